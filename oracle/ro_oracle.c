/*
 * ro_oracle.c -- CPU restatement of radio-observer's STFT / waterfall / bolid-scan
 * hot path.  TEST INFRASTRUCTURE ONLY (see ro_oracle.h): the product never
 * routes through this file.  PARITY UNPINNED (see ro_oracle.h for why).
 *
 * Every function names the reference lines (file:line under /root/reference)
 * whose behaviour it restates.  Nothing here is copied from the reference:
 * the arithmetic is re-derived from SURVEY.md Appendix A and from reading the
 * cited lines.
 */
#include "ro_oracle.h"

#include <dlfcn.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------------- */
/* bin / rate helpers                                                        */
/* ------------------------------------------------------------------------- */

/* src/FFTBackend.cpp:108-109 : overlap is clamped into [0, bins-1]. */
int ro_oracle_clamp_overlap(int bins, int overlap)
{
    if (overlap < 0) return 0;
    if (overlap >= bins) return bins - 1;
    return overlap;
}

/* src/FFTBackend.cpp:150-151 : float / float. */
float ro_oracle_fft_sample_rate(int sample_rate, int bins, int overlap)
{
    int ov = ro_oracle_clamp_overlap(bins, overlap);
    return (float)sample_rate / (float)(bins - ov);
}

/* src/FFTBackend.h:159-178 : the quotient is a float divide, "+ 0.5" promotes
 * to double, the product with (float)bins is double, then truncated. */
int ro_oracle_frequency_to_bin(int bins, int sample_rate, float frequency)
{
    float sr = (float)sample_rate;
    float n = (float)bins;
    float q = frequency / sr;
    double v = (double)n * ((double)q + 0.5);
    int bin = (int)v;
    if (bin < 0) return 0;
    if (bin >= bins) return bins - 1;
    return bin;
}

/* src/FFTBackend.h:134-147 : b/n is float; "-0.5 + ..." is double; the product
 * with sr is double and narrows to float on return. */
float ro_oracle_bin_to_frequency(int bins, int sample_rate, int bin)
{
    float b = (float)bin;
    float sr = (float)sample_rate;
    float n = (float)bins;
    float ratio = b / n;
    double v = (double)sr * (-0.5 + (double)ratio);
    return (float)v;
}

/* src/FFTBackend.h:197-200 : double * float -> double -> int. */
int ro_oracle_time_to_fft_samples(double seconds, float fft_sample_rate)
{
    return (int)(seconds * (double)fft_sample_rate);
}

/* src/WaterfallBackend.h:84-87 with Recorder::getFFTSampleRate() returning int
 * (src/WaterfallBackend.cpp:29-32): the float rate is truncated first. */
int ro_oracle_recorder_fft_samples_to_raw(int rows, float fft_sample_rate, int sample_rate)
{
    int irate = (int)fft_sample_rate;
    double v = ((double)rows / (double)irate) * (double)sample_rate;
    return (int)v;
}

/* src/WaterfallBackend.h:283-287 : float rate kept. */
int ro_oracle_backend_fft_samples_to_raw(int rows, float fft_sample_rate, int sample_rate)
{
    double v = ((double)rows / (double)fft_sample_rate) * (double)sample_rate;
    return (int)v;
}

/* ------------------------------------------------------------------------- */
/* window tables                                                             */
/* ------------------------------------------------------------------------- */

/* src/FFTBackend.cpp:165-184 : a0..a3 are float constants, PI = 4*atan(1.0),
 * the angle is ((k*PI)*(float)i)/(float)(bins-1) in double, sum in double,
 * narrowed to float on store. */
void ro_oracle_window_nuttall(int bins, float *w)
{
    const double pi = 4.0 * atan(1.0);
    const float a0 = 0.355768f, a1 = 0.487396f, a2 = 0.144232f, a3 = 0.012604f;
    const double d = (double)(float)(bins - 1);
    for (int i = 0; i < bins; i++) {
        double fi = (double)(float)i;
        double c1 = cos(2.0 * pi * fi / d);
        double c2 = cos(4.0 * pi * fi / d);
        double c3 = cos(6.0 * pi * fi / d);
        double v = (double)a0 - (double)a1 * c1 + (double)a2 * c2 - (double)a3 * c3;
        w[i] = (float)v;
    }
}

/* src/FFTBackend.cpp:157-163 (commented-out Hann): 0.5*(1 - cos(2*PI*(float)i/(float)(bins-1))). */
void ro_oracle_window_hann(int bins, float *w)
{
    const double pi = 4.0 * atan(1.0);
    const double d = (double)(float)(bins - 1);
    for (int i = 0; i < bins; i++) {
        double fi = (double)(float)i;
        w[i] = (float)(0.5 * (1.0 - cos(2.0 * pi * fi / d)));
    }
}

/* ------------------------------------------------------------------------- */
/* framing                                                                   */
/* ------------------------------------------------------------------------- */

/* src/FFTBackend.cpp:211-212,241-247 : first row after `bins` samples, then
 * one per hop; leftovers shorter than a hop produce nothing. */
int64_t ro_oracle_row_count(int64_t samples, int bins, int overlap)
{
    int ov = ro_oracle_clamp_overlap(bins, overlap);
    int64_t hop = bins - ov;
    if (samples < bins) return 0;
    return (samples - bins) / hop + 1;
}

/* ------------------------------------------------------------------------- */
/* FP64 FFT                                                                  */
/* ------------------------------------------------------------------------- */

/* Forward unnormalised transform X[k] = sum_n x[n] exp(-2 pi i k n / N), the
 * definition of FFTW_FORWARD (src/FFTBackend.cpp:120).  libfftw3 itself is
 * not in this image; any correct FP64 FFT agrees with it to ~1e-15 |x|.
 * Implementation: iterative radix-2 decimation in time, per-size cached
 * twiddle and bit-reversal tables. */
typedef struct {
    int n;
    int log2n;
    double *tw;       /* n/2 entries (re,im) */
    uint32_t *rev;
} fft_plan_t;

#define MAX_PLANS 24
static fft_plan_t g_plans[MAX_PLANS];
static int g_plan_count = 0;

static const fft_plan_t *get_plan(int n)
{
    for (int i = 0; i < g_plan_count; i++)
        if (g_plans[i].n == n) return &g_plans[i];
    if (g_plan_count >= MAX_PLANS) return NULL;
    int l = 0;
    while ((1 << l) < n) l++;
    if ((1 << l) != n || n < 2) return NULL;
    fft_plan_t *p = &g_plans[g_plan_count];
    p->n = n;
    p->log2n = l;
    p->tw = (double *)malloc(sizeof(double) * (size_t)n);
    p->rev = (uint32_t *)malloc(sizeof(uint32_t) * (size_t)n);
    if (!p->tw || !p->rev) return NULL;
    const long double two_pi = 8.0L * atanl(1.0L);
    for (int k = 0; k < n / 2; k++) {
        long double a = -two_pi * (long double)k / (long double)n;
        p->tw[2 * k] = (double)cosl(a);
        p->tw[2 * k + 1] = (double)sinl(a);
    }
    for (int i = 0; i < n; i++) {
        uint32_t r = 0;
        for (int b = 0; b < l; b++)
            if (i & (1 << b)) r |= 1u << (l - 1 - b);
        p->rev[i] = r;
    }
    g_plan_count++;
    return p;
}

/* ---- optional engine: the reference's own FFT library, when the host has it.
 * src/FFTBackend.cpp:117-120 plans fftw_plan_dft_1d(bins, in_, out_, FFTW_FORWARD, FFTW_ESTIMATE) on fftw_malloc'ed
 * arrays and calls fftw_execute per row (:236).  libfftw3 is not vendored with the reference and not installed in
 * this image; ro_oracle_use_fftw() dlopens libfftw3.so.3 if the box running the CPU baseline happens to have it and
 * from then on ro_oracle_fft_f64 executes through it (new-array execute on buffers from ro_oracle_fft_alloc, which
 * are fftw_malloc'ed like the plan's own).  Never required: without the library everything stays on the radix-2
 * transform below, and the two agree to ~1e-15 |x| (tests/test_oracle.py checks that when the library is present). */
typedef void *(*fftw_plan_fn)(int, void *, void *, int, unsigned);
typedef void (*fftw_exec_fn)(void *, void *, void *);
typedef void *(*fftw_malloc_fn)(size_t);
typedef void (*fftw_free_fn)(void *);
static struct {
    void *handle;
    fftw_plan_fn plan_dft_1d;
    fftw_exec_fn execute_dft;
    fftw_malloc_fn malloc_;
    fftw_free_fn free_;
    int on;
    int engine;              /* 1 = libfftw3, 2 = MKL's FFTW3 interface */
    int n[MAX_PLANS];
    void *plan[MAX_PLANS];
    int count;
} g_fftw;

int ro_oracle_use_fftw(int on)
{
    if (!on) { g_fftw.on = 0; return 0; }
    if (!g_fftw.handle) {
        void *h = dlopen("libfftw3.so.3", RTLD_NOW | RTLD_LOCAL);
        if (!h) h = dlopen("libfftw3.so", RTLD_NOW | RTLD_LOCAL);
        g_fftw.engine = 1;
        if (!h) {
            /* second choice: a vendor FFT behind the same fftw3 API (MKL's FFTW3 interface lives in libmkl_rt); not
             * the reference's library, but the API and plan flags are the reference's call sites verbatim.  One
             * thread per call: the baseline's threads are the bench's own. */
            static const char *mkl[] = {"libmkl_rt.so.2", "libmkl_rt.so.1", "libmkl_rt.so", "/opt/conda/lib/libmkl_rt.so.2",
                                        "/opt/conda/lib/libmkl_rt.so.1", "/opt/conda/lib/libmkl_rt.so"};
            for (size_t i = 0; i < sizeof(mkl) / sizeof(mkl[0]) && !h; i++) h = dlopen(mkl[i], RTLD_NOW | RTLD_LOCAL);
            g_fftw.engine = 2;
            if (h) {
                void (*set_threads)(int) = (void (*)(int))dlsym(h, "MKL_Set_Num_Threads");
                if (set_threads) set_threads(1);
            }
        }
        if (!h) { g_fftw.engine = 0; return 0; }
        g_fftw.plan_dft_1d = (fftw_plan_fn)dlsym(h, "fftw_plan_dft_1d");
        g_fftw.execute_dft = (fftw_exec_fn)dlsym(h, "fftw_execute_dft");
        g_fftw.malloc_ = (fftw_malloc_fn)dlsym(h, "fftw_malloc");
        g_fftw.free_ = (fftw_free_fn)dlsym(h, "fftw_free");
        if (!g_fftw.plan_dft_1d || !g_fftw.execute_dft || !g_fftw.malloc_ || !g_fftw.free_) { dlclose(h); g_fftw.engine = 0; return 0; }
        g_fftw.handle = h;
    }
    g_fftw.on = 1;
    return 1;
}

int ro_oracle_fftw_active(void) { return g_fftw.on ? g_fftw.engine : 0; }

/* plan for `bins` (FFTW_FORWARD = -1, FFTW_ESTIMATE = 1 << 6); call once per size before threads share it */
int ro_oracle_fft_prepare(int bins)
{
    if (!g_fftw.on) return get_plan(bins) ? 0 : -1;
    for (int i = 0; i < g_fftw.count; i++)
        if (g_fftw.n[i] == bins) return 0;
    if (g_fftw.count >= MAX_PLANS) return -1;
    void *a = g_fftw.malloc_(sizeof(double) * 2 * (size_t)bins), *b = g_fftw.malloc_(sizeof(double) * 2 * (size_t)bins);
    if (!a || !b) return -1;
    void *pl = g_fftw.plan_dft_1d(bins, a, b, -1, 1u << 6);
    g_fftw.free_(a);
    g_fftw.free_(b);
    if (!pl) return -1;
    g_fftw.n[g_fftw.count] = bins;
    g_fftw.plan[g_fftw.count] = pl;
    g_fftw.count++;
    return 0;
}

/* scratch for ro_oracle_fft_f64: aligned the way the active engine wants it */
static void *fft_alloc(size_t bytes) { return g_fftw.on ? g_fftw.malloc_(bytes) : malloc(bytes); }
static void fft_free(void *p) { if (g_fftw.on) g_fftw.free_(p); else free(p); }

int ro_oracle_fft_f64(int bins, const double *in, double *out);

/* Lengths that are not a power of two (FFTW takes any N, src/FFTBackend.cpp:120; src/BolidRecorder.h:35 suggests
 * 32728): Bluestein's identity n k = (n^2 + k^2 - (k - n)^2) / 2 turns the N-point DFT into a circular convolution
 * of length M >= 2 N - 1 (a power of two), done with the radix-2 transform above, all in double with the chirp
 * exp(-pi i n^2 / N) from long double and its angle reduced exactly (n^2 mod 2N in integers).  Agrees with the
 * O(N^2) long-double DFT below to ~1e-13 of the largest bin (tests/test_oracle.py). */
static int fft_bluestein(int n, const double *in, double *out)
{
    int m = 2;
    while (m < 2 * n - 1) m <<= 1;
    double *c = (double *)malloc(sizeof(double) * 2 * (size_t)n);
    double *a = (double *)calloc(2 * (size_t)m, sizeof(double)), *fa = (double *)malloc(sizeof(double) * 2 * (size_t)m);
    double *b = (double *)calloc(2 * (size_t)m, sizeof(double)), *fb = (double *)malloc(sizeof(double) * 2 * (size_t)m);
    int rc = -2;
    if (c && a && fa && b && fb) {
        const long double pi = 4.0L * atanl(1.0L);
        for (int i = 0; i < n; i++) {
            const long long r = ((long long)i * i) % (2LL * n);
            const long double ang = -pi * (long double)r / (long double)n;
            c[2 * i] = (double)cosl(ang);
            c[2 * i + 1] = (double)sinl(ang);
            a[2 * i] = in[2 * i] * c[2 * i] - in[2 * i + 1] * c[2 * i + 1];
            a[2 * i + 1] = in[2 * i] * c[2 * i + 1] + in[2 * i + 1] * c[2 * i];
            b[2 * i] = c[2 * i];                      /* conj(c), wrapped around m */
            b[2 * i + 1] = -c[2 * i + 1];
            if (i) { b[2 * (m - i)] = b[2 * i]; b[2 * (m - i) + 1] = b[2 * i + 1]; }
        }
        rc = ro_oracle_fft_f64(m, a, fa) | ro_oracle_fft_f64(m, b, fb);
        if (rc == 0) {
            /* y = IFFT(fa fb) = conj(FFT(conj(fa fb))) / m */
            for (int i = 0; i < m; i++) {
                const double re = fa[2 * i] * fb[2 * i] - fa[2 * i + 1] * fb[2 * i + 1];
                const double im = fa[2 * i] * fb[2 * i + 1] + fa[2 * i + 1] * fb[2 * i];
                a[2 * i] = re;
                a[2 * i + 1] = -im;
            }
            rc = ro_oracle_fft_f64(m, a, fa);
            for (int k = 0; k < n && rc == 0; k++) {
                const double yr = fa[2 * k] / m, yi = -fa[2 * k + 1] / m;
                out[2 * k] = yr * c[2 * k] - yi * c[2 * k + 1];
                out[2 * k + 1] = yr * c[2 * k + 1] + yi * c[2 * k];
            }
        }
    }
    free(c); free(a); free(fa); free(b); free(fb);
    return rc;
}

int ro_oracle_fft_f64(int bins, const double *in, double *out)
{
    if (g_fftw.on) {
        if (ro_oracle_fft_prepare(bins) != 0) return -1;
        for (int i = 0; i < g_fftw.count; i++)
            if (g_fftw.n[i] == bins) {
                /* fftw_execute_dft wants the alignment of the planning arrays: callers inside this file pass
                 * fft_alloc'ed buffers; anything else is copied through aligned scratch */
                if ((((uintptr_t)in | (uintptr_t)out) & 63) == 0) {
                    g_fftw.execute_dft(g_fftw.plan[i], (void *)in, out);
                } else {
                    double *a = (double *)g_fftw.malloc_(sizeof(double) * 4 * (size_t)bins);
                    if (!a) return -2;
                    memcpy(a, in, sizeof(double) * 2 * (size_t)bins);
                    g_fftw.execute_dft(g_fftw.plan[i], a, a + 2 * (size_t)bins);
                    memcpy(out, a + 2 * (size_t)bins, sizeof(double) * 2 * (size_t)bins);
                    g_fftw.free_(a);
                }
                return 0;
            }
        return -1;
    }
    if (bins >= 2 && (bins & (bins - 1)) != 0) return fft_bluestein(bins, in, out);
    const fft_plan_t *p = get_plan(bins);
    if (!p) return -1;
    const int n = bins;
    for (int i = 0; i < n; i++) {
        uint32_t r = p->rev[i];
        out[2 * r] = in[2 * i];
        out[2 * r + 1] = in[2 * i + 1];
    }
    for (int half = 1; half < n; half <<= 1) {
        const int step = n / (2 * half);
        for (int base = 0; base < n; base += 2 * half) {
            for (int j = 0; j < half; j++) {
                const double wr = p->tw[2 * j * step];
                const double wi = p->tw[2 * j * step + 1];
                double *a = out + 2 * (base + j);
                double *b = out + 2 * (base + j + half);
                const double tr = b[0] * wr - b[1] * wi;
                const double ti = b[0] * wi + b[1] * wr;
                b[0] = a[0] - tr;
                b[1] = a[1] - ti;
                a[0] += tr;
                a[1] += ti;
            }
        }
    }
    return 0;
}

void ro_oracle_dft_direct(int bins, const double *in, double *out)
{
    const long double two_pi = 8.0L * atanl(1.0L);
    for (int k = 0; k < bins; k++) {
        long double sr = 0.0L, si = 0.0L;
        for (int n = 0; n < bins; n++) {
            long long m = ((long long)k * n) % bins;
            long double a = -two_pi * (long double)m / (long double)bins;
            long double c = cosl(a), s = sinl(a);
            sr += (long double)in[2 * n] * c - (long double)in[2 * n + 1] * s;
            si += (long double)in[2 * n] * s + (long double)in[2 * n + 1] * c;
        }
        out[2 * k] = (double)sr;
        out[2 * k + 1] = (double)si;
    }
}

/* ------------------------------------------------------------------------- */
/* one row                                                                   */
/* ------------------------------------------------------------------------- */

/* gain:   src/FFTBackend.cpp:78-79   (imag += gain, phase shift 0)
 * window: src/FFTBackend.cpp:229-232 (double sample * float coefficient)
 * abs:    src/WaterfallBackend.cpp:492-505 (double sqrt, narrowed; fft-shift) */
static int row_from_window(int bins, const double *win /* gain already applied */,
                           const float *w, float *row, double *spectrum,
                           double *scratch_in, double *scratch_out)
{
    for (int i = 0; i < bins; i++) {
        scratch_in[2 * i] = win[2 * i] * (double)w[i];
        scratch_in[2 * i + 1] = win[2 * i + 1] * (double)w[i];
    }
    if (ro_oracle_fft_f64(bins, scratch_in, scratch_out) != 0) return -1;
    const int half = bins / 2;
    for (int k = 0; k < bins; k++) {
        double re = scratch_out[2 * k], im = scratch_out[2 * k + 1];
        float m = (float)sqrt(re * re + im * im);
        int col = (k < half) ? (half + k) : (k - half);
        row[col] = m;
    }
    if (spectrum) memcpy(spectrum, scratch_out, sizeof(double) * 2 * (size_t)bins);
    return 0;
}

int ro_oracle_row(int bins, const double *iq, const float *w, double gain,
                  float *row, double *spectrum)
{
    double *buf = (double *)fft_alloc(sizeof(double) * 6 * (size_t)bins);
    if (!buf) return -2;
    double *win = buf, *in = buf + 2 * (size_t)bins, *out = buf + 4 * (size_t)bins;
    for (int i = 0; i < bins; i++) {
        win[2 * i] = iq[2 * i];
        win[2 * i + 1] = iq[2 * i + 1] + gain;
    }
    int rc = row_from_window(bins, win, w, row, spectrum, in, out);
    fft_free(buf);
    return rc;
}

int64_t ro_oracle_stft(const double *iq, int64_t samples, int bins, int overlap,
                       const float *w, double gain,
                       int64_t first_row, int64_t max_rows, float *rows)
{
    const int ov = ro_oracle_clamp_overlap(bins, overlap);
    const int64_t hop = bins - ov;
    const int64_t total = ro_oracle_row_count(samples, bins, ov);
    if (first_row < 0) return -1;
    int64_t count = total - first_row;
    if (count < 0) count = 0;
    if (count > max_rows) count = max_rows;
    double *buf = (double *)fft_alloc(sizeof(double) * 6 * (size_t)bins);
    if (!buf) return -2;
    double *win = buf, *in = buf + 2 * (size_t)bins, *out = buf + 4 * (size_t)bins;
    for (int64_t r = 0; r < count; r++) {
        const double *src = iq + 2 * (first_row + r) * hop;
        for (int i = 0; i < bins; i++) {
            win[2 * i] = src[2 * i];
            win[2 * i + 1] = src[2 * i + 1] + gain;
        }
        if (row_from_window(bins, win, w, rows + r * (int64_t)bins, NULL, in, out) != 0) {
            fft_free(buf);
            return -1;
        }
    }
    fft_free(buf);
    return count;
}

int64_t ro_oracle_stft_f32(const float *iq, int64_t samples, int bins, int overlap,
                           const float *w, double gain,
                           int64_t first_row, int64_t max_rows, float *rows)
{
    const int ov = ro_oracle_clamp_overlap(bins, overlap);
    const int64_t hop = bins - ov;
    const int64_t total = ro_oracle_row_count(samples, bins, ov);
    if (first_row < 0) return -1;
    int64_t count = total - first_row;
    if (count < 0) count = 0;
    if (count > max_rows) count = max_rows;
    double *buf = (double *)fft_alloc(sizeof(double) * 6 * (size_t)bins);
    if (!buf) return -2;
    double *win = buf, *in = buf + 2 * (size_t)bins, *out = buf + 4 * (size_t)bins;
    for (int64_t r = 0; r < count; r++) {
        const float *src = iq + 2 * (first_row + r) * hop;
        /* src/RawStream.cpp:61-62 : float32 widened to the double Complex */
        for (int i = 0; i < bins; i++) {
            win[2 * i] = (double)src[2 * i];
            win[2 * i + 1] = (double)src[2 * i + 1] + gain;
        }
        if (row_from_window(bins, win, w, rows + r * (int64_t)bins, NULL, in, out) != 0) {
            fft_free(buf);
            return -1;
        }
    }
    fft_free(buf);
    return count;
}

/* ------------------------------------------------------------------------- */
/* WFTime                                                                    */
/* ------------------------------------------------------------------------- */

#define USEC_PER_SEC 1000000L

/* src/WFTime.h:92-103 (WFTime::add). */
void ro_oracle_wftime_add(int64_t sec, int64_t usec, int64_t add_sec, int64_t add_usec,
                          int64_t *out_sec, int64_t *out_usec)
{
    int64_t us = usec + add_usec % USEC_PER_SEC;
    int64_t s = sec + add_sec;
    s += add_usec / USEC_PER_SEC;
    s += us / USEC_PER_SEC;
    us %= USEC_PER_SEC;
    *out_sec = s;
    *out_usec = us;
}

/* src/WFTime.h:105-114 (WFTime::addSamples): whole seconds by integer divide,
 * the remainder through double and TRUNCATED to microseconds. */
void ro_oracle_wftime_add_samples(int64_t sec, int64_t usec, uint64_t count, int rate,
                                  int64_t *out_sec, int64_t *out_usec)
{
    uint64_t whole = count / (uint64_t)rate;
    uint64_t rem = count % (uint64_t)rate;
    long micro = (long)(((double)rem / (double)rate) * (double)USEC_PER_SEC);
    ro_oracle_wftime_add(sec, usec, (int64_t)whole, (int64_t)micro, out_sec, out_usec);
}

/* ------------------------------------------------------------------------- */
/* streaming emulation                                                       */
/* ------------------------------------------------------------------------- */

typedef struct { int mark; int64_t sec, usec; } raw_handle_t;

struct ro_oracle_stream {
    int bins, overlap, sample_rate;
    double gain;
    float *w;
    double *window;        /* window_  : bins complex                       */
    raw_handle_t *handles; /* windowRaw_                                    */
    int in_mark;           /* inMark_ - window_                             */
    /* Frontend state (src/Frontend.cpp:16-52) */
    int64_t start_sec, start_usec;
    uint64_t data_offset;
    /* FFTBackend::info_ */
    uint64_t row_offset;
    /* raw IQ ring bookkeeping (RingBuffer2D<float>(2, 1 MiB, n), src/FFTBackend.h:129-132) */
    int raw_capacity;
    int raw_head;
    double *scratch;
};

ro_oracle_stream_t *ro_oracle_stream_create(int bins, int overlap, int sample_rate,
                                            int64_t start_sec, int64_t start_usec,
                                            double gain, int raw_capacity_rows)
{
    if (bins < 2 || (!(bins & (bins - 1)) && !get_plan(bins))) return NULL;     /* (other lengths: fft_bluestein) */
    ro_oracle_stream_t *s = (ro_oracle_stream_t *)calloc(1, sizeof(*s));
    if (!s) return NULL;
    s->bins = bins;
    s->overlap = ro_oracle_clamp_overlap(bins, overlap);
    s->sample_rate = sample_rate;
    s->gain = gain;
    s->w = (float *)malloc(sizeof(float) * (size_t)bins);
    s->window = (double *)calloc(2 * (size_t)bins, sizeof(double));
    s->handles = (raw_handle_t *)calloc((size_t)bins, sizeof(raw_handle_t));
    s->scratch = (double *)malloc(sizeof(double) * 4 * (size_t)bins);
    ro_oracle_window_nuttall(bins, s->w);
    s->start_sec = start_sec;
    s->start_usec = start_usec;
    /* raw ring: width 2 floats, 1 MiB chunks -> chunkRows = 131072, capacity rounded up
     * to whole chunks (src/RingBuffer.h:428-457). */
    {
        int chunk_rows = (1024 * 1024) / 8;
        int want = raw_capacity_rows > 0 ? raw_capacity_rows : 1;
        int chunks = want / chunk_rows + ((want % chunk_rows) ? 1 : 0);
        s->raw_capacity = chunks * chunk_rows;
    }
    return s;
}

void ro_oracle_stream_destroy(ro_oracle_stream_t *s)
{
    if (!s) return;
    free(s->w);
    free(s->window);
    free(s->handles);
    free(s->scratch);
    free(s);
}

int ro_oracle_stream_pending(const ro_oracle_stream_t *s) { return s->in_mark; }

/* copy `count` samples into window_ at in_mark with the gain correction
 * (src/FFTBackend.cpp:216, :78-79) and record raw handles (:217-223). */
static void stream_take(ro_oracle_stream_t *s, const double *src, int count,
                        int64_t tsec, int64_t tusec)
{
    for (int i = 0; i < count; i++) {
        int pos = s->in_mark + i;
        s->window[2 * pos] = src[2 * i];
        s->window[2 * pos + 1] = src[2 * i + 1] + s->gain;
        s->raw_head = (s->raw_head + 1) % s->raw_capacity;   /* push(), then mark() */
        s->handles[pos].mark = s->raw_head;
        ro_oracle_wftime_add_samples(tsec, tusec, (uint64_t)i, s->sample_rate,
                                     &s->handles[pos].sec, &s->handles[pos].usec);
    }
}

int ro_oracle_stream_process(ro_oracle_stream_t *s, const double *iq, int n,
                             float *rows_out, ro_oracle_row_info_t *info_out, int max_rows)
{
    /* Frontend::process hands the backend dataInfo_ (src/Frontend.cpp:41-45) whose
     * timeOffset was computed after the previous call (:47-51). */
    int64_t tsec, tusec;
    ro_oracle_wftime_add_samples(s->start_sec, s->start_usec, s->data_offset, s->sample_rate,
                                 &tsec, &tusec);
    if (s->data_offset == 0) { tsec = s->start_sec; tusec = s->start_usec; }

    const int bins = s->bins, ov = s->overlap;
    int size = n;
    const double *src = iq;
    int produced = 0;

    while (size >= bins - s->in_mark) {                    /* src/FFTBackend.cpp:211 */
        int count = bins - s->in_mark;
        stream_take(s, src, count, tsec, tusec);
        int64_t rsec = s->handles[0].sec, rusec = s->handles[0].usec;   /* :225 */

        if (produced >= max_rows) return -3;
        if (row_from_window(bins, s->window, s->w, rows_out + (size_t)produced * bins, NULL,
                            s->scratch, s->scratch + 2 * (size_t)bins) != 0)
            return -1;

        /* overlap kept at the front (:241-242) */
        memmove(s->window, s->window + 2 * (size_t)(bins - ov), sizeof(double) * 2 * (size_t)ov);
        memmove(s->handles, s->handles + (bins - ov), sizeof(raw_handle_t) * (size_t)ov);
        s->in_mark = ov;
        size -= count;
        src += 2 * (size_t)count;

        info_out[produced].offset = s->row_offset;
        info_out[produced].time_sec = rsec;
        info_out[produced].time_usec = rusec;
        info_out[produced].raw_mark = s->handles[0].mark;               /* :251 */
        produced++;

        ro_oracle_wftime_add_samples(tsec, tusec, (uint64_t)count, s->sample_rate,
                                     &tsec, &tusec);                   /* :255 */
        s->row_offset++;                                                /* :256 */
    }
    if (size > 0) {                                                     /* :261-273 */
        stream_take(s, src, size, tsec, tusec);
        s->in_mark += size;
    }
    s->data_offset += (uint64_t)n;                                      /* Frontend.cpp:47 */
    return produced;
}

/* ------------------------------------------------------------------------- */
/* BolidRecorder scans                                                       */
/* ------------------------------------------------------------------------- */

/* src/BolidRecorder.cpp:302-310 : the comparator subtracts in float and looks
 * at the sign; unordered (NaN) compares as equal. */
static int cmp_float_by_difference(const void *pa, const void *pb)
{
    float d = *(const float *)pa - *(const float *)pb;
    if (d > 0.0f) return 1;
    if (d < 0.0f) return -1;
    return 0;
}

/* src/BolidRecorder.cpp:313-320 : ascending sort, element floor(len/4), times 2
 * (double product narrowed to float: exact). */
float ro_oracle_noise(float *buffer, int length)
{
    qsort(buffer, (size_t)length, sizeof(float), cmp_float_by_difference);
    return (float)((double)buffer[length / 4] * 2.0);
}

/* src/BolidRecorder.cpp:323-335 : ">=" keeps the LAST maximal index. */
int ro_oracle_peak(const float *buffer, int length)
{
    int best = 0;
    for (int i = 0; i < length; i++)
        if (buffer[i] >= buffer[best]) best = i;
    return best;
}

/* src/BolidRecorder.cpp:338-347 : double accumulation in index order, one
 * double divide, narrowed. */
float ro_oracle_average(const float *buffer, int length)
{
    double acc = 0.0;
    for (int i = 0; i < length; i++) acc += (double)buffer[i];
    return (float)(acc / (double)length);
}

void ro_oracle_scan_row(const float *row, int low_noise, int noise_width,
                        int low_detect, int detect_width, int avg_bins,
                        ro_oracle_scan_t *out)
{
    float *copy = (float *)malloc(sizeof(float) * (size_t)(noise_width > 0 ? noise_width : 1));
    memcpy(copy, row + low_noise, sizeof(float) * (size_t)noise_width);     /* :123 */
    out->noise = ro_oracle_noise(copy, noise_width);                          /* :124 */
    free(copy);
    out->peak = ro_oracle_peak(row + low_detect, detect_width);               /* :125 */
    out->average = ro_oracle_average(row + low_detect + out->peak - avg_bins / 2, avg_bins); /* :126-132 */
}

/* src/BolidRecorder.cpp:80-104 */
void ro_oracle_bolid_bands(int bins, int sample_rate, float fft_sample_rate,
                           float min_detect_fq, float max_detect_fq,
                           float min_noise_fq, float max_noise_fq,
                           double advance_time, double jitter_time,
                           float avg_freq_range, double noise_metadata_time,
                           ro_oracle_bands_t *out)
{
    /* the constructor orders the detect frequencies (src/BolidRecorder.h:161) */
    if (min_detect_fq > max_detect_fq) { float t = min_detect_fq; min_detect_fq = max_detect_fq; max_detect_fq = t; }
    int lo = ro_oracle_frequency_to_bin(bins, sample_rate, min_detect_fq);
    int hi = ro_oracle_frequency_to_bin(bins, sample_rate, max_detect_fq);
    if (lo > hi) { int t = lo; lo = hi; hi = t; }
    out->low_detect = lo;
    out->detect_width = hi - lo;
    lo = ro_oracle_frequency_to_bin(bins, sample_rate, min_noise_fq);
    hi = ro_oracle_frequency_to_bin(bins, sample_rate, max_noise_fq);
    out->low_noise = lo < hi ? lo : hi;
    out->noise_width = (lo < hi ? hi : lo) - out->low_noise;
    out->advance = ro_oracle_time_to_fft_samples(advance_time, fft_sample_rate);
    out->jitter = ro_oracle_time_to_fft_samples(jitter_time, fft_sample_rate);
    out->avg_bins = ro_oracle_frequency_to_bin(bins, sample_rate, avg_freq_range)
                  - ro_oracle_frequency_to_bin(bins, sample_rate, 0.0f);
    out->noise_metadata_rows = ro_oracle_time_to_fft_samples(noise_metadata_time, fft_sample_rate);
}

/* ------------------------------------------------------------------------- */
/* BolidRecorder FSM                                                         */
/* ------------------------------------------------------------------------- */

void ro_oracle_fsm_init(ro_oracle_fsm_t *f, int advance, int jitter, float fft_sample_rate,
                        int sample_rate, float min_detect_fq, float max_detect_fq)
{
    memset(f, 0, sizeof(*f));
    f->state = RO_ORACLE_STATE_INIT;       /* src/BolidRecorder.cpp:106-108 */
    f->advance = advance;
    f->jitter = jitter;
    f->fft_sample_rate = fft_sample_rate;
    f->sample_rate = sample_rate;
    if (min_detect_fq > max_detect_fq) { float t = min_detect_fq; min_detect_fq = max_detect_fq; max_detect_fq = t; }
    f->min_detect_fq = min_detect_fq;
    f->max_detect_fq = max_detect_fq;
}

void ro_oracle_fsm_update(ro_oracle_fsm_t *f, float n, float a, float peak_fq, int mark,
                          ro_oracle_event_t *ev)
{
    memset(ev, 0, sizeof(*ev));
    const int detect = ((double)a > (double)n * 2.0);          /* :135 */
    switch (f->state) {
    case RO_ORACLE_STATE_INIT:                                  /* :172-183 */
        if (detect) {
            f->peak_freq = peak_fq;
            f->noise = n;
            f->magnitude = a;
            f->duration = 1;
            f->snap_start = mark - f->advance;
            f->snap_length = 2 * f->advance;
            f->state = RO_ORACLE_STATE_BOLID;
        }
        break;
    case RO_ORACLE_STATE_BOLID:                                 /* :185-193 */
        if (detect) {
            f->duration += 1;
        } else {
            f->snap_length += f->duration;
            f->duration = 1;
            f->state = RO_ORACLE_STATE_BOLID_ENDED;
        }
        break;
    case RO_ORACLE_STATE_BOLID_ENDED:                           /* :195-267 */
        f->duration += 1;
        if (detect) {
            f->state = RO_ORACLE_STATE_BOLID;
        } else if (f->duration >= f->jitter) {
            ev->fired = 1;
            ev->snap_start = f->snap_start;
            ev->snap_length = f->snap_length;
            ev->duration_s = (float)(f->snap_length - 2 * f->advance) / f->fft_sample_rate;  /* :209 */
            ev->noise = f->noise;
            ev->peak_freq = f->peak_freq;
            ev->magnitude = f->magnitude;
            {
                float quarter = (f->max_detect_fq - f->min_detect_fq) / 4;                    /* :241-242 */
                ev->fmin = f->peak_freq - quarter;
                ev->fmax = f->peak_freq + quarter;
            }
            ev->raw_length = ro_oracle_recorder_fft_samples_to_raw(f->snap_length, f->fft_sample_rate,
                                                                   f->sample_rate);          /* :246 */
            f->state = RO_ORACLE_STATE_INIT;
        }
        break;
    default:
        f->state = RO_ORACLE_STATE_INIT;
        break;
    }
}

/* ------------------------------------------------------------------------- */
/* RingBuffer2D bookkeeping                                                  */
/* ------------------------------------------------------------------------- */

typedef struct { int start, end, alive, dirty; } ring_res_t;

struct ro_oracle_ring2d {
    int width, chunk_rows, chunk_count, capacity;
    int head, size;
    ring_res_t *res;
    int res_count, res_cap;
    int *free_list;
    int free_count, free_cap;
};

/* src/RingBuffer.h:428-457 */
ro_oracle_ring2d_t *ro_oracle_ring2d_create(int elem_size, int width, int chunk_bytes, int capacity)
{
    ro_oracle_ring2d_t *r = (ro_oracle_ring2d_t *)calloc(1, sizeof(*r));
    if (!r) return NULL;
    int row_bytes = elem_size * width;
    r->width = width;
    r->chunk_rows = chunk_bytes / row_bytes + ((chunk_bytes % row_bytes) ? 1 : 0);
    if (capacity >= 0) {
        r->chunk_count = capacity / r->chunk_rows + ((capacity % r->chunk_rows) > 0 ? 1 : 0);
        r->capacity = r->chunk_count * r->chunk_rows;
    }
    return r;
}

void ro_oracle_ring2d_destroy(ro_oracle_ring2d_t *r)
{
    if (!r) return;
    free(r->res);
    free(r->free_list);
    free(r);
}

int ro_oracle_ring2d_capacity(const ro_oracle_ring2d_t *r) { return r->capacity; }
int ro_oracle_ring2d_chunk_rows(const ro_oracle_ring2d_t *r) { return r->chunk_rows; }
int ro_oracle_ring2d_get_size(const ro_oracle_ring2d_t *r) { return r->size; }
int ro_oracle_ring2d_is_full(const ro_oracle_ring2d_t *r)
{
    return (r->size >= r->capacity) && (r->capacity > 0);       /* :412-415 */
}

/* src/RingBuffer.h:360-369 */
int ro_oracle_ring2d_normalize(const ro_oracle_ring2d_t *r, int mark)
{
    while (mark < 0) mark += r->capacity;
    return mark % r->capacity;
}

static int ring_in_range(const ro_oracle_ring2d_t *r, int index, int start, int end)
{
    index = ro_oracle_ring2d_normalize(r, index);               /* :562-573 */
    start = ro_oracle_ring2d_normalize(r, start);
    end = ro_oracle_ring2d_normalize(r, end);
    if (end > start) return (index >= start) && (index < end);
    return (index >= start) || (index < end);
}

/* src/RingBuffer.h:482-496 */
int ro_oracle_ring2d_push(ro_oracle_ring2d_t *r)
{
    int written = r->head;
    r->head = (r->head + 1) % r->capacity;
    if (!ro_oracle_ring2d_is_full(r)) r->size++;
    for (int i = 0; i < r->res_count; i++)
        if (r->res[i].alive && ring_in_range(r, r->head, r->res[i].start, r->res[i].end))
            r->res[i].dirty = 1;
    return written;
}

int ro_oracle_ring2d_mark(const ro_oracle_ring2d_t *r) { return r->head; }   /* :505-509 */

/* src/RingBuffer.h:543-551 : equal indices give the full capacity. */
int ro_oracle_ring2d_size_between(const ro_oracle_ring2d_t *r, int start, int end)
{
    start = ro_oracle_ring2d_normalize(r, start);
    end = ro_oracle_ring2d_normalize(r, end);
    if (end > start) return end - start;
    return (r->capacity - start) + end;
}

int ro_oracle_ring2d_size_from(const ro_oracle_ring2d_t *r, int start)       /* :555-560 */
{
    return ro_oracle_ring2d_size_between(r, start, r->head);
}

/* src/RingBuffer.h:583-601, with Reservation::init storing end = 0 (:524-529). */
int ro_oracle_ring2d_reserve(ro_oracle_ring2d_t *r, int start, int end)
{
    int handle;
    start = ro_oracle_ring2d_normalize(r, start);
    end = ro_oracle_ring2d_normalize(r, end);
    (void)end;
    if (r->free_count > 0) {
        handle = r->free_list[--r->free_count];
    } else {
        if (r->res_count == r->res_cap) {
            r->res_cap = r->res_cap ? 2 * r->res_cap : 8;
            r->res = (ring_res_t *)realloc(r->res, sizeof(ring_res_t) * (size_t)r->res_cap);
        }
        handle = r->res_count++;
    }
    r->res[handle].start = start;
    r->res[handle].end = 0;
    r->res[handle].alive = 1;
    r->res[handle].dirty = 0;
    return handle;
}

int ro_oracle_ring2d_free_reservation(ro_oracle_ring2d_t *r, int handle)    /* :610-616 */
{
    if (handle < 0 || handle >= r->res_count) return 0;
    r->res[handle].alive = 0;
    if (r->free_count == r->free_cap) {
        r->free_cap = r->free_cap ? 2 * r->free_cap : 8;
        r->free_list = (int *)realloc(r->free_list, sizeof(int) * (size_t)r->free_cap);
    }
    r->free_list[r->free_count++] = handle;
    return 1;
}

int ro_oracle_ring2d_is_dirty(const ro_oracle_ring2d_t *r, int handle)      /* :617-620 */
{
    if (handle < 0 || handle >= r->res_count) return -1;
    return r->res[handle].dirty;
}

/* ------------------------------------------------------------------------- */
/* offline ln transform                                                      */
/* ------------------------------------------------------------------------- */

/* fits2png:46 takes numpy.log of the non-zero float32 pixels; the fused GPU
 * output keeps the row shape, so zeros map to -inf here (the viewer would
 * drop them).  float32 log of a float32 pixel. */
void ro_oracle_ln_rows(const float *rows, int64_t count, float *out)
{
    for (int64_t i = 0; i < count; i++) out[i] = logf(rows[i]);
}

/* The viewer's grey image of one band image (fits2png:476-477, :444-445, :495-497):
 * min / max of the ln over the non-zero pixels, level = (ln - min) / (max - min) * 255
 * in float32, stored into a uint8 array (C truncation).  `image` is rows x cols,
 * contiguous.  Zero pixels and a flat image give level 0 (the viewer has no such
 * pixels / divides by zero there). */
void ro_oracle_ln_levels(const float *image, int64_t count, float *ln_out, uint8_t *u8_out, float *minmax)
{
    float mn = INFINITY, mx = -INFINITY;
    int any = 0;
    for (int64_t i = 0; i < count; i++) {
        const float l = logf(image[i]);
        if (ln_out) ln_out[i] = l;
        if (image[i] != 0.0f) {
            if (l < mn) mn = l;
            if (l > mx) mx = l;
            any = 1;
        }
    }
    (void)any;
    if (minmax) {
        minmax[0] = mn;
        minmax[1] = mx;
    }
    if (!u8_out) return;
    const float span = mx - mn;
    for (int64_t i = 0; i < count; i++) {
        const float l = logf(image[i]);
        const float level = (l - mn) / span * 255.0f;
        u8_out[i] = (image[i] != 0.0f && span > 0.0f) ? (uint8_t)(int)level : (uint8_t)0;
    }
}
