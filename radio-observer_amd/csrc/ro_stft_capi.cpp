// ro_stft_capi.cpp -- implementation of the C ABI in include/ro_stft.h.
//
// Host side of the MI355X STFT path: owns the window / twiddle tables in HBM,
// the streaming staging buffers, and launches the kernels of ro_kernels.hip.
// There is no CPU compute path in this file: rows only ever come out of the
// HIP kernels.
#include <hip/hip_runtime.h>

#include <dlfcn.h>

#include <algorithm>
#include <cmath>
#include <complex>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <ctime>
#include <deque>
#include <string>
#include <mutex>
#include <vector>

#include "../../include/ro_stft.h"
#include "ro_kernels.h"
#include "ro_narrow.h"

// a -DRO_DIAG=1 build (tools/ab_build.sh) reads its run-time knobs (RO_BIG_FORM, RO_F64_SCRATCH_MB) from the environment
#if defined(RO_DIAG) && !defined(RO_DIAG_KNOBS)
#define RO_DIAG_KNOBS 1
#endif

// scratch of the large transforms' scratch form (the folded sub-rows between the kernels), MiB per block
// largest bins / 16384 the one-kernel form of the large transforms is used for (see ro_stft_create)
#ifndef RO_DIF_MAX_DEC
#define RO_DIF_MAX_DEC 4
#endif

#ifndef RO_SPEC_SCRATCH_MB
#define RO_SPEC_SCRATCH_MB 2048
#endif
// the four-step form's scratch (one block of Z between its two kernels), MiB AT MOST: the block grows to what a
// launch asks for (a streaming handle at Ionozor's shape launches a handful of rows and holds a few MiB, not the
// limit).  1 GiB measured best for resident launches -- blocks inside the 256 MiB Infinity Cache were 3 % faster for the
// row kernel and 13 % slower for the column kernel (profiles/r04_fourstep.txt).  Diagnostic builds: RO_FOUR_SCRATCH_MB
#ifndef RO_FOUR_SCRATCH_MB
#define RO_FOUR_SCRATCH_MB 1024
#endif
// RO_PRECISION_F64: MiB per complex-double scratch block (two blocks); the passes of one chunk run back to back, and a
// chunk that stays inside the 256 MiB Infinity Cache keeps most of the trip between them off HBM: 2.75-2.87 x 10^6
// rows/s at the C3 shape with 128 against 2.46 with 512, 2.36 with 256, 2.50 with 64, 1.96 with 32 (too few workgroups
// per launch); the same bits whatever the chunk (profiles/r04_strict_chunk.txt, tools/r4/strict_sweep.py)
#ifndef RO_F64_SCRATCH_MB
#define RO_F64_SCRATCH_MB 128
#endif
// RO_PRECISION_F64_ONE_LAUNCH (ro_f64fused.hip): a ring of this many rows of 32768 bins per XCD (scaled so that the
// ring's bytes stay the same at the other sizes), this many workgroups per CU.  4 rows is the least that keeps an XCD's
// 32 workgroups busy, and all its L2 serves (profiles/r05_f64_one_launch.txt)
// streaming path: sets of device + pinned staging buffers a handle rotates through (batches that can be in flight at once)
#ifndef RO_GRAPH_TIME_EVERY
#define RO_GRAPH_TIME_EVERY 8
#endif
#ifndef RO_STREAM_SLOTS
#define RO_STREAM_SLOTS 3
#endif
// streaming path: a full latency-bound batch with a row sink runs as one captured graph per slot (run_stream_batch)
#ifndef RO_STREAM_GRAPH
#define RO_STREAM_GRAPH 1
#endif
#ifndef RO_F64_RING_ROWS
#define RO_F64_RING_ROWS 4
#endif
#ifndef RO_F64_WGS_PER_CU
#define RO_F64_WGS_PER_CU 1
#endif

namespace {

thread_local std::string g_error;

int fail(int code, const char *fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    g_error = buf;
    return code;
}

#define HIP_TRY(expr)                                                                     \
    do {                                                                                  \
        hipError_t e_ = (expr);                                                           \
        if (e_ != hipSuccess)                                                             \
            return fail(RO_ERR_HIP, "%s failed: %s", #expr, hipGetErrorString(e_));       \
    } while (0)

// window tables: same arithmetic as FFTBackend::startStream (src/FFTBackend.cpp:156-186):
// float coefficients, double pi = 4*atan(1), (float)i and (float)(bins-1) widened to double,
// evaluation in double, one narrowing on store.
void build_window(int kind, int bins, float *w)
{
    const double pi = 4.0 * std::atan(1.0);
    const double denom = (double)(float)(bins - 1);
    if (kind == RO_WINDOW_HANN) {
        for (int i = 0; i < bins; ++i)
            w[i] = (float)(0.5 * (1.0 - std::cos(2.0 * pi * (double)(float)i / denom)));
        return;
    }
    const float a0 = 0.355768f, a1 = 0.487396f, a2 = 0.144232f, a3 = 0.012604f;
    for (int i = 0; i < bins; ++i) {
        const double x = (double)(float)i;
        w[i] = (float)((double)a0 - (double)a1 * std::cos(2.0 * pi * x / denom) +
                       (double)a2 * std::cos(4.0 * pi * x / denom) -
                       (double)a3 * std::cos(6.0 * pi * x / denom));
    }
}

// per-stage twiddle tables in the layout apply_twiddles() reads:
//   stage with radix R after sub-transforms of length NS: entry (r-1)*NS + k = exp(-2 pi i r k / (NS R))
// evaluated in long double and rounded once to float.
std::vector<float2> build_twiddles(int bins)
{
    int radix[4];
    std::vector<float2> tw;
    if (!ro::stft_radices(bins, radix)) return tw;
    const long double two_pi = 8.0L * atanl(1.0L);
    long ns = radix[0];
    for (int s = 1; s < 4; ++s) {
        const int R = radix[s];
        if (R <= 1) continue;
        for (int r = 1; r < R; ++r)
            for (long k = 0; k < ns; ++k) {
                const long double ang = -two_pi * (long double)((long long)r * k) / (long double)(ns * R);
                tw.push_back(make_float2((float)cosl(ang), (float)sinl(ang)));
            }
        ns *= R;
    }
    return tw;
}

// in-place forward FFT of a power-of-two length in double (table preparation only: the chirp-z filter)
void host_fft(std::vector<std::complex<double>> &x)
{
    const size_t n = x.size();
    for (size_t i = 1, j = 0; i < n; ++i) {
        size_t bit = n >> 1;
        for (; j & bit; bit >>= 1) j ^= bit;
        j ^= bit;
        if (i < j) std::swap(x[i], x[j]);
    }
    const long double two_pi = 8.0L * atanl(1.0L);
    for (size_t len = 2; len <= n; len <<= 1) {
        std::vector<std::complex<double>> w(len / 2);
        for (size_t k = 0; k < len / 2; ++k) {
            const long double ang = -two_pi * (long double)k / (long double)len;
            w[k] = std::complex<double>((double)cosl(ang), (double)sinl(ang));
        }
        for (size_t i = 0; i < n; i += len)
            for (size_t k = 0; k < len / 2; ++k) {
                const std::complex<double> u = x[i + k], v = x[i + k + len / 2] * w[k];
                x[i + k] = u + v;
                x[i + k + len / 2] = u - v;
            }
    }
}

// exp(-2 pi i m / N) in double, correctly rounded from long double (strict-precision path)
std::vector<double2> build_full_twiddles_f64(int bins)
{
    std::vector<double2> tw((size_t)bins);
    const long double two_pi = 8.0L * atanl(1.0L);
    for (int m = 0; m < bins; ++m) {
        const long double ang = -two_pi * (long double)m / (long double)bins;
        tw[(size_t)m] = make_double2((double)cosl(ang), (double)sinl(ang));
    }
    return tw;
}

// One launch worth of finished rows on their way to the caller.  The buffers are pinned host
// memory (hipHostMalloc) recycled through a free list; `done` fires when the device-to-host
// copies have landed, so ro_stft_push never waits for the GPU -- only ro_stft_fetch does.
struct Batch {
    int64_t first_row = 0;
    int64_t rows = 0;
    float *data = nullptr;                     // capacity_rows x out_cols, pinned
    float *ln = nullptr;                       // capacity_rows x out_cols (tile_ln), pinned
    float *minmax = nullptr;                   // capacity_rows x 2 (tile_ln), pinned
    ro_scan_record_t *records = nullptr;       // capacity_rows, pinned
    int64_t capacity_rows = 0;
    int64_t consumed = 0;                      // rows already fetched
    hipEvent_t done = nullptr;
    hipEvent_t k0 = nullptr, k1 = nullptr;     // around the kernels of this batch (timing counters)
    bool pending = false;                      // `done` not yet waited for
    bool timed = true;                         // k0 / k1 were recorded around this batch's kernels
};

}  // namespace

struct ro_stft {
    ro_stft_config_t cfg{};
    int bins = 0, overlap = 0, hop = 0;
    int device = 0;
    std::string device_name;
    std::vector<float> window;
    float *d_window = nullptr;
    float *d_window_k = nullptr;       // kernel-order copy (single-pass plans)
    float *d_window_k32 = nullptr;     // ... in the order of the N = 32768 magnitude-row kernel (bins = 32768)
    float2 *d_twiddles = nullptr;
    float4 *d_twiddles_k = nullptr;    // packed copy for the radix-16/32 stages
    hipStream_t stream = nullptr;

    // streaming state.  Three HIP streams and RO_STREAM_SLOTS slots of device buffers: while the kernels of batch n run on
    // `stream`, batch n+1 is uploaded on `s_in` and batch n-1 goes home on `s_out`.
    int batch_rows = 0;
    // Samples are staged where the upload reads them: in the pinned buffer (h_in) of the slot the next batch will use
    // (slot = batch_seq % RO_STREAM_SLOTS), from its first byte.  A batch uploads the front of it and the samples later rows still
    // need -- the overlap and whatever came in behind the batch's last row -- are carried over to the other slot.
    int stage_fmt = RO_IQ_F32;                 // what is staged: RO_IQ_F32 (8 B per sample) or RO_IQ_I16 (4 B)
    bool stage_fmt_set = false;
    size_t  staged_have = 0;                   // live samples at the front of slot[batch_seq % RO_STREAM_SLOTS].h_in
    int64_t stream_sample0 = 0;                // stream index of the first of them
    // row sink (ro_stft_set_row_sink): finished rows go straight into the caller's ring (ro_pinned_alloc memory)
    float  *sink = nullptr;
    int64_t sink_stride = 0, sink_cap = 0, sink_first = 0;
    struct Slot {
        void  *d_iq = nullptr;                 // batch input  ((batch_rows-1)*hop + bins samples, 8 B each at most)
        float *d_rows = nullptr;               // batch output (batch_rows x bins)
        float *d_tile = nullptr;               // batch_rows x tile_cols when a tile is configured
        float *d_ln = nullptr;                 // ... its log and the rows' min / max of it (tile_ln)
        float *d_minmax = nullptr;
        ro_scan_record_t *d_records = nullptr;
        void  *h_in = nullptr;                 // pinned upload staging
        hipEvent_t uploaded = nullptr;         // H2D of this slot done (h_in reusable, kernels may start)
        hipEvent_t staging_free = nullptr;     // what the host waits for before it writes h_in again: `uploaded`, or the `done`
                                               // event of the graphed batch that last used the slot (not owned)
        hipEvent_t computed = nullptr;         // kernels of this slot done (d_iq reusable, D2H may start)
        hipEvent_t drained = nullptr;          // D2H of this slot done (d_rows / d_tile / d_records reusable)
        // latency-bound batches with a row sink (run_stream_batch): upload + kernels of a FULL batch of this slot as one
        // graph on the slot's own stream, captured from the very calls the plain path makes
        hipStream_t    gstream = nullptr;
        hipGraphExec_t gexec = nullptr;
        int            graph_fmt = -1;         // stage format the graph was captured for
        int64_t        uses = 0;               // batches this slot has run (the first one warms every lazy initialisation)
        bool           on_gstream = false;     // the slot's last batch ran on gstream (else on the three chained streams)
    } slot[RO_STREAM_SLOTS];
    bool slots_ready = false;
    hipStream_t s_in = nullptr, s_out = nullptr;
    bool graph_refused = false;                // stream capture of a batch failed once on this runtime: plain path only
    int out_first = 0, out_cols = 0;           // columns of every row that travel to the host (the tile, or all)
    int64_t batch_seq = 0;
    std::vector<Batch *> batch_pool;           // recycled pinned batches
    int64_t rows_emitted = 0;                  // stream index of the next row to compute
    std::deque<Batch *> ready;
    int64_t rows_ready = 0;
    int64_t stat_samples = 0, stat_rows = 0, stat_launches = 0;
    double stat_kernel_ms = 0.0;
    // per-call counters in the spirit of FFTBackend's RunningAverage2 trio (src/FFTBackend.h:86-92, :208-235)
    ro_stft_timing_t timing{};
    double push_ms_sum = 0.0, batch_ms_sum = 0.0, fetch_ms_sum = 0.0;
    int64_t timed_batches = 0, timed_rows = 0; // the batches behind batch_ms_sum (graphed batches are timed one in RO_GRAPH_TIME_EVERY)
    double  last_batch_ms = 0.0;               // ... and the last one's time, the estimate for the ones in between
    int64_t graph_batches = 0;
    int     diag_time_every = 0, diag_done_only = 0, diag_direct = 0;   // (-DRO_DIAG: tools/r5/host_calls_ab.py)
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    unsigned long long *d_stamps = nullptr;    // diagnostic builds (RO_STAMPS) only

    unsigned *d_ln_keys = nullptr;     // 16 pairs of min / max keys of ro_stft_ln_tile_resident, used in turn
    unsigned ln_calls = 0;
    // large transforms (bins > 32768 = dec x sub_bins, decimation in frequency on the N = 32768 plan; see ro_stft_create)
    bool    big = false;
    int     sub_bins = 0, dec = 0;
    float2 *d_tw_combine = nullptr;    // [dec][sub_bins]: the rotations exp(-2 pi i q m / bins)
    bool    dif = false;               // dec <= RO_DIF_MAX_DEC: one kernel sums the row's blocks itself (MODE 3)
    float  *d_window_dif = nullptr;    // ... [dec][sub_bins]: window block r in the sub-plan's kernel order
    float2 *d_dif_tw = nullptr;        // ... exp(-2 pi i j / dec)
    float2 *d_dif_shift = nullptr;     // ... [dec][16]: the bin shift q / dec as the stages' twiddles (StftArgs::dif_shift)
    float2 *d_spec = nullptr;          // ... the folded sub-rows, [spec_rows][dec][sub_bins] float2
    float  *d_ones = nullptr;          // ... a window of ones (the fold has applied the real one)
    int64_t spec_rows = 0;
    float2 *d_spec2 = nullptr;         // complex spectra of a large size: the sub-rows' spectra before they are interleaved
    // bins = 262144, 524288: the magnitude rows as a four-step FFT (ro_fourstep.hip): column kernel, scratch, row kernel
    bool    four = false;
    float  *d_four_window = nullptr;   // the window in the column kernel's order
    float2 *d_four_tw_a = nullptr, *d_four_tw_b = nullptr, *d_four_tw_r = nullptr;     // ro::FourArgs
    float  *d_four_z = nullptr;        // [four_rows][bins] complex
    int64_t four_rows = 0;
    // lengths that are not a power of two (even 258 .. 524286): Bluestein's chirp-z form on an inner handle of the
    // power-of-two length czt_m >= 2 bins - 1 (see ro::CztArgs)
    bool    czt = false;
    int     czt_m = 0;
    ro_stft *inner = nullptr;
    float2 *d_cw = nullptr;            // [bins] window[i] * exp(-pi i i^2 / bins)
    float2 *d_bc = nullptr;            // [czt_m] conj(FFT_M(conj(chirp), wrapped)) / czt_m
    float2 *d_czt_a = nullptr, *d_czt_A = nullptr;     // [czt_rows][czt_m] each
    float  *d_czt_mag = nullptr;                       // [czt_rows][czt_m]
    int64_t czt_rows = 0;

    // tile_ln: partial min / max of the fused epilogue's two tile waves (rows x 4 floats), grown on demand
    float  *d_ln_part = nullptr;
    int64_t ln_part_rows = 0;

    // strict precision (RO_PRECISION_F64): double twiddle table + two complex-double scratch blocks
    bool     f64 = false;
    double2 *d_tw_f64 = nullptr;
    // ... bins 4096 ... 65536: the row in a CU's registers, no scratch (ro_f64reg.hip); its window order and twiddle tables
    bool     f64reg = false;
    float   *d_f64r_window = nullptr;
    double2 *d_f64r_tw[4] = {nullptr, nullptr, nullptr, nullptr};
    double2 *d_scratch_d[2] = {nullptr, nullptr};
    int64_t  scratch_rows_d = 0;
    // ... or, for bins = 16^3 r2, all four passes in one launch with the intermediate in an XCD's L2 (ro_f64fused.hip):
    // 8 rings of f64_ring_rows rows, the launch's control block, and its give-up word mirrored into pinned host memory
    double2  *d_f64_ring = nullptr;
    unsigned *d_f64_ctl = nullptr;
    unsigned *h_f64_err = nullptr;
    int       f64_ring_rows = 0, f64_wgs_per_cu = 0;
    bool      f64_one_launch = false;
};

namespace {

int check_bands(const ro_stft *h, const ro_bands_t &b)
{
    if (b.noise_width <= 0 || b.detect_width <= 0 || b.avg_bins <= 0)
        return fail(RO_ERR_INVALID, "bands: widths and avg_bins must be positive");
    if (b.low_noise < 0 || b.low_noise + b.noise_width > h->bins)
        return fail(RO_ERR_INVALID, "bands: noise band [%d,+%d) outside [0,%d)", b.low_noise,
                    b.noise_width, h->bins);
    if (b.low_detect < 0 || b.low_detect + b.detect_width > h->bins)
        return fail(RO_ERR_INVALID, "bands: detect band [%d,+%d) outside [0,%d)", b.low_detect,
                    b.detect_width, h->bins);
    return RO_OK;
}

int validate_resident(const ro_stft *h, const void *d_iq, int format, int64_t samples,
                      int64_t first_row, int64_t rows, const float *d_rows, int64_t row_stride,
                      const float *d_tile, const ro_scan_record_t *d_records)
{
    if (!h) return fail(RO_ERR_INVALID, "null handle");
    if (format != RO_IQ_F32 && format != RO_IQ_I16)
        return fail(RO_ERR_INVALID, "resident input must be RO_IQ_F32 or RO_IQ_I16 (got %d)", format);
    if (rows < 0 || first_row < 0) return fail(RO_ERR_INVALID, "negative row range");
    if (rows == 0) return RO_OK;
    if (!d_iq) return fail(RO_ERR_INVALID, "null input pointer");
    if (!d_rows) return fail(RO_ERR_INVALID, "d_rows is required (the band tile is cut from the rows)");
    if (d_rows && row_stride < h->bins)
        return fail(RO_ERR_INVALID, "row_stride %lld < bins %d", (long long)row_stride, h->bins);
    if (d_tile && h->cfg.tile_cols <= 0) return fail(RO_ERR_INVALID, "tile output requested but tile_cols == 0");
    if (d_records && !h->cfg.enable_scan) return fail(RO_ERR_INVALID, "records requested but enable_scan == 0");
    if (d_records && !d_rows) return fail(RO_ERR_INVALID, "records need full rows (d_rows)");
    // every workgroup reads samples [r*hop, r*hop + bins): the last one must stay inside the buffer
    const int64_t last = (first_row + rows - 1) * (int64_t)h->hop + h->bins;
    if (last > samples)
        return fail(RO_ERR_INVALID, "rows [%lld,+%lld) need %lld samples, buffer holds %lld",
                    (long long)first_row, (long long)rows, (long long)last, (long long)samples);
    if (rows > (int64_t)0x0fffffff) return fail(RO_ERR_INVALID, "too many rows in one launch");
    return RO_OK;
}

ro::StftArgs make_stft_args(const ro_stft *h, const void *d_iq, int64_t first_row, int64_t rows,
                            float *d_rows, int64_t row_stride, float *d_tile = nullptr,
                            ro_scan_record_t *d_records = nullptr, float *d_ln = nullptr)
{
    ro::StftArgs a{};
    a.iq = d_iq;
    a.window = h->d_window;
    a.window_k = h->d_window_k;
    a.window_k32 = h->d_window_k32;
    a.twiddles = h->d_twiddles;
    a.twiddles_k = h->d_twiddles_k;
    a.rows_out = d_rows;
    a.first_row = first_row;
    a.rows = rows;
    a.row_stride = row_stride;
    a.hop = h->hop;
    a.gain = (float)h->cfg.iq_gain;
    a.stamps = h->d_stamps;
    a.spare_cus = h->cfg.spare_cus_per_xcd;
    // plans with a fused epilogue scan / tile take them here; for the others the caller launches the separate kernels
    if (!h->f64 && ro::stft_fuses_scan(h->bins)) {
        a.records = d_records;
        a.low_noise = h->cfg.bands.low_noise;
        a.noise_width = h->cfg.bands.noise_width;
        a.low_detect = h->cfg.bands.low_detect;
        a.detect_width = h->cfg.bands.detect_width;
        a.avg_bins = h->cfg.bands.avg_bins;
        a.tile_out = d_tile;
        a.tile_first = h->cfg.tile_first_col;
        a.tile_cols = h->cfg.tile_cols;
        a.ln_out = d_tile ? d_ln : nullptr;
        a.ln_part = h->d_ln_part;
    }
    return a;
}

ro::TileArgs make_tile_args(const ro_stft *h, const float *d_rows, int64_t row_stride, int64_t rows,
                            float *d_tile)
{
    ro::TileArgs t{};
    t.rows_in = d_rows;
    t.tile_out = d_tile;
    t.rows = rows;
    t.row_stride = row_stride;
    t.first = h->cfg.tile_first_col;
    t.cols = h->cfg.tile_cols;
    return t;
}

ro::ScanArgs make_scan_args(const ro_stft *h, const float *d_rows, int64_t row_stride, int64_t rows,
                            ro_scan_record_t *d_records)
{
    ro::ScanArgs s{};
    s.rows_in = d_rows;
    s.records = d_records;
    s.rows = rows;
    s.row_stride = row_stride;
    s.bins = h->bins;
    s.low_noise = h->cfg.bands.low_noise;
    s.noise_width = h->cfg.bands.noise_width;
    s.low_detect = h->cfg.bands.low_detect;
    s.detect_width = h->cfg.bands.detect_width;
    s.avg_bins = h->cfg.bands.avg_bins;
    return s;
}

// window -> FFT -> |X| for rows [first_row, +rows): the single-pass kernel, or for bins > 32768
// the one-kernel or the scratch form (ro_stft_create), in chunks that fit the scratch blocks
// d_tile / d_records (either may be null): produced here too, by the transform's own epilogue where the plan fuses them
// (N = 32768), by tile_kernel / scan_kernel behind it otherwise
int launch_transform(ro_stft *h, const void *d_iq, int format, int64_t first_row, int64_t rows, float *d_rows,
                     int64_t row_stride, hipStream_t s, float *d_tile = nullptr, ro_scan_record_t *d_records = nullptr,
                     float *d_ln = nullptr);

// d_ln / d_minmax: the tile's log and the rows' min / max of it (tile_ln); need d_tile
int launch_tile_and_scan(ro_stft *h, const float *d_rows, int64_t row_stride, int64_t rows, float *d_tile,
                         ro_scan_record_t *d_records, hipStream_t s, float *d_ln = nullptr, float *d_minmax = nullptr)
{
    const bool want_ln = d_tile && (d_ln || d_minmax);
    if (!h->f64 && ro::stft_fuses_scan(h->bins)) {                  // tile, log and records written by the transform
        if (want_ln && d_minmax) HIP_TRY(ro::launch_ln_finish(h->d_ln_part, d_minmax, rows, s));
        return RO_OK;
    }
    if (d_tile) HIP_TRY(ro::launch_tile(make_tile_args(h, d_rows, row_stride, rows, d_tile), s));
    if (want_ln) HIP_TRY(ro::launch_ln_rows(d_tile, d_ln, d_minmax, rows, h->cfg.tile_cols, s));
    if (d_records) HIP_TRY(ro::launch_scan(make_scan_args(h, d_rows, row_stride, rows, d_records), s));
    return RO_OK;
}

// scratch of the fused log: the two tile waves' partial min / max, 4 floats per row of the launch
int ensure_ln_part(ro_stft *h, int64_t rows)
{
    if (rows <= h->ln_part_rows) return RO_OK;
    // sized generously the first time (16 bytes per row) and doubled after that, so that the device-wide wait a
    // regrow needs -- an earlier launch may still be writing the old block -- happens at most a few times per handle
    int64_t want = std::max<int64_t>(rows, std::max<int64_t>(65536, 2 * h->ln_part_rows));
    if (h->d_ln_part) { (void)hipDeviceSynchronize(); (void)hipFree(h->d_ln_part); h->d_ln_part = nullptr; h->ln_part_rows = 0; }
    HIP_TRY(hipMalloc(&h->d_ln_part, (size_t)want * 4 * sizeof(float)));
    h->ln_part_rows = want;
    return RO_OK;
}

// scratch of the large sizes' scratch form (and of their complex spectra): rows per chunk and the blocks
int ensure_big_scratch(ro_stft *h)
{
    if (!h->spec_rows) {
        h->spec_rows = std::max<int64_t>(1, ((int64_t)RO_SPEC_SCRATCH_MB << 20) / ((int64_t)h->bins * 8));
        if (h->spec_rows > 65535) h->spec_rows = 65535;
    }
    if (!h->d_spec) HIP_TRY(hipMalloc(&h->d_spec, (size_t)h->spec_rows * h->bins * sizeof(float2)));
    if (!h->d_spec2) HIP_TRY(hipMalloc(&h->d_spec2, (size_t)h->spec_rows * h->bins * sizeof(float2)));
    return RO_OK;
}

// complex spectra of rows [first_row, +rows) of a large size (bins = dec x 32768), bin k at element k of each row:
// fold_kernel, the N = 32768 kernel in spectra mode on its rows, interleave2_kernel
int launch_spectra_big(ro_stft *h, const void *d_iq, int format, int64_t first_row, int64_t rows, float2 *d_out,
                       int64_t out_stride, hipStream_t s)
{
    int rc = ensure_big_scratch(h);
    if (rc != RO_OK) return rc;
    for (int64_t done = 0; done < rows; done += h->spec_rows) {
        const int64_t n = std::min(h->spec_rows, rows - done);
        ro::FoldArgs f{};
        f.iq = d_iq;
        f.window = h->d_window;
        f.rot = h->d_tw_combine;
        f.out = h->d_spec;
        f.first_row = first_row + done;
        f.rows = n;
        f.hop = h->hop;
        f.m = h->sub_bins;
        f.dec = h->dec;
        f.gain = (float)h->cfg.iq_gain;
        HIP_TRY(ro::launch_fold(format, f, s));
        ro::StftArgs a = make_stft_args(h, h->d_spec, 0, n * h->dec, nullptr, 0);
        a.window = h->d_ones;
        a.window_k = h->d_ones;
        a.window_k32 = h->d_ones;
        a.hop = h->sub_bins;
        a.gain = 0.0f;
        a.spec_out = h->d_spec2;
        a.spec_stride = h->sub_bins;
        HIP_TRY(ro::launch_stft(h->sub_bins, RO_FMT_F32, a, s));
        ro::Interleave2Args t{};
        t.in = h->d_spec2;
        t.out = d_out + done * out_stride;
        t.rows = n;
        t.out_stride = out_stride;
        t.m = h->sub_bins;
        t.dec = h->dec;
        HIP_TRY(ro::launch_interleave2(t, s));
    }
    return RO_OK;
}

// a length that is not a power of two: chirp-z on the inner handle (see ro::CztArgs), in chunks that fit the scratch
int launch_transform_czt(ro_stft *h, const void *d_iq, int format, int64_t first_row, int64_t rows, float *d_rows,
                         int64_t row_stride, hipStream_t s)
{
    ro_stft *in = h->inner;
    const int M = h->czt_m;
    if (!h->d_czt_mag) {                                            // (each block on its own: a failed call can be retried)
        h->czt_rows = std::min<int64_t>(65535, std::max<int64_t>(1, ((int64_t)1 << 30) / ((int64_t)M * 8)));
        if (!h->d_czt_a) HIP_TRY(hipMalloc(&h->d_czt_a, (size_t)h->czt_rows * M * sizeof(float2)));
        if (!h->d_czt_A) HIP_TRY(hipMalloc(&h->d_czt_A, (size_t)h->czt_rows * M * sizeof(float2)));
        HIP_TRY(hipMalloc(&h->d_czt_mag, (size_t)h->czt_rows * M * sizeof(float)));
    }
    for (int64_t done = 0; done < rows; done += h->czt_rows) {
        const int64_t n = std::min(h->czt_rows, rows - done);
        ro::CztArgs c{};
        c.iq = d_iq;
        c.cw = h->d_cw;
        c.bc = h->d_bc;
        c.a = h->d_czt_a;
        c.first_row = first_row + done;
        c.rows = n;
        c.row_stride = row_stride;
        c.hop = h->hop;
        c.n = h->bins;
        c.m = M;
        c.gain = (float)h->cfg.iq_gain;
        HIP_TRY(ro::launch_czt_pre(format, c, s));
        // A = FFT_M(a): the inner handle's rows are the M-sample blocks of d_czt_a (overlap 0, a window of ones)
        if (!in->big) {
            ro::StftArgs a = make_stft_args(in, h->d_czt_a, 0, n, nullptr, 0);
            a.spec_out = h->d_czt_A;
            a.spec_stride = M;
            HIP_TRY(ro::launch_stft(M, RO_FMT_F32, a, s));
        } else {
            int rc = launch_spectra_big(in, h->d_czt_a, RO_FMT_F32, 0, n, h->d_czt_A, M, s);
            if (rc != RO_OK) return rc;
        }
        c.a = h->d_czt_A;
        HIP_TRY(ro::launch_czt_mul(c, s));                          // conj(A B) / M, in place
        int rc = launch_transform(in, h->d_czt_A, RO_FMT_F32, 0, n, h->d_czt_mag, M, s, nullptr, nullptr, nullptr);
        if (rc != RO_OK) return rc;
        c.mag = h->d_czt_mag;
        c.rows_out = d_rows + done * row_stride;
        HIP_TRY(ro::launch_czt_out(c, s));
    }
    return RO_OK;
}

// RO_PRECISION_F64 at bins = 16^3 r2 (8192 ... 65536): one persistent launch, the complex-double intermediate of a row
// in the L2 of the XCD that makes it (ro_f64fused.hip).  The launch's give-up word travels to pinned host memory behind
// the kernel; a launch that gave up is reported by the NEXT call on the handle (and by ro_stft_destroy's caller never:
// tests and bench compare rows).
int launch_transform_f64_fused(ro_stft *h, const void *d_iq, int format, int64_t first_row, int64_t rows, float *d_rows,
                               int64_t row_stride, hipStream_t s)
{
    if (h->h_f64_err && *h->h_f64_err != 0) {
        const unsigned code = *h->h_f64_err;
        *h->h_f64_err = 0;
        return fail(RO_ERR_HIP, "the previous RO_PRECISION_F64 launch gave up waiting for another workgroup (code %u: 1 = row "
                                "map, 2 = ring slot still being read, 3 = row's first half not complete); its rows are incomplete", code);
    }
    if (!h->d_f64_ring) {
        int ring = RO_F64_RING_ROWS * 32768 / h->bins, wgs = RO_F64_WGS_PER_CU;
#ifdef RO_DIAG_KNOBS
        if (const char *e = getenv("RO_F64_RING_ROWS")) ring = atoi(e);
        if (const char *e = getenv("RO_F64_WGS")) wgs = atoi(e);
#endif
        h->f64_ring_rows = std::max(2, std::min(ring, ro::f64_fused_max_ring_rows()));
        h->f64_wgs_per_cu = std::max(1, std::min(wgs, 2));
        HIP_TRY(hipMalloc(&h->d_f64_ring, (size_t)8 * h->f64_ring_rows * h->bins * sizeof(double2)));
        HIP_TRY(hipMalloc(&h->d_f64_ctl, ro::f64_fused_ctl_bytes()));
        HIP_TRY(hipHostMalloc(reinterpret_cast<void **>(&h->h_f64_err), sizeof(unsigned), hipHostMallocDefault));
        *h->h_f64_err = 0;
    }
    ro::BigArgsD b{};
    b.iq = d_iq;
    b.window = h->d_window;
    b.tw = h->d_tw_f64;
    b.first_row = first_row;
    b.rows = rows;
    b.row_stride = row_stride;
    b.hop = h->hop;
    b.n = h->bins;
    b.gain = h->cfg.iq_gain;
    b.rows_out = d_rows;
    HIP_TRY(ro::launch_f64_fused(format, b, h->d_f64_ring, h->d_f64_ctl, h->f64_ring_rows, h->f64_wgs_per_cu, s));
    HIP_TRY(hipMemcpyAsync(h->h_f64_err, h->d_f64_ctl + 1, sizeof(unsigned), hipMemcpyDeviceToHost, s));
    return RO_OK;
}

// RO_PRECISION_F64: every size as radix-16 passes in double through HBM scratch, in chunks that fit it
int launch_transform_f64(ro_stft *h, const void *d_iq, int format, int64_t first_row, int64_t rows, float *d_rows,
                         int64_t row_stride, hipStream_t s)
{
    {
        bool fused = h->f64_one_launch;
#ifdef RO_DIAG_KNOBS
        if (const char *e = getenv("RO_F64_FUSED")) fused = atoi(e) != 0 && ro::f64_fused_supported(h->bins);
#endif
        if (fused) return launch_transform_f64_fused(h, d_iq, format, first_row, rows, d_rows, row_stride, s);
    }
    if (h->f64reg) {
        ro::F64RegArgs r{};
        r.iq = d_iq;
        r.window_k = h->d_f64r_window;
        r.tw0 = h->d_f64r_tw[0];
        r.tw1 = h->d_f64r_tw[1];
        r.tw2 = h->d_f64r_tw[2];
        r.tw3 = h->d_f64r_tw[3];
        r.rows_out = d_rows;
        r.first_row = first_row;
        r.rows = rows;
        r.row_stride = row_stride;
        r.hop = h->hop;
        r.gain = h->cfg.iq_gain;
        HIP_TRY(ro::launch_f64reg(h->bins, format, r, s));
        return RO_OK;
    }
    if (!h->d_scratch_d[0]) {
        int64_t mib = RO_F64_SCRATCH_MB;
#ifdef RO_DIAG
        if (const char *e = getenv("RO_F64_SCRATCH_MB")) mib = std::max(1, atoi(e));     // tools/r4/strict_sweep.sh
#endif
        h->scratch_rows_d = std::max<int64_t>(1, (mib << 20) / ((int64_t)h->bins * 16));
        for (int i = 0; i < 2; ++i)
            HIP_TRY(hipMalloc(&h->d_scratch_d[i], (size_t)h->scratch_rows_d * h->bins * sizeof(double2)));
    }
    int radix[8];
    const int passes = ro::f64_radices(h->bins, radix);
    if (passes < 2) return fail(RO_ERR_UNSUPPORTED, "no FP64 pass plan for bins = %d", h->bins);
    for (int64_t done = 0; done < rows; done += h->scratch_rows_d) {
        const int64_t n = std::min(h->scratch_rows_d, rows - done);
        ro::BigArgsD b{};
        b.iq = d_iq;
        b.window = h->d_window;
        b.tw = h->d_tw_f64;
        b.first_row = first_row + done;
        b.rows = n;
        b.row_stride = row_stride;
        b.hop = h->hop;
        b.n = h->bins;
        b.gain = h->cfg.iq_gain;
        // two passes per kernel where the pair fits its LDS tile (radix 16 followed by any radix, bins >= 4096): half
        // the trips through the scratch blocks
        int ns = 1, hop_idx = 0;
        b.rows_out = d_rows + done * row_stride;
        for (int p = 0; p < passes;) {
            const bool pair = h->bins >= 4096 && p + 1 < passes && radix[p] == 16;
            const int last_p = pair ? p + 1 : p;
            b.ns = ns;
            b.in = h->d_scratch_d[(hop_idx + 1) & 1];
            b.out = h->d_scratch_d[hop_idx & 1];
            if (pair) HIP_TRY(ro::launch_f64_pair(radix[p + 1], p == 0, last_p == passes - 1, format, b, s));
            else HIP_TRY(ro::launch_f64_pass(radix[p], p == 0, p == passes - 1, format, b, s));
            for (int q = p; q <= last_p; ++q) ns *= radix[q];
            p = last_p + 1;
            ++hop_idx;
        }
    }
    return RO_OK;
}

int launch_transform(ro_stft *h, const void *d_iq, int format, int64_t first_row, int64_t rows, float *d_rows,
                     int64_t row_stride, hipStream_t s, float *d_tile, ro_scan_record_t *d_records, float *d_ln)
{
    if (h->f64) return launch_transform_f64(h, d_iq, format, first_row, rows, d_rows, row_stride, s);
    if (h->czt) return launch_transform_czt(h, d_iq, format, first_row, rows, d_rows, row_stride, s);
    if (!h->big) {
        if (d_tile && d_ln && ro::stft_fuses_scan(h->bins)) {
            int rc = ensure_ln_part(h, rows);
            if (rc != RO_OK) return rc;
        }
        ro::StftArgs a = make_stft_args(h, d_iq, first_row, rows, d_rows, row_stride, d_tile, d_records, d_ln);
        HIP_TRY(ro::launch_stft(h->bins, format, a, s));
        return RO_OK;
    }
    if (h->four) {
        {
            int64_t mib = RO_FOUR_SCRATCH_MB;
#ifdef RO_DIAG_KNOBS
            if (const char *e = getenv("RO_FOUR_SCRATCH_MB")) mib = std::max<int64_t>(4, atoll(e));
#endif
            const int64_t limit = std::max<int64_t>(1, (mib << 20) / ((int64_t)h->bins * 8));
            const int64_t want = std::min(limit, rows);
            if (want > h->four_rows) {
                // (launches on one handle are ordered on the caller's stream; the old block may still be in use there)
                if (h->d_four_z) {
                    HIP_TRY(hipStreamSynchronize(s));
                    HIP_TRY(hipFree(h->d_four_z));
                    h->d_four_z = nullptr;
                    h->four_rows = 0;
                }
                HIP_TRY(hipMalloc(&h->d_four_z, (size_t)want * h->bins * 2 * sizeof(float)));
                h->four_rows = want;
            }
        }
        for (int64_t done = 0; done < rows; done += h->four_rows) {
            ro::FourArgs f{};
            f.iq = d_iq;
            f.first_row = first_row + done;
            f.rows = std::min(h->four_rows, rows - done);
            f.hop = h->hop;
            f.gain = (float)h->cfg.iq_gain;
            f.n1 = h->bins / 1024;
            f.window_a = h->d_four_window;
            f.tw_a = h->d_four_tw_a;
            f.tw_b = h->d_four_tw_b;
            f.tw_r = h->d_four_tw_r;
            f.z = h->d_four_z;
            f.rows_out = d_rows + done * row_stride;
            f.row_stride = row_stride;
            f.spare_cus = h->cfg.spare_cus_per_xcd;
            HIP_TRY(ro::launch_fourstep(format, f, s));
        }
        return RO_OK;
    }
    if (h->dec > 1 && h->dif) {
        // one kernel, no scratch: kernel row srow * dec + q makes the bins q + dec k' of stream row srow
        int log2 = 0;
        while ((1 << log2) < h->dec) ++log2;
        ro::StftArgs a = make_stft_args(h, d_iq, first_row, rows * h->dec, d_rows, row_stride);
        a.window = h->d_window;
        a.window_k = h->d_window_dif;
        a.dec = h->dec;
        a.dec_log2 = log2;
        a.dif_tw = h->d_dif_tw;
        a.dif_shift = h->d_dif_shift;
        a.big_form = 1;
        HIP_TRY(ro::launch_stft(h->sub_bins, format, a, s));
        return RO_OK;
    }
    return fail(RO_ERR_STATE, "internal: no transform plan for bins = %d", h->bins);
}

Batch *acquire_batch(ro_stft *h)
{
    if (!h->batch_pool.empty()) {
        Batch *b = h->batch_pool.back();
        h->batch_pool.pop_back();
        return b;
    }
    Batch *b = new (std::nothrow) Batch();
    if (!b) return nullptr;
    b->capacity_rows = h->batch_rows;
    if ((!h->sink &&
         hipHostMalloc(reinterpret_cast<void **>(&b->data), (size_t)b->capacity_rows * h->out_cols * sizeof(float),
                       hipHostMallocDefault) != hipSuccess) ||
        hipHostMalloc(reinterpret_cast<void **>(&b->records), (size_t)b->capacity_rows * sizeof(ro_scan_record_t),
                      hipHostMallocDefault) != hipSuccess ||
        (h->cfg.tile_ln &&
         (hipHostMalloc(reinterpret_cast<void **>(&b->ln), (size_t)b->capacity_rows * h->out_cols * sizeof(float),
                        hipHostMallocDefault) != hipSuccess ||
          hipHostMalloc(reinterpret_cast<void **>(&b->minmax), (size_t)b->capacity_rows * 2 * sizeof(float),
                        hipHostMallocDefault) != hipSuccess)) ||
        hipEventCreateWithFlags(&b->done, hipEventDisableTiming) != hipSuccess ||
        hipEventCreate(&b->k0) != hipSuccess || hipEventCreate(&b->k1) != hipSuccess) {
        if (b->data) (void)hipHostFree(b->data);
        if (b->ln) (void)hipHostFree(b->ln);
        if (b->minmax) (void)hipHostFree(b->minmax);
        if (b->records) (void)hipHostFree(b->records);
        if (b->done) (void)hipEventDestroy(b->done);
        if (b->k0) (void)hipEventDestroy(b->k0);
        if (b->k1) (void)hipEventDestroy(b->k1);
        delete b;
        return nullptr;
    }
    return b;
}

void release_batch(ro_stft *h, Batch *b)
{
    b->consumed = 0;
    b->rows = 0;
    b->pending = false;
    h->batch_pool.push_back(b);
}

void destroy_batch(Batch *b)
{
    if (b->data) (void)hipHostFree(b->data);
    if (b->ln) (void)hipHostFree(b->ln);
    if (b->minmax) (void)hipHostFree(b->minmax);
    if (b->records) (void)hipHostFree(b->records);
    if (b->done) (void)hipEventDestroy(b->done);
    if (b->k0) (void)hipEventDestroy(b->k0);
    if (b->k1) (void)hipEventDestroy(b->k1);
    delete b;
}

void free_stream_slots(ro_stft *h)
{
    for (auto &sl : h->slot) {
        if (sl.d_iq) (void)hipFree(sl.d_iq);
        if (sl.d_rows) (void)hipFree(sl.d_rows);
        if (sl.d_tile) (void)hipFree(sl.d_tile);
        if (sl.d_ln) (void)hipFree(sl.d_ln);
        if (sl.d_minmax) (void)hipFree(sl.d_minmax);
        if (sl.d_records) (void)hipFree(sl.d_records);
        if (sl.h_in) (void)hipHostFree(sl.h_in);
        if (sl.uploaded) (void)hipEventDestroy(sl.uploaded);
        if (sl.computed) (void)hipEventDestroy(sl.computed);
        if (sl.drained) (void)hipEventDestroy(sl.drained);
        if (sl.gexec) (void)hipGraphExecDestroy(sl.gexec);
        if (sl.gstream) (void)hipStreamDestroy(sl.gstream);
        sl = ro_stft::Slot();
    }
    if (h->s_in) (void)hipStreamDestroy(h->s_in);
    if (h->s_out) (void)hipStreamDestroy(h->s_out);
    h->s_in = h->s_out = nullptr;
    h->slots_ready = false;
}

// streaming buffers, all or nothing: a failure half way frees what was allocated, and the next push tries again
int ensure_stream_slots(ro_stft *h)
{
    if (h->slots_ready) return RO_OK;
    HIP_TRY(hipSetDevice(h->device));
    const size_t in_samples = (size_t)(h->batch_rows - 1) * h->hop + h->bins;
    hipError_t e = hipSuccess;
    auto ok = [&](hipError_t r) { if (e == hipSuccess) e = r; return e == hipSuccess; };
    // Three streams chained by events: the upload of batch n + 1 overlaps the kernels of batch n and the download of
    // batch n - 1.  (Round 5 measured everything in order on ONE stream for latency-bound batches -- eight runtime calls
    // fewer per batch: push 3.9 -> 2.7 us per call, and the same 6.7e4 rows/s, because a batch then occupies the stream
    // for its whole upload -> kernel -> download chain, ~90 us; with three streams a second batch in flight overlaps it.)
    ok(hipStreamCreateWithFlags(&h->s_in, hipStreamNonBlocking)) && ok(hipStreamCreateWithFlags(&h->s_out, hipStreamNonBlocking));
    for (auto &sl : h->slot) {
        ok(hipMalloc(&sl.d_iq, in_samples * 2 * sizeof(float))) &&
            ok(hipMalloc(&sl.d_rows, (size_t)h->batch_rows * h->bins * sizeof(float))) &&
            ok(hipMalloc(&sl.d_records, (size_t)h->batch_rows * sizeof(ro_scan_record_t))) &&
            ok(hipHostMalloc(&sl.h_in, in_samples * 2 * sizeof(float), hipHostMallocDefault)) &&
            ok(hipEventCreateWithFlags(&sl.uploaded, hipEventDisableTiming)) &&
            ok(hipEventCreateWithFlags(&sl.computed, hipEventDisableTiming)) &&
            ok(hipEventCreateWithFlags(&sl.drained, hipEventDisableTiming));
        if (h->cfg.tile_cols > 0)
            ok(hipMalloc(&sl.d_tile, (size_t)h->batch_rows * h->cfg.tile_cols * sizeof(float)));
        if (h->cfg.tile_ln)
            ok(hipMalloc(&sl.d_ln, (size_t)h->batch_rows * h->cfg.tile_cols * sizeof(float))) &&
                ok(hipMalloc(&sl.d_minmax, (size_t)h->batch_rows * 2 * sizeof(float)));
    }
    if (e != hipSuccess) {
        free_stream_slots(h);
        return fail(RO_ERR_HIP, "allocating the streaming buffers failed: %s", hipGetErrorString(e));
    }
    h->slots_ready = true;
    return RO_OK;
}

size_t stage_sample_bytes(const ro_stft *h) { return h->stage_fmt == RO_IQ_I16 ? 4 : 8; }

// The one place the host waits for the GPU on the streaming path: a batch's download has finished.  Its kernel time
// (GPU events around its kernels) goes into the counters of ro_stft_timing / ro_stft_stats the first time round.
int await_batch(ro_stft *h, Batch *b)
{
    if (!b->pending) return RO_OK;
    HIP_TRY(hipEventSynchronize(b->done));
    b->pending = false;
    float ms = 0.f;
    h->timing.batches += 1;
    h->timing.batch_rows += b->rows;
    if (b->timed && hipEventElapsedTime(&ms, b->k0, b->k1) == hipSuccess) {
        h->stat_kernel_ms += ms;
        h->timed_batches += 1;
        h->timed_rows += b->rows;
        h->batch_ms_sum += ms;
        h->last_batch_ms = ms;
        h->timing.batch_gpu_ms_max = std::max(h->timing.batch_gpu_ms_max, (double)ms);
    } else if (!b->timed) {
        h->stat_kernel_ms += h->last_batch_ms;      // (an untimed graphed batch: the same graph as the last timed one)
    }
    return RO_OK;
}

// run one batch of the streaming path: rows [rows_emitted, +rows) from the staged samples.  Upload, kernels and
// download are queued on three streams chained by events and the call returns; the host only waits when it is about
// to overwrite a pinned staging buffer whose upload has not finished.
int run_stream_batch(ro_stft *h, int64_t rows)
{
    if (rows <= 0) return RO_OK;
    HIP_TRY(hipSetDevice(h->device));
    const size_t sb = stage_sample_bytes(h);
    const int64_t need = (rows - 1) * (int64_t)h->hop + h->bins;       // samples
    if ((int64_t)h->staged_have < need) return fail(RO_ERR_STATE, "internal: %lld samples staged, %lld needed",
                                                    (long long)h->staged_have, (long long)need);
    // rows that have not been fetched sit in the sink's slots: a batch that would lap them is not launched.  ro_stft_push
    // never gets here in that state (it refuses such a call whole, before staging); ro_stft_flush does, and leaves the
    // staged samples where they are, so a flush repeated after a fetch loses nothing
    if (h->sink && h->rows_ready + rows > h->sink_cap)
        return fail(RO_ERR_STATE, "row sink full: %lld rows wait to be fetched in a ring of %lld slots", (long long)h->rows_ready,
                    (long long)h->sink_cap);
    ro_stft::Slot &sl = h->slot[h->batch_seq % RO_STREAM_SLOTS];                      // (its samples are already in sl.h_in)
    // Back-pressure: one large ro_stft_push must not queue a pinned batch per launch without bound (64 MiB each with
    // full rows).  Batches older than the newest MAX_IN_FLIGHT are waited for here -- they stay in `ready` for the
    // next fetch, their buffers are simply known to be complete.
    constexpr size_t MAX_IN_FLIGHT = 4;
    if (h->ready.size() >= MAX_IN_FLIGHT) {
        const int wrc = await_batch(h, h->ready[h->ready.size() - MAX_IN_FLIGHT]);
        if (wrc != RO_OK) return wrc;
    }
    Batch *b = acquire_batch(h);
    if (!b) return fail(RO_ERR_NOMEM, "out of pinned host memory for a row batch");
    int rc = RO_OK;
    auto step = [&](hipError_t e, const char *what) {
        if (rc == RO_OK && e != hipSuccess) rc = fail(RO_ERR_HIP, "%s failed: %s", what, hipGetErrorString(e));
        return rc == RO_OK;
    };
    // ---- A latency-bound batch (a second of rows: well under a MiB) is all launch overhead -- fourteen runtime calls on
    // three streams for 23 us of GPU work.  With a row sink, a FULL batch of a slot runs as ONE graph (upload and every
    // kernel of the size: captured once from the calls below) on the slot's own stream, followed by the downloads into
    // the sink's slots: seven calls.  The slots' streams overlap a batch with the one or two before it; a slot's
    // own batches are ordered by its stream.  Partial batches (a flush) and the first batch of a slot take the plain path.
    const bool small = (size_t)h->batch_rows * h->out_cols * sizeof(float) <= ((size_t)4 << 20);
    // (float32 power-of-two handles only: the FP64 and chirp-z paths keep per-launch host state -- a give-up word, an inner
    // handle -- that a replayed graph would not see)
    bool graphed = RO_STREAM_GRAPH && small && h->sink && !h->cfg.tile_ln && !h->f64 && !h->czt && rows == h->batch_rows && sl.uses > 0;
    ro_scan_record_t *g_recs = h->cfg.enable_scan ? sl.d_records : nullptr;
    if (graphed && (!sl.gexec || sl.graph_fmt != h->stage_fmt)) {
        if (sl.gexec) { (void)hipGraphExecDestroy(sl.gexec); sl.gexec = nullptr; }
        if (!sl.gstream) step(hipStreamCreateWithFlags(&sl.gstream, hipStreamNonBlocking), "hipStreamCreateWithFlags");
        hipGraph_t g = nullptr;
        if (rc == RO_OK && step(hipStreamBeginCapture(sl.gstream, hipStreamCaptureModeThreadLocal), "hipStreamBeginCapture")) {
            step(hipMemcpyAsync(sl.d_iq, sl.h_in, (size_t)need * sb, hipMemcpyHostToDevice, sl.gstream), "upload");
            if (rc == RO_OK) rc = launch_transform(h, sl.d_iq, h->stage_fmt, 0, rows, sl.d_rows, h->bins, sl.gstream, sl.d_tile, g_recs, sl.d_ln);
            if (rc == RO_OK) rc = launch_tile_and_scan(h, sl.d_rows, h->bins, rows, sl.d_tile, g_recs, sl.gstream, sl.d_ln, sl.d_minmax);
            const hipError_t ce = hipStreamEndCapture(sl.gstream, &g);          // (always: leaves capture mode)
            if (rc == RO_OK) step(ce, "hipStreamEndCapture");
            if (rc == RO_OK) step(hipGraphInstantiate(&sl.gexec, g, nullptr, nullptr, 0), "hipGraphInstantiate");
            if (g) (void)hipGraphDestroy(g);
        }
        if (rc != RO_OK) {                        // no graph on this runtime: the plain path from now on, not an error
            (void)hipGetLastError();
            sl.gexec = nullptr;
            h->graph_refused = true;
            rc = RO_OK;
        }
        sl.graph_fmt = h->stage_fmt;
    }
    graphed = graphed && sl.gexec && !h->graph_refused;
    if (graphed != sl.on_gstream && sl.uses > 0) {
        // the slot changes streams: what its last batch queued has to be over (a handful of times per stream: first graphed
        // batch, a flush)
        if (sl.on_gstream) step(hipStreamSynchronize(sl.gstream), "hipStreamSynchronize");
        else step(hipStreamSynchronize(h->s_out), "hipStreamSynchronize");
    }
    sl.on_gstream = graphed;
    sl.uses += 1;
    if (graphed) {
        hipStream_t gs = sl.gstream;
        // (`uploaded` is recorded by a stream call BEHIND the graph, not by a node inside it: the host waits on it before it
        // stages into this slot's pinned buffer again, and an event that only a queued graph will record still reads as
        // its previous, completed record -- the host then overwrote samples the upload had not read yet: found by the
        // seeded soak of tests/test_gpu_streaming.py)
        // A runtime call costs the host 1 - 4 us here (tools/r5/event_cost.hip: an event record 2.4, the graph 10, a 2-D copy
        // 4) and a batch of a second of rows is 40 us of host time in all, so the timing events around the kernels go round
        // one batch in RO_GRAPH_TIME_EVERY only (the graph is the same every time; ro_stft_timing averages over the timed
        // ones): +8 % rows/s at the Backend's default batch, settings alternated inside one process
        // (profiles/r05_host_calls_ab.txt).  The same A/B says the `uploaded` event has to stay: with the batch's own `done`
        // event as the host's "staging buffer is free again" the host waits for a whole batch two launches back instead of
        // its upload, and the rate halves.
        const int every = h->diag_time_every > 0 ? h->diag_time_every : RO_GRAPH_TIME_EVERY;
        b->timed = h->graph_batches++ % every == 0;
        if (b->timed) step(hipEventRecord(b->k0, gs), "hipEventRecord");
        if (h->diag_direct) {                       // the graph's calls made one by one on the slot's stream
            step(hipMemcpyAsync(sl.d_iq, sl.h_in, (size_t)need * sb, hipMemcpyHostToDevice, gs), "upload");
            if (rc == RO_OK) rc = launch_transform(h, sl.d_iq, h->stage_fmt, 0, rows, sl.d_rows, h->bins, gs, sl.d_tile, g_recs, sl.d_ln);
            if (rc == RO_OK) rc = launch_tile_and_scan(h, sl.d_rows, h->bins, rows, sl.d_tile, g_recs, gs, sl.d_ln, sl.d_minmax);
        } else {
            step(hipGraphLaunch(sl.gexec, gs), "hipGraphLaunch");
        }
        if (b->timed) step(hipEventRecord(b->k1, gs), "hipEventRecord");
        if (!h->diag_done_only) {
            step(hipEventRecord(sl.uploaded, gs), "hipEventRecord");
            sl.staging_free = sl.uploaded;
        } else {
            sl.staging_free = b->done;
        }
        if (rc == RO_OK) {
            const float *src = h->cfg.tile_cols > 0 ? sl.d_tile : sl.d_rows;
            const size_t w = (size_t)h->out_cols * sizeof(float);
            const int64_t s0 = (h->sink_first + h->rows_emitted) % h->sink_cap;
            const int64_t n0 = std::min<int64_t>(rows, h->sink_cap - s0);
            step(hipMemcpy2DAsync(h->sink + s0 * h->sink_stride, (size_t)h->sink_stride * sizeof(float), src, w, w, (size_t)n0,
                                  hipMemcpyDeviceToHost, gs), "download");
            if (n0 < rows)
                step(hipMemcpy2DAsync(h->sink, (size_t)h->sink_stride * sizeof(float), src + (size_t)n0 * h->out_cols, w, w,
                                      (size_t)(rows - n0), hipMemcpyDeviceToHost, gs), "download");
            if (h->cfg.enable_scan)
                step(hipMemcpyAsync(b->records, sl.d_records, (size_t)rows * sizeof(ro_scan_record_t), hipMemcpyDeviceToHost, gs),
                     "download");
        }
        step(hipEventRecord(b->done, gs), "hipEventRecord");
    } else {
    // upload (s_in): after the kernels that last read this slot's d_iq
    step(hipStreamWaitEvent(h->s_in, sl.computed, 0), "hipStreamWaitEvent") &&
        step(hipMemcpyAsync(sl.d_iq, sl.h_in, (size_t)need * sb, hipMemcpyHostToDevice, h->s_in), "upload") &&
        step(hipEventRecord(sl.uploaded, h->s_in), "hipEventRecord");
    sl.staging_free = sl.uploaded;
    b->timed = true;
    // kernels (stream): after the upload, and after the download that last read this slot's outputs
    step(hipStreamWaitEvent(h->stream, sl.uploaded, 0), "hipStreamWaitEvent") &&
        step(hipStreamWaitEvent(h->stream, sl.drained, 0), "hipStreamWaitEvent") &&
        step(hipEventRecord(b->k0, h->stream), "hipEventRecord");
    if (rc == RO_OK) {
        ro_scan_record_t *recs = h->cfg.enable_scan ? sl.d_records : nullptr;
        rc = launch_transform(h, sl.d_iq, h->stage_fmt, 0, rows, sl.d_rows, h->bins, h->stream, sl.d_tile, recs, sl.d_ln);
        if (rc == RO_OK)
            rc = launch_tile_and_scan(h, sl.d_rows, h->bins, rows, sl.d_tile, recs, h->stream, sl.d_ln, sl.d_minmax);
    }
    step(hipEventRecord(b->k1, h->stream), "hipEventRecord") && step(hipEventRecord(sl.computed, h->stream), "hipEventRecord");
    // download (s_out): only the columns somebody asked for travel -- the tile when one is configured
    step(hipStreamWaitEvent(h->s_out, sl.computed, 0), "hipStreamWaitEvent");
    if (rc == RO_OK) {
        const float *src = h->cfg.tile_cols > 0 ? sl.d_tile : sl.d_rows;
        if (h->sink) {
            // straight into the caller's ring: row r of the stream at slot (sink_first + r) mod sink_cap, in at most
            // two runs of consecutive slots
            const size_t w = (size_t)h->out_cols * sizeof(float);
            const int64_t s0 = (h->sink_first + h->rows_emitted) % h->sink_cap;
            const int64_t n0 = std::min<int64_t>(rows, h->sink_cap - s0);
            step(hipMemcpy2DAsync(h->sink + s0 * h->sink_stride, (size_t)h->sink_stride * sizeof(float), src, w, w, (size_t)n0,
                                  hipMemcpyDeviceToHost, h->s_out), "download");
            if (n0 < rows)
                step(hipMemcpy2DAsync(h->sink, (size_t)h->sink_stride * sizeof(float), src + (size_t)n0 * h->out_cols, w, w,
                                      (size_t)(rows - n0), hipMemcpyDeviceToHost, h->s_out), "download");
        } else {
            step(hipMemcpyAsync(b->data, src, (size_t)rows * h->out_cols * sizeof(float), hipMemcpyDeviceToHost, h->s_out),
                 "download");
        }
        if (h->cfg.enable_scan)
            step(hipMemcpyAsync(b->records, sl.d_records, (size_t)rows * sizeof(ro_scan_record_t), hipMemcpyDeviceToHost,
                                h->s_out), "download");
        if (h->cfg.tile_ln) {
            step(hipMemcpyAsync(b->ln, sl.d_ln, (size_t)rows * h->out_cols * sizeof(float), hipMemcpyDeviceToHost, h->s_out),
                 "download");
            step(hipMemcpyAsync(b->minmax, sl.d_minmax, (size_t)rows * 2 * sizeof(float), hipMemcpyDeviceToHost, h->s_out),
                 "download");
        }
    }
    step(hipEventRecord(sl.drained, h->s_out), "hipEventRecord") && step(hipEventRecord(b->done, h->s_out), "hipEventRecord");
    }
    if (rc != RO_OK) {
        // nothing of this batch is handed out; whatever was queued is allowed to finish before the buffers are reused
        (void)hipStreamSynchronize(h->s_in);
        (void)hipStreamSynchronize(h->stream);
        (void)hipStreamSynchronize(h->s_out);
        if (sl.gstream) (void)hipStreamSynchronize(sl.gstream);
        release_batch(h, b);
        return rc;
    }
    b->first_row = h->rows_emitted;
    b->rows = rows;
    b->pending = true;
    h->batch_seq += 1;
    h->stat_launches += 1;
    h->stat_rows += rows;

    // the samples no later row needs are spent: the next row starts rows*hop further on.  What is left -- the overlap
    // and anything behind the batch's last row -- moves to the front of the other slot's staging buffer, whose own
    // upload (the batch before this one) has to be over first; the upload just queued only READS this slot.
    const int64_t consumed = rows * (int64_t)h->hop;
    h->stream_sample0 += consumed;
    h->rows_emitted += rows;
    h->rows_ready += rows;
    h->ready.push_back(b);
    ro_stft::Slot &nx = h->slot[h->batch_seq % RO_STREAM_SLOTS];                      // (batch_seq has moved on)
    const hipError_t we = hipEventSynchronize(nx.staging_free ? nx.staging_free : nx.uploaded);
    const size_t left = h->staged_have - (size_t)consumed;
    std::memcpy(nx.h_in, static_cast<const char *>(sl.h_in) + (size_t)consumed * sb, left * sb);
    h->staged_have = left;
    if (we != hipSuccess) return fail(RO_ERR_HIP, "hipEventSynchronize failed: %s", hipGetErrorString(we));
    return RO_OK;
}

int64_t staged_complete_rows(const ro_stft *h)
{
    const int64_t have = (int64_t)h->staged_have;
    if (have < h->bins) return 0;
    return (have - h->bins) / h->hop + 1;
}

double now_ms()
{
    timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return ts.tv_sec * 1e3 + ts.tv_nsec * 1e-6;
}

}  // namespace

// ---------------------------------------------------------------------------
// library
// ---------------------------------------------------------------------------
extern "C" int ro_abi_version(void) { return RO_ABI_VERSION; }
extern "C" const char *ro_last_error(void) { return g_error.c_str(); }

extern "C" int ro_device_count(void)
{
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) return fail(RO_ERR_HIP, "hipGetDeviceCount: %s", hipGetErrorString(e));
    return n;
}

// ---------------------------------------------------------------------------
// host helpers (FFTBackend's scalar arithmetic; float/double mixing as in the reference)
// ---------------------------------------------------------------------------
extern "C" int ro_clamp_overlap(int bins, int overlap)
{
    if (overlap < 0) return 0;                       // src/FFTBackend.cpp:108
    if (overlap >= bins) return bins - 1;            // :109
    return overlap;
}

extern "C" float ro_fft_sample_rate(int sample_rate, int bins, int overlap)
{
    return (float)sample_rate / (float)(bins - ro_clamp_overlap(bins, overlap));   // :150-151
}

extern "C" int ro_frequency_to_bin(int bins, int sample_rate, float frequency)
{
    // src/FFTBackend.h:169-175: float quotient, double sum and product, truncation, clamp
    const float sr = (float)sample_rate, n = (float)bins;
    const int bin = (int)((double)n * ((double)(frequency / sr) + 0.5));
    if (bin < 0) return 0;
    if (bin >= bins) return bins - 1;
    return bin;
}

extern "C" float ro_bin_to_frequency(int bins, int sample_rate, int bin)
{
    // src/FFTBackend.h:141-145: float quotient, the rest in double, narrowed on return
    const float b = (float)bin, sr = (float)sample_rate, n = (float)bins;
    return (float)((double)sr * (-0.5 + (double)(b / n)));
}

extern "C" int ro_time_to_fft_samples(double seconds, float fft_sample_rate)
{
    return (int)(seconds * (double)fft_sample_rate);             // src/FFTBackend.h:197-200
}

extern "C" int64_t ro_row_count(int64_t samples, int bins, int overlap)
{
    const int64_t hop = bins - ro_clamp_overlap(bins, overlap);
    if (samples < bins) return 0;
    return (samples - bins) / hop + 1;
}

extern "C" int ro_window_table(int kind, int bins, float *out)
{
    if (!out || bins < 2) return fail(RO_ERR_INVALID, "ro_window_table: bad arguments");
    if (kind != RO_WINDOW_NUTTALL && kind != RO_WINDOW_HANN)
        return fail(RO_ERR_INVALID, "ro_window_table: kind %d has no formula", kind);
    build_window(kind, bins, out);
    return RO_OK;
}

// ---------------------------------------------------------------------------
// time-chunk sharding (host arithmetic; the Python side, timeshard.py, calls these)
// ---------------------------------------------------------------------------
extern "C" int ro_shard_rows(int64_t total_rows, int world, int rank, int64_t *first_row, int64_t *rows)
{
    if (total_rows < 0 || world < 1 || rank < 0 || rank >= world || !first_row || !rows)
        return fail(RO_ERR_INVALID, "ro_shard_rows: bad arguments");
    // 128-bit products: rank * total_rows overflows int64 only for absurd sizes, but costs nothing to rule out
    const int64_t lo = (int64_t)(((__int128)rank * total_rows) / world);
    const int64_t hi = (int64_t)(((__int128)(rank + 1) * total_rows) / world);
    *first_row = lo;
    *rows = hi - lo;
    return RO_OK;
}

extern "C" int ro_shard_samples(int64_t first_row, int64_t rows, int bins, int overlap, int64_t *first_sample,
                                int64_t *samples)
{
    if (first_row < 0 || rows < 0 || bins < 2 || !first_sample || !samples)
        return fail(RO_ERR_INVALID, "ro_shard_samples: bad arguments");
    const int64_t hop = bins - ro_clamp_overlap(bins, overlap);
    *first_sample = first_row * hop;
    *samples = rows > 0 ? (rows - 1) * hop + bins : 0;
    return RO_OK;
}

extern "C" int64_t ro_shard_max_rows(int64_t total_rows, int world)
{
    if (total_rows < 0 || world < 1) return fail(RO_ERR_INVALID, "ro_shard_max_rows: bad arguments");
    return (total_rows + world - 1) / world;       // sizes differ by at most one: the largest is the ceiling
}

extern "C" int ro_stitch_rows(const void *gathered, int64_t total_rows, int world, size_t row_bytes, void *out)
{
    if (total_rows < 0 || world < 1 || (total_rows > 0 && (!gathered || !out)))
        return fail(RO_ERR_INVALID, "ro_stitch_rows: bad arguments");
    const int64_t block = ro_shard_max_rows(total_rows, world);
    const char *src = static_cast<const char *>(gathered);
    char *dst = static_cast<char *>(out);
    for (int g = 0; g < world; ++g) {
        int64_t first = 0, rows = 0;
        ro_shard_rows(total_rows, world, g, &first, &rows);
        std::memcpy(dst + (size_t)first * row_bytes, src + (size_t)g * (size_t)block * row_bytes,
                    (size_t)rows * row_bytes);
    }
    return RO_OK;
}

// RCCL, resolved at run time so that the library has no link-time dependency on it: one dlopen / dlsym per process,
// under std::call_once (several host threads may drive their own handles and communicators)
namespace {
struct Rccl {
    int (*all_gather)(const void *, void *, size_t, int, void *, hipStream_t) = nullptr;
    int (*group_start)(void) = nullptr;
    int (*group_end)(void) = nullptr;
    int (*send)(const void *, size_t, int, int, void *, hipStream_t) = nullptr;
    int (*recv)(void *, size_t, int, int, void *, hipStream_t) = nullptr;
};
const Rccl &rccl_api()
{
    static Rccl api;
    static std::once_flag once;
    std::call_once(once, [] {
        void *lib = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
        if (!lib) lib = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
        if (!lib) lib = dlopen("/opt/rocm/lib/librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
        if (!lib) return;
        api.all_gather = reinterpret_cast<decltype(api.all_gather)>(dlsym(lib, "ncclAllGather"));
        api.group_start = reinterpret_cast<decltype(api.group_start)>(dlsym(lib, "ncclGroupStart"));
        api.group_end = reinterpret_cast<decltype(api.group_end)>(dlsym(lib, "ncclGroupEnd"));
        api.send = reinterpret_cast<decltype(api.send)>(dlsym(lib, "ncclSend"));
        api.recv = reinterpret_cast<decltype(api.recv)>(dlsym(lib, "ncclRecv"));
    });
    return api;
}
}  // namespace

// the all-gather itself
extern "C" int ro_allgather_rows(void *nccl_comm, const void *d_local, int64_t local_rows, int64_t total_rows, int world,
                                 int rank, size_t row_bytes, void *d_staging, void *d_gathered, void *stream)
{
    if (!nccl_comm || world < 1 || rank < 0 || rank >= world || total_rows < 0 || row_bytes == 0 || !d_staging ||
        !d_gathered || (local_rows > 0 && !d_local))
        return fail(RO_ERR_INVALID, "ro_allgather_rows: bad arguments");
    int64_t first = 0, mine = 0;
    ro_shard_rows(total_rows, world, rank, &first, &mine);
    if (local_rows != mine)
        return fail(RO_ERR_INVALID, "ro_allgather_rows: rank %d of %d owns %lld of %lld rows, not %lld", rank, world,
                    (long long)mine, (long long)total_rows, (long long)local_rows);
    const Rccl &rccl = rccl_api();
    if (!rccl.all_gather) return fail(RO_ERR_UNSUPPORTED, "librccl (ncclAllGather) not found on this host");
    hipStream_t s = (hipStream_t)stream;
    const int64_t block = ro_shard_max_rows(total_rows, world);
    if (block == 0) return RO_OK;
    const size_t used = (size_t)local_rows * row_bytes, whole = (size_t)block * row_bytes;
    if (used) HIP_TRY(hipMemcpyAsync(d_staging, d_local, used, hipMemcpyDeviceToDevice, s));
    if (whole > used) HIP_TRY(hipMemsetAsync(static_cast<char *>(d_staging) + used, 0, whole - used, s));
    const int rc = rccl.all_gather(d_staging, d_gathered, whole, /*ncclInt8*/ 0, nccl_comm, s);
    if (rc != 0) return fail(RO_ERR_HIP, "ncclAllGather failed with code %d", rc);
    return RO_OK;
}

// gather to ONE rank, rows landing where they belong: ncclSend / ncclRecv in a group, no padding, no stitch
extern "C" int ro_gather_rows(void *nccl_comm, const void *d_local, int64_t local_rows, int64_t total_rows, int world,
                              int rank, int root, size_t row_bytes, void *d_out, void *stream)
{
    if (!nccl_comm || world < 1 || rank < 0 || rank >= world || root < 0 || root >= world || total_rows < 0 ||
        row_bytes == 0 || (local_rows > 0 && !d_local) || (rank == root && total_rows > 0 && !d_out))
        return fail(RO_ERR_INVALID, "ro_gather_rows: bad arguments");
    int64_t first = 0, mine = 0;
    ro_shard_rows(total_rows, world, rank, &first, &mine);
    if (local_rows != mine)
        return fail(RO_ERR_INVALID, "ro_gather_rows: rank %d of %d owns %lld of %lld rows, not %lld", rank, world,
                    (long long)mine, (long long)total_rows, (long long)local_rows);
    const Rccl &rccl = rccl_api();
    if (!rccl.group_start || !rccl.group_end || !rccl.send || !rccl.recv)
        return fail(RO_ERR_UNSUPPORTED, "librccl (ncclSend / ncclRecv) not found on this host");
    const auto group_start = rccl.group_start, group_end = rccl.group_end;
    const auto send = rccl.send;
    const auto recv = rccl.recv;
    hipStream_t s = (hipStream_t)stream;
    if (rank == root && mine > 0)           // the root's own rows: a copy
        HIP_TRY(hipMemcpyAsync(static_cast<char *>(d_out) + (size_t)first * row_bytes, d_local, (size_t)mine * row_bytes,
                               hipMemcpyDeviceToDevice, s));
    int rc = group_start();
    if (rc == 0 && rank != root && mine > 0) rc = send(d_local, (size_t)mine * row_bytes, /*ncclInt8*/ 0, root, nccl_comm, s);
    if (rank == root)
        for (int g = 0; g < world && rc == 0; ++g) {
            int64_t f = 0, n = 0;
            ro_shard_rows(total_rows, world, g, &f, &n);
            if (g != root && n > 0)
                rc = recv(static_cast<char *>(d_out) + (size_t)f * row_bytes, (size_t)n * row_bytes, 0, g, nccl_comm, s);
        }
    const int rc_end = group_end();
    if (rc != 0 || rc_end != 0) return fail(RO_ERR_HIP, "ncclSend / ncclRecv failed with code %d", rc ? rc : rc_end);
    return RO_OK;
}

// the all-gather as a DIRECT exchange: inside one group every rank sends its block to each peer and receives each peer's
// block at its stitched place -- world - 1 point-to-point transfers per rank over world - 1 different xGMI links, no
// ring through one link, no padding, no stitch (what ro_gather_rows does for one root, for all)
extern "C" int ro_allgather_rows_direct(void *nccl_comm, const void *d_local, int64_t local_rows, int64_t total_rows, int world,
                                        int rank, size_t row_bytes, void *d_out, void *stream)
{
    if (!nccl_comm || world < 1 || rank < 0 || rank >= world || total_rows < 0 || row_bytes == 0 ||
        (local_rows > 0 && !d_local) || (total_rows > 0 && !d_out))
        return fail(RO_ERR_INVALID, "ro_allgather_rows_direct: bad arguments");
    int64_t first = 0, mine = 0;
    ro_shard_rows(total_rows, world, rank, &first, &mine);
    if (local_rows != mine)
        return fail(RO_ERR_INVALID, "ro_allgather_rows_direct: rank %d of %d owns %lld of %lld rows, not %lld", rank, world,
                    (long long)mine, (long long)total_rows, (long long)local_rows);
    const Rccl &rccl = rccl_api();
    if (!rccl.group_start || !rccl.group_end || !rccl.send || !rccl.recv)
        return fail(RO_ERR_UNSUPPORTED, "librccl (ncclSend / ncclRecv) not found on this host");
    hipStream_t s = (hipStream_t)stream;
    char *out = static_cast<char *>(d_out);
    if (mine > 0)                            // this rank's own rows: a copy to their place
        HIP_TRY(hipMemcpyAsync(out + (size_t)first * row_bytes, d_local, (size_t)mine * row_bytes, hipMemcpyDeviceToDevice, s));
    int rc = rccl.group_start();
    // peers in the order rank + 1, rank + 2, ...: at every step of the schedule each link pair is used once
    for (int k = 1; k < world && rc == 0; ++k) {
        const int to = (rank + k) % world, from = (rank - k + world) % world;
        int64_t f = 0, n = 0;
        ro_shard_rows(total_rows, world, from, &f, &n);
        if (mine > 0) rc = rccl.send(d_local, (size_t)mine * row_bytes, /*ncclInt8*/ 0, to, nccl_comm, s);
        if (rc == 0 && n > 0) rc = rccl.recv(out + (size_t)f * row_bytes, (size_t)n * row_bytes, 0, from, nccl_comm, s);
    }
    const int rc_end = rccl.group_end();
    if (rc != 0 || rc_end != 0) return fail(RO_ERR_HIP, "ncclSend / ncclRecv failed with code %d", rc ? rc : rc_end);
    return RO_OK;
}

extern "C" int ro_stitch_rows_device(const void *d_gathered, int64_t total_rows, int world, size_t row_bytes, void *d_out,
                                     void *stream)
{
    if (total_rows < 0 || world < 1 || (total_rows > 0 && (!d_gathered || !d_out)))
        return fail(RO_ERR_INVALID, "ro_stitch_rows_device: bad arguments");
    const int64_t block = ro_shard_max_rows(total_rows, world);
    for (int g = 0; g < world; ++g) {
        int64_t first = 0, rows = 0;
        ro_shard_rows(total_rows, world, g, &first, &rows);
        if (rows > 0)
            HIP_TRY(hipMemcpyAsync(static_cast<char *>(d_out) + (size_t)first * row_bytes,
                                   static_cast<const char *>(d_gathered) + (size_t)g * (size_t)block * row_bytes,
                                   (size_t)rows * row_bytes, hipMemcpyDeviceToDevice, (hipStream_t)stream));
    }
    return RO_OK;
}

// lengths that are not a power of two run as a chirp-z transform on the power-of-two length M >= 2 bins - 1 <= 2^20.
// Even lengths only: for an odd size the reference's processFFT leaves the last column of the row unwritten and
// writes one column twice (src/WaterfallBackend.cpp:489-505, halfSize = size / 2) -- there is no defined result to match.
static int czt_length(int bins)
{
    if (bins < 256 || bins >= (1 << 19) || (bins & 1) || (bins & (bins - 1)) == 0) return 0;
    int m = 512;
    while (m < 2 * bins - 1) m <<= 1;
    return m;
}

extern "C" int ro_bins_supported(int bins)
{
    return (ro::stft_supported(bins) || ro::big_supported(bins) || czt_length(bins) > 0) ? 1 : 0;
}

// ---------------------------------------------------------------------------
// handle
// ---------------------------------------------------------------------------
extern "C" int ro_stft_create(const ro_stft_config_t *cfg_in, ro_stft_t **out)
{
    if (!cfg_in || !out) return fail(RO_ERR_INVALID, "ro_stft_create: null argument");
    *out = nullptr;
    const ro_stft_config_t *cfg = cfg_in;          // re-pointed at a full-size copy once struct_size is known
    // ABI growth: fields are only ever appended; a caller built against ABI 1 passes the shorter struct and gets the
    // defaults (0) for what it does not know
    const size_t abi1_size = offsetof(ro_stft_config_t, precision);
    if (cfg->struct_size != sizeof(ro_stft_config_t) && cfg->struct_size != abi1_size)
        return fail(RO_ERR_INVALID, "ro_stft_create: struct_size %u is neither %zu (ABI 2) nor %zu (ABI 1)",
                    cfg->struct_size, sizeof(ro_stft_config_t), abi1_size);
    ro_stft_config_t cfg_full{};
    std::memcpy(&cfg_full, cfg_in, cfg_in->struct_size);
    cfg_full.struct_size = sizeof(ro_stft_config_t);
    cfg = &cfg_full;
    if (cfg->precision != RO_PRECISION_F32 && cfg->precision != RO_PRECISION_F64 && cfg->precision != RO_PRECISION_F64_ONE_LAUNCH)
        return fail(RO_ERR_INVALID, "unknown precision %d", cfg->precision);
    if (cfg->tile_ln != 0 && cfg->tile_ln != 1) return fail(RO_ERR_INVALID, "tile_ln must be 0 or 1");
    if (cfg->tile_ln && cfg->tile_cols <= 0) return fail(RO_ERR_INVALID, "tile_ln needs a tile (tile_cols > 0)");
    if (!ro_bins_supported(cfg->bins))
        return fail(RO_ERR_UNSUPPORTED, "bins = %d has no kernel (powers of two 256..1048576, other even lengths "
                                        "258..524286)", cfg->bins);
    if (czt_length(cfg->bins) && cfg->precision != RO_PRECISION_F32)
        return fail(RO_ERR_UNSUPPORTED, "RO_PRECISION_F64 is available for power-of-two bins only");
    if (cfg->iq_phase_shift != 0)
        return fail(RO_ERR_UNSUPPORTED, "iq_phase_shift != 0 is undefined behaviour in the reference "
                                        "(src/FFTBackend.cpp:67-71) and is not supported");
    if (cfg->sample_rate <= 0) return fail(RO_ERR_INVALID, "sample_rate must be positive");
    if (cfg->spare_cus_per_xcd < 0 || cfg->spare_cus_per_xcd > 16)
        return fail(RO_ERR_INVALID, "spare_cus_per_xcd must be in [0, 16]");
    if (cfg->window_kind == RO_WINDOW_CUSTOM && !cfg->window_table)
        return fail(RO_ERR_INVALID, "RO_WINDOW_CUSTOM needs window_table");
    if (cfg->window_kind < RO_WINDOW_NUTTALL || cfg->window_kind > RO_WINDOW_CUSTOM)
        return fail(RO_ERR_INVALID, "unknown window_kind %d", cfg->window_kind);
    if (cfg->tile_cols < 0 || cfg->tile_first_col < 0 || cfg->tile_first_col + cfg->tile_cols > cfg->bins)
        return fail(RO_ERR_INVALID, "tile [%d,+%d) outside [0,%d)", cfg->tile_first_col, cfg->tile_cols,
                    cfg->bins);

    int ndev = 0;
    HIP_TRY(hipGetDeviceCount(&ndev));
    if (cfg->device < 0 || cfg->device >= ndev)
        return fail(RO_ERR_HIP, "device %d not present (%d devices)", cfg->device, ndev);
    HIP_TRY(hipSetDevice(cfg->device));

    ro_stft *h = new (std::nothrow) ro_stft();
    if (!h) return fail(RO_ERR_NOMEM, "out of host memory");
    h->cfg = *cfg;
    h->cfg.window_table = nullptr;
    h->bins = cfg->bins;
    h->overlap = ro_clamp_overlap(cfg->bins, cfg->overlap);
    h->hop = h->bins - h->overlap;
    h->device = cfg->device;
    if (cfg->enable_scan) {
        int rc = check_bands(h, cfg->bands);
        if (rc != RO_OK) { delete h; return rc; }
    }

    hipDeviceProp_t prop;
    hipError_t e = hipGetDeviceProperties(&prop, cfg->device);
    if (e != hipSuccess) { delete h; return fail(RO_ERR_HIP, "hipGetDeviceProperties: %s", hipGetErrorString(e)); }
    h->device_name = prop.name;
    if (std::string(prop.gcnArchName).find("gfx950") == std::string::npos) {
        delete h;
        return fail(RO_ERR_UNSUPPORTED, "device %d is %s; this library is built for gfx950 only",
                    cfg->device, prop.gcnArchName);
    }
    // The N = 32768 kernel (and what is built on it) wants 132 KiB of LDS for ONE workgroup and spreads its grid over
    // eight XCDs (blockIdx % 8, round-robin dispatch): say so here rather than with an opaque launch error later.  A
    // partition mode with fewer XCDs per device only loses the placement (rows of one XCD's run no longer share an L2).
    if ((prop.maxSharedMemoryPerMultiProcessor > 0 && (size_t)prop.maxSharedMemoryPerMultiProcessor < (size_t)132 * 1024) ||
        prop.multiProcessorCount < 8) {
        const size_t lds = (size_t)prop.maxSharedMemoryPerMultiProcessor;
        const int cus = prop.multiProcessorCount;
        delete h;
        return fail(RO_ERR_UNSUPPORTED, "device %d offers %zu bytes of LDS per CU and %d CUs; the kernels need 135168 for one workgroup and 8",
                    cfg->device, lds, cus);
    }

    // ... and the persistent kernels hand rows out in eight runs, one per XCD (blockIdx % 8 under round-robin dispatch;
    // ro_stft32k.hip, ro_kernels.hip, ro_fourstep.hip).  The device says how many XCDs it has: a partition mode with
    // another number would still compute the same rows but lose the placement silently, so it is refused instead.  (A
    // runtime that does not know the attribute is taken at its word that this is a whole MI355X.)
    {
        int xccs = 0;
        if (hipDeviceGetAttribute(&xccs, hipDeviceAttributeNumberOfXccs, cfg->device) == hipSuccess) {
            if (xccs > 0 && xccs != 8) {                     // (0: a runtime that answers without knowing)
                delete h;
                return fail(RO_ERR_UNSUPPORTED, "device %d reports %d XCDs; the kernels' row placement is laid out for 8 "
                                                "(an MI355X in SPX mode)", cfg->device, xccs);
            }
        } else {
            (void)hipGetLastError();
        }
    }

    h->window.resize(h->bins);
    if (cfg->window_kind == RO_WINDOW_CUSTOM)
        std::memcpy(h->window.data(), cfg->window_table, sizeof(float) * h->bins);
    else
        build_window(cfg->window_kind, h->bins, h->window.data());
    h->big = ro::big_supported(h->bins);
    h->f64 = cfg->precision != RO_PRECISION_F32;
    h->f64_one_launch = cfg->precision == RO_PRECISION_F64_ONE_LAUNCH && ro::f64_fused_supported(cfg->bins);
    h->czt_m = czt_length(h->bins);
    h->czt = h->czt_m > 0;
    if (h->big && !h->f64) {
        // bins = dec x 32768, decimation in frequency on the largest single-pass plan:
        //   X[q + dec k'] = sum_m W_N^(m k') { W_bins^(m q) sum_r W_dec^(r q) w[m + N r] x[m + N r] }
        // 65536, 131072: ONE kernel (ro::stft_kernel MODE 3) that sums the row's dec blocks itself -- dec x the row
        // through L2 per output row, nothing extra through HBM (0.27 / 0.19 of the HBM peak; it was 0.11 / 0.06 at
        // 262144 / 524288, where its 4-byte stores `dec` floats apart cost an L2 write request each).
        // 262144 ... 1048576: the four-step pair of kernels (ro_fourstep.hip) for the magnitude rows.
        // Complex spectra of every large size: fold_kernel writes the braces to scratch, the N = 32768 kernel transforms
        // those rows, interleave2_kernel puts the bins in place (launch_spectra_big).  Rounds 2 and 3 made the magnitude
        // rows of dec >= 8 that way too (0.10-0.13 of the peak; profiles/r04_fourstep.txt has the A/B).
        // (Diagnostic builds: RO_BIG_FORM=dif forces the one-kernel form.)
        const int sub = 32768, dec = h->bins / sub;
        bool four = ro::fourstep_supported(h->bins);
#ifdef RO_DIAG_KNOBS
        if (const char *e = getenv("RO_BIG_FORM")) four = std::strcmp(e, "dif") != 0 && four;
#endif
        h->four = four;
        h->dif = !four;
        h->sub_bins = sub;
        h->dec = dec;
    }
    const int plan_bins = h->czt ? 0 : h->big ? h->sub_bins : h->bins;   // whose stage tables this handle needs (0: none)
    std::vector<float2> tw = plan_bins ? build_twiddles(plan_bins) : std::vector<float2>();
    if (plan_bins && (int)tw.size() != ro::stft_twiddle_count(plan_bins)) {
        delete h;
        return fail(RO_ERR_STATE, "internal: twiddle table size mismatch");
    }

#define CREATE_TRY(expr)                                                                   \
    do {                                                                                   \
        hipError_t e2_ = (expr);                                                           \
        if (e2_ != hipSuccess) {                                                           \
            int rc2_ = fail(RO_ERR_HIP, "%s failed: %s", #expr, hipGetErrorString(e2_));   \
            ro_stft_destroy(h);                                                            \
            return rc2_;                                                                   \
        }                                                                                  \
    } while (0)
    CREATE_TRY(hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking));
    CREATE_TRY(hipEventCreate(&h->ev0));
    CREATE_TRY(hipEventCreate(&h->ev1));
    CREATE_TRY(hipMalloc(&h->d_window, sizeof(float) * h->bins));
    CREATE_TRY(hipMalloc(&h->d_ln_keys, 16 * 2 * sizeof(unsigned)));
    CREATE_TRY(hipMalloc(&h->d_twiddles, sizeof(float2) * std::max<size_t>(tw.size(), 1)));
    CREATE_TRY(hipMemcpy(h->d_window, h->window.data(), sizeof(float) * h->bins, hipMemcpyHostToDevice));
    if (!h->big && !h->czt) {
        std::vector<float> wk((size_t)h->bins);
        if (!ro::stft_window_layout(h->bins, h->window.data(), wk.data())) {
            ro_stft_destroy(h);
            return fail(RO_ERR_UNSUPPORTED, "no window layout for bins=%d", h->bins);
        }
        CREATE_TRY(hipMalloc(&h->d_window_k, sizeof(float) * h->bins));
        CREATE_TRY(hipMemcpy(h->d_window_k, wk.data(), sizeof(float) * h->bins, hipMemcpyHostToDevice));
        if (h->bins == 32768) {
            ro::stft32k_window_layout(h->window.data(), wk.data());
            CREATE_TRY(hipMalloc(&h->d_window_k32, sizeof(float) * h->bins));
            CREATE_TRY(hipMemcpy(h->d_window_k32, wk.data(), sizeof(float) * h->bins, hipMemcpyHostToDevice));
        }
    }
    if (!tw.empty())
        CREATE_TRY(hipMemcpy(h->d_twiddles, tw.data(), sizeof(float2) * tw.size(), hipMemcpyHostToDevice));
    if (plan_bins) {
        const int units = ro::stft_packed_twiddle_count(plan_bins);
        std::vector<float4> pk((size_t)std::max(units, 1));
        if (units > 0) ro::stft_pack_twiddles(plan_bins, tw.data(), pk.data());
        CREATE_TRY(hipMalloc(&h->d_twiddles_k, sizeof(float4) * pk.size()));
        CREATE_TRY(hipMemcpy(h->d_twiddles_k, pk.data(), sizeof(float4) * pk.size(), hipMemcpyHostToDevice));
    }
    if (h->f64) {
        std::vector<double2> full = build_full_twiddles_f64(h->bins);
        CREATE_TRY(hipMalloc(&h->d_tw_f64, sizeof(double2) * full.size()));
        CREATE_TRY(hipMemcpy(h->d_tw_f64, full.data(), sizeof(double2) * full.size(), hipMemcpyHostToDevice));
        h->f64reg = ro::f64reg_supported(h->bins) && !h->f64_one_launch;
#ifdef RO_DIAG
        if (const char *e = getenv("RO_F64_HBM")) h->f64reg = h->f64reg && atoi(e) == 0;   // the through-HBM passes, for A/B
#endif
        if (h->f64reg) {
            ro::F64RegTables t;
            ro::f64reg_tables(h->bins, h->window.data(), t);
            CREATE_TRY(hipMalloc(&h->d_f64r_window, sizeof(float) * t.window_k.size()));
            CREATE_TRY(hipMemcpy(h->d_f64r_window, t.window_k.data(), sizeof(float) * t.window_k.size(), hipMemcpyHostToDevice));
            const std::vector<double2> *tabs[4] = {&t.tw0, &t.tw1, &t.tw2, &t.tw3};
            for (int i = 0; i < 4; ++i) {
                if (tabs[i]->empty()) continue;
                CREATE_TRY(hipMalloc(&h->d_f64r_tw[i], sizeof(double2) * tabs[i]->size()));
                CREATE_TRY(hipMemcpy(h->d_f64r_tw[i], tabs[i]->data(), sizeof(double2) * tabs[i]->size(), hipMemcpyHostToDevice));
            }
        }
    }
    if (h->big && h->dec > 1) {
        // the rotations W_bins^(q m), [dec][sub_bins], each rounded once from long double
        std::vector<float2> tc((size_t)h->bins);
        const long double two_pi = 8.0L * atanl(1.0L);
        for (int r = 0; r < h->dec; ++r)
            for (int k = 0; k < h->sub_bins; ++k) {
                const long double ang = -two_pi * (long double)((long long)r * k) / (long double)h->bins;
                tc[(size_t)r * h->sub_bins + k] = make_float2((float)cosl(ang), (float)sinl(ang));
            }
        CREATE_TRY(hipMalloc(&h->d_tw_combine, sizeof(float2) * tc.size()));
        CREATE_TRY(hipMemcpy(h->d_tw_combine, tc.data(), sizeof(float2) * tc.size(), hipMemcpyHostToDevice));
    }
    if (h->big && h->dec > 1) {
        std::vector<float> ones((size_t)h->sub_bins, 1.0f);
        CREATE_TRY(hipMalloc(&h->d_ones, sizeof(float) * ones.size()));
        CREATE_TRY(hipMemcpy(h->d_ones, ones.data(), sizeof(float) * ones.size(), hipMemcpyHostToDevice));
    }
    if (h->dif) {
        std::vector<float> wk((size_t)h->bins);
        for (int r = 0; r < h->dec; ++r)
            if (!ro::stft_window_layout(h->sub_bins, h->window.data() + (size_t)r * h->sub_bins,
                                        wk.data() + (size_t)r * h->sub_bins)) {
                ro_stft_destroy(h);
                return fail(RO_ERR_UNSUPPORTED, "no window layout for bins=%d", h->sub_bins);
            }
        CREATE_TRY(hipMalloc(&h->d_window_dif, sizeof(float) * wk.size()));
        CREATE_TRY(hipMemcpy(h->d_window_dif, wk.data(), sizeof(float) * wk.size(), hipMemcpyHostToDevice));
        std::vector<float2> td((size_t)h->dec);
        const long double two_pi = 8.0L * atanl(1.0L);
        for (int j = 0; j < h->dec; ++j) {
            // (quarter turns exact: cosl(pi/2) is 6e-20, not 0)
            const long double ang = -two_pi * (long double)j / (long double)h->dec;
            long double c = cosl(ang), sn = sinl(ang);
            if ((4 * j) % h->dec == 0) { c = roundl(c); sn = roundl(sn); }
            td[(size_t)j] = make_float2((float)c, (float)sn);
        }
        CREATE_TRY(hipMalloc(&h->d_dif_tw, sizeof(float2) * td.size()));
        CREATE_TRY(hipMemcpy(h->d_dif_tw, td.data(), sizeof(float2) * td.size(), hipMemcpyHostToDevice));
        // residue q's rotation W_bins^(m q) as a shift of the bin index by q / dec: per stage (32, 32 x 32, 32^3 points
        // behind it) the powers 1, 2, 4, 8, 16 of exp(-2 pi i (q / dec) / M), each rounded once from long double
        std::vector<float2> ts((size_t)h->dec * 16, make_float2(1.0f, 0.0f));
        const long double span[3] = {32.0L, 1024.0L, 32768.0L};
        for (int q = 0; q < h->dec; ++q)
            for (int st = 0; st < 3; ++st)
                for (int i = 0; i < 5; ++i) {
                    const long double ang = -two_pi * ((long double)q / (long double)h->dec) * (long double)(1 << i) / span[st];
                    ts[(size_t)q * 16 + st * 5 + i] = make_float2((float)cosl(ang), (float)sinl(ang));
                }
        CREATE_TRY(hipMalloc(&h->d_dif_shift, sizeof(float2) * ts.size()));
        CREATE_TRY(hipMemcpy(h->d_dif_shift, ts.data(), sizeof(float2) * ts.size(), hipMemcpyHostToDevice));
    }
    if (h->four) {
        std::vector<float> wa;
        std::vector<float2> ta, tb, tr;
        ro::fourstep_tables(h->bins, h->window.data(), wa, ta, tb, tr);
        CREATE_TRY(hipMalloc(&h->d_four_window, sizeof(float) * wa.size()));
        CREATE_TRY(hipMemcpy(h->d_four_window, wa.data(), sizeof(float) * wa.size(), hipMemcpyHostToDevice));
        CREATE_TRY(hipMalloc(&h->d_four_tw_a, sizeof(float2) * ta.size()));
        CREATE_TRY(hipMemcpy(h->d_four_tw_a, ta.data(), sizeof(float2) * ta.size(), hipMemcpyHostToDevice));
        CREATE_TRY(hipMalloc(&h->d_four_tw_b, sizeof(float2) * tb.size()));
        CREATE_TRY(hipMemcpy(h->d_four_tw_b, tb.data(), sizeof(float2) * tb.size(), hipMemcpyHostToDevice));
        CREATE_TRY(hipMalloc(&h->d_four_tw_r, sizeof(float2) * tr.size()));
        CREATE_TRY(hipMemcpy(h->d_four_tw_r, tr.data(), sizeof(float2) * tr.size(), hipMemcpyHostToDevice));
    }
    if (h->czt) {
        const int N = h->bins, M = h->czt_m;
        // the inner handle: length M, overlap 0, a window of ones, no bands / tile
        {
            std::vector<float> ones((size_t)M, 1.0f);
            ro_stft_config_t ic{};
            ic.struct_size = sizeof ic;
            ic.bins = M;
            ic.overlap = 0;
            ic.sample_rate = cfg->sample_rate;
            ic.window_kind = RO_WINDOW_CUSTOM;
            ic.window_table = ones.data();
            ic.device = cfg->device;
            ic.spare_cus_per_xcd = cfg->spare_cus_per_xcd;
            int rc = ro_stft_create(&ic, &h->inner);
            if (rc != RO_OK) { ro_stft_destroy(h); return rc; }
        }
        // chirp c[i] = exp(-pi i i^2 / N), the angle reduced exactly: i^2 mod 2N in integers
        const long double pi = 4.0L * atanl(1.0L);
        auto chirp = [&](int64_t i) {
            const int64_t r = (i * i) % (2 * (int64_t)N);
            const long double ang = -pi * (long double)r / (long double)N;
            return std::complex<double>((double)cosl(ang), (double)sinl(ang));
        };
        std::vector<float2> cw((size_t)N);
        for (int i = 0; i < N; ++i) {
            const std::complex<double> c = chirp(i) * (double)h->window[(size_t)i];
            cw[(size_t)i] = make_float2((float)c.real(), (float)c.imag());
        }
        // B = FFT_M(conj(c) wrapped around M) in double on the host, once; the kernels use conj(B) / M
        std::vector<std::complex<double>> b((size_t)M, std::complex<double>(0.0, 0.0));
        b[0] = std::conj(chirp(0));
        for (int i = 1; i < N; ++i) b[(size_t)i] = b[(size_t)(M - i)] = std::conj(chirp(i));
        host_fft(b);
        std::vector<float2> bc((size_t)M);
        for (int i = 0; i < M; ++i)
            bc[(size_t)i] = make_float2((float)(b[(size_t)i].real() / M), (float)(-b[(size_t)i].imag() / M));
        CREATE_TRY(hipSetDevice(cfg->device));
        CREATE_TRY(hipMalloc(&h->d_cw, sizeof(float2) * cw.size()));
        CREATE_TRY(hipMemcpy(h->d_cw, cw.data(), sizeof(float2) * cw.size(), hipMemcpyHostToDevice));
        CREATE_TRY(hipMalloc(&h->d_bc, sizeof(float2) * bc.size()));
        CREATE_TRY(hipMemcpy(h->d_bc, bc.data(), sizeof(float2) * bc.size(), hipMemcpyHostToDevice));
    }
#undef CREATE_TRY

    // what travels to the host in the streaming path: the tile when one is configured, else whole rows
    h->out_first = cfg->tile_cols > 0 ? cfg->tile_first_col : 0;
    h->out_cols = cfg->tile_cols > 0 ? cfg->tile_cols : h->bins;
    // streaming buffers are allocated lazily by the first push
    h->batch_rows = cfg->max_batch_rows > 0 ? cfg->max_batch_rows
                                            : std::max(1, (64 << 20) / (h->bins * 4));   // ~64 MiB of rows
#ifdef RO_DIAG_KNOBS
    if (const char *e = getenv("RO_GRAPH_TIME_EVERY")) h->diag_time_every = atoi(e);       // tools/r5/host_calls_ab.py
    if (const char *e = getenv("RO_GRAPH_TWO_EVENTS")) h->diag_done_only = atoi(e) == 0;
    if (const char *e = getenv("RO_GRAPH_DIRECT")) h->diag_direct = atoi(e);
#endif
    *out = h;
    return RO_OK;
}

extern "C" int ro_stft_destroy(ro_stft_t *h)
{
    if (!h) return RO_OK;
    (void)hipSetDevice(h->device);
    if (h->s_in) (void)hipStreamSynchronize(h->s_in);
    if (h->stream) (void)hipStreamSynchronize(h->stream);
    if (h->s_out) (void)hipStreamSynchronize(h->s_out);
    free_stream_slots(h);
    if (h->d_window) (void)hipFree(h->d_window);
    if (h->d_window_k) (void)hipFree(h->d_window_k);
    if (h->d_window_k32) (void)hipFree(h->d_window_k32);
    if (h->d_ln_keys) (void)hipFree(h->d_ln_keys);
    if (h->d_twiddles) (void)hipFree(h->d_twiddles);
    if (h->d_twiddles_k) (void)hipFree(h->d_twiddles_k);
    if (h->d_stamps) (void)hipFree(h->d_stamps);
    while (!h->ready.empty()) {
        destroy_batch(h->ready.front());
        h->ready.pop_front();
    }
    for (Batch *b : h->batch_pool) destroy_batch(b);
    if (h->d_tw_combine) (void)hipFree(h->d_tw_combine);
    if (h->d_spec) (void)hipFree(h->d_spec);
    if (h->d_window_dif) (void)hipFree(h->d_window_dif);
    if (h->inner) (void)ro_stft_destroy(h->inner);
    if (h->d_cw) (void)hipFree(h->d_cw);
    if (h->d_bc) (void)hipFree(h->d_bc);
    if (h->d_czt_a) (void)hipFree(h->d_czt_a);
    if (h->d_czt_A) (void)hipFree(h->d_czt_A);
    if (h->d_czt_mag) (void)hipFree(h->d_czt_mag);
    if (h->d_spec2) (void)hipFree(h->d_spec2);
    if (h->d_four_window) (void)hipFree(h->d_four_window);
    if (h->d_four_tw_a) (void)hipFree(h->d_four_tw_a);
    if (h->d_four_tw_b) (void)hipFree(h->d_four_tw_b);
    if (h->d_four_tw_r) (void)hipFree(h->d_four_tw_r);
    if (h->d_four_z) (void)hipFree(h->d_four_z);
    if (h->d_ones) (void)hipFree(h->d_ones);
    if (h->d_dif_tw) (void)hipFree(h->d_dif_tw);
    if (h->d_dif_shift) (void)hipFree(h->d_dif_shift);
    if (h->d_tw_f64) (void)hipFree(h->d_tw_f64);
    if (h->d_f64r_window) (void)hipFree(h->d_f64r_window);
    for (int i = 0; i < 4; ++i)
        if (h->d_f64r_tw[i]) (void)hipFree(h->d_f64r_tw[i]);
    if (h->d_ln_part) (void)hipFree(h->d_ln_part);
    for (int i = 0; i < 2; ++i)
        if (h->d_scratch_d[i]) (void)hipFree(h->d_scratch_d[i]);
    if (h->d_f64_ring) (void)hipFree(h->d_f64_ring);
    if (h->d_f64_ctl) (void)hipFree(h->d_f64_ctl);
    if (h->h_f64_err) (void)hipHostFree(h->h_f64_err);
    if (h->ev0) (void)hipEventDestroy(h->ev0);
    if (h->ev1) (void)hipEventDestroy(h->ev1);
    if (h->stream) (void)hipStreamDestroy(h->stream);
    delete h;
    return RO_OK;
}

// Diagnostic hook (not in include/ro_stft.h): allocate / read the per-workgroup phase stamps
// that a -DRO_STAMPS=1 build of the kernels fills.  A normal build never writes them.
extern "C" int ro_stft_debug_stamps(ro_stft_t *h, unsigned long long *out, int max_words)
{
    if (!h) return fail(RO_ERR_INVALID, "null handle");
    const int words = 4096 * 16;
    if (!h->d_stamps) {
        HIP_TRY(hipSetDevice(h->device));
        HIP_TRY(hipMalloc(&h->d_stamps, sizeof(unsigned long long) * words));
        HIP_TRY(hipMemset(h->d_stamps, 0, sizeof(unsigned long long) * words));
        return RO_OK;
    }
    if (out) {
        HIP_TRY(hipDeviceSynchronize());
        HIP_TRY(hipMemcpy(out, h->d_stamps, sizeof(unsigned long long) * std::min(words, max_words),
                          hipMemcpyDeviceToHost));
    }
    return RO_OK;
}

extern "C" int ro_stft_get_window(const ro_stft_t *h, float *out)
{
    if (!h || !out) return fail(RO_ERR_INVALID, "null argument");
    std::memcpy(out, h->window.data(), sizeof(float) * h->bins);
    return RO_OK;
}

extern "C" int ro_stft_hop(const ro_stft_t *h) { return h ? h->hop : fail(RO_ERR_INVALID, "null handle"); }
extern "C" int ro_stft_bins(const ro_stft_t *h) { return h ? h->bins : fail(RO_ERR_INVALID, "null handle"); }

extern "C" int ro_stft_device_name(const ro_stft_t *h, char *buf, size_t len)
{
    if (!h || !buf || len == 0) return fail(RO_ERR_INVALID, "null argument");
    std::snprintf(buf, len, "%s", h->device_name.c_str());
    return RO_OK;
}

extern "C" int ro_stft_set_bands(ro_stft_t *h, const ro_bands_t *bands)
{
    if (!h || !bands) return fail(RO_ERR_INVALID, "null argument");
    int rc = check_bands(h, *bands);
    if (rc != RO_OK) return rc;
    h->cfg.bands = *bands;
    h->cfg.enable_scan = 1;
    return RO_OK;
}

// ---------------------------------------------------------------------------
// resident path
// ---------------------------------------------------------------------------
extern "C" int ro_stft_run_resident(ro_stft_t *h, const void *d_iq, int format, int64_t samples,
                                    int64_t first_row, int64_t rows, float *d_rows, int64_t row_stride,
                                    float *d_tile, ro_scan_record_t *d_records, void *stream)
{
    int rc = validate_resident(h, d_iq, format, samples, first_row, rows, d_rows, row_stride, d_tile,
                               d_records);
    if (rc != RO_OK || rows == 0) return rc;
    HIP_TRY(hipSetDevice(h->device));
    hipStream_t s = (hipStream_t)stream;      // NULL = the default (null) stream, like any HIP launch
    rc = launch_transform(h, d_iq, format, first_row, rows, d_rows, row_stride, s, d_tile, d_records);
    if (rc != RO_OK) return rc;
    rc = launch_tile_and_scan(h, d_rows, row_stride, rows, d_tile, d_records, s);
    if (rc != RO_OK) return rc;
    h->stat_launches += 1;
    h->stat_rows += rows;
    return RO_OK;
}

extern "C" int ro_stft_run_resident_ln(ro_stft_t *h, const void *d_iq, int format, int64_t samples, int64_t first_row,
                                       int64_t rows, float *d_rows, int64_t row_stride, float *d_tile, float *d_ln_tile,
                                       float *d_ln_minmax, ro_scan_record_t *d_records, void *stream)
{
    int rc = validate_resident(h, d_iq, format, samples, first_row, rows, d_rows, row_stride, d_tile, d_records);
    if (rc != RO_OK || rows == 0) return rc;
    if (!h->cfg.tile_ln) return fail(RO_ERR_STATE, "this handle was not created with tile_ln");
    if (!d_tile) return fail(RO_ERR_INVALID, "the log image is cut from the tile: d_tile is required");
    if (d_ln_minmax && !d_ln_tile) return fail(RO_ERR_INVALID, "the range comes with the log image: pass d_ln_tile too");
    HIP_TRY(hipSetDevice(h->device));
    hipStream_t s = (hipStream_t)stream;
    // the fused epilogue always writes the log when asked for its range (the partial min / max come with it)
    float *ln = d_ln_tile;
    rc = launch_transform(h, d_iq, format, first_row, rows, d_rows, row_stride, s, d_tile, d_records, ln);
    if (rc != RO_OK) return rc;
    rc = launch_tile_and_scan(h, d_rows, row_stride, rows, d_tile, d_records, s, ln, d_ln_minmax);
    if (rc != RO_OK) return rc;
    h->stat_launches += 1;
    h->stat_rows += rows;
    return RO_OK;
}

extern "C" int ro_ln_levels(const float *ln, int64_t count, float mn, float mx, uint8_t *levels_out)
{
    if (count < 0 || (count > 0 && (!ln || !levels_out))) return fail(RO_ERR_INVALID, "ro_ln_levels: bad arguments");
    const float span = mx - mn;
    for (int64_t i = 0; i < count; ++i) {
        // float32 throughout, like numpy in the viewer (fits2png:444-445); -inf = a zero pixel (dropped there)
        const float level = (ln[i] - mn) / span * 255.f;
        levels_out[i] = (std::isfinite(ln[i]) && span > 0.f) ? (uint8_t)(int)level : (uint8_t)0;
    }
    return RO_OK;
}

extern "C" int ro_stft_spectra_resident(ro_stft_t *h, const void *d_iq, int format, int64_t samples,
                                        int64_t first_row, int64_t rows, float *d_spectra, int64_t stride,
                                        void *stream)
{
    // same argument checks as the magnitude path (d_spectra in the place of d_rows)
    int rc = validate_resident(h, d_iq, format, samples, first_row, rows, d_spectra, stride, nullptr, nullptr);
    if (rc != RO_OK) return rc;
    if (h->czt) return fail(RO_ERR_UNSUPPORTED, "complex spectra are available for power-of-two bins");
    if (h->f64) return fail(RO_ERR_UNSUPPORTED, "complex spectra are float32 only (RO_PRECISION_F32)");
    HIP_TRY(hipSetDevice(h->device));
    if (h->big)
        return launch_spectra_big(h, d_iq, format, first_row, rows, reinterpret_cast<float2 *>(d_spectra), stride,
                                  (hipStream_t)stream);
    ro::StftArgs a = make_stft_args(h, d_iq, first_row, rows, nullptr, 0);
    a.spec_out = reinterpret_cast<float2 *>(d_spectra);
    a.spec_stride = stride;
    HIP_TRY(ro::launch_stft(h->bins, format, a, (hipStream_t)stream));
    return RO_OK;
}

extern "C" int ro_stft_scan_resident(ro_stft_t *h, const float *d_rows, int64_t row_stride, int64_t rows,
                                     ro_scan_record_t *d_records, void *stream)
{
    if (!h || !d_rows || !d_records) return fail(RO_ERR_INVALID, "null argument");
    if (!h->cfg.enable_scan) return fail(RO_ERR_STATE, "scan bands not configured");
    if (rows < 0 || row_stride < h->bins) return fail(RO_ERR_INVALID, "bad rows / row_stride");
    HIP_TRY(hipSetDevice(h->device));
    hipStream_t s = (hipStream_t)stream;      // NULL = the default (null) stream, like any HIP launch
    ro::ScanArgs sc = make_scan_args(h, d_rows, row_stride, rows, d_records);
    HIP_TRY(ro::launch_scan(sc, s));
    return RO_OK;
}

extern "C" int ro_stft_ln_tile_resident(ro_stft_t *h, const float *d_rows, int64_t row_stride, int64_t rows,
                                        int first_col, int cols, float *d_ln, uint8_t *d_u8, float *d_minmax,
                                        void *stream)
{
    if (!h || !d_rows) return fail(RO_ERR_INVALID, "null argument");
    if (!d_ln && !d_u8 && !d_minmax) return fail(RO_ERR_INVALID, "no output requested");
    if (rows < 0 || row_stride < h->bins) return fail(RO_ERR_INVALID, "bad rows / row_stride");
    if (first_col < 0 || cols <= 0 || (int64_t)first_col + cols > h->bins)
        return fail(RO_ERR_INVALID, "columns outside the row");
    HIP_TRY(hipSetDevice(h->device));
    ro::LnArgs a;
    a.rows_in = d_rows;
    a.ln_out = d_ln;
    a.u8_out = d_u8;
    a.keys = h->d_ln_keys + 2 * (h->ln_calls++ & 15);      // a pair of its own for each of 16 calls in flight
    a.minmax = d_minmax;
    a.rows = rows;
    a.row_stride = row_stride;
    a.first = first_col;
    a.cols = cols;
    HIP_TRY(ro::launch_ln_tile(a, (hipStream_t)stream));
    return RO_OK;
}

extern "C" int ro_stft_time_resident(ro_stft_t *h, const void *d_iq, int format, int64_t samples,
                                     int64_t first_row, int64_t rows, float *d_rows, int64_t row_stride,
                                     float *d_tile, ro_scan_record_t *d_records, void *stream, int iters,
                                     float *ms_out, float *kernel_ms_out)
{
    int rc = validate_resident(h, d_iq, format, samples, first_row, rows, d_rows, row_stride, d_tile,
                               d_records);
    if (rc != RO_OK) return rc;
    if (iters <= 0 || !ms_out) return fail(RO_ERR_INVALID, "iters must be positive and ms_out non-null");
    HIP_TRY(hipSetDevice(h->device));
    hipStream_t s = (hipStream_t)stream;      // NULL = the default (null) stream, like any HIP launch
    std::vector<hipEvent_t> ev((size_t)iters * 3, nullptr);
    for (auto &e : ev) {
        if (hipEventCreate(&e) != hipSuccess) {
            for (auto &d : ev) if (d) (void)hipEventDestroy(d);
            return fail(RO_ERR_HIP, "hipEventCreate failed");
        }
    }
    for (int i = 0; i < iters && rc == RO_OK; ++i) {
        hipError_t e = hipEventRecord(ev[3 * i], s);
        if (e == hipSuccess) {
            rc = launch_transform(h, d_iq, format, first_row, rows, d_rows, row_stride, s, d_tile, d_records);
            if (rc != RO_OK) break;
            e = hipEventRecord(ev[3 * i + 1], s);
        }
        if (e == hipSuccess) {
            rc = launch_tile_and_scan(h, d_rows, row_stride, rows, d_tile, d_records, s);
            if (rc != RO_OK) break;
            e = hipEventRecord(ev[3 * i + 2], s);
        }
        if (e != hipSuccess) rc = fail(RO_ERR_HIP, "hipEventRecord failed: %s", hipGetErrorString(e));
    }
    if (rc != RO_OK) {                                  // leave no event behind on the error paths
        (void)hipStreamSynchronize(s);
        for (auto &e : ev) (void)hipEventDestroy(e);
        return rc;
    }
    hipError_t es = hipStreamSynchronize(s);
    double k0 = 0.0, k1 = 0.0;
    for (int i = 0; i < iters && es == hipSuccess; ++i) {
        float t = 0.f, t0 = 0.f, t1 = 0.f;
        if ((es = hipEventElapsedTime(&t, ev[3 * i], ev[3 * i + 2])) != hipSuccess) break;
        if ((es = hipEventElapsedTime(&t0, ev[3 * i], ev[3 * i + 1])) != hipSuccess) break;
        if ((es = hipEventElapsedTime(&t1, ev[3 * i + 1], ev[3 * i + 2])) != hipSuccess) break;
        ms_out[i] = t;
        k0 += t0;
        k1 += t1;
    }
    if (es != hipSuccess) {
        for (auto &e : ev) (void)hipEventDestroy(e);
        return fail(RO_ERR_HIP, "timing the resident path failed: %s", hipGetErrorString(es));
    }
    if (kernel_ms_out) {
        kernel_ms_out[0] = (float)(k0 / iters);
        kernel_ms_out[1] = (float)(k1 / iters);
    }
    for (auto &e : ev) (void)hipEventDestroy(e);
    h->stat_launches += iters;
    h->stat_rows += rows * (int64_t)iters;
    return RO_OK;
}

// ---------------------------------------------------------------------------
// streaming path
// ---------------------------------------------------------------------------
extern "C" int ro_stft_push(ro_stft_t *h, const void *iq, int format, int64_t samples, int64_t *rows_ready)
{
    if (!h) return fail(RO_ERR_INVALID, "null handle");
    if (samples < 0 || (samples > 0 && !iq)) return fail(RO_ERR_INVALID, "bad sample buffer");
    if (format != RO_IQ_F32 && format != RO_IQ_I16 && format != RO_IQ_F64)
        return fail(RO_ERR_INVALID, "unknown sample format %d", format);
    const double t0 = now_ms();
    int rc = ensure_stream_slots(h);
    if (rc != RO_OK) return rc;

    // The caller's buffer is only valid during the call (src/WAVStream.cpp:113,123): copy now.  int16 samples stay
    // int16 all the way to the kernel (half the staging memory and PCIe bytes: src/WAVStream.cpp:119-120 hands them
    // over un-normalised, the kernel widens them); float32 and the double Complex are staged as float32 (lossless for
    // every frontend of the reference).  A stream that changes format mid-way is widened to float32 once.
    const bool in_i16 = format == RO_IQ_I16;
    const size_t cap = (size_t)(h->batch_rows - 1) * h->hop + h->bins;        // samples one slot's staging buffer holds
    if (!h->stage_fmt_set) {
        h->stage_fmt = in_i16 ? RO_IQ_I16 : RO_IQ_F32;
        h->stage_fmt_set = true;
    } else if (h->stage_fmt == RO_IQ_I16 && !in_i16) {
        // widen what is staged, in place and from the back (the buffer is sized for 8 bytes per sample)
        char *base = static_cast<char *>(h->slot[h->batch_seq % RO_STREAM_SLOTS].h_in);
        const int16_t *src = reinterpret_cast<const int16_t *>(base);
        float *dst = reinterpret_cast<float *>(base);
        for (size_t i = h->staged_have * 2; i-- > 0;) dst[i] = (float)src[i];
        h->stage_fmt = RO_IQ_F32;
    }
    // With a row sink a push is all or nothing: the batches this call would complete are counted BEFORE anything is
    // staged, and a call whose rows would lap rows that still wait to be fetched is refused whole -- no sample taken,
    // no counter moved -- so the caller fetches and pushes the same buffer again (the streaming analogue of
    // RingBuffer2D::push never overwriting a reserved row silently, src/RingBuffer.h:482-509).
    if (h->sink) {
        const size_t spent = (size_t)h->batch_rows * h->hop;               // samples a batch retires
        size_t have = h->staged_have;
        int64_t batches = 0;
        for (int64_t left = samples; left > 0;) {
            const int64_t take = std::min<int64_t>(left, (int64_t)(cap - have));
            have += (size_t)take;
            left -= take;
            if (have == cap) { ++batches; have -= spent; }
        }
        if (h->rows_ready + batches * (int64_t)h->batch_rows > h->sink_cap)
            return fail(RO_ERR_STATE, "row sink full: this push would complete %lld rows with %lld waiting to be fetched in a "
                                      "ring of %lld slots; nothing was consumed -- fetch, then push the same samples again",
                        (long long)(batches * h->batch_rows), (long long)h->rows_ready, (long long)h->sink_cap);
    }
    const size_t sb = stage_sample_bytes(h);
    const size_t isb = format == RO_IQ_F64 ? 16 : format == RO_IQ_F32 ? 8 : 4;       // bytes per sample as delivered
    const char *in = static_cast<const char *>(iq);
    h->stat_samples += samples;
    for (int64_t left = samples; left > 0;) {
        // into the pinned buffer the next upload reads, converting on the way (no second copy)
        char *dstb = static_cast<char *>(h->slot[h->batch_seq % RO_STREAM_SLOTS].h_in) + h->staged_have * sb;
        const int64_t take = std::min<int64_t>(left, (int64_t)(cap - h->staged_have));
        if (h->stage_fmt == RO_IQ_I16) {
            std::memcpy(dstb, in, (size_t)take * 4);
        } else {
            float *dst = reinterpret_cast<float *>(dstb);
            if (format == RO_IQ_F32) {
                std::memcpy(dst, in, (size_t)take * 2 * sizeof(float));
            } else if (in_i16) {
                const int16_t *src = reinterpret_cast<const int16_t *>(in);
                for (int64_t i = 0; i < take * 2; ++i) dst[i] = (float)src[i];
            } else {
                const double *src = reinterpret_cast<const double *>(in);         // struct Complex
                // (a slot of a few hundred KiB -- a latency-bound batch -- stays in the caches between the calls that
                // fill it and the overlap copy that reads it back; one of many MiB does not, and is written past them)
                // (tools/r5/host_nt.py: at a slot of 590 KiB -- the Backend's default batch -- the two forms cannot be told
                // apart: 1.15 ... 1.52 x 10^5 rows/s with either, from one process to the next)
                size_t nt_from = (size_t)8 << 20;
#ifdef RO_DIAG_KNOBS
                if (const char *e = getenv("RO_STAGE_NT_BYTES")) nt_from = (size_t)atoll(e);
#endif
                const bool past_caches = cap * sb > nt_from;
                for (int64_t at = 0; at < take * 2; at += (int64_t)1 << 30) {     // (the loops count in int)
                    const int n = (int)std::min<int64_t>(take * 2 - at, (int64_t)1 << 30);
                    if (past_caches) ro::narrowToFloatStream(src + at, dst + at, n);
                    else ro::narrowToFloat(src + at, dst + at, n);
                }
            }
        }
        h->staged_have += (size_t)take;
        in += (size_t)take * isb;
        left -= take;
        if (h->staged_have == cap) {                                      // = batch_rows complete rows
            rc = run_stream_batch(h, h->batch_rows);
            if (rc != RO_OK) return rc;
        }
    }
    if (rows_ready) *rows_ready = h->rows_ready;
    const double dt = now_ms() - t0;
    h->timing.push_calls += 1;
    h->push_ms_sum += dt;
    h->timing.push_ms_max = std::max(h->timing.push_ms_max, dt);
    return RO_OK;
}

extern "C" int ro_stft_flush(ro_stft_t *h, int64_t *rows_ready)
{
    if (!h) return fail(RO_ERR_INVALID, "null handle");
    while (h->slots_ready) {
        const int64_t n = std::min<int64_t>(staged_complete_rows(h), h->batch_rows);
        if (n <= 0) break;
        int rc = run_stream_batch(h, n);
        if (rc != RO_OK) return rc;
    }
    if (rows_ready) *rows_ready = h->rows_ready;
    return RO_OK;
}

extern "C" int ro_stft_fetch(ro_stft_t *h, int64_t max_rows, int first_col, int cols, float *rows_out,
                             ro_scan_record_t *records_out, int64_t *first_row_index, int64_t *rows_got)
{
    if (!h || !rows_got) return fail(RO_ERR_INVALID, "null argument");
    if (max_rows < 0) return fail(RO_ERR_INVALID, "negative max_rows");
    if (rows_out && (first_col < h->out_first || cols <= 0 || first_col + cols > h->out_first + h->out_cols))
        return fail(RO_ERR_INVALID, "columns [%d,+%d) outside [%d,+%d) -- what this handle brings to the host%s",
                    first_col, cols, h->out_first, h->out_cols,
                    h->cfg.tile_cols > 0 ? " (the configured tile)" : "");
    if (records_out && !h->cfg.enable_scan) return fail(RO_ERR_STATE, "scan records requested but scan is off");
    if (rows_out && h->sink) return fail(RO_ERR_STATE, "this handle's rows go to its row sink (ro_stft_set_row_sink): pass rows_out = NULL");
    const double t0 = now_ms();
    int64_t got = 0;
    if (first_row_index) *first_row_index = h->rows_emitted;
    if (first_row_index && !h->ready.empty())
        *first_row_index = h->ready.front()->first_row + h->ready.front()->consumed;
    while (got < max_rows && !h->ready.empty()) {
        Batch *b = h->ready.front();
        { const int rc = await_batch(h, b); if (rc != RO_OK) return rc; }
        const int64_t take = std::min(max_rows - got, b->rows - b->consumed);
        for (int64_t r = 0; r < take; ++r) {
            if (rows_out) {
                const float *src = b->data + (size_t)(b->consumed + r) * h->out_cols + (first_col - h->out_first);
                std::memcpy(rows_out + (size_t)(got + r) * cols, src, sizeof(float) * cols);
            }
            if (records_out) records_out[got + r] = b->records[(size_t)(b->consumed + r)];
        }
        b->consumed += take;
        got += take;
        if (b->consumed == b->rows) {
            h->ready.pop_front();
            release_batch(h, b);
        }
    }
    h->rows_ready -= got;
    *rows_got = got;
    const double dt = now_ms() - t0;
    h->timing.fetch_calls += 1;
    h->fetch_ms_sum += dt;
    h->timing.fetch_ms_max = std::max(h->timing.fetch_ms_max, dt);
    return RO_OK;
}

// rows at the head of the output queue whose batches have FINISHED (download included): what ro_stft_fetch hands over
// without waiting.  A caller that fetches only these keeps the next batch's upload and kernels in flight under the
// previous batch's download and under its own per-row work, instead of waiting out every batch it has just launched.
extern "C" int ro_stft_rows_complete(ro_stft_t *h, int64_t *rows)
{
    if (!h || !rows) return fail(RO_ERR_INVALID, "null argument");
    int64_t n = 0;
    for (Batch *b : h->ready) {
        if (b->pending) {
            const hipError_t e = hipEventQuery(b->done);
            if (e == hipErrorNotReady) break;
            if (e != hipSuccess) return fail(RO_ERR_HIP, "hipEventQuery failed: %s", hipGetErrorString(e));
            const int rc = await_batch(h, b);        // finished: book its kernel time once, never query it again
            if (rc != RO_OK) return rc;
        }
        n += b->rows - b->consumed;
    }
    *rows = n;
    return RO_OK;
}

extern "C" int ro_stft_fetch_ln(ro_stft_t *h, int64_t max_rows, float *tile_out, float *ln_out, float *minmax_out,
                                ro_scan_record_t *records_out, int64_t *first_row_index, int64_t *rows_got)
{
    if (!h || !rows_got) return fail(RO_ERR_INVALID, "null argument");
    if (!h->cfg.tile_ln) return fail(RO_ERR_STATE, "this handle was not created with tile_ln");
    if (max_rows < 0) return fail(RO_ERR_INVALID, "negative max_rows");
    if (records_out && !h->cfg.enable_scan) return fail(RO_ERR_STATE, "scan records requested but scan is off");
    const double t0 = now_ms();
    int64_t got = 0;
    if (first_row_index) *first_row_index = h->rows_emitted;
    if (first_row_index && !h->ready.empty())
        *first_row_index = h->ready.front()->first_row + h->ready.front()->consumed;
    const size_t w = (size_t)h->out_cols;
    while (got < max_rows && !h->ready.empty()) {
        Batch *b = h->ready.front();
        { const int rc = await_batch(h, b); if (rc != RO_OK) return rc; }
        const int64_t take = std::min(max_rows - got, b->rows - b->consumed);
        const size_t at = (size_t)b->consumed;
        if (tile_out) std::memcpy(tile_out + (size_t)got * w, b->data + at * w, sizeof(float) * w * (size_t)take);
        if (ln_out) std::memcpy(ln_out + (size_t)got * w, b->ln + at * w, sizeof(float) * w * (size_t)take);
        if (minmax_out) std::memcpy(minmax_out + (size_t)got * 2, b->minmax + at * 2, sizeof(float) * 2 * (size_t)take);
        if (records_out) std::memcpy(records_out + got, b->records + at, sizeof(ro_scan_record_t) * (size_t)take);
        b->consumed += take;
        got += take;
        if (b->consumed == b->rows) {
            h->ready.pop_front();
            release_batch(h, b);
        }
    }
    h->rows_ready -= got;
    *rows_got = got;
    const double dt = now_ms() - t0;
    h->timing.fetch_calls += 1;
    h->fetch_ms_sum += dt;
    h->timing.fetch_ms_max = std::max(h->timing.fetch_ms_max, dt);
    return RO_OK;
}

extern "C" int ro_stft_set_row_sink(ro_stft_t *h, float *base, int64_t row_stride, int64_t capacity_rows, int64_t first_slot)
{
    if (!h) return fail(RO_ERR_INVALID, "null handle");
    if (!h->ready.empty() || h->staged_have > 0)
        return fail(RO_ERR_STATE, "the row sink can only change on an idle stream (after create or ro_stft_reset)");
    if (h->cfg.tile_ln) return fail(RO_ERR_UNSUPPORTED, "a tile_ln handle hands its rows out through ro_stft_fetch_ln");
    HIP_TRY(hipSetDevice(h->device));
    // (batches made with a sink hold no row buffer, batches made without one do: the pool starts over either way)
    while (!h->batch_pool.empty()) { destroy_batch(h->batch_pool.back()); h->batch_pool.pop_back(); }
    h->sink = nullptr;
    if (!base) return RO_OK;
    const int cols = h->cfg.tile_cols > 0 ? h->cfg.tile_cols : h->bins;
    if (row_stride < cols || capacity_rows < 2 * (int64_t)h->batch_rows || first_slot < 0 || first_slot >= capacity_rows)
        return fail(RO_ERR_INVALID, "row sink: stride %lld (rows are %d wide), %lld slots (two batches of %d rows at least), "
                                    "first slot %lld", (long long)row_stride, cols, (long long)capacity_rows, h->batch_rows,
                    (long long)first_slot);
    // The downloads into the ring are asynchronous DMA: the whole range has to be host memory page-locked by THIS
    // process's HIP runtime.  Heap memory is refused here rather than discovered by a copy engine later.
    {
        const size_t bytes = ((size_t)(capacity_rows - 1) * (size_t)row_stride + (size_t)cols) * sizeof(float);
        if (ro_pinned_check(base, bytes) != 1)
            return fail(RO_ERR_INVALID, "row sink: [%p, +%zu bytes) is not page-locked host memory of this process's HIP runtime "
                                        "(use ro_pinned_alloc)", (const void *)base, bytes);
    }
    h->sink = base;
    h->sink_stride = row_stride;
    h->sink_cap = capacity_rows;
    h->sink_first = first_slot;
    return RO_OK;
}

// 1: [p, p + bytes) is host memory page-locked by this process's HIP runtime (ro_pinned_alloc, hipHostMalloc,
// hipHostRegister) -- first and last byte are both known to the runtime as host allocations and lie in ONE mapping
// (equal distance in the runtime's view); 0: it is not (heap, stack, a numpy array, device memory, no device at all).
extern "C" int ro_pinned_check(const void *p, size_t bytes)
{
    if (!p || bytes == 0) return 0;
    const char *lo = static_cast<const char *>(p), *hi = lo + bytes - 1;
    hipPointerAttribute_t a0{}, a1{};
    const hipError_t e0 = hipPointerGetAttributes(&a0, lo);
    const hipError_t e1 = e0 == hipSuccess ? hipPointerGetAttributes(&a1, hi) : e0;
    if (e0 != hipSuccess || e1 != hipSuccess) {
        (void)hipGetLastError();
        return 0;
    }
    return a0.type == hipMemoryTypeHost && a1.type == hipMemoryTypeHost && a0.hostPointer && a1.hostPointer &&
                   static_cast<const char *>(a1.hostPointer) - static_cast<const char *>(a0.hostPointer) == hi - lo
               ? 1 : 0;
}

extern "C" void *ro_pinned_alloc(int device, size_t bytes)
{
    void *p = nullptr;
    if (bytes == 0 || hipSetDevice(device) != hipSuccess) return nullptr;
    if (hipHostMalloc(&p, bytes, hipHostMallocDefault) != hipSuccess) {
        (void)hipGetLastError();
        return nullptr;
    }
    return p;
}

extern "C" void ro_pinned_free(void *p)
{
    if (p) (void)hipHostFree(p);
}

extern "C" int ro_stft_reset(ro_stft_t *h)
{
    if (!h) return fail(RO_ERR_INVALID, "null handle");
    if (h->s_in) (void)hipStreamSynchronize(h->s_in);
    if (h->stream) (void)hipStreamSynchronize(h->stream);
    if (h->s_out) (void)hipStreamSynchronize(h->s_out);
    for (auto &sl : h->slot)
        if (sl.gstream) (void)hipStreamSynchronize(sl.gstream);
    h->staged_have = 0;
    h->stage_fmt_set = false;
    while (!h->ready.empty()) {
        release_batch(h, h->ready.front());
        h->ready.pop_front();
    }
    h->stream_sample0 = 0;
    h->rows_emitted = 0;
    h->rows_ready = 0;
    return RO_OK;
}

extern "C" int ro_stft_timing(ro_stft_t *h, ro_stft_timing_t *out, int reset)
{
    if (!h) return fail(RO_ERR_INVALID, "null handle");
    if (out) {
        *out = h->timing;
        out->push_ms_avg = h->timing.push_calls ? h->push_ms_sum / (double)h->timing.push_calls : 0.0;
        out->batch_gpu_ms_avg = h->timed_batches ? h->batch_ms_sum / (double)h->timed_batches : 0.0;
        out->row_gpu_us_avg = h->timed_rows ? h->batch_ms_sum * 1e3 / (double)h->timed_rows : 0.0;
        out->fetch_ms_avg = h->timing.fetch_calls ? h->fetch_ms_sum / (double)h->timing.fetch_calls : 0.0;
    }
    if (reset) {                                             // FFTBackend::clearProcessingTime, src/FFTBackend.h:231-235
        h->timing = ro_stft_timing_t{};
        h->push_ms_sum = h->batch_ms_sum = h->fetch_ms_sum = 0.0;
        h->timed_batches = h->timed_rows = 0;
    }
    return RO_OK;
}

extern "C" int ro_stft_stats(const ro_stft_t *h, int64_t *samples_in, int64_t *rows_out, int64_t *launches,
                             double *kernel_ms_total)
{
    if (!h) return fail(RO_ERR_INVALID, "null handle");
    if (samples_in) *samples_in = h->stat_samples;
    if (rows_out) *rows_out = h->stat_rows;
    if (launches) *launches = h->stat_launches;
    if (kernel_ms_total) *kernel_ms_total = h->stat_kernel_ms;
    return RO_OK;
}
