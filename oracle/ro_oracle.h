/*
 * ro_oracle.h -- CPU restatement of radio-observer's STFT / waterfall / bolid-scan
 * hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under radio-observer_amd/ (the product)
 * may include, link, load or call this.  Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg use it, and only as the checker / baseline.
 *
 * PARITY UNPINNED: the reference holds no golden vector, known-answer test or
 * fixture for this path (its tests cover RingBuffer bookkeeping only,
 * tests/RingBufferTest.h), and the reference cannot be built here: every
 * translation unit needs the un-vendored `cppapp` submodule and the FFT needs
 * libfftw3, neither of which is in this image.  The ring-buffer invariants
 * that the reference's tests DO pin are restated in tests/test_ring.py and
 * checked against ro_oracle_ring2d_*; everything else is pinned only by
 * independent cross-checks (numpy pocketfft, analytic known answers).
 *
 * All file:line citations are into the reference tree (/root/reference).
 */
#ifndef RO_ORACLE_H
#define RO_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- bin / rate helpers: src/FFTBackend.h:134-200, src/FFTBackend.cpp:150-151 */
float ro_oracle_fft_sample_rate(int sample_rate, int bins, int overlap);
int   ro_oracle_clamp_overlap(int bins, int overlap);            /* src/FFTBackend.cpp:108-109 */
int   ro_oracle_frequency_to_bin(int bins, int sample_rate, float frequency);
float ro_oracle_bin_to_frequency(int bins, int sample_rate, int bin);
int   ro_oracle_time_to_fft_samples(double seconds, float fft_sample_rate);
/* Recorder::fftSamplesToRaw, src/WaterfallBackend.h:84-87 (note: the recorder's
 * getFFTSampleRate() returns int, src/WaterfallBackend.cpp:29-32) */
int   ro_oracle_recorder_fft_samples_to_raw(int rows, float fft_sample_rate, int sample_rate);
/* WaterfallBackend::fftSamplesToRaw, src/WaterfallBackend.h:283-287 (float rate) */
int   ro_oracle_backend_fft_samples_to_raw(int rows, float fft_sample_rate, int sample_rate);

/* ---- window tables: src/FFTBackend.cpp:156-186 */
void ro_oracle_window_nuttall(int bins, float *w);
void ro_oracle_window_hann(int bins, float *w);   /* the commented-out variant, :157-163 */

/* ---- framing: src/FFTBackend.cpp:211-257 */
int64_t ro_oracle_row_count(int64_t samples, int bins, int overlap);

/* ---- forward unnormalised FP64 DFT (stand-in for fftw_execute, src/FFTBackend.cpp:120,236).
 * in/out interleaved (re,im); bins must be a power of two. Returns 0 on success. */
int ro_oracle_fft_f64(int bins, const double *in, double *out);
/* Optional engine behind ro_oracle_fft_f64: libfftw3 itself (what src/FFTBackend.cpp:117-120,236 calls), dlopen'ed when
 * the host has libfftw3.so.3; failing that, a vendor FFT that implements the same fftw3 API (MKL's FFTW3 interface).
 * ro_oracle_use_fftw(1) returns 1 if such an engine is now active, 0 if none was found.
 * ro_oracle_fftw_active(): 0 = the built-in radix-2 transform, 1 = libfftw3, 2 = MKL through its FFTW3 interface. */
int ro_oracle_use_fftw(int on);
int ro_oracle_fftw_active(void);
/* build the plan / tables of one size up front (needed before several threads transform that size) */
int ro_oracle_fft_prepare(int bins);
/* O(N^2) direct DFT in long double, for checking the FFT on small sizes. */
void ro_oracle_dft_direct(int bins, const double *in, double *out);

/* ---- one row: window multiply (:229-232) + DFT + abs/shift (src/WaterfallBackend.cpp:485-505).
 * iq: bins interleaved doubles. row: bins floats. spectrum (optional): bins interleaved doubles. */
int ro_oracle_row(int bins, const double *iq, const float *w, double gain,
                  float *row, double *spectrum);

/* ---- batch STFT over a contiguous stream of `samples` complex doubles.
 * rows: row_count x bins floats. Returns number of rows written or <0. */
int64_t ro_oracle_stft(const double *iq, int64_t samples, int bins, int overlap,
                       const float *w, double gain,
                       int64_t first_row, int64_t max_rows, float *rows);
/* same with float32 interleaved input (the RawStream wire format, src/RawStream.cpp:33,61-62) */
int64_t ro_oracle_stft_f32(const float *iq, int64_t samples, int bins, int overlap,
                           const float *w, double gain,
                           int64_t first_row, int64_t max_rows, float *rows);

/* ---- WFTime arithmetic: src/WFTime.h:92-114 */
void ro_oracle_wftime_add(int64_t sec, int64_t usec, int64_t add_sec, int64_t add_usec,
                          int64_t *out_sec, int64_t *out_usec);
void ro_oracle_wftime_add_samples(int64_t sec, int64_t usec, uint64_t count, int rate,
                                  int64_t *out_sec, int64_t *out_usec);

/* ---- streaming emulation of Frontend::process + FFTBackend::process
 * (src/Frontend.cpp:41-52, src/FFTBackend.cpp:192-279). */
typedef struct ro_oracle_stream ro_oracle_stream_t;

typedef struct {
    uint64_t offset;      /* info_.offset  (row index)              :256 */
    int64_t  time_sec;    /* info_.timeOffset                       :225 */
    int64_t  time_usec;
    int      raw_mark;    /* windowRaw_[0].mark after the memmove   :242,251 */
} ro_oracle_row_info_t;

ro_oracle_stream_t *ro_oracle_stream_create(int bins, int overlap, int sample_rate,
                                            int64_t start_sec, int64_t start_usec,
                                            double gain, int raw_capacity_rows);
void ro_oracle_stream_destroy(ro_oracle_stream_t *s);
/* feed one Frontend::process() call worth of samples; rows produced by this call are appended
 * to rows_out (bins floats each) / info_out, up to max_rows. Returns rows produced or <0. */
int ro_oracle_stream_process(ro_oracle_stream_t *s, const double *iq, int n,
                             float *rows_out, ro_oracle_row_info_t *info_out, int max_rows);
int ro_oracle_stream_pending(const ro_oracle_stream_t *s);  /* samples held in window_ */

/* ---- BolidRecorder static scans: src/BolidRecorder.cpp:302-347 */
float ro_oracle_noise(float *buffer, int length);     /* sorts buffer in place, like the reference */
int   ro_oracle_peak(const float *buffer, int length);
float ro_oracle_average(const float *buffer, int length);

typedef struct {
    float noise;       /* n   :124 */
    int   peak;        /* p   :125 */
    float average;     /* a   :126-132 */
} ro_oracle_scan_t;

/* per-row scan as BolidRecorder::update does it (:121-132). `row` must be addressable from
 * low_detect+peak-avg_bins/2 (may lie below low_detect). */
void ro_oracle_scan_row(const float *row, int low_noise, int noise_width,
                        int low_detect, int detect_width, int avg_bins,
                        ro_oracle_scan_t *out);

/* ---- BolidRecorder::start band setup: src/BolidRecorder.cpp:80-116 */
typedef struct {
    int low_detect, detect_width;
    int low_noise, noise_width;
    int advance, jitter, avg_bins, noise_metadata_rows;
} ro_oracle_bands_t;

void ro_oracle_bolid_bands(int bins, int sample_rate, float fft_sample_rate,
                           float min_detect_fq, float max_detect_fq,
                           float min_noise_fq, float max_noise_fq,
                           double advance_time, double jitter_time,
                           float avg_freq_range, double noise_metadata_time,
                           ro_oracle_bands_t *out);

/* ---- BolidRecorder::update FSM: src/BolidRecorder.cpp:171-273 */
enum { RO_ORACLE_STATE_INIT = 0, RO_ORACLE_STATE_BOLID = 1, RO_ORACLE_STATE_BOLID_ENDED = 2 };

typedef struct {
    int   state;
    float peak_freq, noise, magnitude;
    int   duration;
    int   snap_start, snap_length;   /* nextSnapshot_.start / .length */
    int   advance, jitter;
    float fft_sample_rate;
    int   sample_rate;
    float min_detect_fq, max_detect_fq;
} ro_oracle_fsm_t;

typedef struct {
    int   fired;            /* 1 when the METEOR DETECTED branch ran (:203-266) */
    int   snap_start;       /* nextSnapshot_.start at the time of the event */
    int   snap_length;      /* nextSnapshot_.length (before startWriting clamps it) */
    float duration_s;       /* :209 */
    float noise, peak_freq, magnitude;
    float fmin, fmax;       /* :241-242 */
    int   raw_length;       /* fftSamplesToRaw(length) :246 */
} ro_oracle_event_t;

void ro_oracle_fsm_init(ro_oracle_fsm_t *f, int advance, int jitter, float fft_sample_rate,
                        int sample_rate, float min_detect_fq, float max_detect_fq);
/* mark = buffer_->mark() after the row was pushed (row index + 1 mod capacity). */
void ro_oracle_fsm_update(ro_oracle_fsm_t *f, float n, float a, float peak_fq, int mark,
                          ro_oracle_event_t *ev);

/* ---- RingBuffer2D bookkeeping: src/RingBuffer.h:210-621 (indices only, no payload) */
typedef struct ro_oracle_ring2d ro_oracle_ring2d_t;
ro_oracle_ring2d_t *ro_oracle_ring2d_create(int elem_size, int width, int chunk_bytes, int capacity);
void ro_oracle_ring2d_destroy(ro_oracle_ring2d_t *r);
int  ro_oracle_ring2d_capacity(const ro_oracle_ring2d_t *r);
int  ro_oracle_ring2d_chunk_rows(const ro_oracle_ring2d_t *r);
int  ro_oracle_ring2d_get_size(const ro_oracle_ring2d_t *r);
int  ro_oracle_ring2d_is_full(const ro_oracle_ring2d_t *r);
int  ro_oracle_ring2d_push(ro_oracle_ring2d_t *r);              /* returns row index written */
int  ro_oracle_ring2d_mark(const ro_oracle_ring2d_t *r);
int  ro_oracle_ring2d_normalize(const ro_oracle_ring2d_t *r, int mark);
int  ro_oracle_ring2d_size_from(const ro_oracle_ring2d_t *r, int start);
int  ro_oracle_ring2d_size_between(const ro_oracle_ring2d_t *r, int start, int end);
int  ro_oracle_ring2d_reserve(ro_oracle_ring2d_t *r, int start, int end);
int  ro_oracle_ring2d_free_reservation(ro_oracle_ring2d_t *r, int handle);
int  ro_oracle_ring2d_is_dirty(const ro_oracle_ring2d_t *r, int handle);

/* ---- offline ln / min-max tile: fits2png:46, :444-445, :476-502 (restated, not imported) */
void ro_oracle_ln_rows(const float *rows, int64_t count, float *out);
/* fits2png:444-445,476-477,495-497: min/max of ln over non-zero pixels, 8-bit grey levels */
void ro_oracle_ln_levels(const float *image, int64_t count, float *ln_out, uint8_t *u8_out, float *minmax);

#ifdef __cplusplus
}
#endif
#endif
