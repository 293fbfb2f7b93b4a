// ro_host.h -- what the translation units behind the C ABI share (internal; include/ro_stft.h is the ABI):
// the handle, the batches of the streaming path, the error text and the entry points one file offers the others.
//   ro_abi_helpers.cpp   error text, FFTBackend's public arithmetic, window tables, shard arithmetic, pinned memory
//   ro_exchange.cpp      the RCCL stitch of sharded rows (all-gather, gather to one rank, direct exchange)
//   ro_stream.cpp        push / flush / fetch: staging slots, captured graphs, the row sink
//   ro_czt.cpp           lengths that are not a power of two (chirp-z on an inner handle)
//   ro_stft_capi.cpp     create / destroy, the transform launches, the resident entry points
// There is no CPU compute path in any of them: rows only ever come out of the HIP kernels.
#pragma once

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

namespace ro {
namespace host {

// RO_ERR_* code in, the same code out; the text is what ro_last_error() returns on this thread
int fail(int code, const char *fmt, ...) __attribute__((format(printf, 2, 3)));
const char *last_error_text();

#define HIP_TRY(expr)                                                                     \
    do {                                                                                  \
        hipError_t e_ = (expr);                                                           \
        if (e_ != hipSuccess)                                                             \
            return fail(RO_ERR_HIP, "%s failed: %s", #expr, hipGetErrorString(e_));       \
    } while (0)

inline double now_ms()
{
    timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return ts.tv_sec * 1e3 + ts.tv_nsec * 1e-6;
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

}  // namespace host
}  // namespace ro

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
    std::vector<ro::host::Batch *> batch_pool;           // recycled pinned batches
    int64_t rows_emitted = 0;                  // stream index of the next row to compute
    std::deque<ro::host::Batch *> ready;
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
    // ... bins 256 ... 65536: the row in a CU's registers, no scratch (ro_f64reg.hip); its window order and twiddle tables
    bool     f64reg = false;
    float   *d_f64r_window = nullptr;
    double2 *d_f64r_tw[4] = {nullptr, nullptr, nullptr, nullptr};
    double2 *d_scratch_d[2] = {nullptr, nullptr};
    int64_t  scratch_rows_d = 0;
    // (diagnostic builds, RO_F64_FUSED=1: round 5's one-launch form of the through-HBM passes, ro_f64fused.hip:
    // 8 rings of f64_ring_rows rows, the launch's control block, and its give-up word mirrored into pinned host memory)
    double2  *d_f64_ring = nullptr;
    unsigned *d_f64_ctl = nullptr;
    unsigned *h_f64_err = nullptr;
    int       f64_ring_rows = 0, f64_wgs_per_cu = 0;
};

namespace ro {
namespace host {

// ---- ro_abi_helpers.cpp
void build_window(int kind, int bins, float *w);

// ---- ro_stft_capi.cpp
ro::StftArgs make_stft_args(const ro_stft *h, const void *d_iq, int64_t first_row, int64_t rows, float *d_rows,
                            int64_t row_stride, float *d_tile = nullptr, ro_scan_record_t *d_records = nullptr,
                            float *d_ln = nullptr);
// window -> FFT -> |X| for rows [first_row, +rows): the single-pass kernel, or for bins > 32768
// the one-kernel or the scratch form (ro_stft_create), in chunks that fit the scratch blocks
// d_tile / d_records (either may be null): produced here too, by the transform's own epilogue where the plan fuses them
// (N = 32768), by tile_kernel / scan_kernel behind it otherwise
int launch_transform(ro_stft *h, const void *d_iq, int format, int64_t first_row, int64_t rows, float *d_rows,
                     int64_t row_stride, hipStream_t s, float *d_tile = nullptr, ro_scan_record_t *d_records = nullptr,
                     float *d_ln = nullptr);
// d_ln / d_minmax: the tile's log and the rows' min / max of it (tile_ln); need d_tile
int launch_tile_and_scan(ro_stft *h, const float *d_rows, int64_t row_stride, int64_t rows, float *d_tile,
                         ro_scan_record_t *d_records, hipStream_t s, float *d_ln = nullptr, float *d_minmax = nullptr);
int ensure_ln_part(ro_stft *h, int64_t rows);
int launch_spectra_big(ro_stft *h, const void *d_iq, int format, int64_t first_row, int64_t rows, float2 *d_out,
                       int64_t out_stride, hipStream_t s);

// ---- ro_czt.cpp
int czt_length(int bins);                       // the inner power-of-two length of an even bins that is not one, else 0
int czt_setup(ro_stft *h);                      // the inner handle and the chirp tables of a handle with h->czt
int launch_transform_czt(ro_stft *h, const void *d_iq, int format, int64_t first_row, int64_t rows, float *d_rows,
                         int64_t row_stride, hipStream_t s);

// ---- ro_stream.cpp
void destroy_batch(Batch *b);
void free_stream_slots(ro_stft *h);

}  // namespace host
}  // namespace ro
