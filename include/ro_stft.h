/*
 * ro_stft.h -- C ABI of the MI355X-native STFT / waterfall / bolid-scan path.
 *
 * This is the drop-in boundary for radio-observer's hot path.  Each entry point
 * names the reference interface (file:line under the reference tree) it
 * replaces.  Plain C types only: pointers, sizes, PODs.  No exceptions cross
 * this boundary; every call returns RO_OK (0) or a negative RO_ERR_* code and
 * ro_last_error() gives the text (the reference's own convention is
 * log-and-return, src/FFTBackend.cpp:194, src/WAVStream.cpp:209-213).
 *
 * One handle = one stream = one HIP stream; a handle is not thread-safe, which
 * mirrors the reference's single-caller rule for Backend::process()
 * (src/WAVStream.cpp:115-123, src/JackFrontend.cpp:15-39).
 *
 * There is NO CPU fallback behind this ABI: without a gfx950 device every
 * compute entry point fails with RO_ERR_HIP.
 */
#ifndef RO_STFT_H
#define RO_STFT_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RO_OK               0
#define RO_ERR_INVALID     (-1)   /* bad argument / shape mismatch                */
#define RO_ERR_UNSUPPORTED (-2)   /* e.g. bins not a supported power of two       */
#define RO_ERR_HIP         (-3)   /* HIP runtime error or no device               */
#define RO_ERR_NOMEM       (-4)
#define RO_ERR_STATE       (-5)   /* call not valid in the handle's current state */

#define RO_ABI_VERSION 5

/* window function; the reference hard-codes the 4-term Nuttall
 * (src/FFTBackend.cpp:165-184) and keeps Hann as dead code (:157-163). */
enum { RO_WINDOW_NUTTALL = 0, RO_WINDOW_HANN = 1, RO_WINDOW_CUSTOM = 2 };

/* sample formats accepted by push / resident runs.
 *  F32: interleaved float32 I,Q -- RawStream's wire format (src/RawStream.cpp:33,61-62)
 *  I16: interleaved int16 I,Q, un-normalised -- WAVStream (src/WAVStream.cpp:119-120)
 *  F64: {double real; double imag;} -- struct Complex (src/Backend.h:26-29).
 *       RO_PRECISION_F64 handles of 256 ... 65536 bins stage, upload and multiply
 *       the doubles themselves ((double)sample x (double)w, src/FFTBackend.cpp:
 *       229-232) and also take them device-resident; every other handle accepts
 *       them through ro_stft_push only and narrows them to float32 while staging
 *       (lossless for every frontend the reference has: int16 WAV, float32 raw /
 *       JACK). */
enum { RO_IQ_F32 = 0, RO_IQ_I16 = 1, RO_IQ_F64 = 2 };

/* arithmetic of the transform.
 *  F32: float32 butterflies on float32 samples -- the fast path; rows within 1e-5 of the ROW MAXIMUM of the
 *       reference's double-precision rows (measured 1-3e-7), which is the norm-wise reading of "1e-5 relative".
 *  F64: the reference's own arithmetic type -- double window multiply, double transform, double sqrt, one
 *       narrowing to the float row (src/FFTBackend.cpp:117-120,229-236, src/WaterfallBackend.cpp:492-505).  Bins
 *       256 ... 65536 keep the complex-double row in a compute unit's registers (samples read once, row written
 *       once; below 4096 bins 2 ... 16 rows share a workgroup); 131072 ... 1048576 are passes through HBM scratch.  Rows within 1e-5 of the reference PER BIN
 *       (measured <= 1.2e-7: one float32 ulp, at 60 dB of dynamic range); 2.5 times slower than F32 at 32768 bins.
 *       Complex spectra (ro_stft_spectra_resident) up to 65536 bins.
 *  (Value 2 was ABI 4's RO_PRECISION_F64_ONE_LAUNCH, an experiment that measured slower; ro_stft_create answers
 *  RO_ERR_UNSUPPORTED for it.) */
enum { RO_PRECISION_F32 = 0, RO_PRECISION_F64 = 1 };

/* bin ranges of BolidRecorder::start (src/BolidRecorder.cpp:84-102), in
 * fft-shifted row columns. */
typedef struct ro_bands {
    int32_t low_noise;     /* lowNoiseBin_      */
    int32_t noise_width;   /* noiseWidth_       */
    int32_t low_detect;    /* lowDetectBin_     */
    int32_t detect_width;  /* detectWidth_      */
    int32_t avg_bins;      /* averageBinRange_  */
} ro_bands_t;

/* what BolidRecorder::update computes per row before its state machine
 * (src/BolidRecorder.cpp:121-132): n = noise(), p = peak(), a = average(). */
typedef struct ro_scan_record {
    float   noise;
    int32_t peak;
    float   average;
} ro_scan_record_t;

/* Construction parameters = WaterfallBackend::make's config keys
 * (src/WaterfallBackend.cpp:620-646) plus device-side options. */
typedef struct ro_stft_config {
    uint32_t     struct_size;      /* sizeof(ro_stft_config_t), for ABI growth     */
    int32_t      bins;             /* "bins"     default 32768                      */
    int32_t      overlap;          /* "overlap"  default 0; clamped to [0,bins-1]   */
    int32_t      sample_rate;      /* StreamInfo::sampleRate, default 48000         */
    int32_t      window_kind;      /* RO_WINDOW_*                                   */
    const float *window_table;     /* RO_WINDOW_CUSTOM: bins floats (host)          */
    double       iq_gain;          /* "iq_gain": added to Q (src/FFTBackend.cpp:79) */
    int32_t      iq_phase_shift;   /* must be 0 (non-zero is UB in the reference)   */
    int32_t      device;           /* HIP device ordinal                            */
    int32_t      max_batch_rows;   /* streaming: rows per launch (0 = default)      */
    int32_t      enable_scan;      /* compute ro_scan_record_t per row              */
    ro_bands_t   bands;            /* used when enable_scan                         */
    int32_t      tile_first_col;   /* compact band tile [tile_first_col, +tile_cols)*/
    int32_t      tile_cols;        /* 0 = no tile                                   */
    int32_t      spare_cus_per_xcd;/* CUs per XCD the persistent STFT grid leaves    */
                                   /* free (0..16).  The N >= 16384 kernels fill a   */
                                   /* CU's registers, so nothing else runs beside    */
                                   /* them; 1 lets a concurrent kernel on another    */
                                   /* stream (an RCCL collective) make progress.     */
    int32_t      precision;        /* RO_PRECISION_* (ABI 2; a caller that passes the */
                                   /* ABI-1 struct_size gets RO_PRECISION_F32)        */
    int32_t      tile_ln;          /* 1: with the tile, also its natural log and the  */
                                   /* per-row min / max of that log (the offline      */
                                   /* viewer's FN_LOG and colour range, fits2png:46,  */
                                   /* :476-477) straight from the transform; needs    */
                                   /* tile_cols > 0                                   */
} ro_stft_config_t;

typedef struct ro_stft ro_stft_t;

/* ---- library ------------------------------------------------------------- */
int         ro_abi_version(void);
const char *ro_last_error(void);          /* thread-local text of the last failure */
int         ro_device_count(void);        /* <0 on HIP error                        */

/* ---- pure host helpers: FFTBackend's public arithmetic -------------------- */
/* FFTBackend::FFTBackend overlap clamp, src/FFTBackend.cpp:108-109 */
int     ro_clamp_overlap(int bins, int overlap);
/* fftSampleRate_, src/FFTBackend.cpp:150-151 */
float   ro_fft_sample_rate(int sample_rate, int bins, int overlap);
/* FFTBackend::frequencyToBin, src/FFTBackend.h:159-178 */
int     ro_frequency_to_bin(int bins, int sample_rate, float frequency);
/* FFTBackend::binToFrequency, src/FFTBackend.h:134-147 */
float   ro_bin_to_frequency(int bins, int sample_rate, int bin);
/* FFTBackend::timeToFFTSamples, src/FFTBackend.h:197-200 */
int     ro_time_to_fft_samples(double seconds, float fft_sample_rate);
/* rows produced by FFTBackend::process for a stream of `samples`, :211-257 */
int64_t ro_row_count(int64_t samples, int bins, int overlap);
/* window table as FFTBackend::startStream builds it, :156-186 */
int     ro_window_table(int kind, int bins, float *out);
/* ---- time-chunk sharding of one stream over `world` GPUs ------------------------
 * Rows are independent (row r needs samples [r*hop, r*hop+bins) only,
 * src/FFTBackend.cpp:211-257), so shard g takes the contiguous rows
 * [floor(g*R/world), floor((g+1)*R/world)) and the samples under them; neighbouring
 * shards overlap by the bins-hop samples of halo.  Pure host arithmetic, no device. */
int     ro_shard_rows(int64_t total_rows, int world, int rank, int64_t *first_row, int64_t *rows);
/* first sample and sample count shard [first_row, +rows) must hold (halo included; 0 samples for an empty shard) */
int     ro_shard_samples(int64_t first_row, int64_t rows, int bins, int overlap,
                         int64_t *first_sample, int64_t *samples);
/* rows every rank contributes to an equal-block all-gather: the largest shard */
int64_t ro_shard_max_rows(int64_t total_rows, int world);
/* Stitch the result of an equal-block all-gather (world blocks of ro_shard_max_rows rows of
 * row_bytes each, short shards zero-padded at their end) into total_rows consecutive rows --
 * the row order the FITS writer (src/WaterfallBackend.cpp:141-211) and BolidRecorder's state
 * machine (src/BolidRecorder.cpp:171-273) consume.  Host memory; `out` may not alias `gathered`. */
int     ro_stitch_rows(const void *gathered, int64_t total_rows, int world, size_t row_bytes, void *out);
/* The one exchange step of a multi-GPU run, for a host that owns an RCCL communicator (one process per GPU): the
 * all-gather that puts every shard's rows (band tile, ln tile or 12-byte scan records) on every rank, in the equal-block
 * layout ro_stitch_rows / ro_stitch_rows_device undo.
 *   nccl_comm   the host's ncclComm_t; librccl is dlopen'ed on first use (RO_ERR_UNSUPPORTED if the box has none)
 *   d_local     device, local_rows x row_bytes: this rank's rows (local_rows as ro_shard_rows gives for `rank`)
 *   d_staging   device, ro_shard_max_rows x row_bytes of scratch (the zero-padded send block)
 *   d_gathered  device, world x ro_shard_max_rows x row_bytes
 * Asynchronous on `stream` (copy into the staging block, then ncclAllGather as bytes). */
int     ro_allgather_rows(void *nccl_comm, const void *d_local, int64_t local_rows, int64_t total_rows, int world,
                          int rank, size_t row_bytes, void *d_staging, void *d_gathered, void *stream);
/* The all-gather as the DIRECT exchange SURVEY 5 / 8(e) asks for ("prefer a direct all-to-all-write pattern ... over a
 * single ring": xGMI is point to point, seven links per GPU): inside one ncclGroupStart / ncclGroupEnd every rank
 * ncclSend's its local_rows x row_bytes to each peer and ncclRecv's each peer's rows AT THEIR STITCHED PLACE in d_out
 * (total_rows x row_bytes, device, on every rank) -- no zero-padded staging block, no stitch afterwards; the band the
 * consumer writes (src/WaterfallBackend.cpp:174-205) is complete on every rank when the stream reaches this point.
 * Asynchronous on `stream`. */
int     ro_allgather_rows_direct(void *nccl_comm, const void *d_local, int64_t local_rows, int64_t total_rows, int world,
                                 int rank, size_t row_bytes, void *d_out, void *stream);
/* The schedule of that direct exchange, one step of it: in step k (1 .. world - 1) rank `rank` sends its own rows to
 * *to = (rank + k) mod world and receives the rows of *from = (rank - k) mod world, which are rows
 * [*recv_first_row, +*recv_rows) of the stitched result -- every ordered pair of ranks meets in exactly one step, and
 * in every step every rank sends once and receives once (each xGMI link pair is used once per step).  k = 0 is the
 * rank's own copy (to = from = rank).  ro_allgather_rows_direct, ro_gather_rows (root: every step's receive; the
 * others: the one step whose `to` is the root) and timeshard.gather_rows_direct all walk this one function.
 * Pure host arithmetic. */
int     ro_direct_schedule(int world, int rank, int64_t total_rows, int k, int *to, int *from,
                           int64_t *recv_first_row, int64_t *recv_rows);
/* The same exchange when only ONE rank consumes the rows -- the reference's FITS writer and detector are one process
 * (src/WaterfallBackend.cpp:141-211, src/BolidRecorder.cpp:171-273): every rank sends its local_rows x row_bytes
 * straight to `root` (ncclSend / ncclRecv in one group: world - 1 transfers over world - 1 different links), where
 * they land at their own place in d_out (total_rows x row_bytes, device; ignored on the other ranks).  No padding,
 * no stitch.  Asynchronous on `stream`. */
int     ro_gather_rows(void *nccl_comm, const void *d_local, int64_t local_rows, int64_t total_rows, int world,
                       int rank, int root, size_t row_bytes, void *d_out, void *stream);
/* ro_stitch_rows for device memory: world device-to-device copies on `stream` */
int     ro_stitch_rows_device(const void *d_gathered, int64_t total_rows, int world, size_t row_bytes, void *d_out,
                              void *stream);
/* 1 if `bins` has a kernel in this build: powers of two 256 .. 1048576 (one kernel up to
 * 131072; above, two kernels and one trip through HBM scratch that the handle allocates on first
 * use: 8 bytes per bin and row for up to 1 GiB / (8 bins) rows at a time), and every other EVEN
 * length 258 .. 524286 (FFTW takes any N, src/FFTBackend.cpp:120; src/BolidRecorder.h:35
 * suggests 32728) as a chirp-z transform on the power-of-two length M >= 2 bins - 1 (scratch:
 * 20 M bytes per row for up to 1 GiB / (8 M) rows at a time).  Odd lengths have no defined
 * result in the reference (src/WaterfallBackend.cpp:489-505 leaves the last column unwritten). */
int     ro_bins_supported(int bins);

/* ---- handle --------------------------------------------------------------- */
/* FFTBackend::FFTBackend + startStream (src/FFTBackend.cpp:103-127, :144-189)
 * + WaterfallBackend::startStream's buffer sizing (src/WaterfallBackend.cpp:573-594). */
int ro_stft_create(const ro_stft_config_t *cfg, ro_stft_t **out);
/* FFTBackend::~FFTBackend, src/FFTBackend.cpp:130-141 */
int ro_stft_destroy(ro_stft_t *h);
int ro_stft_get_window(const ro_stft_t *h, float *out /* bins floats, host */);
int ro_stft_hop(const ro_stft_t *h);
int ro_stft_bins(const ro_stft_t *h);
int ro_stft_device_name(const ro_stft_t *h, char *buf, size_t len);
int ro_stft_set_bands(ro_stft_t *h, const ro_bands_t *bands);

/* ---- resident path (benchmarks, multi-GPU shards) --------------------------
 * Replaces, for rows [first_row, first_row+rows) of a stream that is already in
 * HBM, the whole of FFTBackend::process's row loop (src/FFTBackend.cpp:211-257)
 * and WaterfallBackend::processFFT's magnitude/shift (src/WaterfallBackend.cpp:485-505).
 *   d_iq        device pointer to sample 0 of the stream, `format` F32 or I16 (F64 too on RO_PRECISION_F64 handles of
 *               256 ... 65536 bins; RO_ERR_UNSUPPORTED elsewhere)
 *   samples     number of complex samples addressable at d_iq
 *   d_rows      device, rows x row_stride floats (row_stride >= bins); required
 *   d_tile      device, rows x tile_cols floats (compact copy of columns
 *               [tile_first_col, +tile_cols) of d_rows), or NULL
 *   d_records   device, rows records, or NULL (needs enable_scan)
 *   stream      hipStream_t as void* (NULL = the default stream, with its usual ordering rules)
 * The call is asynchronous on `stream`.  One handle = one stream: the handle owns scratch that some paths use
 * (bins > 131072, chirp-z lengths, RO_PRECISION_F64 above 65536 bins, tile_ln), so two launches of ONE handle may only be in flight
 * together when they are ordered on one stream -- like FFTBackend::process, which one thread calls at a time
 * (src/JackFrontend.cpp:19-22 only logs a re-entry).  Use one handle per concurrent stream. */
int ro_stft_run_resident(ro_stft_t *h, const void *d_iq, int format, int64_t samples,
                         int64_t first_row, int64_t rows,
                         float *d_rows, int64_t row_stride,
                         float *d_tile, ro_scan_record_t *d_records, void *stream);

/* ro_stft_run_resident plus the viewer's log image of the tile, produced by the transform itself (tile_ln = 1):
 *   d_ln_tile    device, rows x tile_cols floats: logf(pixel) of every tile pixel (-inf for a zero pixel)
 *   d_ln_minmax  device, rows x 2 floats: min and max of that row's log over its NON-ZERO pixels (+inf, -inf for a
 *                row without any) -- the image's colour range (fits2png:476-477) is the min / max over its rows,
 *                and ro_ln_levels turns log values into the viewer's grey levels once that range is known
 * d_ln_minmax may be NULL.  For bins = 32768 the log is taken in the transform's epilogue, on the magnitudes still in LDS;
 * for the other sizes by a small kernel over the tile. */
int ro_stft_run_resident_ln(ro_stft_t *h, const void *d_iq, int format, int64_t samples,
                            int64_t first_row, int64_t rows, float *d_rows, int64_t row_stride,
                            float *d_tile, float *d_ln_tile, float *d_ln_minmax,
                            ro_scan_record_t *d_records, void *stream);
/* The viewer's grey levels of log values given the image's range: level = (uint8)((ln - mn) / (mx - mn) * 255) in
 * float32 arithmetic, truncating; -inf (a zero pixel) and a flat image give 0 (fits2png:444-445, :495-497).  Host. */
int ro_ln_levels(const float *ln, int64_t count, float mn, float mx, uint8_t *levels_out);

/* The complex spectra themselves instead of their magnitudes: what fftw_execute leaves in out_
 * (src/FFTBackend.cpp:236) and hands to the protected hook FFTBackend::processFFT(const fftw_complex *data,
 * int size, DataInfo, int rawMark) (src/FFTBackend.h:104) -- bin k of row r at d_spectra[r * stride + k] as
 * {float re, float im}, unshifted (k = 0 is DC), unnormalised, after gain and window like the magnitude path.
 * Same kernels with a different last step; rows x stride x 8 bytes are written.  Every size the handle supports:
 * one kernel up to 32768 bins, fold + transform + interleave through the handle's scratch above (power-of-two bins:
 * RO_ERR_UNSUPPORTED for chirp-z lengths).  An RO_PRECISION_F64 handle of 256 ... 65536 bins hands out its double
 * transform, each component narrowed to float once (every bin within a float32 ulp of the reference's fftw_complex,
 * however far below the row's largest); RO_ERR_UNSUPPORTED above.  Asynchronous on `stream`; one launch in flight per
 * handle above 32768 bins in float32 (the scratch is the handle's). */
int ro_stft_spectra_resident(ro_stft_t *h, const void *d_iq, int format, int64_t samples,
                             int64_t first_row, int64_t rows,
                             float *d_spectra /* rows x stride x {re, im} */, int64_t stride, void *stream);

/* BolidRecorder::noise/peak/average over rows already in HBM
 * (src/BolidRecorder.cpp:121-132, :313-347). */
int ro_stft_scan_resident(ro_stft_t *h, const float *d_rows, int64_t row_stride, int64_t rows,
                          ro_scan_record_t *d_records, void *stream);

/* The offline viewer's transform of a band image, on rows already in HBM (fits2png:46 FN_LOG,
 * :444-445 default_color_fn, :476-477 min/max, :495-497 the uint8 store):
 *   ln    = logf(pixel), float32, over columns [first_col, first_col+cols) of `rows` rows
 *   min/max of ln over the NON-ZERO pixels of the whole image (the viewer drops zeros)
 *   level = (uint8)((ln - min) / (max - min) * 255), float32 arithmetic, truncating
 * Zero pixels give -inf in d_ln and level 0; an image with max == min gives level 0 everywhere.
 *   d_ln      device, rows x cols floats, or NULL
 *   d_u8      device, rows x cols bytes,  or NULL
 *   d_minmax  device, 2 floats {min, max}, or NULL
 * Asynchronous on `stream`. */
int ro_stft_ln_tile_resident(ro_stft_t *h, const float *d_rows, int64_t row_stride, int64_t rows,
                             int first_col, int cols, float *d_ln, uint8_t *d_u8, float *d_minmax,
                             void *stream);

/* Times `iters` back-to-back launches of the resident path with HIP events on
 * the launch stream; ms_out[i] = duration of launch i (STFT kernel + scan kernel
 * when records are requested).  kernel_ms_out (optional, 2 floats) receives the
 * average STFT-kernel-only and scan-kernel-only durations. Synchronous. */
int ro_stft_time_resident(ro_stft_t *h, const void *d_iq, int format, int64_t samples,
                          int64_t first_row, int64_t rows,
                          float *d_rows, int64_t row_stride,
                          float *d_tile, ro_scan_record_t *d_records, void *stream,
                          int iters, float *ms_out, float *kernel_ms_out);

/* ---- streaming path (the Backend::process boundary) -------------------------
 * ro_stft_push   = FFTBackend::process (src/FFTBackend.cpp:192-279): takes one
 *                  Frontend::process() call worth of samples (any count), stages
 *                  them, and runs the kernels whenever max_batch_rows rows are
 *                  complete.  *rows_ready = rows waiting in the output queue.
 * ro_stft_flush  = run the kernels on every complete row staged so far
 *                  (WaterfallBackend::endStream, src/WaterfallBackend.cpp:600-607;
 *                  samples short of a hop are dropped, like the reference).
 * ro_stft_fetch  = hand rows to the Recorder side in stream order
 *                  (replaces the synchronous Recorder::update() per row,
 *                  src/WaterfallBackend.cpp:534-536): up to max_rows rows, columns
 *                  [first_col, first_col+cols) of each, plus scan records.  With a tile
 *                  configured (tile_cols > 0) only the tile's columns travel to the host
 *                  (what the FITS writer keeps, src/WaterfallBackend.cpp:176,204) and the
 *                  requested columns must lie inside it.
 *                  *first_row_index = stream index of the first row returned. */
int ro_stft_push(ro_stft_t *h, const void *iq, int format, int64_t samples, int64_t *rows_ready);
int ro_stft_flush(ro_stft_t *h, int64_t *rows_ready);
int ro_stft_fetch(ro_stft_t *h, int64_t max_rows, int first_col, int cols,
                  float *rows_out, ro_scan_record_t *records_out,
                  int64_t *first_row_index, int64_t *rows_got);
/* Rows at the head of the output queue whose batches have finished on the device, download included: ro_stft_fetch
 * hands these over without waiting (it WAITS for anything beyond them).  A host that fetches only what is complete
 * keeps the next batch in flight under the previous one's download and under its own per-row work -- the recorders
 * then see a batch's rows one Backend::process call later than a blocking fetch would show them, never later than the
 * next call or ro_stft_flush + fetch (src/WaterfallBackend.cpp:534-536 runs Recorder::update() inside processFFT;
 * here it runs a batch behind by construction). */
int ro_stft_rows_complete(ro_stft_t *h, int64_t *rows);
/* Row sink: the streaming path's rows go STRAIGHT into the caller's row ring instead of the handle's pinned batches --
 * WaterfallBackend::processFFT writes each finished row into buffer_->push() (src/WaterfallBackend.cpp:488-505); with a
 * sink the device-to-host copy of a batch is that write, and the ring's only copy.  Row r of the stream (counted from
 * ro_stft_create / ro_stft_reset) lands at  base + ((first_slot + r) mod capacity_rows) * row_stride  floats: the
 * handle's full rows, or its tile's columns when one is configured.  The ring has to be page-locked memory from
 * ro_pinned_alloc (the copies are asynchronous; hipPointerGetAttributes is asked about its first and last byte and
 * anything that is not a host allocation of this process's HIP runtime -- heap memory, a numpy array -- is refused with
 * RO_ERR_INVALID) and has to stay allocated until the sink is removed (base = NULL) or
 * the handle destroyed.  (ro_stft_flush launches batch by batch: one that would lap unfetched rows returns RO_ERR_STATE
 * with its samples still staged -- fetch, then flush again.)  With a sink ro_stft_push is ALL OR NOTHING: a call whose samples would complete more rows than
 * the ring has free slots (capacity_rows - rows waiting to be fetched) returns RO_ERR_STATE having consumed nothing;
 * fetch, then push the same buffer again.  A row is in place once ro_stft_fetch has reported it
 * (rows_out = NULL from then on: RO_ERR_STATE otherwise; the scan records still come through records_out), and the
 * handle writes the slots of every launched batch AHEAD of what has been fetched (as many batches as the caller lets
 * stay in flight, never lapping rows that wait to be fetched): capacity_rows >= 2 x max_batch_rows, and
 * whoever reads old rows of the ring (snapshot writers) has to stay that far behind the head, as it has to for push().
 * Only on an idle stream (RO_ERR_STATE otherwise); not for tile_ln handles (RO_ERR_UNSUPPORTED). */
int ro_stft_set_row_sink(ro_stft_t *h, float *base, int64_t row_stride, int64_t capacity_rows, int64_t first_slot);
/* page-locked host memory for such a ring (hipHostMalloc on `device`'s context); NULL when there is none to be had */
void *ro_pinned_alloc(int device, size_t bytes);
void  ro_pinned_free(void *p);
/* 1 when the first and the last byte of [p, p + bytes) are host memory page-locked by this process's HIP runtime and
 * lie the same distance apart in its view (what ro_stft_set_row_sink accepts:
 * the target of the DMA that stands in for processFFT's write into buffer_->push(), src/WaterfallBackend.cpp:488-505),
 * 0 for anything else -- heap or stack memory, device memory, or no HIP device in the process. */
int   ro_pinned_check(const void *p, size_t bytes);
/* ro_stft_fetch for a handle with tile_ln = 1: the whole tile of each row, its log, the row's min / max of the log
 * (2 floats) and the scan record -- any of the four may be NULL. */
int ro_stft_fetch_ln(ro_stft_t *h, int64_t max_rows, float *tile_out, float *ln_out, float *minmax_out,
                     ro_scan_record_t *records_out, int64_t *first_row_index, int64_t *rows_got);
/* FFTBackend::startStream's reset of inMark_/info_ (src/FFTBackend.cpp:148-153) */
int ro_stft_reset(ro_stft_t *h);
/* Per-call timing, the counterpart of FFTBackend's three RunningAverage2 counters and logProcessingTimes /
 * clearProcessingTime (src/FFTBackend.h:86-92, :208-235: average and maximum per process() call, per FFT, per
 * analysis).  push = wall time of one ro_stft_push (the host side of Backend::process); batch = GPU time of the kernels
 * of one batch (HIP events; window + FFT + magnitude, and the band scan where the plan fuses it), known once the batch
 * has been fetched; row = the same per row; fetch = wall time of one ro_stft_fetch (includes waiting for the GPU).
 * `batches` / `batch_rows` count every batch; of the latency-bound batches that run as one captured graph (full batches of
 * at most 4 MiB into a row sink) one in eight carries the timing events -- an event record costs the host as much as a
 * kernel launch -- and batch_gpu_ms_avg / _max and row_gpu_us_avg are over the timed ones. */
typedef struct ro_stft_timing {
    int64_t push_calls;  double push_ms_avg, push_ms_max;
    int64_t batches;     double batch_gpu_ms_avg, batch_gpu_ms_max;
    int64_t batch_rows;  double row_gpu_us_avg;
    int64_t fetch_calls; double fetch_ms_avg, fetch_ms_max;
} ro_stft_timing_t;
int ro_stft_timing(ro_stft_t *h, ro_stft_timing_t *out, int reset /* != 0: clear the counters afterwards */);
/* totals in the spirit of FFTBackend::logProcessingTimes (src/FFTBackend.h:208-229); kernel_ms_total counts an untimed
 * graphed batch (see ro_stft_timing) with the time of the last timed one -- the same graph */
int ro_stft_stats(const ro_stft_t *h, int64_t *samples_in, int64_t *rows_out,
                  int64_t *launches, double *kernel_ms_total);

#ifdef __cplusplus
}
#endif
#endif /* RO_STFT_H */
