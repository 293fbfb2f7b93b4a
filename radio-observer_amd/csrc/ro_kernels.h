// ro_kernels.h -- host-visible launch interface of the gfx950 kernels (internal,
// not part of the C ABI; see include/ro_stft.h for that).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <vector>

#include "../../include/ro_stft.h"

#define RO_FMT_F32 RO_IQ_F32
#define RO_FMT_I16 RO_IQ_I16

namespace ro {

struct StftArgs {
    const void   *iq;          // sample 0 of the stream (device)
    const float  *window;      // bins floats (device), natural order
    const float  *window_k;    // the same coefficients in kernel order (stft_window_layout)
    const float  *window_k32;  // ... in the order of the N = 32768 magnitude-row kernel (stft32k_window_layout)
    const float2 *twiddles;    // per-stage tables (device), see build_twiddles
    const float4 *twiddles_k;  // radix-16/32 stages repacked for 16-byte loads (stft_pack_twiddles)
    float        *rows_out;    // rows x row_stride (magnitude mode)
    float2       *spec_out;    // non-null selects the complex-spectrum mode: rows x spec_stride float2, bin k at [k]
    int64_t       spec_stride;
    int64_t       first_row;
    int64_t       rows;
    int64_t       row_stride;
    int           hop;
    float         gain;
    unsigned long long *stamps; // diagnostic builds only (RO_STAMPS), else nullptr
    int           stagger;     // start delay per workgroup slot, shader cycles (0 = none)
    int           prefetch;    // set by the launcher: touch the next row's new samples ahead of time
    int           spare_cus;   // CUs per XCD the persistent grid leaves to other kernels (ro_stft_config_t)
    int           dec, dec_log2; // MODE 3 (large transform, bins = dec x N): the factor, its log2; else 1, 0
    const float2 *dif_tw;        // MODE 3: exp(-2 pi i j / dec), j < dec
    const float2 *dif_shift;     // MODE 3: [dec][16]: exp(-2 pi i (q / dec) 2^i / M), i < 5, M = 32, 1024, 32768 at 0, 5, 10
    int           big_form;      // 1 selects MODE 3 (a large transform in one kernel), else 0
    // fused per-row band scan (BolidRecorder::noise/peak/average on the row while it is still in LDS): plans that
    // support it (stft_fuses_scan) fill records[row] themselves, the others leave it to launch_scan
    ro_scan_record_t *records; // rows records, or nullptr
    int           low_noise, noise_width, low_detect, detect_width, avg_bins;
    // fused band tile: columns [tile_first, +tile_cols) of every row, compact (rows x tile_cols), or nullptr
    float        *tile_out;
    int           tile_first, tile_cols;
    // ... and its log (ln_out: rows x tile_cols) with the two tile waves' partial min / max of it (ln_part: rows x 4
    // floats = {min, max} of the first and of the second half of the columns); launch_ln_finish folds them per row
    float        *ln_out;
    float        *ln_part;
};

struct TileArgs {
    const float *rows_in;
    float       *tile_out;     // rows x cols
    int64_t      rows;
    int64_t      row_stride;
    int          first, cols;
};

// ln(magnitude) band tile and its 8-bit grey image (the offline viewer's transform, fits2png:46,444-445,476-502)
struct LnArgs {
    const float *rows_in;
    float       *ln_out;       // rows x cols, or nullptr
    uint8_t     *u8_out;       // rows x cols, or nullptr
    unsigned    *keys;         // [0] = min, [1] = max of ln over the non-zero pixels, as order-preserving keys
    float       *minmax;       // the same two as floats, or nullptr
    int64_t      rows;
    int64_t      row_stride;
    int          first, cols;
};

struct ScanArgs {
    const float      *rows_in;
    ro_scan_record_t *records;
    int64_t           rows;
    int64_t           row_stride;
    int               bins;
    int               low_noise, noise_width;
    int               low_detect, detect_width;
    int               avg_bins;
};

// ---- strict precision (RO_PRECISION_F64): the Stockham recurrence as separate radix-16 passes in double, any power of two
struct BigArgsD {
    const void    *iq;         // FIRST pass: sample 0 of the stream
    const float   *window;     // FIRST pass (float32 coefficients, widened like src/FFTBackend.cpp:229-232)
    const double2 *tw;         // exp(-2 pi i m / N), m in [0, N), correctly rounded doubles
    const double2 *in;         // middle / last passes: [rows][N]
    double2       *out;        // first / middle passes: [rows][N]
    float         *rows_out;   // LAST pass: [rows][row_stride]
    int64_t        first_row;
    int64_t        rows;
    int64_t        row_stride;
    int            hop;
    int            n;
    int            ns;
    double         gain;
};
int        f64_radices(int bins, int radices[8]); // passes of the FP64 path for a power of two 256 .. 2^20 (0 = unsupported)
hipError_t launch_f64_pass(int radix, bool first, bool last, int fmt, const BigArgsD &a, hipStream_t s);
// two consecutive passes (radix 16 with a.ns, then radix r2) in one kernel through LDS; n >= 4096
hipError_t launch_f64_pair(int r2, bool first, bool last, int fmt, const BigArgsD &a, hipStream_t s);

// ---- strict precision with the row in a CU's registers (ro_f64reg.hip): bins = D x M, M in {4096, 8192, 16384}, or bins
// 256 ... 2048 as 16 ... 2 rows in the M = 4096 workgroup; no complex-double scratch.  Tables come from f64reg_tables (host), the caller uploads them.
struct F64RegTables {
    std::vector<float>   window_k;   // bins floats in the kernel's order
    std::vector<double2> tw0, tw1, tw2, tw3;
};
struct F64RegArgs {
    const void    *iq;          // sample 0 of the stream
    const float   *window_k;
    const double2 *tw0, *tw1, *tw2, *tw3;
    float         *rows_out;    // [rows][row_stride]
    int64_t        first_row, rows, row_stride;
    int            hop;
    double         gain;
    unsigned long long *stamps; // diagnostic builds only (RO_F64R_STAMPS), else nullptr
    int            spectra;     // 1: rows_out takes the transform itself, {float re, float im} per bin, unshifted; row_stride in pairs
};
bool       f64reg_supported(int bins);           // 256 ... 65536
void       f64reg_tables(int bins, const float *window, F64RegTables &t);
hipError_t launch_f64reg(int bins, int fmt, const F64RegArgs &a, hipStream_t s);

#ifdef RO_DIAG
// (diagnostic builds only: round 5's experiment, not in the product library)
// all four passes of a row in one persistent launch, the intermediate in one XCD's L2 (ro_f64fused.hip):
// bins = 16 x 16 x 16 x r2.  ring = 8 x ring_rows x n complex doubles, ctl = f64_fused_ctl_bytes() bytes (zeroed by the
// launch); a.in / a.out / a.ns are not used.  ctl word [1] != 0 after the launch: a bounded wait gave up.
bool       f64_fused_supported(int bins);
size_t     f64_fused_ctl_bytes();
int        f64_fused_max_ring_rows();
hipError_t launch_f64_fused(int fmt, const BigArgsD &a, double2 *ring, unsigned *ctl, int ring_rows, int wgs_per_cu, hipStream_t s);
#endif

// ---- large transforms as a four-step FFT (bins = n1 x 1024), ro_fourstep.hip
struct FourArgs {
    const void   *iq;          // sample 0 of the stream
    int64_t       first_row, rows;
    int           hop;
    float         gain;
    int           n1;          // bins / 1024
    const float  *window_a;    // bins floats in the column kernel's order (fourstep_tables)
    const float2 *tw_a;        // [32][n1 / 32]: exp(-2 pi i k_l m / n1)
    const float2 *tw_b;        // [n1][16]: powers 2^j, j < 5, of exp(-2 pi i 32 k1 / bins) at 0.., of exp(-2 pi i k1 / bins) at 8..
    const float2 *tw_r;        // [32][8]: powers 2^j, j < 5, of exp(-2 pi i k_r / 1024)
    float        *z;           // scratch: rows x n1 x 2048 floats
    float        *rows_out;    // rows x row_stride, fft-shifted magnitudes
    int64_t       row_stride;
    int           spare_cus;   // CUs per XCD both persistent grids leave to other kernels (ro_stft_config_t)
};
bool fourstep_supported(int bins);
hipError_t launch_fourstep(int format, const FourArgs &a, hipStream_t s);
// the tables of a size (host side), from the natural window
void fourstep_tables(int bins, const float *window, std::vector<float> &window_a, std::vector<float2> &tw_a,
                     std::vector<float2> &tw_b, std::vector<float2> &tw_r);

// ---- large transforms (bins = dec x 32768), see the note in front of fold_kernel
// First step of the scratch form of a large transform (bins = dec x m, decimation in frequency):
//   out[(row dec + q) m + i] = W_bins^(i q) sum_r W_dec^(r q) w[i + m r] x[row hop + i + m r],   i < m, q, r < dec
// every access a run of consecutive i; the N = 32768 kernel then transforms `out`'s rows (no overlap, a window of ones).
struct FoldArgs {
    const void   *iq;
    const float  *window;      // bins floats, natural order
    const float2 *rot;         // [dec][m]: exp(-2 pi i q i / bins)
    float2       *out;         // [rows][dec][m]
    int64_t       first_row, rows;
    int           hop, m, dec;
    float         gain;
};
hipError_t launch_fold(int format, const FoldArgs &a, hipStream_t s);
// Last step: out[row][q + dec j] = in[(row dec + q) m + j] -- the dec sub-rows of a stream row interleaved into the
// row, float2 elements (complex spectra of a large transform: bin q + dec j from sub-row q, element j)
struct Interleave2Args {
    const float2 *in;          // [rows][dec][m]
    float2       *out;         // [rows][out_stride >= dec m]
    int64_t       rows, out_stride;
    int           m, dec;
};
hipError_t launch_interleave2(const Interleave2Args &a, hipStream_t s);

// ---- any even length (FFTW takes any N, src/FFTBackend.cpp:120): Bluestein's chirp-z form of the N-point DFT on top
// of the power-of-two transforms of length M >= 2 N - 1,
//   X[k] = c[k] sum_n (x[n] c[n]) conj(c)[k - n],   c[n] = exp(-pi i n^2 / N)
// i.e. a = x w c (zero-padded to M), A = FFT_M(a), y = IFFT_M(A B) with B = FFT_M(conj(c) wrapped), |X[k]| = |y[k]|.
// The three pointwise steps; the two transforms are the library's own.
struct CztArgs {
    const void   *iq;          // pre: sample 0 of the stream
    const float2 *cw;          // pre: [n] window[i] * c[i]
    const float2 *bc;          // mul: [m] conj(B) / M
    float2       *a;           // pre: out [rows][m];  mul: in/out [rows][m] (in: A, out: conj(A) bc = conj(A B) / M)
    const float  *mag;         // out: [rows][m] magnitudes of FFT_M(conj(A B) / M), fft-shifted by m / 2 like every row
    float        *rows_out;    // out: [rows][row_stride], column (k + n/2) mod n = |X[k]|
    int64_t       first_row, rows, row_stride;
    int           hop, n, m;
    float         gain;
};
hipError_t launch_czt_pre(int format, const CztArgs &a, hipStream_t s);
hipError_t launch_czt_mul(const CztArgs &a, hipStream_t s);
hipError_t launch_czt_out(const CztArgs &a, hipStream_t s);
bool       big_supported(int bins);               // power of two in (32768, 2^20]

bool       stft_supported(int bins);
bool       stft_fuses_scan(int bins);             // the plan writes StftArgs::records / tile_out from its epilogue
int        stft_twiddle_count(int bins);          // float2 entries, <0 if unsupported
bool       stft_radices(int bins, int radices[4]);
int        stft_packed_twiddle_count(int bins);   // float4 units of StftArgs::twiddles_k, <0 if unsupported
bool       stft_pack_twiddles(int bins, const float2 *tw, float4 *out);
bool       stft_window_layout(int bins, const float *w, float *out);   // host: bins floats -> bins floats
hipError_t launch_stft(int bins, int fmt, const StftArgs &a, hipStream_t s);
// the N = 32768 magnitude-row kernel with the fused scan / tile epilogue (ro_stft32k.hip); launch_stft routes to it
hipError_t launch_stft32k(int fmt, const StftArgs &a, hipStream_t s);
void       stft32k_window_layout(const float *w, float *out);         // host: 32768 floats -> StftArgs::window_k32
// (diagnostic builds only: tools/r3/ro_stft_wl.hip, the same structure at N = 16384 / 8192)
hipError_t launch_stft_wl(int bins, int fmt, const StftArgs &a, hipStream_t s);
hipError_t launch_scan(const ScanArgs &a, hipStream_t s);
hipError_t launch_tile(const TileArgs &a, hipStream_t s);
hipError_t launch_ln_tile(const LnArgs &a, hipStream_t s);
// per-row log of a compact tile + min / max (plans without the fused epilogue), and the fold of the fused partials
hipError_t launch_ln_rows(const float *tile, float *ln_out, float *minmax, int64_t rows, int cols, hipStream_t s);
hipError_t launch_ln_finish(const float *ln_part, float *minmax, int64_t rows, hipStream_t s);

}  // namespace ro
