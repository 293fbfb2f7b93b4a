// ro_kernels.hip -- hand-written gfx950 kernels of the STFT / waterfall / scan path.
//
//   stft_kernel : window -> Stockham FFT -> |X| -> fft-shift -> float32 row
//                 replaces src/FFTBackend.cpp:229-236 (window multiply + fftw_execute)
//                 and src/WaterfallBackend.cpp:485-505 (magnitude + shift) of the reference,
//                 and the framing loop :211-257 by addressing row r at sample r*hop.
//   scan_kernel : BolidRecorder::noise / peak / average per row
//                 (src/BolidRecorder.cpp:121-132, :313-347).
//
// One workgroup transforms one row; the whole row lives in the workgroup's
// registers (P = N/T points per thread) and crosses LDS twice (three stages).
// HBM traffic per row is the algorithmic minimum: hop*8 B of new samples
// (overlap re-reads are served by L2 -- consecutive rows are placed on the
// same XCD) plus bins*4 B of magnitudes.
#include "ro_kernels.h"
#include "ro_fft_device.h"

namespace ro {

// ---------------------------------------------------------------------------
// plan
// ---------------------------------------------------------------------------
template <int N_, int T_, int R0_, int R1_, int R2_, int R3_, bool SPLIT_>
struct Plan {
    static constexpr int N = N_, T = T_, P = N_ / T_;
    static constexpr int R0 = R0_, R1 = R1_, R2 = R2_, R3 = R3_;
    static constexpr bool SPLIT = SPLIT_;
    static constexpr int NS1 = R0, NS2 = R0 * R1, NS3 = R0 * R1 * R2;
    static constexpr int TW1 = 0;                                   // float2 offsets into the table
    static constexpr int TW2 = TW1 + (R1 > 1 ? (R1 - 1) * NS1 : 0);
    static constexpr int TW3 = TW2 + (R2 > 1 ? (R2 - 1) * NS2 : 0);
    static constexpr int TW_TOTAL = TW3 + (R3 > 1 ? (R3 - 1) * NS3 : 0);
    static constexpr int LDS_ELEMS = N + N / 32;                    // padded
    static constexpr int LDS_BYTES = LDS_ELEMS * (SPLIT ? 4 : 8);
    static_assert(R0 * R1 * R2 * R3 == N, "radices must multiply to N");
    static_assert(P % R0 == 0 && P % R1 == 0 && P % R2 == 0 && P % R3 == 0, "radix must divide P");
};

// ---------------------------------------------------------------------------
// stage helpers (all indices compile-time after unrolling -> v[] stays in VGPRs)
// ---------------------------------------------------------------------------
template <int P, int R> __device__ __forceinline__ void butterflies(float2 (&v)[P])
{
#pragma unroll
    for (int b = 0; b < P / R; ++b) dif<R>(&v[b * R]);
}

// Buffer-descriptor helpers.  All global traffic of the STFT kernel goes through
// raw buffer instructions: one 32-bit per-lane offset VGPR per stream, the
// per-register part of the address in an SGPR (soffset), and free hardware
// bounds checking (out-of-range loads give 0, stores are dropped).
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void *base, unsigned bytes)
{
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(base), 0, (int)bytes, 0x00020000);
}
__device__ __forceinline__ float2 buf_load_f2(__amdgpu_buffer_rsrc_t r, int voff, int soff)
{
    u32x2 t = __builtin_amdgcn_raw_buffer_load_b64(r, voff, soff, 0);
    return make_float2(__uint_as_float(t.x), __uint_as_float(t.y));
}
__device__ __forceinline__ float buf_load_f(__amdgpu_buffer_rsrc_t r, int voff, int soff)
{
    return __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r, voff, soff, 0));
}
__device__ __forceinline__ void buf_store_f(float x, __amdgpu_buffer_rsrc_t r, int voff, int soff)
{
    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(x), r, voff, soff, 0);
}

// multiply by the inter-stage twiddles of a stage (R, NS); table entry OFF + (r-1)*NS + k.
// Loaded TW_CHUNK at a time behind scheduling fences: left alone, the scheduler
// hoists all R-1 loads above the LDS exchange and the row no longer fits the
// 128-VGPR budget of a 1024-thread workgroup.
constexpr int TW_CHUNK = 8;

// Ordering by fake data dependence: returns `off` unchanged, but the compiler must
// assume it was recomputed from `x`, so loads addressed with the result cannot be
// issued before `x` exists.  (A "memory" clobber does not stop the scheduler from
// clustering buffer loads; this does, and costs no instruction.)
__device__ __forceinline__ int after(int off, float x)
{
    asm volatile("" : "+v"(off) : "v"(x));
    return off;
}

template <int P, int T, int R, int NS, int OFF>
__device__ __forceinline__ void apply_twiddles(float2 (&v)[P], __amdgpu_buffer_rsrc_t tw, int tid)
{
    constexpr int NCH = (R - 1 + TW_CHUNK - 1) / TW_CHUNK;      // chunks per butterfly
#pragma unroll
    for (int b = 0; b < P / R; ++b) {
        // not before the gather that filled v[] has finished
        int koff = after(((tid + T * b) & (NS - 1)) * 8, v[b * R + R - 1].y);
        float2 t[2][TW_CHUNK];
        // two chunks in flight: chunk c+1 is issued before chunk c is consumed
#pragma unroll
        for (int c = 0; c <= NCH; ++c) {
            if (c < NCH) {
                // ... and not before chunk c-2 has been consumed (its registers are reused)
                if (c >= 2) koff = after(koff, v[b * R + (c - 2) * TW_CHUNK + 1].x);
#pragma unroll
                for (int i = 0; i < TW_CHUNK; ++i) {
                    const int r = 1 + c * TW_CHUNK + i;
                    if (r < R) t[c & 1][i] = buf_load_f2(tw, koff, (OFF + (r - 1) * NS) * 8);
                }
            }
            if (c > 0) {
#pragma unroll
                for (int i = 0; i < TW_CHUNK; ++i) {
                    const int r = 1 + (c - 1) * TW_CHUNK + i;
                    if (r < R) v[b * R + r] = cmul(v[b * R + r], t[(c - 1) & 1][i]);
                }
            }
        }
    }
}

// autosort scatter of a finished stage (R, NS) into LDS (element index, padded).
// The padded index i + (i>>5) is affine in r when r*NS never carries into bit 5 on
// its own (NS a multiple of 32) or for the first stage of radix 32 (i = 32*j + r):
// then one base VGPR + immediate offsets address the whole scatter.  Otherwise each
// element computes its own address.
template <int P, int T, int R, int NS, typename E, typename F>
__device__ __forceinline__ void lds_scatter(E *lds, const float2 (&v)[P], int tid, F pick)
{
#pragma unroll
    for (int b = 0; b < P / R; ++b) {
        const int j = tid + T * b;
        const int j0 = (j / NS) * (NS * R) + (j & (NS - 1));
        if constexpr (NS % 32 == 0) {
            E *base = lds + lds_pad(j0);
#pragma unroll
            for (int r = 0; r < R; ++r) base[r * (NS + NS / 32)] = pick(v[b * R + bitrev<R>(r)]);
        } else if constexpr (NS == 1 && R == 32) {
            E *base = lds + 33 * j;
#pragma unroll
            for (int r = 0; r < R; ++r) base[r] = pick(v[b * R + bitrev<R>(r)]);
        } else {
#pragma unroll
            for (int r = 0; r < R; ++r) lds[lds_pad(j0 + r * NS)] = pick(v[b * R + bitrev<R>(r)]);
        }
    }
}

// gather for the next stage of radix R:  v[b*R + r] = lds[j + r*(N/R)]; N/R is a
// multiple of 32 for every plan, so the padded address is base + r*(N/R)*33/32.
template <int N, int P, int T, int R, typename E, typename F>
__device__ __forceinline__ void lds_gather(const E *lds, float2 (&v)[P], int tid, F put)
{
    static_assert((N / R) % 32 == 0, "gather stride must be a multiple of 32");
#pragma unroll
    for (int b = 0; b < P / R; ++b) {
        const E *base = lds + lds_pad(tid + T * b);
#pragma unroll
        for (int r = 0; r < R; ++r) put(v[b * R + r], base[r * (N / R + N / R / 32)]);
    }
}

// full exchange between a finished stage (RA, NS) and the next stage of radix RB.
// SPLIT (row too large for LDS as float2): the real plane goes first; gathering it
// into v[].x leaves v[].y in the old register order for the second scatter.
template <class PL, int RA, int NS, int RB>
__device__ __forceinline__ void exchange(void *smem, float2 (&v)[PL::P], int tid)
{
    constexpr int N = PL::N, P = PL::P, T = PL::T;
    if constexpr (PL::SPLIT) {
        float *lds = reinterpret_cast<float *>(smem);
        lds_scatter<P, T, RA, NS>(lds, v, tid, [](float2 e) { return e.x; });
        __syncthreads();
        lds_gather<N, P, T, RB>(lds, v, tid, [](float2 &d, float s) { d.x = s; });
        __syncthreads();
        lds_scatter<P, T, RA, NS>(lds, v, tid, [](float2 e) { return e.y; });
        __syncthreads();
        lds_gather<N, P, T, RB>(lds, v, tid, [](float2 &d, float s) { d.y = s; });
        __syncthreads();
    } else {
        float2 *lds = reinterpret_cast<float2 *>(smem);
        lds_scatter<P, T, RA, NS>(lds, v, tid, [](float2 e) { return e; });
        __syncthreads();
        lds_gather<N, P, T, RB>(lds, v, tid, [](float2 &d, float2 s) { d = s; });
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------
// sample loads
// ---------------------------------------------------------------------------
template <int FMT> struct Sample;
template <> struct Sample<RO_FMT_F32> {
    static constexpr int BYTES = 8;
    static __device__ __forceinline__ float2 load(__amdgpu_buffer_rsrc_t r, int voff, int soff)
    {
        return buf_load_f2(r, voff, soff);
    }
};
template <> struct Sample<RO_FMT_I16> {
    static constexpr int BYTES = 4;
    static __device__ __forceinline__ float2 load(__amdgpu_buffer_rsrc_t r, int voff, int soff)
    {
        const unsigned u = __builtin_amdgcn_raw_buffer_load_b32(r, voff, soff, 0);
        return make_float2((float)(short)(u & 0xffffu), (float)(short)(u >> 16));
    }
};

// ---------------------------------------------------------------------------
// the STFT kernel
// ---------------------------------------------------------------------------
template <class PL, int FMT>
__global__ __launch_bounds__(PL::T) void stft_kernel(StftArgs a)
{
    constexpr int N = PL::N, T = PL::T, P = PL::P;
    extern __shared__ __attribute__((aligned(16))) char smem[];

    // XCD-aware placement: workgroups b and b+8 share an XCD (round-robin
    // dispatch), so give each XCD one contiguous run of rows -- consecutive
    // rows share (N-hop)/N of their input through that XCD's L2.
    const int64_t per_xcd = (a.rows + 7) / 8;
    const int64_t row = (int64_t)(blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
    if (row >= a.rows) return;

    const int tid = threadIdx.x;
    const int64_t s0 = (a.first_row + row) * (int64_t)a.hop;
    using S = Sample<FMT>;

    const __amdgpu_buffer_rsrc_t rs_iq =
        make_rsrc(reinterpret_cast<const char *>(a.iq) + s0 * S::BYTES, N * S::BYTES);
    const __amdgpu_buffer_rsrc_t rs_win = make_rsrc(a.window, N * 4);
    const __amdgpu_buffer_rsrc_t rs_tw = make_rsrc(a.twiddles, PL::TW_TOTAL * 8);

    float2 v[P];

    // ---- stage 0: samples * window straight from global memory.  All sample loads go
    // out first (they land in the data registers); the window coefficients follow in
    // chunks of WIN_CHUNK, two chunks in flight, so the prologue peaks at P*2 + 2*WIN_CHUNK VGPRs.
    {
        constexpr int R = PL::R0;
        constexpr int WIN_CHUNK = P < 8 ? P : 8;
        constexpr int NCH = P / WIN_CHUNK;
#pragma unroll
        for (int i = 0; i < P; ++i) {
            const int b = i / R, r = i % R;
            v[i] = S::load(rs_iq, (tid + T * b) * S::BYTES, r * (N / R) * S::BYTES);
        }
        float w[2][WIN_CHUNK];
        int woff = tid * 4;
#pragma unroll
        for (int c = 0; c <= NCH; ++c) {
            if (c < NCH) {
                if (c >= 2) woff = after(woff, v[(c - 2) * WIN_CHUNK].x);
#pragma unroll
                for (int q = 0; q < WIN_CHUNK; ++q) {
                    const int i = c * WIN_CHUNK + q, b = i / R, r = i % R;
                    w[c & 1][q] = buf_load_f(rs_win, woff, (T * b + r * (N / R)) * 4);
                }
            }
            if (c > 0) {
#pragma unroll
                for (int q = 0; q < WIN_CHUNK; ++q) {
                    const int i = (c - 1) * WIN_CHUNK + q;
                    const float ww = w[(c - 1) & 1][q];
                    v[i] = make_float2(v[i].x * ww, (v[i].y + a.gain) * ww);
                }
            }
        }
        butterflies<P, R>(v);
    }

    // ---- stage 1
    if constexpr (PL::R1 > 1) {
        exchange<PL, PL::R0, 1, PL::R1>(smem, v, tid);
        apply_twiddles<P, T, PL::R1, PL::NS1, PL::TW1>(v, rs_tw, tid);
        butterflies<P, PL::R1>(v);
    }
    // ---- stage 2
    if constexpr (PL::R2 > 1) {
        exchange<PL, PL::R1, PL::NS1, PL::R2>(smem, v, tid);
        apply_twiddles<P, T, PL::R2, PL::NS2, PL::TW2>(v, rs_tw, tid);
        butterflies<P, PL::R2>(v);
    }
    // ---- stage 3
    if constexpr (PL::R3 > 1) {
        exchange<PL, PL::R2, PL::NS2, PL::R3>(smem, v, tid);
        apply_twiddles<P, T, PL::R3, PL::NS3, PL::TW3>(v, rs_tw, tid);
        butterflies<P, PL::R3>(v);
    }

    // ---- epilogue: |X[k]| to column (k + N/2) mod N   (src/WaterfallBackend.cpp:492-505)
    // k = j + r*(N/RL) with j < N/RL, so the shifted column is j + a per-register constant.
    constexpr int RL = PL::R3 > 1 ? PL::R3 : (PL::R2 > 1 ? PL::R2 : (PL::R1 > 1 ? PL::R1 : PL::R0));
    const bool want_rows = a.rows_out != nullptr;
    const bool want_tile = a.tile_out != nullptr;
    const __amdgpu_buffer_rsrc_t rs_out =
        make_rsrc(want_rows ? a.rows_out + row * a.row_stride : nullptr, want_rows ? N * 4 : 0);
    const __amdgpu_buffer_rsrc_t rs_tile =
        make_rsrc(want_tile ? a.tile_out + row * (int64_t)a.tile_cols : nullptr,
                  want_tile ? a.tile_cols * 4 : 0);
#pragma unroll
    for (int b = 0; b < P / RL; ++b) {
        const int j = tid + T * b;
#pragma unroll
        for (int r = 0; r < RL; ++r) {
            const float2 x = v[b * RL + bitrev<RL>(r)];
            const float m = __builtin_amdgcn_sqrtf(x.x * x.x + x.y * x.y);   // v_sqrt_f32, 1 ulp
            const int cbase = (r * (N / RL) + N / 2) & (N - 1);
            buf_store_f(m, rs_out, j * 4, cbase * 4);
            if (want_tile) {
                const int tc = j + cbase - a.tile_first;
                // out-of-tile lanes get an offset the descriptor's range check drops
                buf_store_f(m, rs_tile, (tc >= 0 && tc < a.tile_cols) ? tc * 4 : 0x40000000, 0);
            }
        }
    }
}

// ---------------------------------------------------------------------------
// per-row band scan (one wavefront per row)
// ---------------------------------------------------------------------------
__device__ __forceinline__ unsigned order_key(float x)
{
    unsigned u = __float_as_uint(x);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float key_to_float(unsigned k)
{
    unsigned u = (k & 0x80000000u) ? (k & 0x7fffffffu) : ~k;
    return __uint_as_float(u);
}

constexpr int SCAN_E = 16;            // noise-band elements cached per lane (band <= 1024)
constexpr int SCAN_WAVES = 4;         // rows per workgroup

__global__ __launch_bounds__(64 * SCAN_WAVES) void scan_kernel(ScanArgs a)
{
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * SCAN_WAVES + (threadIdx.x >> 6);
    if (row >= a.rows) return;
    const float *__restrict__ src = a.rows_in + row * a.row_stride;

    // ---- noise(): element floor(W/4) of the ascending noise band, times two.
    // Order statistic by a 32-step bisection on the order-preserving integer
    // image of the floats: result = largest key K with #(key < K) <= k.
    const float *nb = src + a.low_noise;
    const int W = a.noise_width;
    const int kth = W / 4;
    unsigned keys[SCAN_E];
    const bool cached = W <= 64 * SCAN_E;
    if (cached) {
#pragma unroll
        for (int e = 0; e < SCAN_E; ++e) {
            const int i = lane + 64 * e;
            keys[e] = i < W ? order_key(nb[i]) : 0xffffffffu;   // padding sorts last
        }
    }
    unsigned result = 0;
    for (int bit = 31; bit >= 0; --bit) {
        const unsigned trial = result | (1u << bit);
        int below = 0;
        if (cached) {
#pragma unroll
            for (int e = 0; e < SCAN_E; ++e)
                below += __popcll(__ballot(keys[e] < trial));
        } else {
            for (int i0 = 0; i0 < W; i0 += 64) {
                const int i = i0 + lane;
                const bool lt = i < W && order_key(nb[i]) < trial;
                below += __popcll(__ballot(lt));
            }
        }
        if (below <= kth) result = trial;
    }
    // padding keys (0xffffffff) are never counted as "< trial" unless trial is larger, which
    // cannot happen, so `below` only ever counts real elements.
    const float q = key_to_float(result);
    const float noise = (float)((double)q * 2.0);

    // ---- peak(): last index of the maximum of the detect band
    const float *db = src + a.low_detect;
    float best = 0.f;
    int best_i = -1;
    for (int i = lane; i < a.detect_width; i += 64) {
        const float x = db[i];
        if (best_i < 0 || x >= best) { best = x; best_i = i; }
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        const float ob = __shfl_xor(best, off);
        const int oi = __shfl_xor(best_i, off);
        const bool take = (oi >= 0) && (best_i < 0 || ob > best || (ob == best && oi > best_i));
        if (take) { best = ob; best_i = oi; }
    }
    const int peak = best_i < 0 ? 0 : best_i;

    // ---- average(): sequential double sum in index order, like the reference
    if (lane == 0) {
        const int start = a.low_detect + peak - a.avg_bins / 2;
        double acc = 0.0;
        for (int i = 0; i < a.avg_bins; ++i) {
            const int c = start + i;
            // the reference reads outside the row here when the window leaves it (UB);
            // columns outside [0, bins) contribute nothing in this implementation.
            if (c >= 0 && c < a.bins) acc += (double)src[c];
        }
        ro_scan_record_t rec;
        rec.noise = noise;
        rec.peak = peak;
        rec.average = (float)(acc / (double)a.avg_bins);
        a.records[row] = rec;
    }
}

// ---------------------------------------------------------------------------
// launch table
// ---------------------------------------------------------------------------
template <class PL, int FMT> static hipError_t launch_plan(const StftArgs &a, hipStream_t s)
{
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&stft_kernel<PL, FMT>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, PL::LDS_BYTES);
        if (e != hipSuccess) return e;
        attr_set = true;
    }
    const int64_t per_xcd = (a.rows + 7) / 8;
    const unsigned grid = (unsigned)(per_xcd * 8);
    hipLaunchKernelGGL((stft_kernel<PL, FMT>), dim3(grid), dim3(PL::T), PL::LDS_BYTES, s, a);
    return hipGetLastError();
}

//                      N      T   R0  R1  R2  R3  split
using Plan32768 = Plan<32768, 1024, 32, 32, 32, 1, true>;
using Plan16384 = Plan<16384,  512, 32, 32, 16, 1, true>;
using Plan8192  = Plan< 8192,  256, 32, 32,  8, 1, false>;
using Plan4096  = Plan< 4096,  256, 16, 16, 16, 1, false>;
using Plan2048  = Plan< 2048,  128, 16, 16,  8, 1, false>;
using Plan1024  = Plan< 1024,   64, 16, 16,  4, 1, false>;
using Plan512   = Plan<  512,   64,  8,  8,  8, 1, false>;
using Plan256   = Plan<  256,   64,  4,  4,  4, 4, false>;

template <class PL> static hipError_t launch_fmt(const StftArgs &a, int fmt, hipStream_t s)
{
    if (fmt == RO_FMT_F32) return launch_plan<PL, RO_FMT_F32>(a, s);
    if (fmt == RO_FMT_I16) return launch_plan<PL, RO_FMT_I16>(a, s);
    return hipErrorInvalidValue;
}

bool stft_supported(int bins)
{
    switch (bins) {
    case 256: case 512: case 1024: case 2048: case 4096: case 8192: case 16384: case 32768:
        return true;
    default:
        return false;
    }
}

int stft_twiddle_count(int bins)
{
    switch (bins) {
    case 32768: return Plan32768::TW_TOTAL;
    case 16384: return Plan16384::TW_TOTAL;
    case 8192:  return Plan8192::TW_TOTAL;
    case 4096:  return Plan4096::TW_TOTAL;
    case 2048:  return Plan2048::TW_TOTAL;
    case 1024:  return Plan1024::TW_TOTAL;
    case 512:   return Plan512::TW_TOTAL;
    case 256:   return Plan256::TW_TOTAL;
    default:    return -1;
    }
}

template <class PL> static void fill_radices(int *r) { r[0] = PL::R0; r[1] = PL::R1; r[2] = PL::R2; r[3] = PL::R3; }

bool stft_radices(int bins, int radices[4])
{
    switch (bins) {
    case 32768: fill_radices<Plan32768>(radices); return true;
    case 16384: fill_radices<Plan16384>(radices); return true;
    case 8192:  fill_radices<Plan8192>(radices);  return true;
    case 4096:  fill_radices<Plan4096>(radices);  return true;
    case 2048:  fill_radices<Plan2048>(radices);  return true;
    case 1024:  fill_radices<Plan1024>(radices);  return true;
    case 512:   fill_radices<Plan512>(radices);   return true;
    case 256:   fill_radices<Plan256>(radices);   return true;
    default:    return false;
    }
}

hipError_t launch_stft(int bins, int fmt, const StftArgs &a, hipStream_t s)
{
    if (a.rows <= 0) return hipSuccess;
    switch (bins) {
    case 32768: return launch_fmt<Plan32768>(a, fmt, s);
    case 16384: return launch_fmt<Plan16384>(a, fmt, s);
    case 8192:  return launch_fmt<Plan8192>(a, fmt, s);
    case 4096:  return launch_fmt<Plan4096>(a, fmt, s);
    case 2048:  return launch_fmt<Plan2048>(a, fmt, s);
    case 1024:  return launch_fmt<Plan1024>(a, fmt, s);
    case 512:   return launch_fmt<Plan512>(a, fmt, s);
    case 256:   return launch_fmt<Plan256>(a, fmt, s);
    default:    return hipErrorInvalidValue;
    }
}

hipError_t launch_scan(const ScanArgs &a, hipStream_t s)
{
    if (a.rows <= 0) return hipSuccess;
    const unsigned grid = (unsigned)((a.rows + SCAN_WAVES - 1) / SCAN_WAVES);
    hipLaunchKernelGGL(scan_kernel, dim3(grid), dim3(64 * SCAN_WAVES), 0, s, a);
    return hipGetLastError();
}

}  // namespace ro
