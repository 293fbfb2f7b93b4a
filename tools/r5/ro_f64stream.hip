// ro_f64stream.hip -- RO_PRECISION_F64: two passes of the double transform per launch as a PERSISTENT, software-pipelined
// kernel (gfx950).
//
// f64_pair_kernel (ro_kernels.hip) gives every tile of 4096 points its own workgroup: load 16 points per thread, wait,
// pass p, transpose through LDS, pass p + 1 (its twiddles loaded only now), store, exit -- two workgroups per CU, two
// waves per SIMD, ~25k cycles per tile of which ~3.5k are arithmetic; the two launches of the C3 shape move 38.7 B per
// point across the L2 <-> fabric boundary at 3.7 TB/s where a copy reaches 5 (profiles/r05_f64_one_launch.txt).  Here one
// workgroup per CU (256 threads = one wave per SIMD, the whole register file) walks its share of the tiles and keeps
// the memory system busy under the butterflies:
//
//   * the next tile's 16 points per thread (and, behind the first pair, the 15 twiddles of its pass p) are requested
//     right behind the barrier of the current tile, into the registers pass p has just emptied -- a whole pass p + 1
//     and the stores ahead of their use;
//   * the twiddles of pass p + 1 are requested at the head of the tile, a whole pass p ahead of their use (for the
//     first pair they do not depend on the tile at all: loaded once);
//   * the LDS transposition is double-buffered (2 x 64 KiB), so a tile costs ONE workgroup barrier;
//   * stores are never waited for.
//
// Same butterflies, same table entries, same order of operations as f64_pair_tile (ro_f64_device.h): bit-identical rows
// (tests/test_gpu_strict.py, the rows' hash in tools/r5/f64_sweep.py).
//
// ROUND 5 EXPERIMENT, NOT IN THE PRODUCT BUILD: routed in for every pair of passes it was 2 % SLOWER than one workgroup
// per tile at the C3 shape (5.64 against 5.52 ms per 16384 rows, three interleaved rounds; 8192: +1 %, 65536: -5 %,
// 2^20: -9 %; profiles/r05_f64_stream_ab.txt) -- the two launches are not waiting for their loads, they are at what the
// L2 <-> fabric boundary gives 38.7 B per point of write-then-read (3.8 TB/s).  To build it again: add this file to
// build.py's SOURCES, declare launch_f64_pair_stream in ro_kernels.h and call it where launch_transform_f64 calls
// launch_f64_pair.
#include "ro_kernels.h"
#include "ro_f64_device.h"

#include <mutex>

namespace ro {
namespace f64s {

constexpr int THREADS = 256, TILE = 4096, LDS_BYTES = 2 * TILE * 16;

template <int FMT> struct Raw;
template <> struct Raw<RO_FMT_F32> {
    static constexpr int BYTES = 8;
    static __device__ __forceinline__ void load(__amdgpu_buffer_rsrc_t r, int voff, u32x4 &q)
    {
        const u32x2 x = __builtin_amdgcn_raw_buffer_load_b64(r, voff, 0, 0);
        q.x = x.x;
        q.y = x.y;
    }
    static __device__ __forceinline__ v2f widen(const u32x4 &q) { return (v2f){__uint_as_float(q.x), __uint_as_float(q.y)}; }
};
template <> struct Raw<RO_FMT_I16> {
    static constexpr int BYTES = 4;
    static __device__ __forceinline__ void load(__amdgpu_buffer_rsrc_t r, int voff, u32x4 &q)
    {
        q.x = __builtin_amdgcn_raw_buffer_load_b32(r, voff, 0, 0);
    }
    static __device__ __forceinline__ v2f widen(const u32x4 &q)
    {
        return (v2f){(float)(short)(q.x & 0xffffu), (float)(short)(q.x >> 16)};
    }
};

template <int R2, bool FIRST, bool LAST, int FMT>
__global__ __launch_bounds__(THREADS) void f64_pair_stream_kernel(BigArgsD a)
{
    constexpr int R1 = 16, TPW = TILE / (R1 * R2), NB2 = 16 / R2;
    extern __shared__ __attribute__((aligned(16))) char smem_s[];
    double2 *lds = reinterpret_cast<double2 *>(smem_s);
    const int t = threadIdx.x;
    const int ns = FIRST ? 1 : a.ns;
    const int wg_per_row = a.n / TILE;
    const int64_t total = a.rows * (int64_t)wg_per_row;
    // XCD-aware placement, as in the float32 kernels: workgroups b and b + 8 share an XCD (round-robin dispatch) and take
    // neighbouring tiles of one contiguous run, so the tiles of a row -- which read the same samples in the first pair --
    // meet in one L2.  Placement affects speed only.
    const int64_t per_xcd = (total + 7) / 8;
    const int64_t first = (int64_t)(blockIdx.x & 7) * per_xcd;
    const int64_t end = first + per_xcd < total ? first + per_xcd : total;
    const int64_t stride = gridDim.x >> 3;
    int64_t w = first + (blockIdx.x >> 3);
    if (w >= end) return;

    const int tl = t % TPW, kp = t / TPW;                     // pass p: butterfly k' = kp of tile tile0 + tl
    const int per_row = a.n / R1;
    const int step = FIRST ? 0 : a.n / (ns * R1);
    const int ns2 = ns * R1;
    const int step2 = a.n / (ns2 * R2);

    u32x4 buf[R1];                                            // the tile's 16 points per thread as they arrive
    double2 twp[R1 - 1];                                      // twiddles of pass p (behind the first pair)
    double2 tw1[NB2 * (R2 - 1)];                              // twiddles of pass p + 1

    // everything pass p of workgroup-tile `wt` needs, requested (not waited for)
    auto request = [&](int64_t wt) {
        const int64_t row = wt / wg_per_row;
        const int tile = (int)(wt - row * wg_per_row) * TPW + tl, b = tile / ns, c = tile - b * ns;
        const int j = (b + kp * (a.n / (R1 * R2 * ns))) * ns + c;
        if constexpr (FIRST) {
            const int64_t s0 = (a.first_row + row) * (int64_t)a.hop;
            const __amdgpu_buffer_rsrc_t rs =
                make_rsrc(reinterpret_cast<const char *>(a.iq) + s0 * Raw<FMT>::BYTES, (unsigned)a.n * Raw<FMT>::BYTES);
#pragma unroll
            for (int k = 0; k < R1; ++k) {
                const int n = j + k * per_row;
                Raw<FMT>::load(rs, n * Raw<FMT>::BYTES, buf[k]);
                buf[k].z = __float_as_uint(a.window[n]);
            }
        } else {
            const double2 *in = a.in + row * (int64_t)a.n;
            const int kk = j & (ns - 1);
#pragma unroll
            for (int k = 0; k < R1; ++k) {
                buf[k] = *reinterpret_cast<const u32x4 *>(in + (j + k * per_row));
                if (k > 0) twp[k - 1] = a.tw[(int64_t)k * kk * step];
            }
        }
    };
    // the twiddles of pass p + 1 of workgroup-tile `wt`
    auto request_tw1 = [&](int64_t wt) {
        const int64_t row = wt / wg_per_row;
        const int tile0 = (int)(wt - row * wg_per_row) * TPW;
#pragma unroll
        for (int i = 0; i < NB2; ++i) {
            const int u = t + 256 * i, tl2 = u % TPW, sl = u / TPW;
            const int tile = tile0 + tl2, b = tile / ns, c = tile - b * ns;
            const int kk = sl * ns + c;
#pragma unroll
            for (int k = 1; k < R2; ++k) tw1[i * (R2 - 1) + k - 1] = a.tw[(int64_t)k * kk * step2];
        }
    };

    request(w);
    if constexpr (FIRST) request_tw1(w);                      // ns = 1: c = 0, the same for every tile
    for (int it = 0;; ++it) {
        const int64_t row = w / wg_per_row;
        const int tile0 = (int)(w - row * wg_per_row) * TPW;
        double2 *xb = lds + (it & 1) * TILE;
        if constexpr (!FIRST) request_tw1(w);
        {
            // ---- pass p
            v2d v[R1];
#pragma unroll
            for (int k = 0; k < R1; ++k) {
                if constexpr (FIRST) {
                    const v2f x = Raw<FMT>::widen(buf[k]);
                    const double wn = (double)__uint_as_float(buf[k].z);
                    v[k] = (v2d){(double)x.x * wn, ((double)x.y + a.gain) * wn};       // src/FFTBackend.cpp:78-79, :229-232
                } else {
                    v[k] = (v2d){__hiloint2double((int)buf[k].y, (int)buf[k].x), __hiloint2double((int)buf[k].w, (int)buf[k].z)};
                    if (k > 0) v[k] = cmul_d(v[k], (v2d){twp[k - 1].x, twp[k - 1].y});
                }
            }
            dif_d<R1>(v);
#pragma unroll
            for (int k = 0; k < R1; ++k) {
                const v2d x = v[bitrev<R1>(k)];
                xb[(k * R2 + kp) * TPW + tl] = make_double2(x.x, x.y);
            }
        }
        __syncthreads();
        const int64_t next = w + stride;
        const bool has_next = next < end;
        if (has_next) request(next);                          // under pass p + 1 and the stores
        // ---- pass p + 1
#pragma unroll
        for (int i = 0; i < NB2; ++i) {
            const int u = t + 256 * i, tl2 = u % TPW, sl = u / TPW;
            const int tile = tile0 + tl2, b = tile / ns, c = tile - b * ns;
            const int kk = sl * ns + c;
            v2d v[R2];
#pragma unroll
            for (int k = 0; k < R2; ++k) {
                const double2 x = xb[(sl * R2 + k) * TPW + tl2];
                v[k] = (v2d){x.x, x.y};
                if (k > 0) v[k] = cmul_d(v[k], (v2d){tw1[i * (R2 - 1) + k - 1].x, tw1[i * (R2 - 1) + k - 1].y});
            }
            dif_d<R2>(v);
            const int j0 = b * ns2 * R2 + kk;
            if constexpr (LAST) {
                float *out = a.rows_out + row * a.row_stride;
#pragma unroll
                for (int k = 0; k < R2; ++k) {
                    const v2d x = v[bitrev<R2>(k)];
                    __builtin_nontemporal_store((float)sqrt(x.x * x.x + x.y * x.y),
                                                &out[(j0 + k * ns2 + a.n / 2) & (a.n - 1)]);       // WaterfallBackend.cpp:492-505
                }
            } else {
                double2 *out = a.out + row * (int64_t)a.n;
#pragma unroll
                for (int k = 0; k < R2; ++k) {
                    const v2d x = v[bitrev<R2>(k)];
                    out[j0 + k * ns2] = make_double2(x.x, x.y);
                }
            }
        }
        if (!has_next) break;
        w = next;
    }
}

struct DevicePlan {
    bool ready = false;
    int  cus = 0;
};

template <auto KERNEL> static hipError_t prepare(int &cus)
{
    static std::mutex lock;
    static DevicePlan table[64];
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    if (dev < 0 || dev >= 64) return hipErrorInvalidDevice;
    std::lock_guard<std::mutex> g(lock);
    DevicePlan &d = table[dev];
    if (!d.ready) {
        if ((e = hipFuncSetAttribute(reinterpret_cast<const void *>(KERNEL), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES)) != hipSuccess)
            return e;
        if ((e = hipDeviceGetAttribute(&d.cus, hipDeviceAttributeMultiprocessorCount, dev)) != hipSuccess) return e;
        d.ready = true;
    }
    cus = d.cus;
    return hipSuccess;
}

template <int R2, bool FIRST, bool LAST, int FMT> static hipError_t launch_t(const BigArgsD &a, hipStream_t s)
{
    int cus = 0;
    hipError_t e = prepare<&f64_pair_stream_kernel<R2, FIRST, LAST, FMT>>(cus);
    if (e != hipSuccess) return e;
    const int64_t total = a.rows * (int64_t)(a.n / TILE);
    const int64_t per_xcd = (total + 7) / 8;
    int64_t slots = cus / 8;                                   // one workgroup per CU
    if (slots < 1) slots = 1;
    if (slots > per_xcd) slots = per_xcd;
    hipLaunchKernelGGL((f64_pair_stream_kernel<R2, FIRST, LAST, FMT>), dim3((unsigned)(slots * 8)), dim3(THREADS), LDS_BYTES, s, a);
    return hipGetLastError();
}

}  // namespace f64s

// passes p (radix 16, a.ns) and p + 1 (radix r2) in one persistent launch; n >= 4096.  Same contract as launch_f64_pair.
hipError_t launch_f64_pair_stream(int r2, bool first, bool last, int fmt, const BigArgsD &a, hipStream_t s)
{
    using namespace f64s;
    if (a.rows <= 0) return hipSuccess;
    if (a.n < 4096) return hipErrorInvalidValue;
    if (r2 == 16) {
        if (first && !last)
            return fmt == RO_FMT_I16 ? launch_t<16, true, false, RO_FMT_I16>(a, s) : launch_t<16, true, false, RO_FMT_F32>(a, s);
        if (!first) return last ? launch_t<16, false, true, RO_FMT_F32>(a, s) : launch_t<16, false, false, RO_FMT_F32>(a, s);
    } else if (!first && last) {
        switch (r2) {
        case 8: return launch_t<8, false, true, RO_FMT_F32>(a, s);
        case 4: return launch_t<4, false, true, RO_FMT_F32>(a, s);
        case 2: return launch_t<2, false, true, RO_FMT_F32>(a, s);
        }
    }
    return hipErrorInvalidValue;
}

}  // namespace ro
