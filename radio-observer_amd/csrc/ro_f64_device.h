// ro_f64_device.h -- device functions of the strict-precision path (RO_PRECISION_F64): double butterflies and the
// two-passes-per-tile body shared by f64_pair_kernel (ro_kernels.hip: one launch per pair of passes, the intermediate
// in HBM scratch) and f64_fused_kernel (ro_f64fused.hip: all four passes of a row in one launch, the intermediate in
// the XCD's L2).  Same butterflies, same table entries, same order in both: bit-identical rows.
#pragma once

#include <hip/hip_runtime.h>

#include "ro_kernels.h"
#include "ro_fft_device.h"
#include "ro_device_util.h"

// cache policy of the loads of an intermediate that OTHER workgroups of the same launch wrote (gfx950 aux bits: 1 = sc0,
// 2 = nt, 16 = sc1); tools/r5/f64f_loads.sh measures them
#ifndef RO_F64F_LOAD_AUX
#define RO_F64F_LOAD_AUX 16
#endif

namespace ro {

typedef double v2d __attribute__((ext_vector_type(2)));

__device__ __forceinline__ v2d cmul_d(v2d a, v2d w) { return (v2d){a.x * w.x - a.y * w.y, a.x * w.y + a.y * w.x}; }

// exp(-2 pi i M / 16) for M = 0..7 (the rest by symmetry inside dif_d)
template <int M> __device__ __forceinline__ v2d mul_w16_d(v2d d)
{
    constexpr double C1 = 0.92387953251128675613, C2 = 0.70710678118654752440, C3 = 0.38268343236508977173;
    if constexpr (M == 0) return d;
    else if constexpr (M == 4) return (v2d){d.y, -d.x};                       // * (-i)
    else {
        constexpr double c = (M == 1) ? C1 : (M == 2) ? C2 : (M == 3) ? C3 : (M == 5) ? -C3 : (M == 6) ? -C2 : -C1;
        constexpr double sn = (M == 1) ? C3 : (M == 2) ? C2 : (M == 3) ? C1 : (M == 5) ? C1 : (M == 6) ? C2 : C3;
        return (v2d){d.x * c + d.y * sn, d.y * c - d.x * sn};                 // d * (c - i sn)
    }
}

// in-place decimation-in-frequency DFT of R points (R in {2,4,8,16}); result k sits at v[bitrev_R(k)]
template <int R> __device__ __forceinline__ void dif_d(v2d *v)
{
    if constexpr (R >= 2) {
#pragma unroll
        for (int i = 0; i < R / 2; ++i) {
            const v2d a = v[i], b = v[i + R / 2];
            v[i] = a + b;
            const v2d d = a - b;
            // twiddle W_R^i = W_16^(i * 16 / R)
            switch (i * (16 / R)) {
            case 0: v[i + R / 2] = d; break;
            case 1: v[i + R / 2] = mul_w16_d<1>(d); break;
            case 2: v[i + R / 2] = mul_w16_d<2>(d); break;
            case 3: v[i + R / 2] = mul_w16_d<3>(d); break;
            case 4: v[i + R / 2] = mul_w16_d<4>(d); break;
            case 5: v[i + R / 2] = mul_w16_d<5>(d); break;
            case 6: v[i + R / 2] = mul_w16_d<6>(d); break;
            default: v[i + R / 2] = mul_w16_d<7>(d); break;
            }
        }
        dif_d<R / 2>(v);
        dif_d<R / 2>(v + R / 2);
    }
}


// Two passes of the recurrence on one tile of 4096 points: pass p (radix 16, sub-length ns) and pass p+1 (radix R2,
// sub-length 16 ns) share tiles of 16 R2 points -- the R2 butterflies j = (b + k' N/(16 R2 ns)) ns + c of pass p
// produce exactly the inputs of the 16 butterflies j' = 16 b ns + s ns + c of pass p+1 (s < 16), for every (b, c) --
// so a workgroup of 256 threads takes 4096 / (16 R2) neighbouring tiles starting at tile0, runs pass p, transposes
// through 64 KiB of LDS and runs pass p+1.  `row` counts from a.first_row (samples, a.rows_out); in_row / out_row
// point at the row's N complex doubles (pass p's input unless FIRST, pass p+1's output unless LAST).  SC1IN: the
// input was written by OTHER workgroups of this launch (the fused kernel): every load of it goes past this CU's L1
// (buffer loads with sc1), which another CU's stores never refresh.
template <int R2, bool FIRST, bool LAST, int FMT, bool SC1IN>
__device__ __forceinline__ void f64_pair_tile(const BigArgsD &a, double2 *lds, int64_t row, int tile0,
                                              const double2 *in_row, double2 *out_row)
{
    constexpr int R1 = 16, TPW = 4096 / (R1 * R2), NB2 = 16 / R2;     // tiles per workgroup; pass-(p+1) butterflies per thread
    const int t = threadIdx.x;
    const int ns = a.ns;
    {
        // ---- pass p: thread = butterfly k' of tile tile0 + (t % TPW)
        const int tl = t % TPW, kp = t / TPW;
        const int tile = tile0 + tl, b = tile / ns, c = tile - b * ns;
        const int j = (b + kp * (a.n / (R1 * R2 * ns))) * ns + c;
        const int per_row = a.n / R1;
        v2d v[R1];
        if constexpr (FIRST) {
            using S = Sample<FMT>;
            const int64_t s0 = (a.first_row + row) * (int64_t)a.hop;
            const __amdgpu_buffer_rsrc_t rs =
                make_rsrc(reinterpret_cast<const char *>(a.iq) + s0 * S::BYTES, (unsigned)a.n * S::BYTES);
#pragma unroll
            for (int k = 0; k < R1; ++k) {
                const int n = j + k * per_row;
                const double w = (double)a.window[n];
                const v2f x = S::load(rs, n * S::BYTES, 0);
                v[k] = (v2d){(double)x.x * w, ((double)x.y + a.gain) * w};       // src/FFTBackend.cpp:78-79, :229-232
            }
        } else {
            const int kk = j & (ns - 1);
            const int step = a.n / (ns * R1);
            [[maybe_unused]] const __amdgpu_buffer_rsrc_t ri = make_rsrc(in_row, (unsigned)a.n * 16u);
#pragma unroll
            for (int k = 0; k < R1; ++k) {
                if constexpr (SC1IN) {
                    const u32x4 q = __builtin_amdgcn_raw_buffer_load_b128(ri, (j + k * per_row) * 16, 0, RO_F64F_LOAD_AUX);
                    v[k] = (v2d){__hiloint2double((int)q.y, (int)q.x), __hiloint2double((int)q.w, (int)q.z)};
                } else {
                    const double2 x = in_row[j + k * per_row];
                    v[k] = (v2d){x.x, x.y};
                }
                if (k > 0) {
                    const double2 w = a.tw[(int64_t)k * kk * step];
                    v[k] = cmul_d(v[k], (v2d){w.x, w.y});
                }
            }
        }
        dif_d<R1>(v);
#pragma unroll
        for (int k = 0; k < R1; ++k) {
            const v2d x = v[bitrev<R1>(k)];
            lds[(k * R2 + kp) * TPW + tl] = make_double2(x.x, x.y);
        }
    }
    __syncthreads();
    // ---- pass p+1: butterfly u = (s, tile): reads slot s of the tile's R2 pass-p butterflies
    const int ns2 = ns * R1;
    const int step2 = a.n / (ns2 * R2);
#pragma unroll
    for (int i = 0; i < NB2; ++i) {
        const int u = t + 256 * i, tl = u % TPW, sl = u / TPW;
        const int tile = tile0 + tl, b = tile / ns, c = tile - b * ns;
        const int kk = sl * ns + c;                                   // j' mod (16 ns),  j' = 16 b ns + kk
        v2d v[R2];
#pragma unroll
        for (int k = 0; k < R2; ++k) {
            const double2 x = lds[(sl * R2 + k) * TPW + tl];
            v[k] = (v2d){x.x, x.y};
            if (k > 0) {
                const double2 w = a.tw[(int64_t)k * kk * step2];
                v[k] = cmul_d(v[k], (v2d){w.x, w.y});
            }
        }
        dif_d<R2>(v);
        const int j0 = b * ns2 * R2 + kk;
        if constexpr (LAST) {
            float *out = a.rows_out + row * a.row_stride;
#pragma unroll
            for (int k = 0; k < R2; ++k) {
                const v2d x = v[bitrev<R2>(k)];
                // (the rows are write-once: nt, like every other kernel's row stores -- 4 % on this path; the scratch
                // between the two kernels stays on the default policy, nt there loses the Infinity Cache: -7 ... -20 %)
                __builtin_nontemporal_store((float)sqrt(x.x * x.x + x.y * x.y),
                                            &out[(j0 + k * ns2 + a.n / 2) & (a.n - 1)]);       // WaterfallBackend.cpp:492-505
            }
        } else {
#pragma unroll
            for (int k = 0; k < R2; ++k) {
                const v2d x = v[bitrev<R2>(k)];
                out_row[j0 + k * ns2] = make_double2(x.x, x.y);
            }
        }
    }
}

}  // namespace ro
