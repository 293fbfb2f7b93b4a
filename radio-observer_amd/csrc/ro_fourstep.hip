// ro_fourstep.hip -- the large transforms (bins = N1 x 1024: 262144 and Ionozor's 524288, Ionozor.json:27) as a
// four-step FFT: two kernels, ONE trip through HBM scratch, every global access a run of whole cache lines.
//   replaces src/FFTBackend.cpp:229-236 (window multiply + fftw_execute) and src/WaterfallBackend.cpp:485-505
//   (magnitude + fft-shift) of the reference for those sizes.
//
//   n = 1024 n1 + n2          k = k1 + N1 k2                                   (n1, k1 < N1;  n2, k2 < 1024)
//   X[k1 + N1 k2] = sum_n2 W_1024^(n2 k2) { W_M^(n2 k1) sum_n1 W_N1^(n1 k1) w[n] x[n] }
//
// four_cols_kernel  (the inner sum: N1-point transforms down the columns n2).  A workgroup takes C = 32768 / N1
//   neighbouring columns of one stream row -- 32768 points, 32 per thread, the row kernel's shape.  n1 = m + R2 l
//   (R2 = N1 / 32): thread (m, column) loads its 32 legs l (each wave-load 64 neighbouring samples), multiplies by the
//   window, radix-32 over l -> k_l; an exchange over the workgroup through LDS (one component plane at a time, like
//   exchange 1 of the row kernel); thread (k_l's, column) multiplies by W_N1^(m k_l) and finishes with radix-R2 over m
//   -> k1 = k_l + 32 k_m.  Z[k1][n2] goes to scratch in the order the second kernel reads it (below).
// four_rows_kernel  (the braces' factor and the outer sum: 1024-point transforms along the rows k1).  It IS passes 1
//   and 2 of ro_stft32k.hip -- same thread maps, same LDS layout (ro_k32_lds.h), same planar butterflies, same
//   wave-local exchange 2 -- on 32 rows k1 = 32 g .. 32 g + 31 at a time (wave w: rows 2 w, 2 w + 1), n2 = a + 32 b:
//   the factor W_M^(n2 k1) splits into W_M^(32 b k1), which is pass 1's stage twiddle (two values per wave: scalar
//   loads), and W_M^(a k1), a factor of pass 2's.  The 32 x 1024 magnitudes of a block are bins
//   k1 + N1 (k_r + 32 k_c): the read-back of the image finds FOUR consecutive k1 in a lane and a whole 128-byte line
//   of the fft-shifted row in eight lanes.  Two barriers per block.
//
// Scratch order ("planar pairs"): row k1 = 16 x 32 quads, quad (i, a) = { re Z[a + 32 (2i)], re Z[a + 32 (2i + 1)],
//   im ..., im ... } -- what thread (k1, a) of pass 1 holds in R[i], I[i]: one 16-byte load per register quad, no
//   shuffling.  The column kernel gets there with one v_permlane32_swap per point (lanes c, c + 32 of its waves hold
//   the two mates).
//
// HBM / Infinity Cache traffic per stream row: 8 hop (samples) + 8 M out + 8 M in (scratch) + 4 M (the row) against the
// 28 M of the fold / transform / interleave form it replaces.
#define RO_TIE_SCHED 1
#include "ro_kernels.h"
#include "ro_fft_device.h"
#include "ro_fft_planar.h"
#include "ro_device_util.h"
#include "ro_k32_lds.h"

#include <cmath>
#include <mutex>
#include <vector>

// cache policy of the scratch traffic (gfx950 aux bits: 1 = sc0, 2 = nt, 16 = sc1)
#ifndef RO_FOUR_Z_ST_AUX
#define RO_FOUR_Z_ST_AUX 0
#endif
#ifndef RO_FOUR_Z_LD_AUX
#define RO_FOUR_Z_LD_AUX 2
#endif

namespace ro {
namespace four {

using k32::RQ;
using k32::HB;
using k32::T;
using k32::own_write4;
using k32::own_write_plane;
using k32::lds_vpair;

constexpr int N2 = 1024;
constexpr int BLOCK = 32768;                       // points per workgroup and block, both kernels
constexpr int COLS_LDS = BLOCK * 4;                // one component plane of a block
constexpr int PIPE_UNITS = 6;                      // rows kernel: quads of the next block requested from pass 2's last level

// x w for a twiddle the whole wave shares (SGPRs: no inline asm, which would want it in VGPRs)
__device__ __forceinline__ v2f cmul_u(v2f x, v2f w) { return __builtin_elementwise_fma(x.yx, (v2f){-w.y, w.y}, x * w.xx); }

// ---------------------------------------------------------------------------------------------------------------
// contiguous runs of blocks per XCD (workgroups b, b + 8, ... share an XCD's L2; neighbouring blocks share lines)
struct Run {
    int64_t blk, end, stride;
};
__device__ __forceinline__ Run xcd_run(int64_t nblk)
{
    const int64_t per_xcd = (nblk + 7) / 8;
    const int64_t first = (int64_t)(blockIdx.x & 7) * per_xcd;
    Run r;
    r.end = first + per_xcd < nblk ? first + per_xcd : nblk;
    r.stride = gridDim.x >> 3;
    r.blk = first + (blockIdx.x >> 3);
    return r;
}

// ---------------------------------------------------------------------------------------------------------------
// columns: Z[k1][n2] = sum_n1 W_N1^(n1 k1) w[n] x[n]
template <int FMT, int R2> __global__ __launch_bounds__(T, 1) void four_cols_kernel(FourArgs a)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    using S = Sample<FMT>;
    constexpr int N1 = 32 * R2, C = T / R2, SETS = 32 / R2, M = N1 * N2;
    static_assert(C >= 64, "a wave's lanes are 64 neighbouring columns (the mates of the scratch order are lanes c, c + 32)");
    float *lds = reinterpret_cast<float *>(smem);

    Run run = xcd_run(a.rows * R2);                                  // R2 = 1024 / C column groups per stream row
    if (run.blk >= run.end) return;
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = __builtin_amdgcn_readfirstlane(tid / C);         // m in front of the exchange, the k_l set behind it
    const char *iq = reinterpret_cast<const char *>(a.iq);

    v2f v[32];
    v4f w4[8];
    // a block's samples (legs [L0, L1)) and its window coefficients in this kernel's order (fourstep_tables)
    struct Src {
        __amdgpu_buffer_rsrc_t rs, rw;
        int vo;
    };
    auto source = [&](int64_t blk, bool valid) {
        const int64_t s = blk / R2;
        const int cg = (int)(blk % R2);
        Src src;
        src.rs = make_rsrc(iq + (a.first_row + s) * (int64_t)a.hop * S::BYTES, valid ? (unsigned)M * S::BYTES : 0u);
        src.rw = make_rsrc(a.window_a + (size_t)cg * 8 * T * 4, valid ? 8 * T * 16 : 0);
        src.vo = (N2 * grp + C * cg + (tid % C)) * S::BYTES;
        return src;
    };
    auto load_legs = [&](const Src &src, int vo, auto lo_c, auto hi_c) {
        constexpr int L0 = decltype(lo_c)::value, L1 = decltype(hi_c)::value;
#pragma unroll
        for (int l = L0; l < L1; ++l) v[l] = S::load(src.rs, vo, l * (N2 * R2) * S::BYTES);
    };
    auto load_window = [&](const Src &src, int to) {
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const u32x4 t = __builtin_amdgcn_raw_buffer_load_b128(src.rw, to, q * T * 16, 0);
            w4[q] = (v4f){__uint_as_float(t.x), __uint_as_float(t.y), __uint_as_float(t.z), __uint_as_float(t.w)};
        }
    };
    using c0 = std::integral_constant<int, 0>;
    using c16 = std::integral_constant<int, 16>;
    using c32 = std::integral_constant<int, 32>;
    {
        const Src src = source(run.blk, true);
        load_legs(src, src.vo, c0{}, c32{});
        load_window(src, tid * 16);
    }

    for (;;) {
        const int64_t blk = run.blk, next = blk + run.stride;
        const bool has_next = next < run.end;
        // ---- window (src/FFTBackend.cpp:229-232; Q += gain: :78-79) and the radix-32 over l
        {
            const v2f gain2 = (v2f){0.0f, a.gain};
            if (a.gain != 0.0f) {
#pragma unroll
                for (int l = 0; l < 32; ++l) v[l] = v[l] + gain2;
            }
#pragma unroll
            for (int l = 0; l < 32; ++l) {
                const v4f c4 = w4[l >> 2];
                const float c = (l & 3) == 0 ? c4.x : (l & 3) == 1 ? c4.y : (l & 3) == 2 ? c4.z : c4.w;
                v[l] = v[l] * (v2f){c, c};
            }
        }
        dit<32>(v);                                              // result k_l at v[bitrev32(k_l)]
        // ---- exchange: plane[k_l][tid] <- this thread's k_l; thread (set g, column) reads k_l = SETS g + h, every m
        v2f u[32];                                               // u[R2 h + m]
        wg_sync();                                               // the last block's reads of the plane are done
#pragma unroll
        for (int k = 0; k < 32; ++k) lds[k * T + tid] = v[bitrev<32>(k)].x;
        wg_sync();
#pragma unroll
        for (int h = 0; h < SETS; ++h)
#pragma unroll
            for (int m = 0; m < R2; ++m) u[R2 * h + m].x = lds[(SETS * grp + h) * T + m * C + (tid % C)];
        wg_sync();
#pragma unroll
        for (int k = 0; k < 32; ++k) lds[k * T + tid] = v[bitrev<32>(k)].y;
        wg_sync();
#pragma unroll
        for (int h = 0; h < SETS; ++h)
#pragma unroll
            for (int m = 0; m < R2; ++m) u[R2 * h + m].y = lds[(SETS * grp + h) * T + m * C + (tid % C)];
        // The next block's samples and coefficients: v and w4 are free from here on, but 128 VGPRs do not hold them
        // next to u -- the first half of the legs now, the second when half of u has left, the window at the end.
        const Src nsrc = source(has_next ? next : blk, has_next);
        load_legs(nsrc, nsrc.vo, c0{}, c16{});
        // ---- W_N1^(m k_l) (one table row per k_l, the same for the whole wave: scalar loads), radix-R2 over m, out
        const int64_t s = blk / R2;
        const int cg = (int)(blk % R2);
        // row k1 of stream row s: 2048 floats at ((s N1 + k1) 2048); this wave's 64 columns are a = lane & 31 of
        // b = 2 i + p, i = (C / 64) cg + (wave's half of the group), p = lane >> 5
        const int i_quad = (C / 64) * cg + (C == 128 ? (wave & 1) : 0);
        const __amdgpu_buffer_rsrc_t rz = make_rsrc(a.z + (size_t)s * N1 * 2048, (unsigned)N1 * 2048u * 4u);
        const int lane = tid & 63;
        // lanes < 32 store the pair of imaginary parts (floats 2, 3 of the quad), lanes >= 32 the real parts
        const int zo = ((i_quad * 32 + (lane & 31)) * 4 + (lane < 32 ? 2 : 0)) * 4;
        float last = 0.0f;
#pragma unroll
        for (int h = 0; h < SETS; ++h) {
            const int kl = SETS * grp + h;
            const float2 *ta = a.tw_a + kl * R2;
#pragma unroll
            for (int m = 1; m < R2; ++m) {
                const float2 t = ta[m];
                u[R2 * h + m] = cmul_u(u[R2 * h + m], (v2f){t.x, t.y});
            }
            dit<R2>(&u[R2 * h]);                                 // result k_m at position bitrev_R2(k_m)
#pragma unroll
            for (int km = 0; km < R2; ++km) {
                const v2f z = u[R2 * h + bitrev<R2>(km)];
                // lanes c < 32 and c + 32 hold the mates: (v0, v1) = (im, re) -> lanes < 32 (im, im'), lanes >= 32 (re, re')
                const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(z.y), __float_as_uint(z.x), false, false);
                __builtin_amdgcn_raw_buffer_store_b64((u32x2){sw[0], sw[1]}, rz, zo, (kl + 32 * km) * 8192, RO_FOUR_Z_ST_AUX);
                last = __uint_as_float(sw[1]);
            }
            if (h == SETS / 2 - 1) load_legs(nsrc, after(nsrc.vo, last), c16{}, c32{});
        }
        load_window(nsrc, after(tid * 16, last));
        if (!has_next) break;
        run.blk = next;
    }
}

// ---------------------------------------------------------------------------------------------------------------
// rows: X[k1 + N1 k2] = sum_n2 W_1024^(n2 k2) W_M^(n2 k1) Z[k1][n2], |X| to column (k + M/2) mod M of the row
__global__ __launch_bounds__(T, 1) void four_rows_kernel(FourArgs a)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    using namespace planar;
    const float *lds = reinterpret_cast<const float *>(smem);
    const int G = a.n1 >> 5;                                         // blocks (of 32 rows k1) per stream row
    Run run = xcd_run(a.rows * G);
    if (run.blk >= run.end) return;
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int lane;
    {
        int lt = tid;
        asm volatile("" : "+v"(lt));
        lane = lt & 63;
    }
    // pass 1: thread (row rho = 2 wave + kb, a): lane (a >> 1) + 16 (a & 1) + 32 kb; quads i < 16 of its row
    const int zo = ((2 * wave + (lane >> 5)) * 2048 + (2 * (lane & 15) + ((lane >> 4) & 1)) * 4) * 4;
    v2f R[16], I[16];
    auto z_rsrc = [&](int64_t blk, bool valid) { return make_rsrc(a.z + (size_t)blk * (32 * 2048), valid ? 32u * 2048u * 4u : 0u); };
    auto load_quad = [&](const __amdgpu_buffer_rsrc_t &rs, int off, int i) {
        const u32x4 t = __builtin_amdgcn_raw_buffer_load_b128(rs, off, i * 512, RO_FOUR_Z_LD_AUX);
        R[i] = (v2f){__uint_as_float(t.x), __uint_as_float(t.y)};
        I[i] = (v2f){__uint_as_float(t.z), __uint_as_float(t.w)};
    };
    {
        const __amdgpu_buffer_rsrc_t rs = z_rsrc(run.blk, true);
#pragma unroll
        for (int i = 0; i < 16; ++i) load_quad(rs, zo, i);
    }
    // pass 2: thread (row rho = 2 wave + kbp, k_r = k1p)
    const int k1p = ((lane >> 1) + 4 * (wave >> 1)) & 31, kbp = lane & 1;
    const __amdgpu_buffer_rsrc_t rs_twr = make_rsrc(a.tw_r, 32 * 8 * 8);

    // the image of the block before this one: chunk q = segments k_c = 4 q + (tid >> 8), bins beta = 4 m .. 4 m + 3 of it
    // (m = tid & 255): rows rho = 4 (m & 7) + i of k_r = m >> 3 -- four neighbouring columns
    // 32 g + rho + N1 ((k_r + 32 k_c + 512) & 1023) of the fft-shifted row (src/WaterfallBackend.cpp:492-505)
    int rb_base = RQ * (tid >> 8) + 128 * (tid & 7) + 2 * ((((tid & 255) >> 3) - 4 * (tid & 7)) & 31);
    const int out_vo = (4 * (tid & 7) + a.n1 * ((tid & 255) >> 3) + a.n1 * 32 * (tid >> 8)) * 4;
    const float *prev_out = a.rows_out;
    unsigned prev_bytes = 0;
    auto store_chunk = [&](int q, const __amdgpu_buffer_rsrc_t &rs) {
        int rb = rb_base;
        asm volatile("" : "+v"(rb));
        lds_vpair *p = (lds_vpair *)(lds + rb + 4 * RQ * q);
        const v2f x01 = p[0], x23 = p[32];
        buf_store_f4(x01.x, x01.y, x23.x, x23.y, rs, out_vo, a.n1 * 128 * ((q + 4) & 7) * 4);
    };
    const unsigned mc = (unsigned)wave * 256u, md = mc + (unsigned)HB;

    for (;;) {
        const int64_t blk = run.blk, next = blk + run.stride;
        const bool has_next = next < run.end;
        const int g = (int)(blk % G);
        const int64_t s = blk / G;
        const __amdgpu_buffer_rsrc_t rs_prev = make_rsrc(prev_out, prev_bytes);
        // stage twiddles of the wave's two rows k1 = 32 g + 2 wave (+ 1): [k1][16] = five powers 2^j of W_M^(32 k1) at
        // 0..4 and of W_M^(k1) at 8..12 -- uniform addresses: the scalar cache
        const float2 *tb = a.tw_b + (size_t)(32 * g + 2 * wave) * 16;
        v2f tw1[5], tw2[5];
        {
            const bool odd = lane >= 32;
#pragma unroll
            for (int j = 0; j < 5; ++j) {
                const float2 e = tb[j], o = tb[16 + j];
                tw1[j] = (v2f){odd ? o.x : e.x, odd ? o.y : e.y};
            }
        }
        // ---- pass 1 (over b; mates b, b + 1: planar<0>), the previous block's image leaving between its levels
        level<0, 0>(R, I, tw1[4]);
        store_chunk(0, rs_prev);
        store_chunk(1, rs_prev);
        level<0, 1>(R, I, tw1[3]);
        store_chunk(2, rs_prev);
        store_chunk(3, rs_prev);
        level<0, 2>(R, I, tw1[2]);
        store_chunk(4, rs_prev);
        store_chunk(5, rs_prev);
        level<0, 3>(R, I, tw1[1]);
        store_chunk(6, rs_prev);
        store_chunk(7, rs_prev);
        {
            // pass 2's twiddles: W_1024^(k_r) (a vector load: k_r is this thread's) times W_M^(k1) (the scalar cache)
            v2f tr[6];
#pragma unroll
            for (int q = 0; q < 3; ++q) {
                const u32x4 t = __builtin_amdgcn_raw_buffer_load_b128(rs_twr, k1p * 64, q * 16, 0);
                tr[2 * q] = (v2f){__uint_as_float(t.x), __uint_as_float(t.y)};
                tr[2 * q + 1] = (v2f){__uint_as_float(t.z), __uint_as_float(t.w)};
            }
#pragma unroll
            for (int j = 0; j < 5; ++j) {
                const float2 e = tb[8 + j], o = tb[24 + j];
                tw2[j] = cmul(tr[j], (v2f){kbp ? o.x : e.x, kbp ? o.y : e.y});
            }
        }
        wg_sync();                                               // (a) every wave has read its part of the old image
        last0(R, I, tw1[0], [&](auto jc) {
            constexpr int j = decltype(jc)::value, q = bitrev<32>(2 * j);
            own_write4<q, q + 1, q + 16, q + 17>(mc, md, R[j].x, R[8 + j].x, R[j].y, R[8 + j].y);
        });
        // ---- exchange 2, inside the wave (see ro_stft32k.hip)
        v2f R2[16], I2[16];
        {
            lds_vpair *g2 = (lds_vpair *)(lds + RQ * k1p + 64 * wave + 32 * kbp);
            asm volatile("" ::: "memory");
#pragma unroll
            for (int i = 0; i < 16; ++i) R2[i] = g2[(i >> 1) + 8 * (i & 1)];
            asm volatile("" ::: "memory");
            own_write_plane(mc, md, [&](int k1) {
                constexpr int MB = 0;
                const int p = bitrev<32>(k1);
                return hf<MB>(p) ? I[pr<MB>(p)].y : I[pr<MB>(p)].x;
            });
#pragma unroll
            for (int i = 0; i < 16; ++i) I2[i] = g2[(i >> 1) + 8 * (i & 1)];
        }
        // ---- pass 2 (over a; mates a, a + 2: planar<1>), magnitudes into the image, the next block's quads behind them
        head<1>(R2, I2, tw2[4], tw2[3], tw2[2], tw2[1]);
        {
            const __amdgpu_buffer_rsrc_t rs_next = z_rsrc(has_next ? next : blk, has_next);
            v2f pma = {0.f, 0.f}, pmb = {0.f, 0.f};
            last1(R2, I2, tw2[0], [&](auto uc) {
                constexpr int u = decltype(uc)::value;
                auto mag = [](v2f re, v2f im) {                  // src/WaterfallBackend.cpp:497-503
                    const v2f sq = __builtin_elementwise_fma(im, im, re * re);
                    return (v2f){__builtin_amdgcn_sqrtf(sq.x), __builtin_amdgcn_sqrtf(sq.y)};
                };
                const v2f m_a = mag(R2[2 * u], I2[2 * u]), m_b = mag(R2[2 * u + 1], I2[2 * u + 1]);
                if constexpr (u > 0) {
                    constexpr int r = bitrev<8>(u > 0 ? u - 1 : 0);
                    own_write4<r, r + 8, r + 16, r + 24>(mc, md, pma.x, pma.y, pmb.x, pmb.y);
                }
                pma = m_a;
                pmb = m_b;
                if constexpr (u < PIPE_UNITS) {
                    const int pj = after(zo, m_b.y);
                    load_quad(rs_next, pj, 2 * u);
                    load_quad(rs_next, pj, 2 * u + 1);
                }
            });
            asm volatile("s_nop 7\n\ts_nop 7" ::: "memory");      // the last square roots (transcendental pipe)
            {
                constexpr int r = bitrev<8>(7);
                own_write4<r, r + 8, r + 16, r + 24>(mc, md, pma.x, pma.y, pmb.x, pmb.y);
            }
#pragma unroll
            for (int i = 2 * PIPE_UNITS; i < 16; ++i) load_quad(rs_next, zo, i);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the add-TID image writes (hipcc does not count them)
            wg_sync();                                           // (e) the image of this block is complete
        }
        prev_out = a.rows_out + s * a.row_stride + 32 * g;
        prev_bytes = (unsigned)(a.n1 * N2 - 32 * g) * 4u;
        if (!has_next) break;
        run.blk = next;
    }
    {
        const __amdgpu_buffer_rsrc_t rs_last = make_rsrc(prev_out, prev_bytes);
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            store_chunk(q, rs_last);
            if (q & 1) asm volatile("" ::: "memory");
        }
    }
}

struct DevicePlan {
    bool ready = false;
    int  cus = 0;
};

template <typename K> static hipError_t prepare(K kernel, int lds_bytes, int &cus)
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
        if ((e = hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes)) != hipSuccess)
            return e;
        if ((e = hipDeviceGetAttribute(&d.cus, hipDeviceAttributeMultiprocessorCount, dev)) != hipSuccess) return e;
        d.ready = true;
    }
    cus = d.cus;
    return hipSuccess;
}

static unsigned grid_for(int cus, int64_t nblk)
{
    const int64_t per_xcd = (nblk + 7) / 8;
    int64_t slots = cus / 8;
    if (slots < 1) slots = 1;
    if (slots > per_xcd) slots = per_xcd;
    return (unsigned)(slots * 8);
}

template <int FMT, int R2> static hipError_t launch_cols(const FourArgs &a, hipStream_t s)
{
    int cus = 0;
    hipError_t e = prepare(&four_cols_kernel<FMT, R2>, COLS_LDS, cus);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL((four_cols_kernel<FMT, R2>), dim3(grid_for(cus, a.rows * R2)), dim3(T), COLS_LDS, s, a);
    return hipGetLastError();
}

}  // namespace four

bool fourstep_supported(int bins) { return bins == 262144 || bins == 524288; }

// tables of a size (host; the C ABI uploads them): see FourArgs
void fourstep_tables(int bins, const float *window, std::vector<float> &window_a, std::vector<float2> &tw_a,
                     std::vector<float2> &tw_b, std::vector<float2> &tw_r)
{
    const int n1 = bins / 1024, r2 = n1 / 32, c = 1024 / r2;
    const long double tau = -2.0L * 3.14159265358979323846264338327950288L;
    // window: thread t = m c + col of column group cg reads quad q = legs l = 4 q .. 4 q + 3 at ((cg 8 + q) 1024 + t) 4:
    // w[1024 (m + r2 l) + c cg + col]
    window_a.assign((size_t)bins, 0.0f);
    for (int cg = 0; cg < r2; ++cg)
        for (int q = 0; q < 8; ++q)
            for (int t = 0; t < 1024; ++t)
                for (int e = 0; e < 4; ++e) {
                    const int m = t / c, col = t % c, l = 4 * q + e;
                    window_a[(((size_t)cg * 8 + q) * 1024 + t) * 4 + e] = window[(size_t)1024 * (m + r2 * l) + c * cg + col];
                }
    tw_a.assign((size_t)32 * r2, float2{1.0f, 0.0f});
    for (int kl = 0; kl < 32; ++kl)
        for (int m = 0; m < r2; ++m) {
            const long double ph = tau * (long double)(kl * m) / (long double)n1;
            tw_a[(size_t)kl * r2 + m] = float2{(float)cosl(ph), (float)sinl(ph)};
        }
    tw_b.assign((size_t)n1 * 16, float2{1.0f, 0.0f});
    for (int k1 = 0; k1 < n1; ++k1)
        for (int j = 0; j < 5; ++j) {
            const long double p1 = tau * (long double)((int64_t)32 * k1 * (1 << j) % bins) / (long double)bins;
            const long double p2 = tau * (long double)((int64_t)k1 * (1 << j)) / (long double)bins;
            tw_b[(size_t)k1 * 16 + j] = float2{(float)cosl(p1), (float)sinl(p1)};
            tw_b[(size_t)k1 * 16 + 8 + j] = float2{(float)cosl(p2), (float)sinl(p2)};
        }
    tw_r.assign((size_t)32 * 8, float2{1.0f, 0.0f});
    for (int kr = 0; kr < 32; ++kr)
        for (int j = 0; j < 5; ++j) {
            const long double ph = tau * (long double)(kr * (1 << j)) / 1024.0L;
            tw_r[(size_t)kr * 8 + j] = float2{(float)cosl(ph), (float)sinl(ph)};
        }
}

hipError_t launch_fourstep(int fmt, const FourArgs &a, hipStream_t s)
{
    using namespace four;
    if (a.rows <= 0) return hipSuccess;
    if (!fourstep_supported(a.n1 * 1024) || !a.z || !a.window_a || !a.tw_a || !a.tw_b || !a.tw_r) return hipErrorInvalidValue;
    hipError_t e;
    if (a.n1 == 512)
        e = fmt == RO_FMT_F32 ? launch_cols<RO_FMT_F32, 16>(a, s) : fmt == RO_FMT_I16 ? launch_cols<RO_FMT_I16, 16>(a, s) : hipErrorInvalidValue;
    else
        e = fmt == RO_FMT_F32 ? launch_cols<RO_FMT_F32, 8>(a, s) : fmt == RO_FMT_I16 ? launch_cols<RO_FMT_I16, 8>(a, s) : hipErrorInvalidValue;
    if (e != hipSuccess) return e;
    int cus = 0;
    if ((e = prepare(&four_rows_kernel, k32::LDS_BYTES, cus)) != hipSuccess) return e;
    hipLaunchKernelGGL(four_rows_kernel, dim3(grid_for(cus, a.rows * (a.n1 / 32))), dim3(T), k32::LDS_BYTES, s, a);
    return hipGetLastError();
}

}  // namespace ro
