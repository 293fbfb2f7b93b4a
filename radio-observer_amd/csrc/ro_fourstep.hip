// ro_fourstep.hip -- the large transforms (bins = N1 x 1024: 262144 and Ionozor's 524288, Ionozor.json:27) as a
// four-step FFT: two kernels, ONE trip through HBM scratch, every global access a run of whole cache lines.
//   replaces src/FFTBackend.cpp:229-236 (window multiply + fftw_execute) and src/WaterfallBackend.cpp:485-505
//   (magnitude + fft-shift) of the reference for those sizes.
//
//   n = 1024 n1 + n2          k = k1 + N1 k2                                   (n1, k1 < N1;  n2, k2 < 1024)
//   X[k1 + N1 k2] = sum_n2 W_1024^(n2 k2) { W_M^(n2 k1) sum_n1 W_N1^(n1 k1) w[n] x[n] }
//
// Both kernels run as ONE workgroup of 1024 threads per CU, 32768 points per block, 32 per thread -- the row kernel's
// shape.  (RO_FOUR_WAVES=8 builds them as workgroups of 512 threads, two to a CU, each with half the registers and
// half the LDS, so that one can transform while the other's block is on its way: measured slower, 414 / 337 us per 256
// rows against 397 / 269 -- the kernels move bytes at the speed of a copy either way, and the half-sized blocks of
// the row kernel write half cache lines.  profiles/r04_fourstep.txt.)
//
// four_cols_kernel  (the inner sum: N1-point transforms down the columns n2).  A workgroup takes C = 32768 / N1
//   columns of one stream row (whole pairs of the scratch order, below).  n1 = m + R2 l
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
//   shuffling.  The column kernel gets there with one lane swap per point and 16-byte stores: its waves hold the two
//   mates of a pair in lanes l, l + 32 (C >= 64: v_permlane32_swap) or l, l + 16 (C = 32: v_permlane16_swap).
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

// The ONE diagnostic switch of this file: a -DRO_DIAG=1 build (tools/ab_build.sh) may set RO_FOUR_WAVES=8 (see above).
#if !defined(RO_DIAG) || !defined(RO_FOUR_WAVES)
#undef RO_FOUR_WAVES
#define RO_FOUR_WAVES 16
#endif

namespace ro {
namespace four {

using k32::lds_vpair;

// Cache policy (gfx950 aux bits: 1 = sc0, 2 = nt, 16 = sc1).  Everything that is touched once carries the nt hint --
// the stores and the loads of Z, the row stores, and the legs of a stream row that no later row reads again (the first
// 32 hop / M of them) -- so that an XCD's L2 keeps what IS read again: the legs the next stream row shares and the
// window table.  Column kernel at Ionozor's shape: 5.2 -> 3.5 MiB fetched per stream row (FETCH_SIZE), 402 -> 357 us per
// 256 rows; nt loads of Z: 172 -> 142 us per 128 rows in the row kernel (profiles/r04_fourstep.txt).
constexpr int Z_ST_AUX = 2, Z_LD_AUX = 2;

constexpr int N2 = 1024;
constexpr int WAVES = RO_FOUR_WAVES, T = 64 * WAVES;   // one workgroup of 1024 per CU (8: two of 512, see the header)
static_assert(WAVES == 16 || WAVES == 8, "workgroup shapes");
constexpr int BLOCK = 32 * T;                      // points per workgroup and block, both kernels
constexpr int COLS_LDS = BLOCK * 4;                // one component plane of a block
// rows kernel: cell(q, w, l) = RQ q + 64 w + l as in ro_k32_lds.h, one territory per wave: 1026 floats per row (514 for
// eight waves: = 2 mod 64 as well).  The add-TID writes of rows >= 16 take k32::HB into M0 where 4 RQ 31 does not fit
// the 16 bits of the offset field (ro_k32_lds.h).
constexpr int RQ = 64 * WAVES + 2;
constexpr int HBQ = 4 * RQ * 31 > 65535 ? k32::HB : 0;
static_assert(RQ != 1026 || RQ == k32::RQ, "the row kernel's layout");
constexpr int ROWS_LDS = 32 * RQ * 4;
constexpr int BROWS = 2 * WAVES;                   // rows k1 per block
constexpr int ROT = 64 / WAVES;                    // pass 2's lane rotation per pair of waves (4: ro_stft32k.hip)

// rows QA, QB (< 16) and QC, QD (>= 16) of the wave's own territory (M0 = mc = 256 wave, md = mc + HBQ), and a plane
template <int QA, int QB, int QC, int QD>
__device__ __forceinline__ void own_write4(unsigned mc, unsigned md, float sa, float sb, float sc, float sd)
{
    static_assert(QA < 16 && QB < 16 && QC >= 16 && QD >= 16 && QC < 32 && QD < 32, "row algebra");
    constexpr int R = 4 * RQ;
    static_assert(R * 15 <= 65535 && R * 31 - HBQ <= 65535 && R * 16 - HBQ >= 0 && 15 * 256 + HBQ <= 65535, "M0 / offset split");
    addtid_write4<R * QA, R * QB, R * QC - HBQ, R * QD - HBQ>(mc, md, sa, sb, sc, sd);
}
template <typename F> __device__ __forceinline__ void own_write_plane(unsigned mc, unsigned md, F f)
{
    constexpr int R = 4 * RQ, H = HBQ;
    addtid_write8<0 * R, 1 * R, 2 * R, 3 * R, 4 * R, 5 * R, 6 * R, 7 * R>(mc, f(0), f(1), f(2), f(3), f(4), f(5), f(6), f(7));
    addtid_write8<8 * R, 9 * R, 10 * R, 11 * R, 12 * R, 13 * R, 14 * R, 15 * R>(mc, f(8), f(9), f(10), f(11), f(12), f(13),
                                                                                  f(14), f(15));
    addtid_write8<16 * R - H, 17 * R - H, 18 * R - H, 19 * R - H, 20 * R - H, 21 * R - H, 22 * R - H, 23 * R - H>(
        md, f(16), f(17), f(18), f(19), f(20), f(21), f(22), f(23));
    addtid_write8<24 * R - H, 25 * R - H, 26 * R - H, 27 * R - H, 28 * R - H, 29 * R - H, 30 * R - H, 31 * R - H>(
        md, f(24), f(25), f(26), f(27), f(28), f(29), f(30), f(31));
}
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
//
// Block cg of a stream row (1024 / C of them) and thread t = C grp + lam of the workgroup:
//   C >= 64: columns n2 = C cg + lam: a wave holds a = lane & 31 of b, b + 1 (b even): mates in lanes l, l + 32
//   C = 32:  a = 16 (cg & 1) + (lam & 15) of b = 2 (cg >> 1) + (lam >> 4): mates in lanes l, l + 16; a wave-load of one
//            leg is four runs of 128 bytes
// (fourstep_column() is the same map for the host's window table.)
template <int C> __host__ __device__ constexpr int fourstep_column(int cg, int lam)
{
    return C >= 64 ? C * cg + lam : (16 * (cg & 1) + (lam & 15)) + 32 * (2 * (cg >> 1) + (lam >> 4));
}

template <int FMT, int R2, int NT> __global__ __launch_bounds__(T, 4) void four_cols_kernel(FourArgs a)   // 4 waves per SIMD: 128 VGPRs
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    using S = Sample<FMT>;
    constexpr int N1 = 32 * R2, C = T / R2, SETS = 32 / R2, M = N1 * N2, GROUPS = N2 / C;
    static_assert(C == 128 || C == 64 || C == 32, "the lane swaps of the scratch order");
    float *lds = reinterpret_cast<float *>(smem);

    Run run = xcd_run(a.rows * GROUPS);
    if (run.blk >= run.end) return;
    const int tid = threadIdx.x;
    // m in front of the exchange, the k_l set behind it (C >= 64: the wave's; C = 32: the half-wave's)
    const int grp = C >= 64 ? __builtin_amdgcn_readfirstlane(tid / C) : tid / C;
    const int lam = tid % C;
    const char *iq = reinterpret_cast<const char *>(a.iq);

    v2f v[32];
    v4f w4[8];
    // legs whose samples no later stream row contains: leg l is samples [1024 R2 l, 1024 R2 (l + 1)) of the row, the next
    // row starts at hop
    // (a template parameter: the launcher rounds 32 hop / M down to 32, 16, 8 or 0.  As a run-time value, one uniform
    // branch per leg, the kernel took 385 us per 256 rows at Ionozor's shape instead of 357)
    constexpr int nt_legs = NT;
    // a block's samples (legs [L0, L1)) and its window coefficients in this kernel's order (fourstep_tables)
    struct Src {
        __amdgpu_buffer_rsrc_t rs, rw;
        int vo;
    };
    auto source = [&](int64_t blk, bool valid) {
        const int64_t s = blk / GROUPS;
        const int cg = (int)(blk % GROUPS);
        Src src;
        src.rs = make_rsrc(iq + (a.first_row + s) * (int64_t)a.hop * S::BYTES, valid ? (unsigned)M * S::BYTES : 0u);
        src.rw = make_rsrc(a.window_a + (size_t)cg * 8 * T * 4, valid ? 8 * T * 16 : 0);
        src.vo = (N2 * grp + fourstep_column<C>(cg, lam)) * S::BYTES;
        return src;
    };
    auto load_legs = [&](const Src &src, int vo, auto lo_c, auto hi_c) {
        constexpr int L0 = decltype(lo_c)::value, L1 = decltype(hi_c)::value;
#pragma unroll
        for (int l = L0; l < L1; ++l) {
            if (l < nt_legs) v[l] = S::load_nt(src.rs, vo, l * (N2 * R2) * S::BYTES);
            else v[l] = S::load(src.rs, vo, l * (N2 * R2) * S::BYTES);
        }
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
    const __amdgpu_buffer_rsrc_t rs_twa = make_rsrc(a.tw_a, 32 * R2 * 8);

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
        // ---- exchange: plane[k_l][tid] <- this thread's k_l; thread (set grp, column) reads k_l = SETS grp + h, every m
        v2f u[32];                                               // u[R2 h + m]
        wg_sync();                                               // the last block's reads of the plane are done
#pragma unroll
        for (int k = 0; k < 32; ++k) lds[k * T + tid] = v[bitrev<32>(k)].x;
        wg_sync();
#pragma unroll
        for (int h = 0; h < SETS; ++h)
#pragma unroll
            for (int m = 0; m < R2; ++m) u[R2 * h + m].x = lds[(SETS * grp + h) * T + m * C + lam];
        wg_sync();
#pragma unroll
        for (int k = 0; k < 32; ++k) lds[k * T + tid] = v[bitrev<32>(k)].y;
        wg_sync();
#pragma unroll
        for (int h = 0; h < SETS; ++h)
#pragma unroll
            for (int m = 0; m < R2; ++m) u[R2 * h + m].y = lds[(SETS * grp + h) * T + m * C + lam];
        // The next block's samples and coefficients: v and w4 are free from here on, but 128 VGPRs do not hold them
        // next to u -- the first half of the legs now (N1 = 1024, one set of 32 in u: when a quarter of it has left),
        // the second when half of u has left, the window at the end.
        const Src nsrc = source(has_next ? next : blk, has_next);
        if constexpr (SETS > 1) load_legs(nsrc, nsrc.vo, c0{}, c16{});
        // ---- W_N1^(m k_l), radix-R2 over m, out
        const int64_t s = blk / GROUPS;
        const int cg = (int)(blk % GROUPS);
        // row k1 of stream row s: 2048 floats at (s N1 + k1) 2048; quad (i, a) of this lane's column at (32 i + a) 4.
        const __amdgpu_buffer_rsrc_t rz = make_rsrc(a.z + (size_t)s * N1 * 2048, (unsigned)N1 * 2048u * 4u);
        const int n2 = fourstep_column<C>(cg, lam);
        const int mate_p = (n2 >> 5) & 1;
        // Two rows k_m = 2 kp (A) and 2 kp + 1 (B) leave together.  The lane swap of one component with vdst = B, src = A
        // gives the lanes of p = 0 (B, B') and the lanes of p = 1 (A, A'): a lane stores a whole quad, 16 bytes, of row
        // B resp. A (32 rows of 8192 bytes further down).  (C = 32: the half-waves' k_l sets differ -- their part of the
        // row number is in the lane's offset too.)
        const int zo = ((n2 >> 6) * 32 + (n2 & 31)) * 16 + (mate_p ? 0 : 32 * 8192) + (C == 32 ? SETS * grp * 8192 : 0);
        const int zs = C >= 64 ? SETS * grp * 8192 : 0;
        float last = 0.0f;
#pragma unroll
        for (int h = 0; h < SETS; ++h) {
            if constexpr (C >= 64) {
                // one table row per k_l, the same for the whole wave: scalar loads
                const float2 *ta = a.tw_a + (SETS * grp + h) * R2;
#pragma unroll
                for (int m = 1; m < R2; ++m) {
                    const float2 t = ta[m];
                    u[R2 * h + m] = cmul_u(u[R2 * h + m], (v2f){t.x, t.y});
                }
            } else {
#pragma unroll
                for (int m = 1; m < R2; ++m)
                    u[R2 * h + m] = cmul(u[R2 * h + m], buf_load_f2(rs_twa, SETS * grp * R2 * 8, (h * R2 + m) * 8));
            }
            dit<R2>(&u[R2 * h]);                                 // result k_m at position bitrev_R2(k_m)
#pragma unroll
            for (int kp = 0; kp < R2 / 2; ++kp) {
                const v2f zA = u[R2 * h + bitrev<R2>(2 * kp)], zB = u[R2 * h + bitrev<R2>(2 * kp + 1)];
                u32x4 q;
                if constexpr (C >= 64) {
                    const auto tr = __builtin_amdgcn_permlane32_swap(__float_as_uint(zB.x), __float_as_uint(zA.x), false, false);
                    const auto ti = __builtin_amdgcn_permlane32_swap(__float_as_uint(zB.y), __float_as_uint(zA.y), false, false);
                    q = (u32x4){tr[0], tr[1], ti[0], ti[1]};
                } else {
                    const auto tr = __builtin_amdgcn_permlane16_swap(__float_as_uint(zB.x), __float_as_uint(zA.x), false, false);
                    const auto ti = __builtin_amdgcn_permlane16_swap(__float_as_uint(zB.y), __float_as_uint(zA.y), false, false);
                    q = (u32x4){tr[0], tr[1], ti[0], ti[1]};
                }
                buf_store_u4<Z_ST_AUX>(q, rz, zo, zs + (h + 32 * 2 * kp) * 8192);
                last = __uint_as_float(q.w);
                if (SETS == 1 && kp == R2 / 4 - 1) load_legs(nsrc, after(nsrc.vo, last), c0{}, c16{});
            }
            if (SETS > 1 && h == SETS / 2 - 1) load_legs(nsrc, after(nsrc.vo, last), c16{}, c32{});
        }
        if (SETS == 1) load_legs(nsrc, after(nsrc.vo, last), c16{}, c32{});
        load_window(nsrc, after(tid * 16, last));
        if (!has_next) break;
        run.blk = next;
    }
}

// ---------------------------------------------------------------------------------------------------------------
// rows: X[k1 + N1 k2] = sum_n2 W_1024^(n2 k2) W_M^(n2 k1) Z[k1][n2], |X| to column (k + M/2) mod M of the row
//
// ro_stft32k.hip's maps (rho = k1 - BROWS g plays k0's part, k_r is k1's, k_c is k2's):
//   pass 1:      lane (a >> 1) + 16 (a & 1) + 32 kb of wave w is thread (rho = 2 w + kb, a); slots b = 2 i, 2 i + 1 are
//                quad (i, a) of its scratch row
//   exchange 2:  slot k_r of that thread -> cell(k_r, w, lane); pass-2 lane l' of wave w is thread
//                (rho = 2 w + (l' & 1), k_r = ((l' >> 1) + ROT (w >> 1)) & 31) and reads slots a = 4 u + p, 4 u + p + 2
//                from cell(k_r, w, 2 u + 16 p + 32 kb), + 1
//   image:       slot k_c of that thread -> cell(k_c, w, l').  The rotation (4 per pair of waves with sixteen waves, 8
//                with eight) makes the read-back (two ds_read_b64 per lane: territories w, w + 1 = four neighbouring
//                rho) conflict-free.  tools/r4/emu_four.py checks all of it.
__global__ __launch_bounds__(T, 4) void four_rows_kernel(FourArgs a)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    using namespace planar;
    const float *lds = reinterpret_cast<const float *>(smem);
    const int G = a.n1 / BROWS;                                      // blocks per stream row
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
    const int zo = ((2 * wave + (lane >> 5)) * 2048 + (2 * (lane & 15) + ((lane >> 4) & 1)) * 4) * 4;
    v2f R[16], I[16];
    auto z_rsrc = [&](int64_t blk, bool valid) {
        return make_rsrc(a.z + (size_t)blk * (BROWS * 2048), valid ? (unsigned)BROWS * 2048u * 4u : 0u);
    };
    auto load_quad = [&](const __amdgpu_buffer_rsrc_t &rs, int off, int i) {
        const u32x4 t = __builtin_amdgcn_raw_buffer_load_b128(rs, off, i * 512, Z_LD_AUX);
        R[i] = (v2f){__uint_as_float(t.x), __uint_as_float(t.y)};
        I[i] = (v2f){__uint_as_float(t.z), __uint_as_float(t.w)};
    };
    {
        const __amdgpu_buffer_rsrc_t rs = z_rsrc(run.blk, true);
#pragma unroll
        for (int i = 0; i < 16; ++i) load_quad(rs, zo, i);
    }
    const int k1p = ((lane >> 1) + ROT * (wave >> 1)) & 31, kbp = lane & 1;
    const __amdgpu_buffer_rsrc_t rs_twr = make_rsrc(a.tw_r, 32 * 8 * 8);

    // the image of the block before this one: chunk q = segments k_c = 4 q + t_hi (t_hi = tid / (T / 4)); m = tid mod
    // (T / 4): rows rho = 4 j + i (j = m mod (WAVES / 2)) of k_r = m / (WAVES / 2) -- four neighbouring columns
    // BROWS g + rho + N1 ((k_r + 32 k_c + 512) & 1023) of the fft-shifted row (src/WaterfallBackend.cpp:492-505)
    const int t_hi = tid / (T / 4), rj = tid % (WAVES / 2), r_kr = (tid % (T / 4)) / (WAVES / 2);
    int rb_base = RQ * t_hi + 128 * rj + 2 * ((r_kr - ROT * rj) & 31);
    const int out_vo = (4 * rj + a.n1 * r_kr + a.n1 * 32 * t_hi) * 4;
    const float *prev_out = a.rows_out;
    unsigned prev_bytes = 0;
    auto store_chunk = [&](int q, const __amdgpu_buffer_rsrc_t &rs) {
        int rb = rb_base;
        asm volatile("" : "+v"(rb));
        lds_vpair *p = (lds_vpair *)(lds + rb + 4 * RQ * q);
        const v2f x01 = p[0], x23 = p[32];
        buf_store_f4(x01.x, x01.y, x23.x, x23.y, rs, out_vo, a.n1 * 128 * ((q + 4) & 7) * 4);
    };
    const unsigned mc = (unsigned)wave * 256u, md = mc + (unsigned)HBQ;

    for (;;) {
        const int64_t blk = run.blk, next = blk + run.stride;
        const bool has_next = next < run.end;
        const int g = (int)(blk % G);
        const int64_t s = blk / G;
        const __amdgpu_buffer_rsrc_t rs_prev = make_rsrc(prev_out, prev_bytes);
        // stage twiddles of the wave's two rows k1 = 16 g + 2 wave (+ 1): [k1][16] = five powers 2^j of W_M^(32 k1) at
        // 0..4 and of W_M^(k1) at 8..12 -- uniform addresses: the scalar cache
        const float2 *tb = a.tw_b + (size_t)(BROWS * g + 2 * wave) * 16;
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
        prev_out = a.rows_out + s * a.row_stride + BROWS * g;
        prev_bytes = (unsigned)(a.n1 * N2 - BROWS * g) * 4u;
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

// One table PER KERNEL (the kernel is a template argument: every four_cols_kernel variant and four_rows_kernel have the
// same function type, a table keyed by the type was shared between them and only the first kernel launched on a device
// got its dynamic-LDS attribute).
template <auto KERNEL> static hipError_t prepare(int lds_bytes, int &cus)
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
        if ((e = hipFuncSetAttribute(reinterpret_cast<const void *>(KERNEL), hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes)) != hipSuccess)
            return e;
        if ((e = hipDeviceGetAttribute(&d.cus, hipDeviceAttributeMultiprocessorCount, dev)) != hipSuccess) return e;
        d.ready = true;
    }
    cus = d.cus;
    return hipSuccess;
}

static unsigned grid_for(int cus, int spare_cus, int64_t nblk)
{
    const int64_t per_xcd = (nblk + 7) / 8;
    int64_t slots = (16 / WAVES) * (cus / 8 - (spare_cus > 0 ? spare_cus : 0));     // workgroups per XCD
    if (slots < 1) slots = 1;
    if (slots > per_xcd) slots = per_xcd;
    return (unsigned)(slots * 8);
}

template <int FMT, int R2, int NT> static hipError_t launch_cols_nt(const FourArgs &a, hipStream_t s)
{
    int cus = 0;
    hipError_t e = prepare<&four_cols_kernel<FMT, R2, NT>>(COLS_LDS, cus);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL((four_cols_kernel<FMT, R2, NT>), dim3(grid_for(cus, a.spare_cus, a.rows * (N2 / (T / R2)))), dim3(T), COLS_LDS, s, a);
    return hipGetLastError();
}
template <int FMT, int R2> static hipError_t launch_cols(const FourArgs &a, hipStream_t s)
{
    // legs of a stream row that no later row contains (see the kernel): float32 samples only -- the int16 kernel is
    // built without the hint
    const int nt = FMT == RO_FMT_F32 ? a.hop / (N2 * R2) : 0;
    if constexpr (FMT == RO_FMT_F32) {
        if (nt >= 32) return launch_cols_nt<FMT, R2, 32>(a, s);
        if (nt >= 16) return launch_cols_nt<FMT, R2, 16>(a, s);
        if (nt >= 8) return launch_cols_nt<FMT, R2, 8>(a, s);
    }
    return launch_cols_nt<FMT, R2, 0>(a, s);
}

}  // namespace four

bool fourstep_supported(int bins) { return bins == 262144 || bins == 524288 || (bins == 1048576 && four::WAVES == 16); }

// tables of a size (host; the C ABI uploads them): see FourArgs
void fourstep_tables(int bins, const float *window, std::vector<float> &window_a, std::vector<float2> &tw_a,
                     std::vector<float2> &tw_b, std::vector<float2> &tw_r)
{
    const int n1 = bins / 1024, r2 = n1 / 32, c = four::T / r2, groups = 1024 / c;
    const long double tau = -2.0L * 3.14159265358979323846264338327950288L;
    // window: thread t = m c + lam of block cg reads quad q = legs l = 4 q .. 4 q + 3 at ((cg 8 + q) T + t) 4:
    // w[1024 (m + r2 l) + column(cg, lam)]
    window_a.assign((size_t)bins, 0.0f);
    for (int cg = 0; cg < groups; ++cg)
        for (int q = 0; q < 8; ++q)
            for (int t = 0; t < four::T; ++t)
                for (int e = 0; e < 4; ++e) {
                    const int m = t / c, lam = t % c, l = 4 * q + e;
                    const int col = c >= 64 ? c * cg + lam : four::fourstep_column<32>(cg, lam);
                    window_a[(((size_t)cg * 8 + q) * four::T + t) * 4 + e] = window[(size_t)1024 * (m + r2 * l) + col];
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
    if (fmt != RO_FMT_F32 && fmt != RO_FMT_I16) return hipErrorInvalidValue;
    const bool f = fmt == RO_FMT_F32;
    if (a.n1 == 256) e = f ? launch_cols<RO_FMT_F32, 8>(a, s) : launch_cols<RO_FMT_I16, 8>(a, s);
    else if (a.n1 == 512) e = f ? launch_cols<RO_FMT_F32, 16>(a, s) : launch_cols<RO_FMT_I16, 16>(a, s);
    else if constexpr (WAVES == 16) e = f ? launch_cols<RO_FMT_F32, 32>(a, s) : launch_cols<RO_FMT_I16, 32>(a, s);
    else return hipErrorInvalidValue;
    if (e != hipSuccess) return e;
    int cus = 0;
    if ((e = prepare<&four_rows_kernel>(ROWS_LDS, cus)) != hipSuccess) return e;
    hipLaunchKernelGGL(four_rows_kernel, dim3(grid_for(cus, a.spare_cus, a.rows * (a.n1 / BROWS))), dim3(T), ROWS_LDS, s, a);
    return hipGetLastError();
}

}  // namespace ro
