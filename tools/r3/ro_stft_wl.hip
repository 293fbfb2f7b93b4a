// ro_stft_wl.hip -- ROUND-3 EXPERIMENT, not part of the product build (tools/r3/ab_wl_build.sh builds a variant with it;
// profiles/r03_ab_wl.txt has the result: every GPU test green, no faster than stft_kernel's generic loop at 16384, 15 %
// slower at 8192).  N = 16384 and N = 8192 magnitude rows with stft32k_kernel's structure (ro_stft32k.hip): window ->
// FFT -> |X| -> fft-shift -> float32 row, BolidRecorder's per-row scan and the band tile cut from the row in LDS.
//   replaces src/FFTBackend.cpp:229-236, src/WaterfallBackend.cpp:485-505, src/BolidRecorder.cpp:121-132, :313-347.
//
// N = 1024 S (S = 16, 8), one workgroup of T = 32 S threads per row, 32 points per thread, 128 KiB / 64 KiB of LDS per
// CU for two / four workgroups:
//   n = n1 + T n0,  n1 = a + S b  (a < S, b < 32)        k = k0 + 32 k1 + 1024 k2  (k0, k1 < 32, k2 < S)
//   pass 0 over n0 -> k0   radix 32, thread = column n1
//   pass 1 over b  -> k1   radix 32, thread = (k0, a)
//   pass 2 over a  -> k2   radix S, 32 / S butterflies per thread: (k0, k1 = i + S m), m < 32 / S
// A wave owns G = 64 / S values of k0 and ALL of their a after exchange 1, so exchange 2 is a transposition inside
// each group of S lanes of the wave: no workgroup barrier between the last read of exchange 1 and the complete image.
// The same LDS layout as the N = 32768 kernel, cell(q, w, l) = (64 W + 1) q + 64 w + l with W = T / 64 waves; every
// write is ds_write_addtid_b32:
//   exchange 1:  slot k0 of pass-0 wave w', lane l   ->  cell(w' + W (k0 mod G), k0 / G, l)
//                pass-1 lane (a >> 1) + (S/2) j + 32 (a & 1) of wave w is thread (k0 = G w + j, a); it reads slot b
//                from cell(b / G + W j, w, (a >> 1) + (S/2) (b mod G) + 32 (a & 1))
//   exchange 2:  slot k1 of that thread -> cell(pi(k1), w, lane); pi swaps two bits of k1 so that the column reads of
//                the G groups of a half-wave walk 32 different banks
//                pass-2 lane (il, jl) of wave w is k0 = G w + jl, i = (il + 2 w) mod S and reads slot a of butterfly m
//                from cell(pi(i + S m), w, (a >> 1) + (S/2) jl + 32 (a & 1))
//   image:       bin k0 + 32 (i + S m) + 1024 k2 -> cell(m S + k2, w, lane)
// tools/r3/emu_wl.py restates these maps for S = 32, 16, 8 with numpy and checks them against numpy's FFT and for
// bank conflicts (exchanges: none; the 16-byte-per-lane read-back of the image: two-way, the minimum these maps allow).
// the butterflies' scheduling leash as a scheduling barrier, not an empty asm statement (see tie() in ro_fft_device.h)
#ifndef RO_TIE_SCHED
#define RO_TIE_SCHED 1
#endif
#include "ro_kernels.h"
#include "ro_fft_device.h"
#include "ro_device_util.h"

#include <mutex>

namespace ro {
namespace wl {

template <int S_> struct Shape {
    static constexpr int S = S_, N = 1024 * S, T = 32 * S, W = T / 64, G = 64 / S, H = S / 2, NB2 = 32 / S;
    static constexpr int RQ = 64 * W + 1;
    static constexpr int IMAGE_BYTES = 32 * RQ * 4;
    static constexpr int LDS_BYTES = IMAGE_BYTES + 1024;             // + the fused scan's histogram
    static constexpr int LG = S == 16 ? 4 : 3;                       // log2 S
    // twiddle tables of the 32 . 32 . S plan (ro_kernels.hip, Plan<N, T, 32, 32, S>): generic float2 table and packed units
    static constexpr int TW2 = 31 * 32, TW_TOTAL = TW2 + (S - 1) * 1024;
    static constexpr int PK2 = 3 * 32, PK_TOTAL = PK2 + (S == 16 ? 3 * 1024 : 0);
    static_assert(S == 16 || S == 8, "N = 16384 or 8192");
    // M0 (256 w or 4 RQ w') and the offset field (4 RQ row, or 4 (RQ W j + 64 w)) hold 16 bits each: one M0 value does
    static_assert(4 * RQ * 31 < 65536 && 4 * RQ * (W - 1) < 65536 && 4 * (RQ * W * (G - 1) + 64 * (W - 1)) < 65536,
                  "every cell within reach of M0 + offset");
    // row of slot k1 in exchange 2
    static constexpr int pi(int k1)
    {
        constexpr int lo = S == 16 ? 3 : 2;                          // swap bits (lo, lo + 1)
        const int b0 = (k1 >> lo) & 1, b1 = (k1 >> (lo + 1)) & 1;
        return (k1 & ~(3 << lo)) | (b0 << (lo + 1)) | (b1 << lo);
    }
    // pass-2 lane -> (il, jl) and back
    static __device__ __forceinline__ int lane_il(int lane) { return lane & (S - 1); }
    static __device__ __forceinline__ int lane_jl(int lane)
    {
        if constexpr (S == 16) return lane >> 4;
        else return ((lane >> 3) & 1) | (((lane >> 5) & 1) << 1) | (((lane >> 4) & 1) << 2);
    }
    static __device__ __forceinline__ int lane_of(int il, int jl)
    {
        if constexpr (S == 16) return il + 16 * jl;
        else return il + 8 * (jl & 1) + 32 * ((jl >> 1) & 1) + 16 * ((jl >> 2) & 1);
    }
};

// one add-TID write: value -> byte M0 + OFF + 4 lane
template <int OFF> __device__ __forceinline__ void addtid1(unsigned m0, float a0)
{
    asm volatile("s_mov_b32 m0, %1\n\t"
                 "s_nop 0\n\t"
                 "ds_write_addtid_b32 %0 offset:%2"
                 :
                 : "v"(a0), "s"(m0), "n"(OFF)
                 : "memory", "m0");
}

// exchange 1, slot K0 of this wave (M0 = ma = 4 RQ w'): row w' + W (K0 mod G) of territory K0 / G
template <class SH, int K0> __device__ __forceinline__ void x1_write(unsigned ma, float x)
{
    addtid1<4 * (SH::RQ * SH::W * (K0 % SH::G) + 64 * (K0 / SH::G))>(ma, x);
}
// the wave's own territory (M0 = mc = 256 w): ROW of exchange 2 / the image
template <class SH, int ROW> __device__ __forceinline__ void own_write(unsigned mc, float x) { addtid1<4 * SH::RQ * ROW>(mc, x); }

// eight add-TID writes with one M0 (ro_device_util.h's addtid_write8) for slots Q0 .. Q0 + 7 of a plane
template <class SH, int Q0, typename F> __device__ __forceinline__ void x1_write8(unsigned ma, F f)
{
    constexpr auto off = [](int k0) constexpr { return 4 * (SH::RQ * SH::W * (k0 % SH::G) + 64 * (k0 / SH::G)); };
    addtid_write8<off(Q0), off(Q0 + 1), off(Q0 + 2), off(Q0 + 3), off(Q0 + 4), off(Q0 + 5), off(Q0 + 6), off(Q0 + 7)>(
        ma, f(Q0), f(Q0 + 1), f(Q0 + 2), f(Q0 + 3), f(Q0 + 4), f(Q0 + 5), f(Q0 + 6), f(Q0 + 7));
}
template <class SH, int Q0, bool PI, typename F> __device__ __forceinline__ void own_write8(unsigned mc, F f)
{
    constexpr auto off = [](int q) constexpr { return 4 * SH::RQ * (PI ? SH::pi(q) : q); };
    addtid_write8<off(Q0), off(Q0 + 1), off(Q0 + 2), off(Q0 + 3), off(Q0 + 4), off(Q0 + 5), off(Q0 + 6), off(Q0 + 7)>(
        mc, f(Q0), f(Q0 + 1), f(Q0 + 2), f(Q0 + 3), f(Q0 + 4), f(Q0 + 5), f(Q0 + 6), f(Q0 + 7));
}

// column c of the fft-shifted row in the LDS image (the band scan's view of it)
template <class SH> struct ImageRow {
    const float *img;
    __device__ __forceinline__ float operator()(int c) const
    {
        constexpr int N = SH::N, S = SH::S, G = SH::G;
        const int k = (c + N / 2) & (N - 1), k2 = k >> 10, beta = k & 1023;
        const int k0 = beta & 31, k1 = beta >> 5;
        const int w = k0 / G, j = k0 % G, m = k1 / S, i = k1 % S;
        return img[SH::RQ * (m * S + k2) + 64 * w + SH::lane_of((i - 2 * w) & (S - 1), j)];
    }
};

typedef const volatile __attribute__((address_space(3))) float lds_vfloat;

template <int S, int FMT> __global__ __launch_bounds__(32 * S, 4) void stft_wl_kernel(StftArgs a)
{
    using SH = Shape<S>;
    constexpr int N = SH::N, T = SH::T, W = SH::W, G = SH::G, H = SH::H, NB2 = SH::NB2, RQ = SH::RQ, HL = 16;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    using SM = Sample<FMT>;

    // XCD-aware placement (see stft32k_kernel)
    const int64_t per_xcd = (a.rows + 7) / 8;
    const int64_t xcd_first = (int64_t)(blockIdx.x & 7) * per_xcd;
    const int64_t xcd_end = xcd_first + per_xcd < a.rows ? xcd_first + per_xcd : a.rows;
    const int64_t stride = gridDim.x >> 3;
    int64_t row = xcd_first + (blockIdx.x >> 3);
    if (row >= xcd_end) return;

    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const __amdgpu_buffer_rsrc_t rs_tw = make_rsrc(a.twiddles, SH::TW_TOTAL * 8);
    const __amdgpu_buffer_rsrc_t rs_twk = make_rsrc(a.twiddles_k, SH::PK_TOTAL * 16);
    const char *iq = reinterpret_cast<const char *>(a.iq);
    const float *lds = reinterpret_cast<const float *>(smem);

    v2f v[32];
    // sample loads: lanes l and l + 32 share two neighbouring columns (legs 0..15 / 16..31), 16 bytes per load
    const int po = (((tid & ~63) + 2 * (tid & 31)) + ((tid >> 5) & 1) * HL * (N / 32)) * SM::BYTES;
    auto row_rsrc = [&](int64_t k, bool valid) {
        return make_rsrc(iq + (a.first_row + k) * (int64_t)a.hop * SM::BYTES, valid ? (unsigned)N * SM::BYTES : 0u);
    };
    auto load_row = [&](const __amdgpu_buffer_rsrc_t &rs) {
#pragma unroll
        for (int k = 0; k < HL; ++k) SM::load_pair(rs, po, k * (N / 32) * SM::BYTES, v[k], v[HL + k]);
    };
    load_row(row_rsrc(row, true));
    v4f w4[HL / 2];
    auto load_window = [&](const __amdgpu_buffer_rsrc_t &rs_win) {
#pragma unroll
        for (int k = 0; k < HL; k += 2) {
            const u32x4 t = __builtin_amdgcn_raw_buffer_load_b128(rs_win, tid * 16, (k / 2) * T * 16, 0);
            w4[k / 2] = (v4f){__uint_as_float(t.x), __uint_as_float(t.y), __uint_as_float(t.z), __uint_as_float(t.w)};
        }
    };
    auto win_rsrc = [&](bool valid) { return make_rsrc(a.window_k, valid ? N * 4 : 0); };
    load_window(win_rsrc(true));

    // the image of the row before this one and where it goes; 0 bytes = nothing to store
    const float *prev_out = a.rows_out;
    unsigned prev_bytes = 0;
    // chunk q of the image: bins 4 mg .. 4 mg + 3, mg = tid + T q (eight chunks per row): k0 = 4 (mg & 7) + i4,
    // k1 = (mg & 255) >> 3, k2 = mg >> 8
    int rb_base;
    {
        const int mg = tid, mu = mg & 7, k1 = (mg & 255) >> 3;
        const int w = S == 16 ? mu : mu >> 1, j0 = S == 16 ? 0 : 4 * (mu & 1);
        rb_base = RQ * ((k1 / S) * S + (mg >> 8)) + 64 * w + SH::lane_of((k1 % S - 2 * w) & (S - 1), j0);
    }
    auto store_chunk = [&](int q, const __amdgpu_buffer_rsrc_t &rs) {
        int rb = rb_base;
        asm volatile("" : "+v"(rb));
        // chunk q moves k2 by T q / 256 rows of the image; the four bins of a lane are lanes 16 apart (S = 16) or
        // 8 / 32 / 40 apart (S = 8) in one row
        const float *p = lds + rb + RQ * ((T * q) >> 8);
        constexpr int D1 = S == 16 ? 16 : 8, D2 = S == 16 ? 32 : 32, D3 = S == 16 ? 48 : 40;
        const float x0 = p[0], x1 = p[D1], x2 = p[D2], x3 = p[D3];
        buf_store_f4(x0, x1, x2, x3, rs, tid * 16, ((q * T * 4 + N / 2) & (N - 1)) * 4);
    };

    const unsigned ma = (unsigned)wave * (4u * RQ);              // exchange 1: M0 of this writer wave
    const unsigned mc = (unsigned)wave * 256u;                   // own territory

    for (;;) {
        const __amdgpu_buffer_rsrc_t rs_prev = make_rsrc(prev_out, prev_bytes);
        // ---- window
        {
            const v2f gain2 = (v2f){0.0f, a.gain};          // src/FFTBackend.cpp:78-79: Q += gain
            if (a.gain != 0.0f) {
#pragma unroll
                for (int i = 0; i < 32; ++i) v[i] = v[i] + gain2;
            }
#pragma unroll
            for (int k = 0; k < HL; ++k) {
                v2f &lo = v[k], &hi = v[HL + k];
                const v4f c4 = w4[k / 2];
                const v2f e = lo * ((k & 1) ? c4.zz : c4.xx);
                const v2f o = hi * ((k & 1) ? c4.ww : c4.yy);
                const auto rx = __builtin_amdgcn_permlane32_swap(__float_as_uint(e.x), __float_as_uint(o.x), false, false);
                const auto ry = __builtin_amdgcn_permlane32_swap(__float_as_uint(e.y), __float_as_uint(o.y), false, false);
                lo = (v2f){__uint_as_float(rx[0]), __uint_as_float(ry[0])};
                hi = (v2f){__uint_as_float(rx[1]), __uint_as_float(ry[1])};
            }
        }
        const int64_t next = row + stride;
        const bool has_next = next < xcd_end;

        // ---- pass 0, levels 0..3; the previous row's image goes out between them
        dit32_head(v, [&](auto hc) {
            constexpr int h = decltype(hc)::value;
            store_chunk(2 * h, rs_prev);
            store_chunk(2 * h + 1, rs_prev);
        });
        // pass-1 twiddles of butterfly k0 = G wave + j (units 0 .. 95 of the packed table: {w, w^2} {w^4, w^8} {w^16, -})
        v2f tw1[5];
        {
            int lt = tid;
            asm volatile("" : "+v"(lt));
            const int k0 = G * wave + ((lt & 31) / H);
#pragma unroll
            for (int q = 0; q < 3; ++q) {
                const u32x4 u = __builtin_amdgcn_raw_buffer_load_b128(rs_twk, k0 * 16, q * 32 * 16, 0);
                tw1[2 * q] = (v2f){__uint_as_float(u.x), __uint_as_float(u.y)};
                if (q < 2) tw1[2 * q + 1] = (v2f){__uint_as_float(u.z), __uint_as_float(u.w)};
            }
        }
        wg_sync();                              // (a) every wave has read its part of the old image: LDS is free
        // ---- pass 0, last level: the x plane of exchange 1 leaves as the pairs finish
        dit32_last(v, [&](auto jc) {
            constexpr int j = decltype(jc)::value;
            constexpr int q = bitrev<32>(2 * j);
            x1_write<SH, q>(ma, v[2 * j].x);
            x1_write<SH, q + 1>(ma, v[16 + 2 * j].x);
            x1_write<SH, q + 16>(ma, v[2 * j + 1].x);
            x1_write<SH, q + 17>(ma, v[17 + 2 * j].x);
            return v[17 + 2 * j].y;
        });
        // ---- exchange 1, the rest
        {
            int lt = tid;
            asm volatile("" : "+v"(lt));
            const int l = lt & 63, l5 = l & 31;
            lds_vfloat *g1 = (lds_vfloat *)(lds + RQ * W * (l5 / H) + 64 * wave + (l5 % H) + 32 * (l >> 5));
            auto off = [](int b) constexpr { return SH::RQ * (b / SH::G) + SH::H * (b % SH::G); };
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();                                // (b)
            asm volatile("" ::: "memory");
#pragma unroll
            for (int b = 0; b < 32; ++b) v[b].x = g1[off(b)];
            wg_sync();                                                   // (c)
            auto fy = [&](int k0) { return v[bitrev<32>(k0)].y; };
            x1_write8<SH, 0>(ma, fy);
            x1_write8<SH, 8>(ma, fy);
            x1_write8<SH, 16>(ma, fy);
            x1_write8<SH, 24>(ma, fy);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();                                // (d)
            asm volatile("" ::: "memory");
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                v[i].y = g1[off(i)];
                v[i + 16].y = g1[off(i + 16)];
            }
        }
        // ---- pass 1.  From here to the completed image the wave is on its own.
        fdit32_head(v, tw1[4], tw1[3], tw1[2], tw1[1]);
        // this thread in pass 2: k0 = G wave + jl, k1 = i + S m
        int il_, jl_, i_;
        {
            int lt = tid;
            asm volatile("" : "+v"(lt));
            il_ = SH::lane_il(lt & 63);
            jl_ = SH::lane_jl(lt & 63);
            i_ = (il_ + 2 * wave) & (S - 1);
        }
        // pass-2 twiddles of butterfly m: k = k0 + 32 (i + S m)
        v2f tw2[NB2][TW_SET];
#pragma unroll
        for (int m = 0; m < NB2; ++m) {
            const int k = G * wave + jl_ + 32 * (i_ + S * m);
            if constexpr (S == 16) {
#pragma unroll
                for (int q = 0; q < 3; ++q) {
                    const u32x4 u = __builtin_amdgcn_raw_buffer_load_b128(rs_twk, k * 16, (SH::PK2 + q * 1024) * 16, 0);
                    tw2[m][2 * q] = (v2f){__uint_as_float(u.x), __uint_as_float(u.y)};
                    tw2[m][2 * q + 1] = (v2f){__uint_as_float(u.z), __uint_as_float(u.w)};
                }
            } else {                                                     // radix 8: w, w^2, w^4 (tw_apply's C8 form)
                tw2[m][0] = buf_load_f2(rs_tw, k * 8, (SH::TW2) * 8);
                tw2[m][1] = buf_load_f2(rs_tw, k * 8, (SH::TW2 + 1024) * 8);
                tw2[m][3] = buf_load_f2(rs_tw, k * 8, (SH::TW2 + 3 * 1024) * 8);
            }
        }
        fdit32_last(v, tw1[0], [&](auto jc) {
            constexpr int j = decltype(jc)::value;
            constexpr int q = bitrev<32>(2 * j);
            own_write<SH, SH::pi(q)>(mc, v[2 * j].x);
            own_write<SH, SH::pi(q + 1)>(mc, v[16 + 2 * j].x);
            own_write<SH, SH::pi(q + 16)>(mc, v[2 * j + 1].x);
            own_write<SH, SH::pi(q + 17)>(mc, v[17 + 2 * j].x);
            return v[17 + 2 * j].y;
        });
        // ---- exchange 2 inside the wave: butterfly m reads slot a of row pi(i + S m)
        {
            const float *base = lds + 64 * wave + H * jl_;
            auto col = [](int a) constexpr { return (a >> 1) + 32 * (a & 1); };
            auto fy = [&](int k1) { return v[bitrev<32>(k1)].y; };
            lds_vfloat *g2[NB2];
#pragma unroll
            for (int m = 0; m < NB2; ++m) {
                // pi(i + S m) for a run-time i: the swapped bit pair sits inside i for S = 16 (bits 3 of i and the m bit)
                int k1 = i_ + S * m, lo = S == 16 ? 3 : 2;
                const int b0 = (k1 >> lo) & 1, b1 = (k1 >> (lo + 1)) & 1;
                k1 = (k1 & ~(3 << lo)) | (b0 << (lo + 1)) | (b1 << lo);
                g2[m] = (lds_vfloat *)(base + RQ * k1);
            }
            asm volatile("" ::: "memory");
#pragma unroll
            for (int m = 0; m < NB2; ++m)
#pragma unroll
                for (int r = 0; r < S; ++r) v[m * S + r].x = g2[m][col(r)];
            asm volatile("" ::: "memory");
            own_write8<SH, 0, true>(mc, fy);
            own_write8<SH, 8, true>(mc, fy);
            own_write8<SH, 16, true>(mc, fy);
            own_write8<SH, 24, true>(mc, fy);
#pragma unroll
            for (int m = 0; m < NB2; ++m)
#pragma unroll
                for (int r = 0; r < S; ++r) v[m * S + r].y = g2[m][col(r)];
        }
        // ---- pass 2: 32 / S radix-S butterflies with their stage twiddles
        tw_butterflies<32, S, S == 8>(v, tw2);
        // ---- magnitudes -> image (bin k0 + 32 (i + S m) + 1024 k2 at row m S + k2 of the wave's territory)
        {
            float mg[32];
#pragma unroll
            for (int m = 0; m < NB2; ++m)
#pragma unroll
                for (int k2 = 0; k2 < S; ++k2) {
                    const v2f x = v[m * S + bitrev<S>(k2)];
                    const v2f sq = x * x;
                    mg[m * S + k2] = __builtin_amdgcn_sqrtf(sq.x + sq.y);
                }
            asm volatile("s_nop 7\n\ts_nop 7" ::: "memory");      // the last square roots (transcendental pipe) in front of asm
            auto fm = [&](int q) { return mg[q]; };
            own_write8<SH, 0, false>(mc, fm);
            own_write8<SH, 8, false>(mc, fm);
            own_write8<SH, 16, false>(mc, fm);
            own_write8<SH, 24, false>(mc, fm);
        }
        // the next row's samples and window coefficients (zero-sized descriptors after the last row): two to four
        // workgroups share the CU, the others' butterflies run while these land
        load_row(row_rsrc(has_next ? next : row, has_next));
        load_window(win_rsrc(has_next));
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        wg_sync();                                            // (e) the image of this row is complete
        // ---- BolidRecorder's scan (waves 0, 1) and the band tile (waves 2, 3; S = 8 has four waves in all) on the image
        {
            int lane = tid & 63;
            asm volatile("" : "+v"(lane));
            const ImageRow<SH> img{lds};
            if (a.records != nullptr && wave < 2) {
                int low_noise = a.low_noise, noise_width = a.noise_width, low_detect = a.low_detect;
                int detect_width = a.detect_width, avg_bins = a.avg_bins;
                asm volatile("" : "+s"(low_noise), "+s"(noise_width), "+s"(low_detect), "+s"(detect_width), "+s"(avg_bins));
                if (wave == 0) {
                    unsigned *hist = reinterpret_cast<unsigned *>(smem + SH::IMAGE_BYTES);
                    const float nz = noise_width <= 512 ? scan_noise<8>(img, low_noise, noise_width, hist, lane)
                                                        : scan_noise<0>(img, low_noise, noise_width, hist, lane);
                    if (lane == 0) a.records[row].noise = nz;
                } else {
                    const int pk = scan_peak<8>(img, low_detect, detect_width, lane);
                    const float av = scan_average(img, low_detect + pk - avg_bins / 2, avg_bins, N, lane);
                    if (lane == 0) {
                        a.records[row].peak = pk;
                        a.records[row].average = av;
                    }
                }
            }
            if (a.tile_out != nullptr && (wave == 2 || wave == 3)) {
                int tile_cols = a.tile_cols, tile_first = a.tile_first;
                asm volatile("" : "+s"(tile_cols), "+s"(tile_first));
                const int half = ((tile_cols + 127) >> 7) << 6;
                const int c0 = wave == 2 ? 0 : half;
                const int c1 = wave == 2 ? (half < tile_cols ? half : tile_cols) : tile_cols;
                float *dst = a.tile_out + row * (int64_t)tile_cols;
                if (a.ln_out == nullptr) {
                    for (int c = c0 + lane; c < c1; c += 64) dst[c] = img(tile_first + c);
                } else {
                    float *ldst = a.ln_out + row * (int64_t)tile_cols;
                    unsigned kmin = 0xffffffffu, kmax = 0u;
                    for (int c = c0 + lane; c < c1; c += 64) {
                        const float x = img(tile_first + c);
                        const float l = logf(x);
                        dst[c] = x;
                        ldst[c] = l;
                        if (x != 0.f) {
                            const unsigned key = order_key(l);
                            kmin = min(kmin, key);
                            kmax = max(kmax, key);
                        }
                    }
                    kmin = wave_min_u32(kmin);
                    kmax = wave_max_u32(kmax);
                    if (lane == 0) {
                        float *part = a.ln_part + row * 4 + (wave == 2 ? 0 : 2);
                        part[0] = kmin == 0xffffffffu ? __builtin_inff() : key_to_float(kmin);
                        part[1] = kmax == 0u ? -__builtin_inff() : key_to_float(kmax);
                    }
                }
            }
        }
        prev_out = a.rows_out + row * a.row_stride;
        prev_bytes = N * 4;
        if (!has_next) break;
        row = next;
    }
    // the last row's image (complete: the loop ends behind its barrier); nothing overwrites LDS any more
    {
        const __amdgpu_buffer_rsrc_t rs_last = make_rsrc(prev_out, prev_bytes);
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            store_chunk(q, rs_last);
            if (q & 1) asm volatile("" ::: "memory");
        }
    }
}

struct DevicePlanWl {
    bool ready = false;
    int resident = 0, per_cu = 1;
};

template <int S, int FMT> static hipError_t launch_fmt(const StftArgs &a, hipStream_t s)
{
    using SH = Shape<S>;
    static std::mutex lock;
    static DevicePlanWl table[64];
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    if (dev < 0 || dev >= 64) return hipErrorInvalidDevice;
    DevicePlanWl d;
    {
        std::lock_guard<std::mutex> g(lock);
        DevicePlanWl &t = table[dev];
        if (!t.ready) {
            const void *fn = reinterpret_cast<const void *>(&stft_wl_kernel<S, FMT>);
            if ((e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, SH::LDS_BYTES)) != hipSuccess) return e;
            int cus = 0, per_cu = 0;
            if ((e = hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev)) != hipSuccess) return e;
            if ((e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fn, SH::T, SH::LDS_BYTES)) != hipSuccess) return e;
            t.per_cu = per_cu < 1 ? 1 : per_cu;
            t.resident = cus * t.per_cu;
            t.ready = true;
        }
        d = t;
    }
    const int64_t per_xcd = (a.rows + 7) / 8;
    int64_t slots = d.resident / 8;
    if (a.spare_cus > 0) slots -= (int64_t)a.spare_cus * d.per_cu;
    if (slots < 1) slots = 1;
    if (slots > per_xcd) slots = per_xcd;
    StftArgs b = a;
    b.dec = 1;
    b.dec_log2 = 0;
    b.prefetch = 0;
    b.stagger = 0;
    hipLaunchKernelGGL((stft_wl_kernel<S, FMT>), dim3((unsigned)(slots * 8)), dim3(SH::T), SH::LDS_BYTES, s, b);
    return hipGetLastError();
}

}  // namespace wl

// magnitude rows of N = 16384 / 8192 (launch_stft routes to it)
hipError_t launch_stft_wl(int bins, int fmt, const StftArgs &a, hipStream_t s)
{
    if (a.rows <= 0) return hipSuccess;
    if (a.spec_out != nullptr || a.big_form) return hipErrorInvalidValue;
    if (bins == 16384) {
        if (fmt == RO_FMT_F32) return wl::launch_fmt<16, RO_FMT_F32>(a, s);
        if (fmt == RO_FMT_I16) return wl::launch_fmt<16, RO_FMT_I16>(a, s);
    } else if (bins == 8192) {
        if (fmt == RO_FMT_F32) return wl::launch_fmt<8, RO_FMT_F32>(a, s);
        if (fmt == RO_FMT_I16) return wl::launch_fmt<8, RO_FMT_I16>(a, s);
    }
    return hipErrorInvalidValue;
}

}  // namespace ro
