// ro_stft32k.hip -- the N = 32768 magnitude-row kernel (BASELINE.json configs 3, 4, 5): window -> FFT -> |X| ->
// fft-shift -> float32 row, with BolidRecorder's per-row scan and the band tile cut from the row while it is in LDS.
//   replaces src/FFTBackend.cpp:229-236 (window multiply + fftw_execute), src/WaterfallBackend.cpp:485-505
//   (magnitude + shift) and src/BolidRecorder.cpp:121-132, :313-347 (noise / peak / average) of the reference.
//
// One 1024-thread workgroup per CU, persistent, one row at a time: the complex row (256 KiB) lives in the
// workgroup's registers as 32 points per thread and the transform is three radix-32 passes,
//   n = n1 + 1024 n0,  n1 = a + 32 b          k = k0 + 32 k1 + 1024 k2
//   pass 0 over n0 -> k0   (thread = column n1; the samples arrive coalesced, 16 bytes per lane)
//   pass 1 over b  -> k1   (thread = (k0, a))
//   pass 2 over a  -> k2   (thread = (k0, k1))
// The exchange between pass 0 and pass 1 is a transposition over the whole workgroup (every wave needs data of
// every other wave).  From there on a wave owns two values of k0 and ALL of their a, so the exchange between pass
// 1 and pass 2 is a 32 x 32 transposition inside each half of the wave: no workgroup barrier between the last read
// of exchange 1 and the completed magnitude image -- two thirds of the row's butterflies -- and the waves of a SIMD
// drift apart, one wave's LDS traffic running under another's butterflies.  (The stft_kernel<Plan32768> this replaces
// made both exchanges over the whole workgroup: ten barriers per row held all sixteen waves in the same phase, so
// VALU, LDS and the memory pipe took turns: 0.42 of the HBM roofline.)  Five barriers per row are left.
//
// ONE LDS layout serves the three uses (floats; real and imaginary plane one after the other, a complex row does
// not fit 160 KiB):
//   cell(q, w, l) = 1025 q + 64 w + l        q < 32 a row, w < 16 the wave whose territory it is, l < 64
// * every write is ds_write_addtid_b32 (address = M0 + offset + 4 lane: lane-linear, no address VGPR, twice the
//   rate of ds_write_b32); M0 and the offset field hold 16 bits each and every cell is within their reach;
// * 1025 is odd: a column read (32 rows, one cell each) walks 32 different banks;
// * a wave only ever READS its own territory until the image is complete, and nobody writes into a territory
//   between the barrier in front of exchange 1's reads and the next row: that is what removes the barriers.
//   exchange 1:  slot k0 of pass-0 wave w', lane l  ->  cell(w' + 16 (k0 & 1), k0 >> 1, l)
//                pass-1 lane (a >> 1) + 16 kb + 32 (a & 1) of wave w is thread (k0 = 2 w + kb, a) and reads slot b
//                from cell((b >> 1) + 16 kb, w, (a >> 1) + 16 (b & 1) + 32 (a & 1))
//   exchange 2:  slot k1 of that thread -> cell(k1, w, lane)
//                pass-2 lane l' of wave w is thread (k0 = 2 w + (l' >> 5), k1 = (l' + 4 (w >> 1)) & 31) and reads
//                slot a from cell(k1, w, (a >> 1) + 16 kb + 32 (a & 1))
//   image:       slot k2 of that thread = bin k0 + 32 k1 + 1024 k2 -> cell(k2, w, l')
//                (the rotation by 4 (w >> 1) makes the 16-byte-per-lane read-back of the row conflict-free)
// tools/r3/emu32k.py restates these maps with numpy and checks them against numpy's FFT and for bank conflicts.
#ifndef RO_K32_ABLATE
#define RO_K32_ABLATE 0
#endif
#if RO_K32_ABLATE & 8
#define RO_FFT_NO_BFLY 1
#endif
// the butterflies' scheduling leash as a scheduling barrier, not an empty asm statement (see tie() in ro_fft_device.h):
// 240 fewer s_nop per row and wave, 7 VGPRs fewer, the same bits
#ifndef RO_TIE_SCHED
#define RO_TIE_SCHED 1
#endif
#include "ro_kernels.h"
#include "ro_fft_device.h"
#include "ro_device_util.h"

#include <mutex>

// Diagnostic only: -DRO_STAMPS32K=1 accumulates s_memtime deltas per phase of the row loop (every wave of every
// workgroup) into StftArgs::stamps.  Never timed, never shipped.
#ifndef RO_STAMPS32K
#define RO_STAMPS32K 0
#endif
// butterfly pairs of pass 2's last level that request next-row samples (of 8); the rest is requested behind the scan
#ifndef RO_K32_PIPE_J
#define RO_K32_PIPE_J 6
#endif
// share (percent) of the next row's window coefficients requested right behind the window stage
#ifndef RO_K32_WIN_EARLY_PCT
#define RO_K32_WIN_EARLY_PCT 25
#endif

// Diagnostic builds only (tools/ab_build.sh): RO_K32_ABLATE bits remove one kind of memory traffic to price it (1: no
// twiddle loads, 2: no window loads, 4: no LDS exchanges, 8: no butterflies (-DRO_FFT_NO_BFLY), 16: no row stores;
// results are wrong by design); RO_K32_PRIO tries wave priorities
// between exchange 1 and the image (1: by progress -- 3, 2, 1, 0 through the four butterfly phases; 2: static, youngest
// wave of a SIMD highest).
#ifndef RO_K32_ABLATE
#define RO_K32_ABLATE 0
#endif
#ifndef RO_K32_PRIO
#define RO_K32_PRIO 0
#endif
// the next row's last sample legs and window coefficients are requested in front of the image-complete barrier by
// every wave that does not scan (1), or behind the scan by all (0)
#ifndef RO_K32_EARLY_LOADS
#define RO_K32_EARLY_LOADS 1
#endif
#ifndef RO_K32_TW1_SCALAR
#define RO_K32_TW1_SCALAR 1
#endif

namespace ro {
namespace k32 {

constexpr int N = 32768, T = 1024, H = 16;
constexpr int RQ = 1025;                              // floats per row of the LDS layout
constexpr int IMAGE_BYTES = 32 * RQ * 4;              // 131200
constexpr int LDS_BYTES = IMAGE_BYTES + 1024;         // + the fused scan's radix-select histogram
constexpr int HB = 61568;                             // added to M0 where offset + base would not fit 16 bits
static_assert(15 * 256 + HB <= 65535 && 31 * 4 * RQ - HB <= 65535 && 16 * 4 * RQ - HB >= 0, "rows 16..31: M0 / offset split");
static_assert(15 * 4 * RQ + 4032 <= 65535 && 4 * 16 * RQ + 15 * 256 - 4032 <= 65535, "exchange 1, odd slots: M0 / offset split");

// exchange 1, the four slots q, q+1, q+16, q+17 (q even) a last-level pair finishes: even slots from M0 = ma = 4100 w,
// odd slots from mb = ma + 4032
template <int Q>
__device__ __forceinline__ void x1_write_pair(unsigned ma, unsigned mb, float s_q, float s_q1, float s_q16, float s_q17)
{
    static_assert(Q % 2 == 0 && Q < 16, "slot algebra");
    constexpr int E = 256 * (Q >> 1), O = 4 * 16 * RQ - 4032 + 256 * (Q >> 1);
    addtid_write4<E, E + 2048, O, O + 2048>(ma, mb, s_q, s_q16, s_q1, s_q17);
}
// a whole plane of exchange 1: f(k0) for the 32 slots
template <typename F> __device__ __forceinline__ void x1_write_plane(unsigned ma, unsigned mb, F f)
{
    constexpr int O = 4 * 16 * RQ - 4032;
    addtid_write8<0, 256, 512, 768, 1024, 1280, 1536, 1792>(ma, f(0), f(2), f(4), f(6), f(8), f(10), f(12), f(14));
    addtid_write8<2048, 2304, 2560, 2816, 3072, 3328, 3584, 3840>(ma, f(16), f(18), f(20), f(22), f(24), f(26), f(28), f(30));
    addtid_write8<O, O + 256, O + 512, O + 768, O + 1024, O + 1280, O + 1536, O + 1792>(mb, f(1), f(3), f(5), f(7), f(9), f(11),
                                                                                        f(13), f(15));
    addtid_write8<O + 2048, O + 2304, O + 2560, O + 2816, O + 3072, O + 3328, O + 3584, O + 3840>(
        mb, f(17), f(19), f(21), f(23), f(25), f(27), f(29), f(31));
}
// rows q, q+1, q+16, q+17 of the wave's own territory (exchange 2 and the image): rows < 16 from M0 = mc = 256 w,
// rows >= 16 from md = mc + HB
template <int Q>
__device__ __forceinline__ void own_write_pair(unsigned mc, unsigned md, float s_q, float s_q1, float s_q16, float s_q17)
{
    static_assert(Q % 2 == 0 && Q < 16, "slot algebra");
    constexpr int R = 4 * RQ;
    addtid_write4<R * Q, R * (Q + 1), R * (Q + 16) - HB, R * (Q + 17) - HB>(mc, md, s_q, s_q1, s_q16, s_q17);
}
template <typename F> __device__ __forceinline__ void own_write_plane(unsigned mc, unsigned md, F f)
{
    constexpr int R = 4 * RQ;
    addtid_write8<0 * R, 1 * R, 2 * R, 3 * R, 4 * R, 5 * R, 6 * R, 7 * R>(mc, f(0), f(1), f(2), f(3), f(4), f(5), f(6), f(7));
    addtid_write8<8 * R, 9 * R, 10 * R, 11 * R, 12 * R, 13 * R, 14 * R, 15 * R>(mc, f(8), f(9), f(10), f(11), f(12), f(13),
                                                                                  f(14), f(15));
    addtid_write8<16 * R - HB, 17 * R - HB, 18 * R - HB, 19 * R - HB, 20 * R - HB, 21 * R - HB, 22 * R - HB, 23 * R - HB>(
        md, f(16), f(17), f(18), f(19), f(20), f(21), f(22), f(23));
    addtid_write8<24 * R - HB, 25 * R - HB, 26 * R - HB, 27 * R - HB, 28 * R - HB, 29 * R - HB, 30 * R - HB, 31 * R - HB>(
        md, f(24), f(25), f(26), f(27), f(28), f(29), f(30), f(31));
}

// column c of the fft-shifted row in the LDS image (the band scan's view of it)
struct ImageRow {
    const float *img;
    __device__ __forceinline__ float operator()(int c) const
    {
        const int k = (c + N / 2) & (N - 1), r = k >> 10, beta = k & 1023;
        const int w = (beta & 31) >> 1, kb = beta & 1, k1 = beta >> 5;
        return img[RQ * r + 64 * w + ((k1 - 4 * (w >> 1)) & 31) + 32 * kb];
    }
};

typedef const volatile __attribute__((address_space(3))) float lds_vfloat;

template <int FMT> __global__ __launch_bounds__(T, 1) void stft32k_kernel(StftArgs a)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    using S = Sample<FMT>;

    // XCD-aware placement: workgroups b and b+8 share an XCD (round-robin dispatch), so each XCD gets one contiguous
    // run of rows and its workgroups take consecutive rows of it at the same time -- consecutive rows share
    // (N-hop)/N of their input through that XCD's L2.  Placement affects speed only.
    const int64_t per_xcd = (a.rows + 7) / 8;
    const int64_t xcd_first = (int64_t)(blockIdx.x & 7) * per_xcd;
    const int64_t xcd_end = xcd_first + per_xcd < a.rows ? xcd_first + per_xcd : a.rows;
    const int64_t stride = gridDim.x >> 3;
    int64_t row = xcd_first + (blockIdx.x >> 3);
    if (row >= xcd_end) return;

    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const __amdgpu_buffer_rsrc_t rs_twk = make_rsrc(a.twiddles_k, (3 * 32 + 3 * 1024) * 16);
    const char *iq = reinterpret_cast<const char *>(a.iq);
    const float *lds = reinterpret_cast<const float *>(smem);

    v2f v[32];

    // ---- sample loads.  Lanes l and l + 32 of a wave share two neighbouring columns: lane l < 32 fetches both for
    // legs 0..15, lane l + 32 for legs 16..31, 16 bytes per load (v[k] / v[16 + k] = even / odd column of the leg);
    // the window stage multiplies them in place and v_permlane32_swap gives every lane one whole column:
    // position t of the workgroup ends up with column (t & ~63) + 2 (t & 31) + ((t >> 5) & 1).
    const int po = (((tid & ~63) + 2 * (tid & 31)) + ((tid >> 5) & 1) * H * (N / 32)) * S::BYTES;
    auto row_rsrc = [&](int64_t k, bool valid) {
        return make_rsrc(iq + (a.first_row + k) * (int64_t)a.hop * S::BYTES, valid ? (unsigned)N * S::BYTES : 0u);
    };
    {
        const __amdgpu_buffer_rsrc_t rs = row_rsrc(row, true);
#pragma unroll
        for (int k = 0; k < H; ++k) S::load_pair(rs, po, k * (N / 32) * S::BYTES, v[k], v[H + k]);
    }
    // window coefficients in the kernel's own order (StftArgs::window_k, stft_window_layout): 16 bytes per lane =
    // {even, odd column of leg k, even, odd column of leg k + 1}
    v4f w4[H / 2];
    auto load_window = [&](const __amdgpu_buffer_rsrc_t &rs_win, auto first_c, auto last_c) {
        constexpr int first = decltype(first_c)::value, last = decltype(last_c)::value;
#pragma unroll
        for (int k = first; k < last; k += 2) {
            if constexpr (RO_K32_ABLATE & 2) { w4[k / 2] = (v4f){0.5f, 0.25f, 0.5f, 0.25f}; continue; }
            const u32x4 t = __builtin_amdgcn_raw_buffer_load_b128(rs_win, tid * 16, (k / 2) * T * 16, 0);
            w4[k / 2] = (v4f){__uint_as_float(t.x), __uint_as_float(t.y), __uint_as_float(t.z), __uint_as_float(t.w)};
        }
    };
    constexpr int NW_EARLY = ((H * RO_K32_WIN_EARLY_PCT) / 100) & ~1;
    using c0 = std::integral_constant<int, 0>;
    using cE = std::integral_constant<int, NW_EARLY>;
    using cN = std::integral_constant<int, H>;
    auto win_rsrc = [&](bool valid) { return make_rsrc(a.window_k, valid ? N * 4 : 0); };
    load_window(win_rsrc(true), c0{}, cN{});

    // stage twiddles: {w, w^2} {w^4, w^8} {w^16, -} of butterfly k from the packed table (three 16-byte loads);
    // fdit32 makes the other powers.  Pass 1: k = k0 (32 entries at unit 0), pass 2: k = k0 + 32 k1 (1024 at unit 96).
    v2f tw1[5], tw2[5];
    auto tw_load = [&](v2f (&t)[5], int k, auto pk_c, auto ns_c) {
        constexpr int PK = decltype(pk_c)::value, NS = decltype(ns_c)::value;
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            if constexpr (RO_K32_ABLATE & 1) { t[2 * q] = (v2f){0.6f, 0.8f}; if (q < 2) t[2 * q + 1] = (v2f){0.8f, 0.6f}; continue; }
            const u32x4 u = __builtin_amdgcn_raw_buffer_load_b128(rs_twk, k * 16, (PK + q * NS) * 16, 0);
            t[2 * q] = (v2f){__uint_as_float(u.x), __uint_as_float(u.y)};
            if (q < 2) t[2 * q + 1] = (v2f){__uint_as_float(u.z), __uint_as_float(u.w)};
        }
    };

    unsigned long long st_acc[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, st_prev = 0;
    auto stamp = [&](int k) {
        if constexpr (RO_STAMPS32K) {
            unsigned long long t;
            __builtin_amdgcn_sched_barrier(0);
            asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
            __builtin_amdgcn_sched_barrier(0);
            if (k >= 0) st_acc[k] += t - st_prev;
            st_prev = t;
        }
    };
    stamp(-1);
    if constexpr (RO_STAMPS32K) st_acc[13] = __builtin_amdgcn_s_memrealtime();   // start, on the chip-wide 100 MHz counter

    // the image of the row before this one and where it goes; 0 bytes = nothing to store
    const float *prev_out = a.rows_out;
    unsigned prev_bytes = 0;
    unsigned touch = 0;
    // chunk q of the image: bins 4 m .. 4 m + 3 of segment r = 4 q + (tid >> 8), m = tid & 255, sit in the territories of
    // waves 2 (m & 7) and 2 (m & 7) + 1 (two lanes each); out as 1 KiB per wave-instruction; bin k leaves for column
    // (k + N/2) mod N (src/WaterfallBackend.cpp:492-505)
    // (ONE register holds the thread's chunk-0 address for the whole kernel; laundered per chunk, else hipcc keeps all
    // eight chunk addresses alive through the row -- and recomputed from tid per chunk it cost ~100 VALU ops per row)
    int rb_base = RQ * (tid >> 8) + 128 * (tid & 7) + ((((tid & 255) >> 3) - 4 * (tid & 7)) & 31);
    auto store_chunk = [&](int q, const __amdgpu_buffer_rsrc_t &rs) {
        int rb = rb_base;
        asm volatile("" : "+v"(rb));
        const float *p = lds + rb + 4 * RQ * q;
        const float x0 = p[0], x1 = p[32], x2 = p[64], x3 = p[96];
        if constexpr (RO_K32_ABLATE & 16) asm volatile("" ::"v"(x0), "v"(x1), "v"(x2), "v"(x3));
        else buf_store_f4(x0, x1, x2, x3, rs, tid * 16, ((q * T * 4 + N / 2) & (N - 1)) * 4);
    };
    auto prio = [&](auto pc) {
        constexpr int P = decltype(pc)::value;
        if constexpr (RO_K32_PRIO == 1) __builtin_amdgcn_s_setprio(P);
    };
    // RO_K32_PRIO 3 / 4: one step per butterfly LEVEL (3: passes 1 and 2, 4: pass 0 as well): level n of the row runs at
    // priority 3 - (n mod 4), so a wave that is a level behind outranks the waves ahead of it
    auto lprio = [&](auto nc) {
        constexpr int n = decltype(nc)::value;
        if constexpr (RO_K32_PRIO == 4 || (RO_K32_PRIO == 3 && n >= 5)) __builtin_amdgcn_s_setprio(3 - (n & 3));
    };
    if constexpr (RO_K32_PRIO == 2) {
        if (wave >= 12) __builtin_amdgcn_s_setprio(3);
        else if (wave >= 8) __builtin_amdgcn_s_setprio(2);
        else if (wave >= 4) __builtin_amdgcn_s_setprio(1);
    }
    using p0 = std::integral_constant<int, 0>;
    using p1 = std::integral_constant<int, 1>;
    using p2 = std::integral_constant<int, 2>;
    using p3 = std::integral_constant<int, 3>;

    const unsigned ma = (unsigned)wave * (4u * RQ), mb = ma + 4032u;      // exchange 1: M0 of the even / odd slots
    const unsigned mc = (unsigned)wave * 256u, md = mc + (unsigned)HB;    // own territory: M0 of rows < 16 / >= 16

    for (;;) {
        const __amdgpu_buffer_rsrc_t rs_prev = make_rsrc(prev_out, prev_bytes);
        // ---- window (coefficients and samples were requested a whole epilogue ago)
        {
            const v2f gain2 = (v2f){0.0f, a.gain};          // src/FFTBackend.cpp:78-79: Q += gain
            if (a.gain != 0.0f) {                           // every shipped config has iq_gain = 0: skip the adds
#pragma unroll
                for (int i = 0; i < 32; ++i) v[i] = v[i] + gain2;
            }
#pragma unroll
            for (int k = 0; k < H; ++k) {
                v2f &lo = v[k], &hi = v[H + k];
                const v4f c4 = w4[k / 2];
                const v2f e = lo * ((k & 1) ? c4.zz : c4.xx);          // even column, leg k (16 + k on lanes >= 32)
                const v2f o = hi * ((k & 1) ? c4.ww : c4.yy);          // odd column
                // lanes 0..31 hold legs k, lanes 32..63 legs 16 + k of both columns: the upper half of slot k trades
                // places with the lower half of slot 16 + k
                const auto rx = __builtin_amdgcn_permlane32_swap(__float_as_uint(e.x), __float_as_uint(o.x), false, false);
                const auto ry = __builtin_amdgcn_permlane32_swap(__float_as_uint(e.y), __float_as_uint(o.y), false, false);
                lo = (v2f){__uint_as_float(rx[0]), __uint_as_float(ry[0])};
                hi = (v2f){__uint_as_float(rx[1]), __uint_as_float(ry[1])};
            }
        }
        asm volatile("" ::"v"(touch));                       // see touch_next
        const int64_t next = row + stride;
        const bool has_next = next < xcd_end;
        // the first coefficients of the NEXT row right away: their registers are free for the whole transform
        load_window(win_rsrc(true), c0{}, cE{});
        // The hop new samples of the workgroup's next row are touched (one dword per 128-byte line, value unused) well
        // before the epilogue asks for them: they come from HBM, every other byte of the row from L2, and that one
        // miss latency sat on the critical path of every row.  The register is "used" after the next window stage.
        auto touch_next = [&]() {
            const int64_t s0 = (a.first_row + (has_next ? next : row)) * (int64_t)a.hop + (N - a.hop);
            const __amdgpu_buffer_rsrc_t rs_new =
                make_rsrc(iq + s0 * S::BYTES, (has_next && a.prefetch) ? a.hop * S::BYTES : 0);
            touch = __builtin_amdgcn_raw_buffer_load_b32(rs_new, tid * 128, 0, 0);
        };
        stamp(0);

        // ---- pass 0, levels 0..3; the previous row's image goes out between them (LDS read-back + 16-byte stores)
        lprio(std::integral_constant<int, 0>{});
        dit32_head(v, [&](auto hc) {
            constexpr int h = decltype(hc)::value;
            store_chunk(2 * h, rs_prev);
            store_chunk(2 * h + 1, rs_prev);
            lprio(std::integral_constant<int, h + 1>{});
        });
        if constexpr (RO_K32_TW1_SCALAR) {
            // pass-1 twiddles depend on k0 only -- two values per wave: scalar loads (the scalar cache, not the vector
            // memory pipe, which at this point is busy with the row's samples) and one select per register
            int lt = tid;
            asm volatile("" : "+v"(lt));
            const bool odd = (lt >> 4) & 1;
            const float4 *tk = a.twiddles_k + 2 * wave;
#pragma unroll
            for (int q = 0; q < 3; ++q) {
                v4f e, o;                                   // units q * 32 + k0 for the wave's even and odd k0
                asm volatile("s_load_dwordx4 %0, %2, %3\n\t"
                             "s_load_dwordx4 %1, %2, %4\n\t"
                             "s_waitcnt lgkmcnt(0)"
                             : "=&s"(e), "=&s"(o)
                             : "s"(tk), "n"(q * 32 * 16), "n"(q * 32 * 16 + 16)
                             : "memory");
                tw1[2 * q] = (v2f){odd ? o.x : e.x, odd ? o.y : e.y};
                if (q < 2) tw1[2 * q + 1] = (v2f){odd ? o.z : e.z, odd ? o.w : e.w};
            }
        } else {
            int lt = tid;
            asm volatile("" : "+v"(lt));
            tw_load(tw1, 2 * wave + ((lt >> 4) & 1), std::integral_constant<int, 0>{}, std::integral_constant<int, 32>{});
        }
        stamp(1);
        wg_sync();                              // (a) every wave has read its part of the old image: LDS is free
        stamp(2);
        // ---- pass 0, last level: the x plane of exchange 1 leaves as the pairs finish
        dit32_last(v, [&](auto jc) {
            constexpr int j = decltype(jc)::value;
            if constexpr (!(RO_K32_ABLATE & 4)) x1_write_pair<bitrev<32>(2 * j)>(ma, mb, v[2 * j].x, v[16 + 2 * j].x, v[2 * j + 1].x, v[17 + 2 * j].x);
            return v[17 + 2 * j].y;
        });
        stamp(3);
        // ---- exchange 1, the rest: thread (k0, a) of pass 1 reads slot b from its own territory
        {
            int lt = tid;
            asm volatile("" : "+v"(lt));
            const int l = lt & 63;
            lds_vfloat *g1 = (lds_vfloat *)(lds + RQ * 16 * ((l >> 4) & 1) + 64 * wave + (l & 15) + 32 * (l >> 5));
            auto off = [](int b) constexpr { return RQ * (b >> 1) + 16 * (b & 1); };
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");           // the x-plane writes (hipcc does not count them)
            __builtin_amdgcn_s_barrier();                                // (b)
            asm volatile("" ::: "memory");
#pragma unroll
            for (int b = 0; b < 32; ++b) if constexpr (!(RO_K32_ABLATE & 4)) v[b].x = g1[off(b)];
            wg_sync();                                                   // (c) everyone has its x: the plane may go
            if constexpr (!(RO_K32_ABLATE & 4)) x1_write_plane(ma, mb, [&](int k0) { return v[bitrev<32>(k0)].y; });
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();                                // (d)
            asm volatile("" ::: "memory");
#pragma unroll
            for (int i = 0; i < 16; ++i) {                               // in the order level 0 of pass 1 pairs them
                if constexpr (RO_K32_ABLATE & 4) continue;
                v[i].y = g1[off(i)];
                v[i + 16].y = g1[off(i + 16)];
            }
        }
        stamp(4);
        // ---- pass 1.  From here to the completed image the wave is on its own.
        prio(p3{});
        lprio(std::integral_constant<int, 5>{});
        fdit32_head(v, tw1[4], tw1[3], tw1[2], tw1[1], [&](auto hc) { lprio(std::integral_constant<int, 6 + decltype(hc)::value>{}); });
        touch_next();
        int k1p, kbp;                                                    // this thread in pass 2: (k0 = 2 wave + kbp, k1 = k1p)
        {
            int lt = tid;
            asm volatile("" : "+v"(lt));
            k1p = ((lt & 31) + 4 * (wave >> 1)) & 31;
            kbp = (lt >> 5) & 1;
            tw_load(tw2, 2 * wave + kbp + 32 * k1p, std::integral_constant<int, 96>{}, std::integral_constant<int, 1024>{});
        }
        stamp(5);
        prio(p2{});
        fdit32_last(v, tw1[0], [&](auto jc) {
            constexpr int j = decltype(jc)::value;
            if constexpr (!(RO_K32_ABLATE & 4)) own_write_pair<bitrev<32>(2 * j)>(mc, md, v[2 * j].x, v[16 + 2 * j].x, v[2 * j + 1].x, v[17 + 2 * j].x);
            return v[17 + 2 * j].y;
        });
        stamp(6);
        // ---- exchange 2: a 32 x 32 transposition inside each half of the wave, through the wave's own rows.  One
        // wave's LDS instructions execute in order: no wait between its writes and its reads of the same cells.
        {
            lds_vfloat *g2 = (lds_vfloat *)(lds + RQ * k1p + 64 * wave + 16 * kbp);
            auto off = [](int s) constexpr { return (s >> 1) + 32 * (s & 1); };
            asm volatile("" ::: "memory");
#pragma unroll
            for (int s = 0; s < 32; ++s) if constexpr (!(RO_K32_ABLATE & 4)) v[s].x = g2[off(s)];
            asm volatile("" ::: "memory");
            if constexpr (!(RO_K32_ABLATE & 4)) own_write_plane(mc, md, [&](int k1) { return v[bitrev<32>(k1)].y; });
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                if constexpr (RO_K32_ABLATE & 4) continue;
                v[i].y = g2[off(i)];
                v[i + 16].y = g2[off(i + 16)];
            }
        }
        stamp(7);
        // ---- pass 2
        prio(p1{});
        lprio(std::integral_constant<int, 10>{});
        fdit32_head(v, tw2[4], tw2[3], tw2[2], tw2[1], [&](auto hc) { lprio(std::integral_constant<int, 11 + decltype(hc)::value>{}); });
        prio(p0{});
        stamp(8);
        {
            // Last level with the epilogue folded in.  After butterflies (j, 8 + j) x[2j], x[2j+1], x[16+2j], x[17+2j]
            // are final = bins k0 + 32 k1 + 1024 q for q = qj, qj+16, qj+1, qj+17 (qj = bitrev32(2j)): their magnitudes
            // go to rows q of the wave's territory, and the four freed registers receive legs 2j, 2j+1 of the NEXT
            // row's samples -- for j < PIPE_J; the last legs are requested behind the scan.
            const __amdgpu_buffer_rsrc_t rs_next = row_rsrc(has_next ? next : row, has_next);   // zero-sized after the last row
            // The image writes of pair j are issued one pair late (from done(j + 1), the last ones behind the level):
            // v_sqrt_f32 runs in the transcendental pipe and hipcc pads no hazards in front of inline asm.
            float pm0 = 0.f, pm1 = 0.f, pm16 = 0.f, pm17 = 0.f;
            fdit32_last(v, tw2[0], [&](auto jc) {
                constexpr int j = decltype(jc)::value;
                auto mag = [](v2f x) { const v2f sq = x * x; return __builtin_amdgcn_sqrtf(sq.x + sq.y); };
                const float m0 = mag(v[2 * j]), m16 = mag(v[2 * j + 1]);
                const float m1 = mag(v[16 + 2 * j]), m17 = mag(v[17 + 2 * j]);
                if constexpr (j > 0) own_write_pair<bitrev<32>(2 * (j > 0 ? j - 1 : 0))>(mc, md, pm0, pm1, pm16, pm17);
                pm0 = m0; pm1 = m1; pm16 = m16; pm17 = m17;
                if constexpr (j < RO_K32_PIPE_J) {
                    const int pj = after(po, m17);      // the loads may not start before these magnitudes exist
                    S::load_pair(rs_next, pj, (2 * j) * (N / 32) * S::BYTES, v[2 * j], v[H + 2 * j]);
                    S::load_pair(rs_next, pj, (2 * j + 1) * (N / 32) * S::BYTES, v[2 * j + 1], v[H + 2 * j + 1]);
                }
                return m17;
            });
            asm volatile("s_nop 7\n\ts_nop 7" ::: "memory");      // the last square roots: see above
            own_write_pair<bitrev<32>(14)>(mc, md, pm0, pm1, pm16, pm17);
            // The legs the last level did not request and the rest of the window coefficients.  The waves reach this
            // point up to ~6k cycles apart (the oldest wave of a SIMD first) and then wait for the barrier: whoever is
            // not about to scan asks NOW, so the memory pipe works through most of the next row's 384 KiB while the
            // younger waves are still in their butterflies, instead of starting behind the barrier with all 16 waves
            // in its queue.  (The two scanning waves need these registers for their band.)
            // who has work on the image behind the barrier: waves 0, 1 scan, waves 2, 3 cut the band tile
            const bool image_work = RO_K32_EARLY_LOADS ? ((a.records != nullptr && wave < 2) ||
                                                          (a.tile_out != nullptr && (wave == 2 || wave == 3)))
                                                       : true;
            auto late_loads = [&]() {
#pragma unroll
                for (int k = 2 * RO_K32_PIPE_J; k < H; ++k)
                    S::load_pair(rs_next, po, k * (N / 32) * S::BYTES, v[k], v[H + k]);
                load_window(win_rsrc(has_next), cE{}, cN{});
            };
            auto image_complete = [&]() {
                stamp(9);
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the add-TID image writes (hipcc does not count them)
                wg_sync();                                            // (e) the image of this row is complete
                stamp(10);
            };
            if (!image_work) {
                // (its own branch, barrier included: with one barrier for both paths hipcc keeps the registers of these
                // loads live across the scan code and spills)
                late_loads();
                image_complete();
            } else {
                image_complete();
                // BolidRecorder's per-row scan on the image (src/BolidRecorder.cpp:121-132, :313-347) by waves 0 and 1 --
                // the oldest wave of two SIMDs, which the arbiter serves first -- while the others go on to the next row's
                // window stage; waves 2 and 3 cut the band tile.  Everything derived from the lane number and the band
                // limits is laundered through empty asm: otherwise hipcc hoists those loop invariants in front of the row
                // loop, where they sit in VGPRs of all 16 waves for the whole row.
                {
                    int lane = tid & 63;
                    asm volatile("" : "+v"(lane));
                    const ImageRow img{lds};
                    if (a.records != nullptr && wave < 2) {
                        int low_noise = a.low_noise, noise_width = a.noise_width, low_detect = a.low_detect;
                        int detect_width = a.detect_width, avg_bins = a.avg_bins;
                        asm volatile("" : "+s"(low_noise), "+s"(noise_width), "+s"(low_detect), "+s"(detect_width), "+s"(avg_bins));
                        if (wave == 0) {
                            unsigned *hist = reinterpret_cast<unsigned *>(smem + IMAGE_BYTES);
                            // bands up to 512 columns (the shipped configs: 409 / 410) keep their keys in registers
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
                        const int half = ((tile_cols + 127) >> 7) << 6;            // first wave's share, whole 64s
                        const int c0 = wave == 2 ? 0 : half;
                        const int c1 = wave == 2 ? (half < tile_cols ? half : tile_cols) : tile_cols;
                        float *dst = a.tile_out + row * (int64_t)tile_cols;
                        if (a.ln_out == nullptr) {
                            for (int c = c0 + lane; c < c1; c += 64) dst[c] = img(tile_first + c);
                        } else {
                            // the viewer's log image of the tile (fits2png:46) and this wave's share of the row's min /
                            // max over the non-zero pixels (:476-477), while the magnitudes are still in LDS
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
                late_loads();
            }
        }
        stamp(11);
        prev_out = a.rows_out + row * a.row_stride;
        prev_bytes = N * 4;
        st_acc[15] += 1;
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
    if constexpr (RO_STAMPS32K) {
        st_acc[14] = __builtin_amdgcn_s_memrealtime();             // ... and end (tools/r3/stamps32k.py)
        if (a.stamps && (tid & 63) == 0)
            for (int k = 0; k < 16; ++k) a.stamps[(blockIdx.x * 16 + wave) * 16 + k] = st_acc[k];
    }
}

struct DevicePlan32k {
    bool ready = false;
    int cus = 0;
};

template <int FMT> static hipError_t launch_fmt(const StftArgs &a, hipStream_t s)
{
    static std::mutex lock;
    static DevicePlan32k table[64];
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    if (dev < 0 || dev >= 64) return hipErrorInvalidDevice;
    int cus;
    {
        std::lock_guard<std::mutex> g(lock);
        DevicePlan32k &d = table[dev];
        if (!d.ready) {
            const void *fn = reinterpret_cast<const void *>(&stft32k_kernel<FMT>);
            if ((e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES)) != hipSuccess) return e;
            if ((e = hipDeviceGetAttribute(&d.cus, hipDeviceAttributeMultiprocessorCount, dev)) != hipSuccess) return e;
            d.ready = true;
        }
        cus = d.cus;
    }
    // persistent grid: one workgroup per CU, a multiple of 8 so every XCD gets the same share, never more than rows
    const int64_t per_xcd = (a.rows + 7) / 8;
    int64_t slots = cus / 8;
    if (a.spare_cus > 0) slots -= a.spare_cus;
    if (slots < 1) slots = 1;
    if (slots > per_xcd) slots = per_xcd;
    StftArgs b = a;
    b.dec = 1;
    b.dec_log2 = 0;
    // The touches park hop * BYTES per resident workgroup in the XCD's 4 MiB L2 for most of a row time; past half of it
    // they push out the rows being transformed and every line is fetched twice.
    b.prefetch = slots * (int64_t)a.hop * Sample<FMT>::BYTES <= (2 << 20) ? 1 : 0;
    b.stagger = 0;
    hipLaunchKernelGGL((stft32k_kernel<FMT>), dim3((unsigned)(slots * 8)), dim3(T), LDS_BYTES, s, b);
    return hipGetLastError();
}

}  // namespace k32

hipError_t launch_stft32k(int fmt, const StftArgs &a, hipStream_t s)
{
    if (a.rows <= 0) return hipSuccess;
    if (a.spec_out != nullptr || a.big_form) return hipErrorInvalidValue;
    if (fmt == RO_FMT_F32) return k32::launch_fmt<RO_FMT_F32>(a, s);
    if (fmt == RO_FMT_I16) return k32::launch_fmt<RO_FMT_I16>(a, s);
    return hipErrorInvalidValue;
}

}  // namespace ro
