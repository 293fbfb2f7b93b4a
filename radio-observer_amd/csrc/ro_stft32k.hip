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
// 1 and pass 2 is a 32 x 32 transposition inside the wave: no workgroup barrier between the last read of exchange 1
// and the completed magnitude image -- two thirds of the row's butterflies.  Five barriers per row.
//
// ONE LDS layout serves the three uses (floats; real and imaginary plane one after the other, a complex row does
// not fit 160 KiB):
//   cell(q, w, l) = 1026 q + 64 w + l        q < 32 a row, w < 16 the wave whose territory it is, l < 64
// * every write is ds_write_addtid_b32 (address = M0 + offset + 4 lane: lane-linear, no address VGPR, twice the
//   rate of ds_write_b32); M0 and the offset field hold 16 bits each and every cell is within their reach;
// * every read is ds_read_b64 -- two neighbouring cells of one row, i.e. the same component of two points (256 B/clk
//   where ds_read_b32 gives 128): that is why passes 1 and 2 keep their points on a PLANAR register layout
//   (ro_fft_planar.h) -- R[i] = (re p, re p'), I[i] = (im p, im p') -- and why the lane maps below put the two points a
//   thread wants together into neighbouring lanes of the writing wave;
// * 1026 = 2 mod 64: a column read (32 rows, two cells each) walks all 64 banks, and 16 rows are 32 banks;
// * a wave only ever READS its own territory until the image is complete, and nobody writes into a territory
//   between the barrier in front of exchange 1's reads and the next row: that is what removes the barriers.
//   pass 0:      lane l of wave w' is column n1 = a + 32 b with a = 2 ((l & 31) >> 1) + (l >> 5), b = 2 w' + (l & 1):
//                lanes l, l + 1 (l even) hold the same a and b, b + 1
//   exchange 1:  slot k0 of that thread  ->  cell(w' + 16 (k0 & 1), k0 >> 1, l)
//                pass-1 lane (a >> 1) + 16 (a & 1) + 32 kb of wave w is thread (k0 = 2 w + kb, a) and reads slots
//                b = 2 w', 2 w' + 1 from cell(w' + 16 kb, w, 2 (a >> 1) + 32 (a & 1)), + 1       (mates b, b + 1)
//   exchange 2:  slot k1 of that thread -> cell(k1, w, lane)
//                pass-2 lane l' of wave w is thread (k0 = 2 w + (l' & 1), k1 = ((l' >> 1) + 4 (w >> 1)) & 31) and reads
//                slots a = 4 u + p, 4 u + p + 2 from cell(k1, w, 2 u + 16 p + 32 kb), + 1       (mates a, a + 2)
//   image:       slot k2 of that thread = bin k0 + 32 k1 + 1024 k2 -> cell(k2, w, l'): bins k0 = 2 w, 2 w + 1 of one
//                (k1, k2) are neighbours, and the rotation by 4 (w >> 1) makes the 16-byte-per-lane read-back of the
//                row (two ds_read_b64: territories w, w + 1) conflict-free
// tools/r4/emu32k.py restates these maps and the planar butterflies (with the VOP3P modifiers of ro_fft_planar.h) in
// numpy and checks them against numpy's FFT and for bank conflicts.
// the butterflies' scheduling leash as a scheduling barrier, not an empty asm statement (see tie() in ro_fft_device.h):
// hipcc pads an s_nop around every inline-asm result on gfx950; 240 fewer s_nop per row and wave, 7 VGPRs fewer
#define RO_TIE_SCHED 1
#include "ro_kernels.h"
#include "ro_fft_device.h"
#include "ro_fft_planar.h"
#include "ro_device_util.h"
#include "ro_k32_lds.h"

#include <mutex>

// The ONE diagnostic switch of this file.  A -DRO_DIAG=1 build (tools/ab_build.sh) may set RO_STAMPS32K=1: s_memtime
// deltas per phase of the row loop (every wave of every workgroup) accumulated into StftArgs::stamps
// (tools/r3/stamps32k.py).  Never timed, never shipped.
#if !defined(RO_DIAG) || !defined(RO_STAMPS32K)
#undef RO_STAMPS32K
#define RO_STAMPS32K 0
#endif

namespace ro {
namespace k32 {

// legs of the next row requested from inside pass 2's last level, two per unit (of 16); the rest behind the scan
constexpr int PIPE_UNITS = 6;
// share of the next row's window coefficients requested right behind the window stage (quads of 8)
constexpr int WIN_EARLY = 2;

// column of thread position t in pass 0 (see the header); the window table is laid out with it (stft32k_window_layout)
__host__ __device__ constexpr int column(int t)
{
    const int l = t & 63, j = l & 31;
    return (t & ~63) + 2 * ((j >> 1) + 16 * (j & 1)) + (l >> 5);
}

// column c of the fft-shifted row in the LDS image (the band scan's view of it)
struct ImageRow {
    const float *img;
    __device__ __forceinline__ float operator()(int c) const
    {
        const int k = (c + N / 2) & (N - 1), r = k >> 10, beta = k & 1023;
        const int w = (beta & 31) >> 1, kb = beta & 1, k1 = beta >> 5;
        return img[RQ * r + 64 * w + 2 * ((k1 - 4 * (w >> 1)) & 31) + kb];
    }
};

// levels 1..3 of dit<32> in level order behind a level 0 the caller has done itself, with the hook of dit32_head
template <typename F> __device__ __forceinline__ void dit32_levels123(v2f *v, F hook)
{
    const v2f *tok = &v[31];
    dit_level_order<1>(v, tok, std::make_integer_sequence<int, 2>{});
    hook(std::integral_constant<int, 1>{});
    dit_level_order<2>(v, tok, std::make_integer_sequence<int, 4>{});
    hook(std::integral_constant<int, 2>{});
    dit_level_order<3>(v, tok, std::make_integer_sequence<int, 8>{});
    hook(std::integral_constant<int, 3>{});
}

template <int FMT> __global__ __launch_bounds__(T, 1) void stft32k_kernel(StftArgs a)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    using S = Sample<FMT>;
    using namespace planar;

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
    // v_permlane32_swap gives every lane one whole column, column(t).
    const int po = (column(tid & ~32) + ((tid >> 5) & 1) * H * (N / 32)) * S::BYTES;
    auto row_rsrc = [&](int64_t k, bool valid) {
        return make_rsrc(iq + (a.first_row + k) * (int64_t)a.hop * S::BYTES, valid ? (unsigned)N * S::BYTES : 0u);
    };
    {
        const __amdgpu_buffer_rsrc_t rs = row_rsrc(row, true);
#pragma unroll
        for (int k = 0; k < H; ++k) S::load_pair(rs, po, k * (N / 32) * S::BYTES, v[k], v[H + k]);
    }
    // window coefficients in the kernel's own order (StftArgs::window_k32, stft32k_window_layout): quad q of a thread =
    // the coefficients of legs {2q, 2q + 16, 2q + 1, 2q + 17} of its column -- the two butterflies (2q, 2q + 16) and
    // (2q + 1, 2q + 17) of pass 0's first level, which takes the window multiply in
    v4f w4[H / 2];
    auto load_window = [&](const __amdgpu_buffer_rsrc_t &rs_win, auto first_c, auto last_c) {
        constexpr int first = decltype(first_c)::value, last = decltype(last_c)::value;
#pragma unroll
        for (int q = first; q < last; ++q) {
            const u32x4 t = __builtin_amdgcn_raw_buffer_load_b128(rs_win, tid * 16, q * T * 16, 0);
            w4[q] = (v4f){__uint_as_float(t.x), __uint_as_float(t.y), __uint_as_float(t.z), __uint_as_float(t.w)};
        }
    };
    using c0 = std::integral_constant<int, 0>;
    using cE = std::integral_constant<int, WIN_EARLY>;
    using cN = std::integral_constant<int, H / 2>;
    auto win_rsrc = [&](bool valid) { return make_rsrc(a.window_k32, valid ? N * 4 : 0); };
    load_window(win_rsrc(true), c0{}, cN{});

    // stage twiddles: {w, w^2} {w^4, w^8} {w^16, -} of butterfly k from the packed table (three 16-byte loads);
    // the levels make the other powers.  Pass 1: k = k0 (32 entries at unit 0), pass 2: k = k0 + 32 k1 (1024 at unit 96).
    v2f tw1[5], tw2[5];
    auto tw_load = [&](v2f (&t)[5], int k, auto pk_c, auto ns_c) {
        constexpr int PK = decltype(pk_c)::value, NS = decltype(ns_c)::value;
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            const u32x4 u = __builtin_amdgcn_raw_buffer_load_b128(rs_twk, k * 16, (PK + q * NS) * 16, 0);
            t[2 * q] = (v2f){__uint_as_float(u.x), __uint_as_float(u.y)};
            if (q < 2) t[2 * q + 1] = (v2f){__uint_as_float(u.z), __uint_as_float(u.w)};
        }
    };

    [[maybe_unused]] unsigned long long st_acc[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, st_prev = 0;
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
    // waves 2 (m & 7) (bins 4 m, 4 m + 1: neighbouring cells) and 2 (m & 7) + 1; out as 1 KiB per wave-instruction; bin k
    // leaves for column (k + N/2) mod N (src/WaterfallBackend.cpp:492-505)
    // (ONE register holds the thread's chunk-0 address for the whole kernel; laundered per chunk, else hipcc keeps all
    // eight chunk addresses alive through the row)
    int rb_base = RQ * (tid >> 8) + 128 * (tid & 7) + 2 * ((((tid & 255) >> 3) - 4 * (tid & 7)) & 31);
    auto store_chunk = [&](int q, const __amdgpu_buffer_rsrc_t &rs) {
        int rb = rb_base;
        asm volatile("" : "+v"(rb));
        lds_vpair *p = (lds_vpair *)(lds + rb + 4 * RQ * q);
        const v2f x01 = p[0], x23 = p[32];
        buf_store_f4(x01.x, x01.y, x23.x, x23.y, rs, tid * 16, ((q * T * 4 + N / 2) & (N - 1)) * 4);
    };

    const unsigned ma = (unsigned)wave * (4u * RQ), mb = ma + (unsigned)XB;   // exchange 1: M0 of the even / odd slots
    const unsigned mc = (unsigned)wave * 256u, md = mc + (unsigned)HB;        // own territory: M0 of rows < 16 / >= 16

    for (;;) {
        const __amdgpu_buffer_rsrc_t rs_prev = make_rsrc(prev_out, prev_bytes);
        // ---- every lane gets its column (the samples were requested a whole epilogue ago)
        {
            const v2f gain2 = (v2f){0.0f, a.gain};          // src/FFTBackend.cpp:78-79: Q += gain
            if (a.gain != 0.0f) {                           // every shipped config has iq_gain = 0: skip the adds
#pragma unroll
                for (int i = 0; i < 32; ++i) v[i] = v[i] + gain2;
            }
#pragma unroll
            for (int k = 0; k < H; ++k) {
                v2f &lo = v[k], &hi = v[H + k];
                // lanes 0..31 hold legs k, lanes 32..63 legs 16 + k of both columns: the upper half of slot k trades
                // places with the lower half of slot 16 + k
                const auto rx = __builtin_amdgcn_permlane32_swap(__float_as_uint(lo.x), __float_as_uint(hi.x), false, false);
                const auto ry = __builtin_amdgcn_permlane32_swap(__float_as_uint(lo.y), __float_as_uint(hi.y), false, false);
                lo = (v2f){__uint_as_float(rx[0]), __uint_as_float(ry[0])};
                hi = (v2f){__uint_as_float(rx[1]), __uint_as_float(ry[1])};
            }
        }
        asm volatile("" ::"v"(touch));                       // see touch_next
        // ---- pass 0, level 0 with the window multiply in it (src/FFTBackend.cpp:229-232):
        //   (x_i w_i + x_{i+16} w_{i+16},  x_i w_i - x_{i+16} w_{i+16})  =  t = x_i w_i;  fma(x_{i+16}, +-w_{i+16}, t)
        // three packed operations where multiply, multiply, add, subtract are four
#pragma unroll
        for (int i = 0; i < H; ++i) {
            const v4f c4 = w4[i / 2];
            const v2f wi = (i & 1) ? c4.zz : c4.xx, wj = (i & 1) ? c4.ww : c4.yy;
            const v2f t = v[i] * wi;
            const v2f s = __builtin_elementwise_fma(v[H + i], wj, t);
            v[H + i] = __builtin_elementwise_fma(v[H + i], -wj, t);
            v[i] = s;
            if (i & 1) planar::leash();
        }
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

        // ---- pass 0, levels 1..3; the previous row's image goes out between them (LDS read-back + 16-byte stores)
        store_chunk(0, rs_prev);
        store_chunk(1, rs_prev);
        dit32_levels123(v, [&](auto hc) {
            constexpr int h = decltype(hc)::value;
            store_chunk(2 * h, rs_prev);
            store_chunk(2 * h + 1, rs_prev);
        });
        int lane;
        {
            int lt = tid;
            asm volatile("" : "+v"(lt));
            lane = lt & 63;
        }
        {
            // pass-1 twiddles depend on k0 only -- two values per wave (lanes < 32: k0 = 2 w, the others 2 w + 1): scalar
            // loads (the scalar cache, not the vector memory pipe, which at this point is busy with the row's samples)
            // and one select per register
            const bool odd = lane >= 32;
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
        }
        stamp(1);
        wg_sync();                              // (a) every wave has read its part of the old image: LDS is free
        stamp(2);
        // ---- pass 0, last level: the x plane of exchange 1 leaves as the pairs finish
        dit32_last(v, [&](auto jc) {
            constexpr int j = decltype(jc)::value;
            x1_write_pair<bitrev<32>(2 * j)>(ma, mb, v[2 * j].x, v[16 + 2 * j].x, v[2 * j + 1].x, v[17 + 2 * j].x);
            return v[17 + 2 * j].y;
        });
        stamp(3);
        // ---- exchange 1, the rest: thread (k0, a) of pass 1 reads slots b = 2 w', 2 w' + 1 (one ds_read_b64) from its own
        // territory, for the sixteen writer waves w'
        v2f R[16], I[16];
        {
            lds_vpair *g1 = (lds_vpair *)(lds + RQ * 16 * (lane >> 5) + 64 * wave + 2 * (lane & 15) + 32 * ((lane >> 4) & 1));
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");           // the x-plane writes (hipcc does not count them)
            __builtin_amdgcn_s_barrier();                                // (b)
            asm volatile("" ::: "memory");
#pragma unroll
            for (int i = 0; i < 16; ++i) R[i] = g1[(RQ / 2) * i];
            wg_sync();                                                   // (c) everyone has its x: the plane may go
            x1_write_plane(ma, mb, [&](int k0) { return v[bitrev<32>(k0)].y; });
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();                                // (d)
            asm volatile("" ::: "memory");
#pragma unroll
            for (int i = 0; i < 8; ++i) {                                // in the order level 0 of pass 1 pairs them
                I[i] = g1[(RQ / 2) * i];
                I[i + 8] = g1[(RQ / 2) * (i + 8)];
            }
        }
        stamp(4);
        // ---- pass 1 (mates b, b + 1: planar<0>).  From here to the completed image the wave is on its own.
        planar::head<0>(R, I, tw1[4], tw1[3], tw1[2], tw1[1]);
        touch_next();
        int k1p, kbp;                                                    // this thread in pass 2: (k0 = 2 wave + kbp, k1 = k1p)
        {
            k1p = ((lane >> 1) + 4 * (wave >> 1)) & 31;
            kbp = lane & 1;
            tw_load(tw2, 2 * wave + kbp + 32 * k1p, std::integral_constant<int, 96>{}, std::integral_constant<int, 1024>{});
        }
        stamp(5);
        planar::last0(R, I, tw1[0], [&](auto jc) {
            // positions 2j, 2j + 1 (pair j) and 16 + 2j, 17 + 2j (pair 8 + j) are final: results k1 = q, q + 16, q + 1,
            // q + 17 with q = bitrev32(2j); their real parts leave for exchange 2
            constexpr int j = decltype(jc)::value, q = bitrev<32>(2 * j);
            own_write4<q, q + 1, q + 16, q + 17>(mc, md, R[j].x, R[8 + j].x, R[j].y, R[8 + j].y);
        });
        stamp(6);
        // ---- exchange 2: a 32 x 32 transposition inside each half of the wave, through the wave's own rows.  One
        // wave's LDS instructions execute in order: no wait between its writes and its reads of the same cells.
        v2f R2[16], I2[16];
        {
            lds_vpair *g2 = (lds_vpair *)(lds + RQ * k1p + 64 * wave + 32 * kbp);
            asm volatile("" ::: "memory");
#pragma unroll
            for (int i = 0; i < 16; ++i) R2[i] = g2[(i >> 1) + 8 * (i & 1)];          // pair 2u + p: cells 2u + 16p, + 1
            asm volatile("" ::: "memory");
            own_write_plane(mc, md, [&](int k1) {
                constexpr int MB = 0;
                const int p = bitrev<32>(k1);
                return hf<MB>(p) ? I[pr<MB>(p)].y : I[pr<MB>(p)].x;
            });
#pragma unroll
            for (int i = 0; i < 16; ++i) I2[i] = g2[(i >> 1) + 8 * (i & 1)];
        }
        stamp(7);
        // ---- pass 2 (mates a, a + 2: planar<1>)
        planar::head<1>(R2, I2, tw2[4], tw2[3], tw2[2], tw2[1]);
        stamp(8);
        {
            // Last level with the epilogue folded in.  After unit u positions 4u .. 4u + 3 are final = bins
            // k0 + 32 k1 + 1024 k2 for k2 = r, r + 8 (pair 2u) and r + 16, r + 24 (pair 2u + 1), r = bitrev8(u): their
            // magnitudes go to those rows of the wave's territory, and the eight freed registers receive legs 2u, 2u + 1
            // of the NEXT row's samples -- for u < PIPE_UNITS; the last legs are requested behind the scan.
            const __amdgpu_buffer_rsrc_t rs_next = row_rsrc(has_next ? next : row, has_next);   // zero-sized after the last row
            // The image writes of unit u are issued one unit late (from done(u + 1), the last ones behind the level):
            // v_sqrt_f32 runs in the transcendental pipe and hipcc pads no hazards in front of inline asm.
            v2f pma = {0.f, 0.f}, pmb = {0.f, 0.f};
            planar::last1(R2, I2, tw2[0], [&](auto uc) {
                constexpr int u = decltype(uc)::value;
                // |X| = sqrt(re^2 + im^2) (src/WaterfallBackend.cpp:497-503), two bins per packed operation
                auto mag = [](v2f re, v2f im) {
                    const v2f s = __builtin_elementwise_fma(im, im, re * re);
                    return (v2f){__builtin_amdgcn_sqrtf(s.x), __builtin_amdgcn_sqrtf(s.y)};
                };
                const v2f m_a = mag(R2[2 * u], I2[2 * u]), m_b = mag(R2[2 * u + 1], I2[2 * u + 1]);
                if constexpr (u > 0) {
                    constexpr int r = bitrev<8>(u > 0 ? u - 1 : 0);
                    own_write4<r, r + 8, r + 16, r + 24>(mc, md, pma.x, pma.y, pmb.x, pmb.y);
                }
                pma = m_a;
                pmb = m_b;
                if constexpr (u < PIPE_UNITS) {
                    const int pj = after(po, m_b.y);    // the loads may not start before these magnitudes exist
                    S::load_pair(rs_next, pj, (2 * u) * (N / 32) * S::BYTES, v[2 * u], v[H + 2 * u]);
                    S::load_pair(rs_next, pj, (2 * u + 1) * (N / 32) * S::BYTES, v[2 * u + 1], v[H + 2 * u + 1]);
                }
            });
            asm volatile("s_nop 7\n\ts_nop 7" ::: "memory");      // the last square roots: see above
            {
                constexpr int r = bitrev<8>(7);
                own_write4<r, r + 8, r + 16, r + 24>(mc, md, pma.x, pma.y, pmb.x, pmb.y);
            }
            // The legs the last level did not request and the rest of the window coefficients.  The waves reach this
            // point up to ~6k cycles apart (the oldest wave of a SIMD first) and then wait for the barrier: whoever is
            // not about to scan asks NOW, so the memory pipe works through most of the next row's 384 KiB while the
            // younger waves are still in their butterflies, instead of starting behind the barrier with all 16 waves
            // in its queue.  (The two scanning waves need these registers for their band.)
            // who has work on the image behind the barrier: waves 0, 1 scan, waves 2, 3 cut the band tile
            const bool image_work = (a.records != nullptr && wave < 2) || (a.tile_out != nullptr && (wave == 2 || wave == 3));
            auto late_loads = [&]() {
#pragma unroll
                for (int k = 2 * PIPE_UNITS; k < H; ++k)
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
                    int sl = tid & 63;
                    asm volatile("" : "+v"(sl));
                    const ImageRow img{lds};
                    if (a.records != nullptr && wave < 2) {
                        int low_noise = a.low_noise, noise_width = a.noise_width, low_detect = a.low_detect;
                        int detect_width = a.detect_width, avg_bins = a.avg_bins;
                        asm volatile("" : "+s"(low_noise), "+s"(noise_width), "+s"(low_detect), "+s"(detect_width), "+s"(avg_bins));
                        if (wave == 0) {
                            unsigned *hist = reinterpret_cast<unsigned *>(smem + IMAGE_BYTES);
                            // bands up to 512 columns (the shipped configs: 409 / 410) keep their keys in registers
                            const float nz = noise_width <= 512 ? scan_noise<8>(img, low_noise, noise_width, hist, sl)
                                                                : scan_noise<0>(img, low_noise, noise_width, hist, sl);
                            if (sl == 0) a.records[row].noise = nz;
                        } else {
                            const int pk = scan_peak<8>(img, low_detect, detect_width, sl);
                            const float av = scan_average(img, low_detect + pk - avg_bins / 2, avg_bins, N, sl);
                            if (sl == 0) {
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
                            for (int c = c0 + sl; c < c1; c += 64) dst[c] = img(tile_first + c);
                        } else {
                            // the viewer's log image of the tile (fits2png:46) and this wave's share of the row's min /
                            // max over the non-zero pixels (:476-477), while the magnitudes are still in LDS
                            float *ldst = a.ln_out + row * (int64_t)tile_cols;
                            unsigned kmin = 0xffffffffu, kmax = 0u;
                            for (int c = c0 + sl; c < c1; c += 64) {
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
                            if (sl == 0) {
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
        if constexpr (RO_STAMPS32K) st_acc[15] += 1;
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

// StftArgs::window_k32 from the natural table: thread t of the kernel reads 16 bytes at (q T + t) 16, q < 8 =
// {w[c + 1024 (2q)], w[c + 1024 (2q + 16)], w[c + 1024 (2q + 1)], w[c + 1024 (2q + 17)]}, c = its column
void stft32k_window_layout(const float *w, float *out)
{
    using namespace k32;
    for (int t = 0; t < T; ++t) {
        const int c = column(t);
        for (int q = 0; q < H / 2; ++q) {
            float *o = out + ((size_t)q * T + t) * 4;
            o[0] = w[c + 1024 * (2 * q)];
            o[1] = w[c + 1024 * (2 * q + 16)];
            o[2] = w[c + 1024 * (2 * q + 1)];
            o[3] = w[c + 1024 * (2 * q + 17)];
        }
    }
}

hipError_t launch_stft32k(int fmt, const StftArgs &a, hipStream_t s)
{
    if (a.rows <= 0) return hipSuccess;
    if (a.spec_out != nullptr || a.big_form || a.window_k32 == nullptr) return hipErrorInvalidValue;
    if (fmt == RO_FMT_F32) return k32::launch_fmt<RO_FMT_F32>(a, s);
    if (fmt == RO_FMT_I16) return k32::launch_fmt<RO_FMT_I16>(a, s);
    return hipErrorInvalidValue;
}

}  // namespace ro
