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
#include "ro_device_util.h"
#include "ro_f64_device.h"

#include <cstdlib>
#include <mutex>
#include <type_traits>

// The ONE diagnostic switch of this file.  A -DRO_DIAG=1 build (tools/ab_build.sh) may set RO_STAMPS=1 (s_memtime
// deltas per phase of the row loop, written to StftArgs::stamps by lane 0 of wave 0 of every workgroup; never timed:
// the fences change the overlaps) and gets the run-time knobs of launch_plan (RO_SLOTS, RO_STAGGER, RO_PREFETCH).
// The product build has neither.  (Round 1 and 2's other build-time switches -- ablations, the exchange and load
// variants -- are gone: what they measured is in profiles/r01_ablation.txt and r02_ab_attempts.txt, and the choices
// they settled are the constants below.)
#ifdef RO_DIAG
#ifndef RO_STAMPS
#define RO_STAMPS 0
#endif
#define RO_DIAG_KNOBS 1
#ifndef RO_ROW_PRIO
#define RO_ROW_PRIO 1                 // -DRO_ROW_PRIO=0: the A/B baseline of profiles/r03_row_priority.txt
#endif
// the round-3 experiment tools/r3/ro_stft_wl.hip (stft32k_kernel's structure at N = 16384 / 8192; tools/r3/ab_wl_build.sh
// adds it to the build): correct, and no faster than this file's generic loop (profiles/r03_ab_wl.txt)
#ifndef RO_USE_WL
#define RO_USE_WL 0
#endif
#ifndef RO_WL_FUSE16
#define RO_WL_FUSE16 1
#endif
#ifndef RO_WL_FUSE8
#define RO_WL_FUSE8 0
#endif
#else
#define RO_STAMPS 0
#define RO_ROW_PRIO 1
#define RO_USE_WL 0
#define RO_WL_FUSE16 0
#define RO_WL_FUSE8 0
#endif

namespace ro {

// 16-byte sample loads shared by lane pairs (see load_row) instead of one 8-byte load per sample
constexpr bool RO_PAIRED_LOADS = true;
// window coefficients in the kernel's own order (16-byte loads, see stft_window_layout) instead of the natural table
constexpr bool RO_WIN_PERM = true;
// Plans with registers to spare (N <= 4096: LDS, not VGPRs, limits their occupancy) keep their window coefficients
// and stage twiddles in registers for the whole persistent loop instead of re-reading them from L2 for every row --
// per row only the samples come in and the magnitudes go out (14-35 % faster).
constexpr bool RO_RESIDENT_TABLES = true;
// add-TID plans: lanes l and l+32 share their sample columns and trade halves with v_permlane32_swap_b32 (1 VALU op
// per register) instead of lanes l and l^1 with a DPP move + select (2 ops)
constexpr bool RO_SWAP32 = true;
// The hop new samples of the workgroup's NEXT row are touched (one dword per 128-byte line, value unused) well
// before the epilogue asks for them: they come from HBM, every other byte of the row from L2, and that one miss
// latency sat on the critical path of every row.  Where: 1 = after the window stage, 2 = after the first exchange
// (nothing queues behind the misses there; measured 5 % faster than 1), 3 = after the second.
constexpr int RO_PREFETCH_NEXT = 2;
// the N = 32768 (complex spectra and the one-kernel large transform; magnitude rows run on ro_stft32k.hip), 16384 and
// 8192 plans exchange through ds_write_addtid_b32 (8192: split planes, tables re-read per row, four workgroups of 256
// threads per CU instead of two with window and twiddles resident in 234 VGPRs)
constexpr bool RO_USE_ADDTID = true, RO_ADDTID_16384 = true, RO_ADDTID_8192 = true;
// share (percent) of the next row's window coefficients that is prefetched across the transform (the 1024-thread plan
// has registers for a quarter only)
constexpr int RO_WIN_EARLY_PCT = 50;

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
    // packed table (16-byte units): stage s with radix >= 16 holds 3 x NS units, unit q*NS + k = the q-th pair of
    // twiddles of butterfly k: radix 32 {w,w^2} {w^4,w^8} {w^16,-}; radix 16 {w,w^2} {w^3,w^4} {w^8,w^12}
    static constexpr int pkq(int r) { return (r == 32 || r == 16) ? 3 : 0; }
    static constexpr int PK1 = 0;
    static constexpr int PK2 = PK1 + pkq(R1) * NS1;
    static constexpr int PK3 = PK2 + pkq(R2) * NS2;
    static constexpr int PK_TOTAL = PK3 + pkq(R3) * NS3;
    static constexpr int LDS_ELEMS = N + N / 32;                    // padded
    static constexpr int LDS_BYTES = LDS_ELEMS * (SPLIT ? 4 : 8);
    static_assert(R0 * R1 * R2 * R3 == N, "radices must multiply to N");
    static_assert(P % R0 == 0 && P % R1 == 0 && P % R2 == 0 && P % R3 == 0, "radix must divide P");
};

// the N = 32768 and 16384 plans exchange through ds_write_addtid_b32 (see exchange_addtid): N / 32 "logical threads",
// one radix-32 butterfly each in the first two stages (T threads run P / 32 of them each: 1024 x 1, 512 x 2 or -- N =
// 16384 -- 512 x 1), a last stage of radix N / 1024
template <class PL> constexpr bool plan_addtid()
{
    return RO_USE_ADDTID && (PL::N == 32768 || (PL::N == 16384 && RO_ADDTID_16384) || (PL::N == 8192 && RO_ADDTID_8192)) &&
           PL::T * (PL::P / 32) == PL::N / 32 && PL::T % 64 == 0 && PL::R0 == 32 && PL::R1 == 32 &&
           PL::R2 == PL::N / 1024 && PL::R3 == 1 && PL::SPLIT;
}
template <class PL> constexpr bool plan_swap32() { return plan_addtid<PL>() && RO_SWAP32 && RO_PAIRED_LOADS; }
// dynamic LDS of a plan: the exchange image
template <class PL> constexpr int plan_lds_bytes() { return PL::LDS_BYTES; }

// Paired sample loads: which stage-0 column a thread transforms, and the first sample it fetches.
// Lanes l, l^1 (default) or l, l+32 (swap32) fetch the SAME two adjacent columns with 16-byte loads, one lane
// for legs [0, R0/2), the other for legs [R0/2, R0), then trade halves so each ends up with one whole column.
template <class PL> __host__ __device__ constexpr int plan_column(int tid)
{
    if constexpr (plan_swap32<PL>()) return (tid & ~63) + 2 * (tid & 31) + ((tid >> 5) & 1);
    else return tid;
}
template <class PL> __host__ __device__ constexpr int plan_pair_off(int tid)
{
    const int c = plan_column<PL>(tid);
    return (c & ~1) + (c & 1) * (PL::R0 / 2) * (PL::N / PL::R0);
}

// ---------------------------------------------------------------------------
// stage helpers (all indices compile-time after unrolling -> v[] stays in VGPRs)
// ---------------------------------------------------------------------------

// autosort scatter of a finished stage (R, NS) into LDS (element index, padded).
// The padded index i + (i>>5) is affine in r when r*NS never carries into bit 5 on
// its own (NS a multiple of 32) or for the first stage of radix 32 (i = 32*j + r):
// then one base VGPR + immediate offsets address the whole scatter.  Otherwise each
// element computes its own address.
template <int P, int T, int R, int NS, typename E, typename F>
__device__ __forceinline__ void lds_scatter(E *lds, const v2f (&v)[P], int tid, F pick)
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
__device__ __forceinline__ void lds_gather(const E *lds, v2f (&v)[P], int tid, F put)
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
template <class PL, int RA, int NS, int RB, typename ST>
__device__ __forceinline__ void exchange(void *smem, v2f (&v)[PL::P], int tid, ST sub)
{
    constexpr int N = PL::N, P = PL::P, T = PL::T;
    if constexpr (PL::SPLIT) {
        float *lds = reinterpret_cast<float *>(smem);
        lds_scatter<P, T, RA, NS>(lds, v, tid, [](v2f e) { return e.x; });
        sub(0);
        wg_sync();
        sub(1);
        lds_gather<N, P, T, RB>(lds, v, tid, [](v2f &d, float s) { d.x = s; });
        sub(2);
        wg_sync();
        sub(3);
        lds_scatter<P, T, RA, NS>(lds, v, tid, [](v2f e) { return e.y; });
        sub(0);
        wg_sync();
        sub(1);
        lds_gather<N, P, T, RB>(lds, v, tid, [](v2f &d, float s) { d.y = s; });
        sub(2);
        wg_sync();
        sub(3);
    } else {
        v2f *lds = reinterpret_cast<v2f *>(smem);
        lds_scatter<P, T, RA, NS>(lds, v, tid, [](v2f e) { return e; });
        wg_sync();
        lds_gather<N, P, T, RB>(lds, v, tid, [](v2f &d, v2f s) { d = s; });
        wg_sync();
    }
}

// ---------------------------------------------------------------------------
// N = 32768 exchange with ds_write_addtid_b32.  In-kernel stamps showed that four fifths of an
// exchange is spent waiting for the LDS WRITES (ds_write_b32 moves address + data VGPRs to the
// LDS at 4 cycles per wave-instruction; 128 KiB per plane took ~3.7k cycles), the gathers being
// cheap.  ds_write_addtid_b32 has no address VGPR (address = M0 + offset + 4*lane, 2 cycles per
// wave-instruction), but needs a lane-linear image.  Both exchanges have one:
//   after stage 0 (element i = 32 j + r):           image[r*1025 + j]       gather (j'&31)*1025 + (j'>>5) + 32 r'
//   after stage 1 (i = (j>>5)*1024 + (j&31) + 32r): image[r*1024 + j]       gather (j'>>5)*1024 + (j'&31) + 32 r'
// (j = writing thread, r = its register slot; j', r' = reading thread / slot).  Writes are
// lane-linear, reads hit 32 consecutive banks per half-wave: no conflicts either way.
// M0 holds 16 bits and the offset field 16 bits, so slots 0..15 and 16..31 use two M0 values.
// ---------------------------------------------------------------------------
// scatter 32 floats per lane, slot q (value f(q), q a literal after inlining) to byte q*ROWB + 4*tid
template <int ROWB, typename F> __device__ __forceinline__ void addtid_scatter32(unsigned wave_bytes, F f)
{
    // second half: M0 = wave_bytes + HB, offsets q*ROWB - HB in [0, 65535]; HB chosen so both fit 16 bits
    constexpr int HB = (31 * ROWB - 65532 + 3) / 4 * 4 > 0 ? ((31 * ROWB - 65532 + 3) / 4) * 4 : 0;
    static_assert(HB + 3840 <= 65532 && 31 * ROWB - HB <= 65535 && 16 * ROWB - HB >= 0, "M0 / offset split");
    addtid_write8<0 * ROWB, 1 * ROWB, 2 * ROWB, 3 * ROWB, 4 * ROWB, 5 * ROWB, 6 * ROWB, 7 * ROWB>(
        wave_bytes, f(0), f(1), f(2), f(3), f(4), f(5), f(6), f(7));
    addtid_write8<8 * ROWB, 9 * ROWB, 10 * ROWB, 11 * ROWB, 12 * ROWB, 13 * ROWB, 14 * ROWB, 15 * ROWB>(
        wave_bytes, f(8), f(9), f(10), f(11), f(12), f(13), f(14), f(15));
    addtid_write8<16 * ROWB - HB, 17 * ROWB - HB, 18 * ROWB - HB, 19 * ROWB - HB, 20 * ROWB - HB, 21 * ROWB - HB,
                  22 * ROWB - HB, 23 * ROWB - HB>(wave_bytes + HB, f(16), f(17), f(18), f(19), f(20), f(21), f(22),
                                                  f(23));
    addtid_write8<24 * ROWB - HB, 25 * ROWB - HB, 26 * ROWB - HB, 27 * ROWB - HB, 28 * ROWB - HB, 29 * ROWB - HB,
                  30 * ROWB - HB, 31 * ROWB - HB>(wave_bytes + HB, f(24), f(25), f(26), f(27), f(28), f(29), f(30),
                                                  f(31));
}

// XCH = 1: between stage 0 and 1, XCH = 2: between stage 1 and 2.  TL = N / 32 logical threads (1024 or 512), radix 32
// in stages 0 and 1, R2 = TL / 32 in stage 2 (32 / R2 butterflies per logical thread).  With S = TL / 32:
//   after stage 0 (element i = 32 j + r):               image[r*(TL+1) + j]   gather (j'&31)*(TL+1) + (j'>>5) + S r'
//   after stage 1 (i = (j>>5)*1024 + (j&31) + 32 r):    image[r*TL + j]       gather ((j'>>5) + S b)*TL + (j'&31) + 32 r'
//                                                                             into slot b R2 + r'  (b < 32 / R2, r' < R2)
// With swap32 pairing the stage-0 thread at position t of a row holds column (t&~63) + 2(t&31) + ((t>>5)&1), so
// column c sits at (c&~63) + ((c&63)>>1) + 32 (c&1): column q + S r' at 32 (q&1) + (q>>1) + 64 ((S r')>>6) + ((S r')&63)/2
// -- still one base + a literal per slot.
template <int XCH, bool SWAP32, int NB, int T, typename ST>
__device__ __forceinline__ void exchange_addtid(void *smem, v2f (&v)[32 * NB], int tid, ST sub)
{
    constexpr int TL = T * NB, S = TL / 32, R2 = S;
    constexpr int ROW = XCH == 1 ? TL + 1 : TL;                   // floats per register-slot row of the image
    constexpr bool PERM = XCH == 1 && SWAP32;
    static_assert(S % 2 == 0 && 32 % R2 == 0, "add-TID exchange: 64 | TL");
    const float *lds = reinterpret_cast<const float *>(smem);
    // logical thread j = tid + T*b (b < NB) owns registers v[32b .. 32b+31]
    auto wave_bytes = [&](int b) { return (unsigned)__builtin_amdgcn_readfirstlane((tid + T * b) >> 6) * 256u; };
    auto gbase = [&](int b) {
        const int j = tid + T * b, q = j >> 5;
        return lds + (XCH == 1 ? (j & 31) * (TL + 1) + (PERM ? 32 * (q & 1) + (q >> 1) : q) : (j >> 5) * TL + (j & 31));
    };
    // float offset of gather slot r from gbase
    auto goff = [](int r) constexpr {
        if (XCH == 1) return PERM ? 64 * ((S * r) >> 6) + (((S * r) & 63) >> 1) : S * r;
        return (r / R2) * S * TL + 32 * (r % R2);
    };
    // volatile: keeps the 32 gathers single ds_read_b32 -- merged into ds_read2_b32 they come back as register
    // pairs of one plane and cost a v_mov each to interleave with the other plane (96 VALU ops per row)
    typedef const volatile __attribute__((address_space(3))) float lds_vfloat;
#pragma unroll
    for (int b = 0; b < NB; ++b)
        addtid_scatter32<ROW * 4>(wave_bytes(b), [&](int q) { return v[32 * b + bitrev<32>(q)].x; });
    sub(0);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    sub(1);
#pragma unroll
    for (int b = 0; b < NB; ++b) {
        lds_vfloat *gv = (lds_vfloat *)gbase(b);
#pragma unroll
        for (int r = 0; r < 32; ++r) v[32 * b + r].x = gv[goff(r)];
    }
    sub(2);
    wg_sync();
    sub(3);
#pragma unroll
    for (int b = 0; b < NB; ++b)
        addtid_scatter32<ROW * 4>(wave_bytes(b), [&](int q) { return v[32 * b + bitrev<32>(q)].y; });
    sub(0);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    sub(1);
#pragma unroll
    for (int b = 0; b < NB; ++b) {
        lds_vfloat *gv = (lds_vfloat *)gbase(b);
#pragma unroll
        for (int r = 0; r < 32; ++r) v[32 * b + r].y = gv[goff(r)];
    }
    sub(2);
    wg_sync();
    sub(3);
}

// ---------------------------------------------------------------------------
// the STFT kernel
// ---------------------------------------------------------------------------
// One workgroup per CU walks over its rows (persistent): no workgroup relaunch between
// rows, the row stores drain while the next row is being loaded, and the next row's
// samples are requested from inside the epilogue, each into the register whose
// magnitude has just been stored.
// MODE 0: magnitude rows (the waterfall).  MODE 1: the complex spectrum itself, bin k at element k of the row
// (what fftw_execute leaves in out_ and FFTBackend::processFFT receives, src/FFTBackend.h:104): same transform, the
// epilogue stores v[] as it is -- no magnitude, no shift, no LDS staging.  MODE 3: a large transform in one kernel
// (below).
// waves per SIMD the register allocation must leave room for.  MODE 3 wants two of its 512-thread workgroups on a CU
// (4 waves per SIMD, 128 VGPRs): one sums its blocks -- loads -- while the other runs its butterflies.
constexpr int RO_DIF_WAVES = 4, RO_MINW8192 = 3;
template <class PL, int FMT, int MODE> constexpr int plan_min_waves()
{
    return MODE == 3 ? RO_DIF_WAVES : (PL::N == 8192 && plan_addtid<PL>()) ? RO_MINW8192 : 1;
}

template <class PL, int FMT, int MODE>
__global__ __launch_bounds__(PL::T, (plan_min_waves<PL, FMT, MODE>())) void stft_kernel(StftArgs a)
{
    constexpr int N = PL::N, T = PL::T, P = PL::P;
    constexpr int R0 = PL::R0;
    constexpr bool ADDTID = plan_addtid<PL>();
    // MODE 3: a large transform (bins = dec x N) in ONE kernel, decimation in frequency.  Kernel row k is residue
    // q = k mod dec of stream row k / dec:
    //   X[q + dec k'] = sum_m W_N^(m k') { W_bins^(m q) sum_r W_dec^(r q) w[m + N r] x[m + N r] },   m, k' < N, r < dec
    // i.e. the window stage sums the `dec` contiguous blocks of the row (coalesced 16-byte loads, every workgroup
    // reads the whole row: dec x the loads, all but the first from L2), rotates by W_bins^(m q), and the N-point
    // transform follows unchanged; bin q + dec k' leaves as column q + dec j of the fft-shifted row (4-byte stores
    // `dec` floats apart -- the dec workgroups of a stream row run side by side on one XCD and fill its lines
    // together in that XCD's L2).  No scratch: HBM sees the algorithmic bytes only.
    constexpr bool DIF = MODE == 3;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    using S = Sample<FMT>;

    // XCD-aware placement: workgroups b and b+8 share an XCD (round-robin dispatch), so
    // each XCD gets one contiguous run of rows and its workgroups take consecutive rows
    // of it at the same time -- consecutive rows share (N-hop)/N of their input through
    // that XCD's L2.  Placement affects speed only.
    const int64_t per_xcd = (a.rows + 7) / 8;
    const int64_t xcd_first = (int64_t)(blockIdx.x & 7) * per_xcd;
    const int64_t xcd_end = xcd_first + per_xcd < a.rows ? xcd_first + per_xcd : a.rows;
    const int64_t stride = gridDim.x >> 3;
    int64_t row = xcd_first + (blockIdx.x >> 3);
    if (row >= xcd_end) return;

    // De-phase the workgroups.  They all start together and take the same time per row, so
    // left alone every CU bursts its loads, then its LDS phase, then its stores at the same
    // moment and the memory system alternates between idle and a 256-CU queue.  Slot s of an
    // XCD starts s * stagger cycles late (at most about one row time in total).
    if (a.stagger > 0) {
        const unsigned long long t0 = __builtin_amdgcn_s_memtime();
        const unsigned long long wait = (unsigned long long)(blockIdx.x >> 3) * (unsigned)a.stagger;
        while (__builtin_amdgcn_s_memtime() - t0 < wait) __builtin_amdgcn_s_sleep(16);
    }

    const int tid = threadIdx.x;
    const __amdgpu_buffer_rsrc_t rs_tw = make_rsrc(a.twiddles, PL::TW_TOTAL * 8);
    const __amdgpu_buffer_rsrc_t rs_twk = make_rsrc(a.twiddles_k, PL::PK_TOTAL * 16);
    const char *iq = reinterpret_cast<const char *>(a.iq);

    v2f v[P];

    // Sample loads of one row into v[].  PAIRED (one butterfly per thread): the two lanes
    // of a pair (tid even / odd) share their 2*R0 samples -- the even lane fetches both
    // columns for legs 0..R0/2-1, the odd lane for legs R0/2..R0-1, each with 16-byte loads
    // (the L1 path moves a 16-byte-per-lane instruction as fast as an 8-byte one, so this
    // halves the load time).  v[k] / v[R0/2+k] then hold the even / odd column of leg k
    // (resp. R0/2+k); the window stage multiplies them in place and a DPP swap between the
    // two lanes puts every sample into its natural slot.
    // NB stage-0 butterflies per thread: "logical thread" j = tid + T*b (b < NB) owns v[R0*b .. R0*b + R0-1].  The
    // paired scheme needs one butterfly per logical thread; the add-TID plan runs 1024 of them on 1024 or 512 threads.
    constexpr int NB = P / R0, TL = T * NB;
    constexpr bool PAIRED = (NB == 1 || ADDTID) && (R0 % 2 == 0) && RO_PAIRED_LOADS;
    constexpr int H = R0 / 2;
    constexpr bool SWAP32 = plan_swap32<PL>();
    // descriptor of kernel row k's samples / window coefficients (zero-sized when !valid: the loads become no-ops)
    auto row_rsrc = [&](int64_t k, bool valid) {
        const int64_t srow = DIF ? k >> a.dec_log2 : k;
        return make_rsrc(iq + (a.first_row + srow) * (int64_t)a.hop * S::BYTES, valid ? (unsigned)N * S::BYTES : 0u);
    };
    // the thread index as a value hipcc cannot hoist address arithmetic out of the row loop with (MODE 3 sits at its
    // 128 VGPRs: a few shifts and adds per use are cheaper than an invariant parked in scratch)
    auto fresh_tid = [&]() {
        int lt = tid;
        if constexpr (DIF) asm volatile("" : "+v"(lt));
        return lt;
    };
    auto pair_off = [&](int b) { return plan_pair_off<PL>(fresh_tid() + T * b); };   // first sample logical thread b fetches
    auto load_row = [&](const __amdgpu_buffer_rsrc_t &rs) {
        if constexpr (PAIRED) {
#pragma unroll
            for (int b = 0; b < NB; ++b) {
                const int po = pair_off(b) * S::BYTES;
#pragma unroll
                for (int k = 0; k < H; ++k) {
                    v2f &lo = v[R0 * b + k], &hi = v[R0 * b + H + k];
                    S::load_pair(rs, po, k * (N / R0) * S::BYTES, lo, hi);
                }
            }
        } else {
#pragma unroll
            for (int i = 0; i < P; ++i) {
                // slot i: butterfly i / R0, leg i % R0 of stage 0
                v[i] = S::load(rs, (tid + T * (i / R0)) * S::BYTES, (i % R0) * (N / R0) * S::BYTES);
            }
        }
    };

    // ---- prologue: samples of the first row
    load_row(row_rsrc(row, true));

    constexpr int RL = PL::R3 > 1 ? PL::R3 : (PL::R2 > 1 ? PL::R2 : (PL::R1 > 1 ? PL::R1 : PL::R0));
    unsigned long long st_acc[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, st_prev = 0;
    auto stamp = [&](int k) {
        if constexpr (RO_STAMPS) {
            unsigned long long t;
            __builtin_amdgcn_sched_barrier(0);
            asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
            __builtin_amdgcn_sched_barrier(0);
            if (k >= 0) st_acc[k] += t - st_prev;
            st_prev = t;
        }
    };
    stamp(-1);
    if constexpr (RO_STAMPS == 1) st_acc[13] = st_prev;             // when this workgroup started
    // window coefficients of the row about to be transformed (fetched in the previous
    // epilogue / the prologue, all at once: P floats)
    constexpr bool WPERM = PAIRED && RO_WIN_PERM;
    constexpr int NW = PAIRED ? H : P;
    using wtype = std::conditional_t<PAIRED, v2f, float>;
    wtype w[WPERM ? 1 : NW];
    // kernel-order table: legs 2q, 2q+1 as {even, odd, even, odd} column coefficients, per logical thread
    v4f w4[WPERM ? NB * (NW / 2) : 1];
    // first..last-1 of the NW coefficient registers
    auto load_window = [&](const __amdgpu_buffer_rsrc_t &rs_win, auto first_c, auto last_c) {
        constexpr int first = decltype(first_c)::value, last = decltype(last_c)::value;
        if constexpr (WPERM) {
            // kernel-order table: 16 bytes per lane = the coefficient pairs of legs k, k+1
            static_assert(first % 2 == 0 && last % 2 == 0, "window chunks are pairs of legs");
#pragma unroll
            for (int b = 0; b < NB; ++b) {
#pragma unroll
                for (int k = first; k < last; k += 2) {
                    v4f &d = w4[b * (NW / 2) + k / 2];
                    const u32x4 t =
                        __builtin_amdgcn_raw_buffer_load_b128(rs_win, (tid + T * b) * 16, (k / 2) * TL * 16, 0);
                    d = (v4f){__uint_as_float(t.x), __uint_as_float(t.y), __uint_as_float(t.z), __uint_as_float(t.w)};
                }
            }
        } else if constexpr (PAIRED) {
            static_assert(NB == 1 || WPERM, "several logical threads need the kernel-order window table");
#pragma unroll
            for (int k = first; k < last; ++k) {
                w[k] = buf_load_f2(rs_win, pair_off(0) * 4, k * (N / R0) * 4);
            }
        } else {
#pragma unroll
            for (int i = first; i < last; ++i) {
                w[i] = buf_load_f(rs_win, tid * 4, (T * (i / R0) + (i % R0) * (N / R0)) * 4);
            }
        }
    };
    // Coefficients [0, NW_EARLY) of the next row are requested right after the window stage
    // (their registers are free for the whole transform, so these loads cost nothing); the
    // rest follows in the epilogue.  (Keeping all of them resident instead -- they are the same for every row --
    // makes hipcc spill 31 registers; prefetching 100 % fits but leaves no VGPR to spare and gains 1 %.)
    // (the pipelined plan has fewer registers to spare -- the fused scan's two waves keep their band in registers
    // while the next row's samples are already landing: a quarter; with half, hipcc parks one coefficient quad in
    // scratch for the whole row)
    constexpr int NW_EARLY = DIF ? 0 : ((NW * (N == 32768 ? 25 : RO_WIN_EARLY_PCT)) / 100) & ~1;
    using c0 = std::integral_constant<int, 0>;
    using cE = std::integral_constant<int, NW_EARLY>;
    using cN = std::integral_constant<int, NW>;
    const float *win_tab = WPERM ? a.window_k : a.window;
    auto win_rsrc = [&](int64_t, bool valid) { return make_rsrc(win_tab, valid ? N * 4 : 0); };
    if constexpr (!DIF) load_window(win_rsrc(row, true), c0{}, cN{});
    // window and twiddle tables resident in registers (see RO_RESIDENT_TABLES)
    constexpr bool RES = RO_RESIDENT_TABLES && N <= 8192 && !ADDTID;   // twiddles (and window)
    constexpr bool TW8C = ADDTID && PL::R2 == 8;                                     // see tw_prefetch
    // the 512-thread form of the N = 32768 plan has 256 VGPRs per thread: the window stays, the twiddles do not fit
    constexpr bool RESW = RES;
    v2f tw1[PL::R1 > 1 ? P / PL::R1 : 1][TW_SET];
    v2f tw2[PL::R2 > 1 ? P / PL::R2 : 1][TW_SET];
    v2f tw3[PL::R3 > 1 ? P / PL::R3 : 1][TW_SET];
    if constexpr (RES) {
        if constexpr (PL::R1 > 1) tw_prefetch<P, T, PL::R1, PL::NS1, PL::TW1, PL::PK1>(tw1, rs_tw, rs_twk, tid);
        if constexpr (PL::R2 > 1) tw_prefetch<P, T, PL::R2, PL::NS2, PL::TW2, PL::PK2>(tw2, rs_tw, rs_twk, tid);
        if constexpr (PL::R3 > 1) tw_prefetch<P, T, PL::R3, PL::NS3, PL::TW3, PL::PK3>(tw3, rs_tw, rs_twk, tid);
    }

    unsigned touch = 0;                    // destination of the next-row prefetch touches (touch_next)
    [[maybe_unused]] unsigned rows_done = 0;
    for (;;) {
        if constexpr (RO_ROW_PRIO && N >= 1024) {       // one-wave workgroups of N = 512: 5 % slower with it, 256: even
            // The workgroups sharing a CU do equal work, but the arbiter serves the oldest wave first: left alone the
            // workgroup dispatched first runs at nearly its solo speed, the others on what it leaves, and once it is
            // done the CU runs under-filled to the end of the launch (lifetimes 0.63 ... 1.0 of the launch at N =
            // 16384, 0.51 ... 1.0 at 8192: profiles/r03_workgroup_lifetimes.txt).  Priority by rows done, modulo 4:
            // whoever has fallen behind (by up to three rows) outranks its neighbours until it has caught up, and all
            // workgroups of a CU end within a row of each other: 9 % less time at 16384, 7 % at 4096, 4 % at 8192
            // (profiles/r03_row_priority.txt; a step per HALF row, priority 3 - half-rows done mod 4, was 3-6 % slower at
            // 4096 and 2048 and even at 16384 and 8192).
            switch (rows_done & 3) {
            case 0: __builtin_amdgcn_s_setprio(3); break;
            case 1: __builtin_amdgcn_s_setprio(2); break;
            case 2: __builtin_amdgcn_s_setprio(1); break;
            default: __builtin_amdgcn_s_setprio(0); break;
            }
            ++rows_done;
        }
        // ---- stage 0: window.  Coefficients arrive in chunks of WIN_CHUNK, two chunks in
        // flight, so the stage peaks at 2P + 2*WIN_CHUNK VGPRs (+P while a row waits to be stored).
        // ---- stage 0: window (coefficients and samples were requested a whole epilogue ago)
        {
            const v2f gain2 = (v2f){0.0f, a.gain};          // src/FFTBackend.cpp:78-79: Q += gain
            if (a.gain != 0.0f) {                           // every shipped config has iq_gain = 0: skip the adds
#pragma unroll
                for (int i = 0; i < P; ++i) v[i] = v[i] + gain2;
            }
            if constexpr (PAIRED) {
                const bool odd = tid & 1;
                if constexpr (DIF) {
                    // v[] holds block 0 of the row; add the other dec-1 blocks, each times its window block and
                    // W_dec^(r q).  Everything here is per element, so it happens before the lane swap below, with
                    // the loads' own (lane, slot) -> m map.
                    static_assert(WPERM && NB == 1, "MODE 3 runs on the plans with one paired butterfly per thread");
                    load_window(win_rsrc(row, true), c0{}, cN{});
                    const int q = (int)(row & (a.dec - 1));
                    const char *blk = iq + (a.first_row + (row >> a.dec_log2)) * (int64_t)a.hop * S::BYTES;
                    const int po = pair_off(0);
                    // A unit = KC legs of one block: KC 16-byte sample loads and KC/2 16-byte window loads.  Two
                    // units' registers; the loads of unit u+1 are issued in front of the arithmetic of unit u (and
                    // kept there by a fake dependence on unit u-1's results), so a wave always has one unit in flight.
                    constexpr int KC = 4, NU = H / KC;
                    struct Unit { v2f xl[KC], xh[KC]; v4f wq[KC / 2]; };
                    Unit ua, ub;
                    auto unit_load = [&](Unit &u, const __amdgpu_buffer_rsrc_t &rs_b, const __amdgpu_buffer_rsrc_t &rs_w,
                                         int k0, float dep) {
                        const int pod = after(po, dep), ltd = after(tid, dep);
#pragma unroll
                        for (int j = 0; j < KC; ++j)
                            S::load_pair(rs_b, pod * S::BYTES, (k0 + j) * (N / R0) * S::BYTES, u.xl[j], u.xh[j]);
#pragma unroll
                        for (int j = 0; j < KC / 2; ++j) {
                            const u32x4 t = __builtin_amdgcn_raw_buffer_load_b128(rs_w, ltd * 16, ((k0 / 2) + j) * TL * 16, 0);
                            u.wq[j] = (v4f){__uint_as_float(t.x), __uint_as_float(t.y), __uint_as_float(t.z), __uint_as_float(t.w)};
                        }
                    };
                    auto unit_add = [&](Unit &u, int k0, v2f cxx, v2f cyn) {
#pragma unroll
                        for (int j = 0; j < KC; ++j) {
                            const int k = k0 + j;
                            if (a.gain != 0.0f) { u.xl[j] = u.xl[j] + gain2; u.xh[j] = u.xh[j] + gain2; }
                            const v4f c4 = u.wq[j / 2];
                            const v2f e = u.xl[j] * ((k & 1) ? c4.zz : c4.xx);
                            const v2f o = u.xh[j] * ((k & 1) ? c4.ww : c4.yy);
                            v[k] = __builtin_elementwise_fma(e.yx, cyn, __builtin_elementwise_fma(e, cxx, v[k]));
                            v[H + k] = __builtin_elementwise_fma(o.yx, cyn, __builtin_elementwise_fma(o, cxx, v[H + k]));
                        }
                    };
                    // block r's descriptors; r = dec: the rotation table (8-byte entries: the float sample path), no window
                    auto blk_rsrc = [&](int r) {
                        return r < a.dec ? make_rsrc(blk + (int64_t)r * N * S::BYTES, N * S::BYTES) : make_rsrc(nullptr, 0);
                    };
                    auto wblk_rsrc = [&](int r) { return make_rsrc(win_tab + (int64_t)r * N, r < a.dec ? N * 4 : 0); };
                    static_assert(NU == 4, "the unit schedule below is written out for four units per block");
                    unit_load(ua, blk_rsrc(1), wblk_rsrc(1), 0, v[0].x);
#pragma unroll
                    for (int k = 0; k < H; ++k) {                  // block 0
                        const v4f c4 = w4[k / 2];
                        v[k] = v[k] * ((k & 1) ? c4.zz : c4.xx);
                        v[H + k] = v[H + k] * ((k & 1) ? c4.ww : c4.yy);
                    }
#pragma unroll 1
                    for (int r = 1; r < a.dec; ++r) {
                        const __amdgpu_buffer_rsrc_t rs_b = blk_rsrc(r), rs_w = wblk_rsrc(r);
                        const float2 c = a.dif_tw[(r * q) & (a.dec - 1)];
                        const v2f cxx = (v2f){c.x, c.x}, cyn = (v2f){-c.y, c.y};
                        unit_load(ub, rs_b, rs_w, KC, v[H + 3 * KC].x);
                        unit_add(ua, 0, cxx, cyn);
                        unit_load(ua, rs_b, rs_w, 2 * KC, v[0].x);
                        unit_add(ub, KC, cxx, cyn);
                        unit_load(ub, rs_b, rs_w, 3 * KC, v[H + KC].x);
                        unit_add(ua, 2 * KC, cxx, cyn);
                        unit_load(ua, blk_rsrc(r + 1), wblk_rsrc(r + 1), 0, v[H + 2 * KC].x);     // (zero-sized past the last block)
                        unit_add(ub, 3 * KC, cxx, cyn);
                    }
                    // (no rotation here: W_bins^(m q) is a shift of the bin index by q / dec, and the three stages take it
                    // in their twiddles -- shift_stage below)
                }
#pragma unroll
                for (int b = 0; b < NB; ++b) {
#pragma unroll
                for (int k = 0; k < H; ++k) {
                    v2f &lo = v[R0 * b + k], &hi = v[R0 * b + H + k];
                    v2f we, wo;                                    // coefficient of the even / odd column, both halves
                    if constexpr (DIF) {
                        we = wo = (v2f){1.0f, 1.0f};               // (already applied)
                    } else if constexpr (WPERM) {
                        const v4f c4 = w4[b * (NW / 2) + k / 2];
                        we = (k & 1) ? c4.zz : c4.xx;
                        wo = (k & 1) ? c4.ww : c4.yy;
                    } else {
                        we = w[k].xx;
                        wo = w[k].yy;
                    }
                    const v2f e = DIF ? lo : lo * we;              // even column, leg k (H+k on odd lanes)
                    const v2f o = DIF ? hi : hi * wo;              // odd column
                    if constexpr (SWAP32) {
                        // lanes 0..31 hold legs k, lanes 32..63 legs H+k of both columns: the upper half of slot k
                        // trades places with the lower half of slot H+k (v_permlane32_swap_b32)
                        const auto rx = __builtin_amdgcn_permlane32_swap(__float_as_uint(e.x), __float_as_uint(o.x), false, false);
                        const auto ry = __builtin_amdgcn_permlane32_swap(__float_as_uint(e.y), __float_as_uint(o.y), false, false);
                        lo = (v2f){__uint_as_float(rx[0]), __uint_as_float(ry[0])};
                        hi = (v2f){__uint_as_float(rx[1]), __uint_as_float(ry[1])};
                        continue;
                    }
                    // even lane keeps e in slot k and takes the partner's e (leg H+k) into slot H+k;
                    // odd lane keeps o in slot H+k and takes the partner's o (leg k) into slot k.
                    const v2f pe = (v2f){dpp_quad<0xB1>(e.x), dpp_quad<0xB1>(e.y)};
                    const v2f po = (v2f){dpp_quad<0xB1>(o.x), dpp_quad<0xB1>(o.y)};
                    lo = odd ? po : e;
                    hi = odd ? o : pe;
                }
                }
            } else {
#pragma unroll
                for (int i = 0; i < P; ++i) v[i] = v[i] * (v2f){w[i], w[i]};
            }
        }
        // The coefficients for the NEXT row are requested right away: their registers are free
        // from here on and the loads overlap the whole transform instead of the epilogue's
        // memory burst.  (Unconditional: the same table every row.)
        if constexpr (ADDTID) asm volatile("" ::"v"(touch));     // see touch_next
        const int64_t next = row + stride;
        const bool has_next = next < xcd_end;
        if constexpr (!RESW && !DIF) load_window(win_rsrc(has_next ? next : row, true), c0{}, cE{});
        auto touch_next = [&]() {
            if constexpr (!ADDTID) return;          // only the add-TID plan is launched with a.prefetch
            // samples [next*hop + N - hop, next*hop + N) = hop * BYTES bytes: one dword per 128-byte line and thread
            // (reaches 128 KiB: hop <= N/2 for float samples; whatever lies past the descriptor's end costs nothing).
            // The dword is not wanted, only its line in L2; its register is "used" only after the next window stage
            // (younger loads have been waited for by then, so that use never waits).  One touch per thread, one VGPR:
            // with two, hipcc ran out of registers in the pipelined loop and spilled them behind an s_waitcnt that sat
            // through the HBM miss the touch exists to hide.
            const int64_t s0 = (a.first_row + (has_next ? next : row)) * (int64_t)a.hop + (N - a.hop);
            // switched off (a.prefetch == 0, last row) by a zero-sized descriptor, not by a branch
            const __amdgpu_buffer_rsrc_t rs_new =
                make_rsrc(iq + s0 * S::BYTES, (has_next && a.prefetch) ? a.hop * S::BYTES : 0);
            touch = __builtin_amdgcn_raw_buffer_load_b32(rs_new, tid * 128, 0, 0);
        };
        if constexpr (RO_PREFETCH_NEXT == 1) touch_next();
        stamp(0);                                   // window multiply (+ wait for samples)

        // MODE 3, residue q != 0: the braces carry the factor W_bins^(m q) = W_N^(m q / dec) -- the transform evaluated
        // at the bins k' + q / dec.  In the autosort recurrence that is every stage's twiddle exponent (j mod Ns) moved
        // by q / dec (stage 0, Ns = 1: w^r with w = exp(-2 pi i (q / dec) / R0), the same for every thread), so the
        // rotation costs no table and no loads: 15 uniform constants per residue (StftArgs::dif_shift: {w, w^2, w^4,
        // w^8, w^16} of the shift at each of the three radix-32 stages) -- stage 0 runs its twiddled form, stages 1 and
        // 2 multiply their five held powers by the shift's.  (Until round 3 the rotation was a [dec][N] table read in the
        // window stage, 256 KiB from L2 per row with q != 0: 65536 1.3 % and 131072 3.5 % faster without it,
        // profiles/r03_dif_shift.txt.  Residue 0 runs the twiddled stage 0 as well, with w = 1: the two forms side by
        // side behind a branch on q cost 68 spilled registers.)
        [[maybe_unused]] const int dif_q = DIF ? (int)(row & (a.dec - 1)) : 0;
        [[maybe_unused]] auto shift_stage = [&](v2f (&t)[TW_SET], int stage) {
            const float2 *sh = a.dif_shift + dif_q * 16 + stage * 5;
#pragma unroll
            for (int i = 0; i < 5; ++i) t[i] = cmul(t[i], (v2f){sh[i].x, sh[i].y});
        };
        if constexpr (DIF) {
            static_assert(R0 == 32 && PL::R1 == 32 && PL::R2 == 32 && P == 32, "the shifted stages are written for 32.32.32");
            {
                const float2 *sh = a.dif_shift + dif_q * 16;
                fdit32(v, (v2f){sh[4].x, sh[4].y}, (v2f){sh[3].x, sh[3].y}, (v2f){sh[2].x, sh[2].y},
                       (v2f){sh[1].x, sh[1].y}, (v2f){sh[0].x, sh[0].y});
            }
        } else {
            butterflies<P, R0>(v);
        }
        if constexpr (PL::R1 > 1 && !RES) tw_prefetch<P, T, PL::R1, PL::NS1, PL::TW1, PL::PK1>(tw1, rs_tw, rs_twk, fresh_tid());
        stamp(2);                                   // butterflies 0

        // ---- stage 1
        if constexpr (PL::R1 > 1) {
            if constexpr (ADDTID) exchange_addtid<1, SWAP32, P / 32, T>(smem, v, tid, [&](int k) { if constexpr (RO_STAMPS == 3) stamp(12 + k); });
            else exchange<PL, PL::R0, 1, PL::R1>(smem, v, fresh_tid(), [&](int k) { if constexpr (RO_STAMPS == 3) stamp(12 + k); });
            stamp(3);                               // exchange 1
            if constexpr (DIF) { if (dif_q != 0) shift_stage(tw1[0], 1); }
            tw_butterflies<P, PL::R1>(v, tw1);
            // behind the butterflies: in front of them hipcc's wait for this pass's twiddles (which it believes to be
            // the youngest loads in flight) would sit through the touches' HBM misses as well
            if constexpr (RO_PREFETCH_NEXT == 2) touch_next();
            stamp(4);                               // twiddles + butterflies 1
        }
        // ---- stage 2
        if constexpr (PL::R2 > 1) {
            if constexpr (!RES) tw_prefetch<P, T, PL::R2, PL::NS2, PL::TW2, PL::PK2, TW8C>(tw2, rs_tw, rs_twk, fresh_tid());
            if constexpr (ADDTID) exchange_addtid<2, SWAP32, P / 32, T>(smem, v, tid, [&](int k) { if constexpr (RO_STAMPS == 3) stamp(12 + k); });
            else exchange<PL, PL::R1, PL::NS1, PL::R2>(smem, v, fresh_tid(), [&](int k) { if constexpr (RO_STAMPS == 3) stamp(12 + k); });
            stamp(5);                               // exchange 2
            if constexpr (RO_PREFETCH_NEXT == 3) touch_next();
            if constexpr (DIF) { if (dif_q != 0) shift_stage(tw2[0], 2); }
            tw_butterflies<P, PL::R2, TW8C>(v, tw2);
            stamp(6);                               // twiddles + butterflies 2

        }
        // ---- stage 3
        if constexpr (PL::R3 > 1) {
            if constexpr (!RES) tw_prefetch<P, T, PL::R3, PL::NS3, PL::TW3, PL::PK3>(tw3, rs_tw, rs_twk, tid);
            exchange<PL, PL::R2, PL::NS2, PL::R3>(smem, v, tid, [](int) {});
            tw_butterflies<P, PL::R3>(v, tw3);
        }

        // ---- epilogue: |X[k]| -> column (k + N/2) mod N  (src/WaterfallBackend.cpp:492-505).
        // vmcnt retires in issue order and counts stores, so whatever is loaded after a store
        // cannot be used before that store has been acknowledged (~5k cycles here).  The row
        // therefore leaves the registers through LDS (free at this point): magnitudes are
        // written there in row order, the next row's samples and window coefficients are
        // requested into the freed registers, and only then the row is read back 16 bytes per
        // lane and stored -- 1 KiB per wave-instruction, the stores being the LAST thing in the
        // VMEM queue.
        if constexpr (MODE == 1) {
            // slot r of butterfly b is bin (tid + T b) + r N/RL: 8 bytes per lane, 512 contiguous bytes per wave
            const __amdgpu_buffer_rsrc_t rs_spec =
                make_rsrc(a.spec_out + row * a.spec_stride, (unsigned)N * 8u);
#pragma unroll
            for (int b = 0; b < P / RL; ++b) {
#pragma unroll
                for (int r = 0; r < RL; ++r) {
                    const v2f x = v[b * RL + bitrev<RL>(r)];
                    const u32x2 t = {__float_as_uint(x.x), __float_as_uint(x.y)};
                    __builtin_amdgcn_raw_buffer_store_b64(t, rs_spec, (tid + T * b) * 8, r * (N / RL) * 8, RO_STORE_AUX);
                }
            }
            load_row(row_rsrc(has_next ? next : row, has_next));
            if constexpr (!RESW) load_window(win_rsrc(has_next ? next : row, has_next), cE{}, cN{});
            st_acc[9] += 1;
            if (!has_next) break;
            row = next;
            continue;
        }
        if constexpr (ADDTID) {
            // slot q of logical thread j is column j + TL q: byte 4 TL q + 4 j of the LDS image, i.e. the row in
            // natural order, written lane-linearly (ds_write_addtid_b32); the fft-shift moves into the store offsets
            // (last stage of radix RL < 32: butterfly b2 of the logical thread, output s, is column j + TL (b2 + (32/RL) s))
#pragma unroll
            for (int b = 0; b < NB; ++b) {
                float m[32];
#pragma unroll
                for (int q = 0; q < 32; ++q) {
                    constexpr int B2 = 32 / RL;
                    const v2f x = v[32 * b + (q % B2) * RL + bitrev<RL>(q / B2)];
                    const v2f sq = x * x;
                    m[q] = __builtin_amdgcn_sqrtf(sq.x + sq.y);      // v_sqrt_f32, 1 ulp
                }
                const unsigned wave_bytes = (unsigned)__builtin_amdgcn_readfirstlane((tid + T * b) >> 6) * 256u;
                addtid_scatter32<TL * 4>(wave_bytes, [&](int q) { return m[q]; });
            }
        } else {
            float *lds_m = reinterpret_cast<float *>(smem);
            const int lt = fresh_tid();
#pragma unroll
            for (int b = 0; b < P / RL; ++b) {
#pragma unroll
                for (int r = 0; r < RL; ++r) {
                    const v2f x = v[b * RL + bitrev<RL>(r)];
                    const v2f sq = x * x;
                    const float mag = __builtin_amdgcn_sqrtf(sq.x + sq.y);   // v_sqrt_f32, 1 ulp
                    const int cbase = (r * (N / RL) + N / 2) & (N - 1);
                    lds_m[lt + T * b + cbase] = mag;             // j < N/RL, cbase a multiple of it: no wrap
                }
            }
        }
        stamp(10);                                  // magnitudes -> LDS writes issued
        // a zero-sized descriptor turns the loads into no-ops after the last row
        load_row(row_rsrc(has_next ? next : row, has_next));
        // unconditional (zero-sized descriptor after the last row): a branch here would keep the
        // old coefficients alive next to the new ones
        // (MODE 3 asks for block 0's coefficients at the head of its window stage: across the loop's back edge they
        // were the 32 registers too many)
        if constexpr (!RESW && !DIF) load_window(win_rsrc(has_next ? next : row, has_next), cE{}, cN{});
        stamp(7);                                   // next-row loads issued
        // the add-TID writes sit inside inline asm: hipcc does not count them, so the barrier's own lgkmcnt wait
        // is missing unless it is spelled out (the image is read by OTHER waves right behind the barrier)
        if constexpr (ADDTID) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        wg_sync();
        stamp(11);                                  // barrier 1
        if constexpr (DIF) {
            // image element j is column q + dec j of the shifted row (add-TID plans, whose image is in natural order:
            // column q + dec ((j + N/2) mod N)): one float per lane, `dec` floats apart (default cache policy: the
            // other residues' workgroups fill the same lines)
            const int q = (int)(row & (a.dec - 1));
            const __amdgpu_buffer_rsrc_t rs_out =
                make_rsrc(a.rows_out + (row >> a.dec_log2) * a.row_stride, (unsigned)N * (unsigned)a.dec * 4u);
            const float *lds_m1 = reinterpret_cast<const float *>(smem);
            const int lt = fresh_tid();
            const int vo = (lt * a.dec + q) * 4;
            int step = T * a.dec * 4;
            asm volatile("" : "+s"(step));                  // (else hipcc keeps P multiples of it in SGPRs all row long)
#pragma unroll
            for (int i = 0; i < P; ++i) {
                const int io = ADDTID ? ((i + P / 2) & (P - 1)) : i;
                buf_store_f(lds_m1[lt + T * i], rs_out, vo, io * step);
                if ((i & 7) == 7) asm volatile("" ::: "memory");
            }
        } else {
            const __amdgpu_buffer_rsrc_t rs_out = make_rsrc(a.rows_out + row * a.row_stride, N * 4);
            const float4 *lds_m4 = reinterpret_cast<const float4 *>(smem);
#pragma unroll
            for (int q = 0; q < P / 4; ++q) {
                const float4 x = lds_m4[tid + T * q];
                if constexpr (ADDTID)      // natural-order image: column k leaves for (k + N/2) mod N
                    buf_store_f4(x.x, x.y, x.z, x.w, rs_out, tid * 16, ((q * T * 4 + N / 2) & (N - 1)) * 4);
                else buf_store_f4(x.x, x.y, x.z, x.w, rs_out, tid * 16, q * T * 16);
                // two reads in flight at most: hoisting all P/4 of them would need P more VGPRs
                // while the next row's samples and window are already landing in theirs
                if (q & 1) asm volatile("" ::: "memory");
            }
        }
        stamp(12);                                  // LDS read-back + row stores issued
        wg_sync();                            // LDS is reused by the next row's exchange
        stamp(8);                                   // barrier 2
        st_acc[9] += 1;
        if (!has_next) break;
        row = next;
    }
    if constexpr (RO_STAMPS) {
        if constexpr (RO_STAMPS == 1) {                            // ... when it ended, and where it ran
            unsigned hw, xcc;
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)\n\ts_getreg_b32 %1, hwreg(HW_REG_XCC_ID)" : "=s"(hw), "=s"(xcc));
            st_acc[14] = st_prev;
            st_acc[15] = ((unsigned long long)xcc << 32) | hw;
        }
        if (a.stamps && tid == 0)
            for (int k = 0; k < 16; ++k) a.stamps[blockIdx.x * 16 + k] = st_acc[k];
    }
}

// ---------------------------------------------------------------------------
// Large transforms (bins = dec x 32768: 65536 ... 1048576, the station sizes of Bolidozor.json:45 and Ionozor.json:27).
// A row no longer fits a CU.  Decimation in frequency over the N = 32768 single-pass plan:
//   X[q + dec k'] = sum_m W_N^(m k') { W_bins^(m q) sum_r W_dec^(r q) w[m + N r] x[m + N r] }
// dec <= 4: ONE kernel, stft_kernel MODE 3 -- the braces are summed in its window stage (every workgroup reads the
//   whole row: dec x the loads, from L2), bins leave `dec` floats apart.  HBM sees the algorithmic bytes only.
// dec >= 8: the strided 4-byte stores of that form cost one 64-byte L2 write request per lane and residue: the
//   magnitude rows come from the four-step pair of kernels in ro_fourstep.hip.
// Complex spectra of every large size: three steps through scratch, every access a run of consecutive elements:
//   fold_kernel (the braces, 8 B per bin), the N = 32768 kernel on its rows in spectra mode, interleave2_kernel.
// ---------------------------------------------------------------------------
// ---------------------------------------------------------------------------
// fold_kernel: first step of the scratch form of a large transform (see FoldArgs).  One thread owns two neighbouring
// columns i, i+1 and walks down the rows of its group: per row it reads the 2 R samples of its columns (16 bytes per
// block, consecutive lanes consecutive columns), does the radix-R butterfly across the blocks and writes R pairs.
// Window coefficients and rotations depend on the column only and stay in registers for all rows (R <= 16).
// Bound: HBM -- hop x 8 B in, 8 B per bin out.
// ---------------------------------------------------------------------------
template <int R, int FMT> __global__ __launch_bounds__(256) void fold_kernel(FoldArgs a)
{
    using S = Sample<FMT>;
    constexpr bool RES = R <= 16;                                  // rotations resident (R = 32: re-read per row, from L2)
    const int pair = blockIdx.x * 256 + threadIdx.x;               // columns 2 pair, 2 pair + 1;  2 pair < m
    const int64_t per = (a.rows + gridDim.y - 1) / gridDim.y;
    const int64_t r0 = (int64_t)blockIdx.y * per, r1 = r0 + per < a.rows ? r0 + per : a.rows;
    if (r0 >= r1) return;
    const int m = a.m;
    v2f w[R];
    v4f rot[RES ? R : 1];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const float2 t = reinterpret_cast<const float2 *>(a.window + (int64_t)r * m)[pair];
        w[r] = (v2f){t.x, t.y};
    }
    const float4 *rot4 = reinterpret_cast<const float4 *>(a.rot);
    if constexpr (RES) {
#pragma unroll
        for (int q = 1; q < R; ++q) {
            const float4 t = rot4[(int64_t)q * (m / 2) + pair];
            rot[q] = (v4f){t.x, t.y, t.z, t.w};
        }
    }
    const v2f gain2 = (v2f){0.0f, a.gain};
    const char *iq = reinterpret_cast<const char *>(a.iq);
    for (int64_t row = r0; row < r1; ++row) {
        const __amdgpu_buffer_rsrc_t rs =
            make_rsrc(iq + (a.first_row + row) * (int64_t)a.hop * S::BYTES, (unsigned)m * R * S::BYTES);
        v2f u[R], v[R];
#pragma unroll
        for (int r = 0; r < R; ++r) {
            S::load_pair(rs, pair * 2 * S::BYTES, r * m * S::BYTES, u[r], v[r]);
        }
#pragma unroll
        for (int r = 0; r < R; ++r) {
            if (a.gain != 0.0f) { u[r] = u[r] + gain2; v[r] = v[r] + gain2; }
            u[r] = u[r] * w[r].xx;
            v[r] = v[r] * w[r].yy;
        }
        dif<R>(u);
        dif<R>(v);
        float4 *out = reinterpret_cast<float4 *>(a.out + row * (int64_t)R * m) + pair;
#pragma unroll
        for (int q = 0; q < R; ++q) {
            v2f x = u[bitrev<R>(q)], y = v[bitrev<R>(q)];
            if (q > 0) {
                v4f t;
                if constexpr (RES) t = rot[q];
                else {
                    const float4 l = rot4[(int64_t)q * (m / 2) + pair];
                    t = (v4f){l.x, l.y, l.z, l.w};
                }
                x = cmul(x, t.xy);
                y = cmul(y, t.zw);
            }
            out[(int64_t)q * (m / 2)] = make_float4(x.x, x.y, y.x, y.y);
        }
    }
}


// ---------------------------------------------------------------------------
// Strict precision (ro_stft_config_t::precision = RO_PRECISION_F64): the reference's arithmetic type.  FFTBackend
// multiplies double samples by the float window in double (src/FFTBackend.cpp:229-232), runs FFTW's double transform
// (:117-120, :236) and takes sqrt(re^2 + im^2) in double before narrowing to the float row
// (src/WaterfallBackend.cpp:492-505).  A row of doubles (512 KiB at N = 32768) does not fit a CU's registers, so this
// mode runs the Stockham autosort recurrence as separate passes through HBM for EVERY size, in double: radix-16 passes (the last one 2..16)
// over two complex-double scratch blocks in HBM, twiddles from one correctly rounded exp(-2 pi i m/N) table, the
// butterflies' own constants in double.  Bound: HBM at 32 B per point and pass; never the benchmarked shape.  Its
// rows agree with the oracle's FP64 radix-2 transform to a few 1e-16 of the row maximum, i.e. per bin to ~1e-12 at
// 60 dB of dynamic range -- the per-bin reading of "1e-5 relative" that fp32 butterflies cannot meet.
// ---------------------------------------------------------------------------
template <int R, bool FIRST, bool LAST, int FMT>
__global__ __launch_bounds__(256) void f64_pass_kernel(BigArgsD a)
{
    const int per_row = a.n / R;                                  // butterflies per row
    const int64_t g = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t row = g / per_row;
    if (row >= a.rows) return;
    const int j = (int)(g - row * per_row);
    v2d v[R];
    if constexpr (FIRST) {
        using S = Sample<FMT>;
        const int64_t s0 = (a.first_row + row) * (int64_t)a.hop;
        const __amdgpu_buffer_rsrc_t rs =
            make_rsrc(reinterpret_cast<const char *>(a.iq) + s0 * S::BYTES, (unsigned)a.n * S::BYTES);
#pragma unroll
        for (int k = 0; k < R; ++k) {
            const int n = j + k * per_row;
            const double w = (double)a.window[n];
            const v2f x = S::load(rs, n * S::BYTES, 0);
            v[k] = (v2d){(double)x.x * w, ((double)x.y + a.gain) * w};       // src/FFTBackend.cpp:78-79, :229-232
        }
    } else {
        const double2 *in = a.in + row * (int64_t)a.n;
        const int kk = j & (a.ns - 1);
        const int step = a.n / (a.ns * R);
#pragma unroll
        for (int k = 0; k < R; ++k) {
            const double2 x = in[j + k * per_row];
            v[k] = (v2d){x.x, x.y};
            if (k > 0) {
                const double2 t = a.tw[(int64_t)k * kk * step];
                v[k] = cmul_d(v[k], (v2d){t.x, t.y});
            }
        }
    }
    dif_d<R>(v);
    const int j0 = (j / a.ns) * (a.ns * R) + (j & (a.ns - 1));
    if constexpr (LAST) {
        float *out = a.rows_out + row * a.row_stride;
#pragma unroll
        for (int k = 0; k < R; ++k) {
            const v2d x = v[bitrev<R>(k)];
            out[(j0 + k * a.ns + a.n / 2) & (a.n - 1)] = (float)sqrt(x.x * x.x + x.y * x.y);   // WaterfallBackend.cpp:492-505
        }
    } else {
        double2 *out = a.out + row * (int64_t)a.n;
#pragma unroll
        for (int k = 0; k < R; ++k) {
            const v2d x = v[bitrev<R>(k)];
            out[j0 + k * a.ns] = make_double2(x.x, x.y);
        }
    }
}
// Two passes of the recurrence in one kernel (f64_pair_tile, ro_f64_device.h): one trip through HBM (16 B per point
// each way) instead of two.  Same butterflies, same table entries, same order as f64_pass_kernel twice: bit-identical
// results.
template <int R2, bool FIRST, bool LAST, int FMT>
__global__ __launch_bounds__(256) void f64_pair_kernel(BigArgsD a)
{
    constexpr int TPW = 4096 / (16 * R2);
    extern __shared__ __attribute__((aligned(16))) char smem_d[];
    double2 *lds = reinterpret_cast<double2 *>(smem_d);               // [slot s][k'][tile]: 16 x R2 x TPW
    const int tiles_per_row = a.n / (16 * R2);
    const int64_t wg = blockIdx.x;
    const int64_t row = wg / (tiles_per_row / TPW);
    const int tile0 = (int)(wg - row * (tiles_per_row / TPW)) * TPW;
    f64_pair_tile<R2, FIRST, LAST, FMT, false>(a, lds, row, tile0, a.in + row * (int64_t)a.n, a.out + row * (int64_t)a.n);
}


// ---------------------------------------------------------------------------
// band tile: compact copy of columns [first, first+cols) of every row (what the FITS
// writer keeps, src/WaterfallBackend.cpp:176,204) -- the unit the multi-GPU gather moves.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void tile_kernel(TileArgs a)
{
    const int64_t total = a.rows * (int64_t)a.cols;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t r = i / a.cols;
        const int c = (int)(i - r * a.cols);
        a.tile_out[i] = a.rows_in[r * a.row_stride + a.first + c];
    }
}

// ---------------------------------------------------------------------------
// per-row band scan (one wavefront per row)
// ---------------------------------------------------------------------------
// ---------------------------------------------------------------------------
// ln tile: the viewer's FN_LOG (fits2png:46: numpy.log of the non-zero float32 pixels) over the band
// columns, its min / max (fits2png:476-477), and the grey level (v - min) / (max - min) * 255 cut to
// uint8 (:444-445, :495-497).  Zero pixels (the viewer drops them) give -inf in the ln tile and level 0.
// Pass 1 writes ln and reduces min/max, pass 2 quantises; both are plain streaming kernels.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void ln_reduce_kernel(LnArgs a)
{
    __shared__ unsigned s_min[4], s_max[4];
    const int64_t total = a.rows * (int64_t)a.cols;
    unsigned kmin = 0xffffffffu, kmax = 0u;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t r = i / a.cols;
        const int c = (int)(i - r * a.cols);
        const float v = a.rows_in[r * a.row_stride + a.first + c];
        const float l = logf(v);
        if (a.ln_out) a.ln_out[i] = l;
        if (v != 0.f) {
            const unsigned k = order_key(l);
            kmin = min(kmin, k);
            kmax = max(kmax, k);
        }
    }
    for (int d = 32; d >= 1; d >>= 1) {
        kmin = min(kmin, (unsigned)__shfl_xor((int)kmin, d));
        kmax = max(kmax, (unsigned)__shfl_xor((int)kmax, d));
    }
    if ((threadIdx.x & 63) == 0) {
        s_min[threadIdx.x >> 6] = kmin;
        s_max[threadIdx.x >> 6] = kmax;
    }
    wg_sync();
    if (threadIdx.x == 0) {
        for (int w = 1; w < 4; ++w) {
            kmin = min(kmin, s_min[w]);
            kmax = max(kmax, s_max[w]);
        }
        atomicMin(&a.keys[0], kmin);
        atomicMax(&a.keys[1], kmax);
    }
}

__global__ __launch_bounds__(256) void ln_quant_kernel(LnArgs a)
{
    const float mn = key_to_float(a.keys[0]), mx = key_to_float(a.keys[1]);
    if (a.minmax && blockIdx.x == 0 && threadIdx.x == 0) {
        a.minmax[0] = mn;
        a.minmax[1] = mx;
    }
    if (!a.u8_out) return;
    const float span = mx - mn;
    const int64_t total = a.rows * (int64_t)a.cols;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t r = i / a.cols;
        const int c = (int)(i - r * a.cols);
        const float v = a.rows_in[r * a.row_stride + a.first + c];
        const float l = a.ln_out ? a.ln_out[i] : logf(v);
        const float level = (l - mn) / span * 255.f;               // float32 throughout, like numpy
        a.u8_out[i] = (v != 0.f && span > 0.f) ? (uint8_t)(int)level : (uint8_t)0;
    }
}

// ln of a compact tile, one wave per row, with the row's min / max over the non-zero pixels: what the fused epilogue
// of the N = 32768 plan computes from LDS, for the plans without it (and the FP64 mode)
__global__ __launch_bounds__(256) void ln_rows_kernel(const float *tile, float *ln_out, float *minmax, int64_t rows, int cols)
{
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float *src = tile + row * (int64_t)cols;
    unsigned kmin = 0xffffffffu, kmax = 0u;
    for (int c = lane; c < cols; c += 64) {
        const float x = src[c];
        const float l = logf(x);
        if (ln_out) ln_out[row * (int64_t)cols + c] = l;
        if (x != 0.f) {
            const unsigned key = order_key(l);
            kmin = min(kmin, key);
            kmax = max(kmax, key);
        }
    }
    kmin = wave_min_u32(kmin);
    kmax = wave_max_u32(kmax);
    if (lane == 0 && minmax) {
        minmax[2 * row] = kmin == 0xffffffffu ? __builtin_inff() : key_to_float(kmin);
        minmax[2 * row + 1] = kmax == 0u ? -__builtin_inff() : key_to_float(kmax);
    }
}

// fold the two tile waves' partial min / max of the fused epilogue into one pair per row
__global__ __launch_bounds__(256) void ln_finish_kernel(const float *part, float *minmax, int64_t rows)
{
    const int64_t row = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (row >= rows) return;
    const float4 p = reinterpret_cast<const float4 *>(part)[row];
    minmax[2 * row] = fminf(p.x, p.z);
    minmax[2 * row + 1] = fmaxf(p.y, p.w);
}

constexpr int SCAN_WAVES = 4;         // rows per workgroup

// E: band elements a lane keeps in registers.  Bands up to 64 E columns are loaded ONCE, all loads in flight together;
// wider ones are re-read by every pass of the radix select, one dependent trip to L2 per 64 columns -- at Ionozor's
// 524288 bins (bands of thousands of columns) that was 245 us per launch whatever the row count, 8 % of the size's time.
template <int E> __global__ __launch_bounds__(64 * SCAN_WAVES) void scan_kernel(ScanArgs a)
{
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * SCAN_WAVES + (threadIdx.x >> 6);
    if (row >= a.rows) return;
    const GlobalRow src{a.rows_in + row * a.row_stride};
    __shared__ __attribute__((aligned(16))) unsigned hist[SCAN_WAVES][256];
    unsigned *h = hist[threadIdx.x >> 6];
    const bool cached = a.noise_width <= 64 * E;
    const float noise = cached ? scan_noise<E>(src, a.low_noise, a.noise_width, h, lane)
                               : scan_noise<0>(src, a.low_noise, a.noise_width, h, lane);
    const int peak = scan_peak<E>(src, a.low_detect, a.detect_width, lane);
    const float avg = scan_average(src, a.low_detect + peak - a.avg_bins / 2, a.avg_bins, a.bins, lane);
    if (lane == 0) {
        ro_scan_record_t rec;
        rec.noise = noise;
        rec.peak = peak;
        rec.average = avg;
        a.records[row] = rec;
    }
}

// ---------------------------------------------------------------------------
// launch table
// ---------------------------------------------------------------------------
// Persistent launch: as many workgroups as the device can hold at once (rounded down to a
// multiple of 8 so every XCD gets the same share), never more than there are rows.
// What a plan's kernel needs per DEVICE before its first launch there: the dynamic-LDS attribute set and its occupancy
// known.  One entry per device ordinal, filled under a lock (a host may drive several devices and threads through the C
// ABI's cfg.device; the round-1 version kept these in unsynchronised function-local statics of whichever device
// launched first).
struct DevicePlan {
    bool ready = false;
    int resident = 0;            // workgroups resident on the device (all CUs)
    int per_cu = 1;              // ... per CU
};
constexpr int MAX_DEVICES = 64;

template <class PL, int FMT, int MODE> static hipError_t device_plan(DevicePlan &out)
{
    static std::mutex lock;
    static DevicePlan table[MAX_DEVICES];
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    if (dev < 0 || dev >= MAX_DEVICES) return hipErrorInvalidDevice;
    std::lock_guard<std::mutex> g(lock);
    DevicePlan &d = table[dev];
    if (!d.ready) {
        const void *fn = reinterpret_cast<const void *>(&stft_kernel<PL, FMT, MODE>);
        if ((e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, plan_lds_bytes<PL>())) != hipSuccess)
            return e;
        int cus = 0, per_cu = 0;
        if ((e = hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev)) != hipSuccess) return e;
        if ((e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fn, PL::T, plan_lds_bytes<PL>())) != hipSuccess)
            return e;
        d.per_cu = per_cu < 1 ? 1 : per_cu;
        d.resident = cus * d.per_cu;
        d.ready = true;
    }
    out = d;
    return hipSuccess;
}

#ifdef RO_DIAG_KNOBS
// experiment knobs of the diagnostic builds (tools/sweep_env.sh): never compiled into the product library
static int env_knob(const char *name, int unset)
{
    const char *e = getenv(name);
    return e ? atoi(e) : unset;
}
#endif

template <class PL, int FMT, int MODE> static hipError_t launch_plan(const StftArgs &a, hipStream_t s)
{
    DevicePlan d;
    hipError_t e = device_plan<PL, FMT, MODE>(d);
    if (e != hipSuccess) return e;
    const int64_t per_xcd = (a.rows + 7) / 8;
    int64_t slots = d.resident / 8;                     // workgroups per XCD
    if (a.spare_cus > 0) slots -= (int64_t)a.spare_cus * d.per_cu;
    if (slots < 1) slots = 1;
    if (slots > per_xcd) slots = per_xcd;
#ifdef RO_DIAG_KNOBS
    if (const int cap = env_knob("RO_SLOTS", 0); cap > 0 && slots > cap) slots = cap;      // workgroups per XCD
#endif
    const unsigned grid = (unsigned)(slots * 8);
    StftArgs b = a;
    if (MODE < 2) { b.dec = 1; b.dec_log2 = 0; }
    // The touches park hop*BYTES per resident workgroup in the XCD's 4 MiB L2 for most of a row time.  Past half of
    // it they push out the rows being transformed and every line is fetched twice (seen at overlap 0: FETCH_SIZE x2,
    // 19 % slower).  Plans with several workgroups per CU hide the miss behind each other and gain nothing (measured).
    b.prefetch = (MODE < 2 && plan_addtid<PL>() && slots * (int64_t)a.hop * Sample<FMT>::BYTES <= (2 << 20)) ? 1 : 0;
    b.stagger = 0;
#ifdef RO_DIAG_KNOBS
    if (const int force = env_knob("RO_PREFETCH", -1); force >= 0) b.prefetch = force;
    b.stagger = env_knob("RO_STAGGER", 0);             // start delay per workgroup slot, shader cycles
#endif
    hipLaunchKernelGGL((stft_kernel<PL, FMT, MODE>), dim3(grid), dim3(PL::T), plan_lds_bytes<PL>(), s, b);
    return hipGetLastError();
}

//                      N      T   R0  R1  R2  R3  split
using Plan32768 = Plan<32768, 1024, 32, 32, 32, 1, true>;    // complex spectra and MODE 3 only
using Plan16384 = Plan<16384,  512, 32, 32, 16, 1, true>;
using Plan8192  = Plan< 8192,  256, 32, 32,  8, 1, (RO_ADDTID_8192 && RO_USE_ADDTID)>;
using Plan4096  = Plan< 4096,  256, 16, 16, 16, 1, false>;
using Plan2048  = Plan< 2048,  128, 16, 16,  8, 1, false>;
using Plan1024  = Plan< 1024,   64, 16, 16,  4, 1, false>;
using Plan512   = Plan<  512,   64,  8,  8,  8, 1, false>;
using Plan256   = Plan<  256,   64,  4,  4,  4, 4, false>;

template <class PL> static hipError_t launch_fmt(const StftArgs &a, int fmt, hipStream_t s)
{
    const bool spec = a.spec_out != nullptr;
    if (a.big_form) {                                  // a large transform in one kernel (MODE 3)
        if constexpr (PL::N == 32768) {
            if (spec) return hipErrorInvalidValue;
            if (fmt == RO_FMT_F32) return launch_plan<PL, RO_FMT_F32, 3>(a, s);
            if (fmt == RO_FMT_I16) return launch_plan<PL, RO_FMT_I16, 3>(a, s);
        }
        return hipErrorInvalidValue;
    }
    if constexpr (PL::N == 32768) {
        if (!spec) return launch_stft32k(fmt, a, s);               // magnitude rows: ro_stft32k.hip
        if (fmt == RO_FMT_F32) return launch_plan<PL, RO_FMT_F32, 1>(a, s);
        if (fmt == RO_FMT_I16) return launch_plan<PL, RO_FMT_I16, 1>(a, s);
    } else {
#if RO_USE_WL
        if constexpr (PL::N == 16384 || PL::N == 8192) {
            if (!spec) return launch_stft_wl(PL::N, fmt, a, s);    // magnitude rows: ro_stft_wl.hip
        }
#endif
        if (fmt == RO_FMT_F32) return spec ? launch_plan<PL, RO_FMT_F32, 1>(a, s) : launch_plan<PL, RO_FMT_F32, 0>(a, s);
        if (fmt == RO_FMT_I16) return spec ? launch_plan<PL, RO_FMT_I16, 1>(a, s) : launch_plan<PL, RO_FMT_I16, 0>(a, s);
    }
    return hipErrorInvalidValue;
}

bool stft_fuses_scan(int bins)
{
    // ro_stft32k.hip, ro_stft_wl.hip.  (The scan costs the same few thousand cycles of ONE wave whatever the row length:
    // a sixteenth of the workgroup for a twelfth of the row at 32768 and it hides in the oldest waves' slack; at 8192 it
    // is a quarter of the workgroup for half of the row -- there the separate scan_kernel, one wave per row over all
    // rows at once, is the cheaper form.)
    return bins == 32768 || (RO_USE_WL && ((bins == 16384 && RO_WL_FUSE16) || (bins == 8192 && RO_WL_FUSE8)));
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

// The window table in the order the kernel consumes it (StftArgs::window_k): thread `tid` of the plan reads
// 16 bytes at ((k/2)*T + tid)*16 = {w[c + k*S], w[c+1 + k*S], w[c + (k+1)*S], w[c+1 + (k+1)*S]} with
// S = N/R0 and c = the first sample it fetches (plan_pair_off), k = 0, 2, .. R0/2-2.
template <class PL> static void window_layout(const float *w, float *out)
{
    constexpr int N = PL::N, R0 = PL::R0, H = R0 / 2, S = N / R0;
    constexpr int T = PL::T * (PL::P / R0);                      // logical threads: one stage-0 butterfly each
    static_assert((PL::P == R0 || plan_addtid<PL>()) && H % 2 == 0, "paired plan");
    for (int tid = 0; tid < T; ++tid) {
        const int c = plan_pair_off<PL>(tid);
        for (int k = 0; k < H; ++k) {
            float *o = out + ((size_t)(k / 2) * T + tid) * 4 + 2 * (k % 2);
            o[0] = w[c + k * S];
            o[1] = w[c + 1 + k * S];
        }
    }
}

bool stft_window_layout(int bins, const float *w, float *out)
{
    switch (bins) {
    case 32768: window_layout<Plan32768>(w, out); return true;
    case 16384: window_layout<Plan16384>(w, out); return true;
    case 8192:  window_layout<Plan8192>(w, out);  return true;
    case 4096:  window_layout<Plan4096>(w, out);  return true;
    case 2048:  window_layout<Plan2048>(w, out);  return true;
    case 1024:  window_layout<Plan1024>(w, out);  return true;
    case 512:   window_layout<Plan512>(w, out);   return true;
    case 256:   window_layout<Plan256>(w, out);   return true;
    default:    return false;
    }
}

// packed twiddle table (StftArgs::twiddles_k) from the generic one: for every stage of radix 16 / 32, unit
// q*NS + k = {w^a(k), w^b(k)} with (a, b) = (1,2) (4,8) (16,-) for radix 32 and (1,2) (3,4) (8,12) for radix 16;
// generic entry (c-1)*NS + k = w^c(k).
template <class PL> static void pack_twiddles(const float2 *tw, float4 *out)
{
    const int radix[3] = {PL::R1, PL::R2, PL::R3}, ns[3] = {PL::NS1, PL::NS2, PL::NS3};
    const int off[3] = {PL::TW1, PL::TW2, PL::TW3}, pk[3] = {PL::PK1, PL::PK2, PL::PK3};
    static const int pairs32[3][2] = {{1, 2}, {4, 8}, {16, 16}}, pairs16[3][2] = {{1, 2}, {3, 4}, {8, 12}};
    for (int s = 0; s < 3; ++s) {
        const int nq = PL::pkq(radix[s]);
        const int(*pairs)[2] = radix[s] == 32 ? pairs32 : pairs16;
        for (int q = 0; q < nq; ++q)
            for (int k = 0; k < ns[s]; ++k) {
                const float2 a = tw[off[s] + (pairs[q][0] - 1) * ns[s] + k];
                const float2 b = tw[off[s] + (pairs[q][1] - 1) * ns[s] + k];
                out[pk[s] + q * ns[s] + k] = make_float4(a.x, a.y, b.x, b.y);
            }
    }
}

int stft_packed_twiddle_count(int bins)      // float4 units, <0 if unsupported
{
    switch (bins) {
    case 32768: return Plan32768::PK_TOTAL;
    case 16384: return Plan16384::PK_TOTAL;
    case 8192:  return Plan8192::PK_TOTAL;
    case 4096:  return Plan4096::PK_TOTAL;
    case 2048:  return Plan2048::PK_TOTAL;
    case 1024:  return Plan1024::PK_TOTAL;
    case 512:   return Plan512::PK_TOTAL;
    case 256:   return Plan256::PK_TOTAL;
    default:    return -1;
    }
}

bool stft_pack_twiddles(int bins, const float2 *tw, float4 *out)
{
    switch (bins) {
    case 32768: pack_twiddles<Plan32768>(tw, out); return true;
    case 16384: pack_twiddles<Plan16384>(tw, out); return true;
    case 8192:  pack_twiddles<Plan8192>(tw, out);  return true;
    case 4096:  pack_twiddles<Plan4096>(tw, out);  return true;
    case 2048:  pack_twiddles<Plan2048>(tw, out);  return true;
    case 1024:  pack_twiddles<Plan1024>(tw, out);  return true;
    case 512:   pack_twiddles<Plan512>(tw, out);   return true;
    case 256:   pack_twiddles<Plan256>(tw, out);   return true;
    default:    return false;
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

bool big_supported(int bins)
{
    return bins > 32768 && bins <= (1 << 20) && (bins & (bins - 1)) == 0;
}

// interleave2_kernel: [dec][m] -> [m][dec] per stream row through an LDS tile (reads: dec runs of 1 KiB, writes: one run of
// dec KiB), complex elements.  Bound: HBM, 8 B per bin each way.
template <int R> __global__ __launch_bounds__(256) void interleave2_kernel(Interleave2Args a)
{
    __shared__ float2 tile[R][129];
    const int j0 = blockIdx.x * 128, t = threadIdx.x;
    const int64_t row = blockIdx.y;
    const float2 *in = a.in + row * (int64_t)R * a.m + j0;
#pragma unroll
    for (int i = 0; i < R / 2; ++i) {
        const int e = t + 256 * i;                                 // R x 128 elements, 128 per sub-row
        tile[e >> 7][e & 127] = in[(int64_t)(e >> 7) * a.m + (e & 127)];
    }
    __syncthreads();
    float2 *out = a.out + row * a.out_stride + (int64_t)j0 * R;
#pragma unroll
    for (int i = 0; i < R / 2; ++i) {
        const int e = t + 256 * i;
        out[e] = tile[e % R][e / R];
    }
}

hipError_t launch_interleave2(const Interleave2Args &a, hipStream_t s)
{
    if (a.rows <= 0) return hipSuccess;
    if (a.m % 128 != 0 || a.rows > 65535) return hipErrorInvalidValue;
    const dim3 grid((unsigned)(a.m / 128), (unsigned)a.rows);
    switch (a.dec) {
    case 2:  hipLaunchKernelGGL(interleave2_kernel<2>, grid, dim3(256), 0, s, a); break;
    case 4:  hipLaunchKernelGGL(interleave2_kernel<4>, grid, dim3(256), 0, s, a); break;
    case 8:  hipLaunchKernelGGL(interleave2_kernel<8>, grid, dim3(256), 0, s, a); break;
    case 16: hipLaunchKernelGGL(interleave2_kernel<16>, grid, dim3(256), 0, s, a); break;
    case 32: hipLaunchKernelGGL(interleave2_kernel<32>, grid, dim3(256), 0, s, a); break;
    default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

// ---------------------------------------------------------------------------
// Chirp-z (Bluestein) steps for lengths that are not a power of two; see CztArgs.  Plain streaming kernels, one
// element per thread: HBM-bound and not the benchmarked shape (every shipped config is a power of two).
// ---------------------------------------------------------------------------
template <int FMT> __global__ __launch_bounds__(256) void czt_pre_kernel(CztArgs a)
{
    using S = Sample<FMT>;
    const int i = blockIdx.x * 256 + threadIdx.x;                  // < m
    const int64_t row = blockIdx.y;
    v2f x = (v2f){0.0f, 0.0f};
    if (i < a.n) {
        const char *iq = reinterpret_cast<const char *>(a.iq);
        const __amdgpu_buffer_rsrc_t rs =
            make_rsrc(iq + (a.first_row + row) * (int64_t)a.hop * S::BYTES, (unsigned)a.n * S::BYTES);
        x = S::load(rs, i * S::BYTES, 0);
        x.y += a.gain;                                             // src/FFTBackend.cpp:78-79
        const float2 c = a.cw[i];
        x = cmul(x, (v2f){c.x, c.y});
    }
    a.a[row * (int64_t)a.m + i] = make_float2(x.x, x.y);
}

__global__ __launch_bounds__(256) void czt_mul_kernel(CztArgs a)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    const int64_t row = blockIdx.y;
    float2 *p = a.a + row * (int64_t)a.m + i;
    const float2 x = *p, b = a.bc[i];
    *p = make_float2(x.x * b.x + x.y * b.y, x.x * b.y - x.y * b.x);      // conj(x) * b
}

__global__ __launch_bounds__(256) void czt_out_kernel(CztArgs a)
{
    const int k = blockIdx.x * 256 + threadIdx.x;                  // output column
    const int64_t row = blockIdx.y;
    if (k >= a.n) return;
    // column k holds bin (k + n/2) mod n (src/WaterfallBackend.cpp:492-505, n even); the length-m transform left bin b
    // at its own column (b + m/2) mod m
    const int bin = k >= a.n / 2 ? k - a.n / 2 : k + a.n / 2;
    a.rows_out[row * a.row_stride + k] = a.mag[row * (int64_t)a.m + ((bin + a.m / 2) & (a.m - 1))];
}

hipError_t launch_czt_pre(int format, const CztArgs &a, hipStream_t s)
{
    if (a.rows <= 0) return hipSuccess;
    if (a.m % 256 != 0 || a.rows > 65535) return hipErrorInvalidValue;
    const dim3 grid((unsigned)(a.m / 256), (unsigned)a.rows);
    if (format == RO_FMT_F32) hipLaunchKernelGGL(czt_pre_kernel<RO_FMT_F32>, grid, dim3(256), 0, s, a);
    else if (format == RO_FMT_I16) hipLaunchKernelGGL(czt_pre_kernel<RO_FMT_I16>, grid, dim3(256), 0, s, a);
    else return hipErrorInvalidValue;
    return hipGetLastError();
}

hipError_t launch_czt_mul(const CztArgs &a, hipStream_t s)
{
    if (a.rows <= 0) return hipSuccess;
    if (a.m % 256 != 0 || a.rows > 65535) return hipErrorInvalidValue;
    hipLaunchKernelGGL(czt_mul_kernel, dim3((unsigned)(a.m / 256), (unsigned)a.rows), dim3(256), 0, s, a);
    return hipGetLastError();
}

hipError_t launch_czt_out(const CztArgs &a, hipStream_t s)
{
    if (a.rows <= 0) return hipSuccess;
    if (a.rows > 65535) return hipErrorInvalidValue;
    hipLaunchKernelGGL(czt_out_kernel, dim3((unsigned)((a.n + 255) / 256), (unsigned)a.rows), dim3(256), 0, s, a);
    return hipGetLastError();
}

template <int R> static hipError_t launch_fold_r(int format, const FoldArgs &a, hipStream_t s)
{
    // m / 2 column pairs: m / 512 blocks across; enough row groups down to fill the device a few times over
    const unsigned bx = (unsigned)(a.m / 512);
    int64_t groups = (2048 + bx - 1) / bx;
    if (groups > a.rows) groups = a.rows;
    const dim3 grid(bx, (unsigned)groups);
    if (format == RO_FMT_F32) hipLaunchKernelGGL((fold_kernel<R, RO_FMT_F32>), grid, dim3(256), 0, s, a);
    else if (format == RO_FMT_I16) hipLaunchKernelGGL((fold_kernel<R, RO_FMT_I16>), grid, dim3(256), 0, s, a);
    else return hipErrorInvalidValue;
    return hipGetLastError();
}

hipError_t launch_fold(int format, const FoldArgs &a, hipStream_t s)
{
    if (a.rows <= 0) return hipSuccess;
    if (a.m % 512 != 0) return hipErrorInvalidValue;
    switch (a.dec) {
    case 2:  return launch_fold_r<2>(format, a, s);
    case 4:  return launch_fold_r<4>(format, a, s);
    case 8:  return launch_fold_r<8>(format, a, s);
    case 16: return launch_fold_r<16>(format, a, s);
    case 32: return launch_fold_r<32>(format, a, s);
    default: return hipErrorInvalidValue;
    }
}

int f64_radices(int bins, int radices[8])
{
    if (bins < 256 || bins > (1 << 20) || (bins & (bins - 1)) != 0) return 0;
    int l = 0;
    while ((1 << l) < bins) ++l;
    int n = 0;
    while (l >= 8 || l == 4) {              // radix 16 while the remainder still leaves a last pass of >= 2 ... 16
        radices[n++] = 16;
        l -= 4;
    }
    if (l > 4) {                            // 5..7 bits left: 16 then 2, 4 or 8
        radices[n++] = 16;
        l -= 4;
    }
    if (l > 0) radices[n++] = 1 << l;
    return n;
}

template <int R, bool FIRST, bool LAST, int FMT> static hipError_t launch_f64(const BigArgsD &a, hipStream_t s)
{
    const int64_t total = a.rows * (int64_t)(a.n / R);
    const int64_t blocks = (total + 255) / 256;
    if (blocks > 0x7fffffff) return hipErrorInvalidValue;
    hipLaunchKernelGGL((f64_pass_kernel<R, FIRST, LAST, FMT>), dim3((unsigned)blocks), dim3(256), 0, s, a);
    return hipGetLastError();
}

hipError_t launch_f64_pass(int radix, bool first, bool last, int fmt, const BigArgsD &a, hipStream_t s)
{
    if (a.rows <= 0) return hipSuccess;
    if (first && !last && radix == 16)
        return fmt == RO_FMT_I16 ? launch_f64<16, true, false, RO_FMT_I16>(a, s)
                                 : launch_f64<16, true, false, RO_FMT_F32>(a, s);
    if (!first && !last && radix == 16) return launch_f64<16, false, false, RO_FMT_F32>(a, s);
    if (!first && last) {
        switch (radix) {
        case 2:  return launch_f64<2, false, true, RO_FMT_F32>(a, s);
        case 4:  return launch_f64<4, false, true, RO_FMT_F32>(a, s);
        case 8:  return launch_f64<8, false, true, RO_FMT_F32>(a, s);
        case 16: return launch_f64<16, false, true, RO_FMT_F32>(a, s);
        }
    }
    return hipErrorInvalidValue;
}

template <int R2, bool FIRST, bool LAST, int FMT> static hipError_t launch_f64_pair_t(const BigArgsD &a, hipStream_t s)
{
    const void *fn = reinterpret_cast<const void *>(&f64_pair_kernel<R2, FIRST, LAST, FMT>);
    hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 4096 * 16);
    if (e != hipSuccess) return e;
    const int64_t blocks = a.rows * (int64_t)(a.n / 4096);
    if (blocks > 0x7fffffff) return hipErrorInvalidValue;
    hipLaunchKernelGGL((f64_pair_kernel<R2, FIRST, LAST, FMT>), dim3((unsigned)blocks), dim3(256), 4096 * 16, s, a);
    return hipGetLastError();
}

// passes p (radix 16, a.ns) and p+1 (radix r2) in one kernel; n >= 4096
hipError_t launch_f64_pair(int r2, bool first, bool last, int fmt, const BigArgsD &a, hipStream_t s)
{
    if (a.rows <= 0) return hipSuccess;
    if (a.n < 4096) return hipErrorInvalidValue;
    if (r2 == 16) {
        if (first && !last)
            return fmt == RO_FMT_I16 ? launch_f64_pair_t<16, true, false, RO_FMT_I16>(a, s)
                                     : launch_f64_pair_t<16, true, false, RO_FMT_F32>(a, s);
        if (!first) return last ? launch_f64_pair_t<16, false, true, RO_FMT_F32>(a, s)
                                : launch_f64_pair_t<16, false, false, RO_FMT_F32>(a, s);
    } else if (!first && last) {            // (first && last would be n = 16 r2 < 4096)
        switch (r2) {
        case 8: return launch_f64_pair_t<8, false, true, RO_FMT_F32>(a, s);
        case 4: return launch_f64_pair_t<4, false, true, RO_FMT_F32>(a, s);
        case 2: return launch_f64_pair_t<2, false, true, RO_FMT_F32>(a, s);
        }
    }
    return hipErrorInvalidValue;
}

hipError_t launch_tile(const TileArgs &a, hipStream_t s)
{
    if (a.rows <= 0 || a.cols <= 0) return hipSuccess;
    const int64_t total = a.rows * (int64_t)a.cols;
    int64_t blocks = (total + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(tile_kernel, dim3((unsigned)blocks), dim3(256), 0, s, a);
    return hipGetLastError();
}

hipError_t launch_ln_tile(const LnArgs &a, hipStream_t s)
{
    if (a.rows <= 0 || a.cols <= 0) return hipSuccess;
    const int64_t total = a.rows * (int64_t)a.cols;
    int64_t blocks = (total + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    hipError_t e = hipMemsetAsync(a.keys, 0xff, sizeof(unsigned), s);        // min key = all ones
    if (e != hipSuccess) return e;
    e = hipMemsetAsync(a.keys + 1, 0, sizeof(unsigned), s);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(ln_reduce_kernel, dim3((unsigned)blocks), dim3(256), 0, s, a);
    hipLaunchKernelGGL(ln_quant_kernel, dim3((unsigned)(a.u8_out ? blocks : 1)), dim3(256), 0, s, a);
    return hipGetLastError();
}

hipError_t launch_ln_rows(const float *tile, float *ln_out, float *minmax, int64_t rows, int cols, hipStream_t s)
{
    if (rows <= 0 || cols <= 0) return hipSuccess;
    hipLaunchKernelGGL(ln_rows_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, s, tile, ln_out, minmax, rows, cols);
    return hipGetLastError();
}

hipError_t launch_ln_finish(const float *ln_part, float *minmax, int64_t rows, hipStream_t s)
{
    if (rows <= 0) return hipSuccess;
    hipLaunchKernelGGL(ln_finish_kernel, dim3((unsigned)((rows + 255) / 256)), dim3(256), 0, s, ln_part, minmax, rows);
    return hipGetLastError();
}

hipError_t launch_scan(const ScanArgs &a, hipStream_t s)
{
    if (a.rows <= 0) return hipSuccess;
    const unsigned grid = (unsigned)((a.rows + SCAN_WAVES - 1) / SCAN_WAVES);
    const int widest = a.noise_width > a.detect_width ? a.noise_width : a.detect_width;
    if (widest <= 64 * SCAN_E) hipLaunchKernelGGL(scan_kernel<SCAN_E>, dim3(grid), dim3(64 * SCAN_WAVES), 0, s, a);
    else if (widest <= 64 * 64) hipLaunchKernelGGL(scan_kernel<64>, dim3(grid), dim3(64 * SCAN_WAVES), 0, s, a);
    else hipLaunchKernelGGL(scan_kernel<128>, dim3(grid), dim3(64 * SCAN_WAVES), 0, s, a);   // wider still: batched re-reads
    return hipGetLastError();
}

}  // namespace ro
