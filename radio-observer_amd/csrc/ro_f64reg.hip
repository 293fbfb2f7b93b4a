// ro_f64reg.hip -- RO_PRECISION_F64 with the row in a compute unit's registers: window -> FFT -> |X| -> fft-shift ->
// float32 row in the reference's own arithmetic type (double window multiply, double transform, double square root, one
// narrowing: src/FFTBackend.cpp:117-120, :229-236, src/WaterfallBackend.cpp:492-505) and NO complex-double scratch in
// HBM: the samples are read once, the float row is written once.
//
// bins N = D x M, M = 16 x 16 x 16 x R3 in {4096, 8192, 16384} (R3 = 1, 2, 4), D in {1, 2, 4}.  One workgroup of
// T = M / 16 threads holds sub-row q of a stream row -- the bins q + D k, k < M -- as 16 complex doubles per thread
// (64 VGPRs of the 128 a thread has at 16 waves per CU); D workgroups make a row, side by side on one XCD.
//   sub-row:   y_q[i] = sum_r W_D^(r q) w[i + M r] x[i + M r]          (the fold: decimation in frequency by D)
//              X[q + D k] = sum_i y_q[i] W_N^(i q) W_M^(i k)           (an M-point transform TWISTED by theta = W_N^q)
// A twisted transform  F_L(x; theta)[k] = sum_n x[n] theta^n W_L^(n k)  splits like a plain one (n = n' + L' j, k = kd + R k'):
//              F_L(x; theta)[kd + R k'] = F_L'(y_kd; theta W_L^kd)[k'],   y_kd[n'] = F_R(x[n' + L' .]; theta^L')[kd]
// so every pass is a twisted radix-16 transform of the thread's 16 points whose twist depends on the output digits made so
// far only, never on the remaining input index: no separate twiddle multiply between the passes, and
//   pass 0  twist W_(16 D)^q                 the same for the whole workgroup        (scalar loads)
//   pass 1  twist W_(256 D)^(q + D k0)       k0 = the wave at M = 16384              (scalar loads there)
//   pass 2  twist W_(4096 D)^(q + D K1)      K1 = k0 + 16 k1, 256 values             (eight 16-byte loads per thread)
//   pass 3  (R3 > 1) radix R3, twist W_N^(q + D K2), K2 = K1 + 256 k2                (two 16-byte loads per thread)
// The twisted radix-16 transform is four levels of radix-2 butterflies (a + t b, a - t b), six FMAs each, where block
// beta of level l uses t = w^(8 >> l) W_16^(bitrev_l(beta) (8 >> l)): eight table entries {w^8, w^4, w^2, w^2 W_8, w,
// w W_16, w W_16^2, w W_16^3} and a free factor -i cover all fifteen.
//
// Thread maps (t = thread, Q = T / 16 = 16 R3):
//   pass 0   thread n1 = t, slot n0:                 sample i = n1 + T n0          (coalesced, 8 bytes per lane)
//   pass 1   thread (k0, n2) = (t / Q, t % Q), slot j1:  n1 = n2 + Q j1
//   pass 2   thread (k0, k1, n3), u = t % Q = k1 R3 + n3, slot j2:  n2 = n3 + R3 j2
//   pass 3   thread (k0, k1, g), slots (i, n3):      k2 = g + R3 i;  result k3 at slot i R3 + bitrev(k3)
//   bin      k = k0 + 16 k1 + 256 k2 + 4096 k3
// The exchange after pass 0 crosses the workgroup; from there on the 16 Q values of one k0 stay with the Q threads of
// that k0, which share a wave: exchanges 2 and 3 need no barrier.  LDS holds ONE plane of doubles at a time (a complex
// row of M = 16384 is 256 KiB): real parts, then imaginary parts.  Territory of k0 = ST = T + 16 R3 doubles:
//   exchange 1   cell k0 ST + n1                    write: slot k0, lanes n1 linear; read: lanes n2 linear
//   exchange 2   cell k0 ST + k1 (Q + R3) + n2      the padding R3 makes the stride-R3 reads conflict-free
//   exchange 3   cell k0 ST + k2 (Q + 1) + u
//   image        float cell 2 k0 ST + s Q + rot(u)  magnitudes by slot s, rotated by k0 so that the read-out (lanes =
//                sixteen k0 x four u: 64-byte runs of consecutive bins) is conflict-free as well
// tools/r6/emu_f64r.py restates all of it in numpy against numpy's FFT and counts the bank conflicts (none).
// Five workgroup barriers per sub-row.
//
// bins 256 ... 2048 (N = 256 R): B = 4096 / N rows share the M = 4096 workgroup (f64r_kernel's LOGB).  The workgroup's index
// i = n2 + 16 j1 + 256 n0 carries the row in the LOW bits of the last digit, n2 = b + B m, and means sample m + R j1 + 16 R n0
// of row b.  Passes 0 and 1 (over n0, j1) are then the 4096-point transform's own -- their twists W_256^k0 do not know N --
// and what is left of a row, the digit m, is what the first log2 R levels of pass 2 transform, with the 4096-point table
// again: W_4096^(K1 B m) = W_N^(K1 m).  Every map above stays; slot p of the result is row p mod B, k2 = bitrev(p / B).
//
// SPEC instantiations (ro_stft_spectra_resident on an RO_PRECISION_F64 handle): the transform itself instead of |X| -- its
// real parts, then its imaginary parts, each narrowed to float once, take the image's way out (two barriers more), unshifted.
#include "ro_kernels.h"
#include "ro_fft_device.h"
#include "ro_device_util.h"

#include <mutex>
#include <cmath>
#include <type_traits>

// 1: (float)sqrt(double) as hipcc expands it (about fifteen FP64 operations); 0: float square root + one residual step in
// double (exact to ~1e-14 of a float ulp before the final rounding)
#ifndef RO_F64R_SQRT_EXACT
#define RO_F64R_SQRT_EXACT 0
#endif

// Between barrier (d) and barrier (e) a wave is on its own for two thirds of the sub-row's work, and the arbiter serves the
// oldest wave of a SIMD first.  RO_F64R_PRIO = 1: a wave's priority falls as it gets through the stretch (pass 1: 3,
// pass 2: 2, pass 3: 1, magnitudes: 0) in the kernels of M <= 2^RO_F64R_PRIO_MAX_LOGM.  Measured (tools/r6/f64r_ab.sh,
// profiles/r06_f64r_ab.txt): +3.5 % at M = 4096, where the four waves of a SIMD belong to four workgroups at different
// places of their sub-rows (the one that is furthest behind goes first); -7 % at M = 16384, where they belong to one
// workgroup and meet at its barriers anyway (the magnitudes then wait behind everything).
#ifndef RO_F64R_PRIO
#define RO_F64R_PRIO 1
#endif
#ifndef RO_F64R_PRIO_MAX_LOGM
#define RO_F64R_PRIO_MAX_LOGM 12
#endif

// Cache policy of the 4-byte row stores at D > 1 (a sub-row owns every D-th column: the D workgroups of a row write the
// same lines a few microseconds apart and the line should wait for all of them in L2): gfx950 aux bits, 0 = default,
// 2 = nt like every other row store of the library.
#ifndef RO_F64R_STORE_AUX_D
#define RO_F64R_STORE_AUX_D 0
#endif

// 1: the exchanges' LDS reads are volatile, which keeps hipcc from fusing two ds_read_b64 into one ds_read2_b64 /
// ds_read2st64_b64: the fused forms run at half the bytes per clock (MI355X_MICROARCH.md, LDS table: 128 against 256
// B/clk/CU) and bank their lanes in groups of 16 over 32 banks, for which the strides below are not chosen.
#ifndef RO_F64R_PLAIN_READS
#define RO_F64R_PLAIN_READS 1
#endif

// 1 = the per-lane table entries of passes 2 and 3 are asked for right behind barrier (d), in front of pass 1 (pass 1's own,
// where it is per lane: behind barrier (b)) -- the POWERS {w^8, w^4, w^2, w} only, 16 registers per pass; the other four
// entries are made from them inside the transform.  Later they queue behind the older waves' requests for the
// next sub-row's samples (24 KiB per wave through one in-order memory pipeline): the youngest wave of a SIMD then waited
// ~10k cycles for 160 bytes in the middle of pass 2 (profiles/r06_f64r_stamps.txt).
#ifndef RO_F64R_EARLY_TABLES
#define RO_F64R_EARLY_TABLES 1
#endif

// 1: the new samples of the sub-row AFTER the next one are touched into L2 ahead of time (the next one's are requested into
// registers a barrier ahead anyway).  Measured (profiles/r06_f64r_ab.txt): the touched lines are gone again before they
// are asked for -- FETCH_SIZE doubles (1.33 x algorithmic in all against 1.00 x) and the kernel is 2-3 % slower: off.
#ifndef RO_F64R_TOUCH
#define RO_F64R_TOUCH 0
#endif

// The ONE diagnostic switch of this file.  A -DRO_DIAG=1 build (tools/ab_build.sh) may set RO_F64R_STAMPS=1: s_memtime
// deltas per phase of the sub-row loop, every wave of every workgroup, accumulated into Args::stamps
// (tools/r6/f64r_stamps.py).  Never timed, never shipped.
#if !defined(RO_DIAG) || !defined(RO_F64R_STAMPS)
#undef RO_F64R_STAMPS
#define RO_F64R_STAMPS 0
#endif

namespace ro {
namespace f64r {

typedef double d2 __attribute__((ext_vector_type(2)));

template <int LOGM> struct Geo {
    static constexpr int M = 1 << LOGM, T = M / 16, R3 = M / 4096, Q = T / 16;
    static constexpr int ST = T + 16 * R3;          // doubles per k0 territory
    static constexpr int S2 = Q + (R3 == 1 ? 1 : 2);   // exchange 2: doubles between two k1 slots
    static constexpr int PLANE = 16 * ST;           // doubles
    static constexpr int LDS_BYTES = PLANE * 8;
    static constexpr int L3 = R3 == 1 ? 0 : R3 == 2 ? 1 : 2;
    static_assert(M == 4096 || M == 8192 || M == 16384, "M = 16^3 R3");
};

// (a, b) <- (a + t' b, a - t' b),  t' = t (-i)^ROT: six FMAs
template <int ROT>
__device__ __forceinline__ void bfly(double &ar, double &ai, double &br, double &bi, const double tr, const double ti)
{
    double cr, ci;
    if constexpr (ROT == 0) { cr = tr; ci = ti; } else { cr = ti; ci = -tr; }
    const double p = __builtin_fma(-bi, ci, ar), o1r = __builtin_fma(br, cr, p);
    const double q = __builtin_fma(bi, cr, ai), o1i = __builtin_fma(br, ci, q);
    br = __builtin_fma(2.0, ar, -o1r);
    bi = __builtin_fma(2.0, ai, -o1i);
    ar = o1r;
    ai = o1i;
}

template <int BITS> __host__ __device__ constexpr int brev(int k)
{
    int r = 0;
    for (int i = 0; i < BITS; ++i) r |= ((k >> i) & 1) << (BITS - 1 - i);
    return r;
}

// one block of one level: L = 16 >> LVL points starting at BETA * L
struct TwC;
// (a, b) <- (a + (-i)^ROT b, a - (-i)^ROT b): four additions
template <int ROT> __device__ __forceinline__ void bfly_plain(double &ar, double &ai, double &br, double &bi)
{
    const double tr = ROT == 0 ? br : bi, ti = ROT == 0 ? bi : -br;
    br = ar - tr;
    bi = ai - ti;
    ar = ar + tr;
    ai = ai + ti;
}
template <int LVL, int BETA, typename TW> __device__ __forceinline__ void block16(double *re, double *im, TW &tw)
{
    constexpr int L = 16 >> LVL, H = L / 2, E = brev<LVL>(BETA) * H;         // E = exponent of W16
    constexpr int IDX = LVL == 0 ? 0 : LVL == 1 ? 1 : LVL == 2 ? 2 + (E % 4) / 2 : 4 + E % 4;
    constexpr int ROT = E / 4;
#pragma unroll
    for (int m = 0; m < H; ++m) {
        double &ar = re[BETA * L + m], &ai = im[BETA * L + m], &br = re[BETA * L + m + H], &bi = im[BETA * L + m + H];
        if constexpr (std::is_same<typename std::remove_reference<TW>::type, TwC>::value) {
            // exp(-2 pi i (E % 4) / 16) = (c, -s)
            constexpr double C[4] = {1.0, 0.92387953251128675613, 0.70710678118654752440, 0.38268343236508977173};
            constexpr double S[4] = {0.0, 0.38268343236508977173, 0.70710678118654752440, 0.92387953251128675613};
            if constexpr (E % 4 == 0) bfly_plain<ROT>(ar, ai, br, bi);
            else bfly<ROT>(ar, ai, br, bi, C[E % 4], -S[E % 4]);
        } else {
            bfly<ROT>(ar, ai, br, bi, tw.re(IDX), tw.im(IDX));
        }
    }
}
template <int LVL, typename TW, int... Bs>
__device__ __forceinline__ void level16(double *re, double *im, TW &tw, std::integer_sequence<int, Bs...>)
{
    (block16<LVL, Bs>(re, im, tw), ...);
}
// twisted radix-16 transform in place: result kd at position bitrev4(kd)
template <typename TW> __device__ __forceinline__ void twisted16(double *re, double *im, TW &&tw)
{
    level16<0>(re, im, tw, std::make_integer_sequence<int, 1>{});
    level16<1>(re, im, tw, std::make_integer_sequence<int, 2>{});
    tw.second_half();                                   // (per-lane tables: entries 4..7, the last level's, are asked for here)
    level16<2>(re, im, tw, std::make_integer_sequence<int, 4>{});
    level16<3>(re, im, tw, std::make_integer_sequence<int, 8>{});
}

// its first LEVELS levels only: sixteen / 2^LEVELS twisted radix-2^LEVELS transforms over the TOP bits of the position, one
// per value of the low bits (rows batched into a workgroup: see f64r_kernel's LOGB); result kd at top bits = bitrev(kd)
template <int LEVELS, typename TW> __device__ __forceinline__ void twisted16_top(double *re, double *im, TW &&tw)
{
    if constexpr (LEVELS >= 1) level16<0>(re, im, tw, std::make_integer_sequence<int, 1>{});
    if constexpr (LEVELS >= 2) level16<1>(re, im, tw, std::make_integer_sequence<int, 2>{});
    if constexpr (LEVELS >= 3) {
        tw.second_half();                               // (w^2 W_8, where it is made from w^2)
        level16<2>(re, im, tw, std::make_integer_sequence<int, 4>{});
    }
}

// eight twiddles in VGPRs (per-lane table entry) or SGPRs (an entry the whole wave shares)
template <bool LAZY> struct TwVT {
    d2 t[8];
    __device__ __forceinline__ double re(int i) const { return t[i].x; }
    __device__ __forceinline__ double im(int i) const { return t[i].y; }
    __amdgpu_buffer_rsrc_t rs;
    int voff;
    template <int FIRST> __device__ __forceinline__ void load4()
    {
#pragma unroll
        for (int i = FIRST; i < FIRST + 4; ++i) {
            const u32x4 u = __builtin_amdgcn_raw_buffer_load_b128(rs, voff, i * 16, 0);
            t[i] = (d2){__hiloint2double((int)u.y, (int)u.x), __hiloint2double((int)u.w, (int)u.z)};
        }
    }
    // entry `index` of a table of `entries` (the base is the same for the whole wave, the index is the lane's): the four
    // twiddles of levels 0..2 now, the last level's four from inside the transform -- all eight at once are 32 VGPRs
    // beside the 64 of the points, and where two passes in a row take per-lane tables (M = 4096) hipcc spills
    __device__ __forceinline__ void load(const double2 *table, int entries, int index)
    {
        rs = make_rsrc(table, (unsigned)entries * 128u);
        voff = index * 128;
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (LAZY) {
            load4<0>();
        } else {
            // the powers only: {w^8, w^4, w^2, w}; second_half() makes the other four from them
#pragma unroll
            for (int i : {0, 1, 2, 4}) {
                const u32x4 u = __builtin_amdgcn_raw_buffer_load_b128(rs, voff, i * 16, 0);
                t[i] = (d2){__hiloint2double((int)u.y, (int)u.x), __hiloint2double((int)u.w, (int)u.z)};
            }
        }
    }
    __device__ __forceinline__ void second_half()
    {
        if constexpr (LAZY) {
            __builtin_amdgcn_sched_barrier(0);
            load4<4>();
        } else {
            // w^2 W_8 and w W_16^e, e = 1, 2, 3, in double from the loaded powers (exp(-2 pi i e / 16) = (c, -s)): the
            // table's own entries are these products rounded once, these are rounded twice -- 1e-16 against a bar of 2e-7
            constexpr double C[4] = {1.0, 0.92387953251128675613, 0.70710678118654752440, 0.38268343236508977173};
            constexpr double S[4] = {0.0, 0.38268343236508977173, 0.70710678118654752440, 0.92387953251128675613};
            t[3] = (d2){(t[2].x + t[2].y) * C[2], (t[2].y - t[2].x) * C[2]};
#pragma unroll
            for (int e = 1; e < 4; ++e)
                t[4 + e] = (d2){__builtin_fma(t[4].y, S[e], t[4].x * C[e]), __builtin_fma(-t[4].x, S[e], t[4].y * C[e])};
        }
    }
};
typedef TwVT<true> TwV;
template <int LOGM, int P> __device__ __forceinline__ void set_prio()
{
    if constexpr (RO_F64R_PRIO && LOGM <= RO_F64R_PRIO_MAX_LOGM) __builtin_amdgcn_s_setprio(P);
}
struct TwS {
    d2 t[8];
    __device__ __forceinline__ double re(int i) const { return t[i].x; }
    __device__ __forceinline__ double im(int i) const { return t[i].y; }
    __device__ __forceinline__ void second_half() {}
    // `entry` must be the same for the whole wave
    __device__ __forceinline__ void load(const double2 *entry)
    {
        asm volatile("s_load_dwordx4 %0, %8, 0\n\t"
                     "s_load_dwordx4 %1, %8, 16\n\t"
                     "s_load_dwordx4 %2, %8, 32\n\t"
                     "s_load_dwordx4 %3, %8, 48\n\t"
                     "s_load_dwordx4 %4, %8, 64\n\t"
                     "s_load_dwordx4 %5, %8, 80\n\t"
                     "s_load_dwordx4 %6, %8, 96\n\t"
                     "s_load_dwordx4 %7, %8, 112\n\t"
                     "s_waitcnt lgkmcnt(0)"
                     : "=&s"(t[0]), "=&s"(t[1]), "=&s"(t[2]), "=&s"(t[3]), "=&s"(t[4]), "=&s"(t[5]), "=&s"(t[6]), "=&s"(t[7])
                     : "s"(entry)
                     : "memory");
    }
};

// the twist-free pass (D = 1, pass 0): the eight entries are {1, 1, 1, W_8, 1, W_16, W_8, W_16^3}
struct TwC {
    __device__ __forceinline__ void second_half() {}
};

// |X| as floats: src/WaterfallBackend.cpp:497-503 takes sqrt in double and narrows once.  Here: s = re^2 + im^2 in double;
// y = rsq_f32(float(s)), r = float(s) y is the root within a few ulp; with e = s - r^2 (the product is exact in double, one
// rounding) the root is r + e / (2 r) up to e^2 / (8 r^3) < 2^-46 r, and the float operation r + (e y) / 2 rounds that
// value once -- the correctly rounded float except where the root lies within ~1e-13 of a rounding boundary.  Outside the
// range where float(s) is a normal number with headroom: the plain way, decided ONCE for the sixteen values of the whole
// wave (a per-value branch costs more than the arithmetic; a per-lane one hipcc turns into both ways plus a select).
__device__ __forceinline__ void magnitudes16(const double *re, const double *im, float *out)
{
    double s[16];
    float sf[16];
    float lo = __builtin_inff(), hi = 0.f;            // (float min / max: 32-bit operations)
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        s[i] = __builtin_fma(im[i], im[i], re[i] * re[i]);
        sf[i] = (float)s[i];
        lo = __builtin_fminf(lo, sf[i]);
        hi = __builtin_fmaxf(hi, sf[i]);
    }
    // (a NaN passes min and max unseen and needs no watching: rsq, the products and the residual below hand it on, and NaN
    // is what sqrt(NaN) gives)
    const bool odd = !(lo > 0x1p-100f && hi < 0x1p100f);
    if (RO_F64R_SQRT_EXACT || __builtin_amdgcn_ballot_w64(odd) != 0) {
#pragma unroll
        for (int i = 0; i < 16; ++i) out[i] = (float)sqrt(s[i]);
    } else {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const float y = __builtin_amdgcn_rsqf(sf[i]);
            const float r = sf[i] * y;
            const double rd = (double)r;
            const float e = (float)__builtin_fma(-rd, rd, s[i]);
            out[i] = __builtin_fmaf(e * y, 0.5f, r);
        }
    }
}

// Sample formats: float32 and int16 pairs as everywhere in the library (Sample<>: a float pair per sample), and, for this
// kernel only, the reference's own struct Complex {double real; double imag;} (src/Backend.h:26-29) -- RO_IQ_F64 samples
// multiplied as (double)sample x (double)w like src/FFTBackend.cpp:229-232, with nothing narrowed on the way.
#define RO_FMT_F64 RO_IQ_F64
template <int FMT> struct Smp : Sample<FMT> {
    typedef v2f type;
};
template <> struct Smp<RO_FMT_F64> {
    typedef d2 type;
    static constexpr int BYTES = 16;
    static __device__ __forceinline__ d2 load(__amdgpu_buffer_rsrc_t r, int voff, int soff)
    {
        const u32x4 u = __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0);
        return (d2){__hiloint2double((int)u.y, (int)u.x), __hiloint2double((int)u.w, (int)u.z)};
    }
};

// The raw material of one sub-row in registers: the samples of PR blocks (16 each) and their window coefficients per
// thread, requested a whole barrier wait ahead of the fold (PR = D for D <= 2, 2 of the 4 blocks at D = 4, none for
// double samples: they would not fit, and the fold asks for what is missing itself)
template <int PR, int FMT> struct Raw {
    typename Smp<FMT>::type x[PR > 0 ? PR : 1][16];
    v4f w[PR > 0 ? PR : 1][4];
};
// (TS = samples between two slots of a thread, MS = samples between two blocks; voff_iq / voff_w = the lane's byte offsets
// into the samples and into the window table)
template <int TS, int MS, int PR, int FMT>
__device__ __forceinline__ void raw_load(Raw<PR, FMT> &r, const __amdgpu_buffer_rsrc_t &rs_iq, const __amdgpu_buffer_rsrc_t &rs_w,
                                         int voff_iq, int voff_w)
{
    using S = Smp<FMT>;
#pragma unroll
    for (int rr = 0; rr < PR; ++rr) {
#pragma unroll
        for (int n0 = 0; n0 < 16; ++n0) r.x[rr][n0] = S::load(rs_iq, voff_iq, (TS * n0 + MS * rr) * S::BYTES);
#pragma unroll
        for (int sg = 0; sg < 4; ++sg) {
            const u32x4 c = __builtin_amdgcn_raw_buffer_load_b128(rs_w, voff_w, (rr * 4 + sg) * TS * 16, 0);
            r.w[rr][sg] = (v4f){__uint_as_float(c.x), __uint_as_float(c.y), __uint_as_float(c.z), __uint_as_float(c.w)};
        }
    }
}

// The fold of sub-row q: slot n0 of thread t = sum_r W_D^(r q) w[i + M r] (x[i + M r] + i gain), i = t + T n0
// (src/FFTBackend.cpp:78-79: Q += gain; :229-232: the window multiply, double x (double)float).  Products of a float32 or
// int16 sample and a float32 coefficient are exact in double.  W_D^(r q) = (-i)^ph, ph = r q (4 / D) mod 4, is a swap and
// two sign flips of the sample AS IT ARRIVED (exact), the same for the whole workgroup: no copy of the fold per q (hipcc
// hoists what such copies have in common -- every conversion -- in front of them, and the doubles of a whole sub-row do
// not fit).
template <int D> struct Rot {
    unsigned swap[D], sign_r[D], sign_i[D];      // per term r: exchange the halves; xor masks of the new real / imaginary half
    double   gain_r[D], gain_i[D];               // what the reference's Q += gain becomes behind the rotation
    __device__ __forceinline__ Rot(int q, double gain)
    {
#pragma unroll
        for (int r = 0; r < D; ++r) {
            const int ph = ((r * q) % D) * (4 / D);
            swap[r] = ph & 1;
            sign_r[r] = (ph == 2 || ph == 3) ? 0x80000000u : 0u;
            sign_i[r] = (ph == 1 || ph == 2) ? 0x80000000u : 0u;
            // (xr + i (xi + g)) (-i)^ph: ph 0: imag + g; 1: real + g; 2: imag - g; 3: real - g
            gain_r[r] = ph == 1 ? gain : ph == 3 ? -gain : 0.0;
            gain_i[r] = ph == 0 ? gain : ph == 2 ? -gain : 0.0;
        }
    }
};
// (-i)^ph x as a pair of doubles
template <int D, int R> __device__ __forceinline__ d2 rotated(v2f x, const Rot<D> &rot)
{
    float fr = x.x, fi = x.y;
    if constexpr (R > 0 && D > 1) {
        if constexpr (D > 2) {
            const float a = rot.swap[R] ? fi : fr, b = rot.swap[R] ? fr : fi;
            fr = a;
            fi = b;
        }
        fr = __uint_as_float(__float_as_uint(fr) ^ rot.sign_r[R]);
        fi = __uint_as_float(__float_as_uint(fi) ^ rot.sign_i[R]);
    }
    return (d2){(double)fr, (double)fi};
}
template <int D, int R> __device__ __forceinline__ d2 rotated(d2 x, const Rot<D> &rot)
{
    double dr = x.x, di = x.y;
    if constexpr (R > 0 && D > 1) {
        if constexpr (D > 2) {
            const double a = rot.swap[R] ? di : dr, b = rot.swap[R] ? dr : di;
            dr = a;
            di = b;
        }
        dr = __hiloint2double(__double2hiint(dr) ^ (int)rot.sign_r[R], __double2loint(dr));
        di = __hiloint2double(__double2hiint(di) ^ (int)rot.sign_i[R], __double2loint(di));
    }
    return (d2){dr, di};
}
template <int D, bool GAIN, int R, typename X>
__device__ __forceinline__ void fold_term(double &sr, double &si, X x, float w, const Rot<D> &rot)
{
    const d2 xd = rotated<D, R>(x, rot);
    double xr = xd.x, xi = xd.y;
    if constexpr (GAIN) {
        if constexpr (D > 2) xr += rot.gain_r[R];
        xi += rot.gain_i[R];                      // (D <= 2: the rotation is a sign, the gain stays on the imaginary half)
    }
    const double wd = (double)w;
    if constexpr (R == 0) {
        sr = xr * wd;
        si = xi * wd;
    } else {
        sr = __builtin_fma(xr, wd, sr);
        si = __builtin_fma(xi, wd, si);
    }
}
// terms R0 + Rs of a slot (term 0 starts the sum, the others add to it)
template <int D, bool GAIN, int R0, typename X, int... Rs>
__device__ __forceinline__ void fold_slot(double &sr, double &si, const X (&x)[D], const float (&w)[D], const Rot<D> &rot,
                                          std::integer_sequence<int, Rs...>)
{
    (fold_term<D, GAIN, R0 + Rs>(sr, si, x[R0 + Rs], w[R0 + Rs], rot), ...);
}

// Blocks 0 .. PR - 1 from the registers of raw_load ...
template <int D, int PR, bool GAIN, int FMT>
__device__ __forceinline__ void fold_raw(double *re, double *im, const Raw<PR, FMT> &r, const Rot<D> &rot)
{
    typedef typename Smp<FMT>::type X;
#pragma unroll
    for (int n0 = 0; n0 < 16; ++n0) {
        X x[D];
        float w[D];
#pragma unroll
        for (int rr = 0; rr < PR; ++rr) {
            x[rr] = r.x[rr][n0];
            w[rr] = r.w[rr][n0 / 4][n0 % 4];
        }
        fold_slot<D, GAIN, 0>(re[n0], im[n0], x, w, rot, std::make_integer_sequence<int, PR>{});
        // (two slots at a time: left to itself the scheduler converts all the samples first and the doubles do not fit)
        if (n0 & 1) __builtin_amdgcn_sched_barrier(0);
    }
}
// ... blocks R0 .. D - 1 straight from memory, CH slots at a time (D = 4: two blocks of samples fit the registers ahead of
// time, the other two are asked for here; double samples: all of them)
template <int TS, int MS, int D, int R0, bool GAIN, int FMT>
__device__ __forceinline__ void fold_mem(double *re, double *im, const __amdgpu_buffer_rsrc_t &rs_iq,
                                         const __amdgpu_buffer_rsrc_t &rs_w, int voff_iq, int voff_w, const Rot<D> &rot)
{
    using S = Smp<FMT>;
    typedef typename S::type X;
    constexpr int CH = (sizeof(X) * (D - R0) > 32) ? 2 : 4;          // slots per request group: at most 32 VGPRs of samples
#pragma unroll
    for (int sg = 0; sg < 4; ++sg) {
        v4f w[D];
#pragma unroll
        for (int r = R0; r < D; ++r) {
            const u32x4 c = __builtin_amdgcn_raw_buffer_load_b128(rs_w, voff_w, (r * 4 + sg) * TS * 16, 0);
            w[r] = (v4f){__uint_as_float(c.x), __uint_as_float(c.y), __uint_as_float(c.z), __uint_as_float(c.w)};
        }
#pragma unroll
        for (int h = 0; h < 4 / CH; ++h) {
            X x[CH][D];
#pragma unroll
            for (int r = R0; r < D; ++r)
#pragma unroll
                for (int e = 0; e < CH; ++e) x[e][r] = S::load(rs_iq, voff_iq, (TS * (4 * sg + CH * h + e) + MS * r) * S::BYTES);
#pragma unroll
            for (int e = 0; e < CH; ++e) {
                float we[D];
#pragma unroll
                for (int r = R0; r < D; ++r) we[r] = w[r][CH * h + e];
                fold_slot<D, GAIN, R0>(re[4 * sg + CH * h + e], im[4 * sg + CH * h + e], x[e], we, rot,
                                       std::make_integer_sequence<int, D - R0>{});
            }
            if constexpr (CH < 4) __builtin_amdgcn_sched_barrier(0);
        }
    }
}

// v_permlane32_swap / v_permlane16_swap of two doubles: a 2 x 2 transposition between lane bit 5 (4) and the pair
template <int BIT> __device__ __forceinline__ void lane_swap(double &a, double &b)
{
    unsigned alo = (unsigned)__double2loint(a), ahi = (unsigned)__double2hiint(a);
    unsigned blo = (unsigned)__double2loint(b), bhi = (unsigned)__double2hiint(b);
    if constexpr (BIT == 5) {
        const auto l = __builtin_amdgcn_permlane32_swap(alo, blo, false, false);
        const auto h = __builtin_amdgcn_permlane32_swap(ahi, bhi, false, false);
        alo = l[0]; blo = l[1]; ahi = h[0]; bhi = h[1];
    } else {
        const auto l = __builtin_amdgcn_permlane16_swap(alo, blo, false, false);
        const auto h = __builtin_amdgcn_permlane16_swap(ahi, bhi, false, false);
        alo = l[0]; blo = l[1]; ahi = h[0]; bhi = h[1];
    }
    a = __hiloint2double((int)ahi, (int)alo);
    b = __hiloint2double((int)bhi, (int)blo);
}

struct Args {
    const void    *iq;
    const float   *window_k;     // [D][4][T][4]: w[t + T (4 sg + e) + M r] at ((r 4 + sg) T + t) 4 + e
    const double2 *tw0;          // [D][8]
    const double2 *tw1;          // [D][16][8]
    const double2 *tw2;          // [D][256][8]
    const double2 *tw3;          // [D][256 R3][2] (R3 > 1)
    float         *rows_out;
    int64_t        first_row, rows, row_stride;
    int            hop;
    double         gain;
    unsigned long long *stamps;  // diagnostic builds only (RO_F64R_STAMPS), else nullptr
    int            spectra;      // 1: rows_out takes {re, im} float pairs per bin, unshifted (row_stride counts pairs)
};

template <int LOGM, int D, int FMT, bool GAIN, int LOGB = 0, bool SPEC = false>
__global__ __launch_bounds__((1 << LOGM) / 16, 4) void f64r_kernel(Args a)
{
    using G = Geo<LOGM>;
    using S = Smp<FMT>;
    [[maybe_unused]] constexpr int M = G::M, T = G::T, R3 = G::R3, Q = G::Q, ST = G::ST, N = M * D;
    // LOGB > 0: bins NB = 4096 >> LOGB (2048 ... 256), B = 2^LOGB ROWS in the M = 4096 workgroup.  The workgroup's index
    // i = n2 + 16 j1 + 256 n0 carries the row in the low bits of n2 = b + B m and sample m + R j1 + 16 R n0 of row b
    // (R = 16 / B): passes 0 and 1 are the 4096-point transform's own, with its own tables, and the row's last digit m is
    // what the first LV = log2 R levels of pass 2 transform -- with the 4096-point table again, W_4096^(K1 B m) =
    // W_NB^(K1 m); slot p of the result is row p mod B, k2 = bitrev(p / B).  A "row" of the loop below is then a tile of B.
    static_assert(LOGB == 0 || (LOGM == 12 && D == 1 && LOGB <= 4), "rows are batched into the M = 4096 workgroup only");
    [[maybe_unused]] constexpr int B = 1 << LOGB, NB = N >> LOGB, TS = T >> LOGB, LV = 4 - LOGB;
    // blocks of the next sub-row's samples requested a barrier ahead (double samples: none, 64 VGPRs a block)
    constexpr int PR = FMT == RO_FMT_F64 ? 0 : D <= 2 ? D : 2;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    double *plane = reinterpret_cast<double *>(smem);
    float *image = reinterpret_cast<float *>(smem);
#if RO_F64R_PLAIN_READS
    typedef const volatile __attribute__((address_space(3))) double lds_vdouble;
    lds_vdouble *rplane = (lds_vdouble *)plane;
#else
    const double *rplane = plane;
#endif

    // XCD-aware placement (speed only): workgroups b and b + 8 share an XCD under round-robin dispatch; each XCD takes
    // a contiguous run of rows, its workgroups take the D sub-rows of consecutive rows at the same time
    const int64_t units = (a.rows + B - 1) >> LOGB;
    const int64_t per_xcd = (units + 7) / 8;
    const int64_t xcd_first = (int64_t)(blockIdx.x & 7) * per_xcd;
    const int64_t xcd_end = xcd_first + per_xcd < units ? xcd_first + per_xcd : units;
    const int slots = gridDim.x >> 3;                  // a multiple of D
    const int slot = blockIdx.x >> 3;
    const int q = slot % D;
    const int64_t row_step = slots / D;
    int64_t row = xcd_first + slot / D;
    if (row >= xcd_end) return;

    const int t = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const char *iq = reinterpret_cast<const char *>(a.iq);

    // Thread roles -- pass 1 (k0, n2 = u), pass 2 (k0, k1, n3) with u = n3 16 + k1 -- and the LDS cells that follow from them
    // (see the header) are recomputed from the thread number where they are used, behind an empty asm statement: as
    // loop invariants hipcc keeps some twenty of them in VGPRs for the whole kernel and spills the transform's.
    auto fresh = [&]() {
        int x = t;
        asm volatile("" : "+v"(x));
        return x;
    };
    const double2 *tw0 = a.tw0 + q * 8;
    const double2 *tw1 = a.tw1 + q * 16 * 8;
    const double2 *tw2 = a.tw2 + q * 256 * 8;
    const Rot<D> rot(q, a.gain);

    [[maybe_unused]] unsigned long long st_acc[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, st_prev = 0;
    auto stamp = [&](int k) {
        if constexpr (RO_F64R_STAMPS) {
            unsigned long long now;
            __builtin_amdgcn_sched_barrier(0);
            asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(now)::"memory");
            __builtin_amdgcn_sched_barrier(0);
            if (k >= 0) st_acc[k] += now - st_prev;
            st_prev = now;
        }
    };
    stamp(-1);

    // ---- the read-out of a finished image: column (k + N/2) mod N of the row holds |X[k]| (src/WaterfallBackend.cpp:
    // 492-505).  A slot's share of the bin only has bits above the lane's share, and N/2 flips the top one.
    // position p of a thread -> its share of the sub-row bin (see the header / tools/r6/emu_f64r.py)
    auto slot_bin = [](int p) constexpr {
        if (R3 == 1) return 256 * brev<4>(p);
        if (R3 == 2) return 512 * brev<3>(p & 7) + 4096 * (p >> 3);
        return 1024 * (((p >> 1) & 1) + 2 * (p & 1)) + 4096 * (((p >> 3) & 1) + 2 * ((p >> 2) & 1));
    };
    // D = 1: 16-byte stores.  lane = c + 4 ul (k0 = 4 c + e, e the dword of the store; ul = u & 15 = k1): a wave's store
    // is 1 KiB of consecutive columns.  code = wave + waves it = (u >> 4) + R3 s.
    // D > 1: 4-byte stores (the sub-row owns every D-th column): k0 = lane & 15, u = (lane >> 4) | (wave << 2), s = it.
    float m[16];
    auto readout_lds = [&]() {
        const int lane = fresh() & 63;
        if constexpr (D == 1) {
            const int rc = lane & 3, ul = lane >> 2;
#pragma unroll
            for (int it = 0; it < 4; ++it) {
                const int code = wave + (T / 64) * it, uh = code % R3, s = code / R3, ru = ul + 16 * uh;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int rk0 = 4 * rc + e;
                    int cell;
                    if constexpr (R3 == 1) cell = 2 * rk0 * ST + (s ^ (rc & 1)) * 16 + ((ru + 8 * (rc >> 1)) & 15);
                    else cell = 2 * rk0 * ST + s * Q + ((ru + 2 * rk0) & (Q - 1));
                    m[4 * it + e] = image[cell];
                }
            }
        } else {
            const int rk0 = lane & 15, ru = (lane >> 4) | (wave << 2);
            const int imr = 2 * rk0 * ST + ((ru + 2 * rk0) & (Q - 1));
#pragma unroll
            for (int s = 0; s < 16; ++s) m[s] = image[imr + Q * s];
        }
    };
    // SPEC (ro_stft_spectra_resident): the image holds the real parts, then the imaginary parts, of the transform narrowed to
    // float; bin k of the row goes to floats 2 k + part of its line of row_stride {re, im} pairs, unshifted (k = 0 is DC:
    // what fftw_execute leaves in out_, src/FFTBackend.cpp:236), as 4-byte stores on the default cache policy (the two
    // halves of a pair meet in L2)
    auto spec_store = [&](int64_t prow, int part) {
        const int lane = fresh() & 63;
        if constexpr (LOGB > 0) {
            const int rc = lane & 3, ul = lane >> 2;
#pragma unroll
            for (int it = 0; it < 4; ++it) {
                const int s = wave + 4 * it;
                const int64_t orow = prow * B + (s & (B - 1));
                const __amdgpu_buffer_rsrc_t rsb = make_rsrc(a.rows_out + orow * a.row_stride * 2, orow < a.rows ? NB * 8 : 0);
                int sb = 0;
#pragma unroll
                for (int p = 0; p < 16; ++p) sb = s == p ? 256 * brev<LV>(p >> LOGB) : sb;
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(m[4 * it + e]), rsb, (4 * rc + 16 * ul + sb + e) * 8 + part * 4, 0, 0);
            }
        } else {
            const __amdgpu_buffer_rsrc_t rs = make_rsrc(a.rows_out + prow * a.row_stride * 2, N * 8);
            if constexpr (D == 1) {
                const int rc = lane & 3, ul = lane >> 2;
#pragma unroll
                for (int it = 0; it < 4; ++it) {
                    const int code = wave + (T / 64) * it, uh = code % R3, s = code / R3;
                    const int g = R3 == 4 ? (uh >> 1) + 2 * (uh & 1) : uh;
                    int sb = 0;
#pragma unroll
                    for (int p = 0; p < 16; ++p) sb = s == p ? slot_bin(p) : sb;
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(m[4 * it + e]), rs, (4 * rc + 16 * ul + 256 * g + sb + e) * 8 + part * 4, 0, 0);
                }
            } else {
                const int rk0 = lane & 15, ru = (lane >> 4) | (wave << 2);
                const int rn3 = ru >> 4, rg = R3 == 4 ? (rn3 >> 1) + 2 * (rn3 & 1) : rn3;
                const int voff = (q + D * (rk0 + 16 * (ru & 15) + 256 * rg)) * 8 + part * 4;
#pragma unroll
                for (int s = 0; s < 16; ++s)
                    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(m[s]), rs, voff, D * slot_bin(s) * 8, 0);
            }
        }
    };
    auto readout_store = [&](int64_t prow) {
        [[maybe_unused]] const __amdgpu_buffer_rsrc_t rs = make_rsrc(a.rows_out + prow * a.row_stride, LOGB == 0 ? N * 4 : 0);
        const int lane = fresh() & 63;
        if constexpr (LOGB > 0) {
            // slot s = wave + 4 it of the image belongs to row s mod B of the tile and holds its bins 256 bitrev(s / B) + 0 .. 255
            const int rc = lane & 3, ul = lane >> 2;
#pragma unroll
            for (int it = 0; it < 4; ++it) {
                const int s = wave + 4 * it;
                const int64_t orow = prow * B + (s & (B - 1));
                const __amdgpu_buffer_rsrc_t rsb = make_rsrc(a.rows_out + orow * a.row_stride, orow < a.rows ? NB * 4 : 0);
                int sb = 0;
#pragma unroll
                for (int p = 0; p < 16; ++p) sb = s == p ? 256 * brev<LV>(p >> LOGB) : sb;
                const int col = ((4 * rc + 16 * ul + sb) + NB / 2) & (NB - 1);
                buf_store_f4(m[4 * it], m[4 * it + 1], m[4 * it + 2], m[4 * it + 3], rsb, col * 4, 0);
            }
        } else if constexpr (D == 1) {
            const int rc = lane & 3, ul = lane >> 2;
#pragma unroll
            for (int it = 0; it < 4; ++it) {
                const int code = wave + (T / 64) * it, uh = code % R3, s = code / R3;
                // bin of dword 0: 4 rc + 16 ul + 256 g'(uh) + slot_bin(s); g' of writer n3 = uh
                const int g = R3 == 4 ? (uh >> 1) + 2 * (uh & 1) : uh;
                int sb = 0;                               // slot_bin(s) for a run-time s: a 16-way select the compiler folds per it
#pragma unroll
                for (int p = 0; p < 16; ++p) sb = s == p ? slot_bin(p) : sb;
                const int col = ((4 * rc + 16 * ul + 256 * g + sb) + N / 2) & (N - 1);
                buf_store_f4(m[4 * it], m[4 * it + 1], m[4 * it + 2], m[4 * it + 3], rs, col * 4, 0);
            }
        } else {
            const int rk0 = lane & 15, ru = (lane >> 4) | (wave << 2);
            const int rn3 = ru >> 4, rg = R3 == 4 ? (rn3 >> 1) + 2 * (rn3 & 1) : rn3;
            const int voff = (q + D * (rk0 + 16 * (ru & 15) + 256 * rg)) * 4;
#pragma unroll
            for (int s = 0; s < 16; ++s)
                __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(m[s]), rs, voff, ((D * slot_bin(s)) ^ (N / 2)) * 4, RO_F64R_STORE_AUX_D);
        }
    };

    const __amdgpu_buffer_rsrc_t rs_w = make_rsrc(a.window_k, (unsigned)NB * 4);
    auto iq_rsrc = [&](int64_t r, bool valid) {
        if constexpr (LOGB == 0) {
            return make_rsrc(iq + (a.first_row + r) * (int64_t)a.hop * S::BYTES, valid ? (unsigned)N * S::BYTES : 0u);
        } else {
            // the tile's rows as ONE range (they overlap or follow one another): rows r B .. r B + vb - 1, vb = the valid ones
            const int64_t left = a.rows - r * B;
            const unsigned vb = left < B ? (unsigned)left : (unsigned)B;
            return make_rsrc(iq + (a.first_row + r * B) * (int64_t)a.hop * S::BYTES,
                             valid ? ((vb - 1u) * (unsigned)a.hop + (unsigned)NB) * S::BYTES : 0u);
        }
    };
    // the lane's place in the samples and in the window table, from its thread number
    auto lane_off = [&](int th, int &voff_iq, int &voff_w) {
        if constexpr (LOGB == 0) {
            voff_iq = th * S::BYTES;
            voff_w = th * 16;
        } else {
            const int b = th & (B - 1), ti = ((th & 15) >> LOGB) + (16 >> LOGB) * (th >> 4);
            voff_iq = (b * a.hop + ti) * S::BYTES;
            voff_w = ti * 16;
        }
    };
    Raw<PR, FMT> raw;
    {
        int vi, vw;
        lane_off(t, vi, vw);
        raw_load<TS, M, PR, FMT>(raw, iq_rsrc(row, true), rs_w, vi, vw);
    }

    constexpr int TOUCHES = D < 2 ? D : 2;               // x T x 128 bytes of new samples touched ahead (hop <= N / 2 whole)
    unsigned touch[TOUCHES] = {};
    for (;;) {
        const int64_t next = row + row_step;
        const bool has_next = next < xcd_end;
#pragma unroll
        for (int i = 0; i < TOUCHES; ++i) asm volatile("" ::"v"(touch[i]));   // the "use" of the touches below (long complete)
        // (nothing of the fold in front of this point: hoisted above the barrier and the read-out, its conversions hold
        // the samples twice)
        __builtin_amdgcn_sched_barrier(0);
        double re[16], im[16];
        // ---- the fold
        if constexpr (PR > 0) fold_raw<D, PR, GAIN, FMT>(re, im, raw, rot);
        if constexpr (PR < D) {
            int vi, vw;
            lane_off(fresh(), vi, vw);
            fold_mem<TS, M, D, PR, GAIN, FMT>(re, im, iq_rsrc(row, true), rs_w, vi, vw, rot);
        }
        stamp(3);
        // ---- pass 0, and the real parts leave for exchange 1
        if constexpr (D == 1) {
            TwC tw;
            twisted16(re, im, tw);
        } else {
            TwS tw;
            tw.load(tw0);
            twisted16(re, im, tw);
        }
        double xr[16], xi[16];
        const int x1w = fresh();                         // exchange 1: writer cell n1 = t, reader cell k0 ST + u
        const int x1r = (x1w / Q) * ST + x1w % Q;
#pragma unroll
        for (int k = 0; k < 16; ++k) plane[x1w + k * ST] = re[brev<4>(k)];
        stamp(4);
        wg_sync();                                      // (b)
        [[maybe_unused]] TwS tw1s;                       // M = 16384: the wave is one k0, its pass-1 table entry in SGPRs
        [[maybe_unused]] TwVT<!RO_F64R_EARLY_TABLES> tw1v;   // M < 16384: per lane (its powers asked for here)
        if constexpr (Q == 64) tw1s.load(tw1 + wave * 8);
        else if constexpr (RO_F64R_EARLY_TABLES) tw1v.load(tw1, 16, fresh() / Q);
#pragma unroll
        for (int j = 0; j < 16; ++j) xr[j] = rplane[x1r + Q * j];
        stamp(5);
        wg_sync();                                      // (c) everyone has its real parts
#pragma unroll
        for (int k = 0; k < 16; ++k) plane[x1w + k * ST] = im[brev<4>(k)];
        stamp(6);
        wg_sync();                                      // (d)
#pragma unroll
        for (int j = 0; j < 16; ++j) xi[j] = rplane[x1r + Q * j];
        asm volatile("" ::: "memory");
        set_prio<LOGM, 3>();
        // this thread from here on: (k0, u) in pass 1, (k0, k1, n3) in pass 2, g' = bitrev(n3) in pass 3
        const int tr_ = fresh();
        const int k0 = tr_ / Q, u = tr_ % Q, k1 = u & 15, n3 = u >> 4;
        const int K1 = k0 + 16 * k1;
        constexpr bool EARLY = RO_F64R_EARLY_TABLES;
        TwVT<!EARLY> tw2v;
        [[maybe_unused]] d2 a1, a2;                      // pass 3's {a', a'^2}, a' = W_N^(q + D (K1 + 256 g'))
        auto load_tw3 = [&]() {
            if constexpr (R3 > 1) {
                const int gp = R3 == 4 ? (n3 >> 1) + 2 * (n3 & 1) : n3;
                const __amdgpu_buffer_rsrc_t rs = make_rsrc(a.tw3 + (int64_t)q * 256 * R3 * 2, 256 * R3 * 32);
                const u32x4 c1 = __builtin_amdgcn_raw_buffer_load_b128(rs, (K1 * R3 + gp) * 32, 0, 0);
                const u32x4 c2 = __builtin_amdgcn_raw_buffer_load_b128(rs, (K1 * R3 + gp) * 32, 16, 0);
                a1 = (d2){__hiloint2double((int)c1.y, (int)c1.x), __hiloint2double((int)c1.w, (int)c1.z)};
                a2 = (d2){__hiloint2double((int)c2.y, (int)c2.x), __hiloint2double((int)c2.w, (int)c2.z)};
            }
        };
        if constexpr (EARLY) {
            if constexpr (LV > 0) tw2v.load(tw2, 256, K1);
            load_tw3();
        }
        // ---- pass 1.  From here to the completed image a wave touches its own territories only.
        if constexpr (Q == 64) {
            twisted16(xr, xi, tw1s);
        } else {
            if constexpr (!RO_F64R_EARLY_TABLES) tw1v.load(tw1, 16, k0);
            twisted16(xr, xi, tw1v);
        }
        stamp(7);
        set_prio<LOGM, 2>();
        // ---- exchange 2 (one wave's LDS instructions execute in order: no wait between its writes and its reads); the first
        // half of pass 2's table entry is asked for in front of it
        if constexpr (!EARLY && LV > 0) tw2v.load(tw2, 256, K1);
        const int x2w = k0 * ST + u, x2r = k0 * ST + k1 * G::S2 + n3;
#pragma unroll
        for (int k = 0; k < 16; ++k) plane[x2w + k * G::S2] = xr[brev<4>(k)];
        asm volatile("" ::: "memory");
#pragma unroll
        for (int j = 0; j < 16; ++j) re[j] = rplane[x2r + R3 * j];
        asm volatile("" ::: "memory");
#pragma unroll
        for (int k = 0; k < 16; ++k) plane[x2w + k * G::S2] = xi[brev<4>(k)];
        asm volatile("" ::: "memory");
#pragma unroll
        for (int j = 0; j < 16; ++j) im[j] = rplane[x2r + R3 * j];
        asm volatile("" ::: "memory");
        stamp(8);
        // ---- pass 2 (rows batched into the workgroup: what is left of their transform)
        if constexpr (LOGB == 0) twisted16(re, im, tw2v);
        else twisted16_top<LV>(re, im, tw2v);
        set_prio<LOGM, 1>();
        stamp(9);
        if constexpr (R3 > 1) {
            // ---- exchange 3 without LDS: the R3 threads of one (k0, k1) sit 16 lanes apart, so a 2 x 2 transposition
            // between a lane bit and a position bit is one v_permlane*_swap per dword
            if constexpr (R3 == 4) {
#pragma unroll
                for (int p = 0; p < 8; ++p) {
                    lane_swap<5>(re[p], re[p + 8]);
                    lane_swap<5>(im[p], im[p + 8]);
                }
#pragma unroll
                for (int p = 0; p < 16; ++p)
                    if (!(p & 4)) {
                        lane_swap<4>(re[p], re[p + 4]);
                        lane_swap<4>(im[p], im[p + 4]);
                    }
            } else {
#pragma unroll
                for (int p = 0; p < 8; ++p) {
                    lane_swap<4>(re[p], re[p + 8]);
                    lane_swap<4>(im[p], im[p + 8]);
                }
            }
            stamp(10);
            // ---- pass 3: butterfly i has twist theta = a' W_16^i, a' = W_N^(q + D (K1 + 256 g')) (table: {a', a'^2})
            if constexpr (!EARLY) load_tw3();
            constexpr double C16[8] = {1.0, 0.92387953251128675613, 0.70710678118654752440, 0.38268343236508977173,
                                       0.0, -0.38268343236508977173, -0.70710678118654752440, -0.92387953251128675613};
            constexpr double S16[8] = {0.0, 0.38268343236508977173, 0.70710678118654752440, 0.92387953251128675613,
                                       1.0, 0.92387953251128675613, 0.70710678118654752440, 0.38268343236508977173};
#pragma unroll
            for (int i = 0; i < 16 / R3; ++i) {
                // theta = a1 (c - i s), c + i s = exp(2 pi i i / 16)
                const double c = C16[i], sn = S16[i];
                const double tr = i == 0 ? a1.x : a1.x * c + a1.y * sn, ti = i == 0 ? a1.y : a1.y * c - a1.x * sn;
                if constexpr (R3 == 2) {
                    const int r = brev<3>(i);                       // inputs n3 = 0, 1 at positions r, 8 + r
                    bfly<0>(re[r], im[r], re[8 + r], im[8 + r], tr, ti);
                } else {
                    // inputs n3 at positions 8 (n3 >> 1) + 4 (n3 & 1) + 2 (i & 1) + (i >> 1); theta^2 = a2 W_8^i
                    const int b = 2 * (i & 1) + (i >> 1);
                    const double c2 = C16[2 * i], s2 = S16[2 * i];
                    const double ur = i == 0 ? a2.x : a2.x * c2 + a2.y * s2, ui = i == 0 ? a2.y : a2.y * c2 - a2.x * s2;
                    bfly<0>(re[b], im[b], re[b + 8], im[b + 8], ur, ui);                 // n3 = 0, 2
                    bfly<0>(re[b + 4], im[b + 4], re[b + 12], im[b + 12], ur, ui);       // n3 = 1, 3
                    bfly<0>(re[b], im[b], re[b + 4], im[b + 4], tr, ti);                 // n3 = 0, 1
                    bfly<1>(re[b + 8], im[b + 8], re[b + 12], im[b + 12], tr, ti);       // n3 = 2, 3
                }
            }
            stamp(11);
        }
        set_prio<LOGM, 0>();
        // ---- magnitudes into the image (own territory): cell 2 k0 ST + s Q + rot(u), see the header
        int imw;
        if constexpr (R3 == 1) imw = 2 * k0 * ST + ((u + 8 * (k0 >> 3)) & 15);       // + (s ^ ((k0 >> 2) & 1)) 16
        else imw = 2 * k0 * ST + ((u + 2 * k0) & (Q - 1));                            // + s Q
        const int imw_flip = R3 == 1 ? (k0 >> 2) & 1 : 0;
        float mg[16];
        auto image_write = [&]() {
#pragma unroll
            for (int s = 0; s < 16; ++s) {
                if constexpr (R3 == 1) image[imw + 16 * (s ^ imw_flip)] = mg[s];
                else image[imw + Q * s] = mg[s];
            }
        };
        if constexpr (SPEC) {
            // the real parts take the image's way out first (two barriers more per sub-row), the imaginary parts the
            // magnitudes' place below
#pragma unroll
            for (int s = 0; s < 16; ++s) mg[s] = (float)re[s];
            image_write();
            wg_sync();
            readout_lds();
            wg_sync();
            spec_store(row, 0);
#pragma unroll
            for (int s = 0; s < 16; ++s) mg[s] = (float)im[s];
        } else {
            magnitudes16(re, im, mg);
        }
        image_write();
        const float last = mg[15];
        // The hop new samples of a later sub-row are touched (one dword per 128-byte line, value unused: it is "used" in front
        // of the next fold) so that the requests below find them in L2: they come from HBM, every other byte of the row from
        // L2.  Here, because loads return in order: behind this point nothing waits for a load before the next fold.
        {
            const int64_t far = next + row_step;
            if (RO_F64R_TOUCH && far < xcd_end) {
                const int64_t s0 = (a.first_row + far) * (int64_t)a.hop + (N - a.hop);
                const __amdgpu_buffer_rsrc_t rs_new = make_rsrc(iq + s0 * S::BYTES, (unsigned)a.hop * S::BYTES);
                const int tt = after(fresh(), last);
#pragma unroll
                for (int i = 0; i < TOUCHES; ++i) touch[i] = __builtin_amdgcn_raw_buffer_load_b32(rs_new, tt * 128, i * T * 128, 0);
            }
        }
        stamp(12);
        __builtin_amdgcn_sched_barrier(0);
        // ---- the next sub-row's samples and window: asked for now, needed behind the barrier and the read-out.  (The
        // requests may not start before the last magnitude exists: their registers are the transform's.)
        {
            int vi, vw;
            lane_off(after(fresh(), last), vi, vw);
            raw_load<TS, M, PR, FMT>(raw, iq_rsrc(has_next ? next : row, has_next), rs_w, vi, vw);
        }
        wg_sync();                                      // (e) the image of this sub-row is complete
        stamp(13);
        // ---- the image: out of LDS, then LDS is free for the next sub-row's exchange, then on its way to the row
        readout_lds();
        stamp(0);
        wg_sync();                                      // (a) the image has been read
        stamp(1);
        if constexpr (SPEC) spec_store(row, 1);
        else readout_store(row);
        stamp(2);
        if constexpr (RO_F64R_STAMPS) st_acc[15] += 1;
        if (!has_next) break;
        row = next;
    }
    if constexpr (RO_F64R_STAMPS) {
        if (a.stamps && (t & 63) == 0)
            for (int k = 0; k < 16; ++k) a.stamps[((size_t)blockIdx.x * (T / 64) + wave) * 16 + k] = st_acc[k];
    }
}

template <int LOGM, int D, int FMT, int LOGB = 0> static hipError_t launch_one(const Args &a, hipStream_t s)
{
    using G = Geo<LOGM>;
    static std::mutex lock;
    static int cus_of[64];
    static bool ready[64];
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    if (dev < 0 || dev >= 64) return hipErrorInvalidDevice;
    int cus;
    {
        std::lock_guard<std::mutex> g(lock);
        if (!ready[dev]) {
            const void *fn[2] = {reinterpret_cast<const void *>(&f64r_kernel<LOGM, D, FMT, false, LOGB>),
                                 reinterpret_cast<const void *>(&f64r_kernel<LOGM, D, FMT, true, LOGB>)};
            for (int i = 0; i < 2; ++i)
                if ((e = hipFuncSetAttribute(fn[i], hipFuncAttributeMaxDynamicSharedMemorySize, G::LDS_BYTES)) != hipSuccess) return e;
            if ((e = hipDeviceGetAttribute(&cus_of[dev], hipDeviceAttributeMultiprocessorCount, dev)) != hipSuccess) return e;
            ready[dev] = true;
        }
        cus = cus_of[dev];
    }
    // persistent grid: 1024 / T workgroups per CU, per XCD a multiple of D, never more than the XCD's share of sub-rows
    const int64_t per_xcd = (((a.rows + (1 << LOGB) - 1) >> LOGB) + 7) / 8;      // (LOGB > 0: tiles of 2^LOGB rows)
    int64_t slots = (int64_t)(cus / 8) * (1024 / G::T);
    if (slots > per_xcd * D) slots = per_xcd * D;
    slots = slots / D * D;
    if (slots < D) slots = D;
    // (the reference's "iq_gain" is 0 in every shipped config: the additions exist only in the kernel that needs them)
    if (a.spectra) {
        // (one instantiation per plan and format: with a gain of 0.0 the additions change nothing)
        static std::mutex slock;
        static bool sready[64];
        {
            std::lock_guard<std::mutex> g(slock);
            if (!sready[dev]) {
                if ((e = hipFuncSetAttribute(reinterpret_cast<const void *>(&f64r_kernel<LOGM, D, FMT, true, LOGB, true>),
                                             hipFuncAttributeMaxDynamicSharedMemorySize, G::LDS_BYTES)) != hipSuccess) return e;
                sready[dev] = true;
            }
        }
        hipLaunchKernelGGL((f64r_kernel<LOGM, D, FMT, true, LOGB, true>), dim3((unsigned)(slots * 8)), dim3(G::T), G::LDS_BYTES, s, a);
    } else if (a.gain != 0.0) hipLaunchKernelGGL((f64r_kernel<LOGM, D, FMT, true, LOGB>), dim3((unsigned)(slots * 8)), dim3(G::T), G::LDS_BYTES, s, a);
    else hipLaunchKernelGGL((f64r_kernel<LOGM, D, FMT, false, LOGB>), dim3((unsigned)(slots * 8)), dim3(G::T), G::LDS_BYTES, s, a);
    return hipGetLastError();
}

template <int FMT> static hipError_t launch_fmt(int m_log2, int dec, int logb, const Args &a, hipStream_t s)
{
#define RO_F64R_CASE(LM, DD) \
    if (m_log2 == LM && dec == DD) return launch_one<LM, DD, FMT>(a, s)
    if (logb == 1) return launch_one<12, 1, FMT, 1>(a, s);
    if (logb == 2) return launch_one<12, 1, FMT, 2>(a, s);
    if (logb == 3) return launch_one<12, 1, FMT, 3>(a, s);
    if (logb == 4) return launch_one<12, 1, FMT, 4>(a, s);
    RO_F64R_CASE(12, 1);
    RO_F64R_CASE(13, 1);
    RO_F64R_CASE(14, 1);
    RO_F64R_CASE(14, 2);
    RO_F64R_CASE(14, 4);
#undef RO_F64R_CASE
    return hipErrorInvalidValue;
}

static bool plan(int bins, int &m_log2, int &dec, int &logb)
{
    logb = 0;
    switch (bins) {
    case 256: m_log2 = 12; dec = 1; logb = 4; return true;       // 16 rows in the 4096-point workgroup
    case 512: m_log2 = 12; dec = 1; logb = 3; return true;
    case 1024: m_log2 = 12; dec = 1; logb = 2; return true;
    case 2048: m_log2 = 12; dec = 1; logb = 1; return true;
    case 4096: m_log2 = 12; dec = 1; return true;
    case 8192: m_log2 = 13; dec = 1; return true;
    case 16384: m_log2 = 14; dec = 1; return true;
    case 32768: m_log2 = 14; dec = 2; return true;
    case 65536: m_log2 = 14; dec = 4; return true;
    default: return false;
    }
}

static double2 wexp(long double num, long double den)       // exp(-2 pi i num / den), correctly rounded from long double
{
    const long double ang = -2.0L * 3.14159265358979323846264338327950288L * num / den;
    return make_double2((double)cosl(ang), (double)sinl(ang));
}

// {w^8, w^4, w^2, w^2 W16^2, w, w W16, w W16^2, w W16^3} for w = W_n^e, exponents kept as exact integers over 16 n
static void tw8(int64_t n, int64_t e, double2 *out)
{
    const int64_t den = 16 * n;
    auto wp = [&](int p, int c) { return wexp((long double)((16 * e * p + c * n) % den), (long double)den); };
    out[0] = wp(8, 0);
    out[1] = wp(4, 0);
    out[2] = wp(2, 0);
    out[3] = wp(2, 2);
    out[4] = wp(1, 0);
    out[5] = wp(1, 1);
    out[6] = wp(1, 2);
    out[7] = wp(1, 3);
}

}  // namespace f64r

bool f64reg_supported(int bins)
{
    int m, d, b;
    return f64r::plan(bins, m, d, b);
}

void f64reg_tables(int bins, const float *window, F64RegTables &t)
{
    int m_log2 = 0, D = 0, logb = 0;
    if (!f64r::plan(bins, m_log2, D, logb)) return;
    // (rows batched into the workgroup, logb > 0: the window in the order of ONE row's threads, T = bins / 16; the twiddle
    // tables are the 4096-point ones)
    const int M = (1 << m_log2) >> logb, T = M / 16, R3 = (1 << m_log2) / 4096;
    t.window_k.assign((size_t)bins, 0.f);
    for (int r = 0; r < D; ++r)
        for (int sg = 0; sg < 4; ++sg)
            for (int th = 0; th < T; ++th)
                for (int e = 0; e < 4; ++e)
                    t.window_k[(((size_t)r * 4 + sg) * T + th) * 4 + e] = window[th + T * (4 * sg + e) + M * r];
    t.tw0.resize((size_t)D * 8);
    t.tw1.resize((size_t)D * 16 * 8);
    t.tw2.resize((size_t)D * 256 * 8);
    t.tw3.assign(R3 > 1 ? (size_t)D * 256 * R3 * 2 : 0, make_double2(0, 0));
    for (int q = 0; q < D; ++q) {
        f64r::tw8(16 * D, q, &t.tw0[(size_t)q * 8]);
        for (int k0 = 0; k0 < 16; ++k0) f64r::tw8(256 * D, q + D * k0, &t.tw1[((size_t)q * 16 + k0) * 8]);
        for (int K1 = 0; K1 < 256; ++K1) f64r::tw8((int64_t)4096 * D, q + D * K1, &t.tw2[((size_t)q * 256 + K1) * 8]);
        if (R3 > 1)
            for (int K1 = 0; K1 < 256; ++K1)
                for (int g = 0; g < R3; ++g) {
                    const int64_t e = q + (int64_t)D * (K1 + 256 * g);
                    double2 *o = &t.tw3[(((size_t)q * 256 + K1) * R3 + g) * 2];
                    o[0] = f64r::wexp((long double)(e % bins), (long double)bins);
                    o[1] = f64r::wexp((long double)((2 * e) % bins), (long double)bins);
                }
    }
}

hipError_t launch_f64reg(int bins, int fmt, const F64RegArgs &a, hipStream_t s)
{
    if (a.rows <= 0) return hipSuccess;
    int m_log2 = 0, dec = 0, logb = 0;
    if (!f64r::plan(bins, m_log2, dec, logb)) return hipErrorInvalidValue;
    f64r::Args b;
    b.iq = a.iq;
    b.window_k = a.window_k;
    b.tw0 = a.tw0;
    b.tw1 = a.tw1;
    b.tw2 = a.tw2;
    b.tw3 = a.tw3;
    b.rows_out = a.rows_out;
    b.first_row = a.first_row;
    b.rows = a.rows;
    b.row_stride = a.row_stride;
    b.hop = a.hop;
    b.gain = a.gain;
    b.stamps = a.stamps;
    b.spectra = a.spectra;
    if (fmt == RO_FMT_F32) return f64r::launch_fmt<RO_FMT_F32>(m_log2, dec, logb, b, s);
    if (fmt == RO_FMT_I16) return f64r::launch_fmt<RO_FMT_I16>(m_log2, dec, logb, b, s);
    if (fmt == RO_IQ_F64) return f64r::launch_fmt<RO_IQ_F64>(m_log2, dec, logb, b, s);
    return hipErrorInvalidValue;
}

}  // namespace ro
