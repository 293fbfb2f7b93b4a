// ro_fft_planar.h -- the twiddled radix-32 stage (fdit32 of ro_fft_device.h) on a PLANAR register layout (gfx950).
//
// Why: an LDS exchange that writes planes of floats (ds_write_addtid_b32: lane-linear, no address VGPR) can be read
// back two floats at a time (ds_read_b64: 256 B/clk/CU against the 128 of ds_read_b32) only if the two floats of one
// lane are the same component of two different points.  So the 32 points of a thread are held as 16 + 16 register
// pairs  R[i] = (re p, re p'),  I[i] = (im p, im p')  -- the two points of a pair are "mates" -- and every packed
// operation works on the same component of two points instead of on (re, im) of one.
//
// Positions, levels, exponents and results are those of fdit32: level L pairs positions p, p + (16 >> L), block U of
// level L multiplies its second operands by W32^E g_L, E = bitrev_L(U) (16 >> L), result k ends at position
// bitrev_32(k).  Mates differ in bit MB of the position (MB = 0: p, p + 1; MB = 1: p, p + 2):
//   pair index  pr<MB>(p),  half  hf<MB>(p)
// Three kinds of butterfly, by where the mates sit relative to the level's partner bit PB = 4 - L:
//   joint  (MB < PB)  both mates are in the same block: (A, B) <- (A + e B, A - e B) on two pairs, one twiddle for both
//                      halves; six packed FMAs for two butterflies -- the count of the (re, im) form;
//   mixed  (MB > PB)  the mates are in neighbouring blocks U, U + 1 whose exponents differ by 8 (a factor -i): the same
//                      six FMAs, the low half using e = w, the high half e = -i w (VOP3P op_sel / neg modifiers);
//   inpair (MB = PB)  the two mates ARE the partners: R = (a.re, b.re), I = (a.im, b.im); four FMAs per butterfly
//                      (one more than the other forms: a level of 16 butterflies costs 64 packed operations, not 48).
// Sums are formed exactly as cmadd / cmadd_mi form them (same operations in the same order); differences are 2a - s
// in the joint and mixed forms (as in fdit32) and computed directly in the inpair form.
#pragma once

#include "ro_fft_device.h"

namespace ro {
namespace planar {

template <int MB> __host__ __device__ constexpr int pr(int p) { return MB == 0 ? (p >> 1) : (((p >> 2) << 1) | (p & 1)); }
template <int MB> __host__ __device__ constexpr int hf(int p) { return (p >> MB) & 1; }

// the scheduling leash of ro_fft_device.h's tie(): VALU instructions do not cross it, memory and scalar ones may
__device__ __forceinline__ void leash() { __builtin_amdgcn_sched_barrier(0x4 | 0x10 | 0x80); }

// ---- joint: (A, B) <- (A + e B, A - e B), e = w (MI = false) or -i w = (w.y, -w.x) (MI = true), both halves alike
// RO_PLANAR_ABLATE (a -DRO_DIAG=1 build only, tools/r5/radix4_bound.sh): the joint butterflies of levels 0 and 1 of
// every head() leave out their last packed FMA -- 16 per stage, what an FMA radix-4 butterfly (22 instead of 24 real
// FMAs per four points) would save on those two pairs of levels.  The rows are wrong; only the time is of interest.
#if defined(RO_DIAG) && defined(RO_PLANAR_ABLATE)
#define RO_PLANAR_DROP(L, n) ((L) < 2)
#else
#define RO_PLANAR_DROP(L, n) false
#endif
template <bool MI, bool DROP = false> __device__ __forceinline__ void bf_joint(v2f &AR, v2f &AI, v2f &BR, v2f &BI, v2f w)
{
    // (the six operations in THIS order, a leash behind each: two dependent packed operations back to back cost an
    // s_nop on gfx950 -- hipcc's hazard recognizer pads it -- and left alone the scheduler pairs them up)
    v2f ur, ui, sr, si;
    if constexpr (!MI) {
        ur = __builtin_elementwise_fma(BR, w.xx, AR);  leash();
        ui = __builtin_elementwise_fma(BI, w.xx, AI);  leash();
        sr = __builtin_elementwise_fma(BI, -w.yy, ur); leash();
        si = __builtin_elementwise_fma(BR, w.yy, ui);  leash();
    } else {
        ur = __builtin_elementwise_fma(BR, w.yy, AR);  leash();
        ui = __builtin_elementwise_fma(BI, w.yy, AI);  leash();
        sr = __builtin_elementwise_fma(BI, w.xx, ur);  leash();
        si = __builtin_elementwise_fma(BR, -w.xx, ui); leash();
    }
    BR = __builtin_elementwise_fma(AR, (v2f){2.0f, 2.0f}, -sr); leash();
    if constexpr (!DROP) BI = __builtin_elementwise_fma(AI, (v2f){2.0f, 2.0f}, -si);
    else BI = AI;
    AR = sr;
    AI = si;
}

// ---- mixed: the low halves use e = w, the high halves e = -i w
//   s.re = fma(B.im, -e.y, fma(B.re, e.x, A.re))     low: e = (w.x, w.y)   high: e = (w.y, -w.x)
//   s.im = fma(B.re,  e.y, fma(B.im, e.x, A.im))
__device__ __forceinline__ void bf_mixed(v2f &AR, v2f &AI, v2f &BR, v2f &BI, v2f w)
{
    const v2f ur = __builtin_elementwise_fma(BR, w, AR);       // (B.re w.x, B.re' w.y) + A.re
    leash();
    const v2f ui = __builtin_elementwise_fma(BI, w, AI);
    leash();
    v2f sr, si;
    // low: -B.im w.y   high: +B.im' w.x
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,1,0] op_sel_hi:[1,0,1] neg_lo:[0,1,0]" : "=v"(sr) : "v"(BI), "v"(w), "v"(ur));
    // low: +B.re w.y   high: -B.re' w.x
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,1,0] op_sel_hi:[1,0,1] neg_hi:[0,1,0]" : "=v"(si) : "v"(BR), "v"(w), "v"(ui));
    BR = __builtin_elementwise_fma(AR, (v2f){2.0f, 2.0f}, -sr);
    leash();
    BI = __builtin_elementwise_fma(AI, (v2f){2.0f, 2.0f}, -si);
    AR = sr;
    AI = si;
}

// ---- inpair: R = (a.re, b.re), I = (a.im, b.im)  <-  (a + e b, a - e b), e = w or -i w
//   u   = (a.re + b.re e.x, a.re - b.re e.x)      R' = (u.lo - b.im e.y,  u.hi + b.im e.y)
//   u'  = (a.im + b.im e.x, a.im - b.im e.x)      I' = (u'.lo + b.re e.y, u'.hi - b.re e.y)
template <bool MI> __device__ __forceinline__ void bf_inpair(v2f &R, v2f &I, v2f w)
{
    v2f u, u2, r, i;
    if constexpr (!MI) {        // e = (w.x, w.y)
        asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0] neg_hi:[0,1,0]" : "=v"(u) : "v"(R), "v"(w), "v"(R));
        asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0] neg_hi:[0,1,0]" : "=v"(u2) : "v"(I), "v"(w), "v"(I));
        asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[1,1,1] neg_lo:[0,1,0]" : "=v"(r) : "v"(I), "v"(w), "v"(u));
        asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[1,1,1] neg_hi:[0,1,0]" : "=v"(i) : "v"(R), "v"(w), "v"(u2));
    } else {                    // e = (w.y, -w.x)
        asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[1,1,0] neg_hi:[0,1,0]" : "=v"(u) : "v"(R), "v"(w), "v"(R));
        asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[1,1,0] neg_hi:[0,1,0]" : "=v"(u2) : "v"(I), "v"(w), "v"(I));
        asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[1,0,1] neg_hi:[0,1,0]" : "=v"(r) : "v"(I), "v"(w), "v"(u));
        asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[1,0,1] neg_lo:[0,1,0]" : "=v"(i) : "v"(R), "v"(w), "v"(u2));
    }
    R = r;
    I = i;
}

// exponent of block U at level L
template <int L> __host__ __device__ constexpr int block_exp(int U) { return bitrev_bits<L>(U) * (16 >> L); }

// butterfly number n (0..15) of level L: block U = n / (S/2), offset n % (S/2); done once per pair of mates
template <int MB, int L, int n> __device__ __forceinline__ void bf_at(v2f (&R)[16], v2f (&I)[16], const v2f (&tw)[8])
{
    constexpr int S = 32 >> L, PB = 4 - L;
    constexpr int U = n / (S / 2), pa = U * S + n % (S / 2), pb = pa + S / 2;
    constexpr int E = block_exp<L>(U);
    if constexpr (PB == MB) {
        static_assert(pr<MB>(pa) == pr<MB>(pb) && hf<MB>(pa) == 0 && hf<MB>(pb) == 1, "partners are mates");
        bf_inpair<(E >= 8)>(R[pr<MB>(pa)], I[pr<MB>(pa)], tw[E & 7]);
        leash();
    } else if constexpr (hf<MB>(pa) == 0) {
        constexpr int pm = pa | (1 << MB);                     // the mate of pa; its partner is the mate of pb
        static_assert(pr<MB>(pm) == pr<MB>(pa) && pr<MB>(pm + S / 2) == pr<MB>(pb), "mates share their pairs");
        constexpr int Em = block_exp<L>(pm / S);
        if constexpr (MB < PB) {
            static_assert(Em == E, "mates in one block");
            bf_joint<(E >= 8), RO_PLANAR_DROP(L, n)>(R[pr<MB>(pa)], I[pr<MB>(pa)], R[pr<MB>(pb)], I[pr<MB>(pb)], tw[E & 7]);
        } else {
            static_assert(Em == E + 8 && E < 8, "mates in neighbouring blocks: a factor -i");
            bf_mixed(R[pr<MB>(pa)], I[pr<MB>(pa)], R[pr<MB>(pb)], I[pr<MB>(pb)], tw[E]);
        }
        leash();
    }
}

template <int MB, int L, int... Ns>
__device__ __forceinline__ void bf_all(v2f (&R)[16], v2f (&I)[16], const v2f (&tw)[8], std::integer_sequence<int, Ns...>)
{
    (bf_at<MB, L, Ns>(R, I, tw), ...);
}

// level L of the stage; g = w^(16 >> L).  The constants' products W32^e g as in fdit_level.
template <int MB, int L> __device__ __forceinline__ void level(v2f (&R)[16], v2f (&I)[16], v2f g)
{
    v2f tw[8];
    tw[0] = g;
    constexpr int STEP = 16 >> L;
    if constexpr (STEP <= 4) tw[4] = mul_w32<4>(g);
    if constexpr (STEP <= 2) { tw[2] = mul_w32<2>(g); tw[6] = mul_w32<6>(g); }
    if constexpr (STEP <= 1) { tw[1] = mul_w32<1>(g); tw[3] = mul_w32<3>(g); tw[5] = mul_w32<5>(g); tw[7] = mul_w32<7>(g); }
    bf_all<MB, L>(R, I, tw, std::make_integer_sequence<int, 16>{});
}

// levels 0..3
template <int MB> __device__ __forceinline__ void head(v2f (&R)[16], v2f (&I)[16], v2f g16, v2f g8, v2f g4, v2f g2)
{
    level<MB, 0>(R, I, g16);
    level<MB, 1>(R, I, g8);
    level<MB, 2>(R, I, g4);
    level<MB, 3>(R, I, g2);
}

// ---- last level, MB = 0 (inpair), in fdit32_last's order: blocks J and 8 + J, then done(J): positions 2J, 2J + 1,
// 16 + 2J, 17 + 2J = pairs J and 8 + J are final.  Blocks J, J + 1 (J even) and 8 + J, 9 + J share two products (their
// exponents differ by 8: the MI form), which are made where they are first used.
template <int J, typename F> __device__ __forceinline__ void last0_unit(v2f (&R)[16], v2f (&I)[16], v2f g1, v2f (&w)[2], F &done)
{
    constexpr int E0 = bitrev_bits<4>(J), E1 = bitrev_bits<4>(8 + J);
    static_assert(E1 == E0 + 1, "blocks J and 8 + J use neighbouring exponents");
    if constexpr (J % 2 == 0) {
        w[0] = mul_w32<E0 & 7>(g1);
        w[1] = mul_w32<E1 & 7>(g1);
    }
    bf_inpair<(E0 >= 8)>(R[J], I[J], w[0]);
    leash();
    bf_inpair<(E1 >= 8)>(R[8 + J], I[8 + J], w[1]);
    leash();
    done(std::integral_constant<int, J>{});
}
template <typename F, int... Js>
__device__ __forceinline__ void last0_all(v2f (&R)[16], v2f (&I)[16], v2f g1, F &done, std::integer_sequence<int, Js...>)
{
    v2f w[2];
    (last0_unit<Js>(R, I, g1, w, done), ...);
}
template <typename F> __device__ __forceinline__ void last0(v2f (&R)[16], v2f (&I)[16], v2f g1, F done)
{
    last0_all(R, I, g1, done, std::make_integer_sequence<int, 8>{});
}

// ---- last level, MB = 1 (mixed): unit u = blocks 2u, 2u + 1 = positions 4u .. 4u + 3 = pairs 2u (positions 4u, 4u + 2)
// and 2u + 1 (4u + 1, 4u + 3); then done(u).  One product per unit: W32^bitrev3(u) g1.
template <int U, typename F> __device__ __forceinline__ void last1_unit(v2f (&R)[16], v2f (&I)[16], v2f g1, F &done)
{
    constexpr int E = bitrev_bits<4>(2 * U);
    static_assert(E < 8 && bitrev_bits<4>(2 * U + 1) == E + 8, "blocks 2u, 2u + 1: exponents E, E + 8");
    const v2f w = mul_w32<E>(g1);
    bf_mixed(R[2 * U], I[2 * U], R[2 * U + 1], I[2 * U + 1], w);
    leash();
    done(std::integral_constant<int, U>{});
}
template <typename F, int... Us>
__device__ __forceinline__ void last1_all(v2f (&R)[16], v2f (&I)[16], v2f g1, F &done, std::integer_sequence<int, Us...>)
{
    (last1_unit<Us>(R, I, g1, done), ...);
}
template <typename F> __device__ __forceinline__ void last1(v2f (&R)[16], v2f (&I)[16], v2f g1, F done)
{
    last1_all(R, I, g1, done, std::make_integer_sequence<int, 8>{});
}

}  // namespace planar
}  // namespace ro
