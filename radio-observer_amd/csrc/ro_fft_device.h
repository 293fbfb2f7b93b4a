// ro_fft_device.h -- in-register radix-R butterflies and the Stockham stage
// plumbing shared by every STFT kernel instantiation (gfx950 only).
//
// Layout of one transform:  N = R0*R1*R2(*R3) points, T threads, P = N/T
// points per thread held in registers as float2 v[P].  Stage s (radix R,
// Ns = product of earlier radices) follows the autosort recurrence
//     in :  v[b*R + r] = x[j + r*(N/R)] * w(r*(j mod Ns)/(Ns*R)),   j = tid + T*b
//     out:  y[(j/Ns)*Ns*R + (j mod Ns) + r*Ns] = DFT_R(v)[r]
// so every LDS / global access of a wavefront is 64 consecutive elements.
// The exchange between stages goes through LDS; the element index i is padded
// to i + (i>>5) so that the stride-R writes of the first stage hit 32 different
// banks (ds_write_b32/b64 bank = dword address mod 32).
#pragma once

#include <hip/hip_runtime.h>
#include <utility>

// The butterflies' scheduling leash (tie(), below) comes in two forms, chosen per translation unit BEFORE this header
// is included: RO_TIE_SCHED = 0 an empty asm statement (ro_kernels.hip), 1 a scheduling barrier (ro_stft32k.hip).
// Everything in this header that depends on the choice lives in an inline namespace named after it, so the two
// translation units of the library define differently NAMED functions, not one inline function with two bodies.
#ifndef RO_TIE_SCHED
#define RO_TIE_SCHED 0
#endif

namespace ro {
#if RO_TIE_SCHED
inline namespace tie_sched {
#else
inline namespace tie_asm {
#endif

// cos(k*pi/16), k = 0..8
#define RO_C1 0.98078528040323044913f
#define RO_C2 0.92387953251128675613f
#define RO_C3 0.83146961230254523708f
#define RO_C4 0.70710678118654752440f
#define RO_C5 0.55557023301960222474f
#define RO_C6 0.38268343236508977173f
#define RO_C7 0.19509032201612826785f

template <int M> struct W32;   // exp(-2*pi*i*M/32) = (c, -s)
template <> struct W32<1>  { static constexpr float c = RO_C1, s = RO_C7; };
template <> struct W32<2>  { static constexpr float c = RO_C2, s = RO_C6; };
template <> struct W32<3>  { static constexpr float c = RO_C3, s = RO_C5; };
template <> struct W32<5>  { static constexpr float c = RO_C5, s = RO_C3; };
template <> struct W32<6>  { static constexpr float c = RO_C6, s = RO_C2; };
template <> struct W32<7>  { static constexpr float c = RO_C7, s = RO_C1; };
template <> struct W32<9>  { static constexpr float c = -RO_C7, s = RO_C1; };
template <> struct W32<10> { static constexpr float c = -RO_C6, s = RO_C2; };
template <> struct W32<11> { static constexpr float c = -RO_C5, s = RO_C3; };
template <> struct W32<13> { static constexpr float c = -RO_C3, s = RO_C5; };
template <> struct W32<14> { static constexpr float c = -RO_C2, s = RO_C6; };
template <> struct W32<15> { static constexpr float c = -RO_C1, s = RO_C7; };

// Complex values are clang 2-vectors: v2f arithmetic lowers to the packed VALU ops
// (v_pk_add_f32 / v_pk_mul_f32 / v_pk_fma_f32, two flops per lane per issue slot), the
// .yx / .xx swizzles fold into op_sel, and constant pairs live in SGPRs.
typedef float v2f __attribute__((ext_vector_type(2)));
typedef float v4f __attribute__((ext_vector_type(4)));

// a * w for a twiddle held in registers: (ax wx - ay wy, ay wx + ax wy) in two packed ops.
// hipcc builds the (-wy, wy) operand with an extra v_xor; the VOP3P modifiers do it for free:
// op_sel / op_sel_hi pick the half of each source that feeds the low / high result lane and
// neg_lo negates a source for the low lane only.  (Pure asm, no side effects: the scheduler is
// free to move it; packed VALU ops need no software wait states between each other.)
__device__ __forceinline__ v2f cmul(v2f a, v2f w)
{
    v2f t, r;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[1,0]" : "=v"(t) : "v"(a), "v"(w));                 // a * (wx, wx)
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_lo:[0,1,0]"           // (ay,ax)*(-wy,wy) + t
        : "=v"(r)
        : "v"(a), "v"(w), "v"(t));
    return r;
}

// d * exp(-2*pi*i*M/32) = (x c + y s, y c - x s)
template <int M> __device__ __forceinline__ v2f mul_w32(v2f d)
{
    if constexpr (M == 0) {
        return d;
    } else if constexpr (M == 8) {            // * (-i) = (y, -x)
        return d.yx * (v2f){1.0f, -1.0f};
    } else {
        constexpr float c = (M == 4) ? RO_C4 : (M == 12) ? -RO_C4 : W32<M == 4 || M == 12 ? 1 : M>::c;
        constexpr float s = (M == 4 || M == 12) ? RO_C4 : W32<M == 4 || M == 12 ? 1 : M>::s;
        const v2f t = d * (v2f){c, c};
        return __builtin_elementwise_fma(d.yx, (v2f){s, -s}, t);
    }
}

// Scheduling leash.  hipcc's scheduler interleaves all independent butterflies of a
// level (temporaries for each) and the 1024-thread kernel no longer fits its 128 VGPRs.
// tie() makes x look recomputed from dep (no instruction is emitted), which chains
// every SEQ_G-th butterfly behind the previous one in program order: at most SEQ_G
// butterflies' temporaries are live, and four waves per SIMD cover the lost ILP.
#ifndef RO_SEQ_G
#define RO_SEQ_G 2
#endif
constexpr int SEQ_G = RO_SEQ_G;

// RO_TIE_SCHED = 1: the leash is a scheduling barrier for VALU instructions (memory and scalar instructions may still
// cross it) instead of an empty asm statement.  hipcc (ROCm 7.2) assumes that ANY inline asm result may be a
// "dst_sel-forwarded" value on gfx950 and puts an s_nop 0 in front of the next VALU instruction that reads it -- and
// another one in front of the asm when its input comes from a packed op: ~300 s_nop per row and wave in the butterflies.
__device__ __forceinline__ void tie(v2f &x, const v2f &dep)
{
#if RO_TIE_SCHED
    __builtin_amdgcn_sched_barrier(0x4 | 0x10 | 0x80);
#else
    asm volatile("" : "+v"(x) : "v"(dep));
#endif
}

// One decimation-in-frequency level of a size-R sub-transform: butterfly I.
template <int R, int I> __device__ __forceinline__ void dif_bfly(v2f *v, const v2f *&tok)
{
    if constexpr (I % SEQ_G == 0) tie(v[I], *tok);
    const v2f a = v[I], b = v[I + R / 2];
    v[I] = a + b;
    v[I + R / 2] = mul_w32<I * (32 / R)>(a - b);
    tok = &v[I + R / 2];
}

template <int R, int... Is>
__device__ __forceinline__ void dif_level(v2f *v, const v2f *&tok, std::integer_sequence<int, Is...>)
{
    (dif_bfly<R, Is>(v, tok), ...);
}

template <int R> __device__ __forceinline__ void dif_rec(v2f *v, const v2f *&tok)
{
    if constexpr (R >= 2) {
        dif_level<R>(v, tok, std::make_integer_sequence<int, R / 2>{});
        dif_rec<R / 2>(v, tok);
        dif_rec<R / 2>(v + R / 2, tok);
    }
}

// In-place DFT of R points (R in {2,4,8,16,32}); result k sits at v[bitrev_R(k)].
template <int R> __device__ __forceinline__ void dif(v2f *v)
{
    const v2f *tok = &v[R - 1];
    dif_rec<R>(v, tok);
}

// ---------------------------------------------------------------------------
// Decimation-in-time form of the same in-register transform (what the single-pass kernels use; the DIF form above
// remains for the fold kernel of the large transforms): same positions and pairs per
// level as dif<R>, result k again at v[bitrev_R(k)], but the constant twiddle sits BEFORE the butterfly, on the
// second operand, and is the same for a whole block: block `u` at depth l uses W32^E with E = bitrev_l(u) * (16 >> l),
// i.e. the two halves of a block with exponent E continue with E/2 and E/2 + 8.  That form fuses:
//   a' = a + w b    two packed FMAs  (b.xx * (c,-s) + a, then b.yy * (s,c) + that)
//   b' = a - w b  = 2 a - a'         one packed FMA
// three issue slots per butterfly instead of four (add, sub, two for the product), two instead of three for w = -i.
// ---------------------------------------------------------------------------
// acc + x * w for a twiddle held in registers: two packed FMAs (same modifier trick as cmul)
__device__ __forceinline__ v2f cmadd(v2f x, v2f w, v2f acc)
{
    v2f t, r;
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[1,0,1]" : "=v"(t) : "v"(x), "v"(w), "v"(acc));     // x * (wx, wx) + acc
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_lo:[0,1,0]"               // (xy,xx)*(-wy,wy) + t
        : "=v"(r)
        : "v"(x), "v"(w), "v"(t));
    return r;
}

// acc + x * (-i w): the same two FMAs with the halves of w swapped and one sign moved
__device__ __forceinline__ v2f cmadd_mi(v2f x, v2f w, v2f acc)
{
    v2f t, r;
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,1,0] op_sel_hi:[1,1,1]" : "=v"(t) : "v"(x), "v"(w), "v"(acc));   // x * (wy, wy) + acc
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[0,0,1] neg_hi:[0,1,0]"                          // (xy,xx)*(wx,-wx) + t
        : "=v"(r)
        : "v"(x), "v"(w), "v"(t));
    return r;
}

// (a, b) <- (a + W32^E b, a - W32^E b)
template <int E> __device__ __forceinline__ void dit_pair(v2f &a, v2f &b)
{
    if constexpr (E == 0) {
        const v2f s = a + b;
        b = a - b;
        a = s;
    } else if constexpr (E == 8) {            // w b = (b.y, -b.x)
        const v2f s = __builtin_elementwise_fma(b.yx, (v2f){1.0f, -1.0f}, a);
        b = __builtin_elementwise_fma(b.yx, (v2f){-1.0f, 1.0f}, a);
        a = s;
    } else {                                  // w b = (b.x c + b.y s, b.y c - b.x s)
        constexpr float c = (E == 4) ? RO_C4 : (E == 12) ? -RO_C4 : W32<E == 4 || E == 12 ? 1 : E>::c;
        constexpr float sn = (E == 4 || E == 12) ? RO_C4 : W32<E == 4 || E == 12 ? 1 : E>::s;
        const v2f t = __builtin_elementwise_fma(b.xx, (v2f){c, -sn}, a);
        const v2f s = __builtin_elementwise_fma(b.yy, (v2f){sn, c}, t);
        b = __builtin_elementwise_fma(a, (v2f){2.0f, 2.0f}, -s);
        a = s;
    }
}

template <int R, int E, int I> __device__ __forceinline__ void dit_bfly(v2f *v, const v2f *&tok)
{
    if constexpr (I % SEQ_G == 0) tie(v[I], *tok);
    dit_pair<E>(v[I], v[I + R / 2]);
    tok = &v[I + R / 2];
}

template <int R, int E, int... Is>
__device__ __forceinline__ void dit_level(v2f *v, const v2f *&tok, std::integer_sequence<int, Is...>)
{
    (dit_bfly<R, E, Is>(v, tok), ...);
}

template <int R, int E> __device__ __forceinline__ void dit_rec(v2f *v, const v2f *&tok)
{
    if constexpr (R >= 2) {
        dit_level<R, E>(v, tok, std::make_integer_sequence<int, R / 2>{});
        dit_rec<R / 2, E / 2>(v, tok);
        dit_rec<R / 2, E / 2 + 8>(v + R / 2, tok);
    }
}

// In-place DFT of R points, decimation in time; result k sits at v[bitrev_R(k)].
template <int R> __device__ __forceinline__ void dit(v2f *v)
{
    const v2f *tok = &v[R - 1];
    dit_rec<R, 0>(v, tok);
}

// the transform minus its first level (the caller has done the R/2 butterflies v[i], v[i + R/2] itself)
template <int R> __device__ __forceinline__ void dit_after_first_level(v2f *v)
{
    const v2f *tok = &v[R - 1];
    dit_rec<R / 2, 0>(v, tok);
    dit_rec<R / 2, 8>(v + R / 2, tok);
}

// ---------------------------------------------------------------------------
// Radix-32 stage WITH its stage twiddles (element r of the butterfly enters multiplied by w^r, w per thread), the
// twiddles factored through the levels:  x_r w^r + x_{r+16} w^{r+16} = w^r (x_r + w^16 x_{r+16}), so level l only
// needs g_l = w^(16 >> l) on its second operands (times the block's constant W32^E) and the common factor w^r
// shrinks to w^0 = 1 at the last level.  Every butterfly is cmadd + (2a - a') = 3 issue slots; the constants cost
// 11 products per stage (E and E+8 share one: -i is a modifier), against 24 composed twiddles + 31 products when
// the twiddles are applied up front.  Only w, w^2, w^4, w^8, w^16 are loaded.  Result k at v[bitrev_32(k)].
// ---------------------------------------------------------------------------
template <int BITS> __host__ __device__ constexpr int bitrev_bits(int k)
{
    int r = 0;
    for (int b = 0; b < BITS; ++b) { r = (r << 1) | (k & 1); k >>= 1; }
    return r;
}

template <int L, int U, int I> __device__ __forceinline__ void fdit_bfly(v2f *x, const v2f (&tw)[8], const v2f *&tok)
{
    constexpr int S = 32 >> L, E = bitrev_bits<L>(U) * (16 >> L), base = U * S;
    v2f &a = x[base + I], &b = x[base + I + S / 2];
    if constexpr (I % SEQ_G == 0) tie(a, *tok);
    const v2f s = (E >= 8) ? cmadd_mi(b, tw[E & 7], a) : cmadd(b, tw[E & 7], a);
    b = __builtin_elementwise_fma(a, (v2f){2.0f, 2.0f}, -s);
    a = s;
    tok = &b;
}

template <int L, int U, int... Is>
__device__ __forceinline__ void fdit_block(v2f *x, const v2f (&tw)[8], const v2f *&tok, std::integer_sequence<int, Is...>)
{
    (fdit_bfly<L, U, Is>(x, tw, tok), ...);
}

template <int L, int... Us>
__device__ __forceinline__ void fdit_blocks(v2f *x, const v2f (&tw)[8], const v2f *&tok, std::integer_sequence<int, Us...>)
{
    (fdit_block<L, Us>(x, tw, tok, std::make_integer_sequence<int, (16 >> L)>{}), ...);
}

// level L of the stage; g = w^(16 >> L)
template <int L> __device__ __forceinline__ void fdit_level(v2f *x, v2f g, const v2f *&tok)
{
    v2f tw[8];
    tw[0] = g;
    constexpr int STEP = 16 >> L;                       // exponents in use at this level: multiples of STEP below 8
    if constexpr (STEP <= 4) tw[4] = mul_w32<4>(g);
    if constexpr (STEP <= 2) { tw[2] = mul_w32<2>(g); tw[6] = mul_w32<6>(g); }
    if constexpr (STEP <= 1) { tw[1] = mul_w32<1>(g); tw[3] = mul_w32<3>(g); tw[5] = mul_w32<5>(g); tw[7] = mul_w32<7>(g); }
    fdit_blocks<L>(x, tw, tok, std::make_integer_sequence<int, (1 << L)>{});
}

// x[0..32) <- DFT32 of (x[r] w^r); g16..g1 = w^16, w^8, w^4, w^2, w
__device__ __forceinline__ void fdit32(v2f *x, v2f g16, v2f g8, v2f g4, v2f g2, v2f g1)
{
    const v2f *tok = &x[31];
    fdit_level<0>(x, g16, tok);
    fdit_level<1>(x, g8, tok);
    fdit_level<2>(x, g4, tok);
    fdit_level<3>(x, g2, tok);
    fdit_level<4>(x, g1, tok);
}

// fdit32 with a hook into its last level: that level's 16 butterflies (x[2U], x[2U+1]) run in the order
// U = 0, 8, 1, 9, ... and done(j) is called after the pair (j, 8 + j): x[2j], x[2j+1], x[16+2j], x[17+2j] are final
// then (bins bitrev32 of those positions) and their registers free for whatever the caller loads into them next.
__device__ __forceinline__ void tie(v2f &x, float dep)
{
#if RO_TIE_SCHED
    __builtin_amdgcn_sched_barrier(0x4 | 0x10 | 0x80);
#else
    asm volatile("" : "+v"(x) : "v"(dep));
#endif
}

// one last-level butterfly (a, b) <- (a + W32^E g1 b, a - W32^E g1 b), the product w = W32^(E & 7) g1 given
template <int E> __device__ __forceinline__ void fdit_last_bfly(v2f &a, v2f &b, v2f w)
{
    const v2f s = (E >= 8) ? cmadd_mi(b, w, a) : cmadd(b, w, a);
    b = __builtin_elementwise_fma(a, (v2f){2.0f, 2.0f}, -s);
    a = s;
}

// Blocks J and 8 + J of the last level have exponents E = bitrev4(J) and E + 1, and J, J + 1 (J even) use the same two
// products W32^(E & 7) g1 (E and E + 8 differ by -i, a modifier): the products are made two at a time, right where
// they are first used -- 4 registers live instead of the 16 that all eight of them held through the level.
// done(J) returns a float the next pair is chained behind (the scheduling leash of tie(); it cannot be one of the
// x[] just finished: the caller has already started to reload those registers).
template <int J, typename F>
__device__ __forceinline__ void fdit_last_pair(v2f *x, v2f g1, v2f (&w)[2], float &chain, F &done)
{
    constexpr int E0 = bitrev_bits<4>(J), E1 = bitrev_bits<4>(8 + J);
    static_assert(E1 == E0 + 1, "blocks J and 8 + J use neighbouring exponents");
    if constexpr (J % 2 == 0) {
        w[0] = mul_w32<E0 & 7>(g1);
        w[1] = mul_w32<E1 & 7>(g1);
    }
    tie(x[2 * J], chain);
    fdit_last_bfly<E0>(x[2 * J], x[2 * J + 1], w[0]);
    tie(x[16 + 2 * J], x[2 * J + 1]);
    fdit_last_bfly<E1>(x[16 + 2 * J], x[17 + 2 * J], w[1]);
    chain = done(std::integral_constant<int, J>{});
}

template <typename F, int... Js>
__device__ __forceinline__ void fdit_last_level(v2f *x, v2f g1, F &done, std::integer_sequence<int, Js...>)
{
    v2f w[2];
    float chain = x[31].x;
    (fdit_last_pair<Js>(x, g1, w, chain, done), ...);
}

// fdit32 in two parts: levels 0..3, then the last level with the hook (the caller may branch between them)
template <typename F> __device__ __forceinline__ void fdit32_head(v2f *x, v2f g16, v2f g8, v2f g4, v2f g2, F hook)
{
    const v2f *tok = &x[31];
    fdit_level<0>(x, g16, tok);
    hook(std::integral_constant<int, 0>{});
    fdit_level<1>(x, g8, tok);
    hook(std::integral_constant<int, 1>{});
    fdit_level<2>(x, g4, tok);
    hook(std::integral_constant<int, 2>{});
    fdit_level<3>(x, g2, tok);
    hook(std::integral_constant<int, 3>{});
}
__device__ __forceinline__ void fdit32_head(v2f *x, v2f g16, v2f g8, v2f g4, v2f g2)
{
    fdit32_head(x, g16, g8, g4, g2, [](auto) {});
}

template <typename F> __device__ __forceinline__ void fdit32_last(v2f *x, v2f g1, F done)
{
    fdit_last_level(x, g1, done, std::make_integer_sequence<int, 8>{});
}

// dit<32> in LEVEL order (dit_rec walks the same butterflies depth first), in two parts like fdit32: levels 0..3 with a
// hook after each of them (four evenly spaced places for the caller to slip other work -- row stores -- between the
// butterflies), then the last level with the pair hook of fdit_last_pair.  Block U of level L has exponent
// bitrev_L(U) * (16 >> L), as in the recursion.
template <int L, int... Us>
__device__ __forceinline__ void dit_level_order(v2f *v, const v2f *&tok, std::integer_sequence<int, Us...>)
{
    constexpr int S = 32 >> L;
    (dit_level<S, bitrev_bits<L>(Us) * (16 >> L)>(v + Us * S, tok, std::make_integer_sequence<int, S / 2>{}), ...);
}

template <typename F> __device__ __forceinline__ void dit32_head(v2f *v, F hook)
{
    const v2f *tok = &v[31];
    dit_level_order<0>(v, tok, std::make_integer_sequence<int, 1>{});
    hook(std::integral_constant<int, 0>{});
    dit_level_order<1>(v, tok, std::make_integer_sequence<int, 2>{});
    hook(std::integral_constant<int, 1>{});
    dit_level_order<2>(v, tok, std::make_integer_sequence<int, 4>{});
    hook(std::integral_constant<int, 2>{});
    dit_level_order<3>(v, tok, std::make_integer_sequence<int, 8>{});
    hook(std::integral_constant<int, 3>{});
}

template <int J, typename F> __device__ __forceinline__ void dit_last_pair(v2f *x, float &chain, F &done)
{
    tie(x[2 * J], chain);
    dit_pair<bitrev_bits<4>(J)>(x[2 * J], x[2 * J + 1]);
    tie(x[16 + 2 * J], x[2 * J + 1]);
    dit_pair<bitrev_bits<4>(8 + J)>(x[16 + 2 * J], x[17 + 2 * J]);
    chain = done(std::integral_constant<int, J>{});
}

template <typename F, int... Js>
__device__ __forceinline__ void dit_last_level(v2f *x, F &done, std::integer_sequence<int, Js...>)
{
    float chain = x[31].x;
    (dit_last_pair<Js>(x, chain, done), ...);
}

// last level of dit<32>: pairs (j, 8 + j) in the order j = 0..7, done(j) after each (see fdit_last_pair)
template <typename F> __device__ __forceinline__ void dit32_last(v2f *x, F done)
{
    dit_last_level(x, done, std::make_integer_sequence<int, 8>{});
}

template <int R> __host__ __device__ constexpr int bitrev(int k)
{
    int r = 0;
    for (int b = 1; b < R; b <<= 1) { r = (r << 1) | (k & 1); k >>= 1; }
    return r;
}

__device__ __forceinline__ constexpr int lds_pad(int i) { return i + (i >> 5); }

}  // inline namespace tie_sched / tie_asm
}  // namespace ro
