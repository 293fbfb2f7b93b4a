// ro_device_util.h -- device helpers shared by the gfx950 kernels (ro_kernels.hip, ro_stft32k.hip): raw buffer
// loads / stores, the add-TID LDS writes, sample formats, and BolidRecorder's per-row band scan.
#pragma once

#include <hip/hip_runtime.h>

#include "ro_kernels.h"
#include "ro_fft_device.h"

// cache policy of the row stores (gfx950 aux bits: 1 = sc0, 2 = nt, 16 = sc1); nt measured 2.5 % faster (rows are write-once)
#ifndef RO_STORE_AUX
#define RO_STORE_AUX 2
#endif

// 0 only in tests/test_isa_cpu.py's negative build: the wait states behind the 16-byte stores left out, to show that
// the test sees the hazard they cover
#ifndef RO_STORE_NOP
#define RO_STORE_NOP 1
#endif

namespace ro {

// Buffer-descriptor helpers.  All global traffic of the STFT kernel goes through
// raw buffer instructions: one 32-bit per-lane offset VGPR per stream, the
// per-register part of the address in an SGPR (soffset), and free hardware
// bounds checking (out-of-range loads give 0, stores are dropped).
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void *base, unsigned bytes)
{
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(base), 0, (int)bytes, 0x00020000);
}
__device__ __forceinline__ v2f buf_load_f2(__amdgpu_buffer_rsrc_t r, int voff, int soff)
{
    u32x2 t = __builtin_amdgcn_raw_buffer_load_b64(r, voff, soff, 0);
    return (v2f){__uint_as_float(t.x), __uint_as_float(t.y)};
}
__device__ __forceinline__ float buf_load_f(__amdgpu_buffer_rsrc_t r, int voff, int soff)
{
    return __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r, voff, soff, 0));
}
__device__ __forceinline__ void buf_store_f(float x, __amdgpu_buffer_rsrc_t r, int voff, int soff)
{
    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(x), r, voff, soff, 0);
}
__device__ __forceinline__ void buf_store_f4(float x0, float x1, float x2, float x3,
                                             __amdgpu_buffer_rsrc_t r, int voff, int soff)
{
    u32x4 t = {__float_as_uint(x0), __float_as_uint(x1), __float_as_uint(x2), __float_as_uint(x3)};
    __builtin_amdgcn_raw_buffer_store_b128(t, r, voff, soff, RO_STORE_AUX);
    // A 16-byte store goes on reading its data registers after it has issued: a VALU write to them in the next two
    // wait states changes what the last lanes store.  The rule is the ISA guide's table of software-inserted wait states
    // ("VMEM store of more than 64 bits followed by a write of the VGPRs holding its data": 1 wait state, 2 on gfx940 and
    // later), which EXEMPTS buffer stores whose offset comes from an SGPR; hipcc implements both the rule and the
    // exemption (GCNHazardRecognizer::createsVALUHazard / checkVALUHazardsHelper, VALUWaitStates = 2 with gfx940
    // instructions) and counts every inline-asm statement -- the empty ones of the scheduling leash included -- as a
    // wait state.  What was observed here departs from the exemption: with the SGPR-soffset form used for every row
    // store and the data registers recycled at once by the butterflies around the pipelined stores, one row in ten left
    // with lanes 12..15 of every 16 carrying the NEXT values of those registers; two real wait states behind the store
    // ended it.  The asm keeps the store's register tuple alive across them -- the tuple as ONE operand: given the four
    // dwords as four operands, hipcc has kept the registers they came from alive instead, copied them into a fresh
    // tuple for the store and moved the statement in front of it (seen in ro_fourstep.hip's first 16-byte stores);
    // tests/test_isa_cpu.py checks the emitted ISA of every kernel for the pattern (and that it would see it: a build
    // without this s_nop is red).
#if RO_STORE_NOP
    asm volatile("s_nop 1" ::"v"(t));
#endif
}
// the same store (and the same two wait states behind it) for four raw dwords and a cache policy of the caller's choice
template <int AUX> __device__ __forceinline__ void buf_store_u4(u32x4 t, __amdgpu_buffer_rsrc_t r, int voff, int soff)
{
    __builtin_amdgcn_raw_buffer_store_b128(t, r, voff, soff, AUX);
#if RO_STORE_NOP
    asm volatile("s_nop 1" ::"v"(t));
#endif
}

// value of lane (quad_perm) of the same register, DPP: no LDS, full-rate VALU
template <int CTRL> __device__ __forceinline__ float dpp_quad(float x)
{
    return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(x), CTRL, 0xF, 0xF, false));
}

// Workgroup barrier that the optimiser may not move memory operations across.  __syncthreads() alone is not enough:
// with a branch behind an exchange, hipcc sank the plain LDS gather loads of the exchange below its closing barrier
// (two s_barrier back to back in the ISA, the ds_reads after them) and rows went wrong at random once several
// workgroups shared a CU (tests/test_gpu_fullsize.py caught it).  The empty asm statements claim to touch memory,
// so no load or store crosses them in either direction.
__device__ __forceinline__ void wg_sync()
{
    asm volatile("" ::: "memory");
    __syncthreads();
    asm volatile("" ::: "memory");
}

// Ordering by fake data dependence: returns `off` unchanged, but the compiler must
// assume it was recomputed from `x`, so loads addressed with the result cannot be
// issued before `x` exists.  (A "memory" clobber does not stop the scheduler from
// clustering buffer loads; this does, and costs no instruction.)
__device__ __forceinline__ int after(int off, float x)
{
    asm volatile("" : "+v"(off) : "v"(x));
    return off;
}

// ---------------------------------------------------------------------------
// stage helpers shared by the STFT kernels (all indices compile-time after unrolling -> v[] stays in VGPRs)
// ---------------------------------------------------------------------------
template <int P, int R> __device__ __forceinline__ void butterflies(v2f (&v)[P])
{
#pragma unroll
    for (int b = 0; b < P / R; ++b) {
        dit<R>(&v[b * R]);
    }
}

__device__ __forceinline__ v2f tw_load(__amdgpu_buffer_rsrc_t tw, int koff, int entry)
{
    return buf_load_f2(tw, koff, entry * 8);
}

// Stage twiddles.  A radix-R stage needs w^r, r = 1..R-1, per butterfly, w = exp(-2 pi i k / (NS R)) depending on
// the thread.  Radix <= 8 loads them all (8-byte loads from the generic table).  Radix 16 / 32 load a few powers
// from the packed table (three 16-byte loads) and get the rest by multiplication -- radix 32 needs only
// w, w^2, w^4, w^8, w^16 (fdit32), radix 16 holds w, w^2, w^3, w^4, w^8, w^12 and composes w^(4m+c).  At most two
// extra roundings (~1e-7) on top of the table's correctly rounded entries.  The loads are split from their use so
// that they are issued BEFORE the LDS exchange of the stage and land while the workgroup sits in its barriers.
constexpr int TW_SET = 7;     // twiddles held per butterfly: R=32: 5, R=16: 6, R<=8: R-1

// C8 (radix 8 only): load w, w^2, w^4 and let tw_apply make the other four by multiplication -- 6 registers held per
// butterfly across the exchange instead of 14 (the N = 8192 add-TID plan has four radix-8 butterflies per thread)
template <int P, int T, int R, int NS, int OFF, int PK, bool C8 = false>
__device__ __forceinline__ void tw_prefetch(v2f (&t)[P / R][TW_SET], __amdgpu_buffer_rsrc_t tw,
                                            __amdgpu_buffer_rsrc_t twk, int tid)
{
    static_assert(R == 32 || R == 16 || R <= 8, "unsupported radix");
#pragma unroll
    for (int b = 0; b < P / R; ++b) {
        const int koff = ((tid + T * b) & (NS - 1)) * 8;
        if constexpr (R >= 16) {
#pragma unroll
            for (int q = 0; q < 3; ++q) {
                const u32x4 u = __builtin_amdgcn_raw_buffer_load_b128(twk, koff * 2, (PK + q * NS) * 16, 0);
                const v2f lo = (v2f){__uint_as_float(u.x), __uint_as_float(u.y)};
                const v2f hi = (v2f){__uint_as_float(u.z), __uint_as_float(u.w)};
                // radix 32: t = {w, w^2, w^4, w^8, w^16}; radix 16: t = {w, w^2, w^3, w^4, w^8, w^12}
                if (q == 0) { t[b][0] = lo; t[b][1] = hi; }
                else if (q == 1) { t[b][2] = lo; t[b][3] = hi; }
                else { t[b][4] = lo; if constexpr (R == 16) t[b][5] = hi; }
            }
        } else if constexpr (C8 && R == 8) {
            t[b][0] = tw_load(tw, koff, OFF);
            t[b][1] = tw_load(tw, koff, OFF + NS);
            t[b][3] = tw_load(tw, koff, OFF + 3 * NS);
        } else {
#pragma unroll
            for (int r = 1; r < R; ++r) t[b][r - 1] = tw_load(tw, koff, OFF + (r - 1) * NS);
        }
    }
}

// stage twiddles applied up front: x[r] *= w^r, all R-1 powers held (radix <= 8; larger radices go through
// tw_butterflies' fused forms)
template <int P, int R, bool C8 = false>
__device__ __forceinline__ void tw_apply(v2f (&v)[P], const v2f (&t)[P / R][TW_SET])
{
#pragma unroll
    for (int b = 0; b < P / R; ++b) {
        v2f *x = &v[b * R];
        if constexpr (C8 && R == 8) {
            const v2f w1 = t[b][0], w2 = t[b][1], w4 = t[b][3], w3 = cmul(w1, w2);
            x[1] = cmul(x[1], w1);
            x[2] = cmul(x[2], w2);
            x[3] = cmul(x[3], w3);
            x[4] = cmul(x[4], w4);
            x[5] = cmul(x[5], cmul(w4, w1));
            x[6] = cmul(x[6], cmul(w4, w2));
            x[7] = cmul(x[7], cmul(w4, w3));
            continue;
        }
#pragma unroll
        for (int r = 1; r < R; ++r) x[r] = cmul(x[r], t[b][R <= 8 ? r - 1 : r % 5]);
    }
}

// Twiddles and butterflies of one stage.  Radix 32: the twiddles are factored through the levels (fdit32 in
// ro_fft_device.h).  Radix 16 fuses them into the first level:
//   A = x[r] w^r (two ops),  a' = A + x[r+R/2] w^(r+R/2) (two FMAs),  b' = 2A - a' (one)
// five issue slots per pair where twiddling both and then adding / subtracting takes six.
template <int P, int R, bool C8 = false>
__device__ __forceinline__ void tw_butterflies(v2f (&v)[P], const v2f (&t)[P / R][TW_SET])
{
    if constexpr (R >= 16) {
#pragma unroll
        for (int b = 0; b < P / R; ++b) {
            v2f *x = &v[b * R];
            constexpr int H = R / 2;
            // radix 16: w^r for r = 4m + c (c, m in 0..3) from the held set t = {w, w^2, w^3, w^4, w^8, w^12}
#pragma unroll
            for (int m = 0; m < 4; ++m) {
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const int r = 4 * m + c;
                    if constexpr (R == 32) {
                        continue;                                   // handled below (fdit32)
                    } else {
                        // R == 16: pairs (r, 8 + r) for r < 8; r >= 8 is the partner's twiddle w^(8+r')
                        if (r >= H) continue;
                        const v2f wlo = (r == 0) ? (v2f){1.0f, 0.0f}
                                        : (m == 0) ? t[b][c - 1] : (c == 0) ? t[b][2 + m] : cmul(t[b][2 + m], t[b][c - 1]);
                        const int rh = r + H, mh = rh / 4, ch = rh % 4;
                        const v2f whi = (ch == 0) ? t[b][2 + mh] : cmul(t[b][2 + mh], t[b][ch - 1]);
                        const v2f A = (r == 0) ? x[0] : cmul(x[r], wlo);
                        const v2f s = cmadd(x[rh], whi, A);
                        x[rh] = __builtin_elementwise_fma(A, (v2f){2.0f, 2.0f}, -s);
                        x[r] = s;
                    }
                }
            }
            if constexpr (R == 32) fdit32(x, t[b][4], t[b][3], t[b][2], t[b][1], t[b][0]);
            else dit_after_first_level<R>(x);
        }
    } else {
        tw_apply<P, R, C8>(v, t);
        butterflies<P, R>(v);
    }
}
// ---------------------------------------------------------------------------
// ds_write_addtid_b32: LDS address = M0[15:0] + 16-bit offset + 4 * lane, no address VGPR, 2 cycles per
// wave-instruction (ds_write_b32 moves address + data at 4).  The image it writes is lane-linear.
// ---------------------------------------------------------------------------
template <int O0, int O1, int O2, int O3, int O4, int O5, int O6, int O7>
__device__ __forceinline__ void addtid_write8(unsigned m0, float a0, float a1, float a2, float a3, float a4,
                                              float a5, float a6, float a7)
{
    // "SALU writes M0 -> LDS add-TID instruction" needs one wait state; hipcc pads nothing inside asm
    asm volatile("s_mov_b32 m0, %8\n\t"
                 "s_nop 0\n\t"
                 "ds_write_addtid_b32 %0 offset:%9\n\t"
                 "ds_write_addtid_b32 %1 offset:%10\n\t"
                 "ds_write_addtid_b32 %2 offset:%11\n\t"
                 "ds_write_addtid_b32 %3 offset:%12\n\t"
                 "ds_write_addtid_b32 %4 offset:%13\n\t"
                 "ds_write_addtid_b32 %5 offset:%14\n\t"
                 "ds_write_addtid_b32 %6 offset:%15\n\t"
                 "ds_write_addtid_b32 %7 offset:%16"
                 :
                 : "v"(a0), "v"(a1), "v"(a2), "v"(a3), "v"(a4), "v"(a5), "v"(a6), "v"(a7), "s"(m0), "n"(O0), "n"(O1),
                   "n"(O2), "n"(O3), "n"(O4), "n"(O5), "n"(O6), "n"(O7)
                 : "memory", "m0");      // M0 is hipcc's too (LDS-DMA): it must know the asm leaves another value there
}

// four add-TID writes with two M0 values: a0, a1 at offsets OA0, OA1 from M0 = ma, then b0, b1 at OB0, OB1 from mb
template <int OA0, int OA1, int OB0, int OB1>
__device__ __forceinline__ void addtid_write4(unsigned ma, unsigned mb, float a0, float a1, float b0, float b1)
{
    asm volatile("s_mov_b32 m0, %4\n\t"
                 "s_nop 0\n\t"
                 "ds_write_addtid_b32 %0 offset:%6\n\t"
                 "ds_write_addtid_b32 %1 offset:%7\n\t"
                 "s_mov_b32 m0, %5\n\t"
                 "s_nop 0\n\t"
                 "ds_write_addtid_b32 %2 offset:%8\n\t"
                 "ds_write_addtid_b32 %3 offset:%9"
                 :
                 : "v"(a0), "v"(a1), "v"(b0), "v"(b1), "s"(ma), "s"(mb), "n"(OA0), "n"(OA1), "n"(OB0), "n"(OB1)
                 : "memory", "m0");
}

// ---------------------------------------------------------------------------
// per-row band scan: BolidRecorder::noise / peak / average (src/BolidRecorder.cpp:313-347), one wavefront per row.
// Shared by scan_kernel (rows in HBM) and the fused epilogue of the N = 32768 plan (row still in LDS).
// ---------------------------------------------------------------------------
__device__ __forceinline__ unsigned order_key(float x)
{
    unsigned u = __float_as_uint(x);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float key_to_float(unsigned k)
{
    unsigned u = (k & 0x80000000u) ? (k & 0x7fffffffu) : ~k;
    return __uint_as_float(u);
}

// inclusive prefix sum over the 64 lanes with DPP moves (VALU only; a __shfl_up ladder is six dependent LDS round
// trips): shifts by 1, 2, 4, 8 inside each row of 16 lanes, then the row totals are passed on with row_bcast
__device__ __forceinline__ unsigned wave_inclusive_sum(unsigned x)
{
    int v = (int)x;
    v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, true);     // row_shr:1, out-of-row lanes read 0
    v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, true);     // row_shr:2
    v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, true);     // row_shr:4
    v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, true);     // row_shr:8
    v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, true);     // row_bcast:15 -> rows 1 and 3
    v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xc, 0xf, true);     // row_bcast:31 -> rows 2 and 3
    return (unsigned)v;
}


// row accessors: column c of the fft-shifted row
struct GlobalRow {
    const float *p;
    __device__ __forceinline__ float operator()(int c) const { return p[c]; }
};
// the natural-order LDS image of the N = 32768 epilogue: bin k at element k, shifted column c is bin (c + N/2) mod N
template <int N> struct ImageRow {
    const float *img;
    __device__ __forceinline__ float operator()(int c) const { return img[(c + N / 2) & (N - 1)]; }
};

constexpr int SCAN_BATCH = 16;        // loads in flight per lane where a band is re-read instead of kept
constexpr int SCAN_E = 16;            // noise-band elements cached per lane by the CACHED form (band <= 1024)

// maximum / minimum over the 64 lanes, VALU only (the same DPP ladder as wave_inclusive_sum with max in the place of
// +: an inclusive prefix maximum whose last lane holds the total; 0 is the identity the out-of-row lanes read).  The
// __shfl_xor form these replace is six DEPENDENT ds_bpermute round trips (~120 cycles each) per reduction -- the
// fused scan's two waves spent most of their time in them.
__device__ __forceinline__ unsigned wave_max_u32(unsigned x)
{
    int v = (int)x;
    auto mx = [](int a, int b) { return (int)max((unsigned)a, (unsigned)b); };
    v = mx(v, __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, true));     // row_shr:1
    v = mx(v, __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, true));     // row_shr:2
    v = mx(v, __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, true));     // row_shr:4
    v = mx(v, __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, true));     // row_shr:8
    v = mx(v, __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, true));     // row_bcast:15 -> rows 1 and 3
    v = mx(v, __builtin_amdgcn_update_dpp(0, v, 0x143, 0xc, 0xf, true));     // row_bcast:31 -> rows 2 and 3
    return (unsigned)__builtin_amdgcn_readlane(v, 63);
}
__device__ __forceinline__ unsigned wave_min_u32(unsigned x) { return ~wave_max_u32(~x); }

// noise(): element floor(W/4) of the ascending noise band, times two (src/BolidRecorder.cpp:313-317).
// Order statistic by radix select on the order-preserving integer image of the floats, exact (the result is an
// element of the band, bit for bit).  The bits on which ALL keys agree (wave min ^ max) are skipped -- a noise band
// shares its sign and most of its exponent, and histogramming those bits first put every key on the same LDS word
// (64-way ds_add_u32 collisions: 55 % of the old kernel's LDS cycles).  Below them, 8 bits per pass: every pass
// histograms the digit of the keys that still match the prefix (256 bins in LDS, `h`), a wave scan over the bins
// finds the bin holding rank k, k drops by the count below it; a bin holding ONE key ends the search early (that
// key is looked up), which is the usual exit after two passes.
// CACHED: keys live in registers (W <= 64 * SCAN_E); otherwise every pass re-reads the row (cheap from LDS).
template <int E, class Row>      // E > 0: keys cached in E registers per lane (W <= 64 E); E = 0: re-read every pass
__device__ __forceinline__ float scan_noise(Row row, int low_noise, int W, unsigned *h, int lane)
{
    constexpr bool CACHED = E > 0;
    if (W <= 0) return key_to_float(0xffffffffu) * 2.0f;        // the reference indexes an empty array (undefined)
    unsigned keys[CACHED ? E : 1];
    if constexpr (CACHED) {
#pragma unroll
        for (int e = 0; e < E; ++e) {
            const int i = lane + 64 * e;
            // clamped index instead of a guarded load: no divergent branch, never outside the band
            keys[e] = order_key(row(low_noise + (i < W ? i : W - 1)));
        }
    }
    // f(key, valid) for every element slot of this lane (slots past the band: valid = false)
    auto for_keys = [&](auto f) {
        if constexpr (CACHED) {
#pragma unroll
            for (int e = 0; e < E; ++e)
                if (64 * e < W) f(keys[e], lane + 64 * e < W);
        } else {
            // sixteen loads in flight at a time (one trip to the row per 1024 columns, not per 64)
            for (int i0 = 0; i0 < W; i0 += 64 * SCAN_BATCH) {
                float x[SCAN_BATCH];
#pragma unroll
                for (int b = 0; b < SCAN_BATCH; ++b) {
                    const int i = i0 + 64 * b + lane;
                    x[b] = row(low_noise + (i < W ? i : W - 1));
                }
#pragma unroll
                for (int b = 0; b < SCAN_BATCH; ++b)
                    if (i0 + 64 * b < W) f(order_key(x[b]), i0 + 64 * b + lane < W);
            }
        }
    };
    unsigned kmin = 0xffffffffu, kmax = 0u;
    for_keys([&](unsigned key, bool valid) {
        kmin = min(kmin, valid ? key : 0xffffffffu);
        kmax = max(kmax, valid ? key : 0u);
    });
    kmin = wave_min_u32(kmin);
    kmax = wave_max_u32(kmax);
    unsigned result = kmin;                              // all keys equal: that value
    if (kmin != kmax) {
        int top = 31 - __clz((int)(kmin ^ kmax));        // highest bit on which two keys differ (wave-uniform)
        unsigned prefix = kmin;                          // bits above `top` are common to every key
        int k = W / 4;
        for (;;) {
            const int shift = top >= 7 ? top - 7 : 0;
            const unsigned mask = (2u << (top - shift)) - 1u;
            {
                // (a zero hipcc cannot hoist: as a loop invariant of the fused epilogue it went to scratch)
                unsigned z = 0u;
                asm volatile("" : "+v"(z));
                reinterpret_cast<uint4 *>(h)[lane] = make_uint4(z, z, z, z);
            }
            for_keys([&](unsigned key, bool valid) {
                // keys whose higher bits equal the prefix (shifting by 32 is not defined: top == 31 matches all);
                // the others add 0 -- no divergent branch around the atomic
                const bool match = top == 31 || ((key ^ prefix) >> (top + 1)) == 0u;
                atomicAdd(&h[(key >> shift) & mask], (valid && match) ? 1u : 0u);
            });
            const uint4 b = reinterpret_cast<const uint4 *>(h)[lane];          // bins 4 lane .. 4 lane + 3
            const unsigned s = b.x + b.y + b.z + b.w;
            const unsigned inc = wave_inclusive_sum(s);
            const unsigned exc = inc - s;
            const bool mine = exc <= (unsigned)k && (unsigned)k < inc;         // exactly one lane: total > k
            unsigned below = exc, bin = 0, count = b.x;
            if ((unsigned)k >= below + b.x) {
                below += b.x; bin = 1; count = b.y;
                if ((unsigned)k >= below + b.y) {
                    below += b.y; bin = 2; count = b.z;
                    if ((unsigned)k >= below + b.z) { below += b.z; bin = 3; count = b.w; }
                }
            }
            const int owner = __ffsll((long long)__ballot(mine)) - 1;                  // wave-uniform
            const unsigned digit = (unsigned)__builtin_amdgcn_readlane((int)(4 * lane + bin), owner);
            const unsigned in_bin = (unsigned)__builtin_amdgcn_readlane((int)count, owner);
            k -= __builtin_amdgcn_readlane((int)below, owner);
            prefix = (prefix & ~(mask << shift)) | (digit << shift);
            if (shift == 0) { result = prefix; break; }
            if (in_bin == 1u) {
                // one key carries this prefix: it is the answer, whatever its lower bits are
                unsigned found = 0u;
                for_keys([&](unsigned key, bool valid) {
                    if (valid && ((key ^ prefix) >> shift) == 0u) found = key;
                });
                result = wave_max_u32(found);            // every other lane holds 0 (and no key is 0: order_key)
                break;
            }
            top = shift - 1;
        }
    }
    return (float)((double)key_to_float(result) * 2.0);
}

// peak(): last index of the maximum of the detect band (src/BolidRecorder.cpp:323-335: `>=`, so ties go to the
// highest index).  Per-lane arg-max-last in index order, then a cross-lane reduction with "larger index wins".
template <int E, class Row>      // E > 0 and DW <= 64 E: all loads first; else one load per step
__device__ __forceinline__ int scan_peak(Row row, int low_detect, int DW, int lane)
{
    float best = 0.f;
    int best_i = -1;
    if (E > 0 && DW <= 64 * E) {
        // all loads first (one miss latency, not one per 64 columns)
        float xs[E > 0 ? E : 1];
#pragma unroll
        for (int e = 0; e < E; ++e) {
            const int i = lane + 64 * e;
            xs[e] = row(low_detect + (i < DW ? i : (DW > 0 ? DW - 1 : 0)));
        }
#pragma unroll
        for (int e = 0; e < E; ++e) {
            const int i = lane + 64 * e;
            if (i < DW && (best_i < 0 || xs[e] >= best)) { best = xs[e]; best_i = i; }
        }
    } else {
        for (int i0 = 0; i0 < DW; i0 += 64 * SCAN_BATCH) {
            float xs[SCAN_BATCH];
#pragma unroll
            for (int b = 0; b < SCAN_BATCH; ++b) {
                const int i = i0 + 64 * b + lane;
                xs[b] = row(low_detect + (i < DW ? i : DW - 1));
            }
#pragma unroll
            for (int b = 0; b < SCAN_BATCH; ++b) {
                const int i = i0 + 64 * b + lane;
                if (i < DW && (best_i < 0 || xs[b] >= best)) { best = xs[b]; best_i = i; }
            }
        }
    }
    // cross-lane: the largest value, then the largest index among the lanes that hold it (two DPP reductions)
    const unsigned bk = best_i < 0 ? 0u : order_key(best);             // order_key never gives 0 for a real value
    const unsigned top = wave_max_u32(bk);
    const unsigned cand = (best_i >= 0 && bk == top) ? (unsigned)best_i + 1u : 0u;
    best_i = (int)wave_max_u32(cand) - 1;
    return best_i < 0 ? 0 : best_i;
}

// average(): sequential double sum in index order, like the reference (src/BolidRecorder.cpp:338-347; window start
// :126-132).  64 columns are fetched at a time (one load per lane), then every lane adds them up in the same order
// from the other lanes' registers.  The reference reads outside the row when the window leaves it (UB); columns
// outside [0, bins) contribute nothing here.
template <class Row>
__device__ __forceinline__ float scan_average(Row row, int start, int avg_bins, int bins, int lane)
{
    double acc = 0.0;
    for (int base = 0; base < avg_bins; base += 64) {
        const int c = start + base + lane;
        const float x = (base + lane < avg_bins && c >= 0 && c < bins) ? row(c) : 0.f;
        const int n = avg_bins - base < 64 ? avg_bins - base : 64;
        for (int i = 0; i < n; ++i) {
            // v_readlane with a uniform lane number: a few cycles, where __shfl was an LDS round trip per element of
            // this dependent chain (out-of-row columns were loaded as 0 and add nothing, like before)
            acc += (double)__int_as_float(__builtin_amdgcn_readlane(__float_as_int(x), i));
        }
    }
    return (float)(acc / (double)avg_bins);
}

// ---------------------------------------------------------------------------
// sample loads
// ---------------------------------------------------------------------------
template <int FMT> struct Sample;
template <> struct Sample<RO_FMT_F32> {
    static constexpr int BYTES = 8;
    static __device__ __forceinline__ v2f load(__amdgpu_buffer_rsrc_t r, int voff, int soff)
    {
        return buf_load_f2(r, voff, soff);
    }
    // ... with the non-temporal hint: a sample nobody reads again
    static __device__ __forceinline__ v2f load_nt(__amdgpu_buffer_rsrc_t r, int voff, int soff)
    {
        const u32x2 t = __builtin_amdgcn_raw_buffer_load_b64(r, voff, soff, 2);
        return (v2f){__uint_as_float(t.x), __uint_as_float(t.y)};
    }
    // two adjacent samples with one 16-byte load
    static __device__ __forceinline__ void load_pair(__amdgpu_buffer_rsrc_t r, int voff, int soff, v2f &s0, v2f &s1)
    {
        const u32x4 t = __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0);
        s0 = (v2f){__uint_as_float(t.x), __uint_as_float(t.y)};
        s1 = (v2f){__uint_as_float(t.z), __uint_as_float(t.w)};
    }
};
template <> struct Sample<RO_FMT_I16> {
    static constexpr int BYTES = 4;
    static __device__ __forceinline__ v2f load(__amdgpu_buffer_rsrc_t r, int voff, int soff)
    {
        const unsigned u = __builtin_amdgcn_raw_buffer_load_b32(r, voff, soff, 0);
        return (v2f){(float)(short)(u & 0xffffu), (float)(short)(u >> 16)};
    }
    static __device__ __forceinline__ v2f load_nt(__amdgpu_buffer_rsrc_t r, int voff, int soff)
    {
        const unsigned u = __builtin_amdgcn_raw_buffer_load_b32(r, voff, soff, 2);
        return (v2f){(float)(short)(u & 0xffffu), (float)(short)(u >> 16)};
    }
    static __device__ __forceinline__ void load_pair(__amdgpu_buffer_rsrc_t r, int voff, int soff, v2f &s0, v2f &s1)
    {
        const u32x2 t = __builtin_amdgcn_raw_buffer_load_b64(r, voff, soff, 0);
        s0 = (v2f){(float)(short)(t.x & 0xffffu), (float)(short)(t.x >> 16)};
        s1 = (v2f){(float)(short)(t.y & 0xffffu), (float)(short)(t.y >> 16)};
    }
};

}  // namespace ro
