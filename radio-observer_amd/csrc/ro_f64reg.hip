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
#include "ro_kernels.h"
#include "ro_fft_device.h"
#include "ro_device_util.h"

#include <mutex>
#include <cmath>

// 1: (float)sqrt(double) as hipcc expands it (about fifteen FP64 operations); 0: float square root + one residual step in
// double (exact to ~1e-14 of a float ulp before the final rounding)
#ifndef RO_F64R_SQRT_EXACT
#define RO_F64R_SQRT_EXACT 0
#endif

namespace ro {
namespace f64r {

typedef double d2 __attribute__((ext_vector_type(2)));

template <int LOGM> struct Geo {
    static constexpr int M = 1 << LOGM, T = M / 16, R3 = M / 4096, Q = T / 16;
    static constexpr int ST = T + 16 * R3;          // doubles per k0 territory
    static constexpr int S2 = Q + R3, S3 = Q + 1;
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
template <int LVL, int BETA, typename TW> __device__ __forceinline__ void block16(double *re, double *im, const TW &tw)
{
    constexpr int L = 16 >> LVL, H = L / 2, E = brev<LVL>(BETA) * H;         // E = exponent of W16
    constexpr int IDX = LVL == 0 ? 0 : LVL == 1 ? 1 : LVL == 2 ? 2 + (E % 4) / 2 : 4 + E % 4;
    constexpr int ROT = E / 4;
#pragma unroll
    for (int m = 0; m < H; ++m) bfly<ROT>(re[BETA * L + m], im[BETA * L + m], re[BETA * L + m + H], im[BETA * L + m + H], tw.re(IDX), tw.im(IDX));
}
template <int LVL, typename TW, int... Bs>
__device__ __forceinline__ void level16(double *re, double *im, const TW &tw, std::integer_sequence<int, Bs...>)
{
    (block16<LVL, Bs>(re, im, tw), ...);
}
// twisted radix-16 transform in place: result kd at position bitrev4(kd)
template <typename TW> __device__ __forceinline__ void twisted16(double *re, double *im, const TW &tw)
{
    level16<0>(re, im, tw, std::make_integer_sequence<int, 1>{});
    level16<1>(re, im, tw, std::make_integer_sequence<int, 2>{});
    level16<2>(re, im, tw, std::make_integer_sequence<int, 4>{});
    level16<3>(re, im, tw, std::make_integer_sequence<int, 8>{});
}

// eight twiddles in VGPRs (per-lane table entry) or SGPRs (an entry the whole wave shares)
struct TwV {
    d2 t[8];
    __device__ __forceinline__ double re(int i) const { return t[i].x; }
    __device__ __forceinline__ double im(int i) const { return t[i].y; }
    // entry `index` of a table of `entries` (the base is the same for the whole wave, the index is the lane's)
    __device__ __forceinline__ void load(const double2 *table, int entries, int index)
    {
        const __amdgpu_buffer_rsrc_t rs = make_rsrc(table, (unsigned)entries * 128u);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const u32x4 u = __builtin_amdgcn_raw_buffer_load_b128(rs, index * 128, i * 16, 0);
            t[i] = (d2){__hiloint2double((int)u.y, (int)u.x), __hiloint2double((int)u.w, (int)u.z)};
        }
    }
};
struct TwS {
    d2 t[8];
    __device__ __forceinline__ double re(int i) const { return t[i].x; }
    __device__ __forceinline__ double im(int i) const { return t[i].y; }
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

// |X| as a float: src/WaterfallBackend.cpp:497-503 takes sqrt in double and narrows once
__device__ __forceinline__ float magnitude(double re, double im)
{
    const double s = __builtin_fma(im, im, re * re);
#if RO_F64R_SQRT_EXACT
    return (float)sqrt(s);
#else
    // r = sqrt_f32(float(s)) is within an ulp or two; with e = s - r^2 (exact product, one rounding) the root is
    // r + e / (2 r) up to e^2 / (8 r^3) < 2^-49 r, and the float sum r + c rounds that value once.  Outside the
    // range where float(s) is a normal number with headroom: the plain way.
    if (!(s > 0x1p-100 && s < 0x1p100)) return (float)sqrt(s);
    const float r = __builtin_amdgcn_sqrtf((float)s);
    const double rd = (double)r;
    const float e = (float)__builtin_fma(-rd, rd, s);
    const float h = 0.5f * __builtin_amdgcn_rcpf(r);
    return __builtin_fmaf(e, h, r);
#endif
}

// The fold of sub-row QQ: slot n0 of thread t = sum_r W_D^(r QQ) w[i + M r] (x[i + M r] + i gain), i = t + T n0
// (src/FFTBackend.cpp:78-79: Q += gain; :229-232: the window multiply, double x (double)float).  Products of a float32 or
// int16 sample and a float32 coefficient are exact in double.
template <int LOGM, int D, int QQ, int FMT>
__device__ __forceinline__ void fold(double *re, double *im, const __amdgpu_buffer_rsrc_t &rs_iq,
                                     const __amdgpu_buffer_rsrc_t &rs_w, int t, bool has_gain, double gain)
{
    using G = Geo<LOGM>;
    using S = Sample<FMT>;
    constexpr int M = G::M, T = G::T;
#pragma unroll
    for (int sg = 0; sg < 4; ++sg) {
        v2f x[D][4];
        v4f w[D];
#pragma unroll
        for (int r = 0; r < D; ++r) {
#pragma unroll
            for (int e = 0; e < 4; ++e) x[r][e] = S::load(rs_iq, t * S::BYTES, (T * (4 * sg + e) + M * r) * S::BYTES);
            const u32x4 c = __builtin_amdgcn_raw_buffer_load_b128(rs_w, t * 16, (r * 4 + sg) * T * 16, 0);
            w[r] = (v4f){__uint_as_float(c.x), __uint_as_float(c.y), __uint_as_float(c.z), __uint_as_float(c.w)};
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            double sr = 0.0, si = 0.0;
#pragma unroll
            for (int r = 0; r < D; ++r) {
                const double xr = (double)x[r][e].x;
                double xi = (double)x[r][e].y;
                if (has_gain) xi += gain;
                const double wd = (double)w[r][e];
                const int ph = ((r * QQ) % D) * (4 / D);                  // the factor is (-i)^ph
                if (r == 0) {
                    sr = xr * wd;
                    si = xi * wd;
                } else if (ph == 0) {
                    sr = __builtin_fma(xr, wd, sr);
                    si = __builtin_fma(xi, wd, si);
                } else if (ph == 1) {                                      // -i (xr + i xi) = xi - i xr
                    sr = __builtin_fma(xi, wd, sr);
                    si = __builtin_fma(-xr, wd, si);
                } else if (ph == 2) {
                    sr = __builtin_fma(-xr, wd, sr);
                    si = __builtin_fma(-xi, wd, si);
                } else {                                                   // i (xr + i xi) = -xi + i xr
                    sr = __builtin_fma(-xi, wd, sr);
                    si = __builtin_fma(xr, wd, si);
                }
            }
            re[4 * sg + e] = sr;
            im[4 * sg + e] = si;
        }
    }
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
};

template <int LOGM, int D, int FMT> __global__ __launch_bounds__((1 << LOGM) / 16, 4) void f64r_kernel(Args a)
{
    using G = Geo<LOGM>;
    using S = Sample<FMT>;
    constexpr int M = G::M, T = G::T, R3 = G::R3, Q = G::Q, ST = G::ST, N = M * D;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    double *plane = reinterpret_cast<double *>(smem);
    float *image = reinterpret_cast<float *>(smem);

    // XCD-aware placement (speed only): workgroups b and b + 8 share an XCD under round-robin dispatch; each XCD takes
    // a contiguous run of rows, its workgroups take the D sub-rows of consecutive rows at the same time
    const int64_t per_xcd = (a.rows + 7) / 8;
    const int64_t xcd_first = (int64_t)(blockIdx.x & 7) * per_xcd;
    const int64_t xcd_end = xcd_first + per_xcd < a.rows ? xcd_first + per_xcd : a.rows;
    const int slots = gridDim.x >> 3;                  // a multiple of D
    const int slot = blockIdx.x >> 3;
    const int q = slot % D;
    const int64_t row_step = slots / D;
    int64_t row = xcd_first + slot / D;
    if (row >= xcd_end) return;

    const int t = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int lane = t & 63;
    const char *iq = reinterpret_cast<const char *>(a.iq);

    // ---- thread roles (fixed for the kernel)
    const int k0 = t / Q, u = t % Q, k1 = u / R3, n3 = u % R3;
    const int K1 = k0 + 16 * k1;
    // LDS cells (doubles / floats; see the header)
    const int x1w = t, x1r = k0 * ST + u;
    const int x2w = k0 * ST + u, x2r = k0 * ST + k1 * G::S2 + n3;
    const int x3w = k0 * ST + u, x3r = k0 * ST + n3 * G::S3 + k1 * R3;
    // image: writer (k0, s, u), reader (k0 = lane & 15, s = it, u = (lane >> 4) | (wave << 2))
    const int rk0 = lane & 15, ru = (lane >> 4) | (wave << 2);
    int imw, imr, imw_odd = 0, imr_odd = 0;
    if constexpr (R3 == 1) {
        // cell = 2 k0 ST + (s ^ (k0 & 1)) 16 + ((u + 2 (k0 >> 1)) & 15): even / odd s from two bases
        const int bw = 2 * k0 * ST + ((u + 2 * (k0 >> 1)) & 15), br_ = 2 * rk0 * ST + ((ru + 2 * (rk0 >> 1)) & 15);
        imw = bw + 16 * (k0 & 1);
        imw_odd = bw - 16 * (k0 & 1);
        imr = br_ + 16 * (rk0 & 1);
        imr_odd = br_ - 16 * (rk0 & 1);
    } else {
        imw = 2 * k0 * ST + ((u + 2 * k0) & (Q - 1));
        imr = 2 * rk0 * ST + ((ru + 2 * rk0) & (Q - 1));
    }
    // read-out: bin = rk0 + 16 rk1 + 256 (rg + R3 i) + 4096 k3 for slot s = i R3 + p (R3 > 1), rk0 + 16 ru + 256 bitrev4(s) else
    const int rbin = rk0 + 16 * (ru / R3) + 256 * (ru % R3);

    const double2 *tw0 = a.tw0 + q * 8;
    const double2 *tw1 = a.tw1 + q * 16 * 8;
    const double2 *tw2 = a.tw2 + q * 256 * 8;
    const bool has_gain = a.gain != 0.0;

    auto readout = [&](int64_t prow) {
        const __amdgpu_buffer_rsrc_t rs = make_rsrc(a.rows_out + prow * a.row_stride, N * 4);
        const int voff = (q + D * rbin) * 4;
        float m[16];
#pragma unroll
        for (int s = 0; s < 16; ++s) {
            if constexpr (R3 == 1) m[s] = image[((s & 1) ? imr_odd : imr) + 16 * s];
            else m[s] = image[imr + Q * s];
        }
#pragma unroll
        for (int s = 0; s < 16; ++s) {
            // column (k + N/2) mod N of the row holds |X[k]| (src/WaterfallBackend.cpp:492-505): the slot's share of
            // the bin only has bits above the lane's share, and N/2 flips the top one
            const int sbin = R3 == 1 ? 256 * brev<4>(s) : 256 * R3 * (s / R3) + 4096 * brev<G::L3>(s % R3);
            const int soff = ((D * sbin) ^ (N / 2)) * 4;
            __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(m[s]), rs, voff, soff, RO_STORE_AUX);
        }
    };

    bool have_prev = false;
    int64_t prev_row = 0;
    for (;;) {
        double re[16], im[16];
        // ---- samples and window of this sub-row; the fold (q is the same for the whole workgroup: one copy per q, the
        // factors W_D^(r q) = (-i)^(r q (4 / D)) are then signs and swaps inside the FMAs)
        {
            const __amdgpu_buffer_rsrc_t rs_iq =
                make_rsrc(iq + (a.first_row + row) * (int64_t)a.hop * S::BYTES, (unsigned)N * S::BYTES);
            const __amdgpu_buffer_rsrc_t rs_w = make_rsrc(a.window_k, (unsigned)N * 4);
            if constexpr (D == 1) {
                fold<LOGM, D, 0, FMT>(re, im, rs_iq, rs_w, t, has_gain, a.gain);
            } else if constexpr (D == 2) {
                if (q == 0) fold<LOGM, D, 0, FMT>(re, im, rs_iq, rs_w, t, has_gain, a.gain);
                else fold<LOGM, D, 1, FMT>(re, im, rs_iq, rs_w, t, has_gain, a.gain);
            } else {
                if (q == 0) fold<LOGM, D, 0, FMT>(re, im, rs_iq, rs_w, t, has_gain, a.gain);
                else if (q == 1) fold<LOGM, D, 1, FMT>(re, im, rs_iq, rs_w, t, has_gain, a.gain);
                else if (q == 2) fold<LOGM, D, 2, FMT>(re, im, rs_iq, rs_w, t, has_gain, a.gain);
                else fold<LOGM, D, 3, FMT>(re, im, rs_iq, rs_w, t, has_gain, a.gain);
            }
        }
        // ---- the image of the sub-row before this one leaves while the samples arrive
        if (have_prev) readout(prev_row);

        // ---- pass 0
        {
            TwS tw;
            tw.load(tw0);
            twisted16(re, im, tw);
        }
        wg_sync();                                      // (a) the old image has been read: LDS is free
        // ---- exchange 1
        double xr[16], xi[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) plane[x1w + k * ST] = re[brev<4>(k)];
        wg_sync();                                      // (b)
#pragma unroll
        for (int j = 0; j < 16; ++j) xr[j] = plane[x1r + Q * j];
        wg_sync();                                      // (c) everyone has its real parts
#pragma unroll
        for (int k = 0; k < 16; ++k) plane[x1w + k * ST] = im[brev<4>(k)];
        wg_sync();                                      // (d)
#pragma unroll
        for (int j = 0; j < 16; ++j) xi[j] = plane[x1r + Q * j];
        asm volatile("" ::: "memory");
        // ---- pass 1.  From here to the completed image a wave touches its own territories only.
        if constexpr (Q == 64) {
            TwS tw;                                      // the wave is one k0
            tw.load(tw1 + wave * 8);
            twisted16(xr, xi, tw);
        } else {
            TwV tw;
            tw.load(tw1, 16, k0);
            twisted16(xr, xi, tw);
        }
        // ---- exchange 2 (one wave's LDS instructions execute in order: no wait between its writes and its reads)
#pragma unroll
        for (int k = 0; k < 16; ++k) plane[x2w + k * G::S2] = xr[brev<4>(k)];
        asm volatile("" ::: "memory");
#pragma unroll
        for (int j = 0; j < 16; ++j) re[j] = plane[x2r + R3 * j];
        asm volatile("" ::: "memory");
#pragma unroll
        for (int k = 0; k < 16; ++k) plane[x2w + k * G::S2] = xi[brev<4>(k)];
        asm volatile("" ::: "memory");
#pragma unroll
        for (int j = 0; j < 16; ++j) im[j] = plane[x2r + R3 * j];
        asm volatile("" ::: "memory");
        // ---- pass 2
        {
            TwV tw;
            tw.load(tw2, 256, K1);
            twisted16(re, im, tw);
        }
        if constexpr (R3 > 1) {
            // ---- exchange 3: slot k2 -> cell k2 S3 + u; thread g = n3 reads k2 = g + R3 i, all n3
#pragma unroll
            for (int k = 0; k < 16; ++k) plane[x3w + k * G::S3] = re[brev<4>(k)];
            asm volatile("" ::: "memory");
#pragma unroll
            for (int i = 0; i < 16 / R3; ++i)
#pragma unroll
                for (int m = 0; m < R3; ++m) xr[i * R3 + m] = plane[x3r + i * R3 * G::S3 + m];
            asm volatile("" ::: "memory");
#pragma unroll
            for (int k = 0; k < 16; ++k) plane[x3w + k * G::S3] = im[brev<4>(k)];
            asm volatile("" ::: "memory");
#pragma unroll
            for (int i = 0; i < 16 / R3; ++i)
#pragma unroll
                for (int m = 0; m < R3; ++m) xi[i * R3 + m] = plane[x3r + i * R3 * G::S3 + m];
            asm volatile("" ::: "memory");
            // ---- pass 3: butterfly i has twist theta = a' W_16^i, a' = W_N^(q + D (K1 + 256 g)) (table: {a', a'^2})
            d2 a1, a2;
            {
                const __amdgpu_buffer_rsrc_t rs = make_rsrc(a.tw3 + (int64_t)q * 256 * R3 * 2, 256 * R3 * 32);
                const u32x4 c1 = __builtin_amdgcn_raw_buffer_load_b128(rs, (K1 * R3 + n3) * 32, 0, 0);
                const u32x4 c2 = __builtin_amdgcn_raw_buffer_load_b128(rs, (K1 * R3 + n3) * 32, 16, 0);
                a1 = (d2){__hiloint2double((int)c1.y, (int)c1.x), __hiloint2double((int)c1.w, (int)c1.z)};
                a2 = (d2){__hiloint2double((int)c2.y, (int)c2.x), __hiloint2double((int)c2.w, (int)c2.z)};
            }
            constexpr double C16[8] = {1.0, 0.92387953251128675613, 0.70710678118654752440, 0.38268343236508977173,
                                       0.0, -0.38268343236508977173, -0.70710678118654752440, -0.92387953251128675613};
            constexpr double S16[8] = {0.0, 0.38268343236508977173, 0.70710678118654752440, 0.92387953251128675613,
                                       1.0, 0.92387953251128675613, 0.70710678118654752440, 0.38268343236508977173};
#pragma unroll
            for (int i = 0; i < 16 / R3; ++i) {
                // theta = a1 (c - i s), c + i s = exp(2 pi i i / 16)
                const double c = C16[i], s = S16[i];
                const double tr = i == 0 ? a1.x : a1.x * c + a1.y * s, ti = i == 0 ? a1.y : a1.y * c - a1.x * s;
                if constexpr (R3 == 2) {
                    bfly<0>(xr[2 * i], xi[2 * i], xr[2 * i + 1], xi[2 * i + 1], tr, ti);
                } else {
                    // theta^2 = a2 W_8^i
                    const double c2 = C16[2 * i], s2 = S16[2 * i];
                    const double ur = i == 0 ? a2.x : a2.x * c2 + a2.y * s2, ui = i == 0 ? a2.y : a2.y * c2 - a2.x * s2;
                    bfly<0>(xr[4 * i], xi[4 * i], xr[4 * i + 2], xi[4 * i + 2], ur, ui);
                    bfly<0>(xr[4 * i + 1], xi[4 * i + 1], xr[4 * i + 3], xi[4 * i + 3], ur, ui);
                    bfly<0>(xr[4 * i], xi[4 * i], xr[4 * i + 1], xi[4 * i + 1], tr, ti);
                    bfly<1>(xr[4 * i + 2], xi[4 * i + 2], xr[4 * i + 3], xi[4 * i + 3], tr, ti);
                }
            }
            // ---- magnitudes into the image (own territory)
#pragma unroll
            for (int s = 0; s < 16; ++s) image[imw + Q * s] = magnitude(xr[s], xi[s]);
        } else {
#pragma unroll
            for (int s = 0; s < 16; ++s) image[((s & 1) ? imw_odd : imw) + 16 * s] = magnitude(re[s], im[s]);
        }
        wg_sync();                                      // (e) the image of this sub-row is complete
        have_prev = true;
        prev_row = row;
        row += row_step;
        if (row >= xcd_end) break;
    }
    readout(prev_row);
}

template <int LOGM, int D, int FMT> static hipError_t launch_one(const Args &a, hipStream_t s)
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
            const void *fn = reinterpret_cast<const void *>(&f64r_kernel<LOGM, D, FMT>);
            if ((e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, G::LDS_BYTES)) != hipSuccess) return e;
            if ((e = hipDeviceGetAttribute(&cus_of[dev], hipDeviceAttributeMultiprocessorCount, dev)) != hipSuccess) return e;
            ready[dev] = true;
        }
        cus = cus_of[dev];
    }
    // persistent grid: 1024 / T workgroups per CU, per XCD a multiple of D, never more than the XCD's share of sub-rows
    const int64_t per_xcd = (a.rows + 7) / 8;
    int64_t slots = (int64_t)(cus / 8) * (1024 / G::T);
    if (slots > per_xcd * D) slots = per_xcd * D;
    slots = slots / D * D;
    if (slots < D) slots = D;
    hipLaunchKernelGGL((f64r_kernel<LOGM, D, FMT>), dim3((unsigned)(slots * 8)), dim3(G::T), G::LDS_BYTES, s, a);
    return hipGetLastError();
}

template <int FMT> static hipError_t launch_fmt(int m_log2, int dec, const Args &a, hipStream_t s)
{
#define RO_F64R_CASE(LM, DD) \
    if (m_log2 == LM && dec == DD) return launch_one<LM, DD, FMT>(a, s)
    RO_F64R_CASE(12, 1);
    RO_F64R_CASE(13, 1);
    RO_F64R_CASE(14, 1);
    RO_F64R_CASE(14, 2);
    RO_F64R_CASE(14, 4);
#undef RO_F64R_CASE
    return hipErrorInvalidValue;
}

static bool plan(int bins, int &m_log2, int &dec)
{
    switch (bins) {
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
    int m, d;
    return f64r::plan(bins, m, d);
}

void f64reg_tables(int bins, const float *window, F64RegTables &t)
{
    int m_log2 = 0, D = 0;
    if (!f64r::plan(bins, m_log2, D)) return;
    const int M = 1 << m_log2, T = M / 16, R3 = M / 4096;
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
    int m_log2 = 0, dec = 0;
    if (!f64r::plan(bins, m_log2, dec)) return hipErrorInvalidValue;
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
    if (fmt == RO_FMT_F32) return f64r::launch_fmt<RO_FMT_F32>(m_log2, dec, b, s);
    if (fmt == RO_FMT_I16) return f64r::launch_fmt<RO_FMT_I16>(m_log2, dec, b, s);
    return hipErrorInvalidValue;
}

}  // namespace ro
