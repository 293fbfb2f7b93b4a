// ro_narrow.h -- the one narrowing of struct Complex (two doubles) to the float pairs of the raw ring, which are the GPU
// path's RO_IQ_F32 as well: FFTBackend::floatToInt(Complex, float*) (src/FFTBackend.h:258-262) a call's worth at a time; shared by the host mirror's pushRaw (host/HipWaterfallBackend.cpp) and
// ro_stft_push's RO_IQ_F64 deliveries (ro_stft_capi.cpp).
// At -O2 g++ leaves the plain loop scalar, and it was most of the host's time per row.  The level is chosen once at run
// time (the library is built on one machine and runs on another): AVX-512 (one conversion makes the eight floats of a
// store), AVX (two conversions and an insert per store: the PC samples of tools/r5/host_sample.py had 63 % of the host
// thread's time in that loop at 256 rows per launch), SSE2, scalar.  cvtpd2ps rounds like the cast under the default
// rounding mode (tests/test_host_cpu.py holds every level against it).
#pragma once
#if defined(__x86_64__) && defined(__GNUC__)
#include <immintrin.h>
#include <cstdint>
#define RO_NARROW_X86 1
#endif

namespace ro {

inline void narrowScalar(const double *src, float *dst, int count)
{
    for (int i = 0; i < count; ++i) dst[i] = (float)src[i];
}

#ifdef RO_NARROW_X86
inline void narrowSse2(const double *src, float *dst, int count)
{
    int i = 0;
    for (; i + 4 <= count; i += 4) {
        const __m128 lo = _mm_cvtpd_ps(_mm_loadu_pd(src + i)), hi = _mm_cvtpd_ps(_mm_loadu_pd(src + i + 2));
        _mm_storeu_ps(dst + i, _mm_movelh_ps(lo, hi));
    }
    narrowScalar(src + i, dst + i, count - i);
}

__attribute__((target("avx"))) inline void narrowAvx(const double *src, float *dst, int count)
{
    int i = 0;
    for (; i + 8 <= count; i += 8) {
        const __m128 lo = _mm256_cvtpd_ps(_mm256_loadu_pd(src + i)), hi = _mm256_cvtpd_ps(_mm256_loadu_pd(src + i + 4));
        _mm256_storeu_ps(dst + i, _mm256_set_m128(hi, lo));
    }
    for (; i < count; ++i) dst[i] = (float)src[i];
}

__attribute__((target("avx512f"))) inline void narrowAvx512(const double *src, float *dst, int count)
{
    int i = 0;
    for (; i + 32 <= count; i += 32) {
        const __m256 a = _mm512_cvtpd_ps(_mm512_loadu_pd(src + i)), b = _mm512_cvtpd_ps(_mm512_loadu_pd(src + i + 8));
        const __m256 c = _mm512_cvtpd_ps(_mm512_loadu_pd(src + i + 16)), d = _mm512_cvtpd_ps(_mm512_loadu_pd(src + i + 24));
        _mm256_storeu_ps(dst + i, a);
        _mm256_storeu_ps(dst + i + 8, b);
        _mm256_storeu_ps(dst + i + 16, c);
        _mm256_storeu_ps(dst + i + 24, d);
    }
    for (; i + 8 <= count; i += 8) _mm256_storeu_ps(dst + i, _mm512_cvtpd_ps(_mm512_loadu_pd(src + i)));
    for (; i < count; ++i) dst[i] = (float)src[i];
}
#endif

#ifdef RO_NARROW_X86
// The same loops with non-temporal stores, for destinations that are written once and not read by this core soon -- the
// raw ring (megabytes, read again only when an event is captured) and a staging buffer too large for the caches (the
// copy engine reads it from memory): an ordinary store first READS the line it is about to overwrite, which doubled the
// memory traffic of the host thread.  Scalar head until dst is aligned to the store; sfence behind the call, so that
// whoever is told about the data next (a DMA launch) finds it in memory.
__attribute__((target("avx"))) inline void narrowAvxStream(const double *src, float *dst, int count)
{
    int i = 0;
    while (i < count && ((uintptr_t)(dst + i) & 31)) { dst[i] = (float)src[i]; ++i; }
    for (; i + 8 <= count; i += 8) {
        const __m128 lo = _mm256_cvtpd_ps(_mm256_loadu_pd(src + i)), hi = _mm256_cvtpd_ps(_mm256_loadu_pd(src + i + 4));
        _mm256_stream_ps(dst + i, _mm256_set_m128(hi, lo));
    }
    for (; i < count; ++i) dst[i] = (float)src[i];
    _mm_sfence();
}

__attribute__((target("avx512f"))) inline void narrowAvx512Stream(const double *src, float *dst, int count)
{
    int i = 0;
    while (i < count && ((uintptr_t)(dst + i) & 63)) { dst[i] = (float)src[i]; ++i; }
    for (; i + 16 <= count; i += 16) {
        const __m256 a = _mm512_cvtpd_ps(_mm512_loadu_pd(src + i)), b = _mm512_cvtpd_ps(_mm512_loadu_pd(src + i + 8));
        _mm512_stream_ps(dst + i, _mm512_castpd_ps(_mm512_insertf64x4(_mm512_castpd256_pd512(_mm256_castps_pd(a)), _mm256_castps_pd(b), 1)));
    }
    for (; i < count; ++i) dst[i] = (float)src[i];
    _mm_sfence();
}
#endif

// 3: AVX-512, 2: AVX, 1: SSE2, 0: scalar -- the best this CPU has
inline int narrowLevel()
{
#ifdef RO_NARROW_X86
    static const int level = __builtin_cpu_supports("avx512f") ? 3 : __builtin_cpu_supports("avx") ? 2 : 1;
    return level;
#else
    return 0;
#endif
}

// `level` above what the CPU has is the caller's mistake (the tests ask narrowLevel() first)
inline void narrowWith(int level, const double *src, float *dst, int count)
{
#ifdef RO_NARROW_X86
    if (level >= 3) return narrowAvx512(src, dst, count);
    if (level == 2) return narrowAvx(src, dst, count);
    if (level == 1) return narrowSse2(src, dst, count);
#endif
    (void)level;
    narrowScalar(src, dst, count);
}

inline void narrowToFloat(const double *src, float *dst, int count) { narrowWith(narrowLevel(), src, dst, count); }

// stream = true: the non-temporal form where the CPU has one (levels 2 and 3), else the ordinary loops
inline void narrowWith(int level, bool stream, const double *src, float *dst, int count)
{
#ifdef RO_NARROW_X86
    if (stream && level >= 3) return narrowAvx512Stream(src, dst, count);
    if (stream && level == 2) return narrowAvxStream(src, dst, count);
#endif
    narrowWith(level, src, dst, count);
}

inline void narrowToFloatStream(const double *src, float *dst, int count) { narrowWith(narrowLevel(), true, src, dst, count); }

}  // namespace ro
