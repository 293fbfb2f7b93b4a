// Micro-benchmark: issue cost (shader cycles per wave-instruction and SIMD) of the FP64 VALU forms the strict-precision
// kernel (csrc/ro_f64reg.hip) is made of, and of the conversions and the float square root / reciprocal around them, as a
// function of the waves per SIMD; then the LDS forms of its exchanges (ds_write_b64 / ds_read_b64, lane-linear).
// One workgroup per CU; 256 / 512 / 1024 threads = 1 / 2 / 4 waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)

template <int KIND> __global__ void valu_k(unsigned long long *out, double seed, int iters)
{
    double a[8], b = seed * 0.5, c = 0.25;
    float f[8];
    for (int i = 0; i < 8; ++i) { a[i] = seed + i; f[i] = (float)seed + i; }
    __shared__ double lds[1024 * 17];
    const int t = threadIdx.x;
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        if constexpr (KIND == 0) {
#define X(i) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
            REP8(X) REP8(X)
#undef X
        } else if constexpr (KIND == 1) {
#define X(i) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(a[i]) : "v"(b));
            REP8(X) REP8(X)
#undef X
        } else if constexpr (KIND == 2) {
#define X(i) asm volatile("v_add_f64 %0, %0, %1" : "+v"(a[i]) : "v"(b));
            REP8(X) REP8(X)
#undef X
        } else if constexpr (KIND == 3) {     // dependent chain
#define X(i) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(a[0]) : "v"(b), "v"(c));
            REP8(X) REP8(X)
#undef X
        } else if constexpr (KIND == 4) {     // SGPR-pair operand
#define X(i) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(a[i]) : "s"(b), "v"(c));
            REP8(X) REP8(X)
#undef X
        } else if constexpr (KIND == 5) {
#define X(i) asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(a[i]) : "v"(f[i]));
            REP8(X) REP8(X)
#undef X
        } else if constexpr (KIND == 6) {
#define X(i) asm volatile("v_cvt_f32_f64 %0, %1" : "=v"(f[i]) : "v"(a[i]));
            REP8(X) REP8(X)
#undef X
        } else if constexpr (KIND == 7) {
#define X(i) asm volatile("v_sqrt_f32 %0, %0" : "+v"(f[i]));
            REP8(X) REP8(X)
#undef X
        } else if constexpr (KIND == 8) {
#define X(i) asm volatile("v_sqrt_f64 %0, %0" : "+v"(a[i]));
            REP8(X) REP8(X)
#undef X
        } else if constexpr (KIND == 9) {
#define X(i) asm volatile("v_rsq_f64 %0, %0" : "+v"(a[i]));
            REP8(X) REP8(X)
#undef X
        } else if constexpr (KIND == 10) {    // two dependent chains
#define X(i) asm volatile("v_fma_f64 %0, %0, %2, %3\n\tv_fma_f64 %1, %1, %2, %3" : "+v"(a[0]), "+v"(a[1]) : "v"(b), "v"(c));
            REP8(X)
#undef X
        } else if constexpr (KIND == 11) {    // ds_write_b64, lane-linear, 16 per iteration
#define X(i) lds[t + 1024 * i] = a[i];
            REP8(X)
#undef X
#define X(i) lds[t + 1024 * (8 + i)] = a[i];
            REP8(X)
#undef X
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        } else if constexpr (KIND == 12) {    // ds_read_b64, lane-linear, 16 per iteration
#define X(i) a[i] += lds[t + 1024 * i];
            REP8(X)
#undef X
#define X(i) a[i] += lds[t + 1024 * (8 + i)];
            REP8(X)
#undef X
            asm volatile("" ::: "memory");
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    double acc = 0;
    for (int i = 0; i < 8; ++i) acc += a[i] + f[i];
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * 16 + (threadIdx.x >> 6)] = t1 - t0;
    if (acc == 12345.678) out[0] = 0;
}

template <int KIND> void run(const char *name, int instr_per_iter, unsigned long long *d, int cus)
{
    const int iters = 2000;
    hipFuncSetAttribute(reinterpret_cast<const void *>(&valu_k<KIND>), hipFuncAttributeMaxDynamicSharedMemorySize, 0);
    for (int threads : {256, 512, 1024}) {
        hipLaunchKernelGGL(valu_k<KIND>, dim3(cus), dim3(threads), 0, 0, d, 1.0, iters);
        hipLaunchKernelGGL(valu_k<KIND>, dim3(cus), dim3(threads), 0, 0, d, 1.0, iters);
        if (hipDeviceSynchronize() != hipSuccess) { printf("%s: launch failed\n", name); return; }
        std::vector<unsigned long long> h((size_t)cus * 16);
        hipMemcpy(h.data(), d, h.size() * 8, hipMemcpyDeviceToHost);
        const int waves = threads / 64;
        double sum = 0;
        for (int b = 0; b < cus; ++b)
            for (int w = 0; w < waves; ++w) sum += (double)h[(size_t)b * 16 + w];
        const double per_wave = sum / (cus * waves);
        const double wave_instr = (double)iters * instr_per_iter;
        printf("%-44s %d waves/SIMD: %6.2f cycles per instruction and wave, %6.2f per instruction and SIMD\n", name,
               waves / 4, per_wave / wave_instr, per_wave / wave_instr / (waves / 4));
    }
}

int main()
{
    int cus = 0;
    hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0);
    unsigned long long *d;
    hipMalloc(&d, (size_t)cus * 16 * 8);
    run<0>("v_fma_f64, 8 independent", 16, d, cus);
    run<4>("v_fma_f64 SGPR operand, 8 independent", 16, d, cus);
    run<1>("v_mul_f64, 8 independent", 16, d, cus);
    run<2>("v_add_f64, 8 independent", 16, d, cus);
    run<3>("v_fma_f64, dependent chain", 16, d, cus);
    run<10>("v_fma_f64, two dependent chains", 16, d, cus);
    run<5>("v_cvt_f64_f32", 16, d, cus);
    run<6>("v_cvt_f32_f64", 16, d, cus);
    run<7>("v_sqrt_f32", 16, d, cus);
    run<8>("v_sqrt_f64", 16, d, cus);
    run<9>("v_rsq_f64", 16, d, cus);
    run<11>("ds_write_b64 lane-linear (+wait per 16)", 16, d, cus);
    run<12>("ds_read_b64 lane-linear", 16, d, cus);
    return 0;
}
