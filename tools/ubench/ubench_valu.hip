// Micro-benchmark: issue cost (shader cycles per wave-instruction and SIMD) of the fp32 VALU forms the STFT butterflies
// can be written in, as a function of the waves per SIMD: v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32 (two fp32 per
// lane and instruction) against v_fma_f32 / v_mul_f32 / v_add_f32, independent chains (8 accumulators per lane) and
// one dependent chain.  One workgroup per CU; 256 / 512 / 1024 threads = 1 / 2 / 4 waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float v2f __attribute__((ext_vector_type(2)));

#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)

template <int KIND> __global__ void valu_k(unsigned long long *out, float seed, int iters)
{
    v2f a[8], b = (v2f){seed, seed * 0.5f}, c = (v2f){0.25f, 0.125f};
    float s[16];
    for (int i = 0; i < 8; ++i) a[i] = (v2f){seed + i, seed - i};
    for (int i = 0; i < 16; ++i) s[i] = seed + i;
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        if constexpr (KIND == 0) {            // 8 independent v_pk_fma_f32
#define X(i) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
            REP8(X) REP8(X)
#undef X
        } else if constexpr (KIND == 1) {     // 16 independent v_fma_f32
#define X(i) asm volatile("v_fma_f32 %0, %0, %2, %3\n\tv_fma_f32 %1, %1, %2, %3" : "+v"(s[2 * i]), "+v"(s[2 * i + 1]) : "v"(b.x), "v"(c.x));
            REP8(X) REP8(X)
#undef X
        } else if constexpr (KIND == 2) {     // dependent chain of v_pk_fma_f32
#define X(i) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(a[0]) : "v"(b), "v"(c));
            REP8(X) REP8(X)
#undef X
        } else if constexpr (KIND == 3) {     // dependent chain of v_fma_f32
#define X(i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(s[0]) : "v"(b.x), "v"(c.x));
            REP8(X) REP8(X)
#undef X
        } else if constexpr (KIND == 4) {     // 8 independent v_pk_mul_f32
#define X(i) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
            REP8(X) REP8(X)
#undef X
        } else if constexpr (KIND == 5) {     // 8 independent v_pk_add_f32
#define X(i) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
            REP8(X) REP8(X)
#undef X
        } else if constexpr (KIND == 6) {     // v_pk_fma_f32 with op_sel / neg modifiers as the butterflies use them
#define X(i) asm volatile("v_pk_fma_f32 %0, %0, %1, %2 op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_lo:[0,1,0]" : "+v"(a[i]) : "v"(b), "v"(c));
            REP8(X) REP8(X)
#undef X
        } else if constexpr (KIND == 7) {     // two chains interleaved (what SEQ_G = 2 leaves): dependent pairs
#define X(i) asm volatile("v_pk_fma_f32 %0, %0, %2, %3\n\tv_pk_fma_f32 %1, %1, %2, %3" : "+v"(a[0]), "+v"(a[1]) : "v"(b), "v"(c));
            REP8(X)
#undef X
        } else if constexpr (KIND == 8) {     // v_permlane32_swap (window stage)
#define X(i) asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(s[2 * i]), "+v"(s[2 * i + 1]));
            REP8(X) REP8(X)
#undef X
        } else if constexpr (KIND == 9) {     // pk_fma with an SGPR-pair constant operand (constant twiddles)
#define X(i) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "s"(b), "v"(c));
            REP8(X) REP8(X)
#undef X
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float acc = 0;
    for (int i = 0; i < 8; ++i) acc += a[i].x + a[i].y;
    for (int i = 0; i < 16; ++i) acc += s[i];
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * 16 + (threadIdx.x >> 6)] = t1 - t0;
    if (acc == 12345.678f) out[0] = 0;
}

template <int KIND> void run(const char *name, int instr_per_iter, unsigned long long *d, int cus)
{
    const int iters = 2000;
    for (int threads : {256, 512, 1024}) {
        hipLaunchKernelGGL(valu_k<KIND>, dim3(cus), dim3(threads), 0, 0, d, 1.0f, iters);
        hipLaunchKernelGGL(valu_k<KIND>, dim3(cus), dim3(threads), 0, 0, d, 1.0f, iters);
        hipDeviceSynchronize();
        std::vector<unsigned long long> h((size_t)cus * 16);
        hipMemcpy(h.data(), d, h.size() * 8, hipMemcpyDeviceToHost);
        const int waves = threads / 64;
        double sum = 0;
        for (int b = 0; b < cus; ++b)
            for (int w = 0; w < waves; ++w) sum += (double)h[(size_t)b * 16 + w];
        const double per_wave = sum / (cus * waves);                       // cycles one wave needed for the loop
        const double wave_instr = (double)iters * instr_per_iter;
        // per SIMD: waves/4 waves share it
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
    run<0>("v_pk_fma_f32, 8 independent", 16, d, cus);
    run<6>("v_pk_fma_f32 op_sel/neg, 8 independent", 16, d, cus);
    run<9>("v_pk_fma_f32 SGPR operand, 8 independent", 16, d, cus);
    run<1>("v_fma_f32, 16 independent", 32, d, cus);
    run<2>("v_pk_fma_f32, dependent chain", 16, d, cus);
    run<7>("v_pk_fma_f32, two dependent chains", 16, d, cus);
    run<3>("v_fma_f32, dependent chain", 16, d, cus);
    run<4>("v_pk_mul_f32, 8 independent", 16, d, cus);
    run<5>("v_pk_add_f32, 8 independent", 16, d, cus);
    run<8>("v_permlane32_swap_b32", 16, d, cus);
    return 0;
}
