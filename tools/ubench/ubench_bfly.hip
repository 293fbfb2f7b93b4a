// Micro-benchmark: the STFT kernel's own radix-32 butterfly code (ro_fft_device.h: dit<32>, fdit32) in a loop on
// registers, no memory traffic: shader cycles per radix-32 pass and wave at 1 / 2 / 4 waves per SIMD, and the cycles
// per packed VALU instruction that implies.  Tells whether the butterflies as hipcc schedules them reach the issue rate
// tools/ubench/ubench_valu.hip measures for bare v_pk_fma_f32 streams.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include "../../radio-observer_amd/csrc/ro_fft_device.h"
using namespace ro;

template <int KIND> __global__ __launch_bounds__(1024) void bfly_k(unsigned long long *out, float seed, int iters)
{
    v2f v[32];
    for (int i = 0; i < 32; ++i) v[i] = (v2f){seed + i + threadIdx.x * 1e-3f, seed - i};
    v2f g1 = (v2f){0.999f, -0.01f}, g2 = (v2f){0.998f, -0.02f}, g4 = (v2f){0.99f, -0.04f}, g8 = (v2f){0.98f, -0.08f},
        g16 = (v2f){0.96f, -0.16f};
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        if constexpr (KIND == 0) dit<32>(v);
        else if constexpr (KIND == 1) fdit32(v, g16, g8, g4, g2, g1);
        else {
            fdit32_head(v, g16, g8, g4, g2);
            fdit32_last(v, g1, [&](auto jc) { return v[17 + 2 * decltype(jc)::value].y; });
        }
#pragma unroll
        for (int i = 0; i < 32; ++i) v[i] = v[i] * (v2f){0.03125f, 0.03125f};      // keep the values bounded: 32 more ops
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float acc = 0;
    for (int i = 0; i < 32; ++i) acc += v[i].x + v[i].y;
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * 16 + (threadIdx.x >> 6)] = t1 - t0;
    if (acc == 12345.678f) out[0] = 0;
}

template <int KIND> void run(const char *name, int ops, unsigned long long *d, int cus)
{
    const int iters = 200;
    for (int threads : {256, 512, 1024}) {
        for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL(bfly_k<KIND>, dim3(cus), dim3(threads), 0, 0, d, 1.0f, iters);
        (void)hipDeviceSynchronize();
        std::vector<unsigned long long> h((size_t)cus * 16);
        (void)hipMemcpy(h.data(), d, h.size() * 8, hipMemcpyDeviceToHost);
        const int waves = threads / 64;
        double sum = 0;
        for (int b = 0; b < cus; ++b)
            for (int w = 0; w < waves; ++w) sum += (double)h[(size_t)b * 16 + w];
        const double per_pass = sum / (cus * waves) / iters;
        printf("%-34s %d waves/SIMD: %7.0f cycles per pass and wave = %5.2f per packed op and wave, %5.2f per op and SIMD (%d ops)\n",
               name, waves / 4, per_pass, per_pass / ops, per_pass / ops / (waves / 4), ops);
    }
}

int main()
{
    int cus = 0;
    (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0);
    unsigned long long *d;
    (void)hipMalloc(&d, (size_t)cus * 16 * 8);
    // op counts: dit<32> 80 butterflies (2 ops for w = 1 / -i, 3 otherwise) ~ 208 + 32 scaling; fdit32 240 + 22 + 32
    run<0>("dit<32> (constant twiddles)", 240, d, cus);
    run<1>("fdit32 (stage twiddles)", 294, d, cus);
    run<2>("fdit32_head + fdit32_last", 294, d, cus);
    return 0;
}
