// Micro-benchmark: per-CU store / load throughput as a function of how many CUs are active.
// Each workgroup (1024 threads) streams `rows` rows of 128 KiB (stores) or 256 KiB (loads) with
// 16-byte-per-lane buffer instructions, exactly the shapes the STFT kernel uses.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ __amdgpu_buffer_rsrc_t rsrc(const void *p, unsigned bytes)
{
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(p), 0, (int)bytes, 0x00020000);
}
__global__ __launch_bounds__(1024) void store_k(float *out, int rows_per_wg, int total_rows)
{
    const int tid = threadIdx.x;
    u32x4 v = {(unsigned)tid, 1u, 2u, 3u};
    for (int i = 0; i < rows_per_wg; ++i) {
        const long row = ((long)blockIdx.x + (long)i * gridDim.x) % total_rows;
        __amdgpu_buffer_rsrc_t r = rsrc(out + row * 32768, 32768 * 4);
#pragma unroll
        for (int q = 0; q < 8; ++q) __builtin_amdgcn_raw_buffer_store_b128(v, r, tid * 16, q * 16384, 0);
    }
}
__global__ __launch_bounds__(1024) void load_k(const float *in, float *sink, int rows_per_wg, int total_rows, int hop_floats)
{
    const int tid = threadIdx.x;
    u32x4 acc = {0, 0, 0, 0};
    for (int i = 0; i < rows_per_wg; ++i) {
        const long row = ((long)blockIdx.x + (long)i * gridDim.x) % total_rows;
        __amdgpu_buffer_rsrc_t r = rsrc(in + row * hop_floats, 65536 * 4);
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            u32x4 t = __builtin_amdgcn_raw_buffer_load_b128(r, tid * 16, q * 16384, 0);
            acc.x ^= t.x; acc.y ^= t.y; acc.z ^= t.z; acc.w ^= t.w;
        }
    }
    if (acc.x == 0x12345678u && acc.y == 77u) sink[tid] = 1.f;
}
__global__ __launch_bounds__(1024) void load32_k(const float *in, float *sink, int rows_per_wg)
{
    const int tid = threadIdx.x;
    unsigned acc = 0;
    __amdgpu_buffer_rsrc_t r = rsrc(in, 32768 * 4);
    for (int i = 0; i < rows_per_wg; ++i) {
#pragma unroll
        for (int q = 0; q < 32; ++q) {
            unsigned t = __builtin_amdgcn_raw_buffer_load_b32(r, tid * 4, q * 4096, 0);
            acc ^= t;
        }
        asm volatile("" ::: "memory");
    }
    if (acc == 0x12345678u) sink[tid] = 1.f;
}
int main()
{
    const int total_rows = 16384;
    float *buf; float *in; float *sink;
    hipMalloc(&buf, (size_t)total_rows * 32768 * 4);
    hipMalloc(&in, ((size_t)total_rows * 16384 + 65536) * 4);
    hipMalloc(&sink, 4096);
    hipMemset(in, 0, ((size_t)total_rows * 16384 + 65536) * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    int grids[] = {8, 32, 64, 128, 256, 512};
    for (int g : grids) {
        const int rpw = 64;
        for (int pass = 0; pass < 2; ++pass) {
            hipEventRecord(e0);
            hipLaunchKernelGGL(store_k, dim3(g), dim3(1024), 0, 0, buf, rpw, total_rows);
            hipEventRecord(e1); hipEventSynchronize(e1);
        }
        float ms; hipEventElapsedTime(&ms, e0, e1);
        double bytes = (double)g * rpw * 131072.0;
        printf("store  grid %4d: %.3f ms  %.2f TB/s total  %.1f GB/s per WG  (%.1f B/clk/WG @2.1GHz)\n", g, ms,
               bytes / ms / 1e9, bytes / ms / 1e6 / g, bytes / ms / 1e6 / g / 2.1);
    }
    for (int hop : {16384, 65536}) for (int g : grids) {
        const int rpw = 64;
        for (int pass = 0; pass < 2; ++pass) {
            hipEventRecord(e0);
            hipLaunchKernelGGL(load_k, dim3(g), dim3(1024), 0, 0, in, sink, rpw, total_rows / 4, hop);
            hipEventRecord(e1); hipEventSynchronize(e1);
        }
        float ms; hipEventElapsedTime(&ms, e0, e1);
        double bytes = (double)g * rpw * 262144.0;
        printf("load hop=%5d floats grid %4d: %.3f ms  %.2f TB/s total  %.1f GB/s per WG  (%.1f B/clk/WG)\n", hop, g, ms,
               bytes / ms / 1e9, bytes / ms / 1e6 / g, bytes / ms / 1e6 / g / 2.1);
    }
    // L2-resident: every workgroup re-reads the same 256 KiB (hop 0), and a 128 KiB table as b32 loads (the window shape)
    for (int g : grids) {
        const int rpw = 64;
        for (int pass = 0; pass < 2; ++pass) {
            hipEventRecord(e0);
            hipLaunchKernelGGL(load_k, dim3(g), dim3(1024), 0, 0, in, sink, rpw, 1, 0);
            hipEventRecord(e1); hipEventSynchronize(e1);
        }
        float ms; hipEventElapsedTime(&ms, e0, e1);
        double bytes = (double)g * rpw * 262144.0;
        printf("load same 256 KiB (L2 hits) grid %4d: %.3f ms  %.2f TB/s total  %.1f GB/s per WG  (%.1f B/clk/WG)\n", g, ms,
               bytes / ms / 1e9, bytes / ms / 1e6 / g, bytes / ms / 1e6 / g / 2.1);
    }
    for (int g : grids) {
        const int rpw = 64;
        for (int pass = 0; pass < 2; ++pass) {
            hipEventRecord(e0);
            hipLaunchKernelGGL(load32_k, dim3(g), dim3(1024), 0, 0, in, sink, rpw);
            hipEventRecord(e1); hipEventSynchronize(e1);
        }
        float ms; hipEventElapsedTime(&ms, e0, e1);
        double bytes = (double)g * rpw * 131072.0;
        printf("load same 128 KiB as b32 (window shape) grid %4d: %.3f ms  %.2f TB/s total  %.1f GB/s per WG  (%.1f B/clk/WG)\n", g, ms,
               bytes / ms / 1e9, bytes / ms / 1e6 / g, bytes / ms / 1e6 / g / 2.1);
    }
    return 0;
}
