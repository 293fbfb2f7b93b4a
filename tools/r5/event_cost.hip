// GPU box: what one HIP runtime call of the streaming path costs the calling thread
// (hipcc --offload-arch=gfx950 -O2 -o tools/r5/event_cost tools/r5/event_cost.hip on the CPU box; the binary travels with gpurun).
// hipEventQuery on an event that is not ready / ready, hipEventRecord, hipStreamWaitEvent, a 4-byte hipMemcpyAsync,
// hipStreamWriteValue32 into pinned memory, hipGraphLaunch of an empty-kernel graph.  Diagnostic.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
__global__ void spin(long long ticks) { const long long t0 = wall_clock64(); while (wall_clock64() - t0 < ticks) {} }
__global__ void nop() {}
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { std::printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
int main()
{
    hipStream_t s, s2;
    CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
    hipEvent_t ev, ev2;
    CK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    CK(hipEventCreate(&ev2));
    unsigned *flag = nullptr, *dflag = nullptr;
    CK(hipHostMalloc((void **)&flag, 64, hipHostMallocDefault));
    CK(hipMalloc((void **)&dflag, 64));
    *flag = 0;
    const int N = 20000;
    // not ready: a kernel that spins 100 ms (100 MHz counter) in front of the event
    hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, s, 10000000LL);
    CK(hipEventRecord(ev, s));
    double t = now();
    int notready = 0;
    for (int i = 0; i < N; ++i) notready += hipEventQuery(ev) == hipErrorNotReady;
    std::printf("hipEventQuery, not ready (%d of %d): %.3f us\n", notready, N, (now() - t) / N * 1e6);
    CK(hipStreamSynchronize(s));
    t = now();
    for (int i = 0; i < N; ++i) (void)hipEventQuery(ev);
    std::printf("hipEventQuery, ready: %.3f us\n", (now() - t) / N * 1e6);
    t = now();
    for (int i = 0; i < N; ++i) (void)hipEventRecord(ev, s);
    std::printf("hipEventRecord (no timing): %.3f us\n", (now() - t) / N * 1e6);
    CK(hipStreamSynchronize(s));
    t = now();
    for (int i = 0; i < N; ++i) (void)hipEventRecord(ev2, s);
    std::printf("hipEventRecord (timing): %.3f us\n", (now() - t) / N * 1e6);
    CK(hipStreamSynchronize(s));
    t = now();
    for (int i = 0; i < N; ++i) (void)hipStreamWaitEvent(s2, ev, 0);
    std::printf("hipStreamWaitEvent: %.3f us\n", (now() - t) / N * 1e6);
    CK(hipStreamSynchronize(s2));
    t = now();
    for (int i = 0; i < N; ++i) (void)hipMemcpyAsync(flag, dflag, 4, hipMemcpyDeviceToHost, s);
    std::printf("hipMemcpyAsync 4 bytes D2H: %.3f us\n", (now() - t) / N * 1e6);
    CK(hipStreamSynchronize(s));
    t = now();
    hipError_t we = hipSuccess;
    for (int i = 0; i < N && we == hipSuccess; ++i) we = hipStreamWriteValue32(s, flag, (unsigned)i + 1, 0);
    std::printf("hipStreamWriteValue32 into pinned memory: %.3f us (%s)\n", (now() - t) / N * 1e6, hipGetErrorString(we));
    CK(hipStreamSynchronize(s));
    std::printf("  flag after the stream: %u\n", *flag);
    // does the value arrive behind the work in front of it?  spin 20 ms, then the write; poll
    hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, s, 2000000LL);
    we = hipStreamWriteValue32(s, flag, 0xABCD, 0);
    t = now();
    while (*(volatile unsigned *)flag != 0xABCD && now() - t < 1.0) {}
    std::printf("  value visible %.1f ms after its call (a 20 ms kernel in front of it)\n", (now() - t) * 1e3);
    // ---- the same calls in bursts of 16 behind a synchronised stream: what the CALL costs the host when the queue is empty
    // (the loops above fill the queue and then run at the device's rate)
    {
        hipGraph_t g0;
        hipGraphExec_t ge0;
        CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
        CK(hipMemcpyAsync(dflag, flag, 4, hipMemcpyHostToDevice, s));
        hipLaunchKernelGGL(nop, dim3(256), dim3(1024), 0, s);
        hipLaunchKernelGGL(nop, dim3(1), dim3(64), 0, s);
        CK(hipStreamEndCapture(s, &g0));
        CK(hipGraphInstantiate(&ge0, g0, nullptr, nullptr, 0));
        float *h2 = nullptr, *d2 = nullptr;
        CK(hipHostMalloc((void **)&h2, 8 << 20, hipHostMallocDefault));
        CK(hipMalloc((void **)&d2, 8 << 20));
        const int B = 16, T = 200;
        const char *names[] = {"hipEventRecord (no timing)", "hipEventRecord (timing)", "hipMemcpyAsync 72 B D2H", "hipMemcpy2DAsync 6 x 128 KiB D2H",
                               "hipGraphLaunch (upload + two kernels)", "kernel launch", "hipStreamWriteValue32", "hipMemcpyAsync 590 KiB H2D"};
        for (int op = 0; op < 8; ++op) {
            double sum = 0;
            for (int tr = 0; tr < T; ++tr) {
                CK(hipStreamSynchronize(s));
                const double t0 = now();
                for (int i = 0; i < B; ++i) {
                    switch (op) {
                    case 0: (void)hipEventRecord(ev, s); break;
                    case 1: (void)hipEventRecord(ev2, s); break;
                    case 2: (void)hipMemcpyAsync(flag, dflag, 72, hipMemcpyDeviceToHost, s); break;
                    case 3: (void)hipMemcpy2DAsync(h2, 131072 + 64, d2, 131072, 131072, 6, hipMemcpyDeviceToHost, s); break;
                    case 4: (void)hipGraphLaunch(ge0, s); break;
                    case 5: hipLaunchKernelGGL(nop, dim3(256), dim3(1024), 0, s); break;
                    case 6: (void)hipStreamWriteValue32(s, flag, 7, 0); break;
                    case 7: (void)hipMemcpyAsync(d2, h2, 590 << 10, hipMemcpyHostToDevice, s); break;
                    }
                }
                sum += now() - t0;
            }
            std::printf("burst of %d, %-40s %.3f us per call\n", B, names[op], sum / T / B * 1e6);
        }
        CK(hipStreamSynchronize(s));
    }
    hipGraph_t g;
    hipGraphExec_t ge;
    CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
    hipLaunchKernelGGL(nop, dim3(1), dim3(64), 0, s);
    hipLaunchKernelGGL(nop, dim3(1), dim3(64), 0, s);
    CK(hipStreamEndCapture(s, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    t = now();
    for (int i = 0; i < N / 10; ++i) (void)hipGraphLaunch(ge, s);
    std::printf("hipGraphLaunch (two empty kernels): %.3f us\n", (now() - t) / (N / 10) * 1e6);
    CK(hipStreamSynchronize(s));
    return 0;
}
