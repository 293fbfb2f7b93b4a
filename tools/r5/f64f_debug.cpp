// GPU box: the fused FP64 kernel alone, with short spin limits and a trace of every workgroup, against a host FP64 DFT
// of a few bins.  Build: hipcc -O3 --offload-arch=gfx950 -std=c++17 -fno-slp-vectorize -DRO_F64F_DEBUG=1
//   -DRO_F64F_SPIN_LIMIT=4096 -I radio-observer_amd/csrc tools/r5/f64f_debug.cpp radio-observer_amd/csrc/ro_f64fused.hip -o build/f64f_debug
// usage: f64f_debug [rows] [ring_rows] [wgs_per_cu] [launches]
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "ro_kernels.h"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 2; } } while (0)

int main(int argc, char **argv)
{
    const int rows = argc > 1 ? atoi(argv[1]) : 8, ring_rows = argc > 2 ? atoi(argv[2]) : 6, wgs = argc > 3 ? atoi(argv[3]) : 2;
    const int n = 32768, hop = 8192;
    const size_t samples = (size_t)n + (size_t)hop * (rows - 1);
    std::vector<float> iq(samples * 2), win(n, 1.0f);
    unsigned st = 12345;
    for (auto &x : iq) { st = st * 1664525u + 1013904223u; x = (float)((int)(st >> 8) % 2001 - 1000) / 1000.0f; }
    std::vector<double2> tw(n);
    for (int m = 0; m < n; ++m) {
        const long double ph = -2.0L * 3.14159265358979323846264338327950288L * m / n;
        tw[m] = double2{(double)cosl(ph), (double)sinl(ph)};
    }
    float *d_iq, *d_win, *d_rows; double2 *d_tw, *d_ring; unsigned *d_ctl;
    CK(hipMalloc(&d_iq, iq.size() * 4)); CK(hipMalloc(&d_win, n * 4)); CK(hipMalloc(&d_tw, n * 16));
    CK(hipMalloc(&d_rows, (size_t)rows * n * 4)); CK(hipMalloc(&d_ring, (size_t)8 * ring_rows * n * 16));
    CK(hipMalloc(&d_ctl, ro::f64_fused_ctl_bytes()));
    CK(hipMemcpy(d_iq, iq.data(), iq.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_win, win.data(), n * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_tw, tw.data(), n * 16, hipMemcpyHostToDevice));
    CK(hipMemset(d_rows, 0xff, (size_t)rows * n * 4));
    ro::BigArgsD b{};
    b.iq = d_iq; b.window = d_win; b.tw = d_tw; b.first_row = 0; b.rows = rows; b.row_stride = n; b.hop = hop; b.n = n; b.gain = 0.0;
    b.rows_out = d_rows;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int iters = argc > 4 ? atoi(argv[4]) : 1;
    float ms = 0;
    for (int it = 0; it < iters; ++it) {
        CK(hipEventRecord(e0, 0));
        CK(ro::launch_f64_fused(RO_FMT_F32, b, d_ring, d_ctl, ring_rows, wgs, 0));
        CK(hipEventRecord(e1, 0));
        CK(hipDeviceSynchronize());
        CK(hipEventElapsedTime(&ms, e0, e1));
        if (iters > 1) { printf("launch %d: %.3f ms\n", it, ms); fflush(stdout); }
    }
    std::vector<unsigned> ctl(ro::f64_fused_ctl_bytes() / 4);
    CK(hipMemcpy(ctl.data(), d_ctl, ctl.size() * 4, hipMemcpyDeviceToHost));
    printf("rows %d ring %d wgs/cu %d: %.3f ms, next_row %u, error %u\n", rows, ring_rows, wgs, ms, ctl[0], ctl[1]);
    for (int x = 0; x < 8; ++x) {
        const unsigned *xc = &ctl[32 + x * 1024];
        printf("xcd %d: tickets %u  a_cnt", x, xc[0]);
        for (int s = 0; s < ring_rows && s < 8; ++s) printf(" %u", xc[32 + 2 * s]);
        printf("  b_cnt");
        for (int s = 0; s < ring_rows && s < 8; ++s) printf(" %u", xc[33 + 2 * s]);
        printf("  map");
        for (int m = 0; m < 4; ++m) printf(" (%u,%u)", xc[160 + 2 * m + 1], xc[160 + 2 * m]);
        printf("\n");
    }
#ifndef RO_F64F_DEBUG
    std::vector<unsigned> pad(16 * 1024, 0);
    const unsigned *dbg = pad.data();
#else
    const unsigned *dbg = &ctl[32 + 8 * 1024];
#endif
    int shown = 0, hist[8] = {0};
    double seg[6] = {0, 0, 0, 0, 0, 0}, tickets = 0;
    for (int w = 0; w < 1024 && dbg[w * 16] != 0; ++w) {
        hist[dbg[w * 16 + 3] & 7]++;
        tickets += dbg[w * 16 + 1];
        for (int k = 0; k < 6; ++k) seg[k] += 64.0 * dbg[w * 16 + 8 + k];
        if (dbg[w * 16 + 3] != 5 && shown < 24) {
            printf("  wg %d xcc %u drawn %u last tk %u state %u row %u want %u cnt_word %u\n", w, dbg[w * 16] - 100, dbg[w * 16 + 1],
                   dbg[w * 16 + 2], dbg[w * 16 + 3], dbg[w * 16 + 4], dbg[w * 16 + 5], dbg[w * 16 + 6]);
            ++shown;
        }
    }
    if (tickets > 0)
        printf("thread-0 cycles per ticket (last launch): ticket atomic %.0f, row map %.0f, dependency wait %.0f | per tile: A work+drain %.0f, "
               "B work %.0f, null %.0f  (sums over %g tickets: %.3g %.3g %.3g %.3g %.3g %.3g)\n",
               seg[0] / tickets, seg[1] / tickets, seg[2] / tickets, seg[3] / (tickets / 2), seg[4] / (tickets / 2), seg[5] / tickets,
               tickets, seg[0], seg[1], seg[2], seg[3], seg[4], seg[5]);
    printf("workgroup end states: exited %d, gave up %d, other %d %d %d %d\n", hist[5], hist[6], hist[1], hist[2], hist[3], hist[4]);
    // spot check: row r, a few bins against a direct double DFT (window of ones, fft-shifted magnitude)
    std::vector<float> out((size_t)rows * n);
    CK(hipMemcpy(out.data(), d_rows, out.size() * 4, hipMemcpyDeviceToHost));
    double worst = 0;
    for (int r : {0, rows / 2, rows - 1})
        for (int k : {0, 1, 1234, 16384, 32767}) {
            long double re = 0, im = 0;
            for (int i = 0; i < n; ++i) {
                const long double ph = -2.0L * 3.14159265358979323846264338327950288L * (long double)(((int64_t)i * k) % n) / n;
                const long double xr = iq[2 * ((size_t)r * hop + i)], xi = iq[2 * ((size_t)r * hop + i) + 1];
                re += xr * cosl(ph) - xi * sinl(ph);
                im += xr * sinl(ph) + xi * cosl(ph);
            }
            const double want = (double)sqrtl(re * re + im * im), got = out[(size_t)r * n + ((k + n / 2) & (n - 1))];
            worst = fmax(worst, fabs(got - want) / fmax(want, 1e-30));
        }
    printf("spot check (15 bins): worst relative error %.3g\n", worst);
    return 0;
}
