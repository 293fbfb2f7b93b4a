#!/usr/bin/env python3
"""GPU box: shader clock and package power (amdgpu hwmon, bench.py's sampler) while the register-resident FP64 kernel runs
back to back for a few seconds at one shape -- is it the package's power cap that sets its clock, as under the float32
headline kernel?   python tools/r6/f64r_power.py [BINS OVERLAP ROWS [SECONDS]] | --f32"""
import importlib
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402

ro = importlib.import_module("radio-observer_amd")


def run(bins, overlap, R, seconds, precision):
    hop = bins - overlap
    T = bins + (R - 1) * hop
    d_iq = torch.randn((T, 2), dtype=torch.float32, device="cuda")
    d_rows = torch.empty((R, bins), dtype=torch.float32, device="cuda")
    s = torch.cuda.current_stream()
    sampler = bench.ClockPowerSampler(torch, 0)
    with ro.Stft(bins=bins, overlap=overlap, precision=precision) as st:
        for _ in range(5):
            st.run_resident(d_iq, ro.RO_IQ_F32, T, 0, R, d_rows, stream=s.cuda_stream)
        torch.cuda.synchronize()
        sampler.start()
        t0 = time.perf_counter()
        n = 0
        while time.perf_counter() - t0 < seconds:
            for _ in range(20):
                st.run_resident(d_iq, ro.RO_IQ_F32, T, 0, R, d_rows, stream=s.cuda_stream)
            torch.cuda.synchronize()
            n += 20
        t1 = time.perf_counter()
        sampler.stop()
    late = [r for r in sampler.samples if r[0] > t0 + 0.4 * (t1 - t0)]
    mhz = np.array([r[1] for r in late])
    w = np.array([r[2] for r in late])
    alg = hop * 8 + bins * 4
    print("%s bins %6d overlap %6d rows %6d: %5d launches in %.2f s = %.3e rows/s = %.3f of 8 TB/s; last 60 %%: sclk %.0f MHz (min %.0f max %.0f), "
          "package %.0f W (min %.0f max %.0f), %d samples"
          % ("FP64" if precision else "FP32", bins, overlap, R, n, t1 - t0, n * R / (t1 - t0), alg * n * R / (t1 - t0) / 8e12,
             np.nanmean(mhz), np.nanmin(mhz), np.nanmax(mhz), np.nanmean(w), np.nanmin(w), np.nanmax(w), len(late)), flush=True)


def main():
    if "--f32" in sys.argv:                                    # the float32 kernels of the station configs' shapes instead
        for b, o, r in ((65536, 49152, 8192), (524288, 262144, 512), (4096, 2048, 65536), (16384, 12288, 16384)):
            run(b, o, r, 3.0, ro.RO_PRECISION_F32)
        return
    secs = float(sys.argv[4]) if len(sys.argv) > 4 else 3.0
    if len(sys.argv) > 3:
        shapes = [(int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]))]
    else:
        shapes = [(32768, 24576, 16384), (4096, 2048, 65536), (1024, 512, 262144), (65536, 49152, 8192)]
    for b, o, r in shapes:
        run(b, o, r, secs, ro.RO_PRECISION_F64)
    run(32768, 24576, 16384, secs, ro.RO_PRECISION_F32)        # the float32 headline kernel beside them


if __name__ == "__main__":
    main()
