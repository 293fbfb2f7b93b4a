#!/usr/bin/env python3
"""GPU box: the C3 kernel on fewer CUs (spare_cus_per_xcd = 0 ... 24 of each XCD's 32 left idle): rows/s, rows/s per
active CU, package power and sclk (sysfs hwmon files) in the middle of a long run of launches.  If the rate were bound by what
a CU can do, rows/s per CU would not depend on how many of them work; under a package power cap it rises as CUs are
taken away (the clock rises)."""
import importlib, os, sys, threading, time
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
ro = importlib.import_module("radio-observer_amd")
bins, overlap, R = 32768, 24576, 16384
hop = bins - overlap
samples = bins + hop * (R - 1)
iq = torch.randn((samples, 2), device="cuda", dtype=torch.float32)
rows = torch.empty((R, bins), device="cuda", dtype=torch.float32)
s = torch.cuda.current_stream().cuda_stream


def smi():
    """package power and sclk of this device from the amdgpu hwmon files in sysfs -- plain file reads (bench.py's
    ClockPowerSampler): the first form of this script ran rocm-smi as a child of a process that holds the GPU, which
    bench.py and tests/test_bench_cpu.py forbid (a fork + exec out of such a process is a hazard on this pool)"""
    import bench
    smp = bench.ClockPowerSampler(torch, 0)
    if not smp.files:
        return float("nan"), -1
    mhz = smp._read(smp.files[0]) / 1e6
    w = smp._read(smp.files[1]) / 1e6 if smp.files[1] else float("nan")
    return w, int(mhz)


print("spare CUs per XCD | active CUs | ms per launch | rows/s | rows/s per active CU | package W | sclk MHz")
for spare in (0, 4, 8, 12, 16):
    with ro.Stft(bins=bins, overlap=overlap, spare_cus_per_xcd=spare) as st:
        for _ in range(30):
            st.run_resident(iq, ro.RO_IQ_F32, samples, 0, R, rows, stream=s)
        torch.cuda.synchronize()
        seen = []
        th = threading.Thread(target=lambda: (time.sleep(1.5), seen.append(smi())))
        th.start()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        n = int(3000 * (32 - spare) / 32) + 200
        e0.record()
        for _ in range(n):
            st.run_resident(iq, ro.RO_IQ_F32, samples, 0, R, rows, stream=s)
        e1.record()
        torch.cuda.synchronize()
        th.join()
        ms = e0.elapsed_time(e1) / n
        act = 8 * (32 - spare)
        w, clk = seen[0] if seen else (float("nan"), -1)
        print("%17d | %10d | %13.4f | %.4g | %20.4g | %9.0f | %8d" % (spare, act, ms, R / ms * 1e3, R / ms * 1e3 / act, w, clk), flush=True)
