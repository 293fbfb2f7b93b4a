#!/usr/bin/env python3
"""GPU box: run a -DRO_DIAG=1 -DRO_F64R_STAMPS=1 build (RO_STFT_LIB) of f64r_kernel on one shape and print the share of
each phase of the sub-row loop (s_memtime ticks, all waves of all workgroups).  usage: f64r_stamps.py BINS OVERLAP ROWS"""
import ctypes as C, importlib, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
ro = importlib.import_module("radio-observer_amd")
lib = ro.library()
lib.ro_stft_debug_stamps.restype = C.c_int
lib.ro_stft_debug_stamps.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
bins, overlap, R = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
hop = bins - overlap
samples = bins + hop * (R - 1)
iq = torch.randn((samples, 2), device="cuda", dtype=torch.float32)
rows = torch.empty((R, bins), device="cuda", dtype=torch.float32)
st = ro.Stft(bins=bins, overlap=overlap, precision=ro.RO_PRECISION_F64)
run = lambda: st.run_resident(iq, ro.RO_IQ_F32, samples, 0, R, rows, stream=torch.cuda.current_stream().cuda_stream)
for _ in range(10):
    run()
torch.cuda.synchronize()
lib.ro_stft_debug_stamps(st._h, None, 0)          # allocate; from now on the kernel records
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
run()
e1.record()
torch.cuda.synchronize()
kernel_us = e0.elapsed_time(e1) * 1000.0
buf = np.zeros(4096 * 16, dtype=np.uint64)
lib.ro_stft_debug_stamps(st._h, buf.ctypes.data_as(C.c_void_p), buf.size)
D = {4096: 1, 8192: 1, 16384: 1, 32768: 2, 65536: 4}[bins]
M = bins // D
waves = M // 16 // 64
a = buf.reshape(-1, waves, 16).astype(np.float64)        # [workgroup][wave][stamp]
a = a[a[:, 0, 15] > 0]
names = ["image out of LDS", "barrier a (image read)", "row stores", "fold (+ wait for the samples)",
         "pass 0 + x1 write re", "barrier b + x1 read re", "barrier c + x1 write im", "barrier d + x1 read im + pass 1",
         "exchange 2", "pass 2 (+ its table loads)", "lane swaps (exchange 3)", "pass 3 (+ its table loads)",
         "magnitudes + image write", "next samples requested + barrier e"]
subrows = a[:, :, 15]
per = a[:, :, :14].sum((0, 1)) / subrows.sum()
tot = per.sum()
print("bins %d overlap %d rows %d: %d workgroups x %d waves, %.1f sub-rows per workgroup; launch %.1f us; %.0f ticks per sub-row"
      " and wave (100 MHz x ticks: s_memtime counts shader cycles)" % (bins, overlap, R, a.shape[0], waves, subrows[:, 0].mean(), kernel_us, tot))
for k, n in enumerate(names):
    if per[k] > 0:
        print("  %-40s %8.0f  %5.1f %%" % (n, per[k], 100 * per[k] / tot))
# oldest (wave 0) against youngest wave
for w in (0, waves - 1):
    pw = a[:, w, :14].sum(0) / a[:, w, 15].sum()
    print("  wave %2d: %s" % (w, " ".join("%5.0f" % x for x in pw)))
