#!/usr/bin/env python3
"""GPU box: run a -DRO_STAMPS32K=1 build (RO_STFT_LIB) of stft32k_kernel on the C3 shape and print the share of each
phase of the row loop (s_memtime ticks of wave 0 and of wave 15 of every workgroup, averaged)."""
import ctypes as C, importlib, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
ro = importlib.import_module("radio-observer_amd")
lib = ro.library()
lib.ro_stft_debug_stamps.restype = C.c_int
lib.ro_stft_debug_stamps.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
bins, overlap, R = 32768, 24576, 16384
scan = "--scan" in sys.argv
hop = bins - overlap
samples = bins + hop * (R - 1)
iq = torch.randn((samples, 2), device="cuda", dtype=torch.float32)
rows = torch.empty((R, bins), device="cuda", dtype=torch.float32)
recs = torch.zeros((R, 3), device="cuda", dtype=torch.float32) if scan else None
st = ro.Stft(bins=bins, overlap=overlap, bands=bench.make_bands(ro) if scan else None)
run = lambda: st.run_resident(iq, ro.RO_IQ_F32, samples, 0, R, rows, d_records=recs, stream=torch.cuda.current_stream().cuda_stream)
for _ in range(20):
    run()
torch.cuda.synchronize()
lib.ro_stft_debug_stamps(st._h, None, 0)          # allocate; from now on the kernel records
run()
torch.cuda.synchronize()
buf = np.zeros(4096 * 16, dtype=np.uint64)
lib.ro_stft_debug_stamps(st._h, buf.ctypes.data_as(C.c_void_p), buf.size)
a = buf.reshape(-1, 16)[:512].astype(np.float64)
names = ["window (+sample / window wait)", "pass 0 levels 0-3 + old image out", "barrier a (image free)",
         "pass 0 last level + x writes", "exchange 1 rest (barriers b, c, d)", "pass 1 levels 0-3 (+touch, tw2 loads)",
         "pass 1 last level + x writes", "exchange 2 (no barrier)", "pass 2 levels 0-3",
         "pass 2 last level + mags + loads", "wait + barrier e (image complete)", "scan / tile + late loads"]
for which, label in ((0, "wave 0"), (1, "wave 15")):
    w = a[which::2]
    w = w[w[:, 15] > 0]
    per_row = w[:, :12].sum(0) / w[:, 15].sum()
    tot = per_row.sum()
    print("%s (scan fused: %s): workgroups %d, rows/wg %.1f, ticks/row %.0f" % (label, scan, len(w), w[:, 15].mean(), tot))
    for n, t in zip(names, per_row):
        print("  %-42s %8.0f ticks  %5.1f %%" % (n, t, 100 * t / tot))
