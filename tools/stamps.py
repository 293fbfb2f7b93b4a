#!/usr/bin/env python3
"""GPU box: run a -DRO_STAMPS=1 build (RO_STFT_LIB) on the C3 shape (or: stamps.py BINS OVERLAP ROWS) and print the share of each
phase of the row loop (s_memtime ticks of wave 0 per workgroup, averaged)."""
import ctypes as C, importlib, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ro = importlib.import_module("radio-observer_amd")
lib = ro.library()
lib.ro_stft_debug_stamps.restype = C.c_int
lib.ro_stft_debug_stamps.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
bins, overlap, R = (int(x) for x in (sys.argv[1:4] if len(sys.argv) > 3 else (32768, 24576, 16384)))
hop = bins - overlap
samples = bins + hop * (R - 1)
iq = torch.randn((samples, 2), device="cuda", dtype=torch.float32)
rows = torch.empty((R, bins), device="cuda", dtype=torch.float32)
st = ro.Stft(bins=bins, overlap=overlap)
for _ in range(2):
    st.run_resident(iq, ro.RO_IQ_F32, samples, 0, R, rows, stream=torch.cuda.current_stream().cuda_stream)
torch.cuda.synchronize()
lib.ro_stft_debug_stamps(st._h, None, 0)          # allocate; from now on the kernel records
# the launch that is read back comes last of a run long enough for the power governor to settle (the stamps of a launch
# overwrite those of the one before); its duration from events, so ticks / duration = the clock it ran at
K = int(os.environ.get("STAMP_LAUNCHES", "400"))
ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
for i in range(K):
    if i == K - 1:
        ev[0].record()
    st.run_resident(iq, ro.RO_IQ_F32, samples, 0, R, rows, stream=torch.cuda.current_stream().cuda_stream)
ev[1].record()
torch.cuda.synchronize()
launch_us = ev[0].elapsed_time(ev[1]) * 1e3
buf = np.zeros(4096 * 16, dtype=np.uint64)
lib.ro_stft_debug_stamps(st._h, buf.ctypes.data_as(C.c_void_p), buf.size)
a = buf.reshape(-1, 16)
a = a[a[:, 9] > 0].astype(np.float64)
idx = [0, 1, 2, 3, 4, 5, 6, 10, 7, 11, 12, 8]
names = ["window mult (+sample wait)", "-", "butterflies0 + tw prefetch", "exchange 1", "tw + butterflies 1",
         "exchange 2", "tw + butterflies 2", "magnitudes -> LDS", "next-row loads issue", "barrier 1",
         "LDS read-back + stores issue", "barrier 2"]
per_row = a[:, idx].sum(0) / a[:, 9].sum()
tot = per_row.sum()
print("workgroups %d, rows/wg %.1f, ticks/row %.0f; the launch took %.1f us = %.0f ticks/us per workgroup (s_memtime against the launch's duration)"
      % (len(a), a[:, 9].mean(), tot, launch_us, tot * a[:, 9].mean() / launch_us))
for n, t in zip(names, per_row):
    print("  %-30s %8.0f ticks  %5.1f %%" % (n, t, 100 * t / tot))
