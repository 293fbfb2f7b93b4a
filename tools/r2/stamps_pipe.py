#!/usr/bin/env python3
"""GPU box: phase shares of the PIPELINED row loop (N = 32768) from a -DRO_STAMPS=1 build (RO_STFT_LIB): s_memtime ticks of
wave 0 per workgroup, averaged.  Same instrument as tools/stamps.py (round 1), phases of the round-2 loop."""
import ctypes as C, importlib, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
ro = importlib.import_module("radio-observer_amd")
lib = ro.library()
lib.ro_stft_debug_stamps.restype = C.c_int
lib.ro_stft_debug_stamps.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
bins, overlap, R = 32768, 24576, 16384
hop = bins - overlap
samples = bins + hop * (R - 1)
iq = torch.randn((samples, 2), device="cuda", dtype=torch.float32)
rows = torch.empty((R, bins), device="cuda", dtype=torch.float32)
recs = torch.zeros((R, 3), device="cuda", dtype=torch.float32)
with_scan = "--scan" in sys.argv
bands = ro.Bands(low_noise=22528, noise_width=409, low_detect=23415, detect_width=410, avg_bins=27) if with_scan else None
st = ro.Stft(bins=bins, overlap=overlap, bands=bands)
kw = dict(d_records=recs) if with_scan else {}
for _ in range(25):
    st.run_resident(iq, ro.RO_IQ_F32, samples, 0, R, rows, stream=torch.cuda.current_stream().cuda_stream, **kw)
torch.cuda.synchronize()
lib.ro_stft_debug_stamps(st._h, None, 0)          # allocate; from now on the kernel records
st.run_resident(iq, ro.RO_IQ_F32, samples, 0, R, rows, stream=torch.cuda.current_stream().cuda_stream, **kw)
torch.cuda.synchronize()
buf = np.zeros(4096 * 16, dtype=np.uint64)
lib.ro_stft_debug_stamps(st._h, buf.ctypes.data_as(C.c_void_p), buf.size)
a = buf.reshape(-1, 16)
a = a[a[:, 9] > 0].astype(np.float64)
idx = [0, 2, 7, 10, 3, 4, 11, 12, 5, 6, 13, 14, 15, 8]
names = ["window mult (+sample/window wait)", "pass 0 levels 0-3 + prev row read-back/stores", "barrier (image free)",
         "pass 0 last level + x scatter", "exchange 1 rest (3 barriers)", "pass 1 levels 0-3 (+touch, tw loads)",
         "barrier (exch 1 gathered)", "pass 1 last level + x scatter", "exchange 2 rest (3 barriers)",
         "pass 2 levels 0-3", "barrier (exch 2 gathered)", "pass 2 last level + mags + image + loads",
         "barrier (image complete)", "scan/tile waves + late loads issue"]
per_row = a[:, idx].sum(0) / a[:, 9].sum()
tot = per_row.sum()
print("scan fused: %s; workgroups %d, rows/wg %.1f, ticks/row %.0f" % (with_scan, len(a), a[:, 9].mean(), tot))
for n, t in zip(names, per_row):
    print("  %-45s %8.0f ticks  %5.1f %%" % (n, t, 100 * t / tot))
