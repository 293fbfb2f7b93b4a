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
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20):
    run()
e1.record()
torch.cuda.synchronize()
kernel_us = e0.elapsed_time(e1) * 1000.0 / 20
buf = np.zeros(4096 * 16, dtype=np.uint64)
lib.ro_stft_debug_stamps(st._h, buf.ctypes.data_as(C.c_void_p), buf.size)
a = buf.reshape(256, 16, 16).astype(np.float64)          # [workgroup][wave][stamp]
names = ["window (+sample / window wait)", "pass 0 levels 0-3 + old image out", "barrier a (image free)",
         "pass 0 last level + x writes", "exchange 1 rest (barriers b, c, d)", "pass 1 levels 0-3 (+touch, tw2 loads)",
         "pass 1 last level + x writes", "exchange 2 (no barrier)", "pass 2 levels 0-3",
         "pass 2 last level + mags + loads", "wait + barrier e (image complete)", "scan / tile + late loads"]
rows_wg = a[:, 0, 15].mean()
per = a[:, :, :12].sum(0) / a[:, :, 15].sum(0)[:, None]          # [wave][stamp] ticks per row
tot = per.sum(1)
print("scan fused: %s; workgroups 256, rows/wg/launch %.1f, ticks/row %.0f; launch %.1f us -> %.2f us/row, clock ~ %.2f GHz" % (
    scan, rows_wg, tot.mean(), kernel_us, kernel_us / rows_wg, tot.mean() * rows_wg / kernel_us / 1000.0))
wg_tot = a[:, 0, :12].sum(1)                                  # wave 0 of each workgroup: its ticks from first row to last
print("workgroup lifetime (wave 0, ticks): mean %.0f min %.0f max %.0f -> mean / max = %.3f of the CU time is used; by XCD (workgroup %% 8): %s"
      % (wg_tot.mean(), wg_tot.min(), wg_tot.max(), wg_tot.mean() / wg_tot.max(),
         " ".join("%.0f" % wg_tot[x::8].mean() for x in range(8))))
u = buf.reshape(256, 16, 16)
start, end = u[:, 0, 13].astype(np.float64), u[:, 0, 14].astype(np.float64)
t0 = start.min()
print("on the chip-wide 100 MHz counter: workgroups START over %.1f us, END over %.1f us, first start to last end %.1f us (the launch by events: %.1f us);"
      " mean lifetime %.1f us" % ((start.max() - t0) / 100, (end.max() - end.min()) / 100, (end.max() - t0) / 100, kernel_us, (end - start).mean() / 100))
print("by XCD (workgroup %% 8): end after the first start, us: %s;  lifetime us: %s;  ticks/us: %s" % (
    " ".join("%.0f" % ((end[x::8].max() - t0) / 100) for x in range(8)),
    " ".join("%.0f" % ((end[x::8] - start[x::8]).mean() / 100) for x in range(8)),
    " ".join("%.0f" % (wg_tot[x::8] / ((end[x::8] - start[x::8]) / 100)).mean() for x in range(8))))
print("per wave, ticks per row: e->a work = scan/late loads + window + pass 0 head; a wait; a->d; d->e work; e wait")
for w in range(16):
    ea = per[w, 11] + per[w, 0] + per[w, 1]
    print("  wave %2d  e->a %6.0f (scan/late %5.0f window %5.0f head %5.0f)  wait a %5.0f  a->d %5.0f  d->e %6.0f  wait e %5.0f" % (
        w, ea, per[w, 11], per[w, 0], per[w, 1], per[w, 2], per[w, 3] + per[w, 4], per[w, 5:10].sum(), per[w, 10]))
if "--brief" not in sys.argv:
    for w in (0, 1, 15):
        print("wave %d" % w)
        for n, t in zip(names, per[w]):
            print("  %-42s %8.0f ticks  %5.1f %%" % (n, t, 100 * t / tot[w]))
