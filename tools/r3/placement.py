#!/usr/bin/env python3
"""GPU box: where and when the workgroups of stft_kernel's persistent grid run (a -DRO_DIAG=1 -DRO_STAMPS=1 build via
RO_STFT_LIB): XCC / SE / CU of every workgroup from HW_REG_HW_ID, its start and end on the XCD's s_memtime counter.
Prints workgroups per CU, how long each was resident relative to the launch, and how many CUs ran their workgroups
one after the other instead of side by side.  usage: placement.py BINS OVERLAP ROWS"""
import ctypes as C, importlib, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
ro = importlib.import_module("radio-observer_amd")
lib = ro.library()
lib.ro_stft_debug_stamps.restype = C.c_int
lib.ro_stft_debug_stamps.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
bins, overlap, R = (int(x) for x in sys.argv[1:4])
hop = bins - overlap
samples = bins + hop * (R - 1)
iq = torch.randn((samples, 2), device="cuda", dtype=torch.float32)
rows = torch.empty((R, bins), device="cuda", dtype=torch.float32)
st = ro.Stft(bins=bins, overlap=overlap)
s = torch.cuda.current_stream().cuda_stream
for _ in range(2):
    st.run_resident(iq, ro.RO_IQ_F32, samples, 0, R, rows, stream=s)
torch.cuda.synchronize()
lib.ro_stft_debug_stamps(st._h, None, 0)
K = int(os.environ.get("STAMP_LAUNCHES", "200"))
ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
for i in range(K):
    if i == K - 1:
        ev[0].record()
    st.run_resident(iq, ro.RO_IQ_F32, samples, 0, R, rows, stream=s)
ev[1].record()
torch.cuda.synchronize()
us = ev[0].elapsed_time(ev[1]) * 1e3
buf = np.zeros(4096 * 16, dtype=np.uint64)
lib.ro_stft_debug_stamps(st._h, buf.ctypes.data_as(C.c_void_p), buf.size)
a = buf.reshape(-1, 16)
a = a[a[:, 9] > 0]
hw = a[:, 15] & 0xffffffff
xcc = (a[:, 15] >> 32) & 0xf
cu, sh, se = (hw >> 8) & 0xf, (hw >> 12) & 1, (hw >> 13) & 7
start, end = a[:, 13].astype(np.float64), a[:, 14].astype(np.float64)
print("%d workgroups, %.1f rows each; launch %.1f us" % (len(a), a[:, 9].mean(), us))
place = {}
for i in range(len(a)):
    place.setdefault((int(xcc[i]), int(se[i]), int(sh[i]), int(cu[i])), []).append(i)
per_cu = np.bincount([len(v) for v in place.values()])
print("CUs in use: %d; CUs with k workgroups: %s" % (len(place), {k: int(n) for k, n in enumerate(per_cu) if n}))
life = end - start
# s_memtime runs at the shader clock but its zero differs between clock domains, so times are only compared inside a
# cluster of workgroups whose starts lie within one launch length of each other
order = np.argsort(start)
clusters, cur = [], [order[0]]
for i in order[1:]:
    if start[i] - start[cur[0]] > 4 * life.max():
        clusters.append(cur); cur = []
    cur.append(i)
clusters.append(cur)
print("workgroup lifetime: mean %.0f ticks, min %.0f, max %.0f (%.0f ticks/us if the slowest spans the launch)"
      % (life.mean(), life.min(), life.max(), life.max() / us))
print("%d clock domains%s" % (len(clusters), "" if "--clusters" in sys.argv else " (--clusters lists them)"))
for c in (clusters if "--clusters" in sys.argv else []):
    c = np.array(c)
    t0, t1 = start[c].min(), end[c].max()
    span = t1 - t0
    st_sorted = np.sort(start[c] - t0)
    print("  %3d workgroups (XCC ids %s): busy %.0f ticks = %.0f ticks/us; starts spread over %.3f of it (median start %.3f); "
          "ends from %.3f to 1; mean lifetime %.3f -> %.1f %% of the CU time of the launch is used"
          % (len(c), sorted(set(xcc[c].tolist())), span, span / us, st_sorted[-1] / span, np.median(st_sorted) / span,
             (end[c].min() - t0) / span, (life[c] / span).mean(), 100 * (life[c] / span).mean()))
serial = 0
for v in place.values():
    v = sorted(v, key=lambda i: start[i])
    for p, q in zip(v, v[1:]):
        if start[q] >= end[p]:
            serial += 1
print("pairs of workgroups that shared a CU one AFTER the other: %d" % serial)
