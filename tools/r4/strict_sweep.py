#!/usr/bin/env python3
"""GPU box: the RO_PRECISION_F64 path at the C3 shape (2048 rows per step, what bench.py's strict_precision entry
times) for one scratch-chunk size -- does a chunk whose complex-double scratch stays inside the 256 MiB Infinity
Cache keep the trip between the two pass kernels off HBM?  Needs a -DRO_DIAG=1 build (RO_STFT_LIB=...), which reads
RO_F64_SCRATCH_MB from the environment.  Prints rows/s, ms per step and a hash of the rows (bit-identical
whatever the chunk).  usage: RO_F64_SCRATCH_MB=64 RO_STFT_LIB=build/ab/libro_stft_diag.so strict_sweep.py [steps]"""
import hashlib, importlib, os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
ro = importlib.import_module("radio-observer_amd")
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
bins, overlap, R = 32768, 24576, 2048
samples = bins + (bins - overlap) * (R - 1)
iq = bench.synth_iq(torch, samples, 0xC3, "cuda:0")
rows = torch.empty((R, bins), dtype=torch.float32, device="cuda")
recs = torch.zeros((R, 3), dtype=torch.float32, device="cuda")
s = torch.cuda.current_stream().cuda_stream
with ro.Stft(bins=bins, overlap=overlap, bands=bench.make_bands(ro), precision=ro.RO_PRECISION_F64) as st:
    for _ in range(5):
        st.run_resident(iq, ro.RO_IQ_F32, samples, 0, R, rows, d_records=recs, stream=s)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(steps):
        st.run_resident(iq, ro.RO_IQ_F32, samples, 0, R, rows, d_records=recs, stream=s)
    e1.record()
    torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / steps
h = hashlib.sha256(rows.cpu().numpy().tobytes()).hexdigest()[:16]
print("RO_F64_SCRATCH_MB=%s rows/step %d  %.4f ms/step  %.4g rows/s  rows_hash %s"
      % (os.environ.get("RO_F64_SCRATCH_MB", "default"), R, ms, R / (ms * 1e-3), h), flush=True)
