#!/usr/bin/env python3
"""GPU box: 20 launches of the RO_PRECISION_F64 path on 2048 rows of the C3 shape (what bench.py's strict_precision
entry times), for a kernel trace of that path alone."""
import importlib, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
ro = importlib.import_module("radio-observer_amd")
bins, overlap, R = 32768, 24576, 2048
samples = bins + (bins - overlap) * (R - 1)
iq = bench.synth_iq(torch, samples, 0xC3, "cuda:0")
rows = torch.empty((R, bins), dtype=torch.float32, device="cuda")
recs = torch.zeros((R, 3), dtype=torch.float32, device="cuda")
with ro.Stft(bins=bins, overlap=overlap, bands=bench.make_bands(ro), precision=ro.RO_PRECISION_F64) as st:
    for _ in range(20):
        st.run_resident(iq, ro.RO_IQ_F32, samples, 0, R, rows, d_records=recs, stream=torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
print("ok")
