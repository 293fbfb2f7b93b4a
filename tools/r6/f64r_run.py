#!/usr/bin/env python3
"""GPU box: LAUNCHES launches of the RO_PRECISION_F64 path on ROWS rows of one shape (synthetic noise + carrier), nothing
else on the device: for a kernel trace or a counter pass of that path alone.  usage: f64r_run.py BINS OVERLAP ROWS [LAUNCHES]"""
import importlib, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
ro = importlib.import_module("radio-observer_amd")
bins, overlap, R = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
launches = int(sys.argv[4]) if len(sys.argv) > 4 else 6
samples = bins + (bins - overlap) * (R - 1)
iq = bench.synth_iq(torch, samples, 0xC3, "cuda:0")
rows = torch.empty((R, bins), dtype=torch.float32, device="cuda")
with ro.Stft(bins=bins, overlap=overlap, precision=ro.RO_PRECISION_F64) as st:
    for _ in range(launches):
        st.run_resident(iq, ro.RO_IQ_F32, samples, 0, R, rows, stream=torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
print("ok")
