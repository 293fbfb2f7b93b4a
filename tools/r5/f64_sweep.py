#!/usr/bin/env python3
"""GPU box: the RO_PRECISION_F64 path at the C3 shape for one setting of its knobs.  Needs a -DRO_DIAG=1 build
(RO_STFT_LIB=build/ab/libro_stft_diag.so), which reads from the environment
    RO_F64_FUSED      1 = all four passes in one launch, the intermediate in an XCD's L2 (ro_f64fused.hip); 0 = two launches
    RO_F64_RING_ROWS  rows of 32768 bins in an XCD's ring (fused form)
    RO_F64_WGS        workgroups per CU (fused form: 1 or 2)
    RO_F64_SCRATCH_MB chunk of the two-launch form
Prints rows/s, ms per step and a hash of the rows (bit-identical whatever the form).
usage: [knobs] RO_STFT_LIB=... f64_sweep.py [steps] [rows] [bins] [overlap]"""
import hashlib, importlib, os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
ro = importlib.import_module("radio-observer_amd")
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
R = int(sys.argv[2]) if len(sys.argv) > 2 else 16384
bins = int(sys.argv[3]) if len(sys.argv) > 3 else 32768
overlap = int(sys.argv[4]) if len(sys.argv) > 4 else bins * 3 // 4
samples = bins + (bins - overlap) * (R - 1)
iq = bench.synth_iq(torch, samples, 0xC3, "cuda:0")
rows = torch.empty((R, bins), dtype=torch.float32, device="cuda")
s = torch.cuda.current_stream().cuda_stream
with ro.Stft(bins=bins, overlap=overlap, precision=ro.RO_PRECISION_F64) as st:
    for _ in range(3):
        st.run_resident(iq, ro.RO_IQ_F32, samples, 0, R, rows, stream=s)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(steps):
        st.run_resident(iq, ro.RO_IQ_F32, samples, 0, R, rows, stream=s)
    e1.record()
    torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / steps
h = hashlib.sha256(rows.cpu().numpy().tobytes()).hexdigest()[:16]
knobs = " ".join("%s=%s" % (k, os.environ[k]) for k in ("RO_F64_FUSED", "RO_F64_STREAM", "RO_F64_RING_ROWS", "RO_F64_WGS", "RO_F64_SCRATCH_MB")
                 if k in os.environ)
print("%-48s bins %d rows/step %d  %.4f ms/step  %.4g rows/s  frac %.4f  rows_hash %s"
      % (knobs or "defaults", bins, R, ms, R / (ms * 1e-3), (8 * (bins - overlap) + 4 * bins) * R / (ms * 1e-3) / 8e12, h), flush=True)
