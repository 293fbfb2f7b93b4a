#!/usr/bin/env python3
"""GPU box, under `rocprofv3 --pmc FETCH_SIZE` (and again with WRITE_SIZE): launch the STFT kernel
with overlap = 0, where every input byte is read exactly once, so the counter can be calibrated on
this kernel's own access pattern (MI355X_MICROARCH.md §HBM: FETCH_SIZE under-reports wide reads).
Known bytes per launch: rows*bins*8 in (+ bins*4 window + twiddles, negligible), rows*bins*4 out."""
import importlib, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ro = importlib.import_module("radio-observer_amd")
bins, R = 32768, 8192
iq = torch.randn((bins * R, 2), device="cuda", dtype=torch.float32)
rows = torch.empty((R, bins), device="cuda", dtype=torch.float32)
with ro.Stft(bins=bins, overlap=0) as st:
    for _ in range(3):
        st.run_resident(iq, ro.RO_IQ_F32, bins * R, 0, R, rows, stream=torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
print("calibration launch: read %d bytes, wrote %d bytes" % (bins * R * 8, bins * R * 4))
