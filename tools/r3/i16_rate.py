#!/usr/bin/env python3
"""GPU box: the C3 kernel on int16 I/Q (the WAV / sound-card format, src/WAVStream.cpp:62-97) against float32 I/Q, same shape."""
import importlib, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
ro = importlib.import_module("radio-observer_amd")
bins, overlap, R = 32768, 24576, 16384
hop = bins - overlap
samples = bins + hop * (R - 1)
f32 = torch.randn((samples, 2), device="cuda", dtype=torch.float32)
i16 = torch.randint(-20000, 20000, (samples, 2), device="cuda", dtype=torch.int16)
rows = torch.empty((R, bins), device="cuda", dtype=torch.float32)
s = torch.cuda.current_stream().cuda_stream
with ro.Stft(bins=bins, overlap=overlap) as st:
    for rnd in range(3):
        for name, buf, fmt, bps in (("float32", f32, ro.RO_IQ_F32, 8), ("int16", i16, ro.RO_IQ_I16, 4)):
            for _ in range(30):
                st.run_resident(buf, fmt, samples, 0, R, rows, stream=s)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(300):
                st.run_resident(buf, fmt, samples, 0, R, rows, stream=s)
            e1.record(); torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / 300
            alg = (hop * bps + bins * 4) * R
            print("round %d %-8s %.4f ms per launch, %.4g rows/s, algorithmic %.0f B/row -> %.3f of the 8 TB/s peak" % (
                rnd, name, ms, R / ms * 1e3, alg / R, alg / (ms * 1e-3) / 8e12), flush=True)
