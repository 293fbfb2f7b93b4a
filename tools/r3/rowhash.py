#!/usr/bin/env python3
"""GPU box: hashes of what one build of libro_stft.so makes of the C3/C4 bench input (rows, scan records, band tile),
so two builds can be compared bit for bit:  RO_STFT_LIB=build/ab/libro_stft_X.so python3 tools/r3/rowhash.py [rows] [f32|i16]"""
import hashlib, importlib, os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench
ro = importlib.import_module("radio-observer_amd")
rows = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
fmt = sys.argv[2] if len(sys.argv) > 2 else "f32"
bins, overlap = 32768, 24576
hop = bins - overlap
samples = bins + hop * (rows - 1)
iq = bench.synth_iq(torch, samples, 0xC3, "cuda:0")
fmt_id = ro.RO_IQ_F32
if fmt == "i16":
    iq = (iq * 300).round().clamp(-32768, 32767).to(torch.int16).contiguous()
    fmt_id = ro.RO_IQ_I16
bands = bench.make_bands(ro)
first, cols = 23278, 615
d_rows = torch.zeros((rows, bins), dtype=torch.float32, device="cuda:0")
d_recs = torch.zeros((rows, 3), dtype=torch.float32, device="cuda:0")
d_tile = torch.zeros((rows, cols), dtype=torch.float32, device="cuda:0")
with ro.Stft(bins=bins, overlap=overlap, device=0, bands=bands, tile=(first, cols)) as st:
    st.run_resident(iq, fmt_id, samples, 0, rows, d_rows, d_tile=d_tile, d_records=d_recs,
                    stream=torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
h = lambda t: hashlib.sha256(t.cpu().numpy().tobytes()).hexdigest()[:16]
r = d_rows.cpu().numpy()
print("lib", os.environ.get("RO_STFT_LIB", "default"), "rows", rows, fmt, "rows_hash", h(d_rows), "records_hash", h(d_recs),
      "tile_hash", h(d_tile), "tile==rows", bool(np.array_equal(r[:, first:first + cols], d_tile.cpu().numpy())),
      "finite", bool(np.isfinite(r).all()), "row0[:3]", r[0, :3], "rowmax", float(r.max()))
