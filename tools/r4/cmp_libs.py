#!/usr/bin/env python3
"""GPU box: rows of one shape from two builds of libro_stft.so side by side -- where do they differ?
usage: cmp_libs.py LIB_A LIB_B BINS OVERLAP ROWS"""
import os
import subprocess
import sys
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CHILD = r'''
import sys, importlib, numpy as np, torch
sys.path.insert(0, %r)
ro = importlib.import_module("radio-observer_amd")
bins, overlap, rows, out = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
rng = np.random.default_rng(5)
n = bins + (rows - 1) * (bins - overlap)
iq = rng.standard_normal((n, 2)).astype(np.float32)
d = torch.from_numpy(iq).cuda()
with ro.Stft(bins=bins, overlap=overlap, sample_rate=96000, device=0) as st:
    o = torch.empty((rows, bins), dtype=torch.float32, device="cuda")
    st.run_resident(d.data_ptr(), ro.RO_IQ_F32, n, 0, rows, o.data_ptr())
    torch.cuda.synchronize()
    np.save(out, o.cpu().numpy())
''' % ROOT


def run(lib, args, out):
    env = dict(os.environ, RO_STFT_LIB=lib)
    subprocess.run([sys.executable, "-c", CHILD, *args, out], check=True, env=env)
    return np.load(out)


if __name__ == "__main__":
    la, lb, bins, overlap, rows = sys.argv[1:6]
    a = run(la, [bins, overlap, rows], "/tmp/cmp_a.npy")
    b = run(lb, [bins, overlap, rows], "/tmp/cmp_b.npy")
    bad = np.argwhere(a != b)
    print("shape", a.shape, "differing elements", len(bad), "max |a-b|", float(np.abs(a - b).max()), "max a", float(a.max()))
    if len(bad):
        cols = np.unique(bad[:, 1])
        print("rows with differences", np.unique(bad[:, 0])[:16], "columns", len(cols), cols[:32], "mod 1024:", np.unique(cols % 1024)[:32],
              "div n1:", np.unique(cols // (int(bins) // 1024))[:32])
        for r, c in bad[:8]:
            print(r, c, a[r, c], b[r, c])
