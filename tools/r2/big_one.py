"""GPU box: one small large-transform launch against numpy's FFT.  usage: big_one.py BINS OVERLAP ROWS"""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import importlib
import numpy as np, torch
ro = importlib.import_module("radio-observer_amd")
bins, overlap, nrows = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
hop = bins - overlap
rng = np.random.default_rng(1)
n = bins + (nrows - 1) * hop
iq = (rng.standard_normal((n, 2)) * 0.3).astype(np.float32)
d_iq = torch.from_numpy(iq).cuda()
d_rows = torch.full((nrows, bins), float("nan"), dtype=torch.float32, device="cuda")
with ro.Stft(bins=bins, overlap=overlap) as st:
    st.run_resident(d_iq, ro.RO_IQ_F32, n, 0, nrows, d_rows, stream=torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
got = d_rows.cpu().numpy()
w = ro.window_table(ro.RO_WINDOW_NUTTALL, bins).astype(np.float64)
x = iq[:, 0].astype(np.float64) + 1j * iq[:, 1].astype(np.float64)
worst = 0.0
for r in range(nrows):
    X = np.fft.fftshift(np.abs(np.fft.fft(x[r * hop:r * hop + bins] * w)))
    worst = max(worst, float(np.max(np.abs(got[r] - X)) / np.max(X)))
print("bins=%d rows=%d max err rel to row max %.3g finite=%s" % (bins, nrows, worst, np.isfinite(got).all()))
