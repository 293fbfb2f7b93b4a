"""GPU box: soak of the four-step sizes (262144, 524288, 1048576) over shapes the test suite does not use -- overlaps
0 / 25 % / 50 % / 75 % / 87.5 % / bins - 2, odd hops, row strides wider than the row, int16 frames, launches cut at odd
rows (one launch == four uneven launches, bit for bit: the scratch blocks and the nt-leg variants of the column kernel
change with the cut), CUs left out.  A few rows of every shape against the oracle.  Exits non-zero on the first
mismatch."""
import importlib, os, sys
import numpy as np, torch
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
ro = importlib.import_module("radio-observer_amd")
import ro_oracle as oracle

def run(st, iq, fmt, first, n, out, stride):
    st.run_resident(iq, fmt, iq.shape[0], first, n, out, row_stride=stride, stream=torch.cuda.current_stream().cuda_stream)

g = torch.Generator(device="cuda"); g.manual_seed(4321)
shapes = []
for bins, R in ((262144, 1100), (524288, 600), (1048576, 300)):
    for ov in (0, bins // 4, bins // 2, bins - bins // 4, bins - bins // 8, bins - 2, bins // 2 + 12345, bins - 1001):
        rows = R if bins - ov >= bins // 8 else 4 * R
        shapes.append((bins, ov, min(rows, (1 << 31) // (bins - ov) // 8)))
for k, (bins, ov, R) in enumerate(shapes):
    hop = bins - ov
    R = max(8, min(R, (3 << 30) // (8 * hop) ))                       # at most 3 GiB of samples
    samples = bins + hop * (R - 1)
    i16 = k % 3 == 2
    if i16:
        iq = torch.randint(-20000, 20000, (samples, 2), generator=g, device="cuda", dtype=torch.int16)
        fmt = ro.RO_IQ_I16
    else:
        iq = torch.randn((samples, 2), generator=g, device="cuda", dtype=torch.float32)
        fmt = ro.RO_IQ_F32
    stride = bins + (64 if k % 2 else 0)
    rows = torch.empty((R, stride), dtype=torch.float32, device="cuda")
    kw = dict(spare_cus_per_xcd=(0, 3, 9)[k % 3])
    with ro.Stft(bins=bins, overlap=ov, **kw) as st:
        run(st, iq, fmt, 0, R, rows, stride); torch.cuda.synchronize()
        cuts = [0, R // 7 + 3, R // 2 + 11, R - 5, R]
        part = torch.empty((max(b - a for a, b in zip(cuts, cuts[1:])), stride), dtype=torch.float32, device="cuda")
        for a, b in zip(cuts, cuts[1:]):
            part.fill_(float("nan")); run(st, iq, fmt, a, b - a, part, stride); torch.cuda.synchronize()
            if not torch.equal(part[:b - a, :bins].view(torch.int32), rows[a:b, :bins].view(torch.int32)):
                print("MISMATCH shard", bins, ov, R, a, b); sys.exit(1)
    worst = 0.0
    for r in (0, R // 3, R - 1):
        x = iq[r * hop:r * hop + bins].cpu().numpy().astype(np.float32)
        want = oracle.stft(x, bins, ov)[0]
        got = rows[r, :bins].cpu().numpy()
        worst = max(worst, float(np.abs(got - want).max() / want.max()))
    print("bins=%d overlap=%d rows=%d %s stride=%d spare=%d: cut-invariant, oracle err %.3g" % (
        bins, ov, R, "int16" if i16 else "float32", stride, kw["spare_cus_per_xcd"], worst), flush=True)
    if worst > 1e-5:
        sys.exit(2)
    del rows, iq, part
print("soak ok")
