"""GPU box: shard invariance (one launch == three uneven launches, bit for bit) and a few oracle rows for the plans
touched at the end of round 2, over overlaps the test suite does not use.  Prints one line per shape; exits non-zero
on the first mismatch."""
import importlib, os, sys
import numpy as np, torch
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
ro = importlib.import_module("radio-observer_amd")
import ro_oracle as oracle

def run(st, iq, first, n, out):
    st.run_resident(iq, ro.RO_IQ_F32, iq.shape[0], first, n, out, stream=torch.cuda.current_stream().cuda_stream)

shapes = []
for bins, R in ((8192, 20000), (16384, 12000), (32768, 6000), (65536, 3000), (131072, 1500), (262144, 1300), (32728, 1500), (12000, 4000)):
    for ov in (0, bins // 2, bins - bins // 8, bins - 2):
        if ov == bins - 2 and bins > 32768:
            continue
        shapes.append((bins, ov, R if ov else max(64, R // 4)))
g = torch.Generator(device="cuda"); g.manual_seed(1234)
for bins, ov, R in shapes:
    hop = bins - ov
    samples = bins + hop * (R - 1)
    iq = torch.randn((samples, 2), generator=g, device="cuda", dtype=torch.float32)
    rows = torch.empty((R, bins), dtype=torch.float32, device="cuda")
    with ro.Stft(bins=bins, overlap=ov) as st:
        run(st, iq, 0, R, rows); torch.cuda.synchronize()
        cuts = [0, R // 7 + 3, R // 2 + 11, R]
        part = torch.empty((max(b - a for a, b in zip(cuts, cuts[1:])), bins), dtype=torch.float32, device="cuda")
        for a, b in zip(cuts, cuts[1:]):
            part.fill_(float("nan")); run(st, iq, a, b - a, part); torch.cuda.synchronize()
            if not torch.equal(part[:b - a].view(torch.int32), rows[a:b].view(torch.int32)):
                print("MISMATCH shard", bins, ov, R, a, b); sys.exit(1)
    worst = 0.0
    host = None
    for r in (0, R // 3, R - 1):
        x = iq[r * hop:r * hop + bins].cpu().numpy()
        want = oracle.stft(x, bins, ov)[0]
        got = rows[r].cpu().numpy()
        worst = max(worst, float(np.abs(got - want).max() / want.max()))
    print("bins=%d overlap=%d rows=%d shard-invariant, oracle err %.3g" % (bins, ov, R, worst), flush=True)
    if worst > 1e-5:
        sys.exit(2)
    del rows, iq
print("soak ok")
