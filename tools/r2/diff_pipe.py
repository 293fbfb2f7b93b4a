"""diagnostic: rows of the default library vs a -DRO_PIPE=0 build of the same sources, bit for bit; prints where they differ"""
import ctypes, importlib, os, subprocess, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
ref = os.path.join(ROOT, "build", "r2", "libro_stft_nopipe.so")
def run(lib, R, iq):
    env = dict(os.environ, RO_STFT_LIB=lib) if lib else dict(os.environ)
    code = r'''
import importlib, sys, numpy as np, torch
sys.path.insert(0, %r)
ro = importlib.import_module("radio-observer_amd")
iq = torch.from_numpy(np.load("/tmp/iq.npy")).cuda()
R = %d
rows = torch.empty((R, 32768), dtype=torch.float32, device="cuda")
with ro.Stft(bins=32768, overlap=24576) as st:
    st.run_resident(iq, ro.RO_IQ_F32, iq.shape[0], 0, R, rows, stream=torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
np.save(sys.argv[1], rows.cpu().numpy())
''' % (ROOT, R)
    out = "/tmp/rows_%s.npy" % ("ref" if lib == ref else "new")
    subprocess.check_call([sys.executable, "-c", code, out], env=env)
    return np.load(out)
R = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
rng = np.random.default_rng(1)
iq = rng.standard_normal((32768 + 8192 * (R - 1), 2)).astype(np.float32)
np.save("/tmp/iq.npy", iq)
a = run(os.environ.get("RO_TEST_LIB"), R, iq); b = run(ref, R, iq)
bad = np.argwhere(a.view(np.uint32) != b.view(np.uint32))
print("rows", R, "mismatches", len(bad))
if len(bad):
    print(" cols", len(np.unique(bad[:, 1])), "rows", len(np.unique(bad[:, 0])))
    print(" c//1024:", {int(k): int(v) for k, v in zip(*np.unique(bad[:, 1] // 1024, return_counts=True))})
    print(" c%64:", {int(k): int(v) for k, v in zip(*np.unique(bad[:, 1] % 64, return_counts=True))})
    for r, c in bad[:4]:
        print(" ", r, c, a[r, c], b[r, c])
