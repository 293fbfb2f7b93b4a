import ctypes as C, importlib, os, sys
import numpy as np, torch
sys.path.insert(0, "/root/repo" if os.path.exists("/root/repo") else ".")
ro = importlib.import_module("radio-observer_amd")
lib = ro.library()
lib.ro_stft_debug_stamps.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
bins, overlap, R = 32768, 24576, 16384
hop = bins - overlap; samples = bins + hop * (R - 1)
iq = torch.randn((samples, 2), device="cuda", dtype=torch.float32)
rows = torch.empty((R, bins), device="cuda", dtype=torch.float32)
st = ro.Stft(bins=bins, overlap=overlap)
for _ in range(2):
    st.run_resident(iq, ro.RO_IQ_F32, samples, 0, R, rows, stream=torch.cuda.current_stream().cuda_stream)
torch.cuda.synchronize()
lib.ro_stft_debug_stamps(st._h, None, 0)
st.run_resident(iq, ro.RO_IQ_F32, samples, 0, R, rows, stream=torch.cuda.current_stream().cuda_stream)
torch.cuda.synchronize()
buf = np.zeros(4096 * 16, dtype=np.uint64)
lib.ro_stft_debug_stamps(st._h, buf.ctypes.data_as(C.c_void_p), buf.size)
a = buf.reshape(-1, 16); a = a[a[:, 9] > 0].astype(np.float64)
per = a.sum(0) / a[:, 9].sum()
print("per row (both exchanges, both planes summed): scatter issue %.0f, wait+barrier %.0f, gather issue %.0f, wait+barrier %.0f" % (per[12], per[13], per[14], per[15]))
print("stamp3/5 (rest of exchanges)", per[3], per[5])
