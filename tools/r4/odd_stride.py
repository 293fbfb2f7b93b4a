import importlib, sys, numpy as np, torch
import os; sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
ro = importlib.import_module("radio-observer_amd")
g = torch.Generator(device="cuda"); g.manual_seed(7)
for bins, ov, R in ((32768, 24576, 40), (4096, 2048, 50), (65536, 49152, 12), (524288, 262144, 9), (1048576, 524288, 5)):
    hop = bins - ov
    iq = torch.randn((bins + hop * (R - 1), 2), generator=g, device="cuda", dtype=torch.float32)
    ref = torch.empty((R, bins), dtype=torch.float32, device="cuda")
    with ro.Stft(bins=bins, overlap=ov) as st:
        st.run_resident(iq, ro.RO_IQ_F32, iq.shape[0], 0, R, ref); torch.cuda.synchronize()
        for extra in (1, 2, 3, 5):
            stride = bins + extra
            buf = torch.full((R * stride + 8,), float("nan"), dtype=torch.float32, device="cuda")
            for off in (0, 1, 3):
                out = buf[off:off + R * stride]
                st.run_resident(iq, ro.RO_IQ_F32, iq.shape[0], 0, R, out.data_ptr(), row_stride=stride); torch.cuda.synchronize()
                got = out.view(R, stride)[:, :bins]
                ok = torch.equal(got.view(torch.int32), ref.view(torch.int32))
                pad_ok = bool(torch.isnan(out.view(R, stride)[:, bins:]).all())
                print(bins, "stride +%d base +%d floats:" % (extra, off), "rows equal" if ok else "ROWS DIFFER", "padding untouched" if pad_ok else "PADDING WRITTEN", flush=True)
                buf.fill_(float("nan"))
