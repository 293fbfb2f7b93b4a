#!/usr/bin/env python3
"""GPU box: the register-resident FP64 kernel (csrc/ro_f64reg.hip) against the oracle at every size it takes, then its
rate at C2 and C3 with HIP events.  `python tools/r6/f64r_check.py [--rate-only] [--sizes 4096,32768]`"""
import argparse
import importlib
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402

ro = importlib.import_module("radio-observer_amd")


def rows_of(iq, bins, overlap, fmt=None, **kw):
    fmt = ro.RO_IQ_F32 if fmt is None else fmt
    d_iq = torch.from_numpy(np.ascontiguousarray(iq)).cuda()
    rows = ro.row_count(iq.shape[0], bins, overlap)
    d_rows = torch.full((rows, bins), float("nan"), dtype=torch.float32, device="cuda")
    with ro.Stft(bins=bins, overlap=overlap, precision=ro.RO_PRECISION_F64, **kw) as st:
        st.run_resident(d_iq, fmt, iq.shape[0], 0, rows, d_rows, stream=torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
    return d_rows.cpu().numpy()


def parity(sizes):
    import ro_oracle
    from util import add_tone, noise_iq
    ro_oracle.lib()
    ok = True
    for bins in sizes:
        for overlap, R in ((bins * 3 // 4, 70), (0, 9), (bins - 2, 300)):
            hop = bins - overlap
            rng = np.random.default_rng(bins + overlap)
            iq = add_tone(noise_iq(rng, bins + (R - 1) * hop), 7000.0, 1000.0)
            got = rows_of(iq, bins, overlap)
            want = ro_oracle.stft(iq, bins, overlap)
            w64 = want.astype(np.float64)
            e = np.abs(got.astype(np.float64) - w64) / np.maximum(w64, 1e-300)
            nanc = int(np.isnan(got).sum())
            print("bins %6d overlap %6d rows %4d: per-bin max %.3g  identical floats %.5f  nan %d"
                  % (bins, overlap, got.shape[0], np.nanmax(e) if e.size else 0, (got == want).mean(), nanc), flush=True)
            ok = ok and nanc == 0 and e.max() <= 2e-7
        # int16 + gain + custom window
        rng = np.random.default_rng(3)
        hop = min(1024, bins // 4)
        i16 = rng.integers(-20000, 20000, size=(bins + 37 * hop, 2), dtype=np.int16)
        w = rng.random(bins).astype(np.float32)
        got = rows_of(i16, bins, bins - hop, fmt=ro.RO_IQ_I16, window_table=w, iq_gain=123.5)
        want = ro_oracle.stft(i16.astype(np.float64), bins, bins - hop, w=w, gain=123.5)
        e = np.abs(got.astype(np.float64) - want) / np.maximum(want.astype(np.float64), 1e-300)
        print("bins %6d int16 + gain + custom window: per-bin max %.3g" % (bins, e.max()), flush=True)
        ok = ok and e.max() <= 2e-7
    return ok


def rate(bins, overlap, R, steps=12, spectra=False):
    hop = bins - overlap
    T = bins + (R - 1) * hop
    g = torch.Generator(device="cuda").manual_seed(bins)
    d_iq = torch.randn((T, 2), dtype=torch.float32, device="cuda", generator=g)
    d_rows = torch.empty((R, bins, 2) if spectra else (R, bins), dtype=torch.float32, device="cuda")
    s = torch.cuda.current_stream()
    with ro.Stft(bins=bins, overlap=overlap, precision=ro.RO_PRECISION_F64) as st:
        launch = st.spectra_resident if spectra else st.run_resident
        for _ in range(3):
            launch(d_iq, ro.RO_IQ_F32, T, 0, R, d_rows, stream=s.cuda_stream)
        torch.cuda.synchronize()
        evs = [torch.cuda.Event(enable_timing=True) for _ in range(steps + 1)]
        evs[0].record(s)
        for i in range(steps):
            launch(d_iq, ro.RO_IQ_F32, T, 0, R, d_rows, stream=s.cuda_stream)
            evs[i + 1].record(s)
        torch.cuda.synchronize()
    ms = np.array([evs[i].elapsed_time(evs[i + 1]) for i in range(steps)])
    alg = hop * 8 + bins * (8 if spectra else 4)
    print("%sbins %6d overlap %6d rows %6d: %.3f ms per launch (min %.3f)  %.3e rows/s  %.1f GB/s algorithmic = %.3f of 8 TB/s"
          % ("complex spectra, " if spectra else "", bins, overlap, R, ms.mean(), ms.min(), R / (ms.mean() * 1e-3), alg * R / (ms.mean() * 1e-3) / 1e9,
             alg * R / (ms.mean() * 1e-3) / 8e12), flush=True)


def main():
    p = argparse.ArgumentParser()
    p.add_argument("--rate-only", action="store_true")
    p.add_argument("--spectra", action="store_true", help="the rates of ro_stft_spectra_resident instead (algorithmic bytes: hop 8 + bins 8)")
    p.add_argument("--sizes", default="256,512,1024,2048,4096,8192,16384,32768,65536")
    a = p.parse_args()
    sizes = [int(x) for x in a.sizes.split(",")]
    ok = True
    if not a.rate_only:
        ok = parity(sizes)
        print("parity:", "OK" if ok else "FAILED", flush=True)
    if ok:
        shapes = {256: (128, 1 << 20), 512: (256, 1 << 19), 1024: (512, 1 << 18), 2048: (1024, 1 << 17), 4096: (2048, 65536), 8192: (6144, 32768), 16384: (12288, 16384), 32768: (24576, 16384), 65536: (49152, 8192),
                  131072: (98304, 2048), 262144: (196608, 1024), 524288: (262144, 512), 1048576: (524288, 256)}
        for bins in sizes:
            rate(bins, *shapes[bins], spectra=a.spectra)
    return 0 if ok else 1


if __name__ == "__main__":
    sys.exit(main())
