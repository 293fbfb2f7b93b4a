#!/usr/bin/env python3
"""Generate the golden fixtures in this directory.

What pins them: the reference has no golden vectors for this path and cannot be built in
this image (cppapp + libfftw3 absent), so these fixtures are produced by the CPU oracle
(oracle/ro_oracle.c) and are only written after an INDEPENDENT implementation agrees:
numpy's pocketfft (FP64) for the spectra, numpy sort/argmax for the scans.  They are
regression vectors for the oracle and known answers for the HIP path -- "parity unpinned"
in the sense of DESIGN.md.  Run:  python tests/golden/make_golden.py
"""
import hashlib
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import ro_oracle as O                       # noqa: E402
from util import add_chirp, add_tone, noise_iq  # noqa: E402

FS = 48000


def numpy_rows(iq, bins, overlap, w, gain=0.0):
    """independent restatement with numpy only (no oracle code)"""
    x = iq.astype(np.float64)
    z = x[:, 0] + 1j * (x[:, 1] + gain)
    hop = bins - overlap
    n = (len(z) - bins) // hop + 1
    out = np.empty((n, bins), np.float32)
    for r in range(n):
        spec = np.fft.fft(z[r * hop:r * hop + bins] * w.astype(np.float64))
        out[r] = np.abs(np.fft.fftshift(spec)).astype(np.float32)
    return out


def check_rows(rows, ref):
    err = np.abs(rows.astype(np.float64) - ref.astype(np.float64)).max(axis=1) / ref.max(axis=1)
    # float32 rounding of two FP64 results that agree to ~1e-15: identical or 1 ulp apart
    assert err.max() < 2e-7, err.max()


def main():
    out = {}
    # ---- G1 window tables
    for kind in ("nuttall", "hann"):
        for n in (1024, 4096, 32768):
            w = O.window(n, kind)
            out["win_%s_%d_sha256" % (kind, n)] = np.frombuffer(
                hashlib.sha256(w.tobytes()).digest(), dtype=np.uint8)
            idx = np.linspace(0, n - 1, 64).astype(np.int64)
            out["win_%s_%d_idx" % (kind, n)] = idx
            out["win_%s_%d_val" % (kind, n)] = w[idx]
    # ---- G3 spectra
    rng = np.random.default_rng(0xC1)
    c1 = add_tone(noise_iq(rng, 1024 + 3 * 512, 300.0), 10400.0, 8000.0)
    c1 = np.clip(np.rint(c1), -32768, 32767).astype(np.int16)
    rows = O.stft(c1.astype(np.float64), 1024, 512)
    check_rows(rows, numpy_rows(c1, 1024, 512, O.window(1024)))
    out["c1_iq_i16"], out["c1_rows"] = c1, rows

    rng = np.random.default_rng(0xC2)
    c2 = noise_iq(rng, 4096 + 3 * 2048)
    rows = O.stft(c2, 4096, 2048)
    check_rows(rows, numpy_rows(c2, 4096, 2048, O.window(4096)))
    out["c2_iq_f32"], out["c2_rows"] = c2, rows
    rows_h = O.stft(c2, 4096, 2048, w=O.window(4096, "hann"))
    check_rows(rows_h, numpy_rows(c2, 4096, 2048, O.window(4096, "hann")))
    out["c2_rows_hann"] = rows_h

    rng = np.random.default_rng(0xC3)
    c3 = add_tone(noise_iq(rng, 32768 + 8192), 10600.0, 30.0)
    rows = O.stft(c3, 32768, 24576)
    check_rows(rows, numpy_rows(c3, 32768, 24576, O.window(32768)))
    out["c3_iq_f32"], out["c3_rows"] = c3, rows

    # ---- G5 scan records on the C4 band layout (radio-observer.json), incl. ties
    b = O.bolid_bands(32768, FS, 24576, 10300, 10900, 9000, 9600, 2, 5, 40)
    rng = np.random.default_rng(0xC4)
    band_lo = min(b.low_noise, b.low_detect) - 64
    band_hi = max(b.low_noise + b.noise_width, b.low_detect + b.detect_width) + 64
    srows = np.zeros((24, 32768), np.float32)
    srows[:, band_lo:band_hi] = np.abs(rng.standard_normal((24, band_hi - band_lo))).astype(np.float32)
    for r in range(8):                               # equal maxima -> last index
        idx = rng.choice(b.detect_width, 2 + r % 3, replace=False)
        srows[r, b.low_detect + idx] = 9.0
    srows[8:12, band_lo:band_hi] = rng.integers(0, 4, (4, band_hi - band_lo)).astype(np.float32)
    srows[12, b.low_detect] = 50.0                   # peak at the band's first bin
    n, p, a = O.scan_rows(srows, b.low_noise, b.noise_width, b.low_detect, b.detect_width, b.avg_bins)
    for r in range(24):                              # independent check with numpy
        nb = np.sort(srows[r, b.low_noise:b.low_noise + b.noise_width])
        assert n[r] == np.float32(nb[b.noise_width // 4] * 2.0)
        db = srows[r, b.low_detect:b.low_detect + b.detect_width]
        assert p[r] == np.flatnonzero(db == db.max()).max()
        s0 = b.low_detect + p[r] - b.avg_bins // 2
        acc = 0.0
        for v in srows[r, s0:s0 + b.avg_bins]:
            acc += float(v)
        assert a[r] == np.float32(acc / b.avg_bins)
    out["scan_band"] = np.array([band_lo, band_hi], np.int64)
    out["scan_rows_band"] = srows[:, band_lo:band_hi]
    out["scan_bands"] = np.array([b.low_noise, b.noise_width, b.low_detect, b.detect_width, b.avg_bins], np.int32)
    out["scan_noise"], out["scan_peak"], out["scan_average"] = n, p, a

    # ---- G6 FSM: scripted detect sequence, advance 11, jitter 29 (radio-observer.json @ C3)
    rate = O.lib().ro_oracle_fft_sample_rate(FS, 32768, 24576)
    fsm = O.BolidFsm(b.advance, b.jitter, rate, FS, 10300.0, 10900.0)
    script = [0] * 40 + [1] * 14 + [0] * 10 + [1] * 3 + [0] * 60 + [1] * 1 + [0] * 40
    events = []
    for i, d in enumerate(script):
        nn, aa = 1.0, (5.0 if d else 1.0)            # detect <=> a > 2 n
        ev = fsm.update(nn, aa, 10500.0 + i, (i + 1) % 2816)
        if ev.fired:
            events.append([i, ev.snap_start, ev.snap_length, ev.raw_length, ev.duration_s,
                           ev.peak_freq, ev.fmin, ev.fmax])
    out["fsm_script"] = np.array(script, np.int8)
    out["fsm_events"] = np.array(events, np.float64)

    # ---- C4 end-to-end: noise + one 2 s chirp, 41 rows
    rng = np.random.default_rng(0xC4)
    hop = 8192
    c4 = noise_iq(rng, 32768 + 40 * hop)
    add_chirp(c4, 9 * hop + 16384, 2.0, 10800.0, -100.0, 3.0)
    rows = O.stft(c4, 32768, 24576)
    n, p, a = O.scan_rows(rows, b.low_noise, b.noise_width, b.low_detect, b.detect_width, b.avg_bins)
    out["c4_seed_note"] = np.array([0xC4], np.int64)
    out["c4_detect"] = (a.astype(np.float64) > n.astype(np.float64) * 2.0)
    out["c4_margin"] = a.astype(np.float64) / (2.0 * n.astype(np.float64))
    out["c4_noise"], out["c4_peak"], out["c4_average"] = n, p, a

    path = os.path.join(HERE, "hotpath_golden.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes;", "c4 detect rows:",
          np.flatnonzero(out["c4_detect"]).tolist(), "min margin on detect %.2f, max off %.2f"
          % (out["c4_margin"][out["c4_detect"]].min(), out["c4_margin"][~out["c4_detect"]].max()))
    print("fsm events:", events)


if __name__ == "__main__":
    main()
