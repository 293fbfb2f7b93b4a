"""The committed fixtures (tests/golden/hotpath_golden.npz, made by make_golden.py) against
the oracle: guards the oracle against regressions on this box's libm / compiler."""
import hashlib
import os

import numpy as np
import pytest

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "hotpath_golden.npz")


@pytest.fixture(scope="module")
def gold():
    return np.load(G)


def test_window_hashes(oracle, gold):
    for kind in ("nuttall", "hann"):
        for n in (1024, 4096, 32768):
            w = oracle.window(n, kind)
            assert np.array_equal(w[gold["win_%s_%d_idx" % (kind, n)]], gold["win_%s_%d_val" % (kind, n)])
            digest = np.frombuffer(hashlib.sha256(w.tobytes()).digest(), dtype=np.uint8)
            assert np.array_equal(digest, gold["win_%s_%d_sha256" % (kind, n)])


def test_spectra(oracle, gold):
    assert np.array_equal(oracle.stft(gold["c1_iq_i16"].astype(np.float64), 1024, 512), gold["c1_rows"])
    assert np.array_equal(oracle.stft(gold["c2_iq_f32"], 4096, 2048), gold["c2_rows"])
    assert np.array_equal(oracle.stft(gold["c2_iq_f32"], 4096, 2048, w=oracle.window(4096, "hann")),
                          gold["c2_rows_hann"])
    assert np.array_equal(oracle.stft(gold["c3_iq_f32"], 32768, 24576), gold["c3_rows"])


def test_scan_records(oracle, gold):
    lo, hi = gold["scan_band"]
    rows = np.zeros((gold["scan_rows_band"].shape[0], 32768), np.float32)
    rows[:, lo:hi] = gold["scan_rows_band"]
    ln, nw, ld, dw, avg = [int(x) for x in gold["scan_bands"]]
    n, p, a = oracle.scan_rows(rows, ln, nw, ld, dw, avg)
    assert np.array_equal(n, gold["scan_noise"]) and np.array_equal(p, gold["scan_peak"])
    assert np.array_equal(a, gold["scan_average"])


def test_fsm_events(oracle, gold):
    b = oracle.bolid_bands(32768, 48000, 24576, 10300, 10900, 9000, 9600, 2, 5, 40)
    rate = oracle.lib().ro_oracle_fft_sample_rate(48000, 32768, 24576)
    fsm = oracle.BolidFsm(b.advance, b.jitter, rate, 48000, 10300.0, 10900.0)
    events = []
    for i, d in enumerate(gold["fsm_script"]):
        ev = fsm.update(1.0, 5.0 if d else 1.0, 10500.0 + i, (i + 1) % 2816)
        if ev.fired:
            events.append([i, ev.snap_start, ev.snap_length, ev.raw_length, ev.duration_s, ev.peak_freq,
                           ev.fmin, ev.fmax])
    assert np.array_equal(np.array(events, np.float64), gold["fsm_events"])
