"""Shared helpers for the parity tests (test infrastructure)."""
import numpy as np

# radio-observer.json's recorder settings (radio-observer.json:62-87)
JSON_BOLID = dict(min_detect=10300.0, max_detect=10900.0, min_noise=9000.0, max_noise=9600.0,
                  advance_time=2.0, jitter_time=5.0, avg_freq_range=40.0)


def noise_iq(rng, samples, sigma=1.0):
    """float32 interleaved [samples, 2] white Gaussian noise."""
    return (rng.standard_normal((samples, 2)) * sigma).astype(np.float32)


def add_tone(iq, freq, amp, fs=48000, phase=0.0):
    t = np.arange(iq.shape[0], dtype=np.float64)
    ph = 2.0 * np.pi * freq * t / fs + phase
    iq[:, 0] += (amp * np.cos(ph)).astype(np.float32)
    iq[:, 1] += (amp * np.sin(ph)).astype(np.float32)
    return iq


def add_chirp(iq, start_sample, duration_s, f0, rate_hz_per_s, amp, fs=48000):
    n = int(duration_s * fs)
    n = min(n, iq.shape[0] - start_sample)
    if n <= 0:
        return iq
    t = np.arange(n, dtype=np.float64) / fs
    ph = 2.0 * np.pi * (f0 * t + 0.5 * rate_hz_per_s * t * t)
    iq[start_sample:start_sample + n, 0] += (amp * np.cos(ph)).astype(np.float32)
    iq[start_sample:start_sample + n, 1] += (amp * np.sin(ph)).astype(np.float32)
    return iq


def rel_to_row_max(got, want):
    """max over rows of max_k |got-want| / max_k |want|  (the norm-wise 1e-5 bar)."""
    got = np.asarray(got, dtype=np.float64)
    want = np.asarray(want, dtype=np.float64)
    err = np.abs(got - want).max(axis=-1)
    ref = np.abs(want).max(axis=-1)
    ref = np.where(ref == 0, 1.0, ref)
    return float((err / ref).max())
