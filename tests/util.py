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


# ---- C5: the 8-hour synthetic stream, generated on the device ---------------------------------
# Counter-based so that any shard can regenerate its own slice without the rest of the stream: the stream is cut into
# chunks of C5_CHUNK samples, chunk c is torch's Philox generator seeded with (0xC5 << 32) + c, and the chirps are a
# pure function of the sample index.  C4's signal model: sigma = 1 noise, a linear chirp every 30 s starting at 20 s,
# 10 800 Hz falling 100 Hz/s, amplitude 3 sigma, durations cycling through 0.5 / 1 / 2 / 4 s.
C5_CHUNK = 1 << 24
C5_SAMPLES = 8 * 3600 * 48000
C5_DURATIONS = (0.5, 1.0, 2.0, 4.0)


def c5_chirps(total_samples=C5_SAMPLES, fs=48000):
    """[(start_sample, samples)] of every chirp that fits the stream"""
    out, i = [], 0
    while True:
        s0 = int((20.0 + 30.0 * i) * fs)
        n = int(C5_DURATIONS[i % 4] * fs)
        if s0 + n > total_samples:
            return out
        out.append((s0, n))
        i += 1


def c5_slice(torch, first_sample, samples, total_samples=C5_SAMPLES, device="cuda", fs=48000):
    """float32 [samples, 2] = samples [first_sample, +samples) of the C5 stream, generated on `device`"""
    out = torch.empty((samples, 2), dtype=torch.float32, device=device)
    g = torch.Generator(device=device)
    c0, c1 = first_sample // C5_CHUNK, (first_sample + samples - 1) // C5_CHUNK
    for c in range(c0, c1 + 1):
        g.manual_seed((0xC5 << 32) + c)
        chunk = torch.randn((C5_CHUNK, 2), generator=g, device=device, dtype=torch.float32)
        lo = max(first_sample, c * C5_CHUNK)
        hi = min(first_sample + samples, (c + 1) * C5_CHUNK)
        out[lo - first_sample:hi - first_sample] = chunk[lo - c * C5_CHUNK:hi - c * C5_CHUNK]
        del chunk
    for s0, n in c5_chirps(total_samples, fs):
        lo, hi = max(s0, first_sample), min(s0 + n, first_sample + samples)
        if lo >= hi:
            continue
        t = torch.arange(lo - s0, hi - s0, device=device, dtype=torch.float64) / fs
        ph = 2.0 * np.pi * (10800.0 * t + 0.5 * -100.0 * t * t)
        out[lo - first_sample:hi - first_sample, 0] += (3.0 * torch.cos(ph)).to(torch.float32)
        out[lo - first_sample:hi - first_sample, 1] += (3.0 * torch.sin(ph)).to(torch.float32)
    return out
