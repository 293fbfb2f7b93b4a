"""CPU tests of the oracle itself (oracle/ro_oracle.c): independent cross-checks (numpy
pocketfft, direct DFT, numpy sort) and the known answers recorded from the reference in
SURVEY.md (Appendix A-3, Appendix D, §8 a10/a11 -- values observed when the survey session
ran the reference's own FFTBackend::process / helpers)."""
import numpy as np
import pytest


def test_fft_matches_numpy_and_direct_dft(oracle):
    rng = np.random.default_rng(0)
    for n in (2, 8, 64, 1024, 4096, 32768):
        x = rng.standard_normal(n) + 1j * rng.standard_normal(n)
        got, ref = oracle.fft(x), np.fft.fft(x)
        assert np.abs(got - ref).max() <= 2e-15 * np.abs(ref).max() * np.log2(n) + 1e-300
    x = rng.standard_normal(256) + 1j * rng.standard_normal(256)
    assert np.abs(oracle.dft_direct(x) - oracle.fft(x)).max() < 1e-12


def test_fft_of_lengths_that_are_not_a_power_of_two(oracle):
    """FFTW takes any N (src/FFTBackend.cpp:120): the oracle's stand-in is Bluestein over its radix-2 transform"""
    rng = np.random.default_rng(11)
    for n in (6, 10, 258, 1000, 3000, 32728):
        x = rng.standard_normal(n) + 1j * rng.standard_normal(n)
        got = oracle.fft(x)
        assert np.abs(got - np.fft.fft(x)).max() <= 1e-12 * np.abs(got).max()
        if n <= 1000:
            assert np.abs(got - oracle.dft_direct(x)).max() <= 1e-12 * np.abs(got).max()


def test_fft_is_forward_unnormalised(oracle):
    """FFTW_FORWARD: X[k] = sum x[n] exp(-2 pi i k n / N), no 1/N (src/FFTBackend.cpp:120)."""
    n = 64
    x = np.zeros(n, np.complex128)
    x[1] = 1.0
    X = oracle.fft(x)
    assert np.allclose(X, np.exp(-2j * np.pi * np.arange(n) / n), atol=1e-15)
    assert np.allclose(oracle.fft(np.ones(n, np.complex128))[0], n)


def test_window_known_answers(oracle):
    """w[0] ~ -1.86e-9, max 1.0f, mean 0.35575714 at N=32768."""
    w = oracle.window(32768)
    assert w.dtype == np.float32
    assert w[0] == np.float32(-1.8626451e-09)
    assert w.max() == np.float32(1.0)
    assert abs(float(w.astype(np.float64).mean()) - 0.35575714) < 5e-9
    assert (w <= 1.0).all()                                 # the reference asserts this (FFTBackend.cpp:185)
    # formula in double with float coefficients and float-converted i, N-1
    i = np.arange(32768, dtype=np.float32).astype(np.float64)
    d = float(np.float32(32767))
    a = [np.float32(v).astype(np.float64) for v in (0.355768, 0.487396, 0.144232, 0.012604)]
    pi = 4.0 * np.arctan(1.0)
    ref = (a[0] - a[1] * np.cos(2 * pi * i / d) + a[2] * np.cos(4 * pi * i / d) - a[3] * np.cos(6 * pi * i / d))
    # numpy's cos may differ from libm's in the last ulp of the double: allow 1 float ulp
    assert np.abs(w.astype(np.float64) - ref).max() <= 6e-8
    h = oracle.window(4096, "hann")
    assert h[0] == 0.0 and abs(h[2047] - 1.0) < 1e-6


def test_framing_known_answers_from_survey(oracle):
    """driving FFTBackend::process with 1024-sample calls, N=1024,
    overlap 512, T=6444 gave 11 rows, info.offset 0..10, row times 0, 0.010666, ..., 0.031999
    (truncated), rawMark 513, 1025, ..."""
    rng = np.random.default_rng(1)
    T = 6444
    iq = rng.standard_normal(T) + 1j * rng.standard_normal(T)
    s = oracle.Stream(1024, 512)
    rows, infos = [], []
    for i in range(0, T, 1024):
        r, inf = s.process(iq[i:i + 1024])
        rows.append(r)
        infos += inf
    rows = np.concatenate(rows)
    assert rows.shape == (11, 1024) and oracle.row_count(T, 1024, 512) == 11
    assert [i[0] for i in infos] == list(range(11))
    assert [(i[1], i[2]) for i in infos[:4]] == [(0, 0), (0, 10666), (0, 21333), (0, 31999)]
    assert [i[3] for i in infos[:3]] == [513, 1025, 1537]
    # rows equal the batch STFT and numpy's fft of the windowed frames
    assert np.array_equal(rows, oracle.stft(iq, 1024, 512))
    w = oracle.window(1024).astype(np.float64)
    ref = np.abs(np.fft.fftshift(np.fft.fft(iq[3 * 512:3 * 512 + 1024] * w)))
    assert np.abs(rows[3] - ref).max() / ref.max() < 1e-7


@pytest.mark.parametrize("chunk", [1, 7, 512, 1000, 4096, 5000])
def test_rows_do_not_depend_on_call_chunking(oracle, chunk):
    """FFTBackend::process keeps leftovers in window_ (src/FFTBackend.cpp:261-273)."""
    rng = np.random.default_rng(2)
    T = 3 * 1024 + 333
    iq = rng.standard_normal(T) + 1j * rng.standard_normal(T)
    s = oracle.Stream(1024, 768)
    rows = [s.process(iq[i:i + chunk])[0] for i in range(0, T, chunk)]
    rows = np.concatenate(rows)
    assert np.array_equal(rows, oracle.stft(iq, 1024, 768))
    assert rows.shape[0] == oracle.row_count(T, 1024, 768) == (T - 1024) // 256 + 1


def test_row_count_edges(oracle):
    assert oracle.row_count(0, 1024, 512) == 0
    assert oracle.row_count(1023, 1024, 512) == 0
    assert oracle.row_count(1024, 1024, 512) == 1
    assert oracle.row_count(1535, 1024, 512) == 1
    assert oracle.row_count(1536, 1024, 512) == 2
    assert oracle.row_count(4096, 1024, 5000) == 4096 - 1024 + 1      # overlap clamps to N-1 (cpp:109)
    assert oracle.row_count(4096, 1024, -3) == 4                       # and to 0 (cpp:108)


def test_bin_mapping_known_answers(oracle):
    """bins for the frequencies of radio-observer.json at N=32768."""
    L = oracle.lib()
    want = {10300: 23415, 10900: 23825, 9000: 22528, 9600: 22937, 10100: 23278, 11000: 23893,
            12000: 24576, 0: 16384, 40: 16411}
    for f, b in want.items():
        assert L.ro_oracle_frequency_to_bin(32768, 48000, float(f)) == b
    assert L.ro_oracle_frequency_to_bin(32768, 48000, 1e9) == 32767      # clamp
    assert L.ro_oracle_frequency_to_bin(32768, 48000, -1e9) == 0
    assert L.ro_oracle_bin_to_frequency(32768, 48000, 16384) == 0.0
    assert L.ro_oracle_bin_to_frequency(32768, 48000, 0) == -24000.0
    assert abs(L.ro_oracle_fft_sample_rate(48000, 32768, 24576) - 5.859375) == 0.0
    assert L.ro_oracle_fft_sample_rate(48000, 1024, 512) == 93.75
    assert L.ro_oracle_time_to_fft_samples(2.0, np.float32(5.859375)) == 11
    assert L.ro_oracle_time_to_fft_samples(5.0, np.float32(5.859375)) == 29
    b = oracle.bolid_bands(32768, 48000, 24576, 10300, 10900, 9000, 9600, 2, 5, 40)
    assert (b.low_detect, b.detect_width, b.low_noise, b.noise_width, b.advance, b.jitter, b.avg_bins) == \
        (23415, 410, 22528, 409, 11, 29, 27)
    # BolidRecorder.cpp:102-104: averageBinRange_ is 0 at N=1024 (the reference asserts)
    assert oracle.bolid_bands(1024, 48000, 512, 10300, 10900, 9000, 9600, 2, 5, 40).avg_bins == 0


def test_wftime_truncates_to_microseconds(oracle):
    import ctypes as C
    L = oracle.lib()
    s, u = C.c_int64(), C.c_int64()
    L.ro_oracle_wftime_add_samples(0, 0, 1536, 48000, C.byref(s), C.byref(u))
    assert (s.value, u.value) == (0, 32000)
    L.ro_oracle_wftime_add_samples(0, 0, 512, 48000, C.byref(s), C.byref(u))
    assert (s.value, u.value) == (0, 10666)                 # 10666.67 truncated (WFTime.h:111-112)
    L.ro_oracle_wftime_add_samples(5, 999999, 48000 * 3 + 1, 48000, C.byref(s), C.byref(u))
    assert (s.value, u.value) == (9, 19)                    # 999999 + 20 us carries a second


def test_scan_functions(oracle):
    import ctypes as C
    L = oracle.lib()
    rng = np.random.default_rng(3)
    for width in (1, 4, 5, 51, 409, 410):
        x = rng.random(width).astype(np.float32)
        buf = x.copy()
        n = L.ro_oracle_noise(buf.ctypes.data_as(C.POINTER(C.c_float)), width)
        assert n == np.float32(np.sort(x)[width // 4] * 2.0)
        assert np.array_equal(buf, np.sort(x))              # sorts its argument, like qsort on the copy
        a = L.ro_oracle_average(x.ctypes.data_as(C.POINTER(C.c_float)), width)
        assert a == np.float32(np.sum(x.astype(np.float64)) / width) or abs(a - x.mean()) < 1e-6
    x = np.array([1, 5, 2, 5, 5, 0], np.float32)
    assert L.ro_oracle_peak(x.ctypes.data_as(C.POINTER(C.c_float)), 6) == 4     # last maximum
    assert L.ro_oracle_peak(x.ctypes.data_as(C.POINTER(C.c_float)), 3) == 1


def test_fsm_states_and_event(oracle):
    """BolidRecorder::update's state machine (src/BolidRecorder.cpp:171-273)."""
    rate = np.float32(5.859375)
    f = oracle.BolidFsm(11, 29, rate, 48000, 10300.0, 10900.0)
    det = lambda d, mark: f.update(1.0, 5.0 if d else 1.0, 10500.0, mark)
    for i in range(20):
        assert not det(0, i + 1).fired and f.f.state == 0
    det(1, 21)                                               # INIT -> BOLID, start = mark - advance
    assert f.f.state == 1 and f.f.snap_start == 21 - 11 and f.f.snap_length == 22
    for i in range(5):
        det(1, 22 + i)
    assert f.f.duration == 6
    det(0, 27)                                               # BOLID -> ENDED, length += duration
    assert f.f.state == 2 and f.f.snap_length == 28 and f.f.duration == 1
    det(1, 28)                                               # ENDED -> BOLID again (jitter not reached)
    assert f.f.state == 1 and f.f.duration == 2
    det(0, 29)
    assert f.f.state == 2 and f.f.snap_length == 30
    fired = None
    for i in range(40):
        ev = det(0, 30 + i)
        if ev.fired:
            fired = (i, ev)
            break
    assert fired is not None
    i, ev = fired
    assert i == 27                                           # duration counts 2..29 -> fires when >= jitter
    assert ev.snap_start == 10 and ev.snap_length == 30
    assert ev.duration_s == np.float32(8.0) / rate           # (length - 2*advance) / fftRate
    assert ev.raw_length == int((30 / 5.0) * 48000)          # recorder's int fft rate (WaterfallBackend.cpp:29-32)
    assert (ev.fmin, ev.fmax) == (10500.0 - 150.0, 10500.0 + 150.0)
    assert f.f.state == 0


def test_ln_levels_match_the_viewers_numpy_formula(oracle):
    """fits2png is numpy: FN_LOG = numpy.log of the non-zero float32 pixels (:46), min/max of that (:476-477),
    default_color_fn (v - min) / (max - min) (:444-445) times 255 stored into a uint8 array (:495-497)."""
    rng = np.random.default_rng(46)
    image = (np.abs(rng.standard_normal((37, 615))) * 10.0 ** rng.uniform(-3, 4, (37, 1))).astype(np.float32)
    image[5, 7:19] = 0.0
    ln, u8, (mn, mx) = oracle.ln_levels(image)
    nz = image != 0
    data = np.log(image[nz])                                            # float32 in, float32 out
    assert data.dtype == np.float32
    assert np.abs(ln[nz] - data).max() <= 1e-6                          # numpy's SIMD logf vs libm: <= 2 ulp here
    assert abs(mn - data.min()) <= 1e-6 and abs(mx - data.max()) <= 1e-6
    assert np.all(np.isneginf(ln[~nz]))
    pixels = np.zeros(int(nz.sum()), np.uint8)
    row = (ln[nz] - np.float32(mn)) / (np.float32(mx) - np.float32(mn)) * 255
    assert row.dtype == np.float32
    pixels[:] = row                                                     # the viewer's store: C truncation
    assert np.array_equal(u8[nz], pixels) and (u8[~nz] == 0).all()
    assert u8.max() == 255 and u8.min() == 0
