"""GPU parity of the complex-spectrum output (ro_stft_spectra_resident) against the oracle's FP64 spectrum of the
same windowed samples -- what fftw_execute leaves in out_ and FFTBackend::processFFT receives
(src/FFTBackend.cpp:229-236, src/FFTBackend.h:104).  Tolerance: |X_gpu - X_oracle| <= 1e-5 * max_k |X_oracle| per row
(the norm-wise reading of north_star's 1e-5, as for the magnitudes).  Also: the magnitude rows are exactly the
fft-shifted absolute values of these spectra computed the way the kernel computes them."""
import numpy as np
import pytest

from util import add_tone, noise_iq

pytestmark = pytest.mark.gpu

ALL_BINS = [256, 512, 1024, 2048, 4096, 8192, 16384, 32768]


def spectra_gpu(ro, torch, iq, bins, overlap, fmt=None, **kw):
    fmt = ro.RO_IQ_F32 if fmt is None else fmt
    d_iq = torch.from_numpy(np.ascontiguousarray(iq)).cuda()
    samples = iq.shape[0]
    rows = ro.row_count(samples, bins, overlap)
    spec = torch.full((rows, bins, 2), float("nan"), dtype=torch.float32, device="cuda")
    mag = torch.empty((rows, bins), dtype=torch.float32, device="cuda")
    s = torch.cuda.current_stream().cuda_stream
    with ro.Stft(bins=bins, overlap=overlap, **kw) as st:
        st.spectra_resident(d_iq, fmt, samples, 0, rows, spec, stream=s)
        st.run_resident(d_iq, fmt, samples, 0, rows, mag, stream=s)
        torch.cuda.synchronize()
        w = st.window
    return spec.cpu().numpy(), mag.cpu().numpy(), w


@pytest.mark.parametrize("bins", ALL_BINS)
def test_spectra_match_oracle(ro, oracle, torch_cuda, bins):
    rng = np.random.default_rng(bins + 1)
    overlap = bins // 2
    hop = bins - overlap
    nrows = 6 if bins >= 8192 else 20
    iq = add_tone(noise_iq(rng, bins + hop * (nrows - 1)), 7000.0, 5.0)
    spec, mag, w = spectra_gpu(ro, torch_cuda, iq, bins, overlap)
    z = iq[:, 0].astype(np.float64) + 1j * iq[:, 1].astype(np.float64)
    got = spec[..., 0].astype(np.float64) + 1j * spec[..., 1].astype(np.float64)
    for r in range(nrows):
        _, want = oracle.row_with_spectrum(z[r * hop:r * hop + bins], w)
        assert np.abs(got[r] - want).max() <= 1e-5 * np.abs(want).max(), (bins, r)
    # the waterfall row is |X| at (k + N/2) mod N (src/WaterfallBackend.cpp:492-505): same kernel, same values
    sq = spec[..., 0] * spec[..., 0] + spec[..., 1] * spec[..., 1]
    absx = np.sqrt(sq.astype(np.float32))
    shifted = np.roll(absx, bins // 2, axis=1)
    assert np.abs(shifted - mag).max() <= 2e-6 * mag.max()


@pytest.mark.parametrize("bins,nrows", [(65536, 3), (262144, 2)])
def test_large_spectra_match_oracle(ro, oracle, torch_cuda, bins, nrows):
    """complex spectra above 32768 (fold + the N = 32768 kernel in spectra mode + interleave), bin k at element k"""
    rng = np.random.default_rng(bins % 991)
    overlap = bins // 2
    hop = bins - overlap
    iq = add_tone(noise_iq(rng, bins + hop * (nrows - 1)), 7000.0, 5.0)
    spec, mag, w = spectra_gpu(ro, torch_cuda, iq, bins, overlap)
    z = iq[:, 0].astype(np.float64) + 1j * iq[:, 1].astype(np.float64)
    got = spec[..., 0].astype(np.float64) + 1j * spec[..., 1].astype(np.float64)
    for r in range(nrows):
        _, want = oracle.row_with_spectrum(z[r * hop:r * hop + bins], w)
        assert np.abs(got[r] - want).max() <= 1e-5 * np.abs(want).max(), (bins, r)
    absx = np.abs(got)
    assert np.abs(np.roll(absx, bins // 2, axis=1) - mag).max() <= 1e-5 * mag.max()


def test_spectra_int16_gain_stride_and_range(ro, oracle, torch_cuda):
    bins, overlap = 1024, 512
    rng = np.random.default_rng(3)
    i16 = rng.integers(-3000, 3000, size=(1024 * 5, 2)).astype(np.int16)
    torch = torch_cuda
    d_iq = torch.from_numpy(i16).cuda()
    rows = ro.row_count(i16.shape[0], bins, overlap)
    stride = bins + 8
    spec = torch.full((rows, stride, 2), 7.0, dtype=torch.float32, device="cuda")
    with ro.Stft(bins=bins, overlap=overlap, iq_gain=2.5) as st:
        st.spectra_resident(d_iq, ro.RO_IQ_I16, i16.shape[0], 2, 3, spec[2:], stride=stride,
                            stream=torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        w = st.window
        with pytest.raises(ro.StftError):
            st.spectra_resident(d_iq, ro.RO_IQ_I16, i16.shape[0], 0, rows + 1, spec)      # past the samples
    s = spec.cpu().numpy()
    assert (s[:2] == 7.0).all() and (s[5:] == 7.0).all() and (s[2:5, bins:] == 7.0).all()  # nothing else touched
    z = i16[:, 0].astype(np.float64) + 1j * i16[:, 1].astype(np.float64)
    for r in (2, 3, 4):
        _, want = oracle.row_with_spectrum(z[r * 512:r * 512 + bins], w, gain=2.5)
        got = s[r, :bins, 0].astype(np.float64) + 1j * s[r, :bins, 1]
        assert np.abs(got - want).max() <= 1e-5 * np.abs(want).max()


def test_spectra_unsupported_for_chirp_z_lengths(ro, torch_cuda):
    torch = torch_cuda
    bins = 32728
    d_iq = torch.zeros((bins, 2), dtype=torch.float32, device="cuda")
    spec = torch.empty((1, bins, 2), dtype=torch.float32, device="cuda")
    with ro.Stft(bins=bins, overlap=0) as st:
        with pytest.raises(ro.StftError) as e:
            st.spectra_resident(d_iq, ro.RO_IQ_F32, bins, 0, 1, spec)
        assert e.value.code == -2


def per_component(got, want):
    """worst |component error| of a complex row against the larger component of the bin (a double transform narrowed once
    per component is within half a float32 ulp of that)"""
    scale = np.maximum(np.abs(want.real), np.abs(want.imag))
    err = np.maximum(np.abs(got.real - want.real), np.abs(got.imag - want.imag))
    ok = scale > 1e-30
    return float((err[ok] / scale[ok]).max())


@pytest.mark.parametrize("bins,overlap,nrows", [(256, 192, 41), (512, 0, 19), (1024, 512, 23), (2048, 1024, 9), (4096, 2048, 20),
                                                (8192, 6144, 9), (16384, 8192, 7), (32768, 24576, 7), (65536, 49152, 5)])
def test_fp64_mode_spectra_are_the_double_transform(ro, oracle, torch_cuda, bins, overlap, nrows):
    """RO_PRECISION_F64 handles hand out the reference's fftw_complex row itself (src/FFTBackend.cpp:236), each component
    narrowed to float once: EVERY bin within a float32 ulp of the oracle's double spectrum, 60 dB under a carrier too
    (the float32 transform's bar is 1e-5 of the row's largest bin); the magnitude rows are |.| of the same values."""
    rng = np.random.default_rng(bins + 7)
    hop = bins - overlap
    iq = add_tone(noise_iq(rng, bins + hop * (nrows - 1)), 9100.0, 1000.0)
    spec, mag, w = spectra_gpu(ro, torch_cuda, iq, bins, overlap, precision=ro.RO_PRECISION_F64)
    assert np.isfinite(spec).all()
    z = iq[:, 0].astype(np.float64) + 1j * iq[:, 1].astype(np.float64)
    got = spec[..., 0].astype(np.float64) + 1j * spec[..., 1].astype(np.float64)
    worst = 0.0
    for r in range(nrows):
        _, want = oracle.row_with_spectrum(z[r * hop:r * hop + bins], w)
        worst = max(worst, per_component(got[r], want))
        m = np.roll(np.abs(want), bins // 2)
        assert (np.abs(mag[r] - m) <= 2e-7 * m).all()
    assert worst <= 1.2e-7, worst


def test_fp64_mode_spectra_int16_gain_doubles_stride_and_range(ro, oracle, torch_cuda):
    """the other sample formats (int16 with a gain; struct Complex's doubles), a padded stride and a row range in the middle
    of the stream: nothing else written"""
    torch = torch_cuda
    s = torch.cuda.current_stream().cuda_stream
    for bins, overlap in ((1024, 768), (16384, 12288), (32768, 16384)):
        hop = bins - overlap
        rng = np.random.default_rng(bins)
        total = 9
        i16 = rng.integers(-3000, 3000, size=(bins + hop * (total - 1), 2)).astype(np.int16)
        f64 = rng.standard_normal((bins + hop * (total - 1), 2)) * (1.0 + 2.0 ** -30)
        stride = bins + 8
        for samples, fmt, gain in ((i16, ro.RO_IQ_I16, 2.5), (f64, ro.RO_IQ_F64, 0.0)):
            spec = torch.full((total, stride, 2), 7.0, dtype=torch.float32, device="cuda")
            d_iq = torch.from_numpy(samples).cuda()
            with ro.Stft(bins=bins, overlap=overlap, iq_gain=gain, precision=ro.RO_PRECISION_F64) as st:
                st.spectra_resident(d_iq, fmt, samples.shape[0], 2, 5, spec[2:], stride=stride, stream=s)
                torch.cuda.synchronize()
                w = st.window
            out = spec.cpu().numpy()
            assert (out[:2] == 7.0).all() and (out[7:] == 7.0).all() and (out[2:7, bins:] == 7.0).all()
            z = samples[:, 0].astype(np.float64) + 1j * samples[:, 1].astype(np.float64)
            for r in range(2, 7):
                _, want = oracle.row_with_spectrum(z[r * hop:r * hop + bins], w, gain=gain)
                got = out[r, :bins, 0].astype(np.float64) + 1j * out[r, :bins, 1]
                assert per_component(got, want) <= 1.2e-7, (bins, fmt, r)


def test_fp64_mode_spectra_stop_at_65536_bins(ro, torch_cuda):
    torch = torch_cuda
    bins = 131072
    d_iq = torch.zeros((bins, 2), dtype=torch.float32, device="cuda")
    spec = torch.empty((1, bins, 2), dtype=torch.float32, device="cuda")
    with ro.Stft(bins=bins, overlap=0, precision=ro.RO_PRECISION_F64) as st:
        with pytest.raises(ro.StftError) as e:
            st.spectra_resident(d_iq, ro.RO_IQ_F32, bins, 0, 1, spec)
        assert e.value.code == -2
