"""GPU parity: HIP STFT rows (through the C ABI) vs the FP64 oracle.

Tolerance (BASELINE.md "Parity bars"): |row_gpu - row_oracle| <= 1e-5 * max_k |row_oracle|
for every row -- the norm-wise reading of north_star's "1e-5 relative on spectral
magnitudes" (fp32 butterflies cannot meet a per-bin 1e-5 on bins 60 dB
under a carrier; per-bin statistics are printed for the record).
"""
import numpy as np
import pytest

from util import add_tone, noise_iq, rel_to_row_max

pytestmark = pytest.mark.gpu

TOL = 1e-5
ALL_BINS = [256, 512, 1024, 2048, 4096, 8192, 16384, 32768]


def gpu_rows(ro, torch, iq, bins, overlap, fmt=None, rows=None, first_row=0, **kw):
    """run the resident path on a host array; returns float32 [rows, bins]."""
    fmt = ro.RO_IQ_F32 if fmt is None else fmt
    d_iq = torch.from_numpy(np.ascontiguousarray(iq)).cuda()
    samples = iq.shape[0]
    total = ro.row_count(samples, bins, overlap)
    rows = total - first_row if rows is None else rows
    d_rows = torch.full((max(rows, 1), bins), float("nan"), dtype=torch.float32, device="cuda")
    with ro.Stft(bins=bins, overlap=overlap, **kw) as st:
        st.run_resident(d_iq, fmt, samples, first_row, rows, d_rows,
                        stream=torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
    return d_rows[:rows].cpu().numpy()


@pytest.mark.parametrize("bins", ALL_BINS)
def test_noise_rows_match_oracle(ro, oracle, torch_cuda, bins):
    rng = np.random.default_rng(bins)
    overlap = bins // 2
    nrows = 9
    iq = noise_iq(rng, bins + (nrows - 1) * (bins - overlap) + 5)
    got = gpu_rows(ro, torch_cuda, iq, bins, overlap)
    want = oracle.stft(iq, bins, overlap)
    assert got.shape == want.shape == (nrows, bins)
    assert np.isfinite(got).all()
    err = rel_to_row_max(got, want)
    print("bins=%d rel-to-row-max err %.3g" % (bins, err))
    assert err <= TOL


@pytest.mark.parametrize("bins,overlap,nrows", [(65536, 49152, 5), (131072, 0, 2), (262144, 131072, 3),
                                                 (524288, 262144, 3), (1048576, 786432, 2)])
def test_large_transforms(ro, oracle, torch_cuda, bins, overlap, nrows):
    """Bolidozor.json:45-46 (65536 / 49152) and Ionozor.json:27-28 (524288 / 262144): decimation in frequency on the
    N = 32768 kernel in one kernel up to 131072, the four-step pair of kernels (csrc/ro_fourstep.hip: N1 = 256, 512,
    1024 columns of 1024) through scratch above."""
    rng = np.random.default_rng(bins % 1000)
    hop = bins - overlap
    iq = add_tone(noise_iq(rng, bins + (nrows - 1) * hop + 3), 10600.0, 20.0, fs=96000)
    got = gpu_rows(ro, torch_cuda, iq, bins, overlap, sample_rate=96000)
    want = oracle.stft(iq, bins, overlap)
    assert got.shape == want.shape == (nrows, bins)
    err = rel_to_row_max(got, want)
    print("bins=%d rel-to-row-max err %.3g" % (bins, err))
    assert err <= TOL
    assert abs(int(got[0].argmax()) - ro.frequency_to_bin(bins, 96000, 10600.0)) <= 1


def test_large_transform_many_rows_cross_scratch_chunks(ro, oracle, torch_cuda):
    """more rows than one scratch block holds (512 rows at N=262144, the four-step form), and as many on the
    one-kernel form (N=65536), whose launch is not chunked."""
    torch = torch_cuda
    for bins in (262144, 65536):
        overlap, nrows = bins - 256, 1100
        rng = np.random.default_rng(8)
        iq = noise_iq(rng, bins + (nrows - 1) * 256)
        got = gpu_rows(ro, torch, iq, bins, overlap)
        assert np.isfinite(got).all()
        for r in (0, 511, 512, 1023, 1024, 1099):
            want = oracle.stft(iq[r * 256:r * 256 + bins], bins, overlap)[0]
            assert rel_to_row_max(got[r][None], want[None]) <= TOL
        del got


@pytest.mark.parametrize("bins", [65536, 262144, 524288, 1048576])
def test_large_transform_int16_gain_and_custom_window(ro, oracle, torch_cuda, bins):
    """both forms of the large transforms (one kernel at 65536; the four-step pair with 128, 64 and 32 columns per
    workgroup at 262144, 524288, 1048576) with WAV frames, I/Q gain (src/FFTBackend.cpp:78-79) and a caller's window
    table"""
    overlap, nrows = bins // 2, (4 if bins <= 262144 else 3)
    rng = np.random.default_rng(bins % 977)
    n = bins + (nrows - 1) * (bins - overlap) + 11
    i16 = rng.integers(-20000, 20000, size=(n, 2), dtype=np.int16)
    w = (0.25 + rng.random(bins)).astype(np.float32)
    got = gpu_rows(ro, torch_cuda, i16, bins, overlap, fmt=ro.RO_IQ_I16, window_table=w, iq_gain=37.5)
    want = oracle.stft(i16.astype(np.float32), bins, overlap, w=w, gain=37.5)
    assert got.shape == want.shape == (nrows, bins)
    assert rel_to_row_max(got, want) <= TOL


@pytest.mark.parametrize("bins,overlap,nrows", [(32768, 24576, 300), (262144, 131072, 40), (524288, 262144, 19)])
def test_cus_left_to_other_kernels_do_not_change_a_bit(ro, torch_cuda, bins, overlap, nrows):
    """ro_stft_config_t::spare_cus_per_xcd shrinks the persistent grids (N = 32768 kernel; both kernels of the four-step
    form): the rows are the same bits whatever the grid"""
    rng = np.random.default_rng(bins % 991)
    iq = noise_iq(rng, bins + (nrows - 1) * (bins - overlap))
    full = gpu_rows(ro, torch_cuda, iq, bins, overlap)
    for spare in (5, 16):
        assert np.array_equal(gpu_rows(ro, torch_cuda, iq, bins, overlap, spare_cus_per_xcd=spare), full), spare


@pytest.mark.parametrize("bins,overlap,nrows", [(4096, 2048, 50), (32768, 24576, 40), (65536, 49152, 12), (524288, 262144, 9)])
def test_row_stride_and_base_need_no_alignment(ro, torch_cuda, bins, overlap, nrows):
    """the 16-byte row stores of every form at a row stride and a base that are multiples of 4 bytes only: the same
    bits, and not a float outside the rows"""
    torch = torch_cuda
    rng = np.random.default_rng(bins % 983)
    iq = noise_iq(rng, bins + (nrows - 1) * (bins - overlap))
    ref = gpu_rows(ro, torch, iq, bins, overlap)
    d_iq = torch.from_numpy(np.ascontiguousarray(iq)).cuda()
    with ro.Stft(bins=bins, overlap=overlap) as st:
        for extra, off in ((1, 0), (3, 1), (5, 3)):
            stride = bins + extra
            buf = torch.full((nrows * stride + 8,), float("nan"), dtype=torch.float32, device="cuda")
            out = buf[off:off + nrows * stride]
            st.run_resident(d_iq, ro.RO_IQ_F32, iq.shape[0], 0, nrows, out.data_ptr(), row_stride=stride,
                            stream=torch.cuda.current_stream().cuda_stream)
            torch.cuda.synchronize()
            got = out.view(nrows, stride).cpu().numpy()
            assert np.array_equal(got[:, :bins], ref), (extra, off)
            assert np.isnan(got[:, bins:]).all() and np.isnan(buf[:off].cpu().numpy()).all(), (extra, off)


@pytest.mark.parametrize("bins,overlap,nrows", [(258, 0, 9), (1000, 600, 7), (12000, 9000, 5), (32728, 24546, 5),
                                                 (100000, 50000, 3), (524286, 262143, 2)])
def test_lengths_that_are_not_a_power_of_two(ro, oracle, torch_cuda, bins, overlap, nrows):
    """FFTW takes any N (src/FFTBackend.cpp:120; src/BolidRecorder.h:35 suggests 32728): even lengths run as a
    chirp-z transform on the power-of-two kernels (M = 512 ... 2^20: single pass, one-kernel large form, scratch form)"""
    assert ro.bins_supported(bins)
    rng = np.random.default_rng(bins % 1009)
    hop = bins - overlap
    iq = add_tone(noise_iq(rng, bins + (nrows - 1) * hop + 5), 10600.0, 20.0)
    got = gpu_rows(ro, torch_cuda, iq, bins, overlap)
    want = oracle.stft(iq, bins, overlap)
    assert got.shape == want.shape == (nrows, bins)
    assert np.isfinite(got).all()
    err = rel_to_row_max(got, want)
    print("bins=%d rel-to-row-max err %.3g" % (bins, err))
    assert err <= TOL
    assert abs(int(got[0].argmax()) - ro.frequency_to_bin(bins, 48000, 10600.0)) <= 1


def test_not_a_power_of_two_int16_gain_window_scan_and_tile(ro, oracle, torch_cuda):
    """the chirp-z path behind the same entry points: WAV frames, I/Q gain, a caller's window, band tile and scan records"""
    torch = torch_cuda
    bins, overlap, nrows = 32728, 24546, 6
    hop = bins - overlap
    rng = np.random.default_rng(77)
    i16 = rng.integers(-20000, 20000, size=(bins + (nrows - 1) * hop, 2), dtype=np.int16)
    w = (0.25 + rng.random(bins)).astype(np.float32)
    bands = ro.Bands(low_noise=100, noise_width=300, low_detect=16000, detect_width=500, avg_bins=27)
    d_iq = torch.from_numpy(i16).cuda()
    d_rows = torch.zeros((nrows, bins), dtype=torch.float32, device="cuda")
    d_tile = torch.zeros((nrows, 615), dtype=torch.float32, device="cuda")
    d_recs = torch.zeros((nrows, 3), dtype=torch.float32, device="cuda")
    with ro.Stft(bins=bins, overlap=overlap, window_table=w, iq_gain=37.5, bands=bands, tile=(20000, 615)) as st:
        st.run_resident(d_iq, ro.RO_IQ_I16, i16.shape[0], 0, nrows, d_rows, d_tile=d_tile, d_records=d_recs,
                        stream=torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
    rows = d_rows.cpu().numpy()
    want = oracle.stft(i16.astype(np.float32), bins, overlap, w=w, gain=37.5)
    assert rel_to_row_max(rows, want) <= TOL
    assert np.array_equal(d_tile.cpu().numpy(), rows[:, 20000:20615])
    got = d_recs.cpu().numpy().view(ro.capi.SCAN_DTYPE).reshape(-1)
    n, p, a = oracle.scan_rows(rows, 100, 300, 16000, 500, 27)
    assert np.array_equal(got["peak"], p) and np.array_equal(got["noise"], n) and np.array_equal(got["average"], a)


@pytest.mark.parametrize("bins,overlap", [(1024, 512), (4096, 2048), (32768, 24576), (32768, 0),
                                           (4096, 4095), (2048, 100), (32768, 32767)])
def test_overlap_variants(ro, oracle, torch_cuda, bins, overlap):
    rng = np.random.default_rng(7)
    hop = bins - overlap
    nrows = 5
    iq = noise_iq(rng, bins + (nrows - 1) * hop + (hop - 1))
    got = gpu_rows(ro, torch_cuda, iq, bins, overlap)
    want = oracle.stft(iq, bins, overlap)
    assert got.shape == want.shape
    assert rel_to_row_max(got, want) <= TOL


def test_c3_carrier_60db(ro, oracle, torch_cuda):
    """C3 signal model: sigma=1 noise + CW 30 sigma at +10.6 kHz."""
    bins, overlap = 32768, 24576
    rng = np.random.default_rng(0xC3)
    iq = add_tone(noise_iq(rng, bins + 7 * 8192), 10600.0, 30.0)
    got = gpu_rows(ro, torch_cuda, iq, bins, overlap)
    want = oracle.stft(iq, bins, overlap)
    err = rel_to_row_max(got, want)
    perbin = np.abs(got.astype(np.float64) - want) / np.maximum(want, 1e-30)
    print("C3: rel-to-row-max %.3g; per-bin rel err: median %.3g, p99 %.3g, max %.3g, frac>1e-5 %.3g"
          % (err, np.median(perbin), np.quantile(perbin, 0.99), perbin.max(), (perbin > 1e-5).mean()))
    assert err <= TOL
    # the carrier lands where FFTBackend::frequencyToBin says (src/FFTBackend.h:159-178)
    assert abs(int(got[0].argmax()) - ro.frequency_to_bin(bins, 48000, 10600.0)) <= 1


def test_int16_wav_like_input(ro, oracle, torch_cuda):
    """C1 signal model on the device's int16 path: un-normalised int16 I/Q
    (src/WAVStream.cpp:119-120): tone amp 8000 @ 10.4 kHz + sigma=300 noise."""
    bins, overlap = 1024, 512
    rng = np.random.default_rng(0xC1)
    n = 1024 * 12
    f = add_tone(noise_iq(rng, n, 300.0), 10400.0, 8000.0)
    i16 = np.clip(np.rint(f), -32768, 32767).astype(np.int16)
    got = gpu_rows(ro, torch_cuda, i16, bins, overlap, fmt=ro.RO_IQ_I16)
    want = oracle.stft(i16.astype(np.float64), bins, overlap)
    assert got.shape == want.shape == (23, bins)
    assert rel_to_row_max(got, want) <= TOL


def test_hann_and_custom_window(ro, oracle, torch_cuda):
    bins, overlap = 4096, 2048
    rng = np.random.default_rng(0xC2)
    iq = noise_iq(rng, bins * 3)
    got = gpu_rows(ro, torch_cuda, iq, bins, overlap, window=ro.RO_WINDOW_HANN)
    want = oracle.stft(iq, bins, overlap, w=oracle.window(bins, "hann"))
    assert rel_to_row_max(got, want) <= TOL
    w = rng.random(bins).astype(np.float32)
    got = gpu_rows(ro, torch_cuda, iq, bins, overlap, window_table=w)
    want = oracle.stft(iq, bins, overlap, w=w)
    assert rel_to_row_max(got, want) <= TOL


def test_iq_gain_is_added_to_q(ro, oracle, torch_cuda):
    """src/FFTBackend.cpp:78-79: out.imag = in.imag + gain."""
    bins, overlap = 2048, 1024
    rng = np.random.default_rng(5)
    iq = noise_iq(rng, bins * 4)
    got = gpu_rows(ro, torch_cuda, iq, bins, overlap, iq_gain=0.25)
    want = oracle.stft(iq, bins, overlap, gain=0.25)
    assert rel_to_row_max(got, want) <= TOL
    plain = oracle.stft(iq, bins, overlap)
    assert rel_to_row_max(got, plain) > 1e-3          # and it does change the result


def test_known_answers(ro, oracle, torch_cuda):
    """Analytic vectors: unit impulse -> |X[k]| = w[n0]; DC -> sum(w) in column bins/2."""
    bins = 1024
    w = oracle.window(bins).astype(np.float64)
    iq = np.zeros((bins, 2), np.float32)
    iq[300, 0] = 1.0
    got = gpu_rows(ro, torch_cuda, iq, bins, 0)
    assert np.allclose(got[0], w[300], rtol=2e-6, atol=0)
    iq = np.zeros((bins, 2), np.float32)
    iq[:, 0] = 1.0
    got = gpu_rows(ro, torch_cuda, iq, bins, 0)[0]
    assert got.argmax() == bins // 2                    # DC sits at column N/2 (src/WaterfallBackend.cpp:492-497)
    assert abs(got[bins // 2] - w.sum()) <= 1e-6 * w.sum()
    # a tone exactly on bin +37 shows up at column N/2 + 37, one on -37 at N/2 - 37
    for kbin in (37, -37):
        t = np.arange(bins) * (2 * np.pi * kbin / bins)
        iq = np.stack([np.cos(t), np.sin(t)], 1).astype(np.float32)
        got = gpu_rows(ro, torch_cuda, iq, bins, 0)[0]
        assert got.argmax() == bins // 2 + kbin


def test_first_row_and_partial_ranges(ro, oracle, torch_cuda):
    bins, overlap = 4096, 3072
    rng = np.random.default_rng(11)
    iq = noise_iq(rng, bins + 40 * 1024)
    want = oracle.stft(iq, bins, overlap)
    got = gpu_rows(ro, torch_cuda, iq, bins, overlap, first_row=13, rows=17)
    assert rel_to_row_max(got, want[13:30]) <= TOL
    # rows do not depend on where the launch starts: bit-identical to the full run
    full = gpu_rows(ro, torch_cuda, iq, bins, overlap)
    assert np.array_equal(full[13:30], got)


def test_resident_shape_checks(ro, torch_cuda):
    torch = torch_cuda
    bins, overlap = 1024, 512
    d_iq = torch.zeros((2048, 2), dtype=torch.float32, device="cuda")
    d_rows = torch.zeros((8, bins), dtype=torch.float32, device="cuda")
    with ro.Stft(bins=bins, overlap=overlap) as st:
        with pytest.raises(ro.StftError):          # 4 rows need 1024 + 3*512 = 2560 samples
            st.run_resident(d_iq, ro.RO_IQ_F32, 2048, 0, 4, d_rows)
        with pytest.raises(ro.StftError):          # stride too small
            st.run_resident(d_iq, ro.RO_IQ_F32, 2048, 0, 3, d_rows, row_stride=512)
        with pytest.raises(ro.StftError):          # records without scan bands
            st.run_resident(d_iq, ro.RO_IQ_F32, 2048, 0, 3, d_rows, d_records=d_rows)
        st.run_resident(d_iq, ro.RO_IQ_F32, 2048, 0, 3, d_rows)
        torch.cuda.synchronize()
    with pytest.raises(ro.StftError):
        ro.Stft(bins=1001)
    with pytest.raises(ro.StftError):
        ro.Stft(bins=1024, iq_phase_shift=3)


def test_tile_output_matches_rows(ro, torch_cuda):
    torch = torch_cuda
    bins, overlap = 32768, 24576
    rng = np.random.default_rng(3)
    iq = noise_iq(rng, bins + 5 * 8192)
    first, cols = 22528, 2048                      # 9-12 kHz recorder band (radio-observer.json:75-76)
    d_iq = torch.from_numpy(iq).cuda()
    d_rows = torch.zeros((6, bins), dtype=torch.float32, device="cuda")
    d_tile = torch.full((6, cols), -1.0, dtype=torch.float32, device="cuda")
    with ro.Stft(bins=bins, overlap=overlap, tile=(first, cols)) as st:
        st.run_resident(d_iq, ro.RO_IQ_F32, iq.shape[0], 0, 6, d_rows, d_tile=d_tile)
        torch.cuda.synchronize()
    assert torch.equal(d_tile, d_rows[:, first:first + cols])
