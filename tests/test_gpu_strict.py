"""RO_PRECISION_F64: the reference's arithmetic type (double window multiply, double transform, double sqrt, one
narrowing to the float row: src/FFTBackend.cpp:117-120,229-236, src/WaterfallBackend.cpp:492-505) on the GPU.

The bar here is the PER-BIN reading of north_star's "1e-5 relative on spectral magnitudes":
|row_gpu[k] - row_oracle[k]| <= 1e-5 * row_oracle[k] for every bin k of every row, which float32 butterflies miss on
bins 60 dB under a carrier (tests/test_gpu_stft.py::test_c3_carrier_60db prints that) and the FP64 mode meets with
seven orders of magnitude to spare.  The oracle's transform is an FP64 radix-2 FFT; two correct double transforms
differ by a few 1e-16 of the row maximum, so most float rows come out bit-identical.

Bins 256 ... 65536 run csrc/ro_f64reg.hip (the complex-double row in a CU's registers; below 4096 bins 2 ... 16 rows share
a workgroup), the larger powers of two the passes through HBM scratch (ro_kernels.hip)."""
import numpy as np
import pytest

from util import add_tone, noise_iq

pytestmark = pytest.mark.gpu

PER_BIN = 1e-5


def strict_rows(ro, torch, iq, bins, overlap, fmt=None, precision=None, **kw):
    fmt = ro.RO_IQ_F32 if fmt is None else fmt
    precision = ro.RO_PRECISION_F64 if precision is None else precision
    d_iq = torch.from_numpy(np.ascontiguousarray(iq)).cuda()
    rows = ro.row_count(iq.shape[0], bins, overlap)
    d_rows = torch.full((rows, bins), float("nan"), dtype=torch.float32, device="cuda")
    with ro.Stft(bins=bins, overlap=overlap, precision=precision, **kw) as st:
        st.run_resident(d_iq, fmt, iq.shape[0], 0, rows, d_rows, stream=torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
    return d_rows.cpu().numpy()


def per_bin(got, want):
    want = want.astype(np.float64)
    return np.abs(got.astype(np.float64) - want) / np.maximum(want, 1e-300)


def test_c3_carrier_60db_per_bin(ro, oracle, torch_cuda):
    """C3 signal model (sigma = 1 noise + CW 30 sigma): EVERY bin within 1e-5 of the oracle, relative to that bin."""
    bins, overlap = 32768, 24576
    rng = np.random.default_rng(0xC3)
    iq = add_tone(noise_iq(rng, bins + 7 * 8192), 10600.0, 30.0)
    got = strict_rows(ro, torch_cuda, iq, bins, overlap)
    want = oracle.stft(iq, bins, overlap)
    e = per_bin(got, want)
    print("FP64 mode, C3: per-bin rel err max %.3g, median %.3g; bit-identical floats %.4f"
          % (e.max(), np.median(e), (got == want).mean()))
    assert e.max() <= PER_BIN
    assert e.max() <= 2e-7                                   # i.e. at most one float32 ulp anywhere
    # the float32 path on the same input, for the record: norm-wise fine, per bin not
    with ro.Stft(bins=bins, overlap=overlap) as st:
        d_iq = torch_cuda.from_numpy(iq).cuda()
        d_rows = torch_cuda.empty((8, bins), dtype=torch_cuda.float32, device="cuda")
        st.run_resident(d_iq, ro.RO_IQ_F32, iq.shape[0], 0, 8, d_rows,
                        stream=torch_cuda.cuda.current_stream().cuda_stream)
        torch_cuda.cuda.synchronize()
    f = per_bin(d_rows.cpu().numpy(), want)
    print("float32 mode, same input: per-bin rel err max %.3g, frac > 1e-5: %.4f" % (f.max(), (f > PER_BIN).mean()))
    # ... while its rel-to-row-max error is 1e-7 (test_gpu_stft).  What the float32 mode does per bin on this input is
    # pinned, not just printed: measured 2.5 % of the bins over 1e-5 and a worst bin at 9.1e-4 (the weakest bins of a row
    # whose carrier is 60 dB above them); bench.py's `parity` carries the same two numbers next to the norm-wise one.
    assert (f > PER_BIN).mean() <= 0.03
    assert f.max() <= 2e-3


@pytest.mark.parametrize("bins,overlap", [(256, 128), (512, 0), (1024, 512), (2048, 1536), (4096, 2048),
                                          (8192, 6144), (16384, 12288), (32768, 0), (65536, 49152), (131072, 65536),
                                          (1048576, 0)])
def test_every_size_per_bin(ro, oracle, torch_cuda, bins, overlap):
    rng = np.random.default_rng(bins)
    hop = bins - overlap
    iq = add_tone(noise_iq(rng, bins + 5 * hop), 7000.0, 1000.0)          # 60 dB carrier again
    got = strict_rows(ro, torch_cuda, iq, bins, overlap)
    want = oracle.stft(iq, bins, overlap)
    assert got.shape == want.shape
    assert per_bin(got, want).max() <= 2e-7


def test_int16_gain_and_custom_window(ro, oracle, torch_cuda):
    bins, overlap = 4096, 3072
    rng = np.random.default_rng(3)
    i16 = rng.integers(-20000, 20000, size=(bins + 9 * 1024, 2), dtype=np.int16)
    w = rng.random(bins).astype(np.float32)
    got = strict_rows(ro, torch_cuda, i16, bins, overlap, fmt=ro.RO_IQ_I16, window_table=w, iq_gain=123.5)
    want = oracle.stft(i16.astype(np.float64), bins, overlap, w=w, gain=123.5)
    assert per_bin(got, want).max() <= 2e-7


def test_scan_records_and_tile_in_strict_mode(ro, oracle, torch_cuda):
    """records and band tile come from the separate kernels here (the fused epilogue is float32 only)"""
    from test_gpu_scan import json_bands
    bins, overlap, hop = 32768, 24576, 8192
    rng = np.random.default_rng(9)
    iq = noise_iq(rng, bins + 19 * hop)
    bands = json_bands(ro, oracle)
    torch = torch_cuda
    d_iq = torch.from_numpy(iq).cuda()
    R = 20
    rows = torch.empty((R, bins), dtype=torch.float32, device="cuda")
    recs = torch.zeros((R, 3), dtype=torch.float32, device="cuda")
    tile = torch.empty((R, 615), dtype=torch.float32, device="cuda")
    with ro.Stft(bins=bins, overlap=overlap, bands=bands, tile=(23278, 615), precision=ro.RO_PRECISION_F64) as st:
        st.run_resident(d_iq, ro.RO_IQ_F32, iq.shape[0], 0, R, rows, d_tile=tile, d_records=recs,
                        stream=torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
    got = rows.cpu().numpy()
    assert np.array_equal(tile.cpu().numpy(), got[:, 23278:23278 + 615])
    n, p, a = oracle.scan_rows(got, bands.low_noise, bands.noise_width, bands.low_detect, bands.detect_width,
                               bands.avg_bins)
    rec = recs.cpu().numpy().view(ro.capi.SCAN_DTYPE).reshape(-1)
    assert np.array_equal(rec["peak"], p) and np.array_equal(rec["noise"], n) and np.array_equal(rec["average"], a)


@pytest.mark.parametrize("bins,overlap,R", [(256, 255, 3001), (256, 128, 40003), (512, 0, 4999), (1024, 512, 20001),
                                            (2048, 1536, 7001), (4096, 2048, 5000), (8192, 6144, 1500),
                                            (16384, 12288, 900), (32768, 24576, 1500), (65536, 49152, 300)])
def test_register_form_many_rows_and_any_split(ro, oracle, torch_cuda, bins, overlap, R):
    """f64r_kernel is persistent: a workgroup takes many sub-rows in a row, the samples of the next one requested while
    the image of this one is still in LDS, and the D sub-rows of a stream row come from D workgroups; below 4096 bins a
    workgroup takes 4096 / bins rows at a time (row counts that leave the last tile part empty).  Enough rows that
    every workgroup loops many times: every bin of every row against the oracle, and the same bits whatever the launch's
    row range (uneven cuts: first rows of an XCD's run, a single row, a launch smaller than the grid)."""
    torch = torch_cuda
    hop = bins - overlap
    rng = np.random.default_rng(bins + R)
    iq = add_tone(noise_iq(rng, bins + (R - 1) * hop), 10600.0, 30.0)
    got = strict_rows(ro, torch, iq, bins, overlap)
    want = oracle.stft(iq, bins, overlap)
    assert per_bin(got, want).max() <= 2e-7
    d_iq = torch.from_numpy(iq).cuda()
    with ro.Stft(bins=bins, overlap=overlap, precision=ro.RO_PRECISION_F64) as st:
        s = torch.cuda.current_stream().cuda_stream
        cuts = ((0, 1), (1, 7), (8, R // 2), (8 + R // 2, R - 8 - R // 2))
        for first, n in cuts:
            part = torch.full((n, bins + 3), float("nan"), dtype=torch.float32, device="cuda")     # (a padded row stride too)
            st.run_resident(d_iq, ro.RO_IQ_F32, iq.shape[0], first, n, part, row_stride=bins + 3, stream=s)
            torch.cuda.synchronize()
            p = part.cpu().numpy()
            assert np.array_equal(p[:, :bins], got[first:first + n])
            assert np.isnan(p[:, bins:]).all()                  # nothing written beside the rows


def test_register_form_beside_another_kernel(ro, oracle, torch_cuda):
    """the persistent grid assumes nothing about residency or placement: the same launches while the float32 transform of
    another handle keeps the CUs busy on a second stream; every bin against the oracle."""
    torch = torch_cuda
    rng = np.random.default_rng(78)
    busy_bins, busy_rows = 32768, 4096
    busy_iq = torch.from_numpy(noise_iq(rng, busy_bins + 8192 * (busy_rows - 1))).cuda()
    busy_out = torch.empty((busy_rows, busy_bins), dtype=torch.float32, device="cuda")
    side = torch.cuda.Stream()
    with ro.Stft(bins=busy_bins, overlap=24576) as busy:
        for bins, overlap, R in ((32768, 24576, 600), (4096, 2048, 3000), (65536, 49152, 200), (1024, 512, 9001)):
            hop = bins - overlap
            iq = add_tone(noise_iq(rng, bins + (R - 1) * hop), 7000.0, 300.0)
            d_iq = torch.from_numpy(iq).cuda()
            out = torch.full((R, bins), float("nan"), dtype=torch.float32, device="cuda")
            torch.cuda.synchronize()
            with ro.Stft(bins=bins, overlap=overlap, precision=ro.RO_PRECISION_F64) as st:
                for _ in range(3):
                    busy.run_resident(busy_iq, ro.RO_IQ_F32, busy_iq.shape[0], 0, busy_rows, busy_out, stream=side.cuda_stream)
                st.run_resident(d_iq, ro.RO_IQ_F32, iq.shape[0], 0, R, out, stream=torch.cuda.current_stream().cuda_stream)
                for _ in range(2):
                    busy.run_resident(busy_iq, ro.RO_IQ_F32, busy_iq.shape[0], 0, busy_rows, busy_out, stream=side.cuda_stream)
                torch.cuda.synchronize()
            want = oracle.stft(iq, bins, overlap)
            assert per_bin(out.cpu().numpy(), want).max() <= 2e-7, bins


@pytest.mark.parametrize("bins", [256, 1024, 4096, 16384, 32768, 65536])
def test_register_form_int16_gain_and_custom_window(ro, oracle, torch_cuda, bins):
    """int16 frames (un-normalised, WAVStream), the I/Q gain (its own kernel instantiation) and a caller's window; magnitudes
    beyond float32's range of squares take the plain square root (the fast one works on float(re^2 + im^2))"""
    overlap = bins - bins // 8
    rng = np.random.default_rng(5)
    i16 = rng.integers(-20000, 20000, size=(bins + 40 * (bins // 8), 2), dtype=np.int16)
    w = rng.random(bins).astype(np.float32)
    got = strict_rows(ro, torch_cuda, i16, bins, overlap, fmt=ro.RO_IQ_I16, window_table=w, iq_gain=-77.25)
    want = oracle.stft(i16.astype(np.float64), bins, overlap, w=w, gain=-77.25)
    assert per_bin(got, want).max() <= 2e-7
    # huge and tiny samples: re^2 + im^2 leaves the range where its float is a normal number with headroom
    for scale in (1e17, 1e-17, 0.0):
        iq = (noise_iq(np.random.default_rng(7), bins + 3 * (bins // 8)) * scale).astype(np.float32)
        got = strict_rows(ro, torch_cuda, iq, bins, overlap)
        want = oracle.stft(iq, bins, overlap)
        assert np.isfinite(got).all()
        ok = want > 0
        assert per_bin(got[ok], want[ok]).max() <= 2e-7 if ok.any() else True
        assert np.array_equal(got[~ok], want[~ok])


@pytest.mark.parametrize("bins,overlap", [(256, 64), (1024, 512), (4096, 2048), (16384, 8192), (32768, 24576), (65536, 32768)])
def test_true_double_samples(ro, oracle, torch_cuda, bins, overlap):
    """struct Complex is two doubles (src/Backend.h:26-29) and the reference multiplies them as such (src/FFTBackend.cpp:
    229-232): with RO_PRECISION_F64 at these sizes RO_IQ_F64 samples reach the kernel un-narrowed, resident and through
    ro_stft_push.  The input is built so that narrowing it to float32 shows while two correct double transforms still
    agree per bin: a carrier 3000 x the noise -- float32 quantises the sum in steps of 2e-4 of the noise, 1e-4 of a noise
    bin's magnitude, and the rounding of a double transform stays 1e-10 of it."""
    torch = torch_cuda
    hop = bins - overlap
    R = 9
    T = bins + (R - 1) * hop
    n = np.arange(T, dtype=np.float64)
    rng = np.random.default_rng(bins)
    z = 3000.0 * np.exp(2j * np.pi * 0.11 * n) + (rng.standard_normal(T) + 1j * rng.standard_normal(T))
    iq = np.ascontiguousarray(np.stack([z.real, z.imag], axis=1))           # float64 [T, 2]
    want = oracle.stft(iq, bins, overlap)
    # resident doubles
    d_iq = torch.from_numpy(iq).cuda()
    d_rows = torch.full((R, bins), float("nan"), dtype=torch.float32, device="cuda")
    with ro.Stft(bins=bins, overlap=overlap, precision=ro.RO_PRECISION_F64) as st:
        st.run_resident(d_iq, ro.RO_IQ_F64, T, 0, R, d_rows, stream=torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        got = d_rows.cpu().numpy()
        assert per_bin(got, want).max() <= 2e-7
        # the same samples narrowed to float32 first: far outside the bar somewhere (the test can tell the difference)
        f32 = iq.astype(np.float32)
        st.run_resident(torch.from_numpy(f32).cuda(), ro.RO_IQ_F32, T, 0, R, d_rows, stream=torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        assert per_bin(d_rows.cpu().numpy(), want).max() > 1e-5
    # call by call: pieces of float64, with an int16 and a float32 piece in front (the staged samples are widened in place)
    lead = 1000
    head16 = rng.integers(-3000, 3000, size=(lead, 2), dtype=np.int16)
    head32 = rng.standard_normal((lead, 2)).astype(np.float32)
    full = np.concatenate([head16.astype(np.float64), head32.astype(np.float64), iq], axis=0)
    want2 = oracle.stft(full, bins, overlap)
    with ro.Stft(bins=bins, overlap=overlap, precision=ro.RO_PRECISION_F64, max_batch_rows=2) as st:
        st.push(head16)
        st.push(head32)
        for at in range(0, T, 7777):
            st.push(iq[at:at + 7777])
        st.flush()
        first, rows, _ = st.fetch(1000)
        assert first == 0 and rows.shape[0] == want2.shape[0]
        assert per_bin(rows, want2).max() <= 2e-7
    # a float32 handle, or an FP64 handle of another size, refuses resident doubles and narrows pushed ones as ever
    with ro.Stft(bins=bins, overlap=overlap) as st:
        with pytest.raises(ro.StftError) as e:
            st.run_resident(d_iq, ro.RO_IQ_F64, T, 0, R, d_rows)
        assert e.value.code == -2


@pytest.mark.parametrize("bins,overlap", [(512, 256), (1024, 512), (4096, 3072), (32768, 24576)])
def test_a_nan_sample_poisons_its_rows_only(ro, oracle, torch_cuda, bins, overlap):
    """fftw_execute and sqrt hand a NaN on to every bin of the rows that contain the sample (src/FFTBackend.cpp:229-236,
    src/WaterfallBackend.cpp:497-503) and to no other row -- also where several rows share a workgroup (bins < 4096) and
    where the magnitudes' fast square root decides once per wave"""
    hop = bins - overlap
    R = 37
    rng = np.random.default_rng(bins)
    iq = noise_iq(rng, bins + (R - 1) * hop)
    at = 11 * hop + bins // 3                                  # inside rows 11 - (bins // hop - 1) + ... 11
    iq[at, 1] = np.nan
    got = strict_rows(ro, torch_cuda, iq, bins, overlap)
    hit = np.array([r * hop <= at < r * hop + bins for r in range(R)])
    assert hit.sum() >= 2 and not hit.all()
    assert np.isnan(got[hit]).all()
    clean = iq.copy()
    clean[at, 1] = 0.0
    want = oracle.stft(clean, bins, overlap)
    assert per_bin(got[~hit], want[~hit]).max() <= 2e-7
