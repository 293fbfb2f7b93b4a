"""GPU parity of the band-scan kernel vs BolidRecorder::noise/peak/average as restated by the
oracle (src/BolidRecorder.cpp:121-132, :313-347).  Integer/index work and exact float results:
the bar is bit-exact on all three record fields."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def scan_gpu(ro, torch, rows, bands):
    bins = rows.shape[1]
    d_rows = torch.from_numpy(rows).cuda()
    d_recs = torch.zeros((rows.shape[0], 3), dtype=torch.float32, device="cuda")
    with ro.Stft(bins=bins, overlap=0, bands=bands) as st:
        st.scan_resident(d_rows, rows.shape[0], d_recs, stream=torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
    return d_recs.cpu().numpy().view(ro.capi.SCAN_DTYPE).reshape(-1)


def check(ro, oracle, torch, rows, bands):
    got = scan_gpu(ro, torch, rows, bands)
    n, p, a = oracle.scan_rows(rows, bands.low_noise, bands.noise_width, bands.low_detect,
                               bands.detect_width, bands.avg_bins)
    assert np.array_equal(got["peak"], p)
    assert np.array_equal(got["noise"].view(np.uint32), n.view(np.uint32))
    assert np.array_equal(got["average"].view(np.uint32), a.view(np.uint32))


def json_bands(ro, oracle, bins=32768, overlap=24576):
    b = oracle.bolid_bands(bins, 48000, overlap, 10300, 10900, 9000, 9600, 2, 5, 40)
    return ro.Bands(low_noise=b.low_noise, noise_width=b.noise_width, low_detect=b.low_detect,
                    detect_width=b.detect_width, avg_bins=b.avg_bins)


def test_scan_random_rows_radio_observer_json(ro, oracle, torch_cuda):
    rng = np.random.default_rng(1)
    bands = json_bands(ro, oracle)
    assert (bands.low_detect, bands.detect_width, bands.low_noise, bands.noise_width, bands.avg_bins) == \
        (23415, 410, 22528, 409, 27)
    rows = np.abs(rng.standard_normal((257, 32768))).astype(np.float32) * 100
    check(ro, oracle, torch_cuda, rows, bands)


def test_scan_ties_take_last_index(ro, oracle, torch_cuda):
    """peak() uses >= so equal maxima resolve to the highest index (src/BolidRecorder.cpp:329-332)."""
    rng = np.random.default_rng(2)
    bands = json_bands(ro, oracle)
    rows = rng.random((64, 32768)).astype(np.float32)
    for r in range(64):
        idx = rng.choice(bands.detect_width, size=1 + r % 5, replace=False)
        rows[r, bands.low_detect + idx] = 7.5
    got = scan_gpu(ro, torch_cuda, rows, bands)
    for r in range(64):
        band = rows[r, bands.low_detect:bands.low_detect + bands.detect_width]
        assert got["peak"][r] == np.flatnonzero(band == band.max()).max()
    check(ro, oracle, torch_cuda, rows, bands)
    # a row of all-equal values: last index wins, quartile is that value
    rows[:] = 3.25
    got = scan_gpu(ro, torch_cuda, rows, bands)
    assert (got["peak"] == bands.detect_width - 1).all() and (got["noise"] == 6.5).all()


def test_scan_duplicates_and_quartile_index(ro, oracle, torch_cuda):
    """noise() = sorted[len/4] * 2 with heavy duplication in the band (src/BolidRecorder.cpp:313-317)."""
    rng = np.random.default_rng(3)
    bands = json_bands(ro, oracle)
    rows = rng.integers(0, 6, size=(128, 32768)).astype(np.float32)     # many equal values, zeros
    check(ro, oracle, torch_cuda, rows, bands)
    rows = (rng.standard_normal((128, 32768)) * 1e-3).astype(np.float32)  # signed values too
    check(ro, oracle, torch_cuda, rows, bands)


@pytest.mark.parametrize("bins,nw,dw,avg", [(4096, 51, 51, 3), (4096, 1, 1, 1), (32768, 1024, 777, 27),
                                           (32768, 1025, 64, 2), (32768, 5000, 3000, 101),
                                           (1024, 63, 65, 5), (32768, 16384, 16384, 27),
                                           (32768, 4096, 4096, 5), (32768, 4097, 100, 5), (32768, 8192, 8193, 3),
                                           (524288, 3277, 6554, 55), (524288, 20000, 9000, 7)])
def test_scan_band_shapes(ro, oracle, torch_cuda, bins, nw, dw, avg):
    """bands kept in registers (scan_kernel<E>: up to 1024, 4096, 8192 columns) and bands re-read in batches, odd
    widths, the limits between the forms, Ionozor-sized rows."""
    rng = np.random.default_rng(nw * 7 + dw)
    ln = int(rng.integers(0, bins - nw + 1))
    ld = int(rng.integers(avg, bins - dw - avg + 1))
    bands = ro.Bands(low_noise=ln, noise_width=nw, low_detect=ld, detect_width=dw, avg_bins=avg)
    rows = np.abs(rng.standard_normal((33, bins))).astype(np.float32)
    check(ro, oracle, torch_cuda, rows, bands)


@pytest.mark.parametrize("dw", [3000, 8192, 20000])
def test_scan_ties_and_duplicates_in_wide_bands(ro, oracle, torch_cuda, dw):
    """the >= rule (last index of the maximum) and the quartile among many equal values where a lane holds 47, 128 or
    a batched 313 columns of the band"""
    rng = np.random.default_rng(dw)
    bins = 65536
    bands = ro.Bands(low_noise=1000, noise_width=dw, low_detect=30000, detect_width=dw, avg_bins=9)
    rows = rng.integers(0, 9, size=(24, bins)).astype(np.float32)
    for r in range(24):
        rows[r, 30000 + rng.choice(dw, size=1 + r % 7, replace=False)] = 11.0
    got = scan_gpu(ro, torch_cuda, rows, bands)
    for r in range(24):
        band = rows[r, 30000:30000 + dw]
        assert got["peak"][r] == np.flatnonzero(band == 11.0).max()
    check(ro, oracle, torch_cuda, rows, bands)


def test_scan_average_window_below_detect_band(ro, oracle, torch_cuda):
    """average() starts at lowDetect + p - avg/2, which lies below the detect band when the
    peak is its first bin (src/BolidRecorder.cpp:126-132)."""
    bands = json_bands(ro, oracle)
    rng = np.random.default_rng(4)
    rows = rng.random((16, 32768)).astype(np.float32)
    rows[:, bands.low_detect] = 50.0                       # peak index 0
    got = scan_gpu(ro, torch_cuda, rows, bands)
    assert (got["peak"] == 0).all()
    check(ro, oracle, torch_cuda, rows, bands)


def test_fused_records_equal_standalone_scan(ro, oracle, torch_cuda):
    """records produced by run_resident == scan of the rows it wrote == oracle scan of those rows."""
    torch = torch_cuda
    from util import noise_iq
    bins, overlap = 32768, 24576
    bands = json_bands(ro, oracle)
    iq = noise_iq(np.random.default_rng(9), bins + 31 * 8192)
    d_iq = torch.from_numpy(iq).cuda()
    d_rows = torch.zeros((32, bins), dtype=torch.float32, device="cuda")
    d_recs = torch.zeros((32, 3), dtype=torch.float32, device="cuda")
    with ro.Stft(bins=bins, overlap=overlap, bands=bands) as st:
        st.run_resident(d_iq, ro.RO_IQ_F32, iq.shape[0], 0, 32, d_rows, d_records=d_recs)
        # the handle's own stream was used (stream=None): synchronise the device before reading
        torch.cuda.synchronize()
    rows = d_rows.cpu().numpy()
    got = d_recs.cpu().numpy().view(ro.capi.SCAN_DTYPE).reshape(-1)
    n, p, a = oracle.scan_rows(rows, bands.low_noise, bands.noise_width, bands.low_detect,
                               bands.detect_width, bands.avg_bins)
    assert np.array_equal(got["peak"], p) and np.array_equal(got["noise"], n) and np.array_equal(got["average"], a)
