"""GPU, BASELINE.json's full sizes (C3/C4: N=32768, 75 % overlap, 16384 rows; C2: N=4096, 50 %, 65536 rows), where
the FP64 oracle would take minutes: size-independent properties of the transform plus oracle rows at a few places.

  * shard invariance  -- any split of the row range gives the same bits (what the multi-GPU time chunks rely on)
  * shift invariance  -- row r of a stream == row 0 of the stream that starts r*hop samples later, bit for bit
  * scaling           -- rows(2^k x) == 2^k rows(x), bit for bit (every operation of the path is homogeneous)
  * Parseval          -- sum_k row[k]^2 == N * sum_n (w[n] |x[r hop + n]|)^2 to fp32 accuracy, EVERY row
  * scan records      -- bit-exact against the oracle's scan of the same rows, on a sample of rows
The other plans run at sizes that keep every CU busy with several workgroups at once -- the regime in which a
compiler-reordered barrier once corrupted rows that every small-size parity test got right.
"""
import numpy as np
import pytest

from util import add_chirp, add_tone, noise_iq, rel_to_row_max

pytestmark = pytest.mark.gpu


def synth(torch, samples, seed, carrier=None):
    g = torch.Generator(device="cuda")
    g.manual_seed(seed)
    iq = torch.randn((samples, 2), generator=g, device="cuda", dtype=torch.float32)
    if carrier:
        f, amp = carrier
        t = torch.arange(samples, device="cuda", dtype=torch.float64)
        ph = 2.0 * np.pi * f / 48000.0 * t
        iq[:, 0] += (amp * torch.cos(ph)).float()
        iq[:, 1] += (amp * torch.sin(ph)).float()
    return iq


def run(ro, torch, st, iq, first, rows, out):
    st.run_resident(iq, ro.RO_IQ_F32, iq.shape[0], first, rows, out, stream=torch.cuda.current_stream().cuda_stream)


@pytest.mark.parametrize("bins,overlap,R,seed", [(32768, 24576, 16384, 0xC3), (4096, 2048, 65536, 0xC2),
                                                 (16384, 12288, 16384, 5), (8192, 6144, 32768, 6),
                                                 (2048, 0, 65536, 7), (1024, 512, 131072, 0xC1), (512, 384, 65536, 8),
                                                 (256, 128, 262144, 9),
                                                 (65536, 49152, 4096, 10),      # Bolidozor.json:45-46: one-kernel form
                                                 (262144, 196608, 2600, 11),    # four-step form, six scratch blocks
                                                 (32728, 24546, 2500, 12)])     # chirp-z, three chunks
def test_full_size_properties(ro, oracle, torch_cuda, bins, overlap, R, seed):
    torch = torch_cuda
    hop = bins - overlap
    samples = bins + hop * (R - 1)
    iq = synth(torch, samples, seed, carrier=(10600.0, 30.0))
    rows = torch.empty((R, bins), dtype=torch.float32, device="cuda")
    with ro.Stft(bins=bins, overlap=overlap) as st:
        run(ro, torch, st, iq, 0, R, rows)
        torch.cuda.synchronize()
        assert bool(torch.isfinite(rows).all())

        # ---- shard invariance: three uneven shards (the second one starts mid-XCD-run) == one launch
        part = torch.empty((R // 2 + 5, bins), dtype=torch.float32, device="cuda")
        for first, n in ((0, 1237), (1237, R // 2 + 5), (1237 + R // 2 + 5, R - 1237 - R // 2 - 5)):
            part.fill_(float("nan"))
            run(ro, torch, st, iq, first, n, part)
            torch.cuda.synchronize()
            assert torch.equal(part[:n].view(torch.int32), rows[first:first + n].view(torch.int32)), (first, n)

        # ---- shift invariance: row r == row 0 of the stream that starts at sample r*hop
        one = torch.empty((1, bins), dtype=torch.float32, device="cuda")
        for r in (1, min(4097, R - 2), R - 1):
            sub = iq[r * hop:r * hop + bins]
            run(ro, torch, st, sub, 0, 1, one)
            torch.cuda.synchronize()
            assert torch.equal(one[0].view(torch.int32), rows[r].view(torch.int32)), r

        # ---- scaling by a power of two is exact
        iq8 = iq[:bins + hop * 63] * 8.0
        r8 = torch.empty((64, bins), dtype=torch.float32, device="cuda")
        run(ro, torch, st, iq8, 0, 64, r8)
        torch.cuda.synchronize()
        assert torch.equal(r8.view(torch.int32), (rows[:64] * 8.0).view(torch.int32))

        # ---- Parseval, every row: sum_k |X_k|^2 = N sum_n |w_n x_n|^2
        w = torch.from_numpy(st.window).cuda().double()
        p = (iq.double() ** 2).sum(dim=1)                                           # |x_n|^2
        if bins <= 32768 and bins & (bins - 1) == 0:
            want = torch.nn.functional.conv1d(p.view(1, 1, -1), (w * w).view(1, 1, -1), stride=hop).view(-1) * bins
        else:       # (long windows: the same sums as strided matrix-vector products, a few hundred rows at a time)
            want = torch.cat([torch.mv(p.as_strided((min(256, R - r0), bins), (hop, 1), r0 * hop), w * w)
                              for r0 in range(0, R, 256)]) * bins
        got = (rows.double() ** 2).sum(dim=1)
        assert want.shape[0] == R
        rel = ((got - want).abs() / want).max().item()
        assert rel < 2e-6, rel

    # ---- oracle rows at a few places of the run (norm-wise 1e-5, like the small-size parity tests)
    pick = [0, 1, R // 2, R - 1]
    host = iq.cpu().numpy()
    for r in pick:
        want_row = oracle.stft(host[r * hop:r * hop + bins], bins, overlap)[0]
        assert rel_to_row_max(rows[r].cpu().numpy()[None], want_row[None]) <= 1e-5


def test_full_size_scan_records_and_detection(ro, oracle, torch_cuda):
    """C4 at full size: noise + one chirp every 30 s; scan records of all 16384 rows against the oracle's scan of the
    same GPU rows (bit-exact), and the detector's decision a > 2n flags exactly the rows the chirps cover."""
    from test_gpu_scan import json_bands
    torch = torch_cuda
    bins, overlap, hop, R = 32768, 24576, 8192, 16384
    samples = bins + hop * (R - 1)
    rng = np.random.default_rng(0xC4)
    iq = noise_iq(rng, samples)
    starts = []
    t = 20.0
    durs = [0.5, 1.0, 2.0, 4.0]
    while t * 48000 + 4 * 48000 < samples:
        add_chirp(iq, int(t * 48000), durs[len(starts) % 4], 10800.0, -100.0, 3.0)
        starts.append(t)
        t += 30.0
    bands = json_bands(ro, oracle)
    d_iq = torch.from_numpy(iq).cuda()
    rows = torch.empty((R, bins), dtype=torch.float32, device="cuda")
    recs = torch.zeros((R, 3), dtype=torch.float32, device="cuda")
    with ro.Stft(bins=bins, overlap=overlap, bands=bands) as st:
        st.run_resident(d_iq, ro.RO_IQ_F32, samples, 0, R, rows, d_records=recs,
                        stream=torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
    got = recs.cpu().numpy().view(ro.capi.SCAN_DTYPE).reshape(-1)
    lo = min(bands.low_noise, bands.low_detect - bands.avg_bins)
    hi = max(bands.low_noise + bands.noise_width, bands.low_detect + bands.detect_width + bands.avg_bins)
    band = rows[:, lo:hi].cpu().numpy()                                     # the columns the scan can touch
    full = np.zeros((R, bins), np.float32)
    full[:, lo:hi] = band
    n, p, a = oracle.scan_rows(full, bands.low_noise, bands.noise_width, bands.low_detect, bands.detect_width,
                               bands.avg_bins)
    assert np.array_equal(got["peak"], p)
    assert np.array_equal(got["noise"].view(np.uint32), n.view(np.uint32))
    assert np.array_equal(got["average"].view(np.uint32), a.view(np.uint32))
    detect = got["average"].astype(np.float64) > 2.0 * got["noise"].astype(np.float64)
    # every chirp produces a run of detected rows that starts within 4 rows (one window length) of its onset
    for i, t0 in enumerate(starts):
        r0 = int(t0 * 48000) // hop
        assert detect[r0:r0 + 4].any(), (i, t0)
        assert not detect[max(0, r0 - 40):r0 - 4].any(), (i, t0)
    assert 0 < detect.sum() < R // 10


@pytest.mark.parametrize("bins,overlap,R", [(32768, 24576, 6000), (4096, 2048, 40000), (1024, 512, 100000)])
def test_int16_samples_under_load(ro, oracle, torch_cuda, bins, overlap, R):
    """The int16 (WAV) sample format has its own kernel instantiations: same shard-invariance check with every CU
    busy, plus oracle rows at three places."""
    torch = torch_cuda
    hop = bins - overlap
    samples = bins + hop * (R - 1)
    g = torch.Generator(device="cuda")
    g.manual_seed(bins)
    i16 = torch.randint(-20000, 20000, (samples, 2), generator=g, device="cuda", dtype=torch.int16)
    rows = torch.empty((R, bins), dtype=torch.float32, device="cuda")
    s = torch.cuda.current_stream().cuda_stream
    with ro.Stft(bins=bins, overlap=overlap) as st:
        st.run_resident(i16, ro.RO_IQ_I16, samples, 0, R, rows, stream=s)
        part = torch.empty((R // 3 + 1, bins), dtype=torch.float32, device="cuda")
        for first in (0, R // 3 + 1, 2 * (R // 3 + 1)):
            n = min(R // 3 + 1, R - first)
            st.run_resident(i16, ro.RO_IQ_I16, samples, first, n, part, stream=s)
            torch.cuda.synchronize()
            assert torch.equal(part[:n].view(torch.int32), rows[first:first + n].view(torch.int32)), first
    host = i16.cpu().numpy().astype(np.float64)
    for r in (0, R // 2, R - 1):
        want = oracle.stft(host[r * hop:r * hop + bins], bins, overlap)[0]
        assert rel_to_row_max(rows[r].cpu().numpy()[None], want[None]) <= 1e-5
