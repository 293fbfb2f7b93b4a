"""GPU: the streaming half of the C ABI (ro_stft_push / flush / fetch) -- the Backend::process
boundary -- against the oracle, for every sample format a reference frontend produces."""
import os

import numpy as np
import pytest

from util import add_tone, noise_iq, rel_to_row_max

pytestmark = pytest.mark.gpu


def stream(ro, iq_chunks, bins, overlap, **kw):
    rows, recs, firsts = [], [], []
    with ro.Stft(bins=bins, overlap=overlap, **kw) as st:
        for c in iq_chunks:
            st.push(c)
            while True:
                first, r, rec = st.fetch(5)
                if len(r) == 0:
                    break
                rows.append(r); firsts.append(first)
                if rec is not None:
                    recs.append(rec)
        st.flush()
        while True:
            first, r, rec = st.fetch(1000)
            if len(r) == 0:
                break
            rows.append(r); firsts.append(first)
            if rec is not None:
                recs.append(rec)
        stats = st.stats()
    return (np.concatenate(rows) if rows else np.zeros((0, bins), np.float32)), recs, firsts, stats


@pytest.mark.parametrize("bins,overlap,chunk,batch", [(1024, 512, 1024, 3), (4096, 2048, 4096, 0),
                                                       (4096, 3072, 333, 7), (32768, 24576, 4096, 5),
                                                       (65536, 49152, 50000, 4),     # Bolidozor.json:45-46, one kernel
                                                       (262144, 196608, 100000, 5),  # the four-step form
                                                       (32728, 24546, 9999, 3)])     # src/BolidRecorder.h:35, chirp-z
def test_push_fetch_float32(ro, oracle, bins, overlap, chunk, batch):
    rng = np.random.default_rng(chunk)
    hop = bins - overlap
    T = bins + 12 * hop + 17
    iq = noise_iq(rng, T)
    chunks = [iq[i:i + chunk] for i in range(0, T, chunk)]
    got, _, firsts, stats = stream(ro, chunks, bins, overlap, max_batch_rows=batch)
    want = oracle.stft(iq, bins, overlap)
    assert got.shape == want.shape == (13, bins)
    assert rel_to_row_max(got, want) <= 1e-5
    assert firsts[0] == 0 and stats["samples_in"] == T and stats["rows_out"] == 13


def test_push_formats_agree(ro, oracle):
    """struct Complex (double), float32 and int16 inputs of the same values give identical rows."""
    bins, overlap = 2048, 1024
    rng = np.random.default_rng(1)
    f = add_tone(noise_iq(rng, bins * 5, 200.0), 5000.0, 9000.0)
    i16 = np.clip(np.rint(f), -32768, 32767).astype(np.int16)
    a, *_ = stream(ro, [i16], bins, overlap)
    b, *_ = stream(ro, [i16.astype(np.float32)], bins, overlap)
    c, *_ = stream(ro, [i16[:, 0].astype(np.float64) + 1j * i16[:, 1]], bins, overlap)
    d, *_ = stream(ro, [(i16[:, 0] + 1j * i16[:, 1]).astype(np.complex64)], bins, overlap)
    assert np.array_equal(a, b) and np.array_equal(a, c) and np.array_equal(a, d)
    assert rel_to_row_max(a, oracle.stft(i16.astype(np.float64), bins, overlap)) <= 1e-5


def test_fetch_columns_and_records(ro, oracle):
    bins, overlap = 32768, 24576
    b = oracle.bolid_bands(bins, 48000, overlap, 10300, 10900, 9000, 9600, 2, 5, 40)
    bands = ro.Bands(low_noise=b.low_noise, noise_width=b.noise_width, low_detect=b.low_detect,
                     detect_width=b.detect_width, avg_bins=b.avg_bins)
    rng = np.random.default_rng(2)
    iq = noise_iq(rng, bins + 9 * 8192)
    with ro.Stft(bins=bins, overlap=overlap, bands=bands, max_batch_rows=4) as st:
        assert st.push(iq) == 8                       # two full batches are ready, two rows still staged
        first, band, rec = st.fetch(3, first_col=22528, cols=2048)
        assert first == 0 and band.shape == (3, 2048) and rec.shape == (3,)
        assert st.flush() == 7
        first2, rest, rec2 = st.fetch(100)
        assert first2 == 3 and rest.shape == (7, bins)
        assert st.fetch(10)[1].shape[0] == 0
    want = oracle.stft(iq, bins, overlap)
    assert rel_to_row_max(band, want[:3, 22528:24576]) <= 1e-5 * want[:3].max() / want[:3, 22528:24576].max() + 1e-5
    assert rel_to_row_max(rest, want[3:]) <= 1e-5
    n, p, a = oracle.scan_rows(rest, b.low_noise, b.noise_width, b.low_detect, b.detect_width, b.avg_bins)
    assert np.array_equal(rec2["noise"], n) and np.array_equal(rec2["peak"], p) and np.array_equal(rec2["average"], a)


def test_short_and_empty_streams(ro):
    with ro.Stft(bins=1024, overlap=512) as st:
        assert st.push(np.zeros((0, 2), np.float32)) == 0
        assert st.push(np.zeros((1023, 2), np.float32)) == 0
        assert st.flush() == 0                        # less than one window: nothing comes out
        assert st.push(np.zeros((1, 2), np.float32)) == 0
        assert st.flush() == 1
        first, rows, _ = st.fetch(10)
        assert first == 0 and rows.shape == (1, 1024) and not rows.any()
        st.reset()
        assert st.flush() == 0


def test_int16_stream_stays_int16_and_format_switch(ro, oracle):
    """int16 samples (WAV) are staged and uploaded as int16 and widened by the kernel; a stream that switches to
    float samples mid-way is widened once on the host -- same rows either way."""
    bins, overlap = 4096, 3072
    rng = np.random.default_rng(12)
    i16 = rng.integers(-30000, 30000, size=(bins + 40 * 1024 + 5, 2), dtype=np.int16)
    want = oracle.stft(i16.astype(np.float64), bins, overlap)
    a, *_ = stream(ro, [i16[i:i + 1024] for i in range(0, len(i16), 1024)], bins, overlap, max_batch_rows=6)
    assert a.shape == want.shape and rel_to_row_max(a, want) <= 1e-5
    half = 17 * 1024 + 3
    mixed = [i16[:half], i16[half:].astype(np.float32)]
    b, *_ = stream(ro, mixed, bins, overlap, max_batch_rows=6)
    assert np.array_equal(a, b)


def test_tile_only_transport_and_timing(ro, oracle):
    """With a tile configured only its columns travel to the host; fetch outside it is refused.  The timing
    counters (FFTBackend::logProcessingTimes' counterpart) count calls and batches."""
    bins, overlap, hop = 32768, 24576, 8192
    rng = np.random.default_rng(13)
    iq = noise_iq(rng, bins + 21 * hop)
    tile = (23278, 615)
    with ro.Stft(bins=bins, overlap=overlap, tile=tile, max_batch_rows=4) as st:
        for i in range(0, len(iq), 4096):
            st.push(iq[i:i + 4096])
        st.flush()
        with pytest.raises(ro.StftError):
            st.fetch(1, first_col=0, cols=bins)
        first, band, _ = st.fetch(1000)
        assert first == 0 and band.shape == (22, 615)
        t = st.timing()
        assert t["push_calls"] == (len(iq) + 4095) // 4096 and t["batches"] == 6 and t["batch_rows"] == 22
        assert 0 < t["batch_gpu_ms_avg"] <= t["batch_gpu_ms_max"] and t["row_gpu_us_avg"] > 0
        assert t["push_ms_max"] >= t["push_ms_avg"] > 0 and t["fetch_calls"] == 1
        st.timing(reset=True)
        assert st.timing()["push_calls"] == 0
    want = oracle.stft(iq, bins, overlap)[:, tile[0]:tile[0] + tile[1]]
    full = oracle.stft(iq, bins, overlap)
    assert np.abs(band.astype(np.float64) - want).max() <= 1e-5 * full.max()


@pytest.mark.parametrize("bins,overlap,batch,slots,first_slot,tile", [(4096, 2048, 3, 8, 5, None), (32768, 24576, 4, 11, 0, None),
                                                                      (4096, 3072, 2, 7, 6, (1000, 300))])
def test_row_sink_receives_the_rows_in_ring_order(ro, oracle, bins, overlap, batch, slots, first_slot, tile):
    """ro_stft_set_row_sink: the rows land in the caller's ring by DMA -- row r in slot (first_slot + r) mod capacity, the
    batch that wraps in two pieces -- and ro_stft_fetch (rows_out = NULL) only reports them; a ring too small for two
    batches, a changed sink on a busy stream and rows_out with a sink are refused."""
    rng = np.random.default_rng(bins + batch)
    hop = bins - overlap
    R = 3 * slots + 1                                   # the ring wraps three times
    T = bins + (R - 1) * hop + 5
    iq = noise_iq(rng, T)
    want = oracle.stft(iq, bins, overlap)
    cols = tile[1] if tile else bins
    pinned = ro.PinnedArray(slots, cols + 3)            # page-locked (ro_pinned_alloc); a stride wider than the row
    ring = pinned.array
    ring[:] = np.nan
    small = ro.PinnedArray(2 * batch - 1, cols)
    with ro.Stft(bins=bins, overlap=overlap, max_batch_rows=batch, tile=tile) as st:
        with pytest.raises(ro.StftError):
            st.set_row_sink(small.array)
        # heap memory is refused before any copy engine sees it; page-locked memory of this runtime is accepted
        heap = np.zeros((slots, cols + 3), np.float32)
        assert ro.library().ro_pinned_check(heap.ctypes.data, heap.nbytes) == 0
        assert ro.library().ro_pinned_check(ring.ctypes.data, ring.nbytes) == 1
        # (a range that runs off the END of a pinned allocation is only refused when what follows is not page-locked too:
        # neighbouring hipHostMalloc blocks are; the check is about DMA safety, not about allocation boundaries)
        assert ro.library().ro_pinned_check(heap.ctypes.data + 4, heap.nbytes - 4) == 0
        with pytest.raises(ro.StftError) as e:
            st.set_row_sink(heap, first_slot)
        assert e.value.code == -1 and "page-locked" in str(e.value)
        st.set_row_sink(ring, first_slot)
        seen = 0
        for i in range(0, T, 3000):
            st.push(iq[i:i + 3000])
            while True:
                first, got, _ = st.fetch_records(batch)
                if got == 0:
                    break
                assert first == seen
                for r in range(first, first + got):     # a reported row is in place (and nobody has lapped it yet)
                    slot = (first_slot + r) % slots
                    ref = want[r, tile[0]:tile[0] + tile[1]] if tile else want[r]
                    assert np.abs(ring[slot, :cols] - ref).max() <= 1e-5 * want[r].max()
                    assert np.isnan(ring[slot, cols:]).all()                     # the padding of the stride is not written
                seen += got
        with pytest.raises(ro.StftError):
            st.set_row_sink(None)                       # samples are staged: not idle
        with pytest.raises(ro.StftError):
            st.fetch(1)                                 # rows_out with a sink
        st.flush()
        while True:
            first, got, _ = st.fetch_records(1000)
            if got == 0:
                break
            seen += got
        assert seen == R
        st.reset()
        # rows nobody fetches stay in the ring's slots: a push that would lap them is refused WHOLE -- nothing consumed,
        # nothing launched -- and the same samples pushed again in pieces with a fetch in between arrive without a gap
        big = iq[:bins + (slots + batch) * hop]
        before = st.stats()["samples_in"]
        with pytest.raises(ro.StftError) as e:
            st.push(big)
        assert "row sink full" in str(e.value) and "nothing was consumed" in str(e.value)
        assert st.stats()["samples_in"] == before and st.fetch_records(10 * slots)[1] == 0
        ring[:] = np.nan
        seen, step = 0, batch * hop + 7                   # pieces that complete a batch or so: always room after a fetch
        for at in range(0, len(big), step):
            st.push(big[at:at + step])
            while True:
                first, got, _ = st.fetch_records(10 * slots)
                if got == 0:
                    break
                assert first == seen
                for r in range(first, first + got):
                    ref = want[r, tile[0]:tile[0] + tile[1]] if tile else want[r]
                    assert np.abs(ring[(first_slot + r) % slots, :cols] - ref).max() <= 1e-5 * want[r].max()
                seen += got
        st.flush()
        first, got, _ = st.fetch_records(10 * slots)
        assert first == seen and seen + got == slots + batch + 1
        st.reset()
        st.set_row_sink(None)
        st.push(iq[:bins + hop])
        st.flush()
        first, rows, _ = st.fetch(10)                   # back to the copying path
        assert first == 0 and rows.shape[0] == 2 and np.abs(rows[1, :cols] - (want[1, tile[0]:tile[0] + tile[1]] if tile else want[1])).max() <= 1e-5 * want[1].max()
    del ring
    pinned.close()
    small.close()


@pytest.mark.parametrize("batch", [1, 2])
def test_four_step_sizes_back_to_back_with_a_sink(ro, torch_cuda, batch):
    """bins = 262144 hands Z from its column kernel to its row kernel through ONE scratch block per handle: batches of such
    a handle may never run side by side on the slots' streams (they are not graphed; every batch is ordered on the handle's
    stream).  Many small batches pushed in ONE call, no fetch in between, against the resident rows of the same samples."""
    torch = torch_cuda
    bins, overlap = 262144, 196608
    hop = bins - overlap
    R = 12
    rng = np.random.default_rng(batch)
    iq = noise_iq(rng, bins + (R - 1) * hop)
    d_iq = torch.from_numpy(iq).cuda()
    want = torch.empty((R, bins), dtype=torch.float32, device="cuda")
    with ro.Stft(bins=bins, overlap=overlap) as st:
        st.run_resident(d_iq, ro.RO_IQ_F32, iq.shape[0], 0, R, want, stream=torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
    want = want.cpu().numpy()
    pinned = ro.PinnedArray(R + 2, bins)
    ring = pinned.array
    ring[:] = np.nan
    with ro.Stft(bins=bins, overlap=overlap, max_batch_rows=batch) as st:
        st.set_row_sink(ring, 0)
        for rep in range(2):                                 # (the second time every slot has run before)
            st.push(iq)                                      # R rows = R / batch launches queued back to back
            st.flush()
            seen = 0
            while True:
                first, got, _ = st.fetch_records(100)
                if got == 0:
                    break
                seen += got
            assert seen == R
            assert np.array_equal(ring[:R], want), rep
            ring[:] = np.nan
            st.reset()
            st.set_row_sink(ring, 0)
    del ring
    pinned.close()


def test_one_hip_runtime_in_the_test_process(ro, torch_cuda):
    """The GPU tests import torch first (conftest), so libro_stft.so resolves `libamdhip64.so.7` against the copy torch
    has loaded: ONE HIP runtime in the process, device pointers, streams and page-locked memory mean the same thing to
    torch and to the product."""
    ro.library()
    import hostlib
    hostlib.host_library()                              # libro_host.so + the harness: linked against the same SONAME
    assert len(ro.hip_runtimes()) == 1, ro.hip_runtimes()
    ro.require_one_hip_runtime()


def test_rows_complete_counts_only_finished_batches(ro, oracle):
    """ro_stft_rows_complete: the rows ro_stft_fetch hands over without waiting.  Nothing before a batch is launched;
    after a flush and a wait, everything; and what it reports can be fetched at once and matches the oracle."""
    bins, overlap, batch = 4096, 2048, 5
    hop = bins - overlap
    rng = np.random.default_rng(21)
    R = 3 * batch + 2
    iq = noise_iq(rng, bins + (R - 1) * hop)
    want = oracle.stft(iq, bins, overlap)
    with ro.Stft(bins=bins, overlap=overlap, max_batch_rows=batch) as st:
        assert st.rows_complete() == 0
        st.push(iq[:bins + (batch - 2) * hop])               # not a whole batch yet: nothing launched
        assert st.rows_complete() == 0
        st.push(iq[bins + (batch - 2) * hop:])
        st.flush()
        import time
        for _ in range(2000):                                # the launches finish by themselves: poll, never fetch
            if st.rows_complete() == R:
                break
            time.sleep(0.001)
        assert st.rows_complete() == R
        first, rows, _ = st.fetch(4)                         # a partial fetch leaves the rest reported
        assert first == 0 and rows.shape[0] == 4 and st.rows_complete() == R - 4
        first, rest, _ = st.fetch(1000)
        assert first == 4 and rest.shape[0] == R - 4 and st.rows_complete() == 0
        got = np.concatenate([rows, rest])
    assert (np.abs(got.astype(np.float64) - want).max(axis=1) / want.max(axis=1)).max() <= 1e-5


@pytest.mark.parametrize("seed", [int(x) for x in os.environ.get("RO_SOAK_SEEDS", "1,2,3").split(",")])
def test_row_sink_soak_random_calls_and_fetches(ro, oracle, seed):
    """The streaming path with a row sink under calls of random size and fetches at random moments (seeded): pushes that
    would lap unfetched rows are refused whole and repeated after a fetch, only finished rows are fetched in between,
    and every row of the stream arrives once, in order, in its slot, equal to the oracle's."""
    rng = np.random.default_rng(1000 + seed)
    bins, overlap = 2048, 1536
    hop = bins - overlap
    batch = int(rng.integers(2, 7))
    slots = int(rng.integers(2 * batch, 4 * batch + 3))
    R = 40 * slots + int(rng.integers(0, slots))
    T = bins + (R - 1) * hop + int(rng.integers(0, hop))
    iq = noise_iq(rng, T)
    want = oracle.stft(iq, bins, overlap)
    pinned = ro.PinnedArray(slots, bins)
    ring = pinned.array
    ring[:] = np.nan
    seen = refused = 0

    def take(st, limit):
        nonlocal seen
        while limit > 0:
            first, got, _ = st.fetch_records(min(limit, 1 + int(rng.integers(0, 2 * batch))))
            if got == 0:
                break
            assert first == seen
            for r in range(first, first + got):
                assert np.abs(ring[r % slots] - want[r]).max() <= 1e-5 * want[r].max(), r
            seen += got
            limit -= got

    with ro.Stft(bins=bins, overlap=overlap, max_batch_rows=batch) as st:
        st.set_row_sink(ring, 0)
        at = 0
        while at < T:
            n = int(rng.integers(1, (slots + batch) * hop))
            piece = iq[at:at + n]
            try:
                st.push(piece)
                at += len(piece)
            except ro.StftError as e:
                assert e.code == -5 and "nothing was consumed" in str(e)
                refused += 1
                take(st, 10 ** 9)                              # everything in flight (this one waits), then the same piece again
                if st.stats()["rows_out"] - seen == 0 and len(piece) > (slots - batch) * hop:
                    piece = piece[:(slots - batch) * hop]      # a piece no empty ring could take: cut it
                st.push(piece)
                at += len(piece)
            if rng.random() < 0.6:
                take(st, st.rows_complete())                   # only what has finished: never waits
        for _ in range(100):                                   # a flush whose batch would lap unfetched rows is refused too:
            try:                                               # its samples stay staged, fetch and flush again
                st.flush()
                break
            except ro.StftError as e:
                assert e.code == -5 and "row sink full" in str(e)
                refused += 1
                take(st, 10 ** 9)
        take(st, 10 ** 9)
        assert seen == R == st.stats()["rows_out"]
    assert refused > 0                                         # the soak did meet the refusal
    del ring
    pinned.close()
