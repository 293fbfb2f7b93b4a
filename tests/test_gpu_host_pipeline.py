"""GPU: the host-side mirror of the reference interface (Frontend -> Backend::process ->
Recorder::update) fed like the reference's frontends feed it, against the oracle's emulation of
Frontend::process + FFTBackend::process + BolidRecorder::update."""
import numpy as np
import pytest

from hostlib import HostPipeline, host_library
from util import add_chirp, add_tone, noise_iq, rel_to_row_max

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def _need_host():
    assert host_library() is not None, "tests/harness/libro_host_harness.so missing: run __graft_entry__.build()"


@pytest.mark.parametrize("bins,overlap,chunk,batch", [(1024, 512, 1024, 4), (1024, 512, 4096, 0),
                                                       (4096, 2048, 1000, 3), (2048, 0, 777, 1),
                                                       (32768, 24576, 4096, 8),
                                                       (65536, 49152, 4096, 4),      # Bolidozor.json:45-46
                                                       (32728, 24546, 1024, 3)])     # src/BolidRecorder.h:35
def test_stream_rows_times_and_marks(oracle, bins, overlap, chunk, batch):
    """WAVStream hands over 1024 samples per call, RawStream up to 4096, JACK arbitrary counts
    (src/WAVStream.cpp:190, src/RawStream.cpp:32, src/JackFrontend.cpp:35): rows, DataInfo and
    rawMark must not depend on the chunking except where the reference's timestamps do."""
    rng = np.random.default_rng(bins + chunk)
    hop = bins - overlap
    T = bins + 9 * hop + hop // 3
    iq = noise_iq(rng, T)
    z = iq[:, 0].astype(np.float64) + 1j * iq[:, 1].astype(np.float64)
    start = (1700000000, 250000)
    p = HostPipeline(bins, overlap, start=start, max_batch_rows=batch, snapshot_length=1)
    o = oracle.Stream(bins, overlap, start=start, raw_capacity_rows=p.raw_capacity())
    want_rows, want_info = [], []
    for i in range(0, T, chunk):
        p.process(z[i:i + chunk])
        r, inf = o.process(z[i:i + chunk])
        want_rows.append(r)
        want_info += inf
    p.end()
    assert p.error == ""
    want_rows = np.concatenate(want_rows)
    assert p.rows == want_rows.shape[0] == 10
    cap = p.ring_capacity()
    assert p.ring_mark() == 10 % cap
    got = np.stack([p.ring_row(p.ring_mark() - 10 + i) for i in range(10)])       # at(mark-1) = newest row
    assert rel_to_row_max(got, want_rows) <= 1e-5
    for i in range(10):
        assert p.row_info(i) == want_info[i], (i, p.row_info(i), want_info[i])
    # rawHandles_ is written one slot AHEAD of its row (src/WaterfallBackend.cpp:507, Appendix B-4)
    for i in range(10):
        m, s, u = p.raw_handle((i + 1) % cap)
        assert (m, s, u) == (want_info[i][3], want_info[i][1], want_info[i][2])
    p.close()


def test_default_batch_is_bounded_by_latency(oracle):
    """max_batch_rows = 0 (the default of every shipped config): rows reach the recorders within one second of
    stream time and well inside the raw ring, while the stream is running -- not only at endStream()."""
    bins, overlap, hop, fs = 4096, 3072, 1024, 48000
    rng = np.random.default_rng(4)
    T = 6 * fs                                                   # six seconds: 278 rows
    iq = noise_iq(rng, T)
    z = iq[:, 0].astype(np.float64) + 1j * iq[:, 1].astype(np.float64)
    p = HostPipeline(bins, overlap, max_batch_rows=0, snapshot_length=1)
    rate = fs / hop                                              # 46.875 rows/s
    assert 1 <= p.batch_rows() <= int(np.ceil(rate))
    assert p.batch_rows() * hop + bins <= p.raw_capacity()
    worst = 0
    for i in range(0, T, 4096):
        p.process(z[i:i + 4096])
        fed = min(i + 4096, T)
        complete = (fed - bins) // hop + 1 if fed >= bins else 0
        worst = max(worst, complete - p.rows)
    # never four batches behind: at most three batches are launched and not yet handed over (between two calls only rows
    # that have finished on the device are fetched -- the launches overlap the next calls -- and anything beyond is
    # waited for), plus what is being staged for the next launch
    # (more than one outstanding batch is allowed where that lag is a small part of both rings -- three in every shipped
    # config; this test's one-second snapshots make the rings short: the bound covers every case)
    assert worst < 4 * p.batch_rows(), (worst, p.batch_rows())
    p.end()
    assert p.rows == (T - bins) // hop + 1 and p.error == ""
    p.close()


def test_one_long_process_call_on_a_wrapped_ring(oracle):
    """A file replayed in ONE Backend::process call (the whole stream at once) on a ring that wraps many times: the backend
    hands the call over in pieces that stay inside the ring, rows land in the ring's slots by DMA ahead of the push() that
    publishes them, and what is in flight is bounded -- every row arrives, in order, with the stamps of the chunked run,
    and the ring ends holding the newest rows (the oracle's), not rows from a lap earlier or later."""
    bins, overlap, hop = 4096, 3072, 1024
    rng = np.random.default_rng(11)
    R = 1500
    T = bins + (R - 1) * hop + 17
    iq = noise_iq(rng, T)
    z = iq[:, 0].astype(np.float64) + 1j * iq[:, 1].astype(np.float64)
    start = (1700000000, 0)
    p = HostPipeline(bins, overlap, start=start, max_batch_rows=0, snapshot_length=1)
    cap = p.ring_capacity()
    assert R > 3 * cap, (R, cap)                                  # the ring wraps more than three times
    o = oracle.Stream(bins, overlap, start=start, raw_capacity_rows=p.raw_capacity())
    want_rows, want_info = o.process(z)                           # one call there too: the same time stamps
    p.process(z)
    p.end()
    assert p.error == "" and p.rows == R == want_rows.shape[0]
    assert p.ring_mark() == R % cap
    keep = min(cap - 1, 64)
    got = np.stack([p.ring_row(p.ring_mark() - keep + i) for i in range(keep)])
    assert rel_to_row_max(got, want_rows[R - keep:]) <= 1e-5
    for i in (0, 1, cap - 1, cap, 2 * cap + 3, R - 1):
        assert p.row_info(i) == want_info[i], (i, p.row_info(i), want_info[i])
    p.close()


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_pipeline_soak_random_call_sizes(oracle, seed):
    """Backend::process with calls of random size (one sample ... several batches' worth, seeded) at the latency-bound
    default batch -- graph-launched batches, rows fetched only when finished, up to two batches in flight, the ring
    wrapping: every row's DataInfo and raw mark equal the oracle's for the same chunking, and the ring ends holding the
    newest rows."""
    rng = np.random.default_rng(500 + seed)
    bins, overlap, hop = 2048, 1536, 512
    R = 1200 + int(rng.integers(0, 300))
    T = bins + (R - 1) * hop + int(rng.integers(0, hop))
    iq = noise_iq(rng, T)
    z = iq[:, 0].astype(np.float64) + 1j * iq[:, 1].astype(np.float64)
    start = (1700000000 + seed, 123456)
    p = HostPipeline(bins, overlap, start=start, max_batch_rows=0, snapshot_length=2)
    o = oracle.Stream(bins, overlap, start=start, raw_capacity_rows=p.raw_capacity())
    want_rows, want_info = [], []
    at = 0
    while at < T:
        n = int(rng.choice([1, 7, 100, 512, 1024, 4096, 20000, 90000]))
        n = max(1, int(rng.integers(1, n + 1)))
        chunk = z[at:at + n]
        p.process(chunk)
        r, inf = o.process(chunk)
        want_rows.append(r)
        want_info += inf
        at += len(chunk)
    p.end()
    assert p.error == ""
    want_rows = np.concatenate(want_rows)
    assert p.rows == R == want_rows.shape[0]
    cap = p.ring_capacity()
    keep = min(cap - 1, 200)
    got = np.stack([p.ring_row(p.ring_mark() - keep + i) for i in range(keep)])
    assert rel_to_row_max(got, want_rows[R - keep:]) <= 1e-5
    for i in range(R):
        assert p.row_info(i) == want_info[i], (i, p.row_info(i), want_info[i])
    p.close()


def test_bolid_detection_through_the_pipeline(oracle):
    """C4: chirps in noise through Frontend -> HipWaterfallBackend -> BolidRecorder; the events
    must equal the oracle's FSM driven by the oracle's FP64 rows."""
    bins, overlap, hop = 32768, 24576, 8192
    rng = np.random.default_rng(0xC4)
    rows = 120
    iq = noise_iq(rng, bins + (rows - 1) * hop)
    add_chirp(iq, 20 * hop, 2.0, 10800.0, -100.0, 3.0)
    add_chirp(iq, 70 * hop + 1234, 1.0, 10700.0, -100.0, 3.0)
    z = iq[:, 0].astype(np.float64) + 1j * iq[:, 1].astype(np.float64)
    p = HostPipeline(bins, overlap, max_batch_rows=16, snapshot_length=60)
    for i in range(0, len(z), 4096):
        p.process(z[i:i + 4096])
    p.end()
    assert p.error == "" and p.rows == rows
    ld, dw, ln, nw, adv, jit, avg = p.bands()
    assert (ld, dw, ln, nw, adv, jit, avg) == (23415, 410, 22528, 409, 11, 29, 27)
    cap = p.ring_capacity()
    assert cap == 352 * 8                                     # ceil(60 * 5.859375) * 8, 8 rows per 1 MiB chunk
    # oracle side
    want = oracle.stft(iq, bins, overlap)
    n, pk, a = oracle.scan_rows(want, ln, nw, ld, dw, avg)
    rate = oracle.lib().ro_oracle_fft_sample_rate(48000, bins, overlap)
    fsm = oracle.BolidFsm(adv, jit, rate, 48000, 10300.0, 10900.0)
    expect = []
    for r in range(rows):
        ev = fsm.update(n[r], a[r], oracle.lib().ro_oracle_bin_to_frequency(bins, 48000, ld + int(pk[r])),
                        (r + 1) % cap)
        if ev.fired:
            expect.append((r, ev.snap_start, ev.snap_length, ev.raw_length, ev.duration_s, ev.peak_freq,
                           ev.fmin, ev.fmax))
    got = [(e.row, e.start, e.length, e.rawLength, e.duration, e.peakFreq, e.fmin, e.fmax) for e in p.events()]
    assert len(expect) == 2, expect
    assert got == expect                                      # event rows, lengths, peak bins: bit-exact
    assert p.state() == fsm.f.state
    # noise / magnitude of the events agree to fp32-FFT accuracy
    margin = a.astype(np.float64) / (2 * n.astype(np.float64))
    assert np.abs(margin - 1).min() > 1e-3                    # no marginal row in this signal
    p.close()


def test_backend_without_scan_recorder_still_delivers_rows(oracle):
    """averageBinRange_ is 0 at N=1024 (src/BolidRecorder.cpp:102-104: the reference asserts);
    here the recorder simply asks for no scan and the rows still flow."""
    bins, overlap = 1024, 512
    rng = np.random.default_rng(5)
    iq = noise_iq(rng, 1024 * 6)
    i16 = np.clip(np.rint(add_tone(iq * 300, 10400.0, 8000.0)), -32768, 32767)
    z = i16[:, 0] + 1j * i16[:, 1]
    p = HostPipeline(bins, overlap, max_batch_rows=2, snapshot_length=1)
    for i in range(0, len(z), 1024):
        p.process(z[i:i + 1024])
    p.end()
    assert p.rows == 11 and p.events() == []
    want = oracle.stft(z, bins, overlap)
    got = np.stack([p.ring_row(p.ring_mark() - 11 + i) for i in range(11)])
    assert rel_to_row_max(got, want) <= 1e-5
    p.close()


def test_c1_wav_to_fits_end_to_end(oracle, tmp_path):
    """BASELINE config 1 (with the 2-channel stand-in for the reference's unreadable mono case, SURVEY.md
    §0-9): 16-bit I/Q WAV at 48 kHz -> WAVStream -> FFT bins 1024 / overlap 512 on the GPU ->
    SnapshotRecorder -> FITS; the image must equal the oracle's rows over [low_freq, hi_freq)."""
    import ctypes as C
    from test_host_cpu import read_fits, wav_bytes
    L = host_library()
    L.ro_host_wav_to_fits.restype = C.c_int64
    L.ro_host_wav_to_fits.argtypes = [C.c_char_p, C.c_int64, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float,
                                      C.c_float, C.c_char_p, C.c_char_p, C.c_char_p, C.c_int, C.c_char_p, C.c_int]
    rng = np.random.default_rng(0xC1)
    frames = 1024 * 200                                           # a multiple of 1024 frames
    f = add_tone(noise_iq(rng, frames, 300.0), 10400.0, 8000.0)
    i16 = np.clip(np.rint(f), -32768, 32767).astype(np.int16)
    payload = wav_bytes(i16, rate=48000)
    files, err = C.create_string_buffer(16384), C.create_string_buffer(1024)
    rows = L.ro_host_wav_to_fits(payload, len(payload), 1024, 512, 16, 1, 9000.0, 12000.0, str(tmp_path).encode(),
                                 b"c1test", files, 16384, err, 1024)
    assert err.value == b"", err.value
    want = oracle.stft(i16.astype(np.float64), 1024, 512)
    assert rows == want.shape[0] == (frames - 1024) // 512 + 1
    names = files.value.decode().split()
    lo = oracle.lib().ro_oracle_frequency_to_bin(1024, 48000, 9000.0)
    hi = oracle.lib().ro_oracle_frequency_to_bin(1024, 48000, 12000.0)
    snap_rows = int(np.ceil(1 * 93.75))                           # 94 rows per one-second snapshot
    assert len(names) == int(np.ceil(rows / snap_rows))
    got = np.concatenate([read_fits(n)[1] for n in names])
    assert got.shape == (rows, hi - lo)
    ref = want[:, lo:hi]
    assert np.abs(got - ref).max() <= 1e-5 * want.max()
    hdr = read_fits(names[1])[0]
    assert float(hdr["CRVAL1"]) == 9000.0 and abs(float(hdr["CDELT2"]) - 1000.0 / 93.75) < 1e-9
    # the tone sits where frequencyToBin puts it
    assert abs(int(want[5].argmax()) - oracle.lib().ro_oracle_frequency_to_bin(1024, 48000, 10400.0)) <= 1


def test_event_capture_files_band_image_and_raw_iq(oracle, tmp_path):
    """SURVEY.md §8(f) raw-I/Q event capture: a detected chirp makes BolidRecorder queue a snapshot with
    includeRawData (src/BolidRecorder.cpp:262-263); the band image must be the GPU rows the detector saw and the
    raw image the float32 samples behind them (src/WaterfallBackend.cpp:214-267)."""
    from test_host_cpu import read_fits
    bins, overlap, hop = 32768, 24576, 8192
    rng = np.random.default_rng(0xF3)
    rows = 90
    iq = noise_iq(rng, bins + (rows - 1) * hop)
    add_chirp(iq, 25 * hop, 2.0, 10800.0, -100.0, 3.0)
    z = iq[:, 0].astype(np.float64) + 1j * iq[:, 1].astype(np.float64)
    t0 = (1700000000, 250000)
    p = HostPipeline(bins, overlap, start=t0, max_batch_rows=8, snapshot_length=60, out_dir=tmp_path)
    for i in range(0, len(z), 4096):
        p.process(z[i:i + 4096])
    p.end()
    assert p.error == "" and p.rows == rows
    evs = p.events()
    assert len(evs) == 1
    e = evs[0]
    blid, raws = p.bolid_files(False), p.bolid_files(True)
    assert len(blid) == 1 and len(raws) == 1
    lo = oracle.lib().ro_oracle_frequency_to_bin(bins, 48000, 9000.0)
    hi = oracle.lib().ro_oracle_frequency_to_bin(bins, 48000, 12000.0)
    want = oracle.stft(iq, bins, overlap)
    hdr, data, _ = read_fits(blid[0])
    assert data.shape == (e.length, hi - lo)
    assert np.abs(data - want[e.start:e.start + e.length, lo:hi]).max() <= 1e-5 * want.max()
    # the raw window: handle[start] carries the mark stamped with row start-1 (rawHandles_ one slot ahead)
    hdr, data, _ = read_fits(raws[0])
    L = int(e.length / float(int(48000.0 / hop)) * 48000)
    first = (e.start * hop + 1) % p.raw_capacity()
    assert e.rawLength == L and data.shape == (L, 2)
    avail = min(L, len(iq) - first)                      # a capture may run past the end of a finished stream
    assert avail > L // 2 and np.array_equal(data[:avail], iq[first:first + avail].astype(np.float32))
    st = oracle.Stream(bins, overlap, 48000, t0)
    infos = []
    for i in range(0, len(z), 4096):
        infos += st.process(z[i:i + 4096])[1]
    _, sec, usec, _ = infos[e.start - 1]
    assert int(hdr["CRVAL2"]) == int(sec * 1000 + usec / 1000.0)
    p.close()
