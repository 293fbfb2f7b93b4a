"""BASELINE.json configs[4] ("C5") through the HIP path on one GPU: the 8-hour stream (1 382 400 000 samples,
168 747 rows of N = 32768 at 75 % overlap, C4's noise + chirp model), transformed

  (a) as one sequence of launches over the whole stream resident in HBM, and
  (b) as the 8 time-chunk shards of an 8-GPU run -- ro_shard_rows / ro_shard_samples, every shard REGENERATING its
      own slice (halo included) from the counter-based generator, its rows addressed from 0 -- whose band tiles and
      scan records are padded to equal blocks, laid out as an all-gather leaves them and stitched with the code the
      RCCL path uses (timeshard.pad_block / timeshard.stitch = ro_stitch_rows).

The band tile and the (n, p, a) stream of (b) must equal (a) bit for bit (rows are independent:
src/FFTBackend.cpp:211-257); the stitched records then drive the product's BolidRecorder state machine
(src/BolidRecorder.cpp:171-273) to exactly the oracle FSM's events, and every injected chirp is detected.
"""
import ctypes as C

import numpy as np
import pytest

from util import C5_SAMPLES, JSON_BOLID, c5_chirps, c5_slice, rel_to_row_max

pytestmark = pytest.mark.gpu

BINS, OVERLAP, HOP, FS = 32768, 24576, 8192, 48000
WORLD = 8
BLOCK = 16384                      # rows per launch


def run_blocks(ro, torch, st, iq, first_row, rows, tile, recs, scratch):
    s = torch.cuda.current_stream().cuda_stream
    for b in range(0, rows, BLOCK):
        n = min(BLOCK, rows - b)
        st.run_resident(iq, ro.RO_IQ_F32, iq.shape[0], first_row + b, n, scratch, d_tile=tile[b:b + n],
                        d_records=recs[b:b + n], stream=s)


def test_c5_eight_shards_equal_one_stream(ro, oracle, torch_cuda):
    from hostlib import BolidEvent, host_library
    from test_gpu_scan import json_bands
    torch = torch_cuda
    sh = ro.sharding()
    R = ro.row_count(C5_SAMPLES, BINS, OVERLAP)
    assert R == 168747
    bands = json_bands(ro, oracle)
    lo = ro.frequency_to_bin(BINS, FS, 9000.0)               # BolidRecorder's snapshot band (radio-observer.json:75-76)
    hi = ro.frequency_to_bin(BINS, FS, 12000.0)
    tile = (lo, hi - lo)
    assert tile == (22528, 2048)

    scratch = torch.empty((BLOCK, BINS), dtype=torch.float32, device="cuda")
    with ro.Stft(bins=BINS, overlap=OVERLAP, bands=bands, tile=tile) as st:
        # ---- (a) the whole stream, one handle, consecutive launches
        iq = c5_slice(torch, 0, C5_SAMPLES)
        tile_a = torch.empty((R, tile[1]), dtype=torch.float32, device="cuda")
        recs_a = torch.zeros((R, 3), dtype=torch.float32, device="cuda")
        run_blocks(ro, torch, st, iq, 0, R, tile_a, recs_a, scratch)
        torch.cuda.synchronize()
        assert bool(torch.isfinite(tile_a).all())

        # ---- (b) eight shards, each on its own regenerated slice
        max_rows = ro.shard_max_rows(R, WORLD)
        g_tile = torch.empty((WORLD * max_rows, tile[1]), dtype=torch.float32, device="cuda")
        g_recs = torch.empty((WORLD * max_rows, 3), dtype=torch.float32, device="cuda")
        seen = 0
        for g in range(WORLD):
            first, rows = sh.shard_rows(R, WORLD, g)
            s0, ns = sh.shard_samples(first, rows, BINS, HOP)
            assert first == seen and s0 == first * HOP and ns == (rows - 1) * HOP + BINS
            seen += rows
            mine = c5_slice(torch, s0, ns)
            assert torch.equal(mine, iq[s0:s0 + ns])                         # the generator is slice-invariant
            t_g = torch.empty((rows, tile[1]), dtype=torch.float32, device="cuda")
            r_g = torch.zeros((rows, 3), dtype=torch.float32, device="cuda")
            run_blocks(ro, torch, st, mine, 0, rows, t_g, r_g, scratch)      # shard-local row numbering
            torch.cuda.synchronize()
            g_tile[g * max_rows:(g + 1) * max_rows] = sh.pad_block(t_g, R, WORLD)
            g_recs[g * max_rows:(g + 1) * max_rows] = sh.pad_block(r_g, R, WORLD)
            del mine, t_g, r_g
        assert seen == R
        tile_b = sh.stitch(g_tile, R, WORLD)
        recs_b = sh.stitch(g_recs, R, WORLD)
        assert tile_b.shape == tile_a.shape and recs_b.shape == recs_a.shape
        assert torch.equal(tile_b.view(torch.int32), tile_a.view(torch.int32))
        assert torch.equal(recs_b.view(torch.int32), recs_a.view(torch.int32))
        # the host-memory form a C++ driver would use gives the same stream
        recs_host = ro.stitch_rows(g_recs.cpu().numpy(), R, WORLD)
        assert np.array_equal(recs_host.view(np.uint32), recs_a.cpu().numpy().view(np.uint32))

        # ---- oracle rows + oracle scan around one chirp of every duration, and in the last shard
        chirps = c5_chirps()
        assert len(chirps) == 960
        pick = [chirps[i][0] // HOP for i in (0, 1, 2, 3, 957)]
        rec_np = recs_host.view(ro.capi.SCAN_DTYPE).reshape(-1)
        for r0 in pick:
            r = np.arange(r0 - 2, r0 + 8)
            seg = iq[int(r[0]) * HOP:int(r[-1]) * HOP + BINS].cpu().numpy()
            want = oracle.stft(seg, BINS, OVERLAP)
            n, p, a = oracle.scan_rows(want, bands.low_noise, bands.noise_width, bands.low_detect,
                                       bands.detect_width, bands.avg_bins)
            got = tile_a[int(r[0]):int(r[-1]) + 1].cpu().numpy()
            assert rel_to_row_max(got, want[:, tile[0]:tile[0] + tile[1]]) <= 1e-5
            d_want = a.astype(np.float64) > 2.0 * n.astype(np.float64)
            d_got = rec_np["average"][r].astype(np.float64) > 2.0 * rec_np["noise"][r].astype(np.float64)
            assert np.array_equal(d_got, d_want), r0                      # detected rows: bit-exact
            assert np.array_equal(rec_np["peak"][r][d_want], p[d_want]), r0   # peak bins on detected rows
    del iq, scratch, g_tile, g_recs

    # ---- every injected chirp is detected, and nothing else is
    detect = rec_np["average"].astype(np.float64) > 2.0 * rec_np["noise"].astype(np.float64)
    covered = np.zeros(R, bool)
    for s0, n in chirps:
        r_first = max(0, (s0 - BINS) // HOP + 1)             # first row whose window holds a sample of the chirp
        r_last = min(R - 1, (s0 + n - 1) // HOP)
        covered[r_first:r_last + 1] = True
        assert detect[s0 // HOP:s0 // HOP + 4].any(), s0
    assert not detect[~covered].any()

    # ---- the stitched stream through the product's state machine == the oracle FSM
    rate = ro.fft_sample_rate(FS, BINS, OVERLAP)
    L = host_library()
    L.ro_host_bolid_replay.restype = C.c_int64
    L.ro_host_bolid_replay.argtypes = [C.c_int, C.c_int, C.c_int, C.c_float, C.c_float, C.c_float, C.c_float,
                                       C.c_double, C.c_double, C.c_float, C.c_void_p, C.c_int64,
                                       C.POINTER(BolidEvent), C.c_int]
    buf = (BolidEvent * 2048)()
    recs_c = np.ascontiguousarray(recs_host)
    n_ev = L.ro_host_bolid_replay(BINS, OVERLAP, FS, JSON_BOLID["min_detect"], JSON_BOLID["max_detect"],
                                  JSON_BOLID["min_noise"], JSON_BOLID["max_noise"], JSON_BOLID["advance_time"],
                                  JSON_BOLID["jitter_time"], JSON_BOLID["avg_freq_range"],
                                  C.c_void_p(recs_c.ctypes.data), R, buf, 2048)
    f = oracle.BolidFsm(11, 29, rate, FS, 10300.0, 10900.0)
    want_ev = []
    cap = 2816                                                # row ring: ceil(60 s x 5.859 rows/s) x 8 (a16)
    for i in range(R):
        ev = f.update(rec_np["noise"][i], rec_np["average"][i],
                      ro.bin_to_frequency(BINS, FS, bands.low_detect + int(rec_np["peak"][i])), (i + 1) % cap)
        if ev.fired:
            want_ev.append((i, ev.snap_start, ev.snap_length, ev.peak_freq))
    got_ev = [(buf[i].row, buf[i].start, buf[i].length, buf[i].peakFreq) for i in range(min(n_ev, 2048))]
    assert n_ev == len(want_ev) == len(got_ev)
    assert got_ev == want_ev
    # chirps are 30 s apart and an event closes 29 rows (5 s) after its last detected row: one event per chirp,
    # except that the stream ends before the last one's jitter time has passed or not -- at most one short
    assert len(chirps) - 1 <= n_ev <= len(chirps)
