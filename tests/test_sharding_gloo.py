"""The N > 1 path on CPU: two gloo ranks shard one stream by time chunk, transform their rows,
all-gather band tiles + scan records and stitch them; the result must equal the single-process
run.  (On the GPUs the per-rank transform is the HIP kernel and the backend is RCCL; here the
oracle stands in for the kernel so the sharding / halo / stitch logic is what is tested.)"""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BINS, OVERLAP, HOP = 4096, 3072, 1024
TILE = (2900, 300)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _signal(total_rows):
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from util import add_chirp, noise_iq
    rng = np.random.default_rng(77)
    iq = noise_iq(rng, BINS + (total_rows - 1) * HOP)
    add_chirp(iq, 9 * HOP, 0.2, 10650.0, -200.0, 4.0)
    return iq


def _bands(O):
    b = O.bolid_bands(BINS, 48000, OVERLAP, 10300, 10900, 9000, 9600, 0.05, 0.1, 400)
    return b


def _worker(rank, world, port, total_rows, out_dir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import importlib
    import ro_oracle as O
    ro = importlib.import_module("radio-observer_amd")
    sh = ro.sharding()
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        iq = _signal(total_rows)
        first, rows = sh.shard_rows(total_rows, world, rank)
        s0, ns = sh.shard_samples(first, rows, BINS, HOP)
        mine = iq[s0:s0 + ns]                                   # this rank only ever touches its slice
        local = O.stft(mine, BINS, OVERLAP) if rows else np.zeros((0, BINS), np.float32)
        b = _bands(O)
        n, p, a = O.scan_rows(local, b.low_noise, b.noise_width, b.low_detect, b.detect_width, b.avg_bins)
        recs = np.zeros((rows, 3), np.float32)
        recs[:, 0] = n
        recs[:, 1] = p.view(np.float32)                         # ro_scan_record_t: 12 bytes, peak is an int32
        recs[:, 2] = a
        tile = torch.from_numpy(np.ascontiguousarray(local[:, TILE[0]:TILE[0] + TILE[1]]))
        g_tile, _ = sh.gather_rows(tile, total_rows)
        # the direct exchange (point-to-point transfers, rows landing stitched) must leave exactly the same band
        assert torch.equal(sh.gather_rows_direct(tile, total_rows), g_tile)
        assert torch.equal(sh.gather_rows_direct(torch.from_numpy(recs), total_rows), sh.gather_rows(torch.from_numpy(recs), total_rows)[0])
        stitch, work = sh.gather_rows(torch.from_numpy(recs), total_rows, async_op=True)
        work.wait()
        g_recs = stitch()
        np.savez(os.path.join(out_dir, "rank%d.npz" % rank), tile=g_tile.numpy(), recs=g_recs.numpy(),
                 first=first, rows=rows)
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,total_rows", [(2, 41), (2, 40), (3, 7)])
def test_sharded_equals_single(tmp_path, oracle, ro, world, total_rows):
    port = _free_port()
    mp.spawn(_worker, args=(world, port, total_rows, str(tmp_path)), nprocs=world, join=True)
    iq = _signal(total_rows)
    full = oracle.stft(iq, BINS, OVERLAP)
    b = _bands(oracle)
    n, p, a = oracle.scan_rows(full, b.low_noise, b.noise_width, b.low_detect, b.detect_width, b.avg_bins)
    seen = 0
    for r in range(world):
        z = np.load(os.path.join(str(tmp_path), "rank%d.npz" % r))
        # every rank ends up with the whole band, in row order, bit-identical to the 1-process run
        assert np.array_equal(z["tile"], full[:, TILE[0]:TILE[0] + TILE[1]])
        assert np.array_equal(z["recs"][:, 0], n) and np.array_equal(z["recs"][:, 2], a)
        assert np.array_equal(z["recs"][:, 1].copy().view(np.int32), p)
        assert int(z["first"]) == seen
        seen += int(z["rows"])
    assert seen == total_rows
    # the stitched (n, p, a) stream drives the state machine exactly like the unsharded one
    if total_rows > 20:                                          # the chirp starts at row 9
        assert (a.astype(np.float64) > 2.0 * n.astype(np.float64)).any()


def _worker_c5(rank, world, port, total_rows, out_dir):
    """every rank holds its shard of C5's 168 747 scan records (three words a row: the row's index and two functions of
    it) and of a narrow tile; the direct exchange and the gather to one rank must leave them stitched on the receiver"""
    sys.path.insert(0, ROOT)
    import importlib
    ro = importlib.import_module("radio-observer_amd")
    sh = ro.sharding()
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        first, rows = sh.shard_rows(total_rows, world, rank)
        idx = torch.arange(first, first + rows, dtype=torch.int64)
        recs = torch.stack([idx.to(torch.float32), (idx % 977).to(torch.float32), (idx * 3 % 1013).to(torch.float32)], dim=1)
        want_idx = torch.arange(total_rows, dtype=torch.int64)
        want = torch.stack([want_idx.to(torch.float32), (want_idx % 977).to(torch.float32), (want_idx * 3 % 1013).to(torch.float32)], dim=1)
        got = sh.gather_rows_direct(recs, total_rows)
        ok = torch.equal(got, want)
        padded, _ = sh.gather_rows(recs, total_rows)                    # the equal-block all-gather + stitch: the same rows
        ok = ok and torch.equal(padded, want)
        rooted, _ = sh.gather_rows(recs, total_rows, root=world - 1)
        ok = ok and ((rooted is None) if rank != world - 1 else torch.equal(rooted, want))
        open(os.path.join(out_dir, "ok%d" % rank), "w").write("1" if ok else "0")
        dist.barrier()
    finally:
        dist.destroy_process_group()


def test_c5_split_over_eight_ranks(tmp_path, ro):
    """BASELINE config 5's geometry on eight gloo ranks: 168 747 rows = 8 x 21 093 + 3, so three shards are a row longer --
    the direct exchange (ro_direct_schedule step by step), the padded all-gather and the gather to one rank all end with
    the rows in stream order on their receivers."""
    world, total_rows = 8, 168747
    port = _free_port()
    mp.spawn(_worker_c5, args=(world, port, total_rows, str(tmp_path)), nprocs=world, join=True)
    for r in range(world):
        assert open(os.path.join(str(tmp_path), "ok%d" % r)).read() == "1", r


def test_shard_arithmetic(ro):
    sh = ro.sharding()
    for total in (0, 1, 7, 8, 168747):
        for world in (1, 2, 3, 8):
            shards = sh.all_shards(total, world)
            assert shards[0][0] == 0 and sum(r for _, r in shards) == total
            for g in range(1, world):
                assert shards[g][0] == shards[g - 1][0] + shards[g - 1][1]
            assert max(r for _, r in shards) - min(r for _, r in shards) <= 1
    # C5: 8 h at 48 kHz, N=32768/75 %: 168 747 rows, ~21 093 per GPU, 192 KiB of halo
    assert ro.row_count(8 * 3600 * 48000, 32768, 24576) == 168747
    first, rows = sh.shard_rows(168747, 8, 3)
    s0, ns = sh.shard_samples(first, rows, 32768, 8192)
    assert rows in (21093, 21094) and ns == (rows - 1) * 8192 + 32768
    nxt, _ = sh.shard_rows(168747, 8, 4)
    assert s0 + ns - nxt * 8192 == 32768 - 8192                 # overlap with the next shard = halo
