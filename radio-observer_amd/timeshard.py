"""Time-chunk sharding of one sample stream over N GPUs (one process per GPU).

Rows are independent: row r needs samples [r*hop, r*hop + bins) and nothing else
(src/FFTBackend.cpp:211-257), so rank g of G takes the contiguous rows
[floor(g*R/G), floor((g+1)*R/G)) and the samples under them -- neighbouring shards overlap
by the bins-hop samples of halo.  No collective runs inside the transform; the only exchange
is the stitch at the end of a chunk: an all-gather (RCCL on GPUs, gloo in the CPU tests) of
each rank's band tile [rows_g x tile_cols] and scan records [rows_g x 3 words], after which
every rank holds the waterfall band and the (n, p, a) stream in row order -- what the
reference's FITS writer (src/WaterfallBackend.cpp:141-211) and BolidRecorder's state machine
(src/BolidRecorder.cpp:171-273) consume.
"""
import torch
import torch.distributed as dist


def shard_rows(total_rows, world, rank):
    """(first_row, rows) of `rank`: contiguous, sizes differ by at most one."""
    lo = (rank * total_rows) // world
    hi = ((rank + 1) * total_rows) // world
    return lo, hi - lo


def shard_samples(first_row, rows, bins, hop):
    """(first_sample, samples) a shard must hold, halo included; (x, 0) for an empty shard."""
    if rows <= 0:
        return first_row * hop, 0
    return first_row * hop, (rows - 1) * hop + bins


def all_shards(total_rows, world):
    return [shard_rows(total_rows, world, g) for g in range(world)]


def gather_rows(local, total_rows, group=None, async_op=False):
    """All-gather per-rank row blocks [rows_g, C] into [total_rows, C] in row order.

    Shards may differ by one row, all_gather_into_tensor needs equal blocks: every rank
    contributes max_rows rows (its own, zero-padded) and the padding is cut on arrival.
    Returns (stitched, work) -- `work` is None unless async_op."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    shards = all_shards(total_rows, world)
    max_rows = max(r for _, r in shards)
    cols = local.shape[1]
    assert local.shape[0] == shards[rank][1], (local.shape, shards[rank])
    if local.shape[0] == max_rows:
        send = local.contiguous()
    else:
        send = torch.zeros((max_rows, cols), dtype=local.dtype, device=local.device)
        send[:local.shape[0]] = local
    recv = torch.empty((world * max_rows, cols), dtype=local.dtype, device=local.device)
    work = dist.all_gather_into_tensor(recv, send, group=group, async_op=async_op)

    def stitch():
        if all(r == max_rows for _, r in shards):
            return recv
        return torch.cat([recv[g * max_rows:g * max_rows + r] for g, (_, r) in enumerate(shards)], dim=0)

    if async_op:
        return stitch, work
    return stitch(), None
