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

The shard arithmetic itself lives behind the C ABI (ro_shard_rows / ro_shard_samples /
ro_shard_max_rows / ro_stitch_rows, include/ro_stft.h) so that a C++ host needs no Python;
this module is the torch.distributed plumbing around it.
"""
import torch
import torch.distributed as dist

from . import capi


def shard_rows(total_rows, world, rank):
    """(first_row, rows) of `rank`: contiguous, sizes differ by at most one."""
    return capi.shard_rows(total_rows, world, rank)


def shard_samples(first_row, rows, bins, hop):
    """(first_sample, samples) a shard must hold, halo included; (x, 0) for an empty shard."""
    return capi.shard_samples(first_row, rows, bins, bins - hop)


def all_shards(total_rows, world):
    return [shard_rows(total_rows, world, g) for g in range(world)]


def pad_block(local, total_rows, world):
    """This rank's rows as the equal-sized block an all-gather needs: ro_shard_max_rows rows, zero-padded."""
    max_rows = capi.shard_max_rows(total_rows, world)
    if local.shape[0] == max_rows:
        return local.contiguous()
    send = torch.zeros((max_rows,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    send[:local.shape[0]] = local
    return send


def stitch(recv, total_rows, world):
    """[world * max_rows, C] as an equal-block all-gather leaves it -> [total_rows, C] in row order
    (the device-tensor form of ro_stitch_rows: the padding of the short shards is cut)."""
    shards = all_shards(total_rows, world)
    max_rows = capi.shard_max_rows(total_rows, world)
    if all(r == max_rows for _, r in shards):
        return recv
    return torch.cat([recv[g * max_rows:g * max_rows + r] for g, (_, r) in enumerate(shards)], dim=0)


def gather_rows(local, total_rows, group=None, async_op=False, root=None):
    """All-gather per-rank row blocks [rows_g, C] into [total_rows, C] in row order.

    Shards may differ by one row, all_gather_into_tensor needs equal blocks: every rank
    contributes max_rows rows (its own, zero-padded) and the padding is cut on arrival.
    root = r gathers to rank r only (the other ranks get None): 1/world of the inbound traffic
    per non-root rank, for when only one rank writes the FITS files.
    Returns (stitched, work) -- `work` is None unless async_op, and `stitched` is then a
    callable to be invoked after work.wait()."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    shards = all_shards(total_rows, world)
    assert local.shape[0] == shards[rank][1], (local.shape, shards[rank])
    send = pad_block(local, total_rows, world)
    max_rows = send.shape[0]
    if root is None:
        recv = torch.empty((world * max_rows,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
        work = dist.all_gather_into_tensor(recv, send, group=group, async_op=async_op)
    else:
        recv = None
        parts = None
        if rank == root:
            recv = torch.empty((world * max_rows,) + tuple(local.shape[1:]), dtype=local.dtype,
                               device=local.device)
            parts = list(recv.view((world, max_rows) + tuple(local.shape[1:])).unbind(0))
        work = dist.gather(send, parts, dst=root, group=group, async_op=async_op)

    def finish():
        return None if recv is None else stitch(recv, total_rows, world)

    if async_op:
        return finish, work
    return finish(), None


def gather_rows_direct(local, total_rows, group=None, out=None):
    """The all-gather as a DIRECT exchange (xGMI is point to point): every rank sends its rows to each peer
    and receives each peer's rows at their stitched place -- the torch.distributed form of ro_allgather_rows_direct
    (the SAME schedule, step by step out of ro_direct_schedule: to rank + k, from rank - k, k = 1 .. world - 1, one batch
    of point-to-point operations).  No padding, no stitch.  Returns [total_rows, C] in row order (written into `out`
    when given)."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    shards = all_shards(total_rows, world)
    first, mine = shards[rank]
    assert local.shape[0] == mine, (local.shape, shards[rank])
    if out is None:
        out = torch.empty((total_rows,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    src = local.contiguous()
    out[first:first + mine].copy_(src)
    ops = []
    for k in range(1, world):
        to, frm, f, n = capi.direct_schedule(world, rank, total_rows, k)
        if mine > 0:
            ops.append(dist.P2POp(dist.isend, src, to, group))
        if n > 0:
            ops.append(dist.P2POp(dist.irecv, out[f:f + n], frm, group))
    if ops:
        for req in dist.batch_isend_irecv(ops):
            req.wait()
    return out
