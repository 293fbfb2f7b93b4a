"""GPU: ro_allgather_rows / ro_stitch_rows_device, the C-ABI form of the multi-GPU stitch, over a real RCCL
communicator.  One GPU is what this box has, so the communicator has ONE rank (ncclCommInitRank through ctypes): that
checks the plumbing -- dlopen of librccl, padding into the staging block, the byte-typed ncclAllGather on the caller's
stream, the device-side stitch -- not the fabric; the world > 1 arithmetic is covered by the CPU tests of
ro_shard_rows / ro_stitch_rows and by tests/test_gpu_c5.py, which lays out eight shards exactly as this call does."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


class NcclUniqueId(C.Structure):
    _fields_ = [("internal", C.c_char * 128)]


def test_allgather_rows_on_a_single_rank_communicator(ro, torch_cuda):
    torch = torch_cuda
    try:
        rccl = C.CDLL("librccl.so.1")
    except OSError:
        pytest.skip("no librccl on this box")
    uid = NcclUniqueId()
    assert rccl.ncclGetUniqueId(C.byref(uid)) == 0
    comm = C.c_void_p()
    rccl.ncclCommInitRank.argtypes = [C.POINTER(C.c_void_p), C.c_int, NcclUniqueId, C.c_int]
    torch.cuda.set_device(0)
    assert rccl.ncclCommInitRank(C.byref(comm), 1, uid, 0) == 0
    try:
        L = ro.library()
        rows, cols = 1237, 615
        local = torch.randn((rows, cols), device="cuda", dtype=torch.float32)
        staging = torch.empty((ro.shard_max_rows(rows, 1), cols), device="cuda", dtype=torch.float32)
        gathered = torch.empty_like(staging)
        out = torch.zeros((rows, cols), device="cuda", dtype=torch.float32)
        s = torch.cuda.current_stream().cuda_stream
        rc = L.ro_allgather_rows(comm, C.c_void_p(local.data_ptr()), rows, rows, 1, 0, cols * 4,
                                 C.c_void_p(staging.data_ptr()), C.c_void_p(gathered.data_ptr()), C.c_void_p(s))
        assert rc == 0, L.ro_last_error()
        assert L.ro_stitch_rows_device(C.c_void_p(gathered.data_ptr()), rows, 1, cols * 4, C.c_void_p(out.data_ptr()),
                                       C.c_void_p(s)) == 0
        torch.cuda.synchronize()
        assert torch.equal(out, local)
        # the gather to one rank (what a host with a single consumer of the rows uses): rows land stitched
        out2 = torch.zeros((rows, cols), device="cuda", dtype=torch.float32)
        rc = L.ro_gather_rows(comm, C.c_void_p(local.data_ptr()), rows, rows, 1, 0, 0, cols * 4,
                              C.c_void_p(out2.data_ptr()), C.c_void_p(s))
        assert rc == 0, L.ro_last_error()
        torch.cuda.synchronize()
        assert torch.equal(out2, local)
        # the all-gather as direct point-to-point transfers (one group of sends / receives): rows land stitched on every rank
        out3 = torch.zeros((rows, cols), device="cuda", dtype=torch.float32)
        rc = L.ro_allgather_rows_direct(comm, C.c_void_p(local.data_ptr()), rows, rows, 1, 0, cols * 4,
                                        C.c_void_p(out3.data_ptr()), C.c_void_p(s))
        assert rc == 0, L.ro_last_error()
        torch.cuda.synchronize()
        assert torch.equal(out3, local)
        assert L.ro_allgather_rows_direct(comm, C.c_void_p(local.data_ptr()), rows - 1, rows, 1, 0, cols * 4,
                                          C.c_void_p(out3.data_ptr()), C.c_void_p(s)) == -1
        assert L.ro_gather_rows(comm, C.c_void_p(local.data_ptr()), rows, rows, 1, 0, 1, cols * 4,
                                C.c_void_p(out2.data_ptr()), C.c_void_p(s)) == -1          # root outside the world
        # a row count that is not this rank's share is refused before anything is queued
        assert L.ro_allgather_rows(comm, C.c_void_p(local.data_ptr()), rows - 1, rows, 1, 0, cols * 4,
                                   C.c_void_p(staging.data_ptr()), C.c_void_p(gathered.data_ptr()), C.c_void_p(s)) == -1
    finally:
        rccl.ncclCommDestroy.argtypes = [C.c_void_p]
        rccl.ncclCommDestroy(comm)


def test_stitch_rows_device_undoes_the_padding(ro, torch_cuda):
    torch = torch_cuda
    L = ro.library()
    total, world, cols = 41, 3, 7
    full = torch.arange(total * cols, device="cuda", dtype=torch.float32).reshape(total, cols)
    m = ro.shard_max_rows(total, world)
    gathered = torch.zeros((world * m, cols), device="cuda", dtype=torch.float32)
    for g in range(world):
        first, rows = ro.shard_rows(total, world, g)
        gathered[g * m:g * m + rows] = full[first:first + rows]
    out = torch.empty_like(full)
    assert L.ro_stitch_rows_device(C.c_void_p(gathered.data_ptr()), total, world, cols * 4, C.c_void_p(out.data_ptr()),
                                   C.c_void_p(torch.cuda.current_stream().cuda_stream)) == 0
    torch.cuda.synchronize()
    assert torch.equal(out, full)
