// ro_exchange.cpp -- the one exchange step of the time-chunk split: the stitch of every rank's rows for the
// FITS writer and the detector's state machine (src/WaterfallBackend.cpp:174-205, src/BolidRecorder.cpp:171-273), for a
// C++ host that owns an ncclComm_t.  librccl is resolved at run time; nothing here runs inside the transform.
#include "ro_host.h"

using namespace ro::host;

// RCCL, resolved at run time so that the library has no link-time dependency on it: one dlopen / dlsym per process,
// under std::call_once (several host threads may drive their own handles and communicators)
namespace {
struct Rccl {
    int (*all_gather)(const void *, void *, size_t, int, void *, hipStream_t) = nullptr;
    int (*group_start)(void) = nullptr;
    int (*group_end)(void) = nullptr;
    int (*send)(const void *, size_t, int, int, void *, hipStream_t) = nullptr;
    int (*recv)(void *, size_t, int, int, void *, hipStream_t) = nullptr;
};
const Rccl &rccl_api()
{
    static Rccl api;
    static std::once_flag once;
    std::call_once(once, [] {
        void *lib = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
        if (!lib) lib = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
        if (!lib) lib = dlopen("/opt/rocm/lib/librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
        if (!lib) return;
        api.all_gather = reinterpret_cast<decltype(api.all_gather)>(dlsym(lib, "ncclAllGather"));
        api.group_start = reinterpret_cast<decltype(api.group_start)>(dlsym(lib, "ncclGroupStart"));
        api.group_end = reinterpret_cast<decltype(api.group_end)>(dlsym(lib, "ncclGroupEnd"));
        api.send = reinterpret_cast<decltype(api.send)>(dlsym(lib, "ncclSend"));
        api.recv = reinterpret_cast<decltype(api.recv)>(dlsym(lib, "ncclRecv"));
    });
    return api;
}
}  // namespace

// the all-gather itself
extern "C" int ro_allgather_rows(void *nccl_comm, const void *d_local, int64_t local_rows, int64_t total_rows, int world,
                                 int rank, size_t row_bytes, void *d_staging, void *d_gathered, void *stream)
{
    if (!nccl_comm || world < 1 || rank < 0 || rank >= world || total_rows < 0 || row_bytes == 0 || !d_staging ||
        !d_gathered || (local_rows > 0 && !d_local))
        return fail(RO_ERR_INVALID, "ro_allgather_rows: bad arguments");
    int64_t first = 0, mine = 0;
    ro_shard_rows(total_rows, world, rank, &first, &mine);
    if (local_rows != mine)
        return fail(RO_ERR_INVALID, "ro_allgather_rows: rank %d of %d owns %lld of %lld rows, not %lld", rank, world,
                    (long long)mine, (long long)total_rows, (long long)local_rows);
    const Rccl &rccl = rccl_api();
    if (!rccl.all_gather) return fail(RO_ERR_UNSUPPORTED, "librccl (ncclAllGather) not found on this host");
    hipStream_t s = (hipStream_t)stream;
    const int64_t block = ro_shard_max_rows(total_rows, world);
    if (block == 0) return RO_OK;
    const size_t used = (size_t)local_rows * row_bytes, whole = (size_t)block * row_bytes;
    if (used) HIP_TRY(hipMemcpyAsync(d_staging, d_local, used, hipMemcpyDeviceToDevice, s));
    if (whole > used) HIP_TRY(hipMemsetAsync(static_cast<char *>(d_staging) + used, 0, whole - used, s));
    const int rc = rccl.all_gather(d_staging, d_gathered, whole, /*ncclInt8*/ 0, nccl_comm, s);
    if (rc != 0) return fail(RO_ERR_HIP, "ncclAllGather failed with code %d", rc);
    return RO_OK;
}

// gather to ONE rank, rows landing where they belong: ncclSend / ncclRecv in a group, no padding, no stitch
extern "C" int ro_gather_rows(void *nccl_comm, const void *d_local, int64_t local_rows, int64_t total_rows, int world,
                              int rank, int root, size_t row_bytes, void *d_out, void *stream)
{
    if (!nccl_comm || world < 1 || rank < 0 || rank >= world || root < 0 || root >= world || total_rows < 0 ||
        row_bytes == 0 || (local_rows > 0 && !d_local) || (rank == root && total_rows > 0 && !d_out))
        return fail(RO_ERR_INVALID, "ro_gather_rows: bad arguments");
    int64_t first = 0, mine = 0;
    ro_shard_rows(total_rows, world, rank, &first, &mine);
    if (local_rows != mine)
        return fail(RO_ERR_INVALID, "ro_gather_rows: rank %d of %d owns %lld of %lld rows, not %lld", rank, world,
                    (long long)mine, (long long)total_rows, (long long)local_rows);
    const Rccl &rccl = rccl_api();
    if (!rccl.group_start || !rccl.group_end || !rccl.send || !rccl.recv)
        return fail(RO_ERR_UNSUPPORTED, "librccl (ncclSend / ncclRecv) not found on this host");
    const auto group_start = rccl.group_start, group_end = rccl.group_end;
    const auto send = rccl.send;
    const auto recv = rccl.recv;
    hipStream_t s = (hipStream_t)stream;
    if (rank == root && mine > 0)           // the root's own rows: a copy
        HIP_TRY(hipMemcpyAsync(static_cast<char *>(d_out) + (size_t)first * row_bytes, d_local, (size_t)mine * row_bytes,
                               hipMemcpyDeviceToDevice, s));
    int rc = group_start();
    // the direct schedule with one receiver: the root takes every step's receive, every other rank sends in the one step
    // whose `to` is the root
    for (int k = 1; k < world && rc == 0; ++k) {
        int to = 0, from = 0;
        int64_t f = 0, n = 0;
        if (ro_direct_schedule(world, rank, total_rows, k, &to, &from, &f, &n) != RO_OK) return RO_ERR_INVALID;
        if (rank == root) {
            if (n > 0) rc = recv(static_cast<char *>(d_out) + (size_t)f * row_bytes, (size_t)n * row_bytes, 0, from, nccl_comm, s);
        } else if (to == root && mine > 0) {
            rc = send(d_local, (size_t)mine * row_bytes, /*ncclInt8*/ 0, root, nccl_comm, s);
        }
    }
    const int rc_end = group_end();
    if (rc != 0 || rc_end != 0) return fail(RO_ERR_HIP, "ncclSend / ncclRecv failed with code %d", rc ? rc : rc_end);
    return RO_OK;
}

// the all-gather as a DIRECT exchange: inside one group every rank sends its block to each peer and receives each peer's
// block at its stitched place -- world - 1 point-to-point transfers per rank over world - 1 different xGMI links, no
// ring through one link, no padding, no stitch (what ro_gather_rows does for one root, for all)
extern "C" int ro_allgather_rows_direct(void *nccl_comm, const void *d_local, int64_t local_rows, int64_t total_rows, int world,
                                        int rank, size_t row_bytes, void *d_out, void *stream)
{
    if (!nccl_comm || world < 1 || rank < 0 || rank >= world || total_rows < 0 || row_bytes == 0 ||
        (local_rows > 0 && !d_local) || (total_rows > 0 && !d_out))
        return fail(RO_ERR_INVALID, "ro_allgather_rows_direct: bad arguments");
    int64_t first = 0, mine = 0;
    ro_shard_rows(total_rows, world, rank, &first, &mine);
    if (local_rows != mine)
        return fail(RO_ERR_INVALID, "ro_allgather_rows_direct: rank %d of %d owns %lld of %lld rows, not %lld", rank, world,
                    (long long)mine, (long long)total_rows, (long long)local_rows);
    const Rccl &rccl = rccl_api();
    if (!rccl.group_start || !rccl.group_end || !rccl.send || !rccl.recv)
        return fail(RO_ERR_UNSUPPORTED, "librccl (ncclSend / ncclRecv) not found on this host");
    hipStream_t s = (hipStream_t)stream;
    char *out = static_cast<char *>(d_out);
    if (mine > 0)                            // this rank's own rows: a copy to their place
        HIP_TRY(hipMemcpyAsync(out + (size_t)first * row_bytes, d_local, (size_t)mine * row_bytes, hipMemcpyDeviceToDevice, s));
    int rc = rccl.group_start();
    // ro_direct_schedule: in step k every rank sends to rank + k and receives from rank - k
    for (int k = 1; k < world && rc == 0; ++k) {
        int to = 0, from = 0;
        int64_t f = 0, n = 0;
        if (ro_direct_schedule(world, rank, total_rows, k, &to, &from, &f, &n) != RO_OK) return RO_ERR_INVALID;
        if (mine > 0) rc = rccl.send(d_local, (size_t)mine * row_bytes, /*ncclInt8*/ 0, to, nccl_comm, s);
        if (rc == 0 && n > 0) rc = rccl.recv(out + (size_t)f * row_bytes, (size_t)n * row_bytes, 0, from, nccl_comm, s);
    }
    const int rc_end = rccl.group_end();
    if (rc != 0 || rc_end != 0) return fail(RO_ERR_HIP, "ncclSend / ncclRecv failed with code %d", rc ? rc : rc_end);
    return RO_OK;
}

extern "C" int ro_stitch_rows_device(const void *d_gathered, int64_t total_rows, int world, size_t row_bytes, void *d_out,
                                     void *stream)
{
    if (total_rows < 0 || world < 1 || (total_rows > 0 && (!d_gathered || !d_out)))
        return fail(RO_ERR_INVALID, "ro_stitch_rows_device: bad arguments");
    const int64_t block = ro_shard_max_rows(total_rows, world);
    for (int g = 0; g < world; ++g) {
        int64_t first = 0, rows = 0;
        ro_shard_rows(total_rows, world, g, &first, &rows);
        if (rows > 0)
            HIP_TRY(hipMemcpyAsync(static_cast<char *>(d_out) + (size_t)first * row_bytes,
                                   static_cast<const char *>(d_gathered) + (size_t)g * (size_t)block * row_bytes,
                                   (size_t)rows * row_bytes, hipMemcpyDeviceToDevice, (hipStream_t)stream));
    }
    return RO_OK;
}
