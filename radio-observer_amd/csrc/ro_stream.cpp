// ro_stream.cpp -- the streaming half of the C ABI: ro_stft_push stages the caller's samples, full batches run on rotating
// slots of device and pinned buffers (a latency-bound batch with a row sink as ONE captured graph per slot), finished rows
// travel home behind events and ro_stft_fetch / the row sink hands them over.  Replaces the framing loop of
// src/FFTBackend.cpp:192-279 and the row ring's producer side, src/WaterfallBackend.cpp:485-541.
#include "ro_host.h"

using namespace ro::host;

namespace {

Batch *acquire_batch(ro_stft *h)
{
    if (!h->batch_pool.empty()) {
        Batch *b = h->batch_pool.back();
        h->batch_pool.pop_back();
        return b;
    }
    Batch *b = new (std::nothrow) Batch();
    if (!b) return nullptr;
    b->capacity_rows = h->batch_rows;
    if ((!h->sink &&
         hipHostMalloc(reinterpret_cast<void **>(&b->data), (size_t)b->capacity_rows * h->out_cols * sizeof(float),
                       hipHostMallocDefault) != hipSuccess) ||
        hipHostMalloc(reinterpret_cast<void **>(&b->records), (size_t)b->capacity_rows * sizeof(ro_scan_record_t),
                      hipHostMallocDefault) != hipSuccess ||
        (h->cfg.tile_ln &&
         (hipHostMalloc(reinterpret_cast<void **>(&b->ln), (size_t)b->capacity_rows * h->out_cols * sizeof(float),
                        hipHostMallocDefault) != hipSuccess ||
          hipHostMalloc(reinterpret_cast<void **>(&b->minmax), (size_t)b->capacity_rows * 2 * sizeof(float),
                        hipHostMallocDefault) != hipSuccess)) ||
        hipEventCreateWithFlags(&b->done, hipEventDisableTiming) != hipSuccess ||
        hipEventCreate(&b->k0) != hipSuccess || hipEventCreate(&b->k1) != hipSuccess) {
        if (b->data) (void)hipHostFree(b->data);
        if (b->ln) (void)hipHostFree(b->ln);
        if (b->minmax) (void)hipHostFree(b->minmax);
        if (b->records) (void)hipHostFree(b->records);
        if (b->done) (void)hipEventDestroy(b->done);
        if (b->k0) (void)hipEventDestroy(b->k0);
        if (b->k1) (void)hipEventDestroy(b->k1);
        delete b;
        return nullptr;
    }
    return b;
}

void release_batch(ro_stft *h, Batch *b)
{
    b->consumed = 0;
    b->rows = 0;
    b->pending = false;
    h->batch_pool.push_back(b);
}

size_t stage_sample_bytes(const ro_stft *h) { return h->stage_fmt == RO_IQ_I16 ? 4 : h->stage_fmt == RO_IQ_F64 ? 16 : 8; }
// bytes per sample the slots are sized for: doubles only where the handle's kernel takes them (RO_PRECISION_F64 at
// 256 ... 65536 bins)
size_t slot_sample_bytes(const ro_stft *h) { return h->f64reg ? 16 : 8; }

// streaming buffers, all or nothing: a failure half way frees what was allocated, and the next push tries again
int ensure_stream_slots(ro_stft *h)
{
    if (h->slots_ready) return RO_OK;
    HIP_TRY(hipSetDevice(h->device));
    const size_t in_samples = (size_t)(h->batch_rows - 1) * h->hop + h->bins;
    hipError_t e = hipSuccess;
    auto ok = [&](hipError_t r) { if (e == hipSuccess) e = r; return e == hipSuccess; };
    // Three streams chained by events: the upload of batch n + 1 overlaps the kernels of batch n and the download of
    // batch n - 1.  (Round 5 measured everything in order on ONE stream for latency-bound batches -- eight runtime calls
    // fewer per batch: push 3.9 -> 2.7 us per call, and the same 6.7e4 rows/s, because a batch then occupies the stream
    // for its whole upload -> kernel -> download chain, ~90 us; with three streams a second batch in flight overlaps it.)
    ok(hipStreamCreateWithFlags(&h->s_in, hipStreamNonBlocking)) && ok(hipStreamCreateWithFlags(&h->s_out, hipStreamNonBlocking));
    for (auto &sl : h->slot) {
        ok(hipMalloc(&sl.d_iq, in_samples * slot_sample_bytes(h))) &&
            ok(hipMalloc(&sl.d_rows, (size_t)h->batch_rows * h->bins * sizeof(float))) &&
            ok(hipMalloc(&sl.d_records, (size_t)h->batch_rows * sizeof(ro_scan_record_t))) &&
            ok(hipHostMalloc(&sl.h_in, in_samples * slot_sample_bytes(h), hipHostMallocDefault)) &&
            ok(hipEventCreateWithFlags(&sl.uploaded, hipEventDisableTiming)) &&
            ok(hipEventCreateWithFlags(&sl.computed, hipEventDisableTiming)) &&
            ok(hipEventCreateWithFlags(&sl.drained, hipEventDisableTiming));
        if (h->cfg.tile_cols > 0)
            ok(hipMalloc(&sl.d_tile, (size_t)h->batch_rows * h->cfg.tile_cols * sizeof(float)));
        if (h->cfg.tile_ln)
            ok(hipMalloc(&sl.d_ln, (size_t)h->batch_rows * h->cfg.tile_cols * sizeof(float))) &&
                ok(hipMalloc(&sl.d_minmax, (size_t)h->batch_rows * 2 * sizeof(float)));
    }
    if (e != hipSuccess) {
        free_stream_slots(h);
        return fail(RO_ERR_HIP, "allocating the streaming buffers failed: %s", hipGetErrorString(e));
    }
    h->slots_ready = true;
    return RO_OK;
}

// The one place the host waits for the GPU on the streaming path: a batch's download has finished.  Its kernel time
// (GPU events around its kernels) goes into the counters of ro_stft_timing / ro_stft_stats the first time round.
int await_batch(ro_stft *h, Batch *b)
{
    if (!b->pending) return RO_OK;
    HIP_TRY(hipEventSynchronize(b->done));
    b->pending = false;
    float ms = 0.f;
    h->timing.batches += 1;
    h->timing.batch_rows += b->rows;
    if (b->timed && hipEventElapsedTime(&ms, b->k0, b->k1) == hipSuccess) {
        h->stat_kernel_ms += ms;
        h->timed_batches += 1;
        h->timed_rows += b->rows;
        h->batch_ms_sum += ms;
        h->last_batch_ms = ms;
        h->timing.batch_gpu_ms_max = std::max(h->timing.batch_gpu_ms_max, (double)ms);
    } else if (!b->timed) {
        h->stat_kernel_ms += h->last_batch_ms;      // (an untimed graphed batch: the same graph as the last timed one)
    }
    return RO_OK;
}

// run one batch of the streaming path: rows [rows_emitted, +rows) from the staged samples.  Upload, kernels and
// download are queued on three streams chained by events and the call returns; the host only waits when it is about
// to overwrite a pinned staging buffer whose upload has not finished.
int run_stream_batch(ro_stft *h, int64_t rows)
{
    if (rows <= 0) return RO_OK;
    HIP_TRY(hipSetDevice(h->device));
    const size_t sb = stage_sample_bytes(h);
    const int64_t need = (rows - 1) * (int64_t)h->hop + h->bins;       // samples
    if ((int64_t)h->staged_have < need) return fail(RO_ERR_STATE, "internal: %lld samples staged, %lld needed",
                                                    (long long)h->staged_have, (long long)need);
    // rows that have not been fetched sit in the sink's slots: a batch that would lap them is not launched.  ro_stft_push
    // never gets here in that state (it refuses such a call whole, before staging); ro_stft_flush does, and leaves the
    // staged samples where they are, so a flush repeated after a fetch loses nothing
    if (h->sink && h->rows_ready + rows > h->sink_cap)
        return fail(RO_ERR_STATE, "row sink full: %lld rows wait to be fetched in a ring of %lld slots", (long long)h->rows_ready,
                    (long long)h->sink_cap);
    ro_stft::Slot &sl = h->slot[h->batch_seq % RO_STREAM_SLOTS];                      // (its samples are already in sl.h_in)
    // Back-pressure: one large ro_stft_push must not queue a pinned batch per launch without bound (64 MiB each with
    // full rows).  Batches older than the newest MAX_IN_FLIGHT are waited for here -- they stay in `ready` for the
    // next fetch, their buffers are simply known to be complete.
    constexpr size_t MAX_IN_FLIGHT = 4;
    if (h->ready.size() >= MAX_IN_FLIGHT) {
        const int wrc = await_batch(h, h->ready[h->ready.size() - MAX_IN_FLIGHT]);
        if (wrc != RO_OK) return wrc;
    }
    Batch *b = acquire_batch(h);
    if (!b) return fail(RO_ERR_NOMEM, "out of pinned host memory for a row batch");
    int rc = RO_OK;
    auto step = [&](hipError_t e, const char *what) {
        if (rc == RO_OK && e != hipSuccess) rc = fail(RO_ERR_HIP, "%s failed: %s", what, hipGetErrorString(e));
        return rc == RO_OK;
    };
    // ---- A latency-bound batch (a second of rows: well under a MiB) is all launch overhead -- fourteen runtime calls on
    // three streams for 23 us of GPU work.  With a row sink, a FULL batch of a slot runs as ONE graph (upload and every
    // kernel of the size: captured once from the calls below) on the slot's own stream, followed by the downloads into
    // the sink's slots: seven calls.  The slots' streams overlap a batch with the one or two before it; a slot's
    // own batches are ordered by its stream.  Partial batches (a flush) and the first batch of a slot take the plain path.
    const bool small = (size_t)h->batch_rows * h->out_cols * sizeof(float) <= ((size_t)4 << 20);
    // (float32 power-of-two single-kernel handles only: the FP64 and chirp-z paths keep per-launch host state -- scratch
    // blocks, an inner handle -- that a replayed graph would not see, and the four-step sizes hand Z from their column
    // kernel to their row kernel through ONE scratch block per handle, which two slots' graphs on two streams would share)
    bool graphed = RO_STREAM_GRAPH && small && h->sink && !h->cfg.tile_ln && !h->f64 && !h->czt && !h->four &&
                   rows == h->batch_rows && sl.uses > 0;
    ro_scan_record_t *g_recs = h->cfg.enable_scan ? sl.d_records : nullptr;
    if (graphed && (!sl.gexec || sl.graph_fmt != h->stage_fmt)) {
        if (sl.gexec) { (void)hipGraphExecDestroy(sl.gexec); sl.gexec = nullptr; }
        if (!sl.gstream) step(hipStreamCreateWithFlags(&sl.gstream, hipStreamNonBlocking), "hipStreamCreateWithFlags");
        hipGraph_t g = nullptr;
        if (rc == RO_OK && step(hipStreamBeginCapture(sl.gstream, hipStreamCaptureModeThreadLocal), "hipStreamBeginCapture")) {
            step(hipMemcpyAsync(sl.d_iq, sl.h_in, (size_t)need * sb, hipMemcpyHostToDevice, sl.gstream), "upload");
            if (rc == RO_OK) rc = launch_transform(h, sl.d_iq, h->stage_fmt, 0, rows, sl.d_rows, h->bins, sl.gstream, sl.d_tile, g_recs, sl.d_ln);
            if (rc == RO_OK) rc = launch_tile_and_scan(h, sl.d_rows, h->bins, rows, sl.d_tile, g_recs, sl.gstream, sl.d_ln, sl.d_minmax);
            const hipError_t ce = hipStreamEndCapture(sl.gstream, &g);          // (always: leaves capture mode)
            if (rc == RO_OK) step(ce, "hipStreamEndCapture");
            if (rc == RO_OK) step(hipGraphInstantiate(&sl.gexec, g, nullptr, nullptr, 0), "hipGraphInstantiate");
            if (g) (void)hipGraphDestroy(g);
        }
        if (rc != RO_OK) {                        // no graph on this runtime: the plain path from now on, not an error
            (void)hipGetLastError();
            sl.gexec = nullptr;
            h->graph_refused = true;
            rc = RO_OK;
        }
        sl.graph_fmt = h->stage_fmt;
    }
    graphed = graphed && sl.gexec && !h->graph_refused;
    if (graphed != sl.on_gstream && sl.uses > 0) {
        // the slot changes streams: what its last batch queued has to be over (a handful of times per stream: first graphed
        // batch, a flush)
        if (sl.on_gstream) step(hipStreamSynchronize(sl.gstream), "hipStreamSynchronize");
        else step(hipStreamSynchronize(h->s_out), "hipStreamSynchronize");
    }
    sl.on_gstream = graphed;
    sl.uses += 1;
    if (graphed) {
        hipStream_t gs = sl.gstream;
        // (`uploaded` is recorded by a stream call BEHIND the graph, not by a node inside it: the host waits on it before it
        // stages into this slot's pinned buffer again, and an event that only a queued graph will record still reads as
        // its previous, completed record -- the host then overwrote samples the upload had not read yet: found by the
        // seeded soak of tests/test_gpu_streaming.py)
        // A runtime call costs the host 1 - 4 us here (tools/r5/event_cost.hip: an event record 2.4, the graph 10, a 2-D copy
        // 4) and a batch of a second of rows is 40 us of host time in all, so the timing events around the kernels go round
        // one batch in RO_GRAPH_TIME_EVERY only (the graph is the same every time; ro_stft_timing averages over the timed
        // ones): +8 % rows/s at the Backend's default batch, settings alternated inside one process
        // (profiles/r05_host_calls_ab.txt).  The same A/B says the `uploaded` event has to stay: with the batch's own `done`
        // event as the host's "staging buffer is free again" the host waits for a whole batch two launches back instead of
        // its upload, and the rate halves.
        const int every = h->diag_time_every > 0 ? h->diag_time_every : RO_GRAPH_TIME_EVERY;
        b->timed = h->graph_batches++ % every == 0;
        if (b->timed) step(hipEventRecord(b->k0, gs), "hipEventRecord");
        if (h->diag_direct) {                       // the graph's calls made one by one on the slot's stream
            step(hipMemcpyAsync(sl.d_iq, sl.h_in, (size_t)need * sb, hipMemcpyHostToDevice, gs), "upload");
            if (rc == RO_OK) rc = launch_transform(h, sl.d_iq, h->stage_fmt, 0, rows, sl.d_rows, h->bins, gs, sl.d_tile, g_recs, sl.d_ln);
            if (rc == RO_OK) rc = launch_tile_and_scan(h, sl.d_rows, h->bins, rows, sl.d_tile, g_recs, gs, sl.d_ln, sl.d_minmax);
        } else {
            step(hipGraphLaunch(sl.gexec, gs), "hipGraphLaunch");
        }
        if (b->timed) step(hipEventRecord(b->k1, gs), "hipEventRecord");
        if (!h->diag_done_only) {
            step(hipEventRecord(sl.uploaded, gs), "hipEventRecord");
            sl.staging_free = sl.uploaded;
        } else {
            sl.staging_free = b->done;
        }
        if (rc == RO_OK) {
            const float *src = h->cfg.tile_cols > 0 ? sl.d_tile : sl.d_rows;
            const size_t w = (size_t)h->out_cols * sizeof(float);
            const int64_t s0 = (h->sink_first + h->rows_emitted) % h->sink_cap;
            const int64_t n0 = std::min<int64_t>(rows, h->sink_cap - s0);
            step(hipMemcpy2DAsync(h->sink + s0 * h->sink_stride, (size_t)h->sink_stride * sizeof(float), src, w, w, (size_t)n0,
                                  hipMemcpyDeviceToHost, gs), "download");
            if (n0 < rows)
                step(hipMemcpy2DAsync(h->sink, (size_t)h->sink_stride * sizeof(float), src + (size_t)n0 * h->out_cols, w, w,
                                      (size_t)(rows - n0), hipMemcpyDeviceToHost, gs), "download");
            if (h->cfg.enable_scan)
                step(hipMemcpyAsync(b->records, sl.d_records, (size_t)rows * sizeof(ro_scan_record_t), hipMemcpyDeviceToHost, gs),
                     "download");
        }
        step(hipEventRecord(b->done, gs), "hipEventRecord");
    } else {
    // upload (s_in): after the kernels that last read this slot's d_iq
    step(hipStreamWaitEvent(h->s_in, sl.computed, 0), "hipStreamWaitEvent") &&
        step(hipMemcpyAsync(sl.d_iq, sl.h_in, (size_t)need * sb, hipMemcpyHostToDevice, h->s_in), "upload") &&
        step(hipEventRecord(sl.uploaded, h->s_in), "hipEventRecord");
    sl.staging_free = sl.uploaded;
    b->timed = true;
    // kernels (stream): after the upload, and after the download that last read this slot's outputs
    step(hipStreamWaitEvent(h->stream, sl.uploaded, 0), "hipStreamWaitEvent") &&
        step(hipStreamWaitEvent(h->stream, sl.drained, 0), "hipStreamWaitEvent") &&
        step(hipEventRecord(b->k0, h->stream), "hipEventRecord");
    if (rc == RO_OK) {
        ro_scan_record_t *recs = h->cfg.enable_scan ? sl.d_records : nullptr;
        rc = launch_transform(h, sl.d_iq, h->stage_fmt, 0, rows, sl.d_rows, h->bins, h->stream, sl.d_tile, recs, sl.d_ln);
        if (rc == RO_OK)
            rc = launch_tile_and_scan(h, sl.d_rows, h->bins, rows, sl.d_tile, recs, h->stream, sl.d_ln, sl.d_minmax);
    }
    step(hipEventRecord(b->k1, h->stream), "hipEventRecord") && step(hipEventRecord(sl.computed, h->stream), "hipEventRecord");
    // download (s_out): only the columns somebody asked for travel -- the tile when one is configured
    step(hipStreamWaitEvent(h->s_out, sl.computed, 0), "hipStreamWaitEvent");
    if (rc == RO_OK) {
        const float *src = h->cfg.tile_cols > 0 ? sl.d_tile : sl.d_rows;
        if (h->sink) {
            // straight into the caller's ring: row r of the stream at slot (sink_first + r) mod sink_cap, in at most
            // two runs of consecutive slots
            const size_t w = (size_t)h->out_cols * sizeof(float);
            const int64_t s0 = (h->sink_first + h->rows_emitted) % h->sink_cap;
            const int64_t n0 = std::min<int64_t>(rows, h->sink_cap - s0);
            step(hipMemcpy2DAsync(h->sink + s0 * h->sink_stride, (size_t)h->sink_stride * sizeof(float), src, w, w, (size_t)n0,
                                  hipMemcpyDeviceToHost, h->s_out), "download");
            if (n0 < rows)
                step(hipMemcpy2DAsync(h->sink, (size_t)h->sink_stride * sizeof(float), src + (size_t)n0 * h->out_cols, w, w,
                                      (size_t)(rows - n0), hipMemcpyDeviceToHost, h->s_out), "download");
        } else {
            step(hipMemcpyAsync(b->data, src, (size_t)rows * h->out_cols * sizeof(float), hipMemcpyDeviceToHost, h->s_out),
                 "download");
        }
        if (h->cfg.enable_scan)
            step(hipMemcpyAsync(b->records, sl.d_records, (size_t)rows * sizeof(ro_scan_record_t), hipMemcpyDeviceToHost,
                                h->s_out), "download");
        if (h->cfg.tile_ln) {
            step(hipMemcpyAsync(b->ln, sl.d_ln, (size_t)rows * h->out_cols * sizeof(float), hipMemcpyDeviceToHost, h->s_out),
                 "download");
            step(hipMemcpyAsync(b->minmax, sl.d_minmax, (size_t)rows * 2 * sizeof(float), hipMemcpyDeviceToHost, h->s_out),
                 "download");
        }
    }
    step(hipEventRecord(sl.drained, h->s_out), "hipEventRecord") && step(hipEventRecord(b->done, h->s_out), "hipEventRecord");
    }
    if (rc != RO_OK) {
        // nothing of this batch is handed out; whatever was queued is allowed to finish before the buffers are reused
        (void)hipStreamSynchronize(h->s_in);
        (void)hipStreamSynchronize(h->stream);
        (void)hipStreamSynchronize(h->s_out);
        for (ro_stft::Slot &o : h->slot)
            if (o.gstream) (void)hipStreamSynchronize(o.gstream);
        release_batch(h, b);
        return rc;
    }
    b->first_row = h->rows_emitted;
    b->rows = rows;
    b->pending = true;
    h->batch_seq += 1;
    h->stat_launches += 1;
    h->stat_rows += rows;

    // the samples no later row needs are spent: the next row starts rows*hop further on.  What is left -- the overlap
    // and anything behind the batch's last row -- moves to the front of the other slot's staging buffer, whose own
    // upload (the batch before this one) has to be over first; the upload just queued only READS this slot.
    const int64_t consumed = rows * (int64_t)h->hop;
    h->stream_sample0 += consumed;
    h->rows_emitted += rows;
    h->rows_ready += rows;
    h->ready.push_back(b);
    ro_stft::Slot &nx = h->slot[h->batch_seq % RO_STREAM_SLOTS];                      // (batch_seq has moved on)
    const hipError_t we = hipEventSynchronize(nx.staging_free ? nx.staging_free : nx.uploaded);
    const size_t left = h->staged_have - (size_t)consumed;
    std::memcpy(nx.h_in, static_cast<const char *>(sl.h_in) + (size_t)consumed * sb, left * sb);
    h->staged_have = left;
    if (we != hipSuccess) return fail(RO_ERR_HIP, "hipEventSynchronize failed: %s", hipGetErrorString(we));
    return RO_OK;
}

int64_t staged_complete_rows(const ro_stft *h)
{
    const int64_t have = (int64_t)h->staged_have;
    if (have < h->bins) return 0;
    return (have - h->bins) / h->hop + 1;
}

}  // namespace

namespace ro {
namespace host {

void destroy_batch(Batch *b)
{
    if (b->data) (void)hipHostFree(b->data);
    if (b->ln) (void)hipHostFree(b->ln);
    if (b->minmax) (void)hipHostFree(b->minmax);
    if (b->records) (void)hipHostFree(b->records);
    if (b->done) (void)hipEventDestroy(b->done);
    if (b->k0) (void)hipEventDestroy(b->k0);
    if (b->k1) (void)hipEventDestroy(b->k1);
    delete b;
}

void free_stream_slots(ro_stft *h)
{
    for (auto &sl : h->slot) {
        if (sl.d_iq) (void)hipFree(sl.d_iq);
        if (sl.d_rows) (void)hipFree(sl.d_rows);
        if (sl.d_tile) (void)hipFree(sl.d_tile);
        if (sl.d_ln) (void)hipFree(sl.d_ln);
        if (sl.d_minmax) (void)hipFree(sl.d_minmax);
        if (sl.d_records) (void)hipFree(sl.d_records);
        if (sl.h_in) (void)hipHostFree(sl.h_in);
        if (sl.uploaded) (void)hipEventDestroy(sl.uploaded);
        if (sl.computed) (void)hipEventDestroy(sl.computed);
        if (sl.drained) (void)hipEventDestroy(sl.drained);
        if (sl.gexec) (void)hipGraphExecDestroy(sl.gexec);
        if (sl.gstream) (void)hipStreamDestroy(sl.gstream);
        sl = ro_stft::Slot();
    }
    if (h->s_in) (void)hipStreamDestroy(h->s_in);
    if (h->s_out) (void)hipStreamDestroy(h->s_out);
    h->s_in = h->s_out = nullptr;
    h->slots_ready = false;
}

}  // namespace host
}  // namespace ro

// ---------------------------------------------------------------------------
// streaming path
// ---------------------------------------------------------------------------
extern "C" int ro_stft_push(ro_stft_t *h, const void *iq, int format, int64_t samples, int64_t *rows_ready)
{
    if (!h) return fail(RO_ERR_INVALID, "null handle");
    if (samples < 0 || (samples > 0 && !iq)) return fail(RO_ERR_INVALID, "bad sample buffer");
    if (format != RO_IQ_F32 && format != RO_IQ_I16 && format != RO_IQ_F64)
        return fail(RO_ERR_INVALID, "unknown sample format %d", format);
    const double t0 = now_ms();
    int rc = ensure_stream_slots(h);
    if (rc != RO_OK) return rc;

    // The caller's buffer is only valid during the call (src/WAVStream.cpp:113,123): copy now.  int16 samples stay
    // int16 all the way to the kernel (half the staging memory and PCIe bytes: src/WAVStream.cpp:119-120 hands them
    // over un-normalised, the kernel widens them); float32 stays float32; the double Complex is staged as float32
    // (lossless for every frontend of the reference) -- except on a handle whose kernel multiplies doubles
    // (RO_PRECISION_F64 at 256 ... 65536 bins), where it stays what src/Backend.h:26-29 says it is.  A stream that
    // changes format mid-way is widened once to the wider of the two.
    const bool in_i16 = format == RO_IQ_I16;
    const int want_fmt = in_i16 ? RO_IQ_I16 : (format == RO_IQ_F64 && h->f64reg) ? RO_IQ_F64 : RO_IQ_F32;
    const size_t cap = (size_t)(h->batch_rows - 1) * h->hop + h->bins;        // samples one slot's staging buffer holds
    auto rank_of = [](int f) { return f == RO_IQ_I16 ? 0 : f == RO_IQ_F32 ? 1 : 2; };
    if (!h->stage_fmt_set) {
        h->stage_fmt = want_fmt;
        h->stage_fmt_set = true;
    } else if (rank_of(want_fmt) > rank_of(h->stage_fmt)) {
        // widen what is staged, in place and from the back (the buffer is sized for the widest format the handle stages)
        char *base = static_cast<char *>(h->slot[h->batch_seq % RO_STREAM_SLOTS].h_in);
        const size_t n = h->staged_have * 2;
        if (h->stage_fmt == RO_IQ_I16 && want_fmt == RO_IQ_F32) {
            const int16_t *src = reinterpret_cast<const int16_t *>(base);
            float *dst = reinterpret_cast<float *>(base);
            for (size_t i = n; i-- > 0;) dst[i] = (float)src[i];
        } else if (h->stage_fmt == RO_IQ_I16) {
            const int16_t *src = reinterpret_cast<const int16_t *>(base);
            double *dst = reinterpret_cast<double *>(base);
            for (size_t i = n; i-- > 0;) dst[i] = (double)src[i];
        } else {
            const float *src = reinterpret_cast<const float *>(base);
            double *dst = reinterpret_cast<double *>(base);
            for (size_t i = n; i-- > 0;) dst[i] = (double)src[i];
        }
        h->stage_fmt = want_fmt;
    }
    // With a row sink a push is all or nothing: the batches this call would complete are counted BEFORE anything is
    // staged, and a call whose rows would lap rows that still wait to be fetched is refused whole -- no sample taken,
    // no counter moved -- so the caller fetches and pushes the same buffer again (the streaming analogue of
    // RingBuffer2D::push never overwriting a reserved row silently, src/RingBuffer.h:482-509).
    if (h->sink) {
        const size_t spent = (size_t)h->batch_rows * h->hop;               // samples a batch retires
        size_t have = h->staged_have;
        int64_t batches = 0;
        for (int64_t left = samples; left > 0;) {
            const int64_t take = std::min<int64_t>(left, (int64_t)(cap - have));
            have += (size_t)take;
            left -= take;
            if (have == cap) { ++batches; have -= spent; }
        }
        if (h->rows_ready + batches * (int64_t)h->batch_rows > h->sink_cap)
            return fail(RO_ERR_STATE, "row sink full: this push would complete %lld rows with %lld waiting to be fetched in a "
                                      "ring of %lld slots; nothing was consumed -- fetch, then push the same samples again",
                        (long long)(batches * h->batch_rows), (long long)h->rows_ready, (long long)h->sink_cap);
    }
    const size_t sb = stage_sample_bytes(h);
    const size_t isb = format == RO_IQ_F64 ? 16 : format == RO_IQ_F32 ? 8 : 4;       // bytes per sample as delivered
    const char *in = static_cast<const char *>(iq);
    h->stat_samples += samples;
    for (int64_t left = samples; left > 0;) {
        // into the pinned buffer the next upload reads, converting on the way (no second copy)
        char *dstb = static_cast<char *>(h->slot[h->batch_seq % RO_STREAM_SLOTS].h_in) + h->staged_have * sb;
        const int64_t take = std::min<int64_t>(left, (int64_t)(cap - h->staged_have));
        if (h->stage_fmt == RO_IQ_I16) {
            std::memcpy(dstb, in, (size_t)take * 4);
        } else if (h->stage_fmt == RO_IQ_F64) {
            double *dst = reinterpret_cast<double *>(dstb);
            if (format == RO_IQ_F64) {
                std::memcpy(dst, in, (size_t)take * 2 * sizeof(double));
            } else if (in_i16) {
                const int16_t *src = reinterpret_cast<const int16_t *>(in);
                for (int64_t i = 0; i < take * 2; ++i) dst[i] = (double)src[i];
            } else {
                const float *src = reinterpret_cast<const float *>(in);
                for (int64_t i = 0; i < take * 2; ++i) dst[i] = (double)src[i];
            }
        } else {
            float *dst = reinterpret_cast<float *>(dstb);
            if (format == RO_IQ_F32) {
                std::memcpy(dst, in, (size_t)take * 2 * sizeof(float));
            } else if (in_i16) {
                const int16_t *src = reinterpret_cast<const int16_t *>(in);
                for (int64_t i = 0; i < take * 2; ++i) dst[i] = (float)src[i];
            } else {
                const double *src = reinterpret_cast<const double *>(in);         // struct Complex
                // (a slot of a few hundred KiB -- a latency-bound batch -- stays in the caches between the calls that
                // fill it and the overlap copy that reads it back; one of many MiB does not, and is written past them)
                // (tools/r5/host_nt.py: at a slot of 590 KiB -- the Backend's default batch -- the two forms cannot be told
                // apart: 1.15 ... 1.52 x 10^5 rows/s with either, from one process to the next)
                size_t nt_from = (size_t)8 << 20;
#ifdef RO_DIAG_KNOBS
                if (const char *e = getenv("RO_STAGE_NT_BYTES")) nt_from = (size_t)atoll(e);
#endif
                const bool past_caches = cap * sb > nt_from;
                for (int64_t at = 0; at < take * 2; at += (int64_t)1 << 30) {     // (the loops count in int)
                    const int n = (int)std::min<int64_t>(take * 2 - at, (int64_t)1 << 30);
                    if (past_caches) ro::narrowToFloatStream(src + at, dst + at, n);
                    else ro::narrowToFloat(src + at, dst + at, n);
                }
            }
        }
        h->staged_have += (size_t)take;
        in += (size_t)take * isb;
        left -= take;
        if (h->staged_have == cap) {                                      // = batch_rows complete rows
            rc = run_stream_batch(h, h->batch_rows);
            if (rc != RO_OK) return rc;
        }
    }
    if (rows_ready) *rows_ready = h->rows_ready;
    const double dt = now_ms() - t0;
    h->timing.push_calls += 1;
    h->push_ms_sum += dt;
    h->timing.push_ms_max = std::max(h->timing.push_ms_max, dt);
    return RO_OK;
}

extern "C" int ro_stft_flush(ro_stft_t *h, int64_t *rows_ready)
{
    if (!h) return fail(RO_ERR_INVALID, "null handle");
    while (h->slots_ready) {
        const int64_t n = std::min<int64_t>(staged_complete_rows(h), h->batch_rows);
        if (n <= 0) break;
        int rc = run_stream_batch(h, n);
        if (rc != RO_OK) return rc;
    }
    if (rows_ready) *rows_ready = h->rows_ready;
    return RO_OK;
}

extern "C" int ro_stft_fetch(ro_stft_t *h, int64_t max_rows, int first_col, int cols, float *rows_out,
                             ro_scan_record_t *records_out, int64_t *first_row_index, int64_t *rows_got)
{
    if (!h || !rows_got) return fail(RO_ERR_INVALID, "null argument");
    if (max_rows < 0) return fail(RO_ERR_INVALID, "negative max_rows");
    if (rows_out && (first_col < h->out_first || cols <= 0 || first_col + cols > h->out_first + h->out_cols))
        return fail(RO_ERR_INVALID, "columns [%d,+%d) outside [%d,+%d) -- what this handle brings to the host%s",
                    first_col, cols, h->out_first, h->out_cols,
                    h->cfg.tile_cols > 0 ? " (the configured tile)" : "");
    if (records_out && !h->cfg.enable_scan) return fail(RO_ERR_STATE, "scan records requested but scan is off");
    if (rows_out && h->sink) return fail(RO_ERR_STATE, "this handle's rows go to its row sink (ro_stft_set_row_sink): pass rows_out = NULL");
    const double t0 = now_ms();
    int64_t got = 0;
    if (first_row_index) *first_row_index = h->rows_emitted;
    if (first_row_index && !h->ready.empty())
        *first_row_index = h->ready.front()->first_row + h->ready.front()->consumed;
    while (got < max_rows && !h->ready.empty()) {
        Batch *b = h->ready.front();
        { const int rc = await_batch(h, b); if (rc != RO_OK) return rc; }
        const int64_t take = std::min(max_rows - got, b->rows - b->consumed);
        for (int64_t r = 0; r < take; ++r) {
            if (rows_out) {
                const float *src = b->data + (size_t)(b->consumed + r) * h->out_cols + (first_col - h->out_first);
                std::memcpy(rows_out + (size_t)(got + r) * cols, src, sizeof(float) * cols);
            }
            if (records_out) records_out[got + r] = b->records[(size_t)(b->consumed + r)];
        }
        b->consumed += take;
        got += take;
        if (b->consumed == b->rows) {
            h->ready.pop_front();
            release_batch(h, b);
        }
    }
    h->rows_ready -= got;
    *rows_got = got;
    const double dt = now_ms() - t0;
    h->timing.fetch_calls += 1;
    h->fetch_ms_sum += dt;
    h->timing.fetch_ms_max = std::max(h->timing.fetch_ms_max, dt);
    return RO_OK;
}

// rows at the head of the output queue whose batches have FINISHED (download included): what ro_stft_fetch hands over
// without waiting.  A caller that fetches only these keeps the next batch's upload and kernels in flight under the
// previous batch's download and under its own per-row work, instead of waiting out every batch it has just launched.
extern "C" int ro_stft_rows_complete(ro_stft_t *h, int64_t *rows)
{
    if (!h || !rows) return fail(RO_ERR_INVALID, "null argument");
    int64_t n = 0;
    for (Batch *b : h->ready) {
        if (b->pending) {
            const hipError_t e = hipEventQuery(b->done);
            if (e == hipErrorNotReady) break;
            if (e != hipSuccess) return fail(RO_ERR_HIP, "hipEventQuery failed: %s", hipGetErrorString(e));
            const int rc = await_batch(h, b);        // finished: book its kernel time once, never query it again
            if (rc != RO_OK) return rc;
        }
        n += b->rows - b->consumed;
    }
    *rows = n;
    return RO_OK;
}

extern "C" int ro_stft_fetch_ln(ro_stft_t *h, int64_t max_rows, float *tile_out, float *ln_out, float *minmax_out,
                                ro_scan_record_t *records_out, int64_t *first_row_index, int64_t *rows_got)
{
    if (!h || !rows_got) return fail(RO_ERR_INVALID, "null argument");
    if (!h->cfg.tile_ln) return fail(RO_ERR_STATE, "this handle was not created with tile_ln");
    if (max_rows < 0) return fail(RO_ERR_INVALID, "negative max_rows");
    if (records_out && !h->cfg.enable_scan) return fail(RO_ERR_STATE, "scan records requested but scan is off");
    const double t0 = now_ms();
    int64_t got = 0;
    if (first_row_index) *first_row_index = h->rows_emitted;
    if (first_row_index && !h->ready.empty())
        *first_row_index = h->ready.front()->first_row + h->ready.front()->consumed;
    const size_t w = (size_t)h->out_cols;
    while (got < max_rows && !h->ready.empty()) {
        Batch *b = h->ready.front();
        { const int rc = await_batch(h, b); if (rc != RO_OK) return rc; }
        const int64_t take = std::min(max_rows - got, b->rows - b->consumed);
        const size_t at = (size_t)b->consumed;
        if (tile_out) std::memcpy(tile_out + (size_t)got * w, b->data + at * w, sizeof(float) * w * (size_t)take);
        if (ln_out) std::memcpy(ln_out + (size_t)got * w, b->ln + at * w, sizeof(float) * w * (size_t)take);
        if (minmax_out) std::memcpy(minmax_out + (size_t)got * 2, b->minmax + at * 2, sizeof(float) * 2 * (size_t)take);
        if (records_out) std::memcpy(records_out + got, b->records + at, sizeof(ro_scan_record_t) * (size_t)take);
        b->consumed += take;
        got += take;
        if (b->consumed == b->rows) {
            h->ready.pop_front();
            release_batch(h, b);
        }
    }
    h->rows_ready -= got;
    *rows_got = got;
    const double dt = now_ms() - t0;
    h->timing.fetch_calls += 1;
    h->fetch_ms_sum += dt;
    h->timing.fetch_ms_max = std::max(h->timing.fetch_ms_max, dt);
    return RO_OK;
}

extern "C" int ro_stft_set_row_sink(ro_stft_t *h, float *base, int64_t row_stride, int64_t capacity_rows, int64_t first_slot)
{
    if (!h) return fail(RO_ERR_INVALID, "null handle");
    if (!h->ready.empty() || h->staged_have > 0)
        return fail(RO_ERR_STATE, "the row sink can only change on an idle stream (after create or ro_stft_reset)");
    if (h->cfg.tile_ln) return fail(RO_ERR_UNSUPPORTED, "a tile_ln handle hands its rows out through ro_stft_fetch_ln");
    HIP_TRY(hipSetDevice(h->device));
    // (batches made with a sink hold no row buffer, batches made without one do: the pool starts over either way)
    while (!h->batch_pool.empty()) { destroy_batch(h->batch_pool.back()); h->batch_pool.pop_back(); }
    h->sink = nullptr;
    if (!base) return RO_OK;
    const int cols = h->cfg.tile_cols > 0 ? h->cfg.tile_cols : h->bins;
    if (row_stride < cols || capacity_rows < 2 * (int64_t)h->batch_rows || first_slot < 0 || first_slot >= capacity_rows)
        return fail(RO_ERR_INVALID, "row sink: stride %lld (rows are %d wide), %lld slots (two batches of %d rows at least), "
                                    "first slot %lld", (long long)row_stride, cols, (long long)capacity_rows, h->batch_rows,
                    (long long)first_slot);
    // The downloads into the ring are asynchronous DMA: the whole range has to be host memory page-locked by THIS
    // process's HIP runtime.  Heap memory is refused here rather than discovered by a copy engine later.
    {
        const size_t bytes = ((size_t)(capacity_rows - 1) * (size_t)row_stride + (size_t)cols) * sizeof(float);
        if (ro_pinned_check(base, bytes) != 1)
            return fail(RO_ERR_INVALID, "row sink: [%p, +%zu bytes) is not page-locked host memory of this process's HIP runtime "
                                        "(use ro_pinned_alloc)", (const void *)base, bytes);
    }
    h->sink = base;
    h->sink_stride = row_stride;
    h->sink_cap = capacity_rows;
    h->sink_first = first_slot;
    return RO_OK;
}

extern "C" int ro_stft_reset(ro_stft_t *h)
{
    if (!h) return fail(RO_ERR_INVALID, "null handle");
    if (h->s_in) (void)hipStreamSynchronize(h->s_in);
    if (h->stream) (void)hipStreamSynchronize(h->stream);
    if (h->s_out) (void)hipStreamSynchronize(h->s_out);
    for (auto &sl : h->slot)
        if (sl.gstream) (void)hipStreamSynchronize(sl.gstream);
    h->staged_have = 0;
    h->stage_fmt_set = false;
    while (!h->ready.empty()) {
        release_batch(h, h->ready.front());
        h->ready.pop_front();
    }
    h->stream_sample0 = 0;
    h->rows_emitted = 0;
    h->rows_ready = 0;
    return RO_OK;
}
