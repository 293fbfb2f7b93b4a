// host_capi.cpp -- plain-C harness over the host mirror, for the ctypes tests only
// (tests/test_ring.py, tests/test_gpu_host_pipeline.py).  Not part of the product ABI.
#include <cstring>
#include <vector>

#include "BolidRecorder.h"
#include "HipWaterfallBackend.h"
#include "RingBuffer.h"

using namespace ro;

extern "C" {

// ---- RingBuffer2D<float>
void *ro_host_ring_create(int width, int chunk, int capacity)
{
    return capacity < 0 ? new RingBuffer2D<float>(width, chunk) : new RingBuffer2D<float>(width, chunk, capacity);
}
void ro_host_ring_destroy(void *r) { delete static_cast<RingBuffer2D<float> *>(r); }
#define RING(r) static_cast<RingBuffer2D<float> *>(r)
int ro_host_ring_capacity(void *r) { return RING(r)->getCapacity(); }
int ro_host_ring_chunk_rows(void *r) { return RING(r)->getChunkRows(); }
int ro_host_ring_get_size(void *r) { return RING(r)->getSize(); }
int ro_host_ring_is_full(void *r) { return RING(r)->isFull() ? 1 : 0; }
int ro_host_ring_push(void *r) { int m = RING(r)->mark(); RING(r)->push(); return m; }
int ro_host_ring_mark(void *r) { return RING(r)->mark(); }
int ro_host_ring_normalize(void *r, int m) { return RING(r)->normalizeRowIndex(m); }
int ro_host_ring_size_from(void *r, int s) { return RING(r)->size(s); }
int ro_host_ring_size_between(void *r, int s, int e) { return RING(r)->size(s, e); }
int ro_host_ring_reserve(void *r, int s, int e) { return RING(r)->reserve(s, e); }
int ro_host_ring_free_reservation(void *r, int h) { return RING(r)->freeReservation(h) ? 1 : 0; }
int ro_host_ring_is_dirty(void *r, int h) { return RING(r)->isDirty(h) ? 1 : 0; }

// ---- Frontend -> HipWaterfallBackend -> BolidRecorder
struct Pipeline {
    HipWaterfallBackend backend;
    BolidRecorder bolid;
    FrontendDriver frontend;
    Pipeline(const WaterfallConfig &w, const BolidConfig &b) : backend(w), bolid(&backend, b), frontend(&backend)
    {
        backend.addRecorder(&bolid);
        backend.keepRowLog(true);
    }
};

void *ro_host_pipeline_create(int bins, int overlap, int sample_rate, int64_t start_sec, int64_t start_usec,
                              int max_batch_rows, int snapshot_length, float lo_det, float hi_det, float lo_noise,
                              float hi_noise, double advance_time, double jitter_time, float avg_range)
{
    WaterfallConfig w;
    w.bins = bins;
    w.overlap = overlap;
    w.max_batch_rows = max_batch_rows;
    BolidConfig b;
    b.snapshot_length = snapshot_length;
    b.low_detect_freq = lo_det;
    b.hi_detect_freq = hi_det;
    b.low_noise_freq = lo_noise;
    b.hi_noise_freq = hi_noise;
    b.advance_time = advance_time;
    b.jitter_time = jitter_time;
    b.avg_freq_range = avg_range;
    Pipeline *p = new Pipeline(w, b);
    StreamInfo si;
    si.sampleRate = sample_rate;
    si.timeOffset = WFTime(start_sec, start_usec);
    p->frontend.startStream(si);
    return p;
}
void ro_host_pipeline_destroy(void *p) { delete static_cast<Pipeline *>(p); }
#define PIPE(p) static_cast<Pipeline *>(p)

// one Frontend::process() call: n complex doubles
void ro_host_pipeline_process(void *p, const double *iq, int n)
{
    std::vector<Complex> v((size_t)n);
    std::memcpy(v.data(), iq, sizeof(Complex) * (size_t)n);
    PIPE(p)->frontend.process(v);
}
void ro_host_pipeline_end(void *p) { PIPE(p)->frontend.endStream(); }
int64_t ro_host_pipeline_rows(void *p) { return PIPE(p)->backend.rowsDelivered(); }
const char *ro_host_pipeline_error(void *p) { return PIPE(p)->backend.lastError().c_str(); }
int ro_host_pipeline_ring_capacity(void *p) { return PIPE(p)->backend.buffer().getCapacity(); }
int ro_host_pipeline_ring_mark(void *p) { return PIPE(p)->backend.buffer().mark(); }
int ro_host_pipeline_raw_capacity(void *p) { return PIPE(p)->backend.rawCapacity(); }
void ro_host_pipeline_ring_row(void *p, int mark, float *out)
{
    std::memcpy(out, PIPE(p)->backend.buffer().at(mark), sizeof(float) * (size_t)PIPE(p)->backend.getBins());
}
int ro_host_pipeline_row_info(void *p, int64_t i, uint64_t *offset, int64_t *sec, int64_t *usec, int *raw_mark)
{
    const auto &log = PIPE(p)->backend.rowLog();
    if (i < 0 || i >= (int64_t)log.size()) return -1;
    *offset = log[(size_t)i].offset;
    *sec = log[(size_t)i].time.sec;
    *usec = log[(size_t)i].time.usec;
    *raw_mark = log[(size_t)i].rawMark;
    return 0;
}
void ro_host_pipeline_raw_handle(void *p, int mark, int *raw_mark, int64_t *sec, int64_t *usec)
{
    const auto &h = PIPE(p)->backend.rawHandles();
    const RawDataHandle &x = h[(size_t)mark % h.size()];
    *raw_mark = x.mark;
    *sec = x.time.sec;
    *usec = x.time.usec;
}
void ro_host_pipeline_bands(void *p, int *out7)
{
    const BolidRecorder &b = PIPE(p)->bolid;
    out7[0] = b.lowDetectBin(); out7[1] = b.detectWidth(); out7[2] = b.lowNoiseBin(); out7[3] = b.noiseWidth();
    out7[4] = b.advance(); out7[5] = b.jitter(); out7[6] = b.averageBinRange();
}
int ro_host_pipeline_events(void *p, BolidEvent *out, int max)
{
    const auto &ev = PIPE(p)->bolid.events();
    const int n = (int)std::min<size_t>(ev.size(), (size_t)max);
    for (int i = 0; i < n; ++i) out[i] = ev[(size_t)i];
    return (int)ev.size();
}
int ro_host_pipeline_state(void *p) { return (int)PIPE(p)->bolid.state(); }

}  // extern "C"
