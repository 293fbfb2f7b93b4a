// host_capi.cpp -- plain-C harness over the product's host-side C++ objects (radio-observer_amd/host/, linked as
// libro_host.so), for the ctypes tests (tests/test_ring.py, tests/test_host_cpu.py, tests/test_gpu_host_pipeline.py)
// and bench.py's streaming leg.  Test infrastructure: not part of the product, not in its libraries.
#include <cmath>
#include <cstdint>
#include <cstring>
#include <ctime>
#include <vector>

#include <sstream>

#include "BolidRecorder.h"
#include "Frontends.h"
#include "HipWaterfallBackend.h"
#include "../../radio-observer_amd/csrc/ro_narrow.h"
#include "RingBuffer.h"
#include "SnapshotRecorder.h"

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
// n pushes at once (row i of the call filled with first + i in column 0); returns the mark before
int ro_host_ring_push_run(void *r, int n, float first)
{
    const int m = RING(r)->mark();
    const int w = RING(r)->getWidth();
    RING(r)->pushRun(n, [&](float *rows, int count, int done) {
        for (int i = 0; i < count; ++i) rows[(size_t)i * w] = first + (float)(done + i);
    });
    return m;
}
void ro_host_ring_push_written(void *r, int n) { RING(r)->pushWritten(n); }
void ro_host_ring_mark_ahead(void *r, int n) { RING(r)->markAhead(n); }
float ro_host_ring_at0(void *r, int mark) { return RING(r)->at(mark)[0]; }
void ro_host_ring_set0(void *r, int mark, float x) { RING(r)->at(mark)[0] = x; }
int ro_host_ring_normalize(void *r, int m) { return RING(r)->normalizeRowIndex(m); }
int ro_host_ring_size_from(void *r, int s) { return RING(r)->size(s); }
int ro_host_ring_size_between(void *r, int s, int e) { return RING(r)->size(s, e); }
int ro_host_ring_reserve(void *r, int s, int e) { return RING(r)->reserve(s, e); }
int ro_host_ring_free_reservation(void *r, int h) { return RING(r)->freeReservation(h) ? 1 : 0; }
int ro_host_ring_is_dirty(void *r, int h) { return RING(r)->isDirty(h) ? 1 : 0; }

// ---- Frontend -> HipWaterfallBackend -> {SnapshotRecorder, BolidRecorder}
struct Rig {
    HipWaterfallBackend backend;
    SnapshotRecorder snap;
    BolidRecorder bolid;
    FrontendDriver frontend;
    Rig(const WaterfallConfig &w, const BolidConfig &b, const SnapshotConfig &sc, bool with_snapshot)
        : backend(w), snap(&backend, sc), bolid(&backend, b), frontend(&backend)
    {
        if (with_snapshot) backend.addRecorder(&snap);       // same order as radio-observer.json:52-88
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
    w.metadata_path = "";
    BolidConfig b;
    b.snapshot_length = snapshot_length;
    b.low_detect_freq = lo_det;
    b.hi_detect_freq = hi_det;
    b.low_noise_freq = lo_noise;
    b.hi_noise_freq = hi_noise;
    b.advance_time = advance_time;
    b.jitter_time = jitter_time;
    b.avg_freq_range = avg_range;
    b.write_files = false;
    Rig *p = new Rig(w, b, SnapshotConfig(), false);
    StreamInfo si;
    si.sampleRate = sample_rate;
    si.timeOffset = WFTime(start_sec, start_usec);
    p->frontend.startStream(si);
    return p;
}
// same, plus a SnapshotRecorder writing FITS files of [lo_snap, hi_snap) Hz every snapshot_length seconds
void *ro_host_pipeline_create_snap(int bins, int overlap, int sample_rate, int64_t start_sec, int64_t start_usec,
                                   int max_batch_rows, int snapshot_length, float lo_snap, float hi_snap,
                                   const char *out_dir, const char *origin)
{
    WaterfallConfig w;
    w.bins = bins;
    w.overlap = overlap;
    w.max_batch_rows = max_batch_rows;
    w.origin = origin;
    w.metadata_path = out_dir;
    SnapshotConfig sc;
    sc.output_dir = out_dir;
    sc.snapshot_length = snapshot_length;
    sc.low_freq = lo_snap;
    sc.hi_freq = hi_snap;
    BolidConfig b;
    b.output_dir = out_dir;
    b.snapshot_length = snapshot_length;
    b.low_detect_freq = 10300; b.hi_detect_freq = 10900; b.low_noise_freq = 9000; b.hi_noise_freq = 9600;
    b.advance_time = 2; b.jitter_time = 5;
    Rig *p = new Rig(w, b, sc, true);
    StreamInfo si;
    si.sampleRate = sample_rate;
    si.timeOffset = WFTime(start_sec, start_usec);
    p->frontend.startStream(si);
    return p;
}
int ro_host_pipeline_files(void *p, char *buf, int len)
{
    std::string all;
    for (const auto &f : static_cast<Rig *>(p)->snap.filesWritten()) all += f + "\n";
    std::snprintf(buf, (size_t)len, "%s", all.c_str());
    return (int)static_cast<Rig *>(p)->snap.filesWritten().size();
}
static int joinNames(const std::vector<std::string> &v, char *buf, int len)
{
    std::string all;
    for (const auto &f : v) all += f + "\n";
    std::snprintf(buf, (size_t)len, "%s", all.c_str());
    return (int)v.size();
}
// files of the detector: raw = 0 the band snapshots ("blid"), raw = 1 the raw I/Q captures ("raws")
int ro_host_pipeline_bolid_files(void *p, int raw, char *buf, int len)
{
    const BolidRecorder &b = static_cast<Rig *>(p)->bolid;
    return joinNames(raw ? b.rawFilesWritten() : b.filesWritten(), buf, len);
}
// pin WFTime::now() for the event lines; the metadata CSV's current file name ("" before the first entry)
void ro_host_pipeline_set_clock(void *p, int64_t sec, int64_t usec)
{
    static_cast<Rig *>(p)->backend.setClock(WFTime(sec, usec));
}
int ro_host_pipeline_metadata_file(void *p, char *buf, int len)
{
    CsvLog *log = static_cast<Rig *>(p)->backend.getMetadataFile();
    std::snprintf(buf, (size_t)len, "%s", log ? log->currentFile().c_str() : "");
    return log ? 1 : 0;
}
void ro_host_pipeline_destroy(void *p) { delete static_cast<Rig *>(p); }
#define PIPE(p) static_cast<Rig *>(p)

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
int ro_host_pipeline_batch_rows(void *p) { return PIPE(p)->backend.batchRows(); }


// ---- frontends against a recording backend (no GPU): what exactly does Backend::process() receive?
struct RecordingBackend : public Backend {
    std::vector<int> callSizes;
    std::vector<DataInfo> infos;
    std::vector<Complex> samples;
    int started = 0, ended = 0;
    void startStream(StreamInfo info) override { Backend::startStream(info); started++; }
    void process(const std::vector<Complex> &data, DataInfo info) override
    {
        callSizes.push_back((int)data.size());
        infos.push_back(info);
        samples.insert(samples.end(), data.begin(), data.end());
    }
    void endStream() override { ended++; }
};

struct FrontendRun {
    RecordingBackend backend;
    WAVFormat format;
    std::string error, inf1;
    bool ok = false;
};

// kind 0: WAVStream over `bytes`; kind 1: RawStream (float32 I,Q) at `sample_rate`
void *ro_host_frontend_run(int kind, const char *bytes, int64_t n, int sample_rate, int64_t start_sec,
                           int64_t start_usec)
{
    FrontendRun *r = new FrontendRun();
    std::istringstream in(std::string(bytes, (size_t)n));
    // through Pipeline, the way the reference's main() wires them (src/Pipeline.cpp:11-19)
    ro::Pipeline pipe;
    pipe.setBackend(&r->backend);
    if (kind == 0) {
        WAVStream w(in);
        pipe.setFrontend(&w);
        pipe.run();
        r->ok = w.ok();
        r->format = w.format();
        r->error = w.lastError();
        r->inf1 = w.inf1();
    } else {
        RawStream w(in, nullptr, sample_rate, WFTime(start_sec, start_usec));
        pipe.setFrontend(&w);
        pipe.run();
        r->ok = true;
    }
    return r;
}
void ro_host_frontend_free(void *r) { delete static_cast<FrontendRun *>(r); }
#define FR(r) static_cast<FrontendRun *>(r)
int ro_host_frontend_ok(void *r) { return FR(r)->ok ? 1 : 0; }
const char *ro_host_frontend_error(void *r) { return FR(r)->error.c_str(); }
const char *ro_host_frontend_inf1(void *r) { return FR(r)->inf1.c_str(); }
int ro_host_frontend_calls(void *r) { return (int)FR(r)->backend.callSizes.size(); }
int ro_host_frontend_started(void *r) { return FR(r)->backend.started * 10 + FR(r)->backend.ended; }
int ro_host_frontend_sample_rate(void *r) { return FR(r)->backend.getStreamInfo().sampleRate; }
void ro_host_frontend_format(void *r, int *out6)
{
    const WAVFormat &f = FR(r)->format;
    out6[0] = f.audioFormat; out6[1] = f.channelCount; out6[2] = f.sampleRate; out6[3] = f.byteRate;
    out6[4] = f.blockAlign; out6[5] = f.bitsPerSample;
}
int ro_host_frontend_call(void *r, int i, uint64_t *offset, int64_t *sec, int64_t *usec)
{
    *offset = FR(r)->backend.infos[(size_t)i].offset;
    *sec = FR(r)->backend.infos[(size_t)i].timeOffset.sec;
    *usec = FR(r)->backend.infos[(size_t)i].timeOffset.usec;
    return FR(r)->backend.callSizes[(size_t)i];
}
int64_t ro_host_frontend_samples(void *r, double *out, int64_t max)
{
    const auto &s = FR(r)->backend.samples;
    const int64_t n = std::min<int64_t>((int64_t)s.size(), max);
    if (out && n > 0) std::memcpy(out, s.data(), sizeof(Complex) * (size_t)n);
    return (int64_t)s.size();
}

// ---- C1 end to end: WAV bytes -> WAVStream -> HipWaterfallBackend (GPU) -> SnapshotRecorder -> FITS files
int64_t ro_host_wav_to_fits(const char *bytes, int64_t n, int bins, int overlap, int max_batch_rows,
                            int snapshot_length, float lo, float hi, const char *out_dir, const char *origin,
                            char *files, int files_len, char *err, int err_len)
{
    WaterfallConfig w;
    w.bins = bins;
    w.overlap = overlap;
    w.max_batch_rows = max_batch_rows;
    w.origin = origin;
    w.metadata_path = out_dir;
    SnapshotConfig sc;
    sc.output_dir = out_dir;
    sc.snapshot_length = snapshot_length;
    sc.low_freq = lo;
    sc.hi_freq = hi;
    HipWaterfallBackend backend(w);
    SnapshotRecorder snap(&backend, sc);
    backend.addRecorder(&snap);
    std::istringstream in(std::string(bytes, (size_t)n));
    WAVStream wav(in);
    ro::Pipeline pipe;                                   // Frontend -> Pipeline -> Backend, as in the reference's main()
    pipe.setFrontend(&wav);
    pipe.setBackend(&backend);
    pipe.run();
    const bool ok = wav.ok();
    std::string all;
    for (const auto &f : snap.filesWritten()) all += f + "\n";
    std::snprintf(files, (size_t)files_len, "%s", all.c_str());
    std::snprintf(err, (size_t)err_len, "%s%s", ok ? "" : wav.lastError().c_str(), backend.lastError().c_str());
    return backend.rowsDelivered();
}

// ---- recorders on hand-fed rows (no GPU): cadence, FITS files, detector
struct ManualRig {
    ManualWaterfall source;
    SnapshotRecorder snap;
    BolidRecorder bolid;
    std::ostringstream lines;
    ManualRig(const WaterfallConfig &w, const SnapshotConfig &s, const BolidConfig &b)
        : source(w), snap(&source, s), bolid(&source, b)
    {
        source.addRecorder(&snap);
        source.addRecorder(&bolid);
        bolid.setOutput(&lines);
    }
};
void *ro_host_manual_create(int bins, int overlap, int sample_rate, int snapshot_length, float lo_snap, float hi_snap,
                            const char *out_dir, const char *origin, double advance_time, double jitter_time)
{
    WaterfallConfig w;
    w.bins = bins;
    w.overlap = overlap;
    w.origin = origin;
    w.metadata_path = out_dir;
    SnapshotConfig sc;
    sc.output_dir = out_dir;
    sc.snapshot_length = snapshot_length;
    sc.low_freq = lo_snap;
    sc.hi_freq = hi_snap;
    BolidConfig b;
    b.output_dir = out_dir;
    b.snapshot_length = snapshot_length;
    b.low_detect_freq = 10300; b.hi_detect_freq = 10900; b.low_noise_freq = 9000; b.hi_noise_freq = 9600;
    b.advance_time = advance_time; b.jitter_time = jitter_time;
    ManualRig *m = new ManualRig(w, sc, b);
    StreamInfo si;
    si.sampleRate = sample_rate;
    m->source.startStream(si);
    return m;
}
#define RIG(m) static_cast<ManualRig *>(m)
void ro_host_manual_destroy(void *m) { delete RIG(m); }
void ro_host_manual_push(void *m, const float *row, float n, int p, float a, int64_t sec, int64_t usec, int raw_mark)
{
    ro_scan_record_t s{n, p, a};
    RIG(m)->source.pushRow(row, &s, WFTime(sec, usec), raw_mark);
}
// raw samples behind the rows (n complex doubles), as FFTBackend::process keeps them
void ro_host_manual_push_samples(void *m, const double *iq, int n)
{
    RIG(m)->source.pushSamples(reinterpret_cast<const Complex *>(iq), (size_t)n);
}
int ro_host_manual_bolid_files(void *m, int raw, char *buf, int len)
{
    const BolidRecorder &b = RIG(m)->bolid;
    return joinNames(raw ? b.rawFilesWritten() : b.filesWritten(), buf, len);
}
void ro_host_manual_set_clock(void *m, int64_t sec, int64_t usec) { RIG(m)->source.setClock(WFTime(sec, usec)); }
int ro_host_manual_metadata_file(void *m, char *buf, int len)
{
    CsvLog *log = RIG(m)->source.getMetadataFile();
    std::snprintf(buf, (size_t)len, "%s", log ? log->currentFile().c_str() : "");
    return log ? 1 : 0;
}
// the "met;...#" lines of src/BolidRecorder.cpp:250-257 collected so far
int ro_host_manual_stdout(void *m, char *buf, int len)
{
    std::snprintf(buf, (size_t)len, "%s", RIG(m)->lines.str().c_str());
    return (int)RIG(m)->lines.str().size();
}
int ro_host_manual_raw_capacity(void *m) { return RIG(m)->source.rawBuffer().getCapacity(); }
// sample `mark` of the raw ring (a float pair), as the recorders' raw capture reads it
void ro_host_manual_raw_at(void *m, int mark, float *out2)
{
    const float *p = RIG(m)->source.rawBuffer().at(mark);
    out2[0] = p[0];
    out2[1] = p[1];
}
int ro_host_manual_raw_mark(void *m) { return RIG(m)->source.rawBuffer().mark(); }
// the narrowing loops of pushRaw one level at a time (csrc/ro_narrow.h): returns the best level this CPU has
int ro_host_narrow(int level, const double *src, float *dst, int count)
{
    // (levels 10 + l: the non-temporal form of level l)
    if (level >= 10 && level - 10 <= narrowLevel()) narrowWith(level - 10, true, src, dst, count);
    else if (level >= 0 && level <= narrowLevel()) narrowWith(level, src, dst, count);
    return narrowLevel();
}
void ro_host_manual_end(void *m) { RIG(m)->source.endStream(); }
int ro_host_manual_info(void *m, int *out6)
{
    out6[0] = RIG(m)->source.buffer().getCapacity();
    out6[1] = RIG(m)->snap.snapshotRows();
    out6[2] = RIG(m)->snap.leftBin();
    out6[3] = RIG(m)->snap.rightBin();
    out6[4] = (int)RIG(m)->snap.snapshotsQueued().size();
    out6[5] = (int)RIG(m)->bolid.state();
    return 0;
}
int ro_host_manual_files(void *m, char *buf, int len)
{
    std::string all;
    for (const auto &f : RIG(m)->snap.filesWritten()) all += f + "\n";
    std::snprintf(buf, (size_t)len, "%s", all.c_str());
    return (int)RIG(m)->snap.filesWritten().size();
}
int ro_host_manual_snapshot(void *m, int i, int *start, int *length)
{
    const auto &q = RIG(m)->snap.snapshotsQueued();
    if (i < 0 || i >= (int)q.size()) return -1;
    *start = q[(size_t)i].start;
    *length = q[(size_t)i].length;
    return 0;
}
int ro_host_manual_events(void *m, BolidEvent *out, int max)
{
    const auto &ev = RIG(m)->bolid.events();
    const int n = (int)std::min<size_t>(ev.size(), (size_t)max);
    for (int i = 0; i < n; ++i) out[i] = ev[(size_t)i];
    return (int)ev.size();
}


// ---- the product's BolidRecorder driven by a stream of scan records (the stitched (n, p, a) stream of a
// multi-GPU run: tests/test_gpu_c5.py): no files, zero rows in the ring, events out
int64_t ro_host_bolid_replay(int bins, int overlap, int sample_rate, float lo_det, float hi_det, float lo_noise,
                             float hi_noise, double advance_time, double jitter_time, float avg_range,
                             const ro_scan_record_t *recs, int64_t n, BolidEvent *out, int max)
{
    WaterfallConfig w;
    w.bins = bins;
    w.overlap = overlap;
    w.metadata_path = "";
    w.keep_raw = false;
    BolidConfig b;
    b.write_files = false;
    b.low_detect_freq = lo_det; b.hi_detect_freq = hi_det; b.low_noise_freq = lo_noise; b.hi_noise_freq = hi_noise;
    b.advance_time = advance_time; b.jitter_time = jitter_time; b.avg_freq_range = avg_range;
    ManualWaterfall source(w);
    BolidRecorder bolid(&source, b);
    source.addRecorder(&bolid);
    StreamInfo si;
    si.sampleRate = sample_rate;
    source.startStream(si);
    const std::vector<float> row((size_t)bins, 0.f);
    const int hop = bins - ro_clamp_overlap(bins, overlap);
    WFTime t(0, 0);
    for (int64_t r = 0; r < n; ++r) {
        source.pushRow(row.data(), &recs[r], t, (int)((r + 1) * hop + 1));
        t = t.addSamples(hop, sample_rate);
    }
    source.endStream();
    const auto &ev = bolid.events();
    const int64_t m = std::min<int64_t>((int64_t)ev.size(), max);
    for (int64_t i = 0; i < m; ++i) out[i] = ev[(size_t)i];
    return (int64_t)ev.size();
}

// ---- bench.py's streaming leg: the drop-in path at full speed.  Frontend::process -> HipWaterfallBackend::process
// (ro_stft_push of RO_IQ_F64 into pinned staging, kernels, ro_stft_fetch of full rows) -> Recorder::update per row,
// driven exactly as src/RawStream.cpp:44-66 drives the reference: `block` samples of vector<Complex> per call.
// Handle creation (startStream) and the first `warm_calls` calls are outside the timed region; the region ends
// behind endStream, when every row has been through BolidRecorder::update.
// stats: [0] seconds  [1] samples  [2] rows delivered  [3] process() calls  [4] rows per kernel launch
//        [5] mean ms per process() call  [6] max ms per process() call  [7] events fired
//
// RO_HOST_SAMPLE=<file>: a program-counter sampler over the timed region (a 4 kHz timer on the calling thread's CPU
// time): where the host thread spends its time, by shared object and nearest exported symbol -- there is no perf on the
// GPU boxes.  Diagnostic only; tools/r5/host_sample.sh runs it.
}  // extern "C"
#include <dlfcn.h>
#include <signal.h>
#include <sys/syscall.h>
#include <sys/time.h>
#include <ucontext.h>
#include <unistd.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <string>
namespace {
constexpr int SAMPLE_MAX = 1 << 16;
void *g_samples[SAMPLE_MAX];
volatile int g_nsamples = 0;
void on_prof(int, siginfo_t *, void *uc)
{
    const int i = g_nsamples;
    if (i < SAMPLE_MAX) {
        g_samples[i] = (void *)((ucontext_t *)uc)->uc_mcontext.gregs[REG_RIP];
        g_nsamples = i + 1;
    }
}
timer_t g_timer;
void sampler_start()
{
    g_nsamples = 0;
    struct sigaction sa;
    std::memset(&sa, 0, sizeof sa);
    sa.sa_sigaction = on_prof;
    sa.sa_flags = SA_SIGINFO | SA_RESTART;
    sigaction(SIGPROF, &sa, nullptr);
    sigevent sev;                                     // this thread's CPU time, delivered to this thread
    std::memset(&sev, 0, sizeof sev);
    sev.sigev_notify = SIGEV_THREAD_ID;
    sev.sigev_signo = SIGPROF;
    sev._sigev_un._tid = (pid_t)syscall(SYS_gettid);
    timer_create(CLOCK_THREAD_CPUTIME_ID, &sev, &g_timer);
    itimerspec its;
    its.it_interval.tv_sec = 0;  its.it_interval.tv_nsec = 250000;
    its.it_value = its.it_interval;
    timer_settime(g_timer, 0, &its, nullptr);
}
void sampler_stop(const char *path)
{
    timer_delete(g_timer);
    signal(SIGPROF, SIG_IGN);
    std::map<std::string, int> hist;
    for (int i = 0; i < g_nsamples; ++i) {
        Dl_info di;
        std::string key = "?";
        if (dladdr(g_samples[i], &di) && di.dli_fname) {
            const char *base = std::strrchr(di.dli_fname, '/');
            key = base ? base + 1 : di.dli_fname;
            key += " : ";
            key += di.dli_sname ? di.dli_sname : "(static)";
            if (!di.dli_sname) {
                char off[32];
                std::snprintf(off, sizeof off, " +0x%zx", (size_t)((char *)g_samples[i] - (char *)di.dli_fbase) & ~(size_t)0xff);
                key += off;
            }
        }
        hist[key] += 1;
    }
    std::vector<std::pair<int, std::string>> v;
    for (auto &kv : hist) v.push_back({kv.second, kv.first});
    std::sort(v.rbegin(), v.rend());
    if (FILE *f = std::fopen(path, "w")) {
        std::fprintf(f, "# %d samples (250 us of the calling thread's CPU time each)\n", g_nsamples);
        for (auto &e : v) std::fprintf(f, "%6d %5.1f%%  %s\n", e.first, 100.0 * e.first / std::max(1, (int)g_nsamples), e.second.c_str());
        std::fclose(f);
    }
}
}  // namespace
extern "C" {
int ro_host_stream_bench(int bins, int overlap, int sample_rate, int block, double seconds, int max_batch_rows,
                         int warm_calls, double *stats)
{
    if (bins <= 0 || block <= 0 || seconds <= 0 || !stats) return -1;
    WaterfallConfig w;
    w.bins = bins;
    w.overlap = overlap;
    w.max_batch_rows = max_batch_rows;
    w.metadata_path = "";
    BolidConfig b;                                   // radio-observer.json:62-87
    b.snapshot_length = 60;
    b.low_detect_freq = 10300; b.hi_detect_freq = 10900; b.low_noise_freq = 9000; b.hi_noise_freq = 9600;
    b.advance_time = 2; b.jitter_time = 5; b.avg_freq_range = 40;
    b.write_files = false;
    HipWaterfallBackend backend(w);
    BolidRecorder bolid(&backend, b);
    backend.addRecorder(&bolid);
    FrontendDriver frontend(&backend);
    // blocks of sigma = 1 noise (the content does not change the work).  The reference's frontends hand over ONE vector
    // they have just filled (src/RawStream.cpp:36, :59-66: `outputBuffer`, 4096 samples = 64 KiB, rewritten before every
    // process() call), so what Backend::process reads is in the calling core's caches: two blocks alternate here.
    // (Rounds 3-5 rotated sixteen -- 1 MiB, which a core's 1 MiB L2 does not hold next to everything else, so every
    // call read its samples from L3; RO_STREAM_NBLK=16 brings that back for the record.)
    const char *nblk_env = std::getenv("RO_STREAM_NBLK");
    const int NBLK = nblk_env && std::atoi(nblk_env) > 0 ? std::atoi(nblk_env) : 2;
    std::vector<std::vector<Complex>> blocks((size_t)NBLK, std::vector<Complex>((size_t)block));
    uint64_t lcg = 0x9E3779B97F4A7C15ull;
    auto uni = [&]() { lcg = lcg * 6364136223846793005ull + 1442695040888963407ull; return (double)(lcg >> 11) * (1.0 / 9007199254740992.0); };
    for (auto &blk : blocks)
        for (auto &c : blk) {
            const double u1 = uni() + 1e-300, u2 = uni();
            const double r = std::sqrt(-2.0 * std::log(u1));
            c.real = (float)(r * std::cos(6.283185307179586 * u2));
            c.imag = (float)(r * std::sin(6.283185307179586 * u2));
        }
    StreamInfo si;
    si.sampleRate = sample_rate;
    frontend.startStream(si);
    if (!backend.lastError().empty()) return -2;
    int64_t calls = 0;
    for (int i = 0; i < warm_calls; ++i) frontend.process(blocks[(size_t)(calls++ % NBLK)]);
    const int64_t rows0 = backend.rowsDelivered();
    ro_stft_timing_t tm;
    backend.timing(&tm, true);                                       // the counters of the timed region only
    auto now = []() { timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec; };
    const char *sample_path = std::getenv("RO_HOST_SAMPLE");
    if (sample_path) sampler_start();
    const double t0 = now();
    double worst = 0.0;
    int64_t timed = 0;
    for (;;) {
        const double a = now();
        frontend.process(blocks[(size_t)(calls++ % NBLK)]);
        const double d = now() - a;
        worst = d > worst ? d : worst;
        ++timed;
        if (a + d - t0 >= seconds) break;
    }
    std::memset(&tm, 0, sizeof tm);
    backend.timing(&tm, false);
    frontend.endStream();
    const double dt = now() - t0;
    if (sample_path) sampler_stop(sample_path);
    stats[8] = tm.push_ms_avg;  stats[9] = tm.fetch_ms_avg;  stats[10] = tm.batch_gpu_ms_avg;  stats[11] = tm.row_gpu_us_avg;
    stats[12] = (double)tm.push_calls;  stats[13] = (double)tm.fetch_calls;  stats[14] = (double)tm.batches;
    stats[15] = backend.rowsByDma() ? 1.0 : 0.0;
    stats[0] = dt;
    stats[1] = (double)timed * (double)block;
    stats[2] = (double)(backend.rowsDelivered() - rows0);
    stats[3] = (double)timed;
    stats[4] = (double)backend.batchRows();
    stats[5] = 1e3 * dt / (double)timed;
    stats[6] = 1e3 * worst;
    stats[7] = (double)bolid.events().size();
    return backend.lastError().empty() ? 0 : -3;
}

}  // extern "C"
