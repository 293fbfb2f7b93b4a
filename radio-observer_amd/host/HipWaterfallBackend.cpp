#if defined(__SSE2__)
#include <emmintrin.h>
#endif
#include "HipWaterfallBackend.h"
#include "../csrc/ro_narrow.h"

#include <sys/time.h>

#include <algorithm>
#include <cstdio>
#include <cmath>
#include <cstring>

namespace ro {

static int wrapIndex(int value, int size)            // src/utils.cpp:12-18
{
    while (value < 0) value += size;
    return value % size;
}

int Recorder::getSampleRate() const { return backend_->streamInfo().sampleRate; }
int Recorder::getFFTSampleRate() const { return (int)backend_->getFFTSampleRate(); }
int Recorder::fftMarkToRaw(int mark) const { return (*rawHandles_)[wrapIndex(mark, (int)rawHandles_->size())].mark; }
WFTime Recorder::fftMarkToTime(int mark) const { return (*rawHandles_)[wrapIndex(mark, (int)rawHandles_->size())].time; }

// the row ring of a GPU-fed backend lives in page-locked memory: its rows arrive by DMA (ro_stft_set_row_sink)
static void *pinnedAlloc(void *ctx, size_t bytes) { return ro_pinned_alloc(*static_cast<int *>(ctx), bytes); }
static void pinnedFree(void *, void *p) { ro_pinned_free(p); }

WaterfallBase::WaterfallBase(const WaterfallConfig &cfg) : cfg_(cfg)
{
    bins_ = cfg.bins;
    overlap_ = ro_clamp_overlap(cfg.bins, cfg.overlap);          // src/FFTBackend.cpp:108-109
    hop_ = bins_ - overlap_;
}

void WaterfallBase::addRecorder(Recorder *recorder)
{
    recorders_.push_back(recorder);
    recorder->setBuffer(&buffer_, &rawBuffer_, &bufferMutex_, &rawHandles_);      // src/WaterfallBackend.cpp:566
}

bool WaterfallBase::beginStream(const StreamInfo &info, ro_bands_t *bands)
{
    info_ = info;
    fftSampleRate_ = ro_fft_sample_rate(info.sampleRate, bins_, overlap_);     // FFTBackend.cpp:150-151
    rowsDelivered_ = 0;
    rowsFetched_ = 0;
    rowLog_.clear();
    // src/WaterfallBackend.cpp:577-588
    int bufferSize = 1;
    for (Recorder *r : recorders_) bufferSize = std::max(bufferSize, r->requestBufferSize());
    buffer_.resize(bins_, cfg_.buffer_chunk_size, bufferSize);
    rawHandles_.assign(buffer_.getCapacity(), RawDataHandle());
    {   // resizeRawBuffer(fftSamplesToRaw(bufferSize)): RingBuffer2D<float>(2, 1 MiB, n) (FFTBackend.h:129-132)
        const int want = fftSamplesToRaw(bufferSize);
        const int chunkRows = (1024 * 1024) / 8;
        rawCapacity_ = std::max(1, (want / chunkRows + (want % chunkRows ? 1 : 0)) * chunkRows);
        if (cfg_.keep_raw) rawBuffer_.resize(2, 1024 * 1024, want);
    }
    for (Recorder *r : recorders_) r->start();                                   // :591-593
    bool any = false;
    for (Recorder *r : recorders_)
        if (!any && r->scanBands(bands)) any = true;
    return any;
}

CsvLog *WaterfallBase::getMetadataFile()
{
    if (cfg_.metadata_path.empty()) return nullptr;
    if (!metadataFile_) {
        std::string p = cfg_.metadata_path;
        if (p.back() != '/') p += "/";
        metadataFile_.reset(new CsvLog(p + "%Y%m%d%H%M%S_" + cfg_.origin + "_meta.csv",
                                       "file name; noise; peak f.; mag.; duration"));
    }
    return metadataFile_.get();
}

WFTime WaterfallBase::now() const
{
    if (useFixedClock_) return fixedClock_;
    timeval tv;
    gettimeofday(&tv, nullptr);
    return WFTime((int64_t)tv.tv_sec, (int64_t)tv.tv_usec);
}

void WaterfallBase::pushRaw(const Complex *data, size_t n)
{
    if (!cfg_.keep_raw || rawBuffer_.getCapacity() == 0) return;
    // one ring row per sample (FFTBackend::floatToInt(Complex, float*), FFTBackend.h:258-262), a call's worth at a time:
    // push() per sample cost more than everything else Backend::process does on the host
    rawBuffer_.pushRun((int)n, [&](float *rows, int count, int done) {
        // struct Complex = {double real, imag}; written past the caches: the ring is read again only by an event's capture
        narrowToFloatStream(&data[done].real, rows, 2 * count);
    });
}

void WaterfallBase::finishStream()
{
    for (Recorder *r : recorders_) r->stop();                                    // :604-606
}

// WaterfallBackend::processFFT minus the arithmetic (src/WaterfallBackend.cpp:485-541)
void WaterfallBase::processRow(const float *row, const ro_scan_record_t *scan, DataInfo info, int rawMark)
{
    {
        // the ring's bookkeeping is shared with the recorders' worker threads (reservations, size); the row itself is
        // written into a slot no queued snapshot covers
        std::lock_guard<std::mutex> g(bufferMutex_);
        float *dst = buffer_.push();                                             // :488
        if (row) std::memcpy(dst, row, sizeof(float) * (size_t)bins_);           // (nullptr: already there -- the row sink)
        rawHandles_[buffer_.mark()] = RawDataHandle(rawMark, info.timeOffset);   // :507 (one slot ahead)
    }
    if (scan) currentScan_ = *scan;
    rowsDelivered_++;
    if (keepLog_) rowLog_.push_back(RowInfo{info.offset, info.timeOffset, rawMark});
    for (Recorder *r : recorders_) r->update();                                  // :534-536
}

HipWaterfallBackend::HipWaterfallBackend(const WaterfallConfig &cfg) : WaterfallBase(cfg)
{
    buffer_.setStorage(pinnedAlloc, pinnedFree, &cfg_.device);       // (without a device: nullptr, i.e. the heap)
}

HipWaterfallBackend::~HipWaterfallBackend()
{
    if (stft_) ro_stft_destroy(stft_);
}

void HipWaterfallBackend::startStream(StreamInfo info)
{
    Backend::startStream(info);
    samplesIn_ = 0;
    inMark_ = 0;
    nextStampRow_ = 0;
    rowTimes_.clear();

    // the device side: one ro_stft handle per stream; scan bands from whichever recorder wants them
    if (stft_) { ro_stft_destroy(stft_); stft_ = nullptr; }
    ro_stft_config_t c;
    std::memset(&c, 0, sizeof(c));
    scanEnabled_ = beginStream(info, &c.bands);
    c.struct_size = sizeof(c);
    c.bins = bins_;
    c.overlap = overlap_;
    c.sample_rate = info.sampleRate;
    c.window_kind = RO_WINDOW_NUTTALL;
    c.iq_gain = cfg_.iq_gain;
    c.iq_phase_shift = cfg_.iq_phase_shift;
    c.device = cfg_.device;
    // Rows reach Recorder::update() a batch late (the GPU wants more than one row per launch; recorders only look
    // backwards).  The library's own default batch is sized for throughput (~64 MiB of rows: 87 s of stream at
    // N = 32768 / 75 %); behind the Backend interface the default is bounded by LATENCY instead: at most one second
    // of rows, and never more than an eighth of the row ring or a quarter of the raw-sample ring, so that an event's
    // snapshot and raw capture (which reach back `advance` rows from the row being delivered) still find their data.
    batchRows_ = cfg_.max_batch_rows;
    if (batchRows_ <= 0) {
        const int64_t one_second = (int64_t)std::ceil((double)fftSampleRate_);
        const int64_t ring = buffer_.getCapacity() / 8;
        const int64_t raw = cfg_.keep_raw ? (int64_t)rawCapacity_ / hop_ / 4 : one_second;
        batchRows_ = (int)std::max<int64_t>(1, std::min<int64_t>(std::min(one_second, ring), std::min<int64_t>(raw, 4096)));
    }
    if (cfg_.keep_raw && (int64_t)batchRows_ * hop_ + bins_ > rawCapacity_)
        std::fprintf(stderr, "HipWaterfallBackend: batch of %d rows (%lld samples) exceeds the raw ring (%d samples): "
                             "raw captures of events will hold newer samples than their rows\n",
                     batchRows_, (long long)batchRows_ * hop_ + bins_, rawCapacity_);
    c.max_batch_rows = batchRows_;
    // batches that may be in flight and not yet handed to the recorders (drain): as many (up to three) as keep the lag --
    // those in flight plus the one being staged -- a small part of both rings
    auto lagFits = [&](int64_t batches) {       // `batches` of lag (in flight + the one being staged) stay a small part of both rings
        return batches * (int64_t)batchRows_ <= buffer_.getCapacity() / 4 &&
               (!cfg_.keep_raw || batches * (int64_t)batchRows_ * hop_ + bins_ <= rawCapacity_ / 2);
    };
    // (three at the most: the streaming path rotates through three slots of device and staging buffers)
    maxOutstanding_ = lagFits(4) ? 3 : lagFits(3) ? 2 : 1;
    c.enable_scan = scanEnabled_ ? 1 : 0;
    if (ro_stft_create(&c, &stft_) != RO_OK) {
        // the reference logs and carries on (LOG_ERROR + return); so does this: no rows will come
        lastError_ = ro_last_error();
        std::fprintf(stderr, "HipWaterfallBackend: %s\n", lastError_.c_str());
        stft_ = nullptr;
        return;
    }
    // Finished rows go from the GPU straight into the row ring's slots -- processFFT's write into buffer_->push()
    // (src/WaterfallBackend.cpp:488-505) is the device-to-host copy itself: row r of the stream lands in slot
    // (mark at the start of the stream + r) mod capacity, where the push() of processRow then finds it.  Without
    // page-locked memory for the ring (or a ring shorter than two batches) the rows take the copying path
    // (ro_stft_fetch into fetchRows_).
    rowSink_ = buffer_.storageIsCustom() &&
               ro_stft_set_row_sink(stft_, buffer_.data(), bins_, buffer_.getCapacity(), buffer_.mark()) == RO_OK;
}

// Times of row starts.  FFTBackend::process stamps every sample with
// timeOffset.addSamples(i) while it copies a "take" (the samples that complete the next row),
// then advances timeOffset by the take (src/FFTBackend.cpp:216-223, :255); a row's time is the
// stamp of its first sample (:225).  Only row starts are needed, so they are computed per take.
void HipWaterfallBackend::stampRowStarts(int64_t takeBegin, int64_t takeEnd, const WFTime &t)
{
    while (nextStampRow_ * (int64_t)hop_ < takeEnd) {
        const int64_t s = nextStampRow_ * (int64_t)hop_;
        if (s >= takeBegin) rowTimes_.push_back(t.addSamples((SampleCount)(s - takeBegin), info_.sampleRate));
        nextStampRow_++;
    }
}

void HipWaterfallBackend::process(const std::vector<Complex> &data, DataInfo info)
{
    if (!stft_ || data.empty()) return;
    // ---- host bookkeeping, same control flow as src/FFTBackend.cpp:209-273 (no per-sample work)
    {
        int64_t size = (int64_t)data.size();
        int64_t pos = samplesIn_;
        WFTime timeOffset = info.timeOffset;
        while (size >= bins_ - inMark_) {
            const int count = bins_ - inMark_;
            stampRowStarts(pos, pos + count, timeOffset);
            inMark_ = overlap_;
            size -= count;
            pos += count;
            timeOffset = timeOffset.addSamples((SampleCount)count, info_.sampleRate);
        }
        if (size > 0) {
            stampRowStarts(pos, pos + size, timeOffset);
            inMark_ += (int)size;
        }
        samplesIn_ += (int64_t)data.size();
    }
    // ---- the samples themselves go to the GPU path.  struct Complex is two doubles: they are narrowed twice from the
    // caller's (cache-hot) vector, once into the raw ring with stores that go past the caches and once, by ro_stft_push
    // (RO_IQ_F64), straight into the pinned staging buffer the upload reads.  (Round 4 narrowed into the raw ring and pushed
    // its float rows: the second pass was a memcpy out of a ring that every store had first to read from memory -- the PC
    // samples of tools/r5/host_sample.py had 75 % of the host thread in those two loops at 256 rows per launch.)
    // A call of ordinary size goes over in one piece; a very long one (a file replayed in one block) in pieces that each
    // complete at most a quarter of the row ring, the rows handed to the recorders in between -- the rows of a batch
    // land in the ring's slots before they are fetched, so what is in flight has to stay well inside it.
    const size_t piece = std::max<size_t>((size_t)hop_, (size_t)(buffer_.getCapacity() / 4) * (size_t)hop_);
    for (size_t at = 0; at < data.size(); at += piece) {
        const Complex *src = data.data() + at;
        const size_t n = std::min(piece, data.size() - at);
        int64_t ready = 0;
        if (rowSink_) {
            // The rows this piece completes may be on their way into the slots ahead of the ring's head as soon as
            // ro_stft_push has queued their batch: whoever holds a reservation there (a queued snapshot that has
            // fallen a whole ring behind) has to learn it BEFORE the DMA can start, not when the rows are handed
            // over.  (Rows that complete without filling a batch are marked a batch early: conservative.)
            const int64_t upTo = samplesIn_ - (int64_t)data.size() + (int64_t)(at + n);
            const int64_t complete = upTo >= bins_ ? (upTo - bins_) / hop_ + 1 : 0;
            int64_t ahead = complete - rowsDelivered_;
            if (ahead > buffer_.getCapacity() / 2) {
                // rows that are complete on the device but not handed over yet occupy the ring's free slots: before
                // they and this piece could fill it (a push that would lap them is refused whole), everything in
                // flight is waited for and handed over
                drain(false, true);
                ahead = complete - rowsDelivered_;
            }
            if (ahead > 0) {
                std::lock_guard<std::mutex> g(bufferMutex_);
                buffer_.markAhead((int)std::min<int64_t>(ahead, buffer_.getCapacity()));
            }
        }
        pushRaw(src, n);
        const int rc = ro_stft_push(stft_, src, RO_IQ_F64, (int64_t)n, &ready);
        if (rc != RO_OK) {
            lastError_ = ro_last_error();
            std::fprintf(stderr, "HipWaterfallBackend: %s\n", lastError_.c_str());
            return;
        }
        // (rows that are complete but not yet handed over count against the sink's free slots: if this piece could be
        // refused for it, everything in flight is handed over first -- see the check in front of the push)
        drain(false);
    }
}

void HipWaterfallBackend::endStream()
{
    Backend::endStream();
    if (stft_) drain(true);
    finishStream();
}

// Hand finished rows to the recorders in stream order.  `flush` runs the kernels on every
// complete row still staged (end of stream, or a caller that wants minimum latency).
void HipWaterfallBackend::drain(bool flush, bool wait)
{
    int64_t ready = 0;
    if (flush && ro_stft_flush(stft_, &ready) != RO_OK) {
        // (a flush whose last batch would lap rows still waiting in the sink is refused with its samples kept: hand over
        // everything that is in flight, then flush again -- once is enough, the ring is empty then)
        drain(false, true);
        if (ro_stft_flush(stft_, &ready) != RO_OK) {
            lastError_ = ro_last_error();
            return;
        }
    }
    const int64_t CH = 64;
    if (!rowSink_) fetchRows_.resize((size_t)CH * bins_);
    fetchRecs_.resize((size_t)CH);
    // Between two process() calls only what has FINISHED on the device is handed over: the batch this call has just
    // launched stays in flight under the recorders' work and under the next call's staging (two batches overlap: one
    // downloading, one uploading / transforming), and its rows reach the recorders with the next call.  At the end of
    // the stream (flush) everything is waited for.
    int64_t budget = -1;
    if (!flush && !wait) {
        if (ro_stft_rows_complete(stft_, &budget) != RO_OK) {
            lastError_ = ro_last_error();
            return;
        }
    }
    for (;;) {
        int64_t first = 0, got = 0;
        if (budget == 0) {
            // ... but never more than maxOutstanding_ batches stay launched-and-not-handed-over (three where the rings are
            // long against a batch -- every shipped config: a second of rows against eight snapshots -- two or one where
            // they are not): the recorders look back `advance` rows into the row ring and the raw ring, which have to hold that lag
            // plus the batch being staged (startStream).  What is beyond is waited for.  (In real time the batch of a
            // second ago finished long before this call.)
            int64_t launched = 0;
            if (ro_stft_stats(stft_, nullptr, &launched, nullptr, nullptr) != RO_OK) break;
            if (launched - rowsFetched_ <= (int64_t)maxOutstanding_ * batchRows_) break;
            budget = launched - rowsFetched_ - (int64_t)maxOutstanding_ * batchRows_;
        }
        const int64_t want = budget < 0 ? CH : std::min<int64_t>(CH, budget);
        if (ro_stft_fetch(stft_, want, 0, bins_, rowSink_ ? nullptr : fetchRows_.data(), scanEnabled_ ? fetchRecs_.data() : nullptr,
                          &first, &got) != RO_OK) {
            lastError_ = ro_last_error();
            return;
        }
        if (got == 0) break;
        if (budget > 0) budget -= got;
        rowsFetched_ += got;
        for (int64_t i = 0; i < got; ++i) {
            const int64_t r = first + i;
            DataInfo di;
            di.offset = (SampleCount)r;                                          // info_.offset, :256
            di.timeOffset = rowTimes_.empty() ? WFTime() : rowTimes_.front();
            if (!rowTimes_.empty()) rowTimes_.pop_front();
            // windowRaw_[0].mark after the overlap memmove (src/FFTBackend.cpp:242,251): the mark
            // (pushes so far, mod capacity) of the next row's first sample, or of this row's when
            // there is no overlap to move.
            const int64_t s = (overlap_ > 0 ? (r + 1) : r) * (int64_t)hop_;
            const int rawMark = (int)((s + 1) % rawCapacity_);
            processRow(rowSink_ ? nullptr : &fetchRows_[(size_t)i * bins_], scanEnabled_ ? &fetchRecs_[(size_t)i] : nullptr, di,
                       rawMark);
        }
    }
}

}  // namespace ro
