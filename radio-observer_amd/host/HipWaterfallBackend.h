// HipWaterfallBackend.h -- drop-in for the reference's WaterfallBackend (and the FFTBackend
// under it) on an MI355X: same Backend / Recorder surface, the arithmetic runs in the HIP
// kernels behind include/ro_stft.h.
//
//   reference                                        here
//   FFTBackend::process       (src/FFTBackend.cpp:192-279)  -> ro_stft_push + row bookkeeping
//   fftw_execute + window     (:229-236)                    -> stft_kernel (GPU)
//   WaterfallBackend::processFFT (src/WaterfallBackend.cpp:485-541) -> deliverRow()
//   Recorder::update() per row (:534-536)                   -> same call, same order, batched later
#pragma once

#include <deque>
#include <memory>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/ro_stft.h"
#include "Backend.h"
#include "CsvLog.h"
#include "RingBuffer.h"

namespace ro {

struct RawDataHandle {                       // src/FFTBackend.h:43-49
    int    mark = 0;
    WFTime time;
    RawDataHandle() {}
    RawDataHandle(int m, WFTime t) : mark(m), time(t) {}
};

class WaterfallBase;

// src/WaterfallBackend.h:42-103
class Recorder {
public:
    explicit Recorder(WaterfallBase *backend) : backend_(backend) {}
    virtual ~Recorder() {}
    void setBuffer(RingBuffer2D<float> *buffer, RingBuffer2D<float> *rawBuffer, std::mutex *bufferMutex,
                   std::vector<RawDataHandle> *rawHandles)
    {
        buffer_ = buffer;
        rawBuffer_ = rawBuffer;
        bufferMutex_ = bufferMutex;
        rawHandles_ = rawHandles;
    }
    int getSampleRate() const;
    int getFFTSampleRate() const;            // int, like the reference (src/WaterfallBackend.cpp:29-32)
    int fftMarkToRaw(int mark) const;        // :35-39
    WFTime fftMarkToTime(int mark) const;    // :42-46
    int fftSamplesToRaw(int sampleCount) const
    {
        return (int)(((double)sampleCount / (double)getFFTSampleRate()) * (double)getSampleRate());
    }
    virtual int  requestBufferSize() { return 0; }
    virtual void start() {}
    virtual void stop() {}
    virtual void update() = 0;
    // the scan bands a recorder wants computed on the GPU for every row (none by default)
    virtual bool scanBands(ro_bands_t *) const { return false; }

protected:
    WaterfallBase              *backend_;
    RingBuffer2D<float>        *buffer_ = nullptr;
    RingBuffer2D<float>        *rawBuffer_ = nullptr;     // raw I/Q, 2 floats per sample (src/FFTBackend.h:99)
    std::mutex                 *bufferMutex_ = nullptr;   // guards the row ring's bookkeeping (WaterfallBackend.h:249)
    std::vector<RawDataHandle> *rawHandles_ = nullptr;
};

// WaterfallBackend::make's config keys (src/WaterfallBackend.cpp:620-646)
struct WaterfallConfig {
    int         bins = 32768;
    int         overlap = 0;
    std::string origin = "debug";
    std::string metadata_path = ".";         // "" = keep no metadata CSV (test rigs)
    int         buffer_chunk_size = 1024 * 1024;
    double      iq_gain = 0.0;
    int         iq_phase_shift = 0;
    // device-side options (no counterpart in the reference)
    int         device = 0;
    int         max_batch_rows = 0;          // rows per kernel launch; small = low latency
    bool        keep_raw = true;             // keep the raw I/Q ring (the reference always does: FFTBackend.cpp:217-223)
};

// Everything WaterfallBackend is to its recorders (src/WaterfallBackend.h:239-290 and the FFTBackend
// accessors they call, src/FFTBackend.h:110-200): the row ring, the raw handles, the bin/Hz/time
// helpers, and the per-row delivery of src/WaterfallBackend.cpp:485-541 minus the arithmetic.  The HIP
// backend feeds it from the GPU; ManualWaterfall lets host-only tests feed rows by hand.
class WaterfallBase {
public:
    explicit WaterfallBase(const WaterfallConfig &cfg);
    virtual ~WaterfallBase() {}

    // ---- FFTBackend's public surface (src/FFTBackend.h:110-200)
    int   getBins() const { return bins_; }
    float getFFTSampleRate() const { return fftSampleRate_; }
    SampleType getGain() const { return cfg_.iq_gain; }
    StreamInfo streamInfo() const { return info_; }
    float binToFrequency(int bin) const { return ro_bin_to_frequency(bins_, info_.sampleRate, bin); }
    float binToFrequency() const { return binToFrequency(1) - binToFrequency(0); }
    int   frequencyToBin(float f) const { return ro_frequency_to_bin(bins_, info_.sampleRate, f); }
    double fftSamplesToTime(int samples) const { return (double)samples / (double)fftSampleRate_; }
    int   timeToFFTSamples(double t) const { return ro_time_to_fft_samples(t, fftSampleRate_); }
    int   fftSamplesToRaw(int sampleCount) const                     // src/WaterfallBackend.h:283-287
    {
        return (int)(((double)sampleCount / (double)fftSampleRate_) * (double)info_.sampleRate);
    }
    std::string getOrigin() const { return cfg_.origin; }
    const WaterfallConfig &config() const { return cfg_; }

    void addRecorder(Recorder *recorder);                            // src/WaterfallBackend.cpp:563-567

    // <metadata_path>/%Y%m%d%H%M%S_<origin>_meta.csv, rotated hourly (src/WaterfallBackend.cpp:466-482);
    // nullptr when metadata_path is empty
    CsvLog *getMetadataFile();
    // NoiseMessage (src/BolidMessage.h:19-49): the detector publishes (noise, peak f, magnitude) every row
    // (src/BolidRecorder.cpp:137-138); listening snapshot recorders keep the latest (WaterfallBackend.cpp:270-277)
    struct Noise { float noise = 0, peakFrequency = 0, magnitude = 0; };
    void publishNoise(float n, float peakFq, float mag) { lastNoise_ = Noise{n, peakFq, mag}; }
    const Noise &lastNoise() const { return lastNoise_; }
    // WFTime::now() (src/WFTime.h:173-178); tests pin it with setClock
    WFTime now() const;
    void setClock(WFTime fixed) { fixedClock_ = fixed; useFixedClock_ = true; }

    // ---- what the row being delivered looks like (valid inside Recorder::update())
    const ro_scan_record_t &currentScan() const { return currentScan_; }
    bool  scanEnabled() const { return scanEnabled_; }
    int64_t currentRowIndex() const { return rowsDelivered_ - 1; }   // DataInfo::offset of that row

    // ---- inspection (tests)
    RingBuffer2D<float> &buffer() { return buffer_; }
    RingBuffer2D<float> &rawBuffer() { return rawBuffer_; }
    const std::vector<RawDataHandle> &rawHandles() const { return rawHandles_; }
    int64_t rowsDelivered() const { return rowsDelivered_; }
    int rawCapacity() const { return rawCapacity_; }
    struct RowInfo { uint64_t offset; WFTime time; int rawMark; };
    const std::vector<RowInfo> &rowLog() const { return rowLog_; }
    void keepRowLog(bool on) { keepLog_ = on; }

protected:
    // src/WaterfallBackend.cpp:573-594 (ring sizing, recorders' start()); returns the scan bands a
    // recorder asked for through *bands (true if any)
    bool beginStream(const StreamInfo &info, ro_bands_t *bands);
    void finishStream();                                             // :600-607
    // the reference's hook (src/FFTBackend.h:104) sees complex spectra; here the finished magnitude row
    // is handed over instead (the spectrum never leaves the GPU)
    virtual void processRow(const float *row, const ro_scan_record_t *scan, DataInfo info, int rawMark);

    WaterfallConfig cfg_;
    StreamInfo info_;
    int   bins_, overlap_, hop_;
    float fftSampleRate_ = 0.f;
    bool  scanEnabled_ = false;

    // raw samples as (float)re, (float)im, one ring row per sample (src/FFTBackend.cpp:217-223); `spans` (optional)
    // receives where they went -- at most two runs of consecutive ring rows -- and the call returns how many
    void pushRaw(const Complex *data, size_t n);

    std::unique_ptr<CsvLog>    metadataFile_;
    Noise  lastNoise_;
    WFTime fixedClock_;
    bool   useFixedClock_ = false;
    RingBuffer2D<float>        buffer_;
    std::mutex                 bufferMutex_;     // WaterfallBackend::bufferMutex_: recorders' workers share the ring
    RingBuffer2D<float>        rawBuffer_;
    std::vector<RawDataHandle> rawHandles_;
    std::vector<Recorder *>    recorders_;
    int     rawCapacity_ = 1;
    int     maxOutstanding_ = 1;      // batches launched and not yet handed to the recorders (drain)
    int64_t rowsDelivered_ = 0;
    int64_t rowsFetched_ = 0;        // rows taken out of the handle's queue (== rowsDelivered_ between calls)
    ro_scan_record_t currentScan_{};
    std::vector<RowInfo> rowLog_;
    bool keepLog_ = false;
};

// host-only source: rows (and scan records) are pushed by the caller.  Used by the CPU tests of the
// recorders; never part of a data path.
class ManualWaterfall : public WaterfallBase {
public:
    explicit ManualWaterfall(const WaterfallConfig &cfg) : WaterfallBase(cfg) {}
    void startStream(const StreamInfo &info) { ro_bands_t b; scanEnabled_ = beginStream(info, &b); }
    void pushSamples(const Complex *data, size_t n) { pushRaw(data, n); }
    void pushRow(const float *row, const ro_scan_record_t *scan, WFTime time, int rawMark)
    {
        DataInfo di;
        di.offset = (SampleCount)rowsDelivered_;
        di.timeOffset = time;
        processRow(row, scan, di, rawMark);
    }
    void endStream() { finishStream(); }
};

class HipWaterfallBackend : public Backend, public WaterfallBase {
public:
    explicit HipWaterfallBackend(const WaterfallConfig &cfg);
    ~HipWaterfallBackend() override;

    void startStream(StreamInfo info) override;                      // :573-594 + FFTBackend.cpp:144-189
    void process(const std::vector<Complex> &data, DataInfo info) override;
    void endStream() override;                                       // :600-607
    const std::string &lastError() const { return lastError_; }
    int batchRows() const { return batchRows_; }                     // rows per kernel launch of this stream
    bool rowsByDma() const { return rowSink_; }                      // the rows land in the ring's slots (ro_stft_set_row_sink)
    // FFTBackend::logProcessingTimes' numbers (src/FFTBackend.h:208-229) for this stream: ro_stft_timing of its handle
    bool timing(ro_stft_timing_t *out, bool reset) const { return stft_ && ro_stft_timing(stft_, out, reset ? 1 : 0) == RO_OK; }

private:
    void drain(bool flush, bool wait = false);
    void stampRowStarts(int64_t takeBegin, int64_t takeEnd, const WFTime &t);

    ro_stft_t *stft_ = nullptr;
    int batchRows_ = 0;
    bool rowSink_ = false;           // rows arrive in the ring's slots by DMA (ro_stft_set_row_sink)
    std::string lastError_;

    // framing bookkeeping that stays on the host (timestamps, raw marks: O(1) per row)
    int64_t samplesIn_ = 0;          // samples received so far
    int     inMark_ = 0;             // samples held towards the next row (inMark_ - window_)
    int64_t nextStampRow_ = 0;       // next row whose first-sample time is still unknown
    std::deque<WFTime> rowTimes_;    // time of the first sample of rows not yet delivered
    std::vector<float> fetchRows_;
    std::vector<ro_scan_record_t> fetchRecs_;
};

}  // namespace ro
