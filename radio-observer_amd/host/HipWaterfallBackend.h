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
#include <string>
#include <vector>

#include "../../include/ro_stft.h"
#include "Backend.h"
#include "RingBuffer.h"

namespace ro {

struct RawDataHandle {                       // src/FFTBackend.h:43-49
    int    mark = 0;
    WFTime time;
    RawDataHandle() {}
    RawDataHandle(int m, WFTime t) : mark(m), time(t) {}
};

class HipWaterfallBackend;

// src/WaterfallBackend.h:42-103
class Recorder {
public:
    explicit Recorder(HipWaterfallBackend *backend) : backend_(backend) {}
    virtual ~Recorder() {}
    void setBuffer(RingBuffer2D<float> *buffer, std::vector<RawDataHandle> *rawHandles)
    {
        buffer_ = buffer;
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
    HipWaterfallBackend        *backend_;
    RingBuffer2D<float>        *buffer_ = nullptr;
    std::vector<RawDataHandle> *rawHandles_ = nullptr;
};

// WaterfallBackend::make's config keys (src/WaterfallBackend.cpp:620-646)
struct WaterfallConfig {
    int         bins = 32768;
    int         overlap = 0;
    std::string origin = "debug";
    std::string metadata_path = ".";
    int         buffer_chunk_size = 1024 * 1024;
    double      iq_gain = 0.0;
    int         iq_phase_shift = 0;
    // device-side options (no counterpart in the reference)
    int         device = 0;
    int         max_batch_rows = 0;          // rows per kernel launch; small = low latency
};

class HipWaterfallBackend : public Backend {
public:
    explicit HipWaterfallBackend(const WaterfallConfig &cfg);
    ~HipWaterfallBackend() override;

    // ---- FFTBackend's public surface (src/FFTBackend.h:110-200)
    int   getBins() const { return bins_; }
    float getFFTSampleRate() const { return fftSampleRate_; }
    SampleType getGain() const { return cfg_.iq_gain; }
    float binToFrequency(int bin) const { return ro_bin_to_frequency(bins_, streamInfo_.sampleRate, bin); }
    float binToFrequency() const { return binToFrequency(1) - binToFrequency(0); }
    int   frequencyToBin(float f) const { return ro_frequency_to_bin(bins_, streamInfo_.sampleRate, f); }
    double fftSamplesToTime(int samples) const { return (double)samples / (double)fftSampleRate_; }
    int   timeToFFTSamples(double t) const { return ro_time_to_fft_samples(t, fftSampleRate_); }
    int   fftSamplesToRaw(int sampleCount) const                     // src/WaterfallBackend.h:283-287
    {
        return (int)(((double)sampleCount / (double)fftSampleRate_) * (double)streamInfo_.sampleRate);
    }
    std::string getOrigin() const { return cfg_.origin; }

    void addRecorder(Recorder *recorder);                            // src/WaterfallBackend.cpp:563-567

    void startStream(StreamInfo info) override;                      // :573-594 + FFTBackend.cpp:144-189
    void process(const std::vector<Complex> &data, DataInfo info) override;
    void endStream() override;                                       // :600-607

    // ---- what the row being delivered looks like (valid inside Recorder::update())
    const ro_scan_record_t &currentScan() const { return currentScan_; }
    bool  scanEnabled() const { return scanEnabled_; }
    int64_t currentRowIndex() const { return rowsDelivered_ - 1; }   // DataInfo::offset of that row

    // ---- inspection (tests)
    RingBuffer2D<float> &buffer() { return buffer_; }
    const std::vector<RawDataHandle> &rawHandles() const { return rawHandles_; }
    int64_t rowsDelivered() const { return rowsDelivered_; }
    int rawCapacity() const { return rawCapacity_; }
    const std::string &lastError() const { return lastError_; }
    struct RowInfo { uint64_t offset; WFTime time; int rawMark; };
    const std::vector<RowInfo> &rowLog() const { return rowLog_; }
    void keepRowLog(bool on) { keepLog_ = on; }

protected:
    // the reference's hook (src/FFTBackend.h:104) sees complex spectra; this backend hands
    // over the finished magnitude row instead (the spectrum never leaves the GPU).
    virtual void processRow(const float *row, const ro_scan_record_t *scan, DataInfo info, int rawMark);

private:
    void drain(bool flush);
    void stampRowStarts(int64_t takeBegin, int64_t takeEnd, const WFTime &t);

    WaterfallConfig cfg_;
    int   bins_, overlap_, hop_;
    float fftSampleRate_ = 0.f;
    ro_stft_t *stft_ = nullptr;
    bool  scanEnabled_ = false;
    std::string lastError_;

    RingBuffer2D<float>        buffer_;
    std::vector<RawDataHandle> rawHandles_;
    std::vector<Recorder *>    recorders_;

    // framing bookkeeping that stays on the host (timestamps, raw marks: O(1) per row)
    int64_t samplesIn_ = 0;          // samples received so far
    int     inMark_ = 0;             // samples held towards the next row (inMark_ - window_)
    int64_t nextStampRow_ = 0;       // next row whose first-sample time is still unknown
    std::deque<WFTime> rowTimes_;    // time of the first sample of rows not yet delivered
    int     rawCapacity_ = 1;
    int64_t rowsDelivered_ = 0;
    ro_scan_record_t currentScan_{};
    std::vector<float> fetchRows_;
    std::vector<ro_scan_record_t> fetchRecs_;
    std::vector<RowInfo> rowLog_;
    bool keepLog_ = false;
};

}  // namespace ro
