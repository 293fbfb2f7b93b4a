// BolidRecorder.h -- the detector of src/BolidRecorder.{h,cpp}: band setup (start(), :80-116),
// the per-row decision and the three-state machine (update(), :119-273).  The per-row
// noise()/peak()/average() scan (:313-347) is not done here: it arrives as the GPU's
// ro_scan_record_t for the row being delivered.
#pragma once

#include <iosfwd>
#include <string>
#include <vector>

#include "SnapshotRecorder.h"

namespace ro {

// BolidRecorder::make's config keys and defaults (src/BolidRecorder.cpp:360-381)
struct BolidConfig {
    std::string output_dir = ".";
    std::string output_type = "blid";
    bool   compress_output = true;
    bool   write_files = true;              // see SnapshotConfig
    int    snapshot_length = 60;
    float  low_freq = 9000, hi_freq = 12000;
    float  low_detect_freq = 10000, hi_detect_freq = 10900;
    float  low_noise_freq = 9000, hi_noise_freq = 10000;
    double advance_time = 1, jitter_time = 1;
    float  avg_freq_range = 40;
    float  threshold = 2.0f;                 // parsed and never used, like the reference (:379, h:84)
    double noise_metadata_time = 3600;
};

struct BolidEvent {                          // what :223-263 writes to CSV / BolidMessage / stdout
    int64_t row;                             // row (DataInfo::offset) on which the event fired
    int     start, length;                   // nextSnapshot_.start / .length
    float   duration;                        // seconds, :209
    float   noise, peakFreq, magnitude;
    float   fmin, fmax;                      // :241-242
    int     rawLength;                       // fftSamplesToRaw(length), :246
};

struct NoiseSample { float noise, peakFreq, magnitude; };   // NoiseMessage payload, :137-138

class BolidRecorder : public SnapshotRecorder {
public:
    enum State { STATE_INIT, STATE_BOLID, STATE_BOLID_ENDED };

    BolidRecorder(WaterfallBase *backend, const BolidConfig &cfg);

    void start() override;
    void update() override;
    bool scanBands(ro_bands_t *b) const override;

    const std::vector<BolidEvent> &events() const { return events_; }
    const NoiseSample &lastNoise() const { return lastNoise_; }
    State state() const { return state_; }
    void setOutput(std::ostream *os) { out_ = os; }          // the "met;...#" line (:250-257)

    int lowDetectBin() const { return lowDetectBin_; }
    int detectWidth() const { return detectWidth_; }
    int lowNoiseBin() const { return lowNoiseBin_; }
    int noiseWidth() const { return noiseWidth_; }
    int advance() const { return advance_; }
    int jitter() const { return jitter_; }
    int averageBinRange() const { return averageBinRange_; }

private:
    BolidConfig cfg_;
    float minDetectFq_, maxDetectFq_;
    int   lowDetectBin_ = 0, detectWidth_ = 0, lowNoiseBin_ = 0, noiseWidth_ = 0;
    int   advance_ = 0, jitter_ = 0, averageBinRange_ = 0, noiseMetadataRows_ = 0;
    State state_ = STATE_INIT;
    float peakFreq_ = 0, noise_ = 0, magnitude_ = 0;
    int   duration_ = 0;
    std::vector<BolidEvent> events_;
    NoiseSample lastNoise_{};
    std::ostream *out_ = nullptr;
};

}  // namespace ro
