#include "BolidRecorder.h"

#include <sstream>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <ostream>

namespace ro {

static SnapshotConfig snapshotPart(const BolidConfig &c)
{
    SnapshotConfig s;
    s.output_dir = c.output_dir;
    s.output_type = c.output_type;
    s.compress_output = c.compress_output;
    s.write_files = c.write_files;
    s.listen_to_noise = false;                                               // src/BolidRecorder.h:145
    s.snapshot_length = c.snapshot_length;
    s.low_freq = c.low_freq;
    s.hi_freq = c.hi_freq;
    return s;
}

BolidRecorder::BolidRecorder(WaterfallBase *backend, const BolidConfig &cfg)
    : SnapshotRecorder(backend, snapshotPart(cfg)), cfg_(cfg)
{
    writeUnfinished_ = false;                                                // src/BolidRecorder.h:160
    minDetectFq_ = std::min(cfg.low_detect_freq, cfg.hi_detect_freq);        // ORDER_PAIR, BolidRecorder.h:161
    maxDetectFq_ = std::max(cfg.low_detect_freq, cfg.hi_detect_freq);
}

void BolidRecorder::start()
{
    int lo = backend_->frequencyToBin(minDetectFq_), hi = backend_->frequencyToBin(maxDetectFq_);   // :84-88
    if (lo > hi) std::swap(lo, hi);
    lowDetectBin_ = lo;
    detectWidth_ = hi - lo;
    lo = backend_->frequencyToBin(cfg_.low_noise_freq);                                             // :90-95
    hi = backend_->frequencyToBin(cfg_.hi_noise_freq);
    lowNoiseBin_ = std::min(lo, hi);
    noiseWidth_ = std::max(lo, hi) - lowNoiseBin_;
    advance_ = backend_->timeToFFTSamples(cfg_.advance_time);                                       // :100-103
    jitter_ = backend_->timeToFFTSamples(cfg_.jitter_time);
    averageBinRange_ = backend_->frequencyToBin(cfg_.avg_freq_range) - backend_->frequencyToBin(0);
    noiseMetadataRows_ = backend_->timeToFFTSamples(cfg_.noise_metadata_time);
    state_ = STATE_INIT;                                                                            // :106-108
    events_.clear();
    SnapshotRecorder::start();                                                                      // :115
}

bool BolidRecorder::scanBands(ro_bands_t *b) const
{
    // the reference asserts averageBinRange_ > 0 (:104); without it there is nothing to scan
    if (noiseWidth_ <= 0 || detectWidth_ <= 0 || averageBinRange_ <= 0) return false;
    b->low_noise = lowNoiseBin_;
    b->noise_width = noiseWidth_;
    b->low_detect = lowDetectBin_;
    b->detect_width = detectWidth_;
    b->avg_bins = averageBinRange_;
    return true;
}

void BolidRecorder::update()
{
    if (!backend_->scanEnabled()) return;
    const ro_scan_record_t &s = backend_->currentScan();             // n, p, a of :124-132
    const float n = s.noise, a = s.average;
    const int   p = s.peak;
    const float peakFq = backend_->binToFrequency(lowDetectBin_ + p);   // :133
    const bool  detect = ((double)a > (double)n * 2.0);                 // :135
    lastNoise_ = NoiseSample{n, peakFq, a};                             // NoiseMessage, :137-138
    backend_->publishNoise(n, peakFq, a);

    switch (state_) {
    case STATE_INIT:                                                    // :172-183
        if (detect) {
            peakFreq_ = peakFq;
            noise_ = n;
            magnitude_ = a;
            duration_ = 1;
            nextSnapshot_.start = buffer_->mark() - advance_;                    // :178-180
            nextSnapshot_.length = 2 * advance_;
            nextSnapshot_.fileName = getFileName(fftMarkToTime(nextSnapshot_.start));
            state_ = STATE_BOLID;
        }
        break;
    case STATE_BOLID:                                                   // :185-193
        if (detect) {
            duration_ += 1;
        } else {
            nextSnapshot_.length += duration_;
            duration_ = 1;
            state_ = STATE_BOLID_ENDED;
        }
        break;
    case STATE_BOLID_ENDED:                                             // :195-267
        duration_ += 1;
        if (detect) {
            state_ = STATE_BOLID;
        } else if (duration_ >= jitter_) {
            BolidEvent ev;
            ev.row = backend_->currentRowIndex();
            ev.start = nextSnapshot_.start;
            ev.length = nextSnapshot_.length;
            ev.duration = (float)(nextSnapshot_.length - 2 * advance_) / (float)backend_->getFFTSampleRate();   // :209
            ev.noise = noise_;
            ev.peakFreq = peakFreq_;
            ev.magnitude = magnitude_;
            ev.fmin = peakFreq_ - (maxDetectFq_ - minDetectFq_) / 4;                                    // :241
            ev.fmax = peakFreq_ + (maxDetectFq_ - minDetectFq_) / 4;
            ev.rawLength = fftSamplesToRaw(nextSnapshot_.length);                                       // :246
            events_.push_back(ev);
            const WFTime t = backend_->now();                                                           // :221
            if (CsvLog *log = backend_->getMetadataFile()) {                                            // :223-234
                std::ostringstream entry;
                entry << baseName(nextSnapshot_.fileName) << ";" << noise_ << ";" << peakFreq_ << ";" << magnitude_
                      << ";" << ev.duration;
                log->write(t, entry.str());
            }
            if (out_) {                                                                                 // :250-257
                (*out_) << "met;[" << t.sec << "s, " << t.usec << "us];" << ev.noise << ";" << ev.peakFreq << ";"
                        << ev.magnitude << ";" << ev.fmin << ";" << ev.fmax << ";" << ev.duration << ";"
                        << ev.rawLength << "#" << std::endl;
            }
            nextSnapshot_.includeRawData = true;                                                        // :262-263
            startWriting();
            state_ = STATE_INIT;
        }
        break;
    }
}

}  // namespace ro
