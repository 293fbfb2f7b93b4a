#include "BolidRecorder.h"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <ostream>

namespace ro {

BolidRecorder::BolidRecorder(WaterfallBase *backend, const BolidConfig &cfg) : Recorder(backend), cfg_(cfg)
{
    minDetectFq_ = std::min(cfg.low_detect_freq, cfg.hi_detect_freq);        // ORDER_PAIR, BolidRecorder.h:161
    maxDetectFq_ = std::max(cfg.low_detect_freq, cfg.hi_detect_freq);
}

int BolidRecorder::requestBufferSize()
{
    const float rate = backend_->getFFTSampleRate();
    snapshotRows_ = (int)std::ceil(cfg_.snapshot_length * rate);
    if (snapshotRows_ < 1) snapshotRows_ = 1;
    return snapshotRows_ * 8;
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

    switch (state_) {
    case STATE_INIT:                                                    // :172-183
        if (detect) {
            peakFreq_ = peakFq;
            noise_ = n;
            magnitude_ = a;
            duration_ = 1;
            snapStart_ = buffer_->mark() - advance_;
            snapLength_ = 2 * advance_;
            state_ = STATE_BOLID;
        }
        break;
    case STATE_BOLID:                                                   // :185-193
        if (detect) {
            duration_ += 1;
        } else {
            snapLength_ += duration_;
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
            ev.start = snapStart_;
            ev.length = snapLength_;
            ev.duration = (float)(snapLength_ - 2 * advance_) / (float)backend_->getFFTSampleRate();   // :209
            ev.noise = noise_;
            ev.peakFreq = peakFreq_;
            ev.magnitude = magnitude_;
            ev.fmin = peakFreq_ - (maxDetectFq_ - minDetectFq_) / 4;                                    // :241
            ev.fmax = peakFreq_ + (maxDetectFq_ - minDetectFq_) / 4;
            ev.rawLength = fftSamplesToRaw(snapLength_);                                                // :246
            events_.push_back(ev);
            if (out_) {                                                                                 // :250-257
                (*out_) << "met;" << ev.row << ";" << ev.noise << ";" << ev.peakFreq << ";" << ev.magnitude << ";"
                        << ev.fmin << ";" << ev.fmax << ";" << ev.duration << ";" << ev.rawLength << "#" << std::endl;
            }
            state_ = STATE_INIT;
        }
        break;
    }
}

}  // namespace ro
