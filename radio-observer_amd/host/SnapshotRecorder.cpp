#include "SnapshotRecorder.h"

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <ctime>
#include <sstream>

#include "FITSWriter.h"

namespace ro {

static std::string formatTime(const WFTime &t, const char *fmt)       // src/WFTime.cpp:19-34 (UTC)
{
    std::time_t s = (std::time_t)t.sec;
    char buf[256];
    const size_t n = std::strftime(buf, sizeof(buf), fmt, std::gmtime(&s));
    return std::string(buf, n);
}

std::string baseName(const std::string &path)                        // cppapp Path::basename
{
    const size_t at = path.find_last_of('/');
    return at == std::string::npos ? path : path.substr(at + 1);
}

static std::string joinPath(const std::string &a, const std::string &b)
{
    if (a.empty()) return b;
    if (a.back() == '/') return a + b;
    return a + "/" + b;
}

SnapshotRecorder::SnapshotRecorder(WaterfallBase *backend, const SnapshotConfig &cfg) : Recorder(backend), cfg_(cfg)
{
    leftFrequency_ = std::min(cfg.low_freq, cfg.hi_freq);            // ORDER_PAIR, WaterfallBackend.h:205
    rightFrequency_ = std::max(cfg.low_freq, cfg.hi_freq);
}

SnapshotRecorder::~SnapshotRecorder() { joinWorker(); }

void SnapshotRecorder::joinWorker()
{
    if (worker_.joinable()) {
        snapshots_.close();
        worker_.join();
    }
}

int SnapshotRecorder::requestBufferSize()
{
    const float rate = backend_->getFFTSampleRate();
    snapshotRows_ = (int)std::ceil(cfg_.snapshot_length * rate);
    if (snapshotRows_ < 1) snapshotRows_ = 1;
    return snapshotRows_ * 8;
}

void SnapshotRecorder::start()
{
    if (leftFrequency_ == rightFrequency_) {                          // :368-374
        const StreamInfo info = backend_->streamInfo();
        leftFrequency_ = -(float)info.sampleRate / 2.0f;
        rightFrequency_ = (float)info.sampleRate / 2.0f;
        leftBin_ = 0;
        rightBin_ = backend_->getBins();
    } else {
        leftBin_ = backend_->frequencyToBin(leftFrequency_);
        rightBin_ = backend_->frequencyToBin(rightFrequency_);
    }
    nextSnapshot_ = Snapshot();
    nextSnapshot_.fileName = getFileName(fftMarkToTime(nextSnapshot_.start));   // :394-395 (epoch-zero name on
    joinWorker();                                                                //  the first file: App. B-4)
    queued_.clear();
    {
        std::lock_guard<std::mutex> g(listMutex_);
        written_.clear();
        writtenRaw_.clear();
    }
    snapshots_.reopen();
    worker_ = std::thread(&SnapshotRecorder::threadMethod, this);               // :396
}

std::string SnapshotRecorder::getFileName(const char *typ, WFTime time) const
{
    char name[1024];
    std::snprintf(name, sizeof(name), "%s%03d_%s_%s.%s", formatTime(time, "%Y%m%d%H%M%S").c_str(),
                  (int)(time.usec / 1000), backend_->getOrigin().c_str(), typ, "fits");
    return joinPath(cfg_.output_dir, name);                           // :332-335 (the directory is joined twice
}                                                                     //  in the reference's write(); once here)

std::string SnapshotRecorder::getFileName(WFTime time) const { return getFileName(cfg_.output_type.c_str(), time); }

void SnapshotRecorder::startWriting()
{
    {
        std::lock_guard<std::mutex> g(*bufferMutex_);                                               // :109-119
        if (nextSnapshot_.length == 0) nextSnapshot_.length = buffer_->size(nextSnapshot_.start);  // :111-112
        if (snapshotRows_ < nextSnapshot_.length) nextSnapshot_.length = snapshotRows_;            // :113-114
        nextSnapshot_.reservation = buffer_->reserve(nextSnapshot_.start, nextSnapshot_.start + nextSnapshot_.length);   // :117
    }
    const int end = nextSnapshot_.start + nextSnapshot_.length;
    queued_.push_back(nextSnapshot_);
    snapshots_.send(nextSnapshot_);                                                                 // :120
    nextSnapshot_ = Snapshot();
    nextSnapshot_.start = end;                                                                      // :122
    nextSnapshot_.fileName = getFileName(fftMarkToTime(nextSnapshot_.start));                       // :125
}

// The worker (src/WaterfallBackend.cpp:60-104): takes what update() queued, writes every snapshot whose rows are
// complete, keeps the others for the next turn.  It reads ring rows without the lock, like the reference: the rows
// of a queued snapshot are reserved and the ring is eight snapshots long; only the ring's bookkeeping (size,
// reservations) is touched under the buffer mutex.
void SnapshotRecorder::threadMethod()
{
    std::vector<Snapshot> work;
    for (;;) {
        const bool open = snapshots_.drain(work, /*block=*/true, work.empty() ? 50 : 2);
        std::vector<Snapshot> keep;
        for (const Snapshot &s : work) {
            bool complete;
            {
                std::lock_guard<std::mutex> g(*bufferMutex_);
                complete = buffer_->size(s.start) >= s.length;                                      // :76
            }
            if (complete) {
                if (cfg_.write_files) {
                    write(s);
                    if (s.includeRawData) writeRaw(s);                                              // :80-81
                }
                std::lock_guard<std::mutex> g(*bufferMutex_);
                // The reference keeps the reservation's dirty flag and never reads it (src/RingBuffer.h:482-495, :617-620).
                // Here a writer that has been lapped says so: with the row sink the ring's slots are written by DMA up to
                // a batch ahead of push(), so a snapshot that fell a ring behind may hold rows newer than its header.
                if (buffer_->isDirty(s.reservation)) {
                    dirtySnapshots_++;
                    std::fprintf(stderr, "SnapshotRecorder: %s was lapped by the row ring while it waited to be written; "
                                         "its oldest rows are newer than its header says\n", s.fileName.c_str());
                }
                buffer_->freeReservation(s.reservation);                                            // :87-90
            } else if (open) {
                keep.push_back(s);                     // rows still to come; at the end an unfinished one is dropped
            }
        }
        work.swap(keep);
        if (!open) break;
    }
}

void SnapshotRecorder::update()
{
    bool due;
    {
        std::lock_guard<std::mutex> g(*bufferMutex_);
        due = buffer_->size(nextSnapshot_.start) >= snapshotRows_ + 2;                             // :417-426
    }
    if (due) startWriting();
}

void SnapshotRecorder::stop()
{
    bool any;
    {
        std::lock_guard<std::mutex> g(*bufferMutex_);
        any = buffer_->size(nextSnapshot_.start) >= 0;                                             // :402 (always true)
    }
    if (any && writeUnfinished_) startWriting();                                                   // :402-403
    joinWorker();                                                                                  // :404-411
}

bool SnapshotRecorder::write(const Snapshot &s)
{
    const WFTime time = fftMarkToTime(s.start);                       // :143
    const float fftSampleRate = backend_->getFFTSampleRate();
    if (cfg_.listen_to_noise) {                                       // :157-167
        if (CsvLog *log = backend_->getMetadataFile()) {
            const WaterfallBase::Noise &nz = backend_->lastNoise();
            std::ostringstream entry;
            entry << baseName(s.fileName) << ";" << nz.noise << ";" << nz.peakFrequency << ";" << nz.magnitude << ";" << 0;
            log->write(time, entry.str());
        }
    }
    FITSWriter w;
    if (!w.open("!" + s.fileName)) return false;
    const int width = rightBin_ - leftBin_;
    w.createImage(width, s.length);
    w.comment("File created by radio-observer_amd (MI355X STFT path).");       // :130-135 writeHeader()
    w.comment("See https://github.com/MLAB-project/radio-observer.");
    w.writeHeader("ORIGIN", backend_->getOrigin().c_str(), "");
    w.date();
    w.writeHeader("DATE-OBS", formatTime(time, "%Y-%m-%dT%H:%M:%S").c_str(), "observation date (UTC)");
    w.writeHeader("CTYPE2", "TIME", "in seconds");
    w.writeHeader("CRPIX2", 1, "");
    w.writeHeader("CRVAL2", (long long)time.toMilliseconds(), "unix time of the first FFT row in this file in ms");
    w.writeHeader("CDELT2", 1000.0 / (double)fftSampleRate, "time difference between two FFT samples in ms");
    w.writeHeader("CTYPE1", "FREQ", "in Hz");
    w.writeHeader("CRPIX1", 1.f, "");
    w.writeHeader("CRVAL1", (float)leftFrequency_, "frequency, in Hz, of the leftmost pixel in the image");
    w.writeHeader("CDELT1", (float)backend_->binToFrequency(), "frequency difference between two neighbouring pixels in Hz");
    int rowIndex = s.start;
    for (int y = 0; y < s.length; ++y, ++rowIndex) w.write(y, 1, buffer_->at(rowIndex) + leftBin_);   // :203-205
    const bool ok = w.close();
    if (ok) {
        std::lock_guard<std::mutex> g(listMutex_);
        written_.push_back(s.fileName);
    }
    return ok;
}

// the raw I/Q behind a snapshot: a 2 x L float image (src/WaterfallBackend.cpp:214-267)
bool SnapshotRecorder::writeRaw(const Snapshot &s)
{
    if (!rawBuffer_ || rawBuffer_->getCapacity() == 0) return false;
    const int start = fftMarkToRaw(s.start);                          // :216
    const int length = fftSamplesToRaw(s.length);                     // :217 (the recorder's int fft rate)
    const WFTime time = fftMarkToTime(s.start);
    const float sampleRate = (float)backend_->streamInfo().sampleRate;
    const std::string name = getFileName("raws", time);
    FITSWriter w;
    if (!w.open("!" + name)) return false;
    w.createImage(2, length);
    w.comment("File created by radio-observer_amd (MI355X STFT path).");
    w.comment("See https://github.com/MLAB-project/radio-observer.");
    w.writeHeader("ORIGIN", backend_->getOrigin().c_str(), "");
    w.date();
    w.writeHeader("DATE-OBS", formatTime(time, "%Y-%m-%dT%H:%M:%S").c_str(), "observation date (UTC)");
    w.writeHeader("CTYPE2", "TIME", "in seconds");
    w.writeHeader("CRPIX2", 1, "");
    w.writeHeader("CRVAL2", (long long)time.toMilliseconds(), "unix time of the first IQ sample in this file in ms");
    w.writeHeader("CDELT2", 1000.0f / sampleRate, "time difference between two IQ samples in ms");
    w.writeHeader("CTYPE1", "CHAN", "in Hz");
    w.writeHeader("CRPIX1", 1.f, "");
    w.writeHeader("CRVAL1", 0, "");
    w.writeHeader("CDELT1", 1, "");
    int rowIndex = start;
    for (int y = 0; y < length; ++y, ++rowIndex) w.write(y, 1, rawBuffer_->at(rowIndex));            // :258-260
    const bool ok = w.close();
    if (ok) {
        std::lock_guard<std::mutex> g(listMutex_);
        writtenRaw_.push_back(name);
    }
    return ok;
}

}  // namespace ro
