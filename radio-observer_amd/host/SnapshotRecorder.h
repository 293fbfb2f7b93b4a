// SnapshotRecorder.h -- the fixed-cadence FITS recorder of src/WaterfallBackend.{h,cpp}:107-458.
// Same cadence and file contents, and the same division of labour: update() (the caller's thread, inside
// Backend::process) only decides and queues; a worker thread per recorder, fed through a Channel, waits for a
// queued snapshot's rows to be complete and writes the file (src/WaterfallBackend.cpp:60-104, :396; src/Channel.h),
// so a slow disk never stalls Backend::process.
#pragma once

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <deque>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "HipWaterfallBackend.h"

namespace ro {

// SnapshotRecorder::make's config keys and defaults (src/WaterfallBackend.cpp:436-458)
struct SnapshotConfig {
    std::string output_dir = ".";
    std::string output_type = "snap";
    bool  compress_output = true;       // accepted; files are always written uncompressed (FITSWriter.h)
    int   snapshot_length = 60;
    float low_freq = 0, hi_freq = 0;    // equal = full band (:368-374)
    bool  write_files = true;           // false = keep the cadence/bookkeeping, touch no files (test rigs)
    bool  listen_to_noise = true;       // "snapshot" factory: true (:448-457); the detector passes false
};

std::string baseName(const std::string &path);

struct Snapshot {                        // src/WaterfallBackend.h:117-150
    int start = 0, length = 0, reservation = -1;
    bool includeRawData = false;
    std::string fileName;
    int end() const { return start + length; }
};

// the mutex + condition-variable queue between update() and the worker (src/Channel.h:26-108)
template <class T> class Channel {
public:
    void send(const T &item)
    {
        { std::lock_guard<std::mutex> g(m_); q_.push_back(item); }
        cv_.notify_one();
    }
    void close()
    {
        { std::lock_guard<std::mutex> g(m_); closing_ = true; }
        cv_.notify_all();
    }
    void reopen() { std::lock_guard<std::mutex> g(m_); closing_ = false; q_.clear(); }
    // Moves everything queued into `out`; when `block`, waits up to wait_ms for something to arrive.  Returns whether
    // the channel is still open.  (send() and close() come from one thread, so a drain that sees the channel closed
    // has also seen everything that was ever sent.)
    bool drain(std::vector<T> &out, bool block, int wait_ms)
    {
        std::unique_lock<std::mutex> g(m_);
        if (block && q_.empty() && !closing_) cv_.wait_for(g, std::chrono::milliseconds(wait_ms));
        for (const T &x : q_) out.push_back(x);
        q_.clear();
        return !closing_;
    }

private:
    std::mutex m_;
    std::condition_variable cv_;
    std::deque<T> q_;
    bool closing_ = false;
};

class SnapshotRecorder : public Recorder {
public:
    SnapshotRecorder(WaterfallBase *backend, const SnapshotConfig &cfg);
    ~SnapshotRecorder() override;

    int  requestBufferSize() override;   // :339-347
    void start() override;               // :364-397
    void stop() override;                // :400-412
    void update() override;              // :415-427

    std::string getFileName(WFTime time) const;                      // :320-336
    // (copies taken under the lock: the worker appends to the lists while the stream runs)
    std::vector<std::string> filesWritten() const { std::lock_guard<std::mutex> g(listMutex_); return written_; }
    std::vector<std::string> rawFilesWritten() const { std::lock_guard<std::mutex> g(listMutex_); return writtenRaw_; }
    int dirtySnapshots() const { return dirtySnapshots_.load(); }     // snapshots written after the ring had lapped them
    const std::vector<Snapshot> &snapshotsQueued() const { return queued_; }
    int snapshotRows() const { return snapshotRows_; }
    int leftBin() const { return leftBin_; }
    int rightBin() const { return rightBin_; }

protected:
    void startWriting();                 // :107-127
    bool write(const Snapshot &s);       // :141-211
    bool writeRaw(const Snapshot &s);    // :214-267
    std::string getFileName(const char *typ, WFTime time) const;
    void threadMethod();                 // :60-104
    void joinWorker();

    SnapshotConfig cfg_;
    float leftFrequency_, rightFrequency_;
    bool  writeUnfinished_ = true;
    int   snapshotRows_ = 1, leftBin_ = 0, rightBin_ = 0;
    Snapshot nextSnapshot_;
    std::vector<Snapshot> queued_;       // everything startWriting() has queued so far (inspection)
    std::vector<std::string> written_, writtenRaw_;
    std::atomic<int> dirtySnapshots_{0};
    mutable std::mutex listMutex_;
    Channel<Snapshot> snapshots_;
    std::thread worker_;
};

}  // namespace ro
