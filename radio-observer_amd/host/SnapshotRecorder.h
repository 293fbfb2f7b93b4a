// SnapshotRecorder.h -- the fixed-cadence FITS recorder of src/WaterfallBackend.{h,cpp}:107-458.
// Same cadence and file contents; the file is written from update()/stop() on the caller's thread
// instead of the reference's worker thread (src/WaterfallBackend.cpp:60-104) -- rows are already
// complete in the ring when a snapshot is queued here, so there is nothing to wait for.
#pragma once

#include <string>
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

class SnapshotRecorder : public Recorder {
public:
    SnapshotRecorder(WaterfallBase *backend, const SnapshotConfig &cfg);

    int  requestBufferSize() override;   // :339-347
    void start() override;               // :364-397
    void stop() override;                // :400-412
    void update() override;              // :415-427

    std::string getFileName(WFTime time) const;                      // :320-336
    const std::vector<std::string> &filesWritten() const { return written_; }
    const std::vector<std::string> &rawFilesWritten() const { return writtenRaw_; }
    const std::vector<Snapshot> &snapshotsQueued() const { return queued_; }
    int snapshotRows() const { return snapshotRows_; }
    int leftBin() const { return leftBin_; }
    int rightBin() const { return rightBin_; }

protected:
    void startWriting();                 // :107-127
    bool write(const Snapshot &s);       // :141-211
    bool writeRaw(const Snapshot &s);    // :214-267
    std::string getFileName(const char *typ, WFTime time) const;
    void drainPending(bool final);

    SnapshotConfig cfg_;
    float leftFrequency_, rightFrequency_;
    bool  writeUnfinished_ = true;
    int   snapshotRows_ = 1, leftBin_ = 0, rightBin_ = 0;
    Snapshot nextSnapshot_;
    std::vector<Snapshot> pending_, queued_;
    std::vector<std::string> written_, writtenRaw_;
};

}  // namespace ro
