// Backend.h -- host-side mirror of the reference's sample-stream boundary.
//
// Same names, argument meaning and call order as the reference's
//   struct Complex / StreamInfo / DataInfo / class Backend   (src/Backend.h:26-102)
//   struct WFTime                                            (src/WFTime.h:35-114)
// so that a frontend written against the reference drives this backend unchanged.
// cppapp (DIObject, Ref<>) is not part of this tree: ownership is plain C++ here and the
// DI factory keys are listed in INTEGRATION.md.
#pragma once

#include <cstdint>
#include <vector>

namespace ro {

typedef double   SampleType;    // src/common_types.h:15-17
typedef uint64_t SampleCount;
typedef int      SampleRate;

// seconds + microseconds, with the reference's truncating sample arithmetic
struct WFTime {
    int64_t sec = 0, usec = 0;
    WFTime() {}
    WFTime(int64_t s, int64_t us) : sec(s), usec(us) {}

    // src/WFTime.h:92-103
    WFTime add(int64_t seconds, int64_t microseconds) const
    {
        const int64_t M = 1000000;
        int64_t us = usec + microseconds % M;
        int64_t s = sec + seconds + microseconds / M + us / M;
        return WFTime(s, us % M);
    }
    // src/WFTime.h:105-114: whole seconds by integer division, the rest through double, truncated
    WFTime addSamples(SampleCount count, SampleRate rate) const
    {
        const SampleCount whole = count / (SampleCount)rate, rem = count % (SampleCount)rate;
        const long micro = (long)(((double)rem / (double)rate) * 1000000.0);
        return add((int64_t)whole, (int64_t)micro);
    }
    double toMilliseconds() const { return (double)sec * 1000.0 + (double)usec / 1000.0; }
    bool operator==(const WFTime &o) const { return sec == o.sec && usec == o.usec; }
};

struct Complex { double real, imag; };                      // src/Backend.h:26-29

struct StreamInfo {                                          // src/Backend.h:35-59
    bool   knownLength = false;
    int    length = 0;
    int    sampleRate = 48000;
    WFTime timeOffset;
    double samplesToTime(int samples) const { return (double)samples / (double)sampleRate; }
};

struct DataInfo {                                            // src/Backend.h:65-77
    SampleCount offset = 0;
    WFTime      timeOffset;
};

class Backend {                                              // src/Backend.h:83-102
public:
    virtual ~Backend() {}
    StreamInfo getStreamInfo() const { return streamInfo_; }
    virtual void startStream(StreamInfo info) { streamInfo_ = info; }
    virtual void process(const std::vector<Complex> &data, DataInfo info) = 0;
    virtual void endStream() {}

protected:
    StreamInfo streamInfo_;
};

// What a Frontend does around the backend (src/Frontend.cpp:16-52): keeps the running
// DataInfo and recomputes timeOffset from the stream start after every call.
class FrontendDriver {
public:
    explicit FrontendDriver(Backend *b) : backend_(b) {}
    void startStream(const StreamInfo &info)
    {
        streamInfo_ = info;
        backend_->startStream(info);
        dataInfo_.offset = 0;
        dataInfo_.timeOffset = info.timeOffset;
    }
    void process(const std::vector<Complex> &data)
    {
        backend_->process(data, dataInfo_);
        dataInfo_.offset += data.size();
        dataInfo_.timeOffset = streamInfo_.timeOffset.addSamples(dataInfo_.offset, streamInfo_.sampleRate);
    }
    void endStream() { backend_->endStream(); }

private:
    Backend   *backend_;
    StreamInfo streamInfo_;
    DataInfo   dataInfo_;
};

// src/Frontend.h:24-60, src/Frontend.cpp:16-60: a sample source.  run() reads its input and feeds the backend
// through startStream() / process() / endStream(); stop() asks a running frontend to finish.
class Frontend {
public:
    Frontend() {}
    virtual ~Frontend() {}
    void setBackend(Backend *backend) { backend_ = backend; }
    Backend *getBackend() const { return backend_; }
    virtual void run() = 0;
    virtual void stop() { stopping_ = true; }                       // src/Frontend.cpp:55-58

protected:
    void startStream()                                              // src/Frontend.cpp:16-27
    {
        if (!backend_) return;
        backend_->startStream(streamInfo_);
        dataInfo_.offset = 0;
        dataInfo_.timeOffset = streamInfo_.timeOffset;
    }
    void endStream() { if (backend_) backend_->endStream(); }       // :30-38
    void process(const std::vector<Complex> &data)                  // :41-52
    {
        if (!backend_) return;
        backend_->process(data, dataInfo_);
        dataInfo_.offset += data.size();
        dataInfo_.timeOffset = streamInfo_.timeOffset.addSamples(dataInfo_.offset, streamInfo_.sampleRate);
    }

    Backend   *backend_ = nullptr;
    StreamInfo streamInfo_;
    DataInfo   dataInfo_;
    bool       stopping_ = false;
};

// src/Pipeline.h:24-58, src/Pipeline.cpp:11-36: binds a frontend to a backend and runs it.  (The reference's
// agents -- MetadataAgent and friends -- have empty bodies and are not mirrored; ownership is the caller's.)
class Pipeline {
public:
    Frontend *getFrontend() const { return frontend_; }
    void setFrontend(Frontend *frontend) { frontend_ = frontend; }
    Backend *getBackend() const { return backend_; }
    void setBackend(Backend *backend) { backend_ = backend; }
    void run()                                                      // src/Pipeline.cpp:11-19
    {
        if (!frontend_) return;
        frontend_->setBackend(backend_);
        frontend_->run();
    }
    void stop() { if (frontend_) frontend_->stop(); }               // :22-36

private:
    Frontend *frontend_ = nullptr;
    Backend  *backend_ = nullptr;
};

}  // namespace ro
