// Frontends.h -- sample sources that drive a Backend the way the reference's do:
//   WAVStream  (src/WAVStream.cpp:101-245): RIFF/WAVE, 16-bit PCM, 2 channels = I,Q, values handed over
//              un-normalised as doubles, 1024 frames per Backend::process() call (:112-123, :190)
//   RawStream  (src/RawStream.cpp:30-69): interleaved little-endian float32 I,Q, up to 4096 frames per call
// Both sit on std::istream so that tests can feed them from memory.  Deviations from the reference's
// undefined or buggy corners are deliberate and listed at the code.
#pragma once

#include <istream>
#include <string>
#include <vector>

#include "Backend.h"

namespace ro {

struct WAVFormat {                       // src/WAVStream.h:33-49
    int audioFormat = 0, channelCount = 0, sampleRate = 0, byteRate = 0, blockAlign = 0, bitsPerSample = 0;
};

class WAVStream : public Frontend {
public:
    explicit WAVStream(std::istream &in, Backend *backend = nullptr) : in_(in) { setBackend(backend); }
    // src/WAVStream.cpp:198-245.  Where the reference logs an error and returns, lastError() holds the text.
    void run() override;
    bool ok() const { return error_.empty(); }
    const WAVFormat &format() const { return format_; }
    const std::string &inf1() const { return inf1_; }
    const std::string &lastError() const { return error_; }
    int64_t framesDelivered() const { return frames_; }

private:
    bool readDataSubchunk(int64_t size);
    template <class T> T readScalar()
    {
        T v = T();
        in_.read(reinterpret_cast<char *>(&v), sizeof(T));
        return v;
    }
    std::string readString(int length);

    std::istream  &in_;
    WAVFormat      format_;
    std::string    inf1_, error_;
    bool           dataRead_ = false;
    int64_t        frames_ = 0;
    static const int kBlockFrames = 1024;            // dataBufferSize_, src/WAVStream.cpp:190
};

class RawStream : public Frontend {
public:
    RawStream(std::istream &in, Backend *backend, int sampleRate, WFTime start = WFTime())
        : in_(in), sampleRate_(sampleRate), start_(start) { setBackend(backend); }
    void run() override;                              // src/RawStream.cpp:30-69
    int64_t framesDelivered() const { return frames_; }

private:
    std::istream  &in_;
    int            sampleRate_;
    WFTime         start_;
    int64_t        frames_ = 0;
    static const int kBlockFrames = 4096;             // bufferSize, src/RawStream.cpp:32
};

}  // namespace ro
