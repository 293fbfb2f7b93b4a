#include "Frontends.h"

#include <cstdint>
#include <cstring>

namespace ro {

std::string WAVStream::readString(int length)
{
    std::string s((size_t)length, '\0');
    in_.read(&s[0], length);
    const size_t z = s.find('\0');
    if (z != std::string::npos) s.resize(z);
    return s;
}

// src/WAVStream.cpp:198-245 (run) and :158-183 (readSubchunk)
void WAVStream::run()
{
    streamInfo_ = StreamInfo();
    if (readString(4) != "RIFF") { error_ = "Invalid chunk ID. Stream may not be in WAV format."; return; }
    int64_t chunkSize = (int64_t)readScalar<uint32_t>();
    if (readString(4) != "WAVE") { error_ = "Invalid chunk format. Stream may not be in WAV format."; return; }
    chunkSize -= 4;
    dataRead_ = false;
    while (chunkSize > 0 && in_.good()) {
        const std::string id = readString(4);
        const int64_t size = (int64_t)readScalar<uint32_t>();
        if (!in_.good()) break;
        if (id == "fmt ") {                                               // :62-85
            format_.audioFormat = readScalar<int16_t>();
            format_.channelCount = readScalar<int16_t>();
            format_.sampleRate = readScalar<int32_t>();
            format_.byteRate = readScalar<int32_t>();
            format_.blockAlign = readScalar<int16_t>();
            format_.bitsPerSample = readScalar<int16_t>();
            streamInfo_.sampleRate = format_.sampleRate;
            // the reference reads 16 bytes whatever `size` says; an extended fmt chunk would desynchronise
            // its parser -- the extension is skipped here
            if (size > 16) in_.ignore(size - 16);
        } else if (id == "inf1") {                                        // :91-97
            // untrusted length: keep at most 64 KiB of it (a size above INT_MAX once meant a huge std::string)
            const uint32_t keep = size < 65536u ? size : 65536u;
            inf1_ = readString((int)keep);
            if (size > keep) in_.ignore(size - keep);
        } else if (id == "data") {                                        // :169-176
            if (!dataRead_) {
                startStream();
                dataRead_ = true;
            }
            if (!readDataSubchunk(size)) break;
        } else {
            in_.ignore(size);                                             // :146-150
        }
        // the reference subtracts size + 4 (:182), four bytes short of the 8-byte sub-chunk header
        // (Appendix B-1); the correct accounting is used here
        chunkSize -= size + 8;
    }
    if (dataRead_) endStream();                                           // :243-244
}

// src/WAVStream.cpp:101-143
bool WAVStream::readDataSubchunk(int64_t size)
{
    if (format_.bitsPerSample != 16) {                                    // :103-106
        error_ = "Can only read 16 bits per sample! Stopping now.";
        return false;
    }
    if (format_.channelCount != 2 || format_.blockAlign != 4) {
        // the reference indexes dataBuffer_[2*s], [2*s+1] whatever the channel count is (:119-120): a mono
        // file makes it read past its buffer (Appendix B-1).  Only 2-channel I/Q files are accepted here.
        error_ = "Only 2-channel (I/Q) 16-bit PCM is supported.";
        return false;
    }
    std::vector<int16_t> raw((size_t)kBlockFrames * 2);
    std::vector<Complex> out;
    int64_t remaining = size;
    while (remaining >= format_.blockAlign && in_.good() && !stopping_) {
        const int64_t want = std::min<int64_t>(remaining, (int64_t)kBlockFrames * format_.blockAlign);
        in_.read(reinterpret_cast<char *>(raw.data()), want);
        const int64_t got = in_.gcount();
        const int frames = (int)(got / format_.blockAlign);
        if (frames <= 0) break;
        // the reference's tail path (:126-138) doubles the sample count and re-reads stale data; the tail is
        // delivered once, with its true length
        out.resize((size_t)frames);
        for (int s = 0; s < frames; ++s) {
            out[(size_t)s].real = (double)raw[(size_t)s * 2];             // :119  un-normalised int16
            out[(size_t)s].imag = (double)raw[(size_t)s * 2 + 1];         // :120
        }
        process(out);
        frames_ += frames;
        remaining -= got;
        if (got < want) break;
    }
    if (remaining > 0 && remaining < format_.blockAlign) in_.ignore(remaining);
    return true;
}

// src/RawStream.cpp:30-69
void RawStream::run()
{
    std::vector<float> raw((size_t)kBlockFrames * 2);
    std::vector<Complex> out;
    streamInfo_ = StreamInfo();
    streamInfo_.sampleRate = sampleRate_;
    streamInfo_.timeOffset = start_;          // the reference stamps WFTime::now() (:40); injectable here
    startStream();
    while (!stopping_) {                                                  // :44
        in_.read(reinterpret_cast<char *>(raw.data()), (std::streamsize)(raw.size() * sizeof(float)));
        const int64_t frames = in_.gcount() / (int64_t)(sizeof(float) * 2);   // :58
        if (frames <= 0) break;
        out.resize((size_t)frames);
        for (int64_t i = 0; i < frames; ++i) {
            out[(size_t)i].real = raw[(size_t)i * 2];                     // :61
            out[(size_t)i].imag = raw[(size_t)i * 2 + 1];                 // :62
        }
        process(out);
        frames_ += frames;
        if (!in_.good()) break;
    }
    endStream();
}

}  // namespace ro
