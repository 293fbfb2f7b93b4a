// FITSWriter.h -- minimal writer for the one kind of file the reference produces: a primary HDU holding a
// 2-D float32 image (src/FITSWriter.cpp:40-163 via cfitsio; cfitsio itself is not a dependency here).
// FITS 4.0: 80-character cards in 2880-byte blocks, big-endian IEEE data, zero padding.
// The reference's default "[compress]" tile compression (src/WaterfallBackend.cpp:171-172) is lossy for float
// images and is not reproduced: files are written uncompressed (= compress_output: false).
#pragma once

#include <cstdint>
#include <cstdio>
#include <string>
#include <vector>

namespace ro {

class FITSWriter {
public:
    bool open(const std::string &fileName);              // a leading '!' (cfitsio "overwrite") is accepted and dropped
    void createImage(long width, long height);           // BITPIX = -32  (FLOAT_IMG)
    void writeHeader(const char *key, const char *value, const char *comment);
    void writeHeader(const char *key, int value, const char *comment);
    void writeHeader(const char *key, long long value, const char *comment);
    void writeHeader(const char *key, float value, const char *comment);
    void writeHeader(const char *key, double value, const char *comment);
    void comment(const char *text);
    void date();                                          // DATE = file creation time, UTC (fits_write_date)
    void write(long y, long count, const float *data);   // `count` rows starting at row y
    bool close();
    bool ok() const { return ok_; }

private:
    void card(const std::string &text);
    void endHeader();

    FILE *file_ = nullptr;
    std::vector<std::string> cards_;
    long width_ = 0, height_ = 0;
    bool headerWritten_ = false, ok_ = true;
    long dataStart_ = 0;
};

}  // namespace ro
