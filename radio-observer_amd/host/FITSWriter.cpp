#include "FITSWriter.h"

#include <cstring>
#include <ctime>

namespace ro {

static std::string pad80(std::string s)
{
    if (s.size() > 80) s.resize(80);
    s.append(80 - s.size(), ' ');
    return s;
}

static std::string keyValue(const char *key, const std::string &value, const char *comment)
{
    char buf[128];
    std::snprintf(buf, sizeof(buf), "%-8.8s= %20s", key, value.c_str());
    std::string s(buf);
    if (comment && *comment) s += std::string(" / ") + comment;
    return pad80(s);
}

bool FITSWriter::open(const std::string &fileName)
{
    std::string name = fileName;
    if (!name.empty() && name[0] == '!') name.erase(0, 1);
    const size_t br = name.find('[');                       // cfitsio extended syntax, e.g. "[compress]"
    if (br != std::string::npos) name.resize(br);
    file_ = std::fopen(name.c_str(), "wb");
    cards_.clear();
    headerWritten_ = false;
    ok_ = file_ != nullptr;
    return ok_;
}

void FITSWriter::createImage(long width, long height)
{
    width_ = width;
    height_ = height;
    card(keyValue("SIMPLE", "T", "file does conform to FITS standard"));
    card(keyValue("BITPIX", "-32", "number of bits per data pixel"));
    card(keyValue("NAXIS", "2", "number of data axes"));
    card(keyValue("NAXIS1", std::to_string(width), "length of data axis 1"));
    card(keyValue("NAXIS2", std::to_string(height), "length of data axis 2"));
    card(keyValue("EXTEND", "T", "FITS dataset may contain extensions"));
}

void FITSWriter::card(const std::string &text) { cards_.push_back(pad80(text)); }

void FITSWriter::writeHeader(const char *key, const char *value, const char *comment)
{
    std::string v = value;
    std::string q;
    for (char c : v) { q += c; if (c == '\'') q += '\''; }
    if (q.size() < 8) q.append(8 - q.size(), ' ');
    char buf[128];
    std::snprintf(buf, sizeof(buf), "%-8.8s= '%s'", key, q.c_str());
    std::string s(buf);
    if (s.size() < 30) s.append(30 - s.size(), ' ');
    if (comment && *comment) s += std::string(" / ") + comment;
    card(s);
}
void FITSWriter::writeHeader(const char *key, int value, const char *comment) { card(keyValue(key, std::to_string(value), comment)); }
void FITSWriter::writeHeader(const char *key, long long value, const char *comment) { card(keyValue(key, std::to_string(value), comment)); }
void FITSWriter::writeHeader(const char *key, float value, const char *comment)
{
    char b[40];
    std::snprintf(b, sizeof(b), "%.7G", (double)value);         // cfitsio's TFLOAT precision
    std::string v = b;
    if (v.find_first_of(".EN") == std::string::npos) v += ".";  // keep it a FITS real
    card(keyValue(key, v, comment));
}
void FITSWriter::writeHeader(const char *key, double value, const char *comment)
{
    char b[40];
    std::snprintf(b, sizeof(b), "%.15G", value);                // cfitsio's TDOUBLE precision
    std::string v = b;
    if (v.find_first_of(".EN") == std::string::npos) v += ".";
    card(keyValue(key, v, comment));
}
void FITSWriter::comment(const char *text)
{
    std::string t = text;
    for (size_t i = 0; i < t.size() || i == 0; i += 72) {
        card("COMMENT " + t.substr(i, 72));
        if (t.empty()) break;
    }
}
void FITSWriter::date()
{
    std::time_t now = std::time(nullptr);
    char b[32];
    std::strftime(b, sizeof(b), "%Y-%m-%dT%H:%M:%S", std::gmtime(&now));
    writeHeader("DATE", b, "file creation date (YYYY-MM-DDThh:mm:ss UT)");
}

void FITSWriter::endHeader()
{
    if (headerWritten_ || !file_) return;
    std::string block;
    for (const auto &c : cards_) block += c;
    block += pad80("END");
    block.append((2880 - block.size() % 2880) % 2880, ' ');
    ok_ = ok_ && std::fwrite(block.data(), 1, block.size(), file_) == block.size();
    dataStart_ = (long)block.size();
    headerWritten_ = true;
}

void FITSWriter::write(long y, long count, const float *data)
{
    endHeader();
    if (!file_ || y < 0 || y + count > height_) { ok_ = false; return; }
    std::vector<unsigned char> be((size_t)count * width_ * 4);
    for (long i = 0; i < count * width_; ++i) {
        uint32_t u;
        std::memcpy(&u, &data[i], 4);
        be[(size_t)i * 4 + 0] = (unsigned char)(u >> 24);
        be[(size_t)i * 4 + 1] = (unsigned char)(u >> 16);
        be[(size_t)i * 4 + 2] = (unsigned char)(u >> 8);
        be[(size_t)i * 4 + 3] = (unsigned char)u;
    }
    std::fseek(file_, dataStart_ + y * width_ * 4, SEEK_SET);
    ok_ = ok_ && std::fwrite(be.data(), 1, be.size(), file_) == be.size();
}

bool FITSWriter::close()
{
    if (!file_) return false;
    endHeader();
    const long dataBytes = width_ * height_ * 4;
    const long total = dataStart_ + dataBytes;
    const long padTo = ((total + 2879) / 2880) * 2880;
    std::fseek(file_, 0, SEEK_END);
    long have = std::ftell(file_);
    if (have < padTo) {
        std::vector<unsigned char> z((size_t)(padTo - have), 0);
        ok_ = ok_ && std::fwrite(z.data(), 1, z.size(), file_) == z.size();
    }
    ok_ = (std::fclose(file_) == 0) && ok_;
    file_ = nullptr;
    return ok_;
}

}  // namespace ro
