#include "CsvLog.h"

#include <ctime>

namespace ro {

// WFTime::getHour(true).format(fmt) (src/WFTime.h:146-159, src/WFTime.cpp:19-34): minutes and seconds
// cleared in LOCAL time, the result formatted as UTC -- the two agree when TZ is UTC, like the stations'
std::string CsvLog::getFileName(WFTime time) const
{
    std::time_t stamp = (std::time_t)time.sec;
    std::tm parts;
    localtime_r(&stamp, &parts);
    parts.tm_min = 0;
    parts.tm_sec = 0;
    stamp = std::mktime(&parts);
    gmtime_r(&stamp, &parts);
    char buf[1024];
    const size_t n = std::strftime(buf, sizeof(buf), format_.c_str(), &parts);
    return std::string(buf, n);
}

void CsvLog::write(WFTime time, const std::string &entry)
{
    const std::string name = getFileName(time);
    if (!out_.is_open() || name != name_) {                                   // src/CsvLog.cpp:12-24
        bool exists = false;
        {
            std::ifstream probe(name.c_str());
            exists = probe.good();
        }
        if (out_.is_open()) out_.close();
        out_.open(name.c_str(), std::ios_base::out | std::ios_base::binary | std::ios_base::app);
        name_ = name;
        if (!exists) out_ << "# " << header_ << std::endl;
        out_.flush();
    }
    out_ << entry << std::endl;
    out_.flush();
}

}  // namespace ro
