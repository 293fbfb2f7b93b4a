// CsvLog.h -- the hourly-rotated metadata CSV of src/CsvLog.{h,cpp}: one file per wall-clock hour,
// "# <header>" as the first line of a new file, one entry per line, flushed after every write.
#pragma once

#include <fstream>
#include <string>

#include "Backend.h"

namespace ro {

class CsvLog {
public:
    CsvLog(const std::string &fileNameFormat, const std::string &header)      // src/CsvLog.cpp:33-37
        : format_(fileNameFormat), header_(header) {}

    std::string getFileName(WFTime time) const;                               // :40-43
    void write(WFTime time, const std::string &entry);                        // :46-53
    const std::string &currentFile() const { return name_; }

private:
    std::string   format_, header_, name_;
    std::ofstream out_;
};

}  // namespace ro
