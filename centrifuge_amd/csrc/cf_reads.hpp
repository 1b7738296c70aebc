// cf_reads.hpp — read ingest of the command-line front end: FASTA / FASTQ / raw /
// command-line sequences into base codes 0..4, names, qualities and per-read seeds.
// Behaviour follows the reference's parsers (pat.cpp:725-850 FASTA, :852-1100 FASTQ,
// :1290+ raw; alphabet.cpp:36-58,298-319); the code is our own.
#pragma once
#include <cstdint>
#include <cstdio>
#include <memory>
#include <string>
#include <vector>

namespace cfamd {

enum class ReadFormat { Fasta, Fastq, Raw, CmdLine };

struct ReadRec {
    std::string name;
    std::vector<uint8_t> seq;     // 0..3 = ACGT, 4 = N
    std::vector<uint8_t> qual;    // empty = all 'I' (FASTA, pat.cpp:828)
};

// One input stream (a list of files read one after another, or the -c sequences).
class ReadSource {
public:
    ReadSource(std::vector<std::string> files, ReadFormat fmt, int trim5, int trim3);
    ~ReadSource();
    bool next(ReadRec &r);        // false at the end of the last file; throws std::runtime_error
    uint64_t count() const { return readCnt_; }

private:
    int get();
    int peek();
    bool openNext();
    bool nextFasta(ReadRec &r);
    bool nextFastq(ReadRec &r);
    bool nextRaw(ReadRec &r);

    std::vector<std::string> files_;
    ReadFormat fmt_;
    int trim5_, trim3_;
    size_t fileIdx_ = 0;
    std::FILE *f_ = nullptr;
    bool pipe_ = false;
    std::vector<unsigned char> buf_;
    size_t pos_ = 0, len_ = 0;
    bool first_ = true;
    uint64_t readCnt_ = 0;
    bool rawFirst_ = true;                 // raw format: the first character of the input is still to be checked
};

}  // namespace cfamd
