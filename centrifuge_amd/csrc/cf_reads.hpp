// cf_reads.hpp — the sequential read sources of the command-line front end: raw (one sequence per line) and
// command-line sequences into base codes 0..4, names and qualities.  Behaviour follows the reference's parsers
// (RawPatternSource pat.h:1478-1585, VectorPatternSource pat.cpp:456-546; alphabet.cpp:298-319); the code is our
// own.  FASTA / FASTQ files go through the chunk parsers of cf_ingest.cpp.
#pragma once
#include <cstdint>
#include <cstdio>
#include <memory>
#include <string>
#include <vector>

namespace cfamd {

class ByteSource;

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
    bool nextRaw(ReadRec &r);

    std::vector<std::string> files_;
    ReadFormat fmt_;
    int trim5_, trim3_;
    size_t fileIdx_ = 0;
    std::unique_ptr<ByteSource> f_;      // the open file (plain / stdin / gzip / bzip2, cf_bytesource.hpp)
    std::vector<unsigned char> buf_;
    size_t pos_ = 0, len_ = 0;
    uint64_t readCnt_ = 0;
    bool rawFirst_ = true;                 // raw format: the first character of the input is still to be checked
};

}  // namespace cfamd
