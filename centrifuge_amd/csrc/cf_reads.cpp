// cf_reads.cpp — see cf_reads.hpp
#include "cf_reads.hpp"
#include "cf_bytesource.hpp"

#include <cctype>
#include <cstring>
#include <stdexcept>

namespace cfamd {

namespace {
// alphabet.cpp:298-319 (asc2dna: everything but ACGTN becomes A)
inline uint8_t dnaCode(int c) {
    switch (c) {
        case 'C': case 'c': return 1;
        case 'G': case 'g': return 2;
        case 'T': case 't': return 3;
        case 'N': case 'n': return 4;
        default: return 0;
    }
}
// drop the last n elements (SStringExpandable::trimEnd): everything when there are no more than n
void trimEnd(std::vector<uint8_t> &v, int n) {
    if (n <= 0) return;
    const size_t k = static_cast<size_t>(n);
    if (v.size() > k) v.resize(v.size() - k); else v.clear();
}
}  // namespace

ReadSource::ReadSource(std::vector<std::string> files, ReadFormat fmt, int trim5, int trim3)
    : files_(std::move(files)), fmt_(fmt), trim5_(trim5), trim3_(trim3), buf_(1 << 22) {}

ReadSource::~ReadSource() = default;

bool ReadSource::openNext() {
    f_.reset();
    if (fmt_ == ReadFormat::CmdLine || fileIdx_ >= files_.size()) return false;
    f_.reset(new ByteSource(files_[fileIdx_++]));           // throws when the file cannot be opened
    pos_ = len_ = 0;
    return true;
}

int ReadSource::peek() {
    if (pos_ >= len_) {
        if (!f_) return -1;
        len_ = f_->read(reinterpret_cast<char *>(buf_.data()), buf_.size());
        pos_ = 0;
        if (len_ == 0) return -1;
    }
    return buf_[pos_];
}
int ReadSource::get() {
    const int c = peek();
    if (c >= 0) pos_++;
    return c;
}

bool ReadSource::next(ReadRec &r) {
    r.name.clear(); r.seq.clear(); r.qual.clear();
    if (fmt_ == ReadFormat::Fasta || fmt_ == ReadFormat::Fastq)
        throw std::runtime_error("FASTA / FASTQ input goes through the chunk parsers of cf_ingest.cpp");
    if (fmt_ == ReadFormat::CmdLine) {
        if (fileIdx_ >= files_.size()) return false;
        // -c: the "file names" are the reads, `seq` or `seq:qual` (VectorPatternSource, pat.cpp:456-546): trimmed by
        // characters, then EVERY remaining character becomes a base through asc2dna (anything but ACGTN is an A);
        // missing qualities are 'I', surplus ones are cut
        const std::string &tok = files_[fileIdx_++];
        const size_t colon = tok.find(':');
        std::string sq = tok.substr(0, colon), vq = colon == std::string::npos ? std::string() : tok.substr(colon + 1);
        const size_t cut = (size_t)std::max(0, trim3_) + (size_t)std::max(0, trim5_);
        if (sq.size() <= cut) sq.clear();
        else { if (trim5_ > 0) sq.erase(0, (size_t)trim5_); if (trim3_ > 0) sq.erase(sq.size() - (size_t)trim3_); }
        if (vq.size() > cut) { if (trim5_ > 0) vq.erase(0, (size_t)trim5_); if (trim3_ > 0) vq.erase(vq.size() - (size_t)trim3_); }
        vq.resize(sq.size(), 'I');
        for (char ch : sq) r.seq.push_back(dnaCode((unsigned char)ch));
        if (colon != std::string::npos) r.qual.assign(vq.begin(), vq.end());
        r.name = std::to_string(readCnt_);
        readCnt_++;
        return true;
    }
    for (;;) {
        if (!f_ && !openNext()) return false;
        bool ok = nextRaw(r);
        if (ok) return true;
        f_.reset();
    }
}

// RawPatternSource::read (pat.h:1493-1584): one sequence per line — the first whitespace-free token of the
// line; every character of it counts toward the 5' trim, only letters become bases ('.' is not a base here:
// that translation belongs to colour space); the rest of the line is skipped.
bool ReadSource::nextRaw(ReadRec &r) {
    int c = get();
    while (c == '\n' || c == '\r') c = get();
    if (c < 0) return false;
    if (rawFirst_) {                                         // sanity check of the file's first character (pat.h:1511-1529)
        static const char *kDna = "ABCDGHKMNRSTVWXYabcdghkmnrstvwxy-";          // asc2dnacat > 0 (alphabet.cpp:36-58)
        if (!std::strchr(kDna, c)) {
            std::string m = "Error: reads file does not look like a Raw file";
            if (c == '>') m += "\nReads file looks like a FASTA file; please use -f";
            if (c == '@') m += "\nReads file looks like a FASTQ file; please use -q";
            throw std::runtime_error(m);
        }
        rawFirst_ = false;
    }
    int chs = 0;
    while (c >= 0 && !std::isspace(c)) {
        if (std::isalpha(c) && chs >= trim5_) r.seq.push_back(dnaCode(c));
        chs++;
        c = get();
    }
    while (c >= 0 && c != '\n' && c != '\r') c = get();      // peekToEndOfLine
    if (trim3_ > 0) { trimEnd(r.seq, trim3_); }
    r.name = std::to_string(readCnt_);
    readCnt_++;
    return true;
}

}  // namespace cfamd
