// cf_ingest.hpp — multi-threaded read ingest of the front end.
//
// The reference parses one read per mutex acquisition (pat.h:786-845).  Here an I/O
// thread cuts each input stream into ~16 MiB chunks at record boundaries, a pool of
// parser threads turns chunks into structure-of-arrays batches (base codes, offsets,
// names, qualities, per-read seeds), and next() hands the chunks back in file order.
// Record semantics are those of cf_reads.cpp (FastaPatternSource / FastqPatternSource,
// pat.cpp:725-1100); FASTQ records must be four lines each on this path.
#pragma once
#include <cstring>
#include <condition_variable>
#include <cstdint>
#include <cstdio>
#include <deque>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "cf_reads.hpp"

namespace cfamd {

// A buffer in pinned host memory (cf_host_alloc) when a HIP device is there — what makes the slot's uploads asynchronous DMA —
// else plain memory.  Keeps its capacity while the ReadSoA that owns it goes round between the parser threads and the batches.
template <typename T>
struct HostBuf {
    T *p = nullptr;
    size_t n = 0, cap = 0;
    bool pinned = false;
    HostBuf() = default;
    HostBuf(const HostBuf &) = delete;
    HostBuf &operator=(const HostBuf &) = delete;
    HostBuf(HostBuf &&o) noexcept { *this = std::move(o); }
    HostBuf &operator=(HostBuf &&o) noexcept {
        if (this != &o) { release(); p = o.p; n = o.n; cap = o.cap; pinned = o.pinned; o.p = nullptr; o.n = o.cap = 0; }
        return *this;
    }
    ~HostBuf() { release(); }
    void release();
    void reserve(size_t count);          // contents are NOT kept
    void grow(size_t count, size_t keep);   // the first `keep` elements are
};

// The reads of a ReadSoA in the packed form the batch slot takes (cf_packed_reads of centrifuge_amd.h): 2-bit words, 32 bases
// per word, every read starting on a word; the N mask in its sparse form; lengths and seeds.  Made by the parser thread that
// parsed the chunk (ReadSoA::pack), so the GPU thread uploads 3/8 byte per base straight from pinned memory.
struct PackedSoA {
    HostBuf<uint64_t> words, nIdx;
    HostBuf<uint32_t> lens, seeds, nMsk;
    uint64_t nReads = 0, nWords = 0, nBases = 0, nN = 0;
    uint32_t maxLen = 0;
    bool valid = false;                  // false: the byte form (seq / off) is all there is (batches assembled record by record)
    bool appendable = true;              // an empty batch, or one that so far took whole mate pairs from packed chunks (ReadSoA::appendInterleaved)
    std::vector<uint64_t> woff;          // pack(): word offset of every read (nReads + 1) — what interleaving mates needs
};

// reads in structure-of-arrays form: read i = seq[off[i], off[i+1]), names[nameOff[i], nameOff[i+1])
struct ReadSoA {
    PackedSoA pk;
    void pack();                            // pk from seq / off / seeds
    std::vector<uint8_t> seq;
    std::vector<uint64_t> off{0};
    std::string names;
    std::vector<uint64_t> nameOff{0};
    std::vector<uint8_t> qual;              // parallel to seq when hasQual
    std::vector<uint32_t> seeds;
    std::vector<uint32_t> unnamedKeep;      // reads without a name that must stay unnamed (FASTQ records without a base letter)
    bool hasQual = false;

    size_t size() const { return off.size() - 1; }
    void clear() {
        seq.clear(); off.assign(1, 0); names.clear(); nameOff.assign(1, 0); qual.clear(); seeds.clear(); unnamedKeep.clear();
        pk.valid = false; pk.appendable = true; pk.nReads = pk.nWords = pk.nBases = pk.nN = 0; pk.maxLen = 0;
    }
    void push(const uint8_t *s, const uint8_t *q, size_t len, const char *name, size_t nameLen, uint32_t seed);
    // bulk append of records [i0, i1) of another batch
    void appendRange(const ReadSoA &o, size_t i0, size_t i1);
    // bulk append of cnt mate pairs, interleaved: a[ia], b[ib], a[ia+1], b[ib+1], ...
    void appendInterleaved(const ReadSoA &a, size_t ia, const ReadSoA &b, size_t ib, size_t cnt);
    bool hasEmptyName() const { for (size_t i = 0; i + 1 < nameOff.size(); i++) if (nameOff[i + 1] == nameOff[i]) return true; return false; }
    void appendRecord(const ReadSoA &o, size_t i) {
        const size_t len = o.off[i + 1] - o.off[i];
        push(o.seq.data() + o.off[i], o.hasQual ? o.qual.data() + o.off[i] : nullptr, len, o.names.data() + o.nameOff[i],
             o.nameOff[i + 1] - o.nameOff[i], o.seeds[i]);
    }
};

class ChunkedReader {
public:
    // pack: the parser threads also make every chunk's packed form (ReadSoA::pk)
    // startOffset: where in the (first, plain) file to begin — a record start (a run that changes over from the device text path)
    ChunkedReader(std::vector<std::string> files, ReadFormat fmt, int trim5, int trim3, uint32_t globalSeed, int threads, bool pack = false, uint64_t startOffset = 0);
    ~ChunkedReader();
    // Next chunk of parsed reads in input order; false at the end.  Reads whose name was empty
    // come back with an empty name (the caller substitutes the read's ordinal, pat.cpp:838-842).
    bool next(ReadSoA &out);

private:
    // a block of file bytes: plain memory that is never value-initialised (a std::vector<char> zero-fills 32 MiB before
    // every read) and keeps its capacity while it goes round between the I/O thread and the parsers
    struct CharBuf {
        std::unique_ptr<char[]> p;
        size_t len = 0, cap = 0;
        void ensure(size_t n) {
            if (n <= cap) return;
            std::unique_ptr<char[]> q(new char[n + 64]);
            if (len) std::memcpy(q.get(), p.get(), len);
            p = std::move(q); cap = n;
        }
    };
    // a block is either bytes (data) or a range of a regular file the parser reads itself (fd >= 0: foff, flen)
    struct Raw { uint64_t seq; CharBuf data; bool first; bool last = false; int fd = -1; uint64_t foff = 0, flen = 0; };
    void ioLoop();
    void parseLoop();
    void parseSequential(ReadSoA &out, size_t maxReads);

    std::vector<std::string> files_;
    ReadFormat fmt_;
    int trim5_, trim3_;
    uint32_t globalSeed_;
    bool parallel_;
    bool pack_ = false;
    uint64_t startOffset_ = 0;
    std::unique_ptr<ReadSource> seqSrc_;     // raw / command-line formats: sequential path

    // Buffers go round: a parsed chunk handed out by next() leaves the caller's previous arrays behind, the parser threads
    // fill those next (and the raw file blocks likewise) — no chunk pays for fresh pages (first-touch faults cost several
    // times the copy itself).
    std::vector<ReadSoA> soaPool_;
    std::vector<CharBuf> rawPool_;
    int rangeFd_ = -1;                       // plain files: blocks are ranges the parsers pread themselves
    std::string rangePath_;
    int busy_ = 0;                           // parsers holding a block (guarded by mu_)

    std::mutex mu_;
    std::condition_variable cv_;
    std::deque<Raw> work_;
    std::map<uint64_t, ReadSoA> done_;
    uint64_t produced_ = 0, nextOut_ = 0;
    bool ioDone_ = false, stop_ = false;
    std::string error_;
    std::thread io_;
    std::vector<std::thread> parsers_;
    size_t maxInFlight_ = 8;
};

// a plain file dealt out as ranges (the reader's own blocks, and the front end's device text path): where the block that starts at
// pos — a record start — ends (a record start, or the end of the file), and the bytes of a range
uint64_t nextRecordCut(int fd, uint64_t pos, uint64_t fsize, size_t kBlock, bool fasta, const std::string &path);
void readFileRange(int fd, char *dst, size_t n, uint64_t off, const std::string &path);
// (mates on the text path: the second file is cut where it holds as many records as the first file's block)
uint64_t countByte(const char *p, size_t n, char c);
uint64_t behindNthByte(const char *p, size_t n, char c, uint64_t k);       // ~0: fewer than k occurrences

// one chunk of complete records -> SoA (exposed for the tests)
// lastOfFile: the chunk ends where the file ends — a record whose name line runs into the end of the file (or is
// followed by nothing but line ends) is not a read (FastaPatternSource::read bails out, pat.cpp:764-783)
void parseFastaChunk(const char *p, const char *e, bool firstOfFile, int trim5, int trim3, uint32_t globalSeed, ReadSoA &out, bool lastOfFile = false);
void parseFastqChunk(const char *p, const char *e, bool firstOfFile, int trim5, int trim3, uint32_t globalSeed, ReadSoA &out, bool lastOfFile = true);

}  // namespace cfamd
