// cf_build_host.hpp — host side of the index builder: reference ingest (FASTA or
// in-memory sequences) into the joined text + fragment table, and the writers of
// the parts of <base>.{1,3,4}.cf that do not need the suffix array.
// Format facts cite /root/reference; the code is our own.
#pragma once
#include <cstdint>
#include <cstdio>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "../../include/centrifuge_amd_build.h"

namespace cfamd {

// one stretch of unambiguous bases (RefRecord, ref_read.h): `off` ambiguous
// characters precede it, `first` = it opens a new sequence
struct RefRec {
    uint64_t off, len;
    bool first;
};

struct JoinedRef {
    // joined text: only A,C,G,T, one code per byte.  Either owned (`store`) or, for
    // gap-free in-memory input, a borrowed pointer into the caller's buffer.
    const uint8_t *text = nullptr;
    std::vector<uint8_t> store;
    uint64_t len = 0;
    std::vector<RefRec> szs;
    std::vector<std::string> refnames;      // header lines of the non-empty sequences
    std::vector<uint64_t> plen;             // per sequence: bases + ambiguous chars (bt2_idx.h:3274-3284)
    std::vector<uint64_t> rstarts;          // 3 per fragment: joined off, seq idx, off in seq (bt2_io.h:989-1027)
    std::vector<uint64_t> seqJoinedStart;   // per sequence: joined offset of its first base
    uint64_t nPat = 0, nFrag = 0;
};

// throws std::runtime_error
void ingestFasta(const std::vector<std::string> &paths, JoinedRef &out);
void ingestMemory(const uint8_t *codes, const uint64_t *seqOff, const char *const *names, uint64_t nSeq, JoinedRef &out);

// <base>.3.cf (bt2_idx.h:1316-1504)
void writeTaxonomyFile(const std::string &path, const JoinedRef &ref, const char *conversionTable,
                       const char *taxonomyTree, const char *nameTable, const char *sizeTable);

// little-endian scalar writers
template <typename T>
inline void put(std::FILE *f, T v) {
    if (std::fwrite(&v, sizeof v, 1, f) != 1) throw std::runtime_error("short write");
}
inline void putBytes(std::FILE *f, const void *p, size_t n) {
    if (n && std::fwrite(p, 1, n, f) != n) throw std::runtime_error("short write");
}

}  // namespace cfamd
