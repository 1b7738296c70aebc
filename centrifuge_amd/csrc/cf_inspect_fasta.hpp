// cf_inspect_fasta.hpp — FASTA records out of the restored joined text (centrifuge-inspect's
// default mode).  Shared by cf_inspect.cpp and the CPU test harness (tests/emu), which feeds it
// the text its one-lane emulation of the restore kernels produced.
#pragma once
#include <cstdint>
#include <algorithm>
#include <cstdio>
#include <cstring>
#include <stdexcept>
#include <string>

#include "cf_index.hpp"

namespace cfamd {

struct Out {                                                                // buffered output stream
    std::FILE *f;
    std::string buf;
    explicit Out(std::FILE *fp) : f(fp) {}
    void flushIf(size_t lim = 1u << 20) { if (buf.size() >= lim) { std::fwrite(buf.data(), 1, buf.size(), f); buf.clear(); } }
    ~Out() { flushIf(1); std::fflush(f); }
};

// FASTA body with the reference's wrapping (print_fasta_record, centrifuge_inspect.cpp:191-211)
inline void fastaRecord(Out &o, const std::string &name, const std::string &seq, int across) {
    o.buf.push_back('>');
    o.buf += name;
    o.buf.push_back('\n');
    if (across > 0) {
        size_t i = 0;
        const size_t w = (size_t)across;
        while (i + w < seq.size()) { o.buf.append(seq, i, w); o.buf.push_back('\n'); i += w; o.flushIf(); }
        if (i < seq.size()) { o.buf.append(seq, i, std::string::npos); o.buf.push_back('\n'); }
    } else {
        o.buf += seq;
        o.buf.push_back('\n');
    }
    o.flushIf();
}

// characters [i0, i0 + cnt) of the 2-bit packed text appended as letters, four per table lookup
inline void unpackAppend(std::string &seq, const uint8_t *packed, uint64_t i0, uint64_t cnt) {
    static const struct Tab { char t[256][4]; Tab() { for (int b = 0; b < 256; b++) for (int j = 0; j < 4; j++) t[b][j] = "ACGT"[(b >> (2 * j)) & 3]; } } tab;
    const size_t at = seq.size();
    seq.resize(at + cnt);
    char *d = &seq[at];
    uint64_t i = i0;
    const uint64_t e = i0 + cnt;
    for (; i < e && (i & 3); i++) *d++ = "ACGT"[(packed[i >> 2] >> (2 * (i & 3))) & 3];
    for (; i + 4 <= e; i += 4, d += 4) std::memcpy(d, tab.t[packed[i >> 2]], 4);
    for (; i < e; i++) *d++ = "ACGT"[(packed[i >> 2] >> (2 * (i & 3))) & 3];
}

// print_index_sequences (centrifuge_inspect.cpp:369-430) over the restored joined text: one
// record per sequence that owns a fragment, gaps between fragments and both ends filled with N.
inline void printSequences(const HostIndex &h, const uint8_t *packed, int across, std::FILE *fp) {
    Out o(fp);
    const uint64_t nFrag = h.rstarts.size() / 3;
    uint64_t cur = ~0ull, curLen = 0, lastOff = 0;
    bool first = true;
    std::string seq;
    auto flush = [&]() {
        if (cur == ~0ull) return;
        if (seq.size() < curLen) seq.append(curLen - seq.size(), 'N');
        fastaRecord(o, cur < h.refnames.size() ? h.refnames[cur] : std::string(), seq, across);
    };
    for (uint64_t fi = 0; fi < nFrag; fi++) {
        const uint64_t lo = h.rstarts[3 * fi], hi = fi + 1 < nFrag ? h.rstarts[3 * fi + 3] : h.g.len;
        const uint64_t tidx = h.rstarts[3 * fi + 1], toff0 = h.rstarts[3 * fi + 2];
        if (tidx >= h.plen.size()) throw std::runtime_error("fragment table names a sequence that does not exist");
        const uint64_t tlen = h.plen[tidx];
        if (hi <= lo || toff0 >= tlen) continue;                 // positions at or past the sequence length are dropped (:390)
        const uint64_t cnt = std::min(hi - lo, tlen - toff0);
        if (cur != tidx) {
            flush();
            cur = tidx; seq.clear(); curLen = tlen; lastOff = 0; first = true;
            seq.reserve(tlen);
        }
        // the fragment's first character decides the gap in front of it (:405-408); the rest follow it directly
        const uint64_t adj = (first && toff0 > 0) ? toff0 + 1 : toff0;
        if (adj - lastOff > 1) seq.append(adj - lastOff - 1, 'N');
        unpackAppend(seq, packed, lo, cnt);
        lastOff = toff0 + cnt - 1;
        first = false;
    }
    if (cur < h.refnames.size()) flush();
}


}  // namespace cfamd
