// cf_inspect_fasta.hpp — FASTA records out of the restored joined text (centrifuge-inspect's
// default mode).  Shared by cf_inspect.cpp and the CPU test harness (tests/emu), which feeds it
// the text its one-lane emulation of the restore kernels produced.
#pragma once
#include <cstdint>
#include <cstdio>
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
        for (uint64_t i = lo; i < hi; i++) {
            const uint64_t toff = toff0 + (i - lo);
            if (toff >= tlen) continue;
            if (cur != tidx) {
                flush();
                cur = tidx; seq.clear(); curLen = tlen; lastOff = 0; first = true;
                seq.reserve(tlen);
            }
            const uint64_t adj = (first && toff > 0) ? toff + 1 : toff;
            if (adj - lastOff > 1) seq.append(adj - lastOff - 1, 'N');
            seq.push_back("ACGT"[(packed[i >> 2] >> (2 * (i & 3))) & 3]);
            lastOff = toff;
            first = false;
        }
    }
    if (cur < h.refnames.size()) flush();
}


}  // namespace cfamd
