// cf_ingest.cpp — see cf_ingest.hpp
#include "cf_ingest.hpp"
#include <cerrno>
#include <unistd.h>
#include "cf_bytesource.hpp"

#include <algorithm>
#include <cctype>
#include <cstdlib>
#include <cstring>
#include <stdexcept>

#include "../../include/centrifuge_amd.h"
#include "cf_knobs.hpp"

#if defined(__x86_64__)
#include <immintrin.h>
#endif

namespace cfamd {

namespace {

struct Tables {
    uint8_t keep[256], code[256], alpha[256];
    Tables() {
        std::memset(keep, 0, sizeof keep); std::memset(code, 0, sizeof code); std::memset(alpha, 0, sizeof alpha);
        for (const char *s = "ABCDGHKMNRSTVWXYabcdghkmnrstvwxy-"; *s; s++) keep[(unsigned char)*s] = 1;   // alphabet.cpp:36-58
        code[(unsigned char)'C'] = code[(unsigned char)'c'] = 1;                                             // alphabet.cpp:298-319
        code[(unsigned char)'G'] = code[(unsigned char)'g'] = 2;
        code[(unsigned char)'T'] = code[(unsigned char)'t'] = 3;
        code[(unsigned char)'N'] = code[(unsigned char)'n'] = 4;
        for (int c = 0; c < 256; c++) alpha[c] = std::isalpha(c) ? 1 : 0;
    }
};
const Tables kT;

// ---- wide paths for the two things nearly every input byte is: an upper-case A/C/G/T, or a quality character.
//
// acgtRun: as many whole 32-byte groups at c as consist of upper-case A, C, G, T only -> their codes at w, and the bases'
// term of the read's seed (genRandSeed, pat.h:55-91: r ^= code << 2(i mod 16)) folded into r for a run that starts at base
// number i.  Returns the bytes taken (a multiple of 32); whatever follows — a lower-case or ambiguity letter, a newline,
// the last bytes of a line — goes through the byte loop of the caller.  (x >> 1) & 3 sends A,C,G,T to 0,1,3,2.
#if defined(__x86_64__)
__attribute__((target("avx2"))) static size_t acgtRunAvx2(const char *c, const char *end, uint8_t *w, uint32_t &r, uint32_t i) {
    const __m256i m3 = _mm256_set1_epi8(3);
    const __m256i tExp = _mm256_setr_epi8('A', 'C', 'T', 'G', 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 'A', 'C', 'T', 'G', 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0);
    const __m256i tCode = _mm256_setr_epi8(0, 1, 3, 2, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 1, 3, 2, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0);
    const __m256i k14 = _mm256_set1_epi16(0x0401), k116 = _mm256_set1_epi32(0x00100001);
    const unsigned s = (i & 15u) << 1;
    uint32_t acc = 0;
    const char *c0 = c;
    while (end - c >= 32) {
        const __m256i x = _mm256_loadu_si256(reinterpret_cast<const __m256i *>(c));
        const __m256i v = _mm256_and_si256(_mm256_srli_epi16(x, 1), m3);
        if (_mm256_movemask_epi8(_mm256_cmpeq_epi8(_mm256_shuffle_epi8(tExp, v), x)) != -1) break;
        const __m256i code = _mm256_shuffle_epi8(tCode, v);
        _mm256_storeu_si256(reinterpret_cast<__m256i *>(w), code);
        // 32 codes -> two 32-bit words of 16 two-bit fields each
        const __m256i p4 = _mm256_madd_epi16(_mm256_maddubs_epi16(code, k14), k116);          // per dword: 4 codes in 8 bits
        const __m256i p8 = _mm256_packus_epi16(_mm256_packus_epi32(p4, p4), _mm256_setzero_si256());   // per 128-bit lane: 4 bytes
        acc ^= (uint32_t)_mm256_extract_epi32(p8, 0) ^ (uint32_t)_mm256_extract_epi32(p8, 4);
        c += 32; w += 32;
    }
    r ^= s ? (acc << s) | (acc >> (32 - s)) : acc;           // the run's fields all sit 2(i mod 16) bits up, cyclically
    return (size_t)(c - c0);
}
static const bool kHaveAvx2 = __builtin_cpu_supports("avx2");
#else
static const bool kHaveAvx2 = false;
static size_t acgtRunAvx2(const char *, const char *, uint8_t *, uint32_t &, uint32_t) { return 0; }
#endif
inline size_t acgtRun(const char *c, const char *end, uint8_t *w, uint32_t &r, uint32_t i) {
    return kHaveAvx2 && end - c >= 32 ? acgtRunAvx2(c, end, w, r, i) : 0;
}
// the qualities' term of the seed: r ^= q[j] << 8(j mod 4) = the XOR of the string's little-endian dwords
inline uint32_t qualFold(const uint8_t *q, size_t n) {
    uint64_t a = 0;
    size_t j = 0;
    for (; j + 8 <= n; j += 8) { uint64_t x; std::memcpy(&x, q + j, 8); a ^= x; }
    uint32_t r = (uint32_t)a ^ (uint32_t)(a >> 32);
    for (; j < n; j++) r ^= (uint32_t)q[j] << ((j & 3) << 3);
    return r;
}
// smallest byte of a quality string (all bytes < 128 in any sane input; a byte >= 128 just takes the slow check)
inline bool anyBelow33(const uint8_t *q, size_t n) {
    uint64_t hit = 0;
    size_t j = 0;
    for (; j + 8 <= n; j += 8) { uint64_t x; std::memcpy(&x, q + j, 8); hit |= ((x - 0x2121212121212121ull) | x) & 0x8080808080808080ull; }
    if (hit) return true;
    for (; j < n; j++) if (q[j] < 33) return true;
    return false;
}

inline const char *lineEnd(const char *p, const char *e) {
    while (p < e && *p != '\n' && *p != '\r') p++;
    return p;
}
inline const char *skipNewlines(const char *p, const char *e) {
    while (p < e && (*p == '\n' || *p == '\r')) p++;
    return p;
}

}  // namespace

// ---- the packed form
template <typename T> void HostBuf<T>::release() {
    if (p) { if (pinned) cf_host_free(p); else std::free(p); }
    p = nullptr; n = cap = 0;
}
template <typename T> void HostBuf<T>::reserve(size_t count) {
    if (count <= cap) return;
    release();
    count += count / 8 + 64;
    void *q = nullptr;
    if (cf_host_alloc(&q, count * sizeof(T)) == CF_OK && q) pinned = true;      // no device (the CPU tests): plain memory
    else { q = std::malloc(count * sizeof(T)); pinned = false; if (!q) throw std::bad_alloc(); }
    p = static_cast<T *>(q); cap = count;
}
template <typename T> void HostBuf<T>::grow(size_t count, size_t keep) {
    if (count <= cap) return;
    HostBuf<T> bigger;
    bigger.reserve(std::max(count, 2 * cap));
    if (keep && p) std::memcpy(bigger.p, p, keep * sizeof(T));
    *this = std::move(bigger);
}
template struct HostBuf<uint64_t>;
template struct HostBuf<uint32_t>;

namespace {
// 32 base codes (0..3, anything above = N) -> 64 bits of 2-bit fields (N -> 0) and the 32 N bits
#if defined(__x86_64__)
__attribute__((target("avx2"))) inline void pack32Avx2(const uint8_t *c, uint64_t &w, uint32_t &m) {
    const __m256i x = _mm256_loadu_si256(reinterpret_cast<const __m256i *>(c));
    // unsigned: a code of 128 or more is an N as well, as in the scalar path (c > 3)
    const __m256i isN = _mm256_xor_si256(_mm256_cmpeq_epi8(_mm256_min_epu8(x, _mm256_set1_epi8(3)), x), _mm256_set1_epi8((char)0xff));
    m = (uint32_t)_mm256_movemask_epi8(isN);
    const __m256i code = _mm256_andnot_si256(isN, x);
    const __m256i p4 = _mm256_madd_epi16(_mm256_maddubs_epi16(code, _mm256_set1_epi16(0x0401)), _mm256_set1_epi32(0x00100001));   // per dword: 4 codes in 8 bits
    const __m256i p8 = _mm256_packus_epi16(_mm256_packus_epi32(p4, p4), _mm256_setzero_si256());                                 // per 128-bit lane: 4 bytes
    w = (uint64_t)(uint32_t)_mm256_extract_epi32(p8, 0) | ((uint64_t)(uint32_t)_mm256_extract_epi32(p8, 4) << 32);
}
#endif
inline void packTail(const uint8_t *c, size_t n, uint64_t &w, uint32_t &m) {          // n < 32 (or any n without AVX2: n <= 32)
    w = 0; m = 0;
    for (size_t j = 0; j < n; j++) {
        if (c[j] > 3) m |= 1u << j; else w |= (uint64_t)c[j] << (2 * j);
    }
}
}  // namespace

void ReadSoA::pack() {
    const size_t nr = size();
    uint64_t nw = 0;
    uint32_t mx = 0;
    for (size_t i = 0; i < nr; i++) { const uint64_t L = off[i + 1] - off[i]; nw += (L + 31) >> 5; mx = (uint32_t)std::max<uint64_t>(mx, std::min<uint64_t>(L, 0xffffffffull)); }
    pk.words.reserve(nw + 1); pk.lens.reserve(nr + 1); pk.seeds.reserve(nr + 1);
    // the sparse N mask: words that hold an N are few; the lists grow as needed (contents kept by hand)
    uint64_t nN = 0;
    auto pushN = [&](uint64_t idx, uint32_t m) {
        if (nN >= pk.nIdx.cap || nN >= pk.nMsk.cap) {
            HostBuf<uint64_t> a; HostBuf<uint32_t> b_;
            a.reserve(2 * nN + 1024); b_.reserve(2 * nN + 1024);
            if (nN) { std::memcpy(a.p, pk.nIdx.p, nN * 8); std::memcpy(b_.p, pk.nMsk.p, nN * 4); }
            pk.nIdx = std::move(a); pk.nMsk = std::move(b_);
        }
        pk.nIdx.p[nN] = idx; pk.nMsk.p[nN] = m; nN++;
    };
    uint64_t at = 0;
    pk.woff.resize(nr + 1);
    for (size_t i = 0; i < nr; i++) {
        const uint8_t *c = seq.data() + off[i];
        const uint64_t L = off[i + 1] - off[i];
        pk.woff[i] = at;
        pk.lens.p[i] = (uint32_t)L; pk.seeds.p[i] = seeds[i];
        uint64_t j = 0;
        for (; j + 32 <= L; j += 32, at++) {
            uint64_t w; uint32_t m;
#if defined(__x86_64__)
            if (kHaveAvx2) pack32Avx2(c + j, w, m); else packTail(c + j, 32, w, m);
#else
            packTail(c + j, 32, w, m);
#endif
            pk.words.p[at] = w;
            if (m) pushN(at, m);
        }
        if (j < L) {
            uint64_t w; uint32_t m;
            packTail(c + j, (size_t)(L - j), w, m);
            pk.words.p[at] = w;
            if (m) pushN(at, m);
            at++;
        }
    }
    pk.woff[nr] = at;
    pk.nReads = nr; pk.nWords = nw; pk.nBases = seq.size(); pk.nN = nN; pk.maxLen = mx;
    pk.valid = true; pk.appendable = false;
}

void ReadSoA::push(const uint8_t *s, const uint8_t *q, size_t len, const char *name, size_t nameLen, uint32_t seed) {
    pk.valid = false; pk.appendable = false;
    if (q && !hasQual) {                      // first read with qualities: earlier reads (none in practice) get 'I'
        qual.assign(seq.size(), (uint8_t)'I');
        hasQual = true;
    }
    seq.insert(seq.end(), s, s + len);
    if (hasQual) { if (q) qual.insert(qual.end(), q, q + len); else qual.insert(qual.end(), len, (uint8_t)'I'); }
    off.push_back(seq.size());
    names.append(name, nameLen);
    nameOff.push_back(names.size());
    seeds.push_back(seed);
}

void ReadSoA::appendRange(const ReadSoA &o, size_t i0, size_t i1) {
    pk.valid = false; pk.appendable = false;
    if (i1 <= i0) return;
    if (o.hasQual && !hasQual) { qual.assign(seq.size(), (uint8_t)'I'); hasQual = true; }
    const uint64_t s0 = o.off[i0], s1 = o.off[i1], n0 = o.nameOff[i0], n1 = o.nameOff[i1];
    const uint64_t sBase = seq.size(), nBase = names.size();
    seq.insert(seq.end(), o.seq.begin() + (long)s0, o.seq.begin() + (long)s1);
    if (hasQual) {
        if (o.hasQual) qual.insert(qual.end(), o.qual.begin() + (long)s0, o.qual.begin() + (long)s1);
        else qual.insert(qual.end(), s1 - s0, (uint8_t)'I');
    }
    names.append(o.names, n0, n1 - n0);
    const size_t at = off.size(), cnt = i1 - i0;
    off.resize(at + cnt);
    nameOff.resize(at + cnt);
    uint64_t *po = off.data() + at, *pn = nameOff.data() + at;
    const uint64_t *so = o.off.data() + i0 + 1, *sn = o.nameOff.data() + i0 + 1;
    const uint64_t dOff = sBase - s0, dName = nBase - n0;            // (modular: the sums are exact)
    for (size_t i = 0; i < cnt; i++) { po[i] = so[i] + dOff; pn[i] = sn[i] + dName; }
    seeds.insert(seeds.end(), o.seeds.begin() + (long)i0, o.seeds.begin() + (long)i1);
}

// the packed words of cnt mate pairs, interleaved like the bytes: every read starts on a word, so a read is a run of whole words
static void appendPackedInterleaved(PackedSoA &d, const PackedSoA &a, size_t ia, const PackedSoA &b, size_t ib, size_t cnt) {
    const uint64_t wa = a.woff[ia + cnt] - a.woff[ia], wb = b.woff[ib + cnt] - b.woff[ib];
    d.words.grow(d.nWords + wa + wb + 1, d.nWords);
    d.lens.grow(d.nReads + 2 * cnt + 1, d.nReads); d.seeds.grow(d.nReads + 2 * cnt + 1, d.nReads);
    const PackedSoA *src[2] = {&a, &b};
    const size_t idx0[2] = {ia, ib};
    size_t nAt[2];                                       // cursors into the sources' sparse N lists (sorted by word)
    for (int m = 0; m < 2; m++) nAt[m] = (size_t)(std::lower_bound(src[m]->nIdx.p, src[m]->nIdx.p + src[m]->nN, src[m]->woff[idx0[m]]) - src[m]->nIdx.p);
    for (size_t i = 0; i < cnt; i++) {
        for (int m = 0; m < 2; m++) {
            const PackedSoA &o = *src[m];
            const size_t j = idx0[m] + i;
            const uint64_t w0 = o.woff[j], nw = o.woff[j + 1] - w0;
            std::memcpy(d.words.p + d.nWords, o.words.p + w0, nw * 8);
            while (nAt[m] < o.nN && o.nIdx.p[nAt[m]] < w0 + nw) {          // the read's words that hold an N, at their new place
                d.nIdx.grow(d.nN + 1, d.nN); d.nMsk.grow(d.nN + 1, d.nN);
                d.nIdx.p[d.nN] = d.nWords + (o.nIdx.p[nAt[m]] - w0); d.nMsk.p[d.nN] = o.nMsk.p[nAt[m]];
                d.nN++; nAt[m]++;
            }
            d.lens.p[d.nReads] = o.lens.p[j]; d.seeds.p[d.nReads] = o.seeds.p[j];
            d.maxLen = std::max(d.maxLen, o.lens.p[j]);
            d.nBases += o.lens.p[j];
            d.nReads++; d.nWords += nw;
        }
    }
}

void ReadSoA::appendInterleaved(const ReadSoA &a, size_t ia, const ReadSoA &b, size_t ib, size_t cnt) {
    if (cnt == 0) return;
    if (pk.appendable && a.pk.valid && b.pk.valid && pk.nReads == size()) { appendPackedInterleaved(pk, a.pk, ia, b.pk, ib, cnt); pk.valid = true; }
    else { pk.valid = false; pk.appendable = false; }
    const bool q = a.hasQual || b.hasQual || hasQual;
    if (q && !hasQual) { qual.assign(seq.size(), (uint8_t)'I'); hasQual = true; }
    const uint64_t sa = a.off[ia + cnt] - a.off[ia], sb = b.off[ib + cnt] - b.off[ib];
    const uint64_t na = a.nameOff[ia + cnt] - a.nameOff[ia], nb = b.nameOff[ib + cnt] - b.nameOff[ib];
    size_t sAt = seq.size(), nAt = names.size();
    const size_t rAt = off.size();
    seq.resize(sAt + sa + sb);
    if (hasQual) qual.resize(sAt + sa + sb);
    names.resize(nAt + na + nb);
    off.resize(rAt + 2 * cnt); nameOff.resize(rAt + 2 * cnt);
    const size_t dAt = seeds.size();
    seeds.resize(dAt + 2 * cnt);
    uint8_t *ps = seq.data(), *pq = hasQual ? qual.data() : nullptr;
    char *pn = &names[0];
    uint64_t *po = off.data() + rAt, *pno = nameOff.data() + rAt;
    uint32_t *pd = seeds.data() + dAt;
    const ReadSoA *src[2] = {&a, &b};
    const size_t idx0[2] = {ia, ib};
    for (size_t i = 0; i < cnt; i++) {
        for (int m = 0; m < 2; m++) {
            const ReadSoA &o = *src[m];
            const size_t j = idx0[m] + i;
            const uint64_t s0 = o.off[j], len = o.off[j + 1] - s0, n0 = o.nameOff[j], nl = o.nameOff[j + 1] - n0;
            std::memcpy(ps + sAt, o.seq.data() + s0, len);
            if (pq) { if (o.hasQual) std::memcpy(pq + sAt, o.qual.data() + s0, len); else std::memset(pq + sAt, 'I', len); }
            std::memcpy(pn + nAt, o.names.data() + n0, nl);
            sAt += len; nAt += nl;
            *po++ = sAt; *pno++ = nAt; *pd++ = o.seeds[j];
        }
    }
}

// FastaPatternSource::read (pat.cpp:725-850) over a chunk of whole records.  Any '>' starts a
// record, as in the reference (it peeks for '>' after every character).
void parseFastaChunk(const char *p, const char *e, bool firstOfFile, int trim5, int trim3, uint32_t globalSeed, ReadSoA &out, bool lastOfFile) {
    if (firstOfFile) {
        for (;;) {
            p = skipNewlines(p, e);
            if (p < e && (*p == '#' || *p == ';')) p = lineEnd(p, e); else break;
        }
        if (p < e && *p != '>') throw std::runtime_error("Error: reads file does not look like a FASTA file");
    }
    if (trim5 == 0 && trim3 == 0 && !out.hasQual) {
        // Fast path: bases are translated straight into the batch array (sized for the worst case once, cut
        // back at the end) and the read's seed (genRandSeed, pat.h:55-91) is folded in the same pass.
        const uint32_t seed0 = (globalSeed + 101u) * 59u * 61u * 67u * 71u * 73u * 79u * 83u;
        const size_t base0 = out.seq.size();
        out.seq.resize(base0 + (size_t)(e - p));
        uint8_t *const s0 = out.seq.data();
        uint8_t *w = s0 + base0;
        while (p < e) {
            if (*p != '>') { p++; continue; }
            ++p;
            const char *name = p;
            while (p < e && *p != '\n' && *p != '\r' && *p != '>') p++;
            const size_t nameLen = (size_t)(p - name);
            p = skipNewlines(p, e);
            if (lastOfFile && p >= e) break;                 // the file ends in this record's name line: not a read
            const char *recEnd = static_cast<const char *>(std::memchr(p, '>', (size_t)(e - p)));
            if (!recEnd) recEnd = e;
            uint32_t r = seed0;
            uint32_t i = 0;
            for (const char *c = p; c < recEnd;) {
                const size_t run = acgtRun(c, recEnd, w, r, i);       // whole groups of plain bases
                c += run; w += run; i += (uint32_t)run;
                // then byte by byte to the end of the line (or past whatever stopped the run)
                const char *stop = recEnd - c > 40 ? c + 40 : recEnd;
                for (; c < stop; c++) {
                    const unsigned char ch = (unsigned char)*c;
                    const uint32_t k = kT.keep[ch], code = kT.code[ch];
                    *w = (uint8_t)code;
                    r ^= (k ? code : 0u) << ((i & 15) << 1);
                    w += k; i += k;
                    if (ch == '\n') { c++; break; }
                }
            }
            // qualities of a FASTA read are all 'I': their term depends on the length only
            uint32_t q = ((i >> 2) & 1) ? 0x49494949u : 0u;
            for (uint32_t j = 0; j < (i & 3); j++) q ^= 0x49u << (j << 3);
            r ^= q;
            for (size_t j = 0; j < nameLen; j++) {
                const int pc = (int)(signed char)name[j];
                if (pc == '/') break;
                r ^= (uint32_t)pc << ((j & 3) << 3);
            }
            out.off.push_back((uint64_t)(w - s0));
            out.names.append(name, nameLen);
            out.nameOff.push_back(out.names.size());
            out.seeds.push_back(r);
            p = recEnd;
        }
        out.seq.resize((size_t)(w - s0));
        return;
    }
    out.seq.reserve(out.seq.size() + (size_t)(e - p));
    std::vector<uint8_t> tmp;
    while (p < e) {
        if (*p != '>') { p++; continue; }                      // (only stray bytes before the first record of a later chunk)
        ++p;
        const char *name = p;
        while (p < e && *p != '\n' && *p != '\r' && *p != '>') p++;
        const size_t nameLen = (size_t)(p - name);
        p = skipNewlines(p, e);
        if (lastOfFile && p >= e) break;                     // the file ends in this record's name line: not a read
        const char *recEnd = static_cast<const char *>(std::memchr(p, '>', (size_t)(e - p)));
        if (!recEnd) recEnd = e;
        tmp.clear();
        int begin = 0;
        for (const char *c = p; c < recEnd; c++) {
            const unsigned char ch = (unsigned char)*c;
            if (kT.keep[ch] && begin++ >= trim5) tmp.push_back(kT.code[ch]);
        }
        if (trim3 > 0) { if (tmp.size() > (size_t)trim3) tmp.resize(tmp.size() - (size_t)trim3); else tmp.clear(); }
        const uint32_t seed = cf_gen_rand_seed(tmp.data(), nullptr, tmp.size(), name, nameLen, globalSeed);
        out.push(tmp.data(), nullptr, tmp.size(), name, nameLen, seed);
        p = recEnd;
    }
}

// FastqPatternSource::read (pat.cpp:852-1100): name line, every letter up to the '+' line (over any number of lines), the
// '+' line, ONE line of phred33 character qualities
void parseFastqChunk(const char *p, const char *e, bool firstOfFile, int trim5, int trim3, uint32_t globalSeed, ReadSoA &out, bool lastOfFile) {
    if (firstOfFile && p < e && *p != '@') { p = lineEnd(p, e); p = skipNewlines(p, e); }
    std::vector<uint8_t> s, q;
    out.hasQual = true;
    if (out.qual.size() < out.seq.size()) out.qual.resize(out.seq.size(), (uint8_t)'I');
    if (trim5 == 0 && trim3 == 0) {
        // Fast path: bases and qualities go straight into the batch arrays (sized for the worst case once, cut
        // back at the end), the seed (genRandSeed, pat.h:55-91) is folded in the same passes.  Same checks and
        // messages as the general loop below.
        const uint32_t seed0 = (globalSeed + 101u) * 59u * 61u * 67u * 71u * 73u * 79u * 83u;
        const size_t base0 = out.seq.size();
        out.seq.resize(base0 + (size_t)(e - p));
        out.qual.resize(base0 + (size_t)(e - p));
        uint8_t *const s0 = out.seq.data(), *const q0 = out.qual.data();
        size_t at = base0;
        auto fail = [&](const std::string &m) { out.seq.resize(at); out.qual.resize(at); throw std::runtime_error(m); };
        bool eaten = false;                                  // the record before took this record's first character
        while (p < e) {
            if (!eaten) {
                p = skipNewlines(p, e);
                if (p >= e) break;
                if (*p != '@') fail("Error: reads file does not look like a FASTQ file");
                ++p;
            }
            eaten = false;
            const char *name = p;
            p = lineEnd(p, e);
            const size_t nameLen = (size_t)(p - name);
            p = skipNewlines(p, e);
            if (p >= e) break;                               // the input ends in a name line: not a read (pat.cpp:887-900)
            // the sequence: every letter up to the first '+', over as many lines as it takes (pat.cpp:932-975 reads
            // character by character and skips what is not a letter, line ends included)
            uint32_t r = seed0;
            uint8_t *w = s0 + at;
            uint32_t i = 0;
            bool plus = false;
            while (p < e && !plus) {
                if (*p == '+') { plus = true; break; }
                const char *le = lineEnd(p, e);
                const size_t run = acgtRun(p, le, w, r, i);           // the line's leading groups of plain bases
                w += run; i += (uint32_t)run;
                const char *c = p + run;
                for (; c < le; c++) {
                    unsigned char ch = (unsigned char)*c;
                    if (ch == '+') { plus = true; break; }
                    if (ch == '.') ch = 'N';
                    const uint32_t k = kT.alpha[ch], code = kT.code[ch];
                    *w = (uint8_t)code;
                    r ^= (k ? code : 0u) << ((i & 15) << 1);
                    w += k; i += k;
                }
                p = plus ? c : skipNewlines(le, e);
            }
            const size_t n = (size_t)(w - (s0 + at));
            const char *le;
            if (!plus) {
                if (lastOfFile) break;                           // the input ends inside a sequence: not a read (pat.cpp:966-969)
                fail("Error: reads file does not look like a FASTQ file");
            }
            p = lineEnd(p, e);
            p = skipNewlines(p, e);
            if (n > 0) {
                le = lineEnd(p, e);
                size_t nq = (size_t)(le - p);
                if (nq && anyBelow33(reinterpret_cast<const uint8_t *>(p), nq)) {   // report the first offender the way the general loop does
                    for (const char *c = p; c < le; c++) {
                        const unsigned char ch = (unsigned char)*c;
                        if (ch == ' ') fail("Error: reads file contains a pattern with a space in the quality string");
                        if (ch < 33) fail("Saw ASCII character " + std::to_string((int)ch) + " but expected 33-based Phred qual.");
                    }
                }
                if (nq < n) fail("Error: Read " + std::string(name, nameLen) + " has more read characters than quality values.");
                if (nq > n + 1) fail("Error: Read " + std::string(name, nameLen) + " has more quality values than read characters.");
                std::memcpy(q0 + at, p, n);
                r ^= qualFold(q0 + at, n);
                p = le;
            } else {
                // A record without a single base letter leaves the reference's reader early (pat.cpp:985-993): the
                // next character — the next record's '@' in a well-formed file — is taken unseen, and a missing
                // name is NOT replaced by the read's ordinal.
                if (p < e) { p++; eaten = true; }
                if (nameLen == 0) out.unnamedKeep.push_back((uint32_t)(out.off.size() - 1));
            }
            for (size_t j = 0; j < nameLen; j++) {
                const int pc = (int)(signed char)name[j];
                if (pc == '/') break;
                r ^= (uint32_t)pc << ((j & 3) << 3);
            }
            at += n;
            out.off.push_back((uint64_t)at);
            out.names.append(name, nameLen);
            out.nameOff.push_back(out.names.size());
            out.seeds.push_back(r);
        }
        out.seq.resize(at); out.qual.resize(at);
        return;
    }
    bool eaten = false;
    while (p < e) {
        if (!eaten) {
            p = skipNewlines(p, e);
            if (p >= e) break;
            if (*p != '@') throw std::runtime_error("Error: reads file does not look like a FASTQ file");
            ++p;
        }
        eaten = false;
        const char *name = p;
        p = lineEnd(p, e);
        const size_t nameLen = (size_t)(p - name);
        p = skipNewlines(p, e);
        if (p >= e) break;
        s.clear(); q.clear();
        int charsRead = 0;
        const bool emptyLine = p < e && *p == '+';               // empty sequence line was swallowed with the newlines
        // every letter up to the first '+', over as many lines as it takes (pat.cpp:932-975)
        while (p < e && *p != '+') {
            unsigned char ch = (unsigned char)*p++;
            if (ch == '.') ch = 'N';
            if (kT.alpha[ch]) { if (charsRead >= trim5) s.push_back(kT.code[ch]); charsRead++; }
        }
        if (p >= e) {
            if (lastOfFile) break;                               // the input ends inside a sequence: not a read (pat.cpp:966-969)
            throw std::runtime_error("Error: reads file does not look like a FASTQ file");
        }
        const char *le;
        p = lineEnd(p, e);
        p = skipNewlines(p, e);
        if (trim3 > 0) { if (s.size() > (size_t)trim3) s.resize(s.size() - (size_t)trim3); else s.clear(); }
        // pat.cpp:985: an empty sequence line, or (without a 5' trim) a line without a base letter
        const bool noLetters = emptyLine || (trim5 == 0 && charsRead == 0);
        if (noLetters) {
            if (p < e) { p++; eaten = true; }
            if (nameLen == 0) out.unnamedKeep.push_back((uint32_t)(out.off.size() - 1));
        } else {
            le = lineEnd(p, e);
            int qualsRead = 0;
            for (const char *c = p; c < le; c++) {
                const unsigned char ch = (unsigned char)*c;
                if (ch == ' ') throw std::runtime_error("Error: reads file contains a pattern with a space in the quality string");
                if (qualsRead >= trim5) {
                    if (ch < 33) throw std::runtime_error("Saw ASCII character " + std::to_string((int)ch) + " but expected 33-based Phred qual.");
                    q.push_back(ch);
                }
                qualsRead++;
            }
            p = le;
            if (trim3 > 0) { if (q.size() > (size_t)trim3) q.resize(q.size() - (size_t)trim3); else q.clear(); }
            const std::string nm(name, nameLen);
            if (q.size() < s.size()) throw std::runtime_error("Error: Read " + nm + " has more read characters than quality values.");
            if (q.size() > s.size() + 1) throw std::runtime_error("Error: Read " + nm + " has more quality values than read characters.");
            if (q.size() > s.size()) q.resize(s.size());
        }
        const uint32_t seed = cf_gen_rand_seed(s.data(), q.empty() ? nullptr : q.data(), s.size(), name, nameLen, globalSeed);
        out.push(s.data(), q.empty() && s.empty() ? reinterpret_cast<const uint8_t *>("") : q.data(), s.size(), name, nameLen, seed);
    }
}

// ----------------------------------------------------------------------------------------
ChunkedReader::ChunkedReader(std::vector<std::string> files, ReadFormat fmt, int trim5, int trim3, uint32_t globalSeed, int threads, bool pack, uint64_t startOffset)
    : files_(std::move(files)), fmt_(fmt), trim5_(trim5), trim3_(trim3), globalSeed_(globalSeed),
      parallel_(fmt == ReadFormat::Fasta || fmt == ReadFormat::Fastq), pack_(pack), startOffset_(startOffset) {
    if (!parallel_) { seqSrc_.reset(new ReadSource(files_, fmt_, trim5_, trim3_)); return; }
    const int n = std::max(1, threads);
    maxInFlight_ = (size_t)n * 2 + 2;
    io_ = std::thread([this] { ioLoop(); });
    for (int i = 0; i < n; i++) parsers_.emplace_back([this] { parseLoop(); });
}

ChunkedReader::~ChunkedReader() {
    { std::lock_guard<std::mutex> lk(mu_); stop_ = true; }
    cv_.notify_all();
    if (io_.joinable()) io_.join();
    for (auto &t : parsers_) if (t.joinable()) t.join();
}

// Offset of the last record start in [1, len) of a stretch of the file (which may begin in the middle of a line), 0 = none.
// FASTA: any '>' starts a record.  FASTQ: a line starting with '@' that is followed by sequence lines and then a complete
// line starting with '+' (a quality line may start with '@' too; the line after it never looks like a sequence).
// lenient: take the last name line that has sequence lines and a complete '+' line behind it without asking the quality
// lines to match — the fallback for a stretch that has grown past a few blocks without an accepted candidate (a file whose
// quality strings do not match its sequences record after record: the parser deals with those, as the reference's does; the
// reader must not swallow the whole file into one block looking for a cut point).
static size_t lastRecordStart(const char *bp, size_t len, bool fasta, bool lenient = false) {
    if (fasta) {
        for (size_t i = len; i-- > 1;) if (bp[i] == '>') return i;
        return 0;
    }
    size_t i = len;
    while (i > 1) {
        size_t ls = i - 1;
        while (ls > 0 && bp[ls - 1] != '\n') ls--;                   // start of the line containing i-1
        if (ls > 0 && bp[ls] == '@') {
            // a name line: what follows it are sequence lines — anything the parser takes as one (it skips whatever is no base
            // letter: blanks, digits, colour-space characters), i.e. any line that does not start with '@': the next record's
            // name line, which follows a quality line that merely starts with '@', is the one thing that is not — up to a
            // complete line that starts with '+'
            const char *b = bp, *e = b + len;
            const char *l = skipNewlines(lineEnd(b + ls, e), e);
            size_t seqLen = 0;                                       // characters on the candidate's sequence lines
            for (;;) {
                if (l >= e) break;
                const char *le = lineEnd(l, e);
                if (*l == '+') {
                    if (le >= e) break;
                    // ... and, so that a block of wrapped quality lines (one starting with '@', a later one with '+') is not taken for
                    // a record: the quality lines behind the '+' must hold exactly as many characters as the sequence lines did, and
                    // all of them must lie inside the stretch (a candidate that cannot be checked is passed over: there are
                    // thousands of record starts before it)
                    if (lenient && seqLen > 0) return ls;
                    const char *q = skipNewlines(le, e);
                    size_t qLen = 0;
                    bool ok = false;
                    while (q < e && qLen < seqLen) {
                        const char *qe = lineEnd(q, e);
                        if (qe >= e) break;                      // the line runs out of the stretch
                        size_t n = (size_t)(qe - q);
                        while (n && (q[n - 1] == '\r')) n--;
                        qLen += n;
                        q = skipNewlines(qe, e);
                    }
                    ok = qLen == seqLen && seqLen > 0;
                    if (ok) return ls;
                    break;
                }
                if (le >= e || *l == '@') break;
                // what the parser keeps of a sequence line: letters, a '.' as the N it becomes — up to a '+' in mid-line, where the
                // parser's sequence ends and its '+' line begins
                const char *c = l;
                for (; c < le && *c != '+'; c++) { const unsigned char ch = (unsigned char)*c; seqLen += kT.alpha[ch == '.' ? (unsigned char)'N' : ch]; }
                l = c < le ? c : skipNewlines(le, e);
            }
        }
        i = ls;
    }
    return 0;
}

static void preadFull(int fd, char *dst, size_t n, uint64_t off, const std::string &path) {
    while (n) {
        const ssize_t got = ::pread(fd, dst, n, (off_t)off);
        if (got < 0) { if (errno == EINTR) continue; throw std::runtime_error("Error: I/O error while reading \"" + path + "\""); }
        if (got == 0) throw std::runtime_error("Error: \"" + path + "\" changed while it was read");
        dst += got; off += (uint64_t)got; n -= (size_t)got;
    }
}
void readFileRange(int fd, char *dst, size_t n, uint64_t off, const std::string &path) { preadFull(fd, dst, n, off, path); }

// ---- mates on the device text path: the block of the second file that holds as many records as the first file's block does, by
//      the rule the device's record pass counts by — a FASTA record starts at every '>', a FASTQ record is four lines.  (Whether the
//      records have the plain form is the device's to say; these only count bytes, 32 at a time.)
#if defined(__x86_64__)
__attribute__((target("avx2"))) static uint64_t countByteAvx2(const char *p, size_t n, char c) {
    const __m256i k = _mm256_set1_epi8(c);
    uint64_t total = 0;
    size_t i = 0;
    for (; i + 32 <= n; i += 32)
        total += (uint64_t)__builtin_popcount((unsigned)_mm256_movemask_epi8(_mm256_cmpeq_epi8(_mm256_loadu_si256(reinterpret_cast<const __m256i *>(p + i)), k)));
    for (; i < n; i++) total += p[i] == c;
    return total;
}
#endif
uint64_t countByte(const char *p, size_t n, char c) {
#if defined(__x86_64__)
    if (kHaveAvx2) return countByteAvx2(p, n, c);
#endif
    uint64_t total = 0;
    for (size_t i = 0; i < n; i++) total += p[i] == c;
    return total;
}
// offset just behind the k-th occurrence of c in p[0, n) (k >= 1), or ~0 when there are fewer
uint64_t behindNthByte(const char *p, size_t n, char c, uint64_t k) {
    size_t i = 0;
    uint64_t seen = 0;
    while (i < n) {                                                  // whole 4 KiB pieces while the target lies beyond them
        const size_t step = std::min<size_t>(4096, n - i);
        const uint64_t here = countByte(p + i, step, c);
        if (seen + here >= k) break;
        seen += here; i += step;
    }
    for (; i < n; i++) if (p[i] == c && ++seen == k) return (uint64_t)i + 1;
    return ~0ull;
}

// Where the block of a plain file that starts at pos (a record start) ends: the last record start in (pos, pos + kBlock] — a look
// at the last 256 KiB of the block, further back if need be — or, for a record larger than the block, in the blocks behind it;
// the end of the file when that comes first.
uint64_t nextRecordCut(int fd, uint64_t pos, uint64_t fsize, size_t kBlock, bool fasta, const std::string &path) {
    uint64_t end = pos + kBlock, cut = 0;
    std::vector<char> win;
    for (;;) {
        if (end >= fsize) return fsize;
        for (uint64_t T = std::min<uint64_t>(256u << 10, kBlock); cut == 0; T *= 8) {
            const uint64_t ws = end - pos > T ? end - T : pos;
            win.resize((size_t)(end - ws));
            preadFull(fd, win.data(), win.size(), ws, path);
            const size_t local = lastRecordStart(win.data(), win.size(), fasta);
            if (local) cut = ws + local;
            if (ws == pos) {
                // nothing in the whole stretch passes the check: past a few blocks, the lenient rule (see lastRecordStart)
                if (!cut && !fasta && end - pos > 4 * (uint64_t)kBlock) { const size_t l2 = lastRecordStart(win.data(), win.size(), false, true); if (l2) cut = ws + l2; }
                break;
            }
        }
        if (cut) return cut;
        end += kBlock;                           // one record larger than the block
    }
}

void ChunkedReader::ioLoop() {
    // bytes per block: 32 MiB = ~280 k reads of 100 bases (CF_INGEST_BLOCK: the tests cut the input into many small blocks)
    const size_t kBlock = cfamd::cf_knob("CF_INGEST_BLOCK") ? std::max<size_t>(4096, std::strtoull(cfamd::cf_knob("CF_INGEST_BLOCK"), nullptr, 10)) : (size_t)(32u << 20);
    try {
        for (const std::string &path : files_) {
            ByteSource src(path, (int)std::max<size_t>(1, parsers_.size()));      // plain / stdin / gzip (in-process) / bzip2; throws when it cannot be opened
            int fd = -1; uint64_t fsize = 0;
            if ((!cfamd::cf_knob("CF_INGEST_STREAM") || startOffset_) && src.regularFile(fd, fsize)) {
                // A plain file is dealt out as RANGES: this thread only finds where records start (a look at the last
                // 256 KiB of every block), the parser threads read their range themselves — the copy out of the page cache,
                // the one serial pass that was left, runs on as many threads as parse.
                rangeFd_ = fd; rangePath_ = path;
                uint64_t pos = std::min(startOffset_, fsize);
                startOffset_ = 0;                            // (the first file only)
                bool first = pos == 0;
                while (pos < fsize) {
                    const uint64_t cut = nextRecordCut(fd, pos, fsize, kBlock, fmt_ == ReadFormat::Fasta, path);
                    Raw r;
                    r.first = first; first = false;
                    r.last = cut == fsize;
                    r.fd = fd; r.foff = pos; r.flen = cut - pos;
                    pos = cut;
                    std::unique_lock<std::mutex> lk(mu_);
                    cv_.wait(lk, [&] { return stop_ || produced_ - nextOut_ < maxInFlight_; });
                    if (stop_) return;
                    r.seq = produced_++;
                    work_.push_back(std::move(r));
                    lk.unlock();
                    cv_.notify_all();
                }
                // the descriptor must outlive the parsers' reads: wait until every block of this file is parsed
                std::unique_lock<std::mutex> lk(mu_);
                cv_.wait(lk, [&] { return stop_ || (work_.empty() && busy_ == 0); });
                if (stop_) return;
                continue;
            }
            // The file is read straight into the block a parser will get (one copy: the read itself); what follows the
            // last whole record is carried over to the front of the next block.
            std::vector<char> carry;
            CharBuf buf;
            bool first = true, eof = false;
            while (!eof) {
                if (!buf.p) {
                    { std::lock_guard<std::mutex> lk(mu_); if (!rawPool_.empty()) { buf = std::move(rawPool_.back()); rawPool_.pop_back(); } }
                    buf.len = 0;
                    buf.ensure(carry.size() + kBlock);
                    if (!carry.empty()) std::memcpy(buf.p.get(), carry.data(), carry.size());
                    buf.len = carry.size();
                    carry.clear();
                } else buf.ensure(buf.len + kBlock);           // one record larger than a block: keep reading into the same one
                const size_t got = src.read(buf.p.get() + buf.len, kBlock);
                buf.len += got;
                eof = got < kBlock;
                const char *bp = buf.p.get();
                size_t cut = buf.len;
                if (!eof) {                                  // last record start inside the buffer
                    cut = lastRecordStart(bp, buf.len, fmt_ == ReadFormat::Fasta);
                    if (cut == 0 && fmt_ != ReadFormat::Fasta && buf.len > 4 * kBlock) cut = lastRecordStart(bp, buf.len, false, true);
                    if (cut == 0) continue;                  // one record larger than the block: keep reading
                }
                carry.assign(bp + cut, bp + buf.len);
                Raw r;
                r.first = first; first = false;
                r.last = eof;
                buf.len = cut;
                r.data = std::move(buf);
                buf = CharBuf();
                if (r.data.len == 0) { std::lock_guard<std::mutex> lk(mu_); if (rawPool_.size() < 64) rawPool_.push_back(std::move(r.data)); continue; }
                std::unique_lock<std::mutex> lk(mu_);
                cv_.wait(lk, [&] { return stop_ || produced_ - nextOut_ < maxInFlight_; });
                if (stop_) return;
                r.seq = produced_++;
                work_.push_back(std::move(r));
                lk.unlock();
                cv_.notify_all();
            }
        }
    } catch (const std::exception &ex) {
        std::lock_guard<std::mutex> lk(mu_);
        if (error_.empty()) error_ = ex.what();
    }
    { std::lock_guard<std::mutex> lk(mu_); ioDone_ = true; }
    cv_.notify_all();
}

void ChunkedReader::parseLoop() {
    for (;;) {
        Raw r;
        {
            std::unique_lock<std::mutex> lk(mu_);
            cv_.wait(lk, [&] { return stop_ || !work_.empty() || ioDone_; });
            if (stop_ || (work_.empty() && ioDone_)) return;
            if (work_.empty()) continue;
            r = std::move(work_.front());
            work_.pop_front();
            busy_++;
            if (r.fd >= 0 && !rawPool_.empty()) { r.data = std::move(rawPool_.back()); rawPool_.pop_back(); }
        }
        ReadSoA out;
        { std::lock_guard<std::mutex> lk(mu_); if (!soaPool_.empty()) { out = std::move(soaPool_.back()); soaPool_.pop_back(); } }
        out.clear(); out.hasQual = false;
        try {
            if (r.fd >= 0) {
                r.data.len = 0;
                r.data.ensure((size_t)r.flen);
                preadFull(r.fd, r.data.p.get(), (size_t)r.flen, r.foff, rangePath_);
                r.data.len = (size_t)r.flen;
            }
            const char *p = r.data.p.get(), *e = p + r.data.len;
            if (fmt_ == ReadFormat::Fasta) parseFastaChunk(p, e, r.first, trim5_, trim3_, globalSeed_, out, r.last);
            else parseFastqChunk(p, e, r.first, trim5_, trim3_, globalSeed_, out, r.last);
            if (pack_) out.pack();
        } catch (const std::exception &ex) {
            std::lock_guard<std::mutex> lk(mu_);
            if (error_.empty()) error_ = ex.what();
        }
        {
            std::lock_guard<std::mutex> lk(mu_);
            done_[r.seq] = std::move(out);
            if (rawPool_.size() < 64) rawPool_.push_back(std::move(r.data));
            busy_--;
        }
        cv_.notify_all();
    }
}

void ChunkedReader::parseSequential(ReadSoA &out, size_t maxReads) {
    ReadRec r;
    while (out.size() < maxReads && seqSrc_->next(r)) {
        const uint32_t seed = cf_gen_rand_seed(r.seq.data(), r.qual.empty() ? nullptr : r.qual.data(), r.seq.size(), r.name.data(),
                                               r.name.size(), globalSeed_);
        out.push(r.seq.data(), r.qual.empty() ? nullptr : r.qual.data(), r.seq.size(), r.name.data(), r.name.size(), seed);
    }
}

bool ChunkedReader::next(ReadSoA &out) {
    out.clear();
    out.hasQual = false;
    if (!parallel_) {
        parseSequential(out, 1u << 16);
        return out.size() > 0;
    }
    std::unique_lock<std::mutex> lk(mu_);
    cv_.wait(lk, [&] { return !error_.empty() || done_.count(nextOut_) || (ioDone_ && nextOut_ == produced_); });
    if (!error_.empty()) throw std::runtime_error(error_);
    auto it = done_.find(nextOut_);
    if (it == done_.end()) return false;
    if (out.seq.capacity() && soaPool_.size() < 64) soaPool_.push_back(std::move(out));      // the caller's old arrays: the next chunk is parsed into them
    out = std::move(it->second);
    done_.erase(it);
    nextOut_++;
    lk.unlock();
    cv_.notify_all();
    return true;
}

}  // namespace cfamd
