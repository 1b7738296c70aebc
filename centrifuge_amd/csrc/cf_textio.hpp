// cf_textio.hpp — the front end's ingest and egress ON THE DEVICE (round 6).
//
// Ingest: a block of whole FASTA / FASTQ records goes up as the file holds it (113 bytes per 100-base read); the kernels below find
// the records, check that every one of them has the PLAIN form — the one form in which this file and the host parser
// (cf_ingest.cpp: FastaPatternSource::read pat.cpp:725-850, FastqPatternSource::read pat.cpp:852-1100) cannot disagree —, and make
// what the batch needs: lengths, the per-read seeds (genRandSeed, pat.h:55-91), the 2-bit words and N masks.  A block with ANY
// record outside the plain form is not parsed here at all: the status says so and the caller hands that block to the host parser
// (whose semantics are the reference's, record by record).  The plain form:
//   FASTA  the block starts with '>'; every '>' starts a record; a record is a non-empty name line without '\r', then one or more
//          lines whose characters are A C G T N in either case, at least one of them
//   FASTQ  four lines per record: '@' + non-empty name without '\r'; a non-empty line of A C G T N in either case; a line that
//          starts with '+'; as many quality characters (all >= 33) as bases; the block ends with the last record's '\n'
// Egress: the default eight columns of AlnSinkSam::appendMate (aln_sink.h:2279-2337; centrifuge.cpp:520) are formatted here from
// the narrow rows the batch leaves on the device, the readID copied out of the uploaded block (aln_sink.h:2203-2217), into a
// buffer the host only write()s; and the part of SpeciesMetrics (aln_sink.h:142-172) the per-taxon counters of count_body do not
// hold — the perfect single assignments and the perfect multi-assignment tuples the EM runs on — is tallied by the same pass.
//
// Every body is one thread per item and runs in the CPU harness (tests/emu) as a plain loop; the cross-lane parts are sums only.
#pragma once
#include "cf_platform.hpp"

namespace cfamd {

constexpr uint32_t kTextPiece = 64;                     // bytes per thread of the two marker passes
constexpr uint32_t kTextPad = 128;                      // zero bytes the uploaded block is followed by (whole-piece and whole-word loads)
enum : uint32_t { kTextFasta = 0, kTextFastq = 1 };
// why a block is not in the plain form (TextStatus::flags; any bit = the host parses it)
enum : uint32_t {
    kTxBadStart = 1u, kTxNoNameEnd = 2u, kTxEmptyName = 4u, kTxCarriageReturn = 8u, kTxBadBase = 16u, kTxEmptySeq = 32u,
    kTxBadPlus = 64u, kTxQualLen = 128u, kTxBadQual = 256u, kTxLineCount = 512u, kTxTooMany = 1024u,
    kTxMateCount = 2048u                                 // (made by the host: the two blocks of a paired upload hold different numbers of records)
};

constexpr uint32_t kTextStripes = 64;                   // the block's sums are kept in that many places (a wavefront adds to one of them:
                                                        // ten thousand atomics on ONE address took most of the record pass), added up by the host
struct TextStatus {
    unsigned long long nWords[kTextStripes], nBases[kTextStripes];   // sums over the records: packed words, bases
    unsigned long long outBytes;                        // egress: bytes of the formatted rows
    uint32_t maxLen, flags;
    uint32_t tupleWords, pad;                           // egress: words of the tuple list that are filled
    unsigned long long words() const { unsigned long long t = 0; for (uint32_t i = 0; i < kTextStripes; i++) t += nWords[i]; return t; }
    unsigned long long bases() const { unsigned long long t = 0; for (uint32_t i = 0; i < kTextStripes; i++) t += nBases[i]; return t; }
};

CF_DEV uint64_t tx_load8(const uint8_t *base, uint64_t off) {
    const uint64_t a = off & ~7ull;
    const uint32_t sh = (uint32_t)(off & 7) * 8;
    const uint64_t lo = cf_load8(base + a);
    if (sh == 0) return lo;
    return (lo >> sh) | (cf_load8(base + a + 8) << (64 - sh));
}
// the bytes of a block one after the other, fetched as aligned 8-byte words
struct TxCursor {
    const uint8_t *base;
    uint64_t at, w;
    CF_DEV void seek(const uint8_t *b, uint64_t p) { base = b; at = p; w = cf_load8(b + (p & ~7ull)) >> ((p & 7) * 8); }
    CF_DEV uint32_t next() {
        const uint32_t c = (uint32_t)w & 0xffu;
        at++;
        if ((at & 7) == 0) w = cf_load8(base + at); else w >>= 8;
        return c;
    }
};
// 0x80 in every byte of x that equals the marker (exact: no carries between bytes)
CF_DEV uint64_t tx_match8(uint64_t x, uint32_t marker) {
    const uint64_t y = x ^ (0x0101010101010101ull * marker);
    const uint64_t t = (y & 0x7f7f7f7f7f7f7f7full) + 0x7f7f7f7f7f7f7f7full;
    return ~(t | y | 0x7f7f7f7f7f7f7f7full);
}

// ---- pass 1 and 2: where the markers are ('>' of FASTA: the record starts; '\n' of FASTQ: the line ends), in file order.
// Thread t owns the bytes [64 t, 64 t + 64): it counts its markers (cnt), and — after the exclusive sums of the counts (base) —
// writes their positions.  The block is followed by zero bytes, which are no marker.
struct DTextMark {
    const uint8_t *text;
    uint64_t nBytes;
    uint32_t marker;
    uint32_t *cnt;               // per piece
    const uint64_t *base;        // exclusive sums of cnt (nPieces + 1)
    uint32_t *pos;               // marker positions
    uint64_t posCap;
};
CF_DEV void text_count_body(const DTextMark &m, uint64_t t) {
    const uint64_t o = t * kTextPiece;
    if (o >= m.nBytes) return;
    uint32_t n = 0;
#pragma unroll
    for (uint32_t k = 0; k < kTextPiece / 16; k++) {
        const u64x2 v = cf_load16(m.text + o + 16 * k);
        n += (uint32_t)cf_popc64(tx_match8(v.x, m.marker)) + (uint32_t)cf_popc64(tx_match8(v.y, m.marker));
    }
    m.cnt[t] = n;
}
CF_DEV void text_mark_body(const DTextMark &m, uint64_t t) {
    const uint64_t o = t * kTextPiece;
    if (o >= m.nBytes) return;
    uint64_t at = m.base[t];
#pragma unroll
    for (uint32_t k = 0; k < kTextPiece / 16; k++) {
        const u64x2 v = cf_load16(m.text + o + 16 * k);
        uint64_t a = tx_match8(v.x, m.marker), b = tx_match8(v.y, m.marker);
        while (a) { const int i = cf_ctz64(a) >> 3; if (at < m.posCap) m.pos[at] = (uint32_t)(o + 16 * k + (uint32_t)i); at++; a &= a - 1; }
        while (b) { const int i = cf_ctz64(b) >> 3; if (at < m.posCap) m.pos[at] = (uint32_t)(o + 16 * k + 8 + (uint32_t)i); at++; b &= b - 1; }
    }
}

// ---- pass 3: one thread per record — the plain-form checks, its length, its seed, where its bases and its readID lie.
struct DTextRec {
    const uint8_t *text;
    uint64_t nBytes;
    const uint32_t *pos;         // FASTA: record starts; FASTQ: line ends (four per record)
    const uint64_t *total;       // the number of markers (the last of pass 1's exclusive sums): known on the device only
    uint64_t posCap;             // markers `pos` holds: a block with more is left to the host
    uint32_t recCap;             // records the per-record arrays hold
    uint32_t format;
    uint32_t seed0;              // (globalSeed + 101) * 59 * 61 * 67 * 71 * 73 * 79 * 83 (pat.h:60-68)
    uint32_t *rlen, *seeds;
    uint32_t *seqOff;            // first byte of the record's sequence line(s)
    uint32_t *idOff, *idLen;     // the readID: the name up to the first white space, a trailing /1 /2 /3 removed (aln_sink.h:2203-2217)
    TextStatus *st;
    // mates: two blocks in one buffer (text = this block's first byte, textBase = its place in the buffer: the places left for the
    // later passes count from the buffer's start), record r of block `mate` is read stride * r + mate of the batch
    uint32_t textBase, stride, mate;
};
CF_DEV bool tx_isspace(uint32_t c) { return c == ' ' || (c >= 9 && c <= 13); }
// base letter -> 0..3, 4 = N, 5 = not a plain base letter
CF_DEV uint32_t tx_code(uint32_t c) {
    const uint32_t u = c & 0xdfu;                          // upper case
    const uint32_t v = (u >> 1) & 3u;                      // A C T G -> 0 1 2 3
    if (u == 'A' || u == 'C' || u == 'G' || u == 'T') return v ^ (v >> 1);
    return u == 'N' ? 4u : 5u;
}
// Eight bytes at once (the sequence lines are nearly all of a block): their 2-bit codes gathered into 16 bits (an N and anything
// else as 0), which of them are N, which are no plain base letter at all, which are a '\n' — one bit per byte each.
CF_DEV uint32_t tx_movemask(uint64_t m80) { return (uint32_t)(((m80 >> 7) * 0x0102040810204080ull) >> 56); }
CF_DEV void tx_codes8(uint64_t x, uint32_t &c16, uint32_t &n8, uint32_t &bad8, uint32_t &nl8) {
    const uint64_t u = x & 0xdfdfdfdfdfdfdfdfull;                     // upper case
    const uint64_t mA = tx_match8(u, 'A'), mC = tx_match8(u, 'C'), mG = tx_match8(u, 'G'), mT = tx_match8(u, 'T'), mN = tx_match8(u, 'N');
    uint64_t c = ((mC | mT) >> 7) | ((mG | mT) >> 6);                 // A C G T -> 0 1 2 3 in the low bits of every byte
    c = (c | (c >> 6)) & 0x000f000f000f000full;
    c = (c | (c >> 12)) & 0x000000ff000000ffull;
    c = (c | (c >> 24)) & 0xffffull;
    c16 = (uint32_t)c;
    n8 = tx_movemask(mN);
    bad8 = tx_movemask(~(mA | mC | mG | mT | mN) & 0x8080808080808080ull);
    nl8 = tx_movemask(tx_match8(x, '\n'));
}
CF_DEV uint32_t tx_rotl32(uint32_t v, uint32_t s) { s &= 31u; return s ? (v << s) | (v >> (32u - s)) : v; }
// the seed's term of eight bases that start at base number i (genRandSeed, pat.h:69-75: r ^= code << 2(i mod 16), in 32-bit
// arithmetic): the 2-bit codes land in their fields cyclically; an N is code 4, whose bit lies one field up and falls off the top
CF_DEV uint32_t tx_seed8(uint32_t c16, uint32_t n8, uint32_t i) {
    uint32_t r = tx_rotl32(c16, 2u * (i & 15u));
    while (n8) { const uint32_t j = (uint32_t)cf_ctz32(n8); r ^= 4u << (((i + j) & 15u) << 1); n8 &= n8 - 1; }
    return r;
}
// 0x80 in every byte of x that is below 33 (no carries between bytes)
CF_DEV uint64_t tx_below33(uint64_t x) {
    const uint64_t t = (x & 0x7f7f7f7f7f7f7f7full) + 0x5f5f5f5f5f5f5f5full;       // >= 0x80 where the low seven bits are >= 33
    return ~(t | x) & 0x8080808080808080ull;
}
// the bases of a record from byte `pos` to byte `e` (line ends skipped): their number, their term of the seed, the plain-form check
CF_DEV uint32_t tx_bases(const uint8_t *text, uint64_t pos, uint64_t e, uint32_t &seed, uint32_t &len) {
    while (pos < e) {
        const uint64_t x = tx_load8(text, pos);
        if (e - pos >= 8) {
            uint32_t c16, n8, bad8, nl8;
            tx_codes8(x, c16, n8, bad8, nl8);
            if (nl8 == 0) {
                if (bad8) return kTxBadBase;
                seed ^= tx_seed8(c16, n8, len);
                len += 8; pos += 8;
                continue;
            }
        }
        const uint32_t ch = (uint32_t)x & 0xffu;
        pos++;
        if (ch == '\n') continue;
        const uint32_t code = tx_code(ch);
        if (code > 4) return kTxBadBase;
        seed ^= code << ((len & 15u) << 1);
        len++;
    }
    return 0;
}
// the name line from `from` to its '\n' (which must lie before `lim`): the name's term of the seed, the readID's length
CF_DEV uint32_t tx_name(TxCursor &c, uint64_t lim, uint32_t &r, uint32_t &nameLen, uint32_t &idLen) {
    uint32_t flags = 0, j = 0, ws = 0xffffffffu, p1 = 0, p2 = 0;
    bool slash = false;
    for (;;) {
        if (c.at >= lim) { flags |= kTxNoNameEnd; break; }
        const uint32_t ch = c.next();
        if (ch == '\n') break;
        if (ch == '\r') flags |= kTxCarriageReturn;
        if (ch == '/') slash = true;
        if (!slash) r ^= (uint32_t)(int32_t)(int8_t)ch << ((j & 3u) << 3);
        if (ws == 0xffffffffu && tx_isspace(ch)) ws = j;
        p2 = p1; p1 = ch;
        j++;
    }
    nameLen = j;
    uint32_t nl = j;
    if (j >= 2 && p2 == '/' && (p1 == '1' || p1 == '2' || p1 == '3')) nl -= 2;
    idLen = ws < nl ? ws : nl;
    if (j == 0) flags |= kTxEmptyName;
    return flags;
}
struct TextCounts { uint64_t nMarkers; uint32_t nRec; bool fits; };
CF_DEV TextCounts text_counts(const DTextRec &d) {
    TextCounts c;
    c.nMarkers = *d.total;
    const uint64_t rec = d.format == kTextFasta ? c.nMarkers : c.nMarkers >> 2;
    c.fits = c.nMarkers <= d.posCap && rec <= d.recCap;
    c.nRec = c.fits ? (uint32_t)rec : 0u;
    return c;
}
CF_DEV void text_record_body(const DTextRec &d, uint32_t r) {
    uint32_t flags = 0, len = 0;
    const TextCounts tc = text_counts(d);
    const bool live = r < tc.nRec;
    if (r == 0) {
        // the checks on the block as a whole
        if (!tc.fits) flags |= kTxTooMany;
        else if (d.nBytes) {
            if (d.text[0] != (d.format == kTextFasta ? '>' : '@')) flags |= kTxBadStart;
            if (d.format == kTextFastq && ((tc.nMarkers & 3u) || d.text[d.nBytes - 1] != '\n')) flags |= kTxLineCount;
            if (tc.nRec == 0) flags |= kTxBadStart;
        }
    }
    const uint64_t w = (uint64_t)d.stride * r + d.mate;                 // the read's number in the batch
    if (live) {
        uint32_t seed = d.seed0, nameLen = 0, idLen = 0;
        TxCursor c;
        if (d.format == kTextFasta) {
            const uint64_t s = d.pos[r], e = r + 1 < tc.nRec ? (uint64_t)d.pos[r + 1] : d.nBytes;
            c.seek(d.text, s + 1);
            flags |= tx_name(c, e, seed, nameLen, idLen);
            d.idOff[w] = d.textBase + (uint32_t)(s + 1); d.idLen[w] = idLen;
            d.seqOff[w] = d.textBase + (uint32_t)c.at;
            flags |= tx_bases(d.text, c.at, e, seed, len);
            // the qualities of a FASTA read are 'I' throughout: their term depends on the length only
            uint32_t q = ((len >> 2) & 1u) ? 0x49494949u : 0u;
            for (uint32_t j = 0; j < (len & 3u); j++) q ^= 0x49u << (j << 3);
            seed ^= q;
        } else {
            const uint64_t ls = r ? (uint64_t)d.pos[4 * (uint64_t)r - 1] + 1 : 0;
            const uint64_t n0 = d.pos[4 * (uint64_t)r], n1 = d.pos[4 * (uint64_t)r + 1], n2 = d.pos[4 * (uint64_t)r + 2], n3 = d.pos[4 * (uint64_t)r + 3];
            if (d.text[ls] != '@') flags |= kTxBadStart;
            c.seek(d.text, ls + 1);
            flags |= tx_name(c, n0 + 1, seed, nameLen, idLen);
            d.idOff[w] = d.textBase + (uint32_t)(ls + 1); d.idLen[w] = idLen;
            d.seqOff[w] = d.textBase + (uint32_t)(n0 + 1);
            flags |= tx_bases(d.text, n0 + 1, n1, seed, len);
            if (n2 <= n1 + 1 || d.text[n1 + 1] != '+') flags |= kTxBadPlus;
            if (n3 - n2 != n1 - n0) flags |= kTxQualLen;
            else {
                // the qualities' term (r ^= q[j] << 8(j mod 4)): the string's little-endian dwords folded together
                uint64_t pos = n2 + 1;
                uint32_t j = 0;
                while (pos < n3) {
                    uint64_t x = tx_load8(d.text, pos);
                    const uint32_t take = n3 - pos >= 8 ? 8u : (uint32_t)(n3 - pos);
                    if (take < 8) x = (x & ((1ull << (8 * take)) - 1)) | (0x2121212121212121ull << (8 * take));   // (the bytes past the line: no offence, and folded out again)
                    if (tx_below33(x)) flags |= kTxBadQual;
                    if (take < 8) x &= (1ull << (8 * take)) - 1;
                    seed ^= tx_rotl32((uint32_t)x ^ (uint32_t)(x >> 32), 8u * (j & 3u));
                    j += take; pos += take;
                }
            }
        }
        if (len == 0) flags |= kTxEmptySeq;
        d.rlen[w] = len; d.seeds[w] = seed;
    }
    // the block's sums: over the wavefront first, one set of atomics per wavefront
    unsigned long long words = live ? (len + 31u) >> 5 : 0u, bases = live ? len : 0u;
    uint32_t mx = live ? len : 0u;
    for (int m = CF_WAVE / 2; m > 0; m >>= 1) {
        words += cf_shfl_xor(words, m); bases += cf_shfl_xor(bases, m);
        const uint32_t o = cf_shfl_xor(mx, m); mx = o > mx ? o : mx;
        flags |= cf_shfl_xor(flags, m);
    }
    if (cf_lane() == 0) {
        const uint32_t stripe = (r / CF_WAVE) & (kTextStripes - 1);
        if (words) cf_atomic_add(&d.st->nWords[stripe], words);
        if (bases) cf_atomic_add(&d.st->nBases[stripe], bases);
        if (mx > d.st->maxLen) cf_atomic_max(&d.st->maxLen, mx);      // (the plain read only spares atomics that would change nothing)
        if (flags) cf_atomic_or(&d.st->flags, flags);
    }
}

// ---- pass 4 (behind the exclusive sums of the reads' word counts): the 2-bit words and the N masks, one thread per record.
struct DTextPack {
    const uint8_t *text;
    const uint32_t *seqOff, *rlen;
    const uint64_t *woff;
    uint64_t *bases;
    uint32_t *nmask;
    uint32_t nReads;
};
CF_DEV void text_pack_body(const DTextPack &d, uint32_t r) {
    if (r >= d.nReads) return;
    const uint32_t L = d.rlen[r];
    const uint64_t wo = d.woff[r];
    uint64_t pos = d.seqOff[r];
    uint32_t i = 0;
    for (uint32_t k = 0; 32 * k < L; k++) {
        uint64_t w = 0;
        uint32_t m = 0;
        uint32_t j = 0;
        while (j < 32 && i < L) {
            const uint64_t x = tx_load8(d.text, pos);
            if ((j & 7u) == 0 && L - i >= 8) {                        // eight bases at once, unless a line ends among them
                uint32_t c16, n8, bad8, nl8;
                tx_codes8(x, c16, n8, bad8, nl8);
                if (nl8 == 0) { w |= (uint64_t)c16 << (2 * j); m |= n8 << j; j += 8; i += 8; pos += 8; continue; }
            }
            const uint32_t ch = (uint32_t)x & 0xffu;
            pos++;
            if (ch == '\n') continue;                                // (a FASTA sequence over several lines)
            const uint32_t code = tx_code(ch);
            if (code > 3) m |= 1u << j; else w |= (uint64_t)code << (2 * j);
            j++; i++;
        }
        d.bases[wo + k] = w;
        d.nmask[wo + k] = m;
    }
}

// ---- egress: the default columns
//   readID seqID taxID score 2ndBestScore hitLength queryLength numMatches   (centrifuge.cpp:520; aln_sink.h:2279-2337)
// one thread per query: fmt_size_body leaves the bytes its rows take, fmt_write_body (behind the exclusive sums) writes them.
struct TextRow { uint32_t uniqueID, tidx, score, hitLen; };          // = NarrowRow
struct DTextFmt {
    const uint8_t *text;
    const uint32_t *idOff, *idLen, *rlen;
    const TextRow *rows;
    const uint64_t *rowFirst;
    const uint8_t *qinfo;        // rows of the query (six bits), "took part" bits of its mates
    const uint32_t *score2, *maxScore;
    uint32_t nQueries, paired;
    // the strings a row repeats: per reference its uid, per taxon its seqID when the row names no reference (the rank's name) and its taxID
    const uint8_t *strs;
    const uint32_t *uidOff, *rankOff, *taxOff;
    const uint8_t *taxLeaf;
    uint32_t nRefs, nTaxa, idxZero;
    uint32_t *size;              // per query
    const uint64_t *outOff;      // exclusive sums of size
    uint8_t *out;
    uint64_t outCap;
    // the tally that is not in the per-taxon counters (aln_sink.h:142-172): perfect single assignments per taxon, and the list of
    // perfect multi-assignment tuples (n, then n taxon indices)
    unsigned long long *single;
    uint32_t *tuples;
    uint32_t tuplesCap;
    TextStatus *st;
};
CF_DEV uint32_t tx_digits(uint32_t v) {
    return v < 10u ? 1u : v < 100u ? 2u : v < 1000u ? 3u : v < 10000u ? 4u : v < 100000u ? 5u : v < 1000000u ? 6u : v < 10000000u ? 7u : v < 100000000u ? 8u : v < 1000000000u ? 9u : 10u;
}
CF_DEV uint8_t *tx_put(uint8_t *w, uint32_t v) {
    const uint32_t n = tx_digits(v);
    for (uint32_t i = n; i-- > 0;) { w[i] = (uint8_t)('0' + v % 10u); v /= 10u; }
    return w + n;
}
CF_DEV uint8_t *tx_copy(uint8_t *w, const uint8_t *s, uint32_t n) {
    for (uint32_t i = 0; i < n; i++) w[i] = s[i];
    return w + n;
}
CF_DEV void fmt_size_body(const DTextFmt &f, uint32_t q) {
    if (q >= f.nQueries) return;
    const uint32_t ra = f.paired ? 2 * q : q;
    const uint32_t qlen = f.rlen[ra] + (f.paired ? f.rlen[ra + 1] : 0u);
    const uint32_t n = f.qinfo[q] & 0x3fu, id = f.idLen[ra], s2 = f.score2[q];
    // what every row of the query has: readID \t ... 2ndBestScore \t ... queryLength \t numMatches \n
    const uint32_t common = id + 1 + tx_digits(s2) + 1 + tx_digits(qlen) + 1 + tx_digits(n ? n : 1u) + 1;
    uint32_t total;
    if (n == 0) total = common + 12 + 1 + 1 + 1 + 1 + 1 + 1 + 1;       // unclassified \t 0 \t 0 \t [score2 \t] 0 \t
    else {
        total = n * common;
        const uint64_t f0 = f.rowFirst[q];
        for (uint32_t i = 0; i < n; i++) {
            const TextRow row = f.rows[f0 + i];
            const uint32_t t = row.tidx < f.nTaxa ? row.tidx : 0u;
            const bool viaUid = f.taxLeaf[t] && row.uniqueID < f.nRefs;
            total += (viaUid ? f.uidOff[row.uniqueID + 1] - f.uidOff[row.uniqueID] : f.rankOff[t + 1] - f.rankOff[t]) + 1 +
                     (f.taxOff[t + 1] - f.taxOff[t]) + 1 + tx_digits(row.score) + 1 + tx_digits(row.hitLen) + 1;
        }
    }
    f.size[q] = total;
}
// lds: kFmtLds + 8 bytes of LDS of the thread's WAVEFRONT (or nullptr).  A lane writes its rows a byte at a time, its neighbour's rows
// start ~41 bytes further on: straight to HBM that is one store instruction per byte and 64 scattered bytes per instruction.  The 64
// queries of a wavefront print one contiguous stretch of the text, so the rows are put together in LDS — at the stretch's own phase
// within a dword — and go out as whole aligned dwords, 256 bytes per store instruction (a stretch that does not fit goes the old way).
constexpr uint32_t kFmtLds = 8192;
#ifndef CF_HOST_EMU
// (the kernel is launched in blocks of 256 threads: cf_device.hip k_fmt_write)
CF_DEV uint8_t *fmt_wave_lds() { __shared__ __attribute__((aligned(16))) uint8_t lds[256 / CF_WAVE][kFmtLds + 16]; return lds[threadIdx.x / CF_WAVE]; }
#endif
CF_DEV void fmt_write_body(const DTextFmt &f, uint32_t q, uint8_t *lds = nullptr) {
#ifndef CF_HOST_EMU
    if (!lds) lds = fmt_wave_lds();
#endif
    const bool live = q < f.nQueries;
    uint32_t oneTaxon = 0xffffffffu;                                     // the taxon this query is a perfect single assignment of
    const uint32_t lane = cf_lane(), q0 = q - lane;
    const uint32_t qe = q0 + CF_WAVE < f.nQueries ? q0 + CF_WAVE : f.nQueries;
    const uint64_t segBegin = q0 < f.nQueries ? f.outOff[q0] : 0, segEnd = q0 < f.nQueries ? f.outOff[qe] : 0;
    const uint32_t phase = (uint32_t)(segBegin & 3u);
    const bool viaLds = lds != nullptr && segEnd > segBegin && segEnd - segBegin + phase <= kFmtLds && segEnd <= f.outCap;
    if (live) {
        const uint32_t ra = f.paired ? 2 * q : q;
        const uint32_t qlen = f.rlen[ra] + (f.paired ? f.rlen[ra + 1] : 0u);
        const uint32_t n = f.qinfo[q] & 0x3fu, idn = f.idLen[ra], s2 = f.score2[q], ms = f.maxScore[q];
        const uint8_t *id = f.text + f.idOff[ra];
        const uint64_t o = f.outOff[q];
        if (o + f.size[q] <= f.outCap) {
            uint8_t *w = viaLds ? lds + phase + (uint32_t)(o - segBegin) : f.out + o;
            if (n == 0) {
                w = tx_copy(w, id, idn);
                const uint8_t kU[] = {'\t', 'u', 'n', 'c', 'l', 'a', 's', 's', 'i', 'f', 'i', 'e', 'd', '\t', '0', '\t', '0', '\t'};
                for (uint32_t i = 0; i < sizeof kU; i++) *w++ = kU[i];
                w = tx_put(w, s2);
                *w++ = '\t'; *w++ = '0'; *w++ = '\t';
                w = tx_put(w, qlen); *w++ = '\t';
                *w++ = '1'; *w++ = '\n';
            } else {
                const uint64_t f0 = f.rowFirst[q];
                for (uint32_t i = 0; i < n; i++) {
                    const TextRow row = f.rows[f0 + i];
                    const uint32_t t = row.tidx < f.nTaxa ? row.tidx : 0u;
                    w = tx_copy(w, id, idn); *w++ = '\t';
                    if (f.taxLeaf[t] && row.uniqueID < f.nRefs) w = tx_copy(w, f.strs + f.uidOff[row.uniqueID], f.uidOff[row.uniqueID + 1] - f.uidOff[row.uniqueID]);
                    else w = tx_copy(w, f.strs + f.rankOff[t], f.rankOff[t + 1] - f.rankOff[t]);
                    *w++ = '\t';
                    w = tx_copy(w, f.strs + f.taxOff[t], f.taxOff[t + 1] - f.taxOff[t]); *w++ = '\t';
                    w = tx_put(w, row.score); *w++ = '\t';
                    w = tx_put(w, s2); *w++ = '\t';
                    w = tx_put(w, row.hitLen); *w++ = '\t';
                    w = tx_put(w, qlen); *w++ = '\t';
                    w = tx_put(w, n); *w++ = '\n';
                }
            }
        }
        // SpeciesMetrics::addSpeciesCounts (aln_sink.h:142-172), the part the per-taxon counters do not hold: only perfect hits feed the EM
        if (n == 0) oneTaxon = f.idxZero;
        else {
            const uint64_t f0 = f.rowFirst[q];
            if (n == 1) { const TextRow row = f.rows[f0]; if (ms != 0xffffffffu && row.score >= ms && row.tidx < f.nTaxa) oneTaxon = row.tidx; }
            else {
                bool all = ms != 0xffffffffu;
                for (uint32_t i = 0; i < n && all; i++) { const TextRow row = f.rows[f0 + i]; all = row.score >= ms && row.tidx < f.nTaxa; }
                if (all) {
                    const uint32_t at = cf_atomic_add(&f.st->tupleWords, n + 1);
                    if (at + n + 1 <= f.tuplesCap) { f.tuples[at] = n; for (uint32_t i = 0; i < n; i++) f.tuples[at + 1 + i] = f.rows[f0 + i].tidx; }
                }
            }
        }
    }
    if (viaLds) {
        // every lane's rows are in LDS (the operations of a wavefront on its LDS retire in order; the fence keeps the compiler's
        // hands off the order): out they go, dword j of the stretch by lane j mod 64
        cf_compiler_fence();
        const uint32_t total = phase + (uint32_t)(segEnd - segBegin);
        uint8_t *const dst = f.out + (segBegin - phase);                  // a dword boundary of the output
        for (uint32_t j = lane; 4 * j < total; j += CF_WAVE) {
            const uint32_t lo = 4 * j, hi = lo + 4;
            if (lo >= phase && hi <= total) {
                uint32_t v;
                __builtin_memcpy(&v, lds + lo, 4);
                *reinterpret_cast<uint32_t *>(dst + lo) = v;
            } else
                for (uint32_t k = lo < phase ? phase : lo; k < (hi < total ? hi : total); k++) dst[k] = lds[k];
        }
        cf_compiler_fence();
    }
    // one atomic per taxon and wavefront: the lanes that name the same taxon as the lowest waiting lane are counted by it
    bool waiting = oneTaxon != 0xffffffffu;
    for (;;) {
        const uint64_t w = cf_ballot(waiting);
        if (!w) break;
        const uint32_t lead = cf_shfl(oneTaxon, cf_ctz64(w));
        const uint64_t same = cf_ballot(waiting && oneTaxon == lead);
        if (waiting && oneTaxon == lead) {
            if ((uint32_t)cf_ctz64(same) == cf_lane()) cf_atomic_add(&f.single[lead], (unsigned long long)cf_popc64(same));
            waiting = false;
        }
    }
    if (live && q + 1 == f.nQueries) f.st->outBytes = f.outOff[f.nQueries];
}

}  // namespace cfamd
