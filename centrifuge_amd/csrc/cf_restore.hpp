// cf_restore.hpp — the joined text back out of the index: inverse BWT as many short LF walks.
//
// The reference inverts the BWT with ONE chain of n dependent LF steps from row n down to the
// '$' row (Ebwt::restore, bt2_util.h:150-168) — hours at the scale of the shipped indexes.  Here
// every K-th row is a *mark*.  Pass 1 starts one walk per mark (plus row n) and runs it until the
// next mark (or the '$' row), recording the number of steps and the mark it reached; the marks
// then form one linked list in text order, ranked with pointer doubling (restore_rank_body), which
// gives every segment its place in the text; pass 2 repeats the walks and writes the characters,
// 16 per 32-bit word, into the 2-bit packed text.  2n LF steps instead of n, but ~10^5 of them in
// flight at any time: the same HBM-random-read regime as the classification walk kernel, with
// the same chain layout (G lanes per chain, persistent waves pulling segments from a queue).
#pragma once
#include "cf_kernels.hpp"

namespace cfamd {

constexpr uint32_t kRestoreTerm = 0xffffffffu;       // "next segment" of the walk that reached the '$' row

struct DRestore {
    uint64_t n;              // text length; rows are 0..n
    uint32_t shift;          // marks are the rows with row % 2^shift == 0
    uint32_t nMarked;        // (n >> shift) + 1
    uint32_t nSeg;           // nMarked, +1 when row n is not a mark (its walk is segment nMarked)
    uint32_t *cursor;        // work queue
    uint64_t *segLen;        // pass 1 out: characters the segment emits
    uint32_t *segNext;       // pass 1 out: segment that continues the text leftwards, or kRestoreTerm
    const uint64_t *segEnd;  // pass 2 in: text position one past the segment's first character
    uint32_t *text;          // pass 2 out: 2-bit packed text, zeroed; char i at bits 2(i%16) of word i/16
    uint64_t maxSteps;       // a walk longer than this means the index is inconsistent
    uint32_t *err;           // set to 1 in that case
    // pass 2, optional: the suffix array and its inverse, sampled — every walk knows the text position of the row it is at
    uint64_t *saPos;         // value row >> posShift = SA[row] for rows that are multiples of 2^posShift (or null); 40-bit trios (trio_put), zeroed
    uint64_t *isa;           // isa[pos >> posShift] = the row of the suffix at pos, for such positions
    uint32_t posShift;
    uint32_t isaShift;       // ... the inverse sample at every 2^isaShift-th position (>= posShift: DIndex::isaRate)
};

CF_DEV uint64_t restore_start_row(const DRestore &r, uint32_t seg) { return seg < r.nMarked ? (uint64_t)seg << r.shift : r.n; }

// one walk per segment; WRITE = false: lengths and links, WRITE = true: characters
template <int G, bool WRITE>
CF_DEV void restore_body(const DIndex &ix, const DRestore &r) {
    const int sub = Grp<G>::sub();
    const uint32_t lane = cf_lane();
    const uint32_t leaderLane = lane & ~(uint32_t)(G - 1);
    const uint64_t markMask = (1ull << r.shift) - 1;
    bool busy = false;
    uint64_t row = 0, steps = 0, pos = 0;
    uint32_t item = 0, acc = 0;
    uint64_t wnext = 0, wend = 0;
    bool exhausted = false;
    const uint64_t total = r.nSeg;
    for (;;) {
        const uint64_t idleMask = cf_ballot(!busy && sub == 0);
        if (idleMask) {
            if (wnext >= wend && !exhausted) {
                uint32_t base = 0;
                if (lane == 0) base = cf_atomic_add(r.cursor, (uint32_t)kSearchChunk);
                base = cf_first_lane_u32(base);
                if (base >= total) { exhausted = true; wnext = wend = 0; }
                else { wnext = base; wend = base + kSearchChunk < total ? base + kSearchChunk : total; }
            }
            const uint64_t avail = wend - wnext;
            const uint32_t nIdle = (uint32_t)cf_popc64(idleMask);
            if (!busy) {
                const uint32_t rnk = (uint32_t)cf_popc64(idleMask & ((1ull << leaderLane) - 1));
                if (rnk < avail) {
                    item = (uint32_t)(wnext + rnk);
                    row = restore_start_row(r, item);
                    steps = 0; acc = 0;
                    if (WRITE) pos = r.segEnd[item];
                    if (row == ix.zOff) {
                        if (!WRITE && sub == 0) { r.segLen[item] = 0; r.segNext[item] = kRestoreTerm; }
                        if (WRITE && r.saPos && sub == 0) {           // a mark that is the '$' row: SA = 0, nothing to walk
                            const uint64_t pm = (1ull << r.posShift) - 1;
                            if ((row & pm) == 0) trio_put(r.saPos, row >> r.posShift, pos);
                            if ((pos & ((1ull << r.isaShift) - 1)) == 0) trio_put(r.isa, pos >> r.isaShift, row);
                        }
                    } else busy = true;
                }
            }
            wnext += nIdle < avail ? nIdle : avail;
        }
        if (cf_ballot(busy) == 0) {
            if (exhausted) break;
            continue;
        }
        // ---- loads: the side of the row and the row's own BWT byte (same 128-byte line)
        uint64_t sS = 0;
        uint32_t o = 0, own = 0;
        Side<G> sd;
        if (WRITE && busy && r.saPos && sub == 0) {           // SA[row] = pos at this point of the walk
            const uint64_t pm = (1ull << r.posShift) - 1;
            if ((row & pm) == 0) trio_put(r.saPos, row >> r.posShift, pos);
            if ((pos & ((1ull << r.isaShift) - 1)) == 0) trio_put(r.isa, pos >> r.isaShift, row);
        }
        if (busy) {
            sS = side_of(ix, row);
            o = (uint32_t)(row - sS * kSideChars);
            const uint8_t *p = ix.sides + sS * 128;
            side_load<G>(sd, p);
            own = p[o >> 2];
        }
        // ---- processing: c = bwt[row]; row = LF(row, c)  (bt2_idx.h:2941-2963)
        if (busy) {
            const int c = (int)((own >> (2 * (o & 3))) & 3u);
            const uint32_t pat = pat32(c);
            uint64_t t;
            if (G == 2) {
                const bool mine = sub == (c >> 1);
                const uint64_t oc = (c & 1) ? sd.v[8 / G - 1].y : sd.v[8 / G - 1].x;
                uint64_t pt = side_count1<G>(sd, pat, o) + (mine ? oc : 0ull);
                pt += swap1_64(pt);
                t = pt;
            } else t = side_occ<G>(sd, c) + Grp<G>::sum(side_count1<G>(sd, pat, o));
            if (c == 0 && sS == ix.zSide && ix.zIn < o) t--;
            row = t + fchr_of(ix, c);
            steps++;
            if (WRITE) {                                      // text[pos-1] = c, words filled from the top down
                pos--;
                acc |= (uint32_t)c << (2 * (uint32_t)(pos & 15));
                if ((pos & 15) == 0) { if (sub == 0 && acc) cf_atomic_or(&r.text[pos >> 4], acc); acc = 0; }
            }
            const bool atEnd = row == ix.zOff, atMark = (row & markMask) == 0;
            if (atEnd || atMark || steps > r.maxSteps) {
                if (sub == 0) {
                    if (steps > r.maxSteps) *r.err = 1;
                    if (WRITE && r.saPos && atEnd) {              // the '$' row: no walk starts there (SA = 0)
                        const uint64_t pm = (1ull << r.posShift) - 1;
                        if ((row & pm) == 0) trio_put(r.saPos, row >> r.posShift, pos);
                        if ((pos & ((1ull << r.isaShift) - 1)) == 0) trio_put(r.isa, pos >> r.isaShift, row);
                    }
                    if (WRITE) { if (acc) cf_atomic_or(&r.text[pos >> 4], acc); }
                    else { r.segLen[item] = steps; r.segNext[item] = atEnd ? kRestoreTerm : (uint32_t)(row >> r.shift); }
                }
                busy = false;
            }
        }
    }
}

// One round of pointer doubling over the list of segments (Wyllie's list ranking):
// sum[s] becomes the characters emitted from segment s to the end of the list.  Element nSeg is
// the terminator (sum 0, next = itself); kRestoreTerm links are redirected to it beforehand.
CF_DEV void restore_rank_body(const uint64_t *sumIn, const uint32_t *nextIn, uint64_t *sumOut, uint32_t *nextOut, uint32_t nElem, uint32_t s) {
    if (s >= nElem) return;
    const uint32_t nx = nextIn[s];
    sumOut[s] = sumIn[s] + sumIn[nx];
    nextOut[s] = nextIn[nx];
}

}  // namespace cfamd
