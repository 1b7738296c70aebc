// cf_kernels.hpp — the classification hot path as CDNA4 (gfx950) kernels.
//
// Integer pointer chasing over the FM index, bounded by the rate of random
// memory requests (no MFMA).  The kernels of a batch — none of them needs
// the host between them: the sizes one kernel produces for the next (work items,
// hit slots, rows) stay on the device in BatchStatus:
//
//   k_convert    (byte input only) 1 byte per base -> the packed form below.
//   k_plan       filters, hit capacities; two scans; work list + the 16-byte item records (DPlan::itemMeta).  Reads arrive
//                (or are made) PACKED: 2-bit words + N mask words, 32 bases per word, every read starting on a word.
//   k_pack       strand records (the reads as 2-bit words in search order + N masks) for the two-lane kernel; the one-lane
//                kernel makes its record itself, in LDS, from the packed read (S_REC / S_REC2).
//   k_search2    one chain per (read, strand), persistent waves on a chunked work queue: the chain of
//                partialSearch calls (hi_aligner.h:902-1031) driven as in
//                Classifier::searchForwardAndReverse (classifier.h:666-772), as a per-chain
//                state machine with ONE load phase per iteration.  Over the occurrence planes
//                (DIndex::planes) a chain is one lane — 64 chains per wavefront, one 16-byte
//                load per LF step, and over the pair planes (DIndex::planes2) one load per TWO bases;
//                over the sides two lanes (2 x 4 x global_load_dwordx4 per
//                step, rank through a per-lane LDS prefix table).  Calls start from the wide
//                ftab; unique matches are finished against the text (S_POS / S_TXT / S_ISA).
//   k_search     the same search with G lanes per chain and the read fetched from
//                its packed words: reads longer than 256 bp (no strand record).
//   k_post_fast  one lane per query, the common case in registers (<= 8 hits a strand, plan <= 4 hits): extend /
//                twin-removal / trim (classifier.h:790-895), strand choice (:898-941), the
//                libstdc++-exact sort (:267) as stable ranks, and the plan of which SA rows get
//                resolved (:253-299,366).  Everything per query is laid out by FIELD (DBatch::nhml, qflag, qplan ...):
//                these kernels are bound by (load instructions x lines touched), not bytes.
//   k_post       the same for the queries k_post_fast deferred (BatchStatus::nSlowPost), any size.
//   k_window     which queries' rows fit the row workspace in this pass (normally all).
//   k_walk3      one lane per SA row: walk left to a row of the resolve table
//                (group_walk.h:1154, bt2_idx.h:1980-2014, 2941-2963); k_walk2, 2-lane chains,
//                builds that table at load time and serves the debug tap.
//   k_score_fast one lane per query, <= 8 rows and <= 4 map entries in registers: hit map, (len-15)^2 scores, the
//                climb up the taxonomy (classifier.h:305-520), selection with the
//                per-read LCG (aln_sink.h:1860-1927); k_score the same for the deferred queries, any size.
//   k_count      the per-taxon counters (aln_sink.h:142-172) from the finished rows, binned in LDS.
//   k_compact    the rows of a batch packed back to back for the copy out.
//
// The bodies are written against cf_platform.hpp so tests/emu can single-step
// them on a CPU; the product only ever runs them through hipcc.
#pragma once
#include "cf_platform.hpp"

namespace cfamd {

constexpr uint32_t kNone32 = 0xffffffffu;
constexpr uint64_t kNone64 = ~0ull;
constexpr uint32_t kSideChars = 384;     // 96 bytes of 2-bit BWT per 128-byte side
constexpr int kSearchChunk = CF_WAVE;    // work items a wavefront claims per atomic: one per lane (search2_body keeps a claimed chunk's item records one per lane)

// ------------------------------------------------------------- device views
struct RefInfo { uint64_t tax; uint32_t tidx, pid; };
static_assert(sizeof(RefInfo) == 16, "RefInfo layout");
struct DIndex {
    const uint8_t *sides;        // numSides x 128 B: [96 B BWT][u64 occ A,C,G,T]
    const uint64_t *ftab, *eftab;
    uint64_t fchr0, fchr1, fchr2, fchr3;
    uint64_t len, zOff, zSide;
    uint32_t zIn;                // zOff % 384
    int32_t ftabChars, offRate, offw;
    const void *offs;            // u16 or u32 SA sample: reference-sequence index
    // Occurrence planes, made at load time (occ_planes_body): the BWT once more, as one bit vector per character.  Rows are
    // cut into groups of 64; group g has four 16-byte entries, entry c = {u64 bits: bit j set where bwt[64 g + j] == c (the
    // '$' row is in none), u64 base = fchr[c] + #{ i < 64 g : bwt[i] == c }}.  LF(row, c) = base + popcount of the bits
    // below row % 64: ONE 16-byte load per step and lane and a handful of instructions, where a 128-byte side takes eight
    // loads and popcounts over 24 dwords.  The search kernel's cost is counted in (load instruction x line touched) — the
    // rate at which a CU's L1 takes divergent requests, ~0.16 per cycle (tools/microbench/lane_loads.hip) — so one chain per
    // lane is affordable only with one load per step.  8 bits per base (the sides: 2.7); same LF values by construction.
    const uint8_t *planes;
    // Pair planes, made at load time from the planes (pair_planes_body): the same for TWO bases at once.  Group g has sixteen
    // 16-byte entries, entry 4 c1 + c0 = {u64 bits: bit j set where the suffix of row 64 g + j is preceded by c1 and that by c0,
    // u64 base = LF(c0, LF(c1, 64 g))}: LF(c0, LF(c1, row)) = base + popcount of the bits below row % 64 — the rows preceded by c1
    // map, in order, to consecutive rows, among which LF(c0, .) counts those preceded by c0.  One request extends a range by two
    // bases; 4 bytes per base.  Where ranges stay wide for long (strain clusters, repeats) that halves the range steps.
    const uint8_t *planes2;
    // Wide ftab, made at load time (wide_ftab_body): entry [fi] = what a partialSearch call knows after the wideChars bases fi
    // (10-mer ftab lookup + wideChars - ftabChars LF steps), in 8 bytes: the SA range at the DEEPEST depth D in
    // [ftabChars, wideChars] at which it is still non-empty — top (40 bits) | D - ftabChars (4 bits) | code (4 bits) | payload (16 bits).
    // code 1 .. 14 = bot - top, and — when D = wideChars — payload = the NEXT-PAIRS MASK: bit 4 c1 + c0 set iff the range survives
    // the two further bases c1, c0 (so bits 4 c1 .. 4 c1 + 3 all clear iff it does not survive c1 alone): a call whose next base
    // leads nowhere ends at the entry, without a step, and one whose base after next does skips the pair request (wide_ftab_body).
    // Round 5: an entry of up to FOUR rows (code 1 .. 4, D = wideChars: 86 % of the entries a matching read meets on a text of
    // ~2 x 4^wideChars bases) carries, instead of the mask, the rows' CONTEXT: the 8 (one row), 4 (two) or 2 (three, four) bases — row
    // r at bits 2 nb r + 2 i — that precede the suffixes in the text, nearest first — the bases the search would extend by.  Compared with the read's next
    // bases in registers they tell how far the call goes on (the longest of the rows' matches) without a request: the call's hit
    // length, and with it — for a hit nobody will read (lazy hits, search2_body) — everything the restart rule needs; a hit that is
    // kept still steps, but knows where to stop (no failing step).  An entry whose context would cross the start of the text goes
    // without one (code 15).
    // code 15: payload = bot - top (no mask), 0xffff = range too large for the entry, take the step-by-step path.  code 0 (the
    // whole entry 0): the 10-mer itself does not occur (ftab miss).  D = wideChars: the search goes on from there; D < wideChars:
    // the range died inside, and {range, D} is the hit the step-by-step path ends with.  One 8-byte read instead of the widest,
    // mostly two-sided, LF steps of the call — and, with the mask, instead of the one or two requests most calls of a strand
    // that matches nothing end with (its ranges are one or two rows wide at wideChars).
    const uint64_t *wide;
    int32_t wideChars;           // 0 = no wide table
    // Text verification of unique matches (search2_body, S_POS / S_TXT / S_ISA), all three made at load time by the inverse-BWT
    // walks (cf_restore.hpp): the joined text 2-bit packed (char i at bits 2(i%32) of word i/32), SA[row] for every
    // 2^posRate-th row and the row of the suffix at every 2^posRate-th position.  posRate < 0: not built.  The two samples hold
    // 40-bit values (texts < 2^40 bases) three to a 16-byte piece (trio_get / trio_put): one aligned 16-byte load per
    // lookup as before, 5.33 bytes per value instead of 8 — at the nt scale a third of the text tables' 130 GB.
    const uint64_t *text;
    const uint64_t *saPos;
    const uint64_t *isa;
    int32_t posRate;
    // ... the inverse sample at every 2^isaRate-th position, isaRate >= posRate (round 6).  With the hits of unique matches in their
    // position form (HitP) hardly anything reads the inverse sample any more — the resolver near the ends of a sequence, a match
    // that runs into the start of the text — so it is kept coarse (every 8th position where the SA sample holds every row) and its
    // room goes to the SA sample, which every verification reads; small ranges against the text (multiRows) keep both at every row.
    int32_t isaRate;
    // Text verification of SMALL RANGES (round 4, off unless multiRows > 0; needs posRate == 0: SA and inverse SA at every row /
    // position).  A range of R <= multiRows rows that has held its size for multiMinRun bases — relatives: strains of a cluster —
    // is finished like a single row: SA[top + i] and the text windows of every row i give how far that row's suffix goes on
    // matching (M_i); the call ends at the largest of them, with the rows that reach it — LF keeps rows in order, so they are the
    // rows ISA[p_first - Mmax] .. + S - 1, first = the lowest such row: R + windows + 1 requests instead of a pair-plane step per
    // two bases for as long as the relatives agree (31 of a repeat-rich read's 43.5 requests).  See search2_body, S_POS / S_TXT.
    uint32_t multiRows, multiMinRun;
    uint32_t verifyMinRun;       // successful single-row steps in a row before a unique match is handed to the text (default 0 since the next-pairs masks: the strand that matches nothing hardly ever gets to a one-row range, config 2 measured 6.32 -> 6.09 ms)
    // what the walk kernel resolves rows with: the file's own sample (walkOffs = offs, walkRate = offRate), or the dense
    // table made from it at load time (every 2^walkRate-th row, walkRate < offRate; see walk2_body)
    const void *walkOffs;
    int32_t walkRate;
    const uint64_t *boundRow;    // sorted .4.cf rows
    const uint32_t *boundRef;
    const uint32_t *boundBits;   // prefilter bitset over row >> boundShift
    uint64_t lastBoundary;
    uint32_t nBound;
    int32_t boundShift;
    // taxonomy tables
    // Position -> reference (round 6, hits in the position form: HitP above).  The walk-left of a row whose suffix starts at text
    // position p ends, k <= walkMax steps on, at a row of the file's sample, a boundary row or the '$' row, and answers with the
    // sequence that holds position p - k + 11 (the sample stores the sequence of "position + 11", bt2_idx.h:3640-3669; a boundary
    // row stands 11 bases before its sequence, :3504; the '$' row answers 0 = the first sequence).  So wherever p lies at least
    // walkMax bases behind the start of its sequence s and at least 12 before its end, the answer is s whatever k is — and
    // walkMax is known: exactly where the resolve table holds every row (it is made by taking that very walk from every row:
    // walk2_body in its table modes, DBatch::walkMaxOut), else as the longest segment of the inverse-BWT walks that make the text
    // tables (they run from one row that is a multiple of 2^shift, shift >= offRate, to the next: no walk-left is longer than the
    // segment it starts in).  Elsewhere (that many bases at the head of a sequence, twelve at its end) the resolver goes the old
    // way: the row from the inverse sample, its walk.  posFrag == nullptr: not made, no hit takes the form.
    //   posBucket[p >> posShift] = the fragment that holds position (p >> posShift) << posShift (fragments: the pieces of the
    //   sequences between their gaps, joined: Ebwt::_rstarts); posFrag[f] = {joined start, sequence}; posSeq[s] = {first, end} of
    //   sequence s in the joined text
    const uint32_t *posBucket;
    const u64x2 *posFrag, *posSeq;
    uint32_t posShift, nPosFrag, walkMax;
    const RefInfo *refInfo;      // per reference: taxid, dense taxon index of it, path index or kNone32 — one 16-byte gather
    const uint64_t *paths;       // nPath x 10 taxids
    const uint32_t *pathTidx;    // nPath x 10 dense taxon indices
    uint32_t nRef, tidxOne;      // dense index of taxid 1
    int32_t small;               // every row >> 7 fits 32 bits: side = row / 384 by a 32-bit multiply
};

struct DParams {
    uint32_t k, m, inc, ihits, rankSlot;
    int32_t traverse;
    const uint8_t *refExcluded;  // per reference flag or nullptr
    const uint64_t *hostSet;     // sorted taxids of the host closure
    uint32_t nHostSet;
};

struct Hit {                     // one partial hit (BWTHit hi_aligner.h:58-142) + the rows planned for it, as the kernels compute with it
    uint64_t top, bot;
    uint32_t bwoff, len;
    uint32_t nelt;               // rows to resolve for this hit (0 = skipped)
};

// The same hit as it is stored: 16 bytes (one global_store_dwordx4 per push, half the traffic of the k_post / k_emit /
// k_score reads).  w0 = top:40 | len:24,  w1 = size:40 | bwoff:24.
// top < 2^40 - 1 and size = bot - top < 2^40 (texts up to 1.1e12 bases); len, bwoff < 2^24 - 1 (0xffffff stands for the kNone32
// of a reset hit, hi_aligner.h:63-71): reads of up to 16,777,213 bases — contigs, long reads (rounds 1-3 kept 16-bit fields and
// refused reads of 65,535 bases or more, which the reference classifies).  An unresolved (dummy) hit carries top = bot = MASK in
// the reference: here top = all ones and size = 0.  The rows planned for a hit are NOT in the record: the common-case kernels
// keep them with the query (PlanHit), the general ones recompute them from the strand's maxG (QHead::maxG, plan_nelt).
struct HitP { uint64_t w0, w1; };
// Round 6, the POSITION FORM of a one-row hit: bit 39 of the size field set, the top field = the TEXT POSITION of the hit's suffix
// instead of its suffix-array row.  A unique match that the search finishes against the text (S_POS / S_TXT) ends at a text
// position; its row — which nothing reads but the resolver of rows to references — took one more random request (the inverse
// sample, S_ISA).  With DIndex::posFrag the resolver answers from the position instead (resolve_pos), and the search stores
// what it has.  Everything else sees a hit of one row (hp_size masks the bit); rows and positions travel together as 64-bit
// values with bit 63 marking a position (hp_row, PlanHit::top, DBatch::rowVal).  Texts of up to 2^39 bases.
constexpr uint64_t kHitPosForm = 1ull << 39, kRowIsPos = 1ull << 63;
// per (read, strand): hits pushed (31 bits) | one of them has minHitLen << 31 (k_post skips strands that cannot score)
CF_DEV uint32_t nhml_make(uint32_t nHits, bool hasLong) { return (nHits & 0x7fffffffu) | (hasLong ? 0x80000000u : 0u); }
CF_DEV uint32_t nhml_n(uint32_t v) { return v & 0x7fffffffu; }
CF_DEV bool nhml_long(uint32_t v) { return (v >> 31) != 0; }
static_assert(sizeof(HitP) == 16, "HitP layout");
constexpr uint64_t kHit40 = (1ull << 40) - 1;
constexpr uint32_t kHit24 = (1u << 24) - 1;
constexpr uint32_t kMaxReadLen = kHit24 - 2;                 // longest read the hit records hold

CF_DEV HitP hit_pack(const Hit &h) {
    const bool dummy = h.top == kNone64;
    const bool pos = !dummy && (h.top & kRowIsPos) != 0;           // (Hit::top carries the mark of hp_row)
    const uint64_t size = dummy ? 0 : pos ? (1ull | kHitPosForm) : h.bot - h.top;
    const uint64_t bw = h.bwoff == kNone32 ? (uint64_t)kHit24 : (uint64_t)(h.bwoff & kHit24);
    HitP p;
    p.w0 = (dummy ? kHit40 : (h.top & kHit40)) | ((uint64_t)(h.len & kHit24) << 40);
    p.w1 = (size & kHit40) | (bw << 40);
    return p;
}
CF_DEV uint32_t hp_len(const HitP &p) { return (uint32_t)(p.w0 >> 40); }
CF_DEV uint64_t hp_size(const HitP &p) { return p.w1 & (kHit40 & ~kHitPosForm); }
CF_DEV bool hp_posform(const HitP &p) { return (p.w1 & kHitPosForm) != 0; }
CF_DEV uint32_t hp_bwoff(const HitP &p) { const uint32_t b = (uint32_t)(p.w1 >> 40); return b == kHit24 ? kNone32 : b; }
CF_DEV uint64_t hp_top(const HitP &p) { return p.w0 & kHit40; }
CF_DEV uint64_t hp_row(const HitP &p) { return (p.w0 & kHit40) | (hp_posform(p) ? kRowIsPos : 0ull); }      // the first row — or the position, marked — as the resolver takes it
CF_DEV bool hp_dummy(const HitP &p) { return (p.w0 & kHit40) == kHit40 && (p.w1 & kHit40) == 0; }
CF_DEV void hp_set_len(HitP &p, uint32_t len) { p.w0 = (p.w0 & kHit40) | ((uint64_t)(len & kHit24) << 40); }
CF_DEV void hp_set_bwoff(HitP &p, uint32_t bw) { p.w1 = (p.w1 & kHit40) | ((uint64_t)(bw == kNone32 ? kHit24 : (bw & kHit24)) << 40); }
CF_DEV Hit hit_unpack(const HitP &p) {
    Hit h;
    const bool dummy = hp_dummy(p);
    h.top = dummy ? kNone64 : hp_row(p);
    h.bot = dummy ? kNone64 : hp_row(p) + hp_size(p);
    h.len = hp_len(p); h.bwoff = hp_bwoff(p); h.nelt = 0;
    return h;
}
// rows resolved for a hit of `len` bases over `size` rows under the strand's maxG (getGenomeIdx classifier.h:592-593, :299)
CF_DEV uint32_t plan_nelt(uint64_t len, uint64_t size, uint64_t maxG, uint64_t m, uint64_t ihits) {
    if (len <= m || size == 0) return 0;
    const uint64_t nelt = size < maxG ? size : maxG;
    return nelt > ihits ? 0u : (uint32_t)nelt;               // (those rows are never used)
}

// The per-query kernels are bound by the rate at which a CU's L1 takes (load instruction x line touched), like the search
// kernel: a 64-byte record per query read field by field costs 64 lines per instruction and wave, the same field
// in an array of its own 4.  So what the stages hand each other lies in arrays by field (qflag, qplan[j], o1tax ...), and
// only the rare general paths keep records.
//
// A planned hit — one whose rows are resolved and scored — as k_emit and the score kernels need it: kept per query when the
// query has at most kInlinePlan of them (nearly all do; pairs plan one or two per mate), so that neither goes back to the hit
// pool, where the hits their loops merely pass over (short ones, unresolved ones) would each cost a dependent global load —
// and so that the common-case post kernel (post_fast_body) need not write its sorted, planned hit list back at all.
struct PlanHit {
    uint64_t top;                // first row
    uint32_t nelt;               // rows planned for it
    uint32_t meta;               // len (16 bits) | mate index rdi << 16 | strand f << 17 | ts << 18 (the scoring loop's time stamp of the hit)
};
CF_DEV uint32_t plan_meta(uint32_t len, int rdi, int f, uint32_t ts) { return (len & 0xffffu) | ((uint32_t)rdi << 16) | ((uint32_t)f << 17) | (ts << 18); }
CF_DEV uint32_t pm_len(uint32_t m) { return m & 0xffffu; }
CF_DEV int pm_rdi(uint32_t m) { return (int)((m >> 16) & 1u); }
CF_DEV int pm_f(uint32_t m) { return (int)((m >> 17) & 1u); }
CF_DEV uint32_t pm_ts(uint32_t m) { return m >> 18; }
constexpr uint32_t kPlanTsMax = (1u << 14) - 1;
constexpr uint32_t kInlinePlan = 4;
constexpr uint32_t kPlanNotInline = 7;

// qflag[q]: planned hits in qplan (0 .. kInlinePlan, or kPlanNotInline: more than fit, read the hit lists) | paired << 3 |
// firstMate << 4 | nMates << 5.  The rows planned are qRows[q].
CF_DEV uint32_t qf_make(uint32_t nPlan, uint32_t paired, uint32_t firstMate, uint32_t nMates) { return nPlan | (paired << 3) | (firstMate << 4) | (nMates << 5); }
CF_DEV uint32_t qf_nplan(uint32_t f) { return f & 7u; }
CF_DEV bool qf_paired(uint32_t f) { return (f >> 3) & 1u; }
CF_DEV uint32_t qf_first(uint32_t f) { return (f >> 4) & 1u; }
CF_DEV int qf_nmates(uint32_t f) { return (int)((f >> 5) & 3u); }

struct QHead {                   // per query, written by k_post (the general kernel) for the queries whose plan is not inline
    uint32_t nProc[2][2];        // [mate][strand]: hits the scoring loop visits (break included)
    uint8_t lo[2], hi[2];        // strands chosen per mate
    uint8_t brk[2];              // bit f: the loop over strand f of that mate ended through `break`
    uint8_t pad2[2];
    uint32_t maxG[2][2];         // [mate][strand]: maxG while that strand's hits were planned (classifier.h:253-265), saturated — emit_body and
                                 // score_body get a hit's rows back from it (plan_nelt): nothing above ihits matters
};
static_assert(sizeof(QHead) == 40, "QHead layout");

struct HmEntry {                 // HitCount classifier.h:30-121, 72 bytes
    uint64_t taxID;
    uint32_t uniqueID, pid;      // reference index or kNone32; path index or kNone32
    uint32_t sc[2][2];
    uint32_t hl[2][2];
    uint32_t ts, score, hitLen, tidx;
    uint8_t rank, pad[7];
};

struct TcEntry { uint64_t tid; uint32_t cnt, tidx; };

struct OutRow { uint64_t taxID; uint32_t uniqueID, score, hitLen, tidx; };
struct NarrowRow { uint32_t uniqueID, tidx, score, hitLen; };
static_assert(sizeof(NarrowRow) == 16, "NarrowRow layout");
constexpr uint32_t kFieldRows = 4;

struct OpCounts { unsigned long long nFtab, nPair, nPair2, nSingle, nWalk, nRows, nFtabWide, nVerify, nTextLoads, nPosHits; };

// Device-side status of a batch: everything the host used to fetch in the middle of a batch (sizes of the
// work list, of the hit pool, of the row workspace) lives here, is produced and consumed by kernels, and
// travels to the host once, behind the last kernel.
struct BatchStatus {
    uint32_t nItems;             // (read, strand) work items = 2 x classified reads
    uint32_t flags;              // kStHitsOverflow | kStLenOverflow
    uint64_t hitsNeed;           // hit slots the batch's plan asks for
    uint64_t rowsTotal;          // SA rows planned by k_post over all queries
    uint64_t rowLo, rowHi;       // rows [rowLo, rowHi) are resolved and scored in the current pass ...
    uint32_t qLo, qHi;           // ... they belong to queries [qLo, qHi)
    uint64_t rowsOut;            // printed rows (total of nOut)
    uint64_t needRows;           // rows of query qLo when it alone exceeds the workspace (else 0)
    uint32_t nSlowPost;          // queries the common-case post kernel left to post_body (DBatch::slowPost)
    uint32_t nSlowScore;         // queries the common-case score kernel left to score_body in the current pass (DBatch::slowScore)
};
constexpr uint32_t kStHitsOverflow = 1u, kStLenOverflow = 2u, kStWordsOverflow = 4u;

struct DBatch {
    // reads, packed: read r owns the 32-base words [woff[r], woff[r+1]); base i sits at bits 2(i%32) of word i/32
    // (codes 0..3 = ACGT, an N carries code 0 and its bit in nmask)
    const uint64_t *bases;
    const uint32_t *nmask;
    const uint32_t *rlen;        // nReads
    const uint64_t *woff;        // nReads + 1
    const uint32_t *seeds;
    const uint8_t *pass;         // per read: takes part in classification
    const uint32_t *items;       // reads that are searched (nItems/2 entries)
    const uint32_t *slotOf;      // per read: index into items or kNone32
    const uint64_t *hitBase;     // per read: first Hit of the fw list; rc list at +hitCap
    const uint32_t *hitCap;      // per read: capacity of each strand list
    HitP *hits;
    uint32_t *nhml;              // per item (2*slot + strand): hits pushed | longest of them << 16 (k_post skips strands that cannot
                                 // score) — one word, one store per strand
    uint32_t *qflag;             // per query (qf_make)
    PlanHit *qplan;              // kInlinePlan arrays of qplanStride entries: planned hit j of query q = qplan[j * qplanStride + q]
    uint64_t qplanStride;
    QHead *qhead;                // per query; filled only where the plan is not inline
    uint32_t *qRows;             // per query (+1 slot), rows planned
    const uint64_t *qBase;       // exclusive scan of qRows
    uint64_t *rowVal;            // row workspace of the current pass: entry i = row rowLo + i
    uint32_t *rowRef;
    HmEntry *hm;
    TcEntry *tc;
    OutRow *out;                 // k slots per query: the rows of a query that prints MORE than kFieldRows
    // the rows of a query that prints up to kFieldRows (nearly every query: one, or — repeat-rich collections — a few), by FIELD:
    // row j of query q at [j * oStride + q] — taxID | uniqueID, score << 32 | hitLen, taxon index << 32.  (24-byte records in k
    // slots per query cost the score kernels, k_compact and k_count 64 lines per instruction and wave; a field 4.)
    uint64_t *o1tax, *o1a, *o1b;
    uint64_t oStride;
    uint32_t *nOut, *score2;
    unsigned long long *counts;  // 2 x nTaxa: n_reads then n_unique
    uint32_t nTaxa;
    uint32_t nReads, nQueries;
    int32_t paired;
    unsigned long long *cursor;  // [0] search queue, [1] walk queue
    BatchStatus *st;
    uint64_t hitsCap, rowsCap;   // capacities of the hit pool / of the row workspace (rows per pass)
    uint32_t genShift;           // walk2_body in its table-building modes: work item i stands for row i << genShift
    uint32_t lazyHits;           // search2_body: bit 0 = hits reach the hit pool only once their strand has one of minHitLen (see there); bit 2 = one-row
                                 // hits that end in the text go out in the position form (HitP; needs DIndex::posFrag)
    uint32_t *walkMaxOut;        // walk2_body in its table modes: the longest walk (atomic max), or nullptr
    uint32_t directRefs;         // the resolve table holds EVERY row (walkRate 0): the common-case score kernel takes a row's reference
                                 // straight from it — no k_emit, no k_walk3, no rowVal / rowRef traffic for the 99 % of the queries it
                                 // finishes; only the queries it leaves get their rows resolved into rowRef (resolve_query_body)
    OpCounts *ops;
    uint32_t *slowPost, *slowScore;   // nQueries each: the queries the common-case kernels hand to the general ones
    uint8_t *postDeferred;            // per query: 1 = the common-case post kernel left it to post_body (written by k_post_fast alone, so that a
                                      // score kernel running BESIDE post_body can tell without touching what post_body is writing); may be null
    // k_search2: one packed record per (read, strand) item, written by k_pack (see StrandRec below)
    const uint8_t *recs;
    // ... or, for the one-lane kernel, no records at all: itemMeta[item] = {word offset of the read, L, hit-list base of the strand,
    // read} (k_plan_fill), and the chain makes its strand record in LDS from the packed read itself (S_REC / S_REC2 of search2_body)
    const uint32_t *itemMeta;
    // Round 6: behind the packed reads — in the same array, DPlan::revDelta words on — the plan leaves the FORWARD strands' words
    // already in search order (rev_word; N-free reads only).  The forward item of such a read carries that word offset, both its
    // items the flag kItemPre: S_REC2 then loads the strand's words as they will lie in its record (no pointer of its own: the
    // kernel's scalar registers are all taken).
    uint32_t recWords;           // W: 2-bit words per strand (4: reads <= 128 bp, 6: <= 192 bp, 8: <= 256 bp); 0 = records not built
};

// ------------------------------------------------------------ group helpers
template <int G>
struct Grp {
    static CF_DEV int sub() { return G == 1 ? 0 : (int)(cf_lane() & (G - 1)); }
    static CF_DEV uint32_t sum(uint32_t v) {
#pragma unroll
        for (int m = 1; m < G; m <<= 1) v += cf_shfl_xor(v, m);
        return v;
    }
    static CF_DEV uint64_t bcast64(uint64_t v, int srcSub) {
        if (G == 1) return v;
        const int src = (int)(cf_lane() & ~(uint32_t)(G - 1)) | srcSub;
        const uint32_t lo = cf_shfl((uint32_t)v, src), hi = cf_shfl((uint32_t)(v >> 32), src);
        return ((uint64_t)hi << 32) | lo;
    }
};

CF_DEV uint64_t fchr_of(const DIndex &ix, int c) {
    return c == 0 ? ix.fchr0 : c == 1 ? ix.fchr1 : c == 2 ? ix.fchr2 : ix.fchr3;
}

// even bits set where the 2-bit char equals c (bt2_idx.h:505-517)
CF_DEV uint64_t match_mask(uint64_t w, int c) {
    const uint64_t t = ((c & 1) ? 0ull : 0x5555555555555555ull) | ((c & 2) ? 0ull : 0xaaaaaaaaaaaaaaaaull);
    const uint64_t x = w ^ t;
    return x & (x >> 1) & 0x5555555555555555ull;
}

// matches among the first n (0..64) chars of a 16-byte chunk (lo = chars 0..31)
CF_DEV uint32_t cnt_prefix(uint64_t mlo, uint64_t mhi, int n) {
    n = n < 0 ? 0 : (n > 64 ? 64 : n);
    const int nlo = n < 32 ? n : 32, nhi = n - nlo;
    const uint64_t klo = nlo >= 32 ? ~0ull : ((1ull << (2 * nlo)) - 1);
    const uint64_t khi = nhi >= 32 ? ~0ull : ((1ull << (2 * nhi)) - 1);
    return (uint32_t)(cf_popc64(mlo & klo) + cf_popc64(mhi & khi));
}

// One 128-byte side held cooperatively by a G-lane group: lane `sub` owns the
// 16-byte chunks sub, sub+G, ...  With G = 8 the group's loads form exactly one
// aligned 128-byte request.
template <int G>
struct Side {
    u64x2 v[8 / G];
};

template <int G>
CF_DEV void side_load(Side<G> &s, const uint8_t *p) {
    const int sub = Grp<G>::sub();
#pragma unroll
    for (int i = 0; i < 8 / G; i++) s.v[i] = cf_load16(p + 16 * (sub + i * G));
}

// this lane's share of #{ j < off : bwt[j] == c }
template <int G>
CF_DEV uint32_t side_count(const Side<G> &s, int c, uint32_t off) {
    const int sub = Grp<G>::sub();
    uint32_t acc = 0;
#pragma unroll
    for (int i = 0; i < 8 / G; i++) {
        const int j = sub + i * G;
        if (j < 6) acc += cnt_prefix(match_mask(s.v[i].x, c), match_mask(s.v[i].y, c), (int)off - 64 * j);
    }
    return acc;
}

// occ[c] of the side, broadcast to the whole group (chunks 6,7 hold A,C | G,T)
template <int G>
CF_DEV uint64_t side_occ(const Side<G> &s, int c) {
    constexpr int i6 = 6 / G, s6 = 6 % G, i7 = 7 / G, s7 = 7 % G;
    const uint64_t c6 = (c & 1) ? s.v[i6].y : s.v[i6].x;
    const uint64_t c7 = (c & 1) ? s.v[i7].y : s.v[i7].x;
    return Grp<G>::bcast64((c & 2) ? c7 : c6, (c & 2) ? s7 : s6);
}

// (LF(top,c), LF(bot,c)) — rank = fchr[c] + occ[side][c] + in-side count, minus
// one when c == A and the '$' (stored as an A) lies before the locus
// (bt2_idx.h:2192-2227).  A one-row range is served by the same formula:
// LF(top+1,c) - LF(top,c) is 1 exactly when bwt[top] == c and top != zOff,
// which is mapLF1's success condition (bt2_idx.h:2910-2934).
template <int G>
CF_DEV void rank_pair(const DIndex &ix, int c, uint64_t top, uint64_t bot, uint64_t &t, uint64_t &b, bool &twoSides) {
    const uint64_t sT = top / kSideChars;
    const uint32_t oT = (uint32_t)(top - sT * kSideChars);
    const uint64_t spread = bot - top;
    const bool same = (uint64_t)oT + spread <= kSideChars;
    twoSides = !same;
    Side<G> st;
    side_load<G>(st, ix.sides + sT * 128);
    uint32_t acc, oB;
    uint64_t occT, occB, sB;
    if (same) {
        sB = sT;
        oB = oT + (uint32_t)spread;
        acc = side_count<G>(st, c, oT) | (side_count<G>(st, c, oB) << 16);
        occT = occB = side_occ<G>(st, c);
    } else {
        sB = bot / kSideChars;
        oB = (uint32_t)(bot - sB * kSideChars);
        Side<G> sb;
        side_load<G>(sb, ix.sides + sB * 128);
        acc = side_count<G>(st, c, oT) | (side_count<G>(sb, c, oB) << 16);
        occT = side_occ<G>(st, c);
        occB = side_occ<G>(sb, c);
    }
    acc = Grp<G>::sum(acc);
    uint32_t cT = acc & 0xffffu, cB = acc >> 16;
    if (c == 0) {
        if (sT == ix.zSide && ix.zIn < oT) cT--;
        if (sB == ix.zSide && ix.zIn < oB) cB--;
    }
    const uint64_t f = fchr_of(ix, c);
    t = f + occT + cT;
    b = f + occB + cB;
}

// LF(row, bwt[row]) for the walk-left (bt2_idx.h:2941-2963); row != zOff
template <int G>
CF_DEV uint64_t lf_own(const DIndex &ix, uint64_t row) {
    const uint64_t s = row / kSideChars;
    const uint32_t o = (uint32_t)(row - s * kSideChars);
    const uint8_t *p = ix.sides + s * 128;
    Side<G> sd;
    side_load<G>(sd, p);
    const int c = (p[o >> 2] >> (2 * (o & 3))) & 3;     // same line as the side: no extra HBM fetch
    uint32_t cnt = Grp<G>::sum(side_count<G>(sd, c, o));
    if (c == 0 && s == ix.zSide && ix.zIn < o) cnt--;
    return fchr_of(ix, c) + side_occ<G>(sd, c) + cnt;
}

// How repeat-rich is the indexed collection?  (cf_index_open's planner asks before it chooses the tables: whether search ranges
// stay a few rows wide for long — relatives: strains of a cluster, shared operons — decides whether finishing small ranges
// against the text pays for the SA / inverse-SA samples at every row, DIndex::multiRows.)  One sample = two NEIGHBOURING rows
// row, row + 1: do their suffixes share the `depth` bases that PRECEDE them — the bases a backward search would extend a range
// holding both by?  Neighbouring rows with the same BWT character map to neighbouring rows, so one LF chain serves both.
// Over the file's sides (the only table there is when the planner runs).  An i.i.d. text answers "no" for all but a handful of
// samples, the repeat-rich stand-in "yes" for half of them.
CF_DEV bool repeat_probe_body(const DIndex &ix, uint64_t row, uint32_t depth) {
    for (uint32_t d = 0; d < depth; d++) {
        if (row + 1 > ix.len || row == ix.zOff || row + 1 == ix.zOff) return false;
        const uint64_t s0 = row / kSideChars, s1 = (row + 1) / kSideChars;
        const uint32_t o0 = (uint32_t)(row - s0 * kSideChars), o1 = (uint32_t)(row + 1 - s1 * kSideChars);
        const int c0 = (ix.sides[s0 * 128 + (o0 >> 2)] >> (2 * (o0 & 3))) & 3, c1 = (ix.sides[s1 * 128 + (o1 >> 2)] >> (2 * (o1 & 3))) & 3;
        if (c0 != c1) return false;
        row = lf_own<1>(ix, row);
    }
    return true;
}

// bits of v below position o (0..64)
CF_DEV uint32_t popc_below(uint64_t v, uint32_t o) {
    const uint32_t lo = (uint32_t)v, hi = (uint32_t)(v >> 32);
    const uint32_t klo = o >= 32 ? 0xffffffffu : ((1u << o) - 1u);
    const uint32_t khi = o >= 64 ? 0xffffffffu : (o > 32 ? ((1u << (o - 32)) - 1u) : 0u);
    return (uint32_t)cf_popc32(lo & klo) + (uint32_t)cf_popc32(hi & khi);
}

// 40-bit values three to a 16-byte piece: value i lives in piece i / 3 at bits 40 (i % 3) .. + 39.  Written once each into a
// zeroed table (atomic OR: three writers share a piece), read with one aligned 16-byte load.
CF_DEV uint64_t trio_get(const u64x2 &v, uint32_t k) {
    const uint64_t m = (1ull << 40) - 1;
    return k == 0 ? (v.x & m) : k == 1 ? (((v.x >> 40) | (v.y << 24)) & m) : ((v.y >> 16) & m);
}
CF_DEV uint64_t trio_at(const uint64_t *tab, uint64_t i) {
    const uint64_t pc = i / 3;
    return trio_get(cf_load16(reinterpret_cast<const uint8_t *>(tab + 2 * pc)), (uint32_t)(i - 3 * pc));
}
CF_DEV void trio_put(uint64_t *tab, uint64_t i, uint64_t val) {
    const uint64_t pc = i / 3;
    const uint32_t k = (uint32_t)(i - 3 * pc);
    val &= (1ull << 40) - 1;
    uint64_t *w = tab + 2 * pc;
    if (k == 0) { if (val) cf_atomic_or64(w, val); }
    else if (k == 1) { if (val << 40) cf_atomic_or64(w, val << 40); if (val >> 24) cf_atomic_or64(w + 1, val >> 24); }
    else if (val) cf_atomic_or64(w + 1, val << 16);
}
constexpr uint64_t trio_words(uint64_t count) { return 2 * ((count + 2) / 3) + 2; }      // u64 words of a table of `count` values (+ one piece)

// rank_pair over whatever the index holds: the occurrence planes when they were made (one 16-byte entry per row group — every
// lane of a G-lane chain reads the same entry, one lookup), else the sides.  With the planes an index can do WITHOUT its sides
// in HBM (DIndex::sides == nullptr: cf_index_open drops them when that buys a denser table elsewhere — the nt-scale index).
template <int G>
CF_DEV void rank_any(const DIndex &ix, int c, uint64_t top, uint64_t bot, uint64_t &t, uint64_t &b, bool &twoSides) {
    if (ix.planes) {
        const u64x2 et = cf_load16(ix.planes + (top >> 6) * 64 + 16 * c);
        t = et.y + popc_below(et.x, (uint32_t)top & 63u);
        twoSides = (bot >> 6) != (top >> 6);
        if (!twoSides) b = et.y + popc_below(et.x, (uint32_t)bot & 63u);
        else { const u64x2 eb = cf_load16(ix.planes + (bot >> 6) * 64 + 16 * c); b = eb.y + popc_below(eb.x, (uint32_t)bot & 63u); }
    } else rank_pair<G>(ix, c, top, bot, t, b, twoSides);
}

// the character that precedes the suffix of `row` (its BWT character) and the row of the suffix that starts with it: one LF step
// with the row's own character (bt2_idx.h:2941-2963) over the planes — the four entries of the row's group are one 64-byte line,
// the character is the one whose bit is set at the row — or the sides.  false: `row` is the '$' row (nothing precedes the text)
CF_DEV bool lf_own_any(const DIndex &ix, uint64_t &row, int &c) {
    if (row == ix.zOff) return false;
    if (ix.planes) {
        const uint8_t *p = ix.planes + (row >> 6) * 64;
        const uint32_t o = (uint32_t)row & 63u;
        const u64x2 e0 = cf_load16(p), e1 = cf_load16(p + 16), e2 = cf_load16(p + 32), e3 = cf_load16(p + 48);
        c = ((e0.x >> o) & 1) ? 0 : ((e1.x >> o) & 1) ? 1 : ((e2.x >> o) & 1) ? 2 : 3;
        const u64x2 e = c == 0 ? e0 : c == 1 ? e1 : c == 2 ? e2 : e3;
        row = e.y + popc_below(e.x, o);
    } else {
        const uint64_t s = row / kSideChars;
        const uint32_t o = (uint32_t)(row - s * kSideChars);
        c = (ix.sides[s * 128 + (o >> 2)] >> (2 * (o & 3))) & 3;
        row = lf_own<1>(ix, row);
    }
    return true;
}

CF_DEV uint64_t ftab_hi(const DIndex &ix, uint64_t i) {     // bt2_idx.h:1880-1897
    const uint64_t v = ix.ftab[i];
    return v <= ix.len ? v : ix.eftab[(v ^ kNone64) * 2 + 1];
}
CF_DEV uint64_t ftab_lo(const DIndex &ix, uint64_t i) {     // bt2_idx.h:1953-1970
    const uint64_t v = ix.ftab[i];
    return v <= ix.len ? v : ix.eftab[(v ^ kNone64) * 2];
}

// ------------------------------------------------------------ read access
// A strand walks its read one base per LF step, so a register window of one packed word (32 bases + their
// N bits) serves 32 steps per load.  Char j of the forward strand is base j; of the reverse-complement
// strand base L-1-j, complemented (sstring.h:2928-2934: N stays N).
struct ReadWin {
    uint64_t w, idx;
    uint32_t m;
};

CF_DEV int strand_char(const DBatch &b, uint64_t wbase, uint32_t L, bool fw, uint32_t j, ReadWin &win) {
    const uint32_t pos = fw ? j : L - 1 - j;
    const uint64_t wi = wbase + (pos >> 5);
    if (wi != win.idx) { win.w = b.bases[wi]; win.m = b.nmask[wi]; win.idx = wi; }
    const uint32_t sh = pos & 31;
    if ((win.m >> sh) & 1u) return 4;
    const int c = (int)((win.w >> (2 * sh)) & 3);
    return fw ? c : (c ^ 3);
}

// ------------------------------------------------------- byte-stream helpers
// 8 bytes of a byte array from any offset, as two aligned 8-byte loads (arrays carry >= 16 bytes of padding)
CF_DEV uint64_t load8_any(const uint8_t *base, uint64_t off) {
    const uint64_t a = off & ~7ull;
    const uint32_t sh = (uint32_t)(off & 7) * 8;
    const uint64_t lo = cf_load8(base + a);
    if (sh == 0) return lo;
    const uint64_t hi = cf_load8(base + a + 8);
    return (lo >> sh) | (hi << (64 - sh));
}
// bit 0 of every byte set where the byte is non-zero
CF_DEV uint64_t nonzero_bytes(uint64_t t) { t |= t >> 4; t |= t >> 2; t |= t >> 1; return t & 0x0101010101010101ull; }
CF_DEV uint64_t bswap64(uint64_t v) {
    v = ((v & 0x00ff00ff00ff00ffull) << 8) | ((v >> 8) & 0x00ff00ff00ff00ffull);
    v = ((v & 0x0000ffff0000ffffull) << 16) | ((v >> 16) & 0x0000ffff0000ffffull);
    return (v << 32) | (v >> 32);
}

// ------------------------------------------------------- packing the reads
// 8 base codes (one per byte: 0..3 = ACGT, anything above = N) -> 16 bits of 2-bit codes (N -> 0) + 8 N bits
CF_DEV void squeeze8(uint64_t x, uint32_t &c16, uint32_t &n8) {
    const uint64_t isN = nonzero_bytes(x & 0xfcfcfcfcfcfcfcfcull);
    uint64_t c = x & 0x0303030303030303ull & ~(isN * 3);
    c = (c | (c >> 6)) & 0x000f000f000f000full;           // 2 bits of every byte -> 16 contiguous bits
    c = (c | (c >> 12)) & 0x000000ff000000ffull;
    c = (c | (c >> 24)) & 0xffffull;
    c16 = (uint32_t)c;
    n8 = (uint32_t)((isN * 0x0102040810204080ull) >> 56);
}
// every bit of m doubled: bit i -> bits 2i, 2i+1
CF_DEV uint64_t spread_pairs(uint32_t m) {
    uint64_t x = m;
    x = (x | (x << 16)) & 0x0000ffff0000ffffull;
    x = (x | (x << 8)) & 0x00ff00ff00ff00ffull;
    x = (x | (x << 4)) & 0x0f0f0f0f0f0f0f0full;
    x = (x | (x << 2)) & 0x3333333333333333ull;
    x = (x | (x << 1)) & 0x5555555555555555ull;
    return x * 3;
}
// the 32 bit pairs of a word in reverse order
CF_DEV uint64_t pair_reverse(uint64_t x) {
    x = cf_brev64(x);
    return ((x >> 1) & 0x5555555555555555ull) | ((x & 0x5555555555555555ull) << 1);
}

// The 1-byte-per-base input of cf_batch_create into the packed form every kernel works on.  One thread per
// read (rlen and woff are made before: woff = exclusive sum of ceil(len / 32)).
struct DConvert {
    const uint8_t *seq;          // base codes, padded by >= 16 bytes
    const uint64_t *off;         // nReads + 1
    const uint64_t *woff;
    uint64_t *bases;
    uint32_t *nmask;
    uint32_t nReads;
};
CF_DEV void convert_body(const DConvert &c, uint32_t r) {
    if (r >= c.nReads) return;
    const uint64_t o = c.off[r], L = c.off[r + 1] - o, wo = c.woff[r];
    for (uint64_t k = 0; 32 * k < L; k++) {
        uint64_t w = 0;
        uint32_t m = 0;
#pragma unroll
        for (uint32_t g = 0; g < 4; g++) {
            const uint64_t j0 = 32 * k + 8 * g;
            if (j0 >= L) break;
            uint64_t x = load8_any(c.seq, o + j0);
            if (L - j0 < 8) x &= (1ull << (8 * (L - j0))) - 1;
            uint32_t c16, n8;
            squeeze8(x, c16, n8);
            w |= (uint64_t)c16 << (16 * g);
            m |= n8 << (8 * g);
        }
        c.bases[wo + k] = w;
        c.nmask[wo + k] = m;
    }
}
// The DENSE packed input (cf_dense_reads: reads of ONE length, four bases per byte, every read starting on a byte — 25 bytes per
// 100-base read across the host link instead of the 32 + 4 of the word form and its length) into the word form every kernel works
// on.  One thread per (read, 32-base word); thread (r, 0) also writes the read's length.  `dense` is padded by >= 16 bytes.
struct DUnpack {
    const uint8_t *dense;
    uint64_t *bases;
    uint32_t *rlen;
    uint32_t nReads, readLen;
    uint64_t *rev;               // or nullptr: where the forward strands' words in search order go (rev_word; DPlan::revDelta words behind `bases`)
};
CF_DEV void dense_unpack_body(const DUnpack &u, uint64_t t) {
    const uint32_t W = (u.readLen + 31) >> 5, bpr = (u.readLen + 3) >> 2;
    if (W == 0) { if (t < u.nReads) u.rlen[t] = 0; return; }
    const uint64_t r = t / W;
    const uint32_t k = (uint32_t)(t - r * W);
    if (r >= u.nReads) return;
    uint64_t w = load8_any(u.dense, r * bpr + 8ull * k);
    const uint32_t have = u.readLen - 32 * k;                        // bases of this word that exist (>= 1)
    if (have < 32) w &= (1ull << (2 * have)) - 1;                    // (the next read's bytes, and the last byte's unused bit pairs)
    u.bases[r * W + k] = w;
    if (k == 0) u.rlen[r] = u.readLen;
    if (u.rev) {
        // rev_word(k) straight from the dense bytes (the thread has the read's place at hand: a pass of its own over the words cost
        // 0.3 - 0.4 ms per 10 M reads, this a store): the 32-base window that ends at base L-1-32k, its pairs reversed
        const uint32_t L = u.readLen;
        const int32_t s0 = (int32_t)(L - 1 - 32u * k) - 31;
        uint64_t x;
        if (s0 >= 0) {
            const uint64_t at = r * bpr + ((uint32_t)s0 >> 2);
            const uint32_t sh = 2u * ((uint32_t)s0 & 3u);
            x = load8_any(u.dense, at) >> sh;
            if (sh) x |= (load8_any(u.dense, at + 8) & 0xffull) << (64 - sh);
        } else x = load8_any(u.dense, r * bpr) << (2 * (uint32_t)(-s0));
        uint64_t rv = pair_reverse(x);
        if (have < 32) rv &= (1ull << (2 * have)) - 1;
        u.rev[r * W + k] = rv;
    }
}

// read lengths of the byte input (the packed input brings them along)
CF_DEV void rlen_body(const uint64_t *off, uint32_t *rlen, uint32_t nReads, uint32_t r) {
    if (r >= nReads) return;
    const uint64_t l = off[r + 1] - off[r];
    rlen[r] = l > 0xffffffffull ? 0xffffffffu : (uint32_t)l;
}

// -------------------------------------------------------------- batch plan
// The per-batch work plan, made on the device from the packed reads (the host never walks the bases):
// which reads are classified (Scoring::nFilter scoring.cpp:104-117 with nCeil = 0.15 len, scoring.h:61-63,
// and the length filter of centrifuge.cpp:2562-2577), how many hit slots a strand can need, then — after
// one exclusive scan of the capacities — the work list and the hit-list bases.  One thread per read; a read's N count is
// the popcount of its mask words, and neighbouring lanes read neighbouring words.
struct DPlan {
    const uint32_t *nmask;
    const uint32_t *rlen;
    const uint64_t *woff;
    uint32_t nReads;
    int32_t ftabChars;
    uint32_t maxLenAllowed; // the launch was specialised for reads up to this long
    uint64_t nWords;        // packed words the caller uploaded: a read whose words would lie beyond them is a caller error
    uint8_t *pass;          // [nReads]
    uint32_t *hitCap;       // [nReads]      hits a strand's list can hold; 0 = the read is not classified.  The scan input:
                            //               slot = #non-zero before, hit-list base = 2 x sum before (cf_scan.hpp SCAN_HITS)
    uint32_t *slotOf;       // [nReads + 1]  exclusive count of classified reads, then slot or kNone32 in place
    uint64_t *hitBase;      // [nReads + 1]  exclusive sum of 2 * hitCap
    uint32_t *items;        // [#classified]
    uint32_t *itemMeta;     // [2 x #classified x 4] or nullptr (DBatch::itemMeta)
    uint64_t *bases;        // the packed reads; and, revDelta words on in the same array (0: not made) ...
    uint32_t revDelta;      // ... the forward strands' words in search order (rev_word; see DBatch::itemMeta)
    BatchStatus *st;
    uint64_t hitsCap;       // hit slots the pool holds
};

CF_DEV void plan_body(const DPlan &p, uint32_t r) {
    if (r < p.nReads) {
        const uint64_t L = p.rlen[r], wo = p.woff[r];
        uint32_t nN = 0;
        // the lengths promise more words than were uploaded (n_words too small): flagged before anything reads past the
        // buffers, and the read is kept away from every kernel (cf_batch_wait reports CF_ERR_ARG)
        const bool inside = wo + ((L + 31) >> 5) <= p.nWords;
        if (!inside) cf_atomic_or(&p.st->flags, kStWordsOverflow);
        for (uint64_t k = 0; inside && 32 * k < L; k++) {
            uint32_t m = p.nmask[wo + k];
            if (L - 32 * k < 32) m &= (1u << (L - 32 * k)) - 1u;                // bits past the read do not count
            nN += (uint32_t)cf_popc32(m);
        }
        const uint64_t maxns = (uint64_t)(0.0 + (double)0.15f * (double)L);
        bool ok = inside && L >= 2 && nN <= maxns;
        // a read longer than the launch was specialised for is a caller error: flagged, and kept away from the kernels
        if (ok && L > p.maxLenAllowed) { cf_atomic_or(&p.st->flags, kStLenOverflow); ok = false; }
        // Every partialSearch call either swallows >= ftabChars N-free bases or ends on an N
        // (hi_aligner.h:934-978), which bounds the hits per strand.
        const uint32_t cap = ok ? (uint32_t)(nN + (L - nN) / (uint64_t)p.ftabChars + 2) : 0u;
        p.pass[r] = ok ? 1 : 0;
        p.hitCap[r] = cap;
    }
}

// Word k of a read's FORWARD strand in search order (char j of a strand record = the j-th base from the strand's right end, so for
// the forward strand base L-1-j): the 32-base window of the read that ends at base L-1-32k, its pairs reversed.  rw = the read's
// packed words.  The reverse-complement strand needs no such word: its record is the read's own words complemented.  Made once
// per read by the plan (a streaming pass: k_plan_fill) so that the search kernel's S_REC2 state — whose code every wavefront
// carries through nearly every iteration for the few lanes that are in it — is two loads and a complement instead of W funnel
// shifts and bit reversals (round 6; round 5 measured what the in-loop transform costs: search 5.43 -> 5.05 ms without it).
CF_DEV uint64_t rev_word(const uint64_t *rw, uint32_t L, uint32_t k) {
    const uint32_t have = L - 32u * k;                                    // chars of this word that exist (>= 1)
    const int32_t s0 = (int32_t)(L - 1 - 32u * k) - 31;                   // first base of the window
    uint64_t x;
    if (s0 >= 0) {
        const uint32_t wi = (uint32_t)s0 >> 5, sh = (uint32_t)s0 & 31;
        x = rw[wi] >> (2 * sh);
        if (sh) x |= rw[wi + 1] << (64 - 2 * sh);
    } else x = rw[0] << (2 * (uint32_t)(-s0));                            // the window starts before the read: zeros below base 0
    uint64_t w = pair_reverse(x);
    if (have < 32) w &= (1ull << (2 * have)) - 1;
    return w;
}

constexpr uint32_t kItemHasN = 0x80000000u;              // itemMeta word 1 = L | this (reads are shorter than 65535 bases)
constexpr uint32_t kItemPre = 0x40000000u;               // ... | this: an N-free read whose forward words the plan has made (DPlan::revDelta)
CF_DEV void plan_fill_body(const DPlan &p, uint32_t r) {
    if (r > p.nReads) return;
    if (r == p.nReads) {                                  // the scans' totals: sizes of the work list and of the hit pool
        const uint64_t need = p.hitBase[r];
        p.st->hitsNeed = need;
        if (need > p.hitsCap) { cf_atomic_or(&p.st->flags, kStHitsOverflow); p.st->nItems = 0; }
        else p.st->nItems = 2u * p.slotOf[r];
        return;
    }
    const uint32_t slot = p.slotOf[r];
    if (p.pass[r]) {
        p.items[slot] = r;
        if (p.itemMeta) {                                 // the two strands' work items
            const uint32_t wo = (uint32_t)p.woff[r], L = p.rlen[r], hb = (uint32_t)p.hitBase[r];
            // does the read hold an N at all?  (Hardly any does: the search kernel then leaves the mask words where they are.)
            uint32_t any = 0;
            for (uint32_t k = 0; 32 * k < L; k++) {
                uint32_t mk = p.nmask[p.woff[r] + k];
                if (L - 32 * k < 32) mk &= (1u << (L - 32 * k)) - 1u;
                any |= mk;
            }
            const bool pre = p.revDelta && !any;          // (a read with an N keeps the in-kernel transform: its mask words turn with it)
            const uint32_t Lf = L | (any ? kItemHasN : 0u) | (pre ? kItemPre : 0u);
            uint32_t *m = p.itemMeta + 8 * (size_t)slot;
            m[0] = pre ? wo + p.revDelta : wo; m[1] = Lf; m[2] = hb; m[3] = r;
            m[4] = wo; m[5] = Lf; m[6] = hb + p.hitCap[r]; m[7] = r;
        }
    } else p.slotOf[r] = kNone32;
}

// ... and the words themselves: thread t = word t % wordsPerRead of read t / wordsPerRead (wordsPerRead = the batch's record
// width, 4 / 6 / 8: every read of the batch is at most that long).  A kernel of its own behind plan_fill_body — one thread per
// read writing its 4 - 8 words cost 0.3 - 0.5 ms per 10 M reads (strided 8-byte stores).  Only for batches that did not come in
// the dense form: dense_unpack_body makes these words as it unpacks
CF_DEV void rev_words_body(const DPlan &p, uint32_t wordsPerRead, uint64_t t) {
    const uint64_t r = t / wordsPerRead;
    const uint32_t k = (uint32_t)(t - r * wordsPerRead);
    if (r >= p.nReads || !p.revDelta || !p.pass[r]) return;
    const uint32_t L = p.rlen[r];
    if (32u * k >= L) return;                         // (made for a read with an N as well — its items do not point here: cheaper than asking)
    const uint64_t wo = p.woff[r];
    p.bases[wo + p.revDelta + k] = rev_word(p.bases + wo, L, k);
}

constexpr uint32_t kMaxScoreNever = 0xffffffffu;
// max_score of a query: sum over the mates that take part of (len-15)^2 (classifier.h:530-536)
CF_DEV void plan_maxscore_body(const uint32_t *rlen, const uint8_t *pass, uint32_t nQueries, int paired, uint32_t *maxScore, uint32_t q) {
    if (q >= nQueries) return;
    const uint32_t r0 = paired ? 2 * q : q;
    // (int64_t in the reference, compared with 32-bit scores: a value of 2^32 or more — a read beyond 65,550 bases — is one no
    // score reaches, and is handed on as kMaxScoreNever, which is no sum of two squares: cf_report_add knows it)
    const uint64_t L0 = rlen[r0];
    const uint64_t s0 = L0 > 15 ? (L0 - 15) * (L0 - 15) : 0u;
    const bool p0 = pass[r0] != 0;
    uint64_t v = p0 ? s0 : 0u;
    if (paired) {
        const uint64_t L1 = rlen[r0 + 1];
        const uint64_t s1 = L1 > 15 ? (L1 - 15) * (L1 - 15) : 0u;
        const bool p1 = pass[r0 + 1] != 0;
        v = (p0 && p1) ? s0 + s1 : p0 ? s0 : p1 ? s1 : 0u;
    }
    maxScore[q] = v >= 0xffffffffull ? kMaxScoreNever : (uint32_t)v;
}

// result egress: the rows of query q moved to their place in the dense list — from the by-field arrays when it prints up to
// kFieldRows rows, from its k slots when more.  One thread per query (+ one for the total).
struct DCompact {
    const OutRow *out;
    const uint64_t *o1tax, *o1a, *o1b;
    uint64_t oStride;
    const uint32_t *nOut;
    const uint64_t *rowFirst;
    uint32_t k, nQueries;
    OutRow *dst;
    BatchStatus *st;
    // the NARROW result format (cf_batch_set_result_format): 16-byte rows — the taxID is a function of the taxon index — and one
    // byte per query: rows printed | mate 1 took part << 6 | mate 2 << 7 (max_score is a function of those and the lengths)
    NarrowRow *dstNarrow;            // != nullptr: narrow rows instead of dst
    uint8_t *qinfo;
    const uint8_t *pass;
    int32_t paired;
};
CF_DEV OutRow row_of_one(uint64_t tax, uint64_t a, uint64_t bb) {
    OutRow o; o.taxID = tax; o.uniqueID = (uint32_t)a; o.score = (uint32_t)(a >> 32); o.hitLen = (uint32_t)bb; o.tidx = (uint32_t)(bb >> 32);
    return o;
}
CF_DEV void compact_body(const DCompact &c, uint32_t q) {
    if (q == c.nQueries && c.st) c.st->rowsOut = c.rowFirst[q];
    if (q >= c.nQueries) return;
    const uint32_t n = c.nOut[q] < c.k ? c.nOut[q] : c.k;
    const uint64_t f = c.rowFirst[q];
    if (c.dstNarrow) {
        if (n <= kFieldRows) {
#pragma unroll
            for (uint32_t i = 0; i < kFieldRows; i++) if (i < n) {
                const uint64_t a = c.o1a[i * c.oStride + q], bb = c.o1b[i * c.oStride + q];
                c.dstNarrow[f + i] = NarrowRow{(uint32_t)a, (uint32_t)(bb >> 32), (uint32_t)(a >> 32), (uint32_t)bb};
            }
        } else for (uint32_t i = 0; i < n; i++) { const OutRow o = c.out[(uint64_t)q * c.k + i]; c.dstNarrow[f + i] = NarrowRow{o.uniqueID, o.tidx, o.score, o.hitLen}; }
        const uint32_t r0 = c.paired ? 2 * q : q;
        c.qinfo[q] = (uint8_t)(n | (c.pass[r0] ? 0x40u : 0u) | ((c.paired && c.pass[r0 + 1]) ? 0x80u : 0u));
        return;
    }
    if (n <= kFieldRows) {
#pragma unroll
        for (uint32_t i = 0; i < kFieldRows; i++) if (i < n) c.dst[f + i] = row_of_one(c.o1tax[i * c.oStride + q], c.o1a[i * c.oStride + q], c.o1b[i * c.oStride + q]);
    } else for (uint32_t i = 0; i < n; i++) c.dst[f + i] = c.out[(uint64_t)q * c.k + i];
}

// ------------------------------------------------------------------ search
// Start of one partialSearch call at `cur` (hi_aligner.h:928-978).  Returns
//   0 = a dummy hit (top = bot = MASK) of length `len` was decided,
//   1 = ftab range [top,bot) is non-empty, extension starts at dep.
CF_DEV int ps_begin(const DIndex &ix, const DBatch &b, uint64_t wbase, uint32_t L, bool fw, uint32_t cur,
                    ReadWin &win, uint64_t &top, uint64_t &bot, uint32_t &dep, uint32_t &len, uint32_t &newCur,
                    bool &usedFtab) {
    const uint32_t ftc = (uint32_t)ix.ftabChars;
    const uint32_t left = L - cur;
    usedFtab = false;
    if (left < ftc) { len = L - cur; newCur = L; return 0; }
    uint64_t fi = 0;
    for (uint32_t i = 0; i < ftc; i++) {                // i = 0 is the rightmost char of the window
        const int c = strand_char(b, wbase, L, fw, L - cur - 1 - i, win);
        if (c > 3) { len = i + 1; newCur = cur + i + 1; return 0; }
        fi |= (uint64_t)c << (2 * i);                   // leftmost char ends up most significant
    }
    usedFtab = true;
    top = ftab_hi(ix, fi);
    bot = ftab_lo(ix, fi + 1);
    dep = cur + ftc;
    if (bot <= top) { len = ftc; newCur = dep; return 0; }
    return 1;
}

enum : int { MODE_IDLE = 0, MODE_CALL = 1, MODE_EXT = 2 };

template <int G>
CF_DEV void search_body(const DIndex &ix, const DParams &pr, const DBatch &b) {
    const int sub = Grp<G>::sub();
    const uint32_t lane = cf_lane();
    const uint32_t leaderLane = lane & ~(uint32_t)(G - 1);
    // group state (identical in the G lanes of a group)
    int mode = MODE_IDLE;
    uint32_t item = 0, L = 0, cur = 0, offset = 0, dep = 0, nh = 0, mxl = 0;
    uint64_t wbase = 0, top = 0, bot = 0;
    bool fw = true;
    HitP *hl = nullptr;
    ReadWin win{0, kNone64, 0};
    const uint32_t nItems = b.st->nItems;
    // per-wave work queue
    uint32_t wnext = 0, wend = 0;
    bool exhausted = false;
    unsigned long long cFtab = 0, cPair = 0, cPair2 = 0, cSingle = 0;

    for (;;) {
        const uint64_t idleMask = cf_ballot(mode == MODE_IDLE && sub == 0);
        if (idleMask) {
            if (wnext >= wend && !exhausted) {
                uint32_t base = 0;
                if (lane == 0) base = (uint32_t)cf_atomic_add(&b.cursor[0], (unsigned long long)kSearchChunk);
                base = cf_first_lane_u32(base);
                if (base >= nItems) { exhausted = true; wnext = wend = 0; }
                else { wnext = base; wend = base + kSearchChunk < nItems ? base + kSearchChunk : nItems; }
            }
            const uint32_t avail = wend - wnext;
            const uint32_t nIdle = (uint32_t)cf_popc64(idleMask);
            if (mode == MODE_IDLE) {
                const uint32_t rnk = (uint32_t)cf_popc64(idleMask & ((1ull << leaderLane) - 1));
                if (rnk < avail) {
                    item = wnext + rnk;
                    const uint32_t rd = b.items[item >> 1];
                    fw = (item & 1) == 0;
                    wbase = b.woff[rd];
                    L = b.rlen[rd];
                    hl = b.hits + b.hitBase[rd] + (fw ? 0u : b.hitCap[rd]);
                    cur = 0; nh = 0; mxl = 0; win.idx = kNone64;
                    mode = MODE_CALL;
                }
            }
            wnext += nIdle < avail ? nIdle : avail;
        }
        if (cf_ballot(mode != MODE_IDLE) == 0) {
            if (exhausted) break;
            continue;
        }
        // ---- one step per group
        bool push = false;
        uint64_t pTop = kNone64, pBot = kNone64;
        uint32_t pLen = 0;
        if (mode == MODE_CALL) {
            offset = cur;
            uint32_t len = 0, newCur = 0;
            bool usedFtab;
            const int r = ps_begin(ix, b, wbase, L, fw, cur, win, top, bot, dep, len, newCur, usedFtab);
            if (usedFtab) cFtab++;
            if (r == 0) { push = true; pLen = len; cur = newCur; }
            else mode = MODE_EXT;
        }
        if (mode == MODE_EXT) {
            bool stop = dep >= L;
            if (!stop) {
                const int c = strand_char(b, wbase, L, fw, L - dep - 1, win);
                if (c > 3) stop = true;
                else {
                    uint64_t t, bb; bool two;
                    if (bot - top > 1) { cPair++; } else { cSingle++; }
                    rank_any<G>(ix, c, top, bot, t, bb, two);
                    if (two) cPair2++;
                    if (bb <= t) stop = true;
                    else { top = t; bot = bb; dep++; stop = dep >= L; }
                }
            }
            if (stop) { push = true; pTop = top; pBot = bot; pLen = dep - offset; cur = dep; }
        }
        if (push) {
            if (sub == 0) {
                Hit h; h.top = pTop; h.bot = pBot; h.bwoff = offset; h.len = pLen; h.nelt = 0;
                hl[nh] = hit_pack(h);
            }
            nh++;
            mxl = pLen > mxl ? pLen : mxl;
            // classifier.h:686-766: done, or skip the mismatching base and go on
            bool done = cur >= L;
            if (!done) {
                if (pLen > pr.inc) cur += 1;
                done = cur + pr.m >= L;
            }
            if (done) {
                if (sub == 0) b.nhml[item] = nhml_make(nh, mxl >= pr.m);
                mode = MODE_IDLE;
            } else mode = MODE_CALL;
        }
    }
    if (b.ops && sub == 0 && (cFtab | cPair | cSingle)) {
        cf_atomic_add(&b.ops->nFtab, cFtab); cf_atomic_add(&b.ops->nPair, cPair);
        cf_atomic_add(&b.ops->nPair2, cPair2); cf_atomic_add(&b.ops->nSingle, cSingle);
    }
}


// ---------------------------------------------------------- search, version 2
// One memory round trip per iteration for every chain of the wavefront.
//
// k_search above serialises, inside each iteration, the latencies of whatever
// its chains happen to be doing (queue refill -> read bytes -> ftab -> side).
// Here every chain is a small state machine and each iteration has ONE block of
// loads — a strand record, an ftab pair, or the side(s) of an LF step, chosen
// per chain — followed by ALU-only processing, so the wavefront waits once per
// iteration.  The strand lives in LDS as 2-bit words in search order (char j =
// j-th base from the right end of the searched strand) plus an N bit mask; the
// 10-mer ftab index is a 20-bit funnel shift of two LDS words.
//
// StrandRec (global, one per item = 2*slot + strand), W = recWords:
//   u64 words[W] | u32 nmask[W] | pad | last 16 bytes: u32 L | u32 hitIdx | u32 read | u32 0   (64, 96 or 128 B)
constexpr int rec_bytes(int W) { return ((12 * W + 16 + 31) / 32) * 32; }
constexpr int rec_lds_stride(int W) { return rec_bytes(W) + 8; }          // in LDS (search2_body)

// one thread per (item, word): 32 search-order chars of the strand from the packed read.  Char j of a strand
// record is the j-th base from the RIGHT end of the searched strand: for the forward strand base L-1-j (the
// read's pairs reversed), for the reverse complement the complement of base j (the read's pairs as they are).
// word k of item's record (w, m) and — for k == 0 — the record's meta chunk; false: no such item
CF_DEV bool pack_word(const DBatch &b, uint32_t item, uint32_t k, uint64_t &w, uint32_t &m, uint32_t *meta) {
    if (item >= b.st->nItems) return false;
    const uint32_t rd = b.items[item >> 1];
    const bool fw = (item & 1) == 0;
    const uint64_t wo = b.woff[rd];
    const uint32_t L = b.rlen[rd];
    w = 0; m = 0;
    if (32 * k < L) {
        const uint32_t have = L - 32 * k;                         // chars of this word that exist (>= 1)
        if (fw) {
            const int32_t s0 = (int32_t)(L - 1 - 32 * k) - 31;   // first base of the 32-base window that ends at base L-1-32k
            uint64_t x;
            uint32_t mm;
            if (s0 >= 0) {
                const uint32_t wi = (uint32_t)s0 >> 5, sh = (uint32_t)s0 & 31;
                x = b.bases[wo + wi] >> (2 * sh);
                mm = b.nmask[wo + wi] >> sh;
                if (sh) { x |= b.bases[wo + wi + 1] << (64 - 2 * sh); mm |= b.nmask[wo + wi + 1] << (32 - sh); }
            } else {                                              // the window starts before the read: zeros below base 0
                const uint32_t neg = (uint32_t)(-s0);             // 1..31
                x = b.bases[wo] << (2 * neg);
                mm = b.nmask[wo] << neg;
            }
            w = pair_reverse(x);
            m = cf_brev32(mm);
        } else {
            w = ~b.bases[wo + k];
            m = b.nmask[wo + k];
        }
        if (have < 32) { w &= (1ull << (2 * have)) - 1; m &= (1u << have) - 1u; }
        w &= ~spread_pairs(m);                                    // an N carries code 0
    }
    if (k == 0) {
        meta[0] = L;
        meta[1] = (uint32_t)(b.hitBase[rd] + (fw ? 0u : b.hitCap[rd]));
        meta[2] = rd; meta[3] = 0;
    }
    return true;
}
// the word into a record image at `rec` (the record itself, or the block's LDS copy of it: k_pack)
CF_DEV void pack_store(uint8_t *rec, uint32_t W, uint32_t k, uint64_t w, uint32_t m, const uint32_t *meta) {
    reinterpret_cast<uint64_t *>(rec)[k] = w;
    reinterpret_cast<uint32_t *>(rec + 8 * W)[k] = m;
    if (k == 0) {
        uint32_t *dst = reinterpret_cast<uint32_t *>(rec + rec_bytes((int)W) - 16);
        dst[0] = meta[0]; dst[1] = meta[1]; dst[2] = meta[2]; dst[3] = meta[3];
    }
    if (k == W - 1)                                               // the pad between the masks and the meta chunk
        for (uint32_t z = 12 * W; z < (uint32_t)rec_bytes((int)W) - 16; z += 4) *reinterpret_cast<uint32_t *>(rec + z) = 0;
}
CF_DEV void pack_body(const DBatch &b, uint8_t *recs, uint32_t W, uint32_t t) {
    const uint32_t item = t / W, k = t % W;
    uint64_t w; uint32_t m, meta[4];
    if (pack_word(b, item, k, w, m, meta)) pack_store(recs + (uint64_t)item * rec_bytes((int)W), W, k, w, m, meta);
}

CF_DEV uint64_t side_of(const DIndex &ix, uint64_t row) {
    if (ix.small) return (uint64_t)(((uint64_t)(uint32_t)(row >> 7) * 0xAAAAAAABull) >> 33);   // (row>>7)/3
    return row / kSideChars;
}

// 32-bit formulation of the masked popcount (full-rate VALU ops only): xor with pat32(c) turns
// the 2-bit chars equal to c into 11; a 16-char dword contributes its matches below bit 2n.
CF_DEV uint32_t pat32(int c) { return ((c & 1) ? 0u : 0x55555555u) | ((c & 2) ? 0u : 0xaaaaaaaau); }
CF_DEV uint32_t cnt16(uint32_t w, uint32_t pat, int n2) {
    const uint32_t x = w ^ pat;
    const uint32_t m = x & (x >> 1) & 0x55555555u;
    const int k = n2 < 0 ? 0 : (n2 > 31 ? 31 : n2);          // bit 31 is never a match bit, so 31 stands for "all"
    return (uint32_t)cf_popc32(m & ((1u << k) - 1u));
}
// this lane's share of #{ j < o : bwt[j] == c } in one side
template <int G>
CF_DEV uint32_t side_count1(const Side<G> &s, uint32_t pat, uint32_t o) {
    const int sub = Grp<G>::sub();
    uint32_t acc = 0;
#pragma unroll
    for (int i = 0; i < 8 / G; i++) {
        const int j = sub + i * G;
        if (j < 6) {
            const int n2 = 2 * ((int)o - 64 * j);
            acc += cnt16((uint32_t)s.v[i].x, pat, n2) + cnt16((uint32_t)(s.v[i].x >> 32), pat, n2 - 32) +
                   cnt16((uint32_t)s.v[i].y, pat, n2 - 64) + cnt16((uint32_t)(s.v[i].y >> 32), pat, n2 - 96);
        }
    }
    return acc;
}
// even bits of x (bit 2j -> bit j)
CF_DEV uint32_t squeeze_even(uint64_t x) {
    x &= 0x5555555555555555ull;
    x = (x | (x >> 1)) & 0x3333333333333333ull;
    x = (x | (x >> 2)) & 0x0f0f0f0f0f0f0f0full;
    x = (x | (x >> 4)) & 0x00ff00ff00ff00ffull;
    x = (x | (x >> 8)) & 0x0000ffff0000ffffull;
    x = (x | (x >> 16)) & 0x00000000ffffffffull;
    return (uint32_t)x;
}
// The 24 plane entries of side s (thread s): 6 groups of 64 rows x 4 characters.  The running counts start from the side's
// own occ[]; the '$' (stored as an A, bt2_idx.h:2192-2227) is no character: its bit is cleared and it is not counted.
CF_DEV void occ_planes_body(const DIndex &ix, uint8_t *planes, uint64_t s, uint64_t nSides) {
    if (s >= nSides) return;
    const uint64_t *p = reinterpret_cast<const uint64_t *>(ix.sides + s * 128);
    uint64_t run[4] = {p[12] + ix.fchr0, p[13] + ix.fchr1, p[14] + ix.fchr2, p[15] + ix.fchr3};
    uint64_t *out = reinterpret_cast<uint64_t *>(planes + s * 384);
    for (int g = 0; g < 6; g++) {
        for (int c = 0; c < 4; c++) {
            uint64_t bits = (uint64_t)squeeze_even(match_mask(p[2 * g], c)) | ((uint64_t)squeeze_even(match_mask(p[2 * g + 1], c)) << 32);
            if (c == 0 && s == ix.zSide && ix.zIn / 64 == (uint32_t)g) bits &= ~(1ull << (ix.zIn & 63));
            out[8 * g + 2 * c] = bits;
            out[8 * g + 2 * c + 1] = run[c];
            run[c] += (uint64_t)cf_popc64(bits);
        }
    }
}

// The 16 pair-plane entries of row group g (thread g), from the planes.
CF_DEV void pair_planes_body(const DIndex &ix, uint8_t *planes2, uint64_t g, uint64_t nGroups) {
    if (g >= nGroups) return;
    const uint64_t *p1 = reinterpret_cast<const uint64_t *>(ix.planes + g * 64);       // {bits, base} x 4
    uint64_t bits[16];
    for (int i = 0; i < 16; i++) bits[i] = 0;
    for (uint32_t o = 0; o < 64; o++) {
        if (64 * g + o > ix.len) break;                          // rows past the text (padding of the last side)
        int c1 = -1;
        for (int c = 0; c < 4; c++) if ((p1[2 * c] >> o) & 1) c1 = c;
        if (c1 < 0) continue;                                    // the '$' row: preceded by nothing
        const uint64_t j = p1[2 * c1 + 1] + popc_below(p1[2 * c1], o);                 // LF(c1, row): the row of the suffix one base longer
        const uint64_t *pj = reinterpret_cast<const uint64_t *>(ix.planes + (j >> 6) * 64);
        int c0 = -1;
        for (int c = 0; c < 4; c++) if ((pj[2 * c] >> (j & 63)) & 1) c0 = c;
        if (c0 < 0) continue;                                    // that suffix is the whole text
        bits[4 * c1 + c0] |= 1ull << o;
    }
    uint64_t *out = reinterpret_cast<uint64_t *>(planes2 + g * 256);
    for (int pr = 0; pr < 16; pr++) {
        const int c1 = pr >> 2, c0 = pr & 3;
        const uint64_t r1 = p1[2 * c1 + 1];                      // LF(c1, 64 g)
        const uint64_t *e = reinterpret_cast<const uint64_t *>(ix.planes + (r1 >> 6) * 64 + 16 * c0);
        out[2 * pr] = bits[pr];
        out[2 * pr + 1] = e[1] + popc_below(e[0], (uint32_t)r1 & 63u);                 // LF(c0, LF(c1, 64 g))
    }
}

CF_DEV uint64_t swap1_64(uint64_t v) {
    return ((uint64_t)cf_swap1((uint32_t)(v >> 32)) << 32) | cf_swap1((uint32_t)v);
}

// Rank through a per-lane LDS table.  Counting matches below an offset by masking every dword costs
// ~7 VALU ops per dword and count; of a side's 24 dwords only ONE is cut by the offset, the others
// count fully or not at all.  So each lane stores the match masks of its dwords and their running
// popcount (exclusive prefix) once per side, and a count is: prefix[l] + popc(mask[l] & low bits),
// with l picked by the offset — two LDS reads and a handful of ALU ops per count.
template <int G>
struct RankTab {
    static constexpr int NB = (G == 1) ? 6 : (G == 2) ? 3 : (G == 4) ? 2 : 1;   // chunk slots of a lane that can hold BWT
    static constexpr int ND = 4 * NB;                                           // match dwords per lane
    static constexpr bool WIDE = ND * 16 > 255;                                 // prefix entries: u8 unless they could overflow
    static constexpr int PW = WIDE ? (ND + 2) / 2 : (ND + 4) / 4;               // dwords holding the ND + 1 prefix entries
    static constexpr int MW = ND / 2;                                           // match masks use the even bits only: two per dword
    static constexpr int WORDS = ((MW + PW + 3) / 4) * 4;                       // per-lane LDS dwords (16-byte multiple)
};

template <int G>
CF_DEV void rank_tab_build(const Side<G> &s, uint32_t pat, uint32_t *scr) {
    using T = RankTab<G>;
    const int sub = Grp<G>::sub();
    // masks are written pair by pair as they are made (nothing of the table stays in registers); the running
    // popcount is kept packed the way it is stored: entry e = matches in the dwords before dword e
    uint32_t run = 0, acc = 0;                                   // acc: the prefix dword being filled
    constexpr int PER = T::WIDE ? 2 : 4, BITS = T::WIDE ? 16 : 8;
#pragma unroll
    for (int i = 0; i < T::NB; i++) {
        const bool bwt = sub + i * G < 6;                     // chunks 6, 7 hold occ[]
        const uint32_t w[4] = {(uint32_t)s.v[i].x, (uint32_t)(s.v[i].x >> 32), (uint32_t)s.v[i].y, (uint32_t)(s.v[i].y >> 32)};
        uint32_t mm[4];
#pragma unroll
        for (int e = 0; e < 4; e++) {
            const uint32_t x = w[e] ^ pat;
            mm[e] = bwt ? (x & (x >> 1) & 0x55555555u) : 0u;
            const int d = 4 * i + e;                             // entry d = run before this dword
            acc |= run << (BITS * (d % PER));
            if (d % PER == PER - 1) { scr[T::MW + d / PER] = acc; acc = 0; }
            run += (uint32_t)cf_popc32(mm[e]);
        }
        scr[2 * i] = mm[0] | (mm[1] << 1);
        scr[2 * i + 1] = mm[2] | (mm[3] << 1);
    }
    acc |= run << (BITS * (T::ND % PER));                        // entry ND: all of this lane's dwords
    scr[T::MW + T::ND / PER] = acc;
    cf_compiler_fence();
}

// this lane's share of #{ j < o : bwt[j] == c } from the table of rank_tab_build
template <int G>
CF_DEV uint32_t rank_tab_count(const uint32_t *scr, uint32_t o) {
    using T = RankTab<G>;
    const uint32_t sub = (uint32_t)Grp<G>::sub();
    const uint32_t gd = o >> 4, jj = gd >> 2;                                   // dword and chunk the offset falls into
    const bool mine = jj < 6 && (jj & (uint32_t)(G - 1)) == sub;
    uint32_t below = jj > sub ? (jj - sub + (uint32_t)(G - 1)) / (uint32_t)G : 0u;   // my chunks wholly below the offset
    below = below > (uint32_t)T::NB ? (uint32_t)T::NB : below;
    const uint32_t l = mine ? 4 * (jj / (uint32_t)G) + (gd & 3) : 4 * below;
    const uint32_t pre = T::WIDE ? reinterpret_cast<const uint16_t *>(scr + T::MW)[l] : reinterpret_cast<const uint8_t *>(scr + T::MW)[l];
    const uint32_t md = mine ? (scr[l >> 1] >> (l & 1)) & 0x55555555u : 0u;
    return pre + (uint32_t)cf_popc32(md & ((1u << (2 * (o & 15))) - 1u));
}

// start of a partialSearch call at `cur` from the LDS copy of the strand (hi_aligner.h:928-978):
// 0 = dummy hit of length `len` decided (newCur set), 1 = look up ftab[fi]
template <class LM> CF_DEV int ps_begin2(const uint64_t *lw, LM lm, uint32_t L, uint32_t cur, uint32_t ftc, uint32_t wide, uint64_t &fi,
                     uint32_t &len, uint32_t &newCur) {
    const uint32_t left = L - cur;
    if (left < ftc) { len = left; newCur = L; return 0; }
    const uint32_t k = cur >> 5, sh = cur & 31;
    uint32_t m = lm[k] >> sh;
    if (sh) m |= lm[k + 1] << (32 - sh);
    const uint32_t m10 = m & ((1u << ftc) - 1);
    if (m10) { const uint32_t i = (uint32_t)cf_ctz32(m10); len = i + 1; newCur = cur + i + 1; return 0; }
    uint64_t v = lw[k] >> (2 * sh);
    if (sh) v |= lw[k + 1] << (64 - 2 * sh);
    // 2 = the next `wide` bases exist and are N-free: look the whole wide-mer up (its low 2*ftc bits are the ftab index)
    if (wide && left >= wide && (m & ((1u << wide) - 1)) == 0) { fi = v & ((1ull << (2 * wide)) - 1); return 2; }
    fi = v & ((1ull << (2 * ftc)) - 1);
    return 1;
}

// One entry of the wide ftab: thread t = the wide-mer whose low 2*ftabChars bits are an ftab index and whose higher bit
// pairs are the bases the search would extend by next, in order (hi_aligner.h:946-1008 done ahead of time).
constexpr uint64_t kWideSizeMax = 0xfffffull;               // "does not fit": the caller steps (what wide_size returns for it)
constexpr uint64_t kWideMaskRows = 14;                       // ranges up to this many rows carry their size in the code, and a mask
#ifndef CF_WIDE_CTX_ROWS
#define CF_WIDE_CTX_ROWS 4
#endif
constexpr uint64_t kWideCtxRows = CF_WIDE_CTX_ROWS;          // ... of these, ranges up to this many rows carry the rows' context instead (0: masks only — a build for A/B runs)
constexpr uint32_t wide_ctx_bases(uint32_t rows) { return rows == 1 ? 8u : rows == 2 ? 4u : 2u; }   // bases of context per row (16 bits of payload): row r at bits 2 nb r
CF_DEV uint64_t wide_entry(uint64_t top, uint64_t size, uint32_t depthOverFtab, uint64_t cap, bool masked, uint32_t mask) {
    if (size == 0) return 0;
    uint64_t code, payload;
    if (size >= cap || size >= 0xffffull) { code = 15; payload = 0xffff; }
    else if (size <= kWideMaskRows && masked) { code = size; payload = mask; }
    else { code = 15; payload = size; }
    return top | ((uint64_t)depthOverFtab << 40) | (code << 44) | (payload << 48);
}
CF_DEV uint64_t wide_size(uint64_t e) {                      // 0: ftab miss; kWideSizeMax: does not fit
    const uint64_t code = (e >> 44) & 15u, payload = e >> 48;
    return code != 15 ? code : payload == 0xffff ? kWideSizeMax : payload;
}
CF_DEV bool wide_masked(uint64_t e) { const uint64_t code = (e >> 44) & 15u; return code != 0 && code != 15; }   // (means something when D = wideChars)
// cap: ranges of `cap` rows or more are stored as "does not fit" (kWideSizeMax in production; the tests lower it)
CF_DEV void wide_ftab_body(const DIndex &ix, uint32_t wideChars, uint64_t *table, uint64_t t, uint64_t cap = kWideSizeMax) {
    if (t >= (1ull << (2 * wideChars))) return;
    const uint32_t ftc = (uint32_t)ix.ftabChars;
    const uint64_t fi = t & ((1ull << (2 * ftc)) - 1);
    uint64_t top = ftab_hi(ix, fi), bot = ftab_lo(ix, fi + 1);
    if (bot <= top) { table[t] = 0; return; }
    uint32_t j = ftc;
    for (; j < wideChars; j++) {
        const int c = (int)((t >> (2 * j)) & 3);
        uint64_t nt, nb; bool two;
        rank_any<1>(ix, c, top, bot, nt, nb, two);
        if (nb <= nt) break;
        top = nt; bot = nb;
    }
    // the next-pairs mask of a small range that is alive at wideChars: the very steps the search would take (so the '$' row and
    // every other rule of the step are in it).  A base the range survives although no pair with it does — the occurrence at the
    // start of the text — cannot be told from the mask: such an entry goes without one.
    uint32_t mask = 0;
    bool masked = false;
    if (kWideCtxRows > 0 && j == wideChars && bot - top <= kWideCtxRows && bot - top < cap) {
        // up to four rows: the bases that precede their suffixes, nearest first (the rows' own LF chains)
        masked = true;
        const uint32_t rows = (uint32_t)(bot - top), nb = wide_ctx_bases(rows);
#pragma unroll 1
        for (uint32_t r = 0; r < rows && masked; r++) {
            uint64_t row = top + r;
#pragma unroll 1
            for (uint32_t i = 0; i < nb; i++) {
                int c;
                if (!lf_own_any(ix, row, c)) { masked = false; break; }      // the start of the text: no context (code 15)
                mask |= (uint32_t)c << (2 * nb * r + 2 * i);
            }
        }
    } else if (j == wideChars && bot - top <= kWideMaskRows && bot - top < cap) {
        masked = true;
#pragma unroll 1
        for (int c1 = 0; c1 < 4; c1++) {
            uint64_t t1, b1; bool two;
            rank_any<1>(ix, c1, top, bot, t1, b1, two);
            if (b1 <= t1) continue;
            uint32_t m4 = 0;
#pragma unroll 1
            for (int c0 = 0; c0 < 4; c0++) {
                uint64_t t2, b2;
                rank_any<1>(ix, c0, t1, b1, t2, b2, two);
                if (b2 > t2) m4 |= 1u << c0;
            }
            if (!m4) masked = false;
            mask |= m4 << (4 * c1);
        }
    }
    table[t] = wide_entry(top, bot - top, j - ftc, cap, masked, mask);
}

enum : int { S_IDLE = 0, S_REC = 1, S_CALL = 2, S_FTAB = 3, S_EXT = 4, S_EXTB = 5, S_FTABW = 6, S_POS = 7, S_TXT = 8, S_ISA = 9, S_REC2 = 10 };

// Text verification of a unique match.  Once the SA range of a partialSearch call is down to ONE row, every further base is one
// LF step = one random 128-byte request, for as long as the read keeps matching — 68 of the 93 requests per read on the
// benchmark's reads.  But a single row is a single text position: when the chain sits on a row of the SA sample
// (every 2^posRate-th), it reads SA[row] = p (S_POS), compares the bases still to come with the text left of p, 64 per request
// (S_TXT: the read's search-order words against the pair-reversed text words, stop at the first difference or N), and then
// needs the ROW of the suffix where the match ended — the hit the reference would hold, whose row getGenomeIdx resolves:
// the inverse sample gives the row of the next sampled position at or right of it (S_ISA), from where at most
// 2^posRate - 1 ordinary LF steps (over bases already known to match) reach it.  The state that results — row, depth — is
// exactly the one the step-by-step path passes through, and the ordinary step that follows finds the same mismatch, N or
// read end.  Tried only after a few single-row steps succeeded in a row (a chance match dies within a step or two), and
// kept only when it saves steps (>= 4 matched); otherwise the chain just keeps stepping.
// Lazy hits.  The hit list of a strand is read by nobody unless the strand has a hit of minHitLen (k_post: such a strand cannot
// score, extend, or win the strand choice) — and that is most strands: the one that does not match ends every call after
// 14-18 bases.  Their 16-byte records were a fifth of the search kernel's time (partial-line writes to HBM beside its reads).
// So a strand starts lazy: its first kLazyHits hits wait in LDS, later ones are not kept at all; the first hit of minHitLen
// flushes what waits and switches the strand to direct stores — or, when hits have already been let go, sends the chain
// through the strand once more, direct from the start (rare: three short hits and then a long one).
constexpr uint32_t kLazyHits = 2;
// ... but ONE for the one-lane kernel's 192- and 256-base records: 31 / 39 KB of LDS per block instead of 35 / 43 = five / four
// blocks per CU instead of four / three.  Measured (profiles/r04g_*): config 4 (2 x 150 bp) 6.10 -> 6.54e8 mates/s, config 5 (250 bp)
// 2.06 -> 2.32e8 reads/s.  A strand whose first two hits are short and whose third is long then searches once more.
constexpr uint32_t lazy_hits(int G, int W) { return G == 1 && W >= 6 ? 1u : kLazyHits; }
constexpr uint32_t kVerifyMinLeft = 12;      // bases still to come for the detour to be worth three requests
// (successful single-row steps before it is tried: DIndex::verifyMinRun, 0 by default)

// COUNT: also tally the LF steps / ftab lookups into b.ops (the instrumented pass behind
// cf_batch_opcounts); the production launch carries no counters.
// BLOCKS (G = 1 only): LF steps over the occurrence planes (DIndex::planes): one 16-byte load and two masked popcounts per
// step, no per-lane LDS table
template <int G, int W, bool COUNT, bool BLOCKS = false, int LZN = 0, bool MULTI = false>
CF_DEV void search2_body(const DIndex &ix, const DParams &pr, const DBatch &b, uint8_t *ldsBlock) {
    static_assert(!BLOCKS || G == 1, "the planes are read one chain per lane");
    constexpr int PER = 8 / G;                       // 16-byte chunks of a side per lane
    constexpr int RB = rec_bytes(W);
    constexpr int RCH = RB / (16 * G);               // chunks of a strand record per lane
    constexpr int NV = BLOCKS ? (RCH > 2 ? RCH : 2) : PER;   // 16-byte registers of the load slot
    static_assert(RCH >= 1 && RCH <= NV, "record does not fit the load slot");
    constexpr uint32_t kRawPieces = W / 2 + (W + 3) / 4;         // 16-byte pieces of a read's W words and W mask words
    static_assert(G != 1 || (int)kRawPieces <= NV, "packed read does not fit the load slot");
    struct Slot { u64x2 v[NV]; };
    const int sub = Grp<G>::sub();
    const uint32_t lane = cf_lane();
    const uint32_t leaderLane = lane & ~(uint32_t)(G - 1);
    // LDS stride of a record: RB + 8 bytes, an ODD number of 8-byte units.  All lanes read the same field of their own
    // record at once; with the records RB = 64 / 96 / 128 bytes apart those reads fall into 2 - 4 of the 32 banks
    // (a 16-way conflict on every one of the ~20 LDS reads of an iteration — the LDS pipe, shared by the CU, was what the
    // kernel waited for); an odd stride spreads them over all banks.
    constexpr int RBL = rec_lds_stride(W);
    uint8_t *lrec = ldsBlock + (size_t)(cf_local_thread() / G) * RBL;
    constexpr uint32_t LZ = LZN ? (uint32_t)LZN : lazy_hits(G, W);
    // the chain's first LZ hits while its strand has none of minHitLen yet (see the push step); behind the records
    // and the rank tables
    uint64_t *lhit = reinterpret_cast<uint64_t *>(ldsBlock + (size_t)(cf_block_threads() / G) * RBL +
                                                  (BLOCKS ? (size_t)0 : (size_t)cf_block_threads() * 4 * RankTab<G>::WORDS)) +
                     (size_t)(cf_local_thread() / G) * 2 * LZ;
    // per-lane rank table behind the block's strand records (cf_threads_per_block() / G chains)
    uint32_t *scr = BLOCKS ? nullptr : reinterpret_cast<uint32_t *>(ldsBlock + (size_t)(cf_block_threads() / G) * RBL) + (size_t)cf_local_thread() * RankTab<G>::WORDS;
    const uint64_t *lw = reinterpret_cast<const uint64_t *>(lrec);
    const uint32_t *lm = reinterpret_cast<const uint32_t *>(lrec + 8 * W);
    // the record's last 16 bytes: {L, hitIdx} as packed by k_pack, then the work item — chain constants that
    // live in LDS, not in registers (one ds_read where they are needed)
    uint32_t *lmeta = reinterpret_cast<uint32_t *>(lrec + RB - 16);
    const uint32_t ftc = (uint32_t)ix.ftabChars;
    // chain state, identical in the G lanes of a chain
    int mode = S_IDLE;
    uint32_t cur = 0, dep = 0;
    // hits pushed so far (bits 0-7) | longest of them (bits 8-19) | offset of the running call (bits 20-31):
    // three small numbers (reads <= 256 bases; the launcher checks the hit capacity) in one register
    uint32_t nhmx = 0;
    uint64_t top = 0, bot = 0;
    // One register pair for two values that are never alive together: the ftab index between S_CALL and
    // S_FTAB, and LF(top) of a two-sided step while it waits for the bot side (S_EXT -> S_EXTB).
    uint64_t aux = 0;
    // text verification: bit 0 = not (again) in this call, bit 1 = at least one full 64-base window matched, bit 4 = endDep holds, bits 8.. = successful
    // single-row steps in a row.  aux holds the text position during S_POS .. S_ISA (a single row never needs it for S_EXTB).
    uint32_t vf = 0;
    uint32_t endDep = 0;                             // vf bit 4: the text showed where the unique match ends — at this depth the next base fails
    // verification of a small range (DIndex::multiRows): bit 31 = under way; bits 0-3 = the row being compared (i), 4-7 = rows that
    // reach the longest match so far (S), 8-19 = that length (Mmax), 20-23 = rows of the range (R).  While it runs, endDep holds
    // the depth it started at, and bot the text position where the first of the longest matches ended (the range is top .. top + R)
    uint32_t mv = 0;
    uint32_t lz = 0;                                 // bit 0: the strand's hits are still held back (lazy hits); bit 1: the chain's read holds an N (or may: records made by k_pack); bit 2: one-row hits that end in the text go out in the position form (DBatch::lazyHits bit 2)
    uint32_t wnext = 0, wend = 0;
    bool exhausted = false;
    // The one-lane kernel's work items (DBatch::itemMeta, 16 bytes each) come in with the chunk: the lanes of a wavefront that
    // claims 64 items load them TOGETHER, lane l the record of item base + l (eight lines, one instruction), and a lane that
    // takes an item later gets its record from the lane that holds it (three ds_bpermute).  Until round 4 every chain fetched its
    // own record in an iteration of its own (S_REC): one of ~14 iterations per strand, and a request.
    constexpr bool SELF = G == 1;                    // (taken when the batch has item records: b.itemMeta)
    u64x2 cmeta{0, 0};                               // the record of item cbase + lane
    uint32_t cbase = 0, pend = 0;                    // pend: the item a chain in S_REC waits for (its chunk's records are in flight)
    unsigned long long cFtab = 0, cPair = 0, cPair2 = 0, cSingle = 0, cFtabW = 0, cVerify = 0, cText = 0, cPos = 0;
    const int32_t posRate = ix.posRate;
    const int32_t isaRate = MULTI ? posRate : ix.isaRate;          // (the small-range kernels run where both samples hold every row)
    const uint32_t nItems = b.st->nItems;            // made by the plan kernels of this batch (0 when the hit pool is too small)
    const uint32_t wideChars = (uint32_t)ix.wideChars;

    for (;;) {
        // ---- refill idle chains from the per-wave queue
        uint32_t item = 0;                           // only meaningful in the iteration that fetches the record
        const bool selfRec = SELF && b.itemMeta != nullptr;
        // a chain's item record out of the lanes' registers: the chain's constants go to LDS, the word offset stays in aux, and the
        // chain is where S_REC used to leave it — about to load the read's words (wave-uniform: every lane makes the call)
        auto takeMeta = [&](bool mine, uint32_t it) {
            const int src = (int)((mine ? it - cbase : lane) & (uint32_t)(CF_WAVE - 1));
            const uint32_t m0 = cf_shfl((uint32_t)cmeta.x, src), m1 = cf_shfl((uint32_t)(cmeta.x >> 32), src), m2 = cf_shfl((uint32_t)cmeta.y, src);
            if (mine) {
                aux = (uint64_t)m0 | ((uint64_t)(m1 & (kItemHasN | kItemPre)) << 32);         // (bit 63 = kItemHasN: the read holds an N; bit 62 = kItemPre)
                lmeta[0] = m1 & ~(kItemHasN | kItemPre); lmeta[1] = m2; lmeta[2] = it;
                cf_compiler_fence();
                mode = S_REC2;
            }
        };
        if (selfRec && cf_ballot(mode == S_REC)) takeMeta(mode == S_REC, pend);      // records claimed an iteration ago: landed by now
        const uint64_t idleMask = cf_ballot(mode == S_IDLE && sub == 0);
        if (idleMask) {
            bool fresh = false;
            if (wnext >= wend && !exhausted) {
                uint32_t base = 0;
                if (lane == 0) base = (uint32_t)cf_atomic_add(&b.cursor[0], (unsigned long long)kSearchChunk);
                base = cf_first_lane_u32(base);
                if (base >= nItems) { exhausted = true; wnext = wend = 0; }
                else {
                    wnext = base; wend = base + kSearchChunk < nItems ? base + kSearchChunk : nItems;
                    if (selfRec) {                   // (waited for with the iteration's other loads; used from the next iteration on)
                        cbase = base; fresh = true;
                        cmeta = base + lane < nItems ? cf_load16(reinterpret_cast<const uint8_t *>(b.itemMeta + 4 * (size_t)(base + lane))) : u64x2{0, 0};
                    }
                }
            }
            const uint32_t avail = wend - wnext;
            const uint32_t nIdle = (uint32_t)cf_popc64(idleMask);
            bool took = false;
            if (mode == S_IDLE) {
                const uint32_t rnk = (uint32_t)cf_popc64(idleMask & ((1ull << leaderLane) - 1));
                if (rnk < avail) { item = wnext + rnk; mode = S_REC; pend = item; took = true; }
            }
            wnext += nIdle < avail ? nIdle : avail;
            if (selfRec && !fresh && cf_ballot(took)) takeMeta(took, item);     // the chunk's records are in the registers already
        }
        if (cf_ballot(mode != S_IDLE) == 0) {
            if (exhausted) break;
            continue;
        }
        // ---- the iteration's loads.  Every state wants 16-byte pieces from ONE address: an SA / inverse-SA sample, a wide or
        //      10-mer ftab entry, a plane entry of an LF step (1 piece), a text window (2), a strand record (RCH), or — without
        //      the planes — ONE side of an LF step (a step whose top and bot fall into different groups / sides takes two
        //      iterations, S_EXT then S_EXTB).  The states only choose the address and the number of pieces; the loads are
        //      issued once, below, for all states, and waited for once: one memory round trip per iteration.  (Loads issued
        //      state by state share their destination registers, so each would wait for the previous state's to land.)
        // Round 6: the iteration in TWO builds — with the strand records' N masks, and with every mask read folded to zero (a
        // quarter fewer instructions: the tests for an N at the call's start, in the entry's context, before every step and pair,
        // in the text window).  Which one runs is the wavefront's choice per iteration: the second unless one of its chains is on a
        // read that holds an N (lz bit 1; hardly any read does: 0.1 % of the benchmark's, a wavefront in sixteen at any time).
        // (a chain that is about to make its record counts from now on: it goes on into its first call within this iteration)
        const bool anyN = cf_ballot((mode != S_IDLE && (lz & 2u) != 0) || (mode == S_REC2 && (aux >> 63) != 0) || (mode == S_REC && !selfRec)) != 0;
        auto iter = [&](auto lmx) __attribute__((always_inline)) {
        Slot sa;
        sa.v[0] = u64x2{0, 0};
        uint64_t sS = 0;                             // the group / side loaded in this iteration
        uint32_t oT = 0, oB = 0;                     // offsets of top / bot inside it (whichever apply)
        bool same = true, stepN = false;
        int c = 0;
        const uint8_t *ldp = nullptr;
        uint32_t nch = 0, strd = 16;
        // (vf bit 4 before any verification: the wide-ftab entry's context has shown where the call ends, a few bases on — no detour)
        if (posRate >= 0 && mode == S_EXT && !(vf & 25u) && bot - top == 1 && (vf >> 8) >= ix.verifyMinRun &&
            (top & ((1ull << posRate) - 1)) == 0 && lmeta[0] - dep >= kVerifyMinLeft) {
            mode = S_POS;
            if (COUNT) cVerify++;
        } else if (MULTI && G == 1 && posRate == 0 && ix.multiRows && mode == S_EXT && !(vf & 25u) && bot - top >= 2 && bot - top <= ix.multiRows &&
                   (vf >> 8) >= ix.multiMinRun && lmeta[0] - dep >= kVerifyMinLeft) {
            mv = 0x80000000u | ((uint32_t)(bot - top) << 20);     // i = 0, S = 0, Mmax = 0
            endDep = dep;
            mode = S_POS;
            if (COUNT) cVerify++;
        }
        if (mode == S_POS) {
            const uint64_t row = top + ((MULTI && (mv >> 31)) ? (mv & 15u) : 0u);
            if (COUNT && (MULTI && (mv >> 31)) && (mv & 15u)) cText++;         // (every row's SA read beyond the first: counted with the windows)
            ldp = reinterpret_cast<const uint8_t *>(ix.saPos + 2 * ((row >> posRate) / 3)); nch = 1;      // (trio piece)
        } else if (mode == S_ISA) {
            ldp = reinterpret_cast<const uint8_t *>(ix.isa + 2 * ((aux >> isaRate) / 3)); nch = 1;
        } else if (mode == S_TXT) {
            if constexpr (G == 1) {
                // ONE 16-byte load: the text word that holds position aux - 1 and the one before it — the 33 .. 64 bases left of
                // aux (all of [0, aux) when aux < 64).  (Four words would always hold 64, but cost a second request.)
                const uint64_t b0 = aux >= 64 ? ((aux - 1) >> 5) - 1 : 0;
                ldp = reinterpret_cast<const uint8_t *>(ix.text) + 8 * b0; nch = 1;
            } else {
                // the 128 bases of the four text words that hold the 64 left of position aux (all of [0, aux) when aux < 64)
                const uint64_t b0 = aux >= 64 ? (aux - 64) >> 5 : 0;
                ldp = reinterpret_cast<const uint8_t *>(ix.text) + 8 * b0 + (size_t)sub * (32 / G); nch = 2 / G;
            }
            if (COUNT) cText++;
        } else if (mode == S_REC) {
            if (G == 1 && b.itemMeta) nch = 0;       // (the chunk's item records are on their way: nothing to ask for)
            else { ldp = b.recs + (uint64_t)item * RB + (size_t)sub * (RB / G); nch = RCH; }
        } else if (G == 1 && mode == S_REC2) {
            // the read's W packed words, then (below) its W mask words: aux = the read's word offset
            // (its mask words only when it holds an N: else they are zero, and one request fewer)
            // (the forward item of an N-free read: the offset is that of its words in search order, kItemPre)
            ldp = reinterpret_cast<const uint8_t *>(b.bases + (uint32_t)aux); nch = (aux >> 63) ? kRawPieces : W / 2;
        } else if (mode == S_FTAB) {
            ldp = reinterpret_cast<const uint8_t *>(ix.ftab + aux); nch = 1;          // {ftab[aux], ftab[aux + 1]}
        } else if (mode == S_FTABW) {
            ldp = reinterpret_cast<const uint8_t *>(ix.wide + aux); nch = 1;
        } else if (mode == S_EXT || mode == S_EXTB) {
            c = (int)((lw[dep >> 5] >> (2 * (dep & 31))) & 3);
            stepN = mode == S_EXT && ((lmx[dep >> 5] >> (dep & 31)) & 1u) != 0;
            if (!stepN) {
                if constexpr (BLOCKS) {
                    const uint64_t row = mode == S_EXT ? top : bot;
                    sS = row >> 6;
                    if (mode == S_EXT) {
                        oT = (uint32_t)row & 63u;
                        const uint64_t spread = bot - top;
                        same = (uint64_t)oT + spread <= 64;
                        oB = same ? oT + (uint32_t)spread : 0u;
                        // two bases with one request (pair planes) when the base after this one exists and is no N — unless the
                        // pair has just come back empty (vf bit 3): then this base alone, and the call ends (see below)
                        const uint32_t d1 = dep + 1;
                        const bool pair = ix.planes2 && !(vf & 8u) && d1 < lmeta[0] && ((lmx[d1 >> 5] >> (d1 & 31)) & 1u) == 0 &&
                                          !((vf & 16u) && d1 >= endDep);              // (the base at endDep is known to fail: no pair across it)
                        vf = pair ? (vf | 4u) : (vf & ~4u);
                    } else oB = (uint32_t)row & 63u;
                    if (vf & 4u) {
                        const uint32_t d1 = dep + 1;
                        const int c0 = (int)((lw[d1 >> 5] >> (2 * (d1 & 31))) & 3);
                        ldp = ix.planes2 + sS * 256 + 16 * (4 * c + c0); nch = 1;
                    } else { ldp = ix.planes + sS * 64 + 16 * c; nch = 1; }
                } else {
                    if (mode == S_EXT) {
                        sS = side_of(ix, top);
                        oT = (uint32_t)(top - sS * kSideChars);
                        const uint64_t spread = bot - top;
                        same = (uint64_t)oT + spread <= kSideChars;
                        oB = same ? oT + (uint32_t)spread : 0u;
                    } else {
                        sS = side_of(ix, bot);
                        oB = (uint32_t)(bot - sS * kSideChars);
                    }
                    ldp = ix.sides + sS * 128 + 16 * sub; nch = PER; strd = 16 * G;   // chunks sub, sub + G, ... (Side<G>)
                }
            }
        }
        if (nch) sa.v[0] = cf_load16(ldp);
        if (nch > 1) {                               // text windows, records, sides, packed reads
#pragma unroll
            for (int i = 1; i < NV; i++) {
                if ((uint32_t)i >= nch) continue;
                if (G == 1 && mode == S_REC2 && i >= W / 2) sa.v[i] = cf_load16(reinterpret_cast<const uint8_t *>(b.nmask + (uint32_t)aux) + 16 * (i - W / 2));
                else sa.v[i] = cf_load16(ldp + (size_t)i * strd);
            }
        }
        cf_wait_vmem();                              // the one wait of the iteration (cf_platform.hpp)
        const u64x2 ft = sa.v[0];                    // what the one-piece states asked for
        // ---- processing (ALU + LDS only, apart from the rare eftab indirection)
        bool push = false;
        uint64_t pTop = kNone64, pBot = kNone64;
        uint32_t pLen = 0;
        // a row of a small range is done: its match went on for mi bases and ended at text position pe.  Next row, or — after the
        // last — the range the step-by-step path ends this call with (S_ISA)
        auto rowDone = [&](uint32_t mi, uint64_t pe) {
            const uint32_t mmax = (mv >> 8) & 0xfffu;
            if (mi > mmax || !(mv & 0xf0u)) { mv = (mv & 0xfff0000fu) | (mi << 8) | 0x10u; bot = pe; }      // a new longest (or the first row): S = 1
            else if (mi == mmax) mv += 0x10u;
            mv += 1u;                                             // i++
            if ((mv & 15u) < ((mv >> 20) & 15u)) { dep = endDep; mode = S_POS; }
            else { dep = endDep + ((mv >> 8) & 0xfffu); aux = bot; mode = S_ISA; }
            // (a range of which ONE row matches longest could go out in the position form as well — built and measured in round 6:
            // the three registers it takes spill in the small-range kernels, 12 bytes of scratch in the loop, and the repeat-rich
            // preset's search went from 6.2 to 7.6 ms: profiles/r06h_*)
        };
        if (mode == S_POS) {
            const uint64_t row = top + ((MULTI && (mv >> 31)) ? (mv & 15u) : 0u);
            aux = trio_get(ft, (uint32_t)((row >> posRate) % 3));   // SA[row]: the bases to come lie left of it in the text
            if (MULTI && (mv >> 31)) { if (aux == 0) rowDone(0, 0); else mode = S_TXT; }
            else if (aux == 0) { vf |= 1u; mode = S_EXT; } else mode = S_TXT;
        } else if (mode == S_TXT) {
            const uint64_t p = aux;
            const uint32_t L = lmeta[0], left = L - dep;
            uint32_t cmp = left < 64 ? left : 64;
            uint64_t x0, x1;
            uint32_t winBases;                                     // bases left of p the window holds (a full window = that many matched)
            if constexpr (G == 1) {
                const uint64_t w0 = sa.v[0].x, w1 = sa.v[0].y;
                // p inside the two-word window: 33 .. 64, or p itself when p < 64.  The bases left of it, nearest first, are the pairs
                // [e-32, e) and what lies below them, each reversed; pairs below the window read as zero (cmp keeps off them)
                const uint32_t e = (uint32_t)(p - ((p >= 64 ? ((p - 1) >> 5) - 1 : 0) << 5));
                winBases = e;
                if (e >= 32) {
                    const uint32_t sh = e - 32;                   // 0 .. 32
                    x0 = sh == 0 ? w0 : sh == 32 ? w1 : (w0 >> (2 * sh)) | (w1 << (64 - 2 * sh));
                    x1 = sh ? w0 << (2 * (32 - sh)) : 0ull;
                } else {                                          // (p < 32) the e bases of w0, shifted to the top
                    x0 = e ? w0 << (2 * (32 - e)) : 0ull;
                    x1 = 0;
                }
            } else {
            // the four window words in both lanes of the chain
            uint64_t w0, w1, w2;                                  // (the fourth word of the 32 bytes is never reached: e <= 95)
            if (G == 2) {
                const uint64_t ox = swap1_64(sa.v[0].x), oy = swap1_64(sa.v[0].y);
                w0 = sub == 0 ? sa.v[0].x : ox; w1 = sub == 0 ? sa.v[0].y : oy;
                w2 = sub == 0 ? ox : sa.v[0].x;
            } else { w0 = sa.v[0].x; w1 = sa.v[0].y; w2 = sa.v[2 / G - 1].x; }
            winBases = p < 64 ? (uint32_t)p : 64u;
            // p inside the window: 64..95, or p itself when p < 64.  The 64 bases left of it, nearest first, are the pairs
            // [e-32, e) and [e-64, e-32) of the window, each reversed; pairs below position 0 read as zero (cmp keeps off them)
            const uint32_t e = (uint32_t)(p - ((p >= 64 ? (p - 64) >> 5 : 0) << 5));
            if (e >= 64) {                                        // e - 32 in 32..63: pairs from w1|w2; e - 64 in 0..31: from w0|w1
                const uint32_t sh = e & 31;
                x0 = sh ? (w1 >> (2 * sh)) | (w2 << (64 - 2 * sh)) : w1;
                x1 = sh ? (w0 >> (2 * sh)) | (w1 << (64 - 2 * sh)) : w0;
            } else if (e >= 32) {                                 // (p < 64) e - 32 in 0..31: from w0|w1; the rest lies below 0
                const uint32_t sh = e & 31;
                x0 = sh ? (w0 >> (2 * sh)) | (w1 << (64 - 2 * sh)) : w0;
                x1 = sh ? w0 << (2 * (32 - sh)) : 0ull;
            } else {                                              // (p < 32) the e bases of w0, shifted to the top
                x0 = e ? w0 << (2 * (32 - e)) : 0ull;
                x1 = 0;
            }
            }
            if (winBases < cmp) cmp = winBases;                   // no further than the window (and the text) reaches
            // the read's next bases in search order, and their N bits
            const uint32_t k = dep >> 5, sh = dep & 31;
            uint64_t q0 = lw[k] >> (2 * sh), q1 = lw[k + 1] >> (2 * sh);
            uint32_t n0 = lmx[k] >> sh, n1 = lmx[k + 1] >> sh;
            if (sh) { q0 |= lw[k + 1] << (64 - 2 * sh); q1 |= lw[k + 2] << (64 - 2 * sh); n0 |= lmx[k + 1] << (32 - sh); n1 |= lmx[k + 2] << (32 - sh); }
            uint64_t d0 = pair_reverse(x0) ^ q0, d1 = pair_reverse(x1) ^ q1;
            d0 = (d0 | (d0 >> 1)) & 0x5555555555555555ull;
            d1 = (d1 | (d1 >> 1)) & 0x5555555555555555ull;
            if (n0 | n1) { d0 |= spread_pairs(n0) & 0x5555555555555555ull; d1 |= spread_pairs(n1) & 0x5555555555555555ull; }   // an N ends the match as well
            uint32_t M = d0 ? (uint32_t)cf_ctz64(d0) >> 1 : 32u + (d1 ? (uint32_t)cf_ctz64(d1) >> 1 : 32u);
            if (M > cmp) M = cmp;
            // the match ends at text position pe; q = the sampled position at or right of it
            const uint64_t pe = p - M, pm = (1ull << isaRate) - 1;
            const uint64_t q = (pe + pm) & ~pm;
            // a difference inside the compared span: the row's next base after M more is not the read's — the step there would
            // come back empty, so the call ends when the chain gets there (S_ISA, or the steps back from the sample), unasked
            if (MULTI && (mv >> 31)) {                                       // a row of a small range: on to its next window, or the row is done
                if (M == winBases && left > M && p > M) { dep += M; aux = p - M; }
                else rowDone(dep + M - endDep, pe);
            } else {
            if (M < cmp) { vf |= 16u; endDep = dep + M; }
            if ((lz & 4u) && (M < cmp || dep + M >= L)) {
                // The call ends where the match ended — a difference inside the compared span, or the read's end — and the hit is
                // ONE row: the suffix at text position pe.  Its row is nobody's business but the resolver's, which answers from the
                // position (DIndex::posFrag, resolve_pos): the hit goes out in its position form (HitP), and the inverse-sample
                // request — with the steps back from a sampled position, where the samples are not at every one — is not made
                vf |= 1u;
                push = true; pTop = kRowIsPos | pe; pBot = pTop + 1; pLen = dep + M - (nhmx >> 20); cur = dep + M;
                if (COUNT) cPos++;
            } else if (M == winBases && left > M && p > M) {      // the whole window matches and there is more of both: next window
                dep += M; aux = p - M; vf |= 2u;
            } else if ((M < 4 || M < q - pe) && !(vf & 2u)) {     // not worth it, or the way back from q would be longer than
                vf |= 1u; mode = S_EXT;                           // what was matched: keep stepping from where the chain is
            } else {
                // the state the step-by-step path has at q: depth = (depth at pe) - (q - pe), row = the row of the suffix at q
                dep = dep + M - (uint32_t)(q - pe);
                aux = q;
                mode = S_ISA;
            }
            }
        } else if (mode == S_ISA) {
            top = trio_get(ft, (uint32_t)((aux >> isaRate) % 3));
            if (MULTI && (mv >> 31)) {
                // the rows that matched longest, in their old order (LF keeps it): the first of them is the row of the suffix at
                // aux.  Every one of them fails at the next base (a difference, an N, the start of the text) or the read is over:
                // the call ends here, as the step-by-step path would after its failing step
                bot = top + ((mv >> 4) & 15u);
                mv = 0; vf |= 1u;
                push = true; pTop = top; pBot = bot; pLen = dep - (nhmx >> 20); cur = dep;
            } else {
            bot = top + 1;
            vf |= 1u;
            if (dep >= lmeta[0] || ((vf & 16u) && dep == endDep)) { push = true; pTop = top; pBot = bot; pLen = dep - (nhmx >> 20); cur = dep; }
            else mode = S_EXT;
            }
        } else if (G == 1 && mode == S_REC && b.itemMeta) {
            // (waits for its chunk's item records: takeMeta at the top of the next iteration)
        } else if (G == 1 && mode == S_REC2) {
            // The strand record from the packed read (what k_pack writes for the other kernels): char j of a record is the j-th base
            // from the RIGHT end of the searched strand — for the forward strand base L-1-j (32-base windows of the read, pairs
            // reversed), for the reverse complement the complement of base j (the read's words as they are).  The raw words go to
            // LDS first: the forward strand's windows start at any base.
            const uint32_t L = lmeta[0];
            const bool fwd = (lmeta[2] & 1u) == 0;
            uint64_t *rw = reinterpret_cast<uint64_t *>(lrec);
            uint32_t *rm = reinterpret_cast<uint32_t *>(lrec + 8 * W);
            if ((aux >> 62) & 1u) {
                // the N-free read (nearly every read): the forward strand's words came in search order (rev_word, made by the plan),
                // the other strand's are the read's own, complemented; no mask
#pragma unroll
                for (int k = 0; k < W; k++) {
                    const uint64_t v = (k & 1) ? sa.v[k / 2].y : sa.v[k / 2].x;
                    uint64_t w = 0;
                    if (32u * (uint32_t)k < L) {
                        const uint32_t have = L - 32u * (uint32_t)k;
                        w = fwd ? v : ~v;
                        if (have < 32) w &= (1ull << (2 * have)) - 1;
                    }
                    rw[k] = w; rm[k] = 0u;
                }
            } else {
#pragma unroll
            for (int i = 0; i < W / 2; i++) { rw[2 * i] = sa.v[i].x; rw[2 * i + 1] = sa.v[i].y; }
#pragma unroll
            for (int i = 0; i < (W + 3) / 4; i++) {
                const bool hasN = (aux >> 63) != 0;
                const uint64_t a = hasN ? sa.v[W / 2 + i].x : 0ull, bq = hasN ? sa.v[W / 2 + i].y : 0ull;
                if (4 * i + 0 < W) rm[4 * i + 0] = (uint32_t)a;
                if (4 * i + 1 < W) rm[4 * i + 1] = (uint32_t)(a >> 32);
                if (4 * i + 2 < W) rm[4 * i + 2] = (uint32_t)bq;
                if (4 * i + 3 < W) rm[4 * i + 3] = (uint32_t)(bq >> 32);
            }
            cf_compiler_fence();
            uint64_t fw_[W]; uint32_t fm_[W];
#pragma unroll
            for (int k = 0; k < W; k++) {
                uint64_t w = 0; uint32_t m = 0;
                if (32u * (uint32_t)k < L) {
                    const uint32_t have = L - 32u * (uint32_t)k;          // chars of this word that exist (>= 1)
                    if (fwd) {
                        const int32_t s0 = (int32_t)(L - 1 - 32u * (uint32_t)k) - 31;   // first base of the 32-base window that ends at base L-1-32k
                        uint64_t x; uint32_t mm;
                        if (s0 >= 0) {
                            const uint32_t wi = (uint32_t)s0 >> 5, sh = (uint32_t)s0 & 31;
                            x = lw[wi] >> (2 * sh); mm = lmx[wi] >> sh;
                            if (sh) { x |= lw[wi + 1] << (64 - 2 * sh); mm |= lmx[wi + 1] << (32 - sh); }
                        } else {                                          // the window starts before the read: zeros below base 0
                            const uint32_t neg = (uint32_t)(-s0);
                            x = lw[0] << (2 * neg); mm = lmx[0] << neg;
                        }
                        w = pair_reverse(x); m = cf_brev32(mm);
                    } else { w = ~lw[k]; m = lmx[k]; }
                    if (have < 32) { w &= (1ull << (2 * have)) - 1; m &= (1u << have) - 1u; }
                    w &= ~spread_pairs(m);                                // an N carries code 0
                }
                fw_[k] = w; fm_[k] = m;
            }
            cf_compiler_fence();
#pragma unroll
            for (int k = 0; k < W; k++) { rw[k] = fw_[k]; rm[k] = fm_[k]; }
            }
            cf_compiler_fence();
            cur = 0; nhmx = 0; lz = (b.lazyHits & 5u) | ((aux >> 63) ? 2u : 0u);
            mode = S_CALL;
        } else if (mode == S_REC) {
            uint64_t *dst = reinterpret_cast<uint64_t *>(lrec + (size_t)sub * (RB / G));     // 8-byte aligned (odd stride)
#pragma unroll
            for (int i = 0; i < RCH; i++) { dst[2 * i] = sa.v[i].x; dst[2 * i + 1] = sa.v[i].y; }
            cf_compiler_fence();                     // the words / masks are read back below through other types
            static_assert(12 * W <= RB - 16, "words and masks must not reach into the meta chunk");
            if (sub == G - 1) lmeta[2] = item;       // same lane, after its 16-byte store of the chunk
            cf_compiler_fence();
            cur = 0; nhmx = 0; lz = (b.lazyHits & 5u) | 2u;
            mode = S_CALL;
        } else if (mode == S_FTABW) {
            const uint64_t size = wide_size(ft.x);
            if (size == kWideSizeMax) {                          // range too large for an entry: step by step from the 10-mer
                aux &= (1ull << (2 * ftc)) - 1;
                mode = S_FTAB;
                if (COUNT) cFtab++;
            } else if (size == 0) {                              // the 10-mer does not occur (the S_FTAB miss)
                push = true; pLen = ftc; cur += ftc;
            } else {                                             // the range the step-by-step path holds after D bases
                const uint32_t D = ftc + (uint32_t)((ft.x >> 40) & 15u);
                top = ft.x & ((1ull << 40) - 1); bot = top + size;
                dep = cur + D;
                const uint32_t L = lmeta[0];
                bool ends = D < wideChars || dep >= L;           // died at D + 1, or read end
                // what the entry knows about the bases to come.  more = kUnknownMore: nothing; else the call goes on for exactly
                // `more` further bases and ends there (every surviving row fails at the next base, or the read / an N stops it)
                constexpr uint32_t kUnknownMore = 0xffu;
                uint32_t more = kUnknownMore;
                if (!ends && wide_masked(ft.x) && ((lmx[dep >> 5] >> (dep & 31)) & 1u) == 0) {
                    const uint32_t code = (uint32_t)(ft.x >> 44) & 15u, d1 = dep + 1;
                    if (kWideCtxRows > 0 && code <= kWideCtxRows) {
                        // one or two rows with their CONTEXT (the bases that precede their suffixes): how far do they go on matching
                        // the read's next bases (search order, out of the strand record)?
                        const uint32_t nb = wide_ctx_bases(code);
                        const uint32_t k = dep >> 5, sh = dep & 31;
                        uint32_t q = (uint32_t)(lw[k] >> (2 * sh)), nm = lmx[k] >> sh;
                        if (sh > 24) { q |= (uint32_t)(lw[k + 1] << (64 - 2 * sh)); nm |= lmx[k + 1] << (32 - sh); }
                        uint32_t lim = L - dep < nb ? L - dep : nb;                       // bases that exist ...
                        nm &= (1u << nb) - 1u;
                        if (nm) { const uint32_t fn = (uint32_t)cf_ctz32(nm); if (fn < lim) lim = fn; }   // ... before the next N
                        const uint32_t pay = (uint32_t)(ft.x >> 48);
                        uint32_t mmax = 0, nmax = 0, rmax = 0;      // the longest of the rows' matches, how many rows reach it, the last of them
#pragma unroll
                        for (uint32_t r = 0; r < (kWideCtxRows ? (uint32_t)kWideCtxRows : 1u); r++) {
                            if (r >= code) continue;
                            uint32_t x = ((pay >> (2 * nb * r)) ^ q) & ((1u << (2 * nb)) - 1u);
                            x = (x | (x >> 1)) & 0x5555u;
                            uint32_t mr = x ? (uint32_t)cf_ctz32(x) >> 1 : nb;
                            if (mr > lim) mr = lim;
                            if (mr > mmax) { mmax = mr; nmax = 1; rmax = r; } else if (mr == mmax) nmax++;
                        }
                        // ONE row goes on beyond the context while the others — chance co-occurrences of the wide-mer — have each
                        // shown a difference: within these nb bases the range comes down to that row's image whatever is done, and
                        // nothing is recorded before then.  The chain follows that row alone from here: no pair steps to shake the
                        // others off, and (a single row) straight to the text where the row is sampled
                        if (mmax == nb && nmax == 1) { top += rmax; bot = top + 1; }
                        if (mmax < nb) more = mmax;              // (mmax == nb: the context is used up and a row still matches — unknown)
                    } else {
                        // the next-pairs mask: none of the rows is preceded by the read's next base -> the step would come back empty,
                        // the call ends here; it is, but not by that base and the one after it (or the read ends there, or an N
                        // follows) -> one further base, and there the call ends
                        const int c1 = (int)((lw[dep >> 5] >> (2 * (dep & 31))) & 3);
                        const uint32_t m4 = (uint32_t)(ft.x >> (48 + 4 * c1)) & 15u;
                        if (!m4) more = 0;
                        else if (d1 >= L || ((lmx[d1 >> 5] >> (d1 & 31)) & 1u) != 0 || !((m4 >> ((lw[d1 >> 5] >> (2 * (d1 & 31))) & 3)) & 1u)) more = 1;
                    }
                    if (more == 0) ends = true;
                    else if (more != kUnknownMore) {
                        // The call ends `more` bases on.  A hit nobody will read — the strand is still lazy, it holds its LZ hits
                        // already, the hit is shorter than minHitLen: emitHit below stores nothing for it — needs no range: its
                        // length is all the restart rule asks for, and the steps that would make the range are not taken.
                        if ((lz & 1u) && (nhmx & 0xffu) >= LZ && D + more < pr.m) {
                            push = true; pTop = top; pBot = bot; pLen = D + more; cur = dep + more;      // (the range is not the hit's: never stored)
                        } else {
                            // a hit that is kept: the steps are taken, but they stop where the entry says (no failing step); over the pair
                            // planes a single further base is stepped alone (vf bit 3: and there the call ends)
                            vf |= 16u; endDep = dep + more;
                            if (BLOCKS && ix.planes2 && more == 1) vf |= 8u;
                            mode = S_EXT;
                        }
                    } else mode = S_EXT;
                } else if (!ends) mode = S_EXT;
                if (ends) { push = true; pTop = top; pBot = bot; pLen = D; cur = dep; }
            }
        } else if (mode == S_FTAB) {
            top = ft.x <= ix.len ? ft.x : ix.eftab[(ft.x ^ kNone64) * 2 + 1];       // ftabHi bt2_idx.h:1880-1897
            bot = ft.y <= ix.len ? ft.y : ix.eftab[(ft.y ^ kNone64) * 2];           // ftabLo bt2_idx.h:1953-1970
            dep = cur + ftc;
            if (bot <= top) { push = true; pLen = ftc; cur = dep; }
            else if (dep >= lmeta[0]) { push = true; pTop = top; pBot = bot; pLen = dep - (nhmx >> 20); cur = dep; }
            else mode = S_EXT;
        } else if (mode == S_EXT || mode == S_EXTB) {
            bool stop = stepN;
            if (!stepN) {
                if (COUNT && mode == S_EXT) {
                    if (bot - top > 1) cPair++; else cSingle++;
                    if (!same) cPair2++;
                }
                // both counts on the one loaded side; the caller's state says which of them mean something
                if constexpr (!BLOCKS) rank_tab_build<G>(reinterpret_cast<const Side<G> &>(sa), pat32(c), scr);
                uint64_t t, bb;
                if constexpr (BLOCKS) {                          // entry = {bits, fchr[c] + occ before the group}
                    t = sa.v[0].y + popc_below(sa.v[0].x, oT);
                    bb = sa.v[0].y + popc_below(sa.v[0].x, oB);
                } else if constexpr (G == 2) {
                    // lane c>>1 of the pair owns occ[c] (chunks 6 | 7); partial = count (+ occ), summed over
                    // the pair with two DPP moves per 64-bit value
                    const uint64_t occ = sub == (c >> 1) ? ((c & 1) ? sa.v[8 / G - 1].y : sa.v[8 / G - 1].x) : 0ull;
                    uint64_t pT = rank_tab_count<G>(scr, oT) + occ;
                    uint64_t pB = rank_tab_count<G>(scr, oB) + occ;
                    pT += swap1_64(pT);
                    pB += swap1_64(pB);
                    t = pT; bb = pB;
                } else {
                    uint32_t acc = rank_tab_count<G>(scr, oT) | (rank_tab_count<G>(scr, oB) << 16);
                    acc = Grp<G>::sum(acc);
                    const uint64_t occ = side_occ<G>(reinterpret_cast<const Side<G> &>(sa), c);
                    t = occ + (acc & 0xffffu);
                    bb = occ + (acc >> 16);
                }
                if constexpr (!BLOCKS) {
                    if (c == 0 && sS == ix.zSide) {
                        if (ix.zIn < oT) t--;
                        if (ix.zIn < oB) bb--;
                    }
                    const uint64_t f = fchr_of(ix, c);
                    t += f; bb += f;
                }
                if (mode == S_EXT && !same) {                    // top side done; the bot side comes next iteration
                    aux = t;
                    mode = S_EXTB;
                } else {
                    if (mode == S_EXTB) { t = aux; mode = S_EXT; }
                    const bool pair = BLOCKS && (vf & 4u) != 0;
                    if (bb <= t) {
                        // an empty pair says nothing about its first base: that one alone next (vf bit 3), and whatever it gives is
                        // where the call ends — the base after it is known to fail
                        if (pair) vf = (vf & ~4u) | 8u;
                        else stop = true;
                    } else {
                        vf = bb - t == bot - top ? vf + (pair ? 0x200u : 0x100u) : (vf & 0xffu);    // steps in a row that kept the range's size (one row: single-row steps)
                        top = t; bot = bb; dep += pair ? 2u : 1u; stop = dep >= lmeta[0] || (vf & 8u) != 0 || ((vf & 16u) && dep == endDep);
                        vf &= ~4u;
                    }
                }
            }
            if (stop) { push = true; pTop = top; pBot = bot; pLen = dep - (nhmx >> 20); cur = dep; }
        }
        // the hit record {w0, w1} of `hlen` bases, the strand's (nhmx & 0xff)-th: to the pool, or held back (lazy hits).
        // true: the strand turned out to matter after hits were let go — the chain must search it again, direct
        auto emitHit = [&](uint64_t w0, uint64_t w1, uint32_t hlen) -> bool {
            const uint32_t j = nhmx & 0xffu;
            HitP *dst = b.hits + ((uint64_t)lmeta[1] + j);
            if (!(lz & 1u)) { if (sub == 0) cf_store16_stream(dst, w0, w1); return false; }
            if (hlen >= pr.m) {
                if (j > LZ) return true;
                if (sub == 0) {
#pragma unroll
                    for (uint32_t t = 0; t < LZ; t++) if (t < j) cf_store16_stream(dst - j + t, lhit[2 * t], lhit[2 * t + 1]);
                    cf_store16_stream(dst, w0, w1);
                }
                lz &= ~1u;
                return false;
            }
            if (j < LZ && sub == 0) { lhit[2 * j] = w0; lhit[2 * j + 1] = w1; }
            return false;
        };
        // a finished call: store the hit, then done / restart rule (classifier.h:686-766)
        if (push) {
            const uint32_t L = lmeta[0];
            const bool dummy = pTop == kNone64;              // HitP{top, size, bwoff, len, nelt = 0}
            const bool posf = !dummy && (pTop & kRowIsPos) != 0;        // (a hit in its position form: S_TXT)
            if (emitHit((dummy ? kHit40 : (pTop & kHit40)) | ((uint64_t)pLen << 40), (dummy ? 0ull : posf ? (1ull | kHitPosForm) : pBot - pTop) | ((uint64_t)(nhmx >> 20) << 40), pLen)) {
                cur = 0; nhmx = 0; lz &= ~1u; mode = S_CALL;    // once more from the strand's right end, every hit stored
            } else {
                { const uint32_t mx = (nhmx >> 8) & 0xfffu; nhmx = (nhmx & 0xfff000ffu) + 1u + ((pLen > mx ? pLen : mx) << 8); }
                bool done = cur >= L;
                if (!done) {
                    if (pLen > pr.inc) cur += 1;
                    done = cur + pr.m >= L;
                }
                if (done) { if (sub == 0) b.nhml[lmeta[2]] = nhml_make(nhmx & 0xffu, ((nhmx >> 8) & 0xfffu) >= pr.m); mode = S_IDLE; }
                else mode = S_CALL;
            }
        }
        // begin the next partialSearch call (no memory access; a dummy hit keeps the chain in S_CALL)
        if (mode == S_CALL) {
            const uint32_t L = lmeta[0];
            nhmx = (nhmx & 0xfffffu) | (cur << 20);
            vf = 0; mv = 0;
            uint32_t len = 0, newCur = 0;
            const int how = ps_begin2(lw, lmx, L, cur, ftc, wideChars, aux, len, newCur);
            if (how == 2) { mode = S_FTABW; if (COUNT) cFtabW++; }
            else if (how == 1) { mode = S_FTAB; if (COUNT) cFtab++; }
            else {
                // an unresolved hit of `len` bases
                if (emitHit(kHit40 | ((uint64_t)len << 40), (uint64_t)(nhmx >> 20) << 40, len)) {
                    cur = 0; nhmx = 0; lz &= ~1u;               // (stays in S_CALL: the strand starts over next iteration)
                } else {
                    { const uint32_t mx = (nhmx >> 8) & 0xfffu; nhmx = (nhmx & 0xfff000ffu) + 1u + ((len > mx ? len : mx) << 8); }
                    cur = newCur;
                    bool done = cur >= L;
                    if (!done) {
                        if (len > pr.inc) cur += 1;
                        done = cur + pr.m >= L;
                    }
                    if (done) { if (sub == 0) b.nhml[lmeta[2]] = nhml_make(nhmx & 0xffu, ((nhmx >> 8) & 0xfffu) >= pr.m); mode = S_IDLE; }
                }
            }
        }
        };
        struct ZeroMask { CF_DEV uint32_t operator[](uint32_t) const { return 0u; } };
        if (anyN) iter(lm); else iter(ZeroMask{});
    }
    if (COUNT && b.ops && sub == 0 && (cFtab | cPair | cSingle | cFtabW)) {
        cf_atomic_add(&b.ops->nFtabWide, cFtabW);
        cf_atomic_add(&b.ops->nVerify, cVerify); cf_atomic_add(&b.ops->nTextLoads, cText); cf_atomic_add(&b.ops->nPosHits, cPos);
        cf_atomic_add(&b.ops->nFtab, cFtab); cf_atomic_add(&b.ops->nPair, cPair);
        cf_atomic_add(&b.ops->nPair2, cPair2); cf_atomic_add(&b.ops->nSingle, cSingle);
    }
}

// A whole partialSearch call, used by k_post's extension step (one lane, a chain of dependent loads: what the queries with a
// long hit on both strands wait for).  With the derived tables at hand it takes the same short cuts as search2_body, each of
// which leaves the state the step-by-step path passes through: the wide ftab answers the call's first wideChars bases, the
// planes serve an LF step with one or two 16-byte loads, and a range that is down to ONE row on a row of the SA sample is
// finished against the text (see "Text verification" at search2_body) — ~15 dependent loads instead of one per base.
template <int G>
CF_DEV void ps_whole(const DIndex &ix, const DBatch &b, uint64_t wbase, uint32_t L, bool fw, uint32_t cur,
                     Hit &h) {
    ReadWin win{0, kNone64, 0};
    uint64_t top = kNone64, bot = kNone64;
    uint32_t dep = 0, len = 0, newCur = 0;
    bool usedFtab;
    h.bwoff = cur; h.nelt = 0;
    const uint32_t ftc = (uint32_t)ix.ftabChars, wc = (uint32_t)ix.wideChars;
    bool started = false;
    if (ix.wide && L - cur >= wc) {                              // the wide-mer, when its bases exist and hold no N
        uint64_t fi = 0;
        bool clean = true;
        for (uint32_t i = 0; i < wc && clean; i++) {
            const int c = strand_char(b, wbase, L, fw, L - cur - 1 - i, win);
            if (c > 3) clean = false; else fi |= (uint64_t)c << (2 * i);
        }
        if (clean) {
            const uint64_t e = ix.wide[fi], size = wide_size(e);
            if (size == 0) { h.top = h.bot = kNone64; h.len = ftc; return; }            // the 10-mer does not occur
            if (size != kWideSizeMax) {
                const uint32_t D = ftc + (uint32_t)((e >> 40) & 15u);
                top = e & ((1ull << 40) - 1); bot = top + size; dep = cur + D;
                if (D < wc) { h.top = top; h.bot = bot; h.len = D; return; }              // the range died at base D + 1
                started = true;
            }
        }
    }
    if (!started && !ps_begin(ix, b, wbase, L, fw, cur, win, top, bot, dep, len, newCur, usedFtab)) {
        h.top = h.bot = kNone64; h.len = len; return;
    }
    bool tried = false;                                          // text verification: once per call
    uint32_t run = 0;                                            // successful single-row steps in a row
    while (dep < L) {
        if (ix.posRate >= 0 && !tried && bot - top == 1 && run >= ix.verifyMinRun && (top & ((1ull << ix.posRate) - 1)) == 0 &&
            L - dep >= kVerifyMinLeft) {
            tried = true;
            const uint64_t p = trio_at(ix.saPos, top >> ix.posRate);      // the bases to come lie left of it in the text
            uint64_t tw = 0, twi = kNone64;
            uint32_t M = 0;
            while (dep + M < L && M < p) {
                const uint64_t pos = p - 1 - M;
                if ((pos >> 5) != twi) { twi = pos >> 5; tw = ix.text[twi]; }
                const int rc = strand_char(b, wbase, L, fw, L - (dep + M) - 1, win);
                if (rc > 3 || (uint64_t)rc != ((tw >> (2 * (pos & 31))) & 3)) break;
                M++;
            }
            const uint64_t pe = p - M, pm = (1ull << ix.isaRate) - 1, q = (pe + pm) & ~pm;
            if ((b.lazyHits & 4u) && (dep + M >= L || M < p)) {
                // the call ends where the match ended (the read's end, or a difference — not the start of the text): the hit is the
                // suffix at pe, handed back in its position form (HitP) — no inverse-sample read, no steps back from it
                h.top = kRowIsPos | pe; h.bot = h.top + 1; h.len = dep + M - cur;
                return;
            }
            if (M >= 4 && M >= q - pe) {                         // the state the step-by-step path has at q (row of that suffix, its depth)
                dep = dep + M - (uint32_t)(q - pe);
                top = trio_at(ix.isa, q >> ix.isaRate); bot = top + 1;
                continue;
            }
        }
        const int c = strand_char(b, wbase, L, fw, L - dep - 1, win);
        if (c > 3) break;
        uint64_t t, bb;
        { bool two; rank_any<G>(ix, c, top, bot, t, bb, two); }
        if (bb <= t) break;
        run = (bb - t == 1 && bot - top == 1) ? run + 1 : 0;
        top = t; bot = bb; dep++;
    }
    h.top = top; h.bot = bot; h.len = dep - cur;
}

// -------------------------------------------------- libstdc++ std::sort order
// hit._partialHits.sort(compareBWTHits()) is std::sort (ds.h:775-779): tie
// order between equivalent hits is whatever libstdc++'s introsort does, and it
// decides hit-map insertion order.  Restated from the algorithm's published
// structure (bits/stl_algo.h of g++ 11: threshold 16, depth limit 2*floor(log2 n),
// median-of-3 to first, unguarded Hoare partition, heapsort fallback, final
// insertion sort); checked against std::sort in tests/test_sort_order.py.
CF_DEV bool hit_less(const HitP &a, const HitP &b) {         // classifier.h:1058-1086
    const uint64_t as = hp_size(a), bs = hp_size(b);
    const uint64_t al = hp_len(a), bl = hp_len(b);
    if (al >= 22 || bl >= 22) {
        if (al >= 22 && bl >= 22) { if (as < bs) return true; if (as > bs) return false; }
        if (bl < al) return true;
        if (bl > al) return false;
    }
    if (bl * as < al * bs) return true;
    if (bl * as > al * bs) return false;
    if (as < bs) return true;
    if (as > bs) return false;
    if (bl < al) return true;
    return false;
}
CF_DEV void hit_swap(HitP &a, HitP &b) { const HitP t = a; a = b; b = t; }

CF_DEV void ss_linear_insert(HitP *a, int last) {
    const HitP val = a[last];
    int next = last - 1;
    while (hit_less(val, a[next])) { a[last] = a[next]; last = next; --next; }
    a[last] = val;
}
CF_DEV void ss_insertion(HitP *a, int first, int last) {
    if (first == last) return;
    for (int i = first + 1; i != last; ++i) {
        if (hit_less(a[i], a[first])) {
            const HitP val = a[i];
            for (int k = i; k > first; --k) a[k] = a[k - 1];
            a[first] = val;
        } else ss_linear_insert(a, i);
    }
}
CF_DEV void ss_push_heap(HitP *a, int first, int hole, int top, const HitP &val) {
    int parent = (hole - 1) / 2;
    while (hole > top && hit_less(a[first + parent], val)) {
        a[first + hole] = a[first + parent]; hole = parent; parent = (hole - 1) / 2;
    }
    a[first + hole] = val;
}
CF_DEV void ss_adjust_heap(HitP *a, int first, int hole, int len, const HitP &val) {
    const int top = hole;
    int child = hole;
    while (child < (len - 1) / 2) {
        child = 2 * (child + 1);
        if (hit_less(a[first + child], a[first + (child - 1)])) child--;
        a[first + hole] = a[first + child]; hole = child;
    }
    if ((len & 1) == 0 && child == (len - 2) / 2) {
        child = 2 * (child + 1);
        a[first + hole] = a[first + (child - 1)]; hole = child - 1;
    }
    ss_push_heap(a, first, hole, top, val);
}
CF_DEV void ss_heapsort(HitP *a, int first, int last) {
    const int len = last - first;
    if (len >= 2) {
        int parent = (len - 2) / 2;
        for (;;) { const HitP v = a[first + parent]; ss_adjust_heap(a, first, parent, len, v); if (parent == 0) break; parent--; }
    }
    while (last - first > 1) {
        --last;
        const HitP v = a[last]; a[last] = a[first];
        ss_adjust_heap(a, first, 0, last - first, v);
    }
}
CF_DEV void std_sort_hits(HitP *a, int n) {
    if (n <= 1) return;
    int lg = 0;
    for (int t = n; t > 1; t >>= 1) lg++;
    // introsort loop; the recursion on the right part becomes an explicit stack
    // (disjoint ranges, so the order in which they are finished does not matter)
    int stF[48], stL[48], stD[48], sp = 0;
    int first = 0, last = n, depth = 2 * lg;
    for (;;) {
        while (last - first > 16) {
            if (depth == 0) { ss_heapsort(a, first, last); break; }
            --depth;
            const int mid = first + (last - first) / 2;
            {   // median of (first+1, mid, last-1) to first
                const int x = first + 1, y = mid, z = last - 1;
                if (hit_less(a[x], a[y])) {
                    if (hit_less(a[y], a[z])) hit_swap(a[first], a[y]);
                    else if (hit_less(a[x], a[z])) hit_swap(a[first], a[z]);
                    else hit_swap(a[first], a[x]);
                } else if (hit_less(a[x], a[z])) hit_swap(a[first], a[x]);
                else if (hit_less(a[y], a[z])) hit_swap(a[first], a[z]);
                else hit_swap(a[first], a[y]);
            }
            int lo = first + 1, hi = last;
            for (;;) {                                   // unguarded partition around a[first]
                while (hit_less(a[lo], a[first])) ++lo;
                --hi;
                while (hit_less(a[first], a[hi])) --hi;
                if (!(lo < hi)) break;
                hit_swap(a[lo], a[hi]);
                ++lo;
            }
            if (sp < 48) { stF[sp] = lo; stL[sp] = last; stD[sp] = depth; sp++; }
            last = lo;
        }
        if (sp == 0) break;
        --sp; first = stF[sp]; last = stL[sp]; depth = stD[sp];
    }
    if (n > 16) {
        ss_insertion(a, 0, 16);
        for (int i = 16; i != n; ++i) ss_linear_insert(a, i);
    } else ss_insertion(a, 0, n);
}

// -------------------------------------------------------------------- post
CF_DEV void hit_reset(HitP &h) { Hit z; z.top = z.bot = 0; z.bwoff = kNone32; z.len = 0; z.nelt = 0; h = hit_pack(z); }   // hi_aligner.h:63-71

// trim overlaps inside one strand (classifier.h:873-895)
CF_DEV void post_trim(HitP *h, uint32_t n) {
    if (n < 2) return;
    for (uint32_t i = 0; i + 1 < n; i++) {
        for (uint32_t j = i + 1; j < n; j++) {
            const uint32_t bi = hp_bwoff(h[i]), bj = hp_bwoff(h[j]), li = hp_len(h[i]), lj = hp_len(h[j]);
            if (bi >= bj) { hp_set_len(h[i], 0); break; }
            if ((uint64_t)bi + li <= bj) break;
            if (li >= lj) {
                const uint32_t e = bj + lj;
                hp_set_bwoff(h[j], bi + li); hp_set_len(h[j], e - (bi + li));
            } else hp_set_len(h[i], bj - bi);
        }
    }
}

// extend / twin removal / trim for one mate (classifier.h:790-895)
// hs / n: the mate's two hit lists — in the hit pool, or the copies post_body keeps in its lane's scratch (LDS)
CF_DEV void post_fix(const DIndex &ix, const DParams &pr, const DBatch &b, uint32_t rd, HitP *const hs[2], const uint32_t n[2]) {
    const uint64_t wbase = b.woff[rd], m = pr.m;
    const uint32_t L = b.rlen[rd];
    // sum[fwi] of classifier.h:663-725: lengths of the hits >= minHitLen as they were pushed
    uint64_t sum[2] = {0, 0};
    for (int f = 0; f < 2; f++) for (uint32_t i = 0; i < n[f]; i++) if (hp_len(hs[f][i]) >= m) sum[f] += hp_len(hs[f][i]);
    if (sum[0] >= m && sum[1] >= m) {
        // extend overlapping fw / rc hits (classifier.h:790-847)
        for (uint32_t i = 0; i < n[0]; i++) {
            HitP &hit = hs[0][i];
            const uint64_t len = hp_len(hit), l = hp_bwoff(hit), r = l + len;
            for (uint32_t j = 0; j < n[1]; j++) {
                HitP &rc = hs[1][j];
                const uint64_t rclen = hp_len(rc);
                if (len < m && rclen < m) continue;
                const uint64_t rc_l = (uint64_t)L - hp_bwoff(rc) - rclen, rc_r = rc_l + rclen;
                if (r <= rc_l) continue;
                if (rc_r <= l) continue;
                if (l == rc_l && r == rc_r) continue;
                if (l < rc_l && r > rc_r) continue;
                if (l > rc_l && r < rc_r) continue;
                if (l > rc_l) {
                    Hit t; ps_whole<1>(ix, b, wbase, L, true, (uint32_t)rc_l, t);
                    if (t.len == len + l - rc_l) hit = hit_pack(t);
                }
                if (r > rc_r) {
                    Hit t; ps_whole<1>(ix, b, wbase, L, false, (uint32_t)(L - r), t);
                    if (t.len == rclen + r - rc_r) rc = hit_pack(t);
                }
            }
        }
        // drop fw/rc twins that map too often (classifier.h:849-870)
        for (uint32_t i = 0; i < n[0]; i++) {
            HitP &hit = hs[0][i];
            const uint64_t len = hp_len(hit), l = hp_bwoff(hit), r = l + len;
            for (uint32_t j = 0; j < n[1]; j++) {
                HitP &rc = hs[1][j];
                const uint64_t rclen = hp_len(rc), rc_l = (uint64_t)L - hp_bwoff(rc) - rclen, rc_r = rc_l + rclen;
                if (rc_l < l) break;
                if (len != rclen) continue;
                if (l == rc_l && r == rc_r && hp_size(hit) + hp_size(rc) > pr.ihits) {
                    hit_reset(hit); hit_reset(rc); break;
                }
            }
        }
    }
    post_trim(hs[0], n[0]);
    post_trim(hs[1], n[1]);
}
CF_DEV void post_fix(const DIndex &ix, const DParams &pr, const DBatch &b, uint32_t rd) {      // in place (the debug tap)
    const uint32_t slot = b.slotOf[rd];
    HitP *hs[2] = {b.hits + b.hitBase[rd], b.hits + b.hitBase[rd] + b.hitCap[rd]};
    const uint32_t n[2] = {nhml_n(b.nhml[2 * slot]), nhml_n(b.nhml[2 * slot + 1])};
    post_fix(ix, pr, b, rd, hs, n);
}

// n hit records from src to dst, four loads in flight at a time (a plain loop waits for every record before it asks for the next)
CF_DEV void hits_copy(HitP *dst, const HitP *src, uint32_t n) {
    uint32_t i = 0;
    for (; i + 4 <= n; i += 4) { const HitP a = src[i], b2 = src[i + 1], c = src[i + 2], d = src[i + 3]; dst[i] = a; dst[i + 1] = b2; dst[i + 2] = c; dst[i + 3] = d; }
    for (; i < n; i++) dst[i] = src[i];
}

// scratch / scratchCap (round 6): room for scratchCap hit records that is this lane's own — LDS on the device.  The general kernel
// lives on the latency of dependent accesses: extension, twin removal, trim, the strand choice, std::sort and the plan go over a
// mate's hit lists again and again (hundreds of 16-byte reads and writes, each the next one's condition), and in the hit pool
// every one of them is a trip to L2 or HBM.  A mate whose lists fit is worked on in the scratch — in with a few loads in flight,
// the chosen strands' lists back out at the end (emit_body and score_body read them there) — and only ps_whole's chain goes
// to memory.  nullptr / lists that do not fit: in place, as before.
CF_DEV void post_body(const DIndex &ix, const DParams &pr, const DBatch &b, uint32_t q, HitP *scratch = nullptr, uint32_t scratchCap = 0) {
    if (b.st->flags & kStHitsOverflow) { b.qRows[q] = 0; return; }   // nothing was searched; the host re-runs the batch with a larger pool
    QHead qi;
    for (int a = 0; a < 2; a++) { qi.lo[a] = qi.hi[a] = 0; qi.nProc[a][0] = qi.nProc[a][1] = 0; qi.brk[a] = 0; qi.pad2[a] = 0; qi.maxG[a][0] = qi.maxG[a][1] = 0; }
    uint32_t nPlanned = 0;
    const uint32_t r0 = b.paired ? 2 * q : q;
    const bool p0 = b.pass[r0] != 0, p1 = b.paired ? b.pass[r0 + 1] != 0 : false;
    uint32_t rds[2] = {r0, r0 + 1};
    int nm;
    uint32_t isPaired = 0, firstMate = 0;
    if (b.paired && p0 && p1) { nm = 2; isPaired = 1; }           // centrifuge.cpp:2678-2690
    else if (p0) nm = 1;
    else if (p1) { nm = 1; rds[0] = r0 + 1; firstMate = 1; }
    else nm = 0;
    const uint64_t k = pr.k, m = pr.m;
    uint64_t maxG = k;                                            // classifier.h:228
    uint32_t rowsTotal = 0;
    uint32_t tsBase = 0;                                          // the scoring loop's time stamp (classifier.h:232) at the head of a list
    bool tsWide = false;
    for (int rdi = 0; rdi < nm; rdi++) {
        const uint32_t rd = rds[rdi], slot = b.slotOf[rd];
        HitP *const pool[2] = {b.hits + b.hitBase[rd], b.hits + b.hitBase[rd] + b.hitCap[rd]};
        HitP *hs[2] = {pool[0], pool[1]};
        const uint32_t hm0 = b.nhml[2 * slot], hm1 = b.nhml[2 * slot + 1];
        const uint32_t n[2] = {nhml_n(hm0), nhml_n(hm1)};
        // A strand whose longest hit is below minHitLen cannot score, cannot trigger the cross-strand
        // extension / twin removal (both need >= minHitLen on BOTH strands, classifier.h:790) and loses
        // the strand choice; trimming only ever shortens hits.  So: neither strand long -> the mate
        // contributes nothing; one strand long -> only that strand's list is read, trimmed and planned.
        const bool long0 = nhml_long(hm0), long1 = nhml_long(hm1);
        if (!long0 && !long1) continue;
        // the lists that will be read (a strand without a long hit is not) into the lane's scratch when they fit
        const uint32_t need = (long0 ? n[0] : 0u) + (long1 ? n[1] : 0u);
        const bool staged = scratch != nullptr && need <= scratchCap;
        if (staged) {
            hs[0] = scratch; hs[1] = scratch + (long0 ? n[0] : 0u);
            if (long0) hits_copy(hs[0], pool[0], n[0]);
            if (long1) hits_copy(hs[1], pool[1], n[1]);
        }
        if (long0 && long1) post_fix(ix, pr, b, rd, hs, n);
        else post_trim(hs[long0 ? 0 : 1], n[long0 ? 0 : 1]);
        // strand choice (classifier.h:898-941)
        uint64_t tot[2] = {0, 0}, mx[2] = {0, 0};
        for (int f = 0; f < 2; f++) if (f == 0 ? long0 : long1) for (uint32_t i = 0; i < n[f]; i++) {
            const uint64_t len = hp_len(hs[f][i]);
            if (len < m) continue;
            tot[f] += (len - 15) * (len - 15);
            if (len > mx[f]) mx[f] = len;
        }
        int lo, hi;
        if (tot[0] != tot[1]) { lo = tot[0] > tot[1] ? 0 : 1; hi = lo + 1; }
        else if (mx[0] != mx[1]) { lo = mx[0] > mx[1] ? 0 : 1; hi = lo + 1; }
        else { lo = 0; hi = 2; }
        qi.lo[rdi] = (uint8_t)lo; qi.hi[rdi] = (uint8_t)hi;
        for (int f = lo; f < hi; f++) {
            HitP *h = hs[f];
            for (uint32_t i = 0; i < n[f]; i++)                      // classifier.h:253-265
                if (hp_len(h[i]) >= m && hp_size(h[i]) > maxG) maxG = hp_size(h[i]);
            if (maxG > k) maxG += k;
            qi.maxG[rdi][f] = maxG > 0xffffffffull ? 0xffffffffu : (uint32_t)maxG;
            std_sort_hits(h, (int)n[f]);                             // classifier.h:267
            uint64_t cnt = 0;
            uint32_t i = 0;
            for (; i < n[f]; i++) {                                  // classifier.h:270-372, plan only
                const uint64_t len = hp_len(h[i]), size = hp_size(h[i]);
                const uint64_t nelt = plan_nelt(len, size, maxG, m, pr.ihits);   // (emit / score get the same number from QHead::maxG)
                if (nelt == 0) continue;
                if (nPlanned < kInlinePlan) {                            // straight into the query's record
                    PlanHit ph;
                    ph.top = hp_row(h[i]); ph.nelt = (uint32_t)nelt; ph.meta = plan_meta((uint32_t)len, rdi, f, tsBase + i);
                    if (tsBase + i > kPlanTsMax || len > 0xffffu) tsWide = true;
                    b.qplan[(uint64_t)nPlanned * b.qplanStride + q] = ph;
                }
                nPlanned++;
                rowsTotal += (uint32_t)nelt;
                cnt += nelt;
                if (cnt >= maxG) { i++; qi.brk[rdi] |= (uint8_t)(1u << f); break; }   // :366
            }
            qi.nProc[rdi][f] = i;
            // the iteration that left through `break` did not run ts++ (classifier.h:366-367)
            tsBase += i - ((qi.brk[rdi] >> f) & 1u);
            if (staged) hits_copy(pool[f], h, n[f]);                 // the sorted list where emit_body / score_body look for it
        }
    }
    const uint32_t nPlan = (nPlanned <= kInlinePlan && !tsWide) ? nPlanned : kPlanNotInline;
    b.qflag[q] = qf_make(nPlan, isPaired, firstMate, (uint32_t)nm);
    if (nPlan == kPlanNotInline) b.qhead[q] = qi;                // (inline plans are all k_emit and the score kernels read)
    b.qRows[q] = rowsTotal;
}

// A wavefront's lanes append the queries they leave to the general kernel to a list: one atomic per wavefront.  Every lane of
// the wavefront must make the call.
CF_DEV void defer_push(uint32_t *list, uint32_t *counter, bool defer, uint32_t q) {
    const uint64_t mask = cf_ballot(defer);
    if (!mask) return;
    const uint32_t lane = cf_lane();
    const int leader = cf_ctz64(mask);
    uint32_t base = 0;
    if ((int)lane == leader) base = cf_atomic_add(counter, (uint32_t)cf_popc64(mask));
    base = cf_shfl(base, leader);
    if (defer) list[base + (uint32_t)cf_popc64(mask & ((1ull << lane) - 1))] = q;
}

// post_body for the common query, in registers: every mate that takes part has ONE strand with a hit of minHitLen (the other
// strand cannot score, extend or win the strand choice), at most kPostFastHits hits on it, and the query plans at most
// kInlinePlan hits.  Such a strand's hits come straight from the search — ascending, non-overlapping offsets (checked) — so
// post_trim changes nothing, the strand wins the choice (classifier.h:898-941), and what is left is the order of
// std::sort (:267) and the row plan (:253-299,366).  Up to 16 hits std::sort IS its insertion sort (ds.h:775, threshold
// 16), i.e. the stable order under compareBWTHits: the rank of a hit = the hits that are less + the equivalent ones before it,
// one comparison per pair, no data-dependent moves.  The list is loaded with kPostFastHits independent 16-byte loads (the
// general kernel chases it load by dependent load) and is NOT written back: k_emit and the score kernels take a query's
// planned hits (DBatch::qflag / qplan).  Returns true when the query is left to post_body, having written nothing.
constexpr int kPostFastHits = 8;
CF_DEV bool post_fast_body(const DIndex &ix, const DParams &pr, const DBatch &b, uint32_t q) {
    (void)ix;
    if (b.st->flags & kStHitsOverflow) { b.qRows[q] = 0; return false; }   // nothing was searched; the host re-runs the batch with a larger pool
    // (the per-mate results are kept in scalars and put into the record at the end: an array indexed by the mate would be
    // moved to LDS by the compiler, and the kernel is to run beside a search kernel that has the LDS to itself)
    uint32_t fOf0 = 2, fOf1 = 2, np0 = 0, np1 = 0, brk0 = 0, brk1 = 0;          // strand (2 = none), hits visited, left through break
    uint8_t isPaired = 0, firstMate = 0;
    const uint32_t r0 = b.paired ? 2 * q : q;
    const bool p0 = b.pass[r0] != 0, p1 = b.paired ? b.pass[r0 + 1] != 0 : false;
    uint32_t rd0 = r0;
    int nm;
    if (b.paired && p0 && p1) { nm = 2; isPaired = 1; }          // centrifuge.cpp:2678-2690
    else if (p0) nm = 1;
    else if (p1) { nm = 1; rd0 = r0 + 1; firstMate = 1; }
    else nm = 0;
    const uint64_t k = pr.k, m = pr.m;
    uint64_t maxG = k;                                            // classifier.h:228 (carried over the mates)
    uint32_t rowsTotal = 0, tsBase = 0, nPlanned = 0;
    PlanHit pl0{0, 0, 0}, pl1{0, 0, 0}, pl2{0, 0, 0}, pl3{0, 0, 0};
    static_assert(kInlinePlan == 4, "four plan slots below");
    bool defer = false;
    for (int rdi = 0; rdi < nm && !defer; rdi++) {
        const uint32_t rd = rd0 + (uint32_t)rdi, slot = b.slotOf[rd];
        const uint32_t hm0 = b.nhml[2 * slot], hm1 = b.nhml[2 * slot + 1];
        const bool long0 = nhml_long(hm0), long1 = nhml_long(hm1);
        if (!long0 && !long1) continue;                           // the mate contributes nothing
        if (long0 && long1) { defer = true; break; }              // cross-strand extension / twin removal: post_fix
        const int f = long0 ? 0 : 1;
        const uint32_t n = nhml_n(f ? hm1 : hm0);
        if (n > (uint32_t)kPostFastHits) { defer = true; break; }
        const uint8_t *hp = reinterpret_cast<const uint8_t *>(b.hits + b.hitBase[rd] + (f ? b.hitCap[rd] : 0u));
        uint64_t w0[kPostFastHits], w1[kPostFastHits];
#pragma unroll
        for (int i = 0; i < kPostFastHits; i++) {
            u64x2 v{0, 0};
            if ((uint32_t)i < n) v = cf_load16(hp + 16 * i);
            w0[i] = v.x; w1[i] = v.y;
        }
        bool apart = true;                                        // post_trim is a no-op exactly when neighbours do not overlap
        uint64_t mg = maxG;
#pragma unroll
        for (int i = 0; i < kPostFastHits; i++) {
            if ((uint32_t)i >= n) continue;
            const HitP hi{w0[i], w1[i]};
            const uint64_t len = hp_len(hi), size = hp_size(hi);
            if (len >= m && size > mg) mg = size;                 // classifier.h:253-265
            if (i + 1 < kPostFastHits && (uint32_t)(i + 1) < n) {
                const HitP hj{w0[i + 1], w1[i + 1]};
                const uint64_t bi = hp_bwoff(hi), bj = hp_bwoff(hj);
                if (bi >= bj || bi + len > bj) apart = false;
            }
        }
        if (!apart) { defer = true; break; }
        maxG = mg;
        if (maxG > k) maxG += k;
        uint32_t rank[kPostFastHits];
#pragma unroll
        for (int i = 0; i < kPostFastHits; i++) rank[i] = 0;
#pragma unroll
        for (int i = 0; i < kPostFastHits; i++) {
#pragma unroll
            for (int j = i + 1; j < kPostFastHits; j++) {
                if ((uint32_t)j < n) {                            // the later one goes first only when it is less
                    const bool jFirst = hit_less(HitP{w0[j], w1[j]}, HitP{w0[i], w1[i]});
                    rank[i] += jFirst ? 1u : 0u;
                    rank[j] += jFirst ? 0u : 1u;
                }
            }
        }
        uint64_t cnt = 0;
        uint32_t visited = n;
        bool stopped = false, brk = false;
#pragma unroll
        for (int r = 0; r < kPostFastHits; r++) {                 // classifier.h:270-372, plan only, in sorted order
            if ((uint32_t)r >= n || stopped) continue;
            uint64_t c0 = 0, c1 = 0;
#pragma unroll
            for (int i = 0; i < kPostFastHits; i++) if ((uint32_t)i < n && rank[i] == (uint32_t)r) { c0 = w0[i]; c1 = w1[i]; }
            const HitP c{c0, c1};
            const uint64_t len = hp_len(c), size = hp_size(c);
            uint64_t nelt = 0;
            if (!(len <= m || size == 0)) {
                nelt = size < maxG ? size : maxG;                 // getGenomeIdx classifier.h:592-593
                if (nelt > pr.ihits) nelt = 0;                    // :299
            }
            if (nelt == 0) continue;
            const PlanHit ph{hp_row(c), (uint32_t)nelt, plan_meta((uint32_t)len, rdi, f, tsBase + (uint32_t)r)};
            if (tsBase + (uint32_t)r > kPlanTsMax || len > 0xffffu) defer = true;      // (a PlanHit holds 16 bits of length: the general kernel)
            if (nPlanned == 0) pl0 = ph; else if (nPlanned == 1) pl1 = ph; else if (nPlanned == 2) pl2 = ph; else if (nPlanned == 3) pl3 = ph;
            nPlanned++;
            rowsTotal += (uint32_t)nelt;
            cnt += nelt;
            if (cnt >= maxG) { visited = (uint32_t)r + 1; brk = true; stopped = true; }   // :366
        }
        if (rdi == 0) { fOf0 = (uint32_t)f; np0 = visited; brk0 = brk ? 1u : 0u; }
        else { fOf1 = (uint32_t)f; np1 = visited; brk1 = brk ? 1u : 0u; }
        tsBase += visited - (brk ? 1u : 0u);                      // the iteration that left through `break` did not run ts++
    }
    if (defer || nPlanned > kInlinePlan) return true;
    // (the record of the general paths — strands, hits visited, breaks — is not needed: the plan is inline)
    (void)np0; (void)np1; (void)brk0; (void)brk1; (void)fOf0; (void)fOf1;
    b.qflag[q] = qf_make(nPlanned, isPaired, firstMate, (uint32_t)nm);
    if (nPlanned > 0) b.qplan[q] = pl0;
    if (nPlanned > 1) b.qplan[b.qplanStride + q] = pl1;
    if (nPlanned > 2) b.qplan[2 * b.qplanStride + q] = pl2;
    if (nPlanned > 3) b.qplan[3 * b.qplanStride + q] = pl3;
    b.qRows[q] = rowsTotal;
    return false;
}

// The row window of a pass: queries [qLo, qHi) whose planned rows fit the row workspace together.  Normally one
// pass covers the batch; a batch that plans more rows than the workspace holds (repeat-rich reads: up to
// ihits rows per hit) is finished in further passes, each started by the host with the previous qHi.
// One thread.
CF_DEV void row_window_body(const DBatch &b, uint32_t qLo, bool keepSlow = false) {
    BatchStatus &st = *b.st;
    const uint32_t nq = b.nQueries;
    st.rowsTotal = b.qBase[nq];
    st.needRows = 0;
    if (!keepSlow) st.nSlowScore = 0;                            // the pass's list of queries for the general score kernel (keepSlow: the
                                                                 // early score kernel has made it already, for every query of the batch)
    if (st.flags & kStHitsOverflow) { st.qLo = st.qHi = 0; st.rowLo = st.rowHi = 0; return; }
    if (qLo > nq) qLo = nq;
    const uint64_t rowLo = b.qBase[qLo];
    // largest qHi in [qLo, nq] with qBase[qHi] - rowLo <= rowsCap
    uint32_t lo = qLo, hi = nq;
    while (lo < hi) {
        const uint32_t md = lo + (hi - lo + 1) / 2;
        if (b.qBase[md] - rowLo <= b.rowsCap) lo = md; else hi = md - 1;
    }
    if (lo == qLo && qLo < nq) st.needRows = b.qBase[qLo + 1] - rowLo;      // this query alone does not fit
    st.qLo = qLo; st.qHi = lo;
    st.rowLo = rowLo; st.rowHi = b.qBase[lo];
}

// rows of every planned hit, in query order: put(index in the pass's row workspace, row)
template <typename Put>
CF_DEV void for_each_planned_row(const DParams &pr, const DBatch &b, uint32_t q, Put put) {
    if (q < b.st->qLo || q >= b.st->qHi) return;
    const uint32_t nRows = b.qRows[q];
    if (nRows == 0) return;
    const uint32_t qf = b.qflag[q], nPlan = qf_nplan(qf);
    const uint64_t base = b.qBase[q] - b.st->rowLo;
    uint32_t rowoff = 0;                                 // rows of the hits before this one, in the order k_post planned them
    if (nPlan != kPlanNotInline) {
#pragma unroll
        for (uint32_t j = 0; j < kInlinePlan; j++) {
            if (j >= nPlan) continue;
            const PlanHit ph = b.qplan[(uint64_t)j * b.qplanStride + q];
            for (uint32_t e = 0; e < ph.nelt; e++) put(base + rowoff + e, ph.top + e);
            rowoff += ph.nelt;
        }
        return;
    }
    const QHead *qp = &b.qhead[q];                       // (read field by field: a local copy indexed by mate / strand would go to LDS)
    const uint32_t r0 = (b.paired ? 2 * q : q) + qf_first(qf);
    const int nMates = qf_nmates(qf);
    for (int rdi = 0; rdi < nMates; rdi++) {
        const uint32_t rd = r0 + rdi;
        const int lo = qp->lo[rdi], hi = qp->hi[rdi];
        for (int f = lo; f < hi; f++) {
            const HitP *h = b.hits + b.hitBase[rd] + (f ? b.hitCap[rd] : 0u);
            const uint32_t np = qp->nProc[rdi][f];
            const uint64_t mg = qp->maxG[rdi][f];
            for (uint32_t i = 0; i < np; i++) {
                const HitP hp = h[i];
                const uint32_t ne = plan_nelt(hp_len(hp), hp_size(hp), mg, pr.m, pr.ihits);
                for (uint32_t e = 0; e < ne; e++) put(base + rowoff + e, hp_row(hp) + e);
                rowoff += ne;
            }
        }
    }
}
CF_DEV void emit_body(const DParams &pr, const DBatch &b, uint32_t q) {
    for_each_planned_row(pr, b, q, [&](uint64_t at, uint64_t row) { b.rowVal[at] = row; });
}

// -------------------------------------------------------------------- walk
// bt2_idx.h:1980-2014 tryOffset, minus the LF step
CF_DEV bool try_offset(const DIndex &ix, uint64_t row, uint32_t &ref) {
    if (row == ix.zOff) { ref = 0; return true; }
    if ((row & ((1ull << ix.offRate) - 1)) == 0) {
        const uint64_t e = row >> ix.offRate;
        ref = ix.offw ? static_cast<const uint32_t *>(ix.offs)[e] : static_cast<const uint16_t *>(ix.offs)[e];
        return true;
    }
    if (ix.lastBoundary > 0 && row <= ix.lastBoundary) {
        const uint64_t blk = row >> ix.boundShift;
        if ((ix.boundBits[blk >> 5] >> (blk & 31)) & 1u) {
            uint32_t lo = 0, hi = ix.nBound;
            while (lo < hi) { const uint32_t md = (lo + hi) >> 1; if (ix.boundRow[md] < row) lo = md + 1; else hi = md; }
            if (lo < ix.nBound && ix.boundRow[lo] == row) {
                ref = ix.offw ? ix.boundRef[lo] : (ix.boundRef[lo] & 0xffffu);
                return true;
            }
        }
    }
    return false;
}

// The walk: like search2_body, one block of loads per iteration for every chain —
// the row to resolve, an SA-sample entry, or {side, own BWT byte, boundary prefilter word} of a
// walk-left step (issued together, the side speculatively) — then ALU-only processing.
//
// The loop "while (tryOffset(row) fails) row = LF(row)" has no memory: what it returns depends on the row it is at, not on
// how it got there.  So its answer can be tabulated for more rows than the file's sample holds (every 16th): at load time
// this very kernel, in its table-building modes, walks from every 2^walkRate-th row with the file's sample and stores
// where it ends; the batch walks then stop at the first row of that denser table — on average 2^walkRate - 1 steps
// instead of 2^offRate - 1 (15) — and read the same reference index the full walk would have reached.
enum : int { W_IDLE = 0, W_FETCH = 1, W_STEP = 2, W_SAMPLE = 3 };
enum : int { WALK_BATCH = 0, WALK_TABLE16 = 1, WALK_TABLE32 = 2 };    // rows of a batch -> rowRef | every 2^genShift-th row -> u16 / u32 table

template <int G, bool COUNT, int MODE = WALK_BATCH>
CF_DEV void walk2_body(const DIndex &ix, const DBatch &b) {
    const int sub = Grp<G>::sub();
    const uint32_t lane = cf_lane();
    const uint32_t leaderLane = lane & ~(uint32_t)(G - 1);
    int mode = W_IDLE;
    uint64_t row = 0, item = 0;
    uint64_t wnext = 0, wend = 0;
    bool exhausted = false;
    unsigned long long cWalk = 0;
    uint32_t stepsHere = 0, stepsMax = 0;                        // table modes: steps of the walk under way, longest walk of this lane (DBatch::walkMaxOut)
    const uint64_t total = b.st->rowHi - b.st->rowLo;            // rows of this pass (row_window_body)
    const uint64_t sampleMask = (1ull << ix.walkRate) - 1;
    auto put = [&](uint64_t it, uint32_t ref) {                  // the answer for work item `it`
        if (MODE == WALK_TABLE16) reinterpret_cast<uint16_t *>(b.rowRef)[it] = (uint16_t)ref;
        else b.rowRef[it] = ref;
    };
    // where a row goes next (tryOffset's order, bt2_idx.h:1980-2014): '$' row -> reference 0,
    // sampled row -> read the sample, anything else -> a walk-left step (with the boundary check)
    auto classify = [&](uint64_t r) {
        if (r == ix.zOff) { if (sub == 0) put(item, 0); mode = W_IDLE; }
        else mode = (r & sampleMask) == 0 ? W_SAMPLE : W_STEP;
    };
    for (;;) {
        const uint64_t idleMask = cf_ballot(mode == W_IDLE && sub == 0);
        if (idleMask) {
            if (wnext >= wend && !exhausted) {
                uint64_t base = 0;
                if (lane == 0) base = cf_atomic_add(&b.cursor[1], (unsigned long long)kSearchChunk);
                base = ((uint64_t)cf_first_lane_u32((uint32_t)(base >> 32)) << 32) | cf_first_lane_u32((uint32_t)base);
                if (base >= total) { exhausted = true; wnext = wend = 0; }
                else { wnext = base; wend = base + kSearchChunk < total ? base + kSearchChunk : total; }
            }
            const uint64_t avail = wend - wnext;
            const uint32_t nIdle = (uint32_t)cf_popc64(idleMask);
            if (mode == W_IDLE) {
                const uint32_t rnk = (uint32_t)cf_popc64(idleMask & ((1ull << leaderLane) - 1));
                if (rnk < avail) { item = wnext + rnk; mode = W_FETCH; }
            }
            wnext += nIdle < avail ? nIdle : avail;
        }
        if (cf_ballot(mode != W_IDLE) == 0) {
            if (exhausted) break;
            continue;
        }
        // ---- loads
        uint64_t rv = 0, sS = 0;
        uint32_t o = 0, bits = 0, samp = 0, own = 0;
        bool chk = false;
        Side<G> sd;
        u64x2 pe0{0, 0}, pe1{0, 0}, pe2{0, 0}, pe3{0, 0};
        if (mode == W_FETCH) rv = MODE == WALK_BATCH ? b.rowVal[item] : item << b.genShift;
        else if (mode == W_SAMPLE) {
            const uint64_t e = row >> ix.walkRate;
            samp = ix.offw ? static_cast<const uint32_t *>(ix.walkOffs)[e] : static_cast<const uint16_t *>(ix.walkOffs)[e];
        } else if (mode == W_STEP) {
            if (ix.sides) {
                sS = side_of(ix, row);
                o = (uint32_t)(row - sS * kSideChars);
                const uint8_t *p = ix.sides + sS * 128;
                side_load<G>(sd, p);
                own = p[o >> 2];                              // same 128-byte line as the side
            } else {                                          // (the sides were dropped: the four plane entries of the row's group, one line)
                const uint8_t *p = ix.planes + (row >> 6) * 64;
                pe0 = cf_load16(p); pe1 = cf_load16(p + 16); pe2 = cf_load16(p + 32); pe3 = cf_load16(p + 48);
            }
            chk = ix.lastBoundary > 0 && row <= ix.lastBoundary;
            if (chk) { const uint64_t blk = row >> ix.boundShift; bits = (ix.boundBits[blk >> 5] >> (blk & 31)) & 1u; }
        }
        // ---- processing
        if (mode == W_FETCH) { row = rv; stepsHere = 0; classify(row); }
        else if (mode == W_SAMPLE) { if (sub == 0) put(item, samp); mode = W_IDLE; }
        else if (mode == W_STEP) {
            bool resolved = false;
            if (chk && bits) {                                // rare: the row may be a genome-boundary row (.4.cf)
                uint32_t lo = 0, hi = ix.nBound;
                while (lo < hi) { const uint32_t md = (lo + hi) >> 1; if (ix.boundRow[md] < row) lo = md + 1; else hi = md; }
                if (lo < ix.nBound && ix.boundRow[lo] == row) {
                    if (sub == 0) put(item, ix.offw ? ix.boundRef[lo] : (ix.boundRef[lo] & 0xffffu));
                    resolved = true; mode = W_IDLE;
                }
            }
            if (!resolved && !ix.sides) {                     // the character is the one whose bit is set at the row
                const uint32_t ob = (uint32_t)row & 63u;
                const u64x2 e = ((pe0.x >> ob) & 1) ? pe0 : ((pe1.x >> ob) & 1) ? pe1 : ((pe2.x >> ob) & 1) ? pe2 : pe3;
                row = e.y + popc_below(e.x, ob);
                if (COUNT) cWalk++;
                if (MODE != WALK_BATCH) { stepsHere++; if (stepsHere > stepsMax) stepsMax = stepsHere; }
                classify(row);
            } else if (!resolved) {                           // row = LF(row, bwt[row]) (bt2_idx.h:2941-2963)
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
                if (COUNT) cWalk++;
                if (MODE != WALK_BATCH) { stepsHere++; if (stepsHere > stepsMax) stepsMax = stepsHere; }
                classify(row);
            }
        }
    }
    if (COUNT && b.ops && sub == 0 && cWalk) cf_atomic_add(&b.ops->nWalk, cWalk);
    if (MODE != WALK_BATCH && b.walkMaxOut && stepsMax) cf_atomic_max(b.walkMaxOut, stepsMax);
}

// The batch walk, one LANE per row: with the dense resolve table a row is 0-3 LF steps away from its answer (most are 0-1), so
// the chain machinery of walk2_body (work queue, two lanes per chain, one state per iteration) costs more than the walk —
// 2.8 ms per 14 M rows of which the steps were a fraction.  Here a lane loads its row, loops "table row? boundary row? else
// one LF step over a side it loads whole" (lf_own<1>: eight 16-byte loads of one line), and stores the reference index; the
// rows of a wave are neighbours in rowVal / rowRef, so both ends are coalesced.  walk2_body remains the table builder
// (long walks, millions of chains) and the debug tap.
// the reference of a hit in the position form (DIndex::posFrag): its sequence where the walk-left cannot leave it (true), else
// nothing (false: resolve_pos goes by the row)
CF_DEV bool resolve_pos_fast(const DIndex &ix, uint64_t pos, uint32_t &ref) {
    uint32_t lo = ix.posBucket[pos >> ix.posShift], hi = ix.posBucket[(pos >> ix.posShift) + 1] + 1;      // the fragment lies in [lo, hi)
    if (hi > ix.nPosFrag) hi = ix.nPosFrag;
    while (hi - lo > 1) { const uint32_t md = (lo + hi) >> 1; if (ix.posFrag[md].x <= pos) lo = md; else hi = md; }
    const uint32_t seq = (uint32_t)ix.posFrag[lo].y;
    const u64x2 span = ix.posSeq[seq];
    ref = seq;
    return pos >= span.x + ix.walkMax && pos + 12 <= span.y;
}
CF_DEV uint32_t resolve_plain_row(const DIndex &ix, uint64_t row, uint32_t &steps, uint32_t forced = 0);
// ... near an end of its sequence: the row of the suffix at pos — from the inverse sample at the next sampled position and the
// steps back from it with the rows' own characters, as the search did before the position form — then its walk.  (The steps back
// are `forced` steps of the walk's own loop: one copy of the LF code, no registers for a second.)
CF_DEV uint32_t resolve_pos_slow(const DIndex &ix, uint64_t pos) {
    const uint64_t pm = (1ull << ix.isaRate) - 1, q = (pos + pm) & ~pm;
    uint32_t steps = 0;
    return resolve_plain_row(ix, trio_at(ix.isa, q >> ix.isaRate), steps, (uint32_t)(q - pos));
}
CF_DEV uint32_t resolve_pos(const DIndex &ix, uint64_t pos) {
    uint32_t ref;
    return resolve_pos_fast(ix, pos, ref) ? ref : resolve_pos_slow(ix, pos);
}

// the walk-left of ONE row (tryOffset's order, bt2_idx.h:1980-2014, 2941-2963): '$' row, table row, boundary row, else a step
// forced: that many LF steps first, whatever the rows on the way are (resolve_pos_slow)
CF_DEV uint32_t resolve_plain_row(const DIndex &ix, uint64_t row, uint32_t &steps, uint32_t forced) {
    const uint64_t sampleMask = (1ull << ix.walkRate) - 1;
    uint32_t ref = 0;
    for (;;) {
        if (forced == 0 && row == ix.zOff) { ref = 0; break; }
        if (forced == 0 && (row & sampleMask) == 0) {
            const uint64_t e = row >> ix.walkRate;
            ref = ix.offw ? static_cast<const uint32_t *>(ix.walkOffs)[e] : static_cast<const uint16_t *>(ix.walkOffs)[e];
            break;
        }
        if (forced == 0 && ix.lastBoundary > 0 && row <= ix.lastBoundary) {
            const uint64_t blk = row >> ix.boundShift;
            if ((ix.boundBits[blk >> 5] >> (blk & 31)) & 1u) {
                uint32_t lo = 0, hi = ix.nBound;
                while (lo < hi) { const uint32_t md = (lo + hi) >> 1; if (ix.boundRow[md] < row) lo = md + 1; else hi = md; }
                if (lo < ix.nBound && ix.boundRow[lo] == row) { ref = ix.offw ? ix.boundRef[lo] : (ix.boundRef[lo] & 0xffffu); break; }
            }
        }
        if (ix.planes) {                                                       // LF with the row's own character (bt2_idx.h:2941-2963):
            // the four entries of the row's group are one 64-byte line; the character is the one whose bit is set at the row
            const uint8_t *p = ix.planes + (row >> 6) * 64;
            const uint32_t o = (uint32_t)row & 63u;
            const u64x2 e0 = cf_load16(p), e1 = cf_load16(p + 16), e2 = cf_load16(p + 32), e3 = cf_load16(p + 48);
            const u64x2 e = ((e0.x >> o) & 1) ? e0 : ((e1.x >> o) & 1) ? e1 : ((e2.x >> o) & 1) ? e2 : e3;
            row = e.y + popc_below(e.x, o);
        } else row = lf_own<1>(ix, row);
        if (forced) forced--; else steps++;
    }
    return ref;
}
// ... of a planned row as the plans carry it: a suffix-array row, or — marked — the text position of a hit in its position form
CF_DEV uint32_t resolve_row(const DIndex &ix, uint64_t row, uint32_t &steps) {
    return (row & kRowIsPos) ? resolve_pos(ix, row & ~kRowIsPos) : resolve_plain_row(ix, row, steps);
}
template <bool COUNT>
CF_DEV void walk3_body(const DIndex &ix, const DBatch &b, uint64_t i) {
    const uint64_t total = b.st->rowHi - b.st->rowLo;
    if (i >= total) return;
    uint32_t steps = 0;
    b.rowRef[i] = resolve_row(ix, b.rowVal[i], steps);
    if (COUNT && steps) cf_atomic_add(&b.ops->nWalk, (unsigned long long)steps);
}
// ... and of every planned row of ONE query, straight into rowRef: the queries the common-case score kernel left, when it
// takes its own rows' references from the table (DBatch::directRefs) and nothing was emitted or walked for the batch
CF_DEV void resolve_query_body(const DIndex &ix, const DParams &pr, const DBatch &b, uint32_t q) {
    for_each_planned_row(pr, b, q, [&](uint64_t at, uint64_t row) { uint32_t steps = 0; b.rowRef[at] = resolve_row(ix, row, steps); });
}

// ------------------------------------------------------------------- score
CF_DEV bool host_has(const DParams &pr, uint64_t tid) {
    uint32_t lo = 0, hi = pr.nHostSet;
    while (lo < hi) { const uint32_t md = (lo + hi) >> 1; if (pr.hostSet[md] < tid) lo = md + 1; else hi = md; }
    return lo < pr.nHostSet && pr.hostSet[lo] == tid;
}

CF_DEV uint32_t lcg_next(uint32_t &last) {                       // random_source.h:52-61
    last = 1664525u * last + 1013904223u;
    uint32_t ret = last >> 16;
    last = 1664525u * last + 1013904223u;
    return ret ^ last;
}

CF_DEV uint32_t path_len(const HmEntry &e) { return e.pid == kNone32 ? 0u : 10u; }
CF_DEV uint64_t path_at(const DIndex &ix, const HmEntry &e, uint32_t slot) { return ix.paths[(uint64_t)e.pid * 10 + slot]; }
CF_DEV uint32_t path_tidx_at(const DIndex &ix, const HmEntry &e, uint32_t slot) { return ix.pathTidx[(uint64_t)e.pid * 10 + slot]; }

// the taxon a reference is counted under (addHitToHitMap classifier.h:982-1001): its own, or the first one at or above the
// classification rank on its path
CF_DEV void ref_taxon_of(const DIndex &ix, const DParams &pr, const RefInfo &ri, uint64_t &tax, uint32_t &tidx, uint32_t &pid, uint32_t &rank) {
    tax = ri.tax; tidx = ri.tidx; pid = ri.pid;
    const uint32_t plen = pid == kNone32 ? 0u : 10u;
    rank = pr.rankSlot;
    if (rank > 0) {
        for (; rank < plen; rank++) {
            const uint64_t t = ix.paths[(uint64_t)pid * 10 + rank];
            if (t != 0) { tax = t; tidx = ix.pathTidx[(uint64_t)pid * 10 + rank]; break; }
        }
    }
}
CF_DEV void ref_taxon(const DIndex &ix, const DParams &pr, uint32_t ref, uint64_t &tax, uint32_t &tidx, uint32_t &pid, uint32_t &rank) {
    ref_taxon_of(ix, pr, ix.refInfo[ref], tax, tidx, pid, rank);
}

// SpeciesMetrics::addSpeciesCounts (aln_sink.h:142-172).  A global atomic is carried out at the memory side of the fabric (the
// XCDs' L2s are not coherent with each other), one transaction each: two per query — 20 M per batch — were what the score
// kernels issued per query.  A query that prints up to kFieldRows rows (or the "unclassified" row: taxon index 0) is counted from
// what the score kernel wrote anyway (nOut, the rows' taxon indices in o1b): count_body adds a chunk of queries up in LDS and
// sends one atomic pair per taxon seen and block.  Every printed row counts as a read of its taxon, the row of a query that
// prints ONE also as a unique read.  Queries with more than kFieldRows rows are few and use the atomics directly (score_body).
constexpr uint32_t kCountSlotBits = 12, kCountSlots = 1u << kCountSlotBits, kCountChunk = 32768, kCountProbes = 8, kCountEmpty = 0xffffffffu;
// block `chunk`: the queries [chunk * kCountChunk, + kCountChunk) of the pass's window, ONE pass over them whatever the number
// of taxa (a real taxonomy has 10^4 - 10^6 nodes, the synthetic ones a few thousand).  lds = keys[nSlots], reads[nSlots],
// unique[nSlots]; nSlots = 1 << slotBits <= kCountSlots.  `direct` (the host sets it when nTaxa <= nSlots): slot = taxon index;
// else an open hash (multiplicative, linear probing, kCountProbes tries) and a taxon that finds no slot is counted by the far atomics.
CF_DEV void count_body(const DBatch &b, uint32_t *lds, uint32_t chunk, uint32_t slotBits, bool direct) {
    const uint32_t t = cf_local_thread(), nt = cf_block_threads();
    const uint32_t nSlots = 1u << slotBits;
    uint32_t *keys = lds, *cnt = lds + nSlots, *unq = lds + 2 * nSlots;
    for (uint32_t i = t; i < nSlots; i += nt) { keys[i] = kCountEmpty; cnt[i] = 0; unq[i] = 0; }
    cf_block_sync();
    const uint32_t q0 = chunk * kCountChunk, qLo = b.st->qLo, qHi = b.st->qHi;
    for (uint32_t i = t; i < kCountChunk; i += nt) {
        const uint32_t q = q0 + i;
        if (q < qLo || q >= qHi) continue;
        const uint32_t no = b.nOut[q];
        if (no > kFieldRows) continue;                           // (counted by the score kernel, row by row)
        const bool one = no <= 1;
#pragma unroll
        for (uint32_t j = 0; j < kFieldRows; j++) {
            if (j >= (no ? no : 1u)) continue;
            const uint32_t tidx = no ? (uint32_t)(b.o1b[j * b.oStride + q] >> 32) : 0u;   // no rows: the "unclassified" row, taxid 0
            if (tidx >= b.nTaxa) continue;                       // not on a well-formed index
            const uint32_t h = direct ? tidx : (tidx * 0x9e3779b1u) >> (32u - slotBits);
            bool placed = false;
            for (uint32_t pr = 0; pr < kCountProbes && !placed; pr++) {
                const uint32_t s = (h + pr) & (nSlots - 1);
                uint32_t k = keys[s];
                if (k == kCountEmpty) k = cf_atomic_cas(&keys[s], kCountEmpty, tidx);     // the old key: empty = the slot is ours now
                if (k == kCountEmpty || k == tidx) { cf_atomic_add(&cnt[s], 1u); if (one) cf_atomic_add(&unq[s], 1u); placed = true; }
            }
            if (!placed) {
                cf_atomic_add(&b.counts[tidx], 1ull);
                if (one) cf_atomic_add(&b.counts[b.nTaxa + tidx], 1ull);
            }
        }
    }
    cf_block_sync();
    for (uint32_t i = t; i < nSlots; i += nt) {
        const uint32_t k = keys[i], n = cnt[i], u = unq[i];
        if (k == kCountEmpty || n == 0) continue;
        cf_atomic_add(&b.counts[k], (unsigned long long)n);
        if (u) cf_atomic_add(&b.counts[b.nTaxa + k], (unsigned long long)u);
    }
}

// score_body for the common query, in registers: its planned hits are inline (qplan: at most kInlinePlan), it resolved at most
// kScoreFastRows rows, they lead to at most kFastEntries hit-map entries, and those are no more than k (no climb:
// classifier.h:399).  Everything the general kernel does for such a query — hit map with the ts rule (classifier.h:305-378,
// 982-1050), scores, host list (:385-394), 2ndBest, selectByScore with the per-read LCG (aln_sink.h:1860-1927) — on a handful
// of registers with compile-time indices: no hit-map or parent-count scratch in memory, no walk over the hit lists, the
// row of a query that prints one by field.  Returns true when the query is left to score_body, having written nothing.
constexpr uint32_t kScoreFastRows = 8, kFastEntries = 4;
// EARLY (round 5; needs DBatch::directRefs): the kernel runs right behind the common-case post kernel, BESIDE the general one (which
// is a few thousand chains of dependent loads: latency, hardly any bandwidth — 0.9 of the repeat-rich batch's 12 ms), before the rows
// are counted and the row window of the pass is known.  A query the common-case post kernel finished needs neither: its plan is
// inline, its references come straight from the resolve table.  A query it left (postDeferred) is not looked at — the general
// post kernel is writing its plan at this very moment — and goes to the general score kernel, like every other query this one
// leaves; until that has scored them they print nothing (nOut = 0).
template <bool EARLY = false>
CF_DEV bool score_fast_body(const DIndex &ix, const DParams &pr, const DBatch &b, uint32_t q) {
    if (EARLY) {
        if (b.postDeferred[q] || (b.st->flags & kStHitsOverflow)) { b.nOut[q] = 0; b.score2[q] = 0; return !(b.st->flags & kStHitsOverflow); }
    } else if (q < b.st->qLo || q >= b.st->qHi) {                // not in this pass's row window: scored in a later pass (or it was in
        if (b.st->qLo == 0) b.nOut[q] = 0;                       // an earlier one).  Until then the query prints nothing, so that the
        return false;                                            // compaction behind the first pass stays inside its buffers
    }
    const uint32_t qf = b.qflag[q], nPlan = qf_nplan(qf), nRows = b.qRows[q];
    if (nPlan == kPlanNotInline || nRows > kScoreFastRows) { if (EARLY) { b.nOut[q] = 0; b.score2[q] = 0; } return true; }
    const uint64_t base = EARLY ? 0 : b.qBase[q] - b.st->rowLo;        // (EARLY: the rows are not counted yet — and not needed: directRefs)
    uint64_t eTax[kFastEntries];
    uint32_t eRef[kFastEntries], eTidx[kFastEntries], eTs[kFastEntries], eSc[kFastEntries][4], eHl[kFastEntries][4];
#pragma unroll
    for (uint32_t z = 0; z < kFastEntries; z++) {
        eTax[z] = 0; eRef[z] = 0; eTidx[z] = 0; eTs[z] = 0;
#pragma unroll
        for (int c = 0; c < 4; c++) { eSc[z][c] = 0; eHl[z][c] = 0; }
    }
    // ---- the plan, the rows and what the rows lead to, as ROUNDS of independent loads (round 6).  Row after row a lane ran a chain
    // of four or five dependent trips per row — plan, bucket, fragment, sequence span, reference record — and its wavefront waited
    // for the lane with the most rows; here every round issues the loads of all of the query's rows (at most kScoreFastRows) before
    // anything looks at what came back.  Same rows in the same order into the same hit map.
    uint64_t pTop[kInlinePlan];
    uint32_t pNelt[kInlinePlan], pMeta[kInlinePlan];
#pragma unroll
    for (uint32_t j = 0; j < kInlinePlan; j++) {
        PlanHit ph{0, 0, 0};
        if (j < nPlan) ph = b.qplan[(uint64_t)j * b.qplanStride + q];
        pTop[j] = ph.top; pNelt[j] = ph.nelt; pMeta[j] = ph.meta;
    }
    static_assert(kInlinePlan == 4, "three thresholds below");
    const uint32_t c1 = pNelt[0], c2 = c1 + pNelt[1], c3 = c2 + pNelt[2];
    uint32_t total = c3 + pNelt[3];                               // (= nRows: every planned hit's rows)
    if (total > kScoreFastRows) total = kScoreFastRows;
    // (kGather rows at a time: nearly every query has no more, and eight rows' worth of loads in flight cost a wavefront per SIMD)
    constexpr uint32_t kGather = 4;
    uint32_t nh = 0;
    for (uint32_t s0 = 0; s0 < total; s0 += kGather) {
        uint64_t rowv[kGather];                                // row s of the query: row e of planned hit j, hits in plan order
        uint32_t rmeta[kGather], ref[kGather];
#pragma unroll
        for (uint32_t i = 0; i < kGather; i++) {
            const uint32_t s = s0 + i;
            const bool g1 = s >= c1, g2 = s >= c2, g3 = s >= c3;
            rowv[i] = (g3 ? pTop[3] : g2 ? pTop[2] : g1 ? pTop[1] : pTop[0]) + (s - (g3 ? c3 : g2 ? c2 : g1 ? c1 : 0u));
            rmeta[i] = g3 ? pMeta[3] : g2 ? pMeta[2] : g1 ? pMeta[1] : pMeta[0];
            ref[i] = 0xffffffffu;
        }
        // (Every load of a round is UNCONDITIONAL: a slot that has nothing to fetch reads the first element of the table — one line
        // for all such lanes of the wavefront — and its value is not looked at.  A load inside a divergent branch gets its wait
        // inside the branch, or at the next branch that writes a register the allocator shares with it: a round's loads then go out
        // one at a time.)
        bool act[kGather], isPos[kGather];
#pragma unroll
        for (uint32_t i = 0; i < kGather; i++) { act[i] = s0 + i < total; isPos[i] = act[i] && (rowv[i] & kRowIsPos) != 0; }
        if (b.directRefs) {                                          // the table holds every row (and 0 at the '$' row): the walk IS this read
            uint32_t tref[kGather];
            if (ix.offw) {
#pragma unroll
                for (uint32_t i = 0; i < kGather; i++) tref[i] = static_cast<const uint32_t *>(ix.walkOffs)[act[i] && !isPos[i] ? rowv[i] : 0ull];
            } else {
#pragma unroll
                for (uint32_t i = 0; i < kGather; i++) tref[i] = static_cast<const uint16_t *>(ix.walkOffs)[act[i] && !isPos[i] ? rowv[i] : 0ull];
            }
            if (ix.posFrag) {
                // a row in its position form (resolve_pos_fast, the same steps for all of them side by side): its bucket's fragments ...
                uint64_t lh[kGather];                                // (posBucket[bk], posBucket[bk + 1]: one 8-byte load)
#pragma unroll
                for (uint32_t i = 0; i < kGather; i++)
                    lh[i] = cf_load8(reinterpret_cast<const uint8_t *>(ix.posBucket + (isPos[i] ? (rowv[i] & ~kRowIsPos) >> ix.posShift : 0ull)));
                uint32_t lo[kGather], hi[kGather];
                bool narrow = false;
#pragma unroll
                for (uint32_t i = 0; i < kGather; i++) {
                    lo[i] = isPos[i] ? (uint32_t)lh[i] : 0u;
                    hi[i] = isPos[i] ? (uint32_t)(lh[i] >> 32) + 1 : 0u;
                    if (hi[i] > ix.nPosFrag) hi[i] = ix.nPosFrag;
                    narrow = narrow || hi[i] - lo[i] > 1;
                }
                while (narrow) {                                      // ... the one that holds the position (rare: a bucket is 16 K positions)
                    uint64_t fx[kGather];
#pragma unroll
                    for (uint32_t i = 0; i < kGather; i++) fx[i] = ix.posFrag[hi[i] - lo[i] > 1 ? (lo[i] + hi[i]) >> 1 : 0u].x;
                    narrow = false;
#pragma unroll
                    for (uint32_t i = 0; i < kGather; i++) {
                        if (hi[i] - lo[i] > 1) { const uint32_t md = (lo[i] + hi[i]) >> 1; if (fx[i] <= (rowv[i] & ~kRowIsPos)) lo[i] = md; else hi[i] = md; }
                        narrow = narrow || hi[i] - lo[i] > 1;
                    }
                }
                // ... its sequence, and whether the walk-left from the position provably stays inside it (else: the general kernel)
                uint32_t sq[kGather];
#pragma unroll
                for (uint32_t i = 0; i < kGather; i++) sq[i] = (uint32_t)ix.posFrag[lo[i]].y;
                u64x2 span[kGather];
#pragma unroll
                for (uint32_t i = 0; i < kGather; i++) span[i] = ix.posSeq[isPos[i] ? sq[i] : 0u];
                bool leave = false;
#pragma unroll
                for (uint32_t i = 0; i < kGather; i++) {
                    const uint64_t pos = rowv[i] & ~kRowIsPos;
                    if (isPos[i] && !(pos >= span[i].x + ix.walkMax && pos + 12 <= span[i].y)) leave = true;
                    if (isPos[i]) tref[i] = sq[i];
                }
                if (leave) { if (EARLY) { b.nOut[q] = 0; b.score2[q] = 0; } return true; }
            }
#pragma unroll
            for (uint32_t i = 0; i < kGather; i++) if (act[i]) ref[i] = tref[i];
        } else {
#pragma unroll
            for (uint32_t i = 0; i < kGather; i++) { const uint32_t v = b.rowRef[act[i] ? base + s0 + i : 0ull]; if (act[i]) ref[i] = v; }
        }
        // the references' records (ref_taxon) and their places on the exclusion list
        RefInfo ri[kGather];
        uint8_t rex[kGather];
#pragma unroll
        for (uint32_t i = 0; i < kGather; i++) ri[i] = ix.refInfo[ref[i] < ix.nRef ? ref[i] : 0u];
#pragma unroll
        for (uint32_t i = 0; i < kGather; i++) rex[i] = pr.refExcluded ? pr.refExcluded[ref[i] < ix.nRef ? ref[i] : 0u] : (uint8_t)0;
#pragma unroll
        for (uint32_t i = 0; i < kGather; i++) {
            if (s0 + i >= total) continue;
            if (ref[i] >= ix.nRef) continue;                         // not on a well-formed index
            if (rex[i]) continue;                                    // classifier.h:339
            const uint32_t len = pm_len(rmeta[i]), ts = pm_ts(rmeta[i]);
            const uint32_t col = 2u * (uint32_t)pm_rdi(rmeta[i]) + (uint32_t)pm_f(rmeta[i]);     // [mate][strand]
            const uint32_t sc = (len - 15) * (len - 15);             // classifier.h:332
            uint64_t tax; uint32_t tidx, pid, rank;
            ref_taxon_of(ix, pr, ri[i], tax, tidx, pid, rank);
            // addHitToHitMap classifier.h:982-1050: the entry of this reference (of this taxon at a classification rank), or a new one
            uint32_t at = kFastEntries;
#pragma unroll
            for (uint32_t z = 0; z < kFastEntries; z++)
                if (z < nh && at == kFastEntries && (pr.rankSlot == 0 ? eRef[z] == ref[i] : eTax[z] == tax)) at = z;
            bool add = true;
            if (at == kFastEntries) {
                if (nh == kFastEntries) { if (EARLY) { b.nOut[q] = 0; b.score2[q] = 0; } return true; }   // a fifth entry: the general kernel
                at = nh++;
#pragma unroll
                for (uint32_t z = 0; z < kFastEntries; z++) if (z == at) { eTax[z] = tax; eRef[z] = ref[i]; eTidx[z] = tidx; eTs[z] = ts; }
            } else {
#pragma unroll
                for (uint32_t z = 0; z < kFastEntries; z++) if (z == at) { add = eTs[z] != ts; eTs[z] = ts; }   // once per hit and entry
            }
            if (add) {
#pragma unroll
                for (uint32_t z = 0; z < kFastEntries; z++) {
#pragma unroll
                    for (uint32_t c = 0; c < 4; c++) if (z == at && c == col) { eSc[z][c] += sc; eHl[z][c] += len; }
                }
            }
        }
    }
    if (nh > pr.k) { if (EARLY) { b.nOut[q] = 0; b.score2[q] = 0; } return true; }   // more entries than -k: the climb (classifier.h:399-515)
    // finalize (classifier.h:86-120, 380-382)
    const bool paired = qf_paired(qf);
    uint32_t score[kFastEntries], hitLen[kFastEntries];
#pragma unroll
    for (uint32_t z = 0; z < kFastEntries; z++) {
        score[z] = eSc[z][0] > eSc[z][1] ? eSc[z][0] : eSc[z][1];
        hitLen[z] = eHl[z][0] > eHl[z][1] ? eHl[z][0] : eHl[z][1];
        if (paired) { score[z] += eSc[z][2] > eSc[z][3] ? eSc[z][2] : eSc[z][3]; hitLen[z] += eHl[z][2] > eHl[z][3] ? eHl[z][2] : eHl[z][3]; }
    }
    // host logic (classifier.h:385-394)
    bool onlyHost = false;
    if (pr.nHostSet) {
        uint32_t best = 0;
#pragma unroll
        for (uint32_t z = 0; z < kFastEntries; z++) {
            if (z >= nh) continue;
            if (score[z] > best) { best = score[z]; onlyHost = host_has(pr, eTax[z]); }
            else if (score[z] == best) onlyHost = onlyHost || host_has(pr, eTax[z]);
        }
    }
    // results in hit-map order (:537-565): res[i] = the entry of result i
    uint32_t res[kFastEntries], rs[kFastEntries], nres = 0;
#pragma unroll
    for (uint32_t z = 0; z < kFastEntries; z++) { res[z] = 0; rs[z] = 0; }
#pragma unroll
    for (uint32_t z = 0; z < kFastEntries; z++) {
        if (z >= nh) continue;
        if (onlyHost && !host_has(pr, eTax[z])) continue;
#pragma unroll
        for (uint32_t i = 0; i < kFastEntries; i++) if (i == nres) { res[i] = z; rs[i] = score[z]; }
        nres++;
    }
    const uint32_t r0 = b.paired ? 2 * q : q;
    if (nres == 0) {                                             // the "unclassified" row
        b.nOut[q] = 0; b.score2[q] = 0;
        return false;
    }
    // 2ndBest over all results (aligner_result.h:398-431)
    uint32_t score2 = 0;
    {
        uint32_t bst = 0, sec = 0; bool hb = false, hs = false;
#pragma unroll
        for (uint32_t i = 0; i < kFastEntries; i++) {
            if (i >= nres) continue;
            const uint32_t v = rs[i];
            if (!hb || v > bst) { sec = bst; hs = hb; bst = v; hb = true; }
            else if (!hs || v > sec) { sec = v; hs = true; }
        }
        score2 = hs ? sec : 0;
    }
    // selectByScore (aln_sink.h:1860-1927): descending (score, result index) — pos[p] = the result at place p — then the tie
    // streaks shuffled with the per-read LCG (ds.h:784-795)
    uint32_t pos[kFastEntries], ps[kFastEntries];
#pragma unroll
    for (uint32_t p = 0; p < kFastEntries; p++) { pos[p] = 0; ps[p] = 0; }
#pragma unroll
    for (uint32_t i = 0; i < kFastEntries; i++) {
        if (i >= nres) continue;
        uint32_t place = 0;                                      // results that come before result i
#pragma unroll
        for (uint32_t j = 0; j < kFastEntries; j++) if (j < nres && j != i && (rs[j] > rs[i] || (rs[j] == rs[i] && j > i))) place++;
#pragma unroll
        for (uint32_t p = 0; p < kFastEntries; p++) if (p == place) { pos[p] = i; ps[p] = rs[i]; }
    }
    if (nres > 1) {
        uint32_t rnd = b.seeds[r0];
        if (paired) rnd ^= b.seeds[r0 + 1];                      // centrifuge.cpp:2608-2613
        uint32_t streak = 0;
#pragma unroll
        for (uint32_t i = 1; i <= kFastEntries; i++) {
            if (i > nres) continue;
            bool tie = false;
#pragma unroll
            for (uint32_t p = 1; p < kFastEntries; p++) if (p == i && i < nres) tie = ps[p] == ps[p - 1];
            if (tie) { if (streak == 0) streak = 1; streak++; }
            else {
                if (streak > 1) {
                    const uint32_t begin = i - streak;
#pragma unroll
                    for (uint32_t z = 0; z + 1 < kFastEntries; z++) {
                        if (z < begin || z + 1 >= begin + streak) continue;
                        const uint32_t left = streak - (z - begin);
                        const uint32_t r = lcg_next(rnd) % left;
#pragma unroll
                        for (uint32_t t = 1; t < kFastEntries; t++)
                            if (r == t && z + t < kFastEntries) { const uint32_t x = pos[z]; pos[z] = pos[z + t < kFastEntries ? z + t : z]; pos[z + t < kFastEntries ? z + t : z] = x; }
                    }
                }
                streak = 0;
            }
        }
    }
    uint32_t num = nres < pr.k ? nres : pr.k;                    // aln_sink.h:2442-2458
#pragma unroll
    for (uint32_t i = 0; i + 1 < kFastEntries; i++) if (i + 1 < num && ps[i] != ps[i + 1]) num = i + 1;
    // the rows, by field (k_count counts them: aln_sink.h:142-172)
#pragma unroll
    for (uint32_t i = 0; i < kFastEntries; i++) {
        if (i >= num) continue;
        uint32_t z = 0;
#pragma unroll
        for (uint32_t p = 0; p < kFastEntries; p++) if (p == i) z = pos[p];
        uint32_t en = 0;
#pragma unroll
        for (uint32_t y = 0; y < kFastEntries; y++) if (y == z) en = res[y];
        uint64_t tax = 0; uint32_t ref = 0, tidx = 0, sco = 0, hle = 0;
#pragma unroll
        for (uint32_t y = 0; y < kFastEntries; y++) if (y == en) { tax = eTax[y]; ref = eRef[y]; tidx = eTidx[y]; sco = score[y]; hle = hitLen[y]; }
        static_assert(kFastEntries <= kFieldRows, "every row of the common-case kernel goes out by field");
        b.o1tax[i * b.oStride + q] = tax; b.o1a[i * b.oStride + q] = (uint64_t)ref | ((uint64_t)sco << 32); b.o1b[i * b.oStride + q] = (uint64_t)hle | ((uint64_t)tidx << 32);
    }
    b.nOut[q] = num; b.score2[q] = score2;
    return false;
}

// bytes of scratch a lane of score_body needs for a query of up to `rows` planned rows: its hit map, its parent counts / result
// order, its rows' references
constexpr uint32_t score_scratch_bytes(uint32_t rows) { return rows * (uint32_t)(sizeof(HmEntry) + sizeof(TcEntry) + sizeof(uint32_t)); }
// scratch / scratchRows (round 6): this lane's own room (LDS on the device) for a query of up to scratchRows planned rows.  Like
// post_body the kernel is a chain of dependent accesses — the dedup of a hit's references, the linear searches of the hit map
// (one per reference), the climb's passes over map and parent counts — and in the row workspace each is a trip to L2 or HBM; in
// the scratch only the taxonomy gathers (refInfo, paths) are.  Nothing comes back out: the printed rows go where they always went.
// rowsMin / rowsMax: the launch takes the queries that planned that many rows (the kernel runs twice: the many small queries in
// place, side by side in as many lanes as the device holds — and the few large ones, whose passes over hit map and parent counts
// are what the whole kernel used to wait for, one per wavefront with their state in LDS)
CF_DEV void score_body(const DIndex &ix, const DParams &pr, const DBatch &b, uint32_t q, uint8_t *scratch = nullptr, uint32_t scratchRows = 0,
                       uint32_t rowsMin = 0, uint32_t rowsMax = 0xffffffffu) {
    if (q < b.st->qLo || q >= b.st->qHi) return;                 // not in this pass's row window
    if (b.qRows[q] < rowsMin || b.qRows[q] > rowsMax) return;
    const uint32_t qf = b.qflag[q], nPlanQ = qf_nplan(qf);
    const bool qPaired = qf_paired(qf);
    const uint32_t k = pr.k;
    OutRow *out = b.out + (uint64_t)q * k;
    const uint64_t base = b.qBase[q] - b.st->rowLo;
    HmEntry *hm = b.hm + base;
    TcEntry *tc = b.tc + base;
    uint32_t *refBase = b.rowRef + base;
    {
        const uint32_t rowsQ = b.qRows[q];
        if (scratch != nullptr && rowsQ <= scratchRows) {
            hm = reinterpret_cast<HmEntry *>(scratch);
            tc = reinterpret_cast<TcEntry *>(scratch + sizeof(HmEntry) * scratchRows);
            uint32_t *lr = reinterpret_cast<uint32_t *>(scratch + (sizeof(HmEntry) + sizeof(TcEntry)) * scratchRows);
            uint32_t i = 0;
            for (; i + 4 <= rowsQ; i += 4) { const uint32_t a = refBase[i], c1 = refBase[i + 1], c2 = refBase[i + 2], c3 = refBase[i + 3]; lr[i] = a; lr[i + 1] = c1; lr[i + 2] = c2; lr[i + 3] = c3; }
            for (; i < rowsQ; i++) lr[i] = refBase[i];
            refBase = lr;
        }
    }
    uint32_t nh = 0;
    const uint32_t r0 = (b.paired ? 2 * q : q);
    uint32_t rowoff = 0;                                             // rows of the hits before this one (k_post's plan order)
    // one planned hit into the hit map: its rows' references, distinct, in first-seen order (classifier.h:305-372)
    auto addHit = [&](uint32_t ne, uint32_t len, int rdi, int f, uint32_t ts) {
        uint32_t *refs = refBase + rowoff;
        rowoff += ne;
        uint32_t nid = 0;
        for (uint32_t e = 0; e < ne; e++) {                                      // classifier.h:305-326
            const uint32_t ref = refs[e];
            bool found = false;
            for (uint32_t z = 0; z < nid && !found; z++) found = refs[z] == ref;
            if (!found) refs[nid++] = ref;
        }
        const uint32_t sc = (len - 15) * (len - 15);                             // classifier.h:332
        for (uint32_t z = 0; z < nid; z++) {
            const uint32_t ref = refs[z];
            if (ref >= ix.nRef) continue;                                        // not on a well-formed index
            if (pr.refExcluded && pr.refExcluded[ref]) continue;                 // classifier.h:339
            // addHitToHitMap classifier.h:982-1050
            uint64_t tax; uint32_t tidx, pid, rank;
            ref_taxon(ix, pr, ref, tax, tidx, pid, rank);
            uint32_t idx = 0;
            for (; idx < nh; idx++) {
                const bool same = rank == 0 ? (hm[idx].uniqueID == ref) : (hm[idx].taxID == tax);
                if (same) {
                    if (hm[idx].ts != ts) { hm[idx].sc[rdi][f] += sc; hm[idx].hl[rdi][f] += len; hm[idx].ts = ts; }
                    break;
                }
            }
            if (idx >= nh) {
                HmEntry e;
                e.taxID = tax; e.uniqueID = ref; e.pid = pid; e.tidx = tidx;
                e.sc[0][0] = e.sc[0][1] = e.sc[1][0] = e.sc[1][1] = 0;
                e.hl[0][0] = e.hl[0][1] = e.hl[1][0] = e.hl[1][1] = 0;
                e.sc[rdi][f] = sc; e.hl[rdi][f] = len;
                e.ts = ts; e.score = 0; e.hitLen = 0; e.rank = (uint8_t)rank;
                for (int z2 = 0; z2 < 7; z2++) e.pad[z2] = 0;
                hm[nh++] = e;
            }
        }
    };
    if (nPlanQ != kPlanNotInline) {                                  // the planned hits as k_post left them with the query
        for (uint32_t j = 0; j < nPlanQ; j++) {
            const PlanHit ph = b.qplan[(uint64_t)j * b.qplanStride + q];
            addHit(ph.nelt, pm_len(ph.meta), pm_rdi(ph.meta), pm_f(ph.meta), pm_ts(ph.meta));
        }
    } else {
        const QHead *qp = &b.qhead[q];
        uint32_t ts = 0;                                             // classifier.h:232
        for (int rdi = 0; rdi < qf_nmates(qf); rdi++) {
            const uint32_t rd = r0 + qf_first(qf) + rdi;
            for (int f = qp->lo[rdi]; f < qp->hi[rdi]; f++) {
                const HitP *h = b.hits + b.hitBase[rd] + (f ? b.hitCap[rd] : 0u);
                const uint32_t np = qp->nProc[rdi][f];
                const uint64_t mg = qp->maxG[rdi][f];
                for (uint32_t i = 0; i < np; i++, ts++) {
                    const HitP hp = h[i];
                    const uint32_t ne = plan_nelt(hp_len(hp), hp_size(hp), mg, pr.m, pr.ihits);
                    if (ne == 0) continue;
                    addHit(ne, hp_len(hp), rdi, f, ts);
                }
                // the iteration that left through `break` did not run ts++ (classifier.h:366-367)
                if ((qp->brk[rdi] >> f) & 1) ts--;
            }
        }
    }
    // finalize (classifier.h:86-120, 380-382)
    for (uint32_t i = 0; i < nh; i++) {
        HmEntry &e = hm[i];
        const uint32_t s0 = e.sc[0][0] > e.sc[0][1] ? e.sc[0][0] : e.sc[0][1];
        const uint32_t l0 = e.hl[0][0] > e.hl[0][1] ? e.hl[0][0] : e.hl[0][1];
        if (qPaired) {
            const uint32_t s1 = e.sc[1][0] > e.sc[1][1] ? e.sc[1][0] : e.sc[1][1];
            const uint32_t l1 = e.hl[1][0] > e.hl[1][1] ? e.hl[1][0] : e.hl[1][1];
            e.score = s0 + s1; e.hitLen = l0 + l1;
        } else { e.score = s0; e.hitLen = l0; }
    }
    // host logic (classifier.h:385-394)
    bool onlyHost = false;
    if (pr.nHostSet) {
        uint32_t best = 0;
        for (uint32_t i = 0; i < nh; i++) {
            if (hm[i].score > best) { best = hm[i].score; onlyHost = host_has(pr, hm[i].taxID); }
            else if (hm[i].score == best) onlyHost = onlyHost || host_has(pr, hm[i].taxID);
        }
    }
    bool unclassified = false;
    if (!onlyHost && nh > k) {                                                   // classifier.h:399-515
        uint32_t bs = hm[0].score;
        for (uint32_t i = 1; i < nh; i++) if (bs < hm[i].score) bs = hm[i].score;
        for (int i = 0; i < (int)nh; i++) {                                      // :409-417
            if (hm[i].score < bs) { if (i + 1 < (int)nh) hm[i] = hm[nh - 1]; nh--; i--; }
        }
        if (!pr.traverse && nh > k) unclassified = true;                         // :419-425
        uint32_t rank = 0;
        while (!unclassified && nh > k) {                                        // :428-514
            uint32_t ntc = 0;
            for (uint32_t i = 0; i < nh; i++) {
                HmEntry &e = hm[i];
                const uint32_t plen = path_len(e);
                while (e.rank < rank) {
                    if ((uint32_t)e.rank + 1 >= plen) { e.rank = 255; break; }
                    e.rank += 1; e.taxID = path_at(ix, e, e.rank); e.tidx = path_tidx_at(ix, e, e.rank);
                }
                if (e.rank > rank) continue;
                uint64_t parent; uint32_t ptidx;
                if (rank + 1 >= plen) { parent = 1; ptidx = ix.tidxOne; }
                else { parent = path_at(ix, e, rank + 1); ptidx = path_tidx_at(ix, e, rank + 1); }
                if (parent == 0) continue;
                uint32_t j = 0;
                for (; j < ntc; j++) if (tc[j].tid == parent) { tc[j].cnt++; break; }
                if (j == ntc) { tc[ntc].cnt = 1; tc[ntc].tid = parent; tc[ntc].tidx = ptidx; ntc++; }
            }
            if (ntc == 0) {
                if (rank < path_len(hm[0])) { rank++; continue; }
                break;
            }
            for (uint32_t a = 1; a < ntc; a++) {                                 // sort (count, taxid) ascending, :467
                const TcEntry v = tc[a];
                uint32_t c = a;
                while (c > 0 && (tc[c - 1].cnt > v.cnt || (tc[c - 1].cnt == v.cnt && tc[c - 1].tid > v.tid))) { tc[c] = tc[c - 1]; c--; }
                tc[c] = v;
            }
            uint32_t j = ntc;
            while (j-- > 0) {
                const uint64_t parent = tc[j].tid;
                for (uint32_t i = 0; i < nh; i++) {
                    HmEntry &e = hm[i];
                    if (e.rank != rank) continue;
                    const uint32_t plen = path_len(e);
                    const uint64_t cp = (rank + 1 >= plen) ? 1 : path_at(ix, e, rank + 1);
                    if (parent == cp) { e.uniqueID = kNone32; e.rank = (uint8_t)(rank + 1); e.taxID = parent; e.tidx = tc[j].tidx; }
                }
                bool first = true;
                for (uint32_t i = 0; i < nh; i++) {                              // :489-506
                    if (parent == hm[i].taxID) {
                        if (!first) { if (i + 1 < nh) hm[i] = hm[nh - 1]; nh--; i--; }
                        else first = false;
                    }
                }
                if (nh <= k) break;
            }
            ++rank;
            if (rank > path_len(hm[0])) break;
        }
    }
    if (!onlyHost && nh > k) unclassified = true;                                // :516-520
    uint32_t nOut = 0, score2 = 0;
    if (!unclassified) {
        // results in hit-map order (:537-565), then selectByScore aln_sink.h:1860-1927;
        // tc[] is reused as the (score, idx) buffer
        uint32_t nres = 0;
        for (uint32_t i = 0; i < nh; i++) {
            if (onlyHost && !host_has(pr, hm[i].taxID)) continue;
            tc[nres].cnt = hm[i].score; tc[nres].tidx = nres; tc[nres].tid = i; nres++;
        }
        if (nres > 0) {
            // 2ndBest over all results (aligner_result.h:398-431)
            {
                uint32_t bst = 0, sec = 0; bool hb = false, hs = false;
                for (uint32_t i = 0; i < nres; i++) {
                    const uint32_t s = tc[i].cnt;
                    if (!hb || s > bst) { sec = bst; hs = hb; bst = s; hb = true; }
                    else if (!hs || s > sec) { sec = s; hs = true; }
                }
                score2 = hs ? sec : 0;
            }
            for (uint32_t a = 1; a < nres; a++) {                                // descending (score, idx)
                const TcEntry v = tc[a];
                uint32_t c = a;
                while (c > 0 && (tc[c - 1].cnt < v.cnt || (tc[c - 1].cnt == v.cnt && tc[c - 1].tidx < v.tidx))) { tc[c] = tc[c - 1]; c--; }
                tc[c] = v;
            }
            // shuffle tie streaks with the per-read LCG (ds.h:784-795)
            uint32_t rnd = b.seeds[r0];
            if (qPaired) rnd ^= b.seeds[r0 + 1];                                 // centrifuge.cpp:2608-2613
            uint32_t streak = 0;
            for (uint32_t i = 1; i <= nres; i++) {
                if (i < nres && tc[i].cnt == tc[i - 1].cnt) { if (streak == 0) streak = 1; streak++; }
                else {
                    if (streak > 1) {
                        const uint32_t begin = i - streak;
                        uint32_t left = streak;
                        for (uint32_t z = begin; z < begin + streak - 1; z++) {
                            const uint32_t r = lcg_next(rnd) % left;
                            if (r > 0) { const TcEntry t = tc[z]; tc[z] = tc[z + r]; tc[z + r] = t; }
                            left--;
                        }
                    }
                    streak = 0;
                }
            }
            uint32_t num = nres < k ? nres : k;                                  // aln_sink.h:2442-2458
            for (uint32_t i = 0; i + 1 < num; i++) if (tc[i].cnt != tc[i + 1].cnt) { num = i + 1; break; }
            for (uint32_t i = 0; i < num; i++) {
                const HmEntry &e = hm[tc[i].tid];
                if (num <= kFieldRows) {                                         // by field (k_compact, k_count read them there)
                    b.o1tax[i * b.oStride + q] = e.taxID; b.o1a[i * b.oStride + q] = (uint64_t)e.uniqueID | ((uint64_t)e.score << 32);
                    b.o1b[i * b.oStride + q] = (uint64_t)e.hitLen | ((uint64_t)e.tidx << 32);
                } else {
                    OutRow o; o.taxID = e.taxID; o.uniqueID = e.uniqueID; o.score = e.score; o.hitLen = e.hitLen; o.tidx = e.tidx;
                    out[i] = o;
                }
            }
            nOut = num;
        }
    }
    b.nOut[q] = nOut;
    b.score2[q] = score2;
    // SpeciesMetrics::addSpeciesCounts (aln_sink.h:142-172): per printed row; up to kFieldRows rows through count_body
    if (nOut > kFieldRows && b.counts) for (uint32_t i = 0; i < nOut; i++) cf_atomic_add(&b.counts[out[i].tidx], 1ull);
}

}  // namespace cfamd
