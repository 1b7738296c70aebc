// tests/emu/emu.cpp — TEST HARNESS ONLY (never linked into libcentrifuge_amd.so).
//
// Compiles the kernel bodies of centrifuge_amd/csrc/cf_kernels.hpp with
// CF_HOST_EMU (a one-lane "wavefront", see cf_platform.hpp) and steps them on
// the CPU in the same order the device layer launches them, so that the search
// state machine, extend/twin/trim, the std::sort restatement, the row plan, the
// hit map, the climb and the selection can be checked against the oracle in the
// `-m "not gpu"` tests.  Built with CF_EMU_WAVE64 as well (libcfemu64.so) the
// search kernels run as a wavefront of 64 lanes — fibers that meet at the
// cross-lane primitives, emu_run_wave below — one chain per lane over the
// planes, two lanes per chain over the sides.  Several wavefronts on one work
// queue can only be exercised on a GPU (tests marked gpu).
#define CF_HOST_EMU 1
#include <algorithm>
#include <cstdio>
#include <cstring>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../centrifuge_amd/csrc/cf_index.hpp"
#include "../../centrifuge_amd/csrc/cf_kernels.hpp"
#include "../../centrifuge_amd/csrc/cf_plan.hpp"
#include "../../centrifuge_amd/csrc/cf_restore.hpp"
#include "../../centrifuge_amd/csrc/cf_inspect_fasta.hpp"
#include "../../centrifuge_amd/csrc/cf_textio.hpp"

namespace cfamd { thread_local EmuCtx g_emu; }
using namespace cfamd;

#ifdef CF_EMU_WAVE64
// ---- the 64-lane wavefront (cf_platform.hpp, CF_EMU_WAVE64): one fiber per lane, a scheduler that runs every lane up to its
//      next cross-lane primitive (or its return) and then forms that primitive's results over the lanes still alive
#include <functional>
#include <ucontext.h>
namespace {
struct EmuWaveRt {
    static constexpr int N = CF_WAVE;
    static constexpr size_t kStack = 512u << 10;
    ucontext_t sched{}, lane[N]{};
    std::vector<char> stacks;
    bool done[N]{}, waiting[N]{};
    int cur = -1;
    int op[N]{}, src[N]{};
    uint64_t in[N]{}, out[N]{};
    std::function<void()> fn;
    uint64_t collectives = 0;
};
thread_local EmuWaveRt *g_wave = nullptr;
void emuLaneMain() {
    EmuWaveRt *w = g_wave;
    const int me = w->cur;
    w->fn();
    w->done[me] = true;
    swapcontext(&w->lane[me], &w->sched);         // never resumed
}
}  // namespace
namespace cfamd {
int emu_wave_lane() { return g_wave ? g_wave->cur : -1; }
uint64_t emu_collective(int op, uint64_t v, int src) {
    EmuWaveRt *w = g_wave;
    const int me = w->cur;
    w->op[me] = op; w->in[me] = v; w->src[me] = src; w->waiting[me] = true;
    swapcontext(&w->lane[me], &w->sched);
    w->cur = me;                                   // (the scheduler set it before it came back here)
    return w->out[me];
}
}  // namespace cfamd
// one wavefront over `fn` (the kernel body with its arguments bound): returns the number of cross-lane primitives it met
static uint64_t emu_run_wave(std::function<void()> fn) {
    auto w = std::make_unique<EmuWaveRt>();
    w->fn = std::move(fn);
    w->stacks.assign(EmuWaveRt::kStack * EmuWaveRt::N, 0);
    for (int l = 0; l < EmuWaveRt::N; l++) {
        getcontext(&w->lane[l]);
        w->lane[l].uc_stack.ss_sp = w->stacks.data() + EmuWaveRt::kStack * (size_t)l;
        w->lane[l].uc_stack.ss_size = EmuWaveRt::kStack;
        w->lane[l].uc_link = nullptr;
        makecontext(&w->lane[l], emuLaneMain, 0);
    }
    EmuWaveRt *const outer = g_wave;
    g_wave = w.get();
    for (;;) {
        for (int l = 0; l < EmuWaveRt::N; l++) {
            if (w->done[l] || w->waiting[l]) continue;
            w->cur = l;
            swapcontext(&w->sched, &w->lane[l]);   // runs until the lane waits at a primitive or returns
            w->cur = -1;
        }
        // the exchanges of lane pairs (cf_swap1: two lanes per chain, both in the same branch) first: they need their partner only
        bool released = false;
        for (int l = 0; l < EmuWaveRt::N; l++) {
            if (w->done[l] || !w->waiting[l] || w->op[l] != EMU_OP_SWAP1) continue;
            const int p = l ^ 1;
            if (w->done[p] || w->op[p] != EMU_OP_SWAP1 || !w->waiting[p]) { std::fprintf(stderr, "emu_run_wave: lane %d exchanges with lane %d, which is elsewhere (done %d, waiting %d, op %d)\n", l, p, (int)w->done[p], (int)w->waiting[p], w->op[p]); std::abort(); }
            if (p < l) continue;                   // (the pair is released once, by its lower lane)
            w->out[l] = w->in[p]; w->out[p] = w->in[l];
            w->waiting[l] = w->waiting[p] = false;
            w->op[l] = w->op[p] = 0;
            released = true;
            w->collectives++;
        }
        if (released) continue;
        // ... then the lanes at a fence: every lane has now reached a fence, a primitive or its end — the lockstep the fence stands for
        for (int l = 0; l < EmuWaveRt::N; l++)
            if (!w->done[l] && w->waiting[l] && w->op[l] == EMU_OP_FENCE) { w->waiting[l] = false; w->op[l] = 0; released = true; }
        if (released) continue;
        int first = -1;
        for (int l = 0; l < EmuWaveRt::N; l++) if (!w->done[l]) { first = l; break; }
        if (first < 0) break;                      // every lane has returned
        uint64_t mask = 0;
        for (int l = 0; l < EmuWaveRt::N; l++) {
            if (w->done[l]) continue;
            if (w->op[l] != w->op[first]) { std::fprintf(stderr, "emu_run_wave: divergent collective (lane %d at op %d, lane %d at op %d)\n", first, w->op[first], l, w->op[l]); std::abort(); }
            if (w->in[l] & 1ull) mask |= 1ull << l;
        }
        for (int l = 0; l < EmuWaveRt::N; l++) {
            if (w->done[l]) continue;
            switch (w->op[l]) {
                case EMU_OP_BALLOT: w->out[l] = mask; break;
                case EMU_OP_FIRST: w->out[l] = w->in[first]; break;
                default: { const int s = w->src[l] & (EmuWaveRt::N - 1); w->out[l] = w->done[s] ? w->in[l] : w->in[s]; break; }   // (a lane that has left: its register holds whatever it held)
            }
            w->waiting[l] = false;
        }
        w->collectives++;
    }
    g_wave = outer;
    return w->collectives;
}
static uint64_t g_waveCollectives = 0;
extern "C" uint64_t emu_wave_collectives() { return g_waveCollectives; }      // cross-lane primitives of the last search (the tests ask that there were some)
extern "C" int emu_wave_lanes() { return CF_WAVE; }
#else
extern "C" uint64_t emu_wave_collectives() { return 0; }
extern "C" int emu_wave_lanes() { return 1; }
#endif

struct EmuIndex {
    HostIndex h;
    std::vector<uint8_t> sides, offs, dense;
    std::vector<RefInfo> refInfo;
    std::vector<uint64_t> wide, text, saPos, isa;
    std::vector<uint8_t> blocks, blocks2;
    std::vector<uint64_t> ftab, eftab;
    IndexTables t;
    DIndex d{};
    uint32_t walkMaxSeen = 0;                 // longest walk of the last emu_densify
    uint32_t restoreShift = 0, restoreMaxSeg = 0;   // the inverse-BWT walks of the last emu_textify
    std::vector<uint32_t> posBucket;          // emu_posify
    std::vector<u64x2> posFrag, posSeq;
};
static uint32_t g_isaExtra = 0;               // emu_textify: the inverse sample that many steps coarser than the SA sample (the device layer: 3 with position-form hits)
static uint32_t g_posShift = 14;              // positions per bucket of posBucket, log2 (the tests shrink it: several fragments per bucket, buckets without one)

static void slurp(std::FILE *f, uint64_t bytes, void *dst) {
    if (std::fread(dst, 1, bytes, f) != bytes) throw std::runtime_error("short read");
}

extern "C" {

void *emu_open(const char *base) {
    try {
        auto ix = std::make_unique<EmuIndex>();
        ix->h.load(base, [&](Section s, std::FILE *f, uint64_t bytes) {
            switch (s) {
                case Section::Sides: ix->sides.resize(bytes + 128); slurp(f, bytes, ix->sides.data()); break;
                case Section::Ftab: ix->ftab.resize(bytes / 8); slurp(f, bytes, ix->ftab.data()); break;
                case Section::Eftab: ix->eftab.resize(bytes / 8); slurp(f, bytes, ix->eftab.data()); break;
                case Section::SaSample: ix->offs.resize(bytes + 8); slurp(f, bytes, ix->offs.data()); break;
            }
        });
        ix->t = makeIndexTables(ix->h);
        DIndex &d = ix->d;
        fillIndexScalars(ix->h, ix->t, d);
        d.sides = ix->sides.data(); d.ftab = ix->ftab.data(); d.eftab = ix->eftab.data(); d.offs = ix->offs.data();
        d.walkOffs = d.offs; d.posRate = -1;
        d.boundRow = ix->h.boundRow.data(); d.boundRef = ix->h.boundRef.data(); d.boundBits = ix->t.boundBits.data();
        ix->refInfo.assign(ix->h.uidTid.size() + 1, RefInfo{0, 0, kNone32});
        for (size_t i = 0; i < ix->h.uidTid.size(); i++) ix->refInfo[i] = RefInfo{ix->h.uidTid[i], ix->t.refTidx[i], ix->t.refPath[i]};
        d.refInfo = ix->refInfo.data();
        d.paths = ix->t.paths.data(); d.pathTidx = ix->t.pathTidx.data();
        return ix.release();
    } catch (const std::exception &e) {
        std::fprintf(stderr, "emu_open: %s\n", e.what());
        return nullptr;
    }
}
void emu_close(void *p) { delete static_cast<EmuIndex *>(p); }

uint64_t emu_num_taxa(void *p) { return static_cast<EmuIndex *>(p)->h.taxa.size(); }
uint64_t emu_taxon_id(void *p, uint64_t i) { return static_cast<EmuIndex *>(p)->h.taxa[i]; }
const char *emu_format_seqid(void *p, uint32_t u, uint64_t t) { return static_cast<EmuIndex *>(p)->h.formatSeqId(u, t); }

uint64_t emu_rank(void *p, int c, uint64_t row) {
    uint64_t t, b; bool two;
    rank_pair<1>(static_cast<EmuIndex *>(p)->d, c, row, row, t, b, two);
    return t;
}

// LF(c0, LF(c1, row)) from the pair planes (rows 0 .. len + 1)
uint64_t emu_pair_rank(void *p, int c1, int c0, uint64_t row) {
    const EmuIndex &ix = *static_cast<EmuIndex *>(p);
    uint64_t g = row >> 6; uint32_t o = (uint32_t)row & 63u;
    const uint64_t nGroups = (ix.h.g.len + 64) / 64;
    if (g >= nGroups) { g = nGroups - 1; o = (uint32_t)(row - 64 * g); }        // (row = 64 g: the whole of the last group)
    const uint64_t *e = reinterpret_cast<const uint64_t *>(ix.d.planes2 + g * 256 + 16 * (4 * c1 + c0));
    return e[1] + popc_below(e[0], o);
}
uint64_t emu_num_rows(void *p) { return static_cast<EmuIndex *>(p)->h.g.len + 1; }

static int g_searchVersion = 2;
static uint32_t g_verifyMinRun = 0;
static uint32_t g_lazyHits = 1;               // classification runs hold hits back as the device does; the search tap never
static int g_walkVersion = 3;                  // 3 = one lane per row (the batch walk), 2 = the chain kernel
static uint64_t g_rowsCap = ~0ull >> 1;          // rows per pass of the row stage (tests shrink it to drive several passes)
static uint32_t g_postScratch = 24, g_scoreScratchRows = 16;    // the general kernels' per-lane scratch (k_post / k_score keep it in LDS); 0 = in place
static int g_revWords = 1;                       // ... and finds the forward strands' words in search order beside the packed reads (DBatch::revBases, rev_word; 0: the in-kernel transform for every read)
static int g_selfRecords = 1;                    // the one-lane kernel builds its strand records from the packed reads (else: pack_body's)
static uint32_t g_countSlotBits = 0;            // 0: the product's slot count; small = probing and overflow to the far atomics
static int g_postFast = 1, g_scoreFast = 1;      // the common-case kernels first (as the device layer launches them), or the general ones alone
static int g_earlyScore = 0;                     // the common-case score kernel right behind the common-case post kernel (enqueuePost's early mode: CF_EARLY_SCORE=1, off by default — measured a loss)

struct Work {
    BatchPlan plan;
    std::vector<uint8_t> seq, recs;
    std::vector<uint64_t> off, qBase, rowVal, bases, woff;
    std::vector<uint32_t> seeds, nhml, rowRef, nOut, score2, nmask, rlen, qRows, slowPost, slowScore, itemMeta;
    std::vector<unsigned long long> cursor;
    BatchStatus st{};
    std::vector<HitP> hits;
    std::vector<PlanHit> qplan;
    std::vector<QHead> qhead;
    std::vector<uint32_t> qflag;
    std::vector<uint64_t> o1tax, o1a, o1b;
    std::vector<HmEntry> hm;
    std::vector<TcEntry> tc;
    std::vector<OutRow> out;
    std::vector<unsigned long long> counts;
    OpCounts ops{};
    DBatch d{};
};

// the reads as the device holds them: wcount_body -> scan -> convert_body, as cf_batch_create launches them
static void packReads(const uint8_t *seq, const uint64_t *off, uint64_t nReads, Work &w) {
    const uint64_t nbases = off[nReads];
    w.seq.assign(nbases + 16, 0);
    if (nbases) std::memcpy(w.seq.data(), seq, nbases);
    w.off.assign(off, off + nReads + 1);
    w.rlen.assign(nReads + 1, 0);
    for (uint32_t r = 0; r < nReads + 3; r++) rlen_body(w.off.data(), w.rlen.data(), (uint32_t)nReads, r);
    w.woff.assign(nReads + 1, 0);
    uint64_t t = 0;
    for (uint64_t r = 0; r <= nReads; r++) { w.woff[r] = t; if (r < nReads) t += ((uint64_t)w.rlen[r] + 31) >> 5; }
    w.bases.assign(t + 16, 0xdeadbeefdeadbeefull); w.nmask.assign(t + 16, 0xdeadbeefu);   // poison: every word must be written
    DConvert c{w.seq.data(), w.off.data(), w.woff.data(), w.bases.data(), w.nmask.data(), (uint32_t)nReads};
    for (uint32_t r = 0; r < nReads + 3; r++) convert_body(c, r);
}

static void setup(EmuIndex &ix, const DParams &pr, const uint8_t *seq, const uint64_t *off, const uint32_t *seeds,
                  uint64_t nReads, int paired, Work &w) {
    w.plan = makeBatchPlan(seq, off, nReads, ix.h.g.ftabChars);
    packReads(seq, off, nReads, w);
    w.seeds.assign(seeds, seeds + nReads); w.seeds.push_back(0);
    const uint64_t nQ = paired ? nReads / 2 : nReads;
    w.hits.resize(w.plan.hitsTotal + 1);
    w.nhml.assign(2 * w.plan.items.size() + 1, 0);
    w.qflag.assign(nQ + 1, 0xdeadbeefu); w.qhead.resize(nQ + 1); w.qplan.assign((nQ + 1) * kInlinePlan, PlanHit{0xdeaddeaddeadull, 0xdeadu, 0xdeadu});
    w.o1tax.assign((nQ + 1) * kFieldRows, 0xdead); w.o1a.assign((nQ + 1) * kFieldRows, 0xdead); w.o1b.assign((nQ + 1) * kFieldRows, 0xdead);
    w.qRows.assign(nQ + 1, 0); w.qBase.assign(nQ + 1, 0);
    w.out.resize(nQ * pr.k + 1); w.nOut.assign(nQ + 1, 0); w.score2.assign(nQ + 1, 0);
    w.cursor.assign(4, 0);
    w.slowPost.assign(nQ + 1, 0xdeadbeefu); w.slowScore.assign(nQ + 1, 0xdeadbeefu);
    w.counts.assign(2 * ix.h.taxa.size(), 0);
    w.st = BatchStatus{};
    w.st.nItems = (uint32_t)(2 * w.plan.items.size());
    DBatch &d = w.d;
    d.bases = w.bases.data(); d.nmask = w.nmask.data(); d.rlen = w.rlen.data(); d.woff = w.woff.data();
    d.seeds = w.seeds.data(); d.pass = w.plan.pass.data();
    d.items = w.plan.items.data(); d.slotOf = w.plan.slotOf.data(); d.hitBase = w.plan.hitBase.data();
    d.hitCap = w.plan.hitCap.data(); d.hits = w.hits.data(); d.nhml = w.nhml.data(); d.qflag = w.qflag.data(); d.qhead = w.qhead.data(); d.qplan = w.qplan.data(); d.qplanStride = nQ + 1;
    d.o1tax = w.o1tax.data(); d.o1a = w.o1a.data(); d.o1b = w.o1b.data(); d.oStride = nQ + 1;
    d.qRows = w.qRows.data(); d.qBase = w.qBase.data(); d.out = w.out.data(); d.nOut = w.nOut.data();
    d.score2 = w.score2.data(); d.counts = w.counts.data(); d.nTaxa = (uint32_t)ix.h.taxa.size();
    d.nReads = (uint32_t)nReads; d.nQueries = (uint32_t)nQ;
    d.paired = paired; d.cursor = w.cursor.data(); d.ops = &w.ops; d.st = &w.st;
    d.hitsCap = w.plan.hitsTotal; d.rowsCap = g_rowsCap;
    d.slowPost = w.slowPost.data(); d.slowScore = w.slowScore.data();
}

// the search stage: k_search2's body (strand records, one-lane chains) when the reads fit its
// records, else k_search's byte-window body — the same selection as the device layer
static void runSearch(EmuIndex &ix, const DParams &pr, Work &w) {
    uint32_t W = g_searchVersion == 2 ? w.plan.recWords() : 0;
    if ((uint64_t)(0.15 * w.plan.maxLen) + w.plan.maxLen / (uint64_t)std::max(1, ix.h.g.ftabChars) + 3 >= 255) W = 0;   // as the device layer
    if (W && w.st.nItems) {
        w.d.recWords = W; w.d.recs = nullptr; w.d.itemMeta = nullptr;
        if (ix.d.planes && g_selfRecords) {           // as the device layer: the one-lane kernel over the planes makes its records itself
            w.itemMeta.assign(4 * (size_t)w.st.nItems + 8, 0xdeadbeefu);
            for (uint32_t it = 0; it < w.st.nItems; it++) {
                const uint32_t rd = w.plan.items[it >> 1];
                uint32_t *m = w.itemMeta.data() + 4 * (size_t)it;
                uint32_t any = 0;                     // (as plan_fill_body: does the read hold an N)
                for (uint32_t k = 0; 32 * k < w.rlen[rd]; k++) {
                    uint32_t mk = w.nmask[w.woff[rd] + k];
                    if (w.rlen[rd] - 32 * k < 32) mk &= (1u << (w.rlen[rd] - 32 * k)) - 1u;
                    any |= mk;
                }
                m[0] = (uint32_t)w.woff[rd]; m[1] = w.rlen[rd] | (any ? kItemHasN : 0u); m[2] = (uint32_t)(w.plan.hitBase[rd] + ((it & 1) ? w.plan.hitCap[rd] : 0u)); m[3] = rd;
            }
            w.d.itemMeta = w.itemMeta.data();
            if (g_revWords) {                         // as k_plan_fill: the kernel's own plan_fill_body over every read; the forward words behind the reads' own, poisoned first
                const uint32_t nW = (uint32_t)w.woff[w.d.nReads];
                w.bases.resize(nW + 16); w.bases.resize(2 * (size_t)(nW + 16), 0xfeedfacefeedfaceull);
                w.d.bases = w.bases.data();
                DPlan p{};
                std::vector<uint8_t> pass = w.plan.pass;                       // (plan_fill_body writes items and retires the slots of skipped reads: copies)
                std::vector<uint32_t> slotOf = w.plan.slotOf, hitCap = w.plan.hitCap, items = w.plan.items;
                std::vector<uint64_t> hitBase = w.plan.hitBase;
                pass.resize(w.d.nReads + 1); slotOf.resize(w.d.nReads + 1); hitCap.resize(w.d.nReads + 1); items.resize(w.d.nReads + 1); hitBase.resize(w.d.nReads + 1);
                BatchStatus st{};
                p.nmask = w.nmask.data(); p.rlen = w.rlen.data(); p.woff = w.woff.data(); p.nReads = (uint32_t)w.d.nReads; p.pass = pass.data(); p.hitCap = hitCap.data();
                p.slotOf = slotOf.data(); p.hitBase = hitBase.data(); p.items = items.data(); p.itemMeta = w.itemMeta.data(); p.st = &st; p.hitsCap = ~0ull;
                p.bases = w.bases.data(); p.revDelta = nW + 16;
                const std::vector<uint32_t> byHand(w.itemMeta.begin(), w.itemMeta.begin() + 4 * (size_t)w.st.nItems);
                for (uint32_t r = 0; r < w.d.nReads; r++) plan_fill_body(p, r);
                for (uint64_t t = 0; t < (uint64_t)(w.d.nReads + 2) * W; t++) rev_words_body(p, W, t);        // (k_rev_words, grid rounded up)
                for (uint32_t it = 0; it < w.st.nItems; it++) {                  // the two ways of making the item records agree (but for the round-6 fields)
                    const uint32_t *a = byHand.data() + 4 * (size_t)it, *m = w.itemMeta.data() + 4 * (size_t)it;
                    const bool pre = (m[1] & kItemPre) != 0;
                    if (m[1] != (a[1] | (pre ? kItemPre : 0u)) || m[2] != a[2] || m[3] != a[3] || m[0] != a[0] + ((pre && !(it & 1)) ? p.revDelta : 0u) || (pre && (m[1] & kItemHasN))) std::abort();
                }
            }
        } else {
            w.recs.assign((size_t)w.st.nItems * rec_bytes((int)W), 0);
            for (uint32_t t = 0; t < (w.st.nItems + 3) * W; t++) pack_body(w.d, w.recs.data(), W, t);
            w.d.recs = w.recs.data();
        }
        std::vector<uint8_t> lds((size_t)CF_WAVE * (rec_lds_stride((int)W) + 4 * RankTab<1>::WORDS + 4 * RankTab<2>::WORDS + 16 * kLazyHits) + 64, 0);
#ifdef CF_EMU_WAVE64
        // the wavefront of 64 lanes: every lane runs the body, the cross-lane primitives are rendezvous (emu_run_wave)
#define SEARCH2(...) g_waveCollectives = emu_run_wave([&] { search2_body<__VA_ARGS__>(ix.d, pr, w.d, lds.data()); })
#else
#define SEARCH2(...) search2_body<__VA_ARGS__>(ix.d, pr, w.d, lds.data())
#endif
        if (ix.d.planes) {
            if (ix.d.multiRows && W == 4) SEARCH2(1, 4, true, true, 0, true);     // as the device layer: kernels of their own
            else if (ix.d.multiRows && W == 6) SEARCH2(1, 6, true, true, 0, true);
            else if (ix.d.multiRows) SEARCH2(1, 8, true, true, 0, true);
            else if (W == 4) SEARCH2(1, 4, true, true);
            else if (W == 6) SEARCH2(1, 6, true, true);
            else SEARCH2(1, 8, true, true);
        }
#ifdef CF_EMU_WAVE64
        // over the sides the device runs TWO lanes per chain (k_search2<2, W>: each lane loads half a side, the pair sums its counts
        // with DPP swaps): 32 chains per wavefront here as there
        else if (w.d.recs && W == 4) SEARCH2(2, 4, true);
        else if (w.d.recs && W == 6) SEARCH2(2, 6, true);
        else if (w.d.recs) SEARCH2(2, 8, true);
#endif
        else if (W == 4) SEARCH2(1, 4, true);
        else if (W == 6) SEARCH2(1, 6, true);
        else SEARCH2(1, 8, true);
#undef SEARCH2
    } else search_body<1>(ix.d, pr, w.d);
}

static uint32_t g_lastSlowPost = 0, g_lastSlowScore = 0;
void emu_set_search_version(int v) { g_searchVersion = v; }
void emu_set_fast_kernels(int post, int score) { g_postFast = post; g_scoreFast = score; }
void emu_set_count_slot_bits(unsigned bits) { g_countSlotBits = bits > kCountSlotBits ? kCountSlotBits : bits; }
void emu_set_self_records(int on) { g_selfRecords = on; }
void emu_set_rev_words(int on) { g_revWords = on; }
void emu_set_general_scratch(uint32_t postHits, uint32_t scoreRows) { g_postScratch = postHits; g_scoreScratchRows = scoreRows; }
void emu_last_slow(uint32_t *post, uint32_t *score) { *post = g_lastSlowPost; *score = g_lastSlowScore; }
void emu_set_verify_min_run(uint32_t v) { g_verifyMinRun = v; }
static uint32_t g_multiRows = 0, g_multiMinRun = 2;
void emu_set_multi_verify(uint32_t rows, uint32_t minRun) { g_multiRows = rows > 15 ? 15 : rows; g_multiMinRun = minRun; }
void emu_set_walk_version(int v) { g_walkVersion = v; }
void emu_set_lazy_hits(uint32_t v) { g_lazyHits = v; }
static int g_directRefs = 1;
void emu_set_multi_verify(uint32_t rows, uint32_t minRun);
void emu_set_direct_refs(int on) { g_directRefs = on; }
// the occurrence planes (occ_planes_body); on = 0 drops them again (the search then reads the sides)
int emu_planify(void *p, int on) {
    EmuIndex &ix = *static_cast<EmuIndex *>(p);
    ix.d.planes = nullptr; ix.d.planes2 = nullptr;
    ix.d.sides = ix.sides.data();                       // (the planes are made from the sides)
    if (!on) return 1;
    const uint64_t nSides = ix.h.g.numSides;
    ix.blocks.assign(nSides * 384 + 64, 0xee);
    for (uint64_t s = 0; s < nSides + 3; s++) occ_planes_body(ix.d, ix.blocks.data(), s, nSides);
    ix.d.planes = ix.blocks.data();
    return 1;
}

// the sides out of the device view (cf_index_options::sides = -1): needs the planes; on = 0 puts them back
int emu_drop_sides(void *p, int on) {
    EmuIndex &ix = *static_cast<EmuIndex *>(p);
    if (on && !ix.d.planes) return 0;
    ix.d.sides = on ? nullptr : ix.sides.data();
    return 1;
}

// the pair planes (pair_planes_body) from the planes; on = 0 (or no planes) drops them
int emu_planify2(void *p, int on) {
    EmuIndex &ix = *static_cast<EmuIndex *>(p);
    ix.d.planes2 = nullptr;
    if (!on || !ix.d.planes) return on ? 0 : 1;
    const uint64_t nGroups = (ix.h.g.len + 64) / 64;
    ix.blocks2.assign(nGroups * 256 + 64, 0xee);
    for (uint64_t g = 0; g < nGroups + 3; g++) pair_planes_body(ix.d, ix.blocks2.data(), g, nGroups);
    ix.d.planes2 = ix.blocks2.data();
    return 1;
}

// the dense resolve table as the device layer makes it at load time: walk2_body in its table-building mode from every
// 2^rate-th row with the file's sample; rate >= offRate (or < 0) goes back to the file's sample
int emu_densify(void *p, int rate) {
    EmuIndex &ix = *static_cast<EmuIndex *>(p);
    DIndex &d = ix.d;
    d.walkOffs = d.offs; d.walkRate = d.offRate;
    if (rate < 0 || rate >= d.offRate) return 0;
    const uint64_t count = (d.len >> rate) + 1;
    std::vector<uint8_t> table((count + 2) * (d.offw ? 4 : 2), 0xee);
    unsigned long long cursor[4] = {0, 0, 0, 0};
    BatchStatus st{};
    st.rowLo = 0; st.rowHi = count;
    DBatch b{};
    b.rowRef = reinterpret_cast<uint32_t *>(table.data()); b.cursor = cursor; b.st = &st; b.genShift = (uint32_t)rate;
    uint32_t walkMax = 0;
    b.walkMaxOut = &walkMax;
    if (d.offw) walk2_body<1, false, WALK_TABLE32>(d, b); else walk2_body<1, false, WALK_TABLE16>(d, b);
    ix.dense.swap(table);
    d.walkOffs = ix.dense.data(); d.walkRate = rate;
    ix.walkMaxSeen = walkMax;
    d.posFrag = nullptr;                                // (made again by emu_posify: it rests on this table's longest walk)
    return 1;
}
// position -> reference (DIndex::posFrag, as makePosTables of the device layer): needs the text tables (any sample rate) and a bound
// on the walk-left.  on = 0 takes the tables away again (no hit then takes the position form); returns 1 when made
int emu_posify(void *p, int on) {
    EmuIndex &ix = *static_cast<EmuIndex *>(p);
    DIndex &d = ix.d;
    d.posFrag = nullptr; d.posBucket = nullptr; d.posSeq = nullptr; d.nPosFrag = 0; d.walkMax = 0; d.posShift = 14;
    if (!on || d.posRate < 0 || !d.isa) return 0;
    uint32_t bound;                                      // as makePosTables: exact from a resolve table at every row, else the longest restore segment
    if (d.walkRate == 0 && d.walkOffs != d.offs) bound = ix.walkMaxSeen;
    else if (ix.restoreMaxSeg && (int)ix.restoreShift >= d.offRate) bound = ix.restoreMaxSeg;
    else return 0;
    const uint64_t n = d.len, nFrag = ix.h.rstarts.size() / 3;
    if (nFrag == 0) return 0;
    ix.posFrag.assign(nFrag, u64x2{0, 0}); ix.posSeq.assign(ix.h.nPat + 1, u64x2{0, 0});
    for (uint64_t i = 0; i < nFrag; i++) ix.posFrag[i] = u64x2{ix.h.rstarts[3 * i], ix.h.rstarts[3 * i + 1]};
    for (uint64_t i = 0; i < nFrag; i++) {
        const uint64_t sq = ix.posFrag[i].y;
        if (i == 0 || ix.posFrag[i - 1].y != sq) {
            uint64_t j = i;
            while (j < nFrag && ix.posFrag[j].y == sq) j++;
            ix.posSeq[sq] = u64x2{ix.posFrag[i].x, j < nFrag ? ix.posFrag[j].x : n};
        }
    }
    const uint32_t sh = g_posShift;
    const uint64_t nB = (n >> sh) + 2;
    ix.posBucket.assign(nB + 1, 0);
    uint64_t f = 0;
    for (uint64_t b = 0; b <= nB; b++) { while (f + 1 < nFrag && ix.posFrag[f + 1].x <= (b << sh)) f++; ix.posBucket[b] = (uint32_t)f; }
    d.posBucket = ix.posBucket.data(); d.posFrag = ix.posFrag.data(); d.posSeq = ix.posSeq.data();
    d.posShift = sh; d.nPosFrag = (uint32_t)nFrag; d.walkMax = bound;
    return 1;
}
void emu_set_pos_shift(uint32_t sh) { g_posShift = sh; }
void emu_set_isa_extra(uint32_t steps) { g_isaExtra = steps; }
uint32_t emu_walk_max(void *p) { return static_cast<EmuIndex *>(p)->walkMaxSeen; }
void emu_set_rows_cap(uint64_t v) { g_rowsCap = v ? v : (~0ull >> 1); }
void emu_set_early_score(int on) { g_earlyScore = on; }

int emu_classify(void *p, const cf_params *cp, const uint8_t *seq, const uint64_t *off, const uint32_t *seeds,
                 uint64_t nReads, int paired, cf_row *rows, uint32_t *nRows, uint32_t *score2, cf_opcounts *ops,
                 uint64_t *countsOut) {
    try {
        EmuIndex &ix = *static_cast<EmuIndex *>(p);
        DParams pr;
        const ClassifierTables ct = makeClassifier(ix.h, *cp, pr);
        if (!ct.refExcluded.empty()) pr.refExcluded = ct.refExcluded.data();
        if (!ct.hostSet.empty()) { pr.hostSet = ct.hostSet.data(); pr.nHostSet = (uint32_t)ct.hostSet.size(); }
        Work w;
        setup(ix, pr, seq, off, seeds, nReads, paired, w);
        g_emu.tid = 0; g_emu.nthreads = 1;
        w.d.lazyHits = g_lazyHits | (ix.d.posFrag ? 4u : 0u);        // (as bindBatch: hits in the position form where the index resolves positions)
        // (what a held-back hit would have overwritten must not look like a hit: the pool starts out poisoned)
        if (g_lazyHits) for (auto &h : w.hits) { h.w0 = 0xdeaddeaddeaddeadull; h.w1 = 0xdeaddeaddeaddeadull; }
        runSearch(ix, pr, w);
        g_lastSlowScore = 0;
        // k_post_fast over every query, then k_post over the queries it listed
        w.st.nSlowPost = 0;
        std::vector<uint8_t> pdef(w.d.nQueries + 1, 0xee);
        w.d.postDeferred = pdef.data();
        for (uint32_t q = 0; q < w.d.nQueries; q++) {
            const bool df = g_postFast ? post_fast_body(ix.d, pr, w.d, q) : true;
            pdef[q] = df ? 1 : 0;
            defer_push(w.d.slowPost, &w.st.nSlowPost, df, q);
        }
        // as the device layer (enqueuePost): with the resolve table at every row the common-case score kernel runs HERE — behind the
        // common-case post kernel, before (on the device: beside) the general one, before the rows are counted — and leaves its list
        const bool early = g_earlyScore && g_postFast && g_scoreFast && g_directRefs && ix.d.walkRate == 0 && ix.d.walkOffs != ix.d.offs && w.d.nQueries;
        w.st.nSlowScore = 0;
        if (early) {
            w.d.directRefs = 1;
            // (what the general post kernel has not written yet must not be read: poison the plan of the queries left to it)
            for (uint32_t q = 0; q < w.d.nQueries; q++) if (pdef[q]) { w.qflag[q] = 0xeeeeeeeeu; w.qRows[q] = 0xeeeeeeeu; }
            for (uint32_t q = 0; q < w.d.nQueries; q++) defer_push(w.d.slowScore, &w.st.nSlowScore, score_fast_body<true>(ix.d, pr, w.d, q), q);
        }
        {   // as k_post: a lane's scratch for the mate's hit lists (g_postScratch records; lists that do not fit are worked on in place)
            std::vector<HitP> scratch(g_postScratch + 1);
            for (auto &h : scratch) { h.w0 = 0xa5a5a5a5a5a5a5a5ull; h.w1 = 0x5a5a5a5a5a5a5a5aull; }
            for (uint32_t i = 0; i < w.st.nSlowPost; i++) post_body(ix.d, pr, w.d, w.d.slowPost[i], g_postScratch ? scratch.data() : nullptr, g_postScratch);
        }
        g_lastSlowPost = w.st.nSlowPost;
        uint64_t total = 0;
        for (uint32_t q = 0; q <= w.d.nQueries; q++) { w.qBase[q] = total; total += w.qRows[q]; }
        total = w.qBase[w.d.nQueries];
        // the row stage, pass by pass as cf_batch_wait drives it: window -> emit -> walk -> score
        uint32_t qLo = 0;
        bool firstPass = true;
        do {
            const bool earlyPass = early && firstPass;
            firstPass = false;
            row_window_body(w.d, qLo, earlyPass);
            if (w.st.qHi == w.st.qLo && w.st.qLo < w.d.nQueries) { w.d.rowsCap = w.st.needRows; row_window_body(w.d, qLo, earlyPass); }   // grow to the one query that does not fit
            const uint64_t rows = w.st.rowHi - w.st.rowLo;
            w.rowVal.assign(rows + 1, 0); w.rowRef.assign(rows + 1, 0); w.hm.assign(rows + 1, HmEntry{}); w.tc.assign(rows + 1, TcEntry{});
            w.d.rowVal = w.rowVal.data(); w.d.rowRef = w.rowRef.data(); w.d.hm = w.hm.data(); w.d.tc = w.tc.data();
            w.cursor[1] = 0;
            // as the device layer: with the resolve table at every row the common-case score kernel reads references from it, and only
            // the queries it leaves get their rows resolved (no emit, no walk)
            w.d.directRefs = (uint32_t)(g_directRefs && g_scoreFast && ix.d.walkRate == 0 && ix.d.walkOffs != ix.d.offs);
            if (!w.d.directRefs) {
                for (uint32_t q = 0; q < w.d.nQueries; q++) emit_body(pr, w.d, q);
                if (g_walkVersion == 2) walk2_body<1, true>(ix.d, w.d);
                else for (uint64_t i = 0; i < rows + 3; i++) walk3_body<true>(ix.d, w.d, i);
            } else { std::fill(w.rowVal.begin(), w.rowVal.end(), 0xeeeeeeeeeeeeeeeeull); std::fill(w.rowRef.begin(), w.rowRef.end(), 0xeeeeeeeeu); }
            if (!earlyPass) for (uint32_t q = 0; q < w.d.nQueries; q++) defer_push(w.d.slowScore, &w.st.nSlowScore, g_scoreFast ? score_fast_body(ix.d, pr, w.d, q) : true, q);
            if (w.d.directRefs) for (uint32_t i = 0; i < w.st.nSlowScore; i++) resolve_query_body(ix.d, pr, w.d, w.d.slowScore[i]);
            {   // as k_score: a lane's scratch for hit map, parent counts and references (queries of up to g_scoreScratchRows rows)
                std::vector<uint64_t> scratch(score_scratch_bytes(g_scoreScratchRows) / 8 + 2, 0xa5a5a5a5a5a5a5a5ull);
                for (uint32_t i = 0; i < w.st.nSlowScore; i++)
                    score_body(ix.d, pr, w.d, w.d.slowScore[i], g_scoreScratchRows ? reinterpret_cast<uint8_t *>(scratch.data()) : nullptr, g_scoreScratchRows);
            }
            g_lastSlowScore += w.st.nSlowScore;
            {   // k_count: one block per chunk of queries
                std::vector<uint32_t> slots(3 * kCountSlots, 0xabababab);
                const uint32_t bits = g_countSlotBits ? g_countSlotBits : kCountSlotBits;
                for (uint32_t c = 0; c * kCountChunk < w.d.nQueries + 1; c++) count_body(w.d, slots.data(), c, bits, w.d.nTaxa <= (1u << bits));
            }
            qLo = w.st.qHi;
        } while (qLo < w.d.nQueries);
        static_assert(sizeof(cf_row) == sizeof(OutRow), "row layout");
        std::memcpy(rows, w.out.data(), (size_t)w.d.nQueries * pr.k * sizeof(OutRow));
        for (uint32_t q = 0; q < w.d.nQueries; q++)                  // up to kFieldRows rows of a query lie by field (compact_body reads them there)
            if (w.nOut[q] <= kFieldRows) for (uint32_t i = 0; i < w.nOut[q]; i++) {
                const size_t at = (size_t)i * w.d.oStride + q;
                const OutRow o = row_of_one(w.o1tax[at], w.o1a[at], w.o1b[at]); std::memcpy(rows + (size_t)q * pr.k + i, &o, sizeof o);
            }
        std::memcpy(nRows, w.nOut.data(), (size_t)w.d.nQueries * 4);
        std::memcpy(score2, w.score2.data(), (size_t)w.d.nQueries * 4);
        if (ops) {
            ops->n_ftab = w.ops.nFtab; ops->n_pair = w.ops.nPair; ops->n_pair2 = w.ops.nPair2;
            ops->n_single = w.ops.nSingle; ops->n_walk = w.ops.nWalk; ops->n_rows = total; ops->n_ftab_wide = w.ops.nFtabWide; ops->n_verify = w.ops.nVerify; ops->n_text_loads = w.ops.nTextLoads; ops->n_pos_hits = w.ops.nPosHits;
        }
        if (countsOut) std::memcpy(countsOut, w.counts.data(), w.counts.size() * 8);
        return 0;
    } catch (const std::exception &e) {
        std::fprintf(stderr, "emu_classify: %s\n", e.what());
        return 1;
    }
}

// hit lists of one read after search + extend/twin/trim (the cf_debug_search tap)
int emu_search(void *p, const cf_params *cp, const uint8_t *seq, uint64_t len, cf_hit *hf, cf_hit *hr,
               uint32_t maxHits, uint32_t nhits[2]) {
    EmuIndex &ix = *static_cast<EmuIndex *>(p);
    DParams pr;
    makeClassifier(ix.h, *cp, pr);
    const uint64_t off[2] = {0, len};
    const uint32_t seed = 0;
    Work w;
    setup(ix, pr, seq, off, &seed, 1, 0, w);
    nhits[0] = nhits[1] = 0;
    if (w.st.nItems == 0) return 0;
    w.d.lazyHits = 0;
    runSearch(ix, pr, w);
    post_fix(ix.d, pr, w.d, 0);
    cf_hit *o[2] = {hf, hr};
    for (int f = 0; f < 2; f++) {
        nhits[f] = nhml_n(w.nhml[f]);
        for (uint32_t i = 0; i < nhits[f] && i < maxHits; i++) {
            const Hit h = hit_unpack(w.hits[(size_t)f * w.plan.hitCap[0] + i]);
            o[f][i].top = h.top; o[f][i].bot = h.bot; o[f][i].bwoff = h.bwoff; o[f][i].len = h.len;
        }
    }
    return 0;
}

uint32_t emu_resolve(void *p, uint64_t row) {
    EmuIndex &ix = *static_cast<EmuIndex *>(p);
    uint32_t ref;
    while (!try_offset(ix.d, row, ref)) row = lf_own<1>(ix.d, row);
    return ref;
}

// the wide ftab as the device layer makes it at load time (wide_ftab_body over all 4^k wide-mers); k <= ftabChars: off
static uint64_t g_wideCap = kWideSizeMax;     // ranges this large are left to the step-by-step path
void emu_set_wide_cap(uint64_t c) { g_wideCap = c < kWideSizeMax ? c : kWideSizeMax; }
int emu_widen(void *p, int k) {
    EmuIndex &ix = *static_cast<EmuIndex *>(p);
    ix.d.wide = nullptr; ix.d.wideChars = 0;
    if (k <= ix.d.ftabChars || k > 13) return 0;
    const uint64_t entries = 1ull << (2 * k);
    ix.wide.assign(entries + 2, 0xeeeeeeeeeeeeeeeeull);
    for (uint64_t t = 0; t < entries + 5; t++) wide_ftab_body(ix.d, (uint32_t)k, ix.wide.data(), t, g_wideCap);
    ix.d.wide = ix.wide.data(); ix.d.wideChars = k;
    return 1;
}

// one row through walk2_body as a batch walk (uses the dense table when emu_densify made one)
uint32_t emu_resolve_walk(void *p, uint64_t row) {
    EmuIndex &ix = *static_cast<EmuIndex *>(p);
    unsigned long long cursor[4] = {0, 0, 0, 0};
    BatchStatus st{};
    st.rowLo = 0; st.rowHi = 1;
    uint64_t rv = row; uint32_t ref = 0xdeadbeefu;
    DBatch b{};
    b.rowVal = &rv; b.rowRef = &ref; b.cursor = cursor; b.st = &st;
    walk2_body<1, false>(ix.d, b);
    return ref;
}

void emu_sort_hits(cf_hit *hits, uint32_t n) {
    std::vector<HitP> t(n + 1);
    for (uint32_t i = 0; i < n; i++) { Hit h; h.top = hits[i].top; h.bot = hits[i].bot; h.bwoff = hits[i].bwoff; h.len = hits[i].len; h.nelt = 0; t[i] = hit_pack(h); }
    std_sort_hits(t.data(), (int)n);
    for (uint32_t i = 0; i < n; i++) { const Hit h = hit_unpack(t[i]); hits[i].top = h.top; hits[i].bot = h.bot; hits[i].bwoff = h.bwoff; hits[i].len = h.len; }
}

// the same list through libstdc++'s own std::sort with the same comparator (what the reference runs, ds.h:775-779)
void emu_std_sort_hits(cf_hit *hits, uint32_t n) {
    std::vector<HitP> t(n);
    for (uint32_t i = 0; i < n; i++) { Hit h; h.top = hits[i].top; h.bot = hits[i].bot; h.bwoff = hits[i].bwoff; h.len = hits[i].len; h.nelt = 0; t[i] = hit_pack(h); }
    std::sort(t.begin(), t.end(), [](const HitP &a, const HitP &b) { return hit_less(a, b); });
    for (uint32_t i = 0; i < n; i++) { const Hit h = hit_unpack(t[i]); hits[i].top = h.top; hits[i].bot = h.bot; hits[i].bwoff = h.bwoff; hits[i].len = h.len; }
}

// The device-side batch plan (plan_body -> scans -> plan_fill_body, plan_maxscore_body) against the host
// plan the rest of this harness uses (makeBatchPlan) and the max_score rule of classifier.h:530-536.
// Returns 0 when every array agrees, else the number of the first array that differs.
int emu_plan_check(const uint8_t *seq, const uint64_t *off, uint64_t nReads, int ftabChars, int paired) {
    const BatchPlan hp = makeBatchPlan(seq, off, nReads, ftabChars);
    Work w;
    packReads(seq, off, nReads, w);
    // the packed reads must say what the bytes say: lengths, codes (N -> 0), N bits, nothing past a read's end
    for (uint64_t r = 0; r < nReads; r++) {
        const uint64_t L = off[r + 1] - off[r];
        if (w.rlen[r] != L) return 20;
        for (uint64_t i = 0; i < 32 * (w.woff[r + 1] - w.woff[r]); i++) {
            const uint64_t wi = w.woff[r] + (i >> 5);
            const uint32_t code = (uint32_t)((w.bases[wi] >> (2 * (i & 31))) & 3), nb = (w.nmask[wi] >> (i & 31)) & 1u;
            const uint8_t c = i < L ? seq[off[r] + i] : 0;
            if (code != (c > 3 ? 0u : c) || nb != (c > 3 ? 1u : 0u)) return 21;
        }
    }
    std::vector<uint8_t> pass(nReads + 1, 9);
    std::vector<uint32_t> hitCap(nReads + 1, 77), slotOf(nReads + 1, 77), items(nReads + 1, 77);
    std::vector<uint64_t> hitBase(nReads + 1, 77);
    BatchStatus st{};
    DPlan p{};
    p.nmask = w.nmask.data(); p.rlen = w.rlen.data(); p.woff = w.woff.data();
    p.nReads = (uint32_t)nReads; p.ftabChars = ftabChars; p.maxLenAllowed = 0xffffffffu; p.pass = pass.data(); p.hitCap = hitCap.data();
    p.slotOf = slotOf.data(); p.hitBase = hitBase.data(); p.items = items.data();
    p.st = &st; p.hitsCap = hp.hitsTotal; p.nWords = w.woff[nReads];
    for (uint32_t r = 0; r < nReads + 7; r++) plan_body(p, r);               // a grid rounded up past nReads + 1
    uint32_t a = 0; uint64_t b = 0;
    for (uint64_t r = 0; r <= nReads; r++) { slotOf[r] = a; hitBase[r] = b; if (r < nReads) { a += hitCap[r] != 0; b += 2ull * hitCap[r]; } }   // SCAN_HITS
    for (uint32_t r = 0; r < nReads + 7; r++) plan_fill_body(p, r);
    if (st.nItems != 2 * hp.items.size()) return 1;
    if (st.hitsNeed != hp.hitsTotal || st.flags) return 2;
    for (uint64_t r = 0; r < nReads; r++) {
        if (pass[r] != hp.pass[r]) return 4;
        if (slotOf[r] != hp.slotOf[r]) return 5;
        if (hp.pass[r] && (hitCap[r] != hp.hitCap[r] || hitBase[r] != hp.hitBase[r])) return 6;
    }
    for (uint32_t i = 0; i < st.nItems / 2; i++) if (items[i] != hp.items[i]) return 7;
    auto readHasN = [&](uint32_t rd) {
        for (uint32_t k = 0; 32 * k < w.rlen[rd]; k++) {
            uint32_t mk = w.nmask[w.woff[rd] + k];
            if (w.rlen[rd] - 32 * k < 32) mk &= (1u << (w.rlen[rd] - 32 * k)) - 1u;
            if (mk) return true;
        }
        return false;
    };
    {   // the work items' constants for the kernel that makes its own strand records
        std::vector<uint32_t> meta(8 * (nReads + 1), 0xabababab);
        p.itemMeta = meta.data();
        for (uint32_t r = 0; r < nReads + 7; r++) plan_fill_body(p, r);
        for (uint32_t i = 0; i < st.nItems; i++) {
            const uint32_t rd = hp.items[i >> 1];
            const uint32_t *m = meta.data() + 4 * (size_t)i;
            if (m[0] != w.woff[rd] || (m[1] & ~kItemHasN) != w.rlen[rd] || ((m[1] & kItemHasN) != 0) != readHasN(rd) || m[3] != rd || m[2] != (uint32_t)(hp.hitBase[rd] + ((i & 1) ? hp.hitCap[rd] : 0u))) return 16;
        }
        p.itemMeta = nullptr;
        for (uint64_t r = 0; r < nReads; r++) if (hp.pass[r]) slotOf[r] = hp.slotOf[r];     // (plan_fill_body turned the slots of skipped reads into kNone32 once already)
    }
    // a pool one slot too small is flagged and nothing is searched; a launch specialised for shorter reads is
    // flagged and the reads it cannot take are kept out of the work list
    if (hp.hitsTotal > 0) {
        auto replan = [&](BatchStatus &sx) {
            p.st = &sx;
            for (uint32_t r = 0; r < nReads + 7; r++) plan_body(p, r);
            a = 0; b = 0;
            for (uint64_t r = 0; r <= nReads; r++) { slotOf[r] = a; hitBase[r] = b; if (r < nReads) { a += hitCap[r] != 0; b += 2ull * hitCap[r]; } }
            for (uint32_t r = 0; r < nReads + 7; r++) plan_fill_body(p, r);
        };
        BatchStatus s2{}, s3{};
        p.hitsCap = hp.hitsTotal - 1;
        replan(s2);
        if (s2.nItems != 0 || s2.flags != kStHitsOverflow || s2.hitsNeed != hp.hitsTotal) return 9;
        p.hitsCap = hp.hitsTotal; p.maxLenAllowed = (uint32_t)hp.maxLen - 1;
        replan(s3);
        if (!(s3.flags & kStLenOverflow)) return 10;
        uint32_t shorter = 0;
        for (uint32_t r : hp.items) shorter += (off[r + 1] - off[r]) < hp.maxLen;
        if (s3.nItems != 2 * shorter) return 13;
        // fewer packed words uploaded than the lengths promise: flagged, and the reads past the end stay out of the work list
        if (w.woff[nReads] > 0) {
            BatchStatus s4{};
            p.maxLenAllowed = 0xffffffffu; p.nWords = w.woff[nReads] - 1;
            replan(s4);
            if (!(s4.flags & kStWordsOverflow)) return 14;
            uint32_t inside = 0;
            for (uint32_t r : hp.items) inside += w.woff[r + 1] <= p.nWords;
            if (s4.nItems != 2 * inside) return 15;
            p.nWords = w.woff[nReads];
        }
    }
    const uint64_t nQ = paired ? nReads / 2 : nReads;
    std::vector<uint32_t> ms(nQ + 1, 5);
    for (uint32_t q = 0; q < nQ + 3; q++) plan_maxscore_body(w.rlen.data(), hp.pass.data(), (uint32_t)nQ, paired, ms.data(), q);
    auto perfect = [&](uint64_t r) { const uint64_t L = off[r + 1] - off[r]; return L > 15 ? (uint32_t)((L - 15) * (L - 15)) : 0u; };
    for (uint64_t q = 0; q < nQ; q++) {
        const uint64_t r0 = paired ? 2 * q : q;
        const bool p0 = hp.pass[r0] != 0, p1 = paired ? hp.pass[r0 + 1] != 0 : false;
        const uint32_t want = (paired && p0 && p1) ? perfect(r0) + perfect(r0 + 1) : p0 ? perfect(r0) : p1 ? perfect(r0 + 1) : 0u;
        if (ms[q] != want) return 8;
    }
    // strand records: char j of a record = j-th base from the right end of the searched strand
    for (uint32_t W : {4u, 6u, 8u}) {
        if (hp.maxLen > 32 * W || hp.items.empty()) continue;
        Work v;
        packReads(seq, off, nReads, v);
        v.st.nItems = (uint32_t)(2 * hp.items.size());
        v.d.bases = v.bases.data(); v.d.nmask = v.nmask.data(); v.d.rlen = v.rlen.data(); v.d.woff = v.woff.data();
        v.d.items = hp.items.data(); v.d.hitBase = hp.hitBase.data(); v.d.hitCap = hp.hitCap.data(); v.d.st = &v.st;
        std::vector<uint8_t> recs((size_t)v.st.nItems * rec_bytes((int)W), 0xa5);
        for (uint32_t t = 0; t < (v.st.nItems + 2) * W; t++) pack_body(v.d, recs.data(), W, t);
        for (uint32_t item = 0; item < v.st.nItems; item++) {
            const uint32_t rd = hp.items[item >> 1];
            const bool fw = (item & 1) == 0;
            const uint64_t L = off[rd + 1] - off[rd];
            const uint8_t *rec = recs.data() + (size_t)item * rec_bytes((int)W);
            const uint64_t *lw = reinterpret_cast<const uint64_t *>(rec);
            const uint32_t *lm = reinterpret_cast<const uint32_t *>(rec + 8 * W);
            const uint32_t *meta = reinterpret_cast<const uint32_t *>(rec + rec_bytes((int)W) - 16);
            if (meta[0] != L || meta[2] != rd || meta[1] != (uint32_t)(hp.hitBase[rd] + (fw ? 0u : hp.hitCap[rd]))) return 11;
            for (uint32_t j = 0; j < 32 * W; j++) {
                uint32_t wantC = 0, wantN = 0;
                if (j < L) {
                    const uint8_t c = fw ? seq[off[rd] + (L - 1 - j)] : seq[off[rd] + j];
                    wantN = c > 3; wantC = c > 3 ? 0u : (fw ? c : (uint32_t)(c ^ 3));
                }
                if (((lw[j >> 5] >> (2 * (j & 31))) & 3) != wantC || ((lm[j >> 5] >> (j & 31)) & 1u) != wantN) return 12;
            }
        }
    }
    return 0;
}

// compact_body against the obvious loop; returns 0 when the packed rows agree
int emu_compact_check(const cf_row *rows, const uint32_t *nRows, uint32_t k, uint32_t nQ) {
    std::vector<uint64_t> first(nQ + 1, 0);
    for (uint32_t q = 0; q < nQ; q++) first[q + 1] = first[q] + nRows[q];
    std::vector<OutRow> dst(first[nQ] + 1);
    // as the score kernels leave them: the row of a query that prints one by field (its k slots poisoned), several in the slots
    std::vector<OutRow> slots(reinterpret_cast<const OutRow *>(rows), reinterpret_cast<const OutRow *>(rows) + (size_t)nQ * k);
    const uint64_t stride = nQ + 1;
    std::vector<uint64_t> o1tax(stride * kFieldRows, 0), o1a(stride * kFieldRows, 0), o1b(stride * kFieldRows, 0);
    for (uint32_t q = 0; q < nQ; q++) if (nRows[q] <= kFieldRows) for (uint32_t i = 0; i < nRows[q]; i++) {
        const OutRow o = slots[(size_t)q * k + i];
        const size_t at = (size_t)i * stride + q;
        o1tax[at] = o.taxID; o1a[at] = (uint64_t)o.uniqueID | ((uint64_t)o.score << 32); o1b[at] = (uint64_t)o.hitLen | ((uint64_t)o.tidx << 32);
        std::memset(&slots[(size_t)q * k + i], 0xee, sizeof(OutRow));
    }
    const DCompact c{slots.data(), o1tax.data(), o1a.data(), o1b.data(), stride, nRows, first.data(), k, nQ, dst.data(), nullptr};
    for (uint32_t q = 0; q < nQ + 5; q++) compact_body(c, q);
    uint64_t w = 0;
    for (uint32_t q = 0; q < nQ; q++)
        for (uint32_t i = 0; i < nRows[q]; i++, w++)
            if (std::memcmp(&dst[w], &rows[(uint64_t)q * k + i], sizeof(OutRow)) != 0) return 1;
    return 0;
}

// the inverse BWT in the order cf_index_restore launches it: pass 1, link fix-up, pointer-doubling
// rounds, pass 2 (one-lane chains).  Returns 0, or 2 when the walks do not add up to the text.
int emu_restore(void *p, uint32_t shift, uint8_t *packed, uint64_t nBytes) {
    EmuIndex &ix = *static_cast<EmuIndex *>(p);
    const uint64_t n = ix.h.g.len;
    if (nBytes < n / 4 + 1) return 1;
    DRestore r{};
    r.n = n; r.shift = shift;
    r.nMarked = (uint32_t)(n >> shift) + 1;
    r.nSeg = r.nMarked + ((n & ((1ull << shift) - 1)) ? 1u : 0u);
    r.maxSteps = n + 1;
    const uint32_t nElem = r.nSeg + 1;
    std::vector<uint64_t> sumA(nElem, 0), sumB(nElem, 0);
    std::vector<uint32_t> nextA(nElem, 0), nextB(nElem, 0), text((n + 15) / 16 + 1, 0);
    uint32_t cursor = 0, err = 0;
    r.cursor = &cursor; r.segLen = sumA.data(); r.segNext = nextA.data(); r.err = &err; r.text = text.data();
    g_emu.tid = 0; g_emu.nthreads = 1;
    restore_body<1, false>(ix.d, r);
    for (uint32_t s = 0; s < r.nSeg; s++) if (nextA[s] == kRestoreTerm) nextA[s] = r.nSeg;
    sumA[r.nSeg] = 0; nextA[r.nSeg] = r.nSeg;
    uint64_t *si = sumA.data(), *so = sumB.data(); uint32_t *ni = nextA.data(), *no = nextB.data();
    for (uint64_t span = 1; span < nElem; span <<= 1) {
        for (uint32_t s = 0; s < nElem; s++) restore_rank_body(si, ni, so, no, nElem, s);
        std::swap(si, so); std::swap(ni, no);
    }
    if (err || si[r.nSeg - 1] != n) return 2;
    r.segEnd = si;
    cursor = 0;
    restore_body<1, true>(ix.d, r);
    if (err) return 2;
    std::memcpy(packed, text.data(), n / 4 + 1);
    return 0;
}

// the text-verification tables as the device layer makes them at load time: pass 1, ranking, pass 2 of the inverse BWT with
// the sampled SA / ISA outputs switched on (restore_body).  rate < 0: off
int emu_textify(void *p, int rate) {
    EmuIndex &ix = *static_cast<EmuIndex *>(p);
    ix.d.text = nullptr; ix.d.saPos = nullptr; ix.d.isa = nullptr; ix.d.posRate = -1;
    ix.d.posFrag = nullptr;                              // (emu_posify rests on the inverse sample: made again after this)
    if (rate < 0) return 0;
    const uint64_t n = ix.h.g.len;
    const uint32_t shift = 4;
    DRestore r{};
    r.n = n; r.shift = shift;
    r.nMarked = (uint32_t)(n >> shift) + 1;
    r.nSeg = r.nMarked + ((n & ((1ull << shift) - 1)) ? 1u : 0u);
    r.maxSteps = n + 1;
    const uint32_t nElem = r.nSeg + 1;
    std::vector<uint64_t> sumA(nElem, 0), sumB(nElem, 0);
    std::vector<uint32_t> nextA(nElem, 0), nextB(nElem, 0);
    ix.text.assign((n + 31) / 32 + 8, 0);
    // (the inverse sample at its own rate, g_isaExtra steps coarser — DIndex::isaRate; small ranges against the text keep it at the SA sample's)
    const int isaRate = (g_multiRows >= 2 && rate == 0) ? rate : std::min(6, rate + (int)g_isaExtra);
    ix.saPos.assign(trio_words((n >> rate) + 2), 0); ix.isa.assign(trio_words((n >> isaRate) + 2), 0);
    uint32_t cursor = 0, err = 0;
    r.cursor = &cursor; r.segLen = sumA.data(); r.segNext = nextA.data(); r.err = &err; r.text = reinterpret_cast<uint32_t *>(ix.text.data());
    g_emu.tid = 0; g_emu.nthreads = 1;
    restore_body<1, false>(ix.d, r);
    ix.restoreShift = shift; ix.restoreMaxSeg = 0;               // (k_restore_link: the longest segment bounds every walk-left)
    for (uint32_t s2 = 0; s2 < r.nSeg; s2++) if (sumA[s2] > ix.restoreMaxSeg) ix.restoreMaxSeg = (uint32_t)sumA[s2];
    for (uint32_t s2 = 0; s2 < r.nSeg; s2++) if (nextA[s2] == kRestoreTerm) nextA[s2] = r.nSeg;
    sumA[r.nSeg] = 0; nextA[r.nSeg] = r.nSeg;
    uint64_t *si = sumA.data(), *so = sumB.data(); uint32_t *ni = nextA.data(), *no = nextB.data();
    for (uint64_t span = 1; span < nElem; span <<= 1) {
        for (uint32_t s2 = 0; s2 < nElem; s2++) restore_rank_body(si, ni, so, no, nElem, s2);
        std::swap(si, so); std::swap(ni, no);
    }
    if (err || si[r.nSeg - 1] != n) return -2;
    r.segEnd = si;
    r.saPos = ix.saPos.data(); r.isa = ix.isa.data(); r.posShift = (uint32_t)rate; r.isaShift = (uint32_t)isaRate;
    cursor = 0;
    restore_body<1, true>(ix.d, r);
    if (err) return -2;
    // the two samples are each other's inverse where both are defined, and SA is a permutation of the positions
    uint64_t sum = 0;
    for (uint64_t i = 0; i <= (n >> rate); i++) {
        const uint64_t pos = trio_at(ix.saPos.data(), i);
        if (pos > n) return -3;
        sum += pos;
        if ((pos & ((1ull << isaRate) - 1)) == 0 && trio_at(ix.isa.data(), pos >> isaRate) != (i << rate)) return -4;
    }
    for (uint64_t i = 0; i <= (n >> isaRate); i++) {
        const uint64_t row = trio_at(ix.isa.data(), i);
        if (row > n) return -3;
        if ((row & ((1ull << rate) - 1)) == 0 && trio_at(ix.saPos.data(), row >> rate) != (i << isaRate)) return -4;
    }
    if (rate == 0 && sum != n * (n + 1) / 2) return -3;
    ix.d.text = ix.text.data(); ix.d.saPos = ix.saPos.data(); ix.d.isa = ix.isa.data(); ix.d.posRate = rate; ix.d.isaRate = isaRate;
    ix.d.verifyMinRun = g_verifyMinRun;
    ix.d.multiRows = rate == 0 ? g_multiRows : 0u; ix.d.multiMinRun = g_multiMinRun;
    return 1;
}

// dense_unpack_body (cf_batch_upload_dense_async's kernel): the dense form into 32-base words + lengths; `dense` padded by 16 bytes.
// narrow = compact_body's narrow rows of ONE query given by field (what the score kernels leave): returns the qinfo byte
void emu_dense_unpack(const uint8_t *dense, uint32_t nReads, uint32_t readLen, uint64_t *bases, uint32_t *rlen) {
    const DUnpack u{dense, bases, rlen, nReads, readLen, nullptr};
    const uint64_t W = (readLen + 31) / 32;
    for (uint64_t t = 0; t < (uint64_t)nReads * (W ? W : 1) + 5; t++) dense_unpack_body(u, t);
}
// ... with the forward strands' words in search order beside them (DUnpack::rev), checked against rev_word over the unpacked
// words — the two ways the device layer makes them (k_dense_unpack, k_rev_words): 0 = they agree, else 1 + the first word that does not
uint64_t emu_dense_unpack_rev(const uint8_t *dense, uint32_t nReads, uint32_t readLen, uint64_t *bases, uint32_t *rlen, uint64_t *rev) {
    const DUnpack u{dense, bases, rlen, nReads, readLen, rev};
    const uint64_t W = (readLen + 31) / 32;
    for (uint64_t t = 0; t < (uint64_t)nReads * (W ? W : 1) + 5; t++) dense_unpack_body(u, t);
    for (uint64_t r = 0; r < nReads; r++)
        for (uint32_t k = 0; k < W; k++)
            if (rev[r * W + k] != rev_word(bases + r * W, readLen, k)) return 1 + r * W + k;
    return 0;
}

// centrifuge-inspect's FASTA mode over the emulated restore, written to `path`
int emu_inspect_fasta(void *p, uint32_t shift, int across, const char *path) {
    EmuIndex &ix = *static_cast<EmuIndex *>(p);
    std::vector<uint8_t> packed(ix.h.g.len / 4 + 1);
    const int rc = emu_restore(p, shift, packed.data(), packed.size());
    if (rc) return rc;
    std::FILE *f = std::fopen(path, "wb");
    if (!f) return 3;
    try { printSequences(ix.h, packed.data(), across, f); } catch (const std::exception &e) { std::fprintf(stderr, "emu_inspect_fasta: %s\n", e.what()); std::fclose(f); return 4; }
    std::fclose(f);
    return 0;
}

// ---- the text forms (cf_textio.hpp): the device layer's launches of cf_batch_upload_text / the plan stage / cf_batch_wait_text,
//      one body call per thread; in the 64-lane build the bodies with cross-lane sums run as wavefronts of 64 fibers
}  // extern "C"
template <typename F>
static void emuThreads(uint64_t n, F body) {
#ifdef CF_EMU_WAVE64
    for (uint64_t base = 0; base < n; base += CF_WAVE) g_waveCollectives += emu_run_wave([&, base] { body((uint32_t)(base + (uint64_t)emu_wave_lane())); });
#else
    for (uint64_t t = 0; t < n; t++) body((uint32_t)t);
#endif
}
extern "C" {
// text: the block followed by >= kTextPad zero bytes, 8-byte aligned.  status = {nWords, nBases, maxLen, flags}.  Returns the number of
// records (0 with flags set: the block is not in the plain form)
uint32_t emu_text_parse2(const uint8_t *text, uint64_t nBytes, int format, uint32_t globalSeed, uint64_t posCap, uint32_t recCap,
                         uint32_t *rlen, uint32_t *seeds, uint32_t *seqOff, uint32_t *idOff, uint32_t *idLen, uint64_t *status,
                         uint32_t textBase, uint32_t stride, uint32_t mate);
uint32_t emu_text_parse(const uint8_t *text, uint64_t nBytes, int format, uint32_t globalSeed, uint64_t posCap, uint32_t recCap,
                        uint32_t *rlen, uint32_t *seeds, uint32_t *seqOff, uint32_t *idOff, uint32_t *idLen, uint64_t *status) {
    return emu_text_parse2(text, nBytes, format, globalSeed, posCap, recCap, rlen, seeds, seqOff, idOff, idLen, status, 0, 1, 0);
}
// ... one block of a pair of blocks: `text` points at the block (textBase bytes into the buffer the later passes see), its record r
// is read stride * r + mate of the batch; the status words are added to
uint32_t emu_text_parse2(const uint8_t *text, uint64_t nBytes, int format, uint32_t globalSeed, uint64_t posCap, uint32_t recCap,
                         uint32_t *rlen, uint32_t *seeds, uint32_t *seqOff, uint32_t *idOff, uint32_t *idLen, uint64_t *status,
                         uint32_t textBase, uint32_t stride, uint32_t mate) {
    const uint64_t nPieces = (nBytes + kTextPiece - 1) / kTextPiece;
    std::vector<uint32_t> cnt(nPieces + 1, 0), pos(posCap + 1, 0);
    std::vector<uint64_t> base(nPieces + 1, 0);
    const DTextMark m{text, nBytes, format == (int)kTextFasta ? (uint32_t)'>' : (uint32_t)'\n', cnt.data(), base.data(), pos.data(), posCap};
    for (uint64_t t = 0; t < nPieces + 3; t++) text_count_body(m, t);
    for (uint64_t t = 0; t < nPieces; t++) base[t + 1] = base[t] + cnt[t];
    for (uint64_t t = 0; t < nPieces + 3; t++) text_mark_body(m, t);
    TextStatus st{};
    const uint32_t seed0 = (globalSeed + 101u) * 59u * 61u * 67u * 71u * 73u * 79u * 83u;
    const DTextRec d{text, nBytes, pos.data(), &base[nPieces], posCap, recCap, (uint32_t)format, seed0, rlen, seeds, seqOff, idOff, idLen, &st, textBase, stride, mate};
    emuThreads((uint64_t)recCap + 70, [&](uint32_t r) { text_record_body(d, r); });
    status[0] += st.words(); status[1] += st.bases(); status[2] = std::max<uint64_t>(status[2], st.maxLen); status[3] |= st.flags;
    if (st.flags) return 0;
    return (uint32_t)(format == (int)kTextFasta ? base[nPieces] : base[nPieces] >> 2);
}
void emu_text_pack(const uint8_t *text, uint32_t nReads, const uint32_t *seqOff, const uint32_t *rlen, uint64_t *bases, uint32_t *nmask) {
    std::vector<uint64_t> woff(nReads + 1, 0);
    for (uint32_t r = 0; r < nReads; r++) woff[r + 1] = woff[r] + ((rlen[r] + 31) >> 5);
    const DTextPack d{text, seqOff, rlen, woff.data(), bases, nmask, nReads};
    for (uint32_t r = 0; r < nReads + 5; r++) text_pack_body(d, r);
}
// the default columns of nQueries queries from narrow rows; returns the bytes of text (out holds outCap), tupleWords the filled
// words of `tuples`; single: per-taxon counters that are added to
uint64_t emu_text_format(const uint8_t *text, const uint32_t *idOff, const uint32_t *idLen, const uint32_t *rlen, const cf_row16 *rows,
                         const uint8_t *qinfo, const uint32_t *score2, const uint32_t *maxScore, uint32_t nQueries, int paired,
                         const uint8_t *strs, const uint32_t *uidOff, const uint32_t *rankOff, const uint32_t *taxOff, const uint8_t *taxLeaf,
                         uint32_t nRefs, uint32_t nTaxa, uint32_t idxZero, uint8_t *out, uint64_t outCap, unsigned long long *single,
                         uint32_t *tuples, uint32_t tuplesCap, uint32_t *tupleWords) {
    std::vector<uint64_t> rowFirst(nQueries + 1, 0), outOff(nQueries + 1, 0);
    for (uint32_t q = 0; q < nQueries; q++) rowFirst[q + 1] = rowFirst[q] + (qinfo[q] & 0x3fu);
    std::vector<uint32_t> size(nQueries + 1, 0);
    TextStatus st{};
    DTextFmt f{};
    f.text = text; f.idOff = idOff; f.idLen = idLen; f.rlen = rlen;
    static_assert(sizeof(TextRow) == sizeof(cf_row16), "TextRow layout");
    f.rows = reinterpret_cast<const TextRow *>(rows); f.rowFirst = rowFirst.data(); f.qinfo = qinfo; f.score2 = score2; f.maxScore = maxScore;
    f.nQueries = nQueries; f.paired = paired ? 1u : 0u;
    f.strs = strs; f.uidOff = uidOff; f.rankOff = rankOff; f.taxOff = taxOff; f.taxLeaf = taxLeaf; f.nRefs = nRefs; f.nTaxa = nTaxa; f.idxZero = idxZero;
    f.size = size.data(); f.outOff = outOff.data(); f.out = out; f.outCap = outCap; f.single = single; f.tuples = tuples; f.tuplesCap = tuplesCap; f.st = &st;
    for (uint32_t q = 0; q < nQueries + 5; q++) fmt_size_body(f, q);
    for (uint32_t q = 0; q < nQueries; q++) outOff[q + 1] = outOff[q] + size[q];
    // (a wavefront's LDS: one buffer per wavefront of the 64-lane build — its fibers run one after the other —, per thread in the one-lane build)
    std::vector<uint64_t> ldsAll(((uint64_t)nQueries + 70 + CF_WAVE) / CF_WAVE * ((kFmtLds + 16) / 8 + 1));
    emuThreads((uint64_t)nQueries + 70, [&](uint32_t q) { fmt_write_body(f, q, reinterpret_cast<uint8_t *>(ldsAll.data() + (uint64_t)(q / CF_WAVE) * ((kFmtLds + 16) / 8 + 1))); });
    *tupleWords = st.tupleWords;
    return nQueries ? st.outBytes : 0;
}

}  // extern "C"
