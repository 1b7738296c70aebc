// cf_device.hip — HBM layout, kernel launches and the C ABI (include/centrifuge_amd.h).
// gfx950 only; there is no CPU path behind any compute entry point.
#include <hip/hip_runtime.h>
#include <sched.h>
#include <dlfcn.h>

#include <algorithm>
#include <cmath>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <mutex>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/centrifuge_amd.h"
#include "cf_index.hpp"
#include "cf_kernels.hpp"
#include "cf_scan.hpp"
#include "cf_textio.hpp"
#include "cf_restore.hpp"
#include "cf_plan.hpp"
#include "cf_knobs.hpp"

using namespace cfamd;

namespace {

thread_local std::string g_err;

struct HipError : std::runtime_error {
    using std::runtime_error::runtime_error;
};
struct ArgError : std::runtime_error {
    using std::runtime_error::runtime_error;
};

#define HIP_OK(expr)                                                                              \
    do {                                                                                          \
        hipError_t e_ = (expr);                                                                   \
        if (e_ != hipSuccess)                                                                     \
            throw HipError(std::string(#expr) + ": " + hipGetErrorString(e_));                    \
    } while (0)

// ---- a device allocation that frees itself
// set while cf_slot_estimate_bytes runs a slot's sizing without a device: allocations only add up
static thread_local uint64_t *g_dryBytes = nullptr;
template <typename T>
struct DevBuf {
    T *p = nullptr;
    size_t n = 0;
    DevBuf() = default;
    DevBuf(const DevBuf &) = delete;
    DevBuf &operator=(const DevBuf &) = delete;
    ~DevBuf() { release(); }
    void release() { if (p) (void)hipFree(p); p = nullptr; n = 0; }
    void alloc(size_t count) {
        release();
        if (count == 0) count = 1;
        if (g_dryBytes) { *g_dryBytes += count * sizeof(T); n = count; return; }      // (cf_slot_estimate_bytes: sizes only)
        HIP_OK(hipMalloc(reinterpret_cast<void **>(&p), count * sizeof(T)));
        n = count;
    }
    void ensure(size_t count) { if (count > n) alloc(count + count / 8); }
    void upload(const std::vector<T> &v) {
        alloc(v.size());
        if (!v.empty()) HIP_OK(hipMemcpy(p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice));
    }
    size_t bytes() const { return n * sizeof(T); }
};

// ------------------------------------------------------------------ kernels
template <int G>
__global__ void __launch_bounds__(256) k_search(DIndex ix, DParams pr, DBatch b) { search_body<G>(ix, pr, b); }

__global__ void __launch_bounds__(256) k_rlen(const uint64_t *off, uint32_t *rlen, uint32_t nReads) { rlen_body(off, rlen, nReads, cf_global_thread()); }
__global__ void __launch_bounds__(256) k_convert(DConvert c) { convert_body(c, cf_global_thread()); }
__global__ void __launch_bounds__(256) k_dense_unpack(DUnpack u) { dense_unpack_body(u, (uint64_t)blockIdx.x * blockDim.x + threadIdx.x); }
__global__ void __launch_bounds__(256) k_pack(DBatch b, uint8_t *recs, uint32_t W) { pack_body(b, recs, W, cf_global_thread()); }

// W = 4 (reads <= 128 bases): 62 VGPRs and 20 KB of LDS per block = 8 waves/SIMD, with the natural register
// allocation (no launch-bounds pressure: forcing it spilled and ran 1.6x slower, profiles/r01_sweeps.txt).
// W = 6 (<= 192 bases, e.g. 150 bp mates): 96-byte records, 24 KB of LDS = 6 blocks per CU.
// W = 8 (<= 256 bases): 64 VGPRs, 28 KB of LDS = 5 blocks per CU.
// The occupancy asked of the compiler is what the LDS allows anyway: 25.6 / 29.7 / 33.8 KB per block = 6 / 5 / 4 blocks per CU, i.e.
// up to 85 / 102 / 128 VGPRs (round 4 asked for 7 waves throughout, which the body — 76 VGPRs — cannot meet: six warnings per build);
// the instrumented build (COUNT) takes what it needs
template <int G, int W, bool COUNT>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(COUNT ? 1 : W == 4 ? 6 : W == 6 ? 5 : 4, 8))) k_search2(DIndex ix, DParams pr, DBatch b) {
    // strand records of the block's chains, then one rank table per lane
    __shared__ __attribute__((aligned(16))) uint8_t lds[(256 / G) * rec_lds_stride(W) + 256 * 4 * RankTab<G>::WORDS + (256 / G) * 16 * kLazyHits];
    search2_body<G, W, COUNT>(ix, pr, b, lds);
}

// one chain per lane over the occurrence planes (DIndex::planes): 64 chains per wave, LDS = the strand records only
template <int W, bool COUNT, int LZ = 0, bool MULTI = false>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu((W == 4 && LZ == 1) ? 7 : (W == 4 && MULTI) ? 6 : (W == 6 && MULTI) ? 5 : 1, 8))) k_search2_l1(DIndex ix, DParams pr, DBatch b) {
    __shared__ __attribute__((aligned(16))) uint8_t lds[256 * rec_lds_stride(W) + 256 * 16 * (LZ ? LZ : (int)lazy_hits(1, W))];
    search2_body<1, W, COUNT, true, LZ, MULTI>(ix, pr, b, lds);
}
__global__ void __launch_bounds__(256) k_pair_planes(DIndex ix, uint8_t *planes2, uint64_t nGroups) {
    pair_planes_body(ix, planes2, (uint64_t)blockIdx.x * blockDim.x + threadIdx.x, nGroups);
}
__global__ void __launch_bounds__(256) k_occ_planes(DIndex ix, uint8_t *planes, uint64_t nSides) {
    occ_planes_body(ix, planes, (uint64_t)blockIdx.x * blockDim.x + threadIdx.x, nSides);
}
// repeat_probe_body over pseudo-random rows: hits[0] = samples whose two neighbouring rows share their preceding `depth` bases
__global__ void __launch_bounds__(256) k_repeat_probe(DIndex ix, uint32_t nSamples, uint32_t depth, uint32_t *hits) {
    const uint32_t i = cf_global_thread();
    bool yes = false;
    if (i < nSamples && ix.len > 2) {
        uint64_t x = 0x9e3779b97f4a7c15ull * (i + 1);
        x ^= x >> 29; x *= 0xbf58476d1ce4e5b9ull; x ^= x >> 32;
        yes = repeat_probe_body(ix, x % (ix.len - 1), depth);
    }
    const uint64_t m = cf_ballot(yes);
    if (m && cf_lane() == (uint32_t)cf_ctz64(m)) (void)cf_atomic_add(hits, (uint32_t)cf_popc64(m));
}

// the common-case post / score kernels: one lane per query, everything in registers; the queries they cannot take go to a list
__global__ void __launch_bounds__(256) k_post_fast(DIndex ix, DParams pr, DBatch b) {
    const uint32_t q = cf_global_thread();
    const bool d = q < b.nQueries ? post_fast_body(ix, pr, b, q) : false;
    if (q < b.nQueries && b.postDeferred) b.postDeferred[q] = d ? 1 : 0;
    defer_push(b.slowPost, &b.st->nSlowPost, d, q);
}
template <bool EARLY>
__global__ void __launch_bounds__(256) k_score_fast(DIndex ix, DParams pr, DBatch b) {
    const uint32_t q = cf_global_thread();
    const bool d = q < b.nQueries ? score_fast_body<EARLY>(ix, pr, b, q) : false;
    defer_push(b.slowScore, &b.st->nSlowScore, d, q);
}
// (CF_POST_FAST=0 / CF_SCORE_FAST=0: every query goes to the general kernel)
__global__ void __launch_bounds__(256) k_list_all(uint32_t *list, uint32_t *counter, uint32_t n) {
    const uint32_t q = cf_global_thread();
    if (q < n) list[q] = q;
    if (q == 0) *counter = n;
}
// the general kernels, over the listed queries (their number is on the device)
// (dynamic LDS: capHits hit records of scratch per lane, post_body)
// (three wavefronts per SIMD: the body's natural allocation is a few registers above the 168 that allows, and the kernel lives on
// the number of queries it has in flight)
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(3, 8))) k_post(DIndex ix, DParams pr, DBatch b, uint32_t capHits) {
    extern __shared__ __attribute__((aligned(16))) uint8_t ldsPost[];
    HitP *scratch = capHits ? reinterpret_cast<HitP *>(ldsPost) + (size_t)cf_local_thread() * capHits : nullptr;
    const uint32_t n = b.st->nSlowPost;
    for (uint32_t i = cf_global_thread(); i < n; i += cf_global_threads()) post_body(ix, pr, b, b.slowPost[i], scratch, capHits);
}
__global__ void __launch_bounds__(64) k_postfix_only(DIndex ix, DParams pr, DBatch b) {   // debug tap
    const uint32_t i = cf_global_thread();
    if (i < b.st->nItems / 2) post_fix(ix, pr, b, b.items[i]);
}
__global__ void k_window(DBatch b, uint32_t qLo, bool keepSlow) { if (cf_global_thread() == 0) row_window_body(b, qLo, keepSlow); }
__global__ void __launch_bounds__(256) k_emit(DParams pr, DBatch b) {
    const uint32_t q = cf_global_thread();
    if (q < b.nQueries) emit_body(pr, b, q);
}
template <int G, bool COUNT>
__global__ void __launch_bounds__(256) k_walk2(DIndex ix, DBatch b) { walk2_body<G, COUNT>(ix, b); }
template <int MODE>
__global__ void __launch_bounds__(256) k_walk_table(DIndex ix, DBatch b) { walk2_body<2, false, MODE>(ix, b); }
// The dense resolve table from the SUFFIX ARRAY instead of walks (round 6; densifyIndex, where the text tables hold SA[row] for every
// row).  The walk-left from the row of text position p passes the rows of p - 1, p - 2, ... and answers at the first STOP row — the
// '$' row, a row of the file's sample, a boundary row, in tryOffset's order (bt2_idx.h:1980-2014) — so its answer is that of the
// nearest stop at or left of p IN THE TEXT: mark the stops by their positions (stopVal[pos], one bit per position), then
// table[row] = stopVal[the last marked position <= SA[row]].  One streamed pass over the rows with ~2 random reads each instead of a walk
// of 15 dependent steps per row; the longest walk (the bound the position form of hits rests on) is the widest gap seen.
// (grid-stride loops: a launch of more than 2^32 threads is refused, and an index has more rows than that)
template <typename T>
__global__ void __launch_bounds__(256) k_stop_samples(DIndex ix, T *stopVal, uint64_t *stopBits, uint64_t nSamples) {
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nSamples; i += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t pos = trio_at(ix.saPos, i << ix.offRate);
        stopVal[pos] = static_cast<const T *>(ix.offs)[i];
        cf_atomic_or64(&stopBits[pos >> 6], 1ull << (pos & 63));
    }
}
template <typename T>
__global__ void __launch_bounds__(256) k_stop_bounds(DIndex ix, T *stopVal, uint64_t *stopBits) {
    for (uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; j < ix.nBound; j += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t pos = trio_at(ix.saPos, ix.boundRow[j]);
        stopVal[pos] = (T)ix.boundRef[j];
        cf_atomic_or64(&stopBits[pos >> 6], 1ull << (pos & 63));
    }
}
template <typename T>
__global__ void k_stop_end(DIndex ix, T *stopVal, uint64_t *stopBits) {       // the '$' row: the suffix at position 0, reference 0
    const uint64_t pos = trio_at(ix.saPos, ix.zOff);
    stopVal[pos] = 0;
    cf_atomic_or64(&stopBits[pos >> 6], 1ull << (pos & 63));
}
template <typename T>
__global__ void __launch_bounds__(256) k_table_by_position(DIndex ix, const T *stopVal, const uint64_t *stopBits, T *table, uint64_t nRows, uint32_t *walkMax) {
    uint32_t gap = 0;
    for (uint64_t row = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; row < nRows; row += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t pos = trio_at(ix.saPos, row);
        uint64_t w = pos >> 6;
        uint64_t m = stopBits[w] & (~0ull >> (63 - (pos & 63)));
        while (m == 0 && w > 0) m = stopBits[--w];             // (position 0 is a stop whenever the '$' row has one: the loop ends there at the latest)
        const uint64_t pred = m ? (w << 6) + 63 - (uint64_t)__builtin_clzll(m) : 0;
        table[row] = stopVal[pred];
        const uint32_t g = (uint32_t)(pos - pred);
        gap = g > gap ? g : gap;
    }
    for (int d = 32; d > 0; d >>= 1) { const uint32_t o = __shfl_xor(gap, d, 64); gap = o > gap ? o : gap; }
    if ((threadIdx.x & 63) == 0 && gap) atomicMax(walkMax, gap);
}
// ... and a spot check of the finished table against the walk itself (the file's sample is still what DIndex::walkOffs names):
// 64 K rows spread over the table; a difference sends densifyIndex back to the walks
template <typename T>
__global__ void __launch_bounds__(256) k_table_spot_check(DIndex ix, const T *table, uint64_t nRows, uint32_t nCheck, uint32_t *bad) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nCheck) return;
    const uint64_t row = i < 64 ? (i < 32 ? (uint64_t)i : nRows - 1 - (i - 32)) % nRows : (0x9e3779b97f4a7c15ull * (i + 1)) % nRows;
    uint32_t steps = 0;
    const uint32_t want = resolve_plain_row(ix, row, steps);
    if ((uint32_t)table[row] != (sizeof(T) == 2 ? (want & 0xffffu) : want)) atomicAdd(bad, 1u);
}

template <bool COUNT>
__global__ void __launch_bounds__(256) k_walk3(DIndex ix, DBatch b) {
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < b.st->rowHi - b.st->rowLo; i += (uint64_t)gridDim.x * blockDim.x)
        walk3_body<COUNT>(ix, b, i);
}
// PLANES: the build for an index whose planes are made (or not) — the range steps and the rows' own LF steps then carry the code of ONE
// of their two forms (rank_any / lf_own_any choose at run time: both forms in one kernel were 256 registers and 72 bytes of scratch, one
// wavefront per SIMD for a kernel that is chains of dependent loads)
template <bool PLANES>
__global__ void __launch_bounds__(256) k_wide_ftab(DIndex ix, uint32_t wideChars, uint64_t *table) {
    __builtin_assume(PLANES ? ix.planes != nullptr : ix.planes == nullptr);
    for (uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; t < (1ull << (2 * wideChars)); t += (uint64_t)gridDim.x * blockDim.x)
        wide_ftab_body(ix, wideChars, table, t);
}

// the planned rows of the queries the common-case score kernel left, resolved into rowRef (DBatch::directRefs)
__global__ void __launch_bounds__(64) k_resolve_slow(DIndex ix, DParams pr, DBatch b) {
    const uint32_t n = b.st->nSlowScore;
    for (uint32_t i = cf_global_thread(); i < n; i += cf_global_threads()) resolve_query_body(ix, pr, b, b.slowScore[i]);
}
// (dynamic LDS: score_scratch_bytes(capRows) of scratch per ACTIVE lane, score_body; every `sparse`-th lane of a wavefront works —
// the kernel lives on latency, not on lanes, and the scratch of 64 lanes would leave one wavefront per CU)
__global__ void __launch_bounds__(64) k_score(DIndex ix, DParams pr, DBatch b, uint32_t capRows, uint32_t sparse, uint32_t rowsMin, uint32_t rowsMax) {
    extern __shared__ __attribute__((aligned(16))) uint8_t ldsScore[];
    const uint32_t n = b.st->nSlowScore;
    if (cf_local_thread() % sparse) return;
    const uint32_t me = cf_global_thread() / sparse, all = cf_global_threads() / sparse;
    uint8_t *scratch = capRows ? ldsScore + (size_t)(cf_local_thread() / sparse) * score_scratch_bytes(capRows) : nullptr;
    for (uint32_t i = me; i < n; i += all) score_body(ix, pr, b, b.slowScore[i], scratch, capRows, rowsMin, rowsMax);
}

// the words of the sparse N mask into the (otherwise zero) dense one; mask == nullptr: those words back to zero
__global__ void __launch_bounds__(256) k_scatter_nmask(const uint64_t *idx, const uint32_t *mask, uint64_t n, uint64_t nWords, uint32_t *nmask) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n && idx[i] < nWords) nmask[idx[i]] = mask ? mask[i] : 0u;
}
// per-taxon counters of a pass from the row taxa the score kernels left (count_body): block = a chunk of queries.  1024 threads: a
// batch is a few hundred chunks — one block per CU — and a thread's loop is two dependent loads per query (nOut, then the row's
// taxon): with four wavefronts on a CU that latency was the kernel (0.16 ms for 100 MB)
__global__ void __launch_bounds__(1024) k_count(DBatch b, uint32_t slotBits, bool direct) {
    __shared__ uint32_t slots[3 * kCountSlots];
    count_body(b, slots, blockIdx.x, slotBits, direct);
}
__global__ void __launch_bounds__(256) k_plan(DPlan p) { plan_body(p, cf_global_thread()); }
__global__ void __launch_bounds__(256) k_plan_fill(DPlan p) { plan_fill_body(p, cf_global_thread()); }
// the forward strands' words in search order (rev_word), one thread per (read, word): neighbouring lanes read and write neighbouring words
__global__ void __launch_bounds__(256) k_rev_words(DPlan p, uint32_t wordsPerRead) { rev_words_body(p, wordsPerRead, (uint64_t)blockIdx.x * blockDim.x + threadIdx.x); }
__global__ void __launch_bounds__(256) k_plan_maxscore(const uint32_t *rlen, const uint8_t *pass, uint32_t nQueries, int paired, uint32_t *maxScore) {
    plan_maxscore_body(rlen, pass, nQueries, paired, maxScore, cf_global_thread());
}
__global__ void __launch_bounds__(256) k_compact(DCompact c) {
    compact_body(c, cf_global_thread());
}
// the text forms (cf_textio.hpp): ingest and egress of the front end on the device, one thread per piece / record / query
__global__ void __launch_bounds__(256) k_text_count(DTextMark m) { text_count_body(m, (uint64_t)blockIdx.x * blockDim.x + threadIdx.x); }
__global__ void __launch_bounds__(256) k_text_mark(DTextMark m) { text_mark_body(m, (uint64_t)blockIdx.x * blockDim.x + threadIdx.x); }
__global__ void __launch_bounds__(256) k_text_records(DTextRec d) { text_record_body(d, cf_global_thread()); }
__global__ void __launch_bounds__(256) k_text_pack(DTextPack d) { text_pack_body(d, cf_global_thread()); }
__global__ void __launch_bounds__(256) k_fmt_size(DTextFmt f) { fmt_size_body(f, cf_global_thread()); }
__global__ void __launch_bounds__(256) k_fmt_write(DTextFmt f) { fmt_write_body(f, cf_global_thread()); }
template <int G, bool WRITE>
__global__ void __launch_bounds__(256) k_restore(DIndex ix, DRestore r) { restore_body<G, WRITE>(ix, r); }
__global__ void __launch_bounds__(256) k_restore_rank(const uint64_t *sumIn, const uint32_t *nextIn, uint64_t *sumOut, uint32_t *nextOut, uint32_t nElem) {
    restore_rank_body(sumIn, nextIn, sumOut, nextOut, nElem, cf_global_thread());
}
// links that reached the '$' row point at the terminator element (index nSeg: sum 0, next = itself)
// (... and the longest segment: no walk-left from any row is longer, makePosTables)
__global__ void __launch_bounds__(256) k_restore_link(uint64_t *sum, uint32_t *next, uint32_t nSeg, uint32_t *maxLen) {
    const uint32_t s = cf_global_thread();
    if (s < nSeg) {
        if (next[s] == kRestoreTerm) next[s] = nSeg;
        const uint64_t len = sum[s];                              // pass 1 left the segment's length here
        uint32_t l32 = len > 0xffffffffull ? 0xffffffffu : (uint32_t)len;
        for (int m = 32; m > 0; m >>= 1) { const uint32_t o = cf_shfl_xor(l32, m); if (o > l32) l32 = o; }
        if (cf_lane() == 0 && l32) cf_atomic_max(maxLen, l32);
    }
    else if (s == nSeg) { sum[s] = 0; next[s] = nSeg; }
}

template <int G>
__global__ void k_debug_rank(DIndex ix, const uint8_t *chars, const uint64_t *rows, uint64_t n, uint64_t *out) {
    const uint64_t i = (uint64_t)cf_global_thread() / G;
    // keep whole groups converged: every lane of a live group runs the cooperative load
    const uint64_t ii = i < n ? i : n - 1;
    uint64_t t, bb; bool two;
    if (ix.sides) rank_pair<G>(ix, chars[ii] & 3, rows[ii], rows[ii], t, bb, two);
    else rank_any<G>(ix, chars[ii] & 3, rows[ii], rows[ii], t, bb, two);          // (the sides were dropped)
    if (i < n && Grp<G>::sub() == 0) out[i] = t;
}

// roofline denominator: every group chases `steps` dependent pseudo-random sides
__global__ void __launch_bounds__(256) k_random_sides(const uint8_t *sides, uint64_t numSides, uint32_t steps,
                                                        uint64_t seed, unsigned long long *sink) {
    const uint32_t grp = cf_global_thread() >> 3, sub = cf_lane() & 7;
    uint64_t x = seed + 0x9e3779b97f4a7c15ull * (grp + 1);
    unsigned long long acc = 0;
    for (uint32_t s = 0; s < steps; s++) {
        x ^= x >> 12; x ^= x << 25; x ^= x >> 27;
        const uint64_t side = (x * 0x2545f4914f6cdd1dull) % numSides;
        const u64x2 v = cf_load16(sides + side * 128 + 16 * sub);
        // fold the loaded data back into the chain so the next address depends on it
        uint32_t f = (uint32_t)(v.x ^ v.y);
        f += __shfl_xor(f, 1, 64); f += __shfl_xor(f, 2, 64); f += __shfl_xor(f, 4, 64);
        x += f & 1u;
        acc += f;
    }
    if (acc == 0x1234567u) sink[0] = acc;
}

}  // namespace

// ------------------------------------------------------------------ objects
struct TablePlanRecord { int K = 0, textRate = -1, planes = 0, resolveRate = -1, pair = 0, dropSides = 0; bool valid = false; };
struct cf_index {
    HostIndex h;
    int device = -1;             // -1: host-only view
    DIndex d{};
    DevBuf<uint8_t> planes;                     // occurrence planes (DIndex::planes), made at load
    float planesMs = 0;
    DevBuf<uint8_t> planes2;                    // pair planes (DIndex::planes2), made at load from the planes
    float planes2Ms = 0;
    DevBuf<uint64_t> wide;                      // wide ftab (DIndex::wide), made at load
    float wideMs = 0;
    DevBuf<uint32_t> text;                      // 2-bit joined text + sampled SA / inverse SA: text verification (DIndex::text ..)
    DevBuf<uint64_t> saPos, isa;
    float textMs = 0;
    DevBuf<uint8_t> sides, offs, dense;         // dense: the resolve table the walk stops at (every 2^denseRate-th row), made at load
    int denseRate = -1;
    uint32_t walkMaxSeen = 0;                   // longest walk the table build took (exact over ALL rows when denseRate == 0)
    bool denseByPos = false;                    // the resolve table was made from the stops' text positions (k_table_by_position), not by walks
    uint32_t restoreShift = 0, restoreMaxSeg = 0;  // the inverse-BWT walks of the last restoreCore: marks every 2^shift rows, longest segment
    DevBuf<uint32_t> posBucket;                 // position -> reference (DIndex::posFrag; makePosTables)
    DevBuf<u64x2> posFrag, posSeq;
    float denseMs = 0;
    DevBuf<uint64_t> ftab, eftab, boundRow, paths;
    DevBuf<RefInfo> refInfo;
    DevBuf<uint32_t> boundRef, boundBits, pathTidx;
    uint64_t deviceBytes = 0, fileBytes = 0, budgetSeen = 0;
    cf_index_options opt{};                     // cf_index_open_ex (all zero: automatic)
    bool planned = false;                       // the options are the table planner's choice: made as they are while they fit
    bool planDropSides = false, sidesDropped = false;   // the sides leave HBM once the tables that are made from them exist
    uint64_t droppedBytes = 0;                  // file sections that left HBM (sides, SA sample)
    bool wantTextRate0 = false;                 // planner probe: text tables at every row or none (small_range_rows)
    // the share of neighbouring suffix-array rows whose suffixes are preceded by the same 24 bases (k_repeat_probe; -1 = not
    // measured): ~0 for an i.i.d.-like collection, 0.3 - 0.6 where strains of a cluster / shared operons keep search ranges a few
    // rows wide for most of a read — what the planner prices the small ranges against the text with
    double repeatFrac = -1.0;
    uint32_t multiRowsPlan = 0;                 // the planner's word on small ranges against the text (rows; 0 = not in effect)
    TablePlanRecord plan{};                     // what the planner chose (cf_index_describe compares it with what was made)
    int plannedTextRate = -2;                   // the planner's text rate (-2 = none recorded: textifyIndex reads the option field)
    int numCUs = 256;
    // resident blocks per CU of the persistent search kernels on THIS device, by record size (64 / 96 / 128 bytes): asked of
    // the runtime once, when the index is opened (launches may come from several threads, and devices may differ)
    int occSides[3] = {0, 0, 0}, occPlanes[3] = {0, 0, 0};
    int occMulti[3] = {0, 0, 0};                // ... of the small-range variants of the planes kernel
    int occLazy[3] = {0, 0, 0};                 // ... of the other lazy-hit count (CF_LAZY_N)
};

struct cf_classifier {
    cf_index *ix = nullptr;
    cf_params p{};
    std::vector<uint64_t> hostList, exclList;
    DParams d{};
    DevBuf<uint8_t> refExcluded;
    DevBuf<uint64_t> hostSet;
    DevBuf<unsigned long long> counts;           // per taxon: reads, unique reads (count_body), perfect single assignments (fmt_write_body)
    // the strings a formatted row repeats (cf_batch_wait_text), made at the first use
    DevBuf<uint8_t> fmtStrs, fmtLeaf;
    DevBuf<uint32_t> fmtUidOff, fmtRankOff, fmtTaxOff;
    uint32_t fmtIdxZero = 0;
    bool fmtMade = false;
    std::mutex fmtMu;
};

// pinned host memory that frees itself (results a batch hands back, staging of the byte input)
template <typename T>
struct PinBuf {
    T *p = nullptr;
    size_t n = 0;
    PinBuf() = default;
    PinBuf(const PinBuf &) = delete;
    PinBuf &operator=(const PinBuf &) = delete;
    ~PinBuf() { if (p) (void)hipHostFree(p); }
    void ensure(size_t count) {
        if (count <= n) return;
        if (p) (void)hipHostFree(p);
        p = nullptr; n = 0;
        count += count / 8;
        if (g_dryBytes) { n = count; return; }
        HIP_OK(hipHostMalloc(reinterpret_cast<void **>(&p), count * sizeof(T), hipHostMallocDefault));
        n = count;
    }
};

// One batch slot: the device workspace of a batch and the pinned host buffers its results come back in.  A slot is
// made once (cf_batch_alloc, or cf_batch_create for the one-shot form) and reused for any number of batches: every
// buffer only ever grows.  Nothing between "reads in" and "results out" needs the host: see BatchStatus.
struct cf_batch {
    cf_classifier *cl = nullptr;
    // the batch that is loaded
    uint64_t nReads = 0, nQueries = 0, nWords = 0;
    int paired = 0;
    uint32_t maxLenHost = 0;                 // upper bound of the read lengths (chooses the search kernel)
    uint32_t recWords = 0;
    bool selfRecords = false;                // the search kernel builds the strand records (no k_pack)
    bool revMade = false;                    // ... and the dense unpack has made them already (else k_rev_words does)
    uint32_t revDelta = 0;                   // DPlan::revDelta: where, behind the packed reads, the forward strands' search-order words lie (0: not made)
    bool loaded = false, planned = false, running = false, finished = false, downloaded = false;
    bool fromBytes = false;                  // the resident reads came as 1 byte per base (seq / off8) and are packed by the plan stage
    // device
    DevBuf<uint8_t> seq, pass, recs;
    DevBuf<uint64_t> bases, woff, off8, hitBase, qBase, rowVal, rowFirst, tileA;
    DevBuf<uint32_t> nmask, rlen, seeds, items, slotOf, hitCap, nhml, rowRef, nOut, score2, maxScore, qRows, tileC, slowPost, slowScore, qflag, itemMeta;
    DevBuf<HitP> hits;
    DevBuf<PlanHit> qplan;
    DevBuf<QHead> qhead;
    DevBuf<uint64_t> o1tax, o1a, o1b;
    DevBuf<uint8_t> dense;                                 // the dense input (cf_dense_reads) as it came, unpacked into bases / rlen
    uint32_t densePending = 0;                             // read length of a dense upload whose words are still to be made (by the plan stage)
    DevBuf<uint8_t> qinfo;                                 // narrow results: one byte per query
    DevBuf<uint8_t> postDeferred;                          // per query: left to the general post kernel (DBatch::postDeferred)
    hipEvent_t evPostFast = nullptr, evPost = nullptr;     // the early score kernel beside the general post kernel (enqueueClassify)
    int resultFormat = CF_RESULTS_ROWS;
    // the text forms: the uploaded block (it stays: the readIDs are copied out of it), what the record pass leaves per read, the
    // formatted rows
    DevBuf<uint8_t> text, textOut;
    DevBuf<uint32_t> txCnt, txPos, txSeqOff, txIdOff, txIdLen, txSize, txTuples, txTileC;
    DevBuf<uint64_t> txBase, txOutOff, txTileA;
    DevBuf<TextStatus> txSt;
    PinBuf<TextStatus> hTxSt;
    PinBuf<uint64_t> hTxTotal;
    PinBuf<uint8_t> hTextOut;
    PinBuf<uint32_t> hTuples;
    bool fromText = false;                   // the resident reads came as text (the plan stage packs them: k_text_pack)
    bool rowsStay = false;                   // cf_batch_wait_text: the rows are formatted on the device, none cross the link
    bool textDone = false;                   // ... and have been (a second wait hands the same text back: the tally is made once)
    uint64_t textBytes = 0, tupleWords = 0;
    DevBuf<uint64_t> nIdx; DevBuf<uint32_t> nMsk;          // sparse N mask of the batch being uploaded
    const uint32_t *nmaskZeroOf = nullptr;                 // the mask buffer that is all zero but for the nSparsePrev words listed in nIdx
    uint64_t nSparsePrev = 0, nmaskZeroN = 0;
    DevBuf<HmEntry> hm;
    DevBuf<TcEntry> tc;
    DevBuf<OutRow> out, outCompact;
    DevBuf<OpCounts> ops;
    DevBuf<unsigned long long> cursor;
    DevBuf<BatchStatus> st;
    uint64_t hitsCapLimit = 0, rowsCapLimit = 0;        // test knobs (cf_batch_set_limits): 0 = none
    DPlan pl{};
    DBatch d{};
    // host (pinned)
    PinBuf<BatchStatus> hSt;
    PinBuf<OpCounts> hOps;
    PinBuf<OutRow> hRows;
    PinBuf<uint32_t> hNOut, hScore2, hMaxScore;
    PinBuf<uint8_t> hQInfo;
    uint64_t rowsSpec = 0;                   // rows the download brought along before the total was known
    uint64_t rowsOut = 0, rowsTotal = 0;
    double rowsPerQuery = 0;                 // printed rows per query of the slot's last batch (0 = none yet): sizes the next download
    double plannedPerQuery = 0;              // planned SA rows per query of the slot's last batch: sizes the row workspace
    uint32_t passes = 0;                     // passes of the row stage the last batch took
    float planMs = 0;
    float ms[5] = {0, 0, 0, 0, 0};
    OpCounts lastOps{};
    bool opsValid = false;
    hipStream_t stream = nullptr;            // stream of the batch in flight
    // CF_TAIL_STREAM=1: the per-query kernels (post .. compact) of a batch run on the slot's OWN stream, behind an event the
    // search kernel leaves on the caller's, so that the tail of batch i runs beside the search of batch i+1.  Measured
    // (profiles/r03d_*): with the per-query kernels and the search both bound by the rate at which the CUs' L1s take
    // (load x line), running them side by side gains nothing (14.6 ms per step against 13.4 on one stream) — what the second
    // stream had hidden in the previous round's pipeline were memsets, which are gone.  Off by default.
    hipStream_t tail = nullptr;
    hipEvent_t ev[10] = {};                  // 0..4 stage marks of classify, 5/6 plan, 7 done, 8 uploaded, 9 classified
    hipEvent_t evLate = nullptr;             // CF_TAIL_STREAM=2: the common-case score kernel is done, the tail may start
    bool evInit = false;
    ~cf_batch() { if (evInit) { for (auto &e : ev) (void)hipEventDestroy(e); (void)hipEventDestroy(evLate); (void)hipEventDestroy(evPostFast); (void)hipEventDestroy(evPost); } if (tail) (void)hipStreamDestroy(tail); }
};

namespace {

void streamSection(std::FILE *f, uint64_t bytes, uint8_t *dst) {
    // 32 MiB pinned staging, double buffered
    constexpr size_t kChunk = 32u << 20;
    uint8_t *stage[2] = {nullptr, nullptr};
    hipStream_t st;
    HIP_OK(hipStreamCreate(&st));
    hipEvent_t done[2];
    for (int i = 0; i < 2; i++) {
        HIP_OK(hipHostMalloc(reinterpret_cast<void **>(&stage[i]), kChunk, hipHostMallocDefault));
        HIP_OK(hipEventCreate(&done[i]));
    }
    uint64_t pos = 0;
    int cur = 0;
    bool used[2] = {false, false};
    try {
        while (pos < bytes) {
            const size_t n = static_cast<size_t>(std::min<uint64_t>(kChunk, bytes - pos));
            if (used[cur]) HIP_OK(hipEventSynchronize(done[cur]));
            if (std::fread(stage[cur], 1, n, f) != n) throw std::runtime_error("short read while streaming an index section");
            HIP_OK(hipMemcpyAsync(dst + pos, stage[cur], n, hipMemcpyHostToDevice, st));
            HIP_OK(hipEventRecord(done[cur], st));
            used[cur] = true;
            pos += n;
            cur ^= 1;
        }
        HIP_OK(hipStreamSynchronize(st));
    } catch (...) {
        for (int i = 0; i < 2; i++) { (void)hipHostFree(stage[i]); (void)hipEventDestroy(done[i]); }
        (void)hipStreamDestroy(st);
        throw;
    }
    for (int i = 0; i < 2; i++) { (void)hipHostFree(stage[i]); (void)hipEventDestroy(done[i]); }
    (void)hipStreamDestroy(st);
}

void uploadIndex(cf_index &ix, const std::string &base) {
    HostIndex &h = ix.h;
    h.load(base, [&](Section s, std::FILE *f, uint64_t bytes) {
        switch (s) {
            case Section::Sides:
                ix.sides.alloc(bytes + 128);
                streamSection(f, bytes, ix.sides.p);
                break;
            case Section::Ftab:
                ix.ftab.alloc(bytes / 8);
                streamSection(f, bytes, reinterpret_cast<uint8_t *>(ix.ftab.p));
                break;
            case Section::Eftab:
                ix.eftab.alloc(bytes / 8);
                streamSection(f, bytes, reinterpret_cast<uint8_t *>(ix.eftab.p));
                break;
            case Section::SaSample:
                ix.offs.alloc(bytes + 8);
                streamSection(f, bytes, ix.offs.p);
                break;
        }
    });
    const IndexTables t = makeIndexTables(h);
    ix.boundRow.upload(h.boundRow);
    ix.boundRef.upload(h.boundRef);
    ix.boundBits.upload(t.boundBits);
    {
        std::vector<RefInfo> ri(h.uidTid.size() + 1, RefInfo{0, 0, kNone32});
        for (size_t i = 0; i < h.uidTid.size(); i++) ri[i] = RefInfo{h.uidTid[i], t.refTidx[i], t.refPath[i]};
        ix.refInfo.upload(ri);
    }
    ix.paths.upload(t.paths);
    ix.pathTidx.upload(t.pathTidx);

    DIndex &d = ix.d;
    fillIndexScalars(h, t, d);
    d.sides = ix.sides.p; d.ftab = ix.ftab.p; d.eftab = ix.eftab.p;
    d.offs = ix.offs.p; d.walkOffs = ix.offs.p;
    d.boundRow = ix.boundRow.p; d.boundRef = ix.boundRef.p; d.boundBits = ix.boundBits.p;
    d.refInfo = ix.refInfo.p;
    d.paths = ix.paths.p; d.pathTidx = ix.pathTidx.p;
    ix.deviceBytes = ix.sides.bytes() + ix.ftab.bytes() + ix.eftab.bytes() + ix.offs.bytes() + ix.boundRow.bytes() +
                     ix.boundRef.bytes() + ix.boundBits.bytes() + ix.refInfo.bytes() + ix.paths.bytes() + ix.pathTidx.bytes();
}

int envInt(const char *name, int dflt);
int persistentBlocks(const cf_index &ix, uint64_t groups, int blocksPerCU, int lanes);

// HBM a derived table may still take: what is free on the device, and no more than what is left of the caller's budget
size_t freeFor(const cf_index &ix) {
    size_t freeB = 0, totalB = 0;
    HIP_OK(hipMemGetInfo(&freeB, &totalB));
    if (ix.opt.hbm_budget_bytes) {
        const uint64_t left = ix.opt.hbm_budget_bytes > ix.deviceBytes ? ix.opt.hbm_budget_bytes - ix.deviceBytes : 0;
        freeB = (size_t)std::min<uint64_t>(freeB, left);
    }
    return freeB;
}

// ---- which derived tables to make.  Each table alone is easy to size; which COMBINATION buys most under a budget depends on
// the index: a 103 Gbp text cannot have the planes (n bytes) beside text tables at every 2nd row — but it can beside text
// tables at every 32nd row without a resolve table, and that halves the cost of a read (an LF step over the sides costs
// the CU's L1 four (load x line) pairs per chain, one over the planes).  So cf_index_open enumerates the combinations
// (wide-ftab bases, text-table rate, planes, resolve-table rate, pair planes: a few hundred), prices each with the model below
// — (load x line) pairs per 100-base read, fitted to the measured op counts of configs 2, 4 and 5 — and takes the
// cheapest one that fits what the device (or the caller's budget) leaves after the files and a reserve for the batch
// slots.  Fields of cf_index_options the caller set, and the environment knobs, are constraints of the search.
struct TablePlan { int K, textRate, planes, resolveRate, pair; double cost; uint64_t bytes; int dropSides = 0; };

double tableCost(double log4n, int ftc, int offRate, int K, int textRate, int planes, int resolveRate, int pair, double repeatFrac = 0.0, bool multi = false) {
    const double calls = 6.5, rows = 1.42;
    // an LF step over the sides (two lanes, four loads each) against one over the planes: 4 x the (load x line) pairs, but measured
    // (round 2: the same step counts over sides and planes, 13.2 vs 11.4 ms; config 5: planes without text tables 0.66 x sides
    // with them at 2.3 x the steps) it is ~1.4 x the time — a chain waits for its one round trip either way
    const double step = planes ? 1.0 : 1.4;
    const int k = K > ftc ? K : ftc;
    const double twoRow = calls * (std::max(0.0, log4n - k) + 1.4) * (pair ? 0.62 : 1.0);
    // (samples at every row, rate 0: no step to a sampled row before the text, none back from the inverse sample after it)
    // (round 6, hits in the position form: no inverse-sample read, no steps back from a sampled position — what is left of a
    // coarser sample's price is the way TO a sampled row — and the rows of one-row hits are resolved without a walk; measured on
    // configs 4 and 5, profiles/r06g_*: single-row steps 11.8 per 150-base mate at every 8th row, 28.7 per 250-base read at every 16th)
    const bool posForm = textRate >= 0 && envInt("CF_POS_HITS", 1) != 0;
    const double single = textRate < 0 ? 67.6 : textRate == 0 ? 6.0 : 6.0 + (posForm ? 0.45 : 0.7) * (double)(1u << textRate);
    const double verify = textRate < 0 ? 0.0 : posForm ? 3.6 : 5.0;
    const double lookups = calls * (K > ftc ? 1.0 : 2.0);       // wide entry, or the 10-mer pair (+ its first steps in twoRow)
    const double records = 8.0;
    const double walkSteps = resolveRate == 0 ? 0.0 : ((double)(1u << resolveRate) - 1.0) / 2.0;
    const double walk = (posForm ? 0.4 : 1.0) * rows * (1.0 + walkSteps * 2.0 * step + (resolveRate == 0 ? 0.0 : 0.5));
    // A repeat-rich collection (repeatFrac: the share of neighbouring rows that share their preceding 24 bases, cf_index::repeatFrac):
    // the matching strand's range stays a few rows wide for that share of the read's bases beyond the lookup — a step per base, per
    // two over the pair planes (the repeat-rich stand-in: 31 of its 43.5 requests) — unless small ranges are finished against the
    // text (multi: SA of every row + its windows + one inverse-SA read; 31 -> 14 measured at repeatFrac 0.6)
    const double wideSteps = multi ? 0.0 : repeatFrac * std::max(0.0, 100.0 - k) * (pair ? 0.5 : 1.0);
    const double multiReq = multi ? 23.0 * repeatFrac : 0.0;
    (void)offRate;
    return (twoRow + single + wideSteps) * step + verify + lookups + records + walk + multiReq;
}

// Round 6: what the tables cost to MAKE, in seconds — so that a caller who says how large its job is (cf_index_options::
// expected_reads) gets the plan that finishes the job soonest, not the one that would classify an endless stream fastest.  Fitted to
// the build times cf_index_describe reports: the wide ftab 1.2 - 1.7 s for 4^16 entries; the text tables (one
// inverse-BWT pass whatever the sample rate) 2.6 - 3.0 s and the resolve table at every row 2.8 s at 8.6 Gbp by walks (0.4 s from the stop rows'
// positions, where the text tables hold SA[row] for every row: densifyIndex); planes 25 ms, pair
// planes 0.3 s.  A cost unit of tableCost is 0.022 ns of a read's time (sides alone: 218 units = round 1's 2.0e8 reads/s; all
// tables: 25 units = the search's 0.55 ns).
constexpr double kSecondsPerCostUnit = 0.022e-9;
static double tableBuildSeconds(uint64_t n, int ftc, int offRate, int K, int textRate, int planes, int resolveRate, int pair) {
    double s = 0;
    if (K > ftc) s += 0.36e-9 * std::pow(4.0, (double)K);
    if (textRate >= 0) s += 0.35e-9 * (double)n;
    if (resolveRate < offRate) s += (resolveRate == 0 && textRate == 0 ? 0.05e-9 : 0.33e-9) * (double)(n >> resolveRate);   // (every row beside SA at every row: from the stops' positions, 0.4 s at 8.6 Gbp)
    if (planes) s += 0.003e-9 * (double)n;
    if (pair) s += 0.035e-9 * (double)n;
    return s;
}

// marks of the inverse-BWT walks (restoreCore): every 2^shift-th row
static uint32_t restoreShiftFor(uint64_t n) {
    int lg = 0;
    while ((n >> lg) > 1) lg++;
    const char *es = cfamd::cf_knob("CF_RESTORE_SHIFT");
    uint32_t sh = es ? (uint32_t)std::atoi(es) : (uint32_t)std::min(10, std::max(4, lg - 18));
    while ((n >> sh) + 3 >= 0xffffffffull) sh++;                                  // 32-bit segment ids
    return sh;
}
// The inverse sample's rate beside an SA sample at every 2^textRate-th row (DIndex::isaRate): three steps coarser (at most every
// 64th position) where the hits of unique matches take their position form — whatever then still asks for a row from a position
// is rare and can take the steps back — else the same.  multi: small ranges against the text read it for every range (rate 0).
static int isaRateFor(uint64_t n, int offRate, int textRate, bool multi) {
    if (textRate < 0 || multi || !envInt("CF_POS_HITS", 1) || (int)restoreShiftFor(n) < offRate) return textRate;
    if (cfamd::cf_knob("CF_ISA_RATE")) return std::max(textRate, std::min(6, envInt("CF_ISA_RATE", textRate)));
    return std::min(6, textRate + 3);
}
static uint64_t textTableBytes(uint64_t n, int offRate, int textRate, bool multi) {
    if (textRate < 0) return 0;
    return 8 * (trio_words((n >> textRate) + 2) + trio_words((n >> isaRateFor(n, offRate, textRate, multi)) + 2)) + n / 4 + (n >> 5) + 512;
}

// rows of the small ranges that are finished against the text (DIndex::multiRows): cf_index_options::small_range_rows (n = that many,
// -1 = off, 0 = automatic: four where the probe finds the collection repeat-rich), or CF_MULTI_VERIFY
constexpr double kRepeatFracMin = 0.10;
static uint32_t smallRangeRows(const cf_index &ix) {
    if (cfamd::cf_knob("CF_MULTI_VERIFY")) return (uint32_t)std::clamp(envInt("CF_MULTI_VERIFY", 0), 0, 15);
    if (ix.opt.small_range_rows < 0) return 0;
    if (ix.opt.small_range_rows > 0) return (uint32_t)std::clamp(ix.opt.small_range_rows, 2, 15);
    return ix.repeatFrac >= kRepeatFracMin ? 4u : 0u;          // automatic: where the collection's ranges stay wide (k_repeat_probe)
}
// the repeat fraction the cost model prices with: measured, or — the caller asks for small ranges without a device to measure
// on (cf_debug_plan_tables) — that of the repeat-rich stand-in
static double repeatFracOf(const cf_index &ix) {
    return ix.repeatFrac >= 0 ? ix.repeatFrac : (ix.opt.small_range_rows > 0 || cfamd::cf_knob("CF_MULTI_VERIFY")) ? 0.6 : 0.0;
}

// env knob (if set) or option field (if not 0) as a constraint: returns true and the value the enumeration must keep to
bool fixedKnob(const char *env, int32_t opt, int &v) {
    if (cfamd::cf_knob(env)) { v = envInt(env, 0); return true; }
    if (opt != 0) { v = opt; return true; }
    return false;
}

// lateRoom: bytes that become free only once the planes, the text tables and the resolve table exist (the sides of a plan that
// drops them: cf_index_open_ex makes those three first, lets the sides go, then makes the wide ftab and the pair planes)
static TablePlan planTablesIn(const cf_index &ix, uint64_t room, bool needPlanes, uint64_t lateRoom = 0) {
    const double repeatFrac = repeatFracOf(ix);
    const uint64_t n = ix.h.g.len;
    const int ftc = ix.h.g.ftabChars, offRate = ix.h.g.offRate;
    const uint64_t width = ix.h.offw ? 4 : 2;
    const double log4n = n > 1 ? std::log((double)n) / std::log(4.0) : 0.0;
    int kAuto = 0;
    for (uint64_t m = n; m >= 4; m >>= 2) kAuto++;
    kAuto = std::min(kAuto, 16);
    // what may be tried per table: everything (automatic), the caller's / the environment's value or nothing (a fixed value is
    // made while it fits), or nothing (switched off).  "Nothing" = K <= ftabChars, text rate -1, resolve rate = offRate.
    int v;
    std::vector<int> Ks, Ts, Ps, Rs, Qs;
    if (fixedKnob("CF_WIDE_FTAB", ix.opt.wide_ftab_chars, v)) { if (v > ftc && v <= 16 && n < (1ull << 40)) Ks.push_back(v); }
    else if (n < (1ull << 40)) for (int K = kAuto; K > ftc && K >= kAuto - 3; K--) Ks.push_back(K);
    Ks.push_back(ftc);
    if (fixedKnob("CF_TEXT_VERIFY_RATE", ix.opt.text_verify_rate, v)) { if (v >= 0 && v <= 5 && n >= 64 && (v > 0 || cfamd::cf_knob("CF_TEXT_VERIFY_RATE"))) Ts.push_back(v); }
    else if (n >= 64 && ix.wantTextRate0) Ts.push_back(0);         // (small ranges against the text: the samples at every row, or — planTables — not at all)
    else if (n >= 64) for (int r = 0; r <= 5; r++) Ts.push_back(r);   // (every row — 10.7 bytes per base — where the room is there: round 5)
    Ts.push_back(-1);
    if (fixedKnob("CF_OCC_PLANES", ix.opt.occ_planes, v)) { if (v > 0) Ps.push_back(1); } else Ps.push_back(1);
    Ps.push_back(0);
    if (fixedKnob("CF_DENSE_SA_RATE", ix.opt.resolve_rate, v)) {
        const int rr = cfamd::cf_knob("CF_DENSE_SA_RATE") ? v : v - 1;          // (the option field holds rate + 1, -1 = none)
        if (rr >= 0 && rr < offRate) Rs.push_back(rr);
    } else for (int rr = 0; rr <= 3 && rr < offRate; rr++) Rs.push_back(rr);
    Rs.push_back(offRate);
    if (fixedKnob("CF_PAIR_PLANES", ix.opt.pair_planes, v)) { if (v > 0) Qs.push_back(1); } else Qs.push_back(1);
    Qs.push_back(0);
    const uint64_t planesB = ix.h.g.numSides * 384, pairB = ((n + 64) / 64 + 1) * 256;
    TablePlan best{0, -1, 0, offRate, 0, 1e300, 0};
    bool any = false;
    for (int K : Ks) for (int tr : Ts) for (int pl : Ps) for (int rr : Rs) for (int pp : Qs) {
        if (pp && !pl) continue;                                 // the pair planes are made from the planes
        if (needPlanes && !pl) continue;
        const uint64_t wideB = K > ftc ? (8ull << (2 * K)) + 16 : 0;
        const uint64_t textB = textTableBytes(n, offRate, tr, tr == 0 && (ix.wantTextRate0 || smallRangeRows(ix) >= 2));      // (as textifyIndex will make them)
        // (a denser resolve table REPLACES the file's SA sample in HBM — no kernel reads that once the table exists —, so only
        // the difference counts against the room)
        const uint64_t offsB = ((n >> offRate) + 1) * width;
        const uint64_t resB = rr >= offRate ? 0 : ((n >> rr) + 3) * width - std::min<uint64_t>(offsB, ((n >> rr) + 3) * width);
        const uint64_t early = textB + (pl ? planesB : 0) + resB, bytes = early + wideB + (pp ? pairB : 0);
        if (early > room || bytes > room + lateRoom) continue;
        const bool multi = tr == 0 && pl && ix.wantTextRate0;          // (small ranges against the text: the planes kernel, samples at every row)
        double c = tableCost(log4n, ftc, offRate, K, tr, pl, rr, pp, repeatFrac, multi);
        // a job of known size: the seconds the tables take to make, in the cost units those seconds are worth to each of its reads
        if (ix.opt.expected_reads) c += tableBuildSeconds(n, ftc, offRate, K, tr, pl, rr, pp) / ((double)ix.opt.expected_reads * kSecondsPerCostUnit);
        if (!any || c < best.cost - 1e-9 || (std::fabs(c - best.cost) <= 1e-9 && bytes < best.bytes)) { best = TablePlan{K > ftc ? K : 0, tr, pl, rr, pp, c, bytes}; any = true; }
    }
    if (!any) best.cost = 1e300;
    return best;
}

// ... and with the planes an index can do without its SIDES in HBM (every LF step reads the planes; the sides are the same BWT
// once more, 1/3 byte per base): the plan with the sides' bytes added to the room and the planes required is taken when — and
// only when — it is cheaper than the best plan that keeps them (the nt-scale index: the planes instead of the two-lane kernel).
// cf_index_options::sides: 1 = always keep them, -1 = drop them whenever the planes are made.
static TablePlan planTablesSides(const cf_index &ix, uint64_t room);
TablePlan planTables(const cf_index &ixIn, uint64_t room) {
    // small ranges against the text want the samples at every row (10.7 bytes per base instead of 5.3): that plan when the model —
    // which knows what they save on a collection this repeat-rich (tableCost) — prices it below the usual one
    if (smallRangeRows(ixIn) >= 2 && !cfamd::cf_knob("CF_TEXT_VERIFY_RATE") && ixIn.opt.text_verify_rate == 0) {
        cf_index probe;
        probe.h.g = ixIn.h.g; probe.h.offw = ixIn.h.offw; probe.opt = ixIn.opt; probe.repeatFrac = ixIn.repeatFrac; probe.wantTextRate0 = true;
        const TablePlan t0 = planTablesSides(probe, room), usual = planTablesSides(ixIn, room);
        return t0.textRate == 0 && t0.planes && t0.cost < usual.cost ? t0 : usual;
    }
    return planTablesSides(ixIn, room);
}
static TablePlan planTablesSides(const cf_index &ix, uint64_t room) {
    const TablePlan keep = planTablesIn(ix, room, false);
    const int pol = cfamd::cf_knob("CF_DROP_SIDES") ? (envInt("CF_DROP_SIDES", 0) ? -1 : 1) : ix.opt.sides;
    if (pol > 0) return keep;
    TablePlan drop = planTablesIn(ix, room, true, ix.h.g.numSides * 128);
    drop.dropSides = 1;
    if (drop.cost >= 1e300) return keep;
    if (pol < 0) return drop;
    return drop.cost < keep.cost - 1e-9 ? drop : keep;
}

void dropSides(cf_index &ix) {
    if (!ix.d.planes || !ix.sides.p) return;
    ix.deviceBytes -= ix.sides.bytes(); ix.droppedBytes += ix.sides.bytes();
    ix.sides.release();
    ix.d.sides = nullptr;
    ix.sidesDropped = true;
}

// The dense resolve table (walk2_body): the answer of the walk-left loop for every 2^rate-th row, computed by the walk
// kernel itself from the file's SA sample.  rate: CF_DENSE_SA_RATE (0 = every row ... offRate = the file's own sample,
// i.e. off); default 1 (every 2nd row: n bytes with a u16 sample, a walk of 1.4 steps on average instead of 15) as
// long as the table stays under half of the HBM that is still free (it is made last).
void densifyIndex(cf_index &ix) {
    const int offRate = ix.h.g.offRate;
    // every row when that fits a third of what is free (a resolved row is then ONE table read: no LF step, no boundary check),
    // else every 2nd / 4th / 8th as long as the table stays under half of it
    int rate = cfamd::cf_knob("CF_DENSE_SA_RATE") ? envInt("CF_DENSE_SA_RATE", 1) : ix.opt.resolve_rate < 0 ? offRate : ix.opt.resolve_rate > 0 ? ix.opt.resolve_rate - 1 : 0;
    if (rate < 0 || rate >= offRate) return;
    const size_t width = ix.h.offw ? 4 : 2;
    const size_t freeB = freeFor(ix);
    if (ix.planned) { if (((ix.h.g.len >> rate) + 2) * width > freeB) return; }     // the planner's choice: made as it is while it fits
    else {
        if (rate == 0 && (ix.h.g.len + 2) * width > freeB / 3) rate = 1;
        while (rate < offRate && ((ix.h.g.len >> rate) + 2) * width > freeB / 2) rate++;
    }
    if (rate >= offRate) return;
    const uint64_t count = (ix.h.g.len >> rate) + 1;         // rows 0 .. len
    ix.dense.alloc((count + 2) * width);
    DevBuf<unsigned long long> cursor; DevBuf<BatchStatus> st;
    cursor.alloc(4); st.alloc(1);
    HIP_OK(hipMemset(cursor.p, 0, 32));
    BatchStatus hs{};
    hs.rowLo = 0; hs.rowHi = count;
    HIP_OK(hipMemcpy(st.p, &hs, sizeof hs, hipMemcpyHostToDevice));
    DBatch b{};
    b.rowRef = reinterpret_cast<uint32_t *>(ix.dense.p); b.cursor = cursor.p; b.st = st.p; b.genShift = (uint32_t)rate;
    DevBuf<uint32_t> walkMax;                        // the longest walk (at rate 0 — a walk from every row — the bound resolve_pos rests on)
    walkMax.alloc(1);
    HIP_OK(hipMemset(walkMax.p, 0, 4));
    b.walkMaxOut = walkMax.p;
    hipEvent_t e0, e1;
    HIP_OK(hipEventCreate(&e0)); HIP_OK(hipEventCreate(&e1));
    HIP_OK(hipEventRecord(e0, nullptr));
    const dim3 gr(persistentBlocks(ix, count, 8, 2)), bl(256);
    // every row, and the text tables hold SA[row] for every row: the table from the positions of the stop rows (k_table_by_position)
    // — while its scratch (a value per position, a bit per position) fits what is free beside the table; CF_DENSE_BY_POS=0: by walks
    const uint64_t n1 = ix.h.g.len + 1;
    bool byPos = rate == 0 && ix.d.saPos && ix.d.posRate == 0 && ix.offs.p && envInt("CF_DENSE_BY_POS", 1) != 0 &&
                 freeFor(ix) > (n1 + 64) * width + n1 / 8 + (2ull << 30);
    if (byPos) {
        DevBuf<uint8_t> stopVal; DevBuf<uint64_t> stopBits;
        stopVal.alloc((n1 + 64) * width); stopBits.alloc(n1 / 64 + 2);
        HIP_OK(hipMemsetAsync(stopBits.p, 0, (n1 / 64 + 2) * 8, nullptr));
        const uint64_t nSamples = (ix.h.g.len >> offRate) + 1;
        const uint64_t cap = (uint64_t)ix.numCUs * 64;                  // blocks of 256: eight per SIMD, grid-stride beyond
        const dim3 gs((unsigned)std::min<uint64_t>((nSamples + 255) / 256, cap)), gb((unsigned)std::min<uint64_t>(((uint64_t)ix.d.nBound + 255) / 256, cap)), gt((unsigned)std::min<uint64_t>((count + 255) / 256, cap));
        const bool bounds = ix.d.lastBoundary > 0 && ix.d.nBound > 0;           // (tryOffset looks at the boundary rows only then)
        if (ix.h.offw) {
            uint32_t *sv = reinterpret_cast<uint32_t *>(stopVal.p);
            if (bounds) hipLaunchKernelGGL(k_stop_bounds<uint32_t>, gb, bl, 0, nullptr, ix.d, sv, stopBits.p);
            hipLaunchKernelGGL(k_stop_samples<uint32_t>, gs, bl, 0, nullptr, ix.d, sv, stopBits.p, nSamples);
            hipLaunchKernelGGL(k_stop_end<uint32_t>, dim3(1), dim3(1), 0, nullptr, ix.d, sv, stopBits.p);
            hipLaunchKernelGGL(k_table_by_position<uint32_t>, gt, bl, 0, nullptr, ix.d, (const uint32_t *)sv, (const uint64_t *)stopBits.p, reinterpret_cast<uint32_t *>(ix.dense.p), count, walkMax.p);
        } else {
            uint16_t *sv = reinterpret_cast<uint16_t *>(stopVal.p);
            if (bounds) hipLaunchKernelGGL(k_stop_bounds<uint16_t>, gb, bl, 0, nullptr, ix.d, sv, stopBits.p);
            hipLaunchKernelGGL(k_stop_samples<uint16_t>, gs, bl, 0, nullptr, ix.d, sv, stopBits.p, nSamples);
            hipLaunchKernelGGL(k_stop_end<uint16_t>, dim3(1), dim3(1), 0, nullptr, ix.d, sv, stopBits.p);
            hipLaunchKernelGGL(k_table_by_position<uint16_t>, gt, bl, 0, nullptr, ix.d, (const uint16_t *)sv, (const uint64_t *)stopBits.p, reinterpret_cast<uint16_t *>(ix.dense.p), count, walkMax.p);
        }
        HIP_OK(hipGetLastError());
        DevBuf<uint32_t> bad; bad.alloc(1);
        HIP_OK(hipMemsetAsync(bad.p, 0, 4, nullptr));
        const uint32_t nCheck = 65536;
        if (ix.h.offw) hipLaunchKernelGGL(k_table_spot_check<uint32_t>, dim3(nCheck / 256), bl, 0, nullptr, ix.d, reinterpret_cast<const uint32_t *>(ix.dense.p), count, nCheck, bad.p);
        else hipLaunchKernelGGL(k_table_spot_check<uint16_t>, dim3(nCheck / 256), bl, 0, nullptr, ix.d, reinterpret_cast<const uint16_t *>(ix.dense.p), count, nCheck, bad.p);
        HIP_OK(hipGetLastError());
        uint32_t nBad = 0;
        HIP_OK(hipMemcpy(&nBad, bad.p, 4, hipMemcpyDeviceToHost));      // (synchronises; the scratch goes out of scope below)
        if (nBad || envInt("CF_DENSE_BY_POS", 1) == 2) {                // (2: the test of this way back)
            byPos = false;
            HIP_OK(hipMemset(walkMax.p, 0, 4));
        }
    }
    if (byPos) {}
    else if (ix.h.offw) hipLaunchKernelGGL(k_walk_table<WALK_TABLE32>, gr, bl, 0, nullptr, ix.d, b);
    else hipLaunchKernelGGL(k_walk_table<WALK_TABLE16>, gr, bl, 0, nullptr, ix.d, b);
    ix.denseByPos = byPos;
    HIP_OK(hipEventRecord(e1, nullptr));
    HIP_OK(hipEventSynchronize(e1));
    HIP_OK(hipGetLastError());
    HIP_OK(hipEventElapsedTime(&ix.denseMs, e0, e1));
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    HIP_OK(hipMemcpy(&ix.walkMaxSeen, walkMax.p, 4, hipMemcpyDeviceToHost));
    ix.d.walkOffs = ix.dense.p; ix.d.walkRate = rate;
    ix.denseRate = rate;
    ix.deviceBytes += ix.dense.bytes();
    // the file's own sample has done its work: every kernel resolves rows through the table now
    ix.deviceBytes -= ix.offs.bytes(); ix.droppedBytes += ix.offs.bytes();
    ix.offs.release();
    ix.d.offs = nullptr;
}

// Position -> reference (DIndex::posFrag, resolve_pos): made whenever the text tables are (any sample rate) and a bound on the
// walk-left is at hand — exact where the resolve table holds every row (densifyIndex took the walk from every row: config 2's
// plan), else the longest segment of the inverse-BWT walks that made the text tables.
// A few MB: one u32 per 16 K positions, 16 bytes per fragment and per sequence.  CF_POS_HITS=0: not made (no hit takes the form).
void makePosTables(cf_index &ix) {
    ix.d.posFrag = nullptr; ix.d.posBucket = nullptr; ix.d.posSeq = nullptr; ix.d.nPosFrag = 0; ix.d.walkMax = 0; ix.d.posShift = 14;
    if (!envInt("CF_POS_HITS", 1) || ix.d.posRate < 0 || !ix.d.isa) return;
    // the bound on the walk-left: exact from the resolve table's build when that took the walk from every row, else the longest
    // segment of the inverse-BWT walks (valid while their marks are a subset of the sample's rows)
    uint32_t bound = 0;
    if (ix.denseRate == 0 && ix.dense.p) bound = ix.walkMaxSeen;
    else if (ix.restoreMaxSeg && ix.restoreMaxSeg != 0xffffffffu && (int)ix.restoreShift >= ix.h.g.offRate) bound = ix.restoreMaxSeg;
    else return;
    const uint64_t n = ix.h.g.len, nFrag = ix.h.rstarts.size() / 3;
    if (nFrag == 0 || nFrag >= 0xfffffff0ull || n >= (1ull << 39)) return;
    std::vector<u64x2> frag(nFrag), seq(ix.h.nPat + 1, u64x2{0, 0});
    for (uint64_t i = 0; i < nFrag; i++) {
        const uint64_t at = ix.h.rstarts[3 * i], sq = ix.h.rstarts[3 * i + 1];
        if (sq >= ix.h.nPat || at > n || (i && at < ix.h.rstarts[3 * (i - 1)])) return;      // not a list this code understands: no position form
        frag[i] = u64x2{at, sq};
    }
    // a sequence's span in the joined text: from its first fragment to the first fragment of the next sequence that has one
    for (uint64_t i = 0; i < nFrag; i++) {
        const uint64_t sq = frag[i].y;
        if (i == 0 || frag[i - 1].y != sq) {
            if (seq[sq].y != 0) return;                                                      // (a sequence in two runs of fragments)
            seq[sq].x = frag[i].x;
            uint64_t j = i;
            while (j < nFrag && frag[j].y == sq) j++;
            seq[sq].y = j < nFrag ? frag[j].x : n;
        }
    }
    const uint32_t sh = 14;
    const uint64_t nB = (n >> sh) + 2;
    std::vector<uint32_t> bucket(nB + 1);
    uint64_t f = 0;
    for (uint64_t b = 0; b <= nB; b++) {                      // the last fragment that starts at or before the bucket's first position
        const uint64_t p0 = b << sh;
        while (f + 1 < nFrag && frag[f + 1].x <= p0) f++;
        bucket[b] = (uint32_t)f;
    }
    ix.posBucket.upload(bucket); ix.posFrag.upload(frag); ix.posSeq.upload(seq);
    ix.deviceBytes += ix.posBucket.bytes() + ix.posFrag.bytes() + ix.posSeq.bytes();
    ix.d.posBucket = ix.posBucket.p; ix.d.posFrag = ix.posFrag.p; ix.d.posSeq = ix.posSeq.p;
    ix.d.posShift = sh; ix.d.nPosFrag = (uint32_t)nFrag; ix.d.walkMax = bound;
}

// The occurrence planes (occ_planes_body): 384 bytes per side (8 bits per base) next to the side's 128, one thread per side.
// CF_OCC_PLANES=0: not made (the search kernel then reads the sides, two lanes per chain); also skipped when they would
// take more than 60 % of the HBM that is free once the wide ftab and the text tables are made.
void planifyIndex(cf_index &ix) {
    if (cfamd::cf_knob("CF_OCC_PLANES") ? !envInt("CF_OCC_PLANES", 1) : ix.opt.occ_planes < 0) return;
    const uint64_t nSides = ix.h.g.numSides;
    const size_t freeB = freeFor(ix);
    if ((double)nSides * 384 > (ix.planned ? 1.0 : 0.6) * (double)freeB) return;
    ix.planes.alloc(nSides * 384);
    hipEvent_t e0, e1;
    HIP_OK(hipEventCreate(&e0)); HIP_OK(hipEventCreate(&e1));
    HIP_OK(hipEventRecord(e0, nullptr));
    hipLaunchKernelGGL(k_occ_planes, dim3((unsigned)((nSides + 255) / 256)), dim3(256), 0, nullptr, ix.d, ix.planes.p, nSides);
    HIP_OK(hipEventRecord(e1, nullptr));
    HIP_OK(hipEventSynchronize(e1));
    HIP_OK(hipGetLastError());
    HIP_OK(hipEventElapsedTime(&ix.planesMs, e0, e1));
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    ix.d.planes = ix.planes.p;
    ix.deviceBytes += ix.planes.bytes();
}

// The pair planes (pair_planes_body): 4 bytes per base next to the planes' 1, one thread per group of 64 rows.  Made last, when
// the planes exist and the table takes at most a third of what is still free (CF_PAIR_PLANES=0 / 1 decides by hand).
void pairPlanifyIndex(cf_index &ix) {
    if (!ix.d.planes) return;
    const bool forced = cfamd::cf_knob("CF_PAIR_PLANES") ? envInt("CF_PAIR_PLANES", 0) != 0 : ix.opt.pair_planes > 0;
    if (cfamd::cf_knob("CF_PAIR_PLANES") ? !envInt("CF_PAIR_PLANES", 1) : ix.opt.pair_planes < 0) return;
    const uint64_t nGroups = (ix.h.g.len + 64) / 64;             // rows 0 .. len
    const size_t freeB = freeFor(ix);
    if ((double)nGroups * 256 > (ix.planned ? 1.0 : forced ? 0.9 : 1.0 / 3) * (double)freeB) return;
    // Rows 0 .. len have a group.  No range ever ends beyond row len: the empty suffix sorts LAST (row len), so an LF step gives at
    // most fchr[4] = len, and the last 10-mer's bot is eftab[2 ftabChars - 2] = len (bt2_idx.h:1953-1970) — a step's top and bot
    // both index groups < nGroups.  One zeroed group behind them all the same: a damaged index must not send a load off the table.
    ix.planes2.alloc((nGroups + 1) * 256);
    HIP_OK(hipMemset(ix.planes2.p + nGroups * 256, 0, 256));
    hipEvent_t e0, e1;
    HIP_OK(hipEventCreate(&e0)); HIP_OK(hipEventCreate(&e1));
    HIP_OK(hipEventRecord(e0, nullptr));
    hipLaunchKernelGGL(k_pair_planes, dim3((unsigned)((nGroups + 255) / 256)), dim3(256), 0, nullptr, ix.d, ix.planes2.p, nGroups);
    HIP_OK(hipEventRecord(e1, nullptr));
    HIP_OK(hipEventSynchronize(e1));
    HIP_OK(hipGetLastError());
    HIP_OK(hipEventElapsedTime(&ix.planes2Ms, e0, e1));
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    ix.d.planes2 = ix.planes2.p;
    ix.deviceBytes += ix.planes2.bytes();
}

// The wide ftab (wide_ftab_body).  Bases per entry: CF_WIDE_FTAB (0 = off), default = floor(log4 n) — about one row left per
// entry, so that a call that dies early (the wrong strand, a read from elsewhere) is answered by the lookup alone — at most
// 16 (8 bytes x 4^16 = 34 GB) and only when the table stays under a sixth of the free HBM.
void widenFtab(cf_index &ix) {
    const int ftc = ix.h.g.ftabChars;
    int k = cfamd::cf_knob("CF_WIDE_FTAB") ? envInt("CF_WIDE_FTAB", -1) : ix.opt.wide_ftab_chars < 0 ? 0 : ix.opt.wide_ftab_chars > 0 ? ix.opt.wide_ftab_chars : -1;
    if (k < 0) {
        k = 0;
        for (uint64_t m = ix.h.g.len; m >= 4; m >>= 2) k++;
        k = std::min(k, 16);
    }
    if (k <= ftc || k > 16 || ix.h.g.len >= (1ull << 40)) return;
    const size_t freeB = freeFor(ix);
    while (k > ftc && (8ull << (2 * k)) > (ix.planned ? freeB : freeB / 6)) k--;
    if (k <= ftc) return;
    const uint64_t entries = 1ull << (2 * k);
    ix.wide.alloc(entries + 2);                  // (the search kernel reads 16 bytes at an entry)
    hipEvent_t e0, e1;
    HIP_OK(hipEventCreate(&e0)); HIP_OK(hipEventCreate(&e1));
    HIP_OK(hipEventRecord(e0, nullptr));
    const dim3 gw((unsigned)std::min<uint64_t>((entries + 255) / 256, 1u << 22));
    if (ix.d.planes) hipLaunchKernelGGL(k_wide_ftab<true>, gw, dim3(256), 0, nullptr, ix.d, (uint32_t)k, ix.wide.p);
    else hipLaunchKernelGGL(k_wide_ftab<false>, gw, dim3(256), 0, nullptr, ix.d, (uint32_t)k, ix.wide.p);
    HIP_OK(hipEventRecord(e1, nullptr));
    HIP_OK(hipEventSynchronize(e1));
    HIP_OK(hipGetLastError());
    HIP_OK(hipEventElapsedTime(&ix.wideMs, e0, e1));
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    ix.d.wide = ix.wide.p; ix.d.wideChars = k;
    ix.deviceBytes += ix.wide.bytes();
}

// (every CF_* knob goes through cf_knob: honoured only under CF_DEBUG_KNOBS=1, cf_knobs.hpp)
int envInt(const char *name, int dflt) {
    const char *v = cfamd::cf_knob(name);
    return v && *v ? std::atoi(v) : dflt;
}

// resident blocks per CU of the persistent kernels (tuning knob; the default is the measured best)
int blocksPerCU() { static const int b = envInt("CF_BLOCKS_PER_CU", 8); return b; }
// CF_SEARCH_V=1 forces the packed-word search kernel (k_search) that reads > 256 bases always take
int searchVersion() { static const int v = envInt("CF_SEARCH_V", 2); return v; }

int persistentBlocks(const cf_index &ix, uint64_t groups, int blocksPerCU, int lanes) {
    const uint64_t want = (groups * lanes + 255) / 256;
    const uint64_t cap = (uint64_t)ix.numCUs * blocksPerCU;
    return (int)std::max<uint64_t>(1, std::min(want, cap));
}

template <typename F>
cf_status guard(F &&f) {
    try {
        f();
        return CF_OK;
    } catch (const HipError &e) { g_err = e.what(); return CF_ERR_HIP;
    } catch (const ArgError &e) { g_err = e.what(); return CF_ERR_ARG;
    } catch (const std::bad_alloc &) { g_err = "out of host memory"; return CF_ERR_NOMEM;
    } catch (const std::exception &e) {
        g_err = e.what();
        return g_err.find("cannot open") != std::string::npos ? CF_ERR_IO : CF_ERR_FORMAT;
    }
}

// resident blocks per CU of the persistent search kernels (cf_index::occSides / occPlanes)
void queryOccupancy(cf_index &ix) {
    auto ask = [](auto kernel, int dflt) { int n = 0; return hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, kernel, 256, 0) == hipSuccess && n > 0 ? n : dflt; };
    const int d = blocksPerCU();
    ix.occSides[0] = ask(k_search2<2, 4, false>, d); ix.occSides[1] = ask(k_search2<2, 6, false>, d); ix.occSides[2] = ask(k_search2<2, 8, false>, d);
    ix.occPlanes[0] = ask(k_search2_l1<4, false>, 4); ix.occPlanes[1] = ask(k_search2_l1<6, false>, 4); ix.occPlanes[2] = ask(k_search2_l1<8, false>, 4);
    ix.occMulti[0] = ask(k_search2_l1<4, false, 0, true>, 4); ix.occMulti[1] = ask(k_search2_l1<6, false, 0, true>, 4); ix.occMulti[2] = ask(k_search2_l1<8, false, 0, true>, 4);
    ix.occLazy[0] = ask(k_search2_l1<4, false, 1>, 4); ix.occLazy[1] = ask(k_search2_l1<6, false, 2>, 4); ix.occLazy[2] = ask(k_search2_l1<8, false, 2>, 4);
    (void)hipGetLastError();
}

// the search kernel of a batch: k_search2 (strand records in LDS, one memory round trip per
// iteration) when every read fits its records, else the packed-word kernel k_search.  The number of work
// items is on the device (BatchStatus::nItems); the grid is sized by the reads the batch holds.
// Returns true when the launched kernel tallied the op counters (k_search always does; k_search2
// only in its instrumented build, which cf_batch_opcounts runs on demand).
bool launchSearch(cf_classifier *cl, cf_batch *bt, hipStream_t st, int blocksCap = 0, bool count = false) {
    cf_index &ix = *cl->ix;
    const bool v2 = bt->recWords == 4 || bt->recWords == 6 || bt->recWords == 8;
    const int wi = bt->recWords == 4 ? 0 : bt->recWords == 6 ? 1 : 2;
    int perCU = blocksPerCU();
    // persistent kernel: exactly the blocks that are resident at once (registers and LDS decide: 8 per CU for
    // 128-base records, 6 for 192-base and 5 for 256-base ones); more would only queue up behind them
    if (v2 && !cfamd::cf_knob("CF_BLOCKS_PER_CU") && ix.occSides[wi] > 0) perCU = ix.occSides[wi];
    int blocks = persistentBlocks(ix, 2 * bt->nReads, perCU, 2);
    if (blocksCap) blocks = std::min(blocks, blocksCap);
    const DBatch &d = bt->d;
    const dim3 gr(blocks), bl(256);
    if (v2 && ix.d.planes) {
        int per = ix.occPlanes[wi] > 0 ? ix.occPlanes[wi] : 4;
        if (cfamd::cf_knob("CF_BLOCKS_PER_CU")) per = std::min(per, blocksPerCU());
        int nb = persistentBlocks(ix, 2 * bt->nReads, per, 1);
        if (blocksCap) nb = std::min(nb, blocksCap);
        const dim3 g1(nb);
        // (CF_LAZY_N = 2: two lazy hits per strand for the 192- and 256-base records instead of one — LDS per block, and with it
        // the blocks per CU, against strands searched twice; measured per workload)
        static const int lazyN = envInt("CF_LAZY_N", 0);
        // a variant's own grid: the blocks that are resident at once with ITS registers and LDS (asked once per index and device: queryOccupancy)
        auto gridOf = [&](int pc) { const int n = persistentBlocks(ix, 2 * bt->nReads, pc > 0 ? pc : per, 1); return dim3(blocksCap ? std::min(blocksCap, n) : n); };
        if (ix.d.multiRows) {      // small ranges against the text (small_range_rows): kernels of their own — the states cost registers
            const dim3 gm = gridOf(ix.occMulti[wi]);
            if (bt->recWords == 4) { if (count) hipLaunchKernelGGL((k_search2_l1<4, true, 0, true>), gm, bl, 0, st, ix.d, cl->d, d); else hipLaunchKernelGGL((k_search2_l1<4, false, 0, true>), gm, bl, 0, st, ix.d, cl->d, d); }
            else if (bt->recWords == 6) { if (count) hipLaunchKernelGGL((k_search2_l1<6, true, 0, true>), gm, bl, 0, st, ix.d, cl->d, d); else hipLaunchKernelGGL((k_search2_l1<6, false, 0, true>), gm, bl, 0, st, ix.d, cl->d, d); }
            else { if (count) hipLaunchKernelGGL((k_search2_l1<8, true, 0, true>), gm, bl, 0, st, ix.d, cl->d, d); else hipLaunchKernelGGL((k_search2_l1<8, false, 0, true>), gm, bl, 0, st, ix.d, cl->d, d); }
        } else if (bt->recWords == 4) {
            if (count) hipLaunchKernelGGL((k_search2_l1<4, true>), g1, bl, 0, st, ix.d, cl->d, d);
            else if (lazyN == 1) hipLaunchKernelGGL((k_search2_l1<4, false, 1>), gridOf(ix.occLazy[0]), bl, 0, st, ix.d, cl->d, d);   // one lazy hit, 22.5 KB of LDS, 72 VGPRs: seven blocks per CU instead of six
            else hipLaunchKernelGGL((k_search2_l1<4, false>), g1, bl, 0, st, ix.d, cl->d, d);
        } else if (bt->recWords == 6) {
            if (count) hipLaunchKernelGGL((k_search2_l1<6, true>), g1, bl, 0, st, ix.d, cl->d, d);
            else if (lazyN == 2) hipLaunchKernelGGL((k_search2_l1<6, false, 2>), gridOf(ix.occLazy[1]), bl, 0, st, ix.d, cl->d, d);
            else hipLaunchKernelGGL((k_search2_l1<6, false>), g1, bl, 0, st, ix.d, cl->d, d);
        } else {
            if (count) hipLaunchKernelGGL((k_search2_l1<8, true>), g1, bl, 0, st, ix.d, cl->d, d);
            else if (lazyN == 2) hipLaunchKernelGGL((k_search2_l1<8, false, 2>), gridOf(ix.occLazy[2]), bl, 0, st, ix.d, cl->d, d);
            else hipLaunchKernelGGL((k_search2_l1<8, false>), g1, bl, 0, st, ix.d, cl->d, d);
        }
        return count;
    }
    if (v2) {
        if (bt->recWords == 4) { if (count) hipLaunchKernelGGL((k_search2<2, 4, true>), gr, bl, 0, st, ix.d, cl->d, d); else hipLaunchKernelGGL((k_search2<2, 4, false>), gr, bl, 0, st, ix.d, cl->d, d); }
        else if (bt->recWords == 6) { if (count) hipLaunchKernelGGL((k_search2<2, 6, true>), gr, bl, 0, st, ix.d, cl->d, d); else hipLaunchKernelGGL((k_search2<2, 6, false>), gr, bl, 0, st, ix.d, cl->d, d); }
        else { if (count) hipLaunchKernelGGL((k_search2<2, 8, true>), gr, bl, 0, st, ix.d, cl->d, d); else hipLaunchKernelGGL((k_search2<2, 8, false>), gr, bl, 0, st, ix.d, cl->d, d); }
        return count;
    }
    hipLaunchKernelGGL(k_search<2>, gr, bl, 0, st, ix.d, cl->d, d);
    return true;
}

// the walk over the rows of the current pass (their number is on the device: BatchStatus::rowLo/rowHi): one lane per row,
// grid-stride over what the pass holds (CF_WALK_V=2: the chain kernel k_walk2 instead)
bool launchWalk(cf_classifier *cl, cf_batch *bt, hipStream_t st, bool count = false) {
    cf_index &ix = *cl->ix;
    static const int wv = envInt("CF_WALK_V", 3);
    const uint64_t guess = std::min<uint64_t>(bt->d.rowsCap, std::max<uint64_t>(4 * bt->nQueries, 1024));
    const DBatch &d = bt->d;
    if (wv == 2) {
        const dim3 gr(persistentBlocks(ix, guess, blocksPerCU(), 2)), bl(256);
        if (count) hipLaunchKernelGGL((k_walk2<2, true>), gr, bl, 0, st, ix.d, d); else hipLaunchKernelGGL((k_walk2<2, false>), gr, bl, 0, st, ix.d, d);
        return count;
    }
    const dim3 gr((unsigned)std::min<uint64_t>((guess + 255) / 256, (uint64_t)ix.numCUs * 64)), bl(256);
    if (count) hipLaunchKernelGGL(k_walk3<true>, gr, bl, 0, st, ix.d, d); else hipLaunchKernelGGL(k_walk3<false>, gr, bl, 0, st, ix.d, d);
    return count;
}

// Inverse BWT (see cf_restore.hpp): pass 1 (segment lengths + links), list ranking, pass 2 (characters and, when asked for,
// the sampled suffix array and its inverse).  The 2-bit text stays on the device in `text`.
void restoreCore(cf_index &ix, DevBuf<uint32_t> &text, uint64_t *saPos, uint64_t *isa, uint32_t posShift, uint32_t isaShift) {
    const uint64_t n = ix.h.g.len;
    DRestore r{};
    r.n = n;
    r.shift = restoreShiftFor(n);
    r.nMarked = (uint32_t)(n >> r.shift) + 1;
    r.nSeg = r.nMarked + ((n & ((1ull << r.shift) - 1)) ? 1u : 0u);
    const uint32_t startSeg = r.nSeg - 1;                                      // the walk that starts at row n
    r.maxSteps = std::min<uint64_t>(n + 1, (1ull << r.shift) * 8192ull);
    const uint32_t nElem = r.nSeg + 1;
    DevBuf<uint64_t> sumA, sumB; DevBuf<uint32_t> nextA, nextB, cur, err;
    sumA.alloc(nElem); sumB.alloc(nElem); nextA.alloc(nElem); nextB.alloc(nElem); cur.alloc(4); err.alloc(1);
    const uint64_t words = (n + 15) / 16 + 16;                                 // + padding: the search kernel reads 32-byte windows
    text.alloc(words);
    HIP_OK(hipMemsetAsync(text.p, 0, words * 4, 0));
    HIP_OK(hipMemsetAsync(cur.p, 0, 16, 0));
    HIP_OK(hipMemsetAsync(err.p, 0, 4, 0));
    r.cursor = cur.p; r.segLen = sumA.p; r.segNext = nextA.p; r.err = err.p; r.text = text.p;
    const dim3 gr(persistentBlocks(ix, r.nSeg, blocksPerCU(), 2)), bl(256);
    const bool verbose = cfamd::cf_knob("CF_RESTORE_VERBOSE") != nullptr;
    struct Events {                                                            // released on every way out
        hipEvent_t e[4] = {};
        ~Events() { for (auto &x : e) if (x) (void)hipEventDestroy(x); }
    } evs;
    hipEvent_t *ev = evs.e;
    for (int i = 0; i < 4; i++) HIP_OK(hipEventCreate(&ev[i]));
    HIP_OK(hipEventRecord(ev[0], 0));
    hipLaunchKernelGGL((k_restore<2, false>), gr, bl, 0, 0, ix.d, r);
    HIP_OK(hipEventRecord(ev[1], 0));
    const dim3 ge((nElem + 255) / 256);
    HIP_OK(hipMemsetAsync(cur.p + 2, 0, 4, 0));                                // (cur[2]: the longest segment)
    hipLaunchKernelGGL(k_restore_link, ge, bl, 0, 0, sumA.p, nextA.p, r.nSeg, cur.p + 2);
    ix.restoreShift = r.shift;
    uint64_t *si = sumA.p, *so = sumB.p; uint32_t *ni = nextA.p, *no = nextB.p;
    for (uint64_t span = 1; span < nElem; span <<= 1) {
        hipLaunchKernelGGL(k_restore_rank, ge, bl, 0, 0, si, ni, so, no, nElem);
        std::swap(si, so); std::swap(ni, no);
    }
    uint64_t total = 0; uint32_t e = 0;
    HIP_OK(hipMemcpy(&total, si + startSeg, 8, hipMemcpyDeviceToHost));
    HIP_OK(hipMemcpy(&e, err.p, 4, hipMemcpyDeviceToHost));
    if (e || total != n) throw std::runtime_error("cf_index_restore: the BWT does not invert to one text of the stated length (damaged index)");
    HIP_OK(hipMemcpy(&ix.restoreMaxSeg, cur.p + 2, 4, hipMemcpyDeviceToHost));
    r.segEnd = si;
    r.saPos = saPos; r.isa = isa; r.posShift = posShift; r.isaShift = isaShift;
    HIP_OK(hipMemsetAsync(cur.p, 0, 16, 0));
    HIP_OK(hipEventRecord(ev[2], 0));
    hipLaunchKernelGGL((k_restore<2, true>), gr, bl, 0, 0, ix.d, r);
    HIP_OK(hipEventRecord(ev[3], 0));
    HIP_OK(hipDeviceSynchronize());
    HIP_OK(hipGetLastError());
    if (verbose) {                                                             // each pass touches one 128-byte side per character
        float p1 = 0, rk = 0, p2 = 0;
        HIP_OK(hipEventElapsedTime(&p1, ev[0], ev[1])); HIP_OK(hipEventElapsedTime(&rk, ev[1], ev[2])); HIP_OK(hipEventElapsedTime(&p2, ev[2], ev[3]));
        std::fprintf(stderr, "cf_index_restore: n=%llu marks every %u rows, %u segments; pass1 %.1f ms (%.2f TB/s), ranking %.1f ms, pass2 %.1f ms (%.2f TB/s)\n",
                     (unsigned long long)n, 1u << r.shift, r.nSeg, p1, 128.0 * n / (p1 * 1e-3) / 1e12, rk, p2, 128.0 * n / (p2 * 1e-3) / 1e12);
    }
    HIP_OK(hipMemcpy(&e, err.p, 4, hipMemcpyDeviceToHost));
    if (e) throw std::runtime_error("cf_index_restore: damaged index");
}

// The tables of the search kernel's text verification (DIndex::text / saPos / isa): the inverse-BWT walks above know the text
// position of every row they visit, so the 2-bit text, SA[row] for every 2^rate-th row and the row of every 2^rate-th
// position come out of one run of them.  rate: CF_TEXT_VERIFY_RATE (-1 = off), default 1; raised until the tables
// (16 bytes per sampled row / position + n/4 of text) fit half of the free HBM, given up beyond 5 (every 32nd row: a unique
// match then steps ~32 times to a sampled row and < 32 back from the inverse sample — still a fraction of a 250-base read's
// single-row steps).
void textifyIndex(cf_index &ix) {
    int rate = cfamd::cf_knob("CF_TEXT_VERIFY_RATE") ? envInt("CF_TEXT_VERIFY_RATE", 1) : ix.opt.text_verify_rate < 0 ? -1 : ix.opt.text_verify_rate > 0 ? ix.opt.text_verify_rate : 1;
    if (ix.planned && ix.plannedTextRate == 0 && !cfamd::cf_knob("CF_TEXT_VERIFY_RATE")) rate = 0;
    if (rate < 0 || ix.h.g.len < 64) return;
    const size_t freeB = freeFor(ix);
    const uint64_t n = ix.h.g.len;
    // (small ranges against the text — asked for, and possible only with the samples at every row — keep the inverse sample there too)
    auto multiAt = [&](int r) { return r == 0 && smallRangeRows(ix) >= 2; };
    while (rate <= 5 && textTableBytes(n, ix.h.g.offRate, rate, multiAt(rate)) > (ix.planned ? freeB : freeB / 2)) rate++;
    if (rate > 5) return;
    const int isaRate = isaRateFor(n, ix.h.g.offRate, rate, multiAt(rate));
    const auto t0 = std::chrono::steady_clock::now();
    ix.saPos.alloc(trio_words((n >> rate) + 2)); ix.isa.alloc(trio_words((n >> isaRate) + 2));       // 40-bit values, three to 16 bytes
    HIP_OK(hipMemsetAsync(ix.saPos.p, 0, ix.saPos.bytes(), 0)); HIP_OK(hipMemsetAsync(ix.isa.p, 0, ix.isa.bytes(), 0));
    restoreCore(ix, ix.text, ix.saPos.p, ix.isa.p, (uint32_t)rate, (uint32_t)isaRate);
    ix.d.isaRate = isaRate;
    ix.textMs = (float)(std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() * 1e3);
    ix.d.text = reinterpret_cast<const uint64_t *>(ix.text.p); ix.d.saPos = ix.saPos.p; ix.d.isa = ix.isa.p; ix.d.posRate = rate;
    ix.d.verifyMinRun = (uint32_t)std::max(0, envInt("CF_TEXT_VERIFY_MIN_RUN", 0));
    // small ranges against the text (DIndex::multiRows): off by default — cf_index_options::small_range_rows (or CF_MULTI_VERIFY) asks
    // for it, and it takes effect when the samples are at every row (the planner makes them so when that fits)
    ix.d.multiRows = rate == 0 && smallRangeRows(ix) >= 2 ? smallRangeRows(ix) : 0u;
    ix.d.multiMinRun = (uint32_t)std::max(0, envInt("CF_MULTI_MIN_RUN", 0));
    ix.deviceBytes += ix.text.bytes() + ix.saPos.bytes() + ix.isa.bytes();
}

// cf_index::repeatFrac: 16 K pairs of neighbouring rows, 24 bases back (k_repeat_probe) — a millisecond, over the file's sides
void probeRepeats(cf_index &ix) {
    constexpr uint32_t kSamples = 1u << 14, kDepth = 24;
    DevBuf<uint32_t> hits;
    hits.alloc(1);
    HIP_OK(hipMemset(hits.p, 0, 4));
    hipLaunchKernelGGL(k_repeat_probe, dim3(kSamples / 256), dim3(256), 0, nullptr, ix.d, kSamples, kDepth, hits.p);
    uint32_t h = 0;
    HIP_OK(hipMemcpy(&h, hits.p, 4, hipMemcpyDeviceToHost));
    HIP_OK(hipGetLastError());
    ix.repeatFrac = (double)h / (double)kSamples;
}

bool haveDevice() {
    int n = 0;
    return hipGetDeviceCount(&n) == hipSuccess && n > 0;
}

}  // namespace

// host view of an index for the host-only modules of the library (cf_report.cpp)
const HostIndex &cf_index_host(const cf_index *ix) { return ix->h; }

// ======================================================================= C ABI
extern "C" {

const char *cf_strerror(cf_status s) {
    switch (s) {
        case CF_OK: return "ok";
        case CF_ERR_IO: return "index file I/O error";
        case CF_ERR_FORMAT: return "index format error";
        case CF_ERR_NO_DEVICE: return "no HIP device (this library has no CPU path)";
        case CF_ERR_HIP: return "HIP runtime error";
        case CF_ERR_ARG: return "bad argument";
        case CF_ERR_NOMEM: return "out of memory";
        default: return "unknown status";
    }
}
const char *cf_last_error(void) { return g_err.c_str(); }

cf_status cf_index_open_host(const char *basename, cf_index **out) {
    if (!basename || !out) return CF_ERR_ARG;
    *out = nullptr;
    auto ix = std::make_unique<cf_index>();
    cf_status st = guard([&] { ix->h.load(basename, nullptr); });
    if (st == CF_OK) *out = ix.release();
    return st;
}

cf_status cf_index_open(const char *basename, int device, cf_index **out) { return cf_index_open_ex(basename, device, nullptr, out); }

cf_status cf_index_open_ex(const char *basename, int device, const cf_index_options *opt, cf_index **out) {
    if (!basename || !out) return CF_ERR_ARG;
    *out = nullptr;
    if (!haveDevice()) { g_err = "no HIP device visible"; return CF_ERR_NO_DEVICE; }
    auto ix = std::make_unique<cf_index>();
    if (opt) ix->opt = *opt;
    cf_status st = guard([&] {
        HIP_OK(hipSetDevice(device));
        hipDeviceProp_t prop;
        HIP_OK(hipGetDeviceProperties(&prop, device));
        ix->numCUs = prop.multiProcessorCount;
        ix->device = device;
        {
            size_t freeB = 0, totalB = 0;
            HIP_OK(hipMemGetInfo(&freeB, &totalB));
            ix->budgetSeen = ix->opt.hbm_budget_bytes ? std::min<uint64_t>(ix->opt.hbm_budget_bytes, freeB) : freeB;
        }
        uploadIndex(*ix, basename);
        ix->fileBytes = ix->deviceBytes;
        if (ix->opt.hbm_budget_bytes && ix->fileBytes > ix->opt.hbm_budget_bytes)
            throw ArgError("the index files alone need more device memory than hbm_budget_bytes allows");
        ix->d.posRate = -1;
        if (ix->opt.small_range_rows == 0 && !cfamd::cf_knob("CF_MULTI_VERIFY") && ix->h.g.len >= (1u << 16)) probeRepeats(*ix);
        if (envInt("CF_TABLE_PLANNER", 1)) {          // (CF_TABLE_PLANNER=0: the fixed priorities and shares of rounds 2 - 3 instead)
            // what the tables may take: the budget (or the device's free memory) less the files' sections, and — without a budget —
            // a reserve for the batch slots (a fifth of the device, at least 48 GB — three slots of 10 M mates of 150 bases take 35 GB —
            // but at most half of what is free)
            size_t freeB = 0, totalB = 0;
            HIP_OK(hipMemGetInfo(&freeB, &totalB));
            uint64_t room;
            if (ix->opt.hbm_budget_bytes) room = std::min<uint64_t>(ix->opt.hbm_budget_bytes - ix->fileBytes, freeB);
            else {
                // (never more than half of what is free: a 16 - 48 GB card, or a device another process shares, still gets tables)
                const uint64_t reserve = std::min<uint64_t>(std::max<uint64_t>(48ull << 30, totalB / 5), freeB / 2);
                room = freeB - reserve;
            }
            const TablePlan tp = planTables(*ix, room);
            ix->opt.wide_ftab_chars = tp.K ? tp.K : -1;
            ix->opt.text_verify_rate = tp.textRate < 0 ? -1 : tp.textRate;
            ix->plannedTextRate = tp.textRate;                       // (0 = every row: the option field cannot say that, it means "automatic")
            ix->opt.occ_planes = tp.planes ? 1 : -1;
            ix->opt.resolve_rate = tp.resolveRate >= ix->h.g.offRate ? -1 : tp.resolveRate + 1;
            ix->opt.pair_planes = tp.pair ? 1 : -1;
            ix->planned = true;
            ix->planDropSides = tp.dropSides != 0;
            ix->plan = TablePlanRecord{tp.K, tp.textRate, tp.planes, tp.resolveRate, tp.pair, tp.dropSides, true};
        }
        if (ix->planned) {
            // the planes first (the wide ftab is then made over them: one load per step instead of eight), the tables that read the
            // sides next, and — where the plan counts on their room — the sides out of HBM before the last tables are allocated
            planifyIndex(*ix);
            textifyIndex(*ix);
            densifyIndex(*ix);
            if (ix->planDropSides) dropSides(*ix);
            widenFtab(*ix);
            pairPlanifyIndex(*ix);
            makePosTables(*ix);
        } else {
            widenFtab(*ix);
            textifyIndex(*ix);
            planifyIndex(*ix);
            densifyIndex(*ix);
            pairPlanifyIndex(*ix);
            makePosTables(*ix);
        }
        queryOccupancy(*ix);
    });
    if (st == CF_OK) *out = ix.release();
    return st;
}

void cf_index_close(cf_index *ix) { delete ix; }

// the table planner on its own (no device): what cf_index_open would make of an index of n bases under `room` bytes for the tables
cf_status cf_debug_plan_tables(uint64_t n, int ftab_chars, int off_rate, int sa_width, uint64_t room, const cf_index_options *opt,
                               int32_t out[6], double *cost, uint64_t *bytes) {
    if (!out) return CF_ERR_ARG;
    cf_index ix;
    ix.h.g.len = n; ix.h.g.ftabChars = ftab_chars; ix.h.g.offRate = off_rate; ix.h.offw = sa_width == 4;
    ix.h.g.numSides = ((n / 4 + 1) + 95) / 96;
    if (opt) ix.opt = *opt;
    const TablePlan tp = planTables(ix, room);
    out[0] = tp.K; out[1] = tp.textRate; out[2] = tp.planes; out[3] = tp.resolveRate; out[4] = tp.pair; out[5] = tp.dropSides;
    if (cost) *cost = tp.cost;
    if (bytes) *bytes = tp.bytes;
    return CF_OK;
}

cf_status cf_index_describe(const cf_index *ix, cf_index_config *c) {
    if (!ix || !c) return CF_ERR_ARG;
    std::memset(c, 0, sizeof *c);
    const uint64_t n = ix->h.g.len;
    c->text_len = n; c->budget_bytes = ix->budgetSeen; c->file_section_bytes = ix->fileBytes;
    c->wide_ftab_bytes = ix->wide.bytes(); c->wide_ftab_chars = ix->d.wideChars;
    c->text_bytes = ix->text.bytes() + ix->saPos.bytes() + ix->isa.bytes() + ix->posBucket.bytes() + ix->posFrag.bytes() + ix->posSeq.bytes();      // (+ position -> reference, makePosTables)
    c->text_verify_rate = ix->device >= 0 ? ix->d.posRate : -1;
    c->planes_bytes = ix->planes.bytes(); c->occ_planes = ix->d.planes ? 1 : 0;
    c->pair_planes_bytes = ix->planes2.bytes(); c->pair_planes = ix->d.planes2 ? 1 : 0;
    c->resolve_bytes = ix->dense.bytes(); c->resolve_rate = ix->denseRate >= 0 ? ix->denseRate : ix->h.g.offRate;
    c->sides_dropped = ix->sidesDropped ? 1 : 0;
    c->file_bytes_dropped = ix->droppedBytes;
    c->small_range_rows = ix->device >= 0 ? (int32_t)ix->d.multiRows : 0;
    c->repeat_fraction = ix->repeatFrac;
    // 1: every table came out as the planner chose it (0: one did not fit when its turn came and was made coarser or not at all;
    // -1: no plan — CF_TABLE_PLANNER=0 or a host-only view)
    c->plan_realised = !ix->plan.valid ? -1 :
        (ix->d.wideChars == ix->plan.K && (ix->device >= 0 ? ix->d.posRate : -1) == ix->plan.textRate && (ix->d.planes != nullptr) == (ix->plan.planes != 0) &&
         (ix->denseRate >= 0 ? ix->denseRate : ix->h.g.offRate) == ix->plan.resolveRate && (ix->d.planes2 != nullptr) == (ix->plan.pair != 0) &&
         ix->sidesDropped == (ix->plan.dropSides != 0)) ? 1 : 0;
    c->total_bytes = ix->deviceBytes;
    c->build_ms = ix->planesMs + ix->planes2Ms + ix->wideMs + ix->textMs + ix->denseMs;
    // the request model of DESIGN.md 5 (constants measured on the config-2 workload: 6.5 partialSearch calls and 1.4 resolved rows
    // per 100-base read): two-row steps until a call's range is one row, single-row steps / verification reads, one table
    // lookup per call, two strand records, the walk
    const double log4n = n > 1 ? std::log((double)n) / std::log(4.0) : 0.0;
    const int K = c->wide_ftab_chars ? c->wide_ftab_chars : ix->h.g.ftabChars;
    const double calls = 6.5, rows = 1.42;
    const double twoRow = calls * (std::max(0.0, log4n - K) + 1.4);
    const double r = c->text_verify_rate;
    const double single = r < 0 ? 67.6 : 1.5 * (1.0 + ((double)(1u << (int)r) - 1.0)) + 4.0, verify = r < 0 ? 0.0 : 5.0;
    const double walk = rows * (double)(1u << c->resolve_rate);
    c->est_requests_per_100bp_read = twoRow + single + verify + calls + 2.0 + walk;
    return CF_OK;
}

uint64_t cf_index_text_len(const cf_index *ix) { return ix->h.g.len; }
uint64_t cf_index_num_refs(const cf_index *ix) { return ix->h.uid.size(); }
uint64_t cf_index_num_taxa(const cf_index *ix) { return ix->h.taxa.size(); }
uint64_t cf_index_device_bytes(const cf_index *ix) { return ix->deviceBytes; }
int cf_index_compressed(const cf_index *ix) { return ix->h.compressed ? 1 : 0; }
int cf_index_sa_width(const cf_index *ix) { return ix->h.offw ? 4 : 2; }
int cf_index_wide_ftab_chars(const cf_index *ix) { return ix->d.wideChars; }
int cf_index_occ_planes(const cf_index *ix) { return ix->d.planes ? 1 : 0; }
double cf_index_occ_planes_build_ms(const cf_index *ix) { return ix->planesMs; }
int cf_index_text_verify_rate(const cf_index *ix) { return ix->device >= 0 ? ix->d.posRate : -1; }
double cf_index_text_verify_build_ms(const cf_index *ix) { return ix->textMs; }
int cf_index_resolve_rate(const cf_index *ix) { return ix->denseRate >= 0 ? ix->denseRate : ix->h.g.offRate; }
double cf_index_resolve_build_ms(const cf_index *ix) { return ix->denseMs; }
uint32_t cf_index_walk_bound(const cf_index *ix) { return ix->d.posFrag ? ix->d.walkMax : 0u; }
int cf_index_resolve_by_position(const cf_index *ix) { return ix->denseByPos ? 1 : 0; }
const char *cf_index_uid(const cf_index *ix, uint64_t r) { return r < ix->h.uid.size() ? ix->h.uid[r].c_str() : ""; }
uint64_t cf_index_ref_taxid(const cf_index *ix, uint64_t r) { return r < ix->h.uidTid.size() ? ix->h.uidTid[r] : 0; }
uint64_t cf_index_taxon_id(const cf_index *ix, uint64_t i) { return i < ix->h.taxa.size() ? ix->h.taxa[i] : 0; }
const char *cf_format_seqid(const cf_index *ix, uint32_t u, uint64_t t) { return ix->h.formatSeqId(u, t); }
int cf_tax_rank(const cf_index *ix, uint64_t t) { const TaxNode *n = ix->h.findNode(t); return n ? n->rank : 0; }
const char *cf_tax_rank_string(int rank) { return rankString(rank); }
const char *cf_tax_name(const cf_index *ix, uint64_t t) { return ix->h.name(t); }
uint64_t cf_tax_size(const cf_index *ix, uint64_t t) { return ix->h.size(t); }

cf_status cf_params_default(cf_params *p) {
    if (!p) return CF_ERR_ARG;
    std::memset(p, 0, sizeof *p);
    p->khits = 5; p->min_hitlen = 22; p->rank_slot = 0; p->tree_traverse = 1;
    return CF_OK;
}

cf_status cf_classifier_create(cf_index *ix, const cf_params *p, cf_classifier **out) {
    if (!ix || !p || !out || p->khits < 1 || p->min_hitlen < 15 || p->rank_slot < 0 || p->rank_slot > 9) return CF_ERR_ARG;
    *out = nullptr;
    if (ix->device < 0) { g_err = "index was opened host-only"; return CF_ERR_NO_DEVICE; }
    auto cl = std::make_unique<cf_classifier>();
    cf_status st = guard([&] {
        HIP_OK(hipSetDevice(ix->device));
        cl->ix = ix; cl->p = *p;
        cl->hostList.assign(p->host_taxids, p->host_taxids + std::max(0, p->n_host));
        cl->exclList.assign(p->exclude_taxids, p->exclude_taxids + std::max(0, p->n_exclude));
        cl->p.host_taxids = cl->hostList.data(); cl->p.exclude_taxids = cl->exclList.data();
        const ClassifierTables t = makeClassifier(ix->h, cl->p, cl->d);
        if (cl->d.ihits >= 0x7fffffffu) throw ArgError("-k too large");
        if (ix->h.g.len >= (1ull << 40)) throw ArgError("index too large: hit records hold 40-bit rows");
        if (!t.refExcluded.empty()) { cl->refExcluded.upload(t.refExcluded); cl->d.refExcluded = cl->refExcluded.p; }
        if (!t.hostSet.empty()) { cl->hostSet.upload(t.hostSet); cl->d.hostSet = cl->hostSet.p; cl->d.nHostSet = (uint32_t)t.hostSet.size(); }
        cl->counts.alloc(3 * ix->h.taxa.size());
        HIP_OK(hipMemset(cl->counts.p, 0, cl->counts.bytes()));
    });
    if (st == CF_OK) *out = cl.release();
    return st;
}
void cf_classifier_destroy(cf_classifier *c) { delete c; }

uint32_t cf_gen_rand_seed(const uint8_t *seq, const uint8_t *qual, uint64_t len, const char *name, uint64_t nlen,
                          uint32_t seed) {
    // pat.h:55-91; shifts of an `int` beyond the value bits wrap in 32-bit arithmetic
    uint32_t r = (seed + 101u) * 59u * 61u * 67u * 71u * 73u * 79u * 83u;
    for (uint64_t i = 0; i < len; i++) r ^= (uint32_t)seq[i] << ((i & 15) << 1);
    for (uint64_t i = 0; i < len; i++) r ^= (uint32_t)(qual ? qual[i] : (uint8_t)'I') << ((i & 3) << 3);
    for (uint64_t i = 0; i < nlen; i++) {
        const int p = (int)(signed char)name[i];
        if (p == '/') break;
        r ^= (uint32_t)p << ((i & 3) << 3);
    }
    return r;
}

// ------------------------------------------------------------------ a batch, stage by stage
// Everything below enqueues on one stream and returns; only waitBatch() blocks.

// Sizes every buffer of the slot for a batch of nReads reads in nWords packed words (maxLen = longest read,
// nBases = sum of the lengths or 0 when unknown).  Buffers only grow; nothing here touches a stream.
static void sizeBatch(cf_batch *bt, uint64_t nReads, uint64_t nWords, uint64_t nBases, uint32_t maxLen, int paired) {
    cf_classifier *cl = bt->cl;
    if (paired && (nReads & 1)) throw ArgError("a paired batch needs an even number of reads");
    if (nReads >= 0x7fffffffull) throw ArgError("a batch holds fewer than 2^31 reads");
    if (maxLen > kMaxReadLen) throw ArgError("reads of more than 16,777,213 bases are not supported (24-bit offsets in the hit records)");
    const uint32_t ftc = (uint32_t)std::max(1, cl->ix->h.g.ftabChars);
    bt->nReads = nReads; bt->paired = paired ? 1 : 0; bt->nQueries = paired ? nReads / 2 : nReads; bt->nWords = nWords;
    bt->maxLenHost = maxLen;
    const uint64_t nq = bt->nQueries;
    bt->bases.ensure(nWords + 16); bt->nmask.ensure(nWords + 16);      // (the search kernel reads a read's words in whole 16-byte pieces, W words from its first)
    bt->rlen.ensure(nReads + 16); bt->seeds.ensure(nReads + 1); bt->woff.ensure(nReads + 1);
    bt->pass.ensure(nReads + 1); bt->hitCap.ensure(nReads + 16);
    bt->slotOf.ensure(nReads + 1); bt->hitBase.ensure(nReads + 1); bt->items.ensure(nReads + 1);
    bt->nhml.ensure(2 * nReads + 1);
    bt->maxScore.ensure(nq + 1); bt->qflag.ensure(nq + 1); bt->qhead.ensure(nq + 1); bt->qRows.ensure(nq + 16); bt->qBase.ensure(nq + 1);
    bt->qplan.ensure((nq + 1) * kInlinePlan); bt->o1tax.ensure((nq + 1) * kFieldRows); bt->o1a.ensure((nq + 1) * kFieldRows); bt->o1b.ensure((nq + 1) * kFieldRows);
    bt->slowPost.ensure(nq + 1); bt->slowScore.ensure(nq + 1);
    if (envInt("CF_EARLY_SCORE", 0)) bt->postDeferred.ensure(nq + 16);      // (only the early-score experiment reads it: off by default, and then neither allocated nor written)
    bt->out.ensure(nq * (uint64_t)cl->d.k + 1); bt->nOut.ensure(nq + 16); bt->score2.ensure(nq + 1); bt->rowFirst.ensure(nq + 1);
    bt->cursor.ensure(4); bt->ops.ensure(1); bt->st.ensure(1);
    bt->tileA.ensure(scan_tiles_for(std::max(nReads, nq)) + 1); bt->tileC.ensure(scan_tiles_for(std::max(nReads, nq)) + 1);
    // strand records of k_search2 (2-bit search-order words + N masks) when every read fits them.  k_search2 keeps a
    // strand's hit count in 8 bits: hits per strand <= #N + (L - #N) / ftabChars + 2 with #N <= 0.15 L for a
    // classified read (only an index with a very short ftab can get near that)
    bt->recWords = searchVersion() != 2 ? 0u : maxLen <= 128 ? 4u : maxLen <= 192 ? 6u : maxLen <= 256 ? 8u : 0u;
    if ((uint64_t)(0.15 * maxLen) + maxLen / (uint64_t)ftc + 3 >= 255) bt->recWords = 0;
    // the one-lane kernel (over the planes) makes its strand records itself from the packed reads and 16 bytes per work item
    // (DBatch::itemMeta); the others read the records k_pack writes
    bt->selfRecords = bt->recWords && cl->ix->d.planes && nWords < 0xffffffffull && envInt("CF_SELF_RECORDS", 1);
    if (bt->selfRecords) bt->itemMeta.ensure(8 * nReads + 8);
    // ... and finds the forward strands' words in search order behind the packed reads, in the same array (DPlan::revDelta; word
    // offsets stay 32 bits)
    bt->revDelta = bt->selfRecords && 2 * nWords + 64 < 0xffffffffull && envInt("CF_REV_WORDS", 1) ? (uint32_t)nWords + 16u : 0u;
    if (bt->revDelta) bt->bases.ensure(2 * (nWords + 16));      // (sizeBatch runs before the reads are copied in)
    else if (bt->recWords) bt->recs.ensure(2 * nReads * (uint64_t)rec_bytes((int)bt->recWords) + 64);
    // hit pool: a strand's list holds #N + (L - #N)/ftc + 2 hits.  Sized for N-free reads plus 2 % (+ 4096); a batch
    // rich in N asks for more through BatchStatus::hitsNeed and is re-run once with a pool of that size.
    uint64_t hitsWant = 2 * ((nBases ? nBases : 32 * nWords) / ftc + 2 * nReads);
    hitsWant += hitsWant / 50 + 4096;
    if (bt->hitsCapLimit) hitsWant = std::min(hitsWant, bt->hitsCapLimit);
    if (bt->hits.n < hitsWant) bt->hits.ensure(hitsWant);
    if (bt->recWords && bt->hits.n > 0xffffffffull) throw ArgError("batch too large: its hit lists need 32-bit offsets, split it");
    // row workspace: rows per pass.  ~100 bytes per row; 4 rows per query cover ordinary reads (1-2 rows) and repeat-rich
    // collections (3.3 measured) — 8 were 9 GB of a 10 M-read slot's 21 —, a batch that plans more is finished in further passes
    // (waitBatch), and the slot then remembers what its batches plan and sizes the workspace for that
    uint64_t rowsWant = std::max<uint64_t>(4 * nq, 1u << 16);
    if (bt->plannedPerQuery > 0) rowsWant = std::max<uint64_t>(rowsWant, (uint64_t)((double)nq * bt->plannedPerQuery * 1.15));
    if (const int per = envInt("CF_ROWS_PER_QUERY", 0)) rowsWant = std::max<uint64_t>((uint64_t)per * nq, 1024);
    if (bt->rowsCapLimit) rowsWant = bt->rowsCapLimit;
    if (bt->rowVal.n < rowsWant || bt->rowsCapLimit) { bt->rowVal.ensure(rowsWant); bt->rowRef.ensure(rowsWant); bt->hm.ensure(rowsWant); bt->tc.ensure(rowsWant); }
    bt->outCompact.ensure(nq * (uint64_t)cl->d.k + 1);
    // pinned results
    bt->hSt.ensure(1); bt->hOps.ensure(1);
    bt->hNOut.ensure(nq + 1); bt->hScore2.ensure(nq + 1); bt->hMaxScore.ensure(nq + 1);
    if (bt->resultFormat == CF_RESULTS_NARROW) { bt->qinfo.ensure(nq + 16); bt->hQInfo.ensure(nq + 16); }
    // rows copied back before their number is known: a quarter more than one per query, or a tenth more than the slot's last
    // batch printed (a slot that met reads with several assignments each keeps the larger pinned buffer and asks for more)
    // a slot that has seen a batch asks for what that one printed per query and 4 % more (the rows beyond it, if any, are fetched by
    // cf_batch_wait: a few per cent of a batch, into the same pinned buffer) — the quarter of margin of a first batch is 60 MB of
    // a 10 M-read batch's 360 MB across PCIe, which the kernels now outrun
    bt->rowsSpec = bt->rowsPerQuery > 0 ? (uint64_t)((double)nq * bt->rowsPerQuery * 1.04) + 4096 : nq + nq / 4 + 1024;
    bt->rowsSpec = std::min<uint64_t>(bt->rowsSpec, nq * (uint64_t)cl->d.k);
    bt->hRows.ensure(std::max<uint64_t>(bt->rowsSpec, nq + nq / 4 + 1024));
    if (g_dryBytes) return;
    if (!bt->evInit) { for (auto &e : bt->ev) HIP_OK(hipEventCreate(&e)); HIP_OK(hipEventCreate(&bt->evLate)); HIP_OK(hipEventCreate(&bt->evPostFast)); HIP_OK(hipEventCreate(&bt->evPost)); bt->evInit = true; }
    // the slot's own stream: the general post kernel runs on it beside the early score kernel (enqueueClassify; CF_EARLY_SCORE=0: not),
    // and the CF_TAIL_STREAM experiments
    if (!bt->tail && (envInt("CF_TAIL_STREAM", 0) || envInt("CF_EARLY_SCORE", 0))) HIP_OK(hipStreamCreateWithFlags(&bt->tail, hipStreamNonBlocking));
}

// device views of the slot's buffers (after any growth)
static void bindBatch(cf_batch *bt) {
    cf_classifier *cl = bt->cl;
    DPlan &pl = bt->pl;
    pl.nmask = bt->nmask.p; pl.rlen = bt->rlen.p; pl.woff = bt->woff.p; pl.nReads = (uint32_t)bt->nReads; pl.ftabChars = cl->ix->h.g.ftabChars;
    pl.maxLenAllowed = bt->maxLenHost;
    pl.nWords = bt->nWords;
    pl.pass = bt->pass.p; pl.hitCap = bt->hitCap.p; pl.slotOf = bt->slotOf.p;
    pl.hitBase = bt->hitBase.p; pl.items = bt->items.p; pl.st = bt->st.p;
    pl.itemMeta = bt->selfRecords ? bt->itemMeta.p : nullptr;
    pl.bases = bt->bases.p; pl.revDelta = bt->revDelta;
    pl.hitsCap = bt->hitsCapLimit ? std::min<uint64_t>(bt->hitsCapLimit, bt->hits.n) : bt->hits.n;
    DBatch &d = bt->d;
    d.bases = bt->bases.p; d.nmask = bt->nmask.p; d.rlen = bt->rlen.p; d.woff = bt->woff.p; d.seeds = bt->seeds.p;
    d.pass = bt->pass.p; d.items = bt->items.p; d.slotOf = bt->slotOf.p; d.hitBase = bt->hitBase.p; d.hitCap = bt->hitCap.p;
    // (bit 2: one-row hits that end in the text go out in the position form — where the index can resolve a position, DIndex::posFrag)
    d.lazyHits = (uint32_t)(envInt("CF_LAZY_HITS", 1) != 0) | (cl->ix->d.posFrag ? 4u : 0u);
    // the resolve table at every row: the common-case score kernel reads references straight from it (no emit, no walk)
    d.directRefs = (uint32_t)(cl->ix->dense.p && cl->ix->d.walkRate == 0 && envInt("CF_SCORE_FAST", 1) != 0 && envInt("CF_DIRECT_REFS", 1) != 0);
    d.hits = bt->hits.p; d.nhml = bt->nhml.p; d.qflag = bt->qflag.p; d.qhead = bt->qhead.p; d.qplan = bt->qplan.p; d.qplanStride = bt->qplan.n / kInlinePlan; d.qRows = bt->qRows.p; d.qBase = bt->qBase.p;
    d.rowVal = bt->rowVal.p; d.rowRef = bt->rowRef.p; d.hm = bt->hm.p; d.tc = bt->tc.p;
    d.out = bt->out.p; d.nOut = bt->nOut.p; d.score2 = bt->score2.p;
    d.counts = cl->counts.p; d.nTaxa = (uint32_t)cl->ix->h.taxa.size();
    d.nReads = (uint32_t)bt->nReads; d.nQueries = (uint32_t)bt->nQueries; d.paired = bt->paired;
    d.cursor = bt->cursor.p; d.st = bt->st.p; d.ops = bt->ops.p;
    d.slowPost = bt->slowPost.p; d.slowScore = bt->slowScore.p; d.postDeferred = envInt("CF_EARLY_SCORE", 0) ? bt->postDeferred.p : nullptr;
    d.o1tax = bt->o1tax.p; d.o1a = bt->o1a.p; d.o1b = bt->o1b.p; d.oStride = bt->o1tax.n / kFieldRows;
    d.hitsCap = pl.hitsCap;
    d.rowsCap = bt->rowsCapLimit ? std::min<uint64_t>(bt->rowsCapLimit, bt->rowVal.n) : bt->rowVal.n;
    d.recs = bt->recWords && !bt->selfRecords ? bt->recs.p : nullptr; d.recWords = bt->recWords;
    d.itemMeta = bt->selfRecords ? bt->itemMeta.p : nullptr;
}

// The batch plan, all on the device: filters and hit capacities (k_plan), work list and hit-list bases (two exclusive
// scans + k_plan_fill, which also leaves the work-list and hit-pool sizes in BatchStatus), max_score per query, strand records.
static void enqueuePlan(cf_batch *bt, hipStream_t st) {
    const uint64_t nReads = bt->nReads;
    const DPlan &pl = bt->pl;
    HIP_OK(hipStreamWaitEvent(st, bt->ev[8], 0));      // the upload may have gone through another (copy) stream
    if (bt->densePending) {                            // a dense upload: its bytes into the word form (once: a re-plan of resident reads finds them made)
        const uint32_t L = bt->densePending - 1;
        const DUnpack u{bt->dense.p, bt->bases.p, bt->rlen.p, (uint32_t)nReads, L, bt->revDelta ? bt->bases.p + bt->revDelta : nullptr};
        bt->revMade = bt->revDelta != 0;
        const uint64_t threads = nReads * std::max<uint64_t>(((uint64_t)L + 31) >> 5, 1);
        hipLaunchKernelGGL(k_dense_unpack, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, st, u);
        bt->densePending = 0;
    }
    HIP_OK(hipEventRecord(bt->ev[5], st));
    HIP_OK(hipMemsetAsync(bt->st.p, 0, sizeof(BatchStatus), st));
    // word offsets of the reads = exclusive sums of ceil(len / 32); byte input gets its lengths and is packed here
    const dim3 gp((unsigned)((nReads + 1 + 255) / 256)), bl(256);
    if (bt->fromBytes && nReads) hipLaunchKernelGGL(k_rlen, gp, bl, 0, st, bt->off8.p, bt->rlen.p, (uint32_t)nReads);
    scan_enqueue<SCAN_WORDS>(bt->rlen.p, nReads, bt->woff.p, nullptr, bt->tileA.p, bt->tileC.p, st);
    if (bt->fromBytes && nReads) {
        DConvert c{bt->seq.p, bt->off8.p, bt->woff.p, bt->bases.p, bt->nmask.p, (uint32_t)nReads};
        hipLaunchKernelGGL(k_convert, dim3((unsigned)((nReads + 255) / 256)), dim3(256), 0, st, c);
    }
    if (bt->fromText && nReads) {                      // a text block: the record pass left lengths and places, the words are made here
        const DTextPack c{bt->text.p, bt->txSeqOff.p, bt->rlen.p, bt->woff.p, bt->bases.p, bt->nmask.p, (uint32_t)nReads};
        hipLaunchKernelGGL(k_text_pack, dim3((unsigned)((nReads + 255) / 256)), dim3(256), 0, st, c);
    }
    hipLaunchKernelGGL(k_plan, gp, bl, 0, st, pl);
    scan_enqueue<SCAN_HITS>(bt->hitCap.p, nReads, bt->hitBase.p, bt->slotOf.p, bt->tileA.p, bt->tileC.p, st);      // hit-list bases + work-list slots in one scan
    hipLaunchKernelGGL(k_plan_fill, gp, bl, 0, st, pl);
    if (pl.revDelta && nReads && !bt->revMade) {
        const uint64_t threads = nReads * (uint64_t)bt->recWords;
        hipLaunchKernelGGL(k_rev_words, dim3((unsigned)((threads + 255) / 256)), bl, 0, st, pl, bt->recWords);
    }
    if (bt->nQueries) hipLaunchKernelGGL(k_plan_maxscore, dim3((unsigned)((bt->nQueries + 255) / 256)), bl, 0, st, bt->rlen.p, bt->pass.p,
                                         (uint32_t)bt->nQueries, bt->paired, bt->maxScore.p);
    if (bt->recWords && !bt->selfRecords && nReads) {
        const uint64_t threads = 2 * nReads * (uint64_t)bt->recWords;
        hipLaunchKernelGGL(k_pack, dim3((unsigned)((threads + 255) / 256)), bl, 0, st, bt->d, bt->recs.p, bt->recWords);
    }
    HIP_OK(hipEventRecord(bt->ev[6], st));
    bt->planned = true;
}

// grid of the general per-query kernels: they stride over a list whose length is on the device
static dim3 listGrid(const cf_index &ix, uint64_t nq) { return dim3((unsigned)std::max<uint64_t>(1, std::min<uint64_t>((nq + 63) / 64, (uint64_t)ix.numCUs * 32))); }

// Round 5, built, validated (emulator order, the whole GPU suite) and MEASURED A LOSS — off unless CF_EARLY_SCORE=1: the EARLY score
// kernel.  With the resolve table at every row (directRefs) the common-case score kernel needs nothing but what the common-case
// post kernel leaves with a query — so it can run right behind it, on the batch's stream, while the general post kernel (the
// queries with a long hit on both strands: chains of a few thousand dependent loads, 0.3 - 0.9 ms of latency for 0.2 - 0.8 % of
// the queries) works BESIDE it on the slot's own stream; the two meet before the rows are counted.  Measured (profiles/r05e_*, same
// box): config 2 post + score 1.73 -> 2.51 ms per 10 M reads, the repeat-rich text 4.13 -> 6.19 — a kernel that lives on the latency
// of dependent loads runs several times longer beside one that keeps the memory system busy than alone on an idle one, and the
// batch waits for it at the join.  (The same lesson as CF_TAIL_STREAM in rounds 3 and 4, from the other side.)
static bool earlyScoreMode(const cf_batch *bt) {
    static const bool on = envInt("CF_EARLY_SCORE", 0) != 0 && envInt("CF_POST_FAST", 1) != 0 && envInt("CF_SCORE_FAST", 1) != 0 && envInt("CF_TAIL_STREAM", 0) == 0;
    return on && bt->d.directRefs != 0 && bt->tail != nullptr && bt->nQueries != 0;
}

// extend / trim / strand choice / sort / row plan of every query: the common-case kernel, then the general one over what it left
static void enqueuePost(cf_batch *bt, hipStream_t st) {
    cf_classifier *cl = bt->cl;
    cf_index &ix = *cl->ix;
    const DBatch &d = bt->d;
    const uint32_t nq = (uint32_t)bt->nQueries;
    if (!nq) return;
    static const bool fast = envInt("CF_POST_FAST", 1) != 0;
    if (fast) hipLaunchKernelGGL(k_post_fast, dim3((nq + 255) / 256), dim3(256), 0, st, ix.d, cl->d, d);
    else hipLaunchKernelGGL(k_list_all, dim3((nq + 255) / 256), dim3(256), 0, st, bt->slowPost.p, &bt->st.p->nSlowPost, nq);
    if (earlyScoreMode(bt)) {
        HIP_OK(hipEventRecord(bt->evPostFast, st));
        HIP_OK(hipStreamWaitEvent(bt->tail, bt->evPostFast, 0));
        hipLaunchKernelGGL(k_post, listGrid(ix, nq), dim3(64), 0, bt->tail, ix.d, cl->d, d, 0u);
        HIP_OK(hipEventRecord(bt->evPost, bt->tail));
        hipLaunchKernelGGL(k_score_fast<true>, dim3((nq + 255) / 256), dim3(256), 0, st, ix.d, cl->d, d);
        HIP_OK(hipStreamWaitEvent(st, bt->evPost, 0));
        return;
    }
    // the lane's scratch (post_body): both strands' lists of a mate — #N + (L - #N) / ftabChars + 2 hits a strand, N-free reads in mind
    // (a mate richer in N than that works in place) — 24 records = 24 KB of LDS per wavefront for 100-base reads, 54 KB for 250.
    // Built, validated (emulator: both paths in one batch; the whole GPU suite) and MEASURED A LOSS (round 6, profiles/r06d_*): off
    // unless CF_POST_LDS=1.  What the kernel waits for is ps_whole's chain of HBM misses, not its passes over the hit lists (those hit
    // L2), and the LDS the scratch takes leaves fewer wavefronts per CU to wait side by side: config 2 post 0.80 -> 0.80 ms, config 4
    // 1.40 -> 1.56, config 5 (250-base reads: 54 KB per wavefront, two of them per CU) 1.85 -> 3.65.
    static const bool postLds = envInt("CF_POST_LDS", 0) != 0;
    const uint32_t capHits = postLds ? (uint32_t)std::min<uint64_t>(60, 2 * ((uint64_t)bt->maxLenHost / (uint64_t)std::max(1, ix.h.g.ftabChars) + 2)) : 0u;
    hipLaunchKernelGGL(k_post, listGrid(ix, nq), dim3(64), (size_t)64 * capHits * sizeof(HitP), st, ix.d, cl->d, d, capHits);
}

// one pass of the row stage over the queries from qLo on: window -> emit -> walk -> score.  early: the common-case score kernel has
// run already (enqueuePost) and left its list of queries for the general one
static bool enqueueRowPass(cf_batch *bt, uint32_t qLo, hipStream_t st, bool marks, hipStream_t late = nullptr, bool early = false) {
    cf_classifier *cl = bt->cl;
    cf_index &ix = *cl->ix;
    const DBatch &d = bt->d;
    const uint32_t nq = (uint32_t)bt->nQueries;
    static const bool fast = envInt("CF_SCORE_FAST", 1) != 0;
    HIP_OK(hipMemsetAsync(bt->cursor.p + 1, 0, 8, st));
    hipLaunchKernelGGL(k_window, dim3(1), dim3(64), 0, st, d, qLo, early);
    const bool direct = d.directRefs != 0;                  // (then the stage marks of "post" and "walk" end here: nothing is emitted or walked)
    if (nq && !direct) hipLaunchKernelGGL(k_emit, dim3((nq + 255) / 256), dim3(256), 0, st, cl->d, d);
    if (marks) HIP_OK(hipEventRecord(bt->ev[2], st));
    const bool counted = nq && !direct ? launchWalk(cl, bt, st) : true;
    if (marks) HIP_OK(hipEventRecord(bt->ev[3], st));
    if (nq) {
        if (early) {}                                       // (k_score_fast<true> ran behind k_post_fast)
        else if (fast) hipLaunchKernelGGL(k_score_fast<false>, dim3((nq + 255) / 256), dim3(256), 0, st, ix.d, cl->d, d);
        else hipLaunchKernelGGL(k_list_all, dim3((nq + 255) / 256), dim3(256), 0, st, bt->slowScore.p, &bt->st.p->nSlowScore, nq);   // (score_body skips what lies outside the window)
        if (late && late != st) {                // the rest of the batch on the slot's own stream (CF_TAIL_STREAM=2, see enqueueClassify)
            HIP_OK(hipEventRecord(bt->evLate, st));
            HIP_OK(hipStreamWaitEvent(late, bt->evLate, 0));
            st = late;
        }
        if (direct) hipLaunchKernelGGL(k_resolve_slow, listGrid(ix, nq), dim3(64), 0, st, ix.d, cl->d, d);
        // the lane's scratch (score_body): hit map, parent counts and references of a query of up to capRows planned rows, every
        // second lane at work: 47 KB of LDS per wavefront = three of them per CU.  As with k_post: measured a loss (repeat-rich
        // preset: score 2.42 -> 2.63 ms) — the kernel's time is its slowest query's, the queries that take long are the ones with more
        // rows than a scratch holds, and the others gain nothing from finishing sooner while fewer lanes are resident.  Off unless
        // CF_SCORE_LDS_ROWS=<n>.
        static const uint32_t capRows = (uint32_t)std::clamp(envInt("CF_SCORE_LDS_ROWS", 0), 0, 64), sparse = (uint32_t)std::clamp(envInt("CF_SCORE_LDS_SPARSE", 2), 1, 64);
        // (round 6 also tried the queries with many rows in a launch of their own, one per wavefront with 128 rows of state in LDS,
        // the others in place: score 2.45 -> 2.76 / 2.87 ms on the repeat-rich preset with the line drawn at 12 / 24 rows, profiles/r06k_*)
        hipLaunchKernelGGL(k_score, listGrid(ix, nq), dim3(64), (size_t)(64 / sparse) * score_scratch_bytes(capRows), st, ix.d, cl->d, d, capRows, capRows ? sparse : 1u, 0u, 0xffffffffu);
        static const uint32_t slotBits = (uint32_t)std::clamp(envInt("CF_COUNT_SLOT_BITS", (int)kCountSlotBits), 1, (int)kCountSlotBits);   // (tests: few slots = probing, overflow)
        hipLaunchKernelGGL(k_count, dim3((nq + kCountChunk - 1) / kCountChunk), dim3(1024), 0, st, d, slotBits, d.nTaxa <= (1u << slotBits));
    }
    if (marks) HIP_OK(hipEventRecord(bt->ev[4], st));
    return counted;
}

// printed rows of all queries back to back (query order): on average 1-2 rows per query travel instead of k slots
static void enqueueCompact(cf_batch *bt, hipStream_t st) {
    const uint32_t nq = (uint32_t)bt->nQueries;
    const dim3 g((nq + 1 + 255) / 256), bl(256);
    scan_enqueue<SCAN_PLAIN>(bt->nOut.p, nq, bt->rowFirst.p, nullptr, bt->tileA.p, bt->tileC.p, st);
    // (outCompact has room for all k slots of every query: the number of printed rows is not known on the host here)
    const bool narrow = bt->resultFormat == CF_RESULTS_NARROW;        // (16-byte rows in the same buffer: it holds 24 bytes per row)
    const DCompact c{bt->out.p, bt->o1tax.p, bt->o1a.p, bt->o1b.p, bt->d.oStride, bt->nOut.p, bt->rowFirst.p, (uint32_t)bt->cl->d.k, nq, bt->outCompact.p, bt->st.p,
                     narrow ? reinterpret_cast<NarrowRow *>(bt->outCompact.p) : nullptr, bt->qinfo.p, bt->pass.p, bt->paired};
    hipLaunchKernelGGL(k_compact, g, bl, 0, st, c);
}

static void enqueueClassify(cf_batch *bt, hipStream_t st) {
    cf_classifier *cl = bt->cl;
    HIP_OK(hipMemsetAsync(bt->cursor.p, 0, 32, st));
    HIP_OK(hipMemsetAsync(bt->ops.p, 0, sizeof(OpCounts), st));
    // (qRows: every query's entry is written by one of the post kernels; nOut: by one of the score kernels — queries outside
    // the first pass's row window, or all of them when the hit pool was too small, get a zero there: until they are scored
    // they print nothing, so the compaction behind the first pass stays inside its buffers.  Only the debug modes that
    // skip the common-case kernels still need the memsets.)
    static const bool allFast = envInt("CF_POST_FAST", 1) != 0 && envInt("CF_SCORE_FAST", 1) != 0;
    if (!allFast) {
        HIP_OK(hipMemsetAsync(bt->qRows.p, 0, 4 * (bt->nQueries + 1), st));
        HIP_OK(hipMemsetAsync(bt->nOut.p, 0, 4 * (bt->nQueries + 1), st));
    }
    HIP_OK(hipMemsetAsync(&bt->st.p->nSlowPost, 0, 8, st));            // nSlowPost, nSlowScore (the rest of the status block is the plan's)
    HIP_OK(hipEventRecord(bt->ev[0], st));
    bool counted = true;
    if (bt->nReads) counted = launchSearch(cl, bt, st) && counted;
    HIP_OK(hipEventRecord(bt->ev[1], st));
    // CF_TAIL_STREAM=1: every per-query kernel on the slot's own stream (see cf_batch::tail; measured in round 3, a loss).
    // CF_TAIL_STREAM=2 (round 4): only what follows the common-case score kernel — the general score kernel (a few thousand
    // one-lane chains of dependent loads: latency, hardly any bandwidth), k_count and the compaction — so that the next batch's
    // search starts while they run; the stage marks of "score" then include whatever ran beside it
    static const int tailMode = envInt("CF_TAIL_STREAM", 0);
    hipStream_t ts = bt->tail && tailMode == 1 ? bt->tail : st;
    hipStream_t late = bt->tail && tailMode == 2 ? bt->tail : ts;
    if (ts != st) HIP_OK(hipStreamWaitEvent(ts, bt->ev[1], 0));
    const bool early = earlyScoreMode(bt);
    enqueuePost(bt, ts);
    scan_enqueue<SCAN_PLAIN>(bt->qRows.p, bt->nQueries, bt->qBase.p, nullptr, bt->tileA.p, bt->tileC.p, ts);
    counted = enqueueRowPass(bt, 0, ts, true, late, early) && counted;
    enqueueCompact(bt, late);
    HIP_OK(hipEventRecord(bt->ev[9], late));
    bt->opsValid = counted;
    bt->passes = 1;
    bt->running = true; bt->finished = false; bt->downloaded = false;
    bt->stream = st;
}

// results and status into the slot's pinned host buffers, then the "done" event
static void enqueueDownload(cf_batch *bt, hipStream_t st) {
    const uint64_t nq = bt->nQueries;
    HIP_OK(hipStreamWaitEvent(st, bt->ev[9], 0));      // the kernels may have run on another (compute) stream
    HIP_OK(hipMemcpyAsync(bt->hSt.p, bt->st.p, sizeof(BatchStatus), hipMemcpyDeviceToHost, st));
    HIP_OK(hipMemcpyAsync(bt->hOps.p, bt->ops.p, sizeof(OpCounts), hipMemcpyDeviceToHost, st));
    if (nq && !bt->rowsStay) {
        const uint64_t spec = std::min<uint64_t>(bt->rowsSpec, nq * (uint64_t)bt->cl->d.k);
        if (bt->resultFormat == CF_RESULTS_NARROW) {             // 16-byte rows, one byte + 2ndBestScore per query
            HIP_OK(hipMemcpyAsync(bt->hQInfo.p, bt->qinfo.p, nq, hipMemcpyDeviceToHost, st));
            HIP_OK(hipMemcpyAsync(bt->hScore2.p, bt->score2.p, nq * 4, hipMemcpyDeviceToHost, st));
            HIP_OK(hipMemcpyAsync(bt->hRows.p, bt->outCompact.p, spec * sizeof(NarrowRow), hipMemcpyDeviceToHost, st));
        } else {
            HIP_OK(hipMemcpyAsync(bt->hNOut.p, bt->nOut.p, nq * 4, hipMemcpyDeviceToHost, st));
            HIP_OK(hipMemcpyAsync(bt->hScore2.p, bt->score2.p, nq * 4, hipMemcpyDeviceToHost, st));
            HIP_OK(hipMemcpyAsync(bt->hMaxScore.p, bt->maxScore.p, nq * 4, hipMemcpyDeviceToHost, st));
            HIP_OK(hipMemcpyAsync(bt->hRows.p, bt->outCompact.p, spec * sizeof(OutRow), hipMemcpyDeviceToHost, st));
        }
    }
    HIP_OK(hipEventRecord(bt->ev[7], st));
    bt->downloaded = true;
}

// Blocks until the batch in flight is done, finishes what the single asynchronous pass could not (a hit pool that
// was too small: the batch is run again with the pool it asked for; more planned rows than the row workspace
// holds: further passes of the row stage), and leaves the results in the slot's pinned buffers.
static void waitBatch(cf_batch *bt) {
    if (!bt->running) throw ArgError("no batch in flight on this slot");
    if (bt->finished) return;
    hipStream_t st = bt->stream;
    if (!bt->downloaded) enqueueDownload(bt, st);
    HIP_OK(hipEventSynchronize(bt->ev[7]));
    HIP_OK(hipGetLastError());
    bool redo = false;
    auto fetchStatus = [&] {
        HIP_OK(hipEventSynchronize(bt->ev[9]));            // (a re-run's per-query kernels live on the slot's own stream)
        HIP_OK(hipMemcpyAsync(bt->hSt.p, bt->st.p, sizeof(BatchStatus), hipMemcpyDeviceToHost, st));
        HIP_OK(hipStreamSynchronize(st));
        HIP_OK(hipGetLastError());
    };
    // The stage marks live on the kernels' stream, which may not be the stream of the "done" event: its own last
    // event is waited for as well (no-op when the work is done), which also brings the marks' status up to date.
    HIP_OK(hipEventSynchronize(bt->ev[9]));
    auto lapse = [&](float &ms, int a, int b) { if (hipEventElapsedTime(&ms, bt->ev[a], bt->ev[b]) != hipSuccess) { ms = 0; (void)hipGetLastError(); } };
    lapse(bt->planMs, 5, 6);
    lapse(bt->ms[0], 0, 1); lapse(bt->ms[1], 1, 2); lapse(bt->ms[2], 2, 3); lapse(bt->ms[3], 3, 4); lapse(bt->ms[4], 0, 4);
    if (bt->hSt.p->flags & kStLenOverflow) throw ArgError("a read is longer than the max_len the batch was submitted with");
    if (bt->hSt.p->flags & kStWordsOverflow) throw ArgError("the read lengths need more packed words than n_words says were uploaded");
    if (bt->hSt.p->flags & kStHitsOverflow) {
        // the reads carry more N than the pool allowed for: it is grown to what the plan asked for, and the batch
        // (whose first attempt searched and scored nothing) runs again
        const uint64_t need = bt->hSt.p->hitsNeed;
        if (bt->recWords && need > 0xffffffffull) throw ArgError("batch too large: its hit lists need 32-bit offsets, split it");
        bt->hitsCapLimit = 0;
        bt->hits.ensure(need + 1);
        bindBatch(bt);
        enqueuePlan(bt, st);
        enqueueClassify(bt, st);
        fetchStatus();
        if (bt->hSt.p->flags & kStHitsOverflow) throw std::logic_error("hit pool still too small after growing it");
        redo = true;
    }
    while (bt->hSt.p->qHi < bt->nQueries) {
        if (bt->hSt.p->qHi == bt->hSt.p->qLo) {          // one query plans more rows than the workspace holds: grow to it
            const uint64_t need = bt->hSt.p->needRows;
            bt->rowsCapLimit = 0;
            bt->rowVal.ensure(need); bt->rowRef.ensure(need); bt->hm.ensure(need); bt->tc.ensure(need);
            bindBatch(bt);
        }
        enqueueRowPass(bt, bt->hSt.p->qHi, st, false);
        bt->passes++;
        bt->opsValid = false;
        fetchStatus();
        redo = true;
    }
    if (redo) {                                            // the first download saw an unfinished batch
        enqueueCompact(bt, st);
        HIP_OK(hipEventRecord(bt->ev[9], st));
        enqueueDownload(bt, st);
        HIP_OK(hipEventSynchronize(bt->ev[7]));
        HIP_OK(hipGetLastError());
    }
    bt->rowsOut = bt->hSt.p->rowsOut;
    bt->rowsTotal = bt->hSt.p->rowsTotal;
    if (bt->nQueries) { bt->rowsPerQuery = (double)bt->rowsOut / (double)bt->nQueries; bt->plannedPerQuery = (double)bt->rowsTotal / (double)bt->nQueries; }
    if (bt->rowsOut > bt->rowsSpec && !bt->rowsStay) {     // more printed rows than the download brought along
        const uint64_t have = bt->rowsSpec;
        const size_t rb = bt->resultFormat == CF_RESULTS_NARROW ? sizeof(NarrowRow) : sizeof(OutRow);
        if (bt->rowsOut > bt->hRows.n) {                   // (and more than the pinned buffer holds: a larger one, with room to spare)
            PinBuf<OutRow> bigger;
            bigger.ensure(bt->rowsOut + bt->rowsOut / 8);
            std::memcpy(bigger.p, bt->hRows.p, have * rb);
            std::swap(bt->hRows.p, bigger.p); std::swap(bt->hRows.n, bigger.n);
        }
        HIP_OK(hipMemcpy(reinterpret_cast<uint8_t *>(bt->hRows.p) + have * rb, reinterpret_cast<const uint8_t *>(bt->outCompact.p) + have * rb, (bt->rowsOut - have) * rb, hipMemcpyDeviceToHost));
    }
    bt->lastOps = *bt->hOps.p;
    bt->lastOps.nRows = bt->rowsTotal;
    bt->finished = true;
}

// reads of the 1-byte-per-base form: staged to the device, then packed there
static void uploadBytes(cf_batch *bt, const uint8_t *seq, const uint64_t *off, const uint32_t *seeds, uint64_t nReads, int paired, hipStream_t st) {
    uint64_t nWords = 0, maxLen = 0;
    for (uint64_t r = 0; r < nReads; r++) {
        if (off[r + 1] < off[r]) throw ArgError("read offsets must be non-decreasing");
        const uint64_t L = off[r + 1] - off[r];
        nWords += (L + 31) >> 5;
        maxLen = std::max(maxLen, L);
    }
    if (maxLen > 0xffffffffull) throw ArgError("a read is longer than 2^32 bases");
    const uint64_t nbases = off[nReads] - off[0];
    if (nbases && !seq) throw ArgError("null sequence buffer");
    sizeBatch(bt, nReads, nWords, nbases, (uint32_t)maxLen, paired);
    bt->seq.ensure(nbases + 16); bt->off8.ensure(nReads + 1);
    bindBatch(bt);
    HIP_OK(hipMemsetAsync(bt->seq.p + nbases, 0, 16, st));
    if (nbases) HIP_OK(hipMemcpyAsync(bt->seq.p, seq + off[0], nbases, hipMemcpyHostToDevice, st));
    // offsets relative to the first read
    if (off[0] == 0) HIP_OK(hipMemcpyAsync(bt->off8.p, off, (nReads + 1) * 8, hipMemcpyHostToDevice, st));
    else {
        std::vector<uint64_t> rel(nReads + 1);
        for (uint64_t r = 0; r <= nReads; r++) rel[r] = off[r] - off[0];
        HIP_OK(hipMemcpy(bt->off8.p, rel.data(), (nReads + 1) * 8, hipMemcpyHostToDevice));
    }
    if (nReads) HIP_OK(hipMemcpyAsync(bt->seeds.p, seeds, nReads * 4, hipMemcpyHostToDevice, st));
    HIP_OK(hipEventRecord(bt->ev[8], st));
    bt->fromBytes = true; bt->fromText = false; bt->densePending = 0; bt->revMade = false; bt->nmaskZeroOf = nullptr;         // (k_convert writes every mask word)
    bt->loaded = true; bt->planned = false; bt->running = false; bt->finished = false;
}

// sparse N mask: zeros, then the few words that hold an N.  The mask buffer is kept zero between batches: a slot that took a sparse
// mask last time only takes back the words it set then (their indices are still on the device) instead of clearing 4 bytes per word again.
static void uploadSparseMask(cf_batch *bt, const uint64_t *idx, const uint32_t *mask, uint64_t nNWords, uint64_t nWords, hipStream_t st) {
    if (bt->nmaskZeroOf != bt->nmask.p || bt->nmaskZeroN != bt->nmask.n) {
        HIP_OK(hipMemsetAsync(bt->nmask.p, 0, bt->nmask.bytes(), st));
        bt->nmaskZeroOf = bt->nmask.p; bt->nmaskZeroN = bt->nmask.n; bt->nSparsePrev = 0;
    } else if (bt->nSparsePrev) {
        hipLaunchKernelGGL(k_scatter_nmask, dim3((unsigned)((bt->nSparsePrev + 255) / 256)), dim3(256), 0, st, bt->nIdx.p, nullptr,
                           bt->nSparsePrev, (uint64_t)bt->nmask.n, bt->nmask.p);
        bt->nSparsePrev = 0;
    }
    if (nNWords) {
        bt->nSparsePrev = nNWords;
        bt->nIdx.ensure(nNWords); bt->nMsk.ensure(nNWords);
        HIP_OK(hipMemcpyAsync(bt->nIdx.p, idx, nNWords * 8, hipMemcpyHostToDevice, st));
        HIP_OK(hipMemcpyAsync(bt->nMsk.p, mask, nNWords * 4, hipMemcpyHostToDevice, st));
        hipLaunchKernelGGL(k_scatter_nmask, dim3((unsigned)((nNWords + 255) / 256)), dim3(256), 0, st, bt->nIdx.p, bt->nMsk.p,
                           nNWords, nWords, bt->nmask.p);
    }
}

static void uploadPacked(cf_batch *bt, const cf_packed_reads *in, hipStream_t st) {
    if (in->n_reads && (!in->len || !in->seeds)) throw ArgError("null length / seed array");
    if (in->n_words && !in->bases) throw ArgError("null packed-base array");
    if (!in->nmask && in->n_nwords && (!in->nword_idx || !in->nword_mask)) throw ArgError("null sparse N-mask arrays");
    sizeBatch(bt, in->n_reads, in->n_words, in->n_bases, in->max_len, in->paired);
    bindBatch(bt);
    if (in->n_words) {
        HIP_OK(hipMemcpyAsync(bt->bases.p, in->bases, in->n_words * 8, hipMemcpyHostToDevice, st));
        if (in->nmask) { HIP_OK(hipMemcpyAsync(bt->nmask.p, in->nmask, in->n_words * 4, hipMemcpyHostToDevice, st)); bt->nmaskZeroOf = nullptr; }
        else uploadSparseMask(bt, in->nword_idx, in->nword_mask, in->n_nwords, in->n_words, st);
    }
    if (in->n_reads) {
        HIP_OK(hipMemcpyAsync(bt->rlen.p, in->len, in->n_reads * 4, hipMemcpyHostToDevice, st));
        HIP_OK(hipMemcpyAsync(bt->seeds.p, in->seeds, in->n_reads * 4, hipMemcpyHostToDevice, st));
    }
    HIP_OK(hipEventRecord(bt->ev[8], st));             // the upload stage is copies only: it can live on a copy stream
    bt->fromBytes = false; bt->fromText = false; bt->densePending = 0; bt->revMade = false;
    bt->loaded = true; bt->planned = false; bt->running = false; bt->finished = false;
}

// the dense form (cf_dense_reads): the byte stream up, then unpacked into the word form on the device (one thread per word)
static void uploadDense(cf_batch *bt, const cf_dense_reads *in, hipStream_t st) {
    if (in->n_reads && (!in->bases4 || !in->seeds)) throw ArgError("null dense-base / seed array");
    if (in->n_nwords && (!in->nword_idx || !in->nword_mask)) throw ArgError("null sparse N-mask arrays");
    if (in->read_len > kMaxReadLen) throw ArgError("reads of more than 16,777,213 bases are not supported (24-bit offsets in the hit records)");
    const uint64_t W = ((uint64_t)in->read_len + 31) >> 5, bpr = ((uint64_t)in->read_len + 3) >> 2;
    const uint64_t nWords = in->n_reads * W, nBytes = in->n_reads * bpr;
    sizeBatch(bt, in->n_reads, nWords, in->n_reads * (uint64_t)in->read_len, in->read_len, in->paired);
    bt->dense.ensure(nBytes + 32);
    bindBatch(bt);
    if (in->n_reads) {
        HIP_OK(hipMemcpyAsync(bt->dense.p, in->bases4, nBytes, hipMemcpyHostToDevice, st));
        HIP_OK(hipMemsetAsync(bt->dense.p + nBytes, 0, 32, st));             // (the unpack kernel reads whole 8-byte pieces)
        HIP_OK(hipMemcpyAsync(bt->seeds.p, in->seeds, in->n_reads * 4, hipMemcpyHostToDevice, st));
        if (nWords) uploadSparseMask(bt, in->nword_idx, in->nword_mask, in->n_nwords, nWords, st);
    }
    // The words are made by the PLAN stage, on the kernels' stream (enqueuePlan): a kernel on the copy stream waits for CUs the
    // persistent search kernel of the batch before holds, and everything behind it on that stream — the next slots' copies — waits
    // with it (measured: k_dense_unpack 0.14 ms alone, up to 5.7 ms there; host to host 1.00 against 1.16e9 reads/s for the word form)
    bt->densePending = in->n_reads ? in->read_len + 1 : 0; bt->revMade = false;
    HIP_OK(hipEventRecord(bt->ev[8], st));
    bt->fromBytes = false; bt->fromText = false;
    bt->loaded = true; bt->planned = false; bt->running = false; bt->finished = false;
}


// a block of FASTA / FASTQ text (cf_text_reads): up as it is, then on the device — where its records start, the plain-form checks,
// lengths, seeds and places per record (cf_textio.hpp) — and ONE wait for the status: the number of reads sizes the slot.  A block
// that is not in the plain form leaves the slot without a batch (info->irregular says why): the caller's host parser takes it.
static void uploadText(cf_batch *bt, const cf_text_reads *in, hipStream_t st, cf_text_info *info) {
    if (in->format != CF_TEXT_FASTA && in->format != CF_TEXT_FASTQ) throw ArgError("cf_text_reads::format is CF_TEXT_FASTA or CF_TEXT_FASTQ");
    const int nBlocks = in->text2 ? 2 : 1;
    const char *src[2] = {in->text, in->text2};
    const uint64_t nBs[2] = {in->n_bytes, in->text2 ? in->n_bytes2 : 0};
    if ((nBs[0] && !src[0]) || (nBs[1] && !src[1])) throw ArgError("null text block");
    if (nBs[0] + nBs[1] >= 0xffff0000ull) throw ArgError("the text blocks of a batch hold fewer than 2^32 bytes (32-bit places in them)");
    const bool fasta = in->format == CF_TEXT_FASTA;
    // The blocks lie one behind the other in one buffer, each followed by zero bytes; the per-piece counts, their sums and the marker
    // places likewise.  Room for records of 32 bytes on average (a name and 22 bases take that): a block of shorter ones is the
    // host parser's.
    uint64_t at[2], pieces[2], pieceAt[2], recCap[2], posCap[2], posAt[2], textBytes = 0, nPieces = 0, posTotal = 0, recMax = 0;
    for (int k = 0; k < nBlocks; k++) {
        at[k] = textBytes; pieces[k] = (nBs[k] + kTextPiece - 1) / kTextPiece; pieceAt[k] = nPieces;
        recCap[k] = nBs[k] / 32 + 1024; posCap[k] = fasta ? recCap[k] : 4 * recCap[k]; posAt[k] = posTotal;
        textBytes += pieces[k] * kTextPiece + kTextPad; nPieces += pieces[k] + 16; posTotal += posCap[k] + 16;
        recMax = std::max(recMax, recCap[k]);
    }
    const uint64_t readCap = recMax * (uint64_t)nBlocks;
    bt->text.ensure(textBytes);
    bt->txCnt.ensure(nPieces + 16); bt->txBase.ensure(nPieces + 16); bt->txPos.ensure(posTotal + 16);
    bt->txTileA.ensure(scan_tiles_for(std::max(pieces[0], pieces[nBlocks - 1])) + 1); bt->txTileC.ensure(scan_tiles_for(std::max(pieces[0], pieces[nBlocks - 1])) + 1);
    bt->rlen.ensure(readCap + 16); bt->seeds.ensure(readCap + 16);
    bt->txSeqOff.ensure(readCap + 16); bt->txIdOff.ensure(readCap + 16); bt->txIdLen.ensure(readCap + 16);
    bt->txSt.ensure(1); bt->hTxSt.ensure(1); bt->hTxTotal.ensure(2);
    *info = cf_text_info{};
    bt->loaded = false; bt->planned = false; bt->running = false; bt->finished = false;
    HIP_OK(hipMemsetAsync(bt->txSt.p, 0, sizeof(TextStatus), st));
    const uint32_t seed0 = (in->global_seed + 101u) * 59u * 61u * 67u * 71u * 73u * 79u * 83u;
    const dim3 bl(256);
    for (int k = 0; k < nBlocks; k++) {
        uint8_t *text = bt->text.p + at[k];
        HIP_OK(hipMemsetAsync(text + nBs[k], 0, pieces[k] * kTextPiece + kTextPad - nBs[k], st));
        if (nBs[k]) HIP_OK(hipMemcpyAsync(text, src[k], nBs[k], hipMemcpyHostToDevice, st));
        uint32_t *cnt = bt->txCnt.p + pieceAt[k], *pos = bt->txPos.p + posAt[k];
        uint64_t *base = bt->txBase.p + pieceAt[k];
        const DTextMark m{text, nBs[k], fasta ? (uint32_t)'>' : (uint32_t)'\n', cnt, base, pos, posCap[k]};
        const dim3 gp((unsigned)std::max<uint64_t>(1, (pieces[k] + 255) / 256));
        if (pieces[k]) hipLaunchKernelGGL(k_text_count, gp, bl, 0, st, m);
        scan_enqueue<SCAN_PLAIN>(cnt, pieces[k], base, nullptr, bt->txTileA.p, bt->txTileC.p, st);
        if (pieces[k]) hipLaunchKernelGGL(k_text_mark, gp, bl, 0, st, m);
        DTextRec d{text, nBs[k], pos, base + pieces[k], posCap[k], (uint32_t)recCap[k], (uint32_t)in->format, seed0,
                   bt->rlen.p, bt->seeds.p, bt->txSeqOff.p, bt->txIdOff.p, bt->txIdLen.p, bt->txSt.p, (uint32_t)at[k], (uint32_t)nBlocks, (uint32_t)k};
        hipLaunchKernelGGL(k_text_records, dim3((unsigned)((recCap[k] + 255) / 256)), bl, 0, st, d);
        HIP_OK(hipMemcpyAsync(bt->hTxTotal.p + k, base + pieces[k], 8, hipMemcpyDeviceToHost, st));
    }
    HIP_OK(hipMemcpyAsync(bt->hTxSt.p, bt->txSt.p, sizeof(TextStatus), hipMemcpyDeviceToHost, st));
    HIP_OK(hipStreamSynchronize(st));
    HIP_OK(hipGetLastError());
    const TextStatus ts = *bt->hTxSt.p;
    uint64_t nRec[2] = {0, 0};
    for (int k = 0; k < nBlocks; k++) nRec[k] = fasta ? bt->hTxTotal.p[k] : bt->hTxTotal.p[k] >> 2;
    if (ts.flags) { info->irregular = ts.flags; return; }
    if (nBlocks == 2 && nRec[0] != nRec[1]) { info->irregular = kTxMateCount; return; }
    uint64_t nq = nRec[0];
    if (in->max_reads && nq > in->max_reads) nq = in->max_reads;     // (the sums below then cover a few reads too many: upper bounds, as they may be)
    const uint64_t nReads = nq * (uint64_t)nBlocks;
    sizeBatch(bt, nReads, ts.words(), ts.bases(), ts.maxLen, nBlocks == 2);
    bindBatch(bt);
    HIP_OK(hipEventRecord(bt->ev[8], st));
    bt->fromText = true; bt->fromBytes = false; bt->densePending = 0; bt->revMade = false; bt->nmaskZeroOf = nullptr;   // (k_text_pack writes every mask word)
    bt->loaded = true;
    info->n_reads = nReads; info->n_bases = ts.bases(); info->max_len = ts.maxLen;
}

// ======================================================================= batch C ABI
cf_status cf_host_alloc(void **p, size_t bytes) {
    if (!p) return CF_ERR_ARG;
    *p = nullptr;
    if (!haveDevice()) { g_err = "no HIP device visible"; return CF_ERR_NO_DEVICE; }
    return guard([&] { HIP_OK(hipHostMalloc(p, bytes ? bytes : 1, hipHostMallocDefault)); });
}
void cf_host_free(void *p) { if (p) (void)hipHostFree(p); }

cf_status cf_batch_alloc(cf_classifier *cl, uint64_t maxReads, uint64_t maxWords, cf_batch **out) {
    if (!cl || !out) return CF_ERR_ARG;
    *out = nullptr;
    auto bt = std::make_unique<cf_batch>();
    cf_status st = guard([&] {
        HIP_OK(hipSetDevice(cl->ix->device));
        bt->cl = cl;
        const uint64_t per = maxReads ? (32 * maxWords + maxReads - 1) / maxReads : 0;        // bases per read, rounded up to words
        sizeBatch(bt.get(), maxReads & ~1ull, maxWords, 0, (uint32_t)std::min<uint64_t>(per, 128), 0);
        bt->nReads = bt->nQueries = bt->nWords = 0;
    });
    if (st == CF_OK) *out = bt.release();
    return st;
}

// Device memory a slot of cf_batch_alloc(max_reads, max_words) takes, without a device: what a caller subtracts from the HBM it
// offers the index (cf_index_options::hbm_budget_bytes) when it knows its slots — cf_index_open's own reserve for them is a
// fifth of the device, which three slots of 10 M mates of 150 bases (3 x 15.5 GB) nearly use up and slots of 4 M reads waste.
cf_status cf_slot_estimate_bytes(uint64_t maxReads, uint64_t maxWords, int khits, int ftabChars, int occPlanes, uint64_t *bytes) {
    if (!bytes || khits < 1 || ftabChars < 1) return CF_ERR_ARG;
    *bytes = 0;
    return guard([&] {
        cf_index ix;
        ix.h.g.ftabChars = ftabChars;
        if (occPlanes) ix.d.planes = reinterpret_cast<const uint8_t *>(&ix);      // (only asked whether it is there)
        cf_classifier cl;
        cl.ix = &ix; cl.d.k = (uint32_t)khits;
        uint64_t total = 0;
        g_dryBytes = &total;
        try {
            cf_batch bt;
            bt.cl = &cl;
            const uint64_t per = maxReads ? (32 * maxWords + maxReads - 1) / maxReads : 0;
            sizeBatch(&bt, maxReads & ~1ull, maxWords, 0, (uint32_t)std::min<uint64_t>(per, 128), 0);
            bt.evInit = false;
        } catch (...) { g_dryBytes = nullptr; throw; }
        g_dryBytes = nullptr;
        *bytes = total;
    });
}

cf_status cf_batch_upload_packed_async(cf_batch *bt, const cf_packed_reads *in, void *streamv) {
    if (!bt || !in) return CF_ERR_ARG;
    return guard([&] {
        HIP_OK(hipSetDevice(bt->cl->ix->device));
        if (bt->running && !bt->finished) throw ArgError("the slot still has a batch in flight: cf_batch_wait first");
        uploadPacked(bt, in, static_cast<hipStream_t>(streamv));
    });
}

cf_status cf_batch_upload_dense_async(cf_batch *bt, const cf_dense_reads *in, void *streamv) {
    if (!bt || !in) return CF_ERR_ARG;
    return guard([&] {
        HIP_OK(hipSetDevice(bt->cl->ix->device));
        if (bt->running && !bt->finished) throw ArgError("the slot still has a batch in flight: cf_batch_wait first");
        uploadDense(bt, in, static_cast<hipStream_t>(streamv));
    });
}

cf_status cf_batch_set_result_format(cf_batch *bt, int format) {
    if (!bt || (format != CF_RESULTS_ROWS && format != CF_RESULTS_NARROW)) return CF_ERR_ARG;
    if (format == CF_RESULTS_NARROW && bt->cl->d.k > 63) { g_err = "the narrow result format holds a query's row count in six bits: -k <= 63"; return CF_ERR_ARG; }
    if (bt->running && !bt->finished) { g_err = "the slot still has a batch in flight: cf_batch_wait first"; return CF_ERR_ARG; }
    return guard([&] {
        HIP_OK(hipSetDevice(bt->cl->ix->device));
        // the host buffers of a finished batch hold rows of the OLD format: after a change they are not results any more (the
        // accessors would read 16-byte rows as 24-byte ones); the batch is classified again to get them in the new form
        if (format != bt->resultFormat) { bt->finished = false; bt->running = false; bt->downloaded = false; }
        bt->resultFormat = format;
        if (format == CF_RESULTS_NARROW) { bt->qinfo.ensure(bt->nOut.n + 16); bt->hQInfo.ensure(bt->nOut.n + 16); }
    });
}

cf_status cf_classify_async(cf_classifier *cl, cf_batch *bt, void *streamv) {
    if (!cl || !bt || bt->cl != cl) return CF_ERR_ARG;
    return guard([&] {
        HIP_OK(hipSetDevice(cl->ix->device));
        if (!bt->loaded) throw ArgError("no reads were uploaded into this slot");
        hipStream_t st = static_cast<hipStream_t>(streamv);
        bt->rowsStay = false; bt->textDone = false;
        if (!bt->planned) enqueuePlan(bt, st);
        enqueueClassify(bt, st);
    });
}

// the whole device side of a batch once more on the reads the slot already holds: plan + kernels, nothing uploaded (a caller whose
// reads are produced on the device, or stay there over several runs with other options; bench.py's "inputs resident in HBM")
cf_status cf_batch_reclassify_async(cf_classifier *cl, cf_batch *bt, void *streamv) {
    if (!cl || !bt || bt->cl != cl) return CF_ERR_ARG;
    return guard([&] {
        HIP_OK(hipSetDevice(cl->ix->device));
        if (!bt->loaded) throw ArgError("no reads were uploaded into this slot");
        if (bt->running && !bt->finished) throw ArgError("the slot still has a batch in flight: cf_batch_wait first");
        hipStream_t st = static_cast<hipStream_t>(streamv);
        bt->rowsStay = false; bt->textDone = false;
        enqueuePlan(bt, st);
        enqueueClassify(bt, st);
    });
}

cf_status cf_batch_download_async(cf_batch *bt, void *streamv) {
    if (!bt) return CF_ERR_ARG;
    return guard([&] {
        HIP_OK(hipSetDevice(bt->cl->ix->device));
        if (!bt->running) throw ArgError("cf_classify_async first");
        enqueueDownload(bt, static_cast<hipStream_t>(streamv));
    });
}

cf_status cf_batch_submit(cf_batch *bt, const cf_packed_reads *in, void *streamv) {
    cf_status s = cf_batch_upload_packed_async(bt, in, streamv);
    if (s == CF_OK) s = cf_classify_async(bt->cl, bt, streamv);
    if (s == CF_OK) s = cf_batch_download_async(bt, streamv);
    return s;
}

cf_status cf_batch_wait_narrow(cf_batch *bt, cf_results_narrow *res) {
    if (!bt) return CF_ERR_ARG;
    if (bt->resultFormat != CF_RESULTS_NARROW) { g_err = "cf_batch_wait_narrow on a slot whose result format is not CF_RESULTS_NARROW"; return CF_ERR_ARG; }
    const cf_status rc = guard([&] {
        HIP_OK(hipSetDevice(bt->cl->ix->device));
        if (bt->rowsStay && bt->finished) throw ArgError("the batch was finished by cf_batch_wait_text: its rows were not downloaded");
        waitBatch(bt);
        if (res) {
            static_assert(sizeof(cf_row16) == sizeof(NarrowRow), "cf_row16 layout");
            res->rows = reinterpret_cast<const cf_row16 *>(bt->hRows.p);
            res->qinfo = bt->hQInfo.p; res->score2 = bt->hScore2.p;
            res->n_queries = bt->nQueries; res->total_rows = bt->rowsOut;
            res->planned_sa_rows = bt->rowsTotal; res->row_passes = bt->passes;
            res->slow_post = bt->hSt.p->nSlowPost; res->slow_score = bt->hSt.p->nSlowScore;
        }
    });
    if (rc != CF_OK) {                                     // the batch is lost; the slot takes the next one
        (void)hipStreamSynchronize(bt->stream);
        bt->running = false; bt->finished = false;
    }
    return rc;
}

uint32_t cf_narrow_max_score(uint8_t qinfo, uint32_t len1, uint32_t len2, int paired) {
    // plan_maxscore_body (classifier.h:530-536), from the "took part" bits of the query's byte
    const uint64_t s0 = len1 > 15 ? (uint64_t)(len1 - 15) * (len1 - 15) : 0u, s1 = len2 > 15 ? (uint64_t)(len2 - 15) * (len2 - 15) : 0u;
    const bool p0 = (qinfo & 0x40u) != 0, p1 = paired && (qinfo & 0x80u) != 0;
    const uint64_t v = (p0 && p1) ? s0 + s1 : p0 ? s0 : p1 ? s1 : 0u;
    return v >= 0xffffffffull ? kMaxScoreNever : (uint32_t)v;
}

cf_status cf_results_narrow_expand(const cf_index *ix, const cf_results_narrow *r, const uint32_t *len, uint32_t uniformLen, int paired,
                                   cf_row *rows, uint32_t *nRows, uint32_t *maxScore) {
    if (!ix || !r || (r->total_rows && !rows) || (r->n_queries && (!nRows || !maxScore))) return CF_ERR_ARG;
    const uint64_t nTaxa = ix->h.taxa.size();
    for (uint64_t i = 0; i < r->total_rows; i++) {
        const cf_row16 &n = r->rows[i];
        if (n.taxon_idx >= nTaxa) { g_err = "cf_results_narrow_expand: a row's taxon index lies outside the index's taxon table"; return CF_ERR_ARG; }
        rows[i].tax_id = ix->h.taxa[n.taxon_idx]; rows[i].unique_id = n.unique_id; rows[i].score = n.score; rows[i].hit_len = n.hit_len; rows[i].taxon_idx = n.taxon_idx;
    }
    for (uint64_t q = 0; q < r->n_queries; q++) {
        const uint64_t r0 = paired ? 2 * q : q;
        nRows[q] = r->qinfo[q] & 0x3fu;
        maxScore[q] = cf_narrow_max_score(r->qinfo[q], len ? len[r0] : uniformLen, paired ? (len ? len[r0 + 1] : uniformLen) : 0u, paired);
    }
    return CF_OK;
}

cf_status cf_batch_wait(cf_batch *bt, cf_results *res) {
    if (!bt) return CF_ERR_ARG;
    // a format mix-up is the caller's slip, not the batch's: refused before anything that would give the batch up
    if (res && bt->resultFormat != CF_RESULTS_ROWS) { g_err = "the slot's result format is CF_RESULTS_NARROW: cf_batch_wait_narrow"; return CF_ERR_ARG; }
    const cf_status rc = guard([&] {
        HIP_OK(hipSetDevice(bt->cl->ix->device));
        if (bt->rowsStay && bt->finished) throw ArgError("the batch was finished by cf_batch_wait_text: its rows were not downloaded");
        waitBatch(bt);
        if (res) {
            static_assert(sizeof(cf_row) == sizeof(OutRow), "cf_row layout");
            res->rows = reinterpret_cast<const cf_row *>(bt->hRows.p);
            res->n_rows = bt->hNOut.p; res->score2 = bt->hScore2.p; res->max_score = bt->hMaxScore.p;
            res->n_queries = bt->nQueries; res->total_rows = bt->rowsOut;
            res->planned_sa_rows = bt->rowsTotal; res->row_passes = bt->passes;
            res->slow_post = bt->hSt.p->nSlowPost; res->slow_score = bt->hSt.p->nSlowScore;
        }
    });
    if (rc != CF_OK) {                                     // the batch is lost; the slot takes the next one
        (void)hipStreamSynchronize(bt->stream);
        bt->running = false; bt->finished = false;
    }
    return rc;
}

cf_status cf_batch_upload_text(cf_batch *bt, const cf_text_reads *in, void *streamv, cf_text_info *info) {
    if (!bt || !in || !info) return CF_ERR_ARG;
    return guard([&] {
        HIP_OK(hipSetDevice(bt->cl->ix->device));
        if (bt->running && !bt->finished) throw ArgError("the slot still has a batch in flight: cf_batch_wait first");
        uploadText(bt, in, static_cast<hipStream_t>(streamv), info);
    });
}

// the strings a formatted row repeats, on the device: per reference its uid, per taxon the seqID of a row that names no
// reference (classifier.h:546-557 + aln_sink.h:2219-2234) and the taxID column (aln_sink.h:2236-2250)
static void makeFormatTables(cf_classifier *cl) {
    std::lock_guard<std::mutex> lk(cl->fmtMu);
    if (cl->fmtMade) return;
    const HostIndex &h = cl->ix->h;
    const size_t nTaxa = h.taxa.size(), nRefs = h.uid.size();
    std::vector<uint8_t> strs, leaf(nTaxa, 0);
    std::vector<uint32_t> uidOff(nRefs + 1), rankOff(nTaxa + 1), taxOff(nTaxa + 1);
    auto put = [&](const std::string &x) { strs.insert(strs.end(), x.begin(), x.end()); };
    for (size_t r = 0; r < nRefs; r++) { uidOff[r] = (uint32_t)strs.size(); put(h.uid[r]); }
    uidOff[nRefs] = (uint32_t)strs.size();
    for (size_t i = 0; i < nTaxa; i++) {
        const uint64_t t = h.taxa[i];
        rankOff[i] = (uint32_t)strs.size();
        const char *viaMerged = h.formatSeqId(CF_MERGED, t);
        put(viaMerged);
        leaf[i] = nRefs && h.formatSeqId(0, t) != viaMerged ? 1 : 0;
    }
    rankOff[nTaxa] = (uint32_t)strs.size();
    for (size_t i = 0; i < nTaxa; i++) {
        const uint64_t t = h.taxa[i];
        taxOff[i] = (uint32_t)strs.size();
        put(std::to_string(t & 0xffffffffull));
        if (t >> 32) { strs.push_back('.'); put(std::to_string(t >> 32)); }
    }
    taxOff[nTaxa] = (uint32_t)strs.size();
    if (strs.size() >= 0xffffffffull) throw ArgError("the index's reference and rank names exceed 4 GB");
    strs.resize(strs.size() + 16, 0);
    cl->fmtStrs.upload(strs); cl->fmtLeaf.upload(leaf); cl->fmtUidOff.upload(uidOff); cl->fmtRankOff.upload(rankOff); cl->fmtTaxOff.upload(taxOff);
    cl->fmtIdxZero = h.taxonIndex(0);
    cl->fmtMade = true;
}

// The results of a batch that came as text, as text: the default columns formatted on the device from the rows the kernels left
// there (none of them crosses the link), and the perfect multi-assignment tuples the report's EM needs beside the device's counters.
cf_status cf_batch_wait_text(cf_batch *bt, cf_results_text *res) {
    if (!bt || !res) return CF_ERR_ARG;
    if (bt->resultFormat != CF_RESULTS_NARROW) { g_err = "cf_batch_wait_text needs a slot whose result format is CF_RESULTS_NARROW"; return CF_ERR_ARG; }
    if (!bt->fromText) { g_err = "cf_batch_wait_text: the slot's reads did not come as text (cf_batch_upload_text): there are no readIDs on the device"; return CF_ERR_ARG; }
    const cf_status rc = guard([&] {
        cf_classifier *cl = bt->cl;
        HIP_OK(hipSetDevice(cl->ix->device));
        if (!bt->finished) bt->rowsStay = true;
        else if (!bt->rowsStay) throw ArgError("the batch was already finished by another wait");
        waitBatch(bt);
        if (bt->textDone) {
            res->text = reinterpret_cast<const char *>(bt->hTextOut.p); res->n_bytes = bt->textBytes;
            res->tuples = bt->hTuples.p; res->n_tuple_words = bt->tupleWords;
            res->n_queries = bt->nQueries; res->total_rows = bt->rowsOut; res->planned_sa_rows = bt->rowsTotal; res->row_passes = bt->passes;
            res->slow_post = bt->hSt.p->nSlowPost; res->slow_score = bt->hSt.p->nSlowScore;
            return;
        }
        makeFormatTables(cl);
        hipStream_t st = bt->stream;
        const uint64_t nq = bt->nQueries;
        const uint32_t nTaxa = (uint32_t)cl->ix->h.taxa.size();
        const uint64_t tuplesCap = std::min<uint64_t>(nq * ((uint64_t)cl->d.k + 1) + 16, 0xffffffffull);
        bt->txSize.ensure(nq + 16); bt->txOutOff.ensure(nq + 16); bt->txTuples.ensure(tuplesCap);
        bt->txTileA.ensure(scan_tiles_for(nq) + 1); bt->txTileC.ensure(scan_tiles_for(nq) + 1);
        HIP_OK(hipMemsetAsync(bt->txSt.p, 0, sizeof(TextStatus), st));
        DTextFmt f{};
        f.text = bt->text.p; f.idOff = bt->txIdOff.p; f.idLen = bt->txIdLen.p; f.rlen = bt->rlen.p;
        static_assert(sizeof(TextRow) == sizeof(NarrowRow), "TextRow layout");
        f.rows = reinterpret_cast<const TextRow *>(bt->outCompact.p); f.rowFirst = bt->rowFirst.p; f.qinfo = bt->qinfo.p;
        f.score2 = bt->score2.p; f.maxScore = bt->maxScore.p; f.nQueries = (uint32_t)nq; f.paired = (uint32_t)bt->paired;
        f.strs = cl->fmtStrs.p; f.uidOff = cl->fmtUidOff.p; f.rankOff = cl->fmtRankOff.p; f.taxOff = cl->fmtTaxOff.p; f.taxLeaf = cl->fmtLeaf.p;
        f.nRefs = (uint32_t)cl->ix->h.uid.size(); f.nTaxa = nTaxa; f.idxZero = cl->fmtIdxZero;
        f.size = bt->txSize.p; f.outOff = bt->txOutOff.p; f.single = cl->counts.p + 2 * (size_t)nTaxa;
        f.tuples = bt->txTuples.p; f.tuplesCap = (uint32_t)tuplesCap; f.st = bt->txSt.p;
        const dim3 bl(256), gq((unsigned)std::max<uint64_t>(1, (nq + 255) / 256));
        uint64_t total = 0;
        if (nq) {
            hipLaunchKernelGGL(k_fmt_size, gq, bl, 0, st, f);
            scan_enqueue<SCAN_PLAIN>(bt->txSize.p, nq, bt->txOutOff.p, nullptr, bt->txTileA.p, bt->txTileC.p, st);
            HIP_OK(hipMemcpyAsync(bt->hTxTotal.p, bt->txOutOff.p + nq, 8, hipMemcpyDeviceToHost, st));
            HIP_OK(hipStreamSynchronize(st));
            total = *bt->hTxTotal.p;
            bt->textOut.ensure(total + 16); bt->hTextOut.ensure(total + 16);
            f.out = bt->textOut.p; f.outCap = total;
            hipLaunchKernelGGL(k_fmt_write, gq, bl, 0, st, f);
            HIP_OK(hipMemcpyAsync(bt->hTextOut.p, bt->textOut.p, total, hipMemcpyDeviceToHost, st));
            HIP_OK(hipMemcpyAsync(bt->hTxSt.p, bt->txSt.p, sizeof(TextStatus), hipMemcpyDeviceToHost, st));
            HIP_OK(hipStreamSynchronize(st));
            HIP_OK(hipGetLastError());
        }
        const uint64_t tw = nq ? bt->hTxSt.p->tupleWords : 0;
        if (tw > tuplesCap) throw std::logic_error("the tuple list outgrew its buffer");
        if (tw) {
            bt->hTuples.ensure(tw);
            HIP_OK(hipMemcpy(bt->hTuples.p, bt->txTuples.p, tw * 4, hipMemcpyDeviceToHost));
        }
        bt->textDone = true; bt->textBytes = total; bt->tupleWords = tw;
        res->text = reinterpret_cast<const char *>(bt->hTextOut.p); res->n_bytes = total;
        res->tuples = bt->hTuples.p; res->n_tuple_words = tw;
        res->n_queries = nq; res->total_rows = bt->rowsOut; res->planned_sa_rows = bt->rowsTotal; res->row_passes = bt->passes;
        res->slow_post = bt->hSt.p->nSlowPost; res->slow_score = bt->hSt.p->nSlowScore;
    });
    if (rc != CF_OK) {                                     // the batch is lost; the slot takes the next one
        (void)hipStreamSynchronize(bt->stream);
        bt->running = false; bt->finished = false;
    }
    return rc;
}

cf_status cf_batch_set_limits(cf_batch *bt, uint64_t hitSlots, uint64_t rowsPerPass) {
    if (!bt) return CF_ERR_ARG;
    bt->hitsCapLimit = hitSlots; bt->rowsCapLimit = rowsPerPass;
    return CF_OK;
}

// The one-shot form: a slot sized for these reads, the reads uploaded and packed, the plan made.
cf_status cf_batch_create(cf_classifier *cl, const uint8_t *seq, const uint64_t *off, const uint32_t *seeds,
                          uint64_t nReads, int paired, cf_batch **out) {
    if (!cl || !off || !seeds || !out || (paired && (nReads & 1)) || nReads >= 0x7fffffffull) return CF_ERR_ARG;
    *out = nullptr;
    auto bt = std::make_unique<cf_batch>();
    cf_status st = guard([&] {
        HIP_OK(hipSetDevice(cl->ix->device));
        bt->cl = cl;
        uploadBytes(bt.get(), seq, off, seeds, nReads, paired, nullptr);
        enqueuePlan(bt.get(), nullptr);
        HIP_OK(hipStreamSynchronize(nullptr));
        HIP_OK(hipGetLastError());
    });
    if (st == CF_OK) *out = bt.release();
    return st;
}
cf_status cf_batch_upload(cf_batch *bt, const uint8_t *seq, const uint64_t *off, const uint32_t *seeds, uint64_t nReads, int paired, void *streamv) {
    if (!bt || !off || !seeds || (paired && (nReads & 1))) return CF_ERR_ARG;
    return guard([&] {
        HIP_OK(hipSetDevice(bt->cl->ix->device));
        if (bt->running && !bt->finished) throw ArgError("the slot still has a batch in flight: cf_batch_wait first");
        uploadBytes(bt, seq, off, seeds, nReads, paired, static_cast<hipStream_t>(streamv));
    });
}
void cf_batch_destroy(cf_batch *b) { delete b; }
uint64_t cf_batch_num_queries(const cf_batch *b) { return b->nQueries; }

// The device-side preparation of a batch again, from its resident packed reads: plan + strand records.  A caller
// that measures (or re-uses the resident reads) runs it as the first stage of a pass.
cf_status cf_batch_plan(cf_batch *bt, void *streamv) {
    if (!bt) return CF_ERR_ARG;
    return guard([&] {
        HIP_OK(hipSetDevice(bt->cl->ix->device));
        if (!bt->loaded) throw ArgError("no reads were uploaded into this slot");
        hipStream_t st = static_cast<hipStream_t>(streamv);
        enqueuePlan(bt, st);
        HIP_OK(hipEventSynchronize(bt->ev[6]));
        HIP_OK(hipGetLastError());
        HIP_OK(hipEventElapsedTime(&bt->planMs, bt->ev[5], bt->ev[6]));
    });
}
cf_status cf_batch_plan_ms(const cf_batch *bt, float *ms) {
    if (!bt || !ms) return CF_ERR_ARG;
    *ms = bt->planMs;
    return CF_OK;
}

cf_status cf_classify(cf_classifier *cl, cf_batch *bt, void *streamv) {
    cf_status s = cf_classify_async(cl, bt, streamv);
    if (s == CF_OK) s = cf_batch_download_async(bt, streamv);
    if (s == CF_OK) s = cf_batch_wait(bt, nullptr);
    return s;
}

static void needFinished(const cf_batch *bt) {
    if (!bt->finished) throw ArgError("the batch has not been classified (or waited for)");
    if (bt->resultFormat != CF_RESULTS_ROWS) throw ArgError("the slot's result format is CF_RESULTS_NARROW: cf_batch_wait_narrow / cf_results_narrow_expand");
}

cf_status cf_batch_results(cf_batch *bt, cf_row *rows, uint32_t *nRows, uint32_t *score2) {
    if (!bt || !rows || !nRows || !score2) return CF_ERR_ARG;
    return guard([&] {
        HIP_OK(hipSetDevice(bt->cl->ix->device));
        needFinished(bt);
        // the k-slots-per-query form, from the packed rows the slot brought back
        const uint32_t k = bt->cl->d.k;
        uint64_t at = 0;
        for (uint64_t q = 0; q < bt->nQueries; q++) {
            const uint32_t n = std::min(bt->hNOut.p[q], k);
            std::memcpy(rows + q * k, bt->hRows.p + at, (size_t)n * sizeof(OutRow));
            at += n;
        }
        std::memcpy(nRows, bt->hNOut.p, bt->nQueries * 4);
        std::memcpy(score2, bt->hScore2.p, bt->nQueries * 4);
    });
}

cf_status cf_batch_max_scores(const cf_batch *bt, uint32_t *out) {
    if (!bt || !out) return CF_ERR_ARG;
    return guard([&] {
        HIP_OK(hipSetDevice(bt->cl->ix->device));
        if (bt->finished && bt->resultFormat == CF_RESULTS_ROWS) std::memcpy(out, bt->hMaxScore.p, bt->nQueries * 4);
        else if (bt->nQueries) {
            if (!bt->planned) throw ArgError("the batch has no plan yet");
            HIP_OK(hipMemcpy(out, bt->maxScore.p, bt->nQueries * 4, hipMemcpyDeviceToHost));
        }
    });
}

cf_status cf_batch_num_rows(cf_batch *bt, uint64_t *total) {
    if (!bt || !total) return CF_ERR_ARG;
    return guard([&] { needFinished(bt); *total = bt->rowsOut; });
}

cf_status cf_batch_results_compact(cf_batch *bt, cf_row *rows, uint64_t rowsCap, uint32_t *nRows, uint32_t *score2) {
    if (!bt || !nRows || !score2 || (!rows && rowsCap)) return CF_ERR_ARG;
    return guard([&] {
        needFinished(bt);
        if (rowsCap < bt->rowsOut) throw ArgError("cf_batch_results_compact: the row buffer is smaller than cf_batch_num_rows");
        std::memcpy(rows, bt->hRows.p, bt->rowsOut * sizeof(OutRow));
        std::memcpy(nRows, bt->hNOut.p, bt->nQueries * 4);
        std::memcpy(score2, bt->hScore2.p, bt->nQueries * 4);
    });
}

cf_status cf_batch_timings(const cf_batch *bt, float ms[5]) {
    if (!bt || !ms) return CF_ERR_ARG;
    std::memcpy(ms, bt->ms, sizeof bt->ms);
    return CF_OK;
}
cf_status cf_batch_opcounts(cf_batch *bt, cf_opcounts *o) {
    if (!bt || !o) return CF_ERR_ARG;
    if (!bt->finished) { g_err = "the batch has not been classified"; return CF_ERR_ARG; }
    if (!bt->opsValid) {
        // The production kernels carry no counters: tally once with their instrumented builds.  The
        // work is a pure function of the batch, so the counts are those of the timed launches.  (The
        // search pass rewrites the hit lists of the batch; rows already in `out` are not touched.  A batch
        // whose row stage took several passes reports the walk steps of its last pass only.)
        cf_classifier *cl = bt->cl;
        const cf_status st = guard([&] {
            HIP_OK(hipSetDevice(cl->ix->device));
            HIP_OK(hipMemset(bt->cursor.p, 0, 32));
            HIP_OK(hipMemset(bt->ops.p, 0, sizeof(OpCounts)));
            if (bt->nReads) launchSearch(cl, bt, nullptr, 0, true);
            if (bt->nQueries && !bt->d.directRefs) launchWalk(cl, bt, nullptr, true);      // (direct references: no row was walked)
            HIP_OK(hipDeviceSynchronize());
            HIP_OK(hipGetLastError());
            HIP_OK(hipMemcpy(&bt->lastOps, bt->ops.p, sizeof(OpCounts), hipMemcpyDeviceToHost));
            bt->lastOps.nRows = bt->rowsTotal;
            bt->opsValid = true;
        });
        if (st != CF_OK) return st;
    }
    o->n_ftab = bt->lastOps.nFtab; o->n_pair = bt->lastOps.nPair; o->n_pair2 = bt->lastOps.nPair2;
    o->n_single = bt->lastOps.nSingle; o->n_walk = bt->lastOps.nWalk; o->n_rows = bt->lastOps.nRows;
    o->n_ftab_wide = bt->lastOps.nFtabWide; o->n_verify = bt->lastOps.nVerify; o->n_text_loads = bt->lastOps.nTextLoads;
    o->n_pos_hits = bt->lastOps.nPosHits;
    return CF_OK;
}

cf_status cf_counts_reset(cf_classifier *cl) {
    if (!cl) return CF_ERR_ARG;
    return guard([&] { HIP_OK(hipSetDevice(cl->ix->device)); HIP_OK(hipMemset(cl->counts.p, 0, cl->counts.bytes())); });
}
cf_status cf_counts_get(cf_classifier *cl, uint64_t *nReads, uint64_t *nUnique) {
    if (!cl || !nReads || !nUnique) return CF_ERR_ARG;
    return guard([&] {
        const size_t n = cl->ix->h.taxa.size();
        HIP_OK(hipSetDevice(cl->ix->device));
        HIP_OK(hipMemcpy(nReads, cl->counts.p, n * 8, hipMemcpyDeviceToHost));
        HIP_OK(hipMemcpy(nUnique, cl->counts.p + n, n * 8, hipMemcpyDeviceToHost));
    });
}
cf_status cf_counts_get_single(cf_classifier *cl, uint64_t *nSingle) {
    if (!cl || !nSingle) return CF_ERR_ARG;
    return guard([&] {
        const size_t n = cl->ix->h.taxa.size();
        HIP_OK(hipSetDevice(cl->ix->device));
        HIP_OK(hipMemcpy(nSingle, cl->counts.p + 2 * n, n * 8, hipMemcpyDeviceToHost));
    });
}
void *cf_counts_device(cf_classifier *cl) { return cl ? cl->counts.p : nullptr; }

// ---- RCCL, bound at first use (dlopen): the library itself carries no link-time RCCL dependency
extern "C++" {
namespace {
struct Rccl {
    // rccl.h: ncclResult_t f(...)
    int (*allReduce)(const void *, void *, size_t, int, int, void *, hipStream_t) = nullptr;
    int (*commInitAll)(void **, int, const int *) = nullptr;
    int (*commDestroy)(void *) = nullptr;
    int (*groupStart)() = nullptr;
    int (*groupEnd)() = nullptr;
    bool ok = false;
};
const Rccl &rccl() {
    static const Rccl r = [] {
        Rccl x;
        void *h = dlopen("librccl.so.1", RTLD_LAZY | RTLD_LOCAL);
        if (!h) h = dlopen("librccl.so", RTLD_LAZY | RTLD_LOCAL);
        if (!h) return x;
        x.allReduce = reinterpret_cast<decltype(x.allReduce)>(dlsym(h, "ncclAllReduce"));
        x.commInitAll = reinterpret_cast<decltype(x.commInitAll)>(dlsym(h, "ncclCommInitAll"));
        x.commDestroy = reinterpret_cast<decltype(x.commDestroy)>(dlsym(h, "ncclCommDestroy"));
        x.groupStart = reinterpret_cast<decltype(x.groupStart)>(dlsym(h, "ncclGroupStart"));
        x.groupEnd = reinterpret_cast<decltype(x.groupEnd)>(dlsym(h, "ncclGroupEnd"));
        x.ok = x.allReduce && x.commInitAll && x.commDestroy && x.groupStart && x.groupEnd;
        return x;
    }();
    return r;
}
}  // namespace
}  // extern "C++"

cf_status cf_counts_allreduce(cf_classifier *cl, void *comm, void *streamv) {
    if (!cl || !comm) return CF_ERR_ARG;
    if (!rccl().ok) { g_err = "librccl.so.1 could not be loaded"; return CF_ERR_HIP; }
    return guard([&] {
        HIP_OK(hipSetDevice(cl->ix->device));
        constexpr int kNcclUint64 = 5, kNcclSum = 0;          // rccl.h: ncclDataType_t / ncclRedOp_t
        const int rc = rccl().allReduce(cl->counts.p, cl->counts.p, 3 * cl->ix->h.taxa.size(), kNcclUint64, kNcclSum, comm,
                                        static_cast<hipStream_t>(streamv));
        if (rc != 0) throw HipError("ncclAllReduce failed with ncclResult_t " + std::to_string(rc));
    });
}

// One process, several GPUs (the C++ front end's --gpus N): the communicators of all devices at once and the
// all-reduce of every device's counters as one RCCL group.
cf_status cf_comm_init_all(int n, const int *devices, void **comms) {
    if (n < 1 || !devices || !comms) return CF_ERR_ARG;
    if (!rccl().ok) { g_err = "librccl.so.1 could not be loaded"; return CF_ERR_HIP; }
    return guard([&] {
        // (CF_TEST_FAIL_COMM=1, behind the knob gate: the communicators cannot be made — the drivers' error path, tests/test_gpu_cli.py)
        if (envInt("CF_TEST_FAIL_COMM", 0)) throw HipError("ncclCommInitAll failed with ncclResult_t 2 (CF_TEST_FAIL_COMM)");
        const int rc = rccl().commInitAll(comms, n, devices);
        if (rc != 0) throw HipError("ncclCommInitAll failed with ncclResult_t " + std::to_string(rc));
    });
}
void cf_comm_destroy(void *comm) { if (comm && rccl().ok) (void)rccl().commDestroy(comm); }

cf_status cf_counts_allreduce_group(cf_classifier *const *cls, void *const *comms, int n) {
    if (n < 1 || !cls || !comms) return CF_ERR_ARG;
    if (!rccl().ok) { g_err = "librccl.so.1 could not be loaded"; return CF_ERR_HIP; }
    return guard([&] {
        if (rccl().groupStart() != 0) throw HipError("ncclGroupStart failed");
        cf_status st = CF_OK;
        for (int i = 0; i < n && st == CF_OK; i++) st = cf_counts_allreduce(cls[i], comms[i], nullptr);
        const int rc = rccl().groupEnd();
        if (st != CF_OK) throw HipError(g_err);
        if (rc != 0) throw HipError("ncclGroupEnd failed with ncclResult_t " + std::to_string(rc));
        for (int i = 0; i < n; i++) { HIP_OK(hipSetDevice(cls[i]->ix->device)); HIP_OK(hipStreamSynchronize(nullptr)); }
    });
}

// HIP streams for callers that do not link the HIP runtime themselves (the C++ front end)
cf_status cf_stream_create(int device, void **stream) {
    if (!stream) return CF_ERR_ARG;
    *stream = nullptr;
    if (!haveDevice()) { g_err = "no HIP device visible"; return CF_ERR_NO_DEVICE; }
    return guard([&] {
        HIP_OK(hipSetDevice(device));
        hipStream_t s;
        HIP_OK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
        *stream = s;
    });
}
void cf_stream_destroy(void *stream) { if (stream) (void)hipStreamDestroy(static_cast<hipStream_t>(stream)); }
int cf_device_count(void) { int n = 0; return hipGetDeviceCount(&n) == hipSuccess ? n : 0; }

// NUMA node of a device's PCIe link, from sysfs (-1: unknown — no such device, a VM without the topology, one node)
int cf_device_numa_node(int device) {
    char bdf[64] = {0};
    if (device < 0 || device >= cf_device_count()) return -1;
    if (hipDeviceGetPCIBusId(bdf, (int)sizeof bdf - 1, device) != hipSuccess) { (void)hipGetLastError(); return -1; }      // (leave no error behind for the next call's check)
    for (char *c = bdf; *c; c++) *c = (char)std::tolower((unsigned char)*c);
    std::FILE *f = std::fopen((std::string("/sys/bus/pci/devices/") + bdf + "/numa_node").c_str(), "r");
    if (!f) return -1;
    int node = -1;
    if (std::fscanf(f, "%d", &node) != 1) node = -1;
    std::fclose(f);
    return node;
}

cf_status cf_thread_bind_near_device(int device, int *nodeOut) {
    if (nodeOut) *nodeOut = -1;
    const int node = cf_device_numa_node(device);
    if (node < 0) return CF_OK;
    std::FILE *f = std::fopen(("/sys/devices/system/node/node" + std::to_string(node) + "/cpulist").c_str(), "r");
    if (!f) return CF_OK;
    char buf[4096] = {0};
    const size_t got = std::fread(buf, 1, sizeof buf - 1, f);
    std::fclose(f);
    buf[got] = 0;
    cpu_set_t have, want;
    CPU_ZERO(&have); CPU_ZERO(&want);
    if (sched_getaffinity(0, sizeof have, &have) != 0) return CF_OK;
    int nWant = 0;
    for (const char *p = buf; *p;) {                          // "0-63,128-191"
        char *e = nullptr;
        const long lo = std::strtol(p, &e, 10);
        if (e == p) break;
        long hi = lo;
        p = e;
        if (*p == '-') { hi = std::strtol(p + 1, &e, 10); p = e; }
        for (long c = lo; c <= hi && c < CPU_SETSIZE; c++) if (c >= 0 && CPU_ISSET((int)c, &have)) { CPU_SET((int)c, &want); nWant++; }
        while (*p == ',' || *p == '\n' || *p == ' ') p++;
    }
    if (nWant == 0) return CF_OK;                             // the node's CPUs are not ours to run on: stay where we are
    if (sched_setaffinity(0, sizeof want, &want) != 0) return CF_OK;
    if (nodeOut) *nodeOut = node;
    return CF_OK;
}

// ---------------------------------------------------------------- debug taps
cf_status cf_debug_search(cf_classifier *cl, const uint8_t *seq, uint64_t len, cf_hit *hf, cf_hit *hr,
                          uint32_t maxHits, uint32_t nhits[2]) {
    if (!cl || !seq || !hf || !hr || !nhits) return CF_ERR_ARG;
    nhits[0] = nhits[1] = 0;
    const uint64_t off[2] = {0, len};
    const uint32_t seed = 0;
    cf_batch *bt = nullptr;
    cf_status st = cf_batch_create(cl, seq, off, &seed, 1, 0, &bt);
    if (st != CF_OK) return st;
    std::unique_ptr<cf_batch> own(bt);
    return guard([&] {
        BatchStatus hs{};
        HIP_OK(hipMemcpy(&hs, bt->st.p, sizeof hs, hipMemcpyDeviceToHost));
        if (hs.nItems == 0) return;                          // the read does not pass the filters
        HIP_OK(hipMemset(bt->cursor.p, 0, 32));
        HIP_OK(hipMemset(bt->ops.p, 0, sizeof(OpCounts)));
        bt->d.lazyHits = 0;                                  // the tap shows every hit of both strands
        launchSearch(cl, bt, nullptr, 1);
        hipLaunchKernelGGL(k_postfix_only, dim3(1), dim3(64), 0, 0, cl->ix->d, cl->d, bt->d);
        HIP_OK(hipDeviceSynchronize());
        HIP_OK(hipGetLastError());
        uint32_t n[2], cap = 0;
        HIP_OK(hipMemcpy(n, bt->nhml.p, 8, hipMemcpyDeviceToHost));
        n[0] &= 0x7fffffffu; n[1] &= 0x7fffffffu;              // (hits | has one of minHitLen << 31: nhml_make)
        HIP_OK(hipMemcpy(&cap, bt->hitCap.p, 4, hipMemcpyDeviceToHost));
        std::vector<HitP> all(2 * (size_t)cap);
        HIP_OK(hipMemcpy(all.data(), bt->hits.p, all.size() * sizeof(HitP), hipMemcpyDeviceToHost));
        cf_hit *o[2] = {hf, hr};
        for (int f = 0; f < 2; f++) {
            nhits[f] = n[f];
            for (uint32_t i = 0; i < n[f] && i < maxHits; i++) {
                const HitP &p = all[f * cap + i];                      // HitP layout (cf_kernels.hpp), unpacked on the host
                const uint64_t top = p.w0 & kHit40, size = p.w1 & kHit40;
                const bool dummy = top == kHit40 && size == 0;
                const uint32_t bw = (uint32_t)(p.w1 >> 40);
                o[f][i].top = dummy ? kNone64 : top; o[f][i].bot = dummy ? kNone64 : top + size;
                o[f][i].bwoff = bw == kHit24 ? kNone32 : bw; o[f][i].len = (uint32_t)(p.w0 >> 40);
            }
        }
    });
}

cf_status cf_debug_resolve(cf_index *ix, const uint64_t *rows, uint64_t n, uint32_t *refs) {
    if (!ix || !rows || !refs) return CF_ERR_ARG;
    if (ix->device < 0) return CF_ERR_NO_DEVICE;
    return guard([&] {
        HIP_OK(hipSetDevice(ix->device));
        DevBuf<uint64_t> r; DevBuf<uint32_t> o; DevBuf<unsigned long long> cur; DevBuf<BatchStatus> st;
        r.upload(std::vector<uint64_t>(rows, rows + n)); o.alloc(n); cur.alloc(4); st.alloc(1);
        HIP_OK(hipMemset(cur.p, 0, 32));
        BatchStatus hs{};
        hs.rowLo = 0; hs.rowHi = n;
        HIP_OK(hipMemcpy(st.p, &hs, sizeof hs, hipMemcpyHostToDevice));
        DBatch d{};
        d.rowVal = r.p; d.rowRef = o.p; d.cursor = cur.p; d.st = st.p;
        hipLaunchKernelGGL((k_walk2<2, false>), dim3(persistentBlocks(*ix, n, 4, 2)), dim3(256), 0, 0, ix->d, d);
        HIP_OK(hipDeviceSynchronize());
        HIP_OK(hipGetLastError());
        HIP_OK(hipMemcpy(refs, o.p, n * 4, hipMemcpyDeviceToHost));
    });
}

// the batch's prefix sums on their own (cf_scan.hpp): mode 0 = ceil(x / 32), 1 = x, 2 = 2x with the count of non-zero x
cf_status cf_debug_scan(int device, int mode, const uint32_t *in, uint64_t n, uint64_t *sums, uint32_t *counts) {
    if ((!in && n) || !sums || mode < 0 || mode > 2 || (mode == 2 && !counts)) return CF_ERR_ARG;
    if (!haveDevice()) return CF_ERR_NO_DEVICE;
    return guard([&] {
        HIP_OK(hipSetDevice(device));
        DevBuf<uint32_t> din, dc, tc; DevBuf<uint64_t> da, ta;
        din.alloc(n + 16); da.alloc(n + 1); dc.alloc(n + 1); ta.alloc(scan_tiles_for(n) + 1); tc.alloc(scan_tiles_for(n) + 1);
        if (n) HIP_OK(hipMemcpy(din.p, in, n * 4, hipMemcpyHostToDevice));
        if (mode == SCAN_WORDS) scan_enqueue<SCAN_WORDS>(din.p, n, da.p, dc.p, ta.p, tc.p, nullptr);
        else if (mode == SCAN_PLAIN) scan_enqueue<SCAN_PLAIN>(din.p, n, da.p, dc.p, ta.p, tc.p, nullptr);
        else scan_enqueue<SCAN_HITS>(din.p, n, da.p, dc.p, ta.p, tc.p, nullptr);
        HIP_OK(hipDeviceSynchronize());
        HIP_OK(hipGetLastError());
        HIP_OK(hipMemcpy(sums, da.p, (n + 1) * 8, hipMemcpyDeviceToHost));
        if (mode == SCAN_HITS) HIP_OK(hipMemcpy(counts, dc.p, (n + 1) * 4, hipMemcpyDeviceToHost));
    });
}

static cf_status debugRank(cf_index *ix, const uint8_t *chars, const uint64_t *rows, uint64_t n, uint64_t *out, int g) {
    if (!ix || !chars || !rows || !out || n == 0) return CF_ERR_ARG;
    if (ix->device < 0) return CF_ERR_NO_DEVICE;
    return guard([&] {
        HIP_OK(hipSetDevice(ix->device));
        DevBuf<uint8_t> c; DevBuf<uint64_t> r, o;
        c.upload(std::vector<uint8_t>(chars, chars + n)); r.upload(std::vector<uint64_t>(rows, rows + n)); o.alloc(n);
        const uint64_t threads = n * g;
        if (g == 8) hipLaunchKernelGGL(k_debug_rank<8>, dim3((int)((threads + 255) / 256)), dim3(256), 0, 0, ix->d, c.p, r.p, n, o.p);
        else hipLaunchKernelGGL(k_debug_rank<1>, dim3((int)((threads + 255) / 256)), dim3(256), 0, 0, ix->d, c.p, r.p, n, o.p);
        HIP_OK(hipDeviceSynchronize());
        HIP_OK(hipGetLastError());
        HIP_OK(hipMemcpy(out, o.p, n * 8, hipMemcpyDeviceToHost));
    });
}
cf_status cf_debug_rank(cf_index *ix, const uint8_t *chars, const uint64_t *rows, uint64_t n, uint64_t *out) {
    return debugRank(ix, chars, rows, n, out, 8);
}
cf_status cf_debug_rank1(cf_index *ix, const uint8_t *chars, const uint64_t *rows, uint64_t n, uint64_t *out) {
    return debugRank(ix, chars, rows, n, out, 1);
}

// Inverse BWT (see cf_restore.hpp): pass 1 (segment lengths + links), list ranking, pass 2 (characters).
cf_status cf_index_restore(cf_index *ix, uint8_t *packed, uint64_t nBytes) {
    if (!ix || !packed) return CF_ERR_ARG;
    if (ix->device < 0) return CF_ERR_NO_DEVICE;
    const uint64_t n = ix->h.g.len;
    if (nBytes < n / 4 + 1) { g_err = "cf_index_restore: the output buffer must hold len/4 + 1 bytes"; return CF_ERR_ARG; }
    return guard([&] {
        HIP_OK(hipSetDevice(ix->device));
        if (ix->text.p) {                                    // the text is already resident (text verification tables)
            HIP_OK(hipMemcpy(packed, ix->text.p, n / 4 + 1, hipMemcpyDeviceToHost));
            return;
        }
        if (!ix->sides.p) throw ArgError("cf_index_restore: the index was opened without its BWT sides in HBM (cf_index_options::sides = 1 keeps them)");
        DevBuf<uint32_t> text;
        restoreCore(*ix, text, nullptr, nullptr, 0, 0);
        HIP_OK(hipMemcpy(packed, text.p, n / 4 + 1, hipMemcpyDeviceToHost));
    });
}

cf_status cf_debug_random_read_gbps(cf_index *ix, uint64_t nLoads, int steps, double *gbps) {
    if (!ix || !gbps || steps < 1) return CF_ERR_ARG;
    if (ix->device < 0) return CF_ERR_NO_DEVICE;
    return guard([&] {
        HIP_OK(hipSetDevice(ix->device));
        DevBuf<unsigned long long> sink; sink.alloc(1);
        const uint64_t groups = std::max<uint64_t>(1, nLoads / (uint64_t)steps);
        const int blocks = (int)((groups * 8 + 255) / 256);
        hipEvent_t a, b;
        HIP_OK(hipEventCreate(&a)); HIP_OK(hipEventCreate(&b));
        // (the index's own sides; where they were dropped, the planes: 128-byte lines of the same kind of table, three times as many)
        const uint8_t *tab = ix->sides.p ? ix->sides.p : ix->planes.p;
        const uint64_t nLines = ix->sides.p ? ix->h.g.numSides : ix->h.g.numSides * 3;
        hipLaunchKernelGGL(k_random_sides, dim3(blocks), dim3(256), 0, 0, tab, nLines, 2u, 1ull, sink.p);
        HIP_OK(hipEventRecord(a, 0));
        hipLaunchKernelGGL(k_random_sides, dim3(blocks), dim3(256), 0, 0, tab, nLines, (uint32_t)steps, 7ull, sink.p);
        HIP_OK(hipEventRecord(b, 0));
        HIP_OK(hipEventSynchronize(b));
        float ms = 0;
        HIP_OK(hipEventElapsedTime(&ms, a, b));
        (void)hipEventDestroy(a); (void)hipEventDestroy(b);
        *gbps = (double)(blocks * 32ull * (uint64_t)steps) * 128.0 / (ms * 1e-3) / 1e9;
    });
}

}  // extern "C"
