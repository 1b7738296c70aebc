// cf_build.hip — GPU index construction: suffix array of the joined reference by
// chunked radix sort of 29-mer keys, tie refinement by further 29-mers for a bounded number of rounds and then by
// prefix doubling over an inverse suffix array (log rounds: repeat-rich references), then the BWT sides, the
// SA sample, ftab/eftab and the genome-boundary rows, written in the reference's
// on-disk format (Ebwt::buildToDisk bt2_idx.h:3377-3840; initFromVector :1249-1642).
// gfx950 only; C ABI in include/centrifuge_amd_build.h.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cstring>
#include <map>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/centrifuge_amd_build.h"
#include "cf_build_host.hpp"
#include "cf_knobs.hpp"
#include "cf_prims.hpp"

using namespace cfamd;

namespace {

thread_local std::string g_berr;
thread_local double g_btime[4] = {0, 0, 0, 0};

struct HipErr : std::runtime_error { using std::runtime_error::runtime_error; };
#define HIPB(expr)                                                                                 \
    do {                                                                                           \
        hipError_t e_ = (expr);                                                                    \
        if (e_ != hipSuccess) throw HipErr(std::string(#expr) + ": " + hipGetErrorString(e_));     \
    } while (0)

template <typename T>
struct Dev {
    T *p = nullptr;
    size_t n = 0;
    Dev() = default;
    Dev(const Dev &) = delete;
    Dev &operator=(const Dev &) = delete;
    ~Dev() { if (p) (void)hipFree(p); }
    void alloc(size_t count) {
        if (p) (void)hipFree(p);
        p = nullptr; n = 0;
        HIPB(hipMalloc(reinterpret_cast<void **>(&p), std::max<size_t>(1, count) * sizeof(T)));
        n = count;
    }
    void ensure(size_t count) { if (count > n) alloc(count + count / 16 + 64); }
    void release() { if (p) (void)hipFree(p); p = nullptr; n = 0; }
};

constexpr int kKeyChars = 29;                 // chars per 64-bit key: 58 bits + 6-bit end marker
constexpr int kBinChars = 12;                 // chunking granularity: 4^12 prefix bins
constexpr uint64_t kBins = 1ull << (2 * kBinChars);
constexpr uint64_t kRefOverlap = 11;          // bt2_idx.h:3504

// ------------------------------------------------------------- device helpers
struct Packed {
    const uint64_t *w;                        // big-endian 2-bit text: char i at bits 62-2(i&31) of word i>>5; tail = all ones
    uint64_t n;
};

// 32 chars starting at p, first char in the top bit pair; positions >= n read as T (3)
__device__ __forceinline__ uint64_t window(const Packed &t, uint64_t p) {
    if (p >= t.n) return ~0ull;
    const uint64_t k = p >> 5;
    const uint32_t s = 2u * (uint32_t)(p & 31);
    uint64_t v = t.w[k] << s;
    if (s) v |= t.w[k + 1] >> (64 - s);
    return v;
}

// Sort key of the suffix at p (terminator sorts AFTER every base): 29 chars padded
// with T, then 29 - min(29, remaining): of two suffixes with equal padded chars the
// one that ended earlier carries '$' where the other carries a real T, so it is larger.
__device__ __forceinline__ uint64_t sortKey(const Packed &t, uint64_t p) {
    const uint64_t rem = p < t.n ? t.n - p : 0;
    const uint64_t mark = rem >= (uint64_t)kKeyChars ? 0 : (uint64_t)kKeyChars - rem;
    return (window(t, p) & ~63ull) | mark;
}

__device__ __forceinline__ uint32_t charAt(const Packed &t, uint64_t p) {
    return (uint32_t)(t.w[p >> 5] >> (62 - 2 * (p & 31))) & 3u;
}

// ------------------------------------------------------------------- kernels
__global__ void __launch_bounds__(256) kb_pack(const uint8_t *text, uint64_t n, uint64_t *out, uint64_t nWords) {
    for (uint64_t w = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; w < nWords; w += (uint64_t)gridDim.x * blockDim.x) {
        uint64_t v = 0;
        const uint64_t base = w << 5;
        for (int j = 0; j < 32; j++) {
            const uint64_t i = base + j;
            const uint64_t c = i < n ? (uint64_t)(text[i] & 3) : 3ull;
            v |= c << (62 - 2 * j);
        }
        out[w] = v;
    }
}

__global__ void __launch_bounds__(256) kb_hist(Packed t, unsigned long long *bins) {
    for (uint64_t p = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; p <= t.n; p += (uint64_t)gridDim.x * blockDim.x)
        atomicAdd(&bins[window(t, p) >> (64 - 2 * kBinChars)], 1ull);
}

// suffixes whose prefix bin lies in [binLo, binHi): (key, position), unordered.  A block handles
// tiles of 256 x 16 positions and reserves its output range with ONE atomic per tile (a single
// global counter saturates near 10^8 atomics/s, so per-wave reservations made this pass the
// builder's bottleneck).
constexpr int kCollectPer = 16;
__global__ void __launch_bounds__(256) kb_collect(Packed t, uint64_t binLo, uint64_t binHi, uint64_t *keys, uint64_t *vals,
                                                   unsigned long long *counter) {
    __shared__ uint32_t scanTmp[4];
    __shared__ unsigned long long tileBase;
    const uint64_t tile = 256ull * kCollectPer;
    const uint64_t nTiles = (t.n + 1 + tile - 1) / tile;
    for (uint64_t ti = blockIdx.x; ti < nTiles; ti += gridDim.x) {
        uint64_t key[kCollectPer];
        uint32_t mask = 0, cnt = 0;
#pragma unroll
        for (int j = 0; j < kCollectPer; j++) {
            const uint64_t p = ti * tile + (uint64_t)j * 256 + threadIdx.x;
            key[j] = 0;
            if (p <= t.n) {
                key[j] = sortKey(t, p);
                const uint64_t bin = key[j] >> (64 - 2 * kBinChars);
                if (bin >= binLo && bin < binHi) { mask |= 1u << j; cnt++; }
            }
        }
        uint32_t off, total;
        block_exclusive_sum256(cnt, off, total, scanTmp);
        if (threadIdx.x == 0) tileBase = total ? atomicAdd(counter, (unsigned long long)total) : 0ull;
        __syncthreads();
        uint64_t at = tileBase + off;
#pragma unroll
        for (int j = 0; j < kCollectPer; j++)
            if ((mask >> j) & 1u) { keys[at] = key[j]; vals[at] = ti * tile + (uint64_t)j * 256 + threadIdx.x; at++; }
        __syncthreads();
    }
}

// after the first sort: mark members of tie groups and group heads
__global__ void __launch_bounds__(256) kb_flag_first(const uint64_t *keys, uint32_t n, uint32_t *tie, uint32_t *head) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint64_t k = keys[i];
    const bool eqPrev = i > 0 && keys[i - 1] == k;
    const bool eqNext = i + 1 < n && keys[i + 1] == k;
    tie[i] = (eqPrev || eqNext) ? 1u : 0u;
    head[i] = (!eqPrev && eqNext) ? 1u : 0u;
}

// compact tie members: slot j gets (position, destination row in the chunk, head flag)
__global__ void __launch_bounds__(256) kb_compact_first(const uint64_t *vals, uint32_t n, const uint32_t *tie, const uint32_t *tieIdx,
                                                         const uint32_t *head, uint64_t *tpos, uint32_t *tdst, uint32_t *thead) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n || !tie[i]) return;
    const uint32_t j = tieIdx[i];
    tpos[j] = vals[i]; tdst[j] = i; thead[j] = head[i];
}

__global__ void __launch_bounds__(256) kb_iota(uint32_t *v, uint32_t m) {
    const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j < m) v[j] = j;
}
// out[i] = src[idx[i]]
template <typename T>
__global__ void __launch_bounds__(256) kb_gather(const T *src, const uint32_t *idx, uint32_t m, T *out) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < m) out[i] = src[idx[i]];
}

__global__ void __launch_bounds__(256) kb_round_keys(Packed t, const uint64_t *tpos, uint32_t m, uint64_t depth, uint64_t *tkey) {
    const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j < m) tkey[j] = sortKey(t, tpos[j] + depth);
}

// after a segmented sort: write the slots back to the chunk SA and flag the ties that remain
__global__ void __launch_bounds__(256) kb_round_flag(const uint64_t *tkey, const uint64_t *tpos, const uint32_t *tdst, const uint32_t *thead,
                                                      uint32_t m, uint64_t *sa, uint32_t *tie, uint32_t *head) {
    const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= m) return;
    sa[tdst[j]] = tpos[j];
    const uint64_t k = tkey[j];
    const bool eqPrev = !thead[j] && tkey[j - 1] == k;            // slot 0 is always a head
    const bool eqNext = j + 1 < m && !thead[j + 1] && tkey[j + 1] == k;
    tie[j] = (eqPrev || eqNext) ? 1u : 0u;
    head[j] = (!eqPrev && eqNext) ? 1u : 0u;
}

__global__ void __launch_bounds__(256) kb_round_compact(const uint64_t *tpos, const uint32_t *tdst, uint32_t m, const uint32_t *tie,
                                                         const uint32_t *tieIdx, const uint32_t *head,
                                                         uint64_t *tpos2, uint32_t *tdst2, uint32_t *thead2) {
    const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= m || !tie[j]) return;
    const uint32_t q = tieIdx[j];
    tpos2[q] = tpos[j]; tdst2[q] = tdst[j]; thead2[q] = head[j];
}

// ---- prefix doubling over the suffixes the 29-mer rounds left tied (buildOnGpu, phase 2)
// The inverse suffix array holds, for every text position, the row of its suffix — final for a resolved suffix, the first
// row of its tie group for one that is still tied — in 40 bits: a u32 array and, for texts of 2^32 rows or more, a u8 one.
struct Isa {
    uint32_t *lo;
    uint8_t *hi;                  // nullptr: every row fits 32 bits
};
__device__ __forceinline__ void isaPut(const Isa &r, uint64_t pos, uint64_t row) {
    r.lo[pos] = (uint32_t)row;
    if (r.hi) r.hi[pos] = (uint8_t)(row >> 32);
}
__device__ __forceinline__ uint64_t isaGet(const Isa &r, uint64_t pos) {
    return (uint64_t)r.lo[pos] | (r.hi ? (uint64_t)r.hi[pos] << 32 : 0ull);
}
constexpr uint64_t kSkipRow = ~0ull;          // chunk SA entry of a row whose suffix went to the leftover list

// the tied slots a chunk is left with after its 29-mer rounds go to the leftover list; their rows are emitted at the very end
__global__ void __launch_bounds__(256) kb_spill(const uint64_t *tpos, const uint32_t *tdst, const uint32_t *thead, uint32_t m, uint64_t rowBase,
                                                 uint64_t *sa, uint64_t *lpos, uint64_t *lrow, uint32_t *lhead) {
    const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= m) return;
    lpos[j] = tpos[j]; lrow[j] = rowBase + tdst[j]; lhead[j] = thead[j];
    sa[tdst[j]] = kSkipRow;
}
// headRow[g] = row of the first slot of group g (gid = inclusive sum of the head flags: 1-based)
__global__ void __launch_bounds__(256) kb_head_rows(const uint64_t *lrow, const uint32_t *lhead, const uint32_t *gid, uint32_t m, uint64_t *headRow) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < m && lhead[i]) headRow[gid[i] - 1] = lrow[i];
}
__global__ void __launch_bounds__(256) kb_isa_groups(Isa r, const uint64_t *lpos, const uint32_t *gid, const uint64_t *headRow, uint32_t m) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < m) isaPut(r, lpos[i], headRow[gid[i] - 1]);
}
// sort key of a doubling round: the rank of the suffix h further on (a suffix that ends before that is alone in its group by
// now — its end marker set it apart — and may take any key)
__global__ void __launch_bounds__(256) kb_double_keys(Isa r, const uint64_t *lpos, uint32_t m, uint64_t h, uint64_t n, uint64_t *key) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= m) return;
    const uint64_t p = lpos[i] + h;
    key[i] = isaGet(r, p < n ? p : n);
}
// after the two sorts: a slot starts a group where it did before or where its key differs from the slot before it
__global__ void __launch_bounds__(256) kb_double_heads(const uint64_t *key, const uint32_t *oldHead, uint32_t m, uint32_t *newHead, unsigned long long *ties) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t hd = (i >= m || i == 0 || oldHead[i] || key[i - 1] != key[i]) ? 1u : 0u;
    if (i < m) newHead[i] = hd;
    const unsigned long long tied = __ballot(!hd);           // one atomic per wavefront
    if ((threadIdx.x & 63) == 0 && tied) atomicAdd(ties, (unsigned long long)__popcll(tied));
}

struct EmitArgs {
    Packed t;
    const uint64_t *sa;           // chunk SA (kb_emit) / positions of the leftover list (kb_emit_list)
    const uint64_t *rows;         // kb_emit_list: the rows that go with them
    uint64_t rowBase;
    uint32_t count;
    uint8_t *bwt;                 // one BWT char per row
    void *sample; int sampleWide; int offRate;
    const uint64_t *fragStart; const uint32_t *fragSeq; uint32_t nFrag;
    const uint64_t *marks; const uint32_t *markRef; uint32_t nMarks;
    uint64_t *boundRow; uint32_t *boundRef; unsigned long long *boundCount;
    uint64_t *shortRow; uint32_t nShort;      // rows of the suffixes with fewer than ftabChars chars: index = n - pos
    uint64_t *zOff;
    Isa isa; int haveIsa;         // inverse suffix array for the doubling rounds (phase 2), when it is kept
};

// everything the index keeps of one suffix-array row
__device__ __forceinline__ void emitRow(const EmitArgs &a, uint64_t pos, uint64_t row) {
    const uint64_t n = a.t.n;
    // BWT char: the '$' row is stored as an A and remembered as zOff (bt2_idx.h:3571-3584)
    if (pos == 0) { a.bwt[row] = 0; *a.zOff = row; }
    else a.bwt[row] = (uint8_t)charAt(a.t, pos - 1);
    // SA sample = reference index of text offset pos + 11 (bt2_idx.h:3640-3669)
    if ((row & ((1ull << a.offRate) - 1)) == 0) {
        uint32_t tidx = 0;
        if (pos > 0) {
            uint64_t adj = pos + kRefOverlap;
            if (adj >= n) adj = pos;
            if (adj >= n) adj--;
            uint32_t lo = 0, hi = a.nFrag;                 // last fragment starting at or before adj
            while (hi - lo > 1) { const uint32_t md = (lo + hi) >> 1; if (a.fragStart[md] <= adj) lo = md; else hi = md; }
            tidx = a.fragSeq[lo];
        }
        const uint64_t e = row >> a.offRate;
        if (a.sampleWide) static_cast<uint32_t *>(a.sample)[e] = tidx;
        else static_cast<uint16_t *>(a.sample)[e] = (uint16_t)tidx;
    }
    // genome-boundary rows (.4.cf, bt2_idx.h:3562-3567)
    {
        uint32_t lo = 0, hi = a.nMarks;
        while (lo < hi) { const uint32_t md = (lo + hi) >> 1; if (a.marks[md] < pos) lo = md + 1; else hi = md; }
        if (lo < a.nMarks && a.marks[lo] == pos) {
            const unsigned long long at = atomicAdd(a.boundCount, 1ull);
            a.boundRow[at] = row; a.boundRef[at] = a.markRef[lo];
        }
    }
    if (n - pos < a.nShort) a.shortRow[n - pos] = row;
}

__global__ void __launch_bounds__(256) kb_emit(EmitArgs a) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= a.count) return;
    const uint64_t pos = a.sa[i], row = a.rowBase + i;
    if (pos == kSkipRow) return;                           // still tied: emitted from the leftover list (kb_emit_list)
    emitRow(a, pos, row);
    if (a.haveIsa) isaPut(a.isa, pos, row);
}
__global__ void __launch_bounds__(256) kb_emit_list(EmitArgs a) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < a.count) emitRow(a, a.sa[i], a.rows[i]);
}

// one thread per 32-char word of a side: pack little-end-first (char j at bits 2j..2j+1,
// bt2_idx.h:3688-3707) and count the chars ('$' row not counted, padding counted as A)
__global__ void __launch_bounds__(256) kb_side_words(const uint8_t *bwt, uint64_t nWords, uint64_t zOff, uint8_t *sides, uint32_t *wordCnt) {
    const uint64_t w = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
    if (w >= nWords) return;
    const uint64_t row0 = w * 32;
    const ulonglong2 *src = reinterpret_cast<const ulonglong2 *>(bwt + row0);
    const ulonglong2 a = src[0], b = src[1];
    const uint64_t q[4] = {a.x, a.y, b.x, b.y};
    uint64_t v = 0;
    uint32_t c[4] = {0, 0, 0, 0};
    for (int j = 0; j < 32; j++) {
        const uint32_t ch = (uint32_t)(q[j >> 3] >> (8 * (j & 7))) & 3u;
        v |= (uint64_t)ch << (2 * j);
        if (row0 + j != zOff) c[ch]++;
    }
    const uint64_t side = w / 12, k = w % 12;
    *reinterpret_cast<uint64_t *>(sides + side * 128 + 8 * k) = v;
    wordCnt[w] = c[0] | (c[1] << 8) | (c[2] << 16) | (c[3] << 24);
}

__global__ void __launch_bounds__(256) kb_side_counts(const uint32_t *wordCnt, uint64_t nSides, unsigned long long *cnt /* 4 x nSides */) {
    const uint64_t s = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
    if (s >= nSides) return;
    uint32_t c[4] = {0, 0, 0, 0};
    for (int k = 0; k < 12; k++) {
        const uint32_t x = wordCnt[s * 12 + k];
        c[0] += x & 255; c[1] += (x >> 8) & 255; c[2] += (x >> 16) & 255; c[3] += x >> 24;
    }
    for (int ch = 0; ch < 4; ch++) cnt[(uint64_t)ch * nSides + s] = c[ch];
}

__global__ void __launch_bounds__(256) kb_side_occ(const unsigned long long *occ /* 4 x nSides, exclusive */, uint64_t nSides, uint8_t *sides) {
    const uint64_t s = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
    if (s >= nSides) return;
    uint64_t *o = reinterpret_cast<uint64_t *>(sides + s * 128 + 96);
    for (int ch = 0; ch < 4; ch++) o[ch] = occ[(uint64_t)ch * nSides + s];
}

inline int blocksFor(uint64_t n, int per = 256) { return (int)std::min<uint64_t>((n + per - 1) / per, 1u << 20); }   // grid-stride kernels
inline unsigned blocksExact(uint64_t n) { return (unsigned)((n + 255) / 256); }                                       // one thread per element

double now() {
    return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

struct BuildOut {
    std::vector<uint8_t> sides, sample;
    std::vector<uint64_t> ftab, eftab;
    uint64_t zOff = 0, fchr[5] = {0, 0, 0, 0, 0};
    std::vector<std::pair<uint64_t, uint32_t>> bounds;
};

void buildOnGpu(const JoinedRef &ref, int offRate, int ftabChars, uint64_t chunkMax, bool verbose, BuildOut &out) {
    const uint64_t n = ref.len;
    const uint64_t nWords = (n + 31) / 32 + 2;
    Dev<uint64_t> dPacked;
    dPacked.alloc(nWords);
    {
        Dev<uint8_t> dText;
        dText.alloc(n);
        HIPB(hipMemcpy(dText.p, ref.text, n, hipMemcpyHostToDevice));
        hipLaunchKernelGGL(kb_pack, dim3(blocksFor(nWords)), dim3(256), 0, 0, dText.p, n, dPacked.p, nWords);
        HIPB(hipDeviceSynchronize());
    }
    const Packed t{dPacked.p, n};

    // ---- prefix-bin histogram -> chunk plan, ftab counts
    std::vector<unsigned long long> bins(kBins);
    {
        Dev<unsigned long long> dBins;
        dBins.alloc(kBins);
        HIPB(hipMemset(dBins.p, 0, kBins * 8));
        hipLaunchKernelGGL(kb_hist, dim3(blocksFor(n + 1, 256 * 16)), dim3(256), 0, 0, t, dBins.p);
        HIPB(hipMemcpy(bins.data(), dBins.p, kBins * 8, hipMemcpyDeviceToHost));
    }
    struct Chunk { uint64_t lo, hi, count; };
    std::vector<Chunk> chunks;
    {
        uint64_t lo = 0, cnt = 0;
        for (uint64_t b = 0; b < kBins; b++) {
            if (cnt > 0 && cnt + bins[b] > chunkMax) { chunks.push_back({lo, b, cnt}); lo = b; cnt = 0; }
            cnt += bins[b];
        }
        chunks.push_back({lo, kBins, cnt});
    }
    uint64_t maxCount = 0;
    for (const auto &c : chunks) maxCount = std::max(maxCount, c.count);
    if (maxCount >= 0x7fffff00ull) throw std::runtime_error("a prefix bin holds too many suffixes for one GPU pass (highly repetitive reference)");
    if (verbose) std::fprintf(stderr, "[cf-build] n=%llu, %zu chunk(s), largest %llu suffixes\n", (unsigned long long)n, chunks.size(), (unsigned long long)maxCount);

    // ---- device buffers
    const uint64_t numSides = ((n / 4 + 1) + 95) / 96;
    const uint64_t bwtRows = numSides * 384;
    const bool wide = ref.nPat > 65535;                     // bt2_idx.h:1322
    const uint64_t offsLen = (n + 1 + (1ull << offRate) - 1) >> offRate;
    Dev<uint8_t> dBwt, dSample;
    dBwt.alloc(bwtRows + 64);
    HIPB(hipMemset(dBwt.p, 0, bwtRows + 64));
    dSample.alloc(offsLen * (wide ? 4 : 2) + 8);
    Dev<uint64_t> kIn, kOut, vIn, vOut, tposB, dFragStart, dMarks, dBoundRow, dShortRow, dZoff;
    Dev<uint32_t> tie, tieIdx, head, tdstA, tdstB, theadA, theadB, gid, idxA, idxB, dFragSeq, dMarkRef, dBoundRef, isaLo;
    Dev<unsigned long long> dCounter, dBoundCount;
    Dev<uint8_t> tmp, isaHi;
    kIn.alloc(maxCount); kOut.alloc(maxCount); vIn.alloc(maxCount); vOut.alloc(maxCount);
    tie.alloc(maxCount + 1); tieIdx.alloc(maxCount + 1); head.alloc(maxCount + 1);
    dCounter.alloc(1); dBoundCount.alloc(1); dZoff.alloc(1);
    HIPB(hipMemset(dBoundCount.p, 0, 8));
    // fragment table and boundary marks (bt2_idx.h:3499-3535)
    {
        std::vector<uint64_t> fs(ref.nFrag);
        std::vector<uint32_t> fq(ref.nFrag);
        for (uint64_t i = 0; i < ref.nFrag; i++) { fs[i] = ref.rstarts[3 * i]; fq[i] = (uint32_t)ref.rstarts[3 * i + 1]; }
        dFragStart.alloc(fs.size()); dFragSeq.alloc(fq.size());
        HIPB(hipMemcpy(dFragStart.p, fs.data(), fs.size() * 8, hipMemcpyHostToDevice));
        HIPB(hipMemcpy(dFragSeq.p, fq.data(), fq.size() * 4, hipMemcpyHostToDevice));
        std::map<uint64_t, uint32_t> marks;                 // later sequences overwrite (std::map assignment)
        for (uint64_t i = 0; i < ref.nPat; i++) {
            const uint64_t ro = ref.seqJoinedStart[i];
            marks[ro < kRefOverlap ? 0 : ro - kRefOverlap] = (uint32_t)i;
        }
        std::vector<uint64_t> mk; std::vector<uint32_t> mr;
        for (const auto &kv : marks) { mk.push_back(kv.first); mr.push_back(kv.second); }
        dMarks.alloc(mk.size()); dMarkRef.alloc(mr.size());
        HIPB(hipMemcpy(dMarks.p, mk.data(), mk.size() * 8, hipMemcpyHostToDevice));
        HIPB(hipMemcpy(dMarkRef.p, mr.data(), mr.size() * 4, hipMemcpyHostToDevice));
        dBoundRow.alloc(mk.size()); dBoundRef.alloc(mk.size());
    }
    const uint32_t nShort = (uint32_t)std::min<uint64_t>((uint64_t)ftabChars, n + 1);   // suffixes n-ftabChars+1 .. n
    dShortRow.alloc(nShort);

    size_t tmpBytes = 0;
    {
        size_t b1 = 0, b2 = 0, b3 = 0, b4 = 0;
        HIPB(sort_pairs(nullptr, b1, kIn.p, kOut.p, vIn.p, vOut.p, maxCount, 0, 64));
        b2 = device_scan_bytes<uint32_t>(maxCount + 1);
        HIPB(sort_pairs(nullptr, b3, kIn.p, kOut.p, tie.p, tieIdx.p, maxCount, 0, 64));
        HIPB(sort_pairs(nullptr, b4, tie.p, tieIdx.p, tie.p, tieIdx.p, maxCount, 0, 32));
        tmpBytes = std::max(std::max(b1, b2), std::max(b3, b4));
    }
    tmp.alloc(tmpBytes);

    // ---- the two ways out of a tie.  Suffixes that tie on their first 29 bases are refined 29 bases at a time
    // (phase 1: the key of a tied suffix at p becomes the 29-mer at p + depth; the tied slots are sorted by (group, key) as TWO
    // plain radix sorts — by key, then stably by group — so that one giant group (a homopolymer tract) or hundreds of millions
    // of tiny ones (strain clusters) cost the same per element).  That alone needs LCP / 29 rounds, and repeat-rich references
    // (strains 0.1 % apart, shared operons of kilobases) hold LCPs in the thousands.  So a chunk runs a bounded number of such
    // rounds, and what is still tied then — a small fraction — goes to a leftover list that is finished AFTER the last chunk by
    // prefix doubling (phase 2): with the row of every suffix at hand (the inverse suffix array `isa`: the final row of a
    // resolved suffix, the first row of its group for a tied one), a group that agrees on its first h bases is ordered by
    // the rows of the suffixes h further on, which doubles h per round (Larsson-Sadakane).  The inverse suffix array costs
    // 4-5 bytes per base; a reference too large for it (beyond ~40 Gbp) stays with phase 1 alone, as before.
    const char *er = cfamd::cf_knob("CF_BUILD_ROUNDS");
    const uint32_t maxRounds1 = er ? (uint32_t)std::max(1, std::atoi(er)) : 24u;          // 29-mer rounds before a chunk may spill
    const bool wideRows = n + 1 > 0xffffffffull;
    bool haveIsa = false;
    {
        size_t freeB = 0, totalB = 0;
        HIPB(hipMemGetInfo(&freeB, &totalB));
        const char *ed = cfamd::cf_knob("CF_BUILD_DOUBLING");
        const uint64_t need = (n + 2) * (wideRows ? 5ull : 4ull);
        // beside it the refinement buffers of a chunk (40 bytes per tied slot) and the leftover list must still fit
        haveIsa = (!ed || std::atoi(ed) != 0) && need + 44ull * maxCount + (12ull << 30) < freeB;
        if (haveIsa) {
            isaLo.alloc(n + 2);
            if (wideRows) isaHi.alloc(n + 2);
        }
        if (verbose) std::fprintf(stderr, "[cf-build] inverse suffix array for prefix doubling: %s (%.1f GB of %.1f GB free)\n", haveIsa ? "kept" : "not kept",
                                  need / 1e9, freeB / 1e9);
    }
    const Isa isa{isaLo.p, wideRows ? isaHi.p : nullptr};
    struct Leftover { Dev<uint64_t> pos, row; Dev<uint32_t> head; uint32_t m = 0; uint64_t depth = 0; };
    std::vector<std::unique_ptr<Leftover>> leftovers;

    EmitArgs ea{};
    ea.t = t; ea.bwt = dBwt.p;
    ea.sample = dSample.p; ea.sampleWide = wide ? 1 : 0; ea.offRate = offRate;
    ea.fragStart = dFragStart.p; ea.fragSeq = dFragSeq.p; ea.nFrag = (uint32_t)ref.nFrag;
    ea.marks = dMarks.p; ea.markRef = dMarkRef.p; ea.nMarks = (uint32_t)dMarks.n;
    ea.boundRow = dBoundRow.p; ea.boundRef = dBoundRef.p; ea.boundCount = dBoundCount.p;
    ea.shortRow = dShortRow.p; ea.nShort = nShort; ea.zOff = dZoff.p;
    ea.isa = isa; ea.haveIsa = haveIsa ? 1 : 0;

    auto bitsFor = [](uint64_t v) { int b = 1; while (b < 64 && (v >> b)) b++; return b; };
    // slots 0 .. m-1 hold (key[j], grp[j]) with the groups contiguous and ascending: perm = the slots in (group, key) order.
    // keyOut / scratch arrays are caller-provided; returns the array that holds the permutation (permA or permB).
    auto sortByGroupAndKey = [&](const uint64_t *key, uint64_t *keyScratch, int keyBits, const uint32_t *grp, uint32_t nGroups, uint32_t m,
                                 uint32_t *permA, uint32_t *permB, uint32_t *g2a, uint32_t *g2b) -> uint32_t * {
        hipLaunchKernelGGL(kb_iota, dim3(blocksExact(m)), dim3(256), 0, 0, permA, m);
        size_t sb = tmp.n;
        HIPB(sort_pairs(tmp.p, sb, key, keyScratch, permA, permB, m, 0, keyBits));
        hipLaunchKernelGGL(kb_gather<uint32_t>, dim3(blocksExact(m)), dim3(256), 0, 0, grp, permB, m, g2a);
        sb = tmp.n;
        HIPB(sort_pairs(tmp.p, sb, g2a, g2b, permB, permA, m, 0, bitsFor(nGroups)));
        return permA;
    };

    double tCollect = 0, tSort = 0, tRefine = 0, tEmit = 0;
    auto lap = [&](double &acc, double &t0) { if (verbose) { (void)hipDeviceSynchronize(); const double t = now(); acc += t - t0; t0 = t; } };
    uint64_t rowBase = 0;
    for (size_t ci = 0; ci < chunks.size(); ci++) {
        const Chunk &c = chunks[ci];
        if (c.count == 0) continue;
        const uint32_t cnt = (uint32_t)c.count;
        double tl = verbose ? now() : 0;
        HIPB(hipMemset(dCounter.p, 0, 8));
        hipLaunchKernelGGL(kb_collect, dim3(8192), dim3(256), 0, 0, t, c.lo, c.hi, kIn.p, vIn.p, dCounter.p);
        lap(tCollect, tl);
        size_t tb = tmp.n;
        HIPB(sort_pairs(tmp.p, tb, kIn.p, kOut.p, vIn.p, vOut.p, cnt, 0, 64));
        lap(tSort, tl);
        // ---- tie groups of the first sort
        hipLaunchKernelGGL(kb_flag_first, dim3(blocksExact(cnt)), dim3(256), 0, 0, kOut.p, cnt, tie.p, head.p);
        HIPB(hipMemsetAsync(tie.p + cnt, 0, 4));
        HIPB((device_scan<uint32_t, false>(tmp.p, tie.p, tieIdx.p, (uint64_t)cnt + 1)));
        uint32_t m = 0;
        HIPB(hipMemcpy(&m, tieIdx.p + cnt, 4, hipMemcpyDeviceToHost));
        uint64_t *sa = vOut.p;
        if (m) {
            // the first sort's inputs are free now: they hold the tied slots' positions and keys (tpos = vIn, keys = kIn / kOut)
            tposB.ensure(m); tdstA.ensure(m); tdstB.ensure(m); theadA.ensure(m + 1); theadB.ensure(m + 1);
            gid.ensure(m + 1); idxA.ensure(m); idxB.ensure(m);
            uint64_t *tpos = vIn.p, *tposAlt = tposB.p;
            hipLaunchKernelGGL(kb_compact_first, dim3(blocksExact(cnt)), dim3(256), 0, 0, vOut.p, cnt, tie.p, tieIdx.p, head.p, tpos, tdstA.p, theadA.p);
            uint64_t depth = 0;
            uint32_t *tdst = tdstA.p, *tdstAlt = tdstB.p, *thead = theadA.p, *theadAlt = theadB.p;
            uint32_t rounds = 0;
            while (m) {
                if (haveIsa && rounds >= maxRounds1) break;                 // the rest of this chunk is left to the doubling rounds
                depth += kKeyChars;
                if (depth > n + kKeyChars) throw std::runtime_error("suffix refinement did not converge");
                // group of every slot (1-based inclusive count of the head flags) and their number
                HIPB((device_scan<uint32_t, true>(tmp.p, thead, gid.p, m)));
                uint32_t nGroups = 0;
                HIPB(hipMemcpy(&nGroups, gid.p + (m - 1), 4, hipMemcpyDeviceToHost));
                hipLaunchKernelGGL(kb_round_keys, dim3(blocksExact(m)), dim3(256), 0, 0, t, tpos, m, depth, kIn.p);
                const uint32_t *perm = sortByGroupAndKey(kIn.p, kOut.p, 64, gid.p, nGroups, m, idxA.p, idxB.p, tie.p, tieIdx.p);
                hipLaunchKernelGGL(kb_gather<uint64_t>, dim3(blocksExact(m)), dim3(256), 0, 0, tpos, perm, m, tposAlt);
                hipLaunchKernelGGL(kb_gather<uint64_t>, dim3(blocksExact(m)), dim3(256), 0, 0, kIn.p, perm, m, kOut.p);
                std::swap(tpos, tposAlt);                                   // positions and keys (kOut) in slot order now
                hipLaunchKernelGGL(kb_round_flag, dim3(blocksExact(m)), dim3(256), 0, 0, kOut.p, tpos, tdst, thead, m, sa, tie.p, head.p);
                HIPB(hipMemsetAsync(tie.p + m, 0, 4));
                HIPB((device_scan<uint32_t, false>(tmp.p, tie.p, tieIdx.p, (uint64_t)m + 1)));
                uint32_t m2 = 0;
                HIPB(hipMemcpy(&m2, tieIdx.p + m, 4, hipMemcpyDeviceToHost));
                if (m2) {
                    hipLaunchKernelGGL(kb_round_compact, dim3(blocksExact(m)), dim3(256), 0, 0, tpos, tdst, m, tie.p, tieIdx.p, head.p,
                                       tposAlt, tdstAlt, theadAlt);
                    std::swap(tpos, tposAlt); std::swap(tdst, tdstAlt); std::swap(thead, theadAlt);
                }
                m = m2;
                rounds++;
            }
            if (m) {                                                        // spill: these rows are emitted after the doubling rounds
                auto lo = std::make_unique<Leftover>();
                lo->pos.alloc(m); lo->row.alloc(m); lo->head.alloc(m);
                lo->m = m; lo->depth = depth + kKeyChars;                   // its groups agree on that many bases
                hipLaunchKernelGGL(kb_spill, dim3(blocksExact(m)), dim3(256), 0, 0, tpos, tdst, thead, m, rowBase, sa, lo->pos.p, lo->row.p, lo->head.p);
                leftovers.push_back(std::move(lo));
            }
            if (verbose) std::fprintf(stderr, "[cf-build] chunk %zu/%zu: %u suffixes, %u refinement round(s), %u left tied\n", ci + 1, chunks.size(), cnt, rounds, m);
        }
        lap(tRefine, tl);
        ea.sa = sa; ea.rows = nullptr; ea.rowBase = rowBase; ea.count = cnt;
        hipLaunchKernelGGL(kb_emit, dim3(blocksExact(cnt)), dim3(256), 0, 0, ea);
        HIPB(hipDeviceSynchronize());
        lap(tEmit, tl);
        rowBase += cnt;
    }
    if (rowBase != n + 1) throw std::runtime_error("internal: suffix count mismatch");
    // release the sort workspace before the doubling rounds and the side pass
    kIn.release(); kOut.release(); vIn.release(); vOut.release(); tposB.release();
    tie.release(); tieIdx.release(); head.release(); tdstA.release(); tdstB.release(); theadA.release(); theadB.release();
    gid.release(); idxA.release(); idxB.release();

    // ---- phase 2: prefix doubling over the leftover list
    double tDouble = 0;
    if (!leftovers.empty()) {
        double tl = now();
        uint64_t M64 = 0, h = ~0ull;
        for (const auto &lo : leftovers) { M64 += lo->m; h = std::min(h, lo->depth); }
        if (M64 >= 0x7fffff00ull) throw std::runtime_error("too many suffixes still tied after the 29-mer rounds for one doubling pass (raise CF_BUILD_ROUNDS)");
        const uint32_t M = (uint32_t)M64;
        Dev<uint64_t> lpos, lposAlt, lrow, key, keyAlt, headRow;
        Dev<uint32_t> lhead, lheadAlt, lgid, pA, pB, g2a, g2b;
        Dev<unsigned long long> dTies;
        lpos.alloc(M); lposAlt.alloc(M); lrow.alloc(M); key.alloc(M); keyAlt.alloc(M); headRow.alloc(M);
        lhead.alloc(M); lheadAlt.alloc(M); lgid.alloc(M); pA.alloc(M); pB.alloc(M); g2a.alloc(M); g2b.alloc(M); dTies.alloc(1);
        {   // the chunks' lists one after the other: ascending rows, groups contiguous
            uint64_t at = 0;
            for (auto &lo : leftovers) {
                HIPB(hipMemcpy(lpos.p + at, lo->pos.p, (size_t)lo->m * 8, hipMemcpyDeviceToDevice));
                HIPB(hipMemcpy(lrow.p + at, lo->row.p, (size_t)lo->m * 8, hipMemcpyDeviceToDevice));
                HIPB(hipMemcpy(lhead.p + at, lo->head.p, (size_t)lo->m * 4, hipMemcpyDeviceToDevice));
                at += lo->m;
                lo.reset();
            }
        }
        size_t b1 = 0, b2 = 0, b3 = 0;
        HIPB(sort_pairs(nullptr, b1, key.p, keyAlt.p, pA.p, pB.p, M, 0, 64));
        HIPB(sort_pairs(nullptr, b2, g2a.p, g2b.p, pA.p, pB.p, M, 0, 32));
        b3 = device_scan_bytes<uint32_t>(M);
        if (std::max(b1, std::max(b2, b3)) > tmp.n) tmp.alloc(std::max(b1, std::max(b2, b3)));
        const int rowBits = bitsFor(n + 1);
        uint32_t *hd = lhead.p, *hdAlt = lheadAlt.p;
        uint64_t *pos = lpos.p, *posAlt = lposAlt.p;
        uint32_t nGroups = 0;
        // group ids and the inverse suffix array entries of the tied suffixes: the first row of their group
        auto regroup = [&] {
            HIPB((device_scan<uint32_t, true>(tmp.p, hd, lgid.p, M)));
            HIPB(hipMemcpy(&nGroups, lgid.p + (M - 1), 4, hipMemcpyDeviceToHost));
            hipLaunchKernelGGL(kb_head_rows, dim3(blocksExact(M)), dim3(256), 0, 0, lrow.p, hd, lgid.p, M, headRow.p);
            hipLaunchKernelGGL(kb_isa_groups, dim3(blocksExact(M)), dim3(256), 0, 0, isa, pos, lgid.p, headRow.p, M);
        };
        regroup();
        uint32_t rounds = 0;
        for (;;) {
            hipLaunchKernelGGL(kb_double_keys, dim3(blocksExact(M)), dim3(256), 0, 0, isa, pos, M, h, n, key.p);
            const uint32_t *perm = sortByGroupAndKey(key.p, keyAlt.p, rowBits, lgid.p, nGroups, M, pA.p, pB.p, g2a.p, g2b.p);
            hipLaunchKernelGGL(kb_gather<uint64_t>, dim3(blocksExact(M)), dim3(256), 0, 0, pos, perm, M, posAlt);
            hipLaunchKernelGGL(kb_gather<uint64_t>, dim3(blocksExact(M)), dim3(256), 0, 0, key.p, perm, M, keyAlt.p);
            std::swap(pos, posAlt);
            HIPB(hipMemset(dTies.p, 0, 8));
            hipLaunchKernelGGL(kb_double_heads, dim3(blocksExact(M)), dim3(256), 0, 0, keyAlt.p, hd, M, hdAlt, dTies.p);
            std::swap(hd, hdAlt);
            unsigned long long ties = 0;
            HIPB(hipMemcpy(&ties, dTies.p, 8, hipMemcpyDeviceToHost));
            rounds++;
            if (verbose) std::fprintf(stderr, "[cf-build] doubling round %u (depth %llu): %llu of %u leftover suffixes still tied\n", rounds, (unsigned long long)h, ties, M);
            if (ties == 0) break;
            if (rounds > 64 || h > n) throw std::runtime_error("prefix doubling did not converge");
            regroup();
            h *= 2;
        }
        ea.sa = pos; ea.rows = lrow.p; ea.rowBase = 0; ea.count = M;
        hipLaunchKernelGGL(kb_emit_list, dim3(blocksExact(M)), dim3(256), 0, 0, ea);
        HIPB(hipDeviceSynchronize());
        tDouble = now() - tl;
    }
    isaLo.release(); isaHi.release();
    if (verbose) std::fprintf(stderr, "[cf-build] collect %.2fs, first sort %.2fs, 29-mer refinement %.2fs, emit %.2fs, prefix doubling %.2fs\n", tCollect, tSort, tRefine, tEmit, tDouble);
    HIPB(hipMemcpy(&out.zOff, dZoff.p, 8, hipMemcpyDeviceToHost));

    // ---- sides: pack 2-bit BWT + cumulative occ (bt2_idx.h:3700-3730)
    const uint64_t nSideWords = numSides * 12;
    Dev<uint8_t> dSides;
    Dev<uint32_t> dWordCnt;
    Dev<unsigned long long> dCnt, dOcc;
    dSides.alloc(numSides * 128); dWordCnt.alloc(nSideWords); dCnt.alloc(4 * numSides + 1); dOcc.alloc(4 * numSides + 1);
    hipLaunchKernelGGL(kb_side_words, dim3((unsigned)((nSideWords + 255) / 256)), dim3(256), 0, 0, dBwt.p, nSideWords, out.zOff, dSides.p, dWordCnt.p);
    hipLaunchKernelGGL(kb_side_counts, dim3((unsigned)((numSides + 255) / 256)), dim3(256), 0, 0, dWordCnt.p, numSides, dCnt.p);
    uint64_t totals[4];
    for (int ch = 0; ch < 4; ch++) {
        const size_t sb = device_scan_bytes<unsigned long long>(numSides);
        if (sb > tmp.n) tmp.alloc(sb);
        HIPB((device_scan<unsigned long long, false>(tmp.p, dCnt.p + ch * numSides, dOcc.p + ch * numSides, numSides)));
        unsigned long long lastOcc = 0, lastCnt = 0;
        HIPB(hipMemcpy(&lastOcc, dOcc.p + ch * numSides + numSides - 1, 8, hipMemcpyDeviceToHost));
        HIPB(hipMemcpy(&lastCnt, dCnt.p + ch * numSides + numSides - 1, 8, hipMemcpyDeviceToHost));
        totals[ch] = lastOcc + lastCnt;
    }
    hipLaunchKernelGGL(kb_side_occ, dim3((unsigned)((numSides + 255) / 256)), dim3(256), 0, 0, dOcc.p, numSides, dSides.p);
    HIPB(hipDeviceSynchronize());
    HIPB(hipGetLastError());
    out.sides.resize(numSides * 128);
    HIPB(hipMemcpy(out.sides.data(), dSides.p, out.sides.size(), hipMemcpyDeviceToHost));
    out.sample.resize(offsLen * (wide ? 4 : 2));
    HIPB(hipMemcpy(out.sample.data(), dSample.p, out.sample.size(), hipMemcpyDeviceToHost));
    // fchr (bt2_idx.h:3760-3779): text-char counts; padding rows were counted as A
    totals[0] -= bwtRows - (n + 1);
    out.fchr[0] = 0;
    for (int ch = 0; ch < 4; ch++) out.fchr[ch + 1] = out.fchr[ch] + totals[ch];
    if (out.fchr[4] != n) throw std::runtime_error("internal: BWT char counts do not add up");
    // boundary rows
    {
        unsigned long long nb = 0;
        HIPB(hipMemcpy(&nb, dBoundCount.p, 8, hipMemcpyDeviceToHost));
        std::vector<uint64_t> br(nb); std::vector<uint32_t> bf(nb);
        if (nb) {
            HIPB(hipMemcpy(br.data(), dBoundRow.p, nb * 8, hipMemcpyDeviceToHost));
            HIPB(hipMemcpy(bf.data(), dBoundRef.p, nb * 4, hipMemcpyDeviceToHost));
        }
        out.bounds.resize(nb);
        for (size_t i = 0; i < nb; i++) out.bounds[i] = {br[i], bf[i]};
        std::sort(out.bounds.begin(), out.bounds.end());
    }
    // ---- ftab / eftab (bt2_idx.h:3586-3620, 3781-3821)
    const uint64_t ftabLen = (1ull << (2 * ftabChars)) + 1;
    std::vector<uint64_t> ftab(ftabLen, 0);
    std::vector<uint8_t> absorb(ftabLen, 0);
    {
        const int sh = 2 * (kBinChars - ftabChars);
        for (uint64_t b = 0; b < kBins; b++) ftab[(b >> sh) + 1] += bins[b];
        // the suffixes shorter than ftabChars were binned by their T-padded prefix: take them out again
        std::vector<uint64_t> shortRow(nShort);
        HIPB(hipMemcpy(shortRow.data(), dShortRow.p, nShort * 8, hipMemcpyDeviceToHost));
        struct Short { uint64_t row, code; };
        std::vector<Short> sh2;
        for (uint32_t r = 0; r < nShort; r++) {            // r = remaining chars, suffix at n - r
            uint64_t code = 0;
            for (int j = 0; j < ftabChars; j++) code = (code << 2) | (j < (int)r ? ref.text[n - r + j] : 3u);
            ftab[code + 1]--;
            sh2.push_back({shortRow[r], code});
        }
        std::sort(sh2.begin(), sh2.end(), [](const Short &x, const Short &y) { return x.row < y.row; });
        // absorb runs of short suffixes into the next populated transition
        uint32_t acc = 0;
        for (size_t i = 0; i < sh2.size(); i++) {
            acc++;
            if (i + 1 < sh2.size() && sh2[i + 1].row == sh2[i].row + 1) continue;
            uint64_t nb = sh2[i].code + 1;                 // first populated bin strictly after the padded prefix
            while (nb + 1 < ftabLen && ftab[nb + 1] == 0) nb++;
            if (nb + 1 < ftabLen) absorb[nb] = (uint8_t)acc; else absorb[ftabLen - 1] = (uint8_t)acc;
            acc = 0;
        }
    }
    std::vector<uint64_t> eftab(2 * (uint64_t)ftabChars, 0);
    {
        auto ftabHi = [&](uint64_t i) { return ftab[i] <= n ? ftab[i] : eftab[(ftab[i] ^ ~0ull) * 2 + 1]; };
        uint64_t cur = 0;
        for (uint64_t i = 1; i < ftabLen; i++) {
            const uint64_t lo = ftab[i] + ftabHi(i - 1);
            if (absorb[i] > 0) {
                if (cur * 2 + 1 >= eftab.size()) throw std::runtime_error("internal: eftab overflow");
                eftab[cur * 2] = lo; eftab[cur * 2 + 1] = lo + absorb[i];
                ftab[i] = (cur++) ^ ~0ull;
            } else ftab[i] = lo;
        }
    }
    out.ftab = std::move(ftab); out.eftab = std::move(eftab);
}

void writeIndexFiles(const std::string &base, const JoinedRef &ref, int offRate, int ftabChars, const BuildOut &o) {
    auto open = [](const std::string &p) {
        std::FILE *f = std::fopen(p.c_str(), "wb");
        if (!f) throw std::runtime_error("cannot open index file for writing: " + p);
        return f;
    };
    {   // .1.cf (bt2_io.h:854-880 header; bt2_idx.h:3262-3290 plen; bt2_io.h:989-1027 rstarts; bt2_idx.h:3745-3829; :1612-1616 names)
        std::FILE *f = open(base + ".1.cf");
        try {
            put<int32_t>(f, 1); put<uint64_t>(f, ref.len); put<int32_t>(f, 7); put<int32_t>(f, 2);
            put<int32_t>(f, offRate); put<int32_t>(f, ftabChars); put<int32_t>(f, -1);
            put<uint64_t>(f, ref.nPat);
            putBytes(f, ref.plen.data(), ref.plen.size() * 8);
            put<uint64_t>(f, ref.nFrag);
            putBytes(f, ref.rstarts.data(), ref.rstarts.size() * 8);
            putBytes(f, o.sides.data(), o.sides.size());
            put<uint64_t>(f, o.zOff);
            for (int i = 0; i < 5; i++) put<uint64_t>(f, o.fchr[i]);
            putBytes(f, o.ftab.data(), o.ftab.size() * 8);
            putBytes(f, o.eftab.data(), o.eftab.size() * 8);
            for (const auto &nm : ref.refnames) { putBytes(f, nm.data(), nm.size()); put<uint8_t>(f, '\n'); }
            put<uint8_t>(f, 0);
        } catch (...) { std::fclose(f); throw; }
        if (std::fclose(f) != 0) throw std::runtime_error("error closing .1.cf");
    }
    {   // .2.cf
        std::FILE *f = open(base + ".2.cf");
        try { put<int32_t>(f, 1); putBytes(f, o.sample.data(), o.sample.size()); } catch (...) { std::fclose(f); throw; }
        if (std::fclose(f) != 0) throw std::runtime_error("error closing .2.cf");
    }
    {   // .4.cf (bt2_idx.h:3535, 3745-3752)
        std::FILE *f = open(base + ".4.cf");
        try {
            put<int32_t>(f, 1); put<uint64_t>(f, o.bounds.size());
            for (const auto &b : o.bounds) { put<uint64_t>(f, b.first); put<uint32_t>(f, b.second); }
        } catch (...) { std::fclose(f); throw; }
        if (std::fclose(f) != 0) throw std::runtime_error("error closing .4.cf");
    }
}

}  // namespace

extern "C" {

cf_status cf_build_input_default(cf_build_input *in) {
    if (!in) return CF_ERR_ARG;
    std::memset(in, 0, sizeof *in);
    in->off_rate = 4; in->ftab_chars = 10;
    return CF_OK;
}

cf_status cf_build_timings(double sec[4]) {
    if (!sec) return CF_ERR_ARG;
    std::memcpy(sec, g_btime, sizeof g_btime);
    return CF_OK;
}

const char *cf_build_last_error(void) { return g_berr.c_str(); }

cf_status cf_build_index(const cf_build_input *in, const char *outBase, int device) {
    if (!in || !outBase || in->off_rate < 0 || in->off_rate > 20 || in->ftab_chars < 1 || in->ftab_chars > kBinChars) {
        g_berr = "bad argument (ftab_chars must be 1..12)";
        return CF_ERR_ARG;
    }
    int nd = 0;
    if (hipGetDeviceCount(&nd) != hipSuccess || nd <= 0) { g_berr = "no HIP device visible"; return CF_ERR_NO_DEVICE; }
    try {
        const double t0 = now();
        HIPB(hipSetDevice(device));
        JoinedRef ref;
        if (in->fasta_paths && in->n_fasta > 0) {
            std::vector<std::string> paths(in->fasta_paths, in->fasta_paths + in->n_fasta);
            ingestFasta(paths, ref);
        } else ingestMemory(in->codes, in->seq_off, in->seq_names, in->n_seq, ref);
        const double t1 = now();
        BuildOut out;
        // suffixes per GPU pass: every pass re-scans the packed text, so large references use larger passes
        const uint64_t chunk = in->chunk_suffixes ? in->chunk_suffixes : (ref.len > (1ull << 31) ? (1ull << 30) : (1ull << 28));
        buildOnGpu(ref, in->off_rate, in->ftab_chars, chunk, in->verbose != 0, out);
        const double t2 = now();
        writeIndexFiles(outBase, ref, in->off_rate, in->ftab_chars, out);
        writeTaxonomyFile(std::string(outBase) + ".3.cf", ref, in->conversion_table, in->taxonomy_tree, in->name_table, in->size_table);
        const double t3 = now();
        g_btime[0] = t1 - t0; g_btime[1] = t2 - t1; g_btime[2] = t3 - t2; g_btime[3] = t3 - t0;
        return CF_OK;
    } catch (const HipErr &e) { g_berr = e.what(); return CF_ERR_HIP;
    } catch (const std::bad_alloc &) { g_berr = "out of host memory"; return CF_ERR_NOMEM;
    } catch (const std::exception &e) {
        g_berr = e.what();
        return g_berr.find("cannot open") != std::string::npos ? CF_ERR_IO : CF_ERR_FORMAT;
    }
}

}  // extern "C"
