// cf_device.hip — HBM layout, kernel launches and the C ABI (include/centrifuge_amd.h).
// gfx950 only; there is no CPU path behind any compute entry point.
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>
#include <dlfcn.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/centrifuge_amd.h"
#include "cf_index.hpp"
#include "cf_kernels.hpp"
#include "cf_restore.hpp"
#include "cf_plan.hpp"

using namespace cfamd;

namespace {

thread_local std::string g_err;

struct HipError : std::runtime_error {
    using std::runtime_error::runtime_error;
};

#define HIP_OK(expr)                                                                              \
    do {                                                                                          \
        hipError_t e_ = (expr);                                                                   \
        if (e_ != hipSuccess)                                                                     \
            throw HipError(std::string(#expr) + ": " + hipGetErrorString(e_));                    \
    } while (0)

// ---- a device allocation that frees itself
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

__global__ void __launch_bounds__(256) k_pack(DBatch b, uint8_t *recs, uint32_t W) { pack_body(b, recs, W, cf_global_thread()); }

// W = 4 (reads <= 128 bases): 62 VGPRs and 20 KB of LDS per block = 8 waves/SIMD, with the natural register
// allocation (no launch-bounds pressure: forcing it spilled and ran 1.6x slower, profiles/r01_sweeps.txt).
// W = 6 (<= 192 bases, e.g. 150 bp mates): 96-byte records, 24 KB of LDS = 6 blocks per CU.
// W = 8 (<= 256 bases): 64 VGPRs, 28 KB of LDS = 5 blocks per CU.
template <int G, int W, bool COUNT>
__global__ void __launch_bounds__(256) k_search2(DIndex ix, DParams pr, DBatch b) {
    // strand records of the block's chains, then one rank table per lane
    __shared__ __attribute__((aligned(16))) uint8_t lds[(256 / G) * rec_bytes(W) + 256 * 4 * RankTab<G>::WORDS];
    search2_body<G, W, COUNT>(ix, pr, b, lds);
}

__global__ void __launch_bounds__(64) k_post(DIndex ix, DParams pr, DBatch b) {
    const uint32_t q = cf_global_thread();
    if (q < b.nQueries) post_body(ix, pr, b, q);
}
__global__ void __launch_bounds__(64) k_postfix_only(DIndex ix, DParams pr, DBatch b) {   // debug tap
    const uint32_t i = cf_global_thread();
    if (i < b.nItems / 2) post_fix(ix, pr, b, b.items[i]);
}
__global__ void __launch_bounds__(256) k_emit(DBatch b) {
    const uint32_t q = cf_global_thread();
    if (q < b.nQueries) emit_body(b, q);
}
template <int G>
__global__ void __launch_bounds__(256) k_walk(DIndex ix, DBatch b) { walk_body<G>(ix, b); }
template <int G, bool COUNT>
__global__ void __launch_bounds__(256) k_walk2(DIndex ix, DBatch b) { walk2_body<G, COUNT>(ix, b); }

__global__ void __launch_bounds__(64) k_score(DIndex ix, DParams pr, DBatch b) {
    const uint32_t q = cf_global_thread();
    if (q < b.nQueries) score_body(ix, pr, b, q);
}

// debug: out[i] = LF(rows[i], chars[i]) with a G-lane group per element
__global__ void __launch_bounds__(256) k_plan(DPlan p) { plan_body(p, cf_global_thread()); }
__global__ void __launch_bounds__(256) k_plan_fill(DPlan p) { plan_fill_body(p, cf_global_thread()); }
__global__ void __launch_bounds__(256) k_plan_maxscore(const uint64_t *off, const uint8_t *pass, uint32_t nQueries, int paired, uint32_t *maxScore) {
    plan_maxscore_body(off, pass, nQueries, paired, maxScore, cf_global_thread());
}
__global__ void __launch_bounds__(256) k_compact(const OutRow *out, const uint32_t *nOut, const uint64_t *rowFirst, uint32_t k, uint32_t nQueries, OutRow *dst) {
    compact_body(out, nOut, rowFirst, k, nQueries, dst, cf_global_thread());
}
__global__ void __launch_bounds__(256) k_widen(const uint32_t *in, uint64_t *out, uint32_t n) {      // scan input of the row compaction
    const uint32_t i = cf_global_thread();
    if (i < n) out[i] = in[i]; else if (i == n) out[i] = 0;
}

template <int G, bool WRITE>
__global__ void __launch_bounds__(256) k_restore(DIndex ix, DRestore r) { restore_body<G, WRITE>(ix, r); }
__global__ void __launch_bounds__(256) k_restore_rank(const uint64_t *sumIn, const uint32_t *nextIn, uint64_t *sumOut, uint32_t *nextOut, uint32_t nElem) {
    restore_rank_body(sumIn, nextIn, sumOut, nextOut, nElem, cf_global_thread());
}
// links that reached the '$' row point at the terminator element (index nSeg: sum 0, next = itself)
__global__ void __launch_bounds__(256) k_restore_link(uint64_t *sum, uint32_t *next, uint32_t nSeg) {
    const uint32_t s = cf_global_thread();
    if (s < nSeg) { if (next[s] == kRestoreTerm) next[s] = nSeg; }
    else if (s == nSeg) { sum[s] = 0; next[s] = nSeg; }
}

template <int G>
__global__ void k_debug_rank(DIndex ix, const uint8_t *chars, const uint64_t *rows, uint64_t n, uint64_t *out) {
    const uint64_t i = (uint64_t)cf_global_thread() / G;
    // keep whole groups converged: every lane of a live group runs the cooperative load
    const uint64_t ii = i < n ? i : n - 1;
    uint64_t t, bb; bool two;
    rank_pair<G>(ix, chars[ii] & 3, rows[ii], rows[ii], t, bb, two);
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
struct cf_index {
    HostIndex h;
    int device = -1;             // -1: host-only view
    DIndex d{};
    DevBuf<uint8_t> sides, offs;
    DevBuf<uint64_t> ftab, eftab, boundRow, refTax, paths;
    DevBuf<uint32_t> boundRef, boundBits, refPath, refTidx, pathTidx;
    uint64_t deviceBytes = 0;
    int numCUs = 256;
};

struct cf_classifier {
    cf_index *ix = nullptr;
    cf_params p{};
    std::vector<uint64_t> hostList, exclList;
    DParams d{};
    DevBuf<uint8_t> refExcluded;
    DevBuf<uint64_t> hostSet;
    DevBuf<unsigned long long> counts;
};

struct cf_batch {
    cf_classifier *cl = nullptr;
    uint64_t nReads = 0, nQueries = 0, nItems = 0, nHitsCap = 0;
    int paired = 0;
    DevBuf<uint8_t> seq, pass, recs;
    uint32_t recWords = 0;
    DevBuf<uint64_t> off, hitBase, qRows, qBase, rowVal, cap2, rowFirst;
    DevBuf<uint32_t> seeds, items, slotOf, hitCap, nHits, maxLen, rowRef, nOut, score2, cursor, flag, maxScore, planMax;
    DevBuf<OutRow> outCompact;
    DPlan pl{};                              // device view of the plan buffers
    uint32_t planMaxLen = 0;
    float planMs = 0;                        // k_plan .. k_pack of the last cf_batch_plan
    bool compacted = false;                  // rowFirst / outCompact hold the rows of the last cf_classify
    uint64_t rowsOut = 0;
    DevBuf<Hit> hits;
    DevBuf<QInfo> qinfo;
    DevBuf<HmEntry> hm;
    DevBuf<TcEntry> tc;
    DevBuf<OutRow> out;
    DevBuf<OpCounts> ops;
    DevBuf<uint8_t> scanTmp;
    DBatch d{};
    float ms[5] = {0, 0, 0, 0, 0};
    OpCounts lastOps{};
    bool opsValid = false;                   // lastOps holds the counts of the last cf_classify
    uint64_t lastRows = 0;
    hipEvent_t ev[6] = {};
    bool evInit = false;
    ~cf_batch() { if (evInit) for (auto &e : ev) (void)hipEventDestroy(e); }
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
    ix.refTax.upload(h.uidTid);
    ix.refPath.upload(t.refPath);
    ix.refTidx.upload(t.refTidx);
    ix.paths.upload(t.paths);
    ix.pathTidx.upload(t.pathTidx);

    DIndex &d = ix.d;
    fillIndexScalars(h, t, d);
    d.sides = ix.sides.p; d.ftab = ix.ftab.p; d.eftab = ix.eftab.p;
    d.offs = ix.offs.p;
    d.boundRow = ix.boundRow.p; d.boundRef = ix.boundRef.p; d.boundBits = ix.boundBits.p;
    d.refTax = ix.refTax.p; d.refPath = ix.refPath.p; d.refTidx = ix.refTidx.p;
    d.paths = ix.paths.p; d.pathTidx = ix.pathTidx.p;
    ix.deviceBytes = ix.sides.bytes() + ix.ftab.bytes() + ix.eftab.bytes() + ix.offs.bytes() + ix.boundRow.bytes() +
                     ix.boundRef.bytes() + ix.boundBits.bytes() + ix.refTax.bytes() + ix.refPath.bytes() +
                     ix.refTidx.bytes() + ix.paths.bytes() + ix.pathTidx.bytes();
}

int envInt(const char *name, int dflt) {
    const char *v = std::getenv(name);
    return v && *v ? std::atoi(v) : dflt;
}

// lanes per (read, strand) chain in k_search / per SA row in k_walk, and resident blocks per CU
// (tuning knobs; defaults are the measured best)
int searchLanes() { static const int g = envInt("CF_SEARCH_G", 2); return g; }
int walkLanes() { static const int g = envInt("CF_WALK_G", 2); return g; }
int blocksPerCU() { static const int b = envInt("CF_BLOCKS_PER_CU", 8); return b; }

int searchVersion() { static const int v = envInt("CF_SEARCH_V", 2); return v; }
int walkVersion() { static const int v = envInt("CF_WALK_V", 2); return v; }

int persistentBlocks(const cf_index &ix, uint64_t groups, int blocksPerCU, int lanes = 8) {
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
    } catch (const std::bad_alloc &) { g_err = "out of host memory"; return CF_ERR_NOMEM;
    } catch (const std::exception &e) {
        g_err = e.what();
        return g_err.find("cannot open") != std::string::npos ? CF_ERR_IO : CF_ERR_FORMAT;
    }
}

// the search kernel of a batch: k_search2 (strand records in LDS, one memory round trip per
// iteration) when every read fits its records, else the byte-window kernel k_search
// Returns true when the launched kernel tallied the op counters (k_search always does; k_search2
// only in its instrumented build, which cf_batch_opcounts runs on demand).
bool launchSearch(cf_classifier *cl, cf_batch *bt, hipStream_t st, int blocksCap = 0, bool count = false) {
    cf_index &ix = *cl->ix;
    const bool v2 = searchVersion() == 2 && (bt->recWords == 4 || bt->recWords == 6 || bt->recWords == 8);
    const int g = v2 ? 2 : searchLanes();                                        // k_search2 is built for 2 lanes per chain
    int perCU = blocksPerCU();
    if (v2 && !std::getenv("CF_BLOCKS_PER_CU")) {
        // persistent kernel: exactly the blocks that are resident at once (registers and LDS decide: 8 per CU for
        // 128-base records, 6 for 192-base and 5 for 256-base ones); more would only queue up behind them
        static int occ[3] = {0, 0, 0};
        int &o = occ[bt->recWords == 4 ? 0 : bt->recWords == 6 ? 1 : 2];
        if (!o) {
            int n = 0;
            const hipError_t e = bt->recWords == 4   ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, k_search2<2, 4, false>, 256, 0)
                                 : bt->recWords == 6 ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, k_search2<2, 6, false>, 256, 0)
                                                     : hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, k_search2<2, 8, false>, 256, 0);
            o = (e == hipSuccess && n > 0) ? n : perCU;
        }
        perCU = o;
    }
    int blocks = persistentBlocks(ix, bt->nItems, perCU, g);
    if (blocksCap) blocks = std::min(blocks, blocksCap);
    const DBatch &d = bt->d;
    const dim3 gr(blocks), bl(256);
    if (v2) {
        if (bt->recWords == 4) { if (count) hipLaunchKernelGGL((k_search2<2, 4, true>), gr, bl, 0, st, ix.d, cl->d, d); else hipLaunchKernelGGL((k_search2<2, 4, false>), gr, bl, 0, st, ix.d, cl->d, d); }
        else if (bt->recWords == 6) { if (count) hipLaunchKernelGGL((k_search2<2, 6, true>), gr, bl, 0, st, ix.d, cl->d, d); else hipLaunchKernelGGL((k_search2<2, 6, false>), gr, bl, 0, st, ix.d, cl->d, d); }
        else { if (count) hipLaunchKernelGGL((k_search2<2, 8, true>), gr, bl, 0, st, ix.d, cl->d, d); else hipLaunchKernelGGL((k_search2<2, 8, false>), gr, bl, 0, st, ix.d, cl->d, d); }
        return count;
    }
    if (g == 2) hipLaunchKernelGGL(k_search<2>, gr, bl, 0, st, ix.d, cl->d, d);
    else if (g == 4) hipLaunchKernelGGL(k_search<4>, gr, bl, 0, st, ix.d, cl->d, d);
    else hipLaunchKernelGGL(k_search<8>, gr, bl, 0, st, ix.d, cl->d, d);
    return true;
}

bool launchWalk(cf_classifier *cl, cf_batch *bt, hipStream_t st, uint64_t totalRows, bool count = false) {
    cf_index &ix = *cl->ix;
    const int g = walkVersion() == 2 ? 2 : walkLanes();
    const dim3 gr(persistentBlocks(ix, totalRows, blocksPerCU(), g)), bl(256);
    const DBatch &d = bt->d;
    if (walkVersion() == 2) {                                                    // k_walk2 likewise
        if (count) hipLaunchKernelGGL((k_walk2<2, true>), gr, bl, 0, st, ix.d, d); else hipLaunchKernelGGL((k_walk2<2, false>), gr, bl, 0, st, ix.d, d);
        return count;
    }
    if (g == 2) hipLaunchKernelGGL(k_walk<2>, gr, bl, 0, st, ix.d, d);
    else if (g == 4) hipLaunchKernelGGL(k_walk<4>, gr, bl, 0, st, ix.d, d);
    else hipLaunchKernelGGL(k_walk<8>, gr, bl, 0, st, ix.d, d);
    return true;
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

cf_status cf_index_open(const char *basename, int device, cf_index **out) {
    if (!basename || !out) return CF_ERR_ARG;
    *out = nullptr;
    if (!haveDevice()) { g_err = "no HIP device visible"; return CF_ERR_NO_DEVICE; }
    auto ix = std::make_unique<cf_index>();
    cf_status st = guard([&] {
        HIP_OK(hipSetDevice(device));
        hipDeviceProp_t prop;
        HIP_OK(hipGetDeviceProperties(&prop, device));
        ix->numCUs = prop.multiProcessorCount;
        ix->device = device;
        uploadIndex(*ix, basename);
    });
    if (st == CF_OK) *out = ix.release();
    return st;
}

void cf_index_close(cf_index *ix) { delete ix; }

uint64_t cf_index_text_len(const cf_index *ix) { return ix->h.g.len; }
uint64_t cf_index_num_refs(const cf_index *ix) { return ix->h.uid.size(); }
uint64_t cf_index_num_taxa(const cf_index *ix) { return ix->h.taxa.size(); }
uint64_t cf_index_device_bytes(const cf_index *ix) { return ix->deviceBytes; }
int cf_index_compressed(const cf_index *ix) { return ix->h.compressed ? 1 : 0; }
int cf_index_sa_width(const cf_index *ix) { return ix->h.offw ? 4 : 2; }
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
        if (!t.refExcluded.empty()) { cl->refExcluded.upload(t.refExcluded); cl->d.refExcluded = cl->refExcluded.p; }
        if (!t.hostSet.empty()) { cl->hostSet.upload(t.hostSet); cl->d.hostSet = cl->hostSet.p; cl->d.nHostSet = (uint32_t)t.hostSet.size(); }
        cl->counts.alloc(2 * ix->h.taxa.size());
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

// The batch plan on the device: filters and hit capacities (k_plan), work list and hit-list bases (two
// exclusive scans + k_plan_fill), max_score per query; its three scalars come back to size the batch.
static void planOnDevice(cf_batch *bt, hipStream_t st, uint32_t &nPass, uint64_t &hitsTotal, uint32_t &maxLen) {
    const uint64_t nReads = bt->nReads;
    const DPlan &pl = bt->pl;
    HIP_OK(hipMemsetAsync(bt->planMax.p, 0, 4, st));
    const dim3 gp((unsigned)((nReads + 1 + 255) / 256)), bl(256);
    hipLaunchKernelGGL(k_plan, gp, bl, 0, st, pl);
    size_t t1 = 0, t2 = 0;
    HIP_OK(hipcub::DeviceScan::ExclusiveSum(nullptr, t1, bt->flag.p, bt->slotOf.p, (int)(nReads + 1), st));
    HIP_OK(hipcub::DeviceScan::ExclusiveSum(nullptr, t2, bt->cap2.p, bt->hitBase.p, (int)(nReads + 1), st));
    if (std::max(t1, t2) > bt->scanTmp.n) bt->scanTmp.alloc(std::max(t1, t2));
    size_t tb = bt->scanTmp.n;
    HIP_OK(hipcub::DeviceScan::ExclusiveSum(bt->scanTmp.p, tb, bt->flag.p, bt->slotOf.p, (int)(nReads + 1), st));
    tb = bt->scanTmp.n;
    HIP_OK(hipcub::DeviceScan::ExclusiveSum(bt->scanTmp.p, tb, bt->cap2.p, bt->hitBase.p, (int)(nReads + 1), st));
    HIP_OK(hipMemcpyAsync(&nPass, bt->slotOf.p + nReads, 4, hipMemcpyDeviceToHost, st));
    HIP_OK(hipMemcpyAsync(&hitsTotal, bt->hitBase.p + nReads, 8, hipMemcpyDeviceToHost, st));
    HIP_OK(hipMemcpyAsync(&maxLen, bt->planMax.p, 4, hipMemcpyDeviceToHost, st));
    hipLaunchKernelGGL(k_plan_fill, gp, bl, 0, st, pl);
    if (bt->nQueries) hipLaunchKernelGGL(k_plan_maxscore, dim3((unsigned)((bt->nQueries + 255) / 256)), bl, 0, st, bt->off.p, bt->pass.p,
                                         (uint32_t)bt->nQueries, bt->paired, bt->maxScore.p);
    HIP_OK(hipStreamSynchronize(st));                // the plan's three scalars size the rest of the batch
    HIP_OK(hipGetLastError());
}

// strand records of k_search2: 2-bit search-order words + N masks, packed from the resident reads
static void packRecords(cf_batch *bt, hipStream_t st) {
    if (!bt->recWords || !bt->nItems) return;
    const uint64_t threads = bt->nItems * (uint64_t)bt->recWords;
    hipLaunchKernelGGL(k_pack, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, st, bt->d, bt->recs.p, bt->recWords);
}

cf_status cf_batch_create(cf_classifier *cl, const uint8_t *seq, const uint64_t *off, const uint32_t *seeds,
                          uint64_t nReads, int paired, cf_batch **out) {
    if (!cl || !off || !seeds || !out || (paired && (nReads & 1)) || nReads >= 0x7fffffffull) return CF_ERR_ARG;
    *out = nullptr;
    auto bt = std::make_unique<cf_batch>();
    cf_status st = guard([&] {
        HIP_OK(hipSetDevice(cl->ix->device));
        bt->cl = cl; bt->nReads = nReads; bt->paired = paired ? 1 : 0;
        bt->nQueries = paired ? nReads / 2 : nReads;
        const uint64_t nbases = off[nReads];
        if (nbases && !seq) throw std::runtime_error("null sequence buffer");
        // uploads: the reads as handed over (1 byte per base), their offsets and seeds — nothing else
        bt->seq.alloc(nbases + 16);
        HIP_OK(hipMemsetAsync(bt->seq.p + nbases, 0, 16, 0));
        if (nbases) HIP_OK(hipMemcpyAsync(bt->seq.p, seq, nbases, hipMemcpyHostToDevice, 0));
        bt->off.alloc(nReads + 1); bt->seeds.alloc(nReads + 1);
        HIP_OK(hipMemcpyAsync(bt->off.p, off, (nReads + 1) * 8, hipMemcpyHostToDevice, 0));
        if (nReads) HIP_OK(hipMemcpyAsync(bt->seeds.p, seeds, nReads * 4, hipMemcpyHostToDevice, 0));
        HIP_OK(hipMemsetAsync(bt->seeds.p + nReads, 0, 4, 0));
        // the plan, on the device (plan_body): filters, hit capacity per read, work list, hit-list bases
        bt->pass.alloc(nReads); bt->hitCap.alloc(nReads + 1); bt->flag.alloc(nReads + 1); bt->cap2.alloc(nReads + 1);
        bt->slotOf.alloc(nReads + 1); bt->hitBase.alloc(nReads + 1); bt->planMax.alloc(1); bt->items.alloc(nReads);
        DPlan &pl = bt->pl;
        pl.seq = bt->seq.p; pl.off = bt->off.p; pl.nReads = (uint32_t)nReads; pl.ftabChars = cl->ix->h.g.ftabChars;
        pl.pass = bt->pass.p; pl.hitCap = bt->hitCap.p; pl.flag = bt->flag.p; pl.cap2 = bt->cap2.p; pl.slotOf = bt->slotOf.p;
        pl.hitBase = bt->hitBase.p; pl.items = bt->items.p; pl.maxLen = bt->planMax.p;
        bt->maxScore.alloc(bt->nQueries);
        uint32_t nPass = 0, maxLen = 0; uint64_t hitsTotal = 0;
        planOnDevice(bt.get(), nullptr, nPass, hitsTotal, maxLen);
        bt->planMaxLen = maxLen;
        bt->nItems = 2ull * nPass;
        bt->nHitsCap = hitsTotal;
        BatchPlan plan;                                  // only its record-width rule is used on this path
        plan.hitsTotal = hitsTotal; plan.maxLen = maxLen;
        bt->hits.alloc(hitsTotal);
        bt->nHits.alloc(bt->nItems);
        bt->maxLen.alloc(bt->nItems);
        bt->qinfo.alloc(bt->nQueries);
        bt->qRows.alloc(bt->nQueries + 1); bt->qBase.alloc(bt->nQueries + 1);
        bt->out.alloc(bt->nQueries * (uint64_t)cl->d.k);
        bt->nOut.alloc(bt->nQueries); bt->score2.alloc(bt->nQueries);
        bt->cursor.alloc(4); bt->ops.alloc(1);
        {   // scan workspace: row plan of cf_classify, row compaction of cf_batch_results_compact (both over nQueries + 1 u64)
            size_t tmpBytes = 0;
            HIP_OK(hipcub::DeviceScan::ExclusiveSum(nullptr, tmpBytes, bt->qRows.p, bt->qBase.p, (int)(bt->nQueries + 1)));
            if (tmpBytes > bt->scanTmp.n) bt->scanTmp.alloc(tmpBytes);
        }
        for (auto &e : bt->ev) HIP_OK(hipEventCreate(&e));
        bt->evInit = true;
        DBatch &d = bt->d;
        d.seq = bt->seq.p; d.off = bt->off.p; d.seeds = bt->seeds.p; d.pass = bt->pass.p; d.items = bt->items.p;
        d.slotOf = bt->slotOf.p; d.hitBase = bt->hitBase.p; d.hitCap = bt->hitCap.p; d.hits = bt->hits.p;
        d.nHits = bt->nHits.p; d.maxLen = bt->maxLen.p; d.qinfo = bt->qinfo.p; d.qRows = bt->qRows.p; d.qBase = bt->qBase.p;
        d.out = bt->out.p; d.nOut = bt->nOut.p; d.score2 = bt->score2.p;
        d.counts = cl->counts.p; d.nTaxa = (uint32_t)cl->ix->h.taxa.size();
        d.nReads = (uint32_t)nReads; d.nQueries = (uint32_t)bt->nQueries; d.nItems = (uint32_t)bt->nItems;
        d.paired = bt->paired; d.cursor = bt->cursor.p; d.ops = bt->ops.p;
        // strand records of k_search2: 2-bit search-order words + N masks, packed once per batch
        bt->recWords = plan.recWords();
        // k_search2 keeps a strand's hit count in 8 bits: hits per strand <= #N + (L - #N) / ftabChars + 2 with
        // #N <= 0.15 L for a classified read (only an index with a very short ftab can get near that)
        if ((uint64_t)(0.15 * maxLen) + maxLen / (uint64_t)std::max(1, cl->ix->h.g.ftabChars) + 3 >= 255) bt->recWords = 0;
        if (bt->recWords && bt->nItems) {
            bt->recs.alloc(bt->nItems * (uint64_t)rec_bytes((int)bt->recWords));
            d.recs = bt->recs.p; d.recWords = bt->recWords;
            packRecords(bt.get(), nullptr);
            HIP_OK(hipDeviceSynchronize());
            HIP_OK(hipGetLastError());
        }
    });
    if (st == CF_OK) *out = bt.release();
    return st;
}
void cf_batch_destroy(cf_batch *b) { delete b; }
uint64_t cf_batch_num_queries(const cf_batch *b) { return b->nQueries; }

// The device-side preparation of a batch again, from its resident reads: plan + strand records.  cf_batch_create
// has done it once; a caller that measures (or re-uses the resident reads) runs it as the first stage of a pass.
cf_status cf_batch_plan(cf_batch *bt, void *streamv) {
    if (!bt) return CF_ERR_ARG;
    return guard([&] {
        HIP_OK(hipSetDevice(bt->cl->ix->device));
        hipStream_t st = static_cast<hipStream_t>(streamv);
        HIP_OK(hipEventRecord(bt->ev[5], st));
        uint32_t nPass = 0, maxLen = 0; uint64_t hitsTotal = 0;
        planOnDevice(bt, st, nPass, hitsTotal, maxLen);
        if (2ull * nPass != bt->nItems || hitsTotal != bt->nHitsCap || maxLen != bt->planMaxLen)
            throw std::runtime_error("cf_batch_plan: the resident reads changed since cf_batch_create");
        packRecords(bt, st);
        HIP_OK(hipEventRecord(bt->ev[0], st));       // cf_classify re-records ev[0]; read the pair before that
        HIP_OK(hipEventSynchronize(bt->ev[0]));
        HIP_OK(hipGetLastError());
        HIP_OK(hipEventElapsedTime(&bt->planMs, bt->ev[5], bt->ev[0]));
    });
}
cf_status cf_batch_plan_ms(const cf_batch *bt, float *ms) {
    if (!bt || !ms) return CF_ERR_ARG;
    *ms = bt->planMs;
    return CF_OK;
}

cf_status cf_classify(cf_classifier *cl, cf_batch *bt, void *streamv) {
    if (!cl || !bt || bt->cl != cl) return CF_ERR_ARG;
    return guard([&] {
        cf_index &ix = *cl->ix;
        HIP_OK(hipSetDevice(ix.device));
        hipStream_t st = static_cast<hipStream_t>(streamv);
        DBatch &d = bt->d;
        bt->compacted = false;
        HIP_OK(hipMemsetAsync(bt->cursor.p, 0, 16, st));
        HIP_OK(hipMemsetAsync(bt->ops.p, 0, sizeof(OpCounts), st));
        HIP_OK(hipMemsetAsync(bt->qRows.p, 0, 8 * (bt->nQueries + 1), st));
        HIP_OK(hipEventRecord(bt->ev[0], st));
        bool counted = true;
        if (bt->nItems) counted = launchSearch(cl, bt, st) && counted;
        HIP_OK(hipEventRecord(bt->ev[1], st));
        const int qBlocks64 = (int)((bt->nQueries + 63) / 64);
        if (bt->nQueries) hipLaunchKernelGGL(k_post, dim3(qBlocks64), dim3(64), 0, st, ix.d, cl->d, d);
        size_t tmpBytes = bt->scanTmp.n;
        HIP_OK(hipcub::DeviceScan::ExclusiveSum(bt->scanTmp.p, tmpBytes, bt->qRows.p, bt->qBase.p,
                                                (int)(bt->nQueries + 1), st));
        uint64_t totalRows = 0;
        HIP_OK(hipMemcpyAsync(&totalRows, bt->qBase.p + bt->nQueries, 8, hipMemcpyDeviceToHost, st));
        HIP_OK(hipStreamSynchronize(st));            // the one host round trip of a batch: size the row workspace
        bt->rowVal.ensure(totalRows); bt->rowRef.ensure(totalRows); bt->hm.ensure(totalRows); bt->tc.ensure(totalRows);
        d.rowVal = bt->rowVal.p; d.rowRef = bt->rowRef.p; d.hm = bt->hm.p; d.tc = bt->tc.p;
        d.nRowsTotal = totalRows;
        bt->lastRows = totalRows;
        if (bt->nQueries && totalRows)
            hipLaunchKernelGGL(k_emit, dim3((int)((bt->nQueries + 255) / 256)), dim3(256), 0, st, d);
        HIP_OK(hipEventRecord(bt->ev[2], st));
        if (totalRows) counted = launchWalk(cl, bt, st, totalRows) && counted;
        HIP_OK(hipEventRecord(bt->ev[3], st));
        if (bt->nQueries) hipLaunchKernelGGL(k_score, dim3(qBlocks64), dim3(64), 0, st, ix.d, cl->d, d);
        HIP_OK(hipEventRecord(bt->ev[4], st));
        HIP_OK(hipMemcpyAsync(&bt->lastOps, bt->ops.p, sizeof(OpCounts), hipMemcpyDeviceToHost, st));
        HIP_OK(hipStreamSynchronize(st));
        HIP_OK(hipGetLastError());
        bt->lastOps.nRows = totalRows;
        bt->opsValid = counted;
        HIP_OK(hipEventElapsedTime(&bt->ms[0], bt->ev[0], bt->ev[1]));
        HIP_OK(hipEventElapsedTime(&bt->ms[1], bt->ev[1], bt->ev[2]));
        HIP_OK(hipEventElapsedTime(&bt->ms[2], bt->ev[2], bt->ev[3]));
        HIP_OK(hipEventElapsedTime(&bt->ms[3], bt->ev[3], bt->ev[4]));
        HIP_OK(hipEventElapsedTime(&bt->ms[4], bt->ev[0], bt->ev[4]));
    });
}

cf_status cf_batch_results(cf_batch *bt, cf_row *rows, uint32_t *nRows, uint32_t *score2) {
    if (!bt || !rows || !nRows || !score2) return CF_ERR_ARG;
    static_assert(sizeof(cf_row) == sizeof(OutRow), "cf_row layout");
    return guard([&] {
        HIP_OK(hipSetDevice(bt->cl->ix->device));
        HIP_OK(hipMemcpy(rows, bt->out.p, bt->nQueries * (uint64_t)bt->cl->d.k * sizeof(OutRow), hipMemcpyDeviceToHost));
        HIP_OK(hipMemcpy(nRows, bt->nOut.p, bt->nQueries * 4, hipMemcpyDeviceToHost));
        HIP_OK(hipMemcpy(score2, bt->score2.p, bt->nQueries * 4, hipMemcpyDeviceToHost));
    });
}

cf_status cf_batch_max_scores(const cf_batch *bt, uint32_t *out) {
    if (!bt || !out) return CF_ERR_ARG;
    return guard([&] {
        HIP_OK(hipSetDevice(bt->cl->ix->device));
        if (bt->nQueries) HIP_OK(hipMemcpy(out, bt->maxScore.p, bt->nQueries * 4, hipMemcpyDeviceToHost));
    });
}

// Rows of all queries back to back (query order): the egress a front end wants — on average 1-2 rows
// per query travel instead of k slots.  The compaction (scan of the row counts + one move kernel) runs
// on the first call after a cf_classify.
static void compactRows(cf_batch *bt) {
    if (bt->compacted) return;
    const uint32_t nq = (uint32_t)bt->nQueries;
    bt->rowFirst.ensure(nq + 1); bt->cap2.ensure(nq + 1);
    const dim3 g((nq + 1 + 255) / 256), bl(256);
    hipLaunchKernelGGL(k_widen, g, bl, 0, 0, bt->nOut.p, bt->cap2.p, nq);
    size_t tb = 0;
    HIP_OK(hipcub::DeviceScan::ExclusiveSum(nullptr, tb, bt->cap2.p, bt->rowFirst.p, (int)(nq + 1)));
    if (tb > bt->scanTmp.n) bt->scanTmp.alloc(tb);
    tb = bt->scanTmp.n;
    HIP_OK(hipcub::DeviceScan::ExclusiveSum(bt->scanTmp.p, tb, bt->cap2.p, bt->rowFirst.p, (int)(nq + 1)));
    uint64_t total = 0;
    HIP_OK(hipMemcpy(&total, bt->rowFirst.p + nq, 8, hipMemcpyDeviceToHost));
    bt->outCompact.ensure(total);
    if (nq && total) hipLaunchKernelGGL(k_compact, g, bl, 0, 0, bt->out.p, bt->nOut.p, bt->rowFirst.p, (uint32_t)bt->cl->d.k, nq, bt->outCompact.p);
    HIP_OK(hipDeviceSynchronize());
    HIP_OK(hipGetLastError());
    bt->rowsOut = total;
    bt->compacted = true;
}

cf_status cf_batch_num_rows(cf_batch *bt, uint64_t *total) {
    if (!bt || !total) return CF_ERR_ARG;
    return guard([&] { HIP_OK(hipSetDevice(bt->cl->ix->device)); compactRows(bt); *total = bt->rowsOut; });
}

cf_status cf_batch_results_compact(cf_batch *bt, cf_row *rows, uint64_t rowsCap, uint32_t *nRows, uint32_t *score2) {
    if (!bt || !nRows || !score2 || (!rows && rowsCap)) return CF_ERR_ARG;
    return guard([&] {
        HIP_OK(hipSetDevice(bt->cl->ix->device));
        compactRows(bt);
        if (rowsCap < bt->rowsOut) throw std::runtime_error("cf_batch_results_compact: the row buffer is smaller than cf_batch_num_rows");
        if (bt->rowsOut) HIP_OK(hipMemcpy(rows, bt->outCompact.p, bt->rowsOut * sizeof(OutRow), hipMemcpyDeviceToHost));
        if (bt->nQueries) {
            HIP_OK(hipMemcpy(nRows, bt->nOut.p, bt->nQueries * 4, hipMemcpyDeviceToHost));
            HIP_OK(hipMemcpy(score2, bt->score2.p, bt->nQueries * 4, hipMemcpyDeviceToHost));
        }
    });
}

cf_status cf_batch_timings(const cf_batch *bt, float ms[5]) {
    if (!bt || !ms) return CF_ERR_ARG;
    std::memcpy(ms, bt->ms, sizeof bt->ms);
    return CF_OK;
}
cf_status cf_batch_opcounts(cf_batch *bt, cf_opcounts *o) {
    if (!bt || !o) return CF_ERR_ARG;
    if (!bt->opsValid) {
        // The production kernels carry no counters: tally once with their instrumented builds.  The
        // work is a pure function of the batch, so the counts are those of the timed launches.  (The
        // search pass rewrites the hit lists of the batch; rows already in `out` are not touched.)
        cf_classifier *cl = bt->cl;
        const cf_status st = guard([&] {
            HIP_OK(hipSetDevice(cl->ix->device));
            HIP_OK(hipMemset(bt->cursor.p, 0, 16));
            HIP_OK(hipMemset(bt->ops.p, 0, sizeof(OpCounts)));
            if (bt->nItems) launchSearch(cl, bt, nullptr, 0, true);
            if (bt->lastRows) launchWalk(cl, bt, nullptr, bt->lastRows, true);
            HIP_OK(hipDeviceSynchronize());
            HIP_OK(hipGetLastError());
            const uint64_t rows = bt->lastRows;
            HIP_OK(hipMemcpy(&bt->lastOps, bt->ops.p, sizeof(OpCounts), hipMemcpyDeviceToHost));
            bt->lastOps.nRows = rows;
            bt->opsValid = true;
        });
        if (st != CF_OK) return st;
    }
    o->n_ftab = bt->lastOps.nFtab; o->n_pair = bt->lastOps.nPair; o->n_pair2 = bt->lastOps.nPair2;
    o->n_single = bt->lastOps.nSingle; o->n_walk = bt->lastOps.nWalk; o->n_rows = bt->lastOps.nRows;
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
void *cf_counts_device(cf_classifier *cl) { return cl ? cl->counts.p : nullptr; }

cf_status cf_counts_allreduce(cf_classifier *cl, void *comm, void *streamv) {
    if (!cl || !comm) return CF_ERR_ARG;
    // ncclResult_t ncclAllReduce(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t)
    using AllReduceFn = int (*)(const void *, void *, size_t, int, int, void *, hipStream_t);
    static AllReduceFn fn = [] {
        void *h = dlopen("librccl.so.1", RTLD_LAZY | RTLD_LOCAL);
        if (!h) h = dlopen("librccl.so", RTLD_LAZY | RTLD_LOCAL);
        return h ? reinterpret_cast<AllReduceFn>(dlsym(h, "ncclAllReduce")) : nullptr;
    }();
    if (!fn) { g_err = "librccl.so.1 (ncclAllReduce) could not be loaded"; return CF_ERR_HIP; }
    return guard([&] {
        HIP_OK(hipSetDevice(cl->ix->device));
        constexpr int kNcclUint64 = 5, kNcclSum = 0;          // rccl.h: ncclDataType_t / ncclRedOp_t
        const int rc = fn(cl->counts.p, cl->counts.p, 2 * cl->ix->h.taxa.size(), kNcclUint64, kNcclSum, comm,
                          static_cast<hipStream_t>(streamv));
        if (rc != 0) throw HipError("ncclAllReduce failed with ncclResult_t " + std::to_string(rc));
    });
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
    if (bt->nItems == 0) return CF_OK;
    return guard([&] {
        HIP_OK(hipMemset(bt->cursor.p, 0, 16));
        HIP_OK(hipMemset(bt->ops.p, 0, sizeof(OpCounts)));
        launchSearch(cl, bt, nullptr, 1);
        hipLaunchKernelGGL(k_postfix_only, dim3(1), dim3(64), 0, 0, cl->ix->d, cl->d, bt->d);
        HIP_OK(hipDeviceSynchronize());
        HIP_OK(hipGetLastError());
        uint32_t n[2];
        HIP_OK(hipMemcpy(n, bt->nHits.p, 8, hipMemcpyDeviceToHost));
        std::vector<Hit> all(bt->nHitsCap);
        HIP_OK(hipMemcpy(all.data(), bt->hits.p, all.size() * sizeof(Hit), hipMemcpyDeviceToHost));
        const uint32_t cap = (uint32_t)(bt->nHitsCap / 2);
        cf_hit *o[2] = {hf, hr};
        for (int f = 0; f < 2; f++) {
            nhits[f] = n[f];
            for (uint32_t i = 0; i < n[f] && i < maxHits; i++) {
                const Hit &h = all[f * cap + i];
                o[f][i].top = h.top; o[f][i].bot = h.bot; o[f][i].bwoff = h.bwoff; o[f][i].len = h.len;
            }
        }
    });
}

cf_status cf_debug_resolve(cf_index *ix, const uint64_t *rows, uint64_t n, uint32_t *refs) {
    if (!ix || !rows || !refs) return CF_ERR_ARG;
    if (ix->device < 0) return CF_ERR_NO_DEVICE;
    return guard([&] {
        HIP_OK(hipSetDevice(ix->device));
        DevBuf<uint64_t> r; DevBuf<uint32_t> o, cur;
        r.upload(std::vector<uint64_t>(rows, rows + n)); o.alloc(n); cur.alloc(4);
        HIP_OK(hipMemset(cur.p, 0, 16));
        DBatch d{};
        d.rowVal = r.p; d.rowRef = o.p; d.cursor = cur.p; d.nRowsTotal = n;
        if (walkVersion() == 2) hipLaunchKernelGGL((k_walk2<2, false>), dim3(persistentBlocks(*ix, n, 4, 2)), dim3(256), 0, 0, ix->d, d);
        else hipLaunchKernelGGL(k_walk<8>, dim3(persistentBlocks(*ix, n, 4)), dim3(256), 0, 0, ix->d, d);
        HIP_OK(hipDeviceSynchronize());
        HIP_OK(hipGetLastError());
        HIP_OK(hipMemcpy(refs, o.p, n * 4, hipMemcpyDeviceToHost));
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
        DRestore r{};
        r.n = n;
        int lg = 0;
        while ((n >> lg) > 1) lg++;
        const char *es = std::getenv("CF_RESTORE_SHIFT");
        r.shift = es ? (uint32_t)std::atoi(es) : (uint32_t)std::min(10, std::max(4, lg - 18));
        if ((n >> r.shift) + 3 >= 0xffffffffull) throw std::runtime_error("cf_index_restore: index too large for 32-bit segment ids");
        r.nMarked = (uint32_t)(n >> r.shift) + 1;
        r.nSeg = r.nMarked + ((n & ((1ull << r.shift) - 1)) ? 1u : 0u);
        const uint32_t startSeg = r.nSeg - 1;                                      // the walk that starts at row n
        r.maxSteps = std::min<uint64_t>(n + 1, (1ull << r.shift) * 8192ull);
        const uint32_t nElem = r.nSeg + 1;
        DevBuf<uint64_t> sumA, sumB; DevBuf<uint32_t> nextA, nextB, cur, err, text;
        sumA.alloc(nElem); sumB.alloc(nElem); nextA.alloc(nElem); nextB.alloc(nElem); cur.alloc(4); err.alloc(1);
        const uint64_t words = (n + 15) / 16 + 1;
        text.alloc(words);
        HIP_OK(hipMemsetAsync(text.p, 0, words * 4, 0));
        HIP_OK(hipMemsetAsync(cur.p, 0, 16, 0));
        HIP_OK(hipMemsetAsync(err.p, 0, 4, 0));
        r.cursor = cur.p; r.segLen = sumA.p; r.segNext = nextA.p; r.err = err.p; r.text = text.p;
        const dim3 gr(persistentBlocks(*ix, r.nSeg, blocksPerCU(), 2)), bl(256);
        const bool verbose = std::getenv("CF_RESTORE_VERBOSE") != nullptr;
        struct Events {                                                            // released on every way out
            hipEvent_t e[4] = {};
            ~Events() { for (auto &x : e) if (x) (void)hipEventDestroy(x); }
        } evs;
        hipEvent_t *ev = evs.e;
        for (int i = 0; i < 4; i++) HIP_OK(hipEventCreate(&ev[i]));
        HIP_OK(hipEventRecord(ev[0], 0));
        hipLaunchKernelGGL((k_restore<2, false>), gr, bl, 0, 0, ix->d, r);
        HIP_OK(hipEventRecord(ev[1], 0));
        const dim3 ge((nElem + 255) / 256);
        hipLaunchKernelGGL(k_restore_link, ge, bl, 0, 0, sumA.p, nextA.p, r.nSeg);
        uint64_t *si = sumA.p, *so = sumB.p; uint32_t *ni = nextA.p, *no = nextB.p;
        for (uint64_t span = 1; span < nElem; span <<= 1) {
            hipLaunchKernelGGL(k_restore_rank, ge, bl, 0, 0, si, ni, so, no, nElem);
            std::swap(si, so); std::swap(ni, no);
        }
        uint64_t total = 0; uint32_t e = 0;
        HIP_OK(hipMemcpy(&total, si + startSeg, 8, hipMemcpyDeviceToHost));
        HIP_OK(hipMemcpy(&e, err.p, 4, hipMemcpyDeviceToHost));
        if (e || total != n) throw std::runtime_error("cf_index_restore: the BWT does not invert to one text of the stated length (damaged index)");
        r.segEnd = si;
        HIP_OK(hipMemsetAsync(cur.p, 0, 16, 0));
        HIP_OK(hipEventRecord(ev[2], 0));
        hipLaunchKernelGGL((k_restore<2, true>), gr, bl, 0, 0, ix->d, r);
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
        hipLaunchKernelGGL(k_random_sides, dim3(blocks), dim3(256), 0, 0, ix->sides.p, ix->h.g.numSides, 2u, 1ull, sink.p);
        HIP_OK(hipEventRecord(a, 0));
        hipLaunchKernelGGL(k_random_sides, dim3(blocks), dim3(256), 0, 0, ix->sides.p, ix->h.g.numSides, (uint32_t)steps, 7ull, sink.p);
        HIP_OK(hipEventRecord(b, 0));
        HIP_OK(hipEventSynchronize(b));
        float ms = 0;
        HIP_OK(hipEventElapsedTime(&ms, a, b));
        (void)hipEventDestroy(a); (void)hipEventDestroy(b);
        *gbps = (double)(blocks * 32ull * (uint64_t)steps) * 128.0 / (ms * 1e-3) / 1e9;
    });
}

}  // extern "C"
