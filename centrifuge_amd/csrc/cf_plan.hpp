// cf_plan.hpp — host-side preparation shared by the device layer (cf_device.hip)
// and the CPU single-step harness of the unit tests (tests/emu): the flat
// taxonomy tables and the classifier parameters (once per index / classifier),
// and the harness's host version of the per-batch work plan.
#pragma once
#include <algorithm>
#include <cstdint>
#include <cstdlib>
#include <vector>

#include "../../include/centrifuge_amd.h"
#include "cf_index.hpp"
#include "cf_kernels.hpp"
#include "cf_knobs.hpp"

namespace cfamd {

struct IndexTables {
    std::vector<uint32_t> boundBits, refPath, refTidx, pathTidx;
    std::vector<uint64_t> paths;
    int boundShift = 8;
};

inline IndexTables makeIndexTables(const HostIndex &h) {
    IndexTables t;
    // Prefilter over the .4.cf boundary rows.  The reference keeps a bitset over
    // row >> shift in front of a std::map (bt2_idx.h:826-850, 2003); any prefilter
    // is only an accelerator of "row is a key of the map".
    const uint64_t m = h.boundRow.size();
    while (m > 0 && ((h.lastBoundary + 1) >> (t.boundShift + 1)) >= 128 * m && t.boundShift < 40) t.boundShift++;
    t.boundBits.assign((((h.lastBoundary + 1) >> t.boundShift) + 32) / 32 + 1, 0);
    for (uint64_t r : h.boundRow) t.boundBits[(r >> t.boundShift) >> 5] |= 1u << ((r >> t.boundShift) & 31);
    const size_t nref = h.uid.size();
    t.refPath.resize(nref + 1);
    t.refTidx.resize(nref + 1);
    for (size_t i = 0; i < nref; i++) {
        t.refPath[i] = h.findPath(h.uidTid[i]);
        t.refTidx[i] = h.taxonIndex(h.uidTid[i]);
    }
    t.paths.assign(h.paths.size() * kPathSlots + 1, 0);
    t.pathTidx.assign(h.paths.size() * kPathSlots + 1, 0);
    for (size_t i = 0; i < h.paths.size(); i++)
        for (int s = 0; s < kPathSlots; s++) {
            t.paths[i * kPathSlots + s] = h.paths[i][s];
            t.pathTidx[i * kPathSlots + s] = h.taxonIndex(h.paths[i][s]);   // tree node or 0: always dense
        }
    return t;
}

// scalar part of the device view; the pointers are filled by the caller
inline void fillIndexScalars(const HostIndex &h, const IndexTables &t, DIndex &d) {
    d.fchr0 = h.fchr[0]; d.fchr1 = h.fchr[1]; d.fchr2 = h.fchr[2]; d.fchr3 = h.fchr[3];
    d.len = h.g.len; d.zOff = h.zOff; d.zSide = h.zOff / kSideChars; d.zIn = (uint32_t)(h.zOff % kSideChars);
    d.ftabChars = h.g.ftabChars; d.offRate = h.g.offRate; d.offw = h.offw ? 1 : 0;
    d.walkRate = d.offRate;                               // the caller points walkOffs at offs (or at its dense table)
    d.lastBoundary = h.lastBoundary; d.nBound = (uint32_t)h.boundRow.size(); d.boundShift = t.boundShift;
    d.nRef = (uint32_t)h.uid.size(); d.tidxOne = h.taxonIndex(1);
    // side = row / 384 by a 32-bit multiply when row >> 7 fits 32 bits; CF_FORCE_WIDE_SIDE=1 takes the 64-bit division
    // on any index (how the tests reach the path indexes beyond 5.5e11 bases take)
    d.small = (((h.g.len + 1024) >> 7) < 0xffffffffull && !cfamd::cf_knob("CF_FORCE_WIDE_SIDE")) ? 1 : 0;
}

struct ClassifierTables {
    std::vector<uint8_t> refExcluded;     // empty when no --exclude-taxids
    std::vector<uint64_t> hostSet;        // sorted; empty when no --host-taxids
};

// Classifier ctor (classifier.h:135-202) + ReportingParams (aln_sink.h:570-588)
inline ClassifierTables makeClassifier(const HostIndex &h, const cf_params &p, DParams &d) {
    ClassifierTables t;
    d = DParams{};
    d.k = (uint32_t)p.khits; d.m = (uint32_t)p.min_hitlen;
    d.inc = (2 * d.m <= 33) ? 10 : (2 * d.m - 33);                       // classifier.h:226
    d.ihits = std::max<uint32_t>(d.k, 5) * (h.compressed ? 4 : 40);
    d.rankSlot = (uint32_t)p.rank_slot; d.traverse = p.tree_traverse;
    if (p.n_exclude > 0) {
        t.refExcluded.resize(h.uid.size() + 1, 0);
        for (size_t r = 0; r < h.uid.size(); r++)
            t.refExcluded[r] = h.inClosure(h.uidTid[r], p.exclude_taxids, p.n_exclude) ? 1 : 0;
    }
    if (p.n_host > 0)
        for (const auto &n : h.tree)
            if (h.inClosure(n.tid, p.host_taxids, p.n_host)) t.hostSet.push_back(n.tid);
    return t;
}

struct BatchPlan {
    std::vector<uint8_t> pass;
    std::vector<uint32_t> items, slotOf, hitCap;
    std::vector<uint64_t> hitBase;
    uint64_t hitsTotal = 0;
    uint64_t maxLen = 0;                  // longest searched read
    // 2-bit words per strand record of k_search2 (0: a read is too long for it, use k_search)
    uint32_t recWords() const { return hitsTotal >= 0xffffffffull ? 0u : maxLen <= 128 ? 4u : maxLen <= 192 ? 6u : maxLen <= 256 ? 8u : 0u; }
};

// Scoring::nFilter (scoring.cpp:104-117) with nCeil = 0 + 0.15f*len (scoring.h:61-63)
// and the length filter of centrifuge.cpp:2562-2577 (multiseedMms = 0).
inline bool matePasses(const uint8_t *s, uint64_t len, uint32_t &nN) {
    nN = 0;
    for (uint64_t i = 0; i < len; i++) nN += s[i] == 4;
    if (len < 2) return false;
    const uint64_t maxns = (uint64_t)(0.0 + (double)0.15f * (double)len);
    return nN <= maxns;
}

// Host restatement of the batch plan.  The device layer plans on the GPU (plan_body / plan_fill_body in
// cf_kernels.hpp, two scans) and uses only BatchPlan::recWords(); this function is what the CPU test
// harness (tests/emu) runs and what tests/test_plan_emu.py checks the device bodies against.
inline BatchPlan makeBatchPlan(const uint8_t *seq, const uint64_t *off, uint64_t nReads, int ftabChars) {
    BatchPlan p;
    p.pass.assign(nReads + 1, 0);
    p.slotOf.assign(nReads + 1, kNone32);
    p.hitCap.assign(nReads + 1, 0);
    p.hitBase.assign(nReads + 1, 0);
    for (uint64_t r = 0; r < nReads; r++) {
        uint32_t nN;
        const uint64_t L = off[r + 1] - off[r];
        p.pass[r] = matePasses(seq + off[r], L, nN) ? 1 : 0;
        if (!p.pass[r]) continue;
        p.slotOf[r] = (uint32_t)p.items.size();
        p.items.push_back((uint32_t)r);
        p.maxLen = std::max<uint64_t>(p.maxLen, L);
        // Every partialSearch call either swallows >= ftabChars N-free bases or
        // ends on an N (hi_aligner.h:934-978), which bounds the hits per strand.
        p.hitCap[r] = (uint32_t)(nN + (L - nN) / (uint64_t)ftabChars + 2);
        p.hitBase[r] = p.hitsTotal;
        p.hitsTotal += 2ull * p.hitCap[r];
    }
    return p;
}

}  // namespace cfamd
