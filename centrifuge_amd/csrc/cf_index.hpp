// cf_index.hpp — host-side view of a Centrifuge index (<base>.{1,2,3,4}.cf).
//
// Reads the on-disk format written by the reference's centrifuge-build
// (layout: bt2_io.h:138-685 for .1/.2.cf, bt2_idx.h:623-853 for .3/.4.cf;
// SURVEY.md §3.4) into flat POD arrays.  The multi-gigabyte sections (BWT
// sides, ftab, SA sample) are not kept on the host: they are handed to a
// sink as (section, FILE*, byte count) so the device layer can stream them
// straight into HBM.
#pragma once
#include <array>
#include <cstdint>
#include <cstdio>
#include <functional>
#include <string>
#include <unordered_map>
#include <vector>

namespace cfamd {

constexpr uint64_t kMask64 = ~0ull;
constexpr int kPathSlots = 10;            // taxonomy.h:63
constexpr uint32_t kNoPath = 0xffffffffu;

enum class Section { Sides, Ftab, Eftab, SaSample };
// The sink must consume exactly `bytes` bytes from `f`.
using SectionSink = std::function<void(Section, std::FILE *f, uint64_t bytes)>;

struct TaxNode {
    uint64_t tid, parent;
    uint8_t rank;      // taxonomy.h:15-47 enum order
    uint8_t leaf;      // tid occurs in the uid table (bt2_idx.h:673)
};

struct Geometry {                         // EbwtParams::init bt2_idx.h:133-167
    uint64_t len = 0;
    int32_t lineRate = 0, offRate = 0, ftabChars = 0;
    uint64_t numSides = 0, sidesBytes = 0, ftabLen = 0, eftabLen = 0, offsLen = 0;
};

class HostIndex {
public:
    // Throws std::runtime_error on I/O or format errors.  With a null sink the
    // big sections are skipped (taxonomy-only view).
    void load(const std::string &base, const SectionSink &sink);

    Geometry g;
    uint64_t nPat = 0, zOff = 0;
    int32_t flags = 0;                    // header word as stored (negative: -flags holds the EBWT_* bits)
    std::vector<uint64_t> plen;           // unambiguous length of every reference sequence
    std::vector<uint64_t> rstarts;        // 3 per fragment: joined offset, sequence idx, offset in the sequence (bt2_idx.h:2016-2062)
    std::vector<std::string> refnames;    // trailing name section of .1.cf (bt2_io.h:746-763)
    uint64_t fchr[5] = {0, 0, 0, 0, 0};
    bool offw = false;                    // SA sample is u32 (nPat > 65535), bt2_io.h:280
    bool compressed = false;              // >= 10 uids start with "cid", bt2_idx.h:648-663

    std::vector<std::string> uid;         // per reference sequence
    std::vector<uint64_t> uidTid;
    std::vector<TaxNode> tree;            // sorted by tid
    std::vector<std::pair<uint64_t, std::string>> names;   // sorted by tid
    std::vector<std::pair<uint64_t, uint64_t>> sizes;      // sorted by tid, after roll-up
    // path table (TaxonomyPathTable::buildPaths taxonomy.h:96-149), sorted by tid
    std::vector<uint64_t> pathTid;
    std::vector<std::array<uint64_t, kPathSlots>> paths;
    // .4.cf genome-boundary rows (bt2_idx.h:789-853), sorted by row
    std::vector<uint64_t> boundRow;
    std::vector<uint32_t> boundRef;
    uint64_t lastBoundary = 0;
    // dense taxon table for counters: sorted unique {0, 1} ∪ tree ∪ uid taxids
    std::vector<uint64_t> taxa;

    const TaxNode *findNode(uint64_t tid) const;
    uint32_t findPath(uint64_t tid) const;            // index into paths or kNoPath
    uint32_t taxonIndex(uint64_t tid) const;          // index into taxa (must exist)
    const char *name(uint64_t tid) const;
    uint64_t size(uint64_t tid) const;
    // tid is a tree node with one of `list` on its root path (classifier.h:157-201)
    bool inClosure(uint64_t tid, const uint64_t *list, int n) const;
    // seqID column: classifier.h:546-557 + aln_sink.h:2219-2234
    const char *formatSeqId(uint32_t uniqueId, uint64_t taxId) const;

private:
    void load1(const std::string &path, const SectionSink &sink);
    void load2(const std::string &path, const SectionSink &sink);
    void load3(const std::string &path);
    void load4(const std::string &path);
    void buildPaths();
    void rollUpSizes();
};

const char *rankString(int rank);                     // taxonomy.h:207-239
int rankToSlot(int rank);                             // taxonomy.h:66-93, -1 if none

}  // namespace cfamd
