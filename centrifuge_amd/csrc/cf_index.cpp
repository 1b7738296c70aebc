// cf_index.cpp — parser for the reference's index file format (host side).
// Format facts cite /root/reference; the code is our own.
#include "cf_index.hpp"

#include <algorithm>
#include <cctype>
#include <cstring>
#include <memory>
#include <stdexcept>

namespace cfamd {

namespace {

struct FileCloser {
    void operator()(std::FILE *f) const { if (f) std::fclose(f); }
};
using File = std::unique_ptr<std::FILE, FileCloser>;

File openOrThrow(const std::string &p) {
    File f(std::fopen(p.c_str(), "rb"));
    if (!f) throw std::runtime_error("cannot open index file " + p);
    return f;
}

template <typename T>
T get(std::FILE *f, const std::string &what) {
    T v;
    if (std::fread(&v, sizeof v, 1, f) != 1) throw std::runtime_error("short read in " + what);
    return v;
}

void skip(std::FILE *f, uint64_t n, const std::string &what) {
    if (fseeko(f, static_cast<off_t>(n), SEEK_CUR) != 0) throw std::runtime_error("seek failed in " + what);
}

}  // namespace

const char *rankString(int rank) {
    static const char *const kNames[] = {
        "no rank", "strain", "species", "genus", "family", "order", "class", "phylum", "kingdom",
        "no rank" /* domain has no case in the reference's switch */, "forma", "infraclass", "infraorder",
        "parvorder", "subclass", "subfamily", "subgenus", "subkingdom", "suborder", "subphylum",
        "subspecies", "subtribe", "superclass", "superfamily", "superkingdom", "superorder",
        "superphylum", "tribe", "varietas", "life"};
    return (rank >= 0 && rank < 30) ? kNames[rank] : "no rank";
}

int rankToSlot(int rank) {
    switch (rank) {
        case 1: case 20: return 0;          // strain, subspecies
        case 2: return 1;                   // species
        case 3: return 2;                   // genus
        case 4: return 3;                   // family
        case 5: return 4;                   // order
        case 6: return 5;                   // class
        case 7: return 6;                   // phylum
        case 8: return 7;                   // kingdom
        case 24: return 8;                  // superkingdom
        case 9: return 9;                   // domain
        default: return -1;
    }
}

void HostIndex::load(const std::string &base, const SectionSink &sink) {
    load1(base + ".1.cf", sink);
    load2(base + ".2.cf", sink);
    load3(base + ".3.cf");
    load4(base + ".4.cf");
    // dense taxon universe
    taxa.clear();
    taxa.push_back(0);
    taxa.push_back(1);
    for (const auto &n : tree) taxa.push_back(n.tid);
    for (uint64_t t : uidTid) taxa.push_back(t);
    std::sort(taxa.begin(), taxa.end());
    taxa.erase(std::unique(taxa.begin(), taxa.end()), taxa.end());
}

// .1.cf: [i32 1][u64 len][i32 lineRate][i32 linesPerSide][i32 offRate][i32 ftabChars][i32 flags]
//        [u64 nPat][u64 plen[nPat]][u64 nFrag][u64 rstarts[3 nFrag]][sides][u64 zOff][u64 fchr[5]]
//        [u64 ftab[]][u64 eftab[]][names]          (bt2_io.h:138-526)
void HostIndex::load1(const std::string &path, const SectionSink &sink) {
    File f = openOrThrow(path);
    if (get<int32_t>(f.get(), path) != 1) throw std::runtime_error(path + ": not a little-endian 64-bit index");
    g.len = get<uint64_t>(f.get(), path);
    g.lineRate = get<int32_t>(f.get(), path);
    (void)get<int32_t>(f.get(), path);
    g.offRate = get<int32_t>(f.get(), path);
    g.ftabChars = get<int32_t>(f.get(), path);
    flags = get<int32_t>(f.get(), path);
    if (g.lineRate != 7) throw std::runtime_error(path + ": unsupported lineRate (128-byte sides expected)");
    if (g.ftabChars < 1 || g.ftabChars > 14 || g.offRate < 0 || g.offRate > 30)
        throw std::runtime_error(path + ": implausible header");
    const uint64_t bwtSz = g.len / 4 + 1, sideBwtSz = 128 - 32;
    g.numSides = (bwtSz + sideBwtSz - 1) / sideBwtSz;
    g.sidesBytes = g.numSides * 128;
    g.ftabLen = (1ull << (2 * g.ftabChars)) + 1;
    g.eftabLen = 2ull * static_cast<uint64_t>(g.ftabChars);
    g.offsLen = (g.len + 1 + (1ull << g.offRate) - 1) >> g.offRate;
    nPat = get<uint64_t>(f.get(), path);
    if (nPat > g.len + 1) throw std::runtime_error(path + ": implausible sequence count");
    plen.resize(nPat);
    if (nPat && std::fread(plen.data(), 8, nPat, f.get()) != nPat) throw std::runtime_error("short read in " + path + " (plen)");
    offw = nPat > 65535;
    const uint64_t nFrag = get<uint64_t>(f.get(), path);
    if (nFrag > g.len + 1) throw std::runtime_error(path + ": implausible fragment count");
    rstarts.resize(3 * nFrag);
    if (nFrag && std::fread(rstarts.data(), 8, 3 * nFrag, f.get()) != 3 * nFrag) throw std::runtime_error("short read in " + path + " (rstarts)");
    if (sink) sink(Section::Sides, f.get(), g.sidesBytes); else skip(f.get(), g.sidesBytes, path);
    zOff = get<uint64_t>(f.get(), path);
    for (auto &v : fchr) v = get<uint64_t>(f.get(), path);
    if (sink) {
        sink(Section::Ftab, f.get(), 8 * g.ftabLen);
        sink(Section::Eftab, f.get(), 8 * g.eftabLen);
    } else {
        // taxonomy-only view: the tables are not read, but a truncated file is still an error
        const off_t here = ftello(f.get());
        if (fseeko(f.get(), 0, SEEK_END) != 0 || ftello(f.get()) < here + static_cast<off_t>(8 * (g.ftabLen + g.eftabLen)))
            throw std::runtime_error("short read in " + path + " (ftab section)");
        if (fseeko(f.get(), here + static_cast<off_t>(8 * (g.ftabLen + g.eftabLen)), SEEK_SET) != 0)
            throw std::runtime_error("seek failed in " + path);
    }
    // reference names: '\n'-separated, ended by '\0' or the end of the file (bt2_io.h:746-763)
    refnames.clear();
    for (;;) {
        const int c = std::fgetc(f.get());
        if (c == EOF || c == 0) break;
        if (c == '\n') refnames.emplace_back();
        else { if (refnames.empty()) refnames.emplace_back(); refnames.back().push_back(static_cast<char>(c)); }
    }
    if (!refnames.empty() && refnames.back().empty()) refnames.pop_back();
}

// .2.cf: [i32 1][offs[offsLen]] with u16 or u32 elements (bt2_io.h:528-641)
void HostIndex::load2(const std::string &path, const SectionSink &sink) {
    if (!sink) return;
    File f = openOrThrow(path);
    if (get<int32_t>(f.get(), path) != 1) throw std::runtime_error(path + ": bad endian word");
    sink(Section::SaSample, f.get(), g.offsLen * (offw ? 4 : 2));
}

// .3.cf (bt2_idx.h:623-787)
void HostIndex::load3(const std::string &path) {
    File f = openOrThrow(path);
    (void)get<int32_t>(f.get(), path);
    const uint64_t nref = get<uint64_t>(f.get(), path);
    // every count read from the file is checked against the bytes that are left before it sizes anything
    const auto checkCount = [&](uint64_t count, uint64_t minBytesEach, const char *what) {
        const off_t here = ftello(f.get());
        if (fseeko(f.get(), 0, SEEK_END) != 0) throw std::runtime_error("seek failed in " + path);
        const off_t end = ftello(f.get());
        if (fseeko(f.get(), here, SEEK_SET) != 0) throw std::runtime_error("seek failed in " + path);
        if (count > (uint64_t)(end - here) / minBytesEach) throw std::runtime_error(path + ": " + what + " count exceeds the file size");
    };
    checkCount(nref, 9, "sequence");
    uid.clear(); uidTid.clear();
    uid.reserve(nref); uidTid.reserve(nref);
    uint64_t ncid = 0;
    for (uint64_t i = 0; i < nref; i++) {
        // The reference extracts uid bytes with formatted `>>`, which drops every
        // whitespace byte; only '\0' (or EOF) ends a uid.
        std::string u;
        for (;;) {
            int c = std::fgetc(f.get());
            if (c == EOF || c == 0) break;
            if (std::isspace(c)) continue;
            u.push_back(static_cast<char>(c));
        }
        if (u.compare(0, 3, "cid") == 0) ncid++;
        uid.push_back(std::move(u));
        uidTid.push_back(get<uint64_t>(f.get(), path));
    }
    compressed = ncid >= 10;
    const uint64_t ntid = get<uint64_t>(f.get(), path);
    checkCount(ntid, 18, "taxonomy node");
    tree.clear(); tree.reserve(ntid);
    for (uint64_t i = 0; i < ntid; i++) {
        TaxNode n{};
        n.tid = get<uint64_t>(f.get(), path);
        n.parent = get<uint64_t>(f.get(), path);
        n.rank = static_cast<uint8_t>(get<uint16_t>(f.get(), path));
        tree.push_back(n);
    }
    std::stable_sort(tree.begin(), tree.end(), [](const TaxNode &a, const TaxNode &b) { return a.tid < b.tid; });
    {   // map semantics: a repeated tid keeps the value assigned last
        size_t w = 0;
        for (size_t i = 0; i < tree.size(); i++) {
            if (w > 0 && tree[w - 1].tid == tree[i].tid) tree[w - 1] = tree[i];
            else tree[w++] = tree[i];
        }
        tree.resize(w);
    }
    for (uint64_t t : uidTid)
        if (auto *n = const_cast<TaxNode *>(findNode(t))) n->leaf = 1;
    const uint64_t nname = get<uint64_t>(f.get(), path);
    checkCount(nname, 9, "name");
    names.clear(); names.reserve(nname);
    for (uint64_t i = 0; i < nname; i++) {
        const uint64_t tid = get<uint64_t>(f.get(), path);
        std::string s;
        int c;
        while ((c = std::fgetc(f.get())) != EOF && std::isspace(c)) {}
        while (c != EOF && !std::isspace(c)) { s.push_back(c == '@' ? ' ' : static_cast<char>(c)); c = std::fgetc(f.get()); }
        names.emplace_back(tid, std::move(s));
    }
    std::stable_sort(names.begin(), names.end(), [](const auto &a, const auto &b) { return a.first < b.first; });
    const uint64_t nsize = get<uint64_t>(f.get(), path);
    checkCount(nsize, 16, "size");
    sizes.clear(); sizes.reserve(nsize);
    for (uint64_t i = 0; i < nsize; i++) {
        const uint64_t tid = get<uint64_t>(f.get(), path);
        const uint64_t sz = get<uint64_t>(f.get(), path);
        sizes.emplace_back(tid, sz);
    }
    std::stable_sort(sizes.begin(), sizes.end(), [](const auto &a, const auto &b) { return a.first < b.first; });
    rollUpSizes();
    buildPaths();
}

// Average genome size of species..phylum nodes over the below-species entries
// under them (bt2_idx.h:704-745).  "Below species" = rank numbered 0 by
// taxonomy.h:165-205: strain, subspecies and the never-numbered RANK_LIFE.
void HostIndex::rollUpSizes() {
    std::vector<uint64_t> sum(tree.size(), 0), cnt(tree.size(), 0);
    for (const auto &e : sizes) {
        uint64_t c = e.first;
        const TaxNode *nd = findNode(c);
        if (!nd || nd->parent == c) continue;
        const bool below = nd->rank == 1 || nd->rank == 20 || nd->rank == 29;
        if (!((nd->rank == 0 && nd->leaf) || below)) continue;
        c = nd->parent;
        for (size_t steps = 0;; steps++) {
            if (steps > tree.size()) throw std::runtime_error("taxonomy tree of the index has a parent cycle");
            const TaxNode *p = findNode(c);
            if (!p) break;
            if (p->rank >= 2 && p->rank <= 7) { sum[p - tree.data()] += e.second; cnt[p - tree.data()]++; }
            if (c == p->parent) break;
            c = p->parent;
        }
    }
    const size_t n0 = sizes.size();
    for (size_t t = 0; t < tree.size(); t++) {
        if (!cnt[t]) continue;
        const uint64_t tid = tree[t].tid, v = sum[t] / cnt[t];
        auto it = std::lower_bound(sizes.begin(), sizes.begin() + n0, tid,
                                   [](const auto &a, uint64_t k) { return a.first < k; });
        if (it != sizes.begin() + n0 && it->first == tid) it->second = v;
        else sizes.emplace_back(tid, v);
    }
    std::sort(sizes.begin(), sizes.end());
}

void HostIndex::buildPaths() {
    std::vector<uint64_t> tids(uidTid);
    std::sort(tids.begin(), tids.end());
    tids.erase(std::unique(tids.begin(), tids.end()), tids.end());
    pathTid.clear(); paths.clear();
    for (uint64_t leafTid : tids) {
        if (!findNode(leafTid)) continue;              // no path for a taxid absent from the tree
        std::array<uint64_t, kPathSlots> p{};
        uint64_t tid = leafTid;
        bool first = true;
        for (size_t steps = 0;; steps++) {
            if (steps > tree.size()) throw std::runtime_error("taxonomy tree of the index has a parent cycle");
            const TaxNode *nd = findNode(tid);
            if (!nd) break;
            const int slot = (first && nd->rank == 0) ? 0 : rankToSlot(nd->rank);
            if (slot >= 0 && p[slot] == 0) p[slot] = tid;
            first = false;
            if (nd->parent == tid) break;
            tid = nd->parent;
        }
        pathTid.push_back(leafTid);
        paths.push_back(p);
    }
}

// .4.cf: [i32 1][u64 m]{u64 saRow, u32 refIdx} x m  (bt2_idx.h:789-853); optional file
void HostIndex::load4(const std::string &path) {
    boundRow.clear(); boundRef.clear(); lastBoundary = 0;
    File f(std::fopen(path.c_str(), "rb"));
    if (!f) return;
    (void)get<int32_t>(f.get(), path);
    const uint64_t m = get<uint64_t>(f.get(), path);
    {
        const off_t here = ftello(f.get());
        if (fseeko(f.get(), 0, SEEK_END) != 0 || (uint64_t)(ftello(f.get()) - here) / 12 < m) throw std::runtime_error(path + ": boundary count exceeds the file size");
        if (fseeko(f.get(), here, SEEK_SET) != 0) throw std::runtime_error("seek failed in " + path);
    }
    std::vector<std::pair<uint64_t, uint32_t>> b;
    b.reserve(m);
    for (uint64_t i = 0; i < m; i++) {
        const uint64_t row = get<uint64_t>(f.get(), path);
        const uint32_t ref = get<uint32_t>(f.get(), path);
        b.emplace_back(row, ref);
        lastBoundary = std::max(lastBoundary, row);
    }
    std::stable_sort(b.begin(), b.end(), [](const auto &x, const auto &y) { return x.first < y.first; });
    for (size_t i = 0; i < b.size(); i++) {
        if (!boundRow.empty() && boundRow.back() == b[i].first) { boundRef.back() = b[i].second; continue; }
        boundRow.push_back(b[i].first);
        boundRef.push_back(b[i].second);
    }
}

const TaxNode *HostIndex::findNode(uint64_t tid) const {
    auto it = std::lower_bound(tree.begin(), tree.end(), tid, [](const TaxNode &n, uint64_t k) { return n.tid < k; });
    return (it != tree.end() && it->tid == tid) ? &*it : nullptr;
}

uint32_t HostIndex::findPath(uint64_t tid) const {
    auto it = std::lower_bound(pathTid.begin(), pathTid.end(), tid);
    return (it != pathTid.end() && *it == tid) ? static_cast<uint32_t>(it - pathTid.begin()) : kNoPath;
}

uint32_t HostIndex::taxonIndex(uint64_t tid) const {
    auto it = std::lower_bound(taxa.begin(), taxa.end(), tid);
    if (it == taxa.end() || *it != tid) throw std::logic_error("taxid outside the dense taxon table");
    return static_cast<uint32_t>(it - taxa.begin());
}

const char *HostIndex::name(uint64_t tid) const {
    auto it = std::lower_bound(names.begin(), names.end(), tid, [](const auto &a, uint64_t k) { return a.first < k; });
    // map semantics: last assignment wins among equal keys
    const char *r = "";
    for (; it != names.end() && it->first == tid; ++it) r = it->second.c_str();
    return r;
}

uint64_t HostIndex::size(uint64_t tid) const {
    auto it = std::lower_bound(sizes.begin(), sizes.end(), tid, [](const auto &a, uint64_t k) { return a.first < k; });
    return (it != sizes.end() && it->first == tid) ? it->second : 0;
}

bool HostIndex::inClosure(uint64_t tid, const uint64_t *list, int n) const {
    if (n <= 0 || !findNode(tid)) return false;
    uint64_t t = tid;
    for (size_t steps = 0;; steps++) {
        if (steps > tree.size()) return false;                   // a parent cycle (load3 rejects those; belt and braces)
        for (int i = 0; i < n; i++) if (list[i] == t) return true;
        const TaxNode *nd = findNode(t);
        if (!nd || nd->parent == t) return false;
        t = nd->parent;
    }
}

const char *HostIndex::formatSeqId(uint32_t uniqueId, uint64_t taxId) const {
    const TaxNode *nd = findNode(taxId);
    const bool leaf = nd ? nd->leaf != 0 : true;
    if (leaf && uniqueId < uid.size()) return uid[uniqueId].c_str();
    return rankString(nd ? nd->rank : 0);
}

}  // namespace cfamd
