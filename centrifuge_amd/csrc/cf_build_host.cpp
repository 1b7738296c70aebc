// cf_build_host.cpp — reference ingest and the SA-independent index-file writers.
// Behaviour restated from the reference's builder (files cited inline); code is our own.
#include "cf_build_host.hpp"

#include <algorithm>
#include <atomic>
#include <cstring>
#include <fstream>
#include <map>
#include <set>
#include <sstream>
#include <stdexcept>
#include <thread>

namespace cfamd {

namespace {

// alphabet.cpp:36-58: 1 = A/C/G/T (either case), 2 = IUPAC ambiguity code incl. N, 3 = '-'
uint8_t charCat(int c) {
    switch (c) {
        case 'A': case 'C': case 'G': case 'T': case 'a': case 'c': case 'g': case 't': return 1;
        case 'B': case 'D': case 'H': case 'K': case 'M': case 'N': case 'R': case 'S': case 'V': case 'W': case 'X': case 'Y':
        case 'b': case 'd': case 'h': case 'k': case 'm': case 'n': case 'r': case 's': case 'v': case 'w': case 'x': case 'y':
            return 2;
        case '-': return 3;
        default: return 0;
    }
}
uint8_t baseCode(int c) {
    switch (c) {
        case 'C': case 'c': return 1;
        case 'G': case 'g': return 2;
        case 'T': case 't': return 3;
        default: return 0;
    }
}

template <typename F>
void parallelFor(uint64_t n, F &&f) {
    unsigned nt = std::min<uint64_t>(std::max(1u, std::min(32u, std::thread::hardware_concurrency())), std::max<uint64_t>(1, n));
    if (nt <= 1) { for (uint64_t i = 0; i < n; i++) f(i); return; }
    std::atomic<uint64_t> next{0};
    std::vector<std::thread> th;
    for (unsigned t = 0; t < nt; t++)
        th.emplace_back([&] { for (uint64_t i; (i = next.fetch_add(1)) < n;) f(i); });
    for (auto &t : th) t.join();
}

// Records of one sequence given a classifier of its symbols into base / gap.
// Mirrors fastaRefReadSize (ref_read.cpp:28-186): alternating gap runs (off) and
// base runs (len); a trailing gap run becomes a record with len 0.
struct SeqScan {
    std::vector<RefRec> recs;
    uint64_t bases = 0, total = 0;
};

void finishSequences(JoinedRef &out, const std::vector<SeqScan> &scans, const std::vector<std::string> &names) {
    out.szs.clear(); out.refnames.clear(); out.plen.clear(); out.rstarts.clear(); out.seqJoinedStart.clear();
    uint64_t tot = 0;
    for (size_t s = 0; s < scans.size(); s++) {
        if (scans[s].bases == 0) {
            // A sequence of gaps only never becomes a pattern: its name is dropped (bt2_idx.h:3318-3322) and
            // szsToDisk adds its length to the pattern before it (bt2_idx.h:3273-3281).  With no pattern before it the
            // reference adds to plen[-1] — a stray write whose visible effect is that the length is simply lost.
            if (!out.plen.empty()) out.plen.back() += scans[s].total;
            for (const RefRec &r : scans[s].recs) out.szs.push_back(r);
            continue;
        }
        // an empty name is replaced by the sequence's ordinal (bt2_idx.h:3310-3316)
        out.refnames.push_back(names[s].empty() ? std::to_string(out.refnames.size()) : names[s]);
        out.plen.push_back(scans[s].total);
        out.seqJoinedStart.push_back(tot);
        uint64_t off = 0;
        for (const RefRec &r : scans[s].recs) {
            out.szs.push_back(r);
            if (r.len == 0) continue;                      // szsToDisk bt2_io.h:1000
            if (r.first) off = 0;
            off += r.off;
            out.rstarts.push_back(tot);
            out.rstarts.push_back(out.plen.size() - 1);
            out.rstarts.push_back(off);
            tot += r.len;
            off += r.len;
        }
    }
    out.len = tot;
    out.nPat = out.plen.size();
    out.nFrag = out.rstarts.size() / 3;
    if (out.nPat == 0 || tot == 0) throw std::runtime_error("no reference sequence");
}

}  // namespace

void ingestFasta(const std::vector<std::string> &paths, JoinedRef &out) {
    std::vector<SeqScan> scans;
    std::vector<std::string> names;
    out.store.clear();
    for (const std::string &p : paths) {
        std::FILE *f = std::fopen(p.c_str(), "rb");
        if (!f) throw std::runtime_error("cannot open reference file " + p);
        std::vector<char> buf(1 << 22);
        bool inHeader = false, haveSeq = false;
        std::string name;
        SeqScan cur;
        uint64_t gap = 0, run = 0;
        bool firstRec = true;
        auto closeRun = [&] {
            if (run) { cur.recs.push_back(RefRec{gap, run, firstRec}); firstRec = false; cur.bases += run; cur.total += gap + run; gap = run = 0; }
        };
        auto closeSeq = [&] {
            if (!haveSeq) return;
            closeRun();
            if (gap) { cur.recs.push_back(RefRec{gap, 0, firstRec}); cur.total += gap; gap = 0; }
            if (cur.total == 0) {
                // An empty record: the reference drops its name (ref_read.h: consecutive '>' lines)
            } else { scans.push_back(std::move(cur)); names.push_back(name); }
            cur = SeqScan{}; firstRec = true; haveSeq = false;
        };
        size_t got;
        while ((got = std::fread(buf.data(), 1, buf.size(), f)) > 0) {
            for (size_t i = 0; i < got; i++) {
                const int c = static_cast<unsigned char>(buf[i]);
                if (inHeader) {
                    if (c == '\n' || c == '\r') inHeader = false; else name.push_back(static_cast<char>(c));
                    continue;
                }
                if (c == '>') { closeSeq(); inHeader = true; haveSeq = true; name.clear(); continue; }
                if (!haveSeq) {
                    if (c == '\n' || c == '\r' || c == ' ' || c == '\t') continue;
                    std::fclose(f);
                    throw std::runtime_error("reference file does not seem to be a FASTA file: " + p);
                }
                const uint8_t cat = charCat(c);
                if (cat == 1) {
                    run++;
                    out.store.push_back(baseCode(c));
                } else if (cat >= 2) {
                    closeRun();
                    gap++;
                }
            }
        }
        closeSeq();
        std::fclose(f);
    }
    finishSequences(out, scans, names);
    out.text = out.store.data();
}

void ingestMemory(const uint8_t *codes, const uint64_t *seqOff, const char *const *namesIn, uint64_t nSeq, JoinedRef &out) {
    if (!codes || !seqOff || !namesIn || nSeq == 0) throw std::runtime_error("no in-memory sequences given");
    // the offsets are lengths by difference: check them before anything subtracts them
    for (uint64_t s = 0; s < nSeq; s++)
        if (seqOff[s + 1] < seqOff[s]) throw std::runtime_error("sequence offsets must be non-decreasing");
    std::vector<SeqScan> scans(nSeq);
    std::vector<std::string> names(nSeq);
    std::atomic<uint64_t> gapsTotal{0};
    parallelFor(nSeq, [&](uint64_t s) {
        const uint8_t *p = codes + seqOff[s];
        const uint64_t L = seqOff[s + 1] - seqOff[s];
        names[s] = namesIn[s] ? namesIn[s] : "";
        SeqScan &sc = scans[s];
        uint64_t amb = 0;
        for (uint64_t i = 0; i < L; i++) amb += p[i] > 3;
        sc.total = L; sc.bases = L - amb;
        if (amb == 0) { if (L) sc.recs.push_back(RefRec{0, L, true}); return; }
        gapsTotal += amb;
        uint64_t gap = 0, run = 0; bool first = true;
        for (uint64_t i = 0; i < L; i++) {
            if (p[i] <= 3) run++;
            else { if (run) { sc.recs.push_back(RefRec{gap, run, first}); first = false; gap = run = 0; } gap++; }
        }
        if (run) { sc.recs.push_back(RefRec{gap, run, first}); first = false; gap = 0; }
        if (gap) sc.recs.push_back(RefRec{gap, 0, first});
    });
    finishSequences(out, scans, names);
    if (gapsTotal.load() == 0) {
        out.store.clear();
        out.text = codes + seqOff[0];                      // gap-free: the joined text is the input itself
        return;
    }
    out.store.resize(out.len);
    // Destination of INPUT sequence s = the bases of the sequences before it.  (out.seqJoinedStart is indexed by
    // kept pattern: all-gap and empty sequences have no entry there, so it must not be indexed by s.)
    std::vector<uint64_t> dstOf(nSeq + 1, 0);
    for (uint64_t s = 0; s < nSeq; s++) dstOf[s + 1] = dstOf[s] + scans[s].bases;
    if (dstOf[nSeq] != out.len) throw std::runtime_error("internal: joined length mismatch");
    parallelFor(nSeq, [&](uint64_t s) {
        const uint8_t *p = codes + seqOff[s];
        uint8_t *d = out.store.data() + dstOf[s];
        uint64_t pos = 0;
        for (const RefRec &r : scans[s].recs) { pos += r.off; std::memcpy(d, p + pos, r.len); d += r.len; pos += r.len; }
    });
    out.text = out.store.data();
}

namespace {

// bt2_idx.h:2999-3009: the uid is the header up to the first blank or the second '|'
std::string uidOf(const std::string &header) {
    size_t nd = 0, j = 0;
    for (; j < header.size(); j++) {
        if (header[j] == ' ') break;
        if (header[j] == '|') nd++;
        if (nd == 2) break;
    }
    return header.substr(0, j);
}

uint64_t tidOf(const std::string &s) {                       // bt2_idx.h:3011-3027
    uint64_t t1 = 0, t2 = 0;
    bool dot = false;
    for (char ch : s) {
        if (ch == '.') { dot = true; continue; }
        const uint32_t num = static_cast<uint32_t>(ch - '0');
        if (dot) t2 = t2 * 10 + num; else t1 = t1 * 10 + num;
    }
    return t1 | (t2 << 32);
}

uint8_t rankId(const std::string &r) {                        // taxonomy.h:241-301, enum :15-47
    static const char *const kNames[] = {"", "strain", "species", "genus", "family", "order", "class", "phylum", "kingdom", "",
                                         "forma", "infraclass", "infraorder", "parvorder", "subclass", "subfamily", "subgenus",
                                         "subkingdom", "suborder", "subphylum", "subspecies", "subtribe", "superclass",
                                         "superfamily", "superkingdom", "superorder", "superphylum", "tribe", "varietas", "life"};
    for (int i = 1; i < 30; i++) if (i != 9 && r == kNames[i]) return static_cast<uint8_t>(i);
    return 0;
}

struct TreeNode { uint64_t parent; uint8_t rank; };

}  // namespace

void writeTaxonomyFile(const std::string &path, const JoinedRef &ref, const char *conversionTable, const char *taxonomyTree,
                       const char *nameTable, const char *sizeTable) {
    // ---- uid -> taxid (bt2_idx.h:1330-1366): token stream, first assignment wins
    std::set<std::string> uids;
    for (const auto &n : ref.refnames) uids.insert(uidOf(n));
    std::map<std::string, uint64_t> uidToTid;
    {
        std::ifstream in(conversionTable ? conversionTable : "");
        if (!in.is_open()) throw std::runtime_error(std::string("cannot open conversion table ") + (conversionTable ? conversionTable : "(null)"));
        while (!in.eof()) {
            std::string uid;
            in >> uid;
            if (uid.empty() || uid[0] == '#') continue;
            std::string stid;
            in >> stid;
            const uint64_t tid = tidOf(stid);
            if (!uids.count(uid)) continue;
            if (uidToTid.count(uid)) continue;
            uidToTid[uid] = tid;
        }
    }
    std::FILE *f = std::fopen(path.c_str(), "wb");
    if (!f) throw std::runtime_error("cannot open index file for writing: " + path);
    try {
        std::set<uint64_t> tids;
        put<int32_t>(f, 1);
        put<uint64_t>(f, ref.refnames.size());
        for (const auto &n : ref.refnames) {
            const std::string uid = uidOf(n);
            putBytes(f, uid.data(), uid.size());
            put<uint8_t>(f, 0);
            auto it = uidToTid.find(uid);
            if (it != uidToTid.end()) { put<uint64_t>(f, it->second); tids.insert(it->second); }
            else put<uint64_t>(f, 0);
        }
        // ---- taxonomy tree (taxonomy.h:322-348), pruned to the ancestors of used taxids (bt2_idx.h:1395-1421)
        std::map<uint64_t, TreeNode> tree;
        {
            std::ifstream in(taxonomyTree ? taxonomyTree : "");
            if (!in.is_open()) throw std::runtime_error(std::string("cannot open taxonomy tree ") + (taxonomyTree ? taxonomyTree : "(null)"));
            std::string line;
            while (std::getline(in, line)) {
                if (line.empty() || line[0] == '#') continue;
                std::istringstream cl(line);
                uint64_t tid = 0, parent = 0; char dummy; std::string rank;
                cl >> tid >> dummy >> parent >> dummy >> rank;
                if (tree.count(tid)) continue;
                tree[tid] = TreeNode{parent, rankId(rank)};
            }
        }
        std::set<uint64_t> color;
        for (uint64_t t : tids) {
            uint64_t tid = t;
            while (tree.count(tid)) {
                const uint64_t par = tree[tid].parent;
                color.insert(tid);
                if (par == tid) break;
                tid = par;
            }
        }
        put<uint64_t>(f, color.size());
        for (uint64_t tid : color) {
            put<uint64_t>(f, tid);
            put<uint64_t>(f, tree[tid].parent);
            put<uint16_t>(f, tree[tid].rank);
        }
        // ---- names (bt2_idx.h:1423-1462): scientific names of the kept nodes, blanks become '@'
        std::map<uint64_t, std::string> names;
        if (nameTable && nameTable[0]) {
            std::ifstream in(nameTable);
            if (!in.is_open()) throw std::runtime_error(std::string("cannot open name table ") + nameTable);
            std::string line;
            while (std::getline(in, line)) {
                if (line.empty() || line[0] == '#') continue;
                if (line.find("scientific name") == std::string::npos) continue;
                std::istringstream cl(line);
                uint64_t tid = 0; char dummy; std::string sci;
                cl >> tid >> dummy >> sci;
                if (!color.count(tid)) continue;
                std::string tmp;
                while (cl >> tmp) {
                    if (tmp == "|") break;
                    sci.push_back('@');
                    sci += tmp;
                }
                names[tid] = sci;
            }
        }
        put<uint64_t>(f, names.size());
        for (const auto &kv : names) {
            put<uint64_t>(f, kv.first);
            putBytes(f, kv.second.data(), kv.second.size());
            put<uint8_t>(f, '\n');
        }
        // ---- sizes (bt2_idx.h:1464-1504)
        std::map<uint64_t, uint64_t> sizes;
        for (size_t i = 0; i < ref.refnames.size(); i++) {
            auto it = uidToTid.find(uidOf(ref.refnames[i]));
            if (it == uidToTid.end()) continue;
            sizes[it->second] += ref.plen[i];
        }
        if (sizeTable && sizeTable[0]) {
            std::ifstream in(sizeTable);
            if (!in.is_open()) throw std::runtime_error(std::string("cannot open size table ") + sizeTable);
            while (!in.eof()) {
                std::string stid;
                in >> stid;
                if (stid.empty() || stid[0] == '#') continue;
                uint64_t sz = 0;
                in >> sz;
                sizes[tidOf(stid)] = sz;
            }
        }
        put<uint64_t>(f, sizes.size());
        for (const auto &kv : sizes) { put<uint64_t>(f, kv.first); put<uint64_t>(f, kv.second); }
    } catch (...) { std::fclose(f); throw; }
    if (std::fclose(f) != 0) throw std::runtime_error("error closing " + path);
}

}  // namespace cfamd

// Host-only half of an index build: the sequence bookkeeping of the FASTA / in-memory input and <out>.3.cf
// (uid table, pruned taxonomy, names, sizes).  No device involved; the CPU tests compare the file with the
// reference builder's on awkward taxonomies.
extern "C" cf_status cf_build_taxonomy(const cf_build_input *in, const char *outBase, char *err, uint64_t errCap) {
    auto fail = [&](cf_status st, const std::string &m) { if (err && errCap) { std::snprintf(err, (size_t)errCap, "%s", m.c_str()); } return st; };
    if (!in || !outBase) return fail(CF_ERR_ARG, "bad argument");
    try {
        cfamd::JoinedRef ref;
        if (in->fasta_paths && in->n_fasta > 0) {
            std::vector<std::string> paths(in->fasta_paths, in->fasta_paths + in->n_fasta);
            cfamd::ingestFasta(paths, ref);
        } else cfamd::ingestMemory(in->codes, in->seq_off, in->seq_names, in->n_seq, ref);
        cfamd::writeTaxonomyFile(std::string(outBase) + ".3.cf", ref, in->conversion_table, in->taxonomy_tree, in->name_table, in->size_table);
        return CF_OK;
    } catch (const std::bad_alloc &) { return fail(CF_ERR_NOMEM, "out of host memory");
    } catch (const std::exception &e) {
        const std::string m = e.what();
        return fail(m.find("cannot open") != std::string::npos ? CF_ERR_IO : CF_ERR_FORMAT, m);
    }
}

// The builder's view of its input, for tests: what goes into the header and name section of <base>.1.cf and the
// joined text itself.  File: u64 len, u64 nPat, plen[nPat], u64 nFrag, rstarts[3 nFrag], names ('\n' after each,
// then '\0'), text[len] (codes 0..3).  No device involved.
extern "C" cf_status cf_build_describe(const cf_build_input *in, const char *path, char *err, uint64_t errCap) {
    auto fail = [&](cf_status st, const std::string &m) { if (err && errCap) { std::snprintf(err, (size_t)errCap, "%s", m.c_str()); } return st; };
    if (!in || !path) return fail(CF_ERR_ARG, "bad argument");
    try {
        cfamd::JoinedRef ref;
        if (in->fasta_paths && in->n_fasta > 0) {
            std::vector<std::string> paths(in->fasta_paths, in->fasta_paths + in->n_fasta);
            cfamd::ingestFasta(paths, ref);
        } else cfamd::ingestMemory(in->codes, in->seq_off, in->seq_names, in->n_seq, ref);
        std::FILE *f = std::fopen(path, "wb");
        if (!f) return fail(CF_ERR_IO, std::string("cannot open ") + path);
        auto put64 = [&](uint64_t v) { std::fwrite(&v, 8, 1, f); };
        put64(ref.len); put64(ref.nPat);
        if (!ref.plen.empty()) std::fwrite(ref.plen.data(), 8, ref.plen.size(), f);
        put64(ref.nFrag);
        if (!ref.rstarts.empty()) std::fwrite(ref.rstarts.data(), 8, ref.rstarts.size(), f);
        for (const auto &nm : ref.refnames) { std::fwrite(nm.data(), 1, nm.size(), f); std::fputc('\n', f); }
        std::fputc(0, f);
        const uint8_t *t = ref.text ? ref.text : ref.store.data();
        if (ref.len) std::fwrite(t, 1, ref.len, f);
        if (std::fclose(f) != 0) return fail(CF_ERR_IO, "error closing the description file");
        return CF_OK;
    } catch (const std::bad_alloc &) { return fail(CF_ERR_NOMEM, "out of host memory");
    } catch (const std::exception &e) {
        const std::string m = e.what();
        return fail(m.find("cannot open") != std::string::npos ? CF_ERR_IO : CF_ERR_FORMAT, m);
    }
}
