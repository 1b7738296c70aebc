// cf_report.cpp — per-taxon summary of a run: the counters and the `observed`
// multiset of SpeciesMetrics::addSpeciesCounts (aln_sink.h:142-172), the
// SQUAREM-accelerated EM abundance of calculateAbundance (aln_sink.h:196-495) and
// the report file of centrifuge.cpp:3231-3319.  Host code;
// the loops keep the reference's iteration order (std::map order) so that the
// double arithmetic, and with it the 6-digit abundance column, comes out the same.
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <limits>
#include <map>
#include <set>
#include <string>
#include <vector>

#include "../../include/centrifuge_amd.h"
#include "cf_index.hpp"

using namespace cfamd;

const HostIndex &cf_index_host(const cf_index *);   // cf_device.hip

namespace {

struct Counts { uint64_t nReads = 0, nUnique = 0; };

// key order of SpeciesMetrics::IDs (aln_sink.h:60-70): shorter tuples first, then element-wise
struct IdsLess {
    bool operator()(const std::vector<uint64_t> &a, const std::vector<uint64_t> &b) const {
        if (a.size() != b.size()) return a.size() < b.size();
        for (size_t i = 0; i < a.size(); i++) if (a[i] != b[i]) return a[i] < b[i];
        return false;
    }
};
using Observed = std::map<std::vector<uint64_t>, uint64_t, IdsLess>;

// one E+M step (aln_sink.h:196-272)
void emStep(const Observed &observed, const std::map<uint64_t, std::vector<uint64_t>> &ancestors,
            const std::map<uint64_t, uint64_t> &tidToNum, const std::vector<double> &p, std::vector<double> &pn,
            const std::vector<size_t> &len) {
    std::fill(pn.begin(), pn.end(), 0.0);
    for (const auto &kv : observed) {
        const std::vector<uint64_t> &ids = kv.first;
        const uint64_t count = kv.second;
        double psum = 0.0;
        for (uint64_t tid : ids) {
            auto it = tidToNum.find(tid);
            if (it != tidToNum.end()) psum += p[it->second];
            else {
                auto a = ancestors.find(tid);
                if (a == ancestors.end()) continue;
                for (uint64_t c : a->second) {
                    auto ci = tidToNum.find(c);
                    if (ci == tidToNum.end()) continue;
                    psum += p[ci->second];
                }
            }
        }
        if (psum == 0.0) continue;
        for (uint64_t tid : ids) {
            auto it = tidToNum.find(tid);
            if (it != tidToNum.end()) pn[it->second] += (count * (p[it->second] / psum));
            else {
                auto a = ancestors.find(tid);
                if (a == ancestors.end()) continue;
                for (uint64_t c : a->second) {
                    auto ci = tidToNum.find(c);
                    if (ci == tidToNum.end()) continue;
                    pn[ci->second] += (count * (p[ci->second] / psum));
                }
            }
        }
    }
    double sum = 0.0;
    for (size_t i = 0; i < pn.size(); i++) sum += (pn[i] / len[i]);
    for (size_t i = 0; i < pn.size(); i++) pn[i] = pn[i] / len[i] / sum;
}

}  // namespace

struct cf_report {
    const HostIndex *h = nullptr;
    std::map<uint64_t, Counts> counts;
    Observed observed;
    // Hot-path accumulators indexed by the dense taxon index of cf_row (no map lookups per row);
    // flush() folds them into the ordered maps the EM and the writers iterate.
    std::vector<Counts> dense;
    std::vector<uint64_t> denseSingle;       // observed tuples of size one

    void flush() {
        for (size_t i = 0; i < dense.size(); i++) {
            if (dense[i].nReads || dense[i].nUnique) { Counts &c = counts[h->taxa[i]]; c.nReads += dense[i].nReads; c.nUnique += dense[i].nUnique; dense[i] = Counts{}; }
            if (denseSingle[i]) { observed[std::vector<uint64_t>{h->taxa[i]}] += denseSingle[i]; denseSingle[i] = 0; }
        }
    }
    std::map<uint64_t, double> abundanceLen;
    size_t emIterations = 0;
    double emDiff = 0.0;

    bool sizeOf(uint64_t tid, uint64_t &out) const {
        auto it = std::lower_bound(h->sizes.begin(), h->sizes.end(), tid, [](const auto &a, uint64_t k) { return a.first < k; });
        if (it == h->sizes.end() || it->first != tid) return false;
        out = it->second;
        return true;
    }

    void calculateAbundance() {                                       // aln_sink.h:274-495
        std::set<uint64_t> leaves;
        for (const auto &kv : observed)
            for (uint64_t tid : kv.first) {
                const TaxNode *nd = h->findNode(tid);
                if (!nd || !nd->leaf) continue;
                leaves.insert(tid);
            }
        std::map<uint64_t, std::vector<uint64_t>> ancestors;
        for (const auto &kv : observed)
            for (uint64_t tid : kv.first) {
                if (leaves.count(tid) || ancestors.count(tid)) continue;
                std::vector<uint64_t> &ch = ancestors[tid];
                for (uint64_t leaf : leaves) {
                    uint64_t t = leaf;
                    for (;;) {
                        const TaxNode *nd = h->findNode(t);
                        if (!nd) break;
                        if (tid == nd->parent) ch.push_back(leaf);
                        if (t == nd->parent) break;
                        t = nd->parent;
                    }
                }
                std::sort(ch.begin(), ch.end());
            }
        std::map<uint64_t, uint64_t> tidToNum;
        std::vector<double> p;
        std::vector<size_t> len;
        for (const auto &kv : observed) {
            const std::vector<uint64_t> &ids = kv.first;
            const uint64_t count = kv.second;
            for (uint64_t tid : ids) {
                if (!leaves.count(tid)) continue;
                auto it = tidToNum.find(tid);
                if (it == tidToNum.end()) {
                    tidToNum[tid] = p.size();
                    p.push_back(1.0 / ids.size() * count);
                    uint64_t sz;
                    len.push_back(sizeOf(tid, sz) ? (size_t)sz : std::numeric_limits<size_t>::max());
                } else p[it->second] += (1.0 / ids.size() * count);
            }
        }
        {
            double sum = 0.0;
            for (size_t i = 0; i < p.size(); i++) sum += (p[i] / len[i]);
            for (size_t i = 0; i < p.size(); i++) p[i] = (p[i] / len[i]) / sum;
        }
        std::vector<double> pn(p.size()), pn2(p.size()), pr(p.size()), pv(p.size());
        size_t iter = 0;
        double diff = 0.0;
        for (;;) {                                                     // SQUAREM iteration :409-443
            emStep(observed, ancestors, tidToNum, p, pn, len);
            emStep(observed, ancestors, tidToNum, pn, pn2, len);
            double ssr = 0.0, ssv = 0.0;
            for (size_t i = 0; i < p.size(); i++) {
                pr[i] = pn[i] - p[i];
                ssr += (pr[i] * pr[i]);
                pv[i] = pn2[i] - pn[i] - pr[i];
                ssv += (pv[i] * pv[i]);
            }
            if (ssv > 0.0) {
                const double gamma = -std::sqrt(ssr / ssv);
                for (size_t i = 0; i < p.size(); i++) pn2[i] = std::max(0.0, p[i] - 2 * gamma * pr[i] + gamma * gamma * pv[i]);
                emStep(observed, ancestors, tidToNum, pn2, pn, len);
            }
            diff = 0.0;
            for (size_t i = 0; i < p.size(); i++) diff += (p[i] > pn[i] ? p[i] - pn[i] : pn[i] - p[i]);
            if (diff < 0.0000000001) break;
            if (++iter >= 10000) break;
            p = pn;
        }
        emIterations = iter; emDiff = diff;
        abundanceLen.clear();
        for (const auto &kv : tidToNum) abundanceLen[kv.first] = p[kv.second];
    }
};

extern "C" {

cf_status cf_report_create(const cf_index *ix, cf_report **out) {
    if (!ix || !out) return CF_ERR_ARG;
    *out = nullptr;
    try {
        auto *r = new cf_report();
        r->h = &cf_index_host(ix);
        *out = r;
        return CF_OK;
    } catch (...) { return CF_ERR_NOMEM; }
}

void cf_report_destroy(cf_report *r) { delete r; }

cf_status cf_report_add(cf_report *r, const cf_row *rows, const uint32_t *nRows, const uint32_t *maxScore, uint64_t nQueries,
                        uint32_t khits) {
    if (!r || !nRows || !maxScore || (!rows && khits)) return CF_ERR_ARG;
    try {
        const size_t nTaxa = r->h->taxa.size();
        if (r->dense.size() != nTaxa) { r->dense.assign(nTaxa, Counts{}); r->denseSingle.assign(nTaxa, 0); }
        const uint32_t idxZero = r->h->taxonIndex(0);
        std::vector<uint64_t> ids;
        const cf_row *packed = rows;                  // khits == 0: rows back to back (cf_batch_results_compact)
        for (uint64_t q = 0; q < nQueries; q++) {
            const uint32_t n = nRows[q];
            if (n == 0) {                       // the "unclassified" row: taxID 0, score 0, max_score 0 (classifier.h:619-626)
                Counts &c = r->dense[idxZero];
                c.nReads++; c.nUnique++;
                r->denseSingle[idxZero]++;
                continue;
            }
            const cf_row *row = khits ? rows + q * (uint64_t)khits : packed;
            packed += n;
            if (n == 1 && row->taxon_idx < nTaxa) {                      // the common case: one assignment
                Counts &c = r->dense[row->taxon_idx];
                c.nReads++; c.nUnique++;
                if (maxScore[q] != 0xffffffffu && row->score >= maxScore[q]) r->denseSingle[row->taxon_idx]++;   // only perfect hits feed the EM
                // (0xffffffff: a max_score of 2^32 or more — int64_t in the reference, aln_sink.h:145-160 — which no 32-bit score reaches)
                continue;
            }
            ids.clear();
            for (uint32_t i = 0; i < n; i++) {
                if (row[i].taxon_idx < nTaxa) { Counts &c = r->dense[row[i].taxon_idx]; c.nReads++; if (n == 1) c.nUnique++; }
                else { Counts &c = r->counts[row[i].tax_id]; c.nReads++; if (n == 1) c.nUnique++; }
                if (maxScore[q] != 0xffffffffu && row[i].score >= maxScore[q]) ids.push_back(row[i].tax_id);
            }
            if (ids.size() == n) {
                std::sort(ids.begin(), ids.end());
                r->observed[ids]++;
            }
        }
        return CF_OK;
    } catch (...) { return CF_ERR_NOMEM; }
}

// cf_report_add over the narrow rows of a batch (aln_sink.h:142-172 once more): the taxID of a row is its taxon's, max_score
// follows from the query's byte and the mates' lengths (classifier.h:530-536)
cf_status cf_report_add_narrow(cf_report *r, const cf_row16 *rows, const uint8_t *qinfo, const uint32_t *len, uint32_t uniformLen,
                               int paired, uint64_t nQueries) {
    if (!r || (nQueries && !qinfo)) return CF_ERR_ARG;
    try {
        const size_t nTaxa = r->h->taxa.size();
        if (r->dense.size() != nTaxa) { r->dense.assign(nTaxa, Counts{}); r->denseSingle.assign(nTaxa, 0); }
        const uint32_t idxZero = r->h->taxonIndex(0);
        std::vector<uint64_t> ids;
        const cf_row16 *row = rows;
        for (uint64_t q = 0; q < nQueries; q++) {
            const uint32_t n = qinfo[q] & 0x3fu;
            if (n == 0) {
                Counts &c = r->dense[idxZero];
                c.nReads++; c.nUnique++;
                r->denseSingle[idxZero]++;
                continue;
            }
            if (!rows) return CF_ERR_ARG;
            const uint64_t r0 = paired ? 2 * q : q;
            const uint32_t ms = cf_narrow_max_score(qinfo[q], len ? len[r0] : uniformLen, paired ? (len ? len[r0 + 1] : uniformLen) : 0u, paired);
            if (n == 1) {
                if (row->taxon_idx >= nTaxa) return CF_ERR_ARG;
                Counts &c = r->dense[row->taxon_idx];
                c.nReads++; c.nUnique++;
                if (ms != 0xffffffffu && row->score >= ms) r->denseSingle[row->taxon_idx]++;
                row++;
                continue;
            }
            ids.clear();
            for (uint32_t i = 0; i < n; i++) {
                if (row[i].taxon_idx >= nTaxa) return CF_ERR_ARG;
                r->dense[row[i].taxon_idx].nReads++;
                if (ms != 0xffffffffu && row[i].score >= ms) ids.push_back(r->h->taxa[row[i].taxon_idx]);
            }
            if (ids.size() == n) {
                std::sort(ids.begin(), ids.end());
                r->observed[ids]++;
            }
            row += n;
        }
        return CF_OK;
    } catch (...) { return CF_ERR_NOMEM; }
}

// SpeciesMetrics::reset (aln_sink.h:84-91): the per-taxon counters start over, the table of observed
// perfect-hit tuples does not — so under --separator the abundance of a later input is estimated from
// the tuples of all inputs so far, exactly as the reference's reports show.
cf_status cf_report_reset_counts(cf_report *r) {
    if (!r) return CF_ERR_ARG;
    try {
        r->flush();                           // fold the dense accumulators first: their observed part stays
        r->counts.clear();
        return CF_OK;
    } catch (...) { return CF_ERR_NOMEM; }
}

cf_status cf_report_adopt_counts(cf_report *r, const uint64_t *nReads, const uint64_t *nUnique, uint64_t nTaxa) {
    if (!r || !nReads || !nUnique || nTaxa != r->h->taxa.size()) return CF_ERR_ARG;
    try {
        r->flush();
        for (uint64_t i = 0; i < nTaxa; i++) {
            const auto it = r->counts.find(r->h->taxa[i]);
            const uint64_t a = it == r->counts.end() ? 0 : it->second.nReads, b = it == r->counts.end() ? 0 : it->second.nUnique;
            if (a != nReads[i] || b != nUnique[i]) return CF_ERR_FORMAT;
        }
        for (uint64_t i = 0; i < nTaxa; i++)
            if (nReads[i] || nUnique[i]) { Counts &c = r->counts[r->h->taxa[i]]; c.nReads = nReads[i]; c.nUnique = nUnique[i]; }
        return CF_OK;
    } catch (...) { return CF_ERR_NOMEM; }
}

// the perfect multi-assignment tuples of batches formatted on the device (fmt_write_body): n, then n dense taxon indices
cf_status cf_report_add_tuples(cf_report *r, const uint32_t *tuples, uint64_t nWords) {
    if (!r || (nWords && !tuples)) return CF_ERR_ARG;
    try {
        const size_t nTaxa = r->h->taxa.size();
        std::vector<uint64_t> ids;
        for (uint64_t i = 0; i < nWords;) {
            const uint32_t n = tuples[i++];
            if (n < 2 || i + n > nWords) return CF_ERR_ARG;
            ids.clear();
            for (uint32_t j = 0; j < n; j++) { if (tuples[i + j] >= nTaxa) return CF_ERR_ARG; ids.push_back(r->h->taxa[tuples[i + j]]); }
            std::sort(ids.begin(), ids.end());
            r->observed[ids]++;
            i += n;
        }
        return CF_OK;
    } catch (...) { return CF_ERR_NOMEM; }
}

cf_status cf_report_adopt_device_tally(cf_report *r, const uint64_t *nReads, const uint64_t *nUnique, const uint64_t *nSingle, uint64_t nTaxa) {
    if (!r || !nReads || !nUnique || !nSingle || nTaxa != r->h->taxa.size()) return CF_ERR_ARG;
    try {
        r->flush();
        // the device counted every batch of the run, the host at most some of them
        for (uint64_t i = 0; i < nTaxa; i++) {
            const auto it = r->counts.find(r->h->taxa[i]);
            const uint64_t a = it == r->counts.end() ? 0 : it->second.nReads, b = it == r->counts.end() ? 0 : it->second.nUnique;
            if (a > nReads[i] || b > nUnique[i]) return CF_ERR_FORMAT;
        }
        for (uint64_t i = 0; i < nTaxa; i++) {
            if (nReads[i] || nUnique[i]) { Counts &c = r->counts[r->h->taxa[i]]; c.nReads = nReads[i]; c.nUnique = nUnique[i]; }
            if (nSingle[i]) r->observed[std::vector<uint64_t>{r->h->taxa[i]}] += nSingle[i];
        }
        return CF_OK;
    } catch (...) { return CF_ERR_NOMEM; }
}

cf_status cf_report_add_counts(cf_report *r, const uint64_t *taxa, const uint64_t *nReads, const uint64_t *nUnique, uint64_t n) {
    if (!r || !taxa || !nReads || !nUnique) return CF_ERR_ARG;
    try {
        for (uint64_t i = 0; i < n; i++) {
            if (!nReads[i] && !nUnique[i]) continue;
            Counts &c = r->counts[taxa[i]];
            c.nReads += nReads[i]; c.nUnique += nUnique[i];
        }
        return CF_OK;
    } catch (...) { return CF_ERR_NOMEM; }
}

// Flat image of a report for shipping between the per-GPU processes of a node:
// u64 nCounts, {u64 tax, u64 nReads, u64 nUnique} x nCounts, u64 nObserved, {u64 len, u64 ids[len], u64 count} x nObserved
cf_status cf_report_serialize(cf_report *r, uint64_t *buf, uint64_t capWords, uint64_t *needWords) {
    if (!r || !needWords) return CF_ERR_ARG;
    r->flush();
    uint64_t need = 2 + 3 * r->counts.size();
    for (const auto &kv : r->observed) need += 2 + kv.first.size();
    *needWords = need;
    if (!buf || capWords < need) return buf ? CF_ERR_ARG : CF_OK;
    uint64_t *w = buf;
    *w++ = r->counts.size();
    for (const auto &kv : r->counts) { *w++ = kv.first; *w++ = kv.second.nReads; *w++ = kv.second.nUnique; }
    *w++ = r->observed.size();
    for (const auto &kv : r->observed) {
        *w++ = kv.first.size();
        for (uint64_t id : kv.first) *w++ = id;
        *w++ = kv.second;
    }
    return CF_OK;
}

cf_status cf_report_merge(cf_report *r, const uint64_t *buf, uint64_t nWords) {            // SpeciesMetrics::merge aln_sink.h:109-140
    if (!r || !buf) return CF_ERR_ARG;
    try {
        const uint64_t *w = buf, *end = buf + nWords;
        if (w >= end) return CF_ERR_ARG;
        uint64_t n = *w++;
        if ((uint64_t)(end - w) < 3 * n + 1) return CF_ERR_ARG;
        for (uint64_t i = 0; i < n; i++, w += 3) { Counts &c = r->counts[w[0]]; c.nReads += w[1]; c.nUnique += w[2]; }
        n = *w++;
        for (uint64_t i = 0; i < n; i++) {
            if (w >= end) return CF_ERR_ARG;
            const uint64_t len = *w++;
            if ((uint64_t)(end - w) < len + 1) return CF_ERR_ARG;
            std::vector<uint64_t> ids(w, w + len);
            w += len;
            r->observed[ids] += *w++;
        }
        return CF_OK;
    } catch (...) { return CF_ERR_NOMEM; }
}

cf_status cf_report_write(cf_report *r, const char *path, int abundance, uint64_t *emIterations, double *emDiff) {
    if (!r || !path) return CF_ERR_ARG;
    try {
        r->flush();
        r->abundanceLen.clear();
        if (abundance) r->calculateAbundance();
        if (emIterations) *emIterations = r->emIterations;
        if (emDiff) *emDiff = r->emDiff;
        std::FILE *f = std::fopen(path, "wb");
        if (!f) return CF_ERR_IO;
        std::fputs("name\ttaxID\ttaxRank\tgenomeSize\tnumReads\tnumUniqueReads\tabundance\n", f);
        const HostIndex &h = *r->h;
        for (const auto &kv : r->counts) {
            const uint64_t taxid = kv.first;
            if (taxid == 0) continue;
            // name, or the number itself when the name table has none (centrifuge.cpp:3266-3271)
            auto ni = std::lower_bound(h.names.begin(), h.names.end(), taxid, [](const auto &a, uint64_t k) { return a.first < k; });
            if (ni != h.names.end() && ni->first == taxid) std::fputs(h.name(taxid), f);
            else std::fprintf(f, "%llu", (unsigned long long)taxid);
            std::fprintf(f, "\t%llu\t", (unsigned long long)taxid);
            const TaxNode *nd = h.findNode(taxid);
            const int rank = nd ? nd->rank : 0;
            const bool leaf = nd ? nd->leaf != 0 : false;
            std::fputs((rank == 0 && leaf) ? "leaf" : rankString(rank), f);
            uint64_t gs = 0;
            if (!r->sizeOf(taxid, gs)) gs = 0;
            // the reference's counters are uint32_t (aln_sink.h:45-51)
            std::fprintf(f, "\t%llu\t%u\t%u\t", (unsigned long long)gs, (uint32_t)kv.second.nReads, (uint32_t)kv.second.nUnique);
            auto ab = r->abundanceLen.find(taxid);
            if (ab != r->abundanceLen.end()) std::fprintf(f, "%g", ab->second);       // ostream << double = %g, precision 6
            else std::fputs("0.0", f);
            std::fputc('\n', f);
        }
        if (std::fclose(f) != 0) return CF_ERR_IO;
        return CF_OK;
    } catch (...) { return CF_ERR_NOMEM; }
}

}  // extern "C"
