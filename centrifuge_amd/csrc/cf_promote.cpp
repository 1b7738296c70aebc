// cf_promote.cpp — `centrifuge-promote`: lift the taxIDs of a classification file to a rank, or merge
// a read's assignments into their lowest common ancestor.  Same command line and bytes on stdout as
// the reference's Perl script `centrifuge-promote` (cited by its line numbers); the taxonomy and the
// conversion table come straight from <index>.3.cf instead of two `centrifuge-inspect` child
// processes (centrifuge-promote:23-40).
//
//   centrifuge-promote <index> <centrifuge output> <level | lca>  > output
#include <cinttypes>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <stdexcept>
#include <string>
#include <unordered_map>
#include <unordered_set>
#include <vector>

#include "cf_index.hpp"

using namespace cfamd;

namespace {

struct Tax {
    std::unordered_map<uint64_t, uint64_t> parent;        // as `centrifuge-inspect --taxonomy-tree` lists it (the root is its own parent)
    std::unordered_map<uint64_t, std::string> level;
};

bool asId(const std::string &s, uint64_t &v) {
    if (s.empty()) return false;
    v = 0;
    for (char c : s) { if (c < '0' || c > '9') return false; v = v * 10 + (uint64_t)(c - '0'); }
    return true;
}

// PromoteTaxId (centrifuge-promote:44-58): the ancestor-or-self of tid at `level`, 0 if there is none
uint64_t promote(const Tax &tx, uint64_t tid, const std::string &level) {
    for (int guard = 0; guard < 100000; guard++) {
        auto lv = tx.level.find(tid);
        if (tid == 0 || lv == tx.level.end()) return 0;
        if (lv->second == level) return tid;
        if (tid <= 1) return 0;
        auto p = tx.parent.find(tid);
        if (p == tx.parent.end() || p->second == tid) return 0;      // (the script would recurse for ever on a non-root self-parent)
        tid = p->second;
    }
    return 0;
}

uint64_t lca(const Tax &tx, uint64_t a, uint64_t b) {                // centrifuge-promote:60-90
    if (a == 0) return b;
    if (b == 0) return a;
    if (a == b) return a;
    std::unordered_set<uint64_t> path;
    while (a != 0) {                                                 // Perl: `$a ge 1` on the decimal string
        path.insert(a);
        auto p = tx.parent.find(a);
        if (p == tx.parent.end()) { std::fprintf(stderr, "Couldn't find parent of taxID %" PRIu64 " - directly assigned to root.\n", a); break; }
        if (p->second == a) break;
        a = p->second;
    }
    while (b > 1) {
        if (path.count(b)) return b;
        auto p = tx.parent.find(b);
        if (p == tx.parent.end()) { std::fprintf(stderr, "Couldn't find parent of taxID %" PRIu64 " - directly assigned to root.\n", b); break; }
        if (p->second == b) break;
        b = p->second;
    }
    return 1;
}

// Perl's split /\t+/: fields between runs of tabs, trailing empty fields dropped (a leading one is kept)
std::vector<std::string> splitTabs(const std::string &s) {
    std::vector<std::string> f;
    size_t i = 0;
    const size_t n = s.size();
    if (n == 0) return f;
    for (;;) {
        size_t j = i;
        while (j < n && s[j] != '\t') j++;
        f.emplace_back(s, i, j - i);
        if (j >= n) break;
        while (j < n && s[j] == '\t') j++;
        if (j >= n) break;                                           // the line ended in tabs: no trailing empty field
        i = j;
    }
    while (!f.empty() && f.back().empty()) f.pop_back();
    return f;
}

std::string joinTabs(const std::vector<std::string> &f) {
    std::string o;
    for (size_t i = 0; i < f.size(); i++) { if (i) o.push_back('\t'); o += f[i]; }
    return o;
}

struct Promoter {
    const Tax &tx;
    std::string level;
    std::string out;

    void flushOut(bool force) { if (force || out.size() > (1u << 20)) { std::fwrite(out.data(), 1, out.size(), stdout); out.clear(); } }

    // OutputPromotedLines (centrifuge-promote:92-148): the rows of one read
    void group(const std::vector<std::string> &lines) {
        if (lines.empty()) return;
        std::vector<std::string> newLines;
        uint64_t numMatches = 0;
        if (level != "lca") {
            std::unordered_set<std::string> seen;
            for (const auto &ln : lines) {
                std::vector<std::string> c = splitTabs(ln);
                if (c.size() < 3) c.resize(3);
                uint64_t tid = 0;
                const bool num = asId(c[2], tid);
                const uint64_t up = num ? promote(tx, tid, level) : 0;
                std::string newTid = up <= 1 ? c[2] : std::to_string(up);
                std::string newLevel = c[1];
                uint64_t nt = 0;
                if (asId(newTid, nt) && nt >= 1) { auto lv = tx.level.find(nt); if (lv != tx.level.end()) newLevel = lv->second; }
                if (seen.count(newTid)) continue;
                seen.insert(newTid);
                numMatches++;
                c[2] = newTid; c[1] = newLevel;
                newLines.push_back(joinTabs(c));
            }
        } else {
            numMatches = 1;
            std::vector<std::string> c = splitTabs(lines[0]);
            if (c.size() < 3) c.resize(3);
            uint64_t l = 0;
            asId(c[2], l);
            for (size_t i = 1; i < lines.size(); i++) {
                std::vector<std::string> d = splitTabs(lines[i]);
                uint64_t t = 0;
                if (d.size() > 2) asId(d[2], t);
                l = lca(tx, l, t);
            }
            const std::string ls = std::to_string(l);
            if (ls != c[2]) { auto lv = tx.level.find(l); c[1] = lv == tx.level.end() ? std::string() : lv->second; }
            c[2] = ls;
            newLines.push_back(joinTabs(c));
        }
        for (const auto &nl : newLines) {                             // the script splits once more and sets the last column
            std::vector<std::string> c = splitTabs(nl);
            if (!c.empty()) c.back() = std::to_string(numMatches);
            out += joinTabs(c);
            out.push_back('\n');
        }
        flushOut(false);
    }
};

}  // namespace

int main(int argc, char **argv) {
    if (argc < 2) {
        std::fprintf(stderr, "Usage: centrifuge-promote centrifuge_index_name centrifuge_output level > output\n\n"
                             "Promote the taxonomy id to specified level in Centrifuge output.\n"
                             "\tIf level equals \"lca\", this will merge the multiassignment to their lowest common ancestor.\n");
        return 255;
    }
    if (argc < 4) { std::fprintf(stderr, "centrifuge-promote: index, classification file and level are required\n"); return 255; }
    try {
        Tax tx;
        {
            HostIndex h;
            std::string base = argv[1];
            if (std::FILE *t = std::fopen((base + ".1.cf").c_str(), "rb")) std::fclose(t);
            else if (const char *e = std::getenv("CENTRIFUGE_INDEXES")) base = std::string(e) + "/" + argv[1];
            h.load(base, nullptr);
            for (const auto &nd : h.tree) { tx.parent[nd.tid] = nd.parent; tx.level[nd.tid] = rankString(nd.rank); }
        }
        std::FILE *f = std::fopen(argv[2], "rb");
        if (!f) { std::fprintf(stderr, "centrifuge-promote: cannot open %s\n", argv[2]); return 2; }
        std::setvbuf(f, nullptr, _IOFBF, 1 << 20);
        Promoter pr{tx, argv[3], {}};
        char *buf = nullptr; size_t cap = 0; ssize_t r;
        if ((r = getline(&buf, &cap, f)) >= 0) pr.out.append(buf, (size_t)r);       // header, as it is
        std::string prev;
        bool first = true;
        std::vector<std::string> lines;
        while ((r = getline(&buf, &cap, f)) >= 0) {
            size_t n = (size_t)r;
            if (n && buf[n - 1] == '\n') n--;
            std::string ln(buf, n);
            const size_t t = ln.find('\t');
            const std::string id = ln.substr(0, t);
            if (!first && id == prev) lines.push_back(std::move(ln));
            else {
                prev = id;
                pr.group(lines);
                lines.clear();
                lines.push_back(std::move(ln));
            }
            first = false;
        }
        pr.group(lines);
        pr.flushOut(true);
        std::free(buf);
        std::fclose(f);
    } catch (const std::exception &e) {
        std::fprintf(stderr, "centrifuge-promote: %s\n", e.what());
        return 1;
    }
    return 0;
}
