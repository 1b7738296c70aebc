// cf_kreport.cpp — `centrifuge-kreport`: Kraken-style report from classification output
// (SURVEY.md §8f row 4).  Same command line and bytes on stdout as the reference's Perl script
// `centrifuge-kreport` (cited below by its line numbers), but the taxonomy comes straight from
// <index>.3.cf instead of two `centrifuge-inspect` child processes (centrifuge-kreport:233-260),
// and a 10^8-row classification file is read at file-system speed instead of Perl speed.
//
//   centrifuge-kreport -x <index> [--no-lca] [--show-zeros] [--is-count-table]
//                      [--min-score N] [--min-length N] [<centrifuge output file>...]
#include <algorithm>
#include <cerrno>
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

constexpr uint64_t kJunk = ~0ull;          // a taxID field that is not a number (e.g. the header line of a second file)

struct Opts {
    std::string index;
    bool noLca = false, showZeros = false, countTable = false, haveMinScore = false, haveMinLength = false;
    double minScore = 0, minLength = 0;
    std::vector<std::string> files;
};

[[noreturn]] void usage(int code) {                                        // centrifuge-kreport:37-60
    std::fputs("\nUsage: centrifuge-kreport -x <index name> OPTIONS <centrifuge output file(s)>\n\n"
               "centrifuge-kreport creates Kraken-style reports from centrifuge out files.\n\n"
               "Options:\n"
               "    -x INDEX            (REQUIRED) Centrifuge index\n\n"
               "    --no-lca             Do not report the LCA of multiple assignments, but report count fractions at the taxa.\n"
               "    --show-zeros         Show clades that have zero reads, too\n"
               "    --is-count-table     The format of the file is 'taxID<tab>COUNT' instead of the standard\n"
               "                         Centrifuge output format\n\n"
               "    --min-score SCORE    Require a minimum score for reads to be counted\n"
               "    --min-length LENGTH  Require a minimum alignment length to the read\n  \n  ", stderr);
    std::exit(code);
}

Opts parse(int argc, char **argv) {
    Opts o;
    for (int i = 1; i < argc; i++) {
        std::string a = argv[i], v;
        bool hasV = false;
        if (a == "--") { for (i++; i < argc; i++) o.files.push_back(argv[i]); break; }
        if (a.size() > 1 && a[0] == '-') {
            std::string n = a.substr(a[1] == '-' ? 2 : 1);                // Getopt::Long takes -opt and --opt alike
            const size_t eq = n.find('=');
            if (eq != std::string::npos) { v = n.substr(eq + 1); n = n.substr(0, eq); hasV = true; }
            auto val = [&]() -> std::string { if (hasV) return v; if (i + 1 >= argc) usage(64); return argv[++i]; };
            if (n == "x") o.index = val();
            else if (n == "no-lca") o.noLca = true;
            else if (n == "show-zeros") o.showZeros = true;
            else if (n == "is-count-table") o.countTable = true;
            else if (n == "min-score") { o.minScore = std::strtod(val().c_str(), nullptr); o.haveMinScore = true; }
            else if (n == "min-length") { o.minLength = std::strtod(val().c_str(), nullptr); o.haveMinLength = true; }
            else if (n == "help" || n == "h") usage(0);
            else { std::fprintf(stderr, "Unknown option: %s\n", n.c_str()); usage(64); }
        } else o.files.push_back(a);
    }
    if (o.index.empty()) usage(64);
    return o;
}

// concatenation of the input files ('-' or none = stdin) — Perl's <>
struct Lines {
    std::vector<std::string> files;
    size_t next = 0;
    std::FILE *f = nullptr;
    char *buf = nullptr;
    size_t cap = 0;
    explicit Lines(std::vector<std::string> fs) : files(std::move(fs)) { if (files.empty()) files.push_back("-"); }
    ~Lines() { if (f && f != stdin) std::fclose(f); std::free(buf); }
    // line without its '\n'; false at the end of the last file
    bool get(const char *&p, size_t &n) {
        for (;;) {
            if (!f) {
                if (next >= files.size()) return false;
                const std::string &path = files[next++];
                if (path == "-") f = stdin;
                else {
                    f = std::fopen(path.c_str(), "rb");
                    if (!f) { std::fprintf(stderr, "Can't open %s: %s.\n", path.c_str(), std::strerror(errno)); continue; }
                    std::setvbuf(f, nullptr, _IOFBF, 1 << 20);
                }
            }
            const ssize_t r = getline(&buf, &cap, f);
            if (r < 0) { if (f != stdin) std::fclose(f); f = nullptr; continue; }
            n = (size_t)r;
            if (n && buf[n - 1] == '\n') n--;
            p = buf;
            return true;
        }
    }
};

bool parseId(const char *p, size_t n, uint64_t &v) {
    if (n == 0) return false;
    v = 0;
    for (size_t i = 0; i < n; i++) { if (p[i] < '0' || p[i] > '9') return false; v = v * 10 + (uint64_t)(p[i] - '0'); }
    return true;
}

struct Taxonomy {
    std::vector<uint64_t> tid, parent;          // ascending tid: the order `centrifuge-inspect --taxonomy-tree` lists them
    std::vector<uint8_t> rank;
    std::unordered_map<uint64_t, uint32_t> at;  // tid -> position
    std::unordered_map<uint64_t, std::vector<uint64_t>> children;
    std::unordered_map<uint64_t, std::string> names;
    std::unordered_set<uint64_t> warned;

    bool parentOf(uint64_t a, uint64_t &p) const { auto it = at.find(a); if (it == at.end()) return false; p = parent[it->second]; return true; }
    void warnNoParent(uint64_t a) { std::fprintf(stderr, "Couldn't find parent of taxID %" PRIu64 " - directly assigned to root.\n", a); }

    bool inTree(uint64_t a) {                                              // centrifuge-kreport:161-175
        if (a == kJunk) return true;
        while (a > 1) {
            uint64_t p;
            if (!parentOf(a, p)) { warnNoParent(a); return false; }
            if (a == p) break;
            a = p;
        }
        return true;
    }
    uint64_t lca(uint64_t a, uint64_t b) {                                 // centrifuge-kreport:177-203
        if (a == 0) return b;
        if (b == 0) return a;
        if (a == b) return a;
        std::unordered_set<uint64_t> path;
        while (a != 0) {                                                   // Perl: `$a ge 1` on the decimal string
            path.insert(a);
            uint64_t p;
            if (a == kJunk || !parentOf(a, p)) { warnNoParent(a); break; }
            if (a == p) break;
            a = p;
        }
        while (b > 1 && b != kJunk) {
            if (path.count(b)) return b;
            uint64_t p;
            if (!parentOf(b, p)) { warnNoParent(b); break; }
            if (b == p) break;
            b = p;
        }
        return 1;
    }
};

const char *rankCode(const char *r) {                                      // centrifuge-kreport:205-218
    static const std::pair<const char *, const char *> k[] = {{"species", "S"}, {"genus", "G"}, {"family", "F"}, {"order", "O"},
                                                              {"class", "C"}, {"phylum", "P"}, {"kingdom", "K"}, {"superkingdom", "D"}};
    for (const auto &e : k) if (std::strcmp(r, e.first) == 0) return e.second;
    return "-";
}

struct Report {
    Taxonomy &tx;
    std::unordered_map<uint64_t, double> taxo, clade;
    double seqCount = 0;
    bool showZeros = false;
    std::string out;

    double cladeOf(uint64_t t) const { auto it = clade.find(t); return it == clade.end() ? 0.0 : it->second; }
    double taxoOf(uint64_t t) const { auto it = taxo.find(t); return it == taxo.end() ? 0.0 : it->second; }

    void sum(uint64_t node, int depth) {                                   // dfs_summation, centrifuge-kreport:220-230
        if (depth > 4096) throw std::runtime_error("taxonomy tree is deeper than 4096 levels (a cycle?)");
        auto it = tx.children.find(node);
        if (it == tx.children.end()) return;
        for (uint64_t c : it->second) {
            sum(c, depth + 1);
            clade[node] += cladeOf(c);
        }
    }
    void line(double cladeCnt, double taxoCnt, const char *code, uint64_t id, int depth, const std::string &name) {
        char b[128];
        std::snprintf(b, sizeof b, "%6.2f\t%lld\t%lld\t%s\t%" PRIu64 "\t", cladeCnt * 100 / seqCount, (long long)cladeCnt, (long long)taxoCnt, code, id);
        out += b;
        out.append((size_t)depth * 2, ' ');
        out += name;
        out.push_back('\n');
        if (out.size() > (1u << 20)) { std::fwrite(out.data(), 1, out.size(), stdout); out.clear(); }
    }
    void report(uint64_t node, int depth) {                                // dfs_report, centrifuge-kreport:138-159
        const double c = cladeOf(node);
        if (c == 0 && !showZeros) return;
        auto at = tx.at.find(node);
        auto nm = tx.names.find(node);
        line(c, taxoOf(node), rankCode(at == tx.at.end() ? "" : rankString(tx.rank[at->second])), node, depth, nm == tx.names.end() ? std::string() : nm->second);
        auto it = tx.children.find(node);
        if (it == tx.children.end()) return;
        std::vector<uint64_t> kids = it->second;
        std::stable_sort(kids.begin(), kids.end(), [&](uint64_t a, uint64_t b) { return cladeOf(a) > cladeOf(b); });   // Perl's sort is a stable merge sort
        for (uint64_t k : kids) report(k, depth + 1);
    }
};

}  // namespace

int main(int argc, char **argv) {
    const Opts o = parse(argc, argv);
    if (o.files.empty()) std::fprintf(stderr, "Reading centrifuge out file from STDIN ... \n");
    try {
        Taxonomy tx;
        std::fprintf(stderr, "Loading taxonomy ...\n");
        {
            HostIndex h;
            std::string base = o.index;                                      // adjustEbwtBase bt2_idx.cpp:38-66, as centrifuge-inspect resolves it
            if (std::FILE *t = std::fopen((base + ".1.cf").c_str(), "rb")) std::fclose(t);
            else if (const char *e = std::getenv("CENTRIFUGE_INDEXES")) base = std::string(e) + "/" + o.index;
            h.load(base, nullptr);
            std::fprintf(stderr, "Loading names file ...\n");
            for (const auto &e : h.names) tx.names[e.first] = e.second;     // a repeated taxid keeps the last name
            std::fprintf(stderr, "Loading nodes file ...\n");
            for (const auto &nd : h.tree) {                                  // centrifuge-kreport:247-259
                const uint64_t par = nd.tid == 1 ? 0 : nd.parent;
                tx.at[nd.tid] = (uint32_t)tx.tid.size();
                tx.tid.push_back(nd.tid); tx.parent.push_back(par); tx.rank.push_back(nd.rank);
                tx.children[par].push_back(nd.tid);
            }
        }
        Report rp{tx};
        rp.showZeros = o.showZeros;
        rp.taxo[0] = 0;
        Lines in(o.files);
        const char *p; size_t n;
        if (o.countTable) {                                                  // centrifuge-kreport:74-79
            while (in.get(p, n)) {
                std::string s(p, n);
                char *save = nullptr;
                char *a = strtok_r(&s[0], " \t\r\n\f\v", &save);
                char *b = a ? strtok_r(nullptr, " \t\r\n\f\v", &save) : nullptr;
                if (!a) continue;
                uint64_t id;
                if (!parseId(a, std::strlen(a), id)) id = kJunk;
                const double c = b ? std::strtod(b, nullptr) : 0.0;
                rp.taxo[id] = c;
                rp.seqCount += c;
            }
        } else {
            int cRead = 0, cTax = 0, cScore = 0, cHit = 0, cNum = 0;        // a missing column reads column 0, as `$cols[undef]` does
            if (in.get(p, n)) {
                size_t i = 0; int col = 0;
                while (i <= n) {
                    size_t j = i; while (j < n && p[j] != '\t') j++;
                    const std::string name(p + i, j - i);
                    if (name == "readID") cRead = col; else if (name == "taxID") cTax = col; else if (name == "score") cScore = col;
                    else if (name == "hitLength") cHit = col; else if (name == "numMatches") cNum = col;
                    col++; i = j + 1;
                }
            }
            std::string prevRead; bool havePrev = false; uint64_t prevTax = 0;
            std::vector<std::pair<const char *, size_t>> f;
            while (in.get(p, n)) {
                f.clear();
                for (size_t i = 0; i <= n;) { size_t j = i; while (j < n && p[j] != '\t') j++; f.emplace_back(p + i, j - i); i = j + 1; }
                auto fld = [&](int c) -> std::pair<const char *, size_t> { return (size_t)c < f.size() ? f[c] : std::pair<const char *, size_t>{"", 0}; };
                auto num = [&](int c) { const auto x = fld(c); return std::strtod(std::string(x.first, x.second).c_str(), nullptr); };
                if (o.haveMinLength && num(cHit) < o.minLength) continue;
                if (o.haveMinScore && num(cScore) < o.minScore) continue;
                const auto rd = fld(cRead), tf = fld(cTax);
                uint64_t tax;
                if (!parseId(tf.first, tf.second, tax)) tax = kJunk;
                if (!tx.inTree(tax)) tax = 1;
                if (o.noLca) {                                               // centrifuge-kreport:106-108
                    const double nm = num(cNum);
                    if (nm == 0) { std::fprintf(stderr, "Illegal division by zero (numMatches column) in the classification file.\n"); return 255; }
                    rp.taxo[tax] += 1 / nm;
                    rp.seqCount += 1 / nm;
                } else if (havePrev && prevRead.size() == rd.second && std::memcmp(prevRead.data(), rd.first, rd.second) == 0) {
                    rp.taxo[prevTax] -= 1;                                   // :110-113
                    prevTax = tx.lca(prevTax, tax);
                    rp.taxo[prevTax] += 1;
                } else {
                    rp.taxo[tax] += 1;                                       // :115-117
                    rp.seqCount += 1;
                    prevTax = tax;
                }
                prevRead.assign(rd.first, rd.second); havePrev = true;
            }
        }
        rp.clade = rp.taxo;
        rp.sum(1, 0);
        if (!(rp.seqCount > 0)) { std::fprintf(stderr, "No sequence matches with given settings at %s line 133.\n", argv[0]); return 255; }
        rp.line(rp.cladeOf(0), rp.taxoOf(0), "U", 0, 0, "unclassified");
        rp.report(1, 0);
        std::fwrite(rp.out.data(), 1, rp.out.size(), stdout);
        std::fflush(stdout);
    } catch (const std::exception &e) {
        std::fprintf(stderr, "centrifuge-kreport: %s\n", e.what());
        return 1;
    }
    return 0;
}
