// cf_inspect.cpp — `centrifuge-inspect-bin`: what is inside an index (the data
// format on the index side of the path).  Same command line and bytes on stdout as the
// reference's centrifuge_inspect.cpp:
//   -n/--names            reference names                      (centrifuge_inspect.cpp:434-443)
//   -s/--summary          flags, SA sampling, ftab, lengths    (:447-481)
//   --conversion-table    uid <tab> taxid                      (:520-531)
//   --taxonomy-tree       tid | parent | rank                  (:532-537)
//   --name-table          taxid <tab> name                     (:538-549)
//   --size-table          taxid <tab> size                     (:550-562)
//   (default)             FASTA of the indexed sequences       (:369-430, print_fasta_record :191-211)
// The table modes read the host-only view of the index; only the FASTA mode touches the GPU:
// the joined text comes back from cf_index_restore (the inverse BWT as thousands of parallel
// LF walks instead of the reference's single chain, bt2_util.h:150-168) and is cut into
// records with the fragment table exactly as joinedToTextOff does (bt2_idx.h:3891-3961).
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/centrifuge_amd.h"
#include "cf_index.hpp"
#include "cf_inspect_fasta.hpp"

using namespace cfamd;

namespace {

struct Opts {
    bool names = false, summary = false, conv = false, tree = false, nameTab = false, sizeTab = false, verbose = false, version = false;
    int across = 60, device = 0;
    std::string wrapper, base;
};

void usage(std::FILE *out, const Opts &o) {
    std::fprintf(out,
        "Centrifuge version 1.0.4-compatible (centrifuge_amd, MI355X-native)\n"
        "Usage: centrifuge-inspect [options]* <cf_base>\n"
        "  <cf_base>         cf filename minus trailing .1.cf/.2.cf/.3.cf\n"
        "\n"
        "  By default, prints FASTA records of the indexed nucleotide sequences to\n"
        "  standard out.  With -n, just prints names.  With -s, just prints a summary of\n"
        "  the index parameters and sequences.\n"
        "\n"
        "Options:\n"
        "  -a/--across <int>  Number of characters across in FASTA output (default: 60)\n"
        "  -n/--names         Print reference sequence names only\n"
        "  -s/--summary       Print summary incl. ref names, lengths, index properties\n"
        "  --conversion-table Print conversion table\n"
        "  --taxonomy-tree    Print taxonomy tree\n"
        "  --name-table       Print names corresponding to taxonomic IDs\n"
        "  --size-table       Print the lengths of the sequences belonging to the same taxonomic ID\n"
        "  --device <int>     GPU used to reconstruct the sequences (default: 0)\n"
        "  -v/--verbose       Verbose output (for debugging)\n"
        "  -h/--help          print this usage message\n");
    if (o.wrapper.empty())
        std::fprintf(stderr, "\n*** Warning ***\n'centrifuge-inspect-bin' was run directly.  It is recommended to use the wrapper script instead.\n\n");
}

[[noreturn]] void fail(const std::string &m, int code = 1) {
    std::fprintf(stderr, "%s\n", m.c_str());
    std::exit(code);
}

Opts parse(int argc, char **argv) {
    Opts o;
    std::vector<std::string> pos;
    for (int i = 1; i < argc; i++) {
        std::string a = argv[i], v;
        bool hasV = false;
        if (a.rfind("--", 0) == 0) {
            const size_t eq = a.find('=');
            if (eq != std::string::npos) { v = a.substr(eq + 1); a = a.substr(0, eq); hasV = true; }
        }
        auto val = [&]() -> std::string {
            if (hasV) return v;
            if (i + 1 >= argc) { usage(stderr, o); std::exit(1); }
            return argv[++i];
        };
        if (a == "-n" || a == "--names") o.names = true;
        else if (a == "-s" || a == "--summary") o.summary = true;
        else if (a == "--conversion-table") o.conv = true;
        else if (a == "--taxonomy-tree") o.tree = true;
        else if (a == "--name-table") o.nameTab = true;
        else if (a == "--size-table") o.sizeTab = true;
        else if (a == "-e" || a == "--ebwt-ref") {}                         // the only way this tool reconstructs anyway
        else if (a == "-v" || a == "--verbose") o.verbose = true;
        else if (a == "--version") o.version = true;
        else if (a == "--wrapper") o.wrapper = val();
        else if (a == "--device") o.device = std::atoi(val().c_str());
        else if (a == "-a" || a == "--across") {
            const std::string s = val();
            char *end = nullptr;
            const long l = std::strtol(s.c_str(), &end, 10);
            if (l < -1) { std::fprintf(stderr, "-a/--across arg must be at least 1\n"); usage(stderr, o); std::exit(1); }
            o.across = (int)l;
        }
        else if (a == "-h" || a == "--help" || a == "--usage") { usage(stdout, o); std::exit(0); }
        else if (a.size() > 1 && a[0] == '-') { usage(stderr, o); std::exit(1); }
        else pos.push_back(a);
    }
    if (o.version) { std::printf("%s version 1.0.4-compatible (centrifuge_amd)\n64-bit\n", argv[0]); std::exit(0); }
    if (pos.empty()) { std::fprintf(stderr, "No index name given!\n"); usage(stderr, o); std::exit(1); }
    o.base = pos[0];
    return o;
}

std::string findIndex(const std::string &base) {                            // adjustEbwtBase bt2_idx.cpp:38-66
    auto exists = [](const std::string &p) { std::FILE *f = std::fopen((p + ".1.cf").c_str(), "rb"); if (f) std::fclose(f); return f != nullptr; };
    if (exists(base)) return base;
    if (const char *e = std::getenv("CENTRIFUGE_INDEXES")) { const std::string p = std::string(e) + "/" + base; if (exists(p)) return p; }
    fail("Could not locate a Centrifuge index corresponding to basename \"" + base + "\"");
}

void appendTaxId(std::string &s, uint64_t tid) {                            // lo32[.hi32], centrifuge_inspect.cpp:524-529
    s += std::to_string(tid & 0xffffffffull);
    if (tid >> 32) { s.push_back('.'); s += std::to_string(tid >> 32); }
}

}  // namespace

int main(int argc, char **argv) {
    const Opts o = parse(argc, argv);
    try {
        const std::string base = findIndex(o.base);
        if (o.verbose) std::printf("Input ebwt file: \"%s\"\nOutput file: \"\"\nLocal endianness: little\nAssertions: disabled\n", o.base.c_str());
        HostIndex h;
        h.load(base, nullptr);
        Out out{stdout};
        std::string &s = out.buf;
        if (o.names) {
            for (const auto &n : h.refnames) { s += n; s.push_back('\n'); out.flushIf(); }
        } else if (o.summary) {
            s += "Flags\t" + std::to_string(-(int64_t)h.flags) + "\n";
            s += "SA-Sample\t1 in " + std::to_string(1ull << h.g.offRate) + "\n";
            s += "FTab-Chars\t" + std::to_string(h.g.ftabChars) + "\n";
            for (size_t i = 0; i < h.refnames.size(); i++) {
                s += "Sequence-" + std::to_string(i + 1) + "\t" + h.refnames[i] + "\t" + std::to_string(i < h.plen.size() ? h.plen[i] : 0) + "\n";
                out.flushIf();
            }
        } else if (o.conv) {
            for (size_t i = 0; i < h.uid.size(); i++) { s += h.uid[i]; s.push_back('\t'); appendTaxId(s, h.uidTid[i]); s.push_back('\n'); out.flushIf(); }
        } else if (o.tree) {
            for (const auto &n : h.tree) {
                s += std::to_string(n.tid) + "\t|\t" + std::to_string(n.parent) + "\t|\t" + rankString(n.rank) + "\n";
                out.flushIf();
            }
        } else if (o.nameTab) {
            for (size_t i = 0; i < h.names.size(); i++) {
                if (i + 1 < h.names.size() && h.names[i + 1].first == h.names[i].first) continue;       // map semantics: last one wins
                appendTaxId(s, h.names[i].first); s.push_back('\t'); s += h.names[i].second; s.push_back('\n');
                out.flushIf();
            }
        } else if (o.sizeTab) {
            for (const auto &e : h.sizes) { appendTaxId(s, e.first); s.push_back('\t'); s += std::to_string(e.second); s.push_back('\n'); out.flushIf(); }
        } else {
            cf_index *ix = nullptr;
            if (cf_index_open(base.c_str(), o.device, &ix) != CF_OK) fail(std::string("centrifuge-inspect: ") + cf_last_error());
            std::vector<uint8_t> packed(h.g.len / 4 + 1);
            if (cf_index_restore(ix, packed.data(), packed.size()) != CF_OK) fail(std::string("centrifuge-inspect: ") + cf_last_error());
            cf_index_close(ix);
            out.flushIf(1);
            printSequences(h, packed.data(), o.across, stdout);
        }
    } catch (const std::exception &e) {
        std::fprintf(stderr, "Error: Encountered exception: '%s'\nCommand: ", e.what());
        for (int i = 0; i < argc; i++) std::fprintf(stderr, "%s ", argv[i]);
        std::fprintf(stderr, "\n");
        return 1;
    }
    return 0;
}
