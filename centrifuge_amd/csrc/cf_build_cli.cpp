// cf_build_cli.cpp — `centrifuge-build-bin`, the drop-in front end of the GPU index builder.
// Same command line as the reference's builder for what the classification path consumes
// (usage centrifuge_build.cpp:118-180, options :183-380): reference FASTA file(s) (comma
// separated) or -c sequences, --conversion-table, --taxonomy-tree, --name-table,
// --size-table, -o/--offrate, -t/--ftabchars; the options that only tune the reference's
// blockwise suffix sorter (-p, --bmax*, --dcv, --nodc, -a/--noauto, --packed, --seed) are
// accepted and ignored.  All work goes through cf_build_index (include/centrifuge_amd_build.h).
#include <cctype>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/centrifuge_amd_build.h"

extern "C" const char *cf_build_last_error(void);

namespace {
// the entry point is a library call (centrifuge_build.cpp:550-556): failures travel as exceptions up to it
[[noreturn]] void die(const std::string &m) { throw std::runtime_error(m); }
std::vector<std::string> splitComma(const std::string &s) {
    std::vector<std::string> out;
    size_t b = 0;
    while (b <= s.size()) {
        size_t e = s.find(',', b);
        if (e == std::string::npos) e = s.size();
        if (e > b) out.push_back(s.substr(b, e - b));
        b = e + 1;
    }
    return out;
}
int run(int argc, const char **argv) {
    cf_build_input in;
    cf_build_input_default(&in);
    std::string conv, tree, names, sizes;
    std::vector<std::string> pos;
    bool cmdline = false;
    int device = 0;
    for (int i = 1; i < argc; i++) {
        std::string a = argv[i], v;
        bool hasV = false;
        if (a.rfind("--", 0) == 0) { const size_t eq = a.find('='); if (eq != std::string::npos) { v = a.substr(eq + 1); a = a.substr(0, eq); hasV = true; } }
        auto val = [&]() -> std::string { if (hasV) return v; if (i + 1 >= argc) die("option " + a + " requires an argument"); return argv[++i]; };
        if (a == "--conversion-table") conv = val();
        else if (a == "--taxonomy-tree") tree = val();
        else if (a == "--name-table") names = val();
        else if (a == "--size-table") sizes = val();
        else if (a == "-o" || a == "--offrate") in.off_rate = std::atoi(val().c_str());
        else if (a == "-t" || a == "--ftabchars") in.ftab_chars = std::atoi(val().c_str());
        else if (a == "-c") cmdline = true;
        else if (a == "-f") {}
        else if (a == "-v" || a == "--verbose") in.verbose = 1;
        else if (a == "--device") device = std::atoi(val().c_str());
        else if (a == "-p" || a == "--threads" || a == "--bmax" || a == "--bmaxmultsqrt" || a == "--bmaxdivn" || a == "--dcv" ||
                 a == "--seed" || a == "-l" || a == "--linerate" || a == "--kmer-count") (void)val();
        else if (a == "-q" || a == "--quiet" || a == "-a" || a == "--noauto" || a == "--nodc" || a == "--packed" || a == "-r" ||
                 a == "--noref" || a == "-3" || a == "--justref" || a == "--ntoa" || a == "--big" || a == "--little") {}
        else if (a == "-h" || a == "--help") {
            std::puts("Usage: centrifuge-build-bin [options]* --conversion-table <table_in> --taxonomy-tree <taxonomy_in> "
                      "[--name-table <table_in2>] <reference_in> <cf_index_base>\n"
                      "  GPU builder (centrifuge_amd); writes <cf_index_base>.{1,2,3,4}.cf identical to the reference builder's");
            return 0;
        } else if (a.size() > 1 && a[0] == '-') die("centrifuge-build-bin: unrecognized option '" + a + "'");
        else pos.push_back(a);
    }
    if (pos.size() < 2) die("No input sequence or sequence file specified! / No output file specified!");
    if (conv.empty() || tree.empty()) die("Error: --conversion-table and --taxonomy-tree must be specified");
    const std::vector<std::string> refs = splitComma(pos[0]);
    std::vector<const char *> fa;
    std::vector<uint8_t> codes;
    std::vector<uint64_t> off{0};
    std::vector<std::string> seqNames;
    std::vector<const char *> seqNamePtrs;
    if (cmdline) {                                       // -c: the "files" are the sequences, named 0,1,... (centrifuge_build.cpp:411-418)
        for (size_t i = 0; i < refs.size(); i++) {
            for (char ch : refs[i]) {
                switch (ch) {
                    case 'A': case 'a': codes.push_back(0); break;
                    case 'C': case 'c': codes.push_back(1); break;
                    case 'G': case 'g': codes.push_back(2); break;
                    case 'T': case 't': codes.push_back(3); break;
                    default: if (std::isalpha((unsigned char)ch) || ch == '-') codes.push_back(4);
                }
            }
            off.push_back(codes.size());
            seqNames.push_back(std::to_string(i));
        }
        for (const auto &n : seqNames) seqNamePtrs.push_back(n.c_str());
        in.codes = codes.data(); in.seq_off = off.data(); in.seq_names = seqNamePtrs.data(); in.n_seq = seqNames.size();
    } else {
        for (const auto &r : refs) fa.push_back(r.c_str());
        in.fasta_paths = fa.data(); in.n_fasta = (int32_t)fa.size();
    }
    in.conversion_table = conv.c_str(); in.taxonomy_tree = tree.c_str();
    in.name_table = names.empty() ? nullptr : names.c_str();
    in.size_table = sizes.empty() ? nullptr : sizes.c_str();
    const cf_status st = cf_build_index(&in, pos[1].c_str(), device);
    if (st != CF_OK) die(std::string("centrifuge-build-bin: ") + cf_strerror(st) + ": " + cf_build_last_error());
    double t[4];
    cf_build_timings(t);
    if (in.verbose) std::fprintf(stderr, "Total time for call to driver() for forward index: %.1fs (GPU suffix sort + BWT %.1fs)\n", t[3], t[1]);
    return 0;
}
}  // namespace

// The reference's C entry point of the builder (centrifuge_build.cpp:550-556, declared
// centrifuge_build_main.cpp:30-32): borrows argv, returns non-zero with a message on stderr.
extern "C" int centrifuge_build(int argc, const char **argv) {
    try {
        return run(argc, argv);
    } catch (const std::exception &e) {
        std::fprintf(stderr, "%s\n", e.what());
        return 1;
    } catch (...) {
        std::fprintf(stderr, "Error: Encountered an unknown exception\n");
        return 1;
    }
}
