// cf_cli.cpp — `centrifuge-class`, the drop-in front end of the classification path.
//
// Keeps the reference's command line for this path (usage centrifuge.cpp:737-868,
// option table :530-695, positional forms :3385-3428), the TSV of
// AlnSinkSam::appendMate (aln_sink.h:2279-2337; header centrifuge.cpp:2985-2992) and
// the report file (centrifuge.cpp:3231-3319).  All classification work goes through
// the C ABI of libcentrifuge_amd.so (include/centrifuge_amd.h); this file is host
// plumbing: option parsing, read ingest, batching, formatting.
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <cerrno>
#include <atomic>

#include <algorithm>
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <future>
#include <deque>
#include <map>
#include <memory>
#include <mutex>
#include <stdexcept>
#include <string>
#include <thread>
#include <type_traits>
#include <vector>

#include "../../include/centrifuge_amd.h"
#include "cf_ingest.hpp"
#include "cf_reads.hpp"
#include "cf_knobs.hpp"
#include "cf_bytesource.hpp"

using namespace cfamd;

namespace {

enum Col { C_READ_ID, C_SEQ_ID, C_TAX_ID, C_TAX_RANK, C_TAX_NAME, C_SCORE, C_SCORE2, C_HIT_LEN, C_QUERY_LEN, C_NUM_MATCHES,
           C_SEQ, C_QUAL, C_SEQ1, C_QUAL1, C_SEQ2, C_QUAL2, C_PLACEHOLDER, C_ZERO };

struct Opts {
    std::string index, outFile, reportFile = "centrifuge_report.tsv";
    std::vector<std::string> queries, mates1, mates2;
    ReadFormat format = ReadFormat::Fastq;              // centrifuge.cpp:300 (FASTQ is the default)
    int khits = 5, minHitLen = 22, threads = 1, trim5 = 0, trim3 = 0, device = 0;
    int gpus = 1;                                       // --gpus N | all: devices device .. device+N-1, the index replicated on each
    int slots = 2;                                      // --slots: GPU threads (batch slots, each with its stream) per device
    bool slotsSet = false;
    bool hostIo = false;                                // --host-io: the parser pool and the formatter threads for every input (no device text path)
    int smallRangeRows = 0;                             // --small-range-rows: cf_index_options::small_range_rows (0 = automatic, -1 = off)
    double hbmBudgetGb = 0;                             // --hbm-budget-gb: cf_index_options::hbm_budget_bytes (0 = what the device has free)
    long long expectedReads = -1;                       // --expected-reads: cf_index_options::expected_reads (-1 = estimated from the input files' sizes, 0 = unknown: the tables that make a read cheapest)
    std::vector<int> gpuList;                           // --gpu-list a,b,..: the devices by number (a number may repeat: logical workers on one GPU)
    uint64_t skip = 0, upto = ~0ull, batch = 1u << 20;
    uint32_t seed = 0;
    bool traverse = true, abundance = true, timing = false, quiet = false, dumpReads = false, ingestBench = false, samFormat = false, separator = false;
    std::string rank = "strain";
    std::vector<uint64_t> hostTaxids, excludeTaxids;
    std::vector<std::string> colNames = {"readID", "seqID", "taxID", "score", "2ndBestScore", "hitLength", "queryLength", "numMatches"};
    std::vector<int> cols;
};

// Leaves centrifuge() with a return code (and a message for stderr): the entry point is a library
// call (centrifuge.cpp:3338-3345), so nothing below may exit() or let an exception escape.
struct CliExit : std::runtime_error {
    int code;
    CliExit(const std::string &m, int rc) : std::runtime_error(m), code(rc) {}
};
[[noreturn]] void die(const std::string &m, int rc = 1) { throw CliExit(m, rc); }

void usage(std::FILE *f) {
    std::fputs(
        "Centrifuge-compatible classifier, MI355X-native hot path (centrifuge_amd)\n"
        "Usage:\n"
        "  centrifuge-class [options]* -x <cf-idx> {-1 <m1> -2 <m2> | -U <r>} [-S <filename>] [--report-file <report>]\n\n"
        "  <cf-idx>   Index filename prefix (minus trailing .X.cf)\n"
        "  <m1>/<m2>  Files with #1 / #2 mates (comma-separated lists)\n"
        "  <r>        Files with unpaired reads (comma-separated list; '-' = stdin; .gz/.bz2 are piped)\n"
        " Input:   -q (FASTQ, default)  -f (FASTA)  -r (one sequence per line)  -c (sequences on the command line)\n"
        "          -s/--skip <int>  -u/--upto <int>  -5/--trim5 <int>  -3/--trim3 <int>\n"
        " Classification:  -k <int> (5)  --min-hitlen <int> (22)  --host-taxids <t,..>  --exclude-taxids <t,..>\n"
        "          --classification-rank <strain|species|genus|family|order|class|phylum>  --no-traverse\n"
        " Output:  -S <file>  --report-file <file> (centrifuge_report.tsv)  --no-abundance  --tab-fmt-cols <c,..>  --out-fmt tab|sam  -t/--time\n"
        " Other:   -p/--threads <int> (host formatting threads)  --seed <int>  --batch <int>  --reorder --mm (accepted)\n"
        " GPUs:    --gpus <N|all> (index replicated on N devices from --device <int> on, batches dealt to them, per-taxon counters\n"
        "          all-reduced with RCCL, output in input order)  --gpu-list <d,..>  --slots <int> (batches in flight per device, 2)\n"
        "          --host-io (reads are parsed and rows printed by host threads for every input; default: plain FASTA / FASTQ files go up as\n"
        "          text and the default columns come back as text, formatted on the device — same bytes either way)\n"
        " Index:   --hbm-budget-gb <float> (device memory the index may take, files + derived tables; default: what is free less a\n"
        "          reserve for the batch slots)  --small-range-rows <-1|0|2..15> (search ranges of up to that many rows are finished\n"
        "          against the text; 0 = decided from how repeat-rich the indexed collection is, -1 = off; results do not depend on it)\n"
        "          --expected-reads <int> (size of the job: the derived tables of the index take seconds to make and are made only\n"
        "          as far as that many reads repay them; default: estimated from the input files' sizes; 0 = every table that fits)\n",
        f);
}

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

int colOf(const std::string &n) {
    static const std::pair<const char *, int> kMap[] = {
        {"readID", C_READ_ID}, {"seqID", C_SEQ_ID}, {"taxLevel", C_TAX_RANK}, {"taxRank", C_TAX_RANK}, {"taxID", C_TAX_ID},
        {"taxName", C_TAX_NAME}, {"score", C_SCORE}, {"2ndBestScore", C_SCORE2}, {"hitLength", C_HIT_LEN},
        {"queryLength", C_QUERY_LEN}, {"numMatches", C_NUM_MATCHES}, {"readSeq", C_SEQ}, {"readQual", C_QUAL},
        {"readSeq1", C_SEQ1}, {"readQual1", C_QUAL1}, {"readSeq2", C_SEQ2}, {"readQual2", C_QUAL2},
        {"SEQ1", C_SEQ1}, {"QUAL1", C_QUAL1}, {"SEQ2", C_SEQ2}, {"QUAL2", C_QUAL2},
        // SAM-style names (centrifuge.cpp:497-508)
        {"QNAME", C_READ_ID}, {"FLAG", C_ZERO}, {"RNAME", C_TAX_ID}, {"POS", C_ZERO}, {"MAPQ", C_ZERO}, {"CIGAR", C_PLACEHOLDER},
        {"RNEXT", C_SEQ_ID}, {"PNEXT", C_ZERO}, {"TLEN", C_QUERY_LEN}, {"SEQ", C_SEQ}, {"QUAL", C_QUAL}};
    for (const auto &kv : kMap) if (n == kv.first) return kv.second;
    die("Column definition " + n + " invalid.");
}

int rankSlot(const std::string &r) {
    static const char *const kRanks[] = {"strain", "species", "genus", "family", "order", "class", "phylum"};
    for (int i = 0; i < 7; i++) if (r == kRanks[i]) return i;
    die("Error: " + r + " (--classification-rank) should be one of strain, species, genus, family, order, class, and phylum");
}

Opts parse(int argc, const char **argv) {
    Opts o;
    std::vector<std::string> pos;
    auto need = [&](int &i, const std::string &name) -> std::string {
        if (i + 1 >= argc) die("option " + name + " requires an argument");
        return argv[++i];
    };
    for (int i = 1; i < argc; i++) {
        std::string a = argv[i], v;
        bool hasV = false;
        if (a.rfind("--", 0) == 0) {
            const size_t eq = a.find('=');
            if (eq != std::string::npos) { v = a.substr(eq + 1); a = a.substr(0, eq); hasV = true; }
        }
        auto val = [&]() { return hasV ? v : need(i, a); };
        if (a == "-x" || a == "--index") o.index = val();
        else if (a == "-U") { for (auto &s : splitComma(val())) o.queries.push_back(s); }
        else if (a == "-1") { for (auto &s : splitComma(val())) o.mates1.push_back(s); }
        else if (a == "-2") { for (auto &s : splitComma(val())) o.mates2.push_back(s); }
        else if (a == "-S" || a == "--output") o.outFile = val();
        else if (a == "--report-file") o.reportFile = val();
        else if (a == "-q") o.format = ReadFormat::Fastq;
        else if (a == "-f") o.format = ReadFormat::Fasta;
        else if (a == "-r") o.format = ReadFormat::Raw;
        else if (a == "-c") o.format = ReadFormat::CmdLine;
        else if (a == "-k") o.khits = std::atoi(val().c_str());
        else if (a == "--min-hitlen") o.minHitLen = std::atoi(val().c_str());
        else if (a == "-p" || a == "--threads") { o.threads = std::atoi(val().c_str()); if (o.threads < 1) die("-p/--threads arg must be at least 1"); }
        else if (a == "-s" || a == "--skip") o.skip = std::strtoull(val().c_str(), nullptr, 10);
        else if (a == "-u" || a == "--upto" || a == "--qupto") { o.upto = std::strtoull(val().c_str(), nullptr, 10); if (o.upto < 1) die("-u/--qupto arg must be at least 1"); }
        else if (a == "-5" || a == "--trim5") o.trim5 = std::atoi(val().c_str());
        else if (a == "-3" || a == "--trim3") o.trim3 = std::atoi(val().c_str());
        else if (a == "--seed") o.seed = (uint32_t)std::strtoul(val().c_str(), nullptr, 10);
        else if (a == "--host-taxids") { for (auto &s : splitComma(val())) o.hostTaxids.push_back(std::strtoull(s.c_str(), nullptr, 10)); }
        else if (a == "--exclude-taxids") { for (auto &s : splitComma(val())) o.excludeTaxids.push_back(std::strtoull(s.c_str(), nullptr, 10)); }
        else if (a == "--classification-rank") o.rank = val();
        else if (a == "--no-traverse") o.traverse = false;
        else if (a == "--no-abundance") o.abundance = false;
        else if (a == "--tab-fmt-cols") o.colNames = splitComma(val());
        else if (a == "--out-fmt") {                                          // centrifuge.cpp:1457-1469
            const std::string f = val();
            if (f == "sam") { o.samFormat = true; o.colNames = splitComma("QNAME,FLAG,RNAME,POS,MAPQ,CIGAR,RNEXT,PNEXT,TLEN,SEQ,QUAL"); }
            else if (f != "default" && f != "tab") die("Invalid output format " + f + "!");
        }
        else if (a == "-t" || a == "--time") o.timing = true;
        else if (a == "--separator") o.separator = true;
        else if (a == "--quiet") o.quiet = true;
        else if (a == "--dump-reads") o.dumpReads = true;          // ingest only: name, bases, qualities, seed per read (tests)
        else if (a == "--ingest-bench") o.dumpReads = o.ingestBench = true;   // ingest only, nothing printed but the rate (tools/ingest_rate.py)
        else if (a == "--device") o.device = std::atoi(val().c_str());
        else if (a == "--gpus") { const std::string g = val(); o.gpus = g == "all" ? -1 : std::atoi(g.c_str()); if (o.gpus == 0 || o.gpus < -1) die("--gpus arg must be a positive number or 'all'"); }
        else if (a == "--gpu-list") { for (auto &x : splitComma(val())) o.gpuList.push_back(std::atoi(x.c_str())); }
        else if (a == "--slots") { o.slots = std::atoi(val().c_str()); o.slotsSet = true; if (o.slots < 1) die("--slots arg must be at least 1"); }
        else if (a == "--host-io") o.hostIo = true;
        else if (a == "--small-range-rows") { o.smallRangeRows = std::atoi(val().c_str()); if (o.smallRangeRows < -1 || o.smallRangeRows == 1 || o.smallRangeRows > 15) die("--small-range-rows arg must be -1 (off), 0 (automatic) or 2 .. 15"); }
        else if (a == "--expected-reads") { o.expectedReads = std::atoll(val().c_str()); if (o.expectedReads < 0) die("--expected-reads arg must not be negative"); }
        else if (a == "--hbm-budget-gb") { o.hbmBudgetGb = std::atof(val().c_str()); if (o.hbmBudgetGb < 0) die("--hbm-budget-gb arg must not be negative"); }
        else if (a == "--batch") o.batch = std::max<uint64_t>(1, std::strtoull(val().c_str(), nullptr, 10));
        else if (a == "--reorder" || a == "--mm" || a == "--non-deterministic" || a == "--qc-filter" || a == "--phred33" ||
                 a == "--ignore-quals" || a == "--nofw" || a == "--norc" || a == "--no-1mm-upfront") {}      // accepted, no effect on this path
        else if (a == "--min-totallen" || a == "--met-file" || a == "--met" || a == "--un" || a == "--al") (void)val();
        else if (a == "-h" || a == "--help") { usage(stdout); throw CliExit("", 0); }
        else if (a == "--version") { std::puts("centrifuge-class (centrifuge_amd, MI355X-native path; Centrifuge 1.0.4 compatible)"); throw CliExit("", 0); }
        else if (a.size() > 1 && a[0] == '-' && a != "-") die("centrifuge-class: unrecognized option '" + a + "'");
        else pos.push_back(a);
    }
    // positional forms (centrifuge.cpp:3385-3428)
    size_t pi = 0;
    if (o.index.empty() && !o.dumpReads) {
        if (pi >= pos.size()) { usage(stderr); die("No index, query, or output file specified!"); }
        o.index = pos[pi++];
    }
    const bool got = !o.queries.empty() || !o.mates1.empty();
    if (pi >= pos.size()) { if (!got) { usage(stderr); die("***\nError: Must specify at least one read input with -U/-1/-2"); } }
    else if (!got) o.queries = splitComma(pos[pi++]);
    if (pi < pos.size() && o.outFile.empty()) {
        o.outFile = pos[pi++];
        std::fprintf(stderr, "Warning: Output file '%s' was specified without -S.  This will not work in future Centrifuge versions.  Please use -S instead.\n", o.outFile.c_str());
    }
    if (pi < pos.size()) die("Extra parameter(s) specified: " + pos[pi]);
    if (o.mates1.size() != o.mates2.size()) die("Error: " + std::to_string(o.mates1.size()) + " mate files/sequences were specified with -1, but " +
                                                std::to_string(o.mates2.size()) + " mate files/sequences were specified with -2.  The same number of mate files/sequences must be specified with -1 and -2.");
    if (o.minHitLen < 15) die("--min-hitlen arg must be at least 15");         // centrifuge.cpp:1402
    if (o.khits < 1) die("-k arg must be at least 1");
    (void)rankSlot(o.rank);                                                  // an unknown rank is an option error (centrifuge.cpp:1432-1445)
    if (o.upto + o.skip > o.upto) o.upto += o.skip;                          // -u counts after -s (centrifuge.cpp:1628-1633)
    for (const auto &c : o.colNames) o.cols.push_back(colOf(c));
    return o;
}

std::string findIndex(const std::string &base) {                            // adjustEbwtBase bt2_idx.cpp:38-66
    auto exists = [](const std::string &p) { std::FILE *f = std::fopen((p + ".1.cf").c_str(), "rb"); if (f) std::fclose(f); return f != nullptr; };
    if (exists(base)) return base;
    if (const char *e = std::getenv("CENTRIFUGE_INDEXES")) { const std::string p = std::string(e) + "/" + base; if (exists(p)) return p; }
    die("Could not locate a Centrifuge index corresponding to basename \"" + base + "\"");
}

#define CF_TRY(expr)                                                                                        \
    do {                                                                                                    \
        cf_status s_ = (expr);                                                                              \
        if (s_ != CF_OK) die(std::string("centrifuge-class: ") + cf_strerror(s_) + ": " + cf_last_error()); \
    } while (0)

// one batch on its way through parse -> classify -> format: reads in structure-of-arrays form,
// mates of a pair adjacent
struct Batch {
    ReadSoA r;
    // filled by the GPU stage
    std::vector<cf_row> rows;                         // packed, query order
    // ... or, when the results crossed the link in their narrow form and the default columns are printed, the narrow rows as they
    // came (16 bytes each) and the queries' bytes: the formatter reads them as they are (round 6; round 5 widened every row on the
    // GPU thread first: 0.57 of a 50 M-read run's 0.87 s)
    std::vector<cf_row16> rows16;
    std::vector<uint8_t> qinfo;
    bool narrowRows = false;
    std::vector<uint64_t> rowFirst;                   // rows[rowFirst[q] .. +nRows[q]) belong to query q
    std::vector<uint32_t> nRows, score2, maxScore;
    uint64_t nq = 0;
    bool paired = false;                              // mates of a pair adjacent in r
    int endOfInput = -1;                              // >= 0: no reads, marks the end of input number `endOfInput` (--separator)
    uint64_t seq = 0;                                 // position in the input: batches are printed in this order
    // a block of a plain FASTA / FASTQ file that goes up as TEXT (the device text path: classifyText): the range of the file, the
    // block's number within its input
    bool isText = false, tFirst = false, tLast = false;
    int tFd = -1;
    uint64_t tOff = 0, tLen = 0, tIdx = 0;
    const std::string *tPath = nullptr;
    int tFd2 = -1;                                    // mates: the second file's range that holds the same records (tLen2 = 0 with tFd2 < 0: unpaired)
    uint64_t tOff2 = 0, tLen2 = 0;
    const std::string *tPath2 = nullptr;
    ReadSoA r2;                                       // ... and its reads when the host parser takes the block
};

// Numbers handed in by position (the blocks of an input in file order), each caller learning the sum of all earlier positions:
// the ordinal of a block's first read (as soon as the blocks before it are parsed), the place of its text in the output (as soon
// as the blocks before it are formatted).  enter() waits for the caller's turn, leave() ends it.
struct OrderedSum {
    std::mutex mu;
    std::condition_variable cv;
    uint64_t next = 0, sum = 0;
    bool failed = false;
    void reset() { std::lock_guard<std::mutex> lk(mu); next = 0; sum = 0; }
    uint64_t enter(uint64_t idx) {
        std::unique_lock<std::mutex> lk(mu);
        cv.wait(lk, [&] { return failed || next == idx; });
        if (failed) throw std::runtime_error("another thread of the run failed");
        return sum;
    }
    void leave(uint64_t v) { { std::lock_guard<std::mutex> lk(mu); sum += v; next++; } cv.notify_all(); }
    void fail() { { std::lock_guard<std::mutex> lk(mu); failed = true; } cv.notify_all(); }
};

void appendReadId(std::string &o, const char *name, size_t n) {             // aln_sink.h:2203-2217
    if (n >= 2 && name[n - 2] == '/' && (name[n - 1] == '1' || name[n - 1] == '2' || name[n - 1] == '3')) n -= 2;
    for (size_t i = 0; i < n; i++) { if (std::isspace((unsigned char)name[i])) break; o.push_back(name[i]); }
}
inline void appendNum(std::string &o, uint64_t v) {                        // decimal digits without a temporary string
    char b[24];
    int n = 0;
    do { b[n++] = (char)('0' + v % 10); v /= 10; } while (v);
    while (n) o.push_back(b[--n]);
}
void appendTaxId(std::string &o, uint64_t t) {                             // aln_sink.h:2236-2250
    appendNum(o, t & 0xffffffffull);
    if (t >> 32) { o.push_back('.'); appendNum(o, t >> 32); }
}
void appendSeq(std::string &o, const ReadSoA &r, size_t i) {
    for (uint64_t k = r.off[i]; k < r.off[i + 1]; k++) o.push_back("ACGTN"[r.seq[k] > 4 ? 4 : r.seq[k]]);
}
void appendQual(std::string &o, const ReadSoA &r, size_t i) {
    if (!r.hasQual) o.append(r.off[i + 1] - r.off[i], 'I');
    else o.append(reinterpret_cast<const char *>(r.qual.data()) + r.off[i], r.off[i + 1] - r.off[i]);
}

// ---- the default eight columns, fast: raw writes into a buffer sized up front, decimal digits two at a time, and the
//      two strings a row repeats — seqID and taxID of its (unique_id, taxon) — made once per taxon / reference
inline char *putNum(char *w, uint64_t v) {
    static const char kPairs[] =
        "00010203040506070809101112131415161718192021222324252627282930313233343536373839404142434445464748495051525354555657585960616263646566676869"
        "707172737475767778798081828384858687888990919293949596979899";
    char b[24];
    int n = 0;
    while (v >= 100) { const unsigned d = (unsigned)(v % 100); v /= 100; b[n++] = kPairs[2 * d + 1]; b[n++] = kPairs[2 * d]; }
    if (v >= 10) { b[n++] = kPairs[2 * v + 1]; b[n++] = kPairs[2 * v]; } else b[n++] = (char)('0' + v);
    while (n) *w++ = b[--n];
    return w;
}
struct FormatTables {                                   // built once per run from the host view of the index
    std::vector<std::string> taxCols;                   // per dense taxon index: "<taxID as lo32[.hi32]>"
    std::vector<std::string> rankName;                  // per dense taxon index: the seqID of a merged assignment / a non-leaf taxon
    std::vector<uint8_t> taxLeaf;                       // per dense taxon index: the uid is printed (leaf, or taxID absent from the tree)
    std::vector<std::pair<const char *, uint32_t>> uid; // per reference: uid text
    size_t maxSeqId = 16, maxTax = 24;
};

struct StageTimes { double create = 0, classify = 0, results = 0, report = 0, format = 0, write = 0, produce = 0, wait = 0, read = 0, parse = 0, hostParse = 0; };

// One entry of the device list: its own replica of the index in that device's HBM and its classifier (whose
// per-taxon counters live on the device).  A device number may appear twice: two logical GPUs on one.
struct Device {
    int id = 0;
    cf_index *ix = nullptr;
    cf_classifier *clf = nullptr;
    double openS = 0;                               // seconds this device took to open its replica
    int numaNode = -1;                              // NUMA node of its PCIe link (-1: unknown): its loader and GPU threads run there
};
// One GPU thread: a batch slot and a stream on its device, and the thread's own tally of what it classified
// (SpeciesMetrics per thread, merged at the end: aln_sink.h:109-140).
struct GpuThread {
    Device *dev = nullptr;
    cf_batch *slot = nullptr;
    void *stream = nullptr;
    cf_report *rep = nullptr;
    StageTimes tm;
    char *tin = nullptr;                            // the device text path: the block as the file holds it, in pinned memory
    size_t tinCap = 0;
    uint64_t textBlocks = 0, hostBlocks = 0;        // blocks that went up as text / were parsed on the host (not in the plain form)
};

struct Runner {
    const Opts &o;
    StageTimes tm;
    double indexOpenS = 0;                          // wall time of all replicas' cf_index_open (they run at once)
    double beforeOpenS = 0;                         // ... and what the call took before it got there (the HIP runtime's start)
    std::vector<Device> devs;
    std::vector<GpuThread> gts;
    cf_index *ix = nullptr;                         // devs[0].ix: the host-side tables every formatter reads
    cf_report *rep = nullptr;                       // --separator: the one report, fed in output order
    // the formatter threads' own tallies (round 6): a batch whose rows came narrow is tallied range by range by the threads that
    // format it — off the GPU threads, where it was a quarter of their time — and merged with the others at the end
    std::vector<cf_report *> fmtReps;
    std::FILE *out = stdout;
    struct OutBuf;
    // the device text path: blocks of a plain FASTA / FASTQ file go up as text, the default columns come back as text
    bool textCapable = false;                       // the run's options allow it (the inputs decide one by one)
    OrderedSum readChain, outChain;                 // read ordinals and output places of an input's blocks, in file order
    int outFd = -1;                                 // the output's descriptor while an input goes through the text path
    bool outRegular = false;                        // ... a regular file: every GPU thread writes its block at its own place (pwrite)
    uint64_t outBase = 0;                           // ... where this input's text starts in it
    // CF_CLI_MAP_OUTPUT=1 (measured a loss, off): the blocks COPIED into the file through mappings of their own places.  Buffered
    // writes to one file queue up behind the file's lock whatever the number of writers — 2 GB of text take 0.47 s from four
    // threads, from eight, from sixteen: 4.3 GB/s, the run's ceiling at 1.06e8 reads/s —, but the page faults of a mapping queue up
    // behind the process's address-space lock and cost more each: 1.0 s from four threads, search wall 0.72 against 0.54 s
    // (profiles/r06n_cli_sweeps.txt)
    std::atomic<bool> outMap{false};
    std::mutex growMu;
    uint64_t outSize = 0;                           // the file's size as grown so far (ahead of the text; cut back at the input's end)
    std::atomic<bool> uptoReached{false};           // -u: the blocks so far hold that many reads
    std::atomic<uint64_t> textBatches{0};           // batches whose rows were formatted (and tallied) on the device
    std::vector<OutBuf *> hostOut;                  // per GPU thread: the text of a block that was parsed and formatted on the host

    ~Runner() {                                     // error paths leave through here as well
        try { waitWrite(); } catch (...) {}          // (a run that ends on an error: the writer must be done with the file before it is closed)
        if (out && out != stdout) std::fclose(out);
        if (rep) cf_report_destroy(rep);
        for (auto &g : gts) { if (g.slot) cf_batch_destroy(g.slot); if (g.rep) cf_report_destroy(g.rep); if (g.stream) cf_stream_destroy(g.stream); if (g.tin) cf_host_free(g.tin); }
        for (cf_report *r : fmtReps) if (r) cf_report_destroy(r);
        for (OutBuf *b : hostOut) delete b;
        for (auto &d : devs) { if (d.clf) cf_classifier_destroy(d.clf); if (d.ix) cf_index_close(d.ix); }
    }

    FormatTables ft;
    // a formatter thread's output: plain memory that keeps its capacity from batch to batch and is never zero-filled
    // (a std::string grown to the worst case of a range writes three times the bytes the rows take)
    struct OutBuf {
        std::unique_ptr<char[]> p;
        size_t cap = 0, len = 0;
        char *room(size_t n) { if (n > cap) { p.reset(new char[n + 64]); cap = n; } len = 0; return p.get(); }
    };
    bool defaultCols = false;
    // one buffer per formatter thread, kept from batch to batch — in TWO sets: while the writer below puts one batch's text into the
    // file, the formatter threads fill the other set with the next batch's (round 5: the write was a third of the output stage)
    std::vector<OutBuf> fmtSets[2];
    int fmtCur = 0;
    std::future<void> writeFut;                       // the write in flight (at most one: the file is written in order)
    double writeBusy = 0;                             // seconds the writer spent in fwrite (the output thread only waits for it)
    void waitWrite() { if (writeFut.valid()) writeFut.get(); }      // (rethrows what the writer threw)

    void makeFormatTables() {
        static const int kDefault[] = {C_READ_ID, C_SEQ_ID, C_TAX_ID, C_SCORE, C_SCORE2, C_HIT_LEN, C_QUERY_LEN, C_NUM_MATCHES};
        defaultCols = o.cols.size() == 8 && std::equal(o.cols.begin(), o.cols.end(), kDefault);
        if (!defaultCols) return;
        const uint64_t nTaxa = cf_index_num_taxa(ix), nRefs = cf_index_num_refs(ix);
        ft.taxCols.resize(nTaxa); ft.rankName.resize(nTaxa); ft.taxLeaf.resize(nTaxa);
        for (uint64_t i = 0; i < nTaxa; i++) {
            const uint64_t t = cf_index_taxon_id(ix, i);
            appendTaxId(ft.taxCols[i], t);
            // cf_format_seqid's rule (classifier.h:546-557 + aln_sink.h:2219-2234): the uid for a leaf (or a taxID the tree
            // does not know) when the row still names a reference, else the rank string
            const char *viaMerged = cf_format_seqid(ix, CF_MERGED, t);
            ft.rankName[i] = viaMerged;
            ft.taxLeaf[i] = nRefs && cf_format_seqid(ix, 0, t) != viaMerged ? 1 : 0;
            ft.maxSeqId = std::max(ft.maxSeqId, ft.rankName[i].size());
            ft.maxTax = std::max(ft.maxTax, ft.taxCols[i].size());
        }
        ft.uid.resize(nRefs);
        for (uint64_t r = 0; r < nRefs; r++) {
            const char *u = cf_index_uid(ix, r);
            ft.uid[r] = {u, (uint32_t)std::strlen(u)};
            ft.maxSeqId = std::max<size_t>(ft.maxSeqId, ft.uid[r].second);
        }
    }

    // the default columns: readID seqID taxID score 2ndBestScore hitLength queryLength numMatches (centrifuge.cpp:520)
    template <typename Row>
    void formatDefault(const Batch &b, const std::vector<Row> &rows, const std::vector<uint32_t> &nRows,
                       const std::vector<uint32_t> &score2, uint64_t q0, uint64_t q1, OutBuf &ob) const {
        constexpr bool kWide = std::is_same<Row, cf_row>::value;      // (a narrow row's taxon always lies in the dense table: cf_results_narrow)
        const bool paired = b.paired;
        const int per = paired ? 2 : 1;
        const ReadSoA &r = b.r;
        const uint64_t nTaxa = ft.taxCols.size();
        // room for every row of the range: names + the widest seqID / taxID + six numbers
        uint64_t nRowsOut = 0, nameBytes = 0;
        for (uint64_t q = q0; q < q1; q++) {
            const uint32_t n = std::max<uint32_t>(1, nRows[q]);
            nRowsOut += n;
            nameBytes += n * (r.nameOff[q * per + 1] - r.nameOff[q * per]);
            // a taxon outside the dense table (a malformed index) is formatted by cf_format_seqid below: its string is not
            // covered by maxSeqId, so its length is added here
            if constexpr (kWide) for (uint32_t i = 0; i < nRows[q]; i++) {
                const Row &row = rows[b.rowFirst[q] + i];
                if (row.taxon_idx >= nTaxa) nameBytes += std::strlen(cf_format_seqid(ix, row.unique_id, row.tax_id));
            }
        }
        char *const w0 = ob.room(nameBytes + nRowsOut * (ft.maxSeqId + ft.maxTax + 6 * 20 + 9));
        char *w = w0;
        for (uint64_t q = q0; q < q1; q++) {
            const size_t ra = q * per, rb = ra + 1;
            const uint64_t qlen = (r.off[ra + 1] - r.off[ra]) + (paired ? r.off[rb + 1] - r.off[rb] : 0);
            const uint32_t n = std::max<uint32_t>(1, nRows[q]);
            // readID: up to the first whitespace, a trailing /1 /2 /3 removed (aln_sink.h:2203-2217)
            const char *name = r.names.data() + r.nameOff[ra];
            size_t nl = r.nameOff[ra + 1] - r.nameOff[ra];
            if (nl >= 2 && name[nl - 2] == '/' && (name[nl - 1] == '1' || name[nl - 1] == '2' || name[nl - 1] == '3')) nl -= 2;
            for (size_t i = 0; i < nl; i++) if (std::isspace((unsigned char)name[i])) { nl = i; break; }
            for (uint32_t i = 0; i < n; i++) {
                std::memcpy(w, name, nl); w += nl;
                *w++ = '\t';
                if (nRows[q] == 0) {
                    std::memcpy(w, "unclassified\t0\t0\t", 17); w += 17;
                    w = putNum(w, score2[q]);
                    *w++ = '\t'; *w++ = '0'; *w++ = '\t';
                } else {
                    const Row &row = rows[b.rowFirst[q] + i];
                    if (!kWide || row.taxon_idx < nTaxa) {
                        if (ft.taxLeaf[row.taxon_idx] && row.unique_id < ft.uid.size()) { std::memcpy(w, ft.uid[row.unique_id].first, ft.uid[row.unique_id].second); w += ft.uid[row.unique_id].second; }
                        else { const std::string &rn = ft.rankName[row.taxon_idx]; std::memcpy(w, rn.data(), rn.size()); w += rn.size(); }
                        *w++ = '\t';
                        const std::string &tc = ft.taxCols[row.taxon_idx];
                        std::memcpy(w, tc.data(), tc.size()); w += tc.size();
                    } else if constexpr (kWide) {             // a taxon outside the dense table (not on a well-formed index)
                        const char *sid = cf_format_seqid(ix, row.unique_id, row.tax_id);
                        const size_t sl = std::strlen(sid);
                        std::memcpy(w, sid, sl); w += sl;
                        *w++ = '\t';
                        w = putNum(w, row.tax_id & 0xffffffffull);
                        if (row.tax_id >> 32) { *w++ = '.'; w = putNum(w, row.tax_id >> 32); }
                    }
                    *w++ = '\t';
                    w = putNum(w, row.score); *w++ = '\t';
                    w = putNum(w, score2[q]); *w++ = '\t';
                    w = putNum(w, row.hit_len); *w++ = '\t';
                }
                w = putNum(w, qlen); *w++ = '\t';
                w = putNum(w, n); *w++ = '\n';
            }
        }
        ob.len = (size_t)(w - w0);
    }

    void formatRange(const Batch &b, const std::vector<cf_row> &rows, const std::vector<uint32_t> &nRows,
                     const std::vector<uint32_t> &score2, uint64_t q0, uint64_t q1, std::string &s) const {
        const bool paired = b.paired;
        const int per = paired ? 2 : 1;
        const ReadSoA &r = b.r;
        for (uint64_t q = q0; q < q1; q++) {
            const size_t ra = q * per, rb = ra + 1;
            const uint64_t qlen = (r.off[ra + 1] - r.off[ra]) + (paired ? r.off[rb + 1] - r.off[rb] : 0);
            const uint32_t n = std::max<uint32_t>(1, nRows[q]);
            for (uint32_t i = 0; i < n; i++) {
                const bool uncl = nRows[q] == 0;
                const cf_row *row = uncl ? nullptr : &rows[b.rowFirst[q] + i];
                const uint64_t tax = uncl ? 0 : row->tax_id;
                bool firstField = true;
                for (int c : o.cols) {
                    if (!firstField) s.push_back('\t');
                    firstField = false;
                    switch (c) {
                        case C_READ_ID: appendReadId(s, r.names.data() + r.nameOff[ra], r.nameOff[ra + 1] - r.nameOff[ra]); break;
                        case C_SEQ_ID: s += uncl ? "unclassified" : cf_format_seqid(ix, row->unique_id, tax); break;
                        case C_TAX_ID: appendTaxId(s, tax); break;
                        case C_TAX_RANK: s += cf_tax_rank_string(cf_tax_rank(ix, tax)); break;
                        case C_TAX_NAME: s += cf_tax_name(ix, tax); break;
                        case C_SCORE: appendNum(s, uncl ? 0u : row->score); break;
                        case C_SCORE2: appendNum(s, score2[q]); break;
                        case C_HIT_LEN: appendNum(s, uncl ? 0u : row->hit_len); break;
                        case C_QUERY_LEN: appendNum(s, qlen); break;
                        case C_NUM_MATCHES: appendNum(s, n); break;
                        case C_SEQ: appendSeq(s, r, ra); if (paired) { s.push_back('_'); appendSeq(s, r, rb); } break;
                        case C_QUAL: appendQual(s, r, ra); if (paired) { s.push_back('_'); appendQual(s, r, rb); } break;
                        case C_SEQ1: appendSeq(s, r, ra); break;
                        case C_QUAL1: appendQual(s, r, ra); break;
                        case C_SEQ2: if (paired) appendSeq(s, r, rb); break;
                        case C_QUAL2: if (paired) appendQual(s, r, rb); break;
                        case C_PLACEHOLDER: s += "*0"; break;      // the reference's switch falls through "" -> "*" -> "0" (aln_sink.h:2322-2324)
                        case C_ZERO: s.push_back('0'); break;
                    }
                }
                s.push_back('\n');
            }
        }
    }

    // GPU stage of one batch on one GPU thread: reads into the thread's slot, the kernels, results out of the slot's
    // pinned buffers into the batch (the slot takes the next batch while this one is being printed)
    void classify(Batch &b, GpuThread &g) {
        const uint64_t nReads = b.r.size();
        if (nReads == 0) return;
        auto t0 = std::chrono::steady_clock::now();
        auto lap = [&](double &acc) { const auto t = std::chrono::steady_clock::now(); acc += std::chrono::duration<double>(t - t0).count(); t0 = t; };
        // Results cross the link in their narrow form (16-byte rows, five bytes per query: cf_results_narrow) whenever the batch has
        // its lengths at hand as an array and -k fits the form's six bits; they are widened into the batch's own buffers below — the
        // copy out of the slot's pinned memory that the wide form needs as well
        const bool packed = b.r.pk.valid && b.r.pk.nReads == nReads;
        const bool narrow = packed && o.khits <= 63;
        CF_TRY(cf_batch_set_result_format(g.slot, narrow ? CF_RESULTS_NARROW : CF_RESULTS_ROWS));
        if (packed) {
            // the chunk's packed form, made by the parser thread that parsed it: 3/8 byte per base straight from pinned memory
            // (2-bit words + the few words of the N mask that are not zero), nothing to pack on the device
            const PackedSoA &pk = b.r.pk;
            cf_packed_reads in{};
            in.bases = pk.words.p; in.nmask = nullptr; in.len = pk.lens.p; in.seeds = pk.seeds.p;
            in.n_reads = pk.nReads; in.n_words = pk.nWords; in.n_bases = pk.nBases; in.max_len = pk.maxLen; in.paired = b.paired ? 1 : 0;
            in.nword_idx = pk.nIdx.p; in.nword_mask = pk.nMsk.p; in.n_nwords = pk.nN;
            CF_TRY(cf_batch_upload_packed_async(g.slot, &in, g.stream));
        } else
            CF_TRY(cf_batch_upload(g.slot, b.r.seq.empty() ? reinterpret_cast<const uint8_t *>("") : b.r.seq.data(), b.r.off.data(), b.r.seeds.data(),
                                   nReads, b.paired ? 1 : 0, g.stream));
        CF_TRY(cf_classify_async(g.dev->clf, g.slot, g.stream));
        CF_TRY(cf_batch_download_async(g.slot, g.stream));
        lap(g.tm.create);
        if (narrow) {
            cf_results_narrow res;
            CF_TRY(cf_batch_wait_narrow(g.slot, &res));
            lap(g.tm.classify);
            b.nq = res.n_queries;
            b.narrowRows = defaultCols;
            if (b.narrowRows) {
                // the default columns are formatted straight from the narrow rows: out of the slot's pinned memory as they are
                if (b.rows16.size() < res.total_rows) b.rows16.resize(res.total_rows);
                if (b.qinfo.size() < b.nq) b.qinfo.resize(b.nq);
                if (b.score2.size() < b.nq) b.score2.resize(b.nq);
                if (res.total_rows) std::memcpy(b.rows16.data(), res.rows, res.total_rows * sizeof(cf_row16));
                if (b.nq) { std::memcpy(b.qinfo.data(), res.qinfo, b.nq); std::memcpy(b.score2.data(), res.score2, b.nq * 4); }
                lap(g.tm.results);
                // (the tally: by the formatter threads, range by range — emit — when they have reports of their own)
                if (g.rep && fmtReps.empty()) { CF_TRY(cf_report_add_narrow(g.rep, b.rows16.data(), b.qinfo.data(), b.r.pk.lens.p, 0, b.paired ? 1 : 0, b.nq)); lap(g.tm.report); }
                return;
            }
            if (b.rows.size() < res.total_rows) b.rows.resize(res.total_rows);
            // (each array by its own size: a recycled batch may have held narrow rows last time, which size nRows and score2 alone)
            if (b.nRows.size() < b.nq) b.nRows.resize(b.nq);
            if (b.score2.size() < b.nq) b.score2.resize(b.nq);
            if (b.maxScore.size() < b.nq) b.maxScore.resize(b.nq);
            CF_TRY(cf_results_narrow_expand(g.dev->ix, &res, b.r.pk.lens.p, 0, b.paired ? 1 : 0, b.rows.data(), b.nRows.data(), b.maxScore.data()));
            if (b.nq) std::memcpy(b.score2.data(), res.score2, b.nq * 4);
        } else {
            b.narrowRows = false;
            cf_results res;
            CF_TRY(cf_batch_wait(g.slot, &res));
            lap(g.tm.classify);
            b.nq = res.n_queries;
            // a recycled batch keeps its (already mapped) buffers
            if (b.rows.size() < res.total_rows) b.rows.resize(res.total_rows);
            if (b.nRows.size() < b.nq) b.nRows.resize(b.nq);
            if (b.score2.size() < b.nq) b.score2.resize(b.nq);
            if (b.maxScore.size() < b.nq) b.maxScore.resize(b.nq);
            if (res.total_rows) std::memcpy(b.rows.data(), res.rows, res.total_rows * sizeof(cf_row));
            if (b.nq) {
                std::memcpy(b.nRows.data(), res.n_rows, b.nq * 4);
                std::memcpy(b.score2.data(), res.score2, b.nq * 4);
                std::memcpy(b.maxScore.data(), res.max_score, b.nq * 4);
            }
        }
        lap(g.tm.results);
        if (g.rep) { CF_TRY(cf_report_add(g.rep, b.rows.data(), b.nRows.data(), b.maxScore.data(), b.nq, 0)); lap(g.tm.report); }
    }

    static void writeAll(int fd, const char *p, size_t n, bool positioned, uint64_t at) {
        while (n) {
            const ssize_t w = positioned ? ::pwrite(fd, p, n, (off_t)at) : ::write(fd, p, n);
            if (w < 0) { if (errno == EINTR) continue; die("error writing the classification output"); }
            p += w; n -= (size_t)w; at += (uint64_t)w;
        }
    }

    // n bytes at `at` of the output file through a mapping of that stretch; false = the file cannot be mapped (the caller writes)
    bool copyIntoFile(const char *text, uint64_t n, uint64_t at) {
        if (n == 0) return true;
        const uint64_t end = at + n;
        {
            std::lock_guard<std::mutex> lk(growMu);
            if (end > outSize) {
                const uint64_t want = end + (256ull << 20);            // (sparse until written; the input's end cuts it back)
                if (::ftruncate(outFd, (off_t)want) != 0) { outMap = false; return false; }
                outSize = want;
            }
        }
        static const uint64_t pg = (uint64_t)sysconf(_SC_PAGESIZE);
        const uint64_t ms = at & ~(pg - 1);
        void *m = ::mmap(nullptr, (size_t)(end - ms), PROT_READ | PROT_WRITE, MAP_SHARED, outFd, (off_t)ms);
        if (m == MAP_FAILED) { outMap = false; return false; }
        std::memcpy(static_cast<char *>(m) + (at - ms), text, (size_t)n);
        ::munmap(m, (size_t)(end - ms));
        return true;
    }

    // The device text path, one block on one GPU thread from the file to the output: the block's bytes into the thread's pinned
    // buffer, up as they are (cf_batch_upload_text: records, lengths, seeds and packed words are made on the device), the kernels,
    // and the default columns back as text (cf_batch_wait_text) — written at the block's own place in the output.  A block with a
    // record outside the plain form is parsed by the host parser instead (and its rows formatted here): same bytes out.
    void classifyText(Batch &b, GpuThread &g, size_t gi) {
        auto t0 = std::chrono::steady_clock::now();
        auto lap = [&](double &acc) { const auto t = std::chrono::steady_clock::now(); acc += std::chrono::duration<double>(t - t0).count(); t0 = t; };
        const bool mates = b.tFd2 >= 0;
        const uint64_t at2 = (b.tLen + 4095) & ~4095ull;               // the second file's block behind the first in the same buffer
        const uint64_t need = mates ? at2 + b.tLen2 : b.tLen;
        if (need + 64 > g.tinCap) {
            if (g.tin) cf_host_free(g.tin);
            g.tin = nullptr; g.tinCap = 0;
            void *q = nullptr;
            const size_t want = (size_t)(need + need / 8 + 4096);
            CF_TRY(cf_host_alloc(&q, want));
            g.tin = static_cast<char *>(q); g.tinCap = want;
        }
        readFileRange(b.tFd, g.tin, (size_t)b.tLen, b.tOff, *b.tPath);
        if (mates) readFileRange(b.tFd2, g.tin + at2, (size_t)b.tLen2, b.tOff2, *b.tPath2);
        lap(g.tm.read);
        const bool fasta = o.format == ReadFormat::Fasta;
        CF_TRY(cf_batch_set_result_format(g.slot, CF_RESULTS_NARROW));
        cf_text_reads in{};
        in.text = g.tin; in.n_bytes = b.tLen; in.format = fasta ? CF_TEXT_FASTA : CF_TEXT_FASTQ; in.global_seed = o.seed; in.max_reads = 0;
        if (mates) { in.text2 = g.tin + at2; in.n_bytes2 = b.tLen2; }
        const uint64_t per = mates ? 2 : 1;
        cf_text_info info{};
        const bool tryDevice = !(cfamd::cf_knob("CF_CLI_TEXT_HOST_PARSE") && std::atoi(cfamd::cf_knob("CF_CLI_TEXT_HOST_PARSE")));   // (the tests: every block through the fallback)
        if (tryDevice) CF_TRY(cf_batch_upload_text(g.slot, &in, g.stream, &info)); else info.irregular = 1;
        lap(g.tm.parse);
        bool onHost = info.irregular != 0;
        uint64_t nReads = info.n_reads / per;                           // queries: reads, or pairs
        if (onHost) {
            // the host parser's semantics are the reference's for every record: this block alone pays for what it holds
            b.r.clear(); b.r.hasQual = false;
            try {
                if (fasta) parseFastaChunk(g.tin, g.tin + b.tLen, b.tFirst, 0, 0, o.seed, b.r, b.tLast);
                else parseFastqChunk(g.tin, g.tin + b.tLen, b.tFirst, 0, 0, o.seed, b.r, b.tLast);
                if (mates) {
                    b.r2.clear(); b.r2.hasQual = false;
                    if (fasta) parseFastaChunk(g.tin + at2, g.tin + at2 + b.tLen2, b.tFirst, 0, 0, o.seed, b.r2, b.tLast);
                    else parseFastqChunk(g.tin + at2, g.tin + at2 + b.tLen2, b.tFirst, 0, 0, o.seed, b.r2, b.tLast);
                }
            } catch (const std::exception &e) {
                // a record the parser refuses: the run ends with the message of the FIRST such record of the file, as the reference's
                // does — the blocks before this one are parsed first (one of them failing ends this wait with its message standing)
                const std::string msg = e.what();
                (void)readChain.enter(b.tIdx);
                die(msg);
            }
            nReads = b.r.size();
            if (mates) {
                // (the two blocks were cut to hold the same records; at the end of the files one may hold fewer: the messages of the other path)
                if (b.r2.size() != nReads && !b.tLast)
                    die("Error: the mate files' records stop lining up block by block near byte " + std::to_string(b.tOff2) + " of " + *b.tPath2 +
                        " (records of other than four lines in one file only?): run with --host-io");
                if (b.r2.size() < nReads) die("Error, fewer reads in file specified with -2 than in file specified with -1");
                if (b.r2.size() > nReads) die("Error, fewer reads in file specified with -1 than in file specified with -2");
            }
            lap(g.tm.hostParse);
        }
        // the ordinal of the block's first read: -u, and the names of reads that have none (pat.cpp:838-842)
        const uint64_t base = readChain.enter(b.tIdx);
        readChain.leave(nReads);
        const uint64_t take = base >= o.upto ? 0 : std::min<uint64_t>(nReads, o.upto - base);
        if (base + nReads >= o.upto) uptoReached = true;
        const char *text = "";
        uint64_t nText = 0;
        if (take == 0) { /* (past -u: nothing of this block is printed) */ }
        else if (!onHost) {
            if (take < nReads) {                                        // the block -u ends in: once more, its first reads only
                in.max_reads = take;
                CF_TRY(cf_batch_upload_text(g.slot, &in, g.stream, &info));
                if (info.irregular || info.n_reads != take * per) die("internal error: a block changed between two parses");
            }
            CF_TRY(cf_classify_async(g.dev->clf, g.slot, g.stream));
            lap(g.tm.create);
            cf_results_text res{};
            CF_TRY(cf_batch_wait_text(g.slot, &res));
            lap(g.tm.classify);
            if (g.rep && res.n_tuple_words) CF_TRY(cf_report_add_tuples(g.rep, res.tuples, res.n_tuple_words));
            text = res.text; nText = res.n_bytes;
            b.nq = res.n_queries;
            textBatches++; g.textBlocks++;
            lap(g.tm.report);
        } else {
            g.hostBlocks++;
            if (mates || take < nReads || b.r.hasEmptyName()) {
                // reads past -u go, reads without a name are named after their ordinal, mates are laid side by side: the block's records one by one
                ReadSoA src;
                std::swap(src, b.r);
                b.r.clear(); b.r.hasQual = false;
                auto one = [&](const ReadSoA &c, uint64_t i) {
                    if (c.nameOff[i + 1] > c.nameOff[i] || std::find(c.unnamedKeep.begin(), c.unnamedKeep.end(), (uint32_t)i) != c.unnamedKeep.end()) { b.r.appendRecord(c, i); return; }
                    const std::string nm = std::to_string(base + i);
                    const uint64_t len = c.off[i + 1] - c.off[i];
                    const uint8_t *q = c.hasQual ? c.qual.data() + c.off[i] : nullptr;
                    b.r.push(c.seq.data() + c.off[i], q, len, nm.data(), nm.size(), cf_gen_rand_seed(c.seq.data() + c.off[i], q, len, nm.data(), nm.size(), o.seed));
                };
                for (uint64_t i = 0; i < take; i++) { one(src, i); if (mates) one(b.r2, i); }
            }
            b.r.pack();
            b.paired = mates;
            const PackedSoA &pk = b.r.pk;
            cf_packed_reads pin{};
            pin.bases = pk.words.p; pin.nmask = nullptr; pin.len = pk.lens.p; pin.seeds = pk.seeds.p;
            pin.n_reads = pk.nReads; pin.n_words = pk.nWords; pin.n_bases = pk.nBases; pin.max_len = pk.maxLen; pin.paired = mates ? 1 : 0;
            pin.nword_idx = pk.nIdx.p; pin.nword_mask = pk.nMsk.p; pin.n_nwords = pk.nN;
            CF_TRY(cf_batch_upload_packed_async(g.slot, &pin, g.stream));
            CF_TRY(cf_classify_async(g.dev->clf, g.slot, g.stream));
            CF_TRY(cf_batch_download_async(g.slot, g.stream));
            lap(g.tm.create);
            cf_results_narrow res;
            CF_TRY(cf_batch_wait_narrow(g.slot, &res));
            lap(g.tm.classify);
            b.nq = res.n_queries;
            b.narrowRows = true;
            if (b.rows16.size() < res.total_rows) b.rows16.resize(res.total_rows);
            if (b.qinfo.size() < b.nq) b.qinfo.resize(b.nq);
            if (b.score2.size() < b.nq) b.score2.resize(b.nq);
            if (b.nRows.size() < b.nq) b.nRows.resize(b.nq);
            if (b.rowFirst.size() < b.nq + 1) b.rowFirst.resize(b.nq + 1);
            if (res.total_rows) std::memcpy(b.rows16.data(), res.rows, res.total_rows * sizeof(cf_row16));
            if (b.nq) { std::memcpy(b.qinfo.data(), res.qinfo, b.nq); std::memcpy(b.score2.data(), res.score2, b.nq * 4); }
            uint64_t f = 0;
            for (uint64_t q = 0; q < b.nq; q++) { const uint32_t n = b.qinfo[q] & 0x3fu; b.nRows[q] = n; b.rowFirst[q] = f; f += n; }
            b.rowFirst[b.nq] = f;
            lap(g.tm.results);
            if (g.rep) CF_TRY(cf_report_add_narrow(g.rep, b.rows16.data(), b.qinfo.data(), b.r.pk.lens.p, 0, mates ? 1 : 0, b.nq));
            lap(g.tm.report);
            OutBuf &ob = *hostOut[gi];
            if (b.nq) formatDefault(b, b.rows16, b.nRows, b.score2, 0, b.nq, ob); else ob.len = 0;
            text = ob.p.get(); nText = ob.len;
            lap(g.tm.format);
        }
        // the block's place in the output: behind the blocks before it
        const uint64_t at = outChain.enter(b.tIdx);
        if (outRegular) { outChain.leave(nText); if (!(outMap && copyIntoFile(text, nText, outBase + at))) writeAll(outFd, text, (size_t)nText, true, outBase + at); }
        else { try { writeAll(outFd, text, (size_t)nText, false, 0); } catch (...) { outChain.leave(nText); throw; } outChain.leave(nText); }
        lap(g.tm.write);
        b.nq = 0;                                          // (the output stage has nothing left to do for this batch)
    }

    // the report file with its stderr lines (centrifuge.cpp:3134-3141,3231-3319; aln_sink.h:471-472)
    template <typename Hms>
    void writeReport(cf_report *r, const std::string &path, const Hms &hms) {
        std::fprintf(stderr, "report file %s\n", path.c_str());
        uint64_t it = 0; double diff = 0;
        const auto ta = std::chrono::steady_clock::now();
        const cf_status st = cf_report_write(r, path.c_str(), o.abundance ? 1 : 0, &it, &diff);
        if (st != CF_OK) die("Error: could not write the report file " + path);
        if (o.abundance) {
            std::fprintf(stderr, "Number of iterations in EM algorithm: %llu\n", (unsigned long long)it);
            std::fprintf(stderr, "Probability diff. (P - P_prev) in the last iteration: %g\n", diff);
            std::fprintf(stderr, "Calculating abundance: %s\n", hms(std::chrono::duration<double>(std::chrono::steady_clock::now() - ta).count()).c_str());
        }
    }

    // --separator: the end of input number `idx` — a marker line in the classification output, that input's own
    // report (centrifuge_report_<idx>.tsv), and the counters start over (centrifuge.cpp:3128-3226)
    template <typename Hms>
    void endInput(int idx, const Hms &hms) {
        waitWrite();
        std::fputs("#File_End_Here\n", out);
        std::fflush(out);
        writeReport(rep, "centrifuge_report_" + std::to_string(idx) + ".tsv", hms);
        CF_TRY(cf_report_reset_counts(rep));              // counters only: the reference keeps its observed tuples (aln_sink.h:84-91)
    }

    // output stage: (--separator: counters / observed tuples in output order,) TSV formatting on `threads` threads, ordered write
    void emit(Batch &b) {
        if (b.nq == 0) return;
        auto t0 = std::chrono::steady_clock::now();
        auto lap = [&](double &acc) { const auto t = std::chrono::steady_clock::now(); acc += std::chrono::duration<double>(t - t0).count(); t0 = t; };
        const uint64_t nq = b.nq;
        if (rep) {
            if (b.narrowRows) CF_TRY(cf_report_add_narrow(rep, b.rows16.data(), b.qinfo.data(), b.r.pk.lens.p, 0, b.paired ? 1 : 0, nq));
            else CF_TRY(cf_report_add(rep, b.rows.data(), b.nRows.data(), b.maxScore.data(), nq, 0));
        }
        if (b.rowFirst.size() < nq + 1) b.rowFirst.resize(nq + 1);
        if (b.narrowRows) {                              // (row counts out of the queries' bytes, along with the row offsets)
            if (b.nRows.size() < nq) b.nRows.resize(nq);
            uint64_t f = 0;
            for (uint64_t q = 0; q < nq; q++) { const uint32_t n = b.qinfo[q] & 0x3fu; b.nRows[q] = n; b.rowFirst[q] = f; f += n; }
            b.rowFirst[nq] = f;
        } else { uint64_t f = 0; for (uint64_t q = 0; q < nq; q++) { b.rowFirst[q] = f; f += b.nRows[q]; } b.rowFirst[nq] = f; }
        lap(tm.report);
        const int nt = (int)std::min<uint64_t>((uint64_t)o.threads, std::max<uint64_t>(1, nq / 4096));
        std::vector<std::string> parts(defaultCols ? 0 : nt);
        std::vector<OutBuf> &fmtBufs = fmtSets[fmtCur];
        if ((int)fmtBufs.size() < nt) fmtBufs.resize(nt);
        std::vector<std::thread> th;
        std::vector<cf_status> tallySt((size_t)nt, CF_OK);
        const bool tallyHere = b.narrowRows && !rep && (int)fmtReps.size() >= nt;
        for (int t = 0; t < nt; t++) {
            const uint64_t q0 = nq * t / nt, q1 = nq * (t + 1) / nt;
            auto one = [&, t, q0, q1] {
                if (tallyHere)
                    tallySt[(size_t)t] = cf_report_add_narrow(fmtReps[(size_t)t], b.rows16.data() + b.rowFirst[q0], b.qinfo.data() + q0,
                                                              b.r.pk.lens.p + q0 * (b.paired ? 2 : 1), 0, b.paired ? 1 : 0, q1 - q0);
                if (defaultCols && b.narrowRows) formatDefault(b, b.rows16, b.nRows, b.score2, q0, q1, fmtBufs[t]);
                else if (defaultCols) formatDefault(b, b.rows, b.nRows, b.score2, q0, q1, fmtBufs[t]);
                else { parts[t].reserve((q1 - q0) * 48); formatRange(b, b.rows, b.nRows, b.score2, q0, q1, parts[t]); }
            };
            if (nt == 1) one(); else th.emplace_back(one);
        }
        for (auto &x : th) x.join();
        for (const cf_status st_ : tallySt) CF_TRY(st_);
        lap(tm.format);
        waitWrite();                                     // the batch before this one is in the file (and its buffers are free again)
        if (defaultCols) {
            // this batch's text goes out while the next one is formatted into the other set of buffers
            std::vector<OutBuf> *set = &fmtBufs;
            writeFut = std::async(std::launch::async, [this, set, nt] {
                const auto w0 = std::chrono::steady_clock::now();
                for (int t = 0; t < nt; t++) {
                    const size_t n = (*set)[t].len;
                    if (n && std::fwrite((*set)[t].p.get(), 1, n, out) != n) die("error writing the classification output");
                }
                writeBusy += std::chrono::duration<double>(std::chrono::steady_clock::now() - w0).count();
            });
            fmtCur ^= 1;
        } else {
            for (int t = 0; t < nt; t++) {
                const size_t n = parts[t].size();
                if (n && std::fwrite(parts[t].data(), 1, n, out) != n) die("error writing the classification output");
            }
        }
        lap(tm.write);
    }

    // The end of a run on N devices: the per-thread tallies merged into one report (observed tuples for the
    // EM), and the per-taxon counters the kernels kept on every device summed — ONE RCCL all-reduce group over the devices
    // (a host sum when the list names one device twice: RCCL wants distinct GPUs) — and checked against the merged tally
    // before they become the report's numReads / numUniqueReads.
    cf_report *finishReport() {
        cf_report *final = gts[0].rep;
        std::vector<cf_report *> others;
        for (size_t t = 1; t < gts.size(); t++) others.push_back(gts[t].rep);
        for (cf_report *r : fmtReps) others.push_back(r);
        for (cf_report *r : others) {
            uint64_t need = 0;
            CF_TRY(cf_report_serialize(r, nullptr, 0, &need));
            std::vector<uint64_t> img(need);
            CF_TRY(cf_report_serialize(r, img.data(), need, &need));
            CF_TRY(cf_report_merge(final, img.data(), need));
        }
        const uint64_t nTaxa = cf_index_num_taxa(ix);
        std::vector<uint64_t> nReads(nTaxa, 0), nUnique(nTaxa, 0);
        bool distinct = true;
        for (size_t i = 0; i < devs.size(); i++) for (size_t j = 0; j < i; j++) distinct = distinct && devs[i].id != devs[j].id;
        const bool viaRccl = distinct && (devs.size() > 1 || cfamd::cf_knob("CF_CLI_RCCL"));
        if (viaRccl) {
            std::vector<int> ids;
            std::vector<cf_classifier *> cls;
            for (auto &d : devs) { ids.push_back(d.id); cls.push_back(d.clf); }
            std::vector<void *> comms(devs.size(), nullptr);
            CF_TRY(cf_comm_init_all((int)devs.size(), ids.data(), comms.data()));
            const cf_status st = cf_counts_allreduce_group(cls.data(), comms.data(), (int)devs.size());
            for (void *c : comms) cf_comm_destroy(c);
            CF_TRY(st);
            CF_TRY(cf_counts_get(devs[0].clf, nReads.data(), nUnique.data()));        // every device now holds the sum
            if (!o.quiet && o.timing) std::fprintf(stderr, "Per-taxon counters all-reduced over %zu GPU(s) with RCCL\n", devs.size());
        } else {
            std::vector<uint64_t> a(nTaxa), b(nTaxa);
            for (auto &d : devs) {
                CF_TRY(cf_counts_get(d.clf, a.data(), b.data()));
                for (uint64_t i = 0; i < nTaxa; i++) { nReads[i] += a[i]; nUnique[i] += b[i]; }
            }
        }
        if (textBatches) {
            // Batches whose rows never reached the host (the device text path): the devices' counters ARE the tally — every batch of the
            // run is in them —, the perfect single assignments come from the devices as well (summed with the others), the perfect
            // tuples were added batch by batch; what the host did see (blocks outside the plain form) may not exceed them
            std::vector<uint64_t> nSingle(nTaxa, 0), c(nTaxa);
            if (viaRccl) CF_TRY(cf_counts_get_single(devs[0].clf, nSingle.data()));
            else for (auto &d : devs) { CF_TRY(cf_counts_get_single(d.clf, c.data())); for (uint64_t i = 0; i < nTaxa; i++) nSingle[i] += c[i]; }
            if (cf_report_adopt_device_tally(final, nReads.data(), nUnique.data(), nSingle.data(), nTaxa) != CF_OK)
                die("internal error: the per-taxon counters of the devices are below the classified rows the host saw");
            return final;
        }
        // the self-check of every run: two tallies of the same reads — the devices' counters (summed by RCCL) and the rows the
        // output stage saw — must agree taxon by taxon (tests/test_report.py feeds cf_report_adopt_counts a tally that is off by one)
        if (cf_report_adopt_counts(final, nReads.data(), nUnique.data(), nTaxa) != CF_OK)
            die("internal error: the per-taxon counters of the devices disagree with the classified rows");
        return final;
    }
};

int run(int argc, const char **argv) {
    const Opts o = parse(argc, argv);
    const auto t0 = std::chrono::steady_clock::now();
    auto secs = [&](std::chrono::steady_clock::time_point a) { return std::chrono::duration<double>(std::chrono::steady_clock::now() - a).count(); };
    auto hms = [](double s) { char b[32]; const int t = (int)s; std::snprintf(b, sizeof b, "%02d:%02d:%02d", t / 3600, (t / 60) % 60, t % 60); return std::string(b); };

    Runner R{o};
    // The reference works through its inputs one at a time, mate files first (centrifuge.cpp:3007-3040): read
    // ordinals, -s/-u and the names of unnamed reads start over with every input.
    struct Input { std::string f1, f2; bool paired; };
    std::vector<Input> inputs;
    for (size_t i = 0; i < o.mates1.size(); i++) inputs.push_back({o.mates1[i], o.mates2[i], true});
    for (const auto &q : o.queries) inputs.push_back({q, std::string(), false});
    if (!o.dumpReads) {
        const std::string base = findIndex(o.index);
        // the device list: --gpu-list, or --gpus N devices from --device on
        std::vector<int> ids = o.gpuList;
        if (ids.empty()) {
            int n = o.gpus;
            if (n < 0) { n = cf_device_count() - o.device; if (n < 1) die("centrifuge-class: no HIP device (this program has no CPU path)"); }
            for (int i = 0; i < n; i++) ids.push_back(o.device + i);
        }
        // --separator reports per input, in input order: one device, one GPU thread, the tally kept by the output stage
        const bool ordered = o.separator;
        if (ordered && ids.size() > 1) die("--separator works on one GPU: drop --gpus / --gpu-list");
        // The size of the job, for the index planner (cf_index_options::expected_reads): the input files' bytes over the fewest
        // bytes a read of that format can reasonably take (an over-estimate errs towards more tables); compressed input counts
        // four-fold, input whose size is not to be had (stdin, a pipe; reads on the command line aside) as unknown: every table
        // that fits.  A replica is planned for its share of the job (its build time does not shrink with the device count).
        uint64_t expected = 0;
        if (o.expectedReads >= 0) expected = (uint64_t)o.expectedReads;
        else if (o.format == ReadFormat::CmdLine) expected = 1 + o.queries.size() + 2 * o.mates1.size();
        else {
            const double perRead = o.format == ReadFormat::Fastq ? 100.0 : o.format == ReadFormat::Fasta ? 60.0 : 30.0;
            double reads = 0;
            bool unknown = false;
            auto add = [&](const std::string &f) {
                struct stat sb;
                if (f == "-" || ::stat(f.c_str(), &sb) != 0 || !S_ISREG(sb.st_mode)) { unknown = true; return; }
                const bool packed = f.size() > 3 && (f.compare(f.size() - 3, 3, ".gz") == 0 || (f.size() > 4 && f.compare(f.size() - 4, 4, ".bz2") == 0));
                reads += (double)sb.st_size * (packed ? 4.0 : 1.0) / perRead;
            };
            for (const auto &in : inputs) { add(in.f1); if (in.paired) add(in.f2); }
            expected = unknown ? 0 : (uint64_t)reads + 1;
        }
        expected /= std::max<size_t>(1, ids.size());
        auto tl = std::chrono::steady_clock::now();
        R.beforeOpenS = secs(t0);
        R.devs.resize(ids.size());
        {   // every device loads its replica of the index at the same time
            std::vector<std::thread> th;
            std::vector<std::string> errs(ids.size());
            for (size_t i = 0; i < ids.size(); i++) {
                R.devs[i].id = ids[i];
                th.emplace_back([&, i] {
                    const auto td = std::chrono::steady_clock::now();
                    struct Done { double &s; std::chrono::steady_clock::time_point t; ~Done() { s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t).count(); } } done{R.devs[i].openS, td};
                    cf_index_options io;
                    std::memset(&io, 0, sizeof io);
                    io.small_range_rows = o.smallRangeRows; io.hbm_budget_bytes = (uint64_t)(o.hbmBudgetGb * 1e9); io.expected_reads = expected;
                    // the loader thread runs (and places its pinned staging buffers) on the NUMA node the GPU hangs off
                    (void)cf_thread_bind_near_device(ids[i], &R.devs[i].numaNode);
                    // (CF_TEST_FAIL_OPEN=<i>, behind the knob gate: replica i comes back out of memory — the error path of one replica
                    // among N, which the one-GPU test box cannot provoke for real)
                    const char *fo = cfamd::cf_knob("CF_TEST_FAIL_OPEN");
                    if (fo && std::atoi(fo) == (int)i) { errs[i] = "centrifuge-class: HIP error: hipMalloc: out of memory (CF_TEST_FAIL_OPEN)"; return; }
                    const cf_status s_ = cf_index_open_ex(base.c_str(), ids[i], &io, &R.devs[i].ix);
                    if (s_ != CF_OK) errs[i] = std::string("centrifuge-class: ") + cf_strerror(s_) + ": " + cf_last_error();
                });
            }
            for (auto &t : th) t.join();
            // one replica's failure ends the run: the message names the device, the replicas that did open are closed by ~Runner
            for (size_t i = 0; i < errs.size(); i++) if (!errs[i].empty()) die(errs[i] + " (device " + std::to_string(ids[i]) + ", replica " + std::to_string(i + 1) + " of " + std::to_string(ids.size()) + ")");
        }
        R.ix = R.devs[0].ix;
        R.makeFormatTables();
        R.indexOpenS = secs(tl);
        if (o.timing) {
            std::fprintf(stderr, "Time loading forward index: %s\n", hms(R.indexOpenS).c_str());
            // every device reads the files through the page cache and makes its own tables, all at once: the wall time is the slowest one's
            std::string per;
            for (const auto &d : R.devs) { char b[48]; std::snprintf(b, sizeof b, "%sdevice %d %.2f", per.empty() ? "" : ", ", d.id, d.openS); per += b; }
            std::fprintf(stderr, "Index open seconds: %s\n", per.c_str());
            {   // what the planner made of it on the first device (cf_index_describe)
                cf_index_config cfg;
                if (cf_index_describe(R.devs[0].ix, &cfg) == CF_OK)
                    std::fprintf(stderr, "Index tables: wide ftab %d bases, text tables rate %d, planes %d, pair planes %d, resolve table rate %d, small ranges %d rows; %.1f GB on the device, made in %.2f s\n",
                                 cfg.wide_ftab_chars, cfg.text_verify_rate, cfg.occ_planes, cfg.pair_planes, cfg.resolve_rate, cfg.small_range_rows, (double)cfg.total_bytes / 1e9, cfg.build_ms / 1e3);
            }
            std::fprintf(stderr, "Job size offered to the index planner: %llu reads per device%s\n", (unsigned long long)expected, expected ? "" : " (unknown: every table that fits)");
            per.clear();
            for (const auto &d : R.devs) { char b[48]; std::snprintf(b, sizeof b, "%sdevice %d node %d", per.empty() ? "" : ", ", d.id, d.numaNode); per += b; }
            std::fprintf(stderr, "NUMA placement of the loader and GPU threads: %s\n", per.c_str());
        }
        cf_params p;
        cf_params_default(&p);
        p.khits = o.khits; p.min_hitlen = o.minHitLen; p.rank_slot = rankSlot(o.rank); p.tree_traverse = o.traverse ? 1 : 0;
        p.host_taxids = o.hostTaxids.data(); p.n_host = (int32_t)o.hostTaxids.size();
        p.exclude_taxids = o.excludeTaxids.data(); p.n_exclude = (int32_t)o.excludeTaxids.size();
        for (auto &d : R.devs) CF_TRY(cf_classifier_create(d.ix, &p, &d.clf));
        // The device text path (round 6): whole blocks of a plain FASTA / FASTQ file up as text, the default columns back as text —
        // without trimming or a skip, the default columns, -k <= 63 (the narrow rows' six bits).  Every GPU thread
        // then also reads its blocks and writes its text, so there are more of them (each with a slot on the device).
        R.textCapable = !ordered && R.defaultCols && (o.format == ReadFormat::Fasta || o.format == ReadFormat::Fastq) &&
                        o.trim5 == 0 && o.trim3 == 0 && o.skip == 0 && o.khits <= 63 && !o.hostIo &&
                        !(cfamd::cf_knob("CF_CLI_DEVICE_TEXT") && !std::atoi(cfamd::cf_knob("CF_CLI_DEVICE_TEXT")));
        const int slots = ordered ? 1 : (R.textCapable && !o.slotsSet) ? std::max(2, std::min(6, o.threads / 2)) : o.slots;
        R.gts.resize(R.devs.size() * (size_t)slots);
        for (size_t t = 0; t < R.gts.size(); t++) {
            GpuThread &g = R.gts[t];
            g.dev = &R.devs[t / (size_t)slots];
            CF_TRY(cf_stream_create(g.dev->id, &g.stream));
            CF_TRY(cf_batch_alloc(g.dev->clf, 0, 0, &g.slot));
            if (!ordered) CF_TRY(cf_report_create(R.ix, &g.rep));
        }
        if (R.textCapable) { for (size_t t = 0; t < R.gts.size(); t++) R.hostOut.push_back(new Runner::OutBuf()); }
        if (ordered) CF_TRY(cf_report_create(R.ix, &R.rep));
        else if (R.defaultCols) {                            // (narrow rows are tallied by the threads that format them)
            R.fmtReps.assign((size_t)std::max(1, o.threads), nullptr);
            for (auto &r : R.fmtReps) CF_TRY(cf_report_create(R.ix, &r));
        }
        if (!o.outFile.empty()) {
            R.out = std::fopen(o.outFile.c_str(), "w+b");         // (readable as well: the device text path maps the file)
            if (!R.out) die("Error: Could not open alignment output file " + o.outFile);
        }
        // header (centrifuge.cpp:2985-2992); none under --out-fmt sam
        if (!o.samFormat) {
            std::string h;
            for (size_t i = 0; i < o.colNames.size(); i++) { if (i) h.push_back('\t'); h += o.colNames[i]; }
            h.push_back('\n');
            std::fwrite(h.data(), 1, h.size(), R.out);
        }
    }

    // ---- three stages: assemble (this thread, fed by the ingest pool) -> GPU threads (batches dealt to whichever
    //      is free: mates stay together, a batch never splits) -> output stage (batches back in input order)
    std::mutex mu;
    std::condition_variable cv;
    std::deque<std::unique_ptr<Batch>> queue;
    std::map<uint64_t, std::unique_ptr<Batch>> done;   // classified batches waiting for their turn to be printed
    uint64_t nextSeq = 0, nextOut = 0, emitted = 0;    // batches submitted / taken by the output stage / done with it
    std::vector<std::unique_ptr<Batch>> spare;         // printed batches go back to the reader: their buffers are already mapped
    bool producerDone = false;
    size_t gpuRunning = R.gts.size();
    std::string workerError;
    const size_t inFlightCap = 2 * std::max<size_t>(1, R.gts.size()) + 2;
    std::vector<std::thread> workers;
    for (size_t wi = 0; wi < R.gts.size(); wi++) workers.emplace_back([&, wi] {
        GpuThread &g = R.gts[wi];
        // a GPU thread lives on its device's NUMA node: the slot's pinned buffers are first touched (and the narrow rows copied
        // out of them) there — what `numactl --cpunodebind` does for a one-GPU process, per thread
        if (g.dev) (void)cf_thread_bind_near_device(g.dev->id, nullptr);
        try {
            for (;;) {
                std::unique_ptr<Batch> b;
                {
                    std::unique_lock<std::mutex> lk(mu);
                    cv.wait(lk, [&] { return !queue.empty() || producerDone || !workerError.empty(); });
                    if (!workerError.empty() || queue.empty()) break;
                    b = std::move(queue.front());
                    queue.pop_front();
                }
                cv.notify_all();
                if (b->endOfInput < 0) { if (b->isText) R.classifyText(*b, g, wi); else R.classify(*b, g); }
                {
                    std::lock_guard<std::mutex> lk(mu);
                    const uint64_t sq = b->seq;
                    done[sq] = std::move(b);
                }
                cv.notify_all();
            }
        } catch (const std::exception &e) {
            { std::lock_guard<std::mutex> lk(mu); if (workerError.empty()) workerError = e.what(); }
            R.readChain.fail(); R.outChain.fail();           // (threads waiting for this one's block must not wait for ever)
        }
        { std::lock_guard<std::mutex> lk(mu); gpuRunning--; }
        cv.notify_all();
    });
    std::thread writer([&] {
        try {
            for (;;) {
                std::unique_ptr<Batch> b;
                {
                    std::unique_lock<std::mutex> lk(mu);
                    cv.wait(lk, [&] { return done.count(nextOut) || gpuRunning == 0 || !workerError.empty(); });
                    if (!workerError.empty()) return;
                    auto it = done.find(nextOut);
                    if (it == done.end()) return;                 // the GPU threads are gone and the next batch never came
                    b = std::move(it->second);
                    done.erase(it);
                    nextOut++;
                }
                cv.notify_all();
                if (b->endOfInput >= 0) R.endInput(b->endOfInput, hms); else if (!b->isText) R.emit(*b);
                { std::lock_guard<std::mutex> lk(mu); spare.push_back(std::move(b)); emitted++; }
                cv.notify_all();
            }
        } catch (const std::exception &e) { std::lock_guard<std::mutex> lk(mu); if (workerError.empty()) workerError = e.what(); }
        cv.notify_all();
    });
    auto joinAll = [&] { for (auto &w : workers) w.join(); writer.join(); };
    // a batch enters the pipeline: at most inFlightCap of them between here and the printed output
    auto submit = [&](std::unique_ptr<Batch> b) -> bool {
        std::unique_lock<std::mutex> lk(mu);
        b->seq = nextSeq++;
        cv.wait(lk, [&] { return b->seq < nextOut + inFlightCap || !workerError.empty(); });
        if (!workerError.empty()) return false;
        queue.push_back(std::move(b));
        lk.unlock();
        cv.notify_all();
        return true;
    };
    // every batch submitted so far is printed (false: the run is failing)
    auto drain = [&]() -> bool {
        std::unique_lock<std::mutex> lk(mu);
        cv.wait(lk, [&] { return emitted == nextSeq || !workerError.empty(); });
        return workerError.empty();
    };
    auto ts = std::chrono::steady_clock::now();
    uint64_t benchReads = 0, benchBases = 0;
    try {
      size_t lastSeq = 0, lastNames = 0, lastReads = 0;
      bool lastQual = false, aborted = false;
      for (size_t fi = 0; fi < inputs.size() && !aborted; fi++) {
        const Input &in = inputs[fi];
        const bool paired = in.paired;
        struct stat isb;
        uint64_t resume1 = 0, resume2 = 0, resumeId = 0;              // where the parser pool takes over from the text path (mates only)
        if (R.textCapable && paired && !o.dumpReads && in.f1 != "-" && in.f2 != "-" && ::stat(in.f1.c_str(), &isb) == 0 && S_ISREG(isb.st_mode) &&
            ::stat(in.f2.c_str(), &isb) == 0 && S_ISREG(isb.st_mode)) {
            // Mates on the device text path: the first file is cut like an unpaired one; the second where it holds as many records
            // as the first file's block — counted here, 32 bytes at a time over mappings of the two files, by the rule the device's
            // record pass counts by (a FASTA record starts at every '>', a FASTQ record is four lines).  Files that do not keep to
            // that (wrapped FASTQ lines, blank lines) change over to the parser pool where they stop doing so.
            ByteSource src1(in.f1, 1), src2(in.f2, 1);
            int fd1 = -1, fd2 = -1; uint64_t fs1 = 0, fs2 = 0;
            if (src1.regularFile(fd1, fs1) && src2.regularFile(fd2, fs2) && fs1 && fs2) {
                if (!drain()) { aborted = true; break; }
                R.waitWrite();
                std::fflush(R.out);
                R.outFd = fileno(R.out);
                struct stat sb;
                R.outRegular = ::fstat(R.outFd, &sb) == 0 && S_ISREG(sb.st_mode);
                R.outBase = R.outRegular ? (uint64_t)ftello(R.out) : 0;
                R.outSize = R.outRegular ? (uint64_t)sb.st_size : 0;
                R.outMap = R.outRegular && cfamd::cf_knob("CF_CLI_MAP_OUTPUT") && std::atoi(cfamd::cf_knob("CF_CLI_MAP_OUTPUT"));
                R.readChain.reset(); R.outChain.reset(); R.uptoReached = false;
                const size_t kBlock = cfamd::cf_knob("CF_TEXT_BLOCK") ? std::max<size_t>(4096, std::strtoull(cfamd::cf_knob("CF_TEXT_BLOCK"), nullptr, 10)) : (size_t)(32u << 20);
                void *m1 = ::mmap(nullptr, (size_t)fs1, PROT_READ, MAP_SHARED, fd1, 0), *m2 = ::mmap(nullptr, (size_t)fs2, PROT_READ, MAP_SHARED, fd2, 0);
                struct Unmap { void *a, *b; size_t na, nb; ~Unmap() { if (a != MAP_FAILED) ::munmap(a, na); if (b != MAP_FAILED) ::munmap(b, nb); } } unmap{m1, m2, (size_t)fs1, (size_t)fs2};
                bool changeOver = m1 == MAP_FAILED || m2 == MAP_FAILED;
                const char *p1 = static_cast<const char *>(m1), *p2 = static_cast<const char *>(m2);
                const bool fasta = o.format == ReadFormat::Fasta;
                uint64_t pos1 = 0, pos2 = 0, idx = 0, pairs = 0;
                while (!changeOver && pos1 < fs1 && !R.uptoReached) {
                    const auto tp0 = std::chrono::steady_clock::now();
                    const uint64_t cut1 = nextRecordCut(fd1, pos1, fs1, kBlock, fasta, in.f1);
                    uint64_t n = 0, end2 = 0;
                    if (fasta) {
                        n = countByte(p1 + pos1, (size_t)(cut1 - pos1), '>');
                        if (p1[pos1] != '>' || pos2 >= fs2 || p2[pos2] != '>') { changeOver = true; break; }
                        const uint64_t e = behindNthByte(p2 + pos2, (size_t)(fs2 - pos2), '>', n + 1);
                        if (e != ~0ull) end2 = pos2 + e - 1;
                        else if (countByte(p2 + pos2, (size_t)(fs2 - pos2), '>') == n) end2 = fs2;
                        else { changeOver = true; break; }              // the second file holds fewer records: the other path says so
                    } else {
                        const uint64_t nl = countByte(p1 + pos1, (size_t)(cut1 - pos1), '\n');
                        if ((nl & 3) || p1[cut1 - 1] != '\n' || p1[pos1] != '@' || pos2 >= fs2 || p2[pos2] != '@') { changeOver = true; break; }
                        n = nl >> 2;
                        const uint64_t e = behindNthByte(p2 + pos2, (size_t)(fs2 - pos2), '\n', 4 * n);
                        if (e == ~0ull) { changeOver = true; break; }
                        end2 = pos2 + e;
                        if (end2 < fs2 && p2[end2] != '@') { changeOver = true; break; }
                        // ... and the four lines before the cut must have a record's shape (lines wrapped anywhere near the block's
                        // end shift everything behind them: name, bases, '+', as many qualities)
                        uint64_t ls[5];
                        ls[4] = end2;
                        bool shape = true;
                        for (int k = 3; k >= 0 && shape; k--) {
                            uint64_t q = ls[k + 1] - 1;                 // the '\n' that ends line k
                            while (q > pos2 && p2[q - 1] != '\n') q--;
                            ls[k] = q;
                            shape = q >= pos2 && (k == 0 || q > pos2);
                        }
                        if (!shape || p2[ls[0]] != '@' || p2[ls[2]] != '+' || ls[2] - ls[1] != ls[4] - ls[3]) { changeOver = true; break; }
                    }
                    if ((cut1 == fs1) != (end2 == fs2)) { changeOver = true; break; }      // one file goes on where the other ends
                    std::unique_ptr<Batch> b;
                    { std::lock_guard<std::mutex> lk(mu); if (!spare.empty()) { b = std::move(spare.back()); spare.pop_back(); } }
                    if (!b) b = std::make_unique<Batch>();
                    b->nq = 0; b->endOfInput = -1; b->paired = true; b->narrowRows = false;
                    b->isText = true; b->tFd = fd1; b->tOff = pos1; b->tLen = cut1 - pos1; b->tFirst = pos1 == 0; b->tLast = cut1 == fs1; b->tIdx = idx++; b->tPath = &in.f1;
                    b->tFd2 = fd2; b->tOff2 = pos2; b->tLen2 = end2 - pos2; b->tPath2 = &in.f2;
                    pos1 = cut1; pos2 = end2; pairs += n;
                    const auto tp1 = std::chrono::steady_clock::now();
                    R.tm.produce += std::chrono::duration<double>(tp1 - tp0).count();
                    if (!submit(std::move(b))) { aborted = true; break; }
                    R.tm.wait += std::chrono::duration<double>(std::chrono::steady_clock::now() - tp1).count();
                }
                if (aborted || !drain()) { aborted = true; break; }
                if (R.outRegular && R.outSize > R.outBase + R.outChain.sum && ::ftruncate(R.outFd, (off_t)(R.outBase + R.outChain.sum)) != 0) die("error writing the classification output");
                if (R.outRegular && fseeko(R.out, (off_t)(R.outBase + R.outChain.sum), SEEK_SET) != 0) die("error writing the classification output");
                if (!changeOver || R.uptoReached) continue;
                resume1 = pos1; resume2 = pos2; resumeId = pairs;     // the parser pool goes on from here
            }
        }
        if (R.textCapable && !paired && !o.dumpReads && in.f1 != "-" && ::stat(in.f1.c_str(), &isb) == 0 && S_ISREG(isb.st_mode)) {
            // The device text path: a plain file (not stdin, a pipe or a compressed one) is dealt out to the GPU threads as ranges
            // that start and end at record starts; each reads its range, sends it up as it is and writes the text that comes back.
            ByteSource src(in.f1, 1);                             // (throws when the file cannot be opened, as the other path does)
            int fd = -1; uint64_t fsize = 0;
            if (src.regularFile(fd, fsize)) {
                if (!drain()) { aborted = true; break; }          // nothing of an earlier input is still on its way
                R.waitWrite();
                std::fflush(R.out);
                R.outFd = fileno(R.out);
                struct stat sb;
                R.outRegular = ::fstat(R.outFd, &sb) == 0 && S_ISREG(sb.st_mode);
                R.outBase = R.outRegular ? (uint64_t)ftello(R.out) : 0;
                R.outSize = R.outRegular ? (uint64_t)sb.st_size : 0;
                R.outMap = R.outRegular && cfamd::cf_knob("CF_CLI_MAP_OUTPUT") && std::atoi(cfamd::cf_knob("CF_CLI_MAP_OUTPUT"));
                R.readChain.reset(); R.outChain.reset(); R.uptoReached = false;
                const size_t kBlock = cfamd::cf_knob("CF_TEXT_BLOCK") ? std::max<size_t>(4096, std::strtoull(cfamd::cf_knob("CF_TEXT_BLOCK"), nullptr, 10)) : (size_t)(64u << 20);
                uint64_t pos = 0, idx = 0;
                while (pos < fsize && !R.uptoReached) {
                    const auto tp0 = std::chrono::steady_clock::now();
                    const uint64_t cut = nextRecordCut(fd, pos, fsize, kBlock, o.format == ReadFormat::Fasta, in.f1);
                    std::unique_ptr<Batch> b;
                    { std::lock_guard<std::mutex> lk(mu); if (!spare.empty()) { b = std::move(spare.back()); spare.pop_back(); } }
                    if (!b) b = std::make_unique<Batch>();
                    b->nq = 0; b->endOfInput = -1; b->paired = false; b->narrowRows = false;
                    b->isText = true; b->tFd = fd; b->tOff = pos; b->tLen = cut - pos; b->tFirst = pos == 0; b->tLast = cut == fsize; b->tIdx = idx++; b->tPath = &in.f1;
                    b->tFd2 = -1; b->tLen2 = 0;
                    pos = cut;
                    const auto tp1 = std::chrono::steady_clock::now();
                    R.tm.produce += std::chrono::duration<double>(tp1 - tp0).count();
                    if (!submit(std::move(b))) { aborted = true; break; }
                    R.tm.wait += std::chrono::duration<double>(std::chrono::steady_clock::now() - tp1).count();
                }
                if (aborted || !drain()) { aborted = true; break; }
                // the output continues behind this input's text
                if (R.outRegular && R.outSize > R.outBase + R.outChain.sum && ::ftruncate(R.outFd, (off_t)(R.outBase + R.outChain.sum)) != 0) die("error writing the classification output");
                if (R.outRegular && fseeko(R.out, (off_t)(R.outBase + R.outChain.sum), SEEK_SET) != 0) die("error writing the classification output");
                continue;
            }
        }
        // single-end chunks travel whole (they become the batch): the parser threads also make their packed form; mates are
        // interleaved into the batch pair by pair, their packed words along with their bytes (every read starts on a word).
        // A batch put together record by record (-s / -u windows, unnamed reads) goes up as bytes.  (CF_CLI_PACKED=0: bytes always; with --dump-reads
        // the knob CF_DUMP_FROM_PACKED=1 prints the bases back out of the packed form — the tests' window on it.)
        const bool dumpPacked = o.dumpReads && cfamd::cf_knob("CF_DUMP_FROM_PACKED") && std::atoi(cfamd::cf_knob("CF_DUMP_FROM_PACKED"));
        const bool wantPacked = o.dumpReads ? dumpPacked : !(cfamd::cf_knob("CF_CLI_PACKED") && !std::atoi(cfamd::cf_knob("CF_CLI_PACKED")));
        ChunkedReader s1({in.f1}, o.format, o.trim5, o.trim3, o.seed, o.threads, wantPacked, resume1);
        std::unique_ptr<ChunkedReader> s2;
        if (paired) s2.reset(new ChunkedReader({in.f2}, o.format, o.trim5, o.trim3, o.seed, o.threads, wantPacked, resume2));
        ReadSoA c1, c2;
        size_t i1 = 0, i2 = 0;
        bool c1Named = false, c2Named = false; // the current chunk of the stream has no unnamed read (bulk path allowed)
        auto fetch = [&](ChunkedReader &src, ReadSoA &c, size_t &i) {
            while (i >= c.size()) {
                if (!src.next(c)) return false;
                i = 0;
                if (&c == &c1) c1Named = !c1.hasEmptyName(); else c2Named = !c2.hasEmptyName();
            }
            return true;
        };
        // a record into the batch; a read without a name is named after its ordinal (pat.cpp:838-842)
        auto take = [&](Batch &b, const ReadSoA &c, size_t i, uint64_t id) {
            if (c.nameOff[i + 1] > c.nameOff[i]) { b.r.appendRecord(c, i); return; }
            // (a FASTQ record without a base letter leaves the reader before the default name is set, pat.cpp:985-993)
            if (std::find(c.unnamedKeep.begin(), c.unnamedKeep.end(), (uint32_t)i) != c.unnamedKeep.end()) { b.r.appendRecord(c, i); return; }
            const std::string nm = std::to_string(id);
            const uint64_t len = c.off[i + 1] - c.off[i];
            const uint8_t *q = c.hasQual ? c.qual.data() + c.off[i] : nullptr;
            b.r.push(c.seq.data() + c.off[i], q, len, nm.data(), nm.size(),
                     cf_gen_rand_seed(c.seq.data() + c.off[i], q, len, nm.data(), nm.size(), o.seed));
        };
        uint64_t rdid = resumeId;
        uint64_t dumpWordAt = 0, dumpNAt = 0, dumpBatches = 0, dumpFromPacked = 0;
        bool more = true;
        while (more) {
            const auto tp0 = std::chrono::steady_clock::now();
            std::unique_ptr<Batch> b;
            { std::lock_guard<std::mutex> lk(mu); if (!spare.empty()) { b = std::move(spare.back()); spare.pop_back(); } }
            if (b) { b->r.clear(); b->r.hasQual = false; b->nq = 0; b->endOfInput = -1; b->isText = false; }
            else {                                  // a new batch starts with the footprint of the last one: no regrowth copies
                b = std::make_unique<Batch>();
                b->r.seq.reserve(lastSeq); b->r.names.reserve(lastNames); b->r.off.reserve(lastReads + 1);
                b->r.nameOff.reserve(lastReads + 1); b->r.seeds.reserve(lastReads);
                if (lastQual) b->r.qual.reserve(lastSeq);
            }
            b->paired = paired;
            while (b->r.size() < o.batch * (paired ? 2 : 1)) {
                if (!fetch(s1, c1, i1)) {
                    if (paired && fetch(*s2, c2, i2)) die("Error, fewer reads in file specified with -1 than in file specified with -2");
                    more = false;
                    break;
                }
                if (!paired && i1 == 0 && b->r.size() == 0 && c1Named && c1.size() <= o.batch && rdid >= o.skip && rdid + c1.size() <= o.upto && c1.size() > 1) {
                    // a whole parsed chunk, untouched by -s / -u / --batch: it BECOMES the batch (its arrays are moved, not copied —
                    // the parser threads did all the work; this thread only hands chunks on)
                    const uint64_t cnt = c1.size();
                    std::swap(b->r, c1);
                    c1.clear(); c1.hasQual = false;
                    i1 = 0; rdid += cnt;
                    break;
                }
                if (!paired) {
                    // bulk path: as many of the chunk's remaining records as fit the batch and the -s/-u window
                    const uint64_t room = o.batch - b->r.size();
                    const uint64_t lim = rdid >= o.skip && rdid < o.upto ? std::min<uint64_t>({(uint64_t)(c1.size() - i1), room, o.upto - rdid}) : 0;
                    if (c1Named && lim > 1) {
                        b->r.appendRange(c1, i1, i1 + lim);
                        i1 += lim; rdid += lim;
                        continue;
                    }
                }
                if (paired && !fetch(*s2, c2, i2)) die("Error, fewer reads in file specified with -2 than in file specified with -1");
                if (paired && c1Named && c2Named && rdid >= o.skip && rdid < o.upto) {
                    // bulk path for mates: as many whole pairs as both chunks, the batch and the -u window allow
                    const uint64_t room = (o.batch * 2 - b->r.size()) / 2;
                    const uint64_t lim = std::min<uint64_t>({(uint64_t)(c1.size() - i1), (uint64_t)(c2.size() - i2), room, o.upto - rdid});
                    if (lim > 1) {
                        b->r.appendInterleaved(c1, i1, c2, i2, lim);
                        i1 += lim; i2 += lim; rdid += lim;
                        continue;
                    }
                }
                const uint64_t id = rdid++;
                if (id >= o.upto) { more = false; break; }
                if (id >= o.skip) {
                    take(*b, c1, i1, id);
                    if (paired) take(*b, c2, i2, id);
                }
                i1++; i2++;
            }
            if (o.ingestBench) {                     // count, and hand the batch's arrays back as the pipeline's last stage would
                benchReads += b->r.size(); benchBases += b->r.seq.size();
                std::lock_guard<std::mutex> lk(mu); spare.push_back(std::move(b));
                continue;
            }
            if (o.dumpReads) {
                if (dumpPacked) { dumpBatches++; if (b->r.pk.valid) dumpFromPacked++; }
                for (size_t i = 0; i < b->r.size(); i++) {
                    std::string ln(b->r.names.data() + b->r.nameOff[i], b->r.nameOff[i + 1] - b->r.nameOff[i]);
                    ln.push_back('\t');
                    if (dumpPacked && b->r.pk.valid) {            // the bases as the packed form holds them
                        const PackedSoA &pk = b->r.pk;
                        if (i == 0) { dumpWordAt = 0; dumpNAt = 0; }
                        const uint32_t L = pk.lens.p[i];
                        for (uint32_t j = 0; j < L; j++) {
                            const uint64_t wi = dumpWordAt + (j >> 5);
                            while (dumpNAt < pk.nN && pk.nIdx.p[dumpNAt] < wi) dumpNAt++;
                            const bool isN = dumpNAt < pk.nN && pk.nIdx.p[dumpNAt] == wi && ((pk.nMsk.p[dumpNAt] >> (j & 31)) & 1u);
                            ln.push_back(isN ? 'N' : "ACGT"[(pk.words.p[wi] >> (2 * (j & 31))) & 3]);
                        }
                        dumpWordAt += (L + 31) >> 5;
                        if (pk.seeds.p[i] != b->r.seeds[i] || pk.nReads != b->r.size()) die("internal error: packed form out of step");
                    } else appendSeq(ln, b->r, i);
                    ln.push_back('\t'); appendQual(ln, b->r, i);
                    ln += "\t" + std::to_string(b->r.seeds[i]) + "\n";
                    std::fwrite(ln.data(), 1, ln.size(), stdout);
                }
                continue;
            }
            lastSeq = std::max(lastSeq, b->r.seq.size()); lastNames = std::max(lastNames, b->r.names.size());
            lastReads = std::max(lastReads, b->r.size()); lastQual = lastQual || b->r.hasQual;
            const auto tp1 = std::chrono::steady_clock::now();
            R.tm.produce += std::chrono::duration<double>(tp1 - tp0).count();
            if (!submit(std::move(b))) { aborted = true; break; }
            R.tm.wait += std::chrono::duration<double>(std::chrono::steady_clock::now() - tp1).count();
        }
        if (dumpPacked) std::fprintf(stderr, "dumped from the packed form: %llu of %llu batches\n", (unsigned long long)dumpFromPacked, (unsigned long long)dumpBatches);
        if (o.separator && !o.dumpReads && !aborted) {          // marker behind the input's last batch (centrifuge.cpp:3128-3226)
            auto mk = std::make_unique<Batch>();
            mk->endOfInput = (int)fi;
            if (!submit(std::move(mk))) aborted = true;
        }
      }
    } catch (const std::exception &e) {
        { std::lock_guard<std::mutex> lk(mu); producerDone = true; if (workerError.empty()) workerError = e.what(); }
        cv.notify_all();
        joinAll();
        die(e.what());
    }
    { std::lock_guard<std::mutex> lk(mu); producerDone = true; }
    cv.notify_all();
    joinAll();
    if (!workerError.empty()) die(workerError);
    R.waitWrite();                                   // the last batch's text is in the file
    if (o.ingestBench) std::fprintf(stderr, "ingest: %llu reads, %llu bases in %.3f s\n", (unsigned long long)benchReads, (unsigned long long)benchBases, secs(ts));
    if (o.dumpReads) return 0;
    if (o.timing) {
        StageTimes g;
        for (const auto &t : R.gts) { g.create += t.tm.create; g.classify += t.tm.classify; g.results += t.tm.results; g.report += t.tm.report; }
        std::fprintf(stderr, "Multiseed full-index search: %s\n", hms(secs(ts)).c_str());
        if (R.textBatches) {
            uint64_t tb = 0, hb = 0;
            for (const auto &t : R.gts) { g.read += t.tm.read; g.parse += t.tm.parse; g.hostParse += t.tm.hostParse; g.write += t.tm.write; g.format += t.tm.format; tb += t.textBlocks; hb += t.hostBlocks; }
            std::fprintf(stderr, "Device text path: %llu block(s) parsed and printed on the device, %llu on the host (not in the plain form); %zu GPU thread(s), seconds summed over them: "
                                 "read %.2f, upload + parse %.2f, host parse %.2f, enqueue %.2f, kernels + format + download %.2f, tuples %.2f, host format %.2f, write %.2f\n",
                         (unsigned long long)tb, (unsigned long long)hb, R.gts.size(), g.read, g.parse, g.hostParse, g.create, g.classify, g.report, g.format, g.write);
        }
        std::fprintf(stderr, "Stage seconds: index open %.2f, search wall %.2f; %zu GPU thread(s) on %zu device(s): submit (upload + enqueue) %.2f, kernels + download %.2f, results %.2f, tally %.2f; "
                             "output thread: tally %.2f, format %.2f, waiting for the writer %.2f (writer thread: write %.2f); reader thread: assemble %.2f, waiting for the pipeline %.2f\n",
                     R.indexOpenS, secs(ts), R.gts.size(), R.devs.size(), g.create, g.classify, g.results, g.report, R.tm.report, R.tm.format, R.tm.write, R.writeBusy, R.tm.produce, R.tm.wait);
    }
    R.waitWrite();
    if (R.out != stdout) { std::FILE *f = R.out; R.out = stdout; if (std::fclose(f) != 0) die("error closing the classification output"); }
    else std::fflush(stdout);
    if (!o.separator && !o.reportFile.empty()) R.writeReport(R.finishReport(), o.reportFile, hms);   // one coalesced report (centrifuge.cpp:3231-3319)
    if (o.timing) {
        std::fprintf(stderr, "Overall time: %s\n", hms(secs(t0)).c_str());
        std::fprintf(stderr, "Overall seconds: %.2f (from the start of the call to here; of which before the index open %.2f)\n", secs(t0), R.beforeOpenS);
    }
    return 0;
}

}  // namespace

// The reference's C entry point (centrifuge.cpp:3338-3345, declared centrifuge_main.cpp:29-32): borrows
// argv, serially re-entrant, returns non-zero with a message on stderr; no exception or exit() escapes.
extern "C" int centrifuge(int argc, const char **argv) {
    try {
        return run(argc, argv);
    } catch (const CliExit &e) {
        if (e.what()[0]) std::fprintf(stderr, "%s\n", e.what());
        std::fflush(stdout);
        return e.code;
    } catch (const std::exception &e) {
        std::fprintf(stderr, "Error: Encountered exception: '%s'\n", e.what());
        return 1;
    } catch (...) {
        std::fprintf(stderr, "Error: Encountered an unknown exception\n");
        return 1;
    }
}
