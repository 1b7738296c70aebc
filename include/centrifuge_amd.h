/*
 * centrifuge_amd.h — C ABI of the MI355X-native Centrifuge classification path.
 *
 * The reference (DaehwanKimLab/centrifuge 1.0.4) has no plugin / FFI seam for
 * this path: its only C symbol is
 *     extern "C" int centrifuge(int argc, const char **argv)   centrifuge.cpp:3338-3345
 * and the only polymorphic seam is
 *     virtual int HI_Aligner::go(...)                          hi_aligner.h:791-802
 * overridden once by Classifier::go (classifier.h:212).  This header is the
 * boundary a maintainer binds instead of that operator: plain pointers and
 * sizes, no C++ or torch types.  Every entry point names the reference
 * interface it replaces.  INTEGRATION.md shows the reference-side stub.
 *
 * Conventions: the caller owns all host buffers; the library owns device
 * memory; no exception crosses the ABI (status codes + cf_strerror); results
 * of a batch come back in input order (equivalent to the reference's
 * --reorder, the only order in which it is deterministic across -p).
 * There is NO CPU fallback: without a HIP device every compute entry point
 * returns CF_ERR_NO_DEVICE.
 */
#ifndef CENTRIFUGE_AMD_H
#define CENTRIFUGE_AMD_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef int cf_status;
enum {
    CF_OK = 0,
    CF_ERR_IO = 1,          /* cannot open / short index file                  */
    CF_ERR_FORMAT = 2,      /* index file not understood                       */
    CF_ERR_NO_DEVICE = 3,   /* no HIP device: the product has no CPU path      */
    CF_ERR_HIP = 4,         /* a HIP runtime call failed (see cf_last_error)   */
    CF_ERR_ARG = 5,
    CF_ERR_NOMEM = 6
};
const char *cf_strerror(cf_status);
const char *cf_last_error(void);           /* thread-local detail of the last failure */

/* ---------------------------------------------------------------- index
 * Replaces Ebwt<uint64_t>(...) + Ebwt::loadIntoMemory
 * (centrifuge.cpp:2878,2950; bt2_idx.h:566-853; bt2_io.h:42-685).
 * cf_index_open_host parses <base>.{1,2,3,4}.cf into host memory only (no GPU
 * needed: taxonomy, names, uid table for output formatting);
 * cf_index_open additionally lays the BWT sides / ftab / SA sample / taxonomy
 * tables out in the HBM of `device`. */
typedef struct cf_index cf_index;
cf_status cf_index_open(const char *basename, int device, cf_index **out);
cf_status cf_index_open_host(const char *basename, cf_index **out);
void      cf_index_close(cf_index *);

/* The derived tables (made on the device from the files' content when the index is opened; they change no result) trade HBM
 * for random memory requests: wide ftab, text + SA / inverse-SA samples, occurrence planes, pair planes, resolve table.  WHICH
 * of them, and how dense, is decided by a planner: it enumerates the combinations, prices each with a model of what a read
 * then costs (fitted to measured op counts) and takes the cheapest that fits the room — with cf_index_open what
 * the device has free less the files and a reserve for the batch slots (a fifth of the device, at least 48 GB, never more than half of what is free), with cf_index_open_ex the
 * caller's budget for ALL the index may occupy (files' sections + tables), e.g. to share a GPU or to run many slots.  Fields
 * of cf_index_options: 0 = automatic, -1 = off, a value = that, as long as it fits.  For tests and experiments a set of
 * environment knobs (CF_WIDE_FTAB, CF_TEXT_VERIFY_RATE, CF_OCC_PLANES, CF_DENSE_SA_RATE, CF_PAIR_PLANES ...: the list is in
 * csrc/cf_knobs.hpp) overrides the fields — read ONLY while CF_DEBUG_KNOBS=1 is set as well, so that nothing in a user's
 * environment reaches them.  cf_index_describe reports what was made and what it costs. */
typedef struct {
    uint64_t hbm_budget_bytes;  /* 0 = whatever is free on the device                                                     */
    int32_t  wide_ftab_chars;   /* 0 = automatic (floor(log4 n), at most 16), -1 = none, else bases per entry (<= 16)      */
    int32_t  text_verify_rate;  /* 0 = automatic (densest that fits: 1 .. 5), -1 = none, else the rate                    */
    int32_t  occ_planes;        /* 0 = automatic, -1 = none, 1 = wanted                                                   */
    int32_t  resolve_rate;      /* 0 = automatic (every row if it fits, else every 2nd, 4th, 8th), -1 = the file's sample, */
                                /* else rate + 1 (1 = every row, 2 = every 2nd ...)                                       */
    int32_t  pair_planes;       /* 0 = automatic, -1 = none, 1 = wanted (needs the planes; 4 bytes per base)               */
    int32_t  sides;             /* the file's BWT sides in HBM once the tables are made: 0 = automatic (they leave when the planes  */
                                /* exist and their room buys a cheaper plan: the nt-scale index), 1 = keep, -1 = drop with planes  */
    int32_t  small_range_rows;  /* n (2 .. 15) = a search range of up to n rows that keeps its size — relatives: the strains of a      */
                                /* cluster — is finished against the text like a single row (SA of every row + text windows + one      */
                                /* inverse-SA read).  Needs the SA / inverse-SA samples at EVERY row (10.7 bytes per base instead of   */
                                /* 5.3) and the planes; the planner grants it when its model prices that plan below the usual one,     */
                                /* else the option has no effect.  0 = automatic: cf_index_open measures how repeat-rich the            */
                                /* collection is (the share of neighbouring suffix-array rows preceded by the same 24 bases,            */
                                /* cf_index_config::repeat_fraction) and asks for 4 rows from a share of 0.10; -1 = off.  Reads of     */
                                /* more than 256 bases go through the byte-window kernel, which steps such ranges (same results).  On  */
                                /* the 8.6 Gbp repeat-rich stand-in: 43.5 -> 26 requests per 100-base read, 6.4 -> 7.8e8 reads/s        */
    int32_t  reserved_;
    uint64_t expected_reads;    /* round 6: the size of the JOB the index is opened for, in reads; 0 = unknown / a long-running caller   */
                                /* (the tables that make a read cheapest, as before).  The derived tables take 6 - 18 s to make and pay  */
                                /* only over ~1e9 reads: with a job size the planner minimises build time + reads x time per read, both  */
                                /* from its model (the wide ftab costs 0.36 ns per entry, text and resolve tables 0.33 ns per base, a   */
                                /* read 0.022 ns per cost unit) — a 1 M-read job gets the planes and a 13-mer ftab and starts at once.   */
                                /* centrifuge-class estimates it from the sizes of its input files (--expected-reads overrides)          */
} cf_index_options;
typedef struct {
    uint64_t text_len;
    uint64_t budget_bytes;          /* the budget the tables were chosen under (the device's free memory when none was given)   */
    uint64_t file_section_bytes;    /* sides, ftab, SA sample, boundary rows, taxonomy                                          */
    uint64_t wide_ftab_bytes;  int32_t wide_ftab_chars;      /* 0 = not made                                                    */
    uint64_t text_bytes;       int32_t text_verify_rate;     /* 2-bit text + SA / inverse-SA samples; -1 = not made             */
    uint64_t planes_bytes;     int32_t occ_planes;
    uint64_t pair_planes_bytes; int32_t pair_planes;         /* two bases per LF request (CF_PAIR_PLANES)                       */
    uint64_t resolve_bytes;    int32_t resolve_rate;         /* rows are resolved at every 2^rate-th row (offRate = file's own) */
    int32_t  sides_dropped;         /* 1: the BWT sides left HBM after the tables were made (cf_index_restore then needs the text tables) */
    uint64_t total_bytes;
    double   build_ms;              /* all derived tables together                                                              */
    /* a model of the random requests one 100-base read costs in the search and walk kernels with this configuration (DESIGN.md
       5 has the measured curve): what a caller sizing a budget can expect, not a measurement */
    double   est_requests_per_100bp_read;
    uint64_t file_bytes_dropped;    /* of file_section_bytes: what left HBM once a derived table replaced it (the SA sample behind a  */
                                    /* denser resolve table; the sides when sides_dropped) — total_bytes no longer holds it            */
    int32_t  small_range_rows;      /* rows up to which a search range is finished against the text (0 = not in effect)               */
    int32_t  plan_realised;         /* 1 = every table was made as the planner chose it, 0 = one did not fit when its turn came (made  */
                                    /* coarser, or not at all), -1 = no plan (CF_TABLE_PLANNER=0, host-only view)                      */
    double   repeat_fraction;       /* share of neighbouring suffix-array rows whose suffixes are preceded by the same 24 bases, over   */
                                    /* 16 K sampled rows (-1 = not measured: small_range_rows was given): ~0 for an i.i.d.-like text   */
} cf_index_config;
cf_status cf_index_open_ex(const char *basename, int device, const cf_index_options *opt /* NULL = all automatic */, cf_index **out);
cf_status cf_index_describe(const cf_index *, cf_index_config *out);
/* the table planner on its own, no device needed (tests, capacity planning): what cf_index_open would make of an index of n
 * bases with `room` bytes for the tables -> out = {wide-ftab bases (0 = none), text rate (-1 = none), planes, resolve rate
 * (off_rate = the file's sample), pair planes, sides dropped}, the model's cost (L1 load x line pairs per 100-base read), the tables' bytes */
cf_status cf_debug_plan_tables(uint64_t n, int ftab_chars, int off_rate, int sa_width, uint64_t room, const cf_index_options *opt,
                               int32_t out[6], double *cost, uint64_t *bytes);

uint64_t    cf_index_text_len(const cf_index *);      /* EbwtParams::_len                */
uint64_t    cf_index_num_refs(const cf_index *);      /* |uid_to_tid|                    */
uint64_t    cf_index_num_taxa(const cf_index *);      /* size of the dense taxon table   */
uint64_t    cf_index_device_bytes(const cf_index *);  /* bytes resident in HBM           */
int         cf_index_compressed(const cf_index *);    /* bt2_idx.h:648-663               */
int         cf_index_sa_width(const cf_index *);      /* 2 or 4 bytes per SA sample      */
/* SA rows are resolved against a table of every 2^rate-th row: the file's own sample (rate = offRate, 4) or the denser
 * one cf_index_open derives from it on the device (CF_DENSE_SA_RATE, default 2: a quarter of the walk-left steps) */
int         cf_index_resolve_rate(const cf_index *);
/* bases per entry of the wide ftab cf_index_open derives on the device (CF_WIDE_FTAB; 0 = none: the file's 10-mer ftab only) */
int         cf_index_wide_ftab_chars(const cf_index *);
/* 1 when cf_index_open derived the occurrence planes (CF_OCC_PLANES; per 64 rows and character: 64 match bits + the LF base,
   8 bits per base): the search kernel then runs one chain per lane with one 16-byte load per LF step; 0: it reads the sides */
int         cf_index_occ_planes(const cf_index *);
double      cf_index_occ_planes_build_ms(const cf_index *);
/* unique matches are verified against the 2-bit text through SA / inverse-SA samples of every 2^rate-th row / position,
 * derived on the device when the index is opened (CF_TEXT_VERIFY_RATE, default 2; -1 = not built) */
int         cf_index_text_verify_rate(const cf_index *);
double      cf_index_text_verify_build_ms(const cf_index *);
double      cf_index_resolve_build_ms(const cf_index *);
/* round 6: the bound on the walk-left (bt2_idx.h:1980-2014: steps until the '$' row, a row of the file's sample or a boundary row) that
 * the position form of hits rests on — exact (the longest walk from ANY row) where the resolve table holds every row; 0 = no hit takes the form.
 * cf_index_resolve_by_position: 1 when that table was made from the stop rows' text positions (every row, SA[row] at every row) instead of by walks */
uint32_t    cf_index_walk_bound(const cf_index *);
int         cf_index_resolve_by_position(const cf_index *);
const char *cf_index_uid(const cf_index *, uint64_t ref);
uint64_t    cf_index_ref_taxid(const cf_index *, uint64_t ref);
uint64_t    cf_index_taxon_id(const cf_index *, uint64_t dense_idx);
/* seqID column text (classifier.h:546-557 + aln_sink.h:2219-2234) */
const char *cf_format_seqid(const cf_index *, uint32_t unique_id, uint64_t tax_id);
int         cf_tax_rank(const cf_index *, uint64_t tax_id);     /* taxonomy.h:15-47 code, 0 if absent */
const char *cf_tax_rank_string(int rank);                       /* taxonomy.h:207-239 */
const char *cf_tax_name(const cf_index *, uint64_t tax_id);     /* "" if none */
uint64_t    cf_tax_size(const cf_index *, uint64_t tax_id);     /* 0 if none  */

/* The joined text back out of the BWT — replaces Ebwt::restore (bt2_util.h:150-168), the
 * engine of centrifuge-inspect's FASTA mode (centrifuge_inspect.cpp:369-430).  Needs an index
 * opened on a device.  packed: cf_index_text_len/4 + 1 bytes, 2 bits per character, character i
 * at bits 2(i%4) of byte i/4 (codes 0..3 = ACGT), i.e. the BWT's own packing. */
cf_status   cf_index_restore(cf_index *, uint8_t *packed, uint64_t n_bytes);

/* ----------------------------------------------------------- classifier
 * Replaces the per-thread Classifier(...) + ReportingParams(khits, compressed)
 * (centrifuge.cpp:2365-2374; classifier.h:135-202; aln_sink.h:570-588). */
typedef struct {
    int32_t khits;            /* -k                       default 5  centrifuge.cpp:321 */
    int32_t min_hitlen;       /* --min-hitlen (>= 15)     default 22 centrifuge.cpp:473 */
    int32_t rank_slot;        /* --classification-rank as path slot: 0 strain, 1 species,
                                 2 genus, 3 family, 4 order, 5 class, 6 phylum           */
    int32_t tree_traverse;    /* 0 = --no-traverse                                       */
    const uint64_t *host_taxids;    int32_t n_host;      /* --host-taxids    */
    const uint64_t *exclude_taxids; int32_t n_exclude;   /* --exclude-taxids */
} cf_params;
cf_status cf_params_default(cf_params *);

typedef struct cf_classifier cf_classifier;
cf_status cf_classifier_create(cf_index *, const cf_params *, cf_classifier **out);
void      cf_classifier_destroy(cf_classifier *);

/* ---------------------------------------------------------------- batch
 * Replaces PatternSourcePerThread::nextReadPair + initRead/initReads for a
 * whole batch (centrifuge.cpp:2447,2678-2690; hi_aligner.h:739-785), and the
 * way the reference overlaps parsing, search and output across its threads
 * (centrifuge.cpp:2342-2755, outq.cpp:51-100): a cf_batch is a reusable SLOT
 * — device workspace plus pinned result buffers that only ever grow — and a
 * batch moves through it as three asynchronous stages on a caller stream:
 *
 *     cf_batch_upload_packed_async -> cf_classify_async -> cf_batch_download_async      (cf_batch_submit = all three)
 *     ... the caller fills / submits other slots on other streams ...
 *     cf_batch_wait                                                                      (the only call that blocks)
 *
 * Between "reads in" and "results out" the device never waits for the host:
 * the sizes one kernel makes for the next stay on the device.  With two or
 * three slots in flight, the upload of batch i+1, the kernels of batch i and
 * the download of batch i-1 overlap.
 *
 * Reads cross the boundary PACKED ("2-bit bases + N mask"):
 * read r owns the 32-base words [w(r), w(r+1)), w(r) = sum over the reads
 * before it of ceil(len/32); base i of the read sits in bits 2(i%32)..+1 of
 * bases[w(r) + i/32] (codes 0..3 = A,C,G,T; an N carries code 0) and, when it
 * is an N, in bit i%32 of nmask[w(r) + i/32].  3/8 byte per base instead of 1.
 * paired != 0: reads 2q and 2q+1 are the mates of query q.  seeds[r] =
 * genRandSeed of read r (cf_gen_rand_seed).  The N and length filters
 * (scoring.cpp:104-117, centrifuge.cpp:2550-2577) are applied on the device. */
typedef struct cf_batch cf_batch;
typedef struct {
    const uint64_t *bases;     /* n_words packed 2-bit words                                        */
    const uint32_t *nmask;     /* n_words N-mask words                                              */
    const uint32_t *len;       /* n_reads read lengths                                              */
    const uint32_t *seeds;     /* n_reads per-read seeds                                            */
    uint64_t n_reads, n_words; /* n_words >= sum of ceil(len/32): checked on the device, cf_batch_wait fails otherwise */
    uint64_t n_bases;          /* sum of len, or 0 if not known (sizes the hit pool more tightly)   */
    uint32_t max_len;          /* >= every len (a longer read makes cf_batch_wait fail); <= 16,777,213: */
                               /* hit records keep 24-bit offsets, longer reads are CF_ERR_ARG      */
    int32_t  paired;
    /* Sparse form of the N mask, used when nmask == NULL: only the words that hold an N (most batches have a handful;
       the dense mask is 2/7 of the bytes that cross PCIe).  nword_idx[i] = index of a word, nword_mask[i] = its mask
       word; every other word's mask is 0.  n_nwords == 0: no N in the batch. */
    const uint64_t *nword_idx;
    const uint32_t *nword_mask;
    uint64_t n_nwords;
} cf_packed_reads;
typedef struct {
    uint64_t tax_id;
    uint32_t unique_id;       /* reference-sequence index or CF_MERGED */
    uint32_t score;
    uint32_t hit_len;         /* hitLength column */
    uint32_t taxon_idx;       /* dense index of tax_id (cf_index_taxon_id) */
} cf_row;
typedef struct {              /* results of a batch, in the slot's pinned host memory: valid until the slot's next upload */
    const cf_row *rows;       /* printed rows of all queries back to back, query order               */
    const uint32_t *n_rows;   /* per query; 0 = the single "unclassified" row                        */
    const uint32_t *score2;   /* per query: 2ndBestScore column                                      */
    const uint32_t *max_score;/* per query: classifier.h:530-536 (perfect-hit test of the abundance EM) */
    uint64_t n_queries, total_rows;
    uint64_t planned_sa_rows; /* SA rows the batch resolved                                          */
    uint32_t row_passes;      /* passes of the row stage (1 unless the rows exceeded the workspace)  */
    uint32_t slow_post;       /* diagnostics: queries the common-case post / score kernels left to   */
    uint32_t slow_score;      /* the general ones (score: in the last pass of the row stage)         */
} cf_results;

/* ---- the NARROW forms of both directions (round 5): what crosses the host link when the kernels outrun it.  A 100-base read
 * in the word form above costs 40 bytes in (four words, its length, its seed) and 36 out (a 24-byte row, three words per query);
 * dense in and narrow out it costs 29 + 21, and the classification of a batch is the same to the bit (tests/test_async_abi.py).
 *
 * Dense reads: every read of the batch has ONE length (what a sequencer run delivers untrimmed); four bases per byte — base i
 * of read r in bits 2(i%4)..+1 of bases4[r * ceil(read_len/4) + i/4], codes as above — every read starting on a byte, no
 * length array; the N mask in its sparse form, indexed by the WORD the device will hold the base in (read r owns words
 * [r * ceil(read_len/32), ...): word index r * ceil(read_len/32) + i/32, bit i%32).  Unpacked into the word form on the device. */
typedef struct {
    const uint8_t  *bases4;    /* n_reads * ceil(read_len / 4) bytes                                  */
    const uint32_t *seeds;     /* n_reads per-read seeds                                              */
    uint64_t n_reads;
    uint32_t read_len;         /* <= 16,777,213                                                       */
    int32_t  paired;
    const uint64_t *nword_idx; /* sparse N mask as in cf_packed_reads; n_nwords == 0: no N            */
    const uint32_t *nword_mask;
    uint64_t n_nwords;
} cf_dense_reads;
cf_status cf_batch_upload_dense_async(cf_batch *, const cf_dense_reads *, void *hip_stream);

/* Narrow results: 16-byte rows (tax_id = cf_index_taxon_id(taxon_idx)) and, per query, ONE byte — rows printed (bits 0-5) |
 * first mate took part in the classification (bit 6) | second mate did (bit 7) — and 2ndBestScore; max_score is a function of
 * the mates that took part and their lengths (cf_narrow_max_score).  Needs -k <= 63.  The format is a property of the slot:
 * set it before cf_classify_async; cf_batch_wait_narrow then replaces cf_batch_wait (which refuses a narrow slot). */
typedef struct { uint32_t unique_id, taxon_idx, score, hit_len; } cf_row16;
typedef struct {
    const cf_row16 *rows;      /* printed rows of all queries back to back, query order               */
    const uint8_t  *qinfo;     /* per query: n_rows | mate-1 passed << 6 | mate-2 passed << 7         */
    const uint32_t *score2;
    uint64_t n_queries, total_rows, planned_sa_rows;
    uint32_t row_passes, slow_post, slow_score;
} cf_results_narrow;
#define CF_RESULTS_ROWS   0    /* cf_row + n_rows + score2 + max_score (the default)                  */
#define CF_RESULTS_NARROW 1
cf_status cf_batch_set_result_format(cf_batch *, int format);
cf_status cf_batch_wait_narrow(cf_batch *, cf_results_narrow *out);
/* classifier.h:530-536 from a query's qinfo byte and the lengths of its mates (len2 ignored unless paired) */
uint32_t  cf_narrow_max_score(uint8_t qinfo, uint32_t len1, uint32_t len2, int paired);
/* the narrow results of a batch as cf_row / n_rows / max_score arrays (rows: total_rows entries; the others n_queries): for
 * callers that want the wide form after the link has been crossed.  len: the batch's read lengths (n_reads), or NULL with
 * uniform_len for a dense batch */
cf_status cf_results_narrow_expand(const cf_index *, const cf_results_narrow *, const uint32_t *len, uint32_t uniform_len, int paired,
                                   cf_row *rows, uint32_t *n_rows, uint32_t *max_score);

/* ---- the TEXT forms of both directions (round 6): ingest and egress on the device, for a front end whose host cores cannot parse
 * and print as fast as the kernels classify (reference: FastaPatternSource::read pat.cpp:725-850, FastqPatternSource::read
 * pat.cpp:852-1100, genRandSeed pat.h:55-91 in; AlnSinkSam::appendMate aln_sink.h:2279-2337, the readID rule aln_sink.h:2203-2217
 * and SpeciesMetrics::addSpeciesCounts aln_sink.h:142-172 out).
 *
 * In: a block of WHOLE records as the file holds them (for mates: the two files' blocks that hold the same records).  The device finds the records, makes lengths, seeds and packed
 * words — if every record of the block has the plain form (FASTA: '>' + non-empty name line without '\r', then lines of
 * A C G T N in either case, at least one base; any '>' starts a record.  FASTQ: four lines per record — '@' + name, bases, a line
 * starting with '+', as many quality characters >= 33 as bases —, the block ending with a '\n').  That is the form in which the
 * reference's parsers and this one cannot differ; a block with any other record is NOT parsed: info->irregular names the reason,
 * the slot holds no batch, and the caller parses that block on the host (cf_batch_upload_packed_async).  The call waits for the
 * parse (one status word crosses the link): the number of reads sizes the slot. */
#define CF_TEXT_FASTA 0
#define CF_TEXT_FASTQ 1
typedef struct {
    const char *text;          /* n_bytes of the file, from a record start to a record end; pinned memory makes the copy a DMA */
    uint64_t n_bytes;          /* < 2^32 - 65536                                                       */
    int32_t  format;           /* CF_TEXT_FASTA | CF_TEXT_FASTQ                                        */
    uint32_t global_seed;      /* --seed (cf_gen_rand_seed's last argument)                            */
    uint64_t max_reads;        /* 0 = every record; else only the block's first max_reads records (pairs, with text2) (-u) */
    const char *text2;         /* mates: the block of the second file that holds the SAME NUMBER of records (any other number */
    uint64_t n_bytes2;         /* is `irregular`); NULL = unpaired.  Reads 2q and 2q+1 of the batch are the mates of query q   */
} cf_text_reads;
typedef struct {
    uint64_t n_reads, n_bases;
    uint32_t max_len;
    uint32_t irregular;        /* 0 = the slot holds the block's reads; else why the block is not in the plain form (a bit set) */
} cf_text_info;
cf_status cf_batch_upload_text(cf_batch *, const cf_text_reads *, void *hip_stream, cf_text_info *info);
/* Out: the batch's rows as the text centrifuge prints by default — readID seqID taxID score 2ndBestScore hitLength queryLength
 * numMatches, one line per row, query order — formatted on the device from the rows the kernels left there (no row crosses the
 * link) with the readIDs copied out of the uploaded block; in the slot's pinned memory, valid until its next upload.  Needs the
 * slot in the narrow result format and a batch that came through cf_batch_upload_text; replaces cf_batch_wait.  The same pass
 * tallies what the report needs beyond the device's per-taxon counters (cf_counts_get): the perfect single assignments per taxon
 * (cf_counts_get_single) and the perfect multi-assignment tuples — `tuples`: n, then n dense taxon indices, ... — for
 * cf_report_add_tuples. */
typedef struct {
    const char *text;
    uint64_t n_bytes;
    const uint32_t *tuples;
    uint64_t n_tuple_words;
    uint64_t n_queries, total_rows, planned_sa_rows;
    uint32_t row_passes, slow_post, slow_score;
} cf_results_text;
cf_status cf_batch_wait_text(cf_batch *, cf_results_text *out);

/* pinned (page-locked) host memory: what makes the transfers of the async calls truly asynchronous */
cf_status cf_host_alloc(void **p, size_t bytes);
void      cf_host_free(void *p);

/* a slot for batches of up to about max_reads reads in max_words packed words (hints: a larger batch grows it) */
cf_status cf_batch_alloc(cf_classifier *, uint64_t max_reads, uint64_t max_words, cf_batch **out);
/* device bytes such a slot takes (no device needed): a caller that knows its slots sizes cf_index_options::hbm_budget_bytes
 * as free memory - slots instead of leaving cf_index_open its default reserve of a fifth of the device */
cf_status cf_slot_estimate_bytes(uint64_t max_reads, uint64_t max_words, int khits, int ftab_chars, int occ_planes, uint64_t *bytes);
cf_status cf_batch_upload_packed_async(cf_batch *, const cf_packed_reads *, void *hip_stream);
cf_status cf_classify_async(cf_classifier *, cf_batch *, void *hip_stream);
/* plan + kernels once more over the reads the slot already holds (no upload): for reads that are produced on the device or stay
 * there — and what bench.py times ("inputs resident in HBM") */
cf_status cf_batch_reclassify_async(cf_classifier *, cf_batch *, void *hip_stream);
cf_status cf_batch_download_async(cf_batch *, void *hip_stream);
cf_status cf_batch_submit(cf_batch *, const cf_packed_reads *, void *hip_stream);
cf_status cf_batch_wait(cf_batch *, cf_results *out /* may be NULL */);
/* the same slot fed with 1 byte per base (codes 0..4 = A,C,G,T,N, alphabet.cpp:298-319; read r = seq[off[r], off[r+1])):
 * staged to the device and packed there */
cf_status cf_batch_upload(cf_batch *, const uint8_t *seq, const uint64_t *off, const uint32_t *seeds, uint64_t n_reads,
                          int paired, void *hip_stream);
/* test knob: cap the hit pool (slots) and the row workspace (rows per pass) of a slot, 0 = no cap — drives the
 * re-run / multi-pass paths of cf_batch_wait on small inputs */
cf_status cf_batch_set_limits(cf_batch *, uint64_t hit_slots, uint64_t rows_per_pass);

/* One-shot form: a slot sized for exactly these reads (1 byte per base), uploaded, packed and planned before
 * the call returns; classify it with cf_classify. */
cf_status cf_batch_create(cf_classifier *, const uint8_t *seq, const uint64_t *off,
                          const uint32_t *seeds, uint64_t n_reads, int paired, cf_batch **out);
void      cf_batch_destroy(cf_batch *);
uint64_t  cf_batch_num_queries(const cf_batch *);

/* pat.h:55-91 genRandSeed.  qual may be NULL (FASTA: all 'I', pat.cpp:828). */
uint32_t cf_gen_rand_seed(const uint8_t *seq, const uint8_t *qual, uint64_t len,
                          const char *name, uint64_t name_len, uint32_t global_seed);

/* ------------------------------------------------------------- classify
 * Replaces Classifier::go + AlnSinkWrap::finishRead for every query of the
 * batch (centrifuge.cpp:2693,2715; classifier.h:212-571; aln_sink.h:1633-1927),
 * including the per-taxon counters of SpeciesMetrics::addSpeciesCounts
 * (aln_sink.h:142-172), which accumulate in the classifier until reset.
 * Runs the HIP kernels on `stream` (a hipStream_t, or NULL for the default
 * stream) and returns after they complete: cf_classify_async + cf_batch_download_async + cf_batch_wait. */
cf_status cf_classify(cf_classifier *, cf_batch *, void *stream);

#define CF_MERGED 0xffffffffu   /* unique_id of an assignment merged up the taxonomy */

/* The device-side preparation of a batch once more, from its resident reads (cf_batch_create has done it
 * once): the plan — N filter and length filter of centrifuge.cpp:2550-2577, hit capacities, work list — and the
 * strand records.  For callers that count this stage into a measured pass (bench.py does) or re-use resident
 * reads; cf_batch_plan_ms gives its device time (HIP events on the stream it ran on). */
cf_status cf_batch_plan(cf_batch *, void *hip_stream);
cf_status cf_batch_plan_ms(const cf_batch *, float *ms);

/* Copy results to the host: rows[q*khits + i] for i < n_rows[q], already in
 * print order; n_rows[q] == 0 means the single "unclassified" row;
 * score2[q] = 2ndBestScore column. */
cf_status cf_batch_results(cf_batch *, cf_row *rows, uint32_t *n_rows, uint32_t *score2);

/* The same rows packed back to back in query order (query q owns the n_rows[q] rows after those of
 * the queries before it): what the sink prints is 1-2 rows per read, so this moves a fraction of
 * the k-slot layout over PCIe and the caller sizes its buffer by cf_batch_num_rows instead of
 * n_queries * khits.  rows_cap = capacity of `rows` in rows (>= cf_batch_num_rows). */
cf_status cf_batch_num_rows(cf_batch *, uint64_t *total_rows);
cf_status cf_batch_results_compact(cf_batch *, cf_row *rows, uint64_t rows_cap, uint32_t *n_rows, uint32_t *score2);

/* max_score of every query (classifier.h:530-536): sum over the mates that passed the
 * filters of (len-15)^2; a printed row with score >= max_score is a perfect hit and
 * feeds the abundance EM (aln_sink.h:158-171).  Computed on the device with the batch plan.  The reference keeps it in an
 * int64_t and compares it with 32-bit scores: a value of 2^32 or more (a read beyond 65,550 bases) comes back as 0xffffffff,
 * "never reached" — no sum of two squares — and cf_report_add treats it so. */
cf_status cf_batch_max_scores(const cf_batch *, uint32_t *max_score);

/* Per-kernel device time of the last cf_classify on this batch, from HIP
 * events recorded on the launch stream: ms[0] search, [1] post/sort/plan,
 * [2] SA walk (resolve), [3] score/reduce/select, [4] whole call. */
cf_status cf_batch_timings(const cf_batch *, float ms[5]);
/* Work done by the last cf_classify: LF steps of the search kernel, of which
 * two-sided; ftab lookups; walk steps; rows.  The production kernels carry no
 * counters; the first call after a cf_classify re-runs the search and walk
 * kernels of the batch in their instrumented builds (same work, deterministic). */
typedef struct {
    uint64_t n_ftab, n_pair, n_pair2, n_single, n_walk, n_rows;
    uint64_t n_ftab_wide;     /* partialSearch calls started from the wide ftab (one 8-byte read) */
    uint64_t n_verify;        /* unique matches handed to the text comparison: one SA-sample read + one inverse-sample read each */
    uint64_t n_text_loads;    /* 32-byte text windows they compared */
    uint64_t n_pos_hits;      /* of n_verify: matches whose hit went out in its position form — the inverse-sample read was not made (round 6) */
} cf_opcounts;
cf_status cf_batch_opcounts(cf_batch *, cf_opcounts *);

/* ------------------------------------------------------------- counters
 * Dense per-taxon {n_reads, n_unique_reads} (ReadCounts aln_sink.h:45-51),
 * length cf_index_num_taxa each, accumulated on the device by cf_classify.
 * cf_counts_device exposes the device buffer (2*num_taxa u64: n_reads then
 * n_unique) so a caller can all-reduce it in place with RCCL across the
 * per-GPU processes of a node (the only collective of the path). */
cf_status cf_counts_reset(cf_classifier *);
cf_status cf_counts_get(cf_classifier *, uint64_t *n_reads, uint64_t *n_unique);
/* a third block of num_taxa u64 behind those two: the perfect single assignments per taxon (unclassified reads under taxon 0)
 * of the batches whose rows were formatted on the device (cf_batch_wait_text), summed by the all-reduce along with the others */
cf_status cf_counts_get_single(cf_classifier *, uint64_t *n_single);
void     *cf_counts_device(cf_classifier *);
/* The path's only collective: in-place sum of the counters over the ranks of
 * an RCCL communicator — ncclAllReduce(counts, counts, 2*num_taxa, ncclUint64, ncclSum, comm,
 * stream), enqueued on `stream` (hipStream_t or NULL).  `nccl_comm` is an ncclComm_t the caller
 * created (ncclCommInitRank, one process per GPU).  librccl is bound at first use (dlopen), so
 * the library itself carries no link-time RCCL dependency.  Replaces the mutexed
 * SpeciesMetrics::merge across threads (aln_sink.h:109-140). */
cf_status cf_counts_allreduce(cf_classifier *, void *nccl_comm, void *stream);

/* One process driving several GPUs (centrifuge-class --gpus N; the reference starts its workers from C++,
 * centrifuge.cpp:2806-2813): the communicators of all devices at once (ncclCommInitAll), and the all-reduce of every
 * device's counters as ONE RCCL group (ncclGroupStart / ncclAllReduce x n / ncclGroupEnd), synchronised on return. */
cf_status cf_comm_init_all(int n, const int *devices, void **comms /* n entries out */);
void      cf_comm_destroy(void *comm);
cf_status cf_counts_allreduce_group(cf_classifier *const *classifiers, void *const *comms, int n);
/* HIP streams / device count for callers that do not link the HIP runtime themselves */
cf_status cf_stream_create(int device, void **hip_stream);
void      cf_stream_destroy(void *hip_stream);
int       cf_device_count(void);
/* NUMA placement of the host side of a device (round 6; the reference's workers are plain threads, centrifuge.cpp:2806-2813, and
 * leave placement to the kernel): the node the GPU's PCIe link hangs off (/sys/bus/pci/devices/<bdf>/numa_node; -1 = unknown),
 * and the calling THREAD bound to that node's CPUs (within the affinity it already has) — pinned buffers are placed by first
 * touch, and a GPU thread or index loader on the far node copies at ~60 % of the local rate.  node_out may be NULL;
 * CF_OK with *node_out = -1 when the topology is not to be had (nothing changed). */
int       cf_device_numa_node(int device);
cf_status cf_thread_bind_near_device(int device, int *node_out);

/* ---------------------------------------------------------------- report
 * Replaces SpeciesMetrics (aln_sink.h:56-507) on the host side of a run and the
 * report writer of centrifuge.cpp:3231-3319: per-taxon numReads / numUniqueReads,
 * the `observed` multiset of perfect-hit taxID tuples, the SQUAREM-EM abundance
 * (aln_sink.h:274-495) and the report TSV.  Needs only a host view of the index
 * (cf_index_open_host is enough).  cf_report_add takes the rows of
 * cf_batch_results (khits = the classifier's k) or of cf_batch_results_compact
 * (khits = 0: packed rows) plus cf_batch_max_scores; cf_report_add_counts adds dense
 * counters instead (e.g. the RCCL-reduced cf_counts_get of other ranks). */
typedef struct cf_report cf_report;
cf_status cf_report_create(const cf_index *, cf_report **out);
void      cf_report_destroy(cf_report *);
cf_status cf_report_add(cf_report *, const cf_row *rows, const uint32_t *n_rows, const uint32_t *max_score,
                        uint64_t n_queries, uint32_t khits);
/* the same from the NARROW results of a batch (cf_results_narrow: 16-byte rows, one byte per query) without widening them first:
 * len = the batch's read lengths (n_reads), or NULL with uniform_len; max_score as cf_narrow_max_score gives it */
cf_status cf_report_add_narrow(cf_report *, const cf_row16 *rows, const uint8_t *qinfo, const uint32_t *len, uint32_t uniform_len,
                               int paired, uint64_t n_queries);
cf_status cf_report_add_counts(cf_report *, const uint64_t *taxids, const uint64_t *n_reads, const uint64_t *n_unique, uint64_t n);
/* metrics.reset() between the inputs of a --separator run (centrifuge.cpp:3225; SpeciesMetrics::reset,
 * aln_sink.h:84-91): the counters start over, the observed-tuple table is kept, as in the reference. */
cf_status cf_report_reset_counts(cf_report *);
/* The per-taxon counters of the report against dense device counters (cf_counts_get, after the all-reduce over the
 * GPUs of a run): returns CF_OK when every taxon agrees — the rows the host saw and the counters the kernels kept are
 * two tallies of the same reads — and makes the device counters the report's own. */
cf_status cf_report_adopt_counts(cf_report *, const uint64_t *n_reads, const uint64_t *n_unique, uint64_t n_taxa);
/* The tally of a run whose rows (all, or some of its batches') never reached the host (cf_batch_wait_text): the tuples of those
 * batches as they come, and at the end the device's counters — which cover every batch of the run — as the report's own, with
 * the perfect single assignments of cf_counts_get_single added to the observed tuples.  (No cross-check here: the host saw no rows
 * to check against.) */
cf_status cf_report_add_tuples(cf_report *, const uint32_t *tuples, uint64_t n_words);
cf_status cf_report_adopt_device_tally(cf_report *, const uint64_t *n_reads, const uint64_t *n_unique, const uint64_t *n_single, uint64_t n_taxa);
/* Ship a report between the per-GPU processes of a node (SpeciesMetrics::merge,
 * aln_sink.h:109-140): serialize into `cap_words` u64 words (call with buf = NULL to
 * size it), merge adds a serialized report into this one. */
cf_status cf_report_serialize(cf_report *, uint64_t *buf, uint64_t cap_words, uint64_t *need_words);
cf_status cf_report_merge(cf_report *, const uint64_t *buf, uint64_t n_words);
/* abundance != 0 runs the EM (--no-abundance turns it off); the two outputs are the
 * numbers the reference prints on stderr (aln_sink.h:471-472), either may be NULL */
cf_status cf_report_write(cf_report *, const char *path, int abundance, uint64_t *em_iterations, double *em_diff);

/* ------------------------------------------------- debug / parity taps */
typedef struct { uint64_t top, bot; uint32_t bwoff, len; } cf_hit;
/* Hit lists of read r after the search kernel and after the post kernel's
 * extend/twin/trim (classifier.h:646-896), before sorting.  max_hits is the
 * capacity of each output array; counts come back in nhits[2]. */
cf_status cf_debug_search(cf_classifier *, const uint8_t *seq, uint64_t len,
                          cf_hit *hits_fw, cf_hit *hits_rc, uint32_t max_hits, uint32_t nhits[2]);
/* rows -> reference-sequence index (group_walk.h:1154 / bt2_idx.h:1980) */
cf_status cf_debug_resolve(cf_index *, const uint64_t *rows, uint64_t n, uint32_t *refs);
/* out[i] = LF(rows[i], chars[i])  (bt2_idx.h:2192-2227) */
cf_status cf_debug_rank(cf_index *, const uint8_t *chars, const uint64_t *rows, uint64_t n, uint64_t *out);
/* same through the single-lane rank used by the post kernel's re-search */
cf_status cf_debug_rank1(cf_index *, const uint8_t *chars, const uint64_t *rows, uint64_t n, uint64_t *out);
/* the prefix sums of a batch on their own: n u32 items -> n + 1 exclusive sums of ceil(x/32) (mode 0), x (mode 1) or 2x
 * plus the exclusive counts of non-zero items (mode 2) */
cf_status cf_debug_scan(int device, int mode, const uint32_t *in, uint64_t n, uint64_t *sums, uint32_t *counts);
/* random 128-byte-side read bandwidth of the device over the resident sides
 * (the roofline denominator of SURVEY.md §8d): GB/s over `n_loads` loads. */
cf_status cf_debug_random_read_gbps(cf_index *, uint64_t n_loads, int dependent_steps, double *gbps);

/* ------------------------------------------------------------ program entry
 * The whole classifier program as a call: the reference's own C symbol
 * `extern "C" int centrifuge(int argc, const char **argv)` (centrifuge.cpp:3338-3345, declared
 * centrifuge_main.cpp:29-32; SURVEY.md §8b boundary 2).  argv as for centrifuge-class; borrows
 * argv; serially re-entrant (no option state survives a call); returns non-zero with a message
 * on stderr; never calls exit() and lets no C++ exception out. */
int centrifuge(int argc, const char **argv);

#ifdef __cplusplus
}
#endif
#endif
