/*
 * centrifuge_amd_build.h — C ABI of the MI355X-native index builder.
 *
 * Replaces the reference's `centrifuge-build-bin` for the files the
 * classification path consumes:
 *     driver<>()                     centrifuge_build.cpp:398-520
 *     Ebwt<index_t>::initFromVector  bt2_idx.h:1249-1642   (.1/.2/.3.cf, names)
 *     Ebwt<index_t>::buildToDisk     bt2_idx.h:3377-3840   (sides, ftab, SA sample, .4.cf)
 * The suffix array is built on the GPU (chunked radix sort of 29-mer keys with
 * tie refinement); the output files are byte-identical to the reference
 * builder's for the same inputs (tests/test_build_*.py).  There is no CPU path:
 * without a HIP device cf_build_index returns CF_ERR_NO_DEVICE.
 */
#ifndef CENTRIFUGE_AMD_BUILD_H
#define CENTRIFUGE_AMD_BUILD_H
#include <stddef.h>
#include <stdint.h>
#include "centrifuge_amd.h"
#ifdef __cplusplus
extern "C" {
#endif

typedef struct {
    /* ---- reference sequences: either FASTA files ... (centrifuge-build <ref.fa,...>) */
    const char *const *fasta_paths;   /* n_fasta paths, or NULL                         */
    int32_t n_fasta;
    /* ---- ... or sequences already in memory (synthetic data, no FASTA round trip):
     * codes[seq_off[i] .. seq_off[i+1]) are the bases of sequence i, 0..3 = A,C,G,T,
     * anything else = ambiguous (a gap, as asc2dnacat >= 2, alphabet.cpp:36-58);
     * seq_names[i] is its FASTA header line without '>'.                             */
    const uint8_t *codes;
    const uint64_t *seq_off;
    const char *const *seq_names;
    uint64_t n_seq;
    /* ---- taxonomy inputs, same files as the reference's options                     */
    const char *conversion_table;     /* --conversion-table  (uid <tab> taxid)          */
    const char *taxonomy_tree;        /* --taxonomy-tree     (nodes.dmp)                */
    const char *name_table;           /* --name-table        (names.dmp), may be NULL   */
    const char *size_table;           /* --size-table, may be NULL                      */
    int32_t off_rate;                 /* -o, default 4   (centrifuge_build.cpp:95)      */
    int32_t ftab_chars;               /* -t, default 10  (centrifuge_build.cpp:96)      */
    uint64_t chunk_suffixes;          /* suffixes sorted per GPU pass; 0 = default      */
    int32_t verbose;
} cf_build_input;

cf_status cf_build_input_default(cf_build_input *);

/* Writes <out_base>.1.cf .2.cf .3.cf .4.cf using HIP device `device`. */
cf_status cf_build_index(const cf_build_input *in, const char *out_base, int device);

/* Seconds spent in the phases of the last cf_build_index on this thread:
 * [0] parse/join, [1] GPU suffix sort + BWT/SA-sample emission, [2] file writing, [3] total */
cf_status cf_build_timings(double sec[4]);
const char *cf_build_last_error(void);      /* thread-local detail of the last cf_build_index failure */

/* The host-only half of a build: sequence bookkeeping + <out_base>.3.cf (uid table, pruned taxonomy, names,
 * sizes: bt2_idx.h:1375-1504).  Needs no device.  err (optional) receives the message of a failure. */
cf_status cf_build_taxonomy(const cf_build_input *in, const char *out_base, char *err, uint64_t err_cap);

/* The builder's view of its input, for tests (no device): u64 len, u64 n_seq, plen[], u64 n_frag, rstarts[3 n_frag],
 * the names ('\n' after each, then '\0'), the joined text (len codes 0..3) — what the header and name section of
 * <base>.1.cf are written from (bt2_io.h:854-880, bt2_idx.h:3262-3290, bt2_io.h:989-1027). */
cf_status cf_build_describe(const cf_build_input *in, const char *path, char *err, uint64_t err_cap);

/* The whole builder program as a call: the reference's own C symbol (centrifuge_build.cpp:550-556,
 * declared centrifuge_build_main.cpp:30-32).  argv as for centrifuge-build-bin; borrows argv;
 * returns non-zero with a message on stderr, never exits or throws. */
int centrifuge_build(int argc, const char **argv);

#ifdef __cplusplus
}
#endif
#endif
