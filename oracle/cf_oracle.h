/*
 * cf_oracle.h — CPU restatement of Centrifuge's per-read classification path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under centrifuge_amd/ (the product) may
 * include, link or call this.  Only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg use it, and only as the checker.
 *
 * Parity status: PINNED.  tests/test_oracle_golden.py checks this restatement
 * byte-for-byte against the TSVs the compiled, unmodified reference
 * (oracle/_ref/centrifuge-class, built by oracle/Makefile from the sources
 * under /root/reference) produced for the reference's own worked example
 * (example/ + MANUAL:1012-1028) and for generated indexes/reads — the truth
 * tables committed under tests/golden/ (made by tests/golden/make_golden.py) —
 * and tests/test_variants.py checks it against the reference binary run on the
 * spot (compressed index, u32 SA sample, -o / -t variants) wherever
 * oracle/_ref is present.
 *
 * Every function cites the reference file:line whose behaviour it restates
 * (paths relative to /root/reference).
 */
#ifndef CF_ORACLE_H
#define CF_ORACLE_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct cfo_index cfo_index;

typedef struct {
    int32_t  khits;          /* -k, default 5            centrifuge.cpp:321  */
    int32_t  min_hitlen;     /* --min-hitlen, default 22 centrifuge.cpp:473  */
    int32_t  rank_slot;      /* path slot of --classification-rank: 0 strain,
                                1 species ... 6 phylum   taxonomy.h:66-93    */
    int32_t  tree_traverse;  /* 0 with --no-traverse                         */
    const uint64_t *host_taxids;    int32_t n_host;     /* --host-taxids     */
    const uint64_t *exclude_taxids; int32_t n_exclude;  /* --exclude-taxids  */
} cfo_params;

#define CFO_MERGED 0xffffffffu     /* uniqueID of an entry merged up the tree */

typedef struct {
    uint64_t tax_id;
    uint32_t unique_id;      /* reference-sequence index, or CFO_MERGED      */
    uint32_t score;
    uint32_t hit_len;        /* (uint64)summedHitLen                         */
    uint32_t pad;
} cfo_row;

/* one partial hit, as left by searchForwardAndReverse (after extend/twin/trim) */
typedef struct {
    uint64_t top, bot;
    uint32_t bwoff, len;
} cfo_hit;

/* per-read operation counters used for the algorithmic-bytes figure
 * (SURVEY.md §8(d)): see cfo_counters_get(). */
typedef struct {
    uint64_t n_ftab, n_pair, n_pair2, n_single, n_walk, n_rows, n_ranges, n_reads;
} cfo_opcounts;

cfo_index *cfo_index_open(const char *basename);
void       cfo_index_close(cfo_index *);
const char *cfo_last_error(void);

/* index facts (for tests / formatting) */
uint64_t cfo_index_len(const cfo_index *);
uint64_t cfo_index_nref(const cfo_index *);
int      cfo_index_compressed(const cfo_index *);
int      cfo_index_offw(const cfo_index *);
const char *cfo_index_uid(const cfo_index *, uint64_t ref);
uint64_t cfo_index_ref_taxid(const cfo_index *, uint64_t ref);
/* seqID column: uid if tree node of taxID is a leaf or absent, else rank string
 * (aln_sink.h:2219-2234, classifier.h:557).  Returns a static/owned string. */
const char *cfo_format_seqid(const cfo_index *, uint32_t unique_id, uint64_t tax_id);
/* taxonomy lookups for the report */
int      cfo_tax_rank(const cfo_index *, uint64_t tax_id);         /* RANK_* or 0 */
const char *cfo_tax_rank_string(int rank);
const char *cfo_tax_name(const cfo_index *, uint64_t tax_id);      /* "" if none  */
uint64_t cfo_tax_size(const cfo_index *, uint64_t tax_id);         /* 0 if none   */

/* pat.h:55-91 — per-read seed from sequence codes (0..4), qualities, name. */
uint32_t cfo_gen_rand_seed(const uint8_t *seq, const uint8_t *qual, uint64_t len,
                           const char *name, uint64_t namelen, uint32_t seed);

/* Scoring::nFilter (scoring.cpp:104-117) + length filter (centrifuge.cpp:2562-2577):
 * 1 if the mate takes part in classification. */
int cfo_mate_passes(const uint8_t *seq, uint64_t len);

/*
 * Classify n_queries reads (paired=0) or pairs (paired=1; mates of query q are
 * reads 2q and 2q+1).  seq holds base codes 0..4 (A,C,G,T,N); read r occupies
 * seq[off[r] .. off[r+1]).  seeds[r] is genRandSeed of read r.
 * Output per query q: n_rows[q] (0 = "unclassified"), rows[q*k .. q*k+n_rows)
 * already in print order, score2[q] = 2ndBestScore.
 * Returns 0 on success.
 */
int cfo_classify(const cfo_index *, const cfo_params *,
                 const uint8_t *seq, const uint64_t *off, const uint32_t *seeds,
                 uint64_t n_queries, int paired,
                 cfo_row *rows, uint32_t *n_rows, uint32_t *score2,
                 cfo_opcounts *ops /* may be NULL; accumulated */);

/* Debug/parity taps for the GPU kernels. */
/* hits of one mate after searchForwardAndReverse; returns counts in nhits[2].
 * hits_fw / hits_rc must hold at least len+2 entries. */
int cfo_search(const cfo_index *, const cfo_params *, const uint8_t *seq, uint64_t len,
               cfo_hit *hits_fw, cfo_hit *hits_rc, uint32_t nhits[2]);
/* resolve one BW row to a reference-sequence index (group_walk.h / tryOffset) */
uint64_t cfo_resolve_row(const cfo_index *, uint64_t row);
/* rank(c,row) = LF(row,c)  (bt2_idx.h:2192-2227) */
uint64_t cfo_rank(const cfo_index *, int c, uint64_t row);
/* libstdc++ std::sort order of hits under compareBWTHits (classifier.h:1058) */
void cfo_sort_hits(cfo_hit *hits, uint32_t n);

#ifdef __cplusplus
}
#endif
#endif
