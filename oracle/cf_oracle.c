/*
 * cf_oracle.c — CPU restatement (plain C11) of Centrifuge 1.0.4's per-read
 * FM-index classification path.  TEST INFRASTRUCTURE ONLY — see cf_oracle.h.
 *
 * Written from the behaviour of the cited reference lines;
 * no reference source is copied.  Array-based, single-threaded, no dependency
 * on the product code in centrifuge_amd/.
 */
#define _GNU_SOURCE
#define _FILE_OFFSET_BITS 64
#include "cf_oracle.h"
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <ctype.h>

#define MASK64 0xffffffffffffffffull
#define NRANKS 10                      /* taxonomy.h:63 */

static char g_err[512];
const char *cfo_last_error(void) { return g_err; }
#define FAIL(...) do { snprintf(g_err, sizeof g_err, __VA_ARGS__); g_err[sizeof g_err - 1] = 0; goto fail; } while (0)

/* ------------------------------------------------------------------ index */

typedef struct { uint64_t tid, parent; uint8_t rank, leaf; } tnode;
typedef struct { uint64_t tid; char *name; } tname;
typedef struct { uint64_t tid, size; } tsize;
typedef struct { uint64_t row; uint32_t ref; } bound;

struct cfo_index {
    /* .1.cf header + derived geometry  (bt2_idx.h:133-167, bt2_io.h:138-207) */
    uint64_t len; int32_t lineRate, offRate, ftabChars;
    uint64_t numSides, ebwtTotLen, ftabLen, eftabLen, offsLen;
    uint64_t nPat, nFrag; uint64_t *plen;
    uint8_t *ebwt; uint64_t zOff, fchr[5]; uint64_t *ftab, *eftab;
    /* .2.cf  (bt2_io.h:528-641): u16 when nPat <= 65535 else u32 */
    int offw; uint16_t *offs16; uint32_t *offs32;
    /* .3.cf  (bt2_idx.h:623-787) */
    uint64_t nref; char **uid; uint64_t *uid_tid; int compressed;
    uint64_t ntree; tnode *tree;             /* sorted by tid */
    uint64_t nname; tname *names;            /* sorted by tid */
    uint64_t nsize; tsize *sizes;            /* sorted by tid */
    /* path table (taxonomy.h:96-160) */
    uint64_t npath; uint64_t *path_tid; uint64_t (*paths)[NRANKS]; /* sorted by tid */
    /* .4.cf  (bt2_idx.h:789-853) */
    uint64_t nbound; bound *bounds; uint64_t lastBoundary;   /* sorted by row */
    char seqid_buf[64];
};

static int rd(FILE *f, void *p, size_t n) { return fread(p, 1, n, f) == n ? 0 : -1; }

static int cmp_tnode(const void *a, const void *b) {
    uint64_t x = ((const tnode *)a)->tid, y = ((const tnode *)b)->tid; return x < y ? -1 : x > y; }
static int cmp_tname(const void *a, const void *b) {
    uint64_t x = ((const tname *)a)->tid, y = ((const tname *)b)->tid; return x < y ? -1 : x > y; }
static int cmp_tsize(const void *a, const void *b) {
    uint64_t x = ((const tsize *)a)->tid, y = ((const tsize *)b)->tid; return x < y ? -1 : x > y; }
static int cmp_bound(const void *a, const void *b) {
    uint64_t x = ((const bound *)a)->row, y = ((const bound *)b)->row; return x < y ? -1 : x > y; }

static const tnode *tree_find(const cfo_index *ix, uint64_t tid) {
    uint64_t lo = 0, hi = ix->ntree;
    while (lo < hi) { uint64_t m = (lo + hi) / 2;
        if (ix->tree[m].tid < tid) lo = m + 1; else hi = m; }
    return (lo < ix->ntree && ix->tree[lo].tid == tid) ? &ix->tree[lo] : NULL;
}

/* taxonomy.h:151-160 getPath: NULL = empty path (size 0) */
static const uint64_t *path_find(const cfo_index *ix, uint64_t tid) {
    uint64_t lo = 0, hi = ix->npath;
    while (lo < hi) { uint64_t m = (lo + hi) / 2;
        if (ix->path_tid[m] < tid) lo = m + 1; else hi = m; }
    return (lo < ix->npath && ix->path_tid[lo] == tid) ? ix->paths[lo] : NULL;
}

/* taxonomy.h:66-93 rank_to_pathID */
static int rank_to_slot(int rank) {
    switch (rank) {
        case 1: case 20: return 0;   /* strain, subspecies */
        case 2: return 1; case 3: return 2; case 4: return 3; case 5: return 4;
        case 6: return 5; case 7: return 6; case 8: return 7;
        case 24: return 8;           /* superkingdom */
        case 9: return 9;            /* domain */
        default: return -1;
    }
}

/* taxonomy.h:96-149 buildPaths: one 10-slot path per distinct uid-table taxid
 * that is in the tree; stored sorted by taxid for getPath's lookup. */
static int build_paths_fast(cfo_index *ix) {
    /* mark first occurrences using a sorted (tid, idx) list */
    typedef struct { uint64_t tid, idx; } ti;
    ti *a = malloc(sizeof(ti) * ix->nref);
    if (!a) return -1;
    for (uint64_t i = 0; i < ix->nref; i++) { a[i].tid = ix->uid_tid[i]; a[i].idx = i; }
    qsort(a, ix->nref, sizeof(ti), cmp_tsize);
    uint64_t n = 0;
    ix->path_tid = malloc(sizeof(uint64_t) * (ix->nref + 1));
    ix->paths = malloc(sizeof(uint64_t[NRANKS]) * (ix->nref + 1));
    if (!ix->path_tid || !ix->paths) { free(a); return -1; }
    for (uint64_t i = 0; i < ix->nref; i++) {
        if (i > 0 && a[i].tid == a[i - 1].tid) continue;
        uint64_t tid = a[i].tid;
        if (!tree_find(ix, tid)) continue;
        uint64_t *p = ix->paths[n]; memset(p, 0, sizeof(uint64_t) * NRANKS);
        ix->path_tid[n++] = tid;
        int first = 1;
        for (;;) {
            const tnode *nd = tree_find(ix, tid);
            if (!nd) break;
            int slot = (first && nd->rank == 0) ? 0 : rank_to_slot(nd->rank);
            if (slot >= 0 && slot < NRANKS && p[slot] == 0) p[slot] = tid;
            first = 0;
            if (nd->parent == tid) break;
            tid = nd->parent;
        }
    }
    free(a);
    ix->npath = n;   /* already sorted by tid */
    return 0;
}

/* taxonomy.h:165-205 tax_rank_num: only "is numbered below species" matters
 * here (bt2_idx.h:722-723).  Numbered 0: strain, subspecies, and RANK_LIFE
 * (never assigned, so it keeps the zero of static storage). */
static int rank_below_species(int rank) { return rank == 1 || rank == 20 || rank == 29 /* RANK_LIFE: never numbered */; }

cfo_index *cfo_index_open(const char *base) {
    cfo_index *ix = calloc(1, sizeof *ix);
    FILE *f = NULL; char path[4096];
    if (!ix) return NULL;
    /* ---- .1.cf  (bt2_io.h:138-526) */
    snprintf(path, sizeof path, "%s.1.cf", base);
    f = fopen(path, "rb"); if (!f) FAIL("cannot open %s", path);
    int32_t one, lps, flags;
    if (rd(f, &one, 4) || one != 1) FAIL("%s: bad endian word", path);
    if (rd(f, &ix->len, 8) || rd(f, &ix->lineRate, 4) || rd(f, &lps, 4) ||
        rd(f, &ix->offRate, 4) || rd(f, &ix->ftabChars, 4) || rd(f, &flags, 4)) FAIL("%s: short header", path);
    if (ix->lineRate != 7) FAIL("%s: lineRate %d unsupported (expect 7)", path, ix->lineRate);
    {   /* EbwtParams::init bt2_idx.h:133-167 */
        uint64_t bwtSz = ix->len / 4 + 1, sideSz = 1ull << ix->lineRate, sideBwtSz = sideSz - 32;
        ix->numSides = (bwtSz + sideBwtSz - 1) / sideBwtSz;
        ix->ebwtTotLen = ix->numSides * sideSz;
        ix->ftabLen = (1ull << (2 * ix->ftabChars)) + 1;
        ix->eftabLen = 2ull * ix->ftabChars;
        ix->offsLen = (ix->len + 1 + (1ull << ix->offRate) - 1) >> ix->offRate;
    }
    if (rd(f, &ix->nPat, 8)) FAIL("%s: short", path);
    ix->plen = malloc(8 * (ix->nPat + 1));
    if (!ix->plen || rd(f, ix->plen, 8 * ix->nPat)) FAIL("%s: plen", path);
    ix->offw = ix->nPat > 65535;                                  /* bt2_io.h:280 */
    if (rd(f, &ix->nFrag, 8)) FAIL("%s: short", path);
    if (fseeko(f, (off_t)(24 * ix->nFrag), SEEK_CUR)) FAIL("%s: rstarts", path);
    ix->ebwt = malloc(ix->ebwtTotLen);
    if (!ix->ebwt || rd(f, ix->ebwt, ix->ebwtTotLen)) FAIL("%s: ebwt", path);
    if (rd(f, &ix->zOff, 8) || rd(f, ix->fchr, 40)) FAIL("%s: zOff/fchr", path);
    ix->ftab = malloc(8 * ix->ftabLen); ix->eftab = malloc(8 * ix->eftabLen);
    if (!ix->ftab || !ix->eftab || rd(f, ix->ftab, 8 * ix->ftabLen) || rd(f, ix->eftab, 8 * ix->eftabLen))
        FAIL("%s: ftab", path);
    fclose(f); f = NULL;
    /* ---- .2.cf */
    snprintf(path, sizeof path, "%s.2.cf", base);
    f = fopen(path, "rb"); if (!f) FAIL("cannot open %s", path);
    if (rd(f, &one, 4) || one != 1) FAIL("%s: bad endian word", path);
    if (ix->offw) { ix->offs32 = malloc(4 * ix->offsLen); if (!ix->offs32 || rd(f, ix->offs32, 4 * ix->offsLen)) FAIL("%s: offs", path); }
    else          { ix->offs16 = malloc(2 * ix->offsLen); if (!ix->offs16 || rd(f, ix->offs16, 2 * ix->offsLen)) FAIL("%s: offs", path); }
    fclose(f); f = NULL;
    /* ---- .3.cf */
    snprintf(path, sizeof path, "%s.3.cf", base);
    f = fopen(path, "rb"); if (!f) FAIL("cannot open %s", path);
    if (rd(f, &one, 4) || rd(f, &ix->nref, 8)) FAIL("%s: short", path);
    ix->uid = calloc(ix->nref + 1, sizeof(char *)); ix->uid_tid = malloc(8 * (ix->nref + 1));
    if (!ix->uid || !ix->uid_tid) FAIL("oom");
    uint64_t ncid = 0;
    for (uint64_t i = 0; i < ix->nref; i++) {
        char buf[4096]; size_t n = 0;
        for (;;) {   /* `in3 >> c` skips every whitespace byte, so only '\0' (or EOF)
                      * terminates a uid: bt2_idx.h:642-647 */
            int c = fgetc(f);
            if (c == EOF || c == 0) break;
            if (isspace(c)) continue;
            if (n + 1 < sizeof buf) buf[n++] = (char)c;
        }
        buf[n] = 0;
        ix->uid[i] = strdup(buf);
        if (strncmp(buf, "cid", 3) == 0) ncid++;
        if (rd(f, &ix->uid_tid[i], 8)) FAIL("%s: uid table", path);
    }
    ix->compressed = ncid >= 10;                                 /* bt2_idx.h:661-663 */
    if (rd(f, &ix->ntree, 8)) FAIL("%s: tree", path);
    ix->tree = malloc(sizeof(tnode) * (ix->ntree + 1)); if (!ix->tree) FAIL("oom");
    for (uint64_t i = 0; i < ix->ntree; i++) {
        uint16_t rk;
        if (rd(f, &ix->tree[i].tid, 8) || rd(f, &ix->tree[i].parent, 8) || rd(f, &rk, 2)) FAIL("%s: tree", path);
        ix->tree[i].rank = (uint8_t)rk; ix->tree[i].leaf = 0;
    }
    qsort(ix->tree, ix->ntree, sizeof(tnode), cmp_tnode);
    {   /* std::map semantics: later duplicates overwrite; keep the last of equal tids */
        uint64_t w = 0;
        for (uint64_t i = 0; i < ix->ntree; i++) {
            if (w > 0 && ix->tree[w - 1].tid == ix->tree[i].tid) ix->tree[w - 1] = ix->tree[i];
            else ix->tree[w++] = ix->tree[i];
        }
        ix->ntree = w;
    }
    for (uint64_t i = 0; i < ix->nref; i++) {                    /* leaf = tid in uid table, :673 */
        tnode *nd = (tnode *)tree_find(ix, ix->uid_tid[i]); if (nd) nd->leaf = 1;
    }
    if (rd(f, &ix->nname, 8)) FAIL("%s: names", path);
    ix->names = calloc(ix->nname + 1, sizeof(tname)); if (!ix->names) FAIL("oom");
    for (uint64_t i = 0; i < ix->nname; i++) {
        char buf[4096]; size_t n = 0; int c;
        if (rd(f, &ix->names[i].tid, 8)) FAIL("%s: names", path);
        /* in3 >> name: skip leading whitespace, read to whitespace; then skip 1 byte */
        while ((c = fgetc(f)) != EOF && isspace(c)) {}
        while (c != EOF && !isspace(c)) { if (n + 1 < sizeof buf) buf[n++] = (char)c; c = fgetc(f); }
        buf[n] = 0;   /* the terminating whitespace byte ('\n') has been consumed == seekg(1) */
        for (size_t k = 0; k < n; k++) if (buf[k] == '@') buf[k] = ' ';
        ix->names[i].name = strdup(buf);
    }
    qsort(ix->names, ix->nname, sizeof(tname), cmp_tname);
    if (rd(f, &ix->nsize, 8)) FAIL("%s: sizes", path);
    ix->sizes = calloc(ix->nsize + ix->ntree + 1, sizeof(tsize)); if (!ix->sizes) FAIL("oom");
    for (uint64_t i = 0; i < ix->nsize; i++)
        if (rd(f, &ix->sizes[i].tid, 8) || rd(f, &ix->sizes[i].size, 8)) FAIL("%s: sizes", path);
    fclose(f); f = NULL;
    qsort(ix->sizes, ix->nsize, sizeof(tsize), cmp_tsize);
    {   /* genome-size roll-up, bt2_idx.h:704-745 */
        uint64_t n0 = ix->nsize;
        uint64_t *sum = calloc(ix->ntree + 1, 8), *cnt = calloc(ix->ntree + 1, 8);
        if (!sum || !cnt) FAIL("oom");
        for (uint64_t i = 0; i < n0; i++) {
            uint64_t c = ix->sizes[i].tid; const tnode *nd = tree_find(ix, c);
            if (!nd || nd->parent == c) continue;
            if (!((nd->rank == 0 && nd->leaf) || rank_below_species(nd->rank))) continue;
            c = nd->parent;
            for (;;) {
                const tnode *p = tree_find(ix, c); if (!p) break;
                if (p->rank >= 2 && p->rank <= 7) { sum[p - ix->tree] += ix->sizes[i].size; cnt[p - ix->tree]++; }
                if (c == p->parent) break;
                c = p->parent;
            }
        }
        for (uint64_t t = 0; t < ix->ntree; t++) if (cnt[t]) {
            uint64_t tid = ix->tree[t].tid, v = sum[t] / cnt[t];
            uint64_t lo = 0, hi = n0;
            while (lo < hi) { uint64_t md = (lo + hi) / 2; if (ix->sizes[md].tid < tid) lo = md + 1; else hi = md; }
            if (lo < n0 && ix->sizes[lo].tid == tid) ix->sizes[lo].size = v;
            else { ix->sizes[ix->nsize].tid = tid; ix->sizes[ix->nsize].size = v; ix->nsize++; }
        }
        free(sum); free(cnt);
        qsort(ix->sizes, ix->nsize, sizeof(tsize), cmp_tsize);
    }
    if (build_paths_fast(ix)) FAIL("oom paths");
    /* ---- .4.cf (optional) */
    snprintf(path, sizeof path, "%s.4.cf", base);
    f = fopen(path, "rb");
    if (f) {
        uint64_t m = 0;
        if (rd(f, &one, 4) || rd(f, &m, 8)) m = 0;
        ix->bounds = malloc(sizeof(bound) * (m + 1)); if (!ix->bounds) FAIL("oom");
        for (uint64_t i = 0; i < m; i++) {
            if (rd(f, &ix->bounds[i].row, 8) || rd(f, &ix->bounds[i].ref, 4)) FAIL("%s: short", path);
            if (ix->bounds[i].row > ix->lastBoundary) ix->lastBoundary = ix->bounds[i].row;
        }
        ix->nbound = m;
        qsort(ix->bounds, m, sizeof(bound), cmp_bound);
        {   /* std::map: a repeated key keeps the LAST value assigned; qsort is
             * not stable, so resolve duplicates deterministically is impossible
             * here — the builder never writes duplicate rows. */
        }
        fclose(f); f = NULL;
    }
    return ix;
fail:
    if (f) fclose(f);
    cfo_index_close(ix);
    return NULL;
}

void cfo_index_close(cfo_index *ix) {
    if (!ix) return;
    free(ix->plen); free(ix->ebwt); free(ix->ftab); free(ix->eftab); free(ix->offs16); free(ix->offs32);
    if (ix->uid) for (uint64_t i = 0; i < ix->nref; i++) free(ix->uid[i]);
    free(ix->uid); free(ix->uid_tid); free(ix->tree);
    if (ix->names) for (uint64_t i = 0; i < ix->nname; i++) free(ix->names[i].name);
    free(ix->names); free(ix->sizes); free(ix->path_tid); free(ix->paths); free(ix->bounds);
    free(ix);
}

uint64_t cfo_index_len(const cfo_index *ix) { return ix->len; }
uint64_t cfo_index_nref(const cfo_index *ix) { return ix->nref; }
int cfo_index_compressed(const cfo_index *ix) { return ix->compressed; }
int cfo_index_offw(const cfo_index *ix) { return ix->offw; }
const char *cfo_index_uid(const cfo_index *ix, uint64_t r) { return r < ix->nref ? ix->uid[r] : ""; }
uint64_t cfo_index_ref_taxid(const cfo_index *ix, uint64_t r) { return r < ix->nref ? ix->uid_tid[r] : 0; }

int cfo_tax_rank(const cfo_index *ix, uint64_t tid) { const tnode *n = tree_find(ix, tid); return n ? n->rank : 0; }

const char *cfo_tax_rank_string(int rank) {          /* taxonomy.h:207-239 */
    static const char *s[] = { "no rank", "strain", "species", "genus", "family", "order", "class", "phylum",
        "kingdom", "no rank" /* domain */, "forma", "infraclass", "infraorder", "parvorder", "subclass",
        "subfamily", "subgenus", "subkingdom", "suborder", "subphylum", "subspecies", "subtribe", "superclass",
        "superfamily", "superkingdom", "superorder", "superphylum", "tribe", "varietas", "life" };
    return (rank >= 0 && rank < 30) ? s[rank] : "no rank";
}

const char *cfo_tax_name(const cfo_index *ix, uint64_t tid) {
    uint64_t lo = 0, hi = ix->nname;
    while (lo < hi) { uint64_t m = (lo + hi) / 2; if (ix->names[m].tid < tid) lo = m + 1; else hi = m; }
    return (lo < ix->nname && ix->names[lo].tid == tid) ? ix->names[lo].name : "";
}
uint64_t cfo_tax_size(const cfo_index *ix, uint64_t tid) {
    uint64_t lo = 0, hi = ix->nsize;
    while (lo < hi) { uint64_t m = (lo + hi) / 2; if (ix->sizes[m].tid < tid) lo = m + 1; else hi = m; }
    return (lo < ix->nsize && ix->sizes[lo].tid == tid) ? ix->sizes[lo].size : 0;
}

/* classifier.h:546-557 + aln_sink.h:2219-2234 */
const char *cfo_format_seqid(const cfo_index *ix, uint32_t unique_id, uint64_t tax_id) {
    const tnode *nd = tree_find(ix, tax_id);
    int rank = nd ? nd->rank : 0, leaf = nd ? nd->leaf : 1;
    if (leaf && unique_id != CFO_MERGED && unique_id < ix->nref) return ix->uid[unique_id];
    return cfo_tax_rank_string(rank);
}

/* ------------------------------------------------------------ rank and LF */

static inline int popc64(uint64_t x) { return __builtin_popcountll(x); }

/* number of chars == c among the first `off` chars of side `s`
 * (bt2_idx.h:505-517 countInU64, :2364-2425 countUpTo; chars little-end first) */
static inline uint64_t count_upto(const uint8_t *side, int c, uint32_t off) {
    static const uint64_t ctab[4] = { MASK64, 0xaaaaaaaaaaaaaaaaull, 0x5555555555555555ull, 0 };
    uint64_t cnt = 0; uint32_t w = 0;
    for (; off >= 32; off -= 32, w++) {
        uint64_t x; memcpy(&x, side + 8 * w, 8); x ^= ctab[c];
        cnt += popc64(x & (x >> 1) & 0x5555555555555555ull);
    }
    if (off) {
        uint64_t x; memcpy(&x, side + 8 * w, 8); x ^= ctab[c];
        x = x & (x >> 1) & 0x5555555555555555ull;
        cnt += popc64(x & ((1ull << (2 * off)) - 1));
    }
    return cnt;
}

/* bt2_idx.h:2192-2227 countBt2Side → LF(row,c) */
uint64_t cfo_rank(const cfo_index *ix, int c, uint64_t row) {
    uint64_t s = row / 384; uint32_t off = (uint32_t)(row % 384);
    const uint8_t *side = ix->ebwt + s * 128;
    uint64_t cnt = count_upto(side, c, off), occ;
    if (c == 0 && ix->zOff / 384 == s && (uint32_t)(ix->zOff % 384) < off) cnt--;   /* '$' stored as A */
    memcpy(&occ, side + 96 + 8 * c, 8);
    return ix->fchr[c] + occ + cnt;
}

static inline int bwt_char(const cfo_index *ix, uint64_t row) {  /* rowL bt2_idx.h:2737-2752 */
    uint64_t s = row / 384; uint32_t off = (uint32_t)(row % 384);
    return (ix->ebwt[s * 128 + (off >> 2)] >> (2 * (off & 3))) & 3;
}

static inline uint64_t ftab_hi(const cfo_index *ix, uint64_t i) {  /* bt2_idx.h:1880-1897 */
    uint64_t v = ix->ftab[i]; return v <= ix->len ? v : ix->eftab[(v ^ MASK64) * 2 + 1]; }
static inline uint64_t ftab_lo(const cfo_index *ix, uint64_t i) {  /* bt2_idx.h:1953-1970 */
    uint64_t v = ix->ftab[i]; return v <= ix->len ? v : ix->eftab[(v ^ MASK64) * 2]; }

/* bt2_idx.h:1980-2014 tryOffset */
static inline uint64_t try_offset(const cfo_index *ix, uint64_t row) {
    if (row == ix->zOff) return 0;
    if ((row & (MASK64 << ix->offRate)) == row) {
        uint64_t e = row >> ix->offRate;
        return ix->offw ? ix->offs32[e] : ix->offs16[e];
    }
    if (ix->lastBoundary > 0 && row <= ix->lastBoundary) {
        uint64_t lo = 0, hi = ix->nbound;
        while (lo < hi) { uint64_t m = (lo + hi) / 2; if (ix->bounds[m].row < row) lo = m + 1; else hi = m; }
        if (lo < ix->nbound && ix->bounds[lo].row == row)
            return ix->offw ? ix->bounds[lo].ref : (uint16_t)ix->bounds[lo].ref;
    }
    return MASK64;
}

static cfo_opcounts *g_ops;   /* optional op counters (single-threaded oracle) */

/* group_walk.h:1154-1209 advanceElement, observable per-row semantics */
uint64_t cfo_resolve_row(const cfo_index *ix, uint64_t row) {
    for (;;) {
        uint64_t r = try_offset(ix, row);
        if (r != MASK64) return r;
        row = cfo_rank(ix, bwt_char(ix, row), row);      /* mapLF1(row&,l) bt2_idx.h:2941 */
        if (g_ops) g_ops->n_walk++;
    }
}

/* ---------------------------------------------------------------- search */

typedef struct { uint64_t top, bot, bwoff, len; } hit_t;          /* hi_aligner.h:58-142 */
typedef struct { uint64_t cur; int done; uint32_t n; hit_t *h; } strand_t;  /* ReadBWTHit :149-319 */

/* strand char i (0 = leftmost): fw = read, rc = reverse complement, N stays 4
 * (read.h:138-151, sstring.h:2928-2934) */
static inline int sch(const uint8_t *s, uint64_t L, int fw, uint64_t i) {
    if (fw) return s[i];
    int c = s[L - 1 - i]; return c > 3 ? 4 : (c ^ 3);
}

static inline void push_hit(strand_t *st, uint64_t top, uint64_t bot, uint64_t bwoff, uint64_t len) {
    hit_t *h = &st->h[st->n++]; h->top = top; h->bot = bot; h->bwoff = (uint32_t)bwoff; h->len = (uint32_t)len;
}

/* HI_Aligner::partialSearch hi_aligner.h:902-1031 */
static void partial_search(const cfo_index *ix, const uint8_t *s, uint64_t L, int fw, strand_t *st) {
    const uint64_t ftc = (uint64_t)ix->ftabChars;
    uint64_t offset = st->cur, dep = offset, left = L - dep, top, bot;
    if (left < ftc) {                                            /* :934-944 */
        st->cur = L; push_hit(st, MASK64, MASK64, offset, st->cur - offset); st->done = 1; return;
    }
    for (uint64_t i = 0; i < ftc; i++) {                          /* :946-961 */
        if (sch(s, L, fw, L - dep - 1 - i) > 3) {
            st->cur += i + 1; push_hit(st, MASK64, MASK64, offset, st->cur - offset);
            if (st->cur >= L) st->done = 1;
            return;
        }
    }
    {                                                            /* :964 ftabLoHi, bt2_idx.h:1830-1851 */
        uint64_t fi = 0, p = L - dep - ftc;
        for (uint64_t i = 0; i < ftc; i++) fi = (fi << 2) | (uint64_t)sch(s, L, fw, p + i);
        top = ftab_hi(ix, fi); bot = ftab_lo(ix, fi + 1);
        if (g_ops) g_ops->n_ftab++;
    }
    dep += ftc;
    if (bot <= top) {                                            /* :966-978 */
        st->cur = dep; push_hit(st, MASK64, MASK64, offset, st->cur - offset);
        if (st->cur >= L) st->done = 1;
        return;
    }
    while (dep < L) {                                            /* :981-1007 */
        int c = sch(s, L, fw, L - dep - 1);
        uint64_t t, b;
        if (c > 3) break;
        if (bot - top > 1) {                                     /* mapLF on both loci */
            t = cfo_rank(ix, c, top); b = cfo_rank(ix, c, bot);
            if (g_ops) { g_ops->n_pair++; if ((top % 384) + (bot - top) >= 384) g_ops->n_pair2++; }  /* bt2_idx.h:339 */
        } else {                                                 /* mapLF1 bt2_idx.h:2910-2934 */
            if (g_ops) g_ops->n_single++;
            if (bwt_char(ix, top) != c || top == ix->zOff) { t = b = 0; }
            else { t = cfo_rank(ix, c, top); b = t + 1; }
        }
        if (b <= t) break;
        top = t; bot = b; dep++;
    }
    push_hit(st, top, bot, offset, dep - offset);                /* :1010-1029 */
    st->cur = dep;
    if (st->cur >= L) st->done = 1;
}

static inline void hit_reset(hit_t *h) { h->top = h->bot = 0; h->bwoff = MASK64; h->len = 0; }  /* :63-71 */

/* Classifier::searchForwardAndReverse classifier.h:646-896.
 * st[0] = fw strand, st[1] = rc strand; both with enough capacity. */
static void search_fw_rc(const cfo_index *ix, const uint8_t *s, uint64_t L, uint64_t m, uint64_t inc,
                         uint64_t ihits, strand_t st[2], hit_t *tmpbuf) {
    uint64_t sum[2] = { 0, 0 };
    /* The reference alternates strands (:666-772); each strand's chain of
     * partialSearch calls depends only on that strand's state, so running them
     * one after the other yields the same hit lists. */
    for (int fwi = 0; fwi < 2; fwi++) {
        strand_t *h = &st[fwi];
        while (!h->done) {
            partial_search(ix, s, L, fwi == 0, h);
            hit_t *last = &h->h[h->n - 1];
            if (h->done) { if (last->len >= m) sum[fwi] += last->len; break; }
            if (last->len >= m) sum[fwi] += last->len;
            if (last->len > inc) h->cur += 1;                    /* :727-761 */
            if (h->cur + m >= L) { h->done = 1; break; }         /* :762-766 */
        }
    }
    if (sum[0] >= m && sum[1] >= m) {
        /* extend, :790-847 */
        for (uint32_t i = 0; i < st[0].n; i++) {
            hit_t *hit = &st[0].h[i];
            uint64_t len = hit->len, l = hit->bwoff, r = hit->bwoff + len;
            for (uint32_t j = 0; j < st[1].n; j++) {
                hit_t *rc = &st[1].h[j];
                uint64_t rclen = rc->len;
                if (len < m && rclen < m) continue;
                uint64_t rc_l = L - rc->bwoff - rc->len, rc_r = rc_l + rclen;
                if (r <= rc_l) continue;
                if (rc_r <= l) continue;
                if (l == rc_l && r == rc_r) continue;
                if (l < rc_l && r > rc_r) continue;
                if (l > rc_l && r < rc_r) continue;
                if (l > rc_l) {
                    strand_t t = { rc_l, 0, 0, tmpbuf };
                    partial_search(ix, s, L, 1, &t);
                    if (t.h[0].len == len + l - rc_l) *hit = t.h[0];
                }
                if (r > rc_r) {
                    strand_t t = { L - r, 0, 0, tmpbuf };
                    partial_search(ix, s, L, 0, &t);
                    if (t.h[0].len == rclen + r - rc_r) *rc = t.h[0];
                }
            }
        }
        /* twins, :849-870 */
        for (uint32_t i = 0; i < st[0].n; i++) {
            hit_t *hit = &st[0].h[i];
            uint64_t len = hit->len, l = hit->bwoff, r = hit->bwoff + len;
            for (uint32_t j = 0; j < st[1].n; j++) {
                hit_t *rc = &st[1].h[j];
                uint64_t rclen = rc->len, rc_l = L - rc->bwoff - rc->len, rc_r = rc_l + rclen;
                if (rc_l < l) break;
                if (len != rclen) continue;
                if (l == rc_l && r == rc_r && (hit->bot - hit->top) + (rc->bot - rc->top) > ihits) {
                    hit_reset(hit); hit_reset(rc); break;
                }
            }
        }
    }
    /* trim, :873-895 */
    for (int fwi = 0; fwi < 2; fwi++) {
        strand_t *h = &st[fwi];
        if (h->n < 2) continue;
        for (uint32_t i = 0; i + 1 < h->n; i++) {
            hit_t *a = &h->h[i];
            for (uint32_t j = i + 1; j < h->n; j++) {
                hit_t *b = &h->h[j];
                if (a->bwoff >= b->bwoff) { a->len = 0; break; }
                if (a->bwoff + a->len <= b->bwoff) break;
                if (a->len >= b->len) { uint64_t e = b->bwoff + b->len; b->bwoff = a->bwoff + a->len; b->len = e - b->bwoff; }
                else a->len = b->bwoff - a->bwoff;
            }
        }
    }
}

/* ------------------------------------------- libstdc++ std::sort, restated */
/* classifier.h:1058-1086 compareBWTHits (the literal 22 is the reference's) */
static int hit_less(const hit_t *a, const hit_t *b) {
    uint64_t as = a->bot - a->top, bs = b->bot - b->top;
    if (a->len >= 22 || b->len >= 22) {
        if (a->len >= 22 && b->len >= 22) { if (as < bs) return 1; if (as > bs) return 0; }
        if (b->len < a->len) return 1;
        if (b->len > a->len) return 0;
    }
    if (b->len * as < a->len * bs) return 1;
    if (b->len * as > a->len * bs) return 0;
    if (as < bs) return 1;
    if (as > bs) return 0;
    if (b->len < a->len) return 1;
    if (b->len > a->len) return 0;
    return 0;
}
#define SWAPH(x, y) do { hit_t _t = (x); (x) = (y); (y) = _t; } while (0)

/* bits/stl_algo.h (g++ 11): __unguarded_linear_insert / __insertion_sort */
static void ss_unguarded_linear_insert(hit_t *a, long last) {
    hit_t val = a[last]; long next = last - 1;
    while (hit_less(&val, &a[next])) { a[last] = a[next]; last = next; --next; }
    a[last] = val;
}
static void ss_insertion_sort(hit_t *a, long first, long last) {
    if (first == last) return;
    for (long i = first + 1; i != last; ++i) {
        if (hit_less(&a[i], &a[first])) {
            hit_t val = a[i];
            memmove(&a[first + 1], &a[first], sizeof(hit_t) * (size_t)(i - first));
            a[first] = val;
        } else ss_unguarded_linear_insert(a, i);
    }
}
/* bits/stl_heap.h: __push_heap / __adjust_heap / make_heap / sort_heap */
static void ss_push_heap(hit_t *a, long first, long hole, long top, hit_t val) {
    long parent = (hole - 1) / 2;
    while (hole > top && hit_less(&a[first + parent], &val)) {
        a[first + hole] = a[first + parent]; hole = parent; parent = (hole - 1) / 2;
    }
    a[first + hole] = val;
}
static void ss_adjust_heap(hit_t *a, long first, long hole, long len, hit_t val) {
    const long top = hole; long child = hole;
    while (child < (len - 1) / 2) {
        child = 2 * (child + 1);
        if (hit_less(&a[first + child], &a[first + (child - 1)])) child--;
        a[first + hole] = a[first + child]; hole = child;
    }
    if ((len & 1) == 0 && child == (len - 2) / 2) {
        child = 2 * (child + 1);
        a[first + hole] = a[first + (child - 1)]; hole = child - 1;
    }
    ss_push_heap(a, first, hole, top, val);
}
static void ss_heapsort(hit_t *a, long first, long last) {      /* __partial_sort(first,last,last) */
    long len = last - first;
    if (len >= 2) {                                              /* __make_heap */
        long parent = (len - 2) / 2;
        for (;;) { hit_t v = a[first + parent]; ss_adjust_heap(a, first, parent, len, v); if (parent == 0) break; parent--; }
    }
    while (last - first > 1) {                                   /* __sort_heap / __pop_heap */
        --last;
        hit_t v = a[last]; a[last] = a[first];
        ss_adjust_heap(a, first, 0, last - first, v);
    }
}
static void ss_move_median_to_first(hit_t *a, long result, long x, long y, long z) {
    if (hit_less(&a[x], &a[y])) {
        if (hit_less(&a[y], &a[z])) SWAPH(a[result], a[y]);
        else if (hit_less(&a[x], &a[z])) SWAPH(a[result], a[z]);
        else SWAPH(a[result], a[x]);
    } else if (hit_less(&a[x], &a[z])) SWAPH(a[result], a[x]);
    else if (hit_less(&a[y], &a[z])) SWAPH(a[result], a[z]);
    else SWAPH(a[result], a[y]);
}
static long ss_unguarded_partition(hit_t *a, long first, long last, long pivot) {
    for (;;) {
        while (hit_less(&a[first], &a[pivot])) ++first;
        --last;
        while (hit_less(&a[pivot], &a[last])) --last;
        if (!(first < last)) return first;
        SWAPH(a[first], a[last]);
        ++first;
    }
}
static void ss_introsort_loop(hit_t *a, long first, long last, long depth) {
    while (last - first > 16) {
        if (depth == 0) { ss_heapsort(a, first, last); return; }
        --depth;
        long mid = first + (last - first) / 2;
        ss_move_median_to_first(a, first, first + 1, mid, last - 1);
        long cut = ss_unguarded_partition(a, first + 1, last, first);
        ss_introsort_loop(a, cut, last, depth);
        last = cut;
    }
}
static void std_sort_hits(hit_t *a, long n) {                    /* std::sort, ds.h:775-779 */
    if (n <= 0) return;
    long lg = 0; for (long t = n; t > 1; t >>= 1) lg++;
    ss_introsort_loop(a, 0, n, 2 * lg);
    if (n > 16) {                                                /* __final_insertion_sort */
        ss_insertion_sort(a, 0, 16);
        for (long i = 16; i != n; ++i) ss_unguarded_linear_insert(a, i);
    } else ss_insertion_sort(a, 0, n);
}

void cfo_sort_hits(cfo_hit *hits, uint32_t n) {
    hit_t *t = malloc(sizeof(hit_t) * (n + 1));
    for (uint32_t i = 0; i < n; i++) { t[i].top = hits[i].top; t[i].bot = hits[i].bot; t[i].bwoff = hits[i].bwoff; t[i].len = hits[i].len; }
    std_sort_hits(t, n);
    for (uint32_t i = 0; i < n; i++) { hits[i].top = t[i].top; hits[i].bot = t[i].bot; hits[i].bwoff = (uint32_t)t[i].bwoff; hits[i].len = (uint32_t)t[i].len; }
    free(t);
}

/* -------------------------------------------------------------- classify */

typedef struct {                                                 /* HitCount classifier.h:30-121 */
    uint64_t uniqueID, taxID;
    uint32_t score, scores[2][2], timeStamp;
    uint64_t hitLen, hitLens[2][2];
    const uint64_t *path; uint32_t pathlen;                      /* 10 or 0 */
    uint8_t rank, leaf; uint32_t num_leaves;
} hcount;

typedef struct { hcount *e; uint32_t n, cap; } hitmap;

static hcount *hm_push(hitmap *hm) {
    if (hm->n == hm->cap) { hm->cap = hm->cap ? hm->cap * 2 : 64; hm->e = realloc(hm->e, sizeof(hcount) * hm->cap); }
    hcount *h = &hm->e[hm->n++]; memset(h, 0, sizeof *h); h->leaf = 1; h->num_leaves = 1; return h;
}

/* classifier.h:157-201: tid is in the closure iff it is a tree node with a listed id on its root path */
static int in_closure(const cfo_index *ix, uint64_t tid, const uint64_t *list, int n) {
    if (n <= 0 || !tree_find(ix, tid)) return 0;
    uint64_t t = tid;
    for (;;) {
        for (int i = 0; i < n; i++) if (list[i] == t) return 1;
        const tnode *nd = tree_find(ix, t);
        if (!nd) return 0;
        if (t == nd->parent) return 0;
        t = nd->parent;
    }
}

/* classifier.h:982-1050 addHitToHitMap */
static void add_hit(const cfo_index *ix, const cfo_params *pr, hitmap *hm, int rdi, int fwi,
                    uint64_t uniqueID, uint64_t taxID, uint32_t ts, uint32_t score, uint64_t hitlen) {
    const uint64_t *path = path_find(ix, taxID);
    uint32_t plen = path ? NRANKS : 0;
    uint32_t rank = (uint32_t)pr->rank_slot;
    if (rank > 0) {
        for (; rank < plen; rank++) if (path[rank] != 0) { taxID = path[rank]; break; }
    }
    uint32_t idx = 0;
    for (; idx < hm->n; idx++) {
        hcount *h = &hm->e[idx];
        int same = (rank == 0) ? (uniqueID == h->uniqueID) : (taxID == h->taxID);
        if (same) {
            if (h->timeStamp != ts) {
                h->scores[rdi][fwi] += score; h->hitLens[rdi][fwi] += hitlen; h->timeStamp = ts;
            }
            break;
        }
    }
    if (idx >= hm->n) {
        hcount *h = hm_push(hm);
        h->uniqueID = uniqueID; h->taxID = taxID; h->scores[rdi][fwi] = score; h->hitLens[rdi][fwi] = hitlen;
        h->timeStamp = ts; h->path = path; h->pathlen = plen; h->rank = (uint8_t)rank;
    }
}

typedef struct { uint32_t n, cap; uint64_t *ids; } idlist;

/* LCG random_source.h:36-61 */
static inline uint32_t lcg_next(uint32_t *last) {
    uint32_t ret; *last = 1664525u * *last + 1013904223u; ret = *last >> 16;
    *last = 1664525u * *last + 1013904223u; ret ^= *last; return ret;
}

typedef struct { uint32_t score; uint32_t idx; } sbuf;

static int classify_query(const cfo_index *ix, const cfo_params *pr,
                          const uint8_t *seqs[2], const uint64_t lens[2], const int pass[2], int is_pair,
                          uint32_t seedA, uint32_t seedB,
                          cfo_row *rows, uint32_t *n_rows, uint32_t *score2) {
    const uint64_t k = (uint64_t)pr->khits, m = (uint64_t)pr->min_hitlen;
    const uint64_t ihits = (k > 5 ? k : 5) * (ix->compressed ? 4 : 40);       /* aln_sink.h:580-588 */
    const uint64_t inc = (2 * m <= 33) ? 10 : (2 * m - 33);                   /* classifier.h:226 */
    *n_rows = 0; *score2 = 0;
    /* which mates take part: centrifuge.cpp:2678-2690 */
    const uint8_t *rs[2]; uint64_t rl[2]; int nm = 0, paired = 0;
    if (is_pair && pass[0] && pass[1]) { rs[0] = seqs[0]; rl[0] = lens[0]; rs[1] = seqs[1]; rl[1] = lens[1]; nm = 2; paired = 1; }
    else if (pass[0]) { rs[0] = seqs[0]; rl[0] = lens[0]; nm = 1; }
    else if (is_pair && pass[1]) { rs[0] = seqs[1]; rl[0] = lens[1]; nm = 1; }
    else return 0;                                               /* reportUnclassified */
    uint32_t rnd = paired ? (seedA ^ seedB) : seedA;            /* centrifuge.cpp:2608-2613 */

    hitmap hm = { 0 }; int rc = 0;
    uint64_t maxG = k; uint32_t ts = 0;                          /* classifier.h:228,232 */
    uint64_t *refs = NULL; size_t refs_cap = 0;
    for (int rdi = 0; rdi < nm; rdi++) {
        const uint8_t *s = rs[rdi]; uint64_t L = rl[rdi];
        hit_t *buf = malloc(sizeof(hit_t) * (2 * (L + 2) + 4));
        strand_t st[2] = { { 0, 0, 0, buf }, { 0, 0, 0, buf + L + 2 } };
        search_fw_rc(ix, s, L, m, inc, ihits, st, buf + 2 * (L + 2));
        /* getForwardOrReverseHit :898-941 */
        uint64_t tot[2] = { 0, 0 }, mx[2] = { 0, 0 };
        for (int f = 0; f < 2; f++) for (uint32_t i = 0; i < st[f].n; i++) {
            uint64_t len = st[f].h[i].len; if (len < m) continue;
            tot[f] += (len - 15) * (len - 15); if (len > mx[f]) mx[f] = len;
        }
        int lo, hi;
        if (tot[0] != tot[1]) { lo = tot[0] > tot[1] ? 0 : 1; hi = lo + 1; }
        else if (mx[0] != mx[1]) { lo = mx[0] > mx[1] ? 0 : 1; hi = lo + 1; }
        else { lo = 0; hi = 2; }
        for (int fwi = lo; fwi < hi; fwi++) {
            strand_t *h = &st[fwi];
            for (uint32_t i = 0; i < h->n; i++)                  /* :253-261 */
                if (h->h[i].len >= m && h->h[i].bot - h->h[i].top > maxG) maxG = h->h[i].bot - h->h[i].top;
            if (maxG > k) maxG += k;                             /* :263-265 */
            std_sort_hits(h->h, h->n);                           /* :267 */
            uint64_t cnt = 0;
            for (uint32_t hi2 = 0; hi2 < h->n; hi2++, ts++) {    /* :270-372 */
                const hit_t *ph = &h->h[hi2];
                uint64_t len = ph->len, size = ph->bot - ph->top;
                if (len <= m) continue;
                if (size == 0) continue;
                uint64_t nelt = size < maxG ? size : maxG;       /* getGenomeIdx :592-593 */
                if (nelt > ihits) continue;                      /* :299 (resolution result unused) */
                if (nelt > refs_cap) { refs_cap = nelt * 2; refs = realloc(refs, 8 * refs_cap); }
                uint32_t nid = 0;
                if (g_ops) { g_ops->n_ranges++; g_ops->n_rows += nelt; }
                for (uint64_t e = 0; e < nelt; e++, cnt++) {     /* :305-326 */
                    uint64_t ref = cfo_resolve_row(ix, ph->top + e);
                    int found = 0;
                    for (uint32_t q = 0; q < nid && !found; q++) found = refs[q] == ref;
                    if (!found) refs[nid++] = ref;
                }
                uint32_t sc = (uint32_t)((len - 15) * (len - 15));             /* :332 */
                for (uint32_t q = 0; q < nid; q++) {
                    uint64_t ref = refs[q];
                    uint64_t tax = ref < ix->nref ? ix->uid_tid[ref] : 0;
                    if (in_closure(ix, tax, pr->exclude_taxids, pr->n_exclude)) continue;   /* :339 */
                    add_hit(ix, pr, &hm, rdi, fwi, ref, tax, ts, sc, len);
                }
                if (cnt >= maxG) break;                          /* :366 */
            }
        }
        free(buf);
    }
    free(refs);
    /* finalize :86-120, :380-382 */
    for (uint32_t i = 0; i < hm.n; i++) {
        hcount *h = &hm.e[i];
#define MAXU(a, b) ((a) > (b) ? (a) : (b))
        if (paired) { h->score = MAXU(h->scores[0][0], h->scores[0][1]) + MAXU(h->scores[1][0], h->scores[1][1]);
                      h->hitLen = MAXU(h->hitLens[0][0], h->hitLens[0][1]) + MAXU(h->hitLens[1][0], h->hitLens[1][1]); }
        else        { h->score = MAXU(h->scores[0][0], h->scores[0][1]); h->hitLen = MAXU(h->hitLens[0][0], h->hitLens[0][1]); }
    }
    /* host logic :385-394 */
    int64_t best = 0; int only_host = 0;
    for (uint32_t i = 0; i < hm.n; i++) {
        if ((int64_t)hm.e[i].score > best) { best = hm.e[i].score; only_host = in_closure(ix, hm.e[i].taxID, pr->host_taxids, pr->n_host); }
        else if ((int64_t)hm.e[i].score == best) only_host |= in_closure(ix, hm.e[i].taxID, pr->host_taxids, pr->n_host);
    }
    if (!only_host && hm.n > k) {                                /* :399-515 */
        uint32_t bs = hm.e[0].score;
        for (uint32_t i = 1; i < hm.n; i++) if (bs < hm.e[i].score) bs = hm.e[i].score;
        for (int i = 0; i < (int)hm.n; i++) {                    /* :409-417 */
            if (hm.e[i].score < bs) { if (i + 1 < (int)hm.n) hm.e[i] = hm.e[hm.n - 1]; hm.n--; i--; }
        }
        if (!pr->tree_traverse && hm.n > k) goto done;           /* :419-425 unclassified */
        uint32_t rank = 0;                                       /* uint8_t in the reference; never wraps */
        typedef struct { uint32_t cnt; uint64_t tid; } tc;
        tc *tcs = malloc(sizeof(tc) * (hm.n + 1));
        while (hm.n > k) {                                       /* :428-514 */
            uint32_t ntc = 0;
            for (uint32_t i = 0; i < hm.n; i++) {
                hcount *h = &hm.e[i];
                while (h->rank < rank) {
                    if ((uint32_t)h->rank + 1 >= h->pathlen) { h->rank = 255; break; }
                    h->rank += 1; h->taxID = h->path[h->rank]; h->leaf = 0;
                }
                if (h->rank > rank) continue;
                uint64_t parent = (rank + 1 >= h->pathlen) ? 1 : h->path[rank + 1];
                if (parent == 0) continue;
                uint32_t j = 0;
                for (; j < ntc; j++) if (tcs[j].tid == parent) { tcs[j].cnt++; break; }
                if (j == ntc) { tcs[ntc].cnt = 1; tcs[ntc].tid = parent; ntc++; }
            }
            if (ntc == 0) {
                if (rank < hm.e[0].pathlen) { rank++; continue; } else break;
            }
            for (uint32_t a = 1; a < ntc; a++) {                 /* sort (count, taxid) ascending, :467 */
                tc v = tcs[a]; uint32_t b = a;
                while (b > 0 && (tcs[b - 1].cnt > v.cnt || (tcs[b - 1].cnt == v.cnt && tcs[b - 1].tid > v.tid))) { tcs[b] = tcs[b - 1]; b--; }
                tcs[b] = v;
            }
            uint32_t j = ntc;
            while (j-- > 0) {
                uint64_t parent = tcs[j].tid;
                for (uint32_t i = 0; i < hm.n; i++) {
                    hcount *h = &hm.e[i];
                    if (h->rank != rank) continue;
                    uint64_t cp = (rank + 1 >= h->pathlen) ? 1 : h->path[rank + 1];
                    if (parent == cp) { h->uniqueID = MASK64; h->rank = (uint8_t)(rank + 1); h->taxID = parent; h->leaf = 0; }
                }
                int first = 1; uint32_t rep = hm.n;
                for (uint32_t i = 0; i < hm.n; i++) {            /* :489-506 */
                    if (parent == hm.e[i].taxID) {
                        if (!first) {
                            hm.e[rep].num_leaves += hm.e[i].num_leaves;
                            if (i + 1 < hm.n) hm.e[i] = hm.e[hm.n - 1];
                            hm.n--; i--;
                        } else { first = 0; rep = i; }
                    }
                }
                if (hm.n <= k) break;
            }
            ++rank;
            if (rank > hm.e[0].pathlen) break;
        }
        free(tcs);
    }
    if (!only_host && hm.n > k) goto done;                       /* :516-520 unclassified */
    {
        /* emit :537-565, then AlnSinkWrap::finishRead → selectByScore aln_sink.h:1860-1927 */
        uint32_t nres = 0; uint32_t *res = malloc(4 * (hm.n + 1));
        for (uint32_t i = 0; i < hm.n; i++) {
            if (only_host && !in_closure(ix, hm.e[i].taxID, pr->host_taxids, pr->n_host)) continue;
            res[nres++] = i;
        }
        if (nres > 0) {
            sbuf *b = malloc(sizeof(sbuf) * nres);
            for (uint32_t i = 0; i < nres; i++) { b[i].score = hm.e[res[i]].score; b[i].idx = i; }
            for (uint32_t a = 1; a < nres; a++) {                /* ascending (score, idx) */
                sbuf v = b[a]; uint32_t c = a;
                while (c > 0 && (b[c - 1].score > v.score || (b[c - 1].score == v.score && b[c - 1].idx > v.idx))) { b[c] = b[c - 1]; c--; }
                b[c] = v;
            }
            for (uint32_t a = 0; a < nres / 2; a++) { sbuf t = b[a]; b[a] = b[nres - 1 - a]; b[nres - 1 - a] = t; }   /* reverse */
            /* shuffle tie streaks, ds.h:784-795 */
            uint32_t streak = 0;
            for (uint32_t i = 1; i <= nres; i++) {
                if (i < nres && b[i].score == b[i - 1].score) { if (streak == 0) streak = 1; streak++; }
                else {
                    if (streak > 1) {
                        uint32_t begin = i - streak, left = streak;
                        for (uint32_t q = begin; q < begin + streak - 1; q++) {
                            uint32_t r = lcg_next(&rnd) % left;
                            if (r > 0) { sbuf t = b[q]; b[q] = b[q + r]; b[q + r] = t; }
                            left--;
                        }
                    }
                    streak = 0;
                }
            }
            uint32_t num = nres < k ? nres : (uint32_t)k;         /* getReport aln_sink.h:2442-2458 */
            for (uint32_t i = 0; i + 1 < num; i++) if (b[i].score != b[i + 1].score) { num = i + 1; break; }
            /* 2ndBest: aligner_result.h:398-431 over ALL results */
            int64_t bst = INT64_MIN, sec = INT64_MIN;
            for (uint32_t i = 0; i < nres; i++) {
                int64_t sc = hm.e[res[i]].score;
                if (sc > bst) { sec = bst; bst = sc; } else if (sc > sec) sec = sc;
            }
            *score2 = sec == INT64_MIN ? 0 : (uint32_t)sec;
            for (uint32_t i = 0; i < num; i++) {
                const hcount *h = &hm.e[res[b[i].idx]];
                rows[i].tax_id = h->taxID;
                rows[i].unique_id = (h->uniqueID < ix->nref) ? (uint32_t)h->uniqueID : CFO_MERGED;
                rows[i].score = h->score; rows[i].hit_len = (uint32_t)h->hitLen; rows[i].pad = 0;
            }
            *n_rows = num;
            free(b);
        }
        free(res);
    }
done:
    free(hm.e);
    return rc;
}

int cfo_mate_passes(const uint8_t *seq, uint64_t len) {
    /* lenfilt centrifuge.cpp:2562-2577 (multiseedMms = 0) */
    if (len < 2) return 0;
    /* nFilter scoring.cpp:104-117 with nCeil = 0 + 0.15f * len (scoring.h:61-63) */
    uint64_t maxns = (uint64_t)(0.0 + (double)0.15f * (double)len), ns = 0;
    for (uint64_t i = 0; i < len; i++) if (seq[i] == 4) { if (++ns > maxns) return 0; }
    return 1;
}

int cfo_classify(const cfo_index *ix, const cfo_params *pr,
                 const uint8_t *seq, const uint64_t *off, const uint32_t *seeds,
                 uint64_t nq, int paired, cfo_row *rows, uint32_t *n_rows, uint32_t *score2, cfo_opcounts *ops) {
    g_ops = ops;
    for (uint64_t q = 0; q < nq; q++) {
        const uint8_t *s[2] = { 0, 0 }; uint64_t l[2] = { 0, 0 }; int pass[2] = { 0, 0 };
        uint64_t r0 = paired ? 2 * q : q;
        s[0] = seq + off[r0]; l[0] = off[r0 + 1] - off[r0]; pass[0] = cfo_mate_passes(s[0], l[0]);
        if (paired) { s[1] = seq + off[r0 + 1]; l[1] = off[r0 + 2] - off[r0 + 1]; pass[1] = cfo_mate_passes(s[1], l[1]); }
        classify_query(ix, pr, s, l, pass, paired, seeds[r0], paired ? seeds[r0 + 1] : 0,
                       rows + q * (uint64_t)pr->khits, &n_rows[q], &score2[q]);
        if (ops) ops->n_reads += paired ? 2 : 1;
    }
    g_ops = NULL;
    return 0;
}

int cfo_search(const cfo_index *ix, const cfo_params *pr, const uint8_t *seq, uint64_t L,
               cfo_hit *hf, cfo_hit *hr, uint32_t nh[2]) {
    const uint64_t k = (uint64_t)pr->khits, m = (uint64_t)pr->min_hitlen;
    const uint64_t ihits = (k > 5 ? k : 5) * (ix->compressed ? 4 : 40);
    const uint64_t inc = (2 * m <= 33) ? 10 : (2 * m - 33);
    hit_t *buf = malloc(sizeof(hit_t) * (2 * (L + 2) + 4));
    strand_t st[2] = { { 0, 0, 0, buf }, { 0, 0, 0, buf + L + 2 } };
    search_fw_rc(ix, seq, L, m, inc, ihits, st, buf + 2 * (L + 2));
    cfo_hit *out[2] = { hf, hr };
    for (int f = 0; f < 2; f++) {
        nh[f] = st[f].n;
        for (uint32_t i = 0; i < st[f].n; i++) {
            out[f][i].top = st[f].h[i].top; out[f][i].bot = st[f].h[i].bot;
            out[f][i].bwoff = (uint32_t)st[f].h[i].bwoff; out[f][i].len = (uint32_t)st[f].h[i].len;
        }
    }
    free(buf);
    return 0;
}

/* pat.h:55-91 genRandSeed; 32-bit wrap-around arithmetic on `int` shifts */
uint32_t cfo_gen_rand_seed(const uint8_t *seq, const uint8_t *qual, uint64_t len,
                           const char *name, uint64_t namelen, uint32_t seed) {
    uint32_t rseed = (seed + 101u) * 59u * 61u * 67u * 71u * 73u * 79u * 83u;
    for (uint64_t i = 0; i < len; i++) rseed ^= ((uint32_t)seq[i] << ((i & 15) << 1));
    for (uint64_t i = 0; i < len; i++) rseed ^= ((uint32_t)(qual ? qual[i] : 'I') << ((i & 3) << 3));
    for (uint64_t i = 0; i < namelen; i++) {
        int p = (unsigned char)name[i];       /* BTString is char-based: see note */
        if (p == '/') break;
        rseed ^= ((uint32_t)p << ((i & 3) << 3));
    }
    return rseed;
}
