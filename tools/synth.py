#!/usr/bin/env python3
"""Deterministic synthetic inputs for parity tests and the benchmark.

Recipe follows SURVEY.md §8(d): G genomes x L bp of uniform-random ACGT in
"genera" of `genus_size` members that are one ancestor with `divergence`
substitutions (creates multi-genome hits, ties and >k cases); taxonomy
root(1) -> genus(100+g) -> species(1000+i); uid `seq<i>` (or `cid|<i>` for the
"compressed" variant, bt2_idx.h:648-663).  Reads are sampled uniformly over
genomes and strands, a fraction carries one substitution, plus optional random
(unclassifiable) reads and reads containing N runs.

Nothing here reads /root/reference; the files it writes are the inputs of the
reference's own `centrifuge-build` (FASTA + conversion table + nodes/names.dmp).
"""
import argparse
import os
import numpy as np

ACGT = np.frombuffer(b"ACGT", dtype=np.uint8)
COMP = np.zeros(256, dtype=np.uint8)
for a, b in zip(b"ACGTN", b"TGCAN"):
    COMP[a] = b


def make_genomes(n_genomes, length, genus_size=8, divergence=0.05, seed=12345):
    """Returns uint8 array [n_genomes, length] of ASCII bases."""
    rng = np.random.default_rng(seed)
    out = np.empty((n_genomes, length), dtype=np.uint8)
    for g0 in range(0, n_genomes, genus_size):
        anc = rng.integers(0, 4, size=length, dtype=np.uint8)
        for i in range(g0, min(g0 + genus_size, n_genomes)):
            codes = anc.copy()
            mut = rng.random(length) < divergence
            codes[mut] = (codes[mut] + rng.integers(1, 4, size=int(mut.sum()), dtype=np.uint8)) & 3
            out[i] = ACGT[codes]
    return out



def make_repeat_genomes(n_genomes, length, seed=5, n_operons=8, operon_len=5000):
    """Repeat-rich stand-in, as ASCII like make_genomes: genera of 8 whose members come in clusters of 4 near-identical
    strains (0.4 - 1 % apart), `n_operons` shared stretches of `operon_len` bases pasted into a third of the genomes
    each, and low-complexity tracts (homopolymers, dinucleotide repeats of 50-400 bases) over 0.5 % of every genome:
    suffix ties hundreds to thousands of bases deep (what real bacterial collections look like to a suffix sorter)."""
    rng = np.random.default_rng(seed)
    g = np.empty((n_genomes, length), dtype=np.uint8)
    for g0 in range(0, n_genomes, 8):
        m = min(8, n_genomes - g0)
        anc = rng.integers(0, 4, length, dtype=np.uint8)
        for j in range(m):
            if j % 4 == 0:
                mut = rng.random(length) < 0.05
                g[g0 + j] = (anc + mut * rng.integers(1, 4, length, dtype=np.uint8)) & 3
            else:
                mut = rng.random(length) < 0.001 * (1 + 3 * (j % 4))
                g[g0 + j] = (g[g0 + j - j % 4] + mut * rng.integers(1, 4, length, dtype=np.uint8)) & 3
    if length > operon_len + 1:
        ops = rng.integers(0, 4, (n_operons, operon_len), dtype=np.uint8)
        for o in range(n_operons):
            for _ in range(max(2, n_genomes // 3)):
                gi = int(rng.integers(0, n_genomes))
                p = int(rng.integers(0, length - operon_len))
                g[gi, p:p + operon_len] = ops[o]
    if length > 800:
        for gi in range(n_genomes):
            for _ in range(max(1, int(0.005 * length / 200))):
                p = int(rng.integers(0, length - 400))
                ln = int(rng.integers(50, 400))
                a, b = rng.integers(0, 4, 2)
                pat = np.full(ln, a, dtype=np.uint8)
                if rng.random() < 0.5:
                    pat[1::2] = b
                g[gi, p:p + ln] = pat
    return ACGT[g]


def write_reference(outdir, genomes, genus_size=8, uid_prefix="seq", line=80,
                    ranks=("genus", "species"), n_in_genomes=0, seed=99):
    """Writes genomes.fa, conv.tsv, nodes.dmp, names.dmp into outdir."""
    os.makedirs(outdir, exist_ok=True)
    n, L = genomes.shape
    rng = np.random.default_rng(seed)
    with open(os.path.join(outdir, "genomes.fa"), "wb") as f:
        for i in range(n):
            f.write(b">%s%d synthetic genome %d\n" % (uid_prefix.encode(), i, i))
            g = genomes[i]
            if n_in_genomes and i % 3 == 0:
                g = g.copy()                   # N stretches split the sequence into fragments
                for _ in range(n_in_genomes):
                    p = int(rng.integers(0, max(1, L - 40)))
                    g[p:p + int(rng.integers(1, 30))] = ord("N")
            body = g.tobytes()
            for p in range(0, L, line):
                f.write(body[p:p + line] + b"\n")
    write_taxonomy(outdir, n, genus_size, uid_prefix, ranks)


def write_taxonomy(outdir, n, genus_size=8, uid_prefix="seq", ranks=("genus", "species")):
    """conv.tsv, nodes.dmp, names.dmp for n genomes (no sequences)."""
    os.makedirs(outdir, exist_ok=True)
    with open(os.path.join(outdir, "conv.tsv"), "w") as f:
        for i in range(n):
            f.write("%s%d\t%d\n" % (uid_prefix, i, 1000 + i))
    ngen = (n + genus_size - 1) // genus_size
    with open(os.path.join(outdir, "nodes.dmp"), "w") as f:
        f.write("1\t|\t1\t|\tno rank\n")
        f.write("2\t|\t1\t|\tsuperkingdom\n")
        f.write("50\t|\t2\t|\tfamily\n")
        for g in range(ngen):
            f.write("%d\t|\t50\t|\t%s\n" % (100 + g, ranks[0]))
        for i in range(n):
            f.write("%d\t|\t%d\t|\t%s\n" % (1000 + i, 100 + i // genus_size, ranks[1]))
    with open(os.path.join(outdir, "names.dmp"), "w") as f:
        f.write("1\t|\troot\t|\t\t|\tscientific name\t|\n")
        f.write("2\t|\tBacteria\t|\t\t|\tscientific name\t|\n")
        f.write("50\t|\tSynthaceae\t|\t\t|\tscientific name\t|\n")
        for g in range(ngen):
            f.write("%d\t|\tSynthus%d\t|\t\t|\tscientific name\t|\n" % (100 + g, g))
        for i in range(n):
            f.write("%d\t|\tSynthus%d species%d\t|\t\t|\tscientific name\t|\n" % (1000 + i, i // genus_size, i))


# the NCBI rank vocabulary, top down; a random lineage keeps this order, leaves ranks out and puts "no rank" / unknown strings between
RANKS_TOP_DOWN = ["superkingdom", "kingdom", "subkingdom", "superphylum", "phylum", "subphylum", "superclass", "class", "subclass",
                  "infraclass", "superorder", "order", "suborder", "infraorder", "parvorder", "superfamily", "family", "subfamily",
                  "tribe", "subtribe", "genus", "subgenus", "species group", "species subgroup", "species", "subspecies", "varietas",
                  "forma", "strain"]
RANKS_ODD = ["no rank", "no rank", "clade", "domain", "life", "serotype", ""]


def write_random_taxonomy(outdir, rng, n_clusters, per, uid_prefix="seq"):
    """conv.tsv, nodes.dmp, names.dmp of a RANDOM tree for n_clusters x per sequences (uids <prefix><i>, cluster = i // per):
    lineages of random depth over the whole rank vocabulary, sequences on leaves, on inner nodes, several on one node, on taxIDs
    the tree does not know, taxIDs beyond 32 bits, names for some nodes only, now and then a sequence missing from the
    conversion table.  Returns (taxID per sequence, {taxID: (parent, rank)})."""
    used = {0, 1}

    def new_id():
        while True:
            t = int(rng.integers(2, 3000000)) if rng.random() < 0.93 else int(rng.integers(1 << 32, 1 << 40))
            if t not in used:
                used.add(t)
                return t
    nodes = {1: (1, "no rank")}                               # tid -> (parent, rank)

    def lineage(parent, lo, hi, p_keep):
        """a chain of nodes under `parent` through RANKS_TOP_DOWN[lo:hi]; returns the chain (top first)"""
        chain = []
        for r in RANKS_TOP_DOWN[lo:hi]:
            if rng.random() < 0.12:
                t = new_id(); nodes[t] = (parent, str(rng.choice(RANKS_ODD))); parent = t; chain.append(t)
            if rng.random() < p_keep:
                t = new_id(); nodes[t] = (parent, r); parent = t; chain.append(t)
        return chain
    top = lineage(1, 0, int(rng.integers(0, 12)), float(rng.choice([0.2, 0.5, 0.9])))       # what all clusters share
    top_end = top[-1] if top else 1
    seq_tid = []
    for c in range(n_clusters):
        split = int(rng.integers(8, 24))
        mid = lineage(top_end, min(split, 12), int(rng.integers(20, 26)), float(rng.choice([0.3, 0.6, 0.95])))
        anchor_pool = [top_end] + mid
        for i in range(per):
            how = rng.random()
            if how < 0.55:                                    # its own leaf under the cluster's lineage (any depth below the anchor)
                tail = lineage(anchor_pool[-1], int(rng.integers(22, 27)), len(RANKS_TOP_DOWN), 0.5)
                if not tail:
                    t = new_id(); nodes[t] = (anchor_pool[-1], str(rng.choice(["species", "strain", "no rank", "subspecies"]))); tail = [t]
                seq_tid.append(tail[-1])
            elif how < 0.75:                                  # an inner node of the lineage (a genome filed under its genus, say)
                seq_tid.append(int(rng.choice(anchor_pool)))
            elif how < 0.9 and seq_tid:                       # the node another sequence already sits on
                seq_tid.append(int(rng.choice(seq_tid)))
            else:                                             # a taxID the tree does not know
                seq_tid.append(new_id())
    if rng.random() < 0.2:                                    # unused branches beside the used ones (pruned by the builder)
        lineage(1, 0, 10, 0.5)
    with open(os.path.join(outdir, "conv.tsv"), "w") as f:
        for i, t in enumerate(seq_tid):
            if rng.random() < 0.97:                           # (a sequence missing from the table: as the builder decides)
                f.write("%s%d\t%d\n" % (uid_prefix, i, t))
    with open(os.path.join(outdir, "nodes.dmp"), "w") as f:
        for t, (p, r) in nodes.items():
            f.write("%d\t|\t%d\t|\t%s\t|\n" % (t, p, r))
    with open(os.path.join(outdir, "names.dmp"), "w") as f:
        for t in nodes:
            if rng.random() < 0.8:
                f.write("%d\t|\tname of %d\t|\t\t|\tscientific name\t|\n" % (t, t))
    return seq_tid, nodes


def sample_reads(genomes, n_reads, read_len=100, mut_frac=0.63, random_frac=0.01,
                 n_frac=0.001, seed=777, paired=False, frag=(250, 400)):
    """Returns (names, seqs) or for paired ((names, seqs1), (names, seqs2)); seqs are bytes."""
    rng = np.random.default_rng(seed)
    G, L = genomes.shape
    names, s1, s2 = [], [], []
    gi = rng.integers(0, G, size=n_reads)
    strand = rng.integers(0, 2, size=n_reads)
    kind = rng.random(n_reads)
    for i in range(n_reads):
        g = int(gi[i])
        if paired:
            fl = int(rng.integers(frag[0], frag[1] + 1))
            fl = max(fl, read_len)
            p = int(rng.integers(0, L - fl + 1))
            fragseq = genomes[g, p:p + fl]
            if strand[i]:
                fragseq = COMP[fragseq[::-1]]
            m1 = fragseq[:read_len].copy()
            m2 = COMP[fragseq[::-1]][:read_len].copy()
            mates = [m1, m2]
        else:
            p = int(rng.integers(0, L - read_len + 1))
            r = genomes[g, p:p + read_len].copy()
            if strand[i]:
                r = COMP[r[::-1]]
            mates = [r]
        if kind[i] < random_frac:
            mates = [ACGT[rng.integers(0, 4, size=len(m), dtype=np.uint8)] for m in mates]
            tag = "rnd"
        else:
            tag = str(g)
            for m in mates:
                if rng.random() < mut_frac:
                    q = int(rng.integers(0, len(m)))
                    m[q] = ACGT[(np.searchsorted(ACGT, m[q]) + int(rng.integers(1, 4))) & 3]
                if rng.random() < n_frac:
                    q = int(rng.integers(0, len(m) - 3))
                    m[q:q + int(rng.integers(1, 4))] = ord("N")
        names.append("r%d_%s" % (i, tag))
        s1.append(mates[0].tobytes())
        if paired:
            s2.append(mates[1].tobytes())
    if paired:
        return (names, s1), (names, s2)
    return names, s1


def write_fasta(path, names, seqs, suffix=""):
    with open(path, "wb") as f:
        for n, s in zip(names, seqs):
            f.write(b">" + n.encode() + suffix.encode() + b"\n" + s + b"\n")


def write_fastq(path, names, seqs, suffix="", seed=5):
    rng = np.random.default_rng(seed)
    with open(path, "wb") as f:
        for n, s in zip(names, seqs):
            q = (rng.integers(2, 41, size=len(s), dtype=np.uint8) + 33).tobytes()
            f.write(b"@" + n.encode() + suffix.encode() + b"\n" + s + b"\n+\n" + q + b"\n")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("outdir")
    ap.add_argument("--genomes", type=int, default=64)
    ap.add_argument("--length", type=int, default=1000000)
    ap.add_argument("--genus-size", type=int, default=8)
    ap.add_argument("--divergence", type=float, default=0.05)
    ap.add_argument("--reads", type=int, default=100000)
    ap.add_argument("--read-len", type=int, default=100)
    ap.add_argument("--paired", action="store_true")
    ap.add_argument("--uid-prefix", default="seq")
    ap.add_argument("--seed", type=int, default=12345)
    a = ap.parse_args()
    g = make_genomes(a.genomes, a.length, a.genus_size, a.divergence, a.seed)
    write_reference(a.outdir, g, a.genus_size, a.uid_prefix)
    if a.reads:
        if a.paired:
            (n, s1), (_, s2) = sample_reads(g, a.reads, a.read_len, paired=True)
            write_fasta(os.path.join(a.outdir, "reads_1.fa"), n, s1, "/1")
            write_fasta(os.path.join(a.outdir, "reads_2.fa"), n, s2, "/2")
        else:
            n, s = sample_reads(g, a.reads, a.read_len)
            write_fasta(os.path.join(a.outdir, "reads.fa"), n, s)


if __name__ == "__main__":
    main()
