"""Offline fuzz (not collected by pytest): random TAXONOMIES under the classification path.  The other classify fuzzer keeps one
tree shape (root - superkingdom - family - genus - species); here the tree is random: lineages of random depth with ranks drawn
from the whole NCBI vocabulary (sub- / super- / infra- ranks, "no rank" in between, unknown strings), sequences attached to
leaves, to inner nodes, several to one node, to taxIDs that are not in the tree, taxIDs beyond 32 bits, names for some nodes
only — so the 10-slot paths (taxonomy.h:96-149), the key lifting of --classification-rank (classifier.h:1002-1010), the climb
(classifier.h:427-514), the seqID rule of the TSV (uid / rank string) and the report's names and ranks all see shapes the
synthetic recipe never makes.  Genomes come in clusters of relatives so that reads hit several of them and -k 1..3 forces
the climb.  The kernel bodies (tests/emu) and the host report against the compiled reference (oracle/_ref).
usage: fuzz_taxonomy.py <seconds> [seed0]"""
import os, sys, tempfile, time, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests')); sys.path.insert(0, os.path.join(ROOT, 'tools'))
import numpy as np
import synth, common
from centrifuge_amd import reads, capi
from oracle import oracle as O
from emu import emu
import test_report as TR

# top-down; a lineage keeps this order, leaves ranks out at random and puts "no rank" / unknown strings in between
ORDERED = ["superkingdom", "kingdom", "subkingdom", "superphylum", "phylum", "subphylum", "superclass", "class", "subclass",
           "infraclass", "superorder", "order", "suborder", "infraorder", "parvorder", "superfamily", "family", "subfamily",
           "tribe", "subtribe", "genus", "subgenus", "species group", "species subgroup", "species", "subspecies", "varietas",
           "forma", "strain"]
ODD = ["no rank", "no rank", "clade", "domain", "life", "serotype", ""]

t_end = time.time() + float(sys.argv[1])
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 0
it = 0; bad = 0
while time.time() < t_end:
    rng = np.random.default_rng(seed0 + it); it += 1
    n_clusters = int(rng.integers(1, 5)); per = int(rng.integers(2, 7)); G = n_clusters * per
    L = int(rng.integers(1200, 3500))
    div = float(rng.choice([0.0, 0.005, 0.02, 0.05]))
    d = tempfile.mkdtemp(prefix="fzt")
    g = synth.make_genomes(G, L, genus_size=per, divergence=div, seed=int(rng.integers(1 << 30)))
    synth.write_reference(d, g, genus_size=per)              # genomes.fa (uids seq<i>); its taxonomy files are replaced below

    used = {0, 1}
    def new_id():
        while True:
            t = int(rng.integers(2, 3000000)) if rng.random() < 0.93 else int(rng.integers(1 << 32, 1 << 40))
            if t not in used:
                used.add(t); return t
    nodes = {1: (1, "no rank")}                               # tid -> (parent, rank)
    def lineage(parent, lo, hi, p_keep):
        """a chain of nodes under `parent` through ORDERED[lo:hi]; returns the chain (top first)"""
        chain = []
        for r in ORDERED[lo:hi]:
            if rng.random() < 0.12:
                t = new_id(); nodes[t] = (parent, str(rng.choice(ODD))); parent = t; chain.append(t)
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
                tail = lineage(anchor_pool[-1], int(rng.integers(22, 27)), len(ORDERED), 0.5)
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
    with open(d + "/conv.tsv", "w") as f:
        for i, t in enumerate(seq_tid):
            if rng.random() < 0.97:                           # (a sequence missing from the table gets taxID 0 ... as the builder decides)
                f.write("seq%d\t%d\n" % (i, t))
    with open(d + "/nodes.dmp", "w") as f:
        for t, (p, r) in nodes.items():
            f.write("%d\t|\t%d\t|\t%s\t|\n" % (t, p, r))
    with open(d + "/names.dmp", "w") as f:
        for t in nodes:
            if rng.random() < 0.8:
                f.write("%d\t|\tname of %d\t|\t\t|\tscientific name\t|\n" % (t, t))
    try:
        O.ref_build(d, threads=2)
    except subprocess.CalledProcessError:
        subprocess.run(["rm", "-rf", d]); continue              # the reference builder refused the input: nothing to compare
    rl = int(rng.choice([60, 100, 150]))
    nm, s = synth.sample_reads(g, 120, min(rl, L // 2), random_frac=0.05, n_frac=0.05, seed=int(rng.integers(1 << 30)))
    synth.write_fasta(d + "/r.fa", nm, s); files = [d + "/r.fa"]
    kw = {"k": int(rng.choice([1, 1, 2, 3, 5, 20])), "min_hitlen": int(rng.choice([16, 22, 22, 30])),
          "rank": str(rng.choice(list(capi.RANK_SLOTS))), "traverse": bool(rng.random() < 0.8)}
    pool = sorted(set(seq_tid) | set(nodes))
    if rng.random() < 0.25: kw["host"] = [int(x) for x in rng.choice(pool, size=min(len(pool), int(rng.integers(1, 3))), replace=False)]
    if rng.random() < 0.25: kw["exclude"] = [int(x) for x in rng.choice(pool, size=min(len(pool), int(rng.integers(1, 3))), replace=False)]
    a = ["-k", str(kw["k"]), "--min-hitlen", str(kw["min_hitlen"]), "--classification-rank", kw["rank"]]
    if not kw["traverse"]: a.append("--no-traverse")
    if kw.get("host"): a += ["--host-taxids", ",".join(map(str, kw["host"]))]
    if kw.get("exclude"): a += ["--exclude-taxids", ",".join(map(str, kw["exclude"]))]
    try:
        want = O.ref_classify(d + "/idx", d + "/w.tsv", d + "/w.rep", extra=a, u=files[0])
    except subprocess.CalledProcessError:
        subprocess.run(["rm", "-rf", d]); continue
    e = emu.Emu(d + "/idx")
    names, ql, seq, off, seeds, pr = reads.load(files, False)
    got = None
    for fp, fs in ((1, 1), (0, 0)):                           # the common-case kernels in front, then the general ones alone
        emu.lib().emu_set_search_version(2)
        emu.lib().emu_set_fast_kernels(fp, fs)
        rows, n_rows, s2_ = e.classify(seq, off, seeds, paired=False, **kw)
        got = reads.format_tsv(e.seqid, names, ql, rows, n_rows, s2_)
        if got == want and fp:                                # the report (names, ranks, counters, EM) from the same rows
            hix = capi.Index(d + "/idx", host_only=True); rep = capi.Report(hix); orc = O.Oracle(d + "/idx")
            rep.add(rows, n_rows, TR.max_scores(orc, seq, off, pr), kw["k"]); rep.write(d + "/m.rep"); rep.close(); hix.close()
            if open(d + "/m.rep").read() != open(d + "/w.rep").read():
                got = "REPORT DIFFERS"
                print(common.first_diff(open(d + "/m.rep").read(), open(d + "/w.rep").read()), flush=True)
        if got != want:
            bad += 1
            print("MISMATCH iter", it - 1, "seed", seed0 + it - 1, "fast", fp, fs, kw, "clusters,per,L,div", n_clusters, per, L, div, d, flush=True)
            print(common.first_diff(got, want), flush=True)
            break
    e.close()
    if got == want:
        subprocess.run(["rm", "-rf", d])
emu.lib().emu_set_fast_kernels(1, 1)
print("iterations", it, "bad", bad)
