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

    seq_tid, nodes = synth.write_random_taxonomy(d, rng, n_clusters, per)
    try:
        O.ref_build(d, threads=2)
    except subprocess.CalledProcessError:
        subprocess.run(["rm", "-rf", d]); continue              # the reference builder refused the input: nothing to compare
    rl = int(rng.choice([60, 100, 150]))
    paired = bool(rng.random() < 0.3)
    if paired:
        (nm, s1), (_, s2) = synth.sample_reads(g, 120, min(rl, L // 4), paired=True, random_frac=0.05, n_frac=0.05, seed=int(rng.integers(1 << 30)))
        synth.write_fasta(d + "/r1.fa", nm, s1, "/1"); synth.write_fasta(d + "/r2.fa", nm, s2, "/2"); files = [d + "/r1.fa", d + "/r2.fa"]
    else:
        nm, s = synth.sample_reads(g, 120, min(rl, L // 2), random_frac=0.05, n_frac=0.05, seed=int(rng.integers(1 << 30)))
        if rng.random() < 0.5:                                  # ragged lengths (trimmed reads)
            s = [x[:int(rng.integers(20, len(x) + 1))] for x in s]
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
        want = O.ref_classify(d + "/idx", d + "/w.tsv", d + "/w.rep", extra=a, **(dict(m1=files[0], m2=files[1]) if paired else dict(u=files[0])))
    except subprocess.CalledProcessError:
        subprocess.run(["rm", "-rf", d]); continue
    e = emu.Emu(d + "/idx")
    names, ql, seq, off, seeds, pr = reads.load(files, False)
    got = None
    for fp, fs in ((1, 1), (0, 0)):                           # the common-case kernels in front, then the general ones alone
        emu.lib().emu_set_search_version(2)
        emu.lib().emu_set_fast_kernels(fp, fs)
        rows, n_rows, s2_ = e.classify(seq, off, seeds, paired=pr, **kw)
        got = reads.format_tsv(e.seqid, names, ql, rows, n_rows, s2_)
        if got == want and fp:                                # the report (names, ranks, counters, EM) from the same rows
            hix = capi.Index(d + "/idx", host_only=True); rep = capi.Report(hix); orc = O.Oracle(d + "/idx")
            rep.add(rows, n_rows, TR.max_scores(orc, seq, off, pr), kw["k"]); rep.write(d + "/m.rep"); rep.close(); hix.close()
            if open(d + "/m.rep").read() != open(d + "/w.rep").read():
                got = "REPORT DIFFERS"
                print(common.first_diff(open(d + "/m.rep").read(), open(d + "/w.rep").read()), flush=True)
        if got != want:
            bad += 1
            print("MISMATCH iter", it - 1, "seed", seed0 + it - 1, "fast", fp, fs, kw, "clusters,per,L,div", n_clusters, per, L, div, "paired", paired, d, flush=True)
            print(common.first_diff(got, want), flush=True)
            break
    e.close()
    if got == want:
        subprocess.run(["rm", "-rf", d])
emu.lib().emu_set_fast_kernels(1, 1)
print("iterations", it, "bad", bad)
