#!/usr/bin/env python3
"""Regenerates the golden fixtures under tests/golden/ with the compiled,
unmodified reference (oracle/_ref, built by oracle/Makefile).

Run in the build container (needs /root/reference for the worked example and
oracle/_ref/centrifuge-{build-bin,class}).  Output: two small archives,

  example.tar.xz      the reference's worked example (example/ + MANUAL:1012-1028):
                      rebuilt index, the 16 reads, and the reference's TSV/report
                      for a set of option variants;
  synth_small.tar.xz  a 24-genome synthetic index with repeats / low-complexity /
                      N-containing genomes, single-end, paired-end and FASTQ
                      reads, and the reference's outputs for k = 1, 5, genus
                      rank, host / exclude filters.

Each archive holds `cases.json`: [{name, args, reads: [files], tsv, report}].
The GPU tests and smoke() read only these archives (never /root/reference).
"""
import io
import json
import os
import shutil
import subprocess
import sys
import tarfile
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import synth  # noqa: E402
from oracle import oracle as O  # noqa: E402

REF = os.path.join(ROOT, "oracle", "_ref")


def run_case(d, idx, name, args, reads, cases):
    tsv, rep = os.path.join(d, name + ".tsv"), os.path.join(d, name + ".report.tsv")
    cmd = [os.path.join(REF, "centrifuge-class"), "-p", "1", "-x", idx, "-S", tsv, "--report-file", rep] + args
    if len(reads) == 1:
        cmd += ["-U", os.path.join(d, reads[0])]
    else:
        cmd += ["-1", os.path.join(d, reads[0]), "-2", os.path.join(d, reads[1])]
    subprocess.run(cmd, check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    cases.append({"name": name, "args": args, "reads": reads, "tsv": name + ".tsv", "report": name + ".report.tsv"})


def pack(d, out):
    with tarfile.open(out, "w:xz", preset=9) as t:
        for f in sorted(os.listdir(d)):
            t.add(os.path.join(d, f), arcname=f)
    print(out, os.path.getsize(out), "bytes")


def make_example():
    ex = "/root/reference/example"
    d = tempfile.mkdtemp()
    subprocess.run([os.path.join(REF, "centrifuge-build-bin"), "--conversion-table", ex + "/reference/gi_to_tid.dmp",
                    "--taxonomy-tree", ex + "/reference/nodes.dmp", "--name-table", ex + "/reference/names.dmp",
                    ex + "/reference/test.fa", os.path.join(d, "idx")], check=True, stdout=subprocess.DEVNULL,
                   stderr=subprocess.DEVNULL)
    shutil.copy(ex + "/reads/input.fa", os.path.join(d, "reads.fa"))
    # extra reads cut from the example genomes: lengths around the 22-mer rule, junk flanks, mates
    seqs = {}
    name = None
    for ln in open(ex + "/reference/test.fa"):
        if ln.startswith(">"):
            name = ln[1:].split()[0]
            seqs[name] = ""
        else:
            seqs[name] += ln.strip()
    g4 = seqs[[k for k in seqs if k.startswith("gi|4")][0]]
    g7 = seqs[[k for k in seqs if k.startswith("gi|7")][0]]
    with open(os.path.join(d, "len.fa"), "w") as f:
        for L in (15, 21, 22, 23, 24, 31, 40):
            f.write(">len%d\n%s\n" % (L, g4[100:100 + L]))
        f.write(">junkL\n%s%s\n" % ("ACGTTGCAAGGCTTAACCGG", g4[200:223]))
        f.write(">junkR\n%s%s\n" % (g4[200:223], "ACGTTGCAAGGCTTAACCGG"))
        f.write(">allN\n%s\n>rand\nACGATCGATCGGGATATTAGCGCGATATCGGCGATATAGCTAGCTAGGGATCTCTAGAGA\n" % ("N" * 50))
        f.write(">iupac\n%sRYKM%s\n" % (g7[50:90], g7[94:130]))
    comp = str.maketrans("ACGTN", "TGCAN")
    with open(os.path.join(d, "m1.fa"), "w") as f1, open(os.path.join(d, "m2.fa"), "w") as f2:
        f1.write(">p1/1\n%s\n" % g4[10:90]); f2.write(">p1/2\n%s\n" % g4[250:330].translate(comp)[::-1])
        f1.write(">p2/1\n%s\n" % g7[10:90]); f2.write(">p2/2\n%s\n" % ("N" * 80))
        f1.write(">p3/1\n%s\n" % ("N" * 50)); f2.write(">p3/2\n%s\n" % ("N" * 60))
        f1.write(">p4/1\n%s\n" % ("N" * 50)); f2.write(">p4/2\n%s\n" % g7[300:380])
    cases = []
    idx = os.path.join(d, "idx")
    for nm, args in [("default", ["-f"]), ("k1", ["-f", "-k", "1"]), ("k1_notrav", ["-f", "-k", "1", "--no-traverse"]),
                     ("genus", ["-f", "--classification-rank", "genus"]),
                     ("species", ["-f", "--classification-rank", "species"]),
                     ("host", ["-f", "--host-taxids", "9913"]), ("exclude", ["-f", "--exclude-taxids", "9913"]),
                     ("minhit79", ["-f", "--min-hitlen", "79"]), ("minhit80", ["-f", "--min-hitlen", "80"])]:
        run_case(d, idx, nm, args, ["reads.fa"], cases)
    run_case(d, idx, "lengths", ["-f"], ["len.fa"], cases)
    run_case(d, idx, "pairs", ["-f"], ["m1.fa", "m2.fa"], cases)
    json.dump(cases, open(os.path.join(d, "cases.json"), "w"), indent=1)
    pack(d, os.path.join(HERE, "example.tar.xz"))
    shutil.rmtree(d)


def make_synth_small():
    d = tempfile.mkdtemp()
    rng = np.random.default_rng(5)
    G, L = 24, 6000
    g = synth.make_genomes(G, L, genus_size=8, divergence=0.01, seed=7)
    rep = synth.ACGT[rng.integers(0, 4, 300, dtype=np.uint8)]
    rep2 = synth.ACGT[rng.integers(0, 4, 60, dtype=np.uint8)]
    for i in range(G):
        for _ in range(10):
            p = int(rng.integers(0, L - 300)); g[i, p:p + 300] = rep
        for _ in range(3):
            p = int(rng.integers(0, L - 60)); g[i, p:p + 60] = rep2
        p = int(rng.integers(0, L - 200)); g[i, p:p + 200] = ord("A")
        p = int(rng.integers(0, L - 200)); g[i, p:p + 200] = np.frombuffer(b"AC" * 100, dtype=np.uint8)
    synth.write_reference(d, g, n_in_genomes=2)
    O.ref_build(d, threads=1)
    names, seqs = synth.sample_reads(g, 1500, 100, random_frac=0.03, n_frac=0.15, seed=11)

    def add(nm, s):
        names.append(nm); seqs.append(s)
    add("polyA", b"A" * 100); add("polyAC", b"AC" * 50); add("rep", rep[:100].tobytes())
    add("rep2", rep2.tobytes() + rep[:40].tobytes()); add("allN", b"N" * 100)
    add("n15", b"N" * 15 + g[0, :85].tobytes()); add("n16", b"N" * 16 + g[0, :84].tobytes())
    for Lr in (1, 2, 9, 10, 11, 22, 23, 33):
        add("len%d" % Lr, g[3, 500:500 + Lr].tobytes())
    for j in range(100):
        r = g[int(rng.integers(0, G)), 100 + j * 50: 200 + j * 50].copy()
        r[rng.integers(0, 100, int(rng.integers(1, 16)))] = ord("N"); add("nrich%d" % j, r.tobytes())
    for j in range(100):
        a = g[int(rng.integers(0, G)), 200 + j * 30: 250 + j * 30]
        b = synth.COMP[g[int(rng.integers(0, G)), 1000 + j * 30: 1050 + j * 30][::-1]]
        add("chim%d" % j, a.tobytes() + b.tobytes())
    for j in range(100):
        a = g[int(rng.integers(0, G)), 300 + j * 30: 360 + j * 30]
        add("pal%d" % j, a.tobytes() + synth.COMP[a[::-1]][10:50].tobytes())
    synth.write_fasta(os.path.join(d, "reads.fa"), names, seqs)
    n250, s250 = synth.sample_reads(g, 300, 250, random_frac=0.03, n_frac=0.5, seed=12)
    for j in range(100):
        r = g[int(rng.integers(0, G)), 100 + j * 20: 350 + j * 20].copy()
        r[rng.integers(0, 250, int(rng.integers(10, 38)))] = ord("N"); n250.append("nr%d" % j); s250.append(r.tobytes())
    synth.write_fasta(os.path.join(d, "reads250.fa"), n250, s250)
    (n, s1), (_, s2) = synth.sample_reads(g, 400, 150, paired=True, random_frac=0.03, n_frac=0.3, seed=4)
    s2 = [(b"N" * 150 if i % 40 == 0 else x) for i, x in enumerate(s2)]
    s1 = [(b"N" * 150 if i % 55 == 0 else x) for i, x in enumerate(s1)]
    synth.write_fasta(os.path.join(d, "r1.fa"), n, s1, "/1")
    synth.write_fasta(os.path.join(d, "r2.fa"), n, s2, "/2")
    synth.write_fastq(os.path.join(d, "reads.fq"), *synth.sample_reads(g, 300, 100, seed=9))
    for f in ("genomes.fa", "conv.tsv", "nodes.dmp", "names.dmp"):
        os.remove(os.path.join(d, f))
    cases = []
    idx = os.path.join(d, "idx")
    for nm, args in [("k5", ["-f"]), ("k1", ["-f", "-k", "1"]), ("k2", ["-f", "-k", "2"]), ("k50", ["-f", "-k", "50"]),
                     ("genus", ["-f", "--classification-rank", "genus"]),
                     ("family_k1", ["-f", "--classification-rank", "family", "-k", "1"]),
                     ("host", ["-f", "--host-taxids", "1005,102"]), ("exclude", ["-f", "--exclude-taxids", "1005,102"]),
                     ("minhit15", ["-f", "--min-hitlen", "15"]), ("k1_notrav", ["-f", "-k", "1", "--no-traverse"])]:
        run_case(d, idx, nm, args, ["reads.fa"], cases)
    run_case(d, idx, "r250_k5", ["-f"], ["reads250.fa"], cases)
    run_case(d, idx, "r250_k1", ["-f", "-k", "1"], ["reads250.fa"], cases)
    run_case(d, idx, "pe_k5", ["-f"], ["r1.fa", "r2.fa"], cases)
    run_case(d, idx, "pe_k1", ["-f", "-k", "1"], ["r1.fa", "r2.fa"], cases)
    run_case(d, idx, "fastq", ["-q"], ["reads.fq"], cases)
    json.dump(cases, open(os.path.join(d, "cases.json"), "w"), indent=1)
    pack(d, os.path.join(HERE, "synth_small.tar.xz"))
    shutil.rmtree(d)


if __name__ == "__main__":
    if not O.have_ref():
        sys.exit("oracle/_ref is not built (make -C oracle ref)")
    make_example()
    make_synth_small()
