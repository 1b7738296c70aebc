#!/usr/bin/env python3
"""Golden outputs of the reference's index/report tools: centrifuge-inspect
(compiled reference, oracle/_ref/centrifuge-inspect-bin), centrifuge-kreport and centrifuge-promote (the reference's
Perl scripts, run from a scratch copy next to a shim `centrifuge-inspect`, because it looks for the
inspector in its own directory: centrifuge-kreport:23,235,245).

Run in the build container (needs /root/reference, perl, oracle/_ref).  Output: tools.tar.xz with

  gaps.{1,2,3,4}.cf   a tiny index whose sequences start / end with N runs, one all-N sequence,
                      one 30 bp sequence (fragment-table edge cases of the FASTA mode)
  inspect/<index>.<mode>.txt   stdout of every inspector mode on example, synth_small and gaps
  kreport/<index>.<case>.<variant>.txt   stdout of centrifuge-kreport over the golden TSVs
  promote/<index>.<case>.<level>.txt     stdout of centrifuge-promote (the reference's Perl script) over them
  cases.json          [{tool, index, args, input, out}]

tests/test_tools.py reads only the archives (never /root/reference).
"""
import json
import os
import shutil
import stat
import subprocess
import sys
import tarfile
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import common  # noqa: E402
import synth  # noqa: E402
from oracle import oracle as O  # noqa: E402

REF = os.path.join(ROOT, "oracle", "_ref")
INSPECT_MODES = [("names", ["-n"]), ("summary", ["-s"]), ("conv", ["--conversion-table"]), ("tree", ["--taxonomy-tree"]),
                 ("nametab", ["--name-table"]), ("sizetab", ["--size-table"]), ("fasta", []), ("fasta25", ["-a", "25"]),
                 ("fasta0", ["-a", "0"])]
PROMOTE_LEVELS = ["species", "genus", "family", "superkingdom", "lca", "phylum"]
KREPORT_VARIANTS = [("lca", []), ("nolca", ["--no-lca"]), ("zeros", ["--show-zeros"]), ("minscore", ["--min-score", "300"]),
                    ("minlen_nolca", ["--min-length", "60", "--no-lca"])]


def make_gaps(d):
    rng = np.random.default_rng(21)
    r = lambda n: synth.ACGT[rng.integers(0, 4, n, dtype=np.uint8)].tobytes()   # noqa: E731
    seqs = [("seq0 leading and trailing gaps", b"N" * 7 + r(300) + b"N" * 12 + r(150) + b"N" * 33),
            ("seq1 all gaps", b"N" * 90),
            ("seq2 short", r(30)),
            ("seq3 plain", r(500)),
            ("seq4 many gaps", b"N" + r(40) + b"N" + r(41) + b"NN" + r(42) + b"N" * 70 + r(43) + b"N"),
            ("seq5 last with trailing gap", r(200) + b"N" * 5)]
    with open(os.path.join(d, "genomes.fa"), "wb") as f:
        for nm, s in seqs:
            f.write(b">" + nm.encode() + b"\n")
            for p in range(0, len(s), 70):
                f.write(s[p:p + 70] + b"\n")
    synth.write_taxonomy(d, len(seqs), genus_size=3)
    O.ref_build(d, base="gaps", threads=1)
    for f in ("genomes.fa", "conv.tsv", "nodes.dmp", "names.dmp"):
        os.remove(os.path.join(d, f))


def main():
    if not O.have_ref() or not os.path.exists("/root/reference/centrifuge-kreport"):
        sys.exit("needs oracle/_ref and /root/reference")
    d = tempfile.mkdtemp()
    scratch = tempfile.mkdtemp()
    shutil.copy("/root/reference/centrifuge-kreport", scratch)
    shutil.copy("/root/reference/centrifuge-promote", scratch)
    shim = os.path.join(scratch, "centrifuge-inspect")
    with open(shim, "w") as f:
        f.write('#!/bin/sh\nexec %s/centrifuge-inspect-bin "$@"\n' % REF)
    os.chmod(shim, os.stat(shim).st_mode | stat.S_IEXEC)
    make_gaps(d)
    os.makedirs(os.path.join(d, "inspect")); os.makedirs(os.path.join(d, "kreport")); os.makedirs(os.path.join(d, "promote"))
    cases = []
    idx = {"gaps": os.path.join(d, "gaps")}
    for arch in ("example", "synth_small"):
        idx[arch] = os.path.join(common.golden(arch)[0], "idx")
    for name, base in idx.items():
        for mode, args in INSPECT_MODES:
            out = "inspect/%s.%s.txt" % (name, mode)
            r = subprocess.run([os.path.join(REF, "centrifuge-inspect-bin")] + args + [base], capture_output=True, check=True)
            open(os.path.join(d, out), "wb").write(r.stdout)
            cases.append({"tool": "inspect", "index": name, "args": args, "out": out})
    for arch in ("example", "synth_small"):
        gd, gc = common.golden(arch)
        for c in gc:
            for var, args in KREPORT_VARIANTS:
                r = subprocess.run(["perl", os.path.join(scratch, "centrifuge-kreport"), "-x", idx[arch]] + args + [os.path.join(gd, c["tsv"])],
                                   capture_output=True)
                if r.returncode != 0:
                    continue                      # "No sequence matches with given settings"
                out = "kreport/%s.%s.%s.txt" % (arch, c["name"], var)
                open(os.path.join(d, out), "wb").write(r.stdout)
                cases.append({"tool": "kreport", "index": arch, "args": args, "input": c["tsv"], "out": out})
    for arch in ("example", "synth_small"):
        gd, gc = common.golden(arch)
        for c in gc:
            for lv in PROMOTE_LEVELS:
                r = subprocess.run(["perl", os.path.join(scratch, "centrifuge-promote"), idx[arch], os.path.join(gd, c["tsv"]), lv], capture_output=True, check=True)
                out = "promote/%s.%s.%s.txt" % (arch, c["name"], lv)
                open(os.path.join(d, out), "wb").write(r.stdout)
                cases.append({"tool": "promote", "index": arch, "args": [lv], "input": c["tsv"], "out": out})
    # a count table, and two files in one call (the second file's header line is counted as a read
    # of an unprintable taxon, as the Perl script does)
    gd, _ = common.golden("synth_small")
    ct = os.path.join(d, "counts.txt")
    open(ct, "w").write("1000\t5\n1001\t2.5\n0\t3\n100\t1\n77777\t4\n")
    r = subprocess.run(["perl", os.path.join(scratch, "centrifuge-kreport"), "-x", idx["synth_small"], "--is-count-table", ct], capture_output=True, check=True)
    open(os.path.join(d, "kreport/synth_small.counts.txt"), "wb").write(r.stdout)
    cases.append({"tool": "kreport", "index": "synth_small", "args": ["--is-count-table"], "input": "@counts.txt", "out": "kreport/synth_small.counts.txt"})
    r = subprocess.run(["perl", os.path.join(scratch, "centrifuge-kreport"), "-x", idx["synth_small"], os.path.join(gd, "k5.tsv"), os.path.join(gd, "pe_k5.tsv")],
                       capture_output=True, check=True)
    open(os.path.join(d, "kreport/synth_small.two_files.txt"), "wb").write(r.stdout)
    cases.append({"tool": "kreport", "index": "synth_small", "args": [], "input": "k5.tsv,pe_k5.tsv", "out": "kreport/synth_small.two_files.txt"})
    json.dump(cases, open(os.path.join(d, "cases.json"), "w"), indent=1)
    out = os.path.join(HERE, "tools.tar.xz")
    with tarfile.open(out, "w:xz", preset=9) as t:
        for root, _, files in sorted(os.walk(d)):
            for f in sorted(files):
                p = os.path.join(root, f)
                t.add(p, arcname=os.path.relpath(p, d))
    print(out, os.path.getsize(out), "bytes,", len(cases), "cases")
    shutil.rmtree(d); shutil.rmtree(scratch)


if __name__ == "__main__":
    main()
