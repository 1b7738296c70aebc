#!/usr/bin/env python3
"""End-to-end wall time of the centrifuge-class front ends (ours vs the reference binary) on one
FASTA file: index built by the GPU builder, reads sampled from the genomes."""
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch  # noqa: E402
import bench  # noqa: E402
import synth  # noqa: E402
from centrifuge_amd import capi  # noqa: E402
from oracle import oracle as O  # noqa: E402


def main():
    G, L, n = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
    d = "/tmp/cf_e2e"
    os.makedirs(d, exist_ok=True)
    g = bench.gpu_genomes(torch, G, L)
    codes = bench.gpu_sample_reads(torch, g, n, 100, seed=5).cpu().numpy()
    host = g.cpu().numpy()
    del g
    torch.cuda.empty_cache()
    synth.write_taxonomy(d, G)
    capi.build_index(os.path.join(d, "idx"), codes=host.reshape(-1), seq_off=np.arange(G + 1, dtype=np.uint64) * np.uint64(L),
                     seq_names=[b"seq%d x" % i for i in range(G)], conversion_table=os.path.join(d, "conv.tsv"),
                     taxonomy_tree=os.path.join(d, "nodes.dmp"), name_table=os.path.join(d, "names.dmp"))
    bench.write_fasta(os.path.join(d, "reads.fa"), bench.read_names(n), codes)
    ours = os.path.join(ROOT, "centrifuge_amd", "bin", "centrifuge-class")
    ref = os.path.join(O.REF_DIR, "centrifuge-class")
    noref = len(sys.argv) > 4 and sys.argv[4] == "noref"
    runs = [("ours -p 8", ours, 8), ("ours -p 1", ours, 1)] + ([] if noref else [("reference -p 8", ref, 8)])
    for tag, exe, p in runs:
        t0 = time.time()
        r = subprocess.run([exe, "-f", "-t", "-p", str(p), "--reorder", "-x", os.path.join(d, "idx"), "-U", os.path.join(d, "reads.fa"),
                            "-S", os.path.join(d, tag.split()[0] + ".tsv"), "--report-file", os.path.join(d, tag.split()[0] + ".rep")],
                           capture_output=True, text=True)
        dt = time.time() - t0
        print("%-16s wall %.2fs -> %.3g reads/s  rc=%d | %s" % (tag, dt, n / dt, r.returncode,
              " ; ".join(l for l in r.stderr.splitlines() if "ime" in l or "search" in l or "Stage" in l)))
    if noref:
        return
    same = open(os.path.join(d, "ours.tsv")).read() == open(os.path.join(d, "reference.tsv")).read()
    same_rep = open(os.path.join(d, "ours.rep")).read() == open(os.path.join(d, "reference.rep")).read()
    print("TSV identical: %s, report identical: %s (%d reads)" % (same, same_rep, n))


if __name__ == "__main__":
    main()
