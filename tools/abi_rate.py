#!/usr/bin/env python3
"""PCIe-inclusive rate of the C ABI (GPU box): host buffers in (cf_batch_create: upload + device-side
plan + strand records), cf_classify, packed rows out (cf_batch_results_compact + cf_batch_max_scores),
on one batch of synthetic reads against a synthetic index.  This is NOT bench.py's metric (that one
starts with the reads resident in HBM); DESIGN.md quotes it beside it.  Prints one JSON line."""
import argparse
import json
import os
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import bench  # noqa: E402
import synth  # noqa: E402
from centrifuge_amd import capi  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--genomes", type=int, default=256)
    ap.add_argument("--genome-len", type=int, default=1048576)
    ap.add_argument("--reads", type=int, default=10_000_000)
    ap.add_argument("--reps", type=int, default=4)
    a = ap.parse_args()
    import torch
    d = tempfile.mkdtemp(prefix="cf_abi_")
    g = bench.gpu_genomes(torch, a.genomes, a.genome_len)
    codes = bench.gpu_sample_reads(torch, g, a.reads, 100, seed=5)
    host = g.cpu().numpy()
    del g
    torch.cuda.empty_cache()
    synth.write_taxonomy(d, a.genomes)
    base = os.path.join(d, "idx")
    capi.build_index(base, codes=host.reshape(-1), seq_off=np.arange(a.genomes + 1, dtype=np.uint64) * np.uint64(a.genome_len),
                     seq_names=[b"seq%d x" % i for i in range(a.genomes)], conversion_table=os.path.join(d, "conv.tsv"),
                     taxonomy_tree=os.path.join(d, "nodes.dmp"), name_table=os.path.join(d, "names.dmp"))
    ix = capi.Index(base)
    clf = capi.Classifier(ix)
    n = a.reads
    seq = np.ascontiguousarray(codes.reshape(-1))
    off = np.arange(n + 1, dtype=np.uint64) * np.uint64(100)
    seeds = bench.seeds_for(codes, bench.read_names(n))
    best = None
    for rep in range(a.reps):
        t0 = time.perf_counter()
        b = clf.batch(seq, off, seeds, False)
        t1 = time.perf_counter()
        b.classify()
        t2 = time.perf_counter()
        rows, first, n_rows, s2 = b.results_compact()
        ms = b.max_scores()
        t3 = time.perf_counter()
        b.close()
        t4 = time.perf_counter()
        cur = {"create_ms": (t1 - t0) * 1e3, "classify_ms": (t2 - t1) * 1e3, "results_ms": (t3 - t2) * 1e3, "destroy_ms": (t4 - t3) * 1e3,
               "total_ms": (t4 - t0) * 1e3, "rows": int(len(rows))}
        print("[abi] rep %d: %s" % (rep, {k: round(v, 1) for k, v in cur.items()}), file=sys.stderr)
        if best is None or cur["total_ms"] < best["total_ms"]:
            best = cur
    best = {k: (round(v, 2) if isinstance(v, float) else v) for k, v in best.items()}
    best.update({"reads": n, "read_len": 100, "reads_per_s_pcie_inclusive": round(n / (best["total_ms"] * 1e-3)),
                 "host_bytes_in": int(seq.nbytes + off.nbytes + seeds.nbytes), "host_bytes_out": int(rows.nbytes + n_rows.nbytes + s2.nbytes + ms.nbytes)})
    print(json.dumps(best))


if __name__ == "__main__":
    main()
