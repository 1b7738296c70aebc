#!/usr/bin/env python3
"""centrifuge-inspect at the scale of the benchmark index (GPU box): builds the synthetic index of
bench.py (default 2048 x 4 Mbp = 8.6 Gbp), inverts its BWT with cf_index_restore, checks the text
against the genomes it was built from (bit for bit), and times the inspector's FASTA mode end to end.
Prints one JSON line.  CF_RESTORE_VERBOSE=1 adds the per-pass kernel times on stderr."""
import argparse
import json
import os
import subprocess
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
    ap.add_argument("--genomes", type=int, default=2048)
    ap.add_argument("--genome-len", type=int, default=4194304)
    ap.add_argument("--fasta", action="store_true", help="also time the inspector binary writing FASTA to /dev/null")
    ap.add_argument("--keep", default=None, help="directory to build the index in (kept for later profiling runs)")
    a = ap.parse_args()
    import torch
    if a.keep:
        os.makedirs(a.keep, exist_ok=True)
    d = a.keep or tempfile.mkdtemp(prefix="cf_inspect_scale_")
    g = bench.gpu_genomes(torch, a.genomes, a.genome_len)
    host = g.cpu().numpy()
    del g
    torch.cuda.empty_cache()
    synth.write_taxonomy(d, a.genomes)
    names = [b"seq%d synthetic genome %d" % (i, i) for i in range(a.genomes)]
    off = np.arange(a.genomes + 1, dtype=np.uint64) * np.uint64(a.genome_len)
    base = os.path.join(d, "idx")
    bt = capi.build_index(base, codes=host.reshape(-1), seq_off=off, seq_names=names, conversion_table=os.path.join(d, "conv.tsv"),
                          taxonomy_tree=os.path.join(d, "nodes.dmp"), name_table=os.path.join(d, "names.dmp"))
    ix = capi.Index(base)
    n = ix.text_len
    t0 = time.time()
    packed = ix.restore()
    t_restore = time.time() - t0
    t0 = time.time()
    packed = ix.restore()
    t_restore2 = time.time() - t0
    ix.close()
    flat = host.reshape(-1)
    assert n == flat.size
    m = n // 4 * 4
    q = flat[:m].reshape(-1, 4)
    want = (q[:, 0] | (q[:, 1] << 2) | (q[:, 2] << 4) | (q[:, 3] << 6)).astype(np.uint8)
    same = bool(np.array_equal(want, packed[: m // 4]))
    tail = 0
    for j in range(n - m):
        tail |= int(flat[m + j]) << (2 * j)
    same = same and int(packed[m // 4]) == tail
    out = {"text_len": int(n), "build_s": bt[3], "restore_s_first": round(t_restore, 3), "restore_s": round(t_restore2, 3),
           "restore_gbp_per_s": round(n / t_restore2 / 1e9, 3), "text_identical_to_input": same}
    if a.fasta:
        exe = os.path.join(ROOT, "centrifuge_amd", "bin", "centrifuge-inspect")
        t0 = time.time()
        r = subprocess.run([exe, base], stdout=subprocess.DEVNULL, stderr=subprocess.PIPE)
        out["inspect_fasta_wall_s"] = round(time.time() - t0, 2)
        out["inspect_rc"] = r.returncode
        sys.stderr.write(r.stderr.decode()[-2000:])
        # the same through a pipe, counting the bytes: every record is '>' name '\n' + the bases in lines of 60
        t0 = time.time()
        r = subprocess.run("%s %s 2>/dev/null | wc -c" % (exe, base), shell=True, capture_output=True)
        out["inspect_fasta_piped_wall_s"] = round(time.time() - t0, 2)
        want_bytes = sum(len(nm) + 2 for nm in names) + a.genomes * (a.genome_len + (a.genome_len + 59) // 60)
        out["fasta_bytes"] = int(r.stdout.split()[0])
        out["fasta_bytes_expected"] = want_bytes
    print(json.dumps(out))


if __name__ == "__main__":
    main()
