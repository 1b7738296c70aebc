#!/usr/bin/env python3
"""Scale probe for the GPU index builder: synthetic genomes generated on the GPU
(torch), built with cf_build_index, sizes and phase timings printed."""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import synth  # noqa: E402
from centrifuge_amd import capi  # noqa: E402


def gpu_genomes(n_genomes, length, genus_size=8, divergence=0.05, seed=12345, device="cuda"):
    """[n_genomes, length] base codes 0..3 in pinned host memory + the device copy."""
    gen = torch.Generator(device=device)
    gen.manual_seed(seed)
    dev = torch.empty((n_genomes, length), dtype=torch.uint8, device=device)
    for g0 in range(0, n_genomes, genus_size):
        anc = torch.randint(0, 4, (length,), dtype=torch.uint8, device=device, generator=gen)
        for i in range(g0, min(g0 + genus_size, n_genomes)):
            mut = torch.rand(length, device=device, generator=gen) < divergence
            add = torch.randint(1, 4, (length,), dtype=torch.uint8, device=device, generator=gen)
            dev[i] = (anc + add * mut) & 3
    return dev


def main():
    """build_scale.py GENOMES LENGTH [OUTDIR [iid|repeat]] — `repeat` = bench.py's repeat-rich recipe (config 2r)"""
    G, L = int(sys.argv[1]), int(sys.argv[2])
    out = sys.argv[3] if len(sys.argv) > 3 else "/tmp/cf_scale"
    recipe = sys.argv[4] if len(sys.argv) > 4 else "iid"
    os.makedirs(out, exist_ok=True)
    t0 = time.time()
    if recipe == "repeat":
        import bench
        dev = bench.gpu_genomes(torch, G, L, recipe="repeat")
    else:
        dev = gpu_genomes(G, L)
    host = dev.cpu().numpy()
    torch.cuda.synchronize()
    t1 = time.time()
    synth.write_taxonomy(out, G)
    names = [b"seq%d synthetic genome %d" % (i, i) for i in range(G)]
    off = np.arange(G + 1, dtype=np.uint64) * np.uint64(L)
    del dev
    torch.cuda.empty_cache()
    t = capi.build_index(os.path.join(out, "idx"), codes=host.reshape(-1), seq_off=off, seq_names=names,
                         conversion_table=os.path.join(out, "conv.tsv"), taxonomy_tree=os.path.join(out, "nodes.dmp"),
                         name_table=os.path.join(out, "names.dmp"), verbose=True)
    sz = sum(os.path.getsize(os.path.join(out, "idx.%d.cf" % k)) for k in (1, 2, 3, 4))
    print("recipe %s:" % recipe, end=" ")
    print("genomes %d x %d = %.3f Gbp: generate %.1fs, build parse %.1fs gpu %.1fs write %.1fs total %.1fs, index %.3f GB" %
          (G, L, G * L / 1e9, t1 - t0, t[0], t[1], t[2], t[3], sz / 1e9))


if __name__ == "__main__":
    main()
