"""N > 1: two processes over the gloo backend (CPU) run the sharded path end to end —
contiguous shards, ONE all-reduce of the dense per-taxon counters, report images merged
on rank 0 — and must reproduce the reference's single-process TSV and report."""
import os
import subprocess
import sys
import tempfile

import numpy as np
import pytest

import common
from centrifuge_amd import dist as cfd


def test_shards_cover_the_queries():
    for n in (0, 1, 7, 8, 1000003):
        for w in (1, 2, 3, 8):
            r = [cfd.shard(n, k, w) for k in range(w)]
            assert r[0][0] == 0 and r[-1][1] == n
            assert all(r[i][1] == r[i + 1][0] for i in range(w - 1))
            sizes = [b - a for a, b in r]
            assert max(sizes) - min(sizes) <= 1


@pytest.mark.parametrize("arch,name,world", [("synth_small", "k5", 2), ("synth_small", "pe_k1", 2), ("example", "default", 3)])
def test_sharded_run_matches_reference(arch, name, world):
    d, cases = common.golden(arch)
    c = [x for x in cases if x["name"] == name][0]
    with tempfile.TemporaryDirectory() as out:
        port = 29500 + (os.getpid() % 2000)
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world),
               "--master-addr", "127.0.0.1", "--master-port", str(port),
               os.path.join(common.ROOT, "tests", "dist_worker.py"), arch, name, out]
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-3000:]
        from centrifuge_amd import reads
        tsv = reads.HEADER + "".join(open(os.path.join(out, "body_%d.tsv" % k)).read() for k in range(world))
        assert tsv == open(os.path.join(d, c["tsv"])).read()
        assert open(os.path.join(out, "report.tsv")).read() == open(os.path.join(d, c["report"])).read()
        # the all-reduced dense counters say the same as the report (numReads, numUniqueReads per taxon)
        counts = np.load(os.path.join(out, "counts.npy"))
        from emu import emu
        e = emu.Emu(os.path.join(d, "idx"))
        ntax = e.L.emu_num_taxa(e.h)
        mine = {e.L.emu_taxon_id(e.h, i): (int(counts[i]), int(counts[ntax + i])) for i in range(ntax)
                if counts[i] and e.L.emu_taxon_id(e.h, i) != 0}
        rep = {}
        for ln in open(os.path.join(d, c["report"])).read().splitlines()[1:]:
            f = ln.split("\t")
            rep[int(f[1])] = (int(f[4]), int(f[5]))
        assert mine == rep
