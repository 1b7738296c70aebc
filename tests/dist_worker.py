"""Worker of tests/test_dist_gloo.py: one rank of the sharded classification path on CPU.
The kernels are single-stepped by the CPU harness (tests/emu — test infrastructure); what
is under test is the N > 1 plumbing of centrifuge_amd/dist.py: sharding, the counter
all-reduce and the report merge, over the gloo backend."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tools")):
    if p not in sys.path:
        sys.path.insert(0, p)

import common  # noqa: E402
from centrifuge_amd import capi, reads, dist as cfd  # noqa: E402
from emu import emu  # noqa: E402
from test_report import max_scores  # noqa: E402
from oracle import oracle as O  # noqa: E402


def main():
    arch, name, outdir = sys.argv[1], sys.argv[2], sys.argv[3]
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    d, cases = common.golden(arch)
    c = [x for x in cases if x["name"] == name][0]
    kw, fastq = common.case_kwargs(c["args"])
    base = os.path.join(d, "idx")
    names, qlens, seq, off, seeds, paired = reads.load([os.path.join(d, f) for f in c["reads"]], fastq)
    nq = len(names)
    lo, hi = cfd.shard(nq, rank, world)
    per = 2 if paired else 1
    r0, r1 = lo * per, hi * per
    s_off = (off[r0:r1 + 1] - off[r0]).astype(np.uint64)
    s_seq = seq[int(off[r0]):int(off[r1])] if r1 > r0 else np.zeros(1, dtype=np.uint8)
    e = emu.Emu(base)
    rows, n_rows, score2, cnt = e.classify(s_seq, s_off, seeds[r0:r1], paired=paired, counts=True, **kw)
    # the one collective of the path: dense per-taxon counters, summed in place
    counts = torch.from_numpy(cnt.astype(np.int64))
    cfd.allreduce_counts(dist, counts)
    # per-rank report image -> rank 0
    ix = capi.Index(base, host_only=True)
    rep = capi.Report(ix)
    orc = O.Oracle(base)
    ms = max_scores(orc, s_seq, s_off, paired)
    rep.add(rows, n_rows, ms, kw.get("k", 5))
    body = reads.format_tsv(e.seqid, names[lo:hi], qlens[lo:hi], rows, n_rows, score2)[len(reads.HEADER):]
    with open(os.path.join(outdir, "body_%d.tsv" % rank), "w") as f:
        f.write(body)
    if cfd.merge_reports(dist, rep, rank, world):
        rep.write(os.path.join(outdir, "report.tsv"))
        np.save(os.path.join(outdir, "counts.npy"), counts.numpy())
    # bench.py's bookkeeping over the ranks: every rank's timed region on every rank (the step time is the slowest rank's)
    times = cfd.per_rank(dist, 1.5 + rank, rank, world)
    assert times == [1.5 + r for r in range(world)] and max(times) == 0.5 + world, times
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
