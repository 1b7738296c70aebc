"""One process per GPU: how the classification path shards over the GPUs of a node.

Reads are independent (Classifier::go keeps no cross-read state, classifier.h:224), so
the index is replicated in every GPU's HBM, each rank classifies a contiguous shard of
the queries, and nothing is exchanged on the data path.  Two things meet at the end
(SURVEY.md §8e): the dense per-taxon counters are summed in place with ONE all-reduce
(RCCL over xGMI on the GPUs: backend "nccl"; gloo in the CPU tests), and rank 0 merges
the small per-rank report images (counters + observed tuples, SpeciesMetrics::merge
aln_sink.h:109-140) to run the EM and write the report.  Per-read rows stay in shard
order, so concatenating the ranks' TSV bodies by rank reproduces the --reorder output.
"""
import numpy as np


def shard(n_queries, rank, world):
    """Contiguous query range [lo, hi) of `rank`: shards differ by at most one query."""
    base, extra = divmod(n_queries, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def allreduce_counts(dist, counts):
    """In-place SUM all-reduce of the dense [2 * n_taxa] counter tensor (device tensor on the
    GPUs — it aliases cf_counts_device — or a CPU tensor under gloo)."""
    dist.all_reduce(counts, op=dist.ReduceOp.SUM)
    return counts


def merge_reports(dist, report, rank, world, device=None):
    """Gather every rank's serialized report on rank 0 and merge it there (sizes first,
    then the padded images through one all-gather).  Returns True on the rank holding the
    merged report.  `device`: where the collective's tensors live — None = CPU (gloo); under
    the "nccl" backend (RCCL) collectives only take device tensors, so pass the rank's
    torch.device("cuda", i): the (small) images make one round trip through HBM."""
    import torch
    if device is None and dist.get_backend() == "nccl":
        device = torch.device("cuda", torch.cuda.current_device())
    img = torch.from_numpy(report.serialize().view(np.int64))
    n = torch.tensor([img.numel()], dtype=torch.int64, device=device)
    sizes = [torch.zeros(1, dtype=torch.int64, device=device) for _ in range(world)]
    dist.all_gather(sizes, n)
    sizes = [int(s.item()) for s in sizes]
    cap = max(max(sizes), 1)
    pad = torch.zeros(cap, dtype=torch.int64, device=device)
    pad[:img.numel()] = img.to(pad.device)
    bufs = [torch.zeros(cap, dtype=torch.int64, device=device) for _ in range(world)]
    dist.all_gather(bufs, pad)
    if rank != 0:
        return False
    for r in range(1, world):
        report.merge(bufs[r][:sizes[r]].cpu().numpy().view(np.uint64))
    return True


def per_rank(dist, value, rank, world, device=None):
    """every rank's `value` (a float: its timed region, its index-open time ...) on every rank, as a list by rank — one SUM
    all-reduce of a one-hot vector.  bench.py's step time is the MAX of the timed regions (the slowest rank), and the list
    shows a straggler.  `device` as in merge_reports."""
    import torch
    if device is None and dist.get_backend() == "nccl":
        device = torch.device("cuda", torch.cuda.current_device())
    t = torch.zeros(world, dtype=torch.float64, device=device)
    t[rank] = float(value)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return [float(x) for x in t.tolist()]
