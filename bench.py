#!/usr/bin/env python3
"""bench.py — classified reads/s of the MI355X-native classification path.

A "step" is one pass of the hot path (search -> post -> walk -> score kernels,
cf_classify) over one batch of synthetic 100 bp reads whose packed bases, seeds
and workspace are already resident in HBM (cf_batch_create ran before the timed
region); the index is resident too.  N > 1: one process per GPU (torchrun), the
index replicated per GPU, reads sharded (each rank classifies its own batch: weak
scaling) and ONE collective per step — an RCCL all-reduce over xGMI of the dense
per-taxon counters.  Rank 0 prints one JSON line (contract in the task brief).

Workload: BASELINE.json config 2 is "p_compressed (~4.2 GB) + 10M synthetic 100 bp
reads on 1 x MI355X".  p_compressed cannot be downloaded here, so the index is a
synthetic stand-in of the same size class (default 2048 genomes x 4 Mbp = 8.6 Gbp
-> ~3.9 GB of index, genera of 8 genomes at 5 % divergence; SURVEY.md §8d recipe),
generated on the GPU and built inside the run by our own GPU builder
(cf_build_index; byte-identical to the reference's centrifuge-build, see
tests/test_gpu_build.py).  Index construction is outside the timed path.
"""
import argparse
import json
import os
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "tools")):
    if p not in sys.path:
        sys.path.insert(0, p)

HBM_PEAK_GBPS = 8000.0          # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
NAME_DIGITS = 9


def log(*a):
    if int(os.environ.get("RANK", "0")) == 0:
        print("[bench]", *a, file=sys.stderr, flush=True)


# ------------------------------------------------------------------ synthetic data (on the GPU, torch = plumbing)
def gpu_genomes(torch, n_genomes, length, genus_size=8, divergence=0.05, seed=12345):
    """[n_genomes, length] base codes 0..3 on the current device: genera of `genus_size`
    members, each member = the genus ancestor with `divergence` substitutions."""
    gen = torch.Generator(device="cuda")
    gen.manual_seed(seed)
    dev = torch.empty((n_genomes, length), dtype=torch.uint8, device="cuda")
    for g0 in range(0, n_genomes, genus_size):
        anc = torch.randint(0, 4, (length,), dtype=torch.uint8, device="cuda", generator=gen)
        for i in range(g0, min(g0 + genus_size, n_genomes)):
            mut = torch.rand(length, device="cuda", generator=gen) < divergence
            add = torch.randint(1, 4, (length,), dtype=torch.uint8, device="cuda", generator=gen)
            dev[i] = (anc + add * mut) & 3
    return dev


def gpu_sample_reads(torch, genomes, n_reads, read_len, seed, mut_frac=0.63, random_frac=0.01, n_frac=0.001):
    """SURVEY §8(d) read recipe -> codes [n_reads, read_len] (0..4, numpy): uniform over genomes
    and strands, 63 % with one substitution, 1 % random reads, 0.1 % with a short N run."""
    gen = torch.Generator(device="cuda")
    gen.manual_seed(seed)
    G, L = genomes.shape
    flat = genomes.reshape(-1)
    out = torch.empty((n_reads, read_len), dtype=torch.uint8, device="cuda")
    ar = torch.arange(read_len, device="cuda")
    CH = 1 << 21
    for s in range(0, n_reads, CH):
        e = min(n_reads, s + CH)
        m = e - s
        gi = torch.randint(0, G, (m,), device="cuda", generator=gen)
        pos = torch.randint(0, L - read_len + 1, (m,), device="cuda", generator=gen)
        r = flat[(gi * L + pos)[:, None] + ar[None, :]]
        rc = torch.rand(m, device="cuda", generator=gen) < 0.5
        r = torch.where(rc[:, None], 3 - r.flip(1), r)
        mut = torch.rand(m, device="cuda", generator=gen) < mut_frac
        mp = torch.randint(0, read_len, (m,), device="cuda", generator=gen)
        add = torch.randint(1, 4, (m,), dtype=torch.uint8, device="cuda", generator=gen)
        rows = torch.arange(m, device="cuda")
        r[rows, mp] = torch.where(mut, (r[rows, mp] + add) & 3, r[rows, mp])
        rnd = torch.rand(m, device="cuda", generator=gen) < random_frac
        r = torch.where(rnd[:, None], torch.randint(0, 4, (m, read_len), dtype=torch.uint8, device="cuda", generator=gen), r)
        nn = torch.rand(m, device="cuda", generator=gen) < n_frac
        q = torch.randint(0, read_len - 3, (m,), device="cuda", generator=gen)
        ln = torch.randint(1, 4, (m,), device="cuda", generator=gen)
        nmask = nn[:, None] & (ar[None, :] >= q[:, None]) & (ar[None, :] < (q + ln)[:, None])
        r = torch.where(nmask, torch.full_like(r, 4), r)
        out[s:e] = r
    return out.cpu().numpy()


def gpu_sample_pairs(torch, genomes, n_pairs, read_len, seed, frag=(250, 400), mut_frac=0.63, random_frac=0.01, n_frac=0.001):
    """FR pairs from fragments of frag[0]..frag[1] bp: mate 1 = the fragment's
    first read_len bases, mate 2 = the reverse complement of its last read_len bases; the whole pair is
    flipped with p = 1/2; per-mate substitution / N recipe as for single reads.  -> codes [2 n_pairs, read_len]"""
    gen = torch.Generator(device="cuda")
    gen.manual_seed(seed)
    G, L = genomes.shape
    flat = genomes.reshape(-1)
    out = torch.empty((2 * n_pairs, read_len), dtype=torch.uint8, device="cuda")
    ar = torch.arange(read_len, device="cuda")
    CH = 1 << 20
    for s in range(0, n_pairs, CH):
        e = min(n_pairs, s + CH)
        m = e - s
        gi = torch.randint(0, G, (m,), device="cuda", generator=gen)
        fl = torch.randint(max(frag[0], read_len), frag[1] + 1, (m,), device="cuda", generator=gen)
        pos = (torch.rand(m, device="cuda", generator=gen) * (L - fl + 1).double()).long().clamp_(max=L - frag[1] - 1)
        left = flat[(gi * L + pos)[:, None] + ar[None, :]]
        right = 3 - flat[(gi * L + pos + fl - read_len)[:, None] + ar[None, :]].flip(1)
        flip = torch.rand(m, device="cuda", generator=gen) < 0.5
        m1 = torch.where(flip[:, None], right, left)
        m2 = torch.where(flip[:, None], left, right)
        r = torch.stack([m1, m2], dim=1).reshape(2 * m, read_len)
        k = 2 * m
        rows = torch.arange(k, device="cuda")
        mut = torch.rand(k, device="cuda", generator=gen) < mut_frac
        mp = torch.randint(0, read_len, (k,), device="cuda", generator=gen)
        add = torch.randint(1, 4, (k,), dtype=torch.uint8, device="cuda", generator=gen)
        r[rows, mp] = torch.where(mut, (r[rows, mp] + add) & 3, r[rows, mp])
        rnd = (torch.rand(m, device="cuda", generator=gen) < random_frac).repeat_interleave(2)
        r = torch.where(rnd[:, None], torch.randint(0, 4, (k, read_len), dtype=torch.uint8, device="cuda", generator=gen), r)
        nn = torch.rand(k, device="cuda", generator=gen) < n_frac
        q = torch.randint(0, read_len - 3, (k,), device="cuda", generator=gen)
        ln = torch.randint(1, 4, (k,), device="cuda", generator=gen)
        nmask = nn[:, None] & (ar[None, :] >= q[:, None]) & (ar[None, :] < (q + ln)[:, None])
        r = torch.where(nmask, torch.full_like(r, 4), r)
        out[2 * s:2 * e] = r
    return out.cpu().numpy()


def read_names(n):
    """fixed-width names r000000000 ... as a [n, 1+NAME_DIGITS] byte matrix"""
    idx = np.arange(n, dtype=np.int64)
    m = np.empty((n, 1 + NAME_DIGITS), dtype=np.uint8)
    m[:, 0] = ord("r")
    for k in range(NAME_DIGITS):
        m[:, NAME_DIGITS - k] = (idx // 10 ** k) % 10 + 48
    return m


def seeds_for(codes, names, global_seed=0):
    """genRandSeed (pat.h:55-91), vectorised for equal-length FASTA reads and fixed-width names."""
    n, L = codes.shape
    r = np.full(n, ((global_seed + 101) * 59 * 61 * 67 * 71 * 73 * 79 * 83) & 0xffffffff, dtype=np.uint32)
    CH = 1 << 20
    sh = ((np.arange(L) & 15) << 1).astype(np.uint32)
    for s in range(0, n, CH):
        r[s:s + CH] ^= np.bitwise_xor.reduce(codes[s:s + CH].astype(np.uint32) << sh[None, :], axis=1)
    q = np.uint32(0)
    for i in range(L):
        q ^= np.uint32(ord("I") << ((i & 3) << 3))
    r ^= q
    sh = ((np.arange(names.shape[1]) & 3) << 3).astype(np.uint32)
    r ^= np.bitwise_xor.reduce(names.astype(np.uint32) << sh[None, :], axis=1)      # names hold no '/'
    return r


def write_fasta(path, names, codes, suffix=b""):
    n, L = codes.shape
    w = names.shape[1]
    sfx = len(suffix)
    rec = np.empty((n, w + sfx + L + 3), dtype=np.uint8)
    rec[:, 0] = ord(">")
    rec[:, 1:1 + w] = names
    if sfx:
        rec[:, 1 + w:1 + w + sfx] = np.frombuffer(suffix, dtype=np.uint8)
    rec[:, 1 + w + sfx] = 10
    rec[:, 2 + w + sfx:2 + w + sfx + L] = np.frombuffer(b"ACGTN", dtype=np.uint8)[codes]
    rec[:, -1] = 10
    rec.tofile(path)


# ------------------------------------------------------------------ CPU baseline = the unmodified reference
def cpu_baseline(base, workdir, codes, names, procs, threads, k, paired=False):
    """oracle/_ref/centrifuge-class (the reference, compiled from its own sources) on a bounded
    sample of the same reads.  The reference stops scaling at ~8 threads per process on this box
    (its read parser and output queue are mutexed), so the box is filled with `procs` processes
    x `threads` threads on disjoint shards.  Index load is measured by the same processes on a
    1-read file and subtracted.  Returns (reads/s, tsv of shard 0, details)."""
    from oracle import oracle as O
    exe = os.path.join(O.REF_DIR, "centrifuge-class")
    n = len(names)                       # queries (reads, or pairs with the mates adjacent in `codes`)
    per = (n + procs - 1) // procs
    shards = [(i * per, min(n, (i + 1) * per)) for i in range(procs) if i * per < n]

    def put(tag, s, e):
        if not paired:
            write_fasta(os.path.join(workdir, "cpu_%s.fa" % tag), names[s:e], codes[s:e])
            return ["-U", os.path.join(workdir, "cpu_%s.fa" % tag)]
        write_fasta(os.path.join(workdir, "cpu_%s_1.fa" % tag), names[s:e], codes[2 * s:2 * e:2], b"/1")
        write_fasta(os.path.join(workdir, "cpu_%s_2.fa" % tag), names[s:e], codes[2 * s + 1:2 * e:2], b"/2")
        return ["-1", os.path.join(workdir, "cpu_%s_1.fa" % tag), "-2", os.path.join(workdir, "cpu_%s_2.fa" % tag)]

    one = put("one", 0, 1)
    inputs = [put(str(i), s, e) for i, (s, e) in enumerate(shards)]

    def run(files, tag):
        t0 = time.time()
        ps = [subprocess.Popen([exe, "-f", "-p", str(threads), "--reorder", "-k", str(k), "-x", base] + f +
                               ["-S", os.path.join(workdir, "cpu_%s_%d.tsv" % (tag, i)),
                                "--report-file", os.path.join(workdir, "cpu_%s_%d.rep" % (tag, i))],
                               stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL) for i, f in enumerate(files)]
        for p in ps:
            if p.wait() != 0:
                raise RuntimeError("reference centrifuge-class failed")
        return time.time() - t0

    t_load = run([one] * len(shards), "load")
    t_all = run(inputs, "run")
    search = max(t_all - t_load, 1e-3)
    tsv0 = open(os.path.join(workdir, "cpu_run_0.tsv")).read()
    return n / search, tsv0, shards[0][1], {"wall_s": t_all, "index_load_s": t_load, "search_s": search}


def effective_cores():
    """CPUs this container may actually use: min(affinity, cgroup cpu.max quota)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            n = min(n, max(1, int(q) // int(per)))
    except Exception:
        pass
    return n


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--genomes", type=int, default=int(os.environ.get("CF_BENCH_GENOMES", 2048)))
    ap.add_argument("--genome-len", type=int, default=int(os.environ.get("CF_BENCH_GENOME_LEN", 4194304)))
    ap.add_argument("--reads", type=int, default=int(os.environ.get("CF_BENCH_READS", 10000000)),
                    help="reads per GPU per step")
    ap.add_argument("--read-len", type=int, default=100)
    ap.add_argument("--paired", action="store_true", help="reads are FR pairs (mates adjacent); --reads counts mates")
    ap.add_argument("--cpu-sample", type=int, default=int(os.environ.get("CF_BENCH_CPU_SAMPLE", 1000000)))
    ap.add_argument("--cpu-threads", type=int, default=8, help="threads per reference process")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--split", type=int, default=int(os.environ.get("CF_BENCH_SPLIT", 1)),
                    help="classify the step's batch as this many sub-batches on concurrent HIP streams")
    a = ap.parse_args()

    import torch
    from centrifuge_amd import capi, reads as rd, dist as cfd
    import synth

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench: no GPU — the classification path has no CPU fallback")
    torch.cuda.set_device(local)
    dist = None
    if world > 1 or os.environ.get("CF_BENCH_FORCE_DIST"):     # the env knob drives the collective path with one rank (tests)
        import torch.distributed as dist
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))
    nproc = effective_cores()

    workdir = os.path.join(os.environ.get("CF_BENCH_DIR") or tempfile.gettempdir(), "cf_bench_%d_%d" % (a.genomes, a.genome_len))
    os.makedirs(workdir, exist_ok=True)
    base = os.path.join(workdir, "idx")
    have_index = all(os.path.exists(base + ".%d.cf" % k) for k in (1, 2, 3, 4))
    reads_cache = os.path.join(workdir, "reads_%d_%d_%d%s.npy" % (a.reads, a.read_len, rank, "_pe" if a.paired else ""))
    per = 2 if a.paired else 1
    if a.reads % per:
        raise SystemExit("bench: --reads must be even with --paired")

    # ---- synthetic genomes (every rank, same seed) and this rank's shard of the reads; a second run
    #      in the same work directory (profiling passes) reuses the index and the sampled reads
    t0 = time.time()
    genomes = None
    if have_index and os.path.exists(reads_cache):
        codes = np.load(reads_cache)
        log("reusing the index and reads cached in %s" % workdir)
    else:
        genomes = gpu_genomes(torch, a.genomes, a.genome_len)
        codes = (gpu_sample_pairs(torch, genomes, a.reads // 2, a.read_len, seed=777 + rank) if a.paired else
                 gpu_sample_reads(torch, genomes, a.reads, a.read_len, seed=777 + rank))
        torch.cuda.synchronize()
        if os.environ.get("CF_BENCH_DIR"):
            np.save(reads_cache, codes)
        log("genomes %d x %d bp + %d reads generated on the GPU in %.1fs" % (a.genomes, a.genome_len, a.reads, time.time() - t0))

    # ---- index: rank 0 builds it with the GPU builder, everybody loads its own HBM replica
    build_s = None
    if rank == 0 and not have_index:
        host = genomes.cpu().numpy()
        del genomes
        torch.cuda.empty_cache()
        synth.write_taxonomy(workdir, a.genomes)
        names_g = [b"seq%d synthetic genome %d" % (i, i) for i in range(a.genomes)]
        goff = np.arange(a.genomes + 1, dtype=np.uint64) * np.uint64(a.genome_len)
        bt = capi.build_index(base + ".tmp", codes=host.reshape(-1), seq_off=goff, seq_names=names_g, device=local,
                              conversion_table=os.path.join(workdir, "conv.tsv"), taxonomy_tree=os.path.join(workdir, "nodes.dmp"),
                              name_table=os.path.join(workdir, "names.dmp"))
        for k in (1, 2, 3, 4):
            os.replace(base + ".tmp.%d.cf" % k, base + ".%d.cf" % k)
        build_s = bt[3]
        del host
        log("index built on the GPU in %.1fs (suffix sort + BWT %.1fs)" % (bt[3], bt[1]))
    else:
        del genomes
        torch.cuda.empty_cache()
    if dist is not None:
        dist.barrier()
    t0 = time.time()
    ix = capi.Index(base, device=local)
    clf = capi.Classifier(ix)
    log("index in HBM: %.2f GB, text %.2f Gbp, load %.1fs" % (ix.device_bytes / 1e9, ix.text_len / 1e9, time.time() - t0))

    # ---- batch resident in HBM before the timed region
    nq_all = a.reads // per
    ns = min(nq_all, a.cpu_sample // per) if rank == 0 and not a.no_cpu else 0      # sampled queries
    names = read_names(ns)
    seeds = np.zeros(a.reads, dtype=np.uint32)
    if ns:                                      # the mates of a pair share the name ("/1", "/2" are not hashed)
        seeds[:ns * per] = seeds_for(codes[:ns * per], np.repeat(names, per, axis=0))
    off = (np.arange(a.reads + 1, dtype=np.uint64) * np.uint64(a.read_len))
    # the step's batch, optionally as S sub-batches whose kernels overlap on S HIP streams (the
    # latency-bound per-query kernels of one sub-batch fill the gaps of the other's search kernel)
    S = max(1, a.split)
    cut = [per * (nq_all * i // S) for i in range(S + 1)]
    batches = [clf.batch(codes[cut[i]:cut[i + 1]].reshape(-1), off[:cut[i + 1] - cut[i] + 1], seeds[cut[i]:cut[i + 1]], paired=a.paired)
               for i in range(S)]
    streams = [torch.cuda.Stream() for _ in range(S)]
    pool = None
    if S > 1:
        from concurrent.futures import ThreadPoolExecutor
        pool = ThreadPoolExecutor(S)
    counts_ptr = clf.counts_device_ptr()
    n_taxa = ix.num_taxa

    class _Raw:                   # expose the library's device counters to torch (RCCL all-reduce in place)
        __cuda_array_interface__ = {"shape": (2 * n_taxa,), "typestr": "<i8", "data": (counts_ptr, False), "version": 2}
    counts_t = torch.as_tensor(_Raw(), device=torch.device("cuda", local)) if dist is not None else None

    plan_ms = [0.0]

    def one(i):                   # one pass over a resident batch: plan + strand records, then the four stages
        plan_ms[0] += batches[i].plan(streams[i].cuda_stream)
        batches[i].classify(streams[i].cuda_stream)

    def step():
        # A step starts from the raw reads resident in HBM (1 byte per base, offsets, seeds): the device-side
        # plan (filters, hit capacities, work list) and the strand records are part of every timed step.
        if S == 1:
            one(0)
        else:                     # the calls block until their kernels are done; ctypes drops the GIL
            list(pool.map(one, range(S)))
        if dist is not None:
            with torch.cuda.stream(streams[0]):
                cfd.allreduce_counts(dist, counts_t)       # the one collective of the path (RCCL over xGMI)

    def step_times():
        return sum(np.array(b.timings()) for b in batches)

    def step_ops():               # instrumented re-run of the search / walk kernels, outside the timed region
        o = capi.OpCounts()
        for b in batches:
            x = b.opcounts()
            for f, _ in capi.OpCounts._fields_:
                setattr(o, f, getattr(o, f) + getattr(x, f))
        return o

    for _ in range(a.warmup):
        step()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    kms = np.zeros(5)
    ops = None
    plan_ms[0] = 0.0
    t0 = time.perf_counter()
    for _ in range(a.steps):
        step()
        kms += step_times()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if dist is not None:
        tt = torch.tensor([dt], device="cuda")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    kms /= max(1, a.steps)
    plan_step_ms = plan_ms[0] / max(1, a.steps)
    ops = step_ops()

    if rank == 0:
        total_reads = a.reads * world * a.steps
        value = total_reads / dt
        # dominant kernel = k_search; algorithmic bytes per launch (SURVEY.md §8d formula, search part):
        # 128 B per distinct side touched per LF step + 16 B per ftab lookup + packed read in
        search_bytes = 128 * (ops.n_pair + ops.n_pair2 + ops.n_single) + 16 * ops.n_ftab + \
            ((a.read_len + 3) // 4 + (a.read_len + 7) // 8) * a.reads
        achieved = search_bytes / (kms[0] * 1e-3) / 1e9
        whole_bytes = ops.algorithmic_bytes(ix.sa_width, a.reads, a.read_len)
        rand_gbps = ix.random_read_gbps(1 << 26, 64)
        res = {
            "metric": "classified reads/sec (whole node) on 100bp synthetic reads vs p_compressed; HBM GB/s achieved",
            "value": value, "unit": "reads/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": dt / a.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u64", "data": "synthetic",
            "config": {"workload": "p_compressed stand-in: synthetic index %d genomes x %d bp = %.2f Gbp (%.2f GB resident in HBM; "
                                   "p_compressed itself is ~4.2 GB and not downloadable here), %d x %d bp %s reads per GPU per step, -k 5" %
                                   (a.genomes, a.genome_len, ix.text_len / 1e9, ix.device_bytes / 1e9, a.reads, a.read_len, "PE (FR pairs, mates counted)" if a.paired else "SE"),
                       "index_bytes": ix.device_bytes, "reads_per_gpu_per_step": a.reads, "read_len": a.read_len,
                       "index_build_s_gpu": build_s,
                       "parallelism": "index replicated per GPU, reads sharded, RCCL all-reduce of per-taxon counters"},
            "roofline": {"bound": "hbm", "kernel": "k_search2", "achieved": achieved, "peak": HBM_PEAK_GBPS,
                         "unit": "GB/s", "frac": achieved / HBM_PEAK_GBPS, "traffic": None,
                         "kernel_ms": kms[0], "algorithmic_bytes_per_launch": search_bytes,
                         "algorithmic_bytes_per_read_whole_path": whole_bytes / a.reads,
                         "whole_path_GBps": whole_bytes / ((plan_step_ms + kms[4]) * 1e-3) / 1e9,
                         "measured_random_128B_read_GBps": rand_gbps,
                         "frac_of_measured_random": achieved / rand_gbps if rand_gbps else None},
            "streams": S,
            "kernels_ms": {"plan": plan_step_ms, "search": kms[0], "post": kms[1], "walk": kms[2], "score": kms[3],
                           "total": plan_step_ms + kms[4]},
            "ops_per_read": {"ftab": ops.n_ftab / a.reads, "pair": ops.n_pair / a.reads, "pair2": ops.n_pair2 / a.reads,
                             "single": ops.n_single / a.reads, "walk": ops.n_walk / a.reads, "rows": ops.n_rows / a.reads},
        }
        # HBM traffic of the dominant kernel from the committed rocprofv3 PMC passes of this very workload
        try:
            pm = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))
            if (pm["genomes"], pm["genome_len"], pm["reads"], pm["read_len"]) == (a.genomes, a.genome_len, a.reads, a.read_len):
                res["roofline"]["traffic"] = pm["traffic_bytes_per_launch"]
                res["roofline"]["traffic_source"] = "profiles/pmc_traffic.json (rocprofv3 --pmc FETCH_SIZE x2 + WRITE_SIZE, kernel %s)" % pm["kernel"]
        except Exception:
            pass
        if not a.no_cpu:
            try:
                procs = max(1, nproc // a.cpu_threads)       # usable cores (cgroup quota) / threads per process
                qps, tsv0, n0, det = cpu_baseline(base, workdir, codes[:ns * per], names, procs, a.cpu_threads, 5, a.paired)
                rps = qps * per
                res["cpu_baseline"] = {"value": rps, "unit": "reads/s", "cores": procs * a.cpu_threads, "kind": "reference",
                                       "sample": "first %d reads of rank 0's batch%s; %d processes x %d threads of the reference's "
                                                 "centrifuge-class (--reorder) on disjoint shards, search time = wall %.1fs minus "
                                                 "index load %.1fs measured the same way; FASTA parse included" %
                                                 (ns * per, " (pairs, -1/-2)" if a.paired else "", procs, a.cpu_threads, det["wall_s"], det["index_load_s"]), **det}
                # parity on the benchmark sample itself: GPU rows of shard 0 vs the reference's TSV
                parts = [b.results() for b in batches]
                rows, n_rows, score2 = (np.concatenate([x[i] for x in parts]) for i in range(3))
                nm = [bytes(x) for x in names[:n0]]
                got = rd.format_tsv(ix.seqid, nm, [a.read_len * per] * n0, rows[:n0], n_rows[:n0], score2[:n0])
                res["cpu_baseline"]["gpu_rows_identical_on_sample"] = (got == tsv0)
                res["cpu_baseline"]["parity_checked_reads"] = n0
            except Exception as e:          # the baseline is reported, never required for the metric
                res["cpu_baseline"] = {"value": None, "unit": "reads/s", "cores": nproc, "kind": "reference",
                                       "sample": "failed: %r" % (e,)}
        print(json.dumps(res))
    for b in batches:
        b.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
