#!/usr/bin/env python3
"""bench.py — classified reads/s of the MI355X-native classification path.

A "step" is one pass of the hot path over one batch of synthetic reads.  `value` is timed as the bench contract prescribes: the
inputs — PACKED reads (2-bit words + N mask, lengths, seeds) — are RESIDENT IN HBM when the timed region starts (every slot
holds the read set its warm-up steps uploaded); a step = plan + every kernel of a batch (cf_batch_reclassify_async) + its printed
rows and per-query columns downloaded into pinned host memory, `--inflight` slots (default 3) in flight: kernels on one stream,
downloads on another; the index is resident in HBM.  Beside it, in the same run and the same JSON line, `host_to_host`: the same
K steps through the whole asynchronous slot ABI with the reads coming from pinned host memory (upload, kernels, download on three
streams — SURVEY.md 8(d)'s scope, what rounds 1-3 reported as `value`): the PCIe-inclusive rate, bound by the host link where
the kernels outrun it.  Since round 5 both legs use the narrow forms of the boundary by default (`--wire narrow`: cf_dense_reads
in — four bases per byte, one length for the batch — and CF_RESULTS_NARROW out — 16-byte rows, five bytes per query: 29 + 21 bytes
per 100-base read across PCIe instead of 40 + 36; `--wire wide` = the word form and cf_row).  The per-kernel HIP-event times behind `roofline` come from the timed steps (all kernels of all slots
share one stream, so a batch's event intervals are its kernels' own durations) and, in `device_resident`, from one slot alone.

N > 1: one process per GPU (torchrun), the index replicated per GPU, reads sharded (each rank classifies its own
batches: weak scaling), ONE collective inside the timed region — the RCCL all-reduce over xGMI of the dense
per-taxon counters — and, after it, the per-rank report images merged on rank 0 (observed tuples for the EM).
Rank 0 prints one JSON line (contract in the task brief).

Workloads (`--config`, BASELINE.json `configs`): 2 = "p_compressed (~4.2 GB) + 10M synthetic 100 bp SE reads on
1 x MI355X" — the headline; 4 = p+h+v scale, 2 x 150 bp pairs; 5 = nt scale, 250 bp reads; 2r = config 2 on a
repeat-rich stand-in (strain clusters, shared operons, low-complexity tracts; cf_index_open finds it repeat-rich and finishes
small search ranges against the text; 2r- = the same with that switched off).  The real indexes cannot be
downloaded here, so each is a synthetic stand-in of the same size class (SURVEY.md §8d recipe), generated on the
GPU and built inside the run by our own GPU builder (cf_build_index; byte-identical to the reference's
centrifuge-build, tests/test_gpu_build.py).  Config 2's uids start with "cid" like p_compressed's, so the index is
`compressed()` (bt2_idx.h:648-663) and the classifier runs with ihits = 20 as it does on the real one.  Index
construction is outside the timed path.
"""
import argparse
import json
import os
import shutil
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "tools")):
    if p not in sys.path:
        sys.path.insert(0, p)

# SURVEY.md 8(d) byte model of the reference's step-by-step algorithm, per read, on the preset's workload (op counts of the
# round-1 kernels, which did exactly those steps: ftab 6.52, pair 51.17, pair2 22.47, single 67.61, walk 21.24, rows 1.415)
REF_MODEL = {"2": {"bytes_per_read": 128 * (51.1746692 + 22.4683682 + 67.6126641 + 21.2351075) + 16 * 6.5204704 + 2 * 1.4154131 + 25 + 13 + 32,
                   "search_bytes_per_read": 128 * (51.1746692 + 22.4683682 + 67.6126641) + 16 * 6.5204704 + 25 + 13}}
HBM_PEAK_GBPS = 8000.0          # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
NAME_DIGITS = 9


def log(*a):
    if int(os.environ.get("RANK", "0")) == 0:
        print("[bench]", *a, file=sys.stderr, flush=True)


# ------------------------------------------------------------------ synthetic data (on the GPU, torch = plumbing)
def gpu_genomes(torch, n_genomes, length, genus_size=8, divergence=0.05, seed=12345, recipe="iid"):
    """[n_genomes, length] base codes 0..3 on the current device.
    recipe "iid": genera of `genus_size` members, each member = the genus ancestor with `divergence` substitutions.
    recipe "repeat" (what real bacterial collections look like to an FM index: SA ranges stay wide for long):
    genera of 8 as above at 5 %, but every genome comes as a cluster of 4 near-identical STRAINS (0.1-1 % apart:
    half of the genomes are strains of another), 64 shared 5 kb "operons" are pasted into 10 % of the genomes each
    (cross-genus repeats), and 0.5 % of every genome is low-complexity tracts (homopolymers, dinucleotide repeats)."""
    gen = torch.Generator(device="cuda")
    gen.manual_seed(seed)
    dev = torch.empty((n_genomes, length), dtype=torch.uint8, device="cuda")
    strain = 4 if recipe == "repeat" else 1
    for g0 in range(0, n_genomes, genus_size):
        m = min(genus_size, n_genomes - g0)
        anc = torch.randint(0, 4, (length,), dtype=torch.uint8, device="cuda", generator=gen)
        # the genus at once: every member = the ancestor with its own substitutions
        mut = torch.rand((m, length), device="cuda", generator=gen) < divergence
        add = torch.randint(1, 4, (m, length), dtype=torch.uint8, device="cuda", generator=gen)
        mem = (anc[None, :] + add * mut) & 3
        if strain > 1:                                    # members 1..3 of every cluster of 4 become strains of member 0
            for j in range(m):
                if j % strain:
                    d = 0.001 * (1 + 3 * (j % strain))
                    mu = torch.rand(length, device="cuda", generator=gen) < d
                    ad = torch.randint(1, 4, (length,), dtype=torch.uint8, device="cuda", generator=gen)
                    mem[j] = (mem[j - j % strain] + ad * mu) & 3
        dev[g0:g0 + m] = mem
    if recipe == "repeat":
        n_op, op_len = 64, 5000
        ops = torch.randint(0, 4, (n_op, op_len), dtype=torch.uint8, device="cuda", generator=gen)
        per = max(1, n_genomes // 10)
        for o in range(n_op):
            gs = torch.randint(0, n_genomes, (per,), device="cuda", generator=gen)
            ps = torch.randint(0, length - op_len, (per,), device="cuda", generator=gen)
            idx = ps[:, None] + torch.arange(op_len, device="cuda")[None, :]
            dev[gs[:, None], idx] = ops[o][None, :]
        n_tr = max(1, int(0.005 * length / 200))          # tracts of ~200 bp
        ar = torch.arange(400, device="cuda")
        for gi in range(n_genomes):
            ps = torch.randint(0, length - 400, (n_tr,), device="cuda", generator=gen)
            ln = torch.randint(50, 400, (n_tr,), device="cuda", generator=gen)
            a = torch.randint(0, 4, (n_tr,), dtype=torch.uint8, device="cuda", generator=gen)
            b = torch.randint(0, 4, (n_tr,), dtype=torch.uint8, device="cuda", generator=gen)
            di = torch.rand(n_tr, device="cuda", generator=gen) < 0.5
            pat = torch.where(di[:, None] & (ar[None, :] % 2 == 1), b[:, None], a[:, None])
            keep = ar[None, :] < ln[:, None]
            idx = ps[:, None] + ar[None, :]
            row = dev[gi]
            row[idx[keep]] = pat[keep]
    return dev


def gpu_sample_reads(torch, genomes, n_reads, read_len, seed, mut_frac=0.63, random_frac=0.01, n_frac=0.001):
    """SURVEY §8(d) read recipe -> codes [n_reads, read_len] (0..4, on the device): uniform over genomes
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
    return out


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
    return out


def gpu_pack(torch, codes):
    """[n, L] base codes on the device -> (bases int64 [n * W], nmask int64 [n * W]) in the packed layout of
    cf_packed_reads (W = ceil(L / 32) words per read; bit patterns, so the sign of the int64 means nothing)"""
    n, L = codes.shape
    W = (L + 31) // 32
    sh2 = 2 * torch.arange(32, device="cuda", dtype=torch.int64)
    sh1 = torch.arange(32, device="cuda", dtype=torch.int64)
    bases = torch.empty(n * W, dtype=torch.int64, device="cuda")
    nmask = torch.empty(n * W, dtype=torch.int64, device="cuda")
    CH = 1 << 20
    for s in range(0, n, CH):
        e = min(n, s + CH)
        pad = torch.zeros((e - s, W * 32), dtype=torch.uint8, device="cuda")
        pad[:, :L] = codes[s:e]
        isn = pad > 3
        c = torch.where(isn, torch.zeros_like(pad), pad).to(torch.int64).view(e - s, W, 32)
        bases[s * W:e * W] = (c << sh2).sum(dim=2).reshape(-1)
        nmask[s * W:e * W] = (isn.view(e - s, W, 32).to(torch.int64) << sh1).sum(dim=2).reshape(-1)
    return bases, nmask


def gpu_dense(torch, codes):
    """[n, L] base codes on the device -> uint8 [n * ceil(L / 4)]: cf_dense_reads' four bases per byte, every read on a byte (an N
    carries code 0; its bit travels in the sparse N mask)"""
    n, L = codes.shape
    bpr = (L + 3) // 4
    out = torch.empty(n * bpr, dtype=torch.uint8, device="cuda")
    CH = 1 << 21
    for s in range(0, n, CH):
        e = min(n, s + CH)
        pad = torch.zeros((e - s, bpr * 4), dtype=torch.uint8, device="cuda")
        pad[:, :L] = torch.where(codes[s:e] > 3, torch.zeros_like(codes[s:e]), codes[s:e])
        q = pad.view(e - s, bpr, 4)
        out[s * bpr:e * bpr] = (q[:, :, 0] | (q[:, :, 1] << 2) | (q[:, :, 2] << 4) | (q[:, :, 3] << 6)).reshape(-1)
    return out


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
    1-read file and subtracted.  Returns (reads/s, the TSV of the whole sample = the shards' bodies in
    order, queries in it, details)."""
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
    parts = [open(os.path.join(workdir, "cpu_run_%d.tsv" % i)).read() for i in range(len(shards))]
    hdr = parts[0][:parts[0].index("\n") + 1]
    tsv = hdr + "".join(x[len(hdr):] for x in parts)
    return n / search, tsv, n, {"wall_s": t_all, "index_load_s": t_load, "search_s": search}


def cli_end_to_end(torch, base, workdir, P, n_genomes, genome_len, n_total, nproc, local, have_cpu_shard, paired):
    """The product a user runs — centrifuge-class, FASTA in, TSV + report out — on `n_total` reads of the preset's recipe against
    the preset's index (outside every timed region; the parent has let go of its own index replica).  Parity first: the binary's
    TSV and report on shard 0 of the CPU leg's sample, byte for byte against what the reference wrote for that shard.  Then the
    big file: wall time of the whole process (start, index open, every read classified, files written), the stage seconds the
    binary reports itself (-t), and the rate of the classification proper (wall less the index open)."""
    import re
    exe = os.path.join(ROOT, "centrifuge_amd", "bin", "centrifuge-class")
    out = {"reads": 0, "threads": nproc}
    env = dict(os.environ, HIP_VISIBLE_DEVICES=os.environ.get("HIP_VISIBLE_DEVICES", ""))
    if not env["HIP_VISIBLE_DEVICES"]:
        env.pop("HIP_VISIBLE_DEVICES")
    if have_cpu_shard and not paired:
        fa, want_tsv, want_rep = (os.path.join(workdir, f) for f in ("cpu_0.fa", "cpu_run_0.tsv", "cpu_run_0.rep"))
        r = subprocess.run([exe, "-f", "-p", str(nproc), "--device", str(local), "-k", "5", "-x", base, "-U", fa, "-S", os.path.join(workdir, "cli_0.tsv"),
                            "--report-file", os.path.join(workdir, "cli_0.rep")], capture_output=True, text=True, env=env, timeout=600)
        ok = r.returncode == 0
        out["sample_reads"] = max(0, len(open(fa, "rb").read().splitlines()) // 2)
        out["tsv_identical_to_reference_on_sample"] = ok and open(os.path.join(workdir, "cli_0.tsv")).read() == open(want_tsv).read()
        out["report_identical_to_reference_on_sample"] = ok and open(os.path.join(workdir, "cli_0.rep")).read() == open(want_rep).read()
        if not ok:
            out["sample_error"] = (r.stderr or "")[-300:]
    # the big file: reads of the preset's recipe, written 10 M at a time (on a RAM disk when the host can spare it)
    t0 = time.time()
    genomes = gpu_genomes(torch, n_genomes, genome_len, recipe=P["recipe"])
    read_len = P["read_len"]
    big_dir = "/dev/shm/cf_bench_e2e" if shutil.disk_usage("/dev/shm").free > 4 * n_total * (read_len + 13) else workdir
    os.makedirs(big_dir, exist_ok=True)
    fa = os.path.join(big_dir, "e2e_reads.fa")
    done = 0
    with open(fa, "wb") as f:
        while done < n_total:
            n = min(10000000, n_total - done)
            codes = gpu_sample_reads(torch, genomes, n, read_len, seed=4242 + done).cpu().numpy()
            names = read_names(n) if done == 0 else read_names(done + n)[done:]
            tmp = os.path.join(big_dir, "e2e_part.fa")
            write_fasta(tmp, names, codes)
            with open(tmp, "rb") as g:
                shutil.copyfileobj(g, f, 64 << 20)
            os.remove(tmp)
            done += n
            del codes
    del genomes
    torch.cuda.empty_cache()
    out["fasta_bytes"] = os.path.getsize(fa)
    out["generate_s"] = time.time() - t0
    tsv = os.path.join(big_dir, "e2e.tsv")
    t0 = time.time()
    r = subprocess.run([exe, "-f", "-t", "-p", str(nproc), "--device", str(local), "-x", base, "-U", fa, "-S", tsv, "--report-file", os.path.join(big_dir, "e2e.rep")],
                       capture_output=True, text=True, env=env, timeout=900)
    wall = time.time() - t0
    out.update({"reads": n_total, "wall_s": wall, "reads_per_s_whole_process": n_total / wall, "rc": r.returncode})
    err = r.stderr or ""
    m = re.search(r"Time loading forward index: (\d+):(\d+):(\d+)", err)
    st = [l for l in err.splitlines() if l.startswith("Stage seconds")]
    if st:
        out["stage_seconds"] = st[-1][len("Stage seconds: "):]
        mo = re.search(r"index open ([0-9.]+)", st[-1])
        if mo:
            out["index_open_s"] = float(mo.group(1))
            out["reads_per_s_after_index_open"] = n_total / max(1e-3, wall - float(mo.group(1)))
    elif m:
        out["index_open_s_rounded"] = int(m.group(1)) * 3600 + int(m.group(2)) * 60 + int(m.group(3))
    tb = [l for l in err.splitlines() if l.startswith("Index tables:")]
    if tb:
        out["index_tables"] = tb[-1][len("Index tables: "):]
    dt = [l for l in err.splitlines() if l.startswith("Device text path:")]
    if dt:
        out["device_text_path"] = dt[-1][len("Device text path: "):]
    mo_ = re.search(r"Overall seconds: ([0-9.]+) .*before the index open ([0-9.]+)", err)
    if mo_:
        out["seconds_inside_the_call"] = float(mo_.group(1))
        out["seconds_before_the_index_open"] = float(mo_.group(2))
        out["seconds_outside_the_call"] = wall - float(mo_.group(1))        # process start (loader, HIP runtime load) + exit (HIP runtime shutdown)

    def digest(path):
        import hashlib
        h = hashlib.md5()
        with open(path, "rb") as f_:
            for blk in iter(lambda: f_.read(64 << 20), b""):
                h.update(blk)
        return h.hexdigest()
    big_md5 = None
    if r.returncode == 0:
        out["tsv_bytes"] = os.path.getsize(tsv)
        out["tsv_rows"] = sum(1 for _ in open(tsv, "rb")) - 1
        big_md5 = (digest(tsv), digest(os.path.join(big_dir, "e2e.rep")))
    else:
        out["error"] = err[-300:]
    # Round 6: the index is planned for the JOB (cf_index_options::expected_reads, estimated by the binary from its input's size).
    # Beside the run above: the same file with --expected-reads 0 (every table that fits: what rounds 1 - 5 always made), and a
    # small job — the file's first million reads — whose wall time is what a user with one sample waits for; the reference's
    # time for that job follows from the CPU leg (its index load + 1 M reads at its rate)
    def timed(args, key, env2=None):
        t1 = time.time()
        rr = subprocess.run([exe, "-f", "-t", "-p", str(nproc), "--device", str(local), "-x", base] + args, capture_output=True, text=True, env=dict(env, **(env2 or {})), timeout=900)
        w = time.time() - t1
        e2 = rr.stderr or ""
        o2 = {"wall_s": w, "rc": rr.returncode}
        mo2 = re.search(r"Stage seconds: index open ([0-9.]+), search wall ([0-9.]+)", e2)
        if mo2:
            o2["index_open_s"] = float(mo2.group(1))
            o2["search_wall_s"] = float(mo2.group(2))
        d2 = [l for l in e2.splitlines() if l.startswith("Device text path:")]
        if d2:
            o2["device_text_path"] = d2[-1][len("Device text path: "):]
        t2 = [l for l in e2.splitlines() if l.startswith("Index tables:")]
        if t2:
            o2["index_tables"] = t2[-1][len("Index tables: "):]
        out[key] = o2
    try:
        # the same run through the host's parser pool and formatter threads (what rounds 3 - 5 had; CF_CLI_DEVICE_TEXT is a debug knob):
        # its wall time, and its files against the device text path's, byte for byte
        timed(["-U", fa, "-S", tsv, "--report-file", os.path.join(big_dir, "e2e.rep")], "host_parser_and_formatter", {"CF_DEBUG_KNOBS": "1", "CF_CLI_DEVICE_TEXT": "0"})
        hp = out["host_parser_and_formatter"]
        if hp["rc"] == 0 and big_md5:
            hp["same_tsv_and_report_as_the_device_text_path"] = (digest(tsv), digest(os.path.join(big_dir, "e2e.rep"))) == big_md5
        if "index_open_s" in hp:
            hp["reads_per_s_after_index_open"] = n_total / max(1e-3, hp["wall_s"] - hp["index_open_s"])
        # ... and with the rows going nowhere (-S /dev/null): what the text path delivers when no output file's lock holds it back
        timed(["-U", fa, "-S", "/dev/null", "--report-file", os.path.join(big_dir, "e2e.rep")], "output_to_dev_null")
        if "search_wall_s" in out["output_to_dev_null"]:
            out["output_to_dev_null"]["reads_per_s_in_the_search_phase"] = n_total / max(1e-3, out["output_to_dev_null"]["search_wall_s"])
        # (builder's sweeps: CF_BENCH_CLI_SWEEP="name:extra args:ENV=v,ENV=v;..." — more runs of the same file, each under its name)
        for var in [v for v in os.environ.get("CF_BENCH_CLI_SWEEP", "").split(";") if v.strip()]:
            nm_, ar_, en_ = (var.split(":") + ["", ""])[:3]
            timed(["-U", fa, "-S", tsv, "--report-file", os.path.join(big_dir, "e2e.rep")] + ar_.split(), "sweep_" + nm_,
                  dict([("CF_DEBUG_KNOBS", "1")] + [tuple(kv.split("=", 1)) for kv in en_.split(",") if "=" in kv]))
        # (CF_BENCH_CLI_TRACE=<dir>: the same run with every table once more under rocprofv3's kernel trace, for profiles/)
        if os.environ.get("CF_BENCH_CLI_TRACE"):
            td = os.environ["CF_BENCH_CLI_TRACE"]
            subprocess.run(["rocprofv3", "--kernel-trace", "--stats", "--output-format", "csv", "-d", td, "-o", "t", "--", exe, "-f", "-t", "-p", str(nproc), "--device", str(local),
                            "-x", base, "-U", fa, "-S", tsv, "--report-file", os.path.join(big_dir, "e2e.rep"), "--expected-reads", "0"], capture_output=True, text=True, env=env, timeout=900, cwd="/tmp")
        timed(["-U", fa, "-S", tsv, "--report-file", os.path.join(big_dir, "e2e.rep"), "--expected-reads", "0"], "every_table")
        out["every_table"]["reads_per_s_whole_process"] = n_total / out["every_table"]["wall_s"]
        if "index_open_s" in out["every_table"]:
            out["every_table"]["reads_per_s_after_index_open"] = n_total / max(1e-3, out["every_table"]["wall_s"] - out["every_table"]["index_open_s"])
        small = os.path.join(big_dir, "e2e_small.fa")
        n_small = min(1000000, n_total)
        with open(fa, "rb") as f, open(small, "wb") as g:
            for _ in range(2 * n_small):
                g.write(f.readline())
        timed(["-U", small, "-S", tsv, "--report-file", os.path.join(big_dir, "e2e.rep")], "small_job")
        out["small_job"]["reads"] = n_small
        os.remove(small)
        # mates (-1 / -2) through the text path: 10 M pairs of the preset's reads (any two reads make a pair here: the run is timed and
        # compared with the host threads' output, not with a truth), against the same run with --host-io
        n_pairs = min(10000000, n_total // 5)
        if n_pairs >= 1000 and not paired:
            genomes = gpu_genomes(torch, n_genomes, genome_len, recipe=P["recipe"])
            mates = []
            for m_ in (1, 2):
                codes = gpu_sample_reads(torch, genomes, n_pairs, read_len, seed=777 + m_).cpu().numpy()
                f_m = os.path.join(big_dir, "e2e_m%d.fa" % m_)
                write_fasta(f_m, read_names(n_pairs), codes)
                mates.append(f_m)
                del codes
            del genomes
            torch.cuda.empty_cache()
            timed(["-1", mates[0], "-2", mates[1], "-S", tsv, "--report-file", os.path.join(big_dir, "e2e.rep")], "mates")
            md_ = (digest(tsv), digest(os.path.join(big_dir, "e2e.rep"))) if out["mates"]["rc"] == 0 else None
            timed(["-1", mates[0], "-2", mates[1], "-S", tsv, "--report-file", os.path.join(big_dir, "e2e.rep"), "--host-io"], "mates_host_io")
            out["mates"]["pairs"] = n_pairs
            out["mates"]["same_tsv_and_report_as_host_io"] = md_ is not None and out["mates_host_io"]["rc"] == 0 and md_ == (digest(tsv), digest(os.path.join(big_dir, "e2e.rep")))
            for k_ in ("mates", "mates_host_io"):
                if "search_wall_s" in out[k_]:
                    out[k_]["mates_per_s_in_the_search_phase"] = 2 * n_pairs / max(1e-3, out[k_]["search_wall_s"])
            for f_m in mates:
                os.remove(f_m)
    except Exception as e:          # noqa: BLE001  (the legs above are extras: the line is printed without them)
        out["extras_failed"] = repr(e)
    for f_ in (fa, tsv):
        try:
            os.remove(f_)
        except OSError:
            pass
    return out


def bind_near_gpu(torch, local):
    """Run on the CPUs of the NUMA node the GPU hangs off (what `numactl --cpunodebind` does for one rank per GPU): pinned host
    buffers are placed by first touch, and copies to and from a remote node's memory run at ~60 % of the local rate — round 5 saw
    the host-to-host rate of one and the same command come out at 1.00 or 1.18e9 reads/s from process to process.  Returns the
    affinity to restore (the CPU baseline and the command line's run take every core again), or None when the topology is not
    to be had (then nothing changes)."""
    try:
        pr = torch.cuda.get_device_properties(local)
        bdf = "%04x:%02x:%02x.0" % (getattr(pr, "pci_domain_id", 0), pr.pci_bus_id, pr.pci_device_id)
        node = int(open("/sys/bus/pci/devices/%s/numa_node" % bdf).read())
        if node < 0:
            return None
        cpus = set()
        for part in open("/sys/devices/system/node/node%d/cpulist" % node).read().strip().split(","):
            lo, _, hi = part.partition("-")
            cpus.update(range(int(lo), int(hi or lo) + 1))
        old = os.sched_getaffinity(0)
        want = cpus & old
        if not want or want == old:
            return None
        os.sched_setaffinity(0, want)
        log("bound to the %d CPUs of NUMA node %d (GPU %s)" % (len(want), node, bdf))
        return old
    except Exception:
        return None


def host_link(torch, local):
    """The GPU's PCIe link as sysfs reports it (speed, width, NUMA node) — round 5 saw the host-to-host rate of one command come
    out at 1.00 or 1.18e9 reads/s by the box's GPU slot; with the link on record a slow run can be told from a slow link."""
    out = {}
    try:
        pr = torch.cuda.get_device_properties(local)
        bdf = "%04x:%02x:%02x.0" % (getattr(pr, "pci_domain_id", 0), pr.pci_bus_id, pr.pci_device_id)
        out["bdf"] = bdf
        for k in ("current_link_speed", "current_link_width", "max_link_speed", "max_link_width", "numa_node"):
            try:
                out[k] = open("/sys/bus/pci/devices/%s/%s" % (bdf, k)).read().strip()
            except OSError:
                pass
        # the narrowest link on the way up to the root complex (a switch or a bifurcated slot caps what the endpoint negotiates)
        path = os.path.realpath("/sys/bus/pci/devices/%s" % bdf)
        narrowest = None
        while path and path != "/" and "pci" in path:
            try:
                w, sp = int(open(path + "/current_link_width").read()), open(path + "/current_link_speed").read().strip()
                gts = float(sp.split()[0])
                if narrowest is None or w * gts < narrowest[0]:
                    narrowest = (w * gts, "%s x%d at %s" % (os.path.basename(path), w, sp))
            except (OSError, ValueError):
                pass
            path = os.path.dirname(path)
        if narrowest:
            out["narrowest_hop"] = narrowest[1]
    except Exception as e:      # noqa: BLE001
        out["error"] = repr(e)
    return out


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


def kernel_source_sha():
    import hashlib
    h = hashlib.sha256()
    for f in ("cf_kernels.hpp", "cf_device.hip"):
        h.update(open(os.path.join(ROOT, "centrifuge_amd", "csrc", f), "rb").read())
    return h.hexdigest()


def run_other_configs(presets, steps, budget_s):
    """`bench.py --config X` for the other presets, one child process each (own index, own HBM), a short timed region; the
    child's JSON line is cut down to what compares the workloads: rate, step, kernel times, requests per read, parity."""
    out = {}
    t_start = time.time()
    # what a preset is expected to take on one MI355X box (genomes + build + load + CPU reference + timed steps), seconds
    expect = {"2r": 240, "2r-": 150, "4": 300, "5": 660, "2": 150}
    for c in presets:
        c = c.strip()
        left = budget_s - (time.time() - t_start)
        budget_gb = None
        key = c
        if "@" in c:                                   # "2@96": the preset under --hbm-budget-gb 96 (one point of the budget -> throughput curve)
            c, _, g_ = c.partition("@")
            try:
                budget_gb = float(g_)
            except ValueError:
                out[key] = {"skipped": "bad budget"}
                continue
        if c not in PRESETS or (c == "2" and budget_gb is None):
            out[key] = {"skipped": "unknown preset"}
            continue
        if left < expect.get(c, 300):
            out[key] = {"skipped": "would not fit the remaining %.0f s of the other-configs budget" % left}
            continue
        # (warmup = two batches per slot in flight: by then a slot's buffers are allocated and its row download is sized by what the
        #  workload prints — where a read prints more than 1.25 rows, a slot's first batch fetches the rest synchronously into a
        #  re-allocated pinned buffer and its second re-allocates once more for the margin; with 2 warm-up steps for 3 slots all
        #  of that fell into the 8 timed steps of the repeat-rich preset: 27-30 ms per step instead of 19.9)
        cmd = [sys.executable, os.path.abspath(__file__), "--config", c, "--steps", str(steps), "--warmup", "6", "--other-configs", "",
               "--cpu-sample", os.environ.get("CF_BENCH_OTHER_CPU_SAMPLE", "200000")]
        if budget_gb is not None:
            cmd += ["--hbm-budget-gb", str(budget_gb)]
        for k_ in ("genomes", "genome_len", "reads"):                  # CF_BENCH_OTHER_GENOMES_2r=512 ...: smaller stand-ins (tests)
            v_ = os.environ.get("CF_BENCH_OTHER_%s_%s" % (k_.upper(), c))
            if v_:
                cmd += ["--" + k_.replace("_", "-"), v_]
        t0 = time.time()
        log("other config %s: %s" % (key, " ".join(cmd[2:])))
        try:
            env = dict(os.environ)
            for k_ in ("CF_BENCH_GENOMES", "CF_BENCH_GENOME_LEN", "CF_BENCH_READS", "CF_BENCH_CONFIG", "CF_BENCH_FORCE_DIST", "CF_BENCH_HBM_BUDGET_GB",
                       "RANK", "LOCAL_RANK", "WORLD_SIZE", "GROUP_RANK", "ROLE_RANK", "LOCAL_WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT",
                       "TORCHELASTIC_RUN_ID", "TORCHELASTIC_RESTART_COUNT", "TORCHELASTIC_MAX_RESTARTS"):
                env.pop(k_, None)                      # a child is a plain one-GPU run, whatever launched the parent
            r = subprocess.run(cmd, capture_output=True, text=True, timeout=min(left, 2.5 * expect.get(c, 300)), env=env)
            line = [x for x in r.stdout.splitlines() if x.startswith("{")]
            if r.returncode != 0 or not line:
                out[key] = {"failed": "rc %d: %s" % (r.returncode, (r.stderr or "")[-400:]), "wall_s": time.time() - t0}
                continue
            j = json.loads(line[-1])
            ops = j.get("ops_per_read", {})
            cpu = j.get("cpu_baseline", {})
            out[key] = {"workload": j["config"]["workload"], "hbm_budget_gb": budget_gb, "index_resident_gb": j["config"]["index_bytes"] / 1e9, "value": j["value"], "unit": "mates/s" if PRESETS[c]["paired"] else "reads/s",
                      "ms_per_step": j["ms_per_step"], "host_to_host": j.get("host_to_host"), "steps": j["steps"], "reads_per_step": j["config"]["reads_per_gpu_per_step"],
                      "read_len": j["config"]["read_len"], "kernels_ms": j["kernels_ms"],
                      "kernels_ms_one_slot_alone": j["device_resident"]["blocking_api_kernels_ms"],
                      "requests_per_read": sum(ops.get(k_, 0) for k_ in ("ftab", "pair", "pair2", "single", "ftab_wide", "text_loads", "walk")) + 2 * ops.get("verify", 0) - ops.get("pos_hits", 0) + 2,
                      "ops_per_read": ops, "general_kernel_queries": j.get("general_kernel_queries"), "index_bytes": j["config"]["index_bytes"], "index_build_s_gpu": j["config"]["index_build_s_gpu"],
                      "derived_tables": {k_: j["config"].get(k_) for k_ in ("occ_planes", "pair_planes", "wide_ftab_chars", "text_verify_sample_every_nth", "resolve_table_every_nth_row")},
                      "index_options": j["config"].get("index_options"), "small_range_rows_in_effect": j["config"].get("small_range_rows_in_effect"),
                      "repeat_fraction": j["config"].get("repeat_fraction"), "plan_realised": j["config"].get("plan_realised"),
                      "roofline_traffic": j["roofline"].get("traffic"),
                      "search_roofline_frac": j["roofline"]["frac"], "search_frac_of_measured_request_rate": j["roofline"].get("frac_of_measured_request_rate"),
                      "cpu_reference_reads_per_s": cpu.get("value"), "parity_checked_reads": cpu.get("parity_checked_reads"),
                      "gpu_rows_identical": cpu.get("gpu_rows_identical_on_sample"), "wall_s": time.time() - t0}
        except subprocess.TimeoutExpired:
            out[key] = {"failed": "timed out", "wall_s": time.time() - t0}
        except Exception as e:
            out[key] = {"failed": repr(e), "wall_s": time.time() - t0}
    return out


PRESETS = {
    # BASELINE.json configs -> stand-ins of the same size class (genomes x length, reads per GPU per step, read length, pairs, uid prefix, recipe)
    "2": dict(genomes=2048, genome_len=4194304, reads=10000000, read_len=100, paired=False, uid="cid|", recipe="iid",
              what="config 2: p_compressed stand-in (compressed index: uids start with cid)"),
    # (every index is opened with all options automatic.  On the repeat-rich collection cf_index_open's probe finds neighbouring
    #  suffix-array rows sharing their preceding bases and the planner makes the SA / inverse-SA samples at every row: ranges of up
    #  to four relatives are finished against the text (cf_index_options::small_range_rows).  "2r-" is the same workload with that
    #  switched off, so that the line shows both; --small-range-rows overrides either way)
    "2r": dict(genomes=2048, genome_len=4194304, reads=10000000, read_len=100, paired=False, uid="cid|", recipe="repeat",
               what="config 2 on a repeat-rich stand-in (strain clusters at 0.1-1 %, shared 5 kb operons, low-complexity tracts)"),
    "2r-": dict(genomes=2048, genome_len=4194304, reads=10000000, read_len=100, paired=False, uid="cid|", recipe="repeat",
                what="config 2 on the repeat-rich stand-in, small ranges against the text switched off (small_range_rows = -1)",
                index_opts=dict(small_range_rows=-1)),
    "4": dict(genomes=6144, genome_len=4194304, reads=10000000, read_len=150, paired=True, uid="seq", recipe="iid",
              what="config 4: p+h+v stand-in, 2 x 150 bp FR pairs (mates counted)"),
    "5": dict(genomes=24576, genome_len=4194304, reads=4000000, read_len=250, paired=False, uid="seq", recipe="iid",
              what="config 5: nt-scale stand-in, 250 bp reads"),
    # not part of the default run: config 4's reads on a repeat-rich text of the size that still affords the samples at every row
    "4r": dict(genomes=2048, genome_len=4194304, reads=10000000, read_len=150, paired=True, uid="cid|", recipe="repeat",
               what="2 x 150 bp FR pairs (config 4's reads) on the 8.6 Gbp repeat-rich stand-in"),
    "4r-": dict(genomes=2048, genome_len=4194304, reads=10000000, read_len=150, paired=True, uid="cid|", recipe="repeat",
                what="2 x 150 bp FR pairs on the 8.6 Gbp repeat-rich stand-in, small ranges against the text switched off",
                index_opts=dict(small_range_rows=-1)),
}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", default=os.environ.get("CF_BENCH_CONFIG", "2"), choices=sorted(PRESETS))
    ap.add_argument("--genomes", type=int, default=int(os.environ.get("CF_BENCH_GENOMES", 0)))
    ap.add_argument("--genome-len", type=int, default=int(os.environ.get("CF_BENCH_GENOME_LEN", 0)))
    ap.add_argument("--reads", type=int, default=int(os.environ.get("CF_BENCH_READS", 0)), help="reads per GPU per step")
    ap.add_argument("--read-len", type=int, default=0)
    ap.add_argument("--inflight", type=int, default=int(os.environ.get("CF_BENCH_INFLIGHT", 3)), help="batch slots in flight")
    ap.add_argument("--cpu-sample", type=int, default=int(os.environ.get("CF_BENCH_CPU_SAMPLE", 1000000)))
    ap.add_argument("--cpu-threads", type=int, default=8, help="threads per reference process")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--dense-nmask", action="store_true", help="upload the N mask word for word instead of the words that hold an N")
    ap.add_argument("--wire", default=os.environ.get("CF_BENCH_WIRE", "narrow"), choices=["narrow", "wide"],
                    help="what crosses the host link: narrow = cf_dense_reads in (four bases per byte, no length array) and CF_RESULTS_NARROW out "
                         "(16-byte rows, 5 bytes per query); wide = the word form in (cf_packed_reads) and cf_row + three words per query out")
    ap.add_argument("--small-range-rows", type=int, default=None, help="cf_index_options::small_range_rows (default: the preset's, i.e. automatic; -1 = off)")
    ap.add_argument("--other-configs", default=os.environ.get("CF_BENCH_OTHER", "2r,2r-,4,5,2@96"),
                    help="presets run briefly after the headline (config 2, one GPU) and attached to the same JSON line as other_configs; '' = none")
    ap.add_argument("--hbm-budget-gb", type=float, default=float(os.environ.get("CF_BENCH_HBM_BUDGET_GB", 0)),
                    help="device memory the index may take, files + derived tables (cf_index_open_ex); 0 = what is free")
    ap.add_argument("--cli-reads", type=int, default=int(os.environ.get("CF_BENCH_CLI_READS", 50000000)),
                    help="reads of the end-to-end leg (centrifuge-class on a FASTA file of the preset's reads against the preset's index, outside the timed region; 0 = skip)")
    ap.add_argument("--other-steps", type=int, default=int(os.environ.get("CF_BENCH_OTHER_STEPS", 20)))
    ap.add_argument("--other-budget-s", type=float, default=float(os.environ.get("CF_BENCH_OTHER_BUDGET_S", 1300)),
                    help="wall-clock budget of all other_configs runs together (a preset that would not fit is skipped and says so)")
    a = ap.parse_args()
    P = dict(PRESETS[a.config])
    if a.small_range_rows is not None:
        P["index_opts"] = dict(P.get("index_opts") or {}, small_range_rows=a.small_range_rows)
    for k_, v_ in (("genomes", a.genomes), ("genome_len", a.genome_len), ("reads", a.reads), ("read_len", a.read_len)):
        if v_:
            P[k_] = v_
    n_genomes, genome_len, n_reads, read_len, paired = P["genomes"], P["genome_len"], P["reads"], P["read_len"], P["paired"]

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
        import datetime
        # (rank 0 builds the index while the others wait at the first barrier: 80 s of GPU build + 47 GB of files for config 5;
        # the default watchdog of 10 minutes is too close to that on a loaded box)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local), timeout=datetime.timedelta(minutes=60))
    nproc = effective_cores()
    old_affinity = bind_near_gpu(torch, local)

    # where the stand-in's index files go: half a byte per base.  The boxes' root overlay holds ~79 GB (the 47 GB of config 5 did
    # not fit it beside the rest), their /dev/shm 1.5 TB of a 3 TB host: a large index goes there when the host can spare it; if
    # nothing holds it the stand-in is scaled down to what the largest place takes, and the workload string says so
    need = int(0.5 * n_genomes * genome_len) + (2 << 30)
    scaled_note = ""
    root_dir = os.environ.get("CF_BENCH_DIR") or tempfile.gettempdir()
    os.makedirs(root_dir, exist_ok=True)

    def free_of(d_):
        try:
            return shutil.disk_usage(d_).free
        except OSError:
            return 0
    try:
        mem_avail = int([l for l in open("/proc/meminfo") if l.startswith("MemAvailable")][0].split()[1]) * 1024
    except Exception:
        mem_avail = 0
    if need > (8 << 30) and free_of("/dev/shm") > 2 * need and mem_avail > need + (256 << 30):
        root_dir = os.path.join("/dev/shm", "cf_bench")
        os.makedirs(root_dir, exist_ok=True)
    elif free_of(root_dir) < 1.3 * need:
        fit = int(0.7 * free_of(root_dir) / (0.5 * genome_len))
        if fit < 8:
            raise SystemExit("bench: no room for the index files under %s" % root_dir)
        scaled_note = " [scaled down from %d genomes: %s holds %.0f GB]" % (n_genomes, root_dir, free_of(root_dir) / 1e9)
        n_genomes = fit // 8 * 8
    workdir = os.path.join(root_dir, "cf_bench_%d_%d_%s_%s" % (n_genomes, genome_len, P["recipe"], P["uid"].strip("|")))
    os.makedirs(workdir, exist_ok=True)
    base = os.path.join(workdir, "idx")
    have_index = all(os.path.exists(base + ".%d.cf" % k) for k in (1, 2, 3, 4))
    per = 2 if paired else 1
    if n_reads % per:
        raise SystemExit("bench: the number of reads must be even for pairs")
    S = max(1, a.inflight)

    # ---- synthetic genomes (every rank, same seed); reads are sampled from them further down
    t0 = time.time()
    genomes = gpu_genomes(torch, n_genomes, genome_len, recipe=P["recipe"])
    torch.cuda.synchronize()
    log("%d genomes x %d bp (%s) generated on the GPU in %.1fs" % (n_genomes, genome_len, P["recipe"], time.time() - t0))

    # ---- this rank's read sets, one per slot (seeded apart), packed on the GPU and parked in pinned host memory
    W = (read_len + 31) // 32
    narrow = a.wire == "narrow" and not a.dense_nmask
    sets, sample_codes = [], None
    t0 = time.time()
    for j in range(S):
        sd = 777 + 1000 * j + rank
        codes = gpu_sample_pairs(torch, genomes, n_reads // 2, read_len, seed=sd) if paired else gpu_sample_reads(torch, genomes, n_reads, read_len, seed=sd)
        if j == 0 and rank == 0 and not a.no_cpu:
            sample_codes = codes[:min(n_reads, a.cpu_sample // per * per)].cpu().numpy()
        bases_d, nmask_d = gpu_pack(torch, codes)
        pd = None
        if narrow:
            dd = gpu_dense(torch, codes)
            pd = capi.PinnedArray(capi.lib(), np.uint8, n_reads * ((read_len + 3) // 4))
            pd.a[:] = dd.cpu().numpy()
            del dd
        del codes
        pb = capi.PinnedArray(capi.lib(), np.uint64, n_reads * W)
        pl_, ps = capi.PinnedArray(capi.lib(), np.uint32, n_reads), capi.PinnedArray(capi.lib(), np.uint32, n_reads)
        pb.a[:] = bases_d.cpu().numpy().view(np.uint64)
        nm = nmask_d.cpu().numpy().astype(np.uint32)
        if a.dense_nmask:                      # the N mask word for word (n_words x 4 bytes across PCIe)
            pm = capi.PinnedArray(capi.lib(), np.uint32, n_reads * W)
            pm.a[:] = nm
            nw = None
        else:                                  # ... or only the words that hold an N (cf_packed_reads' sparse form)
            ni, nk = capi.sparse_nmask(nm)
            pni, pnk = capi.PinnedArray(capi.lib(), np.uint64, len(ni)), capi.PinnedArray(capi.lib(), np.uint32, len(ni))
            pni.a[:] = ni; pnk.a[:] = nk
            pm, nw = None, (pni, pnk)
        del nm
        pl_.a[:] = read_len
        ps.a[:] = 0
        del bases_d, nmask_d
        sets.append((pb, pm, pl_, ps, nw, pd))
    torch.cuda.synchronize()
    log("%d read sets of %d x %d bp sampled and packed on the GPU in %.1fs" % (S, n_reads, read_len, time.time() - t0))

    # ---- index: rank 0 builds it with the GPU builder, everybody loads its own HBM replica
    build_s = None
    if rank == 0 and not have_index:
        host = genomes.cpu().numpy()
        del genomes
        torch.cuda.empty_cache()
        synth.write_taxonomy(workdir, n_genomes, uid_prefix=P["uid"])
        names_g = [b"%s%d synthetic genome %d" % (P["uid"].encode(), i, i) for i in range(n_genomes)]
        goff = np.arange(n_genomes + 1, dtype=np.uint64) * np.uint64(genome_len)
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
    # the HBM the index may take: this process knows its slots (S of n_reads reads), so it offers the index what is free less
    # exactly those (cf_slot_estimate_bytes) and a margin — cf_index_open's own reserve is a fifth of the device, too little for
    # three slots of 10 M mates (3 x 22.6 GB) and far too much for three of 4 M reads
    budget = int(a.hbm_budget_gb * 1e9)
    if not budget:
        torch.cuda.empty_cache()
        free_b = torch.cuda.mem_get_info(local)[0]
        budget = max(0, int(free_b - S * capi.slot_bytes(n_reads, n_reads * W) - (8 << 30)))
        log("HBM free %.1f GB, %d slots of %.1f GB: the index is offered %.1f GB" % (free_b / 1e9, S, capi.slot_bytes(n_reads, n_reads * W) / 1e9, budget / 1e9))
    ix = capi.Index(base, device=local, hbm_budget=budget, **{k_: v_ for k_, v_ in (P.get("index_opts") or {}).items() if v_})
    index_open_s = time.time() - t0
    clf = capi.Classifier(ix)
    ix_cfg = ix.describe()
    compressed = bool(ix.L.cf_index_compressed(ix.h))
    resolve_rate, resolve_ms = ix.L.cf_index_resolve_rate(ix.h), ix.L.cf_index_resolve_build_ms(ix.h)
    tv_rate, tv_ms = ix.L.cf_index_text_verify_rate(ix.h), ix.L.cf_index_text_verify_build_ms(ix.h)
    log("index in HBM: %.2f GB (resolve table of every %d-th row made in %.0f ms; text + SA / inverse SA samples %s), text %.2f Gbp, compressed=%s, load %.1fs" %
        (ix.device_bytes / 1e9, 1 << resolve_rate, resolve_ms, "of every %d-th row made in %.0f ms" % (1 << tv_rate, tv_ms) if tv_rate >= 0 else "not built",
         ix.text_len / 1e9, compressed, time.time() - t0))

    # ---- the sampled queries of set 0 get the seeds the reference derives (names + bases): parity on the benchmark's own reads
    nq_all = n_reads // per
    ns = (len(sample_codes) // per) if sample_codes is not None else 0
    names = read_names(ns)
    if ns:                                      # the mates of a pair share the name ("/1", "/2" are not hashed)
        sets[0][3].a[:ns * per] = seeds_for(sample_codes, np.repeat(names, per, axis=0))

    slots = [capi.Slot(clf, n_reads, n_reads * W) for _ in range(S)]
    if narrow:
        for s_ in slots:
            s_.set_result_format(capi.RESULTS_NARROW)
    st_up, st_k, st_dn = torch.cuda.Stream(), torch.cuda.Stream(), torch.cuda.Stream()
    streams = (st_up.cuda_stream, st_k.cuda_stream, st_dn.cuda_stream)
    counts_ptr = clf.counts_device_ptr()
    n_taxa = ix.num_taxa

    class _Raw:                   # expose the library's device counters to torch (RCCL all-reduce in place)
        __cuda_array_interface__ = {"shape": (2 * n_taxa,), "typestr": "<i8", "data": (counts_ptr, False), "version": 2}
    counts_t = torch.as_tensor(_Raw(), device=torch.device("cuda", local)) if dist is not None else None

    inflight = [False] * S
    last = [None] * S
    acc = {"kms": np.zeros(5), "plan": 0.0, "n": 0}

    def collect(j):
        """results of slot j's batch + its stage times.  All kernels of all slots run on ONE stream, one after the
        other, so the HIP-event intervals of a batch are its kernels' own durations even while copies overlap them"""
        last[j] = slots[j].wait_narrow(copy=False) if narrow else slots[j].wait(copy=False, offsets=False)        # (row offsets are a host-side pass the pipeline does not need)
        ms, pm = slots[j].timings()
        acc["kms"] += np.array(ms)
        acc["plan"] += pm
        acc["n"] += 1
        inflight[j] = False

    def submit_set(j):
        pb, pm, pl_, ps, nw, pd = sets[j]
        if narrow:
            slots[j].submit_dense(pd.a, ps.a, read_len, paired=paired, nwords=(nw[0].a, nw[1].a), streams=streams)
        else:
            slots[j].submit(pb.a, pm.a if pm is not None else None, pl_.a, ps.a, paired=paired, max_len=read_len, streams=streams,
                            n_bases=n_reads * read_len, nwords=(nw[0].a, nw[1].a) if nw is not None else None)

    def pipeline(n, resident=False):
        """n steps: step i submits read set i % S through slot i % S after collecting what that slot held.  resident: the slot's
        reads are in HBM already (its last submit put them there) — plan + kernels + download only, nothing crosses PCIe inbound"""
        for i in range(n):
            j = i % S
            if inflight[j]:
                collect(j)
            if resident:
                slots[j].resubmit((streams[1], streams[2]))
            else:
                submit_set(j)
            inflight[j] = True
        for j in [(n + k) % S for k in range(S)]:          # drain, oldest first
            if inflight[j]:
                collect(j)

    # ---- warm-up: every slot gets its read set into HBM (and its buffers sized) through the whole host-to-host pipeline ...
    pipeline(max(a.warmup, S))
    torch.cuda.synchronize()
    # ---- ... then the PCIe-inclusive rate (never `value`: reported as host_to_host): packed reads in pinned host memory -> H2D ->
    # plan + kernels -> D2H -> rows in pinned host memory, S batches in flight, the same K steps
    if dist is not None:
        dist.barrier()
    t0 = time.perf_counter()
    pipeline(a.steps)
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()                         # (every rank has finished its K steps: the slowest rank's time, as for `value`)
    h2h_dt = time.perf_counter() - t0
    # ---- the timed region of `value`: the inputs are resident in HBM when it starts (every slot holds its read set since the
    # steps above); a step = plan + every kernel of a batch + its rows downloaded into pinned host memory
    pipeline(min(a.warmup, S), resident=True)
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    acc["kms"][:] = 0
    acc["plan"], acc["n"] = 0.0, 0
    t0 = time.perf_counter()
    pipeline(a.steps, resident=True)
    if dist is not None:
        with torch.cuda.stream(st_k):
            cfd.allreduce_counts(dist, counts_t)       # the one collective of the path (RCCL over xGMI)
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    per_rank_ms = [dt / max(1, a.steps) * 1e3]
    if dist is not None:                       # MAX over ranks is the step time; every rank's own time shows a straggler
        times = cfd.per_rank(dist, dt, rank, world)
        per_rank_ms = [x / max(1, a.steps) * 1e3 for x in times]
        dt = max(times)
        h2h_dt = max(cfd.per_rank(dist, h2h_dt, rank, world))
        index_open_all = cfd.per_rank(dist, index_open_s, rank, world)
    else:
        index_open_all = [index_open_s]

    if old_affinity is not None:                   # (every core again for what follows: the CPU reference, the command line's run)
        os.sched_setaffinity(0, old_affinity)
    # per-kernel HIP-event times: averages over the timed steps
    kms = acc["kms"] / max(1, acc["n"])
    plan_step_ms = acc["plan"] / max(1, acc["n"])
    # ---- untimed: one slot alone through the blocking calls (cf_batch_plan + cf_classify) on its resident reads
    if last[0] is None:                                   # --steps 0 --warmup 0: still give slot 0 a batch
        submit_set(0)
        last[0] = slots[0].wait_narrow(copy=False) if narrow else slots[0].wait(copy=False, offsets=False)
    reps, iso = 3, np.zeros(6)
    t1 = time.perf_counter()
    for _ in range(reps):
        iso[5] += slots[0].plan(st_k.cuda_stream)
        slots[0].classify(st_k.cuda_stream)
        iso[:5] += np.array(slots[0].timings()[0])
    resident_wall = (time.perf_counter() - t1) / reps
    iso /= reps
    if acc["n"] == 0:
        kms, plan_step_ms = iso[:5], iso[5]
    ops = slots[0].opcounts()
    if narrow:                                             # rows of read set 0 (the parity sample lives there), in the wide form
        rows_, n_rows_, s2_, ms_, info_ = slots[0].wait_narrow(expand=(None, read_len, paired))
        first_ = np.zeros(len(n_rows_) + 1, dtype=np.uint64)
        np.cumsum(n_rows_, out=first_[1:])
        res0 = (rows_, first_, n_rows_, s2_, ms_, info_)
    else:
        res0 = slots[0].wait(copy=False)

    # ---- N > 1: the per-rank report images meet on rank 0 (observed tuples for the EM); outside the timed region
    merged_rows = None
    if dist is not None:
        rep = capi.Report(ix)
        rows, first, n_rows, score2, max_score, info = res0
        rep.add(rows, n_rows, max_score, 0)
        if cfd.merge_reports(dist, rep, rank, world):
            rp = os.path.join(workdir, "merged_report.tsv")
            rep.write(rp)
            merged_rows = max(0, len(open(rp).read().splitlines()) - 1)
        rep.close()

    if rank == 0:
        total_reads = n_reads * world * a.steps
        value = total_reads / dt
        # dominant kernel = the search kernel.  Its algorithmic bytes per launch, every request at the granule it is made at:
        # an LF step = one 16-byte plane entry (a 128-byte side without the planes), a wide-ftab entry or an
        # SA / inverse-SA sample 8, a 10-mer ftab pair 16, a text window 32, a strand record in, a 16-byte hit record out
        planes = bool(ix.L.cf_index_occ_planes(ix.h))
        step_b = 16 if planes else 128
        rec_b = 64 if read_len <= 128 else 96 if read_len <= 192 else 128
        calls = ops.n_ftab + ops.n_ftab_wide
        search_bytes = step_b * (ops.n_pair + ops.n_pair2 + ops.n_single) + 16 * ops.n_ftab + 8 * (ops.n_ftab_wide + 2 * ops.n_verify - ops.n_pos_hits) + \
            32 * ops.n_text_loads + 2 * rec_b * n_reads + 16 * calls + 16 * n_reads
        # every load request of the search launch (the limit is ~50 G random requests/s whatever the granule)
        search_requests = ops.n_pair + ops.n_pair2 + ops.n_single + ops.n_ftab + ops.n_ftab_wide + 2 * ops.n_verify - ops.n_pos_hits + ops.n_text_loads + 2 * n_reads        # + one strand record per (read, strand)
        achieved = search_bytes / (kms[0] * 1e-3) / 1e9
        # the whole path = the search's bytes (above) + what the stages behind it must touch: the hit records read back (16 each), a
        # resolve-table / SA-sample entry and a 16-byte reference record per resolved row, the LF steps of the walk, the printed
        # rows (24 bytes) and three words per query out — a superset of the search's figure by construction
        whole_bytes = search_bytes + 16 * calls + (ix.sa_width + 16) * ops.n_rows + (64 if planes else 128) * ops.n_walk + \
            24 * len(res0[0]) + 12 * nq_all
        rand_gbps = ix.random_read_gbps(1 << 26, 64)
        if narrow:
            pcie_in = n_reads * ((read_len + 3) // 4 + 4) + 12 * len(sets[0][4][0].a)
            pcie_out = len(res0[0]) * 16 + nq_all * 5
        else:
            pcie_in = n_reads * (W * 8 + 8) + (n_reads * W * 4 if a.dense_nmask else 12 * len(sets[0][4][0].a))
            pcie_out = len(res0[0]) * 24 + nq_all * 12
        rows_out = int(res0[5]["planned_sa_rows"])
        res = {
            "metric": "classified reads/sec (whole node) on 100bp synthetic reads vs p_compressed; HBM GB/s achieved",
            "value": value, "unit": "reads/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": dt / max(1, a.steps) * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u64", "data": "synthetic",
            "config": {"workload": "%s: synthetic index %d genomes x %d bp = %.2f Gbp (%.2f GB resident in HBM%s), %d x %d bp %s reads per GPU "
                                   "per step, -k 5, %s index (ihits %d); timed with the packed reads resident in HBM: plan + kernels -> rows in pinned host "
                                   "memory, %d batches in flight (host_to_host: the same with the reads coming from pinned host memory)" %
                                   (P["what"] + scaled_note, n_genomes, genome_len, ix.text_len / 1e9, ix.device_bytes / 1e9,
                                    "; p_compressed itself is ~4.2 GB and not downloadable here" if a.config.startswith("2") else "",
                                    n_reads, read_len, "PE (FR pairs, mates counted)" if paired else "SE",
                                    "compressed" if compressed else "uncompressed", 20 if compressed else 200, S),
                       "preset": a.config, "recipe": P["recipe"], "index_bytes": ix.device_bytes, "reads_per_gpu_per_step": n_reads, "read_len": read_len,
                       "index_build_s_gpu": build_s, "index_open_s_per_rank": index_open_all, "inflight": S, "resolve_table_every_nth_row": 1 << resolve_rate, "resolve_table_build_ms": resolve_ms,
                       "text_verify_sample_every_nth": (1 << tv_rate) if tv_rate >= 0 else None, "text_verify_build_ms": tv_ms, "wide_ftab_chars": ix.L.cf_index_wide_ftab_chars(ix.h), "occ_planes": planes, "occ_planes_build_ms": ix.L.cf_index_occ_planes_build_ms(ix.h), "pair_planes": bool(ix_cfg["pair_planes"]),
                       "hbm_budget_gb": a.hbm_budget_gb or None, "hbm_offered_gb": budget / 1e9, "index_options": P.get("index_opts") or None, "small_range_rows_in_effect": int(ix_cfg.get("small_range_rows", 0)), "repeat_fraction": ix_cfg.get("repeat_fraction"), "plan_realised": ix_cfg.get("plan_realised"),
                       "host_link": host_link(torch, local),
                       "host_to_host_reads_per_s": n_reads * world * a.steps / h2h_dt, "host_to_host_ms_per_step": h2h_dt / max(1, a.steps) * 1e3,
                       "read_sets": "%d distinct read sets (one per slot, seeded apart): step i classifies set i %% %d again — throughput is that of fresh reads (no result is cached), but the sets' own lines may still sit in L2 / MALL from three steps before" % (S, S),
                       "index_tables": {k_: ix_cfg[k_] for k_ in ("file_section_bytes", "wide_ftab_bytes", "text_bytes", "planes_bytes", "pair_planes_bytes", "resolve_bytes", "total_bytes", "est_requests_per_100bp_read")},
                       "parallelism": "index replicated per GPU, reads sharded, RCCL all-reduce of per-taxon counters"},
            "per_rank_ms_per_step": per_rank_ms,
            "timing_scope": "inputs resident in HBM when the timed region starts (packed reads uploaded by the warm-up steps): plan/search/post/walk/score/compact -> D2H -> rows in pinned host memory; K steps over S slots in flight",
            "host_to_host": {"reads_per_s": n_reads * world * a.steps / h2h_dt, "ms_per_step": h2h_dt / max(1, a.steps) * 1e3,
                             "note": "the PCIe-inclusive rate (SURVEY 8d's scope, rounds 1-3's `value`): pinned packed reads -> H2D -> the same kernels -> D2H -> pinned rows, "
                                     "same K steps, this rank; bound by the host link where the kernels outrun it (DESIGN.md 5)"},
            "device_resident": {"reads_per_s": n_reads / ((plan_step_ms + kms[4]) * 1e-3), "ms_per_step": plan_step_ms + kms[4],
                                "blocking_api_wall_ms_per_step": resident_wall * 1e3,
                                "blocking_api_kernels_ms": {"plan": iso[5], "search": iso[0], "post": iso[1], "walk": iso[2], "score": iso[3], "total": iso[5] + iso[4]},
                                "note": "plan + kernels of a batch whose packed reads are in HBM: HIP events of the timed steps (all kernels share one "
                                        "stream, so the intervals are the kernels' own durations); the blocking_api figures are one slot alone through "
                                        "cf_batch_plan + cf_classify, each call starting on an idle GPU"},
            "pcie_bytes_per_step": {"in": pcie_in, "out": pcie_out, "per_read": (pcie_in + pcie_out) / n_reads, "wire": a.wire if narrow or a.wire == "wide" else "wide",
                                    "note": "narrow: cf_dense_reads in (four bases per byte, one length for the batch, seeds, sparse N mask), CF_RESULTS_NARROW out (16-byte rows, one "
                                            "byte + 2ndBestScore per query); wide: cf_packed_reads in (32-base words, lengths, seeds), cf_row (24 bytes) + three words per query out — the same "
                                            "classification to the bit (tests/test_async_abi.py)"},
            "roofline": {"bound": "hbm", "kernel": "k_search2", "achieved": achieved, "peak": HBM_PEAK_GBPS,
                         "unit": "GB/s", "frac": achieved / HBM_PEAK_GBPS, "traffic": None,
                         "kernel_ms": kms[0], "algorithmic_bytes_per_launch": search_bytes,
                         "algorithmic_bytes_per_read_whole_path": whole_bytes / n_reads,
                         "whole_path_GBps": whole_bytes / ((plan_step_ms + kms[4]) * 1e-3) / 1e9,
                         "whole_path_frac": whole_bytes / ((plan_step_ms + kms[4]) * 1e-3) / 1e9 / HBM_PEAK_GBPS,
                         "load_requests_per_launch": search_requests, "achieved_Grequests_per_s": search_requests / (kms[0] * 1e-3) / 1e9,
                         "measured_random_Grequests_per_s": rand_gbps / 128.0,
                         "frac_of_measured_request_rate": search_requests / (kms[0] * 1e-3) / 1e9 / (rand_gbps / 128.0) if rand_gbps else None,
                         "measured_random_128B_read_GBps": rand_gbps,
                         "frac_of_measured_random": achieved / rand_gbps if rand_gbps else None,
                         "binding_resource": "random requests per second, not bytes: a CU's L1 takes ~0.16 (load instruction x line) "
                                             "per cycle and HBM ~50 G random lines/s whatever the granule (DESIGN.md 3, tools/microbench/)",
                         "step_bytes": step_b},
            "reference_byte_model": REF_MODEL.get(a.config) and dict(REF_MODEL[a.config], **{
                "equivalent_search_GBps": REF_MODEL[a.config]["search_bytes_per_read"] * n_reads / (kms[0] * 1e-3) / 1e9,
                "equivalent_whole_path_GBps": REF_MODEL[a.config]["bytes_per_read"] * n_reads / ((plan_step_ms + kms[4]) * 1e-3) / 1e9,
                "note": "SURVEY.md 8(d) formula (128 B per LF step, 16 B per ftab lookup) with the op counts of the reference's own "
                        "step-by-step algorithm on this workload (round-1 instrumented pass, profiles/r01k_bench_full_8.6Gbp.json): what "
                        "the same reads cost in the reference's data layout.  Not a physical rate: the derived tables remove most of "
                        "those steps, so it may exceed the HBM peak"}),
            "kernels_ms": {"plan": plan_step_ms, "search": kms[0], "post": kms[1], "walk": kms[2], "score": kms[3],
                           "total": plan_step_ms + kms[4]},
            "ops_per_read": {"ftab": ops.n_ftab / n_reads, "pair": ops.n_pair / n_reads, "pair2": ops.n_pair2 / n_reads,
                             "single": ops.n_single / n_reads, "ftab_wide": ops.n_ftab_wide / n_reads, "verify": ops.n_verify / n_reads, "pos_hits": ops.n_pos_hits / n_reads, "text_loads": ops.n_text_loads / n_reads,
                             "walk": ops.n_walk / n_reads, "rows": rows_out / n_reads,
                             "printed_rows": len(res0[0]) / n_reads},
            "general_kernel_queries": {"post": int(res0[5]["slow_post"]) / max(1, nq_all), "score": int(res0[5]["slow_score"]) / max(1, nq_all),
                                       "note": "share of the queries the common-case post / score kernels (registers only) left to the general ones"},
        }
        if merged_rows is not None:
            res["merged_report_rows"] = merged_rows
        # HBM traffic of the dominant kernel: rocprofv3 PMC passes cannot run inside this process, so the figure comes from the
        # committed passes of this very workload (tools/gpu_profile.sh) — and only while the kernel sources are still the ones
        # it was collected on (sha256 of csrc/cf_kernels.hpp + cf_device.hip recorded beside it); otherwise null
        try:
            pmj = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))
            pm = (pmj.get("presets") or {}).get(a.config)
            if pm is not None:
                same = (pm["genomes"], pm["genome_len"], pm["reads"], pm["read_len"]) == (n_genomes, genome_len, n_reads, read_len)
                if same and pmj.get("kernel_source_sha256") == kernel_source_sha():
                    res["roofline"]["traffic"] = pm["traffic_bytes_per_launch"]
                    res["roofline"]["traffic_GBps"] = pm["traffic_bytes_per_launch"] / (kms[0] * 1e-3) / 1e9      # HBM GB/s the kernel moves (counters / this run's duration)
                    res["roofline"]["traffic_frac_of_peak"] = res["roofline"]["traffic_GBps"] / HBM_PEAK_GBPS
                    res["roofline"]["traffic_over_algorithmic"] = pm["traffic_bytes_per_launch"] / search_bytes
                    res["roofline"]["traffic_source"] = "profiles/pmc_traffic.json (rocprofv3 --pmc, %s, kernel %s, same kernel sources)" % (pmj.get("formula", ""), pm["kernel"])
                    sq = pm.get("sq_counters_per_launch")
                    if sq:
                        # the kernel's OTHER ceiling: a wave64 VALU instruction holds its SIMD for four cycles — 1,024 SIMDs at
                        # <= 2.4 GHz issue <= 6.1e11 of them per second
                        valu_peak = 1024 * 2.4e9 / 4
                        res["roofline"]["instruction_issue"] = {
                            "valu_wave_instructions_per_launch": sq["SQ_INSTS_VALU"], "valu_issue_ms_at_peak": sq["SQ_INSTS_VALU"] / valu_peak * 1e3,
                            "frac_of_kernel_ms": sq["SQ_INSTS_VALU"] / valu_peak * 1e3 / kms[0],
                            "wave_time_shares": {"issuing": sq["SQ_ACTIVE_INST_ANY"] / sq["SQ_WAVE_CYCLES"], "waiting_to_issue": sq["SQ_WAIT_INST_ANY"] / sq["SQ_WAVE_CYCLES"],
                                                 "waiting_for_memory": sq["SQ_WAIT_ANY"] / sq["SQ_WAVE_CYCLES"]},
                            "note": "rocprofv3 SQ counters of the same kernel sources (profiles/pmc_traffic.json): VALU issue time at four cycles per wave-instruction "
                                    "against the kernel's duration, and where the resident waves spend their cycles.  Round 5 read these as 'bound by instruction "
                                    "issue'; round 6 took 45 % of the always-executed instructions out for 7 % of the time: the kernel follows its chains' dependent "
                                    "round trips — requests per read — not its instruction count (DESIGN.md 3)"}
                elif same:
                    res["roofline"]["traffic_source"] = "null: profiles/pmc_traffic.json was collected on other kernel sources (stale)"
        except Exception:
            pass
        if not a.no_cpu and ns:
            try:
                procs = max(1, nproc // a.cpu_threads)       # usable cores (cgroup quota) / threads per process
                qps, tsv, n0, det = cpu_baseline(base, workdir, sample_codes, names, procs, a.cpu_threads, 5, paired)
                rps = qps * per
                res["cpu_baseline"] = {"value": rps, "unit": "reads/s", "cores": procs * a.cpu_threads, "kind": "reference",
                                       "sample": "first %d reads of rank 0's first read set%s; %d processes x %d threads of the reference's "
                                                 "centrifuge-class (--reorder) on disjoint shards, search time = wall %.1fs minus "
                                                 "index load %.1fs measured the same way; FASTA parse included" %
                                                 (ns * per, " (pairs, -1/-2)" if paired else "", procs, a.cpu_threads, det["wall_s"], det["index_load_s"]), **det}
                # parity on the benchmark sample itself: the GPU rows of the WHOLE sample (every CPU shard) vs the reference's TSV
                rows, first, n_rows, score2, max_score, info = res0
                k5 = capi.unpack_rows(rows[:int(first[n0])], first[:n0 + 1], n_rows[:n0], 5)
                nm = [bytes(x) for x in names[:n0]]
                got = rd.format_tsv(ix.seqid, nm, [read_len * per] * n0, k5, n_rows[:n0], score2[:n0])
                res["cpu_baseline"]["gpu_rows_identical_on_sample"] = (got == tsv)
                res["cpu_baseline"]["parity_checked_reads"] = n0 * per
            except Exception as e:          # the baseline is reported, never required for the metric
                res["cpu_baseline"] = {"value": None, "unit": "reads/s", "cores": nproc, "kind": "reference",
                                       "sample": "failed: %r" % (e,)}
    for s_ in slots:
        s_.close()
    if rank == 0:
        # (only beside the headline itself: not when the sizes were overridden, nor in the one-rank process-group test mode)
        headline = not (a.genomes or a.genome_len or a.reads or a.read_len or a.hbm_budget_gb or os.environ.get("CF_BENCH_FORCE_DIST"))
        if world == 1 and a.config == "2" and a.other_configs.strip() and headline:
            # the other workloads of BASELINE.json, each in a process of its own (its index needs the HBM this one holds)
            del slots, sets, last, res0
            clf.close(); ix.close()
            torch.cuda.empty_cache()
            if a.cli_reads > 0:
                try:
                    res["cli_end_to_end"] = cli_end_to_end(torch, base, workdir, P, n_genomes, genome_len, a.cli_reads, nproc, local,
                                                           bool(res.get("cpu_baseline", {}).get("value")), paired)
                except Exception as e:
                    res["cli_end_to_end"] = {"failed": repr(e)}
            res["other_configs"] = run_other_configs([c for c in a.other_configs.split(",") if c.strip()], a.other_steps, a.other_budget_s)
        print(json.dumps(res))
    if dist is not None:
        dist.barrier()
    if rank == 0 and workdir.startswith("/dev/shm/") and not os.environ.get("CF_BENCH_KEEP"):
        shutil.rmtree(workdir, ignore_errors=True)                  # a RAM disk is not a place to leave 47 GB
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
