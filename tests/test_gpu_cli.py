"""The drop-in front end (centrifuge_amd/bin/centrifuge-class) against the reference:
same command line, byte-identical TSV and report file — on the golden cases and,
for the input options, against the reference binary (oracle/_ref) run on the spot."""
import os
import subprocess
import tempfile

import pytest

import common
from oracle import oracle as O

pytestmark = pytest.mark.gpu
CLI = os.path.join(common.ROOT, "centrifuge_amd", "bin", "centrifuge-class")
# CF_TEST_SMALL_RANGE_ROWS (tests/conftest.py): the suite's switch that opens every index with small ranges finished against the text
CLI_X = [CLI] + (["--small-range-rows", os.environ["CF_TEST_SMALL_RANGE_ROWS"]] if os.environ.get("CF_TEST_SMALL_RANGE_ROWS") else [])


def run(exe, args, d):
    out, rep = os.path.join(d, "o.tsv"), os.path.join(d, "r.tsv")
    r = subprocess.run((CLI_X if exe == CLI else [exe]) + args + ["-S", out, "--report-file", rep], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    return open(out).read(), open(rep).read(), r.stderr


def read_args(d, c):
    files = [os.path.join(d, f) for f in c["reads"]]
    return ["-U", files[0]] if len(files) == 1 else ["-1", files[0], "-2", files[1]]


@pytest.mark.parametrize("arch,name", common.all_cases())
def test_cli_matches_golden(arch, name):
    d, cases = common.golden(arch)
    c = [x for x in cases if x["name"] == name][0]
    with tempfile.TemporaryDirectory() as t:
        tsv, rep, err = run(CLI, list(c["args"]) + ["-x", os.path.join(d, "idx")] + read_args(d, c), t)
    assert tsv == open(os.path.join(d, c["tsv"])).read()
    assert rep == open(os.path.join(d, c["report"])).read()
    assert "report file" in err and "Number of iterations in EM algorithm" in err


@pytest.mark.skipif(not O.have_ref(), reason="oracle/_ref not built")
@pytest.mark.parametrize("extra", [
    ["-s", "7", "-u", "90"],
    ["-5", "4", "-3", "6"],
    ["--tab-fmt-cols", "readID,taxID,taxRank,taxName,numMatches,readSeq,readQual"],
    ["--out-fmt", "sam"],
    ["--tab-fmt-cols", "QNAME,CIGAR,FLAG,RNAME,RNEXT,TLEN,SEQ1,QUAL2,readSeq2,taxLevel"],
    ["--seed", "1234", "-k", "3"],
    ["--no-abundance", "--min-hitlen", "30"],
    ["-p", "3", "--reorder"],
    ["-k", "64"],                                            # beyond the narrow result format's six bits: the wide rows cross the link
    ["--small-range-rows", "4", "--hbm-budget-gb", "2"],     # the index options of the command line (ours only: stripped for the reference)
    ["--small-range-rows", "-1"],
])
def test_cli_options_match_reference_binary(extra):
    d, cases = common.golden("synth_small")
    ref_exe = os.path.join(O.REF_DIR, "centrifuge-class")
    for fmt, reads in (("-f", ["-U", os.path.join(d, "reads.fa")]), ("-q", ["-U", os.path.join(d, "reads.fq")]),
                       ("-f", ["-1", os.path.join(d, "r1.fa"), "-2", os.path.join(d, "r2.fa")])):
        args = [fmt, "-x", os.path.join(d, "idx")] + reads + extra
        ours_only = ("--small-range-rows", "--hbm-budget-gb")
        ref_args = [fmt, "-x", os.path.join(d, "idx")] + reads + ([] if extra[0] in ours_only else extra)
        with tempfile.TemporaryDirectory() as t1, tempfile.TemporaryDirectory() as t2:
            want = run(ref_exe, ref_args, t1)
            got = run(CLI, args + ["--batch", "97"], t2)          # small batches: many trips through the C ABI
        assert got[0] == want[0], common.first_diff(got[0], want[0])
        assert got[1] == want[1], common.first_diff(got[1], want[1])


def test_cli_errors_like_the_reference():
    d, _ = common.golden("example")
    r = subprocess.run([CLI, "-f", "-x", os.path.join(d, "nonexistent"), "-U", os.path.join(d, "reads.fa")], capture_output=True, text=True)
    assert r.returncode != 0 and "Could not locate a Centrifuge index" in r.stderr
    r = subprocess.run([CLI, "-x", os.path.join(d, "idx")], capture_output=True, text=True)
    assert r.returncode != 0 and "Must specify at least one read input" in r.stderr


def test_bench_distributed_path_with_one_rank():
    """bench.py under torchrun with the process group forced on: RCCL init, barrier, the in-place
    all-reduce on the tensor aliasing cf_counts_device, the max-over-ranks timing — one rank is
    all a 1-GPU box offers; the driver runs N = 2, 4, 8."""
    import json
    import sys
    env = dict(os.environ, CF_BENCH_FORCE_DIST="1", CF_BENCH_DIR=tempfile.mkdtemp())
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", str(29600 + os.getpid() % 300), os.path.join(common.ROOT, "bench.py"), "--gpus", "1", "--steps", "2",
           "--warmup", "1", "--genomes", "32", "--genome-len", "200000", "--reads", "200000", "--cpu-sample", "20000"]
    r = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1]
    d = json.loads(line)
    assert d["n_gpus"] == 1 and d["value"] > 0 and d["roofline"]["achieved"] > 0
    assert d["cpu_baseline"]["gpu_rows_identical_on_sample"] is True


def test_entry_point_is_serially_reentrant_in_one_process():
    """centrifuge(argc, argv) (centrifuge.cpp:3338-3345) called repeatedly in this process with
    different options: every call gives the golden TSV + report of its own options."""
    import ctypes as C
    from centrifuge_amd import capi
    L = C.CDLL(capi.LIB_PATH)
    L.centrifuge.restype, L.centrifuge.argtypes = C.c_int, [C.c_int, C.POINTER(C.c_char_p)]
    d, cases = common.golden("synth_small")
    order = [c for c in cases if c["name"] in ("k5", "k1", "genus", "pe_k5", "host", "k5")] * 2
    with tempfile.TemporaryDirectory() as t:
        for i, c in enumerate(order):
            out, rep = os.path.join(t, "o%d.tsv" % i), os.path.join(t, "r%d.tsv" % i)
            args = ["centrifuge-class"] + list(c["args"]) + ["-x", os.path.join(d, "idx")] + read_args(d, c) + ["-S", out, "--report-file", rep]
            av = (C.c_char_p * len(args))(*[a.encode() for a in args])
            assert L.centrifuge(len(args), av) == 0
            assert open(out).read() == open(os.path.join(d, c["tsv"])).read(), c["name"]
            assert open(rep).read() == open(os.path.join(d, c["report"])).read(), c["name"]
        # a failing call in between leaves the next one intact
        bad = ["centrifuge-class", "-f", "-x", os.path.join(d, "idx"), "-U", os.path.join(t, "missing.fa"), "-S", os.path.join(t, "x.tsv")]
        av = (C.c_char_p * len(bad))(*[a.encode() for a in bad])
        assert L.centrifuge(len(bad), av) != 0
        c = order[0]
        out, rep = os.path.join(t, "again.tsv"), os.path.join(t, "again.rep")
        args = ["centrifuge-class"] + list(c["args"]) + ["-x", os.path.join(d, "idx")] + read_args(d, c) + ["-S", out, "--report-file", rep]
        av = (C.c_char_p * len(args))(*[a.encode() for a in args])
        assert L.centrifuge(len(args), av) == 0
        assert open(out).read() == open(os.path.join(d, c["tsv"])).read()


def test_arg_file_mode_runs_each_line():
    """centrifuge-class -A <file> (centrifuge_main.cpp:35-63): one argument string per line."""
    d, cases = common.golden("example")
    with tempfile.TemporaryDirectory() as t:
        lines = []
        for c in cases[:3]:
            lines.append(" ".join(list(c["args"]) + ["-x", os.path.join(d, "idx")] + read_args(d, c) +
                                  ["-S", os.path.join(t, c["name"] + ".tsv"), "--report-file", os.path.join(t, c["name"] + ".rep")]))
        argf = os.path.join(t, "args.txt")
        open(argf, "w").write("\n".join(lines) + "\n\n")
        r = subprocess.run([CLI, "-A", argf], capture_output=True, text=True)
        assert r.returncode == 0, r.stderr
        for c in cases[:3]:
            assert open(os.path.join(t, c["name"] + ".tsv")).read() == open(os.path.join(d, c["tsv"])).read()


@pytest.mark.skipif(not O.have_ref(), reason="oracle/_ref not built")
def test_inputs_are_processed_one_at_a_time_like_the_reference():
    """several -U files, mates and singles in one run, -s/-u per input, and --separator (marker line,
    one report per input: centrifuge.cpp:3007-3040,3128-3226) against the reference binary"""
    d, _ = common.golden("synth_small")
    ref_exe = os.path.join(O.REF_DIR, "centrifuge-class")
    u1, u2 = os.path.join(d, "reads.fa"), os.path.join(d, "reads250.fa")
    m1, m2 = os.path.join(d, "r1.fa"), os.path.join(d, "r2.fa")
    variants = [["-f", "-U", u1 + "," + u2], ["-f", "-s", "100", "-u", "300", "-U", u1 + "," + u2],
                ["-f", "-1", m1, "-2", m2, "-U", u1], ["-f", "-k", "2", "-1", m1, "-2", m2, "-U", u2 + "," + u1, "-s", "5"]]
    for args in variants:
        a = args + ["-x", os.path.join(d, "idx")]
        with tempfile.TemporaryDirectory() as t1, tempfile.TemporaryDirectory() as t2:
            want = run(ref_exe, a, t1)
            got = run(CLI, a + ["--batch", "211"], t2)
        assert got[0] == want[0], common.first_diff(got[0], want[0])
        assert got[1] == want[1], common.first_diff(got[1], want[1])
    a = ["-f", "-x", os.path.join(d, "idx"), "-1", m1, "-2", m2, "-U", u1 + "," + u2, "--separator"]
    outs = []
    for exe in (ref_exe, CLI):
        with tempfile.TemporaryDirectory() as t:
            r = subprocess.run([exe] + a + ["-S", os.path.join(t, "o.tsv")], capture_output=True, text=True, cwd=t)
            assert r.returncode == 0, r.stderr
            reps = [open(os.path.join(t, "centrifuge_report_%d.tsv" % i)).read() for i in range(3)]
            assert not os.path.exists(os.path.join(t, "centrifuge_report.tsv"))
            outs.append((open(os.path.join(t, "o.tsv")).read(), reps, [ln for ln in r.stderr.splitlines() if ln.startswith("report file")]))
    assert outs[0][0].count("#File_End_Here\n") == 3
    assert outs[1][0] == outs[0][0], common.first_diff(outs[1][0], outs[0][0])
    assert outs[1][1] == outs[0][1] and outs[1][2] == outs[0][2]


@pytest.mark.parametrize("gpu_args,env", [
    (["--gpu-list", "0,0"], {}),                          # the one device opened as two logical GPUs: two index replicas, host-summed counters
    (["--gpu-list", "0,0,0", "--slots", "1"], {}),
    (["--gpus", "1", "--slots", "3"], {"CF_CLI_RCCL": "1"}),      # one device through ncclCommInitAll + the all-reduce group
    (["--gpus", "all"], {}),
])
@pytest.mark.parametrize("name", ["k5", "pe_k1", "r250_k5", "host"])
def test_cli_on_several_gpus_matches_golden(name, gpu_args, env):
    """centrifuge-class --gpus / --gpu-list (the C++ multi-GPU driver: centrifuge.cpp:2762-2819, aln_sink.h:109-140): batches
    dealt to whichever GPU thread is free, printed in input order; per-thread tallies merged; the devices' per-taxon
    counters reduced (RCCL group / host sum) and checked against the merged tally.  Small batches, so every worker gets
    many and finishes them out of order."""
    d, cases = common.golden("synth_small")
    c = [x for x in cases if x["name"] == name][0]
    with tempfile.TemporaryDirectory() as t:
        out, rep = os.path.join(t, "o.tsv"), os.path.join(t, "r.tsv")
        cmd = CLI_X + list(c["args"]) + ["-x", os.path.join(d, "idx")] + read_args(d, c) + ["-S", out, "--report-file", rep, "--batch", "61", "-p", "2", "-t"] + gpu_args
        r = subprocess.run(cmd, capture_output=True, text=True, env=dict(os.environ, **env))
        assert r.returncode == 0, r.stderr
        assert open(out).read() == open(os.path.join(d, c["tsv"])).read()
        assert open(rep).read() == open(os.path.join(d, c["report"])).read()
        if env:
            assert "all-reduced over 1 GPU(s) with RCCL" in r.stderr


def test_cli_gpu_option_errors():
    d, _ = common.golden("example")
    base = ["-f", "-x", os.path.join(d, "idx"), "-U", os.path.join(d, "reads.fa")]
    r = subprocess.run(CLI_X + base + ["--gpus", "0"], capture_output=True, text=True)
    assert r.returncode == 1 and "--gpus arg must be" in r.stderr
    r = subprocess.run(CLI_X + base + ["--gpu-list", "0,0", "--separator"], capture_output=True, text=True)
    assert r.returncode == 1 and "--separator works on one GPU" in r.stderr
    r = subprocess.run(CLI_X + base + ["--gpu-list", "0,97"], capture_output=True, text=True)
    assert r.returncode == 1


def test_multi_gpu_driver_error_paths_end_the_run_cleanly():
    """what the C++ multi-GPU driver does when things go wrong, on the one test GPU opened as several logical
    devices (the failures themselves are provoked through the knob gate, CF_TEST_FAIL_OPEN / CF_TEST_FAIL_COMM: one device cannot
    run out of memory for one of its replicas only, nor RCCL fail to initialise).  One replica of three that does not open: the
    run ends with exit code 1 and a message naming the device and the replica, within seconds (the replicas that did open are
    closed, no thread is left waiting), no output file is half written.  Communicators that cannot be made: the classification
    output is complete (it is written before the reduction), the run ends with exit code 1 and RCCL's message, no report."""
    import time
    d, cases = common.golden("synth_small")
    c = [x for x in cases if x["name"] == "k5"][0]
    with tempfile.TemporaryDirectory() as t:
        out, rep = os.path.join(t, "o.tsv"), os.path.join(t, "r.tsv")
        base = CLI_X + list(c["args"]) + ["-x", os.path.join(d, "idx")] + read_args(d, c) + ["-S", out, "--report-file", rep, "--batch", "61", "-p", "2", "-t"]
        t0 = time.time()
        r = subprocess.run(base + ["--gpu-list", "0,0,0"], capture_output=True, text=True, timeout=120, env=dict(os.environ, CF_DEBUG_KNOBS="1", CF_TEST_FAIL_OPEN="1"))
        assert r.returncode == 1 and "out of memory" in r.stderr and "device 0, replica 2 of 3" in r.stderr, r.stderr
        assert time.time() - t0 < 60 and not os.path.exists(rep)
        assert not os.path.exists(out) or os.path.getsize(out) == 0
        # the placement line of a sound multi-replica run names every device's node
        ok = subprocess.run(base + ["--gpu-list", "0,0"], capture_output=True, text=True, timeout=300)
        assert ok.returncode == 0 and "NUMA placement of the loader and GPU threads: device 0 node" in ok.stderr, ok.stderr
        assert open(out).read() == open(os.path.join(d, c["tsv"])).read()
        os.remove(out); os.remove(rep)
        r = subprocess.run(base, capture_output=True, text=True, timeout=300, env=dict(os.environ, CF_DEBUG_KNOBS="1", CF_CLI_RCCL="1", CF_TEST_FAIL_COMM="1"))
        assert r.returncode == 1 and "ncclCommInitAll failed" in r.stderr, r.stderr
        assert open(out).read() == open(os.path.join(d, c["tsv"])).read() and not os.path.exists(rep)


def test_numa_helpers_of_the_c_abi():
    """cf_device_numa_node / cf_thread_bind_near_device: the node sysfs names for the GPU's PCIe link, and the calling thread bound
    to (a subset of) its CPUs; an unknown topology changes nothing and is no error"""
    import ctypes as C
    import threading
    from centrifuge_amd import capi
    L = capi.lib()
    node = L.cf_device_numa_node(0)
    assert node >= -1 and L.cf_device_numa_node(4096) == -1
    res = {}

    def work():
        before = os.sched_getaffinity(0)
        got = C.c_int(-7)
        rc = L.cf_thread_bind_near_device(0, C.byref(got))
        res["rc"], res["node"], res["before"], res["after"] = rc, got.value, before, os.sched_getaffinity(0)
    th = threading.Thread(target=work); th.start(); th.join()
    assert res["rc"] == 0 and res["after"] <= res["before"] and len(res["after"]) >= 1
    assert res["node"] in (-1, node)
    if res["node"] < 0:
        assert res["after"] == res["before"]


@pytest.mark.parametrize("gpu_args", [[], ["--gpu-list", "0,0"]])
def test_counter_self_check_passes_on_a_sound_run(gpu_args):
    """every run ends with the devices' per-taxon counters (summed over the devices) compared with the tally of the rows the
    output stage saw (cf_report_adopt_counts): a sound run passes it and prints the golden report.  That a disagreement is fatal
    is tested where it can be provoked without a hook in the product: tests/test_report.py::test_adopted_counters_must_agree"""
    d, cases = common.golden("synth_small")
    c = [x for x in cases if x["name"] == "k5"][0]
    with tempfile.TemporaryDirectory() as t:
        args = CLI_X + list(c["args"]) + gpu_args + ["-x", os.path.join(d, "idx")] + read_args(d, c) + ["-S", os.path.join(t, "o.tsv"), "--report-file", os.path.join(t, "r.tsv")]
        ok = subprocess.run(args, capture_output=True, text=True)
        assert ok.returncode == 0, ok.stderr
        assert open(os.path.join(t, "r.tsv")).read() == open(os.path.join(d, c["report"])).read()


# ---- the command line under random taxonomies (missing 6 / next 2): tests/fuzz/fuzz_taxonomy.py's recipe — random
# trees over the whole rank vocabulary, sequences on leaves / inner nodes / shared nodes / taxIDs the tree does not know, 40-bit
# taxIDs, random -k / --classification-rank / --min-hitlen / host / exclude lists, pairs and ragged read lengths — through the
# BINARY on the device: its own row formatter (formatDefault / formatRange, aln_sink.h:2203-2250) and report against the
# reference binary, 200 seeds in groups of 8.
def _random_taxonomy_case(seed, d):
    import sys
    import numpy as np
    sys.path.insert(0, os.path.join(common.ROOT, "tools"))
    import synth
    from centrifuge_amd import capi
    rng = np.random.default_rng(990000 + seed)
    n_clusters, per = int(rng.integers(1, 5)), int(rng.integers(2, 7))
    L = int(rng.integers(1200, 3500))
    g = synth.make_genomes(n_clusters * per, L, genus_size=per, divergence=float(rng.choice([0.0, 0.005, 0.02, 0.05])), seed=int(rng.integers(1 << 30)))
    synth.write_reference(d, g, genus_size=per)
    seq_tid, nodes = synth.write_random_taxonomy(d, rng, n_clusters, per)
    try:
        O.ref_build(d, threads=2)
    except subprocess.CalledProcessError:
        return None                                              # the reference builder refused the input: nothing to compare
    rl = int(rng.choice([60, 100, 150]))
    if rng.random() < 0.3:
        (nm, s1), (_, s2) = synth.sample_reads(g, 120, min(rl, L // 4), paired=True, random_frac=0.05, n_frac=0.05, seed=int(rng.integers(1 << 30)))
        synth.write_fasta(os.path.join(d, "r1.fa"), nm, s1, "/1"); synth.write_fasta(os.path.join(d, "r2.fa"), nm, s2, "/2")
        files = ["-1", os.path.join(d, "r1.fa"), "-2", os.path.join(d, "r2.fa")]
    else:
        nm, s = synth.sample_reads(g, 120, min(rl, L // 2), random_frac=0.05, n_frac=0.05, seed=int(rng.integers(1 << 30)))
        if rng.random() < 0.5:                                   # ragged lengths (trimmed reads)
            s = [x[:int(rng.integers(20, len(x) + 1))] for x in s]
        synth.write_fasta(os.path.join(d, "r.fa"), nm, s)
        files = ["-U", os.path.join(d, "r.fa")]
    a = ["-f", "-k", str(int(rng.choice([1, 1, 2, 3, 5, 20]))), "--min-hitlen", str(int(rng.choice([16, 22, 22, 30]))),
         "--classification-rank", str(rng.choice(list(capi.RANK_SLOTS)))]
    if rng.random() >= 0.8:
        a.append("--no-traverse")
    pool = sorted(set(seq_tid) | set(nodes))
    if rng.random() < 0.25:
        a += ["--host-taxids", ",".join(str(int(x)) for x in rng.choice(pool, size=min(len(pool), int(rng.integers(1, 3))), replace=False))]
    if rng.random() < 0.25:
        a += ["--exclude-taxids", ",".join(str(int(x)) for x in rng.choice(pool, size=min(len(pool), int(rng.integers(1, 3))), replace=False))]
    if rng.random() < 0.3:
        a += ["--tab-fmt-cols", "readID,seqID,taxID,taxRank,taxName,score,2ndBestScore,hitLength,queryLength,numMatches"]
    return a + ["-x", os.path.join(d, "idx")] + files


@pytest.mark.skipif(not O.have_ref(), reason="oracle/_ref not built")
@pytest.mark.parametrize("group", range(25))
def test_cli_random_taxonomies(group):
    ref_exe = os.path.join(O.REF_DIR, "centrifuge-class")
    compared = 0
    for seed in range(8 * group, 8 * group + 8):
        with tempfile.TemporaryDirectory() as d, tempfile.TemporaryDirectory() as t1, tempfile.TemporaryDirectory() as t2:
            args = _random_taxonomy_case(seed, d)
            if args is None:
                continue
            r = subprocess.run([ref_exe] + args + ["-S", os.path.join(t1, "o.tsv"), "--report-file", os.path.join(t1, "r.tsv")], capture_output=True, text=True)
            if r.returncode != 0:
                continue                                         # (the reference refuses the case: nothing to compare)
            want = open(os.path.join(t1, "o.tsv")).read(), open(os.path.join(t1, "r.tsv")).read()
            got = run(CLI, args + ["--batch", "53"], t2)
            assert got[0] == want[0], (seed, args, common.first_diff(got[0], want[0]))
            assert got[1] == want[1], (seed, args, common.first_diff(got[1], want[1]))
            compared += 1
    assert compared >= 6


# ---- N > 1 on real devices.  The builder's boxes have one GPU, so everything below is skipped there; the
# first box with two or more runs the RCCL paths with as many ranks as it has: the C ABI's group all-reduce, the C++ driver
# (--gpus 2 / all) and bench.py under torchrun with two processes.
def _n_devices():
    from centrifuge_amd import capi
    try:
        return int(capi.lib().cf_device_count())
    except Exception:
        return 0


def _need_two():
    # (asked inside the tests, not at import: the library's HIP runtime must not come up before torch's in the collecting process)
    if _n_devices() < 2:
        pytest.skip("needs two or more GPUs")


def test_counts_allreduce_group_over_real_devices():
    """cf_comm_init_all (ncclCommInitAll) + cf_counts_allreduce_group on distinct devices: every device classifies its own
    slice of the golden reads, and after the group all-reduce every device holds the reference's per-taxon counters"""
    _need_two()
    import ctypes as C
    import numpy as np
    from centrifuge_amd import capi, reads
    n = min(_n_devices(), 8)
    d, cases = common.golden("synth_small")
    c = [x for x in cases if x["name"] == "k5"][0]
    base = os.path.join(d, "idx")
    names, qlens, seq, off, seeds, paired = reads.load([os.path.join(d, f) for f in c["reads"]], False)
    L = capi.lib()
    ixs = [capi.Index(base, device=i) for i in range(n)]
    clfs = [capi.Classifier(ix) for ix in ixs]
    nq = len(names)
    for i in range(n):                                   # rank i: reads i, i + n, ...
        sel = list(range(i, nq, n))
        s2 = np.concatenate([seq[int(off[q]):int(off[q + 1])] for q in sel]) if sel else np.zeros(0, np.uint8)
        o2 = np.concatenate([[0], np.cumsum([int(off[q + 1] - off[q]) for q in sel])]).astype(np.uint64)
        b = clfs[i].batch(s2, o2, np.asarray([seeds[q] for q in sel], dtype=np.uint32), False)
        b.classify(); b.results(); b.close()
    devs = (C.c_int * n)(*range(n))
    comms = (C.c_void_p * n)()
    capi._check(L.cf_comm_init_all(n, devs, comms))
    hs = (C.c_void_p * n)(*[cl.h for cl in clfs])
    capi._check(L.cf_counts_allreduce_group(hs, comms, n))
    want = None
    for cl in clfs:
        got = cl.counts()
        if want is None:
            want = got
        assert np.array_equal(got[0], want[0]) and np.array_equal(got[1], want[1])
    rep = open(os.path.join(d, c["report"])).read().splitlines()[1:]
    ref_counts = {int(f.split("\t")[1]): (int(f.split("\t")[4]), int(f.split("\t")[5])) for f in rep}
    tax = ixs[0].taxon_ids()
    mine = {int(tax[i]): (int(want[0][i]), int(want[1][i])) for i in range(len(tax)) if want[0][i] and tax[i] != 0}
    assert mine == ref_counts
    for i in range(n):
        L.cf_comm_destroy(comms[i])
    for cl in clfs:
        cl.close()
    for ix in ixs:
        ix.close()


@pytest.mark.parametrize("name", ["k5", "pe_k1"])
def test_cli_on_two_real_gpus(name):
    _need_two()
    d, cases = common.golden("synth_small")
    c = [x for x in cases if x["name"] == name][0]
    with tempfile.TemporaryDirectory() as t:
        out, rep = os.path.join(t, "o.tsv"), os.path.join(t, "r.tsv")
        cmd = CLI_X + list(c["args"]) + ["-x", os.path.join(d, "idx")] + read_args(d, c) + ["-S", out, "--report-file", rep, "--batch", "61", "-p", "2", "-t", "--gpus", "2"]
        r = subprocess.run(cmd, capture_output=True, text=True)
        assert r.returncode == 0, r.stderr
        assert open(out).read() == open(os.path.join(d, c["tsv"])).read()
        assert open(rep).read() == open(os.path.join(d, c["report"])).read()
        assert "all-reduced over 2 GPU(s) with RCCL" in r.stderr


def test_bench_under_torchrun_with_two_ranks():
    _need_two()
    import json
    import sys
    env = dict(os.environ, CF_BENCH_DIR=tempfile.mkdtemp())
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(29900 + os.getpid() % 90), os.path.join(common.ROOT, "bench.py"), "--gpus", "2", "--steps", "3",
           "--warmup", "1", "--genomes", "32", "--genome-len", "200000", "--reads", "200000", "--cpu-sample", "20000", "--other-configs", ""]
    r = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    dj = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert dj["n_gpus"] == 2 and dj["value"] > 0 and len(dj["per_rank_ms_per_step"]) == 2
    assert dj["cpu_baseline"]["gpu_rows_identical_on_sample"] is True
    assert dj["merged_report_rows"] > 0
