"""The index / report tools around the hot path: centrifuge-inspect and
centrifuge-kreport against the reference's outputs committed in tests/golden/tools.tar.xz
(made by tests/golden/make_tools_golden.py with the compiled reference inspector and the
reference's Perl kreport).  The table modes and kreport are host code (CPU tests); the
inspector's FASTA mode inverts the BWT on the GPU (gpu tests) — its kernels' bodies and the
record formatter are stepped on the CPU through tests/emu here."""
import os
import subprocess
import tempfile

import numpy as np
import pytest

import common
from emu import emu as E

BIN = os.path.join(common.ROOT, "centrifuge_amd", "bin")
INSPECT = os.path.join(BIN, "centrifuge-inspect")
KREPORT = os.path.join(BIN, "centrifuge-kreport")
PROMOTE = os.path.join(BIN, "centrifuge-promote")


def index_of(name):
    if name == "gaps":
        return os.path.join(common.golden("tools")[0], "gaps")
    return os.path.join(common.golden(name)[0], "idx")


def tool_cases(tool, fasta=None):
    _, cases = common.golden("tools")
    out = []
    for c in cases:
        if c["tool"] != tool:
            continue
        is_fasta = tool == "inspect" and "/%s.fasta" % c["index"] in c["out"]
        if fasta is None or fasta == is_fasta:
            out.append(c)
    return out


def case_id(c):
    return os.path.basename(c["out"])[:-4]


def want(c):
    return open(os.path.join(common.golden("tools")[0], c["out"]), "rb").read()


@pytest.mark.parametrize("c", tool_cases("inspect", fasta=False), ids=case_id)
def test_inspect_tables_match_reference(c):
    r = subprocess.run([INSPECT] + c["args"] + [index_of(c["index"])], capture_output=True)
    assert r.returncode == 0, r.stderr
    assert r.stdout == want(c)


@pytest.mark.parametrize("c", tool_cases("kreport"), ids=case_id)
def test_kreport_matches_reference(c):
    d = common.golden("tools")[0] if c["input"].startswith("@") else common.golden(c["index"])[0]
    files = [os.path.join(d, f) for f in c["input"].lstrip("@").split(",")]
    r = subprocess.run([KREPORT, "-x", index_of(c["index"])] + c["args"] + files, capture_output=True)
    assert r.returncode == 0, r.stderr
    assert r.stdout == want(c)


@pytest.mark.parametrize("c", tool_cases("promote"), ids=case_id)
def test_promote_matches_reference(c):
    d = common.golden(c["index"])[0]
    r = subprocess.run([PROMOTE, index_of(c["index"]), os.path.join(d, c["input"])] + c["args"], capture_output=True)
    assert r.returncode == 0, r.stderr
    assert r.stdout == want(c)


def test_kreport_stdin_and_errors():
    d = common.golden("synth_small")[0]
    c = [x for x in tool_cases("kreport") if x["out"] == "kreport/synth_small.k5.lca.txt"][0]
    r = subprocess.run([KREPORT, "-x", index_of("synth_small")], stdin=open(os.path.join(d, "k5.tsv"), "rb"), capture_output=True)
    assert r.returncode == 0 and r.stdout == want(c) and b"Reading centrifuge out file from STDIN" in r.stderr
    r = subprocess.run([KREPORT, "-x", index_of("synth_small"), "--min-score", "999999", os.path.join(d, "k5.tsv")], capture_output=True)
    assert r.returncode == 255 and b"No sequence matches with given settings" in r.stderr and r.stdout == b""
    r = subprocess.run([KREPORT, os.path.join(d, "k5.tsv")], capture_output=True)
    assert r.returncode == 64 and b"Usage: centrifuge-kreport" in r.stderr
    r = subprocess.run([INSPECT], capture_output=True)
    assert r.returncode == 1 and b"No index name given!" in r.stderr
    r = subprocess.run([INSPECT, "-n", os.path.join(d, "nonexistent")], capture_output=True)
    assert r.returncode == 1 and b"Could not locate a Centrifuge index" in r.stderr


@pytest.mark.parametrize("c", tool_cases("inspect", fasta=True), ids=case_id)
@pytest.mark.parametrize("shift", [2, 5, 10])
def test_restore_kernels_and_fasta_formatter_emulated(c, shift):
    """restore_body / restore_rank_body (one-lane chains) + printSequences on the CPU."""
    across = int(c["args"][1]) if c["args"] else 60
    e = E.Emu(index_of(c["index"]))
    with tempfile.TemporaryDirectory() as t:
        out = os.path.join(t, "o.fa")
        e.inspect_fasta(out, across, shift)
        got = open(out, "rb").read()
    e.close()
    assert got == want(c)


# ------------------------------------------------------------------ GPU
@pytest.mark.gpu
@pytest.mark.parametrize("c", tool_cases("inspect", fasta=True), ids=case_id)
def test_inspect_fasta_matches_reference(c):
    r = subprocess.run([INSPECT] + c["args"] + [index_of(c["index"])], capture_output=True)
    assert r.returncode == 0, r.stderr
    assert r.stdout == want(c)


@pytest.mark.gpu
@pytest.mark.parametrize("shift", [None, 4, 12])
def test_build_then_inspect_round_trip(shift):
    """Size-independent property: sequences -> cf_build_index -> cf_index_restore -> the same
    sequences (48 Mbp; N runs included), at several mark spacings."""
    import sys
    sys.path.insert(0, os.path.join(common.ROOT, "tools"))
    import synth
    from centrifuge_amd import capi
    rng = np.random.default_rng(3)
    G, L = 48, 1_000_000
    g = synth.ACGT[rng.integers(0, 4, (G, L), dtype=np.uint8)]
    for i in range(0, G, 5):
        for _ in range(3):
            p = int(rng.integers(0, L - 5000)); g[i, p:p + int(rng.integers(1, 4000))] = ord("N")
    g[7, :100] = ord("N"); g[9, -50:] = ord("N")
    with tempfile.TemporaryDirectory() as t:
        synth.write_reference(t, g, line=60)
        capi.build_index(os.path.join(t, "idx"), fasta=[os.path.join(t, "genomes.fa")], conversion_table=os.path.join(t, "conv.tsv"),
                         taxonomy_tree=os.path.join(t, "nodes.dmp"), name_table=os.path.join(t, "names.dmp"))
        env = dict(os.environ)
        if shift is not None:
            env["CF_RESTORE_SHIFT"] = str(shift)
        out = os.path.join(t, "out.fa")
        with open(out, "wb") as f:
            r = subprocess.run([INSPECT, os.path.join(t, "idx")], stdout=f, stderr=subprocess.PIPE, env=env)
        assert r.returncode == 0, r.stderr
        a, b = open(out, "rb").read(), open(os.path.join(t, "genomes.fa"), "rb").read()
        assert a == b, common.first_diff(a.decode(), b.decode())


# ------------------------------------------------ live against the reference's Perl scripts (build container only)
REF_SCRIPTS = "/root/reference"


@pytest.mark.skipif(not (os.path.exists(os.path.join(REF_SCRIPTS, "centrifuge-kreport")) and os.path.exists("/usr/bin/perl")),
                    reason="the reference's Perl scripts are only in the build container")
def test_kreport_and_promote_on_awkward_classification_files():
    """taxIDs that are not in the tree, unclassified rows, reads with three and more rows, reordered / extra columns:
    the same bytes as the Perl scripts (run from a scratch copy beside a shim `centrifuge-inspect`)."""
    import shutil
    import stat
    from oracle import oracle as O
    if not O.have_ref():
        pytest.skip("oracle/_ref not built")
    idx = index_of("synth_small")
    with tempfile.TemporaryDirectory() as t:
        for s in ("centrifuge-kreport", "centrifuge-promote"):
            shutil.copy(os.path.join(REF_SCRIPTS, s), t)
        shim = os.path.join(t, "centrifuge-inspect")
        open(shim, "w").write('#!/bin/sh\nexec %s/centrifuge-inspect-bin "$@"\n' % O.REF_DIR)
        os.chmod(shim, os.stat(shim).st_mode | stat.S_IEXEC)
        std = "readID\tseqID\ttaxID\tscore\t2ndBestScore\thitLength\tqueryLength\tnumMatches\n"
        files = {
            "odd_taxids.tsv": std + "r1\tseq0\t1000\t7225\t0\t100\t100\t1\nr2\tx\t424242\t400\t0\t35\t100\t1\nr3\tunclassified\t0\t0\t0\t0\t100\t1\n"
                                    "r4\tseq1\t1001\t900\t900\t45\t100\t3\nr4\tseq9\t1009\t900\t900\t45\t100\t3\nr4\tseq17\t1017\t900\t900\t45\t100\t3\n"
                                    "r5\tgenus\t100\t81\t0\t24\t100\t2\nr5\tseq2\t1002\t81\t0\t24\t100\t2\nr6\ty\t50\t64\t0\t23\t100\t1\n",
            "reordered.tsv": "numMatches\ttaxID\treadID\tseqID\textra\thitLength\tscore\tqueryLength\n"
                             "2\t1003\tq1\tseq3\tzz\t60\t2025\t100\n2\t1004\tq1\tseq4\tzz\t60\t2025\t100\n1\t1010\tq2\tseq10\tzz\t99\t7056\t100\n",
        }
        for fn, text in files.items():
            p = os.path.join(t, fn)
            open(p, "w").write(text)
            for opts in ([], ["--no-lca"], ["--show-zeros"], ["--min-score", "100"], ["--min-length", "40"]):
                want = subprocess.run(["perl", os.path.join(t, "centrifuge-kreport"), "-x", idx] + opts + [p], capture_output=True)
                got = subprocess.run([KREPORT, "-x", idx] + opts + [p], capture_output=True)
                assert got.returncode == want.returncode and got.stdout == want.stdout, (fn, opts, got.stdout[:400], want.stdout[:400])
        p = os.path.join(t, "odd_taxids.tsv")
        for level in ("species", "genus", "family", "lca", "nosuchlevel"):
            want = subprocess.run(["perl", os.path.join(t, "centrifuge-promote"), idx, p, level], capture_output=True)
            got = subprocess.run([PROMOTE, idx, p, level], capture_output=True)
            assert got.stdout == want.stdout, (level, got.stdout[:400], want.stdout[:400])


def test_index_lookup_through_centrifuge_indexes():
    """adjustEbwtBase (bt2_idx.cpp:38-66): a basename that is not a path is looked up under $CENTRIFUGE_INDEXES"""
    import shutil
    d = common.golden("example")[0]
    with tempfile.TemporaryDirectory() as t:
        for e in ("1", "2", "3", "4"):
            shutil.copy(os.path.join(d, "idx.%s.cf" % e), os.path.join(t, "moved.%s.cf" % e))
        env = dict(os.environ, CENTRIFUGE_INDEXES=t)
        r = subprocess.run([INSPECT, "-n", "moved"], capture_output=True, env=env, cwd="/")
        assert r.returncode == 0 and r.stdout == open(os.path.join(common.golden("tools")[0], "inspect/example.names.txt"), "rb").read()
        r = subprocess.run([KREPORT, "-x", "moved", os.path.join(d, "default.tsv")], capture_output=True, env=env, cwd="/")
        assert r.returncode == 0 and r.stdout == open(os.path.join(common.golden("tools")[0], "kreport/example.default.lca.txt"), "rb").read()
        r = subprocess.run([INSPECT, "-n", "moved"], capture_output=True, cwd="/")
        assert r.returncode == 1 and b"Could not locate a Centrifuge index" in r.stderr
