"""centrifuge-class over the device text path (round 6): plain FASTA / FASTQ files go up as text in blocks, the default columns
come back as text — the same bytes as the host parser + host formatter give, as the reference binary gives; blocks that hold a
record outside the plain form take the host parser, one by one."""
import gzip
import os
import re
import subprocess
import tempfile

import numpy as np
import pytest

import common
from oracle import oracle as O

pytestmark = pytest.mark.gpu
CLI = os.path.join(common.ROOT, "centrifuge_amd", "bin", "centrifuge-class")


def run(args, d, env=None, exe=CLI, tag="o"):
    out, rep = os.path.join(d, tag + ".tsv"), os.path.join(d, tag + ".rep")
    p = subprocess.Popen([exe] + args + ["-S", out, "--report-file", rep], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=dict(os.environ, **(env or {})))
    try:
        _, err = p.communicate(timeout=180)
    except subprocess.TimeoutExpired:
        # a run that hangs: where its threads are (rocgdb is in the image), then an end to it
        bt = subprocess.run(["/opt/rocm/bin/rocgdb", "-p", str(p.pid), "-batch", "-ex", "thread apply all bt 14"], capture_output=True, text=True, timeout=120)
        p.kill()
        raise AssertionError("the run hangs: %s\n%s" % (" ".join([exe] + args), bt.stdout[-12000:]))
    assert p.returncode == 0, err
    return open(out, "rb").read(), open(rep, "rb").read(), err


def blocks(err):
    m = re.search(r"Device text path: (\d+) block\(s\) parsed and printed on the device, (\d+) on the host", err)
    return (int(m.group(1)), int(m.group(2))) if m else None


@pytest.mark.parametrize("name,fmt,reads", [("k5", "-f", "reads.fa"), ("fastq", "-q", "reads.fq"), ("r250_k5", "-f", "reads250.fa")])
def test_text_path_is_taken_and_prints_what_the_host_paths_print(name, fmt, reads):
    d, cases = common.golden("synth_small")
    c = [x for x in cases if x["name"] == name][0]
    args = [fmt, "-t", "-p", "4", "-x", os.path.join(d, "idx"), "-U", os.path.join(d, reads)]
    want = open(os.path.join(d, c["tsv"]), "rb").read(), open(os.path.join(d, c["report"]), "rb").read()
    with tempfile.TemporaryDirectory() as t:
        tsv, rep, err = run(args, t)
        assert (tsv, rep) == want and blocks(err) == (1, 0), err
        # many small blocks; every block through the host parser; the path off
        tsv, rep, err = run(args, t, env={"CF_TEXT_BLOCK": "4096"})
        assert (tsv, rep) == want and blocks(err)[0] > 10 and blocks(err)[1] == 0
        tsv, rep, err = run(args, t, env={"CF_TEXT_BLOCK": "8192", "CF_CLI_TEXT_HOST_PARSE": "1"})
        assert (tsv, rep) == want and blocks(err) is None or blocks(err)[0] == 0
        tsv, rep, err = run(args, t, env={"CF_CLI_DEVICE_TEXT": "0"})
        assert (tsv, rep) == want and blocks(err) is None
        # -u inside a block, at a block's end, past the file; one slot; the output into a pipe (written in order)
        for u in ("1", "37", "100000"):
            a = run(args + ["-u", u], t, env={"CF_TEXT_BLOCK": "4096"}, tag="a")
            b = run(args + ["-u", u], t, env={"CF_CLI_DEVICE_TEXT": "0"}, tag="b")
            assert a[:2] == b[:2], u
        tsv1, rep1, _ = run(args + ["--slots", "1"], t, env={"CF_TEXT_BLOCK": "4096"})
        assert (tsv1, rep1) == want
        r = subprocess.run([CLI] + args + ["--report-file", os.path.join(t, "p.rep")], capture_output=True, env=dict(os.environ, CF_TEXT_BLOCK="4096"))
        assert r.returncode == 0 and r.stdout == want[0] and open(os.path.join(t, "p.rep"), "rb").read() == want[1]


@pytest.mark.skipif(not O.have_ref(), reason="oracle/_ref not built")
@pytest.mark.parametrize("fastq", [False, True])
def test_blocks_outside_the_plain_form_take_the_host_parser(fastq):
    """a file of plain records with a few odd ones strewn in (CR LF, lower case, an ambiguity letter, no name, wrapped lines, a
    blank line, a '.' for an N, one quality value too many): small blocks, so that some go up as text and some are parsed on the host —
    against the reference binary, TSV and report"""
    d, _ = common.golden("synth_small")
    src = open(os.path.join(d, "reads.fq" if fastq else "reads.fa"), "rb").read()
    if fastq:
        ls = src.split(b"\n")[:-1]
        recs = [b"\n".join(ls[k:k + 4]) + b"\n" for k in range(0, len(ls), 4)]
    else:
        recs = [b">" + r for r in src.split(b"\n>")]
        recs[0] = recs[0][1:]
        recs = [r if r.endswith(b"\n") else r + b"\n" for r in recs]
    rng = np.random.default_rng(3 + fastq)
    out = []
    for i, r in enumerate(recs):
        kind = int(rng.integers(0, 250))
        lines = r.split(b"\n")[:-1]
        if kind == 0:
            r = b"\r\n".join(lines) + b"\r\n"
        elif kind == 1:
            lines[1] = lines[1].lower(); r = b"\n".join(lines) + b"\n"
        elif kind == 2:
            lines[1] = lines[1][:7] + b"R" + lines[1][8:]; r = b"\n".join(lines) + b"\n"
        elif kind == 3:
            lines[0] = lines[0][:1]; r = b"\n".join(lines) + b"\n"                      # no name: named after its ordinal
        elif kind == 4 and not fastq:
            s = lines[1]; r = lines[0] + b"\n" + b"\n".join(s[k:k + 30] for k in range(0, len(s), 30)) + b"\n"
        elif kind == 5 and not fastq:
            r = r + b"\n"
        elif kind == 6:
            lines[1] = lines[1][:3] + b"." + lines[1][4:]; r = b"\n".join(lines) + b"\n"
        elif kind == 7 and fastq:
            lines[3] = lines[3] + b"I"; r = b"\n".join(lines) + b"\n"
        elif kind == 8:
            lines[0] = lines[0] + b" a comment/2"; r = b"\n".join(lines) + b"\n"
        out.append(r)
    text = b"".join(out)
    ref_exe = os.path.join(O.REF_DIR, "centrifuge-class")
    with tempfile.TemporaryDirectory() as t:
        p = os.path.join(t, "odd.fq" if fastq else "odd.fa")
        open(p, "wb").write(text)
        for extra in ([], ["-u", "777"], ["-k", "1", "--seed", "99"]):
            args = ["-q" if fastq else "-f", "-t", "-x", os.path.join(d, "idx"), "-U", p] + extra
            want = run(args, t, exe=ref_exe, tag="ref")                       # (one thread: the reference's output order)
            got = run(args + ["-p", "3"], t, env={"CF_TEXT_BLOCK": "6000"})
            assert got[0] == want[0], common.first_diff(got[0].decode("latin1"), want[0].decode("latin1"))
            assert got[1] == want[1]
            nb = blocks(got[2])
            assert nb and nb[0] >= 3 and nb[1] >= 3, got[2]


@pytest.mark.skipif(not O.have_ref(), reason="oracle/_ref not built")
def test_plain_and_compressed_inputs_and_mates_in_one_run():
    d, _ = common.golden("synth_small")
    ref_exe = os.path.join(O.REF_DIR, "centrifuge-class")
    u1, u2 = os.path.join(d, "reads.fa"), os.path.join(d, "reads250.fa")
    with tempfile.TemporaryDirectory() as t:
        gz = os.path.join(t, "r.fa.gz")
        with gzip.open(gz, "wb") as f:
            f.write(open(u2, "rb").read())
        plain2 = os.path.join(t, "r2.fa")
        open(plain2, "wb").write(open(u2, "rb").read())
        for reads_ref, reads in ((["-U", u1 + "," + plain2 + "," + u1], ["-U", u1 + "," + gz + "," + u1]),
                                 (["-1", os.path.join(d, "r1.fa"), "-2", os.path.join(d, "r2.fa"), "-U", u1 + "," + plain2],) * 2):
            base = ["-f", "-t", "-x", os.path.join(d, "idx")]
            want = run(base + reads_ref, t, exe=ref_exe, tag="ref")           # (one thread: with -p 4 --reorder and several inputs the reference hangs)
            got = run(base + ["-p", "4"] + reads, t, env={"CF_TEXT_BLOCK": "20000"})
            assert got[:2] == want[:2], common.first_diff(got[0].decode("latin1"), want[0].decode("latin1"))
            assert blocks(got[2])[0] >= 10


@pytest.mark.skipif(not O.have_ref(), reason="oracle/_ref not built")
@pytest.mark.parametrize("fastq", [False, True])
def test_mates_go_up_as_two_text_blocks(fastq):
    """-1 / -2 of plain files: the second file is cut where it holds the records of the first file's block; odd records in either
    file send their blocks to the host parser; a second file that stops keeping to four lines per record changes the rest of the
    input over to the parser pool — against the reference binary, and against this binary's parser pool for -u"""
    d, cases = common.golden("synth_small")
    ref_exe = os.path.join(O.REF_DIR, "centrifuge-class")
    m1, m2 = (open(os.path.join(d, f), "rb").read() for f in ("r1.fa", "r2.fa"))

    def records(src):
        recs = [b">" + r for r in src.split(b"\n>")]
        recs[0] = recs[0][1:]
        return [r if r.endswith(b"\n") else r + b"\n" for r in recs]
    r1, r2 = records(m1), records(m2)
    if fastq:
        def fq(recs, seed):
            rng = np.random.default_rng(seed)
            out = []
            for r in recs:
                name, seq = r.split(b"\n")[:2]
                out.append(b"@" + name[1:] + b"\n" + seq + b"\n+\n" + bytes(int(q) for q in rng.integers(33, 127, len(seq))) + b"\n")
            return out
        r1, r2 = fq(r1, 1), fq(r2, 2)
    fmt, ext = ("-q", ".fq") if fastq else ("-f", ".fa")
    with tempfile.TemporaryDirectory() as t:
        f1, f2 = os.path.join(t, "a" + ext), os.path.join(t, "b" + ext)

        def check(a, b, extra=(), want_blocks=None, block="3000"):
            open(f1, "wb").write(b"".join(a)); open(f2, "wb").write(b"".join(b))
            args = [fmt, "-t", "-x", os.path.join(d, "idx"), "-1", f1, "-2", f2] + list(extra)
            want = run(args, t, exe=ref_exe, tag="ref")
            got = run(args + ["-p", "4"], t, env={"CF_TEXT_BLOCK": block})
            assert got[0] == want[0], common.first_diff(got[0].decode("latin1"), want[0].decode("latin1"))
            assert got[1] == want[1]
            if want_blocks:
                nb = blocks(got[2])
                assert nb and nb[0] >= want_blocks[0] and nb[1] >= want_blocks[1], got[2]
            return got
        check(r1, r2, want_blocks=(10, 0))
        check(r1, r2, extra=["-u", "123", "-k", "2"], want_blocks=(2, 0))
        # odd records in the second file only (CR LF, an ambiguity letter, no name): their blocks take the host parser
        odd = list(r2)
        for i in range(7, len(odd), 41):
            ls = odd[i].split(b"\n")[:-1]
            k = (i // 41) % 3
            if k == 0:
                odd[i] = b"\r\n".join(ls) + b"\r\n"
            elif k == 1:
                ls[1] = ls[1][:5] + b"R" + ls[1][6:]; odd[i] = b"\n".join(ls) + b"\n"
            else:
                ls[0] = ls[0][:1]; odd[i] = b"\n".join(ls) + b"\n"
        check(r1, odd, want_blocks=(3, 3), block="5000")
        # the second file stops keeping to the device's record rule half way (FASTA: no such thing — a '>' is a record; FASTQ: wrapped
        # sequence lines): the rest of the input goes through the parser pool
        if fastq:
            wrapped = list(r2)
            for i in range(len(wrapped) // 2, len(wrapped)):
                ls = wrapped[i].split(b"\n")[:-1]
                wrapped[i] = ls[0] + b"\n" + ls[1][:40] + b"\n" + ls[1][40:] + b"\n" + ls[2] + b"\n" + ls[3] + b"\n"
            got = check(r1, wrapped, block="5000")
            nb = blocks(got[2])
            assert nb and nb[0] >= 3, got[2]
        # unequal files: the messages of the reference
        open(f1, "wb").write(b"".join(r1)); open(f2, "wb").write(b"".join(r2[:-3]))
        r = subprocess.run([CLI, fmt, "-x", os.path.join(d, "idx"), "-1", f1, "-2", f2, "-S", os.path.join(t, "x.tsv")], capture_output=True, text=True,
                           env=dict(os.environ, CF_TEXT_BLOCK="5000"), timeout=180)
        assert r.returncode == 1 and "fewer reads in file specified with -2 than in file specified with -1" in r.stderr


@pytest.mark.parametrize("fastq", [False, True])
def test_random_files_print_the_same_through_both_ways(fastq):
    """the end-to-end form of tests/test_textio_emu.py's safety property: files of the golden reads with random damage (line ends,
    stray '>' '@' '+', blanks, ambiguity letters, deleted bytes) in small blocks — whatever the device takes or refuses, block by
    block, the run prints what the host threads print (--host-io), or fails with the same message"""
    d, _ = common.golden("synth_small")
    src = open(os.path.join(d, "reads.fq" if fastq else "reads.fa"), "rb").read()
    rng = np.random.default_rng(99 + fastq)
    junk = [b"\r", b">", b"@", b"+", b"\n", b".", b"-", b"R", b"n", b" ", b"\t", b"/1", b"\n\n", b"*", b"acgt"]
    same = failed = 0
    with tempfile.TemporaryDirectory() as t:
        p = os.path.join(t, "x.fq" if fastq else "x.fa")
        for trial in range(24):
            cut = int(rng.integers(20000, len(src)))
            if fastq:                                            # (whole records: a FASTQ file cut in mid-record is an error both ways, and says little)
                cut = len(b"\n".join(src[:cut].split(b"\n")[:-1]).rsplit(b"\n@", 1)[0]) + 1
            text = bytearray(src[:cut])
            for _ in range(int(rng.integers(0, 3 if fastq else 12))):
                at = int(rng.integers(0, len(text)))
                if rng.random() < 0.7:
                    text[at:at] = junk[int(rng.integers(0, len(junk)))]
                else:
                    del text[at:at + int(rng.integers(1, 5))]
            open(p, "wb").write(bytes(text))
            args = ["-q" if fastq else "-f", "-p", "4", "-x", os.path.join(d, "idx"), "-U", p]
            outs = []
            for extra, env in ((["--host-io"], {}), ([], {"CF_TEXT_BLOCK": str(int(rng.choice([3000, 9000, 40000])))})):
                r = subprocess.run([CLI] + args + extra + ["-S", os.path.join(t, "o.tsv"), "--report-file", os.path.join(t, "o.rep")], capture_output=True, text=True,
                                   env=dict(os.environ, **env), timeout=180)
                outs.append((r.returncode, open(os.path.join(t, "o.tsv"), "rb").read() if r.returncode == 0 else None,
                             open(os.path.join(t, "o.rep"), "rb").read() if r.returncode == 0 else None,
                             [ln for ln in r.stderr.splitlines() if ln.startswith(("Error", "Saw ASCII"))][:1]))
            assert outs[0] == outs[1], (trial, outs[0][0], outs[1][0], outs[0][3], outs[1][3])
            same += outs[0][0] == 0
            failed += outs[0][0] != 0
    assert same >= (4 if fastq else 8) and same + failed == 24      # (enough damaged files still parse: the comparison did run)
