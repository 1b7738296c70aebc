"""The builder's sequence bookkeeping (FASTA -> names, lengths, fragment table, joined text: fastaRefReadSize
ref_read.cpp:28-186, szsToDisk bt2_idx.h:3255-3345) on the CPU, against the reference builder + inspector run on the
same awkward FASTA files: the header of <base>.1.cf (sequence count, plen, fragment table), its name section, and the
sequences `centrifuge-inspect` prints back (gaps restored) must be what cf_build_describe reports."""
import os
import subprocess
import tempfile

import numpy as np
import pytest

from centrifuge_amd import capi
from oracle import oracle as O

pytestmark = pytest.mark.skipif(not O.have_ref(), reason="oracle/_ref not built")


def rnd(rng, n):
    return bytes(np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, n)]).decode()


def ref_header(path):
    raw = open(path, "rb").read()
    u64 = lambda o: int(np.frombuffer(raw, dtype="<u8", count=1, offset=o)[0])   # noqa: E731
    assert int(np.frombuffer(raw, dtype="<i4", count=1)[0]) == 1
    n = u64(4)
    o = 4 + 8 + 5 * 4
    npat = u64(o)
    plen = np.frombuffer(raw, dtype="<u8", count=npat, offset=o + 8)
    o += 8 + 8 * npat
    nfrag = u64(o)
    rst = np.frombuffer(raw, dtype="<u8", count=3 * nfrag, offset=o + 8).reshape(-1, 3)
    return n, plen, rst


def reconstruct(desc):
    """the FASTA centrifuge-inspect prints (print_index_sequences, centrifuge_inspect.cpp:369-430) from the description"""
    out = []
    seqs = {}
    order = []
    n = desc["len"]
    rst = desc["rstarts"]
    for fi in range(len(rst)):
        lo, tidx, toff = (int(x) for x in rst[fi])
        hi = int(rst[fi + 1][0]) if fi + 1 < len(rst) else n
        if tidx not in seqs:
            seqs[tidx] = bytearray(b"N" * int(desc["plen"][tidx]))
            order.append(tidx)
        seqs[tidx][toff:toff + hi - lo] = np.frombuffer(b"ACGT", dtype=np.uint8)[desc["text"][lo:hi]].tobytes()
    for t in order:
        out.append(b">" + desc["names"][t] + b"\n")
        s = bytes(seqs[t])
        out += [s[i:i + 60] + b"\n" for i in range(0, len(s), 60)]
    return b"".join(out)


def fasta_cases(rng):
    r = lambda n: rnd(rng, n)   # noqa: E731
    return {
        "plain_multiline": ">a first\n%s\n%s\n>b\n%s\n" % (r(70), r(33), r(120)),
        "gaps_everywhere": ">lead\nNNNN%s\n>trail\n%sNNNNN\n>mid\n%sNN%sN%s\n>both\nN%sN\n" % (r(50), r(40), r(30), r(31), r(32), r(45)),
        "all_gap_in_the_middle": ">x\n%s\n>allgap\nNNNNNNNNNN\nNNNN\n>y\n%s\n" % (r(60), r(61)),
        "empty_records_and_blank_lines": ">e1\n>e2\n\n>real\n\n%s\n\n%s\n>e3\n>real2\n%s\n>e4\n" % (r(40), r(20), r(35)),
        "lowercase_iupac_dash": ">m\n%s\nacgtRYKMSWBDHVn-%s\n>n\n%s\n" % (r(30).lower(), r(25), r(50)),
        "crlf": ">c1 x\r\n%s\r\n%s\r\n>c2\r\n%s\r\n" % (r(30), r(30), r(44)),
        "no_final_newline": ">z\n%s\n>w\n%s" % (r(40), r(41)),
        "trailing_gap_then_new_sequence": ">p\n%sNNN\n>q\nNN%s\n" % (r(33), r(34)),
        "short_sequences": ">s1\nA\n>s2\nAC\n>s3\n%s\n" % r(12),
        "first_sequence_all_gaps": ">g0\nNNNNNNNNNNNNNNN\n>g1\n-----------%s\n>g2\nNN\n>g3\n%s\n" % (r(118), r(40)),
        "spaces_inside_sequence": ">t\n%s %s\t%s\n" % (r(10), r(10), r(10)),
    }


@pytest.mark.parametrize("name", sorted(fasta_cases(np.random.default_rng(0))))
def test_bookkeeping_matches_reference(name):
    text = fasta_cases(np.random.default_rng(0))[name]
    with tempfile.TemporaryDirectory() as d:
        fa = os.path.join(d, "g.fa")
        with open(fa, "w", newline="") as f:
            f.write(text)
        open(d + "/conv", "w").write("nothing\t1\n")
        open(d + "/nodes", "w").write("1\t|\t1\t|\tno rank\n")
        open(d + "/names", "w").write("1\t|\troot\t|\t\t|\tscientific name\t|\n")
        r = subprocess.run([os.path.join(O.REF_DIR, "centrifuge-build-bin"), "--conversion-table", d + "/conv", "--taxonomy-tree", d + "/nodes",
                            "--name-table", d + "/names", fa, d + "/ref"], capture_output=True, text=True)
        assert r.returncode == 0, r.stderr[-1500:]
        n, plen, rst = ref_header(d + "/ref.1.cf")
        desc = capi.build_describe([fa])
        assert desc["len"] == n
        assert np.array_equal(desc["plen"], plen), (desc["plen"], plen)
        assert np.array_equal(desc["rstarts"], rst), (desc["rstarts"], rst)
        names = subprocess.run([os.path.join(O.REF_DIR, "centrifuge-inspect-bin"), "-n", d + "/ref"], capture_output=True).stdout
        assert b"".join(x + b"\n" for x in desc["names"]) == names
        fasta = subprocess.run([os.path.join(O.REF_DIR, "centrifuge-inspect-bin"), d + "/ref"], capture_output=True).stdout
        assert reconstruct(desc) == fasta


def test_several_fasta_files():
    rng = np.random.default_rng(4)
    with tempfile.TemporaryDirectory() as d:
        a, b, c = d + "/a.fa", d + "/b.fa", d + "/c.fa"
        open(a, "w").write(">a1\n%s\n>a2\n%sNN" % (rnd(rng, 50), rnd(rng, 30)))          # ends in a gap, no final newline
        open(b, "w").write(">b1\nNN%s\n" % rnd(rng, 40))
        open(c, "w").write("\n\n>c1\n%s\n>c2\n\n" % rnd(rng, 25))
        open(d + "/conv", "w").write("nothing\t1\n")
        open(d + "/nodes", "w").write("1\t|\t1\t|\tno rank\n")
        open(d + "/names", "w").write("1\t|\troot\t|\t\t|\tscientific name\t|\n")
        r = subprocess.run([os.path.join(O.REF_DIR, "centrifuge-build-bin"), "--conversion-table", d + "/conv", "--taxonomy-tree", d + "/nodes",
                            "--name-table", d + "/names", ",".join([a, b, c]), d + "/ref"], capture_output=True, text=True)
        assert r.returncode == 0, r.stderr[-1500:]
        n, plen, rst = ref_header(d + "/ref.1.cf")
        desc = capi.build_describe([a, b, c])
        assert desc["len"] == n and np.array_equal(desc["plen"], plen) and np.array_equal(desc["rstarts"], rst)
        fasta = subprocess.run([os.path.join(O.REF_DIR, "centrifuge-inspect-bin"), d + "/ref"], capture_output=True).stdout
        assert reconstruct(desc) == fasta


def _to_memory(text):
    """the sequences of a FASTA text as the builder's in-memory input: codes (0..3, 4 = gap), offsets, names"""
    names, seqs = [], []
    for rec in text.split(">")[1:]:
        lines = rec.replace("\r", "").split("\n")
        names.append(lines[0].encode())
        seqs.append("".join(lines[1:]).replace(" ", "").replace("\t", ""))
    lut = np.full(256, 4, dtype=np.uint8)
    for i, ch in enumerate("ACGT"):
        lut[ord(ch)] = i
        lut[ord(ch.lower())] = i
    codes = np.concatenate([lut[np.frombuffer(s.encode(), dtype=np.uint8)] for s in seqs] + [np.zeros(0, dtype=np.uint8)])
    off = np.zeros(len(seqs) + 1, dtype=np.uint64)
    np.cumsum([len(s) for s in seqs], out=off[1:])
    return codes, off, names


@pytest.mark.parametrize("name", ["all_gap_in_the_middle", "first_sequence_all_gaps", "gaps_everywhere", "trailing_gap_then_new_sequence",
                                  "short_sequences", "plain_multiline"])
def test_memory_input_matches_fasta_input(name):
    """codes + seq_off input goes through the same bookkeeping as the FASTA path — including all-gap and
    zero-length sequences in the middle, which have no pattern of their own (the copy loop used to
    index the per-pattern joined starts by input sequence number)."""
    text = fasta_cases(np.random.default_rng(0))[name]
    with tempfile.TemporaryDirectory() as d:
        fa = os.path.join(d, "g.fa")
        with open(fa, "w", newline="") as f:
            f.write(text)
        want = capi.build_describe([fa])
    codes, off, names = _to_memory(text)
    got = capi.build_describe(codes=codes, seq_off=off, seq_names=names)
    assert got["len"] == want["len"]
    assert np.array_equal(got["plen"], want["plen"]) and np.array_equal(got["rstarts"], want["rstarts"])
    assert got["names"] == want["names"]
    assert np.array_equal(got["text"], want["text"])


def test_memory_input_awkward_mix():
    """the ASAN reproducer of ADVICE r1: ACGT / NN / ANC / GGTT, plus an empty sequence"""
    seqs = ["ACGT", "NN", "ANC", "", "GGTT", "NNNN", "TTNA"]
    text = "".join(">s%d\n%s\n" % (i, s) for i, s in enumerate(seqs))
    codes, off, names = _to_memory(text)
    got = capi.build_describe(codes=codes, seq_off=off, seq_names=names)
    joined = "".join(s.replace("N", "") for s in seqs)
    assert got["len"] == len(joined)
    assert bytes(np.frombuffer(b"ACGT", dtype=np.uint8)[got["text"]]).decode() == joined


def test_memory_input_rejects_decreasing_offsets():
    codes = np.zeros(16, dtype=np.uint8)
    off = np.array([0, 8, 4, 16], dtype=np.uint64)
    with pytest.raises(capi.CfError):
        capi.build_describe(codes=codes, seq_off=off, seq_names=[b"a", b"b", b"c"])
