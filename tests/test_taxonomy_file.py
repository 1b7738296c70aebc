"""<base>.3.cf (uid table, pruned taxonomy, names, sizes: bt2_idx.h:1375-1504) written by the host half of the
builder (cf_build_taxonomy, no device) against the file the reference builder writes (oracle/_ref, run on the CPU)
for awkward inputs: sequences missing from the conversion table, table entries without a sequence, unused and
orphan nodes, unusual ranks, names with blanks, a size table, "cid" uids (compressed index)."""
import filecmp
import os
import subprocess
import tempfile

import numpy as np
import pytest

from centrifuge_amd import capi
from oracle import oracle as O

pytestmark = pytest.mark.skipif(not O.have_ref(), reason="oracle/_ref not built")


def write(path, text):
    with open(path, "w") as f:
        f.write(text)


def seqs(names, rng, n_runs=()):
    out = []
    for i, nm in enumerate(names):
        s = bytes(np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, 200 + 17 * i)]).decode()
        if i in n_runs:
            s = s[:50] + "N" * 13 + s[63:]
        out.append(">%s\n%s\n" % (nm, s))
    return "".join(out)


CASES = {}


def case(fn):
    CASES[fn.__name__] = fn
    return fn


@case
def plain(d, rng):
    write(d + "/g.fa", seqs(["s0 first", "s1", "s2|v2 x"], rng))
    write(d + "/conv", "s0\t10\ns1\t11\ns2|v2\t12\n")
    write(d + "/nodes", "1\t|\t1\t|\tno rank\n5\t|\t1\t|\tgenus\n10\t|\t5\t|\tspecies\n11\t|\t5\t|\tspecies\n12\t|\t1\t|\tspecies\n")
    write(d + "/names", "1\t|\troot\t|\t\t|\tscientific name\t|\n5\t|\tGenus five\t|\t\t|\tscientific name\t|\n"
                        "10\t|\tGenus five alpha\t|\t\t|\tscientific name\t|\n10\t|\tsynonym ten\t|\t\t|\tsynonym\t|\n"
                        "11\t|\tGenus  five   beta\t|\t\t|\tscientific name\t|\n12\t|\tLonely\t|\t\t|\tscientific name\t|\n")
    return {}


@case
def missing_and_surplus_entries(d, rng):
    write(d + "/g.fa", seqs(["a", "b", "c", "d"], rng, n_runs=(1,)))
    write(d + "/conv", "a\t100\nzz_not_in_fasta\t101\nc\t102\nd\t100\n")           # b has no entry; two sequences share a taxon
    write(d + "/nodes", "1\t|\t1\t|\tno rank\n2\t|\t1\t|\tsuperkingdom\n50\t|\t2\t|\tfamily\n60\t|\t50\t|\tgenus\n"
                        "100\t|\t60\t|\tspecies\n101\t|\t60\t|\tspecies\n102\t|\t999\t|\tstrain\n777\t|\t2\t|\tphylum\n")   # 102: orphan parent, 777 unused
    write(d + "/names", "1\t|\troot\t|\t\t|\tscientific name\t|\n100\t|\tHundred\t|\t\t|\tscientific name\t|\n777\t|\tUnused\t|\t\t|\tscientific name\t|\n")
    return {}


@case
def ranks_and_sizes(d, rng):
    write(d + "/g.fa", seqs(["q%d" % i for i in range(6)], rng))
    write(d + "/conv", "".join("q%d\t%d\n" % (i, 200 + i) for i in range(6)))
    ranks = ["subspecies", "strain", "species", "no rank", "clade", "varietas"]
    write(d + "/nodes", "1\t|\t1\t|\tno rank\n20\t|\t1\t|\tkingdom\n21\t|\t20\t|\tclass\n22\t|\t21\t|\torder\n23\t|\t22\t|\tspecies\n" +
          "".join("%d\t|\t23\t|\t%s\n" % (200 + i, r) for i, r in enumerate(ranks)))
    write(d + "/names", "".join("%d\t|\tName of %d\t|\t\t|\tscientific name\t|\n" % (t, t) for t in (1, 20, 21, 22, 23, 200, 201, 202, 203, 204, 205)))
    write(d + "/sizes", "200\t12345\n203\t777\n999\t5\n")
    return {"size_table": d + "/sizes"}


@case
def compressed_cid_uids(d, rng):
    names = ["cid|%d|x" % i for i in range(12)]
    write(d + "/g.fa", seqs(names, rng))
    write(d + "/conv", "".join("%s\t%d\n" % (n, 300 + i // 2) for i, n in enumerate(names)))
    write(d + "/nodes", "1\t|\t1\t|\tno rank\n30\t|\t1\t|\tgenus\n" + "".join("%d\t|\t30\t|\tspecies\n" % (300 + i) for i in range(6)))
    write(d + "/names", "1\t|\troot\t|\t\t|\tscientific name\t|\n30\t|\tCompressed genus\t|\t\t|\tscientific name\t|\n")
    return {}


@case
def diverging_duplicates_comments_and_ncbi_columns(d, rng):
    write(d + "/g.fa", seqs(["u1", "u2.1 with version", "u3"], rng))
    write(d + "/conv", "# comment line\nu1\t400\nu1\t401\nu2.1\t402.7\n\nu3 403\nu2\t499\n")      # first wins; 402.7 = hi/lo split; blank separated
    full = lambda t, p, r: "%d\t|\t%d\t|\t%s\t|\t\t|\t0\t|\t1\t|\t11\t|\t1\t|\t0\t|\t1\t|\t1\t|\t0\t|\t\t|\n" % (t, p, r)   # noqa: E731
    write(d + "/nodes", full(1, 1, "no rank") + full(40, 1, "family") + full(400, 40, "species") + full(401, 40, "species") +
          full(402, 40, "subspecies") + full(403, 400, "no rank"))
    write(d + "/names", "1\t|\troot\t|\t\t|\tscientific name\t|\n400\t|\tFour hundred\t|\tFour hundred <x>\t|\tscientific name\t|\n"
                        "403\t|\tChild@of 400\t|\t\t|\tscientific name\t|\n40\t|\tFam\t|\t\t|\tauthority\t|\n")
    return {}


@pytest.mark.parametrize("name", sorted(CASES))
def test_taxonomy_file_matches_reference_builder(name):
    rng = np.random.default_rng(len(name))
    with tempfile.TemporaryDirectory() as d:
        extra = CASES[name](d, rng)
        cmd = [os.path.join(O.REF_DIR, "centrifuge-build-bin"), "--conversion-table", d + "/conv", "--taxonomy-tree", d + "/nodes",
               "--name-table", d + "/names"]
        if "size_table" in extra:
            cmd += ["--size-table", extra["size_table"]]
        r = subprocess.run(cmd + [d + "/g.fa", d + "/ref"], capture_output=True, text=True)
        assert r.returncode == 0, r.stderr[-2000:]
        capi.build_taxonomy(d + "/ours", [d + "/g.fa"], d + "/conv", d + "/nodes", d + "/names", extra.get("size_table"))
        # the loader's view of the reference-built index: every table the inspector prints (size roll-up included)
        mine = os.path.join(os.path.dirname(capi.LIB_PATH), "bin", "centrifuge-inspect-bin")
        for mode in ("-n", "-s", "--conversion-table", "--taxonomy-tree", "--name-table", "--size-table"):
            want = subprocess.run([os.path.join(O.REF_DIR, "centrifuge-inspect-bin"), mode, d + "/ref"], capture_output=True).stdout
            got = subprocess.run([mine, mode, d + "/ref"], capture_output=True).stdout
            assert got == want, (name, mode, got[:300], want[:300])
        a, b = open(d + "/ref.3.cf", "rb").read(), open(d + "/ours.3.cf", "rb").read()
        assert a == b, "%s: .3.cf differs (ref %d bytes, ours %d), first at %d" % (
            name, len(a), len(b), next((i for i, (x, y) in enumerate(zip(a, b)) if x != y), min(len(a), len(b))))


def test_corrupt_taxonomy_is_a_format_error_not_a_hang():
    """a .3.cf whose tree has a parent cycle (A -> B -> A) used to hang cf_index_open; a count larger than the
    file used to reach reserve().  Both are format errors now."""
    import struct
    import common
    d, _ = common.golden("example")
    with tempfile.TemporaryDirectory() as t:
        for k in (1, 2, 4):
            os.symlink(os.path.join(d, "idx.%d.cf" % k), os.path.join(t, "x.%d.cf" % k))

        def tax3(nodes, nref_claim=None):
            b = struct.pack("<iQ", 1, 1 if nref_claim is None else nref_claim) + b"seq0\0" + struct.pack("<Q", 10)
            b += struct.pack("<Q", len(nodes))
            for tid, par, rank in nodes:
                b += struct.pack("<QQH", tid, par, rank)
            b += struct.pack("<Q", 0) + struct.pack("<Q", 1) + struct.pack("<QQ", 10, 1000)
            open(os.path.join(t, "x.3.cf"), "wb").write(b)

        tax3([(10, 11, 1), (11, 10, 2)])                       # cycle 10 -> 11 -> 10
        with pytest.raises(capi.CfError, match="cycle"):
            capi.Index(os.path.join(t, "x"), host_only=True)
        tax3([(10, 1, 1), (1, 1, 0)], nref_claim=1 << 60)      # absurd count
        with pytest.raises(capi.CfError, match="exceeds the file size"):
            capi.Index(os.path.join(t, "x"), host_only=True)
        tax3([(10, 1, 1), (1, 1, 0)])                          # and the well-formed one loads
        capi.Index(os.path.join(t, "x"), host_only=True).close()
