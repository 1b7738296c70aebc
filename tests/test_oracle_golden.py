"""The CPU restatement (oracle/) against the golden vectors: the reference's own
worked example (example/ + MANUAL:1012-1028) and generated cases, all produced
by the compiled, unmodified reference (tests/golden/make_golden.py)."""
import os

import numpy as np
import pytest

import common
from oracle import oracle as O


@pytest.mark.parametrize("arch,name", common.all_cases())
def test_oracle_matches_reference_tsv(arch, name):
    d, cases = common.golden(arch)
    c = [x for x in cases if x["name"] == name][0]
    kw, fastq = common.case_kwargs(c["args"])
    orc = O.Oracle(os.path.join(d, "idx"))
    files = [os.path.join(d, f) for f in c["reads"]]
    got = orc.classify_files(*files, fastq=fastq, **kw)
    ref = open(os.path.join(d, c["tsv"])).read()
    assert got == ref, common.first_diff(got, ref)


def test_manual_table_rows():
    """MANUAL:1012-1028: the 7 columns the manual prints for the worked example."""
    d, cases = common.golden("example")
    ref = open(os.path.join(d, "default.tsv")).read().splitlines()[1:]
    assert len(ref) == 16      # 4 reads with a 2-way tie + 8 unique reads
    c1 = [r.split("\t") for r in ref if r.startswith("C_1\t")]
    assert [r[1:6] + [r[7]] for r in c1] == [["gi|7", "9913", "4225", "4225", "80", "2"],
                                              ["gi|4", "9646", "4225", "4225", "80", "2"]]
    uniq = [r.split("\t") for r in ref if r.startswith("1_1\t")]
    assert uniq == [["1_1", "gi|4", "9646", "4225", "0", "80", "80", "1"]]


@pytest.mark.skipif(not (O.have_ref() and os.path.isdir("/root/reference/example")),
                    reason="needs oracle/_ref and /root/reference")
def test_reference_reproduces_golden(tmp_path):
    """The committed truth really is what the compiled reference prints today."""
    d, cases = common.golden("example")
    out = O.ref_classify(os.path.join(d, "idx"), str(tmp_path / "o.tsv"), str(tmp_path / "r.tsv"),
                         u="/root/reference/example/reads/input.fa")
    assert out == open(os.path.join(d, "default.tsv")).read()


def test_seed_and_filters():
    orc = O.Oracle(os.path.join(common.golden("example")[0], "idx"))
    L = orc.L
    n = np.full(20, 4, dtype=np.uint8)
    assert L.cfo_mate_passes(n.ctypes.data, 20) == 0          # all N
    s = np.zeros(100, dtype=np.uint8); s[:15] = 4
    assert L.cfo_mate_passes(s.ctypes.data, 100) == 1         # 15 of 100 passes (0.15f ceiling)
    s[15] = 4
    assert L.cfo_mate_passes(s.ctypes.data, 100) == 0
    one = np.zeros(1, dtype=np.uint8)
    assert L.cfo_mate_passes(one.ctypes.data, 1) == 0         # length filter
