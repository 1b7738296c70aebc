"""The text forms of the batch ABI on the GPU (cf_batch_upload_text / cf_batch_wait_text: include/centrifuge_amd.h): a block of FASTA /
FASTQ text in, the default columns' text out — against the reference's golden TSVs and reports, against the word form of the same
reads, and at scale against oracle/_ref (tests/test_gpu_scale.py)."""
import os
import tempfile

import numpy as np
import pytest

import common
from centrifuge_amd import capi, reads
from test_async_abi import dev_index, load_case, tsv_of

pytestmark = pytest.mark.gpu
HEADER = reads.HEADER


def unpaired_cases():
    return [(a, n) for a, n in common.all_cases() if len([c for c in common.golden(a)[1] if c["name"] == n][0]["reads"]) == 1]


@pytest.mark.parametrize("arch,name", unpaired_cases())
def test_text_in_text_out_matches_the_reference(arch, name):
    d, c, kw, nm, ql, seq, off, seeds, paired = load_case(arch, name)
    if kw.get("k", 5) > 63:
        pytest.skip("the narrow result format holds -k <= 63")
    fastq = common.case_kwargs(c["args"])[1]
    text = open(os.path.join(d, c["reads"][0]), "rb").read()
    ix = dev_index(arch)
    clf = capi.Classifier(ix, **kw)
    clf.reset_counts()
    slot = capi.Slot(clf)
    slot.set_result_format(capi.RESULTS_NARROW)
    info = slot.submit_text(text, capi.TEXT_FASTQ if fastq else capi.TEXT_FASTA)
    want = open(os.path.join(d, c["tsv"])).read()
    if info.irregular:
        # the block holds a record outside the plain form: nothing was submitted, the slot is free for the host parser's reads
        b, m, ln = capi.pack_reads(seq, off)
        slot.set_result_format(capi.RESULTS_ROWS)
        slot.submit(b, m, ln, np.ascontiguousarray(seeds, dtype=np.uint32))
        assert tsv_of(ix, clf.params.khits, nm, ql, slot.wait()) == want
        slot.close(); clf.close()
        pytest.skip("not in the plain form (flags %#x): parsed on the host" % info.irregular)
    assert info.n_reads == len(nm) and info.n_bases == int(off[-1]) and info.max_len == (max(ql) if ql else 0)
    got, tuples, res = slot.wait_text()
    assert HEADER.encode() + got == want.encode(), common.first_diff((HEADER.encode() + got).decode("latin1"), want)
    # a second wait hands the same text back and tallies nothing twice
    again, tuples2, _ = slot.wait_text()
    assert again == got and np.array_equal(tuples, tuples2)
    # the report from what the device tallied alone: its counters, the perfect single assignments, the tuples
    rep = capi.Report(ix)
    rep.add_tuples(tuples)
    n_reads, n_unique = clf.counts()
    rep.adopt_device_tally(n_reads, n_unique, clf.counts_single())
    with tempfile.TemporaryDirectory() as t:
        p = os.path.join(t, "r.tsv")
        rep.write(p)
        assert open(p).read() == open(os.path.join(d, c["report"])).read()
    rep.close()
    # the same slot takes the word form next, and text again after it (the N-mask bookkeeping of the slot)
    b, m, ln = capi.pack_reads(seq, off)
    ni, nk = capi.sparse_nmask(m)
    slot.submit(b, None, ln, np.ascontiguousarray(seeds, dtype=np.uint32), nwords=(ni, nk))
    r16, qinfo, s2, _ = slot.wait_narrow()
    with pytest.raises(capi.CfError):
        slot.wait_text()                                        # reads that did not come as text have no readIDs on the device
    slot.submit(b, None, ln, np.ascontiguousarray(seeds, dtype=np.uint32), nwords=(ni, nk))
    slot.wait_narrow()
    assert not slot.submit_text(text, capi.TEXT_FASTQ if fastq else capi.TEXT_FASTA).irregular
    assert slot.wait_text()[0] == got
    # -u: only the block's first reads
    if len(nm) > 3:
        assert slot.submit_text(text, capi.TEXT_FASTQ if fastq else capi.TEXT_FASTA, max_reads=3).n_reads == 3
        head = slot.wait_text()[0]
        assert got.startswith(head) and head.count(b"\n") == sum(max(1, int(x)) for x in (qinfo[:3] & 0x3f))
    slot.close(); clf.close()


def paired_cases():
    return [(a, n) for a, n in common.all_cases() if len([c for c in common.golden(a)[1] if c["name"] == n][0]["reads"]) == 2]


@pytest.mark.parametrize("arch,name", paired_cases())
def test_mates_as_two_text_blocks_match_the_reference(arch, name):
    d, c, kw, nm, ql, seq, off, seeds, paired = load_case(arch, name)
    t1, t2 = (open(os.path.join(d, f), "rb").read() for f in c["reads"])
    ix = dev_index(arch)
    clf = capi.Classifier(ix, **kw)
    clf.reset_counts()
    slot = capi.Slot(clf)
    slot.set_result_format(capi.RESULTS_NARROW)
    info = slot.submit_text(t1, capi.TEXT_FASTA, text2=t2)
    assert not info.irregular and info.n_reads == 2 * len(nm)
    got, tuples, _ = slot.wait_text()
    want = open(os.path.join(d, c["tsv"])).read()
    assert HEADER.encode() + got == want.encode(), common.first_diff((HEADER.encode() + got).decode("latin1"), want)
    rep = capi.Report(ix)
    rep.add_tuples(tuples)
    n_reads, n_unique = clf.counts()
    rep.adopt_device_tally(n_reads, n_unique, clf.counts_single())
    with tempfile.TemporaryDirectory() as t:
        rep.write(os.path.join(t, "r.tsv"))
        assert open(os.path.join(t, "r.tsv")).read() == open(os.path.join(d, c["report"])).read()
    rep.close()
    # blocks that do not hold the same number of records are refused; -u counts pairs
    cut = t2.index(b"\n>", len(t2) // 2) + 1
    assert slot.submit_text(t1, capi.TEXT_FASTA, text2=t2[:cut]).irregular == 2048
    assert slot.submit_text(t1, capi.TEXT_FASTA, text2=t2, max_reads=3).n_reads == 6
    assert got.startswith(slot.wait_text()[0])
    slot.close(); clf.close()


def test_blocks_outside_the_plain_form_are_refused_and_cost_nothing():
    ix = dev_index("synth_small")
    clf = capi.Classifier(ix)
    slot = capi.Slot(clf)
    slot.set_result_format(capi.RESULTS_NARROW)
    for text, fmt in ((b">a\r\nACGT\r\n", capi.TEXT_FASTA), (b">\nACGT\n", capi.TEXT_FASTA), (b">a\nACRT\n", capi.TEXT_FASTA), (b"x>a\nACGT\n", capi.TEXT_FASTA),
                      (b"@a\nACGT\n+\nIII\n", capi.TEXT_FASTQ), (b"@a\nAC\nGT\n+\nIIII\n", capi.TEXT_FASTQ), (b"@a\nACGT\n+\nIIII", capi.TEXT_FASTQ)):
        info = slot.submit_text(text, fmt)
        assert info.irregular and info.n_reads == 0
        with pytest.raises(capi.CfError):
            slot.wait_text()
    # an empty block is a batch of no reads
    info = slot.submit_text(b"", capi.TEXT_FASTA)
    assert not info.irregular and info.n_reads == 0
    assert slot.wait_text()[0] == b""
    # wrong arguments
    with pytest.raises(capi.CfError):
        slot.submit_text(b">a\nACGT\n", 7)
    wide = capi.Slot(clf)
    assert not wide.submit_text(b">a\nACGT\n", capi.TEXT_FASTA).irregular
    with pytest.raises(capi.CfError):
        wide.wait_text()                                        # the rows must stay narrow on the device
    wide.close(); slot.close(); clf.close()
