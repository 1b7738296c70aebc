"""The host-side report (cf_report: counters, observed tuples, SQUAREM-EM abundance,
report TSV) against the reference's own report files, fed with the rows of the CPU
single-step harness.  Byte-exact, including the 6-digit abundance column."""
import os
import tempfile

import numpy as np
import pytest

import common
from centrifuge_amd import capi, reads
from emu import emu
from oracle import oracle as O


def max_scores(orc, seq, off, paired):
    """classifier.h:530-536 with the mate filters of centrifuge.cpp:2550-2577"""
    n = len(off) - 1
    lens = (off[1:] - off[:-1]).astype(np.int64)
    ok = np.array([bool(orc.L.cfo_mate_passes(seq[int(off[r]):].ctypes.data, int(lens[r]))) if lens[r] else False
                   for r in range(n)])
    perf = np.where(lens > 15, (lens - 15) ** 2, 0)
    def u32(v):                      # 2^32 or more (int64_t in the reference): the value no score reaches (kMaxScoreNever)
        return np.where(v >= 0xffffffff, 0xffffffff, v).astype(np.uint32)
    if not paired:
        return u32(np.where(ok, perf, 0))
    a, b = slice(0, n, 2), slice(1, n, 2)
    both = ok[a] & ok[b]
    return u32(np.where(both, perf[a] + perf[b], np.where(ok[a], perf[a], np.where(ok[b], perf[b], 0))))


@pytest.mark.parametrize("arch,name", common.all_cases())
def test_report_matches_reference(arch, name):
    d, cases = common.golden(arch)
    c = [x for x in cases if x["name"] == name][0]
    kw, fastq = common.case_kwargs(c["args"])
    base = os.path.join(d, "idx")
    e = emu.Emu(base)
    orc = O.Oracle(base)
    names, qlens, seq, off, seeds, paired = reads.load([os.path.join(d, f) for f in c["reads"]], fastq)
    rows, n_rows, score2 = e.classify(seq, off, seeds, paired=paired, **kw)
    ix = capi.Index(base, host_only=True)
    rep = capi.Report(ix)
    ms = max_scores(orc, seq, off, paired)
    # two chunks: the report accumulates across batches
    h = len(n_rows) // 2
    k = kw.get("k", 5)
    rep.add(rows[:h], n_rows[:h], ms[:h], k)
    rep.add(rows[h:], n_rows[h:], ms[h:], k)
    with tempfile.TemporaryDirectory() as t:
        out = os.path.join(t, "rep.tsv")
        rep.write(out)
        got = open(out).read()
    want = open(os.path.join(d, c["report"])).read()
    assert got == want, common.first_diff(got, want)
    rep.close(); ix.close()


def test_reset_between_inputs_keeps_the_observed_tuples():
    """--separator (centrifuge.cpp:3225): SpeciesMetrics::reset clears the counters only (aln_sink.h:84-91), so the
    second input's report has its own read counts and the abundance estimated from the tuples of both inputs."""
    d, cases = common.golden("synth_small")
    c = [x for x in cases if x["name"] == "k5"][0]
    base = os.path.join(d, "idx")
    e = emu.Emu(base)
    orc = O.Oracle(base)
    names, qlens, seq, off, seeds, paired = reads.load([os.path.join(d, f) for f in c["reads"]], False)
    rows, n_rows, score2 = e.classify(seq, off, seeds, paired=False)
    ms = max_scores(orc, seq, off, False)
    h = len(n_rows) // 3
    ix = capi.Index(base, host_only=True)

    def table(path):
        return {ln.split("\t")[1]: ln.rstrip("\n").split("\t") for ln in open(path).read().splitlines()[1:]}
    with tempfile.TemporaryDirectory() as t:
        both, second = capi.Report(ix), capi.Report(ix)
        both.add(rows, n_rows, ms, 5); both.write(os.path.join(t, "both.tsv"))
        second.add(rows[h:], n_rows[h:], ms[h:], 5); second.write(os.path.join(t, "second.tsv"))
        rep = capi.Report(ix)
        rep.add(rows[:h], n_rows[:h], ms[:h], 5); rep.write(os.path.join(t, "a.tsv"))
        rep.reset_counts()
        rep.add(rows[h:], n_rows[h:], ms[h:], 5); rep.write(os.path.join(t, "b.tsv"))
        tb, t2, t12 = table(os.path.join(t, "b.tsv")), table(os.path.join(t, "second.tsv")), table(os.path.join(t, "both.tsv"))
        assert set(tb) == set(t2)                                                  # rows: the taxa of the second input
        assert all(tb[k][:6] == t2[k][:6] for k in tb)                             # ... with its counts
        assert all(tb[k][6] == t12[k][6] for k in tb)                              # ... and the abundance of both inputs' tuples
        assert any(tb[k][6] != t2[k][6] for k in tb)
        for r in (both, second, rep):
            r.close()
    ix.close()


def test_adopted_counters_must_agree():
    """cf_report_adopt_counts — the self-check every centrifuge-class run ends with (cf_cli.cpp: the devices' counters, summed by
    RCCL, against the tally of the rows the output stage saw; SpeciesMetrics::merge aln_sink.h:109-140): counters that agree
    with the rows are taken, counters that are off by one read on one taxon are refused, and the report stays what it was"""
    d, cases = common.golden("synth_small")
    c = [x for x in cases if x["name"] == "k5"][0]
    base = os.path.join(d, "idx")
    e = emu.Emu(base)
    orc = O.Oracle(base)
    names, qlens, seq, off, seeds, paired = reads.load([os.path.join(d, f) for f in c["reads"]], False)
    rows, n_rows, score2 = e.classify(seq, off, seeds, paired=False)
    ms = max_scores(orc, seq, off, False)
    ix = capi.Index(base, host_only=True)
    taxa = [int(x) for x in ix.taxon_ids()]
    at = {t: i for i, t in enumerate(taxa)}
    n_reads, n_unique = np.zeros(len(taxa), dtype=np.uint64), np.zeros(len(taxa), dtype=np.uint64)
    for q in range(len(n_rows)):                                  # aln_sink.h:142-172 on the rows themselves
        for r in range(int(n_rows[q])):
            i = at[int(rows[q, r]["tax_id"])]
            n_reads[i] += 1
            if n_rows[q] == 1:
                n_unique[i] += 1
    if (n_rows == 0).any():
        n_reads[at[0]] += int((n_rows == 0).sum()); n_unique[at[0]] += int((n_rows == 0).sum())
    rep = capi.Report(ix)
    rep.add(rows, n_rows, ms, 5)
    with tempfile.TemporaryDirectory() as t:
        rep.adopt_counts(n_reads, n_unique)
        rep.write(os.path.join(t, "a.tsv"))
        assert open(os.path.join(t, "a.tsv")).read() == open(os.path.join(d, c["report"])).read()
        for arr in (n_reads, n_unique):
            bad = arr.copy()
            bad[int(np.flatnonzero(arr)[0])] += 1
            with pytest.raises(capi.CfError):
                rep.adopt_counts(bad if arr is n_reads else n_reads, bad if arr is n_unique else n_unique)
        with pytest.raises(capi.CfError):
            rep.adopt_counts(n_reads[:-1], n_unique[:-1])           # a counter array of another index
        rep.write(os.path.join(t, "b.tsv"))
        assert open(os.path.join(t, "b.tsv")).read() == open(os.path.join(d, c["report"])).read()
    rep.close(); ix.close()
