"""Parity of the HIP path (through the C ABI of libcentrifuge_amd.so) against the
golden vectors produced by the unmodified reference, and against the oracle on
the kernel taps.  Bit-exact: taxID / score / 2ndBest / hitLength / numMatches /
row order per read, and the per-taxon counters."""
import os

import numpy as np
import pytest

import common
from centrifuge_amd import capi, reads
from oracle import oracle as O

pytestmark = pytest.mark.gpu

_idx = {}


def dev_index(arch):
    if arch not in _idx:
        d, _ = common.golden(arch)
        _idx[arch] = capi.Index(os.path.join(d, "idx"), device=0)
    return _idx[arch]


@pytest.mark.parametrize("arch", ["example", "synth_small"])
def test_rank_tap_matches_oracle(arch):
    d, _ = common.golden(arch)
    ix = dev_index(arch)
    orc = O.Oracle(os.path.join(d, "idx"))
    n = ix.text_len
    rng = np.random.default_rng(1)
    rows = rng.integers(0, n + 1, size=20000, dtype=np.uint64)
    rows[:400] = np.arange(400) % (n + 1)
    chars = rng.integers(0, 4, size=len(rows), dtype=np.uint8)
    want = np.array([orc.L.cfo_rank(orc.h, int(c), int(r)) for c, r in zip(chars, rows)], dtype=np.uint64)
    assert np.array_equal(ix.debug_rank(chars, rows), want)
    assert np.array_equal(ix.debug_rank(chars, rows, single_lane=True), want)


@pytest.mark.parametrize("arch", ["example", "synth_small"])
def test_resolve_tap_matches_oracle(arch):
    d, _ = common.golden(arch)
    ix = dev_index(arch)
    orc = O.Oracle(os.path.join(d, "idx"))
    n = ix.text_len
    rows = np.arange(0, n + 1, dtype=np.uint64) if n < 5000 else \
        np.random.default_rng(2).integers(0, n + 1, size=20000, dtype=np.uint64)
    want = np.array([orc.L.cfo_resolve_row(orc.h, int(r)) for r in rows], dtype=np.uint32)
    assert np.array_equal(ix.debug_resolve(rows), want)


@pytest.mark.parametrize("arch", ["example", "synth_small"])
def test_resolve_table_by_position_equals_the_walks(arch):
    """round 6: where the resolve table holds every row and the text tables SA[row] for every row, the table is made from the TEXT
    POSITIONS of the stop rows of the walk-left ('$' row, the file's sample, boundary rows: bt2_idx.h:1980-2014) — table[row] = the
    value at the nearest stop at or left of SA[row] — instead of by a walk from every row.  Both builds of the same index: every row
    resolves alike (and as the oracle's walk does), and the longest walk — the bound the position form of hits rests on — is the same."""
    d, _ = common.golden(arch)
    orc = O.Oracle(os.path.join(d, "idx"))

    def open_with(by_pos):
        os.environ["CF_DENSE_SA_RATE"], os.environ["CF_TEXT_VERIFY_RATE"], os.environ["CF_DENSE_BY_POS"] = "0", "0", str(by_pos)
        try:
            return capi.Index(os.path.join(d, "idx"), device=0)
        finally:
            del os.environ["CF_DENSE_SA_RATE"], os.environ["CF_TEXT_VERIFY_RATE"], os.environ["CF_DENSE_BY_POS"]
    a, b, c = open_with(1), open_with(0), open_with(2)      # (2: made by position, then sent back to the walks as if the spot check had failed)
    try:
        assert a.L.cf_index_resolve_by_position(a.h) == 1 and b.L.cf_index_resolve_by_position(b.h) == 0 and c.L.cf_index_resolve_by_position(c.h) == 0
        assert a.L.cf_index_resolve_rate(a.h) == 0 and b.L.cf_index_resolve_rate(b.h) == 0
        n = a.text_len
        rows = np.arange(0, n + 1, dtype=np.uint64)
        ra, rb = a.debug_resolve(rows), b.debug_resolve(rows)
        assert np.array_equal(ra, rb)
        some = rows if n < 5000 else np.random.default_rng(5).choice(rows, size=20000, replace=False)
        want = np.array([orc.L.cfo_resolve_row(orc.h, int(r)) for r in some], dtype=np.uint32)
        assert np.array_equal(ra[some.astype(np.int64)], want)
        assert a.L.cf_index_walk_bound(a.h) == b.L.cf_index_walk_bound(b.h) == c.L.cf_index_walk_bound(c.h) > 0
        assert np.array_equal(c.debug_resolve(rows), rb)
    finally:
        a.close(); b.close(); c.close()


def test_search_tap_matches_oracle():
    d, cases = common.golden("synth_small")
    ix = dev_index("synth_small")
    orc = O.Oracle(os.path.join(d, "idx"))
    clf = capi.Classifier(ix)
    p = orc.params()
    recs = reads.read_fasta(os.path.join(d, "reads.fa"))
    import ctypes as C
    for name, codes, _ in recs[:150] + recs[-330:]:
        if len(codes) < 2:
            continue
        hf, hr = clf.debug_search(codes)
        of = (O.Hit * (len(codes) + 2))(); orr = (O.Hit * (len(codes) + 2))(); nh = (C.c_uint32 * 2)()
        # the tap only runs for reads that pass the N filter
        if not orc.L.cfo_mate_passes(codes.ctypes.data, len(codes)):
            assert len(hf) == 0 and len(hr) == 0
            continue
        orc.L.cfo_search(orc.h, C.byref(p), codes.ctypes.data, len(codes), C.addressof(of), C.addressof(orr), nh)
        for got, want, n in ((hf, of, nh[0]), (hr, orr, nh[1])):
            assert len(got) == n, name
            for i in range(n):
                assert (int(got[i]["top"]), int(got[i]["bot"]), int(got[i]["bwoff"]), int(got[i]["len"])) == \
                       (want[i].top, want[i].bot, want[i].bwoff, want[i].len), (name, i)
    clf.close()


@pytest.mark.parametrize("arch,name", common.all_cases())
def test_classify_matches_reference(arch, name):
    d, cases = common.golden(arch)
    c = [x for x in cases if x["name"] == name][0]
    kw, fastq = common.case_kwargs(c["args"])
    ix = dev_index(arch)
    clf = capi.Classifier(ix, **kw)
    names, qlens, seq, off, seeds, paired = reads.load([os.path.join(d, f) for f in c["reads"]], fastq)
    b = clf.batch(seq, off, seeds, paired)
    b.classify()
    rows, n_rows, score2 = b.results()
    got = reads.format_tsv(ix.seqid, names, qlens, rows, n_rows, score2)
    ref = open(os.path.join(d, c["tsv"])).read()
    assert got == ref, common.first_diff(got, ref)
    # counters (aln_sink.h:142-172) against the reference's report
    n_reads, n_uniq = clf.counts()
    taxa = ix.taxon_ids()
    mine = {int(t): (int(a), int(u)) for t, a, u in zip(taxa, n_reads, n_uniq) if t != 0 and a}
    rep = {}
    for ln in open(os.path.join(d, c["report"])).read().splitlines()[1:]:
        f = ln.split("\t")
        rep[int(f[1])] = (int(f[4]), int(f[5]))
    assert mine == rep
    # idempotence: a second pass over the same resident batch — plan and strand records made again from the
    # resident reads (cf_batch_plan), then the four stages — gives the same rows
    assert b.plan() >= 0.0
    b.classify()
    rows2, n_rows2, score22 = b.results()
    assert np.array_equal(n_rows, n_rows2) and np.array_equal(score2, score22)
    for q in range(len(n_rows)):
        assert np.array_equal(rows[q, :n_rows[q]], rows2[q, :n_rows2[q]])
    # the packed egress (cf_batch_results_compact) carries exactly the used slots of the k-slot layout
    crow, first, cn, cs2 = b.results_compact()
    assert np.array_equal(cn, n_rows) and np.array_equal(cs2, score2) and len(crow) == int(n_rows.sum())
    for q in range(len(n_rows)):
        assert np.array_equal(crow[int(first[q]):int(first[q]) + int(cn[q])], rows[q, :n_rows[q]])
    b.close(); clf.close()


def test_replicated_batch_is_order_independent():
    """Size-independent property: classifying a batch made of R shuffled copies of
    the golden reads gives every copy the rows of the original (work-queue order,
    wave packing and batch size must not matter)."""
    d, cases = common.golden("synth_small")
    ix = dev_index("synth_small")
    clf = capi.Classifier(ix)
    names, qlens, seq, off, seeds, paired = reads.load([os.path.join(d, "reads.fa")], False)
    n = len(names)
    b0 = clf.batch(seq, off, seeds, False); b0.classify(); r0, n0, s0 = b0.results(); b0.close()
    R = 40
    rng = np.random.default_rng(3)
    perm = np.concatenate([rng.permutation(n) for _ in range(R)])
    lens = (off[1:] - off[:-1]).astype(np.int64)
    parts = [seq[int(off[i]):int(off[i + 1])] for i in perm]
    off2 = np.zeros(len(perm) + 1, dtype=np.uint64); off2[1:] = np.cumsum(lens[perm])
    seq2 = np.concatenate(parts)
    b = clf.batch(seq2, off2, seeds[perm], False); b.classify(); r, nr, s2 = b.results(); b.close()
    assert np.array_equal(nr, n0[perm]) and np.array_equal(s2, s0[perm])
    for j, i in enumerate(perm):
        assert np.array_equal(r[j, :nr[j]], r0[i, :n0[i]])
    clf.close()


def test_counts_allreduce_over_rccl_single_rank():
    """cf_counts_allreduce drives ncclAllReduce on the device counters; with a one-rank
    communicator (all a 1-GPU box offers) the sum must leave them unchanged."""
    import ctypes as C
    try:
        rccl = C.CDLL("librccl.so.1")
    except OSError:
        pytest.skip("librccl not found")

    class UniqueId(C.Structure):
        _fields_ = [("internal", C.c_char * 128)]

    uid = UniqueId()
    assert rccl.ncclGetUniqueId(C.byref(uid)) == 0
    comm = C.c_void_p()
    rccl.ncclCommInitRank.argtypes = [C.POINTER(C.c_void_p), C.c_int, UniqueId, C.c_int]
    assert rccl.ncclCommInitRank(C.byref(comm), 1, uid, 0) == 0
    d, _ = common.golden("example")
    ix = dev_index("example")
    clf = capi.Classifier(ix)
    names, qlens, seq, off, seeds, paired = reads.load([os.path.join(d, "reads.fa")], False)
    b = clf.batch(seq, off, seeds, paired)
    b.classify()
    before = clf.counts()
    clf.allreduce_counts(comm)
    import torch
    torch.cuda.synchronize()
    after = clf.counts()
    assert np.array_equal(before[0], after[0]) and np.array_equal(before[1], after[1]) and before[0].sum() > 0
    rccl.ncclCommDestroy.argtypes = [C.c_void_p]
    rccl.ncclCommDestroy(comm)
    b.close(); clf.close()


@pytest.mark.parametrize("lengths,paired,k", common.EDGE_CASES)
def test_edge_batches_match_oracle(lengths, paired, k):
    """read lengths at the boundaries of the ftab window, the minimum hit length, the LDS strand
    words and the two strand-record sizes; empty, all-N, homopolymer and N-run reads; pairs with a
    filtered mate — rows, order, 2ndBest must equal the oracle's"""
    d, _ = common.golden("synth_small")
    ix = dev_index("synth_small")
    orc = O.Oracle(os.path.join(d, "idx"))
    recs = reads.read_fasta(os.path.join(d, "reads.fa")) + reads.read_fasta(os.path.join(d, "reads250.fa"))
    rng = np.random.default_rng(11)
    rs = common.edge_reads(recs, lengths, rng)
    if paired and len(rs) % 2:
        rs.append(rs[0])
    seq, off = orc.pack(rs)
    seeds = rng.integers(0, 2 ** 32, size=len(rs), dtype=np.uint32)
    nq = len(rs) // 2 if paired else len(rs)
    want = orc.classify(seq, off, seeds, nq, paired, orc.params(k=k))
    clf = capi.Classifier(ix, k=k)
    b = clf.batch(seq, off, seeds, paired)
    b.classify()
    got = b.results()
    assert np.array_equal(got[1], want[1]) and np.array_equal(got[2], want[2])
    for q in range(nq):
        for r in range(int(want[1][q])):
            g, w = got[0][q, r], want[0][q, r]
            assert (int(g["tax_id"]), int(g["unique_id"]), int(g["score"]), int(g["hit_len"])) == \
                   (int(w["tax_id"]), int(w["unique_id"]), int(w["score"]), int(w["hit_len"])), (q, r)
    b.close(); clf.close()


def test_empty_batch():
    ix = dev_index("example")
    clf = capi.Classifier(ix)
    b = clf.batch(np.zeros(1, dtype=np.uint8), np.zeros(1, dtype=np.uint64), np.zeros(0, dtype=np.uint32), False)
    b.classify()
    rows, n_rows, s2 = b.results()
    assert len(n_rows) == 0
    crow, first, cn, cs2 = b.results_compact()
    assert len(crow) == 0 and len(cn) == 0 and len(b.max_scores()) == 0
    b.close(); clf.close()


@pytest.mark.skipif(not O.have_ref(), reason="oracle/_ref (the compiled reference) is not built")
@pytest.mark.parametrize("k", [5, 1])
def test_contigs_and_long_reads_match_the_reference(tmp_path, k):
    """reads of 65,535 to 300,000 bases (rounds 1-3 refused them): the byte-window search kernel, 24-bit offsets and lengths in
    the hit records, the general post / score kernels — through the slot ABI and through centrifuge-class, rows and report
    against the compiled reference (classifier.h:212-571 has no length bound)"""
    import subprocess
    d = str(tmp_path)
    base, fa = common.long_read_case(d)
    want = O.ref_classify(base, os.path.join(d, "w.tsv"), os.path.join(d, "w.rep"), u=fa, extra=["-k", str(k)], threads=4)
    names, ql, seq, off, seeds, pr = reads.load([fa], False)
    ix = capi.Index(base, device=0)
    clf = capi.Classifier(ix, k=k)
    b, m, ln = capi.pack_reads(seq, off)
    slot = capi.Slot(clf)
    slot.submit(b, m, ln, np.ascontiguousarray(seeds, dtype=np.uint32), paired=False)
    rows, first, n_rows, score2, max_score, info = slot.wait()
    slot.close()
    got = reads.format_tsv(ix.seqid, names, ql, capi.unpack_rows(rows, first, n_rows, k), n_rows, score2)
    assert got == want, common.first_diff(got, want)
    assert int(max_score[5]) == 0xffffffff and int(max_score[6]) == (65535 - 15) ** 2     # 300 kb: beyond 32 bits, "never reached"
    # the byte form of the boundary (cf_batch_create) takes them as well
    bt = clf.batch(seq, off, seeds, False)
    bt.classify()
    r2, n2, s2 = bt.results()
    bt.close()
    assert reads.format_tsv(ix.seqid, names, ql, r2, n2, s2) == want
    clf.close(); ix.close()
    cli = os.path.join(common.ROOT, "centrifuge_amd", "bin", "centrifuge-class")
    out, rep = os.path.join(d, "o.tsv"), os.path.join(d, "o.rep")
    r = subprocess.run([cli, "-f", "-k", str(k), "-x", base, "-U", fa, "-S", out, "--report-file", rep], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert open(out).read() == want
    assert open(rep).read() == open(os.path.join(d, "w.rep")).read()


def test_reads_beyond_the_hit_records_are_refused():
    d, _ = common.golden("example")
    ix = dev_index("example")
    clf = capi.Classifier(ix)
    n = (1 << 24) - 1
    seq = np.zeros(n, dtype=np.uint8)
    with pytest.raises(capi.CfError):
        clf.batch(seq, np.array([0, n], dtype=np.uint64), np.zeros(1, dtype=np.uint32), False)
    clf.close()
