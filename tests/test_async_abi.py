"""The asynchronous slot ABI (cf_batch_alloc / cf_batch_submit / cf_batch_wait, include/centrifuge_amd.h): packed
2-bit input across the boundary, no host round trip inside a batch, reusable slots, several batches in flight, and the
paths cf_batch_wait finishes on its own (a hit pool that was too small; more planned rows than the row workspace
holds).  Every result is compared with the reference's golden TSV / with the one-shot path."""
import os

import numpy as np
import pytest

import common
from centrifuge_amd import capi, reads


def naive_pack(seq, off):
    nw = int(sum((int(off[i + 1] - off[i]) + 31) // 32 for i in range(len(off) - 1)))
    bases, nmask = np.zeros(nw, dtype=np.uint64), np.zeros(nw, dtype=np.uint32)
    w = 0
    for r in range(len(off) - 1):
        s = seq[int(off[r]):int(off[r + 1])]
        for i, c in enumerate(s):
            if c > 3:
                nmask[w + i // 32] |= np.uint32(1 << (i % 32))
            else:
                bases[w + i // 32] |= np.uint64(int(c) << (2 * (i % 32)))
        w += (len(s) + 31) // 32
    return bases, nmask


def test_pack_reads_layout():
    rng = np.random.default_rng(5)
    rs = [rng.integers(0, 5, int(L), dtype=np.uint8) for L in [0, 1, 31, 32, 33, 64, 100, 0, 257, 5]]
    off = np.zeros(len(rs) + 1, dtype=np.uint64)
    off[1:] = np.cumsum([len(r) for r in rs])
    seq = np.concatenate(rs)
    b, m, ln = capi.pack_reads(seq, off)
    wb, wm = naive_pack(seq, off)
    assert np.array_equal(b, wb) and np.array_equal(m, wm)
    assert list(ln) == [len(r) for r in rs]
    b0, m0, l0 = capi.pack_reads(np.zeros(0, dtype=np.uint8), np.zeros(1, dtype=np.uint64))
    assert len(b0) == 0 and len(m0) == 0 and len(l0) == 0


_idx = {}


def dev_index(arch):
    if arch not in _idx:
        d, _ = common.golden(arch)
        _idx[arch] = capi.Index(os.path.join(d, "idx"), device=0)
    return _idx[arch]


def load_case(arch, name):
    d, cases = common.golden(arch)
    c = [x for x in cases if x["name"] == name][0]
    kw, fastq = common.case_kwargs(c["args"])
    nm, ql, seq, off, seeds, paired = reads.load([os.path.join(d, f) for f in c["reads"]], fastq)
    return d, c, kw, nm, ql, seq, off, seeds, paired


def tsv_of(ix, k, nm, ql, res):
    rows, first, n_rows, score2, max_score, info = res
    return reads.format_tsv(ix.seqid, nm, ql, capi.unpack_rows(rows, first, n_rows, k), n_rows, score2)


@pytest.mark.gpu
@pytest.mark.parametrize("arch,name", common.all_cases())
def test_packed_submit_matches_reference(arch, name):
    d, c, kw, nm, ql, seq, off, seeds, paired = load_case(arch, name)
    ix = dev_index(arch)
    clf = capi.Classifier(ix, **kw)
    b, m, ln = capi.pack_reads(seq, off)
    slot = capi.Slot(clf)
    slot.submit(b, m, ln, np.ascontiguousarray(seeds, dtype=np.uint32), paired=paired)
    res = slot.wait()
    got = tsv_of(ix, clf.params.khits, nm, ql, res)
    want = open(os.path.join(d, c["tsv"])).read()
    assert got == want, common.first_diff(got, want)
    assert res[5]["row_passes"] >= 1
    # the N mask in its sparse form (only the words that hold an N; the slot zeroes the rest) gives the same rows
    ni, nk = capi.sparse_nmask(m)
    assert len(ni) == int(np.count_nonzero(m)) and (len(ni) == 0 or len(ni) < len(m))
    slot.submit(b, None, ln, np.ascontiguousarray(seeds, dtype=np.uint32), paired=paired, nwords=(ni, nk))
    res_s = slot.wait()
    assert tsv_of(ix, clf.params.khits, nm, ql, res_s) == want
    # max_score as the one-shot path computes it
    bt = clf.batch(seq, off, seeds, paired)
    bt.classify()
    assert np.array_equal(bt.max_scores(), res[4])
    r2 = bt.results_compact()
    assert np.array_equal(r2[0], res[0]) and np.array_equal(r2[2], res[2]) and np.array_equal(r2[3], res[3])
    bt.close(); slot.close(); clf.close()


@pytest.mark.gpu
def test_slot_reuse_and_batches_in_flight():
    """one slot takes batches of different size and shape one after the other; three slots run concurrently on three
    streams with pinned input; counters accumulate over everything exactly once"""
    import torch
    ix = dev_index("synth_small")
    clf = capi.Classifier(ix)
    cases = [load_case("synth_small", n) for n in ("k5", "r250_k5", "pe_k5", "fastq")]
    wants = [open(os.path.join(x[0], x[1]["tsv"])).read() for x in cases]
    slot = capi.Slot(clf, max_reads=64, max_words=256)
    for rep in range(2):
        for x, want in zip(cases, wants):
            d, c, kw, nm, ql, seq, off, seeds, paired = x
            b, m, ln = capi.pack_reads(seq, off)
            slot.submit(b, m, ln, np.ascontiguousarray(seeds, dtype=np.uint32), paired=paired)
            assert tsv_of(ix, 5, nm, ql, slot.wait()) == want
    slot.close()
    before = clf.counts()
    streams = [torch.cuda.Stream() for _ in range(3)]
    slots = [capi.Slot(clf) for _ in range(3)]
    pinned = []
    for rnd in range(3):
        for i in range(3):
            d, c, kw, nm, ql, seq, off, seeds, paired = cases[(i + rnd) % 3]
            b, m, ln = capi.pack_reads(seq, off)
            arrs = []
            for a in (b, m, ln, np.ascontiguousarray(seeds, dtype=np.uint32)):
                p = capi.PinnedArray(clf.L, a.dtype, len(a))
                p.a[:] = a
                arrs.append(p)
            pinned.append(arrs)
            slots[i].submit(arrs[0].a, arrs[1].a, arrs[2].a, arrs[3].a, paired=paired, stream=streams[i].cuda_stream)
        for i in range(3):
            d, c, kw, nm, ql, seq, off, seeds, paired = cases[(i + rnd) % 3]
            assert tsv_of(ix, 5, nm, ql, slots[i].wait(copy=False)) == wants[(i + rnd) % 3]
    # the three stages of every slot on three shared streams (upload / kernels / download), chained by events
    st3 = tuple(x.cuda_stream for x in streams)
    for rnd in range(4):
        for i in range(3):
            arrs = pinned[(rnd % 3) * 3 + i]
            slots[i].submit(arrs[0].a, arrs[1].a, arrs[2].a, arrs[3].a, paired=cases[(i + rnd % 3) % 3][8], streams=st3)
        for i in range(3):
            x = cases[(i + rnd % 3) % 3]
            assert tsv_of(ix, 5, x[3], x[4], slots[i].wait(copy=False)) == wants[(i + rnd % 3) % 3]
    after = clf.counts()
    one = []
    for x in cases[:3]:
        c0 = capi.Classifier(ix)
        bt = c0.batch(x[5], x[6], x[7], x[8])
        bt.classify()
        one.append(c0.counts())
        bt.close(); c0.close()
    for j in range(2):
        assert np.array_equal(after[j] - before[j], 7 * sum(o[j] for o in one))
    for s in slots:
        s.close()
    for arrs in pinned:
        for p in arrs:
            p.free()
    clf.close()


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["k5", "pe_k1", "r250_k5", "k50"])
@pytest.mark.parametrize("rows_per_pass", [1, 3, 50])
def test_row_workspace_smaller_than_the_batch(name, rows_per_pass):
    """more planned rows than the row workspace holds: cf_batch_wait finishes the batch in further passes (and grows
    the workspace to a single query that exceeds it); rows and counters are those of the one-pass run"""
    d, c, kw, nm, ql, seq, off, seeds, paired = load_case("synth_small", name)
    ix = dev_index("synth_small")
    clf = capi.Classifier(ix, **kw)
    b, m, ln = capi.pack_reads(seq, off)
    slot = capi.Slot(clf)
    slot.set_limits(rows_per_pass=rows_per_pass)
    slot.submit(b, m, ln, np.ascontiguousarray(seeds, dtype=np.uint32), paired=paired)
    res = slot.wait()
    got = tsv_of(ix, clf.params.khits, nm, ql, res)
    want = open(os.path.join(d, c["tsv"])).read()
    assert got == want, common.first_diff(got, want)
    assert res[5]["row_passes"] > 1 and res[5]["planned_sa_rows"] > rows_per_pass
    cnt = clf.counts()
    c0 = capi.Classifier(ix, **kw)
    bt = c0.batch(seq, off, seeds, paired)
    bt.classify()
    ref = c0.counts()
    assert np.array_equal(cnt[0], ref[0]) and np.array_equal(cnt[1], ref[1])
    bt.close(); c0.close(); slot.close(); clf.close()


@pytest.mark.gpu
def test_hit_pool_smaller_than_the_batch():
    """a pool that cannot hold the batch's hit lists: nothing is searched in the first attempt, cf_batch_wait grows the
    pool to what the plan asked for and runs the batch again — once, with every read counted once"""
    d, c, kw, nm, ql, seq, off, seeds, paired = load_case("synth_small", "k5")
    ix = dev_index("synth_small")
    clf = capi.Classifier(ix)
    b, m, ln = capi.pack_reads(seq, off)
    slot = capi.Slot(clf)
    slot.set_limits(hit_slots=100)
    slot.submit(b, m, ln, np.ascontiguousarray(seeds, dtype=np.uint32), paired=paired)
    got = tsv_of(ix, 5, nm, ql, slot.wait())
    assert got == open(os.path.join(d, c["tsv"])).read()
    assert int(clf.counts()[0].sum()) >= len(nm)
    c0 = capi.Classifier(ix)
    bt = c0.batch(seq, off, seeds, paired)
    bt.classify()
    assert np.array_equal(c0.counts()[0], clf.counts()[0])
    bt.close(); c0.close(); slot.close(); clf.close()


@pytest.mark.gpu
def test_n_rich_reads_outgrow_the_estimated_pool():
    """the pool is sized for N-free reads: reads at the 15 % N ceiling need more slots and take the re-run path unaided"""
    from oracle import oracle as O
    d, _ = common.golden("synth_small")
    ix = dev_index("synth_small")
    orc = O.Oracle(os.path.join(d, "idx"))
    recs = reads.read_fasta(os.path.join(d, "reads.fa"))
    rng = np.random.default_rng(8)
    rs = []
    for i in range(400):
        r = recs[i % len(recs)][1].copy()
        pos = rng.choice(len(r), int(0.15 * len(r)), replace=False)
        r[pos] = 4
        rs.append(r)
    seq, off = orc.pack(rs)
    seeds = rng.integers(0, 2 ** 32, size=len(rs), dtype=np.uint32)
    want = orc.classify(seq, off, seeds, len(rs), False, orc.params())
    clf = capi.Classifier(ix)
    b, m, ln = capi.pack_reads(seq, off)
    slot = capi.Slot(clf)
    slot.submit(b, m, ln, seeds)
    rows, first, n_rows, score2, max_score, info = slot.wait()
    assert np.array_equal(n_rows, want[1]) and np.array_equal(score2, want[2])
    got = capi.unpack_rows(rows, first, n_rows, 5)
    for q in range(len(rs)):
        for r in range(int(n_rows[q])):
            assert (int(got[q, r]["tax_id"]), int(got[q, r]["score"]), int(got[q, r]["hit_len"])) == \
                   (int(want[0][q, r]["tax_id"]), int(want[0][q, r]["score"]), int(want[0][q, r]["hit_len"]))
    slot.close(); clf.close()


@pytest.mark.gpu
def test_wrong_max_len_and_misuse_are_errors():
    d, c, kw, nm, ql, seq, off, seeds, paired = load_case("synth_small", "k5")
    ix = dev_index("synth_small")
    clf = capi.Classifier(ix)
    b, m, ln = capi.pack_reads(seq, off)
    slot = capi.Slot(clf)
    with pytest.raises(capi.CfError):
        slot.wait()                                            # nothing in flight
    slot.submit(b, m, ln, np.ascontiguousarray(seeds, dtype=np.uint32), max_len=int(ln.max()) - 1)
    with pytest.raises(capi.CfError, match="max_len"):
        slot.wait()
    # fewer packed words than the lengths imply: flagged by the plan before anything reads past the upload
    sd = np.ascontiguousarray(seeds, dtype=np.uint32)
    slot.submit(b[:-1], m[:-1], ln, sd)
    with pytest.raises(capi.CfError, match="n_words"):
        slot.wait()
    slot.submit(b, m, ln, sd)      # the slot is usable again
    assert tsv_of(ix, 5, nm, ql, slot.wait()) == open(os.path.join(d, c["tsv"])).read()
    slot.close(); clf.close()


@pytest.mark.gpu
@pytest.mark.parametrize("n", [0, 1, 15, 16, 17, 4095, 4096, 4097, 65536, 1048576 + 3, 3 * 4096 * 256 + 5])
def test_batch_prefix_sums(n):
    """cf_scan.hpp (reduce-then-scan over 4096-item tiles): n items -> n + 1 exclusive sums, all three summand maps"""
    L = capi.lib()
    rng = np.random.default_rng(n)
    x = rng.integers(0, 300, size=n, dtype=np.uint32)
    x[rng.random(n) < 0.3] = 0
    if n > 5:
        x[3] = 0xffffffff                                     # sums need 64 bits
    for mode, val in ((0, (x.astype(np.uint64) + 31) >> 5), (1, x.astype(np.uint64)), (2, 2 * x.astype(np.uint64))):
        sums = np.full(n + 1, 7, dtype=np.uint64)
        cnt = np.full(n + 1, 7, dtype=np.uint32)
        st = L.cf_debug_scan(0, mode, x.ctypes.data, n, sums.ctypes.data, cnt.ctypes.data)
        assert st == 0
        want = np.zeros(n + 1, dtype=np.uint64)
        np.cumsum(val, out=want[1:])
        assert np.array_equal(sums, want)
        if mode == 2:
            wc = np.zeros(n + 1, dtype=np.uint32)
            np.cumsum(x != 0, out=wc[1:])
            assert np.array_equal(cnt, wc)


@pytest.mark.gpu
@pytest.mark.parametrize("wide,dense,tv,planes", [(11, 0, 2, 1), (12, 1, 0, 0), (13, 3, -1, 1), (0, 4, 3, 0), (0, 4, 1, 1), (0, 4, -1, 1)])
def test_derived_index_tables_do_not_change_a_row(wide, dense, tv, planes):
    """the wide ftab (CF_WIDE_FTAB bases per entry), the dense resolve table (every 2^CF_DENSE_SA_RATE-th row), the text
    verification tables (CF_TEXT_VERIFY_RATE) and the occurrence planes (CF_OCC_PLANES: the search kernel's one-chain-per-lane
    form; 0 = its two-lanes-per-chain form over the sides) are made from the index on the device when it is opened; the knobs
    force combinations of them here.  Every golden case of the synthetic index, the search / resolve taps against the oracle,
    and fewer LF steps than without them."""
    from oracle import oracle as O
    d, cases = common.golden("synth_small")
    def open_with(w, r, t, p):
        os.environ["CF_WIDE_FTAB"], os.environ["CF_DENSE_SA_RATE"], os.environ["CF_TEXT_VERIFY_RATE"], os.environ["CF_OCC_PLANES"] = str(w), str(r), str(t), str(p)
        try:
            return capi.Index(os.path.join(d, "idx"), device=0)
        finally:
            del os.environ["CF_WIDE_FTAB"], os.environ["CF_DENSE_SA_RATE"], os.environ["CF_TEXT_VERIFY_RATE"], os.environ["CF_OCC_PLANES"]
    ix = open_with(wide, dense, tv, planes)
    assert ix.L.cf_index_wide_ftab_chars(ix.h) == wide and ix.L.cf_index_resolve_rate(ix.h) == dense and ix.L.cf_index_text_verify_rate(ix.h) == tv
    assert ix.L.cf_index_occ_planes(ix.h) == planes
    plain = open_with(0, 4, -1, 0)                            # the file's own tables only
    # the search tap (hit lists after extend / twin / trim) read by read, with and without the tables
    clf_t, clf_p = capi.Classifier(ix), capi.Classifier(plain)
    for rec in reads.read_fasta(os.path.join(d, "reads.fa"))[::23] + reads.read_fasta(os.path.join(d, "reads250.fa"))[::11]:
        a, b = clf_t.debug_search(rec[1]), clf_p.debug_search(rec[1])
        assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
    clf_t.close(); clf_p.close()
    orc = O.Oracle(os.path.join(d, "idx"))
    rows_t = np.random.default_rng(2).integers(0, ix.text_len + 1, size=20000, dtype=np.uint64)
    want = np.array([orc.L.cfo_resolve_row(orc.h, int(r)) for r in rows_t], dtype=np.uint32)
    assert np.array_equal(ix.debug_resolve(rows_t), want)
    for c in cases:
        kw, fastq = common.case_kwargs(c["args"])
        nm, ql, seq, off, seeds, paired = reads.load([os.path.join(d, f) for f in c["reads"]], fastq)
        clf = capi.Classifier(ix, **kw)
        bt = clf.batch(seq, off, seeds, paired)
        bt.classify()
        rows, n_rows, s2 = bt.results()
        got = reads.format_tsv(ix.seqid, nm, ql, rows, n_rows, s2)
        assert got == open(os.path.join(d, c["tsv"])).read(), c["name"]
        if c["name"] == "k5":
            ops = bt.opcounts()
            c0 = capi.Classifier(plain)
            b0 = c0.batch(seq, off, seeds, paired)
            b0.classify()
            o0 = b0.opcounts()
            assert ops.n_rows == o0.n_rows
            if wide:
                assert ops.n_ftab_wide > 0 and ops.n_pair + ops.n_single < o0.n_pair + o0.n_single
            if dense < 4:
                assert ops.n_walk < o0.n_walk
            if tv >= 0:
                assert ops.n_verify > 0 and ops.n_single < o0.n_single
            b0.close(); c0.close()
        bt.close(); clf.close()
    ix.close(); plain.close()


@pytest.mark.gpu
def test_index_budget_chooses_tables_and_changes_no_result():
    """cf_index_open_ex: the derived tables are made only while the caller's HBM budget lasts (and individually on request);
    cf_index_describe says what was made; the rows never change.  (The golden index is tiny — floor(log4 n) < 10 — so the wide
    ftab is asked for by hand, 12 bases: 134 MB, which also gives the budgets something to refuse.)"""
    d, c, kw, nm, ql, seq, off, seeds, paired = load_case("synth_small", "k5")
    base = os.path.join(d, "idx")
    want = open(os.path.join(d, c["tsv"])).read()
    b, m, ln = capi.pack_reads(seq, off)
    sd = np.ascontiguousarray(seeds, dtype=np.uint32)
    ixf = capi.Index(base, device=0, wide_ftab_chars=12)
    full = ixf.describe()
    ixf.close()
    assert full["total_bytes"] == full["file_section_bytes"] + full["wide_ftab_bytes"] + full["text_bytes"] + full["planes_bytes"] + \
        full["pair_planes_bytes"] + full["resolve_bytes"] - full["file_bytes_dropped"], full
    assert 0 < full["file_bytes_dropped"] < full["file_section_bytes"] and full["sides_dropped"] == 0      # the SA sample, behind the resolve table at every row
    assert full["wide_ftab_chars"] == 12 and full["occ_planes"] == 1 and full["pair_planes"] == 1 and full["text_verify_rate"] >= 0 and full["resolve_rate"] == 0, full
    seen = set()
    for kwargs in (dict(hbm_budget=full["file_section_bytes"] + 4096),                  # nothing fits beside the files
                   dict(hbm_budget=full["file_section_bytes"] + 64 * 1024 * 1024, wide_ftab_chars=12),      # the small tables, not 134 MB of wide ftab
                   dict(hbm_budget=full["total_bytes"] * 8, wide_ftab_chars=12),
                   dict(occ_planes=-1), dict(pair_planes=-1), dict(wide_ftab_chars=-1, text_verify_rate=-1), dict(resolve_rate=-1), dict(resolve_rate=3, text_verify_rate=3)):
        ix = capi.Index(base, device=0, **kwargs)
        cfg = ix.describe()
        if "hbm_budget" in kwargs:
            assert cfg["total_bytes"] <= kwargs["hbm_budget"] and cfg["budget_bytes"] <= kwargs["hbm_budget"], (kwargs, cfg)
        if kwargs.get("hbm_budget") == full["file_section_bytes"] + 4096:
            assert cfg["total_bytes"] == cfg["file_section_bytes"], cfg
        if kwargs.get("hbm_budget") == full["total_bytes"] * 8:
            assert cfg["wide_ftab_chars"] == 12 and cfg["pair_planes"] == 1, cfg
        if kwargs.get("occ_planes") == -1:
            assert cfg["occ_planes"] == 0 and cfg["planes_bytes"] == 0 and cfg["pair_planes"] == 0, cfg
        if kwargs.get("pair_planes") == -1:
            assert cfg["occ_planes"] == 1 and cfg["pair_planes"] == 0 and cfg["pair_planes_bytes"] == 0, cfg
        if kwargs.get("wide_ftab_chars") == -1:
            assert cfg["wide_ftab_chars"] == 0 and cfg["text_verify_rate"] == -1, cfg
        if kwargs.get("resolve_rate") == -1:
            assert cfg["resolve_bytes"] == 0 and cfg["resolve_rate"] == 4, cfg
        if kwargs.get("resolve_rate") == 3:
            assert cfg["resolve_rate"] == 2 and cfg["text_verify_rate"] == 3, cfg
        assert cfg["est_requests_per_100bp_read"] > 0
        seen.add((cfg["wide_ftab_chars"], cfg["text_verify_rate"], cfg["occ_planes"], cfg["pair_planes"], cfg["resolve_rate"]))
        clf = capi.Classifier(ix)
        slot = capi.Slot(clf)
        slot.submit(b, m, ln, sd)
        assert tsv_of(ix, 5, nm, ql, slot.wait()) == want, kwargs
        slot.close(); clf.close(); ix.close()
    assert len(seen) >= 5, seen
    with pytest.raises(capi.CfError):
        capi.Index(base, device=0, hbm_budget=1024)                                   # not even the files fit


# ---- round 4: resident reads again, slot sizes, an index without its sides
@pytest.mark.gpu
@pytest.mark.parametrize("arch,name", [("synth_small", "k5"), ("synth_small", "pe_k1"), ("synth_small", "r250_k5"), ("example", "default")])
def test_reclassify_on_resident_reads_gives_the_same_rows(arch, name):
    """cf_batch_reclassify_async: plan + kernels once more over the reads the slot holds since its last upload (what bench.py
    times as `value`: inputs resident in HBM) — same rows every time, counters once per run"""
    d, c, kw, nm, ql, seq, off, seeds, paired = load_case(arch, name)
    ix = dev_index(arch)
    clf = capi.Classifier(ix, **kw)
    b, m, ln = capi.pack_reads(seq, off)
    slot = capi.Slot(clf)
    slot.submit(b, m, ln, np.ascontiguousarray(seeds, dtype=np.uint32), paired=paired)
    want = open(os.path.join(d, c["tsv"])).read()
    assert tsv_of(ix, kw.get("k", 5), nm, ql, slot.wait()) == want
    once = clf.counts()
    for rep in (2, 3):
        slot.resubmit((None, None))
        assert tsv_of(ix, kw.get("k", 5), nm, ql, slot.wait()) == want
        now = clf.counts()
        assert np.array_equal(now[0], rep * once[0]) and np.array_equal(now[1], rep * once[1])
    slot.close(); clf.close()


@pytest.mark.gpu
def test_slot_estimate_matches_what_a_slot_takes():
    """cf_slot_estimate_bytes (no device) against the device memory a slot of that size really takes"""
    import torch
    ix = dev_index("synth_small")
    clf = capi.Classifier(ix)
    torch.cuda.synchronize()
    for reads_, words in ((200000, 800000), (1000000, 5000000)):
        est = capi.slot_bytes(reads_, words, occ_planes=bool(ix.L.cf_index_occ_planes(ix.h)))
        before = torch.cuda.mem_get_info(0)[0]
        slot = capi.Slot(clf, reads_, words)
        after = torch.cuda.mem_get_info(0)[0]
        slot.close()
        took = before - after
        assert 0.9 * est <= took <= 1.1 * est + (64 << 20), (reads_, est, took)      # (allocation granules of the runtime on top)
    clf.close()


@pytest.mark.gpu
@pytest.mark.parametrize("arch,name", [("synth_small", "k5"), ("synth_small", "pe_k1"), ("synth_small", "r250_k5"), ("synth_small", "host")])
def test_index_without_its_sides_on_the_device(arch, name):
    """cf_index_options::sides = -1: the planes made, the BWT sides out of HBM — the wide ftab is made over the planes, every kernel
    steps over them (the 300-base reads of r250's neighbours go through the byte-window kernel), the debug taps and the restore
    still answer; rows as ever"""
    d, c, kw, nm, ql, seq, off, seeds, paired = load_case(arch, name)
    base = os.path.join(d, "idx")
    ix = capi.Index(base, device=0, sides=-1, wide_ftab_chars=12)
    cfg = ix.describe()
    assert cfg["sides_dropped"] == 1 and cfg["occ_planes"] == 1 and cfg["file_bytes_dropped"] > 0 and cfg["wide_ftab_chars"] == 12, cfg
    clf = capi.Classifier(ix, **kw)
    b, m, ln = capi.pack_reads(seq, off)
    slot = capi.Slot(clf)
    slot.submit(b, m, ln, np.ascontiguousarray(seeds, dtype=np.uint32), paired=paired)
    assert tsv_of(ix, kw.get("k", 5), nm, ql, slot.wait()) == open(os.path.join(d, c["tsv"])).read()
    slot.close()
    # the byte-window kernel (reads beyond 256 bases) over the planes: the 250-base reads doubled
    if name == "r250_k5":
        full = dev_index(arch)
        s2 = np.concatenate([np.concatenate([seq[int(off[r]):int(off[r + 1])]] * 2) for r in range(len(nm))])
        o2 = np.concatenate([[0], np.cumsum([2 * int(off[r + 1] - off[r]) for r in range(len(nm))])]).astype(np.uint64)
        got, ref = [], []
        for index, out in ((ix, got), (full, ref)):
            cl2 = capi.Classifier(index, **kw)
            bt = cl2.batch(s2, o2, seeds, False)
            bt.classify()
            r2, n2, sc2 = bt.results()
            out.append(reads.format_tsv(index.seqid, nm, [2 * x for x in ql], r2, n2, sc2))
            bt.close(); cl2.close()
        assert got == ref
    # taps and restore
    full = dev_index(arch)
    rng = np.random.default_rng(3)
    rows = rng.integers(0, ix.text_len + 1, size=4000, dtype=np.uint64)
    chars = rng.integers(0, 4, size=len(rows), dtype=np.uint8)
    assert np.array_equal(ix.debug_rank(chars, rows), full.debug_rank(chars, rows))
    assert np.array_equal(ix.debug_resolve(rows), full.debug_resolve(rows))
    assert np.array_equal(ix.restore(), full.restore())                 # (from the resident text tables)
    clf.close(); ix.close()


def test_dense_pack_layout():
    """capi.dense_pack: four bases per byte, every read on a byte, the sparse N mask in the device's word numbering"""
    rng = np.random.default_rng(11)
    for L in (1, 3, 4, 5, 31, 32, 33, 100, 150, 250):
        codes = rng.integers(0, 4, (7, L), dtype=np.uint8)
        codes[rng.random((7, L)) < 0.03] = 4
        b4, ni, nk = capi.dense_pack(codes)
        bpr, W = (L + 3) // 4, (L + 31) // 32
        assert len(b4) == 7 * bpr
        for r in range(7):
            for i in range(L):
                got = (int(b4[r * bpr + i // 4]) >> (2 * (i % 4))) & 3
                assert got == (0 if codes[r, i] > 3 else int(codes[r, i]))
        want = {}
        for r, i in zip(*np.nonzero(codes > 3)):
            want[r * W + i // 32] = want.get(r * W + i // 32, 0) | (1 << (i % 32))
        assert dict(zip(ni.tolist(), nk.tolist())) == want
    assert capi.lib().cf_narrow_max_score(0x40, 100, 0, 0) == 85 * 85 and capi.lib().cf_narrow_max_score(0xc1, 100, 150, 1) == 85 * 85 + 135 * 135
    assert capi.lib().cf_narrow_max_score(0x80, 100, 150, 1) == 135 * 135 and capi.lib().cf_narrow_max_score(0x01, 100, 150, 1) == 0


def test_narrow_results_expand_on_the_host():
    """cf_results_narrow_expand / cf_narrow_max_score are host code (no device): 16-byte rows + one byte per query back into cf_row,
    n_rows and max_score — the taxID out of the index's taxon table, max_score (classifier.h:530-536) from the "took part" bits and
    the lengths, a taxon index outside the table refused"""
    import ctypes as C
    d, _ = common.golden("synth_small")
    ix = capi.Index(os.path.join(d, "idx"), host_only=True)
    taxa = ix.taxon_ids()
    rng = np.random.default_rng(23)
    nq = 200
    paired = 1
    qn = rng.integers(0, 4, nq).astype(np.uint8)                   # rows printed per query
    part = rng.integers(0, 4, nq).astype(np.uint8)                 # bit 0: mate 1 took part, bit 1: mate 2
    qinfo = (qn | ((part & 1) << 6) | ((part >> 1) << 7)).astype(np.uint8)
    tot = int(qn.sum())
    r16 = np.zeros(tot, dtype=capi.ROW16_DTYPE)
    r16["unique_id"] = rng.integers(0, 50, tot); r16["taxon_idx"] = rng.integers(0, len(taxa), tot)
    r16["score"] = rng.integers(0, 10 ** 6, tot); r16["hit_len"] = rng.integers(15, 500, tot)
    score2 = rng.integers(0, 100, nq).astype(np.uint32)
    lens = rng.integers(10, 300, 2 * nq).astype(np.uint32)
    rn = capi.ResultsNarrow()
    rn.rows, rn.qinfo, rn.score2, rn.n_queries, rn.total_rows = r16.ctypes.data, qinfo.ctypes.data, score2.ctypes.data, nq, tot
    rows, n_rows, ms = np.zeros(tot, dtype=capi.ROW_DTYPE), np.zeros(nq, dtype=np.uint32), np.zeros(nq, dtype=np.uint32)
    capi._check(ix.L.cf_results_narrow_expand(ix.h, C.byref(rn), lens.ctypes.data, 0, paired, rows.ctypes.data, n_rows.ctypes.data, ms.ctypes.data))
    assert np.array_equal(n_rows, qn)
    assert np.array_equal(rows["tax_id"], taxa[r16["taxon_idx"]]) and np.array_equal(rows["taxon_idx"], r16["taxon_idx"])
    for f in ("unique_id", "score", "hit_len"):
        assert np.array_equal(rows[f], r16[f])
    sq = lambda L: (int(L) - 15) ** 2 if L > 15 else 0
    for q in range(nq):
        p0, p1 = part[q] & 1, part[q] >> 1
        want = sq(lens[2 * q]) + sq(lens[2 * q + 1]) if p0 and p1 else sq(lens[2 * q]) if p0 else sq(lens[2 * q + 1]) if p1 else 0
        assert ms[q] == want, q
    # one length for the whole batch (a dense batch), single-end
    capi._check(ix.L.cf_results_narrow_expand(ix.h, C.byref(rn), None, 100, 0, rows.ctypes.data, n_rows.ctypes.data, ms.ctypes.data))
    assert np.array_equal(ms, np.where(part & 1, 85 * 85, 0))
    assert ix.L.cf_narrow_max_score(0x40, 70000, 0, 0) == 0xffffffff          # (2^32 or more: "never reached")
    r16["taxon_idx"][0] = len(taxa)
    with pytest.raises(capi.CfError):
        capi._check(ix.L.cf_results_narrow_expand(ix.h, C.byref(rn), lens.ctypes.data, 0, paired, rows.ctypes.data, n_rows.ctypes.data, ms.ctypes.data))
    ix.close()


def test_dense_unpack_body_gives_the_word_form():
    """dense_unpack_body (the kernel behind cf_batch_upload_dense_async) stepped on the CPU: the dense form of a read set comes
    out as exactly the words capi.pack_reads makes of it — every length around the byte and word boundaries"""
    import ctypes as C
    from emu import emu
    L_ = emu.lib()
    rng = np.random.default_rng(17)
    for L in (0, 1, 3, 4, 5, 8, 31, 32, 33, 63, 64, 65, 100, 128, 150, 250, 257):
        n = 9
        codes = rng.integers(0, 4, (n, L), dtype=np.uint8)
        b4, ni, nk = capi.dense_pack(codes)
        dense = np.concatenate([b4, np.full(32, 0xff, dtype=np.uint8)])            # (whatever lies behind the reads must not leak in)
        W = (L + 31) // 32
        bases, rlen = np.zeros(n * W + 1, dtype=np.uint64), np.zeros(n + 1, dtype=np.uint32)
        L_.emu_dense_unpack(dense.ctypes.data_as(C.c_void_p), n, L, bases.ctypes.data_as(C.c_void_p), rlen.ctypes.data_as(C.c_void_p))
        off = np.arange(n + 1, dtype=np.uint64) * np.uint64(L)
        wb, wm, wl = capi.pack_reads(codes.reshape(-1), off)
        assert np.array_equal(bases[:n * W], wb) and bases[n * W] == 0, L
        assert np.array_equal(rlen[:n], wl) and rlen[n] == 0, L
        # round 6: the same kernel also leaves the forward strands' words in search order (the search kernel's S_REC2 loads them):
        # exactly what rev_word makes of the unpacked words, nothing written past the reads
        L_.emu_dense_unpack_rev.restype = C.c_uint64
        bases2, rlen2, rev = np.zeros(n * W + 1, dtype=np.uint64), np.zeros(n + 1, dtype=np.uint32), np.full(n * W + 1, 0x1234, dtype=np.uint64)
        assert L_.emu_dense_unpack_rev(dense.ctypes.data_as(C.c_void_p), n, L, bases2.ctypes.data_as(C.c_void_p), rlen2.ctypes.data_as(C.c_void_p), rev.ctypes.data_as(C.c_void_p)) == 0, L
        assert np.array_equal(bases2, bases) and rev[n * W] == 0x1234, L


@pytest.mark.gpu
@pytest.mark.parametrize("arch,name", common.all_cases())
def test_dense_reads_and_narrow_results_change_no_row(arch, name):
    """the narrow forms of both directions (cf_batch_upload_dense_async: four bases per byte, no length array;
    CF_RESULTS_NARROW: 16-byte rows, one byte + 2ndBestScore per query) against the word form and the wide rows: the same rows,
    row counts, 2ndBestScore and max_score for every golden case — ragged read sets through the narrow results alone, read sets
    of one length (250 bp with N runs, 2 x 150 bp pairs, FASTQ, the worked example) through both"""
    d, c, kw, nm, ql, seq, off, seeds, paired = load_case(arch, name)
    if kw.get("k", 5) > 63:
        pytest.skip("the narrow result format holds -k <= 63")
    ix = dev_index(arch)
    clf = capi.Classifier(ix, **kw)
    seeds = np.ascontiguousarray(seeds, dtype=np.uint32)
    b, m, ln = capi.pack_reads(seq, off)
    wide = capi.Slot(clf)
    wide.submit(b, m, ln, seeds, paired=paired)
    rows, first, n_rows, score2, max_score, info = wide.wait()
    want = tsv_of(ix, clf.params.khits, nm, ql, (rows, first, n_rows, score2, max_score, info))
    assert want == open(os.path.join(d, c["tsv"])).read()
    slot = capi.Slot(clf)
    slot.set_result_format(capi.RESULTS_NARROW)
    with pytest.raises(capi.CfError):
        capi.Slot.wait(slot)                                  # nothing in flight
    slot.submit(b, m, ln, seeds, paired=paired)
    with pytest.raises(capi.CfError):
        r = capi.Results()
        capi._check(slot.L.cf_batch_wait(slot.h, capi.C.byref(r)))        # a narrow slot is waited for with cf_batch_wait_narrow
    r16, qinfo, s2n, info_n = slot.wait_narrow()                          # ... and the slip did not cost the batch in flight
    assert np.array_equal(qinfo & 0x3f, n_rows) and np.array_equal(s2n, score2) and len(r16) == len(rows)
    for f in ("unique_id", "taxon_idx", "score", "hit_len"):
        assert np.array_equal(r16[f], rows[f]), f
    slot.submit(b, m, ln, seeds, paired=paired)
    rows_x, n_rows_x, s2x, ms_x, _ = slot.wait_narrow(expand=(ln, 0, paired))
    assert np.array_equal(rows_x, rows) and np.array_equal(n_rows_x, n_rows) and np.array_equal(ms_x, max_score) and np.array_equal(s2x, score2)
    lens = np.unique(ln)
    if len(lens) == 1 and len(ln):
        L = int(lens[0])
        codes = np.ascontiguousarray(seq, dtype=np.uint8).reshape(len(ln), L)
        b4, ni, nk = capi.dense_pack(codes)
        for fmt in (capi.RESULTS_NARROW, capi.RESULTS_ROWS):
            s = capi.Slot(clf)
            s.set_result_format(fmt)
            s.submit_dense(b4, seeds, L, paired=paired, nwords=(ni, nk))
            if fmt == capi.RESULTS_NARROW:
                rows_d, n_rows_d, s2d, ms_d, _ = s.wait_narrow(expand=(None, L, paired))
            else:
                rows_d, _f, n_rows_d, s2d, ms_d, _ = s.wait()
            assert np.array_equal(rows_d, rows) and np.array_equal(n_rows_d, n_rows) and np.array_equal(ms_d, max_score) and np.array_equal(s2d, score2)
            # ... and once more through the same slot after a batch in the word form (the mask buffer's bookkeeping)
            s.submit(b, m, ln, seeds, paired=paired)
            (s.wait_narrow() if fmt == capi.RESULTS_NARROW else s.wait())
            s.submit_dense(b4, seeds, L, paired=paired, nwords=(ni, nk))
            if fmt == capi.RESULTS_NARROW:
                rows_e = s.wait_narrow(expand=(None, L, paired))[0]
            else:
                rows_e = s.wait()[0]
            assert np.array_equal(rows_e, rows)
            s.close()
    wide.close(); slot.close(); clf.close()
