"""Parity at scale: indexes far beyond the L2 / Infinity Cache, millions of reads, every row compared with
the unmodified reference (oracle/_ref, 16 threads) — a u32 SA sample (> 65,535 sequences) with the 64-bit side
division forced, and a repeat-rich index (strain clusters, shared operons, low-complexity tracts) queried with
low-complexity reads, once with the row workspace cut down so the batch takes several passes of the row stage."""
import os
import shutil
import sys
import tempfile

import numpy as np
import pytest

import common
from centrifuge_amd import capi, reads as rd
from oracle import oracle as O

sys.path.insert(0, common.ROOT)
sys.path.insert(0, os.path.join(common.ROOT, "tools"))

pytestmark = [pytest.mark.gpu, pytest.mark.skipif(not O.have_ref(), reason="oracle/_ref (the compiled reference) is not built")]


def build(torch, bench, synth, d, n_genomes, genome_len, recipe, uid):
    g = bench.gpu_genomes(torch, n_genomes, genome_len, recipe=recipe)
    host = g.cpu().numpy()
    synth.write_taxonomy(d, n_genomes, uid_prefix=uid)
    base = os.path.join(d, "idx")
    capi.build_index(base, codes=host.reshape(-1), seq_off=np.arange(n_genomes + 1, dtype=np.uint64) * np.uint64(genome_len),
                     seq_names=[b"%s%d g" % (uid.encode(), i) for i in range(n_genomes)], conversion_table=os.path.join(d, "conv.tsv"),
                     taxonomy_tree=os.path.join(d, "nodes.dmp"), name_table=os.path.join(d, "names.dmp"))
    del host
    return g, base


def low_complexity(rng, n, L):
    """homopolymers, di- and tri-nucleotide repeats with a few substitutions"""
    out = np.empty((n, L), dtype=np.uint8)
    for i in range(n):
        period = int(rng.integers(1, 4))
        unit = rng.integers(0, 4, period, dtype=np.uint8)
        out[i] = np.tile(unit, L // period + 1)[:L]
        for _ in range(int(rng.integers(0, 3))):
            out[i, int(rng.integers(0, L))] = rng.integers(0, 4)
    return out


def classify_all(ix, clf, codes, names, seeds, limits=None, wire="words"):
    """wire = "words": the packed word form in, wide rows out (cf_batch_upload_packed_async / cf_batch_wait);
    "narrow": the dense form in (four bases per byte) and 16-byte rows out (cf_batch_upload_dense_async, CF_RESULTS_NARROW) — what
    bench.py's legs and centrifuge-class submit by default since round 5"""
    import torch
    import bench
    n, L = codes.shape
    if wire == "narrow":
        b4, ni, nk = capi.dense_pack(codes)
        slot = capi.Slot(clf)
        slot.set_result_format(capi.RESULTS_NARROW)
        if limits:
            slot.set_limits(**limits)
        slot.submit_dense(b4, seeds, L, paired=False, nwords=(ni, nk))
        rows, n_rows, score2, max_score, info = slot.wait_narrow(expand=(None, L, False))
        slot.close()
        first = np.zeros(len(n_rows) + 1, dtype=np.uint64)
        first[1:] = np.cumsum(n_rows, dtype=np.uint64)
        tsv = rd.format_tsv(ix.seqid, names, [L] * n, capi.unpack_rows(rows, first, n_rows, 5), n_rows, score2)
        return tsv, info
    bd, md = bench.gpu_pack(torch, torch.from_numpy(codes).cuda())          # packed on the GPU (plumbing): 2-bit words + N masks
    b, m = bd.cpu().numpy().view(np.uint64), md.cpu().numpy().astype(np.uint32)
    ln = np.full(n, L, dtype=np.uint32)
    del bd, md
    slot = capi.Slot(clf)
    if limits:
        slot.set_limits(**limits)
    slot.submit(b, m, ln, seeds)
    rows, first, n_rows, score2, max_score, info = slot.wait()
    slot.close()
    tsv = rd.format_tsv(ix.seqid, names, [L] * n, capi.unpack_rows(rows, first, n_rows, 5), n_rows, score2)
    return tsv, info


@pytest.mark.parametrize("shape", ["wide_sa", "repeat"])
def test_every_row_matches_the_reference_at_scale(shape):
    import torch
    import bench
    import synth
    d = tempfile.mkdtemp(prefix="cf_scale_")
    try:
        if shape == "wide_sa":            # 70,000 sequences x 32 Kbp = 2.3 Gbp: u32 SA sample; the 64-bit side division forced
            os.environ["CF_FORCE_WIDE_SIDE"] = "1"
            g, base = build(torch, bench, synth, d, 70000, 32768, "iid", "seq")
            n_reads, n_low = 2000000, 0
        else:                             # 512 x 4 Mbp = 2.1 Gbp, repeat-rich, "cid" uids: compressed index (ihits 20)
            g, base = build(torch, bench, synth, d, 512, 4194304, "repeat", "cid|")
            n_reads, n_low = 1000000, 50000
        codes = bench.gpu_sample_reads(torch, g, n_reads, 100, seed=4242).cpu().numpy()
        del g
        torch.cuda.empty_cache()
        if n_low:
            codes[-n_low:] = low_complexity(np.random.default_rng(6), n_low, 100)
        names = bench.read_names(n_reads)
        seeds = bench.seeds_for(codes, names)
        fa = os.path.join(d, "reads.fa")
        bench.write_fasta(fa, names, codes)
        want = O.ref_classify(base, os.path.join(d, "ref.tsv"), os.path.join(d, "ref.rep"), u=fa, threads=16)
        ix = capi.Index(base, device=0)
        assert ix.sa_width == (4 if shape == "wide_sa" else 2)
        assert bool(ix.L.cf_index_compressed(ix.h)) == (shape == "repeat")
        # the resolve table (round 6): with room for every table it holds every row beside SA[row] at every row, and is then made from
        # the stop rows' TEXT POSITIONS instead of by a walk from every row — the same index once more with the walks: the ends of the
        # table and 200 k random rows resolve alike (u32 references on the 70,000-sequence index), the same bound on the walk-left
        assert ix.L.cf_index_resolve_by_position(ix.h) == 1 and ix.L.cf_index_resolve_rate(ix.h) == 0
        os.environ["CF_DENSE_BY_POS"] = "0"
        try:
            walks = capi.Index(base, device=0)
        finally:
            del os.environ["CF_DENSE_BY_POS"]
        assert walks.L.cf_index_resolve_by_position(walks.h) == 0 and walks.L.cf_index_resolve_rate(walks.h) == 0
        n_rows = ix.text_len + 1
        probe = np.concatenate([np.arange(0, 2000, dtype=np.uint64), np.arange(n_rows - 2000, n_rows, dtype=np.uint64),
                                np.random.default_rng(8).integers(0, n_rows, size=200000, dtype=np.uint64)])
        assert np.array_equal(ix.debug_resolve(probe), walks.debug_resolve(probe))
        assert ix.L.cf_index_walk_bound(ix.h) == walks.L.cf_index_walk_bound(walks.h) > 0
        walks.close()
        clf = capi.Classifier(ix)
        nm = [bytes(x) for x in names]
        got, info = classify_all(ix, clf, codes, nm, seeds)
        assert got == want, common.first_diff(got, want)
        before = clf.counts()
        # the narrow forms of both directions (dense reads in, 16-byte rows out: the default of the bench legs and of centrifuge-class)
        # at the same scale: the same TSV, the same counters once more
        got_n, _ = classify_all(ix, clf, codes, nm, seeds, wire="narrow")
        assert got_n == want, common.first_diff(got_n, want)
        twice = clf.counts()
        assert np.array_equal(twice[0], 2 * before[0]) and np.array_equal(twice[1], 2 * before[1])
        clf.reset_counts()
        # the text forms (round 6; cf_batch_upload_text / cf_batch_wait_text): the FASTA file as it is, in two blocks, parsed and
        # printed on the device — the reference's TSV, and the reference's report from what the device tallied alone
        text = open(fa, "rb").read()
        cut = text.index(b"\n>", len(text) // 2) + 1
        slot = capi.Slot(clf)
        slot.set_result_format(capi.RESULTS_NARROW)
        rep, out = capi.Report(ix), [rd.HEADER.encode()]
        for block in (text[:cut], text[cut:]):
            ti = slot.submit_text(block, capi.TEXT_FASTA)
            assert not ti.irregular and ti.max_len == 100
            t_, tuples, _ = slot.wait_text()
            out.append(t_)
            rep.add_tuples(tuples)
        got_t = b"".join(out).decode()
        assert got_t == want, common.first_diff(got_t, want)
        once = clf.counts()
        assert np.array_equal(once[0], before[0]) and np.array_equal(once[1], before[1])
        rep.adopt_device_tally(once[0], once[1], clf.counts_single())
        rep.write(os.path.join(d, "mine.rep"))
        assert open(os.path.join(d, "mine.rep")).read() == open(os.path.join(d, "ref.rep")).read()
        rep.close(); slot.close()
        del text, out
        clf.reset_counts()
        got_w, _ = classify_all(ix, clf, codes, nm, seeds)                         # (the counters of ONE pass for what follows)
        assert got_w == want
        if shape == "wide_sa":
            # 70,000 species + their genera: far more taxa than k_count has LDS slots, and a chunk of queries touches more of them
            # than fit — the hashed slots and the far atomics side by side; the report the counters give is the reference's
            assert ix.num_taxa > 4096
            rows_rep = open(os.path.join(d, "ref.rep")).read().splitlines()[1:]
            ref_counts = {int(f.split("\t")[1]): (int(f.split("\t")[4]), int(f.split("\t")[5])) for f in rows_rep}
            tax = ix.taxon_ids()
            nz = np.nonzero(before[0])[0]
            mine = {int(tax[i]): (int(before[0][i]), int(before[1][i])) for i in nz if tax[i] != 0}
            assert mine == ref_counts
        if shape == "repeat":
            # the same batch with a row workspace a fraction of what it plans: several passes, same rows, same counters
            got2, info2 = classify_all(ix, clf, codes, nm, seeds, limits=dict(rows_per_pass=max(1000, int(info["planned_sa_rows"]) // 7)))
            assert info2["row_passes"] >= 7 and got2 == want
            after = clf.counts()
            assert np.array_equal(after[0], 2 * before[0]) and np.array_equal(after[1], 2 * before[1])
            # and the report the counters + rows give is the reference's
            rows_rep = open(os.path.join(d, "ref.rep")).read().splitlines()[1:]
            ref_counts = {int(f.split("\t")[1]): (int(f.split("\t")[4]), int(f.split("\t")[5])) for f in rows_rep}
            tax = ix.taxon_ids()
            mine = {int(tax[i]): (int(before[0][i]), int(before[1][i])) for i in range(len(tax)) if before[0][i] and tax[i] != 0}
            assert mine == ref_counts
        clf.close(); ix.close()
    finally:
        os.environ.pop("CF_FORCE_WIDE_SIDE", None)
        shutil.rmtree(d, ignore_errors=True)


# ------------------------------------------------------------------ the other shapes the bench runs
# One 2.1 Gbp index (512 genomes x 4 Mbp in genera of 8 at 5 %) opened four ways, >= 500 k reads each, every row compared with
# the compiled reference: the two-lane kernel over the sides (the form the nt-scale config runs: no planes), 2 x 150 bp FR pairs
# (96-byte strand records), 250 bp reads with the text tables at every 32nd row (128-byte records; lazy hits and text
# verification with a long way back from the inverse sample), 300 bp reads through the byte-window kernel k_search.
@pytest.fixture(scope="module")
def iid_index():
    import torch
    import bench
    import synth
    d = tempfile.mkdtemp(prefix="cf_scale2_")
    g, base = build(torch, bench, synth, d, 512, 4194304, "iid", "seq")
    if not os.environ.get("CF_TEST_SMALL_RANGE_ROWS"):         # genera at 5 %, no strains: the probe finds nothing to finish against the text
        ix = capi.Index(base, device=0)
        cfg = ix.describe()
        ix.close()
        assert 0 <= cfg["repeat_fraction"] < 0.10 and cfg["small_range_rows"] == 0 and cfg["plan_realised"] == 1, cfg
    yield d, g, base
    del g
    torch.cuda.empty_cache()
    shutil.rmtree(d, ignore_errors=True)


def more_substitutions(codes, rng, per_read):
    """`per_read` further substitutions in every read (several partial hits per strand, matches of every length)"""
    n, L = codes.shape
    for _ in range(per_read):
        pos = rng.integers(0, L, n)
        add = rng.integers(1, 4, n).astype(np.uint8)
        ok = codes[np.arange(n), pos] < 4
        codes[np.arange(n), pos] = np.where(ok, (codes[np.arange(n), pos] + add) & 3, codes[np.arange(n), pos])
    return codes


SHAPES = {
    # name: (env for cf_index_open, read length, pairs, reads, extra substitutions per read, expectations on the opened index)
    "sides_two_lanes": ({"CF_OCC_PLANES": "0"}, 100, False, 600000, 0, dict(planes=0)),
    "pairs_150": ({}, 150, True, 600000, 1, dict(planes=1)),
    "len250_text_every_32nd": ({"CF_TEXT_VERIFY_RATE": "5"}, 250, False, 500000, 3, dict(planes=1, text_rate=5)),
    "len250_sides_text_every_32nd": ({"CF_TEXT_VERIFY_RATE": "5", "CF_OCC_PLANES": "0"}, 250, False, 500000, 3, dict(planes=0, text_rate=5)),
    "len300_byte_window": ({}, 300, False, 500000, 2, dict(planes=1)),
    "len100_single_base_steps": ({"CF_PAIR_PLANES": "0"}, 100, False, 600000, 1, dict(planes=1, pair=0)),
}


@pytest.mark.parametrize("shape", sorted(SHAPES))
def test_other_kernel_forms_match_the_reference_at_scale(iid_index, shape):
    import torch
    import bench
    d, g, base = iid_index
    env, L, paired, n_reads, extra, expect = SHAPES[shape]
    rng = np.random.default_rng(len(shape))
    if paired:
        codes = bench.gpu_sample_pairs(torch, g, n_reads // 2, L, seed=99).cpu().numpy()
    else:
        codes = bench.gpu_sample_reads(torch, g, n_reads, L, seed=99 + L).cpu().numpy()
    codes = more_substitutions(codes, rng, extra)
    per = 2 if paired else 1
    nq = n_reads // per
    names = bench.read_names(nq)
    seeds = bench.seeds_for(codes, np.repeat(names, per, axis=0))
    t = os.path.join(d, shape)
    os.makedirs(t, exist_ok=True)
    if paired:
        bench.write_fasta(os.path.join(t, "r1.fa"), names, codes[0::2], b"/1")
        bench.write_fasta(os.path.join(t, "r2.fa"), names, codes[1::2], b"/2")
        want = O.ref_classify(base, os.path.join(t, "ref.tsv"), os.path.join(t, "ref.rep"), m1=os.path.join(t, "r1.fa"), m2=os.path.join(t, "r2.fa"), threads=16)
    else:
        bench.write_fasta(os.path.join(t, "r.fa"), names, codes)
        want = O.ref_classify(base, os.path.join(t, "ref.tsv"), os.path.join(t, "ref.rep"), u=os.path.join(t, "r.fa"), threads=16)
    old = {k: os.environ.get(k) for k in env}
    os.environ.update(env)
    try:
        ix = capi.Index(base, device=0)
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
    assert ix.L.cf_index_occ_planes(ix.h) == expect["planes"]
    assert ix.describe()["pair_planes"] == expect.get("pair", expect["planes"])       # two bases per step wherever the planes are (by default)
    if "text_rate" in expect:
        assert ix.L.cf_index_text_verify_rate(ix.h) == expect["text_rate"]
    clf = capi.Classifier(ix)
    bd, md = bench.gpu_pack(torch, torch.from_numpy(codes).cuda())
    b, m = bd.cpu().numpy().view(np.uint64), md.cpu().numpy().astype(np.uint32)
    del bd, md
    slot = capi.Slot(clf)
    slot.submit(b, m, np.full(n_reads, L, dtype=np.uint32), seeds, paired=paired)
    rows, first, n_rows, score2, max_score, info = slot.wait()
    ops = slot.opcounts()
    slot.close()
    got = rd.format_tsv(ix.seqid, [bytes(x) for x in names], [L * per] * nq, capi.unpack_rows(rows, first, n_rows, 5), n_rows, score2)
    assert got == want, common.first_diff(got, want)
    # the shape really took the path it is named after
    if "text_rate" in expect:
        assert ops.n_verify > n_reads // 4 and ops.n_text_loads > 0
    if shape == "len300_byte_window":
        assert ops.n_ftab_wide == 0 and ops.n_ftab > n_reads          # k_search knows no wide ftab
    # the report (per-taxon counters of the device) is the reference's as well
    counts = clf.counts()
    rep = open(os.path.join(t, "ref.rep")).read().splitlines()[1:]
    ref_counts = {int(f.split("\t")[1]): (int(f.split("\t")[4]), int(f.split("\t")[5])) for f in rep}
    tax = ix.taxon_ids()
    mine = {int(tax[i]): (int(counts[0][i]), int(counts[1][i])) for i in range(len(tax)) if counts[0][i] and tax[i] != 0}
    assert mine == ref_counts
    clf.close(); ix.close()


# ------------------------------------------------------------------ the options at scale
# Everything above runs with the default options.  Here the 2.1 Gbp repeat-rich index (strain clusters at 0.1 - 1 %: reads that
# hit several species of a genus with equal scores) under the options that send queries through the rest of Classifier::go —
# the climb when a read has more best hits than -k (classifier.h:399-520), rank lifting (:982-1001), host / exclude lists
# (:339, :385-394), a lower --min-hitlen, --no-traverse (:419-425) — >= 300 k reads each, every row and the report's counters
# against the compiled reference, and the general score kernel (the one that owns the climb) must really have run.
@pytest.fixture(scope="module")
def repeat_index():
    import torch
    import bench
    import synth
    d = tempfile.mkdtemp(prefix="cf_scale3_")
    g, base = build(torch, bench, synth, d, 512, 4194304, "repeat", "cid|")
    n = 300000
    se = bench.gpu_sample_reads(torch, g, n, 100, seed=31337).cpu().numpy()
    se[-20000:] = low_complexity(np.random.default_rng(9), 20000, 100)
    pe = bench.gpu_sample_pairs(torch, g, n // 2, 125, seed=31338).cpu().numpy()
    # the longer records of the small-range kernels (192- and 256-base strand records): 2 x 150 bp pairs, 250 bp reads
    LONGER["pairs_150"] = more_substitutions(bench.gpu_sample_pairs(torch, g, 100000, 150, seed=31339).cpu().numpy(), np.random.default_rng(150), 1)
    LONGER["len250"] = more_substitutions(bench.gpu_sample_reads(torch, g, 200000, 250, seed=31340).cpu().numpy(), np.random.default_rng(250), 2)
    del g
    torch.cuda.empty_cache()
    names = bench.read_names(n)
    bench.write_fasta(os.path.join(d, "r.fa"), names, se)
    bench.write_fasta(os.path.join(d, "r1.fa"), names[:n // 2], pe[0::2], b"/1")
    bench.write_fasta(os.path.join(d, "r2.fa"), names[:n // 2], pe[1::2], b"/2")
    # opened with everything automatic: cf_index_open finds the collection repeat-rich (neighbouring suffix-array rows that share
    # their preceding 24 bases) and makes the SA / inverse-SA samples at every row — EVERY test below runs with small ranges
    # finished against the text, the general kernels and the options behind them
    ix = capi.Index(base, device=0)
    cfg = ix.describe()
    assert bool(ix.L.cf_index_compressed(ix.h)) and cfg["pair_planes"] == 1
    if not os.environ.get("CF_TEST_SMALL_RANGE_ROWS"):
        assert cfg["repeat_fraction"] > 0.2 and cfg["small_range_rows"] == 4 and cfg["text_verify_rate"] == 0 and cfg["plan_realised"] == 1, cfg
    yield d, base, ix, names, se, pe
    ix.close()
    LONGER.clear()
    shutil.rmtree(d, ignore_errors=True)


LONGER = {}


OPTION_SHAPES = {
    # name: (reference arguments, pairs, least share of the queries the general score kernel must have seen)
    "k1": (["-k", "1"], False, 0.01),
    "k2": (["-k", "2"], False, 0.01),
    "genus_rank": (["--classification-rank", "genus"], False, 0.0),
    "host_and_exclude": (["--host-taxids", "1003,101,1200", "--exclude-taxids", "1001,1017,102"], False, 0.0),
    "min_hitlen_15": (["--min-hitlen", "15"], False, 0.0),
    "k1_no_traverse": (["-k", "1", "--no-traverse"], False, 0.01),
    "pairs_k1": (["-k", "1"], True, 0.01),
    "family_k1_minhit16": (["-k", "1", "--classification-rank", "family", "--min-hitlen", "16"], False, 0.0),
}


@pytest.mark.parametrize("shape", sorted(OPTION_SHAPES))
def test_options_match_the_reference_at_scale(repeat_index, shape):
    import torch
    import bench
    d, base, ix, names, se, pe = repeat_index
    args, paired, slow_min = OPTION_SHAPES[shape]
    kw, _ = common.case_kwargs(args)
    k = kw.get("k", 5)
    codes = pe if paired else se
    n_reads, L = codes.shape
    per = 2 if paired else 1
    nq = n_reads // per
    t = os.path.join(d, shape)
    os.makedirs(t, exist_ok=True)
    files = dict(m1=os.path.join(d, "r1.fa"), m2=os.path.join(d, "r2.fa")) if paired else dict(u=os.path.join(d, "r.fa"))
    want = O.ref_classify(base, os.path.join(t, "ref.tsv"), os.path.join(t, "ref.rep"), threads=16, extra=args, **files)
    clf = capi.Classifier(ix, **kw)
    bd, md = bench.gpu_pack(torch, torch.from_numpy(codes).cuda())
    b, m = bd.cpu().numpy().view(np.uint64), md.cpu().numpy().astype(np.uint32)
    del bd, md
    nm = names[:nq]
    seeds = bench.seeds_for(codes, np.repeat(nm, per, axis=0))
    slot = capi.Slot(clf)
    slot.submit(b, m, np.full(n_reads, L, dtype=np.uint32), seeds, paired=paired)
    rows, first, n_rows, score2, max_score, info = slot.wait()
    slot.close()
    got = rd.format_tsv(ix.seqid, [bytes(x) for x in nm], [L * per] * nq, capi.unpack_rows(rows, first, n_rows, k), n_rows, score2)
    assert got == want, common.first_diff(got, want)
    assert int(info["slow_score"]) >= slow_min * nq, (int(info["slow_score"]), nq)       # the climb really ran on the device
    counts = clf.counts()
    rep = open(os.path.join(t, "ref.rep")).read().splitlines()[1:]
    ref_counts = {int(f.split("\t")[1]): (int(f.split("\t")[4]), int(f.split("\t")[5])) for f in rep}
    tax = ix.taxon_ids()
    mine = {int(tax[i]): (int(counts[0][i]), int(counts[1][i])) for i in range(len(tax)) if counts[0][i] and tax[i] != 0}
    assert mine == ref_counts
    clf.close()


def _submit(clf, codes, seeds, paired):
    import torch
    import bench
    n_reads, L = codes.shape
    bd, md = bench.gpu_pack(torch, torch.from_numpy(codes).cuda())
    b, m = bd.cpu().numpy().view(np.uint64), md.cpu().numpy().astype(np.uint32)
    del bd, md
    slot = capi.Slot(clf)
    slot.submit(b, m, np.full(n_reads, L, dtype=np.uint32), seeds, paired=paired)
    out = slot.wait()
    ops = slot.opcounts()
    slot.close()
    return out, ops


def _requests(ops, n_reads):
    return (ops.n_ftab_wide + ops.n_ftab + ops.n_pair + ops.n_pair2 + ops.n_single + 2 * ops.n_verify + ops.n_text_loads) / float(n_reads)


def test_small_ranges_against_the_text_at_scale(repeat_index):
    """small ranges against the text on the 2.1 Gbp repeat-rich index (clusters of near-identical strains: ranges that stay a few
    rows wide) — the planner's own choice for this collection — against the same index with the option off: SA / inverse SA at
    every row, ranges of up to four rows finished against the text — fewer requests, the same rows and counters as the reference
    (default options, and -k 1 for the climb behind it)"""
    import bench
    d, base, ix, names, se, pe = repeat_index
    ix_off = capi.Index(base, device=0, small_range_rows=-1)
    cfg, cfg_off = ix.describe(), ix_off.describe()
    assert cfg["small_range_rows"] >= 2 and cfg["text_verify_rate"] == 0, cfg
    assert cfg_off["small_range_rows"] == 0 and cfg_off["repeat_fraction"] == -1.0, cfg_off      # (switched off: not probed either)
    n_reads, L = se.shape
    seeds = bench.seeds_for(se, names)
    for args in ([], ["-k", "1"]):
        kw, _ = common.case_kwargs(args)
        k = kw.get("k", 5)
        t = os.path.join(d, "small_ranges_k%d" % k)
        os.makedirs(t, exist_ok=True)
        want = O.ref_classify(base, os.path.join(t, "ref.tsv"), os.path.join(t, "ref.rep"), threads=16, extra=args, u=os.path.join(d, "r.fa"))
        counts = {}
        reqs = {}
        for tag, index in (("with", ix), ("without", ix_off)):
            clf = capi.Classifier(index, **kw)
            (rows, first, n_rows, score2, max_score, info), ops = _submit(clf, se, seeds, False)
            got = rd.format_tsv(index.seqid, [bytes(x) for x in names], [L] * n_reads, capi.unpack_rows(rows, first, n_rows, k), n_rows, score2)
            assert got == want, (tag, args, common.first_diff(got, want))
            counts[tag] = clf.counts()
            reqs[tag] = _requests(ops, n_reads)
            clf.close()
        assert np.array_equal(counts["with"][0], counts["without"][0]) and np.array_equal(counts["with"][1], counts["without"][1])
        assert reqs["with"] < 0.85 * reqs["without"], reqs                  # the small ranges really went to the text
    ix_off.close()


@pytest.mark.parametrize("shape", ["pairs_150", "len250"])
def test_small_ranges_with_longer_records_at_scale(repeat_index, shape):
    """the small-range variants of the 192- and 256-base strand records (k_search2_l1<6, ., 0, true> / <8, ., 0, true>): 2 x 150 bp FR
    pairs and 250 bp reads on the repeat-rich index, rows and counters against the compiled reference, requests against the
    same index with the option off"""
    import bench
    d, base, ix, names, se, pe = repeat_index
    codes = LONGER[shape]
    paired = shape == "pairs_150"
    n_reads, L = codes.shape
    per = 2 if paired else 1
    nq = n_reads // per
    nm = names[:nq]
    seeds = bench.seeds_for(codes, np.repeat(nm, per, axis=0))
    t = os.path.join(d, "longer_" + shape)
    os.makedirs(t, exist_ok=True)
    if paired:
        bench.write_fasta(os.path.join(t, "r1.fa"), nm, codes[0::2], b"/1")
        bench.write_fasta(os.path.join(t, "r2.fa"), nm, codes[1::2], b"/2")
        files = dict(m1=os.path.join(t, "r1.fa"), m2=os.path.join(t, "r2.fa"))
    else:
        bench.write_fasta(os.path.join(t, "r.fa"), nm, codes)
        files = dict(u=os.path.join(t, "r.fa"))
    want = O.ref_classify(base, os.path.join(t, "ref.tsv"), os.path.join(t, "ref.rep"), threads=16, **files)
    ix_off = capi.Index(base, device=0, small_range_rows=-1)
    reqs, counts = {}, {}
    for tag, index in (("with", ix), ("without", ix_off)):
        clf = capi.Classifier(index)
        (rows, first, n_rows, score2, max_score, info), ops = _submit(clf, codes, seeds, paired)
        got = rd.format_tsv(index.seqid, [bytes(x) for x in nm], [L * per] * nq, capi.unpack_rows(rows, first, n_rows, 5), n_rows, score2)
        assert got == want, (tag, common.first_diff(got, want))
        reqs[tag] = _requests(ops, n_reads)
        counts[tag] = clf.counts()
        clf.close()
    ix_off.close()
    rep = open(os.path.join(t, "ref.rep")).read().splitlines()[1:]
    ref_counts = {int(f.split("\t")[1]): (int(f.split("\t")[4]), int(f.split("\t")[5])) for f in rep}
    tax = ix.taxon_ids()
    mine = {int(tax[i]): (int(counts["with"][0][i]), int(counts["with"][1][i])) for i in range(len(tax)) if counts["with"][0][i] and tax[i] != 0}
    assert mine == ref_counts
    assert np.array_equal(counts["with"][0], counts["without"][0]) and np.array_equal(counts["with"][1], counts["without"][1])
    if ix.describe()["small_range_rows"] >= 2:
        assert reqs["with"] < 0.85 * reqs["without"], reqs
