"""The kernel bodies (centrifuge_amd/csrc/cf_kernels.hpp) single-stepped on the
CPU by tests/emu against the golden vectors.  This covers the host logic and the
kernels' control flow without a GPU; the 8-lane cooperative paths are covered by
the gpu-marked tests."""
import ctypes as C
import os

import numpy as np
import pytest

import common
from centrifuge_amd import reads
from emu import emu


def run_case(arch, name, search_version=2):
    emu.lib().emu_set_search_version(search_version)
    d, cases = common.golden(arch)
    c = [x for x in cases if x["name"] == name][0]
    kw, fastq = common.case_kwargs(c["args"])
    e = emu.Emu(os.path.join(d, "idx"))
    # seeds through the oracle-independent host code is a GPU-lib function; the emu
    # tests take them from the same formula implemented in the library when it is built
    names, qlens, seq, off, seeds, paired = reads.load([os.path.join(d, f) for f in c["reads"]], fastq)
    rows, n_rows, score2, cnt = e.classify(seq, off, seeds, paired=paired, counts=True, **kw)
    got = reads.format_tsv(e.seqid, names, qlens, rows, n_rows, score2)
    return d, c, e, got, cnt


@pytest.mark.parametrize("search_version", [2, 1])
@pytest.mark.parametrize("arch,name", common.all_cases())
def test_emulated_kernels_match_reference(arch, name, search_version):
    """search_version 2 = k_search2's body (strand records in "LDS"), 1 = k_search's byte-window body"""
    d, c, e, got, cnt = run_case(arch, name, search_version)
    ref = open(os.path.join(d, c["tsv"])).read()
    assert got == ref, common.first_diff(got, ref)
    # per-taxon counters against the reference's report (numReads, numUniqueReads)
    ntax = e.L.emu_num_taxa(e.h)
    mine = {}
    for i in range(ntax):
        t = e.L.emu_taxon_id(e.h, i)
        if t != 0 and cnt[i]:
            mine[t] = (int(cnt[i]), int(cnt[ntax + i]))
    rep = {}
    for ln in open(os.path.join(d, c["report"])).read().splitlines()[1:]:
        f = ln.split("\t")
        rep[int(f[1])] = (int(f[4]), int(f[5]))
    assert mine == rep


@pytest.mark.parametrize("slot_bits", [1, 2, 4])
@pytest.mark.parametrize("arch,name", common.all_cases())
def test_counters_through_the_hashed_slots(arch, name, slot_bits):
    """k_count with far fewer LDS slots than taxa: the open hash with its probing, and (2 or 4 slots, 8 probes) taxa that
    find no slot and go to the far atomics; the counters are those of the direct-mapped form"""
    d, c, e, got, want = run_case(arch, name)
    assert e.L.emu_num_taxa(e.h) > (1 << slot_bits)
    emu.lib().emu_set_count_slot_bits(slot_bits)
    try:
        d, c, e, got, cnt = run_case(arch, name)
    finally:
        emu.lib().emu_set_count_slot_bits(0)
    assert want.sum() > 0 and np.array_equal(cnt, want)


@pytest.mark.parametrize("seed", range(16))
def test_random_taxonomies(tmp_path, seed):
    """Random trees (tools/synth.py write_random_taxonomy: lineages of random depth over the whole rank vocabulary, "no rank" and
    unknown ranks in between, sequences on leaves / inner nodes / shared nodes / taxIDs the tree does not know, 40-bit taxIDs)
    under clusters of related genomes, random -k / --classification-rank / host / exclude lists: the 10-slot paths, the key
    lifting, the climb, the seqID rule and the report's names against the compiled reference.  (tests/fuzz/fuzz_taxonomy.py
    runs the same recipe for as long as one lets it.)"""
    import sys
    from oracle import oracle as O
    if not O.have_ref():
        pytest.skip("oracle/_ref (the compiled reference) is not built")
    sys.path.insert(0, os.path.join(common.ROOT, "tools"))
    import synth
    import test_report as TR
    from centrifuge_amd import capi
    rng = np.random.default_rng(77000 + seed)
    n_clusters, per = int(rng.integers(1, 5)), int(rng.integers(2, 7))
    L, d = int(rng.integers(1200, 3000)), str(tmp_path)
    g = synth.make_genomes(n_clusters * per, L, genus_size=per, divergence=float(rng.choice([0.0, 0.005, 0.02, 0.05])), seed=int(rng.integers(1 << 30)))
    synth.write_reference(d, g, genus_size=per)
    seq_tid, nodes = synth.write_random_taxonomy(d, rng, n_clusters, per)
    O.ref_build(d, threads=2)
    nm, s = synth.sample_reads(g, 120, 100, random_frac=0.05, n_frac=0.05, seed=int(rng.integers(1 << 30)))
    synth.write_fasta(os.path.join(d, "r.fa"), nm, s)
    kw = {"k": int(rng.choice([1, 1, 2, 3, 5, 20])), "min_hitlen": int(rng.choice([16, 22, 22, 30])),
          "rank": str(rng.choice(list(capi.RANK_SLOTS))), "traverse": bool(rng.random() < 0.8)}
    pool = sorted(set(seq_tid) | set(nodes))
    if rng.random() < 0.25:
        kw["host"] = [int(x) for x in rng.choice(pool, size=min(len(pool), 2), replace=False)]
    if rng.random() < 0.25:
        kw["exclude"] = [int(x) for x in rng.choice(pool, size=min(len(pool), 2), replace=False)]
    a = ["-k", str(kw["k"]), "--min-hitlen", str(kw["min_hitlen"]), "--classification-rank", kw["rank"]]
    if not kw["traverse"]:
        a.append("--no-traverse")
    if kw.get("host"):
        a += ["--host-taxids", ",".join(map(str, kw["host"]))]
    if kw.get("exclude"):
        a += ["--exclude-taxids", ",".join(map(str, kw["exclude"]))]
    base = os.path.join(d, "idx")
    want = O.ref_classify(base, os.path.join(d, "w.tsv"), os.path.join(d, "w.rep"), extra=a, u=os.path.join(d, "r.fa"))
    names, ql, seq, off, seeds, pr = reads.load([os.path.join(d, "r.fa")], False)
    e = emu.Emu(base)
    emu.lib().emu_set_search_version(2)
    try:
        for fp, fs in ((1, 1), (0, 0)):                       # the common-case kernels in front, then the general ones alone
            emu.lib().emu_set_fast_kernels(fp, fs)
            rows, n_rows, s2 = e.classify(seq, off, seeds, paired=False, **kw)
            got = reads.format_tsv(e.seqid, names, ql, rows, n_rows, s2)
            assert got == want, (kw, common.first_diff(got, want))
    finally:
        emu.lib().emu_set_fast_kernels(1, 1)
    hix = capi.Index(base, host_only=True)
    rep = capi.Report(hix)
    rep.add(rows, n_rows, TR.max_scores(O.Oracle(base), seq, off, pr), kw["k"])
    rep.write(os.path.join(d, "m.rep"))
    rep.close(); hix.close(); e.close()
    mine, ref = open(os.path.join(d, "m.rep")).read(), open(os.path.join(d, "w.rep")).read()
    assert mine == ref, common.first_diff(mine, ref)


def test_request_budget_on_a_model_of_config_2(tmp_path):
    """The search kernel is bound by random requests per second, so requests per read IS its cost — and the
    emulator counts them exactly.  A 16 Mbp model of the config-2 stand-in (32 genomes in genera of 8 at 5 %, the bench's read
    recipe, every derived table, K = 12 so that a wide-ftab range holds about as many rows as on the 8.6 Gbp index) costs
    29.9 requests per read at the end of round 3 (the GPU on the real thing: 30.5): a change of a kernel body that adds requests
    shows here, without a GPU.  Rows against the reference as everywhere."""
    import sys
    from oracle import oracle as O
    from centrifuge_amd import capi
    if not O.have_ref():
        pytest.skip("oracle/_ref (the compiled reference) is not built")
    sys.path.insert(0, os.path.join(common.ROOT, "tools"))
    import synth
    d = str(tmp_path)
    g = synth.make_genomes(32, 500000, genus_size=8, divergence=0.05, seed=12345)
    synth.write_reference(d, g, genus_size=8)
    O.ref_build(d, threads=4)
    nm, s = synth.sample_reads(g, 2000, 100, seed=777)
    synth.write_fasta(os.path.join(d, "r.fa"), nm, s)
    base = os.path.join(d, "idx")
    want = O.ref_classify(base, os.path.join(d, "w.tsv"), os.path.join(d, "w.rep"), u=os.path.join(d, "r.fa"), threads=4)
    e, L = emu.Emu(base), emu.lib()
    L.emu_set_search_version(2)
    L.emu_textify(e.h, 1); L.emu_planify(e.h, 1); L.emu_planify2(e.h, 1); L.emu_set_self_records(1)
    L.emu_widen(e.h, 12); L.emu_densify(e.h, 0)
    names, ql, seq, off, seeds, pr = reads.load([os.path.join(d, "r.fa")], False)
    ops = capi.OpCounts()
    rows, n_rows, s2 = e.classify(seq, off, seeds, paired=False, ops=ops)
    got = reads.format_tsv(e.seqid, names, ql, rows, n_rows, s2)
    assert got == want, common.first_diff(got, want)
    n = float(len(names))
    requests = (ops.n_ftab_wide + ops.n_ftab + ops.n_pair + ops.n_pair2 + ops.n_single + 2 * ops.n_verify + ops.n_text_loads) / n
    assert ops.n_walk == 0 and ops.n_ftab / n < 0.01                 # resolve table at every row; calls start from the wide ftab
    assert 15.0 < requests <= 25.0, requests        # (main: 29.9; this branch: 24.4)
    e.close()


def test_counters_with_many_taxa(tmp_path):
    """60,000 sequences, each its own species: far more taxa than k_count has LDS slots (the open hash), and one chunk of
    queries touches more of them than there are slots (the far atomics beside it); rows and counters against the reference"""
    import sys
    from oracle import oracle as O
    if not O.have_ref():
        pytest.skip("oracle/_ref (the compiled reference) is not built")
    sys.path.insert(0, os.path.join(common.ROOT, "tools"))
    import synth
    rng = np.random.default_rng(3)
    n, L = 60000, 160
    g = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, (n, L), dtype=np.uint8)]
    d = str(tmp_path)
    synth.write_reference(d, g, genus_size=8, uid_prefix="seq")
    O.ref_build(d, threads=4)
    which, at = rng.integers(0, n, 12000), rng.integers(0, L - 100, 12000)
    with open(os.path.join(d, "reads.fa"), "w") as f:
        for k in range(len(which)):
            f.write(">r%d\n%s\n" % (k, g[which[k], at[k]:at[k] + 100].tobytes().decode()))
    want = O.ref_classify(os.path.join(d, "idx"), os.path.join(d, "ref.tsv"), os.path.join(d, "ref.rep"), u=os.path.join(d, "reads.fa"), threads=4)
    names, qlens, seq, off, seeds, paired = reads.load([os.path.join(d, "reads.fa")], False)
    emu.lib().emu_set_search_version(2)
    e = emu.Emu(os.path.join(d, "idx"))
    ntax = e.L.emu_num_taxa(e.h)
    assert ntax > 8 * 4096
    rows, n_rows, score2, cnt = e.classify(seq, off, seeds, paired=False, counts=True)
    got = reads.format_tsv(e.seqid, names, qlens, rows, n_rows, score2)
    assert got == want, common.first_diff(got, want)
    mine = {}
    for i in np.nonzero(cnt[:ntax])[0]:
        t = e.L.emu_taxon_id(e.h, int(i))
        if t != 0:
            mine[t] = (int(cnt[i]), int(cnt[ntax + i]))
    rep = {}
    for ln in open(os.path.join(d, "ref.rep")).read().splitlines()[1:]:
        f = ln.split("\t")
        rep[int(f[1])] = (int(f[4]), int(f[5]))
    assert len(rep) > 4096 and mine == rep
    e.close()


@pytest.mark.parametrize("lengths,paired,k", common.EDGE_CASES)
def test_edge_batches_match_oracle_on_cpu(lengths, paired, k):
    """the boundary-length / degenerate batches of the GPU edge test, through the CPU single-step harness"""
    from oracle import oracle as O
    emu.lib().emu_set_search_version(2)
    d, _ = common.golden("synth_small")
    orc = O.Oracle(os.path.join(d, "idx"))
    e = emu.Emu(os.path.join(d, "idx"))
    recs = reads.read_fasta(os.path.join(d, "reads.fa")) + reads.read_fasta(os.path.join(d, "reads250.fa"))
    rng = np.random.default_rng(11)
    rs = common.edge_reads(recs, lengths, rng)
    if paired and len(rs) % 2:
        rs.append(rs[0])
    seq, off = orc.pack(rs)
    seeds = rng.integers(0, 2 ** 32, size=len(rs), dtype=np.uint32)
    nq = len(rs) // 2 if paired else len(rs)
    want = orc.classify(seq, off, seeds, nq, paired, orc.params(k=k))
    got = e.classify(seq, off, seeds, paired=paired, k=k)
    assert np.array_equal(got[1], want[1]) and np.array_equal(got[2], want[2])
    for q in range(nq):
        for r in range(int(want[1][q])):
            g, w = got[0][q, r], want[0][q, r]
            assert (int(g["tax_id"]), int(g["unique_id"]), int(g["score"]), int(g["hit_len"])) == \
                   (int(w["tax_id"]), int(w["unique_id"]), int(w["score"]), int(w["hit_len"])), (q, r)


@pytest.mark.parametrize("cap", [1, 2, 5, 17, 1000])
def test_row_stage_in_several_passes(cap):
    """A batch that plans more SA rows than the row workspace holds is finished in several passes of
    window -> emit -> walk -> score (row_window_body): the rows and the per-taxon counters do not depend on the
    capacity, down to one row per pass (a query larger than the workspace makes the workspace grow to it)."""
    try:
        emu.lib().emu_set_rows_cap(0)
        d, c, e, want, cnt_want = run_case("synth_small", "k5")
        emu.lib().emu_set_rows_cap(cap)
        _, _, _, got, cnt = run_case("synth_small", "k5")
        assert got == want and np.array_equal(cnt, cnt_want)
        assert got == open(os.path.join(d, c["tsv"])).read()
        _, c2, _, got2, _ = run_case("synth_small", "pe_k1")
        assert got2 == open(os.path.join(d, c2["tsv"])).read()
    finally:
        emu.lib().emu_set_rows_cap(0)


@pytest.mark.parametrize("rate", [0, 1, 2, 3])
@pytest.mark.parametrize("arch,name", [("synth_small", "k5"), ("synth_small", "pe_k1"), ("synth_small", "r250_k5"), ("example", "default")])
def test_dense_resolve_table_gives_the_same_rows(arch, name, rate):
    """The walk stops at the first row of a table 2^(4 - rate) times denser than the file's SA sample, filled at load time by
    the full walk itself (walk2_body's table mode): same reference indexes, so same rows and counters, fewer LF steps."""
    from centrifuge_amd import capi
    emu.lib().emu_set_search_version(2)
    d, cases = common.golden(arch)
    c = [x for x in cases if x["name"] == name][0]
    kw, fastq = common.case_kwargs(c["args"])
    e = emu.Emu(os.path.join(d, "idx"))
    names, qlens, seq, off, seeds, paired = reads.load([os.path.join(d, f) for f in c["reads"]], fastq)
    base_ops, ops = capi.OpCounts(), capi.OpCounts()
    e.classify(seq, off, seeds, paired=paired, ops=base_ops, **kw)
    assert emu.lib().emu_densify(e.h, rate) == 1
    rows, n_rows, score2, cnt = e.classify(seq, off, seeds, paired=paired, counts=True, ops=ops, **kw)
    got = reads.format_tsv(e.seqid, names, qlens, rows, n_rows, score2)
    assert got == open(os.path.join(d, c["tsv"])).read()
    assert ops.n_rows == base_ops.n_rows and ops.n_walk <= base_ops.n_walk
    if base_ops.n_walk > 200:
        assert ops.n_walk < base_ops.n_walk * (2 ** rate) / 8       # ~(2^rate - 1) / 15 of the steps
    # every row of the index resolves to what the plain walk gives
    e2 = emu.Emu(os.path.join(d, "idx"))
    rng = np.random.default_rng(3)
    rows_t = np.arange(0, 1074) if arch == "example" else rng.integers(0, 140000, 3000)
    emu.lib().emu_resolve_walk.restype = C.c_uint32
    emu.lib().emu_resolve_walk.argtypes = [C.c_void_p, C.c_uint64]
    for r in rows_t:
        assert emu.lib().emu_resolve_walk(e.h, int(r)) == emu.lib().emu_resolve(e2.h, int(r))


@pytest.mark.parametrize("k,cap", [(11, 0), (12, 0), (13, 0), (13, 3)])
@pytest.mark.parametrize("arch,name", [("synth_small", "k5"), ("synth_small", "pe_k1"), ("synth_small", "r250_k5"), ("synth_small", "minhit15"),
                                       ("synth_small", "fastq"), ("example", "default")])
def test_wide_ftab_gives_the_same_rows(arch, name, k, cap):
    """partialSearch calls started from the wide ftab (one lookup = the range at the deepest of the first k bases at which it is
    non-empty: either the search goes on from depth k, or the call ends right there with the hit the step-by-step path would
    end with): same hits, so same rows; fewer LF steps and no 10-mer lookups.  cap: ranges of that many rows or more are
    stored as "does not fit" and take the step-by-step path (the 20-bit size field of an entry, lowered to reach it)"""
    from centrifuge_amd import capi
    emu.lib().emu_set_search_version(2)
    d, cases = common.golden(arch)
    c = [x for x in cases if x["name"] == name][0]
    kw, fastq = common.case_kwargs(c["args"])
    e = emu.Emu(os.path.join(d, "idx"))
    names, qlens, seq, off, seeds, paired = reads.load([os.path.join(d, f) for f in c["reads"]], fastq)
    base_ops, ops = capi.OpCounts(), capi.OpCounts()
    e.classify(seq, off, seeds, paired=paired, ops=base_ops, **kw)
    emu.lib().emu_set_wide_cap(cap if cap else (1 << 20) - 1)
    try:
        assert emu.lib().emu_widen(e.h, k) == 1
    finally:
        emu.lib().emu_set_wide_cap((1 << 20) - 1)
    rows, n_rows, score2, cnt = e.classify(seq, off, seeds, paired=paired, counts=True, ops=ops, **kw)
    got = reads.format_tsv(e.seqid, names, qlens, rows, n_rows, score2)
    assert got == open(os.path.join(d, c["tsv"])).read()
    assert base_ops.n_ftab_wide == 0 and ops.n_ftab_wide > 0
    assert ops.n_pair + ops.n_single < base_ops.n_pair + base_ops.n_single
    if cap and arch == "synth_small":
        assert ops.n_ftab > 0                      # the capped entries fell back to the 10-mer table
    elif arch == "synth_small":
        assert ops.n_ftab < base_ops.n_ftab / 10   # only calls with fewer than k N-free bases left still use it
    # the search tap (hit lists after extend / twin / trim) is the same, read by read
    if arch == "synth_small" and name == "k5":
        e0 = emu.Emu(os.path.join(d, "idx"))
        for r in range(0, len(names), 37):
            s_ = seq[int(off[r]):int(off[r + 1])]
            a, b = e.search(s_), e0.search(s_)
            assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])


@pytest.mark.parametrize("rate", [0, 2, 3, 5])
@pytest.mark.parametrize("arch,name", [("synth_small", "k5"), ("synth_small", "pe_k1"), ("synth_small", "r250_k5"), ("synth_small", "minhit15"),
                                       ("synth_small", "fastq"), ("synth_small", "k50"), ("example", "default")])
def test_text_verification_gives_the_same_rows(arch, name, rate):
    """unique matches verified against the 2-bit text (S_POS / S_TXT / S_ISA of search2_body, tables from the inverse-BWT
    walks): same hits — rows, lengths, offsets — so same output; far fewer single-row LF steps"""
    from centrifuge_amd import capi
    emu.lib().emu_set_search_version(2)
    d, cases = common.golden(arch)
    c = [x for x in cases if x["name"] == name][0]
    kw, fastq = common.case_kwargs(c["args"])
    e = emu.Emu(os.path.join(d, "idx"))
    names, qlens, seq, off, seeds, paired = reads.load([os.path.join(d, f) for f in c["reads"]], fastq)
    base_ops, ops = capi.OpCounts(), capi.OpCounts()
    e.classify(seq, off, seeds, paired=paired, ops=base_ops, **kw)
    assert emu.lib().emu_textify(e.h, rate) == 1
    rows, n_rows, score2, cnt = e.classify(seq, off, seeds, paired=paired, counts=True, ops=ops, **kw)
    got = reads.format_tsv(e.seqid, names, qlens, rows, n_rows, score2)
    assert got == open(os.path.join(d, c["tsv"])).read()
    assert base_ops.n_verify == 0
    if arch == "synth_small":
        assert ops.n_verify > 0 and ops.n_single < base_ops.n_single / (2 if rate <= 3 else 1)   # (every 32nd row: 100-base reads rarely reach one in time)
    # the search tap (top, bot, bwoff, len of every hit after extend / twin / trim) is the same, read by read
    e0 = emu.Emu(os.path.join(d, "idx"))
    for r in range(0, len(names), 29 if arch == "synth_small" else 1):
        s_ = seq[int(off[r]):int(off[r + 1])]
        a, b = e.search(s_), e0.search(s_)
        assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]), r
    # together with the other derived tables
    assert emu.lib().emu_widen(e.h, 11) == 1 and emu.lib().emu_densify(e.h, 2) == 1
    rows, n_rows, score2 = e.classify(seq, off, seeds, paired=paired, **kw)
    assert reads.format_tsv(e.seqid, names, qlens, rows, n_rows, score2) == got


@pytest.mark.parametrize("arch,name", common.all_cases())
def test_occ_planes_give_the_same_rows(arch, name):
    """LF steps of the search over the occurrence planes (per 64 rows and character: 64 match bits + the LF base; one chain
    per lane, one 16-byte load per step) instead of the sides: same ranks, so same hits and rows; same number of steps"""
    from centrifuge_amd import capi
    emu.lib().emu_set_search_version(2)
    d, cases = common.golden(arch)
    c = [x for x in cases if x["name"] == name][0]
    kw, fastq = common.case_kwargs(c["args"])
    e = emu.Emu(os.path.join(d, "idx"))
    names, qlens, seq, off, seeds, paired = reads.load([os.path.join(d, f) for f in c["reads"]], fastq)
    base_ops, ops = capi.OpCounts(), capi.OpCounts()
    e.classify(seq, off, seeds, paired=paired, ops=base_ops, **kw)
    assert emu.lib().emu_planify(e.h, 1) == 1
    rows, n_rows, score2 = e.classify(seq, off, seeds, paired=paired, ops=ops, **kw)
    assert reads.format_tsv(e.seqid, names, qlens, rows, n_rows, score2) == open(os.path.join(d, c["tsv"])).read()
    assert (ops.n_pair, ops.n_single, ops.n_ftab) == (base_ops.n_pair, base_ops.n_single, base_ops.n_ftab)
    if base_ops.n_pair2:
        assert ops.n_pair2 > base_ops.n_pair2          # a range straddles a 64-row group more often than a 384-char side
    e0 = emu.Emu(os.path.join(d, "idx"))
    for r in range(0, len(names), 41):
        s_ = seq[int(off[r]):int(off[r + 1])]
        a, b = e.search(s_), e0.search(s_)
        assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]), r


@pytest.mark.parametrize("arch,name", common.all_cases())
def test_all_derived_tables_together_on_every_golden_case(arch, name):
    """wide ftab + dense resolve table + text verification, as a device index has them by default: every golden case"""
    emu.lib().emu_set_search_version(2)
    d, cases = common.golden(arch)
    c = [x for x in cases if x["name"] == name][0]
    kw, fastq = common.case_kwargs(c["args"])
    e = emu.Emu(os.path.join(d, "idx"))
    assert emu.lib().emu_textify(e.h, 2) == 1 and emu.lib().emu_widen(e.h, 12) == 1 and emu.lib().emu_densify(e.h, 2) == 1
    assert emu.lib().emu_planify(e.h, 1) == 1
    names, qlens, seq, off, seeds, paired = reads.load([os.path.join(d, f) for f in c["reads"]], fastq)
    rows, n_rows, score2 = e.classify(seq, off, seeds, paired=paired, **kw)
    got = reads.format_tsv(e.seqid, names, qlens, rows, n_rows, score2)
    assert got == open(os.path.join(d, c["tsv"])).read()


@pytest.mark.parametrize("arch,name", common.all_cases())
def test_forward_words_from_the_plan_and_from_the_kernel_agree(arch, name):
    """round 6: the plan (plan_fill_body / rev_word) leaves the forward strands' words in search order beside the packed reads, and
    the one-lane search kernel's S_REC2 state loads them instead of turning the read round itself (reads with an N keep the
    in-kernel transform).  Every golden case both ways: the same rows, the same hits on a sample of the reads, the same steps."""
    from centrifuge_amd import capi
    L = emu.lib()
    L.emu_set_search_version(2)
    d, cases = common.golden(arch)
    c = [x for x in cases if x["name"] == name][0]
    kw, fastq = common.case_kwargs(c["args"])
    e = emu.Emu(os.path.join(d, "idx"))
    assert L.emu_planify(e.h, 1) == 1 and L.emu_widen(e.h, 12) == 1
    names, qlens, seq, off, seeds, paired = reads.load([os.path.join(d, f) for f in c["reads"]], fastq)
    want = open(os.path.join(d, c["tsv"])).read()
    ops = {}
    try:
        for rev in (1, 0):
            L.emu_set_rev_words(rev)
            ops[rev] = capi.OpCounts()
            rows, n_rows, score2 = e.classify(seq, off, seeds, paired=paired, ops=ops[rev], **kw)
            assert reads.format_tsv(e.seqid, names, qlens, rows, n_rows, score2) == want, rev
        for f in ("n_pair", "n_pair2", "n_single", "n_ftab", "n_ftab_wide"):
            assert getattr(ops[1], f) == getattr(ops[0], f), f
    finally:
        L.emu_set_rev_words(1)
        e.close()


@pytest.mark.parametrize("text_rate,dense_rate,isa_extra", [(1, 0, 0), (2, 2, 3), (4, 3, 2), (3, -1, 3), (0, 1, 3)])
@pytest.mark.parametrize("arch,name", [("synth_small", "k5"), ("synth_small", "pe_k1"), ("synth_small", "r250_k5"), ("synth_small", "minhit15"), ("synth_small", "k1"), ("example", "default")])
def test_position_form_with_sampled_text_tables(arch, name, text_rate, dense_rate, isa_extra):
    """... and where the SA / inverse-SA samples are not at every row (the larger presets): the search saves the inverse-sample request
    AND the steps back from the sampled position; the resolver's way by the row takes those steps itself (forced steps of the walk);
    the bound on the walk-left is the longest segment of the inverse-BWT walks unless the resolve table holds every row.  isa_extra:
    the inverse sample that many steps coarser than the SA sample (DIndex::isaRate: hardly anything reads it once the hits take the
    form, so the device layer keeps it three steps coarser and the planner spends the room on the SA sample)"""
    from centrifuge_amd import capi
    L = emu.lib()
    L.emu_set_search_version(2)
    d, cases = common.golden(arch)
    c = [x for x in cases if x["name"] == name][0]
    kw, fastq = common.case_kwargs(c["args"])
    e = emu.Emu(os.path.join(d, "idx"))
    names, qlens, seq, off, seeds, paired = reads.load([os.path.join(d, f) for f in c["reads"]], fastq)
    want = open(os.path.join(d, c["tsv"])).read()
    try:
        L.emu_set_isa_extra(isa_extra)
        assert L.emu_planify(e.h, 1) == 1 and L.emu_widen(e.h, 12) == 1 and L.emu_textify(e.h, text_rate) == 1
        if dense_rate >= 0:
            assert L.emu_densify(e.h, dense_rate) == 1
        ops = {}
        for on in (0, 1):
            L.emu_set_pos_shift(5)
            assert L.emu_posify(e.h, on) == on
            ops[on] = capi.OpCounts()
            for fast in ((1, 1), (0, 0)):
                L.emu_set_fast_kernels(*fast)
                rows, n_rows, score2 = e.classify(seq, off, seeds, paired=paired, ops=ops[on], **kw)
                assert reads.format_tsv(e.seqid, names, qlens, rows, n_rows, score2) == want, (on, fast)
        assert ops[0].n_pos_hits == 0 and ops[1].n_single <= ops[0].n_single
        if ops[0].n_verify > 20:
            assert ops[1].n_pos_hits > 0
    finally:
        L.emu_set_pos_shift(14); L.emu_set_fast_kernels(1, 1); L.emu_set_isa_extra(0)
        e.close()


@pytest.mark.parametrize("shift", [14, 6, 3])
@pytest.mark.parametrize("arch,name", common.all_cases())
def test_hits_in_the_position_form_give_the_same_rows(arch, name, shift):
    """round 6: a unique match that ends in the text goes out as {text position} instead of {suffix-array row} — the inverse-sample
    request is not made — and the resolver answers from the position (resolve_pos: the sequence that holds it wherever the
    walk-left cannot leave that sequence, which the longest walk of the table build bounds exactly; else the row from the inverse
    sample and its walk).  Every golden case with and without the form: the same rows, the same counters; with it, hits do take
    the form and requests go down by exactly their number.  Small buckets (shift 6, 3) put several fragments into one bucket and
    leave buckets without any; the golden indexes' sequences are a few kilobases, so both branches of resolve_pos are taken."""
    from centrifuge_amd import capi
    L = emu.lib()
    L.emu_set_search_version(2)
    d, cases = common.golden(arch)
    c = [x for x in cases if x["name"] == name][0]
    kw, fastq = common.case_kwargs(c["args"])
    e = emu.Emu(os.path.join(d, "idx"))
    names, qlens, seq, off, seeds, paired = reads.load([os.path.join(d, f) for f in c["reads"]], fastq)
    want = open(os.path.join(d, c["tsv"])).read()
    try:
        L.emu_set_isa_extra(3 if shift == 6 else 0)        # (the inverse sample at every 8th position, as the device layer keeps it, or at every one)
        assert L.emu_planify(e.h, 1) == 1 and L.emu_planify2(e.h, 1) == 1 and L.emu_widen(e.h, 12) == 1
        assert L.emu_textify(e.h, 0) == 1 and L.emu_densify(e.h, 0) == 1
        ops = {}
        cnts = {}
        for on in (0, 1):
            L.emu_set_pos_shift(shift)
            assert L.emu_posify(e.h, on) == on
            ops[on] = capi.OpCounts()
            rows, n_rows, score2, cnt = e.classify(seq, off, seeds, paired=paired, ops=ops[on], counts=True, **kw)
            assert reads.format_tsv(e.seqid, names, qlens, rows, n_rows, score2) == want, on
            cnts[on] = cnt
        assert np.array_equal(cnts[0], cnts[1])
        assert ops[0].n_pos_hits == 0
        for f in ("n_pair", "n_pair2", "n_ftab", "n_ftab_wide", "n_verify"):
            assert getattr(ops[1], f) == getattr(ops[0], f), f
        assert ops[1].n_single <= ops[0].n_single          # (a match of a base or two that ends in the window is not stepped out either)
        if max(qlens) <= 256 and ops[0].n_verify > 20:
            assert ops[1].n_pos_hits > ops[0].n_verify // 4, (ops[1].n_pos_hits, ops[0].n_verify)
        assert L.emu_walk_max(e.h) >= 1
        # ... and through the general kernels alone (post_body / emit / walk / score_body read the hits from the pool)
        L.emu_set_fast_kernels(0, 0)
        rows, n_rows, score2 = e.classify(seq, off, seeds, paired=paired, **kw)
        assert reads.format_tsv(e.seqid, names, qlens, rows, n_rows, score2) == want
        L.emu_set_direct_refs(0)                          # rows emitted and walked (k_emit / k_walk3), common-case kernels back on
        L.emu_set_fast_kernels(1, 1)
        rows, n_rows, score2 = e.classify(seq, off, seeds, paired=paired, **kw)
        assert reads.format_tsv(e.seqid, names, qlens, rows, n_rows, score2) == want
    finally:
        L.emu_set_pos_shift(14); L.emu_set_fast_kernels(1, 1); L.emu_set_direct_refs(1); L.emu_set_isa_extra(0)
        e.close()


@pytest.mark.parametrize("lengths,paired,k", common.EDGE_CASES)
def test_edge_batches_with_text_verification(lengths, paired, k):
    from oracle import oracle as O
    emu.lib().emu_set_search_version(2)
    d, _ = common.golden("synth_small")
    orc = O.Oracle(os.path.join(d, "idx"))
    e = emu.Emu(os.path.join(d, "idx"))
    assert emu.lib().emu_textify(e.h, 1) == 1
    recs = reads.read_fasta(os.path.join(d, "reads.fa")) + reads.read_fasta(os.path.join(d, "reads250.fa"))
    rng = np.random.default_rng(11)
    rs = common.edge_reads(recs, lengths, rng)
    if paired and len(rs) % 2:
        rs.append(rs[0])
    seq, off = orc.pack(rs)
    seeds = rng.integers(0, 2 ** 32, size=len(rs), dtype=np.uint32)
    nq = len(rs) // 2 if paired else len(rs)
    want = orc.classify(seq, off, seeds, nq, paired, orc.params(k=k))
    got = e.classify(seq, off, seeds, paired=paired, k=k)
    assert np.array_equal(got[1], want[1]) and np.array_equal(got[2], want[2])
    for q in range(nq):
        for r in range(int(want[1][q])):
            g, w = got[0][q, r], want[0][q, r]
            assert (int(g["tax_id"]), int(g["unique_id"]), int(g["score"]), int(g["hit_len"])) == \
                   (int(w["tax_id"]), int(w["unique_id"]), int(w["score"]), int(w["hit_len"])), (q, r)


@pytest.mark.parametrize("planes", [0, 1, 2])
@pytest.mark.parametrize("paired,k,minhit", [(False, 5, 22), (True, 1, 22), (False, 5, 15), (False, 2, 30)])
def test_lazy_hits_give_the_same_rows_on_reads_that_turn_long_late(planes, paired, k, minhit):
    """the search kernel holds a strand's hits back until one reaches minHitLen (the first two wait in LDS, later ones are let
    go, a strand that turns long after that is searched again): reads with several substitutions near their right end — short
    hits first, the long one late — on both strands, plus N runs and the golden reads, against the oracle; the hit pool
    starts out poisoned, so a hit that was not stored but is read shows"""
    from oracle import oracle as O
    emu.lib().emu_set_search_version(2)
    d, _ = common.golden("synth_small")
    orc = O.Oracle(os.path.join(d, "idx"))
    e = emu.Emu(os.path.join(d, "idx"))
    assert emu.lib().emu_planify(e.h, min(planes, 1)) == 1 and emu.lib().emu_widen(e.h, 12) == 1 and emu.lib().emu_textify(e.h, 1) == 1
    if planes == 2:                                      # two bases per LF request over the pair planes
        assert emu.lib().emu_planify2(e.h, 1) == 1
    recs = reads.read_fasta(os.path.join(d, "reads.fa"))[:400] + reads.read_fasta(os.path.join(d, "reads250.fa"))[:150]
    rng = np.random.default_rng(3)
    rs = []
    for i, (_, codes, _) in enumerate(recs):
        c = np.array(codes, dtype=np.uint8).copy()
        L = len(c)
        if L < 60:
            rs.append(c); continue
        end = i % 2 == 0                                   # substitutions near the right end (forward strand searched from
        nsub = 2 + i % 4                                   # there) or near the left end (its reverse complement is)
        gaps = rng.integers(7, 19, size=nsub)
        pos = np.cumsum(gaps)
        pos = pos[pos < L - 30]
        for p_ in pos:
            q = L - 1 - int(p_) if end else int(p_)
            c[q] = (c[q] + 1 + i % 3) & 3 if c[q] < 4 else 0
        if i % 11 == 0:
            c[L // 2] = 4
        rs.append(c)
    if paired and len(rs) % 2:
        rs.append(rs[0])
    seq, off = orc.pack(rs)
    seeds = rng.integers(0, 2 ** 32, size=len(rs), dtype=np.uint32)
    nq = len(rs) // 2 if paired else len(rs)
    want = orc.classify(seq, off, seeds, nq, paired, orc.params(k=k, min_hitlen=minhit))
    for lazy in (1, 0):
        emu.lib().emu_set_lazy_hits(lazy)
        try:
            got = e.classify(seq, off, seeds, paired=paired, k=k, min_hitlen=minhit)
        finally:
            emu.lib().emu_set_lazy_hits(1)
        assert np.array_equal(got[1], want[1]) and np.array_equal(got[2], want[2]), lazy
        for q in range(nq):
            for r in range(int(want[1][q])):
                g, w = got[0][q, r], want[0][q, r]
                assert (int(g["tax_id"]), int(g["unique_id"]), int(g["score"]), int(g["hit_len"])) == \
                       (int(w["tax_id"]), int(w["unique_id"]), int(w["score"]), int(w["hit_len"])), (lazy, q, r)
    assert int(want[1].sum()) > nq // 2                    # most of these reads still classify


@pytest.mark.parametrize("post_hits,score_rows", [(0, 0), (3, 2), (9, 5), (64, 64)])
@pytest.mark.parametrize("arch,name", common.all_cases())
def test_general_kernels_in_their_lanes_scratch_and_in_place_agree(arch, name, post_hits, score_rows):
    """round 6: the general post / score kernels work on a lane's own scratch (LDS on the device) — a mate's hit lists; a query's
    hit map, parent counts and references — whenever they fit it, in place otherwise.  Every golden case through the general
    kernels alone (the common-case ones off) with no scratch, with scratches so small that most mates and queries do not fit
    (both paths inside one batch), and with room for everything: the same rows, the same counters."""
    L = emu.lib()
    L.emu_set_general_scratch.argtypes = [C.c_uint32, C.c_uint32]
    try:
        L.emu_set_fast_kernels(0, 0)
        L.emu_set_general_scratch(post_hits, score_rows)
        d, c, e, got, cnt = run_case(arch, name)
    finally:
        L.emu_set_fast_kernels(1, 1)
        L.emu_set_general_scratch(24, 16)
    ref = open(os.path.join(d, c["tsv"])).read()
    assert got == ref, common.first_diff(got, ref)
    d2, c2, e2, got2, cnt2 = run_case(arch, name)
    assert np.array_equal(cnt, cnt2)


@pytest.mark.parametrize("post_fast,score_fast", [(0, 0), (1, 0), (0, 1)])
@pytest.mark.parametrize("arch,name", common.all_cases())
def test_common_case_kernels_and_general_kernels_agree(arch, name, post_fast, score_fast):
    """post_fast_body / score_fast_body (registers only) in front of the general kernels is what runs by default (the tests above);
    here the general kernels alone, and each common-case kernel with the other stage general: same rows, same counters — and
    with both on, the common-case kernels really take most queries of a plain case."""
    L = emu.lib()
    try:
        L.emu_set_fast_kernels(post_fast, score_fast)
        d, c, e, got, cnt = run_case(arch, name)
    finally:
        L.emu_set_fast_kernels(1, 1)
    ref = open(os.path.join(d, c["tsv"])).read()
    assert got == ref, common.first_diff(got, ref)
    d2, c2, e2, got2, cnt2 = run_case(arch, name)
    assert np.array_equal(cnt, cnt2)
    sp, ss = C.c_uint32(), C.c_uint32()
    L.emu_last_slow(C.byref(sp), C.byref(ss))
    nq = got2.count("\n") - 1
    if name in ("default", "k5") and nq >= 50:
        assert sp.value < nq // 2 and ss.value < nq // 2, (sp.value, ss.value, nq)


def _chimeras(recs, rng, n):
    """reads made of a genome piece and the reverse complement of another (or the same) one, overlapping by a few bases or not:
    BOTH strands carry long hits, so the query takes the general post kernel — cross-strand extension through ps_whole,
    twin removal, trim"""
    pool = [c for _, c, _ in recs if len(c) >= 100]

    def rc(x):
        y = x[::-1].copy()
        y[y < 4] = 3 - y[y < 4]
        return y
    out = []
    for i in range(n):
        a, b_ = pool[int(rng.integers(0, len(pool)))], pool[int(rng.integers(0, len(pool)))]
        la, lb = int(rng.integers(23, 90)), int(rng.integers(23, 90))
        sa, sb = int(rng.integers(0, len(a) - la + 1)), int(rng.integers(0, len(b_) - lb + 1))
        left, right = a[sa:sa + la].copy(), rc(b_[sb:sb + lb])
        if i % 3 == 0 and len(left) > 30:                     # palindromic junction: the two pieces share bases
            k = int(rng.integers(1, 12))
            right = np.concatenate([rc(left[-k:]), right])
        r = np.concatenate([left, right])
        for _ in range(int(rng.integers(0, 3))):
            r[int(rng.integers(0, len(r)))] = rng.integers(0, 4)
        out.append(r[:250])
    return out


@pytest.mark.parametrize("planes,wide,text,dense", [(0, 0, -1, -1), (1, 0, -1, -1), (0, 12, -1, 1), (0, 0, 0, -1), (1, 12, 1, 0), (1, 13, 2, 2), (0, 11, 3, -1), (2, 0, -1, -1), (2, 12, 1, 0)])
def test_reads_with_hits_on_both_strands_with_every_table(planes, wide, text, dense):
    from oracle import oracle as O
    emu.lib().emu_set_search_version(2)
    d, _ = common.golden("synth_small")
    orc = O.Oracle(os.path.join(d, "idx"))
    e = emu.Emu(os.path.join(d, "idx"))
    if planes:
        assert emu.lib().emu_planify(e.h, 1) == 1
    if planes == 2:
        assert emu.lib().emu_planify2(e.h, 1) == 1
    if wide:
        assert emu.lib().emu_widen(e.h, wide) == 1
    if text >= 0:
        assert emu.lib().emu_textify(e.h, text) == 1
    if dense >= 0:
        assert emu.lib().emu_densify(e.h, dense) == 1
    recs = reads.read_fasta(os.path.join(d, "reads.fa")) + reads.read_fasta(os.path.join(d, "reads250.fa"))
    rng = np.random.default_rng(1000 * planes + 10 * wide + text + 5)
    rs = _chimeras(recs, rng, 300)
    seq, off = orc.pack(rs)
    seeds = rng.integers(0, 2 ** 32, size=len(rs), dtype=np.uint32)
    want = orc.classify(seq, off, seeds, len(rs), False, orc.params(k=5))
    got = e.classify(seq, off, seeds, paired=False, k=5)
    assert np.array_equal(got[1], want[1]) and np.array_equal(got[2], want[2])
    for q in range(len(rs)):
        for r in range(int(want[1][q])):
            g, w = got[0][q, r], want[0][q, r]
            assert (int(g["tax_id"]), int(g["unique_id"]), int(g["score"]), int(g["hit_len"])) == \
                   (int(w["tax_id"]), int(w["unique_id"]), int(w["score"]), int(w["hit_len"])), (q, r)
    sp, ss = C.c_uint32(), C.c_uint32()
    emu.lib().emu_last_slow(C.byref(sp), C.byref(ss))
    assert sp.value > len(rs) // 4                        # they really went through the general post kernel


@pytest.mark.parametrize("wide,text", [(0, -1), (12, 1), (0, 0), (13, 3)])
@pytest.mark.parametrize("arch,name", common.all_cases())
def test_pair_planes_give_the_same_rows_with_fewer_requests(arch, name, wide, text):
    """two bases per LF request over the pair planes (entry 4 c1 + c0 of a 64-row group: the rows preceded by c1 c0 + LF(c0, LF(c1,
    group start))): an empty pair falls back to its first base alone, after which the call ends.  Same rows on every golden
    case — alone and with the wide ftab / text verification — and fewer requests where ranges live long"""
    from centrifuge_amd import capi
    emu.lib().emu_set_search_version(2)
    d, cases = common.golden(arch)
    c = [x for x in cases if x["name"] == name][0]
    kw, fastq = common.case_kwargs(c["args"])
    e = emu.Emu(os.path.join(d, "idx"))
    assert emu.lib().emu_planify(e.h, 1) == 1
    if wide:
        assert emu.lib().emu_widen(e.h, wide) == 1
    if text >= 0:
        assert emu.lib().emu_textify(e.h, text) == 1
    names, qlens, seq, off, seeds, paired = reads.load([os.path.join(d, f) for f in c["reads"]], fastq)
    base_ops, ops = capi.OpCounts(), capi.OpCounts()
    e.classify(seq, off, seeds, paired=paired, ops=base_ops, **kw)
    assert emu.lib().emu_planify2(e.h, 1) == 1
    rows, n_rows, score2 = e.classify(seq, off, seeds, paired=paired, ops=ops, **kw)
    assert reads.format_tsv(e.seqid, names, qlens, rows, n_rows, score2) == open(os.path.join(d, c["tsv"])).read()
    if wide == 0 and text < 0 and len(names) > 50:
        assert ops.n_pair + ops.n_single < 0.75 * (base_ops.n_pair + base_ops.n_single)
    e0 = emu.Emu(os.path.join(d, "idx"))
    for r in range(0, len(names), 37):                   # the search tap: hit lists, not only rows
        s_ = seq[int(off[r]):int(off[r + 1])]
        a, b = e.search(s_), e0.search(s_)
        assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]), r


def test_pair_plane_entries_are_two_single_steps():
    """every entry against its definition: LF(c0, LF(c1, row)) from the pair planes = two steps over the sides"""
    d, _ = common.golden("synth_small")
    e = emu.Emu(os.path.join(d, "idx"))
    assert emu.lib().emu_planify(e.h, 1) == 1 and emu.lib().emu_planify2(e.h, 1) == 1
    L = emu.lib()
    L.emu_pair_rank.restype = C.c_uint64
    L.emu_pair_rank.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_uint64]
    n = L.emu_text_len(e.h) if hasattr(L, "emu_text_len") else None
    rng = np.random.default_rng(4)
    import itertools
    nrows = L.emu_num_rows(e.h)
    rows = sorted(set([0, 1, 63, 64, 65, nrows - 1, nrows] + [int(x) for x in rng.integers(0, nrows + 1, 4000)]))
    for row in rows:
        for c1, c0 in itertools.product(range(4), range(4)):
            want = L.emu_rank(e.h, c0, L.emu_rank(e.h, c1, row))
            assert L.emu_pair_rank(e.h, c1, c0, row) == want, (row, c1, c0)


@pytest.mark.parametrize("k", [5, 1])
def test_contigs_and_long_reads_match_the_reference(tmp_path, k):
    """reads of 65,535 to 300,000 bases (24-bit offsets and lengths in the hit records; hit counts of a strand beyond 16 bits;
    max_score beyond 32 bits; time stamps and hit lengths that do not fit an inline plan) through the byte-window search and the
    general post / score kernels, rows and report against the compiled reference"""
    from oracle import oracle as O
    from centrifuge_amd import capi
    import test_report as TR
    if not O.have_ref():
        pytest.skip("oracle/_ref (the compiled reference) is not built")
    d = str(tmp_path)
    base, fa = common.long_read_case(d)
    want = O.ref_classify(base, os.path.join(d, "w.tsv"), os.path.join(d, "w.rep"), u=fa, extra=["-k", str(k)], threads=4)
    names, ql, seq, off, seeds, pr = reads.load([fa], False)
    assert max(ql) == 300000
    e = emu.Emu(base)
    emu.lib().emu_planify(e.h, 1); emu.lib().emu_densify(e.h, 0)
    rows, n_rows, s2 = e.classify(seq, off, seeds, paired=False, k=k)
    got = reads.format_tsv(e.seqid, names, ql, rows, n_rows, s2)
    assert got == want, common.first_diff(got, want)
    hix = capi.Index(base, host_only=True)
    rep = capi.Report(hix)
    rep.add(rows, n_rows, TR.max_scores(O.Oracle(base), seq, off, pr), k)
    rep.write(os.path.join(d, "m.rep"))
    rep.close(); hix.close(); e.close()
    mine, ref = open(os.path.join(d, "m.rep")).read(), open(os.path.join(d, "w.rep")).read()
    assert mine == ref, common.first_diff(mine, ref)


@pytest.mark.parametrize("search_version,walk_version", [(2, 3), (1, 2)])
@pytest.mark.parametrize("arch,name", [("synth_small", "k5"), ("synth_small", "pe_k1"), ("synth_small", "r250_k5"), ("synth_small", "host"), ("example", "default")])
def test_index_without_its_sides_gives_the_same_rows(arch, name, search_version, walk_version):
    """cf_index_options::sides = -1 (what the planner does for the nt-scale index): once the planes exist the BWT sides leave the
    device view — the wide ftab is made over the planes, the byte-window search, the extension step of the general post kernel
    and both walk kernels take their LF steps there; rows as with the sides"""
    d, cases = common.golden(arch)
    c = [x for x in cases if x["name"] == name][0]
    kw, fastq = common.case_kwargs(c["args"])
    names, qlens, seq, off, seeds, paired = reads.load([os.path.join(d, f) for f in c["reads"]], fastq)
    e, L = emu.Emu(os.path.join(d, "idx")), emu.lib()
    try:
        L.emu_set_search_version(search_version); L.emu_set_walk_version(walk_version)
        assert L.emu_planify(e.h, 1) == 1 and L.emu_textify(e.h, 2) == 1
        assert L.emu_densify(e.h, 2 if walk_version == 2 else 1) >= 0
        assert L.emu_drop_sides(e.h, 1) == 1
        L.emu_widen(e.h, 12)
        rows, n_rows, s2 = e.classify(seq, off, seeds, paired=paired, **kw)
        got = reads.format_tsv(e.seqid, names, qlens, rows, n_rows, s2)
        assert got == open(os.path.join(d, c["tsv"])).read()
    finally:
        L.emu_set_search_version(2); L.emu_set_walk_version(3)
        e.close()


@pytest.mark.parametrize("arch,name", [("synth_small", "k5"), ("synth_small", "k1"), ("synth_small", "pe_k1"), ("synth_small", "r250_k5"), ("synth_small", "genus"), ("example", "default")])
def test_references_straight_from_the_table_give_the_same_rows(arch, name):
    """DBatch::directRefs: with the resolve table at every row the common-case score kernel reads a row's reference from the
    table itself — nothing is emitted or walked, and only the queries it leaves get their rows resolved (resolve_query_body);
    the row workspace is poisoned in that mode, so anything that still read it would show"""
    d, cases = common.golden(arch)
    c = [x for x in cases if x["name"] == name][0]
    kw, fastq = common.case_kwargs(c["args"])
    names, qlens, seq, off, seeds, paired = reads.load([os.path.join(d, f) for f in c["reads"]], fastq)
    want = open(os.path.join(d, c["tsv"])).read()
    e, L = emu.Emu(os.path.join(d, "idx")), emu.lib()
    L.emu_set_direct_refs.argtypes = [C.c_int]
    try:
        assert L.emu_planify(e.h, 1) == 1 and L.emu_densify(e.h, 0) >= 0
        for on in (1, 0):
            L.emu_set_direct_refs(on)
            rows, n_rows, s2 = e.classify(seq, off, seeds, paired=paired, **kw)
            assert reads.format_tsv(e.seqid, names, qlens, rows, n_rows, s2) == want, on
    finally:
        L.emu_set_direct_refs(1)
        e.close()


@pytest.mark.parametrize("arch,name", common.all_cases())
def test_early_score_kernel_order_gives_the_same_rows(arch, name):
    """enqueuePost's early mode (round 5): with the resolve table at every row the common-case score kernel runs right behind the
    common-case post kernel — before the general post kernel has written the plans of the queries left to it (poisoned here until
    it has), before the rows are counted and the row window is known — and the general score kernel takes its list afterwards.
    Every golden case in that order, in the plain one, and with the row workspace cut to a few rows (several passes: the early
    order serves the first pass only); rows and per-taxon counters the same"""
    d, cases = common.golden(arch)
    c = [x for x in cases if x["name"] == name][0]
    kw, fastq = common.case_kwargs(c["args"])
    names, qlens, seq, off, seeds, paired = reads.load([os.path.join(d, f) for f in c["reads"]], fastq)
    want = open(os.path.join(d, c["tsv"])).read()
    e, L = emu.Emu(os.path.join(d, "idx")), emu.lib()
    L.emu_set_early_score.argtypes = [C.c_int]
    L.emu_set_rows_cap.argtypes = [C.c_uint64]
    try:
        assert L.emu_planify(e.h, 1) == 1 and L.emu_densify(e.h, 0) >= 0
        counts = {}
        for early in (1, 0):
            for cap in (0, 7):
                L.emu_set_early_score(early)
                L.emu_set_rows_cap(cap)
                rows, n_rows, s2, cnt = e.classify(seq, off, seeds, paired=paired, counts=True, **kw)
                assert reads.format_tsv(e.seqid, names, qlens, rows, n_rows, s2) == want, (early, cap)
                counts[(early, cap)] = cnt
        for k_ in counts:
            assert np.array_equal(counts[k_], counts[(0, 0)]), k_
    finally:
        L.emu_set_early_score(0)
        L.emu_set_rows_cap(0)
        e.close()


@pytest.mark.parametrize("read_len", [100, 150, 250])
def test_small_ranges_against_the_text_on_a_repeat_rich_model(tmp_path, read_len):
    """DIndex::multiRows: on a 5 Mbp model (16 Mbp: 46.7 against 28.0) of the repeat-rich stand-in (clusters of four strains 0.4 - 1 % apart: ranges that stay a
    few rows wide for most of a read) the search costs 46.7 requests per read stepping, 28 - 30 with the small ranges finished
    against the text (SA of every row + its text windows + one inverse-SA read); rows against the reference either way"""
    import sys
    from oracle import oracle as O
    from centrifuge_amd import capi
    if not O.have_ref():
        pytest.skip("oracle/_ref (the compiled reference) is not built")
    sys.path.insert(0, os.path.join(common.ROOT, "tools"))
    import synth
    d = str(tmp_path)
    g = synth.make_repeat_genomes(32, 160000, seed=7)
    synth.write_reference(d, g, genus_size=8, uid_prefix="cid|")
    O.ref_build(d, threads=4)
    nm, s = synth.sample_reads(g, 800 if read_len == 100 else 400, read_len, seed=11)
    synth.write_fasta(os.path.join(d, "r.fa"), nm, s)
    base = os.path.join(d, "idx")
    want = O.ref_classify(base, os.path.join(d, "w.tsv"), os.path.join(d, "w.rep"), u=os.path.join(d, "r.fa"), threads=4)
    names, ql, seq, off, seeds, pr = reads.load([os.path.join(d, "r.fa")], False)
    e, L = emu.Emu(base), emu.lib()
    L.emu_set_multi_verify.argtypes = [C.c_uint32, C.c_uint32]
    try:
        L.emu_set_search_version(2)
        L.emu_planify(e.h, 1); L.emu_planify2(e.h, 1); L.emu_set_self_records(1); L.emu_widen(e.h, 11); L.emu_densify(e.h, 0)
        cost = {}
        for rows, minrun in ((0, 2), (4, 0), (15, 3)):
            L.emu_set_multi_verify(rows, minrun)
            assert L.emu_textify(e.h, 0) == 1
            # (round 6) the same with the hits of ONE row — a unique match, or a small range of which one row matches longest — in
            # their position form: the same rows, fewer requests still
            assert L.emu_posify(e.h, 1) == 1
            opp = capi.OpCounts()
            rws, n_rows, s2 = e.classify(seq, off, seeds, paired=False, ops=opp)
            assert reads.format_tsv(e.seqid, names, ql, rws, n_rows, s2) == want, (rows, minrun, "position form")
            assert L.emu_posify(e.h, 0) == 0
            ops = capi.OpCounts()
            rws, n_rows, s2 = e.classify(seq, off, seeds, paired=False, ops=ops)
            assert reads.format_tsv(e.seqid, names, ql, rws, n_rows, s2) == want, (rows, minrun)
            cost[(rows, minrun)] = (ops.n_ftab_wide + ops.n_ftab + ops.n_pair + ops.n_pair2 + ops.n_single + 2 * ops.n_verify + ops.n_text_loads) / float(len(names))
        assert cost[(0, 2)] > 40 and cost[(4, 0)] < 0.7 * cost[(0, 2)] and cost[(15, 3)] < 0.75 * cost[(0, 2)], cost      # (128-, 192- and 256-base records alike)
    finally:
        L.emu_set_multi_verify(0, 2)
        e.close()
