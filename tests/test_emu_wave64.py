"""The search kernel's CROSS-LANE code off the GPU (VERDICT r5 missing 5 / weak 1 ii): tests/emu built with CF_EMU_WAVE64 runs
search2_body as a wavefront of 64 lanes — 64 fibers that meet at every cf_ballot / cf_shfl / cf_first_lane_u32 (cf_platform.hpp,
emu_run_wave) — so the ballot ranks of the work queue, the hand-out of a chunk's item records between lanes (three shuffles per
item), idle lanes beside running ones, lazy-hit replay in one lane while its neighbours go on, and the kernel's end (lanes
leaving one by one) are exercised by the CPU suite and the fuzzers, not only on the GPU box.  Everything else of a batch runs as
in the one-lane build, so a difference in the rows is the search kernel's."""
import ctypes as C
import os

import numpy as np
import pytest

import common
from centrifuge_amd import reads
from emu import emu


@pytest.fixture()
def wave64():
    was = emu.use_wave64(True)
    try:
        L = emu.lib()
        assert L.emu_wave_lanes() == 64
        yield L
    finally:
        emu.use_wave64(was)


def _case(arch, name):
    d, cases = common.golden(arch)
    c = [x for x in cases if x["name"] == name][0]
    kw, fastq = common.case_kwargs(c["args"])
    names, qlens, seq, off, seeds, paired = reads.load([os.path.join(d, f) for f in c["reads"]], fastq)
    return d, c, kw, names, qlens, seq, off, seeds, paired


@pytest.mark.parametrize("tables", ["planes", "planes_wide_text", "pairs_wide_text_dense", "sides"])
@pytest.mark.parametrize("arch,name", common.all_cases())
def test_golden_cases_through_a_wavefront_of_64_lanes(wave64, arch, name, tables):
    """every golden case, the search stage as ONE wavefront of 64 chains over the planes (the production kernel), with and without
    the wide ftab / text verification / pair planes / resolve table, and over the sides (one lane per chain): the reference's TSV"""
    L = wave64
    L.emu_set_search_version(2)
    d, c, kw, names, qlens, seq, off, seeds, paired = _case(arch, name)
    e = emu.Emu(os.path.join(d, "idx"))
    try:
        if tables != "sides":
            assert L.emu_planify(e.h, 1) == 1
        if tables in ("planes_wide_text", "pairs_wide_text_dense"):
            assert L.emu_widen(e.h, 12) == 1 and L.emu_textify(e.h, 1 if tables == "planes_wide_text" else 0) == 1
        if tables == "pairs_wide_text_dense":                   # (config 2's plan: hits in the position form as well)
            assert L.emu_planify2(e.h, 1) == 1 and L.emu_densify(e.h, 0) == 1 and L.emu_posify(e.h, 1) == 1
        rows, n_rows, score2 = e.classify(seq, off, seeds, paired=paired, **kw)
        got = reads.format_tsv(e.seqid, names, qlens, rows, n_rows, score2)
        want = open(os.path.join(d, c["tsv"])).read()
        assert got == want, common.first_diff(got, want)
        if len(names) > 64 and max(qlens) <= 256:
            assert L.emu_wave_collectives() > 50          # the lanes really met (queue ballots, record hand-outs)
    finally:
        e.close()


@pytest.mark.parametrize("lengths,paired,k", common.EDGE_CASES)
def test_edge_batches_through_a_wavefront_of_64_lanes(wave64, lengths, paired, k):
    """empty, ragged, all-N, one-read and length-boundary batches: fewer items than lanes, lanes that never get one"""
    from oracle import oracle as O
    L = wave64
    L.emu_set_search_version(2)
    d, _ = common.golden("synth_small")
    base = os.path.join(d, "idx")
    orc, e = O.Oracle(base), emu.Emu(base)
    try:
        assert L.emu_planify(e.h, 1) == 1 and L.emu_widen(e.h, 12) == 1 and L.emu_textify(e.h, 2) == 1
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
    finally:
        e.close()


def test_small_ranges_and_lazy_hits_across_lanes(wave64, tmp_path):
    """a repeat-rich model (strain clusters: ranges of a few rows finished against the text, MULTI kernel) and reads that turn
    long late (a lane replays its strand with direct stores while its 63 neighbours go on): rows against the compiled reference"""
    import sys
    from oracle import oracle as O
    if not O.have_ref():
        pytest.skip("oracle/_ref (the compiled reference) is not built")
    sys.path.insert(0, os.path.join(common.ROOT, "tools"))
    import synth
    L = wave64
    L.emu_set_multi_verify.argtypes = [C.c_uint32, C.c_uint32]
    d = str(tmp_path)
    g = synth.make_genomes(16, 200000, genus_size=4, divergence=0.01, seed=99)
    synth.write_reference(d, g, genus_size=4)
    O.ref_build(d, threads=4)
    nm, s = synth.sample_reads(g, 1500, 100, seed=5)
    synth.write_fasta(os.path.join(d, "r.fa"), nm, s)
    base = os.path.join(d, "idx")
    want = O.ref_classify(base, os.path.join(d, "w.tsv"), os.path.join(d, "w.rep"), u=os.path.join(d, "r.fa"), threads=4)
    e = emu.Emu(base)
    try:
        L.emu_set_search_version(2)
        L.emu_planify(e.h, 1); L.emu_planify2(e.h, 1); L.emu_widen(e.h, 11); L.emu_densify(e.h, 0)
        names, ql, seq, off, seeds, pr = reads.load([os.path.join(d, "r.fa")], False)
        for rows_, minrun in ((0, 2), (4, 0)):
            L.emu_set_multi_verify(rows_, minrun)
            assert L.emu_textify(e.h, 0) == 1
            for lazy, pos in ((1, 0), (0, 0), (1, 1)):           # (pos: one-row hits — and small ranges with ONE longest row — in the position form)
                L.emu_set_lazy_hits(lazy)
                assert L.emu_posify(e.h, pos) == pos
                rws, n_rows, s2 = e.classify(seq, off, seeds, paired=False)
                got = reads.format_tsv(e.seqid, names, ql, rws, n_rows, s2)
                assert got == want, (rows_, lazy, pos, common.first_diff(got, want))
    finally:
        L.emu_set_multi_verify(0, 2); L.emu_set_lazy_hits(1)
        e.close()
