"""Index construction on the GPU (cf_build_index) against the unmodified reference
builder (oracle/_ref/centrifuge-build-bin, test infrastructure): the four files
<base>.{1,2,3,4}.cf must be byte-identical for the same inputs."""
import filecmp
import os
import tempfile

import numpy as np
import pytest

import synth
from centrifuge_amd import capi
from oracle import oracle as O

pytestmark = pytest.mark.gpu

LUT = np.zeros(256, dtype=np.uint8)
for _ch, _v in zip(b"ACGTN", range(5)):
    LUT[_ch] = _v


def _compare(d, ours):
    for ext in ("1", "2", "3", "4"):
        a, b = os.path.join(d, "idx.%s.cf" % ext), ours + ".%s.cf" % ext
        assert os.path.getsize(a) == os.path.getsize(b), "size of .%s.cf: ref %d ours %d" % (ext, os.path.getsize(a), os.path.getsize(b))
        if not filecmp.cmp(a, b, shallow=False):
            x, y = np.fromfile(a, dtype=np.uint8), np.fromfile(b, dtype=np.uint8)
            bad = np.nonzero(x != y)[0]
            raise AssertionError(".%s.cf differs at %d byte(s), first at offset %d" % (ext, len(bad), bad[0]))


class _env:
    """environment knobs of the builder for the length of a with block"""
    def __init__(self, **kw):
        self.kw = {k: str(v) for k, v in kw.items()}

    def __enter__(self):
        self.old = {k: os.environ.get(k) for k in self.kw}
        os.environ.update(self.kw)

    def __exit__(self, *a):
        for k, v in self.old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


def _both(d, g, chunk=0, via="fasta", **wr):
    synth.write_reference(d, g, **wr)
    O.ref_build(d, threads=8)
    ours = os.path.join(d, "ours")
    kw = dict(conversion_table=os.path.join(d, "conv.tsv"), taxonomy_tree=os.path.join(d, "nodes.dmp"),
              name_table=os.path.join(d, "names.dmp"), chunk_suffixes=chunk)
    if via == "fasta":
        capi.build_index(ours, fasta=[os.path.join(d, "genomes.fa")], **kw)
    else:
        n, L = g.shape
        prefix = wr.get("uid_prefix", "seq")
        names = [b"%s%d synthetic genome %d" % (prefix.encode(), i, i) for i in range(n)]
        off = np.arange(n + 1, dtype=np.uint64) * np.uint64(L)
        capi.build_index(ours, codes=LUT[g].reshape(-1), seq_off=off, seq_names=names, **kw)
    _compare(d, ours)


@pytest.mark.skipif(not O.have_ref(), reason="oracle/_ref not built")
@pytest.mark.parametrize("via", ["fasta", "memory"])
def test_small_genera(via):
    with tempfile.TemporaryDirectory() as d:
        _both(d, synth.make_genomes(16, 20000), via=via)


@pytest.mark.skipif(not O.have_ref(), reason="oracle/_ref not built")
def test_many_chunks_and_refinement():
    """chunk size far below the suffix count: many GPU passes; relatives at 5 %
    divergence force several 29-mer refinement rounds."""
    with tempfile.TemporaryDirectory() as d:
        _both(d, synth.make_genomes(24, 30000, divergence=0.01), chunk=50000)


@pytest.mark.skipif(not O.have_ref(), reason="oracle/_ref not built")
def test_gaps_split_sequences_into_fragments():
    with tempfile.TemporaryDirectory() as d:
        _both(d, synth.make_genomes(12, 15000), n_in_genomes=5)


@pytest.mark.skipif(not O.have_ref(), reason="oracle/_ref not built")
def test_exact_duplicates_and_low_complexity():
    """identical genomes (ties run to the end of the text) and homopolymer runs"""
    with tempfile.TemporaryDirectory() as d:
        g = synth.make_genomes(8, 6000)
        g[1] = g[0]
        g[2, 1000:3000] = ord("A")
        g[3, 500:2500] = np.frombuffer(b"AC" * 1000, dtype=np.uint8)
        g[7, -40:] = ord("T")
        _both(d, g, chunk=20000)


@pytest.mark.skipif(not O.have_ref(), reason="oracle/_ref not built")
def test_wide_sa_sample():
    """> 65,535 sequences: the SA sample is u32 (bt2_io.h:280)"""
    with tempfile.TemporaryDirectory() as d:
        _both(d, synth.make_genomes(66000, 40, genus_size=8), via="memory")


@pytest.mark.skipif(not O.have_ref(), reason="oracle/_ref not built")
def test_built_index_classifies_like_the_reference():
    with tempfile.TemporaryDirectory() as d:
        g = synth.make_genomes(16, 20000)
        synth.write_reference(d, g)
        names, seqs = synth.sample_reads(g, 2000, 100)
        synth.write_fasta(os.path.join(d, "reads.fa"), names, seqs)
        ours = os.path.join(d, "ours")
        capi.build_index(ours, fasta=[os.path.join(d, "genomes.fa")], conversion_table=os.path.join(d, "conv.tsv"),
                         taxonomy_tree=os.path.join(d, "nodes.dmp"), name_table=os.path.join(d, "names.dmp"))
        want = O.ref_classify(ours, os.path.join(d, "ref.tsv"), os.path.join(d, "ref_rep.tsv"), u=os.path.join(d, "reads.fa"))
        from centrifuge_amd import reads
        ix = capi.Index(ours, device=0)
        clf = capi.Classifier(ix)
        nm, ql, seq, off, seeds, paired = reads.load([os.path.join(d, "reads.fa")], False)
        b = clf.batch(seq, off, seeds, paired)
        b.classify()
        rows, n_rows, s2 = b.results()
        assert reads.format_tsv(ix.seqid, nm, ql, rows, n_rows, s2) == want
        b.close(); clf.close(); ix.close()


@pytest.mark.skipif(not O.have_ref(), reason="oracle/_ref not built")
def test_long_and_mixed_length_reads_match_the_reference():
    """reads longer than 256 bp do not fit k_search2's strand records: the batch takes the
    byte-window kernel (k_search) instead; mixed lengths exercise ragged offsets"""
    with tempfile.TemporaryDirectory() as d:
        g = synth.make_genomes(16, 20000)
        synth.write_reference(d, g)
        from centrifuge_amd import reads
        parts = []
        for n, L, seed in ((300, 100, 1), (300, 180, 2), (200, 300, 3), (100, 700, 4), (50, 40, 5), (20, 23, 6)):
            names, seqs = synth.sample_reads(g, n, L, seed=seed)
            parts += [("L%d_%s" % (L, nm), s) for nm, s in zip(names, seqs)]
        rng = np.random.default_rng(9)
        order = rng.permutation(len(parts))
        synth.write_fasta(os.path.join(d, "reads.fa"), [parts[i][0] for i in order], [parts[i][1] for i in order])
        ours = os.path.join(d, "ours")
        capi.build_index(ours, fasta=[os.path.join(d, "genomes.fa")], conversion_table=os.path.join(d, "conv.tsv"),
                         taxonomy_tree=os.path.join(d, "nodes.dmp"), name_table=os.path.join(d, "names.dmp"))
        ix = capi.Index(ours, device=0)
        for extra, kw in ((["-k", "1"], {"k": 1}), ([], {})):
            want = O.ref_classify(ours, os.path.join(d, "ref.tsv"), os.path.join(d, "ref_rep.tsv"), u=os.path.join(d, "reads.fa"), extra=extra)
            clf = capi.Classifier(ix, **kw)
            nm, ql, seq, off, seeds, paired = reads.load([os.path.join(d, "reads.fa")], False)
            b = clf.batch(seq, off, seeds, paired)
            b.classify()
            rows, n_rows, s2 = b.results()
            got = reads.format_tsv(ix.seqid, nm, ql, rows, n_rows, s2)
            assert got == want
            b.close(); clf.close()
        ix.close()


@pytest.mark.skipif(not O.have_ref(), reason="oracle/_ref not built")
def test_build_cli_is_a_drop_in_for_centrifuge_build():
    """centrifuge_amd/bin/centrifuge-build-bin with the reference's command line"""
    import subprocess
    exe = os.path.join(os.path.dirname(os.path.abspath(capi.LIB_PATH)), "bin", "centrifuge-build-bin")
    with tempfile.TemporaryDirectory() as d:
        synth.write_reference(d, synth.make_genomes(10, 12000), n_in_genomes=2)
        O.ref_build(d, threads=4)
        r = subprocess.run([exe, "-p", "4", "--conversion-table", os.path.join(d, "conv.tsv"), "--taxonomy-tree", os.path.join(d, "nodes.dmp"),
                            "--name-table", os.path.join(d, "names.dmp"), os.path.join(d, "genomes.fa"), os.path.join(d, "ours")],
                           capture_output=True, text=True)
        assert r.returncode == 0, r.stderr
        _compare(d, os.path.join(d, "ours"))


@pytest.mark.skipif(not O.have_ref(), reason="oracle/_ref not built")
def test_leading_trailing_and_all_gap_sequences():
    """sequences that start / end with N runs, an all-N sequence (no fragment at all), a 30 bp one"""
    rng = np.random.default_rng(21)
    r = lambda n: synth.ACGT[rng.integers(0, 4, n, dtype=np.uint8)].tobytes()   # noqa: E731
    seqs = [("seq0 leading and trailing gaps", b"N" * 7 + r(300) + b"N" * 12 + r(150) + b"N" * 33), ("seq1 all gaps", b"N" * 90),
            ("seq2 short", r(30)), ("seq3 plain", r(500)),
            ("seq4 many gaps", b"N" + r(40) + b"N" + r(41) + b"NN" + r(42) + b"N" * 70 + r(43) + b"N"),
            ("seq5 last with trailing gap", r(200) + b"N" * 5)]
    with tempfile.TemporaryDirectory() as d:
        with open(os.path.join(d, "genomes.fa"), "wb") as f:
            for nm, s in seqs:
                f.write(b">" + nm.encode() + b"\n")
                for p in range(0, len(s), 70):
                    f.write(s[p:p + 70] + b"\n")
        synth.write_taxonomy(d, len(seqs), genus_size=3)
        O.ref_build(d, threads=1)
        ours = os.path.join(d, "ours")
        capi.build_index(ours, fasta=[os.path.join(d, "genomes.fa")], conversion_table=os.path.join(d, "conv.tsv"),
                         taxonomy_tree=os.path.join(d, "nodes.dmp"), name_table=os.path.join(d, "names.dmp"))
        _compare(d, ours)


# ---- tie refinement in log rounds: a bounded number of 29-mer rounds, then prefix doubling over the
# inverse suffix array.  CF_BUILD_ROUNDS = 29-mer rounds before a chunk hands its ties over (default 24; 1 sends nearly every
# tie through the doubling rounds), CF_BUILD_DOUBLING=0 = 29-mer rounds only (the round-2 behaviour, any depth).
@pytest.mark.skipif(not O.have_ref(), reason="oracle/_ref not built")
@pytest.mark.parametrize("rounds", [1, 2, 5])
def test_prefix_doubling_on_duplicates_and_low_complexity(rounds):
    """identical genomes (ties to the end of the text), homopolymer / dinucleotide runs of thousands of bases, many chunks"""
    with tempfile.TemporaryDirectory() as d, _env(CF_BUILD_ROUNDS=rounds):
        g = synth.make_genomes(8, 6000)
        g[1] = g[0]
        g[5, 100:] = g[4, :-100]                      # the same text shifted: ties that end at different places
        g[2, 1000:3000] = ord("A")
        g[3, 500:2500] = np.frombuffer(b"AC" * 1000, dtype=np.uint8)
        g[6, 3000:5000] = np.frombuffer(b"ACG" * 667, dtype=np.uint8)[:2000]
        g[7, -40:] = ord("T")
        _both(d, g, chunk=20000)


@pytest.mark.skipif(not O.have_ref(), reason="oracle/_ref not built")
@pytest.mark.parametrize("rounds", [1, 3])
def test_prefix_doubling_small_repeat_rich(rounds):
    with tempfile.TemporaryDirectory() as d, _env(CF_BUILD_ROUNDS=rounds):
        _both(d, synth.make_repeat_genomes(16, 30000), chunk=100000, via="memory")


@pytest.mark.skipif(not O.have_ref(), reason="oracle/_ref not built")
@pytest.mark.parametrize("rounds", [2, 24])
def test_repeat_rich_64mbp_matches_the_reference_builder(rounds):
    """16 x 4 Mbp of strain clusters, shared 5 kb operons and low-complexity tracts: LCPs in the thousands.  rounds = 24 is the
    default (most ties fall to the 29-mer rounds, the operons and tracts to the doubling), 2 leaves nearly everything to it."""
    with tempfile.TemporaryDirectory() as d, _env(CF_BUILD_ROUNDS=rounds):
        _both(d, synth.make_repeat_genomes(16, 4194304), via="memory")


def test_doubling_and_plain_rounds_build_the_same_256mbp_index():
    """beyond what the reference builder finishes in test time: the same repeat-rich 256 Mbp built twice — ties finished by
    prefix doubling, and by 29-mer rounds alone — must give the same files; the text restored from the index is the input"""
    g = synth.make_repeat_genomes(64, 4194304, seed=8)
    n, L = g.shape
    codes = LUT[g].reshape(-1)
    off = np.arange(n + 1, dtype=np.uint64) * np.uint64(L)
    names = [b"seq%d synthetic genome %d" % (i, i) for i in range(n)]
    with tempfile.TemporaryDirectory() as d:
        synth.write_taxonomy(d, n)
        kw = dict(conversion_table=os.path.join(d, "conv.tsv"), taxonomy_tree=os.path.join(d, "nodes.dmp"), name_table=os.path.join(d, "names.dmp"))
        a, b = os.path.join(d, "doubling"), os.path.join(d, "plain")
        with _env(CF_BUILD_ROUNDS=8):
            ta = capi.build_index(a, codes=codes, seq_off=off, seq_names=names, **kw)
        with _env(CF_BUILD_DOUBLING=0):
            tb = capi.build_index(b, codes=codes, seq_off=off, seq_names=names, **kw)
        for ext in ("1", "2", "3", "4"):
            assert filecmp.cmp(a + ".%s.cf" % ext, b + ".%s.cf" % ext, shallow=False), ext
        print("256 Mbp repeat-rich: doubling %.1fs, 29-mer rounds alone %.1fs" % (ta[1], tb[1]))
        ix = capi.Index(a, device=0)
        packed = ix.restore()
        text = np.zeros(n * L, dtype=np.uint8)
        for k in range(4):
            text[k::4] = (packed[:n * L // 4] >> (2 * k)) & 3
        assert np.array_equal(text, codes)
        ix.close()
