"""Index variants the golden archives do not hold, built on the spot with the reference's own
builder (oracle/_ref, test infrastructure) and classified by the reference binary:
  * a "compressed" index (>= 10 uids starting with `cid`, bt2_idx.h:648-663 => ihits = 20 instead of
    200): hits with more than ihits rows are skipped, twins are dropped (classifier.h:299,849-870);
  * more than 65,535 reference sequences: the SA sample is u32 (bt2_io.h:280);
  * non-default SA sampling (-o) and ftab width (-t).
CPU: the single-stepped kernel bodies (tests/emu), the oracle and the report module must reproduce
the reference's TSV + report.  GPU: the HIP path through the C ABI must."""
import os
import tempfile

import numpy as np
import pytest

import common
import synth
from centrifuge_amd import capi, reads
from oracle import oracle as O

VARIANTS = {
    # name: (genomes, length, genus_size, divergence, uid_prefix, build extra, read lens)
    "compressed": (48, 6000, 24, 0.004, "cid|", [], (100, 60)),
    "wide_sa": (66000, 120, 8, 0.05, "seq", [], (100, 40)),
    "offrate2_ftab7": (24, 8000, 8, 0.05, "seq", ["-o", "2", "-t", "7"], (100, 31)),
    "offrate6": (24, 8000, 8, 0.05, "seq", ["-o", "6"], (100, 150)),
}
_built = {}


def variant(name):
    if name not in _built:
        G, L, gs, div, prefix, extra, lens = VARIANTS[name]
        d = tempfile.mkdtemp(prefix="cf_var_%s_" % name)
        g = synth.make_genomes(G, L, genus_size=gs, divergence=div)
        synth.write_reference(d, g, genus_size=gs, uid_prefix=prefix)
        O.ref_build(d, threads=8, extra=extra)
        names, seqs = [], []
        for i, rl in enumerate(lens):
            n, s = synth.sample_reads(g, 600, rl, seed=100 + i)
            names += ["L%d_%s" % (rl, x) for x in n]
            seqs += s
        synth.write_fasta(os.path.join(d, "reads.fa"), names, seqs)
        out = {}
        for k in (1, 5):
            tsv = O.ref_classify(os.path.join(d, "idx"), os.path.join(d, "k%d.tsv" % k), os.path.join(d, "k%d.rep" % k),
                                 u=os.path.join(d, "reads.fa"), extra=["-k", str(k)])
            out[k] = (tsv, open(os.path.join(d, "k%d.rep" % k)).read())
        _built[name] = (d, out)
    return _built[name]


needs_ref = pytest.mark.skipif(not O.have_ref(), reason="oracle/_ref (the compiled reference) is not built")


@needs_ref
@pytest.mark.parametrize("name", sorted(VARIANTS))
def test_variant_on_cpu(name):
    from emu import emu
    from test_report import max_scores
    d, want = variant(name)
    base = os.path.join(d, "idx")
    e = emu.Emu(base)
    orc = O.Oracle(base)
    ix = capi.Index(base, host_only=True)
    if name == "compressed":
        assert ix.L.cf_index_compressed(ix.h) == 1
    if name == "wide_sa":
        assert ix.sa_width == 4
    nm, ql, seq, off, seeds, paired = reads.load([os.path.join(d, "reads.fa")], False)
    for k in (1, 5):
        for ver in (2, 1):
            emu.lib().emu_set_search_version(ver)
            rows, n_rows, s2 = e.classify(seq, off, seeds, paired=False, k=k)
            got = reads.format_tsv(e.seqid, nm, ql, rows, n_rows, s2)
            assert got == want[k][0], common.first_diff(got, want[k][0])
        assert orc.classify_files(os.path.join(d, "reads.fa"), k=k) == want[k][0]
        rep = capi.Report(ix)
        rep.add(rows, n_rows, max_scores(orc, seq, off, False), k)
        with tempfile.TemporaryDirectory() as t:
            rep.write(os.path.join(t, "r.tsv"))
            assert open(os.path.join(t, "r.tsv")).read() == want[k][1]
        rep.close()
    ix.close()


@needs_ref
@pytest.mark.gpu
@pytest.mark.parametrize("name", sorted(VARIANTS))
def test_variant_on_gpu(name):
    d, want = variant(name)
    ix = capi.Index(os.path.join(d, "idx"), device=0)
    nm, ql, seq, off, seeds, paired = reads.load([os.path.join(d, "reads.fa")], False)
    for k in (1, 5):
        clf = capi.Classifier(ix, k=k)
        b = clf.batch(seq, off, seeds, False)
        b.classify()
        rows, n_rows, s2 = b.results()
        got = reads.format_tsv(ix.seqid, nm, ql, rows, n_rows, s2)
        assert got == want[k][0], common.first_diff(got, want[k][0])
        b.close(); clf.close()
    ix.close()


@needs_ref
@pytest.mark.gpu
@pytest.mark.parametrize("name", ["offrate2_ftab7", "offrate6", "compressed"])
def test_variant_index_from_our_builder(name):
    """the GPU builder with -o / -t and `cid` uids writes the same files as the reference builder"""
    import filecmp
    d, _ = variant(name)
    G, L, gs, div, prefix, extra, lens = VARIANTS[name]
    kw = {}
    for i in range(0, len(extra), 2):
        kw["off_rate" if extra[i] == "-o" else "ftab_chars"] = int(extra[i + 1])
    ours = os.path.join(d, "ours")
    capi.build_index(ours, fasta=[os.path.join(d, "genomes.fa")], conversion_table=os.path.join(d, "conv.tsv"),
                     taxonomy_tree=os.path.join(d, "nodes.dmp"), name_table=os.path.join(d, "names.dmp"), **kw)
    for ext in "1234":
        assert filecmp.cmp(os.path.join(d, "idx.%s.cf" % ext), ours + ".%s.cf" % ext, shallow=False), ext


# ---- random option combinations against the reference binary run on the spot
def _combos(n, seed=2024):
    rng = np.random.default_rng(seed)
    out = []
    for _ in range(n):
        kw = {"k": int(rng.choice([1, 2, 3, 5, 10])), "min_hitlen": int(rng.choice([15, 22, 23, 30, 45])),
              "rank": str(rng.choice(["strain", "species", "genus", "family"])), "traverse": bool(rng.random() < 0.75)}
        if rng.random() < 0.35:
            kw["host"] = [int(x) for x in rng.choice([1000, 1003, 1009, 1017, 100, 102, 50], size=int(rng.integers(1, 3)), replace=False)]
        if rng.random() < 0.35:
            kw["exclude"] = [int(x) for x in rng.choice([1001, 1004, 1010, 1020, 101, 2], size=int(rng.integers(1, 3)), replace=False)]
        out.append(kw)
    return out


def _ref_args(kw):
    a = ["-k", str(kw["k"]), "--min-hitlen", str(kw["min_hitlen"]), "--classification-rank", kw["rank"]]
    if not kw["traverse"]:
        a.append("--no-traverse")
    if kw.get("host"):
        a += ["--host-taxids", ",".join(map(str, kw["host"]))]
    if kw.get("exclude"):
        a += ["--exclude-taxids", ",".join(map(str, kw["exclude"]))]
    return a


_combo_truth = {}


def _truth(i, kw):
    if i not in _combo_truth:
        d, _ = variant("offrate6")
        _combo_truth[i] = O.ref_classify(os.path.join(d, "idx"), os.path.join(d, "c%d.tsv" % i), os.path.join(d, "c%d.rep" % i),
                                         u=os.path.join(d, "reads.fa"), extra=_ref_args(kw))
    return _combo_truth[i]


@needs_ref
@pytest.mark.parametrize("i,kw", list(enumerate(_combos(10))))
def test_random_options_on_cpu(i, kw):
    from emu import emu
    emu.lib().emu_set_search_version(2)
    d, _ = variant("offrate6")
    e = emu.Emu(os.path.join(d, "idx"))
    nm, ql, seq, off, seeds, paired = reads.load([os.path.join(d, "reads.fa")], False)
    rows, n_rows, s2 = e.classify(seq, off, seeds, paired=False, **kw)
    got = reads.format_tsv(e.seqid, nm, ql, rows, n_rows, s2)
    want = _truth(i, kw)
    assert got == want, (kw, common.first_diff(got, want))


@needs_ref
@pytest.mark.gpu
@pytest.mark.parametrize("i,kw", list(enumerate(_combos(10))))
def test_random_options_on_gpu(i, kw):
    d, _ = variant("offrate6")
    ix = capi.Index(os.path.join(d, "idx"), device=0)
    nm, ql, seq, off, seeds, paired = reads.load([os.path.join(d, "reads.fa")], False)
    clf = capi.Classifier(ix, **kw)
    b = clf.batch(seq, off, seeds, False)
    b.classify()
    rows, n_rows, s2 = b.results()
    got = reads.format_tsv(ix.seqid, nm, ql, rows, n_rows, s2)
    want = _truth(i, kw)
    assert got == want, (kw, common.first_diff(got, want))
    b.close(); clf.close(); ix.close()


@needs_ref
@pytest.mark.parametrize("paired", [False, True])
@pytest.mark.parametrize("min_hitlen", [15, 18, 22, 25, 30, 40])
def test_min_hitlen_sweep(min_hitlen, paired):
    """every minHitLen regime (below / at / above the literal 22 of compareBWTHits), 250 bp reads (more
    than 16 hits per strand: introsort path) and pairs (the `ts` quirk on `break` couples the mates),
    against the reference binary.  (An experiment that stored only the hits >= minHitLen in k_search2
    broke exactly here — and bought no time: profiles/r01_sweeps.txt.)"""
    from emu import emu
    d, _ = common.golden("synth_small")
    base = os.path.join(d, "idx")
    files = [os.path.join(d, "r1.fa"), os.path.join(d, "r2.fa")] if paired else [os.path.join(d, "reads250.fa")]
    e = emu.Emu(base)
    nm, ql, seq, off, seeds, pr = reads.load(files, False)
    for k in (1, 5):
        with tempfile.TemporaryDirectory() as t:
            kw = dict(m1=files[0], m2=files[1]) if paired else dict(u=files[0])
            want = O.ref_classify(base, os.path.join(t, "w.tsv"), os.path.join(t, "w.rep"), extra=["-k", str(k), "--min-hitlen", str(min_hitlen)], **kw)
        for ver in (2, 1):
            emu.lib().emu_set_search_version(ver)
            rows, n_rows, s2 = e.classify(seq, off, seeds, paired=pr, k=k, min_hitlen=min_hitlen)
            got = reads.format_tsv(e.seqid, nm, ql, rows, n_rows, s2)
            assert got == want, (k, ver, common.first_diff(got, want))
