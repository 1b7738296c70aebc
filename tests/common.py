"""Shared helpers of the test-suite."""
import json
import os
import tarfile
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")
_cache = {}


def golden(name):
    """Extract tests/golden/<name>.tar.xz once per session -> (dir, cases)."""
    if name not in _cache:
        d = tempfile.mkdtemp(prefix="cf_golden_%s_" % name)
        with tarfile.open(os.path.join(GOLDEN, name + ".tar.xz")) as t:
            t.extractall(d)
        _cache[name] = (d, json.load(open(os.path.join(d, "cases.json"))))
    return _cache[name]


def case_kwargs(args):
    """reference CLI args of a golden case -> (classifier kwargs, fastq)."""
    kw, fastq, i = {}, False, 0
    while i < len(args):
        a = args[i]
        if a == "-f":
            pass
        elif a == "-q":
            fastq = True
        elif a == "-k":
            kw["k"] = int(args[i + 1]); i += 1
        elif a == "--no-traverse":
            kw["traverse"] = False
        elif a == "--classification-rank":
            kw["rank"] = args[i + 1]; i += 1
        elif a == "--host-taxids":
            kw["host"] = [int(x) for x in args[i + 1].split(",")]; i += 1
        elif a == "--exclude-taxids":
            kw["exclude"] = [int(x) for x in args[i + 1].split(",")]; i += 1
        elif a == "--min-hitlen":
            kw["min_hitlen"] = int(args[i + 1]); i += 1
        else:
            raise ValueError(a)
        i += 1
    return kw, fastq


def all_cases():
    out = []
    for arch in ("example", "synth_small"):
        _, cases = golden(arch)
        out += [(arch, c["name"]) for c in cases]
    return out


def first_diff(a, b, n=5):
    la, lb = a.splitlines(), b.splitlines()
    msgs = ["%d vs %d lines" % (len(la), len(lb))]
    for x, y in zip(la, lb):
        if x != y:
            msgs.append("got: %s\nref: %s" % (x, y))
            if len(msgs) > n:
                break
    return "\n".join(msgs)


# ---- degenerate / boundary-length batches shared by the CPU (emu) and GPU parity tests
import numpy as np  # noqa: E402

EDGE_CASES = [
    ([0, 1, 2, 9, 10, 11, 15, 21, 22, 23, 24, 31, 32, 33, 63, 64, 65, 100, 127, 128], False, 5),      # 64-byte strand records
    ([129, 130, 150, 160, 161, 191, 192, 64, 32, 10], False, 5),                                       # 96-byte strand records
    ([129, 193, 200, 255, 256, 64, 32, 10], False, 1),                                                 # 128-byte strand records
    ([257, 300, 100, 22, 513], False, 5),                                                              # byte-window kernel
    ([100, 100, 128, 23, 0, 60, 2, 90, 250, 250], True, 5),                                            # pairs, ragged mates
]


def edge_reads(recs, lengths, rng):
    """golden reads cut / padded to the given lengths, plus degenerate reads"""
    out = []
    pool = [c for _, c, _ in recs if len(c) >= 100]
    for i, L in enumerate(lengths):
        src = pool[i % len(pool)]
        if L <= len(src):
            s = int(rng.integers(0, len(src) - L + 1)) if L else 0
            r = src[s:s + L].copy()
        else:                                   # longer than any golden read: chain several
            r = np.concatenate([pool[(i + j) % len(pool)] for j in range(L // 100 + 2)])[:L].copy()
        out.append(r)
    out += [np.zeros(0, dtype=np.uint8), np.full(50, 4, dtype=np.uint8), np.full(1, 2, dtype=np.uint8),
            np.zeros(40, dtype=np.uint8), np.full(33, 3, dtype=np.uint8)]
    n_run = pool[0][:90].copy(); n_run[40:46] = 4
    out.append(n_run)
    # chimeras: a genome piece followed by the reverse complement of another one, so BOTH strands of the
    # read carry a long hit (cross-strand extension, twin removal, full hit lists needed)
    def rc(x):
        y = x[::-1].copy()
        y[y < 4] = 3 - y[y < 4]
        return y
    for a, b_, la, lb in ((1, 2, 45, 45), (3, 3, 50, 50), (4, 5, 30, 60), (6, 7, 24, 70), (8, 8, 40, 23)):
        out.append(np.concatenate([pool[a % len(pool)][:la], rc(pool[b_ % len(pool)][10:10 + lb])]))
    return out




# ---- contigs and long reads (missing 3): reads far beyond 65,535 bases, which rounds 1-3 refused
def long_read_case(d, seed=2024):
    """An index of 8 genomes x 160 kb in two genera (built by the reference builder in `d`) and r.fa with contigs of 66 - 300 kb:
    a genome piece with a substitution every ~700 bases, the reverse complement of one, a chimera of two genomes, one with N
    runs, one of random bases, plus a few ordinary reads in between.  -> (base, fasta path)"""
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import synth
    from oracle import oracle as O
    rng = np.random.default_rng(seed)
    g = synth.make_genomes(8, 160000, genus_size=4, divergence=0.03, seed=seed)
    synth.write_reference(d, g, genus_size=4)
    O.ref_build(d, threads=4)

    def piece(gi, a, n, every):
        r = g[gi, a:a + n].copy()
        for q in range(int(rng.integers(50, every)), n, every):
            r[q] = synth.ACGT[(np.searchsorted(synth.ACGT, r[q]) + int(rng.integers(1, 4))) & 3]
        return r
    seqs = []
    seqs.append(piece(0, 1000, 100000, 700))
    seqs.append(synth.COMP[piece(5, 30000, 70001, 650)[::-1]])
    seqs.append(np.concatenate([piece(2, 0, 66000, 900), piece(6, 90000, 66000, 800)]))
    n_ = piece(3, 20000, 131072, 500)
    for q in (5000, 65535, 65536, 100000):
        n_[q:q + int(rng.integers(1, 40))] = ord("N")
    seqs.append(n_)
    seqs.append(synth.ACGT[rng.integers(0, 4, size=80000, dtype=np.uint8)])
    seqs.append(np.concatenate([g[1], g[1][:140000]]))                      # 300 kb: a genome and most of it again
    seqs.append(piece(7, 500, 65535, 400))
    seqs.append(piece(4, 77, 65536, 1000))
    nm, short = synth.sample_reads(g, 6, 100, seed=seed + 1)
    names = ["contig%d" % i for i in range(len(seqs))] + nm
    synth.write_fasta(os.path.join(d, "r.fa"), names, [s.tobytes() for s in seqs] + short)
    return os.path.join(d, "idx"), os.path.join(d, "r.fa")
