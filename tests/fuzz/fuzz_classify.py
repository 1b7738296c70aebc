"""Offline fuzz (not collected by pytest): random small indexes (relatives, repeats, low complexity, gaps, -o / -t variants),
random reads (single / paired, 30-250 bp, N-rich) and random options; the kernel bodies stepped on the CPU (tests/emu, both
search versions) against the compiled reference (oracle/_ref) run on the spot.  usage: fuzz_classify.py <seconds> [seed0]"""
import os, sys, tempfile, time, subprocess
ROOT=os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0,ROOT); sys.path.insert(0,os.path.join(ROOT,'tests')); sys.path.insert(0,os.path.join(ROOT,'tools'))
import numpy as np
import synth, common
from centrifuge_amd import reads
from oracle import oracle as O
from emu import emu
from centrifuge_amd import capi
import test_report as TR
if os.environ.get("CF_EMU_WAVE64"):          # the search kernel's wavefront as 64 lanes (tests/emu, CF_EMU_WAVE64): the cross-lane code under the fuzzer
    emu.use_wave64(True)
t_end = time.time() + float(sys.argv[1])
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 0
it = 0; bad = 0
while time.time() < t_end:
    rng = np.random.default_rng(seed0 + it); it += 1
    G = int(rng.integers(3, 20)); L = int(rng.integers(800, 6000)); gs = int(rng.choice([1,2,4,8]))
    div = float(rng.choice([0.0, 0.002, 0.01, 0.05]))
    d = tempfile.mkdtemp(prefix="fz")
    g = synth.make_genomes(G, L, genus_size=gs, divergence=div, seed=int(rng.integers(1<<30)))
    # repeats / low complexity
    for _ in range(int(rng.integers(0, 6))):
        i = int(rng.integers(0, G)); p = int(rng.integers(0, L-200)); n = int(rng.integers(20, 200))
        kind = rng.integers(0, 3)
        if kind == 0: g[i, p:p+n] = ord("ACGT"[int(rng.integers(0,4))])
        elif kind == 1: g[i, p:p+n] = np.frombuffer((b"AC"*100)[:n], dtype=np.uint8)
        else:
            j = int(rng.integers(0, G)); q = int(rng.integers(0, L-200)); g[i, p:p+n] = g[j, q:q+n]
    synth.write_reference(d, g, genus_size=gs, n_in_genomes=int(rng.integers(0,3)))
    extra = []
    if rng.random() < 0.3: extra += ["-o", str(int(rng.choice([2,3,5,6])))]
    if rng.random() < 0.3: extra += ["-t", str(int(rng.choice([6,8,9])))]
    O.ref_build(d, threads=2, extra=tuple(extra))
    rl = int(rng.choice([30, 50, 100, 100, 150, 250]))
    paired = bool(rng.random() < 0.3) and rl <= min(150, L//4)
    n = 150
    if paired:
        (nm, s1), (_, s2) = synth.sample_reads(g, n, rl, paired=True, random_frac=0.05, n_frac=float(rng.choice([0,0.1,0.4])), seed=int(rng.integers(1<<30)))
        synth.write_fasta(d+"/r1.fa", nm, s1, "/1"); synth.write_fasta(d+"/r2.fa", nm, s2, "/2")
        files=[d+"/r1.fa", d+"/r2.fa"]
    else:
        nm, s = synth.sample_reads(g, n, min(rl, L//2), random_frac=0.05, n_frac=float(rng.choice([0,0.1,0.4])), seed=int(rng.integers(1<<30)))
        synth.write_fasta(d+"/r.fa", nm, s); files=[d+"/r.fa"]
    kw = {"k": int(rng.choice([1,2,3,5,10,50])), "min_hitlen": int(rng.choice([15,16,22,23,30,45])),
          "rank": str(rng.choice(["strain","species","genus","family"])), "traverse": bool(rng.random()<0.75)}
    if rng.random()<0.3: kw["host"]=[int(x) for x in rng.choice(np.arange(1000,1000+G), size=min(G,int(rng.integers(1,3))), replace=False)]
    if rng.random()<0.3: kw["exclude"]=[int(x) for x in rng.choice(np.arange(1000,1000+G), size=min(G,int(rng.integers(1,3))), replace=False)]
    a = ["-k", str(kw["k"]), "--min-hitlen", str(kw["min_hitlen"]), "--classification-rank", kw["rank"]]
    if not kw["traverse"]: a.append("--no-traverse")
    if kw.get("host"): a += ["--host-taxids", ",".join(map(str, kw["host"]))]
    if kw.get("exclude"): a += ["--exclude-taxids", ",".join(map(str, kw["exclude"]))]
    rkw = dict(m1=files[0], m2=files[1]) if paired else dict(u=files[0])
    want = O.ref_classify(d+"/idx", d+"/w.tsv", d+"/w.rep", extra=a, **rkw)
    e = emu.Emu(d+"/idx")
    names, ql, seq, off, seeds, pr = reads.load(files, False)
    for ver in (2, 1, 3):
        # 3 = k_search2's body over the tables a device index derives at load time: text verification (random sample
        # rate), wide ftab, dense resolve table — index-dependent ftab / sample rates (-t / -o) included
        if ver == 3:
            if "-t" in extra and int(extra[extra.index("-t") + 1]) > 10: continue
            ftc = int(extra[extra.index("-t") + 1]) if "-t" in extra else 10
            emu.lib().emu_set_verify_min_run(int(rng.integers(0, 4)))
            # small ranges against the text (DIndex::multiRows; takes effect with the samples at every row, rate 0 — drawn a third of the time)
            emu.lib().emu_set_multi_verify.argtypes = [__import__("ctypes").c_uint32, __import__("ctypes").c_uint32]
            emu.lib().emu_set_multi_verify(int(rng.choice([0, 2, 4, 8, 15])), int(rng.integers(0, 5)))
            emu.lib().emu_set_isa_extra(int(rng.choice([0, 3, 1])))         # (the inverse sample coarser than the SA sample, DIndex::isaRate)
            emu.lib().emu_textify(e.h, int(rng.choice([0, 0, 1, 2, 3, 5])))
            pl = int(rng.integers(0, 2))
            emu.lib().emu_planify(e.h, pl)                            # the one-chain-per-lane form over the occurrence planes, or the sides
            if pl: emu.lib().emu_planify2(e.h, int(rng.integers(0, 3)) > 0)   # ... and two bases per step over the pair planes
            emu.lib().emu_set_self_records(int(rng.integers(0, 4)) > 0)        # strand records made by the search kernel itself, or by pack_body
            emu.lib().emu_widen(e.h, ftc + int(rng.integers(1, 3)))
            emu.lib().emu_densify(e.h, int(rng.integers(0, 3)))
            emu.lib().emu_set_pos_shift(int(rng.choice([14, 8, 4])))
            emu.lib().emu_posify(e.h, int(rng.integers(0, 3)) > 0)      # hits in the position form where the tables allow it (text + resolve table at every row; else a no-op)
        emu.lib().emu_set_search_version(2 if ver == 3 else ver)
        # the common-case post / score kernels in front of the general ones (as the device layer runs them), or — version 1 —
        # any combination, the general kernels alone included
        fp, fs = (1, 1) if ver != 1 else (int(rng.integers(0, 2)), int(rng.integers(0, 2)))
        emu.lib().emu_set_fast_kernels(fp, fs)
        rows, n_rows, s2_ = e.classify(seq, off, seeds, paired=pr, **kw)
        got = reads.format_tsv(e.seqid, names, ql, rows, n_rows, s2_)
        if got == want and ver == 2:                      # the report (counters, observed tuples, EM) from the same rows
            hix = capi.Index(d+"/idx", host_only=True); rep = capi.Report(hix); orc = O.Oracle(d+"/idx")
            rep.add(rows, n_rows, TR.max_scores(orc, seq, off, pr), kw["k"]); rep.write(d+"/m.rep"); rep.close(); hix.close()
            if open(d+"/m.rep").read() != open(d+"/w.rep").read():
                got = "REPORT DIFFERS"
                print(common.first_diff(open(d+"/m.rep").read(), open(d+"/w.rep").read()), flush=True)
        if got != want:
            bad += 1
            print("MISMATCH iter", it-1, "seed", seed0+it-1, "ver", ver, kw, "G,L,gs,div", G, L, gs, div, "rl", rl, "paired", paired, extra, d, flush=True)
            print(common.first_diff(got, want), flush=True)
            break
    e.close()
    if got == want:
        subprocess.run(["rm","-rf",d])
emu.lib().emu_set_fast_kernels(1, 1); emu.lib().emu_set_self_records(1)
print("iterations", it, "bad", bad)
