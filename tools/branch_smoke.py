"""A few seconds on the GPU for the branch: the golden synth index with a 12-base wide ftab (next-pairs masks), every derived
table, three golden cases (single, paired, N-rich if present) against their committed TSV through the slot ABI + smoke()."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import common
from centrifuge_amd import capi, reads
t0 = time.time()
ARCH = sys.argv[1] if len(sys.argv) > 1 else "synth_small"
d, cases = common.golden(ARCH)
base = os.path.join(d, "idx")
ix = capi.Index(base, device=0, wide_ftab_chars=12)
print("open %.2fs" % (time.time() - t0), ix.describe())
n_ok = 0
for c in cases:
    kw, fastq = common.case_kwargs(c["args"])
    names, qlens, seq, off, seeds, paired = reads.load([os.path.join(d, f) for f in c["reads"]], fastq)
    clf = capi.Classifier(ix, **kw)
    b, m, ln = capi.pack_reads(seq, off)
    slot = capi.Slot(clf)
    slot.submit(b, m, ln, np.ascontiguousarray(seeds, dtype=np.uint32), paired=paired)
    rows, first, n_rows, score2, max_score, info = slot.wait()
    ops = slot.opcounts()
    got = reads.format_tsv(ix.seqid, names, qlens, capi.unpack_rows(rows, first, n_rows, kw.get("k", 5)), n_rows, score2)
    want = open(os.path.join(d, c["tsv"])).read()
    print(c["name"], "OK" if got == want else "DIFF", "ftab_wide", ops.n_ftab_wide, "pair", ops.n_pair, "single", ops.n_single, "verify", ops.n_verify)
    n_ok += got == want
    slot.close(); clf.close()
    if time.time() - t0 > 12: break
ix.close()
print("cases ok:", n_ok, "in %.1fs" % (time.time() - t0))
import __graft_entry__ as g
g.smoke()
