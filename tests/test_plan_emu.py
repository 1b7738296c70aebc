"""The device-side batch plan (plan_body / plan_fill_body / plan_maxscore_body) and the row
compaction (compact_body), stepped on the CPU through tests/emu, against the host plan
(makeBatchPlan: Scoring::nFilter scoring.cpp:104-117, centrifuge.cpp:2562-2577) and the
max_score rule (classifier.h:530-536)."""
import numpy as np
import pytest

from centrifuge_amd.capi import ROW_DTYPE
from emu import emu as E


def batch(reads):
    off = np.zeros(len(reads) + 1, dtype=np.uint64)
    off[1:] = np.cumsum([len(r) for r in reads])
    seq = np.concatenate(reads) if off[-1] else np.zeros(0, dtype=np.uint8)
    return seq.astype(np.uint8), off


@pytest.mark.parametrize("seed", range(6))
@pytest.mark.parametrize("paired", [False, True])
def test_plan_bodies_match_host_plan(seed, paired):
    rng = np.random.default_rng(seed)
    reads = []
    for _ in range(2 * int(rng.integers(1, 150))):
        L = int(rng.choice([0, 1, 2, 3, 9, 10, 15, 16, 20, 33, 100, 128, 129, 150, 256, 257, 300, int(rng.integers(0, 400))]))
        r = rng.integers(0, 4, L, dtype=np.uint8)
        mode = int(rng.integers(0, 5))
        if mode == 0 and L:                       # around the 15 % N ceiling
            n = int(0.15 * L) + int(rng.integers(-1, 2))
            if n > 0:
                r[rng.choice(L, min(n, L), replace=False)] = 4
        elif mode == 1:
            r[:] = 4
        elif mode == 2 and L > 4:
            r[rng.integers(0, L, 3)] = 4
        reads.append(r)
    seq, off = batch(reads)
    for ftab in (10, 4, 12):
        assert E.plan_check(seq, off, ftab, paired) == 0


def test_plan_of_an_empty_batch_and_of_empty_reads():
    assert E.plan_check(np.zeros(0, dtype=np.uint8), np.zeros(1, dtype=np.uint64)) == 0
    assert E.plan_check(np.zeros(0, dtype=np.uint8), np.zeros(5, dtype=np.uint64), paired=True) == 0


def test_compaction_body():
    rng = np.random.default_rng(3)
    for k in (1, 5, 50):
        nq = 300
        rows = np.zeros((nq, k), dtype=ROW_DTYPE)
        rows.view(np.uint8)[:] = rng.integers(0, 256, rows.view(np.uint8).shape, dtype=np.uint8)
        n_rows = rng.integers(0, k + 1, nq).astype(np.uint32)
        n_rows[:5] = 0
        n_rows[5:10] = k
        assert E.compact_check(rows, n_rows) == 0
