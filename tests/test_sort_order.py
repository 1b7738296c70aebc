"""The libstdc++ std::sort restatement of the post kernel (std_sort_hits, cf_kernels.hpp: introsort with threshold 16,
median-of-3 to first, unguarded Hoare partition, heapsort fallback, final insertion sort) against std::sort itself
(g++'s libstdc++, the library the reference is built with) under the reference's comparator (compareBWTHits,
classifier.h:1058-1086).  The comparator leaves many hits equivalent, and what std::sort does with equivalent elements
decides the order of the hit map and with it the printed rows (classifier.h:267, ds.h:775-779)."""
import ctypes as C

import numpy as np
import pytest

from centrifuge_amd.capi import HIT_DTYPE
from emu import emu


def both(h):
    L = emu.lib()
    L.emu_sort_hits.argtypes = [C.c_void_p, C.c_uint32]
    L.emu_std_sort_hits.argtypes = [C.c_void_p, C.c_uint32]
    a, b = h.copy(), h.copy()
    L.emu_sort_hits(a.ctypes.data, len(a))
    L.emu_std_sort_hits(b.ctypes.data, len(b))
    return a, b


def random_hits(rng, n, style):
    h = np.zeros(n, dtype=HIT_DTYPE)
    if style == "ties":             # few distinct (len, size) classes: almost everything ties
        lens = rng.choice([10, 15, 21, 22, 23, 30], n)
        sizes = rng.choice([0, 1, 2, 5], n)
    elif style == "ratio":          # equal len/size ratios with different values (the cross-multiplied branch)
        k = rng.integers(1, 5, n)
        lens, sizes = 11 * k, 2 * k
    elif style == "sorted":
        lens, sizes = np.sort(rng.integers(1, 120, n))[::-1], np.sort(rng.integers(0, 50, n))
    else:
        lens, sizes = rng.integers(1, 120, n), rng.integers(0, 4000, n)
    h["len"] = lens
    h["top"] = rng.integers(0, 1 << 39, n)
    h["bot"] = h["top"] + sizes.astype(np.uint64)
    h["bwoff"] = np.arange(n)       # the tag that tells equivalent hits apart
    return h


@pytest.mark.parametrize("style", ["ties", "ratio", "sorted", "random"])
def test_sort_restatement_equals_std_sort(style):
    rng = np.random.default_rng(hash(style) % 1000)
    for n in list(range(0, 40)) + [47, 48, 63, 64, 65, 100, 129, 255]:
        for _ in range(12 if n < 70 else 4):
            h = random_hits(rng, n, style)
            a, b = both(h)
            assert np.array_equal(a["bwoff"], b["bwoff"]), (style, n, a["bwoff"], b["bwoff"])
            assert np.array_equal(a, b)


def test_heapsort_fallback_is_reached_and_agrees():
    """median-of-3 killer-ish inputs: organ-pipe and many-duplicates patterns large enough to exhaust 2*log2(n) partitions"""
    rng = np.random.default_rng(1)
    for n in (200, 255):
        for trial in range(6):
            h = np.zeros(n, dtype=HIT_DTYPE)
            v = np.concatenate([np.arange(n // 2), np.arange(n - n // 2)[::-1]]) if trial % 2 == 0 else rng.integers(0, 3, n)
            h["len"] = 30
            h["top"] = 0
            h["bot"] = v.astype(np.uint64)
            h["bwoff"] = np.arange(n)
            a, b = both(h)
            assert np.array_equal(a, b)
