"""ctypes wrapper of tests/emu/libcfemu.so — the CPU single-step harness of the
kernel bodies (TEST ONLY; see emu.cpp)."""
import ctypes as C
import os
import subprocess
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
LIB = os.path.join(HERE, "libcfemu.so")
# the 64-lane build of the same harness (CF_EMU_WAVE64: the search kernel's wavefront as 64 fibers that meet at the cross-lane
# primitives, cf_platform.hpp); use_wave64(True) makes lib() — and with it every Emu made afterwards — that build
LIB64 = os.path.join(HERE, "libcfemu64.so")
_wave64 = False

import sys
sys.path.insert(0, ROOT)
from centrifuge_amd.capi import Params, OpCounts, ROW_DTYPE, HIT_DTYPE, make_params  # noqa: E402


def use_wave64(on):
    """switch lib() between the one-lane and the 64-lane build; returns the previous setting"""
    global _wave64, _lib
    was = _wave64
    if bool(on) != _wave64:
        _wave64, _lib = bool(on), None
    return was


def build():
    global LIB
    LIB = LIB64 if _wave64 else os.path.join(HERE, "libcfemu.so")
    src = [os.path.join(HERE, "emu.cpp"), os.path.join(ROOT, "centrifuge_amd/csrc/cf_index.cpp")]
    deps = src + [os.path.join(ROOT, "centrifuge_amd/csrc", f) for f in
                  ("cf_kernels.hpp", "cf_platform.hpp", "cf_plan.hpp", "cf_index.hpp", "cf_restore.hpp", "cf_inspect_fasta.hpp", "cf_textio.hpp")]
    if os.path.exists(LIB) and all(os.path.getmtime(LIB) >= os.path.getmtime(d) for d in deps):
        return
    # built under a lock and moved into place: several test processes (pytest -n) may get here at once
    import fcntl
    with open(LIB + ".lock", "w") as lk:
        fcntl.flock(lk, fcntl.LOCK_EX)
        if os.path.exists(LIB) and all(os.path.getmtime(LIB) >= os.path.getmtime(d) for d in deps):
            return
        tmp = "%s.%d.tmp" % (LIB, os.getpid())
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-Wall", "-Wno-unused-function",
                               "-Wno-unknown-pragmas", "-fno-strict-aliasing"] + (["-DCF_EMU_WAVE64=1"] if _wave64 else []) + ["-o", tmp] + src)
        os.replace(tmp, LIB)


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(LIB)
        L.emu_open.restype = C.c_void_p
        L.emu_open.argtypes = [C.c_char_p]
        L.emu_close.argtypes = [C.c_void_p]
        L.emu_num_taxa.restype = C.c_uint64
        L.emu_num_taxa.argtypes = [C.c_void_p]
        L.emu_taxon_id.restype = C.c_uint64
        L.emu_taxon_id.argtypes = [C.c_void_p, C.c_uint64]
        L.emu_format_seqid.restype = C.c_char_p
        L.emu_format_seqid.argtypes = [C.c_void_p, C.c_uint32, C.c_uint64]
        L.emu_rank.restype = C.c_uint64
        L.emu_rank.argtypes = [C.c_void_p, C.c_int, C.c_uint64]
        L.emu_resolve.restype = C.c_uint32
        L.emu_resolve.argtypes = [C.c_void_p, C.c_uint64]
        L.emu_classify.restype = C.c_int
        L.emu_classify.argtypes = [C.c_void_p, C.POINTER(Params), C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64,
                                   C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.emu_search.restype = C.c_int
        L.emu_search.argtypes = [C.c_void_p, C.POINTER(Params), C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p,
                                 C.c_uint32, C.c_void_p]
        L.emu_sort_hits.argtypes = [C.c_void_p, C.c_uint32]
        L.emu_set_search_version.argtypes = [C.c_int]
        L.emu_set_rows_cap.argtypes = [C.c_uint64]
        L.emu_set_verify_min_run.argtypes = [C.c_uint32]
        L.emu_set_walk_version.argtypes = [C.c_int]
        L.emu_set_lazy_hits.argtypes = [C.c_uint32]
        L.emu_set_fast_kernels.argtypes = [C.c_int, C.c_int]
        L.emu_set_self_records.argtypes = [C.c_int]
        L.emu_set_count_slot_bits.argtypes = [C.c_uint]
        L.emu_last_slow.argtypes = [C.c_void_p, C.c_void_p]
        L.emu_planify.restype = C.c_int
        L.emu_planify.argtypes = [C.c_void_p, C.c_int]
        L.emu_num_rows.restype = C.c_uint64
        L.emu_num_rows.argtypes = [C.c_void_p]
        L.emu_planify2.restype = C.c_int
        L.emu_planify2.argtypes = [C.c_void_p, C.c_int]
        L.emu_set_wide_cap.argtypes = [C.c_uint64]
        L.emu_textify.restype = C.c_int
        L.emu_textify.argtypes = [C.c_void_p, C.c_int]
        L.emu_widen.restype = C.c_int
        L.emu_widen.argtypes = [C.c_void_p, C.c_int]
        L.emu_drop_sides.restype = C.c_int
        L.emu_drop_sides.argtypes = [C.c_void_p, C.c_int]
        L.emu_densify.restype = C.c_int
        L.emu_densify.argtypes = [C.c_void_p, C.c_int]
        L.emu_plan_check.restype = C.c_int
        L.emu_plan_check.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_int, C.c_int]
        L.emu_compact_check.restype = C.c_int
        L.emu_compact_check.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32]
        L.emu_restore.restype = C.c_int
        L.emu_restore.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint64]
        L.emu_inspect_fasta.restype = C.c_int
        L.emu_inspect_fasta.argtypes = [C.c_void_p, C.c_uint32, C.c_int, C.c_char_p]
        L.emu_posify.restype = C.c_int
        L.emu_posify.argtypes = [C.c_void_p, C.c_int]
        L.emu_set_pos_shift.argtypes = [C.c_uint32]
        L.emu_set_isa_extra.argtypes = [C.c_uint32]
        L.emu_walk_max.restype = C.c_uint32
        L.emu_walk_max.argtypes = [C.c_void_p]
        L.emu_wave_collectives.restype = C.c_uint64
        L.emu_wave_lanes.restype = C.c_int
        L.emu_set_rev_words.argtypes = [C.c_int]
        _lib = L
    return _lib


def plan_check(seq, off, ftab_chars=10, paired=False):
    seq = np.ascontiguousarray(seq, dtype=np.uint8)
    off = np.ascontiguousarray(off, dtype=np.uint64)
    return lib().emu_plan_check(seq.ctypes.data, off.ctypes.data, len(off) - 1, ftab_chars, int(paired))


def compact_check(rows, n_rows):
    rows = np.ascontiguousarray(rows)
    n_rows = np.ascontiguousarray(n_rows, dtype=np.uint32)
    return lib().emu_compact_check(rows.ctypes.data, n_rows.ctypes.data, rows.shape[1], rows.shape[0])


class Emu:
    def __init__(self, basename):
        self.L = lib()
        self.h = self.L.emu_open(basename.encode())
        if not self.h:
            raise RuntimeError("emu_open failed")

    def close(self):
        if self.h:
            self.L.emu_close(self.h)
            self.h = None

    def restore(self, n, shift=4):
        """2-bit packed joined text through the restore kernels' bodies (text length n)."""
        out = np.zeros(n // 4 + 1, dtype=np.uint8)
        rc = self.L.emu_restore(self.h, shift, out.ctypes.data, out.size)
        if rc:
            raise RuntimeError("emu_restore failed (%d)" % rc)
        return out

    def inspect_fasta(self, path, across=60, shift=4):
        rc = self.L.emu_inspect_fasta(self.h, shift, across, path.encode())
        if rc:
            raise RuntimeError("emu_inspect_fasta failed (%d)" % rc)

    def seqid(self, u, t):
        return self.L.emu_format_seqid(self.h, int(u), int(t)).decode("latin1")

    def classify(self, seq, off, seeds, paired=False, ops=None, counts=False, **kw):
        p = make_params(**kw)
        seq = np.ascontiguousarray(seq, dtype=np.uint8)
        off = np.ascontiguousarray(off, dtype=np.uint64)
        seeds = np.ascontiguousarray(seeds, dtype=np.uint32)
        n_reads = len(off) - 1
        nq = n_reads // 2 if paired else n_reads
        rows = np.zeros((nq, p.khits), dtype=ROW_DTYPE)
        n_rows = np.zeros(nq, dtype=np.uint32)
        score2 = np.zeros(nq, dtype=np.uint32)
        cnt = np.zeros(2 * self.L.emu_num_taxa(self.h), dtype=np.uint64) if counts else None
        rc = self.L.emu_classify(self.h, C.byref(p), seq.ctypes.data, off.ctypes.data, seeds.ctypes.data, n_reads,
                                 int(paired), rows.ctypes.data, n_rows.ctypes.data, score2.ctypes.data,
                                 C.addressof(ops) if ops is not None else None,
                                 cnt.ctypes.data if counts else None)
        if rc:
            raise RuntimeError("emu_classify failed")
        if counts:
            return rows, n_rows, score2, cnt
        return rows, n_rows, score2

    def search(self, codes, max_hits=512, **kw):
        """hit lists of one read after search + extend / twin / trim (emu_search)"""
        p = make_params(**kw)
        codes = np.ascontiguousarray(codes, dtype=np.uint8)
        hf, hr = np.zeros(max_hits, dtype=HIT_DTYPE), np.zeros(max_hits, dtype=HIT_DTYPE)
        n = (C.c_uint32 * 2)()
        rc = self.L.emu_search(self.h, C.byref(p), codes.ctypes.data, len(codes), hf.ctypes.data, hr.ctypes.data, max_hits, n)
        if rc:
            raise RuntimeError("emu_search failed")
        return hf[:n[0]], hr[:n[1]]
