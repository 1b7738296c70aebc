"""The C-ABI library loads and exports every symbol include/centrifuge_amd.h
declares; without a GPU the compute entry points fail loudly (no CPU path)."""
import ctypes as C
import os
import re

import pytest

import common
from centrifuge_amd import capi


def declared_symbols():
    syms = set()
    for h in ("centrifuge_amd.h", "centrifuge_amd_build.h"):
        hdr = open(os.path.join(common.ROOT, "include", h)).read()
        hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
        syms |= set(re.findall(r"\b(cf_[a-z0-9_]+|centrifuge|centrifuge_build)\s*\(", hdr))
    return sorted(syms)


def test_library_exports_every_declared_symbol():
    L = C.CDLL(capi.LIB_PATH)
    syms = declared_symbols()
    assert len(syms) >= 35
    for s in syms:
        assert hasattr(L, s), "missing export " + s
    assert set(capi.EXPORTS) <= set(syms)


def test_host_only_index_and_formatting():
    d, _ = common.golden("example")
    ix = capi.Index(os.path.join(d, "idx"), host_only=True)
    assert ix.text_len == 1073 and ix.num_refs == 2 and ix.sa_width == 2
    assert ix.seqid(0, 9646) == "gi|4" and ix.seqid(1, 9913) == "gi|7"
    assert ix.seqid(capi.MERGED, 40674) == "class"
    L = ix.L
    assert L.cf_tax_name(ix.h, 9913) == b"Bos taurus"
    assert L.cf_tax_size(ix.h, 9646) == 556
    assert L.cf_tax_rank_string(L.cf_tax_rank(ix.h, 9913)) == b"species"
    # a classifier needs the device-resident index
    with pytest.raises(capi.CfError):
        capi.Classifier(ix)
    ix.close()


def test_builder_fails_loudly_without_a_device():
    """cf_build_index has no CPU path either"""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(capi.CfError):
        capi.build_index("/tmp/cf_nodev", "/nonexistent", "/nonexistent", fasta=["/nonexistent.fa"])


def test_seed_function_matches_reference_formula():
    import numpy as np
    L = capi.lib()
    s = np.array([0, 1, 2, 3, 4, 0, 1], dtype=np.uint8)
    r = (101 * 59 * 61 * 67 * 71 * 73 * 79 * 83) & 0xffffffff
    for i, p in enumerate(s):
        r ^= (int(p) << ((i & 15) << 1)) & 0xffffffff
    for i in range(len(s)):
        r ^= ord("I") << ((i & 3) << 3)
    for i, ch in enumerate(b"ab"):
        r ^= ch << ((i & 3) << 3)
    assert L.cf_gen_rand_seed(s.ctypes.data, None, len(s), b"ab/1", 4, 0) == r


def test_no_device_is_a_loud_error():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    d, _ = common.golden("example")
    with pytest.raises(capi.CfError) as e:
        capi.Index(os.path.join(d, "idx"))
    assert "no CPU path" in str(e.value) or "no HIP device" in str(e.value)


def test_damaged_index_files_fail_cleanly():
    """truncated / missing / foreign index files: a status code and a message, never a crash
    (the reference prints and sometimes carries on, bt2_io.h:66-74; we stop)"""
    import shutil
    import tempfile
    d, _ = common.golden("synth_small")
    with tempfile.TemporaryDirectory() as t:
        for ext in "1234":
            shutil.copy(os.path.join(d, "idx.%s.cf" % ext), os.path.join(t, "idx.%s.cf" % ext))
        base = os.path.join(t, "idx")
        capi.Index(base, host_only=True).close()                         # intact copy opens
        # 1. truncated primary file
        full = open(base + ".1.cf", "rb").read()
        open(base + ".1.cf", "wb").write(full[:len(full) // 3])
        with pytest.raises(capi.CfError):
            capi.Index(base, host_only=True)
        # 2. wrong endianness / not an index
        open(base + ".1.cf", "wb").write(b"\x00\x00\x00\x01" + full[4:])
        with pytest.raises(capi.CfError):
            capi.Index(base, host_only=True)
        open(base + ".1.cf", "wb").write(full)
        # 3. taxonomy file cut in the middle of the uid table
        tax = open(base + ".3.cf", "rb").read()
        open(base + ".3.cf", "wb").write(tax[:40])
        with pytest.raises(capi.CfError):
            capi.Index(base, host_only=True)
        open(base + ".3.cf", "wb").write(tax)
        # 4. missing primary file
        os.remove(base + ".1.cf")
        with pytest.raises(capi.CfError) as ei:
            capi.Index(base, host_only=True)
        assert "cannot open" in str(ei.value) or "I/O" in str(ei.value)


def _call(fn, args):
    av = (C.c_char_p * len(args))(*[a.encode() for a in args])
    fn.restype, fn.argtypes = C.c_int, [C.c_int, C.POINTER(C.c_char_p)]
    return fn(len(args), av)


def test_program_entry_points_return_instead_of_exiting(tmp_path, capfd):
    """centrifuge(argc, argv) / centrifuge_build(argc, argv): the reference's C symbols
    (centrifuge.cpp:3338, centrifuge_build.cpp:550).  Every failure is a return code + stderr
    message in this very process, and the call can be repeated."""
    L = C.CDLL(capi.LIB_PATH)
    d, _ = common.golden("example")
    for _ in range(2):
        assert _call(L.centrifuge, ["centrifuge-class", "--help"]) == 0
        assert _call(L.centrifuge, ["centrifuge-class", "-f", "-x", os.path.join(d, "nonexistent"), "-U", os.path.join(d, "reads.fa")]) == 1
        assert _call(L.centrifuge, ["centrifuge-class", "-x", os.path.join(d, "idx")]) == 1
        assert _call(L.centrifuge, ["centrifuge-class", "-f", "-k", "0", "-x", os.path.join(d, "idx"), "-U", os.path.join(d, "reads.fa")]) == 1
        assert _call(L.centrifuge, ["centrifuge-class", "--bogus"]) == 1
        assert _call(L.centrifuge, ["centrifuge-class", "-f", "--classification-rank", "bogus", "-x", os.path.join(d, "idx"), "-U", os.path.join(d, "reads.fa")]) == 1
        assert _call(L.centrifuge, ["centrifuge-class", "-f", "-p", "0", "-x", os.path.join(d, "idx"), "-U", os.path.join(d, "reads.fa")]) == 1
        assert _call(L.centrifuge_build, ["centrifuge-build-bin", "--help"]) == 0
        assert _call(L.centrifuge_build, ["centrifuge-build-bin", "only_one_positional"]) == 1
        assert _call(L.centrifuge_build, ["centrifuge-build-bin", "--bogus"]) == 1
    err = capfd.readouterr().err
    assert "Could not locate a Centrifuge index" in err and "Must specify at least one read input" in err
    assert "-k arg must be at least 1" in err and "unrecognized option" in err
    assert "(--classification-rank) should be one of strain, species" in err and "-p/--threads arg must be at least 1" in err
    # the ingest-only mode needs no device: a complete run through the entry point, twice
    fq = tmp_path / "r.fq"
    fq.write_text("@a\nACGT\n+\nIIII\n@b\nGGNA\n+\nI#II\n")
    for _ in range(2):
        assert _call(L.centrifuge, ["centrifuge-class", "-q", "--dump-reads", "-U", str(fq)]) == 0
    # without a GPU the classification itself is a loud error, not a CPU fallback
    import torch
    if not torch.cuda.is_available():
        assert _call(L.centrifuge, ["centrifuge-class", "-f", "-x", os.path.join(d, "idx"), "-U", os.path.join(d, "reads.fa"), "-S", str(tmp_path / "o.tsv")]) == 1
        assert "no HIP device" in capfd.readouterr().err


def test_headers_are_plain_c_and_link(tmp_path):
    """the drop-in boundary is a C ABI: both headers compile as C99 (-pedantic) and a C program links the library"""
    import subprocess
    src = tmp_path / "hdr.c"
    src.write_text('#include "centrifuge_amd.h"\n#include "centrifuge_amd_build.h"\n#include <stdio.h>\n'
                   "int main(void) { cf_params p; if (cf_params_default(&p) != CF_OK) return 2;\n"
                   '  printf("%d %d %s\\n", p.khits, p.min_hitlen, cf_strerror(CF_ERR_NO_DEVICE)); return 0; }\n')
    exe = tmp_path / "hdr"
    libdir = os.path.dirname(capi.LIB_PATH)
    r = subprocess.run(["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-I", os.path.join(common.ROOT, "include"), str(src),
                        "-o", str(exe), "-L", libdir, "-lcentrifuge_amd", "-Wl,-rpath," + libdir], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    out = subprocess.run([str(exe)], capture_output=True, text=True)
    assert out.returncode == 0 and out.stdout.startswith("5 22 ")


def test_table_planner_respects_room_and_constraints():
    """cf_debug_plan_tables (the derived-table planner of cf_index_open, no device): whatever it picks fits the room it is given,
    switched-off tables stay off, a fixed value is made while it fits, more room never costs more (the model's cost)"""
    from centrifuge_amd import capi
    sizes = dict(wide=lambda K: (8 << (2 * K)) if K else 0, planes=lambda n: ((n // 4 + 1 + 95) // 96) * 384)
    last = None
    for gb in (1, 4, 8, 16, 32, 64, 100, 160, 250):
        room = gb * 10 ** 9
        p = capi.plan_tables(int(8.59e9), room)
        sides = ((int(8.59e9) // 4 + 1 + 95) // 96) * 128              # (a plan that lets the sides go may spend their room as well)
        assert p["bytes"] <= room + (sides if p["drop_sides"] else 0) and (p["pair"] == 0 or p["planes"] == 1)
        assert not p["drop_sides"] or p["planes"] == 1
        assert capi.plan_tables(int(8.59e9), room, sides=1)["drop_sides"] == 0 and capi.plan_tables(int(8.59e9), room, sides=1)["bytes"] <= room
        assert p["bytes"] >= sizes["wide"](p["K"]) + (sizes["planes"](int(8.59e9)) if p["planes"] else 0)
        if last is not None:
            assert p["cost"] <= last + 1e-9
        last = p["cost"]
    full = capi.plan_tables(int(8.59e9), 250 * 10 ** 9)
    assert (full["K"], full["text_rate"], full["planes"], full["resolve_rate"], full["pair"]) == (16, 0, 1, 0, 1)      # (samples at every row where there is room: round 5)
    assert capi.plan_tables(int(8.59e9), 170 * 10 ** 9)["text_rate"] == 0                                            # (round 6: the inverse sample is kept coarse — 147 GB of tables instead of 187)
    assert capi.plan_tables(int(8.59e9), 130 * 10 ** 9)["text_rate"] == 1                                            # ... and at every 2nd where there is not
    assert capi.plan_tables(int(8.59e9), 250 * 10 ** 9, occ_planes=-1)["planes"] == 0
    assert capi.plan_tables(int(8.59e9), 250 * 10 ** 9, pair_planes=-1, wide_ftab_chars=-1)["K"] == 0
    q = capi.plan_tables(int(8.59e9), 250 * 10 ** 9, resolve_rate=3, text_verify_rate=3)
    assert (q["resolve_rate"], q["text_rate"]) == (2, 3)
    assert capi.plan_tables(int(8.59e9), 250 * 10 ** 9, resolve_rate=-1)["resolve_rate"] == 4
    tiny = capi.plan_tables(320000, 64 << 20, wide_ftab_chars=12)                 # 134 MB of wide ftab do not fit: the rest is made
    assert tiny["K"] == 0 and tiny["planes"] == 1 and tiny["text_rate"] == 0
    assert capi.plan_tables(320000, 10 ** 9, wide_ftab_chars=12)["K"] == 12
    assert capi.plan_tables(int(8.59e9), 0)["bytes"] == 0
    # small ranges against the text want the samples at every row (94 GB at 8.6 Gbp instead of 48): granted when the model — which
    # prices what they save on a repeat-rich collection (without a device to probe: the stand-in's repeat fraction) — puts that
    # plan below the usual one.  With room for everything they come on top; in 150 GB they are worth more than the pair planes;
    # in 110 GB they do not fit beside the planes and the wide ftab, and the plan is the usual one
    def tables(p):
        return {k: p[k] for k in ("K", "text_rate", "planes", "resolve_rate", "pair", "drop_sides")}
    roomy = capi.plan_tables(int(8.59e9), 240 * 10 ** 9, small_range_rows=4)
    assert roomy["text_rate"] == 0 and (roomy["K"], roomy["planes"], roomy["pair"], roomy["resolve_rate"]) == (16, 1, 1, 0)
    mid = capi.plan_tables(int(8.59e9), 150 * 10 ** 9, small_range_rows=4)
    assert mid["text_rate"] == 0 and (mid["K"], mid["planes"], mid["pair"]) == (16, 1, 0) and mid["bytes"] <= 150 * 10 ** 9
    tight = capi.plan_tables(int(8.59e9), 110 * 10 ** 9, small_range_rows=4)
    assert tight["text_rate"] > 0 and tables(tight) == tables(capi.plan_tables(int(8.59e9), 110 * 10 ** 9))
    assert capi.plan_tables(int(8.59e9), 240 * 10 ** 9, small_range_rows=4, text_verify_rate=2)["text_rate"] == 2      # a fixed rate wins
    # off, and automatic without a probe (no device: the repeat fraction is unknown), are the usual plan
    for gb in (110, 150, 240):
        usual = capi.plan_tables(int(8.59e9), gb * 10 ** 9)
        assert capi.plan_tables(int(8.59e9), gb * 10 ** 9, small_range_rows=-1) == usual
    # a plan that drops the sides may spend their room only on the tables made after they are gone (wide ftab, pair planes): the
    # planes, the text tables and the resolve table are built while the sides are still there, and must fit without their bytes
    n5 = int(1.03e11)
    sides5 = ((n5 // 4 + 1 + 95) // 96) * 128
    for gb in (120, 160, 200, 230):
        p = capi.plan_tables(n5, gb * 10 ** 9)
        if p["drop_sides"]:
            wide = (8 << (2 * p["K"])) + 16 if p["K"] else 0
            pair = ((n5 + 64) // 64 + 1) * 256 if p["pair"] else 0
            assert p["bytes"] - wide - pair <= gb * 10 ** 9 and p["bytes"] <= gb * 10 ** 9 + sides5
