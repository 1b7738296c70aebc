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
        syms |= set(re.findall(r"\b(cf_[a-z0-9_]+)\s*\(", hdr))
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
