import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tools"), os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def _ensure_built():
    """The tests need the C-ABI library and the front-end binaries.  They are built in-tree by
    __graft_entry__.build(); a checkout that has not been built yet gets built here (hipcc
    cross-compiles gfx950 without a GPU)."""
    import subprocess
    lib = os.path.join(ROOT, "centrifuge_amd", "libcentrifuge_amd.so")
    cli = os.path.join(ROOT, "centrifuge_amd", "bin", "centrifuge-class")
    if not (os.path.exists(lib) and os.path.exists(cli)):
        subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "centrifuge_amd", "csrc"), "all"])


# The suite's switches (environment, read by the TEST layers only — capi.Index / tests/test_gpu_cli.py —, never by the library):
#   CF_TEST_SMALL_RANGE_ROWS=4   every index that does not say otherwise is opened with cf_index_options::small_range_rows = 4
#                                (centrifuge-class: --small-range-rows 4), so that the WHOLE `-m gpu` suite runs over the search
#                                path that finishes small ranges against the text; -1 = with it off everywhere (the default is
#                                automatic: on for the repeat-rich indexes of tests/test_gpu_scale.py, off for the others)
#   CF_DEBUG_KNOBS=1             (set here for every test) the library reads its CF_* debug knobs — forced table combinations, kernel
#                                variants: centrifuge_amd/csrc/cf_knobs.hpp — only under this gate; a user's environment cannot reach them
#   CF_EMU_WAVE64=1              (tests/fuzz/fuzz_classify.py; tests/test_emu_wave64.py switches the same thing through emu.use_wave64)
#                                the CPU harness as a wavefront of 64 lanes — libcfemu64.so, the search kernels' cross-lane code off the GPU
os.environ.setdefault("CF_DEBUG_KNOBS", "1")


def pytest_configure(config):
    _ensure_built()
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run by the driver with -m gpu)")


def _have_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return os.path.exists("/dev/kfd")


def pytest_collection_modifyitems(config, items):
    if _have_gpu():
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for it in items:
        if "gpu" in it.keywords:
            it.add_marker(skip)
