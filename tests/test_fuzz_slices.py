"""A short slice of every offline fuzzer (tests/fuzz/*.py) inside the CPU suite: each runs for a few
seconds on fresh seeds — the wall clock picks them — and must report no mismatch.  The fuzzers themselves run for as long as
one lets them (`python tests/fuzz/fuzz_classify.py 600`); CF_FUZZ_SLICE_S lengthens the slices here."""
import os
import re
import subprocess
import sys
import time

import pytest

import common
from oracle import oracle as O

FUZZ = os.path.join(common.ROOT, "tests", "fuzz")
SLICE_S = float(os.environ.get("CF_FUZZ_SLICE_S", "6"))


@pytest.mark.skipif(not O.have_ref(), reason="oracle/_ref (the compiled reference) is not built")
@pytest.mark.parametrize("script,seeded", [("fuzz_classify.py", True), ("fuzz_classify.py:wave64", True), ("fuzz_taxonomy.py", True), ("fuzz_restore.py", False),
                                           ("fuzz_build_input.py", False), ("fuzz_report_tools.py", False), ("fuzz_ingest.py", False)])
def test_fuzzer_slice_finds_no_mismatch(script, seeded):
    env = dict(os.environ)
    if script.endswith(":wave64"):               # the same fuzzer over the 64-lane build of the harness (round 6)
        script = script.split(":")[0]
        env["CF_EMU_WAVE64"] = "1"
    if script == "fuzz_ingest.py" and not os.path.exists(os.path.join(common.ROOT, "centrifuge_amd", "bin", "centrifuge-class")):
        pytest.skip("the front end is not built")
    args = [sys.executable, os.path.join(FUZZ, script), str(SLICE_S)]
    if seeded:
        args.append(str(int(time.time()) % 1000000 * 1000))
    r = subprocess.run(args, capture_output=True, text=True, cwd=os.path.join(common.ROOT, "tests"), timeout=600, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    m = re.search(r"iterations\s+(\d+)\s+bad\s+(\d+)", r.stdout)
    assert m, r.stdout[-2000:]
    assert int(m.group(1)) >= 1 and int(m.group(2)) == 0, r.stdout[-3000:]
