#!/bin/bash
# Validation of the tree with the one-pass k_count: PMC passes + headline line first (cheap), then the whole GPU suite.
set -u
TAG=${1:-r3o}
O=$PWD/gpurun_out/$TAG; mkdir -p $O
bash tools/round3_pmc.sh $TAG > $O/pmc_run.log 2>&1
python - <<P
import json
j=json.loads([l for l in open("$O/bench.json") if l.startswith("{")][-1])
print("cfg2 value %.3e ms/step %.2f" % (j["value"], j["ms_per_step"]), {k: round(v,2) for k,v in j["kernels_ms"].items()}, j["cpu_baseline"].get("gpu_rows_identical_on_sample"), j["roofline"]["traffic"])
P
timeout 660 python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1; grep -E "passed|failed|error" $O/pytest_gpu.log | tail -n 3
