#!/bin/bash
# GPU run 2 of round 3: the GPU suite with the common-case kernels + tail streams, then config 2 three ways
set -u
O=gpurun_out/r3b; mkdir -p $O
export TMPDIR=/tmp CF_BENCH_DIR=/tmp/cfb
timeout 900 python -m pytest tests -x -q -m gpu 2>&1 | tail -15 > $O/pytest_gpu.log
timeout 300 python bench.py --other-configs "" --steps 20 --warmup 5 > $O/bench_cfg2.json 2> $O/bench_cfg2.err
CF_TAIL_STREAM=0 timeout 200 python bench.py --other-configs "" --no-cpu --steps 20 --warmup 5 > $O/bench_cfg2_notail.json 2> $O/bench_cfg2_notail.err
CF_BLOCKS_PER_CU=5 timeout 200 python bench.py --other-configs "" --no-cpu --steps 20 --warmup 5 > $O/bench_cfg2_5blocks.json 2> $O/bench_cfg2_5blocks.err
CF_POST_FAST=0 CF_SCORE_FAST=0 CF_TAIL_STREAM=0 timeout 200 python bench.py --other-configs "" --no-cpu --steps 20 --warmup 5 > $O/bench_cfg2_general_only.json 2> $O/bench_cfg2_general_only.err
cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/trace -o t -- python $GRAFT_REPO_ROOT/bench.py --other-configs "" --no-cpu --steps 10 --warmup 3 > $GRAFT_REPO_ROOT/$O/bench_trace.json 2> $GRAFT_REPO_ROOT/$O/trace.err
cd $GRAFT_REPO_ROOT
find $O -name "*.csv" -size +2M -delete
python - <<'P'
import json,glob
for f in sorted(glob.glob("gpurun_out/r3b/bench_*.json")):
    try:
        j=json.loads([l for l in open(f) if l.startswith("{")][-1])
        print(f.split("/")[-1], "value %.3e ms/step %.2f kernels %s general %s" % (j["value"], j["ms_per_step"], {k: round(v,2) for k,v in j["kernels_ms"].items()}, j.get("general_kernel_queries")))
    except Exception as e: print(f, "unreadable", e)
P
tail -n 4 $O/pytest_gpu.log
