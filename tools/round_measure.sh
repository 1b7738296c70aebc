#!/bin/bash
# Run on the GPU box (through gpurun): the round's closing measurements -> gpurun_out/<tag>/
#   full GPU test suite, bench.py (default: config 2, with the CPU leg), kernel trace + PMC traffic passes.
set -u
TAG=${1:-final}
OUT=$PWD/gpurun_out/$TAG; mkdir -p $OUT
export CF_BENCH_DIR=/tmp/cfb
timeout 1500 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1; grep -E "passed|failed|error" $OUT/pytest_gpu.log | tail -2
timeout 400 python bench.py > $OUT/bench.json 2> $OUT/bench.err; tail -c 300 $OUT/bench.json
PMC_SET=short tools/gpu_profile.sh $TAG > /dev/null 2>&1
cp gpurun_out/prof_$TAG/summary.txt $OUT/rocprofv3_summary.txt 2>/dev/null
grep -E "k_search2_l1<4, false>" $OUT/rocprofv3_summary.txt | head -5
