#!/bin/bash
# Run on the GPU box (through gpurun): kernel-trace stats + PMC passes of bench.py.
# Usage: tools/gpu_profile.sh <tag> [bench args...]   -> gpurun_out/prof_<tag>/
set -u
TAG=${1:-r01}; shift || true
OUT=$PWD/gpurun_out/prof_$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp CF_BENCH_DIR=/tmp/cf_bench_prof
REPO=$PWD
ARGS="--steps 3 --warmup 1 --no-cpu $*"
cd /tmp
python $REPO/bench.py $ARGS > $OUT/bench_plain.json 2> $OUT/bench_plain.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o t -- python $REPO/bench.py $ARGS > $OUT/bench_trace.json 2> $OUT/trace.err
for pmc in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum"; do
  name=$(echo $pmc | tr ' ' '_')
  timeout 600 rocprofv3 --pmc $pmc --output-format csv -d $OUT/pmc_$name -o p -- python $REPO/bench.py $ARGS > $OUT/pmc_$name.json 2> $OUT/pmc_$name.err
done
cd $REPO
python tools/prof_summary.py $OUT > $OUT/summary.txt 2>&1
cat $OUT/summary.txt
