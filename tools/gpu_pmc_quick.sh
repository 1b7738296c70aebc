#!/bin/bash
export CF_DEBUG_KNOBS=1   # the library reads its CF_* knobs only under this gate (csrc/cf_knobs.hpp)
# Run on the GPU box (through gpurun): two short PMC passes (instruction mix; wave / wait cycles) of bench.py under the
# caller's environment (kernel variants are selected by CF_* variables).  Usage: tools/gpu_pmc_quick.sh <tag> [bench args...]
set -u
TAG=${1:-q}; shift || true
OUT=$PWD/gpurun_out/pmc_$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp CF_BENCH_DIR=/tmp/cf_bench_prof
REPO=$PWD
ARGS="--steps 3 --warmup 1 --no-cpu $*"
cd /tmp
PASSES=("SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM_RD" "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY")
for pmc in "${PASSES[@]}" ; do
  name=$(echo $pmc | tr ' ' '_')
  timeout 240 rocprofv3 --pmc $pmc --output-format csv -d $OUT/pmc_$name -o p -- python $REPO/bench.py $ARGS > $OUT/pmc_$name.json 2> $OUT/pmc_$name.err
done
cd $REPO
python tools/prof_summary.py $OUT 2>&1 | grep -E "k_search|k_post|k_score|k_walk3|k_pack|k_plan|scan" > $OUT/summary.txt
find $OUT -name "*.csv" -size +2M -delete
cat $OUT/summary.txt
