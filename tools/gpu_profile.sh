#!/bin/bash
export CF_DEBUG_KNOBS=1   # the library reads its CF_* knobs only under this gate (csrc/cf_knobs.hpp)
# Run on the GPU box (through gpurun): kernel-trace stats + PMC passes of bench.py.
# Usage: tools/gpu_profile.sh <tag> [bench args...]   -> gpurun_out/prof_<tag>/
set -u
TAG=${1:-r01}; shift || true
OUT=$PWD/gpurun_out/prof_$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp CF_BENCH_DIR=/tmp/cf_bench_prof
REPO=$PWD
ARGS="--steps 8 --warmup 3 --no-cpu $*"
cd /tmp
timeout 400 python $REPO/bench.py $ARGS > $OUT/bench_plain.json 2> $OUT/bench_plain.err
timeout 240 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o t -- python $REPO/bench.py $ARGS > $OUT/bench_trace.json 2> $OUT/trace.err
# PMC_SET=short: only the HBM traffic counters and the wave / wait counters
if [ "${PMC_SET:-full}" = short ]; then
  PASSES=("FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY")
else
  PASSES=("FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM_RD")
fi
for pmc in "${PASSES[@]}" ; do
  name=$(echo $pmc | tr ' ' '_')
  timeout 240 rocprofv3 --pmc $pmc --output-format csv -d $OUT/pmc_$name -o p -- python $REPO/bench.py $ARGS > $OUT/pmc_$name.json 2> $OUT/pmc_$name.err
done
cd $REPO
python tools/prof_summary.py $OUT > $OUT/summary.txt 2>&1
# keep the copy-back small: raw per-dispatch tables over 2 MB are dropped after summarising
find $OUT -name "*.csv" -size +2M -delete
cat $OUT/summary.txt
