#!/bin/bash
set -u
TAG=${1:-cfg5}
OUT=$PWD/gpurun_out/$TAG; mkdir -p $OUT
export CF_BENCH_DIR=/tmp/cfb
timeout 800 python bench.py --config 5 --cpu-sample 50000 > $OUT/bench_cfg5.json 2> $OUT/bench_cfg5.err; tail -c 200 $OUT/bench_cfg5.json
grep -h "index in HBM\|index built" $OUT/*.err
