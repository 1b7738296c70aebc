#!/bin/bash
# Run on the GPU box (through gpurun): the other bench presets -> gpurun_out/<tag>/  (each builds its index on the GPU first)
set -u
TAG=${1:-cfg}
OUT=$PWD/gpurun_out/$TAG; mkdir -p $OUT
export CF_BENCH_DIR=/tmp/cfb
timeout 400 python bench.py --config 4 --cpu-sample 100000 > $OUT/bench_cfg4.json 2> $OUT/bench_cfg4.err; tail -c 200 $OUT/bench_cfg4.json; rm -rf /tmp/cfb
timeout 300 python bench.py --config 2r --genomes 512 --cpu-sample 100000 > $OUT/bench_cfg2r.json 2> $OUT/bench_cfg2r.err; tail -c 200 $OUT/bench_cfg2r.json; rm -rf /tmp/cfb
timeout 700 python bench.py --config 5 --cpu-sample 50000 > $OUT/bench_cfg5.json 2> $OUT/bench_cfg5.err; tail -c 200 $OUT/bench_cfg5.json
grep -h "index in HBM\|index built" $OUT/*.err
