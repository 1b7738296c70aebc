#!/bin/bash
export CF_DEBUG_KNOBS=1   # the library reads its CF_* knobs only under this gate (csrc/cf_knobs.hpp)
# A/B sweep of kernel launch knobs on one box (index cached in /tmp across runs); every run is bounded.
# usage: gpu_sweep.sh <tag> <bench args in quotes> "ENV=.. ENV=.." ...
export CF_BENCH_DIR=/tmp/cf_bench_sweep
OUT=gpurun_out/sweep_$1.txt; shift
ARGS=$1; shift
: > $OUT
run() { echo "== $*" >> $OUT; env "$@" timeout 150 python bench.py --steps 3 --warmup 1 --no-cpu $ARGS 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('reads/s %.3e  kernels %s  search GB/s %.0f' % (d['value'], {k: round(v,2) for k,v in d['kernels_ms'].items()}, d['roofline']['achieved']))" >> $OUT 2>&1; }
for cfg in "$@"; do run $cfg; done
cat $OUT
