#!/bin/bash
# rocprofv3 --kernel-trace --stats of the headline command at the final tree (+ tools/timeline.py, tools/prof_summary.py)
set -u
TAG=${1:-r3p}
O=$PWD/gpurun_out/$TAG; mkdir -p $O
export TMPDIR=/tmp
R=$PWD
cd /tmp
timeout 240 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o t -- python $R/bench.py --steps 10 --warmup 3 --no-cpu --other-configs "" > $O/bench_trace.json 2> $O/trace.err
cd $R
python tools/timeline.py $O/trace k_search2_l1 > $O/timeline.txt 2>&1
python tools/prof_summary.py $O > $O/summary.txt 2>&1
cp $(find $O/trace -name "*kernel_stats.csv" | head -n 1) $O/kernel_stats.csv
find $O -name "*.csv" -size +2M -delete
head -n 14 $O/summary.txt; head -n 30 $O/timeline.txt
