#!/bin/bash
# Run on the GPU box (through gpurun): the round's last validation when GPU minutes are short — the full GPU test suite,
# bench.py with the CPU leg, one rocprofv3 kernel trace and the two HBM-traffic PMC passes of the same command.
set -u
TAG=${1:-last}
OUT=$PWD/gpurun_out/$TAG; mkdir -p $OUT
export CF_BENCH_DIR=/tmp/cfb TMPDIR=/tmp
R=$PWD
timeout 900 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1; grep -E "passed|failed|error" $OUT/pytest_gpu.log | tail -2
timeout 300 python bench.py > $OUT/bench.json 2> $OUT/bench.err; tail -c 250 $OUT/bench.json; echo
cd /tmp
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o t -- python $R/bench.py --steps 8 --warmup 3 --no-cpu > $OUT/bench_trace.json 2> $OUT/trace.err
for pmc in FETCH_SIZE WRITE_SIZE; do
  timeout 200 rocprofv3 --pmc $pmc --output-format csv -d $OUT/pmc_$pmc -o p -- python $R/bench.py --steps 8 --warmup 3 --no-cpu > $OUT/pmc_$pmc.json 2> $OUT/pmc_$pmc.err
done
cd $R
python tools/prof_summary.py $OUT > $OUT/summary.txt 2>&1
find $OUT -name "*.csv" -size +2M -delete
grep -E "k_search2_l1<4, false>" $OUT/summary.txt | head -4
