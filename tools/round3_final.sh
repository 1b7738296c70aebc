#!/bin/bash
# Final validation of round 3 on the GPU box: the whole GPU suite, config 5 on its own, the default bench line with 2r and 4,
# the rocprofv3 kernel trace and the two HBM-traffic PMC passes of the config-2 command, the timeline of the pipelined steps.
set -u
TAG=${1:-r3g}
O=$PWD/gpurun_out/$TAG; mkdir -p $O
export CF_BENCH_DIR=/tmp/cfb TMPDIR=/tmp
R=$PWD
timeout 1300 python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1; grep -E "passed|failed|error" $O/pytest_gpu.log | tail -3
timeout 900 python bench.py --config 5 --other-configs "" --steps 8 --warmup 2 --cpu-sample 100000 > $O/bench_cfg5.json 2> $O/bench_cfg5.err; tail -c 200 $O/bench_cfg5.json; echo
rm -rf /tmp/cfb/cf_bench_24576_*          # (47 GB of index files)
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 --other-configs "2r,4" > $O/bench.json 2> $O/bench.err; tail -c 200 $O/bench.json; echo
cd /tmp
timeout 240 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o t -- python $R/bench.py --steps 10 --warmup 3 --no-cpu --other-configs "" > $O/bench_trace.json 2> $O/trace.err
for pmc in FETCH_SIZE WRITE_SIZE; do
  timeout 240 rocprofv3 --pmc $pmc --output-format csv -d $O/pmc_$pmc -o p -- python $R/bench.py --steps 8 --warmup 3 --no-cpu --other-configs "" > $O/pmc_$pmc.json 2> $O/pmc_$pmc.err
done
cd $R
python tools/timeline.py $O/trace k_search2_l1 > $O/timeline.txt 2>&1
python tools/prof_summary.py $O > $O/summary.txt 2>&1
python tools/make_pmc_json.py $O > $O/pmc_traffic.json 2> $O/pmc_traffic.err; cp profiles/pmc_traffic.json $O/pmc_traffic_profiles_copy.json
find $O -name "*.csv" -size +2M -delete
grep -E "k_search2_l1<4, false>" $O/summary.txt | head -6
head -30 $O/timeline.txt
