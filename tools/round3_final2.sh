#!/bin/bash
# Last validation of round 3 on the GPU box (table planner on by default): the whole GPU suite, the driver's bench command
# (headline + 2r, 4, 5 as other_configs), the two HBM-traffic PMC passes of the config-2 command -> profiles/pmc_traffic.json
set -u
TAG=${1:-r3k}
O=$PWD/gpurun_out/$TAG; mkdir -p $O
export TMPDIR=/tmp
R=$PWD
timeout 1300 python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1; grep -E "passed|failed|error" $O/pytest_gpu.log | tail -3
( time timeout 1700 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err ) 2> $O/bench_wall.txt; tail -c 300 $O/bench.json; echo; cat $O/bench_wall.txt
cd /tmp
for pmc in FETCH_SIZE WRITE_SIZE; do
  timeout 240 rocprofv3 --pmc $pmc --output-format csv -d $O/pmc_$pmc -o p -- python $R/bench.py --steps 8 --warmup 3 --no-cpu --other-configs "" > $O/pmc_$pmc.json 2> $O/pmc_$pmc.err
done
cd $R
python tools/make_pmc_json.py $O > $O/pmc_traffic.json 2> $O/pmc_traffic.err; cp profiles/pmc_traffic.json $O/pmc_traffic_profiles_copy.json
find $O -name "*.csv" -size +2M -delete
python - <<P
import json
j=json.loads([l for l in open("$O/bench.json") if l.startswith("{")][-1])
print("cfg2 value %.3e ms/step %.2f" % (j["value"], j["ms_per_step"]), {k: round(v,2) for k,v in j["device_resident"]["blocking_api_kernels_ms"].items()}, j["cpu_baseline"].get("gpu_rows_identical_on_sample"), j["roofline"]["frac"], j["roofline"]["traffic"])
for c,o in j.get("other_configs",{}).items():
    print(c, {k:(round(v,3) if isinstance(v,float) else v) for k,v in o.items() if k in ("value","ms_per_step","requests_per_read","gpu_rows_identical","parity_checked_reads","wall_s","failed","skipped","index_bytes","kernels_ms_one_slot_alone","derived_tables")})
P
