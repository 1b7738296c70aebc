#!/bin/bash
# The driver's bench command (headline + 2r, 4, 5) on the tree as it stands.
set -u
TAG=${1:-r3r}
O=$PWD/gpurun_out/$TAG; mkdir -p $O
( time timeout 1000 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err ) 2> $O/bench_wall.txt; cat $O/bench_wall.txt
python - <<P
import json
j=json.loads([l for l in open("$O/bench.json") if l.startswith("{")][-1])
print("cfg2 value %.3e ms/step %.2f" % (j["value"], j["ms_per_step"]), {k: round(v,2) for k,v in j["kernels_ms"].items()}, j["cpu_baseline"].get("gpu_rows_identical_on_sample"), j["roofline"]["frac"], j["roofline"]["traffic"])
for c,o in j.get("other_configs",{}).items():
    print(c, {k:(round(v,3) if isinstance(v,float) else v) for k,v in o.items() if k in ("value","ms_per_step","requests_per_read","gpu_rows_identical","parity_checked_reads","wall_s","failed","skipped")})
P
