#!/bin/bash
# GPU run 5 of round 3: pair planes, ps_whole short cuts, packed CLI feed, one-request text windows, resolve table at every row
set -u
O=gpurun_out/r3e; mkdir -p $O
export TMPDIR=/tmp CF_BENCH_DIR=/tmp/cfb
timeout 700 python -m pytest tests/test_gpu_parity.py tests/test_async_abi.py tests/test_gpu_cli.py -x -q -m gpu 2>&1 | tail -8 > $O/pytest_gpu_subset.log
timeout 500 python -m pytest tests/test_gpu_scale.py -x -q -m gpu 2>&1 | tail -8 > $O/pytest_gpu_scale.log
timeout 900 python bench.py --other-configs "2r,4" --steps 20 --warmup 5 > $O/bench_cfg2_2r_4.json 2> $O/bench_cfg2_2r_4.err
CF_PAIR_PLANES=0 timeout 200 python bench.py --other-configs "" --no-cpu --steps 20 --warmup 5 > $O/bench_cfg2_nopair.json 2> $O/bench_cfg2_nopair.err
CF_PAIR_PLANES=0 timeout 300 python bench.py --config 2r --other-configs "" --no-cpu --steps 8 --warmup 2 > $O/bench_cfg2r_nopair.json 2> $O/bench_cfg2r_nopair.err
python - <<'P'
import json
for f in ("bench_cfg2_2r_4.json", "bench_cfg2_nopair.json", "bench_cfg2r_nopair.json"):
    try:
        j=json.loads([l for l in open("gpurun_out/r3e/"+f) if l.startswith("{")][-1])
    except Exception as e:
        print(f, "unreadable", e); continue
    print(f, "value %.3e ms/step %.2f kernels %s general %s" % (j["value"], j["ms_per_step"], {k: round(v,2) for k,v in j["kernels_ms"].items()}, {k: v for k, v in j.get("general_kernel_queries").items() if k != "note"}))
    print("   iso", {k: round(v,2) for k,v in j["device_resident"]["blocking_api_kernels_ms"].items()}, "tables", j["config"].get("index_tables"))
    print("   ops", {k: round(v,2) for k,v in j["ops_per_read"].items()}, "req frac", j["roofline"].get("frac_of_measured_request_rate"))
    for c, o in j.get("other_configs", {}).items():
        print(c, {k: (round(v,3) if isinstance(v,float) else v) for k, v in o.items() if k not in ("workload","ops_per_read","derived_tables", "kernels_ms")})
        print("   ops", {k: round(v,2) for k,v in o.get("ops_per_read", {}).items()})
P
tail -n 3 $O/pytest_gpu_subset.log $O/pytest_gpu_scale.log
