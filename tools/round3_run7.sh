#!/bin/bash
# GPU run 7 of round 3: the table planner against the fixed priorities (config 2 under budgets, configs 4 and 5), CLI tests
set -u
O=gpurun_out/r3i; mkdir -p $O
export TMPDIR=/tmp CF_BENCH_DIR=/tmp/cfb
timeout 400 python -m pytest tests/test_gpu_cli.py tests/test_async_abi.py -q -m gpu 2>&1 | tail -6 > $O/pytest_gpu_cli_abi.log
for gb in 16 32 64 100; do
  CF_TABLE_PLANNER=1 timeout 200 python bench.py --other-configs "" --no-cpu --steps 10 --warmup 3 --hbm-budget-gb $gb > $O/bench_cfg2_planner_budget_$gb.json 2> $O/bench_cfg2_planner_budget_$gb.err
done
CF_TABLE_PLANNER=1 timeout 200 python bench.py --other-configs "" --no-cpu --steps 10 --warmup 3 > $O/bench_cfg2_planner.json 2> $O/bench_cfg2_planner.err
CF_TABLE_PLANNER=1 timeout 300 python bench.py --config 4 --other-configs "" --no-cpu --steps 8 --warmup 2 > $O/bench_cfg4_planner.json 2> $O/bench_cfg4_planner.err
CF_TABLE_PLANNER=1 timeout 600 python bench.py --config 5 --other-configs "" --no-cpu --steps 8 --warmup 2 > $O/bench_cfg5_planner.json 2> $O/bench_cfg5_planner.err
python - <<'P'
import json, glob
for f in sorted(glob.glob("gpurun_out/r3i/bench_*.json")):
    try:
        j=json.loads([l for l in open(f) if l.startswith("{")][-1])
    except Exception as e:
        print(f, "unreadable", e); continue
    t=j["config"].get("index_tables") or {}
    print(f.split("/")[-1], "value %.3e ms/step %.2f iso %s resident %.1f GB reqs/read %.1f" % (j["value"], j["ms_per_step"], {k: round(v,2) for k,v in j["device_resident"]["blocking_api_kernels_ms"].items()}, t.get("total_bytes",0)/1e9, j["roofline"]["load_requests_per_launch"]/j["config"]["reads_per_gpu_per_step"]),
          {k: j["config"].get(k) for k in ("wide_ftab_chars","text_verify_sample_every_nth","occ_planes","pair_planes","resolve_table_every_nth_row")})
P
tail -n 4 $O/pytest_gpu_cli_abi.log
