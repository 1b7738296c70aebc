#!/bin/bash
# GPU run 4 of round 3: per-query stage by field, multi-entry register score; configs 2r and 4; timeline
set -u
O=gpurun_out/r3d; mkdir -p $O
export TMPDIR=/tmp CF_BENCH_DIR=/tmp/cfb
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_async_abi.py tests/test_gpu_cli.py tests/test_variants.py -x -q -m gpu 2>&1 | tail -8 > $O/pytest_gpu_subset.log
timeout 400 python -m pytest tests/test_gpu_scale.py -x -q -m gpu -k "repeat or pairs_150 or len250_text" 2>&1 | tail -8 > $O/pytest_gpu_scale.log
timeout 900 python bench.py --other-configs "2r,4" --steps 20 --warmup 5 > $O/bench_cfg2_2r_4.json 2> $O/bench_cfg2_2r_4.err
CF_TAIL_STREAM=0 timeout 200 python bench.py --other-configs "" --no-cpu --steps 20 --warmup 5 > $O/bench_cfg2_notail.json 2> $O/bench_cfg2_notail.err
cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/trace -o t -- python $GRAFT_REPO_ROOT/bench.py --other-configs "" --no-cpu --steps 10 --warmup 3 > $GRAFT_REPO_ROOT/$O/bench_trace.json 2> $GRAFT_REPO_ROOT/$O/trace.err
cd $GRAFT_REPO_ROOT
python tools/timeline.py $O/trace k_search2_l1 > $O/timeline.txt 2>&1
find $O -name "*.csv" -size +2M -delete
python - <<'P'
import json
for f in ("bench_cfg2_2r_4.json", "bench_cfg2_notail.json"):
    j=json.loads([l for l in open("gpurun_out/r3d/"+f) if l.startswith("{")][-1])
    print(f, "value %.3e ms/step %.2f kernels %s general %s" % (j["value"], j["ms_per_step"], {k: round(v,2) for k,v in j["kernels_ms"].items()}, {k: v for k, v in j.get("general_kernel_queries").items() if k != "note"}))
    print("   iso", {k: round(v,2) for k,v in j["device_resident"]["blocking_api_kernels_ms"].items()})
    for c, o in j.get("other_configs", {}).items():
        print(c, {k: (round(v,3) if isinstance(v,float) else v) for k, v in o.items() if k not in ("workload","ops_per_read","derived_tables")})
P
cat $O/timeline.txt | head -70
tail -n 3 $O/pytest_gpu_subset.log $O/pytest_gpu_scale.log
