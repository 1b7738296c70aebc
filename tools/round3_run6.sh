#!/bin/bash
# GPU run 6 of round 3: self-made strand records; HBM budget curve of config 2; the front end with packed reads
set -u
O=gpurun_out/r3f; mkdir -p $O
export TMPDIR=/tmp CF_BENCH_DIR=/tmp/cfb
timeout 700 python -m pytest tests/test_async_abi.py tests/test_gpu_cli.py tests/test_gpu_parity.py -x -q -m gpu 2>&1 | tail -8 > $O/pytest_gpu_subset.log
timeout 500 python -m pytest tests/test_gpu_scale.py -x -q -m gpu 2>&1 | tail -8 > $O/pytest_gpu_scale.log
timeout 300 python bench.py --other-configs "" --steps 20 --warmup 5 > $O/bench_cfg2.json 2> $O/bench_cfg2.err
CF_SELF_RECORDS=0 timeout 200 python bench.py --other-configs "" --no-cpu --steps 20 --warmup 5 > $O/bench_cfg2_kpack.json 2> $O/bench_cfg2_kpack.err
CF_TEXT_VERIFY_MIN_RUN=0 timeout 200 python bench.py --other-configs "" --no-cpu --steps 20 --warmup 5 > $O/bench_cfg2_minrun0.json 2> $O/bench_cfg2_minrun0.err
for gb in 8 16 32 64 100; do
  timeout 200 python bench.py --other-configs "" --no-cpu --steps 10 --warmup 3 --hbm-budget-gb $gb > $O/bench_cfg2_budget_$gb.json 2> $O/bench_cfg2_budget_$gb.err
done
# the front end: 10 M reads end to end, 80 M steady state, packed feed against the byte feed
timeout 300 python tools/cli_e2e.py 256 1000000 10000000 noref > $O/cli_e2e.txt 2>&1
B=$GRAFT_REPO_ROOT/centrifuge_amd/bin/centrifuge-class
if [ -f /tmp/cf_e2e/reads.fa ]; then
  cd /tmp/cf_e2e
  $B -f -p 16 -x idx -U reads.fa -S packed.tsv --report-file packed.rep > /dev/null 2>&1
  CF_CLI_PACKED=0 $B -f -p 16 -x idx -U reads.fa -S bytes.tsv --report-file bytes.rep > /dev/null 2>&1
  cmp packed.tsv bytes.tsv && cmp packed.rep bytes.rep && cmp packed.tsv ours.tsv && echo "TSV and report identical: packed feed, byte feed, -p 8 run" >> $GRAFT_REPO_ROOT/$O/cli_e2e.txt
  for i in 1 2 3 4 5 6 7 8; do cat reads.fa; done > reads80.fa
  for mode in 1 0; do echo "== CF_CLI_PACKED=$mode -p 16" >> $GRAFT_REPO_ROOT/$O/cli_steady.txt; ( time CF_CLI_PACKED=$mode $B -f -x idx -U reads80.fa -S /dev/null --report-file /tmp/cf_e2e/rep.tsv -t -p 16 ) 2>&1 | grep -E "Stage seconds|real|Overall" >> $GRAFT_REPO_ROOT/$O/cli_steady.txt; done
  cd $GRAFT_REPO_ROOT
fi
python - <<'P'
import json, glob
for f in sorted(glob.glob("gpurun_out/r3f/bench_*.json")):
    try:
        j=json.loads([l for l in open(f) if l.startswith("{")][-1])
    except Exception as e:
        print(f, "unreadable", e); continue
    t=j["config"].get("index_tables") or {}
    print(f.split("/")[-1], "value %.3e ms/step %.2f iso %s resident %.1f GB est req %.1f reqs/read %.1f" % (j["value"], j["ms_per_step"], {k: round(v,2) for k,v in j["device_resident"]["blocking_api_kernels_ms"].items()}, t.get("total_bytes",0)/1e9, t.get("est_requests_per_100bp_read",0), j["roofline"]["load_requests_per_launch"]/j["config"]["reads_per_gpu_per_step"]),
          {k: j["config"].get(k) for k in ("wide_ftab_chars","text_verify_sample_every_nth","occ_planes","pair_planes","resolve_table_every_nth_row")})
P
cat $O/cli_e2e.txt | tail -5; cat $O/cli_steady.txt
tail -n 3 $O/pytest_gpu_subset.log $O/pytest_gpu_scale.log
