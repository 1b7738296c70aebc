#!/bin/bash
# One parameterised GPU-session script (replaces the per-run tools/round3_*.sh logs).  Usage, through gpurun:
#   tools/gpu/run.sh <tag> <step> [<step> ...]
# steps: tests | tests:<pytest -k expr> | cli[:<reads>] | curve:<preset>:<gb,...> | bench | bench2 (config 2 alone, no CPU leg) | benchenv:<VAR=val,...> (bench2 under env)
#        | cfg:<2|2r|2r-|4|5|4r> | trace | tracecfg:<preset> | pmc[:<preset>] | calib | smoke | env:<VAR=val,...> (exported for the steps that follow)
# Everything lands in gpurun_out/<tag>/.
set -u
TAG=$1; shift
O=$PWD/gpurun_out/$TAG; mkdir -p $O
export CF_BENCH_DIR=/tmp/cfb TMPDIR=/tmp CF_BENCH_KEEP=1 CF_DEBUG_KNOBS=1      # (the library reads its CF_* knobs only under the gate: cf_knobs.hpp)
R=$PWD
line() { python - "$1" <<'P'
import json, sys
try:
    j = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
except Exception as e:
    print("no bench line in", sys.argv[1], e); sys.exit(0)
r = j.get("roofline", {})
print("%s value %.3e ms/step %.2f" % (j["config"].get("preset", "?"), j["value"], j["ms_per_step"]),
      {k: round(v, 2) for k, v in j.get("kernels_ms", {}).items()},
      "req/read", j.get("requests_per_read", r.get("requests_per_read")), "frac", r.get("frac"), "traffic", r.get("traffic"),
      "parity", j.get("cpu_baseline", {}).get("gpu_rows_identical_on_sample"))
for c, o in j.get("other_configs", {}).items():
    print(" ", c, {k: (round(v, 3) if isinstance(v, float) else v) for k, v in o.items()
                   if k in ("value", "ms_per_step", "requests_per_read", "gpu_rows_identical", "parity_checked_reads", "wall_s", "failed", "skipped", "kernels_ms")})
P
}
for step in "$@"; do
  arg=${step#*:}; name=${step%%:*}
  case $name in
    tests)
      if [ "$arg" != "$step" ]; then K=(-k "$arg"); else K=(); fi
      timeout 1500 python -m pytest tests -m gpu -q "${K[@]}" --durations=15 --timeout=${CF_TEST_TIMEOUT:-900} > $O/pytest_gpu.log 2>&1
      grep -E "passed|failed|error" $O/pytest_gpu.log | tail -3 ;;
    env)
      IFS=, read -ra KV <<< "$arg"; for kv in "${KV[@]}"; do export "$kv"; done; echo "env: $arg" ;;
    unenv)
      IFS=, read -ra KV <<< "$arg"; for kv in "${KV[@]}"; do unset "$kv"; done; echo "unset: $arg" ;;
    calib)         # FETCH_SIZE / WRITE_SIZE on a known byte count in this path's access pattern (random 16-byte requests)
      cd /tmp
      for pmc in FETCH_SIZE WRITE_SIZE; do
        timeout 200 rocprofv3 --pmc $pmc --output-format csv -d $O/calib_$pmc -o p -- $R/tools/microbench/bin/fetch_calib 16 > $O/calib_$pmc.txt 2> $O/calib_$pmc.err
      done
      cd $R
      python tools/pmc_calib.py $O > $O/fetch_size_calibration.txt 2>&1; cat $O/fetch_size_calibration.txt ;;
    smoke)
      timeout 300 python __graft_entry__.py smoke > $O/smoke.log 2>&1; tail -2 $O/smoke.log ;;
    bench)
      ( time timeout 1000 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err ) 2> $O/bench_wall.txt
      grep real $O/bench_wall.txt; line $O/bench.json ;;
    bench2)
      timeout 400 python bench.py --steps 20 --warmup 5 --no-cpu --other-configs "" > $O/bench2.json 2> $O/bench2.err; line $O/bench2.json ;;
    cli)           # the end-to-end leg alone (centrifuge-class on a FASTA file of config 2's reads): cli[:<reads>]
      n=50000000; [ "$arg" != "$name" ] && n=$arg
      timeout 900 python bench.py --steps 3 --warmup 1 --no-cpu --other-configs , --cli-reads $n > $O/bench_cli.json 2> $O/bench_cli.err
      python - $O/bench_cli.json <<'P'
import json, sys
j = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
for k, v in j.get("cli_end_to_end", {}).items():
    print(" ", k, v)
P
      ;;
    clitrace)      # the end-to-end leg with a kernel trace of centrifuge-class itself (every table): clitrace[:<reads>]
      n=20000000; [ "$arg" != "$name" ] && n=$arg
      mkdir -p $O/trace
      CF_BENCH_CLI_TRACE=$O/trace timeout 900 python bench.py --steps 3 --warmup 1 --no-cpu --other-configs , --cli-reads $n > $O/bench_clitrace.json 2> $O/bench_clitrace.err
      python tools/prof_summary.py $O > $O/summary_cli.txt 2>&1
      head -40 $O/summary_cli.txt ;;
    benchargs)     # bench2 with extra arguments (commas for blanks): benchargs:--wire,wide
      f=$O/bench_$(echo "$arg" | tr -c 'A-Za-z0-9_=\n' '_')
      timeout 400 python bench.py --steps 20 --warmup 5 --no-cpu --other-configs "" $(echo "$arg" | tr ',' ' ') > $f.json 2> $f.err; echo "args $arg:"; line $f.json ;;
    benchenv)
      f=$O/bench_$(echo "$arg" | tr -c 'A-Za-z0-9_=\n' '_')
      ( IFS=,; for kv in $arg; do export "$kv"; done
        timeout 400 python bench.py --steps 20 --warmup 5 --no-cpu --other-configs "" > $f.json 2> $f.err ); echo "env $arg:"; line $f.json ;;
    cfg)
      timeout 900 python bench.py --config $arg --other-configs "" --steps 20 --warmup 6 --cpu-sample 200000 > $O/bench_cfg$arg.json 2> $O/bench_cfg$arg.err
      line $O/bench_cfg$arg.json ;;
    cfgq)          # a preset alone, no CPU leg (quick A/B runs)
      timeout 600 python bench.py --config $arg --other-configs "" --steps 20 --warmup 6 --no-cpu > $O/bench_cfgq$arg.json 2> $O/bench_cfgq$arg.err
      line $O/bench_cfgq$arg.json ;;
    tracecfg)      # rocprofv3 kernel trace of another preset: tracecfg:<preset>
      cd /tmp
      timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_$arg -o t -- python $R/bench.py --config $arg --steps 10 --warmup 6 --no-cpu --other-configs "" > $O/bench_trace_$arg.json 2> $O/trace_$arg.err
      cd $R
      python tools/timeline.py $O/trace_$arg k_search2_l1 > $O/timeline_$arg.txt 2>&1
      head -24 $O/timeline_$arg.txt ;;
    trace)
      cd /tmp
      timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o t -- python $R/bench.py --steps 10 --warmup 3 --no-cpu --other-configs "" > $O/bench_trace.json 2> $O/trace.err
      cd $R
      python tools/timeline.py $O/trace k_search2_l1 > $O/timeline.txt 2>&1
      python tools/timeline.py $O/trace k_search2_l1 18 > $O/timeline_resident.txt 2>&1      # (3 + 10 + 3 launches come before the resident-input steps)
      python tools/prof_summary.py $O > $O/summary.txt 2>&1
      grep -E "k_search2_l1" $O/summary.txt | head -4 ;;
    pmc)           # FETCH_SIZE / WRITE_SIZE of a preset's kernels: pmc[:<preset>]  (separate passes; nothing but --pmc)
      [ "$arg" = "$step" ] && arg=2
      cd /tmp
      for pmc in FETCH_SIZE WRITE_SIZE; do
        timeout 400 rocprofv3 --pmc $pmc --output-format csv -d $O/pmc_${pmc}_$arg -o p -- python $R/bench.py --config $arg --steps 6 --warmup 3 --no-cpu --other-configs "" > $O/pmc_${pmc}_$arg.json 2> $O/pmc_${pmc}_$arg.err
      done
      cd $R
      python tools/make_pmc_json.py $O $arg > $O/pmc_traffic_$arg.json 2> $O/pmc_traffic_$arg.err; head -c 400 $O/pmc_traffic_$arg.json; echo
      python tools/pmc_kernels.py $O $arg > $O/pmc_kernels_$arg.txt 2>&1; head -12 $O/pmc_kernels_$arg.txt ;;
    pmcq)          # the same with two timed steps and one warm-up, counters on this library's kernels only (--kernel-include-regex): pmcq:<preset>.
                   # rocprofv3 died on the 103 Gbp preset in rounds 5 and 6 inside one of the ~15,000 torch launches that generate its genomes
                   # (profiles/r06j_pmc_cfg5_rocprofv3_crash.txt: at::native::bitwise_and under the counter service); those launches are not what is measured
      cd /tmp
      for pmc in FETCH_SIZE WRITE_SIZE; do
        timeout 600 rocprofv3 --pmc $pmc --kernel-include-regex "k_search|k_post|k_score|k_walk|k_plan|k_count|k_compact|k_scan|k_resolve|k_window|k_dense|k_rev" --output-format csv -d $O/pmc_${pmc}_$arg -o p -- python $R/bench.py --config $arg --steps 2 --warmup 1 --no-cpu --other-configs "" > $O/pmc_${pmc}_$arg.json 2> $O/pmc_${pmc}_$arg.err
        tail -2 $O/pmc_${pmc}_$arg.err | cut -c1-200
      done
      cd $R
      python tools/make_pmc_json.py $O $arg > $O/pmc_traffic_$arg.json 2> $O/pmc_traffic_$arg.err; head -c 400 $O/pmc_traffic_$arg.json; echo
      python tools/pmc_kernels.py $O $arg > $O/pmc_kernels_$arg.txt 2>&1; head -12 $O/pmc_kernels_$arg.txt ;;
    mix)           # instruction mix and wave states of a preset's kernels (SQ counters, two passes): mix[:<preset>]
      [ "$arg" = "$step" ] && arg=2
      bash tools/gpu_pmc_quick.sh ${TAG}_mix_$arg --config $arg --other-configs , --cli-reads 0 | tail -12 ;;     # ("," = no other preset: an empty argument does not survive the script's word splitting)
    curve)         # budget -> throughput curve of a preset: curve:<preset>:<gb,gb,...>
      preset=${arg%%:*}; gbs=${arg#*:}
      ( IFS=,; for gb in $gbs; do
          timeout 400 python bench.py --config $preset --hbm-budget-gb $gb --steps 10 --warmup 6 --no-cpu --other-configs "" > $O/curve_${preset}_$gb.json 2> $O/curve_${preset}_$gb.err
          python - $O/curve_${preset}_$gb.json $gb <<'P'
import json, sys
try:
    j = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
    c = j["config"]
    print("budget %s GB: %.3e reads/s (host to host %.3e), resident %.1f GB, K %s text 1/%s planes %s pair %s resolve 1/%s" % (sys.argv[2], j["value"], j["host_to_host"]["reads_per_s"],
          c["index_bytes"] / 1e9, c["wide_ftab_chars"], c["text_verify_sample_every_nth"], c["occ_planes"], c["pair_planes"], c["resolve_table_every_nth_row"]))
except Exception as e:
    print("budget", sys.argv[2], "failed", e)
P
        done ) ;;
    *) echo "unknown step $step" ;;
  esac
done
find $O -name "*.csv" -size +2M -delete
