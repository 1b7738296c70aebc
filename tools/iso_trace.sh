#!/bin/bash
export CF_DEBUG_KNOBS=1   # the library reads its CF_* knobs only under this gate (csrc/cf_knobs.hpp)
# Run on the GPU box (through gpurun): kernel trace of bench.py's blocking-API repetitions (one slot alone, no copies beside the
# kernels — under rocprofv3 the pipeline's device-to-host copies become blit kernels that slow whatever runs beside them).
# Prints every kernel of the last repetition with its duration.  Usage: tools/iso_trace.sh <tag> [bench args...]
set -u
TAG=${1:-iso}; shift || true
R=$PWD; OUT=$R/gpurun_out/iso_$TAG; mkdir -p $OUT
export TMPDIR=/tmp CF_BENCH_DIR=${CF_BENCH_DIR:-/tmp/cfb}
cd /tmp
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $OUT -o t -- python $R/bench.py --no-cpu --steps 3 --warmup 1 "$@" > $OUT/bench.json 2> $OUT/err.txt
cd $R
python - "$OUT" <<'PY'
import csv, glob, sys
out = sys.argv[1]
K = list(csv.DictReader(open(glob.glob(out + "/**/t_kernel_trace.csv", recursive=True)[0])))
rows = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in K)
idx = [i for i, r in enumerate(rows) if "k_plan(" in r[2]]
# the last repetition starts a few kernels before its k_plan (read lengths, first scan): take from the end of the previous k_compact
start = max(i for i, r in enumerate(rows) if "k_compact" in r[2] and i < idx[-1]) + 1
t0 = rows[start][0]
with open(out + "/last_rep.txt", "w") as f:
    for s, e, n in rows[start:]:
        f.write("%9.3f %8.3f  %s\n" % ((s - t0) / 1e6, (e - s) / 1e6, n.replace("(anonymous namespace)::", "").replace("cfamd::", "")[:70]))
PY
find $OUT -name "*.csv" -size +1M -delete
cat $OUT/last_rep.txt
