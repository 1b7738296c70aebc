#!/bin/bash
# Re-collects the HBM-traffic PMC passes of the config-2 command (-> profiles/pmc_traffic.json) and one headline bench line,
# for a tree whose kernel sources changed in comments only (same device code, new source sha).
set -u
TAG=${1:-r3n}
O=$PWD/gpurun_out/$TAG; mkdir -p $O
export TMPDIR=/tmp
R=$PWD
cd /tmp
for pmc in FETCH_SIZE WRITE_SIZE; do
  timeout 240 rocprofv3 --pmc $pmc --output-format csv -d $O/pmc_$pmc -o p -- python $R/bench.py --steps 8 --warmup 3 --no-cpu --other-configs "" > $O/pmc_$pmc.json 2> $O/pmc_$pmc.err
done
cd $R
python tools/make_pmc_json.py $O > $O/pmc_traffic.json 2> $O/pmc_traffic.err
find $O -name "*.csv" -size +2M -delete
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 --other-configs "" > $O/bench.json 2> $O/bench.err
tail -c 600 $O/bench.json; echo; head -c 600 $O/pmc_traffic.json; tail -n 3 $O/pmc_traffic.err
