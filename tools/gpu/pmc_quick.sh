set -u
export CF_DEBUG_KNOBS=1   # the library reads its CF_* knobs only under this gate (csrc/cf_knobs.hpp)
O=$PWD/gpurun_out/r4o; mkdir -p $O
export CF_BENCH_DIR=/tmp/cfb TMPDIR=/tmp
R=$PWD
cd /tmp
for pmc in FETCH_SIZE WRITE_SIZE; do
  timeout 100 rocprofv3 --pmc $pmc --output-format csv -d $O/pmc_$pmc -o p -- python $R/bench.py --steps 3 --warmup 3 --no-cpu --other-configs "" > $O/pmc_$pmc.json 2> $O/pmc_$pmc.err
done
cd $R
python tools/make_pmc_json.py $O > $O/pmc_traffic.json 2> $O/pmc_traffic.err; head -c 300 $O/pmc_traffic.json
find $O -name "*.csv" -size +2M -delete
