#!/bin/bash
export CF_DEBUG_KNOBS=1   # the library reads its CF_* knobs only under this gate (csrc/cf_knobs.hpp)
# Run on the GPU box (through gpurun): centrifuge-inspect's FASTA mode (the GPU inverse BWT) on the
# benchmark-scale index: correctness + timing (tools/inspect_scale.py), then rocprofv3 kernel-trace
# stats and the HBM traffic counters of the same command.  -> gpurun_out/prof_<tag>/
set -u
TAG=${1:-r01_inspect}
OUT=$PWD/gpurun_out/prof_$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
REPO=$PWD
IDX=/tmp/cf_inspect_idx
CF_RESTORE_VERBOSE=1 timeout 500 python tools/inspect_scale.py --fasta --keep $IDX > $OUT/inspect_scale.json 2> $OUT/inspect_scale.err
cat $OUT/inspect_scale.json
grep cf_index_restore $OUT/inspect_scale.err
cd /tmp
BIN=$REPO/centrifuge_amd/bin/centrifuge-inspect-bin
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o t -- $BIN --wrapper basic-0 $IDX/idx > /dev/null 2> $OUT/trace.err
for pmc in "FETCH_SIZE" "WRITE_SIZE"; do
  timeout 200 rocprofv3 --pmc $pmc --output-format csv -d $OUT/pmc_$pmc -o p -- $BIN --wrapper basic-0 $IDX/idx > /dev/null 2> $OUT/pmc_$pmc.err
done
cd $REPO
python tools/prof_summary.py $OUT > $OUT/summary.txt 2>&1
find $OUT -name "*.csv" -size +2M -delete
cat $OUT/summary.txt
