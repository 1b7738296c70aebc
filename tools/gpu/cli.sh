#!/bin/bash
# Front-end rates on the GPU box (through gpurun): ingest alone, 10 M reads end to end, 80 M reads steady state -> gpurun_out/<tag>/
set -u
TAG=${1:-cli}
OUT=$PWD/gpurun_out/$TAG; mkdir -p $OUT
TMPDIR=/tmp timeout 300 python tools/ingest_rate.py 32e6 8 16 > $OUT/ingest_rate.txt 2>&1; cat $OUT/ingest_rate.txt
timeout 400 python tools/cli_e2e.py 256 1000000 10000000 noref > $OUT/cli_e2e.txt 2>&1; tail -6 $OUT/cli_e2e.txt
cd /tmp/cf_e2e && for i in 1 2 3 4 5 6 7 8; do cat reads.fa; done > reads80.fa
B=$GRAFT_REPO_ROOT/centrifuge_amd/bin/centrifuge-class
for a in "-p 16" "-p 16 --slots 3"; do echo "== $a" >> $OUT/cli_steady.txt; ( time $B -f -x idx -U reads80.fa -S /dev/null --report-file /tmp/cf_e2e/rep.tsv -t $a ) 2>&1 | grep -E "Stage seconds|real|Overall" >> $OUT/cli_steady.txt; done
cat $OUT/cli_steady.txt
