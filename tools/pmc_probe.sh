#!/bin/bash
# usage: bash tools/pmc_probe.sh <kernel-regex> "<counters pass 1>" "<counters pass 2>" ...
set -u
REPO=$(pwd); OUT=$REPO/gpurun_out/pmc_probe; mkdir -p $OUT; export TMPDIR=/tmp; cd /tmp
RE=$1; shift
i=0
for C in "$@"; do
  i=$((i+1))
  timeout 150 rocprofv3 --pmc $C --kernel-include-regex "$RE" -d $OUT/p$i -o pmc --output-format csv -- python $REPO/bench.py --steps 3 --warmup 2 --profile-frames 0 --no-cpu-baseline > /dev/null 2> $OUT/p$i.err
  python $REPO/tools/pmc_summary.py $(ls $OUT/p$i/*counter_collection.csv | head -1) $OUT/p$i.csv && cat $OUT/p$i.csv
  rm -rf $OUT/p$i
done
