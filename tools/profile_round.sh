#!/bin/bash
# Collects the per-round evidence under gpurun_out/prof_$1: kernel-trace stats of the default bench, then FETCH_SIZE and WRITE_SIZE
# (separate PMC passes, plr:: kernels only). Usage on the GPU box:  bash tools/profile_round.sh r01b
set -u
TAG=${1:-r01}
REPO=$(pwd)
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
python -c "import sys; sys.path.insert(0, '$REPO'); import bench; print(bench.kernel_source_digest())" > $OUT/source_digest.txt
python $REPO/bench.py > $OUT/bench.json 2> $OUT/bench.err
rocprofv3 --kernel-trace --stats -d $OUT/kt -o kt --output-format csv -- python $REPO/bench.py --steps 200 --warmup 20 --no-cpu-baseline > $OUT/bench_under_rocprof.json 2> $OUT/kt.err
python $REPO/tools/summarize_profile.py $(ls $OUT/kt/*kernel_stats.csv | head -1) $OUT/kernel_stats.csv
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $C --kernel-include-regex "plr::" -d $OUT/pmc_$C -o pmc --output-format csv -- python $REPO/bench.py --steps 4 --warmup 2 --profile-frames 0 --no-cpu-baseline > /dev/null 2> $OUT/pmc_$C.err
  python $REPO/tools/pmc_summary.py $(ls $OUT/pmc_$C/*counter_collection.csv | head -1) $OUT/pmc_$C.csv
  rm -rf $OUT/pmc_$C
done
python $REPO/bench.py --steps 30 --warmup 5 --no-cpu-baseline --pass-table 2> $OUT/pass_table.txt > /dev/null
# the other single-GPU frame sizes of BASELINE.json as bench lines (config 1: 1080p, config 5 unpartitioned: 8K)
python $REPO/bench.py --width 1920 --height 1080 --no-cpu-baseline > $OUT/bench_1080p.json 2>> $OUT/bench.err
python $REPO/bench.py --width 7680 --height 4320 --steps 150 --no-cpu-baseline > $OUT/bench_8k.json 2>> $OUT/bench.err
rm -rf $OUT/kt
ls -la $OUT
