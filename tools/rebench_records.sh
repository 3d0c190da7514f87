#!/bin/bash
# The bench records of a round's evidence set, taken AFTER profiles/<tag>_pmc_hbm.csv / _sq_counters.csv of the same sources are in the tree (bench.py stamps
# roofline.traffic_source with the counter file it read): bash tools/rebench_records.sh r04e   -> gpurun_out/rebench/<tag>_bench*.json (copy to profiles/)
TAG=${1:-r00}
cd ${GRAFT_REPO_ROOT:-.}; export TMPDIR=/tmp
R=$(pwd); O=$R/gpurun_out/rebench; mkdir -p $O
python bench.py > $O/${TAG}_bench_with_cpu_baseline.json 2> $O/err.txt
python bench.py --no-cpu-baseline > $O/${TAG}_bench.json 2>> $O/err.txt
python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/${TAG}_bench_driver_flags.json 2>> $O/err.txt
python bench.py --width 1920 --height 1080 --no-cpu-baseline > $O/${TAG}_bench_1080p.json 2>> $O/err.txt
python bench.py --width 7680 --height 4320 --steps 150 --no-cpu-baseline > $O/${TAG}_bench_8k.json 2>> $O/err.txt
python bench.py --producers --steps 100 --warmup 10 --no-cpu-baseline > $O/${TAG}_bench_producers.json 2>> $O/err.txt
cd /tmp; timeout -k 5 600 rocprofv3 --kernel-trace --stats -d /tmp/kt_rebench -o kt --output-format csv -- python $R/bench.py --steps 200 --warmup 20 --no-cpu-baseline > $O/${TAG}_bench_under_rocprof.json 2>> $O/err.txt
cd $R; grep -o '"ms_per_step": [0-9.]*\|"traffic_source": "[^"]*"' $O/${TAG}_*.json
