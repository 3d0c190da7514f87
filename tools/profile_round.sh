#!/bin/bash
# Collects the per-round evidence under gpurun_out/prof_$1: kernel-trace stats of the default bench, then FETCH_SIZE and WRITE_SIZE
# (separate PMC passes, plr:: kernels only), the other frame sizes, and the parity / config-5 / band-cost reports of the SAME build. Usage on the GPU box:  bash tools/profile_round.sh r01b
set -u
TAG=${1:-r01}
REPO=$(pwd)
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
python -c "import sys; sys.path.insert(0, '$REPO'); import bench; print(bench.kernel_source_digest())" > $OUT/source_digest.txt
python $REPO/bench.py > $OUT/bench.json 2> $OUT/bench.err
timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/kt -o kt --output-format csv -- python $REPO/bench.py --steps 200 --warmup 20 --no-cpu-baseline > $OUT/bench_under_rocprof.json 2> $OUT/kt.err
python $REPO/tools/summarize_profile.py $(ls $OUT/kt/*kernel_stats.csv | head -1) $OUT/kernel_stats.csv
for C in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --pmc $C --kernel-include-regex "plr::" -d $OUT/pmc_$C -o pmc --output-format csv -- python $REPO/bench.py --steps 4 --warmup 2 --profile-frames 0 --no-cpu-baseline > /dev/null 2> $OUT/pmc_$C.err
  python $REPO/tools/pmc_summary.py $(ls $OUT/pmc_$C/*counter_collection.csv | head -1) $OUT/pmc_$C.csv
  rm -rf $OUT/pmc_$C
done
# SQ / TA / TCP / TCC counters of every plr:: kernel (one group per pass; a rocprofv3 pass that hangs is cut off): bench.py's valu_roofline / l1_roofline
i=0
for C in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU" \
         "TA_TA_BUSY_sum TCP_PENDING_STALL_CYCLES_sum TCP_TOTAL_CACHE_ACCESSES_sum" "GRBM_GUI_ACTIVE TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  timeout 200 rocprofv3 --pmc $C --kernel-include-regex "plr::" -d $OUT/pmc_sq$i -o pmc --output-format csv -- python $REPO/bench.py --steps 3 --warmup 2 --profile-frames 0 --no-cpu-baseline > /dev/null 2> $OUT/pmc_sq$i.err
  python $REPO/tools/pmc_summary.py $(ls $OUT/pmc_sq$i/*counter_collection.csv | head -1) $OUT/pmc_sq$i.csv
  rm -rf $OUT/pmc_sq$i
done
python $REPO/bench.py --steps 30 --warmup 5 --no-cpu-baseline --pass-table 2> $OUT/pass_table.txt > /dev/null
# the other single-GPU frame sizes of BASELINE.json as bench lines (config 1: 1080p, config 5 unpartitioned: 8K)
python $REPO/bench.py --width 1920 --height 1080 --no-cpu-baseline > $OUT/bench_1080p.json 2>> $OUT/bench.err
python $REPO/bench.py --width 7680 --height 4320 --steps 150 --no-cpu-baseline > $OUT/bench_8k.json 2>> $OUT/bench.err
# the input producers recorded as compute passes (not the headline workload): fast set and exact set, per-pass times
python $REPO/bench.py --producers --steps 100 --warmup 10 --no-cpu-baseline --pass-table 2> $OUT/pass_table_producers.txt > $OUT/bench_producers.json
python $REPO/bench.py --producers --exact --steps 30 --warmup 5 --no-cpu-baseline --pass-table 2> $OUT/pass_table_producers_exact.txt > /dev/null
rm -rf $OUT/kt
# ---- parity evidence FROM THIS BUILD: the full-size suite at the benchmarked size, config 5 at 8K, the band cost table; every
# report carries the kernel source digest the PMC summary was stamped with, and the run fails if the sources changed in between
DIGEST=$(cat $OUT/source_digest.txt)
cd $REPO
{ echo "# kernel source digest: $DIGEST"; echo "# python -m pytest tests/test_parity_fullsize.py -m gpu -s   (3840x2160, 256 x 64^3: bench.py's scene)";
  python -m pytest tests/test_parity_fullsize.py -m gpu -q -s 2>&1 | grep -E "^\.?PARITY|passed|failed" | sed 's/^\.//'; } > $OUT/parity_4k.txt
{ echo "# kernel source digest: $DIGEST"; echo "# python -m pytest tests/test_config5_8k.py -m gpu -s   (7680x4320 as 2 x 2 tiles and as 4 row bands on one GPU, exact mode and default halos, 64 frames, vs the unpartitioned frame and the oracle)";
  python -m pytest tests/test_config5_8k.py -m gpu -q -s 2>&1 | grep -E "CONFIG5 .*(summary|partition:|oracle|halos)|passed|failed" | sed 's/^\.*//'; } > $OUT/config5_8k.txt
# tiles against bands in the three GI exchange modes, measured: the single-GPU replay from this build - every rank timed alone with its neighbours' real data in its
# halos (tools/band_cost.py over the native exchange's in-process transport, round 6); then the requested mode without the two-phase filter, and the loopback replay
# of rounds 3 - 5 for continuity
{ echo "# kernel source digest: $DIGEST";
  for M in "--requested" "" "--exact"; do for G in "--tiles 2x2" ""; do echo "# ---- python tools/band_cost.py 4 $G --balance --passes $M"; python tools/band_cost.py 4 $G --balance --passes $M 2>&1 | grep -v amdgpu.ids | grep -vE "^    partition [0-9]"; done; done
  echo "# ---- PLR_BAND_COST_OVERLAP=0 python tools/band_cost.py 4 --tiles 2x2 --balance --requested   (one exchange callback and one filter execution per point)"; PLR_BAND_COST_OVERLAP=0 python tools/band_cost.py 4 --tiles 2x2 --balance --requested 2>&1 | grep -v amdgpu.ids | grep -vE "^    partition [0-9]";
  echo "# ---- python tools/band_cost.py 4 --tiles 2x2 --balance --loopback   (the replay of rounds 3 - 5: every partition with its own texels in its halos)"; python tools/band_cost.py 4 --tiles 2x2 --balance --loopback 2>&1 | grep -v amdgpu.ids | grep -vE "^    partition [0-9]"; } > $OUT/tile_vs_band.txt
# the 256-frame series of the bounded-halo mode's deviation (profiles/r05g_config5_series.txt): unchanged since round 5, re-taken only on request
[ -n "${PLR_PROFILE_SERIES:-}" ] && bash tools/config5_series.sh 256 > $OUT/config5_series.txt 2>&1
{ echo "# kernel source digest: $DIGEST"; python tools/tail_cost.py 2>&1 | grep -v amdgpu.ids; } > $OUT/tail_cost.txt
NOW=$(python -c "import sys; sys.path.insert(0, '$REPO'); import bench; print(bench.kernel_source_digest())")
if [ "$NOW" != "$DIGEST" ]; then echo "kernel sources changed during the profile run ($DIGEST -> $NOW): evidence is inconsistent" >&2; exit 1; fi
grep -q failed $OUT/parity_4k.txt $OUT/config5_8k.txt && { echo "parity suite failed" >&2; exit 1; }
# the dense scene (tens of instances per culling tile, up to the cap): bench line + culling / trace parity at the benchmark's size
python $REPO/bench.py --scene dense --no-cpu-baseline > $OUT/bench_dense.json 2>> $OUT/bench.err
{ echo "# kernel source digest: $DIGEST"; python tools/parity_dense.py 2>&1 | grep -v amdgpu.ids; } > $OUT/parity_dense.txt
# one rank's frame of the partitioned 8K frame (request lists, 2 x 2 tiles) as a kernel timeline: what runs when on which queue (tools/band_timeline.sh)
bash tools/band_timeline.sh > /dev/null 2>&1
{ echo "# kernel source digest: $DIGEST"; echo "# bash tools/band_timeline.sh   (rocprofv3 --kernel-trace around tools/band_cost.py 4 --tiles 2x2 --requested; start, end, duration, queue, kernel)";
  cat $REPO/gpurun_out/band_timeline/timeline.txt; } > $OUT/band_timeline.txt
ls -la $OUT
