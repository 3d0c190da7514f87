#!/bin/bash
# the latency-bound front of frame N+1 (exposure chain, pyramid, culling, trace) beside the VALU-bound TAA of frame N.
# PLR_TAA_ON_TAIL=1 puts the resolve on the asynchronous tail stream in front of the bloom chain: it then starts together with the next frame's front.
export TMPDIR=/tmp
REPO=$(pwd); OUT=$REPO/gpurun_out/r04_overlap; mkdir -p $OUT; cd /tmp
for V in 0 1 0 1 0 1; do
  PLR_TAA_ON_TAIL=$V python $REPO/bench.py --no-cpu-baseline --steps 600 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('PLR_TAA_ON_TAIL=$V  ms_per_step', d['ms_per_step'], ' TAA pass us', round(1e3*d['passes_ms']['Temporal filtering'],1), ' trace pass us', round(1e3*d['passes_ms']['Indirect diffuse SDF trace'],1))"
done
for V in 0 1; do
  PLR_TAA_ON_TAIL=$V timeout 300 rocprofv3 --kernel-trace -d $OUT/kt$V -o kt --output-format csv -- python $REPO/bench.py --steps 100 --warmup 10 --profile-frames 0 --no-cpu-baseline > /dev/null 2> $OUT/kt$V.err
  python $REPO/tools/overlap_probe.py $(ls $OUT/kt$V/*kernel_trace.csv | head -1) "PLR_TAA_ON_TAIL=$V"
  rm -rf $OUT/kt$V
done
