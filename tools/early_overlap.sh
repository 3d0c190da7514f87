#!/bin/bash
# the deferred shade's direct lighting (early part, include/plr.h plr_set_early_parts) beside the GI chain - where it starts, what it runs beside,
# what each kernel loses. One box: frame times for every start position (alternating, two rounds), then a rocprofv3 kernel trace per position read by tools/overlap_probe.py.
export TMPDIR=/tmp
REPO=$(pwd); OUT=$REPO/gpurun_out/${1:-r06_overlap}; mkdir -p $OUT; cd /tmp
run() { # $1 = PLR_EARLY_PARTS, $2 = PLR_EARLY_AT, $3 = label
  PLR_EARLY_PARTS=$1 PLR_EARLY_AT="$2" python $REPO/bench.py --no-cpu-baseline --steps 400 --profile-frames 0 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('%-58s frame %.4f ms' % ('$3', d['ms_per_step']))"
}
for ROUND in 1 2; do
  run 0 "" "one launch (early parts off)"
  run 2 "self" "two launches back to back (direct, then upscale + combine)"
  run 1 "" "direct from the start of the frame (beside front .. spatial 2)"
  run 1 "SDF trace" "direct beside trace .. spatial 2"
  run 1 "spatial filter" "direct beside spatial 1 .. spatial 2"
  run 1 "temporal filter" "direct beside temporal GI .. spatial 2"
done
trace() {
  PLR_EARLY_PARTS=$1 PLR_EARLY_AT="$2" timeout 300 rocprofv3 --kernel-trace -d $OUT/kt -o kt --output-format csv -- python $REPO/bench.py --steps 100 --warmup 10 --profile-frames 0 --no-cpu-baseline > /dev/null 2> $OUT/kt.err
  python $REPO/tools/overlap_probe.py $(ls $OUT/kt/*kernel_trace.csv | head -1) "$3"
  rm -rf $OUT/kt
}
trace 0 "" "one launch (early parts off)"
trace 2 "self" "two launches back to back"
trace 1 "" "direct from the start of the frame"
trace 1 "SDF trace" "direct beside trace .. spatial 2"
trace 1 "spatial filter" "direct beside spatial 1 .. spatial 2"
