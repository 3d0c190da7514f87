#!/bin/bash
# A/B of library builds on ONE box: alternating runs of bench.py with PLR_LIB pointing at each build (plainrenderer_amd/build.py PLR_BUILD_TAG), three rounds.
#   bash tools/ab_libs.sh "<pass name to print>" libplr.so libplr_taa2.so ...
PASS="$1"; shift
for ROUND in 1 2 3; do
  for L in "$@"; do
    PLR_LIB=$(pwd)/plainrenderer_amd/$L python bench.py --no-cpu-baseline --steps 300 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('$L round $ROUND: frame %.4f ms, %s %.1f us' % (d['ms_per_step'], '$PASS', 1e3 * sum(v for k, v in d['passes_ms'].items() if '$PASS' in k)))"
  done
done
