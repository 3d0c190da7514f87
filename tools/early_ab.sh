#!/bin/bash
# A/B of the early parts (plr_set_early_parts, include/plr.h) on ONE box: alternating bench.py runs with PLR_EARLY_PARTS=0 / 1, three rounds.
#   bash tools/early_ab.sh [steps]   -> one line per run: frame ms and the per-pass hipEvent times of the passes involved
STEPS=${1:-300}
for ROUND in 1 2 3; do
  for E in 0 1; do
    PLR_EARLY_PARTS=$E python bench.py --no-cpu-baseline --steps $STEPS 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read())
p = d['passes_ms']
pick = lambda s: 1e3 * sum(v for k, v in p.items() if s in k)
print('PLR_EARLY_PARTS=$E round $ROUND: frame %.4f ms | shade %.1f us (early part %.1f) | trace %.1f | spatial %.1f | temporal GI %.1f | front %.1f | TAA %.1f' % (
    d['ms_per_step'], pick('Forward shading') - pick('early stream'), pick('early stream'), pick('SDF trace'), pick('spatial filter'), pick('diffuse temporal'), pick('Histogram per tile'), pick('Temporal filtering')))"
  done
done
