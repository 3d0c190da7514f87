#!/bin/bash
for W in 6 4; do
  PLR_EXTRA_FLAGS="-DPLR_SHADE_WAVES=$W" python -c "
from plainrenderer_amd import build; build.build(verbose=False)" > /dev/null 2>&1
  for i in 1 2; do echo "waves $W"; python bench.py --steps 600 --warmup 20 --no-cpu-baseline --pass-table 2>&1 >/dev/null | grep -E "Forward shading|sum of"; done
done
