#!/bin/bash
# config 5 (8K, four bands) over 16 frames of temporal feedback, report only: does the band-vs-unpartitioned deviation converge?
# second run: a sample on a row no neighbour sent gets weight 0 WITHOUT shrinking the disc for the samples after it
mkdir -p gpurun_out/r04_config5
for V in 1 0; do
  echo "# PLR_BAND_ROW_MISS_SHRINKS=$V"
  PLR_BAND_ROW_MISS_SHRINKS=$V PLR_CONFIG5_FRAMES=16 PLR_CONFIG5_REPORT_ONLY=1 timeout 1200 python -m pytest tests/test_config5_8k.py -m gpu -q -s 2>&1 | grep -E "CONFIG5|passed|failed|Error"
done
