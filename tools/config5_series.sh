#!/bin/bash
# Config 5 (8K in 2 x 2 tiles and in four bands) over 64 frames of temporal feedback, report only: where does the partitioned-vs-unpartitioned deviation
# settle?. Then the price of an EXACT mode: the GI halo as large as the image - every denoiser sample finds its texel, the partitioned
# frame equals the unpartitioned one (three frames shown) - as the replay cost of a partition with that halo and the bytes it receives per frame.
#   bash tools/config5_series.sh > profiles/r05_config5_series.txt     (on the GPU box; ~6 minutes)
FRAMES=${1:-64}
echo "# kernel source digest: $(python -c 'import bench; print(bench.kernel_source_digest())')"
echo "# ---- default halos, $FRAMES frames (kept frames: 0-3, every 8th, the last)"
PLR_CONFIG5_FRAMES=$FRAMES PLR_CONFIG5_REPORT_ONLY=1 timeout 2400 python -m pytest tests/test_config5_8k.py -m gpu -q -s 2>&1 | grep -E "CONFIG5|passed|failed|Error" | sed 's/^\.//'
echo "# ---- exact mode: PLR_CONFIG5_GI_HALO = the trace image's larger side (every texel a sample can reach is exchanged), 3 frames"
PLR_CONFIG5_GI_HALO=3840 PLR_CONFIG5_FRAMES=3 PLR_CONFIG5_REPORT_ONLY=1 timeout 2400 python -m pytest tests/test_config5_8k.py -m gpu -q -s 2>&1 | grep -E "CONFIG5|passed|failed|Error" | sed 's/^\.//'
echo "# ---- what the exact mode costs a partition (single-GPU replay, loopback exchange: the local work; the bytes are what one rank receives per frame)"
PLR_BAND_COST_GI_HALO=3840 timeout 1200 python tools/band_cost.py 4 --tiles 2x2 2>&1 | grep -E "unpartitioned 7680|^partition|sum of|received"
PLR_BAND_COST_GI_HALO=3840 timeout 1200 python tools/band_cost.py 4 2>&1 | grep -E "unpartitioned 7680|^partition|sum of|received"
