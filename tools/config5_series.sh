#!/bin/bash
# Config 5 (8K in 2 x 2 tiles and in four bands) over a long series of temporal feedback, report only:
#  1. the default bounded GI halo over FRAMES frames (default 256): where does the partitioned-vs-unpartitioned deviation go? (it does not settle)
#  2. the EXACT mode (band_gi_halo = PLRF_HALO_WHOLE_IMAGE: every GI texel a denoiser sample can reach is exchanged) over 64 frames: equal, bit for bit
#  3. what the exact mode costs a partition: single-GPU replay with the loopback exchange (the local work) and the bytes a rank receives per frame
#   bash tools/config5_series.sh > profiles/r05_config5_series.txt     (on the GPU box; ~5 minutes)
FRAMES=${1:-256}
T=tests/test_config5_8k.py
FILTER='s/^\.*//'
echo "# kernel source digest: $(python -c 'import bench; print(bench.kernel_source_digest())')"
echo "# ---- 1. default halos (128 trace rows at 8K), $FRAMES frames, camera moving 2 mm / 4 mm per frame over a static G-buffer (kept frames: 0-3, every 16th, the last)"
PLR_CONFIG5_KEEP_EVERY=16 PLR_CONFIG5_FRAMES=$FRAMES PLR_CONFIG5_REPORT_ONLY=1 timeout 2400 python -m pytest $T -m gpu -q -s -k halo 2>&1 | grep -E "CONFIG5|passed|failed|Error" | sed "$FILTER"
echo "# ---- 2. exact mode, 64 frames (kept: 0-3, every 8th, the last): asserted equal (resolved colour, swapchain, histogram, exposure)"
PLR_CONFIG5_FRAMES=64 timeout 2400 python -m pytest $T -m gpu -q -s -k exact 2>&1 | grep -E "CONFIG5 .*(summary|partition:|oracle)|passed|failed|Error" | sed "$FILTER"
echo "# ---- 3. what the exact mode costs a partition (single-GPU replay, loopback exchange: the local work; the bytes are what one rank moves per frame)"
PLR_BAND_COST_GI_HALO=3840 timeout 1200 python tools/band_cost.py 4 --tiles 2x2 2>&1 | grep -E "unpartitioned 7680|^partition|sum of|received"
PLR_BAND_COST_GI_HALO=3840 timeout 1200 python tools/band_cost.py 4 2>&1 | grep -E "unpartitioned 7680|^partition|sum of|received"
