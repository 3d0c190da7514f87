#!/bin/bash

python bench.py --no-cpu-baseline --steps 300 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('ms', d['ms_per_step'], 'shade', d['passes_ms']['Indirect lighting upscale + Forward shading (deferred)'])"
bash tools/pmc_probe.sh "upscaleAndShade" "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES" "TA_TA_BUSY_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum" "GRBM_GUI_ACTIVE TCP_TCC_READ_REQ_sum" 2>&1 | grep -v "amdgpu.ids\|^kernel"
PLR_PARITY_SIZE=1920x1088 timeout 1200 python -m pytest tests/test_parity_fullsize.py tests/test_shading.py tests/test_golden_frame.py tests/test_fusion.py -m gpu -x -q -s -k "shad or golden or fusion" 2>&1 | tail -12
