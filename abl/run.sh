#!/bin/bash
# ablation probe: counters of the shade kernel for each variant library
mkdir -p gpurun_out/r04c
for V in base abl1 abl2 abl4 abl8 abl15; do
  cp abl/libplr_$V.so plainrenderer_amd/libplr.so
  echo "=== $V"
  python bench.py --no-cpu-baseline --steps 200 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('ms', d['ms_per_step'], 'shade', d['passes_ms']['Indirect lighting upscale + Forward shading (deferred)'])"
  bash tools/pmc_probe.sh "upscaleAndShade" "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES" "TA_TA_BUSY_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum" "GRBM_GUI_ACTIVE TCP_TCC_READ_REQ_sum" 2>&1 | grep -v "amdgpu.ids\|^kernel"
done
cp abl/libplr_base.so plainrenderer_amd/libplr.so
