set -u
mkdir -p gpurun_out/r03a
export TMPDIR=/tmp
( rocprofv3 -L 2>/dev/null | grep -oE "\b(TA|TCP|TD|TCC|SQ|GRBM|SPI)_[A-Z0-9_a-z\[\]]+" | sort -u ) > gpurun_out/r03a/counters.txt 2>&1
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r03a/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r03a/pytest.log
python bench.py --no-cpu-baseline --pass-table > gpurun_out/r03a/bench.json 2> gpurun_out/r03a/pass_table.txt
bash tools/pmc_probe.sh "sdfDiffuseTraceFast" "TA_BUSY_avr TA_TA_BUSY_sum" "TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum" "TA_BUFFER_WAVEFRONTS_sum TA_FLAT_WAVEFRONTS_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TA_TCP_STATE_READ_sum" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_RD SQ_INSTS_VALU" "GRBM_GUI_ACTIVE" > gpurun_out/r03a/trace_pmc.txt 2>&1
cp gpurun_out/pmc_probe/*.err gpurun_out/r03a/ 2>/dev/null
tail -3 gpurun_out/r03a/pytest.log; cat gpurun_out/r03a/bench.json
