set -u
O=gpurun_out/r03e; mkdir -p $O
export TMPDIR=/tmp
python bench.py --no-cpu-baseline --pass-table > $O/bench.json 2> $O/pass_table.txt
PLR_ASYNC_TAIL=0 python bench.py --no-cpu-baseline --pass-table --steps 300 > $O/bench_inorder.json 2> $O/pass_table_inorder.txt
timeout 900 python -m pytest tests/test_fusion.py tests/test_parity_fullsize.py tests/test_bands.py -m gpu -x -q -s -k "fusion or fused or shading or frame or async or bands or band" > $O/pytest_a.log 2>&1; echo "rc=$?" >> $O/pytest_a.log
timeout 900 python -m pytest tests/test_config5_8k.py -m gpu -x -q -s > $O/pytest_config5.log 2>&1; echo "rc=$?" >> $O/pytest_config5.log
timeout 600 python tools/band_cost.py 4 --passes --balance > $O/band_cost.txt 2>&1
bash tools/pmc_probe.sh "sdfDiffuseTraceFast" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_VMEM_RD" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum" "TCC_HIT_sum TCC_MISS_sum" "SQ_WAVES SQ_INSTS_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_SALU" "GRBM_GUI_ACTIVE" > $O/trace_pmc.txt 2>&1
for f in $O/pytest_a.log $O/pytest_config5.log $O/band_cost.txt; do tail -n 4 $f; done; head -c 400 $O/bench.json; echo; head -c 400 $O/bench_inorder.json; cat $O/trace_pmc.txt | tail -12
