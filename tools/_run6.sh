set -u
O=gpurun_out/r03f; mkdir -p $O
export TMPDIR=/tmp
python bench.py --no-cpu-baseline --pass-table > $O/bench.json 2> $O/pass_table.txt
for h in 64 128 192 256; do
PLR_CONFIG5_GI_HALO=$h PLR_CONFIG5_REPORT_ONLY=1 timeout 600 python -m pytest tests/test_config5_8k.py -m gpu -x -q -s 2>&1 | grep -E "CONFIG5 frame 2|passed|failed|Error" > $O/config5_halo$h.txt
done
timeout 900 python -m pytest tests/test_fusion.py tests/test_parity_fullsize.py -m gpu -x -q -s -k "fusion or fused or async" > $O/pytest_a.log 2>&1; echo "rc=$?" >> $O/pytest_a.log
tail -n 3 $O/pytest_a.log; head -c 300 $O/bench.json; echo; grep -E "upscale|sum of" $O/pass_table.txt; cat $O/config5_halo*.txt
