set -u
O=gpurun_out/r03c; mkdir -p $O
export TMPDIR=/tmp
python bench.py --no-cpu-baseline --pass-table > $O/bench.json 2> $O/pass_table.txt
PLR_ASYNC_TAIL=0 python bench.py --no-cpu-baseline --pass-table --steps 300 > $O/bench_inorder.json 2> $O/pass_table_inorder.txt
timeout 1200 python -m pytest tests/test_fusion.py tests/test_fast_kernels.py tests/test_hiz_bloom_taa.py -m gpu -x -q > $O/pytest_a.log 2>&1; echo "rc=$?" >> $O/pytest_a.log
timeout 900 python -m pytest tests/test_parity_fullsize.py -m gpu -q -s -k "taa or frame or hiz" > $O/pytest_parity.log 2>&1; echo "rc=$?" >> $O/pytest_parity.log
timeout 900 python -m pytest tests/test_config5_8k.py -m gpu -x -q -s > $O/pytest_config5.log 2>&1; echo "rc=$?" >> $O/pytest_config5.log
timeout 600 python -m pytest tests/test_bands.py -m gpu -x -q > $O/pytest_bands.log 2>&1; echo "rc=$?" >> $O/pytest_bands.log
timeout 600 python tools/band_cost.py 4 --passes --balance > $O/band_cost.txt 2>&1
for f in $O/pytest_a.log $O/pytest_parity.log $O/pytest_config5.log $O/pytest_bands.log $O/band_cost.txt; do tail -n 4 $f; done; head -c 400 $O/bench.json; echo; head -c 400 $O/bench_inorder.json
