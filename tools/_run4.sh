set -u
O=gpurun_out/r03d; mkdir -p $O
export TMPDIR=/tmp
python bench.py --no-cpu-baseline --pass-table > $O/bench.json 2> $O/pass_table.txt
PLR_ASYNC_TAIL=0 python bench.py --no-cpu-baseline --pass-table --steps 300 > $O/bench_inorder.json 2> $O/pass_table_inorder.txt
PLR_PASS_FUSION=0 python bench.py --no-cpu-baseline --pass-table --steps 300 > $O/bench_nofusion.json 2> $O/pass_table_nofusion.txt
timeout 900 python -m pytest tests/test_fusion.py tests/test_parity_fullsize.py -m gpu -x -q -s -k "fusion or fused or shading or frame or async" > $O/pytest_a.log 2>&1; echo "rc=$?" >> $O/pytest_a.log
timeout 1200 python -m pytest tests/test_variants_parity.py -m gpu -q -s > $O/pytest_variants.log 2>&1; echo "rc=$?" >> $O/pytest_variants.log
timeout 900 python -m pytest tests/test_config5_8k.py -m gpu -x -q -s > $O/pytest_config5.log 2>&1; echo "rc=$?" >> $O/pytest_config5.log
timeout 600 python tools/band_cost.py 4 --passes --balance > $O/band_cost.txt 2>&1
for f in $O/pytest_a.log $O/pytest_variants.log $O/pytest_config5.log $O/band_cost.txt; do tail -n 4 $f; done; head -c 400 $O/bench.json; echo; head -c 400 $O/bench_inorder.json
