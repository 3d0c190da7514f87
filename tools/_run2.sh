set -u
O=gpurun_out/r03b; mkdir -p $O
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_fusion.py tests/test_fast_kernels.py tests/test_hiz_bloom_taa.py tests/test_sdfgi.py -m gpu -x -q > $O/pytest_a.log 2>&1; echo "rc=$?" >> $O/pytest_a.log
timeout 900 python -m pytest tests/test_parity_fullsize.py -m gpu -q -s > $O/pytest_parity.log 2>&1; echo "rc=$?" >> $O/pytest_parity.log
timeout 600 python -m pytest tests/test_bands.py tests/test_golden_frame.py tests/test_full_frame.py -m gpu -x -q > $O/pytest_bands.log 2>&1; echo "rc=$?" >> $O/pytest_bands.log
python bench.py --no-cpu-baseline --pass-table > $O/bench.json 2> $O/pass_table.txt
timeout 900 python -m pytest tests/test_config5_8k.py -m gpu -x -q -s > $O/pytest_config5.log 2>&1; echo "rc=$?" >> $O/pytest_config5.log
timeout 600 python tools/band_cost.py 4 --passes --balance > $O/band_cost.txt 2>&1
for f in $O/pytest_a.log $O/pytest_parity.log $O/pytest_bands.log $O/pytest_config5.log $O/band_cost.txt; do tail -n 4 $f; done; head -c 600 $O/bench.json
