set -u
O=gpurun_out/r03i; mkdir -p $O
export TMPDIR=/tmp
python bench.py --no-cpu-baseline --pass-table > $O/bench.json 2> $O/pass_table.txt
timeout 1500 python -m pytest tests/test_hiz_bloom_taa.py tests/test_parity_fullsize.py tests/test_variants_parity.py tests/test_fusion.py tests/test_golden_frame.py -m gpu -q -s -k "taa or shad or fused or frame or fusion or async or golden" > $O/pytest_a.log 2>&1; echo "rc=$?" >> $O/pytest_a.log
grep -E "PARITY (taa|shade|fused|frame)|VARIANT taa|passed|failed" $O/pytest_a.log | tail -30; head -c 300 $O/bench.json; echo; grep -E "Temporal filtering|upscale|sum of" $O/pass_table.txt
