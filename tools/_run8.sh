set -u
O=gpurun_out/r03h; mkdir -p $O
export TMPDIR=/tmp
timeout 2400 python -m pytest tests -m gpu -q > $O/pytest_all.log 2>&1; echo "rc=$?" >> $O/pytest_all.log
python bench.py --no-cpu-baseline --pass-table > $O/bench.json 2> $O/pass_table.txt
timeout 600 python tools/band_cost.py 4 --passes --balance > $O/band_cost.txt 2>&1
tail -n 8 $O/pytest_all.log; head -c 300 $O/bench.json; echo; grep -E "unpartitioned 7680|band . of|sum of the bands" $O/band_cost.txt
