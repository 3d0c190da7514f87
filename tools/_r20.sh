#!/bin/bash
# scratch: suite + bench
mkdir -p gpurun_out/r03t
python -m pytest tests/ -q -m gpu -x > gpurun_out/r03t/pytest_all.log 2>&1
grep -E "passed|failed" gpurun_out/r03t/pytest_all.log
python bench.py --steps 600 --warmup 20 --no-cpu-baseline --pass-table > gpurun_out/r03t/bench.json 2> gpurun_out/r03t/pass_table.txt
head -8 gpurun_out/r03t/pass_table.txt; tail -1 gpurun_out/r03t/pass_table.txt
python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('steps',d['steps'],'warmup',d['warmup'],'ms_per_step',d['ms_per_step'])"
