#!/bin/bash
mkdir -p gpurun_out/r03y
python -m pytest tests/ -q -m gpu -x > gpurun_out/r03y/pytest_all.log 2>&1
grep -E "passed|failed" gpurun_out/r03y/pytest_all.log
for i in 1 2; do python bench.py --steps 600 --warmup 20 --no-cpu-baseline --pass-table 2>&1 >/dev/null | grep -E "Forward shading|sum of"; done
