#!/bin/bash
# scratch: the whole GPU suite with durations
mkdir -p gpurun_out/r03s
( time python -m pytest tests/ -q -m gpu --durations=40 ) > gpurun_out/r03s/pytest_all.log 2>&1
grep -E "passed|failed|^real" gpurun_out/r03s/pytest_all.log
