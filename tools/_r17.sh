#!/bin/bash
# scratch: the whole GPU suite with durations
mkdir -p gpurun_out/r03q
( time python -m pytest tests/ -q -m gpu --durations=100 ) > gpurun_out/r03q/pytest_all.log 2>&1
tail -5 gpurun_out/r03q/pytest_all.log
