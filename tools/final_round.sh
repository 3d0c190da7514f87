#!/bin/bash
# The round's evidence from ONE build, on the GPU box:  bash tools/final_round.sh r05c
# -> gpurun_out/prof_<tag>/ (tools/profile_round.sh), gpurun_out/final/pytest_gpu.txt (the whole -m gpu suite), then tools/adopt_evidence.py <tag> <previous tag> here.
set -u
TAG=${1:-r00}
export TMPDIR=/tmp
mkdir -p gpurun_out/final
bash tools/profile_round.sh $TAG > gpurun_out/final/profile_round.log 2>&1; echo "rc=$?" >> gpurun_out/final/profile_round.log
timeout 2400 python -m pytest tests -m gpu -q 2>&1 | grep -E "passed|failed|error" | tail -3 > gpurun_out/final/pytest_gpu.txt
tail -n 3 gpurun_out/final/profile_round.log; cat gpurun_out/final/pytest_gpu.txt
