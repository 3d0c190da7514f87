set -u
O=gpurun_out/r03z; mkdir -p $O
export TMPDIR=/tmp
bash tools/profile_round.sh r03d > $O/profile_round.log 2>&1; echo "rc=$?" >> $O/profile_round.log
timeout 2400 python -m pytest tests -m gpu -q > $O/pytest_all.log 2>&1; echo "rc=$?" >> $O/pytest_all.log
python bench.py > $O/bench_with_cpu.json 2> $O/bench_with_cpu.err
tail -n 4 $O/pytest_all.log; tail -n 3 $O/profile_round.log; head -c 300 gpurun_out/prof_r03d/bench.json
