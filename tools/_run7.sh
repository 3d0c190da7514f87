set -u
O=gpurun_out/r03g; mkdir -p $O
export TMPDIR=/tmp
timeout 1800 python -m pytest tests -m gpu -x -q > $O/pytest_all.log 2>&1; echo "rc=$?" >> $O/pytest_all.log
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1
bash tools/profile_round.sh r03a > $O/profile_round.log 2>&1; echo "rc=$?" >> $O/profile_round.log
tail -n 5 $O/pytest_all.log; tail -n 2 $O/smoke.log; tail -n 5 $O/profile_round.log; head -c 300 gpurun_out/prof_r03a/bench.json
