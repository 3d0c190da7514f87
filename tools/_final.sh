cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
mkdir -p gpurun_out/final
timeout -k 5 1500 python -m pytest tests -m gpu -q 2>&1 | grep -E "passed|failed|error" > gpurun_out/final/pytest_gpu.txt
timeout -k 5 1800 bash tools/profile_round.sh r04e > gpurun_out/final/profile_round.log 2>&1
timeout -k 5 200 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/final/smoke.txt 2>&1
timeout -k 5 200 python bench.py --force-bands --steps 100 --no-cpu-baseline 2>&1 | grep -o '"ms_per_step": [0-9.]*' > gpurun_out/final/force_bands.txt
cat gpurun_out/final/pytest_gpu.txt; tail -2 gpurun_out/final/smoke.txt; cat gpurun_out/final/force_bands.txt; cat gpurun_out/prof_r04e/tile_vs_band_proxy.txt; grep -o '"ms_per_step": [0-9.]*' gpurun_out/prof_r04e/bench*.json
