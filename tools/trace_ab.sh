for V in 0 1; do
  echo "PLR_TRACE_PER_LANE=$V"
  PLR_TRACE_PER_LANE=$V python bench.py --no-cpu-baseline --steps 300 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('ms', d['ms_per_step'], 'trace', d['passes_ms']['Indirect diffuse SDF trace'])"
done
python bench.py --producers --steps 50 --no-cpu-baseline 2>&1 | tail -c 300
PLR_PARITY_SIZE=1920x1088 timeout 900 python -m pytest tests/test_parity_fullsize.py tests/test_sdfgi.py tests/test_fusion.py -m gpu -x -q -s -k "trace or sdfgi or fusion" 2>&1 | grep -E "PARITY trace|passed|failed"
python tools/band_cost.py 4 --balance 2>&1 | grep -v amdgpu | tail -6
