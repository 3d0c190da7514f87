for i in 1 2 3; do python bench.py --no-cpu-baseline --steps 600 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('ms', d['ms_per_step'], 'host', d['host_ms_per_step'], d['host_ms_per_frame_idle_gpu'])"; done
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | grep -E "passed|failed|Error" | tail -5
