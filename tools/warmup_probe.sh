# Why does `bench.py --steps 20 --warmup 5` (the driver's flags) read 3 % more per frame than the default 600 steps? Not the warm-up: a fixed cost per timed region.
#   bash tools/warmup_probe.sh   (one GPU; prints ms per step for 20 timed steps after W warm-up frames, then for K timed steps after 5)
for w in 5 20 50 200 600 5 20; do
  python bench.py --steps 20 --warmup $w --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('warmup %4d steps %4d: %.4f ms per step' % (d['warmup'], d['steps'], d['ms_per_step']))"
done
for s in 20 50 100 200; do
  python bench.py --steps $s --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('warmup %4d steps %4d: %.4f ms per step' % (d['warmup'], d['steps'], d['ms_per_step']))"
done
