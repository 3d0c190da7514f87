# The spatial GI filter's block -> tile walk (device/xcd.h): XCD columns x chunks per XCD, through the build's environment hooks. One GPU, the 4K bench frame:
# ms per frame and the filter's pass time (two executions) per configuration.   bash tools/spatial_walk_ab.sh "1:0 2:1 2:2 4:1 4:2 ..."   (splitX:chunks, 0 = default)
for v in ${1:-1:0 2:1 2:2 2:4 4:1 4:2 4:4 8:2 8:4 1:0}; do
  sx=${v%%:*}; ch=${v#*:}
  PLR_SPATIAL_SPLIT_X=$sx PLR_SPATIAL_CHUNKS=$ch python bench.py --steps 300 --warmup 20 --no-cpu-baseline ${SPATIAL_WALK_ARGS:-} 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read())
p = d.get('passes_ms', {})
sp = [v for k, v in p.items() if 'spatial filter' in k]
print('splitX $sx chunks/XCD $ch: %.4f ms per frame, spatial filter %s us' % (d['ms_per_step'], ' + '.join('%.1f' % (1000 * x) for x in sp)))"
done
