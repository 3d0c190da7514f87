# A/B of the two-phase spatial filter of a partitioned frame with request lists (tools/band_cost.py, 2 x 2 tiles at 8K, equal partition): environment hooks of the build
mkdir -p gpurun_out/r06c_try
for v in "$@"; do
  name=${v%%:*}; kv=${v#*:}
  env $kv python tools/band_cost.py 4 --tiles 2x2 --requested --passes > gpurun_out/r06c_try/cost_${name}.txt 2>&1
  echo "== $name ($kv)"; grep -E "^partition . of|spatial filter  |slowest" gpurun_out/r06c_try/cost_${name}.txt | cut -c1-150
done
