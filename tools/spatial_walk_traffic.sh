# HBM traffic of one spatial-filter launch per walk configuration (rocprofv3 --pmc FETCH_SIZE, the gfx950 correction of profiles/*_pmc_hbm.csv: traffic = 2 x FETCH_SIZE + WRITE_SIZE;
# WRITE_SIZE is 24300 KiB for every walk): bash tools/spatial_walk_traffic.sh "2:3 2:4 ..."
export TMPDIR=/tmp; R=$(pwd); cd /tmp
for v in ${1:-1:2 2:2 2:3 2:4}; do
  sx=${v%%:*}; ch=${v#*:}
  rm -rf /tmp/pmcw
  PLR_SPATIAL_SPLIT_X=$sx PLR_SPATIAL_CHUNKS=$ch timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-include-regex "spatialFilterFast" -d /tmp/pmcw -o pmc --output-format csv -- python $R/bench.py --steps 4 --warmup 2 --profile-frames 0 --no-cpu-baseline > /dev/null 2>&1
  python - <<PY
import csv, glob
f = glob.glob("/tmp/pmcw/**/*counter_collection.csv", recursive=True)[0]
vals = [float(r["Counter_Value"]) for r in csv.DictReader(open(f)) if r["Counter_Name"] == "FETCH_SIZE"]
m = sum(vals) / max(len(vals), 1)
print("splitX $sx chunks $ch: FETCH_SIZE %.1f KiB -> %.1f MB = %.2f x algorithmic" % (m, (2 * m + 24300) * 1024 / 1e6, (2 * m + 24300) * 1024 / 62208000))
PY
done
