# What is the fixed 0.40 ms of bench.py's timed region (profiles/r06_steps_vs_ms.txt)? Kernel trace of the driver's command: the last 20 frames' kernels.
export TMPDIR=/tmp; R=$(pwd); O=$R/gpurun_out/timed_region; mkdir -p $O; cd /tmp
rm -rf /tmp/ktr; timeout 300 rocprofv3 --kernel-trace -d /tmp/ktr -o kt --output-format csv -- python $R/bench.py --steps 20 --warmup 5 --profile-frames 0 --no-cpu-baseline > $O/bench.json 2>/dev/null
python - <<PY
import csv, glob, json
f = glob.glob("/tmp/ktr/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(f)))
starts = [i for i, r in enumerate(rows) if "histogramAndPyramid" in r[2]]
last20 = starts[-20:]
t0 = rows[last20[0]][0]
periods = [(rows[b][0] - rows[a][0]) / 1e3 for a, b in zip(last20, last20[1:])]
end = max(r[1] for r in rows[last20[0]:])
prev_end = max(r[1] for r in rows[:last20[0]])
print("ms_per_step reported by this run: %s" % json.load(open("$O/bench.json"))["ms_per_step"])
print("the 20 timed frames: first kernel start -> last kernel end %.1f us = 20 x %.1f" % ((end - t0) / 1e3, (end - t0) / 2e4))
print("frame period (start to start), 19 intervals: median %.1f us, first three %s, last three %s" % (sorted(periods)[9], ["%.1f" % p for p in periods[:3]], ["%.1f" % p for p in periods[-3:]]))
print("last frame: its first kernel's start -> the run's last kernel's end %.1f us (a period + the tail draining alone)" % ((end - rows[last20[-1]][0]) / 1e3))
print("idle gap in front of the timed region (last kernel of the warm-up -> first timed kernel): %.1f us (synchronise + barrier + the first frame's recording)" % ((t0 - prev_end) / 1e3))
PY
