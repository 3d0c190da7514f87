# One rank's frame of the partitioned 8K frame as a kernel timeline (rocprofv3 --kernel-trace around tools/band_cost.py): what runs when, on which queue, and the gaps.
#   bash tools/band_timeline.sh [band_cost.py arguments, default: 4 --tiles 2x2 --requested]
export TMPDIR=/tmp
REPO=$(pwd); OUT=$REPO/gpurun_out/band_timeline; mkdir -p $OUT; cd /tmp
ARGS="${@:-4 --tiles 2x2 --requested}"
timeout 600 rocprofv3 --kernel-trace -d $OUT/kt -o kt --output-format csv -- python $REPO/tools/band_cost.py $ARGS > $OUT/band_cost.txt 2> $OUT/kt.err
F=$(ls $OUT/kt/*kernel_trace.csv | head -1)
python - > $OUT/timeline.txt <<PY
import csv
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Queue_Id"], r["Kernel_Name"]) for r in csv.DictReader(open("$F"))]
rows.sort()
# the partitions are timed one after the other, the unpartitioned frame last: frames of a partition start with the histogram kernel; print the frame in the middle of
# the third quarter of the histogram launches that belong to partitioned frames (rank 2 of 4 at the default arguments)
starts = [i for i, r in enumerate(rows) if "histogramAndPyramid" in r[3]]
print("# %d kernel dispatches, %d frames" % (len(rows), len(starts)))
# which hardware queue do a rank's launch stream, asynchronous tail and communication stream sit on? (the four ranks and the unpartitioned pipeline of this one process share
# the device's hardware queues; a rank per process has three streams in use and a queue for each.) Runs of consecutive frames with the same triple:
def queue_of(k, k1, what):
    qs = [rows[i][2] for i in range(k, k1) if what in rows[i][3]]
    return max(set(qs), key=qs.count) if qs else "-"
runs = []
for a, b in zip(starts, starts[1:]):
    key = (queue_of(a, b, "histogramAndPyramid"), queue_of(a, b, "bloomDownsample"), queue_of(a, b, "requestScan"))
    us = (rows[b][0] - rows[a][0]) / 1e3
    if runs and runs[-1][0] == key: runs[-1][1].append(us)
    else: runs.append([key, [us]])
print("# frames by (launch queue, tail queue, communication queue): count, median us from frame start to frame start")
for key, v in runs:
    if len(v) >= 8: print("#   launch q%s tail q%s comm q%s: %3d frames, median %7.1f us%s" % (key[0], key[1], key[2], len(v), sorted(v)[len(v) // 2], "   <- the tail shares the launch stream's queue: no overlap" if key[0] == key[1] else ""))
for frac in (0.45, 0.55, 0.65):
    k = starts[int(len(starts) * frac)]
    k1 = [i for i in starts if i > k][0]
    t0 = rows[k][0]
    print("# frame at dispatch %d (%.0f %% of the run): %d dispatches, %.1f us from the first kernel's start to the next frame's" % (k, 100 * frac, k1 - k, (rows[k1][0] - t0) / 1e3))
    lastEnd = {}
    for s, e, q, nm in rows[k:k1 + 2]:
        short = nm.replace("(anonymous namespace)::", "").replace("plr::", "").replace("void ", "").split("(")[0][:70]
        print("%9.1f %9.1f  %7.1f us  q%-3s %s" % ((s - t0) / 1e3, (e - t0) / 1e3, (e - s) / 1e3, q, short))
PY
head -100 $OUT/timeline.txt
rm -rf $OUT/kt
