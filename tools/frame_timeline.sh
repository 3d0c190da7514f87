export TMPDIR=/tmp
REPO=$(pwd); OUT=$REPO/gpurun_out/r04_overlap; mkdir -p $OUT; cd /tmp
timeout 300 rocprofv3 --kernel-trace --memory-copy-trace -d $OUT/kt -o kt --output-format csv -- python $REPO/bench.py --steps 100 --warmup 10 --profile-frames 0 --no-cpu-baseline > /dev/null 2> $OUT/kt.err
ls $OUT/kt
F=$(ls $OUT/kt/*kernel_trace.csv | head -1)
M=$(ls $OUT/kt/*memory_copy_trace.csv | head -1)
head -2 $M
python - <<PY
import csv
rows=[(int(r["Start_Timestamp"]),int(r["End_Timestamp"]),"K q%s %s"%(r["Queue_Id"],r["Kernel_Name"][:50])) for r in csv.DictReader(open("$F"))]
try:
    rows+=[(int(r["Start_Timestamp"]),int(r["End_Timestamp"]),"COPY %s %s B"%(r.get("Direction",""),r.get("Bytes", r.get("Size","?")))) for r in csv.DictReader(open("$M"))]
except Exception as e: print("copy trace:", e)
rows.sort()
n=len(rows)
# find a TAA kernel in the middle and print the 40 events after it
idx=[i for i,r in enumerate(rows) if "temporalFilterStrip" in r[2]]
i0=idx[len(idx)//2]
t0=rows[i0][0]
for s,e,nm in rows[i0:i0+30]: print("%8.1f %8.1f %s"%((s-t0)/1e3,(e-t0)/1e3,nm))
PY
rm -rf $OUT/kt
