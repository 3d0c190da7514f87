"""Join the per-pass counter summaries tools/profile_round.sh leaves in gpurun_out/prof_<tag>/pmc_sq*.csv (tools/pmc_summary.py: one row per kernel, mean
per dispatch) into profiles/<tag>_sq_counters.csv, stamped with the kernel source digest of the build. bench.py's valu_roofline / l1_roofline and
tools/isa_mix.py read it. Usage: python tools/merge_sq_counters.py r04a"""
import csv, glob, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1]
src = os.path.join(ROOT, "gpurun_out", "prof_" + tag)
rows, cols = {}, []
for f in sorted(glob.glob(os.path.join(src, "pmc_sq*.csv"))):
    for r in csv.DictReader(open(f)):
        k = r["kernel"]
        rows.setdefault(k, {"dispatches": r["dispatches"]})
        for c, v in r.items():
            if c in ("kernel", "dispatches"): continue
            if c not in cols: cols.append(c)
            rows[k][c] = v
out = os.path.join(ROOT, "profiles", tag + "_sq_counters.csv")
with open(out, "w") as fh:
    fh.write("# rocprofv3 --pmc <one group per pass> --kernel-include-regex plr:: -- python bench.py --steps 3 --warmup 2 --profile-frames 0; mean per dispatch (tools/pmc_summary.py)\n")
    fh.write("# SQ_* in wave-instructions / quad-cycles, TA_* / TCP_* / TCC_* summed over the chip's 256 CUs / 16 channels x 8 XCDs, GRBM_GUI_ACTIVE summed over the 8 XCDs (cycles)\n")
    dig = os.path.join(src, "source_digest.txt")
    if os.path.exists(dig):
        fh.write("# kernel source digest: %s\n" % open(dig).read().strip())
    fh.write("kernel,dispatches," + ",".join(cols) + "\n")
    for k in sorted(rows, key=lambda k: -float(rows[k].get("SQ_INSTS_VALU", 0) or 0)):
        fh.write(k + "," + rows[k]["dispatches"] + "," + ",".join(rows[k].get(c, "") for c in cols) + "\n")
print(open(out).read()[:3000])
