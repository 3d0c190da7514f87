"""Merge the FETCH_SIZE and WRITE_SIZE summaries tools/profile_round.sh leaves in gpurun_out/prof_<tag>/ into profiles/<tag>_pmc_hbm.csv
and copy the other round evidence next to it. Usage: python tools/merge_pmc_hbm.py r01d"""
import csv, os, shutil, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1]
src = os.path.join(ROOT, "gpurun_out", "prof_" + tag)
def read(name, col):
    return {r["kernel"]: (int(r["dispatches"]), float(r[col])) for r in csv.DictReader(open(os.path.join(src, name)))}
f, w = read("pmc_FETCH_SIZE.csv", "FETCH_SIZE"), read("pmc_WRITE_SIZE.csv", "WRITE_SIZE")
rows = []
for k in f:
    if k not in w: continue
    rows.append((k, f[k][0], f[k][1], w[k][1], int(round((2.0 * f[k][1] + w[k][1]) * 1024.0))))
rows.sort(key=lambda r: -r[4])
out = os.path.join(ROOT, "profiles", tag + "_pmc_hbm.csv")
with open(out, "w") as fh:
    fh.write("# rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, --kernel-include-regex plr::), mean per dispatch, python bench.py --steps 4 --warmup 2\n")
    fh.write("# units: KiB. gfx950 correction (MI355X_MICROARCH.md, HBM): FETCH_SIZE tallies 128-B requests at 64 B -> traffic = 2*FETCH_SIZE + WRITE_SIZE;\n")
    fh.write("# calibrated on tonemapping (reads 4 B/px) and applyBloom (reads 8 B/px): 2*FETCH_SIZE matches the algorithmic reads within 0.5 %\n")
    dig = os.path.join(src, "source_digest.txt")
    if os.path.exists(dig):
        fh.write("# kernel source digest: %s\n" % open(dig).read().strip())
    fh.write("kernel,dispatches,FETCH_SIZE_KiB,WRITE_SIZE_KiB,traffic_bytes\n")
    for r in rows: fh.write("%s,%d,%.1f,%.1f,%d\n" % r)
for a in ("bench.json", "bench_under_rocprof.json", "kernel_stats.csv", "pass_table.txt", "bench_1080p.json", "bench_8k.json", "parity_4k.txt", "config5_8k.txt", "valu_rates.txt", "band_cost.txt", "pass_table_producers.txt", "pass_table_producers_exact.txt", "bench_producers.json", "tile_vs_band.txt", "config5_series.txt", "tail_cost.txt"):
    if os.path.exists(os.path.join(src, a)):
        shutil.copy(os.path.join(src, a), os.path.join(ROOT, "profiles", tag + "_" + a))
print(open(out).read()[:1500])
