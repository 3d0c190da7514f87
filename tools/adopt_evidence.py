"""After `gpurun -- bash tools/profile_round.sh <tag>` (+ the -m gpu suite's summary in gpurun_out/final/pytest_gpu.txt): turn gpurun_out/prof_<tag> into the tracked
profiles/<tag>_* set, retire the previous tag's files and re-point the documents.   python tools/adopt_evidence.py <new tag> <previous tag>
Then commit, run tools/rebench_records.sh <tag> on the GPU box (the bench records are stamped with the counter files they read) and copy gpurun_out/rebench/* to profiles/."""
import csv, glob, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.chdir(ROOT)
tag, prev = sys.argv[1], sys.argv[2]
digest = open("gpurun_out/prof_%s/source_digest.txt" % tag).read().strip()
prev_digest = [l for l in open("profiles/%s_isa_mix.txt" % prev)][0].split("digest")[1].split()[0]
subprocess.check_call([sys.executable, "tools/merge_pmc_hbm.py", tag], stdout=subprocess.DEVNULL)
subprocess.check_call([sys.executable, "tools/merge_sq_counters.py", tag], stdout=subprocess.DEVNULL)
with open("profiles/%s_isa_mix.txt" % tag, "w") as fh:
    subprocess.check_call([sys.executable, "tools/isa_mix.py", "profiles/%s_sq_counters.csv" % tag], stdout=fh, stderr=subprocess.DEVNULL)
# the readable table of the big kernels: the previous one's rows and remarks, the new counters
rows = [r for r in csv.reader(l for l in open("profiles/%s_sq_counters.csv" % tag) if not l.startswith("#"))]
hdr, rows = rows[0], rows[1:]
out = []
for line in open("profiles/%s_big_kernel_counters.txt" % prev).read().splitlines():
    if not line.startswith("plr::"):
        out.append(line.replace(prev, tag).replace(prev_digest, digest))
        continue
    key = line[:60].strip()
    match = [r for r in rows if r[0].replace(",", ";").startswith(key)]
    if not match:  # the kernel's template arguments changed since the previous set: the same kernel by its name in front of them, the launch with the most waves
        match = sorted([r for r in rows if r[0].split("<")[0] == key.split("<")[0]], key=lambda r: -float(dict(zip(hdr, r))["SQ_WAVES"]))
        key = match[0][0].replace(",", ";") if match else key
    if not match:
        continue
    r = match[0]
    d = dict(zip(hdr, r)); f = lambda k: float(d[k]); waves = f("SQ_WAVES"); cyc = f("GRBM_GUI_ACTIVE") / 8
    out.append("%-60s %7.1f %8d %9d %9.1f %9.1f %6.0f %% %8.2f %6.0f %% %6.0f %%" % (
        key[:58], cyc / 2400.0, waves, round(f("SQ_INSTS_VALU") / waves), f("SQ_INSTS_VMEM_RD") / waves, f("SQ_INSTS_LDS") / waves, 100 * f("TA_TA_BUSY_sum") / 256 / cyc,
        f("TCP_TOTAL_CACHE_ACCESSES_sum") / 256 / cyc, 100 * (1 - f("TCP_TCC_READ_REQ_sum") / f("TCP_TOTAL_CACHE_ACCESSES_sum")), 100 * f("TCC_HIT_sum") / (f("TCC_HIT_sum") + f("TCC_MISS_sum"))))
open("profiles/%s_big_kernel_counters.txt" % tag, "w").write("\n".join(out) + "\n")
with open("profiles/%s_pytest_gpu.txt" % tag, "w") as fh:
    fh.write("# python -m pytest tests -m gpu -q   (kernel source digest %s, MI355X box)\n" % digest + open("gpurun_out/final/pytest_gpu.txt").read())
# the reports the profile round wrote as they are
import shutil
for f in ("config5_8k.txt", "kernel_stats.csv", "parity_4k.txt", "parity_dense.txt", "pass_table.txt", "pass_table_producers.txt", "pass_table_producers_exact.txt", "tail_cost.txt",
          "tile_vs_band.txt", "band_timeline.txt", "bench_dense.json"):
    if os.path.exists("gpurun_out/prof_%s/%s" % (tag, f)): shutil.copy("gpurun_out/prof_%s/%s" % (tag, f), "profiles/%s_%s" % (tag, f))
same_round = tag[:3] == prev[:3]  # a previous ROUND's final set stays (the documents' history cites it); an earlier build of this round is replaced
for f in glob.glob("profiles/%s_*" % prev) if same_round else []:
    subprocess.check_call(["git", "rm", "-q", "-f", f])
docs = ["DESIGN.md", "BASELINE.md", "README.md", "INTEGRATION.md", "profiles/README.md"] + glob.glob("profiles/r04_*.txt") + glob.glob("profiles/r05_*.txt")
for p in docs if same_round else []:
    s = open(p).read()
    if prev in s: open(p, "w").write(s.replace(prev, tag))
print("adopted %s (digest %s); %d files" % (tag, digest, len(glob.glob("profiles/%s_*" % tag))))
