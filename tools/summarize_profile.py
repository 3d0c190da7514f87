"""Condense a rocprofv3 output directory into the small CSVs kept under profiles/ (plr:: kernels only; the torch kernels in
a bench run belong to synthetic input generation)."""
import csv
import sys


def main(src_stats, dst, pmc=None):
    rows = [r for r in csv.DictReader(open(src_stats)) if r["Name"].startswith("void plr::") or r["Name"].startswith("plr::")]
    total = sum(float(r["TotalDurationNs"]) for r in rows)
    with open(dst, "w") as f:
        f.write("kernel,calls,avg_us,min_us,max_us,total_ms,pct_of_plr\n")
        for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"])):
            name = r["Name"].replace("void ", "").split("(")[0]
            f.write("%s,%s,%.2f,%.2f,%.2f,%.3f,%.2f\n" % (name, r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3,
                                                       float(r["TotalDurationNs"]) / 1e6, 100 * float(r["TotalDurationNs"]) / total))


if __name__ == "__main__":
    main(*sys.argv[1:])
