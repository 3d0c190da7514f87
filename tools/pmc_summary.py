"""Aggregate a rocprofv3 counter_collection CSV per kernel (mean counter value per dispatch) -> small CSV for profiles/."""
import csv
import sys
from collections import defaultdict


def main(src, dst):
    acc = defaultdict(lambda: defaultdict(float))
    cnt = defaultdict(lambda: defaultdict(int))
    with open(src) as f:
        for r in csv.DictReader(f):
            name = r.get("Kernel_Name", "")
            if "plr::" not in name:
                continue
            name = name.replace("void ", "").split("(")[0]
            c = r["Counter_Name"]
            acc[name][c] += float(r["Counter_Value"])
            cnt[name][c] += 1
    counters = sorted({c for k in acc.values() for c in k})
    with open(dst, "w") as f:
        f.write("kernel,dispatches," + ",".join(counters) + "\n")
        for k in sorted(acc):
            n = max(cnt[k].values())
            f.write(k.replace(",", ";") + "," + str(n) + "," + ",".join("%.1f" % (acc[k][c] / max(cnt[k][c], 1)) for c in counters) + "\n")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
