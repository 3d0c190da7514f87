"""Which kernels of neighbouring frames ran beside each other, from a rocprofv3 --kernel-trace CSV of bench.py.
    python tools/overlap_probe.py <kernel_trace.csv> [label]
Per kernel family: launches, mean duration, and the mean time per launch during which at least one kernel of another family was running on the GPU,
broken down by that family. Frame period = mean distance between consecutive starts of the frame's first kernel."""
import csv, sys
from collections import defaultdict

FAMILIES = [("histogramAndPyramid", "front 1 (histogram + pyramid)"), ("exposureChainAndPyramidTail", "front 2 (exposure + pyramid tail)"), ("frustumAndTileCulling", "culling"),
            ("sdfDiffuseTrace", "trace"), ("spatialFilter", "spatial filter"), ("temporalGiFilter", "temporal GI"), ("upscaleAndShade", "upscale + shade"), ("shadeDirect", "shade: direct (early)"), ("upscaleAndCombine", "shade: upscale + combine"),
            ("temporalFilterStrip", "TAA"), ("bloom", "bloom chain"), ("applyBloomTonemap", "apply + tonemap")]


def family(name):
    for key, label in FAMILIES:
        if key in name:
            return label
    return None


def main(path, label=""):
    rows = []
    for r in csv.DictReader(open(path)):
        f = family(r["Kernel_Name"])
        if f:
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), f))
    rows.sort()
    # drop the first and last 20 % (warm-up, auxiliary frames with per-pass events)
    n = len(rows)
    rows = rows[n // 5: n - n // 5]
    starts = [s for s, e, f in rows if f == FAMILIES[0][1]]
    period = (starts[-1] - starts[0]) / max(len(starts) - 1, 1) / 1e3
    dur, cnt = defaultdict(float), defaultdict(int)
    over = defaultdict(lambda: defaultdict(float))
    for i, (s, e, f) in enumerate(rows):
        dur[f] += (e - s) / 1e3
        cnt[f] += 1
        j = i - 1
        while j >= 0 and rows[j][0] > s - 2_000_000:  # kernels that started up to 2 ms earlier
            s2, e2, f2 = rows[j]
            if f2 != f and e2 > s:
                over[f][f2] += (min(e, e2) - s) / 1e3
            j -= 1
        j = i + 1
        while j < len(rows) and rows[j][0] < e:
            s2, e2, f2 = rows[j]
            if f2 != f:
                over[f][f2] += (min(e, e2) - s2) / 1e3
            j += 1
    # idle time in front of every launch: from the end of the latest kernel that ended before it started, on any stream (0 if another kernel was still running)
    gap, prev_end = defaultdict(float), None
    running_end = 0
    for s_, e_, f_ in rows:
        if running_end and s_ > running_end:
            gap[f_] += (s_ - running_end) / 1e3
        running_end = max(running_end, e_)
    busy = sum(gap.values()) / max(len(starts), 1)
    print("## %s: frame period %.1f us (%d frames); GPU idle between kernels %.1f us per frame" % (label, period, len(starts), busy))
    print("%-34s %8s %10s   %s" % ("kernel family", "launches", "mean us", "mean us per launch beside ..."))
    for key, lab in FAMILIES:
        if not cnt[lab]:
            continue
        o = ", ".join("%s %.1f" % (k, v / cnt[lab]) for k, v in sorted(over[lab].items(), key=lambda kv: -kv[1]) if v / cnt[lab] >= 0.5)
        print("%-34s %8d %10.1f   idle before %.1f; %s" % (lab, cnt[lab], dur[lab] / cnt[lab], gap[lab] / cnt[lab], o))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else "")
