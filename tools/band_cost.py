"""Redundant compute of band rendering, measured on ONE GPU without the exchange: the N bands of the 7680 x (1080 N) frame are rendered one after
the other (each with the halo rows it recomputes, producers split into edge / interior dispatches as the overlapped exchange records them) and
their times are summed and compared with the time of the unpartitioned frame.
    python tools/band_cost.py [N] [--passes] [--balance]   (default N = 4: the 8K frame; --balance: band heights from measured band times, as bench.py does)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import bench
from plainrenderer_amd import RenderBackend, tiling
from plainrenderer_amd.frame import FramePipeline

n = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 4
class A: pass
args = A(); args.grid = 16; args.sdf_res = 64; args.shadow_res = 2048; args.steps = 40; args.warmup = 5; args.profile_frames = 0
w, h = 7680, 1080 * n


def measure(band):
    be = RenderBackend(w, h, device=0)
    kw = dict(band_row_begin=band[0], band_row_end=band[1]) if band else {}
    fp = FramePipeline(be, w, h, shadow_map_res=2048, **kw)
    if band:
        fp.set_exchange_callback(lambda exchange_id, stream: None)  # callbacks that move nothing: the recording is the multi-GPU one
    scene, cams, inputs = bench.build_scene(args, "cuda:0", w, h, band)
    inputs.upload(fp)
    be.waitForGPUIdle()
    for i in range(args.warmup):
        fp.frame(cams[i + 1], 1 / 60, 0.5)
    be.waitForGPUIdle()
    t0 = time.perf_counter()
    for i in range(args.steps):
        fp.frame(cams[i + 6], 1 / 60, 0.5)
    be.waitForGPUIdle()
    ms = (time.perf_counter() - t0) * 1e3 / args.steps
    be.setPassTiming(True)
    acc = {}
    for i in range(8):
        fp.frame(cams[i + 20], 1 / 60, 0.5)
        for name, t in be.getRenderpassTimings():
            acc[name] = acc.get(name, 0.0) + t / 8
    fp.destroy()
    be.shutdown()
    return ms, acc


full, full_passes = measure(None)
print("unpartitioned %dx%d frame: %.3f ms" % (w, h, full))
bounds = tiling.equal_bounds(h, n)
if "--balance" in sys.argv:
    # what bench.py --gpus N does before its timed region (static load balancing from measured band times), here with the bands one after the other
    for it in range(2):
        times = [measure((bounds[i], bounds[i + 1]))[0] for i in range(n)]
        print("partition %s: band times %s ms, slowest / mean = %.3f" % (bounds, ["%.3f" % t for t in times], max(times) / (sum(times) / n)))
        new = tiling.balanced_bounds(h, bounds, times)
        if new == bounds:
            break
        bounds = new
total, band_passes, slowest = 0.0, {}, 0.0
for i in range(n):
    band = (bounds[i], bounds[i + 1])
    ms, acc = measure(band)
    total += ms
    slowest = max(slowest, ms)
    for k, v in acc.items():
        band_passes[k] = band_passes.get(k, 0.0) + v
    print("band %d of %d (rows %d..%d): %.3f ms" % (i, n, band[0], band[1], ms))
def group(d):  # fused launches of the unpartitioned frame count towards their passes' group
    g = {}
    for k, v in d.items():
        key = "GI trace + filters" if ("Indirect" in k) else ("bloom" if "loom" in k or "Tonemap" in k else ("exposure" if "istogram" in k or "expose" in k else k.split(" + ")[0]))
        g[key] = g.get(key, 0.0) + v
    return g
if "--passes" in sys.argv:
    for k in sorted(set(full_passes) | set(band_passes), key=lambda k: -band_passes.get(k, 0.0)):
        print("    %-60s bands %.4f ms (%.4f per band), unpartitioned %.4f ms" % (k[:60], band_passes.get(k, 0.0), band_passes.get(k, 0.0) / n, full_passes.get(k, 0.0)))
gf, gb = group(full_passes), group(band_passes)
for k in sorted(gb, key=lambda k: -gb[k]):
    print("  %-34s bands %.3f ms, unpartitioned %.3f ms (%+.1f %%)" % (k, gb[k], gf.get(k, 0.0), 100.0 * (gb[k] / gf[k] - 1.0) if gf.get(k) else 0.0))
print("sum of the bands %.3f ms = %.3f x the unpartitioned frame: %.1f %% redundant compute; slowest band %.3f ms -> at most %.2fx on %d GPUs before any exchange wait" % (
    total, total / full, 100.0 * (total / full - 1.0), slowest, full / slowest, n))
