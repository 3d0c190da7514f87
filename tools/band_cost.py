"""Compute cost of ONE band of a row-partitioned frame on one GPU, without the exchange: how much of a 4K frame's time a
7680 x ~1080 band of the 7680 x (1080 N) frame costs (halo rows are computed redundantly). Usage: python tools/band_cost.py [N] [band index]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import bench
from plainrenderer_amd import RenderBackend, tiling
from plainrenderer_amd.frame import FramePipeline

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4
index = int(sys.argv[2]) if len(sys.argv) > 2 else 1
class A: pass
args = A(); args.grid = 16; args.sdf_res = 64; args.shadow_res = 2048; args.steps = 40; args.warmup = 5; args.profile_frames = 10
w, h = 7680, 1080 * n
band = tiling.band_rows(h, n, index)
be = RenderBackend(w, h, device=0)
fp = FramePipeline(be, w, h, shadow_map_res=2048, band_row_begin=band[0], band_row_end=band[1])
if "--split" in sys.argv:  # record the producers edge-rows-first as the overlapped exchange does (callbacks that move nothing)
    fp.set_exchange_callback(lambda exchange_id, stream: None)
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
for i in range(args.profile_frames):
    fp.frame(cams[i + 46], 1 / 60, 0.5)
    for name, t in be.getRenderpassTimings():
        acc.setdefault(name, []).append(t)
rows = band[1] - band[0]
print("band %d of %d: rows %d..%d (%d rows of %d), %dx%d pixels = %.3f of a 3840x2160 frame" % (index, n, band[0], band[1], rows, h, w, rows, w * rows / 8294400.0))
print("frame time %.3f ms (no exchange)" % ms)
tot = 0
for name, v in sorted(acc.items(), key=lambda kv: -np.mean(kv[1]) * len(kv[1])):
    t = np.mean(v) * len(v) / args.profile_frames
    tot += t
    print("  %-40s %.4f ms" % (name, t))
print("  sum %.3f ms" % tot)
