"""host-side cost of one frame: total plrf_frame() wall time vs the part inside plr_render_frame (launching)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import bench
from plainrenderer_amd import RenderBackend
from plainrenderer_amd.frame import FramePipeline

class A: pass
args = A(); args.grid = 16; args.sdf_res = 64; args.shadow_res = 2048; args.steps = 200; args.warmup = 10; args.profile_frames = 0
w, h = 3840, 2160
be = RenderBackend(w, h, device=0)
fp = FramePipeline(be, w, h, shadow_map_res=2048)
scene, cams, inputs = bench.build_scene(args, "cuda:0", w, h)
inputs.upload(fp)
be.waitForGPUIdle()
for i in range(10):
    fp.frame(cams[i + 1], 1 / 60, 0.5)
be.waitForGPUIdle()
tot = []; launch = []
for i in range(100):
    be.waitForGPUIdle()
    t0 = time.perf_counter()
    fp.frame(cams[i + 11], 1 / 60, 0.5)
    tot.append(time.perf_counter() - t0)
    launch.append(be.getLastFrameCPUTime())
print("plrf_frame host total: median %.3f ms; inside plr_render_frame: median %.3f ms" % (np.median(tot) * 1e3, np.median(launch)))
