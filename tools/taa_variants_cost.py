"""diagnostic: what each feature of the TAA resolve costs at 4K - the pass's hipEvent time inside the benchmark frame for several settings (the default is the workload)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from plainrenderer_amd import RenderBackend
from plainrenderer_amd.frame import FramePipeline
class A: pass
args = A(); args.grid = 16; args.sdf_res = 64; args.shadow_res = 2048; args.steps = 40; args.warmup = 5; args.profile_frames = 0
w, h = 3840, 2160
cases = [("default (clip, dilate, Bicubic1Tap, tonemap)", {}), ("bilinear history", dict(taa_history_sampling_tech=0)), ("no tonemap", dict(taa_filter_use_tonemapping=0)),
         ("clamp instead of clip", dict(taa_use_clipping=0)), ("no dilation", dict(taa_use_motion_vector_dilation=0)),
         ("bilinear, no tonemap, clamp, no dilation", dict(taa_history_sampling_tech=0, taa_filter_use_tonemapping=0, taa_use_clipping=0, taa_use_motion_vector_dilation=0))]
for label, kw in cases:
    be = RenderBackend(w, h, device=0)
    fp = FramePipeline(be, w, h, shadow_map_res=2048, **kw)
    scene, cams, inputs = bench.build_scene(args, "cuda:0", w, h)
    inputs.upload(fp)
    for i in range(20): fp.frame(cams[i + 1], 1 / 60, 0.5)
    be.setPassTiming(True)
    acc = 0.0
    for i in range(20):
        fp.frame(cams[i + 21], 1 / 60, 0.5)
        acc += dict(be.getRenderpassTimings()).get("Temporal filtering", 0.0)
    print("%-50s TAA %.1f us" % (label, acc / 20 * 1000), flush=True)
    fp.destroy(); be.shutdown()
