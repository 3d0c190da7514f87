"""What the asynchronous tail (bloom chain + tonemap) costs the 4K frame: the benchmark frame with and without those passes recorded (diagnostic; the frame without them
is NOT the workload). python tools/tail_cost.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from plainrenderer_amd import RenderBackend
from plainrenderer_amd.frame import FramePipeline

class A: pass
args = A(); args.grid = 16; args.sdf_res = 64; args.shadow_res = 2048; args.steps = 40; args.warmup = 5; args.profile_frames = 0
w, h = 3840, 2160
for label, kw in (("full frame", {}), ("no bloom chain (apply + tonemap stay)", dict(run_bloom=0)), ("no bloom, no tonemap", dict(run_bloom=0, run_tonemap=0)), ("full frame", {})):
    be = RenderBackend(w, h, device=0)
    fp = FramePipeline(be, w, h, shadow_map_res=2048, **kw)
    scene, cams, inputs = bench.build_scene(args, "cuda:0", w, h)
    inputs.upload(fp)
    for i in range(30): fp.frame(cams[i + 1], 1 / 60, 0.5)
    be.waitForGPUIdle()
    t0 = time.perf_counter()
    n = 400
    for i in range(n): fp.frame(cams[(i % 40) + 6], 1 / 60, 0.5)
    be.waitForGPUIdle()
    print("%-40s %.4f ms per frame" % (label, (time.perf_counter() - t0) * 1e3 / n), flush=True)
    fp.destroy(); be.shutdown()
