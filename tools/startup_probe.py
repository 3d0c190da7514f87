"""Where do the first timed frames of `bench.py --steps 20 --warmup 5` lose their time? The benchmark frame, bench.py's sequence (auxiliary frames, W warm-up frames, a sync),
then K frames with a timestamp event on the launch stream after each: per-frame GPU time from the sync on.   python tools/startup_probe.py [K] [W]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import ctypes as C
import torch
import bench
from plainrenderer_amd import RenderBackend
from plainrenderer_amd.frame import FramePipeline

K = int(sys.argv[1]) if len(sys.argv) > 1 else 25
W = int(sys.argv[2]) if len(sys.argv) > 2 else 5
class A: pass
args = A(); args.grid = 16; args.sdf_res = 64; args.shadow_res = 2048; args.steps = 3 * (45 + K + W); args.warmup = W; args.profile_frames = 0
w, h = 3840, 2160
be = RenderBackend(w, h, device=0)
fp = FramePipeline(be, w, h, shadow_map_res=2048)
scene, cams, inputs = bench.build_scene(args, "cuda:0", w, h, None)
inputs.upload(fp)
p = C.c_void_p()
be._check(be.lib.plr_get_launch_stream(C.byref(p)))
stream = torch.cuda.ExternalStream(p.value)
n = [0]
def frame():
    fp.frame(cams[n[0] + 1], 1 / 60, 0.5); n[0] += 1
for rep in range(3):
    for _ in range(40):
        frame()
    for _ in range(W):
        frame()
    be.waitForGPUIdle(); torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(K + 1)]
    t0 = time.perf_counter()
    ev[0].record(stream)
    host = []
    for i in range(K):
        frame()
        ev[i + 1].record(stream)
        host.append((time.perf_counter() - t0) * 1e3)
    be.waitForGPUIdle(); torch.cuda.synchronize()
    total = (time.perf_counter() - t0) * 1e3
    gpu = [ev[i].elapsed_time(ev[i + 1]) for i in range(K)]
    print("rep %d: wall %.3f ms for %d frames = %.4f per frame; launch-stream time per frame (ms): %s" % (rep, total, K, total / K, " ".join("%.3f" % g for g in gpu)))
    print("        host time at which frame i was recorded (ms): %s" % " ".join("%.2f" % t for t in host))
    print("        launch stream reached the last event at %.3f ms; the rest of the wall time is the asynchronous tail of the last frame + the sync" % sum(gpu))
fp.destroy(); be.shutdown()
