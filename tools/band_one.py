"""One partition of the 8K frame alone, for a kernel trace: python tools/band_one.py <index> [--tiles GXxGY] [--overlap M] [--frames K]
(rocprofv3 --kernel-trace --stats -- python tools/band_one.py 2 --overlap 0: the kernels' own durations, without the pass-timing events of band_cost.py --passes)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from plainrenderer_amd import RenderBackend, tiling
from plainrenderer_amd.frame import FramePipeline

def opt(name, default):
    return sys.argv[sys.argv.index(name) + 1] if name in sys.argv else default
index = int(sys.argv[1])
gx, gy = (int(v) for v in opt("--tiles", "1x4").lower().split("x"))
n = gx * gy
frames = int(opt("--frames", "100"))
class A: pass
args = A(); args.grid = 16; args.sdf_res = 64; args.shadow_res = 2048; args.steps = frames; args.warmup = 5; args.profile_frames = 0
w, h = 7680, 1080 * n
rects = tiling.tile_rects(w, h, gx, gy)
x0, y0, x1, y1 = rects[index]
kw = dict(band_row_begin=y0, band_row_end=y1)
if x0 != 0 or x1 != w:
    kw.update(band_col_begin=x0, band_col_end=x1)
if "--overlap" in sys.argv:
    kw.update(band_overlap_exchange=int(opt("--overlap", "2")))
be = RenderBackend(w, h, device=0)
fp = FramePipeline(be, w, h, shadow_map_res=2048, **kw)
fp.attach_rccl_rects(None, index, n, w, h, rects)
scene, cams, inputs = bench.build_scene(args, "cuda:0", w, h, (y0, y1))
inputs.upload(fp)
for i in range(frames + 5):
    fp.frame(cams[i + 1], 1 / 60, 0.5)
be.waitForGPUIdle()
fp.destroy(); be.shutdown()
