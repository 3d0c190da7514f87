"""culling + SDF trace at the CAP of the culling lists, at the benchmark's size. The scene of `bench.py --scene dense` (256 instances x 64^3 one metre
apart, camera along the field: most culling tiles hold the maximum of 100 instances, sdfCameraTileCulling.comp:42-99) at 3840 x 2160, half-resolution trace:
  * frustum + tile culling (fast set) against the oracle: the same lists, entry for entry;
  * sdfDiffuseTrace (fast set, the benchmarked kernel) against the oracle with decision signatures, the statement of tests/test_parity_fullsize.py::test_gpu_fullsize_trace.
    python tools/parity_dense.py        (one MI355X; the oracle trace runs on the host cores)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle")):
    sys.path.insert(0, p)
import numpy as np
import passes, parity
import pyoracle as orc
from plainrenderer_amd import RenderBackend, pixfmt, synth
from plainrenderer_amd.scene import Camera, GlobalShaderInfo
from util import light_buffer_bytes

W, H = (int(v) for v in os.environ.get("PLR_PARITY_DENSE_SIZE", "3840x2160").split("x"))
TW, TH = W // 2, H // 2
SDF_RES, INFLUENCE = 64, 5.0
U = lambda a: pixfmt.unpack_half(a).astype(np.float64)
orc.set_threads(os.cpu_count() or 1)
sc = synth.SynthScene(grid=16, cell=1.0, seed_id=301)
cam = Camera.look((8.0, -5.0, -6.0), (0.0, 0.35, 1.0), aspect=W / H)
gb = sc.gbuffer(cam, W, H, cam)
inst_bytes0, bb_bytes, vols = sc.sdf_instances(SDF_RES)
noise, sky = synth.blue_noise_standins(), synth.sky_lut()
sun = np.array([0.35, -0.8, 0.45]); sun /= np.linalg.norm(sun)
shadow_info, shadow_maps = sc.shadow_cascades(cam, sun, 2.0, 60.0, 2048)
g = GlobalShaderInfo(frameIndex=6, sunDirection=(*sun.tolist(), 0.0), time=3.0)
g.viewProjectionPrevious = cam.view_projection()
cam.fill_global(g, W, H)
light = light_buffer_bytes(sun_color=(1.0, 0.92, 0.8), prev_exposure=8e-5, sun_strength_exposed=128000 * 8e-5)
fpts, fnrm = cam.frustum_points_normals()
be = RenderBackend(W, H, device=0)
be.setMathMode(True)
vol_idx, noise_idx = passes.make_bindless(be, vols, SDF_RES, noise)
inst_bytes = passes.patch_instance_texture_indices(inst_bytes0, vol_idx)
g.noiseTextureIndices = tuple(noise_idx)
arr, n, keep = passes.orc_bindless(vols, SDF_RES, noise, vol_idx, noise_idx)
gp = g.pack()
# ---- culling
hiz = passes.orc_hiz(gb["depth"], W, H)
culled_o, tiles_o = passes.orc_sdf_culling(inst_bytes0, bb_bytes, fpts, fnrm, INFLUENCE, hiz[4], TW, TH, gp)
_, pyramid, _ = passes.gpu_hiz(be, gb["depth"], W, H)
culled_g, tiles_g, _ = passes.gpu_sdf_culling(be, inst_bytes0, bb_bytes, fpts, fnrm, INFLUENCE, pyramid, TW, TH, gp)
got, ref = tiles_g.reshape(-1, passes.TILE_UINTS), tiles_o.reshape(-1, passes.TILE_UINTS)
counts = ref[:, 0]
import math
used = counts.reshape(-1, math.ceil(W / 32))[:math.ceil(TH / 32), :math.ceil(TW / 32)].reshape(-1)  # the buffer's row stride is the FULL-resolution tile count (sdfCulling.inc:17-20)
lists_equal = bool(np.array_equal(got[:, 0], counts) and all(np.array_equal(got[t, 1:1 + counts[t]], ref[t, 1:1 + counts[t]]) for t in range(ref.shape[0])))
print("PARITY dense_culling  %dx%d trace %dx%d: frustum-culled instances gpu %d oracle %d; tile lists identical %s; instances per tile median %d mean %.1f max %d, tiles at the cap of 100: %d of %d" % (
    W, H, TW, TH, int(culled_g[0]), int(culled_o[0]), lists_equal, int(np.median(used)), used.mean(), used.max(), int((used >= 100).sum()), used.size), flush=True)
assert lists_equal and int(culled_g[0]) == int(culled_o[0])
# ---- trace
args = (gb["depth"], gb["normal"], W, H, TW, TH, sky, 200, 100, light, inst_bytes, tiles_o, INFLUENCE, shadow_info, shadow_maps[2], 2048, gp)
with passes.gpu_signature(be, TW * TH) as sg:
    yg, cg = passes.gpu_sdf_trace(be, *args, strict=True, cascade=2)
t0 = time.perf_counter()
with passes.orc_signature(TW * TH) as so:
    yo, co = passes.orc_sdf_trace(*args, arr, n, strict=True, cascade=2)
t_oracle = time.perf_counter() - t0
ray_flip = ((sg.words ^ so.words) & ~np.uint32(0x7F8)).reshape(TH, TW) != 0
take_flip = ((sg.words ^ so.words) & np.uint32(0x7F8)).reshape(TH, TW) != 0
touched = (parity.dilate3x3(ray_flip) | take_flip).reshape(-1)
got = np.concatenate([U(yg).reshape(-1, 4), U(cg).reshape(-1, 2)], axis=1)
ref = np.concatenate([U(yo).reshape(-1, 4), U(co).reshape(-1, 2)], axis=1)
bad = parity.half_violations(got, ref, floor_frac=2.0 ** -10)
print("PARITY dense_trace    rays_flipped=%.6g take_flipped=%.6g pixels_touched=%.6g clean_violations=%d touched_violations=%.6g (oracle trace: %.1f s on %d host threads)" % (
    ray_flip.mean(), take_flip.mean(), touched.mean(), int((bad & ~touched).sum()), float((bad & touched).mean()), t_oracle, os.cpu_count() or 1), flush=True)
assert not (bad & ~touched).any(), "pixels with identical ray decisions must agree to max(2^-7 |x|, 2^-10 max|x|)"
assert ray_flip.mean() <= 1e-4 and take_flip.mean() <= 1e-4 and np.isfinite(got).all()
be.shutdown()
print("parity_dense ok")
