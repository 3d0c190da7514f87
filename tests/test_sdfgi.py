"""Config 4: SDF diffuse GI (depth downscale, frustum + tile culling, sphere trace, spatial/temporal/spatial denoise, upscale).
CPU: oracle known-answer tests. GPU: every pass against the oracle on one small analytic scene (16 instances x 16^3 volumes,
256x144 G-buffer, half-res trace); half-float outputs are compared bit for bit (decoded, so -0 == +0)."""
import math
import struct

import numpy as np
import pytest

import passes
import pyoracle as orc
from plainrenderer_amd import pixfmt, synth
from plainrenderer_amd.scene import Camera, GlobalShaderInfo
from util import F, light_buffer_bytes

W, H = 256, 144
TW, TH = W // 2, H // 2
SDF_RES = 16
INFLUENCE = 5.0


class Scene:
    pass


@pytest.fixture(scope="module")
def scene():
    s = Scene()
    sc = synth.SynthScene(grid=4, cell=8.0, seed_id=300)
    cam = Camera.look((16.0, -7.0, -6.0), (0.0, 0.16, 1.0), aspect=W / H)
    cam_prev = Camera.look((16.05, -7.0, -6.1), (0.0, 0.16, 1.0), aspect=W / H)
    gb = sc.gbuffer(cam, W, H, cam_prev)
    s.sc, s.cam, s.gb = sc, cam, gb
    s.inst_bytes, s.bb_bytes, s.vols = sc.sdf_instances(SDF_RES)
    s.noise = synth.blue_noise_standins()
    s.sky = synth.sky_lut()
    sun = np.array([0.35, -0.8, 0.45])
    sun /= np.linalg.norm(sun)
    s.sun = sun
    s.shadow_info, s.shadow_maps = sc.shadow_cascades(cam, sun, 2.0, 60.0, 256)
    g = GlobalShaderInfo(frameIndex=6, sunDirection=(*sun.tolist(), 0.0), time=3.0)
    g.viewProjectionPrevious = cam_prev.view_projection()
    cam.fill_global(g, W, H)
    s.g = g
    s.light = light_buffer_bytes(sun_color=(1.0, 0.92, 0.8), prev_exposure=8e-5, sun_strength_exposed=128000 * 8e-5)
    s.fpts, s.fnrm = cam.frustum_points_normals()
    return s


def halves_equal(a, b):
    fa, fb = pixfmt.unpack_half(a), pixfmt.unpack_half(b)
    return bool(np.all((fa == fb) | (np.isnan(fa) & np.isnan(fb))))


def mismatch_fraction(a, b):
    fa, fb = pixfmt.unpack_half(a), pixfmt.unpack_half(b)
    return float(np.mean(~((fa == fb) | (np.isnan(fa) & np.isnan(fb)))))


def oracle_chain(s):
    """oracle results of every stage, cached on the scene"""
    if hasattr(s, "chain"):
        return s.chain
    c = {}
    depth = s.gb["depth"]
    c["half_depth"] = passes.orc_depth_downscale(depth, W, H)
    c["hiz"] = passes.orc_hiz(depth, W, H)
    gp = s.g.pack()
    c["culled"], c["tiles"] = passes.orc_sdf_culling(s.inst_bytes, s.bb_bytes, s.fpts, s.fnrm, INFLUENCE, c["hiz"][4], TW, TH, gp)
    s.chain = c
    return c


# ------------------------------------------------------------------ CPU KATs
def test_kat_scene_is_sane(scene):
    d = scene.gb["depth"]
    assert 0.02 < (d == 0).mean() < 0.6  # some sky, mostly geometry
    assert d.max() < 1.0
    c = oracle_chain(scene)
    assert 4 <= c["culled"][0] <= 16
    counts = c["tiles"].reshape(-1, passes.TILE_UINTS)[:, 0]
    assert counts.max() >= 1 and counts.max() <= 16


def test_kat_depth_downscale_picks_even_texels(scene):
    hd = pixfmt.unpack_half(oracle_chain(scene)["half_depth"]).reshape(TH, TW)
    expect = pixfmt.unpack_half(pixfmt.pack_half(scene.gb["depth"][::2, ::2]))
    assert np.array_equal(hd, expect)


# (the analytic-sphere trace KAT - hit distance = ray-sphere distance within the threshold, hit point on the sphere, radial normal, misses -
#  lives in tests/test_kat.py::test_kat_trace_hits_analytic_sphere_at_the_ray_sphere_distance)


# ------------------------------------------------------------------ GPU parity
@pytest.mark.gpu
def test_gpu_depth_downscale(backend, scene):
    got = passes.gpu_depth_downscale(backend, scene.gb["depth"], W, H)
    assert np.array_equal(got, oracle_chain(scene)["half_depth"])


@pytest.mark.gpu
@pytest.mark.parametrize("use_hiz", [True, False])
def test_gpu_culling_bit_exact(backend, scene, use_hiz):
    c = oracle_chain(scene)
    gp = scene.g.pack()
    _, pyramid, _ = passes.gpu_hiz(backend, scene.gb["depth"], W, H)
    culled_g, tiles_g, _ = passes.gpu_sdf_culling(backend, scene.inst_bytes, scene.bb_bytes, scene.fpts, scene.fnrm, INFLUENCE, pyramid if use_hiz else None, TW, TH, gp,
                                                  use_hiz)
    culled_o, tiles_o = passes.orc_sdf_culling(scene.inst_bytes, scene.bb_bytes, scene.fpts, scene.fnrm, INFLUENCE, c["hiz"][4] if use_hiz else None, TW, TH, gp, use_hiz)
    n = int(culled_o[0])
    assert culled_g[0] == n and np.array_equal(culled_g[1:1 + n], culled_o[1:1 + n])
    tg, to = tiles_g.reshape(-1, passes.TILE_UINTS), tiles_o.reshape(-1, passes.TILE_UINTS)
    assert np.array_equal(tg[:, 0], to[:, 0])
    for t in range(tg.shape[0]):
        k = int(to[t, 0])
        assert np.array_equal(tg[t, 1:1 + k], to[t, 1:1 + k])


@pytest.mark.gpu
def test_gpu_culling_caps_at_100_objects_per_tile(backend):
    # 300 identical instances in front of the camera: every tile keeps the first 100 in list order
    n = 300
    cam = Camera.look((0.0, -2.0, -10.0), (0.0, 0.0, 1.0), aspect=2.0)
    g = GlobalShaderInfo()
    cam.fill_global(g, 128, 64)
    inst_bytes = struct.pack("<4I", n, 0, 0, 0) + b"\0" * (96 * n)
    bb = struct.pack("<8f", -1, -3, 4, 0, 1, -1, 6, 0) * n
    fp, fn = cam.frustum_points_normals()
    culled_g, tiles_g, _ = passes.gpu_sdf_culling(backend, inst_bytes, bb, fp, fn, 5.0, None, 64, 32, g.pack(), False)
    culled_o, tiles_o = passes.orc_sdf_culling(inst_bytes, bb, fp, fn, 5.0, None, 64, 32, g.pack(), False)
    assert culled_g[0] == n == culled_o[0]
    assert np.array_equal(culled_g, culled_o)
    assert np.array_equal(tiles_g, tiles_o)
    assert tiles_g.reshape(-1, passes.TILE_UINTS)[0, 0] == 100


def _trace_inputs(backend, scene):
    c = oracle_chain(scene)
    vol_idx, noise_idx = passes.make_bindless(backend, scene.vols, SDF_RES, scene.noise)
    inst_bytes = passes.patch_instance_texture_indices(scene.inst_bytes, vol_idx)
    scene.g.noiseTextureIndices = tuple(noise_idx)
    arr, n, keep = passes.orc_bindless(scene.vols, SDF_RES, scene.noise, vol_idx, noise_idx)
    return c, inst_bytes, arr, n, keep


@pytest.mark.gpu
@pytest.mark.parametrize("strict,full_res", [(True, False), (False, False), (True, True)])
def test_gpu_trace_bit_exact(backend, scene, strict, full_res):
    c, inst_bytes, arr, n, keep = _trace_inputs(backend, scene)
    gp = scene.g.pack()
    tw, th = (W, H) if full_res else (TW, TH)
    if full_res:
        _, tiles = passes.orc_sdf_culling(scene.inst_bytes, scene.bb_bytes, scene.fpts, scene.fnrm, INFLUENCE, c["hiz"][4], tw, th, gp, True, screen_w=W)
    else:
        tiles = c["tiles"]
    args = (scene.gb["depth"], scene.gb["normal"], W, H, tw, th, scene.sky, 200, 100, scene.light, inst_bytes, tiles, INFLUENCE, scene.shadow_info,
            scene.shadow_maps[2], 256, gp)
    y_g, c_g = passes.gpu_sdf_trace(backend, *args, strict=strict, cascade=2)
    y_o, c_o = passes.orc_sdf_trace(*args, arr, n, strict=strict, cascade=2)
    assert mismatch_fraction(y_g, y_o) == 0.0 and mismatch_fraction(c_g, c_o) == 0.0
    y = pixfmt.unpack_half(y_o).reshape(th, tw, 4)
    assert np.isfinite(y).all()
    if not full_res and strict:
        scene.chain["trace"] = (y_o, c_o)


@pytest.mark.gpu
def test_gpu_denoise_chain_bit_exact(backend, scene):
    c, inst_bytes, arr, n, keep = _trace_inputs(backend, scene)
    gp = scene.g.pack()
    if "trace" not in c:
        c["trace"] = passes.orc_sdf_trace(scene.gb["depth"], scene.gb["normal"], W, H, TW, TH, scene.sky, 200, 100, scene.light, inst_bytes, c["tiles"], INFLUENCE,
                                          scene.shadow_info, scene.shadow_maps[2], 256, gp, arr, n, strict=True, cascade=2)
    y0, c0 = c["trace"]
    hd = c["half_depth"]
    # spatial filter on the input (filterIndex 0), half-res R16F depth
    sa = (y0, c0, TW, TH, hd, F.R16_sFloat, TW, TH, scene.gb["normal"], W, H, gp, 0)
    y1g, c1g = passes.gpu_gi_spatial(backend, *sa)
    y1o, c1o = passes.orc_gi_spatial(*sa)
    assert mismatch_fraction(y1g, y1o) == 0.0 and mismatch_fraction(c1g, c1o) == 0.0
    # temporal filter against a synthetic history
    r = np.random.default_rng(5)
    hy = pixfmt.pack_half(pixfmt.unpack_half(y1o) * r.uniform(0.7, 1.3, y1o.shape).astype(np.float32))
    hc = pixfmt.pack_half(pixfmt.unpack_half(c1o) * r.uniform(0.7, 1.3, c1o.shape).astype(np.float32))
    motion_last = np.roll(scene.gb["motion"], 3, axis=1)
    ta = (y1o, c1o, hy, hc, TW, TH, scene.gb["motion"], motion_last, W, H, gp)
    tg = passes.gpu_gi_temporal(backend, *ta)
    to = passes.orc_gi_temporal(*ta)
    for a, b in zip(tg, to):
        assert mismatch_fraction(a, b) == 0.0
    assert np.array_equal(to[0], to[2]) and np.array_equal(to[1], to[3])
    # spatial filter on the history (filterIndex 1)
    sb = (to[2], to[3], TW, TH, hd, F.R16_sFloat, TW, TH, scene.gb["normal"], W, H, gp, 1)
    y2g, c2g = passes.gpu_gi_spatial(backend, *sb)
    y2o, c2o = passes.orc_gi_spatial(*sb)
    assert mismatch_fraction(y2g, y2o) == 0.0 and mismatch_fraction(c2g, c2o) == 0.0
    # upscale to full resolution
    ua = (y2o, c2o, TW, TH, scene.gb["depth"], hd, W, H, gp)
    yug, cug = passes.gpu_gi_upscale(backend, *ua)
    yuo, cuo = passes.orc_gi_upscale(*ua)
    assert mismatch_fraction(yug, yuo) == 0.0 and mismatch_fraction(cug, cuo) == 0.0
    # full-res trace variant of the spatial filter reads the D32 depth buffer
    yf = np.repeat(np.repeat(y0.reshape(TH, TW, 4), 2, 0), 2, 1)
    cf = np.repeat(np.repeat(c0.reshape(TH, TW, 2), 2, 0), 2, 1)
    sc_ = (yf, cf, W, H, scene.gb["depth"], F.Depth32, W, H, scene.gb["normal"], W, H, gp, 0)
    a = passes.gpu_gi_spatial(backend, *sc_)
    b = passes.orc_gi_spatial(*sc_)
    assert mismatch_fraction(a[0], b[0]) == 0.0 and mismatch_fraction(a[1], b[1]) == 0.0


# ------------------------------------------------------------------ SDF debug visualisation
@pytest.mark.gpu
@pytest.mark.parametrize("mode", [1, 2, 3, 4, 0])
def test_gpu_sdf_debug_visualisation_bit_exact(backend, scene, mode):
    c, inst_bytes, arr, n, keep = _trace_inputs(backend, scene)
    gp = scene.g.pack()
    # the debug view culls at full resolution with influence radius 0 (SDFGI.cpp:337-350); HiZ only for the tile-usage mode
    use_hiz = mode == 2
    _, tiles = passes.orc_sdf_culling(scene.inst_bytes, scene.bb_bytes, scene.fpts, scene.fnrm, 0.0, c["hiz"][4], W, H, gp, use_hiz, screen_w=W)
    args = (W, H, scene.sky, 200, 100, scene.light, inst_bytes, tiles, scene.shadow_info, scene.shadow_maps[2], 256, gp)
    a = passes.gpu_sdf_debug(backend, *args, mode, cascade=2)
    b = passes.orc_sdf_debug(*args, arr, n, mode, cascade=2)
    assert np.array_equal(a, b)
    img = pixfmt.unpack_r11g11b10(b.reshape(-1)).reshape(H, W, 3)
    assert np.isfinite(img).all()
    if mode == 3:
        # normals * 0.5 + 0.5 of unit vectors: hit pixels have |2c - 1| ~ 1 (up to the 6/5-bit mantissas)
        hit = np.abs(np.linalg.norm(img * 2 - 1, axis=-1) - 1) < 0.08
        assert 0.05 < hit.mean() < 0.95
    if mode == 2:
        assert (img[..., 0] == img[..., 1]).mean() > 0.9 and img.max() <= 1.0  # grey = fill ratio of the tile lists
    if mode == 4:
        assert img.max() <= 1.0 and (img > 0).any()


@pytest.mark.gpu
def test_gpu_frame_sdf_debug_view(backend):
    """SDFDebugSettings::visualisationMode != None replaces the frame by the debug view (RenderFrontend.cpp:321-340): exposure from
    the previous debug image, full-resolution culling, visualisation into postProcessBuffers[0], tonemap"""
    from plainrenderer_amd.frame import FramePipeline, SyntheticInputs
    from plainrenderer_amd.scene import Camera
    w, h = 256, 144
    cams = [Camera.look((15.0 + 0.03 * i, -7.0, -6.0 + 0.05 * i), (0.0, 0.16, 1.0), aspect=w / h) for i in range(3)]
    sc = synth.SynthScene(grid=4, cell=8.0, seed_id=960)
    fp = FramePipeline(backend, w, h, shadow_map_res=128, brdf_lut_res=16, froxel_depth=8, max_sdf_instances=64, sdf_debug_mode=1)
    inputs = SyntheticInputs(sc, cams[1], cams[0], w, h, sdf_res=16, shadow_res=128, froxel_depth=8, sun_direction=(0.35, -0.8, 0.45))
    inputs.upload(fp)
    for f in range(2):
        fp.frame(cams[f + 1], 1 / 60.0, 0.5 + f / 60.0)
    be = backend
    gp = bytes(fp.submitted_globals())
    g = orc_global = None
    light = be.downloadStorageBuffer(fp.storage_buffer("light"), 20).tobytes()
    tiles = be.downloadStorageBuffer(fp.storage_buffer("sdfCulledTiles"), 404 * math.ceil(w / 32) * math.ceil(h / 32), dtype=np.uint32)
    arr, n, keep = passes.orc_bindless(inputs.volumes, inputs.sdf_res, inputs.noise, list(inputs.volume_indices),
                                       [int(x) for x in np.frombuffer(gp[240:256], np.int32)])
    exp = passes.orc_sdf_debug(w, h, inputs.sky, 200, 100, light, inputs.instance_bytes_patched, tiles, inputs.shadow_info, inputs.shadow_maps[2], 128, gp, arr, n, 1, cascade=2)
    got = be.downloadImage(fp.image("post0"), 0, np.uint32).reshape(h, w)
    assert np.array_equal(got, exp)
    sw = be.downloadImage(fp.image("swapchain"), 0, np.uint8).reshape(h, w, 4).astype(int)
    ref = passes.orc_tonemap(exp.reshape(-1), w, h, gp).astype(int)
    assert np.abs(sw - ref).max() <= 1
    lit = pixfmt.unpack_r11g11b10(exp.reshape(-1))
    assert np.isfinite(lit).all() and lit.max() > 0
    fp.destroy()


# ------------------------------------------------------------------ tiles at the 100-instance cap (sdfCulling.inc:7-15)
@pytest.fixture(scope="module")
def dense_scene():
    """256 instances packed 1 m apart: the camera looks along the field, 10 of the 24 culling tiles hold the maximum of 100 instances"""
    s = Scene()
    sc = synth.SynthScene(grid=16, cell=1.0, seed_id=301)
    cam = Camera.look((8.0, -5.0, -6.0), (0.0, 0.35, 1.0), aspect=W / H)
    s.sc, s.cam, s.gb = sc, cam, sc.gbuffer(cam, W, H, cam)
    s.inst_bytes, s.bb_bytes, s.vols = sc.sdf_instances(SDF_RES)
    s.noise = synth.blue_noise_standins()
    s.sky = synth.sky_lut()
    sun = np.array([0.35, -0.8, 0.45])
    sun /= np.linalg.norm(sun)
    s.shadow_info, s.shadow_maps = sc.shadow_cascades(cam, sun, 2.0, 60.0, 256)
    g = GlobalShaderInfo(frameIndex=6, sunDirection=(*sun.tolist(), 0.0), time=3.0)
    g.viewProjectionPrevious = cam.view_projection()
    cam.fill_global(g, W, H)
    s.g = g
    s.light = light_buffer_bytes(sun_color=(1.0, 0.92, 0.8), prev_exposure=8e-5, sun_strength_exposed=128000 * 8e-5)
    s.fpts, s.fnrm = cam.frustum_points_normals()
    return s


def test_kat_dense_scene_reaches_the_tile_cap(dense_scene):
    c = oracle_chain(dense_scene)
    counts = c["tiles"].reshape(-1, passes.TILE_UINTS)[:, 0]
    assert c["culled"][0] == 256 and counts.max() == 100 and (counts == 100).sum() >= 8


@pytest.mark.gpu
@pytest.mark.parametrize("fast_set", [False, True])
def test_gpu_culling_and_trace_with_tiles_at_the_cap(backend, dense_scene, fast_set):
    """tiles holding exactly the 100 instances the list can store: the culling writes the same lists as the oracle, the exact trace the same bits, and
    the fast trace (which stages a tile's whole instance table in LDS) agrees within the fast set's tolerance"""
    s = dense_scene
    c, inst_bytes, arr, n, keep = _trace_inputs(backend, s)
    gp = s.g.pack()
    backend.setMathMode(fast_set)
    try:
        _, pyramid, _ = passes.gpu_hiz(backend, s.gb["depth"], W, H)
        culled_g, tiles_g, _ = passes.gpu_sdf_culling(backend, s.inst_bytes, s.bb_bytes, s.fpts, s.fnrm, INFLUENCE, pyramid, TW, TH, gp)
        args = (s.gb["depth"], s.gb["normal"], W, H, TW, TH, s.sky, 200, 100, s.light, inst_bytes, c["tiles"], INFLUENCE, s.shadow_info, s.shadow_maps[2], 256, gp)
        y_g, c_g = passes.gpu_sdf_trace(backend, *args, strict=True, cascade=2)
    finally:
        backend.setMathMode(False)
    got, ref = tiles_g.reshape(-1, passes.TILE_UINTS), c["tiles"].reshape(-1, passes.TILE_UINTS)
    assert np.array_equal(got[:, 0], ref[:, 0])
    for t in range(ref.shape[0]):
        assert np.array_equal(got[t, 1:1 + ref[t, 0]], ref[t, 1:1 + ref[t, 0]]), "instance list of tile %d" % t
    y_o, c_o = passes.orc_sdf_trace(*args, arr, n, strict=True, cascade=2)
    if not fast_set:
        assert mismatch_fraction(y_g, y_o) == 0.0 and mismatch_fraction(c_g, c_o) == 0.0
    else:
        fy, fo = pixfmt.unpack_half(y_g).astype(np.float64), pixfmt.unpack_half(y_o).astype(np.float64)
        assert np.isfinite(fy).all()
        bad = np.abs(fy - fo) > np.maximum(2.0 ** -7 * np.abs(fo), 2.0 ** -10 * np.abs(fo).max())
        print("PARITY trace_at_cap pixels_outside_tolerance=%g" % bad.reshape(TH * TW, -1).any(axis=1).mean())
        assert bad.reshape(TH * TW, -1).any(axis=1).mean() <= 0.01  # rays that resolve differently and their 3x3 neighbours
