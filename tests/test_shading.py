"""Deferred Cook-Torrance shade (re-expression of triangle.frag's lighting) and the BRDF LUT bake.
CPU: oracle KATs. GPU: HIP vs oracle; RGBA16F / R11G11B10 outputs are compared bit for bit."""
import numpy as np
import pytest

import passes
from plainrenderer_amd import pixfmt, synth
from plainrenderer_amd.scene import Camera, GlobalShaderInfo
from test_hiz_bloom_taa import packed_close
from util import light_buffer_bytes

W, H = 192, 108
LUT_RES = 32


def build_scene():
    class S:
        pass
    s = S()
    sc = synth.SynthScene(grid=4, cell=8.0, seed_id=400)
    cam = Camera.look((16.0, -6.0, -5.0), (0.05, 0.2, 1.0), aspect=W / H)
    s.gb = sc.gbuffer(cam, W, H)
    sun = np.array([0.35, -0.8, 0.45])
    sun /= np.linalg.norm(sun)
    s.shadow_info, s.shadow_maps = sc.shadow_cascades(cam, sun, 2.0, 60.0, 256)
    g = GlobalShaderInfo(frameIndex=3, sunDirection=(*sun.tolist(), 0.0))
    cam.fill_global(g, W, H)
    s.g = g
    s.noise = synth.blue_noise_standins()
    s.sky = synth.sky_lut()
    s.froxel, s.froxel_dims = synth.froxel_volume(W, H, 16)
    s.vol_settings = synth.volumetric_settings_bytes(30.0)
    r = np.random.default_rng(9)
    ysh = r.uniform(-0.2, 0.6, (H, W, 4)).astype(np.float32) * 0.02
    ysh[..., 0] = np.abs(ysh[..., 0]) + 0.01
    s.ysh = pixfmt.pack_half(ysh)
    s.cocg = pixfmt.pack_half(r.uniform(-0.004, 0.004, (H, W, 2)).astype(np.float32))
    s.light = light_buffer_bytes(sun_color=(1.0, 0.92, 0.8), prev_exposure=8e-5, sun_strength_exposed=128000 * 8e-5)
    s.lut = passes.orc_brdf_lut(LUT_RES, 2)
    return s


@pytest.fixture(scope="module")
def scene():
    return build_scene()


def test_kat_brdf_lut_properties(scene):
    lut = pixfmt.unpack_half(scene.lut).reshape(LUT_RES, LUT_RES, 4)
    assert np.isfinite(lut).all()
    # y = directional albedo at f0 = 1, x = its Fresnel-weighted part (used as mix(x, y, f0), triangle.frag:329): 0 <= x <= y <= 1
    assert lut[..., 0].min() >= 0 and lut[..., 1].max() <= 1.01 and (lut[..., 0] <= lut[..., 1] + 1e-3).all()
    assert 0.0 < lut[..., 2].min() and lut[..., 2].max() < 1.2
    assert np.all(lut[..., 3] == 0)
    # smooth surfaces looking straight on reflect ~ everything single-scattered: E(mu=1, r~0) ~ 1
    assert lut[-1, 1, 1] > 0.9


def _orc_bindless(noise, idx):
    n = max(idx) + 1
    import pyoracle as orc
    arr = (orc.OrcImage * n)()
    keep = []
    for nz, i in zip(noise, idx):
        im = orc.Img(np.ascontiguousarray(nz), 32, 32, passes.F.RG8)
        keep.append(im)
        arr[i] = im.c
    return arr, n, keep


def test_kat_shading_sky_and_night(scene):
    idx = [0, 1, 2, 3]
    scene.g.noiseTextureIndices = tuple(idx)
    arr, n, keep = _orc_bindless(scene.noise, idx)
    args = (scene.gb, W, H, scene.lut, LUT_RES)
    rest = (scene.shadow_info, scene.shadow_maps, 256, scene.ysh, scene.cocg, scene.froxel, scene.froxel_dims, scene.vol_settings, scene.sky, scene.g.pack(), arr, n)
    day = pixfmt.unpack_r11g11b10(passes.orc_deferred_shading(*args, scene.light, *rest)).reshape(H, W, 3)
    assert np.isfinite(day).all() and day.max() > 0
    night = pixfmt.unpack_r11g11b10(passes.orc_deferred_shading(*args, light_buffer_bytes(sun_strength_exposed=0.0), *rest)).reshape(H, W, 3)
    sky = scene.gb["depth"] == 0
    assert sky.any() and np.array_equal(day[sky], night[sky])           # sky stand-in does not depend on the sun strength
    assert night[~sky].sum() < day[~sky].sum()                           # direct sun light is gone, indirect + fog remain
    assert night[~sky].sum() > 0


@pytest.mark.gpu
@pytest.mark.parametrize("brdf", [0, 1, 2, 3])
def test_gpu_brdf_lut_bit_exact(backend, brdf):
    got, _ = passes.gpu_brdf_lut(backend, LUT_RES, brdf)
    ref = passes.orc_brdf_lut(LUT_RES, brdf)
    assert np.array_equal(pixfmt.unpack_half(got), pixfmt.unpack_half(ref))


@pytest.mark.gpu
@pytest.mark.parametrize("brdf,multi,aa,tech,cascades", [(2, 0, True, 0, 3), (0, 1, False, 0, 3), (1, 2, True, 1, 4), (3, 3, True, 0, 1), (2, 0, False, 1, 2)])
def test_gpu_deferred_shading_matches_oracle(backend, scene, brdf, multi, aa, tech, cascades):
    _, noise_idx = passes.make_bindless(backend, [], 1, scene.noise)
    scene.g.noiseTextureIndices = tuple(noise_idx)
    arr, n, keep = _orc_bindless(scene.noise, noise_idx)
    lut = passes.orc_brdf_lut(LUT_RES, brdf)
    args = (scene.gb, W, H, lut, LUT_RES, scene.light, scene.shadow_info, scene.shadow_maps, 256, scene.ysh, scene.cocg, scene.froxel, scene.froxel_dims,
            scene.vol_settings, scene.sky, scene.g.pack())
    got = passes.gpu_deferred_shading(backend, *args, brdf, multi, aa, tech, cascades)
    ref = passes.orc_deferred_shading(*args, arr, n, brdf, multi, aa, tech, cascades)
    assert packed_close(got, ref, max_fraction=0.0)
