"""PLR_MATH_FAST kernels (restructured arithmetic, FMA, v_rcp/v_rsq/v_exp/v_log, LDS tiling) against the oracle.

These kernels are not bit-identical to the scalar evaluation by design; the stated tolerance per pass is:
  |got - ref| <= max(REL * |ref|, ABS_FLOOR_FRAC * max|ref|) for all but OUTLIER_FRAC of the values (a last-bit difference can
  select a neighbouring texel / flip an on-off-screen or hit test for isolated pixels), the mean absolute error stays below
  MEAN_REL * mean|ref|, and no value deviates by more than OUTLIER_MAX_FRAC * max|ref|.
"""
import numpy as np
import pytest

import passes
from plainrenderer_amd import pixfmt
from test_sdfgi import INFLUENCE, SDF_RES, TH, TW, H, W, F, _trace_inputs, oracle_chain, scene  # noqa: F401 (scene fixture)


def assert_close(got, ref, rel, abs_floor_frac=1e-4, outlier_frac=0.0, outlier_max_frac=0.0, mean_rel=1e-3, what=""):
    got = np.asarray(got, np.float64).reshape(-1)
    ref = np.asarray(ref, np.float64).reshape(-1)
    assert np.isfinite(got).all(), what + ": non-finite values"
    scale = np.abs(ref).max()
    tol = np.maximum(rel * np.abs(ref), abs_floor_frac * scale)
    err = np.abs(got - ref)
    bad = err > tol
    assert bad.mean() <= outlier_frac, "%s: %.4f%% of values outside tolerance (allowed %.4f%%), worst %.3g of scale %.3g" % (
        what, 100 * bad.mean(), 100 * outlier_frac, err.max(), scale)
    if bad.any():
        assert err.max() <= outlier_max_frac * scale, "%s: outlier error %.3g exceeds %.3g" % (what, err.max(), outlier_max_frac * scale)
    assert err.mean() <= mean_rel * np.abs(ref).mean() + 1e-12, "%s: mean error %.3g vs mean value %.3g" % (what, err.mean(), np.abs(ref).mean())


_SHADE_SCENE = None


@pytest.fixture()
def fast(backend):
    backend.setMathMode(True)
    yield backend
    backend.setMathMode(False)


@pytest.mark.gpu
@pytest.mark.parametrize("filter_index", [0, 1])
def test_gpu_fast_spatial_filter(fast, scene, filter_index):
    c, inst_bytes, arr, n, keep = _trace_inputs(fast, scene)
    gp = scene.g.pack()
    if "trace" not in c:
        c["trace"] = passes.orc_sdf_trace(scene.gb["depth"], scene.gb["normal"], W, H, TW, TH, scene.sky, 200, 100, scene.light, inst_bytes, c["tiles"], INFLUENCE,
                                          scene.shadow_info, scene.shadow_maps[2], 256, gp, arr, n, strict=True, cascade=2)
    y0, c0 = c["trace"]
    args = (y0, c0, TW, TH, c["half_depth"], F.R16_sFloat, TW, TH, scene.gb["normal"], W, H, gp, filter_index)
    yg, cg = passes.gpu_gi_spatial(fast, *args)
    yo, co = passes.orc_gi_spatial(*args)
    # half-float outputs: 2^-10 relative is one ulp; isolated pixels may pick a neighbouring texel for one of their 32 samples
    assert_close(pixfmt.unpack_half(yg), pixfmt.unpack_half(yo), rel=2.0 ** -9, outlier_frac=0.02, outlier_max_frac=0.2, mean_rel=2e-3, what="Y_SH")
    assert_close(pixfmt.unpack_half(cg), pixfmt.unpack_half(co), rel=2.0 ** -9, outlier_frac=0.02, outlier_max_frac=0.2, mean_rel=2e-3, what="CoCg")
    # full-res (D32 depth) variant
    yf = np.repeat(np.repeat(y0.reshape(TH, TW, 4), 2, 0), 2, 1)
    cf = np.repeat(np.repeat(c0.reshape(TH, TW, 2), 2, 0), 2, 1)
    args = (yf, cf, W, H, scene.gb["depth"], F.Depth32, W, H, scene.gb["normal"], W, H, gp, filter_index)
    yg, cg = passes.gpu_gi_spatial(fast, *args)
    yo, co = passes.orc_gi_spatial(*args)
    assert_close(pixfmt.unpack_half(yg), pixfmt.unpack_half(yo), rel=2.0 ** -9, outlier_frac=0.02, outlier_max_frac=0.2, mean_rel=2e-3, what="Y_SH full-res")


# ------------------------------------------------------------------ deferred shading
@pytest.mark.gpu
@pytest.mark.parametrize("brdf,multi,aa,tech,cascades", [(2, 0, True, 0, 3), (0, 1, False, 0, 3), (1, 2, True, 1, 4), (3, 3, True, 0, 1)])
def test_gpu_fast_deferred_shading(fast, brdf, multi, aa, tech, cascades):
    import test_shading as ts
    global _SHADE_SCENE
    if _SHADE_SCENE is None:
        _SHADE_SCENE = ts.build_scene()
    sc = _SHADE_SCENE
    _, noise_idx = passes.make_bindless(fast, [], 1, sc.noise)
    sc.g.noiseTextureIndices = tuple(noise_idx)
    arr, n, keep = ts._orc_bindless(sc.noise, noise_idx)
    lut = passes.orc_brdf_lut(ts.LUT_RES, brdf)
    args = (sc.gb, ts.W, ts.H, lut, ts.LUT_RES, sc.light, sc.shadow_info, sc.shadow_maps, 256, sc.ysh, sc.cocg, sc.froxel, sc.froxel_dims, sc.vol_settings, sc.sky,
            sc.g.pack())
    got = pixfmt.unpack_r11g11b10(passes.gpu_deferred_shading(fast, *args, brdf, multi, aa, tech, cascades))
    ref = pixfmt.unpack_r11g11b10(passes.orc_deferred_shading(*args, arr, n, brdf, multi, aa, tech, cascades))
    # R11G11B10: one quantum is 2^-6 (R,G) / 2^-5 (B) relative. Outliers: a PCF tap on the edge of its depth test (1/12 of the sun term)
    assert_close(got, ref, rel=2.0 ** -5, abs_floor_frac=2e-4, outlier_frac=0.01, outlier_max_frac=0.15, mean_rel=2e-3, what="shaded colour")


# ------------------------------------------------------------------ TAA
@pytest.mark.gpu
@pytest.mark.parametrize("clip,dilate,tech,tonemap", [(True, True, 4, True), (False, True, 0, True), (True, False, 1, False), (True, True, 2, True),
                                                      (False, False, 3, True), (True, True, 4, False)])
def test_gpu_fast_taa(fast, clip, dilate, tech, tonemap):
    from test_hiz_bloom_taa import _global, synth_depth, synth_motion
    from plainrenderer_amd.scene import taa_jitter_pixels, taa_resolve_weights
    from util import hdr_image
    w, h = 320, 180
    cur = hdr_image(w, h, 40)
    # a plausible history: the current frame shifted by a pixel and re-quantised (pure noise history makes every pixel a contrast edge)
    hist = np.roll(cur.reshape(h, w), 1, axis=1).copy()
    motion, depth = synth_motion(w, h, 42), synth_depth(w, h, 43)
    wts = taa_resolve_weights(taa_jitter_pixels(5))
    g = _global(w, h)
    og, hg = passes.gpu_taa(fast, cur, hist, motion, depth, w, h, wts, g, clip, dilate, tech, tonemap)
    oo, ho = passes.orc_taa(cur, hist, motion, depth, w, h, wts, g, clip, dilate, tech, tonemap)
    assert np.array_equal(og, hg)
    assert_close(pixfmt.unpack_r11g11b10(og), pixfmt.unpack_r11g11b10(oo), rel=2.0 ** -5, abs_floor_frac=2e-4, outlier_frac=0.005, outlier_max_frac=0.1, mean_rel=2e-3,
                 what="TAA output")


# ------------------------------------------------------------------ bloom
@pytest.mark.gpu
@pytest.mark.parametrize("w,h", [(1920, 1080), (270, 136), (97, 61)])
def test_gpu_fast_bloom(fast, w, h):
    from util import hdr_image
    scene_img = hdr_image(w, h, buffer_id=31)
    out_g, downs_g, ups_g = passes.gpu_bloom(fast, scene_img, w, h)
    out_o, downs_o, ups_o = passes.orc_bloom(scene_img, w, h)
    for i, (a, b) in enumerate(zip(ups_g, ups_o)):
        # each level re-quantises to R11G11B10; a one-quantum flip at a coarse level is carried (attenuated) into the finer ones
        assert_close(pixfmt.unpack_r11g11b10(a), pixfmt.unpack_r11g11b10(b), rel=2.0 ** -5, abs_floor_frac=1e-4, outlier_frac=0.0, mean_rel=2e-3, what="up mip %d" % i)
    assert_close(pixfmt.unpack_r11g11b10(out_g), pixfmt.unpack_r11g11b10(out_o), rel=2.0 ** -5, abs_floor_frac=1e-4, outlier_frac=0.0, mean_rel=2e-3, what="applied")


@pytest.mark.gpu
def test_gpu_fast_bloom_constant_energy(fast):
    w, h = 3840, 2160
    scene_img = pixfmt.pack_r11g11b10(np.full((h, w, 3), 0.5, np.float32))
    out, downs, ups = passes.gpu_bloom(fast, scene_img, w, h, strength=0.25)
    assert np.all(pixfmt.unpack_r11g11b10(ups[0]) == 2.5)
    assert np.all(pixfmt.unpack_r11g11b10(out) == 0.5 * (1 + 4 * 0.25))


# ------------------------------------------------------------------ SDF trace
@pytest.mark.gpu
@pytest.mark.parametrize("strict", [True, False])
def test_gpu_fast_trace(fast, scene, strict):
    c, inst_bytes, arr, n, keep = _trace_inputs(fast, scene)
    gp = scene.g.pack()
    args = (scene.gb["depth"], scene.gb["normal"], W, H, TW, TH, scene.sky, 200, 100, scene.light, inst_bytes, c["tiles"], INFLUENCE, scene.shadow_info,
            scene.shadow_maps[2], 256, gp)
    y_g, c_g = passes.gpu_sdf_trace(fast, *args, strict=strict, cascade=2)
    y_o, c_o = passes.orc_sdf_trace(*args, arr, n, strict=strict, cascade=2)
    # a ray on the edge of the hit threshold / an AABB face may resolve differently; it then changes its own texel and the (up to 8)
    # neighbours sharing it through the 3x3 resolve. Everything else agrees to half-float precision.
    assert_close(pixfmt.unpack_half(y_g), pixfmt.unpack_half(y_o), rel=2.0 ** -8, abs_floor_frac=2e-4, outlier_frac=0.03, outlier_max_frac=1.0, mean_rel=2e-2, what="trace Y_SH")
    assert_close(pixfmt.unpack_half(c_g), pixfmt.unpack_half(c_o), rel=2.0 ** -8, abs_floor_frac=2e-4, outlier_frac=0.03, outlier_max_frac=2.0, mean_rel=2e-2, what="trace CoCg")


# ------------------------------------------------------------------ streaming passes (kernels_fast/stream_fast.hip)
@pytest.mark.gpu
def test_gpu_fast_temporal_gi_and_upscale(fast, scene):
    c, inst_bytes, arr, n, keep = _trace_inputs(fast, scene)
    gp = scene.g.pack()
    if "trace" not in c:
        c["trace"] = passes.orc_sdf_trace(scene.gb["depth"], scene.gb["normal"], W, H, TW, TH, scene.sky, 200, 100, scene.light, inst_bytes, c["tiles"], INFLUENCE,
                                          scene.shadow_info, scene.shadow_maps[2], 256, gp, arr, n, strict=True, cascade=2)
    y0, c0 = c["trace"]
    r = np.random.default_rng(5)
    hy = pixfmt.pack_half(pixfmt.unpack_half(y0) * r.uniform(0.7, 1.3, y0.shape).astype(np.float32))
    hc = pixfmt.pack_half(pixfmt.unpack_half(c0) * r.uniform(0.7, 1.3, c0.shape).astype(np.float32))
    motion_last = np.roll(scene.gb["motion"], 3, axis=1)
    ta = (y0, c0, hy, hc, TW, TH, scene.gb["motion"], motion_last, W, H, gp)
    tg = passes.gpu_gi_temporal(fast, *ta)
    to = passes.orc_gi_temporal(*ta)
    # half-float outputs; alpha switches between branches at thresholds (3 px motion, off-screen), where a rounding difference
    # in the reprojected coordinate changes a pixel visibly
    for a, b, what in zip(tg, to, ("Y_SH", "CoCg", "history Y_SH", "history CoCg")):
        assert_close(pixfmt.unpack_half(a), pixfmt.unpack_half(b), rel=2.0 ** -9, outlier_frac=0.005, outlier_max_frac=0.5, mean_rel=3e-3, what="temporal " + what)
    assert np.array_equal(tg[0], tg[2]) and np.array_equal(tg[1], tg[3])
    ua = (to[2], to[3], TW, TH, scene.gb["depth"], c["half_depth"], W, H, gp)
    yug, cug = passes.gpu_gi_upscale(fast, *ua)
    yuo, cuo = passes.orc_gi_upscale(*ua)
    # the edge test (|depth difference| > 0.5 m) and the closest-texel choice flip for a few pixels on a depth discontinuity
    assert_close(pixfmt.unpack_half(yug), pixfmt.unpack_half(yuo), rel=2.0 ** -9, outlier_frac=0.005, outlier_max_frac=1.0, mean_rel=3e-3, what="upscale Y_SH")
    assert_close(pixfmt.unpack_half(cug), pixfmt.unpack_half(cuo), rel=2.0 ** -9, outlier_frac=0.005, outlier_max_frac=2.0, mean_rel=3e-3, what="upscale CoCg")


@pytest.mark.gpu
@pytest.mark.parametrize("w,h", [(1920, 1080), (97, 61)])
def test_gpu_fast_tonemap_within_one_lsb(fast, w, h):
    from util import hdr_image
    from test_exposure_tonemap import _global
    img = hdr_image(w, h, buffer_id=4, pre_exposure=1e-3)
    g = _global(w, h, time=37.25)
    a = passes.gpu_tonemap(fast, img, w, h, g, F.BGRA8_uNorm).astype(int)
    b = passes.orc_tonemap(img, w, h, g).astype(int)
    d = np.abs(a - b)
    assert d.max() <= 1  # stated tolerance: +-1/255
    assert (d != 0).mean() < 0.03
    assert np.all(a[..., 3] == 255)


@pytest.mark.gpu
def test_fast_r11g11b10_encoder_against_the_exact_one_for_every_float(backend):
    """device/image.h packR11G11B10Fast (every colour write of the PLR_MATH_FAST kernels) against the exact encoder, per channel width, over all
    2^32 float bit patterns: negative values, infinities, NaNs and overflow take the exact encoder; in range the single integer rounding gives the
    same code except where the 2^-112 scaling itself rounds (results below 2^-14) and lands on a tie: one code, a handful of patterns per binade"""
    diff11, diff10, max_diff, last = backend.debugVerifyR11G11B10Fast()
    print("PARITY r11g11b10_fast differing_11bit=%d differing_10bit=%d max_code_diff=%d largest_differing_pattern=0x%08x" % (diff11, diff10, max_diff, last))
    assert max_diff <= 1
    assert last < 0x38800000  # nothing at or above 2^-14 differs
    # of the 0x38800000 positive patterns below 2^-14 fewer than 2^-17 may differ
    assert diff11 < 0x38800000 >> 17 and diff10 < 0x38800000 >> 17


@pytest.mark.gpu
@pytest.mark.parametrize("brdf", [0, 1, 2, 3])
def test_fast_brdf_lut_within_a_half_float_step(backend, brdf):
    """kernels_fast/shading_fast.hip brdfLutFastKernel (hardware sin / cos / sqrt / rcp, tabulated sample directions, the shader's summation order)
    against the oracle's LUT (diffuse model 2 at the reference's 512 x 512): every texel within one half-float step, almost all equal"""
    res = 512 if brdf == 2 else 64
    ref = pixfmt.unpack_half(passes.orc_brdf_lut(res, brdf)).reshape(res, res, 4).astype(np.float64)
    backend.setMathMode(True)
    try:
        raw, _ = passes.gpu_brdf_lut(backend, res, brdf)
    finally:
        backend.setMathMode(False)
    got = pixfmt.unpack_half(raw).reshape(res, res, 4).astype(np.float64)
    assert np.isfinite(got).all()
    step = 2.0 ** (np.floor(np.log2(np.maximum(np.abs(ref), 2.0 ** -14))) - 10)  # half-float spacing at the reference value
    steps = np.abs(got - ref) / step
    print("PARITY brdf_lut_fast brdf=%d res=%d equal=%.5f within_1_step=%.5f max_steps=%.2f" % (brdf, res, (steps == 0).mean(), (steps <= 1).mean(), steps.max()))
    assert steps[..., :3].max() <= 1.0 and (steps == 0).mean() >= 0.99  # measured: no texel further than one step, > 99.8 % equal
    assert (got[..., 3] == 0).all()
