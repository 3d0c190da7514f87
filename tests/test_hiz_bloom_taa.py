"""Config 3: min/max HiZ pyramid, bloom chain, TAA resolve.
CPU: oracle known-answer tests from the shader source. GPU: HIP vs oracle through the C-ABI:
bit exact for HiZ; packed R11G11B10 output compared bit for bit for bloom and TAA (the kernels keep the oracle's
operation order, no FMA contraction), with the stated fallback tolerance of one 11/10-bit quantum on <= 1e-4 of texels."""
import numpy as np
import pytest

import passes
from plainrenderer_amd.scene import GlobalShaderInfo, taa_jitter_pixels, taa_resolve_weights
from util import F, hdr_image, pixfmt, rng


def synth_depth(w, h, buffer_id=10, sky_fraction=0.15):
    """reverse-Z depth: a few planes/blobs plus ~15 % sky (= 0.0)"""
    r = rng(buffer_id)
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float32)
    lin = 2.0 + 40.0 * (yy / max(h, 1)) + 8.0 * np.sin(xx / max(w, 1) * 9.0) ** 2
    blob = ((xx - w * 0.3) ** 2 + (yy - h * 0.6) ** 2) < (min(w, h) * 0.2) ** 2
    lin = np.where(blob, 1.5 + 0.001 * xx, lin)
    n, f = 0.1, 300.0
    d = (n * f / lin - n) / (f - n)  # inverse of linearizeDepth
    sky = r.random((h, w)) < sky_fraction * 0.2
    sky |= yy < h * sky_fraction * 0.8
    return np.where(sky, 0.0, d).astype(np.float32)


def synth_motion(w, h, buffer_id=11, max_px=6.0):
    r = rng(buffer_id)
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float32)
    mx = (np.sin(yy / 37.0) * max_px + r.uniform(-0.5, 0.5, (h, w))) / w
    my = (np.cos(xx / 53.0) * max_px * 0.5 + r.uniform(-0.5, 0.5, (h, w))) / h
    return pixfmt.pack_snorm16(np.stack([mx, my], -1).astype(np.float32))


def _global(w, h, **kw):
    return GlobalShaderInfo(screenResolution=(w, h), **kw).pack()


def packed_close(a, b, max_fraction=1e-4):
    """bit equality, or at most one quantum of the 11/10-bit channel on a tiny fraction of texels"""
    a, b = np.asarray(a).reshape(-1), np.asarray(b).reshape(-1)
    if np.array_equal(a, b):
        return True
    diff = a != b
    if diff.mean() > max_fraction:
        return False
    da = pixfmt.unpack_r11g11b10(a[diff]).astype(np.float64)
    db = pixfmt.unpack_r11g11b10(b[diff]).astype(np.float64)
    tol = np.maximum(np.abs(db) * np.array([2.0 ** -6, 2.0 ** -6, 2.0 ** -5]), 1e-6)
    return bool(np.all(np.abs(da - db) <= tol))


# ------------------------------------------------------------------ oracle KATs (CPU)
def test_kat_hiz_constant_and_sky():
    w, h = 64, 48
    mips = passes.orc_hiz(np.full((h, w), 0.25, np.float32), w, h)
    for m in mips:
        assert np.all(m[..., 0] == 0.25) and np.all(m[..., 1] == 0.25)
    mips = passes.orc_hiz(np.zeros((h, w), np.float32), w, h)
    for m in mips:
        assert np.all(m[..., 0] == 1.0) and np.all(m[..., 1] == 0.0)


def test_kat_hiz_apex_is_global_minmax():
    w, h = 256, 128
    d = synth_depth(w, h)
    mips = passes.orc_hiz(d, w, h)
    assert mips[-1].shape[:2] == (1, 1)
    assert mips[-1][0, 0, 0] == d[d != 0].min()
    assert mips[-1][0, 0, 1] == d.max()
    # level 0 is the 2x2 reduction
    blocks = d.reshape(h // 2, 2, w // 2, 2).transpose(0, 2, 1, 3).reshape(h // 2, w // 2, 4)
    assert np.array_equal(mips[0][..., 1], blocks.max(-1))
    assert np.array_equal(mips[0][..., 0], np.where(blocks == 0, 1.0, blocks).min(-1))


def test_kat_hiz_odd_sizes_cover_every_texel():
    # 270x136 -> 135x68 -> 67x34 -> 33x17 ...: odd intermediate sizes use 3-wide footprints, nothing is dropped
    w, h = 540, 272
    d = synth_depth(w, h, buffer_id=12, sky_fraction=0.0)
    d[-1, -1] = 0.999  # the closest texel sits in the last row/column
    mips = passes.orc_hiz(d, w, h)
    assert mips[-1][0, 0, 1] == np.float32(0.999)
    assert mips[-1][0, 0, 0] == d.min()


def test_kat_bloom_constant_energy():
    # down weights sum to 1; every upsample adds the previous level un-normalised: mip0 = 5c, apply = c(1+4s)
    w, h = 128, 64
    c = 0.5
    scene = pixfmt.pack_r11g11b10(np.full((h, w, 3), c, np.float32))
    out, downs, ups = passes.orc_bloom(scene, w, h, strength=0.05)
    for d in downs:
        assert np.all(pixfmt.unpack_r11g11b10(d) == c)
    assert np.all(pixfmt.unpack_r11g11b10(ups[4]) == c)
    assert np.all(pixfmt.unpack_r11g11b10(ups[0]) == 5 * c)
    assert np.allclose(pixfmt.unpack_r11g11b10(out), c * (1 + 4 * 0.05), rtol=2.0 ** -6)


def test_kat_taa_constant_history_is_identity():
    w, h = 64, 32
    c = pixfmt.pack_r11g11b10(np.full((h, w, 3), 0.75, np.float32))
    wts = taa_resolve_weights(taa_jitter_pixels(3))
    assert abs(float(wts.sum()) - 1.0) < 1e-6
    motion = np.zeros((h, w, 2), np.int16)
    out, hist = passes.orc_taa(c, c, motion, synth_depth(w, h), w, h, wts, _global(w, h))
    assert np.allclose(pixfmt.unpack_r11g11b10(out), 0.75, rtol=2.0 ** -6)
    assert np.array_equal(out, hist)


def test_kat_taa_weights_match_host_formula():
    for i in range(8):
        j = taa_jitter_pixels(i)
        assert np.allclose(passes.orc_taa_weights(j), taa_resolve_weights(j), rtol=2e-6)


def test_kat_taa_camera_cut_ignores_history():
    w, h = 48, 40
    cur = hdr_image(w, h, 20)
    hist = hdr_image(w, h, 21)
    wts = taa_resolve_weights(taa_jitter_pixels(1))
    motion = np.zeros((h, w, 2), np.int16)
    a, _ = passes.orc_taa(cur, hist, motion, synth_depth(w, h), w, h, wts, _global(w, h, cameraCut=True))
    b, _ = passes.orc_taa(cur, hdr_image(w, h, 22), motion, synth_depth(w, h), w, h, wts, _global(w, h, cameraCut=True))
    assert np.array_equal(a, b)


# ------------------------------------------------------------------ GPU parity
@pytest.mark.gpu
@pytest.mark.parametrize("fast_set", [False, True])
@pytest.mark.parametrize("w,h", [(3840, 2160), (1920, 1080), (2560, 1432), (540, 272), (130, 94), (64, 48), (34, 2), (4, 4)])
def test_gpu_hiz_bit_exact(backend, w, h, fast_set):
    """both kernel sets: min / max of exact values, nothing is rounded. The fast set's DPP pyramid takes sides that are multiples of 16 (four
    levels without LDS) or of 8 (three: 1920 x 1080, 2560 x 1432), everything else runs the general kernel"""
    d = synth_depth(w, h, buffer_id=30)
    backend.setMathMode(fast_set)
    try:
        got, _, _ = passes.gpu_hiz(backend, d, w, h)
    finally:
        backend.setMathMode(False)
    ref = passes.orc_hiz(d, w, h)
    assert len(got) == len(ref)
    for m, (a, b) in enumerate(zip(got, ref)):
        assert np.array_equal(a.view(np.uint32), b.view(np.uint32)), "mip %d differs" % m
    assert got[-1][0, 0, 0] == d[d != 0].min() and got[-1][0, 0, 1] == d.max()


@pytest.mark.gpu
def test_gpu_hiz_all_sky_and_constant(backend):
    w, h = 256, 192
    got, _, _ = passes.gpu_hiz(backend, np.zeros((h, w), np.float32), w, h)
    for m in got:
        assert np.all(m[..., 0] == 1.0) and np.all(m[..., 1] == 0.0)
    got, _, _ = passes.gpu_hiz(backend, np.full((h, w), 0.125, np.float32), w, h)
    for m in got:
        assert np.all(m == 0.125)


@pytest.mark.gpu
def test_gpu_hiz_rejects_more_than_11_levels(backend):
    from plainrenderer_amd import PlrError
    with pytest.raises(PlrError):
        passes.gpu_hiz(backend, np.zeros((64, 8192), np.float32), 8192, 64)


@pytest.mark.gpu
@pytest.mark.parametrize("w,h", [(1920, 1080), (270, 136), (97, 61)])
def test_gpu_bloom_matches_oracle(backend, w, h):
    scene = hdr_image(w, h, buffer_id=31)
    out_g, downs_g, ups_g = passes.gpu_bloom(backend, scene, w, h)
    out_o, downs_o, ups_o = passes.orc_bloom(scene, w, h)
    for i, (a, b) in enumerate(zip(downs_g, downs_o)):
        assert packed_close(a, b), "down mip %d" % (i + 1)
    for i, (a, b) in enumerate(zip(ups_g, ups_o)):
        assert packed_close(a, b), "up mip %d" % i
    assert packed_close(out_g, out_o)


@pytest.mark.gpu
def test_gpu_bloom_4k_constant_energy(backend):
    # full size property (no oracle needed): constant in -> c(1+4s) out, exactly representable values
    w, h = 3840, 2160
    scene = pixfmt.pack_r11g11b10(np.full((h, w, 3), 0.5, np.float32))
    out, downs, ups = passes.gpu_bloom(backend, scene, w, h, strength=0.25)
    assert np.all(pixfmt.unpack_r11g11b10(ups[0]) == 2.5)
    assert np.all(pixfmt.unpack_r11g11b10(out) == 0.5 * (1 + 4 * 0.25))


@pytest.mark.gpu
@pytest.mark.parametrize("clip,dilate,tech,tonemap", [(True, True, 4, True), (False, True, 0, True), (True, False, 1, False), (True, True, 2, True),
                                                      (False, False, 3, True), (True, True, 4, False)])
def test_gpu_taa_matches_oracle(backend, clip, dilate, tech, tonemap):
    w, h = 320, 180
    cur, hist = hdr_image(w, h, 40), hdr_image(w, h, 41)
    motion, depth = synth_motion(w, h, 42), synth_depth(w, h, 43)
    wts = taa_resolve_weights(taa_jitter_pixels(5))
    g = _global(w, h)
    og, hg = passes.gpu_taa(backend, cur, hist, motion, depth, w, h, wts, g, clip, dilate, tech, tonemap)
    oo, ho = passes.orc_taa(cur, hist, motion, depth, w, h, wts, g, clip, dilate, tech, tonemap)
    assert packed_close(og, oo)
    assert np.array_equal(og, hg) and np.array_equal(oo, ho)


@pytest.mark.gpu
def test_gpu_taa_edges_and_camera_cut(backend):
    w, h = 131, 77  # not a multiple of the 64x4 wave tiling nor of the reference's 8x8 groups
    cur, hist = hdr_image(w, h, 44), hdr_image(w, h, 45)
    motion = synth_motion(w, h, 46, max_px=40.0)  # large motion: many reprojections leave the image -> gaussian fallback
    depth = synth_depth(w, h, 47)
    wts = taa_resolve_weights(taa_jitter_pixels(2))
    for cut in (False, True):
        g = _global(w, h, cameraCut=cut)
        og, _ = passes.gpu_taa(backend, cur, hist, motion, depth, w, h, wts, g)
        oo, _ = passes.orc_taa(cur, hist, motion, depth, w, h, wts, g)
        assert packed_close(og, oo)


@pytest.mark.gpu
def test_gpu_taa_4k_static_scene_is_identity(backend):
    # full size property: zero motion, history == current == constant -> output == input (resolve weights sum to 1)
    w, h = 3840, 2160
    c = pixfmt.pack_r11g11b10(np.full((h, w, 3), 0.75, np.float32))
    wts = taa_resolve_weights(taa_jitter_pixels(3))
    out, hist = passes.gpu_taa(backend, c, c, np.zeros((h, w, 2), np.int16), synth_depth(w, h, 48), w, h, wts, _global(w, h))
    v = pixfmt.unpack_r11g11b10(out)
    assert np.abs(v - 0.75).max() <= 0.75 * 2.0 ** -5
    assert np.array_equal(out, hist)


# ------------------------------------------------------------------ optional TAA stage
def _supersampling_inputs(w, h, seed=31):
    from util import hdr_image
    r = np.random.default_rng(0x504C4149 + seed)
    cur = hdr_image(w, h, buffer_id=seed, pre_exposure=1.0)
    last = hdr_image(w, h, buffer_id=seed + 1, pre_exposure=1.0)
    # mostly similar frames with a band of strong change (contrast rejection) and a depth discontinuity (depth rejection)
    last = np.where(r.random((h, w)) < 0.7, cur.reshape(h, w), last.reshape(h, w)).astype(np.uint32)
    depth = np.full((h, w), 0.02, np.float32)
    depth[:, w // 2:] = 0.2
    depth_last = depth.copy()
    depth_last[h // 3: h // 2] = 0.004
    motion = np.zeros((h, w, 2), np.float32)
    motion[..., 0] = 1.3 / w
    motion[: h // 8] = 0.9  # reprojects off screen
    return cur.reshape(h, w), last, pixfmt.pack_snorm16(motion), depth, depth_last


def test_kat_color_to_luminance_weights():
    px = pixfmt.pack_r11g11b10(np.array([[1.0, 0.0, 0.0], [0.0, 1.0, 0.0], [0.0, 0.0, 1.0], [0.25, 0.25, 0.25], [4.0, 4.0, 4.0]], np.float32))
    lum = passes.orc_color_to_luminance(px.reshape(1, 5), 5, 1).reshape(-1)
    assert lum.tolist() == [round(0.21 * 255), round(0.72 * 255), round(0.07 * 255), round(0.25 * 255), 255]  # luminance.inc weights, R8 store saturates


@pytest.mark.gpu
@pytest.mark.parametrize("w,h,tonemap", [(256, 144, True), (97, 61, False)])
def test_gpu_temporal_supersampling_bit_exact(backend, w, h, tonemap):
    from test_exposure_tonemap import _global
    cur, last, motion, depth, depth_last = _supersampling_inputs(w, h)
    lc_g = passes.gpu_color_to_luminance(backend, cur, w, h)
    lc_o = passes.orc_color_to_luminance(cur, w, h)
    assert np.array_equal(lc_g, lc_o)
    ll = passes.orc_color_to_luminance(last, w, h)
    g = _global(w, h, time=1.0)
    a = passes.gpu_temporal_supersampling(backend, cur, last, motion, depth, depth_last, lc_o, ll, w, h, g, tonemap)
    b = passes.orc_temporal_supersampling(cur, last, motion, depth, depth_last, lc_o, ll, w, h, g, tonemap)
    assert np.array_equal(a, b)
    # the test data exercises both outcomes of every rejection test
    same = a == cur
    assert 0.05 < same.mean() < 0.95


@pytest.mark.gpu
@pytest.mark.parametrize("w,h,tonemap", [(256, 144, True), (1280, 720, True), (256, 144, False), (97, 61, True)])
def test_gpu_fast_temporal_supersampling_same_decisions_within_one_code(backend, w, h, tonemap):
    """PLR_MATH_FAST kernels of the optional TAA stage (kernels_fast/supersampling_fast.hip): the luminance image is bit exact (its R8 rounding is a discrete
    result), and the supersampled colour takes the oracle's decisions by construction (decision arithmetic in the exact set's form), so EVERY channel of EVERY
    pixel is within one R11G11B10 code - no outliers. 97 x 61 is not a multiple of four wide: colorToLuminance then runs the general kernel, said loudly."""
    import parity
    from test_exposure_tonemap import _global
    cur, last, motion, depth, depth_last = _supersampling_inputs(w, h)
    g = _global(w, h, time=1.0)
    lc_o = passes.orc_color_to_luminance(cur, w, h)
    ll = passes.orc_color_to_luminance(last, w, h)
    b = passes.orc_temporal_supersampling(cur, last, motion, depth, depth_last, lc_o, ll, w, h, g, tonemap)
    backend.setMathMode(True)
    try:
        lc_g = passes.gpu_color_to_luminance(backend, cur, w, h)
        n_general, names = backend.getGeneralKernelExecutions()
        assert n_general == (0 if w % 4 == 0 else 1), (n_general, names)
        a = passes.gpu_temporal_supersampling(backend, cur, last, motion, depth, depth_last, lc_o, ll, w, h, g, tonemap)
        assert backend.getGeneralKernelExecutions()[0] == 0, backend.getGeneralKernelExecutions()
    finally:
        backend.setMathMode(False)
    assert np.array_equal(lc_g, lc_o)
    d = parity.r11g11b10_code_diff(np.asarray(a).reshape(-1), np.asarray(b).reshape(-1))
    print("SUPERSAMPLING fast %dx%d tonemap=%d max_code_diff=%d differing=%.3g" % (w, h, tonemap, int(d.max()), float((d != 0).any(axis=1).mean())))
    assert d.max() <= 1
    taken = np.asarray(a).reshape(h, w) != cur  # both outcomes of the rejection tests occur
    assert 0.05 < taken.mean() < 0.95
