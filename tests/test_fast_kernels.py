"""PLR_MATH_FAST kernels that have no discrete decision (bloom chain, tonemap, the R11G11B10 encoder) against the oracle at small and odd sizes.

The passes whose float rounding can flip a discrete decision (ray hit, nearest texel, PCF tap, edge test) - trace, spatial / temporal GI filters,
upscale, deferred shade, TAA - are held to the storage-quantum statement of tests/parity.py with decision signatures, at the benchmarked size for
the default variant (tests/test_parity_fullsize.py) and at 1920 x 1088 for every other variant the fast set ships (tests/test_variants_parity.py).
There is no outlier allowance anywhere: every value must satisfy |got - ref| <= max(REL * |ref|, ABS_FLOOR_FRAC * max|ref|).
"""
import numpy as np
import pytest

import passes
from plainrenderer_amd import pixfmt
from util import F


def assert_close(got, ref, rel, abs_floor_frac=1e-4, mean_rel=1e-3, what=""):
    got = np.asarray(got, np.float64).reshape(-1)
    ref = np.asarray(ref, np.float64).reshape(-1)
    assert np.isfinite(got).all(), what + ": non-finite values"
    scale = np.abs(ref).max()
    tol = np.maximum(rel * np.abs(ref), abs_floor_frac * scale)
    err = np.abs(got - ref)
    assert not (err > tol).any(), "%s: %d values outside tolerance, worst %.3g of scale %.3g" % (what, int((err > tol).sum()), err.max(), scale)
    assert err.mean() <= mean_rel * np.abs(ref).mean() + 1e-12, "%s: mean error %.3g vs mean value %.3g" % (what, err.mean(), np.abs(ref).mean())


@pytest.fixture()
def fast(backend):
    backend.setMathMode(True)
    yield backend
    backend.setMathMode(False)


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
        assert_close(pixfmt.unpack_r11g11b10(a), pixfmt.unpack_r11g11b10(b), rel=2.0 ** -5, abs_floor_frac=1e-4, mean_rel=2e-3, what="up mip %d" % i)
    assert_close(pixfmt.unpack_r11g11b10(out_g), pixfmt.unpack_r11g11b10(out_o), rel=2.0 ** -5, abs_floor_frac=1e-4, mean_rel=2e-3, what="applied")


@pytest.mark.gpu
def test_gpu_fast_bloom_constant_energy(fast):
    w, h = 3840, 2160
    scene_img = pixfmt.pack_r11g11b10(np.full((h, w, 3), 0.5, np.float32))
    out, downs, ups = passes.gpu_bloom(fast, scene_img, w, h, strength=0.25)
    assert np.all(pixfmt.unpack_r11g11b10(ups[0]) == 2.5)
    assert np.all(pixfmt.unpack_r11g11b10(out) == 0.5 * (1 + 4 * 0.25))


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


# ------------------------------------------------------------------ PCF tap table of the deferred shade
def test_kat_pcf_taps_of_the_oracle_against_float64():
    """the oracle's restatement of triangle.frag:100-110 per noise value: unit-disc offsets (cos, sin)(noise 2 pi + 2 pi i / 12) * sqrt((i + noise / 2) / 12)"""
    taps = passes.orc.kat_pcf_taps().astype(np.float64)
    noise = (np.arange(256) / 255.0)[:, None]
    i = np.arange(12)[None, :]
    d = np.sqrt((i + 0.5 * noise) / 12.0)
    angle = noise * 2 * np.pi + 2 * np.pi * i / 12.0
    assert np.abs(taps[..., 0] - np.cos(angle) * d).max() < 2e-6 and np.abs(taps[..., 1] - np.sin(angle) * d).max() < 2e-6
    assert np.abs(np.hypot(taps[..., 0], taps[..., 1]) - d).max() < 1e-6


@pytest.mark.gpu
def test_gpu_pcf_tap_table_is_the_oracles_bit_for_bit(backend):
    """the table the PLR_MATH_FAST shade reads its twelve shadow taps from holds, for all 256 noise values, the bits the shader's own
    sqrt / sin / cos expressions give (software sin / cos of detmath.h, IEEE sqrt and divide)"""
    got = backend.debugPcfTapTable()
    ref = passes.orc.kat_pcf_taps()
    assert np.array_equal(got.view(np.uint32), ref.view(np.uint32)), "%d of 6144 table floats differ" % int((got.view(np.uint32) != ref.view(np.uint32)).sum())
