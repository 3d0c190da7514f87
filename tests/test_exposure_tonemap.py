"""Config 2: luminance-histogram auto-exposure + tonemap.
CPU part: known-answer tests of the oracle that follow from the shader source.
GPU part: HIP kernels vs oracle through the C-ABI (bit exact for histogram bins and the light buffer,
+-1 LSB of the 8-bit target for the tonemap, whose pow() uses the hardware log2/exp2)."""
import math

import numpy as np
import pytest

import passes
from plainrenderer_amd.scene import GlobalShaderInfo
from util import F, hdr_image, light_buffer_bytes, pixfmt


def _global(w, h, time=37.25, **kw):
    g = GlobalShaderInfo(screenResolution=(w, h), time=time, deltaTime=1.0 / 60.0, **kw)
    return g.pack()


# ------------------------------------------------------------------ oracle KATs (CPU)
def test_kat_constant_image_single_bin():
    w, h = 96, 70  # partial last tile row (70 = 2*32 + 6 >= 4 rows of the tile -> well defined)
    img = pixfmt.pack_r11g11b10(np.full((h, w, 3), 0.25, np.float32))
    per_tile, hist = passes.orc_histogram(img, w, h, light_buffer_bytes(prev_exposure=1e-3))
    assert hist.sum() == w * h
    assert (hist != 0).sum() == 1
    # bin = uint(127 * (log(lum/exposure) - log(min)) / (log(max) - log(min)))
    lum = 0.25 * (0.2126 + 0.7152 + 0.0722) / 1e-3
    expect = int(127 * (math.log(lum) - math.log(0.001)) / (math.log(200000.0) - math.log(0.001)))
    assert abs(int(np.argmax(hist)) - expect) <= 1
    assert per_tile.reshape(-1, 128).sum(axis=1).tolist() == [1024, 1024, 1024, 1024, 1024, 1024, 192, 192, 192]


def test_kat_black_pixels_land_in_bin0():
    w, h = 64, 64
    img = pixfmt.pack_r11g11b10(np.zeros((h, w, 3), np.float32))
    _, hist = passes.orc_histogram(img, w, h, light_buffer_bytes())
    assert hist[0] == w * h and hist[1:].sum() == 0


def test_kat_tonemap_properties():
    # constant image, dither pinned: ACES is monotone and clamps to [0,1]; sRGB knee
    w, h = 8, 1
    vals = np.array([0.0, 1e-4, 0.0031308 / 0.6, 0.01, 0.18, 1.0, 10.0, 1e4], np.float32)
    rgb = np.repeat(vals[None, :, None], 3, axis=2)
    out = passes.orc_tonemap(pixfmt.pack_r11g11b10(rgb), w, h, _global(w, h, time=0.0))
    grey = out[0, :, 1].astype(int)  # G channel
    assert np.all(np.diff(grey) >= -1)  # monotone up to the +-1 LSB dither
    assert grey[0] <= 1 and grey[-1] >= 254
    assert np.all(out[..., 3] == 255)


def test_kat_exposure_ev_floor_and_slew():
    # only bins whose cumulative share lies in [0.5, 0.95) count (preExposeLights.comp:56); target EV is floored at 10 (:72)
    hist = np.zeros(128, np.uint32)
    n = 640 * 360
    hist[30], hist[40], hist[50] = n * 6 // 10, n * 3 // 10, n // 10
    lut = pixfmt.pack_r11g11b10(np.full((4, 4, 3), 0.5, np.float32))
    g = _global(640, 360, exposureAdaptionSpeedEvPerSec=1e6)  # no slew limit
    lb = passes.orc_pre_expose(hist, light_buffer_bytes(prev_exposure=1e-3), lut, 4, 4, g)
    exposure = lb[3]
    ev = math.log2(1.0 / (exposure * 1.2))
    assert ev == pytest.approx(10.0, abs=1e-3)
    assert lb[4] == pytest.approx(128000.0 * exposure, rel=1e-6)
    assert lb[:3].tolist() == pytest.approx([0.5, 0.5, 0.5])
    # slew limited: 2 EV/s * 1/60 s
    g = _global(640, 360)
    lb2 = passes.orc_pre_expose(hist, light_buffer_bytes(prev_exposure=1e-2), lut, 4, 4, g)
    ev_prev = math.log2(1.0 / (1e-2 * 1.2))
    ev_now = math.log2(1.0 / (lb2[3] * 1.2))
    assert ev_now - ev_prev == pytest.approx(2.0 / 60.0, rel=1e-3)
    # reference quirk: if no bin qualifies the mean is 0/0; max(NaN, 10) then resolves to 10 as on GPU hardware
    # (IEEE maxNum), which is how the reference recovers from its all-black first frames
    one = np.zeros(128, np.uint32)
    one[40] = n
    g = _global(640, 360, exposureAdaptionSpeedEvPerSec=1e6)
    lb3 = passes.orc_pre_expose(one, light_buffer_bytes(), lut, 4, 4, g)
    assert math.log2(1.0 / (lb3[3] * 1.2)) == pytest.approx(10.0, abs=1e-3)


# ------------------------------------------------------------------ GPU parity
@pytest.mark.gpu
def test_gpu_histogram_threshold_table_equals_the_formula_for_every_float(backend):
    """the PLR_MATH_FAST per-tile histogram bins by comparing against a bisected threshold table (kernels_fast/histogram_fast.hip): for all 2^32
    float bit patterns - zeros, denormals, infinities and NaNs included - that bin is the bin of histogramPerTile.comp:53-57"""
    assert backend.debugVerifyHistogramThresholds(passes.MIN_LUM, passes.MAX_LUM) == 0
    assert backend.debugVerifyHistogramThresholds(0.5, 37.0) == 0


@pytest.mark.gpu
@pytest.mark.parametrize("fast_set", [False, True])
@pytest.mark.parametrize("w,h", [(1920, 1080), (256, 144), (131, 77), (33, 5)])
def test_gpu_histogram_bit_exact(backend, w, h, fast_set):
    img = hdr_image(w, h, buffer_id=1)
    if fast_set:
        # bright outliers, exact zeros and a few non-finite texels: every class of input the binning can see
        r = np.random.default_rng(11)
        flat = img.reshape(-1).copy()
        flat[r.integers(0, flat.size, max(flat.size // 50, 1))] = 0
        flat[r.integers(0, flat.size, 4)] = 0x7C0 | (0x7C0 << 11) | (0x3E0 << 22)      # +inf in every channel
        flat[r.integers(0, flat.size, 4)] = 0x7FF | (0x7FF << 11) | (0x3FF << 22)      # NaN in every channel
        img = flat.reshape(img.shape)
    lb = light_buffer_bytes(prev_exposure=3.7e-4)
    backend.setMathMode(fast_set)
    try:
        pt_g, h_g = passes.gpu_histogram(backend, img, w, h, lb)
    finally:
        backend.setMathMode(False)
    pt_o, h_o = passes.orc_histogram(img, w, h, lb)
    assert np.array_equal(h_g, h_o)
    assert np.array_equal(pt_g, pt_o)
    if w % 32 == 0 and (h % 32 == 0 or h % 32 >= 4) and not fast_set:  # else the write-back quirk drops bins (see oracle)
        assert int(h_g.sum()) == w * h


@pytest.mark.gpu
def test_gpu_histogram_4k_checksum(backend):
    # full-size property: the 128 counts sum to W*H (a checksum of checksums), bins spread over the range
    w, h = 3840, 2160
    img = hdr_image(w, h, buffer_id=2)
    pt, hist = passes.gpu_histogram(backend, img, w, h, light_buffer_bytes(prev_exposure=3.7e-4))
    assert int(hist.sum()) == w * h
    assert np.array_equal(pt.reshape(-1, 128).sum(axis=0).astype(np.uint32), hist)
    assert (hist > 0).sum() > 32


@pytest.mark.gpu
def test_gpu_pre_expose_bit_exact(backend):
    w, h = 1920, 1080
    img = hdr_image(w, h, buffer_id=3)
    lb = light_buffer_bytes(prev_exposure=3.7e-4)
    _, hist = passes.orc_histogram(img, w, h, lb)
    r = np.random.default_rng(11)
    lut = pixfmt.pack_r11g11b10(r.uniform(0.1, 1.0, (128, 128, 3)).astype(np.float32))
    for sun_y, speed in ((-0.8, 2.0), (-0.123, 1e6), (0.3, 0.5)):
        g = _global(w, h, sunDirection=(0.3, sun_y, 0.5, 0.0), exposureAdaptionSpeedEvPerSec=speed)
        a = passes.gpu_pre_expose(backend, hist, lb, lut, 128, 128, g)
        b = passes.orc_pre_expose(hist, lb, lut, 128, 128, g)
        assert np.array_equal(a.view(np.uint32), b.view(np.uint32)), (a, b)


@pytest.mark.gpu
@pytest.mark.parametrize("w,h", [(1920, 1080), (130, 67)])
def test_gpu_tonemap_within_one_lsb(backend, w, h):
    img = hdr_image(w, h, buffer_id=4, pre_exposure=1e-3)
    g = _global(w, h, time=37.25)
    fmt = None if (w, h) == (1920, 1080) else F.BGRA8_uNorm
    a = passes.gpu_tonemap(backend, img, w, h, g, fmt).astype(int)
    b = passes.orc_tonemap(img, w, h, g).astype(int)
    d = np.abs(a - b)
    assert d.max() <= 1  # stated tolerance: +-1/255
    assert (d != 0).mean() < 0.02
    assert np.all(a[..., 3] == 255)


@pytest.mark.gpu
def test_gpu_tonemap_rgba8_target_swaps_channels(backend):
    w, h = 64, 32
    img = hdr_image(w, h, buffer_id=5)
    g = _global(w, h, time=1.5)
    a = passes.gpu_tonemap(backend, img, w, h, g, F.RGBA8).astype(int)
    b = passes.orc_tonemap(img, w, h, g, F.RGBA8).astype(int)
    assert np.abs(a - b).max() <= 1
