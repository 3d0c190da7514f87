"""Known-answer tests that pin the ORACLE (and, through the sampler probe, the HIP sampler code) to closed forms that follow from the
shader text alone - independent of the restatement they check. The reference holds no vectors for this path, so these
properties are the pin:
  * Catmull-Rom weights form a partition of unity and every history sampler reproduces constants / linear ramps (bicubicSampling.inc:28-181)
  * clipAABB leaves inside points alone and puts outside points on the box along the line to its centre (temporalReprojection.inc:8-30)
  * importanceSampleCosine: unit vectors, cos(theta) = sqrt(xi.x), mean direction 2/3 N (sampling.inc:25-45)
  * a ray traced against an analytic sphere volume stops at the ray-sphere distance +- the hit threshold (SDF.inc:101-184)
  * textureGather order = the offsets table of indirectLightUpscale.comp:42-47
  * the 8 global samplers (global.inc:35-42): nearest / linear x clamp / repeat / border white / border black against an independent numpy
    statement of the Vulkan rules on adversarial coordinates, and the HIP samplers bit for bit against the oracle's.
"""
import math
import struct

import numpy as np
import pytest

import pyoracle as orc
from plainrenderer_amd import pixfmt
from util import F, image_desc_2d


def rng(seed):
    return np.random.default_rng(0x504C4149 + seed)


# ------------------------------------------------------------------ Catmull-Rom / history samplers
def test_kat_catmull_rom_partition_of_unity():
    iuv = rng(1).uniform(2.0, 60.0, (4096, 2)).astype(np.float32)
    w = orc.kat_taa(0, iuv, 2, 16).astype(np.float64)
    # 16-tap weights (bicubicSampling.inc:28-45): four per axis, sum 1
    assert np.abs(w[:, 0:4].sum(1) - 1).max() < 2e-6 and np.abs(w[:, 4:8].sum(1) - 1).max() < 2e-6
    # the bilinear-folded form (bicubicSampling.inc:72-90): w0 + (w1 + w2) + w3 = 1, the folded tap sits between texel 1 and 2
    for o in (8, 12):
        assert np.abs(w[:, o] + w[:, o + 1] + w[:, o + 2] - 1).max() < 2e-6
        assert (w[:, o + 3] >= 0).all() and (w[:, o + 3] <= 1).all()
    # closed form at f = 0 (on a texel centre) and f = 0.5
    k = orc.kat_taa(0, np.array([[10.5, 20.5], [11.0, 21.0]], np.float32), 2, 16)
    assert np.allclose(k[0, 0:4], [0, 1, 0, 0], atol=1e-7) and np.allclose(k[0, 8:12], [0, 1, 0, 0], atol=1e-7)
    assert np.allclose(k[1, 0:4], [-1 / 16, 9 / 16, 9 / 16, -1 / 16], atol=1e-7)
    assert np.allclose(k[1, 8:12], [-1 / 16, 18 / 16, -1 / 16, 0.5], atol=1e-7)


@pytest.mark.parametrize("tech", [0, 1, 2, 3, 4])
def test_kat_history_samplers_reproduce_a_constant(tech):
    """partition of unity through the sampler code itself: a constant image comes back as the constant (1-tap: with a constant neighbourhood)"""
    w, h = 48, 32
    val = np.array([0.75, 2.5, 0.125], np.float32)  # exactly representable in R11G11B10
    img = orc.Img(pixfmt.pack_r11g11b10(np.broadcast_to(val, (h, w, 3)).copy()), w, h, F.R11G11B10_uFloat)
    iuv = rng(2 + tech).uniform(0.0, [w, h], (2048, 2)).astype(np.float32)  # includes footprints that clamp at the border
    nbr = np.broadcast_to(val, (9, 3)).copy()
    out = orc.kat_history_sample(img, tech, iuv, nbr)
    assert np.abs(out - val).max() <= 4e-6 * val.max(), np.abs(out - val).max()


@pytest.mark.parametrize("tech", [1, 2])
def test_kat_bicubic_reproduces_a_linear_ramp(tech):
    """Catmull-Rom interpolation is exact for linear functions; 16 tap fetches texel centres, 9 tap folds taps 1, 2 into one bilinear fetch"""
    w, h = 64, 48
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float32)
    ramp = np.stack([0.25 * xx + 1.0, 0.5 * yy + 2.0, 0.125 * (xx + yy) + 3.0, np.ones_like(xx)], -1).astype(np.float32)
    img = orc.Img(ramp, w, h, F.RGBA32_sFloat)
    iuv = rng(10 + tech).uniform(4.0, [w - 4.0, h - 4.0], (2048, 2)).astype(np.float32)
    out = orc.kat_history_sample(img, tech, iuv).astype(np.float64)
    x, y = iuv[:, 0].astype(np.float64) - 0.5, iuv[:, 1].astype(np.float64) - 0.5  # pixel units -> texel index space
    expect = np.stack([0.25 * x + 1.0, 0.5 * y + 2.0, 0.125 * (x + y) + 3.0], -1)
    # 16 tap: float rounding only. 9 tap: the folded tap position is quantised to 1/256 texel by the bilinear filter (slope * 1/512 per axis)
    tol = 6e-5 if tech == 1 else 0.5 / 512 + 0.125 / 256 + 6e-5
    assert np.abs(out - expect).max() <= tol, np.abs(out - expect).max()


def test_kat_clip_aabb():
    r = rng(20)
    mn = r.uniform(-1, 0, (512, 3)).astype(np.float32)
    mx = mn + r.uniform(0.1, 2, (512, 3)).astype(np.float32)
    centre, ext = 0.5 * (mx + mn), 0.5 * (mx - mn) + 1e-4
    inside = (centre + r.uniform(-0.95, 0.95, (512, 3)) * ext).astype(np.float32)
    out = orc.kat_taa(1, np.concatenate([inside, mn, mx], 1), 9, 3)
    assert np.array_equal(out, inside), "a colour inside the box is returned unchanged"
    direction = r.choice([-1, 1], (512, 3)) * r.uniform(0.2, 1, (512, 3))
    direction /= np.abs(direction).max(1, keepdims=True)  # largest normalised component = 1: on the box surface
    outside = (centre + r.uniform(1.5, 4, (512, 1)) * direction * ext).astype(np.float32)
    out = orc.kat_taa(1, np.concatenate([outside, mn, mx], 1), 9, 3).astype(np.float64)
    n_in, n_out = (outside - centre) / ext, (out - centre) / ext
    assert (np.abs(n_in).max(1) >= 1).all()
    assert np.abs(np.abs(n_out).max(1) - 1).max() < 1e-5, "clipped colours lie on the (epsilon-padded) box surface"
    cross = np.cross(out - centre, outside - centre)
    assert np.abs(cross).max() < 1e-5 and ((out - centre) * (outside - centre)).sum(1).min() > 0, "on the segment from the box centre to the colour"
    # degenerate box (min == max): the result collapses onto the centre to within the epsilon
    p = np.array([[5.0, -3.0, 2.0, 1.0, 1.0, 1.0, 1.0, 1.0, 1.0]], np.float32)
    assert np.abs(orc.kat_taa(1, p, 9, 3) - 1.0).max() <= 1.01e-4
    # tonemap / tonemapReverse are inverse to each other (temporalReprojection.inc:34-40) and use the (0.21, 0.72, 0.07) luma
    c = r.uniform(0, 50, (256, 3)).astype(np.float32)
    t = orc.kat_taa(2, c, 3, 6).astype(np.float64)
    lum = c @ np.array([0.21, 0.72, 0.07])
    assert np.allclose(t[:, :3], c / (1 + lum)[:, None], rtol=2e-6) and np.allclose(t[:, 3:], c, rtol=2e-4)


# ------------------------------------------------------------------ sampling
def test_kat_importance_sample_cosine():
    r = rng(30)
    n_dirs = r.normal(size=(8, 3))
    n_dirs /= np.linalg.norm(n_dirs, axis=1, keepdims=True)
    n_dirs = np.concatenate([n_dirs, [[0, 0, 1], [0, 0, -1], [0.01, 0.0, 0.99995]]])  # both branches of the basis choice (|N.z| >= 0.999)
    g = (np.arange(64) + 0.5) / 64
    xi = np.stack(np.meshgrid(g, g, indexing="ij"), -1).reshape(-1, 2)
    for N in n_dirs:
        N = N / np.linalg.norm(N)
        data = np.concatenate([xi, np.broadcast_to(N, (xi.shape[0], 3))], 1).astype(np.float32)
        L = orc.kat_sampling(0, data, 5, 3).astype(np.float64)
        assert np.abs(np.linalg.norm(L, axis=1) - 1).max() < 2e-6
        assert np.abs(L @ N - np.sqrt(xi[:, 0])).max() < 2e-6, "cos(theta) = sqrt(xi.x)"
        mean = L.mean(0)
        assert np.abs(mean - (2.0 / 3.0) * N).max() < 2e-3, "cosine-weighted hemisphere: E[L] = 2/3 N"
    # azimuth: xi.y = 0 and xi.y = 0.5 are opposite directions around N
    N = np.array([0.3, 0.5, 0.8]); N /= np.linalg.norm(N)
    a = orc.kat_sampling(0, np.array([[0.3, 0.0, *N], [0.3, 0.5, *N]], np.float32), 5, 3).astype(np.float64)
    ta, tb = a[0] - (a[0] @ N) * N, a[1] - (a[1] @ N) * N
    assert np.abs(ta + tb).max() < 1e-6
    # directionToSH_L1 (SphericalHarmonics.inc): normalize((0.282095, -0.488603 y, 0.488603 z, -0.488603 x)) of a unit vector = (0.5, -0.866 y, 0.866 z, -0.866 x)
    v = n_dirs[:8].astype(np.float32)
    sh = orc.kat_sampling(1, v, 3, 4)
    assert np.allclose(sh, np.stack([np.full(8, 0.5), -0.8660254 * v[:, 1], 0.8660254 * v[:, 2], -0.8660254 * v[:, 0]], 1), atol=2e-6)


# ------------------------------------------------------------------ trace against an analytic sphere
def _sphere_instance(res, radius, extent):
    lin = ((np.arange(res) + 0.5) / res - 0.5) * extent
    zz, yy, xx = np.meshgrid(lin, lin, lin, indexing="ij")
    vol = pixfmt.pack_half((np.sqrt(xx * xx + yy * yy + zz * zz) - radius).astype(np.float32))
    inst = struct.pack("<3fI3ff", extent, extent, extent, 0, 0.5, 0.5, 0.5, 0.0) + np.eye(4, dtype=np.float32).tobytes()  # worldToLocal = identity
    return inst, orc.Img(vol, res, res, F.R16_sFloat, d=res)


def test_kat_trace_hits_analytic_sphere_at_the_ray_sphere_distance():
    res, radius, extent = 64, 1.5, 4.0
    inst, vol = _sphere_instance(res, radius, extent)
    r = rng(40)
    d = r.normal(size=(512, 3)); d /= np.linalg.norm(d, axis=1, keepdims=True)
    origin = -d * r.uniform(2.5, 6.0, (512, 1))                      # outside the volume's box, looking at ...
    target = r.normal(size=(512, 3)); target *= r.uniform(0, 1.2, (512, 1)) / np.linalg.norm(target, axis=1, keepdims=True)  # ... a point inside the sphere
    dirs = target - origin; dirs /= np.linalg.norm(dirs, axis=1, keepdims=True)
    out = orc.kat_trace_ray(inst, vol, np.concatenate([origin, dirs], 1)).astype(np.float64)
    b = (origin * dirs).sum(1)
    t_hit = -b - np.sqrt(b * b - ((origin * origin).sum(1) - radius * radius))  # analytic ray-sphere distance
    threshold = math.sqrt(3.0) * (extent / res) * 0.25                         # distanceThreshold = length(extends / resolution) * 0.25 (SDF.inc:144)
    assert (out[:, 0] == 1).all(), "every ray aimed into the sphere hits"
    # the march stops the first time the sampled distance drops below the threshold: between (threshold / cos) before the surface and the surface;
    # half-float storage + trilinear interpolation of a curved field add at most a fraction of a voxel
    slack = 0.35 * extent / res
    assert (out[:, 1] <= t_hit + slack).all() and (out[:, 1] >= t_hit - 4.0 * threshold - slack).all(), (np.abs(out[:, 1] - t_hit).max(), threshold)
    # hitPos = start + dir * (distance + last step): the extrapolated point lies on the sphere to within the threshold
    rad = np.linalg.norm(out[:, 2:5], axis=1)
    assert np.abs(rad - radius).max() <= threshold + slack, np.abs(rad - radius).max()
    # SDF normal at the hit = radial direction
    nrm = out[:, 5:8] / np.linalg.norm(out[:, 5:8], axis=1, keepdims=True)
    assert ((nrm * (out[:, 2:5] / rad[:, None])).sum(1) > 0.995).all()
    # rays that pass the sphere at 1.5 voxel or more never come within the threshold: miss
    side = np.cross(dirs, r.normal(size=(512, 3))); side /= np.linalg.norm(side, axis=1, keepdims=True)
    o2 = side * (radius + 1.5 * extent / res + threshold) - dirs * 5.0
    miss = orc.kat_trace_ray(inst, vol, np.concatenate([o2, dirs], 1))
    assert (miss[:, 0] == 0).all() and (miss[:, 1] == 10000.0).all()
    # a ray that starts inside the box is traced from its origin (no AABB entry step): distance measured from there
    inside = orc.kat_trace_ray(inst, vol, np.array([[0.0, 0.0, -1.9, 0.0, 0.0, 1.0]], np.float32))[0]
    assert inside[0] == 1 and abs(inside[1] - 0.4) <= 4 * threshold + slack


# ------------------------------------------------------------------ samplers: independent numpy statement of the Vulkan rules
def _np_sample(tex, filt, addr, uv):
    """tex: [h, w, c] float64 decoded texels. Vulkan 1.2 'Texel Filtering' with 8 fractional weight bits (DESIGN.md sampler contract)."""
    h, w, _ = tex.shape
    border = np.array([1.0, 1.0, 1.0, 1.0] if addr == 2 else [0.0, 0.0, 0.0, 1.0])[: tex.shape[2]]

    def texel(i, j):
        if addr == 0:
            i, j = min(max(i, 0), w - 1), min(max(j, 0), h - 1)
        elif addr == 1:
            i, j = i % w, j % h
        elif i < 0 or j < 0 or i >= w or j >= h:
            return border
        return tex[j, i]
    out = []
    for u, v in uv:
        u32, v32 = np.float32(np.float32(u) * np.float32(w)), np.float32(np.float32(v) * np.float32(h))
        if filt == 0:
            out.append(texel(int(math.floor(u32)), int(math.floor(v32))))
            continue
        tu = int(math.floor(np.float32(np.float32(u32 - np.float32(0.5)) * np.float32(256.0)) + np.float32(0.5)))
        tv = int(math.floor(np.float32(np.float32(v32 - np.float32(0.5)) * np.float32(256.0)) + np.float32(0.5)))
        i0, a, j0, b = tu >> 8, (tu & 255) / 256.0, tv >> 8, (tv & 255) / 256.0
        if filt == 2:
            out.append(np.array([texel(i0, j0 + 1)[0], texel(i0 + 1, j0 + 1)[0], texel(i0 + 1, j0)[0], texel(i0, j0)[0]]))
            continue
        out.append((1 - a) * (1 - b) * texel(i0, j0) + a * (1 - b) * texel(i0 + 1, j0) + (1 - a) * b * texel(i0, j0 + 1) + a * b * texel(i0 + 1, j0 + 1))
    return np.array(out)


def _adversarial_uv(w, h, seed):
    r = rng(seed)
    k = np.arange(-2, w + 3)
    edges = np.concatenate([k / w, (k + 0.5) / w, np.nextafter((k / w).astype(np.float32), np.float32(-9)), np.nextafter((k / w).astype(np.float32), np.float32(9))])
    u = np.concatenate([edges, r.uniform(-1.5, 2.5, 200), [0.0, 1.0, -0.0, 1e-8, 1 - 1e-8, -3.75, 4.25]])
    v = r.permutation(np.concatenate([np.arange(-2, h + 3) / h, r.uniform(-1.5, 2.5, u.size)]))[: u.size]
    return np.stack([u, v], 1).astype(np.float32)


def _rgba16f_image(w, h, seed):
    vals = rng(seed).uniform(-4, 4, (h, w, 4)).astype(np.float32)
    packed = pixfmt.pack_half(vals)
    return packed, pixfmt.unpack_half(packed).reshape(h, w, 4).astype(np.float64)


@pytest.mark.parametrize("filt", [0, 1, 2])
@pytest.mark.parametrize("addr", [0, 1, 2, 3])
def test_kat_samplers_follow_the_vulkan_rules(filt, addr):
    w, h = 13, 7
    packed, dec = _rgba16f_image(w, h, 50)
    uv = _adversarial_uv(w, h, 51 + filt * 4 + addr)
    got = orc.sampler_eval(orc.Img(packed, w, h, F.RGBA16_sFloat), filt, addr, uv).astype(np.float64)
    ref = _np_sample(dec, filt, addr, uv)
    if filt != 1:
        assert np.array_equal(got if filt == 0 else got, ref if filt == 0 else ref), "nearest / gather return texels verbatim"
    else:
        assert np.abs(got - ref).max() <= 4e-6, np.abs(got - ref).max()


def test_kat_texture_gather_order_matches_the_upscale_offsets():
    """indirectLightUpscale.comp:42-47 pairs gather component i with offsets[i] = (0,1), (1,1), (1,0), (0,0)"""
    w, h = 8, 6
    yy, xx = np.mgrid[0:h, 0:w]
    vals = (xx + 100 * yy).astype(np.float32)
    img = orc.Img(pixfmt.pack_half(vals), w, h, F.R16_sFloat)
    # uv on the corner shared by texels (2..3, 1..2): i0 = 2, j0 = 1
    out = orc.sampler_eval(img, 2, 0, np.array([[3.0 / w, 2.0 / h]], np.float32))[0]
    assert out.tolist() == [2 + 200, 3 + 200, 3 + 100, 2 + 100]


# ------------------------------------------------------------------ HIP samplers against the oracle's, bit for bit (a17)
SAMPLER_FORMATS = [("RGBA16_sFloat", 8), ("R11G11B10_uFloat", 4), ("Depth32", 4), ("Depth16", 2), ("RG16_sNorm", 4), ("RGBA8", 4), ("RG8", 2), ("R16_sFloat", 2)]


@pytest.mark.gpu
@pytest.mark.parametrize("fmt_name,bpt", SAMPLER_FORMATS)
def test_gpu_sampler_probe_2d_bit_exact(backend, fmt_name, bpt):
    fmt = getattr(F, fmt_name)
    w, h = 13, 7
    r = rng(60)
    if fmt_name == "Depth32":
        raw = r.uniform(0, 1, (h, w)).astype(np.float32).view(np.uint8)
    elif fmt_name in ("RGBA16_sFloat", "R16_sFloat"):
        raw = pixfmt.pack_half(r.uniform(-4, 4, (h, w, bpt // 2)).astype(np.float32)).view(np.uint8)
    elif fmt_name == "R11G11B10_uFloat":
        raw = pixfmt.pack_r11g11b10(r.uniform(0, 8, (h, w, 3)).astype(np.float32)).view(np.uint8)
    else:
        raw = r.integers(0, 256, (h, w, bpt), dtype=np.uint8)
    raw = np.ascontiguousarray(raw).reshape(-1)
    gimg = backend.createImage(image_desc_2d(w, h, fmt), raw)
    oimg = orc.Img(raw, w, h, fmt)
    for filt in (0, 1, 2):
        for addr in (0, 1, 2, 3):
            uv = _adversarial_uv(w, h, 70 + filt * 4 + addr)
            a = backend.debugSamplerEval(gimg, filt, addr, uv)
            b = orc.sampler_eval(oimg, filt, addr, uv)
            assert np.array_equal(a.view(np.uint32), b.view(np.uint32)), "%s filter %d address %d: %d of %d results differ" % (
                fmt_name, filt, addr, int((a.view(np.uint32) != b.view(np.uint32)).any(1).sum()), uv.shape[0])


@pytest.mark.gpu
def test_gpu_sampler_probe_3d_bit_exact(backend):
    """R16F SDF volumes (trilinear clamp: the trace) and RGBA16F froxel volumes (trilinear clamp / repeat noise)"""
    from plainrenderer_amd.backend import ImageDescription, ImageType, ImageUsageFlags
    r = rng(80)
    for fmt, ch in ((F.R16_sFloat, 1), (F.RGBA16_sFloat, 4)):
        w, h, d = 9, 6, 5
        raw = pixfmt.pack_half(r.uniform(-2, 2, (d, h, w, ch)).astype(np.float32)).reshape(-1)
        gimg = backend.createImage(ImageDescription(width=w, height=h, depth=d, type=ImageType.Type3D, format=fmt, usageFlags=int(ImageUsageFlags.Sampled)), raw)
        oimg = orc.Img(raw, w, h, fmt, d=d)
        uv2 = _adversarial_uv(w, h, 81)
        z = r.permutation(np.concatenate([np.arange(-2, d + 3) / d, r.uniform(-1.5, 2.5, uv2.shape[0])]))[: uv2.shape[0]]
        uvw = np.concatenate([uv2, z[:, None]], 1).astype(np.float32)
        for filt, addr in ((0, 0), (0, 1), (0, 2), (0, 3), (1, 0), (1, 1)):
            a = backend.debugSamplerEval(gimg, filt, addr, uvw)
            b = orc.sampler_eval(oimg, filt, addr, uvw)
            assert np.array_equal(a.view(np.uint32), b.view(np.uint32)), (filt, addr)


# ------------------------------------------------------------------ fast sky LUT lookup against the oracle's
@pytest.mark.gpu
def test_gpu_fast_sky_lut_lookup_follows_the_oracle(backend):
    """sampleSkyLut (sky.inc:85-116) of the PLR_MATH_FAST set uses polynomial acos / atan (device/fastmath.h). Directions sweep the sphere, densely
    around the horizon (where the LUT's v coordinate is sqrt-steep) and the poles (where |V.y| can exceed 1 by an ulp): per channel within one
    R11G11B10 code of the oracle's lookup wherever the LUT is smooth; the synthetic LUT's 6.7x step between two rows is reported separately."""
    from plainrenderer_amd import synth
    r = rng(90)
    lut = synth.sky_lut()
    n = 400000
    v = r.normal(size=(n, 3))
    v[: n // 2, 1] *= 0.02                      # near the horizon
    v[n // 2: n // 2 + 2000, [0, 2]] *= 1e-4    # near the poles
    v /= np.linalg.norm(v, axis=1, keepdims=True)
    v = v.astype(np.float32)
    v[0], v[1], v[2], v[3] = (0, 1, 0), (0, -1, 0), (0, np.float32(1.0000001), 0), (1, 0, 0)
    gimg = backend.createImage(image_desc_2d(200, 100, F.R11G11B10_uFloat), lut)
    got = backend.debugSkyLutEval(gimg, v)
    ref = orc.kat_sky_lut(orc.Img(lut, 200, 100, F.R11G11B10_uFloat), v)
    assert np.isfinite(got).all()
    d = np.abs(pixfmt.pack_r11g11b10(got).astype(np.int64)[:, None] >> np.array([0, 11, 22]) & np.array([0x7ff, 0x7ff, 0x3ff]))
    d = np.abs(d - (pixfmt.pack_r11g11b10(ref).astype(np.int64)[:, None] >> np.array([0, 11, 22]) & np.array([0x7ff, 0x7ff, 0x3ff])))
    rel = np.abs(got - ref) / np.maximum(np.abs(ref), 1e-9)
    worst = np.argsort(-rel.max(1))[:5]
    print("sky lookup: max code diff %d, >1 code: %d of %d, max rel %.3g at V=%s (got %s ref %s)" % (d.max(), (d > 1).any(1).sum(), n, rel.max(), v[worst[0]], got[worst[0]], ref[worst[0]]))
    assert d.max() <= 1, "fast sky lookup differs from the oracle's by more than one R11G11B10 code"
