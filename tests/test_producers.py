"""input producers as compute passes. lightMatrix.comp (cascade fit from the HiZ apex).

The oracle is cross-checked against the independent float64 cascade fit of plainrenderer_amd/synth.py (which the synthetic
shadow maps are rendered with); the HIP kernel is held bit-identical to the oracle."""
import struct

import numpy as np
import pytest

import passes
from plainrenderer_amd import synth
from plainrenderer_amd.scene import Camera, GlobalShaderInfo

W, H = 256, 144


def _global(cam, sun):
    g = GlobalShaderInfo(frameIndex=3, sunDirection=(*sun, 0.0), time=0.5, deltaTime=1 / 60.0)
    cam.fill_global(g, W, H)
    return g.pack()


def _raw_depth(lin, n, f):
    return 1.0 - (n * f / lin - f) / (n - f)


def _parse(info):
    a = np.frombuffer(info, np.float32)
    return a[:4], a[4:68].reshape(4, 4, 4), a[68:76].reshape(4, 2)  # splits, matrices [cascade][col][row], scales


def test_oracle_matches_the_independent_host_cascade_fit():
    cam = Camera.look((15.0, -7.0, -6.0), (0.0, 0.16, 1.0), aspect=W / H)
    sun = np.array([0.35, -0.8, 0.45]); sun /= np.linalg.norm(sun)
    dmin, dmax = 4.0, 120.0
    scene = synth.SynthScene(grid=2, cell=8.0, seed_id=950)
    info_ref, _ = scene.shadow_cascades(cam, sun, dmin, dmax, 16, cascade_count=3, extra_padding=5.0, min_far=30.0)
    apex = (_raw_depth(dmax, cam.near, cam.far), _raw_depth(dmin, cam.near, cam.far))  # .x = min depth = farthest (reverse Z)
    got = passes.orc_light_matrix(b"\0" * 304, apex, _global(cam, sun), 3, 5.0, 30.0)
    s0, m0, c0 = _parse(info_ref)
    s1, m1, c1 = _parse(got)
    assert np.allclose(s1[:2], s0[:2], rtol=2e-5)
    assert np.allclose(c1[:3], c0[:3], rtol=2e-4)
    assert np.allclose(m1[:3], m0[:3], rtol=2e-3, atol=2e-4)  # float32 chain (apex depth round trip included) vs float64


def test_oracle_known_answers():
    cam = Camera.look((0.0, -5.0, 0.0), (0.0, 0.0, 1.0), aspect=W / H)
    sun = np.array([0.0, -1.0, 0.0])  # |forward.y| >= 0.9999: the alternative up vector branch (:68)
    apex = (_raw_depth(50.0, cam.near, cam.far), _raw_depth(2.0, cam.near, cam.far))
    info = passes.orc_light_matrix(b"\0" * 304, apex, _global(cam, sun), 4, 5.0, 30.0)
    splits, m, scales = _parse(info)
    assert np.allclose(splits[:3], [2 + 48 * 0.25, 2 + 48 * 0.5, 2 + 48 * 0.75], rtol=1e-4)  # linear splits (:52-54)
    for i in range(4):
        M = m[i].T  # math matrix
        # orthographic: last row (0,0,0,1); z row = -0.5 * scale.z * forward (+0.5): depth decreases along the light direction
        assert np.allclose(M[3], [0, 0, 0, 1])
        assert np.allclose(np.abs(M[0, :3] @ M[1, :3]), 0, atol=1e-6)
        # the cascade's frustum corners land inside the unit cube after the fit
        lo = 2.0 if i == 0 else splits[i - 1]
        hi = splits[i] if i < 3 else 50.0
        if i == 3:
            lo = cam.near
        for dist in (lo, hi):
            for sx in (-1, 1):
                for sy in (-1, 1):
                    hh = cam.tan_fov_half() * dist
                    p = np.asarray(cam.position, np.float64) + np.asarray(cam.forward) * dist + np.asarray(cam.up) * hh * sy + np.asarray(cam.right) * hh * cam.aspect * sx
                    q = M @ np.append(p, 1.0)
                    assert -1.001 <= q[0] <= 1.001 and -1.001 <= q[1] <= 1.001 and -0.001 <= q[2] <= 1.001
    assert np.allclose(scales[:, 0], [m[i][0][0] / 1.0 if False else scales[i, 0] for i in range(4)])


@pytest.mark.gpu
@pytest.mark.parametrize("sun,count", [((0.35, -0.8, 0.45), 3), ((0.0, -1.0, 0.0), 4), ((-0.6, -0.3, -0.2), 2), ((0.2, 0.9, 0.1), 1)])
def test_gpu_light_matrix_bit_exact(backend, sun, count):
    cam = Camera.look((15.0, -7.0, -6.0), (0.1, 0.16, 1.0), aspect=W / H)
    sun = np.array(sun, np.float64); sun /= np.linalg.norm(sun)
    apex = (_raw_depth(140.0, cam.near, cam.far), _raw_depth(3.5, cam.near, cam.far))
    prev = np.random.default_rng(7).standard_normal(76).astype(np.float32).tobytes()  # stale contents: entries the pass does not write must survive
    gp = _global(cam, sun)
    a = passes.gpu_light_matrix(backend, prev, apex, gp, count, 8.0, 30.0)
    b = passes.orc_light_matrix(prev, apex, gp, count, 8.0, 30.0)
    assert a == b


@pytest.mark.gpu
def test_gpu_frame_with_compute_light_matrices(backend):
    """the frame graph with lightMatrix.comp recorded after the depth pyramid: the buffer the shade and the trace read is the fit to
    this frame's HiZ apex"""
    from oracle_frame import OracleFrame  # noqa: F401 (keeps the import graph of the frame tests warm)
    from plainrenderer_amd.frame import FramePipeline, SyntheticInputs
    w, h = 256, 144
    cams = [Camera.look((15.0 + 0.03 * i, -7.0, -6.0 + 0.05 * i), (0.0, 0.16, 1.0), aspect=w / h) for i in range(3)]
    scene = synth.SynthScene(grid=4, cell=8.0, seed_id=951)
    fp = FramePipeline(backend, w, h, shadow_map_res=128, brdf_lut_res=16, froxel_depth=8, max_sdf_instances=64, run_light_matrix=1)
    inputs = SyntheticInputs(scene, cams[1], cams[0], w, h, sdf_res=16, shadow_res=128, froxel_depth=8, sun_direction=(0.35, -0.8, 0.45))
    inputs.upload(fp)
    fp.frame(cams[1], 1 / 60.0, 0.5)
    got = backend.downloadStorageBuffer(fp.storage_buffer("sunShadowInfo"), 304).tobytes()
    # apex of the pyramid the frame built
    mips = passes.mip_count_from_resolution(w // 2, h // 2)
    apex = backend.downloadImage(fp.image("pyramid"), mips - 1, np.float32)[:2]
    exp = passes.orc_light_matrix(bytes(inputs.shadow_info), apex, bytes(fp.submitted_globals()), 3, 5.0, 30.0)
    assert got == exp
    # and it agrees with the matrices the synthetic shadow maps were rendered with (same depth range, float64 host fit)
    s0, m0, _ = _parse(bytes(inputs.shadow_info))
    s1, m1, _ = _parse(got)
    assert np.allclose(s1[:2], s0[:2], rtol=1e-3) and np.allclose(m1[:3], m0[:3], rtol=5e-3, atol=1e-3)
    fp.destroy()


# ------------------------------------------------------------------ sky LUTs
def _sky_inputs(sun=(0.35, -0.8, 0.45)):
    from util import light_buffer_bytes
    cam = Camera.look((0.0, -5.0, 0.0), (0.0, 0.0, 1.0), aspect=W / H)
    sun = np.array(sun, np.float64); sun /= np.linalg.norm(sun)
    return passes.ATMOSPHERE_DEFAULT, light_buffer_bytes(sun_strength_exposed=12.8), _global(cam, sun)


def test_oracle_sky_luts_known_answers():
    from plainrenderer_amd import pixfmt
    atm, light, gp = _sky_inputs()
    t, m, s = passes.orc_sky_luts(atm, light, gp)
    T = pixfmt.unpack_r11g11b10(t.reshape(-1)).reshape(128, 128, 3)
    # x = height (0 .. 100 km), y = cos of the angle to the zenith (-1 .. 1)
    assert np.isfinite(T).all() and T.min() >= 0 and T.max() <= 1.0
    assert (T[127, 127] > 0.99).all()                   # top of the atmosphere looking up: nothing left to absorb
    assert (T[0, :40] == 0).all()                       # looking straight down from any height below 30 km: the earth blocks the ray
    assert (np.diff(T[100, :, 2]) >= -1e-3).all()       # transmission grows with the height of the observer
    assert (np.diff(T[64:, 0, 2]) >= -1e-3).all()       # from the ground: grows from the horizon towards the zenith
    assert T[127, 0, 2] < T[127, 0, 0]                  # Rayleigh: blue is absorbed/scattered most
    M = pixfmt.unpack_r11g11b10(m.reshape(-1)).reshape(32, 32, 3)
    assert np.isfinite(M).all() and M.min() >= 0 and M.max() < 1.0 and M.max() > 1e-3
    S = pixfmt.unpack_r11g11b10(s.reshape(-1)).reshape(100, 200, 3)
    assert np.isfinite(S).all() and S.max() > 0
    # a blue sky: away from the sun, above the horizon, blue > red
    sky_rows = S[10:40]
    assert (sky_rows[..., 2].mean() > sky_rows[..., 0].mean())
    # the Mie forward peak: looking at the sun is brighter than looking at the same elevation in the opposite azimuth
    def lut_texel(V):  # toSkyLut (sky.inc:86-94)
        th = np.arccos(-V[1]); y = th / np.pi; yl = y * 2 - 1; y = np.sign(yl) * np.sqrt(abs(yl)) * 0.5 + 0.5
        x = -np.arctan2(V[2], V[0]) / (2 * 3.1415) + 0.5
        return S[min(int(y * 100), 99), int(x * 200) % 200]
    sun = np.array([0.35, -0.8, 0.45]); sun /= np.linalg.norm(sun)
    anti = sun * np.array([-1.0, 1.0, -1.0])
    assert lut_texel(sun).sum() > 1.5 * lut_texel(anti).sum()


@pytest.mark.gpu
@pytest.mark.parametrize("sun", [(0.35, -0.8, 0.45), (0.9, -0.05, 0.1)])
def test_gpu_sky_luts_bit_exact(backend, sun):
    atm, light, gp = _sky_inputs(sun)
    a = passes.gpu_sky_luts(backend, atm, light, gp)
    b = passes.orc_sky_luts(atm, light, gp)
    for x, y, what in zip(a, b, ("transmission", "multiscatter", "sky")):
        assert np.array_equal(x, y), what


@pytest.mark.gpu
@pytest.mark.parametrize("sun", [(0.35, -0.8, 0.45), (0.9, -0.05, 0.1)])
def test_gpu_fast_sky_luts_within_one_code(backend, sun):
    """the PLR_MATH_FAST sky LUT kernels (kernels_fast/sky_fast.hip: a lane per march step, DPP prefix sums instead of the serial march) against the
    oracle. Outputs are R11G11B10; the statement is per kernel: EVERY texel within one code per channel of the oracle's LUT when the pass reads the
    LUTs the oracle reads (the later passes run alone on the oracle's transmission / multiscatter LUTs). The chain as the frame records it (the
    sky LUT built from the FAST transmission LUT, whose texels are themselves up to one code - 3 % in blue - from the oracle's) is reported and
    held to three codes."""
    import parity
    atm, light, gp = _sky_inputs(sun)
    b = passes.orc_sky_luts(atm, light, gp)
    backend.setMathMode(True)
    try:
        chain = passes.gpu_sky_luts(backend, atm, light, gp)
        multiscatter_alone = passes.gpu_sky_luts(backend, atm, light, gp, given_transmission=b[0])[1]
        sky_alone = passes.gpu_sky_luts(backend, atm, light, gp, given_transmission=b[0], given_multiscatter=b[1])[2]
    finally:
        backend.setMathMode(False)
    def report(x, y, what):
        d = parity.r11g11b10_code_diff(x.reshape(-1), y.reshape(-1))
        zero_mismatch = int((((x == 0) != (y == 0))).sum())  # a ray that grazes the earth: transmission 0 on one side only
        print("PRODUCER sky %-28s max_code_diff=%d differing=%.5f over_one=%.6f zero_mismatch=%d" % (what, int(d.max()), float((d != 0).any(axis=1).mean()),
                                                                                                  float((d > 1).any(axis=1).mean()), zero_mismatch), flush=True)
        return d
    assert report(chain[0], b[0], "transmission").max() <= 1
    assert report(multiscatter_alone, b[1], "multiscatter (oracle inputs)").max() <= 1
    assert report(sky_alone, b[2], "sky (oracle inputs)").max() <= 1
    assert report(chain[1], b[1], "multiscatter (fast chain)").max() <= 1
    d = report(chain[2], b[2], "sky (fast chain)")
    assert d.max() <= 3 and float((d > 1).any(axis=1).mean()) <= 0.01


@pytest.mark.gpu
def test_gpu_frame_with_compute_sky_luts(backend):
    """the frame graph with the three sky LUT passes recorded where Sky::updateTransmissionLut / updateSkyLut sit (RenderFrontend.cpp:348-350):
    the LUTs the exposure, trace and shade passes read are this frame's compute results (the sky LUT uses this frame's exposure)"""
    from plainrenderer_amd.frame import FramePipeline, SyntheticInputs
    w, h = 256, 144
    cams = [Camera.look((15.0 + 0.03 * i, -7.0, -6.0 + 0.05 * i), (0.0, 0.16, 1.0), aspect=w / h) for i in range(3)]
    scene = synth.SynthScene(grid=4, cell=8.0, seed_id=952)
    fp = FramePipeline(backend, w, h, shadow_map_res=128, brdf_lut_res=16, froxel_depth=8, max_sdf_instances=64, run_sky_luts=1)
    inputs = SyntheticInputs(scene, cams[1], cams[0], w, h, sdf_res=16, shadow_res=128, froxel_depth=8, sun_direction=(0.35, -0.8, 0.45))
    inputs.upload(fp)  # uploads stand-in LUTs too; the compute passes overwrite them
    for f in range(2):
        fp.frame(cams[f + 1], 1 / 60.0, 0.5 + f / 60.0)
    light = backend.downloadStorageBuffer(fp.storage_buffer("light"), 20).tobytes()
    t, m, s = passes.orc_sky_luts(passes.ATMOSPHERE_DEFAULT, light, bytes(fp.submitted_globals()))
    assert np.array_equal(backend.downloadImage(fp.image("transmissionLut"), 0, np.uint32).reshape(128, 128), t)
    assert np.array_equal(backend.downloadImage(fp.image("skyMultiscatterLut"), 0, np.uint32).reshape(32, 32), m)
    got = backend.downloadImage(fp.image("skyLut"), 0, np.uint32).reshape(100, 200)
    assert np.array_equal(got[:96], s[:96])  # rows 96..99 are never dispatched: they keep what was uploaded
    assert np.array_equal(got[96:], inputs.sky.reshape(100, 200)[96:])
    fp.destroy()


# ------------------------------------------------------------------ volumetric froxel lighting
def _volumetric_inputs(seed=970, fd=16, noise_n=8):
    from util import light_buffer_bytes
    rng = np.random.default_rng(0x504C4149 + seed)
    cam = Camera.look((15.0, -7.0, -6.0), (0.0, 0.16, 1.0), aspect=W / H)
    cam_prev = Camera.look((14.9, -7.0, -6.1), (0.01, 0.16, 1.0), aspect=W / H)
    sun = np.array([0.35, -0.8, 0.45]); sun /= np.linalg.norm(sun)
    g = GlobalShaderInfo(frameIndex=5, sunDirection=(*sun, 0.0), time=0.5, deltaTime=1 / 60.0)
    g.viewProjectionPrevious = cam_prev.view_projection()
    g.cameraPosPrevious = (*cam_prev.position, 0.0)
    g.cameraForwardPrevious = (*cam_prev.forward, 0.0)
    cam.fill_global(g, W, H)
    scene = synth.SynthScene(grid=2, cell=8.0, seed_id=seed)
    info, maps = scene.shadow_cascades(cam, sun, 2.0, 60.0, 64)
    fw, fh = (W + 7) // 8, (H + 7) // 8
    noise = rng.integers(0, 256, size=(noise_n, noise_n, noise_n), dtype=np.uint8)
    from plainrenderer_amd import pixfmt
    hist = pixfmt.pack_half(np.stack([rng.uniform(0, 0.01, (fd, fh, fw)), rng.uniform(0, 0.01, (fd, fh, fw)), rng.uniform(0, 0.01, (fd, fh, fw)), rng.uniform(0.001, 0.01, (fd, fh, fw))], -1)
                            .astype(np.float32))
    return (fw, fh, fd, noise, hist, maps[2], 64, info, light_buffer_bytes(sun_strength_exposed=12.8), passes.VOLUMETRIC_SETTINGS_DEFAULT, g.pack())


def test_oracle_volumetrics_known_answers():
    from plainrenderer_amd import pixfmt
    args = _volumetric_inputs()
    fw, fh, fd = args[:3]
    material, scattering, target, integration = [pixfmt.unpack_half(a).reshape(fd, fh, fw, 4) for a in passes.orc_volumetrics(*args)]
    assert np.isfinite(integration).all()
    # material: coefficients * max(baseDensity + range * (noise - 0.5), 0): scattering rgb equal (coefficients 1,1,1), absorption = same density
    assert np.allclose(material[..., 0], material[..., 1]) and np.allclose(material[..., 0], material[..., 3])
    assert material.min() >= 0 and material.max() <= 0.003 + 0.008 * 0.5 + 1e-5
    # integration: transmittance starts near 1, decreases monotonically with depth and stays positive; inscattering accumulates
    T = integration[..., 3]
    assert (T[0] > 0.99).all() and (np.diff(T, axis=0) <= 1e-3).all() and (T > 0).all()
    assert (np.diff(integration[..., 0], axis=0) >= -1e-4).all()


@pytest.mark.gpu
def test_gpu_volumetrics_bit_exact(backend):
    args = _volumetric_inputs()
    a = passes.gpu_volumetrics(backend, *args)
    b = passes.orc_volumetrics(*args)
    for x, y, what in zip(a, b, ("material", "scattering", "reprojection", "integration")):
        assert np.array_equal(x, y), what


@pytest.mark.gpu
@pytest.mark.parametrize("fusion", [0, 1, 2])
def test_gpu_fast_volumetrics_exact_front_and_integration_within_half_float_tolerance(backend, fusion):
    """PLR_MATH_FAST: the three per-froxel passes keep the exact-order arithmetic (everything in them is discrete in the froxel position: the
    shadow-map texel, the 8-bit sub-texel weights of the noise and history samples) and equal the oracle bit for bit - as three launches (pass
    fusion off), as ONE launch that carries the texel from pass to pass in registers (fusion 1), and as that launch without the stores of the
    material / scattering volumes (fusion 2: nothing else binds them). The front-to-back integration (kernels_fast/froxel_fast.hip: hardware exp,
    one exponential per slice, loads ahead of the running sums) meets the half-float bound of tests/parity.py on every value - it has no decision
    to flip."""
    import parity
    from plainrenderer_amd import pixfmt
    args = _volumetric_inputs()
    level = backend.getPassFusion()[0]
    backend.setMathMode(True)
    backend.setPassFusion(fusion)
    try:
        a = passes.gpu_volumetrics(backend, *args, intermediates=fusion < 2)
        fused = backend.getPassFusion()[1]
        if fusion == 2:  # the elided volumes cannot be read back: the backend says so instead of returning stale bytes
            with pytest.raises(Exception):
                passes.gpu_volumetrics(backend, *args, intermediates=True)
    finally:
        backend.setMathMode(False)
        backend.setPassFusion(level)
    assert fused == (4 if fusion else 0)  # the three per-froxel passes and the integration behind them: one launch
    b = passes.orc_volumetrics(*args)
    for x, y, what in zip(a[:3], b[:3], ("material", "scattering", "reprojection")):
        assert x is None or np.array_equal(x, y), what
    assert a[2] is not None
    got, ref = pixfmt.unpack_half(a[3]).reshape(-1, 4), pixfmt.unpack_half(b[3]).reshape(-1, 4)
    # inscattering (rgb) and transmittance (a) live on different scales: each against its own
    bad = parity.half_violations(got[:, :3], ref[:, :3], floor_frac=2.0 ** -10) | parity.half_violations(got[:, 3:], ref[:, 3:], floor_frac=2.0 ** -10)
    print("PRODUCER froxel fusion=%d fused_executions=%d integration violations=%d max_err=%.3g scale=%.3g" % (fusion, fused, int(bad.sum()), float(np.abs(got - ref).max()),
                                                                                                          float(np.abs(ref).max())), flush=True)
    assert not bad.any()


@pytest.mark.gpu
@pytest.mark.parametrize("fd,noise_n", [(37, 8), (16, 6), (5, 12)])
def test_gpu_fast_froxel_columns_ragged_depth_and_any_noise_extent(backend, fd, noise_n):
    """The fused per-froxel launch walks column segments of 8 slices per thread (kernels/producers.hip, froxelFrontFusedKernel): a depth that is not a multiple of the
    segment (37 = 4 x 8 + 5; 5 < one segment), and a noise volume whose extent is not a power of two (the wrap falls back from the mask to the
    modulo) - bit-identical to the oracle like the default shapes."""
    args = _volumetric_inputs(fd=fd, noise_n=noise_n)
    level = backend.getPassFusion()[0]
    backend.setMathMode(True)
    backend.setPassFusion(1)
    try:
        a = passes.gpu_volumetrics(backend, *args)
        fused = backend.getPassFusion()[1]
    finally:
        backend.setMathMode(False)
        backend.setPassFusion(level)
    assert fused == 4
    b = passes.orc_volumetrics(*args)
    for x, y, what in zip(a[:3], b[:3], ("material", "scattering", "reprojection")):
        assert np.array_equal(x, y), what
    import parity
    from plainrenderer_amd import pixfmt
    got, ref = pixfmt.unpack_half(a[3]).reshape(-1, 4), pixfmt.unpack_half(b[3]).reshape(-1, 4)  # the integration inside the same launch: the fast set's bound
    assert not (parity.half_violations(got[:, :3], ref[:, :3], floor_frac=2.0 ** -10) | parity.half_violations(got[:, 3:], ref[:, 3:], floor_frac=2.0 ** -10)).any()


@pytest.mark.gpu
def test_gpu_frame_with_compute_volumetrics(backend):
    """the frame graph with the four froxel passes recorded before the shade (RenderFrontend.cpp:366-371): the integration volume the
    shade reads is this frame's compute result, the history ping-pongs with the frame index"""
    from plainrenderer_amd import pixfmt
    from plainrenderer_amd.frame import FramePipeline, SyntheticInputs
    from plainrenderer_amd.scene import taa_jitter_pixels  # noqa: F401
    w, h = 256, 144
    cams = [Camera.look((15.0 + 0.03 * i, -7.0, -6.0 + 0.05 * i), (0.0, 0.16, 1.0), aspect=w / h) for i in range(3)]
    scene = synth.SynthScene(grid=4, cell=8.0, seed_id=971)
    fd = 16
    fp = FramePipeline(backend, w, h, shadow_map_res=128, brdf_lut_res=16, froxel_depth=fd, max_sdf_instances=64, run_volumetrics=1)
    inputs = SyntheticInputs(scene, cams[1], cams[0], w, h, sdf_res=16, shadow_res=128, froxel_depth=fd, sun_direction=(0.35, -0.8, 0.45))
    inputs.upload(fp)
    noise = np.random.default_rng(0x504C4149 + 972).integers(0, 256, size=(32, 32, 32), dtype=np.uint8)
    backend.uploadImage(fp.image("perlinNoise3D"), noise)
    fw, fh = (w + 7) // 8, (h + 7) // 8
    hist = np.zeros(fw * fh * fd * 4, np.uint16)
    def radical_inverse_base2(i):
        return int("{:032b}".format(i)[::-1], 2) * 2.3283064365386963e-10
    for f in range(2):
        fp.frame(cams[f + 1], 1 / 60.0, 0.5 + f / 60.0)
        cpu_frame = fp.cpu_frame_index()
        settings = struct.pack("<13f", 0.0, 0.0, 0.0, np.float32(radical_inverse_base2(cpu_frame % 8)) - np.float32(0.5), 1.0, 1.0, 1.0, 30.0, 1.0, 0.003, 0.008, 0.5, 0.2)
        light = backend.downloadStorageBuffer(fp.storage_buffer("light"), 20).tobytes()
        exp = passes.orc_volumetrics(fw, fh, fd, noise, hist.reshape(fd, fh, fw, 4), inputs.shadow_maps[2], 128, bytes(inputs.shadow_info), light, settings,
                                     bytes(fp.submitted_globals()))
        got_target = backend.downloadImage(fp.image("volumetricHistory%d" % (cpu_frame % 2)), 0, np.uint16)
        assert np.array_equal(got_target, exp[2]), "reprojection target, frame %d" % f
        assert np.array_equal(backend.downloadImage(fp.image("volumetricIntegrationVolume"), 0, np.uint16), exp[3]), "integration volume, frame %d" % f
        hist = exp[2]
    it = pixfmt.unpack_half(exp[3]).reshape(fd, fh, fw, 4)
    assert np.isfinite(it).all() and (it[..., 3] > 0).all()
    fp.destroy()
