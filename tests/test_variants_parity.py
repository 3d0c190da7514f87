"""Every variant the PLR_MATH_FAST set SHIPS, not only the benchmarked default, is held to the storage-quantum statement of tests/parity.py
(VERDICT r02 #8): the 4 diffuse BRDFs x 4 direct multiscatter modes of the deferred shade (+ geometric AA off, ambient-only indirect light, 1..4
cascades), the five TAA history samplers x clip / clamp (+ no dilation, no tonemapping), the SDF trace without the strict influence cut-off and
the spatial filter on a full-resolution (D32) grid - at 1920 x 1088 on bench.py's scene (RenderFrontend.h:32-38, Techniques/TAA.h:8-17).
Pixels whose decision signatures agree with the oracle's must meet the bound with NO outlier allowance; flipped decisions are counted under caps.
PLR_VARIANT_SIZE=WxH (multiples of 64) changes the size."""
import os

import numpy as np
import pytest

import parity
import passes
import test_parity_fullsize as full
from plainrenderer_amd import pixfmt
from util import F

W, H = (int(v) for v in os.environ.get("PLR_VARIANT_SIZE", "1920x1088").split("x"))
TW, TH = W // 2, H // 2
U = pixfmt.unpack_half


def report(name, **kv):
    print("VARIANT %-22s %s" % (name, " ".join("%s=%s" % (k, ("%.6g" % v) if isinstance(v, float) else v) for k, v in kv.items())), flush=True)


@pytest.fixture(scope="module")
def vs(backend):
    s = full.build_state(backend, W, H)
    yield s
    s.fp.destroy()
    backend.setMathMode(False)


_LUTS = {}
SHADE_VARIANTS = [(b, m, True, 0, 3) for b in range(4) for m in range(4)] + [(2, 0, False, 0, 3), (2, 0, True, 1, 3), (1, 2, False, 1, 4), (3, 3, True, 0, 1), (0, 1, True, 0, 2)]


@pytest.mark.gpu
@pytest.mark.parametrize("brdf,multi,aa,tech,cascades", SHADE_VARIANTS)
def test_gpu_every_shipped_shade_variant_meets_the_quantum_bound(backend, vs, brdf, multi, aa, tech, cascades):
    c = vs.cap["shade"]
    if brdf not in _LUTS:
        _LUTS[brdf] = vs.ora.brdf_lut if brdf == int(vs.settings.diffuse_brdf) else passes.orc_brdf_lut(512, brdf)
    lut = _LUTS[brdf]
    args = (vs.gb, W, H, lut, 512, c["light"], vs.inputs.shadow_info, vs.inputs.shadow_maps, vs.inputs.shadow_res, c["gi"][0], c["gi"][1], vs.inputs.froxel,
            vs.inputs.froxel_dims, vs.inputs.vol_settings, vs.inputs.sky, vs.gp)
    var = (brdf, multi, aa, tech, cascades)
    with passes.gpu_signature(backend, W * H) as sg:
        got = passes.gpu_deferred_shading(backend, *args, *var)
    arr, n = vs.ora._bindless(passes.orc.global_from_bytes(vs.gp))
    with passes.orc_signature(W * H) as so:
        ref = passes.orc_deferred_shading(*args, arr, n, *var)
    flip = sg.words != so.words
    d = parity.r11g11b10_code_diff(got, ref)
    sky = (so.words & 128) != 0
    report("shade %d/%d/%d/%d/%d" % var, flipped=float(flip.mean()), clean_max_code_diff=int(d[~flip & ~sky].max()), sky_max_code_diff=int(d[~flip & sky].max(initial=0)),
           clean_differing=float((d[~flip] != 0).any(axis=1).mean()))
    assert d[~flip & ~sky].max() <= 1, "same cascade, same number of lit PCF taps: every channel within one R11G11B10 code"
    assert d[~flip & sky].max(initial=0) <= 2
    assert flip.mean() <= 2e-3


TAA_VARIANTS = [(clip, True, tech, True) for tech in range(5) for clip in (True, False)] + [(True, False, 4, True), (True, True, 4, False), (False, False, 0, False), (True, False, 2, False)]


@pytest.mark.gpu
@pytest.mark.parametrize("clip,dilate,tech,tonemap", TAA_VARIANTS)
def test_gpu_every_shipped_taa_variant_meets_the_quantum_bound(backend, vs, clip, dilate, tech, tonemap):
    c = vs.cap["taa"]
    args = (c["inp"], c["history"], vs.gb["motion"], vs.gb["depth"], W, H, c["weights"], vs.gp, clip, dilate, tech, tonemap)
    og, hg = passes.gpu_taa(backend, *args)
    oo, ho = passes.orc_taa(*args)
    d = parity.r11g11b10_code_diff(og, oo)
    report("taa %d/%d/%d/%d" % (clip, dilate, tech, tonemap), max_code_diff=int(d.max()), differing=float((d != 0).any(axis=1).mean()), over_one=float((d > 1).any(axis=1).mean()))
    assert np.array_equal(og, hg)
    # the resolve has no discrete decision a kernel could take differently from the oracle: every channel of every pixel within one code
    # (measured on MI355X: at most 1e-5 of the pixels differ at all, profiles/r03_variants_parity.txt)
    assert d.max() <= 1


@pytest.mark.gpu
def test_gpu_trace_without_the_strict_cutoff_meets_the_bound(backend, vs):
    c = vs.cap["trace"]
    args = (vs.gb["depth"], vs.gb["normal"], W, H, TW, TH, vs.inputs.sky, 200, 100, c["light"], vs.inputs.instance_bytes_patched, c["tiles"], 5.0, vs.inputs.shadow_info,
            vs.inputs.shadow_maps[c["cascade"]], vs.inputs.shadow_res, vs.gp)
    with passes.gpu_signature(backend, TW * TH) as sg:
        yg, cg = passes.gpu_sdf_trace(backend, *args, strict=False, cascade=c["cascade"])
    arr, n = vs.ora._bindless(passes.orc.global_from_bytes(vs.gp))
    with passes.orc_signature(TW * TH) as so:
        yo, co = passes.orc_sdf_trace(*args, arr, n, strict=False, cascade=c["cascade"])
    ray_flip = ((sg.words ^ so.words) & ~np.uint32(0x7F8)).reshape(TH, TW) != 0
    take_flip = ((sg.words ^ so.words) & np.uint32(0x7F8)).reshape(TH, TW) != 0
    touched = (parity.dilate3x3(ray_flip) | take_flip).reshape(-1)
    got = np.concatenate([U(yg).reshape(-1, 4), U(cg).reshape(-1, 2)], axis=1)
    ref = np.concatenate([U(yo).reshape(-1, 4), U(co).reshape(-1, 2)], axis=1)
    bad = parity.half_violations(got, ref, floor_frac=2.0 ** -10)
    report("trace strict=0", rays_flipped=float(ray_flip.mean()), take_flipped=float(take_flip.mean()), clean_violations=int((bad & ~touched).sum()))
    assert not (bad & ~touched).any()
    assert ray_flip.mean() <= 2e-5 and take_flip.mean() <= 2e-5


@pytest.mark.gpu
@pytest.mark.parametrize("filter_index", [0, 1])
def test_gpu_spatial_filter_on_a_full_resolution_grid_meets_the_bound(backend, vs, filter_index):
    """the trace at full resolution (SDFTraceSettings::halfResTrace = false) filters on the D32 depth buffer: the unpacked three-gather kernel"""
    c = vs.cap["spatial%d" % filter_index]
    yf = np.repeat(np.repeat(np.asarray(c["inp"][0]).reshape(TH, TW, 4), 2, 0), 2, 1)
    cf = np.repeat(np.repeat(np.asarray(c["inp"][1]).reshape(TH, TW, 2), 2, 0), 2, 1)
    args = (yf, cf, W, H, vs.gb["depth"], F.Depth32, W, H, vs.gb["normal"], W, H, vs.gp, filter_index)
    with passes.gpu_signature(backend, 2 * W * H) as sg:
        yg, cg = passes.gpu_gi_spatial(backend, *args)
    with passes.orc_signature(2 * W * H) as so:
        yo, co = passes.orc_gi_spatial(*args)
    xw = (sg.words ^ so.words).reshape(-1, 2)
    x = xw[:, 0] | xw[:, 1]
    flipped = np.zeros(x.size, np.int32)
    for b in range(32):
        flipped += ((x >> np.uint32(b)) & np.uint32(1)).astype(np.int32)
    clean = flipped == 0
    got = np.concatenate([U(yg).reshape(-1, 4), U(cg).reshape(-1, 2)], axis=1)
    ref = np.concatenate([U(yo).reshape(-1, 4), U(co).reshape(-1, 2)], axis=1)
    bad = parity.half_violations(got, ref, floor_frac=2.0 ** -10)
    report("spatial%d full-res" % filter_index, sample_flip_rate=float(flipped.sum() / (32.0 * x.size)), clean_violations=int((bad & clean).sum()))
    assert not (bad & clean).any()
    assert flipped.sum() <= 1e-3 * 32 * x.size



class _GlobalBufferTheHostCannotRead(passes.GlobalBinding):
    """A global uniform buffer that was filled - and the fill flushed by a frame - BEFORE it became the global one: the backend's host copy of the global
    data (PassCtx::globalHost) does not cover it, so the launchers cannot resolve the frame's noise texture on the host (PassCtx::hostNoiseView)."""

    def set(self, packed340):
        be = self.be
        ubo = be.createUniformBuffer(340)
        be.setUniformBufferData(ubo, packed340)
        be.newFrame()
        be.prepareForDrawcallRecording()
        be.renderFrame()  # an empty frame: flushes the fill while the other buffer is still the global one
        be.setGlobalDescriptorSetResources(passes.RenderPassResources(uniformBuffers=[passes.UniformBufferResource(ubo, 0)]))


@pytest.mark.gpu
def test_gpu_a_global_buffer_unknown_to_the_host_gets_the_general_kernels_loudly(backend, vs, monkeypatch):
    """the fast shade and trace kernels take the noise texture's view from the host. When the host does not know the global buffer's
    contents they do not chase the pointers in a branch nothing tests: the launchers hand the execution to the general kernels, the backend counts
    it (plr_get_general_kernel_executions), and the results are the exact set's bit for bit."""
    c = vs.cap["shade"]
    shade_args = (vs.gb, W, H, vs.ora.brdf_lut, 512, c["light"], vs.inputs.shadow_info, vs.inputs.shadow_maps, vs.inputs.shadow_res, c["gi"][0], c["gi"][1], vs.inputs.froxel,
                  vs.inputs.froxel_dims, vs.inputs.vol_settings, vs.inputs.sky, vs.gp)
    var = (int(vs.settings.diffuse_brdf), 0, True, 0, 3)
    t = vs.cap["trace"]
    trace_args = (vs.gb["depth"], vs.gb["normal"], W, H, TW, TH, vs.inputs.sky, 200, 100, t["light"], vs.inputs.instance_bytes_patched, t["tiles"], 5.0, vs.inputs.shadow_info,
                  vs.inputs.shadow_maps[t["cascade"]], vs.inputs.shadow_res, vs.gp)
    backend.setMathMode(False)
    shade_exact = passes.gpu_deferred_shading(backend, *shade_args, *var)
    trace_exact = passes.gpu_sdf_trace(backend, *trace_args, strict=True, cascade=t["cascade"])
    backend.setMathMode(True)
    try:
        shade_fast = passes.gpu_deferred_shading(backend, *shade_args, *var)
        assert backend.getGeneralKernelExecutions()[0] == 0, backend.getGeneralKernelExecutions()
        assert not np.array_equal(shade_fast, shade_exact), "the two kernel sets round differently somewhere on a 2-Mpixel frame"
        unknown = _GlobalBufferTheHostCannotRead(backend)
        monkeypatch.setattr(passes, "global_binding", lambda be: unknown)
        shade_got = passes.gpu_deferred_shading(backend, *shade_args, *var)
        n, names = backend.getGeneralKernelExecutions()
        assert n == 1 and "shad" in names.lower(), (n, names)
        trace_got = passes.gpu_sdf_trace(backend, *trace_args, strict=True, cascade=t["cascade"])
        n, names = backend.getGeneralKernelExecutions()
        assert n == 1 and "trace" in names.lower(), (n, names)
    finally:
        backend.setMathMode(True)  # the module's mode (build_state); the fixture's teardown switches it off
    assert np.array_equal(shade_got, shade_exact)
    assert np.array_equal(trace_got[0], trace_exact[0]) and np.array_equal(trace_got[1], trace_exact[1])
