"""CPU suite: detmath accuracy, pixel codecs vs numpy, C-ABI symbol export (no compute calls, no GPU)."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from util import ROOT, pixfmt


def _ulp_err(got, ref64):
    ref32 = ref64.astype(np.float32)
    sp = np.spacing(np.abs(ref32)).astype(np.float64)
    return np.abs(got.astype(np.float64) - ref64) / np.maximum(sp, 1e-45)


def test_detmath_accuracy(oracle):
    r = np.random.default_rng(7)
    x = np.exp(r.uniform(-80, 80, 200000)).astype(np.float32)
    assert _ulp_err(oracle.math_eval(0, x), np.log(x.astype(np.float64))).max() < 2.0
    assert _ulp_err(oracle.math_eval(1, x), np.log2(x.astype(np.float64))).max() < 3.0
    x = r.uniform(-87, 88, 200000).astype(np.float32)
    assert _ulp_err(oracle.math_eval(2, x), np.exp(x.astype(np.float64))).max() < 2.5
    x = r.uniform(-126, 127, 200000).astype(np.float32)
    assert _ulp_err(oracle.math_eval(3, x), np.exp2(x.astype(np.float64))).max() < 2.5
    x = r.uniform(-50, 50, 200000).astype(np.float32)
    assert np.abs(oracle.math_eval(5, x) - np.sin(x.astype(np.float64))).max() < 3e-7
    assert np.abs(oracle.math_eval(6, x) - np.cos(x.astype(np.float64))).max() < 3e-7
    x = r.uniform(-1, 1, 200000).astype(np.float32)
    assert np.abs(oracle.math_eval(7, x) - np.arccos(x.astype(np.float64))).max() < 6e-7
    y = r.uniform(-5, 5, 200000).astype(np.float32)
    x = r.uniform(-5, 5, 200000).astype(np.float32)
    assert np.abs(oracle.math_eval(8, y, x) - np.arctan2(y.astype(np.float64), x.astype(np.float64))).max() < 6e-7


def test_detmath_specials(oracle):
    v = oracle.math_eval(0, np.array([0.0, -1.0, np.inf, 1.0], np.float32))
    assert v[0] == -np.inf and np.isnan(v[1]) and v[2] == np.inf and v[3] == 0.0
    # exact powers of two through log2 / exp2
    k = np.arange(-20, 21).astype(np.float32)
    assert np.array_equal(oracle.math_eval(1, np.exp2(k)), k)
    assert np.array_equal(oracle.math_eval(3, k), np.exp2(k))
    assert oracle.math_eval(4, np.array([0.0], np.float32), np.array([5.0], np.float32))[0] == 0.0
    assert oracle.math_eval(7, np.array([1.0000001, -1.0000001], np.float32)).tolist() == pytest.approx([0.0, np.pi], abs=1e-6)


def test_half_codec_matches_numpy(oracle):
    r = np.random.default_rng(3)
    f = np.concatenate([np.exp(r.uniform(-25, 12, 100000)) * r.choice([-1, 1], 100000), [0, -0.0, 65504, 65520, 1e9, 6e-8, 3e-8, 2.98e-8]]).astype(np.float32)
    h = oracle.codec_eval(2, f, f.size, np.uint16, f.size)
    assert np.array_equal(h, pixfmt.pack_half(f))
    allh = np.arange(65536, dtype=np.uint16)
    b = oracle.codec_eval(3, allh, allh.size, np.float32, allh.size)
    ref = pixfmt.unpack_half(allh)
    assert np.array_equal(b[~np.isnan(ref)], ref[~np.isnan(ref)])


def test_r11g11b10_codec(oracle):
    # every finite code round-trips exactly
    codes = np.arange(2 ** 11, dtype=np.uint32)
    packed = codes | (codes << 11) | ((codes & 1023) << 22)
    dec = oracle.codec_eval(1, packed, packed.size, np.float32, packed.size * 3).reshape(-1, 3)
    enc = oracle.codec_eval(0, dec, packed.size, np.uint32, packed.size)
    fin = ((codes >> 6) < 31) & (((codes & 1023) >> 5) < 31)
    assert np.array_equal(enc[fin], packed[fin])
    # numpy twin agrees on random values, incl. negatives, overflow and the subnormal range
    r = np.random.default_rng(5)
    v = (np.exp(r.uniform(-30, 14, (200000, 3))) * r.choice([-1.0, 1.0], (200000, 3), p=[0.05, 0.95])).astype(np.float32)
    assert np.array_equal(oracle.codec_eval(0, v, v.shape[0], np.uint32, v.shape[0]), pixfmt.pack_r11g11b10(v))
    assert np.array_equal(dec[fin], pixfmt.unpack_r11g11b10(packed)[fin])
    # rules: negative -> 0, overflow -> max finite (65024 / 64512), round to nearest even
    one = oracle.codec_eval(0, np.array([[-1.0, 1e9, 1e9]], np.float32), 1, np.uint32, 1)
    d = pixfmt.unpack_r11g11b10(one)[0]
    assert d.tolist() == [0.0, 65024.0, 64512.0]
    # 1 + 2^-7 is exactly half way between two 6-bit-mantissa neighbours: ties to even (1.0)
    tie = oracle.codec_eval(0, np.array([[1.0 + 2.0 ** -7, 1.0 + 3 * 2.0 ** -7, 0.0]], np.float32), 1, np.uint32, 1)
    d = pixfmt.unpack_r11g11b10(tie)[0]
    assert d[0] == 1.0 and d[1] == 1.0 + 2.0 ** -5


def test_unorm_snorm(oracle):
    x = np.array([0.0, 1.0, 0.5, 0.5 / 255.0, 1.5 / 255.0, 2.5 / 255.0, -3.0, 7.0, np.nan], np.float32)
    got = oracle.codec_eval(4, x, x.size, np.uint8, x.size)
    assert got.tolist() == [0, 255, 128, 0, 2, 2, 0, 255, 0]
    s = np.array([-32768, -32767, 0, 32767], np.int16)
    assert oracle.codec_eval(6, s, s.size, np.float32, s.size).tolist() == [-1.0, -1.0, 0.0, 1.0]


def test_c_abi_exports_every_declared_symbol():
    """libplr.so must load without a GPU and export every function include/plr.h declares."""
    from plainrenderer_amd import backend
    hdr = open(os.path.join(ROOT, "include", "plr.h")).read()
    declared = sorted(set(re.findall(r"\b(plr_[a-z0-9_]+)\s*\(", hdr)))
    assert len(declared) >= 35
    if not os.path.exists(backend.LIB_PATH):
        import __graft_entry__
        __graft_entry__.build()
    lib = C.CDLL(backend.LIB_PATH)
    missing = [s for s in declared if not hasattr(lib, s)]
    assert not missing, missing
    assert sorted(backend.EXPORTED_SYMBOLS) == declared
    # ... and every function the other public headers declare (frame pipeline, RCCL band exchange, SDF bake, image IO)
    import glob
    for path in sorted(glob.glob(os.path.join(ROOT, "include", "*.h"))):
        names = sorted(set(re.findall(r"\b(plr[a-z]*_[a-z0-9_]+)\s*\(", open(path).read())))
        missing = [s for s in names if not hasattr(lib, s)]
        assert not missing, (os.path.basename(path), missing)


def test_backend_fails_loudly_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from plainrenderer_amd import PlrError, RenderBackend
    with pytest.raises(PlrError):
        RenderBackend(64, 64)


def test_every_hot_path_shader_has_a_registered_kernel():
    """the shader -> kernel registry is static: a launcher that loses its registration (an edit gone wrong) is caught here, without a GPU"""
    from plainrenderer_amd import supported_shaders
    have = set(supported_shaders())
    need = {"histogramPerTile.comp", "histogramReset.comp", "histogramCombineTiles.comp", "preExposeLights.comp", "tonemapping.comp", "depthHiZPyramid.comp",
            "temporalFilter.comp", "temporalSupersampling.comp", "colorToLuminance.comp", "bloomDownsample.comp", "bloomUpsample.comp", "applyBloom.comp",
            "depthDownscale.comp", "sdfCameraFrustumCulling.comp", "sdfCameraTileCulling.comp", "sdfDiffuseTrace.comp", "filterIndirectDiffuseSpatial.comp",
            "filterIndirectDiffuseTemporal.comp", "indirectLightUpscale.comp", "brdfLut.comp", "deferredShading.comp", "sdfDebugVisualisation.comp", "lightMatrix.comp",
            "skyTransmissionLut.comp", "skyMultiscatterLut.comp", "skyLut.comp", "froxelVolumeMaterial.comp", "froxelLightScattering.comp",
            "volumeLightingReprojection.comp", "volumetricLightingIntegration.comp"}
    assert need <= have, sorted(need - have)


def test_reference_frontend_setup_compiles_against_the_shim():
    """RenderFrontend::setup calls setGlobalDescriptorSetLayout (RenderBackend.h:73, RenderFrontend.cpp:280-295) before it creates any pass.
    tests/cpp/frontend_setup_excerpt.cpp restates that set-up sequence with the reference's type and member names; it must compile against
    include/plr_render_backend.hpp with a plain host compiler, and link against libplr.so (every plr_* function the shim forwards to is exported)"""
    import subprocess
    import tempfile
    from plainrenderer_amd import backend
    src = os.path.join(ROOT, "tests", "cpp", "frontend_setup_excerpt.cpp")
    with tempfile.TemporaryDirectory() as tmp:
        exe = os.path.join(tmp, "excerpt")
        lib_dir = os.path.dirname(backend.LIB_PATH)
        p = subprocess.run(["g++", "-std=c++17", "-Wall", "-Werror", src, "-o", exe, "-L" + lib_dir, "-lplr", "-Wl,--allow-shlib-undefined"], capture_output=True, text=True, timeout=300)
        assert p.returncode == 0, p.stderr[-3000:]
    hdr = open(os.path.join(ROOT, "include", "plr_render_backend.hpp")).read()
    assert "void setGlobalDescriptorSetLayout(const ShaderLayout& layout)" in hdr and "struct ShaderLayout" in hdr


def test_the_exact_kernel_set_is_a_library_of_its_own():
    """the shipped library holds the benchmarked kernel set; the reference-order (PLR_MATH_EXACT) launch paths of the GI trace, the GI filters, the
    deferred shade, TAA and bloom live in libplr_exact.so, which libplr.so loads on demand. No kernel of csrc/kernels_exact/ is defined in libplr.so, every one of
    them is in libplr_exact.so, and that library resolves the backend's symbols against libplr.so (it carries no copy of the backend)."""
    import glob
    import subprocess
    from plainrenderer_amd import backend, build
    if not (os.path.exists(backend.LIB_PATH) and os.path.exists(build.EXACT_LIB_PATH)):
        import __graft_entry__
        __graft_entry__.build()
    kernels = set()
    for path in glob.glob(os.path.join(ROOT, "plainrenderer_amd", "csrc", "kernels_exact", "*.hip")):
        kernels |= set(re.findall(r"__global__[^;{]*?\bvoid\s+([A-Za-z0-9_]+)\s*\(", open(path).read()))
    assert len(kernels) >= 10, kernels

    def defined(lib):
        out = subprocess.run(["nm", "-D", "--defined-only", lib], capture_output=True, text=True, check=True).stdout
        return out

    main, exact = defined(backend.LIB_PATH), defined(build.EXACT_LIB_PATH)
    for k in sorted(kernels):
        assert k in exact, "%s is not in libplr_exact.so" % k
        assert k not in main, "%s (an exact-set kernel) is defined in libplr.so" % k
    assert "plr_setup" in main and "plr_setup" not in exact
    needed = subprocess.run(["readelf", "-d", build.EXACT_LIB_PATH], capture_output=True, text=True, check=True).stdout
    assert "libplr.so" in needed and "$ORIGIN" in needed
    # it loads beside libplr.so without a GPU (its launchers register themselves with the backend's shader registry at load time)
    C.CDLL(backend.LIB_PATH, mode=C.RTLD_GLOBAL)
    C.CDLL(build.EXACT_LIB_PATH)
