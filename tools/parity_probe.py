"""Per-pass error statistics of the PLR_MATH_FAST kernel set against the oracle at a realistic size (diagnostic that the bounds in
tests/test_parity_fullsize.py were derived from). Every HIP pass is fed exactly what the oracle pass consumed in frame 2 of the
bench workload.   python tools/parity_probe.py [width height]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np
import bench
import passes
from oracle_frame import OracleFrame
from plainrenderer_amd import RenderBackend, pixfmt
from plainrenderer_amd.frame import FramePipeline
from util import F


def stats(name, got, ref, quantum_rel, abs_floor=1e-4):
    got = np.asarray(got, np.float64).reshape(-1); ref = np.asarray(ref, np.float64).reshape(-1)
    err = np.abs(got - ref)
    out = [name, "n=%d" % ref.size, "nonfinite=%d" % int((~np.isfinite(got)).sum())]
    for k in (1.0, 2.0, 4.0):
        tol = np.maximum(k * quantum_rel * np.abs(ref), abs_floor)
        out.append("viol(%gq)=%.5f%%" % (k, 100.0 * float((err > tol).mean())))
    tol7 = np.maximum(2.0 ** -7 * np.abs(ref), abs_floor)
    out.append("viol(2^-7)=%.5f%%" % (100.0 * float((err > tol7).mean())))
    out.append("max_abs=%.3g scale=%.3g mean_rel=%.3g" % (err.max(), np.abs(ref).max(), err.mean() / max(np.abs(ref).mean(), 1e-30)))
    print(" ".join(out), flush=True)


def main():
    w, h = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (1920, 1080)
    class A: pass
    args = A(); args.grid = 16; args.sdf_res = 64; args.shadow_res = 2048; args.steps = 4; args.warmup = 0; args.profile_frames = 0
    be = RenderBackend(w, h, device=0)
    be.setMathMode(True)
    fp = FramePipeline(be, w, h, shadow_map_res=2048)
    scene, cams, inputs = bench.build_scene(args, "cuda:0", w, h)
    inputs.upload(fp)
    ora = OracleFrame(inputs, w, h, 512, fp.settings)
    t0 = time.time()
    for f in range(2):
        fp.frame(cams[f + 1], 1.0 / 60.0, 0.5 + f / 60.0)
        frustum = be.downloadUniformBuffer(fp.uniform_buffer("sdfCameraFrustum"), 192).tobytes()
        ora.capture = f == 1
        ora.frame(fp.submitted_globals(), fp.resolve_weights(), frustum, 5.0)
    print("oracle 2 frames at %dx%d: %.1f s" % (w, h, time.time() - t0), flush=True)
    cap, gp, gb = ora.cap, ora.cap["global"], inputs.gb
    tw, th = w // 2, h // 2
    # whole frame, fast set vs oracle (decision flips propagate through the chain here)
    post = pixfmt.unpack_r11g11b10(be.downloadImage(fp.image("post1"), 0, np.uint32))
    stats("FRAME post1 (end to end)", post, pixfmt.unpack_r11g11b10(ora.post1), 2.0 ** -6)
    sw = be.downloadImage(fp.image("swapchain"), 0, np.uint8).astype(int)
    d = np.abs(sw - ora.swapchain.reshape(-1).astype(int))
    print("FRAME swapchain: max LSB diff %d, >1 LSB: %.4f%%" % (d.max(), 100.0 * (d > 1).mean()), flush=True)
    U = pixfmt.unpack_half
    vol_idx, noise_idx = list(inputs.volume_indices), [int(x) for x in ora_global(gp).noiseTextureIndices]
    # trace
    c = cap["trace"]
    yg, cg = passes.gpu_sdf_trace(be, gb["depth"], gb["normal"], w, h, tw, th, inputs.sky, 200, 100, c["light"], inputs.instance_bytes_patched, c["tiles"], 5.0,
                                  inputs.shadow_info, inputs.shadow_maps[c["cascade"]], inputs.shadow_res, gp, strict=True, cascade=c["cascade"])
    stats("trace Y_SH", U(yg), U(c["out"][0]), 2.0 ** -10); stats("trace CoCg", U(cg), U(c["out"][1]), 2.0 ** -10)
    for k, fi in (("spatial0", 0), ("spatial1", 1)):
        c = cap[k]
        dsrc, dfmt, dw, dh = c["depth"]
        yg, cg = passes.gpu_gi_spatial(be, c["inp"][0], c["inp"][1], tw, th, dsrc, dfmt, dw, dh, gb["normal"], w, h, gp, fi)
        stats(k + " Y_SH", U(yg), U(c["out"][0]), 2.0 ** -10); stats(k + " CoCg", U(cg), U(c["out"][1]), 2.0 ** -10)
    c = cap["temporal"]
    tg = passes.gpu_gi_temporal(be, *c["inp"], tw, th, gb["motion"], gb["motion"], w, h, gp)
    stats("temporal Y_SH", U(tg[0]), U(c["out"][0]), 2.0 ** -10); stats("temporal CoCg", U(tg[1]), U(c["out"][1]), 2.0 ** -10)
    c = cap["upscale"]
    yg, cg = passes.gpu_gi_upscale(be, c["inp"][0], c["inp"][1], tw, th, gb["depth"], c["half_depth"], w, h, gp)
    stats("upscale Y_SH", U(yg), U(c["out"][0]), 2.0 ** -10); stats("upscale CoCg", U(cg), U(c["out"][1]), 2.0 ** -10)
    c = cap["shade"]
    s = fp.settings
    got = passes.gpu_deferred_shading(be, gb, w, h, ora.brdf_lut, 512, c["light"], inputs.shadow_info, inputs.shadow_maps, inputs.shadow_res, c["gi"][0], c["gi"][1],
                                      inputs.froxel, inputs.froxel_dims, inputs.vol_settings, inputs.sky, gp, int(s.diffuse_brdf), int(s.direct_multiscatter),
                                      bool(s.use_geometry_aa), int(s.indirect_lighting_tech), int(s.sun_shadow_cascade_count))
    code_stats("shade", got, c["out"])
    c = cap["taa"]
    og, hg = passes.gpu_taa(be, c["inp"], c["history"], gb["motion"], gb["depth"], w, h, c["weights"], gp, True, True, 4, True)
    code_stats("taa", og, c["out"])
    c = cap["bloom"]
    out_g, _, _ = passes.gpu_bloom(be, c["inp"], w, h, float(s.bloom_strength), float(s.bloom_radius))
    code_stats("bloom", out_g, c["out"])
    c = cap["tonemap"]
    a = passes.gpu_tonemap(be, c["inp"], w, h, gp, F.BGRA8_uNorm).astype(int).reshape(-1)
    d = np.abs(a - c["out"].astype(int).reshape(-1))
    print("tonemap: max LSB diff %d, differing channels %.4f%%" % (d.max(), 100.0 * (d != 0).mean()), flush=True)
    be.shutdown()


def ora_global(gp):
    import pyoracle
    return pyoracle.global_from_bytes(gp)


def code_stats(name, got_u32, ref_u32):
    """R11G11B10 images: distribution of the per-channel code difference (one code step = one storage quantum)"""
    got_u32 = np.asarray(got_u32, np.uint32).reshape(-1); ref_u32 = np.asarray(ref_u32, np.uint32).reshape(-1)
    chans = ((0, 0x7ff), (11, 0x7ff), (22, 0x3ff))
    line = [name]
    for (sh, m), cn in zip(chans, "RGB"):
        d = np.abs(((got_u32 >> sh) & m).astype(np.int64) - ((ref_u32 >> sh) & m).astype(np.int64))
        line.append("%s: >0 %.4f%% >1 %.5f%% >2 %.5f%% >8 %.5f%% max %d" % (cn, 100.0 * (d > 0).mean(), 100.0 * (d > 1).mean(), 100.0 * (d > 2).mean(), 100.0 * (d > 8).mean(), d.max()))
    print(" | ".join(line), flush=True)


if __name__ == "__main__":
    main()
