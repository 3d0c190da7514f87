"""BASELINE.json config 5 at its workload: the full frame (GI + shade + post) at 7680 x 4320, 256 SDF instances x 64^3, rendered as FOUR row bands
with the default halos, the benchmarked (PLR_MATH_FAST) kernel set, pass fusion on and a balanced partition, three frames of temporal feedback.

All four bands run in this process on one GPU (one host thread + one backend per band, tiling.LocalTransport moves the halo rows - the recording,
the dispatch bases, the exchange points and the kernels are exactly those of the 4-GPU run; only the transport differs). Every band is held to the
UNPARTITIONED 8K frame, and a strip that straddles a band boundary to the ORACLE frame (the scalar C++ restatement, run on the host cores).

PLR_CONFIG5_SIZE=WxH (H a multiple of 256) shrinks the frame for debugging. Measured numbers: profiles/r03_config5_8k.txt."""
import copy
import os
import sys
import threading

import numpy as np
import pytest

import parity
from plainrenderer_amd import pixfmt, tiling

W, H = (int(v) for v in os.environ.get("PLR_CONFIG5_SIZE", "7680x4320").split("x"))
N_BANDS = 4
N_FRAMES = int(os.environ.get("PLR_CONFIG5_FRAMES", "3"))  # PLR_CONFIG5_FRAMES=16 PLR_CONFIG5_REPORT_ONLY=1: the convergence series (tools/profile_round.sh)
# The partition is what bench.py --gpus 4 would use: tiling.balanced_bounds of band times MEASURED in this test on this GPU (one calibration round with
# an exchange that moves nothing, as bench.calibrate_partition does). The sky band is cheap, ground-level geometry is not.


class _Args:
    grid, sdf_res, shadow_res, steps, warmup, profile_frames = 16, 64, 2048, N_FRAMES + 1, 0, 0


def _bounds(inputs, cams):
    import time
    from plainrenderer_amd import RenderBackend
    from plainrenderer_amd.frame import FramePipeline
    eq = tiling.equal_bounds(H, N_BANDS)
    times, box = [], {}

    def measure(i):  # on a thread of its own: a backend per thread
        be = RenderBackend(W, H, device=0)
        fp = FramePipeline(be, W, H, shadow_map_res=2048, band_row_begin=eq[i], band_row_end=eq[i + 1])
        fp.set_exchange_callback(lambda exchange_id, stream: None)
        copy.copy(inputs).upload(fp)
        for f in range(3):
            fp.frame(cams[f + 1], 1.0 / 60.0, 0.5)
        be.waitForGPUIdle()
        t0 = time.perf_counter()
        for f in range(10):
            fp.frame(cams[(f % 3) + 1], 1.0 / 60.0, 0.5)
        be.waitForGPUIdle()
        box[i] = (time.perf_counter() - t0) / 10.0
        fp.destroy()
        be.shutdown()

    for i in range(N_BANDS):
        t = threading.Thread(target=measure, args=(i,))
        t.start()
        t.join(timeout=600)
        times.append(box[i])
    bounds = tiling.balanced_bounds(H, eq, times, min_rows=512)
    print("CONFIG5 partition: equal bands %s take %s ms -> balanced bounds %s" % (eq, ["%.3f" % (t * 1e3) for t in times], bounds), flush=True)
    return bounds


def _render(inputs, cams, band, bounds, group, out, capture):
    """one backend + C++ FramePipeline on the calling thread; band = index or None (the unpartitioned frame)"""
    from plainrenderer_amd import RenderBackend
    from plainrenderer_amd.frame import FramePipeline
    key = "full" if band is None else band
    try:
        be = RenderBackend(W, H, device=0)
        be.setMathMode(True)
        kw = dict(shadow_map_res=2048)
        if band is not None:
            kw.update(band_row_begin=bounds[band], band_row_end=bounds[band + 1])
            if "PLR_CONFIG5_GI_HALO" in os.environ:  # experiment hook: trace rows of GI exchanged with each neighbour (default: FramePipeline's)
                kw.update(band_gi_halo=int(os.environ["PLR_CONFIG5_GI_HALO"]))
        fp = FramePipeline(be, W, H, **kw)
        inp = inputs if band is None else copy.copy(inputs)  # (the unpartitioned run's upload leaves the texture-array indices the oracle frame needs)
        inp.upload(fp)
        ex = tiling.Exchange(fp, tiling.LocalTransport(group, band), H, N_BANDS, band, bounds) if band is not None else None
        r0, r1 = (0, H) if band is None else (bounds[band], bounds[band + 1])
        frames = []
        for f in range(N_FRAMES):
            fp.frame(cams[f + 1], 1.0 / 60.0, 0.5 + f / 60.0)
            post = be.downloadImage(fp.image("post1"), 0, np.uint32).reshape(H, W)[r0:r1].copy()
            # (the long report-only series keeps the resolved colour only: 16 frames of both images of five renders would be 17 GB of host memory)
            swap = be.downloadImage(fp.image("swapchain"), 0, np.uint8).reshape(H, W, 4)[r0:r1].copy() if N_FRAMES <= 4 else None
            rec = dict(post=post, swap=swap, hist=be.downloadStorageBuffer(fp.storage_buffer("histogram"), 512, dtype=np.uint32).copy(),
                       light=be.downloadStorageBuffer(fp.storage_buffer("light"), 20, dtype=np.uint8).tobytes())
            if capture is not None and band is None:
                rec["globals"] = fp.submitted_globals()
                rec["weights"] = fp.resolve_weights()
                rec["frustum"] = be.downloadUniformBuffer(fp.uniform_buffer("sdfCameraFrustum"), 192).tobytes()
            frames.append(rec)
        res = dict(frames=frames, fused=be.getPassFusion()[1], settings=fp.settings if band is None else None, calls=list(ex.calls) if ex else [])
        fp.destroy()
        be.shutdown()
        out[key] = res
    except BaseException as e:  # surface failures of worker threads (and unblock the others)
        out[key] = e
        if group is not None:
            group.barrier.abort()
        raise


def _join(threads, out, keys):
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=1500)
    for k in keys:
        assert k in out, "%s did not finish" % (k,)
        if isinstance(out[k], BaseException):
            raise out[k]


@pytest.mark.gpu
def test_gpu_config5_four_bands_of_the_8k_frame_against_the_unpartitioned_frame_and_the_oracle():
    import bench
    from plainrenderer_amd import backend as backend_mod
    assert H % 256 == 0 or H == 4320
    scene, cams, inputs = bench.build_scene(_Args, "cuda:0", W, H)
    bounds = _bounds(inputs, cams)
    assert bounds[0] == 0 and bounds[-1] == H and all(b % 64 == 0 for b in bounds[1:-1])
    out = {}
    _join([threading.Thread(target=_render, args=(inputs, cams, None, bounds, None, out, True))], out, ["full"])
    group = tiling.LocalGroup(N_BANDS, backend_mod._load())
    _join([threading.Thread(target=_render, args=(inputs, cams, i, bounds, group, out, None)) for i in range(N_BANDS)], out, list(range(N_BANDS)))
    full = out["full"]
    worst_within1, worst_swap, worst_moved, worst_exposure = 1.0, 1.0, 0, 0.0
    lines = []
    for f in range(N_FRAMES):
        for i in range(N_BANDS):
            b0, b1 = bounds[i], bounds[i + 1]
            fr, bf = full["frames"][f], out[i]["frames"][f]
            # the all-reduce gives every band the same histogram (bit-exact integer sum) and therefore the same exposure; against the unpartitioned
            # frame a few pixels of the previous frame's colour sit in a neighbouring bin (the denoiser's stated halo deviation, below)
            assert np.array_equal(out[0]["frames"][f]["hist"], bf["hist"]) and out[0]["frames"][f]["light"] == bf["light"], "all-reduce: frame %d band %d" % (f, i)
            assert int(bf["hist"].sum()) == W * H
            moved = int(np.abs(fr["hist"].astype(np.int64) - bf["hist"].astype(np.int64)).sum() // 2)
            ea, eb = np.frombuffer(fr["light"], np.float32), np.frombuffer(bf["light"], np.float32)
            worst_moved = max(worst_moved, moved)
            worst_exposure = max(worst_exposure, float(np.abs(ea - eb).max() / max(float(np.abs(ea).max()), 1e-30)))
            d = parity.r11g11b10_code_diff(bf["post"].reshape(-1), fr["post"][b0:b1].reshape(-1))
            within1 = float((d <= 1).all(axis=1).mean())
            sw = float((np.abs(bf["swap"].astype(np.int16) - fr["swap"][b0:b1].astype(np.int16)) <= 1).all(axis=2).mean()) if bf["swap"] is not None else float("nan")
            # where do the differing pixels sit? (rows from the nearer band edge)
            rows_off = np.nonzero((d.reshape(b1 - b0, W, 3) > 1).any(axis=2).any(axis=1))[0]
            edge_dist = int(np.minimum(rows_off, (b1 - b0 - 1) - rows_off).max()) if rows_off.size else -1
            lines.append("CONFIG5 frame %d band %d rows %d..%d: within one code of the unpartitioned frame %.6f (max code diff %d, furthest differing row %d rows from a band edge), "
                         "swapchain within 1 LSB %.6f, histogram: %d pixels in another bin" % (f, i, b0, b1, within1, int(d.max()), edge_dist, sw, moved))
            worst_within1, worst_swap = min(worst_within1, within1), min(worst_swap, sw) if sw == sw else worst_swap
    print("\n".join(lines), flush=True)
    if os.environ.get("PLR_CONFIG5_REPORT_ONLY"):
        return
    assert worst_moved <= 5e-4 * W * H, "histogram vs the unpartitioned frame: %d pixels in another bin" % worst_moved
    assert worst_exposure <= 1e-3, "exposure vs the unpartitioned frame: relative difference %.2e" % worst_exposure
    # The one stated deviation of band rendering: a disc sample of the GI denoiser beyond the exchanged trace rows gets weight 0. The disc is 1.5 m in
    # WORLD space: at 8K it spans 10275 / depth[m] pixels, i.e. more than any bounded halo on the ground in front of the camera (bands 2 and 3 of
    # this scene: differing pixels sit up to 480 rows from a band edge). Measured after three frames, worst band, by trace rows of GI halo:
    # 64: 98.64 %, 128 (the default at 4320 rows, plrf_default_settings): 99.34 %, 192: 99.38 %, 256: 99.40 % within one code; a mirrored stand-in
    # for the missing samples instead of dropping them: 97.2 % (profiles/r03_config5_8k.txt). The sky band equals the unpartitioned frame exactly.
    assert worst_within1 >= 0.99, "every band within one R11G11B10 code of the unpartitioned frame on >= 99 % of its pixels"
    assert worst_swap >= 0.999
    # the overlapped exchange sequence of a band: histogram, GI trace (begin / end), temporal GI (begin / end), GI history, resolved colour (begin / end)
    B, E = 0x100, 0x200
    assert out[1]["calls"][:8] == [0, 1 | B, 1 | E, 2 | B, 2 | E, 3, 4 | B, 4 | E]
    # band mode keeps the fusions that matter: upscale + shade as one launch, packed GI texels from the producers, the two edge dispatches of a producer as one launch
    assert out[1]["fused"] >= 10, out[1]["fused"]

    # ---- a strip across the boundary of bands 1 and 2 against the ORACLE frame (2 frames: every history populated)
    tests_dir = os.path.dirname(os.path.abspath(__file__))
    if tests_dir not in sys.path:
        sys.path.insert(0, tests_dir)
    from oracle_frame import OracleFrame
    ora = OracleFrame(inputs, W, H, 512, full["settings"])
    for f in range(2):
        fr = full["frames"][f]
        ora.frame(fr["globals"], fr["weights"], fr["frustum"], 5.0)
    edge = bounds[2]
    strip = slice(edge - 64, edge + 64)
    ref = ora.post1.reshape(H, W)[strip].reshape(-1).astype(np.uint32)
    for name, got in (("unpartitioned", full["frames"][1]["post"][strip]),
                      ("bands 1 + 2", np.concatenate([out[1]["frames"][1]["post"][-64:], out[2]["frames"][1]["post"][:64]]))):
        d = parity.r11g11b10_code_diff(np.ascontiguousarray(got).reshape(-1), ref)
        within1, within4 = float((d <= 1).all(axis=1).mean()), float((d <= 4).all(axis=1).mean())
        a, b = pixfmt.unpack_r11g11b10(np.ascontiguousarray(got).reshape(-1)), pixfmt.unpack_r11g11b10(ref)
        rel = float(np.abs(a - b).mean() / max(float(b.mean()), 1e-9))
        print("CONFIG5 %s rows %d..%d vs the oracle frame: within one code %.5f, within four %.5f, mean rel err %.2e" % (name, edge - 64, edge + 64, within1, within4, rel), flush=True)
        assert within1 >= 0.98 and within4 >= 0.99 and rel <= 2e-3
