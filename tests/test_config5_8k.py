"""BASELINE.json config 5 at its workload: the full frame (GI + shade + post) at 7680 x 4320, 256 SDF instances x 64^3, rendered as 2 x 2 SCREEN TILES - the
partition config 5 names - and as FOUR row bands, with the default halos, the benchmarked (PLR_MATH_FAST) kernel set, pass fusion on and a balanced
partition, over temporal feedback.

All four partitions run in this process on one GPU, one host thread + one backend + one pipeline each. The exchange is the NATIVE one (csrc/frontend/band_exchange.cpp:
plans, pack / unpack kernels, communication stream, BEGIN behind the producers' edge signal, END, watchdog) over its in-process transport (round 6,
plr_frame.h plrf_local_attach_rects: device copies where a communicator's ncclSend / ncclRecv go) - the recording, the dispatch bases, the exchange points, the kernels
AND the exchange code are those of the 4-GPU run; only the links differ. transport "python" (the exact mode only) is the older diagnostic path: the same pipelines
with tiling.LocalTransport moving the rectangles from Python behind host synchronisations. Every partition is held to the
UNPARTITIONED 8K frame, and a strip that straddles a boundary to the ORACLE frame (the scalar C++ restatement, run on the host cores).

Three modes (plr_frame.h band_gi_halo):
  "requested" - REQUEST LISTS (round 6, PLRF_HALO_REQUESTED): no GI halo at all; every rank asks the owners for exactly the texels its spatial-filter samples land
            on. The partitioned frame must EQUAL the unpartitioned one, byte for byte, like "exact" - for 6 - 27 MB per rank and frame instead of 150 - 200.
  "exact" - every GI texel is exchanged with every rank (PLRF_HALO_WHOLE_IMAGE): the partitioned frame must EQUAL the unpartitioned one, byte for byte, on every
            kept frame of the series (resolved colour, swapchain of the last frame, histogram, exposure). This is the parity statement of a partitioned frame.
  "halo"  - the default bounded halo (128 trace rows at 8K): a denoiser sample beyond it gets weight 0, the denoised signal is next frame's history, so the
            deviation feeds back and spreads. It does NOT settle: profiles/r05_config5_series.txt follows it over 256 frames (VERDICT r04 item 2 asked for the
            converged value: there is none within 256 frames). The asserts of this mode are therefore regression guards at frame 3 and at the last frame of the
            series (64 by default), a little below the measured values - they say "no worse than measured", not "converged".
PLR_CONFIG5_FRAMES overrides the frame count (tools/config5_series.sh).

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
N_FRAMES = int(os.environ.get("PLR_CONFIG5_FRAMES", "64"))  # PLR_CONFIG5_FRAMES=64 PLR_CONFIG5_REPORT_ONLY=1: the convergence series (tools/config5_series.sh)
# The partition is what bench.py --gpus 4 would use: tiling.balanced_bounds of band times MEASURED in this test on this GPU (one calibration round with
# an exchange that moves nothing, as bench.calibrate_partition does). The sky band is cheap, ground-level geometry is not.


# "halo" mode: share of a partition's pixels within one code of the unpartitioned frame, worst partition - measured (profiles/r05_config5_series.txt, several runs:
# the balanced partition differs a little from run to run) at frame 3: tiles 0.99922 - 0.99924, bands 0.99815 - 0.99863; at frame 63: tiles 0.99635 - 0.99646,
# bands 0.99188 - 0.99620. Guards a little below; a series of another length interpolates the measured decay (no steady state: see the module docstring)
GUARD_FRAME3 = {"tiles2x2": 0.9985, "bands4": 0.9970}
GUARD_FRAME63 = {"tiles2x2": 0.9930, "bands4": 0.9880}


KEEP_EVERY = int(os.environ.get("PLR_CONFIG5_KEEP_EVERY", "8"))
STATIC_CAMERA = bool(os.environ.get("PLR_CONFIG5_STATIC_CAMERA"))  # series hook: the camera of frame 0 for every frame (the G-buffer inputs are those of one pose anyway)


def _kept(f):
    """frames whose images are kept for the comparison: all of a short run, a thinning subset of a long series (host memory: 133 MB per 8K image)"""
    return N_FRAMES <= 8 or f < 4 or (f + 1) % KEEP_EVERY == 0 or f == N_FRAMES - 1


class _Args:
    grid, sdf_res, shadow_res, steps, warmup, profile_frames = 16, 64, 2048, N_FRAMES + 1, 0, 0  # (build_scene makes steps + 28 cameras)


def _partition(inputs, cams, kind):
    """the partition bench.py --gpus 4 would use: rectangles balanced from times MEASURED here (one calibration round with an exchange that moves nothing)"""
    import time
    from plainrenderer_amd import RenderBackend
    from plainrenderer_amd.frame import FramePipeline
    gx, gy = (2, 2) if kind == "tiles2x2" else (1, 4)
    cols, rows = tiling.equal_bounds(W, gx), tiling.equal_bounds(H, gy)
    rects = tiling.tile_rects(W, H, gx, gy, cols, rows)
    times, box = [], {}

    def measure(i):  # on a thread of its own: a backend per thread
        x0, y0, x1, y1 = rects[i]
        be = RenderBackend(W, H, device=0)
        kw = dict(band_col_begin=x0, band_col_end=x1) if gx > 1 else {}
        fp = FramePipeline(be, W, H, shadow_map_res=2048, band_row_begin=y0, band_row_end=y1, **kw)
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
    new_cols, new_rows = tiling.balanced_tile_bounds(W, H, gx, gy, cols, rows, times, min_size=512)
    out = tiling.tile_rects(W, H, gx, gy, new_cols, new_rows)
    print("CONFIG5 %s partition: equal rectangles %s take %s ms -> balanced %s" % (kind, rects, ["%.3f" % (t * 1e3) for t in times], out), flush=True)
    return out


def _render(inputs, cams, band, rects, group, out, capture, mode="halo", transport="native"):
    """one backend + C++ FramePipeline on the calling thread; band = index or None (the unpartitioned frame)"""
    from plainrenderer_amd import RenderBackend
    from plainrenderer_amd.frame import FramePipeline
    key = "full" if band is None else band
    try:
        be = RenderBackend(W, H, device=0)
        be.setMathMode(True)
        if DIAG:
            be.setPassFusion(1)  # every intermediate image is written
        kw = dict(shadow_map_res=2048)
        if band is not None:
            x0, y0, x1, y1 = rects[band]
            kw.update(band_row_begin=y0, band_row_end=y1)
            if x0 != 0 or x1 != W:
                kw.update(band_col_begin=x0, band_col_end=x1)
            if mode == "exact":
                kw.update(band_gi_halo=0xffffffff)  # PLRF_HALO_WHOLE_IMAGE
            if mode == "requested":
                kw.update(band_gi_halo=0xfffffffe)  # PLRF_HALO_REQUESTED
            if "PLR_CONFIG5_GI_HALO" in os.environ:  # experiment hook: trace rows of GI exchanged with each neighbour (default: FramePipeline's)
                kw.update(band_gi_halo=int(os.environ["PLR_CONFIG5_GI_HALO"]))
            for item in filter(None, os.environ.get("PLR_CONFIG5_HALOS", "").split(",")):  # experiment hook: "gi_history=512,post=1024,taa_history=64"
                kw["band_%s_halo" % item.split("=")[0]] = int(item.split("=")[1])
        fp = FramePipeline(be, W, H, **kw)
        if band == 0:
            print("CONFIG5 halos: gi %d gi_history %d color %d post %d taa_history %d" % (fp.settings.band_gi_halo, fp.settings.band_gi_history_halo, fp.settings.band_color_halo,
                                                                                          fp.settings.band_post_halo, fp.settings.band_taa_history_halo), flush=True)
        inp = inputs if band is None else copy.copy(inputs)  # (the unpartitioned run's upload leaves the texture-array indices the oracle frame needs)
        inp.upload(fp)
        ex = None
        if band is not None and transport == "native":
            fp.attach_local_rects(group, band, N_BANDS, W, H, rects)
        elif band is not None:
            ex = tiling.Exchange(fp, tiling.LocalTransport(group, band), H, N_BANDS, band, rects=rects, width=W)
        c0, r0, c1, r1 = (0, 0, W, H) if band is None else rects[band]
        frames = []
        for f in range(N_FRAMES):
            fp.frame(cams[1 if STATIC_CAMERA else f + 1], 1.0 / 60.0, 0.5 + f / 60.0)
            if not _kept(f):
                frames.append(None)
                continue
            post = be.downloadImage(fp.image("post1"), 0, np.uint32).reshape(H, W)[r0:r1, c0:c1].copy()
            # (the long report-only series keeps the resolved colour only: 16 frames of both images of five renders would be 17 GB of host memory)
            swap = be.downloadImage(fp.image("swapchain"), 0, np.uint8).reshape(H, W, 4)[r0:r1, c0:c1].copy() if (N_FRAMES <= 4 or f == N_FRAMES - 1) else None
            stages = {}
            if DIAG and f == DIAG_FRAME:
                for name in DIAG:
                    wi, hi, _, bpp = be.mipSize(fp.image(name), 0)
                    sx, sy = W // wi, H // hi
                    stages[name] = (sx, sy, be.downloadImage(fp.image(name), 0, np.uint8).reshape(hi, wi, bpp)[r0 // sy:-(-r1 // sy), c0 // sx:-(-c1 // sx)].copy())
            rec = dict(post=post, swap=swap, stages=stages, hist=be.downloadStorageBuffer(fp.storage_buffer("histogram"), 512, dtype=np.uint32).copy(),
                       light=be.downloadStorageBuffer(fp.storage_buffer("light"), 20, dtype=np.uint8).tobytes())
            if capture is not None and band is None:
                rec["globals"] = fp.submitted_globals()
                rec["weights"] = fp.resolve_weights()
                rec["frustum"] = be.downloadUniformBuffer(fp.uniform_buffer("sdfCameraFrustum"), 192).tobytes()
            frames.append(rec)
        res = dict(frames=frames, fused=be.getPassFusion()[1], settings=fp.settings if band is None else None, calls=list(ex.calls) if ex else [],
                   exchange=(fp.rccl_stats(), fp.rccl_info()) if (band is not None and transport == "native") else None)
        fp.destroy()
        be.shutdown()
        out[key] = res
    except BaseException as e:  # surface failures of worker threads (and unblock the others)
        out[key] = e
        if group is not None:
            group.abort() if transport == "native" else group.barrier.abort()
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


# PLR_CONFIG5_DIAG="depthHalfRes,giYSH0,..." (image names of FramePipeline::image): with every intermediate kept (pass fusion level 1), where does frame
# PLR_CONFIG5_DIAG_FRAME of each partition first differ from the unpartitioned frame? (report only)
DIAG = [n for n in os.environ.get("PLR_CONFIG5_DIAG", "").split(",") if n]
DIAG_FRAME = int(os.environ.get("PLR_CONFIG5_DIAG_FRAME", "0"))
_FULL = {}  # the unpartitioned 8K frames, rendered once for both partitions


@pytest.mark.gpu
@pytest.mark.parametrize("transport,mode", [("native", "requested"), ("native", "exact"), ("native", "halo"), ("python", "exact")])
@pytest.mark.parametrize("kind", ["tiles2x2", "bands4"])
def test_gpu_config5_the_8k_frame_partitioned_against_the_unpartitioned_frame_and_the_oracle(kind, mode, transport):
    import bench
    from plainrenderer_amd import backend as backend_mod
    from plainrenderer_amd.frame import LocalExchangeGroup
    assert H % 256 == 0 or H == 4320
    if "scene" not in _FULL:
        _FULL["scene"] = bench.build_scene(_Args, "cuda:0", W, H)
    scene, cams, inputs = _FULL["scene"]
    rects = _partition(inputs, cams, kind)
    assert sum((r[2] - r[0]) * (r[3] - r[1]) for r in rects) == W * H and all(v % 64 == 0 or v in (W, H) for r in rects for v in r)
    out = {}
    if "full" not in _FULL:
        _join([threading.Thread(target=_render, args=(inputs, cams, None, rects, None, out, True))], out, ["full"])
        _FULL["full"] = out["full"]
    full = _FULL["full"]
    group = LocalExchangeGroup(N_BANDS) if transport == "native" else tiling.LocalGroup(N_BANDS, backend_mod._load())
    try:
        _join([threading.Thread(target=_render, args=(inputs, cams, i, rects, group, out, None, mode, transport)) for i in range(N_BANDS)], out, list(range(N_BANDS)))
    finally:
        if transport == "native":
            group.destroy()
    if transport == "native":
        (sent, received, groups), info = out[1]["exchange"]
        print("CONFIG5 %s %s native in-process exchange, partition 1: %.1f MB sent, %.1f MB received per frame in %d groups; overlap mode %d, %s" % (
            kind, mode, sent / 1e6, received / 1e6, groups, info["overlap_mode"], "rectangles through pack / unpack kernels" if info["packed_regions"] else "whole rows straight from the images"), flush=True)
        assert sent > 0 and received > 0 and groups >= 4
    worst_within1, worst_swap, worst_moved, worst_exposure, worst_early, worst_code = 1.0, 1.0, 0, 0.0, 1.0, 0
    kind_mode = "%s %s" % (kind, mode) + ("" if transport == "native" else " (python transport)")
    lines = []
    for f in range(N_FRAMES):
        if not _kept(f):
            continue
        for i in range(N_BANDS):
            c0, b0, c1, b1 = rects[i]
            fr, bf = full["frames"][f], out[i]["frames"][f]
            # the all-reduce gives every partition the same histogram (bit-exact integer sum) and therefore the same exposure; against the unpartitioned
            # frame a few pixels of the previous frame's colour sit in a neighbouring bin (the denoiser's stated halo deviation, below)
            assert np.array_equal(out[0]["frames"][f]["hist"], bf["hist"]) and out[0]["frames"][f]["light"] == bf["light"], "all-reduce: frame %d partition %d" % (f, i)
            assert int(bf["hist"].sum()) == W * H
            moved = int(np.abs(fr["hist"].astype(np.int64) - bf["hist"].astype(np.int64)).sum() // 2)
            ea, eb = np.frombuffer(fr["light"], np.float32), np.frombuffer(bf["light"], np.float32)
            worst_moved = max(worst_moved, moved)
            worst_exposure = max(worst_exposure, float(np.abs(ea - eb).max() / max(float(np.abs(ea).max()), 1e-30)))
            ref = np.ascontiguousarray(fr["post"][b0:b1, c0:c1])
            d = parity.r11g11b10_code_diff(bf["post"].reshape(-1), ref.reshape(-1))
            within1 = float((d <= 1).all(axis=1).mean())
            sw = float((np.abs(bf["swap"].astype(np.int16) - fr["swap"][b0:b1, c0:c1].astype(np.int16)) <= 1).all(axis=2).mean()) if bf["swap"] is not None else float("nan")
            # where do the differing pixels sit? (rows / columns from the nearer edge of the partition)
            off = (d.reshape(b1 - b0, c1 - c0, 3) > 1).any(axis=2)
            rows_off, cols_off = np.nonzero(off.any(axis=1))[0], np.nonzero(off.any(axis=0))[0]
            edge_dist = int(np.minimum(rows_off, (b1 - b0 - 1) - rows_off).max()) if rows_off.size else -1
            edge_dist_x = int(np.minimum(cols_off, (c1 - c0 - 1) - cols_off).max()) if cols_off.size else -1
            lines.append("CONFIG5 %s frame %d partition %d [%d..%d) x [%d..%d): within one code of the unpartitioned frame %.6f (max code diff %d, furthest differing row %d / column %d from "
                         "an edge), swapchain within 1 LSB %.6f, histogram: %d pixels in another bin" % (kind_mode, f, i, c0, c1, b0, b1, within1, int(d.max()), edge_dist, edge_dist_x, sw, moved))
            worst_within1, worst_swap = min(worst_within1, within1), min(worst_swap, sw) if sw == sw else worst_swap
            worst_code = max(worst_code, int(d.max()))
            if mode in ("exact", "requested"):
                assert np.array_equal(bf["post"], ref) and fr["hist"].tobytes() == bf["hist"].tobytes() and fr["light"] == bf["light"], "exact partition: frame %d partition %d" % (f, i)
                assert bf["swap"] is None or np.array_equal(bf["swap"], fr["swap"][b0:b1, c0:c1])
            if f <= 3:
                worst_early = min(worst_early, within1)
    for name in DIAG:
        for i in range(N_BANDS):
            c0, b0, c1, b1 = rects[i]
            sx, sy, mine = out[i]["frames"][DIAG_FRAME]["stages"][name]
            whole = full["frames"][DIAG_FRAME]["stages"][name][2][b0 // sy:-(-b1 // sy), c0 // sx:-(-c1 // sx)]
            off = (mine != whole).any(axis=2)
            ys, xs = np.nonzero(off)
            where = "none" if not ys.size else "rows %d..%d columns %d..%d of the partition's %d x %d texels, furthest from an edge: row %d column %d" % (
                ys.min(), ys.max(), xs.min(), xs.max(), off.shape[1], off.shape[0], int(np.minimum(ys, off.shape[0] - 1 - ys).max()), int(np.minimum(xs, off.shape[1] - 1 - xs).max()))
            lines.append("CONFIG5 %s DIAG frame %d %-16s partition %d: %8d texels differ (%s)" % (kind, DIAG_FRAME, name, i, int(off.sum()), where))
    lines.append("CONFIG5 %s summary over %d frames: worst share within one code %.6f (frames 0-3: %.6f), max code diff %d, histogram pixels in another bin <= %d, exposure relative "
                 "difference <= %.2e" % (kind_mode, N_FRAMES, worst_within1, worst_early, worst_code, worst_moved, worst_exposure))
    print("\n".join(lines), flush=True)
    if os.environ.get("PLR_CONFIG5_REPORT_ONLY"):
        return
    if mode in ("exact", "requested"):
        assert worst_code == 0 and worst_moved == 0 and worst_exposure == 0.0 and worst_swap == 1.0
    assert worst_moved <= 1e-3 * W * H, "histogram vs the unpartitioned frame: %d pixels in another bin" % worst_moved
    assert worst_exposure <= 1e-3, "exposure vs the unpartitioned frame: relative difference %.2e" % worst_exposure
    # The one stated deviation of a partitioned frame in "halo" mode: a disc sample of the GI denoiser beyond the exchanged halo gets weight 0. The disc is 1.5 m
    # in WORLD space: at 8K it spans 10275 / depth[m] pixels, i.e. more than any bounded halo on near geometry, and the denoised signal is next frame's history:
    # the share of a partition's pixels within one code of the unpartitioned frame keeps falling (no steady state within 256 frames). Guards, not a converged value:
    last_guard = GUARD_FRAME63[kind] if N_FRAMES <= 64 else 0.85
    assert worst_early >= GUARD_FRAME3[kind], "frames 0-3: every partition within one R11G11B10 code of the unpartitioned frame on >= %.4f of its pixels (got %.5f)" % (GUARD_FRAME3[kind], worst_early)
    assert worst_within1 >= last_guard, "frames 0-%d: >= %.4f of every partition's pixels within one code (got %.5f)" % (N_FRAMES - 1, last_guard, worst_within1)
    assert worst_swap >= 0.99
    # the overlapped exchange sequence of a partition: histogram, GI trace (begin / end), temporal GI (begin / end), GI history, resolved colour (begin / end)
    B, E = 0x100, 0x200
    if transport != "native":
        assert out[1]["calls"][:8] == [0, 1 | B, 1 | E, 2 | B, 2 | E, 3, 4 | B, 4 | E]
    # partition mode keeps the fusions that matter: per-tile pyramid + culling, upscale + shade as one launch, packed GI texels from the producers
    assert out[1]["fused"] >= 10, out[1]["fused"]

    # ---- a strip across a boundary against the ORACLE frame (2 frames: every history populated): for the bands the rows around the boundary of bands 1 and 2,
    # for the tiles the same rows of the two LEFT tiles' boundary (its columns)
    tests_dir = os.path.dirname(os.path.abspath(__file__))
    if tests_dir not in sys.path:
        sys.path.insert(0, tests_dir)
    from oracle_frame import OracleFrame
    if "oracle" not in _FULL:
        ora = OracleFrame(inputs, W, H, 512, full["settings"])
        for f in range(2):
            fr = full["frames"][f]
            ora.frame(fr["globals"], fr["weights"], fr["frustum"], 5.0)
        _FULL["oracle"] = ora.post1.reshape(H, W).copy()
    ora_post = _FULL["oracle"]
    if kind == "bands4":
        upper, lower = 1, 2
    else:
        upper, lower = 0, 2  # the left column of tiles
    edge = rects[upper][3]
    assert rects[lower][1] == edge
    cx0, cx1 = rects[upper][0], min(rects[upper][2], rects[lower][2])
    strip = slice(edge - 64, edge + 64)
    ref = np.ascontiguousarray(ora_post[strip, cx0:cx1]).reshape(-1).astype(np.uint32)
    a_up, a_lo = out[upper]["frames"][1]["post"], out[lower]["frames"][1]["post"]
    stitched = np.concatenate([a_up[-64:, :cx1 - cx0], a_lo[:64, :cx1 - cx0]])
    for name, got in (("unpartitioned", full["frames"][1]["post"][strip, cx0:cx1]), ("partitions %d + %d" % (upper, lower), stitched)):
        d = parity.r11g11b10_code_diff(np.ascontiguousarray(got).reshape(-1), ref)
        within1, within4 = float((d <= 1).all(axis=1).mean()), float((d <= 4).all(axis=1).mean())
        a, b = pixfmt.unpack_r11g11b10(np.ascontiguousarray(got).reshape(-1)), pixfmt.unpack_r11g11b10(ref)
        rel = float(np.abs(a - b).mean() / max(float(b.mean()), 1e-9))
        print("CONFIG5 %s %s rows %d..%d vs the oracle frame: within one code %.5f, within four %.5f, mean rel err %.2e" % (kind, name, edge - 64, edge + 64, within1, within4, rel), flush=True)
        assert within1 >= 0.98 and within4 >= 0.99 and rel <= 2e-3
