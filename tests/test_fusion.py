"""Pass fusion (include/plr.h plr_set_pass_fusion, csrc/backend.h): the backend covers certain ADJACENT recorded executions with fewer kernel
launches. The boundary and the results are unchanged: every frame of a fused run equals the unfused run byte for byte. At level 2 (the default)
the fused upscale + deferred shade keeps the upscaled GI texels in registers when no other pass reads them, and the temporal GI filter writes only the
packed texels the spatial filter behind it gathers: those images are then not written, a download of them fails loudly, and everything else still equals
the unfused run."""
import os

import numpy as np
import pytest

from plainrenderer_amd import synth
from plainrenderer_amd.scene import Camera

LUT_RES = 32


@pytest.mark.gpu
@pytest.mark.parametrize("w,h,half_res", [(1280, 720, 1), (648, 360, 1), (640, 352, 0)])
def test_gpu_fused_frames_equal_unfused_frames_byte_for_byte(backend, w, h, half_res):
    from plainrenderer_amd.frame import FramePipeline, SyntheticInputs
    n_frames = 4
    cams = [Camera.look((15.0 + 0.03 * i, -7.0, -6.0 + 0.05 * i), (0.0, 0.16, 1.0), aspect=w / h) for i in range(n_frames + 1)]
    scene = synth.SynthScene(grid=4, cell=8.0, seed_id=510)
    inputs = None
    results, fused_counts = {}, {}
    try:
        backend.setMathMode(True)
        backend.setEarlyParts(0)  # (the shade as two launches is within one code of the single launch, not byte-identical: test_gpu_the_shade_as_two_launches_... below)
        for fusion in (2, 1, 0):
            backend.setPassFusion(fusion)
            fp = FramePipeline(backend, w, h, shadow_map_res=256, brdf_lut_res=LUT_RES, froxel_depth=16, max_sdf_instances=64, sdf_half_res_trace=half_res)
            if inputs is None:
                inputs = SyntheticInputs(scene, cams[1], cams[0], w, h, sdf_res=16, shadow_res=256, froxel_depth=16, sun_direction=(0.35, -0.8, 0.45))
            inputs.upload(fp)
            out = []
            for f in range(n_frames):
                fp.frame(cams[f + 1], 1.0 / 60.0, 0.5 + f / 60.0)
                enabled, fused = backend.getPassFusion()
                assert enabled == fusion
                fused_counts[fusion] = fused
                names = ["swapchain", "post1", "giYSH0", "giCoCg1", "giHistoryYSH0", "giHistoryYSH1", "giHistoryCoCg0", "giFullResYSH", "pyramid", "depthHalfRes", "taaHistory0",
                         "taaHistory1"]
                if fusion == 2:
                    if half_res:  # the half-res trace has an upscale pass: its output is consumed inside the fused upscale + shade launch
                        with pytest.raises(RuntimeError, match="not written in the last frame"):
                            backend.downloadImage(fp.image("giFullResYSH"), 0, np.uint8)
                    # the temporal GI filter's four outputs: the spatial filter behind it gathers the packed texels the temporal kernel writes for it, and nothing else
                    # reads them before the next frame overwrites them (backend.cpp markElidableBehindConsumer)
                    for gone in ("giYSH0", "giHistoryYSH1"):
                        with pytest.raises(RuntimeError, match="not written in the last frame"):
                            backend.downloadImage(fp.image(gone), 0, np.uint8)
                    names = [n for n in names if n not in ("giFullResYSH", "giYSH0", "giHistoryYSH1")]
                imgs = [backend.downloadImage(fp.image(name), 0, np.uint8).copy() for name in names]
                out.append(imgs + [backend.downloadStorageBuffer(fp.storage_buffer("light"), 20, dtype=np.uint8).copy(),
                                   backend.downloadStorageBuffer(fp.storage_buffer("histogram"), 512, dtype=np.uint8).copy()])
            results[fusion] = (out, names + ["light buffer", "histogram"])
            fp.destroy()
    finally:
        backend.setPassFusion(2)
        backend.setEarlyParts(0)
        backend.setMathMode(False)
    assert fused_counts[0] == 0
    assert fused_counts[1] >= 6, "executions inside fused launches: %d" % fused_counts[1]
    assert fused_counts[2] == fused_counts[1]
    if half_res: assert fused_counts[1] >= 8, "the upscale + deferred shade pair is fused when the trace runs at half resolution"
    for level in (1, 2):
        for f in range(n_frames):
            unfused = dict(zip(results[0][1], results[0][0][f]))
            for a, what in zip(results[level][0][f], results[level][1]):
                assert np.array_equal(a, unfused[what]), "%s differs with pass fusion level %d, frame %d (%dx%d)" % (what, level, f, w, h)


@pytest.mark.gpu
@pytest.mark.parametrize("rows", [(0, 720), (128, 576)])
def test_gpu_the_front_of_a_band_as_two_launches_changes_no_byte(backend, rows):
    """Band rendering records histogramPerTile, reset, combine, [histogram all-reduce callback], preExposeLights, per-tile pyramid, depth downscale, the two culling passes.
    The callback names the histogram buffer: the backend sinks it and the exposure pass behind the pyramid and the culling (plr_set_pass_fusion_reorder), and the seven
    passes in front of it run as two launches (kernels_fast/fused_front.h "a BAND's front"). Every image and buffer equals the unfused frame's, over frames of
    feedback, for a band that is the whole frame and for one in its middle (1280 x 720; compared on the band's rows)."""
    from plainrenderer_amd.frame import FramePipeline, SyntheticInputs
    w, h, n_frames = 1280, 720, 3
    cams = [Camera.look((15.0 + 0.03 * i, -7.0, -6.0 + 0.05 * i), (0.0, 0.16, 1.0), aspect=w / h) for i in range(n_frames + 1)]
    scene = synth.SynthScene(grid=4, cell=8.0, seed_id=512)
    inputs = None
    results, fused_counts = {}, {}
    names = ["swapchain", "post1", "pyramid", "depthHalfRes", "giHistoryYSH0", "giHistoryCoCg0", "taaHistory0", "taaHistory1"]
    try:
        backend.setMathMode(True)
        for mode, (fusion, reorder) in {"two launches": (2, True), "recorded order": (2, False), "unfused": (0, False)}.items():
            backend.setPassFusion(fusion)
            backend.setPassFusionReorder(reorder)
            fp = FramePipeline(backend, w, h, shadow_map_res=256, brdf_lut_res=LUT_RES, froxel_depth=16, max_sdf_instances=64, band_row_begin=rows[0], band_row_end=rows[1])
            fp.set_exchange_callback(lambda exchange_id, stream: None)  # a band with no neighbour to exchange with: the recording is what matters here
            if inputs is None:
                inputs = SyntheticInputs(scene, cams[1], cams[0], w, h, sdf_res=16, shadow_res=256, froxel_depth=16, sun_direction=(0.35, -0.8, 0.45))
            inputs.upload(fp)
            out = []
            for f in range(n_frames):
                fp.frame(cams[f + 1], 1.0 / 60.0, 0.5 + f / 60.0)
                fused_counts[mode] = backend.getPassFusion()[1]
                if fusion:
                    general, which = backend.getGeneralKernelExecutions()
                    assert general == 0, which
                def band_rows_of(name):  # the band's own rows of an image (outside them a launch may or may not write its halo tiles: whole 64-row tiles in the fused front)
                    wi, hi, _, bpp = backend.mipSize(fp.image(name), 0)
                    a = backend.downloadImage(fp.image(name), 0, np.uint8).reshape(hi, wi * bpp)
                    return a[rows[0] * hi // h:rows[1] * hi // h].copy()
                out.append([band_rows_of(name) for name in names] +
                           [backend.downloadStorageBuffer(fp.storage_buffer("light"), 20, dtype=np.uint8).copy(),
                            backend.downloadStorageBuffer(fp.storage_buffer("histogram"), 512, dtype=np.uint8).copy(),
                            backend.downloadStorageBuffer(fp.storage_buffer("sdfCulledTiles"), 4096, dtype=np.uint8).copy()])
            results[mode] = out
            fp.destroy()
    finally:
        backend.setPassFusion(2)
        backend.setPassFusionReorder(True)
        backend.setMathMode(False)
    print("FUSION band front rows %s: executions inside fused launches %s" % (rows, fused_counts), flush=True)
    assert fused_counts["unfused"] == 0
    assert fused_counts["two launches"] >= fused_counts["recorded order"] + 1, fused_counts  # seven executions in one group instead of a pair (reset + combine) and a four
    what = names + ["light buffer", "histogram", "culled tiles"]
    for mode in ("two launches", "recorded order"):
        for f in range(n_frames):
            for a, b, name in zip(results[mode][f], results["unfused"][f], what):
                assert np.array_equal(a, b), "%s differs (%s), frame %d, band rows %s" % (name, mode, f, rows)


@pytest.mark.gpu
@pytest.mark.parametrize("w,h", [(1280, 720), (648, 360)])
def test_gpu_fusion_across_the_callers_pass_order_changes_no_byte(backend, w, h):
    """With the input producers recorded as compute passes (RenderFrontend.cpp:342-405 order) the transmission / multiscatter / sky LUT passes and the
    light matrix sit between the members of the fused frame front. plr_set_pass_fusion_reorder (default on) moves them in front of / behind the group where
    the recorded resources allow it: more executions run inside fused launches, and every image and buffer of every frame equals the run with the
    reordering off - and the run without any fusion."""
    from plainrenderer_amd.frame import FramePipeline, SyntheticInputs
    n_frames = 3
    cams = [Camera.look((15.0 + 0.03 * i, -7.0, -6.0 + 0.05 * i), (0.0, 0.16, 1.0), aspect=w / h) for i in range(n_frames + 1)]
    scene = synth.SynthScene(grid=4, cell=8.0, seed_id=511)
    inputs = None
    results, fused_counts = {}, {}
    names = ["swapchain", "post1", "pyramid", "depthHalfRes", "skyLut", "skyMultiscatterLut", "transmissionLut", "volumetricIntegrationVolume", "taaHistory0", "taaHistory1",
             "giHistoryYSH0", "giHistoryCoCg0"]
    try:
        backend.setMathMode(True)
        for mode, (fusion, reorder) in {"reordered": (2, True), "recorded order": (2, False), "unfused": (0, False)}.items():
            backend.setPassFusion(fusion)
            backend.setPassFusionReorder(reorder)
            # froxel depth 64 (the default: eight column segments in the fused per-froxel + integration launch) at the larger size, 16 (two segments) at the smaller:
            # the integrated volume must not depend on whether the four froxel passes were fused, whatever the number of segments
            froxel_depth = 64 if w >= 1280 else 16
            fp = FramePipeline(backend, w, h, shadow_map_res=256, brdf_lut_res=LUT_RES, froxel_depth=froxel_depth, max_sdf_instances=64, run_sky_luts=1, run_light_matrix=1, run_volumetrics=1)
            if inputs is None:
                inputs = SyntheticInputs(scene, cams[1], cams[0], w, h, sdf_res=16, shadow_res=256, froxel_depth=froxel_depth, sun_direction=(0.35, -0.8, 0.45))
            inputs.upload(fp)
            out = []
            for f in range(n_frames):
                fp.frame(cams[f + 1], 1.0 / 60.0, 0.5 + f / 60.0)
                fused_counts[mode] = backend.getPassFusion()[1]
                if mode == "reordered":
                    general, which = backend.getGeneralKernelExecutions()
                    assert general == 0, which
                out.append([backend.downloadImage(fp.image(name), 0, np.uint8).copy() for name in names] +
                           [backend.downloadStorageBuffer(fp.storage_buffer("light"), 20, dtype=np.uint8).copy(),
                            backend.downloadStorageBuffer(fp.storage_buffer("histogram"), 512, dtype=np.uint8).copy(),
                            backend.downloadStorageBuffer(fp.storage_buffer("sunShadowInfo"), 304, dtype=np.uint8).copy()])
            results[mode] = out
            fp.destroy()
    finally:
        backend.setPassFusion(2)
        backend.setPassFusionReorder(True)
        backend.setMathMode(False)
    print("FUSION reorder: executions inside fused launches %s" % fused_counts, flush=True)
    assert fused_counts["unfused"] == 0
    assert fused_counts["reordered"] >= fused_counts["recorded order"] + 4, fused_counts  # the frame front as one group (8 executions) instead of two pairs of it
    what = names + ["light buffer", "histogram", "sun shadow info"]
    for mode in ("reordered", "recorded order"):
        for f in range(n_frames):
            for a, b, name in zip(results[mode][f], results["unfused"][f], what):
                assert np.array_equal(a, b), "%s differs (%s), frame %d (%dx%d)" % (name, mode, f, w, h)


@pytest.mark.gpu
def test_gpu_fusion_is_off_in_exact_mode_and_when_a_callback_separates_the_passes(backend):
    """PLR_MATH_EXACT runs every pass on its own (the bit-exact kernel set has no fused launchers)"""
    from plainrenderer_amd.frame import FramePipeline, SyntheticInputs
    w, h = 320, 192
    cams = [Camera.look((15.0, -7.0, -6.0), (0.0, 0.16, 1.0), aspect=w / h) for _ in range(2)]
    scene = synth.SynthScene(grid=4, cell=8.0, seed_id=511)
    backend.setMathMode(False)
    fp = FramePipeline(backend, w, h, shadow_map_res=128, brdf_lut_res=16, froxel_depth=8, max_sdf_instances=64)
    SyntheticInputs(scene, cams[1], cams[0], w, h, sdf_res=16, shadow_res=128, froxel_depth=8, sun_direction=(0.35, -0.8, 0.45)).upload(fp)
    fp.frame(cams[1], 1.0 / 60.0, 0.5)
    assert backend.getPassFusion() == (2, 0)
    fp.destroy()


@pytest.mark.gpu
@pytest.mark.parametrize("w,h", [(1280, 720), (648, 360)])
def test_gpu_async_frame_tail_changes_nothing_but_the_schedule(backend, w, h):
    """plr_compute_pass_execution::async_tail (include/plr.h): the C++ FramePipeline flags the bloom chain + tonemap, the backend runs them on a second
    stream beside the NEXT frame's exposure / GI / shade passes and orders the two by the images they touch. Six frames of temporal feedback with the
    tail asynchronous equal the in-order run byte for byte - downloads in between (which join the tail) and at the end only."""
    from plainrenderer_amd.frame import FramePipeline, SyntheticInputs
    n_frames = 6
    cams = [Camera.look((15.0 + 0.03 * i, -7.0, -6.0 + 0.05 * i), (0.0, 0.16, 1.0), aspect=w / h) for i in range(n_frames + 1)]
    scene = synth.SynthScene(grid=4, cell=8.0, seed_id=512)
    names = ["swapchain", "post1", "giHistoryYSH0", "taaHistory0", "taaHistory1", "color0", "color1"]
    inputs, results = None, {}
    try:
        backend.setMathMode(True)
        for mode in ("async", "async, no host sync between frames", "in order"):
            backend.setAsyncTail(mode != "in order")
            fp = FramePipeline(backend, w, h, shadow_map_res=256, brdf_lut_res=LUT_RES, froxel_depth=16, max_sdf_instances=64)
            if inputs is None:
                inputs = SyntheticInputs(scene, cams[1], cams[0], w, h, sdf_res=16, shadow_res=256, froxel_depth=16, sun_direction=(0.35, -0.8, 0.45))
            inputs.upload(fp)
            out = []
            for f in range(n_frames):
                fp.frame(cams[f + 1], 1.0 / 60.0, 0.5 + f / 60.0)
                enabled, n_async = backend.getAsyncTail()
                assert enabled == (mode != "in order")
                # 5 downsamples + 5 upsamples + apply + tonemap
                assert n_async == (12 if enabled else 0), "executions on the tail stream: %d" % n_async
                if mode != "async, no host sync between frames" or f == n_frames - 1:
                    out.append([backend.downloadImage(fp.image(name), 0, np.uint8).copy() for name in names] +
                               [backend.downloadStorageBuffer(fp.storage_buffer("histogram"), 512, dtype=np.uint8).copy()])
            results[mode] = out
            fp.destroy()
    finally:
        backend.setAsyncTail(True)
        backend.setMathMode(False)
    for f in range(n_frames):
        for a, b, what in zip(results["async"][f], results["in order"][f], names + ["histogram"]):
            assert np.array_equal(a, b), "%s differs with the asynchronous tail, frame %d" % (what, f)
    for a, b, what in zip(results["async, no host sync between frames"][0], results["in order"][-1], names + ["histogram"]):
        assert np.array_equal(a, b), "%s differs after %d back-to-back frames with the asynchronous tail" % (what, n_frames)


@pytest.mark.gpu
def test_gpu_trace_through_bricked_volumes_changes_no_bit(backend, monkeypatch):
    """North star "SDF bricks", built in the form the trace could use (csrc/device/sdf_bricks.h): PLR_TRACE_BRICKS=1 makes the fast trace march through copies of
    the SDF volumes laid out in 128-byte bricks of 7 x 4 x 2 cells (a fetch touches 1.9 cache lines instead of 4). Same texels, same results: four frames of
    temporal feedback equal the default run byte for byte. (Measured slower - profiles/r04_not_kept.txt - and therefore not the default. Mode 2 makes a launch
    that cannot take the variant an error, so the equality below is about the variant.)"""
    from plainrenderer_amd.frame import FramePipeline, SyntheticInputs
    w, h, n_frames = 1280, 720, 4
    cams = [Camera.look((15.0 + 0.03 * i, -7.0, -6.0 + 0.05 * i), (0.0, 0.16, 1.0), aspect=w / h) for i in range(n_frames + 1)]
    scene = synth.SynthScene(grid=4, cell=8.0, seed_id=513)
    names = ["swapchain", "post1", "giHistoryYSH0", "giHistoryCoCg0", "taaHistory0", "color0", "color1"]
    inputs, results = None, {}
    try:
        backend.setMathMode(True)
        for mode in ("0", "2"):
            monkeypatch.setenv("PLR_TRACE_BRICKS", mode)
            fp = FramePipeline(backend, w, h, shadow_map_res=256, brdf_lut_res=LUT_RES, froxel_depth=16, max_sdf_instances=64)
            if inputs is None:
                inputs = SyntheticInputs(scene, cams[1], cams[0], w, h, sdf_res=16, shadow_res=256, froxel_depth=16, sun_direction=(0.35, -0.8, 0.45))
            inputs.upload(fp)
            out = []
            for f in range(n_frames):
                fp.frame(cams[f + 1], 1.0 / 60.0, 0.5 + f / 60.0)
                out.append([backend.downloadImage(fp.image(name), 0, np.uint8).copy() for name in names])
            results[mode] = out
            fp.destroy()
    finally:
        monkeypatch.delenv("PLR_TRACE_BRICKS", raising=False)
        backend.setMathMode(False)
    assert len(np.unique(results["0"][-1][2])) > 16, "the GI history is not a constant image"
    for f in range(n_frames):
        for a, b, what in zip(results["0"][f], results["2"][f], names):
            assert np.array_equal(a, b), "%s differs with bricked volumes, frame %d" % (what, f)


@pytest.mark.gpu
def test_gpu_a_frame_recorded_differently_cannot_read_an_image_the_last_frame_left_unwritten(backend):
    """at fusion level 2 the fused upscale + shade and the temporal GI filter leave images unwritten on the assumption that the next frame, recorded
    alike, overwrites them before anything reads them. The stale flag is persistent: a NEXT frame that is recorded differently and samples such an image fails
    loudly instead of reading an older frame's texels, the flag survives frames that do not touch the image, and a frame that writes it clears it"""
    from plainrenderer_amd.backend import ComputePassExecution, ImageResource, RenderPassResources
    from plainrenderer_amd.frame import FramePipeline, SyntheticInputs
    w, h = 648, 360
    cams = [Camera.look((15.0 + 0.03 * i, -7.0, -6.0 + 0.05 * i), (0.0, 0.16, 1.0), aspect=w / h) for i in range(4)]
    scene = synth.SynthScene(grid=4, cell=8.0, seed_id=512)
    backend.setMathMode(True)
    backend.setPassFusion(2)
    fp = FramePipeline(backend, w, h, shadow_map_res=256, brdf_lut_res=LUT_RES, froxel_depth=16, max_sdf_instances=64)
    try:
        SyntheticInputs(scene, cams[1], cams[0], w, h, sdf_res=16, shadow_res=256, froxel_depth=16, sun_direction=(0.35, -0.8, 0.45)).upload(fp)
        fp.frame(cams[1], 1.0 / 60.0, 0.5)
        probe = backend.createComputePass("tonemapping.comp", name="stale reader")
        for stale in ("giFullResYSH", "giHistoryYSH1"):
            for _ in range(2):  # the second time: the flag was not cleared by the failed frame, nor by a frame that does not touch the image
                backend.newFrame()
                exe = ComputePassExecution(handle=probe, resources=RenderPassResources(sampledImages=[ImageResource(fp.image(stale), 0, 1)],
                                                                                          storageImages=[ImageResource(fp.image("swapchain"), 0, 0)]), dispatchCount=(1, 1, 1))
                backend.setComputePassExecution(exe)
                with pytest.raises(RuntimeError, match="reads an image that was not written"):
                    backend.renderFrame()
        # the same frame as before: writes (or elides again) before it reads - no complaint; at fusion level 1 the images are written and readable
        fp.frame(cams[2], 1.0 / 60.0, 0.5)
        backend.setPassFusion(1)
        fp.frame(cams[3], 1.0 / 60.0, 0.5)
        assert backend.downloadImage(fp.image("giFullResYSH"), 0, np.uint8).any()
    finally:
        fp.destroy()
        backend.setPassFusion(2)
        backend.setMathMode(False)


@pytest.mark.gpu
def test_gpu_the_persistent_bloom_chain_changes_no_bit():
    """kernels_fast/bloom_fast.hip bloomChainKernel (measured and not kept: off unless PLR_BLOOM_CHAIN=1): runs of small bloom levels as one persistent launch per
    direction, the levels ordered by arrival counters. The fused-vs-unfused byte comparisons of this file, run in a process that turns it on."""
    import subprocess
    import sys
    env = dict(os.environ, PLR_BLOOM_CHAIN="1")
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.abspath(__file__), "-m", "gpu", "-q", "-x", "-k", "not persistent_bloom_chain"], env=env, capture_output=True, text=True,
                       timeout=1200, cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]


@pytest.mark.gpu
@pytest.mark.parametrize("w,h", [(1280, 720), (648, 360)])
def test_gpu_the_shade_as_two_launches_beside_the_gi_chain_stays_within_one_code(backend, w, h):
    """Early parts (include/plr.h plr_set_early_parts, backend.h EarlyPart): in a full frame the deferred shade's direct lighting is launched on the early stream
    at the frame's start - nothing recorded in front of the shade writes what it reads - and runs beside the frame front and the GI chain; the fused upscale + shade
    launch then only upscales, adds the indirect response and the fog, and packs. Frame 0 (no feedback yet): the colour target is within ONE R11G11B10 code of the
    single launch's on every pixel and every image that does not depend on it is byte-identical. Later frames feed the colour back (histogram -> exposure, TAA
    history) and the TAA resolve clips against a neighbourhood box: the resolved image stays within one code on all but a fraction of a per cent of the pixels."""
    from plainrenderer_amd.frame import FramePipeline, SyntheticInputs
    import parity
    n_frames = 4
    cams = [Camera.look((15.0 + 0.03 * i, -7.0, -6.0 + 0.05 * i), (0.0, 0.16, 1.0), aspect=w / h) for i in range(n_frames + 1)]
    scene = synth.SynthScene(grid=4, cell=8.0, seed_id=510)
    inputs = None
    results = {}
    try:
        backend.setMathMode(True)
        for early in (0, 1):
            backend.setEarlyParts(early)
            fp = FramePipeline(backend, w, h, shadow_map_res=256, brdf_lut_res=LUT_RES, froxel_depth=16, max_sdf_instances=64)
            if inputs is None:
                inputs = SyntheticInputs(scene, cams[1], cams[0], w, h, sdf_res=16, shadow_res=256, froxel_depth=16, sun_direction=(0.35, -0.8, 0.45))
            inputs.upload(fp)
            out = []
            for f in range(n_frames):
                fp.frame(cams[f + 1], 1.0 / 60.0, 0.5 + f / 60.0)
                level, launched = backend.getEarlyParts()
                assert (level, launched) == (early, early), "one early part per frame: the shade's direct lighting"
                assert backend.getGeneralKernelExecutions()[0] == 0
                color = backend.downloadImage(fp.image("color%d" % ((f + 1) % 2)), 0, np.uint32).copy()
                same = {n: backend.downloadImage(fp.image(n), 0, np.uint8).copy() for n in ("giHistoryYSH0", "giHistoryCoCg0", "pyramid", "depthHalfRes")}
                out.append((color, backend.downloadImage(fp.image("post1"), 0, np.uint32).copy(), same))
            results[early] = out
            fp.destroy()
    finally:
        backend.setEarlyParts(0)
        backend.setMathMode(False)
    c0, c1 = results[0][0][0], results[1][0][0]
    assert not np.array_equal(c0, np.zeros_like(c0))
    d = parity.r11g11b10_code_diff(c1, c0)
    assert d.max() <= 1, "frame 0: two launches against one, colour target"
    for f in range(n_frames):
        for name in results[0][f][2]:
            if f == 0 or name in ("pyramid", "depthHalfRes"):
                assert np.array_equal(results[0][f][2][name], results[1][f][2][name]), "%s differs in frame %d" % (name, f)
        dp = parity.r11g11b10_code_diff(results[1][f][1], results[0][f][1])
        # (the resolve clips the history to the neighbourhood's box, in tonemapped YCoCg: one code in a neighbour moves the box. Measured on MI355X: 2.3e-3 - 2.6e-3 of
        #  the pixels beyond one code in frame 0, never more than 3 codes)
        assert (dp > 1).any(axis=1).mean() <= 1e-2 and (f > 0 or dp.max() <= 6), "frame %d: resolved colour, %.2e of the pixels beyond one code, max %d" % (f, (dp > 1).any(axis=1).mean(), dp.max())


@pytest.mark.gpu
def test_gpu_a_host_upload_makes_an_unwritten_image_readable_again(backend):
    """the "left unwritten by a fused launch" flag of an image is persistent (a later frame recorded differently must not read stale texels silently);
    a HOST write - upload, raw copy or write into the allocation - gives the image contents of its own and clears it."""
    from plainrenderer_amd.frame import FramePipeline, SyntheticInputs
    w, h = 648, 360
    cams = [Camera.look((15.0 + 0.03 * i, -7.0, -6.0 + 0.05 * i), (0.0, 0.16, 1.0), aspect=w / h) for i in range(3)]
    scene = synth.SynthScene(grid=4, cell=8.0, seed_id=510)
    try:
        backend.setMathMode(True)
        backend.setPassFusion(2)
        fp = FramePipeline(backend, w, h, shadow_map_res=256, brdf_lut_res=LUT_RES, froxel_depth=16, max_sdf_instances=64)
        SyntheticInputs(scene, cams[1], cams[0], w, h, sdf_res=16, shadow_res=256, froxel_depth=16, sun_direction=(0.35, -0.8, 0.45)).upload(fp)
        fp.frame(cams[1], 1.0 / 60.0, 0.5)
        img = fp.image("giFullResYSH")
        with pytest.raises(RuntimeError, match="not written in the last frame"):
            backend.downloadImage(img, 0, np.uint8)
        fill = np.full(w * h * 8, 7, np.uint8)
        backend.uploadImage(img, fill)
        assert np.array_equal(backend.downloadImage(img, 0, np.uint8), fill)
        fp.frame(cams[2], 1.0 / 60.0, 0.5)  # the next frame elides it again
        with pytest.raises(RuntimeError, match="not written in the last frame"):
            backend.downloadImage(img, 0, np.uint8)
        fp.destroy()
    finally:
        backend.setMathMode(False)
