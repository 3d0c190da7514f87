"""Pass fusion (include/plr.h plr_set_pass_fusion, csrc/backend.h): the backend covers certain ADJACENT recorded executions with fewer kernel
launches. The boundary and the results are unchanged: every frame of a fused run equals the unfused run byte for byte."""
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
        for fusion in (True, False):
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
                imgs = [backend.downloadImage(fp.image(name), 0, np.uint8).copy() for name in names]
                out.append(imgs + [backend.downloadStorageBuffer(fp.storage_buffer("light"), 20, dtype=np.uint8).copy(),
                                   backend.downloadStorageBuffer(fp.storage_buffer("histogram"), 512, dtype=np.uint8).copy()])
            results[fusion] = (out, names + ["light buffer", "histogram"])
            fp.destroy()
    finally:
        backend.setPassFusion(True)
        backend.setMathMode(False)
    assert fused_counts[False] == 0
    assert fused_counts[True] >= 6, "executions inside fused launches: %d" % fused_counts[True]
    for f in range(n_frames):
        for a, b, what in zip(results[True][0][f], results[False][0][f], results[True][1]):
            assert np.array_equal(a, b), "%s differs with pass fusion, frame %d (%dx%d)" % (what, f, w, h)


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
    assert backend.getPassFusion() == (True, 0)
    fp.destroy()
