"""Generates tests/golden/frame_96x54.npz: a two-frame run of the whole hot path (exposure, HiZ, SDF GI, shading, TAA, bloom,
tonemap) on a small synthetic scene. Inputs come from plainrenderer_amd.synth; the per-frame global UBO / TAA resolve weights /
culling frustum are the bytes the C++ host mirror (csrc/frontend) submitted, which is why this script needs a GPU box; the
expected outputs are the ORACLE's (oracle/*.cpp, the scalar restatement of the reference shaders). The reference cannot be built
or run here and holds no golden images, so this fixture pins the oracle and the host logic over time.

    gpurun -- 'python tests/golden/make_frame_golden.py gpurun_out/frame_96x54.npz'   then copy the file into tests/golden/"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)

W, H, LUT = 96, 54, 16
N_FRAMES = 2
FP_ARGS = dict(shadow_map_res=128, brdf_lut_res=LUT, froxel_depth=8, max_sdf_instances=64)


def cameras():
    from plainrenderer_amd.scene import Camera
    return [Camera.look((15.0 + 0.03 * i, -7.0, -6.0 + 0.05 * i), (0.0, 0.16, 1.0), aspect=W / H) for i in range(N_FRAMES + 1)]


def frame_times(f):
    return 1.0 / 60.0, 0.5 + f / 60.0


def main(out_path):
    from oracle_frame import OracleFrame
    from plainrenderer_amd import RenderBackend, synth
    from plainrenderer_amd.frame import FramePipeline, SyntheticInputs
    cams = cameras()
    be = RenderBackend(W, H, device=0)
    be.setMathMode(False)
    scene = synth.SynthScene(grid=4, cell=8.0, seed_id=600)
    fp = FramePipeline(be, W, H, **FP_ARGS)
    inputs = SyntheticInputs(scene, cams[1], cams[0], W, H, sdf_res=16, shadow_res=128, froxel_depth=8, sun_direction=(0.35, -0.8, 0.45))
    inputs.upload(fp)
    d = {"in_" + k: v for k, v in inputs.to_arrays().items()}
    d["volume_indices"] = np.asarray(inputs.volume_indices, np.int64)
    d["settings"] = np.frombuffer(bytes(fp.settings), np.uint8)
    ora = OracleFrame(inputs, W, H, LUT, fp.settings)
    for f in range(N_FRAMES):
        dt, t = frame_times(f)
        fp.frame(cams[f + 1], dt, t)
        g = fp.submitted_globals()
        frustum = be.downloadUniformBuffer(fp.uniform_buffer("sdfCameraFrustum"), 192).tobytes()
        influence = float(be.downloadUniformBuffer(fp.uniform_buffer("sdfInfluenceRange"), 4, dtype=np.float32)[0])
        weights = np.asarray(fp.resolve_weights(), np.float32)
        ora.frame(g, weights, frustum, influence)
        d["f%d_globals" % f] = np.frombuffer(bytes(g), np.uint8)
        d["f%d_frustum" % f] = np.frombuffer(frustum, np.uint8)
        d["f%d_influence" % f] = np.array([influence], np.float32)
        d["f%d_weights" % f] = weights
        d["f%d_light" % f] = np.frombuffer(ora.light, np.uint8)
        d["f%d_hist" % f] = ora.hist.copy()
        d["f%d_hiz4" % f] = np.asarray(ora.hiz[4]).copy()
        d["f%d_tiles" % f] = ora.tiles.copy()
        d["f%d_gi_full_y" % f] = ora.full_y.copy()
        d["f%d_color" % f] = ora.color[ora.rt_index].copy()
        d["f%d_post1" % f] = ora.post1.copy()
        d["f%d_swapchain" % f] = ora.swapchain.copy()
    fp.destroy()
    be.shutdown()
    np.savez_compressed(out_path, **d)
    print("wrote", out_path, os.path.getsize(out_path))


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else os.path.join(os.path.dirname(os.path.abspath(__file__)), "frame_96x54.npz"))
