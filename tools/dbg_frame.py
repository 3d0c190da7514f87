import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests")); sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")); sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "oracle"))
import numpy as np
import passes
from plainrenderer_amd import pixfmt, synth, RenderBackend
from plainrenderer_amd.scene import Camera
from plainrenderer_amd.frame import FramePipeline, SyntheticInputs
from oracle_frame import OracleFrame
W,H=256,144
be = RenderBackend(W,H)
cams=[Camera.look((15.0 + 0.03 * i, -7.0, -6.0 + 0.05 * i), (0.0, 0.16, 1.0), aspect=W / H) for i in range(4)]
scene = synth.SynthScene(grid=4, cell=8.0, seed_id=500)
fp = FramePipeline(be, W, H, shadow_map_res=256, brdf_lut_res=32, froxel_depth=16, max_sdf_instances=64, sdf_half_res_trace=1, run_bloom=int(os.environ.get("BLOOM","1")))
inputs = SyntheticInputs(scene, cams[1], cams[0], W, H, sdf_res=16, shadow_res=256, froxel_depth=16, sun_direction=(0.35, -0.8, 0.45))
inputs.upload(fp)
ora = OracleFrame(inputs, W, H, 32, fp.settings)
for f in range(2):
    fp.frame(cams[f+1], 1/60., 0.5+f/60.)
    g = fp.submitted_globals()
    frustum = be.downloadUniformBuffer(fp.uniform_buffer("sdfCameraFrustum"), 192).tobytes()
    infl = float(be.downloadUniformBuffer(fp.uniform_buffer("sdfInfluenceRange"), 4, dtype=np.float32)[0])
    ora.frame(g, fp.resolve_weights(), frustum, infl)
    cur = ora.rt_index
    a = be.downloadImage(fp.image("post1"),0,np.uint32); b = ora.post1.reshape(-1)
    c = be.downloadImage(fp.image("color%d"%cur),0,np.uint32)
    print("frame",f,"color equal", np.array_equal(c, ora.color[cur].reshape(-1)), "post1 diff count", (a!=b).sum(), "of", a.size)
    d = np.nonzero(a!=b)[0][:8]
    for i in d:
        print("  px", i%W, i//W, pixfmt.unpack_r11g11b10(a[i:i+1]), pixfmt.unpack_r11g11b10(b[i:i+1]), "color", pixfmt.unpack_r11g11b10(c[i:i+1]))
    ha = be.downloadImage(fp.image("taaHistory%d" % ((ora.cpu_frame+1)%2)),0,np.uint32)
    print("  taa hist equal", np.array_equal(ha, ora.taa_hist[(ora.cpu_frame+1)%2].reshape(-1)))
    col = pixfmt.unpack_r11g11b10(c); print("  color stats", col.min(), col.max(), np.isnan(col).sum())
