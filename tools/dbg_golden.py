import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np
from golden import make_frame_golden as gen
from plainrenderer_amd import RenderBackend
from plainrenderer_amd.frame import FramePipeline
import test_golden_frame as t
d, inputs, settings = t.load()
be = RenderBackend(1920, 1080, device=0)
be.setMathMode(False)
fp = FramePipeline(be, gen.W, gen.H, **gen.FP_ARGS)
inputs.upload(fp)
cams = gen.cameras()
for f in range(2):
    dt, tt = gen.frame_times(f)
    fp.frame(cams[f + 1], dt, tt)
    a = np.frombuffer(bytes(fp.submitted_globals()), np.uint8); b = d["f%d_globals" % f]
    diff = np.nonzero(a != b)[0]
    print(f, diff)
    print(np.frombuffer(a.tobytes(), np.float32)[diff // 4], np.frombuffer(b.tobytes(), np.float32)[diff // 4])
    fr = np.frombuffer(be.downloadUniformBuffer(fp.uniform_buffer("sdfCameraFrustum"), 192).tobytes(), np.float32)
    fb = np.frombuffer(d["f%d_frustum" % f].tobytes(), np.float32)
    print("frustum diff idx", np.nonzero(fr.view(np.uint32) != fb.view(np.uint32))[0]); print(fr[:8], fb[:8])
