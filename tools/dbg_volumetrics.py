"""Debug helper: where does the GPU volumetrics chain differ from the oracle? Run on the GPU box from the repo root."""
import os, sys
import numpy as np
root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
for d in ("tests", "oracle", "."):
    sys.path.insert(0, os.path.join(root, d))
import passes
import test_producers as T
from plainrenderer_amd import RenderBackend, pixfmt
be = RenderBackend(1920, 1080, device=0)
be.setMathMode(False)
args = T._volumetric_inputs()
a = passes.gpu_volumetrics(be, *args)
b = passes.orc_volumetrics(*args)
for x, y, what in zip(a, b, ("material", "scattering", "reprojection", "integration")):
    fx, fy = pixfmt.unpack_half(x).reshape(-1, 4), pixfmt.unpack_half(y).reshape(-1, 4)
    bad = np.nonzero((x.reshape(-1, 4) != y.reshape(-1, 4)).any(axis=1))[0]
    print(what, "texels differing:", bad.size, "of", fx.shape[0])
    for i in bad[:6]:
        print("   ", i, fx[i], fy[i], x.reshape(-1, 4)[i], y.reshape(-1, 4)[i])
