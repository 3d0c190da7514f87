"""Child process of tests/test_spatial_walk.py: both spatial GI filter passes over seeded random inputs of the size given on the command line; prints one SHA-256 per pass.
The walk of the filter's blocks over the tiles (device/xcd.h) is chosen by PLR_SPATIAL_SPLIT_X / PLR_SPATIAL_CHUNKS, which the library reads once per process."""
import hashlib
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.join(os.path.dirname(HERE), "oracle"))  # tests/passes.py imports the checker's bindings; nothing of it is called here
import passes  # noqa: E402
from plainrenderer_amd import RenderBackend  # noqa: E402
from plainrenderer_amd.scene import Camera, GlobalShaderInfo  # noqa: E402
from util import F  # noqa: E402

tw, th = int(sys.argv[1]), int(sys.argv[2])
w, h = 2 * tw, 2 * th
rng = np.random.default_rng(20260930)
cam = Camera.look((16.0, -7.0, -6.0), (0.0, 0.16, 1.0), aspect=w / h)
g = GlobalShaderInfo()
cam.fill_global(g, w, h)
ysh = rng.random(th * tw * 4, dtype=np.float32).astype(np.float16).view(np.uint16)
cocg = (rng.random(th * tw * 2, dtype=np.float32) - 0.5).astype(np.float16).view(np.uint16)
# depth: a slope with steps (silhouettes) and noise, inside (0, 1); normals: mostly up, perturbed
yy, xx = np.mgrid[0:th, 0:tw].astype(np.float32)
depth = 0.15 + 0.7 * (yy / th) + 0.1 * ((xx // 97) % 2) + 0.02 * rng.random((th, tw), dtype=np.float32)
depth = np.clip(depth, 0.01, 0.99).astype(np.float16)
n = rng.normal(0.0, 0.25, (h, w, 3)).astype(np.float32) + np.array([0.0, -1.0, 0.0], np.float32)
n /= np.linalg.norm(n, axis=2, keepdims=True)
normal = np.concatenate([np.round((n * 0.5 + 0.5) * 255.0).astype(np.uint8), np.full((h, w, 1), 255, np.uint8)], axis=2)
be = RenderBackend(w, h, device=0)
for fi in (0, 1):
    y, c = passes.gpu_gi_spatial(be, ysh, cocg, tw, th, depth, F.R16_sFloat, tw, th, normal, w, h, g.pack(), fi)
    v = np.ascontiguousarray(y).view(np.float16).astype(np.float32)
    print("pass %d %s finite %.4f nonzero %.4f" % (fi, hashlib.sha256(np.ascontiguousarray(y).tobytes() + np.ascontiguousarray(c).tobytes()).hexdigest(),
                                                     float(np.isfinite(v).mean()), float((v != 0).mean())))
