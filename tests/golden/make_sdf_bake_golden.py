"""Generates tests/golden/sdf_bake_*.npz: SDF volumes baked by the oracle (oracle/sdf_bake.cpp, the scalar restatement of the
reference's AssetPipeline/SceneSDF.cpp) from the procedural meshes in plainrenderer_amd/meshes.py. The reference itself cannot be
built or run in this environment and holds no golden volumes, so these pin the oracle's behaviour over time and
give the GPU bake a fixed target. Run from the repo root:  python tests/golden/make_sdf_bake_golden.py"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import pyoracle  # noqa: E402
from plainrenderer_amd import meshes  # noqa: E402

CASES = {
    "sphere": lambda: meshes.uv_sphere(2.0, 24, 12, centre=(0.3, -0.2, 0.1)),
    "box": lambda: meshes.box((1.5, 1.0, 2.0), centre=(0.0, 0.5, 0.0), subdiv=3),
    "torus": lambda: meshes.torus(2.5, 0.7, 28, 12, centre=(-1.0, 0.0, 2.0)),
}


def main():
    for name, make in CASES.items():
        pos, idx = make()
        mn, mx = meshes.bounds(pos)
        res = pyoracle.sdf_resolution(mn, mx)
        vol = pyoracle.sdf_bake(pos, idx, mn, mx, res)
        out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "sdf_bake_%s.npz" % name)
        np.savez_compressed(out, positions=pos, indices=idx, bb_min=mn, bb_max=mx, res=np.array(res, np.int32), volume=vol)
        print(name, res, vol.shape, os.path.getsize(out))


if __name__ == "__main__":
    main()
