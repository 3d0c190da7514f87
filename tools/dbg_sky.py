import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np
import passes, test_producers as tp
from plainrenderer_amd import RenderBackend, pixfmt
be = RenderBackend(256, 144, device=0); be.setMathMode(False)
atm, light, gp = tp._sky_inputs()
a = passes.gpu_sky_luts(be, atm, light, gp); b = passes.orc_sky_luts(atm, light, gp)
for x, y, what in zip(a, b, ("transmission", "multiscatter", "sky")):
    d = x != y
    print(what, d.sum(), "of", d.size)
    if d.any():
        iy, ix = np.nonzero(d)
        for k in range(min(5, len(iy))):
            print("  ", iy[k], ix[k], pixfmt.unpack_r11g11b10(x[iy[k], ix[k]:ix[k]+1]), pixfmt.unpack_r11g11b10(y[iy[k], ix[k]:ix[k]+1]))
