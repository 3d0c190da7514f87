"""Debug helper: where does the fast upscale differ from the oracle? Run on the GPU box from the repo root."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "oracle"))
import passes
import test_sdfgi as T
from plainrenderer_amd import backend as B, pixfmt

W, H, TW, TH = T.W, T.H, T.TW, T.TH
s = T.scene.__wrapped__() if hasattr(T.scene, "__wrapped__") else T.scene.__pytest_wrapped__.obj()
be = B.RenderBackend(1920, 1080, device=0)
gp = s.g.pack()
r = np.random.default_rng(1)
yy, xx = np.mgrid[0:TH, 0:TW]
y = pixfmt.pack_half(np.stack([xx, yy, xx * 0 + 1, xx * 0 + 2], axis=2).astype(np.float32))
c = pixfmt.pack_half(r.uniform(-1, 1, (TH, TW, 2)).astype(np.float32))
hd = passes.orc_depth_downscale(s.gb["depth"], W, H)
print("half depth", None if hd is None else hd.shape)
ua = (y, c, TW, TH, s.gb["depth"], hd, W, H, gp)
yo, co = passes.orc_gi_upscale(*ua)
for fast in (False, True):
    be.setMathMode(fast)
    yg, cg = passes.gpu_gi_upscale(be, *ua)
    a, b = pixfmt.unpack_half(yg).reshape(H, W, 4), pixfmt.unpack_half(yo).reshape(H, W, 4)
    err = np.abs(a - b).max(axis=2)
    bad = err > 2.0 ** -8 * np.abs(b).max(axis=2)
    print("fast" if fast else "exact", "bad pixels", int(bad.sum()), "of", bad.size)
    ys, xs = np.nonzero(bad)
    if os.environ.get("DBG_PX"):
        pts = [tuple(int(v) for v in t.split(",")) for t in os.environ["DBG_PX"].split(";")]
        xs, ys = np.array([p[0] for p in pts]), np.array([p[1] for p in pts])
    if len(ys):
        print(" x parity", np.bincount(xs & 1, minlength=2), "y parity", np.bincount(ys & 1, minlength=2))
        print(" x range", xs.min(), xs.max(), "y range", ys.min(), ys.max())
        hdf = pixfmt.unpack_half(hd).reshape(TH, TW)
        for i in range(min(16, len(ys))):
            X, Y = xs[i], ys[i]
            k, m = X // 2, Y // 2
            print("  px", X, Y, "k,m", k, m, "got texel", a[Y, X][:2], "ref texel", b[Y, X], "got", a[Y, X], "full depth", s.gb["depth"].reshape(H, W)[Y, X],
                  "half depths rows m-1..m+1, cols k-1..k+1", hdf[m - 1:m + 2, k - 1:k + 2].tolist())
