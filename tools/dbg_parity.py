"""Where do pixels with EQUAL decision signatures still differ? (diagnostic for tests/test_parity_fullsize.py)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import torch  # noqa
import numpy as np
os.environ.setdefault("PLR_PARITY_SIZE", "1920x1088")
import parity, passes
import test_parity_fullsize as T
from plainrenderer_amd import RenderBackend, pixfmt

be = RenderBackend(1920, 1080, device=0)
fs = T.build_state(be)
W, H, TW, TH, U = T.W, T.H, T.TW, T.TH, T.U

# ---- trace
c = fs.cap["trace"]
args = (fs.gb["depth"], fs.gb["normal"], W, H, TW, TH, fs.inputs.sky, 200, 100, c["light"], fs.inputs.instance_bytes_patched, c["tiles"], 5.0, fs.inputs.shadow_info,
        fs.inputs.shadow_maps[c["cascade"]], fs.inputs.shadow_res, fs.gp)
with passes.gpu_signature(be, TW * TH) as sg:
    yg, cg = passes.gpu_sdf_trace(be, *args, strict=True, cascade=c["cascade"])
arr, n = fs.ora._bindless(passes.orc.global_from_bytes(fs.gp))
with passes.orc_signature(TW * TH) as so:
    yo, co = passes.orc_sdf_trace(*args, arr, n, strict=True, cascade=c["cascade"])
x = sg.words ^ so.words
ray_flip = (x & ~np.uint32(0x7F8)).reshape(TH, TW) != 0
take_flip = (x & np.uint32(0x7F8)).reshape(TH, TW) != 0
touched = (parity.dilate3x3(ray_flip) | take_flip).reshape(-1)
got = np.concatenate([U(yg).reshape(-1, 4), U(cg).reshape(-1, 2)], axis=1).astype(np.float64)
ref = np.concatenate([U(yo).reshape(-1, 4), U(co).reshape(-1, 2)], axis=1).astype(np.float64)
err = np.abs(got - ref)
tol = np.maximum(2.0 ** -7 * np.abs(ref), 2.0 ** -10 * np.abs(ref).max())
bad = (err > tol)
hit = (so.words & 1).astype(bool)
# does any neighbour in the group (own or taken) hit? classify by own ray only
print("trace: clean pixels %d; violating clean %d; of those own-ray hit %d, own-ray sky %d" % ((~touched).sum(), (bad.any(1) & ~touched).sum(), (bad.any(1) & ~touched & hit).sum(),
                                                                                                (bad.any(1) & ~touched & ~hit).sum()))
for ch in range(6):
    b = bad[:, ch] & ~touched
    if b.any():
        rel = err[b, ch] / np.maximum(np.abs(ref[b, ch]), 1e-12)
        print("  channel %d: %d bad, |ref| median %.3g, err median %.3g max %.3g, rel err median %.3g max %.3g" % (ch, b.sum(), np.median(np.abs(ref[b, ch])), np.median(err[b, ch]), err[b, ch].max(),
                                                                                                                  np.median(rel), rel.max()))
i = np.argmax(np.where(~touched, err.max(1), 0))
print("  worst clean pixel %d (x=%d y=%d): got %s ref %s sig %08x/%08x" % (i, i % TW, i // TW, got[i], ref[i], sg.words[i], so.words[i]))
sky_clean = ~touched & ~hit
takeall = ((so.words >> 3) & 0xff)
print("  Y (ch0*2) rel err over clean sky-ray pixels with no neighbours taken: ", end="")
sel = sky_clean & (takeall == 0)
if sel.any():
    r0 = err[sel, 0] / np.maximum(np.abs(ref[sel, 0]), 1e-12)
    print("n=%d median %.3g p99 %.3g max %.3g" % (sel.sum(), np.median(r0), np.percentile(r0, 99), r0.max()))
sel = ~touched & hit & (takeall == 0)
if sel.any():
    r0 = err[sel, 0] / np.maximum(np.abs(ref[sel, 0]), 1e-12)
    print("  same for hit-ray pixels: n=%d median %.3g p99 %.3g max %.3g" % (sel.sum(), np.median(r0), np.percentile(r0, 99), r0.max()))

# ---- shade
c, s = fs.cap["shade"], fs.settings
args = (fs.gb, W, H, fs.ora.brdf_lut, 512, c["light"], fs.inputs.shadow_info, fs.inputs.shadow_maps, fs.inputs.shadow_res, c["gi"][0], c["gi"][1], fs.inputs.froxel,
        fs.inputs.froxel_dims, fs.inputs.vol_settings, fs.inputs.sky, fs.gp)
var = (int(s.diffuse_brdf), int(s.direct_multiscatter), bool(s.use_geometry_aa), int(s.indirect_lighting_tech), int(s.sun_shadow_cascade_count))
with passes.gpu_signature(be, W * H) as sg:
    got = passes.gpu_deferred_shading(be, *args, *var)
with passes.orc_signature(W * H) as so:
    ref = passes.orc_deferred_shading(*args, arr, n, *var)
flip = sg.words != so.words
d = parity.r11g11b10_code_diff(got, ref)
clean = ~flip
sky = (so.words & 128) != 0
print("shade: clean sky pixels: max code diff %d, >1: %d of %d; clean geometry pixels: max %d, >1: %d of %d" % (d[clean & sky].max(initial=0), (d[clean & sky] > 1).any(1).sum(), (clean & sky).sum(),
                                                                                                          d[clean & ~sky].max(initial=0), (d[clean & ~sky] > 1).any(1).sum(), (clean & ~sky).sum()))
gv, rv = pixfmt.unpack_r11g11b10(got).reshape(-1, 3), pixfmt.unpack_r11g11b10(ref).reshape(-1, 3)
w = np.where(clean & ~sky, d.max(1), 0)
for i in np.argsort(-w)[:8]:
    lit = (so.words[i] >> 2) & 15
    print("  geometry pixel x=%d y=%d codes diff %s got %s ref %s lit %d cascade %d depth %.6f spec %s albedo %s" % (i % W, i // W, d[i], gv[i], rv[i], lit, so.words[i] & 3, fs.gb["depth"].reshape(-1)[i],
                                                                                                             fs.gb["specular"].reshape(-1, 4)[i], fs.gb["albedo"].reshape(-1, 4)[i]))
w = np.where(clean & sky, d.max(1), 0)
for i in np.argsort(-w)[:4]:
    print("  sky pixel x=%d y=%d codes diff %s got %s ref %s" % (i % W, i // W, d[i], gv[i], rv[i]))
lit = (so.words >> 2) & 15
for lo, hi, name in ((0, 0, "fully shadowed"), (1, 11, "penumbra"), (12, 12, "fully lit")):
    sel = clean & ~sky & (lit >= lo) & (lit <= hi)
    if sel.any():
        print("  %s: n=%d, >1 code: %d, max %d" % (name, sel.sum(), (d[sel] > 1).any(1).sum(), d[sel].max()))
be.shutdown()
