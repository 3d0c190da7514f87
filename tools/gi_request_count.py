"""VERDICT r05 item 2, first step: how large would a REQUEST-LIST GI exchange be? For the 7680 x 4320 frame as 2 x 2 tiles and as 4 row bands, per exchange point (the
traced GI in front of spatial filter 0, the temporally filtered GI in front of spatial filter 1): how many DISTINCT half-resolution GI texels outside
"own rectangle + 128 trace rows / columns" do the 32 disc samples of a rank's pixels land on (filterIndirectDiffuseSpatial.comp:53-105)? The sample positions depend
only on depth, camera and frame index, all known before the trace finishes - this script evaluates them with torch from the unpartitioned frame's half-resolution
depth image and global uniform block (one GPU; float32 tensor arithmetic, the shader's statements) and counts texels, per rank and per peer.
    python tools/gi_request_count.py [frames]
A texel costs 16 bytes in the packed form the filter gathers. For comparison: the whole half-resolution image is 3840 x 2160 = 8.29 M texels = 133 MB per exchange
point, and the exact mode (band_gi_halo = PLRF_HALO_WHOLE_IMAGE) moves 151 - 174 MB per rank and frame over both points (profiles/r06_config5_native_exchange.txt)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import bench
from plainrenderer_amd import RenderBackend, tiling
from plainrenderer_amd.frame import FramePipeline

W, H = (int(v) for v in os.environ.get("PLR_REQUEST_COUNT_SIZE", "7680x4320").split("x"))
FRAMES = int(sys.argv[1]) if len(sys.argv) > 1 else 4
HALO = 64 * ((H + 2159) // 2160)  # trace texels: plrf_default_settings band_gi_halo
dev = "cuda:0"


def wang_hash(seed):
    seed = ((seed ^ 61) ^ (seed >> 16)) & 0xffffffff
    seed = (seed * 9) & 0xffffffff
    seed ^= seed >> 4
    seed = (seed * 0x27d4eb2d) & 0xffffffff
    seed ^= seed >> 15
    return seed


def sample_table(key):
    """noise.inc:28-54: 32 x (sqrt(r0), cos(2 pi r1), sin(2 pi r1)) of the RNG seeded with wang_hash(frameIndexMod4 + filterIndex)"""
    s = wang_hash(key)
    out = []
    def rand():
        nonlocal s
        s ^= (s << 13) & 0xffffffff; s ^= s >> 17; s ^= (s << 5) & 0xffffffff
        return min(max(np.float32(s) * np.float32(2.3283064365386963e-10 * (1 + 2.0 ** -22)), 0.0), 1.0)
    for _ in range(32):
        r0, r1 = rand(), rand()
        out.append((np.sqrt(r0), np.cos(2 * np.pi * r1), np.sin(2 * np.pi * r1)))
    return np.array(out, np.float32)


class A: pass
args = A(); args.grid = 16; args.sdf_res = 64; args.shadow_res = 2048; args.steps = FRAMES + 2; args.warmup = 0; args.profile_frames = 0
be = RenderBackend(W, H, device=0)
fp = FramePipeline(be, W, H, shadow_map_res=2048)
scene, cams, inputs = bench.build_scene(args, dev, W, H)
inputs.upload(fp)
W2, H2 = W // 2, H // 2
grids = {"tiles 2x2": tiling.tile_rects(W, H, 2, 2), "4 row bands": tiling.tile_rects(W, H, 1, 4)}
totals = {}
for f in range(FRAMES):
    fp.frame(cams[f + 1], 1 / 60, 0.5 + f / 60)
    g = np.frombuffer(fp.submitted_globals(), np.uint8)
    fl = lambda off, n=1: torch.tensor(g[off:off + 4 * n].view(np.float32).copy(), device=dev)
    vp = fl(0, 16)  # column major
    cam_pos, right, up, fwd = fl(144, 3), fl(176, 3), fl(192, 3), fl(208, 3)
    tan_h, aspect, near, far = (float(g[o:o + 4].view(np.float32)[0]) for o in (280, 284, 288, 292))
    mod4 = int(g[336:340].view(np.uint32)[0])
    depth = torch.tensor(be.downloadImage(fp.image("depthHalfRes"), 0, np.float16).reshape(H2, W2).astype(np.float32), device=dev)

    def pixel_to_world(u, v):
        x = (u * W2).floor().clamp(0, W2 - 1).long(); y = (v * H2).floor().clamp(0, H2 - 1).long()
        d = depth[y, x]
        lin = near * far / (far + (1.0 - d) * (near - far))
        ndx, ndy = u * 2 - 1, v * 2 - 1
        V = -fwd[None, :] + (tan_h * ndy)[:, None] * up[None, :] - (tan_h * aspect * ndx)[:, None] * right[None, :]
        c2p = -V / V.norm(dim=1, keepdim=True)
        return cam_pos[None, :] + c2p / (c2p @ fwd)[:, None] * lin[:, None]

    for name, rects in grids.items():
        for flt, radius in ((0, 1.5), (1, 1.0)):
            tab = sample_table(mod4 + flt)
            for rank, (x0, y0, x1, y1) in enumerate(rects):
                hx0, hy0, hx1, hy1 = x0 // 2, y0 // 2, (x1 + 1) // 2, (y1 + 1) // 2
                ys, xs = torch.meshgrid(torch.arange(hy0, hy1, device=dev), torch.arange(hx0, hx1, device=dev), indexing="ij")
                xs, ys = xs.reshape(-1), ys.reshape(-1)
                u0, v0 = (xs.float() + 0.5) / W2, (ys.float() + 0.5) / H2
                pc = pixel_to_world(u0, v0)
                T = pc - pixel_to_world(u0 + 1.0 / W2, v0); T = T / T.norm(dim=1, keepdim=True)
                B = pc - pixel_to_world(u0, v0 + 1.0 / H2); B = B / B.norm(dim=1, keepdim=True)
                bitmap = torch.zeros(H2 * W2, dtype=torch.bool, device=dev)
                length = torch.ones_like(u0)
                samples_on_screen = 0
                for i in range(32):
                    d = float(tab[i, 0]) * length
                    ox, oy = float(tab[i, 1]) * d, float(tab[i, 2]) * d
                    sw = pc + radius * (ox[:, None] * T + oy[:, None] * B)
                    cx = vp[0] * sw[:, 0] + vp[4] * sw[:, 1] + vp[8] * sw[:, 2] + vp[12]
                    cy = vp[1] * sw[:, 0] + vp[5] * sw[:, 1] + vp[9] * sw[:, 2] + vp[13]
                    cw = vp[3] * sw[:, 0] + vp[7] * sw[:, 1] + vp[11] * sw[:, 2] + vp[15]
                    su, sv = cx / cw * 0.5 + 0.5, cy / cw * 0.5 + 0.5
                    su = torch.where((su < 0) | (su > 1), u0 - ox, su)
                    sv = torch.where((sv < 0) | (sv > 1), v0 - oy, sv)
                    off = (su < 0) | (su > 1) | (sv < 0) | (sv > 1) | ~torch.isfinite(su) | ~torch.isfinite(sv)
                    length = torch.where(off, length * 0.98, length)
                    tx = (su * W2).floor().clamp(0, W2 - 1).long(); ty = (sv * H2).floor().clamp(0, H2 - 1).long()
                    idx = (ty * W2 + tx)[~off]
                    bitmap[idx] = True
                    samples_on_screen += int((~off).sum())
                bm = bitmap.view(H2, W2)
                inside = torch.zeros_like(bm)
                inside[max(hy0 - HALO, 0):min(hy1 + HALO, H2), max(hx0 - HALO, 0):min(hx1 + HALO, W2)] = True
                beyond = int((bm & ~inside).sum())
                per_peer = []
                for p, (px0, py0, px1, py1) in enumerate(rects):
                    if p != rank:
                        peer = torch.zeros_like(bm)
                        peer[py0 // 2:(py1 + 1) // 2, px0 // 2:(px1 + 1) // 2] = True
                        per_peer.append((p, int((bm & peer & ~inside).sum()), int((bm & peer).sum())))
                halo_texels = int(inside.sum()) - (hy1 - hy0) * (hx1 - hx0)
                key = (name, flt, rank)
                t = totals.setdefault(key, dict(beyond=[], halo=halo_texels, all_outside=[], pixels=(hy1 - hy0) * (hx1 - hx0), per_peer=[]))
                t["beyond"].append(beyond)
                t["all_outside"].append(sum(pp[2] for pp in per_peer))
                t["per_peer"].append(per_peer)
fp.destroy(); be.shutdown()
print("# %dx%d, %d frames (camera path of bench.py), half-resolution GI %dx%d = %.2f M texels; bounded halo = %d trace texels around the rectangle; 16 B per packed texel" % (W, H, FRAMES, W2, H2, W2 * H2 / 1e6, HALO))
print("# distinct texels a rank's samples land on OUTSIDE its rectangle: 'beyond the halo' = what a request list would add to the bounded-halo exchange, 'all outside' = what a request list would move instead of it")
for name in grids:
    for flt in (0, 1):
        print("## %s, spatial filter %d (%s, disc radius %.1f m)" % (name, flt, "input: traced GI" if flt == 0 else "input: temporally filtered GI", 1.5 if flt == 0 else 1.0))
        for rank in range(4):
            t = totals[(name, flt, rank)]
            b, a = np.mean(t["beyond"]), np.mean(t["all_outside"])
            peers = {}
            for frame in t["per_peer"]:
                for p, be_, al in frame:
                    peers.setdefault(p, []).append(al)
            print("  rank %d (%.2f M pixels): beyond the halo %.3f M texels (%.1f MB; min %.3f max %.3f over the frames), all outside %.3f M texels (%.1f MB) = %.0f %% of the rest of the image; "
                  "the bounded halo itself %.3f M texels (%.1f MB); by peer [M texels]: %s" % (
                      rank, t["pixels"] / 1e6, b / 1e6, b * 16 / 1e6, min(t["beyond"]) / 1e6, max(t["beyond"]) / 1e6, a / 1e6, a * 16 / 1e6,
                      100.0 * a / max(W2 * H2 - t["pixels"], 1), t["halo"] / 1e6, t["halo"] * 16 / 1e6, ", ".join("%d: %.3f" % (p, np.mean(v) / 1e6) for p, v in sorted(peers.items()))))
    both = [sum(np.mean(totals[(name, flt, rank)]["all_outside"]) for flt in (0, 1)) for rank in range(4)]
    print("  %s: request lists of both exchange points, per rank and frame: %s MB received (worst rank %.1f MB); the whole-image exchange of the exact mode receives %.1f MB per rank" % (
        name, ", ".join("%.1f" % (v * 16 / 1e6) for v in both), max(both) * 16 / 1e6, 2 * (W2 * H2 - min(totals[(name, 0, r)]["pixels"] for r in range(4))) * 16 / 1e6))
