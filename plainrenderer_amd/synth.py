"""Seeded synthetic inputs for the frame pipeline: an analytic scene (ground plane + a grid of sphere / box
instances), its G-buffer (depth D32 reverse-Z, world normals RGBA8, motion RG16_sNorm, albedo / specular RGBA8), per
instance SDF volumes (R16F), sun shadow cascades (D16 + ShadowCascadeInfo), sky / transmission LUTs, froxel volume and
blue-noise stand-ins. torch is used as an array library only (CPU for the small parity cases, the GPU for 4K inputs).

World convention = the reference's: camera up is -y, so the ground is the plane y = 0 and objects live at y < 0.
"""
import math
import struct
from dataclasses import dataclass

import numpy as np
import torch

from . import pixfmt
from .scene import Camera, to_glm

SEED_BASE = 0x504C4149  # "PLAI"


def _gen(seed_id, device="cpu"):
    g = torch.Generator(device="cpu")
    g.manual_seed(SEED_BASE + seed_id)
    return g


def pad_sdf_bounding_box(bb_min, bb_max):
    """padSDFBoundingBox, Plain/src/Common/sdfUtilities.cpp:5-19"""
    padding = np.maximum(0.075 * (bb_max - bb_min), 0.5)
    return bb_min - padding, bb_max + padding


@dataclass
class Instances:
    center: np.ndarray      # (n,3) world position of the object centre
    half: np.ndarray        # (n,3) half extents (sphere: r,r,r)
    kind: np.ndarray        # (n,) 0 sphere, 1 box
    yaw: np.ndarray         # (n,) rotation about y
    albedo: np.ndarray      # (n,3) sRGB-encoded mean albedo


class SynthScene:
    def __init__(self, grid=16, cell=8.0, seed_id=100, device=None):
        # device None: the GPU when there is one (the ray-marched G-buffer and shadow maps of a test scene take seconds on the host cores, and the GPU test
        # suite builds dozens of them), else the CPU. The scene is the same up to the last bit of the torch kernels' sin / sqrt.
        if device is None:
            device = "cuda:0" if torch.cuda.is_available() else "cpu"
        self.grid, self.cell, self.device = grid, cell, torch.device(device)
        r = np.random.default_rng(SEED_BASE + seed_id)
        n = grid * grid
        kind = (r.random(n) < 0.5).astype(np.int64)
        half = np.zeros((n, 3), np.float32)
        rad = r.uniform(1.2, 2.5, n).astype(np.float32)
        box = r.uniform(0.8, 2.1, (n, 3)).astype(np.float32)
        half[kind == 0] = np.repeat(rad[kind == 0, None], 3, 1)
        half[kind == 1] = box[kind == 1]
        yaw = np.where(kind == 1, r.uniform(0, 2 * math.pi, n), 0.0).astype(np.float32)
        ix, iz = np.meshgrid(np.arange(grid), np.arange(grid), indexing="xy")
        cx = (ix.reshape(-1) + 0.5) * cell + r.uniform(-0.8, 0.8, n)
        cz = (iz.reshape(-1) + 0.5) * cell + r.uniform(-0.8, 0.8, n)
        cy = -half[:, 1] + 0.05  # slightly sunk into the ground
        center = np.stack([cx, cy, cz], 1).astype(np.float32)
        albedo = r.uniform(0.2, 0.8, (n, 3)).astype(np.float32)
        self.inst = Instances(center, half, kind, yaw, albedo)
        t = lambda a, dt=torch.float32: torch.as_tensor(a, dtype=dt, device=self.device)
        self._center, self._half, self._kind, self._yaw = t(center), t(half), t(kind, torch.int64), t(yaw)

    # ---------------------------------------------------------------- analytic distance field
    def _object_distance(self, p, idx):
        """distance of world points p (...,3) to the object idx (...)"""
        c = self._center[idx]
        h = self._half[idx]
        yaw = self._yaw[idx]
        q = p - c
        cs, sn = torch.cos(yaw), torch.sin(yaw)
        lx = cs * q[..., 0] - sn * q[..., 2]
        lz = sn * q[..., 0] + cs * q[..., 2]
        l = torch.stack([lx, q[..., 1], lz], -1)
        d_sphere = torch.linalg.norm(l, dim=-1) - h[..., 0]
        a = torch.abs(l) - h
        d_box = torch.linalg.norm(torch.clamp(a, min=0.0), dim=-1) + torch.clamp(torch.amax(a, dim=-1), max=0.0)
        return torch.where(self._kind[idx] == 0, d_sphere, d_box)

    def distance(self, p):
        """conservative scene distance: ground plane, the object of the grid cell containing p, and the cell walls"""
        g, cell = self.grid, self.cell
        ix = torch.clamp(torch.floor(p[..., 0] / cell), 0, g - 1)
        iz = torch.clamp(torch.floor(p[..., 2] / cell), 0, g - 1)
        idx = (iz * g + ix).long()
        d_obj = self._object_distance(p, idx)
        wall = torch.minimum(torch.minimum(p[..., 0] - ix * cell, (ix + 1) * cell - p[..., 0]), torch.minimum(p[..., 2] - iz * cell, (iz + 1) * cell - p[..., 2]))
        inside = (p[..., 0] >= 0) & (p[..., 0] <= g * cell) & (p[..., 2] >= 0) & (p[..., 2] <= g * cell)
        d_wall = torch.where(inside, torch.clamp(wall, min=0.0) + 0.5, torch.full_like(wall, 1e9))
        d_ground = -p[..., 1]
        return torch.minimum(torch.minimum(d_ground, d_obj), d_wall), torch.minimum(d_ground, d_obj)

    def trace(self, origin, direction, t_max, steps=128):
        t = torch.zeros(origin.shape[:-1], device=self.device)
        hit = torch.zeros_like(t, dtype=torch.bool)
        for _ in range(steps):
            p = origin + direction * t[..., None]
            d_safe, d_true = self.distance(p)
            hit = hit | (d_true < 1e-3 * torch.clamp(t, min=1.0))
            t = torch.where(hit, t, t + torch.clamp(d_safe, min=1e-3))
            t = torch.clamp(t, max=t_max * 1.01)
        hit = hit & (t < t_max)
        return t, hit

    def normal(self, p):
        e = 2e-3
        out = []
        for a in range(3):
            dp = torch.zeros(3, device=self.device)
            dp[a] = e
            out.append(self.distance(p + dp)[1] - self.distance(p - dp)[1])
        n = torch.stack(out, -1)
        return n / torch.clamp(torch.linalg.norm(n, dim=-1, keepdim=True), min=1e-12)

    # ---------------------------------------------------------------- G-buffer
    def gbuffer(self, cam: Camera, w, h, cam_prev: Camera = None, rows=None):
        """rows=(r0, r1): only these rows of the w x h frame (arrays of r1 - r0 rows); the frame's coordinates stay global"""
        dev = self.device
        r0, r1 = rows if rows is not None else (0, h)
        ys, xs = torch.meshgrid(torch.arange(r0, r1, device=dev, dtype=torch.float32), torch.arange(w, device=dev, dtype=torch.float32), indexing="ij")
        full_h, h = h, r1 - r0
        ndc_x = (xs + 0.5) / w * 2 - 1
        ndc_y = (ys + 0.5) / full_h * 2 - 1
        tv = lambda a: torch.as_tensor(np.asarray(a, np.float32), device=dev)
        fwd, up, right, pos = tv(cam.forward), tv(cam.up), tv(cam.right), tv(cam.position)
        tan = cam.tan_fov_half()
        d = fwd[None, None, :] - tan * ndc_y[..., None] * up + tan * cam.aspect * ndc_x[..., None] * right
        dn = d / torch.linalg.norm(d, dim=-1, keepdim=True)
        cos_f = (dn * fwd).sum(-1)
        t, hit = self.trace(pos.expand(h, w, 3), dn, cam.far / torch.clamp(cos_f, min=1e-3))
        lin = t * cos_f
        n_, f_ = cam.near, cam.far
        depth = torch.where(hit, (n_ * f_ / torch.clamp(lin, min=n_) - n_) / (f_ - n_), torch.zeros_like(lin)).float()
        p = pos + dn * t[..., None]
        nrm = self.normal(p)
        nrm = torch.where(hit[..., None], nrm, torch.tensor([0.0, -1.0, 0.0], device=dev).expand(h, w, 3))
        out = {"depth": depth.cpu().numpy().astype(np.float32)}
        n8 = torch.cat([nrm * 0.5 + 0.5, torch.ones(h, w, 1, device=dev)], -1)
        out["normal"] = pixfmt.pack_unorm8(n8.cpu().numpy())
        # procedural material: checker albedo, roughness bands, a few metallic instances
        chk = ((torch.floor(p[..., 0] * 0.5) + torch.floor(p[..., 2] * 0.5) + torch.floor(p[..., 1] * 0.5)) % 2)
        base = 0.35 + 0.4 * chk
        tint = 0.5 + 0.5 * torch.sin(p * 0.37 + torch.tensor([0.0, 2.0, 4.0], device=dev))
        alb = torch.clamp(base[..., None] * (0.6 + 0.4 * tint), 0.02, 0.95)
        out["albedo"] = pixfmt.pack_unorm8(torch.cat([alb, torch.ones(h, w, 1, device=dev)], -1).cpu().numpy())
        rough = 0.25 + 0.6 * (0.5 + 0.5 * torch.sin(p[..., 0] * 0.21 + p[..., 2] * 0.13))
        metal = ((torch.floor(p[..., 0] / self.cell) + torch.floor(p[..., 2] / self.cell)) % 5 == 0).float() * (p[..., 1] < -0.05).float()
        spec = torch.stack([torch.ones_like(rough), rough, metal, torch.ones_like(rough)], -1)
        out["specular"] = pixfmt.pack_unorm8(spec.cpu().numpy())
        # motion = (ndcPrevious - ndcCurrent) * 0.5 on the un-jittered matrices (depthPrepass.frag:34-40)
        if cam_prev is None:
            out["motion"] = np.zeros((h, w, 2), np.int16)
        else:
            vp = torch.as_tensor(cam.view_projection().T.astype(np.float32), device=dev)      # row-major math matrix
            vpp = torch.as_tensor(cam_prev.view_projection().T.astype(np.float32), device=dev)
            ph = torch.cat([p, torch.ones(h, w, 1, device=dev)], -1)
            c0 = ph @ vp.T
            c1 = ph @ vpp.T
            m = (c1[..., :2] / c1[..., 3:4] - c0[..., :2] / c0[..., 3:4]) * 0.5
            m = torch.where(hit[..., None], m, torch.zeros_like(m))
            out["motion"] = pixfmt.pack_snorm16(m.cpu().numpy())
        out["hit"] = hit.cpu().numpy()
        return out

    # ---------------------------------------------------------------- SDF instances (SDFGI::updateSDFScene, Techniques/SDFGI.cpp:260-313)
    def sdf_instances(self, res, first_texture_index=0, distinct=True):
        """Returns (instance_bytes incl. 16-byte header, world_bb_bytes, volumes uint16 (n,res,res,res), local extends)."""
        inst = self.inst
        n = inst.center.shape[0]
        vols = np.zeros((n, res, res, res), np.uint16)
        inst_bytes = struct.pack("<4I", n, 0, 0, 0)
        bb_bytes = b""
        dev = self.device
        lin = (torch.arange(res, device=dev, dtype=torch.float32) + 0.5) / res - 0.5
        zz, yy, xx = torch.meshgrid(lin, lin, lin, indexing="ij")
        for i in range(n):
            h = inst.half[i]
            lmin, lmax = pad_sdf_bounding_box(-h, h)
            ext = (lmax - lmin).astype(np.float32)
            e = torch.as_tensor(ext, device=dev)
            l = torch.stack([xx * e[0], yy * e[1], zz * e[2]], -1)  # voxel centres in local space
            ht = torch.as_tensor(h, device=dev)
            if inst.kind[i] == 0:
                d = torch.linalg.norm(l, dim=-1) - ht[0]
            else:
                a = torch.abs(l) - ht
                d = torch.linalg.norm(torch.clamp(a, min=0.0), dim=-1) + torch.clamp(torch.amax(a, dim=-1), max=0.0)
            vols[i] = pixfmt.pack_half(d.cpu().numpy())
            # worldToLocal = inverse(model * translate(bbOffset)); bbOffset = 0 for these symmetric local boxes
            cs, sn = math.cos(float(inst.yaw[i])), math.sin(float(inst.yaw[i]))
            rot = np.array([[cs, 0, -sn], [0, 1, 0], [sn, 0, cs]], np.float64)  # world -> local
            m = np.eye(4)
            m[:3, :3] = rot
            m[:3, 3] = -rot @ inst.center[i].astype(np.float64)
            inst_bytes += struct.pack("<3fI3ff", ext[0], ext[1], ext[2], first_texture_index + i, *inst.albedo[i].tolist(), 0.0)
            inst_bytes += to_glm(m).tobytes()
            # world AABB of the rotated box, padded
            hw = np.abs(rot.T) @ h.astype(np.float64)
            wmin, wmax = pad_sdf_bounding_box(inst.center[i] - hw, inst.center[i] + hw)
            bb_bytes += struct.pack("<8f", *wmin.tolist(), 0.0, *wmax.tolist(), 0.0)
        return inst_bytes, bb_bytes, vols

    # ---------------------------------------------------------------- sun shadow cascades (lightMatrix.comp:57-138)
    def shadow_cascades(self, cam: Camera, sun_direction, depth_min_lin, depth_max_lin, res, cascade_count=3, extra_padding=5.0, min_far=30.0,
                        sample_radius=0.03):
        sun = np.asarray(sun_direction, np.float64)
        forward = -sun
        up = np.array([0.0, -1.0, 0.0]) if abs(forward[1]) < 0.9999 else np.array([0.0, 0.0, -1.0])
        right = np.cross(forward, up)
        up = np.cross(right, forward)
        V = np.eye(4)
        V[0, :3] = right / np.linalg.norm(right)
        V[1, :3] = up / np.linalg.norm(up)
        V[2, :3] = forward
        corr = np.array([[1, 0, 0, 0], [0, 1, 0, 0], [0, 0, -0.5, 0.5], [0, 0, 0, 1]], np.float64)
        splits = [depth_min_lin + (depth_max_lin - depth_min_lin) * (i + 1) / cascade_count for i in range(cascade_count - 1)]
        mats, scales, maps = [], [], []
        pos, fwd, cup, cright = [np.asarray(a, np.float64) for a in (cam.position, cam.forward, cam.up, cam.right)]
        for i in range(cascade_count):
            lo = depth_min_lin if i == 0 else splits[i - 1]
            hi = splits[i] if i < cascade_count - 1 else 0.0
            if i == cascade_count - 1:
                lo, hi = cam.near, max(depth_max_lin, min_far)
            pts = []
            for dist in (hi, lo):
                c = pos + fwd * dist
                hh = cam.tan_fov_half() * dist
                ww = hh * cam.aspect
                pts += [c + cup * hh + cright * ww, c + cup * hh - cright * ww, c - cup * hh + cright * ww, c - cup * hh - cright * ww]
            pv = (V[:3, :3] @ np.array(pts).T).T
            # lightMatrix.comp:88-89 starts the maximum at FLOAT_MIN (the smallest positive float), not at -FLOAT_MAX: a cascade that lies
            # entirely on the negative side of a light-space axis is fitted as if it reached 0
            mn, mx = pv.min(0), np.maximum(pv.max(0), 1.175494351e-38)
            if i == cascade_count - 1:
                mn, mx = mn - extra_padding, mx + extra_padding
            mn, mx = mn - sample_radius * 2, mx + sample_radius * 2
            scale = 2.0 / (mx - mn)
            offset = -0.5 * (mx + mn) * scale
            P = np.eye(4)
            P[0, 0], P[1, 1], P[2, 2] = scale
            P[:3, 3] = offset
            M = corr @ P @ V
            mats.append(M)
            scales.append(scale[:2])
            maps.append(self._render_shadow_map(M, res))
        while len(mats) < 4:
            mats.append(np.eye(4)); scales.append(np.ones(2)); maps.append(np.zeros((res, res), np.uint16))
        info = struct.pack("<4f", *(splits + [0.0] * (4 - len(splits))))
        for M in mats:
            info += to_glm(M).tobytes()
        for s in scales:
            info += struct.pack("<2f", float(s[0]), float(s[1]))
        assert len(info) == 304
        return info, maps

    def _render_shadow_map(self, M, res):
        dev = self.device
        Minv = np.linalg.inv(M)
        ys, xs = torch.meshgrid(torch.arange(res, device=dev, dtype=torch.float32), torch.arange(res, device=dev, dtype=torch.float32), indexing="ij")
        u = (xs + 0.5) / res * 2 - 1
        v = (ys + 0.5) / res * 2 - 1
        Mi = torch.as_tensor(Minv.astype(np.float32), device=dev)
        near = torch.stack([u, v, torch.ones_like(u), torch.ones_like(u)], -1) @ Mi.T   # z' = 1: closest to the light
        far = torch.stack([u, v, torch.zeros_like(u), torch.ones_like(u)], -1) @ Mi.T
        o = near[..., :3]
        seg = far[..., :3] - o
        length = torch.linalg.norm(seg, dim=-1)
        d = seg / length[..., None]
        t, hit = self.trace(o, d, length, steps=96)
        z = torch.where(hit, 1.0 - t / length, torch.zeros_like(t))
        return pixfmt.pack_unorm16(z.cpu().numpy())


# ---------------------------------------------------------------- LUT / noise stand-ins
def sky_lut(w=200, h=100):
    """analytic gradient, R11G11B10 (the reference's Hillaire sky LUT is an input, SURVEY component 14)"""
    v = (np.arange(h, dtype=np.float32)[:, None] + 0.5) / h
    u = (np.arange(w, dtype=np.float32)[None, :] + 0.5) / w
    horizon = np.exp(-((v - 0.5) ** 2) * 40.0)
    rgb = np.stack([0.2 + 0.8 * horizon + 0.1 * np.sin(u * 6.28), 0.35 + 0.6 * horizon + 0 * u, 0.7 * (1 - v) + 0.4 * horizon + 0 * u], -1)
    # darker below the horizon: a smooth ramp over eight rows (a real sky LUT has no texel-to-texel jumps; a hard step here would turn every
    # 1/256 flip of a bilinear filter weight - a legitimate fixed-point decision - into a several-percent difference between two implementations)
    t = np.clip((v - 0.50) / 0.08, 0.0, 1.0)
    rgb = rgb * (1.0 - 0.85 * (t * t * (3.0 - 2.0 * t)))[..., None]
    return pixfmt.pack_r11g11b10((rgb * 0.02).astype(np.float32))


def transmission_lut(n=128):
    v = (np.arange(n, dtype=np.float32)[:, None] + 0.5) / n
    u = (np.arange(n, dtype=np.float32)[None, :] + 0.5) / n
    rgb = np.stack([0.9 - 0.3 * v + 0 * u, 0.85 - 0.45 * v + 0 * u, 0.8 - 0.6 * v + 0 * u], -1)
    return pixfmt.pack_r11g11b10(np.clip(rgb, 0.05, 1.0).astype(np.float32))


def blue_noise_standins(count=4, size=32, seed_id=200):
    """seeded white-noise RG8 stand-ins for the four 32x32 void-and-cluster textures (RenderFrontend.cpp:1251-1269)"""
    r = np.random.default_rng(SEED_BASE + seed_id)
    return [r.integers(0, 256, (size, size, 2), dtype=np.uint8) for _ in range(count)]


def froxel_volume(w, h, depth_slices=64):
    """RGBA16F integration volume (ceil(W/8) x ceil(H/8) x 64): mild in-scattering, transmittance falling with depth"""
    fw, fh = (w + 7) // 8, (h + 7) // 8
    z = (np.arange(depth_slices, dtype=np.float32) + 0.5) / depth_slices
    trans = np.exp(-z * 0.35)
    ins = (1.0 - trans) * 0.002
    vol = np.zeros((depth_slices, fh, fw, 4), np.float32)
    vol[..., 0] = ins[:, None, None] * 0.9
    vol[..., 1] = ins[:, None, None] * 1.0
    vol[..., 2] = ins[:, None, None] * 1.2
    vol[..., 3] = trans[:, None, None]
    return pixfmt.pack_half(vol), (fw, fh, depth_slices)


def volumetric_settings_bytes(max_distance=30.0):
    """VolumetricLightingSettings, std140 (volumetricFroxelLighting.inc:6-16)"""
    return struct.pack("<13f", 0, 0, 0, 0.0, 1, 1, 1, max_distance, 1.0, 0.0, 0.0, 1.0, 0.0) + b"\0" * 12
