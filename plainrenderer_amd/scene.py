"""Host-side camera / GlobalShaderInfo helpers (numpy), following the reference's conventions:

* GlobalShaderInfo layout: Plain/src/Runtime/Rendering/ResourceDescriptions.h:174-203 == resources/shaders/global.inc:4-33 (340 B)
* view / projection: Plain/src/Runtime/Rendering/Camera.cpp:4-27 (reverse-Z, Y flip)
* TAA jitter and resolve weights: Techniques/TAA.cpp:168-202, Common/Utilities/MathUtils.cpp:25-60
* frustum: ViewFrustum.cpp:4-52, packed as in Techniques/SDFGI.cpp:543-566
"""
import math
import struct
from dataclasses import dataclass, field

import numpy as np

f32 = np.float32


@dataclass
class GlobalShaderInfo:
    viewProjection: np.ndarray = field(default_factory=lambda: np.zeros((4, 4), f32))          # [col][row]
    viewProjectionPrevious: np.ndarray = field(default_factory=lambda: np.zeros((4, 4), f32))
    sunDirection: tuple = (0.0, -1.0, 0.0, 0.0)
    cameraPos: tuple = (0.0, 0.0, 0.0, 0.0)
    cameraPosPrevious: tuple = (0.0, 0.0, 0.0, 0.0)
    cameraRight: tuple = (1.0, 0.0, 0.0, 0.0)
    cameraUp: tuple = (0.0, -1.0, 0.0, 0.0)
    cameraForward: tuple = (0.0, 0.0, -1.0, 0.0)
    cameraForwardPrevious: tuple = (0.0, 0.0, -1.0, 0.0)
    noiseTextureIndices: tuple = (0, 0, 0, 0)
    currentFrameCameraJitter: tuple = (0.0, 0.0)
    previousFrameCameraJitter: tuple = (0.0, 0.0)
    screenResolution: tuple = (0, 0)
    cameraTanFovHalf: float = 1.0
    cameraAspectRatio: float = 1.0
    nearPlane: float = 0.1
    farPlane: float = 100.0
    sunIlluminanceLux: float = 128000.0
    exposureOffset: float = 1.0
    exposureAdaptionSpeedEvPerSec: float = 2.0
    deltaTime: float = 0.016
    time: float = 0.0
    mipBias: float = 0.0
    cameraCut: bool = False
    frameIndex: int = 0

    def pack(self) -> bytes:
        """GPU-side (std140) image of the block: the bool is a 4-byte value at offset 320."""
        vp = np.asarray(self.viewProjection, f32).reshape(16)
        vpp = np.asarray(self.viewProjectionPrevious, f32).reshape(16)
        b = vp.tobytes() + vpp.tobytes()
        for v in (self.sunDirection, self.cameraPos, self.cameraPosPrevious, self.cameraRight, self.cameraUp, self.cameraForward,
                  self.cameraForwardPrevious):
            b += struct.pack("<4f", *[float(x) for x in v])
        b += struct.pack("<4i", *[int(x) for x in self.noiseTextureIndices])
        b += struct.pack("<2f", *[float(x) for x in self.currentFrameCameraJitter])
        b += struct.pack("<2f", *[float(x) for x in self.previousFrameCameraJitter])
        b += struct.pack("<2i", *[int(x) for x in self.screenResolution])
        b += struct.pack("<10f", self.cameraTanFovHalf, self.cameraAspectRatio, self.nearPlane, self.farPlane, self.sunIlluminanceLux,
                         self.exposureOffset, self.exposureAdaptionSpeedEvPerSec, self.deltaTime, self.time, self.mipBias)
        fi = int(self.frameIndex) & 0xFFFFFFFF
        b += struct.pack("<5I", 1 if self.cameraCut else 0, fi, fi % 2, fi % 3, fi % 4)
        assert len(b) == 340
        return b


def normalize(v):
    v = np.asarray(v, f32)
    return (v / f32(np.sqrt(np.dot(v, v)))).astype(f32)


def view_matrix(position, right, up, forward):
    """Camera.cpp:4-12. Returns a [col][row] float32 array (glm layout)."""
    # viewMatrix[0]=right, [1]=up, [2]=-forward as columns, then transposed -> they become rows
    rows = np.eye(4)
    rows[0, :3] = right
    rows[1, :3] = up
    rows[2, :3] = -np.asarray(forward, np.float64)
    t = np.eye(4)
    t[:3, 3] = -np.asarray(position, np.float64)
    mat = rows @ t  # row-major math matrix
    return mat


def projection_matrix(fov_deg, aspect, near, far):
    """Camera.cpp:14-27: glm::perspective (RH, -1..1 depth) then the Vulkan / reverse-Z correction."""
    f = 1.0 / math.tan(math.radians(fov_deg) * 0.5)
    p = np.zeros((4, 4))
    p[0, 0] = f / aspect
    p[1, 1] = f
    p[2, 2] = -(far + near) / (far - near)
    p[2, 3] = -(2.0 * far * near) / (far - near)
    p[3, 2] = -1.0
    corr = np.array([[1, 0, 0, 0], [0, -1, 0, 0], [0, 0, -0.5, 0.5], [0, 0, 0, 1]], np.float64)
    return corr @ p


def to_glm(mat_rowmajor):
    """math (row-major) matrix -> [col][row] float32 as stored by glm / GLSL."""
    return np.ascontiguousarray(np.asarray(mat_rowmajor, np.float64).T.astype(f32))


def radical_inverse(index, base):
    inv = 1.0 / base
    r, f = 0.0, inv
    while index > 0:
        r += (index % base) * f
        index //= base
        f *= inv
    return r


def hammersley2d(index):
    """MathUtils.cpp:25-28 (Halton 2,3)."""
    return np.array([radical_inverse(index, 2), radical_inverse(index, 3)], f32)


def taa_jitter_pixels(frame_index_mod8):
    """TAA.cpp:168-170"""
    return (f32(2.0) * hammersley2d(frame_index_mod8) - f32(1.0)).astype(f32)


def taa_resolve_weights(jitter_px):
    """TAA.cpp:181-202 (float32, row-major over y then x)."""
    w = np.zeros(9, f32)
    total = f32(0)
    i = 0
    for y in (-1, 0, 1):
        for x in (-1, 0, 1):
            dx = f32(jitter_px[0]) - f32(x)
            dy = f32(jitter_px[1]) - f32(y)
            d = f32(np.sqrt(f32(dx * dx + dy * dy)))
            w[i] = f32(np.exp(f32(f32(-2.29) * d * d)))
            total = f32(total + w[i])
            i += 1
    return (w / total).astype(f32)


@dataclass
class Camera:
    position: np.ndarray
    forward: np.ndarray
    up: np.ndarray
    right: np.ndarray
    fov: float = 35.0
    aspect: float = 16.0 / 9.0
    near: float = 0.1
    far: float = 300.0

    @staticmethod
    def look(position, forward, world_up=(0.0, -1.0, 0.0), **kw):
        fwd = normalize(forward)
        right = normalize(np.cross(np.asarray(world_up, f32), fwd))
        up = normalize(np.cross(fwd, right))
        return Camera(np.asarray(position, f32), fwd, up, right, **kw)

    def view_projection(self, jitter_uv=(0.0, 0.0)):
        v = view_matrix(self.position, self.right, self.up, self.forward)
        p = projection_matrix(self.fov, self.aspect, self.near, self.far)
        # TAA.cpp:172-179: jitteredProjection[2][0..1] = offset (glm column 2, rows 0/1)
        p = p.copy()
        p[0, 2] = jitter_uv[0]
        p[1, 2] = jitter_uv[1]
        return to_glm(p @ v)

    def tan_fov_half(self):
        return math.tan(math.radians(self.fov) * 0.5)

    def frustum_points_normals(self):
        """ViewFrustum.cpp:4-52 + SDFGI.cpp:543-566 -> (6x4 points, 6x4 normals), order top, bot, near, far, left, right."""
        pos, fwd, up, right = [np.asarray(a, np.float64) for a in (self.position, self.forward, self.up, self.right)]
        nc, fc = pos + fwd * self.near, pos + fwd * self.far
        t = self.tan_fov_half()
        hn, hf = t * self.near, t * self.far
        wn, wf = hn * self.aspect, hf * self.aspect
        P = {
            "ruf": fc + up * hf + right * wf, "luf": fc + up * hf - right * wf, "rlf": fc - up * hf + right * wf, "llf": fc - up * hf - right * wf,
            "run": nc + up * hn + right * wn, "lun": nc + up * hn - right * wn, "rln": nc - up * hn + right * wn, "lln": nc - up * hn - right * wn,
        }

        def nrm(a, b):
            c = np.cross(a, b)
            return c / np.linalg.norm(c)

        N = {
            "top": nrm(P["ruf"] - P["run"], P["run"] - P["lun"]), "bot": nrm(P["rln"] - P["lln"], P["rlf"] - P["rln"]),
            "right": nrm(P["run"] - P["rln"], P["rlf"] - P["rln"]), "left": nrm(P["llf"] - P["lln"], P["lun"] - P["lln"]),
            "near": nrm(P["run"] - P["rln"], P["rln"] - P["lln"]), "far": nrm(P["rlf"] - P["llf"], P["ruf"] - P["rlf"]),
        }
        pts = np.zeros((6, 4), f32)
        nrms = np.zeros((6, 4), f32)
        for i, (pk, nk) in enumerate([("luf", "top"), ("llf", "bot"), ("lln", "near"), ("llf", "far"), ("llf", "left"), ("rlf", "right")]):
            pts[i, :3] = P[pk]
            nrms[i, :3] = N[nk]
        return pts, nrms

    def fill_global(self, g: GlobalShaderInfo, width, height, jitter_uv=(0.0, 0.0)):
        g.viewProjection = self.view_projection(jitter_uv)
        g.cameraPos = (*[float(x) for x in self.position], 1.0)
        g.cameraRight = (*[float(x) for x in self.right], 0.0)
        g.cameraUp = (*[float(x) for x in self.up], 0.0)
        g.cameraForward = (*[float(x) for x in self.forward], 0.0)
        g.cameraTanFovHalf = float(f32(self.tan_fov_half()))
        g.cameraAspectRatio = float(f32(self.aspect))
        g.nearPlane, g.farPlane = self.near, self.far
        g.screenResolution = (width, height)
        g.currentFrameCameraJitter = tuple(float(x) for x in jitter_uv)
        return g
