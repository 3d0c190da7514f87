"""ORACLE binding (test infrastructure): ctypes access to oracle/_build/liboracle.so.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module. It lives under
oracle/ on purpose: the product package (plainrenderer_amd) never imports it.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "_build", "liboracle.so")


def build(verbose=False):
    r = subprocess.run(["make", "-C", _HERE, "-j4"], capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("oracle build failed:\n" + r.stdout + r.stderr)
    if verbose:
        print(r.stdout)
    return LIB_PATH


class OrcImage(C.Structure):
    _fields_ = [("data", C.c_void_p), ("w", C.c_int32), ("h", C.c_int32), ("d", C.c_int32), ("format", C.c_int32)]


class OrcGlobal(C.Structure):
    _fields_ = [
        ("viewProjection", C.c_float * 16), ("viewProjectionPrevious", C.c_float * 16), ("sunDirection", C.c_float * 4),
        ("cameraPosition", C.c_float * 4), ("cameraPositionPrevious", C.c_float * 4), ("cameraRight", C.c_float * 4),
        ("cameraUp", C.c_float * 4), ("cameraForward", C.c_float * 4), ("cameraForwardPrevious", C.c_float * 4),
        ("noiseTextureIndices", C.c_int32 * 4), ("currentFrameCameraJitter", C.c_float * 2),
        ("previousFrameCameraJitter", C.c_float * 2), ("screenResolution", C.c_int32 * 2), ("cameraTanFovHalf", C.c_float),
        ("cameraAspectRatio", C.c_float), ("nearPlane", C.c_float), ("farPlane", C.c_float), ("sunStrength", C.c_float),
        ("exposureOffset", C.c_float), ("exposureAdaptionSpeedEvPerSec", C.c_float), ("deltaTime", C.c_float), ("time", C.c_float),
        ("mipBias", C.c_float), ("cameraCut", C.c_uint32), ("frameIndex", C.c_uint32), ("frameIndexMod2", C.c_uint32),
        ("frameIndexMod3", C.c_uint32), ("frameIndexMod4", C.c_uint32),
    ]


assert C.sizeof(OrcGlobal) == 340


class OrcLightBuffer(C.Structure):
    _fields_ = [("sunColor", C.c_float * 3), ("previousFrameExposure", C.c_float), ("sunStrengthExposed", C.c_float)]


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            build()
        _lib = C.CDLL(LIB_PATH)
    return _lib


class Img:
    """numpy-backed image: `arr` holds the packed texels (any dtype, contiguous), x fastest then y then z."""

    def __init__(self, arr, w, h, fmt, d=1):
        self.arr = np.ascontiguousarray(arr)
        self.w, self.h, self.d, self.fmt = int(w), int(h), int(d), int(fmt)
        self.c = OrcImage(self.arr.ctypes.data_as(C.c_void_p), self.w, self.h, self.d, self.fmt)

    def ref(self):
        return C.byref(self.c)


def new_image(w, h, fmt, bytes_per_texel, d=1):
    return Img(np.zeros(w * h * d * bytes_per_texel, np.uint8), w, h, fmt, d)


def global_from_bytes(b):
    g = OrcGlobal()
    C.memmove(C.byref(g), bytes(b), 340)
    return g


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def set_threads(n):
    lib().orc_set_threads(C.c_int32(n))


def math_eval(fn, a, b=None):
    a = np.ascontiguousarray(a, np.float32)
    out = np.empty_like(a)
    bp = None
    if b is not None:
        b = np.ascontiguousarray(b, np.float32)
        bp = _p(b)
    lib().orc_math_eval(C.c_int(fn), _p(a), bp, _p(out), C.c_int64(a.size))
    return out


def codec_eval(fn, data, n, out_dtype, out_count):
    data = np.ascontiguousarray(data)
    out = np.empty(out_count, out_dtype)
    lib().orc_codec_eval(C.c_int(fn), _p(data), _p(out), C.c_int64(n))
    return out


# ---- config 1: SDF bake (AssetPipeline/SceneSDF.cpp)
def sdf_resolution(bb_min, bb_max):
    mn = np.ascontiguousarray(bb_min, np.float32); mx = np.ascontiguousarray(bb_max, np.float32)
    res = np.zeros(3, np.int32)
    lib().orc_sdf_resolution(_p(mn), _p(mx), _p(res))
    return tuple(int(r) for r in res)


def sdf_padded_box(bb_min, bb_max):
    mn = np.ascontiguousarray(bb_min, np.float32); mx = np.ascontiguousarray(bb_max, np.float32)
    omn = np.zeros(3, np.float32); omx = np.zeros(3, np.float32)
    lib().orc_sdf_padded_box(_p(mn), _p(mx), _p(omn), _p(omx))
    return omn, omx


def pack_half_glm(values):
    f = lib().orc_pack_half_glm
    f.restype = C.c_uint16
    return np.array([f(C.c_float(float(v))) for v in np.asarray(values, np.float32).ravel()], np.uint16)


def sdf_bake(positions, indices, bb_min, bb_max, res):
    """-> uint16 array [resZ, resY, resX] of half-float bits."""
    pos = np.ascontiguousarray(positions, np.float32).reshape(-1, 3)
    idx = np.ascontiguousarray(indices, np.uint32).ravel()
    mn = np.ascontiguousarray(bb_min, np.float32); mx = np.ascontiguousarray(bb_max, np.float32)
    out = np.zeros(res[0] * res[1] * res[2], np.uint16)
    f = lib().orc_sdf_bake
    f.restype = C.c_int32
    rc = f(_p(pos), C.c_int64(pos.shape[0]), _p(idx), C.c_int64(idx.size), _p(mn), _p(mx), C.c_int32(res[0]), C.c_int32(res[1]), C.c_int32(res[2]), _p(out))
    if rc != 0:
        raise ValueError("orc_sdf_bake failed: %d" % rc)
    return out.reshape(res[2], res[1], res[0])


class decision_signature:
    """with decision_signature(words) as s: <one oracle pass>; s.words -> uint32 array (oracle.h: orc_set_decision_signature)"""

    def __init__(self, words):
        self.words = np.full(int(words), 0xFFFFFFFF, np.uint32)

    def __enter__(self):
        lib().orc_set_decision_signature(_p(self.words), C.c_int64(self.words.size))
        return self

    def __exit__(self, *exc):
        lib().orc_set_decision_signature(None, C.c_int64(0))
        return False


# ---- known-answer probes (tests/test_kat.py)
def kat_taa(fn, data, in_per, out_per):
    a = np.ascontiguousarray(data, np.float32).reshape(-1, in_per)
    out = np.zeros((a.shape[0], out_per), np.float32)
    lib().orc_kat_taa(C.c_int(fn), _p(a), _p(out), C.c_int64(a.shape[0]))
    return out


def kat_history_sample(image, tech, iuvs, nbr=None):
    a = np.ascontiguousarray(iuvs, np.float32).reshape(-1, 2)
    out = np.zeros((a.shape[0], 3), np.float32)
    nb = None if nbr is None else np.ascontiguousarray(nbr, np.float32).reshape(27)
    lib().orc_kat_history_sample(image.ref(), C.c_int32(tech), _p(a), None if nb is None else _p(nb), _p(out), C.c_int64(a.shape[0]))
    return out


def kat_trace_ray(instance_bytes96, volume, rays):
    a = np.ascontiguousarray(rays, np.float32).reshape(-1, 6)
    out = np.zeros((a.shape[0], 9), np.float32)
    inst = C.create_string_buffer(bytes(instance_bytes96), 96)
    lib().orc_kat_trace_ray(inst, volume.ref(), _p(a), _p(out), C.c_int64(a.shape[0]))
    return out


def sampler_eval(image, filter, address, coords):
    a = np.ascontiguousarray(coords, np.float32)
    out = np.zeros((a.shape[0], 4), np.float32)
    lib().orc_sampler_eval(image.ref(), C.c_int32(filter), C.c_int32(address), _p(a), _p(out), C.c_int64(a.shape[0]))
    return out


def kat_sampling(fn, data, in_per, out_per):
    a = np.ascontiguousarray(data, np.float32).reshape(-1, in_per)
    out = np.zeros((a.shape[0], out_per), np.float32)
    lib().orc_kat_sampling(C.c_int(fn), _p(a), _p(out), C.c_int64(a.shape[0]))
    return out


def kat_pcf_taps():
    out = np.zeros((256, 12, 2), np.float32)
    lib().orc_kat_pcf_taps(_p(out))
    return out


def kat_sky_lut(image, dirs):
    a = np.ascontiguousarray(dirs, np.float32).reshape(-1, 3)
    out = np.zeros((a.shape[0], 3), np.float32)
    lib().orc_kat_sky_lut(image.ref(), _p(a), _p(out), C.c_int64(a.shape[0]))
    return out
