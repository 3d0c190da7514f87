"""ctypes mirror of include/plr_sdf_bake.h: the reference's computeSceneSDFTextures (AssetPipeline/SceneSDF.h) on the GPU."""
import ctypes as C

import numpy as np

from .backend import PlrError, _ImageDesc, _load

SDF_BAKE_SYMBOLS = ["plr_sdf_texture_description", "plr_sdf_padded_bounds", "plr_compute_sdf", "plr_compute_scene_sdf_textures",
                    "plr_sdf_last_kernel_ms"]


class _MeshData(C.Structure):
    _fields_ = [("positions", C.c_void_p), ("vertex_count", C.c_uint32), ("indices", C.c_void_p), ("index_count", C.c_uint32)]


class _Aabb(C.Structure):
    _fields_ = [("min", C.c_float * 3), ("max", C.c_float * 3)]


def _check(lib, rc):
    if rc != 0:
        raise PlrError("plr error %d: %s" % (rc, lib.plr_last_error().decode()))


def _aabb(bb_min, bb_max):
    return _Aabb((C.c_float * 3)(*[float(v) for v in bb_min]), (C.c_float * 3)(*[float(v) for v in bb_max]))


def sdf_texture_description(bb_min, bb_max):
    """-> (width, height, depth) chosen by the reference's resolution rule (SceneSDF.cpp:120-131)."""
    lib = _load()
    d = _ImageDesc()
    bb = _aabb(bb_min, bb_max)
    _check(lib, lib.plr_sdf_texture_description(C.byref(bb), C.byref(d)))
    return int(d.width), int(d.height), int(d.depth)


def sdf_padded_bounds(bb_min, bb_max):
    lib = _load()
    bb = _aabb(bb_min, bb_max)
    out = _Aabb()
    _check(lib, lib.plr_sdf_padded_bounds(C.byref(bb), C.byref(out)))
    return np.array(list(out.min), np.float32), np.array(list(out.max), np.float32)


def compute_sdf(positions, indices, bb_min, bb_max, res, device=0):
    """computeSDF for one mesh on the GPU -> uint16 array [resZ, resY, resX] of half-float bits."""
    lib = _load()
    pos = np.ascontiguousarray(positions, np.float32).reshape(-1, 3)
    idx = np.ascontiguousarray(indices, np.uint32).ravel()
    mesh = _MeshData(pos.ctypes.data, pos.shape[0], idx.ctypes.data, idx.size)
    bb = _aabb(bb_min, bb_max)
    out = np.zeros(int(res[0]) * int(res[1]) * int(res[2]), np.uint16)
    _check(lib, lib.plr_compute_sdf(C.c_int(device), C.byref(mesh), C.byref(bb), C.c_uint32(res[0]), C.c_uint32(res[1]), C.c_uint32(res[2]),
                                    C.c_void_p(out.ctypes.data), C.c_size_t(out.nbytes)))
    return out.reshape(res[2], res[1], res[0])


def compute_scene_sdf_textures(meshes, bounds, device=0):
    """meshes: [(positions, indices)], bounds: [(min, max)] -> ([(w, h, d)], [uint16 volume], seconds)."""
    lib = _load()
    n = len(meshes)
    keep = []
    marr = (_MeshData * n)()
    barr = (_Aabb * n)()
    for i, ((p, ix), (mn, mx)) in enumerate(zip(meshes, bounds)):
        pos = np.ascontiguousarray(p, np.float32).reshape(-1, 3)
        idx = np.ascontiguousarray(ix, np.uint32).ravel()
        keep += [pos, idx]
        marr[i] = _MeshData(pos.ctypes.data, pos.shape[0], idx.ctypes.data, idx.size)
        barr[i] = _aabb(mn, mx)
    descs = (_ImageDesc * n)()
    outs = [np.zeros(64 * 64 * 64, np.uint16) for _ in range(n)]
    optr = (C.c_void_p * n)(*[o.ctypes.data for o in outs])
    osz = (C.c_size_t * n)(*[o.nbytes for o in outs])
    secs = C.c_double(0.0)
    _check(lib, lib.plr_compute_scene_sdf_textures(C.c_int(device), marr, barr, C.c_uint32(n), descs, optr, osz, C.byref(secs)))
    res = [(int(d.width), int(d.height), int(d.depth)) for d in descs]
    vols = [o[: r[0] * r[1] * r[2]].reshape(r[2], r[1], r[0]) for o, r in zip(outs, res)]
    return res, vols, secs.value


def last_kernel_ms():
    lib = _load()
    ms = C.c_float(0.0)
    _check(lib, lib.plr_sdf_last_kernel_ms(C.byref(ms)))
    return ms.value
