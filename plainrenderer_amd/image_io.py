"""ctypes mirror of include/plr_image_io.h: the reference's DDS reader / writer (Common/ImageIO.cpp loadDDSFile / writeDDSFile)."""
import ctypes as C

import numpy as np

from .backend import ImageDescription, ImageFormat, ImageType, MipCount, PlrError, _ImageDesc, _load

IMAGE_IO_SYMBOLS = ["plr_write_dds_file", "plr_load_dds_file", "plr_encode_dds", "plr_decode_dds"]


def _check(lib, rc):
    if rc != 0:
        raise PlrError("plr error %d: %s" % (rc, lib.plr_last_error().decode()))


def _cdesc(d: ImageDescription):
    return _ImageDesc(d.width, d.height, d.depth, int(d.type), int(d.format), int(d.usageFlags), int(d.mipCount), d.manualMipCount, int(d.autoCreateMips))


def _pydesc(c):
    return ImageDescription(width=c.width, height=c.height, depth=c.depth, type=ImageType(c.type), format=ImageFormat(c.format), usageFlags=c.usage_flags,
                            mipCount=MipCount(c.mip_count), manualMipCount=c.manual_mip_count, autoCreateMips=bool(c.auto_create_mips))


def encode_dds(desc: ImageDescription, data) -> bytes:
    lib = _load()
    payload = np.ascontiguousarray(data).view(np.uint8).reshape(-1)
    cd = _cdesc(desc)
    n = C.c_size_t()
    _check(lib, lib.plr_encode_dds(C.byref(cd), payload.ctypes.data_as(C.c_void_p), C.c_size_t(payload.size), None, C.c_size_t(0), C.byref(n)))
    out = np.empty(n.value, np.uint8)
    _check(lib, lib.plr_encode_dds(C.byref(cd), payload.ctypes.data_as(C.c_void_p), C.c_size_t(payload.size), out.ctypes.data_as(C.c_void_p), C.c_size_t(out.size), C.byref(n)))
    return out.tobytes()


def decode_dds(file_bytes: bytes):
    """-> (ImageDescription, payload bytes)"""
    lib = _load()
    buf = np.frombuffer(file_bytes, np.uint8)
    cd = _ImageDesc()
    off, size = C.c_size_t(), C.c_size_t()
    _check(lib, lib.plr_decode_dds(buf.ctypes.data_as(C.c_void_p), C.c_size_t(buf.size), C.byref(cd), C.byref(off), C.byref(size)))
    return _pydesc(cd), file_bytes[off.value:off.value + size.value]


def write_dds_file(path, desc: ImageDescription, data):
    lib = _load()
    payload = np.ascontiguousarray(data).view(np.uint8).reshape(-1)
    cd = _cdesc(desc)
    _check(lib, lib.plr_write_dds_file(str(path).encode(), C.byref(cd), payload.ctypes.data_as(C.c_void_p), C.c_size_t(payload.size)))


def load_dds_file(path):
    """-> (ImageDescription, uint8 array)"""
    lib = _load()
    cd = _ImageDesc()
    size = C.c_size_t()
    _check(lib, lib.plr_load_dds_file(str(path).encode(), C.byref(cd), None, C.c_size_t(0), C.byref(size)))
    out = np.empty(size.value, np.uint8)
    _check(lib, lib.plr_load_dds_file(str(path).encode(), C.byref(cd), out.ctypes.data_as(C.c_void_p), C.c_size_t(out.size), C.byref(size)))
    return _pydesc(cd), out
