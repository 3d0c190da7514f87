"""numpy codecs for the packed pixel formats (host-side input generation and result decoding).

Same rules as the kernels (csrc/device/image.h): round-to-nearest-even, 11/10-bit floats clamp negatives to 0
and finite overflow to the maximum finite value.
"""
import numpy as np


def _encode_ufloat(v, M):
    v = np.ascontiguousarray(v, np.float32)
    u = v.view(np.uint32).astype(np.uint64)
    exp_max = 31 << M
    max_finite = (30 << M) | ((1 << M) - 1)
    shift = 23 - M
    is_nan = (u & 0x7fffffff) > 0x7f800000
    neg = (u >> 31) != 0
    inf = u == 0x7f800000
    small = v < np.float32(6.103515625e-05)
    magic = np.float32(1 << (9 - M))
    with np.errstate(invalid="ignore", over="ignore"):
        sub = (np.where(small & ~neg & ~is_nan, v, np.float32(0)) + magic).astype(np.float32).view(np.uint32).astype(np.uint64) - np.uint64(np.float32(magic).view(np.uint32))
    t = u + ((1 << (shift - 1)) - 1) + ((u >> shift) & 1)
    norm = (t >> shift).astype(np.int64) - (112 << M)
    r = np.where(small, sub.astype(np.int64), norm)
    r = np.minimum(r, max_finite)
    r = np.where(inf, exp_max, r)
    r = np.where(neg, 0, r)
    r = np.where(is_nan, exp_max | (1 << (M - 1)), r)
    return r.astype(np.uint32)


def _decode_ufloat(c, M):
    c = c.astype(np.uint32)
    h = (c << (10 - M)).astype(np.uint16)
    return h.view(np.float16).astype(np.float32)


def pack_r11g11b10(rgb):
    """rgb: (..., 3) float32 -> (...) uint32"""
    rgb = np.asarray(rgb, np.float32)
    return (_encode_ufloat(rgb[..., 0], 6) | (_encode_ufloat(rgb[..., 1], 6) << 11) | (_encode_ufloat(rgb[..., 2], 5) << 22)).astype(np.uint32)


def unpack_r11g11b10(p):
    p = np.asarray(p, np.uint32)
    return np.stack([_decode_ufloat(p & 0x7ff, 6), _decode_ufloat((p >> 11) & 0x7ff, 6), _decode_ufloat(p >> 22, 5)], axis=-1)


def pack_half(x):
    with np.errstate(over="ignore"):
        return np.asarray(x, np.float32).astype(np.float16).view(np.uint16)


def unpack_half(u):
    return np.asarray(u, np.uint16).view(np.float16).astype(np.float32)


def pack_unorm8(x):
    x = np.nan_to_num(np.asarray(x, np.float32), nan=0.0)
    return np.rint(np.clip(x, 0.0, 1.0) * np.float32(255.0)).astype(np.uint8)


def unpack_unorm8(u):
    return np.asarray(u, np.uint8).astype(np.float32) / np.float32(255.0)


def pack_snorm16(x):
    x = np.nan_to_num(np.asarray(x, np.float32), nan=0.0)
    return np.rint(np.clip(x, -1.0, 1.0) * np.float32(32767.0)).astype(np.int16)


def unpack_snorm16(i):
    return np.maximum(np.asarray(i, np.int16).astype(np.float32) / np.float32(32767.0), np.float32(-1.0))


def pack_unorm16(x):
    return np.rint(np.clip(np.asarray(x, np.float32), 0.0, 1.0).astype(np.float64) * 65535.0).astype(np.uint16)
