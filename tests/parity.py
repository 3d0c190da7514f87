"""The tolerance statement the PLR_MATH_FAST kernel set is held to, as code.

A pass output is compared with the oracle channel by channel:
  * R11G11B10 images: the packed codes may differ by at most ONE code step per channel (one storage quantum: 2^-6 relative for R and G,
    2^-5 for B) - `r11g11b10_code_diff`.
  * half-float images (GI): |got - ref| <= max(2^-7 * |ref|, ABS_FLOOR) on the decoded values, ABS_FLOOR stated per image
    as a fraction of the image's largest magnitude.
  * RGBA8 swapchain: +-1 LSB.
Pixels are split by their DECISION SIGNATURE (one word per pixel emitted by both the oracle and the HIP kernel, oracle/oracle.h):
where the words agree, every channel must meet the bound - no outlier allowance; where they differ, a float rounding flipped a discrete
decision (ray hit, nearest texel, PCF tap, edge test) and the pixel is counted against a hard cap instead.
"""
import numpy as np


def r11g11b10_code_diff(got_u32, ref_u32):
    """-> int array [..., 3]: |code difference| per channel (codes of positive floats are monotonic in the value)"""
    got = np.asarray(got_u32, np.uint32).reshape(-1)
    ref = np.asarray(ref_u32, np.uint32).reshape(-1)
    out = np.empty((got.size, 3), np.int32)
    for c, (sh, m) in enumerate(((0, 0x7FF), (11, 0x7FF), (22, 0x3FF))):
        out[:, c] = np.abs(((got >> sh) & m).astype(np.int32) - ((ref >> sh) & m).astype(np.int32))
    return out


def dilate3x3(mask2d):
    m = np.asarray(mask2d, bool)
    p = np.pad(m, 1)
    out = np.zeros_like(m)
    h, w = m.shape
    for dy in range(3):
        for dx in range(3):
            out |= p[dy:dy + h, dx:dx + w]
    return out


def half_violations(got, ref, floor_frac, rel=2.0 ** -7):
    """got/ref: decoded float arrays [pixels, channels] -> bool [pixels]: some channel outside max(rel * |ref|, floor_frac * max|ref|)"""
    got = np.asarray(got, np.float64)
    ref = np.asarray(ref, np.float64)
    tol = np.maximum(rel * np.abs(ref), floor_frac * np.abs(ref).max())
    bad = (np.abs(got - ref) > tol) | ~np.isfinite(got)
    return bad.any(axis=-1)
