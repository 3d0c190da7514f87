"""Pass drivers used by the parity tests, __graft_entry__.smoke() and bench.py's cpu_baseline leg.

For every reference pass there is a `gpu_*` function that records the pass through the reference-shaped
RenderBackend API (bindings / spec constants / dispatch counts exactly as the reference frontend assigns them,
file:line cited per function) and an `orc_*` function that evaluates the oracle on the same packed inputs.
"""
import ctypes as C
import math
import struct

import numpy as np

import pyoracle as orc
from util import (ComputePassExecution, ImageFormat, ImageResource, MipCount, RenderPassResources, StorageBufferResource,
                  UniformBufferResource, div_up, image_desc_2d)
from plainrenderer_amd.backend import spec_bool, spec_float, spec_int, spec_uint

F = ImageFormat
N_BINS = 128          # RenderFrontend.cpp:46
MIN_LUM = 0.001       # RenderFrontend.cpp:1066
MAX_LUM = 200000.0    # RenderFrontend.cpp:1067


class GlobalBinding:
    """set 0: the `global` uniform buffer (RenderFrontend.cpp:1158-1184)."""

    def __init__(self, be):
        self.be = be
        self.ubo = be.createUniformBuffer(340)
        be.setGlobalDescriptorSetResources(RenderPassResources(uniformBuffers=[UniformBufferResource(self.ubo, 0)]))

    def set(self, packed340):
        self.be.setUniformBufferData(self.ubo, packed340)


_global_cache = {}


def global_binding(be):
    if id(be) not in _global_cache:
        _global_cache[id(be)] = GlobalBinding(be)
    return _global_cache[id(be)]


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


# ------------------------------------------------------------------------------------------- exposure
def gpu_histogram(be, color_packed, w, h, light_bytes):
    """computeColorBufferHistogram, RenderFrontend.cpp:707-754; pass creation :1618-1688"""
    tiles_x, tiles_y = math.ceil(w / 32.0), math.ceil(h / 32.0)
    n_tiles = tiles_x * tiles_y
    color = be.createImage(image_desc_2d(w, h, F.R11G11B10_uFloat), color_packed)
    light = be.createStorageBuffer(20, light_bytes)
    per_tile = be.createStorageBuffer(n_tiles * N_BINS * 4)
    hist = be.createStorageBuffer(N_BINS * 4, struct.pack("<%dI" % N_BINS, *([123456] * N_BINS)))  # garbage: reset must clear it
    p_tile = be.createComputePass("histogramPerTile.comp", [spec_uint(0, N_BINS), spec_float(1, MIN_LUM), spec_float(2, MAX_LUM), spec_int(3, n_tiles)],
                                  "Histogram per tile")
    p_reset = be.createComputePass("histogramReset.comp", [spec_uint(0, N_BINS)], "Histogram reset")
    p_comb = be.createComputePass("histogramCombineTiles.comp", [spec_uint(0, N_BINS), spec_int(1, n_tiles)], "Histogram combine tiles")
    be.newFrame()
    be.setComputePassExecution(ComputePassExecution(p_tile, RenderPassResources(
        storageBuffers=[StorageBufferResource(per_tile, False, 0), StorageBufferResource(light, True, 3)],
        sampledImages=[ImageResource(color, 0, 2)]), b"", (tiles_x, tiles_y, 1)))
    be.setComputePassExecution(ComputePassExecution(p_reset, RenderPassResources(storageBuffers=[StorageBufferResource(hist, False, 1)]), b"",
                                                    (math.ceil(N_BINS / 64.0), 1, 1)))
    be.setComputePassExecution(ComputePassExecution(p_comb, RenderPassResources(
        storageBuffers=[StorageBufferResource(per_tile, False, 0), StorageBufferResource(hist, False, 1)]), b"",
        (n_tiles, math.ceil(N_BINS / 64.0), 1)))
    be.prepareForDrawcallRecording()
    be.renderFrame()
    return (be.downloadStorageBuffer(per_tile, n_tiles * N_BINS * 4, dtype=np.uint32).copy(),
            be.downloadStorageBuffer(hist, N_BINS * 4, dtype=np.uint32).copy())


def orc_histogram(color_packed, w, h, light_bytes):
    L = orc.lib()
    tiles = math.ceil(w / 32.0) * math.ceil(h / 32.0)
    src = orc.Img(color_packed, w, h, F.R11G11B10_uFloat)
    light = C.create_string_buffer(light_bytes, 20)
    per_tile = np.zeros(tiles * N_BINS, np.uint32)
    hist = np.full(N_BINS, 123456, np.uint32)
    L.orc_histogram_per_tile(src.ref(), light, _p(per_tile), C.c_uint32(N_BINS), C.c_float(MIN_LUM), C.c_float(MAX_LUM))
    L.orc_histogram_reset(_p(hist), C.c_uint32(N_BINS))
    L.orc_histogram_combine_tiles(_p(per_tile), _p(hist), C.c_uint32(N_BINS), C.c_uint32(tiles))
    return per_tile, hist


def gpu_pre_expose(be, histogram, light_bytes, lut_packed, lut_w, lut_h, global_packed):
    """computeExposure, RenderFrontend.cpp:776-790; pass creation :1689-1716"""
    global_binding(be).set(global_packed)
    light = be.createStorageBuffer(20, light_bytes)
    hist = be.createStorageBuffer(N_BINS * 4, np.asarray(histogram, np.uint32).tobytes())
    lut = be.createImage(image_desc_2d(lut_w, lut_h, F.R11G11B10_uFloat), lut_packed)
    p = be.createComputePass("preExposeLights.comp", [spec_int(0, N_BINS), spec_float(1, MIN_LUM), spec_float(2, MAX_LUM)], "Pre-expose lights")
    be.newFrame()
    be.setComputePassExecution(ComputePassExecution(p, RenderPassResources(
        storageBuffers=[StorageBufferResource(hist, False, 1), StorageBufferResource(light, False, 0)],
        sampledImages=[ImageResource(lut, 0, 2)]), b"", (1, 1, 1)))
    be.prepareForDrawcallRecording()
    be.renderFrame()
    return be.downloadStorageBuffer(light, 20, dtype=np.float32).copy()


def orc_pre_expose(histogram, light_bytes, lut_packed, lut_w, lut_h, global_packed):
    L = orc.lib()
    light = C.create_string_buffer(light_bytes, 20)
    hist = np.ascontiguousarray(histogram, np.uint32)
    lut = orc.Img(lut_packed, lut_w, lut_h, F.R11G11B10_uFloat)
    g = orc.global_from_bytes(global_packed)
    L.orc_pre_expose_lights(light, _p(hist), lut.ref(), C.byref(g), C.c_int32(N_BINS), C.c_float(MIN_LUM), C.c_float(MAX_LUM))
    return np.frombuffer(light.raw, np.float32).copy()


def gpu_tonemap(be, color_packed, w, h, global_packed, target_format=None):
    """computeTonemapping, RenderFrontend.cpp:931-945: target = swapchain input image (BGRA8) unless a format is given"""
    global_binding(be).set(global_packed)
    src = be.createImage(image_desc_2d(w, h, F.R11G11B10_uFloat), color_packed)
    if target_format is None:
        dst = be.getSwapchainInputImage()
        d = be.getImageDescription(dst)
        assert (d.width, d.height) == (w, h), "swapchain size must match for this helper"
    else:
        dst = be.createImage(image_desc_2d(w, h, target_format))
    p = be.createComputePass("tonemapping.comp", [], "Tonemap")
    be.newFrame()
    be.setComputePassExecution(ComputePassExecution(p, RenderPassResources(storageImages=[ImageResource(dst, 0, 0)], sampledImages=[ImageResource(src, 0, 1)]),
                                                    b"", (math.ceil(w / 8.0), math.ceil(h / 8.0), 1)))
    be.prepareForDrawcallRecording()
    be.renderFrame()
    return be.downloadImage(dst, 0, np.uint8).reshape(h, w, 4).copy()


def orc_tonemap(color_packed, w, h, global_packed, target_format=F.BGRA8_uNorm):
    L = orc.lib()
    src = orc.Img(color_packed, w, h, F.R11G11B10_uFloat)
    dst = orc.new_image(w, h, target_format, 4)
    g = orc.global_from_bytes(global_packed)
    L.orc_tonemapping(src.ref(), dst.ref(), C.byref(g))
    return dst.arr.reshape(h, w, 4).copy()
