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
        # re-bind every time: a FramePipeline created in between installs its own global uniform buffer
        self.be.setGlobalDescriptorSetResources(RenderPassResources(uniformBuffers=[UniformBufferResource(self.ubo, 0)]))
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
        if (d.width, d.height) != (w, h):  # an earlier test's frame pipeline resized the shared backend's swapchain
            be.recreateSwapchain(w, h)
            dst = be.getSwapchainInputImage()
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


# ------------------------------------------------------------------------------------------- HiZ
def mip_count_from_resolution(w, h, d=1):
    """Common/Utilities/MathUtils.cpp:17-19"""
    return 1 + int(math.floor(math.log2(max(w, h, d))))


def single_pass_mip_chain_dispatch(width, height, mip_count, max_mip_count=11):
    """RenderFrontend::computeSinglePassMipChainDispatchCount, RenderFrontend.cpp:1807-1827"""
    unused = max_mip_count - mip_count
    if unused >= 6:
        return 1, 1
    extent = 32 // (2 ** unused)
    return math.ceil(width / extent), math.ceil(height / extent)


def hiz_mip_sizes(w, h):
    pw, ph = w // 2, h // 2
    n = mip_count_from_resolution(pw, ph)
    return [(max(pw >> m, 1), max(ph >> m, 1)) for m in range(n)]


def gpu_hiz(be, depth_f32, w, h):
    """computeDepthPyramid, RenderFrontend.cpp:804-838; spec constants :1770-1805; pyramid image :1736-1745"""
    pw, ph = w // 2, h // 2
    mip_count = mip_count_from_resolution(pw, ph)
    depth = be.createImage(image_desc_2d(w, h, F.Depth32), np.ascontiguousarray(depth_f32, np.float32))
    pyramid = be.createImage(image_desc_2d(pw, ph, F.RG32_sFloat, MipCount.FullChain))
    sync = be.createStorageBuffer(4, struct.pack("<I", 0))
    dx, dy = single_pass_mip_chain_dispatch(pw, ph, mip_count)
    p = be.createComputePass("depthHiZPyramid.comp", [spec_int(0, mip_count), spec_int(1, w), spec_int(2, h), spec_int(3, dx * dy)], "Depth min/max pyramid")
    unused = 11 - mip_count
    storage = [ImageResource(pyramid, (i - unused) if i >= unused else 0, i) for i in range(11)]
    be.newFrame()
    be.setComputePassExecution(ComputePassExecution(p, RenderPassResources(
        storageImages=storage, sampledImages=[ImageResource(depth, 0, 13), ImageResource(pyramid, 0, 15)],
        storageBuffers=[StorageBufferResource(sync, False, 16)]), b"", (dx, dy, 1)))
    be.prepareForDrawcallRecording()
    be.renderFrame()
    return [be.downloadImage(pyramid, m, np.float32).reshape(s[1], s[0], 2).copy() for m, s in enumerate(hiz_mip_sizes(w, h))], pyramid, depth


def orc_hiz(depth_f32, w, h):
    L = orc.lib()
    sizes = hiz_mip_sizes(w, h)
    depth = orc.Img(np.ascontiguousarray(depth_f32, np.float32), w, h, F.Depth32)
    mips = [orc.new_image(s[0], s[1], F.RG32_sFloat, 8) for s in sizes]
    arr = (orc.OrcImage * len(mips))(*[m.c for m in mips])
    L.orc_depth_hiz_pyramid(depth.ref(), arr, C.c_int32(len(mips)))
    return [m.arr.view(np.float32).reshape(s[1], s[0], 2).copy() for m, s in zip(mips, sizes)]


# ------------------------------------------------------------------------------------------- bloom
BLOOM_MIPS = 6  # Techniques/Bloom.cpp:6


def bloom_mip_size(w, h, m):
    return max(w >> m, 1), max(h >> m, 1)


def gpu_bloom(be, scene_packed, w, h, strength=0.05, radius=1.5):
    """Bloom::computeBloom, Techniques/Bloom.cpp:56-143 (passes created in Bloom::init :8-40)"""
    down_p = [be.createComputePass("bloomDownsample.comp", [], "Bloom downsample mip %d" % (i + 1)) for i in range(BLOOM_MIPS - 1)]
    up_p = [be.createComputePass("bloomUpsample.comp", [spec_bool(0, i == 0)], "Bloom Upsample mip %d" % (BLOOM_MIPS - 2 - i)) for i in range(BLOOM_MIPS - 1)]
    apply_p = be.createComputePass("applyBloom.comp", [], "Apply bloom")
    target = be.createImage(image_desc_2d(w, h, F.R11G11B10_uFloat), scene_packed)
    be.newFrame()
    desc = image_desc_2d(w, h, F.R11G11B10_uFloat, MipCount.Manual, BLOOM_MIPS)
    down = be.createTemporaryImage(desc)
    for i in range(BLOOM_MIPS - 1):
        tw, th = bloom_mip_size(w, h, i + 1)
        be.setComputePassExecution(ComputePassExecution(down_p[i], RenderPassResources(
            storageImages=[ImageResource(down, i + 1, 0)], sampledImages=[ImageResource(target if i == 0 else down, i, 1)]), b"",
            (math.ceil(tw / 8.0), math.ceil(th / 8.0), 1)))
    up = be.createTemporaryImage(desc)
    for i in range(BLOOM_MIPS - 1):
        tm = BLOOM_MIPS - 2 - i
        tw, th = bloom_mip_size(w, h, tm)
        be.setComputePassExecution(ComputePassExecution(up_p[i], RenderPassResources(
            storageImages=[ImageResource(up, tm, 0)], sampledImages=[ImageResource(up, tm + 1, 1), ImageResource(down, tm + 1, 2)]),
            struct.pack("<f", radius), (math.ceil(tw / 8.0), math.ceil(th / 8.0), 1)))
    be.setComputePassExecution(ComputePassExecution(apply_p, RenderPassResources(
        storageImages=[ImageResource(target, 0, 0)], sampledImages=[ImageResource(up, 0, 1)]), struct.pack("<f", strength),
        (math.ceil(w / 8.0), math.ceil(h / 8.0), 1)))
    be.prepareForDrawcallRecording()
    be.renderFrame()
    downs = [be.downloadImage(down, m, np.uint32).copy() for m in range(1, BLOOM_MIPS)]
    ups = [be.downloadImage(up, m, np.uint32).copy() for m in range(0, BLOOM_MIPS - 1)]
    return be.downloadImage(target, 0, np.uint32).copy(), downs, ups


def orc_bloom(scene_packed, w, h, strength=0.05, radius=1.5):
    L = orc.lib()
    target = orc.Img(np.array(scene_packed, np.uint32, copy=True), w, h, F.R11G11B10_uFloat)
    down = [None] + [orc.new_image(*bloom_mip_size(w, h, m), F.R11G11B10_uFloat, 4) for m in range(1, BLOOM_MIPS)]
    up = [orc.new_image(*bloom_mip_size(w, h, m), F.R11G11B10_uFloat, 4) for m in range(0, BLOOM_MIPS)]
    for i in range(BLOOM_MIPS - 1):
        L.orc_bloom_downsample((target if i == 0 else down[i]).ref(), down[i + 1].ref())
    for i in range(BLOOM_MIPS - 1):
        tm = BLOOM_MIPS - 2 - i
        L.orc_bloom_upsample(down[tm + 1].ref(), up[tm + 1].ref(), up[tm].ref(), C.c_int32(1 if i == 0 else 0), C.c_float(radius))
    L.orc_apply_bloom(target.ref(), up[0].ref(), C.c_float(strength))
    return (target.arr.view(np.uint32).copy(), [d.arr.view(np.uint32).copy() for d in down[1:]],
            [u.arr.view(np.uint32).copy() for u in up[:BLOOM_MIPS - 1]])


# ------------------------------------------------------------------------------------------- TAA
def gpu_taa(be, current, history, motion_snorm, depth_f32, w, h, weights9, global_packed, clip=True, dilate=True, tech=4, tonemap=True):
    """TAA::computeTemporalFilter, Techniques/TAA.cpp:139-166; spec constants :204-233"""
    global_binding(be).set(global_packed)
    cur = be.createImage(image_desc_2d(w, h, F.R11G11B10_uFloat), current)
    hist_src = be.createImage(image_desc_2d(w, h, F.R11G11B10_uFloat), history)
    hist_dst = be.createImage(image_desc_2d(w, h, F.R11G11B10_uFloat))
    out = be.createImage(image_desc_2d(w, h, F.R11G11B10_uFloat))
    mot = be.createImage(image_desc_2d(w, h, F.RG16_sNorm), motion_snorm)
    dep = be.createImage(image_desc_2d(w, h, F.Depth32), np.ascontiguousarray(depth_f32, np.float32))
    wbuf = be.createUniformBuffer(36)
    be.setUniformBufferData(wbuf, np.asarray(weights9, np.float32).tobytes())
    p = be.createComputePass("temporalFilter.comp", [spec_bool(0, clip), spec_bool(1, dilate), spec_int(2, tech), spec_bool(3, tonemap)], "Temporal filtering")
    be.newFrame()
    be.setComputePassExecution(ComputePassExecution(p, RenderPassResources(
        storageImages=[ImageResource(out, 0, 1), ImageResource(hist_dst, 0, 2)],
        sampledImages=[ImageResource(cur, 0, 0), ImageResource(hist_src, 0, 3), ImageResource(mot, 0, 4), ImageResource(dep, 0, 5)],
        uniformBuffers=[UniformBufferResource(wbuf, 6)]), b"", (math.ceil(w / 8.0), math.ceil(h / 8.0), 1)))
    be.prepareForDrawcallRecording()
    be.renderFrame()
    return be.downloadImage(out, 0, np.uint32).copy(), be.downloadImage(hist_dst, 0, np.uint32).copy()


def orc_taa(current, history, motion_snorm, depth_f32, w, h, weights9, global_packed, clip=True, dilate=True, tech=4, tonemap=True):
    L = orc.lib()
    cur = orc.Img(current, w, h, F.R11G11B10_uFloat)
    hs = orc.Img(history, w, h, F.R11G11B10_uFloat)
    hd = orc.new_image(w, h, F.R11G11B10_uFloat, 4)
    out = orc.new_image(w, h, F.R11G11B10_uFloat, 4)
    mot = orc.Img(motion_snorm, w, h, F.RG16_sNorm)
    dep = orc.Img(np.ascontiguousarray(depth_f32, np.float32), w, h, F.Depth32)
    wts = np.ascontiguousarray(weights9, np.float32)
    g = orc.global_from_bytes(global_packed)
    L.orc_temporal_filter(cur.ref(), out.ref(), hd.ref(), hs.ref(), mot.ref(), dep.ref(), _p(wts), C.byref(g), C.c_int32(int(clip)), C.c_int32(int(dilate)),
                          C.c_int32(tech), C.c_int32(int(tonemap)))
    return out.arr.view(np.uint32).copy(), hd.arr.view(np.uint32).copy()


def orc_taa_weights(jitter_px):
    L = orc.lib()
    j = np.ascontiguousarray(jitter_px, np.float32)
    w = np.zeros(9, np.float32)
    L.orc_taa_resolve_weights(_p(j), _p(w))
    return w


# ------------------------------------------------------------------------------------------- SDF GI
TILE_UINTS = 101  # CulledInstancesPerTile: count + 100 indices (sdfCulling.inc:7-10)


def gpu_depth_downscale(be, depth_f32, w, h):
    """RenderFrontend::downscaleDepth, RenderFrontend.cpp:873-892"""
    full = be.createImage(image_desc_2d(w, h, F.Depth32), np.ascontiguousarray(depth_f32, np.float32))
    half = be.createImage(image_desc_2d(w // 2, h // 2, F.R16_sFloat))
    p = be.createComputePass("depthDownscale.comp", [], "Depth downscale")
    be.newFrame()
    be.setComputePassExecution(ComputePassExecution(p, RenderPassResources(storageImages=[ImageResource(half, 0, 0)], sampledImages=[ImageResource(full, 0, 1)]),
                                                    b"", (math.ceil((w // 2) / 8.0), math.ceil((h // 2) / 8.0), 1)))
    be.renderFrame()
    return be.downloadImage(half, 0, np.uint16).copy()


def orc_depth_downscale(depth_f32, w, h):
    L = orc.lib()
    full = orc.Img(np.ascontiguousarray(depth_f32, np.float32), w, h, F.Depth32)
    half = orc.new_image(w // 2, h // 2, F.R16_sFloat, 2)
    L.orc_depth_downscale(full.ref(), half.ref())
    return half.arr.view(np.uint16).copy()


def gpu_sdf_culling(be, instance_bytes, bb_bytes, frustum_pts, frustum_nrm, influence, pyramid_image, trace_w, trace_h, global_packed, use_hiz=True,
                    screen_w=None):
    """SDFGI::sdfInstanceCulling, Techniques/SDFGI.cpp:538-630"""
    global_binding(be).set(global_packed)
    n = struct.unpack("<I", instance_bytes[:4])[0]
    inst = be.createStorageBuffer(len(instance_bytes), instance_bytes)
    culled = be.createStorageBuffer(4 + 4 * max(n, 1))
    bbs = be.createStorageBuffer(max(len(bb_bytes), 32), bb_bytes)
    frustum = be.createUniformBuffer(12 * 16)
    infl = be.createUniformBuffer(4)
    tcx, tcy = math.ceil(trace_w / 32.0), math.ceil(trace_h / 32.0)
    stride = math.ceil((screen_w or trace_w * 2) / 32.0)
    n_tiles = stride * tcy
    tiles = be.createStorageBuffer(n_tiles * TILE_UINTS * 4)
    p_fr = be.createComputePass("sdfCameraFrustumCulling.comp", [], "SDF camera frustum culling")
    p_tile = be.createComputePass("sdfCameraTileCulling.comp", [spec_bool(0, use_hiz)], "SDF camera tile culling")
    be.newFrame()
    be.setUniformBufferData(frustum, np.concatenate([frustum_pts.reshape(-1), frustum_nrm.reshape(-1)]).astype(np.float32).tobytes())
    be.setStorageBufferData(culled, struct.pack("<I", 0))
    be.setComputePassExecution(ComputePassExecution(p_fr, RenderPassResources(
        storageBuffers=[StorageBufferResource(inst, True, 0), StorageBufferResource(culled, False, 2), StorageBufferResource(bbs, True, 3)],
        uniformBuffers=[UniformBufferResource(frustum, 1), UniformBufferResource(infl, 4)]), b"", (math.ceil(n / 64.0), 1, 1)))
    be.setUniformBufferData(infl, struct.pack("<f", influence))
    res = RenderPassResources(
        storageBuffers=[StorageBufferResource(culled, True, 0), StorageBufferResource(bbs, True, 1), StorageBufferResource(tiles, False, 2)],
        uniformBuffers=[UniformBufferResource(infl, 3)])
    if pyramid_image is not None:
        res.sampledImages = [ImageResource(pyramid_image, 4, 4)]
    be.setComputePassExecution(ComputePassExecution(p_tile, res, struct.pack("<2I", tcx, tcy), (math.ceil(tcx / 8.0), math.ceil(tcy / 8.0), 1)))
    be.renderFrame()
    return (be.downloadStorageBuffer(culled, 4 + 4 * max(n, 1), dtype=np.uint32).copy(),
            be.downloadStorageBuffer(tiles, n_tiles * TILE_UINTS * 4, dtype=np.uint32).copy(), dict(inst=inst, tiles=tiles, infl=infl, bbs=bbs, culled=culled))


def orc_sdf_culling(instance_bytes, bb_bytes, frustum_pts, frustum_nrm, influence, hiz_mip4, trace_w, trace_h, global_packed, use_hiz=True, screen_w=None):
    L = orc.lib()
    n = struct.unpack("<I", instance_bytes[:4])[0]
    bbs = np.frombuffer(bb_bytes, np.float32).copy()
    culled = np.zeros(1 + max(n, 1), np.uint32)
    fp, fn = np.ascontiguousarray(frustum_pts, np.float32), np.ascontiguousarray(frustum_nrm, np.float32)
    L.orc_sdf_camera_frustum_culling(C.c_uint32(n), _p(fp), _p(fn), _p(bbs), C.c_float(influence), _p(culled))
    tcx, tcy = math.ceil(trace_w / 32.0), math.ceil(trace_h / 32.0)
    stride = math.ceil((screen_w or trace_w * 2) / 32.0)
    tiles = np.zeros(stride * tcy * TILE_UINTS, np.uint32)
    g = orc.global_from_bytes(global_packed)
    mip = None
    if hiz_mip4 is not None:
        a = np.ascontiguousarray(hiz_mip4, np.float32)
        mip = orc.Img(a, a.shape[1], a.shape[0], F.RG32_sFloat)
    L.orc_sdf_camera_tile_culling(_p(culled), _p(bbs), _p(tiles), C.c_float(influence), mip.ref() if mip else None, C.byref(g), C.c_int32(int(use_hiz)),
                                  C.c_uint32(tcx), C.c_uint32(tcy))
    return culled, tiles


class GiImages:
    """images/buffers of one GI test scene on the backend"""
    pass


def make_bindless(be, volumes_u16, sdf_res, noise_list):
    """registers SDF volumes + noise textures as default images; returns (handles, global indices, oracle image array)"""
    from plainrenderer_amd.backend import ImageDescription, ImageType, ImageUsageFlags
    vol_handles, vol_idx = [], []
    for v in volumes_u16:
        d = ImageDescription(width=sdf_res, height=sdf_res, depth=sdf_res, type=ImageType.Type3D, format=F.R16_sFloat, usageFlags=int(ImageUsageFlags.Sampled))
        hnd = be.createImage(d, np.ascontiguousarray(v))
        vol_handles.append(hnd)
        vol_idx.append(be.getImageGlobalTextureArrayIndex(hnd))
    noise_idx = []
    for nz in noise_list:
        hnd = be.createImage(image_desc_2d(nz.shape[1], nz.shape[0], F.RG8), np.ascontiguousarray(nz))
        noise_idx.append(be.getImageGlobalTextureArrayIndex(hnd))
    return vol_idx, noise_idx


def orc_bindless(volumes_u16, sdf_res, noise_list, vol_idx, noise_idx):
    """oracle-side global texture array laid out with the same indices as the backend's"""
    n = max(vol_idx + noise_idx) + 1
    arr = (orc.OrcImage * n)()
    keep = []
    for v, i in zip(volumes_u16, vol_idx):
        im = orc.Img(np.ascontiguousarray(v), sdf_res, sdf_res, F.R16_sFloat, d=sdf_res)
        keep.append(im)
        arr[i] = im.c
    for nz, i in zip(noise_list, noise_idx):
        im = orc.Img(np.ascontiguousarray(nz), nz.shape[1], nz.shape[0], F.RG8)
        keep.append(im)
        arr[i] = im.c
    return arr, n, keep


def patch_instance_texture_indices(instance_bytes, vol_idx):
    b = bytearray(instance_bytes)
    for i, ti in enumerate(vol_idx):
        struct.pack_into("<I", b, 16 + i * 96 + 12, ti)
    return bytes(b)


def gpu_sdf_trace(be, depth_f32, normal_rgba8, w, h, trace_w, trace_h, sky_packed, sky_w, sky_h, light_bytes, instance_bytes, tiles_u32, influence, shadow_info,
                  shadow_map_u16, shadow_res, global_packed, strict=True, cascade=2):
    """SDFGI::diffuseSDFTrace, Techniques/SDFGI.cpp:380-419; spec constants :30-46"""
    global_binding(be).set(global_packed)
    out_ysh = be.createImage(image_desc_2d(trace_w, trace_h, F.RGBA16_sFloat))
    out_cocg = be.createImage(image_desc_2d(trace_w, trace_h, F.RG16_sFloat))
    depth = be.createImage(image_desc_2d(w, h, F.Depth32), np.ascontiguousarray(depth_f32, np.float32))
    normal = be.createImage(image_desc_2d(w, h, F.RGBA8), np.ascontiguousarray(normal_rgba8))
    sky = be.createImage(image_desc_2d(sky_w, sky_h, F.R11G11B10_uFloat), sky_packed)
    shadow = be.createImage(image_desc_2d(shadow_res, shadow_res, F.Depth16), np.ascontiguousarray(shadow_map_u16))
    light = be.createStorageBuffer(20, light_bytes)
    inst = be.createStorageBuffer(len(instance_bytes), instance_bytes)
    tiles = be.createStorageBuffer(tiles_u32.nbytes, tiles_u32.tobytes())
    infl = be.createUniformBuffer(4, struct.pack("<f", influence))
    sinfo = be.createStorageBuffer(304, shadow_info)
    p = be.createComputePass("sdfDiffuseTrace.comp", [spec_bool(0, strict), spec_int(1, cascade)], "Indirect diffuse SDF trace")
    be.newFrame()
    be.setComputePassExecution(ComputePassExecution(p, RenderPassResources(
        storageImages=[ImageResource(out_ysh, 0, 0), ImageResource(out_cocg, 0, 1)],
        sampledImages=[ImageResource(depth, 0, 2), ImageResource(normal, 0, 3), ImageResource(sky, 0, 4), ImageResource(shadow, 0, 10)],
        storageBuffers=[StorageBufferResource(light, True, 5), StorageBufferResource(inst, True, 6), StorageBufferResource(tiles, True, 7),
                        StorageBufferResource(sinfo, True, 9)],
        uniformBuffers=[UniformBufferResource(infl, 8)]), b"", (math.ceil(trace_w / 8.0), math.ceil(trace_h / 8.0), 1)))
    be.renderFrame()
    return be.downloadImage(out_ysh, 0, np.uint16).copy(), be.downloadImage(out_cocg, 0, np.uint16).copy()


def orc_sdf_trace(depth_f32, normal_rgba8, w, h, trace_w, trace_h, sky_packed, sky_w, sky_h, light_bytes, instance_bytes, tiles_u32, influence, shadow_info,
                  shadow_map_u16, shadow_res, global_packed, bindless_arr, n_bindless, strict=True, cascade=2):
    L = orc.lib()
    out_ysh = orc.new_image(trace_w, trace_h, F.RGBA16_sFloat, 8)
    out_cocg = orc.new_image(trace_w, trace_h, F.RG16_sFloat, 4)
    depth = orc.Img(np.ascontiguousarray(depth_f32, np.float32), w, h, F.Depth32)
    normal = orc.Img(np.ascontiguousarray(normal_rgba8), w, h, F.RGBA8)
    sky = orc.Img(sky_packed, sky_w, sky_h, F.R11G11B10_uFloat)
    shadow = orc.Img(np.ascontiguousarray(shadow_map_u16), shadow_res, shadow_res, F.Depth16)
    light = C.create_string_buffer(light_bytes, 20)
    inst = C.create_string_buffer(instance_bytes[16:], len(instance_bytes) - 16)
    tiles = np.ascontiguousarray(tiles_u32, np.uint32)
    sinfo = C.create_string_buffer(shadow_info, 304)
    g = orc.global_from_bytes(global_packed)
    L.orc_sdf_diffuse_trace(out_ysh.ref(), out_cocg.ref(), depth.ref(), normal.ref(), sky.ref(), light, inst, _p(tiles), C.c_float(influence), sinfo, shadow.ref(),
                            bindless_arr, C.c_int32(n_bindless), C.byref(g), C.c_int32(int(strict)), C.c_int32(cascade))
    return out_ysh.arr.view(np.uint16).copy(), out_cocg.arr.view(np.uint16).copy()


def gpu_gi_spatial(be, ysh_u16, cocg_u16, tw, th, depth_arr, depth_fmt, dw, dh, normal_rgba8, w, h, global_packed, filter_index):
    """SDFGI::filterIndirectDiffuse spatial passes, Techniques/SDFGI.cpp:428-451,483-506"""
    global_binding(be).set(global_packed)
    in_y = be.createImage(image_desc_2d(tw, th, F.RGBA16_sFloat), np.ascontiguousarray(ysh_u16))
    in_c = be.createImage(image_desc_2d(tw, th, F.RG16_sFloat), np.ascontiguousarray(cocg_u16))
    out_y = be.createImage(image_desc_2d(tw, th, F.RGBA16_sFloat))
    out_c = be.createImage(image_desc_2d(tw, th, F.RG16_sFloat))
    dep = be.createImage(image_desc_2d(dw, dh, depth_fmt), np.ascontiguousarray(depth_arr))
    nrm = be.createImage(image_desc_2d(w, h, F.RGBA8), np.ascontiguousarray(normal_rgba8))
    p = be.createComputePass("filterIndirectDiffuseSpatial.comp", [spec_int(0, filter_index)], "Indirect diffuse spatial filter")
    be.newFrame()
    be.setComputePassExecution(ComputePassExecution(p, RenderPassResources(
        storageImages=[ImageResource(out_y, 0, 0), ImageResource(out_c, 0, 1)],
        sampledImages=[ImageResource(in_y, 0, 2), ImageResource(in_c, 0, 3), ImageResource(dep, 0, 4), ImageResource(nrm, 0, 5)]), b"",
        (math.ceil(tw / 8.0), math.ceil(th / 8.0), 1)))
    be.renderFrame()
    return be.downloadImage(out_y, 0, np.uint16).copy(), be.downloadImage(out_c, 0, np.uint16).copy()


def orc_gi_spatial(ysh_u16, cocg_u16, tw, th, depth_arr, depth_fmt, dw, dh, normal_rgba8, w, h, global_packed, filter_index):
    L = orc.lib()
    in_y = orc.Img(np.ascontiguousarray(ysh_u16), tw, th, F.RGBA16_sFloat)
    in_c = orc.Img(np.ascontiguousarray(cocg_u16), tw, th, F.RG16_sFloat)
    out_y = orc.new_image(tw, th, F.RGBA16_sFloat, 8)
    out_c = orc.new_image(tw, th, F.RG16_sFloat, 4)
    dep = orc.Img(np.ascontiguousarray(depth_arr), dw, dh, depth_fmt)
    nrm = orc.Img(np.ascontiguousarray(normal_rgba8), w, h, F.RGBA8)
    g = orc.global_from_bytes(global_packed)
    L.orc_filter_indirect_diffuse_spatial(out_y.ref(), out_c.ref(), in_y.ref(), in_c.ref(), dep.ref(), nrm.ref(), C.byref(g), C.c_int32(filter_index))
    return out_y.arr.view(np.uint16).copy(), out_c.arr.view(np.uint16).copy()


def gpu_gi_temporal(be, in_y, in_c, hist_y, hist_c, tw, th, motion_cur, motion_last, w, h, global_packed):
    """temporal filter, Techniques/SDFGI.cpp:452-482"""
    global_binding(be).set(global_packed)
    mk = lambda fmt, data=None, ww=tw, hh=th: be.createImage(image_desc_2d(ww, hh, fmt), None if data is None else np.ascontiguousarray(data))
    i_y, i_c, h_y, h_c = mk(F.RGBA16_sFloat, in_y), mk(F.RG16_sFloat, in_c), mk(F.RGBA16_sFloat, hist_y), mk(F.RG16_sFloat, hist_c)
    t_y, t_c, o_y, o_c = mk(F.RGBA16_sFloat), mk(F.RG16_sFloat), mk(F.RGBA16_sFloat), mk(F.RG16_sFloat)
    mc, ml = mk(F.RG16_sNorm, motion_cur, w, h), mk(F.RG16_sNorm, motion_last, w, h)
    p = be.createComputePass("filterIndirectDiffuseTemporal.comp", [], "Indirect diffuse temporal filter")
    be.newFrame()
    be.setComputePassExecution(ComputePassExecution(p, RenderPassResources(
        storageImages=[ImageResource(t_y, 0, 0), ImageResource(t_c, 0, 1), ImageResource(o_y, 0, 2), ImageResource(o_c, 0, 3)],
        sampledImages=[ImageResource(i_y, 0, 4), ImageResource(i_c, 0, 5), ImageResource(h_y, 0, 6), ImageResource(h_c, 0, 7), ImageResource(mc, 0, 8),
                       ImageResource(ml, 0, 9)]), b"", (math.ceil(tw / 8.0), math.ceil(th / 8.0), 1)))
    be.renderFrame()
    dl = lambda im: be.downloadImage(im, 0, np.uint16).copy()
    return dl(t_y), dl(t_c), dl(o_y), dl(o_c)


def orc_gi_temporal(in_y, in_c, hist_y, hist_c, tw, th, motion_cur, motion_last, w, h, global_packed):
    L = orc.lib()
    I = lambda a, fmt, ww=tw, hh=th: orc.Img(np.ascontiguousarray(a), ww, hh, fmt)
    i_y, i_c, h_y, h_c = I(in_y, F.RGBA16_sFloat), I(in_c, F.RG16_sFloat), I(hist_y, F.RGBA16_sFloat), I(hist_c, F.RG16_sFloat)
    t_y, t_c = orc.new_image(tw, th, F.RGBA16_sFloat, 8), orc.new_image(tw, th, F.RG16_sFloat, 4)
    o_y, o_c = orc.new_image(tw, th, F.RGBA16_sFloat, 8), orc.new_image(tw, th, F.RG16_sFloat, 4)
    mc, ml = I(motion_cur, F.RG16_sNorm, w, h), I(motion_last, F.RG16_sNorm, w, h)
    g = orc.global_from_bytes(global_packed)
    L.orc_filter_indirect_diffuse_temporal(t_y.ref(), t_c.ref(), o_y.ref(), o_c.ref(), i_y.ref(), i_c.ref(), h_y.ref(), h_c.ref(), mc.ref(), ml.ref(), C.byref(g))
    v = lambda im: im.arr.view(np.uint16).copy()
    return v(t_y), v(t_c), v(o_y), v(o_c)


def gpu_gi_upscale(be, ysh, cocg, tw, th, depth_f32, half_depth_u16, w, h, global_packed):
    """upscale, Techniques/SDFGI.cpp:510-535"""
    global_binding(be).set(global_packed)
    s_y = be.createImage(image_desc_2d(tw, th, F.RGBA16_sFloat), np.ascontiguousarray(ysh))
    s_c = be.createImage(image_desc_2d(tw, th, F.RG16_sFloat), np.ascontiguousarray(cocg))
    d_y = be.createImage(image_desc_2d(w, h, F.RGBA16_sFloat))
    d_c = be.createImage(image_desc_2d(w, h, F.RG16_sFloat))
    fd = be.createImage(image_desc_2d(w, h, F.Depth32), np.ascontiguousarray(depth_f32, np.float32))
    hd = be.createImage(image_desc_2d(tw, th, F.R16_sFloat), np.ascontiguousarray(half_depth_u16))
    p = be.createComputePass("indirectLightUpscale.comp", [], "Indirect lighting upscale")
    be.newFrame()
    be.setComputePassExecution(ComputePassExecution(p, RenderPassResources(
        storageImages=[ImageResource(d_y, 0, 0), ImageResource(d_c, 0, 1)],
        sampledImages=[ImageResource(s_y, 0, 2), ImageResource(s_c, 0, 3), ImageResource(fd, 0, 4), ImageResource(hd, 0, 5)]), b"",
        (math.ceil(w / 8.0), math.ceil(h / 8.0), 1)))
    be.renderFrame()
    return be.downloadImage(d_y, 0, np.uint16).copy(), be.downloadImage(d_c, 0, np.uint16).copy()


def orc_gi_upscale(ysh, cocg, tw, th, depth_f32, half_depth_u16, w, h, global_packed):
    L = orc.lib()
    s_y, s_c = orc.Img(np.ascontiguousarray(ysh), tw, th, F.RGBA16_sFloat), orc.Img(np.ascontiguousarray(cocg), tw, th, F.RG16_sFloat)
    d_y, d_c = orc.new_image(w, h, F.RGBA16_sFloat, 8), orc.new_image(w, h, F.RG16_sFloat, 4)
    fd = orc.Img(np.ascontiguousarray(depth_f32, np.float32), w, h, F.Depth32)
    hd = orc.Img(np.ascontiguousarray(half_depth_u16), tw, th, F.R16_sFloat)
    g = orc.global_from_bytes(global_packed)
    L.orc_indirect_light_upscale(d_y.ref(), d_c.ref(), s_y.ref(), s_c.ref(), fd.ref(), hd.ref(), C.byref(g))
    return d_y.arr.view(np.uint16).copy(), d_c.arr.view(np.uint16).copy()


# ------------------------------------------------------------------------------------------- shading
def gpu_brdf_lut(be, res, diffuse_brdf=2):
    """RenderFrontend::computeBRDFLut, RenderFrontend.cpp:1031-1042 (512^2 RGBA16F in the reference)"""
    lut = be.createImage(image_desc_2d(res, res, F.RGBA16_sFloat))
    p = be.createComputePass("brdfLut.comp", [spec_int(0, diffuse_brdf)], "BRDF Lut creation")
    be.newFrame()
    be.setComputePassExecution(ComputePassExecution(p, RenderPassResources(storageImages=[ImageResource(lut, 0, 0)]), b"", (math.ceil(res / 8.0), math.ceil(res / 8.0), 1)))
    be.renderFrame()
    return be.downloadImage(lut, 0, np.uint16).copy(), lut


def orc_brdf_lut(res, diffuse_brdf=2):
    L = orc.lib()
    lut = orc.new_image(res, res, F.RGBA16_sFloat, 8)
    L.orc_brdf_lut(lut.ref(), C.c_int32(diffuse_brdf))
    return lut.arr.view(np.uint16).copy()


def gpu_deferred_shading(be, gb, w, h, brdf_lut_u16, lut_res, light_bytes, shadow_info, shadow_maps, shadow_res, ysh_u16, cocg_u16, froxel_u16, froxel_dims,
                         vol_settings, sky_packed, global_packed, diffuse_brdf=2, multiscatter=0, geometric_aa=True, indirect_tech=0, cascades=3):
    """deferred re-expression of renderForwardShading, RenderFrontend.cpp:894-929; spec constants :1093-1131"""
    from plainrenderer_amd.backend import ImageDescription, ImageType, ImageUsageFlags
    global_binding(be).set(global_packed)
    mk = lambda fmt, data, ww=w, hh=h: be.createImage(image_desc_2d(ww, hh, fmt), np.ascontiguousarray(data))
    color = be.createImage(image_desc_2d(w, h, F.R11G11B10_uFloat))
    depth, normal = mk(F.Depth32, gb["depth"]), mk(F.RGBA8, gb["normal"])
    albedo, spec = mk(F.RGBA8, gb["albedo"]), mk(F.RGBA8, gb["specular"])
    lut = mk(F.RGBA16_sFloat, brdf_lut_u16, lut_res, lut_res)
    smaps = [mk(F.Depth16, m, shadow_res, shadow_res) for m in shadow_maps]
    ysh, cocg = mk(F.RGBA16_sFloat, ysh_u16), mk(F.RG16_sFloat, cocg_u16)
    fw, fh, fd = froxel_dims
    vol = be.createImage(ImageDescription(width=fw, height=fh, depth=fd, type=ImageType.Type3D, format=F.RGBA16_sFloat, usageFlags=int(ImageUsageFlags.Sampled)),
                         np.ascontiguousarray(froxel_u16))
    sky = mk(F.R11G11B10_uFloat, sky_packed, 200, 100)
    light = be.createStorageBuffer(20, light_bytes)
    sinfo = be.createStorageBuffer(304, shadow_info)
    vset = be.createUniformBuffer(64, vol_settings)
    p = be.createComputePass("deferredShading.comp", [spec_int(0, diffuse_brdf), spec_int(1, multiscatter), spec_bool(2, geometric_aa), spec_int(3, indirect_tech),
                                                      spec_uint(4, cascades)], "Forward shading (deferred)")
    be.newFrame()
    sampled = [ImageResource(lut, 0, 3), ImageResource(ysh, 0, 15), ImageResource(cocg, 0, 16), ImageResource(vol, 0, 18), ImageResource(depth, 0, 20),
               ImageResource(normal, 0, 21), ImageResource(albedo, 0, 22), ImageResource(spec, 0, 23), ImageResource(sky, 0, 24)]
    sampled += [ImageResource(smaps[i], 0, 9 + i) for i in range(4)]
    be.setComputePassExecution(ComputePassExecution(p, RenderPassResources(
        storageImages=[ImageResource(color, 0, 0)], sampledImages=sampled,
        storageBuffers=[StorageBufferResource(light, True, 7), StorageBufferResource(sinfo, True, 8)],
        uniformBuffers=[UniformBufferResource(vset, 19)]), b"", (math.ceil(w / 8.0), math.ceil(h / 8.0), 1)))
    be.renderFrame()
    return be.downloadImage(color, 0, np.uint32).copy()


def gpu_upscale_and_shade(be, half_ysh, half_cocg, tw, th, half_depth_u16, gb, w, h, brdf_lut_u16, lut_res, light_bytes, shadow_info, shadow_maps, shadow_res, froxel_u16,
                          froxel_dims, vol_settings, sky_packed, global_packed, diffuse_brdf=2, multiscatter=0, geometric_aa=True, cascades=3, download_upscaled=False):
    """indirectLightUpscale.comp and the deferred shade recorded back to back, the shade sampling the images the upscale writes
    (Techniques/SDFGI.cpp:510-535 followed by RenderFrontend.cpp:894-929): in PLR_MATH_FAST with pass fusion this is ONE launch
    (kernels_fast/shading_fast.hip upscaleAndShadeQuadKernel). Returns the colour image (and the upscaled Y_SH / CoCg images if asked)."""
    from plainrenderer_amd.backend import ImageDescription, ImageType, ImageUsageFlags
    global_binding(be).set(global_packed)
    mk = lambda fmt, data, ww=w, hh=h: be.createImage(image_desc_2d(ww, hh, fmt), np.ascontiguousarray(data))
    s_y, s_c = mk(F.RGBA16_sFloat, half_ysh, tw, th), mk(F.RG16_sFloat, half_cocg, tw, th)
    hd = mk(F.R16_sFloat, half_depth_u16, tw, th)
    d_y, d_c = be.createImage(image_desc_2d(w, h, F.RGBA16_sFloat)), be.createImage(image_desc_2d(w, h, F.RG16_sFloat))
    color = be.createImage(image_desc_2d(w, h, F.R11G11B10_uFloat))
    depth, normal = mk(F.Depth32, np.ascontiguousarray(gb["depth"], np.float32)), mk(F.RGBA8, gb["normal"])
    albedo, spec = mk(F.RGBA8, gb["albedo"]), mk(F.RGBA8, gb["specular"])
    lut = mk(F.RGBA16_sFloat, brdf_lut_u16, lut_res, lut_res)
    smaps = [mk(F.Depth16, m, shadow_res, shadow_res) for m in shadow_maps]
    fw, fh, fd = froxel_dims
    vol = be.createImage(ImageDescription(width=fw, height=fh, depth=fd, type=ImageType.Type3D, format=F.RGBA16_sFloat, usageFlags=int(ImageUsageFlags.Sampled)),
                         np.ascontiguousarray(froxel_u16))
    sky = mk(F.R11G11B10_uFloat, sky_packed, 200, 100)
    light = be.createStorageBuffer(20, light_bytes)
    sinfo = be.createStorageBuffer(304, shadow_info)
    vset = be.createUniformBuffer(64, vol_settings)
    pu = be.createComputePass("indirectLightUpscale.comp", [], "Indirect lighting upscale")
    ps = be.createComputePass("deferredShading.comp", [spec_int(0, diffuse_brdf), spec_int(1, multiscatter), spec_bool(2, geometric_aa), spec_int(3, 0), spec_uint(4, cascades)],
                              "Forward shading (deferred)")
    be.newFrame()
    groups = (math.ceil(w / 8.0), math.ceil(h / 8.0), 1)
    be.setComputePassExecution(ComputePassExecution(pu, RenderPassResources(
        storageImages=[ImageResource(d_y, 0, 0), ImageResource(d_c, 0, 1)],
        sampledImages=[ImageResource(s_y, 0, 2), ImageResource(s_c, 0, 3), ImageResource(depth, 0, 4), ImageResource(hd, 0, 5)]), b"", groups))
    sampled = [ImageResource(lut, 0, 3), ImageResource(d_y, 0, 15), ImageResource(d_c, 0, 16), ImageResource(vol, 0, 18), ImageResource(depth, 0, 20),
               ImageResource(normal, 0, 21), ImageResource(albedo, 0, 22), ImageResource(spec, 0, 23), ImageResource(sky, 0, 24)]
    sampled += [ImageResource(smaps[i], 0, 9 + i) for i in range(4)]
    be.setComputePassExecution(ComputePassExecution(ps, RenderPassResources(
        storageImages=[ImageResource(color, 0, 0)], sampledImages=sampled,
        storageBuffers=[StorageBufferResource(light, True, 7), StorageBufferResource(sinfo, True, 8)],
        uniformBuffers=[UniformBufferResource(vset, 19)]), b"", groups))
    be.renderFrame()
    out = be.downloadImage(color, 0, np.uint32).copy()
    if download_upscaled:
        return out, be.downloadImage(d_y, 0, np.uint16).copy(), be.downloadImage(d_c, 0, np.uint16).copy()
    return out


def orc_deferred_shading(gb, w, h, brdf_lut_u16, lut_res, light_bytes, shadow_info, shadow_maps, shadow_res, ysh_u16, cocg_u16, froxel_u16, froxel_dims, vol_settings,
                         sky_packed, global_packed, bindless_arr, n_bindless, diffuse_brdf=2, multiscatter=0, geometric_aa=True, indirect_tech=0, cascades=3):
    L = orc.lib()
    I = lambda a, fmt, ww=w, hh=h, d=1: orc.Img(np.ascontiguousarray(a), ww, hh, fmt, d)
    color = orc.new_image(w, h, F.R11G11B10_uFloat, 4)
    depth, normal, albedo, spec = I(gb["depth"], F.Depth32), I(gb["normal"], F.RGBA8), I(gb["albedo"], F.RGBA8), I(gb["specular"], F.RGBA8)
    lut = I(brdf_lut_u16, F.RGBA16_sFloat, lut_res, lut_res)
    smaps = [I(m, F.Depth16, shadow_res, shadow_res) for m in shadow_maps]
    sm_arr = (orc.OrcImage * 4)(*[m.c for m in smaps])
    ysh, cocg = I(ysh_u16, F.RGBA16_sFloat), I(cocg_u16, F.RG16_sFloat)
    fw, fh, fd = froxel_dims
    vol = I(froxel_u16, F.RGBA16_sFloat, fw, fh, fd)
    sky = I(sky_packed, F.R11G11B10_uFloat, 200, 100)
    light = C.create_string_buffer(light_bytes, 20)
    sinfo = C.create_string_buffer(shadow_info, 304)
    vset = C.create_string_buffer(vol_settings, 64)
    g = orc.global_from_bytes(global_packed)
    L.orc_deferred_shading(color.ref(), depth.ref(), normal.ref(), albedo.ref(), spec.ref(), lut.ref(), light, sinfo, sm_arr, ysh.ref(), cocg.ref(), vol.ref(), vset,
                           sky.ref(), bindless_arr, C.c_int32(n_bindless), C.byref(g), C.c_int32(diffuse_brdf), C.c_int32(multiscatter), C.c_int32(int(geometric_aa)),
                           C.c_int32(indirect_tech), C.c_uint32(cascades))
    return color.arr.view(np.uint32).copy()


# ------------------------------------------------------------------ input producers
def gpu_light_matrix(be, info_bytes, apex_min_max, global_packed, cascade_count, padding, min_far):
    """lightMatrix.comp: binding 0 = sunShadowInfo (std430, 304 B), storage image 1 = lowest HiZ mip (1x1 RG32F); push = 2 floats"""
    info = be.createStorageBuffer(304, info_bytes)
    apex = be.createImage(image_desc_2d(1, 1, F.RG32_sFloat), np.asarray(apex_min_max, np.float32))
    p = be.createComputePass("lightMatrix.comp", [spec_uint(0, cascade_count)], "Compute light matrix")
    gb = global_binding(be)
    gb.set(global_packed)
    be.newFrame()
    be.setComputePassExecution(ComputePassExecution(p, RenderPassResources(storageBuffers=[StorageBufferResource(info, False, 0)],
                                                                           storageImages=[ImageResource(apex, 0, 1)]), struct.pack("<2f", padding, min_far), (1, 1, 1)))
    be.prepareForDrawcallRecording()
    be.renderFrame()
    return be.downloadStorageBuffer(info, 304).tobytes()


def orc_light_matrix(info_bytes, apex_min_max, global_packed, cascade_count, padding, min_far):
    L = orc.lib()
    info = (C.c_uint8 * 304).from_buffer_copy(info_bytes)
    apex = np.asarray(apex_min_max, np.float32)
    g = orc.global_from_bytes(global_packed)
    L.orc_light_matrix(C.byref(info), _p(apex), C.byref(g), C.c_uint32(cascade_count), C.c_float(padding), C.c_float(min_far))
    return bytes(info)


ATMOSPHERE_DEFAULT = struct.pack("<14f", 0.0058, 0.0135, 0.0331, 6371.0, 0.0058, 0.0135, 0.0331, 100.0, 0.000650, 0.001881, 0.000085, 0.006, 1.11 * 0.006, 0.76)  # Sky.h:6-15


def gpu_sky_luts(be, atmosphere_bytes, light_bytes, global_packed, t_res=128, m_res=32, sky_w=200, sky_h=100, given_transmission=None, given_multiscatter=None):
    """Sky::updateTransmissionLut + Sky::updateSkyLut (Techniques/Sky.cpp:260-316): bindings as recorded there.
    given_transmission / given_multiscatter: packed texels to upload instead of recording that LUT's pass (a later pass alone on known inputs)"""
    t = be.createImage(image_desc_2d(t_res, t_res, F.R11G11B10_uFloat), None if given_transmission is None else np.ascontiguousarray(given_transmission))
    m = be.createImage(image_desc_2d(m_res, m_res, F.R11G11B10_uFloat), None if given_multiscatter is None else np.ascontiguousarray(given_multiscatter))
    s = be.createImage(image_desc_2d(sky_w, sky_h, F.R11G11B10_uFloat))
    atm = be.createUniformBuffer(64, atmosphere_bytes)
    light = be.createStorageBuffer(20, light_bytes)
    pt = be.createComputePass("skyTransmissionLut.comp", [], "Sky transmission lut")
    pm = be.createComputePass("skyMultiscatterLut.comp", [], "Sky multiscatter lut")
    ps = be.createComputePass("skyLut.comp", [], "Sky lut")
    gb = global_binding(be)
    gb.set(global_packed)
    be.newFrame()
    if given_transmission is None:
        be.setComputePassExecution(ComputePassExecution(pt, RenderPassResources(storageImages=[ImageResource(t, 0, 0)], uniformBuffers=[UniformBufferResource(atm, 1)]), b"",
                                                        (t_res // 8, t_res // 8, 1)))
    if given_multiscatter is None:
        be.setComputePassExecution(ComputePassExecution(pm, RenderPassResources(storageImages=[ImageResource(m, 0, 0)], sampledImages=[ImageResource(t, 0, 1)],
                                                                                uniformBuffers=[UniformBufferResource(atm, 3)]), b"", (m_res // 8, m_res // 8, 1)))
    be.setComputePassExecution(ComputePassExecution(ps, RenderPassResources(storageImages=[ImageResource(s, 0, 0)], sampledImages=[ImageResource(t, 0, 1), ImageResource(m, 0, 2)],
                                                                            uniformBuffers=[UniformBufferResource(atm, 4)], storageBuffers=[StorageBufferResource(light, True, 5)]),
                                                    b"", (sky_w // 8, sky_h // 8, 1)))
    be.prepareForDrawcallRecording()
    be.renderFrame()
    return (be.downloadImage(t, 0, np.uint32).reshape(t_res, t_res).copy(), be.downloadImage(m, 0, np.uint32).reshape(m_res, m_res).copy(),
            be.downloadImage(s, 0, np.uint32).reshape(sky_h, sky_w).copy())


def orc_sky_luts(atmosphere_bytes, light_bytes, global_packed, t_res=128, m_res=32, sky_w=200, sky_h=100):
    L = orc.lib()
    t = orc.new_image(t_res, t_res, F.R11G11B10_uFloat, 4)
    m = orc.new_image(m_res, m_res, F.R11G11B10_uFloat, 4)
    s = orc.new_image(sky_w, sky_h, F.R11G11B10_uFloat, 4)
    atm = (C.c_uint8 * 56).from_buffer_copy(atmosphere_bytes[:56])
    light = (C.c_uint8 * 20).from_buffer_copy(light_bytes)
    g = orc.global_from_bytes(global_packed)
    L.orc_sky_transmission_lut(t.ref(), C.byref(atm))
    L.orc_sky_multiscatter_lut(m.ref(), t.ref(), C.byref(atm))
    L.orc_sky_lut(s.ref(), t.ref(), m.ref(), C.byref(atm), C.byref(light), C.byref(g))
    out = [t.arr.view(np.uint32).reshape(t_res, t_res).copy(), m.arr.view(np.uint32).reshape(m_res, m_res).copy(), s.arr.view(np.uint32).reshape(sky_h, sky_w).copy()]
    # the recorded dispatches are size / 8 workgroups (integer division, Sky.cpp:268-313): 100 / 8 = 12 groups leave the last 4 rows of the
    # sky LUT unwritten (still the zeros the image was created with)
    for img, (w_, h_) in zip(out, ((t_res, t_res), (m_res, m_res), (sky_w, sky_h))):
        img[(h_ // 8) * 8:, :] = 0
        img[:, (w_ // 8) * 8:] = 0
    return tuple(out)


# ------------------------------------------------------------------ optional TAA stage
def gpu_color_to_luminance(be, color_packed, w, h):
    src = be.createImage(image_desc_2d(w, h, F.R11G11B10_uFloat), color_packed)
    dst = be.createImage(image_desc_2d(w, h, F.R8))
    p = be.createComputePass("colorToLuminance.comp", [], "Color to Luminance")
    be.newFrame()
    be.setComputePassExecution(ComputePassExecution(p, RenderPassResources(storageImages=[ImageResource(dst, 0, 1)], sampledImages=[ImageResource(src, 0, 0)]), b"",
                                                    (div_up(w, 8), div_up(h, 8), 1)))
    be.prepareForDrawcallRecording()
    be.renderFrame()
    return be.downloadImage(dst, 0, np.uint8).reshape(h, w).copy()


def orc_color_to_luminance(color_packed, w, h):
    src = orc.Img(np.ascontiguousarray(color_packed, np.uint32), w, h, F.R11G11B10_uFloat)
    dst = orc.new_image(w, h, F.R8, 1)
    orc.lib().orc_color_to_luminance(src.ref(), dst.ref())
    return dst.arr.reshape(h, w).copy()


def gpu_temporal_supersampling(be, current, last, motion_snorm, depth_cur, depth_last, lum_cur, lum_last, w, h, global_packed, tonemap=True):
    """TAA::computeTemporalSuperSampling bindings (TAA.cpp:106-136)"""
    mk = lambda fmt, data: be.createImage(image_desc_2d(w, h, fmt), data)
    cur, lst, tgt = mk(F.R11G11B10_uFloat, current), mk(F.R11G11B10_uFloat, last), mk(F.R11G11B10_uFloat, None)
    vel, dc, dl = mk(F.RG16_sNorm, motion_snorm), mk(F.Depth32, depth_cur), mk(F.Depth32, depth_last)
    lc, ll = mk(F.R8, lum_cur), mk(F.R8, lum_last)
    p = be.createComputePass("temporalSupersampling.comp", [spec_bool(0, tonemap)], "Temporal supersampling")
    gb = global_binding(be)
    gb.set(global_packed)
    be.newFrame()
    be.setComputePassExecution(ComputePassExecution(p, RenderPassResources(
        storageImages=[ImageResource(tgt, 0, 3)],
        sampledImages=[ImageResource(cur, 0, 1), ImageResource(lst, 0, 2), ImageResource(vel, 0, 4), ImageResource(dc, 0, 5), ImageResource(dl, 0, 6), ImageResource(lc, 0, 7),
                       ImageResource(ll, 0, 8)]), b"", (div_up(w, 8), div_up(h, 8), 1)))
    be.prepareForDrawcallRecording()
    be.renderFrame()
    return be.downloadImage(tgt, 0, np.uint32).reshape(h, w).copy()


def orc_temporal_supersampling(current, last, motion_snorm, depth_cur, depth_last, lum_cur, lum_last, w, h, global_packed, tonemap=True):
    I = lambda a, dt, fmt: orc.Img(np.ascontiguousarray(a, dt), w, h, fmt)
    cur, lst = I(current, np.uint32, F.R11G11B10_uFloat), I(last, np.uint32, F.R11G11B10_uFloat)
    tgt = orc.new_image(w, h, F.R11G11B10_uFloat, 4)
    vel, dc, dl = I(motion_snorm, np.int16, F.RG16_sNorm), I(depth_cur, np.float32, F.Depth32), I(depth_last, np.float32, F.Depth32)
    lc, ll = I(lum_cur, np.uint8, F.R8), I(lum_last, np.uint8, F.R8)
    g = orc.global_from_bytes(global_packed)
    orc.lib().orc_temporal_supersampling(cur.ref(), lst.ref(), tgt.ref(), vel.ref(), dc.ref(), dl.ref(), lc.ref(), ll.ref(), C.byref(g), C.c_int32(int(tonemap)))
    return tgt.arr.view(np.uint32).reshape(h, w).copy()


def gpu_sdf_debug(be, w, h, sky_packed, sky_w, sky_h, light_bytes, instance_bytes, tiles_u32, shadow_info, shadow_map_u16, shadow_res, global_packed, mode, cascade=2):
    """SDFGI::renderSDFVisualization's visualisation pass, Techniques/SDFGI.cpp:351-369 (the culling passes before it are gpu_sdf_culling)"""
    global_binding(be).set(global_packed)
    out = be.createImage(image_desc_2d(w, h, F.R11G11B10_uFloat))
    sky = be.createImage(image_desc_2d(sky_w, sky_h, F.R11G11B10_uFloat), sky_packed)
    shadow = be.createImage(image_desc_2d(shadow_res, shadow_res, F.Depth16), np.ascontiguousarray(shadow_map_u16))
    light = be.createStorageBuffer(20, light_bytes)
    inst = be.createStorageBuffer(len(instance_bytes), instance_bytes)
    tiles = be.createStorageBuffer(tiles_u32.nbytes, tiles_u32.tobytes())
    culled = be.createStorageBuffer(16)
    sinfo = be.createStorageBuffer(304, shadow_info)
    p = be.createComputePass("sdfDebugVisualisation.comp", [spec_int(0, mode), spec_int(1, cascade)], "Visualize SDF")
    be.newFrame()
    be.setComputePassExecution(ComputePassExecution(p, RenderPassResources(
        storageImages=[ImageResource(out, 0, 0)], sampledImages=[ImageResource(sky, 0, 2), ImageResource(shadow, 0, 7)],
        storageBuffers=[StorageBufferResource(light, True, 1), StorageBufferResource(inst, True, 3), StorageBufferResource(tiles, True, 4), StorageBufferResource(culled, True, 5),
                        StorageBufferResource(sinfo, True, 6)]), b"", (math.ceil(w / 8.0), math.ceil(h / 8.0), 1)))
    be.renderFrame()
    return be.downloadImage(out, 0, np.uint32).reshape(h, w).copy()


def orc_sdf_debug(w, h, sky_packed, sky_w, sky_h, light_bytes, instance_bytes, tiles_u32, shadow_info, shadow_map_u16, shadow_res, global_packed, bindless_arr, n_bindless,
                  mode, cascade=2):
    out = orc.new_image(w, h, F.R11G11B10_uFloat, 4)
    sky = orc.Img(sky_packed, sky_w, sky_h, F.R11G11B10_uFloat)
    shadow = orc.Img(np.ascontiguousarray(shadow_map_u16), shadow_res, shadow_res, F.Depth16)
    light = C.create_string_buffer(light_bytes, 20)
    inst = C.create_string_buffer(instance_bytes[16:], len(instance_bytes) - 16)
    tiles = np.ascontiguousarray(tiles_u32, np.uint32)
    sinfo = C.create_string_buffer(shadow_info, 304)
    g = orc.global_from_bytes(global_packed)
    orc.lib().orc_sdf_debug_visualisation(out.ref(), light, sky.ref(), inst, _p(tiles), sinfo, shadow.ref(), bindless_arr, C.c_int32(n_bindless), C.byref(g), C.c_int32(mode),
                                          C.c_int32(cascade))
    return out.arr.view(np.uint32).reshape(h, w).copy()


VOLUMETRIC_SETTINGS_DEFAULT = struct.pack("<13f", 0.3, -0.2, 0.1, 0.125, 1.0, 1.0, 1.0, 30.0, 1.0, 0.003, 0.008, 0.5, 0.2)  # state + VolumetricsSettings (Volumetrics.h:5-13)


def _desc3d(w, h, d, fmt):
    from plainrenderer_amd.backend import ImageDescription, ImageType
    return ImageDescription(width=w, height=h, depth=d, type=ImageType.Type3D, format=fmt, usageFlags=3, mipCount=MipCount.One)


def gpu_volumetrics(be, fw, fh, fd, noise_u8, history_u16, shadow_map_u16, shadow_res, shadow_info, light_bytes, settings_bytes, global_packed, intermediates=True):
    """Volumetrics::computeVolumetricLighting (Techniques/Volumetrics.cpp:119-243): material -> scattering -> reprojection -> integration.
    intermediates=False: the material / scattering volumes are not downloaded (None in their place) - with pass fusion level 2 the fused per-froxel
    launch does not write them"""
    global_binding(be).set(global_packed)
    mk = lambda data=None: be.createImage(_desc3d(fw, fh, fd, F.RGBA16_sFloat), data)
    material, scattering, target, history, integration = mk(), mk(), mk(), mk(np.ascontiguousarray(history_u16)), mk()
    n = noise_u8.shape[0]
    noise = be.createImage(_desc3d(n, n, n, F.R8), np.ascontiguousarray(noise_u8))
    shadow = be.createImage(image_desc_2d(shadow_res, shadow_res, F.Depth16), np.ascontiguousarray(shadow_map_u16))
    sinfo = be.createStorageBuffer(304, shadow_info)
    light = be.createStorageBuffer(20, light_bytes)
    ub = be.createUniformBuffer(64, settings_bytes)
    pm = be.createComputePass("froxelVolumeMaterial.comp", [], "Froxel volume material")
    ps = be.createComputePass("froxelLightScattering.comp", [], "Froxel light scattering")
    pr = be.createComputePass("volumeLightingReprojection.comp", [], "Volumetric lighting reprojection")
    pi = be.createComputePass("volumetricLightingIntegration.comp", [], "Volumetric light integration")
    g4 = (math.ceil(fw / 4.0), math.ceil(fh / 4.0), math.ceil(fd / 4.0))
    be.newFrame()
    be.setComputePassExecution(ComputePassExecution(pm, RenderPassResources(storageImages=[ImageResource(material, 0, 0)], sampledImages=[ImageResource(noise, 0, 1)],
                                                                            uniformBuffers=[UniformBufferResource(ub, 2)]), b"", g4))
    be.setComputePassExecution(ComputePassExecution(ps, RenderPassResources(storageImages=[ImageResource(scattering, 0, 0)],
                                                                            sampledImages=[ImageResource(shadow, 0, 1), ImageResource(material, 0, 2)],
                                                                            storageBuffers=[StorageBufferResource(sinfo, True, 3), StorageBufferResource(light, True, 4)],
                                                                            uniformBuffers=[UniformBufferResource(ub, 5)]), b"", g4))
    be.setComputePassExecution(ComputePassExecution(pr, RenderPassResources(storageImages=[ImageResource(target, 0, 0)],
                                                                            sampledImages=[ImageResource(scattering, 0, 1), ImageResource(history, 0, 2)],
                                                                            uniformBuffers=[UniformBufferResource(ub, 3)]), b"", g4))
    be.setComputePassExecution(ComputePassExecution(pi, RenderPassResources(storageImages=[ImageResource(integration, 0, 0)], sampledImages=[ImageResource(target, 0, 1)],
                                                                            uniformBuffers=[UniformBufferResource(ub, 2)]), b"", (math.ceil(fw / 8.0), math.ceil(fh / 8.0), 1)))
    be.prepareForDrawcallRecording()
    be.renderFrame()
    return [be.downloadImage(i, 0, np.uint16).copy() if (intermediates or i in (target, integration)) else None for i in (material, scattering, target, integration)]


def orc_volumetrics(fw, fh, fd, noise_u8, history_u16, shadow_map_u16, shadow_res, shadow_info, light_bytes, settings_bytes, global_packed):
    L = orc.lib()
    mk = lambda: orc.new_image(fw, fh, F.RGBA16_sFloat, 8, d=fd)
    material, scattering, target, integration = mk(), mk(), mk(), mk()
    history = orc.Img(np.ascontiguousarray(history_u16), fw, fh, F.RGBA16_sFloat, d=fd)
    n = noise_u8.shape[0]
    noise = orc.Img(np.ascontiguousarray(noise_u8), n, n, F.R8, d=n)
    shadow = orc.Img(np.ascontiguousarray(shadow_map_u16), shadow_res, shadow_res, F.Depth16)
    sinfo = C.create_string_buffer(shadow_info, 304)
    light = C.create_string_buffer(light_bytes, 20)
    st = C.create_string_buffer(settings_bytes[:52], 52)
    g = orc.global_from_bytes(global_packed)
    L.orc_froxel_volume_material(material.ref(), noise.ref(), st, C.byref(g))
    L.orc_froxel_light_scattering(scattering.ref(), shadow.ref(), material.ref(), sinfo, light, st, C.byref(g))
    L.orc_volume_lighting_reprojection(target.ref(), scattering.ref(), history.ref(), st, C.byref(g))
    L.orc_volumetric_lighting_integration(integration.ref(), target.ref(), st)
    return [i.arr.view(np.uint16).copy() for i in (material, scattering, target, integration)]


# ------------------------------------------------------------------ decision signatures (include/plr.h, oracle/oracle.h)
class gpu_signature:
    """with gpu_signature(be, words) as s: <one gpu_* pass>; s.words -> uint32 array written by the fast kernel of that pass"""

    def __init__(self, be, words):
        self.be, self.n = be, int(words)

    def __enter__(self):
        self.be.setDecisionSignature(self.n)
        return self

    def __exit__(self, *exc):
        if exc[0] is None:
            self.words = self.be.readDecisionSignature(self.n)
        self.be.setDecisionSignature(0)
        return False


orc_signature = orc.decision_signature
