"""ctypes binding of the C-ABI in include/plr.h plus a thin host-side mirror of the reference's
RenderBackend interface (Plain/src/Runtime/Rendering/Backend/RenderBackend.h:36-110) and pass-record structs
(ResourceDescriptions.h:9-172), so tests and benchmarks read like the reference's frontend code.

There is no CPU fallback: if libplr.so is missing or no HIP device is present, construction fails loudly.
"""
import ctypes as C
import os
import struct
from dataclasses import dataclass, field
from enum import IntEnum
from typing import List, Optional, Sequence

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("PLR_LIB") or os.path.join(_HERE, "libplr.so")  # PLR_LIB: an experiment's build (plainrenderer_amd/build.py PLR_BUILD_TAG)


class PlrError(RuntimeError):
    pass


class ImageType(IntEnum):
    Type1D = 0
    Type2D = 1
    Type3D = 2
    TypeCube = 3


class MipCount(IntEnum):
    One = 0
    FullChain = 1
    Manual = 2
    FullChainAlreadyInData = 3


class ImageUsageFlags(IntEnum):
    Storage = 1
    Sampled = 2
    Attachment = 4


class ImageFormat(IntEnum):
    R8 = 0
    RG8 = 1
    RGBA8 = 2
    R16_sFloat = 3
    RG16_sFloat = 4
    RG32_sFloat = 5
    RG16_sNorm = 6
    RGBA16_sFloat = 7
    RGBA16_sNorm = 8
    RGBA32_sFloat = 9
    R11G11B10_uFloat = 10
    Depth16 = 11
    Depth32 = 12
    BC1 = 13
    BC3 = 14
    BC5 = 15
    BGRA8_uNorm = 16


FORMAT_BYTES = {
    ImageFormat.R8: 1, ImageFormat.RG8: 2, ImageFormat.RGBA8: 4, ImageFormat.R16_sFloat: 2, ImageFormat.RG16_sFloat: 4,
    ImageFormat.RG32_sFloat: 8, ImageFormat.RG16_sNorm: 4, ImageFormat.RGBA16_sFloat: 8, ImageFormat.RGBA16_sNorm: 8,
    ImageFormat.RGBA32_sFloat: 16, ImageFormat.R11G11B10_uFloat: 4, ImageFormat.Depth16: 2, ImageFormat.Depth32: 4,
    ImageFormat.BGRA8_uNorm: 4,
}


class _ImageHandle(C.Structure):
    _fields_ = [("type", C.c_uint32), ("index", C.c_uint32)]


class _ImageDesc(C.Structure):
    _fields_ = [("width", C.c_uint32), ("height", C.c_uint32), ("depth", C.c_uint32), ("type", C.c_uint32),
                ("format", C.c_uint32), ("usage_flags", C.c_uint32), ("mip_count", C.c_uint32),
                ("manual_mip_count", C.c_uint32), ("auto_create_mips", C.c_uint32)]


class _SamplerDesc(C.Structure):
    _fields_ = [("interpolation", C.c_uint32), ("wrapping", C.c_uint32), ("use_anisotropy", C.c_uint32),
                ("max_anisotropy", C.c_float), ("border_color", C.c_uint32), ("max_mip", C.c_uint32)]


class _ImageResource(C.Structure):
    _fields_ = [("image", _ImageHandle), ("mip_level", C.c_uint32), ("binding", C.c_uint32)]


class _StorageBufferResource(C.Structure):
    _fields_ = [("buffer", C.c_uint32), ("read_only", C.c_uint32), ("binding", C.c_uint32)]


class _UniformBufferResource(C.Structure):
    _fields_ = [("buffer", C.c_uint32), ("binding", C.c_uint32)]


class _SamplerResource(C.Structure):
    _fields_ = [("sampler", C.c_uint32), ("binding", C.c_uint32)]


class _PassResources(C.Structure):
    _fields_ = [("samplers", C.POINTER(_SamplerResource)), ("sampler_count", C.c_uint32),
                ("storage_buffers", C.POINTER(_StorageBufferResource)), ("storage_buffer_count", C.c_uint32),
                ("uniform_buffers", C.POINTER(_UniformBufferResource)), ("uniform_buffer_count", C.c_uint32),
                ("sampled_images", C.POINTER(_ImageResource)), ("sampled_image_count", C.c_uint32),
                ("storage_images", C.POINTER(_ImageResource)), ("storage_image_count", C.c_uint32)]


class _ComputePassExecution(C.Structure):
    _fields_ = [("handle", C.c_uint32), ("resources", _PassResources), ("push_constants", C.c_void_p),
                ("push_constant_size", C.c_uint32), ("dispatch_count", C.c_uint32 * 3), ("dispatch_base", C.c_uint32 * 3), ("valid_rows", C.c_uint32 * 2), ("async_tail", C.c_uint32), ("first_rows", C.c_uint32 * 2),
                ("valid_cols", C.c_uint32 * 2), ("first_cols", C.c_uint32 * 2)]


class _SpecConstant(C.Structure):
    _fields_ = [("location", C.c_uint32), ("data", C.c_void_p), ("size", C.c_uint32)]


class _ComputePassDesc(C.Structure):
    _fields_ = [("src_path_relative", C.c_char_p), ("specialisation_constants", C.POINTER(_SpecConstant)),
                ("specialisation_constant_count", C.c_uint32), ("name", C.c_char_p)]


class _RenderPassTime(C.Structure):
    _fields_ = [("time_ms", C.c_float), ("name", C.c_char_p)]


# ---- host-side mirror of the reference's plain-data structs (same names, same meaning) ----
@dataclass(frozen=True)
class ImageHandle:
    type: int = 0
    index: int = 0xFFFFFFFF


@dataclass
class ImageDescription:
    width: int = 1
    height: int = 0
    depth: int = 0
    type: ImageType = ImageType.Type1D
    format: ImageFormat = ImageFormat.R8
    usageFlags: int = 0
    mipCount: MipCount = MipCount.One
    manualMipCount: int = 1
    autoCreateMips: bool = False


@dataclass
class ImageResource:
    image: ImageHandle
    mipLevel: int
    binding: int


@dataclass
class StorageBufferResource:
    buffer: int
    readOnly: bool
    binding: int


@dataclass
class UniformBufferResource:
    buffer: int
    binding: int


@dataclass
class RenderPassResources:
    samplers: list = field(default_factory=list)
    storageBuffers: List[StorageBufferResource] = field(default_factory=list)
    uniformBuffers: List[UniformBufferResource] = field(default_factory=list)
    sampledImages: List[ImageResource] = field(default_factory=list)
    storageImages: List[ImageResource] = field(default_factory=list)


@dataclass
class ComputePassExecution:
    handle: int = 0xFFFFFFFF
    resources: RenderPassResources = field(default_factory=RenderPassResources)
    pushConstants: bytes = b""
    dispatchCount: Sequence[int] = (1, 1, 1)


@dataclass
class SpecialisationConstant:
    location: int
    data: bytes  # raw bytes with the C++ sizeof: bool -> 1 byte, int/uint/float -> 4 bytes


def spec_bool(location, v):
    return SpecialisationConstant(location, struct.pack("<?", bool(v)))


def spec_int(location, v):
    return SpecialisationConstant(location, struct.pack("<i", int(v)))


def spec_uint(location, v):
    return SpecialisationConstant(location, struct.pack("<I", int(v)))


def spec_float(location, v):
    return SpecialisationConstant(location, struct.pack("<f", float(v)))


def _load():
    if not os.path.exists(LIB_PATH):
        raise PlrError("HIP backend library missing: %s (run `python -c 'import __graft_entry__ as g; g.build()'`). "
                       "There is no CPU fallback." % LIB_PATH)
    lib = C.CDLL(LIB_PATH)
    lib.plr_last_error.restype = C.c_char_p
    return lib


EXPORTED_SYMBOLS = [
    "plr_debug_pcf_tap_table", "plr_setup", "plr_shutdown", "plr_recreate_swapchain", "plr_last_error", "plr_wait_for_gpu_idle", "plr_update_shader_code",
    "plr_resize_images", "plr_new_frame", "plr_set_compute_pass_execution", "plr_prepare_for_drawcall_recording",
    "plr_set_uniform_buffer_data", "plr_set_storage_buffer_data", "plr_set_global_descriptor_set_resources",
    "plr_update_compute_pass_shader_description", "plr_render_frame", "plr_get_image_global_texture_array_index",
    "plr_create_compute_pass", "plr_create_image", "plr_create_uniform_buffer", "plr_create_storage_buffer", "plr_create_sampler",
    "plr_create_temporary_image", "plr_get_swapchain_input_image", "plr_get_memory_stats", "plr_get_renderpass_timings",
    "plr_get_last_frame_cpu_time", "plr_get_image_description", "plr_set_pass_timing", "plr_get_last_frame_gpu_time",
    "plr_replay_frame", "plr_upload_image", "plr_download_image", "plr_download_storage_buffer", "plr_download_uniform_buffer",
    "plr_get_image_device_pointer", "plr_get_storage_buffer_device_pointer", "plr_get_stream", "plr_get_launch_stream", "plr_copy_device_memory", "plr_read_device_memory", "plr_write_device_memory", "plr_get_supported_shaders",
    "plr_debug_math_eval", "plr_debug_codec_eval", "plr_debug_sampler_eval", "plr_debug_sky_lut_eval", "plr_debug_verify_histogram_thresholds", "plr_debug_verify_r11g11b10_fast", "plr_debug_set_decision_signature", "plr_debug_read_decision_signature", "plr_set_math_mode", "plr_get_math_mode", "plr_set_stream_overlap", "plr_get_stream_overlap", "plr_set_pass_fusion", "plr_get_pass_fusion", "plr_set_pass_fusion_reorder", "plr_get_general_kernel_executions", "plr_get_edge_signal", "plr_set_async_tail", "plr_get_async_tail", "plr_set_early_parts", "plr_get_early_parts", "plr_set_host_callback_execution", "plr_set_host_callback_execution_on", "plr_upload_image_rows",
    "plr_copy_device_memory_2d", "plr_set_global_descriptor_set_layout",
]


class RenderBackend:
    """Mirror of the reference's RenderBackend public API (compute subset) over the C-ABI."""

    def __init__(self, width, height, device=0):
        self.lib = _load()
        self._open = False
        self._check(self.lib.plr_setup(C.c_int(device), C.c_uint32(width), C.c_uint32(height)))
        self._open = True
        self.width, self.height = width, height

    # -- plumbing
    def _check(self, rc):
        if rc != 0:
            raise PlrError("plr error %d: %s" % (rc, self.lib.plr_last_error().decode()))

    def shutdown(self):
        if self._open:
            self.lib.plr_shutdown()
            self._open = False

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.shutdown()

    @staticmethod
    def _h(handle: ImageHandle):
        return _ImageHandle(handle.type, handle.index)

    @staticmethod
    def _desc(d: ImageDescription):
        return _ImageDesc(d.width, d.height, d.depth, int(d.type), int(d.format), int(d.usageFlags), int(d.mipCount),
                          d.manualMipCount, int(d.autoCreateMips))

    # -- reference API
    def waitForGPUIdle(self):
        self._check(self.lib.plr_wait_for_gpu_idle())

    def resizeImages(self, images, width, height):
        arr = (_ImageHandle * len(images))(*[self._h(i) for i in images])
        self._check(self.lib.plr_resize_images(arr, C.c_uint32(len(images)), C.c_uint32(width), C.c_uint32(height)))

    def newFrame(self):
        self._check(self.lib.plr_new_frame())

    def setComputePassExecution(self, exe: ComputePassExecution):
        r = exe.resources
        keep = []

        def arr(ctype, items):
            a = (ctype * max(len(items), 1))(*items)
            keep.append(a)
            return a

        sb = arr(_StorageBufferResource, [_StorageBufferResource(b.buffer, int(b.readOnly), b.binding) for b in r.storageBuffers])
        ub = arr(_UniformBufferResource, [_UniformBufferResource(b.buffer, b.binding) for b in r.uniformBuffers])
        si = arr(_ImageResource, [_ImageResource(self._h(i.image), i.mipLevel, i.binding) for i in r.sampledImages])
        st = arr(_ImageResource, [_ImageResource(self._h(i.image), i.mipLevel, i.binding) for i in r.storageImages])
        sm = arr(_SamplerResource, [])
        e = _ComputePassExecution()
        e.handle = exe.handle
        e.resources = _PassResources(sm, 0, sb, len(r.storageBuffers), ub, len(r.uniformBuffers), si, len(r.sampledImages), st,
                                     len(r.storageImages))
        pc = bytes(exe.pushConstants)
        buf = C.create_string_buffer(pc, len(pc)) if pc else None
        e.push_constants = C.cast(buf, C.c_void_p) if buf is not None else None
        e.push_constant_size = len(pc)
        e.dispatch_count = (C.c_uint32 * 3)(*[int(x) for x in exe.dispatchCount])
        e.dispatch_base = (C.c_uint32 * 3)(*[int(x) for x in getattr(exe, "dispatchBase", (0, 0, 0))])
        e.valid_rows = (C.c_uint32 * 2)(*[int(x) for x in getattr(exe, "validRows", (0, 0))])
        e.async_tail = int(bool(getattr(exe, "asyncTail", False)))
        e.first_rows = (C.c_uint32 * 2)(*[int(x) for x in getattr(exe, "firstRows", (0, 0))])
        e.valid_cols = (C.c_uint32 * 2)(*[int(x) for x in getattr(exe, "validCols", (0, 0))])
        e.first_cols = (C.c_uint32 * 2)(*[int(x) for x in getattr(exe, "firstCols", (0, 0))])
        self._check(self.lib.plr_set_compute_pass_execution(C.byref(e)))

    def prepareForDrawcallRecording(self):
        self._check(self.lib.plr_prepare_for_drawcall_recording())

    def setUniformBufferData(self, buffer, data):
        b = bytes(data)
        self._check(self.lib.plr_set_uniform_buffer_data(C.c_uint32(buffer), b, C.c_size_t(len(b))))

    def setStorageBufferData(self, buffer, data):
        b = bytes(data)
        self._check(self.lib.plr_set_storage_buffer_data(C.c_uint32(buffer), b, C.c_size_t(len(b))))

    def setGlobalDescriptorSetResources(self, resources: RenderPassResources):
        ub = (_UniformBufferResource * max(len(resources.uniformBuffers), 1))(
            *[_UniformBufferResource(b.buffer, b.binding) for b in resources.uniformBuffers])
        r = _PassResources(None, 0, None, 0, ub, len(resources.uniformBuffers), None, 0, None, 0)
        self._check(self.lib.plr_set_global_descriptor_set_resources(C.byref(r)))

    def _pass_desc(self, shader, spec, name):
        keep = []
        sc = (_SpecConstant * max(len(spec), 1))()
        for i, s in enumerate(spec):
            buf = C.create_string_buffer(bytes(s.data), len(s.data))
            keep.append(buf)
            sc[i] = _SpecConstant(s.location, C.cast(buf, C.c_void_p), len(s.data))
        d = _ComputePassDesc(shader.encode(), sc, len(spec), name.encode() if name is not None else None)
        keep.append(sc)
        return d, keep

    def createComputePass(self, srcPathRelative, specialisationConstants=(), name=None):
        d, keep = self._pass_desc(srcPathRelative, list(specialisationConstants), name or srcPathRelative)
        out = C.c_uint32()
        self._check(self.lib.plr_create_compute_pass(C.byref(d), C.byref(out)))
        return out.value

    def updateComputePassShaderDescription(self, passHandle, srcPathRelative, specialisationConstants=()):
        d, keep = self._pass_desc(srcPathRelative, list(specialisationConstants), None)
        self._check(self.lib.plr_update_compute_pass_shader_description(C.c_uint32(passHandle), C.byref(d)))

    def renderFrame(self, presentToScreen=False):
        self._check(self.lib.plr_render_frame(C.c_int(int(presentToScreen))))

    def getImageGlobalTextureArrayIndex(self, image):
        out = C.c_uint32()
        self._check(self.lib.plr_get_image_global_texture_array_index(self._h(image), C.byref(out)))
        return out.value

    def createImage(self, description: ImageDescription, initialData=None):
        out = _ImageHandle()
        d = self._desc(description)
        if initialData is not None:
            b = np.ascontiguousarray(initialData)
            self._check(self.lib.plr_create_image(C.byref(d), b.ctypes.data_as(C.c_void_p), C.c_size_t(b.nbytes), C.byref(out)))
        else:
            self._check(self.lib.plr_create_image(C.byref(d), None, C.c_size_t(0), C.byref(out)))
        return ImageHandle(out.type, out.index)

    def createUniformBuffer(self, size, initialData=None):
        out = C.c_uint32()
        b = bytes(initialData) if initialData is not None else None
        self._check(self.lib.plr_create_uniform_buffer(C.c_size_t(size), b, C.byref(out)))
        return out.value

    def createStorageBuffer(self, size, initialData=None):
        out = C.c_uint32()
        b = bytes(initialData) if initialData is not None else None
        self._check(self.lib.plr_create_storage_buffer(C.c_size_t(size), b, C.byref(out)))
        return out.value

    def createTemporaryImage(self, description: ImageDescription):
        out = _ImageHandle()
        d = self._desc(description)
        self._check(self.lib.plr_create_temporary_image(C.byref(d), C.byref(out)))
        return ImageHandle(out.type, out.index)

    def getSwapchainInputImage(self):
        out = _ImageHandle()
        self._check(self.lib.plr_get_swapchain_input_image(C.byref(out)))
        return ImageHandle(out.type, out.index)

    def getMemoryStats(self):
        a, u = C.c_uint64(), C.c_uint64()
        self._check(self.lib.plr_get_memory_stats(C.byref(a), C.byref(u)))
        return a.value, u.value

    def getRenderpassTimings(self):
        n = C.c_uint32(0)
        self._check(self.lib.plr_get_renderpass_timings(None, C.byref(n)))
        if n.value == 0:
            return []
        arr = (_RenderPassTime * n.value)()
        self._check(self.lib.plr_get_renderpass_timings(arr, C.byref(n)))
        return [(arr[i].name.decode(), arr[i].time_ms) for i in range(n.value)]

    def getImageDescription(self, image):
        d = _ImageDesc()
        self._check(self.lib.plr_get_image_description(self._h(image), C.byref(d)))
        return ImageDescription(d.width, d.height, d.depth, ImageType(d.type), ImageFormat(d.format), d.usage_flags, MipCount(d.mip_count),
                                d.manual_mip_count, bool(d.auto_create_mips))

    # -- additions (tests / benchmarks)
    def setStreamOverlap(self, enabled):
        self._check(self.lib.plr_set_stream_overlap(C.c_int(1 if enabled else 0)))

    def getStreamOverlap(self):
        """(enabled, executions of the last frame that ran on a side stream)"""
        en, n = C.c_int(0), C.c_uint32(0)
        self._check(self.lib.plr_get_stream_overlap(C.byref(en), C.byref(n)))
        return bool(en.value), int(n.value)

    def setMathMode(self, fast):
        self._check(self.lib.plr_set_math_mode(C.c_int(1 if fast else 0)))

    def getMathMode(self):
        m = C.c_int()
        self._check(self.lib.plr_get_math_mode(C.byref(m)))
        return m.value

    def setPassTiming(self, enabled):
        self._check(self.lib.plr_set_pass_timing(C.c_int(int(enabled))))

    def getLastFrameGpuTime(self):
        ms = C.c_float()
        self._check(self.lib.plr_get_last_frame_gpu_time(C.byref(ms)))
        return ms.value

    def replayFrame(self, count):
        ms = C.c_float()
        self._check(self.lib.plr_replay_frame(C.c_uint32(count), C.byref(ms)))
        return ms.value

    def mipSize(self, image, mip=0):
        d = self.getImageDescription(image)
        w = max(d.width >> mip, 1)
        h = max(max(d.height, 1) >> mip, 1)
        dep = max(max(d.depth, 1) >> mip, 1)
        return w, h, dep, FORMAT_BYTES[d.format]

    def uploadImage(self, image, data, mip=0):
        b = np.ascontiguousarray(data)
        self._check(self.lib.plr_upload_image(self._h(image), C.c_uint32(mip), b.ctypes.data_as(C.c_void_p), C.c_size_t(b.nbytes)))

    def recreateSwapchain(self, width, height):
        """RenderBackend::recreateSwapchain (RenderBackend.h:38)"""
        self._check(self.lib.plr_recreate_swapchain(C.c_uint32(width), C.c_uint32(height)))
        self.width, self.height = width, height

    def getLastFrameCPUTime(self):
        """RenderBackend::getLastFrameCPUTime: host milliseconds spent inside the last renderFrame (flush + launches)"""
        ms = C.c_float()
        self._check(self.lib.plr_get_last_frame_cpu_time(C.byref(ms)))
        return ms.value

    def uploadImageRows(self, image, row_begin, data, mip=0):
        """data: the rows [row_begin, row_begin + data.shape[0]) of a 2D image"""
        b = np.ascontiguousarray(data)
        self._check(self.lib.plr_upload_image_rows(self._h(image), C.c_uint32(mip), C.c_uint32(row_begin), C.c_uint32(b.shape[0]), b.ctypes.data_as(C.c_void_p),
                                                   C.c_size_t(b.nbytes)))

    def downloadImage(self, image, mip=0, dtype=np.uint8):
        w, h, dep, bpp = self.mipSize(image, mip)
        out = np.empty(w * h * dep * bpp, np.uint8)
        self._check(self.lib.plr_download_image(self._h(image), C.c_uint32(mip), out.ctypes.data_as(C.c_void_p), C.c_size_t(out.nbytes)))
        return out.view(dtype)

    def downloadStorageBuffer(self, buffer, size, offset=0, dtype=np.uint8):
        out = np.empty(size, np.uint8)
        self._check(self.lib.plr_download_storage_buffer(C.c_uint32(buffer), out.ctypes.data_as(C.c_void_p), C.c_size_t(offset), C.c_size_t(size)))
        return out.view(dtype)

    def downloadUniformBuffer(self, buffer, size, offset=0, dtype=np.uint8):
        out = np.empty(size, np.uint8)
        self._check(self.lib.plr_download_uniform_buffer(C.c_uint32(buffer), out.ctypes.data_as(C.c_void_p), C.c_size_t(offset), C.c_size_t(size)))
        return out.view(dtype)

    def imageDevicePointer(self, image, mip=0):
        p, s = C.c_void_p(), C.c_size_t()
        self._check(self.lib.plr_get_image_device_pointer(self._h(image), C.c_uint32(mip), C.byref(p), C.byref(s)))
        return p.value, s.value

    def storageBufferDevicePointer(self, buffer):
        p, s = C.c_void_p(), C.c_size_t()
        self._check(self.lib.plr_get_storage_buffer_device_pointer(C.c_uint32(buffer), C.byref(p), C.byref(s)))
        return p.value, s.value

    def getStream(self):
        p = C.c_void_p()
        self._check(self.lib.plr_get_stream(C.byref(p)))
        return p.value

    def debugMathEval(self, fn, a, b=None):
        a = np.ascontiguousarray(a, np.float32)
        out = np.empty_like(a)
        bp = None
        if b is not None:
            b = np.ascontiguousarray(b, np.float32)
            bp = b.ctypes.data_as(C.c_void_p)
        self._check(self.lib.plr_debug_math_eval(C.c_int(fn), a.ctypes.data_as(C.c_void_p), bp, out.ctypes.data_as(C.c_void_p), C.c_int64(a.size)))
        return out

    def debugSkyLutEval(self, sky_lut, directions):
        d = np.ascontiguousarray(directions, np.float32).reshape(-1, 3)
        out = np.empty_like(d)
        self._check(self.lib.plr_debug_sky_lut_eval(_ImageHandle(sky_lut.type, sky_lut.index), d.ctypes.data_as(C.c_void_p), out.ctypes.data_as(C.c_void_p), C.c_int64(d.shape[0])))
        return out

    def debugPcfTapTable(self):
        """-> (256, 12, 2) float32: the PCF tap table of the PLR_MATH_FAST deferred shade (noise byte, tap) -> unit-disc offset"""
        out = np.empty((256, 12, 2), np.float32)
        self._check(self.lib.plr_debug_pcf_tap_table(out.ctypes.data_as(C.c_void_p), C.c_size_t(out.size)))
        return out

    def debugVerifyHistogramThresholds(self, min_luminance, max_luminance):
        """-> number of float bit patterns (of all 2^32) whose table-driven histogram bin differs from the shader's formula"""
        n = C.c_uint64()
        self._check(self.lib.plr_debug_verify_histogram_thresholds(C.c_float(min_luminance), C.c_float(max_luminance), C.byref(n)))
        return n.value

    def debugVerifyR11G11B10Fast(self):
        """-> (patterns whose 11-bit code differs, whose 10-bit code differs, largest code difference, largest differing input bit pattern) of the
        PLR_MATH_FAST R11G11B10 encoder against the exact one over all 2^32 float bit patterns"""
        out = (C.c_uint64 * 4)()
        self._check(self.lib.plr_debug_verify_r11g11b10_fast(out))
        return tuple(int(v) for v in out)

    def setPassFusion(self, level):
        """0 / False: off; 1 / True: on, every intermediate image written; 2 (the default): on, and a fused launch may keep an intermediate that
        nothing else reads in registers instead of writing it (include/plr.h)"""
        self._check(self.lib.plr_set_pass_fusion(C.c_int(int(level))))

    def setPassFusionReorder(self, enabled):
        """executions recorded between the members of a fusable group are moved out of its way when their resources allow it (include/plr.h); default on"""
        self._check(self.lib.plr_set_pass_fusion_reorder(C.c_int(int(bool(enabled)))))

    def getPassFusion(self):
        """-> (level, executions of the last frame that ran inside a fused launch)"""
        e, n = C.c_int(), C.c_uint32()
        self._check(self.lib.plr_get_pass_fusion(C.byref(e), C.byref(n)))
        return e.value, n.value

    def getEdgeSignal(self):
        """-> (device address of the edge signal word or None, value the last rows-first execution raises it to, value it holds now)"""
        p, v = C.c_void_p(), C.c_uint32()
        self._check(self.lib.plr_get_edge_signal(C.byref(p), C.byref(v)))
        if not p.value:
            return None, v.value, None
        now = C.c_uint32()
        self._check(self.lib.plr_read_device_memory(C.byref(now), p, C.c_size_t(4)))
        return p.value, v.value, now.value

    def getGeneralKernelExecutions(self):
        """-> (count, names): executions of the last frame that ran the general (exact-order) kernel although the math mode is fast"""
        n = C.c_uint32()
        buf = C.create_string_buffer(2048)
        self._check(self.lib.plr_get_general_kernel_executions(C.byref(n), buf, C.c_size_t(len(buf))))
        return n.value, buf.value.decode()

    def setAsyncTail(self, enabled):
        """executions flagged async_tail (the bloom chain + tonemap of the C++ FramePipeline) run on a second stream beside the next frame (include/plr.h)"""
        self._check(self.lib.plr_set_async_tail(C.c_int(int(bool(enabled)))))

    def getAsyncTail(self):
        """-> (enabled, executions of the last frame launched on the tail stream)"""
        e, n = C.c_int(), C.c_uint32()
        self._check(self.lib.plr_get_async_tail(C.byref(e), C.byref(n)))
        return bool(e.value), n.value

    def setEarlyParts(self, enabled):
        """early parts of fused launches (the deferred shade's direct lighting as a launch of its own beside the GI chain; include/plr.h)"""
        self._check(self.lib.plr_set_early_parts(C.c_int(int(enabled))))

    def getEarlyParts(self):
        """-> (level: 0 off, 1 on, 2 forced; early parts launched by the last frame)"""
        e, n = C.c_int(), C.c_uint32()
        self._check(self.lib.plr_get_early_parts(C.byref(e), C.byref(n)))
        return e.value, n.value

    def setDecisionSignature(self, words):
        """decision-signature buffer for the next single-pass frame (include/plr.h); 0 frees it"""
        self._check(self.lib.plr_debug_set_decision_signature(C.c_size_t(int(words))))

    def readDecisionSignature(self, words):
        out = np.empty(int(words), np.uint32)
        self._check(self.lib.plr_debug_read_decision_signature(out.ctypes.data_as(C.c_void_p), C.c_size_t(int(words))))
        return out

    def debugSamplerEval(self, image, filter, address, coords, mip=0):
        """-> (n, 4) float32: the device sampler (0 nearest / 1 linear / 2 gather; 0 clamp / 1 repeat / 2 border white / 3 border black) at coords (n, 2|3)"""
        coords = np.ascontiguousarray(coords, np.float32)
        n = coords.shape[0]
        out = np.empty((n, 4), np.float32)
        self._check(self.lib.plr_debug_sampler_eval(_ImageHandle(image.type, image.index), C.c_uint32(mip), C.c_int(filter), C.c_int(address),
                                                    coords.ctypes.data_as(C.c_void_p), out.ctypes.data_as(C.c_void_p), C.c_int64(n)))
        return out

    def debugCodecEval(self, fn, data, n, out_dtype, out_count):
        data = np.ascontiguousarray(data)
        out = np.empty(out_count, out_dtype)
        self._check(self.lib.plr_debug_codec_eval(C.c_int(fn), data.ctypes.data_as(C.c_void_p), out.ctypes.data_as(C.c_void_p), C.c_int64(n)))
        return out


def supported_shaders():
    lib = _load()
    arr = (C.c_char_p * 128)()
    n = lib.plr_get_supported_shaders(arr, C.c_uint32(128))
    return [arr[i].decode() for i in range(min(n, 128))]
