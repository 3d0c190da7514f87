"""plainrenderer_amd: MI355X-native (HIP, gfx950) backend for PlainRenderer's per-pixel frame pipeline.

The product is the C-ABI shared library `libplr.so` (include/plr.h); this package holds its sources
(csrc/), the build script and a ctypes host-side mirror of the reference's RenderBackend interface.
"""
from .backend import (  # noqa: F401
    ComputePassExecution, ImageDescription, ImageFormat, ImageHandle, ImageResource, ImageType, ImageUsageFlags, MipCount, PlrError,
    RenderBackend, RenderPassResources, SpecialisationConstant, StorageBufferResource, UniformBufferResource, spec_bool, spec_float,
    spec_int, spec_uint, supported_shaders,
)
