// C++ shim that re-creates the reference's RenderBackend interface (compute subset) on top of the C-ABI in plr.h.
//
// This is what a PlainRenderer maintainer drops in for Plain/src/Runtime/Rendering/Backend/RenderBackend.{h,cpp}: the
// struct and member names are the reference's (ResourceDescriptions.h:9-172, RenderHandles.h:4-41, Common/ImageDescription.h,
// RenderBackend.h:36-110), so recorder code written against `gRenderBackend` (Techniques/TAA.cpp, Bloom.cpp, SDFGI.cpp,
// RenderFrontend.cpp) compiles against it with only the graphics calls removed. Errors throw std::runtime_error, which is
// what the reference does on shader failure (RenderBackend.cpp:442-445).
#pragma once
#include <cstdint>
#include <cstring>
#include <limits>
#include <stdexcept>
#include <string>
#include <vector>

#include "plr.h"

namespace plrhost {

const uint32_t invalidIndex = std::numeric_limits<uint32_t>::max();

struct RenderPassHandle { uint32_t index = invalidIndex; };
enum class ImageHandleType : uint8_t { Default, Transient, Swapchain };
struct ImageHandle { ImageHandleType type = ImageHandleType::Default; uint32_t index = invalidIndex; };
struct SamplerHandle { uint32_t index = invalidIndex; };
struct UniformBufferHandle { uint32_t index = invalidIndex; };
struct StorageBufferHandle { uint32_t index = invalidIndex; };

enum class ImageType { Type1D, Type2D, Type3D, TypeCube };
enum class MipCount { One, FullChain, Manual, FullChainAlreadyInData };
enum class ImageUsageFlags : uint32_t { Storage = 0x00000001, Sampled = 0x00000002, Attachment = 0x00000004 };
inline ImageUsageFlags operator|(ImageUsageFlags l, ImageUsageFlags r) { return ImageUsageFlags(uint32_t(l) | uint32_t(r)); }
enum class ImageFormat { R8, RG8, RGBA8, R16_sFloat, RG16_sFloat, RG32_sFloat, RG16_sNorm, RGBA16_sFloat, RGBA16_sNorm, RGBA32_sFloat, R11G11B10_uFloat,
                         Depth16, Depth32, BC1, BC3, BC5, BGRA8_uNorm };

struct ImageDescription {
    uint32_t width = 1;
    uint32_t height = 0;
    uint32_t depth = 0;
    ImageType type = ImageType::Type1D;
    ImageFormat format = ImageFormat::R8;
    ImageUsageFlags usageFlags = (ImageUsageFlags)0;
    MipCount mipCount = MipCount::One;
    uint32_t manualMipCount = 1;
    bool autoCreateMips = false;
};

struct StorageBufferResource {
    StorageBufferResource(const StorageBufferHandle buffer, const bool readOnly, const uint32_t binding) : buffer(buffer), readOnly(readOnly), binding(binding) {}
    StorageBufferHandle buffer;
    bool readOnly;
    uint32_t binding;
};
struct UniformBufferResource {
    UniformBufferResource(const UniformBufferHandle buffer, const uint32_t binding) : buffer(buffer), binding(binding) {}
    UniformBufferHandle buffer;
    uint32_t binding;
};
struct ImageResource {
    ImageResource(const ImageHandle image, const uint32_t mipLevel, const uint32_t binding) : image(image), mipLevel(mipLevel), binding(binding) {}
    ImageHandle image;
    uint32_t mipLevel;
    uint32_t binding;
};
struct SamplerResource {
    SamplerResource(const SamplerHandle sampler, const uint32_t binding) : sampler(sampler), binding(binding) {}
    SamplerHandle sampler;
    uint32_t binding;
};
struct RenderPassResources {
    std::vector<SamplerResource> samplers;
    std::vector<StorageBufferResource> storageBuffers;
    std::vector<UniformBufferResource> uniformBuffers;
    std::vector<ImageResource> sampledImages;
    std::vector<ImageResource> storageImages;
};
struct RenderPassExecution {
    RenderPassHandle handle;
    RenderPassResources resources;
};
struct ComputePassExecution {
    RenderPassExecution genericInfo;
    std::vector<char> pushConstants;
    uint32_t dispatchCount[3] = {1, 1, 1};
    uint32_t dispatchBase[3] = {0, 0, 0}; // extension: first workgroup (band rendering), see plr.h
    uint32_t validRows[2] = {0, 0};       // extension: rows of the input images that hold valid data (band rendering), see plr.h
    bool asyncTail = false;               // extension: part of the frame's asynchronous tail (plr_compute_pass_execution::async_tail), see plr.h
    uint32_t firstRows[2] = {0, 0};       // extension: workgroup rows to produce first + edge signal (plr_compute_pass_execution::first_rows), see plr.h
    uint32_t validCols[2] = {0, 0};       // extension: columns of the input images that hold valid data (tile rendering, plr_compute_pass_execution::valid_cols)
    uint32_t firstCols[2] = {0, 0};       // extension: workgroup columns of the edge that is produced first (tile rendering, plr_compute_pass_execution::first_cols)
};
struct SpecialisationConstant {
    uint32_t location;
    std::vector<char> data;
};
struct ShaderDescription {
    std::string srcPathRelative;
    std::vector<SpecialisationConstant> specialisationConstants;
};
struct ComputePassDescription {
    ShaderDescription shaderDescription;
    std::string name;
};
// Backend/Resources.h:9-15: the bindings a descriptor set declares, by resource kind
struct ShaderLayout {
    std::vector<uint32_t> samplerBindings;
    std::vector<uint32_t> sampledImageBindings;
    std::vector<uint32_t> storageImageBindings;
    std::vector<uint32_t> uniformBufferBindings;
    std::vector<uint32_t> storageBufferBindings;
};
struct UniformBufferDescription { size_t size = 0; void* initialData = nullptr; };
struct StorageBufferDescription { size_t size = 0; void* initialData = nullptr; };
enum class SamplerInterpolation { Nearest, Linear };
enum class SamplerWrapping { Clamp, Color, Repeat };
enum class SamplerBorderColor { White, Black };
struct SamplerDescription {
    SamplerInterpolation interpolation = SamplerInterpolation::Nearest;
    SamplerWrapping wrapping = SamplerWrapping::Repeat;
    bool useAnisotropy = false;
    float maxAnisotropy = 8;
    SamplerBorderColor borderColor = SamplerBorderColor::Black;
    uint32_t maxMip = 1;
};
struct RenderPassTime { float timeMs = 0; std::string name; };

// Common/Utilities/GeneralUtils.cpp:8-12
inline std::vector<char> dataToCharArray(const void* data, const size_t size) {
    std::vector<char> result(size);
    std::memcpy(result.data(), data, size);
    return result;
}

class RenderBackend {
public:
    void setup(int deviceOrdinal, uint32_t width, uint32_t height) { check(plr_setup(deviceOrdinal, width, height)); }
    void shutdown() { plr_shutdown(); }
    void recreateSwapchain(const uint32_t width, const uint32_t height) { check(plr_recreate_swapchain(width, height)); }
    void waitForGPUIdle() { check(plr_wait_for_gpu_idle()); }
    void updateShaderCode() { check(plr_update_shader_code()); }

    void resizeImages(const std::vector<ImageHandle>& images, const uint32_t width, const uint32_t height) {
        std::vector<plr_image_handle> h;
        for (const auto& i : images) h.push_back(toC(i));
        check(plr_resize_images(h.data(), (uint32_t)h.size(), width, height));
    }
    void newFrame() { check(plr_new_frame()); }

    void setComputePassExecution(const ComputePassExecution& execution) {
        const RenderPassResources& r = execution.genericInfo.resources;
        std::vector<plr_storage_buffer_resource> sb;
        std::vector<plr_uniform_buffer_resource> ub;
        std::vector<plr_image_resource> si, st;
        for (const auto& b : r.storageBuffers) sb.push_back({b.buffer.index, b.readOnly ? 1u : 0u, b.binding});
        for (const auto& b : r.uniformBuffers) ub.push_back({b.buffer.index, b.binding});
        for (const auto& i : r.sampledImages) si.push_back({toC(i.image), i.mipLevel, i.binding});
        for (const auto& i : r.storageImages) st.push_back({toC(i.image), i.mipLevel, i.binding});
        plr_compute_pass_execution e{};
        e.handle = execution.genericInfo.handle.index;
        e.resources.storage_buffers = sb.data(); e.resources.storage_buffer_count = (uint32_t)sb.size();
        e.resources.uniform_buffers = ub.data(); e.resources.uniform_buffer_count = (uint32_t)ub.size();
        e.resources.sampled_images = si.data(); e.resources.sampled_image_count = (uint32_t)si.size();
        e.resources.storage_images = st.data(); e.resources.storage_image_count = (uint32_t)st.size();
        e.push_constants = execution.pushConstants.data();
        e.push_constant_size = (uint32_t)execution.pushConstants.size();
        for (int i = 0; i < 3; i++) { e.dispatch_count[i] = execution.dispatchCount[i]; e.dispatch_base[i] = execution.dispatchBase[i]; }
        e.valid_rows[0] = execution.validRows[0]; e.valid_rows[1] = execution.validRows[1];
        e.async_tail = execution.asyncTail ? 1u : 0u;
        e.first_rows[0] = execution.firstRows[0]; e.first_rows[1] = execution.firstRows[1];
        e.valid_cols[0] = execution.validCols[0]; e.valid_cols[1] = execution.validCols[1];
        e.first_cols[0] = execution.firstCols[0]; e.first_cols[1] = execution.firstCols[1];
        check(plr_set_compute_pass_execution(&e));
    }
    void setHostCallbackExecution(plr_host_callback callback, void* user, const char* name) { check(plr_set_host_callback_execution(callback, user, name)); }
    // with the resources the callback's work touches (plr.h plr_set_host_callback_execution_on)
    void setHostCallbackExecution(plr_host_callback callback, void* user, const char* name, const std::vector<ImageHandle>& images, const std::vector<StorageBufferHandle>& buffers) {
        std::vector<plr_image_handle> im;
        std::vector<plr_storage_buffer_handle> sb;
        for (const auto& i : images) im.push_back(toC(i));
        for (const auto& b : buffers) sb.push_back(b.index);
        check(plr_set_host_callback_execution_on(callback, user, name, im.data(), (uint32_t)im.size(), sb.data(), (uint32_t)sb.size()));
    }
    void getImageDevicePointer(const ImageHandle image, uint32_t mipLevel, void** outPtr, size_t* outSize) { check(plr_get_image_device_pointer(toC(image), mipLevel, outPtr, outSize)); }
    void prepareForDrawcallRecording() { check(plr_prepare_for_drawcall_recording()); }
    void setUniformBufferData(const UniformBufferHandle buffer, const void* data, const size_t size) { check(plr_set_uniform_buffer_data(buffer.index, data, size)); }
    void setStorageBufferData(const StorageBufferHandle buffer, const void* data, const size_t size) { check(plr_set_storage_buffer_data(buffer.index, data, size)); }
    // RenderBackend.h:73, "must be set once before creating renderpasses" (RenderFrontend.cpp:280-295): validated against the kernels' fixed set 0 (plr.h)
    void setGlobalDescriptorSetLayout(const ShaderLayout& layout) {
        plr_shader_layout l{};
        l.sampler_bindings = layout.samplerBindings.data(); l.sampler_binding_count = (uint32_t)layout.samplerBindings.size();
        l.sampled_image_bindings = layout.sampledImageBindings.data(); l.sampled_image_binding_count = (uint32_t)layout.sampledImageBindings.size();
        l.storage_image_bindings = layout.storageImageBindings.data(); l.storage_image_binding_count = (uint32_t)layout.storageImageBindings.size();
        l.uniform_buffer_bindings = layout.uniformBufferBindings.data(); l.uniform_buffer_binding_count = (uint32_t)layout.uniformBufferBindings.size();
        l.storage_buffer_bindings = layout.storageBufferBindings.data(); l.storage_buffer_binding_count = (uint32_t)layout.storageBufferBindings.size();
        check(plr_set_global_descriptor_set_layout(&l));
    }
    void setGlobalDescriptorSetResources(const RenderPassResources& resources) {
        std::vector<plr_uniform_buffer_resource> ub;
        std::vector<plr_sampler_resource> sm;
        std::vector<plr_image_resource> si;
        for (const auto& b : resources.uniformBuffers) ub.push_back({b.buffer.index, b.binding});
        for (const auto& s : resources.samplers) sm.push_back({s.sampler.index, s.binding});
        for (const auto& i : resources.sampledImages) si.push_back({toC(i.image), i.mipLevel, i.binding});
        plr_pass_resources r{};
        r.uniform_buffers = ub.data(); r.uniform_buffer_count = (uint32_t)ub.size();
        r.samplers = sm.data(); r.sampler_count = (uint32_t)sm.size();
        r.sampled_images = si.data(); r.sampled_image_count = (uint32_t)si.size();
        check(plr_set_global_descriptor_set_resources(&r));
    }
    void updateComputePassShaderDescription(const RenderPassHandle passHandle, const ShaderDescription& desc) {
        CDesc d(desc, nullptr);
        check(plr_update_compute_pass_shader_description(passHandle.index, &d.c));
    }
    void renderFrame(const bool presentToScreen) { check(plr_render_frame(presentToScreen ? 1 : 0)); }
    uint32_t getImageGlobalTextureArrayIndex(const ImageHandle image) {
        uint32_t idx = 0;
        check(plr_get_image_global_texture_array_index(toC(image), &idx));
        return idx;
    }
    RenderPassHandle createComputePass(const ComputePassDescription& desc) {
        CDesc d(desc.shaderDescription, desc.name.c_str());
        RenderPassHandle h;
        check(plr_create_compute_pass(&d.c, &h.index));
        return h;
    }
    ImageHandle createImage(const ImageDescription& description, const void* initialData, const size_t initialDataSize) {
        plr_image_desc d = toC(description);
        plr_image_handle h;
        check(plr_create_image(&d, initialData, initialDataSize, &h));
        return fromC(h);
    }
    UniformBufferHandle createUniformBuffer(const UniformBufferDescription& desc) {
        UniformBufferHandle h;
        check(plr_create_uniform_buffer(desc.size, desc.initialData, &h.index));
        return h;
    }
    StorageBufferHandle createStorageBuffer(const StorageBufferDescription& desc) {
        StorageBufferHandle h;
        check(plr_create_storage_buffer(desc.size, desc.initialData, &h.index));
        return h;
    }
    SamplerHandle createSampler(const SamplerDescription& description) {
        plr_sampler_desc d{(uint32_t)description.interpolation, (uint32_t)description.wrapping, description.useAnisotropy ? 1u : 0u, description.maxAnisotropy,
                           (uint32_t)description.borderColor, description.maxMip};
        SamplerHandle h;
        check(plr_create_sampler(&d, &h.index));
        return h;
    }
    ImageHandle createTemporaryImage(const ImageDescription& description) {
        plr_image_desc d = toC(description);
        plr_image_handle h;
        check(plr_create_temporary_image(&d, &h));
        return fromC(h);
    }
    ImageHandle getSwapchainInputImage() {
        plr_image_handle h;
        check(plr_get_swapchain_input_image(&h));
        return fromC(h);
    }
    void getMemoryStats(uint64_t* outAllocatedSize, uint64_t* outUsedSize) const { check(plr_get_memory_stats(outAllocatedSize, outUsedSize)); }
    std::vector<RenderPassTime> getRenderpassTimings() const {
        uint32_t n = 0;
        check(plr_get_renderpass_timings(nullptr, &n));
        std::vector<plr_renderpass_time> t(n);
        if (n) check(plr_get_renderpass_timings(t.data(), &n));
        std::vector<RenderPassTime> out;
        for (uint32_t i = 0; i < n; i++) out.push_back({t[i].time_ms, t[i].name ? t[i].name : ""});
        return out;
    }
    float getLastFrameCPUTime() const { float ms = 0; check(plr_get_last_frame_cpu_time(&ms)); return ms; }
    ImageDescription getImageDescription(const ImageHandle handle) {
        plr_image_desc d;
        check(plr_get_image_description(toC(handle), &d));
        ImageDescription r;
        r.width = d.width; r.height = d.height; r.depth = d.depth; r.type = (ImageType)d.type; r.format = (ImageFormat)d.format;
        r.usageFlags = (ImageUsageFlags)d.usage_flags; r.mipCount = (MipCount)d.mip_count; r.manualMipCount = d.manual_mip_count; r.autoCreateMips = d.auto_create_mips != 0;
        return r;
    }

    static plr_image_handle toC(ImageHandle h) { return plr_image_handle{(uint32_t)h.type, h.index}; }
    static ImageHandle fromC(plr_image_handle h) { ImageHandle r; r.type = (ImageHandleType)h.type; r.index = h.index; return r; }

private:
    static void check(int rc) {
        if (rc != PLR_OK) throw std::runtime_error(std::string("plr: ") + plr_last_error());
    }
    static plr_image_desc toC(const ImageDescription& d) {
        return plr_image_desc{d.width, d.height, d.depth, (uint32_t)d.type, (uint32_t)d.format, (uint32_t)d.usageFlags, (uint32_t)d.mipCount, d.manualMipCount,
                              d.autoCreateMips ? 1u : 0u};
    }
    struct CDesc {
        plr_compute_pass_desc c{};
        std::vector<plr_specialisation_constant> sc;
        CDesc(const ShaderDescription& s, const char* name) {
            for (const auto& k : s.specialisationConstants) sc.push_back({k.location, k.data.data(), (uint32_t)k.data.size()});
            c.src_path_relative = s.srcPathRelative.c_str();
            c.specialisation_constants = sc.data();
            c.specialisation_constant_count = (uint32_t)sc.size();
            c.name = name;
        }
    };
};

} // namespace plrhost
