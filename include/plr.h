/* plr.h - C-ABI of the MI355X-native (HIP, gfx950) backend for PlainRenderer's per-pixel frame pipeline.
 *
 * Drop-in boundary: the compute subset of `class RenderBackend`'s public section in the reference
 * (Plain/src/Runtime/Rendering/Backend/RenderBackend.h:36-110) together with the plain-data pass-record
 * structs of Plain/src/Runtime/Rendering/ResourceDescriptions.h:9-172, RenderHandles.h:4-41 and
 * Common/ImageDescription.h:4-35. Every entry point cites the reference member it replaces.
 * A reference-side shim (class RenderBackend implemented over these calls) is shown in INTEGRATION.md and
 * shipped as include/plr_render_backend.hpp.
 *
 * Conventions: every call returns PLR_OK (0) or a negative error code (the reference prints and throws,
 * RenderBackend.cpp:442-445; a C boundary cannot throw); plr_last_error() gives the message.
 * All calls must come from one thread (the reference records compute passes on the main thread only).
 * Payload pointers are copied during the call (reference: dataToCharArray, Common/Utilities/GeneralUtils.cpp:8-12).
 * No torch types; plain pointers and sizes only.
 */
#ifndef PLR_H
#define PLR_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define PLR_OK 0
#define PLR_ERR_INVALID_ARGUMENT (-1)
#define PLR_ERR_HIP (-2)
#define PLR_ERR_UNKNOWN_SHADER (-3)
#define PLR_ERR_BINDING (-4)
#define PLR_ERR_NOT_INITIALISED (-5)
#define PLR_ERR_UNSUPPORTED (-6)

#define PLR_INVALID_INDEX 0xFFFFFFFFu /* RenderHandles.h:4 invalidIndex */

/* Common/ImageDescription.h:4-16 (same order, same values) */
enum plr_image_type { PLR_IMAGE_1D = 0, PLR_IMAGE_2D = 1, PLR_IMAGE_3D = 2, PLR_IMAGE_CUBE = 3 };
enum plr_mip_count { PLR_MIP_ONE = 0, PLR_MIP_FULL_CHAIN = 1, PLR_MIP_MANUAL = 2, PLR_MIP_FULL_CHAIN_ALREADY_IN_DATA = 3 };
enum plr_image_usage { PLR_USAGE_STORAGE = 1, PLR_USAGE_SAMPLED = 2, PLR_USAGE_ATTACHMENT = 4 };
enum plr_image_format {
    PLR_FORMAT_R8 = 0, PLR_FORMAT_RG8, PLR_FORMAT_RGBA8, PLR_FORMAT_R16_SFLOAT, PLR_FORMAT_RG16_SFLOAT,
    PLR_FORMAT_RG32_SFLOAT, PLR_FORMAT_RG16_SNORM, PLR_FORMAT_RGBA16_SFLOAT, PLR_FORMAT_RGBA16_SNORM,
    PLR_FORMAT_RGBA32_SFLOAT, PLR_FORMAT_R11G11B10_UFLOAT, PLR_FORMAT_DEPTH16, PLR_FORMAT_DEPTH32,
    PLR_FORMAT_BC1, PLR_FORMAT_BC3, PLR_FORMAT_BC5, PLR_FORMAT_BGRA8_UNORM
};

/* RenderHandles.h:16-21 ImageHandle {type, index} */
enum plr_image_handle_type { PLR_IMAGE_DEFAULT = 0, PLR_IMAGE_TRANSIENT = 1, PLR_IMAGE_SWAPCHAIN = 2 };
typedef struct plr_image_handle { uint32_t type; uint32_t index; } plr_image_handle;
typedef uint32_t plr_pass_handle;           /* RenderPassHandle.index (high bit = graphic pass, never set here) */
typedef uint32_t plr_uniform_buffer_handle; /* UniformBufferHandle.index */
typedef uint32_t plr_storage_buffer_handle; /* StorageBufferHandle.index */
typedef uint32_t plr_sampler_handle;        /* SamplerHandle.index */

/* ImageDescription, Common/ImageDescription.h:18-30 */
typedef struct plr_image_desc {
    uint32_t width, height, depth;
    uint32_t type;             /* plr_image_type */
    uint32_t format;           /* plr_image_format */
    uint32_t usage_flags;      /* plr_image_usage bits */
    uint32_t mip_count;        /* plr_mip_count */
    uint32_t manual_mip_count; /* only if mip_count == PLR_MIP_MANUAL */
    uint32_t auto_create_mips;
} plr_image_desc;

/* SamplerDescription, ResourceDescriptions.h:161-172 */
typedef struct plr_sampler_desc {
    uint32_t interpolation; /* 0 nearest, 1 linear */
    uint32_t wrapping;      /* 0 clamp, 1 border colour, 2 repeat */
    uint32_t use_anisotropy;
    float max_anisotropy;
    uint32_t border_color; /* 0 white, 1 black */
    uint32_t max_mip;
} plr_sampler_desc;

/* ResourceDescriptions.h:9-45 */
typedef struct plr_image_resource { plr_image_handle image; uint32_t mip_level; uint32_t binding; } plr_image_resource;
typedef struct plr_storage_buffer_resource { plr_storage_buffer_handle buffer; uint32_t read_only; uint32_t binding; } plr_storage_buffer_resource;
typedef struct plr_uniform_buffer_resource { plr_uniform_buffer_handle buffer; uint32_t binding; } plr_uniform_buffer_resource;
typedef struct plr_sampler_resource { plr_sampler_handle sampler; uint32_t binding; } plr_sampler_resource;

/* RenderPassResources, ResourceDescriptions.h:47-53, as flat arrays */
typedef struct plr_pass_resources {
    const plr_sampler_resource* samplers; uint32_t sampler_count;
    const plr_storage_buffer_resource* storage_buffers; uint32_t storage_buffer_count;
    const plr_uniform_buffer_resource* uniform_buffers; uint32_t uniform_buffer_count;
    const plr_image_resource* sampled_images; uint32_t sampled_image_count;
    const plr_image_resource* storage_images; uint32_t storage_image_count;
} plr_pass_resources;

/* ComputePassExecution, ResourceDescriptions.h:57-60,74-78 */
typedef struct plr_compute_pass_execution {
    plr_pass_handle handle;
    plr_pass_resources resources;
    const void* push_constants; uint32_t push_constant_size;
    uint32_t dispatch_count[3];
    /* extension (no reference counterpart): first workgroup of the dispatch, vkCmdDispatchBase semantics. Zero for the
     * reference's recorder code. A band renderer (one GPU per range of screen rows) sets [1] so a pass covers only its rows;
     * [0]: the first workgroup COLUMN (tile rendering, see valid_cols below) for the passes of the per-pixel frame path, the first tile for
     * histogramCombineTiles, an error (PLR_ERR_UNSUPPORTED) for every other pass; [2] must be 0. */
    uint32_t dispatch_base[3];
    /* extension (band rendering): rows [valid_rows[0], valid_rows[1]) of the pass's INPUT images hold valid data - the band's own rows plus the
     * halo rows received from the neighbouring GPUs; {0, 0} = every row (the reference's recorder code). Honoured by filterIndirectDiffuseSpatial,
     * whose world-space disc can reach arbitrarily far on near geometry: a sample that lands on a row outside gets the shader's own off-screen
     * treatment (weight 0, disc shrinks: filterIndirectDiffuseSpatial.comp:100-105) instead of reading rows no neighbour sent. */
    uint32_t valid_rows[2];
    /* extension (no reference counterpart): non-zero = this execution belongs to the frame's ASYNCHRONOUS TAIL. The backend launches it on a second
     * HIP stream, ordered behind everything recorded before it; executions recorded after it - the next frame's included - run beside it unless
     * they touch an image / buffer it touches (the same read / write tracking the Vulkan backend derives its barriers from, RenderBackend.cpp:632-767),
     * in which case the launch stream waits for the tail first. Host reads / writes of device memory (plr_download_*, plr_upload_*,
     * plr_wait_for_gpu_idle, host callbacks) always wait for it. Results are identical with plr_set_async_tail(0), which runs everything in order.
     * The C++ host mirror flags the bloom chain + tonemap: ten short, dependent launches that leave the chip mostly idle, and that nothing reads
     * before the next frame's TAA resolve. */
    uint32_t async_tail;
    /* extension (band rendering, round 4): workgroup rows (dispatch_base units) [dispatch_base[1], first_rows[0]) and [first_rows[1], dispatch_base[1] +
     * dispatch_count[1]) - the rows a neighbouring GPU needs - are to be produced FIRST, and the backend's edge signal is raised as soon as they are
     * complete in memory, while the rest of the SAME launch is still running: a halo exchange can start on another stream (hipStreamWaitValue32 on
     * plr_get_edge_signal) without the pass being split into an edge and an interior launch (two launches with two tails: +36 us per band for the
     * trace at 8K). {0, 0} = off. A kernel that cannot order its rows raises the signal when the whole launch has finished, so waiting for the
     * signal is always correct; results do not depend on it. */
    uint32_t first_rows[2];
    /* extension (tile rendering, round 5: the frame partitioned into screen TILES, BASELINE config 5's 2 x 2): dispatch_base[0] / dispatch_count[0] restrict a pass
     * to the workgroup columns of its tile the way [1] restricts it to rows - honoured by the passes of the per-pixel frame path (histogramPerTile,
     * depthHiZPyramid, depthDownscale, sdfCameraTileCulling, sdfDiffuseTrace, the GI filters, indirectLightUpscale, deferredShading, temporalFilter, the bloom
     * chain, applyBloom, tonemapping; PLR_ERR_UNSUPPORTED for any other pass). valid_cols is valid_rows for columns: columns [valid_cols[0], valid_cols[1]) of
     * the INPUT images hold this frame's data ({0, 0} = all). first_cols is first_rows for columns: workgroup columns [dispatch_base[0], first_cols[0]) and
     * [first_cols[1], dispatch_base[0] + dispatch_count[0]) belong to the edge that is produced first - with first_rows the frame of the tile. */
    uint32_t valid_cols[2];
    uint32_t first_cols[2];
} plr_compute_pass_execution;

/* extension: host function executed in recording order while plr_render_frame launches the recorded passes; it may enqueue
 * work (copies, collectives) on the launch stream it is handed. Non-zero return aborts the frame. Used for the halo exchange
 * between passes when the frame is partitioned over GPUs. */
typedef int (*plr_host_callback)(void* user, void* hip_stream);

/* SpecialisationConstant, ResourceDescriptions.h:112-115: raw bytes with the C++ sizeof (bool = 1 byte) */
typedef struct plr_specialisation_constant { uint32_t location; const void* data; uint32_t size; } plr_specialisation_constant;

/* ShaderDescription + ComputePassDescription, ResourceDescriptions.h:117-120,146-149 */
typedef struct plr_compute_pass_desc {
    const char* src_path_relative; /* e.g. "temporalFilter.comp": selects the precompiled HIP kernel */
    const plr_specialisation_constant* specialisation_constants; uint32_t specialisation_constant_count;
    const char* name; /* debug label, reported by plr_get_renderpass_timings */
} plr_compute_pass_desc;

/* RenderPassTime, Backend/VulkanTimestampQueries.h:17-20; name points into backend-owned storage valid until the pass is destroyed */
typedef struct plr_renderpass_time { float time_ms; const char* name; } plr_renderpass_time;

/* ---- lifetime: RenderBackend::setup / shutdown / recreateSwapchain (RenderBackend.h:36-38).
 * There is no window: the "swapchain input image" is an off-screen BGRA8 image of the given size. */
int plr_setup(int device_ordinal, uint32_t width, uint32_t height);
int plr_shutdown(void);
int plr_recreate_swapchain(uint32_t width, uint32_t height);
const char* plr_last_error(void);

/* RenderBackend::waitForGPUIdle, RenderBackend.h:41 */
int plr_wait_for_gpu_idle(void);
/* RenderBackend::updateShaderCode, RenderBackend.h:44: kernels are precompiled, nothing to reload */
int plr_update_shader_code(void);
/* RenderBackend::resizeImages, RenderBackend.h:47 */
int plr_resize_images(const plr_image_handle* images, uint32_t count, uint32_t width, uint32_t height);
/* RenderBackend::newFrame, RenderBackend.h:50: drops recorded executions and transient images */
int plr_new_frame(void);
/* RenderBackend::setComputePassExecution, RenderBackend.h:56 */
int plr_set_compute_pass_execution(const plr_compute_pass_execution* execution);
/* extension: see plr_host_callback; name is the label reported by plr_get_renderpass_timings */
int plr_set_host_callback_execution(plr_host_callback callback, void* user, const char* name);
/* the same, with the images and storage buffers the callback's work reads or writes (a halo exchange knows its images): the frame's asynchronous tail
 * (async_tail above) then waits for the callback only if it shares one of them, instead of always. n_images = n_buffers = 0: touches nothing. */
int plr_set_host_callback_execution_on(plr_host_callback callback, void* user, const char* name, const plr_image_handle* images, uint32_t n_images,
                                       const plr_storage_buffer_handle* buffers, uint32_t n_buffers);
/* RenderBackend::prepareForDrawcallRecording, RenderBackend.h:59: resolves transient images, validates bindings */
int plr_prepare_for_drawcall_recording(void);
/* RenderBackend::setUniformBufferData / setStorageBufferData, RenderBackend.h:64-70:
 * data is copied now and applied, in call order, at the start of the next plr_render_frame */
int plr_set_uniform_buffer_data(plr_uniform_buffer_handle buffer, const void* data, size_t size);
int plr_set_storage_buffer_data(plr_storage_buffer_handle buffer, const void* data, size_t size);
/* RenderBackend::setGlobalDescriptorSetLayout, RenderBackend.h:73 ("must be set once before creating renderpasses"; ShaderLayout, Backend/Resources.h:9-15; caller:
 * RenderFrontend::setupGlobalShaderInfoLayout, RenderFrontend.cpp:280-295). The kernels' set 0 is fixed - binding 0 the `global` uniform buffer, bindings 1..8 the
 * eight samplers of resources/shaders/global.inc:35-42, binding 9 the (graphics-only) noise texture - so the call VALIDATES the layout against it instead of
 * creating anything: a uniform buffer at binding 0 is required, a sampler outside 1..8, a uniform buffer other than 0 or any storage resource in set 0 is
 * PLR_ERR_BINDING. Once a layout is set, plr_set_global_descriptor_set_resources refuses a resource at a binding the layout does not declare. */
typedef struct plr_shader_layout {
    const uint32_t* sampler_bindings; uint32_t sampler_binding_count;
    const uint32_t* sampled_image_bindings; uint32_t sampled_image_binding_count;
    const uint32_t* storage_image_bindings; uint32_t storage_image_binding_count;
    const uint32_t* uniform_buffer_bindings; uint32_t uniform_buffer_binding_count;
    const uint32_t* storage_buffer_bindings; uint32_t storage_buffer_binding_count;
} plr_shader_layout;
int plr_set_global_descriptor_set_layout(const plr_shader_layout* layout);
/* RenderBackend::setGlobalDescriptorSetResources, RenderBackend.h:75: set 0 = binding 0 `global` UBO plus the
 * eight samplers of resources/shaders/global.inc:35-42 at bindings 1..8 (sampler semantics are fixed by binding) */
int plr_set_global_descriptor_set_resources(const plr_pass_resources* resources);
/* RenderBackend::updateComputePassShaderDescription, RenderBackend.h:79 */
int plr_update_compute_pass_shader_description(plr_pass_handle pass, const plr_compute_pass_desc* desc);
/* RenderBackend::renderFrame, RenderBackend.h:82: applies deferred buffer fills, launches the recorded passes
 * in record order on one HIP stream (in-order = the reference's barrier rule), does not block the host */
int plr_render_frame(int present_to_screen);
/* RenderBackend::getImageGlobalTextureArrayIndex, RenderBackend.h:84 (bindless set 2) */
int plr_get_image_global_texture_array_index(plr_image_handle image, uint32_t* out_index);
/* RenderBackend::createComputePass, RenderBackend.h:88 */
int plr_create_compute_pass(const plr_compute_pass_desc* desc, plr_pass_handle* out_pass);
/* RenderBackend::createImage, RenderBackend.h:93 */
int plr_create_image(const plr_image_desc* desc, const void* initial_data, size_t initial_data_size, plr_image_handle* out_image);
/* RenderBackend::createUniformBuffer / createStorageBuffer, RenderBackend.h:94-95 */
int plr_create_uniform_buffer(size_t size, const void* initial_data, plr_uniform_buffer_handle* out_buffer);
int plr_create_storage_buffer(size_t size, const void* initial_data, plr_storage_buffer_handle* out_buffer);
/* RenderBackend::createSampler, RenderBackend.h:96 */
int plr_create_sampler(const plr_sampler_desc* desc, plr_sampler_handle* out_sampler);
/* RenderBackend::createTemporaryImage, RenderBackend.h:101: valid until the next plr_new_frame */
int plr_create_temporary_image(const plr_image_desc* desc, plr_image_handle* out_image);
/* RenderBackend::getSwapchainInputImage, RenderBackend.h:103 */
int plr_get_swapchain_input_image(plr_image_handle* out_image);
/* RenderBackend::getMemoryStats, RenderBackend.h:105 */
int plr_get_memory_stats(uint64_t* out_allocated_size, uint64_t* out_used_size);
/* RenderBackend::getRenderpassTimings, RenderBackend.h:107: hipEvent time of each pass of the last completed frame
 * (enable with plr_set_pass_timing; off by default because the event pairs serialise the launch stream) */
int plr_get_renderpass_timings(plr_renderpass_time* out_times, uint32_t* inout_count);
/* RenderBackend::getLastFrameCPUTime, RenderBackend.h:108 */
int plr_get_last_frame_cpu_time(float* out_ms);
/* RenderBackend::getImageDescription, RenderBackend.h:110 */
int plr_get_image_description(plr_image_handle image, plr_image_desc* out_desc);

/* ---- additions with no reference counterpart (host <-> HBM transfer for tests and benchmarks) ---- */
/* Kernel arithmetic mode. PLR_MATH_EXACT: every pass evaluates the shader's operations in source order with IEEE divide/sqrt,
 * software transcendentals and no FMA contraction (bit-identical to a scalar IEEE evaluation). PLR_MATH_FAST (default): passes
 * that have a restructured kernel (FMA, v_rcp/v_rsq/v_exp/v_log, LDS tiling, algebraic simplification) use it; results stay
 * within the per-pass tolerances stated in DESIGN.md. Integer/bit-exact outputs (histogram, HiZ, culling) are identical in both. */
#define PLR_MATH_EXACT 0
#define PLR_MATH_FAST 1
int plr_set_math_mode(int mode);
int plr_get_math_mode(int* out_mode);
int plr_set_pass_timing(int enabled);
/* Stream overlap (default off - a cross-stream dependency costs 15-20 us on MI355X, more than the hot path's short independent chains
 * win back; PLR_STREAM_OVERLAP=1 in the environment turns it on at plr_setup). When on, plr_render_frame launches the
 * recorded executions in order, but an execution that has no read-after-write / write-after-read / write-after-write hazard with the
 * tail of the main stream - judged from the allocations it binds - is placed on one of three side HIP streams and runs beside it
 * (the Vulkan backend derives its barriers from the same read/write tracking, RenderBackend.cpp:632-767, and leaves independent passes unordered).
 * All side streams join the main stream before a host callback and at the end of the frame. Results do not depend on the setting.
 * out_overlapped_executions: how many executions of the last plr_render_frame ran on a side stream. */
int plr_set_stream_overlap(int enabled);
/* asynchronous frame tail (plr_compute_pass_execution::async_tail; default on, PLR_ASYNC_TAIL=0 turns it off): out_async_executions = executions of
 * the last plr_render_frame launched on the tail stream. plr_get_last_frame_gpu_time covers the launch stream only. */
int plr_set_async_tail(int enabled);
int plr_get_async_tail(int* out_enabled, uint32_t* out_async_executions);
/* Early parts (round 6; default OFF - measured slower on MI355X, see the end of this comment; PLR_EARLY_PARTS=1 turns it on; PLR_MATH_FAST with pass fusion only). A fused launch may have a part that depends on none of
 * the frame's intermediate results: the deferred shade's DIRECT lighting (triangle.frag:146-290 - material, cascade select, twelve-tap PCF, diffuse + GGX +
 * multiscattering response to the sun, the froxel lookup) needs the G-buffer, the shadow cascades and the LUTs, not the GI chain recorded in front of it
 * (RenderFrontend.cpp:342-405: the shade is recorded behind trace, denoise and upscale). The backend issues such a part as a launch of its own on a third stream
 * as soon as the resources recorded with the executions allow - behind the last execution or buffer fill of the frame that writes anything it reads, at the
 * frame's start if there is none - beside the executions recorded in between, and the sequence itself then runs what is left (upscale + indirect + fog + pack).
 * The recorded executions, their order and their outputs' meaning are unchanged; the colour target is within one R11G11B10 code of the single launch
 * (tests/test_fusion.py), every other image byte-identical. out_early_launches: early parts launched by the last plr_render_frame (0: nothing to run beside,
 * a decision-signature buffer is set, the launcher declined the bindings). enabled = 2: the early part is split off even when nothing is recorded that it could run
 * beside (the parity tests hold the two-launch form to the oracle on a frame that records the pair alone).
 * Measured (profiles/r06_overlap.txt, 4K frame): one launch 0.705 ms; two launches back to back 0.814 (direct 168 us + combine 92 us against 176 us fused: the 36-byte
 * record per pixel makes both halves HBM-heavy); direct lighting beside trace .. second spatial filter 0.770, from the frame's start 0.79 - the kernels it runs beside
 * lose what it gains (trace 104 -> 191 us, spatial filter 104 -> 122): a latency-bound kernel's waves hold the register file, there is no idle capacity to fill. */
int plr_set_early_parts(int enabled);
int plr_get_early_parts(int* out_enabled, uint32_t* out_early_launches);
/* Pass fusion (default level 2, PLR_MATH_FAST only): where the recorded frame contains certain shaders back to back - histogramReset +
 * histogramCombineTiles + preExposeLights; depthHiZPyramid + depthDownscale; sdfCameraFrustumCulling + sdfCameraTileCulling; applyBloom +
 * tonemapping; a GI pass followed by the spatial filter that reads its output - the backend covers them with fewer kernel launches. The
 * boundary is unchanged (one plr_set_compute_pass_execution per reference dispatch) and so are the results (byte-identical with fusion
 * off, tests/test_fusion.py). out_fused_executions: how many executions of the last plr_render_frame ran inside a fused launch.
 * enabled = 2 (PLR_PASS_FUSION=2) additionally lets a fused launcher ELIDE an intermediate image: indirectLightUpscale + the deferred shade run as
 * one kernel that hands the upscaled GI texels over in registers / LDS, and when no other execution recorded for the frame binds the two upscaled
 * images they are not written at all; plr_download_image of such an image fails (it does not return last frame's bytes). Every other image
 * and buffer is byte-identical to level 1 (tests/test_fusion.py compares levels 2, 1 and 0). Fusion and plr_set_stream_overlap(1) exclude each
 * other: the side-stream scheduler launches pass by pass. */
int plr_set_pass_fusion(int enabled);
int plr_get_pass_fusion(int* out_enabled, uint32_t* out_fused_executions);
/* Fusion across the caller's pass order (default on; only with pass fusion on). The caller records in the reference's order
 * (RenderFrontend.cpp:342-405), and with the sky LUT / light matrix passes recorded as compute they sit BETWEEN the members of the fused frame front.
 * Before launching, the backend moves such an execution in front of or behind the group when the resources recorded with the executions say the result
 * cannot change: no allocation shared (with a write on either side) with the group members it moves across, nor with an execution it overtakes. Host
 * callbacks are never crossed. The order of the recorded executions is changed in place; results are byte-identical with the setting off
 * (tests/test_fusion.py). */
int plr_set_pass_fusion_reorder(int enabled);
/* the edge signal of the most recently launched execution with first_rows (valid inside the host callback recorded right behind it): a 32-bit word in
 * signal memory that reaches *out_value (monotonic, compare with >=) once that execution's first rows are complete. *out_signal = NULL: the platform
 * has no stream memory operations - order behind the launch stream instead. */
int plr_get_edge_signal(void** out_signal, uint32_t* out_value);
int plr_get_stream_overlap(int* out_enabled, uint32_t* out_overlapped_executions);
/* In PLR_MATH_FAST an execution outside the configuration its fast kernel was built for - or of a shader without one - runs the general
 * (exact-order) kernel of the shader. That is correct but slow, so it is not silent: out_count = executions of the last plr_render_frame that
 * took the general kernel, out_names (optional) = "<pass name> [<shader>], ..." truncated to names_capacity. bench.py refuses to report a
 * frame whose count is not zero. (Executions covered by a fused launch are not counted: a fused launcher is fast-set code.) */
int plr_get_general_kernel_executions(uint32_t* out_count, char* out_names, size_t names_capacity);
/* GPU time of the last plr_render_frame (hipEvents on the launch stream); blocks until that frame finished */
/* Frames are bracketed by the two events only on demand (an event record is a ~6 us barrier packet on the launch stream): with plr_set_pass_timing(1),
 * or because this getter was called since the previous plr_render_frame. So a caller that polls it once per frame (a frame-time overlay; the reference has no such getter, its
 * getRenderpassTimings, RenderBackend.h:107, reports a past frame in the same way) reads the time of the last frame from its second call on; *out_ms = the most recent bracketed frame's time,
 * 0 before there is one. Never an error for an untimed frame. */
int plr_get_last_frame_gpu_time(float* out_ms);
/* replay the recorded frame `count` times back to back; returns total GPU ms between first launch and last completion */
int plr_replay_frame(uint32_t count, float* out_total_gpu_ms);
int plr_upload_image(plr_image_handle image, uint32_t mip_level, const void* data, size_t size);
/* rows [row_begin, row_begin + row_count) of a 2D mip level (band rendering: a GPU only needs the inputs of its rows) */
int plr_upload_image_rows(plr_image_handle image, uint32_t mip_level, uint32_t row_begin, uint32_t row_count, const void* data, size_t size);
int plr_download_image(plr_image_handle image, uint32_t mip_level, void* out_data, size_t size);
int plr_download_storage_buffer(plr_storage_buffer_handle buffer, void* out_data, size_t offset, size_t size);
int plr_download_uniform_buffer(plr_uniform_buffer_handle buffer, void* out_data, size_t offset, size_t size);
/* Raw interop. async_tail executions run on a second stream: plr_get_stream() first makes the launch stream wait for that tail, so work the
 * caller orders on the returned stream - presenting or reading the swapchain, a collective on an image - is behind everything recorded so far,
 * tail included. Call it AFTER plr_render_frame for that guarantee (a stream obtained earlier and used after a later plr_render_frame is ordered
 * behind the launch stream only: ask again, or plr_wait_for_gpu_idle). The two pointer getters imply NO ordering - an address is stable for the
 * life of the resource and may be asked for at any time; order the accesses on plr_get_stream(), or inside a host callback execution that lists
 * the resource (plr_set_host_callback_execution_on), which the backend orders against the tail itself.
 * plr_get_image_device_pointer: device address / byte size of one mip level (HBM resident; lets a caller fill inputs device-to-device). Fails
 * for an image the last frame's fused launch left unwritten (pass fusion level 2), like plr_download_image. An image whose address was
 * handed out may be written behind the backend's back from then on: launchers stop caching tables derived from it (backend.h contentVersionOf). */
int plr_get_image_device_pointer(plr_image_handle image, uint32_t mip_level, void** out_ptr, size_t* out_size);
int plr_get_storage_buffer_device_pointer(plr_storage_buffer_handle buffer, void** out_ptr, size_t* out_size);
/* the hipStream_t the passes are launched on (async_tail executions: a second stream, joined as described above) */
int plr_get_stream(void** out_hip_stream);
/* the same stream WITHOUT the join: for a caller that fetches the stream every frame only to enqueue work that does not touch what async_tail
 * executions write (plr_get_stream's join makes the launch stream wait for the tail - it serialises what the tail was meant to overlap, ~20-60 us per
 * frame when called between frames). Ordering against the tail is then the caller's business. */
int plr_get_launch_stream(void** out_hip_stream);
/* raw copies ordered on the launch stream of the calling thread's backend: device-to-device (asynchronous), device-to-host and
 * host-to-device (both return when the copy is done). For exchange callbacks that move rows between two backends of one process. */
int plr_copy_device_memory(void* dst, const void* src, size_t size);
/* rows x width_bytes from src (src_pitch bytes per row) to dst (dst_pitch): a rectangle of an image (tile rendering, exchange between two backends of one process) */
int plr_copy_device_memory_2d(void* dst, size_t dst_pitch, const void* src, size_t src_pitch, size_t width_bytes, size_t rows);
int plr_read_device_memory(void* dst_host, const void* src, size_t size);
int plr_write_device_memory(void* dst, const void* src_host, size_t size);
/* lists the shader names the backend has kernels for; returns the count */
int plr_get_supported_shaders(const char** out_names, uint32_t capacity);
/* detmath / codec probes on the device (same function ids as oracle/probes.cpp); pointers are host memory */
int plr_debug_math_eval(int fn, const float* a, const float* b, float* out, int64_t n);
int plr_debug_codec_eval(int fn, const void* in, void* out, int64_t n);
/* decision signatures (parity tests of the PLR_MATH_FAST kernel set). While a buffer of `words` 32-bit words is set (0 frees it), the fast kernels
 * of sdfDiffuseTrace, filterIndirectDiffuseSpatial, indirectLightUpscale and the deferred shade also write one word per output pixel
 * (index y * outputWidth + x) recording the pixel's discrete decisions: ray hit / closest instance / shadow bit, parity of every disc
 * sample's nearest texel, edge + closest-depth texel, shadow cascade + number of lit PCF taps. Bit layout: oracle/oracle.h,
 * orc_set_decision_signature (the oracle emits the same words). A test compares the words to tell apart "same decisions, results must
 * agree to the storage quantum" from "a float rounding flipped a decision". Costs nothing when no buffer is set. */
int plr_debug_set_decision_signature(size_t words);
int plr_debug_read_decision_signature(uint32_t* out_words, size_t words);
/* the PLR_MATH_FAST luminance histogram bins by comparison against a threshold table instead of evaluating the logarithm (kernels_fast/
 * histogram_fast.hip): this checks the table-driven bin against the shader's formula for ALL 2^32 float bit patterns. *out_mismatches must be 0. */
int plr_debug_verify_histogram_thresholds(float min_luminance, float max_luminance, uint64_t* out_mismatches);
/* the PLR_MATH_FAST R11G11B10 encoder (device/image.h packR11G11B10Fast) against the exact one for ALL 2^32 float bit patterns, per channel width:
 * out[0] / out[1] = patterns whose 11-bit / 10-bit code differs, out[2] = largest code difference, out[3] = largest input bit pattern that differs
 * (expected: differences of one code, only below 2^-14 = 0x38800000; see the encoder's comment) */
int plr_debug_verify_r11g11b10_fast(uint64_t* out4);
/* the PLR_MATH_FAST sky LUT lookup (polynomial acos / atan, device/fastmath.h) for n directions (3 floats each) -> n x 3 floats; the oracle's
 * orc_kat_sky_lut takes the same arguments */
int plr_debug_sky_lut_eval(plr_image_handle sky_lut, const float* directions, float* out_rgb, int64_t n);
/* the PLR_MATH_FAST deferred shade reads the twelve PCF taps of calcShadow (triangle.frag:100-110) from a table indexed by the pixel's 8-bit noise
 * value (kernels_fast/pcf_taps.h) instead of evaluating sqrt / sin / cos per pixel: this copies the table the kernel uses to out_xy (host memory,
 * 256 x 12 x 2 floats = unit-disc offsets (cos(angle) d, sin(angle) d) of noise byte k, tap i at [(k * 12 + i) * 2]). The oracle's orc_kat_pcf_taps
 * evaluates the shader's expressions for the same 3072 entries; the two must agree bit for bit. */
int plr_debug_pcf_tap_table(float* out_xy, size_t floats);
/* sampler probe: evaluates one of the global samplers of resources/shaders/global.inc:35-42 on an image, with the device sampler code the pass
 * kernels are built from. filter: 0 nearest, 1 linear, 2 textureGather (component 0); address: 0 clamp-to-edge, 1 repeat, 2 border white,
 * 3 border black. coords: n x 2 (2D) or n x 3 (3D image) normalised coordinates, out: n x 4 floats; both host memory.
 * Same arguments as the oracle's orc_sampler_eval. */
int plr_debug_sampler_eval(plr_image_handle image, uint32_t mip_level, int filter, int address, const float* coords, float* out, int64_t n);

#ifdef __cplusplus
}
#endif
#endif /* PLR_H */
