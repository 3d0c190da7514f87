// C entry points of the frame pipeline (include/plr_frame.h).
#include <algorithm>
#include <cstring>
#include <stdexcept>
#include <string>

#include "../../../include/plr_frame.h"
#include "frame_pipeline.h"

using namespace plrhost;

static thread_local std::string g_ferr;
static int fail(const std::exception& e) { g_ferr = e.what(); return PLR_ERR_INVALID_ARGUMENT; }
#define PLRF_TRY(...) try { __VA_ARGS__; return PLR_OK; } catch (const std::exception& e) { return fail(e); }

extern "C" {

const char* plrf_last_error(void) { return g_ferr.c_str(); }

int plrf_default_settings(plrf_settings* o, uint32_t width, uint32_t height) {
    if (!o) return PLR_ERR_INVALID_ARGUMENT;
    const FramePipelineSettings d;
    std::memset(o, 0, sizeof(*o));
    o->width = width; o->height = height;
    o->shadow_map_res = d.shadowMapRes; o->brdf_lut_res = d.brdfLutRes; o->max_sdf_instances = d.maxSdfInstances; o->froxel_depth = d.froxelDepth;
    o->taa_enabled = d.taa.enabled; o->taa_use_clipping = d.taa.useClipping; o->taa_use_motion_vector_dilation = d.taa.useMotionVectorDilation;
    o->taa_history_sampling_tech = (uint32_t)d.taa.historySamplingTech; o->taa_filter_use_tonemapping = d.taa.filterUseTonemapping;
    o->bloom_enabled = d.bloom.enabled; o->bloom_strength = d.bloom.strength; o->bloom_radius = d.bloom.radius;
    o->sdf_half_res_trace = d.sdfTrace.halfResTrace; o->sdf_strict_influence_radius_cutoff = d.sdfTrace.strictInfluenceRadiusCutoff;
    o->sdf_trace_influence_radius = d.sdfTrace.traceInfluenceRadius;
    o->diffuse_brdf = (uint32_t)d.shading.diffuseBRDF; o->direct_multiscatter = (uint32_t)d.shading.directMultiscatter;
    o->indirect_lighting_tech = (uint32_t)d.shading.indirectLightingTech; o->use_geometry_aa = d.shading.useGeometryAA;
    o->sun_shadow_cascade_count = (uint32_t)d.shading.sunShadowCascadeCount;
    o->run_exposure = o->run_hiz = o->run_gi = o->run_shading = o->run_taa = o->run_bloom = o->run_tonemap = 1;
    o->run_light_matrix = d.runLightMatrix; o->volumetrics_max_distance = d.volumetricsMaxDistance; o->run_sky_luts = d.runSkyLuts; o->run_volumetrics = d.runVolumetrics; o->band_taa_history_halo = d.band.taaHistoryHalo; o->band_overlap_exchange = d.band.overlapExchange ? (d.band.rowsFirst ? 2u : 1u) : 0u;
    o->sdf_debug_mode = (uint32_t)d.sdfDebug.visualisationMode; o->sdf_debug_tile_usage_with_hiz = d.sdfDebug.showCameraTileUsageWithHiZ;
    o->sdf_debug_use_influence_radius = d.sdfDebug.useInfluenceRadiusForDebug;
    o->taa_use_separate_supersampling = d.taa.useSeparateSupersampling; o->taa_supersample_use_tonemapping = d.taa.supersampleUseTonemapping;
    // the denoiser's disc is 1.5 m in WORLD space: its reach in rows grows with the frame height, and so does the halo that keeps the band deviation
    // where it is at 2160 rows (measured at 8K in four bands, tests/test_config5_8k.py: 64 rows 98.6 %, 128 rows 99.3 %, 192 rows 99.4 % of a band's
    // pixels within one code of the unpartitioned frame)
    o->band_gi_halo = BandSettings::giHaloForHeight(height); o->band_gi_history_halo = d.band.giHistoryHalo; o->band_color_halo = d.band.colorHalo; o->band_post_halo = d.band.postHalo;
    return PLR_OK;
}

int plrf_create(const plrf_settings* s, void** out) {
    if (!s || !out) return PLR_ERR_INVALID_ARGUMENT;
    PLRF_TRY({
        FramePipelineSettings f;
        f.width = s->width; f.height = s->height; f.shadowMapRes = s->shadow_map_res; f.brdfLutRes = s->brdf_lut_res; f.maxSdfInstances = s->max_sdf_instances;
        f.froxelDepth = s->froxel_depth;
        f.taa.enabled = s->taa_enabled; f.taa.useClipping = s->taa_use_clipping; f.taa.useMotionVectorDilation = s->taa_use_motion_vector_dilation;
        f.taa.historySamplingTech = (HistorySamplingTech)s->taa_history_sampling_tech; f.taa.filterUseTonemapping = s->taa_filter_use_tonemapping;
        f.bloom.enabled = s->bloom_enabled; f.bloom.strength = s->bloom_strength; f.bloom.radius = s->bloom_radius;
        f.sdfTrace.halfResTrace = s->sdf_half_res_trace; f.sdfTrace.strictInfluenceRadiusCutoff = s->sdf_strict_influence_radius_cutoff;
        f.sdfTrace.traceInfluenceRadius = s->sdf_trace_influence_radius;
        f.shading.diffuseBRDF = (DiffuseBRDF)s->diffuse_brdf; f.shading.directMultiscatter = (DirectSpecularMultiscattering)s->direct_multiscatter;
        f.shading.indirectLightingTech = (IndirectLightingTech)s->indirect_lighting_tech; f.shading.useGeometryAA = s->use_geometry_aa;
        f.shading.sunShadowCascadeCount = (int)s->sun_shadow_cascade_count;
        f.runExposure = s->run_exposure; f.runHiZ = s->run_hiz; f.runGI = s->run_gi; f.runShading = s->run_shading; f.runTAA = s->run_taa;
        f.runBloom = s->run_bloom; f.runTonemap = s->run_tonemap;
        f.band.rowBegin = s->band_row_begin; f.band.rowEnd = s->band_row_end; f.band.giHalo = s->band_gi_halo; f.band.giHistoryHalo = s->band_gi_history_halo;
        f.band.colorHalo = s->band_color_halo; f.band.postHalo = s->band_post_halo;
        f.band.colBegin = s->band_col_begin; f.band.colEnd = s->band_col_end;
        f.runLightMatrix = s->run_light_matrix; f.volumetricsMaxDistance = s->volumetrics_max_distance; f.runSkyLuts = s->run_sky_luts; f.runVolumetrics = s->run_volumetrics; f.band.taaHistoryHalo = s->band_taa_history_halo; f.band.overlapExchange = s->band_overlap_exchange != 0; f.band.rowsFirst = s->band_overlap_exchange >= 2;
        f.sdfDebug.visualisationMode = (SDFVisualisationMode)s->sdf_debug_mode; f.sdfDebug.showCameraTileUsageWithHiZ = s->sdf_debug_tile_usage_with_hiz;
        f.sdfDebug.useInfluenceRadiusForDebug = s->sdf_debug_use_influence_radius;
        f.taa.useSeparateSupersampling = s->taa_use_separate_supersampling; f.taa.supersampleUseTonemapping = s->taa_supersample_use_tonemapping;
        *out = new FramePipeline(f);
    })
}

int plrf_destroy(void* p) { delete (FramePipeline*)p; return PLR_OK; }

int plrf_get_image(void* p, const char* name, plr_image_handle* out) {
    const ImageHandle h = ((FramePipeline*)p)->image(name);
    if (h.index == invalidIndex) { g_ferr = std::string("unknown image '") + name + "'"; return PLR_ERR_INVALID_ARGUMENT; }
    *out = RenderBackend::toC(h);
    return PLR_OK;
}
int plrf_get_storage_buffer(void* p, const char* name, plr_storage_buffer_handle* out) {
    StorageBufferHandle h;
    if (!((FramePipeline*)p)->storageBuffer(name, &h)) { g_ferr = std::string("unknown storage buffer '") + name + "'"; return PLR_ERR_INVALID_ARGUMENT; }
    *out = h.index;
    return PLR_OK;
}
int plrf_get_uniform_buffer(void* p, const char* name, plr_uniform_buffer_handle* out) {
    UniformBufferHandle h;
    if (!((FramePipeline*)p)->uniformBuffer(name, &h)) { g_ferr = std::string("unknown uniform buffer '") + name + "'"; return PLR_ERR_INVALID_ARGUMENT; }
    *out = h.index;
    return PLR_OK;
}
int plrf_add_sdf_volume(void* p, uint32_t res, const void* data, size_t bytes, uint32_t* out) { PLRF_TRY(*out = ((FramePipeline*)p)->addSdfVolume(res, data, bytes)) }
int plrf_add_sdf_volume_dds(void* p, const char* path, uint32_t* out, uint32_t outSize[3]) {
    PLRF_TRY({
        ImageDescription d;
        *out = ((FramePipeline*)p)->addSdfVolumeFromDds(path, &d);
        if (outSize) { outSize[0] = d.width; outSize[1] = d.height; outSize[2] = d.depth; }
    })
}
int plrf_set_sdf_scene(void* p, const void* inst, size_t ib, const void* bb, size_t bbb) { PLRF_TRY(((FramePipeline*)p)->setSdfScene(inst, ib, bb, bbb)) }
int plrf_set_sun_direction(void* p, const float d[3]) { PLRF_TRY(((FramePipeline*)p)->setSunDirection(d)) }
int plrf_set_camera_intrinsic(void* p, float fov, float n, float f) { PLRF_TRY(((FramePipeline*)p)->setCameraIntrinsic(fov, n, f)) }
int plrf_set_camera_cut(void* p) { PLRF_TRY(((FramePipeline*)p)->setCameraCut()) }
int plrf_frame(void* p, const plrf_camera* c, float dt, float time) {
    PLRF_TRY({
        CameraExtrinsic e;
        e.position = {c->position[0], c->position[1], c->position[2]};
        e.forward = {c->forward[0], c->forward[1], c->forward[2]};
        e.up = {c->up[0], c->up[1], c->up[2]};
        e.right = {c->right[0], c->right[1], c->right[2]};
        ((FramePipeline*)p)->frame(e, dt, time);
    })
}
int plrf_get_submitted_globals(void* p, void* out) { std::memcpy(out, &((FramePipeline*)p)->lastSubmittedGlobals(), 340); return PLR_OK; }
int plrf_get_resolve_weights(void* p, float* out) { std::memcpy(out, ((FramePipeline*)p)->lastResolveWeights(), 36); return PLR_OK; }
int plrf_get_cpu_frame_index(void* p, uint64_t* out) { *out = ((FramePipeline*)p)->cpuFrameIndex(); return PLR_OK; }

int plrf_set_exchange_callback(void* p, plrf_exchange_callback cb, void* user) { PLRF_TRY(((FramePipeline*)p)->setExchangeCallback(cb, user)) }
int plrf_get_exchange_items(void* p, int id, plrf_exchange_item* out, uint32_t* inoutCount) {
    if (id < 0 || id >= ExchangeCount || !inoutCount) { g_ferr = "invalid exchange id"; return PLR_ERR_INVALID_ARGUMENT; }
    PLRF_TRY({
        FramePipeline* fp = (FramePipeline*)p;
        const std::vector<ExchangeItem>& items = fp->exchangeItems(id);
        const uint32_t n = std::min<uint32_t>((uint32_t)items.size(), out ? *inoutCount : 0u);
        for (uint32_t i = 0; i < n; i++) {
            const ExchangeItem& it = items[i];
            size_t bytes = 0;
            out[i].image = RenderBackend::toC(it.image);
            fp->backend().getImageDevicePointer(it.image, it.mip, &out[i].device_ptr, &bytes);
            out[i].row_begin = it.rowBegin; out[i].row_end = it.rowEnd; out[i].halo_rows = it.haloRows; out[i].row_bytes = it.rowBytes; out[i].image_rows = it.imageRows;
            out[i].col_begin = it.colBegin; out[i].col_end = it.colEnd; out[i].image_cols = it.imageCols; out[i].texel_bytes = it.texelBytes;
        }
        *inoutCount = (uint32_t)items.size();
    })
}
int plrf_get_histogram_exchange(void* p, void** outPtr, size_t* outBytes) {
    PLRF_TRY({
        FramePipeline* fp = (FramePipeline*)p;
        size_t size = 0;
        if (plr_get_storage_buffer_device_pointer(fp->histogramBuffer().index, outPtr, &size) != PLR_OK) throw std::runtime_error(plr_last_error());
        *outBytes = 128 * sizeof(uint32_t);
    })
}
int plrf_get_gi_request_exchange(void* p, plrf_gi_request* out) {
    PLRF_TRY({
        FramePipeline* fp = (FramePipeline*)p;
        std::memset(out, 0, sizeof(*out));
        const BandSettings& b = fp->settings.band;
        out->enabled = b.enabled() && b.giRequested ? 1 : 0;
        if (!out->enabled) return PLR_OK;
        const SDFGI& gi = fp->sdfGi();
        const ImageDescription td = fp->backend().getImageDescription(gi.m_indirectDiffuse_Y_SH[0]);
        const uint32_t div = fp->settings.sdfTrace.halfResTrace ? 2u : 1u;
        out->image_cols = td.width; out->image_rows = td.height; out->row_words = gi.m_giRequestRowWords;
        out->x0 = b.tiled() ? b.colBegin / div : 0u; out->x1 = b.tiled() ? std::min((b.colEnd + div - 1) / div, td.width) : td.width;
        out->y0 = b.rowBegin / div; out->y1 = std::min((b.rowEnd + div - 1) / div, td.height);
        size_t bytes = 0;
        auto imagePtr = [&](ImageHandle h) { void* ptr = nullptr; fp->backend().getImageDevicePointer(h, 0, &ptr, &bytes); return ptr; };
        for (int i = 0; i < 2; i++)
            if (plr_get_storage_buffer_device_pointer(gi.m_giRequestBitmap[i].index, &out->bitmap[i], &bytes) != PLR_OK) throw std::runtime_error(plr_last_error());
        out->ysh[0] = imagePtr(gi.m_indirectDiffuse_Y_SH[0]); out->cocg[0] = imagePtr(gi.m_indirectDiffuse_CoCg[0]);
        out->ysh[1] = imagePtr(gi.m_indirectDiffuseHistory_Y_SH[1]); out->cocg[1] = imagePtr(gi.m_indirectDiffuseHistory_CoCg[1]);
        out->depth = imagePtr(fp->depthHalfRes());
    })
}
int plrf_band_rows_first(void* p) {
    const FramePipeline* fp = (const FramePipeline*)p;
    return fp && fp->settings.band.enabled() && fp->settings.band.overlapExchange && fp->settings.band.rowsFirst ? 1 : 0;
}
int plrf_get_depth_apex_exchange(void* p, void** outPtr, size_t* outBytes) {
    PLRF_TRY({
        FramePipeline* fp = (FramePipeline*)p;
        size_t size = 0;
        if (plr_get_image_device_pointer(RenderBackend::toC(fp->depthApexImage()), 0, outPtr, &size) != PLR_OK) throw std::runtime_error(plr_last_error());
        *outBytes = 2 * sizeof(float);
    })
}

} // extern "C"
