// SDF diffuse GI for gfx950, part 1: depthDownscale.comp, sdfCameraFrustumCulling.comp, sdfCameraTileCulling.comp, sdfDebugVisualisation.comp
// (+ SDF.inc, sdfCulling.inc, sampling.inc, sunShadowCascades.inc, sky.inc, SphericalHarmonics.inc);
// host side Techniques/SDFGI.cpp:380-419,538-630 and RenderFrontend.cpp:873-892.
//
// Mapping to CDNA4: one wave64 is one 8x8 reference workgroup (the shared-memory ray exchange of resolveColor becomes a
// per-wave LDS slab), four waves of a block share one 32x32-px culling tile so the culled instance list and the 96-byte
// SDFInstance records are wave-uniform scalar loads. SDF volumes (64^3 half floats, 512 KiB each) are fetched with explicit
// trilinear address math; 256 of them (134 MB) sit in the 256 MB Infinity Cache after the first touch.
// Culling is rewritten as ordered wave compaction, which makes the reference's atomic-append order deterministic
// (ascending instance index) without changing which instances survive.
#include "../backend.h"
#include "../device/shading_common.h"
#include "../device/culling_device.h"
#include "../device/hiz_fast_device.h"
#include "../device/sdf_march_device.h"

namespace plr {

// ------------------------------------------------------------------------------------------------
// depthDownscale.comp:12-20: half-res R16F depth = nearest full-res texel at (2*iUV + 0.5) / res
__global__ __launch_bounds__(256) void depthDownscaleKernel(ImgView src, ImgView dst, int coverW, int coverH, int yBase, int xBase) {
    const int x = xBase + (int)(blockIdx.x * 64u + (threadIdx.x & 63u)); // columns [xBase, coverW) (tile rendering: PassCtx::colSpan)
    const int y = yBase + (int)(blockIdx.y * 4u + (threadIdx.x >> 6));
    if (x >= coverW || y >= coverH) return;
    const vec2 texelSize(1.f / (float)src.w, 1.f / (float)src.h);
    const vec2 uv(((float)(x * 2) + 0.5f) * texelSize.x, ((float)(y * 2) + 0.5f) * texelSize.y);
    const float depth = sampleNearest2D<F_D32, CLAMP>(src, uv).x;
    ((uint16_t*)dst.ptr)[(size_t)y * (size_t)dst.w + x] = (uint16_t)floatToHalfBits(depth);
}
static int launchDepthDownscale(const PassCtx& c) {
    if (int rc = c.needStorage(0, F_R16F, "depthDownscale halfResDst")) return rc;
    if (int rc = c.needSampled(1, F_D32, "depthDownscale fullResSrc")) return rc;
    const ImgView& dst = c.storage[0];
    const PassCtx::RowSpan rs = c.rowSpan(dst.h);
    const PassCtx::ColSpan cs = c.colSpan(dst.w);
    const int w = cs.x1, x0 = cs.x0, h = rs.y1, y0 = rs.y0; // columns [x0, w), rows [y0, h)
    if (w <= x0 || h <= y0) return 0;
    depthDownscaleKernel<<<dim3(divUp((unsigned)(w - x0), 64u), divUp((unsigned)(h - y0), 4u)), 256, 0, c.stream>>>(c.sampled[1], dst, w, h, y0, x0);
    PLR_CHECK_LAUNCH(c);
    return 0;
}
PLR_REGISTER_SHADER("depthDownscale.comp", launchDepthDownscale);
// Stand-alone the pass is a stream of one nearest fetch and one half conversion per output texel: nothing for a second kernel to restructure, so this one
// serves both math modes. (The default frame never launches it: the fused frame front writes the half-resolution depth from the pyramid's quad blocks.)
PLR_REGISTER_SHADER_FAST("depthDownscale.comp", launchDepthDownscale);

// ------------------------------------------------------------------------------------------------
// sdfCameraFrustumCulling.comp:36-62 as one block doing an ordered stream compaction.
__global__ __launch_bounds__(1024) void frustumCullingKernel(const uint32_t* __restrict__ instanceBuffer, const FrustumUbo* __restrict__ frustum,
                                                             uint32_t* __restrict__ culled, const BoundingBox* __restrict__ bbs,
                                                             const float* __restrict__ influenceRangeP, uint32_t threadLimit, uint32_t capacity) {
    __shared__ uint32_t waveTotals[16];
    __shared__ uint32_t base;
    const uint32_t t = threadIdx.x, lane = t & 63u, wave = t >> 6;
    const uint32_t instanceCount = min(instanceBuffer[0], threadLimit);
    const float influenceRange = *influenceRangeP;
    if (t == 0) base = culled[0];
    __syncthreads();
    for (uint32_t chunk = 0; chunk < instanceCount; chunk += 1024u) {
        const uint32_t instanceIndex = chunk + t;
        bool inside = false;
        if (instanceIndex < instanceCount) {
            inside = insideFrustum(bbs[instanceIndex], frustum, influenceRange);
        }
        const unsigned long long mask = __ballot(inside);
        const uint32_t before = (uint32_t)__popcll(mask & ((1ull << lane) - 1ull));
        if (lane == 0) waveTotals[wave] = (uint32_t)__popcll(mask);
        __syncthreads();
        uint32_t waveBase = base, total = 0;
        for (uint32_t w = 0; w < 16u; w++) {
            if (w < wave) waveBase += waveTotals[w];
            total += waveTotals[w];
        }
        if (inside && waveBase + before < capacity) culled[1 + waveBase + before] = instanceIndex;
        __syncthreads();
        if (t == 0) base += total;
        __syncthreads();
    }
    if (t == 0) culled[0] = base;
}

static int launchFrustumCulling(const PassCtx& c) {
    if (int rc = c.needSbuf(0, 16, "sdfCameraFrustumCulling instance buffer")) return rc;
    if (int rc = c.needUbuf(1, sizeof(FrustumUbo), "sdfCameraFrustumCulling frustum buffer")) return rc;
    if (int rc = c.needSbuf(2, 8, "sdfCameraFrustumCulling culled instance buffer")) return rc;
    if (int rc = c.needSbuf(3, sizeof(BoundingBox), "sdfCameraFrustumCulling world bounding boxes")) return rc;
    if (int rc = c.needUbuf(4, 4, "sdfCameraFrustumCulling influence range")) return rc;
    const uint32_t capacity = (uint32_t)(c.sbuf[2].size / 4u) - 1u;
    const uint32_t bbCapacity = (uint32_t)(c.sbuf[3].size / sizeof(BoundingBox));
    // invocations exist for dispatch*64 instances (instanceIndex >= instanceCount return early)
    const uint32_t threadLimit = std::min(c.dispatch[0] * 64u, bbCapacity);
    frustumCullingKernel<<<1, 1024, 0, c.stream>>>((const uint32_t*)c.sbuf[0].ptr, (const FrustumUbo*)c.ubuf[1].ptr, (uint32_t*)c.sbuf[2].ptr,
                                                   (const BoundingBox*)c.sbuf[3].ptr, (const float*)c.ubuf[4].ptr, threadLimit, capacity);
    PLR_CHECK_LAUNCH(c);
    return 0;
}
PLR_REGISTER_SHADER("sdfCameraFrustumCulling.comp", launchFrustumCulling);
PLR_REGISTER_SHADER_FAST("sdfCameraFrustumCulling.comp", launchFrustumCulling); // one kernel serves both math modes (the fused launches below use the same device code)

template <bool USE_HIZ>
__global__ __launch_bounds__(256) void tileCullingKernel(const uint32_t* __restrict__ culled, const BoundingBox* __restrict__ bbs, CulledInstancesPerTile* __restrict__ tiles,
                                                         const float* __restrict__ influenceRangeP, ImgView depthMinMax, const GlobalUbo* __restrict__ g,
                                                         uint32_t tileCountX, uint32_t tileCountY, uint32_t domainX, uint32_t domainY, uint32_t tileRow0,
                                                         uint32_t tileCapacity, uint32_t listCapacity, uint32_t tileCol0) {
    const uint32_t culledInstanceCount = min(culled[0], listCapacity);
    cullTile<USE_HIZ>([&](uint32_t i) { return culled[1 + i]; }, culledInstanceCount, threadIdx.x & 63u, blockIdx.x * 4u + (threadIdx.x >> 6), bbs, tiles, *influenceRangeP,
                      [&](vec2 uv) { return sampleNearest2D<F_RG32F, CLAMP>(depthMinMax, uv); }, g, tileCountX, tileCountY, domainX, domainY, tileRow0, tileCapacity, tileCol0);
}

// ---- pass fusion (backend.h): the two culling passes as one launch (device/culling_device.h)
template <bool USE_HIZ>
__global__ __launch_bounds__(256) void frustumAndTileCullingKernel(FusedCullParams p, ImgView depthMinMax) {
    __shared__ uint32_t list[kFusedCullMaxInstances];
    __shared__ uint32_t waveTotals[4];
    __shared__ uint32_t base;
    frustumAndTileCullingBlock<USE_HIZ, 256>(p, blockIdx.x, gridDim.x, list, waveTotals, &base, [&](vec2 uv) { return sampleNearest2D<F_RG32F, CLAMP>(depthMinMax, uv); });
}

static int launchTileCulling(const PassCtx& c) {
    if (int rc = c.needGlobal()) return rc;
    if (int rc = c.needSbuf(0, 8, "sdfCameraTileCulling culled instance buffer")) return rc;
    if (int rc = c.needSbuf(1, sizeof(BoundingBox), "sdfCameraTileCulling world bounding boxes")) return rc;
    if (int rc = c.needSbuf(2, sizeof(CulledInstancesPerTile), "sdfCameraTileCulling per tile buffer")) return rc;
    if (int rc = c.needUbuf(3, 4, "sdfCameraTileCulling influence range")) return rc;
    const bool useHiZ = c.specBool(0, false);
    if (useHiZ) if (int rc = c.needSampled(4, F_RG32F, "sdfCameraTileCulling depthMinMaxTexture")) return rc;
    if (c.push.size() < 8) return c.fail(-1, "sdfCameraTileCulling: push constant cameraTileCount missing");
    uint32_t tileCount[2];
    std::memcpy(tileCount, c.push.data(), 8);
    // invocations exist for dispatch*8 tiles per axis; tiles beyond cameraTileCount return early
    const PassCtx::RowSpan rs = c.rowSpan((int)tileCount[1]); // tile rows [y0, y1) of the recorded dispatch (8x8 tiles per workgroup)
    const PassCtx::ColSpan cs = c.colSpan((int)tileCount[0]); // and its tile columns (tile rendering)
    const uint32_t tcx = (uint32_t)(cs.x1 - cs.x0), tcy = (uint32_t)(rs.y1 - rs.y0), tileRow0 = (uint32_t)rs.y0, tileCol0 = (uint32_t)cs.x0;
    if (cs.x1 <= cs.x0 || rs.y1 <= rs.y0) return 0;
    const uint32_t tileCapacity = (uint32_t)(c.sbuf[2].size / sizeof(CulledInstancesPerTile));
    const uint32_t listCapacity = (uint32_t)(c.sbuf[0].size / 4u) - 1u;
    const dim3 grid(divUp(tcx * tcy, 4u));
    const ImgView hiz = useHiZ ? c.sampled[4] : ImgView{nullptr, 1, 1, 1, F_RG32F};
    // the uv of the HiZ fetch divides by the push-constant tile count; the dispatch only bounds which tiles run
    if (useHiZ) tileCullingKernel<true><<<grid, 256, 0, c.stream>>>((const uint32_t*)c.sbuf[0].ptr, (const BoundingBox*)c.sbuf[1].ptr, (CulledInstancesPerTile*)c.sbuf[2].ptr,
                                                                   (const float*)c.ubuf[3].ptr, hiz, c.global, tileCount[0], tileCount[1], tcx, tcy, tileRow0, tileCapacity, listCapacity, tileCol0);
    else tileCullingKernel<false><<<grid, 256, 0, c.stream>>>((const uint32_t*)c.sbuf[0].ptr, (const BoundingBox*)c.sbuf[1].ptr, (CulledInstancesPerTile*)c.sbuf[2].ptr,
                                                             (const float*)c.ubuf[3].ptr, hiz, c.global, tileCount[0], tileCount[1], tcx, tcy, tileRow0, tileCapacity, listCapacity, tileCol0);
    PLR_CHECK_LAUNCH(c);
    return 0;
}
PLR_REGISTER_SHADER("sdfCameraTileCulling.comp", launchTileCulling);
PLR_REGISTER_SHADER_FAST("sdfCameraTileCulling.comp", launchTileCulling);

int prepareFusedCulling(const PassCtx& fc, const PassCtx& tc, FusedCullParams* out, bool* useHiZOut, ImgView* hizOut) {
    if (!fc.hasSbuf(0) || !fc.hasUbuf(1) || !fc.hasSbuf(2) || !fc.hasSbuf(3) || !fc.hasUbuf(4) || fc.ubuf[1].size < sizeof(FrustumUbo) || fc.sbuf[2].size < 8) return kUseGeneralKernel;
    if (!tc.global || !tc.hasSbuf(0) || !tc.hasSbuf(1) || !tc.hasSbuf(2) || !tc.hasUbuf(3) || tc.push.size() < 8) return kUseGeneralKernel;
    if (tc.sbuf[0].ptr != fc.sbuf[2].ptr) return kUseGeneralKernel; // the tile pass must read the list the frustum pass writes
    const bool useHiZ = tc.specBool(0, false);
    if (useHiZ && (!tc.hasSampled(4) || tc.sampled[4].fmt != F_RG32F)) return kUseGeneralKernel;
    const uint32_t capacity = (uint32_t)(fc.sbuf[2].size / 4u) - 1u;
    if (capacity > kFusedCullMaxInstances) return kUseGeneralKernel;
    const uint32_t bbCapacity = (uint32_t)(fc.sbuf[3].size / sizeof(BoundingBox));
    const uint32_t threadLimit = std::min(fc.dispatch[0] * 64u, bbCapacity);
    uint32_t tileCount[2];
    std::memcpy(tileCount, tc.push.data(), 8);
    const PassCtx::RowSpan rs = tc.rowSpan((int)tileCount[1]);
    const PassCtx::ColSpan cs = tc.colSpan((int)tileCount[0]);
    const uint32_t tcx = (uint32_t)(cs.x1 - cs.x0), tcy = (uint32_t)(rs.y1 - rs.y0), tileRow0 = (uint32_t)rs.y0;
    if (cs.x1 <= cs.x0 || rs.y1 <= rs.y0) return kUseGeneralKernel; // nothing to cull for: let the frustum pass run on its own
    CullScratch* scratch = (CullScratch*)tc.scratch(sizeof(CullScratch)); // zero-initialised, returned to zero by the kernel
    if (!scratch) return tc.fail(-2, "sdfCameraTileCulling: cannot allocate scratch memory");
    out->instanceBuffer = (const uint32_t*)fc.sbuf[0].ptr; out->frustum = (const FrustumUbo*)fc.ubuf[1].ptr; out->culled = (uint32_t*)fc.sbuf[2].ptr;
    out->bbsFrustum = (const BoundingBox*)fc.sbuf[3].ptr; out->influenceFrustumP = (const float*)fc.ubuf[4].ptr; out->threadLimit = threadLimit; out->capacity = capacity;
    out->scratch = scratch; out->bbs = (const BoundingBox*)tc.sbuf[1].ptr; out->tiles = (CulledInstancesPerTile*)tc.sbuf[2].ptr; out->influenceRangeP = (const float*)tc.ubuf[3].ptr;
    out->g = tc.global; out->tileCountX = tileCount[0]; out->tileCountY = tileCount[1]; out->domainX = tcx; out->domainY = tcy; out->tileRow0 = tileRow0; out->tileCol0 = (uint32_t)cs.x0;
    out->tileCapacity = (uint32_t)(tc.sbuf[2].size / sizeof(CulledInstancesPerTile)); out->listCapacity = (uint32_t)(tc.sbuf[0].size / 4u) - 1u;
    *useHiZOut = useHiZ;
    *hizOut = useHiZ ? tc.sampled[4] : ImgView{nullptr, 1, 1, 1, F_RG32F};
    return 0;
}

static int launchFusedCulling(const PassCtx* const* ctxs, size_t count) {
    if (count != 2) return kUseGeneralKernel;
    const PassCtx& tc = *ctxs[1];
    FusedCullParams p;
    bool useHiZ = false;
    ImgView hiz;
    if (int rc = prepareFusedCulling(*ctxs[0], tc, &p, &useHiZ, &hiz)) return rc;
    const dim3 grid(divUp(p.domainX * p.domainY, 4u));
    if (useHiZ) frustumAndTileCullingKernel<true><<<grid, 256, 0, tc.stream>>>(p, hiz);
    else frustumAndTileCullingKernel<false><<<grid, 256, 0, tc.stream>>>(p, hiz);
    PLR_CHECK_LAUNCH(tc);
    return 0;
}
// ---- pass fusion, per-tile pyramid (band rendering; frames beyond the shader's 11 levels): depthHiZPyramid + depthDownscale + the two culling passes recorded back to
// back as TWO launches - the pyramid's quad blocks, then one grid whose first blocks finish the tiles' levels 4 and 5 (hizTileTailThread) and whose other blocks are
// the culling's. A culling tile needs one texel of level 4, which the tail threads of this very launch are writing: it evaluates that texel itself from level 3
// (the same footprint, the same bits), as the culling blocks of the whole-frame front do (kernels/exposure_tonemap.hip).
__global__ __launch_bounds__(256) void tileTailAndCullingKernel(fasthiz::TileTailParams t, uint32_t tailBlocks, FusedCullParams cull, uint32_t cullBlocks) {
    __shared__ uint32_t list[kFusedCullMaxInstances];
    __shared__ uint32_t waveTotals[4];
    __shared__ uint32_t base;
    if (blockIdx.x < tailBlocks) { fasthiz::hizTileTailThread(t, (int)(blockIdx.x * 256u + threadIdx.x)); return; }
    frustumAndTileCullingBlock<true, 256>(cull, blockIdx.x - tailBlocks, cullBlocks, list, waveTotals, &base, [&](vec2 uv) {
        // sampleNearest2D<F_RG32F, CLAMP> of level 4 at uv (device/image.h), the texel evaluated instead of loaded (depthHiZPyramid.comp:52-124)
        const int x = clampi((int)floorf(saneCoord(uv.x * (float)t.w4)), t.w4), y = clampi((int)floorf(saneCoord(uv.y * (float)t.h4)), t.h4);
        const MinMax m = footprint<false>(2 * x, 2 * y, t.w3, t.h3, t.h3 & 1, t.w3 & 1, [&](int sx, int sy) { return t.level3[(size_t)sy * (size_t)t.w3 + (size_t)sx]; });
        return make_float2(m.mn, m.mx);
    });
}

static int launchTilePyramidAndCulling(const PassCtx* const* ctxs, size_t count) {
    if (count != 4) return kUseGeneralKernel;
    fasthiz::Plan plan;
    if (int rc = fasthiz::prepare(*ctxs[0], ctxs[1], &plan)) return rc;
    if (!plan.perTile) return kUseGeneralKernel; // the whole-frame chain has its own home for the culling blocks (the frame front's launch 2)
    FusedCullParams cull;
    bool useHiZ = false;
    ImgView hiz;
    if (int rc = prepareFusedCulling(*ctxs[2], *ctxs[3], &cull, &useHiZ, &hiz)) return rc;
    const fasthiz::TileTailParams& t = plan.tileTail;
    if (!useHiZ || hiz.ptr != (const void*)t.level4 || hiz.w != t.w4 || hiz.h != t.h4) return kUseGeneralKernel; // the tiles must sample the level this launch finishes
    if (int rc = fasthiz::launchQuadBlocks(*ctxs[0], plan)) return rc;
    const int n = (t.col4End - t.col4Begin) * (t.row4End - t.row4Begin) + (t.col5End - t.col5Begin) * (t.row5End - t.row5Begin);
    const uint32_t tailBlocks = n > 0 ? divUp((unsigned)n, 256u) : 0u, cullBlocks = divUp(cull.domainX * cull.domainY, 4u);
    tileTailAndCullingKernel<<<tailBlocks + cullBlocks, 256, 0, ctxs[0]->stream>>>(t, tailBlocks, cull, cullBlocks);
    PLR_CHECK_LAUNCH(*ctxs[0]);
    return 0;
}
PLR_REGISTER_FUSION("depthHiZPyramid + depthDownscale + sdfCameraFrustumCulling + sdfCameraTileCulling (per-tile pyramid)", launchTilePyramidAndCulling, "depthHiZPyramid.comp",
                    "depthDownscale.comp", "sdfCameraFrustumCulling.comp", "sdfCameraTileCulling.comp");
PLR_REGISTER_FUSION("sdfCameraFrustumCulling + sdfCameraTileCulling", launchFusedCulling, "sdfCameraFrustumCulling.comp", "sdfCameraTileCulling.comp");

// (the ray march's device functions: device/sdf_march_device.h; the exact-set sdfDiffuseTrace.comp: kernels_exact/sdf_trace_exact.hip)


// ------------------------------------------------------------------------------------------------
// sdfDebugVisualisation.comp:73-133 (host Techniques/SDFGI.cpp:334-369): primary rays from the camera through the tile's
// instance list. debugMode 1 lit SDF, 2 camera tile usage, 3 normals, 4 raymarching steps.
template <int MODE>
__global__ __launch_bounds__(256) void sdfDebugVisualisationKernel(ImgView imageOut, const LightBuffer* __restrict__ light, ImgView skyLut,
                                                                   const SdfInstanceBuffer* __restrict__ instanceBuffer, const CulledInstancesPerTile* __restrict__ tiles,
                                                                   const ShadowCascadeInfo* __restrict__ shadowInfo, ImgView shadowMap, const ImgView* __restrict__ bindless,
                                                                   uint32_t bindlessCount, const GlobalUbo* __restrict__ g, int shadowCascadeIndex, int coverW, int coverH,
                                                                   int yBase, uint32_t tileCapacity, uint32_t instanceCapacity) {
    const int px = (int)(blockIdx.x * 64u + (threadIdx.x & 63u));
    const int py = yBase + (int)(blockIdx.y * 4u + (threadIdx.x >> 6));
    if (px >= coverW || py >= coverH) return;
    const vec3 cameraToPixel = -VFromiUV(px, py, g);
    const uint32_t tileIndex = min(tileIndexFromTileUV(px / (int)kCullingTileSize, py / (int)kCullingTileSize, g), tileCapacity - 1u);
    const vec3 rayStart = ld3(g->cameraPosition) + g->nearPlane * cameraToPixel;
    TraceResult tr;
    tr.hit = false;
    tr.closestHitDistance = 10000.f;
    tr.hitPos = vec3(0.f); tr.N = vec3(0.f); tr.albedo = vec3(0.f); tr.hitCount = 0;
    const CulledInstancesPerTile* tile = tiles + tileIndex;
    const uint32_t objectCountRaw = tile->objectCount;
    const int objectCount = (int)min(objectCountRaw, kMaxObjectsPerTile);
    for (int i = 0; i < objectCount; i++) {
        const uint32_t instIndex = min(tile->indices[i], instanceCapacity - 1u);
        const SDFInstance& inst = instanceBuffer->instances[instIndex];
        const ImgView view = bindless[min(inst.sdfTextureIndex, bindlessCount - 1u)];
        traceRayTroughSDFInstance<true>(inst, rayStart, view, cameraToPixel, tr);
    }
    vec3 color(0.f);
    if (tr.hit || MODE == 2) {
        if (MODE == 1) {
            // simpleShadow with the nearest / BLACK-border sampler (sdfDebugVisualisation.comp:101)
            vec4 p = mulMat4(shadowInfo->lightMatrices[shadowCascadeIndex], vec4(tr.hitPos, 1.f));
            p = p / p.w;
            const vec2 xy(p.x * 0.5f + 0.5f, p.y * 0.5f + 0.5f);
            const float actualDepth = gclamp(p.z, 0.f, 1.f);
            const float shadowMapDepth = sampleNearest2D<F_D16, BORDER_BLACK>(shadowMap, xy).x;
            const float shadow = actualDepth > shadowMapDepth ? 1.f : 0.f;
            vec3 sunLight = light->sunStrengthExposed * ld3(light->sunColor);
            sunLight = sunLight * shadow;
            const vec3 ambient(0.15f);
            const float NoL = gclamp(dot(tr.N, ld3(g->sunDirection)), 0.f, 1.f);
            color = tr.albedo * (ambient + sunLight * NoL);
        } else if (MODE == 2) {
            const float percentage = (float)objectCountRaw / (float)kMaxObjectsPerTile;
            color = percentage >= 1.f ? vec3(1.f, 0.f, 0.f) : vec3(percentage);
        } else if (MODE == 3) {
            color = tr.N * 0.5f + 0.5f;
        } else if (MODE == 4) {
            color = vec3((float)tr.hitCount / 128.f);
        }
    } else {
        color = sampleSkyLut(cameraToPixel, skyLut);
    }
    Texel<F_R11G11B10>::store(imageOut.ptr, (size_t)py * (size_t)imageOut.w + px, vec4(color, 1.f));
}

static int launchSdfDebugVisualisation(const PassCtx& c) {
    if (int rc = c.needGlobal()) return rc;
    if (int rc = c.needStorage(0, F_R11G11B10, "sdfDebugVisualisation imageOut")) return rc;
    if (int rc = c.needSbuf(1, sizeof(LightBuffer), "sdfDebugVisualisation lightStorageBuffer")) return rc;
    if (int rc = c.needSampled(2, F_R11G11B10, "sdfDebugVisualisation skyLut")) return rc;
    if (int rc = c.needSbuf(3, 16 + sizeof(SDFInstance), "sdfDebugVisualisation sdfInstanceBuffer")) return rc;
    if (int rc = c.needSbuf(4, sizeof(CulledInstancesPerTile), "sdfDebugVisualisation cameraCulledTileBuffer")) return rc;
    if (int rc = c.needSbuf(6, sizeof(ShadowCascadeInfo), "sdfDebugVisualisation sunShadowInfo")) return rc;
    if (int rc = c.needSampled(7, F_D16, "sdfDebugVisualisation shadowMap")) return rc;
    if (!c.bindless || c.bindlessCount == 0) return c.fail(-4, "sdfDebugVisualisation: global texture array (set 2) is empty");
    const int mode = c.specInt(0, 0), cascade = c.specInt(1, 3);
    if (cascade < 0 || cascade > 3) return c.fail(-1, "sdfDebugVisualisation: shadowCascadeIndex must be 0..3");
    const ImgView& out = c.storage[0];
    const PassCtx::RowSpan rs = c.rowSpan(out.h);
    const int w = std::min((int)(c.dispatch[0] * 8u), out.w), h = rs.y1, y0 = rs.y0;
    if (w <= 0 || h <= y0) return 0;
    const uint32_t tileCapacity = (uint32_t)(c.sbuf[4].size / sizeof(CulledInstancesPerTile));
    const uint32_t instanceCapacity = (uint32_t)((c.sbuf[3].size - 16u) / sizeof(SDFInstance));
    const dim3 grid(divUp((unsigned)w, 64u), divUp((unsigned)(h - y0), 4u));
#define PLR_DBG_ARGS out, (const LightBuffer*)c.sbuf[1].ptr, c.sampled[2], (const SdfInstanceBuffer*)c.sbuf[3].ptr, (const CulledInstancesPerTile*)c.sbuf[4].ptr, \
                     (const ShadowCascadeInfo*)c.sbuf[6].ptr, c.sampled[7], c.bindless, c.bindlessCount, c.global, cascade, w, h, y0, tileCapacity, instanceCapacity
    switch (mode) {
        case 1: sdfDebugVisualisationKernel<1><<<grid, 256, 0, c.stream>>>(PLR_DBG_ARGS); break;
        case 2: sdfDebugVisualisationKernel<2><<<grid, 256, 0, c.stream>>>(PLR_DBG_ARGS); break;
        case 3: sdfDebugVisualisationKernel<3><<<grid, 256, 0, c.stream>>>(PLR_DBG_ARGS); break;
        case 4: sdfDebugVisualisationKernel<4><<<grid, 256, 0, c.stream>>>(PLR_DBG_ARGS); break;
        default: sdfDebugVisualisationKernel<0><<<grid, 256, 0, c.stream>>>(PLR_DBG_ARGS); break; // None: hits stay black, misses show the sky
    }
#undef PLR_DBG_ARGS
    PLR_CHECK_LAUNCH(c);
    return 0;
}
PLR_REGISTER_SHADER("sdfDebugVisualisation.comp", launchSdfDebugVisualisation);
// Also what PLR_MATH_FAST runs: a debug view's job is to show what the march DOES (mode 4 colours by step count, mode 3 by the SDF's normal, mode 2 by the tile
// lists): the exact-order march is the instrument, a restructured one would visualise itself. It is not on the frame's path (SDFGI.cpp:334-369 replaces the frame).
PLR_REGISTER_SHADER_FAST("sdfDebugVisualisation.comp", launchSdfDebugVisualisation);

} // namespace plr
