// Camera culling of the SDF instances (sdfCameraFrustumCulling.comp, sdfCameraTileCulling.comp, sdfCulling.inc): the device code shared by
// kernels/sdfgi.hip (the two passes on their own, and fused as a pair) and kernels/exposure_tonemap.hip (launch 2 of the fused frame front hosts the
// culling blocks as well, kernels_fast/fused_front.h). The outputs are integer lists that must be the oracle's exactly, so this header is only
// included by files built with the exact set's flags (no FMA contraction, IEEE divide / sqrt).
#pragma once
#include "../backend.h"
#include "shading_common.h"

namespace plr {

// sdfCameraFrustumCulling.comp:36-62
struct FrustumUbo { float points[6][4]; float normals[6][4]; };
struct CulledList { uint32_t count; uint32_t indices[1]; };

// the shader's test for one instance (:44-58)
PLR_DI bool insideFrustum(const BoundingBox& bb, const FrustumUbo* __restrict__ frustum, float influenceRange) {
    const vec3 bbMin = ld3(bb.bbMin), bbMax = ld3(bb.bbMax);
    const vec3 center = (bbMax + bbMin) * 0.5f;
    const vec3 ext = bbMax - bbMin;
    float radius = gmax(gmax(ext.x, ext.y), ext.z) * 0.5f;
    radius += influenceRange;
    bool inside = true;
    for (int i = 0; i < 6; i++) {
        const bool outsidePlane = dot(center - ld3(frustum->points[i]), ld3(frustum->normals[i])) > radius;
        inside = inside && !outsidePlane;
    }
    return inside;
}

// ------------------------------------------------------------------------------------------------
// sdfCulling.inc:17-20: tile stride from the FULL screen resolution (reproduced as is)
PLR_DI uint32_t tileIndexFromTileUV(int tx, int ty, const GlobalUbo* g) {
    const uint32_t tileCountX = (uint32_t)ceilf((float)g->screenResolution[0] / (float)kCullingTileSize);
    return (uint32_t)tx + (uint32_t)ty * tileCountX;
}

PLR_DI vec3 VFromiUV(int x, int y, const GlobalUbo* g) {
    const vec2 pixelCoor(((float)x / (float)g->screenResolution[0] - 0.5f) * 2.f, ((float)y / (float)g->screenResolution[1] - 0.5f) * 2.f);
    return calculateViewDirectionFromPixel(pixelCoor, ld3(g->cameraForward), ld3(g->cameraUp), ld3(g->cameraRight), g->cameraTanFovHalf, g->cameraAspectRatio);
}

// sdfCameraTileCulling.comp:42-99, one wave per tile: 64 instances are tested per step and appended in list order.
// list(i) returns entry i of the frustum-culled list, listCount its length; depthAt(uv) the depth pyramid's (min, max) at uv (USE_HIZ only).
template <bool USE_HIZ, class List, class DepthAt>
PLR_DI void cullTile(List list, uint32_t listCount, uint32_t lane, uint32_t tileLinear, const BoundingBox* __restrict__ bbs, CulledInstancesPerTile* __restrict__ tiles,
                     float influenceRange, DepthAt depthAt, const GlobalUbo* __restrict__ g, uint32_t tileCountX, uint32_t tileCountY, uint32_t domainX,
                     uint32_t domainY, uint32_t tileRow0, uint32_t tileCapacity, uint32_t tileCol0 = 0u) {
    if (tileLinear >= domainX * domainY) return;
    // tiles [tileCol0, tileCol0 + domainX) x [tileRow0, tileRow0 + domainY): the recorded dispatch (band rendering: its tile rows, tile rendering: its columns too)
    const int tx = (int)(tileCol0 + tileLinear % domainX), ty = (int)(tileRow0 + tileLinear / domainX);
    const uint32_t tileIndex = tileIndexFromTileUV(tx, ty, g);
    if (tileIndex >= tileCapacity) return;
    CulledInstancesPerTile* tile = tiles + tileIndex;
    const int ts = (int)kCullingTileSize;
    const vec3 cameraToPixel = -VFromiUV(tx * ts + ts / 2, ty * ts + ts / 2, g);
    vec3 V_ll = -VFromiUV(tx * ts, ty * ts, g);
    vec3 V_ur = -VFromiUV(tx * ts + ts, ty * ts + ts, g);
    V_ll /= dot(cameraToPixel, V_ll);
    V_ur /= dot(cameraToPixel, V_ur);
    const float coneRadiusPerMeter = distance(V_ll, V_ur) * 0.5f;
    float depthMin = g->nearPlane, depthMax = g->farPlane;
    if (USE_HIZ) {
        const vec2 uv((float)tx / (float)tileCountX, (float)ty / (float)tileCountY);
        const auto mm = depthAt(uv); // (min, max) of the tile's texel of the depth pyramid, nearest + clamp-to-edge
        depthMin = linearizeDepth(mm.y, g->nearPlane, g->farPlane);
        depthMax = linearizeDepth(mm.x, g->nearPlane, g->farPlane);
    }
    const vec3 camFwd = ld3(g->cameraForward), camPos = ld3(g->cameraPosition);
    depthMin *= dot(cameraToPixel, camFwd);
    depthMax *= dot(cameraToPixel, camFwd);
    uint32_t count = 0;
    for (uint32_t chunk = 0; chunk < listCount && count < kMaxObjectsPerTile; chunk += 64u) {
        const uint32_t i = chunk + lane;
        bool pass = false;
        uint32_t inst = 0;
        if (i < listCount) {
            inst = list(i);
            const BoundingBox bb = bbs[inst];
            const vec3 bbMin = ld3(bb.bbMin), bbMax = ld3(bb.bbMax);
            const vec3 center = (bbMax + bbMin) * 0.5f;
            const vec3 ext = (bbMax - bbMin) * 0.5f;
            float radius = gmax(gmax(ext.x, ext.y), ext.z);
            radius += influenceRange;
            float projection = dot(center - camPos, cameraToPixel);
            projection = gclamp(projection, depthMin, depthMax);
            const float d = distance(center, projection * cameraToPixel + camPos);
            pass = d < radius + coneRadiusPerMeter * projection;
        }
        const unsigned long long mask = __ballot(pass);
        const uint32_t pos = count + (uint32_t)__popcll(mask & ((1ull << lane) - 1ull));
        if (pass && pos < kMaxObjectsPerTile) tile->indices[pos] = inst;
        count = min(count + (uint32_t)__popcll(mask), kMaxObjectsPerTile);
    }
    if (lane == 0) tile->objectCount = count;
}

// ---- pass fusion (backend.h): sdfCameraFrustumCulling + sdfCameraTileCulling recorded back to back, as one launch. Every block repeats the
// (tiny) frustum test of all instances into LDS with the same ordered compaction and culls its tiles (one per wave) against that list; block 0 also
// stores the list, and the block that takes the last ticket stores its length - after every block has read the initial length.
constexpr uint32_t kFusedCullMaxInstances = 4096;
struct CullScratch { uint32_t ticket; };
struct FusedCullParams {
    const uint32_t* instanceBuffer; const FrustumUbo* frustum; uint32_t* culled; const BoundingBox* bbsFrustum; const float* influenceFrustumP;
    uint32_t threadLimit, capacity;
    CullScratch* scratch;
    const BoundingBox* bbs; CulledInstancesPerTile* tiles; const float* influenceRangeP;
    const GlobalUbo* g;
    uint32_t tileCountX, tileCountY, domainX, domainY, tileRow0, tileCapacity, listCapacity;
    uint32_t tileCol0; // first tile column of the dispatch (tile rendering)
};
// block `block` of `blocks`, NT threads; list: kFusedCullMaxInstances words of LDS, waveTotals: NT / 64 words, base: one word
template <bool USE_HIZ, uint32_t NT, class DepthAt>
PLR_DI void frustumAndTileCullingBlock(const FusedCullParams& p, uint32_t block, uint32_t blocks, uint32_t* list, uint32_t* waveTotals, uint32_t* base, DepthAt depthAt) {
    constexpr uint32_t kWaves = NT / 64u;
    const uint32_t t = threadIdx.x, lane = t & 63u, wave = t >> 6;
    const uint32_t instanceCount = min(p.instanceBuffer[0], p.threadLimit);
    const float influenceFrustum = *p.influenceFrustumP;
    uint32_t* __restrict__ culled = p.culled;
    const uint32_t base0 = culled[0]; // entries the list already holds (the host zeroes the count every frame)
    if (t == 0) *base = base0;
    __syncthreads();
    for (uint32_t chunk = 0; chunk < instanceCount; chunk += NT) {
        const uint32_t instanceIndex = chunk + t;
        const bool inside = instanceIndex < instanceCount && insideFrustum(p.bbsFrustum[instanceIndex], p.frustum, influenceFrustum);
        const unsigned long long mask = __ballot(inside);
        const uint32_t before = (uint32_t)__popcll(mask & ((1ull << lane) - 1ull));
        if (lane == 0) waveTotals[wave] = (uint32_t)__popcll(mask);
        __syncthreads();
        uint32_t waveBase = *base, total = 0;
        for (uint32_t w = 0; w < kWaves; w++) {
            if (w < wave) waveBase += waveTotals[w];
            total += waveTotals[w];
        }
        const uint32_t pos = waveBase + before;
        if (inside && pos < p.capacity) {
            if (pos < kFusedCullMaxInstances) list[pos] = instanceIndex;
            if (block == 0) culled[1 + pos] = instanceIndex;
        }
        __syncthreads();
        if (t == 0) *base += total;
        __syncthreads();
    }
    const uint32_t finalCount = *base;
    const uint32_t listCount = min(finalCount, p.listCapacity);
    // entries below base0 were in the global list before this launch (block 0 does not touch them)
    cullTile<USE_HIZ>([&](uint32_t i) { return i < base0 ? culled[1 + i] : list[i]; }, listCount, lane, block * kWaves + wave, p.bbs, p.tiles, *p.influenceRangeP, depthAt, p.g,
                      p.tileCountX, p.tileCountY, p.domainX, p.domainY, p.tileRow0, p.tileCapacity, p.tileCol0);
    __syncthreads();
    // every block read culled[0] (base0) before this barrier; the ticket is release / acquire at agent scope so that the last block's store of
    // the new count is ordered after all of those reads
    if (t == 0 && __hip_atomic_fetch_add(&p.scratch->ticket, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT) == blocks - 1u) {
        culled[0] = finalCount;
        __hip_atomic_store(&p.scratch->ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// validates a (sdfCameraFrustumCulling, sdfCameraTileCulling) pair of executions recorded back to back and fills the parameters; *useHiZ / *hiz: the
// tile pass's depth pyramid level. 0 / kUseGeneralKernel / < 0 (kernels/sdfgi.hip)
int prepareFusedCulling(const PassCtx& fc, const PassCtx& tc, FusedCullParams* out, bool* useHiZ, ImgView* hiz);

} // namespace plr
