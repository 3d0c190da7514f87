// sdfDiffuseTrace.comp for gfx950, PLR_MATH_EXACT set (libplr_exact.so): the reference's operation order (+ SDF.inc, sdfCulling.inc, sampling.inc,
// sunShadowCascades.inc, sky.inc, SphericalHarmonics.inc); host side Techniques/SDFGI.cpp:380-419. The benchmarked kernel is kernels_fast/sdf_trace_fast.hip.
//
// Mapping to CDNA4: one wave64 is one 8x8 reference workgroup (the shared-memory ray exchange of resolveColor becomes a
// per-wave LDS slab), four waves of a block share one 32x32-px culling tile so the culled instance list and the 96-byte
// SDFInstance records are wave-uniform scalar loads. SDF volumes (64^3 half floats, 512 KiB each) are fetched with explicit
// trilinear address math; 256 of them (134 MB) sit in the 256 MB Infinity Cache after the first touch.
#include "../backend.h"
#include "../device/shading_common.h"
#include "../device/culling_device.h"
#include "../device/sdf_march_device.h"

namespace plr {

template <bool STRICT_CUTOFF>
__global__ __launch_bounds__(256) void sdfDiffuseTraceKernel(ImgView outYSH, ImgView outCoCg, ImgView depthTexture, ImgView normalTexture, ImgView skyLut,
                                                             const LightBuffer* __restrict__ light, const SdfInstanceBuffer* __restrict__ instanceBuffer,
                                                             const CulledInstancesPerTile* __restrict__ tiles, const float* __restrict__ influenceRangeP,
                                                             const ShadowCascadeInfo* __restrict__ shadowInfo, ImgView shadowMap, const ImgView* __restrict__ bindless,
                                                             uint32_t bindlessCount, const GlobalUbo* __restrict__ g, int shadowCascadeIndex, int groupsX, int groupsY, int groupY0, int groupX0,
                                                             uint32_t tileCapacity, uint32_t instanceCapacity) {
    __shared__ RayInfo sharedRays[4][64];
    const int wave = (int)(threadIdx.x >> 6), lane = (int)(threadIdx.x & 63u);
    // one wave = one 8x8 reference workgroup; the four waves of a block are a 2x2 arrangement inside one culling tile
    const int gx = groupX0 + (int)blockIdx.x * 2 + (wave & 1), gy = groupY0 + (int)blockIdx.y * 2 + (wave >> 1); // workgroups [groupX0, groupsX) x [groupY0, groupsY)
    const bool active = gx < groupsX && gy < groupsY;
    const int lx = lane & 7, ly = lane >> 3;
    const int px = gx * 8 + lx, py = gy * 8 + ly;
    vec3 L(0.f, 0.f, 1.f);
    RayInfo mine{0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (active) {
        const vec2 uv((float)px / (float)outYSH.w, (float)py / (float)outYSH.h);
        const float depth = sampleNearest2D<F_D32, CLAMP>(depthTexture, uv).x;
        const float depthLinear = linearizeDepth(depth, g->nearPlane, g->farPlane);
        const vec2 pixelNDC(uv.x * 2.f - 1.f, uv.y * 2.f - 1.f);
        const vec3 camFwd = ld3(g->cameraForward);
        const vec3 V = -calculateViewDirectionFromPixel(pixelNDC, camFwd, ld3(g->cameraUp), ld3(g->cameraRight), g->cameraTanFovHalf, g->cameraAspectRatio);
        const vec3 pWorld = ld3(g->cameraPosition) + V / dot(V, camFwd) * depthLinear;

        const uint32_t noiseSlot = (uint32_t)g->noiseTextureIndices[g->frameIndexMod4 & 3u];
        const ImgView noiseTex = bindless[min(noiseSlot, bindlessCount - 1u)];
        const vec2 noiseUV((float)px / (float)noiseTex.w, (float)py / (float)noiseTex.h);
        const vec4 nz = sampleNearest2D<F_RG8, REPEAT>(noiseTex, noiseUV);
        const vec2 xi(nz.x, nz.y);
        const vec3 normalTexel = sampleNearest2D<F_RGBA8, CLAMP>(normalTexture, uv).xyz();
        const vec3 N = normalTexel * 2.f - 1.f;
        mine.nx = N.x; mine.ny = N.y; mine.nz = N.z; mine.depth = depthLinear;
        const vec3 rayOrigin = pWorld + N * 0.2f;
        L = importanceSampleCosine(xi, N);

        TraceResult tr;
        tr.hit = false;
        tr.closestHitDistance = 10000.f;
        tr.hitPos = vec3(0.f);
        tr.albedo = vec3(0.f);
        // tileUV = gl_WorkGroupID.xy / (cullingTileSize / 8); wave uniform
        const uint32_t tileIndex = min(tileIndexFromTileUV(gx / (int)(kCullingTileSize / 8u), gy / (int)(kCullingTileSize / 8u), g), tileCapacity - 1u);
        const CulledInstancesPerTile* tile = tiles + tileIndex;
        const int objectCount = (int)min(tile->objectCount, kMaxObjectsPerTile);
        for (int i = 0; i < objectCount; i++) {
            const uint32_t instIndex = min((uint32_t)__builtin_amdgcn_readfirstlane((int)tile->indices[i]), instanceCapacity - 1u);
            const SDFInstance& inst = instanceBuffer->instances[instIndex];
            const uint32_t texIndex = min((uint32_t)__builtin_amdgcn_readfirstlane((int)inst.sdfTextureIndex), bindlessCount - 1u);
            const ImgView sdf = bindless[texIndex];
            traceRayTroughSDFInstance(inst, rayOrigin, sdf, L, tr);
        }
        vec3 hitColor;
        if (tr.hit) {
            const float shadow = simpleShadow(tr.hitPos, shadowInfo->lightMatrices[shadowCascadeIndex], shadowMap);
            const vec3 sunLight = shadow * light->sunStrengthExposed * ld3(light->sunColor);
            hitColor = tr.albedo * sunLight;
            bool hitInRange = tr.closestHitDistance < *influenceRangeP;
            hitInRange = hitInRange || !STRICT_CUTOFF;
            const bool selfIntersection = tr.closestHitDistance < 0.0001f;
            if (!hitInRange || selfIntersection) hitColor = vec3(0.f);
        } else {
            hitColor = sampleSkyLut(L, skyLut);
        }
        mine.cr = hitColor.x; mine.cg = hitColor.y; mine.cb = hitColor.z;
    }
    sharedRays[wave][lane] = mine;
    __syncthreads();
    if (!active) return;

    // resolveColor (:70-116); sharedRays[x][y] of the reference = slab[y * 8 + x]
    float weightTotal = 1.f;
    vec3 color(mine.cr, mine.cg, mine.cb);
    const vec3 myN(mine.nx, mine.ny, mine.nz);
    for (int x = -1; x <= 1; x++)
        for (int y = -1; y <= 1; y++) {
            if (x == 0 && y == 0) continue;
            const int rx = lx + x, ry = ly + y;
            const bool isValidIndex = (rx > 0 && ry > 0) && (rx < 8 && ry < 8); // sic: > 0 (:88)
            if (!isValidIndex) continue;
            const RayInfo nb = sharedRays[wave][ry * 8 + rx];
            const float NoN = gclamp(dot(myN, vec3(nb.nx, nb.ny, nb.nz)), 0.f, 1.f);
            const bool normalsMatch = NoN > 0.9f;
            const bool depthMatch = fabsf(mine.depth - nb.depth) < 0.5f;
            if (normalsMatch && depthMatch) {
                const float weight = (x == 0 ? 1.f : 0.5f) * (y == 0 ? 1.f : 0.5f);
                color += weight * vec3(nb.cr, nb.cg, nb.cb);
                weightTotal += weight;
            }
        }
    color /= weightTotal;
    const vec3 YCoCg = linearToYCoCg(color);
    if (px < outYSH.w && py < outYSH.h) {
        const vec4 sh = directionToSH_L1(L);
        // result_Y_SH = vec4(0) + YCoCg.x * SH
        const vec4 ysh = vec4(0.f) + YCoCg.x * sh;
        const size_t idx = (size_t)py * (size_t)outYSH.w + px;
        Texel<F_RGBA16F>::store(outYSH.ptr, idx, ysh);
        Texel<F_RG16F>::store(outCoCg.ptr, idx, vec4(0.f + YCoCg.y, 0.f + YCoCg.z, 0.f, 0.f));
    }
}

static int launchSdfDiffuseTrace(const PassCtx& c) {
    if (int rc = c.needGlobal()) return rc;
    if (int rc = c.needStorage(0, F_RGBA16F, "sdfDiffuseTrace imageOut_Y_SH")) return rc;
    if (int rc = c.needStorage(1, F_RG16F, "sdfDiffuseTrace imageOut_CoCg")) return rc;
    if (int rc = c.needSampled(2, F_D32, "sdfDiffuseTrace depthTexture")) return rc;
    if (int rc = c.needSampled(3, F_RGBA8, "sdfDiffuseTrace normalTexture")) return rc;
    if (int rc = c.needSampled(4, F_R11G11B10, "sdfDiffuseTrace skyLut")) return rc;
    if (int rc = c.needSbuf(5, sizeof(LightBuffer), "sdfDiffuseTrace lightBuffer")) return rc;
    if (int rc = c.needSbuf(6, 16 + sizeof(SDFInstance), "sdfDiffuseTrace sdfInstanceBuffer")) return rc;
    if (int rc = c.needSbuf(7, sizeof(CulledInstancesPerTile), "sdfDiffuseTrace cameraCulledTileBuffer")) return rc;
    if (int rc = c.needUbuf(8, 4, "sdfDiffuseTrace influenceRangeBuffer")) return rc;
    if (int rc = c.needSbuf(9, sizeof(ShadowCascadeInfo), "sdfDiffuseTrace sunShadowInfo")) return rc;
    if (int rc = c.needSampled(10, F_D16, "sdfDiffuseTrace shadowMap")) return rc;
    if (!c.bindless || c.bindlessCount == 0) return c.fail(-4, "sdfDiffuseTrace: global texture array (set 2) is empty");
    const bool strict = c.specBool(0, false);
    const int cascade = c.specInt(1, 3);
    if (cascade < 0 || cascade > 3) return c.fail(-1, "sdfDiffuseTrace: shadowCascadeIndex must be 0..3");
    const ImgView& out = c.storage[0];
    if (c.storage[1].w != out.w || c.storage[1].h != out.h) return c.fail(-4, "sdfDiffuseTrace: Y_SH and CoCg targets differ in size");
    // workgroup rows [groupY0, groupsY) of the recorded dispatch; a block is 2x2 workgroups inside one culling tile
    const int groupX0 = (int)c.base[0], groupsX = groupX0 + (int)c.dispatch[0], groupY0 = (int)c.base[1], groupsY = groupY0 + (int)c.dispatch[1];
    if (groupsX <= groupX0 || groupsY <= groupY0) return 0;
    if ((groupY0 & 1) || (groupX0 & 1)) return c.fail(-1, "sdfDiffuseTrace: dispatch base must be a multiple of 2 workgroups");
    const uint32_t tileCapacity = (uint32_t)(c.sbuf[7].size / sizeof(CulledInstancesPerTile));
    const uint32_t instanceCapacity = (uint32_t)((c.sbuf[6].size - 16u) / sizeof(SDFInstance));
    const dim3 grid(divUp((unsigned)(groupsX - groupX0), 2u), divUp((unsigned)(groupsY - groupY0), 2u));
#define PLR_TRACE_ARGS c.storage[0], c.storage[1], c.sampled[2], c.sampled[3], c.sampled[4], (const LightBuffer*)c.sbuf[5].ptr,                       \
                       (const SdfInstanceBuffer*)c.sbuf[6].ptr, (const CulledInstancesPerTile*)c.sbuf[7].ptr, (const float*)c.ubuf[8].ptr,            \
                       (const ShadowCascadeInfo*)c.sbuf[9].ptr, c.sampled[10], c.bindless, c.bindlessCount, c.global, cascade, groupsX, groupsY, groupY0, groupX0, \
                       tileCapacity, instanceCapacity
    if (strict) sdfDiffuseTraceKernel<true><<<grid, 256, 0, c.stream>>>(PLR_TRACE_ARGS);
    else sdfDiffuseTraceKernel<false><<<grid, 256, 0, c.stream>>>(PLR_TRACE_ARGS);
#undef PLR_TRACE_ARGS
    PLR_CHECK_LAUNCH(c);
    return 0;
}
PLR_REGISTER_SHADER("sdfDiffuseTrace.comp", launchSdfDiffuseTrace);

} // namespace plr
