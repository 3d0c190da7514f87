// ORACLE (test infrastructure, never shipped, never on the product path).
// Restates: resources/shaders/depthDownscale.comp, sdfCameraFrustumCulling.comp, sdfCameraTileCulling.comp (+ sdfCulling.inc),
// sdfDiffuseTrace.comp (+ SDF.inc, sampling.inc, sunShadowCascades.inc, sky.inc:85-116, SphericalHarmonics.inc),
// filterIndirectDiffuseSpatial.comp, filterIndirectDiffuseTemporal.comp, indirectLightUpscale.comp.
#include "common.h"

using namespace orc;

// depthDownscale.comp:12-20, grid ceil(halfRes/8) x 8x8
extern "C" void orc_depth_downscale(const orc_image* srcP, const orc_image* dstP) {
    const Image &fullResSrc = img(srcP), &halfResDst = img(dstP);
    parallelFor(halfResDst.h, [&](int y0, int y1) {
        for (int y = y0; y < y1; y++)
            for (int x = 0; x < halfResDst.w; x++) {
                const vec2 texelSize = 1.f / vec2((float)fullResSrc.w, (float)fullResSrc.h);
                const vec2 uv = (vec2((float)(x * 2), (float)(y * 2)) + 0.5f) * texelSize;
                const float depth = texture2D(fullResSrc, NEAREST, CLAMP, uv).x;
                imageStore(halfResDst, ivec2(x, y), vec4(depth, 0, 0, 0));
            }
    });
}

// sdfCameraFrustumCulling.comp:36-62. The reference appends with atomicAdd, so the order of the compacted list is
// nondeterministic there; ascending instance index (one of its possible outcomes) is the order defined here.
extern "C" void orc_sdf_camera_frustum_culling(uint32_t instanceCount, const float* frustumPoints, const float* frustumNormals, const float* worldBBs,
                                               float influenceRange, uint32_t* culled) {
    for (uint32_t instanceIndex = 0; instanceIndex < instanceCount; instanceIndex++) {
        const vec3 bbMin = v3(worldBBs + instanceIndex * 8), bbMax = v3(worldBBs + instanceIndex * 8 + 4);
        const vec3 boundingSphereCenter = (bbMax + bbMin) * 0.5f;
        const vec3 bbExtends = (bbMax - bbMin);
        float boundingSphereRadius = gmax(gmax(bbExtends.x, bbExtends.y), bbExtends.z) * 0.5f;
        boundingSphereRadius += influenceRange;
        bool isInsideFrustum = true;
        for (int i = 0; i < 6; i++) {
            const vec3 frustumPoint = v3(frustumPoints + i * 4), frustumNormal = v3(frustumNormals + i * 4);
            const bool isOutsidePlane = dot(boundingSphereCenter - frustumPoint, frustumNormal) > boundingSphereRadius;
            isInsideFrustum = isInsideFrustum && !isOutsidePlane;
        }
        if (isInsideFrustum) {
            const uint32_t indexBufferIndex = culled[0]++;
            culled[1 + indexBufferIndex] = instanceIndex;
        }
    }
}

static const uint32_t cullingTileSize = 32;       // sdfCulling.inc:4
static const uint32_t maxObjectsPerTile = 100;    // sdfCulling.inc:5
static const uint32_t tileStrideUints = 1 + maxObjectsPerTile;

// sdfCulling.inc:17-20 (stride from the FULL screen resolution, whatever the trace resolution is)
static uint32_t tileIndexFromTileUV(ivec2 tileUV, const orc_global* g) {
    const uint32_t tileCountX = (uint32_t)std::ceil((float)g->screenResolution[0] / (float)cullingTileSize);
    return (uint32_t)tileUV.x + (uint32_t)tileUV.y * tileCountX;
}

static vec3 VFromiUV(ivec2 iUV, const orc_global* g) {
    const vec2 pixelCoor = (toVec2(iUV) / vec2((float)g->screenResolution[0], (float)g->screenResolution[1]) - 0.5f) * 2.f;
    return calculateViewDirectionFromPixel(pixelCoor, v3(g->cameraForward), v3(g->cameraUp), v3(g->cameraRight), g->cameraTanFovHalf, g->cameraAspectRatio);
}

// sdfCameraTileCulling.comp:42-99
extern "C" void orc_sdf_camera_tile_culling(const uint32_t* culled, const float* worldBBs, uint32_t* tiles, float influenceRange,
                                            const orc_image* depthMinMaxMip, const orc_global* g, int32_t useHiZ, uint32_t tileCountX, uint32_t tileCountY) {
    const uint32_t culledInstanceCount = culled[0];
    const vec3 camPos = v3(g->cameraPosition), camFwd = v3(g->cameraForward);
    for (uint32_t ty = 0; ty < tileCountY; ty++)
        for (uint32_t tx = 0; tx < tileCountX; tx++) {
            const ivec2 tileUV((int)tx, (int)ty);
            const uint32_t tileIndex = tileIndexFromTileUV(tileUV, g);
            uint32_t* tile = tiles + (size_t)tileIndex * tileStrideUints;
            tile[0] = 0;
            const int ts = (int)cullingTileSize;
            const vec3 cameraToPixel = -VFromiUV(ivec2(tileUV.x * ts + ts / 2, tileUV.y * ts + ts / 2), g);
            vec3 V_ll = -VFromiUV(ivec2(tileUV.x * ts, tileUV.y * ts), g);
            vec3 V_ur = -VFromiUV(ivec2(tileUV.x * ts + ts, tileUV.y * ts + ts), g);
            V_ll /= dot(cameraToPixel, V_ll);
            V_ur /= dot(cameraToPixel, V_ur);
            const float coneRadiusPerMeter = distance(V_ll, V_ur) * 0.5f;
            float depthMin = g->nearPlane;
            float depthMax = g->farPlane;
            const vec2 uv = toVec2(tileUV) / vec2((float)tileCountX, (float)tileCountY);
            if (useHiZ) {
                const vec4 depthMinMax = texture2D(img(depthMinMaxMip), NEAREST, CLAMP, uv);
                depthMin = linearizeDepth(depthMinMax.y, g->nearPlane, g->farPlane);
                depthMax = linearizeDepth(depthMinMax.x, g->nearPlane, g->farPlane);
            }
            depthMin *= dot(cameraToPixel, camFwd);
            depthMax *= dot(cameraToPixel, camFwd);
            for (uint32_t i = 0; i < culledInstanceCount; i++) {
                const uint32_t inst = culled[1 + i];
                const vec3 bbMin = v3(worldBBs + inst * 8), bbMax = v3(worldBBs + inst * 8 + 4);
                if (tile[0] >= maxObjectsPerTile) break;
                const vec3 boundingSphereCenter = (bbMax + bbMin) * 0.5f;
                const vec3 bbExtends = (bbMax - bbMin) * 0.5f;
                float boundingSphereRadius = gmax(gmax(bbExtends.x, bbExtends.y), bbExtends.z);
                boundingSphereRadius += influenceRange;
                float projection = dot(boundingSphereCenter - camPos, cameraToPixel);
                projection = gclamp(projection, depthMin, depthMax);
                const float d = distance(boundingSphereCenter, projection * cameraToPixel + camPos);
                if (d < boundingSphereRadius + coneRadiusPerMeter * projection) {
                    tile[1 + tile[0]] = inst;
                    tile[0]++;
                }
            }
        }
}

// ------------------------------------------------------------------------------------------------ trace
namespace {

struct TraceResult {
    bool hit;
    float closestHitDistance;
    vec3 hitPos, N;
    int hitCount;
    vec3 albedo;
};

float sampleSDF(vec3 uv, const Image& sdf) { return texture3D(sdf, LINEAR, CLAMP, uv).x; }

// SDF.inc:16-25
vec3 normalFromSDF(vec3 uv, vec3 extends, const Image& sdf) {
    const float extendsMax = gmax(extends.x, gmax(extends.y, extends.z));
    const vec3 extendsNormalized = extends / extendsMax;
    const vec3 epsilon = vec3(0.15f) / vec3((float)sdf.w, (float)sdf.h, (float)sdf.d) / extendsNormalized;
    return normalize(vec3(sampleSDF(uv + vec3(epsilon.x, 0, 0), sdf) - sampleSDF(uv - vec3(epsilon.x, 0, 0), sdf),
                          sampleSDF(uv + vec3(0, epsilon.y, 0), sdf) - sampleSDF(uv - vec3(0, epsilon.y, 0), sdf),
                          sampleSDF(uv + vec3(0, 0, epsilon.z), sdf) - sampleSDF(uv - vec3(0, 0, epsilon.z), sdf)));
}

bool isPointInAABB(vec3 p, vec3 mn, vec3 mx) { return p.x >= mn.x && p.y >= mn.y && p.z >= mn.z && p.x <= mx.x && p.y <= mx.y && p.z <= mx.z; }

struct HitResult { bool hit; float t; };

// SDF.inc:42-86
HitResult rayAABBIntersection(vec3 o, vec3 dir, vec3 mn, vec3 mx) {
    HitResult result{false, 100000.f};
    float intersection = o.x < 0.f ? mn.x : mx.x;
    const float tx = (intersection - o.x) / dir.x;
    vec3 p = o + tx * dir;
    if (tx > 0.f && p.y >= mn.y && p.y <= mx.y && p.z >= mn.z && p.z <= mx.z) { result.t = gmin(result.t, tx); result.hit = true; }
    intersection = o.y < 0.f ? mn.y : mx.y;
    const float ty = (intersection - o.y) / dir.y;
    p = o + ty * dir;
    if (ty > 0.f && p.x >= mn.x && p.x <= mx.x && p.z >= mn.z && p.z <= mx.z) { result.t = gmin(result.t, ty); result.hit = true; }
    intersection = o.z < 0.f ? mn.z : mx.z;
    const float tz = (intersection - o.z) / dir.z;
    p = o + tz * dir;
    if (tz > 0.f && p.x >= mn.x && p.x <= mx.x && p.y >= mn.y && p.y <= mx.y) { result.t = gmin(result.t, tz); result.hit = true; }
    return result;
}

// SDF.inc:101-184
void traceRayTroughSDFInstance(const orc_sdf_instance& instance, vec3 rayStartWorld, const Image& sdf, vec3 rayDirectionWorld, TraceResult& tr) {
    const mat4 worldToLocal = toMat4(instance.worldToLocal);
    const vec3 localExtends = v3(instance.localExtends);
    vec3 rayStartLocal = (worldToLocal * vec4(rayStartWorld, 1.f)).xyz();
    const vec3 rayEndLocal = (worldToLocal * vec4(rayStartWorld + rayDirectionWorld, 1.f)).xyz();
    vec3 rayDirection = rayEndLocal - rayStartLocal;
    rayDirection /= length(rayDirection);
    const vec3 sdfMaxLocal = localExtends * 0.5f;
    const vec3 sdfMinLocal = -sdfMaxLocal;
    float hitDistanceLocal = 0.f;
    if (!isPointInAABB(rayStartLocal, sdfMinLocal, sdfMaxLocal)) {
        const HitResult aabbHit = rayAABBIntersection(rayStartLocal, rayDirection, sdfMinLocal, sdfMaxLocal);
        if (aabbHit.hit) { rayStartLocal += aabbHit.t * rayDirection; hitDistanceLocal = aabbHit.t; }
        else return;
    }
    vec3 localSamplePos = rayStartLocal;
    const vec3 sdfResolution((float)sdf.w, (float)sdf.h, (float)sdf.d);
    const float distanceThreshold = length(localExtends / sdfResolution) * 0.25f;
    float dLast = 0.f, d = 0.f;
    const float localToGlobalScale = 1.f / length(vec3(instance.worldToLocal[0], instance.worldToLocal[1], instance.worldToLocal[2]));
    if (localToGlobalScale * hitDistanceLocal > tr.closestHitDistance) return;
    for (int i = 0; i < 128; i++) {
        vec3 localExtendsHalf = localExtends * 0.5f;
        localExtendsHalf += 0.01f;
        if (localSamplePos.x > localExtendsHalf.x || localSamplePos.y > localExtendsHalf.y || localSamplePos.z > localExtendsHalf.z ||
            localSamplePos.x < -localExtendsHalf.x || localSamplePos.y < -localExtendsHalf.y || localSamplePos.z < -localExtendsHalf.z)
            break;
        vec3 sampleUV = localSamplePos / localExtends + 0.5f;
        dLast = d;
        d = sampleSDF(sampleUV, sdf);
        if (d < distanceThreshold) {
            tr.hit = true;
            const float distanceGlobal = hitDistanceLocal * localToGlobalScale;
            if (distanceGlobal < tr.closestHitDistance) {
                tr.closestHitDistance = distanceGlobal;
                tr.hitCount = i;
                const float lastStepSizeLocal = d / (1.f - (d - dLast));
                localSamplePos += rayDirection * lastStepSizeLocal;
                sampleUV = localSamplePos / localExtends + 0.5f;
                tr.N = normalFromSDF(sampleUV, localExtends, sdf);
                // transpose(mat3(worldToLocal)) * N: component i = dot(column i of worldToLocal (xyz), N)
                const float* m = instance.worldToLocal;
                tr.N = vec3(m[0] * tr.N.x + m[1] * tr.N.y + m[2] * tr.N.z, m[4] * tr.N.x + m[5] * tr.N.y + m[6] * tr.N.z,
                            m[8] * tr.N.x + m[9] * tr.N.y + m[10] * tr.N.z);
                tr.albedo = pow(v3(instance.meanAlbedo), vec3(2.2f));
                const float lastStepSizeGlobal = lastStepSizeLocal * localToGlobalScale;
                tr.hitPos = rayStartWorld + rayDirectionWorld * (distanceGlobal + lastStepSizeGlobal);
            }
            break;
        }
        localSamplePos += rayDirection * std::fabs(d);
        hitDistanceLocal += std::fabs(d);
    }
}

// sunShadowCascades.inc:13-20
float simpleShadow(vec3 posWorld, const mat4& lightMatrix, const Image& shadowMap, int addr) {
    vec4 p = lightMatrix * vec4(posWorld, 1.f);
    p = p / p.w;
    const vec2 xy = vec2(p.x, p.y) * 0.5f + 0.5f;
    const float actualDepth = gclamp(p.z, 0.f, 1.f);
    const float shadowMapDepth = texture2D(shadowMap, NEAREST, addr, xy).x;
    return actualDepth > shadowMapDepth ? 1.f : 0.f;
}

struct RayInfo { vec3 normal; float depth; vec3 color; };

} // namespace

// sdfDiffuseTrace.comp:118-207 with resolveColor :70-116. Evaluated workgroup by workgroup (8x8), all 64 invocations of a
// group run the trace, then the shared-memory resolve; invocations outside the image take part in the exchange, their
// stores are dropped. bindless[] = the global texture array (set 2): SDF volumes and the noise textures.
extern "C" void orc_sdf_diffuse_trace(const orc_image* outYSHP, const orc_image* outCoCgP, const orc_image* depthP, const orc_image* normalP,
                                      const orc_image* skyLutP, const orc_light_buffer* light, const orc_sdf_instance* instances, const uint32_t* tiles,
                                      float influenceRange, const orc_shadow_cascade_info* shadowInfo, const orc_image* shadowMapP,
                                      const orc_image* bindless, int32_t nBindless, const orc_global* g, int32_t strictCutoff, int32_t shadowCascadeIndex) {
    const Image &imageOut_Y_SH = img(outYSHP), &imageOut_CoCg = img(outCoCgP), &depthTexture = img(depthP), &normalTexture = img(normalP),
                &skyLut = img(skyLutP), &shadowMap = img(shadowMapP);
    const int groupsX = (imageOut_Y_SH.w + 7) / 8, groupsY = (imageOut_Y_SH.h + 7) / 8;
    const Image& noiseTex = img(&bindless[g->noiseTextureIndices[g->frameIndexMod4]]);
    const mat4 lightMatrix = toMat4(shadowInfo->lightMatrices[shadowCascadeIndex]);
    const vec3 camFwd = v3(g->cameraForward), camPos = v3(g->cameraPosition);
    parallelFor(groupsY, [&](int gy0, int gy1) {
        for (int gy = gy0; gy < gy1; gy++)
            for (int gx = 0; gx < groupsX; gx++) {
                RayInfo sharedRays[8][8];
                vec3 Ls[8][8];
                uint32_t raySig[8][8];
                // tileUV = gl_WorkGroupID.xy / (cullingTileSize / 8)
                const ivec2 tileUV(gx / (int)(cullingTileSize / 8), gy / (int)(cullingTileSize / 8));
                const uint32_t tileIndex = tileIndexFromTileUV(tileUV, g);
                const uint32_t* cullingTile = tiles + (size_t)tileIndex * tileStrideUints;
                for (int ly = 0; ly < 8; ly++)
                    for (int lx = 0; lx < 8; lx++) {
                        const ivec2 iUV(gx * 8 + lx, gy * 8 + ly);
                        const vec2 uv = toVec2(iUV) / vec2((float)imageOut_Y_SH.w, (float)imageOut_Y_SH.h);
                        const float depth = texture2D(depthTexture, NEAREST, CLAMP, uv).x;
                        const float depthLinear = linearizeDepth(depth, g->nearPlane, g->farPlane);
                        const vec2 pixelNDC = uv * 2.f - 1.f;
                        const vec3 V = -calculateViewDirectionFromPixel(pixelNDC, camFwd, v3(g->cameraUp), v3(g->cameraRight), g->cameraTanFovHalf, g->cameraAspectRatio);
                        const vec3 pWorld = camPos + V / dot(V, camFwd) * depthLinear;
                        const vec2 noiseUV = toVec2(iUV) / vec2((float)noiseTex.w, (float)noiseTex.h);
                        const vec4 nz = texture2D(noiseTex, NEAREST, REPEAT, noiseUV);
                        const vec2 xi(nz.x, nz.y);
                        const vec3 normalTexel = texture2D(normalTexture, NEAREST, CLAMP, uv).xyz();
                        const vec3 N = normalTexel * 2.f - 1.f;
                        sharedRays[lx][ly].normal = N;
                        sharedRays[lx][ly].depth = depthLinear;
                        const vec3 rayOrigin = pWorld + N * 0.2f;
                        const vec3 L = importanceSampleCosine(xi, N);
                        Ls[lx][ly] = L;

                        TraceResult tr;
                        tr.hit = false;
                        tr.closestHitDistance = 10000.f;
                        tr.hitCount = 0;
                        const uint32_t objectCount = cullingTile[0];
                        uint32_t closestInstance = 0; // decision signature: owner of the closest hit, + 1
                        for (int i = 0; i < (int)objectCount; i++) {
                            const orc_sdf_instance& instance = instances[cullingTile[1 + i]];
                            const float before = tr.closestHitDistance;
                            traceRayTroughSDFInstance(instance, rayOrigin, img(&bindless[instance.sdfTextureIndex]), L, tr);
                            if (tr.closestHitDistance != before) closestInstance = cullingTile[1 + i] + 1u;
                        }
                        vec3 hitColor;
                        raySig[lx][ly] = (tr.hit ? 1u : 0u) | (closestInstance << 11);
                        if (tr.hit) {
                            const float shadow = simpleShadow(tr.hitPos, lightMatrix, shadowMap, BORDER_WHITE);
                            raySig[lx][ly] |= (shadow != 0.f ? 2u : 0u) | ((!(tr.closestHitDistance < influenceRange || !strictCutoff) || tr.closestHitDistance < 0.0001f) ? 4u : 0u);
                            const vec3 sunLight = shadow * light->sunStrengthExposed * v3(light->sunColor);
                            hitColor = tr.albedo * sunLight;
                            bool hitInRange = tr.closestHitDistance < influenceRange;
                            hitInRange = hitInRange || !strictCutoff;
                            const bool selfIntersection = tr.closestHitDistance < 0.0001f;
                            if (!hitInRange || selfIntersection) hitColor = vec3(0.f);
                        } else {
                            hitColor = sampleSkyLut(L, skyLut);
                        }
                        sharedRays[lx][ly].color = hitColor;
                    }
                // resolveColor (:70-116)
                for (int ly = 0; ly < 8; ly++)
                    for (int lx = 0; lx < 8; lx++) {
                        const vec3 initialColor = sharedRays[lx][ly].color;
                        float weightTotal = 1.f;
                        vec3 color = initialColor;
                        uint32_t takeMask = 0u;
                        int neighbour = -1;
                        for (int x = -1; x <= 1; x++)
                            for (int y = -1; y <= 1; y++) {
                                if (x == 0 && y == 0) continue;
                                neighbour++;
                                const int rx = lx + x, ry = ly + y;
                                const bool isValidIndex = (rx > 0 && ry > 0) && (rx < 8 && ry < 8); // sic: > 0, not >= 0 (:88)
                                if (!isValidIndex) continue;
                                const RayInfo& neighbourRay = sharedRays[rx][ry];
                                const float NoN = gclamp(dot(sharedRays[lx][ly].normal, neighbourRay.normal), 0.f, 1.f);
                                const bool normalsMatch = NoN > 0.9f;
                                const bool depthMatch = std::fabs(sharedRays[lx][ly].depth - neighbourRay.depth) < 0.5f;
                                if (normalsMatch && depthMatch) {
                                    takeMask |= 1u << neighbour;
                                    const float weightX = x == 0 ? 1.f : 0.5f, weightY = y == 0 ? 1.f : 0.5f;
                                    const float weight = weightX * weightY;
                                    color += weight * neighbourRay.color;
                                    weightTotal += weight;
                                }
                            }
                        color /= weightTotal;
                        const vec3 YCoCg = linearToYCoCg(color);
                        vec4 result_Y_SH(0.f);
                        vec2 result_CoCg(0.f);
                        result_Y_SH += YCoCg.x * directionToSH_L1(Ls[lx][ly]);
                        result_CoCg += vec2(YCoCg.y, YCoCg.z);
                        const ivec2 iUV(gx * 8 + lx, gy * 8 + ly);
                        imageStore(imageOut_Y_SH, iUV, result_Y_SH);
                        imageStore(imageOut_CoCg, iUV, vec4(result_CoCg.x, result_CoCg.y, 0.f, 0.f));
                        if (iUV.x < imageOut_Y_SH.w && iUV.y < imageOut_Y_SH.h) writeSig((int64_t)iUV.y * imageOut_Y_SH.w + iUV.x, raySig[lx][ly] | (takeMask << 3));
                    }
            }
    });
}

// ------------------------------------------------------------------------------------------------ denoise
namespace {
vec3 pixelToWorld(vec2 uv, const Image& depthTexture, const orc_global* g) {
    const float depth = texture2D(depthTexture, NEAREST, CLAMP, uv).x;
    const float depthLinear = linearizeDepth(depth, g->nearPlane, g->farPlane);
    const vec2 pixelNDC = uv * 2.f - 1.f;
    const vec3 cameraToPixel = -calculateViewDirectionFromPixel(pixelNDC, v3(g->cameraForward), v3(g->cameraUp), v3(g->cameraRight), g->cameraTanFovHalf, g->cameraAspectRatio);
    return v3(g->cameraPosition) + cameraToPixel / dot(cameraToPixel, v3(g->cameraForward)) * depthLinear;
}
} // namespace

// filterIndirectDiffuseSpatial.comp:30-135
extern "C" void orc_filter_indirect_diffuse_spatial(const orc_image* outYSHP, const orc_image* outCoCgP, const orc_image* inYSHP, const orc_image* inCoCgP,
                                                    const orc_image* depthP, const orc_image* normalP, const orc_global* g, int32_t filterIndex) {
    const Image &imageOut_Y_SH = img(outYSHP), &imageOut_CoCg = img(outCoCgP), &texture_Y_SH = img(inYSHP), &texture_CoCg = img(inCoCgP),
                &depthTexture = img(depthP), &normalTexture = img(normalP);
    const mat4 viewProjection = toMat4(g->viewProjection);
    parallelFor(imageOut_Y_SH.h, [&](int y0, int y1) {
        for (int py = y0; py < y1; py++)
            for (int px = 0; px < imageOut_Y_SH.w; px++) {
                const ivec2 iUV(px, py);
                const vec2 texelSize = 1.f / vec2((float)imageOut_Y_SH.w, (float)imageOut_Y_SH.h);
                const vec2 uv = (toVec2(iUV) + 0.5f) * texelSize;
                const vec3 pCenter = pixelToWorld(uv, depthTexture, g);
                const vec3 pRight = pixelToWorld(uv + vec2(1, 0) * texelSize, depthTexture, g);
                const vec3 pUp = pixelToWorld(uv + vec2(0, 1) * texelSize, depthTexture, g);
                const vec3 tangent = normalize(pCenter - pRight);
                const vec3 bitangent = normalize(pCenter - pUp);
                // the geometric normal is computed and immediately overwritten (:43-44)
                const vec3 N = 2.f * texture2D(normalTexture, NEAREST, CLAMP, uv).xyz() - 1.f;
                const int sampleCount = 32;
                vec4 result_Y_SH(0.f);
                vec2 result_CoCg(0.f);
                float weightTotal = 0.f;
                uint32_t rngState = wang_hash(g->frameIndexMod4 + (uint32_t)filterIndex);
                float radiusWorld = 1.5f;
                if (filterIndex == 1) radiusWorld = 1.f;
                float lengthModifier = 1.f;
                uint32_t sampleParityX = 0u, sampleParityY = 0u; // decision signature
                for (int i = 0; i < sampleCount; i++) {
                    const float d = std::sqrt(rand01(rngState)) * lengthModifier;
                    const float angle = 2.f * pi * rand01(rngState);
                    float sa, ca;
                    det_sincosf(angle, &sa, &ca);
                    const vec2 offset = vec2(ca, sa) * d;
                    const vec3 sampleWorld = pCenter + radiusWorld * (offset.x * tangent + offset.y * bitangent);
                    const vec4 sampleProjected = viewProjection * vec4(sampleWorld, 1.f);
                    vec2 sampleUV = vec2(sampleProjected.x, sampleProjected.y) / sampleProjected.w;
                    sampleUV = sampleUV * 0.5f + 0.5f;
                    sampleUV.x = sampleUV.x < 0.f ? uv.x - offset.x : sampleUV.x;
                    sampleUV.y = sampleUV.y < 0.f ? uv.y - offset.y : sampleUV.y;
                    sampleUV.x = sampleUV.x > 1.f ? uv.x - offset.x : sampleUV.x;
                    sampleUV.y = sampleUV.y > 1.f ? uv.y - offset.y : sampleUV.y;
                    const vec3 pixelWorld = pixelToWorld(sampleUV, depthTexture, g);
                    const float distanceToTangentPlane = std::fabs(dot(N, pixelWorld - pCenter));
                    const float maxDistance = 0.25f;
                    float weight = gclamp(maxDistance / gmax(distanceToTangentPlane, 0.0001f), 0.f, 1.f);
                    weight *= weight;
                    const bool offScreen = sampleUV.x < 0.f || sampleUV.y < 0.f || sampleUV.x > 1.f || sampleUV.y > 1.f;
                    if (offScreen) {
                        weight = 0.f;
                        lengthModifier *= 0.98f;
                    }
                    {
                        // nearest texel of the sample in the Y_SH input (clamp to edge), as texture2D(NEAREST, CLAMP) addresses it
                        int tx = (int)std::floor(saneCoord(sampleUV.x * (float)texture_Y_SH.w)), ty = (int)std::floor(saneCoord(sampleUV.y * (float)texture_Y_SH.h));
                        tx = tx < 0 ? 0 : (tx >= texture_Y_SH.w ? texture_Y_SH.w - 1 : tx);
                        ty = ty < 0 ? 0 : (ty >= texture_Y_SH.h ? texture_Y_SH.h - 1 : ty);
                        sampleParityX |= (uint32_t)((tx + (offScreen ? 1 : 0)) & 1) << i;
                        sampleParityY |= (uint32_t)((ty + (offScreen ? 1 : 0)) & 1) << i;
                    }
                    if (weight > 0.f) {
                        const vec4 sample_Y_SH = texture2D(texture_Y_SH, NEAREST, CLAMP, sampleUV);
                        const vec4 c = texture2D(texture_CoCg, NEAREST, CLAMP, sampleUV);
                        const vec2 sample_CoCg(c.x, c.y);
                        if (isnan4(sample_Y_SH) || isnan2(sample_CoCg)) {
                        } else {
                            result_Y_SH += weight * sample_Y_SH;
                            result_CoCg += weight * sample_CoCg;
                            weightTotal += weight;
                        }
                    }
                }
                weightTotal = gmax(weightTotal, 0.00001f);
                result_Y_SH /= weightTotal;
                result_CoCg /= weightTotal;
                imageStore(imageOut_Y_SH, iUV, result_Y_SH);
                imageStore(imageOut_CoCg, iUV, vec4(result_CoCg.x, result_CoCg.y, 0, 0));
                writeSig(2 * ((int64_t)py * imageOut_Y_SH.w + px), sampleParityX);
                writeSig(2 * ((int64_t)py * imageOut_Y_SH.w + px) + 1, sampleParityY);
            }
    });
}

// filterIndirectDiffuseTemporal.comp:20-86
extern "C" void orc_filter_indirect_diffuse_temporal(const orc_image* targetYSHP, const orc_image* targetCoCgP, const orc_image* historyOutYSHP,
                                                     const orc_image* historyOutCoCgP, const orc_image* inYSHP, const orc_image* inCoCgP,
                                                     const orc_image* historyInYSHP, const orc_image* historyInCoCgP, const orc_image* velocityCurrentP,
                                                     const orc_image* velocityLastP, const orc_global* g) {
    const Image &targetOut_Y_SH = img(targetYSHP), &targetOut_CoCg = img(targetCoCgP), &historyOut_Y_SH = img(historyOutYSHP),
                &historyOut_CoCg = img(historyOutCoCgP), &input_Y_SH = img(inYSHP), &input_CoCg = img(inCoCgP), &historyIn_Y_SH = img(historyInYSHP),
                &historyIn_CoCg = img(historyInCoCgP), &velocityCurrent = img(velocityCurrentP), &velocityLastFrame = img(velocityLastP);
    const vec2 screenRes((float)g->screenResolution[0], (float)g->screenResolution[1]);
    parallelFor(targetOut_Y_SH.h, [&](int y0, int y1) {
        for (int py = y0; py < y1; py++)
            for (int px = 0; px < targetOut_Y_SH.w; px++) {
                const ivec2 iUV(px, py);
                const vec2 texelSize = 1.f / vec2((float)targetOut_Y_SH.w, (float)targetOut_Y_SH.h);
                const vec2 uv = (toVec2(iUV) + 0.5f) * texelSize;
                const vec4 current_Y_SH = texture2D(input_Y_SH, LINEAR, CLAMP, uv);
                vec4 t = texture2D(input_CoCg, LINEAR, CLAMP, uv);
                const vec2 current_CoCg(t.x, t.y);
                t = texture2D(velocityCurrent, LINEAR, CLAMP, uv);
                const vec2 motion(t.x, t.y);
                const vec2 uvReprojected = uv + motion;
                vec4 history_Y_SH = texture2D(historyIn_Y_SH, LINEAR, CLAMP, uvReprojected);
                t = texture2D(historyIn_CoCg, LINEAR, CLAMP, uvReprojected);
                vec2 history_CoCg(t.x, t.y);
                t = texture2D(velocityLastFrame, LINEAR, REPEAT, uvReprojected);
                const vec2 motionLastFrame(t.x, t.y);
                const float motionDifference = std::sqrt(std::fabs(length(motion) - length(motionLastFrame)));
                const float K = 10.f;
                const float motionDifferenceFactor = gclamp(motionDifference * K, 0.f, 1.f);
                const float alphaDefault = 0.8f;
                float alphaMin = 0.6f;
                alphaMin -= 0.3f * std::fabs(length(current_Y_SH) - length(history_Y_SH));
                alphaMin = gmax(alphaMin, 0.f);
                float alpha = gmix(alphaDefault, alphaMin, motionDifferenceFactor);
                const float pixelThreshold = 3.f;
                const vec2 am = abs(motion) * screenRes, al = abs(motionLastFrame) * screenRes;
                if (am.x > pixelThreshold || am.y > pixelThreshold || al.x > pixelThreshold || al.y > pixelThreshold) alpha = alphaMin;
                if (uvReprojected.x < 0.f || uvReprojected.y < 0.f || uvReprojected.x > 1.f || uvReprojected.y > 1.f) alpha = 0.f;
                if (g->cameraCut) alpha = 0.f;
                if (isnan4(current_Y_SH) || isnan2(current_CoCg)) {
                    alpha = 1.f;
                    if (isnan4(history_Y_SH)) history_Y_SH = vec4(0.f);
                    if (isnan2(history_CoCg)) history_CoCg = vec2(0.f);
                }
                const vec4 result_Y_SH = current_Y_SH * (1.f - alpha) + history_Y_SH * alpha;
                const vec2 result_CoCg = current_CoCg * (1.f - alpha) + history_CoCg * alpha;
                imageStore(targetOut_Y_SH, iUV, result_Y_SH);
                imageStore(targetOut_CoCg, iUV, vec4(result_CoCg.x, result_CoCg.y, 0, 0));
                imageStore(historyOut_Y_SH, iUV, result_Y_SH);
                imageStore(historyOut_CoCg, iUV, vec4(result_CoCg.x, result_CoCg.y, 0, 0));
            }
    });
}

// indirectLightUpscale.comp:17-71
extern "C" void orc_indirect_light_upscale(const orc_image* dstYSHP, const orc_image* dstCoCgP, const orc_image* srcYSHP, const orc_image* srcCoCgP,
                                           const orc_image* fullResDepthP, const orc_image* halfResDepthP, const orc_global* g) {
    const Image &fullResDst_Y_SH = img(dstYSHP), &fullResDst_CoCg = img(dstCoCgP), &halfResSrc_Y_SH = img(srcYSHP), &halfResSrc_CoCg = img(srcCoCgP),
                &fullResDepthT = img(fullResDepthP), &halfResDepthT = img(halfResDepthP);
    const vec2 screenRes((float)g->screenResolution[0], (float)g->screenResolution[1]);
    parallelFor(fullResDst_Y_SH.h, [&](int y0, int y1) {
        for (int py = y0; py < y1; py++)
            for (int px = 0; px < fullResDst_Y_SH.w; px++) {
                const ivec2 iUV(px, py);
                const vec2 uv = (toVec2(iUV) + 0.5f) / screenRes;
                float fullResDepth = texture2D(fullResDepthT, NEAREST, CLAMP, uv).x;
                fullResDepth = linearizeDepth(fullResDepth, g->nearPlane, g->farPlane);
                const vec2 halfResTexelSize = 1.f / vec2((float)halfResDepthT.w, (float)halfResDepthT.h);
                vec4 depthSamples = textureGatherR(halfResDepthT, CLAMP, uv);
                for (int i = 0; i < 4; i++) depthSamples[i] = linearizeDepth(depthSamples[i], g->nearPlane, g->farPlane);
                float minDepthDiff = 1000.f;
                vec2 closestDepthTexel(0.f);
                const float edgeDepthThreshold = 0.5f;
                bool isEdge = false;
                const vec2 offsets[4] = {vec2(0, 1), vec2(1, 1), vec2(1, 0), vec2(0, 0)};
                for (int i = 0; i < 4; i++) {
                    const float depthDiff = std::fabs(depthSamples[i] - fullResDepth);
                    isEdge = isEdge || depthDiff > edgeDepthThreshold;
                    if (depthDiff < minDepthDiff) { minDepthDiff = depthDiff; closestDepthTexel = offsets[i]; }
                }
                const vec2 uvClosestTexel = uv + closestDepthTexel * halfResTexelSize;
                vec4 result_Y_SH, cc;
                if (isEdge) {
                    result_Y_SH = texture2D(halfResSrc_Y_SH, NEAREST, CLAMP, uvClosestTexel);
                    cc = texture2D(halfResSrc_CoCg, NEAREST, CLAMP, uvClosestTexel);
                } else {
                    result_Y_SH = texture2D(halfResSrc_Y_SH, LINEAR, CLAMP, uv);
                    cc = texture2D(halfResSrc_CoCg, LINEAR, CLAMP, uv);
                }
                imageStore(fullResDst_Y_SH, iUV, result_Y_SH);
                imageStore(fullResDst_CoCg, iUV, vec4(cc.x, cc.y, 0.f, 0.f));
                writeSig((int64_t)py * fullResDst_Y_SH.w + px, (isEdge ? 1u : 0u) | (closestDepthTexel.x != 0.f ? 2u : 0u) | (closestDepthTexel.y != 0.f ? 4u : 0u));
            }
    });
}

// sdfDebugVisualisation.comp:73-133. debugMode: 1 lit SDF, 2 camera tile usage, 3 normals, 4 raymarching steps.
// Fields of TraceResult the shader leaves uninitialised without a hit start at zero here (they reach no output in that case).
extern "C" void orc_sdf_debug_visualisation(const orc_image* outP, const orc_light_buffer* light, const orc_image* skyLutP, const orc_sdf_instance* instances,
                                            const uint32_t* tiles, const orc_shadow_cascade_info* shadowInfo, const orc_image* shadowMapP, const orc_image* bindless,
                                            int32_t nBindless, const orc_global* g, int32_t debugMode, int32_t shadowCascadeIndex) {
    (void)nBindless;
    const Image &imageOut = img(outP), &skyLut = img(skyLutP), &shadowMap = img(shadowMapP);
    const mat4 lightMatrix = toMat4(shadowInfo->lightMatrices[shadowCascadeIndex]);
    parallelFor(imageOut.h, [&](int y0, int y1) {
        for (int y = y0; y < y1; y++)
            for (int x = 0; x < imageOut.w; x++) {
                const ivec2 uv(x, y);
                const vec3 cameraToPixel = -VFromiUV(uv, g);
                const ivec2 tileUV(x / (int)cullingTileSize, y / (int)cullingTileSize);
                const uint32_t tileIndex = tileIndexFromTileUV(tileUV, g);
                const vec3 rayStart = v3(g->cameraPosition) + g->nearPlane * cameraToPixel;
                TraceResult tr;
                tr.hit = false;
                tr.closestHitDistance = 10000.f;
                tr.hitPos = vec3(0.f); tr.N = vec3(0.f); tr.albedo = vec3(0.f); tr.hitCount = 0;
                const uint32_t* cullingTile = tiles + (size_t)tileIndex * tileStrideUints;
                const int objectCount = (int)cullingTile[0];
                for (int i = 0; i < objectCount; i++) {
                    const orc_sdf_instance& instance = instances[cullingTile[1 + i]];
                    traceRayTroughSDFInstance(instance, rayStart, img(&bindless[instance.sdfTextureIndex]), cameraToPixel, tr);
                }
                const float shadow = simpleShadow(tr.hitPos, lightMatrix, shadowMap, BORDER_BLACK);
                vec3 color(0.f);
                if (tr.hit || debugMode == 2) {
                    if (debugMode == 1) {
                        vec3 sunLight = light->sunStrengthExposed * vec3(light->sunColor[0], light->sunColor[1], light->sunColor[2]);
                        sunLight = sunLight * shadow;
                        const vec3 ambient(0.15f);
                        const float NoL = gclamp(dot(tr.N, vec3(g->sunDirection[0], g->sunDirection[1], g->sunDirection[2])), 0.f, 1.f);
                        color = tr.albedo * (ambient + sunLight * NoL);
                    } else if (debugMode == 2) {
                        const float percentage = (float)cullingTile[0] / (float)maxObjectsPerTile;
                        color = percentage >= 1.f ? vec3(1.f, 0.f, 0.f) : vec3(percentage);
                    } else if (debugMode == 3) {
                        color = tr.N * 0.5f + 0.5f;
                    } else if (debugMode == 4) {
                        color = vec3((float)tr.hitCount / 128.f);
                    }
                } else {
                    color = sampleSkyLut(cameraToPixel, skyLut);
                }
                imageStore(imageOut, uv, vec4(color, 1.f));
            }
    });
}

// ------------------------------------------------------------------------------------------------ known-answer probes (tests/test_kat.py)
// One ray through one SDF instance (SDF.inc:101-184): rays = n x {origin3, direction3} in world space.
// out = n x {hit, closestHitDistance, hitPos3, N3, hitCount} (9 floats).
extern "C" void orc_kat_trace_ray(const orc_sdf_instance* instance, const orc_image* volume, const float* rays, float* out, int64_t n) {
    for (int64_t i = 0; i < n; i++) {
        const float* r = rays + 6 * i;
        TraceResult tr;
        tr.hit = false; tr.closestHitDistance = 10000.f; tr.hitPos = vec3(0.f); tr.N = vec3(0.f); tr.hitCount = 0; tr.albedo = vec3(0.f);
        traceRayTroughSDFInstance(*instance, vec3(r[0], r[1], r[2]), img(volume), vec3(r[3], r[4], r[5]), tr);
        float* o = out + 9 * i;
        o[0] = tr.hit ? 1.f : 0.f; o[1] = tr.closestHitDistance; o[2] = tr.hitPos.x; o[3] = tr.hitPos.y; o[4] = tr.hitPos.z;
        o[5] = tr.N.x; o[6] = tr.N.y; o[7] = tr.N.z; o[8] = (float)tr.hitCount;
    }
}
