// TEST INFRASTRUCTURE ONLY (see oracle.h). CPU restatement of the asset-pipeline SDF bake (BASELINE config 1).
// PARITY UNPINNED: the reference holds no test, golden volume or known answer for this path.
//
// Follows  Plain/src/AssetPipeline/SceneSDF.cpp:55-95 (closest triangle distance), :116-131 (resolution rule), :159-231 (SAT
// triangle/box overlap), :233-251 (index helpers), :253-294 (uniform grid), :296-513 (computeSDF);
// Plain/src/Common/sdfUtilities.cpp:5-19 (padding), Common/VolumeInfo.cpp:4-9, Common/Utilities/MathUtils.cpp:4-15
// (directionToVector), Common/AABB.cpp:160-167 (isPointInAABB).
// Third-party arithmetic: glm (un-vendored submodule Plain/vendor/glm, no pinned commit): glm::min/max/clamp/sign/normalize/
// radians and glm::packHalf (detail::toFloat16: round-half-UP in magnitude, overflow to infinity) are restated from glm's
// published definitions. sin/cos/acos are the detmath contract functions (the reference calls libm).
#include <cmath>
#include <cstdint>
#include <cstring>
#include <limits>
#include <vector>

#include "common.h"
#include "oracle.h"

namespace orc {
namespace {

// glm/detail/func_common.inl
inline float glmMin(float x, float y) { return (y < x) ? y : x; }
inline float glmMax(float x, float y) { return (x < y) ? y : x; }
inline float glmClamp(float x, float lo, float hi) { return glmMin(glmMax(x, lo), hi); }
inline float glmSign(float x) { return (float)((0.f < x) - (x < 0.f)); }
inline vec3 glmMin(vec3 a, vec3 b) { return vec3(glmMin(a.x, b.x), glmMin(a.y, b.y), glmMin(a.z, b.z)); }
inline vec3 glmMax(vec3 a, vec3 b) { return vec3(glmMax(a.x, b.x), glmMax(a.y, b.y), glmMax(a.z, b.z)); }
inline float dot2(vec3 v) { return dot(v, v); }
inline float comp(const vec3& v, int c) { return c == 0 ? v.x : (c == 1 ? v.y : v.z); }

// glm/detail/type_half.inl toFloat16
uint16_t packHalfGlm(float f) {
    uint32_t u; std::memcpy(&u, &f, 4);
    const int i = (int)u;
    const int s = (i >> 16) & 0x00008000;
    int e = ((i >> 23) & 0x000000ff) - (127 - 15);
    int m = i & 0x007fffff;
    if (e <= 0) {
        if (e < -10) return (uint16_t)s;
        m = (m | 0x00800000) >> (1 - e);
        if (m & 0x00001000) m += 0x00002000;
        return (uint16_t)(s | (m >> 13));
    } else if (e == 0xff - (127 - 15)) {
        if (m == 0) return (uint16_t)(s | 0x7c00);
        m >>= 13;
        return (uint16_t)(s | 0x7c00 | m | (m == 0));
    } else {
        if (m & 0x00001000) {
            m += 0x00002000;
            if (m & 0x00800000) { m = 0; e += 1; }
        }
        if (e > 30) return (uint16_t)(s | 0x7c00);
        return (uint16_t)(s | (e << 10) | (m >> 13));
    }
}

struct Tri { vec3 v0, v1, v2, N; };
struct Box { vec3 mn, mx; };
struct Volume { vec3 extends, offset; };

// SceneSDF.cpp:42-53
uint32_t nextPowerOfTwo(uint32_t in) {
    uint32_t out = in;
    out--;
    out |= out >> 1; out |= out >> 2; out |= out >> 4; out |= out >> 8; out |= out >> 16;
    out++;
    return out;
}

// sdfUtilities.cpp:5-19
Box padSDFBoundingBox(const Box& bb) {
    vec3 padding = 0.075f * (bb.mx - bb.mn);
    padding = glmMax(padding, vec3(0.5f));
    Box p;
    p.mn = bb.mn - padding;
    p.mx = bb.mx + padding;
    return p;
}

// VolumeInfo.cpp:4-9
Volume volumeInfoFromBoundingBox(const Box& bb) {
    Volume v;
    v.offset = (bb.mx + bb.mn) * 0.5f;
    v.extends = bb.mx - bb.mn;
    return v;
}

// MathUtils.cpp:4-15; glm::radians(x) = x * 0.01745329251994329576923690768489
vec3 directionToVector(vec2 direction) {
    const float theta = direction.y * 0.01745329251994329576923690768489f;
    const float phi = direction.x * 0.01745329251994329576923690768489f;
    return vec3(det_sinf(theta) * det_cosf(phi), -det_cosf(theta), det_sinf(theta) * det_sinf(phi));
}

// SceneSDF.cpp:233-235
inline int flattenGridIndex(int x, int y, int z, ivec3 res) { return x + y * res.x + z * res.x * res.y; }

// SceneSDF.cpp:237-244
ivec3 pointToCellIndex(vec3 p, const Box& aabb, ivec3 res) {
    const vec3 pRelative = p - aabb.mn;
    const vec3 ext = aabb.mx - aabb.mn;
    vec3 n(pRelative.x / ext.x, pRelative.y / ext.y, pRelative.z / ext.z);
    n = vec3(glmClamp(n.x, 0.f, 0.999f), glmClamp(n.y, 0.f, 0.999f), glmClamp(n.z, 0.f, 0.999f));
    const vec3 t(n.x * (float)res.x, n.y * (float)res.y, n.z * (float)res.z);
    return ivec3((int)std::floor(t.x), (int)std::floor(t.y), (int)std::floor(t.z));
}

// SceneSDF.cpp:246-251
vec3 volumeIndexToCellCenter(int x, int y, int z, ivec3 res, const Volume& vol) {
    const vec3 n(((float)x + 0.5f) / (float)res.x, ((float)y + 0.5f) / (float)res.y, ((float)z + 0.5f) / (float)res.z);
    const vec3 s = n - 0.5f;
    return vec3(s.x * vol.extends.x, s.y * vol.extends.y, s.z * vol.extends.z) + vol.offset;
}

// SceneSDF.cpp:160-173
bool isAxisSeparating(vec3 axis, vec3 bbHalf, vec3 v0, vec3 v1, vec3 v2) {
    const float p0 = dot(axis, v0), p1 = dot(axis, v1), p2 = dot(axis, v2);
    const float r = dot(vec3(std::fabs(axis.x), std::fabs(axis.y), std::fabs(axis.z)), bbHalf);
    const float pMin = glmMin(glmMin(p0, p1), p2);
    const float pMax = glmMax(glmMax(p0, p1), p2);
    return pMin > r || pMax < -r;
}

// SceneSDF.cpp:178-231
bool doTriangleAABBOverlap(vec3 bbCenter, vec3 bbExtends, vec3 v0In, vec3 v1In, vec3 v2In, vec3 N) {
    const vec3 v0 = v0In - bbCenter, v1 = v1In - bbCenter, v2 = v2In - bbCenter;
    const vec3 e[3] = {v1 - v0, v2 - v1, v0 - v2};
    const vec3 bbHalf = bbExtends * 0.5f;
    const vec3 bbN[3] = {vec3(1, 0, 0), vec3(0, 1, 0), vec3(0, 0, 1)};
    for (int k = 0; k < 3; k++)
        for (int a = 0; a < 3; a++)
            if (isAxisSeparating(cross(bbN[a], e[k]), bbHalf, v0, v1, v2)) return false;
    for (int a = 0; a < 3; a++)
        if (isAxisSeparating(bbN[a], bbHalf, v0, v1, v2)) return false;
    if (isAxisSeparating(N, bbHalf, v0, v1, v2)) return false;
    return true;
}

Tri makeTriangle(const float* positions, uint32_t i0, uint32_t i1, uint32_t i2) {
    Tri t;
    t.v0 = vec3(positions[3 * i0], positions[3 * i0 + 1], positions[3 * i0 + 2]);
    t.v1 = vec3(positions[3 * i1], positions[3 * i1 + 1], positions[3 * i1 + 2]);
    t.v2 = vec3(positions[3 * i2], positions[3 * i2 + 1], positions[3 * i2 + 2]);
    t.N = normalize(cross(t.v0 - t.v2, t.v0 - t.v1)); // :273 / :317
    return t;
}

// SceneSDF.cpp:55-95
float computePointTrianglesClosestDistance(vec3 p, const std::vector<Tri>& triangles) {
    float closestD = std::numeric_limits<float>::infinity();
    for (const Tri& t : triangles) {
        const vec3 v1ToP = p - t.v0, v2ToP = p - t.v1, v3ToP = p - t.v2;
        const vec3 v0ToV1 = t.v1 - t.v0, v1ToV2 = t.v2 - t.v1, v2ToV0 = t.v0 - t.v2;
        const vec3 eN0 = cross(v0ToV1, t.N), eN1 = cross(v1ToV2, t.N), eN2 = cross(v2ToV0, t.N);
        const float s1 = glmSign(dot(eN0, -v1ToP)), s2 = glmSign(dot(eN1, -v2ToP)), s3 = glmSign(dot(eN2, -v3ToP));
        const bool onEdge = s1 + s2 + s3 < 2.f;
        const float c1 = glmClamp(dot(v1ToP, v0ToV1) / dot2(v0ToV1), 0.f, 1.f);
        const float c2 = glmClamp(dot(v2ToP, v1ToV2) / dot2(v1ToV2), 0.f, 1.f);
        const float c3 = glmClamp(dot(v3ToP, v2ToV0) / dot2(v2ToV0), 0.f, 1.f);
        const float l1 = dot2(p - (t.v0 + v0ToV1 * c1));
        const float l2 = dot2(p - (t.v1 + v1ToV2 * c2));
        const float l3 = dot2(p - (t.v2 + v2ToV0 * c3));
        float d = onEdge ? glmMin(glmMin(l1, l2), l3) : std::fabs(dot(t.N, v1ToP) * dot(t.N, v1ToP));
        d = std::fabs(d);
        closestD = glmMin(closestD, d);
    }
    return std::sqrt(std::fabs(closestD));
}

} // namespace
} // namespace orc

using namespace orc;

extern "C" void orc_sdf_resolution(const float* bbMin3, const float* bbMax3, int32_t* res3) {
    // SceneSDF.cpp:116-131
    for (int c = 0; c < 3; c++) {
        const float targetRes = (bbMax3[c] - bbMin3[c]) / 0.25f;
        uint32_t r = nextPowerOfTwo((uint32_t)targetRes);
        r = r < 16u ? 16u : (r > 64u ? 64u : r);
        res3[c] = (int32_t)r;
    }
}

extern "C" void orc_sdf_padded_box(const float* bbMin3, const float* bbMax3, float* outMin3, float* outMax3) {
    Box bb{vec3(bbMin3[0], bbMin3[1], bbMin3[2]), vec3(bbMax3[0], bbMax3[1], bbMax3[2])};
    const Box p = padSDFBoundingBox(bb);
    outMin3[0] = p.mn.x; outMin3[1] = p.mn.y; outMin3[2] = p.mn.z;
    outMax3[0] = p.mx.x; outMax3[1] = p.mx.y; outMax3[2] = p.mx.z;
}

extern "C" uint16_t orc_pack_half_glm(float v) { return packHalfGlm(v); }

extern "C" int32_t orc_sdf_bake(const float* positions, int64_t nVerts, const uint32_t* indices, int64_t nIndices, const float* bbMin3,
                                const float* bbMax3, int32_t resX, int32_t resY, int32_t resZ, uint16_t* outHalf) {
    if (nIndices % 3 != 0 || resX <= 0 || resY <= 0 || resZ <= 0) return -1;
    for (int64_t i = 0; i < nIndices; i++) if ((int64_t)indices[i] >= nVerts) return -2;
    const Box aabb{vec3(bbMin3[0], bbMin3[1], bbMin3[2]), vec3(bbMax3[0], bbMax3[1], bbMax3[2])};
    const Box padded = padSDFBoundingBox(aabb);
    const Volume vol = volumeInfoFromBoundingBox(padded);
    const ivec3 gridRes(16, 16, 16);
    const ivec3 res(resX, resY, resZ);
    const vec3 cellSize(vol.extends.x / 16.f, vol.extends.y / 16.f, vol.extends.z / 16.f);

    // ---- buildUniformGrid (:253-294)
    std::vector<std::vector<Tri>> grid(16 * 16 * 16);
    std::vector<Tri> all;
    all.reserve((size_t)nIndices / 3);
    for (int64_t i = 0; i < nIndices; i += 3) {
        const Tri t = makeTriangle(positions, indices[i], indices[i + 1], indices[i + 2]);
        all.push_back(t);
        const vec3 tMin = glmMin(glmMin(t.v0, t.v1), t.v2), tMax = glmMax(glmMax(t.v0, t.v1), t.v2);
        const ivec3 lo = pointToCellIndex(tMin, padded, gridRes), hi = pointToCellIndex(tMax, padded, gridRes);
        for (int x = lo.x; x <= hi.x; x++)
            for (int y = lo.y; y <= hi.y; y++)
                for (int z = lo.z; z <= hi.z; z++) {
                    const vec3 c = volumeIndexToCellCenter(x, y, z, gridRes, vol);
                    if (doTriangleAABBOverlap(c, cellSize, t.v0, t.v1, t.v2, t.N)) grid[(size_t)flattenGridIndex(x, y, z, gridRes)].push_back(t);
                }
    }

    // ---- the 225 ray directions (:351-364) do not depend on the voxel
    const int sampleCount1D = 15;
    vec3 dirs[225];
    for (int sx = 0; sx < sampleCount1D; sx++)
        for (int sy = 0; sy < sampleCount1D; sy++) {
            const float sampleX = (float)sx / (float)(sampleCount1D - 1);
            const float sampleY = (float)sy / (float)(sampleCount1D - 1) * 2.f - 1.f;
            const float phi = sampleX * 2.f * 3.1415f;
            const float theta = det_acosf(sampleY);
            const vec2 angles(phi / 3.1415f * 180.f, theta / 3.1415f * 180.f);
            dirs[sx * sampleCount1D + sy] = directionToVector(angles);
        }

    const float inf = std::numeric_limits<float>::infinity();
    parallelFor(resZ * resY, [&](int r0, int r1) {
        for (int row = r0; row < r1; row++) {
            const int z = row / resY, y = row % resY;
            for (int x = 0; x < resX; x++) {
                const vec3 rayOrigin = volumeIndexToCellCenter(x, y, z, res, vol);
                float closestHitTotal = inf;
                uint32_t backHitCounter = 0;
                for (int ray = 0; ray < 225; ray++) {
                    const vec3 rayDirection = dirs[ray];
                    bool isBackfaceHit = false;
                    float rayClosestHit = inf;
                    bool rayIsInBoundingBox = true;
                    const ivec3 start = pointToCellIndex(rayOrigin, padded, gridRes);
                    uint32_t gi[3] = {(uint32_t)start.x, (uint32_t)start.y, (uint32_t)start.z}; // glm::uvec3 (:372)
                    vec3 cur = rayOrigin;
                    while (rayIsInBoundingBox) {
                        const size_t cellIndex = (size_t)flattenGridIndex((int)gi[0], (int)gi[1], (int)gi[2], gridRes);
                        const vec3 cellMin = padded.mn + vec3((float)gi[0] / 16.f * vol.extends.x, (float)gi[1] / 16.f * vol.extends.y, (float)gi[2] / 16.f * vol.extends.z);
                        const vec3 cellMax = cellMin + cellSize;
                        bool hitTriangle = false;
                        for (const Tri& tri : grid[cellIndex]) {
                            const float NoR = dot(tri.N, rayDirection);
                            if (std::fabs(NoR) < 0.0001f) continue;
                            const float D = dot(tri.N, tri.v0);
                            const float t = (D - dot(tri.N, rayOrigin)) / NoR;
                            if (t < 0.f) continue;
                            const vec3 edge0 = tri.v1 - tri.v0, edge1 = tri.v2 - tri.v1, edge2 = tri.v0 - tri.v2;
                            const vec3 planeIntersection = rayOrigin + rayDirection * t;
                            const vec3 C0 = planeIntersection - tri.v0, C1 = planeIntersection - tri.v1, C2 = planeIntersection - tri.v2;
                            const float d0 = dot(tri.N, cross(C0, edge0)), d1 = dot(tri.N, cross(C1, edge1)), d2 = dot(tri.N, cross(C2, edge2));
                            if (!(d0 >= 0.f && d1 >= 0.f && d2 >= 0.f)) continue;
                            const vec3 hitPos = rayOrigin + t * rayDirection;
                            const bool hitInCurrentCell = hitPos.x <= cellMax.x && hitPos.x >= cellMin.x && hitPos.y <= cellMax.y && hitPos.y >= cellMin.y &&
                                                          hitPos.z <= cellMax.z && hitPos.z >= cellMin.z;
                            if (!hitInCurrentCell) continue;
                            hitTriangle = true;
                            if (t < rayClosestHit) {
                                rayClosestHit = t;
                                isBackfaceHit = dot(rayDirection, tri.N) > 0.f;
                            }
                        }
                        if (hitTriangle) break;
                        float distanceToNext = inf;
                        int intersectedComponent = 0;
                        for (int c = 0; c < 3; c++) {
                            const float dc = comp(rayDirection, c);
                            if (dc == 0.f) continue;
                            float next;
                            if (dc > 0.f) {
                                next = comp(cellMax, c);
                                next = next == comp(cur, c) ? next + comp(cellSize, c) : next;
                            } else {
                                next = comp(cellMin, c);
                                next = next == comp(cur, c) ? next - comp(cellSize, c) : next;
                            }
                            const float dist = (next - comp(cur, c)) / dc;
                            if (dist < distanceToNext) { distanceToNext = dist; intersectedComponent = c; }
                        }
                        cur += distanceToNext * rayDirection;
                        gi[intersectedComponent] += comp(rayDirection, intersectedComponent) > 0.f ? 1u : 0xffffffffu;
                        rayIsInBoundingBox = gi[intersectedComponent] < 16u; // unsigned compare; ">= 0" always holds (:484-486)
                    }
                    if (isBackfaceHit) backHitCounter++;
                    closestHitTotal = glmMin(closestHitTotal, rayClosestHit);
                }
                const float backHitPercentage = (float)backHitCounter / 225.f;
                closestHitTotal *= backHitPercentage > 0.5f ? -1.f : 1.f;
                if (closestHitTotal == inf) closestHitTotal = computePointTrianglesClosestDistance(rayOrigin, all);
                outHalf[(size_t)flattenGridIndex(x, y, z, res)] = packHalfGlm(closestHitTotal);
            }
        }
    });
    return 0;
}
