// SDF bake on the GPU (include/plr_sdf_bake.h; reference AssetPipeline/SceneSDF.cpp:296-513, the CPU asset pipeline's computeSDF).
//
// Work decomposition: one workgroup per 4x4x4 voxel brick, lane = voxel, wave w of the four traces rays w, w+4, ... of the 225.
// All 64 lanes of a wave follow the SAME ray direction from neighbouring origins (at 64^3 a brick is exactly one cell of the
// 16^3 triangle grid), so the grid walk and the per-cell triangle loops are coherent: the triangle records a wave reads are
// the same addresses for (almost) all lanes. Per-voxel results (closest hit over rays, back-face count) are combined through
// LDS; min and integer add make the combination order-free, ties between equal distances (+-0) keep the lowest ray index as a
// sequential loop would. Voxels without any hit run the closest-triangle fallback with the triangle list split over the waves.
//
// The uniform grid is built on the host in triangle order (the per-cell order the reference's push_back produces) and stored
// as CSR: cell -> [offset, offset+count) into a list of triangle indices; triangles are 48-byte records (v0, v1, v2, N).
// Arithmetic follows the source operation order with IEEE divide/sqrt and no contraction (this file is built with the exact
// flag set), sin/cos/acos are the detmath contract functions, the half conversion is glm::packHalf's round-half-up.
#include <chrono>
#include <cmath>
#include <vector>

#include "../backend.h"
#include "../device/image.h"
#include "../../../include/plr_sdf_bake.h"

namespace plr {
namespace sdfbake {

struct Params {
    float bbMin[3];   // padded box
    float ext[3];     // padded extents (= VolumeInfo.extends)
    float offset[3];  // VolumeInfo.offset
    float cell[3];    // uniform grid cell size
    int32_t res[3];
    uint32_t triCount;
    uint32_t bricksX, bricksY, bricksZ;
};

struct Tri { float v0[3], v1[3], v2[3], N[3]; };

constexpr int kGrid = 16;
constexpr int kRays1D = 15, kRays = kRays1D * kRays1D;
constexpr int kRayWaves = 4;

// glm::min / glm::max (func_common.inl): the second operand wins only on a strict compare
PLR_DI float glmMin(float x, float y) { return (y < x) ? y : x; }
PLR_DI float glmMax(float x, float y) { return (x < y) ? y : x; }

// glm::packHalf1x16 (detail::toFloat16): round half up in magnitude, overflow -> infinity
PLR_DI uint32_t packHalfGlm(float v) {
    const uint32_t u = f2u(v);
    const uint32_t s = (u >> 16) & 0x8000u;
    const uint32_t a = u & 0x7fffffffu;
    if (a >= 0x7f800000u) {
        if (a == 0x7f800000u) return s | 0x7c00u;
        const uint32_t m = (a & 0x7fffffu) >> 13;
        return s | 0x7c00u | m | (m == 0u ? 1u : 0u);
    }
    const int e = (int)(a >> 23) - 112;
    if (e <= 0) {
        if (e < -10) return s;
        uint32_t m = ((a & 0x7fffffu) | 0x800000u) >> (1 - e);
        if (m & 0x1000u) m += 0x2000u;
        return s | (m >> 13);
    }
    const uint32_t r = a + 0x1000u; // carries from the dropped bits ripple into mantissa and exponent
    if ((int)(r >> 23) - 112 > 30) return s | 0x7c00u;
    return s | ((r - (112u << 23)) >> 13);
}

PLR_DI vec3 ld(const float* p) { return vec3(p[0], p[1], p[2]); }

// SceneSDF.cpp:237-244
PLR_DI void pointToCell(vec3 p, const Params& P, int* cx, int* cy, int* cz) {
    const vec3 rel = p - ld(P.bbMin);
    const float nx = glmMin(glmMax(rel.x / P.ext[0], 0.f), 0.999f);
    const float ny = glmMin(glmMax(rel.y / P.ext[1], 0.f), 0.999f);
    const float nz = glmMin(glmMax(rel.z / P.ext[2], 0.f), 0.999f);
    *cx = (int)floorf(nx * (float)kGrid); *cy = (int)floorf(ny * (float)kGrid); *cz = (int)floorf(nz * (float)kGrid);
}

__global__ __launch_bounds__(256) void sdfBakeKernel(Params P, const Tri* __restrict__ tris, const uint32_t* __restrict__ cellOffsets,
                                                     const uint32_t* __restrict__ cellTris, uint16_t* __restrict__ out) {
    __shared__ float dirs[kRays][3];
    __shared__ float redT[kRayWaves][64];
    __shared__ int redRay[kRayWaves][64];
    __shared__ uint32_t redBack[kRayWaves][64];
    const int tid = (int)threadIdx.x, wave = tid >> 6, lane = tid & 63;

    // the 225 directions (:351-364): phi = x * 2 * 3.1415, theta = acos(y), through degrees and back (MathUtils.cpp:4-15)
    if (tid < kRays) {
        const int sx = tid / kRays1D, sy = tid % kRays1D;
        const float sampleX = (float)sx / (float)(kRays1D - 1);
        const float sampleY = (float)sy / (float)(kRays1D - 1) * 2.f - 1.f;
        const float phi = sampleX * 2.f * 3.1415f;
        const float theta = det_acosf(sampleY);
        const float thetaR = (theta / 3.1415f * 180.f) * 0.01745329251994329576923690768489f;
        const float phiR = (phi / 3.1415f * 180.f) * 0.01745329251994329576923690768489f;
        float st, ct, sp, cp;
        det_sincosf(thetaR, &st, &ct);
        det_sincosf(phiR, &sp, &cp);
        dirs[tid][0] = st * cp; dirs[tid][1] = -ct; dirs[tid][2] = st * sp;
    }
    __syncthreads();

    const uint32_t brick = blockIdx.x;
    const int bx = (int)(brick % P.bricksX), by = (int)((brick / P.bricksX) % P.bricksY), bz = (int)(brick / (P.bricksX * P.bricksY));
    const int x = bx * 4 + (lane & 3), y = by * 4 + ((lane >> 2) & 3), z = bz * 4 + (lane >> 4);
    const bool inVolume = x < P.res[0] && y < P.res[1] && z < P.res[2];
    const float inf = __builtin_inff();

    // volumeIndexToCellCenter (:246-251)
    const vec3 origin((((float)x + 0.5f) / (float)P.res[0] - 0.5f) * P.ext[0] + P.offset[0], (((float)y + 0.5f) / (float)P.res[1] - 0.5f) * P.ext[1] + P.offset[1],
                      (((float)z + 0.5f) / (float)P.res[2] - 0.5f) * P.ext[2] + P.offset[2]);
    int sx0, sy0, sz0;
    pointToCell(origin, P, &sx0, &sy0, &sz0);

    float closest = inf;
    int closestRay = kRays;
    uint32_t backHits = 0;
    if (inVolume) {
        for (int ray = wave; ray < kRays; ray += kRayWaves) {
            const vec3 dir(dirs[ray][0], dirs[ray][1], dirs[ray][2]);
            bool backface = false;
            float rayClosest = inf;
            int gx = sx0, gy = sy0, gz = sz0;
            vec3 cur = origin;
            for (;;) {
                const uint32_t cellIndex = (uint32_t)(gx + gy * kGrid + gz * kGrid * kGrid);
                const vec3 cellMin(P.bbMin[0] + (float)gx / (float)kGrid * P.ext[0], P.bbMin[1] + (float)gy / (float)kGrid * P.ext[1], P.bbMin[2] + (float)gz / (float)kGrid * P.ext[2]);
                const vec3 cellMax = cellMin + ld(P.cell);
                bool hitTriangle = false;
                const uint32_t t0 = cellOffsets[cellIndex], t1 = cellOffsets[cellIndex + 1];
                for (uint32_t k = t0; k < t1; k++) {
                    const Tri& tri = tris[cellTris[k]];
                    const vec3 N = ld(tri.N), v0 = ld(tri.v0), v1 = ld(tri.v1), v2 = ld(tri.v2);
                    const float NoR = dot(N, dir);
                    if (fabsf(NoR) < 0.0001f) continue;
                    const float D = dot(N, v0);
                    const float t = (D - dot(N, origin)) / NoR;
                    if (t < 0.f) continue;
                    const vec3 pI = origin + dir * t;
                    const float d0 = dot(N, cross(pI - v0, v1 - v0));
                    const float d1 = dot(N, cross(pI - v1, v2 - v1));
                    const float d2 = dot(N, cross(pI - v2, v0 - v2));
                    if (!(d0 >= 0.f && d1 >= 0.f && d2 >= 0.f)) continue;
                    const vec3 hp = origin + t * dir;
                    if (!(hp.x <= cellMax.x && hp.x >= cellMin.x && hp.y <= cellMax.y && hp.y >= cellMin.y && hp.z <= cellMax.z && hp.z >= cellMin.z)) continue;
                    hitTriangle = true;
                    if (t < rayClosest) { rayClosest = t; backface = dot(dir, N) > 0.f; }
                }
                if (hitTriangle) break;
                // next cell boundary (:437-470)
                float distNext = inf;
                int comp = 0;
                {
                    const float dc[3] = {dir.x, dir.y, dir.z}, cmax[3] = {cellMax.x, cellMax.y, cellMax.z}, cmin[3] = {cellMin.x, cellMin.y, cellMin.z}, cu[3] = {cur.x, cur.y, cur.z};
#pragma unroll
                    for (int c = 0; c < 3; c++) {
                        if (dc[c] == 0.f) continue;
                        float next;
                        if (dc[c] > 0.f) { next = cmax[c]; next = next == cu[c] ? next + P.cell[c] : next; }
                        else { next = cmin[c]; next = next == cu[c] ? next - P.cell[c] : next; }
                        const float dist = (next - cu[c]) / dc[c];
                        if (dist < distNext) { distNext = dist; comp = c; }
                    }
                }
                cur = cur + distNext * dir;
                const int stepDir = (comp == 0 ? dir.x : (comp == 1 ? dir.y : dir.z)) > 0.f ? 1 : -1;
                int moved;
                if (comp == 0) { gx += stepDir; moved = gx; } else if (comp == 1) { gy += stepDir; moved = gy; } else { gz += stepDir; moved = gz; }
                if ((uint32_t)moved >= (uint32_t)kGrid) break; // unsigned index left the grid (:484-486)
            }
            if (backface) backHits++;
            if (rayClosest < closest) { closest = rayClosest; closestRay = ray; }
        }
    }
    redT[wave][lane] = closest; redRay[wave][lane] = closestRay; redBack[wave][lane] = backHits;
    __syncthreads();

    // every wave recomputes the voxel's combined result (same for all four): sequential-order semantics of glm::min over rays
    float total = redT[0][lane];
    int totalRay = redRay[0][lane];
    uint32_t back = redBack[0][lane];
#pragma unroll
    for (int w = 1; w < kRayWaves; w++) {
        const float t = redT[w][lane];
        const int r = redRay[w][lane];
        if (t < total || (t == total && r < totalRay)) { total = t; totalRay = r; }
        back += redBack[w][lane];
    }
    const float backHitPercentage = (float)back / (float)kRays;
    total *= backHitPercentage > 0.5f ? -1.f : 1.f;
    const bool noHit = inVolume && total == inf;
    __syncthreads(); // redT is reused below

    // computePointTrianglesClosestDistance (:55-95), triangles w, w+4, ... per wave
    float closestD = inf;
    if (noHit) {
        for (uint32_t i = (uint32_t)wave; i < P.triCount; i += kRayWaves) {
            const Tri& tri = tris[i];
            const vec3 N = ld(tri.N), v0 = ld(tri.v0), v1 = ld(tri.v1), v2 = ld(tri.v2);
            const vec3 p = origin;
            const vec3 v1ToP = p - v0, v2ToP = p - v1, v3ToP = p - v2;
            const vec3 e0 = v1 - v0, e1 = v2 - v1, e2 = v0 - v2;
            const float s1 = gsign(dot(cross(e0, N), -v1ToP)), s2 = gsign(dot(cross(e1, N), -v2ToP)), s3 = gsign(dot(cross(e2, N), -v3ToP));
            const bool onEdge = s1 + s2 + s3 < 2.f;
            const float c1 = glmMin(glmMax(dot(v1ToP, e0) / dot(e0, e0), 0.f), 1.f);
            const float c2 = glmMin(glmMax(dot(v2ToP, e1) / dot(e1, e1), 0.f), 1.f);
            const float c3 = glmMin(glmMax(dot(v3ToP, e2) / dot(e2, e2), 0.f), 1.f);
            const vec3 q1 = p - (v0 + e0 * c1), q2 = p - (v1 + e1 * c2), q3 = p - (v2 + e2 * c3);
            const float l1 = dot(q1, q1), l2 = dot(q2, q2), l3 = dot(q3, q3);
            const float pd = dot(N, v1ToP);
            const float d = fabsf(onEdge ? glmMin(glmMin(l1, l2), l3) : fabsf(pd * pd));
            closestD = glmMin(closestD, d);
        }
    }
    redT[wave][lane] = closestD;
    __syncthreads();
    if (wave == 0 && inVolume) {
        float result = total;
        if (noHit) {
            const float d = glmMin(glmMin(glmMin(redT[0][lane], redT[1][lane]), redT[2][lane]), redT[3][lane]);
            result = sqrtf(fabsf(d));
        }
        out[(size_t)x + (size_t)y * (size_t)P.res[0] + (size_t)z * (size_t)P.res[0] * (size_t)P.res[1]] = (uint16_t)packHalfGlm(result);
    }
}

// ------------------------------------------------------------------------------------------------ host side
struct V3 { float x, y, z; };
static inline V3 sub(V3 a, V3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
static inline V3 crossH(V3 a, V3 b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
static inline float dotH(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
static inline float minH(float x, float y) { return (y < x) ? y : x; }
static inline float maxH(float x, float y) { return (x < y) ? y : x; }

static inline bool axisSeparates(V3 axis, V3 half, V3 v0, V3 v1, V3 v2) { // SceneSDF.cpp:160-173
    const float p0 = dotH(axis, v0), p1 = dotH(axis, v1), p2 = dotH(axis, v2);
    const float r = dotH({std::fabs(axis.x), std::fabs(axis.y), std::fabs(axis.z)}, half);
    return minH(minH(p0, p1), p2) > r || maxH(maxH(p0, p1), p2) < -r;
}

static bool triangleOverlapsBox(V3 centre, V3 size, const Tri& t) { // SceneSDF.cpp:178-231
    const V3 v0 = sub({t.v0[0], t.v0[1], t.v0[2]}, centre), v1 = sub({t.v1[0], t.v1[1], t.v1[2]}, centre), v2 = sub({t.v2[0], t.v2[1], t.v2[2]}, centre);
    const V3 e[3] = {sub(v1, v0), sub(v2, v1), sub(v0, v2)};
    const V3 half{size.x * 0.5f, size.y * 0.5f, size.z * 0.5f};
    const V3 ax[3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}};
    for (const V3& edge : e)
        for (const V3& a : ax)
            if (axisSeparates(crossH(a, edge), half, v0, v1, v2)) return false;
    for (const V3& a : ax)
        if (axisSeparates(a, half, v0, v1, v2)) return false;
    return !axisSeparates({t.N[0], t.N[1], t.N[2]}, half, v0, v1, v2);
}

struct PaddedVolume { V3 mn, mx, ext, offset; };

static PaddedVolume padBounds(const plr_aabb& bb) { // sdfUtilities.cpp:5-19, VolumeInfo.cpp:4-9
    PaddedVolume p;
    const float lo[3] = {bb.min[0], bb.min[1], bb.min[2]}, hi[3] = {bb.max[0], bb.max[1], bb.max[2]};
    float mn[3], mx[3];
    for (int c = 0; c < 3; c++) {
        const float pad = maxH(0.075f * (hi[c] - lo[c]), 0.5f);
        mn[c] = lo[c] - pad;
        mx[c] = hi[c] + pad;
    }
    p.mn = {mn[0], mn[1], mn[2]};
    p.mx = {mx[0], mx[1], mx[2]};
    p.ext = sub(p.mx, p.mn);
    p.offset = {(mx[0] + mn[0]) * 0.5f, (mx[1] + mn[1]) * 0.5f, (mx[2] + mn[2]) * 0.5f};
    return p;
}

static inline void cellOfPoint(V3 p, const PaddedVolume& pv, int out[3]) { // SceneSDF.cpp:237-244
    const float rel[3] = {p.x - pv.mn.x, p.y - pv.mn.y, p.z - pv.mn.z}, ext[3] = {pv.ext.x, pv.ext.y, pv.ext.z};
    for (int c = 0; c < 3; c++) {
        const float n = minH(maxH(rel[c] / ext[c], 0.f), 0.999f);
        out[c] = (int)std::floor(n * (float)kGrid);
    }
}

static uint32_t nextPowerOfTwo(uint32_t v) { // SceneSDF.cpp:42-53
    v--;
    v |= v >> 1; v |= v >> 2; v |= v >> 4; v |= v >> 8; v |= v >> 16;
    return v + 1;
}

static thread_local float g_lastKernelMs = 0.f;

#define BAKE_HIP(x)                                                                                           \
    do {                                                                                                      \
        hipError_t e_ = (x);                                                                                  \
        if (e_ != hipSuccess) { rc = setLastError(-2, std::string(#x) + ": " + hipGetErrorString(e_)); goto done; } \
    } while (0)

static int computeSdf(int device, const plr_mesh_data* mesh, const plr_aabb* bounds, uint32_t w, uint32_t h, uint32_t d, void* outData, size_t outSize) {
    if (!mesh || !bounds || !outData) return setLastError(-1, "plr_compute_sdf: null argument");
    if (w == 0 || h == 0 || d == 0 || w > 1024 || h > 1024 || d > 1024) return setLastError(-1, "plr_compute_sdf: resolution must be 1..1024 per axis");
    if (mesh->index_count % 3u != 0u) return setLastError(-1, "plr_compute_sdf: index count is not a multiple of 3");
    if (mesh->index_count && (!mesh->indices || !mesh->positions)) return setLastError(-1, "plr_compute_sdf: null mesh arrays");
    const size_t voxels = (size_t)w * h * d;
    if (outSize < voxels * 2) return setLastError(-1, "plr_compute_sdf: output buffer too small");
    for (uint32_t i = 0; i < mesh->index_count; i++)
        if (mesh->indices[i] >= mesh->vertex_count) return setLastError(-1, "plr_compute_sdf: vertex index out of range");

    const PaddedVolume pv = padBounds(*bounds);
    const V3 cell{pv.ext.x / (float)kGrid, pv.ext.y / (float)kGrid, pv.ext.z / (float)kGrid};

    // triangles + CSR grid (two passes in triangle order, so the per-cell order is ascending triangle index)
    const uint32_t triCount = mesh->index_count / 3u;
    std::vector<Tri> tris(triCount);
    for (uint32_t i = 0; i < triCount; i++) {
        Tri& t = tris[i];
        for (int c = 0; c < 3; c++) {
            t.v0[c] = mesh->positions[3 * (size_t)mesh->indices[3 * i] + c];
            t.v1[c] = mesh->positions[3 * (size_t)mesh->indices[3 * i + 1] + c];
            t.v2[c] = mesh->positions[3 * (size_t)mesh->indices[3 * i + 2] + c];
        }
        const V3 n = crossH(sub({t.v0[0], t.v0[1], t.v0[2]}, {t.v2[0], t.v2[1], t.v2[2]}), sub({t.v0[0], t.v0[1], t.v0[2]}, {t.v1[0], t.v1[1], t.v1[2]}));
        const float inv = 1.0f / std::sqrt(dotH(n, n)); // glm::normalize = v * inversesqrt(dot(v, v))
        t.N[0] = n.x * inv; t.N[1] = n.y * inv; t.N[2] = n.z * inv;
    }
    std::vector<uint32_t> offsets(kGrid * kGrid * kGrid + 1, 0u);
    std::vector<uint32_t> pairs; // (cell, triangle) in triangle order
    pairs.reserve((size_t)triCount * 4);
    for (uint32_t i = 0; i < triCount; i++) {
        const Tri& t = tris[i];
        const V3 tMin{minH(minH(t.v0[0], t.v1[0]), t.v2[0]), minH(minH(t.v0[1], t.v1[1]), t.v2[1]), minH(minH(t.v0[2], t.v1[2]), t.v2[2])};
        const V3 tMax{maxH(maxH(t.v0[0], t.v1[0]), t.v2[0]), maxH(maxH(t.v0[1], t.v1[1]), t.v2[1]), maxH(maxH(t.v0[2], t.v1[2]), t.v2[2])};
        int lo[3], hi[3];
        cellOfPoint(tMin, pv, lo);
        cellOfPoint(tMax, pv, hi);
        for (int x = lo[0]; x <= hi[0]; x++)
            for (int y = lo[1]; y <= hi[1]; y++)
                for (int z = lo[2]; z <= hi[2]; z++) {
                    const V3 centre{(((float)x + 0.5f) / (float)kGrid - 0.5f) * pv.ext.x + pv.offset.x, (((float)y + 0.5f) / (float)kGrid - 0.5f) * pv.ext.y + pv.offset.y,
                                    (((float)z + 0.5f) / (float)kGrid - 0.5f) * pv.ext.z + pv.offset.z};
                    if (!triangleOverlapsBox(centre, cell, t)) continue;
                    const uint32_t cellIndex = (uint32_t)(x + y * kGrid + z * kGrid * kGrid);
                    pairs.push_back(cellIndex);
                    pairs.push_back(i);
                    offsets[cellIndex + 1]++;
                }
    }
    for (size_t c = 0; c < (size_t)kGrid * kGrid * kGrid; c++) offsets[c + 1] += offsets[c];
    std::vector<uint32_t> cellTris(pairs.size() / 2);
    {
        std::vector<uint32_t> cursor(offsets.begin(), offsets.end() - 1);
        for (size_t k = 0; k < pairs.size(); k += 2) cellTris[cursor[pairs[k]]++] = pairs[k + 1];
    }

    Params P;
    P.bbMin[0] = pv.mn.x; P.bbMin[1] = pv.mn.y; P.bbMin[2] = pv.mn.z;
    P.ext[0] = pv.ext.x; P.ext[1] = pv.ext.y; P.ext[2] = pv.ext.z;
    P.offset[0] = pv.offset.x; P.offset[1] = pv.offset.y; P.offset[2] = pv.offset.z;
    P.cell[0] = cell.x; P.cell[1] = cell.y; P.cell[2] = cell.z;
    P.res[0] = (int32_t)w; P.res[1] = (int32_t)h; P.res[2] = (int32_t)d;
    P.triCount = triCount;
    P.bricksX = divUp(w, 4u); P.bricksY = divUp(h, 4u); P.bricksZ = divUp(d, 4u);

    int rc = 0;
    Tri* dTris = nullptr;
    uint32_t *dOffsets = nullptr, *dCellTris = nullptr;
    uint16_t* dOut = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    hipStream_t stream = nullptr;
    BAKE_HIP(hipSetDevice(device));
    BAKE_HIP(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking));
    BAKE_HIP(hipEventCreate(&ev0));
    BAKE_HIP(hipEventCreate(&ev1));
    BAKE_HIP(hipMalloc(&dTris, std::max<size_t>(tris.size(), 1) * sizeof(Tri)));
    BAKE_HIP(hipMalloc(&dOffsets, offsets.size() * 4));
    BAKE_HIP(hipMalloc(&dCellTris, std::max<size_t>(cellTris.size(), 1) * 4));
    BAKE_HIP(hipMalloc(&dOut, voxels * 2));
    if (!tris.empty()) BAKE_HIP(hipMemcpyAsync(dTris, tris.data(), tris.size() * sizeof(Tri), hipMemcpyHostToDevice, stream));
    BAKE_HIP(hipMemcpyAsync(dOffsets, offsets.data(), offsets.size() * 4, hipMemcpyHostToDevice, stream));
    if (!cellTris.empty()) BAKE_HIP(hipMemcpyAsync(dCellTris, cellTris.data(), cellTris.size() * 4, hipMemcpyHostToDevice, stream));
    BAKE_HIP(hipEventRecord(ev0, stream));
    sdfBakeKernel<<<P.bricksX * P.bricksY * P.bricksZ, 256, 0, stream>>>(P, dTris, dOffsets, dCellTris, dOut);
    BAKE_HIP(hipGetLastError());
    BAKE_HIP(hipEventRecord(ev1, stream));
    BAKE_HIP(hipMemcpyAsync(outData, dOut, voxels * 2, hipMemcpyDeviceToHost, stream));
    BAKE_HIP(hipStreamSynchronize(stream));
    BAKE_HIP(hipEventElapsedTime(&g_lastKernelMs, ev0, ev1));
done:
    if (dTris) (void)hipFree(dTris);
    if (dOffsets) (void)hipFree(dOffsets);
    if (dCellTris) (void)hipFree(dCellTris);
    if (dOut) (void)hipFree(dOut);
    if (ev0) (void)hipEventDestroy(ev0);
    if (ev1) (void)hipEventDestroy(ev1);
    if (stream) (void)hipStreamDestroy(stream);
    return rc;
}

} // namespace sdfbake
} // namespace plr

using namespace plr;

extern "C" int plr_sdf_texture_description(const plr_aabb* bb, plr_image_desc* out) {
    if (!bb || !out) return setLastError(-1, "plr_sdf_texture_description: null argument");
    uint32_t res[3];
    for (int c = 0; c < 3; c++) {
        const float target = (bb->max[c] - bb->min[c]) / 0.25f;
        const uint32_t r = sdfbake::nextPowerOfTwo((uint32_t)target);
        res[c] = r < 16u ? 16u : (r > 64u ? 64u : r);
    }
    out->width = res[0]; out->height = res[1]; out->depth = res[2];
    out->type = PLR_IMAGE_3D;
    out->format = PLR_FORMAT_R16_SFLOAT;
    out->usage_flags = PLR_USAGE_STORAGE | PLR_USAGE_SAMPLED;
    out->mip_count = PLR_MIP_ONE;
    out->manual_mip_count = 1;
    out->auto_create_mips = 0;
    return 0;
}

extern "C" int plr_sdf_padded_bounds(const plr_aabb* bb, plr_aabb* out) {
    if (!bb || !out) return setLastError(-1, "plr_sdf_padded_bounds: null argument");
    const sdfbake::PaddedVolume pv = sdfbake::padBounds(*bb);
    out->min[0] = pv.mn.x; out->min[1] = pv.mn.y; out->min[2] = pv.mn.z;
    out->max[0] = pv.mx.x; out->max[1] = pv.mx.y; out->max[2] = pv.mx.z;
    return 0;
}

extern "C" int plr_compute_sdf(int device, const plr_mesh_data* mesh, const plr_aabb* bb, uint32_t w, uint32_t h, uint32_t d, void* outData, size_t outSize) {
    return sdfbake::computeSdf(device, mesh, bb, w, h, d, outData, outSize);
}

extern "C" int plr_compute_scene_sdf_textures(int device, const plr_mesh_data* meshes, const plr_aabb* bounds, uint32_t count, plr_image_desc* descs,
                                              void* const* outData, const size_t* outSizes, double* outSeconds) {
    if ((!meshes || !bounds || !descs || !outData || !outSizes) && count) return setLastError(-1, "plr_compute_scene_sdf_textures: null argument");
    const auto start = std::chrono::steady_clock::now();
    for (uint32_t i = 0; i < count; i++) {
        if (int rc = plr_sdf_texture_description(&bounds[i], &descs[i])) return rc;
        if (!outData[i]) continue;
        if (int rc = sdfbake::computeSdf(device, &meshes[i], &bounds[i], descs[i].width, descs[i].height, descs[i].depth, outData[i], outSizes[i])) return rc;
    }
    if (outSeconds) *outSeconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - start).count();
    return 0;
}

extern "C" int plr_sdf_last_kernel_ms(float* outMs) {
    if (!outMs) return setLastError(-1, "plr_sdf_last_kernel_ms: null argument");
    *outMs = sdfbake::g_lastKernelMs;
    return 0;
}
