// PLR_MATH_FAST variant of sdfDiffuseTrace.comp (exact variant and the wave / tile mapping: kernels_exact/sdf_trace_exact.hip).
//
// Same sphere trace, same culling tiles, same LDS ray exchange. Arithmetic changes:
//  * the per-step uv = localPos / localExtends + 0.5 is a multiply-add with the precomputed reciprocal extents (three IEEE
//    divisions per step in the exact kernel); other divisions are v_rcp_f32, normalisations v_rsq_f32, FMA contraction is on
//  * the two x-neighbours of every trilinear corner pair are fetched with one 32-bit load when they are adjacent in memory
//  * the world position uses the un-normalised view ray (normalisation cancels), the cosine sample uses v_sin/v_cos,
//    directionToSH_L1 of a unit vector is two constants
// A ray whose closest SDF sample sits within rounding of the hit threshold (or of an AABB face) can resolve differently than in
// the exact kernel; such a pixel and the neighbours that share its ray in the 3x3 resolve change visibly. Stated tolerance:
// tests/test_fast_kernels.py.
//
// One thing is NOT approximate: uv = pixel / imageSize (:121, no half-texel offset) lands exactly on a texel boundary of the full-resolution
// depth / normal images, so floor(uv * size) depends on the last bit of the quotient: it is an IEEE division here as in the shader (an
// approximate quotient fetches the neighbouring G-buffer texel for a third of the pixels). Everything in the march uses v_rcp / v_rsq / v_sqrt.
// PLR_BUILD_FLAGS: -fhip-fp32-correctly-rounded-divide-sqrt
#include "../backend.h"
#include "../device/shading_common.h"
#include "../device/fastmath.h"
#include "../device/sdf_bricks.h"
#include "fused_gi.h"
#include <cstdlib>
#include <map>
#include <vector>

namespace plr {
namespace fasttrace {

PLR_DI float rcpf(float x) { return __builtin_amdgcn_rcpf(x); }
PLR_DI float rsqf(float x) { return __builtin_amdgcn_rsqf(x); }

// Seven waves per SIMD (72 VGPRs, no scratch) instead of the six the allocator settles on by itself (80): the kernel waits on a chain of dependent fetches and
// occupancy is what hides it - measured 99.0 -> 95.0 us; at eight (64 VGPRs) it spills 36 bytes and takes 109 us, at five (an experiment that cost two more
// registers) 108 us (profiles/r04_not_kept.txt, profiles/r04_trace_variants.txt)
#ifndef PLR_TRACE_OCC
#define PLR_TRACE_OCC __attribute__((amdgpu_waves_per_eu(7, 7)))
#endif
// The volume texels live in global memory, but their address comes out of the LDS-staged instance table, so the compiler cannot tell and emits
// flat_load (which also counts on lgkmcnt). PLR_TRACE_GLOBAL_VOLUMES=1 types the pointer address_space(1): global_load with partial vmcnt waits.
#ifndef PLR_TRACE_GLOBAL_VOLUMES
#define PLR_TRACE_GLOBAL_VOLUMES 1 // measured, six runs each (profiles/r04_trace_variants.txt): 98.6 +- 0.4 us against 100.1 +- 0.5 us with flat loads
#endif
#if PLR_TRACE_GLOBAL_VOLUMES
typedef const __attribute__((address_space(1))) uint16_t* VolumeTexels;
#else
typedef const uint16_t* VolumeTexels;
#endif
struct Volume {
    VolumeTexels p;
    int w, h, d;
    float fw, fh, fd;
};

// Trilinear clamp-to-edge fetch for volumes of at least 2 texels per axis, without a branch: the sampler's 8.8 fixed-point coordinate
// is clamped to [0, n-1] and split into a cell index <= n-2 and a weight in [0, 1], so both texels of every pair are always in range
// (outside the first / last texel centre the weight saturates at 0 / 1, which is what clamping the two indices gives). The march starts
// every instance on a face of its box, where the clamped-index form diverges into eight single-texel loads.
// fx = (u * n - 0.5) * 256 + 0.5 arrives already folded into one multiply-add of the local position (see traceInstance)
PLR_DI void cellCoord(float fx, int n, int* i0, float* a) {
    const int ti = clampTo(floorToInt(fx), (n - 1) << 8); // the same integer as (int)floorf(clamp(fx)) clamped: the conversion saturates, NaN -> 0
    *i0 = min(ti >> 8, n - 2);
    *a = (float)(ti - (*i0 << 8)) * (1.0f / 256.0f);
}
PLR_DI float sampleSDFInterior(const Volume& v, vec3 fixedPoint) {
    int i0, j0, k0; float a, b, c;
    cellCoord(fixedPoint.x, v.w, &i0, &a);
    cellCoord(fixedPoint.y, v.h, &j0, &b);
    cellCoord(fixedPoint.z, v.d, &k0, &c);
    const int sl = v.w * v.h;
    const int o00 = k0 * sl + j0 * v.w + i0, o10 = o00 + v.w, o01 = o00 + sl, o11 = o01 + v.w;
    auto pair = [&](int off, float& lo, float& hi) {
        uint32_t u32;
        __builtin_memcpy(&u32, v.p + off, 4);
        lo = halfBitsToFloat(u32 & 0xffffu); hi = halfBitsToFloat(u32 >> 16);
    };
    float t000, t100, t010, t110, t001, t101, t011, t111;
    pair(o00, t000, t100); pair(o10, t010, t110); pair(o01, t001, t101); pair(o11, t011, t111);
    const float x00 = t000 + (t100 - t000) * a, x10 = t010 + (t110 - t010) * a, x01 = t001 + (t101 - t001) * a, x11 = t011 + (t111 - t011) * a;
    const float y0v = x00 + (x10 - x00) * b, y1v = x01 + (x11 - x01) * b;
    return y0v + (y1v - y0v) * c;
}

// the same fetch from the volume's copy in cache-line bricks (device/sdf_bricks.h, the BRICKS variant of the kernel): four 32-bit loads as before, in 1.9 cache
// lines on average instead of always 4
PLR_DI float sampleSDFInteriorBricked(const Volume& v, vec3 fixedPoint) {
    int i0, j0, k0; float a, b, c;
    cellCoord(fixedPoint.x, v.w, &i0, &a);
    cellCoord(fixedPoint.y, v.h, &j0, &b);
    cellCoord(fixedPoint.z, v.d, &k0, &c);
    const int nbx = div7(v.w + 5), nby = (v.h + 3) >> 2;                    // brickGrid(w, h, d)
    const int rowBricks = nbx << 6, sliceBricks = __mul24(nby, rowBricks);  // texels per row / slice of bricks
    const int bx = div7(i0), xo = (bx << 6) + (i0 - 7 * bx);
    const int j1 = j0 + 1, k1 = k0 + 1;
    const int y0 = __mul24(j0 >> 2, rowBricks) + ((j0 & 3) << 3), y1 = __mul24(j1 >> 2, rowBricks) + ((j1 & 3) << 3);
    const int z0 = __mul24(k0 >> 1, sliceBricks) + ((k0 & 1) << 5), z1 = __mul24(k1 >> 1, sliceBricks) + ((k1 & 1) << 5);
    auto pair = [&](int off, float& lo, float& hi) {
        uint32_t u32;
        __builtin_memcpy(&u32, v.p + off, 4);
        lo = halfBitsToFloat(u32 & 0xffffu); hi = halfBitsToFloat(u32 >> 16);
    };
    float t000, t100, t010, t110, t001, t101, t011, t111;
    pair(z0 + y0 + xo, t000, t100); pair(z0 + y1 + xo, t010, t110); pair(z1 + y0 + xo, t001, t101); pair(z1 + y1 + xo, t011, t111);
    const float x00 = t000 + (t100 - t000) * a, x10 = t010 + (t110 - t010) * a, x01 = t001 + (t101 - t001) * a, x11 = t011 + (t111 - t011) * a;
    const float y0v = x00 + (x10 - x00) * b, y1v = x01 + (x11 - x01) * b;
    return y0v + (y1v - y0v) * c;
}

PLR_DI float sampleSDF(const Volume& v, float u, float vv, float ww) {
    int i0, j0, k0; float a, b, c;
    linearCoord(u * v.fw, &i0, &a);
    linearCoord(vv * v.fh, &j0, &b);
    linearCoord(ww * v.fd, &k0, &c);
    const int y0 = clampi(j0, v.h) * v.w, y1 = clampi(j0 + 1, v.h) * v.w;
    const int sl = v.w * v.h;
    const int z0 = clampi(k0, v.d) * sl, z1 = clampi(k0 + 1, v.d) * sl;
    float t000, t100, t010, t110, t001, t101, t011, t111;
    if (i0 >= 0 && i0 + 1 < v.w) {
        // both x texels in range: adjacent halves, one (possibly unaligned) dword load per row
        auto pair = [&](int off, float& lo, float& hi) {
            uint32_t u32;
            __builtin_memcpy(&u32, v.p + off + i0, 4);
            lo = halfBitsToFloat(u32 & 0xffffu); hi = halfBitsToFloat(u32 >> 16);
        };
        pair(z0 + y0, t000, t100); pair(z0 + y1, t010, t110); pair(z1 + y0, t001, t101); pair(z1 + y1, t011, t111);
    } else {
        const int x0 = clampi(i0, v.w), x1 = clampi(i0 + 1, v.w);
        t000 = halfBitsToFloat(v.p[z0 + y0 + x0]); t100 = halfBitsToFloat(v.p[z0 + y0 + x1]);
        t010 = halfBitsToFloat(v.p[z0 + y1 + x0]); t110 = halfBitsToFloat(v.p[z0 + y1 + x1]);
        t001 = halfBitsToFloat(v.p[z1 + y0 + x0]); t101 = halfBitsToFloat(v.p[z1 + y0 + x1]);
        t011 = halfBitsToFloat(v.p[z1 + y1 + x0]); t111 = halfBitsToFloat(v.p[z1 + y1 + x1]);
    }
    const float x00 = t000 + (t100 - t000) * a, x10 = t010 + (t110 - t010) * a, x01 = t001 + (t101 - t001) * a, x11 = t011 + (t111 - t011) * a;
    const float y0v = x00 + (x10 - x00) * b, y1v = x01 + (x11 - x01) * b;
    return y0v + (y1v - y0v) * c;
}

struct TraceResult {
    bool hit;
    float closestHitDistance;
    vec3 hitPos;
    vec3 albedoSrgb; // meanAlbedo of the closest hit; linearised once after the loop
};

PLR_DI bool rayAABBIntersection(vec3 o, vec3 dir, vec3 mx, float* tOut) {
    // the three slab tests as selects (a face test is a handful of instructions; three divergent branches cost more)
    const float tx = ((o.x < 0.f ? -mx.x : mx.x) - o.x) * rcpf(dir.x);
    const float ty = ((o.y < 0.f ? -mx.y : mx.y) - o.y) * rcpf(dir.y);
    const float tz = ((o.z < 0.f ? -mx.z : mx.z) - o.z) * rcpf(dir.z);
    const vec3 px = o + tx * dir, py = o + ty * dir, pz = o + tz * dir;
    const bool hx = tx > 0.f && fabsf(px.y) <= mx.y && fabsf(px.z) <= mx.z;
    const bool hy = ty > 0.f && fabsf(py.x) <= mx.x && fabsf(py.z) <= mx.z;
    const bool hz = tz > 0.f && fabsf(pz.x) <= mx.x && fabsf(pz.y) <= mx.y;
    float t = 100000.f;
    t = hx ? gmin(t, tx) : t;
    t = hy ? gmin(t, ty) : t;
    t = hz ? gmin(t, tz) : t;
    *tOut = t;
    return hx || hy || hz;
}

// A ray against one instance, in two steps (SDF.inc:101-184). enterInstance: the ray in the instance's local frame and its entry into the volume's
// box (SDF.inc:104-125) - cheap, no memory access. marchInstance: the sphere trace from that entry point.
struct LocalRay { vec3 start, dir; float entryDistance; };
PLR_DI bool enterInstance(const SDFInstance& inst, vec3 rayStartWorld, vec3 rayDirectionWorld, LocalRay* out) {
    const float* m = inst.worldToLocal;
    const vec3 localExtends = ld3(inst.localExtends);
    vec3 rayStartLocal(m[0] * rayStartWorld.x + m[4] * rayStartWorld.y + m[8] * rayStartWorld.z + m[12],
                       m[1] * rayStartWorld.x + m[5] * rayStartWorld.y + m[9] * rayStartWorld.z + m[13],
                       m[2] * rayStartWorld.x + m[6] * rayStartWorld.y + m[10] * rayStartWorld.z + m[14]);
    // (M * (start + dir)) - (M * start) = linear part of M applied to dir
    vec3 rayDirection(m[0] * rayDirectionWorld.x + m[4] * rayDirectionWorld.y + m[8] * rayDirectionWorld.z,
                      m[1] * rayDirectionWorld.x + m[5] * rayDirectionWorld.y + m[9] * rayDirectionWorld.z,
                      m[2] * rayDirectionWorld.x + m[6] * rayDirectionWorld.y + m[10] * rayDirectionWorld.z);
    rayDirection = rayDirection * rsqf(dot(rayDirection, rayDirection));
    const vec3 sdfMaxLocal = localExtends * 0.5f;
    float hitDistanceLocal = 0.f;
    const bool inside = fabsf(rayStartLocal.x) <= sdfMaxLocal.x && fabsf(rayStartLocal.y) <= sdfMaxLocal.y && fabsf(rayStartLocal.z) <= sdfMaxLocal.z;
    bool enters = true;
    if (!inside) {
        float t;
        enters = rayAABBIntersection(rayStartLocal, rayDirection, sdfMaxLocal, &t);
        rayStartLocal += t * rayDirection;
        hitDistanceLocal = t;
    }
    out->start = rayStartLocal; out->dir = rayDirection; out->entryDistance = hitDistanceLocal;
    return enters;
}

// the state one lane carries through a march; `go` = this lane is still marching
struct MarchState {
    VolumeTexels p; int w, h, d;
    vec3 pos, dir, fpScale, fpBias, lim, invExt;
    float distanceThreshold, localToGlobalScale, hitDistanceLocal, dv, dLast;
    bool thick, bricked; // bricked: only the BRICKS variant of the kernel sets and reads it
};
template <bool BRICKS = false>
PLR_DI bool beginMarch(const SDFInstance& inst, const ImgView& view, const LocalRay& ray, float closestHitDistance, MarchState* st) {
    const float* m = inst.worldToLocal;
    const vec3 localExtends = ld3(inst.localExtends);
    st->localToGlobalScale = rsqf(m[0] * m[0] + m[1] * m[1] + m[2] * m[2]);
    if (st->localToGlobalScale * ray.entryDistance > closestHitDistance) return false;
    const float fw = (float)view.w, fh = (float)view.h, fd = (float)view.d;
    st->p = (VolumeTexels)(const uint16_t*)view.ptr; st->w = view.w; st->h = view.h; st->d = view.d;
    st->invExt = vec3(rcpf(localExtends.x), rcpf(localExtends.y), rcpf(localExtends.z));
    const vec3 voxel(localExtends.x * rcpf(fw), localExtends.y * rcpf(fh), localExtends.z * rcpf(fd));
    st->distanceThreshold = __builtin_amdgcn_sqrtf(dot(voxel, voxel)) * 0.25f;
    st->lim = localExtends * 0.5f + 0.01f;
    st->pos = ray.start; st->dir = ray.dir;
    st->hitDistanceLocal = ray.entryDistance;
    st->dLast = 0.f; st->dv = 0.f;
    st->thick = view.w >= 2 && view.h >= 2 && view.d >= 2;
    st->bricked = BRICKS && (view.fmt & kBrickedFormatFlag) != 0; // staged by the kernel: only volumes of at least two texels per axis have a bricked copy
    // uvw = pos / extends + 0.5 and the sampler's fixed-point texel coordinate (uvw * n - 0.5) * 256 + 0.5 as one multiply-add per axis
    st->fpScale = vec3(st->invExt.x * 256.f * fw, st->invExt.y * 256.f * fh, st->invExt.z * 256.f * fd);
    st->fpBias = vec3(128.f * fw - 127.5f, 128.f * fh - 127.5f, 128.f * fd - 127.5f);
    return true;
}
// one step of SDF.inc:139-181 for a marching lane; false = this lane's march is over (left the box, or hit: then tr is updated)
template <bool BRICKS = false>
PLR_DI bool marchStep(MarchState& st, const SDFInstance& inst, vec3 rayStartWorld, vec3 rayDirectionWorld, TraceResult& tr) {
    if (fabsf(st.pos.x) > st.lim.x || fabsf(st.pos.y) > st.lim.y || fabsf(st.pos.z) > st.lim.z) return false;
    Volume sdf;
    sdf.p = st.p; sdf.w = st.w; sdf.h = st.h; sdf.d = st.d; sdf.fw = (float)st.w; sdf.fh = (float)st.h; sdf.fd = (float)st.d;
    st.dLast = st.dv;
    const vec3 fixedPoint(st.pos.x * st.fpScale.x + st.fpBias.x, st.pos.y * st.fpScale.y + st.fpBias.y, st.pos.z * st.fpScale.z + st.fpBias.z);
    st.dv = (BRICKS && st.bricked) ? sampleSDFInteriorBricked(sdf, fixedPoint)
          : st.thick ? sampleSDFInterior(sdf, fixedPoint)
                     : sampleSDF(sdf, st.pos.x * st.invExt.x + 0.5f, st.pos.y * st.invExt.y + 0.5f, st.pos.z * st.invExt.z + 0.5f);
    if (st.dv < st.distanceThreshold) {
        tr.hit = true;
        const float distanceGlobal = st.hitDistanceLocal * st.localToGlobalScale;
        if (distanceGlobal < tr.closestHitDistance) {
            tr.closestHitDistance = distanceGlobal;
            const float lastStepSizeLocal = st.dv * rcpf(1.f - (st.dv - st.dLast));
            tr.albedoSrgb = ld3(inst.meanAlbedo);
            tr.hitPos = rayStartWorld + rayDirectionWorld * (distanceGlobal + lastStepSizeLocal * st.localToGlobalScale);
        }
        return false;
    }
    const float step = fabsf(st.dv);
    st.pos = st.pos + st.dir * step;
    st.hitDistanceLocal += step;
    return true;
}

// the whole of one instance for all lanes of a wave that share it (the instance record is wave-uniform): the form of rounds 1-3, kept for the
// launcher's A/B switch (PLR_TRACE_PER_LANE=0)
template <bool BRICKS = false>
PLR_DI void traceInstance(const SDFInstance& inst, const ImgView& view, vec3 rayStartWorld, vec3 rayDirectionWorld, TraceResult& tr) {
    LocalRay ray;
    if (!enterInstance(inst, rayStartWorld, rayDirectionWorld, &ray)) return;
    MarchState st;
    if (!beginMarch<BRICKS>(inst, view, ray, tr.closestHitDistance, &st)) return;
    for (int i = 0; i < 128; i++)
        if (!marchStep<BRICKS>(st, inst, rayStartWorld, rayDirectionWorld, tr)) break;
}

struct SdfInstanceBuffer { uint32_t instanceCount, pad1, pad2, pad3; SDFInstance instances[1]; };
// one culled instance of the block's tile as staged in LDS: the 96-byte record, its volume's view and the record's index (decision signature)
struct StagedInstance { SDFInstance inst; ImgView view; uint32_t instIndex, pad; };
static_assert(sizeof(StagedInstance) == 128 && sizeof(ImgView) == 24, "StagedInstance is 32 dwords");
struct RayInfo { float nx, ny, nz, depth, cr, cg, cb; };

// SIG: also write the decision signature of every pixel (plr_debug_set_decision_signature; bit layout in oracle/oracle.h)
// PACK: also write the packed texel the spatial filter gathers (fused_gi.h); 0 = no, else the format of packDepth (F_R16F / F_D32)
// BRICKS: march through the volumes' copies in cache-line bricks (device/sdf_bricks.h; PLR_TRACE_BRICKS=1: measured, not the default)
template <bool STRICT_CUTOFF, bool SIG, int PACK, bool PER_LANE = false, bool BRICKS = false>
__global__ __launch_bounds__(256) PLR_TRACE_OCC void sdfDiffuseTraceFastKernel(ImgView outYSH, ImgView outCoCg, ImgView depthTexture, ImgView normalTexture, ImgView skyLut,
                                                                 const LightBuffer* __restrict__ light, const SdfInstanceBuffer* __restrict__ instanceBuffer,
                                                                 const CulledInstancesPerTile* __restrict__ tiles, const float* __restrict__ influenceRangeP,
                                                                 const ShadowCascadeInfo* __restrict__ shadowInfo, ImgView shadowMap, const ImgView* __restrict__ bindless,
                                                                 uint32_t bindlessCount, const GlobalUbo* __restrict__ g, int shadowCascadeIndex, int groupsX, int groupsY, int groupY0, int groupX0,
                                                                 uint32_t tileCapacity, uint32_t instanceCapacity, uint32_t* __restrict__ sig,
                                                                 uint4* __restrict__ packedOut, ImgView packDepth, TwoRanges ranges, ImgView hostNoiseTex,
                                                                 const uint16_t* const* __restrict__ brickedVolumes) {
    __shared__ RayInfo sharedRays[4][64];
    uint32_t raySig = 0u;
    const int wave = (int)(threadIdx.x >> 6), lane = (int)(threadIdx.x & 63u);
    int blockCol, blockRow;
    ranges.blockXY(&blockCol, &blockRow); // a launch over two row ranges, or the edge rows (tile rendering: and columns) first (backend.h TwoRanges)
    // workgroups [groupX0, groupsX) x [groupY0, groupsY): the recorded dispatch (tile rendering restricts the columns too, PassCtx::base[0])
    const int gx = groupX0 + blockCol * 2 + (wave & 1), gy = groupY0 + blockRow * 2 + (wave >> 1);
    const bool active = gx < groupsX && gy < groupsY;
    const int lx = lane & 7, ly = lane >> 3;
    const int px = gx * 8 + lx, py = gy * 8 + ly;
    vec3 L(0.f, 0.f, 1.f);
    // ---- the tile's instance table, staged once per block. The four waves of a block lie in one culling tile; walking the tile's list straight
    // from memory is a chain of three dependent loads per instance and wave (list entry -> instance record -> volume view) in front of the
    // first SDF fetch, ~1 k cycles in which a wave has nothing else to do (59 % of this kernel's wave cycles were parked). Here the block
    // fetches all records, then all views, side by side, and the instance loop reads them from LDS.
    __shared__ StagedInstance staged[kMaxObjectsPerTile];
    const uint32_t tileIndex = min((uint32_t)(gx / 4) + (uint32_t)(gy / 4) * (uint32_t)ceilf((float)g->screenResolution[0] / 32.f), tileCapacity - 1u);
    const CulledInstancesPerTile* tile = tiles + tileIndex;
    const int objectCount = (int)min(tile->objectCount, kMaxObjectsPerTile);
    {
        uint32_t* lds = (uint32_t*)staged;
        for (int e = (int)threadIdx.x; e < objectCount * 24; e += 256) {
            const int i = e / 24, j = e - i * 24;
            const uint32_t instIndex = min(tile->indices[i], instanceCapacity - 1u);
            lds[i * 32 + j] = ((const uint32_t*)&instanceBuffer->instances[instIndex])[j];
            if (j == 0) lds[i * 32 + 30] = instIndex;
        }
        __syncthreads();
        for (int e = (int)threadIdx.x; e < objectCount * 6; e += 256) {
            const int i = e / 6, j = e - i * 6;
            const uint32_t texIndex = min(staged[i].inst.sdfTextureIndex, bindlessCount - 1u);
            uint32_t word = ((const uint32_t*)&bindless[texIndex])[j];
            if (BRICKS) { // a volume with a bricked copy: the view's pointer becomes the copy's, flagged in fmt
                const uint64_t bricks = brickedVolumes ? (uint64_t)(uintptr_t)brickedVolumes[texIndex] : 0ull;
                if (bricks) word = j == 0 ? (uint32_t)bricks : j == 1 ? (uint32_t)(bricks >> 32) : j == 5 ? (word | (uint32_t)kBrickedFormatFlag) : word;
            }
            lds[i * 32 + 24 + j] = word;
        }
        __syncthreads();
    }
    RayInfo mine{0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (active) {
        const float u = (float)px / (float)outYSH.w, v = (float)py / (float)outYSH.h; // IEEE quotient: see the note at the top
        const float depth = sampleNearest2D<F_D32, CLAMP>(depthTexture, vec2(u, v)).x;
        // linear depth as the shader rounds it (product, sum and quotient separately; an IEEE quotient in this file): it is one side of the 0.5 m
        // depth test that decides which neighbours' rays a pixel takes in the 3x3 resolve
        float depthLinear;
        {
#pragma clang fp contract(off)
            const float span = g->nearPlane - g->farPlane, t = (-depth + 1.f) * span, den = g->farPlane + t, nf = g->nearPlane * g->farPlane;
            depthLinear = nf / den;
        }
        const vec3 ray = ld3(g->cameraForward) + (-g->cameraTanFovHalf * (v * 2.f - 1.f)) * ld3(g->cameraUp) +
                         (g->cameraTanFovHalf * g->cameraAspectRatio * (u * 2.f - 1.f)) * ld3(g->cameraRight);
        const vec3 pWorld = ld3(g->cameraPosition) + ray * depthLinear;
        // the launcher resolves the frame's noise texture on the host (PassCtx::hostNoiseView): frame index -> texture index -> view are three
        // dependent round trips in front of the ray direction otherwise
        const ImgView noiseTex = hostNoiseTex;
        // exact UNORM8 decode (c / 255, an IEEE quotient in this file): a noise value of 255 must be exactly 1 - then sinTheta is exactly 0 and L = N, and
        // for a horizontal N the sky LUT's v coordinate sits on its sqrt-steep horizon, where 2e-4 rad of direction is half a LUT row
        const uint32_t nzTexel = ((const uint16_t*)noiseTex.ptr)[fastm::texelIndex((uint32_t)fastm::repeatIndex(px, noiseTex.w), (uint32_t)fastm::repeatIndex(py, noiseTex.h), (uint32_t)noiseTex.w)];
        const vec2 nz(decodeUnorm8Newton(nzTexel & 0xffu), decodeUnorm8Newton(nzTexel >> 8)); // = c / 255 for every code, in three instructions instead of the IEEE division's ten
        const uint32_t nTexel = ((const uint32_t*)normalTexture.ptr)[fastm::texelIndex((uint32_t)clampTo(floorToInt(u * (float)normalTexture.w), normalTexture.w - 1),
                                                                                          (uint32_t)clampTo(floorToInt(v * (float)normalTexture.h), normalTexture.h - 1), (uint32_t)normalTexture.w)];
        const vec3 N = vec3(decodeUnorm8Newton(nTexel & 0xffu), decodeUnorm8Newton((nTexel >> 8) & 0xffu), decodeUnorm8Newton((nTexel >> 16) & 0xffu)) * 2.f - 1.f;
        mine.nx = N.x; mine.ny = N.y; mine.nz = N.z; mine.depth = depthLinear;
        const vec3 rayOrigin = pWorld + N * 0.2f;
        {
            // importanceSampleCosine (sampling.inc:25-45): phi = 2 pi xi.y, v_sin/v_cos take revolutions
            const float cosTheta = __builtin_amdgcn_sqrtf(nz.x), sinTheta = __builtin_amdgcn_sqrtf(1.f - nz.x);
            const float sp = __builtin_amdgcn_sinf(nz.y), cp = __builtin_amdgcn_cosf(nz.y);
            const vec3 up = fabsf(N.z) < 0.999f ? vec3(0.f, 0.f, 1.f) : vec3(1.f, 0.f, 0.f);
            vec3 tangent = cross(up, N);
            tangent = tangent * rsqf(dot(tangent, tangent));
            const vec3 bitangent = cross(N, tangent);
            L = (cp * sinTheta) * tangent + (sp * sinTheta) * bitangent + cosTheta * N;
        }
        TraceResult tr;
        tr.hit = false;
        tr.closestHitDistance = 10000.f;
        tr.hitPos = vec3(0.f);
        tr.albedoSrgb = vec3(0.f);
        if (!PER_LANE) {
            for (int i = 0; i < objectCount; i++) {
                const float before = tr.closestHitDistance;
                traceInstance<BRICKS>(staged[i].inst, staged[i].view, rayOrigin, L, tr);
                if (SIG && tr.closestHitDistance != before) raySig = (staged[i].instIndex + 1u) << 11;
            }
        } else {
            // Round 4 experiment, NOT the default - it lost: 120.9 us against 100.3 us for the lockstep walk at 4K (98 VGPRs = 4 waves
            // per SIMD instead of 80 = 6, and the box tests run twice). The premise was wrong for this scene: a wave marches ~6 steps in total
            // (33 vector loads per wave, profiles/r03_trace_counters.txt), so there are no long per-instance marches to de-serialise; what the
            // kernel loses is occupancy at its tail (3.2 resident waves per SIMD on average of 6 possible, profiles/r04*_sq_counters.csv).
            // Walking the tile's instances in wave lockstep costs sum_i max_lanes(steps of instance i): every instance is marched for as long as
            // its slowest ray needs, by a wave whose other rays mostly miss its box (the bench scene: up to six boxes per tile, one or two per ray).
            // Here every lane first collects the instances whose box its ray ENTERS (the box test needs no memory access), then the wave marches in
            // rounds: in round r every lane marches ITS r-th candidate. A round lasts as long as its slowest lane, and there are as many rounds as the
            // busiest ray has candidates - not as many as the tile has instances. Per ray the instances are still visited in list order with the same
            // arithmetic, so the result is the ray's result of the lockstep walk, bit for bit.
            uint64_t candLo = 0ull, candHi = 0ull; // kMaxObjectsPerTile = 100 <= 128 bits
            for (int i = 0; i < objectCount; i++) {
                LocalRay ray;
                const bool enters = enterInstance(staged[i].inst, rayOrigin, L, &ray);
                if (i < 64) candLo |= enters ? (1ull << i) : 0ull;
                else candHi |= enters ? (1ull << (i - 64)) : 0ull;
            }
            while (__builtin_amdgcn_ballot_w64((candLo | candHi) != 0ull) != 0ull) {
                MarchState st;
                bool go = false;
                int mine = 0; // this lane's candidate of the round (0 for lanes without one: they do not march)
                if ((candLo | candHi) != 0ull) {
                    if (candLo) { mine = __builtin_ctzll(candLo); candLo &= candLo - 1ull; }
                    else { mine = 64 + __builtin_ctzll(candHi); candHi &= candHi - 1ull; }
                    LocalRay ray;
                    enterInstance(staged[mine].inst, rayOrigin, L, &ray); // the same statements as in the collection pass: the same bits
                    go = beginMarch(staged[mine].inst, staged[mine].view, ray, tr.closestHitDistance, &st);
                }
                const float before = tr.closestHitDistance;
                for (int step = 0; step < 128; step++) {
                    if (go) go = marchStep(st, staged[mine].inst, rayOrigin, L, tr);
                    if (__builtin_amdgcn_ballot_w64(go) == 0ull) break;
                }
                if (SIG && tr.closestHitDistance != before) raySig = (staged[mine].instIndex + 1u) << 11;
            }
        }
        vec3 hitColor;
        if (tr.hit) {
            const float* lm = shadowInfo->lightMatrices[shadowCascadeIndex];
            const vec4 p = mulMat4(lm, vec4(tr.hitPos, 1.f));
            const float iw = rcpf(p.w);
            const float shadowMapDepth = sampleNearest2D<F_D16, BORDER_WHITE>(shadowMap, vec2(p.x * iw * 0.5f + 0.5f, p.y * iw * 0.5f + 0.5f)).x;
            const float shadow = gclamp(p.z * iw, 0.f, 1.f) > shadowMapDepth ? 1.f : 0.f;
            const vec3 a = tr.albedoSrgb;
            const vec3 albedo(a.x <= 0.f ? 0.f : __builtin_amdgcn_exp2f(2.2f * __builtin_amdgcn_logf(a.x)), a.y <= 0.f ? 0.f : __builtin_amdgcn_exp2f(2.2f * __builtin_amdgcn_logf(a.y)),
                              a.z <= 0.f ? 0.f : __builtin_amdgcn_exp2f(2.2f * __builtin_amdgcn_logf(a.z)));
            hitColor = albedo * ((shadow * light->sunStrengthExposed) * ld3(light->sunColor));
            const bool hitInRange = (tr.closestHitDistance < *influenceRangeP) || !STRICT_CUTOFF;
            if (!hitInRange || tr.closestHitDistance < 0.0001f) hitColor = vec3(0.f);
            if (SIG) raySig |= 1u | (shadow != 0.f ? 2u : 0u) | ((!hitInRange || tr.closestHitDistance < 0.0001f) ? 4u : 0u);
        } else {
            hitColor = fastm::sampleSkyLut(L, skyLut);
        }
        mine.cr = hitColor.x; mine.cg = hitColor.y; mine.cb = hitColor.z;
    }
    sharedRays[wave][lane] = mine;
    // a wave only reads the slice it wrote: no block barrier (the four waves of a tile finish their rays at different times)
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    if (!active) { ranges.edgeDone((int)blockIdx.y); return; } // (wave-uniform: a wave is one 8 x 8 group)

    float weightTotal = 1.f;
    vec3 color(mine.cr, mine.cg, mine.cb);
    const vec3 myN(mine.nx, mine.ny, mine.nz);
    uint32_t takeMask = 0u;
#pragma unroll
    for (int x = -1; x <= 1; x++)
#pragma unroll
        for (int y = -1; y <= 1; y++) {
            if (x == 0 && y == 0) continue;
            const int neighbour = (x + 1) * 3 + (y + 1) - ((x > 0 || (x == 0 && y > 0)) ? 1 : 0); // 0..7 in loop order
            // selects instead of two divergent branches per neighbour (an excluded neighbour must not be added even with weight 0: it may be NaN)
            const int rx = lx + x, ry = ly + y;
            const bool inGroup = (rx > 0 && ry > 0) && (rx < 8 && ry < 8); // sic: > 0 (:88)
            const RayInfo nb = sharedRays[wave][inGroup ? ry * 8 + rx : lane];
            const float NoN = gclamp(dot(myN, vec3(nb.nx, nb.ny, nb.nz)), 0.f, 1.f);
            const bool take = inGroup && NoN > 0.9f && fabsf(mine.depth - nb.depth) < 0.5f;
            const float weight = (x == 0 ? 1.f : 0.5f) * (y == 0 ? 1.f : 0.5f);
            const vec3 sum = color + weight * vec3(nb.cr, nb.cg, nb.cb);
            color = vec3(take ? sum.x : color.x, take ? sum.y : color.y, take ? sum.z : color.z);
            weightTotal = take ? weightTotal + weight : weightTotal;
            if (SIG) takeMask |= take ? (1u << neighbour) : 0u;
        }
    color = color * rcpf(weightTotal);
    const vec3 YCoCg = linearToYCoCg(color);
    if (px < outYSH.w && py < outYSH.h) {
        // directionToSH_L1(L), |L| = 1: (0.5, -0.86603 L.y, 0.86603 L.z, -0.86603 L.x)
        const vec4 ysh(YCoCg.x * 0.5f, YCoCg.x * (-0.8660254f * L.y), YCoCg.x * (0.8660254f * L.z), YCoCg.x * (-0.8660254f * L.x));
        const size_t idx = (size_t)py * (size_t)outYSH.w + px;
        const uint2 yBits = make_uint2(floatToHalfBits(ysh.x) | (floatToHalfBits(ysh.y) << 16), floatToHalfBits(ysh.z) | (floatToHalfBits(ysh.w) << 16));
        const uint32_t cBits = floatToHalfBits(YCoCg.y) | (floatToHalfBits(YCoCg.z) << 16);
        const bool through = ranges.isEdge((int)blockIdx.y); // rows a neighbouring GPU is waiting for: written through (backend.h TwoRanges)
        storeOut((uint2*)outYSH.ptr + idx, yBits, through);
        storeOut((uint32_t*)outCoCg.ptr + idx, cBits, through);
        if (PACK) packedOut[idx] = packGiTexel(yBits, cBits, Texel<PACK == 0 ? F_R16F : PACK>::load(packDepth.ptr, idx).x, g->nearPlane, g->farPlane);
        if (SIG) sig[idx] = raySig | (takeMask << 3);
    }
    ranges.edgeDone((int)blockIdx.y); // rows-first launch of a band (plr.h first_rows): this wave's rows are written
}

// ---- BRICKS variant (PLR_TRACE_BRICKS=1): the volumes' copies in cache-line bricks (device/sdf_bricks.h), kept in the pass's scratch memory:
// [table: one pointer per global texture slot | copies]
// one thread per texel slot of the bricked copy (slots beyond the volume repeat its last texel / row / slice)
__global__ void sdfBrickKernel(const uint16_t* __restrict__ src, uint16_t* __restrict__ dst, int w, int h, int d) {
    const BrickGrid g = brickGrid(w, h, d);
    const size_t slot = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (slot >= (size_t)g.nbx * g.nby * g.nbz * 64u) return;
    const int in = (int)(slot & 63u), ix = in & 7, iy = (in >> 3) & 3, iz = in >> 5;
    size_t brick = slot >> 6;
    const int bx = (int)(brick % (size_t)g.nbx); brick /= (size_t)g.nbx;
    const int by = (int)(brick % (size_t)g.nby), bz = (int)(brick / (size_t)g.nby);
    const int x = min(7 * bx + ix, w - 1), y = min(4 * by + iy, h - 1), z = min(2 * bz + iz, d - 1);
    dst[slot] = src[((size_t)z * h + y) * w + x];
}


struct BrickCache {
    struct Entry { const void* src = nullptr; int w = 0, h = 0, d = 0; uint64_t version = 0; size_t offset = 0; };
    const void* scratch = nullptr;
    std::vector<Entry> entries;             // per global texture slot
    std::vector<const uint16_t*> table;     // host copy of the device table
};
// -> device table (slot -> bricked copy or null), or null: no copies (a host the backend cannot read the texture table of, out of memory).
// A copy is (re)built when its image is new, resized, moved or has new contents (contentVersionOf: uploads, passes that write it); an image whose address was
// handed out (plr_get_image_device_pointer) is never cached and is marched in its image layout.
static const uint16_t* const* brickedVolumeTable(const PassCtx& c) {
    if (!c.bindlessHost || c.bindlessCount == 0 || !c.scratchSlot) return nullptr;
    const uint32_t n = c.bindlessCount;
    auto align = [](size_t v) { return (v + 255u) & ~(size_t)255u; };
    static thread_local std::map<void**, BrickCache> caches; // one backend per host thread; keyed by the pass's scratch slot
    BrickCache& cache = caches[c.scratchSlot];
    std::vector<BrickCache::Entry> want(n);
    size_t total = align((size_t)n * sizeof(void*));
    for (uint32_t i = 0; i < n; i++) {
        const ImgView& v = c.bindlessHost[i];
        if (!v.ptr || v.fmt != F_R16F || v.w < 2 || v.h < 2 || v.d < 2 || v.w >= 8192) continue;
        const uint64_t version = contentVersionOf(v.ptr);
        if (!version) continue;
        want[i].src = v.ptr; want[i].w = v.w; want[i].h = v.h; want[i].d = v.d; want[i].version = version; want[i].offset = total;
        total += align(brickedTexelCount(v.w, v.h, v.d) * sizeof(uint16_t));
    }
    if (total > ((size_t)8 << 30)) return nullptr;
    uint8_t* scratch = (uint8_t*)c.scratch(total); // grow-only; a re-allocation drops every copy
    if (!scratch) return nullptr;
    const bool fresh = cache.scratch != (const void*)scratch || cache.entries.size() != n;
    if (fresh) { cache.entries.assign(n, BrickCache::Entry{}); cache.table.assign(n, nullptr); cache.scratch = scratch; }
    bool tableChanged = fresh;
    for (uint32_t i = 0; i < n; i++) {
        const BrickCache::Entry &w = want[i], &have = cache.entries[i];
        const uint16_t* copy = w.src ? (const uint16_t*)(scratch + w.offset) : nullptr;
        if (w.src && (have.src != w.src || have.w != w.w || have.h != w.h || have.d != w.d || have.version != w.version || have.offset != w.offset)) {
            const size_t slots = brickedTexelCount(w.w, w.h, w.d);
            sdfBrickKernel<<<(unsigned)divUp((unsigned)slots, 256u), 256, 0, c.stream>>>((const uint16_t*)w.src, (uint16_t*)(scratch + w.offset), w.w, w.h, w.d);
            if (hipGetLastError() != hipSuccess) return nullptr;
        }
        cache.entries[i] = w;
        if (cache.table[i] != copy) { cache.table[i] = copy; tableChanged = true; }
    }
    if (tableChanged) { // rare (start-up, a new volume): the table's host copy must outlive the transfer
        if (hipMemcpyAsync(scratch, cache.table.data(), (size_t)n * sizeof(void*), hipMemcpyHostToDevice, c.stream) != hipSuccess) return nullptr;
        if (hipStreamSynchronize(c.stream) != hipSuccess) return nullptr;
    }
    return (const uint16_t* const*)scratch;
}

static int launchImpl(const PassCtx& c) {
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
    ImgView hostNoise{};
    if (!c.hostNoiseView(&hostNoise)) return kUseGeneralKernel; // a host that does not know the global buffer's contents gets the kernel that chases the pointers
    const bool strict = c.specBool(0, false);
    const int cascade = c.specInt(1, 3);
    if (cascade < 0 || cascade > 3) return c.fail(-1, "sdfDiffuseTrace: shadowCascadeIndex must be 0..3");
    const ImgView& out = c.storage[0];
    if (c.storage[1].w != out.w || c.storage[1].h != out.h) return c.fail(-4, "sdfDiffuseTrace: Y_SH and CoCg targets differ in size");
    // workgroup rows [groupY0, groupsY) of the recorded dispatch; a block is 2x2 workgroups inside one culling tile
    const int groupX0 = (int)c.base[0], groupsX = groupX0 + (int)c.dispatch[0], groupY0 = (int)c.base[1];
    int groupsY = groupY0 + (int)c.dispatch[1];
    if (groupsX <= groupX0 || groupsY <= groupY0) return 0;
    if ((groupY0 & 1) || (groupX0 & 1)) return c.fail(-1, "sdfDiffuseTrace: dispatch base must be a multiple of 2 workgroups");
    // rows in units of workgroups (8 pixel rows), blocks of 2 workgroups; a second row range covered by the same launch (pass fusion of the two
    // edge dispatches of band rendering) is expressed as a jump in the block row
    TwoRanges ranges;
    int blockRowsTotal = (int)divUp((unsigned)(groupsY - groupY0), 2u);
    if (c.extraCountY) {
        if ((c.dispatch[1] & 1u) || ((c.extraBaseY - (uint32_t)groupY0) & 1u) || c.extraBaseY < (uint32_t)groupsY) return kUseGeneralKernel;
        ranges.split = blockRowsTotal;
        ranges.gap = (int)(c.extraBaseY - (uint32_t)groupY0) / 2 - blockRowsTotal;
        blockRowsTotal += (int)divUp(c.extraCountY, 2u);
        groupsY = (int)(c.extraBaseY + c.extraCountY);
    }
    // band rendering, rows-first (plr.h first_rows): the edge rows' blocks come first and raise the backend's edge signal (blocks of 16 pixel rows, 4 waves)
    if (!c.extraCountY) ranges.setEdgeFirst(c, groupY0 * 8, groupsY * 8, 16, 8, divUp((unsigned)(groupsX - groupX0), 2u), groupX0 * 8, groupsX * 8, 16);
    const uint32_t tileCapacity = (uint32_t)(c.sbuf[7].size / sizeof(CulledInstancesPerTile));
    const uint32_t instanceCapacity = (uint32_t)((c.sbuf[6].size - 16u) / sizeof(SDFInstance));
    const dim3 grid(divUp((unsigned)(groupsX - groupX0), 2u), (unsigned)blockRowsTotal);
#define PLR_TRACE_ARGS c.storage[0], c.storage[1], c.sampled[2], c.sampled[3], c.sampled[4], (const LightBuffer*)c.sbuf[5].ptr,                       \
                       (const SdfInstanceBuffer*)c.sbuf[6].ptr, (const CulledInstancesPerTile*)c.sbuf[7].ptr, (const float*)c.ubuf[8].ptr,            \
                       (const ShadowCascadeInfo*)c.sbuf[9].ptr, c.sampled[10], c.bindless, c.bindlessCount, c.global, cascade, groupsX, groupsY, groupY0, groupX0, \
                       tileCapacity, instanceCapacity, sig, pack ? pack->packed : nullptr, pack ? pack->depth : ImgView{}, ranges, hostNoise, brickedVolumes
    uint32_t* sig = c.sigFor((size_t)out.w * (size_t)out.h);
    const uint16_t* const* brickedVolumes = nullptr; // only the BRICKS variant below gets a table
    // the spatial filter that reads this pass's output wants packed texels (PassCtx::consumer, fused_gi.h): written here, for the rows of this launch
    SpatialPackTarget packTarget;
    const SpatialPackTarget* pack = !sig && spatialPackTargetOfConsumer(c, 0, 1, &packTarget) == 0 ? &packTarget : nullptr;
    // (what the launch packs is noted as a rectangle: the columns of the dispatch, not whole rows - ADVICE r03)
    if (pack && (pack->depth.w != out.w || pack->depth.h != out.h || (pack->depth.fmt != F_R16F && pack->depth.fmt != F_D32))) pack = nullptr;
    if (pack) {
        // fused with the spatial filter that reads this pass's output (fused_gi.h); never together with a signature run
        if (pack->depth.w != out.w || pack->depth.h != out.h) return kUseGeneralKernel;
        if (pack->depth.fmt == F_R16F) {
            // A/B switch for the measurement in DESIGN.md: PLR_TRACE_PER_LANE=1 runs the per-lane candidate rounds instead of the lockstep instance walk
            // (this variant only). Measured at 4K: 120.9 us against 100.3 us - not the default.
            static const bool perLane = std::getenv("PLR_TRACE_PER_LANE") && std::atoi(std::getenv("PLR_TRACE_PER_LANE")) == 1;
            // PLR_TRACE_BRICKS=1 (read per launch: tests switch it): the BRICKS variant, for the benchmarked configuration only; same results, measured slower
            // (device/sdf_bricks.h, profiles/r04_not_kept.txt)
            // (2: as 1, and a launch that cannot take the variant is an error - how the test knows the variant ran)
            const char* bricksEnv = std::getenv("PLR_TRACE_BRICKS");
            const int bricksMode = bricksEnv ? std::atoi(bricksEnv) : 0;
            if (bricksMode >= 1 && strict && !perLane) brickedVolumes = brickedVolumeTable(c);
            if (bricksMode >= 2 && !brickedVolumes) return c.fail(-1, "sdfDiffuseTrace: PLR_TRACE_BRICKS=2 but this launch cannot march bricked volumes");
            if (brickedVolumes) sdfDiffuseTraceFastKernel<true, false, F_R16F, false, true><<<grid, 256, 0, c.stream>>>(PLR_TRACE_ARGS);
            else if (strict && perLane) sdfDiffuseTraceFastKernel<true, false, F_R16F, true><<<grid, 256, 0, c.stream>>>(PLR_TRACE_ARGS);
            else if (strict) sdfDiffuseTraceFastKernel<true, false, F_R16F><<<grid, 256, 0, c.stream>>>(PLR_TRACE_ARGS);
            else sdfDiffuseTraceFastKernel<false, false, F_R16F><<<grid, 256, 0, c.stream>>>(PLR_TRACE_ARGS);
        } else if (pack->depth.fmt == F_D32) {
            if (strict) sdfDiffuseTraceFastKernel<true, false, F_D32><<<grid, 256, 0, c.stream>>>(PLR_TRACE_ARGS);
            else sdfDiffuseTraceFastKernel<false, false, F_D32><<<grid, 256, 0, c.stream>>>(PLR_TRACE_ARGS);
        } else return kUseGeneralKernel;
    } else if (sig) {
        if (strict) sdfDiffuseTraceFastKernel<true, true, 0><<<grid, 256, 0, c.stream>>>(PLR_TRACE_ARGS);
        else sdfDiffuseTraceFastKernel<false, true, 0><<<grid, 256, 0, c.stream>>>(PLR_TRACE_ARGS);
    } else if (strict) sdfDiffuseTraceFastKernel<true, false, 0><<<grid, 256, 0, c.stream>>>(PLR_TRACE_ARGS);
    else sdfDiffuseTraceFastKernel<false, false, 0><<<grid, 256, 0, c.stream>>>(PLR_TRACE_ARGS);
#undef PLR_TRACE_ARGS
    PLR_CHECK_LAUNCH(c);
    if (pack) {
        const int px0 = std::min(groupX0 * 8, (int)out.w), px1 = std::min(groupsX * 8, (int)out.w);
        if (c.extraCountY) {
            spatialNotePackedRect(c, px0, std::min(groupY0 * 8, (int)out.h), px1, std::min((groupY0 + (int)c.dispatch[1]) * 8, (int)out.h));
            spatialNotePackedRect(c, px0, std::min((int)c.extraBaseY * 8, (int)out.h), px1, std::min(groupsY * 8, (int)out.h));
        } else spatialNotePackedRect(c, px0, std::min(groupY0 * 8, (int)out.h), px1, std::min(groupsY * 8, (int)out.h));
    }
    return 0;
}

static int launch(const PassCtx& c) { return launchImpl(c); }

} // namespace fasttrace

static int fasttrace_launch(const PassCtx& c) { return fasttrace::launch(c); }
PLR_REGISTER_SHADER_FAST("sdfDiffuseTrace.comp", fasttrace_launch);
// band rendering records the rows its neighbours need as two dispatches (above and below the interior): one launch
static int fasttrace_two_ranges(const PassCtx* const* ctxs, size_t count) { return launchOverTwoRowRanges(ctxs, count, fasttrace_launch); }
PLR_REGISTER_FUSION("sdfDiffuseTrace over two row ranges", fasttrace_two_ranges, "sdfDiffuseTrace.comp", "sdfDiffuseTrace.comp");
} // namespace plr
