// PLR_MATH_FAST variant of filterIndirectDiffuseSpatial.comp:30-135 (exact variant: kernels/gi_filters.hip).
//
// Same samples, same weights, same accumulation order; what changes is how each quantity is computed:
//  * pixelToWorld(uv) = camPos + cameraToPixel / dot(cameraToPixel, forward) * depthLinear. cameraToPixel is the normalised
//    vector forward - tan*ndc.y*up + tan*aspect*ndc.x*right; with an orthonormal camera basis the normalisation cancels against
//    the dot product, so the position is camPos + (forward - tan*ndc.y*up + tan*aspect*ndc.x*right) * depthLinear: no sqrt, no
//    divide per sample.
//  * the sample's clip position is affine in the disc offset: clip = VP*pCenter + ox*(VP*R*tangent) + oy*(VP*R*bitangent); the three
//    projected vectors are computed once per pixel (only x, y, w rows are needed).
//  * distance to the tangent plane |dot(N, pixelWorld - pCenter)| = |dot(N, camPos - pCenter) + depthLinear * dot(N, ray)| with
//    dot(N, ray) affine in the sample's NDC.
//  * divisions are v_rcp_f32 multiplies, FMA contraction is on.
// Per sample this is ~45 VALU operations + 3 gathers instead of ~180 + 3. Results differ from the exact kernel by float
// rounding only (stated tolerance in tests/test_fast_kernels.py); nearest-texel selection and the off-screen test are the
// discontinuities where a last-bit difference can pick a neighbouring texel.
#include "../backend.h"
#include "../device/shading_common.h"
#include <cstdlib>

namespace plr {

PLR_DI float rcpf(float x) { return __builtin_amdgcn_rcpf(x); }

// Work-groups are dispatched round-robin over the 8 XCDs (block b -> XCD b % 8), each with a private 4 MB L2. The filter gathers
// from a ~100-pixel disc around every pixel, so neighbouring tiles share most of their footprint: with the natural mapping the
// eight L2s each fetch their own copy (measured 364 MB of L2 fills for 62 MB of inputs). The remap below hands every XCD one
// contiguous horizontal band of tiles, so a band's sliding window stays resident in that XCD's L2.
template <int DEPTH_FMT, int TX>
__global__ __launch_bounds__(256) void spatialFilterFastKernel(ImgView outYSH, ImgView outCoCg, ImgView inYSH, ImgView inCoCg, ImgView depthTexture, ImgView normalTexture,
                                                               const GlobalUbo* __restrict__ g, int filterIndex, int coverW, int coverH, int yBase, int tilesX, int numTiles,
                                                               int chunk) {
    __shared__ float sqrtRand[32], cosA[32], sinA[32];
    if (threadIdx.x < 32) {
        uint32_t rngState = wang_hash(g->frameIndexMod4 + (uint32_t)filterIndex);
        const int i = (int)threadIdx.x;
        float r0 = 0.f, r1 = 0.f;
        for (int k = 0; k <= i; k++) { r0 = rand01(rngState); r1 = rand01(rngState); }
        sqrtRand[i] = sqrtf(r0);
        float s, c;
        det_sincosf(2.f * PLR_GLSL_PI * r1, &s, &c);
        cosA[i] = c; sinA[i] = s;
    }
    __syncthreads();
    constexpr int TY = 256 / TX;
    const int tile = (int)(blockIdx.x & 7u) * chunk + (int)(blockIdx.x >> 3);
    if (tile >= numTiles) return;
    const int px = (tile % tilesX) * TX + (int)(threadIdx.x % TX);
    const int py = yBase + (tile / tilesX) * TY + (int)(threadIdx.x / TX);
    if (px >= coverW || py >= coverH) return;

    const float nearP = g->nearPlane, farP = g->farPlane;
    const float nf = nearP * farP, nmf = nearP - farP;
    const vec3 fwd = ld3(g->cameraForward), up = ld3(g->cameraUp), right = ld3(g->cameraRight), camPos = ld3(g->cameraPosition);
    const float tanH = g->cameraTanFovHalf, tanA = g->cameraTanFovHalf * g->cameraAspectRatio;
    const float dW = (float)depthTexture.w, dH = (float)depthTexture.h;
    const int dwi = depthTexture.w, dhi = depthTexture.h;

    auto depthLinearAt = [&](float u, float v) -> float {
        const int x = min(max((int)floorf(u * dW), 0), dwi - 1), y = min(max((int)floorf(v * dH), 0), dhi - 1);
        const float d = Texel<DEPTH_FMT>::load(depthTexture.ptr, (size_t)y * (size_t)dwi + (size_t)x).x;
        return nf * rcpf(farP + (1.f - d) * nmf);
    };
    auto worldAt = [&](float u, float v) -> vec3 {
        const float lin = depthLinearAt(u, v);
        const float nx = u * 2.f - 1.f, ny = v * 2.f - 1.f;
        const vec3 ray = fwd + (-tanH * ny) * up + (tanA * nx) * right;
        return camPos + ray * lin;
    };

    const float tsx = 1.f / (float)outYSH.w, tsy = 1.f / (float)outYSH.h;
    const float u0 = ((float)px + 0.5f) * tsx, v0 = ((float)py + 0.5f) * tsy;
    const vec3 pCenter = worldAt(u0, v0);
    const vec3 pRight = worldAt(u0 + tsx, v0);
    const vec3 pUp = worldAt(u0, v0 + tsy);
    const vec3 dT = pCenter - pRight, dB = pCenter - pUp;
    const float radiusWorld = filterIndex == 1 ? 1.f : 1.5f;
    const vec3 T = dT * (radiusWorld * __builtin_amdgcn_rsqf(dot(dT, dT)));
    const vec3 B = dB * (radiusWorld * __builtin_amdgcn_rsqf(dot(dB, dB)));
    const int nwi = normalTexture.w, nhi = normalTexture.h;
    vec3 N;
    {
        const int x = min(max((int)floorf(u0 * (float)nwi), 0), nwi - 1), y = min(max((int)floorf(v0 * (float)nhi), 0), nhi - 1);
        N = 2.f * Texel<F_RGBA8>::load(normalTexture.ptr, (size_t)y * (size_t)nwi + (size_t)x).xyz() - 1.f;
    }
    // clip.xyw = P0 + ox * PT + oy * PB
    const float* vp = g->viewProjection;
    auto projXYW = [&](vec3 p, float w) -> vec3 {
        return vec3(vp[0] * p.x + vp[4] * p.y + vp[8] * p.z + vp[12] * w, vp[1] * p.x + vp[5] * p.y + vp[9] * p.z + vp[13] * w,
                    vp[3] * p.x + vp[7] * p.y + vp[11] * p.z + vp[15] * w);
    };
    const vec3 P0 = projXYW(pCenter, 1.f), PT = projXYW(T, 0.f), PB = projXYW(B, 0.f);
    // dot(N, pixelWorld - pCenter) = c0 + lin * (nF + ndcY * nU + ndcX * nR)
    const float c0 = dot(N, camPos - pCenter);
    const float nF = dot(N, fwd), nU = -tanH * dot(N, up), nR = tanA * dot(N, right);

    // dist = |c0 + lin * (k0 + sv * k1 + su * k2)|
    const float k0 = nF - nU - nR, k1 = 2.f * nU, k2 = 2.f * nR;
    vec4 result_Y_SH(0.f);
    float resCo = 0.f, resCg = 0.f;
    float weightTotal = 0.f;
    float lengthModifier = 1.f;
    const uint32_t ywi = (uint32_t)inYSH.w;
    const float yW = (float)inYSH.w, yH = (float)inYSH.h, yWm1 = yW - 1.f, yHm1 = yH - 1.f, dWm1 = dW - 1.f, dHm1 = dH - 1.f;
    const bool sameGrid = depthTexture.w == inYSH.w && depthTexture.h == inYSH.h; // half-res trace: depth and GI images share the texel grid
    const uint2* yshTexels = (const uint2*)inYSH.ptr;
    const uint32_t* cocgTexels = (const uint32_t*)inCoCg.ptr;
#pragma unroll 4
    for (int i = 0; i < 32; i++) {
        const float d = sqrtRand[i] * lengthModifier;
        const float ox = cosA[i] * d, oy = sinA[i] * d;
        const vec3 clip = P0 + ox * PT + oy * PB;
        const float invW = rcpf(clip.z) * 0.5f;
        float su = clip.x * invW + 0.5f, sv = clip.y * invW + 0.5f;
        // mirror at the borders (:86-89): a coordinate outside [0,1] is replaced by uv - offset
        su = fabsf(su - 0.5f) > 0.5f ? u0 - ox : su;
        sv = fabsf(sv - 0.5f) > 0.5f ? v0 - oy : sv;
        // nearest texel: trunc == floor for non-negative coordinates, negative ones clamp to 0 either way
        const uint32_t tx = (uint32_t)(int)__builtin_amdgcn_fmed3f(su * yW, 0.f, yWm1), ty = (uint32_t)(int)__builtin_amdgcn_fmed3f(sv * yH, 0.f, yHm1);
        const uint32_t ti = ty * ywi + tx;
        uint32_t di = ti;
        if (!sameGrid) di = (uint32_t)(int)__builtin_amdgcn_fmed3f(sv * dH, 0.f, dHm1) * (uint32_t)dwi + (uint32_t)(int)__builtin_amdgcn_fmed3f(su * dW, 0.f, dWm1);
        // all three gathers are issued together (the Y_SH / CoCg texels are needed unless the sample is off-screen), so a sample
        // costs one memory round trip instead of two dependent ones
        const float dep = Texel<DEPTH_FMT>::load(depthTexture.ptr, di).x;
        const uint2 yt = yshTexels[ti];
        const uint32_t ct = cocgTexels[ti];
        const float lin = nf * rcpf(farP + (1.f - dep) * nmf);
        const float dist = fabsf(c0 + lin * (k0 + sv * k1 + su * k2));
        float weight = gclamp(0.25f * rcpf(gmax(dist, 0.0001f)), 0.f, 1.f);
        weight *= weight;
        if (fabsf(su - 0.5f) > 0.5f || fabsf(sv - 0.5f) > 0.5f) {
            weight = 0.f;
            lengthModifier *= 0.98f;
        }
        const vec4 s(halfBitsToFloat(yt.x & 0xffffu), halfBitsToFloat(yt.x >> 16), halfBitsToFloat(yt.y & 0xffffu), halfBitsToFloat(yt.y >> 16));
        const float co = halfBitsToFloat(ct & 0xffffu), cg = halfBitsToFloat(ct >> 16);
        // NaN guard (:118): finite half inputs cannot overflow this sum, so it is NaN exactly when a component is NaN
        const float nanProbe = ((s.x + s.y) + (s.z + s.w)) + (co + cg);
        if (weight > 0.f && nanProbe == nanProbe) {
            result_Y_SH = result_Y_SH + weight * s;
            resCo += weight * co;
            resCg += weight * cg;
            weightTotal += weight;
        }
    }
    const float inv = rcpf(gmax(weightTotal, 0.00001f));
    const size_t idx = (size_t)py * (size_t)outYSH.w + px;
    Texel<F_RGBA16F>::store(outYSH.ptr, idx, result_Y_SH * inv);
    Texel<F_RG16F>::store(outCoCg.ptr, idx, vec4(resCo * inv, resCg * inv, 0.f, 0.f));
}

static int launchSpatialFilterFast(const PassCtx& c) {
    if (int rc = c.needGlobal()) return rc;
    if (int rc = c.needStorage(0, F_RGBA16F, "filterIndirectDiffuseSpatial imageOut_Y_SH")) return rc;
    if (int rc = c.needStorage(1, F_RG16F, "filterIndirectDiffuseSpatial imageOut_CoCg")) return rc;
    if (int rc = c.needSampled(2, F_RGBA16F, "filterIndirectDiffuseSpatial texture_Y_SH")) return rc;
    if (int rc = c.needSampled(3, F_RG16F, "filterIndirectDiffuseSpatial texture_CoCg")) return rc;
    if (int rc = c.needSampled(4, -1, "filterIndirectDiffuseSpatial depthTexture")) return rc;
    if (int rc = c.needSampled(5, F_RGBA8, "filterIndirectDiffuseSpatial normalTexture")) return rc;
    if (c.sampled[3].w != c.sampled[2].w || c.sampled[3].h != c.sampled[2].h) return c.fail(-4, "filterIndirectDiffuseSpatial: Y_SH and CoCg inputs differ in size");
    const int filterIndex = c.specInt(0, 0);
    const ImgView& out = c.storage[0];
    const PassCtx::RowSpan rs = c.rowSpan(out.h);
    const int w = std::min((int)(c.dispatch[0] * 8u), out.w), h = rs.y1, y0 = rs.y0; // columns [0, w), rows [y0, h)
    if (w <= 0 || h <= y0) return 0;
    const int tileX = 64; // 64x4 tiles measured best (64: 197 us, 32: 199 us, 16: 205 us per pass at 4K before the instruction diet)
    const int TXv = tileX == 64 ? 64 : (tileX == 16 ? 16 : 32), TYv = 256 / TXv;
    const int tilesX = (int)divUp((unsigned)w, (unsigned)TXv), tilesY = (int)divUp((unsigned)(h - y0), (unsigned)TYv);
    const int numTiles = tilesX * tilesY, chunk = (numTiles + 7) / 8;
    const dim3 grid((unsigned)chunk * 8u);
#define PLR_SPATIAL_LAUNCH(FMT, TXC) spatialFilterFastKernel<FMT, TXC><<<grid, 256, 0, c.stream>>>(out, c.storage[1], c.sampled[2], c.sampled[3], c.sampled[4], c.sampled[5], \
                                                                                              c.global, filterIndex, w, h, y0, tilesX, numTiles, chunk)
    if (c.sampled[4].fmt == F_R16F) {
        if (TXv == 64) PLR_SPATIAL_LAUNCH(F_R16F, 64); else if (TXv == 16) PLR_SPATIAL_LAUNCH(F_R16F, 16); else PLR_SPATIAL_LAUNCH(F_R16F, 32);
    } else if (c.sampled[4].fmt == F_D32) {
        if (TXv == 64) PLR_SPATIAL_LAUNCH(F_D32, 64); else if (TXv == 16) PLR_SPATIAL_LAUNCH(F_D32, 16); else PLR_SPATIAL_LAUNCH(F_D32, 32);
    } else return c.fail(-4, "filterIndirectDiffuseSpatial: depthTexture must be R16_sFloat or Depth32");
#undef PLR_SPATIAL_LAUNCH
    PLR_CHECK_LAUNCH(c);
    return 0;
}
PLR_REGISTER_SHADER_FAST("filterIndirectDiffuseSpatial.comp", launchSpatialFilterFast);

} // namespace plr
