// SDF diffuse GI for gfx950, part 2 (denoise): filterIndirectDiffuseSpatial.comp, filterIndirectDiffuseTemporal.comp,
// indirectLightUpscale.comp; host side Techniques/SDFGI.cpp:421-536.
//
// The spatial filter's RNG is seeded identically for every pixel (wang_hash(frameIndexMod4 + filterIndex), :53), so its 32
// (sqrt(rand), cos, sin) triples are a per-launch constant: each block derives them once into LDS instead of 32 sincos per
// pixel. Everything else is a per-pixel gather (depth, normal, Y_SH, CoCg) through the L1/L2-resident half-res images.
#include "../backend.h"
#include "../device/shading_common.h"

namespace plr {

template <int DEPTH_FMT>
PLR_DI vec3 pixelToWorld(vec2 uv, const ImgView& depthTexture, const GlobalUbo* g) {
    const float depth = sampleNearest2D<DEPTH_FMT, CLAMP>(depthTexture, uv).x;
    const float depthLinear = linearizeDepth(depth, g->nearPlane, g->farPlane);
    const vec2 pixelNDC(uv.x * 2.f - 1.f, uv.y * 2.f - 1.f);
    const vec3 camFwd = ld3(g->cameraForward);
    const vec3 cameraToPixel = -calculateViewDirectionFromPixel(pixelNDC, camFwd, ld3(g->cameraUp), ld3(g->cameraRight), g->cameraTanFovHalf, g->cameraAspectRatio);
    return ld3(g->cameraPosition) + cameraToPixel / dot(cameraToPixel, camFwd) * depthLinear;
}

// filterIndirectDiffuseSpatial.comp:30-135
template <int DEPTH_FMT>
__global__ __launch_bounds__(256) void spatialFilterKernel(ImgView outYSH, ImgView outCoCg, ImgView inYSH, ImgView inCoCg, ImgView depthTexture, ImgView normalTexture,
                                                           const GlobalUbo* __restrict__ g, int filterIndex, int coverW, int coverH, int yBase, int validY0, int validY1, int xBase, int validX0,
                                                           int validX1) {
    __shared__ float sqrtRand[32], cosA[32], sinA[32];
    if (threadIdx.x < 64) {
        // lane i replays the xorshift sequence up to its own pair of draws (2*i + 2 steps at most 64: negligible)
        uint32_t rngState = wang_hash(g->frameIndexMod4 + (uint32_t)filterIndex);
        const int i = (int)threadIdx.x;
        if (i < 32) {
            float r0 = 0.f, r1 = 0.f;
            for (int k = 0; k <= i; k++) { r0 = rand01(rngState); r1 = rand01(rngState); }
            sqrtRand[i] = sqrtf(r0);
            const float angle = 2.f * PLR_GLSL_PI * r1;
            float s, c;
            det_sincosf(angle, &s, &c);
            cosA[i] = c; sinA[i] = s;
        }
    }
    __syncthreads();
    const int px = xBase + (int)(blockIdx.x * 64u + (threadIdx.x & 63u)); // columns [xBase, coverW) (tile rendering: PassCtx::colSpan)
    const int py = yBase + (int)(blockIdx.y * 4u + (threadIdx.x >> 6));
    if (px >= coverW || py >= coverH) return;
    const vec2 texelSize(1.f / (float)outYSH.w, 1.f / (float)outYSH.h);
    const vec2 uv(((float)px + 0.5f) * texelSize.x, ((float)py + 0.5f) * texelSize.y);
    const vec3 pCenter = pixelToWorld<DEPTH_FMT>(uv, depthTexture, g);
    const vec3 pRight = pixelToWorld<DEPTH_FMT>(vec2(uv.x + 1.f * texelSize.x, uv.y + 0.f * texelSize.y), depthTexture, g);
    const vec3 pUp = pixelToWorld<DEPTH_FMT>(vec2(uv.x + 0.f * texelSize.x, uv.y + 1.f * texelSize.y), depthTexture, g);
    const vec3 tangent = normalize(pCenter - pRight);
    const vec3 bitangent = normalize(pCenter - pUp);
    const vec3 N = 2.f * sampleNearest2D<F_RGBA8, CLAMP>(normalTexture, uv).xyz() - 1.f;
    vec4 result_Y_SH(0.f);
    vec2 result_CoCg(0.f);
    float weightTotal = 0.f;
    const float radiusWorld = filterIndex == 1 ? 1.f : 1.5f;
    float lengthModifier = 1.f;
    const float* vp = g->viewProjection;
    for (int i = 0; i < 32; i++) {
        const float d = sqrtRand[i] * lengthModifier;
        const vec2 offset(cosA[i] * d, sinA[i] * d);
        const vec3 sampleWorld = pCenter + radiusWorld * (offset.x * tangent + offset.y * bitangent);
        const vec4 sampleProjected = mulMat4(vp, vec4(sampleWorld, 1.f));
        vec2 sampleUV(sampleProjected.x / sampleProjected.w, sampleProjected.y / sampleProjected.w);
        sampleUV = sampleUV * 0.5f + 0.5f;
        sampleUV.x = sampleUV.x < 0.f ? uv.x - offset.x : sampleUV.x;
        sampleUV.y = sampleUV.y < 0.f ? uv.y - offset.y : sampleUV.y;
        sampleUV.x = sampleUV.x > 1.f ? uv.x - offset.x : sampleUV.x;
        sampleUV.y = sampleUV.y > 1.f ? uv.y - offset.y : sampleUV.y;
        const vec3 pixelWorld = pixelToWorld<DEPTH_FMT>(sampleUV, depthTexture, g);
        const float distanceToTangentPlane = fabsf(dot(N, pixelWorld - pCenter));
        float weight = gclamp(0.25f / gmax(distanceToTangentPlane, 0.0001f), 0.f, 1.f);
        weight *= weight;
        // band rendering (PassCtx::validRows): a sample on a row no neighbouring band has sent gets weight 0. (Reflecting such a
        // sample through the pixel's row instead was measured at 8K in four bands: 97.2 % of a band's pixels within one code of the unpartitioned
        // frame after three frames against 98.6 % for the plain drop - the reflected texel is a worse stand-in than a renormalised smaller disc.)
        const int sampleRow = clampi((int)floorf(saneCoord(sampleUV.y * (float)inYSH.h)), inYSH.h);
        const int sampleCol = clampi((int)floorf(saneCoord(sampleUV.x * (float)inYSH.w)), inYSH.w); // (tile rendering: PassCtx::validCols, the same rule for columns)
        if (sampleUV.x < 0.f || sampleUV.y < 0.f || sampleUV.x > 1.f || sampleUV.y > 1.f) {
            weight = 0.f;
            lengthModifier *= 0.98f;
        } else if (sampleRow < validY0 || sampleRow >= validY1 || sampleCol < validX0 || sampleCol >= validX1) {
            // round 4: weight 0, but the disc does not shrink for the samples after it - the off-screen rule exists because the screen ends there; this row
            // exists, another GPU has it. Measured over 16 frames at 8K in four bands (profiles/r04_config5_series.txt): closer to the unpartitioned frame
            // in every band and frame (worst band after three frames 99.51 % within one code against 99.34 %)
            weight = 0.f;
        }
        if (weight > 0.f) {
            const vec4 sample_Y_SH = sampleNearest2D<F_RGBA16F, CLAMP>(inYSH, sampleUV);
            const vec4 cc = sampleNearest2D<F_RG16F, CLAMP>(inCoCg, sampleUV);
            const vec2 sample_CoCg(cc.x, cc.y);
            if (!(anyNan(sample_Y_SH) || anyNan(sample_CoCg))) {
                result_Y_SH += weight * sample_Y_SH;
                result_CoCg += weight * sample_CoCg;
                weightTotal += weight;
            }
        }
    }
    weightTotal = gmax(weightTotal, 0.00001f);
    result_Y_SH = result_Y_SH / weightTotal;
    result_CoCg = result_CoCg / weightTotal;
    const size_t idx = (size_t)py * (size_t)outYSH.w + px;
    Texel<F_RGBA16F>::store(outYSH.ptr, idx, result_Y_SH);
    Texel<F_RG16F>::store(outCoCg.ptr, idx, vec4(result_CoCg.x, result_CoCg.y, 0.f, 0.f));
}

static int launchSpatialFilter(const PassCtx& c) {
    if (int rc = c.needGlobal()) return rc;
    if (int rc = c.needStorage(0, F_RGBA16F, "filterIndirectDiffuseSpatial imageOut_Y_SH")) return rc;
    if (int rc = c.needStorage(1, F_RG16F, "filterIndirectDiffuseSpatial imageOut_CoCg")) return rc;
    if (int rc = c.needSampled(2, F_RGBA16F, "filterIndirectDiffuseSpatial texture_Y_SH")) return rc;
    if (int rc = c.needSampled(3, F_RG16F, "filterIndirectDiffuseSpatial texture_CoCg")) return rc;
    if (int rc = c.needSampled(4, -1, "filterIndirectDiffuseSpatial depthTexture")) return rc;
    if (int rc = c.needSampled(5, F_RGBA8, "filterIndirectDiffuseSpatial normalTexture")) return rc;
    const int filterIndex = c.specInt(0, 0);
    const ImgView& out = c.storage[0];
    const PassCtx::RowSpan rs = c.rowSpan(out.h);
    const PassCtx::ColSpan cs = c.colSpan(out.w);
    const int w = cs.x1, x0 = cs.x0, h = rs.y1, y0 = rs.y0; // columns [x0, w), rows [y0, h)
    if (w <= x0 || h <= y0) return 0;
    const dim3 grid(divUp((unsigned)(w - x0), 64u), divUp((unsigned)(h - y0), 4u));
    // depth is the half-res R16F copy for a half-res trace, the D32 depth buffer otherwise (Techniques/SDFGI.cpp:423)
    int validY0, validY1;
    c.validRowRange(c.sampled[2].h, &validY0, &validY1);
    int validX0, validX1;
    c.validColRange(c.sampled[2].w, &validX0, &validX1);
    if (c.sampled[4].fmt == F_R16F)
        spatialFilterKernel<F_R16F><<<grid, 256, 0, c.stream>>>(out, c.storage[1], c.sampled[2], c.sampled[3], c.sampled[4], c.sampled[5], c.global, filterIndex, w, h, y0, validY0, validY1, x0, validX0, validX1);
    else if (c.sampled[4].fmt == F_D32)
        spatialFilterKernel<F_D32><<<grid, 256, 0, c.stream>>>(out, c.storage[1], c.sampled[2], c.sampled[3], c.sampled[4], c.sampled[5], c.global, filterIndex, w, h, y0, validY0, validY1, x0, validX0, validX1);
    else return c.fail(-4, "filterIndirectDiffuseSpatial: depthTexture must be R16_sFloat or Depth32");
    PLR_CHECK_LAUNCH(c);
    return 0;
}
PLR_REGISTER_SHADER("filterIndirectDiffuseSpatial.comp", launchSpatialFilter);

// ------------------------------------------------------------------------------------------------
// filterIndirectDiffuseTemporal.comp:20-86
__global__ __launch_bounds__(256) void temporalGiFilterKernel(ImgView targetYSH, ImgView targetCoCg, ImgView historyOutYSH, ImgView historyOutCoCg, ImgView inYSH,
                                                              ImgView inCoCg, ImgView historyInYSH, ImgView historyInCoCg, ImgView velocityCurrent,
                                                              ImgView velocityLast, const GlobalUbo* __restrict__ g, int coverW, int coverH, int yBase, int xBase) {
    const int px = xBase + (int)(blockIdx.x * 64u + (threadIdx.x & 63u)); // columns [xBase, coverW) (tile rendering: PassCtx::colSpan)
    const int py = yBase + (int)(blockIdx.y * 4u + (threadIdx.x >> 6));
    if (px >= coverW || py >= coverH) return;
    const vec2 texelSize(1.f / (float)targetYSH.w, 1.f / (float)targetYSH.h);
    const vec2 uv(((float)px + 0.5f) * texelSize.x, ((float)py + 0.5f) * texelSize.y);
    const vec4 current_Y_SH = sampleLinear2D<F_RGBA16F, CLAMP>(inYSH, uv);
    vec4 t = sampleLinear2D<F_RG16F, CLAMP>(inCoCg, uv);
    const vec2 current_CoCg(t.x, t.y);
    t = sampleLinear2D<F_RG16SN, CLAMP>(velocityCurrent, uv);
    const vec2 motion(t.x, t.y);
    const vec2 uvReprojected = uv + motion;
    vec4 history_Y_SH = sampleLinear2D<F_RGBA16F, CLAMP>(historyInYSH, uvReprojected);
    t = sampleLinear2D<F_RG16F, CLAMP>(historyInCoCg, uvReprojected);
    vec2 history_CoCg(t.x, t.y);
    t = sampleLinear2D<F_RG16SN, REPEAT>(velocityLast, uvReprojected); // sic: linearRepeat (:36)
    const vec2 motionLastFrame(t.x, t.y);
    const float motionDifference = sqrtf(fabsf(length(motion) - length(motionLastFrame)));
    const float motionDifferenceFactor = gclamp(motionDifference * 10.f, 0.f, 1.f);
    float alphaMin = 0.6f;
    alphaMin -= 0.3f * fabsf(length(current_Y_SH) - length(history_Y_SH));
    alphaMin = gmax(alphaMin, 0.f);
    float alpha = gmix(0.8f, alphaMin, motionDifferenceFactor);
    const vec2 screenRes((float)g->screenResolution[0], (float)g->screenResolution[1]);
    const float pixelThreshold = 3.f;
    if (fabsf(motion.x) * screenRes.x > pixelThreshold || fabsf(motion.y) * screenRes.y > pixelThreshold ||
        fabsf(motionLastFrame.x) * screenRes.x > pixelThreshold || fabsf(motionLastFrame.y) * screenRes.y > pixelThreshold)
        alpha = alphaMin;
    if (uvReprojected.x < 0.f || uvReprojected.y < 0.f || uvReprojected.x > 1.f || uvReprojected.y > 1.f) alpha = 0.f;
    if (g->cameraCut) alpha = 0.f;
    if (anyNan(current_Y_SH) || anyNan(current_CoCg)) {
        alpha = 1.f;
        if (anyNan(history_Y_SH)) history_Y_SH = vec4(0.f);
        if (anyNan(history_CoCg)) history_CoCg = vec2(0.f);
    }
    const vec4 result_Y_SH = current_Y_SH * (1.f - alpha) + history_Y_SH * alpha;
    const vec2 result_CoCg = current_CoCg * (1.f - alpha) + history_CoCg * alpha;
    const size_t idx = (size_t)py * (size_t)targetYSH.w + px;
    const vec4 cc(result_CoCg.x, result_CoCg.y, 0.f, 0.f);
    Texel<F_RGBA16F>::store(targetYSH.ptr, idx, result_Y_SH);
    Texel<F_RG16F>::store(targetCoCg.ptr, idx, cc);
    if (px < historyOutYSH.w && py < historyOutYSH.h) {
        const size_t hidx = (size_t)py * (size_t)historyOutYSH.w + px;
        Texel<F_RGBA16F>::store(historyOutYSH.ptr, hidx, result_Y_SH);
        Texel<F_RG16F>::store(historyOutCoCg.ptr, hidx, cc);
    }
}

static int launchTemporalGiFilter(const PassCtx& c) {
    if (int rc = c.needGlobal()) return rc;
    const int ysh[4] = {0, 2, 4, 6}, cocg[4] = {1, 3, 5, 7};
    for (int i = 0; i < 2; i++) {
        if (int rc = c.needStorage(ysh[i], F_RGBA16F, "filterIndirectDiffuseTemporal Y_SH output")) return rc;
        if (int rc = c.needStorage(cocg[i], F_RG16F, "filterIndirectDiffuseTemporal CoCg output")) return rc;
    }
    for (int i = 2; i < 4; i++) {
        if (int rc = c.needSampled(ysh[i], F_RGBA16F, "filterIndirectDiffuseTemporal Y_SH input")) return rc;
        if (int rc = c.needSampled(cocg[i], F_RG16F, "filterIndirectDiffuseTemporal CoCg input")) return rc;
    }
    if (int rc = c.needSampled(8, F_RG16SN, "filterIndirectDiffuseTemporal velocityCurrent")) return rc;
    if (int rc = c.needSampled(9, F_RG16SN, "filterIndirectDiffuseTemporal velocityLastFrame")) return rc;
    const ImgView& out = c.storage[0];
    const PassCtx::RowSpan rs = c.rowSpan(out.h);
    const PassCtx::ColSpan cs = c.colSpan(out.w);
    const int w = cs.x1, x0 = cs.x0, h = rs.y1, y0 = rs.y0; // columns [x0, w), rows [y0, h)
    if (w <= x0 || h <= y0) return 0;
    temporalGiFilterKernel<<<dim3(divUp((unsigned)(w - x0), 64u), divUp((unsigned)(h - y0), 4u)), 256, 0, c.stream>>>(
        c.storage[0], c.storage[1], c.storage[2], c.storage[3], c.sampled[4], c.sampled[5], c.sampled[6], c.sampled[7], c.sampled[8], c.sampled[9], c.global, w, h, y0, x0);
    PLR_CHECK_LAUNCH(c);
    return 0;
}
PLR_REGISTER_SHADER("filterIndirectDiffuseTemporal.comp", launchTemporalGiFilter);

// ------------------------------------------------------------------------------------------------
// indirectLightUpscale.comp:17-71
__global__ __launch_bounds__(256) void indirectLightUpscaleKernel(ImgView dstYSH, ImgView dstCoCg, ImgView srcYSH, ImgView srcCoCg, ImgView fullResDepthT,
                                                                  ImgView halfResDepthT, const GlobalUbo* __restrict__ g, int coverW, int coverH, int yBase, int xBase) {
    const int px = xBase + (int)(blockIdx.x * 64u + (threadIdx.x & 63u)); // columns [xBase, coverW) (tile rendering: PassCtx::colSpan)
    const int py = yBase + (int)(blockIdx.y * 4u + (threadIdx.x >> 6));
    if (px >= coverW || py >= coverH) return;
    const vec2 uv(((float)px + 0.5f) / (float)g->screenResolution[0], ((float)py + 0.5f) / (float)g->screenResolution[1]);
    float fullResDepth = sampleNearest2D<F_D32, CLAMP>(fullResDepthT, uv).x;
    fullResDepth = linearizeDepth(fullResDepth, g->nearPlane, g->farPlane);
    const vec2 halfResTexelSize(1.f / (float)halfResDepthT.w, 1.f / (float)halfResDepthT.h);
    // textureGather: (i0,j1), (i1,j1), (i1,j0), (i0,j0)
    int i0, j0; float fa, fb;
    linearCoord(uv.x * (float)halfResDepthT.w, &i0, &fa);
    linearCoord(uv.y * (float)halfResDepthT.h, &j0, &fb);
    float depthSamples[4];
    depthSamples[0] = addressedTexel2D<F_R16F, CLAMP>(halfResDepthT, i0, j0 + 1).x;
    depthSamples[1] = addressedTexel2D<F_R16F, CLAMP>(halfResDepthT, i0 + 1, j0 + 1).x;
    depthSamples[2] = addressedTexel2D<F_R16F, CLAMP>(halfResDepthT, i0 + 1, j0).x;
    depthSamples[3] = addressedTexel2D<F_R16F, CLAMP>(halfResDepthT, i0, j0).x;
    float minDepthDiff = 1000.f;
    vec2 closestDepthTexel(0.f, 0.f);
    bool isEdge = false;
    const float offx[4] = {0.f, 1.f, 1.f, 0.f}, offy[4] = {1.f, 1.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const float dl = linearizeDepth(depthSamples[i], g->nearPlane, g->farPlane);
        const float depthDiff = fabsf(dl - fullResDepth);
        isEdge = isEdge || depthDiff > 0.5f;
        if (depthDiff < minDepthDiff) { minDepthDiff = depthDiff; closestDepthTexel = vec2(offx[i], offy[i]); }
    }
    vec4 result_Y_SH, cc;
    if (isEdge) {
        const vec2 uvClosestTexel = uv + closestDepthTexel * halfResTexelSize;
        result_Y_SH = sampleNearest2D<F_RGBA16F, CLAMP>(srcYSH, uvClosestTexel);
        cc = sampleNearest2D<F_RG16F, CLAMP>(srcCoCg, uvClosestTexel);
    } else {
        result_Y_SH = sampleLinear2D<F_RGBA16F, CLAMP>(srcYSH, uv);
        cc = sampleLinear2D<F_RG16F, CLAMP>(srcCoCg, uv);
    }
    const size_t idx = (size_t)py * (size_t)dstYSH.w + px;
    Texel<F_RGBA16F>::store(dstYSH.ptr, idx, result_Y_SH);
    Texel<F_RG16F>::store(dstCoCg.ptr, idx, vec4(cc.x, cc.y, 0.f, 0.f));
}

static int launchIndirectLightUpscale(const PassCtx& c) {
    if (int rc = c.needGlobal()) return rc;
    if (int rc = c.needStorage(0, F_RGBA16F, "indirectLightUpscale fullResDst_Y_SH")) return rc;
    if (int rc = c.needStorage(1, F_RG16F, "indirectLightUpscale fullResDst_CoCg")) return rc;
    if (int rc = c.needSampled(2, F_RGBA16F, "indirectLightUpscale halfResSrc_Y_SH")) return rc;
    if (int rc = c.needSampled(3, F_RG16F, "indirectLightUpscale halfResSrc_CoCg")) return rc;
    if (int rc = c.needSampled(4, F_D32, "indirectLightUpscale fullResDepth")) return rc;
    if (int rc = c.needSampled(5, F_R16F, "indirectLightUpscale halfResDepth")) return rc;
    const ImgView& out = c.storage[0];
    const PassCtx::RowSpan rs = c.rowSpan(out.h);
    const PassCtx::ColSpan cs = c.colSpan(out.w);
    const int w = cs.x1, x0 = cs.x0, h = rs.y1, y0 = rs.y0; // columns [x0, w), rows [y0, h)
    if (w <= x0 || h <= y0) return 0;
    indirectLightUpscaleKernel<<<dim3(divUp((unsigned)(w - x0), 64u), divUp((unsigned)(h - y0), 4u)), 256, 0, c.stream>>>(c.storage[0], c.storage[1], c.sampled[2], c.sampled[3],
                                                                                                                 c.sampled[4], c.sampled[5], c.global, w, h, y0, x0);
    PLR_CHECK_LAUNCH(c);
    return 0;
}
PLR_REGISTER_SHADER("indirectLightUpscale.comp", launchIndirectLightUpscale);

} // namespace plr
