// ORACLE (test infrastructure, never shipped, never on the product path).
// Restates: resources/shaders/depthHiZPyramid.comp, bloomDownsample.comp, bloomUpsample.comp, applyBloom.comp,
// temporalFilter.comp (+ temporalReprojection.inc, bicubicSampling.inc, luminance.inc) and the CPU-side resolve
// weights of Plain/src/Runtime/Rendering/Techniques/TAA.cpp:181-202.
#include "common.h"

using namespace orc;

// ------------------------------------------------------------------------------------------------ HiZ
// depthHiZPyramid.comp:52-124. `src` is the depth buffer (fromDepthBuffer) or the previous pyramid level.
static vec2 computeMinMax(ivec2 upperLeft, bool fromDepthBuffer, const Image& src, ivec2 srcRes, bool extraRow, bool extraColumn) {
    const ivec2 offsets[4] = {ivec2(0, 0), ivec2(1, 0), ivec2(0, 1), ivec2(1, 1)};
    float depthMin = 1.f;
    float depthMax = 0.f;
    const vec2 texelSize = vec2(1.f) / toVec2(srcRes);
    vec2 upperLeftUV = toVec2(upperLeft) * texelSize;
    upperLeftUV += texelSize * 0.5f;

    auto accumulate = [&](vec2 uv, bool corner) {
        if (fromDepthBuffer) {
            const float depthTexel = texture2D(src, NEAREST, CLAMP, uv).x;
            if (corner) depthMin = gmin(depthMin, depthTexel * (depthTexel == 0.f ? 1.f : 0.f)); // sic (:114)
            else depthMin = gmin(depthMin, depthTexel + (depthTexel == 0.f ? 1.f : 0.f));
            depthMax = gmax(depthMax, depthTexel);
        } else {
            const vec4 previousMinMax = texture2D(src, NEAREST, CLAMP, uv);
            depthMin = gmin(depthMin, previousMinMax.x + (previousMinMax.y == 0.f ? 1.f : 0.f));
            depthMax = gmax(depthMax, previousMinMax.y);
        }
    };
    for (int texel = 0; texel < 4; texel++) accumulate(upperLeftUV + toVec2(offsets[texel]) * texelSize, false);
    if (extraRow) {
        const vec2 o[2] = {vec2(0, 2), vec2(1, 2)};
        for (int t = 0; t < 2; t++) accumulate(upperLeftUV + o[t] * texelSize, false);
    }
    if (extraColumn) {
        const vec2 o[2] = {vec2(2, 0), vec2(2, 1)};
        for (int t = 0; t < 2; t++) accumulate(upperLeftUV + o[t] * texelSize, false);
    }
    if (extraRow && extraColumn) accumulate(upperLeftUV + vec2(2, 2) * texelSize, true);
    return vec2(depthMin, depthMax);
}

// depthHiZPyramid.comp:130-350 evaluated level by level on completed data. The reference's single dispatch reads
// texels of neighbouring workgroups without a cross-workgroup barrier when an intermediate size is odd (SURVEY a6 ii);
// this is the race-free value. mips[l] must have the size max(prev/2, 1).
extern "C" void orc_depth_hiz_pyramid(const orc_image* depthP, const orc_image* mips, int32_t mipCount) {
    const Image* src = &img(depthP);
    bool fromDepthBuffer = true;
    ivec2 srcMipRes(src->w, src->h);
    for (int l = 0; l < mipCount; l++) {
        const Image& dst = img(&mips[l]);
        const ivec2 currentMipRes(std::max(srcMipRes.x / 2, 1), std::max(srcMipRes.y / 2, 1));
        const bool oddY = (srcMipRes.y % 2) == 1, oddX = (srcMipRes.x % 2) == 1;
        const Image* s = src;
        parallelFor(currentMipRes.y, [&](int y0, int y1) {
            for (int y = y0; y < y1; y++)
                for (int x = 0; x < currentMipRes.x; x++) {
                    const vec2 mm = computeMinMax(ivec2(x * 2, y * 2), fromDepthBuffer, *s, srcMipRes, oddY, oddX);
                    imageStore(dst, ivec2(x, y), vec4(mm.x, mm.y, 0, 0));
                }
        });
        fromDepthBuffer = false;
        src = &dst;
        srcMipRes = currentMipRes;
    }
}

// ------------------------------------------------------------------------------------------------ bloom
// bloomDownsample.comp:12-49
extern "C" void orc_bloom_downsample(const orc_image* sourceP, const orc_image* targetP) {
    const Image& source = img(sourceP);
    const Image& target = img(targetP);
    const ivec2 targetResolution(target.w, target.h);
    parallelFor(target.h, [&](int y0, int y1) {
        for (int y = y0; y < y1; y++)
            for (int x = 0; x < target.w; x++) {
                const ivec2 iUV(x, y);
                const vec2 uv = (toVec2(iUV) + 0.5f) / toVec2(targetResolution);
                const vec2 texelSize = 1.f / vec2((float)source.w, (float)source.h);
                vec3 color(0.f);
                auto tap = [&](vec2 o, float wgt) { color += texture2D(source, LINEAR, CLAMP, uv + texelSize * o).xyz() * wgt; };
                color += texture2D(source, LINEAR, CLAMP, uv).xyz() * 0.125f;
                tap(vec2(0.5f, 0.5f), 0.125f); tap(vec2(0.5f, -0.5f), 0.125f); tap(vec2(-0.5f, 0.5f), 0.125f); tap(vec2(-0.5f, -0.5f), 0.125f);
                tap(vec2(1.5f, 0), 0.0625f); tap(vec2(-1.5f, 0), 0.0625f); tap(vec2(0, 1.5f), 0.0625f); tap(vec2(0, -1.5f), 0.0625f);
                tap(vec2(1.5f, 1.5f), 0.03125f); tap(vec2(1.5f, -1.5f), 0.03125f); tap(vec2(-1.5f, 1.5f), 0.03125f); tap(vec2(-1.5f, -1.5f), 0.03125f);
                imageStore(target, iUV, vec4(color, 0));
            }
    });
}

// bloomUpsample.comp:19-57
extern "C" void orc_bloom_upsample(const orc_image* sourceP, const orc_image* prevP, const orc_image* targetP, int32_t isLowestMip, float blurRadius) {
    const Image& source = img(sourceP);
    const Image& target = img(targetP);
    const ivec2 targetResolution(target.w, target.h);
    parallelFor(target.h, [&](int y0, int y1) {
        for (int y = y0; y < y1; y++)
            for (int x = 0; x < target.w; x++) {
                const ivec2 iUV(x, y);
                const vec2 texelSize = 1.f / vec2((float)source.w, (float)source.h);
                const vec2 sampleStepSize = blurRadius * texelSize;
                const vec2 uv = (toVec2(iUV) + 0.5f) / toVec2(targetResolution);
                vec3 color(0.f);
                color += texture2D(source, LINEAR, CLAMP, uv).xyz() * 0.25f;
                auto tap = [&](vec2 o, float wgt) { color += texture2D(source, LINEAR, CLAMP, uv + sampleStepSize * o).xyz() * wgt; };
                tap(vec2(1, 0), 0.125f); tap(vec2(-1, 0), 0.125f); tap(vec2(0, 1), 0.125f); tap(vec2(0, -1), 0.125f);
                tap(vec2(1, 1), 0.0625f); tap(vec2(1, -1), 0.0625f); tap(vec2(-1, 1), 0.0625f); tap(vec2(-1, -1), 0.0625f);
                if (!isLowestMip) {
                    const Image& prev = img(prevP);
                    auto ptap = [&](vec2 o) { color += texture2D(prev, LINEAR, CLAMP, uv + texelSize * o).xyz() * 0.25f; };
                    ptap(vec2(0.5f, 0.5f)); ptap(vec2(0.5f, -0.5f)); ptap(vec2(-0.5f, 0.5f)); ptap(vec2(-0.5f, -0.5f));
                }
                imageStore(target, iUV, vec4(color, 0));
            }
    });
}

// applyBloom.comp:16-30 (in place)
extern "C" void orc_apply_bloom(const orc_image* targetP, const orc_image* bloomP, float bloomStrength) {
    const Image& target = img(targetP);
    const Image& bloomTexture = img(bloomP);
    const ivec2 targetResolution(target.w, target.h);
    parallelFor(target.h, [&](int y0, int y1) {
        for (int y = y0; y < y1; y++)
            for (int x = 0; x < target.w; x++) {
                const ivec2 iUV(x, y);
                const vec2 uv = (toVec2(iUV) + 0.5f) / toVec2(targetResolution);
                const vec3 bloom = texture2D(bloomTexture, LINEAR, CLAMP, uv).xyz();
                const vec3 scene = texelFetch(target, iUV).xyz();
                const vec3 color = mix(scene, bloom, bloomStrength);
                imageStore(target, iUV, vec4(color, 0));
            }
    });
}

// ------------------------------------------------------------------------------------------------ TAA
namespace {

struct N3 { vec3 v[3][3]; };

// temporalReprojection.inc:8-30
vec3 clipAABB(vec3 target, vec3 bbMin, vec3 bbMax) {
    const vec3 epsilon(0.0001f);
    const vec3 center = 0.5f * (bbMax + bbMin);
    const vec3 extend = 0.5f * (bbMax - bbMin) + epsilon;
    const vec3 toTarget = target - center;
    const vec3 toTargetNorm = toTarget / extend;
    const vec3 a = abs(toTargetNorm);
    const float maxComponent = gmax(a.x, gmax(a.y, a.z));
    if (maxComponent < 1.f) return target;
    return center + toTarget / maxComponent;
}
vec3 tonemap(vec3 color) { return color / (1.f + computeLuminance(color)); }
vec3 tonemapReverse(vec3 color) { return color / (1.f - computeLuminance(color)); }

// temporalReprojection.inc:42-52
N3 sampleNeighbourhood(const Image& tex, vec2 uv, vec2 texelSize, bool useTonemapping) {
    N3 n;
    for (int x = -1; x <= 1; x++)
        for (int y = -1; y <= 1; y++) {
            vec3 color = texture2D(tex, LINEAR, CLAMP, uv + texelSize * vec2((float)x, (float)y)).xyz();
            color = useTonemapping ? tonemap(color) : color;
            n.v[x + 1][y + 1] = color;
        }
    return n;
}

// temporalReprojection.inc:67-83
vec2 getClosestFragmentMotion(ivec2 uv, const Image& depthTexture, const Image& velocityTexture) {
    float closestDepth = 0.f;
    ivec2 closestDepthOffset(0, 0);
    for (int x = -1; x <= 1; x++)
        for (int y = -1; y <= 1; y++) {
            const float depth = texelFetch(depthTexture, ivec2(uv.x + x, uv.y + y)).x;
            if (depth > closestDepth) { closestDepth = depth; closestDepthOffset = ivec2(x, y); }
        }
    const vec4 m = texelFetch(velocityTexture, ivec2(uv.x + closestDepthOffset.x, uv.y + closestDepthOffset.y));
    return vec2(m.x, m.y);
}

// temporalFilter.comp:59-69
float computeNeighbourhoodContrast(const N3& n) {
    const float c = computeLuminance(n.v[1][1]);
    return std::fabs(computeLuminance(n.v[0][0]) - c) + std::fabs(computeLuminance(n.v[1][0]) - c) + std::fabs(computeLuminance(n.v[2][0]) - c) +
           std::fabs(computeLuminance(n.v[0][2]) - c) + std::fabs(computeLuminance(n.v[1][2]) - c) + std::fabs(computeLuminance(n.v[2][2]) - c) +
           std::fabs(computeLuminance(n.v[0][1]) - c) + std::fabs(computeLuminance(n.v[2][1]) - c);
}

// bicubicSampling.inc:4-17
float catmullRomWeight1D(float d) {
    const float d1 = std::fabs(d), d2 = d1 * d1, d3 = d2 * d1;
    if (d1 <= 1.f) return (1.f / 6.f) * (9.f * d3 - 15.f * d2 + 6.f);
    if (d1 <= 2.f) return (1.f / 6.f) * (-3.f * d3 + 15.f * d2 - 24.f * d + 12.f);
    return 0.f;
}
vec2 catmullRomWeight2D(vec2 d) { return vec2(catmullRomWeight1D(d.x), catmullRomWeight1D(d.y)); }

vec3 tex(const Image& t, vec2 uv) { return texture2D(t, LINEAR, CLAMP, uv).xyz(); }

// bicubicSampling.inc:28-67
vec3 bicubicSample16Tap(const Image& src, vec2 iUV, vec2 texelSize) {
    const vec2 uvTrunc = floor(iUV - 0.5f) + 0.5f;
    const vec2 d = iUV - uvTrunc;
    const vec2 w[4] = {catmullRomWeight2D(abs(d) + 1.f), catmullRomWeight2D(abs(d)), catmullRomWeight2D(1.f - abs(d)), catmullRomWeight2D(2.f - abs(d))};
    const vec2 uvs[4] = {(uvTrunc - 1.f) * texelSize, uvTrunc * texelSize, (uvTrunc + 1.f) * texelSize, (uvTrunc + 2.f) * texelSize};
    vec3 r(0.f);
    bool first = true;
    for (int j = 0; j < 4; j++)
        for (int i = 0; i < 4; i++) {
            const vec3 t = tex(src, vec2(uvs[i].x, uvs[j].y)) * w[i].x * w[j].y;
            r = first ? t : r + t;
            first = false;
        }
    return r;
}

struct Cubic { vec2 uvTrunc, w0, w3, wB, t; };
Cubic cubicSetup(vec2 iUV) {
    Cubic c;
    c.uvTrunc = floor(iUV - 0.5f) + 0.5f;
    const vec2 f = iUV - c.uvTrunc, f2 = f * f, f3 = f2 * f;
    c.w0 = -0.5f * f3 + f2 - 0.5f * f;
    const vec2 w1 = 1.5f * f3 - 2.5f * f2 + 1.f;
    const vec2 w2 = -1.5f * f3 + 2.f * f2 + 0.5f * f;
    c.w3 = 0.5f * f3 - 0.5f * f2;
    c.wB = w1 + w2;
    c.t = w2 / c.wB;
    return c;
}

// bicubicSampling.inc:72-107
vec3 bicubicSample9Tap(const Image& src, vec2 iUV, vec2 texelSize) {
    const Cubic c = cubicSetup(iUV);
    const vec2 uv0 = (c.uvTrunc - 1.f) * texelSize, uvT = (c.uvTrunc + c.t) * texelSize, uv3 = (c.uvTrunc + 2.f) * texelSize;
    return tex(src, vec2(uv0.x, uv0.y)) * c.w0.x * c.w0.y + tex(src, vec2(uv0.x, uvT.y)) * c.w0.x * c.wB.y + tex(src, vec2(uv0.x, uv3.y)) * c.w0.x * c.w3.y +
           tex(src, vec2(uvT.x, uv0.y)) * c.wB.x * c.w0.y + tex(src, vec2(uvT.x, uvT.y)) * c.wB.x * c.wB.y + tex(src, vec2(uvT.x, uv3.y)) * c.wB.x * c.w3.y +
           tex(src, vec2(uv3.x, uv0.y)) * c.w3.x * c.w0.y + tex(src, vec2(uv3.x, uvT.y)) * c.w3.x * c.wB.y + tex(src, vec2(uv3.x, uv3.y)) * c.w3.x * c.w3.y;
}

// bicubicSampling.inc:112-145
vec3 bicubicSample5Tap(const Image& src, vec2 iUV, vec2 texelSize) {
    const Cubic c = cubicSetup(iUV);
    const vec2 uv0 = (c.uvTrunc - 1.f) * texelSize, uvT = (c.uvTrunc + c.t) * texelSize, uv3 = (c.uvTrunc + 2.f) * texelSize;
    const vec4 result = vec4(tex(src, vec2(uv0.x, uvT.y)), 1.f) * c.w0.x * c.wB.y + vec4(tex(src, vec2(uvT.x, uv0.y)), 1.f) * c.wB.x * c.w0.y +
                        vec4(tex(src, vec2(uvT.x, uvT.y)), 1.f) * c.wB.x * c.wB.y + vec4(tex(src, vec2(uvT.x, uv3.y)), 1.f) * c.wB.x * c.w3.y +
                        vec4(tex(src, vec2(uv3.x, uvT.y)), 1.f) * c.w3.x * c.wB.y;
    return result.xyz() / result.w;
}

// bicubicSampling.inc:150-181
vec3 bicubicSample1Tap(const Image& src, vec2 iUV, vec2 texelSize, const N3& n) {
    const Cubic c = cubicSetup(iUV);
    const vec2 uvT = (c.uvTrunc + c.t) * texelSize;
    const vec3 historySample = tex(src, uvT);
    const vec4 result = vec4(historySample + n.v[0][1] - n.v[1][1], 1.f) * c.w0.x * c.wB.y + vec4(historySample + n.v[1][0] - n.v[1][1], 1.f) * c.wB.x * c.w0.y +
                        vec4(historySample, 1.f) * c.wB.x * c.wB.y + vec4(historySample + n.v[1][2] - n.v[1][1], 1.f) * c.wB.x * c.w3.y +
                        vec4(historySample + n.v[2][1] - n.v[1][1], 1.f) * c.w3.x * c.wB.y;
    return result.xyz() / result.w;
}

} // namespace

// temporalFilter.comp:84-179. resolveWeights9: w0_0, w1_0, w2_0, w0_1, ... (wX_Y, row-major over y then x, :29-39)
extern "C" void orc_temporal_filter(const orc_image* currentP, const orc_image* outputP, const orc_image* historyDstP, const orc_image* historySrcP,
                                    const orc_image* motionP, const orc_image* depthP, const float* rw, const orc_global* g, int32_t useClipping,
                                    int32_t useMotionVectorDilation, int32_t historySampleTech, int32_t useTonemap) {
    const Image &currentFrame = img(currentP), &outputImage = img(outputP), &historyBufferDst = img(historyDstP), &historyBufferSrc = img(historySrcP),
                &motionBuffer = img(motionP), &depthBuffer = img(depthP);
    const vec2 screenRes((float)g->screenResolution[0], (float)g->screenResolution[1]);
    parallelFor(outputImage.h, [&](int y0, int y1) {
        for (int py = y0; py < y1; py++)
            for (int px = 0; px < outputImage.w; px++) {
                const ivec2 iUV(px, py);
                const vec2 texelSize = 1.f / vec2((float)outputImage.w, (float)outputImage.h);
                const vec2 uv = (toVec2(iUV) + 0.5f) * texelSize;
                const N3 neighbourhood = sampleNeighbourhood(currentFrame, uv, texelSize, useTonemap != 0);
                vec3 mn = neighbourhood.v[0][0], mx = neighbourhood.v[0][0];
                for (int i = 0; i < 3; i++)
                    for (int j = 0; j < 3; j++) { mn = min(mn, neighbourhood.v[i][j]); mx = max(mx, neighbourhood.v[i][j]); }

                // resolveColor (:41-57)
                vec3 currentColor(0.f);
                currentColor += neighbourhood.v[0][0] * rw[0]; currentColor += neighbourhood.v[1][0] * rw[1]; currentColor += neighbourhood.v[2][0] * rw[2];
                currentColor += neighbourhood.v[0][1] * rw[3]; currentColor += neighbourhood.v[1][1] * rw[4]; currentColor += neighbourhood.v[2][1] * rw[5];
                currentColor += neighbourhood.v[0][2] * rw[6]; currentColor += neighbourhood.v[1][2] * rw[7]; currentColor += neighbourhood.v[2][2] * rw[8];

                vec2 motion;
                if (useMotionVectorDilation) motion = getClosestFragmentMotion(iUV, depthBuffer, motionBuffer);
                else { const vec4 m = texelFetch(motionBuffer, iUV); motion = vec2(m.x, m.y); }

                vec3 historySample;
                if (historySampleTech == 0) historySample = tex(historyBufferSrc, uv + motion);
                else {
                    const vec2 uvReprojected = toVec2(iUV) + 0.5f + motion * screenRes;
                    if (historySampleTech == 1) historySample = bicubicSample16Tap(historyBufferSrc, uvReprojected, texelSize);
                    else if (historySampleTech == 2) historySample = bicubicSample9Tap(historyBufferSrc, uvReprojected, texelSize);
                    else if (historySampleTech == 3) historySample = bicubicSample5Tap(historyBufferSrc, uvReprojected, texelSize);
                    else if (historySampleTech == 4) historySample = bicubicSample1Tap(historyBufferSrc, uvReprojected, texelSize, neighbourhood);
                    else historySample = vec3(1, 0, 0);
                }
                if (useTonemap) historySample = tonemap(historySample);
                if (useClipping) historySample = clipAABB(historySample, mn, mx);
                else historySample = clamp(historySample, mn, mx);
                if (isnan3(historySample)) historySample = currentColor;

                const float currentContrast = computeNeighbourhoodContrast(neighbourhood);
                const N3 lastNeighbourhood = sampleNeighbourhood(historyBufferSrc, uv + motion, texelSize, useTonemap != 0);
                const float lastContrast = computeNeighbourhoodContrast(lastNeighbourhood);
                float contrastChange = std::fabs(currentContrast - lastContrast);
                contrastChange = gclamp(contrastChange, 0.f, 1.f);
                const float blendMin = 0.03f, blendMax = 0.13f;
                float blendFactor = gmix(blendMax, blendMin, contrastChange);
                if (g->cameraCut) blendFactor = 1.f;
                const vec2 rp = uv + motion;
                if (rp.x < 0.f || rp.y < 0.f || rp.x > 1.f || rp.y > 1.f) {
                    blendFactor = 1.f;
                    // gaussianFilteredNeighbourhood (:71-82)
                    const N3& n = neighbourhood;
                    currentColor = n.v[0][0] * 0.0625f + n.v[0][2] * 0.0625f + n.v[2][0] * 0.0625f + n.v[2][2] * 0.0625f + n.v[1][0] * 0.125f +
                                   n.v[0][1] * 0.125f + n.v[1][2] * 0.125f + n.v[2][1] * 0.125f + n.v[1][1] * 0.25f;
                }
                vec3 color = mix(historySample, currentColor, blendFactor);
                if (useTonemap) color = tonemapReverse(color);
                imageStore(historyBufferDst, iUV, vec4(color, 1.f));
                imageStore(outputImage, iUV, vec4(color, 1.f));
            }
    });
}

// Techniques/TAA.cpp:181-202 (glm::length, glm::exp on float)
extern "C" void orc_taa_resolve_weights(const float* jitter, float* weights) {
    int index = 0;
    float totalWeight = 0.f;
    for (int y = -1; y <= 1; y++)
        for (int x = -1; x <= 1; x++) {
            const float d = length(vec2(jitter[0], jitter[1]) - vec2((float)x, (float)y));
            const float w = det_expf(-2.29f * d * d);
            weights[index] = w;
            totalWeight += w;
            index++;
        }
    for (int i = 0; i < 9; i++) weights[i] /= totalWeight;
}


// ====================================================================================================================
// Optional TAA stage (TAASettings::useSeparateSupersampling, off by default): colorToLuminance.comp + temporalSupersampling.comp.
// Host: Techniques/TAA.cpp:85-137.

// colorToLuminance.comp:14-21 (dst is R8: the store converts to UNORM8)
extern "C" void orc_color_to_luminance(const orc_image* srcP, const orc_image* dstP) {
    const Image& src = img(srcP);
    const Image& dst = img(dstP);
    parallelFor(dst.h, [&](int y0, int y1) {
        for (int y = y0; y < y1; y++)
            for (int x = 0; x < dst.w; x++) {
                const vec3 color = texelFetch(src, ivec2(x, y)).xyz();
                imageStore(dst, ivec2(x, y), vec4(computeLuminance(color), 0.f, 0.f, 0.f));
            }
    });
}

namespace {
// temporalSupersampling.comp:23-29
float minAbsoluteDifference(float s, vec4 v) {
    return gmin(std::fabs(s) - std::fabs(v.x), gmin(std::fabs(s) - std::fabs(v.y), gmin(std::fabs(s) - std::fabs(v.z), std::fabs(s) - std::fabs(v.w))));
}
// temporalSupersampling.comp:39-55
float getClosestNeighbourhoodDepth(const Image& depthBuffer, vec2 uv, const orc_global* g) {
    const vec2 texelSize(1.f / (float)g->screenResolution[0], 1.f / (float)g->screenResolution[1]);
    const int ox[9] = {-1, 0, 1, -1, 0, 1, -1, 0, 1}, oy[9] = {-1, -1, -1, 0, 0, 0, 1, 1, 1};
    float closestDepth = texture2D(depthBuffer, NEAREST, CLAMP, uv + vec2((float)ox[0], (float)oy[0]) * texelSize).x;
    for (int i = 1; i < 9; i++) closestDepth = gmax(texture2D(depthBuffer, NEAREST, CLAMP, uv + vec2((float)ox[i], (float)oy[i]) * texelSize).x, closestDepth);
    return linearizeDepth(closestDepth, g->nearPlane, g->farPlane);
}
} // namespace

// temporalSupersampling.comp:57-110
extern "C" void orc_temporal_supersampling(const orc_image* currentP, const orc_image* lastP, const orc_image* targetP, const orc_image* velocityP,
                                           const orc_image* currentDepthP, const orc_image* lastDepthP, const orc_image* currentLumP, const orc_image* lastLumP,
                                           const orc_global* g, int32_t useTonemap) {
    const Image &currentFrame = img(currentP), &lastFrame = img(lastP), &target = img(targetP), &velocity = img(velocityP), &currentDepth = img(currentDepthP),
                &lastDepth = img(lastDepthP), &currentLum = img(currentLumP), &lastLum = img(lastLumP);
    parallelFor(target.h, [&](int y0, int y1) {
        for (int y = y0; y < y1; y++)
            for (int x = 0; x < target.w; x++) {
                const ivec2 iUV(x, y);
                const vec2 texelSize(1.f / (float)g->screenResolution[0], 1.f / (float)g->screenResolution[1]);
                const vec2 uvCurrent = (vec2((float)x, (float)y) + vec2(0.5f)) * texelSize;
                const vec2 motion = getClosestFragmentMotion(iUV, currentDepth, velocity);
                const vec2 uvLast = uvCurrent + motion;
                vec3 currentSample = texture2D(currentFrame, LINEAR, CLAMP, uvCurrent).xyz();
                vec3 lastSample = texture2D(lastFrame, LINEAR, CLAMP, uvLast).xyz();
                if (useTonemap) { currentSample = tonemap(currentSample); lastSample = tonemap(lastSample); }
                // acceptLastFrameSample (:57-84)
                const vec4 cl = textureGatherR(currentLum, CLAMP, uvCurrent), ll = textureGatherR(lastLum, CLAMP, uvLast);
                const float contrast = minAbsoluteDifference(cl.x, ll) + minAbsoluteDifference(cl.y, ll) + minAbsoluteDifference(cl.z, ll) + minAbsoluteDifference(cl.w, ll);
                const bool contrastTest = contrast < 0.5f;
                const float cd = getClosestNeighbourhoodDepth(currentDepth, uvCurrent, g), ld = getClosestNeighbourhoodDepth(lastDepth, uvLast, g);
                const bool depthTest = std::fabs(cd - ld) < 1.f;
                const bool outOfScreen = uvLast.x < 0.f || uvLast.y < 0.f || uvLast.x > 1.f || uvLast.y > 1.f;
                const bool acceptSample = contrastTest && depthTest && !outOfScreen;
                const float blendFactor = acceptSample ? 0.5f : 0.f;
                vec3 color = currentSample * (1.f - blendFactor) + lastSample * blendFactor; // mix
                if (useTonemap) color = tonemapReverse(color);
                imageStore(target, iUV, vec4(color, 1.f));
            }
    });
}

// ------------------------------------------------------------------------------------------------ known-answer probes (tests/test_kat.py)
// fn 0: bicubic weights at iUV (in: 2 floats, pixel units) -> 16 floats: the 16-tap weights of the four taps along x, then along y
//       (bicubicSampling.inc:28-45), then cubicSetup's (w0, wB = w1 + w2, w3, t = w2 / wB) along x, then along y (bicubicSampling.inc:72-90)
// fn 1: clipAABB(target, bbMin, bbMax) (in: 9 floats) -> 3 floats (temporalReprojection.inc:8-30)
// fn 2: tonemap then tonemapReverse (in: 3 floats) -> 6 floats (temporalReprojection.inc:34-40)
extern "C" void orc_kat_taa(int fn, const float* in, float* out, int64_t n) {
    for (int64_t i = 0; i < n; i++) {
        if (fn == 0) {
            const vec2 iUV(in[2 * i], in[2 * i + 1]);
            const vec2 uvTrunc = floor(iUV - 0.5f) + 0.5f;
            const vec2 d = iUV - uvTrunc;
            const vec2 w[4] = {catmullRomWeight2D(abs(d) + 1.f), catmullRomWeight2D(abs(d)), catmullRomWeight2D(1.f - abs(d)), catmullRomWeight2D(2.f - abs(d))};
            float* o = out + 16 * i;
            for (int k = 0; k < 4; k++) { o[k] = w[k].x; o[4 + k] = w[k].y; }
            const Cubic c = cubicSetup(iUV);
            o[8] = c.w0.x; o[9] = c.wB.x; o[10] = c.w3.x; o[11] = c.t.x;
            o[12] = c.w0.y; o[13] = c.wB.y; o[14] = c.w3.y; o[15] = c.t.y;
        } else if (fn == 1) {
            const float* p = in + 9 * i;
            const vec3 r = clipAABB(vec3(p[0], p[1], p[2]), vec3(p[3], p[4], p[5]), vec3(p[6], p[7], p[8]));
            out[3 * i] = r.x; out[3 * i + 1] = r.y; out[3 * i + 2] = r.z;
        } else if (fn == 2) {
            const vec3 c(in[3 * i], in[3 * i + 1], in[3 * i + 2]);
            const vec3 t = tonemap(c), r = tonemapReverse(t);
            float* o = out + 6 * i;
            o[0] = t.x; o[1] = t.y; o[2] = t.z; o[3] = r.x; o[4] = r.y; o[5] = r.z;
        }
    }
}

// history samplers of temporalFilter.comp:118-131 on one image: tech 0 bilinear, 1 bicubic 16 tap, 2 9 tap, 3 5 tap, 4 1 tap (its
// neighbourhood = nbr9x3, the current frame's 3x3 colours, [x][y] order). iUV in pixel units (n x 2 floats) -> n x 3 floats
extern "C" void orc_kat_history_sample(const orc_image* srcP, int32_t tech, const float* iUVs, const float* nbr9x3, float* out, int64_t n) {
    const Image& src = img(srcP);
    const vec2 texelSize = 1.f / vec2((float)src.w, (float)src.h);
    N3 nb;
    for (int x = 0; x < 3; x++)
        for (int y = 0; y < 3; y++) nb.v[x][y] = nbr9x3 ? vec3(nbr9x3[(x * 3 + y) * 3], nbr9x3[(x * 3 + y) * 3 + 1], nbr9x3[(x * 3 + y) * 3 + 2]) : vec3(0.f);
    for (int64_t i = 0; i < n; i++) {
        const vec2 iUV(iUVs[2 * i], iUVs[2 * i + 1]);
        vec3 r;
        if (tech == 0) r = tex(src, iUV * texelSize);
        else if (tech == 1) r = bicubicSample16Tap(src, iUV, texelSize);
        else if (tech == 2) r = bicubicSample9Tap(src, iUV, texelSize);
        else if (tech == 3) r = bicubicSample5Tap(src, iUV, texelSize);
        else r = bicubicSample1Tap(src, iUV, texelSize, nb);
        out[3 * i] = r.x; out[3 * i + 1] = r.y; out[3 * i + 2] = r.z;
    }
}
