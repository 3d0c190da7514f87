// TAA resolve for gfx950: temporalFilter.comp (+ temporalReprojection.inc, bicubicSampling.inc, luminance.inc),
// host Techniques/TAA.cpp:139-166. One lane per pixel, 64-pixel row segments per wave.
//
// The 3x3 neighbourhood of the current frame is sampled at exact texel centres, where the 8-bit sub-texel bilinear
// weights are (1,0,0,0): those nine taps are plain clamped fetches. History taps sit at the reprojected (arbitrary)
// position and use the full bilinear footprint. All four specialisation constants are template parameters
// (clip x dilate x 5 history samplers x tonemap = 40 variants, instantiated below).
#include "../backend.h"
#include "../device/shading_common.h"

namespace plr {

struct N3 { vec3 v[3][3]; }; // [x+1][y+1] as in sampleNeighbourhood

PLR_DI vec3 taaTonemap(vec3 c) { return c / (1.f + computeLuminance(c)); }
PLR_DI vec3 taaTonemapReverse(vec3 c) { return c / (1.f - computeLuminance(c)); }

// linear-clamp tap of an R11G11B10 image (uv in [0,1] units)
PLR_DI vec3 historyTap(const ImgView& im, float u, float v) {
    int i0, j0; float a, b;
    linearCoord(u * (float)im.w, &i0, &a);
    linearCoord(v * (float)im.h, &j0, &b);
    const uint32_t* base = (const uint32_t*)im.ptr;
    const int x0 = clampi(i0, im.w), x1 = clampi(i0 + 1, im.w);
    const size_t r0 = (size_t)clampi(j0, im.h) * (size_t)im.w, r1 = (size_t)clampi(j0 + 1, im.h) * (size_t)im.w;
    const float w00 = (1.f - a) * (1.f - b), w10 = a * (1.f - b), w01 = (1.f - a) * b, w11 = a * b;
    vec3 r = unpackR11G11B10(base[r0 + x0]) * w00;
    if (w10 != 0.f) r = r + unpackR11G11B10(base[r0 + x1]) * w10;
    if (w01 != 0.f) r = r + unpackR11G11B10(base[r1 + x0]) * w01;
    if (w11 != 0.f) r = r + unpackR11G11B10(base[r1 + x1]) * w11;
    return r;
}

// temporalReprojection.inc:8-30
PLR_DI vec3 clipAABB(vec3 target, vec3 bbMin, vec3 bbMax) {
    const vec3 center = 0.5f * (bbMax + bbMin);
    const vec3 extend = 0.5f * (bbMax - bbMin) + vec3(0.0001f);
    const vec3 toTarget = target - center;
    const vec3 n = vabs(toTarget / extend);
    const float maxComponent = gmax(n.x, gmax(n.y, n.z));
    if (maxComponent < 1.f) return target;
    return center + toTarget / maxComponent;
}

// temporalFilter.comp:59-69
PLR_DI float neighbourhoodContrast(const N3& n) {
    const float c = computeLuminance(n.v[1][1]);
    return fabsf(computeLuminance(n.v[0][0]) - c) + fabsf(computeLuminance(n.v[1][0]) - c) + fabsf(computeLuminance(n.v[2][0]) - c) +
           fabsf(computeLuminance(n.v[0][2]) - c) + fabsf(computeLuminance(n.v[1][2]) - c) + fabsf(computeLuminance(n.v[2][2]) - c) +
           fabsf(computeLuminance(n.v[0][1]) - c) + fabsf(computeLuminance(n.v[2][1]) - c);
}

// bicubicSampling.inc:4-17
PLR_DI float catmullRomWeight1D(float d) {
    const float d1 = fabsf(d), d2 = d1 * d1, d3 = d2 * d1;
    if (d1 <= 1.f) return (1.f / 6.f) * (9.f * d3 - 15.f * d2 + 6.f);
    if (d1 <= 2.f) return (1.f / 6.f) * (-3.f * d3 + 15.f * d2 - 24.f * d + 12.f);
    return 0.f;
}

struct Cubic { vec2 uvTrunc, w0, w3, wB, t; };
// bicubicSampling.inc:74-85 (shared by the 9/5/1-tap variants)
PLR_DI Cubic cubicSetup(vec2 iUV) {
    Cubic c;
    c.uvTrunc = vec2(floorf(iUV.x - 0.5f) + 0.5f, floorf(iUV.y - 0.5f) + 0.5f);
    const vec2 f = iUV - c.uvTrunc, f2 = f * f, f3 = f2 * f;
    c.w0 = -0.5f * f3 + f2 - 0.5f * f;
    const vec2 w1 = 1.5f * f3 - 2.5f * f2 + 1.f;
    const vec2 w2 = -1.5f * f3 + 2.f * f2 + 0.5f * f;
    c.w3 = 0.5f * f3 - 0.5f * f2;
    c.wB = w1 + w2;
    c.t = w2 / c.wB;
    return c;
}

template <int TECH>
PLR_DI vec3 sampleHistory(const ImgView& hist, vec2 uv, vec2 motion, int px, int py, vec2 texelSize, vec2 screenRes, const N3& n) {
    if (TECH == 0) return historyTap(hist, uv.x + motion.x, uv.y + motion.y);
    const vec2 iUV = vec2((float)px, (float)py) + 0.5f + motion * screenRes;
    if (TECH == 1) { // bicubicSample16Tap, bicubicSampling.inc:28-67
        const vec2 uvTrunc(floorf(iUV.x - 0.5f) + 0.5f, floorf(iUV.y - 0.5f) + 0.5f);
        const vec2 d = iUV - uvTrunc;
        const float ax = fabsf(d.x), ay = fabsf(d.y);
        const float wx[4] = {catmullRomWeight1D(ax + 1.f), catmullRomWeight1D(ax), catmullRomWeight1D(1.f - ax), catmullRomWeight1D(2.f - ax)};
        const float wy[4] = {catmullRomWeight1D(ay + 1.f), catmullRomWeight1D(ay), catmullRomWeight1D(1.f - ay), catmullRomWeight1D(2.f - ay)};
        const float us[4] = {(uvTrunc.x - 1.f) * texelSize.x, uvTrunc.x * texelSize.x, (uvTrunc.x + 1.f) * texelSize.x, (uvTrunc.x + 2.f) * texelSize.x};
        const float vs[4] = {(uvTrunc.y - 1.f) * texelSize.y, uvTrunc.y * texelSize.y, (uvTrunc.y + 1.f) * texelSize.y, (uvTrunc.y + 2.f) * texelSize.y};
        vec3 r(0.f);
        for (int j = 0; j < 4; j++)
            for (int i = 0; i < 4; i++) {
                const vec3 t = historyTap(hist, us[i], vs[j]) * wx[i] * wy[j];
                r = (i == 0 && j == 0) ? t : r + t;
            }
        return r;
    }
    const Cubic c = cubicSetup(iUV);
    const vec2 uv0 = (c.uvTrunc - 1.f) * texelSize, uvT = (c.uvTrunc + c.t) * texelSize, uv3 = (c.uvTrunc + 2.f) * texelSize;
    if (TECH == 2) { // bicubicSample9Tap, :72-107
        return historyTap(hist, uv0.x, uv0.y) * c.w0.x * c.w0.y + historyTap(hist, uv0.x, uvT.y) * c.w0.x * c.wB.y + historyTap(hist, uv0.x, uv3.y) * c.w0.x * c.w3.y +
               historyTap(hist, uvT.x, uv0.y) * c.wB.x * c.w0.y + historyTap(hist, uvT.x, uvT.y) * c.wB.x * c.wB.y + historyTap(hist, uvT.x, uv3.y) * c.wB.x * c.w3.y +
               historyTap(hist, uv3.x, uv0.y) * c.w3.x * c.w0.y + historyTap(hist, uv3.x, uvT.y) * c.w3.x * c.wB.y + historyTap(hist, uv3.x, uv3.y) * c.w3.x * c.w3.y;
    }
    if (TECH == 3) { // bicubicSample5Tap, :112-145
        const vec4 r = vec4(historyTap(hist, uv0.x, uvT.y), 1.f) * c.w0.x * c.wB.y + vec4(historyTap(hist, uvT.x, uv0.y), 1.f) * c.wB.x * c.w0.y +
                       vec4(historyTap(hist, uvT.x, uvT.y), 1.f) * c.wB.x * c.wB.y + vec4(historyTap(hist, uvT.x, uv3.y), 1.f) * c.wB.x * c.w3.y +
                       vec4(historyTap(hist, uv3.x, uvT.y), 1.f) * c.w3.x * c.wB.y;
        return r.xyz() / r.w;
    }
    // bicubicSample1Tap, :150-181: the history tap is not tonemapped yet when the (tonemapped) neighbourhood deltas are added
    const vec3 h = historyTap(hist, uvT.x, uvT.y);
    const vec4 r = vec4(h + n.v[0][1] - n.v[1][1], 1.f) * c.w0.x * c.wB.y + vec4(h + n.v[1][0] - n.v[1][1], 1.f) * c.wB.x * c.w0.y +
                   vec4(h, 1.f) * c.wB.x * c.wB.y + vec4(h + n.v[1][2] - n.v[1][1], 1.f) * c.wB.x * c.w3.y +
                   vec4(h + n.v[2][1] - n.v[1][1], 1.f) * c.w3.x * c.wB.y;
    return r.xyz() / r.w;
}

struct ResolveWeights { float w[9]; };

template <bool CLIP, bool DILATE, int TECH, bool TONEMAP>
__global__ __launch_bounds__(256) void temporalFilterKernel(ImgView current, ImgView output, ImgView historyDst, ImgView historySrc, ImgView motionBuffer,
                                                            ImgView depthBuffer, const ResolveWeights* __restrict__ rwp, const GlobalUbo* __restrict__ g,
                                                            int coverW, int coverH, int yBase, int xBase) {
    const int px = xBase + (int)(blockIdx.x * 64u + (threadIdx.x & 63u)); // columns [xBase, coverW) (tile rendering: PassCtx::colSpan)
    const int py = yBase + (int)(blockIdx.y * 4u + (threadIdx.x >> 6));
    if (px >= coverW || py >= coverH) return;
    const vec2 texelSize(1.f / (float)output.w, 1.f / (float)output.h);
    const vec2 uv(((float)px + 0.5f) * texelSize.x, ((float)py + 0.5f) * texelSize.y);

    // sampleNeighbourhood(currentFrame): centre taps -> clamped fetches
    N3 n;
    const uint32_t* cur = (const uint32_t*)current.ptr;
    for (int x = -1; x <= 1; x++)
        for (int y = -1; y <= 1; y++) {
            const vec3 c = unpackR11G11B10(cur[(size_t)clampi(py + y, current.h) * (size_t)current.w + clampi(px + x, current.w)]);
            n.v[x + 1][y + 1] = TONEMAP ? taaTonemap(c) : c;
        }
    vec3 mn = n.v[0][0], mx = n.v[0][0];
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) { mn = vmin(mn, n.v[i][j]); mx = vmax(mx, n.v[i][j]); }

    const ResolveWeights rw = *rwp;
    vec3 currentColor(0.f);
    currentColor += n.v[0][0] * rw.w[0]; currentColor += n.v[1][0] * rw.w[1]; currentColor += n.v[2][0] * rw.w[2];
    currentColor += n.v[0][1] * rw.w[3]; currentColor += n.v[1][1] * rw.w[4]; currentColor += n.v[2][1] * rw.w[5];
    currentColor += n.v[0][2] * rw.w[6]; currentColor += n.v[1][2] * rw.w[7]; currentColor += n.v[2][2] * rw.w[8];

    // motion (temporalReprojection.inc:67-83): closest = largest reverse-Z depth in the 3x3, out-of-image fetches read 0
    vec2 motion;
    {
        int ox = 0, oy = 0;
        if (DILATE) {
            float closest = 0.f;
            for (int x = -1; x <= 1; x++)
                for (int y = -1; y <= 1; y++) {
                    const float d = texelFetch2D<F_D32>(depthBuffer, px + x, py + y).x;
                    if (d > closest) { closest = d; ox = x; oy = y; }
                }
        }
        const vec4 m = texelFetch2D<F_RG16SN>(motionBuffer, px + ox, py + oy);
        motion = vec2(m.x, m.y);
    }

    const vec2 screenRes((float)g->screenResolution[0], (float)g->screenResolution[1]);
    vec3 historySample = sampleHistory<TECH>(historySrc, uv, motion, px, py, texelSize, screenRes, n);
    if (TONEMAP) historySample = taaTonemap(historySample);
    if (CLIP) historySample = clipAABB(historySample, mn, mx);
    else historySample = vclamp(historySample, mn, mx);
    if (anyNan(historySample)) historySample = currentColor;

    const float currentContrast = neighbourhoodContrast(n);
    N3 ln;
    const vec2 rp = uv + motion;
    for (int x = -1; x <= 1; x++)
        for (int y = -1; y <= 1; y++) {
            const vec3 c = historyTap(historySrc, rp.x + texelSize.x * (float)x, rp.y + texelSize.y * (float)y);
            ln.v[x + 1][y + 1] = TONEMAP ? taaTonemap(c) : c;
        }
    const float lastContrast = neighbourhoodContrast(ln);
    const float contrastChange = gclamp(fabsf(currentContrast - lastContrast), 0.f, 1.f);
    float blendFactor = gmix(0.13f, 0.03f, contrastChange);
    if (g->cameraCut) blendFactor = 1.f;
    if (rp.x < 0.f || rp.y < 0.f || rp.x > 1.f || rp.y > 1.f) {
        blendFactor = 1.f;
        currentColor = n.v[0][0] * 0.0625f + n.v[0][2] * 0.0625f + n.v[2][0] * 0.0625f + n.v[2][2] * 0.0625f + n.v[1][0] * 0.125f + n.v[0][1] * 0.125f +
                       n.v[1][2] * 0.125f + n.v[2][1] * 0.125f + n.v[1][1] * 0.25f;
    }
    vec3 color = vmix(historySample, currentColor, blendFactor);
    if (TONEMAP) color = taaTonemapReverse(color);
    const uint32_t packed = packR11G11B10(color);
    if (px < historyDst.w && py < historyDst.h) ((uint32_t*)historyDst.ptr)[(size_t)py * (size_t)historyDst.w + px] = packed;
    ((uint32_t*)output.ptr)[(size_t)py * (size_t)output.w + px] = packed;
}

typedef void (*TaaKernel)(ImgView, ImgView, ImgView, ImgView, ImgView, ImgView, const ResolveWeights*, const GlobalUbo*, int, int, int, int);

template <bool CLIP, bool DILATE, bool TONEMAP> static TaaKernel pickTech(int tech) {
    switch (tech) {
        case 0: return temporalFilterKernel<CLIP, DILATE, 0, TONEMAP>;
        case 1: return temporalFilterKernel<CLIP, DILATE, 1, TONEMAP>;
        case 2: return temporalFilterKernel<CLIP, DILATE, 2, TONEMAP>;
        case 3: return temporalFilterKernel<CLIP, DILATE, 3, TONEMAP>;
        case 4: return temporalFilterKernel<CLIP, DILATE, 4, TONEMAP>;
        default: return nullptr;
    }
}

static int launchTemporalFilter(const PassCtx& c) {
    if (int rc = c.needGlobal()) return rc;
    if (int rc = c.needSampled(0, F_R11G11B10, "temporalFilter currentFrame")) return rc;
    if (int rc = c.needStorage(1, F_R11G11B10, "temporalFilter outputImage")) return rc;
    if (int rc = c.needStorage(2, F_R11G11B10, "temporalFilter historyBufferDst")) return rc;
    if (int rc = c.needSampled(3, F_R11G11B10, "temporalFilter historyBufferSrc")) return rc;
    if (int rc = c.needSampled(4, F_RG16SN, "temporalFilter motionBuffer")) return rc;
    if (int rc = c.needSampled(5, F_D32, "temporalFilter depthBuffer")) return rc;
    if (int rc = c.needUbuf(6, 36, "temporalFilter resolveWeightBuffer")) return rc;
    const bool clip = c.specBool(0, false), dilate = c.specBool(1, false), tonemap = c.specBool(3, false);
    const int tech = c.specInt(2, 0);
    TaaKernel k = nullptr;
    if (clip) {
        if (dilate) k = tonemap ? pickTech<true, true, true>(tech) : pickTech<true, true, false>(tech);
        else k = tonemap ? pickTech<true, false, true>(tech) : pickTech<true, false, false>(tech);
    } else {
        if (dilate) k = tonemap ? pickTech<false, true, true>(tech) : pickTech<false, true, false>(tech);
        else k = tonemap ? pickTech<false, false, true>(tech) : pickTech<false, false, false>(tech);
    }
    if (!k) return c.fail(-6, "temporalFilter: historySampleTech must be 0..4");
    const ImgView& out = c.storage[1];
    const PassCtx::ColSpan cs = c.colSpan(std::min(out.w, c.sampled[0].w));
    const int w = cs.x1, x0 = cs.x0; // columns [x0, w)
    const PassCtx::RowSpan rs = c.rowSpan(std::min(out.h, c.sampled[0].h));
    const int h = rs.y1, y0 = rs.y0; // rows [y0, h)
    if (w <= x0 || h <= y0) return 0;
    k<<<dim3(divUp((unsigned)(w - x0), 64u), divUp((unsigned)(h - y0), 4u)), 256, 0, c.stream>>>(c.sampled[0], out, c.storage[2], c.sampled[3], c.sampled[4], c.sampled[5],
                                                                                          (const ResolveWeights*)c.ubuf[6].ptr, c.global, w, h, y0, x0);
    PLR_CHECK_LAUNCH(c);
    return 0;
}
PLR_REGISTER_SHADER("temporalFilter.comp", launchTemporalFilter);


// ====================================================================================================================
// Optional TAA stage (TAASettings::useSeparateSupersampling): colorToLuminance.comp + temporalSupersampling.comp (Techniques/TAA.cpp:85-137)

// colorToLuminance.comp:14-21
__global__ __launch_bounds__(256) void colorToLuminanceKernel(ImgView src, ImgView dst, int coverW, int coverH, int yBase) {
    const int px = (int)(blockIdx.x * 64u + (threadIdx.x & 63u));
    const int py = yBase + (int)(blockIdx.y * 4u + (threadIdx.x >> 6));
    if (px >= coverW || py >= coverH) return;
    const vec3 color = texelFetch2D<F_R11G11B10>(src, px, py).xyz();
    Texel<F_R8>::store(dst.ptr, (size_t)py * (size_t)dst.w + px, vec4(computeLuminance(color), 0.f, 0.f, 0.f));
}
static int launchColorToLuminance(const PassCtx& c) {
    if (int rc = c.needSampled(0, F_R11G11B10, "colorToLuminance srcTexture")) return rc;
    if (int rc = c.needStorage(1, F_R8, "colorToLuminance dstImage")) return rc;
    const ImgView& dst = c.storage[1];
    const PassCtx::RowSpan rs = c.rowSpan(dst.h);
    const int w = std::min((int)(c.dispatch[0] * 8u), dst.w), h = rs.y1, y0 = rs.y0;
    if (w <= 0 || h <= y0) return 0;
    colorToLuminanceKernel<<<dim3(divUp((unsigned)w, 64u), divUp((unsigned)(h - y0), 4u)), 256, 0, c.stream>>>(c.sampled[0], dst, w, h, y0);
    PLR_CHECK_LAUNCH(c);
    return 0;
}
PLR_REGISTER_SHADER("colorToLuminance.comp", launchColorToLuminance);

// temporalSupersampling.comp:23-29
PLR_DI float minAbsoluteDifference(float s, vec4 v) {
    return gmin(fabsf(s) - fabsf(v.x), gmin(fabsf(s) - fabsf(v.y), gmin(fabsf(s) - fabsf(v.z), fabsf(s) - fabsf(v.w))));
}
// textureGather component 0 of an R8 image with clamp-to-edge: (i0,j1), (i1,j1), (i1,j0), (i0,j0)
PLR_DI vec4 gatherR8(const ImgView& im, vec2 uv) {
    int i0, j0; float a, b;
    linearCoord(uv.x * (float)im.w, &i0, &a);
    linearCoord(uv.y * (float)im.h, &j0, &b);
    return vec4(addressedTexel2D<F_R8, CLAMP>(im, i0, j0 + 1).x, addressedTexel2D<F_R8, CLAMP>(im, i0 + 1, j0 + 1).x, addressedTexel2D<F_R8, CLAMP>(im, i0 + 1, j0).x,
                addressedTexel2D<F_R8, CLAMP>(im, i0, j0).x);
}
// temporalSupersampling.comp:39-55
PLR_DI float closestNeighbourhoodDepth(const ImgView& depthBuffer, vec2 uv, const GlobalUbo* g) {
    const vec2 texelSize(1.f / (float)g->screenResolution[0], 1.f / (float)g->screenResolution[1]);
    const int ox[9] = {-1, 0, 1, -1, 0, 1, -1, 0, 1}, oy[9] = {-1, -1, -1, 0, 0, 0, 1, 1, 1};
    float closestDepth = sampleNearest2D<F_D32, CLAMP>(depthBuffer, uv + vec2((float)ox[0], (float)oy[0]) * texelSize).x;
#pragma unroll
    for (int i = 1; i < 9; i++) closestDepth = gmax(sampleNearest2D<F_D32, CLAMP>(depthBuffer, uv + vec2((float)ox[i], (float)oy[i]) * texelSize).x, closestDepth);
    return linearizeDepth(closestDepth, g->nearPlane, g->farPlane);
}

// temporalSupersampling.comp:57-110
template <bool TONEMAP>
__global__ __launch_bounds__(256) void temporalSupersamplingKernel(ImgView currentFrame, ImgView lastFrame, ImgView target, ImgView velocityBuffer, ImgView currentDepth,
                                                                   ImgView lastDepth, ImgView currentLum, ImgView lastLum, const GlobalUbo* __restrict__ g, int coverW,
                                                                   int coverH, int yBase) {
    const int px = (int)(blockIdx.x * 64u + (threadIdx.x & 63u));
    const int py = yBase + (int)(blockIdx.y * 4u + (threadIdx.x >> 6));
    if (px >= coverW || py >= coverH) return;
    const vec2 texelSize(1.f / (float)g->screenResolution[0], 1.f / (float)g->screenResolution[1]);
    const vec2 uvCurrent = (vec2((float)px, (float)py) + vec2(0.5f)) * texelSize;
    vec2 motion;
    {   // getClosestFragmentMotion (temporalReprojection.inc:67-83)
        float closest = 0.f;
        int ox = 0, oy = 0;
#pragma unroll
        for (int x = -1; x <= 1; x++)
#pragma unroll
            for (int y = -1; y <= 1; y++) {
                const float d = texelFetch2D<F_D32>(currentDepth, px + x, py + y).x;
                if (d > closest) { closest = d; ox = x; oy = y; }
            }
        const vec4 m = texelFetch2D<F_RG16SN>(velocityBuffer, px + ox, py + oy);
        motion = vec2(m.x, m.y);
    }
    const vec2 uvLast = uvCurrent + motion;
    vec3 currentSample = sampleLinear2D<F_R11G11B10, CLAMP>(currentFrame, uvCurrent).xyz();
    vec3 lastSample = sampleLinear2D<F_R11G11B10, CLAMP>(lastFrame, uvLast).xyz();
    if (TONEMAP) { currentSample = taaTonemap(currentSample); lastSample = taaTonemap(lastSample); }
    const vec4 cl = gatherR8(currentLum, uvCurrent), ll = gatherR8(lastLum, uvLast);
    const float contrast = minAbsoluteDifference(cl.x, ll) + minAbsoluteDifference(cl.y, ll) + minAbsoluteDifference(cl.z, ll) + minAbsoluteDifference(cl.w, ll);
    const bool contrastTest = contrast < 0.5f;
    const float cd = closestNeighbourhoodDepth(currentDepth, uvCurrent, g), ld = closestNeighbourhoodDepth(lastDepth, uvLast, g);
    const bool depthTest = fabsf(cd - ld) < 1.f;
    const bool outOfScreen = uvLast.x < 0.f || uvLast.y < 0.f || uvLast.x > 1.f || uvLast.y > 1.f;
    const float blendFactor = (contrastTest && depthTest && !outOfScreen) ? 0.5f : 0.f;
    vec3 color = currentSample * (1.f - blendFactor) + lastSample * blendFactor;
    if (TONEMAP) color = taaTonemapReverse(color);
    Texel<F_R11G11B10>::store(target.ptr, (size_t)py * (size_t)target.w + px, vec4(color, 1.f));
}
static int launchTemporalSupersampling(const PassCtx& c) {
    if (int rc = c.needGlobal()) return rc;
    if (int rc = c.needSampled(1, F_R11G11B10, "temporalSupersampling currentFrame")) return rc;
    if (int rc = c.needSampled(2, F_R11G11B10, "temporalSupersampling lastFrame")) return rc;
    if (int rc = c.needStorage(3, F_R11G11B10, "temporalSupersampling targetImage")) return rc;
    if (int rc = c.needSampled(4, F_RG16SN, "temporalSupersampling velocityBuffer")) return rc;
    if (int rc = c.needSampled(5, F_D32, "temporalSupersampling currentDepthBuffer")) return rc;
    if (int rc = c.needSampled(6, F_D32, "temporalSupersampling lastDepthBuffer")) return rc;
    if (int rc = c.needSampled(7, F_R8, "temporalSupersampling currentLuminanceTexture")) return rc;
    if (int rc = c.needSampled(8, F_R8, "temporalSupersampling lastLuminanceTexture")) return rc;
    const bool tonemap = c.specBool(0, false);
    const ImgView& out = c.storage[3];
    const PassCtx::RowSpan rs = c.rowSpan(out.h);
    const int w = std::min((int)(c.dispatch[0] * 8u), out.w), h = rs.y1, y0 = rs.y0;
    if (w <= 0 || h <= y0) return 0;
    const dim3 grid(divUp((unsigned)w, 64u), divUp((unsigned)(h - y0), 4u));
    if (tonemap) temporalSupersamplingKernel<true><<<grid, 256, 0, c.stream>>>(c.sampled[1], c.sampled[2], out, c.sampled[4], c.sampled[5], c.sampled[6], c.sampled[7], c.sampled[8], c.global, w, h, y0);
    else temporalSupersamplingKernel<false><<<grid, 256, 0, c.stream>>>(c.sampled[1], c.sampled[2], out, c.sampled[4], c.sampled[5], c.sampled[6], c.sampled[7], c.sampled[8], c.global, w, h, y0);
    PLR_CHECK_LAUNCH(c);
    return 0;
}
PLR_REGISTER_SHADER("temporalSupersampling.comp", launchTemporalSupersampling);

} // namespace plr
