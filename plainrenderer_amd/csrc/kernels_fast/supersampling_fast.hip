// PLR_MATH_FAST kernels of the optional TAA stage (TAASettings::useSeparateSupersampling, Techniques/TAA.cpp:85-137):
// colorToLuminance.comp:14-21 and temporalSupersampling.comp:23-110 (exact set: kernels_exact/taa.hip).
//
// What a pixel of the supersampling pass DECIDES - which 3x3 neighbour's motion vector it takes, where that reprojects to, the 8-bit sub-texel weights of the
// history tap, the four luminance texels of each gather, the closest depth of both neighbourhoods, the contrast and depth rejection tests - is computed with the
// exact set's arithmetic (this file is built without contraction and with IEEE division): the decisions are the oracle's by construction. What changes:
//  * the nine current-frame depths of a pixel come from a (64 + 2) x (4 + 2) tile the block stages in LDS once (6.2 loads per pixel -> 1.5): both the closest
//    fragment's motion vector (texelFetch, zero outside the image) and the closest neighbourhood depth (clamp-to-edge: the maximum over the texels inside the image,
//    and depths are >= 0) read it;
//  * the current sample is the pixel's own texel when the bilinear tap sits on a texel centre (always, when the images have the screen's size: one fetch for four);
//  * tonemap / inverse tonemap divide by Newton-corrected v_rcp_f32 quotients (three instructions per channel) instead of IEEE divisions (ten), in the shader's
//    operation order - the inverse tonemap amplifies every ulp by the pixel's luminance, see the kernel; the R11G11B10 encoder is the fast set's.
// Stated result: every channel of every pixel within one R11G11B10 code of the oracle (tests/test_hiz_bloom_taa.py); the luminance image is bit exact.
// PLR_BUILD_FLAGS: -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt
#include "../backend.h"
#include "../device/shading_common.h"
#include "../device/fastmath.h"

namespace plr {
namespace fastss {

// four pixels per lane: one 16-byte load, one dword store; the arithmetic (and so the R8 rounding) is the exact kernel's
__global__ __launch_bounds__(256) void colorToLuminanceFastKernel(ImgView src, ImgView dst, int quadsPerRow, int coverH, int yBase) {
    const int q = (int)(blockIdx.x * 64u + (threadIdx.x & 63u));
    const int py = yBase + (int)(blockIdx.y * 4u + (threadIdx.x >> 6));
    if (q >= quadsPerRow || py >= coverH) return;
    const uint4 t = ((const uint4*)((const uint32_t*)src.ptr + (size_t)py * (size_t)src.w))[q];
    const uint32_t texels[4] = {t.x, t.y, t.z, t.w};
    uint32_t out = 0u;
#pragma unroll
    for (int k = 0; k < 4; k++) out |= encodeUnorm8(computeLuminance(unpackR11G11B10(texels[k]))) << (8 * k);
    ((uint32_t*)((uint8_t*)dst.ptr + (size_t)py * (size_t)dst.w))[q] = out;
}

static int launchColorToLuminance(const PassCtx& c) {
    if (int rc = c.needSampled(0, F_R11G11B10, "colorToLuminance srcTexture")) return rc;
    if (int rc = c.needStorage(1, F_R8, "colorToLuminance dstImage")) return rc;
    const ImgView &src = c.sampled[0], &dst = c.storage[1];
    const PassCtx::RowSpan rs = c.rowSpan(dst.h);
    const int w = std::min((int)(c.dispatch[0] * 8u), dst.w), h = rs.y1, y0 = rs.y0;
    if (w <= 0 || h <= y0) return 0;
    // whole rows of images whose width is a multiple of four (16-byte loads, dword stores): anything else takes the general kernel
    if (w != dst.w || src.w != dst.w || src.h < h || (dst.w & 3)) return kUseGeneralKernel;
    colorToLuminanceFastKernel<<<dim3(divUp((unsigned)(w / 4), 64u), divUp((unsigned)(h - y0), 4u)), 256, 0, c.stream>>>(src, dst, w / 4, h, y0);
    PLR_CHECK_LAUNCH(c);
    return 0;
}

// temporalSupersampling.comp:23-29
PLR_DI float minAbsoluteDifference(float s, const float v[4]) {
    return gmin(fabsf(s) - fabsf(v[0]), gmin(fabsf(s) - fabsf(v[1]), gmin(fabsf(s) - fabsf(v[2]), fabsf(s) - fabsf(v[3]))));
}
// textureGather component 0 of an R8 image with clamp-to-edge: (i0, j1), (i1, j1), (i1, j0), (i0, j0); c / 255 in its three-instruction form (device/image.h)
PLR_DI void gatherR8(const ImgView& im, float u, float v, float out[4]) {
    int i0, j0; float a, b;
    linearCoord(u * (float)im.w, &i0, &a);
    linearCoord(v * (float)im.h, &j0, &b);
    const uint8_t* p = (const uint8_t*)im.ptr;
    const int x0 = clampi(i0, im.w), x1 = clampi(i0 + 1, im.w);
    const uint32_t r0 = fastm::texelIndex(0u, (uint32_t)clampi(j0, im.h), (uint32_t)im.w), r1 = fastm::texelIndex(0u, (uint32_t)clampi(j0 + 1, im.h), (uint32_t)im.w);
    out[0] = decodeUnorm8Newton(p[r1 + x0]); out[1] = decodeUnorm8Newton(p[r1 + x1]); out[2] = decodeUnorm8Newton(p[r0 + x1]); out[3] = decodeUnorm8Newton(p[r0 + x0]);
}

template <bool TONEMAP>
__global__ __launch_bounds__(256) void temporalSupersamplingFastKernel(ImgView currentFrame, ImgView lastFrame, ImgView target, ImgView velocityBuffer, ImgView currentDepth,
                                                                       ImgView lastDepth, ImgView currentLum, ImgView lastLum, const GlobalUbo* __restrict__ g, float tsx,
                                                                       float tsy, int coverW, int coverH, int yBase) {
    __shared__ float depthTile[6][66]; // current depths of rows by0 - 1 .. by0 + 4, columns bx0 - 1 .. bx0 + 64; 0 outside the image (texelFetch)
    const int lane = (int)(threadIdx.x & 63u), wave = (int)(threadIdx.x >> 6);
    const int bx0 = (int)blockIdx.x * 64, by0 = yBase + (int)blockIdx.y * 4;
    for (int e = (int)threadIdx.x; e < 6 * 66; e += 256) {
        const int r = e / 66, col = e - r * 66, x = bx0 - 1 + col, y = by0 - 1 + r;
        const bool inside = x >= 0 && y >= 0 && x < currentDepth.w && y < currentDepth.h;
        const uint32_t bits = ((const uint32_t*)currentDepth.ptr)[fastm::texelIndex((uint32_t)clampi(x, currentDepth.w), (uint32_t)clampi(y, currentDepth.h), (uint32_t)currentDepth.w)];
        depthTile[r][col] = u2f(bits & (inside ? 0xffffffffu : 0u)); // a mask, not a select: the load stays unconditional
    }
    __syncthreads();
    const int px = bx0 + lane, py = by0 + wave;
    if (px >= coverW || py >= coverH) return;
    const float uCur = ((float)px + 0.5f) * tsx, vCur = ((float)py + 0.5f) * tsy; // texelSize = 1 / screenResolution: the IEEE quotients, from the launcher

    // getClosestFragmentMotion (temporalReprojection.inc:67-83): x outer, y inner, strict comparison from 0; and the neighbourhood's closest depth (:39-55)
    float closest = 0.f;
    int ox = 0, oy = 0;
#pragma unroll
    for (int x = -1; x <= 1; x++)
#pragma unroll
        for (int y = -1; y <= 1; y++) {
            const float d = depthTile[wave + 1 + y][lane + 1 + x];
            if (d > closest) { closest = d; ox = x; oy = y; }
        }
    // (clamp-to-edge neighbourhood: the maximum over the neighbours inside the image = the maximum over the tile with zeros outside, depths being >= 0)
    const float cd = linearizeDepth(closest, g->nearPlane, g->farPlane);
    float mx = 0.f, my = 0.f;
    {
        const int vx = px + ox, vy = py + oy;
        if (vx >= 0 && vy >= 0 && vx < velocityBuffer.w && vy < velocityBuffer.h) {
            const uint32_t u = ((const uint32_t*)velocityBuffer.ptr)[fastm::texelIndex((uint32_t)vx, (uint32_t)vy, (uint32_t)velocityBuffer.w)];
            mx = decodeSnorm16((int32_t)(int16_t)(u & 0xffffu)); my = decodeSnorm16((int32_t)(int16_t)(u >> 16));
        }
    }
    const float uLast = uCur + mx, vLast = vCur + my;

    // closest depth of the 3x3 around the reprojected position in last frame's depth buffer: nearest, clamp-to-edge, the shader's coordinates
    float lastClosest;
    {
        const uint32_t* ld = (const uint32_t*)lastDepth.ptr;
        const float fw = (float)lastDepth.w, fh = (float)lastDepth.h;
        auto tap = [&](int x, int y) {
            const int tx = clampTo(floorToInt((uLast + (float)x * tsx) * fw), lastDepth.w - 1), ty = clampTo(floorToInt((vLast + (float)y * tsy) * fh), lastDepth.h - 1);
            return u2f(ld[fastm::texelIndex((uint32_t)tx, (uint32_t)ty, (uint32_t)lastDepth.w)]);
        };
        float d[9];
        const int oxs[9] = {-1, 0, 1, -1, 0, 1, -1, 0, 1}, oys[9] = {-1, -1, -1, 0, 0, 0, 1, 1, 1};
#pragma unroll
        for (int i = 0; i < 9; i++) d[i] = tap(oxs[i], oys[i]); // all nine in flight
        lastClosest = d[0];
#pragma unroll
        for (int i = 1; i < 9; i++) lastClosest = gmax(d[i], lastClosest);
    }
    const float ldLinear = linearizeDepth(lastClosest, g->nearPlane, g->farPlane);

    float cl[4], ll[4];
    gatherR8(currentLum, uCur, vCur, cl);
    gatherR8(lastLum, uLast, vLast, ll);
    const float contrast = minAbsoluteDifference(cl[0], ll) + minAbsoluteDifference(cl[1], ll) + minAbsoluteDifference(cl[2], ll) + minAbsoluteDifference(cl[3], ll);
    const bool contrastTest = contrast < 0.5f;
    const bool depthTest = fabsf(cd - ldLinear) < 1.f;
    const bool outOfScreen = uLast < 0.f || vLast < 0.f || uLast > 1.f || vLast > 1.f;
    const float blendFactor = (contrastTest && depthTest && !outOfScreen) ? 0.5f : 0.f;

    // the two colour taps (linear, clamp-to-edge): footprints and 8-bit weights as the sampler contract has them (device/image.h linearCoord)
    auto bilinear = [&](const ImgView& im, float u, float v, bool maybeCentre) -> vec3 {
        int i0, j0; float a, b;
        linearCoord(u * (float)im.w, &i0, &a);
        linearCoord(v * (float)im.h, &j0, &b);
        const uint32_t* t = (const uint32_t*)im.ptr;
        const int x0 = clampi(i0, im.w), x1 = clampi(i0 + 1, im.w);
        const uint32_t r0 = fastm::texelIndex(0u, (uint32_t)clampi(j0, im.h), (uint32_t)im.w), r1 = fastm::texelIndex(0u, (uint32_t)clampi(j0 + 1, im.h), (uint32_t)im.w);
        // a tap on a texel centre has weights 1, 0, 0, 0: one fetch (wave-uniform test; the other three texels are finite numbers times zero)
        if (maybeCentre && __builtin_amdgcn_ballot_w64(a != 0.f || b != 0.f) == 0ull) return unpackR11G11B10(t[r0 + x0]);
        const vec3 t00 = unpackR11G11B10(t[r0 + x0]), t10 = unpackR11G11B10(t[r0 + x1]), t01 = unpackR11G11B10(t[r1 + x0]), t11 = unpackR11G11B10(t[r1 + x1]);
        {
#pragma clang fp contract(fast)
            const float w00 = (1.f - a) * (1.f - b), w10 = a * (1.f - b), w01 = (1.f - a) * b, w11 = a * b;
            return vec3(t00.x * w00 + t10.x * w10 + t01.x * w01 + t11.x * w11, t00.y * w00 + t10.y * w10 + t01.y * w01 + t11.y * w11,
                        t00.z * w00 + t10.z * w10 + t01.z * w01 + t11.z * w11);
        }
    };
    vec3 cur = bilinear(currentFrame, uCur, vCur, true), last = bilinear(lastFrame, uLast, vLast, false);
    // tonemap c / (1 + lum(c)) and its inverse c / (1 - lum(c)) in the shader's operation order. The inverse cancels: 1 - lum of a tonemapped colour is 1 / (1 + lum),
    // so ONE ulp in the tonemapped colour is lum ulps in the result - a v_rcp_f32 multiply (1 ulp) measured two R11G11B10 codes off on bright HDR pixels. The
    // quotients are therefore Newton-corrected (the residual is exact in an FMA: the correctly rounded quotient but for near-ties), three instructions per channel
    // behind one shared v_rcp_f32 instead of a ten-instruction IEEE division per channel.
    auto divide3 = [](vec3 c, float den) {
        const float r = __builtin_amdgcn_rcpf(den);
        auto quot = [&](float a) { const float q = a * r; return __builtin_fmaf(__builtin_fmaf(-den, q, a), r, q); };
        return vec3(quot(c.x), quot(c.y), quot(c.z));
    };
    if (TONEMAP) { cur = divide3(cur, 1.f + computeLuminance(cur)); last = divide3(last, 1.f + computeLuminance(last)); }
    vec3 color = cur * (1.f - blendFactor) + last * blendFactor;
    if (TONEMAP) color = divide3(color, 1.f - computeLuminance(color));
    ((uint32_t*)target.ptr)[fastm::texelIndex((uint32_t)px, (uint32_t)py, (uint32_t)target.w)] = packR11G11B10(color);
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
    // the depth tile stands for both of the shader's 3x3 current-depth neighbourhoods only when the depth buffer has the screen's size (its nearest taps at
    // uv +- one texel are then the pixel's neighbours); the host must know the screen size (a host whose global buffer the backend cannot read: general kernel)
    if (!c.globalHost) return kUseGeneralKernel;
    const int sw = c.globalHost->screenResolution[0], sh = c.globalHost->screenResolution[1];
    if (sw <= 0 || sh <= 0 || c.sampled[5].w != sw || c.sampled[5].h != sh || out.w != sw || out.h != sh || sw >= (1 << 15) || sh >= (1 << 15)) return kUseGeneralKernel;
    const float tsx = 1.f / (float)sw, tsy = 1.f / (float)sh; // IEEE single-precision quotients (host code without fast-math): vec2(1) / screenResolution
    const dim3 grid(divUp((unsigned)w, 64u), divUp((unsigned)(h - y0), 4u));
    if (tonemap) temporalSupersamplingFastKernel<true><<<grid, 256, 0, c.stream>>>(c.sampled[1], c.sampled[2], out, c.sampled[4], c.sampled[5], c.sampled[6], c.sampled[7], c.sampled[8], c.global, tsx, tsy, w, h, y0);
    else temporalSupersamplingFastKernel<false><<<grid, 256, 0, c.stream>>>(c.sampled[1], c.sampled[2], out, c.sampled[4], c.sampled[5], c.sampled[6], c.sampled[7], c.sampled[8], c.global, tsx, tsy, w, h, y0);
    PLR_CHECK_LAUNCH(c);
    return 0;
}

} // namespace fastss

static int fastss_color_to_luminance(const PassCtx& c) { return fastss::launchColorToLuminance(c); }
PLR_REGISTER_SHADER_FAST("colorToLuminance.comp", fastss_color_to_luminance);
static int fastss_temporal_supersampling(const PassCtx& c) { return fastss::launchTemporalSupersampling(c); }
PLR_REGISTER_SHADER_FAST("temporalSupersampling.comp", fastss_temporal_supersampling);

} // namespace plr
