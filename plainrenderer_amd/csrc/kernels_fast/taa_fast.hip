// PLR_MATH_FAST variant of temporalFilter.comp (exact variant: kernels_exact/taa.hip).
//
// Same algorithm; restructured where the shader repeats work:
//  * tonemap(c) = c / (1 + lum(c)) is one v_rcp_f32 and three multiplies instead of three IEEE divisions
//  * the contrast term only needs luminances, and luminance is linear: lum(tonemap(c)) = lum(c) / (1 + lum(c)) and
//    lum(bilinear(texels)) = bilinear(lum(texels)). The nine bilinear history taps of sampleNeighbourhood(historyBufferSrc, uv + motion)
//    sit one texel apart, so they share one 4x4 texel footprint and one pair of sub-texel weights: 16 texel decodes and 27 scalar
//    lerps replace 36 decodes and 108 vector multiply-adds
//  * clipAABB uses reciprocals; FMA contraction is on
// A tap whose 8-bit sub-texel weight sits on a quantisation boundary can differ by 1/256 between neighbouring taps in the exact
// kernel; here all nine use the centre tap's weights. Stated tolerance: tests/test_fast_kernels.py.
#include "../backend.h"
#include "../device/shading_common.h"
#include "../device/fastmath.h"
#include "../device/buffer_fetch.h"

namespace plr {
namespace fasttaa {

PLR_DI float rcpf(float x) { return __builtin_amdgcn_rcpf(x); }
PLR_DI float lum(vec3 c) { return c.x * 0.21f + c.y * 0.72f + c.z * 0.07f; }
PLR_DI vec3 tonemapF(vec3 c) { return c * rcpf(1.f + lum(c)); }
PLR_DI vec3 tonemapReverseF(vec3 c) { return c * rcpf(1.f - lum(c)); }

PLR_DI vec3 historyTap(const ImgView& im, float u, float v) {
    int i0, j0; float a, b;
    linearCoord(u * (float)im.w, &i0, &a);
    linearCoord(v * (float)im.h, &j0, &b);
    const uint32_t* base = (const uint32_t*)im.ptr;
    const int x0 = clampi(i0, im.w), x1 = clampi(i0 + 1, im.w);
    const size_t r0 = (size_t)clampi(j0, im.h) * (size_t)im.w, r1 = (size_t)clampi(j0 + 1, im.h) * (size_t)im.w;
    const vec3 t00 = unpackR11G11B10(base[r0 + x0]), t10 = unpackR11G11B10(base[r0 + x1]);
    const vec3 t01 = unpackR11G11B10(base[r1 + x0]), t11 = unpackR11G11B10(base[r1 + x1]);
    const vec3 top = t00 + (t10 - t00) * a, bot = t01 + (t11 - t01) * a;
    return top + (bot - top) * b;
}

// texelFetch of the motion buffer with the out-of-range test as a mask on the loaded word (snorm 0 decodes to 0.0): neither a branch per pixel row
// nor a select the compiler could turn back into one around the load
PLR_DI vec4 motionFetch(const ImgView& im, int x, int y) {
    const bool inside = (uint32_t)x < (uint32_t)im.w && (uint32_t)y < (uint32_t)im.h;
    const uint32_t u = ((const uint32_t*)im.ptr)[fastm::texelIndex((uint32_t)clampi(x, im.w), (uint32_t)clampi(y, im.h), (uint32_t)im.w)] & (inside ? 0xffffffffu : 0u);
    return vec4(decodeSnorm16((int32_t)(int16_t)(u & 0xffffu)), decodeSnorm16((int32_t)(int16_t)(u >> 16)), 0.f, inside ? 1.f : 0.f);
}

// the same through a buffer descriptor of the image's texels (device/buffer_fetch.h: no address arithmetic on the VALU)
PLR_DI vec4 motionFetch(BufferDesc texels, const ImgView& im, int x, int y) {
    const bool inside = (uint32_t)x < (uint32_t)im.w && (uint32_t)y < (uint32_t)im.h;
    const uint32_t u = fetch32(texels, fastm::texelIndex((uint32_t)clampi(x, im.w), (uint32_t)clampi(y, im.h), (uint32_t)im.w)) & (inside ? 0xffffffffu : 0u);
    return vec4(decodeSnorm16((int32_t)(int16_t)(u & 0xffffu)), decodeSnorm16((int32_t)(int16_t)(u >> 16)), 0.f, inside ? 1.f : 0.f);
}

PLR_DI vec3 clipAABB(vec3 target, vec3 bbMin, vec3 bbMax) {
    const vec3 center = 0.5f * (bbMax + bbMin);
    const vec3 extend = 0.5f * (bbMax - bbMin) + vec3(0.0001f);
    const vec3 toTarget = target - center;
    const vec3 n(fabsf(toTarget.x) * rcpf(extend.x), fabsf(toTarget.y) * rcpf(extend.y), fabsf(toTarget.z) * rcpf(extend.z));
    const float maxComponent = __builtin_fmaxf(n.x, __builtin_fmaxf(n.y, n.z)); // v_max3_f32: a NaN operand loses, like gmax, without its branches
    if (maxComponent < 1.f) return target;
    return center + toTarget * rcpf(maxComponent);
}

PLR_DI float catmullRomWeight1D(float d) {
    const float d1 = fabsf(d), d2 = d1 * d1, d3 = d2 * d1;
    if (d1 <= 1.f) return (1.f / 6.f) * (9.f * d3 - 15.f * d2 + 6.f);
    if (d1 <= 2.f) return (1.f / 6.f) * (-3.f * d3 + 15.f * d2 - 24.f * d + 12.f);
    return 0.f;
}

struct ResolveWeights { float w[9]; };

template <bool CLIP, bool DILATE, int TECH, bool TONEMAP>
__global__ __launch_bounds__(256) void temporalFilterFastKernel(ImgView current, ImgView output, ImgView historyDst, ImgView historySrc, ImgView motionBuffer,
                                                                ImgView depthBuffer, const ResolveWeights* __restrict__ rwp, const GlobalUbo* __restrict__ g,
                                                                int coverW, int coverH, int yBase, int xBase) {
    const int px = xBase + (int)(blockIdx.x * 64u + (threadIdx.x & 63u)); // columns [xBase, coverW) (tile rendering: PassCtx::colSpan)
    const int py = yBase + (int)(blockIdx.y * 4u + (threadIdx.x >> 6));
    if (px >= coverW || py >= coverH) return;
    const float tsx = 1.f / (float)output.w, tsy = 1.f / (float)output.h;
    const float u0 = ((float)px + 0.5f) * tsx, v0 = ((float)py + 0.5f) * tsy;

    // current 3x3 (texel centres -> plain fetches), tonemapped; n[x+1][y+1]
    vec3 n[3][3];
    float nl[3][3]; // luminance of the (tonemapped) neighbourhood
    const uint32_t* cur = (const uint32_t*)current.ptr;
#pragma unroll
    for (int x = -1; x <= 1; x++)
#pragma unroll
        for (int y = -1; y <= 1; y++) {
            const vec3 c = unpackR11G11B10(cur[(size_t)clampi(py + y, current.h) * (size_t)current.w + clampi(px + x, current.w)]);
            const float l = lum(c);
            if (TONEMAP) {
                const float s = rcpf(1.f + l);
                n[x + 1][y + 1] = c * s;
                nl[x + 1][y + 1] = l * s;
            } else {
                n[x + 1][y + 1] = c;
                nl[x + 1][y + 1] = l;
            }
        }
    vec3 mn = n[0][0], mx = n[0][0];
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = 0; j < 3; j++) { mn = vmin(mn, n[i][j]); mx = vmax(mx, n[i][j]); }
    const ResolveWeights rw = *rwp;
    vec3 currentColor = n[0][0] * rw.w[0] + n[1][0] * rw.w[1] + n[2][0] * rw.w[2] + n[0][1] * rw.w[3] + n[1][1] * rw.w[4] + n[2][1] * rw.w[5] +
                        n[0][2] * rw.w[6] + n[1][2] * rw.w[7] + n[2][2] * rw.w[8];

    vec2 motion;
    {
        int ox = 0, oy = 0;
        if (DILATE) {
            float closest = 0.f;
#pragma unroll
            for (int x = -1; x <= 1; x++)
#pragma unroll
                for (int y = -1; y <= 1; y++) {
                    const float d = texelFetch2D<F_D32>(depthBuffer, px + x, py + y).x;
                    if (d > closest) { closest = d; ox = x; oy = y; }
                }
        }
        const vec4 m = motionFetch(motionBuffer, px + ox, py + oy);
        motion = vec2(m.x, m.y);
    }
    const float rpx = u0 + motion.x, rpy = v0 + motion.y;

    vec3 historySample;
    if (TECH == 0) historySample = historyTap(historySrc, rpx, rpy);
    else {
        const float ix = (float)px + 0.5f + motion.x * (float)g->screenResolution[0], iy = (float)py + 0.5f + motion.y * (float)g->screenResolution[1];
        const float tx = floorf(ix - 0.5f) + 0.5f, ty = floorf(iy - 0.5f) + 0.5f;
        const float fx = ix - tx, fy = iy - ty;
        if (TECH == 1) {
            const float wx[4] = {catmullRomWeight1D(fx + 1.f), catmullRomWeight1D(fx), catmullRomWeight1D(1.f - fx), catmullRomWeight1D(2.f - fx)};
            const float wy[4] = {catmullRomWeight1D(fy + 1.f), catmullRomWeight1D(fy), catmullRomWeight1D(1.f - fy), catmullRomWeight1D(2.f - fy)};
            vec3 r(0.f);
            for (int j = 0; j < 4; j++)
                for (int i = 0; i < 4; i++) r = r + historyTap(historySrc, (tx + (float)(i - 1)) * tsx, (ty + (float)(j - 1)) * tsy) * (wx[i] * wy[j]);
            historySample = r;
        } else {
            const float fx2 = fx * fx, fx3 = fx2 * fx, fy2 = fy * fy, fy3 = fy2 * fy;
            const float w0x = -0.5f * fx3 + fx2 - 0.5f * fx, w1x = 1.5f * fx3 - 2.5f * fx2 + 1.f, w2x = -1.5f * fx3 + 2.f * fx2 + 0.5f * fx, w3x = 0.5f * fx3 - 0.5f * fx2;
            const float w0y = -0.5f * fy3 + fy2 - 0.5f * fy, w1y = 1.5f * fy3 - 2.5f * fy2 + 1.f, w2y = -1.5f * fy3 + 2.f * fy2 + 0.5f * fy, w3y = 0.5f * fy3 - 0.5f * fy2;
            const float wBx = w1x + w2x, wBy = w1y + w2y;
            const float u0c = (tx - 1.f) * tsx, uT = (tx + w2x * rcpf(wBx)) * tsx, u3 = (tx + 2.f) * tsx;
            const float v0c = (ty - 1.f) * tsy, vT = (ty + w2y * rcpf(wBy)) * tsy, v3 = (ty + 2.f) * tsy;
            if (TECH == 2) {
                historySample = historyTap(historySrc, u0c, v0c) * (w0x * w0y) + historyTap(historySrc, u0c, vT) * (w0x * wBy) + historyTap(historySrc, u0c, v3) * (w0x * w3y) +
                                historyTap(historySrc, uT, v0c) * (wBx * w0y) + historyTap(historySrc, uT, vT) * (wBx * wBy) + historyTap(historySrc, uT, v3) * (wBx * w3y) +
                                historyTap(historySrc, u3, v0c) * (w3x * w0y) + historyTap(historySrc, u3, vT) * (w3x * wBy) + historyTap(historySrc, u3, v3) * (w3x * w3y);
            } else if (TECH == 3) {
                const float wa = w0x * wBy, wb = wBx * w0y, wc = wBx * wBy, wd = wBx * w3y, we = w3x * wBy;
                const vec3 r = historyTap(historySrc, u0c, vT) * wa + historyTap(historySrc, uT, v0c) * wb + historyTap(historySrc, uT, vT) * wc +
                               historyTap(historySrc, uT, v3) * wd + historyTap(historySrc, u3, vT) * we;
                historySample = r * rcpf(wa + wb + wc + wd + we);
            } else {
                // Bicubic1Tap: sum_k w_k (h + d_k) / sum_k w_k = h + (sum_k w_k d_k) / sum_k w_k, with d = neighbour - centre of the
                // (tonemapped) current frame; the history tap itself is not tonemapped yet (bicubicSampling.inc:169-176)
                const vec3 h = historyTap(historySrc, uT, vT);
                const float wa = w0x * wBy, wb = wBx * w0y, wc = wBx * wBy, wd = wBx * w3y, we = w3x * wBy;
                const vec3 c = n[1][1];
                const vec3 acc = (n[0][1] - c) * wa + (n[1][0] - c) * wb + (n[1][2] - c) * wd + (n[2][1] - c) * we;
                historySample = h + acc * rcpf(wa + wb + wc + wd + we);
            }
        }
    }
    if (TONEMAP) historySample = tonemapF(historySample);
    if (CLIP) historySample = clipAABB(historySample, mn, mx);
    else historySample = vclamp(historySample, mn, mx);
    if (anyNan(historySample)) historySample = currentColor;

    const float cc = nl[1][1];
    const float currentContrast = fabsf(nl[0][0] - cc) + fabsf(nl[1][0] - cc) + fabsf(nl[2][0] - cc) + fabsf(nl[0][2] - cc) + fabsf(nl[1][2] - cc) + fabsf(nl[2][2] - cc) +
                                  fabsf(nl[0][1] - cc) + fabsf(nl[2][1] - cc);
    // history neighbourhood luminances from one shared 4x4 footprint
    float lastContrast;
    {
        int i0, j0; float a, b;
        linearCoord(rpx * (float)historySrc.w, &i0, &a);
        linearCoord(rpy * (float)historySrc.h, &j0, &b);
        const uint32_t* hist = (const uint32_t*)historySrc.ptr;
        float tl[4][4]; // [row][col] luminance of texels (i0-1 .. i0+2, j0-1 .. j0+2)
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const size_t row = (size_t)clampi(j0 - 1 + r, historySrc.h) * (size_t)historySrc.w;
#pragma unroll
            for (int c = 0; c < 4; c++) tl[r][c] = lum(unpackR11G11B10(hist[row + clampi(i0 - 1 + c, historySrc.w)]));
        }
        float hl[3][3]; // [x+1][y+1]
#pragma unroll
        for (int y = 0; y < 3; y++) {
            float rowLerp[4];
#pragma unroll
            for (int c = 0; c < 4; c++) rowLerp[c] = tl[y][c] + (tl[y + 1][c] - tl[y][c]) * b;
#pragma unroll
            for (int x = 0; x < 3; x++) {
                const float l = rowLerp[x] + (rowLerp[x + 1] - rowLerp[x]) * a;
                hl[x][y] = TONEMAP ? l * rcpf(1.f + l) : l;
            }
        }
        const float hc = hl[1][1];
        lastContrast = fabsf(hl[0][0] - hc) + fabsf(hl[1][0] - hc) + fabsf(hl[2][0] - hc) + fabsf(hl[0][2] - hc) + fabsf(hl[1][2] - hc) + fabsf(hl[2][2] - hc) +
                       fabsf(hl[0][1] - hc) + fabsf(hl[2][1] - hc);
    }
    const float contrastChange = gclamp(fabsf(currentContrast - lastContrast), 0.f, 1.f);
    float blendFactor = gmix(0.13f, 0.03f, contrastChange);
    if (g->cameraCut) blendFactor = 1.f;
    if (rpx < 0.f || rpy < 0.f || rpx > 1.f || rpy > 1.f) {
        blendFactor = 1.f;
        currentColor = (n[0][0] + n[0][2] + n[2][0] + n[2][2]) * 0.0625f + (n[1][0] + n[0][1] + n[1][2] + n[2][1]) * 0.125f + n[1][1] * 0.25f;
    }
    vec3 color = historySample + (currentColor - historySample) * blendFactor;
    if (TONEMAP) color = tonemapReverseF(color);
    const uint32_t packed = packR11G11B10(color);
    if (px < historyDst.w && py < historyDst.h) ((uint32_t*)historyDst.ptr)[(size_t)py * (size_t)historyDst.w + px] = packed;
    ((uint32_t*)output.ptr)[(size_t)py * (size_t)output.w + px] = packed;
}

// ------------------------------------------------------------------------------------------------------------------------
// Strip kernel (history sampling Bilinear and Bicubic1Tap, the default).
//
// A CU's texture addresser retires roughly one wave-wide load per 16-22 cycles whatever its width, and the kernel above issues
// ~40 of them per pixel row segment (9 colour + 9 depth + 16 + 4 history texels ...): it is bound by load-instruction count.
// Here a wave owns a 62-pixel wide, 4-row tall strip: lane L holds column x0 - 1 + L, so the left and right neighbour columns of
// the 3x3 neighbourhood arrive by DPP wave shifts instead of loads (lanes 0 and 63 only feed their neighbours), and the wave walks
// down its rows with a three-row window in registers, loading each colour / depth row once. The 4x4 history footprint of the
// nine neighbourhood taps is four unaligned 16-byte row loads and also contains the bilinear history tap. Per output row a wave
// issues ~9 loads instead of ~40.
PLR_DI float fromLeft(float v) { return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x138, 0xf, 0xf, false)); }  // wave_shr:1
PLR_DI float fromRight(float v) { return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x130, 0xf, 0xf, false)); } // wave_shl:1
PLR_DI float min3f(float a, float b, float c) { return __builtin_fminf(__builtin_fminf(a, b), c); }
PLR_DI float max3f(float a, float b, float c) { return __builtin_fmaxf(__builtin_fmaxf(a, b), c); }

struct Column { float r, g, b, l, d; }; // (tonemapped) colour, its luminance, raw depth (0 outside the image, as texelFetch)
// (the neighbour columns' depths are not shifted: the closest-depth search shares one (depth, row) pair per column instead, see the kernel)
PLR_DI Column shiftFromLeft(const Column& c) { return {fromLeft(c.r), fromLeft(c.g), fromLeft(c.b), fromLeft(c.l), 0.f}; }
PLR_DI Column shiftFromRight(const Column& c) { return {fromRight(c.r), fromRight(c.g), fromRight(c.b), fromRight(c.l), 0.f}; }

#ifndef PLR_TAA_STRIP_ROWS
#define PLR_TAA_STRIP_ROWS 4 // experiment hook (PLR_EXTRA_FLAGS=-DPLR_TAA_STRIP_ROWS=2): rows per wave; fewer rows = fewer registers, more loads per output row
#endif
constexpr int kStripW = 62, kStripRows = PLR_TAA_STRIP_ROWS;
// History texels a wave stages in LDS (round 3; north star: framebuffer tiles through LDS): {luminance, packed texel} of the bounding box of its
// 62 x 4 pixels' 4x4 reprojection footprints. With a static or slowly moving camera that box is ~66 x 7 texels: a lane decodes ~8 texels per strip
// instead of 16 per pixel = 64 (the sixteen R11G11B10 decodes + luminances per pixel were 40 % of this kernel's instructions). A wave whose
// footprints spread further (fast motion, a disocclusion edge through the strip) takes the direct path: its pixels read their footprints from
// memory as before. The strip's reprojection (depth rows -> closest depth -> motion vectors) is resolved for all four rows first; the box's
// texels are then fetched with all loads of a lane in flight at once, next to the first colour rows of the resolve - a first version that
// staged in a rolled loop (one load, one decode, one store per trip) ran 171 us against 155 us for the unstaged kernel: eight exposed round
// trips per strip outweighed a fifth fewer instructions.
constexpr int kStageTexels = 1024, kStageMaxW = 128, kStagePerLane = kStageTexels / 64;

// minimum over the wave's lanes (lanes that do not take part pass INT_MAX); result valid in every lane
PLR_DI int waveMinI(int v) {
    constexpr int kIdentity = 0x7fffffff;
#define PLR_TAA_MIN_STEP(ctrl, rowMask) v = min(v, __builtin_amdgcn_update_dpp(kIdentity, v, ctrl, rowMask, 0xf, false))
    PLR_TAA_MIN_STEP(0x111, 0xf); PLR_TAA_MIN_STEP(0x112, 0xf); PLR_TAA_MIN_STEP(0x114, 0xf); PLR_TAA_MIN_STEP(0x118, 0xf); // row_shr:1, 2, 4, 8: lane 15 of every row holds the row's minimum
    PLR_TAA_MIN_STEP(0x142, 0xa); // row_bcast:15 into rows 1 and 3
    PLR_TAA_MIN_STEP(0x143, 0xc); // row_bcast:31 into rows 2 and 3: lane 63 holds the wave's minimum
#undef PLR_TAA_MIN_STEP
    return __builtin_amdgcn_readlane(v, 63);
}

// BANDED: the launch may cover two row ranges or run its edge rows first (band rendering, backend.h TwoRanges); a whole-frame launch carries none of that code
// (it cost the 128-register kernel four more spilled dwords)
template <bool CLIP, bool DILATE, int TECH, bool TONEMAP, bool BANDED>
#ifndef PLR_TAA_WAVES
#define PLR_TAA_WAVES 4
#endif
__global__ __launch_bounds__(256, PLR_TAA_WAVES) void temporalFilterStripKernel(ImgView current, ImgView output, ImgView historyDst, ImgView historySrc, ImgView motionBuffer,
                                                                 ImgView depthBuffer, const ResolveWeights* __restrict__ rwp, const GlobalUbo* __restrict__ g,
                                                                 int coverW, int coverH, int yBase, int xBase, TwoRanges ranges) {
    static_assert(TECH == 0 || TECH == 4, "strip kernel: Bilinear and Bicubic1Tap history sampling");
    __shared__ uint2 stage[4][kStageTexels]; // per wave: {luminance bits, packed texel}
    const int lane = (int)(threadIdx.x & 63u), wave = (int)(threadIdx.x >> 6);
    int blockCol = (int)blockIdx.x, blockRow = (int)blockIdx.y;
    if (BANDED) ranges.blockXY(&blockCol, &blockRow); // a launch over two row ranges, or the edge rows (tile rendering: and columns) first (backend.h TwoRanges)
    // the column this lane holds; it is an output column for lanes 1..62. Output columns [xBase, coverW) (tile rendering: PassCtx::colSpan)
    const int px = xBase + blockCol * kStripW + lane - 1;
    const int rowFirst = yBase + (blockRow * 4 + wave) * kStripRows;
    if (rowFirst >= coverH) { if (BANDED) ranges.edgeDone((int)blockIdx.y); return; } // wave-uniform (every wave of a rows-first launch reports in: TwoRanges::edgeDone)
    const bool isOutputLane = lane >= 1 && lane <= kStripW && px < coverW;
    const int xc = clampi(px, current.w);
    const bool xInDepth = px >= 0 && px < depthBuffer.w;
    // the strip's texel fetches go through buffer descriptors (device/buffer_fetch.h): the index is scaled by the addresser, not by a v_lshl_add_u64 per fetch
    const BufferDesc cur = texelBuffer(current.ptr, 4u), dep = texelBuffer(depthBuffer.ptr, 4u), mot = texelBuffer(motionBuffer.ptr, 4u);

    auto fetchRow = [&](int y) -> uint32_t { return fetch32(cur, fastm::texelIndex((uint32_t)xc, (uint32_t)clampi(y, current.h), (uint32_t)current.w)); };
    auto decodeRow = [&](uint32_t texel) -> Column {
        const vec3 c = unpackR11G11B10(texel);
        const float l = lum(c);
        Column o;
        if (TONEMAP) { const float s = rcpf(1.f + l); o.r = c.x * s; o.g = c.y * s; o.b = c.z * s; o.l = l * s; }
        else { o.r = c.x; o.g = c.y; o.b = c.z; o.l = l; }
        o.d = 0.f;
        return o;
    };
    auto loadRow = [&](int y) -> Column { return decodeRow(fetchRow(y)); };
    auto loadDepth = [&](int y) -> float { // raw depth, 0 outside the image (texelFetch)
        // the out-of-image case is a bit mask on the loaded word, not a select: a select lets the compiler sink the load into a branch of its own,
        // with an s_waitcnt vmcnt(0) inside - the six depth rows of a strip were six serial round trips
        const uint32_t bits = fetch32(dep, fastm::texelIndex((uint32_t)clampi(px, depthBuffer.w), (uint32_t)clampi(y, depthBuffer.h), (uint32_t)depthBuffer.w));
        return u2f(bits & ((xInDepth && y >= 0 && y < depthBuffer.h) ? 0xffffffffu : 0u));
    };

    const ResolveWeights rw = *rwp;
    const float tsx = 1.f / (float)output.w, tsy = 1.f / (float)output.h;
    const float resX = (float)g->screenResolution[0], resY = (float)g->screenResolution[1];
    const bool cameraCut = g->cameraCut != 0u;
    const BufferDesc hist = texelBuffer(historySrc.ptr, 4u);
    const int hw = historySrc.w, hh = historySrc.h;

    // ---- phase A: where do the strip's pixels reproject to? (motion vector at the closest depth of the 3x3, temporalFilter.comp:93-117)
    // all six colour rows of the strip's resolve: in flight through phase A (rows clamp to the image, so rows below the dispatch load something valid)
    uint32_t raw[kStripRows + 2];
#pragma unroll
    for (int j = 0; j < kStripRows + 2; j++) raw[j] = fetchRow(rowFirst - 1 + j);
    float mvx[kStripRows], mvy[kStripRows], wa[kStripRows], wb[kStripRows]; // motion, sub-texel weights of the footprint
    int fi[kStripRows], fj[kStripRows];                                     // i0, j0 of the footprint (texels i0 - 1 .. i0 + 2)
    int iLo = 0x7fffffff, iHi = 0x7fffffff, jLo = 0x7fffffff, jHi = 0x7fffffff; // min of i0, min of -i0, ...: one kind of reduction
    {
        float dw[kStripRows + 2];
        if (DILATE) {
#pragma unroll
            for (int j = 0; j < kStripRows + 2; j++) dw[j] = loadDepth(rowFirst - 1 + j); // all six depth rows at once
        }
        // three passes over the strip's rows without a branch between them (a row below the dispatch is computed like the others and masked at the
        // end): the four motion fetches depend on the depths only and go out together - with a wave-uniform `continue` per row the compiler kept
        // each row's fetch and its s_waitcnt vmcnt(0) in a block of its own, four more serial round trips per strip
        int ox[kStripRows], oy[kStripRows];
#pragma unroll
        for (int j = 0; j < kStripRows; j++) {
            // closest (largest reverse-Z) depth of the 3x3, scanned x outer / y inner by the reference loop with a strict comparison from 0: the first
            // column whose maximum is the overall maximum (> 0), and in it the first row that reaches it. Every lane finds (maximum, first row) of its own
            // column; the neighbours' pairs come over DPP: 18 compares / selects / shifts instead of 36 on the nine depths
            ox[j] = oy[j] = 0;
            if (DILATE) {
                float cd = dw[j]; int cy = -1;
                if (dw[j + 1] > cd) { cd = dw[j + 1]; cy = 0; }
                if (dw[j + 2] > cd) { cd = dw[j + 2]; cy = 1; }
                const float ld = fromLeft(cd), rd = fromRight(cd);
                const int ly = __builtin_bit_cast(int, fromLeft(__builtin_bit_cast(float, cy))), ry = __builtin_bit_cast(int, fromRight(__builtin_bit_cast(float, cy)));
                float closest = 0.f;
                if (ld > closest) { closest = ld; ox[j] = -1; oy[j] = ly; }
                if (cd > closest) { closest = cd; ox[j] = 0; oy[j] = cy; }
                if (rd > closest) { closest = rd; ox[j] = 1; oy[j] = ry; }
            }
        }
        vec4 m[kStripRows];
#pragma unroll
        for (int j = 0; j < kStripRows; j++) m[j] = motionFetch(mot, motionBuffer, px + ox[j], rowFirst + j + oy[j]);
#pragma unroll
        for (int j = 0; j < kStripRows; j++) {
            const int py = rowFirst + j;
            const bool rowLive = py < coverH; // wave-uniform
            mvx[j] = rowLive ? m[j].x : 0.f; mvy[j] = rowLive ? m[j].y : 0.f;
            const float rpx = ((float)px + 0.5f) * tsx + m[j].x, rpy = ((float)py + 0.5f) * tsy + m[j].y;
            linearCoord(rpx * (float)hw, &fi[j], &wa[j]);
            linearCoord(rpy * (float)hh, &fj[j], &wb[j]);
            if (!rowLive) { fi[j] = fj[j] = 0; wa[j] = wb[j] = 0.f; }
            if (isOutputLane && rowLive) { iLo = min(iLo, fi[j]); iHi = min(iHi, -fi[j]); jLo = min(jLo, fj[j]); jHi = min(jHi, -fj[j]); }
        }
    }
    // bounding box of the footprints (texels i0 - 1 .. i0 + 2 of every pixel), staged if it fits
    const int orgI = waveMinI(iLo) - 1, endI = -waveMinI(iHi) + 3, orgJ = waveMinI(jLo) - 1, endJ = -waveMinI(jHi) + 3; // [org, end)
    const int stageW = endI - orgI, stageH = endJ - orgJ, stageN = stageW * stageH;
    const bool staged = stageW > 0 && stageH > 0 && stageW <= kStageMaxW && stageN <= kStageTexels; // wave-uniform (a wave without output pixels: not staged)
    uint2* __restrict__ myStage = stage[wave];
    if (staged) {
        // entry e = lane + 64 k is box texel (e % stageW, e / stageW); all of a lane's loads are issued before the first is decoded
        const float invW = 1.f / (float)stageW;
        uint32_t tx[kStagePerLane];
#pragma unroll
        for (int k = 0; k < kStagePerLane; k++) {
            const int e = k * 64 + lane;
            tx[k] = 0u;
            if (k * 64 < stageN) { // wave-uniform
                const int r = (int)(((float)e + 0.5f) * invW), x = e - r * stageW; // exact for e < 1024, stageW <= 128
                if (e < stageN) tx[k] = fetch32(hist, __umul24((uint32_t)clampi(orgJ + r, hh), (uint32_t)hw) + (uint32_t)clampi(orgI + x, hw)); // clamp-to-edge resolved here
            }
        }
#pragma unroll
        for (int k = 0; k < kStagePerLane; k++) {
            const int e = k * 64 + lane;
            if (k * 64 < stageN && e < stageN) myStage[e] = make_uint2(f2u(lum(unpackR11G11B10(tx[k]))), tx[k]);
        }
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); // the wave's own LDS writes before its lanes read each other's entries
    }

    // ---- phase B: the resolve. Three-row window: C[0] = row y-1, C[1] = row y, C[2] = row y+1; L / R = the neighbour lanes' columns
    Column C[3], L[3], R[3];
    C[0] = decodeRow(raw[0]); C[1] = decodeRow(raw[1]);
    L[0] = shiftFromLeft(C[0]); R[0] = shiftFromRight(C[0]);
    L[1] = shiftFromLeft(C[1]); R[1] = shiftFromRight(C[1]);

#pragma unroll
    for (int j = 0; j < kStripRows; j++) {
        const int py = rowFirst + j;
        if (py >= coverH) break; // wave-uniform
        C[2] = decodeRow(raw[j + 2]);
        L[2] = shiftFromLeft(C[2]); R[2] = shiftFromRight(C[2]);

        // n[x+1][y+1] of the reference: x = -1 -> L, 0 -> C, +1 -> R; y = -1..1 -> window row 0..2
        const vec3 mn(min3f(min3f(L[0].r, C[0].r, R[0].r), min3f(L[1].r, C[1].r, R[1].r), min3f(L[2].r, C[2].r, R[2].r)),
                      min3f(min3f(L[0].g, C[0].g, R[0].g), min3f(L[1].g, C[1].g, R[1].g), min3f(L[2].g, C[2].g, R[2].g)),
                      min3f(min3f(L[0].b, C[0].b, R[0].b), min3f(L[1].b, C[1].b, R[1].b), min3f(L[2].b, C[2].b, R[2].b)));
        const vec3 mx(max3f(max3f(L[0].r, C[0].r, R[0].r), max3f(L[1].r, C[1].r, R[1].r), max3f(L[2].r, C[2].r, R[2].r)),
                      max3f(max3f(L[0].g, C[0].g, R[0].g), max3f(L[1].g, C[1].g, R[1].g), max3f(L[2].g, C[2].g, R[2].g)),
                      max3f(max3f(L[0].b, C[0].b, R[0].b), max3f(L[1].b, C[1].b, R[1].b), max3f(L[2].b, C[2].b, R[2].b)));
        // weights: index = (x+1) + 3 * (y+1) (temporalFilter.comp resolve loop, y outer)
        auto wsum = [&](float l0, float c0, float r0, float l1, float c1, float r1, float l2, float c2, float r2) {
            return l0 * rw.w[0] + c0 * rw.w[1] + r0 * rw.w[2] + l1 * rw.w[3] + c1 * rw.w[4] + r1 * rw.w[5] + l2 * rw.w[6] + c2 * rw.w[7] + r2 * rw.w[8];
        };
        vec3 currentColor(wsum(L[0].r, C[0].r, R[0].r, L[1].r, C[1].r, R[1].r, L[2].r, C[2].r, R[2].r),
                          wsum(L[0].g, C[0].g, R[0].g, L[1].g, C[1].g, R[1].g, L[2].g, C[2].g, R[2].g),
                          wsum(L[0].b, C[0].b, R[0].b, L[1].b, C[1].b, R[1].b, L[2].b, C[2].b, R[2].b));

        const float rpx = ((float)px + 0.5f) * tsx + mvx[j], rpy = ((float)py + 0.5f) * tsy + mvy[j];
        const int i0 = fi[j], j0 = fj[j];
        const float a = wa[j], b = wb[j];
        struct { float x, y; } m = {mvx[j], mvy[j]};

        // ---- history: the 4x4 texel footprint of the nine bilinear neighbourhood taps around uv + motion: luminances of all sixteen, colours of
        // the centre 2x2 (= the footprint of the bilinear tap at uv + motion)
        float tl[4][4];
        vec3 tc[2][2]; // colours of the centre 2x2 = footprint of the bilinear tap at uv + motion
        if (staged) {
            // lanes that hold no output pixel (strip borders, beyond the image) may point anywhere: they read the box's first rows
            const int base = isOutputLane ? (j0 - 1 - orgJ) * stageW + (i0 - 1 - orgI) : 0;
#pragma unroll
            for (int r = 0; r < 4; r++)
#pragma unroll
                for (int c = 0; c < 4; c++) {
                    const int e = base + r * stageW + c;
                    if (r >= 1 && r <= 2 && c >= 1 && c <= 2) {
                        const uint2 v = myStage[e];
                        tl[r][c] = u2f(v.x);
                        tc[r - 1][c - 1] = unpackR11G11B10(v.y);
                    } else tl[r][c] = u2f(myStage[e].x);
                }
        } else {
        uint32_t t[4][4]; // [row][col] = texel (i0 - 1 + col, j0 - 1 + row), clamped to the edge
        const bool interior = i0 >= 1 && i0 + 2 < hw;
        if (__builtin_amdgcn_ballot_w64(!interior) == 0ull) {
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const uint4 v = fetch128(hist, fastm::texelIndex((uint32_t)(i0 - 1), (uint32_t)clampi(j0 - 1 + r, hh), (uint32_t)hw)); // four texels of a row (16 bytes from a 4-byte element)
                t[r][0] = v.x; t[r][1] = v.y; t[r][2] = v.z; t[r][3] = v.w;
            }
        } else {
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const uint32_t row = __umul24((uint32_t)clampi(j0 - 1 + r, hh), (uint32_t)hw);
#pragma unroll
                for (int c = 0; c < 4; c++) t[r][c] = fetch32(hist, row + (uint32_t)clampi(i0 - 1 + c, hw));
            }
        }
#pragma unroll
        for (int r = 0; r < 4; r++)
#pragma unroll
            for (int c = 0; c < 4; c++) {
                const vec3 col = unpackR11G11B10(t[r][c]);
                tl[r][c] = lum(col);
                if (r >= 1 && r <= 2 && c >= 1 && c <= 2) tc[r - 1][c - 1] = col;
            }
        }

        vec3 historySample;
        {
            const vec3 top = tc[0][0] + (tc[0][1] - tc[0][0]) * a, bot = tc[1][0] + (tc[1][1] - tc[1][0]) * a;
            const vec3 bilinear = top + (bot - top) * b; // = historyTap(historySrc, rpx, rpy)
            if (TECH == 0) historySample = bilinear;
            else {
                // Bicubic1Tap (bicubicSampling.inc:147-181): one bilinear tap at the bicubic-adjusted position plus the current frame's
                // neighbourhood differences
                const float ix = (float)px + 0.5f + m.x * resX, iy = (float)py + 0.5f + m.y * resY;
                const float tx = floorf(ix - 0.5f) + 0.5f, ty = floorf(iy - 0.5f) + 0.5f;
                const float fx = ix - tx, fy = iy - ty;
                const float fx2 = fx * fx, fx3 = fx2 * fx, fy2 = fy * fy, fy3 = fy2 * fy;
                const float w0x = -0.5f * fx3 + fx2 - 0.5f * fx, w1x = 1.5f * fx3 - 2.5f * fx2 + 1.f, w2x = -1.5f * fx3 + 2.f * fx2 + 0.5f * fx, w3x = 0.5f * fx3 - 0.5f * fx2;
                const float w0y = -0.5f * fy3 + fy2 - 0.5f * fy, w1y = 1.5f * fy3 - 2.5f * fy2 + 1.f, w2y = -1.5f * fy3 + 2.f * fy2 + 0.5f * fy, w3y = 0.5f * fy3 - 0.5f * fy2;
                const float wBx = w1x + w2x, wBy = w1y + w2y;
                const float uT = (tx + w2x * rcpf(wBx)) * tsx, vT = (ty + w2y * rcpf(wBy)) * tsy;
                int i1, j1; float a1, b1;
                linearCoord(uT * (float)hw, &i1, &a1);
                linearCoord(vT * (float)hh, &j1, &b1);
                vec3 h;
                if (i1 == i0 && j1 == j0) {
                    const vec3 tp = tc[0][0] + (tc[0][1] - tc[0][0]) * a1, bt = tc[1][0] + (tc[1][1] - tc[1][0]) * a1;
                    h = tp + (bt - tp) * b1;
                } else h = historyTap(historySrc, uT, vT); // the adjusted tap fell into a neighbouring texel quad (rounding at a texel border)
                const float wa = w0x * wBy, wb = wBx * w0y, wc = wBx * wBy, wd = wBx * w3y, we = w3x * wBy;
                const vec3 cC(C[1].r, C[1].g, C[1].b);
                const vec3 acc = (vec3(L[1].r, L[1].g, L[1].b) - cC) * wa + (vec3(C[0].r, C[0].g, C[0].b) - cC) * wb + (vec3(C[2].r, C[2].g, C[2].b) - cC) * wd +
                                 (vec3(R[1].r, R[1].g, R[1].b) - cC) * we;
                historySample = h + acc * rcpf(wa + wb + wc + wd + we);
            }
        }
        if (TONEMAP) historySample = tonemapF(historySample);
        if (CLIP) historySample = clipAABB(historySample, mn, mx);
        else historySample = vec3(__builtin_fminf(__builtin_fmaxf(historySample.x, mn.x), mx.x), __builtin_fminf(__builtin_fmaxf(historySample.y, mn.y), mx.y),
                                  __builtin_fminf(__builtin_fmaxf(historySample.z, mn.z), mx.z));
        if (anyNan(historySample)) historySample = currentColor;

        const float cc = C[1].l;
        const float currentContrast = fabsf(L[0].l - cc) + fabsf(C[0].l - cc) + fabsf(R[0].l - cc) + fabsf(L[2].l - cc) + fabsf(C[2].l - cc) + fabsf(R[2].l - cc) +
                                      fabsf(L[1].l - cc) + fabsf(R[1].l - cc);
        float lastContrast;
        {
            float hl[3][3]; // [x+1][y+1]
#pragma unroll
            for (int y = 0; y < 3; y++) {
                float rowLerp[4];
#pragma unroll
                for (int c = 0; c < 4; c++) rowLerp[c] = tl[y][c] + (tl[y + 1][c] - tl[y][c]) * b;
#pragma unroll
                for (int x = 0; x < 3; x++) {
                    const float l = rowLerp[x] + (rowLerp[x + 1] - rowLerp[x]) * a;
                    hl[x][y] = TONEMAP ? l * rcpf(1.f + l) : l;
                }
            }
            const float hc = hl[1][1];
            lastContrast = fabsf(hl[0][0] - hc) + fabsf(hl[1][0] - hc) + fabsf(hl[2][0] - hc) + fabsf(hl[0][2] - hc) + fabsf(hl[1][2] - hc) + fabsf(hl[2][2] - hc) +
                           fabsf(hl[0][1] - hc) + fabsf(hl[2][1] - hc);
        }
        const float contrastChange = __builtin_amdgcn_fmed3f(fabsf(currentContrast - lastContrast), 0.f, 1.f);
        float blendFactor = 0.13f + (0.03f - 0.13f) * contrastChange;
        if (cameraCut) blendFactor = 1.f;
        if (rpx < 0.f || rpy < 0.f || rpx > 1.f || rpy > 1.f) {
            blendFactor = 1.f;
            currentColor = (vec3(L[0].r, L[0].g, L[0].b) + vec3(L[2].r, L[2].g, L[2].b) + vec3(R[0].r, R[0].g, R[0].b) + vec3(R[2].r, R[2].g, R[2].b)) * 0.0625f +
                           (vec3(C[0].r, C[0].g, C[0].b) + vec3(L[1].r, L[1].g, L[1].b) + vec3(C[2].r, C[2].g, C[2].b) + vec3(R[1].r, R[1].g, R[1].b)) * 0.125f +
                           vec3(C[1].r, C[1].g, C[1].b) * 0.25f;
        }
        vec3 color = historySample + (currentColor - historySample) * blendFactor;
        if (TONEMAP) color = tonemapReverseF(color);
        const uint32_t packed = packR11G11B10(color);
        if (isOutputLane) {
            const bool through = BANDED && ranges.isEdge((int)blockIdx.y); // rows a neighbouring GPU is waiting for: written through (backend.h TwoRanges)
            if (px < historyDst.w && py < historyDst.h) storeOut((uint32_t*)historyDst.ptr + fastm::texelIndex((uint32_t)px, (uint32_t)py, (uint32_t)historyDst.w), packed, through);
            storeOut((uint32_t*)output.ptr + fastm::texelIndex((uint32_t)px, (uint32_t)py, (uint32_t)output.w), packed, through);
        }
        // slide the window down one row
        C[0] = C[1]; L[0] = L[1]; R[0] = R[1];
        C[1] = C[2]; L[1] = L[2]; R[1] = R[2];
    }
    if (BANDED) ranges.edgeDone((int)blockIdx.y);
}

typedef void (*TaaKernel)(ImgView, ImgView, ImgView, ImgView, ImgView, ImgView, const ResolveWeights*, const GlobalUbo*, int, int, int, int);
typedef void (*TaaStripKernel)(ImgView, ImgView, ImgView, ImgView, ImgView, ImgView, const ResolveWeights*, const GlobalUbo*, int, int, int, int, TwoRanges);
template <bool CLIP, bool DILATE, bool TONEMAP> static TaaKernel pickTech(int tech) {
    switch (tech) {
        case 0: return temporalFilterFastKernel<CLIP, DILATE, 0, TONEMAP>;
        case 1: return temporalFilterFastKernel<CLIP, DILATE, 1, TONEMAP>;
        case 2: return temporalFilterFastKernel<CLIP, DILATE, 2, TONEMAP>;
        case 3: return temporalFilterFastKernel<CLIP, DILATE, 3, TONEMAP>;
        case 4: return temporalFilterFastKernel<CLIP, DILATE, 4, TONEMAP>;
        default: return nullptr;
    }
}

static int launch(const PassCtx& c) {
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
    TaaStripKernel strip = nullptr;
    if (tech == 0 || tech == 4) {
#define PLR_STRIP(T, B) (clip ? (dilate ? (tonemap ? temporalFilterStripKernel<true, true, T, true, B> : temporalFilterStripKernel<true, true, T, false, B>)   \
                                        : (tonemap ? temporalFilterStripKernel<true, false, T, true, B> : temporalFilterStripKernel<true, false, T, false, B>)) \
                              : (dilate ? (tonemap ? temporalFilterStripKernel<false, true, T, true, B> : temporalFilterStripKernel<false, true, T, false, B>)  \
                                        : (tonemap ? temporalFilterStripKernel<false, false, T, true, B> : temporalFilterStripKernel<false, false, T, false, B>)))
        const bool banded = c.extraCountY != 0 || c.firstRows[0] != 0 || c.firstRows[1] != 0 || c.firstCols[0] != 0 || c.firstCols[1] != 0;
        strip = banded ? (tech == 0 ? PLR_STRIP(0, true) : PLR_STRIP(4, true)) : (tech == 0 ? PLR_STRIP(0, false) : PLR_STRIP(4, false));
#undef PLR_STRIP
    }
    const ImgView& out = c.storage[1];
    const PassCtx::ColSpan cs = c.colSpan(std::min(out.w, c.sampled[0].w));
    const int w = cs.x1, x0 = cs.x0; // columns [x0, w)
    // rows [y0, h), in blocks of 16 (4 waves x kStripRows); a second row range in the same launch (pass fusion of band rendering's two edge dispatches)
    TwoRanges ranges;
    int stripBlocks, y0, h;
    const bool expressible = twoRangeBlocks(c, std::min(out.h, c.sampled[0].h), 4 * kStripRows, 8, &ranges, &stripBlocks, &y0, &h) == 0;
    if (w <= x0 || h <= y0) return 0;
    // the strip kernel indexes all per-pixel images with one coordinate: it needs them to be the same size
    const bool sameSize = c.sampled[0].w == out.w && c.sampled[0].h == out.h && c.sampled[3].w == out.w && c.sampled[3].h == out.h && c.sampled[5].w == out.w &&
                          c.sampled[5].h == out.h && out.w >= 4;
    if (c.extraCountY && !(strip && sameSize && expressible)) return kUseGeneralKernel; // two ranges: strip kernel only (else two launches)
    const unsigned stripsX = divUp((unsigned)(w - x0), (unsigned)kStripW);
    if (strip && sameSize && !c.extraCountY) ranges.setEdgeFirst(c, y0, h, 4 * kStripRows, 8, stripsX, x0, w, kStripW); // band / tile, edges first (plr.h first_rows, first_cols)
    if (strip && sameSize)
        strip<<<dim3(stripsX, (unsigned)stripBlocks), 256, 0, c.stream>>>(
            c.sampled[0], out, c.storage[2], c.sampled[3], c.sampled[4], c.sampled[5], (const ResolveWeights*)c.ubuf[6].ptr, c.global, w, h, y0, x0, ranges);
    else
        k<<<dim3(divUp((unsigned)(w - x0), 64u), divUp((unsigned)(h - y0), 4u)), 256, 0, c.stream>>>(c.sampled[0], out, c.storage[2], c.sampled[3], c.sampled[4], c.sampled[5],
                                                                                              (const ResolveWeights*)c.ubuf[6].ptr, c.global, w, h, y0, x0);
    PLR_CHECK_LAUNCH(c);
    return 0;
}

} // namespace fasttaa

static int fasttaa_launch(const PassCtx& c) { return fasttaa::launch(c); }
PLR_REGISTER_SHADER_FAST("temporalFilter.comp", fasttaa_launch);
static int fasttaa_two_ranges(const PassCtx* const* ctxs, size_t count) { return launchOverTwoRowRanges(ctxs, count, fasttaa_launch); }
PLR_REGISTER_FUSION("temporalFilter over two row ranges", fasttaa_two_ranges, "temporalFilter.comp", "temporalFilter.comp");
} // namespace plr
