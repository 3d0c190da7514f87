// PLR_MATH_FAST variants of the streaming passes: applyBloom.comp, tonemapping.comp, indirectLightUpscale.comp,
// filterIndirectDiffuseTemporal.comp (exact variants: kernels_exact/bloom.hip, kernels/exposure_tonemap.hip, kernels_exact/gi_filters.hip).
//
// Every pass of this pipeline is bound by VALU issue on gfx950 (a wave64 FP32 instruction occupies a SIMD for 4 cycles; PMC:
// SQ_INSTS_VALU x 4 cycles / 4 SIMDs accounts for each kernel's time), not by HBM. These variants keep the algorithms and cut the
// instruction count: v_rcp instead of IEEE division sequences, hardware min/max/med3 instead of the NaN-ordered software forms,
// texel-centre taps as plain fetches, values that are constant along a row or shared by the four pixels of a lane computed once,
// and adjacent texels fetched with one wide load (a load instruction costs a CU's texture addresser ~16 cycles whatever its width).
#include "../backend.h"
#include "../device/shading_common.h"
#include "fused_gi.h"
#include "upscale_quad.h"
#include "../device/fastmath.h"

namespace plr {
namespace faststream {

PLR_DI float rcpf(float x) { return __builtin_amdgcn_rcpf(x); }
PLR_DI float clamp01(float x) { return __builtin_amdgcn_fmed3f(x, 0.f, 1.f); }

// ------------------------------------------------------------------------------------------------ applyBloom.comp:16-30
// target = mix(scene, bloom, strength) in place. The bloom image has the target's size, so the bilinear tap at the pixel centre
// is the texel itself. Four pixels (16 B) per lane.
__global__ __launch_bounds__(256) void applyBloomFastKernel(ImgView target, ImgView bloom, float strength, int coverW, int coverH, int yBase, int xBase) {
    const int x0 = xBase + (int)(blockIdx.x * 64u + (threadIdx.x & 63u)) * 4; // columns [xBase, coverW), xBase a multiple of 8 (PassCtx::colSpan)
    const int y = yBase + (int)(blockIdx.y * 4u + (threadIdx.x >> 6));
    if (y >= coverH || x0 >= coverW) return;
    uint32_t* trow = (uint32_t*)target.ptr + (size_t)y * (size_t)target.w;
    const uint32_t* brow = (const uint32_t*)bloom.ptr + (size_t)y * (size_t)bloom.w;
    const int n = min(4, coverW - x0);
    uint32_t s[4] = {0u, 0u, 0u, 0u}, b[4] = {0u, 0u, 0u, 0u};
    const bool wide = n == 4 && (target.w & 3) == 0;
    if (wide) {
        const uint4 sv = *(const uint4*)(trow + x0), bv = *(const uint4*)(brow + x0);
        s[0] = sv.x; s[1] = sv.y; s[2] = sv.z; s[3] = sv.w; b[0] = bv.x; b[1] = bv.y; b[2] = bv.z; b[3] = bv.w;
    } else for (int i = 0; i < n; i++) { s[i] = trow[x0 + i]; b[i] = brow[x0 + i]; }
    uint32_t o[4];
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const vec3 sc = unpackR11G11B10(s[i]), bc = unpackR11G11B10(b[i]);
        o[i] = packR11G11B10(sc + (bc - sc) * strength);
    }
    if (wide) *(uint4*)(trow + x0) = make_uint4(o[0], o[1], o[2], o[3]);
    else for (int i = 0; i < n; i++) trow[x0 + i] = o[i];
}

static int launchApplyBloom(const PassCtx& c) {
    if (int rc = c.needStorage(0, F_R11G11B10, "applyBloom target")) return rc;
    if (int rc = c.needSampled(1, F_R11G11B10, "applyBloom bloomTexture")) return rc;
    if (c.push.size() < 4) return c.fail(-1, "applyBloom: push constant bloomStrength missing");
    const ImgView& target = c.storage[0];
    if (c.sampled[1].w != target.w || c.sampled[1].h != target.h) return kUseGeneralKernel; // a real bilinear tap: general kernel
    float strength;
    std::memcpy(&strength, c.push.data(), 4);
    const PassCtx::RowSpan rs = c.rowSpan(target.h);
    const PassCtx::ColSpan cs = c.colSpan(target.w);
    const int w = cs.x1, x0 = cs.x0, h = rs.y1, y0 = rs.y0;
    if (w <= x0 || h <= y0) return 0;
    applyBloomFastKernel<<<dim3(divUp((unsigned)(w - x0), 256u), divUp((unsigned)(h - y0), 4u)), 256, 0, c.stream>>>(target, c.sampled[1], strength, w, h, y0, x0);
    PLR_CHECK_LAUNCH(c);
    return 0;
}

// ------------------------------------------------------------------------------------------------ tonemapping.comp:17-27
PLR_DI float srgb(float l) {
    const float hi = __builtin_amdgcn_exp2f((1.0f / 2.4f) * __builtin_amdgcn_logf(fabsf(l))) * 1.055f - 0.055f;
    return l <= 0.0031308f ? l * 12.92f : hi;
}
PLR_DI uint32_t unorm8(float v) { return (uint32_t)__float2int_rn(clamp01(v) * 255.0f); } // the inputs here are never NaN after the clamps of ACESFitted

// one pixel of tonemapping.comp: ACESFitted, dither, sRGB, 8-bit pack. nyA / nyB: the row terms of the two dither hashes
template <bool BGRA>
PLR_DI uint32_t tonemapPixel(vec3 c, int x, float time, uint32_t nyA, uint32_t nyB) {
    const uint32_t UI0 = 1597334673u, UI1 = 3812015801u, UI2 = 2798796415u;
    const float UIF = 1.0f / (float)0xffffffffu;
    // ACESFitted (tonemapping.inc:40-49)
    vec3 v(0.59719f * c.x + 0.35458f * c.y + 0.04823f * c.z, 0.07600f * c.x + 0.90834f * c.y + 0.01566f * c.z, 0.02840f * c.x + 0.13383f * c.y + 0.83777f * c.z);
    const vec3 a = v * (v + 0.0245786f) - 0.000090537f;
    const vec3 b = v * (0.983729f * v + 0.4329510f) + 0.238081f;
    v = vec3(a.x * rcpf(b.x), a.y * rcpf(b.y), a.z * rcpf(b.z));
    const vec3 o(clamp01(1.60475f * v.x + -0.53108f * v.y + -0.07367f * v.z), clamp01(-0.10208f * v.x + 1.10813f * v.y + -0.00605f * v.z),
                 clamp01(-0.00327f * v.x + -0.07276f * v.y + 1.07602f * v.z));
    const uint32_t qxA = (uint32_t)(int32_t)(float)(uint32_t)((float)x * time), qxB = (uint32_t)(int32_t)(float)(uint32_t)(((float)x + 165.f) * time);
    const uint32_t mA = (qxA * UI0) ^ nyA ^ (qxA * UI2), mB = (qxB * UI0) ^ nyB ^ (qxB * UI2);
    const vec3 noise = (vec3((float)(mA * UI0), (float)(mA * UI1), (float)(mA * UI2)) + vec3((float)(mB * UI0), (float)(mB * UI1), (float)(mB * UI2))) * UIF - 1.f;
    const vec3 s = vec3(srgb(o.x), srgb(o.y), srgb(o.z)) + noise * (1.f / 255.f);
    uint32_t p = unorm8(s.z) | (unorm8(s.y) << 8) | (unorm8(s.x) << 16) | (255u << 24);
    if (!BGRA) p = (p & 0xff00ff00u) | ((p >> 16) & 0xffu) | ((p & 0xffu) << 16);
    return p;
}
// dither.inc:6-12 / noise.inc:19-24: hash32(q) = fract-free integer hash of uvec2 q; the y terms are shared by all pixels of a row
PLR_DI void tonemapRowTerms(int y, float time, uint32_t* nyA, uint32_t* nyB) {
    const uint32_t UI1 = 3812015801u;
    const uint32_t qyA = (uint32_t)(int32_t)(float)(uint32_t)((float)y * time), qyB = (uint32_t)(int32_t)(float)(uint32_t)(((float)y + 1292.f) * time);
    *nyA = qyA * UI1; *nyB = qyB * UI1;
}

template <bool BGRA>
__global__ __launch_bounds__(256) void tonemappingFastKernel(ImgView src, ImgView dst, const GlobalUbo* __restrict__ g, int coverW, int coverH, int yBase, int xBase) {
    const int x0 = xBase + (int)(blockIdx.x * 64u + (threadIdx.x & 63u)) * 4;
    const int y = yBase + (int)(blockIdx.y * 4u + (threadIdx.x >> 6));
    if (y >= coverH || x0 >= coverW) return;
    const float time = g->time;
    const uint32_t* srow = (const uint32_t*)src.ptr + (size_t)y * (size_t)src.w;
    uint32_t* drow = (uint32_t*)dst.ptr + (size_t)y * (size_t)dst.w;
    const int n = min(4, coverW - x0);
    uint32_t in[4] = {0u, 0u, 0u, 0u}, out[4];
    const bool wide = n == 4 && (src.w & 3) == 0 && (dst.w & 3) == 0;
    if (wide) { const uint4 v = *(const uint4*)(srow + x0); in[0] = v.x; in[1] = v.y; in[2] = v.z; in[3] = v.w; }
    else for (int i = 0; i < n; i++) in[i] = srow[x0 + i];
    uint32_t nyA, nyB;
    tonemapRowTerms(y, time, &nyA, &nyB);
#pragma unroll
    for (int i = 0; i < 4; i++) out[i] = tonemapPixel<BGRA>(unpackR11G11B10(in[i]), x0 + i, time, nyA, nyB);
    if (wide) *(uint4*)(drow + x0) = make_uint4(out[0], out[1], out[2], out[3]);
    else for (int i = 0; i < n; i++) drow[x0 + i] = out[i];
}

// ---- applyBloom + tonemapping of the image it just wrote, in one pass over the pixels (pass fusion, backend.h): the applied colour is
// stored (R11G11B10, the pass's output) and its STORED value - decoded again, as the separate tonemap pass would read it - is tonemapped.
// 16 B per pixel of traffic instead of 20, one launch instead of two.
template <bool BGRA>
__global__ __launch_bounds__(256) void applyBloomTonemapKernel(ImgView target, ImgView bloom, ImgView dst, const GlobalUbo* __restrict__ g, float strength, int coverW, int coverH, int yBase, int xBase) {
    const int x0 = xBase + (int)(blockIdx.x * 64u + (threadIdx.x & 63u)) * 4;
    const int y = yBase + (int)(blockIdx.y * 4u + (threadIdx.x >> 6));
    if (y >= coverH || x0 >= coverW) return;
    const float time = g->time;
    uint32_t* trow = (uint32_t*)target.ptr + (size_t)y * (size_t)target.w;
    const uint32_t* brow = (const uint32_t*)bloom.ptr + (size_t)y * (size_t)bloom.w;
    uint32_t* drow = (uint32_t*)dst.ptr + (size_t)y * (size_t)dst.w;
    const uint4 sv = *(const uint4*)(trow + x0), bv = *(const uint4*)(brow + x0); // the launcher guarantees whole 4-pixel groups and 16-byte aligned rows
    const uint32_t s[4] = {sv.x, sv.y, sv.z, sv.w}, b[4] = {bv.x, bv.y, bv.z, bv.w};
    uint32_t nyA, nyB;
    tonemapRowTerms(y, time, &nyA, &nyB);
    uint32_t o[4], t[4];
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const vec3 sc = unpackR11G11B10(s[i]), bc = unpackR11G11B10(b[i]);
        o[i] = packR11G11B10(sc + (bc - sc) * strength);
        t[i] = tonemapPixel<BGRA>(unpackR11G11B10(o[i]), x0 + i, time, nyA, nyB);
    }
    *(uint4*)(trow + x0) = make_uint4(o[0], o[1], o[2], o[3]);
    *(uint4*)(drow + x0) = make_uint4(t[0], t[1], t[2], t[3]);
}

static int launchTonemapping(const PassCtx& c) {
    if (int rc = c.needGlobal()) return rc;
    if (int rc = c.needSampled(1, F_R11G11B10, "tonemapping imageIn")) return rc;
    if (int rc = c.needStorage(0, -1, "tonemapping imageOut")) return rc;
    const ImgView& src = c.sampled[1];
    const ImgView& dst = c.storage[0];
    if (dst.fmt != F_BGRA8 && dst.fmt != F_RGBA8) return c.fail(-4, "tonemapping imageOut must be BGRA8_uNorm or RGBA8");
    const PassCtx::ColSpan cs = c.colSpan(std::min(dst.w, src.w));
    const int coverW = cs.x1, xBase = cs.x0;
    const PassCtx::RowSpan rs = c.rowSpan(std::min(dst.h, src.h));
    const int coverH = rs.y1, y0 = rs.y0;
    if (coverW <= xBase || coverH <= y0) return 0;
    const dim3 grid(divUp((unsigned)(coverW - xBase), 256u), divUp((unsigned)(coverH - y0), 4u));
    if (dst.fmt == F_BGRA8) tonemappingFastKernel<true><<<grid, 256, 0, c.stream>>>(src, dst, c.global, coverW, coverH, y0, xBase);
    else tonemappingFastKernel<false><<<grid, 256, 0, c.stream>>>(src, dst, c.global, coverW, coverH, y0, xBase);
    PLR_CHECK_LAUNCH(c);
    return 0;
}

// ------------------------------------------------------------------------------------------------ indirectLightUpscale.comp:17-71
// bilinear footprint of an RGBA16F / RG16F / R16F image at (i0..i0+1, j0..j0+1) with clamp-to-edge, the two texels of a row in one load
struct Pair8 { uint2 a, b; };
PLR_DI Pair8 loadPairRGBA16F(const ImgView& im, int x0, int x1, int y) {
    const uint2* row = (const uint2*)im.ptr + (size_t)y * (size_t)im.w;
    Pair8 p;
    if (im.w >= 2) {
        const int xb = min(x0, im.w - 2);
        uint4 v;
        __builtin_memcpy(&v, row + xb, 16);
        const uint2 lo = make_uint2(v.x, v.y), hi = make_uint2(v.z, v.w);
        p.a = x0 == xb ? lo : hi;
        p.b = x1 == xb ? lo : hi;
    } else { p.a = row[x0]; p.b = row[x1]; }
    return p;
}
PLR_DI void loadPairRG16F(const ImgView& im, int x0, int x1, int y, uint32_t* a, uint32_t* b) {
    const uint32_t* row = (const uint32_t*)im.ptr + (size_t)y * (size_t)im.w;
    if (im.w >= 2) {
        const int xb = min(x0, im.w - 2);
        uint2 v;
        __builtin_memcpy(&v, row + xb, 8);
        *a = x0 == xb ? v.x : v.y;
        *b = x1 == xb ? v.x : v.y;
    } else { *a = row[x0]; *b = row[x1]; }
}
using fastquad::halves2;
using fastquad::halves4;
using fastquad::linearDepthRounded;
using fastquad::UpscaledQuad;
using fastquad::upscaleQuad;

struct Bilinear { int x0, x1, y0, y1; float a, b; };
PLR_DI Bilinear bilinearCoords(const ImgView& im, float u, float v) {
    Bilinear q;
    int i0, j0;
    linearCoord(u * (float)im.w, &i0, &q.a);
    linearCoord(v * (float)im.h, &j0, &q.b);
    q.x0 = clampi(i0, im.w); q.x1 = clampi(i0 + 1, im.w); q.y0 = clampi(j0, im.h); q.y1 = clampi(j0 + 1, im.h);
    return q;
}
PLR_DI vec4 bilinearRGBA16F(const ImgView& im, const Bilinear& q) {
    const Pair8 r0 = loadPairRGBA16F(im, q.x0, q.x1, q.y0), r1 = loadPairRGBA16F(im, q.x0, q.x1, q.y1);
    const vec4 t00 = halves4(r0.a), t10 = halves4(r0.b), t01 = halves4(r1.a), t11 = halves4(r1.b);
    const float w00 = (1.f - q.a) * (1.f - q.b), w10 = q.a * (1.f - q.b), w01 = (1.f - q.a) * q.b, w11 = q.a * q.b;
    return t00 * w00 + t10 * w10 + t01 * w01 + t11 * w11;
}
PLR_DI vec2 bilinearRG16F(const ImgView& im, const Bilinear& q) {
    uint32_t a0, b0, a1, b1;
    loadPairRG16F(im, q.x0, q.x1, q.y0, &a0, &b0);
    loadPairRG16F(im, q.x0, q.x1, q.y1, &a1, &b1);
    const float w00 = (1.f - q.a) * (1.f - q.b), w10 = q.a * (1.f - q.b), w01 = (1.f - q.a) * q.b, w11 = q.a * q.b;
    return halves2(a0) * w00 + halves2(b0) * w10 + halves2(a1) * w01 + halves2(b1) * w11;
}

PLR_DI void upscalePixel(const ImgView& dstYSH, const ImgView& dstCoCg, const ImgView& srcYSH, const ImgView& srcCoCg, const ImgView& fullResDepthT,
                         const ImgView& halfResDepthT, const GlobalUbo* __restrict__ g, int px, int py, uint32_t* __restrict__ sig) {
    // no fused multiply-adds in this pass (-ffp-contract=fast-honor-pragmas): see linearDepthRounded; "linear depth - full-res depth" would
    // also be fused for some of the four candidates and not for others, which breaks exact ties (equal half-res depths) arbitrarily
#pragma clang fp contract(off)
    const float u = ((float)px + 0.5f) * rcpf((float)g->screenResolution[0]), v = ((float)py + 0.5f) * rcpf((float)g->screenResolution[1]);
    const float nearP = g->nearPlane, farP = g->farPlane, nf = nearP * farP, nmf = nearP - farP;
    auto linearize = [&](float d) { return linearDepthRounded(d, nf, nmf, farP); };
    const int fx = min(max((int)floorf(u * (float)fullResDepthT.w), 0), fullResDepthT.w - 1), fy = min(max((int)floorf(v * (float)fullResDepthT.h), 0), fullResDepthT.h - 1);
    const float fullResDepth = linearize(((const float*)fullResDepthT.ptr)[(size_t)fy * (size_t)fullResDepthT.w + fx]);
    // textureGather footprint of the half-res depth: (i0,j1), (i1,j1), (i1,j0), (i0,j0); a row's two texels in one 4-byte load
    const Bilinear q = bilinearCoords(halfResDepthT, u, v);
    float d00, d10, d01, d11;
    {
        const uint16_t* hd = (const uint16_t*)halfResDepthT.ptr;
        auto pair = [&](int y, float* a, float* b) {
            const uint16_t* row = hd + (size_t)y * (size_t)halfResDepthT.w;
            if (halfResDepthT.w >= 2) {
                const int xb = min(q.x0, halfResDepthT.w - 2);
                uint32_t w2;
                __builtin_memcpy(&w2, row + xb, 4);
                const float lo = halfBitsToFloat(w2 & 0xffffu), hi = halfBitsToFloat(w2 >> 16);
                *a = q.x0 == xb ? lo : hi;
                *b = q.x1 == xb ? lo : hi;
            } else { *a = halfBitsToFloat(row[q.x0]); *b = halfBitsToFloat(row[q.x1]); }
        };
        pair(q.y0, &d00, &d10);
        pair(q.y1, &d01, &d11);
    }
    const float depthSamples[4] = {d01, d11, d10, d00};
    const float offx[4] = {0.f, 1.f, 1.f, 0.f}, offy[4] = {1.f, 1.f, 0.f, 0.f};
    float minDepthDiff = 1000.f, cx = 0.f, cy = 0.f;
    bool isEdge = false;
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const float depthDiff = fabsf(linearize(depthSamples[i]) - fullResDepth);
        isEdge = isEdge || depthDiff > 0.5f;
        if (depthDiff < minDepthDiff) { minDepthDiff = depthDiff; cx = offx[i]; cy = offy[i]; }
    }
    vec4 ysh;
    vec2 cc;
    if (isEdge) {
        // nearest texel at uv + offset * halfResTexelSize (:60-64); this can be a texel outside the gather footprint
        const float uc = u + cx * rcpf((float)halfResDepthT.w), vc = v + cy * rcpf((float)halfResDepthT.h);
        const int nx = min(max((int)floorf(uc * (float)srcYSH.w), 0), srcYSH.w - 1), ny = min(max((int)floorf(vc * (float)srcYSH.h), 0), srcYSH.h - 1);
        ysh = halves4(((const uint2*)srcYSH.ptr)[(size_t)ny * (size_t)srcYSH.w + nx]);
        const int mx = min(max((int)floorf(uc * (float)srcCoCg.w), 0), srcCoCg.w - 1), my = min(max((int)floorf(vc * (float)srcCoCg.h), 0), srcCoCg.h - 1);
        cc = halves2(((const uint32_t*)srcCoCg.ptr)[(size_t)my * (size_t)srcCoCg.w + mx]);
    } else {
        ysh = bilinearRGBA16F(srcYSH, bilinearCoords(srcYSH, u, v));
        cc = bilinearRG16F(srcCoCg, bilinearCoords(srcCoCg, u, v));
    }
    const size_t idx = (size_t)py * (size_t)dstYSH.w + px;
    Texel<F_RGBA16F>::store(dstYSH.ptr, idx, ysh);
    Texel<F_RG16F>::store(dstCoCg.ptr, idx, vec4(cc.x, cc.y, 0.f, 0.f));
    if (sig) sig[idx] = (isEdge ? 1u : 0u) | (cx != 0.f ? 2u : 0u) | (cy != 0.f ? 4u : 0u); // decision signature (oracle/oracle.h)
}

__global__ __launch_bounds__(256) void indirectLightUpscaleFastKernel(ImgView dstYSH, ImgView dstCoCg, ImgView srcYSH, ImgView srcCoCg, ImgView fullResDepthT,
                                                                      ImgView halfResDepthT, const GlobalUbo* __restrict__ g, int coverW, int coverH, int yBase, int xBase, uint32_t* __restrict__ sig) {
    const int px = xBase + (int)(blockIdx.x * 64u + (threadIdx.x & 63u)); // columns [xBase, coverW) (PassCtx::colSpan)
    const int py = yBase + (int)(blockIdx.y * 4u + (threadIdx.x >> 6));
    if (px >= coverW || py >= coverH) return;
    upscalePixel(dstYSH, dstCoCg, srcYSH, srcCoCg, fullResDepthT, halfResDepthT, g, px, py, sig);
}

// ---- 2x2 outputs per thread when the full-resolution image is exactly twice the half-resolution one: upscaleQuad (upscale_quad.h, shared with
// the fused upscale + deferred shade of shading_fast.hip)
__global__ __launch_bounds__(256) void indirectLightUpscaleQuadKernel(ImgView dstYSH, ImgView dstCoCg, ImgView srcYSH, ImgView srcCoCg, ImgView fullResDepthT,
                                                                      ImgView halfResDepthT, const GlobalUbo* __restrict__ g, int coverW, int coverH, int yBase, int xBase, uint32_t* __restrict__ sig) {
    const int k = (xBase >> 1) + (int)(blockIdx.x * 64u + (threadIdx.x & 63u)); // xBase: a multiple of 8 (PassCtx::colSpan)
    const int m = (yBase >> 1) + (int)(blockIdx.y * 4u + (threadIdx.x >> 6));
    const int X = 2 * k, Y = 2 * m;
    if (X >= coverW || Y >= coverH) return;
    if (g->screenResolution[0] != dstYSH.w || g->screenResolution[1] != dstYSH.h) {
        // uv is defined by the UBO's screen resolution (:19): when that is not the target size the quad reasoning does not hold
        for (int py = 0; py < 2; py++)
            for (int px = 0; px < 2; px++)
                if (X + px < coverW && Y + py < coverH) upscalePixel(dstYSH, dstCoCg, srcYSH, srcCoCg, fullResDepthT, halfResDepthT, g, X + px, Y + py, sig);
        return;
    }
    UpscaledQuad q;
    upscaleQuad(srcYSH, srcCoCg, fullResDepthT, halfResDepthT, g, k, m, &q);
#pragma unroll
    for (int py = 0; py < 2; py++) {
        if (Y + py >= coverH || Y + py < yBase) continue;
        const size_t o = (size_t)(Y + py) * (size_t)dstYSH.w + X;
        if (sig) {
            sig[o] = q.sig[py][0];
            if (X + 1 < coverW) sig[o + 1] = q.sig[py][1];
        }
        if (X + 1 < coverW) {
            *(uint4*)((uint2*)dstYSH.ptr + o) = make_uint4(q.ysh[py][0].x, q.ysh[py][0].y, q.ysh[py][1].x, q.ysh[py][1].y); // X even, even pitch: 16-byte aligned
            *(uint2*)((uint32_t*)dstCoCg.ptr + o) = make_uint2(q.cocg[py][0], q.cocg[py][1]);
        } else {
            ((uint2*)dstYSH.ptr)[o] = q.ysh[py][0];
            ((uint32_t*)dstCoCg.ptr)[o] = q.cocg[py][0];
        }
    }
}

static int launchUpscale(const PassCtx& c) {
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
    const int w = cs.x1, x0 = cs.x0, h = rs.y1, y0 = rs.y0;
    if (w <= x0 || h <= y0) return 0;
    // exact 2x case: screen = full-res target = depth buffer, half-res images exactly half of it (the sub-texel weights are then 0.25 / 0.75)
    const bool regular = out.w == 2 * c.sampled[2].w && out.h == 2 * c.sampled[2].h && c.sampled[3].w == c.sampled[2].w && c.sampled[3].h == c.sampled[2].h &&
                         c.sampled[5].w == c.sampled[2].w && c.sampled[5].h == c.sampled[2].h && c.sampled[4].w == out.w && c.sampled[4].h == out.h &&
                         c.storage[1].w == out.w && c.storage[1].h == out.h && (y0 & 1) == 0 && c.sampled[2].w >= 4;
    uint32_t* sig = c.sigFor((size_t)out.w * (size_t)out.h);
    if (regular)
        indirectLightUpscaleQuadKernel<<<dim3(divUp(divUp((unsigned)(w - x0), 2u), 64u), divUp(divUp((unsigned)(h - y0), 2u), 4u)), 256, 0, c.stream>>>(
            c.storage[0], c.storage[1], c.sampled[2], c.sampled[3], c.sampled[4], c.sampled[5], c.global, w, h, y0, x0, sig);
    else
        indirectLightUpscaleFastKernel<<<dim3(divUp((unsigned)(w - x0), 64u), divUp((unsigned)(h - y0), 4u)), 256, 0, c.stream>>>(c.storage[0], c.storage[1], c.sampled[2], c.sampled[3],
                                                                                                                         c.sampled[4], c.sampled[5], c.global, w, h, y0, x0, sig);
    PLR_CHECK_LAUNCH(c);
    return 0;
}

// ------------------------------------------------------------------------------------------------ filterIndirectDiffuseTemporal.comp:20-86
PLR_DI vec2 bilinearRG16SN(const ImgView& im, float u, float v, bool repeat) {
    int i0, j0; float a, b;
    linearCoord(u * (float)im.w, &i0, &a);
    linearCoord(v * (float)im.h, &j0, &b);
    int x0, x1, y0, y1;
    if (repeat) {
        // repeat addressing without the integer modulo (~28 instructions each for a runtime divisor, four of them per pixel: a quarter of this kernel's instructions).
        // uv + motion lies in [-1, 2] (uv in [0, 1], RG16_sNorm motion in [-1, 1]), so an index is within two image sizes of the image: two conditional steps each way
        // give exactly i mod n
        auto wrap = [](int i, int n) { i = i < 0 ? i + n : i; i = i < 0 ? i + n : i; i = i >= n ? i - n : i; i = i >= n ? i - n : i; return i; };
        x0 = wrap(i0, im.w); x1 = wrap(i0 + 1, im.w); y0 = wrap(j0, im.h); y1 = wrap(j0 + 1, im.h);
    }
    else { x0 = clampi(i0, im.w); x1 = clampi(i0 + 1, im.w); y0 = clampi(j0, im.h); y1 = clampi(j0 + 1, im.h); }
    const uint32_t* base = (const uint32_t*)im.ptr;
    auto tx = [&](int x, int y) {
        const uint32_t t = base[(size_t)y * (size_t)im.w + x];
        return vec2(fmaxf((float)(int16_t)(t & 0xffffu) * (1.f / 32767.f), -1.f), fmaxf((float)(int16_t)(t >> 16) * (1.f / 32767.f), -1.f));
    };
    const float w00 = (1.f - a) * (1.f - b), w10 = a * (1.f - b), w01 = (1.f - a) * b, w11 = a * b;
    return tx(x0, y0) * w00 + tx(x1, y0) * w10 + tx(x0, y1) * w01 + tx(x1, y1) * w11;
}

// PACK: also write the packed texel of the history output for the spatial filter that reads it next (fused_gi.h); 0 = no, else the depth format
// PACKED_ONLY: that packed texel is all anybody will read of this pass's result (PassCtx::elidableStorage on all four outputs, backend.cpp
// markElidableBehindConsumer): the four image stores (24 of the pass's 72 bytes per pixel) are left out
template <int PACK, bool PACKED_ONLY = false>
__global__ __launch_bounds__(256) void temporalGiFilterFastKernel(ImgView targetYSH, ImgView targetCoCg, ImgView historyOutYSH, ImgView historyOutCoCg, ImgView inYSH,
                                                                  ImgView inCoCg, ImgView historyInYSH, ImgView historyInCoCg, ImgView velocityCurrent,
                                                                  ImgView velocityLast, const GlobalUbo* __restrict__ g, int coverW, int coverH, int yBase, int xBase,
                                                                  uint4* __restrict__ packedOut, ImgView packDepth, TwoRanges ranges) {
    int blockCol, blockRow;
    ranges.blockXY(&blockCol, &blockRow); // a launch over two row ranges, or the edge rows (tile rendering: and columns) first (backend.h TwoRanges)
    const int px = xBase + blockCol * 64 + (int)(threadIdx.x & 63u); // columns [xBase, coverW) (PassCtx::colSpan), rows [yBase, coverH)
    const int py = yBase + blockRow * 4 + (int)(threadIdx.x >> 6);
    // (the pixel's work as a lambda: every wave, also one with nothing to do, reports in at the end - TwoRanges::edgeDone, rows-first launches of a band)
    auto pixel = [&]() {
    if (px >= coverW || py >= coverH) return;
    // texel size as the shader has it: the correctly rounded 1 / size (Newton step on v_rcp_f32). With the raw approximation a reprojected coordinate
    // lands on the other side of a 1/256 sub-texel weight step for a few pixels in 10^5, which is visible against the half-float quantum
    const float fw = (float)targetYSH.w, fh = (float)targetYSH.h;
    const float rw0 = rcpf(fw), rh0 = rcpf(fh);
    const float tsx = __builtin_fmaf(__builtin_fmaf(-fw, rw0, 1.f), rw0, rw0), tsy = __builtin_fmaf(__builtin_fmaf(-fh, rh0, 1.f), rh0, rh0);
    const float u = ((float)px + 0.5f) * tsx, v = ((float)py + 0.5f) * tsy;
    // the input images have the target's size: the tap at the pixel centre is the texel itself (weights 1,0,0,0)
    const size_t idx = (size_t)py * (size_t)targetYSH.w + px;
    const vec4 current_Y_SH = halves4(((const uint2*)inYSH.ptr)[idx]);
    const vec2 current_CoCg = halves2(((const uint32_t*)inCoCg.ptr)[idx]);
    const vec2 motion = bilinearRG16SN(velocityCurrent, u, v, false);
    const float ru = u + motion.x, rv = v + motion.y;
    vec4 history_Y_SH = bilinearRGBA16F(historyInYSH, bilinearCoords(historyInYSH, ru, rv));
    vec2 history_CoCg = bilinearRG16F(historyInCoCg, bilinearCoords(historyInCoCg, ru, rv));
    const vec2 motionLast = bilinearRG16SN(velocityLast, ru, rv, true); // sic: linearRepeat (:36)
    const float motionDifference = __builtin_amdgcn_sqrtf(fabsf(__builtin_amdgcn_sqrtf(dot(motion, motion)) - __builtin_amdgcn_sqrtf(dot(motionLast, motionLast))));
    const float motionDifferenceFactor = clamp01(motionDifference * 10.f);
    float alphaMin = 0.6f - 0.3f * fabsf(__builtin_amdgcn_sqrtf(dot(current_Y_SH, current_Y_SH)) - __builtin_amdgcn_sqrtf(dot(history_Y_SH, history_Y_SH)));
    alphaMin = fmaxf(alphaMin, 0.f);
    float alpha = 0.8f + (alphaMin - 0.8f) * motionDifferenceFactor;
    const float resX = (float)g->screenResolution[0], resY = (float)g->screenResolution[1];
    if (fmaxf(fmaxf(fabsf(motion.x), fabsf(motionLast.x)) * resX, fmaxf(fabsf(motion.y), fabsf(motionLast.y)) * resY) > 3.f) alpha = alphaMin;
    if (ru < 0.f || rv < 0.f || ru > 1.f || rv > 1.f) alpha = 0.f;
    if (g->cameraCut) alpha = 0.f;
    if (anyNan(current_Y_SH) || anyNan(current_CoCg)) {
        alpha = 1.f;
        if (anyNan(history_Y_SH)) history_Y_SH = vec4(0.f);
        if (anyNan(history_CoCg)) history_CoCg = vec2(0.f);
    }
    const vec4 result_Y_SH = current_Y_SH * (1.f - alpha) + history_Y_SH * alpha;
    const vec2 result_CoCg = current_CoCg * (1.f - alpha) + history_CoCg * alpha;
    const uint2 py4 = make_uint2(floatToHalfBits(result_Y_SH.x) | (floatToHalfBits(result_Y_SH.y) << 16), floatToHalfBits(result_Y_SH.z) | (floatToHalfBits(result_Y_SH.w) << 16));
    const uint32_t pc = floatToHalfBits(result_CoCg.x) | (floatToHalfBits(result_CoCg.y) << 16);
    const bool through = ranges.isEdge((int)blockIdx.y); // rows a neighbouring GPU is waiting for: written through (backend.h TwoRanges)
    if (!PACKED_ONLY) {
        storeOut((uint2*)targetYSH.ptr + idx, py4, through);
        storeOut((uint32_t*)targetCoCg.ptr + idx, pc, through);
        storeOut((uint2*)historyOutYSH.ptr + idx, py4, through);
        storeOut((uint32_t*)historyOutCoCg.ptr + idx, pc, through);
    }
    if (PACK) packedOut[idx] = packGiTexel(py4, pc, Texel<PACK == 0 ? F_R16F : PACK>::load(packDepth.ptr, idx).x, g->nearPlane, g->farPlane);
    };
    pixel();
    ranges.edgeDone((int)blockIdx.y);
}

static int launchTemporalGi(const PassCtx& c) {
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
    // the centre-tap shortcut and the shared store index need all trace-resolution images to have one size
    for (int b : {1, 2, 3}) if (c.storage[b].w != out.w || c.storage[b].h != out.h) return kUseGeneralKernel;
    for (int b : {4, 5, 6, 7}) if (c.sampled[b].w != out.w || c.sampled[b].h != out.h) return kUseGeneralKernel;
    // rows [y0, h) in blocks of 4; a second row range in the same launch (pass fusion of band rendering's two edge dispatches)
    TwoRanges ranges;
    int blockRows, y0, h;
    if (twoRangeBlocks(c, out.h, 4, 8, &ranges, &blockRows, &y0, &h)) return kUseGeneralKernel;
    const PassCtx::ColSpan cs = c.colSpan(out.w);
    const int w = cs.x1, x0 = cs.x0; // columns [x0, w)
    if (w <= x0 || h <= y0) return 0;
    const dim3 grid(divUp((unsigned)(w - x0), 64u), (unsigned)blockRows);
    if (!c.extraCountY) ranges.setEdgeFirst(c, y0, h, 4, 8, grid.x, x0, w, 64); // band / tile rendering, edges first (plr.h first_rows, first_cols): blocks of 64 x 4 pixels
    // the spatial filter that reads the history output wants packed texels (PassCtx::consumer, fused_gi.h): written here, for the rows of this launch
    SpatialPackTarget packTarget;
    const SpatialPackTarget* pack = spatialPackTargetOfConsumer(c, 2, 3, &packTarget) == 0 ? &packTarget : nullptr;
    if (pack && (pack->depth.w != out.w || pack->depth.h != out.h || (pack->depth.fmt != F_R16F && pack->depth.fmt != F_D32))) pack = nullptr;
#define PLR_TEMPORAL_ARGS c.storage[0], c.storage[1], c.storage[2], c.storage[3], c.sampled[4], c.sampled[5], c.sampled[6], c.sampled[7], c.sampled[8], c.sampled[9], c.global, w, h, y0, x0, \
                          pack ? pack->packed : nullptr, pack ? pack->depth : ImgView{}, ranges
    // a launch that packs every texel of the consumer's input, and whose four output images nothing else will read (fusion level 2): packed texels only
    const bool packedOnly = pack && (c.elidableStorage & 15u) == 15u && y0 == 0 && h == out.h && x0 == 0 && w == out.w && !c.extraCountY && !c.firstRows[0] && !c.firstRows[1] &&
                            !c.firstCols[0] && !c.firstCols[1];
    if (packedOnly) {
        if (pack->depth.fmt == F_R16F) temporalGiFilterFastKernel<F_R16F, true><<<grid, 256, 0, c.stream>>>(PLR_TEMPORAL_ARGS);
        else temporalGiFilterFastKernel<F_D32, true><<<grid, 256, 0, c.stream>>>(PLR_TEMPORAL_ARGS);
        c.elidedStorage = 15u;
    } else if (pack) {
        if (pack->depth.fmt == F_R16F) temporalGiFilterFastKernel<F_R16F><<<grid, 256, 0, c.stream>>>(PLR_TEMPORAL_ARGS);
        else temporalGiFilterFastKernel<F_D32><<<grid, 256, 0, c.stream>>>(PLR_TEMPORAL_ARGS);
    } else temporalGiFilterFastKernel<0><<<grid, 256, 0, c.stream>>>(PLR_TEMPORAL_ARGS);
#undef PLR_TEMPORAL_ARGS
    PLR_CHECK_LAUNCH(c);
    if (pack) {
        // pixel rows of this launch: [y0, first range end) and, for a launch over two row ranges, the second range (twoRangeBlocks)
        if (c.extraCountY) {
            const PassCtx::RowSpan a = c.rowSpan(out.h, 8);
            spatialNotePackedRect(c, x0, a.y0, w, a.y1);
            spatialNotePackedRect(c, x0, std::min((int)c.extraBaseY * 8, (int)out.h), w, h);
        } else spatialNotePackedRect(c, x0, y0, w, h);
    }
    return 0;
}

// fused: applyBloom.comp then tonemapping.comp of the same image over the same rows
static int launchApplyBloomTonemap(const PassCtx* const* ctxs, size_t count) {
    if (count != 2) return kUseGeneralKernel;
    const PassCtx &a = *ctxs[0], &t = *ctxs[1];
    if (!a.hasStorage(0) || !a.hasSampled(1) || !t.hasStorage(0) || !t.hasSampled(1) || !t.global || a.push.size() < 4) return kUseGeneralKernel;
    const ImgView &target = a.storage[0], &bloom = a.sampled[1], &src = t.sampled[1], &dst = t.storage[0];
    if (target.fmt != F_R11G11B10 || bloom.fmt != F_R11G11B10 || (dst.fmt != F_BGRA8 && dst.fmt != F_RGBA8)) return kUseGeneralKernel;
    if (src.ptr != target.ptr || src.w != target.w || src.h != target.h) return kUseGeneralKernel;                 // the tonemap must read what applyBloom wrote
    if (bloom.w != target.w || bloom.h != target.h || dst.w != target.w || dst.h != target.h || (target.w & 3)) return kUseGeneralKernel;
    const PassCtx::RowSpan ra = a.rowSpan(target.h), rt = t.rowSpan(target.h);
    const PassCtx::ColSpan ca = a.colSpan(target.w), ct = t.colSpan(target.w);
    // same rows, same columns, whole 4-pixel groups (the image's width, or a tile's column span: multiples of 8)
    if (ra.y0 != rt.y0 || ra.y1 != rt.y1 || ca.x0 != ct.x0 || ca.x1 != ct.x1 || ((ca.x1 - ca.x0) & 3)) return kUseGeneralKernel;
    if (ra.y1 <= ra.y0 || ca.x1 <= ca.x0) return 0;
    float strength;
    std::memcpy(&strength, a.push.data(), 4);
    const dim3 grid(divUp((unsigned)(ca.x1 - ca.x0), 256u), divUp((unsigned)(ra.y1 - ra.y0), 4u));
    if (dst.fmt == F_BGRA8) applyBloomTonemapKernel<true><<<grid, 256, 0, a.stream>>>(target, bloom, dst, t.global, strength, ca.x1, ra.y1, ra.y0, ca.x0);
    else applyBloomTonemapKernel<false><<<grid, 256, 0, a.stream>>>(target, bloom, dst, t.global, strength, ca.x1, ra.y1, ra.y0, ca.x0);
    PLR_CHECK_LAUNCH(a);
    return 0;
}

} // namespace faststream

// ---- probe: the PLR_MATH_FAST sky LUT lookup (device/fastmath.h) for n directions, same arguments as the oracle's orc_kat_sky_lut
__global__ void skyLutEvalKernel(ImgView lut, const float* __restrict__ dirs, float* __restrict__ out, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const vec3 c = fastm::sampleSkyLut(vec3(dirs[3 * i], dirs[3 * i + 1], dirs[3 * i + 2]), lut);
    out[3 * i] = c.x; out[3 * i + 1] = c.y; out[3 * i + 2] = c.z;
}
int launchSkyLutProbe(const ImgView& lut, const float* dirs, float* out, int64_t n) {
    if (lut.fmt != F_R11G11B10 || lut.d > 1) return setLastError(-1, "plr_debug_sky_lut_eval: the sky LUT must be a 2D R11G11B10 image");
    float *dd = nullptr, *dout = nullptr;
    if (hipMalloc((void**)&dd, n * 12) != hipSuccess || hipMalloc((void**)&dout, n * 12) != hipSuccess) return setLastError(-2, "plr_debug_sky_lut_eval: hipMalloc failed");
    hipMemcpy(dd, dirs, n * 12, hipMemcpyHostToDevice);
    skyLutEvalKernel<<<(unsigned)((n + 255) / 256), 256>>>(lut, dd, dout, n);
    const hipError_t e = hipGetLastError();
    hipMemcpy(out, dout, n * 12, hipMemcpyDeviceToHost);
    hipFree(dd); hipFree(dout);
    return e == hipSuccess ? 0 : setLastError(-2, "plr_debug_sky_lut_eval: launch failed");
}
static int fused_apply_bloom_tonemap(const PassCtx* const* ctxs, size_t count) { return faststream::launchApplyBloomTonemap(ctxs, count); }
PLR_REGISTER_FUSION("applyBloom + tonemapping", fused_apply_bloom_tonemap, "applyBloom.comp", "tonemapping.comp");
static int faststream_apply_bloom(const PassCtx& c) { return faststream::launchApplyBloom(c); }
static int faststream_tonemapping(const PassCtx& c) { return faststream::launchTonemapping(c); }
static int faststream_upscale(const PassCtx& c) { return faststream::launchUpscale(c); }
static int faststream_temporal_gi(const PassCtx& c) { return faststream::launchTemporalGi(c); }
PLR_REGISTER_SHADER_FAST("applyBloom.comp", faststream_apply_bloom);
PLR_REGISTER_SHADER_FAST("tonemapping.comp", faststream_tonemapping);
PLR_REGISTER_SHADER_FAST("indirectLightUpscale.comp", faststream_upscale);
PLR_REGISTER_SHADER_FAST("filterIndirectDiffuseTemporal.comp", faststream_temporal_gi);
static int faststream_temporal_gi_two_ranges(const PassCtx* const* ctxs, size_t count) { return launchOverTwoRowRanges(ctxs, count, faststream_temporal_gi); }
PLR_REGISTER_FUSION("filterIndirectDiffuseTemporal over two row ranges", faststream_temporal_gi_two_ranges, "filterIndirectDiffuseTemporal.comp", "filterIndirectDiffuseTemporal.comp");
} // namespace plr

// ---- exhaustive check of the fast R11G11B10 encoder (include/plr.h plr_debug_verify_r11g11b10_fast)
namespace plr {
__global__ void verifyUFloatKernel(uint32_t first, unsigned long long* __restrict__ out) {
    const uint32_t u = first + blockIdx.x * blockDim.x + threadIdx.x;
    const float v = u2f(u);
    // one channel at a time, the other two zero: a pattern the fast path does not take falls back to the exact encoder inside packR11G11B10Fast
    const uint32_t fr = packR11G11B10Fast(vec3(v, 0.f, 0.f)) & 0x7ffu, er = encodeUFloat<6>(v);
    const uint32_t fb = packR11G11B10Fast(vec3(0.f, 0.f, v)) >> 22, eb = encodeUFloat<5>(v);
    if (fr != er) { atomicAdd(out + 0, 1ull); atomicMax(out + 2, (unsigned long long)(fr > er ? fr - er : er - fr)); atomicMax(out + 3, (unsigned long long)u); }
    if (fb != eb) { atomicAdd(out + 1, 1ull); atomicMax(out + 2, (unsigned long long)(fb > eb ? fb - eb : eb - fb)); atomicMax(out + 3, (unsigned long long)u); }
}
} // namespace plr

extern "C" int plr_debug_verify_r11g11b10_fast(uint64_t* out4) {
    using namespace plr;
    if (!out4) return setLastError(-1, "plr_debug_verify_r11g11b10_fast: null argument");
    unsigned long long* dev = nullptr;
    if (hipMalloc((void**)&dev, 32) != hipSuccess) return setLastError(-2, "plr_debug_verify_r11g11b10_fast: hipMalloc failed");
    hipMemset(dev, 0, 32);
    const uint32_t chunk = 1u << 28;
    for (uint64_t first = 0; first < (1ull << 32); first += chunk) verifyUFloatKernel<<<chunk / 256u, 256>>>((uint32_t)first, dev);
    unsigned long long host[4];
    const hipError_t e = hipMemcpy(host, dev, 32, hipMemcpyDeviceToHost);
    hipFree(dev);
    if (e != hipSuccess) return setLastError(-2, "plr_debug_verify_r11g11b10_fast: kernel failed");
    for (int i = 0; i < 4; i++) out4[i] = host[i];
    return 0;
}
