// Helpers shared by the PLR_MATH_FAST kernels (kernels_fast/*.hip). On gfx950 every pass of this pipeline is bound by VALU issue,
// so these trade the exact-mode definitions of image.h / detmath.h for the cheapest instruction sequence with the same meaning up
// to rounding: multiplication by a reciprocal constant instead of a division, a mask instead of an integer modulo when the size
// is a power of two, 24-bit multiplies for texel addressing.
#pragma once
#include "image.h"
#include "shading_common.h"

namespace plr {
namespace fastm {

// UNORM8 texel -> floats: v_cvt_f32_ubyteN + one multiply per channel (image.h divides by 255). The constant is the float just
// BELOW 1/255 (the nearest float is above it and would map 255 to 1.0000001, and sqrt(1 - x) of that is NaN): results never exceed
// the exact quotient and 255 maps to 0.99999994.
constexpr float kInv255 = 0.0039215683937072754f;
PLR_DI vec4 unorm8x4(uint32_t u) {
    const float k = kInv255;
    return vec4((float)(u & 0xffu) * k, (float)((u >> 8) & 0xffu) * k, (float)((u >> 16) & 0xffu) * k, (float)(u >> 24) * k);
}
PLR_DI vec2 unorm8x2(uint32_t u) {
    const float k = kInv255;
    return vec2((float)(u & 0xffu) * k, (float)((u >> 8) & 0xffu) * k);
}
// repeat addressing of a non-negative or negative coordinate: a mask when n is a power of two (noise textures are 32x32)
PLR_DI int repeatIndex(int i, int n) { return (n & (n - 1)) == 0 ? (i & (n - 1)) : repeati(i, n); }
// texel index y * w + x for images below 2^24 texels per side: v_mad_u32_u24 (full rate) instead of a 32/64-bit multiply (quarter rate)
// saneCoord of image.h (keeps the float -> int conversion of a sampler coordinate defined) as one v_med3_f32; a NaN coordinate becomes -1e6
PLR_DI float clampCoord(float u) { return __builtin_amdgcn_fmed3f(u, -1.0e6f, 1.0e6f); }
PLR_DI uint32_t texelIndex(uint32_t x, uint32_t y, uint32_t w) { return __umul24(y, w) + x; }

// ---- sky LUT lookup (sky.inc:86-94, 112-116) for the fast kernels. The exact path spends ~350 VALU instructions per lookup on the software
// acos / atan2 of detmath.h and on integer modulo for the repeat addressing. Here: polynomial acos (Abramowitz & Stegun 4.4.46,
// |error| <= 2e-8 rad + rounding: the v coordinate is sqrt(|2 theta / pi - 1|), infinitely steep at the horizon, so the 7e-5 rad of the
// shorter 4.4.45 form moved horizon directions by a tenth of a LUT row) and atan (|error| <= 1e-5 rad = 3e-4 LUT columns), and a
// conditional wrap (the u coordinate lies in [0, 1], so the bilinear footprint can only step one texel across the seam).
PLR_DI float acosFast(float x) {
    const float a = __builtin_fminf(fabsf(x), 1.f); // a direction normalised in fp32 can have a component of 1 + 1 ulp: sqrt(1 - a) must not see it
    const float p = 1.5707963050f + a * (-0.2145988016f + a * (0.0889789874f + a * (-0.0501743046f + a * (0.0308918810f + a * (-0.0170881256f + a * (0.0066700901f + a * -0.0012624911f))))));
    const float r = __builtin_amdgcn_sqrtf(1.f - a) * p;
    return x < 0.f ? 3.14159265f - r : r;
}
PLR_DI float atan2Fast(float y, float x) {
    const float ax = fabsf(x), ay = fabsf(y);
    const float mx = __builtin_fmaxf(ax, ay), mn = __builtin_fminf(ax, ay);
    const float t = mn * __builtin_amdgcn_rcpf(__builtin_fmaxf(mx, 1e-30f)), s = t * t;
    float r = t * (0.99997726f + s * (-0.33262347f + s * (0.19354346f + s * (-0.11643287f + s * (0.05265332f + s * -0.01172120f)))));
    r = ay > ax ? 1.57079633f - r : r;
    r = x < 0.f ? 3.14159265f - r : r;
    return y < 0.f ? -r : r;
}
PLR_DI vec3 sampleSkyLut(vec3 V, const ImgView& lut) {
    const float theta = acosFast(-V.y);
    const float yl = theta * (1.f / PLR_GLSL_PI) * 2.f - 1.f;
    const float ys = __builtin_amdgcn_sqrtf(fabsf(yl));
    const float v = __builtin_amdgcn_fmed3f((yl < 0.f ? -ys : ys) * 0.5f + 0.5f, 0.005f, 0.995f);
    const float u = -atan2Fast(V.z, V.x) * (1.f / (2.f * 3.1415f)) + 0.5f;
    int i0, j0; float a, b;
    linearCoord(u * (float)lut.w, &i0, &a);
    linearCoord(v * (float)lut.h, &j0, &b);
    int x0 = i0, x1 = i0 + 1;
    x0 = x0 < 0 ? x0 + lut.w : (x0 >= lut.w ? x0 - lut.w : x0);
    x1 = x1 < 0 ? x1 + lut.w : (x1 >= lut.w ? x1 - lut.w : x1);
    x0 = clampi(x0, lut.w); x1 = clampi(x1, lut.w); // (a coordinate outside [0, 1] cannot occur for a finite direction; stay in bounds anyway)
    int y0 = j0, y1 = j0 + 1; // v is clamped to [0.005, 0.995]: the footprint is at most one texel outside the image
    y0 = y0 < 0 ? y0 + lut.h : (y0 >= lut.h ? y0 - lut.h : y0);
    y1 = y1 < 0 ? y1 + lut.h : (y1 >= lut.h ? y1 - lut.h : y1);
    y0 = clampi(y0, lut.h); y1 = clampi(y1, lut.h);
    const uint32_t* t = (const uint32_t*)lut.ptr;
    const vec3 t00 = unpackR11G11B10(t[texelIndex((uint32_t)x0, (uint32_t)y0, (uint32_t)lut.w)]), t10 = unpackR11G11B10(t[texelIndex((uint32_t)x1, (uint32_t)y0, (uint32_t)lut.w)]);
    const vec3 t01 = unpackR11G11B10(t[texelIndex((uint32_t)x0, (uint32_t)y1, (uint32_t)lut.w)]), t11 = unpackR11G11B10(t[texelIndex((uint32_t)x1, (uint32_t)y1, (uint32_t)lut.w)]);
    const vec3 top = t00 + (t10 - t00) * a, bot = t01 + (t11 - t01) * a;
    return top + (bot - top) * b;
}

} // namespace fastm
} // namespace plr
