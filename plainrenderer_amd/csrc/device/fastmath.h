// Helpers shared by the PLR_MATH_FAST kernels (kernels_fast/*.hip). On gfx950 every pass of this pipeline is bound by VALU issue,
// so these trade the exact-mode definitions of image.h / detmath.h for the cheapest instruction sequence with the same meaning up
// to rounding: multiplication by a reciprocal constant instead of a division, a mask instead of an integer modulo when the size
// is a power of two, 24-bit multiplies for texel addressing.
#pragma once
#include "image.h"

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
PLR_DI uint32_t texelIndex(uint32_t x, uint32_t y, uint32_t w) { return __umul24(y, w) + x; }

} // namespace fastm
} // namespace plr
