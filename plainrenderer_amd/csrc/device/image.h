// Linear HBM images for the HIP kernels: native packed pixel formats, tight row pitch, explicit
// address math instead of texture units so a wave64 reads 64 consecutive texels per row segment.
//
// Fixed-function rules (Vulkan leaves parts of these implementation defined; DESIGN.md
// "sampler & format contract"):
//  * float -> half / 11-bit / 10-bit float is round-to-nearest-even; 11/10-bit: negatives -> 0,
//    finite overflow -> max finite, NaN -> NaN, +Inf -> +Inf
//  * UNORM8 encode round-half-even(clamp(x,0,1)*255), NaN -> 0; SNORM16 decode max(c/32767,-1)
//  * linear filtering has 8 fractional weight bits: t = floor((u-0.5)*256+0.5), i0 = t>>8, a = (t&255)/256
//  * texelFetch/imageLoad out of bounds = 0, imageStore out of bounds is dropped
// Format numbering = ImageFormat in the reference (Plain/src/Common/ImageDescription.h:16).
#pragma once
#include "vecmath.h"

namespace plr {

enum Format : int32_t {
    F_R8 = 0, F_RG8 = 1, F_RGBA8 = 2, F_R16F = 3, F_RG16F = 4, F_RG32F = 5, F_RG16SN = 6, F_RGBA16F = 7,
    F_RGBA16SN = 8, F_RGBA32F = 9, F_R11G11B10 = 10, F_D16 = 11, F_D32 = 12, F_BC1 = 13, F_BC3 = 14, F_BC5 = 15,
    F_BGRA8 = 16
};

struct ImgView {
    void* ptr;
    int32_t w, h, d;
    int32_t fmt;
};

// ---- codecs ----
PLR_DI float halfBitsToFloat(uint32_t h) {
    union { uint16_t u; _Float16 f; } c;
    c.u = (uint16_t)h;
    return (float)c.f; // v_cvt_f32_f16, exact
}
PLR_DI uint32_t floatToHalfBits(float v) {
#ifndef PLR_FAST_SET
    // The value must exist as a rounded fp32 number before it is converted: without this barrier the compiler may select v_fma_mixlo_f16
    // for "product, then conversion" - one rounding straight to half instead of two - even with -ffp-contract=off, and a texel in ten
    // thousand then differs from the scalar evaluation by a half-float ulp (found in froxelLightScattering, tools/dbg_volumetrics.py).
    asm volatile("" : "+v"(v));
#endif
    union { uint16_t u; _Float16 f; } c;
    c.f = (_Float16)v; // v_cvt_f16_f32, round-to-nearest-even in the default mode
    return c.u;
}

// an 11-bit (5e6m) / 10-bit (5e5m) unsigned float is a positive half with the low mantissa bits dropped
PLR_DI vec3 unpackR11G11B10(uint32_t p) {
    return vec3(halfBitsToFloat((p & 0x7ffu) << 4), halfBitsToFloat(((p >> 11) & 0x7ffu) << 4), halfBitsToFloat((p >> 22) << 5));
}

// Branch-free: every pass that writes an R11G11B10 image encodes three of these per pixel, and the branchy form (NaN / sign / infinity /
// subnormal tests as early returns) cost a fifth of the bloom kernels' time in divergent-branch overhead. Both range results are computed
// and the special cases override them in reverse priority, so the value is the same for every input bit pattern.
template <int M> PLR_DI uint32_t encodeUFloat(float v) {
    constexpr uint32_t expMax = 31u << M;
    constexpr uint32_t maxFinite = (30u << M) | ((1u << M) - 1u);
    constexpr int shift = 23 - M;
    const uint32_t u = f2u(v);
    // normal results: round the mantissa to nearest even (meaningless below 2^-14, where the subtraction wraps: not selected)
    const uint32_t t = u + ((1u << (shift - 1)) - 1u) + ((u >> shift) & 1u);
    const uint32_t normal = (t >> shift) - (112u << M);
    // subnormal results: adding 2^(23-14-M) makes the fp32 adder round to the 2^-(14+M) grid (RTE)
    constexpr float magic = (float)(1u << (9 - M));
    const uint32_t subnormal = f2u(v + magic) - f2u(magic);
    uint32_t r = v < 6.103515625e-05f ? subnormal : normal;
    r = r < maxFinite ? r : maxFinite;                                           // finite overflow -> largest finite
    r = u == 0x7f800000u ? expMax : r;                                           // +inf
    r = (u >> 31) ? 0u : r;                                                      // negative, -0, -inf
    r = (u & 0x7fffffffu) > 0x7f800000u ? (expMax | (1u << (M - 1))) : r;        // NaN (either sign)
    return r;
}

PLR_DI uint32_t packR11G11B10Exact(vec3 c) {
    return encodeUFloat<6>(c.x) | (encodeUFloat<6>(c.y) << 11) | (encodeUFloat<5>(c.z) << 22);
}

// PLR_MATH_FAST encoder. encodeUFloat above spends most of its ~19 instructions per channel (11 of them half-rate compares / selects) on inputs
// that almost never occur; every pass that writes a colour image pays that three times per pixel (the bloom chain: half its instructions).
// Here a value v in [0, 64768) is scaled by 2^-112, which moves the 5-bit exponent range onto fp32's biased exponents 1 .. 30 and the
// subnormals of the small format onto fp32's subnormals, so ONE integer round-to-nearest-even of the fp32 bit pattern yields exponent and
// mantissa for normal and subnormal results alike (4 instructions, carries into the exponent included). Anything else - negative, -0,
// infinity, NaN, or large enough to round past the largest finite value of the 10-bit channel - sends the whole wave through the exact
// encoder (one max3 + compare per pixel to find out). Same bits as the exact encoder except for the double rounding in the subnormal range
// (the scaling itself rounds once to fp32's subnormal grid): one code for < 2^-17 of the inputs below 2^-14, none above; counted for all
// 2^32 bit patterns by plr_debug_verify_r11g11b10_fast.
template <int M> PLR_DI uint32_t encodeUFloatInRange(float v) {
    constexpr int shift = 23 - M;
    const uint32_t s = f2u(v * 1.92592994438723585e-34f); // 2^-112, exact for results that stay normal
    return (s + ((1u << (shift - 1)) - 1u) + ((s >> shift) & 1u)) >> shift;
}
constexpr uint32_t kUFloatInRangeLimit = 0x477d0000u; // 64768.0f: 10-bit channel's largest finite value 64512 + half a step (ties round up to infinity)
// out of line: the exact encoder is ~60 instructions and a kernel may encode at dozens of sites
__device__ __attribute__((noinline)) inline uint32_t packR11G11B10OutOfRange(float x, float y, float z) { return packR11G11B10Exact(vec3(x, y, z)); }
PLR_DI uint32_t packR11G11B10Fast(vec3 c) {
    const uint32_t top = max(max(f2u(c.x), f2u(c.y)), f2u(c.z)); // as unsigned integers negative values and NaNs are the largest
    if (__builtin_amdgcn_ballot_w64(top >= kUFloatInRangeLimit) != 0ull) return packR11G11B10OutOfRange(c.x, c.y, c.z); // wave-uniform
    return encodeUFloatInRange<6>(c.x) | (encodeUFloatInRange<6>(c.y) << 11) | (encodeUFloatInRange<5>(c.z) << 22);
}

PLR_DI uint32_t packR11G11B10(vec3 c) {
#ifdef PLR_FAST_SET
    return packR11G11B10Fast(c);
#else
    return packR11G11B10Exact(c);
#endif
}

PLR_DI float decodeUnorm8(uint32_t c) { return (float)c / 255.0f; }
// The same value in three instructions where the file is built with IEEE division (ten): the product with the rounded reciprocal and one Newton step on it.
// The residual c - 255 q is exact in an FMA, and the corrected quotient equals the correctly rounded c / 255 for all 256 codes (plr_debug_math_eval fn 14,
// tests/test_gpu_foundations.py; also holds with the reciprocal one ulp off either way).
PLR_DI float decodeUnorm8Newton(uint32_t c) {
    const float x = (float)c, r = 0x1.010102p-8f, q = x * r;
    return __builtin_fmaf(__builtin_fmaf(-255.f, q, x), r, q);
}
PLR_DI uint32_t encodeUnorm8(float v) {
    if (v != v) return 0u;
    return (uint32_t)__float2int_rn(gclamp(v, 0.f, 1.f) * 255.0f);
}
PLR_DI float decodeSnorm16(int32_t c) { return gmax((float)c / 32767.0f, -1.0f); }
PLR_DI float decodeUnorm16(uint32_t c) { return (float)c / 65535.0f; }

// ---- typed texel access (idx = linear texel index) ----
template <int FMT> struct Texel;
template <> struct Texel<F_R11G11B10> {
    static PLR_DI vec4 load(const void* p, size_t i) { return vec4(unpackR11G11B10(((const uint32_t*)p)[i]), 1.f); }
    static PLR_DI void store(void* p, size_t i, vec4 v) { ((uint32_t*)p)[i] = packR11G11B10(v.xyz()); }
};
template <> struct Texel<F_D32> {
    static PLR_DI vec4 load(const void* p, size_t i) { return vec4(((const float*)p)[i], 0.f, 0.f, 1.f); }
    static PLR_DI void store(void* p, size_t i, vec4 v) { ((float*)p)[i] = v.x; }
};
template <> struct Texel<F_D16> {
    static PLR_DI vec4 load(const void* p, size_t i) { return vec4(decodeUnorm16(((const uint16_t*)p)[i]), 0.f, 0.f, 1.f); }
};
template <> struct Texel<F_R16F> {
    static PLR_DI vec4 load(const void* p, size_t i) { return vec4(halfBitsToFloat(((const uint16_t*)p)[i]), 0.f, 0.f, 1.f); }
    static PLR_DI void store(void* p, size_t i, vec4 v) { ((uint16_t*)p)[i] = (uint16_t)floatToHalfBits(v.x); }
};
template <> struct Texel<F_RG16F> {
    static PLR_DI vec4 load(const void* p, size_t i) {
        const uint32_t u = ((const uint32_t*)p)[i];
        return vec4(halfBitsToFloat(u & 0xffffu), halfBitsToFloat(u >> 16), 0.f, 1.f);
    }
    static PLR_DI void store(void* p, size_t i, vec4 v) { ((uint32_t*)p)[i] = floatToHalfBits(v.x) | (floatToHalfBits(v.y) << 16); }
};
template <> struct Texel<F_RGBA16F> {
    static PLR_DI vec4 load(const void* p, size_t i) {
        const uint2 u = ((const uint2*)p)[i];
        return vec4(halfBitsToFloat(u.x & 0xffffu), halfBitsToFloat(u.x >> 16), halfBitsToFloat(u.y & 0xffffu), halfBitsToFloat(u.y >> 16));
    }
    static PLR_DI void store(void* p, size_t i, vec4 v) {
        uint2 u;
        u.x = floatToHalfBits(v.x) | (floatToHalfBits(v.y) << 16);
        u.y = floatToHalfBits(v.z) | (floatToHalfBits(v.w) << 16);
        ((uint2*)p)[i] = u;
    }
};
template <> struct Texel<F_RG32F> {
    static PLR_DI vec4 load(const void* p, size_t i) { const float2 u = ((const float2*)p)[i]; return vec4(u.x, u.y, 0.f, 1.f); }
    static PLR_DI void store(void* p, size_t i, vec4 v) { ((float2*)p)[i] = make_float2(v.x, v.y); }
};
template <> struct Texel<F_RG16SN> {
    static PLR_DI vec4 load(const void* p, size_t i) {
        const uint32_t u = ((const uint32_t*)p)[i];
        return vec4(decodeSnorm16((int32_t)(int16_t)(u & 0xffffu)), decodeSnorm16((int32_t)(int16_t)(u >> 16)), 0.f, 1.f);
    }
};
template <> struct Texel<F_RGBA8> {
    static PLR_DI vec4 load(const void* p, size_t i) {
        const uint32_t u = ((const uint32_t*)p)[i];
        return vec4(decodeUnorm8(u & 0xffu), decodeUnorm8((u >> 8) & 0xffu), decodeUnorm8((u >> 16) & 0xffu), decodeUnorm8(u >> 24));
    }
    static PLR_DI void store(void* p, size_t i, vec4 v) {
        ((uint32_t*)p)[i] = encodeUnorm8(v.x) | (encodeUnorm8(v.y) << 8) | (encodeUnorm8(v.z) << 16) | (encodeUnorm8(v.w) << 24);
    }
};
template <> struct Texel<F_BGRA8> {
    static PLR_DI vec4 load(const void* p, size_t i) {
        const uint32_t u = ((const uint32_t*)p)[i];
        return vec4(decodeUnorm8((u >> 16) & 0xffu), decodeUnorm8((u >> 8) & 0xffu), decodeUnorm8(u & 0xffu), decodeUnorm8(u >> 24));
    }
    static PLR_DI void store(void* p, size_t i, vec4 v) {
        ((uint32_t*)p)[i] = encodeUnorm8(v.z) | (encodeUnorm8(v.y) << 8) | (encodeUnorm8(v.x) << 16) | (encodeUnorm8(v.w) << 24);
    }
};
template <> struct Texel<F_RG8> {
    static PLR_DI vec4 load(const void* p, size_t i) {
        const uint32_t u = ((const uint16_t*)p)[i];
        return vec4(decodeUnorm8(u & 0xffu), decodeUnorm8(u >> 8), 0.f, 1.f);
    }
};
template <> struct Texel<F_R8> {
    static PLR_DI vec4 load(const void* p, size_t i) { return vec4(decodeUnorm8(((const uint8_t*)p)[i]), 0.f, 0.f, 1.f); }
    static PLR_DI void store(void* p, size_t i, vec4 v) { ((uint8_t*)p)[i] = (uint8_t)encodeUnorm8(v.x); }
};

// ---- samplers (resources/shaders/global.inc:35-42) ----
enum Address { CLAMP = 0, REPEAT = 1, BORDER_WHITE = 2, BORDER_BLACK = 3 };

// clamp to [0, hi], hi >= 0: ONE v_med3_i32. The compiler only forms it from min(max()) when both bounds are constants, and leaves two half-rate
// instructions otherwise (the deferred shade had 108 of them); the asm is not volatile, so it is scheduled and hoisted like any other instruction
PLR_DI int clampTo(int i, int hi) {
    int r;
    asm("v_med3_i32 %0, %1, 0, %2" : "=v"(r) : "v"(i), "v"(hi));
    return r;
}
PLR_DI int clampi(int i, int n) { return clampTo(i, n - 1); } // n >= 1
PLR_DI int repeati(int i, int n) { int m = i % n; return m < 0 ? m + n : m; }
PLR_DI float saneCoord(float u) { return gclamp(u, -1.0e6f, 1.0e6f); }
#ifdef PLR_FAST_SET
// floor and conversion in ONE instruction (v_cvt_flr_i32_f32, gfx9). The conversion saturates and maps NaN to 0, so the range clamp that keeps
// (int)floorf() defined is not needed in front of it: a coordinate beyond +-2^31 ends on the first / last texel after the index clamp either way
PLR_DI int floorToInt(float x) {
    int r;
    asm("v_cvt_flr_i32_f32_e32 %0, %1" : "=v"(r) : "v"(x));
    return r;
}
#else
PLR_DI int floorToInt(float x) { return (int)floorf(saneCoord(x)); }
#endif
PLR_DI void linearCoord(float u, int* i0, float* alpha) {
#ifdef PLR_FAST_SET
    const int ti = floorToInt((u - 0.5f) * 256.0f + 0.5f);
#else
    const int ti = (int)floorf((saneCoord(u) - 0.5f) * 256.0f + 0.5f);
#endif
    *i0 = ti >> 8;
    *alpha = (float)(ti & 255) * (1.0f / 256.0f);
}

template <int FMT, int ADDR> PLR_DI vec4 addressedTexel2D(const ImgView& im, int x, int y) {
    if (ADDR == CLAMP) { x = clampi(x, im.w); y = clampi(y, im.h); }
    else if (ADDR == REPEAT) { x = repeati(x, im.w); y = repeati(y, im.h); }
    else if (x < 0 || y < 0 || x >= im.w || y >= im.h) return ADDR == BORDER_WHITE ? vec4(1.f, 1.f, 1.f, 1.f) : vec4(0.f, 0.f, 0.f, 1.f);
    return Texel<FMT>::load(im.ptr, (size_t)y * (size_t)im.w + (size_t)x);
}

template <int FMT, int ADDR> PLR_DI vec4 sampleNearest2D(const ImgView& im, vec2 uv) {
    const float u = uv.x * (float)im.w, v = uv.y * (float)im.h;
    return addressedTexel2D<FMT, ADDR>(im, (int)floorf(saneCoord(u)), (int)floorf(saneCoord(v)));
}

template <int FMT, int ADDR> PLR_DI vec4 sampleLinear2D(const ImgView& im, vec2 uv) {
    int i0, j0; float a, b;
    linearCoord(uv.x * (float)im.w, &i0, &a);
    linearCoord(uv.y * (float)im.h, &j0, &b);
    const vec4 t00 = addressedTexel2D<FMT, ADDR>(im, i0, j0);
    const vec4 t10 = addressedTexel2D<FMT, ADDR>(im, i0 + 1, j0);
    const vec4 t01 = addressedTexel2D<FMT, ADDR>(im, i0, j0 + 1);
    const vec4 t11 = addressedTexel2D<FMT, ADDR>(im, i0 + 1, j0 + 1);
    const float w00 = (1.f - a) * (1.f - b), w10 = a * (1.f - b), w01 = (1.f - a) * b, w11 = a * b;
    return t00 * w00 + t10 * w10 + t01 * w01 + t11 * w11;
}


template <int FMT, int ADDR> PLR_DI vec4 addressedTexel3D(const ImgView& im, int x, int y, int z) {
    if (ADDR == CLAMP) { x = clampi(x, im.w); y = clampi(y, im.h); z = clampi(z, im.d); }
    else if (ADDR == REPEAT) { x = repeati(x, im.w); y = repeati(y, im.h); z = repeati(z, im.d); }
    else if (x < 0 || y < 0 || z < 0 || x >= im.w || y >= im.h || z >= im.d) return ADDR == BORDER_WHITE ? vec4(1.f, 1.f, 1.f, 1.f) : vec4(0.f, 0.f, 0.f, 1.f);
    return Texel<FMT>::load(im.ptr, ((size_t)z * (size_t)im.h + (size_t)y) * (size_t)im.w + (size_t)x);
}

template <int FMT, int ADDR> PLR_DI vec4 sampleNearest3D(const ImgView& im, vec3 uvw) {
    return addressedTexel3D<FMT, ADDR>(im, (int)floorf(saneCoord(uvw.x * (float)im.w)), (int)floorf(saneCoord(uvw.y * (float)im.h)), (int)floorf(saneCoord(uvw.z * (float)im.d)));
}

// trilinear sample (8-bit sub-texel weights); ADDR = CLAMP or REPEAT; same term order as oracle/image.h texture3D
template <int FMT, int ADDR> PLR_DI vec4 sampleLinear3D(const ImgView& im, vec3 uvw) {
    int i0, j0, k0; float a, b, c;
    linearCoord(uvw.x * (float)im.w, &i0, &a);
    linearCoord(uvw.y * (float)im.h, &j0, &b);
    linearCoord(uvw.z * (float)im.d, &k0, &c);
    auto wrap = [](int i, int n) { return ADDR == REPEAT ? repeati(i, n) : clampi(i, n); };
    const int x0 = wrap(i0, im.w), x1 = wrap(i0 + 1, im.w), y0 = wrap(j0, im.h), y1 = wrap(j0 + 1, im.h), z0 = wrap(k0, im.d), z1 = wrap(k0 + 1, im.d);
    auto T = [&](int x, int y, int z) { return Texel<FMT>::load(im.ptr, ((size_t)z * (size_t)im.h + (size_t)y) * (size_t)im.w + (size_t)x); };
    const float a0 = 1.f - a, b0 = 1.f - b, c0 = 1.f - c;
    vec4 r(0.f); // the sum starts from +0 like the oracle's loop (keeps the sign of an all-zero result)
    r = r + T(x0, y0, z0) * ((a0 * b0) * c0);
    r = r + T(x1, y0, z0) * ((a * b0) * c0);
    r = r + T(x0, y1, z0) * ((a0 * b) * c0);
    r = r + T(x1, y1, z0) * ((a * b) * c0);
    r = r + T(x0, y0, z1) * ((a0 * b0) * c);
    r = r + T(x1, y0, z1) * ((a * b0) * c);
    r = r + T(x0, y1, z1) * ((a0 * b) * c);
    r = r + T(x1, y1, z1) * ((a * b) * c);
    return r;
}

// textureGather, component 0: (i0,j1), (i1,j1), (i1,j0), (i0,j0) of the bilinear footprint (the offsets table of indirectLightUpscale.comp:42-47)
template <int FMT, int ADDR> PLR_DI vec4 gatherR2D(const ImgView& im, vec2 uv) {
    int i0, j0; float a, b;
    linearCoord(uv.x * (float)im.w, &i0, &a);
    linearCoord(uv.y * (float)im.h, &j0, &b);
    return vec4(addressedTexel2D<FMT, ADDR>(im, i0, j0 + 1).x, addressedTexel2D<FMT, ADDR>(im, i0 + 1, j0 + 1).x, addressedTexel2D<FMT, ADDR>(im, i0 + 1, j0).x,
                addressedTexel2D<FMT, ADDR>(im, i0, j0).x);
}

template <int FMT> PLR_DI vec4 texelFetch2D(const ImgView& im, int x, int y) {
    if (x < 0 || y < 0 || x >= im.w || y >= im.h) return vec4(0.f);
    return Texel<FMT>::load(im.ptr, (size_t)y * (size_t)im.w + (size_t)x);
}

} // namespace plr
