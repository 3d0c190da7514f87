// ORACLE (test infrastructure, never shipped, never on the product path).
// Pixel-format codecs and software samplers: the fixed-function rules the reference
// GLSL relies on but that live in the Vulkan driver/hardware, not in the reference tree
// (SURVEY.md Appendix B). Format enum order follows Plain/src/Common/ImageDescription.h:16.
//
// Rules fixed by this build (stated in DESIGN.md "sampler & format contract"):
//  * float -> half / 11-bit / 10-bit float: round-to-nearest-even; 11/10-bit: negatives -> 0,
//    finite overflow -> max finite, NaN -> NaN, +Inf -> +Inf.
//  * UNORM8 encode: round-half-even(clamp(x,0,1)*255), NaN -> 0. SNORM16 decode: max(c/32767,-1).
//  * linear filtering uses 8 fractional weight bits (Vulkan subTexelPrecisionBits = 8, what AMD
//    hardware implements): t = floor((u-0.5)*256 + 0.5); i0 = t >> 8; alpha = (t & 255)/256.
//  * texelFetch / imageLoad out of bounds returns 0; imageStore out of bounds is dropped.
#pragma once
#include "glslmath.h"

namespace orc {

enum Format : int32_t {
    F_R8 = 0, F_RG8 = 1, F_RGBA8 = 2, F_R16F = 3, F_RG16F = 4, F_RG32F = 5, F_RG16SN = 6, F_RGBA16F = 7,
    F_RGBA16SN = 8, F_RGBA32F = 9, F_R11G11B10 = 10, F_D16 = 11, F_D32 = 12, F_BC1 = 13, F_BC3 = 14, F_BC5 = 15,
    F_BGRA8 = 16
};

static inline int formatBytes(int f) {
    switch (f) {
        case F_R8: return 1; case F_RG8: return 2; case F_RGBA8: return 4; case F_R16F: return 2;
        case F_RG16F: return 4; case F_RG32F: return 8; case F_RG16SN: return 4; case F_RGBA16F: return 8;
        case F_RGBA16SN: return 8; case F_RGBA32F: return 16; case F_R11G11B10: return 4; case F_D16: return 2;
        case F_D32: return 4; case F_BGRA8: return 4; default: return 0;
    }
}

// ---- small floats ----
// unsigned float with 5 exponent bits (bias 15) and M mantissa bits
static inline uint32_t encodeUFloat(float v, int M) {
    const uint32_t f = f2u(v);
    const uint32_t expMax = 31u << M;
    const uint32_t maxFinite = (30u << M) | ((1u << M) - 1u);
    const uint32_t ex = (f >> 23) & 0xffu;
    uint32_t man = f & 0x7fffffu;
    if (ex == 255u) {
        if (man) return expMax | (1u << (M - 1)); // NaN
        return (f >> 31) ? 0u : expMax;            // -inf -> 0, +inf -> inf
    }
    if (f >> 31) return 0u; // negative (and -0)
    const int e = (int)ex - 127 + 15;
    if (e >= 31) return maxFinite;
    uint32_t r;
    if (e <= 0) {
        if (ex == 0u) return 0u; // fp32 zero / subnormal: far below the smallest 2^-20 step
        man |= 0x800000u;
        const int shift = (23 - M) + (1 - e);
        if (shift > 24) return 0u;
        r = man >> shift;
        const uint32_t rem = man & ((1u << shift) - 1u);
        const uint32_t half = 1u << (shift - 1);
        if (rem > half || (rem == half && (r & 1u))) r++;
        return r;
    }
    const int shift = 23 - M;
    r = ((uint32_t)e << M) | (man >> shift);
    const uint32_t rem = man & ((1u << shift) - 1u);
    const uint32_t half = 1u << (shift - 1);
    if (rem > half || (rem == half && (r & 1u))) r++;
    if (r >= expMax) return maxFinite;
    return r;
}

static inline float decodeUFloat(uint32_t v, int M) {
    const uint32_t e = v >> M;
    const uint32_t m = v & ((1u << M) - 1u);
    if (e == 0u) return (float)m * (1.0f / (float)(1u << M)) * 6.103515625e-05f; // m/2^M * 2^-14
    if (e == 31u) return m ? u2f(0x7fc00000u) : u2f(0x7f800000u);
    return u2f(((e + 112u) << 23) | (m << (23 - M)));
}

static inline uint32_t packR11G11B10(vec3 c) {
    return encodeUFloat(c.x, 6) | (encodeUFloat(c.y, 6) << 11) | (encodeUFloat(c.z, 5) << 22);
}
static inline vec3 unpackR11G11B10(uint32_t p) {
    return vec3(decodeUFloat(p & 0x7ffu, 6), decodeUFloat((p >> 11) & 0x7ffu, 6), decodeUFloat(p >> 22, 5));
}

static inline uint16_t floatToHalf(float v) {
    const uint32_t f = f2u(v);
    const uint16_t sign = (uint16_t)((f >> 16) & 0x8000u);
    const uint32_t ex = (f >> 23) & 0xffu;
    uint32_t man = f & 0x7fffffu;
    if (ex == 255u) return sign | 0x7c00u | (man ? 0x200u : 0u);
    const int e = (int)ex - 127 + 15;
    if (e >= 31) return sign | 0x7c00u; // RTE overflow -> inf
    uint32_t r;
    if (e <= 0) {
        if (ex == 0u) return sign;
        man |= 0x800000u;
        const int shift = 13 + (1 - e);
        if (shift > 24) return sign;
        r = man >> shift;
        const uint32_t rem = man & ((1u << shift) - 1u);
        const uint32_t half = 1u << (shift - 1);
        if (rem > half || (rem == half && (r & 1u))) r++;
        return sign | (uint16_t)r;
    }
    r = ((uint32_t)e << 10) | (man >> 13);
    const uint32_t rem = man & 0x1fffu;
    if (rem > 0x1000u || (rem == 0x1000u && (r & 1u))) r++;
    return sign | (uint16_t)r; // a carry into exponent 31 yields inf, as IEEE RTE requires
}

static inline float halfToFloat(uint16_t h) {
    const uint32_t sign = ((uint32_t)h & 0x8000u) << 16;
    const uint32_t e = (h >> 10) & 0x1fu;
    const uint32_t m = h & 0x3ffu;
    if (e == 0u) {
        const float v = (float)m * (1.0f / 1024.0f) * 6.103515625e-05f;
        return u2f(f2u(v) | sign);
    }
    if (e == 31u) return u2f(sign | 0x7f800000u | (m << 13));
    return u2f(sign | ((e + 112u) << 23) | (m << 13));
}

static inline float roundHalfEven(float x) { return std::nearbyint(x); } // default FE_TONEAREST
static inline uint8_t encodeUnorm8(float v) {
    if (v != v) return 0;
    return (uint8_t)roundHalfEven(gclamp(v, 0.f, 1.f) * 255.0f);
}
static inline float decodeUnorm8(uint8_t c) { return (float)c / 255.0f; }
static inline float decodeSnorm16(int16_t c) { return gmax((float)c / 32767.0f, -1.0f); }
static inline int16_t encodeSnorm16(float v) {
    if (v != v) return 0;
    return (int16_t)roundHalfEven(gclamp(v, -1.f, 1.f) * 32767.0f);
}
static inline float decodeUnorm16(uint16_t c) { return (float)c / 65535.0f; }

// ---- images ----
struct Image {
    void* data;
    int32_t w, h, d;
    int32_t format;
};

static inline size_t texelIndex(const Image& im, int x, int y, int z) {
    return ((size_t)z * (size_t)im.h + (size_t)y) * (size_t)im.w + (size_t)x;
}

// in-bounds fetch; missing channels follow Vulkan (0,0,0,1)
static inline vec4 loadTexel(const Image& im, int x, int y, int z = 0) {
    const size_t i = texelIndex(im, x, y, z);
    switch (im.format) {
        case F_R8: return vec4(decodeUnorm8(((const uint8_t*)im.data)[i]), 0, 0, 1);
        case F_RG8: { const uint8_t* p = (const uint8_t*)im.data + i * 2; return vec4(decodeUnorm8(p[0]), decodeUnorm8(p[1]), 0, 1); }
        case F_RGBA8: { const uint8_t* p = (const uint8_t*)im.data + i * 4; return vec4(decodeUnorm8(p[0]), decodeUnorm8(p[1]), decodeUnorm8(p[2]), decodeUnorm8(p[3])); }
        case F_BGRA8: { const uint8_t* p = (const uint8_t*)im.data + i * 4; return vec4(decodeUnorm8(p[2]), decodeUnorm8(p[1]), decodeUnorm8(p[0]), decodeUnorm8(p[3])); }
        case F_R16F: return vec4(halfToFloat(((const uint16_t*)im.data)[i]), 0, 0, 1);
        case F_RG16F: { const uint16_t* p = (const uint16_t*)im.data + i * 2; return vec4(halfToFloat(p[0]), halfToFloat(p[1]), 0, 1); }
        case F_RGBA16F: { const uint16_t* p = (const uint16_t*)im.data + i * 4; return vec4(halfToFloat(p[0]), halfToFloat(p[1]), halfToFloat(p[2]), halfToFloat(p[3])); }
        case F_RG32F: { const float* p = (const float*)im.data + i * 2; return vec4(p[0], p[1], 0, 1); }
        case F_RGBA32F: { const float* p = (const float*)im.data + i * 4; return vec4(p[0], p[1], p[2], p[3]); }
        case F_RG16SN: { const int16_t* p = (const int16_t*)im.data + i * 2; return vec4(decodeSnorm16(p[0]), decodeSnorm16(p[1]), 0, 1); }
        case F_R11G11B10: return vec4(unpackR11G11B10(((const uint32_t*)im.data)[i]), 1);
        case F_D16: return vec4(decodeUnorm16(((const uint16_t*)im.data)[i]), 0, 0, 1);
        case F_D32: return vec4(((const float*)im.data)[i], 0, 0, 1);
        default: return vec4(0);
    }
}

static inline void storeTexel(const Image& im, int x, int y, int z, vec4 v) {
    if (x < 0 || y < 0 || z < 0 || x >= im.w || y >= im.h || z >= im.d) return; // dropped
    const size_t i = texelIndex(im, x, y, z);
    switch (im.format) {
        case F_R8: ((uint8_t*)im.data)[i] = encodeUnorm8(v.x); break;
        case F_RG8: { uint8_t* p = (uint8_t*)im.data + i * 2; p[0] = encodeUnorm8(v.x); p[1] = encodeUnorm8(v.y); break; }
        case F_RGBA8: { uint8_t* p = (uint8_t*)im.data + i * 4; p[0] = encodeUnorm8(v.x); p[1] = encodeUnorm8(v.y); p[2] = encodeUnorm8(v.z); p[3] = encodeUnorm8(v.w); break; }
        case F_BGRA8: { uint8_t* p = (uint8_t*)im.data + i * 4; p[2] = encodeUnorm8(v.x); p[1] = encodeUnorm8(v.y); p[0] = encodeUnorm8(v.z); p[3] = encodeUnorm8(v.w); break; }
        case F_R16F: ((uint16_t*)im.data)[i] = floatToHalf(v.x); break;
        case F_RG16F: { uint16_t* p = (uint16_t*)im.data + i * 2; p[0] = floatToHalf(v.x); p[1] = floatToHalf(v.y); break; }
        case F_RGBA16F: { uint16_t* p = (uint16_t*)im.data + i * 4; p[0] = floatToHalf(v.x); p[1] = floatToHalf(v.y); p[2] = floatToHalf(v.z); p[3] = floatToHalf(v.w); break; }
        case F_RG32F: { float* p = (float*)im.data + i * 2; p[0] = v.x; p[1] = v.y; break; }
        case F_RGBA32F: { float* p = (float*)im.data + i * 4; p[0] = v.x; p[1] = v.y; p[2] = v.z; p[3] = v.w; break; }
        case F_RG16SN: { int16_t* p = (int16_t*)im.data + i * 2; p[0] = encodeSnorm16(v.x); p[1] = encodeSnorm16(v.y); break; }
        case F_R11G11B10: ((uint32_t*)im.data)[i] = packR11G11B10(v.xyz()); break;
        case F_D32: ((float*)im.data)[i] = v.x; break;
        default: break;
    }
}

// texelFetch / imageLoad: out of bounds -> 0
static inline vec4 texelFetch(const Image& im, ivec2 p) {
    if (p.x < 0 || p.y < 0 || p.x >= im.w || p.y >= im.h) return vec4(0);
    return loadTexel(im, p.x, p.y, 0);
}
static inline void imageStore(const Image& im, ivec2 p, vec4 v) { storeTexel(im, p.x, p.y, 0, v); }

// ---- samplers (the 8 global samplers of resources/shaders/global.inc:35-42) ----
enum Filter { NEAREST = 0, LINEAR = 1 };
enum Address { CLAMP = 0, REPEAT = 1, BORDER_WHITE = 2, BORDER_BLACK = 3 };

static inline int wrapIndex(int i, int n, int addr, bool* border) {
    if (addr == CLAMP) return i < 0 ? 0 : (i >= n ? n - 1 : i);
    if (addr == REPEAT) { int m = i % n; return m < 0 ? m + n : m; }
    if (i < 0 || i >= n) { *border = true; return 0; }
    return i;
}

static inline vec4 borderColor(int addr) { return addr == BORDER_WHITE ? vec4(1, 1, 1, 1) : vec4(0, 0, 0, 1); }

static inline vec4 addressedTexel(const Image& im, int x, int y, int z, int addr) {
    bool border = false;
    const int xi = wrapIndex(x, im.w, addr, &border);
    const int yi = wrapIndex(y, im.h, addr, &border);
    const int zi = (im.d > 1) ? wrapIndex(z, im.d, addr, &border) : 0;
    if (border) return borderColor(addr);
    return loadTexel(im, xi, yi, zi);
}

// coordinates far outside any image are clamped before the float->int conversion
static inline float saneCoord(float u) { return gclamp(u, -1.0e6f, 1.0e6f); }

// 8-bit sub-texel fixed point: returns base index and alpha numerator/256
static inline void linearCoord(float u, int* i0, float* alpha) {
    const float t = std::floor((saneCoord(u) - 0.5f) * 256.0f + 0.5f);
    const int ti = (int)t;
    *i0 = ti >> 8; // arithmetic shift: floor for negatives
    *alpha = (float)(ti & 255) * (1.0f / 256.0f);
}

static inline vec4 texture2D(const Image& im, int filter, int addr, vec2 uv) {
    const float u = uv.x * (float)im.w;
    const float v = uv.y * (float)im.h;
    if (filter == NEAREST) {
        return addressedTexel(im, (int)std::floor(saneCoord(u)), (int)std::floor(saneCoord(v)), 0, addr);
    }
    int i0, j0; float a, b;
    linearCoord(u, &i0, &a);
    linearCoord(v, &j0, &b);
    const vec4 t00 = addressedTexel(im, i0, j0, 0, addr);
    const vec4 t10 = addressedTexel(im, i0 + 1, j0, 0, addr);
    const vec4 t01 = addressedTexel(im, i0, j0 + 1, 0, addr);
    const vec4 t11 = addressedTexel(im, i0 + 1, j0 + 1, 0, addr);
    const float w00 = (1.f - a) * (1.f - b), w10 = a * (1.f - b), w01 = (1.f - a) * b, w11 = a * b;
    return t00 * w00 + t10 * w10 + t01 * w01 + t11 * w11;
}

static inline vec4 texture3D(const Image& im, int filter, int addr, vec3 uvw) {
    const float u = uvw.x * (float)im.w;
    const float v = uvw.y * (float)im.h;
    const float w = uvw.z * (float)im.d;
    if (filter == NEAREST) {
        return addressedTexel(im, (int)std::floor(saneCoord(u)), (int)std::floor(saneCoord(v)), (int)std::floor(saneCoord(w)), addr);
    }
    int i0, j0, k0; float a, b, c;
    linearCoord(u, &i0, &a);
    linearCoord(v, &j0, &b);
    linearCoord(w, &k0, &c);
    vec4 r(0.f);
    for (int dz = 0; dz < 2; dz++)
        for (int dy = 0; dy < 2; dy++)
            for (int dx = 0; dx < 2; dx++) {
                const float wx = dx ? a : (1.f - a), wy = dy ? b : (1.f - b), wz = dz ? c : (1.f - c);
                r = r + addressedTexel(im, i0 + dx, j0 + dy, k0 + dz, addr) * ((wx * wy) * wz);
            }
    return r;
}

// textureGather component 0: (i0,j1), (i1,j1), (i1,j0), (i0,j0) with i0=floor(u-0.5), j0=floor(v-0.5)
static inline vec4 textureGatherR(const Image& im, int addr, vec2 uv) {
    int i0, j0; float a, b; // same 8-bit fixed-point coordinate as the bilinear footprint
    linearCoord(uv.x * (float)im.w, &i0, &a);
    linearCoord(uv.y * (float)im.h, &j0, &b);
    return vec4(addressedTexel(im, i0, j0 + 1, 0, addr).x, addressedTexel(im, i0 + 1, j0 + 1, 0, addr).x,
                addressedTexel(im, i0 + 1, j0, 0, addr).x, addressedTexel(im, i0, j0, 0, addr).x);
}

} // namespace orc
