// ORACLE (test infrastructure, never shipped, never on the product path).
// Deterministic single-precision math contract ("detmath").
//
// GLSL leaves the precision of log/exp/pow/sin/cos/acos/atan implementation
// defined, so "the reference's result" for e.g. a histogram bin
// (resources/shaders/histogramPerTile.comp:54-56) is only defined up to the
// driver's libm. This build fixes one software definition, built only from
// IEEE-754 binary32 + - * / sqrt (each correctly rounded, no FMA contraction),
// so that the CPU oracle and the HIP kernels produce identical bits. The HIP side
// carries the same algorithms in plainrenderer_amd/csrc/device/detmath.h;
// tests/test_detmath.py checks both against float64 libm (accuracy) and against
// each other on the GPU (bit identity).
//
// Compile with -ffp-contract=off and without -ffast-math.
#pragma once
#include <cmath>
#include <cstdint>
#include <cstring>

namespace orc {

static inline uint32_t f2u(float f) { uint32_t u; std::memcpy(&u, &f, 4); return u; }
static inline float u2f(uint32_t u) { float f; std::memcpy(&f, &u, 4); return f; }

// GLSL 4.60 spec 8.3: min(x,y) = y < x ? y : x ; max(x,y) = x < y ? y : x, "undefined" if an operand is NaN.
// NaN case fixed the way GPU hardware (v_min_f32 / v_max_f32, IEEE minNum / maxNum) resolves it: the non-NaN operand wins.
// The reference relies on this to recover from the 0/0 of its first frames (preExposeLights.comp:64,72).
static inline float gmin(float x, float y) { return (x != x) ? y : ((y != y) ? x : ((y < x) ? y : x)); }
static inline float gmax(float x, float y) { return (x != x) ? y : ((y != y) ? x : ((x < y) ? y : x)); }
static inline float gclamp(float x, float lo, float hi) { return gmin(gmax(x, lo), hi); }
static inline float gsign(float x) { return x > 0.f ? 1.f : (x < 0.f ? -1.f : 0.f); }
static inline float gfract(float x) { return x - std::floor(x); }
static inline float gmix(float a, float b, float t) { return a * (1.f - t) + b * t; }

static const float DET_LN2_HI = 0.693145751953125f;      // 0x3f317180
static const float DET_LN2_LO = 1.42860677e-06f;         // 0x35bfbe8e
static const float DET_LN2 = 0.693147182f;
static const float DET_INV_LN2 = 1.44269504f;
static const float DET_PI = 3.14159274f;
static const float DET_PIO2 = 1.57079637f;
static const float DET_PIO4 = 0.785398185f;

// returns log(m) for the reduced mantissa and the binary exponent through *e
static inline float det_log_reduced(float x, int* eOut) {
    uint32_t ix = f2u(x);
    int e = 0;
    if (ix < 0x00800000u) { // subnormal
        x = x * 8388608.0f;
        ix = f2u(x);
        e = -23;
    }
    e += (int)(ix >> 23) - 127;
    ix = (ix & 0x007fffffu) | 0x3f800000u;
    float m = u2f(ix);
    if (m > 1.41421354f) { m = m * 0.5f; e += 1; }
    const float f = m - 1.0f;
    const float s = f / (2.0f + f);
    const float z = s * s;
    float p = 0.222222224f;
    p = p * z + 0.285714298f;
    p = p * z + 0.400000006f;
    p = p * z + 0.666666687f;
    *eOut = e;
    return f - s * (f - z * p);
}

static inline float det_logf(float x) {
    if (x != x) return x;
    if (x < 0.f) return u2f(0x7fc00000u);
    if (x == 0.f) return u2f(0xff800000u);
    if (f2u(x) == 0x7f800000u) return x;
    int e;
    const float r = det_log_reduced(x, &e);
    const float fe = (float)e;
    return fe * DET_LN2_HI + (fe * DET_LN2_LO + r);
}

static inline float det_log2f(float x) {
    if (x != x) return x;
    if (x < 0.f) return u2f(0x7fc00000u);
    if (x == 0.f) return u2f(0xff800000u);
    if (f2u(x) == 0x7f800000u) return x;
    int e;
    const float r = det_log_reduced(x, &e);
    return (float)e + r * DET_INV_LN2;
}

// e^r for |r| <= ~0.35, then scaled by 2^k in two exact steps (subnormal safe)
static inline float det_exp_poly_scale(float r, int k) {
    float p = 1.98412701e-04f;        // 1/5040
    p = p * r + 1.38888892e-03f;      // 1/720
    p = p * r + 8.33333377e-03f;      // 1/120
    p = p * r + 4.16666679e-02f;      // 1/24
    p = p * r + 1.66666672e-01f;      // 1/6
    p = p * r + 0.5f;
    p = p * r + 1.0f;
    p = p * r + 1.0f;
    const int k1 = k / 2;
    const int k2 = k - k1;
    const float s1 = u2f((uint32_t)(k1 + 127) << 23);
    const float s2 = u2f((uint32_t)(k2 + 127) << 23);
    return (p * s1) * s2;
}

static inline float det_expf(float x) {
    if (x != x) return x;
    if (x > 88.7228394f) return u2f(0x7f800000u);
    if (x < -104.0f) return 0.f;
    const float fk = std::floor(x * DET_INV_LN2 + 0.5f);
    const float r = (x - fk * DET_LN2_HI) - fk * DET_LN2_LO;
    return det_exp_poly_scale(r, (int)fk);
}

static inline float det_exp2f(float x) {
    if (x != x) return x;
    if (x >= 128.0f) return u2f(0x7f800000u);
    if (x < -150.0f) return 0.f;
    const float fk = std::floor(x + 0.5f);
    const float r = (x - fk) * DET_LN2;
    return det_exp_poly_scale(r, (int)fk);
}

// GLSL pow(x,y) = exp2(y*log2(x)), "undefined" for x < 0. A negative base is treated as 0 here: the only negative bases
// the hot path produces are roundoff (1 - VoH with VoH = dot of two unit vectors = 1 + 1ulp, brdf.inc:35 via
// triangle.frag:315-317 when the irradiance SH is exactly zero), where hardware log2 would poison the frame with NaN.
static inline float det_powf(float x, float y) {
    if (x < 0.f) x = 0.f;
    if (x == 0.f) return (y > 0.f) ? 0.f : ((y == 0.f) ? 1.f : u2f(0x7f800000u));
    return det_exp2f(y * det_log2f(x));
}

static inline void det_sincosf(float x, float* sOut, float* cOut) {
    float ax = std::fabs(x);
    if (!(ax < 1.0e6f)) { // out of the reduction's range (also NaN/inf): defined as NaN
        *sOut = u2f(0x7fc00000u); *cOut = u2f(0x7fc00000u); return;
    }
    uint32_t j = (uint32_t)(ax * 1.27323954f); // 4/pi
    j += (j & 1u);
    const float y = (float)j;
    const float z = ((ax - y * 0.78515625f) - y * 2.4187564849853515625e-4f) - y * 3.77489497744594108e-8f;
    const float zz = z * z;
    const float sp = ((-1.9515295891e-4f * zz + 8.3321608736e-3f) * zz - 1.6666654611e-1f) * zz * z + z;
    const float cp = ((2.443315711809948e-5f * zz - 1.388731625493765e-3f) * zz + 4.166664568298827e-2f) * zz * zz - 0.5f * zz + 1.0f;
    const uint32_t q = (j >> 1) & 3u;
    float s, c;
    if (q == 0u) { s = sp; c = cp; }
    else if (q == 1u) { s = cp; c = -sp; }
    else if (q == 2u) { s = -sp; c = -cp; }
    else { s = -cp; c = sp; }
    if (x < 0.f) s = -s;
    *sOut = s; *cOut = c;
}
static inline float det_sinf(float x) { float s, c; det_sincosf(x, &s, &c); return s; }
static inline float det_cosf(float x) { float s, c; det_sincosf(x, &s, &c); return c; }

// asin on |x| <= 0.5 (odd polynomial)
static inline float det_asin_poly(float x) {
    const float z = x * x;
    float p = 4.2163199048e-2f;
    p = p * z + 2.4181311049e-2f;
    p = p * z + 4.5470025998e-2f;
    p = p * z + 7.4953002686e-2f;
    p = p * z + 1.6666752422e-1f;
    return p * z * x + x;
}

// acos with the argument clamped to [-1,1] (GLSL: undefined outside)
static inline float det_acosf(float x) {
    if (x != x) return x;
    x = gclamp(x, -1.f, 1.f);
    if (x < -0.5f) return DET_PI - 2.0f * det_asin_poly(std::sqrt(0.5f * (1.0f + x)));
    if (x > 0.5f) return 2.0f * det_asin_poly(std::sqrt(0.5f * (1.0f - x)));
    return DET_PIO2 - det_asin_poly(x);
}

static inline float det_atan_pos(float t) { // t >= 0
    float yy = 0.f;
    if (t > 2.41421366f) { yy = DET_PIO2; t = -(1.0f / t); }
    else if (t > 0.414213568f) { yy = DET_PIO4; t = (t - 1.0f) / (t + 1.0f); }
    const float z = t * t;
    float p = 8.05374449538e-2f;
    p = p * z - 1.38776856032e-1f;
    p = p * z + 1.99777106478e-1f;
    p = p * z - 3.33329491539e-1f;
    return yy + (p * z * t + t);
}

// GLSL atan(y, x)
static inline float det_atan2f(float y, float x) {
    if (x != x || y != y) return u2f(0x7fc00000u);
    if (x == 0.f) {
        if (y == 0.f) return 0.f;
        return y > 0.f ? DET_PIO2 : -DET_PIO2;
    }
    const float t = y / x;
    float a = det_atan_pos(std::fabs(t));
    if (t < 0.f) a = -a;
    if (x > 0.f) return a;
    return (y >= 0.f) ? a + DET_PI : a - DET_PI;
}

} // namespace orc
