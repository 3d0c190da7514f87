// Deterministic binary32 math for the HIP kernels.
//
// GLSL leaves log/exp/pow/sin/cos/acos/atan precision implementation defined, so results such
// as the histogram bin of resources/shaders/histogramPerTile.comp:54-56 are only defined up to
// the driver's libm. These routines fix one definition from IEEE + - * / sqrt (no FMA, the
// library is compiled with -ffp-contract=off) so kernel output is reproducible bit for bit on
// any IEEE machine. v_log_f32 / v_exp_f32 / v_sin_f32 are deliberately not used here.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace plr {

#define PLR_DI __device__ __forceinline__

PLR_DI uint32_t f2u(float f) { return __float_as_uint(f); }
PLR_DI float u2f(uint32_t u) { return __uint_as_float(u); }

// GLSL 4.60 8.3 definitions; a NaN operand loses (IEEE minNum / maxNum, what v_min_f32 / v_max_f32 do), signed zeros
// follow from the comparison. The reference relies on the NaN rule to recover from the 0/0 of its first frames.
#ifdef PLR_FAST_SET
// kernels_fast/: v_min_f32 / v_max_f32 (a NaN operand loses as well; only the sign of a zero result can differ from the comparison form,
// which the compiler otherwise turns into divergent branches in several kernels)
PLR_DI float gmin(float x, float y) { return __builtin_fminf(x, y); }
PLR_DI float gmax(float x, float y) { return __builtin_fmaxf(x, y); }
#else
// = (x != x) ? y : ((y != y) ? x : ((y < x) ? y : x)) for every operand pair (and likewise for the maximum): minNum / maxNum already return
// the non-NaN operand and the smaller / larger one; the comparison form additionally returns x when the operands compare equal (zeros of
// opposite sign), which the select restores. Written this way it is three instructions; the nested conditionals became divergent branches.
PLR_DI float gmin(float x, float y) { return x == y ? x : __builtin_fminf(x, y); }
PLR_DI float gmax(float x, float y) { return x == y ? x : __builtin_fmaxf(x, y); }
#endif
PLR_DI float gclamp(float x, float lo, float hi) { return gmin(gmax(x, lo), hi); }
PLR_DI float gsign(float x) { return x > 0.f ? 1.f : (x < 0.f ? -1.f : 0.f); }
PLR_DI float gmix(float a, float b, float t) { return a * (1.f - t) + b * t; }

#define PLR_LN2_HI 0.693145751953125f
#define PLR_LN2_LO 1.42860677e-06f
#define PLR_LN2 0.693147182f
#define PLR_INV_LN2 1.44269504f
#define PLR_PI_F 3.14159274f
#define PLR_PIO2_F 1.57079637f
#define PLR_PIO4_F 0.785398185f

PLR_DI float det_log_reduced(float x, int* eOut) {
    uint32_t ix = f2u(x);
    int e = 0;
    if (ix < 0x00800000u) {
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

// special operands override the general result in reverse priority (NaN -> NaN, negative -> NaN, +-0 -> -inf, +inf -> +inf): same values as
// early returns, without four divergent branches per call (the reduction is defined, if meaningless, for every bit pattern)
PLR_DI float det_log_special(float x, float general) {
    float r = f2u(x) == 0x7f800000u ? x : general;
    r = x == 0.f ? u2f(0xff800000u) : r;
    r = x < 0.f ? u2f(0x7fc00000u) : r;
    return x != x ? x : r;
}

PLR_DI float det_logf(float x) {
    int e;
    const float r = det_log_reduced(x, &e);
    const float fe = (float)e;
    return det_log_special(x, fe * PLR_LN2_HI + (fe * PLR_LN2_LO + r));
}

PLR_DI float det_log2f(float x) {
    int e;
    const float r = det_log_reduced(x, &e);
    return det_log_special(x, (float)e + r * PLR_INV_LN2);
}

PLR_DI float det_exp_poly_scale(float r, int k) {
    float p = 1.98412701e-04f;
    p = p * r + 1.38888892e-03f;
    p = p * r + 8.33333377e-03f;
    p = p * r + 4.16666679e-02f;
    p = p * r + 1.66666672e-01f;
    p = p * r + 0.5f;
    p = p * r + 1.0f;
    p = p * r + 1.0f;
    const int k1 = k / 2;
    const int k2 = k - k1;
    const float s1 = u2f((uint32_t)(k1 + 127) << 23);
    const float s2 = u2f((uint32_t)(k2 + 127) << 23);
    return (p * s1) * s2;
}

PLR_DI float det_expf(float x) {
    if (x != x) return x;
    if (x > 88.7228394f) return u2f(0x7f800000u);
    if (x < -104.0f) return 0.f;
    const float fk = floorf(x * PLR_INV_LN2 + 0.5f);
    const float r = (x - fk * PLR_LN2_HI) - fk * PLR_LN2_LO;
    return det_exp_poly_scale(r, (int)fk);
}

PLR_DI float det_exp2f(float x) {
    if (x != x) return x;
    if (x >= 128.0f) return u2f(0x7f800000u);
    if (x < -150.0f) return 0.f;
    const float fk = floorf(x + 0.5f);
    const float r = (x - fk) * PLR_LN2;
    return det_exp_poly_scale(r, (int)fk);
}

// pow(x, y) = exp2(y * log2(x)); GLSL leaves x < 0 undefined, a negative base (roundoff of 1 - cos) is treated as 0
PLR_DI float det_powf(float x, float y) {
    if (x < 0.f) x = 0.f;
    if (x == 0.f) return (y > 0.f) ? 0.f : ((y == 0.f) ? 1.f : u2f(0x7f800000u));
    return det_exp2f(y * det_log2f(x));
}

PLR_DI void det_sincosf(float x, float* sOut, float* cOut) {
    const float ax = fabsf(x);
    if (!(ax < 1.0e6f)) { *sOut = u2f(0x7fc00000u); *cOut = u2f(0x7fc00000u); return; }
    uint32_t j = (uint32_t)(ax * 1.27323954f);
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
PLR_DI float det_sinf(float x) { float s, c; det_sincosf(x, &s, &c); return s; }
PLR_DI float det_cosf(float x) { float s, c; det_sincosf(x, &s, &c); return c; }

PLR_DI float det_asin_poly(float x) {
    const float z = x * x;
    float p = 4.2163199048e-2f;
    p = p * z + 2.4181311049e-2f;
    p = p * z + 4.5470025998e-2f;
    p = p * z + 7.4953002686e-2f;
    p = p * z + 1.6666752422e-1f;
    return p * z * x + x;
}

PLR_DI float det_acosf(float x) {
    if (x != x) return x;
    x = gclamp(x, -1.f, 1.f);
    if (x < -0.5f) return PLR_PI_F - 2.0f * det_asin_poly(sqrtf(0.5f * (1.0f + x)));
    if (x > 0.5f) return 2.0f * det_asin_poly(sqrtf(0.5f * (1.0f - x)));
    return PLR_PIO2_F - det_asin_poly(x);
}

PLR_DI float det_atan_pos(float t) {
    float yy = 0.f;
    if (t > 2.41421366f) { yy = PLR_PIO2_F; t = -(1.0f / t); }
    else if (t > 0.414213568f) { yy = PLR_PIO4_F; t = (t - 1.0f) / (t + 1.0f); }
    const float z = t * t;
    float p = 8.05374449538e-2f;
    p = p * z - 1.38776856032e-1f;
    p = p * z + 1.99777106478e-1f;
    p = p * z - 3.33329491539e-1f;
    return yy + (p * z * t + t);
}

PLR_DI float det_atan2f(float y, float x) {
    if (x != x || y != y) return u2f(0x7fc00000u);
    if (x == 0.f) {
        if (y == 0.f) return 0.f;
        return y > 0.f ? PLR_PIO2_F : -PLR_PIO2_F;
    }
    const float t = y / x;
    float a = det_atan_pos(fabsf(t));
    if (t < 0.f) a = -a;
    if (x > 0.f) return a;
    return (y >= 0.f) ? a + PLR_PI_F : a - PLR_PI_F;
}

} // namespace plr
