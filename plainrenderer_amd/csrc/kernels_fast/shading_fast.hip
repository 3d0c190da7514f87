// PLR_MATH_FAST variant of the deferred shade (exact variant and the description of the re-expression: kernels_exact/shading.hip).
//
// Same lighting model term by term; the arithmetic is restructured for the VALU:
//  * (1-x)^5 Fresnel / CoD terms are x2*x2*x instead of exp2(5*log2(x)); the remaining pow/log/exp use v_log_f32 / v_exp_f32
//  * world position = camPos + (forward - tan*ndc.y*up + tan*aspect*ndc.x*right) * depthLinear (the normalisation of the view ray
//    cancels against the division by dot(ray, forward)); normalisations use v_rsq_f32, divisions v_rcp_f32, FMA contraction is on
//  * the 12 PCF tap directions are one hardware sin/cos of the per-pixel noise angle rotated by a constant 12-entry table
//  * directionToSH_L1 of a unit vector has a constant norm, so its normalize() is two constants
// Output is R11G11B10 (6/5-bit mantissas); stated tolerance in tests/test_fast_kernels.py.
//
// What is NOT relaxed: the geometry that feeds a discrete decision. The world position, the light-space position and the D16 comparison of
// a PCF tap are evaluated in the shader's operation order with correctly rounded quotients / roots (exactViewRay / exactSurface /
// calcShadow below): the synthetic - and any bias-free - shadow map holds the depth of the very surface being shaded, so a third of the
// lit pixels have taps within one D16 step of equality and a last-bit difference in the light-space depth flips them (round 2: 1.55 % of
// the 4K pixels resolved a tap differently from the oracle; now the tap count differs on < 1e-3 of them, profiles/r03_parity_4k.txt).
#include "../backend.h"
#include "../device/shading_common.h"
#include "../device/fastmath.h"
#include "../device/buffer_fetch.h"
#include "upscale_quad.h"
#include "pcf_taps.h"
#include <type_traits>
#include <map>

namespace plr {

namespace fastshade {

#ifndef PLR_SHADE_WAVES
#define PLR_SHADE_WAVES 5 // <= 96 VGPRs. Measured with the uniform blocks on scalar loads (PLR_SHADE_UNIFORM_PARAMS below), fused upscale + shade at 4K: 5 waves (90 VGPRs, no
                          // scratch) 216 us; 4 waves (the allocator then takes 106) 243 us; 6 waves (80 VGPRs, 9 spilled) 253 us
#endif

PLR_DI float rcpf(float x) { return __builtin_amdgcn_rcpf(x); }
PLR_DI float rsqf(float x) { return __builtin_amdgcn_rsqf(x); }
PLR_DI float sqrtv(float x) { return __builtin_amdgcn_sqrtf(x); } // v_sqrt_f32 (1 ulp) without the denormal-range rescue sequence of sqrtf()
// v_min / v_max / v_med3: a NaN operand loses, as in the software forms of detmath.h, in one instruction
PLR_DI float fmin1(float a, float b) { return __builtin_fminf(a, b); }
PLR_DI float fmax1(float a, float b) { return __builtin_fmaxf(a, b); }
PLR_DI float fclamp(float x, float lo, float hi) { return __builtin_fminf(__builtin_fmaxf(x, lo), hi); }
PLR_DI float fmix(float a, float b, float t) { return a + (b - a) * t; }
PLR_DI float log2h(float x) { return __builtin_amdgcn_logf(x); }   // v_log_f32 (base 2)
PLR_DI float exp2h(float x) { return __builtin_amdgcn_exp2f(x); }  // v_exp_f32 (base 2)
PLR_DI vec3 nrm(vec3 v) { return v * rsqf(dot(v, v)); }
PLR_DI float pow5(float x) { x = fmax1(x, 0.f); const float x2 = x * x; return x2 * x2 * x; }
PLR_DI float fpow(float x, float y) { return x <= 0.f ? 0.f : exp2h(y * log2h(x)); }

// ---- correctly rounded quotient / reciprocal / root from v_rcp_f32 / v_rsq_f32 (1 ulp) and one Newton step with fused residuals: the IEEE result
// for every operand this kernel sees (finite, normal, non-zero divisors) in 4-5 instructions instead of the 10-instruction v_div_scale sequence
PLR_DI float rcpRN(float d) { const float r = rcpf(d); return __builtin_fmaf(__builtin_fmaf(-d, r, 1.f), r, r); }
PLR_DI float divRNr(float n, float d, float r) { const float q = n * r; return __builtin_fmaf(__builtin_fmaf(-d, q, n), r, q); } // r = rcpf(d) (1 ulp is enough: the residual is exact)
PLR_DI float divRN(float n, float d) { return divRNr(n, d, rcpf(d)); } // the residual step corrects the raw reciprocal's ulp as well: no Newton step on r first
PLR_DI float sqrtRN(float x) { const float r = rsqf(x), s = x * r; return __builtin_fmaf(__builtin_fmaf(-s, s, x), 0.5f * r, s); } // x > 0

// view direction of a pixel (surface -> camera, unit length) and its screen uv, bit for bit as deferredShadingKernel (kernels/shading.hip)
// computes them: screenToWorld.inc:4-9 with every operation rounded separately. Scalar expressions on purpose: `#pragma clang fp contract`
// is lexical, the vec3 operators of vecmath.h would be contracted with the file's default
struct ViewRay { vec3 Vn; float su, sv; };
PLR_DI ViewRay exactViewRay(const GlobalUbo* __restrict__ g, int px, int py) {
#pragma clang fp contract(off)
    ViewRay o;
    o.su = divRN((float)px + 0.5f, (float)g->screenResolution[0]);
    o.sv = divRN((float)py + 0.5f, (float)g->screenResolution[1]);
    const float ndx = o.su * 2.f - 1.f, ndy = o.sv * 2.f - 1.f;
    const float ty = g->cameraTanFovHalf * ndy, tx = (g->cameraTanFovHalf * g->cameraAspectRatio) * ndx;
    const float vx = (-g->cameraForward[0] + ty * g->cameraUp[0]) - tx * g->cameraRight[0];
    const float vy = (-g->cameraForward[1] + ty * g->cameraUp[1]) - tx * g->cameraRight[1];
    const float vz = (-g->cameraForward[2] + ty * g->cameraUp[2]) - tx * g->cameraRight[2];
    const float inv = rcpRN(sqrtRN((vx * vx + vy * vy) + vz * vz));
    o.Vn = vec3(vx * inv, vy * inv, vz * inv);
    return o;
}
// world position of the surface and its view depth (deferredShadingKernel: linearizeDepth, passPos = camPos + Vcam / dot(Vcam, fwd) * depthLinear,
// pixelDepth = dot(camPos - passPos, -fwd)), same operations, same order
struct Surface { vec3 passPos; float pixelDepth; };
PLR_DI Surface exactSurface(const GlobalUbo* __restrict__ g, vec3 Vn, float depth) {
#pragma clang fp contract(off)
    const float cx = -Vn.x, cy = -Vn.y, cz = -Vn.z; // Vcam
    const float fx = g->cameraForward[0], fy = g->cameraForward[1], fz = g->cameraForward[2];
    const float dvf = (cx * fx + cy * fy) + cz * fz;
    const float den = g->farPlane + (-depth + 1.f) * (g->nearPlane - g->farPlane);
    const float depthLinear = divRN(g->nearPlane * g->farPlane, den);
    const float rd = rcpf(dvf);
    Surface s;
    s.passPos.x = g->cameraPosition[0] + divRNr(cx, dvf, rd) * depthLinear;
    s.passPos.y = g->cameraPosition[1] + divRNr(cy, dvf, rd) * depthLinear;
    s.passPos.z = g->cameraPosition[2] + divRNr(cz, dvf, rd) * depthLinear;
    const float wx = g->cameraPosition[0] - s.passPos.x, wy = g->cameraPosition[1] - s.passPos.y, wz = g->cameraPosition[2] - s.passPos.z;
    s.pixelDepth = (wx * -fx + wy * -fy) + wz * -fz;
    return s;
}

PLR_DI float D_GGX(float NoH, float r) {
    const float a = NoH * r;
    const float k = r * rcpf(1.0f - NoH * NoH + a * a);
    return k * k * (1.0f / PLR_GLSL_PI);
}
PLR_DI float Visibility(float NoV, float NoL, float r) {
    const float r_2 = r * r;
    const float v1 = NoL * sqrtv(NoV * NoV * (1.f - r_2) + r_2);
    const float v2 = NoV * sqrtv(NoL * NoL * (1.f - r_2) + r_2);
    return 0.5f * rcpf(v1 + v2);
}
PLR_DI vec3 F_Schlick(vec3 f0, vec3 f90, float VoH) { return f0 + (f90 - f0) * pow5(1.f - VoH); }
PLR_DI vec3 DisneyDiffuse(vec3 diffuseColor, float NoL, float VoH, float NoV, float r) {
    const float energyBias = 0.5f * r;
    const float energyFactor = fmix(1.f, 1.f / 1.51f, r);
    const float f90 = energyBias + 2.f * VoH * VoH * r;
    const float fl = 1.f + (f90 - 1.f) * pow5(1.f - NoL), fv = 1.f + (f90 - 1.f) * pow5(1.f - NoV);
    return diffuseColor * ((1.f / PLR_GLSL_PI) * fl * fv * energyFactor);
}
PLR_DI vec3 CoDWWIIDiffuse(vec3 diffuseColor, float NoL, float VoH, float NoV, float NoH, float r) {
    const float f0Diffuse = VoH + pow5(1.f - VoH);
    const float f1 = (1.f - 0.75f * pow5(1.f - NoL)) * (1.f - 0.75f * pow5(1.f - NoV));
    const float g = log2h(2.f * rcpf(r * r) - 1.f) * (1.f / 18.f);
    const float t = fclamp(2.2f * g - 0.5f, 0.f, 1.f);
    const float fd = f0Diffuse + (f1 - f0Diffuse) * t;
    const float fb = (34.5f * g * g - 59.f * g + 24.5f) * VoH * exp2h(-fmax1(73.2f * g - 21.2f, 8.9f) * sqrtv(NoH));
    return diffuseColor * ((1.f / PLR_GLSL_PI) * (fd + fb));
}
PLR_DI float Titanfall2DiffuseSingleComponent(float NoL, float LoV, float NoV, float NoH, float r) {
    const float facing = 0.5f + 0.5f * LoV;
    const float rough = facing * (0.9f - 0.4f * facing) * (0.5f + NoH) * rcpf(fmax1(NoH, 0.03f));
    const float smoothDiffuse = 1.05f * (1.f - pow5(1.f - NoL)) * (1.f - pow5(1.f - NoV));
    return (1.f / PLR_GLSL_PI) * fmix(smoothDiffuse, rough, r);
}
PLR_DI vec3 GGXSingleScattering(float r, vec3 f0, float NoH, float NoV, float VoH, float NoL) {
    return (D_GGX(NoH, r) * Visibility(NoV, NoL, r)) * F_Schlick(f0, vec3(1.f), VoH);
}
PLR_DI float ReflectedEnergyAverage(float roughness) {
    const float smoothness = 1.f - sqrtv(roughness);
    float r = -0.0761947f - 0.383026f * smoothness;
    r = 1.04997f + smoothness * r;
    r = 0.409255f + smoothness * r;
    return fmin1(0.999f, r);
}
// both branches evaluated and selected: a divergent branch around the power costs more than the power (two quarter-rate instructions)
PLR_DI float sRGBToLinear1(float c) {
    const float lin = c * (1.f / 12.92f), pw = fpow((c + 0.055f) * (1.f / 1.055f), 2.4f);
    return c <= 0.004045f ? lin : pw;
}

// one RGBA16F texel's four channels times a weight, accumulated: the halves go straight into v_fma_mix_f32 (fp16 operands are converted inside the
// instruction: no v_cvt_f32_f16 per channel, no shift for the upper half - the operand select does it)
PLR_DI _Float16 halfLo(uint32_t w) { return __builtin_bit_cast(_Float16, (uint16_t)(w & 0xffffu)); }
PLR_DI _Float16 halfHi(uint32_t w) { return __builtin_bit_cast(_Float16, (uint16_t)(w >> 16)); }
PLR_DI vec4 accumulateTexel(vec4 acc, uint32_t xy, uint32_t zw, float w) {
    return vec4(__builtin_fmaf((float)halfLo(xy), w, acc.x), __builtin_fmaf((float)halfHi(xy), w, acc.y), __builtin_fmaf((float)halfLo(zw), w, acc.z),
                __builtin_fmaf((float)halfHi(zw), w, acc.w));
}
// Column pair of a clamp-to-edge linear fetch: texels x0 = clamp(i0), x1 = clamp(i0 + 1) always lie in the pair (xb, xb + 1) with xb = clamp(i0, 0, w - 2),
// which one 16-byte load fetches (a load instruction costs the texture addresser the same whatever its width). Inside the image the pair IS
// (x0, x1); at the left edge (i0 < 0) both are texel 0 = the pair's first -> weight 0; at the right edge (i0 >= w - 1) both are texel w - 1 = the
// pair's second -> weight 1. One med3 on the weight replaces a select per fetched dword (round 3: 8 selects per LUT fetch, 32 per froxel fetch).
PLR_DI void edgePair(int i0, float a, int w, int* xb, float* a2) {
    *xb = clampTo(i0, w - 2);
    *a2 = __builtin_amdgcn_fmed3f(a + (float)(i0 - *xb), 0.f, 1.f);
}

// bilinear RGBA16F fetch with clamp-to-edge, weights as the sampler contract states them (image.h sampleLinear2D): w00 = (1-a)(1-b), ...
PLR_DI vec4 bilinearLut(const ImgView& im, float u, float v) {
    int i0, j0; float a, b;
    linearCoord(u * (float)im.w, &i0, &a);
    linearCoord(v * (float)im.h, &j0, &b);
    int xb; float a2;
    edgePair(i0, a, im.w, &xb, &a2); // the launcher sends images narrower than two texels to the general kernel
    const int y0 = clampi(j0, im.h), y1 = clampi(j0 + 1, im.h);
    const BufferDesc texels = texelBuffer(im.ptr, 8u); // (device/buffer_fetch.h: the pair's index goes to the addresser as it is)
    const uint4 q0 = fetch128(texels, __umul24((uint32_t)y0, (uint32_t)im.w) + (uint32_t)xb), q1 = fetch128(texels, __umul24((uint32_t)y1, (uint32_t)im.w) + (uint32_t)xb);
    const float a0 = 1.f - a2, b0 = 1.f - b;
    vec4 r(0.f);
    r = accumulateTexel(r, q0.x, q0.y, a0 * b0);
    r = accumulateTexel(r, q0.z, q0.w, a2 * b0);
    r = accumulateTexel(r, q1.x, q1.y, a0 * b);
    r = accumulateTexel(r, q1.z, q1.w, a2 * b);
    return r;
}

struct ShadeParams {
    ImgView color, depth, normal, albedo, specular, brdfLut, shadowMaps[4], ysh, cocg, volumetricLut, skyLut;
    const LightBuffer* light;
    const ShadowCascadeInfo* shadowInfo;
    const VolumetricLightingSettings* vol;
    const GlobalUbo* g;
    const ImgView* bindless;
    uint32_t bindlessCount;
    uint32_t cascadeCount;
    int coverW, coverH, yBase;
    int xBase; // columns [xBase, coverW) (tile rendering: PassCtx::colSpan; a multiple of 8), rows [yBase, coverH)
    uint32_t* sig; // decision signatures (plr_debug_set_decision_signature) or null
    ImgView noiseTex; // the frame's noise texture, resolved on the host (PassCtx::hostNoiseView)
    const float4* pcfTaps; // [256][6] float4 = [noise byte][tap] {x, y}: unit-disc tap offsets (pcf_taps.h)
    // the same rows re-indexed by the POSITION of the frame's noise texel and transposed, [6][pcfPositions] float4, or null (shadeDerivedTables below)
    const float4* pcfTapsByPosition;
    uint32_t pcfPositions;
    const uint2* lutEnergyFootprint; // [(h - 1)][(w - 1)] of the BRDF LUT: .y of the 2 x 2 texels at (x, y), or null (lutEnergy below)
};

// The four uniform blocks a shade kernel reads are ALSO passed as top-level `const T* __restrict__` kernel arguments, and the kernel puts those into
// its copy of ShadeParams. A pointer inside a by-value struct carries no aliasing information, so every load of a uniform field was a VECTOR
// load (a register per dword and lane for a value all lanes share); through a noalias argument the compiler proves the memory is not written by
// the kernel and issues scalar loads into SGPRs.
#define PLR_SHADE_UNIFORM_PARAMS const GlobalUbo* __restrict__ gUniform, const LightBuffer* __restrict__ lightUniform, const ShadowCascadeInfo* __restrict__ shadowUniform, \
                                 const VolumetricLightingSettings* __restrict__ volUniform
#define PLR_SHADE_ADOPT_UNIFORMS(P) do { (P).g = gUniform; (P).light = lightUniform; (P).shadowInfo = shadowUniform; (P).vol = volUniform; } while (0)
#define PLR_SHADE_UNIFORM_ARGS(P) (P).g, (P).light, (P).shadowInfo, (P).vol

// 12-tap rotated-disc PCF (calcShadow, triangle.frag:92-120). Per tap the shader evaluates
//   d = sqrt((i + noise/2) / 12), angle = 2 pi (i / 12 + noise), uv = base + (cos, sin)(angle) * 0.03 * lightSpaceScale * d,
//   nearest fetch with a black border, shadow += actualDepth >= texel.
// The light-space position (matrix product, perspective division) is the shader's, operation by operation (see the file header), and so
// is the comparison: a D16 texel t decodes to fl(t / 65535) (IEEE quotient), so "actualDepth >= texel" is "t <= T" with
// T = max{t : fl(t / 65535) <= actualDepth}, found once per pixel from the guess floor(actualDepth * 65535) (off by at most one) and two
// correctly rounded quotients.
// Round 4, the taps (the kernel is bound by VALU issue; 12 taps were a quarter of its instructions):
//  * (cos, sin)(angle) d comes from the tap table (pcf_taps.h: the oracle's arithmetic for the 256 noise values): a tap position is two fused
//    multiply-adds instead of a square root, four products and its share of a hardware sine / cosine and their rotation;
//  * the cascade - shadow map, light matrix, scale - is WAVE-UNIFORM here (the caller loops over the cascades present in its wave, almost always
//    one): matrix and scale sit in scalar registers instead of sixteen gathered vector registers, and the map is read through a buffer
//    resource whose range check implements the black border for the rows (an offset outside [0, 2 w h) returns 0 = "lit"); columns outside
//    [0, w) get the offset ~0. No index clamps, one compare for the border instead of two and a select, 32-bit offsets.
// `taps` = the pixel's row of the tap table (two taps per float4); m = lightMatrices[cascade] (uniform).
struct PcfTapRow { float4 q[kPcfTaps / 2]; };
PLR_DI float calcShadow(vec3 pos, const ImgView& shadowMap, const float* __restrict__ m, float lssX, float lssY, const PcfTapRow& taps) {
    float cxy0, cxy1; // tap centre, uv
    uint32_t depthThreshold;
    {
#pragma clang fp contract(off)
        const float x = ((m[0] * pos.x + m[4] * pos.y) + m[8] * pos.z) + m[12];
        const float y = ((m[1] * pos.x + m[5] * pos.y) + m[9] * pos.z) + m[13];
        const float z = ((m[2] * pos.x + m[6] * pos.y) + m[10] * pos.z) + m[14];
        const float w = ((m[3] * pos.x + m[7] * pos.y) + m[11] * pos.z) + m[15];
        const float rw = rcpf(w);
        cxy0 = divRNr(x, w, rw) * 0.5f + 0.5f;
        cxy1 = divRNr(y, w, rw) * 0.5f + 0.5f;
        const float actualDepth = fclamp(divRNr(z, w, rw), 0.f, 1.f);
        const uint32_t t0 = (uint32_t)(actualDepth * 65535.f); // truncation = floor, the value is non-negative
        const float r16 = 1.f / 65535.f; // the compiler's constant is the correctly rounded reciprocal
        const float q1 = divRNr((float)(t0 + 1u), 65535.f, r16), q0 = divRNr((float)t0, 65535.f, r16);
        depthThreshold = q1 <= actualDepth ? t0 + 1u : (q0 <= actualDepth ? t0 : t0 - 1u); // t0 = 0 always passes its own test: no wrap
    }
    const int w = shadowMap.w, h = shadowMap.h; // uniform
    const float fw = (float)w, fh = (float)h;
    // tap centre in texels, kept inside +-1e6 (a NaN becomes -1e6): the texel coordinates then fit the 24-bit multiply below, and a centre that
    // far outside the map has all its taps on the border either way
    const float bxw = fastm::clampCoord(cxy0 * fw), byh = fastm::clampCoord(cxy1 * fh);
    const float sxw = 0.03f * lssX * fw, syh = 0.03f * lssY * fh; // uniform
    const __amdgpu_buffer_rsrc_t map = __builtin_amdgcn_make_buffer_rsrc(shadowMap.ptr, 0, w * h * 2, 0x00020000); // raw buffer, 32-bit data format
    uint32_t lit = 0u;
    auto tap = [&](float ox, float oy) {
        const int xi = floorToInt(__builtin_fmaf(ox, sxw, bxw)), yi = floorToInt(__builtin_fmaf(oy, syh, byh));
        const int t = __mul24(yi, w) + xi; // v_mad_i32_i24; rows outside [0, h) give offsets outside [0, 2 w h): the range check returns 0
        const uint32_t off = (uint32_t)xi < (uint32_t)w ? (uint32_t)(t + t) : 0xffffffffu;
        const uint32_t texel = __builtin_amdgcn_raw_buffer_load_b16(map, (int)off, 0, 0);
        lit += texel <= depthThreshold ? 1u : 0u;
    };
#pragma unroll
    for (int k = 0; k < kPcfTaps / 2; k++) {
        const float4 q = taps.q[k];
        tap(q.x, q.y);
        tap(q.z, q.w);
    }
    return (float)lit * (1.f / 12.f);
}

// BRDF LUT, .y (the reflected energy) at (u, v): the multiscattering lobe looks it up twice per pixel. From the LUT itself that is two 16-byte loads
// (two rows) of which 4 of 32 bytes are used; the footprint copy (shadeDerivedTables) holds, per texel pair (xb, yb), the four .y halves of the 2 x 2
// footprint in 8 bytes: ONE load, and the y edge goes into the weight like the x edge (edgePair). Same texels, same weights.
PLR_DI float lutEnergy(const ShadeParams& P, float u, float v) {
    const ImgView& im = P.brdfLut;
    if (!P.lutEnergyFootprint) return bilinearLut(im, u, v).y; // uniform
    int i0, j0; float a, b;
    linearCoord(u * (float)im.w, &i0, &a);
    linearCoord(v * (float)im.h, &j0, &b);
    int xb, yb; float a2, b2;
    edgePair(i0, a, im.w, &xb, &a2);
    edgePair(j0, b, im.h, &yb, &b2);
    const uint2 q = P.lutEnergyFootprint[__umul24((uint32_t)yb, (uint32_t)(im.w - 1)) + (uint32_t)xb];
    const float a0 = 1.f - a2, b0 = 1.f - b2;
    float e = (float)halfLo(q.x) * (a0 * b0);
    e = __builtin_fmaf((float)halfHi(q.x), a2 * b0, e);
    e = __builtin_fmaf((float)halfLo(q.y), a0 * b2, e);
    e = __builtin_fmaf((float)halfHi(q.y), a2 * b2, e);
    return e;
}

template <int MULTISCATTER>
PLR_DI vec3 specularMultiscatteringLobe(const ShadeParams& P, float r, float NoL, vec3 f0, vec3 singleScatteringLobe, vec3 brdfLut) {
    const float energyOutgoing = brdfLut.y;
    const vec3 fresnelAverage = f0 + (1.f - f0) * (1.f / 21.f);
    if (MULTISCATTER == 0) {
        const float energyAverage = ReflectedEnergyAverage(r);
        const float energyIncoming = lutEnergy(P, r, NoL);
        const float unscaled = (1.f - energyIncoming) * (1.f - energyOutgoing) * rcpf(3.1415f * (1.f - energyAverage));
        const vec3 den = 1.f - fresnelAverage * (1.f - energyAverage);
        const vec3 scaling = (fresnelAverage * fresnelAverage * energyAverage) * vec3(rcpf(den.x), rcpf(den.y), rcpf(den.z));
        return unscaled * scaling;
    } else if (MULTISCATTER == 1) {
        const float lobe = (1.f - energyOutgoing) * (1.f / PLR_GLSL_PI);
        const vec3 den = 1.f - fresnelAverage * (1.f - energyOutgoing);
        return (fresnelAverage * fresnelAverage * (energyOutgoing * lobe)) * vec3(rcpf(den.x), rcpf(den.y), rcpf(den.z));
    } else if (MULTISCATTER == 2) {
        return f0 * (rcpf(energyOutgoing) - 1.f) * singleScatteringLobe;
    }
    return vec3(0.f);
}

// raw texels one pixel's shade reads from the G-buffer and from the (upscaled) indirect light
struct PixelInputs {
    float depth;
    uint32_t albedo, specular;
    uint32_t normal, normalH, normalV; // the pixel's own world normal and those of its horizontal / vertical partner in the 2x2 quad (GeometricAA.inc)
    uint2 ysh;                         // indirectDiffuse_Y_SH texel (RGBA16F)
    uint32_t cocg;                     // indirectDiffuse_CoCg texel (RG16F)
};
// normalize() of the decoded G-buffer normal; the all-zero texel (the only one whose normalisation is NaN: the exact kernel then keeps the raw
// vector) stays zero through the clamp of the squared length - one v_max instead of three compares and three selects
PLR_DI vec3 decodeNormal(uint32_t texel) {
    const vec3 raw = fastm::unorm8x4(texel).xyz() * 2.f - 1.f;
    return raw * rsqf(fmax1(dot(raw, raw), 1e-30f));
}

// ---- one geometry pixel (depth != 0) of the deferred shade, in three parts (round 6: the direct lighting can run as a launch of its own, see shadeDirectKernel):
//  shadeDirect   triangle.frag:146-290 - material, view / half vectors, cascade select + PCF, diffuse + GGX + multiscattering lobe of the sun. Needs no GI.
//  shadeIndirect triangle.frag:294-334 - irradiance from the L1 SH, its diffuse and specular response
//  froxelLookup  volumetricLighting.inc applyVolumetricLighting - the froxel LUT's in-scattering and transmittance at the pixel
// what the indirect half needs to know about the surface
struct SurfaceTerms {
    vec3 N, V, f0, diffuseColor, diffuseBRDFIntegral;
    float r, NoV, energyOutgoing; // r: after geometric AA; energyOutgoing = brdfLut.y at (r, NoV)
    vec3 brdfLut;                 // the LUT texel (INDIRECT_TECH 1 reads .x)
};
struct DirectOut { vec3 direct; float pixelDepth; vec2 noiseTexel; }; // direct = diffuseDirect + specularDirect for a sun of colour `lightColor`
template <int DIFFUSE_BRDF, int MULTISCATTER, bool GEOMETRIC_AA>
PLR_DI DirectOut shadeDirect(const ShadeParams& P, int px, int py, const ViewRay& vr, const PixelInputs& in, vec3 lightColor, SurfaceTerms* st, uint32_t* sigWord) {
    const GlobalUbo* g = P.g;
    const Surface surf = exactSurface(g, vr.Vn, in.depth);
    const vec3 passPos = surf.passPos;
    const float pixelDepth = surf.pixelDepth;

    const vec3 albedoTexel = fastm::unorm8x4(in.albedo).xyz();
    const vec3 specularTexel = fastm::unorm8x4(in.specular).xyz();
    const float metalic = specularTexel.z;
    float r = specularTexel.y;
    r = fmax1(r * r, 0.0045f);
    const vec3 albedo(sRGBToLinear1(albedoTexel.x), sRGBToLinear1(albedoTexel.y), sRGBToLinear1(albedoTexel.z));
    const vec3 diffuseColor = (1.f - metalic) * albedo;
    const vec3 N = decodeNormal(in.normal);
    const vec3 L = nrm(ld3(g->sunDirection));
    const vec3 V = vr.Vn; // normalize(camPos - passPos) is the view ray itself
    const vec3 H = nrm(V + L);
    if (GEOMETRIC_AA) {
        const vec3 N_U = decodeNormal(in.normalH) - N, N_V = decodeNormal(in.normalV) - N; // sign is irrelevant: only squared lengths are used
        const float variance = 0.25f * (dot(N_V, N_V) + dot(N_U, N_U));
        const float kernelRoughness2 = fmin1(2.f * variance, 0.18f);
        r = fclamp(sqrtv(r * r + kernelRoughness2), 0.f, 1.f);
    }
    const float NoH = fmax1(dot(N, H), 0.f);
    const float NdotL = dot(N, L);
    const float NoL = fclamp(NdotL, 0.f, 1.f);
    const float VoH = fabsf(dot(V, H));
    const float LoV = fmax1(dot(L, V), 0.f);
    const float NoV = fmax1(fabsf(dot(N, V)), 0.0001f);
    const vec3 f0 = vmix(vec3(0.04f), albedo, metalic);

    // frame index -> texture index -> view -> texel is four dependent round trips in front of the PCF taps when the kernel chases them itself
    const ImgView noiseTex = P.noiseTex;
    const uint32_t noiseIndex = fastm::texelIndex((uint32_t)fastm::repeatIndex(px, noiseTex.w), (uint32_t)fastm::repeatIndex(py, noiseTex.h), (uint32_t)noiseTex.w);
    const uint32_t noiseWord = fetch16(texelBuffer(noiseTex.ptr, 2u), noiseIndex);
    const vec2 noiseTexel = fastm::unorm8x2(noiseWord);
    const uint32_t noiseByte = noiseWord & 0xffu;

    int cascadeIndex = 0;
#pragma unroll
    for (int cascade = 0; cascade < 3; cascade++) cascadeIndex += (cascade < (int)P.cascadeCount - 1 && pixelDepth >= P.shadowInfo->splits[cascade]) ? 1 : 0;
    // The PCF runs once per cascade PRESENT in the wave (the cascade is a function of the view depth: a 64-pixel row segment almost always has one),
    // with the cascade in a scalar register: see calcShadow
    // The pixel's twelve tap offsets. Indexed by the noise VALUE, the 64 lanes of a wave read 64 scattered 96-byte rows: six loads of ~55 cache lines
    // each, a third of all the L1 accesses of a kernel that the ablation of round 4 showed to be bound by exactly those (profiles/r04_shade_ablation.txt).
    // Indexed by the noise texel's POSITION and transposed, lane l reads 16 bytes next to lane l - 1's: four cache lines per load.
    // One code path for both tables: a uniform base and row stride (scalar registers) and one 32-bit byte offset per lane.
    PcfTapRow tapRow;
    {
        const bool byPosition = P.pcfTapsByPosition != nullptr; // uniform
        const uint8_t* base = byPosition ? (const uint8_t*)P.pcfTapsByPosition : (const uint8_t*)P.pcfTaps;
        const uint32_t rowStride = byPosition ? P.pcfPositions * 16u : 16u;
        const uint32_t laneOffset = byPosition ? noiseIndex * 16u : noiseByte * (uint32_t)(kPcfTaps / 2 * 16);
        // raw buffer loads: the lane's byte offset in a VGPR, the row's in an SGPR - no 64-bit address per row on the VALU (device/buffer_fetch.h)
        const __amdgpu_buffer_rsrc_t table = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, 0x7fffffff, 0x00020000);
#pragma unroll
        for (int k = 0; k < kPcfTaps / 2; k++) {
            const auto q = __builtin_amdgcn_raw_buffer_load_b128(table, (int)laneOffset, (int)((uint32_t)k * rowStride), 0);
            __builtin_memcpy(&tapRow.q[k], &q, 16);
        }
    }
    float sunShadow = 0.f;
    for (bool pending = true; pending;) {
        int c = __builtin_amdgcn_readfirstlane(cascadeIndex);
        const bool mine = cascadeIndex == c;
        // the optimiser would otherwise replace c by the (per-lane) cascadeIndex inside the branch it guards - equal there, but in a vector register
        asm volatile("" : "+s"(c));
        if (mine) {
            const ImgView& m0 = P.shadowMaps[0];
            ImgView shadowMap; // uniform selects (s_cselect): the struct is a kernel argument, a dynamic index would copy it to scratch memory
            shadowMap.ptr = c == 0 ? m0.ptr : c == 1 ? P.shadowMaps[1].ptr : c == 2 ? P.shadowMaps[2].ptr : P.shadowMaps[3].ptr;
            shadowMap.w = c == 0 ? m0.w : c == 1 ? P.shadowMaps[1].w : c == 2 ? P.shadowMaps[2].w : P.shadowMaps[3].w;
            shadowMap.h = c == 0 ? m0.h : c == 1 ? P.shadowMaps[1].h : c == 2 ? P.shadowMaps[2].h : P.shadowMaps[3].h;
            shadowMap.d = 1; shadowMap.fmt = F_D16;
            sunShadow = calcShadow(passPos, shadowMap, P.shadowInfo->lightMatrices[c], P.shadowInfo->lightSpaceScale[c][0], P.shadowInfo->lightSpaceScale[c][1], tapRow);
            pending = false;
        }
    }
    *sigWord = (uint32_t)cascadeIndex | ((uint32_t)(sunShadow * 12.f + 0.5f) << 2) | 64u; // cascade, lit PCF taps, geometry (oracle/oracle.h)
    const vec3 directLighting = (fmax1(NdotL, 0.f) * sunShadow) * lightColor;
    const vec3 brdfLut = bilinearLut(P.brdfLut, r, NoV).xyz();

    vec3 diffuseDirect;
    vec3 diffuseBRDFIntegral(brdfLut.z);
    if (DIFFUSE_BRDF == 0) diffuseDirect = diffuseColor * (1.f / PLR_GLSL_PI) * directLighting;
    else if (DIFFUSE_BRDF == 1) diffuseDirect = DisneyDiffuse(diffuseColor, NoL, VoH, NoV, r) * directLighting;
    else if (DIFFUSE_BRDF == 2) diffuseDirect = CoDWWIIDiffuse(diffuseColor, NoL, VoH, NoV, NoH, r) * directLighting;
    else {
        const float single = Titanfall2DiffuseSingleComponent(NoL, LoV, NoV, NoH, r);
        diffuseDirect = diffuseColor * (single + diffuseColor * (0.1159f * r)) * directLighting;
        float multiIntegral = 0.1159f * r * PLR_GLSL_PI * 2.f;
        multiIntegral *= (1.f - (0.04f + 0.96f * pow5(1.f - NoV)));
        multiIntegral *= 0.94291f;
        diffuseBRDFIntegral = vmin(vec3(brdfLut.z) + diffuseColor * multiIntegral, vec3(1.f));
    }
    diffuseDirect = diffuseDirect * ((1.f - F_Schlick(f0, vec3(1.f), NoV)) * (1.f - F_Schlick(f0, vec3(1.f), NoL)));

    const vec3 singleScatteringLobe = GGXSingleScattering(r, f0, NoH, NoV, VoH, NoL);
    const vec3 multiScatteringLobe = specularMultiscatteringLobe<MULTISCATTER>(P, r, NoL, f0, singleScatteringLobe, brdfLut);
    const vec3 specularDirect = directLighting * (singleScatteringLobe + multiScatteringLobe);

    st->N = N; st->V = V; st->f0 = f0; st->diffuseColor = diffuseColor; st->diffuseBRDFIntegral = diffuseBRDFIntegral;
    st->r = r; st->NoV = NoV; st->energyOutgoing = brdfLut.y; st->brdfLut = brdfLut;
    DirectOut o;
    o.direct = diffuseDirect + specularDirect;
    o.pixelDepth = pixelDepth;
    o.noiseTexel = noiseTexel;
    return o;
}

// the response to the traced irradiance (triangle.frag:294-334); ysh / cocg: the (upscaled) indirectDiffuse_Y_SH / _CoCg texels
template <int MULTISCATTER>
PLR_DI vec3 shadeIndirect(const ShadeParams& P, const SurfaceTerms& st, uint2 ysh, uint32_t cocg) {
    const vec3 N = st.N, V = st.V, f0 = st.f0;
    const float r = st.r, NoV = st.NoV;
    const vec4 irradiance_Y_SH(halfBitsToFloat(ysh.x & 0xffffu), halfBitsToFloat(ysh.x >> 16), halfBitsToFloat(ysh.y & 0xffffu), halfBitsToFloat(ysh.y >> 16));
    const vec2 cc(halfBitsToFloat(cocg & 0xffffu), halfBitsToFloat(cocg >> 16));
    // directionToSH_L1(N) for unit N: normalize((0.28209, -0.48860 N.y, 0.48860 N.z, -0.48860 N.x)) = (0.5, -0.86603 N.y, ...)
    const vec4 shN(0.5f, -0.8660254f * N.y, 0.8660254f * N.z, -0.8660254f * N.x);
    const float irradiance_Y = dot(irradiance_Y_SH, shN);
    const vec3 irradiance = YCoCgToLinear(vec3(irradiance_Y, cc.x, cc.y));
    const vec3 diffuseIndirect = irradiance * st.diffuseColor * st.diffuseBRDFIntegral;
    const vec3 dominantDirection = dominantDirectionFromSH_L1(irradiance_Y_SH);
    const float dominantDirectionLength = fclamp(sqrtv(dot(dominantDirection, dominantDirection)), 0.01f, 1.f);
    const float r_indirect = fmix(1.f, r, sqrtv(dominantDirectionLength));
    const vec3 L_indirect = dominantDirection * rcpf(dominantDirectionLength);
    const vec3 H_indirect = nrm(L_indirect + V);
    const float NoH_indirect = fmax1(dot(N, H_indirect), 0.f);
    const float NoL_indirect = fmax1(dot(N, L_indirect), 0.f);
    const float VoH_indirect = fmax1(dot(V, H_indirect), 0.f);
    const vec3 single_i = GGXSingleScattering(r_indirect, f0, NoH_indirect, NoV, VoH_indirect, NoL_indirect);
    const vec3 multi_i = specularMultiscatteringLobe<MULTISCATTER>(P, r_indirect, NoL_indirect, f0, single_i, vec3(0.f, st.energyOutgoing, 0.f));
    const vec3 specularIndirect = (single_i + multi_i) * YCoCgToLinear(vec3(irradiance_Y_SH.x, cc.x, cc.y));
    return diffuseIndirect + specularIndirect;
}

// applyVolumetricLighting's LUT sample: .xyz in-scattering, .w transmittance (froxel z = log(linear * (e^3 - 1) + 1) / 3)
PLR_DI vec4 froxelLookup(const ShadeParams& P, float su, float sv, vec2 noiseTexel, float pixelDepth) {
    const float nu = su + (noiseTexel.x - 0.5f) * 0.013f, nv = sv + (noiseTexel.y - 0.5f) * 0.013f;
    const float linear = pixelDepth * rcpf(P.vol->maxDistance);
    const float z = log2h(linear * 19.0855369f + 1.f) * (0.693147181f / 3.f);
    const ImgView& vol = P.volumetricLut;
    int i0, j0, k0; float a, b, c;
    linearCoord(nu * (float)vol.w, &i0, &a);
    linearCoord(nv * (float)vol.h, &j0, &b);
    linearCoord(z * (float)vol.d, &k0, &c);
    int xb; float a2;
    edgePair(i0, a, vol.w, &xb, &a2); // the texel pair of every row in one 16-byte load, the x edge in the weight (bilinearLut above)
    const uint32_t y0 = __umul24((uint32_t)clampi(j0, vol.h), (uint32_t)vol.w), y1 = __umul24((uint32_t)clampi(j0 + 1, vol.h), (uint32_t)vol.w);
    const uint32_t sl = (uint32_t)vol.w * (uint32_t)vol.h; // uniform
    const uint32_t z0 = __umul24((uint32_t)clampi(k0, vol.d), sl), z1 = __umul24((uint32_t)clampi(k0 + 1, vol.d), sl); // slices below 2^24 texels
    const BufferDesc texels = texelBuffer(vol.ptr, 8u);
    const uint32_t x0 = (uint32_t)xb;
    const uint4 q00 = fetch128(texels, z0 + y0 + x0), q01 = fetch128(texels, z0 + y1 + x0), q10 = fetch128(texels, z1 + y0 + x0), q11 = fetch128(texels, z1 + y1 + x0); // [z][y]
    // the sampler contract's weights and summation order (image.h sampleLinear3D): eight texels x four channels = 32 v_fma_mix_f32 on the fp16 words
    const float a0 = 1.f - a2, b0 = 1.f - b, c0 = 1.f - c;
    const float wab00 = a0 * b0, wab10 = a2 * b0, wab01 = a0 * b, wab11 = a2 * b;
    vec4 it(0.f);
    it = accumulateTexel(it, q00.x, q00.y, wab00 * c0);
    it = accumulateTexel(it, q00.z, q00.w, wab10 * c0);
    it = accumulateTexel(it, q01.x, q01.y, wab01 * c0);
    it = accumulateTexel(it, q01.z, q01.w, wab11 * c0);
    it = accumulateTexel(it, q10.x, q10.y, wab00 * c);
    it = accumulateTexel(it, q10.z, q10.w, wab10 * c);
    it = accumulateTexel(it, q11.x, q11.y, wab01 * c);
    it = accumulateTexel(it, q11.z, q11.w, wab11 * c);
    return it;
}

// one geometry pixel (depth != 0) of the deferred shade: returns the R11G11B10 colour; *sigWord = decision signature (oracle/oracle.h)
template <int DIFFUSE_BRDF, int MULTISCATTER, bool GEOMETRIC_AA, int INDIRECT_TECH>
PLR_DI uint32_t shadeGeometryPixel(const ShadeParams& P, int px, int py, const ViewRay& vr, const PixelInputs& in, uint32_t* sigWord) {
    SurfaceTerms st;
    const DirectOut d = shadeDirect<DIFFUSE_BRDF, MULTISCATTER, GEOMETRIC_AA>(P, px, py, vr, in, ld3(P.light->sunColor), &st, sigWord);
    vec3 lightingIndirect;
    if (INDIRECT_TECH == 0) lightingIndirect = shadeIndirect<MULTISCATTER>(P, st, in.ysh, in.cocg);
    else {
        const float amb = 0.003f * P.light->sunStrengthExposed;
        const vec3 singleScattering = vmix(vec3(st.brdfLut.x), vec3(st.brdfLut.y), st.f0);
        lightingIndirect = (amb * st.diffuseColor) * st.diffuseBRDFIntegral + singleScattering * amb;
    }
    vec3 outColor = d.direct * P.light->sunStrengthExposed + lightingIndirect;
    const vec4 it = froxelLookup(P, vr.su, vr.sv, d.noiseTexel, d.pixelDepth);
    outColor = outColor * it.w + it.xyz();
    return packR11G11B10(outColor);
}

template <int DIFFUSE_BRDF, int MULTISCATTER, bool GEOMETRIC_AA, int INDIRECT_TECH>
__global__ __launch_bounds__(256, PLR_SHADE_WAVES) void deferredShadingFastKernel(ShadeParams P, PLR_SHADE_UNIFORM_PARAMS) {
    PLR_SHADE_ADOPT_UNIFORMS(P);
    const int px = P.xBase + (int)(blockIdx.x * 64u + (threadIdx.x & 63u));
    const int py = P.yBase + (int)(blockIdx.y * 4u + (threadIdx.x >> 6));
    if (px >= P.coverW || py >= P.coverH) return;
    const ViewRay vr = exactViewRay(P.g, px, py);
    const uint32_t idx = fastm::texelIndex((uint32_t)px, (uint32_t)py, (uint32_t)P.color.w);
    PixelInputs in;
    in.depth = ((const float*)P.depth.ptr)[idx];
    if (in.depth == 0.f) {
        ((uint32_t*)P.color.ptr)[idx] = packR11G11B10(fastm::sampleSkyLut(-vr.Vn, P.skyLut));
        if (P.sig) P.sig[idx] = 128u;
        return;
    }
    // the launcher guarantees that the G-buffer images have the colour target's size: one texel index serves all of them
    in.albedo = ((const uint32_t*)P.albedo.ptr)[idx];
    in.specular = ((const uint32_t*)P.specular.ptr)[idx];
    const uint32_t* normals = (const uint32_t*)P.normal.ptr;
    const uint32_t nw = (uint32_t)P.normal.w;
    const int nx = clampi(px, P.normal.w), ny = clampi(py, P.normal.h);
    in.normal = normals[fastm::texelIndex((uint32_t)nx, (uint32_t)ny, nw)];
    in.normalH = in.normalV = in.normal;
    if (GEOMETRIC_AA) {
        in.normalH = normals[fastm::texelIndex((uint32_t)clampi(px ^ 1, P.normal.w), (uint32_t)ny, nw)];
        in.normalV = normals[fastm::texelIndex((uint32_t)nx, (uint32_t)clampi(py ^ 1, P.normal.h), nw)];
    }
    in.ysh = make_uint2(0u, 0u);
    in.cocg = 0u;
    if (INDIRECT_TECH == 0) {
        const int ix = clampTo(floorToInt(vr.su * (float)P.ysh.w), P.ysh.w - 1), iy = clampTo(floorToInt(vr.sv * (float)P.ysh.h), P.ysh.h - 1);
        in.ysh = ((const uint2*)P.ysh.ptr)[fastm::texelIndex((uint32_t)ix, (uint32_t)iy, (uint32_t)P.ysh.w)];
        in.cocg = ((const uint32_t*)P.cocg.ptr)[fastm::texelIndex((uint32_t)ix, (uint32_t)iy, (uint32_t)P.cocg.w)];
    }
    uint32_t sigWord;
    const uint32_t colour = shadeGeometryPixel<DIFFUSE_BRDF, MULTISCATTER, GEOMETRIC_AA, INDIRECT_TECH>(P, px, py, vr, in, &sigWord);
    ((uint32_t*)P.color.ptr)[idx] = colour;
    if (P.sig) P.sig[idx] = sigWord;
}

// ---- indirectLightUpscale.comp + the deferred shade as ONE launch (pass fusion, backend.h). Same thread layout as the shade
// on its own (a lane is a pixel, a wave a 64-pixel row segment, a block 64 x 4 pixels), so the shade part runs exactly as it does there. The block
// first stages the half-resolution GI texels its pixels can touch - (32 + 2) x (2 + 2) texels: decoded Y_SH / CoCg and the LINEARISED half-res
// depth, 32 bytes each - in LDS: one load, one linearisation and one decode per texel instead of four to nine per pixel. A pixel then resolves its
// upscale (indirectLightUpscale.comp:17-71: closest-depth choice, edge test, bilinear blend with the 2x grid's constant 0.25 / 0.75 weights) from
// LDS and shades with the result in registers. The upscaled images are written only if somebody else could read them (U.storeUpscaled):
// otherwise their 12 B/px of stores and 12 B/px of loads never happen. Operation for operation the upscale is upscaleQuad's (upscale_quad.h),
// so the fused launch and the two separate passes produce the same bytes (tests/test_fusion.py).
// (First built with a thread per 2x2 quad that upscaled the quad in registers and shaded its four pixels in a rolled loop: 281 us against 237 us
//  for the two separate launches at 4K - the loop serialises four pixels' dependent gather chains in one wave. This form keeps one pixel per lane.)
struct FusedUpscale {
    ImgView srcYSH, srcCoCg, halfResDepth, dstYSH, dstCoCg;
    int storeUpscaled;
};
constexpr int kGiTileW = 34, kGiTileH = 4; // half-res texels staged per block
struct GiTexel { float depthLinear, y0, y1, y2, y3, co, cg, pad; };
static_assert(sizeof(GiTexel) == 32, "two 16-byte LDS reads per texel");

template <int DIFFUSE_BRDF, int MULTISCATTER, bool GEOMETRIC_AA>
__global__ __launch_bounds__(256, PLR_SHADE_WAVES) void upscaleAndShadeKernel(ShadeParams P, FusedUpscale U, PLR_SHADE_UNIFORM_PARAMS) {
    PLR_SHADE_ADOPT_UNIFORMS(P);
    __shared__ GiTexel tile[kGiTileH][kGiTileW];
    const int t = (int)threadIdx.x;
    const int X0 = P.xBase + (int)(blockIdx.x * 64u), Y0 = P.yBase + (int)(blockIdx.y * 4u); // both even
    const int k0 = (X0 >> 1) - 1, m0 = (Y0 >> 1) - 1;                               // half-res texel of tile[0][0]
    const GlobalUbo* g = P.g;
    const float nearP = g->nearPlane, farP = g->farPlane, nf = nearP * farP, nmf = nearP - farP;
    if (t < kGiTileW * kGiTileH) {
        const int r = t / kGiTileW, c = t - r * kGiTileW;
        const int hw = U.srcYSH.w, hh = U.srcYSH.h;
        const uint32_t i = fastm::texelIndex((uint32_t)clampi(k0 + c, hw), (uint32_t)clampi(m0 + r, hh), (uint32_t)hw); // clamp-to-edge, as the sampler
        const uint2 ys = fetch64(texelBuffer(U.srcYSH.ptr, 8u), i);
        const uint32_t cc = fetch32(texelBuffer(U.srcCoCg.ptr, 4u), i);
        GiTexel e;
        e.depthLinear = fastquad::linearDepthRounded(halfBitsToFloat(fetch16(texelBuffer(U.halfResDepth.ptr, 2u), i)), nf, nmf, farP);
        e.y0 = halfBitsToFloat(ys.x & 0xffffu); e.y1 = halfBitsToFloat(ys.x >> 16); e.y2 = halfBitsToFloat(ys.y & 0xffffu); e.y3 = halfBitsToFloat(ys.y >> 16);
        e.co = halfBitsToFloat(cc & 0xffffu); e.cg = halfBitsToFloat(cc >> 16); e.pad = 0.f;
        tile[r][c] = e;
    }
    // the pixel's own G-buffer texels and its view ray do not depend on the tile: their loads are in flight while the tile is being staged
    const int px = X0 + (t & 63), py = Y0 + (t >> 6);
    const bool covered = px < P.coverW && py < P.coverH;
    const int cpx = covered ? px : 0, cpy = covered ? py : 0;
    const uint32_t idx = fastm::texelIndex((uint32_t)cpx, (uint32_t)cpy, (uint32_t)P.color.w);
    PixelInputs in;
    in.depth = u2f(fetch32(texelBuffer(P.depth.ptr, 4u), idx));
    in.albedo = fetch32(texelBuffer(P.albedo.ptr, 4u), idx);
    in.specular = fetch32(texelBuffer(P.specular.ptr, 4u), idx);
    const uint32_t* normals = (const uint32_t*)P.normal.ptr;
    in.normal = fetch32(texelBuffer(P.normal.ptr, 4u), idx); // the launcher guarantees the normal image has the colour target's size
    in.normalH = in.normalV = in.normal;
    if (GEOMETRIC_AA) {
        in.normalH = fetch32(texelBuffer(P.normal.ptr, 4u), fastm::texelIndex((uint32_t)clampi(cpx ^ 1, P.normal.w), (uint32_t)cpy, (uint32_t)P.normal.w));
        in.normalV = fetch32(texelBuffer(P.normal.ptr, 4u), fastm::texelIndex((uint32_t)cpx, (uint32_t)clampi(cpy ^ 1, P.normal.h), (uint32_t)P.normal.w));
    }
    const ViewRay vr = exactViewRay(g, cpx, cpy);
    __syncthreads();
    if (!covered) return;
    // ---- the pixel's upscaled GI texel: upscaleQuad's statements for pixel (parity p, q) of quad (k, m) (upscale_quad.h)
    uint32_t upSig;
    {
#pragma clang fp contract(off)
        const int p = px & 1, q = py & 1;
        const int kc = (px >> 1) - k0, mr = (py >> 1) - m0; // the quad's own texel inside the tile (>= 1: the neighbourhood's column / row 0 is kc - 1 / mr - 1)
        const int c0 = kc - 1 + p, r0 = mr - 1 + q;         // texel i0 / j0 of the bilinear footprint
        const GiTexel f00 = tile[r0][c0], f10 = tile[r0][c0 + 1], f01 = tile[r0 + 1][c0], f11 = tile[r0 + 1][c0 + 1];
        const float full = fastquad::linearDepthRounded(in.depth, nf, nmf, farP);
        // textureGather order: (i0, j1), (i1, j1), (i1, j0), (i0, j0)
        const float ds[4] = {f01.depthLinear, f11.depthLinear, f10.depthLinear, f00.depthLinear};
        const int offx[4] = {0, 1, 1, 0}, offy[4] = {1, 1, 0, 0};
        float minDiff = 1000.f;
        int cx = 0, cy = 0;
        bool isEdge = false;
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const float diff = fabsf(ds[i] - full);
            isEdge = isEdge || diff > 0.5f;
            if (diff < minDiff) { minDiff = diff; cx = offx[i]; cy = offy[i]; }
        }
        float y0, y1, y2, y3, co, cg;
        if (isEdge) {
            // nearest texel at uv + offset * halfResTexelSize = half-res texel (k + cx, m + cy), clamped to the image
            const int hw = U.srcYSH.w, hh = U.srcYSH.h;
            const GiTexel n = tile[min((py >> 1) + cy, hh - 1) - m0][min((px >> 1) + cx, hw - 1) - k0];
            y0 = n.y0; y1 = n.y1; y2 = n.y2; y3 = n.y3; co = n.co; cg = n.cg;
        } else {
            const float a = p ? 0.25f : 0.75f, b = q ? 0.25f : 0.75f;
            const float w00 = (1.f - a) * (1.f - b), w10 = a * (1.f - b), w01 = (1.f - a) * b, w11 = a * b;
            auto bl = [&](float t00, float t10, float t01, float t11) { return __builtin_fmaf(t11, w11, __builtin_fmaf(t01, w01, __builtin_fmaf(t10, w10, t00 * w00))); };
            y0 = bl(f00.y0, f10.y0, f01.y0, f11.y0); y1 = bl(f00.y1, f10.y1, f01.y1, f11.y1); y2 = bl(f00.y2, f10.y2, f01.y2, f11.y2); y3 = bl(f00.y3, f10.y3, f01.y3, f11.y3);
            co = bl(f00.co, f10.co, f01.co, f11.co); cg = bl(f00.cg, f10.cg, f01.cg, f11.cg);
        }
        in.ysh = make_uint2(floatToHalfBits(y0) | (floatToHalfBits(y1) << 16), floatToHalfBits(y2) | (floatToHalfBits(y3) << 16));
        in.cocg = floatToHalfBits(co) | (floatToHalfBits(cg) << 16);
        upSig = (isEdge ? 1u : 0u) | (cx ? 2u : 0u) | (cy ? 4u : 0u);
    }
    if (U.storeUpscaled) {
        ((uint2*)U.dstYSH.ptr)[idx] = in.ysh;
        ((uint32_t*)U.dstCoCg.ptr)[idx] = in.cocg;
    }
    if (in.depth == 0.f) {
        ((uint32_t*)P.color.ptr)[idx] = packR11G11B10(fastm::sampleSkyLut(-vr.Vn, P.skyLut));
        if (P.sig) P.sig[idx] = 128u | (upSig << 8);
        return;
    }
    uint32_t sigWord;
    const uint32_t colour = shadeGeometryPixel<DIFFUSE_BRDF, MULTISCATTER, GEOMETRIC_AA, 0>(P, px, py, vr, in, &sigWord);
    ((uint32_t*)P.color.ptr)[idx] = colour;
    if (P.sig) P.sig[idx] = sigWord | (upSig << 8); // decision signature of the fused launch: the shade's word | the upscale's word << 8
}

// ---- Round 6: the fused upscale + shade as TWO launches, so that the half of the shade that needs no GI can run BESIDE the GI chain.
//  (A) shadeDirectKernel - per pixel, everything of triangle.frag:146-290 that depends on the G-buffer, the shadow cascades and the LUTs only: material, view /
//      half vectors, cascade select + twelve-tap PCF, diffuse + GGX + multiscattering response to a sun of UNIT colour, the froxel lookup. It reads no GI image,
//      not the light buffer (sun colour x exposure: the frame front writes it) and takes the global uniform block BY VALUE (the host copy, PassCtx::globalHost:
//      the device buffer is filled by the frame's first launch on the main stream), so the backend may start it on the early stream before the frame front
//      (backend.h EarlyPart). Its result is a 36-byte record per pixel in the pass's scratch memory, three planes:
//        a (16 B)  X.rgb fp32 = (diffuseDirect + specularDirect) / lightColor * transmittance        | half transmittance, half r (after geometric AA)
//        b (16 B)  half in-scattering rgb, half energyOutgoing (brdfLut.y) | half diffuseColor * diffuseBRDFIntegral rgb, half f0.r
//        c ( 4 B)  half f0.g, half f0.b
//      a sky pixel (depth 0): a.x = the packed sky colour, nothing else is read.
//      The direct term stays fp32 (it is most of the pixel and un-exposed: a mirror-like highlight can exceed the fp16 range); the rest only enters the
//      indirect term or the additive fog and is rounded towards zero to fp16: 2^-10 relative on a part of a sum that is stored with 6 / 5 mantissa bits.
//  (B) upscaleAndCombineKernel - the block's GI tile in LDS and the per-pixel upscale exactly as upscaleAndShadeKernel, then shadeIndirect from the record and
//      colour = X * (sunColor * sunStrengthExposed) + (indirect * transmittance + in-scattering)
//      (fused: ((direct * exposure) + indirect) * transmittance + in-scattering; the same sum up to fp32 re-association).
// The pair is chosen by the backend when the early part has something to run beside (launchUpscaleAndShade below); tests/test_fusion.py holds its output to
// within one R11G11B10 code of the single launch, tests/test_parity_fullsize.py to the same caps against the oracle.
struct DirectRecords { uint4* a; uint4* b; uint32_t* c; };
PLR_DI uint32_t packHalves(float lo, float hi) { return __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_pkrtz(lo, hi)); }

template <int DIFFUSE_BRDF, int MULTISCATTER, bool GEOMETRIC_AA>
__global__ __launch_bounds__(256, PLR_SHADE_WAVES) void shadeDirectKernel(ShadeParams P, DirectRecords R, const GlobalUbo G, const ShadowCascadeInfo* __restrict__ shadowUniform,
                                                                          const VolumetricLightingSettings* __restrict__ volUniform) {
    P.g = &G; P.shadowInfo = shadowUniform; P.vol = volUniform; P.light = nullptr;
    const int px = P.xBase + (int)(blockIdx.x * 64u + (threadIdx.x & 63u));
    const int py = P.yBase + (int)(blockIdx.y * 4u + (threadIdx.x >> 6));
    if (px >= P.coverW || py >= P.coverH) return;
    const ViewRay vr = exactViewRay(P.g, px, py);
    const uint32_t idx = fastm::texelIndex((uint32_t)px, (uint32_t)py, (uint32_t)P.color.w);
    PixelInputs in;
    in.depth = u2f(fetch32(texelBuffer(P.depth.ptr, 4u), idx));
    if (in.depth == 0.f) {
        R.a[idx] = make_uint4(packR11G11B10(fastm::sampleSkyLut(-vr.Vn, P.skyLut)), 0u, 0u, 0u);
        return;
    }
    in.albedo = fetch32(texelBuffer(P.albedo.ptr, 4u), idx);
    in.specular = fetch32(texelBuffer(P.specular.ptr, 4u), idx);
    in.normal = fetch32(texelBuffer(P.normal.ptr, 4u), idx); // the launcher guarantees the normal image has the colour target's size
    in.normalH = in.normalV = in.normal;
    if (GEOMETRIC_AA) {
        in.normalH = fetch32(texelBuffer(P.normal.ptr, 4u), fastm::texelIndex((uint32_t)clampi(px ^ 1, P.normal.w), (uint32_t)py, (uint32_t)P.normal.w));
        in.normalV = fetch32(texelBuffer(P.normal.ptr, 4u), fastm::texelIndex((uint32_t)px, (uint32_t)clampi(py ^ 1, P.normal.h), (uint32_t)P.normal.w));
    }
    in.ysh = make_uint2(0u, 0u);
    in.cocg = 0u;
    SurfaceTerms st;
    uint32_t sigWord;
    const DirectOut d = shadeDirect<DIFFUSE_BRDF, MULTISCATTER, GEOMETRIC_AA>(P, px, py, vr, in, vec3(1.f), &st, &sigWord);
    const vec4 it = froxelLookup(P, vr.su, vr.sv, d.noiseTexel, d.pixelDepth);
    const vec3 X = d.direct * it.w;
    const vec3 dci = st.diffuseColor * st.diffuseBRDFIntegral;
    R.a[idx] = make_uint4(f2u(X.x), f2u(X.y), f2u(X.z), packHalves(it.w, st.r));
    R.b[idx] = make_uint4(packHalves(it.x, it.y), packHalves(it.z, st.energyOutgoing), packHalves(dci.x, dci.y), packHalves(dci.z, st.f0.x));
    R.c[idx] = packHalves(st.f0.y, st.f0.z);
}

template <int MULTISCATTER>
__global__ __launch_bounds__(256) void upscaleAndCombineKernel(ShadeParams P, FusedUpscale U, DirectRecords R, const GlobalUbo* __restrict__ gUniform, const LightBuffer* __restrict__ lightUniform) {
    P.g = gUniform; P.light = lightUniform;
    __shared__ GiTexel tile[kGiTileH][kGiTileW];
    const int t = (int)threadIdx.x;
    const int X0 = P.xBase + (int)(blockIdx.x * 64u), Y0 = P.yBase + (int)(blockIdx.y * 4u); // both even
    const int k0 = (X0 >> 1) - 1, m0 = (Y0 >> 1) - 1;                               // half-res texel of tile[0][0]
    const GlobalUbo* g = P.g;
    const float nearP = g->nearPlane, farP = g->farPlane, nf = nearP * farP, nmf = nearP - farP;
    if (t < kGiTileW * kGiTileH) {
        const int r = t / kGiTileW, c = t - r * kGiTileW;
        const int hw = U.srcYSH.w, hh = U.srcYSH.h;
        const uint32_t i = fastm::texelIndex((uint32_t)clampi(k0 + c, hw), (uint32_t)clampi(m0 + r, hh), (uint32_t)hw); // clamp-to-edge, as the sampler
        const uint2 ys = fetch64(texelBuffer(U.srcYSH.ptr, 8u), i);
        const uint32_t cc = fetch32(texelBuffer(U.srcCoCg.ptr, 4u), i);
        GiTexel e;
        e.depthLinear = fastquad::linearDepthRounded(halfBitsToFloat(fetch16(texelBuffer(U.halfResDepth.ptr, 2u), i)), nf, nmf, farP);
        e.y0 = halfBitsToFloat(ys.x & 0xffffu); e.y1 = halfBitsToFloat(ys.x >> 16); e.y2 = halfBitsToFloat(ys.y & 0xffffu); e.y3 = halfBitsToFloat(ys.y >> 16);
        e.co = halfBitsToFloat(cc & 0xffffu); e.cg = halfBitsToFloat(cc >> 16); e.pad = 0.f;
        tile[r][c] = e;
    }
    // the pixel's depth, normal and record do not depend on the tile: their loads are in flight while the tile is being staged
    const int px = X0 + (t & 63), py = Y0 + (t >> 6);
    const bool covered = px < P.coverW && py < P.coverH;
    const int cpx = covered ? px : 0, cpy = covered ? py : 0;
    const uint32_t idx = fastm::texelIndex((uint32_t)cpx, (uint32_t)cpy, (uint32_t)P.color.w);
    const float depth = u2f(fetch32(texelBuffer(P.depth.ptr, 4u), idx));
    const uint32_t normalTexel = fetch32(texelBuffer(P.normal.ptr, 4u), idx);
    const uint4 ra = fetch128(texelBuffer(R.a, 16u), idx), rb = fetch128(texelBuffer(R.b, 16u), idx);
    const uint32_t rc = fetch32(texelBuffer(R.c, 4u), idx);
    __syncthreads();
    if (!covered) return;
    // ---- the pixel's upscaled GI texel: upscaleQuad's statements for pixel (parity p, q) of quad (k, m) (upscale_quad.h), as in upscaleAndShadeKernel
    uint2 ysh;
    uint32_t cocg;
    {
#pragma clang fp contract(off)
        const int p = px & 1, q = py & 1;
        const int kc = (px >> 1) - k0, mr = (py >> 1) - m0;
        const int c0 = kc - 1 + p, r0 = mr - 1 + q;
        const GiTexel f00 = tile[r0][c0], f10 = tile[r0][c0 + 1], f01 = tile[r0 + 1][c0], f11 = tile[r0 + 1][c0 + 1];
        const float full = fastquad::linearDepthRounded(depth, nf, nmf, farP);
        const float ds[4] = {f01.depthLinear, f11.depthLinear, f10.depthLinear, f00.depthLinear};
        const int offx[4] = {0, 1, 1, 0}, offy[4] = {1, 1, 0, 0};
        float minDiff = 1000.f;
        int cx = 0, cy = 0;
        bool isEdge = false;
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const float diff = fabsf(ds[i] - full);
            isEdge = isEdge || diff > 0.5f;
            if (diff < minDiff) { minDiff = diff; cx = offx[i]; cy = offy[i]; }
        }
        float y0, y1, y2, y3, co, cg;
        if (isEdge) {
            const int hw = U.srcYSH.w, hh = U.srcYSH.h;
            const GiTexel n = tile[min((py >> 1) + cy, hh - 1) - m0][min((px >> 1) + cx, hw - 1) - k0];
            y0 = n.y0; y1 = n.y1; y2 = n.y2; y3 = n.y3; co = n.co; cg = n.cg;
        } else {
            const float a = p ? 0.25f : 0.75f, b = q ? 0.25f : 0.75f;
            const float w00 = (1.f - a) * (1.f - b), w10 = a * (1.f - b), w01 = (1.f - a) * b, w11 = a * b;
            auto bl = [&](float t00, float t10, float t01, float t11) { return __builtin_fmaf(t11, w11, __builtin_fmaf(t01, w01, __builtin_fmaf(t10, w10, t00 * w00))); };
            y0 = bl(f00.y0, f10.y0, f01.y0, f11.y0); y1 = bl(f00.y1, f10.y1, f01.y1, f11.y1); y2 = bl(f00.y2, f10.y2, f01.y2, f11.y2); y3 = bl(f00.y3, f10.y3, f01.y3, f11.y3);
            co = bl(f00.co, f10.co, f01.co, f11.co); cg = bl(f00.cg, f10.cg, f01.cg, f11.cg);
        }
        // the fused kernel shades with the texel ROUNDED to the images' fp16 (it is what a stand-alone upscale would have stored): the same here
        ysh = make_uint2(floatToHalfBits(y0) | (floatToHalfBits(y1) << 16), floatToHalfBits(y2) | (floatToHalfBits(y3) << 16));
        cocg = floatToHalfBits(co) | (floatToHalfBits(cg) << 16);
    }
    if (U.storeUpscaled) {
        ((uint2*)U.dstYSH.ptr)[idx] = ysh;
        ((uint32_t*)U.dstCoCg.ptr)[idx] = cocg;
    }
    if (depth == 0.f) {
        ((uint32_t*)P.color.ptr)[idx] = ra.x; // the sky colour, packed by the direct launch
        return;
    }
    SurfaceTerms st;
    st.N = decodeNormal(normalTexel);
    {
        // the view ray for SHADING only (no discrete decision hangs on it): one v_rsq instead of the correctly rounded chain of exactViewRay
        const float su = ((float)px + 0.5f) * rcpf((float)g->screenResolution[0]), sv = ((float)py + 0.5f) * rcpf((float)g->screenResolution[1]);
        const float ndx = su * 2.f - 1.f, ndy = sv * 2.f - 1.f;
        const float ty = g->cameraTanFovHalf * ndy, tx = (g->cameraTanFovHalf * g->cameraAspectRatio) * ndx;
        st.V = nrm(vec3((-g->cameraForward[0] + ty * g->cameraUp[0]) - tx * g->cameraRight[0], (-g->cameraForward[1] + ty * g->cameraUp[1]) - tx * g->cameraRight[1],
                        (-g->cameraForward[2] + ty * g->cameraUp[2]) - tx * g->cameraRight[2]));
    }
    st.NoV = fmax1(fabsf(dot(st.N, st.V)), 0.0001f);
    const float transmittance = (float)halfLo(ra.w);
    st.r = (float)halfHi(ra.w);
    st.energyOutgoing = (float)halfHi(rb.y);
    st.diffuseColor = vec3((float)halfLo(rb.z), (float)halfHi(rb.z), (float)halfLo(rb.w)); // diffuseColor * diffuseBRDFIntegral
    st.diffuseBRDFIntegral = vec3(1.f);
    st.f0 = vec3((float)halfHi(rb.w), (float)halfLo(rc), (float)halfHi(rc));
    st.brdfLut = vec3(0.f);
    const vec3 lightingIndirect = shadeIndirect<MULTISCATTER>(P, st, ysh, cocg);
    const vec3 inScattering((float)halfLo(rb.x), (float)halfHi(rb.x), (float)halfLo(rb.y));
    const vec3 light = ld3(P.light->sunColor) * P.light->sunStrengthExposed;
    const vec3 outColor = vec3(u2f(ra.x), u2f(ra.y), u2f(ra.z)) * light + (lightingIndirect * transmittance + inScattering);
    ((uint32_t*)P.color.ptr)[idx] = packR11G11B10(outColor);
}

typedef void (*ShadeKernel)(ShadeParams, const GlobalUbo*, const LightBuffer*, const ShadowCascadeInfo*, const VolumetricLightingSettings*);
template <int D, int M, bool G> static ShadeKernel pickIndirect(int tech) {
    return tech == 0 ? (ShadeKernel)deferredShadingFastKernel<D, M, G, 0> : (ShadeKernel)deferredShadingFastKernel<D, M, G, 1>;
}
template <int D, int M> static ShadeKernel pickAA(bool aa, int tech) { return aa ? pickIndirect<D, M, true>(tech) : pickIndirect<D, M, false>(tech); }
template <int D> static ShadeKernel pickMulti(int m, bool aa, int tech) {
    switch (m) {
        case 0: return pickAA<D, 0>(aa, tech);
        case 1: return pickAA<D, 1>(aa, tech);
        case 2: return pickAA<D, 2>(aa, tech);
        default: return pickAA<D, 3>(aa, tech);
    }
}
typedef void (*FusedKernel)(ShadeParams, FusedUpscale, const GlobalUbo*, const LightBuffer*, const ShadowCascadeInfo*, const VolumetricLightingSettings*);
template <int D, int M> static FusedKernel pickFusedAA(bool aa) { return aa ? (FusedKernel)upscaleAndShadeKernel<D, M, true> : (FusedKernel)upscaleAndShadeKernel<D, M, false>; }
template <int D> static FusedKernel pickFusedMulti(int m, bool aa) {
    switch (m) {
        case 0: return pickFusedAA<D, 0>(aa);
        case 1: return pickFusedAA<D, 1>(aa);
        case 2: return pickFusedAA<D, 2>(aa);
        default: return pickFusedAA<D, 3>(aa);
    }
}

// ---- tables the shade derives from its inputs, in the pass's scratch memory: [tap table by noise value 24 KB | kNoiseSlots x tap table by noise position]
// The by-position table of a noise texture is rebuilt when the texture's content version changes (backend.h contentVersionOf: creation, uploads,
// any execution that writes it); a texture whose address was handed out (version 0), one with more than kMaxNoisePositions texels, or a frame
// whose noise texture the host cannot name takes the by-value table. Four slots: the reference cycles four noise textures (frameIndexMod4).
constexpr int kNoiseSlots = 4;
constexpr uint32_t kMaxNoisePositions = 4096;
constexpr size_t kNoiseSlotBytes = (size_t)kMaxNoisePositions * (kPcfTaps / 2) * sizeof(float4);
__global__ void lutEnergyFootprintKernel(ImgView lut, uint2* __restrict__ out) {
    const int x = (int)(blockIdx.x * 64u + (threadIdx.x & 63u)), y = (int)(blockIdx.y * 4u + (threadIdx.x >> 6));
    if (x >= lut.w - 1 || y >= lut.h - 1) return;
    const uint2* t = (const uint2*)lut.ptr;
    auto energy = [&](int tx, int ty) { return t[(size_t)ty * (size_t)lut.w + tx].x >> 16; }; // RGBA16F texel: .y = upper half of the first word
    out[(size_t)y * (size_t)(lut.w - 1) + x] = make_uint2(energy(x, y) | (energy(x + 1, y) << 16), energy(x, y + 1) | (energy(x + 1, y + 1) << 16));
}
__global__ void pcfTapsByPositionKernel(const uint16_t* __restrict__ noiseTexels, uint32_t positions, const float4* __restrict__ byValue, float4* __restrict__ byPosition) {
    const uint32_t pos = blockIdx.x * blockDim.x + threadIdx.x;
    if (pos >= positions) return;
    const uint32_t noiseByte = noiseTexels[pos] & 0xffu; // RG8 texel, .r
    // The shadow test counts lit taps: their order is free. Tap i of this texel points at angle 2 pi (noise + i / 12); slot j of the row gets the tap
    // whose direction is nearest to 2 pi j / 12, i = j - round(12 noise) (mod 12). The j-th shadow-map load of a wave then reads, for every lane, a
    // texel on the SAME side of its tap centre - a strip along one direction instead of a whole disc - and touches fewer cache lines.
    const float2* taps = (const float2*)byValue + noiseByte * kPcfTaps;
    const int turn = (int)((float)noiseByte * (12.f / 255.f) + 0.5f);
#pragma unroll
    for (int k = 0; k < kPcfTaps / 2; k++) {
        const int i0 = ((2 * k - turn) % kPcfTaps + kPcfTaps) % kPcfTaps, i1 = ((2 * k + 1 - turn) % kPcfTaps + kPcfTaps) % kPcfTaps;
        byPosition[(uint32_t)k * positions + pos] = make_float4(taps[i0].x, taps[i0].y, taps[i1].x, taps[i1].y);
    }
}
struct ShadeDerived {
    const void* scratchBase = nullptr; // the allocation the entries below describe (a re-allocated scratch starts empty)
    struct Slot { const void* noise = nullptr; int w = 0, h = 0; uint64_t version = 0, lastUse = 0; } slots[kNoiseSlots];
    uint64_t useCounter = 0;
    const void* lut = nullptr; int lutW = 0, lutH = 0; uint64_t lutVersion = 0; // what the energy footprint was derived from
};
static thread_local std::map<const void*, ShadeDerived> g_shadeDerived; // key = the pass's scratch slot (one backend per host thread)
static int shadeDerivedTables(const PassCtx& c, ShadeParams* P, DirectRecords* records = nullptr) {
    // [tap table by value | noise slots | BRDF LUT energy footprint (size follows the LUT: a larger LUT re-allocates the scratch and everything is rebuilt)
    //  | the direct launch's records: planes a, b, c over the colour target (only when the shade runs as two launches)]
    const ImgView& lut = P->brdfLut;
    const size_t footprintOffset = kPcfTapTableBytes + kNoiseSlots * kNoiseSlotBytes;
    const size_t footprintBytes = (size_t)(lut.w - 1) * (size_t)std::max(lut.h - 1, 1) * sizeof(uint2); // (the launcher sends LUTs narrower than two texels to the general kernel)
    const size_t recordsOffset = (footprintOffset + footprintBytes + 255) & ~(size_t)255;
    const size_t pixels = (size_t)P->color.w * (size_t)P->color.h;
    const size_t total = records ? recordsOffset + pixels * 36 : footprintOffset + footprintBytes;
    uint8_t* scratch = (uint8_t*)c.scratch(total);
    if (!scratch) return c.fail(-2, "deferredShading: cannot allocate scratch memory");
    if (records) { records->a = (uint4*)(scratch + recordsOffset); records->b = records->a + pixels; records->c = (uint32_t*)(records->b + pixels); }
    ShadeDerived& d = g_shadeDerived[(const void*)c.scratchSlot];
    if (d.scratchBase != scratch) {
        d = ShadeDerived{};
        d.scratchBase = scratch;
        const hipError_t e = buildPcfTapTable((float2*)scratch, c.stream);
        if (e != hipSuccess) return c.fail(-2, std::string("deferredShading: PCF tap table: ") + hipGetErrorString(e));
    }
    P->pcfTaps = (const float4*)scratch;
    P->pcfTapsByPosition = nullptr;
    P->pcfPositions = 0;
    P->lutEnergyFootprint = nullptr;
    if (const uint64_t lutVersion = lut.h >= 2 ? contentVersionOf(lut.ptr) : 0) { // 0: the LUT may change behind the backend's back - the kernel reads the LUT itself
        if (d.lut != lut.ptr || d.lutW != lut.w || d.lutH != lut.h || d.lutVersion != lutVersion) {
            lutEnergyFootprintKernel<<<dim3(divUp((unsigned)(lut.w - 1), 64u), divUp((unsigned)(lut.h - 1), 4u)), 256, 0, c.stream>>>(lut, (uint2*)(scratch + footprintOffset));
            PLR_CHECK_LAUNCH(c);
            d.lut = lut.ptr; d.lutW = lut.w; d.lutH = lut.h; d.lutVersion = lutVersion;
        }
        P->lutEnergyFootprint = (const uint2*)(scratch + footprintOffset);
    }
    const ImgView& noise = P->noiseTex;
    if (!noise.ptr || noise.fmt != F_RG8) return 0;
    const uint64_t version = contentVersionOf(noise.ptr);
    const uint64_t positions = (uint64_t)noise.w * (uint64_t)noise.h;
    if (version == 0 || positions > kMaxNoisePositions) return 0;
    ShadeDerived::Slot* slot = nullptr;
    for (auto& s : d.slots) if (s.noise == noise.ptr && s.w == noise.w && s.h == noise.h && s.version == version) slot = &s;
    if (!slot) {
        slot = &d.slots[0];
        for (auto& s : d.slots) if (s.lastUse < slot->lastUse) slot = &s; // least recently used
        float4* dst = (float4*)(scratch + kPcfTapTableBytes + (size_t)(slot - d.slots) * kNoiseSlotBytes);
        pcfTapsByPositionKernel<<<divUp((unsigned)positions, 256u), 256, 0, c.stream>>>((const uint16_t*)noise.ptr, (uint32_t)positions, (const float4*)scratch, dst);
        PLR_CHECK_LAUNCH(c);
        slot->noise = noise.ptr; slot->w = noise.w; slot->h = noise.h; slot->version = version;
    }
    slot->lastUse = ++d.useCounter;
    P->pcfTapsByPosition = (const float4*)(scratch + kPcfTapTableBytes + (size_t)(slot - d.slots) * kNoiseSlotBytes);
    P->pcfPositions = (uint32_t)positions;
    return 0;
}

// validates the bindings of a deferred shade execution and fills the kernel parameters; 0, kUseGeneralKernel or an error
static int shadeParamsFor(const PassCtx& c, ShadeParams* out, int* diffuseBRDF, int* multi, bool* aa, int* tech, bool deriveTables = true, DirectRecords* records = nullptr) {
    if (int rc = c.needGlobal()) return rc;
    if (int rc = c.needStorage(0, F_R11G11B10, "deferredShading colour target")) return rc;
    if (int rc = c.needSampled(3, F_RGBA16F, "deferredShading brdfLutTexture")) return rc;
    if (int rc = c.needSbuf(7, sizeof(LightBuffer), "deferredShading lightBuffer")) return rc;
    if (int rc = c.needSbuf(8, sizeof(ShadowCascadeInfo), "deferredShading sunShadowInfo")) return rc;
    for (int i = 0; i < 4; i++) if (int rc = c.needSampled(9 + i, F_D16, "deferredShading shadowMapCascade")) return rc;
    if (int rc = c.needSampled(15, F_RGBA16F, "deferredShading indirectDiffuse_Y_SH")) return rc;
    if (int rc = c.needSampled(16, F_RG16F, "deferredShading indirectDiffuse_CoCg")) return rc;
    if (int rc = c.needSampled(18, F_RGBA16F, "deferredShading volumetricLightingLUT")) return rc;
    if (int rc = c.needUbuf(19, 52, "deferredShading volumetric settings")) return rc;
    if (int rc = c.needSampled(20, F_D32, "deferredShading depth")) return rc;
    if (int rc = c.needSampled(21, F_RGBA8, "deferredShading world normals")) return rc;
    if (int rc = c.needSampled(22, F_RGBA8, "deferredShading albedo")) return rc;
    if (int rc = c.needSampled(23, F_RGBA8, "deferredShading specular")) return rc;
    if (int rc = c.needSampled(24, F_R11G11B10, "deferredShading skyLut")) return rc;
    if (!c.bindless || c.bindlessCount == 0) return c.fail(-4, "deferredShading: global texture array (set 2) is empty");
    if (c.sampled[16].w != c.sampled[15].w || c.sampled[16].h != c.sampled[15].h) return c.fail(-4, "deferredShading: Y_SH and CoCg differ in size");
    // this kernel addresses depth / albedo / specular with the colour target's texel index: other layouts take the general kernel
    for (int b : {20, 22, 23}) if (c.sampled[b].w != c.storage[0].w || c.sampled[b].h != c.storage[0].h) return kUseGeneralKernel;
    // the LUT / froxel fetches read the two texels of a row with one 16-byte load
    if (c.sampled[3].w < 2 || c.sampled[18].w < 2) return kUseGeneralKernel;
    *diffuseBRDF = c.specInt(0, 0); *multi = c.specInt(1, 0); *tech = c.specInt(3, 0);
    *aa = c.specBool(2, false);
    const uint32_t cascades = c.specUint(4, 4u);
    if (*diffuseBRDF < 0 || *diffuseBRDF > 3 || *multi < 0 || *multi > 3 || cascades < 1 || cascades > 4) return c.fail(-1, "deferredShading: specialisation constant out of range");
    ShadeParams P{};
    P.color = c.storage[0]; P.depth = c.sampled[20]; P.normal = c.sampled[21]; P.albedo = c.sampled[22]; P.specular = c.sampled[23];
    P.brdfLut = c.sampled[3];
    for (int i = 0; i < 4; i++) P.shadowMaps[i] = c.sampled[9 + i];
    P.ysh = c.sampled[15]; P.cocg = c.sampled[16]; P.volumetricLut = c.sampled[18]; P.skyLut = c.sampled[24];
    P.light = (const LightBuffer*)c.sbuf[7].ptr; P.shadowInfo = (const ShadowCascadeInfo*)c.sbuf[8].ptr;
    P.vol = (const VolumetricLightingSettings*)c.ubuf[19].ptr; P.g = c.global;
    P.bindless = c.bindless; P.bindlessCount = c.bindlessCount; P.cascadeCount = cascades;
    // frame index -> texture index -> view are three dependent round trips in front of every pixel's shadow taps when a kernel chases them itself; this
    // kernel takes the view from the host, and a host that does not know the global buffer's contents (PassCtx::hostNoiseView) gets the general kernel
    if (!c.hostNoiseView(&P.noiseTex)) return kUseGeneralKernel;
    if (deriveTables) if (int rc = shadeDerivedTables(c, &P, records)) return rc;
    const PassCtx::RowSpan rs = c.rowSpan(P.color.h);
    const PassCtx::ColSpan cs = c.colSpan(P.color.w);
    P.coverW = cs.x1; P.xBase = cs.x0; P.coverH = rs.y1; P.yBase = rs.y0; // columns [xBase, coverW), rows [yBase, coverH)
    P.sig = c.sigFor((size_t)P.color.w * (size_t)P.color.h);
    *out = P;
    return 0;
}

static int launchDeferredShadingFast(const PassCtx& c) {
    ShadeParams P;
    int diffuseBRDF, multi, tech;
    bool aa;
    if (int rc = shadeParamsFor(c, &P, &diffuseBRDF, &multi, &aa, &tech)) return rc;
    ShadeKernel k = nullptr;
    switch (diffuseBRDF) {
        case 0: k = pickMulti<0>(multi, aa, tech); break;
        case 1: k = pickMulti<1>(multi, aa, tech); break;
        case 2: k = pickMulti<2>(multi, aa, tech); break;
        default: k = pickMulti<3>(multi, aa, tech); break;
    }
    if (P.coverW <= P.xBase || P.coverH <= P.yBase) return 0;
    k<<<dim3(divUp((unsigned)(P.coverW - P.xBase), 64u), divUp((unsigned)(P.coverH - P.yBase), 4u)), 256, 0, c.stream>>>(P, PLR_SHADE_UNIFORM_ARGS(P));
    PLR_CHECK_LAUNCH(c);
    return 0;
}

// fused: indirectLightUpscale.comp, then the deferred shade that samples the images it wrote, over the same rows.
// fusedShadeSetup validates the pair and fills the kernel parameters (0, kUseGeneralKernel or an error); deriveTables = false: no launch, no allocation (the early
// part's query); records: the shade runs as two launches and these are the planes between them
struct FusedShade { ShadeParams P; FusedUpscale U; int diffuseBRDF, multi; bool aa; };
static int fusedShadeSetup(const PassCtx* const* ctxs, size_t count, FusedShade* f, bool deriveTables, DirectRecords* records) {
    if (count != 2) return kUseGeneralKernel;
    const PassCtx &u = *ctxs[0], &c = *ctxs[1];
    // the upscale's bindings (launchUpscale, stream_fast.hip) in its "regular" 2x case
    if (!u.global || !u.hasStorage(0) || !u.hasStorage(1) || !u.hasSampled(2) || !u.hasSampled(3) || !u.hasSampled(4) || !u.hasSampled(5)) return kUseGeneralKernel;
    if (u.storage[0].fmt != F_RGBA16F || u.storage[1].fmt != F_RG16F || u.sampled[2].fmt != F_RGBA16F || u.sampled[3].fmt != F_RG16F || u.sampled[4].fmt != F_D32 ||
        u.sampled[5].fmt != F_R16F)
        return kUseGeneralKernel;
    const ImgView& out = u.storage[0];
    const PassCtx::RowSpan ur = u.rowSpan(out.h);
    const bool regular = out.w == 2 * u.sampled[2].w && out.h == 2 * u.sampled[2].h && u.sampled[3].w == u.sampled[2].w && u.sampled[3].h == u.sampled[2].h &&
                         u.sampled[5].w == u.sampled[2].w && u.sampled[5].h == u.sampled[2].h && u.sampled[4].w == out.w && u.sampled[4].h == out.h &&
                         u.storage[1].w == out.w && u.storage[1].h == out.h && (ur.y0 & 1) == 0 && u.sampled[2].w >= 4;
    if (!regular) return kUseGeneralKernel;
    ShadeParams& P = f->P;
    int tech;
    if (int rc = shadeParamsFor(c, &P, &f->diffuseBRDF, &f->multi, &f->aa, &tech, deriveTables, records)) return rc;
    if (tech != 0) return kUseGeneralKernel; // the shade does not read the upscaled images
    // the shade must read exactly what the upscale writes, on the same pixel grid, the same depth buffer, over the same rows and whole rows
    if (c.sampled[15].ptr != out.ptr || c.sampled[16].ptr != u.storage[1].ptr || c.sampled[15].w != out.w || c.sampled[15].h != out.h || P.color.w != out.w ||
        P.color.h != out.h || c.sampled[20].ptr != u.sampled[4].ptr || c.sampled[21].w != out.w || c.sampled[21].h != out.h)
        return kUseGeneralKernel;
    const PassCtx::ColSpan uc = u.colSpan(out.w);
    if (uc.x0 != P.xBase || uc.x1 != P.coverW || ur.y0 != P.yBase || ur.y1 != P.coverH) return kUseGeneralKernel; // the same rows and the same columns
    // uv is defined by the UBO's screen resolution (indirectLightUpscale.comp:19): the quad reasoning needs it to be the target size
    if (c.global != u.global || !u.globalHost || u.globalHost->screenResolution[0] != out.w || u.globalHost->screenResolution[1] != out.h) return kUseGeneralKernel;
    FusedUpscale& U = f->U;
    U.srcYSH = u.sampled[2]; U.srcCoCg = u.sampled[3]; U.halfResDepth = u.sampled[5]; U.dstYSH = u.storage[0]; U.dstCoCg = u.storage[1];
    U.storeUpscaled = 1;
    return 0;
}

typedef void (*DirectKernel)(ShadeParams, DirectRecords, const GlobalUbo, const ShadowCascadeInfo*, const VolumetricLightingSettings*);
template <int D, int M> static DirectKernel pickDirectAA(bool aa) { return aa ? (DirectKernel)shadeDirectKernel<D, M, true> : (DirectKernel)shadeDirectKernel<D, M, false>; }
template <int D> static DirectKernel pickDirectMulti(int m, bool aa) {
    switch (m) {
        case 0: return pickDirectAA<D, 0>(aa);
        case 1: return pickDirectAA<D, 1>(aa);
        case 2: return pickDirectAA<D, 2>(aa);
        default: return pickDirectAA<D, 3>(aa);
    }
}

// the EARLY PART of the pair (backend.h EarlyPart): the direct lighting, as a launch of its own on the early stream
static int earlyShadeDirect(const PassCtx* const* ctxs, size_t count, EarlyPart& part) {
    FusedShade f;
    DirectRecords R{};
    const bool launch = part.mode == EarlyPart::Launch;
    if (int rc = fusedShadeSetup(ctxs, count, &f, launch, launch ? &R : nullptr)) return rc;
    const PassCtx& c = *ctxs[1];
    if (!c.globalHost || c.sigFor(1)) return kUseGeneralKernel; // the kernel takes the uniform block by value; decision signatures are written by the single launch
    if (!launch) {
        // everything the direct launch reads (allocation bases): G-buffer, LUTs, shadow cascades, the two small blocks, the frame's noise texture
        const ShadeParams& P = f.P;
        for (const void* k : {(const void*)P.depth.ptr, (const void*)P.normal.ptr, (const void*)P.albedo.ptr, (const void*)P.specular.ptr, (const void*)P.brdfLut.ptr,
                              (const void*)P.shadowMaps[0].ptr, (const void*)P.shadowMaps[1].ptr, (const void*)P.shadowMaps[2].ptr, (const void*)P.shadowMaps[3].ptr,
                              (const void*)P.volumetricLut.ptr, (const void*)P.skyLut.ptr, (const void*)P.shadowInfo, (const void*)P.vol, (const void*)P.noiseTex.ptr})
            part.reads->push_back(k);
        return 0;
    }
    const ShadeParams& P = f.P;
    if (P.coverH <= P.yBase || P.coverW <= P.xBase) return 0;
    DirectKernel k = nullptr;
    switch (f.diffuseBRDF) {
        case 0: k = pickDirectMulti<0>(f.multi, f.aa); break;
        case 1: k = pickDirectMulti<1>(f.multi, f.aa); break;
        case 2: k = pickDirectMulti<2>(f.multi, f.aa); break;
        default: k = pickDirectMulti<3>(f.multi, f.aa); break;
    }
    k<<<dim3(divUp((unsigned)(P.coverW - P.xBase), 64u), divUp((unsigned)(P.coverH - P.yBase), 4u)), 256, 0, c.stream>>>(P, R, *c.globalHost, P.shadowInfo, P.vol);
    PLR_CHECK_LAUNCH(c);
    return 0;
}

static int launchUpscaleAndShade(const PassCtx* const* ctxs, size_t count) {
    FusedShade f;
    DirectRecords R{};
    const bool late = count == 2 && ctxs[0]->earlyPartDone; // the backend launched earlyShadeDirect for this pair in this frame: the combine is what is left
    if (int rc = fusedShadeSetup(ctxs, count, &f, true, late ? &R : nullptr)) return rc;
    const PassCtx &u = *ctxs[0], &c = *ctxs[1];
    const ShadeParams& P = f.P;
    FusedUpscale& U = f.U;
    if (P.coverH <= P.yBase || P.coverW <= P.xBase) return 0;
    U.storeUpscaled = (u.elidableStorage & 3u) == 3u ? 0 : 1;
    if (!U.storeUpscaled) u.elidedStorage = 3u;
    const dim3 grid(divUp((unsigned)(P.coverW - P.xBase), 64u), divUp((unsigned)(P.coverH - P.yBase), 4u));
    if (late) {
        switch (f.multi) {
            case 0: upscaleAndCombineKernel<0><<<grid, 256, 0, c.stream>>>(P, U, R, P.g, P.light); break;
            case 1: upscaleAndCombineKernel<1><<<grid, 256, 0, c.stream>>>(P, U, R, P.g, P.light); break;
            case 2: upscaleAndCombineKernel<2><<<grid, 256, 0, c.stream>>>(P, U, R, P.g, P.light); break;
            default: upscaleAndCombineKernel<3><<<grid, 256, 0, c.stream>>>(P, U, R, P.g, P.light); break;
        }
        PLR_CHECK_LAUNCH(c);
        return 0;
    }
    FusedKernel k = nullptr;
    switch (f.diffuseBRDF) {
        case 0: k = pickFusedMulti<0>(f.multi, f.aa); break;
        case 1: k = pickFusedMulti<1>(f.multi, f.aa); break;
        case 2: k = pickFusedMulti<2>(f.multi, f.aa); break;
        default: k = pickFusedMulti<3>(f.multi, f.aa); break;
    }
    k<<<grid, 256, 0, c.stream>>>(P, U, PLR_SHADE_UNIFORM_ARGS(P));
    PLR_CHECK_LAUNCH(c);
    return 0;
}

// ------------------------------------------------------------------------------------------------ brdfLut.comp:20-101, PLR_MATH_FAST
// One lane per texel, the shader's 1024 Hammersley samples accumulated in the shader's order (the exact kernel: 4.7 ms, almost all of it
// software sin / cos / pow). Everything that depends on the sample index only is the same for every texel and is tabulated once per block in
// LDS: the GGX azimuth's sine / cosine, the second Hammersley coordinate, and the whole cosine-weighted direction of the diffuse sample (N is
// +z: the tangent frame of sampling.inc:4-45 is tangent = -y, bitangent = +x, so a sample h becomes (h.y, -h.x, h.z)). The loop body is then
// hardware sqrt / rcp / exp2 / log2 on per-texel terms; per-texel invariants (r, NoV) are hoisted by the compiler.
constexpr int kLutSamples = 1024;
template <int DIFFUSE_BRDF>
__global__ __launch_bounds__(256) void brdfLutFastKernel(ImgView lut, int coverW, int coverH) {
    __shared__ float4 ggx[kLutSamples];     // sin(phi), cos(phi), xi.y, -
    __shared__ float4 cosine[kLutSamples];  // L of the cosine-weighted sample
    for (int i = (int)threadIdx.x; i < kLutSamples; i += 256) {
        const float x0 = (float)i * (1.f / (float)kLutSamples);
        uint32_t bits = (uint32_t)i;
        bits = __builtin_bitreverse32(bits); // radicalInverse_VdC (sampling.inc: the five swaps are a 32-bit bit reversal)
        const float x1 = (float)bits * 2.3283064365386963e-10f;
        // the table is 1024 entries per block: the shader's own operations (software sin / cos, IEEE sqrt), so that every texel integrates over the
        // oracle's sample directions - the grazing, low-roughness texels sum a handful of spikes and follow every bit of the directions
        float sg, cg, sp, cp;
        det_sincosf(2.f * PLR_GLSL_PI * x0, &sg, &cg);
        det_sincosf(2.f * PLR_GLSL_PI * x1, &sp, &cp);
        ggx[i] = make_float4(sg, cg, x1, 0.f);
        const float cosT = sqrtf(x0), sinT = sqrtf(1.f - x0);
        cosine[i] = make_float4(sp * sinT, -(cp * sinT), cosT, 0.f);
    }
    __syncthreads();
    const int ux = (int)(blockIdx.x * 16u + (threadIdx.x & 15u)), uy = (int)(blockIdx.y * 16u + (threadIdx.x >> 4));
    if (ux >= coverW || uy >= coverH) return;
    const float r = fmax1((float)ux / (float)lut.w, 0.0001f);
    const float NoV = fmax1((float)uy, 0.1f) / (float)lut.h;
    const vec3 V(sqrtv(1.0f - NoV * NoV), 0.f, NoV);
    float r4m1;
    {
#pragma clang fp contract(off)
        const float r2 = r * r;
        r4m1 = r2 * r2 - 1.f;
    }
    const float fresnelOutV = 1.f - F_Schlick(vec3(0.04f), vec3(1.f), NoV).x;
    vec3 result(0.f);
    for (int i = 0; i < kLutSamples; i++) {
        const float4 gq = ggx[i], cq = cosine[i];
        {
            // the polar angle cancels catastrophically for low roughness (1 - cos^2 with cos within an ulp of 1): its operations are the
            // shader's, separately rounded, with a correctly rounded quotient and roots (Newton step on v_rcp / v_rsq), or the grazing rows
            // of the LUT end up tens of per cent away from the oracle's
            float cosT, sinT;
            {
#pragma clang fp contract(off)
                const float den = 1.f + r4m1 * gq.z, num = 1.f - gq.z;
                const float rd = rcpf(den), q0 = num * rd;
                const float q = __builtin_fmaf(__builtin_fmaf(-den, q0, num), rd, q0);
                auto root = [](float x) { const float r = rsqf(x), s0 = x * r; return x > 0.f ? __builtin_fmaf(__builtin_fmaf(-s0, s0, x), 0.5f * r, s0) : 0.f; };
                cosT = root(q);
                const float c2 = cosT * cosT;
                sinT = root(1.f - c2);
            }
            const vec3 H(gq.x * sinT, -(gq.y * sinT), cosT);
            const float VdotH = dot(V, H);
            const float Lz = 2.f * VdotH * H.z - V.z;
            const float VoH = fmax1(VdotH, 0.f), NoH = fmax1(H.z, 0.f), NoL = fmax1(Lz, 0.f);
            const float F_c = pow5(1.f - VoH);
            const float k = Visibility(NoV, NoL, r) * VoH * NoL * rcpf(NoH);
            const bool lit = NoL > 0.f;
            result.x += lit ? F_c * k : 0.f;
            result.y += lit ? k : 0.f;
        }
        {
            const vec3 L(cq.x, cq.y, cq.z);
            const vec3 H = nrm(V + L);
            const float VoH = fclamp(dot(V, H), 0.f, 1.f), NoL = fmax1(L.z, 0.f), NoH = fmax1(H.z, 0.f);
            const float fresnelInOut = fresnelOutV * (1.f - F_Schlick(vec3(0.04f), vec3(1.f), NoL).x);
            if (DIFFUSE_BRDF == 0) result.z += (1.f / PLR_GLSL_PI) * fresnelInOut;
            else if (DIFFUSE_BRDF == 1) result.z += DisneyDiffuse(vec3(1.f), NoL, VoH, NoV, r).x * fresnelInOut;
            else if (DIFFUSE_BRDF == 2) result.z += CoDWWIIDiffuse(vec3(1.f), NoL, VoH, NoV, NoH, r).x * fresnelInOut;
            else result.z += Titanfall2DiffuseSingleComponent(NoL, fclamp(dot(L, V), 0.f, 1.f), NoV, NoH, r) * fresnelInOut;
        }
    }
    result = result * (1.f / (float)kLutSamples);
    result.x *= 4.f;
    result.y *= 4.f;
    Texel<F_RGBA16F>::store(lut.ptr, (size_t)uy * (size_t)lut.w + ux, vec4(result, 0.f));
}

static int launchBrdfLutFast(const PassCtx& c) {
    if (!c.hasStorage(0) || c.storage[0].fmt != F_RGBA16F) return kUseGeneralKernel;
    const ImgView& lut = c.storage[0];
    const int w = std::min((int)(c.dispatch[0] * 8u), lut.w), h = std::min((int)(c.dispatch[1] * 8u), lut.h);
    if (w <= 0 || h <= 0) return 0;
    const dim3 grid(divUp((unsigned)w, 16u), divUp((unsigned)h, 16u));
    switch (c.specInt(0, 0)) {
        case 0: brdfLutFastKernel<0><<<grid, 256, 0, c.stream>>>(lut, w, h); break;
        case 1: brdfLutFastKernel<1><<<grid, 256, 0, c.stream>>>(lut, w, h); break;
        case 2: brdfLutFastKernel<2><<<grid, 256, 0, c.stream>>>(lut, w, h); break;
        case 3: brdfLutFastKernel<3><<<grid, 256, 0, c.stream>>>(lut, w, h); break;
        default: return kUseGeneralKernel;
    }
    PLR_CHECK_LAUNCH(c);
    return 0;
}

} // namespace fastshade

static int fastshade_launch(const PassCtx& c) { return fastshade::launchDeferredShadingFast(c); }
PLR_REGISTER_SHADER_FAST("deferredShading.comp", fastshade_launch);
static int fastshade_upscale_and_shade(const PassCtx* const* ctxs, size_t count) { return fastshade::launchUpscaleAndShade(ctxs, count); }
PLR_REGISTER_FUSION_WITH_SIGNATURES("indirectLightUpscale + deferredShading", fastshade_upscale_and_shade, "indirectLightUpscale.comp", "deferredShading.comp");
static int fastshade_early_direct(const PassCtx* const* ctxs, size_t count, EarlyPart& part) { return fastshade::earlyShadeDirect(ctxs, count, part); }
PLR_REGISTER_EARLY_PART(fastshade_upscale_and_shade, fastshade_early_direct, "direct lighting");
static int fastshade_brdf_lut(const PassCtx& c) { return fastshade::launchBrdfLutFast(c); }
PLR_REGISTER_SHADER_FAST("brdfLut.comp", fastshade_brdf_lut);
} // namespace plr
