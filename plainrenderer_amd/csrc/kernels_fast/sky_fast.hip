// PLR_MATH_FAST variants of the three sky LUT passes (skyTransmissionLut.comp, skyMultiscatterLut.comp, skyLut.comp; exact variants and the
// shaders' quirks: kernels/producers.hip, oracle/producers.cpp). Recorded every frame when the host asks for the LUTs as compute passes
// (Sky::updateTransmissionLut / updateSkyLut, Techniques/Sky.cpp:260-316; plrf_settings.run_sky_luts).
//
// The shaders run one invocation per LUT texel with a serial ray march inside: 16 k, 1 k and 20 k threads of 40 / 1280 / 30 dependent steps, each
// step a handful of exp(). On a 256-CU chip that is a few hundred waves with one long latency-bound instruction stream each - the multiscatter
// LUT's 1024 invocations run 64 x 20 steps = 0.6 M dependent instructions on sixteen waves. What is serial in them is only a running product
// (the transmittance along the ray) and a running sum; everything a step computes is independent of the other steps. So here
//   * skyTransmissionLut: a WAVE per texel, a lane per march step; the transmittance is exp(-step * sum of the extinctions): one DPP sum;
//   * skyLut: 32 lanes per texel (two texels per wave), a lane per march step; the transmittance in front of a step is the exclusive prefix
//     sum of the extinctions (DPP row scans + row broadcast), the colour one segmented sum;
//   * skyMultiscatterLut: the shader's inner march is a geometric series - its coefficients, LUT coordinates and step length do not depend on
//     the step (sic: height / currentHeight are evaluated outside the loop, skyMultiscatterLut.comp:94) - and its 8 azimuth samples per polar
//     angle are the same sample eight times (sic, :46), so a lane per texel replays the eight distinct marches with the invariants hoisted.
// exp / pow are v_exp_f32 / v_log_f32; the transmission LUT's march positions are origin + i * step instead of the accumulated sum (the sky
// LUT keeps the accumulated sum: its ground half amplifies the last bit, see the kernel). Outputs are R11G11B10: stated tolerance one code per
// channel (tests/test_producers.py).
// Built without FMA contraction and with IEEE "/" and sqrtf: the file calls detmath.h's sine / cosine and needs the exact set's bits from them
// (pragmas are lexical and do not reach the header), and the sky LUT's view ray is the shader's arithmetic bit for bit (see there); the LUTs are
// a few thousand waves, neither is what their time depends on. rcpf() / sqrtF() / expF() below stay the single hardware instructions.
// PLR_BUILD_FLAGS: -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt
#include "../backend.h"
#include "../device/shading_common.h"
#include "../device/fastmath.h"

namespace plr {
namespace fastsky {

struct Atmosphere { // sky.inc:1-10, std140: 56 bytes
    float scatteringRayleighGround[3], earthRadius;
    float extinctionRayleighGround[3], atmosphereHeight;
    float ozoneExtinction[3], scatteringMieGround;
    float extinctionMieGround, mieScatteringExponent;
};
struct Coefficients { vec3 scatterRayleigh, scatterMie, extinction; };

PLR_DI float expF(float x) { return __builtin_amdgcn_exp2f(x * 1.44269504089f); }
PLR_DI vec3 expF(vec3 v) { return vec3(expF(v.x), expF(v.y), expF(v.z)); }
PLR_DI float rcpf(float x) { return __builtin_amdgcn_rcpf(x); }
PLR_DI float sqrtF(float x) { return __builtin_amdgcn_sqrtf(x); }

PLR_DI Coefficients calculateCoefficients(float height, const Atmosphere& a) {
    const float rayleighFactor = expF(-height * (1.f / 8.f));
    const float mieFactor = expF(-height * (1.f / 1.2f));
    const float ozoneFactor = __builtin_fmaxf(0.f, 1.f - fabsf(height - 25.f) * (1.f / 15.f));
    Coefficients c;
    c.scatterRayleigh = rayleighFactor * ld3(a.scatteringRayleighGround);
    c.scatterMie = vec3(mieFactor * a.scatteringMieGround);
    c.extinction = rayleighFactor * ld3(a.extinctionRayleighGround) + vec3(mieFactor * a.extinctionMieGround) + ozoneFactor * ld3(a.ozoneExtinction);
    return c;
}

struct Intersection { vec3 pos; float distance; bool hitEarth; };
// sky.inc rayEarthIntersection with the earth's centre at the origin (every caller passes vec3(0)), in the shader's operation order with every
// operation rounded separately: "does the ray hit the earth" is a discrete decision, and rays that graze it exist on the LUT grids (a horizontal
// ray from height 0: t_earth is +-0 or NaN depending on the last bit of d * d) - a texel on the wrong side is black instead of lit
PLR_DI Intersection rayEarthIntersection(vec3 P, vec3 D, float earthRadius, float atmosphere) {
#pragma clang fp contract(off)
    const float lx = -P.x, ly = -P.y, lz = -P.z;
    const float t_ca = (lx * D.x + ly * D.y) + lz * D.z;
    const float d = sqrtf(((lx * lx + ly * ly) + lz * lz) - t_ca * t_ca);
    const float t_hc_earth = sqrtf(earthRadius * earthRadius - d * d);
    const float t_earth = t_ca - t_hc_earth; // NaN when the ray misses the earth: the comparison below is false, as in the shader
    const float r = earthRadius + atmosphere;
    const float t_hc_atmosphere = sqrtf(r * r - d * d);
    const float t_atmosphere = t_ca + fabsf(t_hc_atmosphere);
    Intersection result;
    result.hitEarth = t_earth >= 0.f;
    result.distance = result.hitEarth ? t_earth : t_atmosphere;
    result.pos = vec3(P.x + result.distance * D.x, P.y + result.distance * D.y, P.z + result.distance * D.z);
    return result;
}
// (inscattering - inscattering * e) / max(ext, 1e-5) with e = exp(-ext * length) supplied by the caller
PLR_DI vec3 integrateInscattering(vec3 inscattering, vec3 ext, vec3 e) {
    const vec3 num = inscattering - inscattering * e;
    return vec3(num.x * rcpf(__builtin_fmaxf(ext.x, 0.00001f)), num.y * rcpf(__builtin_fmaxf(ext.y, 0.00001f)), num.z * rcpf(__builtin_fmaxf(ext.z, 0.00001f)));
}
PLR_DI vec2 computeLutUV(float height, float atmosphereHeight, vec3 up, vec3 direction) { return vec2(height * rcpf(atmosphereHeight), dot(up, direction) * 0.5f + 0.5f); }

// ---- DPP sums. update_dpp(old, src, ...): a lane whose source lane is invalid or whose row is masked out keeps `old` (0: the sum's identity)
#define PLR_SKY_DPP_ADD(v, ctrl, rowMask) v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), ctrl, rowMask, 0xf, false))
// inclusive prefix sum within each 32-lane half of the wave (lanes 0..31, 32..63)
PLR_DI float scan32(float v) {
    PLR_SKY_DPP_ADD(v, 0x111, 0xf); PLR_SKY_DPP_ADD(v, 0x112, 0xf); PLR_SKY_DPP_ADD(v, 0x114, 0xf); PLR_SKY_DPP_ADD(v, 0x118, 0xf); // row_shr:1, 2, 4, 8: prefix within 16-lane rows
    PLR_SKY_DPP_ADD(v, 0x142, 0xa);                                                                                                // row_bcast:15: rows 1 and 3 add the total of rows 0 and 2
    return v;
}
// sum over all 64 lanes, valid in every lane
PLR_DI float sum64(float v) {
    v = scan32(v);
    PLR_SKY_DPP_ADD(v, 0x143, 0xc); // row_bcast:31: lanes 32..63 add the total of lanes 0..31
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}
#undef PLR_SKY_DPP_ADD

// ------------------------------------------------------------------------------------------------ skyTransmissionLut.comp:12-58
constexpr int kTransmissionSteps = 40;
__global__ __launch_bounds__(256) void skyTransmissionLutFastKernel(ImgView lut, const Atmosphere* __restrict__ ap, int coverW, int coverH) {
    const int texel = (int)(blockIdx.x * 4u + (threadIdx.x >> 6)), lane = (int)(threadIdx.x & 63u);
    if (texel >= coverW * coverH) return; // wave-uniform
    const int ux = texel % coverW, uy = texel / coverW;
    const Atmosphere a = *ap;
    const float x = (float)ux * rcpf((float)(lut.w - 1)), y = (float)uy * rcpf((float)(lut.h - 1));
    const float height = a.atmosphereHeight * x;
    const float upDot = __builtin_fmaxf(y * 2.f - 1.f, -0.999f);
    const vec3 V(0.f, -upDot, sqrtF(1.f - upDot * upDot));
    const vec3 P(0.f, -height - a.earthRadius, 0.f);
    const Intersection is = rayEarthIntersection(P - 0.01f, V, a.earthRadius, a.atmosphereHeight);
    const vec3 toP = is.pos - P;
    const float pathLength = __builtin_fmaxf(sqrtF(dot(toP, toP)), 0.01f);
    const float stepLength = pathLength * (1.f / (float)kTransmissionSteps);
    // lane i is march step i: position is.pos - (i + 1) * step; the transmittance is exp(-stepLength * sum of the steps' extinctions)
    vec3 ext(0.f);
    if (lane < kTransmissionSteps) {
        const vec3 pos = is.pos - V * (stepLength * (float)(lane + 1));
        const float currentHeight = __builtin_fmaxf(sqrtF(dot(pos, pos)) - a.earthRadius, 0.f);
        ext = calculateCoefficients(currentHeight, a).extinction;
    }
    const vec3 total(sum64(ext.x), sum64(ext.y), sum64(ext.z));
    if (lane != 0) return;
    const vec3 absorption = is.hitEarth ? vec3(0.f) : expF(total * -stepLength);
    ((uint32_t*)lut.ptr)[(size_t)uy * (size_t)lut.w + ux] = packR11G11B10(absorption);
}

// ------------------------------------------------------------------------------------------------ skyMultiscatterLut.comp:21-120
__global__ __launch_bounds__(64) void skyMultiscatterLutFastKernel(ImgView lut, ImgView transmissionLut, const Atmosphere* __restrict__ ap, int coverW, int coverH) {
    const int texel = (int)(blockIdx.x * 64u + threadIdx.x);
    if (texel >= coverW * coverH) return;
    const int ux = texel % coverW, uy = texel / coverW;
    const Atmosphere a = *ap;
    const float kPi = PLR_GLSL_PI;
    const float x = (float)ux * rcpf((float)lut.w), y = (float)uy * rcpf((float)lut.h);
    const float height = a.atmosphereHeight * x;
    const vec3 P(0.f, -height - a.earthRadius, 0.f);
    const float upDot = y * 2.f - 1.f;
    const vec3 L(0.f, -upDot, sqrtF(1.f - upDot * upDot));
    const float isotropicPhase = 1.f / (4.f * kPi);
    // everything the inner march evaluates per step is the same for all its steps and for all eight polar angles (sic, see the file header):
    const Coefficients c = calculateCoefficients(height, a);
    const vec3 scatteringCo = c.scatterRayleigh + c.scatterMie;
    const vec3 up(0.f, -1.f, 0.f);
    const float currentHeight = -P.y - a.earthRadius; // = height up to rounding
    const vec3 transmissionSun = sampleLinear2D<F_R11G11B10, CLAMP>(transmissionLut, computeLutUV(currentHeight, a.atmosphereHeight, up, L)).xyz();
    const vec3 up0 = P * rcpf(sqrtF(dot(P, P))); // normalize(P - earthCenter)
    const vec3 incomingLight = sampleLinear2D<F_R11G11B10, CLAMP>(transmissionLut, computeLutUV(0.f, a.atmosphereHeight, up0, L)).xyz();
    vec3 L_2nd(0.f), f_ms(0.f);
    for (int i = 0; i < 8; i++) {
        // the eight polar angles with the shader's own sine / cosine (detmath.h): at theta = pi / 2 its cosine is -4.4e-8, not 0, and the ray from
        // height 0 then grazes the earth on the other side of the hit test than a ray with an exactly horizontal direction
        float sinTheta, cosTheta;
        det_sincosf(kPi * (float)i * 0.125f, &sinTheta, &cosTheta);
        const vec3 V(sinTheta * cosTheta, -cosTheta, sinTheta * sinTheta); // sic (:46)
        const Intersection is = rayEarthIntersection(P, V, a.earthRadius, a.atmosphereHeight);
        const float stepSize = is.distance * (1.f / 20.f);
        const vec3 earthHitNormal = is.pos * rcpf(sqrtF(dot(is.pos, is.pos)));
        const float earthNoL = __builtin_amdgcn_fmed3f(dot(earthHitNormal, L), 0.f, 1.f);
        vec3 direct = is.hitEarth ? (0.3f / kPi) * incomingLight * earthNoL : vec3(0.f);
        const vec3 e = expF(c.extinction * -stepSize);
        const vec3 coefficientIntegral = integrateInscattering(scatteringCo, c.extinction, e);
        const vec3 scatterIntegral = coefficientIntegral * transmissionSun * isotropicPhase;
        vec3 L_f(0.f), inscattered(0.f), transmission(1.f);
#pragma unroll
        for (int k = 0; k < 20; k++) { // the shader's accumulation order
            L_f = L_f + coefficientIntegral * transmission;
            inscattered = inscattered + scatterIntegral * transmission;
            transmission = transmission * e;
        }
        direct = direct * transmission;
        const vec3 second = (direct * transmission + inscattered) * sinTheta, first = L_f * sinTheta;
#pragma unroll
        for (int j = 0; j < 8; j++) { f_ms = f_ms + first; L_2nd = L_2nd + second; } // the eight identical azimuth samples, added one by one as the shader does
    }
    f_ms = f_ms * (1.f / 64.f);
    L_2nd = L_2nd * (1.f / 64.f);
    const vec3 F_ms(rcpf(1.f - f_ms.x), rcpf(1.f - f_ms.y), rcpf(1.f - f_ms.z));
    ((uint32_t*)lut.ptr)[(size_t)uy * (size_t)lut.w + ux] = packR11G11B10(L_2nd * F_ms);
}

// ------------------------------------------------------------------------------------------------ skyLut.comp:26-123
constexpr int kSkySteps = 30;
__global__ __launch_bounds__(256) void skyLutFastKernel(ImgView lut, ImgView transmissionLut, ImgView multiscatterLut, const Atmosphere* __restrict__ ap,
                                                      const LightBuffer* __restrict__ light, const GlobalUbo* __restrict__ g, int coverW, int coverH) {
    const int lane = (int)(threadIdx.x & 63u), stepIndex = lane & 31;
    const int texel = (int)(blockIdx.x * 8u + (threadIdx.x >> 5));
    const int texelCount = coverW * coverH;
    const bool live = texel < texelCount; // the second half of the last wave may be idle: it still takes part in the DPP steps
    const int ux = live ? texel % coverW : 0, uy = live ? texel / coverW : 0;
    const Atmosphere a = *ap;
    const float kPi = PLR_GLSL_PI;
    // The view ray in the shader's own arithmetic, bit for bit (IEEE divisions, detmath.h's sine / cosine; the file is built without contraction):
    // the lower half of this LUT looks at the ground from 2 m above it, where the hit distance is t_ca - t_hc_earth - two numbers of a few
    // thousand km whose difference is a few metres, i.e. a handful of float ulps. The value IS that rounding noise (the colour is proportional
    // to it), so a direction that differs in its last bit gives a texel that differs by tens of codes.
    const float x = (float)ux / (float)lut.w, y = (float)uy / (float)lut.h;
    float theta = (1.f - y) - 0.5f;
    theta = gsign(theta) * theta * theta * 2.f;
    theta *= kPi;
    theta += kPi * 0.5f;
    const float phi = (-x + 0.5f) * 2.f * kPi;
    float sinT, cosT, sinP, cosP;
    det_sincosf(theta, &sinT, &cosT);
    det_sincosf(phi, &sinP, &cosP);
    const vec3 V(sinT * cosP, cosT, sinT * sinP);
    const float bias = 0.002f;
    const vec3 P(0.f, -a.earthRadius - bias, 0.f);
    const Intersection is = rayEarthIntersection(P, V, a.earthRadius, a.atmosphereHeight);
    const float stepSize = is.distance / (float)kSkySteps;
    const vec3 L = ld3(g->sunDirection);
    const float VoL = dot(V, L);
    const float phaseRayleigh = 3.f / (16.f * kPi) * (1.f + VoL * VoL);
    const float gM = a.mieScatteringExponent;
    const float nominator = 3.f / (8.f * kPi) * (1.f - gM * gM) * (1.f + VoL * VoL);
    const float base = 1.f + gM * gM - 2.f * gM * VoL;
    const float denominator = (2.f + gM * gM) * (base <= 0.f ? 0.f : __builtin_amdgcn_exp2f(1.5f * __builtin_amdgcn_logf(base)));
    const float phaseMie = nominator * rcpf(denominator);
    const float sunStrengthExposed = light->sunStrengthExposed;
    // lane = march step: everything but the transmittance in front of the step is local to it
    vec3 ext(0.f), scatterIntegral(0.f), multi(0.f);
    if (stepIndex < kSkySteps) {
        // the shader's accumulated march position (P + step + step + ...: 90 additions, nothing beside the exponentials below); the sun-side
        // earth shadow test is a discrete decision on it, in the shader's operation order
        const vec3 step = V * stepSize;
        vec3 pos = P;
        for (int k = 0; k <= stepIndex; k++) pos = pos + step;
        const float upLength = sqrtf(dot(pos, pos));
        const float currentHeight = upLength - a.earthRadius;
        const vec3 up = pos / upLength;
        const vec2 lutUV = computeLutUV(currentHeight, a.atmosphereHeight, up, L);
        const vec3 transmission = sampleLinear2D<F_R11G11B10, CLAMP>(transmissionLut, lutUV).xyz();
        const float t_ca = -dot(pos, L);
        const float dShadow = sqrtf(dot(pos, pos) - t_ca * t_ca);
        const float t_earth = t_ca - sqrtf(a.earthRadius * a.earthRadius - dShadow * dShadow);
        const vec3 incomingLight = (t_earth > 0.f ? 0.f : sunStrengthExposed) * transmission;
        const Coefficients c = calculateCoefficients(currentHeight, a);
        const vec3 inscattering = c.scatterRayleigh * incomingLight * phaseRayleigh + c.scatterMie * incomingLight * phaseMie;
        ext = c.extinction;
        // 1 - exp(-extinction * step) with the shader's own exponential: along the ground the steps are decimetres, the exponent is ~1e-6 and the
        // difference is a few dozen ulps of 1 - the LUT's value there is the rounding of that exponential, and v_exp_f32's differs by whole codes
        scatterIntegral = integrateInscattering(inscattering, ext, vec3(det_expf(-ext.x * stepSize), det_expf(-ext.y * stepSize), det_expf(-ext.z * stepSize)));
        const vec3 multiscattering = sampleLinear2D<F_R11G11B10, CLAMP>(multiscatterLut, lutUV).xyz();
        multi = multiscattering * incomingLight * (c.scatterRayleigh + c.scatterMie) * stepSize * transmission;
    }
    // transmittance in front of step i = exp(-stepSize * sum of the extinctions of steps 0 .. i-1): exclusive prefix sum within the 32-lane half
    const vec3 before(scan32(ext.x) - ext.x, scan32(ext.y) - ext.y, scan32(ext.z) - ext.z);
    const vec3 contribution = scatterIntegral * expF(before * -stepSize) + multi;
    const vec3 color(scan32(contribution.x), scan32(contribution.y), scan32(contribution.z)); // lane 31 of the half holds the sum over its steps
    if (live && stepIndex == 31) ((uint32_t*)lut.ptr)[(size_t)uy * (size_t)lut.w + ux] = packR11G11B10(color);
}

static int cover(const PassCtx& c, const ImgView& lut, int* w, int* h) {
    *w = std::min((int)(c.dispatch[0] * 8u), lut.w);
    *h = std::min((int)(c.dispatch[1] * 8u), lut.h);
    return *w > 0 && *h > 0;
}

static int launchTransmission(const PassCtx& c) {
    if (!c.hasStorage(0) || c.storage[0].fmt != F_R11G11B10 || !c.hasUbuf(1) || c.ubuf[1].size < sizeof(Atmosphere) || c.storage[0].w < 2 || c.storage[0].h < 2) return kUseGeneralKernel;
    int w, h;
    if (!cover(c, c.storage[0], &w, &h)) return 0;
    skyTransmissionLutFastKernel<<<divUp((unsigned)(w * h), 4u), 256, 0, c.stream>>>(c.storage[0], (const Atmosphere*)c.ubuf[1].ptr, w, h);
    PLR_CHECK_LAUNCH(c);
    return 0;
}
static int launchMultiscatter(const PassCtx& c) {
    if (!c.hasStorage(0) || c.storage[0].fmt != F_R11G11B10 || !c.hasSampled(1) || c.sampled[1].fmt != F_R11G11B10 || !c.hasUbuf(3) || c.ubuf[3].size < sizeof(Atmosphere)) return kUseGeneralKernel;
    int w, h;
    if (!cover(c, c.storage[0], &w, &h)) return 0;
    skyMultiscatterLutFastKernel<<<divUp((unsigned)(w * h), 64u), 64, 0, c.stream>>>(c.storage[0], c.sampled[1], (const Atmosphere*)c.ubuf[3].ptr, w, h);
    PLR_CHECK_LAUNCH(c);
    return 0;
}
static int launchSkyLut(const PassCtx& c) {
    if (!c.global || !c.hasStorage(0) || c.storage[0].fmt != F_R11G11B10 || !c.hasSampled(1) || c.sampled[1].fmt != F_R11G11B10 || !c.hasSampled(2) ||
        c.sampled[2].fmt != F_R11G11B10 || !c.hasUbuf(4) || c.ubuf[4].size < sizeof(Atmosphere) || !c.hasSbuf(5) || c.sbuf[5].size < sizeof(LightBuffer))
        return kUseGeneralKernel;
    int w, h;
    if (!cover(c, c.storage[0], &w, &h)) return 0;
    skyLutFastKernel<<<divUp((unsigned)(w * h), 8u), 256, 0, c.stream>>>(c.storage[0], c.sampled[1], c.sampled[2], (const Atmosphere*)c.ubuf[4].ptr, (const LightBuffer*)c.sbuf[5].ptr,
                                                                       c.global, w, h);
    PLR_CHECK_LAUNCH(c);
    return 0;
}

} // namespace fastsky

static int fastsky_transmission(const PassCtx& c) { return fastsky::launchTransmission(c); }
static int fastsky_multiscatter(const PassCtx& c) { return fastsky::launchMultiscatter(c); }
static int fastsky_lut(const PassCtx& c) { return fastsky::launchSkyLut(c); }
PLR_REGISTER_SHADER_FAST("skyTransmissionLut.comp", fastsky_transmission);
PLR_REGISTER_SHADER_FAST("skyMultiscatterLut.comp", fastsky_multiscatter);
PLR_REGISTER_SHADER_FAST("skyLut.comp", fastsky_lut);
} // namespace plr
