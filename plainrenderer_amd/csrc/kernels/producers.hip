// Input-producing compute passes: lightMatrix.comp. They are tiny fixed-size dispatches; the point of having them
// behind the same pass API is that the frame is closed under compute. Exact kernel set: operation order = the shader's (oracle/producers.cpp).
#include "../backend.h"
#include "../device/shading_common.h"

namespace plr {

struct M4 { float c[4][4]; }; // column major: c[col][row]
PLR_DI M4 mulM4(const M4& a, const M4& b) {
    M4 r;
    for (int col = 0; col < 4; col++)
        for (int row = 0; row < 4; row++) r.c[col][row] = a.c[0][row] * b.c[col][0] + a.c[1][row] * b.c[col][1] + a.c[2][row] * b.c[col][2] + a.c[3][row] * b.c[col][3];
    return r;
}

// lightMatrix.comp:57-137, local_size 1x1x1: one thread fits the cascades to the depth range of the frame (HiZ apex)
__global__ void lightMatrixKernel(ShadowCascadeInfo* __restrict__ info, ImgView apex, const GlobalUbo* __restrict__ g, int count, float highestCascadeExtraPadding,
                                  float highestCascadeMinFarPlane) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    const float FLOAT_MAX = 3.402823466e+38f, FLOAT_MIN = 1.175494351e-38f;
    const float shadowSampleRadius = 0.03f; // sunShadowCascades.inc:5
    M4 corr{};
    corr.c[0][0] = 1.f; corr.c[1][1] = 1.f; corr.c[2][2] = -0.5f; corr.c[3][2] = 0.5f; corr.c[3][3] = 1.f;
    const vec3 forward = -ld3(g->sunDirection);
    vec3 up = fabsf(forward.y) < 0.9999f ? vec3(0.f, -1.f, 0.f) : vec3(0.f, 0.f, -1.f);
    const vec3 right = cross(forward, up);
    up = cross(right, forward);
    const vec3 nr = normalize(right), nu = normalize(up);
    M4 V{};
    V.c[0][0] = nr.x; V.c[1][0] = nr.y; V.c[2][0] = nr.z;
    V.c[0][1] = nu.x; V.c[1][1] = nu.y; V.c[2][1] = nu.z;
    V.c[0][2] = forward.x; V.c[1][2] = forward.y; V.c[2][2] = forward.z;
    V.c[3][3] = 1.f;
    const float2 depthMinMax = ((const float2*)apex.ptr)[0]; // imageLoad(depthMinMaxLowestMip, ivec2(0)).rg
    const float depthMaxLinear = linearizeDepth(depthMinMax.x, g->nearPlane, g->farPlane);
    const float depthMinLinear = linearizeDepth(depthMinMax.y, g->nearPlane, g->farPlane);
    for (int i = 0; i < count - 1; i++) info->splits[i] = depthMinLinear + ((depthMaxLinear - depthMinLinear) * (float)(i + 1) / (float)count);
    const vec3 camPos = ld3(g->cameraPosition), camFwd = ld3(g->cameraForward), camUp = ld3(g->cameraUp), camRight = ld3(g->cameraRight);
    for (int i = 0; i < count; i++) {
        vec3 minP(FLOAT_MAX), maxP(FLOAT_MIN); // sic (:88-89)
        float cascadeMinDepth = i > 0 ? info->splits[i - 1] : 0.f;
        float cascadeMaxDepth = i < 4 ? info->splits[i] : 0.f;
        if (i == 0) cascadeMinDepth = depthMinLinear;
        if (i == count - 1) {
            cascadeMinDepth = g->nearPlane;
            cascadeMaxDepth = gmax(depthMaxLinear, highestCascadeMinFarPlane);
        }
        vec3 pts[8];
        const vec3 nearC = camPos + camFwd * cascadeMinDepth, farC = camPos + camFwd * cascadeMaxDepth;
        const float hN = g->cameraTanFovHalf * cascadeMinDepth, hF = g->cameraTanFovHalf * cascadeMaxDepth;
        const float wN = hN * g->cameraAspectRatio, wF = hF * g->cameraAspectRatio;
        pts[0] = farC + camUp * hF + camRight * wF; pts[1] = farC + camUp * hF - camRight * wF;
        pts[2] = farC - camUp * hF + camRight * wF; pts[3] = farC - camUp * hF - camRight * wF;
        pts[4] = nearC + camUp * hN + camRight * wN; pts[5] = nearC + camUp * hN - camRight * wN;
        pts[6] = nearC - camUp * hN + camRight * wN; pts[7] = nearC - camUp * hN - camRight * wN;
        for (int k = 0; k < 8; k++) {
            const vec3 p = pts[k];
            const vec3 t(V.c[0][0] * p.x + V.c[1][0] * p.y + V.c[2][0] * p.z + V.c[3][0] * 1.f, V.c[0][1] * p.x + V.c[1][1] * p.y + V.c[2][1] * p.z + V.c[3][1] * 1.f,
                         V.c[0][2] * p.x + V.c[1][2] * p.y + V.c[2][2] * p.z + V.c[3][2] * 1.f);
            minP = vec3(gmin(minP.x, t.x), gmin(minP.y, t.y), gmin(minP.z, t.z));
            maxP = vec3(gmax(maxP.x, t.x), gmax(maxP.y, t.y), gmax(maxP.z, t.z));
        }
        if (i == count - 1) { minP = minP - highestCascadeExtraPadding; maxP = maxP + highestCascadeExtraPadding; }
        minP = minP - shadowSampleRadius * 2.f;
        maxP = maxP + shadowSampleRadius * 2.f;
        const vec3 d = maxP - minP;
        const vec3 scale(2.f / d.x, 2.f / d.y, 2.f / d.z);
        const vec3 s = maxP + minP;
        const vec3 offset(-0.5f * s.x * scale.x, -0.5f * s.y * scale.y, -0.5f * s.z * scale.z);
        M4 P{};
        P.c[0][0] = scale.x; P.c[1][1] = scale.y; P.c[2][2] = scale.z;
        P.c[3][0] = offset.x; P.c[3][1] = offset.y; P.c[3][2] = offset.z; P.c[3][3] = 1.f;
        const M4 L = mulM4(mulM4(corr, P), V);
        for (int col = 0; col < 4; col++)
            for (int row = 0; row < 4; row++) info->lightMatrices[i][col * 4 + row] = L.c[col][row];
        info->lightSpaceScale[i][0] = scale.x;
        info->lightSpaceScale[i][1] = scale.y;
    }
}

static int launchLightMatrix(const PassCtx& c) {
    if (int rc = c.needGlobal()) return rc;
    if (int rc = c.needSbuf(0, sizeof(ShadowCascadeInfo), "lightMatrix sunShadowInfo")) return rc;
    if (int rc = c.needStorage(1, F_RG32F, "lightMatrix depthMinMaxLowestMip")) return rc;
    if (c.sbuf[0].readOnly) return c.fail(-4, "lightMatrix: sunShadowInfo is bound read-only");
    if (c.push.size() < 8) return c.fail(-1, "lightMatrix: push constants (padding, min far plane) missing");
    const int count = c.specInt(0, 4);
    if (count < 1 || count > 4) return c.fail(-1, "lightMatrix: sunShadowCascadeCount must be 1..4");
    float pc[2];
    std::memcpy(pc, c.push.data(), 8);
    if (c.dispatch[0] == 0 || c.dispatch[1] == 0 || c.dispatch[2] == 0) return 0;
    lightMatrixKernel<<<1, 64, 0, c.stream>>>((ShadowCascadeInfo*)c.sbuf[0].ptr, c.storage[1], c.global, count, pc[0], pc[1]);
    PLR_CHECK_LAUNCH(c);
    return 0;
}
PLR_REGISTER_SHADER("lightMatrix.comp", launchLightMatrix);
// one invocation by definition (local_size 1x1x1), and its output - cascade splits and matrices - feeds discrete decisions of the shade: both math modes
PLR_REGISTER_SHADER_FAST("lightMatrix.comp", launchLightMatrix);


// ====================================================================================================================
// Sky LUTs: sky.inc, volumeShading.inc, skyTransmissionLut.comp, skyMultiscatterLut.comp, skyLut.comp (oracle/producers.cpp has the
// same functions in the same operation order). 8x8 workgroups of the shaders = one 64-lane wave each.
namespace sky {

struct Atmosphere { // sky.inc:1-10, std140: 56 bytes
    float scatteringRayleighGround[3], earthRadius;
    float extinctionRayleighGround[3], atmosphereHeight;
    float ozoneExtinction[3], scatteringMieGround;
    float extinctionMieGround, mieScatteringExponent;
};
struct Coefficients { vec3 scatterRayleigh, scatterMie, extinction; };

PLR_DI Coefficients calculateCoefficients(float height, const Atmosphere& a) {
    const float rayleighFactor = det_expf(-height * (1.f / 8));
    const float mieFactor = det_expf(-height * (1.f / 1.2f));
    const float ozoneFactor = gmax(0.f, 1.f - fabsf(height - 25.f) / 15.f);
    Coefficients c;
    c.scatterRayleigh = rayleighFactor * ld3(a.scatteringRayleighGround);
    c.scatterMie = vec3(mieFactor) * a.scatteringMieGround;
    c.extinction = rayleighFactor * ld3(a.extinctionRayleighGround) + vec3(mieFactor * a.extinctionMieGround) + ozoneFactor * ld3(a.ozoneExtinction);
    return c;
}

struct Intersection { vec3 pos; float distance; bool hitEarth; };
PLR_DI Intersection rayEarthIntersection(vec3 P, vec3 D, vec3 C, float earthRadius, float atmosphere) {
    const vec3 L = C - P;
    const float t_ca = dot(L, D);
    const float d = sqrtf(dot(L, L) - t_ca * t_ca);
    const float t_hc_earth = sqrtf(earthRadius * earthRadius - d * d);
    const float t_earth = t_ca - t_hc_earth;
    const float r = earthRadius + atmosphere;
    const float t_hc_atmosphere = sqrtf(r * r - d * d);
    const float t_atmosphere = t_ca + fabsf(t_hc_atmosphere);
    Intersection result;
    result.hitEarth = t_earth >= 0.f;
    const float t = result.hitEarth ? t_earth : t_atmosphere;
    result.distance = t;
    result.pos = P + t * D;
    return result;
}
PLR_DI vec3 expv(vec3 v) { return vec3(det_expf(v.x), det_expf(v.y), det_expf(v.z)); }
PLR_DI vec3 integrateInscattering(vec3 inscattering, vec3 ext, float length) {
    const vec3 e = expv(-ext * length);
    const vec3 num = inscattering - inscattering * e;
    return vec3(num.x / gmax(ext.x, 0.00001f), num.y / gmax(ext.y, 0.00001f), num.z / gmax(ext.z, 0.00001f));
}
PLR_DI vec2 computeLutUV(float height, float atmosphereHeight, vec3 up, vec3 direction) { return vec2(height / atmosphereHeight, dot(up, direction) * 0.5f + 0.5f); }

__global__ __launch_bounds__(64) void skyTransmissionLutKernel(ImgView lut, const Atmosphere* __restrict__ ap, int coverW, int coverH) {
    const int ux = (int)(blockIdx.x * 8u + (threadIdx.x & 7u)), uy = (int)(blockIdx.y * 8u + (threadIdx.x >> 3));
    if (ux >= coverW || uy >= coverH) return;
    const Atmosphere a = *ap;
    const float x = (float)ux / (float)(lut.w - 1), y = (float)uy / (float)(lut.h - 1);
    const float height = 0.f * (1.f - x) + a.atmosphereHeight * x;
    float upDot = y * 2.f - 1.f;
    upDot = gmax(upDot, -0.999f);
    const vec3 V(0.f, -upDot, sqrtf(1.f - (upDot * upDot)));
    const vec3 P(0.f, -height - a.earthRadius, 0.f);
    const vec3 earthCenter(0.f);
    const Intersection is = rayEarthIntersection(P - 0.01f, V, earthCenter, a.earthRadius, a.atmosphereHeight);
    const float pathLength = gmax(distance(is.pos, P), 0.01f);
    const int sampleCount = 40;
    const float stepLength = pathLength / (float)sampleCount;
    vec3 currentPos = is.pos;
    vec3 absorption(1.f);
    const vec3 step = V * stepLength;
    for (int i = 0; i < sampleCount; i++) {
        currentPos = currentPos - step;
        const float currentHeight = gmax(distance(earthCenter, currentPos) - a.earthRadius, 0.f);
        const Coefficients c = calculateCoefficients(currentHeight, a);
        absorption = absorption * expv(-c.extinction * stepLength);
    }
    absorption = is.hitEarth ? vec3(0.f) : absorption;
    Texel<F_R11G11B10>::store(lut.ptr, (size_t)uy * (size_t)lut.w + ux, vec4(absorption, 0.f));
}

__global__ __launch_bounds__(64) void skyMultiscatterLutKernel(ImgView lut, ImgView transmissionLut, const Atmosphere* __restrict__ ap, int coverW, int coverH) {
    const int ux = (int)(blockIdx.x * 8u + (threadIdx.x & 7u)), uy = (int)(blockIdx.y * 8u + (threadIdx.x >> 3));
    if (ux >= coverW || uy >= coverH) return;
    const Atmosphere a = *ap;
    const float kPi = PLR_GLSL_PI;
    const float x = (float)ux / (float)lut.w, y = (float)uy / (float)lut.h;
    const float height = 0.f * (1.f - x) + a.atmosphereHeight * x;
    const vec3 P(0.f, -height - a.earthRadius, 0.f);
    const vec3 earthCenter(0.f);
    const float upDot = y * 2.f - 1.f;
    const vec3 L(0.f, -upDot, sqrtf(1.f - (upDot * upDot)));
    vec3 L_2nd(0.f), f_ms(0.f);
    const float isotropicPhase = 1.f / (4.f * kPi);
    const int sampleCountSqrt = 8;
    const float sampleCountSqrtRcp = 1.f / (float)sampleCountSqrt;
    for (int i = 0; i < sampleCountSqrt; i++)
        for (int j = 0; j < sampleCountSqrt; j++) {
            const float theta = kPi * (float)i * sampleCountSqrtRcp;
            const float sinTheta = det_sinf(theta), cosTheta = det_cosf(theta);
            vec3 V(sinTheta * cosTheta, -cosTheta, sinTheta * sinTheta); // sic (:46)
            const int innerSampleCount = 20;
            vec3 inscattered(0.f);
            const Intersection is = rayEarthIntersection(P, V, earthCenter, a.earthRadius, a.atmosphereHeight);
            vec3 currentPosition = P;
            const float stepSize = is.distance / (float)innerSampleCount;
            V = V * stepSize;
            vec3 L_f(0.f);
            const vec3 earthAlbedo(0.3f);
            const vec3 earthHitNormal = normalize(is.pos - earthCenter);
            const float earthNoL = gclamp(dot(earthHitNormal, L), 0.f, 1.f);
            const vec3 up0 = normalize(currentPosition - earthCenter);
            const vec2 lutUV0 = computeLutUV(0.f, a.atmosphereHeight, up0, L);
            const vec3 incomingLight = sampleLinear2D<F_R11G11B10, CLAMP>(transmissionLut, lutUV0).xyz();
            const vec3 earthLit = earthAlbedo / kPi * incomingLight * earthNoL;
            vec3 direct = is.hitEarth ? earthLit : vec3(0.f);
            vec3 transmission(1.f);
            const float currentHeight = -currentPosition.y - a.earthRadius;
            for (int k = 0; k < innerSampleCount; k++) {
                currentPosition = currentPosition + V;
                const vec3 up(0.f, -1.f, 0.f);
                const Coefficients c = calculateCoefficients(height, a); // sic (:94)
                const vec3 scatteringCo = c.scatterRayleigh + c.scatterMie;
                const vec2 lutUV = computeLutUV(currentHeight, a.atmosphereHeight, up, L);
                const vec3 transmissionSun = sampleLinear2D<F_R11G11B10, CLAMP>(transmissionLut, lutUV).xyz();
                const vec3 coefficientIntegral = integrateInscattering(scatteringCo, c.extinction, stepSize);
                L_f = L_f + coefficientIntegral * transmission;
                const vec3 scatterIntegral = coefficientIntegral * transmissionSun * isotropicPhase;
                inscattered = inscattered + scatterIntegral * transmission;
                transmission = transmission * expv(-c.extinction * stepSize);
            }
            direct = direct * transmission;
            f_ms = f_ms + L_f * sinTheta;
            L_2nd = L_2nd + (direct * transmission + inscattered) * sinTheta;
        }
    const float sampleCountInverse = 1.f / (float)(sampleCountSqrt * sampleCountSqrt);
    f_ms = f_ms * sampleCountInverse;
    L_2nd = L_2nd * sampleCountInverse;
    const vec3 F_ms(1.f / (1.f - f_ms.x), 1.f / (1.f - f_ms.y), 1.f / (1.f - f_ms.z));
    Texel<F_R11G11B10>::store(lut.ptr, (size_t)uy * (size_t)lut.w + ux, vec4(L_2nd * F_ms, 0.f));
}

__global__ __launch_bounds__(64) void skyLutKernel(ImgView lut, ImgView transmissionLut, ImgView multiscatterLut, const Atmosphere* __restrict__ ap,
                                                  const LightBuffer* __restrict__ light, const GlobalUbo* __restrict__ g, int coverW, int coverH) {
    const int ux = (int)(blockIdx.x * 8u + (threadIdx.x & 7u)), uy = (int)(blockIdx.y * 8u + (threadIdx.x >> 3));
    if (ux >= coverW || uy >= coverH) return;
    const Atmosphere a = *ap;
    const float kPi = PLR_GLSL_PI;
    const float x = (float)ux / (float)lut.w, y = (float)uy / (float)lut.h;
    float theta = (1.f - y) - 0.5f;
    theta = gsign(theta) * theta * theta * 2.f;
    theta *= kPi;
    theta += kPi * 0.5f;
    const float phi = (-x + 0.5f) * 2.f * kPi;
    const vec3 V(det_sinf(theta) * det_cosf(phi), det_cosf(theta), det_sinf(theta) * det_sinf(phi));
    const vec3 earthCenter(0.f);
    const float bias = 0.002f;
    const vec3 P(0.f, -a.earthRadius - bias, 0.f);
    const Intersection is = rayEarthIntersection(P, V, earthCenter, a.earthRadius, a.atmosphereHeight);
    const int sampleCount = 30;
    const float stepSize = is.distance / (float)sampleCount;
    const vec3 L = ld3(g->sunDirection);
    const float VoL = dot(V, L);
    const float phaseRayleigh = 3.f / (16.f * kPi) * (1.f + VoL * VoL);
    const float gM = a.mieScatteringExponent;
    const float nominator = 3.f / (8.f * kPi) * (1.f - gM * gM) * (1.f + VoL * VoL);
    const float denominator = (2.f + gM * gM) * det_powf(1.f + gM * gM - 2.f * gM * VoL, 1.5f);
    const float phaseMie = nominator / denominator;
    vec3 currentPosition = P;
    vec3 absorption(1.f), color(0.f);
    const vec3 step = V * stepSize;
    const float sunStrengthExposed = light->sunStrengthExposed;
    for (int i = 0; i < sampleCount; i++) {
        currentPosition = currentPosition + step;
        vec3 up = currentPosition - earthCenter;
        const float upLength = length(up);
        const float currentHeight = upLength - a.earthRadius;
        up = up / upLength;
        const vec2 lutUV = computeLutUV(currentHeight, a.atmosphereHeight, up, L);
        const vec3 transmission = sampleLinear2D<F_R11G11B10, CLAMP>(transmissionLut, lutUV).xyz();
        vec3 incomingLight = sunStrengthExposed * transmission;
        {
            const vec3 Lc = earthCenter - currentPosition;
            const float t_ca = dot(Lc, L);
            const float d = sqrtf(dot(Lc, Lc) - t_ca * t_ca);
            const float t_hc_earth = sqrtf(a.earthRadius * a.earthRadius - d * d);
            const float t_earth = t_ca - t_hc_earth;
            incomingLight = incomingLight * (t_earth > 0.f ? 0.f : 1.f);
        }
        const Coefficients c = calculateCoefficients(currentHeight, a);
        const vec3 inscattering = c.scatterRayleigh * incomingLight * phaseRayleigh + c.scatterMie * incomingLight * phaseMie;
        const vec3 scatterIntegral = integrateInscattering(inscattering, c.extinction, stepSize);
        color = color + scatterIntegral * absorption;
        absorption = absorption * expv(-c.extinction * stepSize);
        const vec3 multiscattering = sampleLinear2D<F_R11G11B10, CLAMP>(multiscatterLut, lutUV).xyz();
        color = color + multiscattering * incomingLight * (c.scatterRayleigh + c.scatterMie) * stepSize * transmission;
    }
    Texel<F_R11G11B10>::store(lut.ptr, (size_t)uy * (size_t)lut.w + ux, vec4(color, 0.f));
}

static int cover(const PassCtx& c, const ImgView& lut, int* w, int* h) {
    *w = std::min((int)(c.dispatch[0] * 8u), lut.w);
    *h = std::min((int)(c.dispatch[1] * 8u), lut.h);
    return *w > 0 && *h > 0;
}

static int launchTransmission(const PassCtx& c) {
    if (int rc = c.needStorage(0, F_R11G11B10, "skyTransmissionLut lut")) return rc;
    if (int rc = c.needUbuf(1, sizeof(Atmosphere), "skyTransmissionLut atmosphereSettingsBuffer")) return rc;
    int w, h;
    if (!cover(c, c.storage[0], &w, &h)) return 0;
    if (c.storage[0].w < 2 || c.storage[0].h < 2) return c.fail(-4, "skyTransmissionLut: the lut needs at least 2x2 texels (coordinates divide by size - 1)");
    skyTransmissionLutKernel<<<dim3(divUp((unsigned)w, 8u), divUp((unsigned)h, 8u)), 64, 0, c.stream>>>(c.storage[0], (const Atmosphere*)c.ubuf[1].ptr, w, h);
    PLR_CHECK_LAUNCH(c);
    return 0;
}
static int launchMultiscatter(const PassCtx& c) {
    if (int rc = c.needStorage(0, F_R11G11B10, "skyMultiscatterLut multiscatterLut")) return rc;
    if (int rc = c.needSampled(1, F_R11G11B10, "skyMultiscatterLut transmissionLut")) return rc;
    if (int rc = c.needUbuf(3, sizeof(Atmosphere), "skyMultiscatterLut atmosphereSettingsBuffer")) return rc;
    int w, h;
    if (!cover(c, c.storage[0], &w, &h)) return 0;
    skyMultiscatterLutKernel<<<dim3(divUp((unsigned)w, 8u), divUp((unsigned)h, 8u)), 64, 0, c.stream>>>(c.storage[0], c.sampled[1], (const Atmosphere*)c.ubuf[3].ptr, w, h);
    PLR_CHECK_LAUNCH(c);
    return 0;
}
static int launchSkyLut(const PassCtx& c) {
    if (int rc = c.needGlobal()) return rc;
    if (int rc = c.needStorage(0, F_R11G11B10, "skyLut skyLut")) return rc;
    if (int rc = c.needSampled(1, F_R11G11B10, "skyLut transmissionLut")) return rc;
    if (int rc = c.needSampled(2, F_R11G11B10, "skyLut multiscatterLut")) return rc;
    if (int rc = c.needUbuf(4, sizeof(Atmosphere), "skyLut atmosphereSettingsBuffer")) return rc;
    if (int rc = c.needSbuf(5, sizeof(LightBuffer), "skyLut lightStorageBuffer")) return rc;
    int w, h;
    if (!cover(c, c.storage[0], &w, &h)) return 0;
    skyLutKernel<<<dim3(divUp((unsigned)w, 8u), divUp((unsigned)h, 8u)), 64, 0, c.stream>>>(c.storage[0], c.sampled[1], c.sampled[2], (const Atmosphere*)c.ubuf[4].ptr,
                                                                                             (const LightBuffer*)c.sbuf[5].ptr, c.global, w, h);
    PLR_CHECK_LAUNCH(c);
    return 0;
}

} // namespace sky

static int sky_transmission_launch(const PassCtx& c) { return sky::launchTransmission(c); }
static int sky_multiscatter_launch(const PassCtx& c) { return sky::launchMultiscatter(c); }
static int sky_lut_launch(const PassCtx& c) { return sky::launchSkyLut(c); }
PLR_REGISTER_SHADER("skyTransmissionLut.comp", sky_transmission_launch);
PLR_REGISTER_SHADER("skyMultiscatterLut.comp", sky_multiscatter_launch);
PLR_REGISTER_SHADER("skyLut.comp", sky_lut_launch);


// ====================================================================================================================
// Volumetric froxel lighting (oracle/producers.cpp has the same functions in the same operation order): froxelVolumeMaterial.comp,
// froxelLightScattering.comp, volumeLightingReprojection.comp (4x4x4 workgroups = one 64-lane wave), volumetricLightingIntegration.comp.
namespace froxel {

struct VolSettings { // volumetricFroxelLighting.inc:6-16, 52 bytes
    float windSampleOffset[3], sampleOffset;
    float scatteringCoefficients[3], maxDistance;
    float absorptionCoefficient, baseDensity, densityNoiseRange, densityNoiseScale, phaseFunctionG;
};

constexpr float kFroxelK = 3.f;
PLR_DI float froxelUVToDepth(float uvZ, float maxDistance) {
    const float remaped = (det_expf(kFroxelK * uvZ) - 1.f) / (det_expf(kFroxelK) - 1.f);
    return remaped * maxDistance;
}
PLR_DI float depthToFroxelUVZ(float depth, float maxDistance) {
    const float linear = depth / maxDistance;
    return det_logf(linear * (det_expf(kFroxelK) - 1.f) + 1.f) / kFroxelK;
}

// the froxel's view ray scaled to unit depth, V / dot(-V, forward): the part of the world position that depends on (x, y) only
PLR_DI vec3 froxelRayPerDepth(int x, int y, const ImgView& vol, float jitter, const GlobalUbo* g, vec3* Vout, bool ndcForm2) {
    const vec2 uv(((float)x + 0.5f + jitter) / (float)vol.w, ((float)y + 0.5f + jitter) / (float)vol.h);
    const vec2 ndc = ndcForm2 ? vec2(2.f * uv.x - 1.f, 2.f * uv.y - 1.f) : vec2(2.f * (uv.x - 0.5f), 2.f * (uv.y - 0.5f));
    const vec3 fwd = ld3(g->cameraForward);
    const vec3 V = calculateViewDirectionFromPixel(ndc, fwd, ld3(g->cameraUp), ld3(g->cameraRight), g->cameraTanFovHalf, g->cameraAspectRatio);
    if (Vout) *Vout = V;
    return V / dot(-V, fwd);
}
// the part that depends on z only
PLR_DI float froxelSliceDepth(int z, const ImgView& vol, float jitter, float maxDistance) { return froxelUVToDepth(((float)z + 0.5f + jitter) / (float)vol.d, maxDistance); }
PLR_DI vec3 froxelWorldPosition(int x, int y, int z, const ImgView& vol, float jitter, const GlobalUbo* g, float maxDistance, vec3* Vout, bool ndcForm2) {
    return ld3(g->cameraPosition) - froxelRayPerDepth(x, y, vol, jitter, g, Vout, ndcForm2) * froxelSliceDepth(z, vol, jitter, maxDistance);
}

PLR_DI size_t idx3(const ImgView& im, int x, int y, int z) { return ((size_t)z * (size_t)im.h + (size_t)y) * (size_t)im.w + (size_t)x; }

// thread -> froxel for the 4x4x4-workgroup passes: blocks of 64 lanes walk x fastest
PLR_DI bool froxelOfThread(const ImgView& vol, int coverX, int coverY, int coverZ, int* x, int* y, int* z) {
    // 32-bit index arithmetic (the launcher refuses volumes of 2^31 froxels or more): the 64-bit divisions this replaced were software sequences of
    // several hundred instructions each, three per thread
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    const uint32_t n = (uint32_t)coverX * (uint32_t)coverY * (uint32_t)coverZ;
    if (i >= n) return false;
    const uint32_t row = i / (uint32_t)coverX;
    *x = (int)(i - row * (uint32_t)coverX);
    *z = (int)(row / (uint32_t)coverY);
    *y = (int)(row - (uint32_t)*z * (uint32_t)coverY);
    return true;
}

__global__ __launch_bounds__(256) void froxelVolumeMaterialKernel(ImgView material, ImgView noiseTexture, const VolSettings* __restrict__ sp, const GlobalUbo* __restrict__ g,
                                                                  int cx, int cy, int cz) {
    int x, y, z;
    if (!froxelOfThread(material, cx, cy, cz, &x, &y, &z)) return;
    const VolSettings s = *sp;
    const vec3 posWorld = froxelWorldPosition(x, y, z, material, s.sampleOffset, g, s.maxDistance, nullptr, false);
    const float noiseScale = 0.5f;
    const vec3 noiseSample = posWorld * noiseScale + ld3(s.windSampleOffset);
    const float noise = sampleLinear3D<F_R8, REPEAT>(noiseTexture, noiseSample).x;
    vec3 scatteringCoefficient = ld3(s.scatteringCoefficients);
    float absorptionCoefficient = s.absorptionCoefficient;
    float densityMultiplier = s.baseDensity;
    densityMultiplier += s.densityNoiseRange * (noise - 0.5f);
    densityMultiplier = gmax(densityMultiplier, 0.f);
    scatteringCoefficient = scatteringCoefficient * densityMultiplier;
    absorptionCoefficient *= densityMultiplier;
    Texel<F_RGBA16F>::store(material.ptr, idx3(material, x, y, z), vec4(scatteringCoefficient, absorptionCoefficient));
}

__global__ __launch_bounds__(256) void froxelLightScatteringKernel(ImgView out, ImgView shadowMap, ImgView material, const ShadowCascadeInfo* __restrict__ shadowInfo,
                                                                   const LightBuffer* __restrict__ light, const VolSettings* __restrict__ sp, const GlobalUbo* __restrict__ g,
                                                                   int cx, int cy, int cz) {
    int x, y, z;
    if (!froxelOfThread(out, cx, cy, cz, &x, &y, &z)) return;
    const VolSettings s = *sp;
    const float kPi = PLR_GLSL_PI;
    vec3 V;
    const vec3 posWorld = froxelWorldPosition(x, y, z, out, s.sampleOffset, g, s.maxDistance, &V, true);
    vec4 p = mulMat4(shadowInfo->lightMatrices[2], vec4(posWorld, 1.f)); // sic: cascade 2 (:43)
    p = p / p.w;
    const float actualDepth = gclamp(p.z, 0.f, 1.f);
    const float shadowMapDepth = sampleNearest2D<F_D16, BORDER_BLACK>(shadowMap, vec2(p.x, p.y) * 0.5f + 0.5f).x;
    const float shadow = actualDepth > shadowMapDepth ? 1.f : 0.f;
    const float sunStrength = shadow * light->sunStrengthExposed;
    const vec3 L = ld3(g->sunDirection);
    const float VoL = dot(-V, L);
    const float gg = s.phaseFunctionG;
    const float phase = (1.f - gg * gg) / (4.f * kPi * det_powf(1.f + gg * gg - 2.f * gg * VoL, 1.5f));
    const bool inMaterial = x < material.w && y < material.h && z < material.d;
    const vec4 sa = inMaterial ? Texel<F_RGBA16F>::load(material.ptr, idx3(material, x, y, z)) : vec4(0.f);
    const vec3 scatteringCoefficient = sa.xyz();
    const float absorptionCoefficient = sa.w;
    const vec3 constantAmbientLighting(0.02f);
    const vec3 inscattering = (sunStrength * phase * ld3(light->sunColor) + constantAmbientLighting) * scatteringCoefficient;
    const vec3 extinctionCoefficient = scatteringCoefficient + absorptionCoefficient;
    const float transmittance = computeLuminance(extinctionCoefficient);
    Texel<F_RGBA16F>::store(out.ptr, idx3(out, x, y, z), vec4(inscattering, transmittance));
}

__global__ __launch_bounds__(256) void volumeLightingReprojectionKernel(ImgView target, ImgView inputVolume, ImgView historyVolume, const VolSettings* __restrict__ sp,
                                                                        const GlobalUbo* __restrict__ g, int cx, int cy, int cz) {
    int x, y, z;
    if (!froxelOfThread(target, cx, cy, cz, &x, &y, &z)) return;
    const VolSettings s = *sp;
    const bool inInput = x < inputVolume.w && y < inputVolume.h && z < inputVolume.d;
    const vec4 current = inInput ? Texel<F_RGBA16F>::load(inputVolume.ptr, idx3(inputVolume, x, y, z)) : vec4(0.f);
    const vec3 posWorld = froxelWorldPosition(x, y, z, target, 0.f, g, s.maxDistance, nullptr, false);
    vec4 ndcPrevious = mulMat4(g->viewProjectionPrevious, vec4(posWorld, 1.f));
    ndcPrevious = vec4(ndcPrevious.x / ndcPrevious.w, ndcPrevious.y / ndcPrevious.w, ndcPrevious.z / ndcPrevious.w, ndcPrevious.w);
    const vec3 camPrev = ld3(g->cameraPositionPrevious);
    const vec3 V_history = normalize(camPrev - posWorld);
    const float historyDistance = distance(posWorld, camPrev);
    const float historyDepth = historyDistance * dot(-V_history, ld3(g->cameraForwardPrevious));
    const vec3 historyUV(ndcPrevious.x * 0.5f + 0.5f, ndcPrevious.y * 0.5f + 0.5f, depthToFroxelUVZ(historyDepth, s.maxDistance));
    vec4 history = sampleLinear3D<F_RGBA16F, CLAMP>(historyVolume, historyUV);
    float alpha = 0.95f;
    if (historyUV.x > 1.f || historyUV.y > 1.f || historyUV.z > 1.f || historyUV.x < 0.f || historyUV.y < 0.f || historyUV.z < 0.f) alpha = 0.f;
    if (g->cameraCut) history = current;
    const vec4 result = current * (1.f - alpha) + history * alpha;
    Texel<F_RGBA16F>::store(target.ptr, idx3(target, x, y, z), result);
}

// ---- the three per-froxel passes as ONE launch (PLR_MATH_FAST + pass fusion; plr_set_pass_fusion). Each pass reads the previous one's volume at
// its OWN froxel only, so a thread carries the texel through registers - rounded to half floats exactly where the stored volume would round it -
// and the result equals the three separate launches bit for bit (tests/test_producers.py, tests/test_fusion.py). What it saves: two launches,
// the material + scattering volumes' round trip through HBM (4 x 8 bytes per froxel; with fusion level 2 their stores go as well when nothing
// else in the frame binds them), and one of the three world-position chains: froxelVolumeMaterial.comp:34 writes the NDC as 2 * (uv - 0.5),
// froxelLightScattering.comp:36 as 2 * uv - 1 - the same float for every uv in [0, 1) (both subtractions round the exact 2 uv - 1 onto the same
// grid: for uv >= 0.25 both are exact by Sterbenz, below it the product by two is exact), and both passes use the same jitter. The reprojection
// pass evaluates the position without jitter: its chain stays its own.
// The arithmetic is the exact set's: sub-texel weights of the noise / history samples and the shadow-map texel are discrete in the position.
// sampleLinear3D<F_R8, REPEAT>(noise, uvw).x (device/image.h) with the wrap of a power-of-two extent as a mask: the signed modulo of repeati is a software
// sequence of some 38 instructions, six of them per sample - a quarter of the column walk's slice step. Same texels, same values, same weights, same term order.
PLR_DI float sampleNoiseRepeat(const ImgView& im, vec3 uvw) {
    int i0, j0, k0; float a, b, c;
    linearCoord(uvw.x * (float)im.w, &i0, &a);
    linearCoord(uvw.y * (float)im.h, &j0, &b);
    linearCoord(uvw.z * (float)im.d, &k0, &c);
    int x0, x1, y0, y1, z0, z1;
    if ((((im.w & (im.w - 1)) | (im.h & (im.h - 1)) | (im.d & (im.d - 1))) == 0)) { // wave-uniform (two's complement: the mask is the non-negative remainder)
        x0 = i0 & (im.w - 1); x1 = (i0 + 1) & (im.w - 1); y0 = j0 & (im.h - 1); y1 = (j0 + 1) & (im.h - 1); z0 = k0 & (im.d - 1); z1 = (k0 + 1) & (im.d - 1);
    } else {
        x0 = repeati(i0, im.w); x1 = repeati(i0 + 1, im.w); y0 = repeati(j0, im.h); y1 = repeati(j0 + 1, im.h); z0 = repeati(k0, im.d); z1 = repeati(k0 + 1, im.d);
    }
    // (decodeUnorm8Newton: c / 255 correctly rounded for all 256 codes in three instructions instead of the IEEE division's ten, tests/test_gpu_foundations.py)
    auto T = [&](int x, int y, int z) { return decodeUnorm8Newton(((const uint8_t*)im.ptr)[((size_t)z * (size_t)im.h + (size_t)y) * (size_t)im.w + (size_t)x]); };
    const float a0 = 1.f - a, b0 = 1.f - b, c0 = 1.f - c;
    float r = 0.f;
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
PLR_DI float roundToHalf(float v) { return halfBitsToFloat(floatToHalfBits(v)); } // what a texel of an RGBA16F volume gives back
// Round 4: a thread walks a COLUMN SEGMENT of kFroxelSegment slices at one (x, y). A good third of the three shaders' arithmetic depends on (x, y) or on z alone -
// per (x, y) the two view rays (two normalisations, ten IEEE divisions) and the phase function (a software pow and a division), per z the slice depths (an
// exponential and two divisions each) - and is evaluated once per column / once per block and slice (LDS) by the same functions in the same order: the
// bits stay (tests/test_producers.py, tests/test_fusion.py). A wave still covers 64 consecutive x of one slice per step: loads and stores coalesce as before.
// Slices per thread, measured at 4K (480 x 270 x 64 froxels): 8 -> 128 us, 16 -> 133, 32 -> 143, 64 (a whole column) -> 177: the column set-up is ~740 instructions against ~600
// per slice, but the shorter segment keeps twice the waves in flight behind the eight dependent loads of the history sample.
constexpr int kFroxelSegment = 8;
constexpr int kFroxelMaxSegments = 32; // INTEGRATE: a column's segments share a block
// INTEGRATE: volumetricLightingIntegration.comp:15-45 in the same launch (the fourth pass of Volumetrics::computeVolumetricLighting). The front-to-back sums of a
// column are a scan over exactly the texels this launch produces, so the block holds ALL segments of its columns (thread = (column, segment), columnsPerBlock x
// segments <= 256): a thread accumulates its segment's totals while it walks it, the block exchanges them through LDS, and a second walk over the thread's own
// eight texels - read back from the target volume it has just written - adds the segments in front of it. Arithmetic of the integration: the fast set's
// (kernels_fast/froxel_fast.hip: one hardware exponential per slice, no decision in this pass; within the half-float bound of tests/parity.py), the three per-froxel
// passes keep the exact set's. What it saves: a launch whose 2 k waves walked 64 dependent slices each (27 us for 130 MB), and the read of the volume from HBM.
PLR_DI float froxelFastExp(float x) { return __builtin_amdgcn_exp2f(x * 1.44269504089f); }
struct FroxelSliceTerm { float r, g, b, e; }; // a slice's added inscattering and its transmittance factor
PLR_DI FroxelSliceTerm froxelSliceTerm(vec4 texel, float segmentLength) { // texel: (inscattering, extinction) as the volume stores it; the slice's extent in depth
    const float e = froxelFastExp(-texel.w * segmentLength);
    const float kk = (1.f - e) * __builtin_amdgcn_rcpf(__builtin_fmaxf(texel.w, 0.00001f)); // integrateInscattering: (s - s e) / max(ext, 1e-5)
    return {texel.x * kk, texel.y * kk, texel.z * kk, e};
}
template <bool STORE_INTERMEDIATES, bool INTEGRATE>
__global__ __launch_bounds__(256) void froxelFrontFusedKernel(ImgView material, ImgView noiseTexture, ImgView scattering, ImgView shadowMap, const ShadowCascadeInfo* __restrict__ shadowInfo,
                                                              const LightBuffer* __restrict__ light, ImgView target, ImgView historyVolume, const VolSettings* __restrict__ sp,
                                                              const GlobalUbo* __restrict__ g, int cx, int cy, int cz, ImgView integrationVolume, int columnsPerBlock, int segments) {
    // [0]: with the frame's jitter (material, scattering), [1]: without (reprojection); the block's own segment, or (INTEGRATE) every slice of the volume
    __shared__ float sliceDepth[2][INTEGRATE ? kFroxelMaxSegments * kFroxelSegment : kFroxelSegment];
    __shared__ float segmentTotals[INTEGRATE ? 4 : 1][INTEGRATE ? 256 : 1];
    __shared__ float sliceLength[INTEGRATE ? kFroxelMaxSegments * kFroxelSegment : 1]; // froxelUVToDepth((z + 1) / d) - froxelUVToDepth(z / d), the integration's (hardware exponentials)
    const VolSettings s = *sp;
    const int segment = INTEGRATE ? (int)(threadIdx.x / (uint32_t)columnsPerBlock) : (int)blockIdx.y;
    const int columnInBlock = INTEGRATE ? (int)threadIdx.x - segment * columnsPerBlock : (int)threadIdx.x;
    const int z0 = segment * kFroxelSegment, depthBase = INTEGRATE ? 0 : z0;
    for (int e = (int)threadIdx.x; e < 2 * (INTEGRATE ? cz : kFroxelSegment); e += 256) {
        const int n = INTEGRATE ? cz : kFroxelSegment, plain = e >= n ? 1 : 0, k = e - plain * n;
        sliceDepth[plain][k] = froxelSliceDepth(depthBase + k, target, plain ? 0.f : s.sampleOffset, s.maxDistance);
    }
    if (INTEGRATE) {
        const float invZ = __builtin_amdgcn_rcpf((float)target.d), depthScale = s.maxDistance * (1.f / 19.0855369f); // froxelUVToDepth(uv) = (e^(3 uv) - 1) / (e^3 - 1) * maxDistance
        for (int z = (int)threadIdx.x; z < cz; z += 256)
            sliceLength[z] = (froxelFastExp(3.f * ((float)(z + 1) * invZ)) - 1.f) * depthScale - (froxelFastExp(3.f * ((float)z * invZ)) - 1.f) * depthScale;
    }
    __syncthreads();
    const uint32_t column = blockIdx.x * (uint32_t)(INTEGRATE ? columnsPerBlock : 256) + (uint32_t)columnInBlock;
    const bool valid = column < (uint32_t)cx * (uint32_t)cy && (!INTEGRATE || segment < segments);
    if (!INTEGRATE && !valid) return;
    const int y = valid ? (int)(column / (uint32_t)cx) : 0, x = valid ? (int)(column - (uint32_t)y * (uint32_t)cx) : 0;
    const float kPi = PLR_GLSL_PI;
    const vec3 camPos = ld3(g->cameraPosition);
    vec3 V;
    const vec3 rayJittered = froxelRayPerDepth(x, y, target, s.sampleOffset, g, &V, true);
    const vec3 rayPlain = froxelRayPerDepth(x, y, target, 0.f, g, nullptr, false);
    const float VoL = dot(-V, ld3(g->sunDirection));
    const float gg = s.phaseFunctionG;
    const float phase = (1.f - gg * gg) / (4.f * kPi * det_powf(1.f + gg * gg - 2.f * gg * VoL, 1.5f));
    const int zn = valid ? min(kFroxelSegment, cz - z0) : 0;
    for (int k = 0; k < zn; k++) {
        const int z = z0 + k;
        const size_t texel = idx3(target, x, y, z); // the launcher checked: all three volumes have the target's size
        // froxelVolumeMaterial.comp
        const vec3 posWorld = camPos - rayJittered * sliceDepth[0][z - depthBase];
        vec4 sa;
        {
            const vec3 noiseSample = posWorld * 0.5f + ld3(s.windSampleOffset);
            const float noise = sampleNoiseRepeat(noiseTexture, noiseSample);
            float densityMultiplier = s.baseDensity;
            densityMultiplier += s.densityNoiseRange * (noise - 0.5f);
            densityMultiplier = gmax(densityMultiplier, 0.f);
            const vec4 m(ld3(s.scatteringCoefficients) * densityMultiplier, s.absorptionCoefficient * densityMultiplier);
            if (STORE_INTERMEDIATES) Texel<F_RGBA16F>::store(material.ptr, texel, m);
            sa = vec4(roundToHalf(m.x), roundToHalf(m.y), roundToHalf(m.z), roundToHalf(m.w));
        }
        // froxelLightScattering.comp
        vec4 current;
        {
            vec4 p = mulMat4(shadowInfo->lightMatrices[2], vec4(posWorld, 1.f));
            if (p.w != 1.f) p = p / p.w; // (a quotient by one is its numerator: under the orthographic cascade matrix four IEEE divisions are not run)
            const float actualDepth = gclamp(p.z, 0.f, 1.f);
            const float shadowMapDepth = sampleNearest2D<F_D16, BORDER_BLACK>(shadowMap, vec2(p.x, p.y) * 0.5f + 0.5f).x;
            const float sunStrength = (actualDepth > shadowMapDepth ? 1.f : 0.f) * light->sunStrengthExposed;
            const vec3 scatteringCoefficient = sa.xyz();
            const vec3 inscattering = (sunStrength * phase * ld3(light->sunColor) + vec3(0.02f)) * scatteringCoefficient;
            const float transmittance = computeLuminance(scatteringCoefficient + sa.w);
            const vec4 r(inscattering, transmittance);
            if (STORE_INTERMEDIATES) Texel<F_RGBA16F>::store(scattering.ptr, texel, r);
            current = vec4(roundToHalf(r.x), roundToHalf(r.y), roundToHalf(r.z), roundToHalf(r.w));
        }
        // volumeLightingReprojection.comp
        const vec3 posUnjittered = camPos - rayPlain * sliceDepth[1][z - depthBase];
        vec4 ndcPrevious = mulMat4(g->viewProjectionPrevious, vec4(posUnjittered, 1.f));
        ndcPrevious = vec4(ndcPrevious.x / ndcPrevious.w, ndcPrevious.y / ndcPrevious.w, ndcPrevious.z / ndcPrevious.w, ndcPrevious.w);
        const vec3 camPrev = ld3(g->cameraPositionPrevious);
        // normalize(camPrev - pos) and distance(pos, camPrev) take the root of the same sum of squares (the second difference is the first one's negative)
        const vec3 toCamPrev = camPrev - posUnjittered;
        const float historyDistance = sqrtf(dot(toCamPrev, toCamPrev));
        const vec3 V_history = toCamPrev * (1.0f / historyDistance);
        const float historyDepth = historyDistance * dot(-V_history, ld3(g->cameraForwardPrevious));
        const vec3 historyUV(ndcPrevious.x * 0.5f + 0.5f, ndcPrevious.y * 0.5f + 0.5f, depthToFroxelUVZ(historyDepth, s.maxDistance));
        vec4 history = sampleLinear3D<F_RGBA16F, CLAMP>(historyVolume, historyUV);
        float alpha = 0.95f;
        if (historyUV.x > 1.f || historyUV.y > 1.f || historyUV.z > 1.f || historyUV.x < 0.f || historyUV.y < 0.f || historyUV.z < 0.f) alpha = 0.f;
        if (g->cameraCut) history = current;
        const vec4 result = current * (1.f - alpha) + history * alpha;
        Texel<F_RGBA16F>::store(target.ptr, texel, result);
    }
    if (INTEGRATE) {
        // The running totals of a column are accumulated strictly slice by slice, front to back - the association of the stand-alone integration kernel
        // (kernels_fast/froxel_fast.hip): results must not depend on whether the passes were fused (backend.h; ADVICE r04: summing per-segment subtotals first
        // associates differently from the third segment on and moved half-rounded texels). Every thread forms its segment's per-slice terms at once (the
        // exponentials, in parallel), then the segments take turns in order: segment s starts from the totals segment s - 1 ended with, handed over through LDS.
        uint2 mine[kFroxelSegment];
#pragma unroll
        for (int k = 0; k < kFroxelSegment; k++) // the thread's own stores, read back (same thread, same addresses: program order)
            mine[k] = k < zn ? ((const uint2*)target.ptr)[idx3(target, x, y, z0 + k)] : make_uint2(0u, 0u);
        FroxelSliceTerm term[kFroxelSegment];
#pragma unroll
        for (int k = 0; k < kFroxelSegment; k++) {
            const vec4 texel(halfBitsToFloat(mine[k].x & 0xffffu), halfBitsToFloat(mine[k].x >> 16), halfBitsToFloat(mine[k].y & 0xffffu), halfBitsToFloat(mine[k].y >> 16));
            term[k] = froxelSliceTerm(texel, sliceLength[k < zn ? z0 + k : 0]);
        }
        for (int turn = 0; turn < segments; turn++) { // block-uniform trip count
            if (turn == segment && valid) {
                float totalR = 0.f, totalG = 0.f, totalB = 0.f, transmittance = 1.f;
                if (turn > 0) { totalR = segmentTotals[0][columnInBlock]; totalG = segmentTotals[1][columnInBlock]; totalB = segmentTotals[2][columnInBlock]; transmittance = segmentTotals[3][columnInBlock]; }
#pragma unroll
                for (int k = 0; k < kFroxelSegment; k++) {
                    if (k >= zn) continue;
                    totalR += term[k].r; totalG += term[k].g; totalB += term[k].b; transmittance *= term[k].e;
                    ((uint2*)integrationVolume.ptr)[idx3(integrationVolume, x, y, z0 + k)] =
                        make_uint2(floatToHalfBits(totalR) | (floatToHalfBits(totalG) << 16), floatToHalfBits(totalB) | (floatToHalfBits(transmittance) << 16));
                }
                segmentTotals[0][columnInBlock] = totalR; segmentTotals[1][columnInBlock] = totalG; segmentTotals[2][columnInBlock] = totalB; segmentTotals[3][columnInBlock] = transmittance;
            }
            __syncthreads();
        }
    }
}

PLR_DI vec3 integrateInscattering(vec3 inscattering, vec3 ext, float length) {
    const vec3 e(det_expf(-ext.x * length), det_expf(-ext.y * length), det_expf(-ext.z * length));
    const vec3 num = inscattering - inscattering * e;
    return vec3(num.x / gmax(ext.x, 0.00001f), num.y / gmax(ext.y, 0.00001f), num.z / gmax(ext.z, 0.00001f));
}

__global__ __launch_bounds__(64) void volumetricLightingIntegrationKernel(ImgView integrationVolume, ImgView inVolume, const VolSettings* __restrict__ sp, int coverW, int coverH) {
    const int x = (int)(blockIdx.x * 8u + (threadIdx.x & 7u)), y = (int)(blockIdx.y * 8u + (threadIdx.x >> 3));
    if (x >= coverW || y >= coverH) return;
    const VolSettings s = *sp;
    vec3 inscatteringTotal(0.f);
    float transmittance = 1.f;
    const int resZ = integrationVolume.d;
    for (int z = 0; z <= resZ; z++) { // sic: one slice past the volume (the fetch reads 0, the store is dropped)
        const bool inside = x < inVolume.w && y < inVolume.h && z < inVolume.d;
        const vec4 it = inside ? Texel<F_RGBA16F>::load(inVolume.ptr, idx3(inVolume, x, y, z)) : vec4(0.f);
        const float depthStart = froxelUVToDepth((float)z / (float)resZ, s.maxDistance);
        const float depthEnd = froxelUVToDepth((float)(z + 1) / (float)resZ, s.maxDistance);
        const float segmentLength = depthEnd - depthStart;
        const vec3 inscattering = integrateInscattering(it.xyz(), vec3(it.w), segmentLength);
        inscatteringTotal = inscatteringTotal + inscattering;
        transmittance *= det_expf(-it.w * segmentLength);
        if (z < resZ) Texel<F_RGBA16F>::store(integrationVolume.ptr, idx3(integrationVolume, x, y, z), vec4(inscatteringTotal, transmittance));
    }
}

static void cover3(const PassCtx& c, const ImgView& vol, int wg, int* cx, int* cy, int* cz) {
    *cx = std::min((int)(c.dispatch[0] * (unsigned)wg), vol.w); *cy = std::min((int)(c.dispatch[1] * (unsigned)wg), vol.h); *cz = std::min((int)(c.dispatch[2] * (unsigned)wg), vol.d);
}
static unsigned blocksFor(int cx, int cy, int cz) { return (unsigned)(((long long)cx * cy * cz + 255) / 256); }
static bool tooManyFroxels(int cx, int cy, int cz) { return (long long)cx * cy * cz >= (1ll << 31); }

static int launchMaterial(const PassCtx& c) {
    if (int rc = c.needGlobal()) return rc;
    if (int rc = c.needStorage(0, F_RGBA16F, "froxelVolumeMaterial materialVolume")) return rc;
    if (int rc = c.needSampled(1, F_R8, "froxelVolumeMaterial noiseTexture")) return rc;
    if (int rc = c.needUbuf(2, sizeof(VolSettings), "froxelVolumeMaterial SettingsBuffer")) return rc;
    int cx, cy, cz;
    cover3(c, c.storage[0], 4, &cx, &cy, &cz);
    if (cx <= 0 || cy <= 0 || cz <= 0) return 0;
    if (tooManyFroxels(cx, cy, cz)) return c.fail(-6, "froxelVolumeMaterial: more than 2^31 froxels");
    froxelVolumeMaterialKernel<<<blocksFor(cx, cy, cz), 256, 0, c.stream>>>(c.storage[0], c.sampled[1], (const VolSettings*)c.ubuf[2].ptr, c.global, cx, cy, cz);
    PLR_CHECK_LAUNCH(c);
    return 0;
}
static int launchScattering(const PassCtx& c) {
    if (int rc = c.needGlobal()) return rc;
    if (int rc = c.needStorage(0, F_RGBA16F, "froxelLightScattering scatteringTransmittanceVolume")) return rc;
    if (int rc = c.needSampled(1, F_D16, "froxelLightScattering sunShadowMap")) return rc;
    if (int rc = c.needSampled(2, F_RGBA16F, "froxelLightScattering materialVolume")) return rc;
    if (int rc = c.needSbuf(3, sizeof(ShadowCascadeInfo), "froxelLightScattering sunShadowInfo")) return rc;
    if (int rc = c.needSbuf(4, sizeof(LightBuffer), "froxelLightScattering lightStorageBuffer")) return rc;
    if (int rc = c.needUbuf(5, sizeof(VolSettings), "froxelLightScattering SettingsBuffer")) return rc;
    int cx, cy, cz;
    cover3(c, c.storage[0], 4, &cx, &cy, &cz);
    if (cx <= 0 || cy <= 0 || cz <= 0) return 0;
    if (tooManyFroxels(cx, cy, cz)) return c.fail(-6, "froxelLightScattering: more than 2^31 froxels");
    froxelLightScatteringKernel<<<blocksFor(cx, cy, cz), 256, 0, c.stream>>>(c.storage[0], c.sampled[1], c.sampled[2], (const ShadowCascadeInfo*)c.sbuf[3].ptr,
                                                                            (const LightBuffer*)c.sbuf[4].ptr, (const VolSettings*)c.ubuf[5].ptr, c.global, cx, cy, cz);
    PLR_CHECK_LAUNCH(c);
    return 0;
}
static int launchReprojection(const PassCtx& c) {
    if (int rc = c.needGlobal()) return rc;
    if (int rc = c.needStorage(0, F_RGBA16F, "volumeLightingReprojection targetImage")) return rc;
    if (int rc = c.needSampled(1, F_RGBA16F, "volumeLightingReprojection inputVolume")) return rc;
    if (int rc = c.needSampled(2, F_RGBA16F, "volumeLightingReprojection historyVolume")) return rc;
    if (int rc = c.needUbuf(3, sizeof(VolSettings), "volumeLightingReprojection SettingsBuffer")) return rc;
    int cx, cy, cz;
    cover3(c, c.storage[0], 4, &cx, &cy, &cz);
    if (cx <= 0 || cy <= 0 || cz <= 0) return 0;
    if (tooManyFroxels(cx, cy, cz)) return c.fail(-6, "volumeLightingReprojection: more than 2^31 froxels");
    volumeLightingReprojectionKernel<<<blocksFor(cx, cy, cz), 256, 0, c.stream>>>(c.storage[0], c.sampled[1], c.sampled[2], (const VolSettings*)c.ubuf[3].ptr, c.global, cx, cy, cz);
    PLR_CHECK_LAUNCH(c);
    return 0;
}
static int launchIntegration(const PassCtx& c) {
    if (int rc = c.needStorage(0, F_RGBA16F, "volumetricLightingIntegration integrationVolume")) return rc;
    if (int rc = c.needSampled(1, F_RGBA16F, "volumetricLightingIntegration scatteringTransmittanceVolume")) return rc;
    if (int rc = c.needUbuf(2, sizeof(VolSettings), "volumetricLightingIntegration SettingsBuffer")) return rc;
    const ImgView& out = c.storage[0];
    const int w = std::min((int)(c.dispatch[0] * 8u), out.w), h = std::min((int)(c.dispatch[1] * 8u), out.h);
    if (w <= 0 || h <= 0 || c.dispatch[2] == 0) return 0;
    volumetricLightingIntegrationKernel<<<dim3(divUp((unsigned)w, 8u), divUp((unsigned)h, 8u)), 64, 0, c.stream>>>(out, c.sampled[1], (const VolSettings*)c.ubuf[2].ptr, w, h);
    PLR_CHECK_LAUNCH(c);
    return 0;
}

// material -> scattering -> reprojection as recorded by Volumetrics::computeVolumetricLighting: one launch when the three executions bind one
// chain of equally sized volumes over the same dispatch (anything else: kUseGeneralKernel, the three launches)
static int launchFusedFront(const PassCtx* const* ctxs, size_t count) {
    if (count != 3 && count != 4) return kUseGeneralKernel;
    const PassCtx &m = *ctxs[0], &sc = *ctxs[1], &r = *ctxs[2];
    if (!m.global || !m.hasStorage(0) || m.storage[0].fmt != F_RGBA16F || !m.hasSampled(1) || m.sampled[1].fmt != F_R8 || !m.hasUbuf(2) || m.ubuf[2].size < sizeof(VolSettings)) return kUseGeneralKernel;
    if (!sc.hasStorage(0) || sc.storage[0].fmt != F_RGBA16F || !sc.hasSampled(1) || sc.sampled[1].fmt != F_D16 || !sc.hasSampled(2) || !sc.hasSbuf(3) ||
        sc.sbuf[3].size < sizeof(ShadowCascadeInfo) || !sc.hasSbuf(4) || sc.sbuf[4].size < sizeof(LightBuffer) || !sc.hasUbuf(5))
        return kUseGeneralKernel;
    if (!r.hasStorage(0) || r.storage[0].fmt != F_RGBA16F || !r.hasSampled(1) || !r.hasSampled(2) || r.sampled[2].fmt != F_RGBA16F || !r.hasUbuf(3)) return kUseGeneralKernel;
    const ImgView &mat = m.storage[0], &scat = sc.storage[0], &tgt = r.storage[0], &hist = r.sampled[2];
    auto sameSize = [](const ImgView& a, const ImgView& b) { return a.w == b.w && a.h == b.h && a.d == b.d; };
    if (sc.sampled[2].ptr != mat.ptr || r.sampled[1].ptr != scat.ptr || !sameSize(mat, scat) || !sameSize(mat, tgt) || !sameSize(sc.sampled[2], mat) || !sameSize(r.sampled[1], scat))
        return kUseGeneralKernel;
    if (sc.ubuf[5].ptr != m.ubuf[2].ptr || r.ubuf[3].ptr != m.ubuf[2].ptr) return kUseGeneralKernel; // one settings buffer
    if (tgt.ptr == hist.ptr || tgt.ptr == mat.ptr || tgt.ptr == scat.ptr || mat.ptr == scat.ptr) return kUseGeneralKernel;
    for (int k = 0; k < 3; k++)
        if (m.dispatch[k] != sc.dispatch[k] || m.dispatch[k] != r.dispatch[k] || m.base[k] || sc.base[k] || r.base[k]) return kUseGeneralKernel;
    int cx, cy, cz;
    cover3(m, mat, 4, &cx, &cy, &cz);
    if (cx <= 0 || cy <= 0 || cz <= 0) return 0;
    if (tooManyFroxels(cx, cy, cz)) return kUseGeneralKernel;
    const bool elide = (m.elidableStorage & 1u) && (sc.elidableStorage & 1u);
    ImgView integration{};
    int columnsPerBlock = 256, segments = (int)divUp((unsigned)cz, (unsigned)kFroxelSegment);
    if (count == 4) {
        // + volumetricLightingIntegration.comp over the same columns, all slices: the scan over the column happens inside the block
        const PassCtx& in = *ctxs[3];
        if (!in.hasStorage(0) || in.storage[0].fmt != F_RGBA16F || !in.hasSampled(1) || !in.hasUbuf(2)) return kUseGeneralKernel;
        integration = in.storage[0];
        if (in.sampled[1].ptr != tgt.ptr || !sameSize(in.sampled[1], tgt) || !sameSize(integration, tgt) || in.ubuf[2].ptr != m.ubuf[2].ptr || integration.ptr == tgt.ptr ||
            integration.ptr == hist.ptr || integration.ptr == mat.ptr || integration.ptr == scat.ptr)
            return kUseGeneralKernel;
        const int iw = std::min((int)(in.dispatch[0] * 8u), integration.w), ih = std::min((int)(in.dispatch[1] * 8u), integration.h);
        if (iw != cx || ih != cy || in.dispatch[2] == 0 || in.base[0] || in.base[1] || in.base[2] || cz != tgt.d || segments > kFroxelMaxSegments) return kUseGeneralKernel;
        columnsPerBlock = 256 / segments;
    }
    const dim3 grid = count == 4 ? dim3(divUp((unsigned)(cx * cy), (unsigned)columnsPerBlock)) : dim3(blocksFor(cx, cy, 1), (unsigned)segments);
    auto kernel = count == 4 ? (elide ? froxelFrontFusedKernel<false, true> : froxelFrontFusedKernel<true, true>) : (elide ? froxelFrontFusedKernel<false, false> : froxelFrontFusedKernel<true, false>);
    kernel<<<grid, 256, 0, m.stream>>>(mat, m.sampled[1], scat, sc.sampled[1], (const ShadowCascadeInfo*)sc.sbuf[3].ptr, (const LightBuffer*)sc.sbuf[4].ptr, tgt, hist,
                                       (const VolSettings*)m.ubuf[2].ptr, m.global, cx, cy, cz, integration, columnsPerBlock, segments);
    PLR_CHECK_LAUNCH(m);
    if (elide) { m.elidedStorage = 1u; sc.elidedStorage = 1u; }
    return 0;
}

} // namespace froxel

static int froxel_material_launch(const PassCtx& c) { return froxel::launchMaterial(c); }
static int froxel_scattering_launch(const PassCtx& c) { return froxel::launchScattering(c); }
static int froxel_reprojection_launch(const PassCtx& c) { return froxel::launchReprojection(c); }
static int froxel_integration_launch(const PassCtx& c) { return froxel::launchIntegration(c); }
PLR_REGISTER_SHADER("froxelVolumeMaterial.comp", froxel_material_launch);
PLR_REGISTER_SHADER("froxelLightScattering.comp", froxel_scattering_launch);
PLR_REGISTER_SHADER("volumeLightingReprojection.comp", froxel_reprojection_launch);
PLR_REGISTER_SHADER("volumetricLightingIntegration.comp", froxel_integration_launch);
static int froxel_fused_front(const PassCtx* const* ctxs, size_t count) { return froxel::launchFusedFront(ctxs, count); }
static int froxel_fused_front_and_integration(const PassCtx* const* ctxs, size_t count) { return froxel::launchFusedFront(ctxs, count); }
PLR_REGISTER_FUSION("froxelVolumeMaterial + froxelLightScattering + volumeLightingReprojection + volumetricLightingIntegration", froxel_fused_front_and_integration,
                    "froxelVolumeMaterial.comp", "froxelLightScattering.comp", "volumeLightingReprojection.comp", "volumetricLightingIntegration.comp");
PLR_REGISTER_FUSION("froxelVolumeMaterial + froxelLightScattering + volumeLightingReprojection", froxel_fused_front, "froxelVolumeMaterial.comp", "froxelLightScattering.comp",
                    "volumeLightingReprojection.comp");

} // namespace plr
