// TEST INFRASTRUCTURE ONLY (see oracle.h). CPU restatement of the input-producing compute passes.
// PARITY UNPINNED by the reference (no tests / golden data); lightMatrix is cross-checked against the independent host-side
// cascade fit of plainrenderer_amd/synth.py in tests/test_producers.py.
#include <cmath>
#include <cstring>

#include "common.h"
#include "oracle.h"

using namespace orc;

namespace {
struct M4 { float c[4][4]; }; // column major: c[col][row]
// GLSL matrix product, term order k = 0..3 (the kernel uses the same order)
M4 mul(const M4& a, const M4& b) {
    M4 r;
    for (int col = 0; col < 4; col++)
        for (int row = 0; row < 4; row++) r.c[col][row] = a.c[0][row] * b.c[col][0] + a.c[1][row] * b.c[col][1] + a.c[2][row] * b.c[col][2] + a.c[3][row] * b.c[col][3];
    return r;
}
} // namespace

// resources/shaders/lightMatrix.comp:57-137 (one invocation). apexMinMax = texel (0,0) of the lowest HiZ mip: .x = min depth
// (farthest, reverse Z), .y = max depth.
extern "C" void orc_light_matrix(orc_shadow_cascade_info* info, const float* apexMinMax, const orc_global* g, uint32_t sunShadowCascadeCount,
                                 float highestCascadeExtraPadding, float highestCascadeMinFarPlane) {
    const float FLOAT_MAX = 3.402823466e+38f, FLOAT_MIN = 1.175494351e-38f;
    const float shadowSampleRadius = 0.03f; // sunShadowCascades.inc:5
    M4 corr{};
    corr.c[0][0] = 1.f; corr.c[1][1] = 1.f; corr.c[2][2] = -0.5f; corr.c[3][2] = 0.5f; corr.c[3][3] = 1.f; // :59-63 (initialiser lists are columns)
    const vec3 forward = -vec3(g->sunDirection[0], g->sunDirection[1], g->sunDirection[2]);
    vec3 up = std::fabs(forward.y) < 0.9999f ? vec3(0.f, -1.f, 0.f) : vec3(0.f, 0.f, -1.f);
    const vec3 right = cross(forward, up);
    up = cross(right, forward);
    const vec3 nr = normalize(right), nu = normalize(up);
    // V[0].xyz = right, V[1].xyz = up, V[2].xyz = forward, then transposed: rows of V are the basis vectors
    M4 V{};
    V.c[0][0] = nr.x; V.c[1][0] = nr.y; V.c[2][0] = nr.z;
    V.c[0][1] = nu.x; V.c[1][1] = nu.y; V.c[2][1] = nu.z;
    V.c[0][2] = forward.x; V.c[1][2] = forward.y; V.c[2][2] = forward.z;
    V.c[3][3] = 1.f;

    const float depthMaxLinear = linearizeDepth(apexMinMax[0], g->nearPlane, g->farPlane);
    const float depthMinLinear = linearizeDepth(apexMinMax[1], g->nearPlane, g->farPlane);
    const int count = (int)sunShadowCascadeCount;
    for (int i = 0; i < count - 1; i++) info->splits[i] = depthMinLinear + ((depthMaxLinear - depthMinLinear) * (float)(i + 1) / (float)count);

    const vec3 camPos(g->cameraPosition[0], g->cameraPosition[1], g->cameraPosition[2]), camFwd(g->cameraForward[0], g->cameraForward[1], g->cameraForward[2]);
    const vec3 camUp(g->cameraUp[0], g->cameraUp[1], g->cameraUp[2]), camRight(g->cameraRight[0], g->cameraRight[1], g->cameraRight[2]);
    for (int i = 0; i < count; i++) {
        vec3 minP(FLOAT_MAX), maxP(FLOAT_MIN); // sic: FLOAT_MIN is the smallest positive float (:88-89)
        float cascadeMinDepth = i > 0 ? info->splits[i - 1] : 0.f; // the shader reads splits[-1] for i == 0 and overwrites it below
        float cascadeMaxDepth = i < 4 ? info->splits[i] : 0.f;
        if (i == 0) cascadeMinDepth = depthMinLinear;
        if (i == count - 1) {
            cascadeMinDepth = g->nearPlane;
            cascadeMaxDepth = gmax(depthMaxLinear, highestCascadeMinFarPlane);
        }
        // computeFrustumPoints (:29-49)
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
        const M4 L = mul(mul(corr, P), V);
        std::memcpy(info->lightMatrices[i], L.c, 64);
        info->lightSpaceScale[i][0] = scale.x;
        info->lightSpaceScale[i][1] = scale.y;
    }
}

// ====================================================================================================================
// Sky LUTs: resources/shaders/sky.inc, volumeShading.inc, skyTransmissionLut.comp, skyMultiscatterLut.comp, skyLut.comp.
// exp / pow / sin / cos are the detmath contract functions; max / min / clamp resolve NaN like v_max / v_min (detmath.h).
namespace {

struct Atmosphere { // sky.inc:1-10, std140 (vec3 + float packs into 16 bytes): 56 bytes
    float scatteringRayleighGround[3], earthRadius;
    float extinctionRayleighGround[3], atmosphereHeight;
    float ozoneExtinction[3], scatteringMieGround;
    float extinctionMieGround, mieScatteringExponent;
};

struct Coefficients { vec3 scatterRayleigh, scatterMie, extinction; };

Coefficients calculateCoefficients(float height, const Atmosphere& a) { // sky.inc:30-45
    const float rayleighFactor = det_expf(-height * (1.f / 8));
    const float mieFactor = det_expf(-height * (1.f / 1.2f));
    const float ozoneFactor = gmax(0.f, 1.f - std::fabs(height - 25.f) / 15.f);
    Coefficients c;
    c.scatterRayleigh = rayleighFactor * vec3(a.scatteringRayleighGround[0], a.scatteringRayleighGround[1], a.scatteringRayleighGround[2]);
    c.scatterMie = vec3(mieFactor) * a.scatteringMieGround;
    c.extinction = rayleighFactor * vec3(a.extinctionRayleighGround[0], a.extinctionRayleighGround[1], a.extinctionRayleighGround[2]) + vec3(mieFactor * a.extinctionMieGround) +
                   ozoneFactor * vec3(a.ozoneExtinction[0], a.ozoneExtinction[1], a.ozoneExtinction[2]);
    return c;
}

struct Intersection { vec3 pos; float distance; bool hitEarth; };

Intersection rayEarthIntersection(vec3 P, vec3 D, vec3 C, float earthRadius, float atmosphere) { // sky.inc:63-84
    const vec3 L = C - P;
    const float t_ca = dot(L, D);
    const float d = std::sqrt(dot(L, L) - t_ca * t_ca);
    const float t_hc_earth = std::sqrt(earthRadius * earthRadius - d * d);
    const float t_earth = t_ca - t_hc_earth;
    const float r = earthRadius + atmosphere;
    const float t_hc_atmosphere = std::sqrt(r * r - d * d);
    const float t_atmosphere = t_ca + std::fabs(t_hc_atmosphere);
    Intersection result;
    result.hitEarth = t_earth >= 0.f; // false when the ray misses the earth (sqrt of a negative number is NaN)
    const float t = result.hitEarth ? t_earth : t_atmosphere;
    result.distance = t;
    result.pos = P + t * D;
    return result;
}

vec3 expv(vec3 v) { return vec3(det_expf(v.x), det_expf(v.y), det_expf(v.z)); }

vec3 integrateInscattering(vec3 inscattering, vec3 ext, float length) { // volumeShading.inc:27-29
    const vec3 e = expv(-ext * length);
    const vec3 num = inscattering - inscattering * e;
    return vec3(num.x / gmax(ext.x, 0.00001f), num.y / gmax(ext.y, 0.00001f), num.z / gmax(ext.z, 0.00001f));
}

vec2 computeLutUV(float height, float atmosphereHeight, vec3 up, vec3 direction) { return vec2(height / atmosphereHeight, dot(up, direction) * 0.5f + 0.5f); } // sky.inc:105-110

const float kPi = 3.1415926535f; // global.inc:44

} // namespace

// skyTransmissionLut.comp:17-47
extern "C" void orc_sky_transmission_lut(const orc_image* lutP, const void* atmosphereSettings56) {
    const Image& lut = img(lutP);
    Atmosphere a;
    std::memcpy(&a, atmosphereSettings56, sizeof(a));
    parallelFor(lut.h, [&](int r0, int r1) {
        for (int uy = r0; uy < r1; uy++)
            for (int ux = 0; ux < lut.w; ux++) {
                const float x = (float)ux / (float)(lut.w - 1), y = (float)uy / (float)(lut.h - 1);
                const float height = 0.f * (1.f - x) + a.atmosphereHeight * x; // mix(0, H, x)
                float upDot = y * 2.f - 1.f;
                upDot = gmax(upDot, -0.999f);
                const vec3 V(0.f, -upDot, std::sqrt(1.f - (upDot * upDot)));
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
                imageStore(lut, ivec2(ux, uy), vec4(absorption, 0.f));
            }
    });
}

// skyMultiscatterLut.comp:19-123
extern "C" void orc_sky_multiscatter_lut(const orc_image* lutP, const orc_image* transmissionP, const void* atmosphereSettings56) {
    const Image& lut = img(lutP);
    const Image& transmissionLut = img(transmissionP);
    Atmosphere a;
    std::memcpy(&a, atmosphereSettings56, sizeof(a));
    parallelFor(lut.h, [&](int r0, int r1) {
        for (int uy = r0; uy < r1; uy++)
            for (int ux = 0; ux < lut.w; ux++) {
                const float x = (float)ux / (float)lut.w, y = (float)uy / (float)lut.h;
                const float height = 0.f * (1.f - x) + a.atmosphereHeight * x;
                const vec3 P(0.f, -height - a.earthRadius, 0.f);
                const vec3 earthCenter(0.f);
                const float upDot = y * 2.f - 1.f;
                const vec3 L(0.f, -upDot, std::sqrt(1.f - (upDot * upDot)));
                vec3 L_2nd(0.f), f_ms(0.f);
                const float isotropicPhase = 1.f / (4.f * kPi);
                const int sampleCountSqrt = 8;
                const float sampleCountSqrtRcp = 1.f / (float)sampleCountSqrt;
                for (int i = 0; i < sampleCountSqrt; i++)
                    for (int j = 0; j < sampleCountSqrt; j++) {
                        const float theta = kPi * (float)i * sampleCountSqrtRcp; // phi is computed by the shader but never used (:42)
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
                        const vec3 incomingLight = texture2D(transmissionLut, LINEAR, CLAMP, lutUV0).xyz();
                        const vec3 earthLit = earthAlbedo / kPi * incomingLight * earthNoL;
                        vec3 direct = is.hitEarth ? earthLit : vec3(0.f);
                        vec3 transmission(1.f);
                        const float currentHeight = -currentPosition.y - a.earthRadius; // "approximation" branch (:73-77)
                        for (int k = 0; k < innerSampleCount; k++) {
                            currentPosition = currentPosition + V;
                            const vec3 up(0.f, -1.f, 0.f);
                            const Coefficients c = calculateCoefficients(height, a); // sic: height, not currentHeight (:94)
                            const vec3 scatteringCo = c.scatterRayleigh + c.scatterMie;
                            const vec2 lutUV = computeLutUV(currentHeight, a.atmosphereHeight, up, L);
                            const vec3 transmissionSun = texture2D(transmissionLut, LINEAR, CLAMP, lutUV).xyz();
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
                imageStore(lut, ivec2(ux, uy), vec4(L_2nd * F_ms, 0.f));
            }
    });
}

// skyLut.comp:25-96
extern "C" void orc_sky_lut(const orc_image* lutP, const orc_image* transmissionP, const orc_image* multiscatterP, const void* atmosphereSettings56,
                            const orc_light_buffer* light, const orc_global* g) {
    const Image& lut = img(lutP);
    const Image& transmissionLut = img(transmissionP);
    const Image& multiscatterLut = img(multiscatterP);
    Atmosphere a;
    std::memcpy(&a, atmosphereSettings56, sizeof(a));
    parallelFor(lut.h, [&](int r0, int r1) {
        for (int uy = r0; uy < r1; uy++)
            for (int ux = 0; ux < lut.w; ux++) {
                const float x = (float)ux / (float)lut.w, y = (float)uy / (float)lut.h;
                // fromSkyLut (sky.inc:96-103)
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
                const vec3 L(g->sunDirection[0], g->sunDirection[1], g->sunDirection[2]);
                const float VoL = dot(V, L);
                const float phaseRayleigh = 3.f / (16.f * kPi) * (1.f + VoL * VoL);
                const float gM = a.mieScatteringExponent;
                const float nominator = 3.f / (8.f * kPi) * (1.f - gM * gM) * (1.f + VoL * VoL);
                const float denominator = (2.f + gM * gM) * det_powf(1.f + gM * gM - 2.f * gM * VoL, 1.5f);
                const float phaseMie = nominator / denominator;
                vec3 currentPosition = P;
                vec3 absorption(1.f), color(0.f);
                const vec3 step = V * stepSize;
                for (int i = 0; i < sampleCount; i++) {
                    currentPosition = currentPosition + step;
                    vec3 up = currentPosition - earthCenter;
                    const float upLength = length(up);
                    const float currentHeight = upLength - a.earthRadius;
                    up = up / upLength;
                    const vec2 lutUV = computeLutUV(currentHeight, a.atmosphereHeight, up, L);
                    const vec3 transmission = texture2D(transmissionLut, LINEAR, CLAMP, lutUV).xyz();
                    vec3 incomingLight = light->sunStrengthExposed * transmission;
                    {   // shadowRay (:25-35)
                        const vec3 Lc = earthCenter - currentPosition;
                        const float t_ca = dot(Lc, L);
                        const float d = std::sqrt(dot(Lc, Lc) - t_ca * t_ca);
                        const float t_hc_earth = std::sqrt(a.earthRadius * a.earthRadius - d * d);
                        const float t_earth = t_ca - t_hc_earth;
                        incomingLight = incomingLight * (t_earth > 0.f ? 0.f : 1.f);
                    }
                    const Coefficients c = calculateCoefficients(currentHeight, a);
                    const vec3 inscattering = c.scatterRayleigh * incomingLight * phaseRayleigh + c.scatterMie * incomingLight * phaseMie;
                    const vec3 scatterIntegral = integrateInscattering(inscattering, c.extinction, stepSize);
                    color = color + scatterIntegral * absorption;
                    absorption = absorption * expv(-c.extinction * stepSize);
                    const vec3 multiscattering = texture2D(multiscatterLut, LINEAR, CLAMP, lutUV).xyz();
                    color = color + multiscattering * incomingLight * (c.scatterRayleigh + c.scatterMie) * stepSize * transmission;
                }
                imageStore(lut, ivec2(ux, uy), vec4(color, 0.f));
            }
    });
}

// ====================================================================================================================
// Volumetric froxel lighting: froxelVolumeMaterial.comp, froxelLightScattering.comp, volumeLightingReprojection.comp,
// volumetricLightingIntegration.comp with volumetricFroxelLighting.inc, volume shading helpers. Host: Techniques/Volumetrics.cpp:119-243.
namespace {

struct VolSettings { // volumetricFroxelLighting.inc:6-16 (std140, 52 bytes) == orc_volumetric_settings
    float windSampleOffset[3], sampleOffset;
    float scatteringCoefficients[3], maxDistance;
    float absorptionCoefficient, baseDensity, densityNoiseRange, densityNoiseScale, phaseFunctionG;
};

const float kFroxelK = 3.f; // volumetricFroxelLighting.inc:20
float froxelUVToDepth(float uvZ, float maxDistance) { // :23-31
    const float remaped = (det_expf(kFroxelK * uvZ) - 1.f) / (det_expf(kFroxelK) - 1.f);
    return remaped * maxDistance;
}
float depthToFroxelUVZ(float depth, float maxDistance) { // :33-41
    const float linear = depth / maxDistance;
    return det_logf(linear * (det_expf(kFroxelK) - 1.f) + 1.f) / kFroxelK;
}

// the froxel centre's world position as the three producers compute it (:25-29 of each shader); jitter = settings.sampleOffset or 0
vec3 froxelWorldPosition(int x, int y, int z, const Image& vol, float jitter, const orc_global* g, float maxDistance, vec3* Vout, vec3* uvOut, bool ndcForm2) {
    const vec3 uv(((float)x + 0.5f + jitter) / (float)vol.w, ((float)y + 0.5f + jitter) / (float)vol.h, ((float)z + 0.5f + jitter) / (float)vol.d);
    // froxelVolumeMaterial / reprojection: 2 * (uv - 0.5); froxelLightScattering: 2 * uv - 1
    const vec2 ndc = ndcForm2 ? vec2(2.f * uv.x - 1.f, 2.f * uv.y - 1.f) : vec2(2.f * (uv.x - 0.5f), 2.f * (uv.y - 0.5f));
    const vec3 fwd(g->cameraForward[0], g->cameraForward[1], g->cameraForward[2]);
    const vec3 V = calculateViewDirectionFromPixel(ndc, fwd, vec3(g->cameraUp[0], g->cameraUp[1], g->cameraUp[2]), vec3(g->cameraRight[0], g->cameraRight[1], g->cameraRight[2]),
                                                   g->cameraTanFovHalf, g->cameraAspectRatio);
    const vec3 posWorld = vec3(g->cameraPosition[0], g->cameraPosition[1], g->cameraPosition[2]) - V / dot(-V, fwd) * froxelUVToDepth(uv.z, maxDistance);
    if (Vout) *Vout = V;
    if (uvOut) *uvOut = uv;
    return posWorld;
}

void store3D(const Image& im, int x, int y, int z, vec4 v) {
    if (x < 0 || y < 0 || z < 0 || x >= im.w || y >= im.h || z >= im.d) return; // imageStore out of bounds is dropped
    storeTexel(im, x, y, z, v);
}
vec4 fetch3D(const Image& im, int x, int y, int z) {
    if (x < 0 || y < 0 || z < 0 || x >= im.w || y >= im.h || z >= im.d) return vec4(0.f); // texelFetch out of bounds reads 0
    return loadTexel(im, x, y, z);
}

} // namespace

// froxelVolumeMaterial.comp:17-43
extern "C" void orc_froxel_volume_material(const orc_image* materialP, const orc_image* noiseP, const void* settings52, const orc_global* g) {
    const Image &material = img(materialP), &noiseTexture = img(noiseP);
    VolSettings s;
    std::memcpy(&s, settings52, sizeof(s));
    parallelFor(material.d * material.h, [&](int r0, int r1) {
        for (int r = r0; r < r1; r++) {
            const int z = r / material.h, y = r % material.h;
            for (int x = 0; x < material.w; x++) {
                const vec3 posWorld = froxelWorldPosition(x, y, z, material, s.sampleOffset, g, s.maxDistance, nullptr, nullptr, false);
                const float noiseScale = 0.5f;
                const vec3 noiseSample = posWorld * noiseScale + vec3(s.windSampleOffset[0], s.windSampleOffset[1], s.windSampleOffset[2]);
                const float noise = texture3D(noiseTexture, LINEAR, REPEAT, noiseSample).x;
                vec3 scatteringCoefficient(s.scatteringCoefficients[0], s.scatteringCoefficients[1], s.scatteringCoefficients[2]);
                float absorptionCoefficient = s.absorptionCoefficient;
                float densityMultiplier = s.baseDensity;
                densityMultiplier += s.densityNoiseRange * (noise - 0.5f);
                densityMultiplier = gmax(densityMultiplier, 0.f);
                scatteringCoefficient = scatteringCoefficient * densityMultiplier;
                absorptionCoefficient *= densityMultiplier;
                store3D(material, x, y, z, vec4(scatteringCoefficient, absorptionCoefficient));
            }
        }
    });
}

// froxelLightScattering.comp:31-63 (the sun shadow cascade is hard-wired to index 2, :43)
extern "C" void orc_froxel_light_scattering(const orc_image* outP, const orc_image* shadowMapP, const orc_image* materialP, const orc_shadow_cascade_info* shadowInfo,
                                            const orc_light_buffer* light, const void* settings52, const orc_global* g) {
    const Image &out = img(outP), &shadowMap = img(shadowMapP), &material = img(materialP);
    VolSettings s;
    std::memcpy(&s, settings52, sizeof(s));
    mat4 lightMatrix;
    std::memcpy(lightMatrix.c, shadowInfo->lightMatrices[2], 64);
    parallelFor(out.d * out.h, [&](int r0, int r1) {
        for (int r = r0; r < r1; r++) {
            const int z = r / out.h, y = r % out.h;
            for (int x = 0; x < out.w; x++) {
                vec3 V;
                const vec3 posWorld = froxelWorldPosition(x, y, z, out, s.sampleOffset, g, s.maxDistance, &V, nullptr, true);
                // simpleShadow with the nearest / black-border sampler
                vec4 p = lightMatrix * vec4(posWorld, 1.f);
                p = p / p.w;
                const float actualDepth = gclamp(p.z, 0.f, 1.f);
                const float shadowMapDepth = texture2D(shadowMap, NEAREST, BORDER_BLACK, vec2(p.x, p.y) * 0.5f + 0.5f).x;
                const float shadow = actualDepth > shadowMapDepth ? 1.f : 0.f;
                const float sunStrength = shadow * light->sunStrengthExposed;
                const vec3 L(g->sunDirection[0], g->sunDirection[1], g->sunDirection[2]);
                const float VoL = dot(-V, L);
                const float gg = s.phaseFunctionG;
                const float phase = (1.f - gg * gg) / (4.f * kPi * det_powf(1.f + gg * gg - 2.f * gg * VoL, 1.5f)); // phaseGreenstein
                const vec4 sa = fetch3D(material, x, y, z);
                const vec3 scatteringCoefficient = sa.xyz();
                const float absorptionCoefficient = sa.w;
                const vec3 constantAmbientLighting(0.02f);
                const vec3 inscattering = (sunStrength * phase * vec3(light->sunColor[0], light->sunColor[1], light->sunColor[2]) + constantAmbientLighting) * scatteringCoefficient;
                const vec3 extinctionCoefficient = scatteringCoefficient + absorptionCoefficient;
                const float transmittance = computeLuminance(extinctionCoefficient);
                store3D(out, x, y, z, vec4(inscattering, transmittance));
            }
        }
    });
}

// volumeLightingReprojection.comp:18-61
extern "C" void orc_volume_lighting_reprojection(const orc_image* targetP, const orc_image* inputP, const orc_image* historyP, const void* settings52, const orc_global* g) {
    const Image &target = img(targetP), &inputVolume = img(inputP), &historyVolume = img(historyP);
    VolSettings s;
    std::memcpy(&s, settings52, sizeof(s));
    mat4 vpPrev;
    std::memcpy(vpPrev.c, g->viewProjectionPrevious, 64);
    parallelFor(target.d * target.h, [&](int r0, int r1) {
        for (int r = r0; r < r1; r++) {
            const int z = r / target.h, y = r % target.h;
            for (int x = 0; x < target.w; x++) {
                const vec4 current = fetch3D(inputVolume, x, y, z);
                const vec3 posWorld = froxelWorldPosition(x, y, z, target, 0.f, g, s.maxDistance, nullptr, nullptr, false);
                vec4 ndcPrevious = vpPrev * vec4(posWorld, 1.f);
                ndcPrevious = vec4(ndcPrevious.x / ndcPrevious.w, ndcPrevious.y / ndcPrevious.w, ndcPrevious.z / ndcPrevious.w, ndcPrevious.w);
                const vec3 camPrev(g->cameraPositionPrevious[0], g->cameraPositionPrevious[1], g->cameraPositionPrevious[2]);
                const vec3 V_history = normalize(camPrev - posWorld);
                const float historyDistance = distance(posWorld, camPrev);
                const float historyDepth = historyDistance * dot(-V_history, vec3(g->cameraForwardPrevious[0], g->cameraForwardPrevious[1], g->cameraForwardPrevious[2]));
                const vec3 historyUV(ndcPrevious.x * 0.5f + 0.5f, ndcPrevious.y * 0.5f + 0.5f, depthToFroxelUVZ(historyDepth, s.maxDistance));
                vec4 history = texture3D(historyVolume, LINEAR, CLAMP, historyUV);
                float alpha = 0.95f;
                if (historyUV.x > 1.f || historyUV.y > 1.f || historyUV.z > 1.f || historyUV.x < 0.f || historyUV.y < 0.f || historyUV.z < 0.f) alpha = 0.f;
                if (g->cameraCut) history = current;
                const vec4 result = current * (1.f - alpha) + history * alpha; // mix
                store3D(target, x, y, z, result);
            }
        }
    });
}

// volumetricLightingIntegration.comp:17-42
extern "C" void orc_volumetric_lighting_integration(const orc_image* outP, const orc_image* inP, const void* settings52) {
    const Image &integrationVolume = img(outP), &scatteringTransmittanceVolume = img(inP);
    VolSettings s;
    std::memcpy(&s, settings52, sizeof(s));
    parallelFor(integrationVolume.h, [&](int y0, int y1) {
        for (int y = y0; y < y1; y++)
            for (int x = 0; x < integrationVolume.w; x++) {
                vec3 inscatteringTotal(0.f);
                float transmittance = 1.f;
                const int resZ = integrationVolume.d;
                for (int z = 0; z <= resZ; z++) { // sic: one slice past the volume (the fetch reads 0, the store is dropped)
                    const vec4 it = fetch3D(scatteringTransmittanceVolume, x, y, z);
                    const float depthStart = froxelUVToDepth((float)z / (float)resZ, s.maxDistance);
                    const float depthEnd = froxelUVToDepth((float)(z + 1) / (float)resZ, s.maxDistance);
                    const float segmentLength = depthEnd - depthStart;
                    const vec3 inscattering = integrateInscattering(it.xyz(), vec3(it.w), segmentLength);
                    inscatteringTotal = inscatteringTotal + inscattering;
                    transmittance *= det_expf(-it.w * segmentLength);
                    store3D(integrationVolume, x, y, z, vec4(inscatteringTotal, transmittance));
                }
            }
    });
}
