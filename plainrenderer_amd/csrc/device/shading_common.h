// Device helpers shared by several pass kernels (the .inc files of the reference shader tree).
#pragma once
#include "image.h"
#include "types.h"

namespace plr {

#define PLR_GLSL_PI 3.1415926535f // resources/shaders/global.inc:44

// resources/shaders/linearDepth.inc:5-8
PLR_DI float linearizeDepth(float depth, float nearP, float farP) { return nearP * farP / (farP + (-depth + 1.f) * (nearP - farP)); }

// resources/shaders/screenToWorld.inc:4-9 (returns the direction from the surface towards the camera)
PLR_DI vec3 calculateViewDirectionFromPixel(vec2 pixelNDC, vec3 cameraForward, vec3 cameraUp, vec3 cameraRight, float cameraTanFovHalf, float aspectRatio) {
    vec3 V = -cameraForward;
    V += cameraTanFovHalf * pixelNDC.y * cameraUp;
    V -= cameraTanFovHalf * aspectRatio * pixelNDC.x * cameraRight;
    return normalize(V);
}

// resources/shaders/luminance.inc:5-7
PLR_DI float computeLuminance(vec3 c) { return dot(c, vec3(0.21f, 0.72f, 0.07f)); }

// resources/shaders/colorConversion.inc:26-38
PLR_DI vec3 linearToYCoCg(vec3 l) {
    return vec3(l.x * 0.25f + 0.5f * l.y + 0.25f * l.z, l.x * 0.5f - 0.5f * l.z, -l.x * 0.25f + 0.5f * l.y - 0.25f * l.z);
}
PLR_DI vec3 YCoCgToLinear(vec3 c) { return vec3(c.x + c.y - c.z, c.x + c.z, c.x - c.y - c.z); }
// resources/shaders/colorConversion.inc:15-23
PLR_DI vec3 sRGBToLinear(vec3 c) {
    const vec3 lo = c / 12.92f;
    const vec3 hi = vpow(vabs(c + 0.055f) / 1.055f, 2.4f);
    return vec3(c.x <= 0.004045f ? lo.x : hi.x, c.y <= 0.004045f ? lo.y : hi.y, c.z <= 0.004045f ? lo.z : hi.z);
}

// resources/shaders/noise.inc:28-54
PLR_DI uint32_t xorshift32(uint32_t& state) {
    state ^= (state << 13); state ^= (state >> 17); state ^= (state << 5);
    return state;
}
PLR_DI uint32_t wang_hash(uint32_t seed) {
    seed = (seed ^ 61u) ^ (seed >> 16); seed *= 9u; seed = seed ^ (seed >> 4); seed *= 0x27d4eb2du; seed = seed ^ (seed >> 15);
    return seed;
}
PLR_DI float rand01(uint32_t& state) {
    const uint32_t x = xorshift32(state);
    state = x;
    return gclamp((float)x * u2f(0x2f800004u), 0.f, 1.f);
}

// resources/shaders/SphericalHarmonics.inc:5-15
PLR_DI vec4 directionToSH_L1(vec3 V) {
    const float s = sqrtf(PLR_GLSL_PI);
    const float s3 = sqrtf(3.f);
    return normalize(vec4(1.f / (2.f * s), -s3 * V.y / (2.f * s), s3 * V.z / (2.f * s), -s3 * V.x / (2.f * s)));
}
PLR_DI vec3 dominantDirectionFromSH_L1(vec4 c) { return vec3(-c.w, -c.y, c.z); }

// resources/shaders/sampling.inc:25-45
PLR_DI vec3 importanceSampleCosine(vec2 xi, vec3 N) {
    const float phi = 2.f * PLR_GLSL_PI * xi.y;
    const float cosTheta = sqrtf(xi.x);
    const float sinTheta = sqrtf(1.f - xi.x);
    float sp, cp;
    det_sincosf(phi, &sp, &cp);
    const vec3 h(cp * sinTheta, sp * sinTheta, cosTheta);
    const vec3 up = fabsf(N.z) < 0.999f ? vec3(0.f, 0.f, 1.f) : vec3(1.f, 0.f, 0.f);
    const vec3 tangent = normalize(cross(up, N));
    const vec3 bitangent = cross(N, tangent);
    vec3 s(0.f);
    s += h.x * tangent;
    s += h.y * bitangent;
    s += h.z * N;
    return s;
}
// resources/shaders/sampling.inc:4-23
PLR_DI vec3 importanceSampleGGX(vec2 xi, float r, vec3 N) {
    const float r_2 = r * r;
    const float cosTheta = sqrtf((1.f - xi.y) / (1.f + (r_2 * r_2 - 1.f) * xi.y));
    const float sinTheta = sqrtf(1.f - cosTheta * cosTheta);
    const float phi = 2.f * PLR_GLSL_PI * xi.x;
    float sp, cp;
    det_sincosf(phi, &sp, &cp);
    const vec3 h(cp * sinTheta, sp * sinTheta, cosTheta);
    const vec3 up = fabsf(N.z) < 0.999f ? vec3(0.f, 0.f, 1.f) : vec3(1.f, 0.f, 0.f);
    const vec3 tangent = normalize(cross(up, N));
    const vec3 bitangent = cross(N, tangent);
    vec3 s(0.f);
    s += h.x * tangent;
    s += h.y * bitangent;
    s += h.z * N;
    return s;
}

// resources/shaders/sky.inc:85-116
PLR_DI vec2 toSkyLut(vec3 V) {
    const float theta = det_acosf(-(V.y));
    float y = theta / PLR_GLSL_PI;
    const float y_lowRange = y * 2.f - 1.f;
    const float y_lowRangeScaled = gsign(y_lowRange) * sqrtf(fabsf(y_lowRange));
    y = y_lowRangeScaled * 0.5f + 0.5f;
    const float phi = -det_atan2f(V.z, V.x);
    return vec2(phi / (2.f * 3.1415f) + 0.5f, y);
}
PLR_DI vec3 sampleSkyLut(vec3 V, const ImgView& skyLut) {
    vec2 uv = toSkyLut(V);
    uv.y = gclamp(uv.y, 0.005f, 0.995f);
    return sampleLinear2D<F_R11G11B10, REPEAT>(skyLut, uv).xyz();
}

PLR_DI vec3 ld3(const float* p) { return vec3(p[0], p[1], p[2]); }

} // namespace plr
