// ORACLE (test infrastructure, never shipped, never on the product path).
#pragma once
#include "image.h"
#include "oracle.h"
#include <algorithm>
#include <functional>
#include <thread>
#include <vector>

namespace orc {

static inline const Image& img(const orc_image* p) { return *reinterpret_cast<const Image*>(p); }
static_assert(sizeof(orc_image) == sizeof(Image), "orc_image layout");
static_assert(sizeof(orc_global) == 340, "GlobalShaderInfo is 340 bytes (SURVEY a1)");

static inline mat4 toMat4(const float* m) {
    mat4 r;
    for (int c = 0; c < 4; c++) for (int k = 0; k < 4; k++) r.c[c][k] = m[c * 4 + k];
    return r;
}
static inline vec3 v3(const float* p) { return vec3(p[0], p[1], p[2]); }

extern int g_threads;
// decision signatures (orc_set_decision_signature, oracle.h): one word per output pixel of the next pass, or null
extern uint32_t* g_sig;
extern int64_t g_sigWords;
static inline void writeSig(int64_t i, uint32_t word) { if (g_sig && i >= 0 && i < g_sigWords) g_sig[i] = word; }
// rows [0,n) split into contiguous chunks, one std::thread per chunk
void parallelFor(int n, const std::function<void(int, int)>& body);

// global.inc:44
static const float pi = 3.1415926535f;

// resources/shaders/linearDepth.inc:5-8
static inline float linearizeDepth(float depth, float nearP, float farP) {
    return nearP * farP / (farP + (-depth + 1.f) * (nearP - farP));
}

// resources/shaders/screenToWorld.inc:4-9
static inline vec3 calculateViewDirectionFromPixel(vec2 pixelNDC, vec3 cameraForward, vec3 cameraUp, vec3 cameraRight,
                                                   float cameraTanFovHalf, float aspectRatio) {
    vec3 V = -cameraForward;
    V += cameraTanFovHalf * pixelNDC.y * cameraUp;
    V -= cameraTanFovHalf * aspectRatio * pixelNDC.x * cameraRight;
    return normalize(V);
}

// resources/shaders/luminance.inc:5-7
static inline float computeLuminance(vec3 color) { return dot(color, vec3(0.21f, 0.72f, 0.07f)); }

// resources/shaders/colorConversion.inc
static inline vec3 linearTosRGB(vec3 linear) {
    const vec3 lo = linear * 12.92f;
    const vec3 hi = (pow(abs(linear), vec3(1.0f / 2.4f)) * 1.055f) - 0.055f;
    return vec3(linear.x <= 0.0031308f ? lo.x : hi.x, linear.y <= 0.0031308f ? lo.y : hi.y,
                linear.z <= 0.0031308f ? lo.z : hi.z);
}
static inline vec3 sRGBToLinear(vec3 c) {
    const vec3 lo = c / 12.92f;
    const vec3 hi = pow(abs(c + 0.055f) / 1.055f, vec3(2.4f));
    return vec3(c.x <= 0.004045f ? lo.x : hi.x, c.y <= 0.004045f ? lo.y : hi.y, c.z <= 0.004045f ? lo.z : hi.z);
}
static inline vec3 linearToYCoCg(vec3 l) {
    return vec3(l.x * 0.25f + 0.5f * l.y + 0.25f * l.z, l.x * 0.5f - 0.5f * l.z, -l.x * 0.25f + 0.5f * l.y - 0.25f * l.z);
}
static inline vec3 YCoCgToLinear(vec3 c) { return vec3(c.x + c.y - c.z, c.x + c.z, c.x - c.y - c.z); }

// resources/shaders/noise.inc
static inline vec3 hash32(vec2 q) {
    const uint32_t UI0 = 1597334673u, UI1 = 3812015801u, UI2 = 2798796415u;
    // uvec3(ivec3(q.xyx)): float -> int (truncation), reinterpret as uint
    uint32_t nx = (uint32_t)(int32_t)q.x * UI0, ny = (uint32_t)(int32_t)q.y * UI1, nz = (uint32_t)(int32_t)q.x * UI2;
    const uint32_t m = nx ^ ny ^ nz;
    nx = m * UI0; ny = m * UI1; nz = m * UI2;
    const float UIF = 1.0f / (float)0xffffffffu;
    return vec3((float)nx, (float)ny, (float)nz) * UIF;
}
static inline uint32_t xorshift32(uint32_t& state) {
    state ^= (state << 13); state ^= (state >> 17); state ^= (state << 5);
    return state;
}
static inline uint32_t wang_hash(uint32_t seed) {
    seed = (seed ^ 61u) ^ (seed >> 16); seed *= 9u; seed = seed ^ (seed >> 4); seed *= 0x27d4eb2du; seed = seed ^ (seed >> 15);
    return seed;
}
static inline float rand01(uint32_t& state) {
    const uint32_t x = xorshift32(state);
    state = x;
    return gclamp((float)x * u2f(0x2f800004u), 0.f, 1.f);
}

// resources/shaders/SphericalHarmonics.inc
static inline vec4 directionToSH_L1(vec3 V) {
    const float s = std::sqrt(pi);
    const float s3 = std::sqrt(3.f);
    return normalize(vec4(1.f / (2.f * s), -s3 * V.y / (2.f * s), s3 * V.z / (2.f * s), -s3 * V.x / (2.f * s)));
}
static inline vec3 dominantDirectionFromSH_L1(vec4 c) { return vec3(-c.w, -c.y, c.z); }

// resources/shaders/sampling.inc:25-45
static inline vec3 importanceSampleCosine(vec2 xi, vec3 N) {
    const float phi = 2.f * pi * xi.y;
    const float cosTheta = std::sqrt(xi.x);
    const float sinTheta = std::sqrt(1.f - xi.x);
    float sp, cp;
    det_sincosf(phi, &sp, &cp);
    const vec3 sampleHemisphere(cp * sinTheta, sp * sinTheta, cosTheta);
    const vec3 up = std::fabs(N.z) < 0.999f ? vec3(0.f, 0.f, 1.f) : vec3(1.f, 0.f, 0.f);
    const vec3 tangent = normalize(cross(up, N));
    const vec3 bitangent = cross(N, tangent);
    vec3 sampleWorld(0.f);
    sampleWorld += sampleHemisphere.x * tangent;
    sampleWorld += sampleHemisphere.y * bitangent;
    sampleWorld += sampleHemisphere.z * N;
    return sampleWorld;
}
// resources/shaders/sampling.inc:4-23
static inline vec3 importanceSampleGGX(vec2 xi, float r, vec3 N) {
    const float r_2 = r * r;
    const float cosTheta = std::sqrt((1.f - xi.y) / (1.f + (r_2 * r_2 - 1.f) * xi.y));
    const float sinTheta = std::sqrt(1.f - cosTheta * cosTheta);
    const float phi = 2.f * pi * xi.x;
    float sp, cp;
    det_sincosf(phi, &sp, &cp);
    const vec3 sampleHemisphere(cp * sinTheta, sp * sinTheta, cosTheta);
    const vec3 up = std::fabs(N.z) < 0.999f ? vec3(0.f, 0.f, 1.f) : vec3(1.f, 0.f, 0.f);
    const vec3 tangent = normalize(cross(up, N));
    const vec3 bitangent = cross(N, tangent);
    vec3 sampleWorld(0.f);
    sampleWorld += sampleHemisphere.x * tangent;
    sampleWorld += sampleHemisphere.y * bitangent;
    sampleWorld += sampleHemisphere.z * N;
    return sampleWorld;
}

// resources/shaders/sky.inc:85-116
static inline vec2 toSkyLut(vec3 V) {
    const float theta = det_acosf(-(V.y));
    float y = theta / pi;
    const float y_lowRange = y * 2.f - 1.f;
    const float y_lowRangeScaled = gsign(y_lowRange) * std::sqrt(std::fabs(y_lowRange));
    y = y_lowRangeScaled * 0.5f + 0.5f;
    const float phi = -det_atan2f(V.z, V.x);
    return vec2(phi / (2.f * 3.1415f) + 0.5f, y);
}
static inline vec3 sampleSkyLut(vec3 V, const Image& skyLut) {
    vec2 uv = toSkyLut(V);
    uv.y = gclamp(uv.y, 0.005f, 0.995f);
    return texture2D(skyLut, LINEAR, REPEAT, uv).xyz();
}

} // namespace orc
