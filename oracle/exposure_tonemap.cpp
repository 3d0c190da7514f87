// ORACLE (test infrastructure, never shipped, never on the product path).
// Restates: resources/shaders/histogramPerTile.comp, histogramReset.comp, histogramCombineTiles.comp,
// preExposeLights.comp, tonemapping.comp (+ tonemapping.inc, colorConversion.inc, dither.inc, noise.inc).
#include "common.h"

namespace orc {

int g_threads = 1;
uint32_t* g_sig = nullptr;
int64_t g_sigWords = 0;

void parallelFor(int n, const std::function<void(int, int)>& body) {
    const int t = std::max(1, std::min(g_threads, n));
    if (t == 1) { body(0, n); return; }
    std::vector<std::thread> pool;
    const int chunk = (n + t - 1) / t;
    for (int i = 0; i < t; i++) {
        const int a = i * chunk, b = std::min(n, a + chunk);
        if (a >= b) break;
        pool.emplace_back([=, &body] { body(a, b); });
    }
    for (auto& th : pool) th.join();
}

// histogramPerTile.comp:28-30 (note: not luminance.inc's weights)
static inline float colorToLuminance(vec3 color) { return dot(color, vec3(0.2126f, 0.7152f, 0.0722f)); }

// tonemapping.inc:17-49; the literal triples act as rows because of the transpose
static inline vec3 mulRows(const float m[3][3], vec3 c) {
    return vec3(m[0][0] * c.x + m[0][1] * c.y + m[0][2] * c.z, m[1][0] * c.x + m[1][1] * c.y + m[1][2] * c.z,
                m[2][0] * c.x + m[2][1] * c.y + m[2][2] * c.z);
}
static inline vec3 RRTAndODTFit(vec3 v) {
    const vec3 a = v * (v + 0.0245786f) - 0.000090537f;
    const vec3 b = v * (0.983729f * v + 0.4329510f) + 0.238081f;
    return a / b;
}
static inline vec3 ACESFitted(vec3 color) {
    static const float inM[3][3] = {{0.59719f, 0.35458f, 0.04823f}, {0.07600f, 0.90834f, 0.01566f}, {0.02840f, 0.13383f, 0.83777f}};
    static const float outM[3][3] = {{1.60475f, -0.53108f, -0.07367f}, {-0.10208f, 1.10813f, -0.00605f}, {-0.00327f, -0.07276f, 1.07602f}};
    color = mulRows(inM, color);
    color = RRTAndODTFit(color);
    color = mulRows(outM, color);
    return clamp(color, 0.f, 1.f);
}

// dither.inc:6-12
static inline vec3 ditherRGB8(vec3 c, ivec2 uv, float g_time) {
    // uvec2(uv * g_time) then implicit uvec2 -> vec2
    const vec2 q0((float)(uint32_t)((float)uv.x * g_time), (float)(uint32_t)((float)uv.y * g_time));
    vec3 noise = hash32(q0);
    const vec2 q1((float)(uint32_t)(((float)uv.x + 165.f) * g_time), (float)(uint32_t)(((float)uv.y + 1292.f) * g_time));
    noise += hash32(q1);
    noise -= 1.f;
    noise /= 255.f;
    return c + noise;
}

} // namespace orc

using namespace orc;

extern "C" void orc_set_threads(int32_t n) { g_threads = n < 1 ? 1 : n; }
extern "C" void orc_set_decision_signature(uint32_t* words, int64_t count) { orc::g_sig = words; orc::g_sigWords = words ? count : 0; }

// histogramPerTile.comp:32-65, one iteration of the outer loops per 32x32 workgroup
extern "C" void orc_histogram_per_tile(const orc_image* srcP, const orc_light_buffer* light, uint32_t* perTile,
                                       uint32_t nBins, float minLuminance, float maxLuminance) {
    const Image& src = img(srcP);
    const int tilesX = (int)std::ceil((float)src.w / 32.f);
    const int tilesY = (int)std::ceil((float)src.h / 32.f);
    const float minLuminanceLog = det_logf(minLuminance);
    const float maxLuminanceLog = det_logf(maxLuminance);
    parallelFor(tilesY, [&](int ty0, int ty1) {
        std::vector<uint32_t> local(nBins);
        for (int ty = ty0; ty < ty1; ty++)
            for (int tx = 0; tx < tilesX; tx++) {
                std::fill(local.begin(), local.end(), 0u);
                bool wrote[1024];
                // out-of-image invocations return before touching shared memory (:37-39): a bin is only
                // initialised / written back by an in-image invocation with localIndexFlat < nBins
                for (int ly = 0; ly < 32; ly++)
                    for (int lx = 0; lx < 32; lx++) {
                        const ivec2 uv(tx * 32 + lx, ty * 32 + ly);
                        const bool inside = !(uv.x >= src.w || uv.y >= src.h);
                        wrote[ly * 32 + lx] = inside;
                        if (!inside) continue;
                        const vec3 color = texelFetch(src, uv).xyz();
                        const float luminance = colorToLuminance(color) / light->previousFrameExposure;
                        const float luminanceLog = det_logf(luminance);
                        const uint32_t maxIndex = nBins - 1u;
                        const uint32_t bin = (uint32_t)((float)maxIndex * gclamp((luminanceLog - minLuminanceLog) / (maxLuminanceLog - minLuminanceLog), 0.f, 1.f));
                        local[bin] += 1u;
                    }
                const uint32_t tileIndex = (uint32_t)tx + (uint32_t)ty * (uint32_t)tilesX;
                for (uint32_t b = 0; b < nBins && b < 1024u; b++)
                    if (wrote[b]) perTile[tileIndex * nBins + b] = local[b];
            }
    });
}

// histogramReset.comp:11-16
extern "C" void orc_histogram_reset(uint32_t* histogram, uint32_t nBins) {
    for (uint32_t i = 0; i < nBins; i++) histogram[i] = 0u;
}

// histogramCombineTiles.comp:27-34, grid (tiles, ceil(nBins/64)) x 64 threads
extern "C" void orc_histogram_combine_tiles(const uint32_t* perTile, uint32_t* histogram, uint32_t nBins, uint32_t nTiles) {
    const uint32_t groupsY = (uint32_t)std::ceil((float)nBins / 64.f);
    for (uint32_t tile = 0; tile < nTiles; tile++)
        for (uint32_t gy = 0; gy < groupsY; gy++)
            for (uint32_t lx = 0; lx < 64u; lx++) {
                const uint32_t bin = lx + 64u * gy;
                if (bin > nBins) continue; // the reference's off-by-one guard (:29)
                if (bin >= nBins) continue; // unreachable for nBins % 64 == 0; avoids the OOB access otherwise
                histogram[bin] += perTile[tile * nBins + bin];
            }
}

// preExposeLights.comp:28-38
static float offsetFromSceneEV(float sceneEV100) {
    const float darkExp = 2.84f, lightExp = 12.81f, lightOffset = 1.47f, darkOffset = -3.17f;
    const float t = gclamp((sceneEV100 - darkExp) / (lightExp - darkOffset), 0.f, 1.f);
    return gmix(darkOffset, lightOffset, t);
}

// preExposeLights.comp:40-88 (one invocation)
extern "C" void orc_pre_expose_lights(orc_light_buffer* light, const uint32_t* histogram, const orc_image* transmissionLut,
                                      const orc_global* g, int32_t nBins, float minLuminance, float maxLuminance) {
    const float minLuminanceLog = det_logf(minLuminance);
    const float maxLuminanceLog = det_logf(maxLuminance);
    const uint32_t pixelCount = (uint32_t)(g->screenResolution[0] * g->screenResolution[1]);
    float mean = 0.f;
    uint32_t countedPixels = 0u;
    uint32_t currentPixelCount = 0u;
    for (int i = 0; i < nBins; i++) {
        currentPixelCount += histogram[i];
        const float percentage = (float)currentPixelCount / (float)pixelCount;
        if (percentage < 0.95f && percentage >= 0.5f) {
            const float binValueLog = minLuminanceLog + (maxLuminanceLog - minLuminanceLog) * (float)i / (float)((float)nBins - 1.f);
            const float binValueLinear = det_expf(binValueLog);
            mean += (float)histogram[i] * binValueLinear;
            countedPixels += histogram[i];
        }
    }
    mean /= (float)countedPixels;
    const float sceneEV100 = det_log2f(mean * 100.f / 12.5f);
    float exposureOffset = offsetFromSceneEV(sceneEV100);
    exposureOffset += g->exposureOffset;
    float targetEV100 = sceneEV100 - exposureOffset;
    targetEV100 = gmax(targetEV100, 10.f);
    const float previousEV100 = det_log2f(1.f / (gmax(light->previousFrameExposure, 0.000001f) * 1.2f));
    const float evDelta = targetEV100 - previousEV100;
    const float evMaxChange = g->exposureAdaptionSpeedEvPerSec * g->deltaTime;
    const float evChange = gsign(evDelta) * gmin(std::fabs(evDelta), std::fabs(evMaxChange));
    const float currentEV100 = previousEV100 + evChange;
    const float exposure = 1.f / (det_powf(2.f, currentEV100) * 1.2f);
    light->sunStrengthExposed = g->sunStrength * exposure;
    light->previousFrameExposure = exposure;
    const vec2 lutUV(0.f, -g->sunDirection[1] * 0.5f + 0.5f);
    const vec3 c = texture2D(img(transmissionLut), LINEAR, CLAMP, lutUV).xyz();
    light->sunColor[0] = c.x; light->sunColor[1] = c.y; light->sunColor[2] = c.z;
}

// tonemapping.comp:17-27 (8x8 groups over ceil(res/8); stores outside the image are dropped)
extern "C" void orc_tonemapping(const orc_image* srcP, const orc_image* dstP, const orc_global* g) {
    const Image& src = img(srcP);
    const Image& dst = img(dstP);
    parallelFor(dst.h, [&](int y0, int y1) {
        for (int y = y0; y < y1; y++)
            for (int x = 0; x < dst.w; x++) {
                const ivec2 uv(x, y);
                const vec3 linearColor = texelFetch(src, uv).xyz();
                const vec3 tonemapped = ACESFitted(linearColor);
                vec3 sRGB = linearTosRGB(tonemapped);
                sRGB = ditherRGB8(sRGB, uv, g->time);
                imageStore(dst, uv, vec4(sRGB, 1.f));
            }
    });
}
