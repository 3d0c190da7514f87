// See frame_pipeline.h. Pass order, bindings, specialisation constants and dispatch counts follow the cited reference code.
#include "frame_pipeline.h"

#include "../../../include/plr_image_io.h"

#include <algorithm>
#include <cmath>
#include <stdexcept>

namespace plrhost {

// ------------------------------------------------------------------ small math (the reference uses glm)
static Vec3 operator+(Vec3 a, Vec3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
static Vec3 operator-(Vec3 a, Vec3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
static Vec3 operator*(Vec3 a, float s) { return {a.x * s, a.y * s, a.z * s}; }
static float dot(Vec3 a, Vec3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
static Vec3 cross(Vec3 a, Vec3 b) { return {a.y * b.z - b.y * a.z, a.z * b.x - b.z * a.x, a.x * b.y - b.x * a.y}; }
static Vec3 normalize(Vec3 a) { const float l = std::sqrt(dot(a, a)); return {a.x / l, a.y / l, a.z / l}; }
static Mat4 mul(const Mat4& a, const Mat4& b) {
    Mat4 r{};
    for (int c = 0; c < 4; c++)
        for (int row = 0; row < 4; row++) {
            float s = 0.f;
            for (int k = 0; k < 4; k++) s += a.m[k * 4 + row] * b.m[c * 4 + k];
            r.m[c * 4 + row] = s;
        }
    return r;
}
static Mat4 identity() { Mat4 r{}; r.m[0] = r.m[5] = r.m[10] = r.m[15] = 1.f; return r; }

// Camera.cpp:4-12
static Mat4 viewMatrixFromCameraExtrinsic(const CameraExtrinsic& e) {
    Mat4 rot = identity(); // rows = right, up, -forward
    rot.m[0] = e.right.x; rot.m[4] = e.right.y; rot.m[8] = e.right.z;
    rot.m[1] = e.up.x; rot.m[5] = e.up.y; rot.m[9] = e.up.z;
    rot.m[2] = -e.forward.x; rot.m[6] = -e.forward.y; rot.m[10] = -e.forward.z;
    Mat4 t = identity();
    t.m[12] = -e.position.x; t.m[13] = -e.position.y; t.m[14] = -e.position.z;
    return mul(rot, t);
}
// Camera.cpp:14-27: glm::perspective (RH, -1..1) then the Vulkan Y flip / reverse-Z correction
static Mat4 projectionMatrixFromCameraIntrinsic(const CameraIntrinsic& in) {
    const float f = 1.f / std::tan(in.fov * 3.14159265358979f / 180.f * 0.5f);
    Mat4 p{};
    p.m[0] = f / in.aspectRatio;
    p.m[5] = f;
    p.m[10] = -(in.far + in.near) / (in.far - in.near);
    p.m[11] = -1.f;
    p.m[14] = -(2.f * in.far * in.near) / (in.far - in.near);
    Mat4 corr{};
    corr.m[0] = 1.f; corr.m[5] = -1.f; corr.m[10] = -0.5f; corr.m[14] = 0.5f; corr.m[15] = 1.f;
    return mul(corr, p);
}

// Common/Utilities/MathUtils.cpp:17-23
static uint32_t mipCountFromResolution(uint32_t w, uint32_t h, uint32_t d) { return 1 + (uint32_t)std::floor(std::log2((float)std::max(std::max(w, h), d))); }
static void resolutionFromMip(int w, int h, int mip, int* ow, int* oh) { *ow = std::max(w / (1 << mip), 1); *oh = std::max(h / (1 << mip), 1); }
// MathUtils.cpp:25-60
static float radicalInverseBase2(uint32_t in) {
    uint32_t out = (in << 16) | (in >> 16);
    out = ((out & 0x00ff00ff) << 8) | ((out & 0xff00ff00) >> 8);
    out = ((out & 0x0f0f0f0f) << 4) | ((out & 0xf0f0f0f0) >> 4);
    out = ((out & 0x33333333) << 2) | ((out & 0xcccccccc) >> 2);
    out = ((out & 0x55555555) << 1) | ((out & 0xaaaaaaaa) >> 1);
    return float(out) * (float)2.3283064365386963e-10;
}
static float radicalInverseBase3(uint32_t in) {
    const uint32_t base = 3;
    const float inverseBase = 1.f / (float)base;
    uint32_t reversedDigits = 0, current = in;
    float inverseBasePowerN = 1;
    while (current) {
        const uint32_t next = current / base;
        const uint32_t digit = current - next * base;
        reversedDigits = reversedDigits * base + digit;
        inverseBasePowerN *= inverseBase;
        current = next;
    }
    return reversedDigits * inverseBasePowerN;
}

template <class T> static SpecialisationConstant spec(uint32_t location, const T& v) { return {location, dataToCharArray(&v, sizeof(T))}; }

static ImageDescription desc2D(uint32_t w, uint32_t h, ImageFormat f, ImageUsageFlags usage = ImageUsageFlags::Storage | ImageUsageFlags::Sampled,
                               MipCount mips = MipCount::One, uint32_t manual = 1) {
    ImageDescription d;
    d.width = w; d.height = h; d.depth = 1; d.type = ImageType::Type2D; d.format = f; d.usageFlags = usage; d.mipCount = mips; d.manualMipCount = manual;
    return d;
}
// 8x8 workgroups over a w x h image; rows restricts the dispatch to the workgroup rows touching [rows.begin, rows.end)
// cols (tile rendering) does the same for the workgroup columns
static void dispatch8(ComputePassExecution& exe, uint32_t w, uint32_t h, RowRange rows = {}, ColRange cols = {}) {
    const uint32_t lo = std::min(rows.begin, h), hi = std::min(rows.end, h);
    const uint32_t cl = std::min(cols.begin, w), ch = std::min(cols.end, w);
    exe.dispatchBase[0] = cl / 8;
    exe.dispatchCount[0] = ch > cl ? (ch + 7) / 8 - cl / 8 : 0;
    exe.dispatchBase[1] = lo / 8;
    exe.dispatchCount[1] = hi > lo ? (hi + 7) / 8 - lo / 8 : 0;
    exe.dispatchCount[2] = 1;
}

void recordRows(RenderBackend& be, ComputePassExecution& exe, uint32_t w, uint32_t h, RowRange rows, uint32_t halo, const std::function<void()>& edgesDone, bool rowsFirst, ColRange cols) {
    const uint32_t r0 = std::min(rows.begin, h), r1 = std::min(rows.end, h);
    const uint32_t c0 = std::min(cols.begin, w), c1 = std::min(cols.end, w);
    const bool tiled = c0 > 0 || c1 < w;
    // (a tile's producers are one execution each: the split into an edge and an interior execution of round 3 exists for whole rows only)
    if (!edgesDone || halo == 0 || r1 <= r0 || c1 <= c0 || (tiled && !rowsFirst)) {
        dispatch8(exe, w, h, rows, cols);
        be.setComputePassExecution(exe);
        if (edgesDone) edgesDone();
        return;
    }
    // rows the band above / below needs first; a pass's result does not depend on how its rows are split over dispatches
    // (edges rounded up to 16 rows: the coarsest launch granularity of a pass, two 8-row workgroups of the trace)
    halo = (halo + 15u) & ~15u;
    // split points on 16-row boundaries of the image; no edge on a side without a neighbouring band (first / last rows of the image)
    const uint32_t topEnd = r0 == 0 ? r0 : std::min((r0 + halo + 15u) & ~15u, r1);
    const uint32_t bottomBegin = r1 >= h ? r1 : std::max((r1 > halo ? r1 - halo : 0u) & ~15u, topEnd);
    // the same for the columns of a tile: no edge on a side without a neighbouring tile (first / last columns of the image)
    const uint32_t leftEnd = c0 == 0 ? c0 : std::min((c0 + halo + 15u) & ~15u, c1);
    const uint32_t rightBegin = c1 >= w ? c1 : std::max((c1 > halo ? c1 - halo : 0u) & ~15u, leftEnd);
    if (rowsFirst && (topEnd > r0 || bottomBegin < r1 || leftEnd > c0 || rightBegin < c1)) {
        // one launch: the kernel takes the edge rows (and columns) first and the backend raises its edge signal when they are written (plr.h first_rows / first_cols,
        // workgroup rows / columns of 8)
        ComputePassExecution e = exe;
        dispatch8(e, w, h, RowRange{r0, r1}, cols);
        e.firstRows[0] = topEnd / 8u;
        e.firstRows[1] = bottomBegin / 8u;
        if (tiled) { e.firstCols[0] = leftEnd / 8u; e.firstCols[1] = rightBegin / 8u; }
        be.setComputePassExecution(e);
        edgesDone();
        return;
    }
    if (tiled) { // (edges asked for, none needed: a tile without neighbours)
        dispatch8(exe, w, h, rows, cols);
        be.setComputePassExecution(exe);
        edgesDone();
        return;
    }
    auto part = [&](uint32_t a, uint32_t b) {
        if (b <= a) return;
        ComputePassExecution e = exe;
        dispatch8(e, w, h, RowRange{a, b});
        be.setComputePassExecution(e);
    };
    part(r0, topEnd);
    part(bottomBegin, r1);
    edgesDone();
    part(topEnd, bottomBegin);
}
static RowRange scaleRows(RowRange r, uint32_t divisor) { // the rows of a 1/divisor resolution image that cover r
    if (r.end == 0xffffffffu && r.begin == 0) return r;
    return {r.begin / divisor, (r.end + divisor - 1) / divisor};
}

// ------------------------------------------------------------------ TAA (Techniques/TAA.cpp)
static ShaderDescription temporalFilterShaderDescription(const TAASettings& s) { // TAA.cpp:204-233
    ShaderDescription d;
    d.srcPathRelative = "temporalFilter.comp";
    d.specialisationConstants = {spec(0, s.useClipping), spec(1, s.useMotionVectorDilation), spec(2, s.historySamplingTech), spec(3, s.filterUseTonemapping)};
    return d;
}
void TAA::init(RenderBackend& be, int w, int h, const TAASettings& settings) { // TAA.cpp:7-67
    ComputePassDescription d;
    d.name = "Temporal filtering";
    d.shaderDescription = temporalFilterShaderDescription(settings);
    m_temporalFilterPass = be.createComputePass(d);
    for (int i = 0; i < 2; i++) m_historyBuffers[i] = be.createImage(desc2D(w, h, ImageFormat::R11G11B10_uFloat), nullptr, 0);
    for (int i = 0; i < 2; i++) m_sceneLuminance[i] = be.createImage(desc2D(w, h, ImageFormat::R8), nullptr, 0); // TAA.cpp:38-53
    {
        ComputePassDescription sd; // TAA.cpp:20-36, 235-245
        sd.name = "Temporal supersampling";
        sd.shaderDescription.srcPathRelative = "temporalSupersampling.comp";
        sd.shaderDescription.specialisationConstants = {spec(0, settings.supersampleUseTonemapping)};
        m_temporalSupersamplingPass = be.createComputePass(sd);
        ComputePassDescription ld;
        ld.name = "Color to Luminance";
        ld.shaderDescription.srcPathRelative = "colorToLuminance.comp";
        m_colorToLuminancePass = be.createComputePass(ld);
    }
    UniformBufferDescription ub;
    ub.size = sizeof(float) * 9;
    for (auto& b : m_taaResolveWeightBuffers) b = be.createUniformBuffer(ub);
    m_taaResolveWeightBuffer = m_taaResolveWeightBuffers[0];
}
void TAA::computeTemporalFilter(RenderBackend& be, const FrameIndexCounter& fi, ImageHandle colorSrc, const FrameRenderTargets& currentFrame, ImageHandle target,
                                RowRange rows, uint32_t edgeRows, const std::function<void()>& edgesDone, bool rowsFirst, ColRange cols) const {
    // TAA.cpp:139-166
    const size_t frameIndexMod2 = fi.mod2();
    const ImageHandle historySrc = m_historyBuffers[frameIndexMod2];
    const ImageHandle historyDst = m_historyBuffers[(frameIndexMod2 + 1) % 2];
    const ImageDescription td = be.getImageDescription(target);
    ComputePassExecution exe;
    exe.genericInfo.handle = m_temporalFilterPass;
    exe.genericInfo.resources.storageImages = {ImageResource(target, 0, 1), ImageResource(historyDst, 0, 2)};
    exe.genericInfo.resources.sampledImages = {ImageResource(colorSrc, 0, 0), ImageResource(historySrc, 0, 3), ImageResource(currentFrame.motionBuffer, 0, 4),
                                               ImageResource(currentFrame.depthBuffer, 0, 5)};
    exe.genericInfo.resources.uniformBuffers = {UniformBufferResource(m_taaResolveWeightBuffer, 6)};
    // experiment switch (profiles/r04_cross_frame_overlap.txt): the resolve as part of the frame's asynchronous tail, i.e. beside the NEXT frame's
    // exposure chain / depth pyramid / culling / trace, which read nothing it writes. Whole-frame rendering only.
    static const bool onTail = std::getenv("PLR_TAA_ON_TAIL") && std::atoi(std::getenv("PLR_TAA_ON_TAIL")) != 0;
    exe.asyncTail = onTail && !edgesDone && rows.begin == 0 && rows.end == 0xffffffffu;
    recordRows(be, exe, td.width, td.height, rows, edgeRows, edgesDone, rowsFirst, cols);
}
void TAA::computeTemporalSuperSampling(RenderBackend& be, const FrameIndexCounter& fi, const FrameRenderTargets& currentFrame, const FrameRenderTargets& lastFrame,
                                       ImageHandle target, RowRange rows) const { // TAA.cpp:85-137
    const ImageDescription td = be.getImageDescription(target);
    const size_t frameIndexMod2 = fi.mod2();
    const ImageHandle currentLuminance = m_sceneLuminance[frameIndexMod2];
    const ImageHandle historyLuminance = m_sceneLuminance[(frameIndexMod2 + 1) % 2];
    {
        ComputePassExecution exe;
        exe.genericInfo.handle = m_colorToLuminancePass;
        exe.genericInfo.resources.storageImages = {ImageResource(currentLuminance, 0, 1)};
        exe.genericInfo.resources.sampledImages = {ImageResource(currentFrame.colorBuffer, 0, 0)};
        dispatch8(exe, td.width, td.height, rows);
        be.setComputePassExecution(exe);
    }
    {
        ComputePassExecution exe;
        exe.genericInfo.handle = m_temporalSupersamplingPass;
        exe.genericInfo.resources.storageImages = {ImageResource(target, 0, 3)};
        exe.genericInfo.resources.sampledImages = {ImageResource(currentFrame.colorBuffer, 0, 1), ImageResource(lastFrame.colorBuffer, 0, 2), ImageResource(currentFrame.motionBuffer, 0, 4),
                                                   ImageResource(currentFrame.depthBuffer, 0, 5), ImageResource(lastFrame.depthBuffer, 0, 6), ImageResource(currentLuminance, 0, 7),
                                                   ImageResource(historyLuminance, 0, 8)};
        dispatch8(exe, td.width, td.height, rows);
        be.setComputePassExecution(exe);
    }
}
void TAA::jitterInPixels(const FrameIndexCounter& fi, float out[2]) const { // TAA.cpp:168-170
    const uint32_t i = (uint32_t)fi.mod8();
    out[0] = 2.f * radicalInverseBase2(i) - 1.f;
    out[1] = 2.f * radicalInverseBase3(i) - 1.f;
}
void TAA::updateTaaResolveWeights(RenderBackend& be, const float j[2]) { // TAA.cpp:181-202
    std::array<float, 9> weights = {};
    int index = 0;
    float totalWeight = 0.f;
    for (int y = -1; y <= 1; y++)
        for (int x = -1; x <= 1; x++) {
            const float dx = j[0] - (float)x, dy = j[1] - (float)y;
            const float d = std::sqrt(dx * dx + dy * dy);
            const float w = std::exp(-2.29f * d * d);
            weights[index++] = w;
            totalWeight += w;
        }
    for (float& w : weights) w /= totalWeight;
    be.setUniformBufferData(m_taaResolveWeightBuffer, &weights[0], sizeof(float) * 9);
}

// ------------------------------------------------------------------ Bloom (Techniques/Bloom.cpp)
static const int bloomMipCount = 6; // Bloom.cpp:6
void Bloom::init(RenderBackend& be) { // Bloom.cpp:8-40
    for (int i = 0; i < bloomMipCount - 1; i++) {
        ComputePassDescription d;
        d.name = "Bloom downsample mip " + std::to_string(i + 1);
        d.shaderDescription.srcPathRelative = "bloomDownsample.comp";
        m_bloomDownsamplePasses.push_back(be.createComputePass(d));
    }
    for (int i = 0; i < bloomMipCount - 1; i++) {
        ComputePassDescription d;
        d.name = "Bloom Upsample mip " + std::to_string(bloomMipCount - 2 - i);
        d.shaderDescription.srcPathRelative = "bloomUpsample.comp";
        const bool isLowestMip = i == 0;
        d.shaderDescription.specialisationConstants = {spec(0, isLowestMip)};
        m_bloomUpsamplePasses.push_back(be.createComputePass(d));
    }
    ComputePassDescription d;
    d.name = "Apply bloom";
    d.shaderDescription.srcPathRelative = "applyBloom.comp";
    m_applyBloomPass = be.createComputePass(d);
}
// Band rendering: which rows of every level does the bloom of `applyRows` depend on? (the dependency cone of the chain, in each level's own rows)
//   up[t] row y reads up[t+1] at y/2 + 0.25 +- 0.5 texel and down[t+1] at y/2 + 0.25 +- radius, each through a bilinear footprint
//   (bloomUpsample.comp:29-55); down[m+1] row y reads down[m] rows 2y-1 .. 2y+3 (bloomDownsample.comp:21-50; the outermost rows enter with weight 0,
//   but must hold finite values).
// Recording every level over "band +- 320 rows" (the sum of all footprints) cost a middle band 59 % extra bloom work; with the cone the full-resolution
// target has no halo at all and only the cheap coarse levels carry the wide ones.
Bloom::Cone Bloom::dependencyCone(RowRange applyRows, uint32_t height, float radius) {
    Cone c;
    if (applyRows.begin == 0 && applyRows.end == 0xffffffffu) return c; // whole image: every level whole
    auto levelRows = [&](int level) { return std::max(height >> level, 1u); };
    auto grow = [&](RowRange r, uint32_t k, int level) { return RowRange{r.begin > k ? r.begin - k : 0u, std::min(r.end + k, levelRows(level))}; };
    auto half = [](RowRange r) { return RowRange{r.begin / 2, (r.end + 1) / 2}; };
    auto twice = [](RowRange r) { return RowRange{r.begin * 2, r.end * 2}; };
    auto unite = [](RowRange a, RowRange b) { return RowRange{std::min(a.begin, b.begin), std::max(a.end, b.end)}; };
    const uint32_t tent = (uint32_t)std::ceil(radius + 0.25f) + 1u;
    c.up[0] = RowRange{std::min(applyRows.begin, height), std::min(applyRows.end, height)};
    for (int t = 0; t + 1 < bloomMipCount - 1; t++) c.up[t + 1] = grow(half(c.up[t]), 2, t + 1);
    c.down[bloomMipCount - 1] = grow(half(c.up[bloomMipCount - 2]), tent, bloomMipCount - 1);
    for (int m = bloomMipCount - 2; m >= 1; m--) c.down[m] = unite(grow(half(c.up[m - 1]), tent, m), grow(twice(c.down[m + 1]), 2, m));
    c.source = grow(twice(c.down[1]), 2, 0);
    return c;
}
uint32_t Bloom::requiredSourceHalo(uint32_t height, float radius) {
    // widest reach over bands in the middle of the image (64-row aligned begin / end do not change the cone's width)
    const uint32_t b0 = (height / 2) & ~63u, b1 = std::min(b0 + 64u, height);
    const Cone c = dependencyCone(RowRange{b0, b1}, height, radius);
    return std::max(b0 - c.source.begin, c.source.end - b1);
}

void Bloom::computeBloom(RenderBackend& be, ImageHandle targetImage, const BloomSettings& settings, RowRange chainRows, RowRange applyRows, bool asyncTail, ColRange chainCols,
                         ColRange applyCols) const { // Bloom.cpp:56-143
    const ImageDescription td = be.getImageDescription(targetImage);
    const int width = (int)td.width, height = (int)td.height;
    // chainRows = rows of the target that hold valid colour (band + exchanged halo); the chain itself only covers the dependency cone of applyRows
    const Cone cone = dependencyCone(applyRows, td.height, settings.radius);
    const bool banded = !(applyRows.begin == 0 && applyRows.end == 0xffffffffu);
    // tile rendering: the cone of the tile's columns (the footprints are the same in x and y: dependencyCone of a column range and the width)
    const Cone coneX = dependencyCone(applyCols, td.width, settings.radius);
    const bool tiled = !(applyCols.begin == 0 && applyCols.end == 0xffffffffu);
    auto levelCols = [&](ColRange coneCols, int level) {
        const ColRange valid = scaleRows(chainCols, 1u << level);
        return ColRange{std::max(coneCols.begin, valid.begin), std::min(coneCols.end, valid.end)};
    };
    // a halo smaller than the cone (BandSettings::postHalo < requiredSourceHalo): levels stop at the rows derived from valid colour, rows
    // beyond keep what an earlier frame left there (the documented limit of a too-small halo; the default halo covers the cone)
    auto levelRows = [&](RowRange coneRows, int level) {
        const RowRange valid = scaleRows(chainRows, 1u << level);
        return RowRange{std::max(coneRows.begin, valid.begin), std::min(coneRows.end, valid.end)};
    };
    const ImageDescription desc = desc2D(width, height, ImageFormat::R11G11B10_uFloat, ImageUsageFlags::Sampled | ImageUsageFlags::Storage, MipCount::Manual, bloomMipCount);
    const ImageHandle downscaleTexture = be.createTemporaryImage(desc);
    for (int i = 0; i < (int)m_bloomDownsamplePasses.size(); i++) {
        ComputePassExecution exe;
        exe.genericInfo.handle = m_bloomDownsamplePasses[i];
        const int sourceMip = i, targetMip = i + 1;
        exe.genericInfo.resources.storageImages = {ImageResource(downscaleTexture, targetMip, 0)};
        exe.genericInfo.resources.sampledImages = {ImageResource(i == 0 ? targetImage : downscaleTexture, sourceMip, 1)};
        int tw, th;
        resolutionFromMip(width, height, targetMip, &tw, &th);
        dispatch8(exe, tw, th, banded ? levelRows(cone.down[targetMip], targetMip) : RowRange{}, tiled ? levelCols(coneX.down[targetMip], targetMip) : ColRange{});
        exe.asyncTail = asyncTail;
        be.setComputePassExecution(exe);
    }
    const ImageHandle upscaleTexture = be.createTemporaryImage(desc);
    for (int i = 0; i < (int)m_bloomUpsamplePasses.size(); i++) {
        ComputePassExecution exe;
        exe.genericInfo.handle = m_bloomUpsamplePasses[i];
        const int targetMip = bloomMipCount - 2 - i, sourceMip = targetMip + 1;
        exe.genericInfo.resources.storageImages = {ImageResource(upscaleTexture, targetMip, 0)};
        exe.genericInfo.resources.sampledImages = {ImageResource(upscaleTexture, sourceMip, 1), ImageResource(downscaleTexture, sourceMip, 2)};
        int tw, th;
        resolutionFromMip(width, height, targetMip, &tw, &th);
        dispatch8(exe, tw, th, banded ? levelRows(cone.up[targetMip], targetMip) : RowRange{}, tiled ? levelCols(coneX.up[targetMip], targetMip) : ColRange{});
        exe.pushConstants = dataToCharArray(&settings.radius, sizeof(settings.radius));
        exe.asyncTail = asyncTail;
        be.setComputePassExecution(exe);
    }
    ComputePassExecution exe;
    exe.genericInfo.handle = m_applyBloomPass;
    exe.genericInfo.resources.storageImages = {ImageResource(targetImage, 0, 0)};
    exe.genericInfo.resources.sampledImages = {ImageResource(upscaleTexture, 0, 1)};
    dispatch8(exe, width, height, applyRows, applyCols);
    exe.pushConstants = dataToCharArray(&settings.strength, sizeof(settings.strength));
    exe.asyncTail = asyncTail;
    be.setComputePassExecution(exe);
}

// ------------------------------------------------------------------ SDFGI (Techniques/SDFGI.cpp)
static const size_t sdfCameraCullingTileSize = 32; // SDFGI.cpp:10
static const size_t maxSdfObjectsPerTile = 100;    // SDFGI.cpp:11
static const size_t sdfInstanceSize = 96;          // sizeof(SDFInstance), SDFGI.h:31-37

void SDFGI::init(RenderBackend& be, int screenW, int screenH, const SDFTraceSettings& ts, const SDFDebugSettings& ds, int sunShadowCascadeIndex, uint32_t maxInstances) { // SDFGI.cpp:48-258
    const uint32_t tw = ts.halfResTrace ? screenW / 2 : screenW, th = ts.halfResTrace ? screenH / 2 : screenH;
    for (int i = 0; i < 2; i++) {
        m_indirectDiffuse_Y_SH[i] = be.createImage(desc2D(tw, th, ImageFormat::RGBA16_sFloat), nullptr, 0);
        m_indirectDiffuse_CoCg[i] = be.createImage(desc2D(tw, th, ImageFormat::RG16_sFloat), nullptr, 0);
        m_indirectDiffuseHistory_Y_SH[i] = be.createImage(desc2D(tw, th, ImageFormat::RGBA16_sFloat), nullptr, 0);
        m_indirectDiffuseHistory_CoCg[i] = be.createImage(desc2D(tw, th, ImageFormat::RG16_sFloat), nullptr, 0);
    }
    m_indirectLightingFullRes_Y_SH = be.createImage(desc2D(screenW, screenH, ImageFormat::RGBA16_sFloat), nullptr, 0);
    m_indirectLightingFullRes_CoCg = be.createImage(desc2D(screenW, screenH, ImageFormat::RG16_sFloat), nullptr, 0);
    StorageBufferDescription sb;
    sb.size = maxInstances * sdfInstanceSize + sizeof(uint32_t) * 4;
    m_sdfInstanceBuffer = be.createStorageBuffer(sb);
    sb.size = maxInstances * sizeof(uint32_t) + sizeof(uint32_t);
    m_sdfCameraFrustumCulledInstances = be.createStorageBuffer(sb);
    UniformBufferDescription ub;
    ub.size = 12 * 4 * sizeof(float);
    m_cameraFrustumBuffer = be.createUniformBuffer(ub);
    sb.size = maxInstances * 2 * 4 * sizeof(float);
    m_sdfInstanceWorldBBBuffer = be.createStorageBuffer(sb);
    {
        // the reference sizes this for 1920x1080 only (SDFGI.cpp:145-151); sized here for the tile-index range the shaders
        // produce: stride = ceil(screenW / 32) (full-res, sdfCulling.inc:17-20) times the trace image's tile rows
        const size_t strideX = (size_t)std::ceil(screenW / float(sdfCameraCullingTileSize));
        // (the debug visualisation culls at full resolution, SDFGI.cpp:340-350: rows for the full screen height)
        const size_t rows = (size_t)std::ceil(std::max<uint32_t>(th, (uint32_t)screenH) / float(sdfCameraCullingTileSize));
        const size_t tileSize = maxSdfObjectsPerTile * sizeof(uint32_t) + sizeof(uint32_t);
        sb.size = strideX * std::max<size_t>(rows, 1) * tileSize;
        m_sdfCameraCulledTiles = be.createStorageBuffer(sb);
    }
    ub.size = sizeof(float);
    m_sdfTraceInfluenceRangeBuffer = be.createUniformBuffer(ub);
    {
        ComputePassDescription d; // createSDFDebugShaderDescription, SDFGI.cpp:13-28
        d.name = "Visualize SDF";
        d.shaderDescription.srcPathRelative = "sdfDebugVisualisation.comp";
        d.shaderDescription.specialisationConstants = {spec(0, ds.visualisationMode), spec(1, sunShadowCascadeIndex)};
        m_sdfDebugVisualisationPass = be.createComputePass(d);
    }
    {
        ComputePassDescription d;
        d.name = "Indirect diffuse SDF trace";
        d.shaderDescription.srcPathRelative = "sdfDiffuseTrace.comp";
        d.shaderDescription.specialisationConstants = {spec(0, ts.strictInfluenceRadiusCutoff), spec(1, sunShadowCascadeIndex)};
        m_diffuseSDFTracePass = be.createComputePass(d);
    }
    for (int i = 0; i < 2; i++) {
        // the request-list exchange of a partitioned frame (GiBand::requested; no reference counterpart): which texels outside this GPU's rectangle do the disc samples
        // of spatial filter pass i land on?
        ComputePassDescription d;
        d.name = i == 0 ? "Indirect diffuse sample requests (filter 0)" : "Indirect diffuse sample requests (filter 1)";
        d.shaderDescription.srcPathRelative = "giSampleRequests.comp";
        d.shaderDescription.specialisationConstants = {spec(0, i)};
        m_giSampleRequestPass[i] = be.createComputePass(d);
        m_giRequestRowWords = (tw + 31u) / 32u;
        StorageBufferDescription rb;
        rb.size = (size_t)m_giRequestRowWords * th * sizeof(uint32_t);
        m_giRequestBitmap[i] = be.createStorageBuffer(rb);
    }
    for (int i = 0; i < 2; i++) {
        ComputePassDescription d;
        d.name = "Indirect diffuse spatial filter";
        d.shaderDescription.srcPathRelative = "filterIndirectDiffuseSpatial.comp";
        d.shaderDescription.specialisationConstants = {spec(0, i)};
        m_indirectDiffuseFilterSpatialPass[i] = be.createComputePass(d);
    }
    {
        ComputePassDescription d;
        d.name = "Indirect diffuse temporal filter";
        d.shaderDescription.srcPathRelative = "filterIndirectDiffuseTemporal.comp";
        m_indirectDiffuseFilterTemporalPass = be.createComputePass(d);
        d.name = "Indirect lighting upscale";
        d.shaderDescription.srcPathRelative = "indirectLightUpscale.comp";
        m_indirectLightingUpscale = be.createComputePass(d);
        d.name = "SDF camera frustum culling";
        d.shaderDescription.srcPathRelative = "sdfCameraFrustumCulling.comp";
        m_sdfCameraFrustumCulling = be.createComputePass(d);
    }
    for (int hiz = 0; hiz < 2; hiz++) {
        ComputePassDescription d;
        d.name = "SDF camera tile culling";
        d.shaderDescription.srcPathRelative = "sdfCameraTileCulling.comp";
        const bool useHiZ = hiz == 1;
        d.shaderDescription.specialisationConstants = {spec(0, useHiZ)};
        (useHiZ ? m_sdfCameraTileCullingHiZ : m_sdfCameraTileCulling) = be.createComputePass(d);
    }
}

void SDFGI::updateSDFScene(RenderBackend& be, const void* instanceData, size_t instanceBytes, const void* bbData, size_t bbBytes) { // SDFGI.cpp:260-313
    if (instanceBytes >= 4) std::memcpy(&m_sdfInstanceCount, instanceData, 4);
    be.setStorageBufferData(m_sdfInstanceBuffer, instanceData, instanceBytes);
    be.setStorageBufferData(m_sdfInstanceWorldBBBuffer, bbData, bbBytes);
}

SDFGI::IndirectLightingImages SDFGI::getIndirectLightingResults(bool tracedHalfRes) const { // SDFGI.cpp:315-326
    if (tracedHalfRes) return {m_indirectLightingFullRes_Y_SH, m_indirectLightingFullRes_CoCg};
    return {m_indirectDiffuseHistory_Y_SH[0], m_indirectDiffuseHistory_CoCg[0]};
}

void SDFGI::sampleRequests(RenderBackend& be, const SDFTraceDependencies& deps, const SDFTraceSettings& s, const GiBand* band) const {
    // recorded like the spatial filter executions they precede: the same rectangle; what is "valid" is the rectangle itself, every sample beyond it is a request
    const ImageDescription td = be.getImageDescription(m_indirectDiffuse_Y_SH[1]);
    for (int i = 0; i < 2; i++) {
        ComputePassExecution exe;
        exe.genericInfo.handle = m_giSampleRequestPass[i];
        exe.genericInfo.resources.sampledImages = {ImageResource(s.halfResTrace ? deps.depthHalfRes : deps.currentFrame.depthBuffer, 0, 4)};
        exe.genericInfo.resources.storageBuffers = {StorageBufferResource(m_giRequestBitmap[i], false, 6)};
        dispatch8(exe, td.width, td.height, band->traceRows, band->traceCols);
        exe.validRows[0] = std::min(band->traceRows.begin, td.height); exe.validRows[1] = std::min(band->traceRows.end, td.height);
        const uint32_t c0 = std::min(band->traceCols.begin, td.width), c1 = std::min(band->traceCols.end, td.width);
        if (c0 > 0 || c1 < td.width) { exe.validCols[0] = c0; exe.validCols[1] = c1; }
        be.setComputePassExecution(exe);
    }
}

void SDFGI::computeIndirectLighting(RenderBackend& be, const FrameIndexCounter&, const SDFTraceDependencies& deps, const SDFTraceSettings& s, const GiBand* band) const {
    if (band && band->requested) {
        // where the two spatial filter passes' samples land depends on depth, camera and frame index only: asked for now, traded while the trace runs
        sampleRequests(be, deps, s, band);
        band->exchangePoint(band->user, ExchangeGiRequests);
    }
    diffuseSDFTrace(be, deps, s, band);
    filterIndirectDiffuse(be, deps, s, band);
}

void SDFGI::renderSDFVisualization(RenderBackend& be, ImageHandle target, const SDFTraceDependencies& deps, const SDFDebugSettings& ds, const SDFTraceSettings& ts) const {
    // SDFGI.cpp:334-369
    const float sdfInfluenceRadius = ds.useInfluenceRadiusForDebug ? ts.traceInfluenceRadius : 0.f;
    const ImageDescription td = be.getImageDescription(m_indirectLightingFullRes_CoCg);
    const bool useHiZCulling = ds.visualisationMode == SDFVisualisationMode::CameraTileUsage && ds.showCameraTileUsageWithHiZ;
    sdfInstanceCulling(be, deps, (int)td.width, (int)td.height, sdfInfluenceRadius, useHiZCulling, nullptr);
    ComputePassExecution exe;
    exe.genericInfo.handle = m_sdfDebugVisualisationPass;
    exe.genericInfo.resources.storageImages = {ImageResource(target, 0, 0)};
    exe.genericInfo.resources.sampledImages = {ImageResource(deps.skyLut, 0, 2), ImageResource(deps.shadowMap, 0, 7)};
    exe.genericInfo.resources.storageBuffers = {StorageBufferResource(deps.lightBuffer, true, 1), StorageBufferResource(m_sdfInstanceBuffer, true, 3),
                                                StorageBufferResource(m_sdfCameraCulledTiles, true, 4), StorageBufferResource(m_sdfCameraFrustumCulledInstances, true, 5),
                                                StorageBufferResource(deps.sunShadowInfoBuffer, true, 6)};
    dispatch8(exe, td.width, td.height);
    be.setComputePassExecution(exe);
}

void SDFGI::sdfInstanceCulling(RenderBackend& be, const SDFTraceDependencies& deps, int targetW, int targetH, float influenceRadius, bool hiZCulling, const GiBand* band) const {
    // SDFGI.cpp:538-630
    {
        struct GPUFrustumData { float points[6][4]; float normal[6][4]; } frustumData;
        std::memcpy(frustumData.points, deps.frustumPoints, sizeof(frustumData.points));
        std::memcpy(frustumData.normal, deps.frustumNormals, sizeof(frustumData.normal));
        be.setUniformBufferData(m_cameraFrustumBuffer, &frustumData, sizeof(frustumData));
        uint32_t zero = 0;
        be.setStorageBufferData(m_sdfCameraFrustumCulledInstances, &zero, sizeof(zero));
        ComputePassExecution exe;
        exe.genericInfo.handle = m_sdfCameraFrustumCulling;
        exe.genericInfo.resources.storageBuffers = {StorageBufferResource(m_sdfInstanceBuffer, true, 0), StorageBufferResource(m_sdfCameraFrustumCulledInstances, false, 2),
                                                    StorageBufferResource(m_sdfInstanceWorldBBBuffer, true, 3)};
        exe.genericInfo.resources.uniformBuffers = {UniformBufferResource(m_cameraFrustumBuffer, 1), UniformBufferResource(m_sdfTraceInfluenceRangeBuffer, 4)};
        exe.dispatchCount[0] = uint32_t(std::ceil(m_sdfInstanceCount / 64.f));
        exe.dispatchCount[1] = 1;
        exe.dispatchCount[2] = 1;
        be.setComputePassExecution(exe);
    }
    {
        ComputePassExecution exe;
        exe.genericInfo.handle = hiZCulling ? m_sdfCameraTileCullingHiZ : m_sdfCameraTileCulling;
        const uint32_t tileCount[2] = {(uint32_t)std::ceil(targetW / float(sdfCameraCullingTileSize)), (uint32_t)std::ceil(targetH / float(sdfCameraCullingTileSize))};
        const uint32_t localGroupSize = 8;
        exe.dispatchCount[0] = uint32_t(std::ceil(tileCount[0] / float(localGroupSize)));
        exe.dispatchCount[1] = uint32_t(std::ceil(tileCount[1] / float(localGroupSize)));
        exe.dispatchCount[2] = 1;
        if (band) { // only the workgroups (8 tile rows each) that hold tiles of the band's trace rows
            const uint32_t tileRows = (uint32_t)sdfCameraCullingTileSize * localGroupSize;
            const uint32_t lo = std::min(band->traceRows.begin, (uint32_t)targetH), hi = std::min(band->traceRows.end, (uint32_t)targetH);
            exe.dispatchBase[1] = lo / tileRows;
            exe.dispatchCount[1] = hi > lo ? (hi + tileRows - 1) / tileRows - lo / tileRows : 0;
            // tile rendering: and the workgroups (8 tile columns each) that hold tiles of the tile's trace columns
            const uint32_t cl = std::min(band->traceCols.begin, (uint32_t)targetW), ch = std::min(band->traceCols.end, (uint32_t)targetW);
            exe.dispatchBase[0] = cl / tileRows;
            exe.dispatchCount[0] = ch > cl ? (ch + tileRows - 1) / tileRows - cl / tileRows : 0;
        }
        exe.pushConstants = dataToCharArray(&tileCount, sizeof(tileCount));
        exe.genericInfo.resources.storageBuffers = {StorageBufferResource(m_sdfCameraFrustumCulledInstances, true, 0), StorageBufferResource(m_sdfInstanceWorldBBBuffer, true, 1),
                                                    StorageBufferResource(m_sdfCameraCulledTiles, false, 2)};
        be.setUniformBufferData(m_sdfTraceInfluenceRangeBuffer, &influenceRadius, sizeof(influenceRadius));
        exe.genericInfo.resources.uniformBuffers = {UniformBufferResource(m_sdfTraceInfluenceRangeBuffer, 3)};
        const int depthPyramidMipLevel = (int)std::log2(std::ceil(float(sdfCameraCullingTileSize))) - 1; // = 4
        exe.genericInfo.resources.sampledImages = {ImageResource(deps.depthMinMaxPyramid, depthPyramidMipLevel, 4)};
        be.setComputePassExecution(exe);
    }
}

void SDFGI::diffuseSDFTrace(RenderBackend& be, const SDFTraceDependencies& deps, const SDFTraceSettings& s, const GiBand* band) const { // SDFGI.cpp:380-419
    const ImageDescription td = be.getImageDescription(m_indirectDiffuse_CoCg[0]);
    sdfInstanceCulling(be, deps, td.width, td.height, s.traceInfluenceRadius, true, band);
    ComputePassExecution exe;
    exe.genericInfo.handle = m_diffuseSDFTracePass;
    exe.genericInfo.resources.storageImages = {ImageResource(m_indirectDiffuse_Y_SH[0], 0, 0), ImageResource(m_indirectDiffuse_CoCg[0], 0, 1)};
    exe.genericInfo.resources.sampledImages = {ImageResource(deps.currentFrame.depthBuffer, 0, 2), ImageResource(deps.worldSpaceNormals, 0, 3), ImageResource(deps.skyLut, 0, 4),
                                               ImageResource(deps.shadowMap, 0, 10)};
    exe.genericInfo.resources.storageBuffers = {StorageBufferResource(deps.lightBuffer, true, 5), StorageBufferResource(m_sdfInstanceBuffer, true, 6),
                                                StorageBufferResource(m_sdfCameraCulledTiles, true, 7), StorageBufferResource(deps.sunShadowInfoBuffer, true, 9)};
    exe.genericInfo.resources.uniformBuffers = {UniformBufferResource(m_sdfTraceInfluenceRangeBuffer, 8)};
    if (band && band->requested && band->requestedBegin) {
        recordRows(be, exe, td.width, td.height, band->traceRows, 0, nullptr, false, band->traceCols);
        band->requestedBegin(band->user, ExchangeGiTrace); // the owners gather what was asked of them as soon as their trace is done
        return;
    }
    if (band && band->exchangeBegin)
        recordRows(be, exe, td.width, td.height, band->traceRows, band->giHalo, [&] { band->exchangeBegin(band->user, ExchangeGiTrace); }, band->rowsFirst, band->traceCols);
    else recordRows(be, exe, td.width, td.height, band ? band->traceRows : RowRange{}, 0, nullptr, false, band ? band->traceCols : ColRange{});
    if (band && band->exchangePoint) band->exchangePoint(band->user, ExchangeGiTrace); // spatial pass 0 reads neighbouring bands' rays
}

void SDFGI::filterIndirectDiffuse(RenderBackend& be, const SDFTraceDependencies& deps, const SDFTraceSettings& s, const GiBand* band) const { // SDFGI.cpp:421-536
    const ImageHandle depthSrc = s.halfResTrace ? deps.depthHalfRes : deps.currentFrame.depthBuffer;
    const ImageDescription td = be.getImageDescription(m_indirectDiffuse_Y_SH[1]);
    const RowRange rows = band ? band->traceRows : RowRange{};
    const ColRange cols = band ? band->traceCols : ColRange{};
    // band rendering: the spatial filter's inputs are valid on the band's rows and the giHalo rows received from each neighbour; a disc sample
    // beyond them is treated like an off-screen sample (ComputePassExecution::validRows, plr.h) instead of reading rows nobody sent
    // a spatial filter execution; with request lists + overlap: the waves that need no requested texel, the wait for the exchange, the others (push constant: the phase)
    auto recordSpatial = [&](ComputePassExecution& exe, int exchangeId) {
        if (band && band->requested && band->requestedEnd) {
            const int32_t phase1 = 1, phase2 = 2;
            exe.pushConstants = dataToCharArray(&phase1, sizeof(phase1));
            be.setComputePassExecution(exe);
            band->requestedEnd(band->user, exchangeId);
            exe.pushConstants = dataToCharArray(&phase2, sizeof(phase2));
            be.setComputePassExecution(exe);
            return;
        }
        be.setComputePassExecution(exe);
    };
    auto setValidRows = [&](ComputePassExecution& exe, int filterIndex) {
        if (!band) return;
        if (band->requested) { // every texel a sample can land on is either this rectangle's or has been requested and received: nothing is masked (plr.h: {0, 0} = all valid)
            exe.genericInfo.resources.storageBuffers.push_back(StorageBufferResource(m_giRequestBitmap[filterIndex], true, 6));
            return;
        }
        const uint32_t r0 = std::min(rows.begin, td.height), r1 = std::min(rows.end, td.height);
        exe.validRows[0] = r0 > band->giHalo ? r0 - band->giHalo : 0u;
        exe.validRows[1] = std::min(r1 + band->giHalo, td.height);
        // tile rendering: the same for the columns (plr.h valid_cols)
        const uint32_t c0 = std::min(cols.begin, td.width), c1 = std::min(cols.end, td.width);
        if (c0 > 0 || c1 < td.width) {
            exe.validCols[0] = c0 > band->giHalo ? c0 - band->giHalo : 0u;
            exe.validCols[1] = std::min(c1 + band->giHalo, td.width);
        }
    };
    {
        ComputePassExecution exe;
        exe.genericInfo.handle = m_indirectDiffuseFilterSpatialPass[0];
        exe.genericInfo.resources.storageImages = {ImageResource(m_indirectDiffuse_Y_SH[1], 0, 0), ImageResource(m_indirectDiffuse_CoCg[1], 0, 1)};
        exe.genericInfo.resources.sampledImages = {ImageResource(m_indirectDiffuse_Y_SH[0], 0, 2), ImageResource(m_indirectDiffuse_CoCg[0], 0, 3), ImageResource(depthSrc, 0, 4),
                                                   ImageResource(deps.worldSpaceNormals, 0, 5)};
        dispatch8(exe, td.width, td.height, rows, cols);
        setValidRows(exe, 0);
        recordSpatial(exe, ExchangeGiTrace);
    }
    {
        // always History[0] -> History[1]; the reference computes historySrc/DstIndex but never uses them (SDFGI.cpp:457-458)
        ComputePassExecution exe;
        exe.genericInfo.handle = m_indirectDiffuseFilterTemporalPass;
        exe.genericInfo.resources.storageImages = {ImageResource(m_indirectDiffuse_Y_SH[0], 0, 0), ImageResource(m_indirectDiffuse_CoCg[0], 0, 1),
                                                   ImageResource(m_indirectDiffuseHistory_Y_SH[1], 0, 2), ImageResource(m_indirectDiffuseHistory_CoCg[1], 0, 3)};
        exe.genericInfo.resources.sampledImages = {ImageResource(m_indirectDiffuse_Y_SH[1], 0, 4), ImageResource(m_indirectDiffuse_CoCg[1], 0, 5),
                                                   ImageResource(m_indirectDiffuseHistory_Y_SH[0], 0, 6), ImageResource(m_indirectDiffuseHistory_CoCg[0], 0, 7),
                                                   ImageResource(deps.currentFrame.motionBuffer, 0, 8), ImageResource(deps.previousFrame.motionBuffer, 0, 9)};
        if (band && band->requested && band->requestedBegin) { recordRows(be, exe, td.width, td.height, rows, 0, nullptr, false, cols); band->requestedBegin(band->user, ExchangeGiTemporal); }
        else if (band && band->exchangeBegin) recordRows(be, exe, td.width, td.height, rows, band->giHalo, [&] { band->exchangeBegin(band->user, ExchangeGiTemporal); }, band->rowsFirst, cols);
        else recordRows(be, exe, td.width, td.height, rows, 0, nullptr, false, cols);
    }
    if (band && band->exchangePoint && !(band->requested && band->requestedBegin)) band->exchangePoint(band->user, ExchangeGiTemporal); // spatial pass 1 reads neighbouring rows of History[1]
    {
        ComputePassExecution exe;
        exe.genericInfo.handle = m_indirectDiffuseFilterSpatialPass[1];
        exe.genericInfo.resources.storageImages = {ImageResource(m_indirectDiffuseHistory_Y_SH[0], 0, 0), ImageResource(m_indirectDiffuseHistory_CoCg[0], 0, 1)};
        exe.genericInfo.resources.sampledImages = {ImageResource(m_indirectDiffuseHistory_Y_SH[1], 0, 2), ImageResource(m_indirectDiffuseHistory_CoCg[1], 0, 3),
                                                   ImageResource(depthSrc, 0, 4), ImageResource(deps.worldSpaceNormals, 0, 5)};
        setValidRows(exe, 1);
        dispatch8(exe, td.width, td.height, rows, cols);
        recordSpatial(exe, ExchangeGiTemporal);
    }
    // (this exchange is small - 16 rows - and its producer launches a packing pre-pass per dispatch: it is not split / overlapped)
    if (band && band->exchangeWhole) band->exchangeWhole(band->user, ExchangeGiHistory);
    else if (band && band->exchangePoint) band->exchangePoint(band->user, ExchangeGiHistory); // upscale + next frame's reprojection read History[0]
    if (s.halfResTrace) {
        ComputePassExecution exe;
        exe.genericInfo.handle = m_indirectLightingUpscale;
        exe.genericInfo.resources.storageImages = {ImageResource(m_indirectLightingFullRes_Y_SH, 0, 0), ImageResource(m_indirectLightingFullRes_CoCg, 0, 1)};
        exe.genericInfo.resources.sampledImages = {ImageResource(m_indirectDiffuseHistory_Y_SH[0], 0, 2), ImageResource(m_indirectDiffuseHistory_CoCg[0], 0, 3),
                                                   ImageResource(deps.currentFrame.depthBuffer, 0, 4), ImageResource(deps.depthHalfRes, 0, 5)};
        const ImageDescription fd = be.getImageDescription(m_indirectLightingFullRes_Y_SH);
        dispatch8(exe, fd.width, fd.height, band ? band->upscaleRows : RowRange{}, band ? band->upscaleCols : ColRange{});
        be.setComputePassExecution(exe);
    }
}

// ------------------------------------------------------------------ FramePipeline (RenderFrontend.cpp)
static const uint32_t nHistogramBins = 128;       // RenderFrontend.cpp:46
static const uint32_t histogramTileSizeX = 32, histogramTileSizeY = 32;
static const float histogramMinValue = 0.001f, histogramMaxValue = 200000.f; // RenderFrontend.cpp:1066-1067
static const uint32_t noiseTextureWidth = 32, noiseTextureHeight = 32;       // RenderFrontend.cpp:52-54
static const int maxSunShadowCascadeCount = 4;

// RenderFrontend::computeSinglePassMipChainDispatchCount, RenderFrontend.cpp:1807-1827
static void singlePassMipChainDispatchCount(uint32_t width, uint32_t height, uint32_t mipCount, uint32_t maxMipCount, uint32_t out[2]) {
    const uint32_t unusedMips = maxMipCount - mipCount;
    if (unusedMips >= 6) { out[0] = out[1] = 1; return; }
    const uint32_t localThreadGroupExtent = 32 / (uint32_t)std::pow((uint32_t)2, unusedMips);
    out[0] = (uint32_t)std::ceil(float(width) / localThreadGroupExtent);
    out[1] = (uint32_t)std::ceil(float(height) / localThreadGroupExtent);
}

static const uint32_t bandAlignment = 64;      // HiZ tile (64 px), culling tile (32 trace px), histogram tile (32 px), 2^5 bloom texel
static const uint32_t bandPyramidMipCount = 6; // per-tile levels of the HiZ kernel; the hot path reads mip 4 (SDFGI.cpp:567)
static const uint32_t maxPyramidMipCount = 11; // depthHiZPyramid.comp binds at most 11 levels (pyramid base <= 2048)
// A band builds only the levels one 64x64-pixel tile yields: the rest of the chain needs every band and has no consumer on this path.
// A frame whose full chain would exceed the shader's 11 levels (8K un-tiled; the reference cannot render it) gets the same per-tile pyramid.
static bool perTilePyramid(const FramePipelineSettings& s) { return s.band.enabled() || mipCountFromResolution(s.width / 2, s.height / 2, 1) > maxPyramidMipCount; }
static uint32_t pyramidMipCount(const FramePipelineSettings& s) { return perTilePyramid(s) ? bandPyramidMipCount : mipCountFromResolution(s.width / 2, s.height / 2, 1); }

FramePipeline::FramePipeline(const FramePipelineSettings& s) : settings(s) {
    const uint32_t W = s.width, H = s.height;
    if (s.band.enabled()) {
        const BandSettings& b = s.band;
        if (b.rowEnd > H || b.rowBegin % bandAlignment != 0 || (b.rowEnd % bandAlignment != 0 && b.rowEnd != H))
            throw std::runtime_error("band rows must lie inside the frame and start/end on multiples of 64 (or at the last row)");
        if (b.tiled() && (b.colEnd > W || b.colBegin % bandAlignment != 0 || (b.colEnd % bandAlignment != 0 && b.colEnd != W)))
            throw std::runtime_error("tile columns must lie inside the frame and start/end on multiples of 64 (or at the last column)");
    } else if (s.band.tiled()) throw std::runtime_error("tile columns without band rows: set rowBegin / rowEnd too");
    // a halo never needs to be larger than the image (PLRF_HALO_WHOLE_IMAGE: the exact mode of a partitioned frame, every GI texel a denoiser sample can
    // reach is exchanged); clamping here keeps the row arithmetic below in 32 bits
    if (settings.band.giHalo == 0xfffffffeu) { // PLRF_HALO_REQUESTED
        if (!s.sdfTrace.halfResTrace) throw std::runtime_error("band_gi_halo = PLRF_HALO_REQUESTED needs the half-resolution trace (the packed-texel spatial filter)");
        settings.band.giRequested = true;
        settings.band.giHalo = 0;
    }
    settings.band.giHalo = std::min(settings.band.giHalo, std::max(W, H));
    settings.band.giHistoryHalo = std::min(settings.band.giHistoryHalo, std::max(W, H));
    for (int i = 0; i < ExchangeCount; i++) {
        m_exchangeCtx[i] = {this, i};
        m_exchangeCtx[ExchangeCount + i] = {this, i | ExchangeBegin};
        m_exchangeCtx[2 * ExchangeCount + i] = {this, i | ExchangeEnd};
    }
    // the backend is set up by the caller (plr_setup), like gRenderBackend.setup in the reference's main(); match the swapchain
    const ImageDescription sw = m_be.getImageDescription(m_be.getSwapchainInputImage());
    if (sw.width != W || sw.height != H) m_be.recreateSwapchain(W, H);
    m_cameraIntrinsic.aspectRatio = (float)W / (float)H;

    // ---- initBuffers (RenderFrontend.cpp:1449-1500)
    StorageBufferDescription sb;
    sb.size = nHistogramBins * sizeof(uint32_t);
    m_histogramBuffer = m_be.createStorageBuffer(sb);
    float lightInit[5] = {0, 0, 0, 0, 0};
    sb.size = sizeof(lightInit); sb.initialData = lightInit;
    m_lightBuffer = m_be.createStorageBuffer(sb);
    sb.initialData = nullptr;
    const uint32_t tileCount = (uint32_t)std::ceil(W / float(histogramTileSizeX)) * (uint32_t)std::ceil(H / float(histogramTileSizeY));
    sb.size = (size_t)tileCount * nHistogramBins * sizeof(uint32_t); // the reference allocates for 1920x1080 only (:1069-1070)
    m_histogramPerTileBuffer = m_be.createStorageBuffer(sb);
    sb.size = sizeof(uint32_t);
    m_depthPyramidSyncBuffer = m_be.createStorageBuffer(sb);
    sb.size = 16 + 64 * maxSunShadowCascadeCount + 8 * maxSunShadowCascadeCount;
    m_sunShadowInfoBuffer = m_be.createStorageBuffer(sb);
    UniformBufferDescription ub;
    ub.size = sizeof(GlobalShaderInfo);
    m_globalUniformBuffer = m_be.createUniformBuffer(ub);
    ub.size = 64;
    m_volumetricsInfoBuffer = m_be.createUniformBuffer(ub);
    RenderPassResources globalResources;
    globalResources.uniformBuffers = {UniformBufferResource(m_globalUniformBuffer, 0)};
    m_be.setGlobalDescriptorSetResources(globalResources);

    // ---- initImages (RenderFrontend.cpp:1186-1447): G-buffer + colour targets, two sets
    for (int i = 0; i < 2; i++) {
        m_postProcessBuffers[i] = m_be.createImage(desc2D(W, H, ImageFormat::R11G11B10_uFloat), nullptr, 0);
        m_frameRenderTargets[i].colorBuffer = m_be.createImage(desc2D(W, H, ImageFormat::R11G11B10_uFloat), nullptr, 0);
        m_frameRenderTargets[i].motionBuffer = m_be.createImage(desc2D(W, H, ImageFormat::RG16_sNorm), nullptr, 0);
        m_frameRenderTargets[i].depthBuffer = m_be.createImage(desc2D(W, H, ImageFormat::Depth32), nullptr, 0);
    }
    m_worldSpaceNormalImage = m_be.createImage(desc2D(W, H, ImageFormat::RGBA8), nullptr, 0);
    m_albedoImage = m_be.createImage(desc2D(W, H, ImageFormat::RGBA8), nullptr, 0);
    m_specularImage = m_be.createImage(desc2D(W, H, ImageFormat::RGBA8), nullptr, 0);
    m_minMaxDepthPyramid = perTilePyramid(s)
        ? m_be.createImage(desc2D(W / 2, H / 2, ImageFormat::RG32_sFloat, ImageUsageFlags::Storage | ImageUsageFlags::Sampled, MipCount::Manual, bandPyramidMipCount), nullptr, 0)
        : m_be.createImage(desc2D(W / 2, H / 2, ImageFormat::RG32_sFloat, ImageUsageFlags::Storage | ImageUsageFlags::Sampled, MipCount::FullChain), nullptr, 0);
    // band rendering: the depth range of the whole frame (lightMatrix.comp's "lowest mip"), reduced from the band's per-tile pyramids and all-reduced
    m_bandDepthApex = m_be.createImage(desc2D(1, 1, ImageFormat::RG32_sFloat), nullptr, 0);
    m_depthHalfRes = m_be.createImage(desc2D(W / 2, H / 2, ImageFormat::R16_sFloat), nullptr, 0);
    m_brdfLut = m_be.createImage(desc2D(s.brdfLutRes, s.brdfLutRes, ImageFormat::RGBA16_sFloat), nullptr, 0);
    m_skyLut = m_be.createImage(desc2D(200, 100, ImageFormat::R11G11B10_uFloat), nullptr, 0);          // Sky.cpp sky LUT
    m_transmissionLut = m_be.createImage(desc2D(128, 128, ImageFormat::R11G11B10_uFloat), nullptr, 0); // Sky.cpp transmission LUT
    {
        ImageDescription d; // Volumetrics.cpp:8-16
        d.width = (uint32_t)std::ceil(W / 8.f); d.height = (uint32_t)std::ceil(H / 8.f); d.depth = s.froxelDepth;
        d.type = ImageType::Type3D; d.format = ImageFormat::RGBA16_sFloat; d.usageFlags = ImageUsageFlags::Storage | ImageUsageFlags::Sampled;
        m_volumetricIntegrationVolume = m_be.createImage(d, nullptr, 0);
    }
    for (int i = 0; i < maxSunShadowCascadeCount; i++)
        m_shadowMaps[i] = m_be.createImage(desc2D(s.shadowMapRes, s.shadowMapRes, ImageFormat::Depth16, ImageUsageFlags::Sampled), nullptr, 0);
    for (int i = 0; i < 4; i++) {
        m_noiseTextures[i] = m_be.createImage(desc2D(noiseTextureWidth, noiseTextureHeight, ImageFormat::RG8, ImageUsageFlags::Sampled), nullptr, 0);
        m_globalShaderInfo.noiseTextureIndices[i] = (int32_t)m_be.getImageGlobalTextureArrayIndex(m_noiseTextures[i]);
    }

    // ---- initRenderpasses (RenderFrontend.cpp:1618-1760)
    {
        ComputePassDescription d;
        const int nTiles = (int)tileCount;
        d.name = "Histogram per tile";
        d.shaderDescription.srcPathRelative = "histogramPerTile.comp";
        d.shaderDescription.specialisationConstants = {spec(0, nHistogramBins), spec(1, histogramMinValue), spec(2, histogramMaxValue), spec(3, nTiles)};
        m_histogramPerTilePass = m_be.createComputePass(d);
        d.name = "Histogram reset";
        d.shaderDescription.srcPathRelative = "histogramReset.comp";
        d.shaderDescription.specialisationConstants = {spec(0, nHistogramBins)};
        m_histogramResetPass = m_be.createComputePass(d);
        d.name = "Histogram combine tiles";
        d.shaderDescription.srcPathRelative = "histogramCombineTiles.comp";
        d.shaderDescription.specialisationConstants = {spec(0, nHistogramBins), spec(1, nTiles)};
        m_histogramCombinePass = m_be.createComputePass(d);
        d.name = "Pre-expose lights";
        d.shaderDescription.srcPathRelative = "preExposeLights.comp";
        d.shaderDescription.specialisationConstants = {spec(0, (int)nHistogramBins), spec(1, histogramMinValue), spec(2, histogramMaxValue)};
        m_preExposeLightsPass = m_be.createComputePass(d);
        d.name = "Tonemap";
        d.shaderDescription.srcPathRelative = "tonemapping.comp";
        d.shaderDescription.specialisationConstants = {};
        m_tonemappingPass = m_be.createComputePass(d);
        d.name = "Depth downscale";
        d.shaderDescription.srcPathRelative = "depthDownscale.comp";
        m_depthDownscalePass = m_be.createComputePass(d);
        d.name = "BRDF Lut creation";
        d.shaderDescription.srcPathRelative = "brdfLut.comp";
        d.shaderDescription.specialisationConstants = {spec(0, s.shading.diffuseBRDF)};
        m_brdfLutPass = m_be.createComputePass(d);
    }
    {
        // createDepthPyramidShaderDescription, RenderFrontend.cpp:1770-1805
        ComputePassDescription d;
        d.name = "Depth min/max pyramid";
        d.shaderDescription.srcPathRelative = "depthHiZPyramid.comp";
        const uint32_t pw = W / 2, ph = H / 2;
        const uint32_t depthMipCount = pyramidMipCount(s);
        uint32_t dc[2];
        singlePassMipChainDispatchCount(pw, ph, depthMipCount, 11, dc);
        m_depthPyramidThreadgroupCount = dc[0] * dc[1];
        d.shaderDescription.specialisationConstants = {spec(0, depthMipCount), spec(1, W), spec(2, H), spec(3, m_depthPyramidThreadgroupCount)};
        m_depthPyramidPass = m_be.createComputePass(d);
    }
    {
        // createForwardPassShaderDescription, RenderFrontend.cpp:1093-1131, as the deferred compute pass
        ComputePassDescription d;
        d.name = "Forward shading (deferred)";
        d.shaderDescription.srcPathRelative = "deferredShading.comp";
        d.shaderDescription.specialisationConstants = {spec(0, s.shading.diffuseBRDF), spec(1, s.shading.directMultiscatter), spec(2, s.shading.useGeometryAA),
                                                       spec(3, s.shading.indirectLightingTech), spec(4, s.shading.sunShadowCascadeCount)};
        m_deferredShadingPass = m_be.createComputePass(d);
    }
    {
        ComputePassDescription d; // RenderFrontend.cpp:1730-1740
        d.name = "Compute light matrix";
        d.shaderDescription.srcPathRelative = "lightMatrix.comp";
        d.shaderDescription.specialisationConstants = {spec(0, (uint32_t)s.shading.sunShadowCascadeCount)};
        m_lightMatrixPass = m_be.createComputePass(d);
        d.name = "Depth pyramid apex of the band";
        d.shaderDescription.srcPathRelative = "depthPyramidApex.comp"; // no reference shader: see kernels/hiz.hip
        d.shaderDescription.specialisationConstants = {};
        m_depthApexPass = m_be.createComputePass(d);
    }
    {
        // Sky::init (Techniques/Sky.cpp:5-64, 196-258): LUT sizes 128^2, 32^2, 200x100; the transmission and sky LUT images are the ones the
        // shade / trace / exposure passes already read
        m_skyMultiscatterLut = m_be.createImage(desc2D(32, 32, ImageFormat::R11G11B10_uFloat), nullptr, 0);
        UniformBufferDescription aub;
        aub.size = sizeof(AtmosphereSettings);
        m_atmosphereSettingsBuffer = m_be.createUniformBuffer(aub);
        ComputePassDescription d;
        d.name = "Sky transmission lut";
        d.shaderDescription.srcPathRelative = "skyTransmissionLut.comp";
        m_skyTransmissionLutPass = m_be.createComputePass(d);
        d.name = "Sky multiscatter lut";
        d.shaderDescription.srcPathRelative = "skyMultiscatterLut.comp";
        m_skyMultiscatterLutPass = m_be.createComputePass(d);
        d.name = "Sky lut";
        d.shaderDescription.srcPathRelative = "skyLut.comp";
        m_skyLutPass = m_be.createComputePass(d);
    }
    {
        // Volumetrics::init (Techniques/Volumetrics.cpp:19-117): froxel volumes of the integration volume's size; the 32^3 R8 Perlin noise is an
        // input (generate3DPerlinNoise is asset code), uploaded by the caller into "perlinNoise3D"
        ImageDescription fd = m_be.getImageDescription(m_volumetricIntegrationVolume);
        m_scatteringTransmittanceVolume = m_be.createImage(fd, nullptr, 0);
        m_volumetricLightingHistory[0] = m_be.createImage(fd, nullptr, 0);
        m_volumetricLightingHistory[1] = m_be.createImage(fd, nullptr, 0);
        m_volumeMaterialVolume = m_be.createImage(fd, nullptr, 0);
        ImageDescription nd;
        nd.width = nd.height = nd.depth = 32; nd.type = ImageType::Type3D; nd.format = ImageFormat::R8; nd.usageFlags = ImageUsageFlags::Sampled;
        m_perlinNoise3D = m_be.createImage(nd, nullptr, 0);
        ComputePassDescription d;
        d.name = "Froxel volume material"; d.shaderDescription.srcPathRelative = "froxelVolumeMaterial.comp";
        m_froxelVolumeMaterialPass = m_be.createComputePass(d);
        d.name = "Froxel light scattering"; d.shaderDescription.srcPathRelative = "froxelLightScattering.comp";
        m_froxelScatteringTransmittancePass = m_be.createComputePass(d);
        d.name = "Volumetric light integration"; d.shaderDescription.srcPathRelative = "volumetricLightingIntegration.comp";
        m_volumetricLightingIntegration = m_be.createComputePass(d);
        d.name = "Volumetric lighting reprojection"; d.shaderDescription.srcPathRelative = "volumeLightingReprojection.comp";
        m_volumetricLightingReprojection = m_be.createComputePass(d);
    }
    m_taa.init(m_be, W, H, s.taa);
    m_bloom.init(m_be);
    m_sdfGi.init(m_be, W, H, s.sdfTrace, s.sdfDebug, s.shading.sunShadowCascadeCount - 1, s.maxSdfInstances);
    const float influence = s.sdfTrace.traceInfluenceRadius;
    (void)influence;
}

ImageHandle FramePipeline::image(const std::string& n) const {
    if (n == "color0") return m_frameRenderTargets[0].colorBuffer;
    if (n == "color1") return m_frameRenderTargets[1].colorBuffer;
    if (n == "motion0") return m_frameRenderTargets[0].motionBuffer;
    if (n == "motion1") return m_frameRenderTargets[1].motionBuffer;
    if (n == "depth0") return m_frameRenderTargets[0].depthBuffer;
    if (n == "depth1") return m_frameRenderTargets[1].depthBuffer;
    if (n == "post0") return m_postProcessBuffers[0];
    if (n == "post1") return m_postProcessBuffers[1];
    if (n == "normal") return m_worldSpaceNormalImage;
    if (n == "albedo") return m_albedoImage;
    if (n == "specular") return m_specularImage;
    if (n == "pyramid") return m_minMaxDepthPyramid;
    if (n == "depthHalfRes") return m_depthHalfRes;
    if (n == "brdfLut") return m_brdfLut;
    if (n == "skyLut") return m_skyLut;
    if (n == "skyMultiscatterLut") return m_skyMultiscatterLut;
    if (n == "perlinNoise3D") return m_perlinNoise3D;
    if (n == "volumetricHistory0") return m_volumetricLightingHistory[0];
    if (n == "volumetricHistory1") return m_volumetricLightingHistory[1];
    if (n == "scatteringTransmittanceVolume") return m_scatteringTransmittanceVolume;
    if (n == "volumeMaterialVolume") return m_volumeMaterialVolume;
    if (n == "transmissionLut") return m_transmissionLut;
    if (n == "volumetricIntegrationVolume") return m_volumetricIntegrationVolume;
    if (n.size() == 7 && n.compare(0, 6, "shadow") == 0) return m_shadowMaps[(n[6] - '0') & 3];
    if (n.size() == 6 && n.compare(0, 5, "noise") == 0) return m_noiseTextures[(n[5] - '0') & 3];
    if (n == "taaHistory0") return m_taa.m_historyBuffers[0];
    if (n == "taaHistory1") return m_taa.m_historyBuffers[1];
    if (n == "sceneLuminance0") return m_taa.m_sceneLuminance[0];
    if (n == "sceneLuminance1") return m_taa.m_sceneLuminance[1];
    if (n == "giYSH0") return m_sdfGi.m_indirectDiffuse_Y_SH[0];
    if (n == "giYSH1") return m_sdfGi.m_indirectDiffuse_Y_SH[1];
    if (n == "giCoCg0") return m_sdfGi.m_indirectDiffuse_CoCg[0];
    if (n == "giCoCg1") return m_sdfGi.m_indirectDiffuse_CoCg[1];
    if (n == "giHistoryYSH0") return m_sdfGi.m_indirectDiffuseHistory_Y_SH[0];
    if (n == "giHistoryYSH1") return m_sdfGi.m_indirectDiffuseHistory_Y_SH[1];
    if (n == "giHistoryCoCg0") return m_sdfGi.m_indirectDiffuseHistory_CoCg[0];
    if (n == "giHistoryCoCg1") return m_sdfGi.m_indirectDiffuseHistory_CoCg[1];
    if (n == "giFullResYSH") return m_sdfGi.m_indirectLightingFullRes_Y_SH;
    if (n == "giFullResCoCg") return m_sdfGi.m_indirectLightingFullRes_CoCg;
    if (n == "swapchain") return const_cast<RenderBackend&>(m_be).getSwapchainInputImage();
    return ImageHandle{};
}

bool FramePipeline::storageBuffer(const std::string& n, StorageBufferHandle* out) const {
    if (n == "histogram") *out = m_histogramBuffer;
    else if (n == "histogramPerTile") *out = m_histogramPerTileBuffer;
    else if (n == "light") *out = m_lightBuffer;
    else if (n == "sunShadowInfo") *out = m_sunShadowInfoBuffer;
    else if (n == "sdfInstances") *out = m_sdfGi.m_sdfInstanceBuffer;
    else if (n == "sdfCulledInstances") *out = m_sdfGi.m_sdfCameraFrustumCulledInstances;
    else if (n == "sdfCulledTiles") *out = m_sdfGi.m_sdfCameraCulledTiles;
    else if (n == "sdfWorldBBs") *out = m_sdfGi.m_sdfInstanceWorldBBBuffer;
    else return false;
    return true;
}

bool FramePipeline::uniformBuffer(const std::string& n, UniformBufferHandle* out) const {
    if (n == "global") *out = m_globalUniformBuffer;
    else if (n == "volumetricSettings") *out = m_volumetricsInfoBuffer;
    else if (n == "taaResolveWeights") *out = m_taa.m_taaResolveWeightBuffer;
    else if (n == "sdfCameraFrustum") *out = m_sdfGi.m_cameraFrustumBuffer;
    else if (n == "sdfInfluenceRange") *out = m_sdfGi.m_sdfTraceInfluenceRangeBuffer;
    else return false;
    return true;
}

uint32_t FramePipeline::addSdfVolume(uint32_t res, const void* halfData, size_t bytes) {
    ImageDescription d;
    d.width = d.height = d.depth = res;
    d.type = ImageType::Type3D; d.format = ImageFormat::R16_sFloat; d.usageFlags = ImageUsageFlags::Sampled;
    const ImageHandle h = m_be.createImage(d, halfData, bytes);
    m_sdfVolumes.push_back(h);
    return m_be.getImageGlobalTextureArrayIndex(h);
}

uint32_t FramePipeline::addSdfVolumeFromDds(const std::string& path, ImageDescription* outDesc) {
    plr_image_desc cd{};
    size_t size = 0;
    if (plr_load_dds_file(path.c_str(), &cd, nullptr, 0, &size) != PLR_OK) throw std::runtime_error(plr_last_error());
    if (cd.format != PLR_FORMAT_R16_SFLOAT || cd.type != PLR_IMAGE_3D) throw std::runtime_error("SDF texture must be a 3D R16_sFloat DDS: " + path);
    std::vector<uint8_t> data(size);
    if (plr_load_dds_file(path.c_str(), &cd, data.data(), data.size(), &size) != PLR_OK) throw std::runtime_error(plr_last_error());
    ImageDescription d;
    d.width = cd.width; d.height = cd.height; d.depth = cd.depth;
    d.type = ImageType::Type3D; d.format = ImageFormat::R16_sFloat; d.usageFlags = ImageUsageFlags::Sampled;
    d.mipCount = cd.manual_mip_count > 1 ? MipCount::Manual : MipCount::One; d.manualMipCount = cd.manual_mip_count;
    if ((size_t)d.width * d.height * d.depth * 2 > size) throw std::runtime_error("DDS file holds fewer texels than its header declares: " + path);
    const ImageHandle h = m_be.createImage(d, data.data(), data.size());
    m_sdfVolumes.push_back(h);
    if (outDesc) *outDesc = d;
    return m_be.getImageGlobalTextureArrayIndex(h);
}

void FramePipeline::setSdfScene(const void* inst, size_t instBytes, const void* bb, size_t bbBytes) { m_sdfGi.updateSDFScene(m_be, inst, instBytes, bb, bbBytes); }
void FramePipeline::setSunDirection(const float d[3]) { m_globalShaderInfo.sunDirection[0] = d[0]; m_globalShaderInfo.sunDirection[1] = d[1]; m_globalShaderInfo.sunDirection[2] = d[2]; m_globalShaderInfo.sunDirection[3] = 0.f; }
void FramePipeline::setCameraIntrinsic(float fov, float n, float f) { m_cameraIntrinsic.fov = fov; m_cameraIntrinsic.near = n; m_cameraIntrinsic.far = f; }

void FramePipeline::computeColorBufferHistogram(ImageHandle lastFrameColor) { // RenderFrontend.cpp:707-754
    const StorageBufferResource histogramPerTileResource(m_histogramPerTileBuffer, false, 0);
    const StorageBufferResource histogramResource(m_histogramBuffer, false, 1);
    const uint32_t W = settings.width, H = settings.height;
    const RowRange tileRows = scaleRows(bandRows(0), histogramTileSizeY);
    {
        ComputePassExecution exe;
        exe.genericInfo.handle = m_histogramPerTilePass;
        exe.genericInfo.resources.storageBuffers = {histogramPerTileResource, StorageBufferResource(m_lightBuffer, true, 3)};
        exe.genericInfo.resources.sampledImages = {ImageResource(lastFrameColor, 0, 2)};
        exe.dispatchCount[0] = uint32_t(std::ceil((float)W / float(histogramTileSizeX)));
        exe.dispatchCount[1] = uint32_t(std::ceil((float)H / float(histogramTileSizeY)));
        exe.dispatchCount[2] = 1;
        if (settings.band.enabled()) { // the band's tile rows; the global histogram is the sum over bands (ExchangeHistogram)
            exe.dispatchBase[1] = tileRows.begin;
            exe.dispatchCount[1] = tileRows.end - tileRows.begin;
            if (settings.band.tiled()) { // and the tile's tile columns
                const ColRange tileCols = scaleRows(bandCols(0), histogramTileSizeX);
                exe.dispatchBase[0] = tileCols.begin;
                exe.dispatchCount[0] = tileCols.end - tileCols.begin;
            }
        }
        m_be.setComputePassExecution(exe);
    }
    const float binsPerDispatch = 64.f;
    {
        ComputePassExecution exe;
        exe.genericInfo.handle = m_histogramResetPass;
        exe.genericInfo.resources.storageBuffers = {histogramResource};
        exe.dispatchCount[0] = uint32_t(std::ceil(float(nHistogramBins) / binsPerDispatch));
        m_be.setComputePassExecution(exe);
    }
    {
        ComputePassExecution exe;
        exe.genericInfo.handle = m_histogramCombinePass;
        exe.genericInfo.resources.storageBuffers = {histogramPerTileResource, histogramResource};
        exe.dispatchCount[0] = (uint32_t)std::ceil(W / float(histogramTileSizeX)) * (uint32_t)std::ceil(H / float(histogramTileSizeY));
        exe.dispatchCount[1] = uint32_t(std::ceil(float(nHistogramBins) / binsPerDispatch));
        if (settings.band.enabled()) {
            // (tile rendering: the per-tile buffer is indexed with the whole frame's stride, so a tile's entries are not one range of it; the combine runs over the
            //  whole width of the tile's tile rows and the entries of other GPUs' tiles stay what the buffer was created with - zeros: an integer sum)
            const uint32_t tilesX = (uint32_t)std::ceil(W / float(histogramTileSizeX));
            exe.dispatchBase[0] = tileRows.begin * tilesX;
            exe.dispatchCount[0] = (tileRows.end - tileRows.begin) * tilesX;
        }
        m_be.setComputePassExecution(exe);
    }
}

void FramePipeline::computeExposure() { // RenderFrontend.cpp:776-790
    ComputePassExecution exe;
    exe.genericInfo.handle = m_preExposeLightsPass;
    exe.genericInfo.resources.storageBuffers = {StorageBufferResource(m_histogramBuffer, false, 1), StorageBufferResource(m_lightBuffer, false, 0)};
    exe.genericInfo.resources.sampledImages = {ImageResource(m_transmissionLut, 0, 2)};
    m_be.setComputePassExecution(exe);
}

void FramePipeline::computeDepthPyramid(ImageHandle depthBuffer) { // RenderFrontend.cpp:804-838
    ComputePassExecution exe;
    exe.genericInfo.handle = m_depthPyramidPass;
    const uint32_t width = settings.width / 2, height = settings.height / 2, maxMipCount = 11;
    const uint32_t mipCount = pyramidMipCount(settings);
    uint32_t dc[2];
    singlePassMipChainDispatchCount(width, height, mipCount, maxMipCount, dc);
    exe.dispatchCount[0] = dc[0]; exe.dispatchCount[1] = dc[1]; exe.dispatchCount[2] = 1;
    if (perTilePyramid(settings)) { // one workgroup per 32x32 texels of pyramid mip 0 = 64 full-resolution rows
        exe.dispatchCount[0] = (width + 31) / 32;
        exe.dispatchCount[1] = (height + 31) / 32;
        if (settings.band.enabled()) {
            const RowRange groups = scaleRows(bandRows(0), 64);
            exe.dispatchBase[1] = groups.begin;
            exe.dispatchCount[1] = groups.end - groups.begin;
            if (settings.band.tiled()) {
                const ColRange groupsX = scaleRows(bandCols(0), 64);
                exe.dispatchBase[0] = groupsX.begin;
                exe.dispatchCount[0] = groupsX.end - groupsX.begin;
            }
        }
    }
    exe.genericInfo.resources.sampledImages = {ImageResource(depthBuffer, 0, 13), ImageResource(m_minMaxDepthPyramid, 0, 15)};
    exe.genericInfo.resources.storageBuffers = {StorageBufferResource(m_depthPyramidSyncBuffer, false, 16)};
    const uint32_t unusedMipCount = maxMipCount - mipCount;
    for (uint32_t i = 0; i < maxMipCount; i++) {
        const uint32_t mipLevel = i >= unusedMipCount ? i - unusedMipCount : 0;
        exe.genericInfo.resources.storageImages.push_back(ImageResource(m_minMaxDepthPyramid, mipLevel, i));
    }
    m_be.setComputePassExecution(exe);
}

void FramePipeline::downscaleDepth(const FrameRenderTargets& currentTarget) { // RenderFrontend.cpp:873-892
    ComputePassExecution exe;
    exe.genericInfo.handle = m_depthDownscalePass;
    // band: the spatial filters sample half-res depth up to giHalo trace rows away, the upscale one more
    const uint32_t depthHalo = 2 * (settings.band.giHalo + settings.band.giHistoryHalo) + 16;
    dispatch8(exe, settings.width / 2, settings.height / 2, bandRows(depthHalo, 2), bandCols(depthHalo, 2));
    exe.genericInfo.resources.storageImages = {ImageResource(m_depthHalfRes, 0, 0)};
    exe.genericInfo.resources.sampledImages = {ImageResource(currentTarget.depthBuffer, 0, 1)};
    m_be.setComputePassExecution(exe);
}

void FramePipeline::computeDeferredShading(ImageHandle colorTarget, const FrameRenderTargets& current) { // renderForwardShading, RenderFrontend.cpp:894-929
    ComputePassExecution exe;
    exe.genericInfo.handle = m_deferredShadingPass;
    exe.genericInfo.resources.storageBuffers = {StorageBufferResource(m_lightBuffer, true, 7), StorageBufferResource(m_sunShadowInfoBuffer, true, 8)};
    const SDFGI::IndirectLightingImages indirectLight = m_sdfGi.getIndirectLightingResults(settings.sdfTrace.halfResTrace);
    exe.genericInfo.resources.storageImages = {ImageResource(colorTarget, 0, 0)};
    exe.genericInfo.resources.sampledImages = {ImageResource(m_brdfLut, 0, 3), ImageResource(indirectLight.Y_SH, 0, 15), ImageResource(indirectLight.CoCg, 0, 16),
                                               ImageResource(m_volumetricIntegrationVolume, 0, 18),
                                               ImageResource(current.depthBuffer, 0, 20), ImageResource(m_worldSpaceNormalImage, 0, 21),
                                               ImageResource(m_albedoImage, 0, 22), ImageResource(m_specularImage, 0, 23), ImageResource(m_skyLut, 0, 24)};
    for (uint32_t i = 0; i < (uint32_t)maxSunShadowCascadeCount; i++) exe.genericInfo.resources.sampledImages.push_back(ImageResource(m_shadowMaps[i], 0, 9 + i));
    exe.genericInfo.resources.uniformBuffers = {UniformBufferResource(m_volumetricsInfoBuffer, 19)};
    dispatch8(exe, settings.width, settings.height, bandRows(settings.band.colorHalo), bandCols(settings.band.colorHalo));
    m_be.setComputePassExecution(exe);
}

void FramePipeline::computeTonemapping(ImageHandle src) { // RenderFrontend.cpp:931-945
    ComputePassExecution exe;
    exe.genericInfo.handle = m_tonemappingPass;
    exe.genericInfo.resources.storageImages = {ImageResource(m_be.getSwapchainInputImage(), 0, 0)};
    exe.genericInfo.resources.sampledImages = {ImageResource(src, 0, 1)};
    dispatch8(exe, settings.width, settings.height, bandRows(0), bandCols(0));
    exe.asyncTail = asyncPostTail();
    m_be.setComputePassExecution(exe);
}

// band rendering / a frame too large for one pyramid: the top level of the per-tile pyramid (one texel per 64 x 64 pixels), the band's rows of it, reduced to
// one texel; with an exchange callback the bands' texels are then combined (min on .r, max on .g) - SURVEY 8e's second collective
void FramePipeline::computeDepthApexOfTiles() {
    ComputePassExecution exe;
    exe.genericInfo.handle = m_depthApexPass;
    const uint32_t top = bandPyramidMipCount - 1;
    const uint32_t levelRows = std::max((settings.height / 2) >> top, 1u);
    RowRange rows{0, levelRows};
    if (settings.band.enabled()) {
        rows = scaleRows(bandRows(0), 64); // 64 full-resolution rows per texel of that level
        rows.end = std::min(rows.end, levelRows);
    }
    exe.dispatchCount[0] = 1; exe.dispatchCount[2] = 1; // (one workgroup from column 0: whole rows)
    exe.dispatchBase[1] = rows.begin;
    exe.dispatchCount[1] = rows.end - rows.begin;
    if (settings.band.enabled() && settings.band.tiled()) { // the tile's texel columns of that level
        const uint32_t levelCols = std::max((settings.width / 2) >> top, 1u);
        ColRange cols = scaleRows(bandCols(0), 64);
        cols.end = std::min(cols.end, levelCols);
        exe.dispatchBase[0] = cols.begin;
        exe.dispatchCount[0] = cols.end - cols.begin;
        const int32_t columnRange = 1; // the x range is a range of texel columns, also when it is one column from column 0
        exe.pushConstants = dataToCharArray(&columnRange, sizeof(columnRange));
    }
    exe.genericInfo.resources.sampledImages = {ImageResource(m_minMaxDepthPyramid, top, 0)};
    exe.genericInfo.resources.storageImages = {ImageResource(m_bandDepthApex, 0, 1)};
    m_be.setComputePassExecution(exe);
    if (settings.band.enabled() && m_exchangeFn) {
        m_be.setHostCallbackExecution(&FramePipeline::exchangeTrampoline, &m_exchangeCtx[ExchangeDepthApex], "Exchange: depth range all-reduce (min, max)", {m_bandDepthApex}, {});
    }
}

void FramePipeline::computeSunLightMatrices() { // RenderFrontend.cpp:840-872
    ComputePassExecution exe;
    exe.genericInfo.handle = m_lightMatrixPass;
    exe.dispatchCount[0] = exe.dispatchCount[1] = exe.dispatchCount[2] = 1;
    const uint32_t depthPyramidMipCount = mipCountFromResolution(settings.width / 2, settings.height / 2, 1);
    if (perTilePyramid(settings)) exe.genericInfo.resources.storageImages = {ImageResource(m_bandDepthApex, 0, 1)};
    else exe.genericInfo.resources.storageImages = {ImageResource(m_minMaxDepthPyramid, depthPyramidMipCount - 1, 1)};
    exe.genericInfo.resources.storageBuffers = {StorageBufferResource(m_sunShadowInfoBuffer, false, 0)};
    struct LightMatrixPushConstants { float highestCascadePaddingSize; float highestCascadeMinFarPlane; } pc;
    pc.highestCascadePaddingSize = settings.sdfTrace.traceInfluenceRadius;
    pc.highestCascadeMinFarPlane = settings.volumetricsMaxDistance;
    // if strict cutoff enabled all hits outside influence radius are discarded anyways
    if (!settings.sdfTrace.strictInfluenceRadiusCutoff) pc.highestCascadePaddingSize += settings.sdfTrace.additionalSunShadowMapPadding;
    exe.pushConstants = dataToCharArray(&pc, sizeof(pc));
    m_be.setComputePassExecution(exe);
}

void FramePipeline::computeVolumetricLighting(float deltaTime) { // Volumetrics::computeVolumetricLighting, Techniques/Volumetrics.cpp:119-243
    m_volumetricsState.sampleOffset = radicalInverseBase2((uint32_t)m_frameIndex.mod8()) - 0.5f; // hammersley2D(frameIndexMod8).x - 0.5
    for (int i = 0; i < 3; i++) m_volumetricsState.windSampleOffset[i] += windSettings.vector[i] * windSettings.speed * deltaTime;
    VolumetricsBufferContents contents;
    contents.state = m_volumetricsState;
    contents.settings = volumetricsSettings;
    m_be.setUniformBufferData(m_volumetricsInfoBuffer, &contents, sizeof(contents));
    const ImageDescription md = m_be.getImageDescription(m_volumeMaterialVolume);
    auto groups = [&](ComputePassExecution& exe, float groupSize, bool flat) {
        exe.dispatchCount[0] = (uint32_t)std::ceil(md.width / groupSize);
        exe.dispatchCount[1] = (uint32_t)std::ceil(md.height / groupSize);
        exe.dispatchCount[2] = flat ? 1u : (uint32_t)std::ceil(md.depth / groupSize);
    };
    {
        ComputePassExecution exe;
        exe.genericInfo.handle = m_froxelVolumeMaterialPass;
        exe.genericInfo.resources.storageImages = {ImageResource(m_volumeMaterialVolume, 0, 0)};
        exe.genericInfo.resources.sampledImages = {ImageResource(m_perlinNoise3D, 0, 1)};
        exe.genericInfo.resources.uniformBuffers = {UniformBufferResource(m_volumetricsInfoBuffer, 2)};
        groups(exe, 4.f, false);
        m_be.setComputePassExecution(exe);
    }
    {
        ComputePassExecution exe;
        exe.genericInfo.handle = m_froxelScatteringTransmittancePass;
        exe.genericInfo.resources.storageImages = {ImageResource(m_scatteringTransmittanceVolume, 0, 0)};
        exe.genericInfo.resources.sampledImages = {ImageResource(m_shadowMaps[settings.shading.sunShadowCascadeCount - 1], 0, 1), ImageResource(m_volumeMaterialVolume, 0, 2)};
        exe.genericInfo.resources.storageBuffers = {StorageBufferResource(m_sunShadowInfoBuffer, true, 3), StorageBufferResource(m_lightBuffer, true, 4)};
        exe.genericInfo.resources.uniformBuffers = {UniformBufferResource(m_volumetricsInfoBuffer, 5)};
        groups(exe, 4.f, false);
        m_be.setComputePassExecution(exe);
    }
    const size_t frameIndexMod2 = m_frameIndex.mod2();
    const ImageHandle reprojectionTarget = m_volumetricLightingHistory[frameIndexMod2];
    const ImageHandle reprojectionHistory = m_volumetricLightingHistory[(frameIndexMod2 + 1) % 2];
    {
        ComputePassExecution exe;
        exe.genericInfo.handle = m_volumetricLightingReprojection;
        exe.genericInfo.resources.storageImages = {ImageResource(reprojectionTarget, 0, 0)};
        exe.genericInfo.resources.sampledImages = {ImageResource(m_scatteringTransmittanceVolume, 0, 1), ImageResource(reprojectionHistory, 0, 2)};
        exe.genericInfo.resources.uniformBuffers = {UniformBufferResource(m_volumetricsInfoBuffer, 3)};
        groups(exe, 4.f, false);
        m_be.setComputePassExecution(exe);
    }
    {
        ComputePassExecution exe;
        exe.genericInfo.handle = m_volumetricLightingIntegration;
        exe.genericInfo.resources.storageImages = {ImageResource(m_volumetricIntegrationVolume, 0, 0)};
        exe.genericInfo.resources.sampledImages = {ImageResource(reprojectionTarget, 0, 1)};
        exe.genericInfo.resources.uniformBuffers = {UniformBufferResource(m_volumetricsInfoBuffer, 2)};
        groups(exe, 8.f, true);
        m_be.setComputePassExecution(exe);
    }
}

void FramePipeline::updateTransmissionLut() { // Sky::updateTransmissionLut, Techniques/Sky.cpp:260-272
    const uint32_t res = 128;
    ComputePassExecution exe;
    exe.genericInfo.handle = m_skyTransmissionLutPass;
    exe.genericInfo.resources.storageImages = {ImageResource(m_transmissionLut, 0, 0)};
    exe.genericInfo.resources.uniformBuffers = {UniformBufferResource(m_atmosphereSettingsBuffer, 1)};
    exe.dispatchCount[0] = res / 8; exe.dispatchCount[1] = res / 8; exe.dispatchCount[2] = 1;
    m_be.setComputePassExecution(exe);
}

void FramePipeline::updateSkyLut() { // Sky::updateSkyLut, Techniques/Sky.cpp:274-316
    m_be.setUniformBufferData(m_atmosphereSettingsBuffer, &atmosphereSettings, sizeof(atmosphereSettings));
    {
        const uint32_t res = 32;
        ComputePassExecution exe;
        exe.genericInfo.handle = m_skyMultiscatterLutPass;
        exe.genericInfo.resources.storageImages = {ImageResource(m_skyMultiscatterLut, 0, 0)};
        exe.genericInfo.resources.sampledImages = {ImageResource(m_transmissionLut, 0, 1)};
        exe.genericInfo.resources.uniformBuffers = {UniformBufferResource(m_atmosphereSettingsBuffer, 3)};
        exe.dispatchCount[0] = res / 8; exe.dispatchCount[1] = res / 8; exe.dispatchCount[2] = 1;
        m_be.setComputePassExecution(exe);
    }
    {
        const uint32_t w = 200, h = 100; // sic: 100 / 8 = 12 workgroups, the last four rows of the LUT are never written (Sky.cpp:311-312)
        ComputePassExecution exe;
        exe.genericInfo.handle = m_skyLutPass;
        exe.genericInfo.resources.storageImages = {ImageResource(m_skyLut, 0, 0)};
        exe.genericInfo.resources.sampledImages = {ImageResource(m_transmissionLut, 0, 1), ImageResource(m_skyMultiscatterLut, 0, 2)};
        exe.genericInfo.resources.uniformBuffers = {UniformBufferResource(m_atmosphereSettingsBuffer, 4)};
        exe.genericInfo.resources.storageBuffers = {StorageBufferResource(m_lightBuffer, true, 5)};
        exe.dispatchCount[0] = w / 8; exe.dispatchCount[1] = h / 8; exe.dispatchCount[2] = 1;
        m_be.setComputePassExecution(exe);
    }
}

void FramePipeline::computeBRDFLut() { // RenderFrontend.cpp:1031-1042
    ComputePassExecution exe;
    exe.genericInfo.handle = m_brdfLutPass;
    exe.genericInfo.resources.storageImages = {ImageResource(m_brdfLut, 0, 0)};
    dispatch8(exe, settings.brdfLutRes, settings.brdfLutRes);
    m_be.setComputePassExecution(exe);
}

RowRange FramePipeline::bandRows(uint32_t halo, uint32_t divisor) const {
    if (!settings.band.enabled()) return {};
    const uint32_t lo = settings.band.rowBegin > halo ? settings.band.rowBegin - halo : 0;
    const uint32_t hi = std::min(settings.band.rowEnd + halo, settings.height);
    return {lo / divisor, (hi + divisor - 1) / divisor};
}

ColRange FramePipeline::bandCols(uint32_t halo, uint32_t divisor) const {
    if (!settings.band.enabled() || !settings.band.tiled()) return {};
    const uint32_t lo = settings.band.colBegin > halo ? settings.band.colBegin - halo : 0;
    const uint32_t hi = std::min(settings.band.colEnd + halo, settings.width);
    return {lo / divisor, (hi + divisor - 1) / divisor};
}

void FramePipeline::addExchangeItem(int id, ImageHandle image, uint32_t divisor, uint32_t haloRows) {
    const ImageDescription d = m_be.getImageDescription(image);
    ExchangeItem it;
    it.image = image;
    it.rowBegin = settings.band.rowBegin / divisor;
    it.rowEnd = std::min((settings.band.rowEnd + divisor - 1) / divisor, d.height);
    it.haloRows = haloRows;
    it.imageRows = d.height;
    size_t bytes = 0;
    void* ptr = nullptr;
    m_be.getImageDevicePointer(image, 0, &ptr, &bytes);
    it.rowBytes = (uint32_t)(bytes / d.height);
    it.imageCols = d.width;
    it.texelBytes = it.rowBytes / std::max(d.width, 1u);
    it.colBegin = settings.band.tiled() ? settings.band.colBegin / divisor : 0u;
    it.colEnd = settings.band.tiled() ? std::min((settings.band.colEnd + divisor - 1) / divisor, d.width) : d.width;
    m_exchangeItems[id].push_back(it);
}

int FramePipeline::exchangeTrampoline(void* user, void* stream) {
    const ExchangeCtx* ctx = (const ExchangeCtx*)user;
    return ctx->self->m_exchangeFn ? ctx->self->m_exchangeFn(ctx->self->m_exchangeUser, ctx->id, stream) : 0;
}

void FramePipeline::exchangePoint(int id, const char* label) {
    // id may carry a phase (ExchangeBegin / ExchangeEnd)
    const int phase = (id & ExchangeBegin) ? 1 : ((id & ExchangeEnd) ? 2 : 0);
    if (!m_exchangeFn) return;
    // what the exchange moves: the registered images' halo rows, or the histogram buffer (the backend orders the frame's asynchronous tail by it)
    std::vector<ImageHandle> images;
    std::vector<StorageBufferHandle> buffers;
    if ((id & ExchangeIdMask) == ExchangeHistogram) buffers.push_back(m_histogramBuffer);
    else if ((id & ExchangeIdMask) == ExchangeGiRequests) { buffers.push_back(m_sdfGi.m_giRequestBitmap[0]); buffers.push_back(m_sdfGi.m_giRequestBitmap[1]); }
    else for (const ExchangeItem& it : m_exchangeItems[id & ExchangeIdMask]) images.push_back(it.image);
    if (settings.band.giRequested && ((id & ExchangeIdMask) == ExchangeGiTrace || (id & ExchangeIdMask) == ExchangeGiTemporal)) {
        images.push_back(m_depthHalfRes); // the requested texels arrive with their depth
        buffers.push_back(m_sdfGi.m_giRequestBitmap[(id & ExchangeIdMask) == ExchangeGiTrace ? 0 : 1]);
    }
    m_be.setHostCallbackExecution(&FramePipeline::exchangeTrampoline, &m_exchangeCtx[phase * ExchangeCount + (id & ExchangeIdMask)], label, images, buffers);
}

void FramePipeline::prepareRenderpasses() { // RenderFrontend.cpp:313-406
    const FrameRenderTargets previousRenderTarget = m_frameRenderTargets[m_sceneRenderTargetIndex];
    m_sceneRenderTargetIndex = (m_sceneRenderTargetIndex + 1) % 2;
    const FrameRenderTargets currentRenderTarget = m_frameRenderTargets[m_sceneRenderTargetIndex];

    const bool bandMode = settings.band.enabled();
    if (settings.sdfDebug.visualisationMode != SDFVisualisationMode::None) { // RenderFrontend.cpp:321-340
        if (bandMode) throw std::runtime_error("the SDF debug visualisation is not supported in band rendering");
        for (auto& items : m_exchangeItems) items.clear();
        // [renderDepthPrepass: input]
        computeDepthPyramid(currentRenderTarget.depthBuffer);
        computeColorBufferHistogram(m_postProcessBuffers[0]);
        if (settings.runSkyLuts) updateTransmissionLut();
        computeExposure();
        if (settings.runSkyLuts) updateSkyLut();
        if (settings.runLightMatrix) computeSunLightMatrices();
        // [renderSunShadowCascades: input]
        SDFTraceDependencies deps = m_frustumScratch;
        deps.currentFrame = currentRenderTarget;
        deps.previousFrame = previousRenderTarget;
        deps.depthHalfRes = m_depthHalfRes;
        deps.worldSpaceNormals = m_worldSpaceNormalImage;
        deps.skyLut = m_skyLut;
        deps.shadowMap = m_shadowMaps[settings.shading.sunShadowCascadeCount - 1];
        deps.lightBuffer = m_lightBuffer;
        deps.sunShadowInfoBuffer = m_sunShadowInfoBuffer;
        deps.depthMinMaxPyramid = m_minMaxDepthPyramid;
        m_sdfGi.renderSDFVisualization(m_be, m_postProcessBuffers[0], deps, settings.sdfDebug, settings.sdfTrace);
        computeTonemapping(m_postProcessBuffers[0]);
        return;
    }
    if (m_isBRDFLutShaderDescriptionStale) {
        computeBRDFLut();
        m_isBRDFLutShaderDescriptionStale = false;
    }
    const bool band = settings.band.enabled();
    for (auto& items : m_exchangeItems) items.clear();
    if (settings.runExposure) {
        computeColorBufferHistogram(previousRenderTarget.colorBuffer);
        if (band) exchangePoint(ExchangeHistogram, "Exchange: histogram all-reduce");
        if (settings.runSkyLuts) updateTransmissionLut(); // separate from the sky LUT: the exposure pass reads it, the other LUTs depend on the exposure
        computeExposure();
    } else if (settings.runSkyLuts) updateTransmissionLut();
    if (settings.runSkyLuts) updateSkyLut();
    // [renderDepthPrepass: input]
    if (settings.runHiZ) computeDepthPyramid(currentRenderTarget.depthBuffer);
    if (settings.runLightMatrix && settings.runHiZ) {
        if (perTilePyramid(settings)) computeDepthApexOfTiles(); // no apex in a per-tile pyramid: reduce (and, across bands, all-reduce) it
        computeSunLightMatrices();
    }
    // [renderSunShadowCascades: input]
    if (settings.runGI && settings.shading.indirectLightingTech == IndirectLightingTech::SDFTrace) {
        if (settings.sdfTrace.halfResTrace) downscaleDepth(currentRenderTarget);
        SDFTraceDependencies deps = m_frustumScratch; // frustum points/normals from setCameraExtrinsic (fillOutSdfGiDependencies, :1073-1090)
        deps.currentFrame = currentRenderTarget;
        deps.previousFrame = previousRenderTarget;
        deps.depthHalfRes = m_depthHalfRes;
        deps.worldSpaceNormals = m_worldSpaceNormalImage;
        deps.skyLut = m_skyLut;
        deps.shadowMap = m_shadowMaps[settings.shading.sunShadowCascadeCount - 1];
        deps.lightBuffer = m_lightBuffer;
        deps.sunShadowInfoBuffer = m_sunShadowInfoBuffer;
        deps.depthMinMaxPyramid = m_minMaxDepthPyramid;
        if (band) {
            const uint32_t div = settings.sdfTrace.halfResTrace ? 2 : 1;
            GiBand gb;
            gb.traceRows = bandRows(0, div);
            gb.upscaleRows = bandRows(settings.band.colorHalo);
            gb.traceCols = bandCols(0, div);
            gb.upscaleCols = bandCols(settings.band.colorHalo);
            gb.user = this;
            gb.giHalo = settings.band.giHalo; gb.giHistoryHalo = settings.band.giHistoryHalo;
            gb.rowsFirst = settings.band.rowsFirst;
            gb.requested = settings.band.giRequested;
            // registers the images of a GI exchange and records its callback: phase 0 = whole exchange, ExchangeBegin after the producer's
            // edge rows, ExchangeEnd (items already registered) before the consumer
            static const auto giExchange = [](FramePipeline* self, int id, int phase) {
                const SDFGI& gi = self->m_sdfGi;
                const uint32_t d = self->settings.sdfTrace.halfResTrace ? 2 : 1;
                const BandSettings& b = self->settings.band;
                const bool reg = phase != ExchangeEnd;
                const char* what = phase == ExchangeBegin ? " (start)" : (phase == ExchangeEnd ? " (wait)" : "");
                if (id == ExchangeGiRequests) {
                    self->exchangePoint(id, "Exchange: GI sample requests");
                } else if (id == ExchangeGiTrace) {
                    if (reg) { self->addExchangeItem(id, gi.m_indirectDiffuse_Y_SH[0], d, b.giHalo); self->addExchangeItem(id, gi.m_indirectDiffuse_CoCg[0], d, b.giHalo); }
                    self->exchangePoint(id | phase, (std::string("Exchange: traced GI halo rows") + what).c_str());
                } else if (id == ExchangeGiTemporal) {
                    if (reg) { self->addExchangeItem(id, gi.m_indirectDiffuseHistory_Y_SH[1], d, b.giHalo); self->addExchangeItem(id, gi.m_indirectDiffuseHistory_CoCg[1], d, b.giHalo); }
                    self->exchangePoint(id | phase, (std::string("Exchange: temporally filtered GI halo rows") + what).c_str());
                } else {
                    if (reg) { self->addExchangeItem(id, gi.m_indirectDiffuseHistory_Y_SH[0], d, b.giHistoryHalo); self->addExchangeItem(id, gi.m_indirectDiffuseHistory_CoCg[0], d, b.giHistoryHalo); }
                    self->exchangePoint(id | phase, (std::string("Exchange: GI history halo rows") + what).c_str());
                }
            };
            const bool overlap = settings.band.overlapExchange && m_exchangeFn && !settings.band.giRequested; // (a requested texel can be anywhere in its owner's rectangle: no edges-first producers)
            if (settings.band.giRequested && settings.band.overlapExchange && m_exchangeFn) {
                gb.requestedBegin = [](void* user, int id) { giExchange((FramePipeline*)user, id, ExchangeBegin); };
                gb.requestedEnd = [](void* user, int id) { giExchange((FramePipeline*)user, id, ExchangeEnd); };
            }
            if (overlap) {
                gb.exchangeBegin = [](void* user, int id) { giExchange((FramePipeline*)user, id, ExchangeBegin); };
                gb.exchangePoint = [](void* user, int id) { giExchange((FramePipeline*)user, id, ExchangeEnd); };
                gb.exchangeWhole = [](void* user, int id) { giExchange((FramePipeline*)user, id, 0); };
            } else gb.exchangePoint = [](void* user, int id) { giExchange((FramePipeline*)user, id, 0); };
            m_sdfGi.computeIndirectLighting(m_be, m_frameIndex, deps, settings.sdfTrace, &gb);
        } else m_sdfGi.computeIndirectLighting(m_be, m_frameIndex, deps, settings.sdfTrace);
    }
    if (settings.runVolumetrics) {
        if (band) throw std::runtime_error("the volumetric froxel passes are not supported in band rendering");
        computeVolumetricLighting(m_lastDeltaTime);
    }
    if (settings.runShading) computeDeferredShading(currentRenderTarget.colorBuffer, currentRenderTarget);
    // [sky: folded into the deferred pass' sky stand-in]
    ImageHandle currentSrc = currentRenderTarget.colorBuffer;
    bool postExchangeStarted = false;
    if (settings.runTAA && settings.taa.enabled) {
        if (settings.taa.useSeparateSupersampling) { // RenderFrontend.cpp:391-395 (whole-frame only: the stage has no halo plan for band rendering)
            if (band) throw std::runtime_error("useSeparateSupersampling is not supported in band rendering");
            m_taa.computeTemporalSuperSampling(m_be, m_frameIndex, currentRenderTarget, previousRenderTarget, m_postProcessBuffers[0]);
            currentSrc = m_postProcessBuffers[0];
        }
        // the bloom chain reads postHalo rows around the band; next frame's temporal filter reprojects into the history image
        const bool bloomOn = settings.runBloom && settings.bloom.enabled;
        if (band && settings.band.overlapExchange && m_exchangeFn) {
            const uint32_t edge = bloomOn ? std::max(settings.band.postHalo, settings.band.taaHistoryHalo) : settings.band.taaHistoryHalo;
            m_taa.computeTemporalFilter(m_be, m_frameIndex, currentSrc, currentRenderTarget, m_postProcessBuffers[1], bandRows(0), edge, [&] {
                if (bloomOn) addExchangeItem(ExchangePost, m_postProcessBuffers[1], 1, settings.band.postHalo);
                addExchangeItem(ExchangePost, m_taa.historyDst(m_frameIndex), 1, settings.band.taaHistoryHalo);
                exchangePoint(ExchangePost | ExchangeBegin, "Exchange: resolved colour halo rows (start)");
            }, settings.band.rowsFirst, bandCols(0));
            postExchangeStarted = true;
        } else m_taa.computeTemporalFilter(m_be, m_frameIndex, currentSrc, currentRenderTarget, m_postProcessBuffers[1], bandRows(0), 0, nullptr, false, bandCols(0));
        currentSrc = m_postProcessBuffers[1];
    }
    if (band) {
        if (postExchangeStarted) exchangePoint(ExchangePost | ExchangeEnd, "Exchange: resolved colour halo rows (wait)");
        else {
            if (settings.runBloom && settings.bloom.enabled) addExchangeItem(ExchangePost, currentSrc, 1, settings.band.postHalo);
            if (settings.runTAA && settings.taa.enabled) addExchangeItem(ExchangePost, m_taa.historyDst(m_frameIndex), 1, settings.band.taaHistoryHalo);
            if (!m_exchangeItems[ExchangePost].empty()) exchangePoint(ExchangePost, "Exchange: resolved colour halo rows");
        }
    }
    // the bloom chain and the tonemap are the frame's asynchronous tail (plr.h async_tail): nothing reads their outputs before the next frame's TAA
    // resolve, and their ten short launches leave the chip mostly idle - they run beside the next frame's exposure / GI / shade passes. In band
    // rendering too: the exchange callbacks are recorded with the images they move (exchangePoint), and only the resolved-colour exchange shares one.
    if (settings.runBloom && settings.bloom.enabled) m_bloom.computeBloom(m_be, currentSrc, settings.bloom, bandRows(settings.band.postHalo), bandRows(0), asyncPostTail(), bandCols(settings.band.postHalo), bandCols(0));
    if (settings.runTonemap) computeTonemapping(currentSrc);
}

// ViewFrustum.cpp:4-52 + the packing of SDFGI.cpp:543-566 (top, bot, near, far, left, right)
static void computeFrustum(const CameraExtrinsic& e, const CameraIntrinsic& in, float points[6][4], float normals[6][4]) {
    const Vec3 nearC = e.position + e.forward * in.near, farC = e.position + e.forward * in.far;
    const float tanFoV = std::tan(in.fov * 3.14159265358979f / 180.f * 0.5f);
    const float hn = tanFoV * in.near, hf = tanFoV * in.far, wn = hn * in.aspectRatio, wf = hf * in.aspectRatio;
    const Vec3 r_u_f = farC + e.up * hf + e.right * wf, l_u_f = farC + e.up * hf - e.right * wf, r_l_f = farC - e.up * hf + e.right * wf, l_l_f = farC - e.up * hf - e.right * wf;
    const Vec3 r_u_n = nearC + e.up * hn + e.right * wn, l_u_n = nearC + e.up * hn - e.right * wn, r_l_n = nearC - e.up * hn + e.right * wn, l_l_n = nearC - e.up * hn - e.right * wn;
    const Vec3 top = normalize(cross(r_u_f - r_u_n, r_u_n - l_u_n)), bot = normalize(cross(r_l_n - l_l_n, r_l_f - r_l_n));
    const Vec3 right = normalize(cross(r_u_n - r_l_n, r_l_f - r_l_n)), left = normalize(cross(l_l_f - l_l_n, l_u_n - l_l_n));
    const Vec3 nearN = normalize(cross(r_u_n - r_l_n, r_l_n - l_l_n)), farN = normalize(cross(r_l_f - l_l_f, r_u_f - r_l_f));
    const Vec3 P[6] = {l_u_f, l_l_f, l_l_n, l_l_f, l_l_f, r_l_f};
    const Vec3 N[6] = {top, bot, nearN, farN, left, right};
    for (int i = 0; i < 6; i++) {
        points[i][0] = P[i].x; points[i][1] = P[i].y; points[i][2] = P[i].z; points[i][3] = 0.f;
        normals[i][0] = N[i].x; normals[i][1] = N[i].y; normals[i][2] = N[i].z; normals[i][3] = 0.f;
    }
}

void FramePipeline::setCameraExtrinsic(const CameraExtrinsic& extrinsic) { // RenderFrontend.cpp:423-454
    m_globalShaderInfo.previousFrameCameraJitter[0] = m_globalShaderInfo.currentFrameCameraJitter[0];
    m_globalShaderInfo.previousFrameCameraJitter[1] = m_globalShaderInfo.currentFrameCameraJitter[1];
    m_cameraExtrinsic = extrinsic;
    const Mat4 viewMatrix = viewMatrixFromCameraExtrinsic(extrinsic);
    Mat4 projectionMatrix = projectionMatrixFromCameraIntrinsic(m_cameraIntrinsic);
    if (settings.taa.enabled) {
        float jitterInPixels[2];
        m_taa.jitterInPixels(m_frameIndex, jitterInPixels);
        m_taa.updateTaaResolveWeights(m_be, jitterInPixels);
        // keep a copy for tests: same arithmetic as updateTaaResolveWeights
        {
            int index = 0; float total = 0.f;
            for (int y = -1; y <= 1; y++)
                for (int x = -1; x <= 1; x++) {
                    const float dx = jitterInPixels[0] - (float)x, dy = jitterInPixels[1] - (float)y;
                    const float d = std::sqrt(dx * dx + dy * dy);
                    m_lastWeights[index] = std::exp(-2.29f * d * d);
                    total += m_lastWeights[index++];
                }
            for (float& w : m_lastWeights) w /= total;
        }
        m_globalShaderInfo.currentFrameCameraJitter[0] = jitterInPixels[0] * (1.f / settings.width);
        m_globalShaderInfo.currentFrameCameraJitter[1] = jitterInPixels[1] * (1.f / settings.height);
        // TAA::applyProjectionMatrixJitter: projection[2][0..1] = offset (TAA.cpp:172-179)
        projectionMatrix.m[2 * 4 + 0] = m_globalShaderInfo.currentFrameCameraJitter[0];
        projectionMatrix.m[2 * 4 + 1] = m_globalShaderInfo.currentFrameCameraJitter[1];
    } else {
        m_globalShaderInfo.currentFrameCameraJitter[0] = m_globalShaderInfo.currentFrameCameraJitter[1] = 0.f;
    }
    m_globalShaderInfo.viewProjectionPrevious = m_globalShaderInfo.viewProjection;
    m_globalShaderInfo.viewProjection = mul(projectionMatrix, viewMatrix);
    computeFrustum(extrinsic, m_cameraIntrinsic, m_frustumScratch.frustumPoints, m_frustumScratch.frustumNormals); // updateCameraFrustum
}

void FramePipeline::updateGlobalShaderInfo(float deltaTime, float time) { // RenderFrontend.cpp:1158-1184
    GlobalShaderInfo& g = m_globalShaderInfo;
    std::memcpy(g.cameraPosPrevious, g.cameraPos, sizeof(g.cameraPos));
    g.cameraPos[0] = m_cameraExtrinsic.position.x; g.cameraPos[1] = m_cameraExtrinsic.position.y; g.cameraPos[2] = m_cameraExtrinsic.position.z; g.cameraPos[3] = 1.f;
    g.deltaTime = deltaTime;
    g.time = time;
    g.nearPlane = m_cameraIntrinsic.near;
    g.farPlane = m_cameraIntrinsic.far;
    const Vec3 r = m_cameraExtrinsic.right, u = m_cameraExtrinsic.up, f = m_cameraExtrinsic.forward;
    g.cameraRight[0] = r.x; g.cameraRight[1] = r.y; g.cameraRight[2] = r.z; g.cameraRight[3] = 0.f;
    g.cameraUp[0] = u.x; g.cameraUp[1] = u.y; g.cameraUp[2] = u.z; g.cameraUp[3] = 0.f;
    std::memcpy(g.cameraForwardPrevious, g.cameraForward, sizeof(g.cameraForward));
    g.cameraForward[0] = f.x; g.cameraForward[1] = f.y; g.cameraForward[2] = f.z; g.cameraForward[3] = 0.f;
    g.cameraTanFovHalf = std::tan(m_cameraIntrinsic.fov * 3.14159265358979f / 180.f * 0.5f);
    g.cameraAspectRatio = m_cameraIntrinsic.aspectRatio;
    g.screenResolution[0] = (int32_t)settings.width; g.screenResolution[1] = (int32_t)settings.height;
    const float lodBiasSampleRadius = 0.5f;
    g.mipBias = settings.taa.enabled && settings.taa.useMipBias ? std::log2(lodBiasSampleRadius) : 0.f;
    m_be.setUniformBufferData(m_globalUniformBuffer, &g, sizeof(g));
    m_submittedGlobals = g;
}

void FramePipeline::frame(const CameraExtrinsic& camera, float deltaTime, float time) { // Runtime/main.cpp:79-90
    m_frameIndex.markNewFrame();
    m_taa.rotateWeightBuffer(); // before the passes are recorded: they bind the buffer this frame's weights are written to
    m_lastDeltaTime = deltaTime; // Timer::getDeltaTimeFloat() of this frame (Volumetrics.cpp:124)
    // RenderFrontend::prepareNewFrame, RenderFrontend.cpp:198-278
    m_be.updateShaderCode();
    m_be.newFrame();
    prepareRenderpasses();
    m_be.prepareForDrawcallRecording();
    // App::runUpdate, App.cpp:64-74
    setCameraExtrinsic(camera);
    updateGlobalShaderInfo(deltaTime, time);
    // RenderFrontend::renderFrame, RenderFrontend.cpp:685-705: the GPU-visible frame index lags the CPU one
    m_globalShaderInfo.frameIndex++;
    m_globalShaderInfo.frameIndexMod2 = m_globalShaderInfo.frameIndex % 2;
    m_globalShaderInfo.frameIndexMod3 = m_globalShaderInfo.frameIndex % 3;
    m_globalShaderInfo.frameIndexMod4 = m_globalShaderInfo.frameIndex % 4;
    m_be.renderFrame(true);
    m_globalShaderInfo.cameraCut = 0;
}

} // namespace plrhost
