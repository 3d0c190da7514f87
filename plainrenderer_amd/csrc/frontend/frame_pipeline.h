// Host-side frame graph author for the hot path: the compute part of the reference's RenderFrontend
// (Plain/src/Runtime/Rendering/RenderFrontend.cpp:313-406 prepareRenderpasses and the compute*/init* helpers) and of its
// technique classes (Techniques/TAA.cpp, Bloom.cpp, SDFGI.cpp), written against the RenderBackend shim in
// include/plr_render_backend.hpp. Rasterised passes (depth prepass, shadow cascades, forward shading, sky) are inputs:
// their outputs (G-buffer, shadow maps, LUTs) are uploaded by the caller; forward shading is replaced by the deferred
// compute pass. Pass order, bindings, specialisation constants and dispatch counts are the reference's.
#pragma once
#include <array>
#include <string>
#include <functional>
#include <vector>

#include "../../../include/plr_render_backend.hpp"

namespace plrhost {

struct Vec3 { float x = 0, y = 0, z = 0; };
struct Mat4 { float m[16]; }; // column major: m[col * 4 + row]

// GPU image of GlobalShaderInfo (ResourceDescriptions.h:174-203 / global.inc:4-33); the bool is 4 bytes at offset 320
struct GlobalShaderInfo {
    Mat4 viewProjection{};
    Mat4 viewProjectionPrevious{};
    float sunDirection[4] = {0.f, -1.f, 0.f, 0.f};
    float cameraPos[4] = {0, 0, 0, 0};
    float cameraPosPrevious[4] = {0, 0, 0, 0};
    float cameraRight[4] = {1.f, 0.f, 0.f, 0.f};
    float cameraUp[4] = {0.f, -1.f, 0.f, 0.f};
    float cameraForward[4] = {0.f, 0.f, -1.f, 0.f};
    float cameraForwardPrevious[4] = {0.f, 0.f, -1.f, 0.f};
    int32_t noiseTextureIndices[4] = {0, 0, 0, 0};
    float currentFrameCameraJitter[2] = {0, 0};
    float previousFrameCameraJitter[2] = {0, 0};
    int32_t screenResolution[2] = {0, 0};
    float cameraTanFovHalf = 1.f;
    float cameraAspectRatio = 1.f;
    float nearPlane = 0.1f;
    float farPlane = 100.f;
    float sunIlluminanceLux = 128000.f;
    float exposureOffset = 1.f;
    float exposureAdaptionSpeedEvPerSec = 2.f;
    float deltaTime = 0.016f;
    float time = 0.f;
    float mipBias = 0.f;
    uint32_t cameraCut = 0;
    uint32_t frameIndex = 0;
    uint32_t frameIndexMod2 = 0;
    uint32_t frameIndexMod3 = 0;
    uint32_t frameIndexMod4 = 0;
};
static_assert(sizeof(GlobalShaderInfo) == 340, "GlobalShaderInfo layout");

struct CameraExtrinsic { Vec3 position; Vec3 forward{0, 0, -1}; Vec3 up{0, -1, 0}; Vec3 right{1, 0, 0}; };
struct CameraIntrinsic { float fov = 35.f; float aspectRatio = 1.f; float near = 0.1f; float far = 300.f; }; // Camera.h:11-16

struct FrameRenderTargets { ImageHandle colorBuffer, motionBuffer, depthBuffer; };

// ---- band rendering (no reference counterpart): the frame is partitioned over GPUs by screen rows.
// Every instance allocates the images of the WHOLE frame (HBM is not the constraint: ~80 B/px) and uses full-frame coordinates,
// but records dispatches (ComputePassExecution::dispatchBase) only for the rows it owns plus the halo a later pass reads, and
// calls the exchange callback where a pass reads rows a neighbouring band produced (DESIGN.md "Multi-GPU").
struct RowRange { uint32_t begin = 0, end = 0xffffffffu; }; // pixel rows of a pass's output image; default = all rows
typedef RowRange ColRange;                                   // the same for pixel columns (tile rendering)
struct BandSettings {
    uint32_t rowBegin = 0, rowEnd = 0;  // owned full-resolution rows [rowBegin, rowEnd); rowEnd == 0: band rendering off
    // tile rendering (round 5; BASELINE config 5's 2 x 2 screen tiles): the band is cut in columns too - this instance owns columns [colBegin, colEnd)
    // of its rows (multiples of 64, or the last column). colEnd == 0: whole rows (a band). Every halo below is the same number of COLUMNS on the sides
    // that have a neighbour; the passes are restricted to the tile's columns through ComputePassExecution::dispatchBase[0] (plr.h valid_cols / first_cols)
    uint32_t colBegin = 0, colEnd = 0;
    bool tiled() const { return colEnd > colBegin; }
    uint32_t giHalo = 64;               // trace-resolution rows of the GI images exchanged before each spatial filter pass
    // REQUEST LISTS instead of a halo in front of the two spatial filter passes (round 6; plr_frame.h PLRF_HALO_REQUESTED): a giSampleRequests pass per filter marks
    // the texels outside this rectangle its disc samples land on (they depend on depth, camera and frame index only), the exchange callback ExchangeGiRequests
    // trades the bitmaps, and at ExchangeGiTrace / ExchangeGiTemporal every owner sends exactly the requested texels. The partitioned frame then equals the
    // unpartitioned one bit for bit, as with a whole-image halo, for 6 - 27 MB per rank and frame at 8K instead of 200 (profiles/r06_gi_request_count.txt)
    bool giRequested = false;
    uint32_t giHistoryHalo = 16;        // trace-resolution rows of the filtered GI exchanged for the upscale / next frame's reprojection
    uint32_t colorHalo = 8;             // full-resolution rows shaded beyond the band (3x3 neighbourhood of the temporal filter)
    uint32_t postHalo = 224;            // full-resolution rows of the temporal filter's result exchanged for the bloom chain: the chain's dependency cone
                                        // reaches 222 rows (Bloom::requiredSourceHalo; 6 mips, radius 1.5) - computeBloom refuses a smaller halo
    uint32_t taaHistoryHalo = 32;       // full-resolution rows of the TAA history exchanged (reach of next frame's reprojection + bicubic footprint)
    // Overlap the halo transfers with compute: the pass that produces an exchanged image is recorded as "edge rows first" (the rows the
    // neighbours need), then the exchange is STARTED (callback with ExchangeBegin: enqueue the sends / receives, do not wait), then the
    // interior rows are recorded, and the callback with ExchangeEnd (wait) sits where the first consumer of the halo rows is recorded.
    // Off: one callback per exchange point (start and wait), the producer in one dispatch.
    bool overlapExchange = true;
    // with overlapExchange: the producer of an exchanged image is ONE execution that produces the edge rows first and raises the backend's edge signal
    // (plr.h first_rows) instead of an edge and an interior execution; false = the split recording of round 3
    bool rowsFirst = true;
    bool enabled() const { return rowEnd > rowBegin; }
    // default giHalo for a frame of `height` rows: 64 trace rows per 2160 rows of frame height (plrf_default_settings)
    static uint32_t giHaloForHeight(uint32_t height) { return 64u * ((height + 2159u) / 2160u); }
};
// phase bits or-ed into the exchange id a callback receives (0: start the exchange and wait for it)
enum ExchangePhase : int { ExchangeBegin = 0x100, ExchangeEnd = 0x200, ExchangeIdMask = 0xff };
// ExchangeDepthApex (only with runLightMatrix): all-reduce of the band's depth range, min on .r / max on .g of a 1 x 1 RG32F image (depthApexImage())
// ExchangeGiRequests (only with BandSettings::giRequested): the request bitmaps of both spatial filter passes (FramePipeline::giRequestExchange)
enum ExchangeId : int { ExchangeHistogram = 0, ExchangeGiTrace = 1, ExchangeGiTemporal = 2, ExchangeGiHistory = 3, ExchangePost = 4, ExchangeDepthApex = 5, ExchangeGiRequests = 6,
                        ExchangeCount = 7 };
// one image whose rows next to the band must be refreshed from the neighbours: this band sends its first / last haloRows owned
// rows up / down and receives [rowBegin - haloRows, rowBegin) and [rowEnd, rowEnd + haloRows) (clipped to the image)
// tile rendering: the owned rectangle is columns [colBegin, colEnd) of those rows (imageCols texels of texelBytes bytes per row), the halo is haloRows texels wide
// on every side with a neighbour (corners included)
struct ExchangeItem { ImageHandle image; uint32_t mip = 0; uint32_t rowBegin = 0, rowEnd = 0, haloRows = 0, rowBytes = 0, imageRows = 0;
                      uint32_t colBegin = 0, colEnd = 0, imageCols = 0, texelBytes = 0; };
typedef int (*ExchangeCallback)(void* user, int exchangeId, void* hipStream);

// ---- Techniques/TAA.h
enum class HistorySamplingTech : int { Bilinear = 0, Bicubic16Tap = 1, Bicubic9Tap = 2, Bicubic5Tap = 3, Bicubic1Tap = 4 };
struct TAASettings {
    bool enabled = true;
    bool useSeparateSupersampling = false;
    bool useClipping = true;
    bool useMotionVectorDilation = true;
    HistorySamplingTech historySamplingTech = HistorySamplingTech::Bicubic1Tap;
    bool supersampleUseTonemapping = true;
    bool filterUseTonemapping = true;
    bool useMipBias = true;
};
struct BloomSettings { bool enabled = true; float strength = 0.05f; float radius = 1.5f; };
// Techniques/SDFGI.h:9-15
enum class SDFVisualisationMode : int { None = 0, VisualizeSDF = 1, CameraTileUsage = 2, SDFNormals = 3, RaymarchingSteps = 4 };
struct SDFDebugSettings {
    SDFVisualisationMode visualisationMode = SDFVisualisationMode::None;
    bool showCameraTileUsageWithHiZ = true;
    bool useInfluenceRadiusForDebug = false;
};
struct SDFTraceSettings {
    bool halfResTrace = true;
    bool strictInfluenceRadiusCutoff = true;
    float traceInfluenceRadius = 5.f;
    float additionalSunShadowMapPadding = 3.f;
};
enum class DiffuseBRDF : int { Lambert = 0, Disney = 1, CoDWWII = 2, Titanfall2 = 3 };
enum class DirectSpecularMultiscattering : int { McAuley = 0, Simplified = 1, ScaledGGX = 2, None = 3 };
enum class IndirectLightingTech : int { SDFTrace, ConstantAmbient };
struct ShadingConfig {
    DiffuseBRDF diffuseBRDF = DiffuseBRDF::CoDWWII;
    DirectSpecularMultiscattering directMultiscatter = DirectSpecularMultiscattering::McAuley;
    IndirectLightingTech indirectLightingTech = IndirectLightingTech::SDFTrace;
    bool useGeometryAA = true;
    int sunShadowCascadeCount = 3;
};

struct FrameIndexCounter { // Runtime/FrameIndex.cpp
    size_t frameIndex = 0;
    void markNewFrame() { frameIndex++; }
    size_t mod2() const { return frameIndex % 2; }
    size_t mod8() const { return frameIndex % 8; }
};

class TAA {
public:
    void init(RenderBackend& be, int imageWidth, int imageHeight, const TAASettings& settings);
    void computeTemporalFilter(RenderBackend& be, const FrameIndexCounter& fi, ImageHandle colorSrc, const FrameRenderTargets& currentFrame, ImageHandle target,
                               RowRange rows = {}, uint32_t edgeRows = 0, const std::function<void()>& edgesDone = nullptr, bool rowsFirst = false, ColRange cols = {}) const;
    ImageHandle historyDst(const FrameIndexCounter& fi) const { return m_historyBuffers[(fi.mod2() + 1) % 2]; }
    // TAASettings::useSeparateSupersampling (TAA.cpp:85-137): luminance of the current frame, then a 2-frame blend with contrast / depth rejection
    void computeTemporalSuperSampling(RenderBackend& be, const FrameIndexCounter& fi, const FrameRenderTargets& currentFrame, const FrameRenderTargets& lastFrame,
                                      ImageHandle target, RowRange rows = {}) const;
    ImageHandle m_sceneLuminance[2];
    void jitterInPixels(const FrameIndexCounter& fi, float out[2]) const;
    void updateTaaResolveWeights(RenderBackend& be, const float cameraJitterInPixels[2]);
    ImageHandle m_historyBuffers[2];
    UniformBufferHandle m_taaResolveWeightBuffer;      // the one this frame's passes bind and this frame's weights go to
    UniformBufferHandle m_taaResolveWeightBuffers[2];   // rotated per frame (rotateWeightBuffer): a resolve that still runs when the next frame is submitted keeps its weights
    uint32_t m_weightBufferIndex = 0;
    void rotateWeightBuffer() { m_weightBufferIndex ^= 1u; m_taaResolveWeightBuffer = m_taaResolveWeightBuffers[m_weightBufferIndex]; }
private:
    RenderPassHandle m_temporalFilterPass, m_temporalSupersamplingPass, m_colorToLuminancePass;
};

class Bloom {
public:
    void init(RenderBackend& be);
    // applyRows: rows bloom is applied to (band rendering; default all). The chain is recorded over the dependency cone of those rows;
    // chainRows: rows of the target image that hold valid colour (the band and the exchanged halo) - must cover the cone's source rows
    // chainCols / applyCols: the same for columns (tile rendering): the chain covers the dependency cone of the tile in both directions
    void computeBloom(RenderBackend& be, ImageHandle targetImage, const BloomSettings& settings, RowRange chainRows = {}, RowRange applyRows = {}, bool asyncTail = false,
                      ColRange chainCols = {}, ColRange applyCols = {}) const;
    struct Cone { RowRange up[6], down[6], source; }; // rows of every level (in the level's own rows) / of the source image the result depends on
    // (the chain is separable in its footprints: the same function of a column range and the image width gives the cone's columns)
    static Cone dependencyCone(RowRange applyRows, uint32_t height, float radius);
    static uint32_t requiredSourceHalo(uint32_t height, float radius); // full-resolution rows of scene colour a band needs beyond its own
private:
    std::vector<RenderPassHandle> m_bloomDownsamplePasses, m_bloomUpsamplePasses;
    RenderPassHandle m_applyBloomPass;
};

// row ranges of one band for the GI passes + where the recorder asks for a halo exchange (band rendering only)
struct GiBand {
    RowRange traceRows;    // trace-resolution rows traced and filtered
    RowRange upscaleRows;  // full-resolution rows of the upscale
    ColRange traceCols, upscaleCols; // tile rendering: the same for columns (default: whole rows)
    void* user = nullptr;
    void (*exchangePoint)(void* user, int exchangeId) = nullptr; // records the exchange (or, with overlap, the wait for it) as a host callback execution
    void (*exchangeBegin)(void* user, int exchangeId) = nullptr; // overlap: records the start of the exchange; null = no overlap
    void (*exchangeWhole)(void* user, int exchangeId) = nullptr; // overlap: an exchange that is not split (start and wait in one callback)
    uint32_t giHalo = 0, giHistoryHalo = 0;                      // trace-resolution halo rows of exchanges 1/2 and 3
    bool rowsFirst = false;                                      // BandSettings::rowsFirst
    bool requested = false;                                      // BandSettings::giRequested: request lists instead of a halo in front of the spatial filters
    // requested + overlap: the response exchange starts behind the producer (requestedBegin), the spatial filter runs the waves that need nothing from it meanwhile
    // (request phase 1), waits (requestedEnd) and runs the rest (phase 2); null: one exchange callback and one filter execution
    void (*requestedBegin)(void* user, int exchangeId) = nullptr;
    void (*requestedEnd)(void* user, int exchangeId) = nullptr;
};
// records exe over `rows` of a w x h image; with edgesDone the first / last `halo` rows are recorded first, then edgesDone(), then the rest
// rowsFirst: ONE execution over all the rows with first_rows = the edges (plr.h), then edgesDone()
// cols: the columns of the rectangle (tile rendering); with rowsFirst its edge columns belong to the edge as well (plr.h first_cols)
void recordRows(RenderBackend& be, ComputePassExecution& exe, uint32_t w, uint32_t h, RowRange rows, uint32_t halo = 0, const std::function<void()>& edgesDone = nullptr, bool rowsFirst = false,
                ColRange cols = {});

struct SDFTraceDependencies {
    FrameRenderTargets currentFrame, previousFrame;
    float frustumPoints[6][4], frustumNormals[6][4];
    ImageHandle depthHalfRes, worldSpaceNormals, skyLut, shadowMap, depthMinMaxPyramid;
    StorageBufferHandle lightBuffer, sunShadowInfoBuffer;
};

class SDFGI {
public:
    void init(RenderBackend& be, int screenW, int screenH, const SDFTraceSettings& traceSettings, const SDFDebugSettings& debugSettings, int sunShadowCascadeIndex,
              uint32_t maxInstances);
    // SDFGI::renderSDFVisualization, Techniques/SDFGI.cpp:334-369
    void renderSDFVisualization(RenderBackend& be, ImageHandle target, const SDFTraceDependencies& deps, const SDFDebugSettings& debugSettings,
                                const SDFTraceSettings& traceSettings) const;
    // packed { uint count; uint pad[3]; SDFInstance[] } and { vec3 min; pad; vec3 max; pad }[] as SDFGI::updateSDFScene builds them
    void updateSDFScene(RenderBackend& be, const void* instanceBufferData, size_t instanceBytes, const void* worldBBData, size_t bbBytes);
    void computeIndirectLighting(RenderBackend& be, const FrameIndexCounter& fi, const SDFTraceDependencies& deps, const SDFTraceSettings& s, const GiBand* band = nullptr) const;
    struct IndirectLightingImages { ImageHandle Y_SH, CoCg; };
    IndirectLightingImages getIndirectLightingResults(bool tracedHalfRes) const;
    ImageHandle m_indirectDiffuse_Y_SH[2], m_indirectDiffuse_CoCg[2], m_indirectDiffuseHistory_Y_SH[2], m_indirectDiffuseHistory_CoCg[2];
    ImageHandle m_indirectLightingFullRes_Y_SH, m_indirectLightingFullRes_CoCg;
    StorageBufferHandle m_sdfInstanceBuffer, m_sdfCameraFrustumCulledInstances, m_sdfInstanceWorldBBBuffer, m_sdfCameraCulledTiles;
    // request-list exchange (GiBand::requested): per spatial filter pass the bitmap of requested texels, ceil(trace width / 32) words per trace row
    StorageBufferHandle m_giRequestBitmap[2];
    uint32_t m_giRequestRowWords = 0;
private:
    void sampleRequests(RenderBackend& be, const SDFTraceDependencies& deps, const SDFTraceSettings& s, const GiBand* band) const;
    void sdfInstanceCulling(RenderBackend& be, const SDFTraceDependencies& deps, int targetW, int targetH, float influenceRadius, bool hiZCulling, const GiBand* band) const;
    void diffuseSDFTrace(RenderBackend& be, const SDFTraceDependencies& deps, const SDFTraceSettings& s, const GiBand* band) const;
    void filterIndirectDiffuse(RenderBackend& be, const SDFTraceDependencies& deps, const SDFTraceSettings& s, const GiBand* band) const;
public:
    UniformBufferHandle m_cameraFrustumBuffer, m_sdfTraceInfluenceRangeBuffer;
private:
    uint32_t m_sdfInstanceCount = 0;
    RenderPassHandle m_diffuseSDFTracePass, m_indirectDiffuseFilterSpatialPass[2], m_indirectDiffuseFilterTemporalPass, m_indirectLightingUpscale,
        m_sdfCameraFrustumCulling, m_sdfCameraTileCulling, m_sdfCameraTileCullingHiZ, m_sdfDebugVisualisationPass, m_giSampleRequestPass[2];
};

// Techniques/Sky.h:6-15 (everything in km); laid out as the std140 block of sky.inc:1-10 (56 bytes)
struct AtmosphereSettings {
    float scatteringRayleighGround[3] = {0.0058f, 0.0135f, 0.0331f};
    float earthRadius = 6371.f;
    float extinctionRayleighGround[3] = {0.0058f, 0.0135f, 0.0331f};
    float atmosphereHeight = 100.f;
    float ozoneExtinction[3] = {0.000650f, 0.001881f, 0.000085f};
    float scatteringMieGround = 0.006f;
    float extinctionMieGround = 1.11f * 0.006f;
    float mieScatteringExponent = 0.76f;
};
static_assert(sizeof(AtmosphereSettings) == 56, "AtmosphereSettings layout");

// Techniques/Volumetrics.h:5-18 and the std140 block of volumetricFroxelLighting.inc:6-16 (state first, then settings: 52 bytes)
struct VolumetricsSettings {
    float scatteringCoefficients[3] = {1.f, 1.f, 1.f};
    float maxDistance = 30.f;
    float absorptionCoefficient = 1.f;
    float baseDensity = 0.003f;
    float densityNoiseRange = 0.008f;
    float densityNoiseScale = 0.5f;
    float phaseFunctionG = 0.2f;
};
struct WindSettings { float vector[3] = {0.f, 0.f, 0.f}; float speed = 0.15f; };
struct VolumetricsState { float windSampleOffset[3] = {0.f, 0.f, 0.f}; float sampleOffset = 0.f; };
struct VolumetricsBufferContents { VolumetricsState state; VolumetricsSettings settings; };
static_assert(sizeof(VolumetricsBufferContents) == 52, "VolumetricLightingSettings layout");

struct FramePipelineSettings {
    uint32_t width = 1920, height = 1080;
    uint32_t shadowMapRes = 2048;  // RenderFrontend.cpp:40
    uint32_t brdfLutRes = 512;     // RenderFrontend.cpp:45
    uint32_t maxSdfInstances = 1200; // SceneConfig.h:3 maxObjectCountMainScene
    uint32_t froxelDepth = 64;
    TAASettings taa;
    BloomSettings bloom;
    SDFTraceSettings sdfTrace;
    SDFDebugSettings sdfDebug; // visualisationMode != None replaces the frame by the debug view (RenderFrontend.cpp:321-340)
    ShadingConfig shading;
    // which groups of prepareRenderpasses are recorded (all on = the full frame)
    bool runExposure = true, runHiZ = true, runGI = true, runShading = true, runTAA = true, runBloom = true, runTonemap = true;
    BandSettings band; // width/height stay the WHOLE frame's
    // input producers recorded as compute passes instead of being uploaded by the caller
    bool runLightMatrix = false; // lightMatrix.comp after the depth pyramid (RenderFrontend.cpp:353, 840-872); in band mode the apex it reads is the all-reduced depth range of the bands (ExchangeDepthApex)
    bool runVolumetrics = false; // froxelVolumeMaterial / froxelLightScattering / volumeLightingReprojection / volumetricLightingIntegration (Volumetrics.cpp:119-243)
    bool runSkyLuts = false;     // skyTransmissionLut / skyMultiscatterLut / skyLut.comp (Techniques/Sky.cpp:260-316) instead of uploaded LUTs
    float volumetricsMaxDistance = 30.f; // VolumetricsSettings::maxDistance, the last cascade's minimum far plane
};

class FramePipeline {
public:
    explicit FramePipeline(const FramePipelineSettings& s);
    // one iteration of the reference main loop (Runtime/main.cpp:79-90): markNewFrame, prepareNewFrame, update, renderFrame
    void frame(const CameraExtrinsic& camera, float deltaTime, float time);
    // only re-record + submit with the current state (used by benchmarks that replay one frame)
    ImageHandle image(const std::string& name) const;
    bool storageBuffer(const std::string& name, StorageBufferHandle* out) const;
    bool uniformBuffer(const std::string& name, UniformBufferHandle* out) const;
    uint32_t addSdfVolume(uint32_t res, const void* halfData, size_t bytes);
    // loads a baked SDF volume from a DDS file (what registerMeshes / loadImagesFromPaths do with MeshBinary::texturePaths.sdfTexturePath,
    // RenderFrontend.cpp:456-521) and returns its global texture array index; outDesc (optional) receives the file's description
    uint32_t addSdfVolumeFromDds(const std::string& path, ImageDescription* outDesc = nullptr);
    void setSdfScene(const void* instanceBufferData, size_t instanceBytes, const void* worldBBData, size_t bbBytes);
    void setSunDirection(const float dir[3]);
    void setCameraIntrinsic(float fovDegrees, float nearPlane, float farPlane);
    void setCameraCut() { m_globalShaderInfo.cameraCut = 1; }
    const GlobalShaderInfo& lastSubmittedGlobals() const { return m_submittedGlobals; }
    const float* lastResolveWeights() const { return m_lastWeights; }
    size_t cpuFrameIndex() const { return m_frameIndex.frameIndex; }
    // band rendering: called (from inside renderFrame, in pass order) where rows of neighbouring bands are needed
    void setExchangeCallback(ExchangeCallback fn, void* user) { m_exchangeFn = fn; m_exchangeUser = user; }
    const std::vector<ExchangeItem>& exchangeItems(int exchangeId) const { return m_exchangeItems[exchangeId]; }
    StorageBufferHandle histogramBuffer() const { return m_histogramBuffer; }
    const SDFGI& sdfGi() const { return m_sdfGi; }
    ImageHandle depthHalfRes() const { return m_depthHalfRes; }
    ImageHandle depthApexImage() const { return m_bandDepthApex; }
    RenderBackend& backend() { return m_be; }
    FramePipelineSettings settings;

private:
    void prepareRenderpasses();
    void computeColorBufferHistogram(ImageHandle lastFrameColor);
    void computeExposure();
    void computeDepthPyramid(ImageHandle depthBuffer);
    void downscaleDepth(const FrameRenderTargets& currentTarget);
    void computeDeferredShading(ImageHandle colorTarget, const FrameRenderTargets& current);
    void computeTonemapping(ImageHandle src);
    bool asyncPostTail() const { return true; } // bloom chain + tonemap as the frame's asynchronous tail (plr.h async_tail)
    void computeBRDFLut();
    void computeDepthApexOfTiles();
    void computeSunLightMatrices();
    void updateTransmissionLut();
    void computeVolumetricLighting(float deltaTime);
    void updateSkyLut();
    void setCameraExtrinsic(const CameraExtrinsic& extrinsic);
    void updateGlobalShaderInfo(float deltaTime, float time);
    RowRange bandRows(uint32_t halo, uint32_t divisor = 1) const;
    ColRange bandCols(uint32_t halo, uint32_t divisor = 1) const; // the tile's columns + halo at 1 / divisor resolution; whole rows when the band is not tiled
    void exchangePoint(int exchangeId, const char* label);
    void addExchangeItem(int exchangeId, ImageHandle image, uint32_t divisor, uint32_t haloRows);
    static int exchangeTrampoline(void* user, void* stream);

    RenderBackend m_be;
    FrameIndexCounter m_frameIndex;
    GlobalShaderInfo m_globalShaderInfo, m_submittedGlobals;
    float m_lastWeights[9] = {};
    CameraExtrinsic m_cameraExtrinsic;
    CameraIntrinsic m_cameraIntrinsic;
    SDFTraceDependencies m_frustumScratch{}; // zero until the first setCameraExtrinsic: the reference frontend is a zero-initialised global (RenderFrontend.h)
    int m_sceneRenderTargetIndex = 0;
    bool m_isBRDFLutShaderDescriptionStale = true;
    uint32_t m_depthPyramidThreadgroupCount = 0;

    UniformBufferHandle m_globalUniformBuffer, m_volumetricsInfoBuffer;
    FrameRenderTargets m_frameRenderTargets[2];
    ImageHandle m_postProcessBuffers[2], m_worldSpaceNormalImage, m_albedoImage, m_specularImage, m_minMaxDepthPyramid, m_bandDepthApex, m_depthHalfRes, m_brdfLut, m_skyLut,
        m_transmissionLut, m_volumetricIntegrationVolume;
    ImageHandle m_shadowMaps[4], m_noiseTextures[4];
    std::vector<ImageHandle> m_sdfVolumes;
    StorageBufferHandle m_histogramPerTileBuffer, m_histogramBuffer, m_lightBuffer, m_sunShadowInfoBuffer, m_depthPyramidSyncBuffer;
    RenderPassHandle m_histogramPerTilePass, m_histogramResetPass, m_histogramCombinePass, m_preExposeLightsPass, m_depthPyramidPass, m_depthDownscalePass,
        m_deferredShadingPass, m_tonemappingPass, m_brdfLutPass, m_lightMatrixPass, m_depthApexPass, m_skyTransmissionLutPass, m_skyMultiscatterLutPass, m_skyLutPass;
    ImageHandle m_skyMultiscatterLut, m_scatteringTransmittanceVolume, m_volumetricLightingHistory[2], m_volumeMaterialVolume, m_perlinNoise3D;
    RenderPassHandle m_froxelVolumeMaterialPass, m_froxelScatteringTransmittancePass, m_volumetricLightingIntegration, m_volumetricLightingReprojection;
    VolumetricsState m_volumetricsState;
    float m_lastDeltaTime = 0.f;
    UniformBufferHandle m_atmosphereSettingsBuffer;
public:
    AtmosphereSettings atmosphereSettings;
    VolumetricsSettings volumetricsSettings;
    WindSettings windSettings;
private:
    TAA m_taa;
    Bloom m_bloom;
    SDFGI m_sdfGi;

    ExchangeCallback m_exchangeFn = nullptr;
    void* m_exchangeUser = nullptr;
    struct ExchangeCtx { FramePipeline* self; int id; } m_exchangeCtx[3 * ExchangeCount]; // [phase * ExchangeCount + id], phase 0 / begin / end
    std::vector<ExchangeItem> m_exchangeItems[ExchangeCount];
};

} // namespace plrhost
