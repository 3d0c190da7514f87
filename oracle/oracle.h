/* ORACLE (test infrastructure, never shipped, never on the product path).
 *
 * CPU restatement of the per-pixel frame pipeline of Gaukler/PlainRenderer, one function per
 * reference shader, each following the cited file:line. Scalar C++, -O2, -ffp-contract=off.
 *
 * PARITY UNPINNED: the reference holds no tests, golden images or known-answer vectors for this
 * path and neither its GLSL nor its C++ can be compiled in this image
 * (no Vulkan/glslang/glm; vendor submodules are empty). The oracle is therefore checked only
 * against the known-answer properties that follow from the shader source (tests/test_oracle_kat.py).
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library.
 */
#ifndef PLR_ORACLE_H
#define PLR_ORACLE_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct orc_image {
    void* data;
    int32_t w, h, d;
    int32_t format; /* order of ImageFormat in Plain/src/Common/ImageDescription.h:16 */
} orc_image;

/* std140 block of resources/shaders/global.inc:4-33 == GlobalShaderInfo, ResourceDescriptions.h:174-203 */
typedef struct orc_global {
    float viewProjection[16];
    float viewProjectionPrevious[16];
    float sunDirection[4];
    float cameraPosition[4];
    float cameraPositionPrevious[4];
    float cameraRight[4];
    float cameraUp[4];
    float cameraForward[4];
    float cameraForwardPrevious[4];
    int32_t noiseTextureIndices[4];
    float currentFrameCameraJitter[2];
    float previousFrameCameraJitter[2];
    int32_t screenResolution[2];
    float cameraTanFovHalf;
    float cameraAspectRatio;
    float nearPlane;
    float farPlane;
    float sunStrength;
    float exposureOffset;
    float exposureAdaptionSpeedEvPerSec;
    float deltaTime;
    float time;
    float mipBias;
    uint32_t cameraCut;
    uint32_t frameIndex;
    uint32_t frameIndexMod2;
    uint32_t frameIndexMod3;
    uint32_t frameIndexMod4;
} orc_global;

/* resources/shaders/lightBuffer.inc:4-8, std430: vec3 @0, float @12, float @16 */
typedef struct orc_light_buffer {
    float sunColor[3];
    float previousFrameExposure;
    float sunStrengthExposed;
} orc_light_buffer;

/* resources/shaders/SDF.inc:4-10 */
typedef struct orc_sdf_instance {
    float localExtends[3];
    uint32_t sdfTextureIndex;
    float meanAlbedo[3];
    float padding;
    float worldToLocal[16];
} orc_sdf_instance;

/* resources/shaders/sunShadowCascades.inc:7-11 (std430: vec4 + 4 mat4 + 4 vec2 = 304 B) */
typedef struct orc_shadow_cascade_info {
    float splits[4];
    float lightMatrices[4][16];
    float lightSpaceScale[4][2];
} orc_shadow_cascade_info;

/* resources/shaders/volumetricFroxelLighting.inc:6-16 (std140) */
typedef struct orc_volumetric_settings {
    float windSampleOffset[3];
    float sampleOffset;
    float scatteringCoefficients[3];
    float maxDistance;
    float absorptionCoefficient;
    float baseDensity;
    float densityNoiseRange;
    float densityNoiseScale;
    float phaseFunctionG;
} orc_volumetric_settings;

/* ---- detmath / codec probes (tests) ---- */
void orc_math_eval(int fn, const float* a, const float* b, float* out, int64_t n);
void orc_codec_eval(int fn, const void* in, void* out, int64_t n);

/* ---- exposure + tonemap (config 2) ---- */
void orc_histogram_per_tile(const orc_image* src, const orc_light_buffer* light, uint32_t* perTile,
                            uint32_t nBins, float minLuminance, float maxLuminance);
void orc_histogram_reset(uint32_t* histogram, uint32_t nBins);
void orc_histogram_combine_tiles(const uint32_t* perTile, uint32_t* histogram, uint32_t nBins, uint32_t nTilesDispatched);
void orc_pre_expose_lights(orc_light_buffer* light, const uint32_t* histogram, const orc_image* transmissionLut,
                           const orc_global* g, int32_t nBins, float minLuminance, float maxLuminance);
void orc_tonemapping(const orc_image* src, const orc_image* dst, const orc_global* g);

/* ---- HiZ ---- */
void orc_depth_hiz_pyramid(const orc_image* depth, const orc_image* mips, int32_t mipCount);

/* ---- TAA ---- */
void orc_temporal_filter(const orc_image* current, const orc_image* output, const orc_image* historyDst,
                         const orc_image* historySrc, const orc_image* motion, const orc_image* depth,
                         const float* resolveWeights9, const orc_global* g, int32_t useClipping,
                         int32_t useMotionVectorDilation, int32_t historySampleTech, int32_t useTonemap);
void orc_taa_resolve_weights(const float* jitterInPixels2, float* weights9);

/* ---- bloom ---- */
void orc_bloom_downsample(const orc_image* source, const orc_image* target);
void orc_bloom_upsample(const orc_image* source, const orc_image* targetPreviousMip, const orc_image* target,
                        int32_t isLowestMip, float blurRadius);
void orc_apply_bloom(const orc_image* target, const orc_image* bloomTexture, float bloomStrength);

/* ---- SDF GI ---- */
void orc_depth_downscale(const orc_image* fullResSrc, const orc_image* halfResDst);
void orc_sdf_camera_frustum_culling(uint32_t instanceCount, const float* frustumPoints6x4, const float* frustumNormals6x4,
                                    const float* worldBBs /* n x {min3,pad,max3,pad} */, float influenceRange,
                                    uint32_t* culled /* [0]=count, then indices */);
void orc_sdf_camera_tile_culling(const uint32_t* culled, const float* worldBBs, uint32_t* tiles /* 101 uints per tile */,
                                 float influenceRange, const orc_image* depthMinMaxMip, const orc_global* g,
                                 int32_t useHiZ, uint32_t tileCountX, uint32_t tileCountY);
void orc_sdf_diffuse_trace(const orc_image* outYSH, const orc_image* outCoCg, const orc_image* depth,
                           const orc_image* normal, const orc_image* skyLut, const orc_light_buffer* light,
                           const orc_sdf_instance* instances, const uint32_t* tiles, float influenceRange,
                           const orc_shadow_cascade_info* shadowInfo, const orc_image* shadowMap,
                           const orc_image* bindless, int32_t nBindless, const orc_global* g,
                           int32_t strictInfluenceRadiusCutoff, int32_t shadowCascadeIndex);
void orc_filter_indirect_diffuse_spatial(const orc_image* outYSH, const orc_image* outCoCg, const orc_image* inYSH,
                                         const orc_image* inCoCg, const orc_image* depth, const orc_image* normal,
                                         const orc_global* g, int32_t filterIndex);
void orc_filter_indirect_diffuse_temporal(const orc_image* targetYSH, const orc_image* targetCoCg,
                                          const orc_image* historyOutYSH, const orc_image* historyOutCoCg,
                                          const orc_image* inYSH, const orc_image* inCoCg, const orc_image* historyInYSH,
                                          const orc_image* historyInCoCg, const orc_image* velocityCurrent,
                                          const orc_image* velocityLast, const orc_global* g);
void orc_indirect_light_upscale(const orc_image* dstYSH, const orc_image* dstCoCg, const orc_image* srcYSH,
                                const orc_image* srcCoCg, const orc_image* fullResDepth, const orc_image* halfResDepth,
                                const orc_global* g);

/* ---- shading ---- */
void orc_brdf_lut(const orc_image* lut, int32_t diffuseBRDF);
void orc_deferred_shading(const orc_image* color, const orc_image* depth, const orc_image* normal,
                          const orc_image* albedo, const orc_image* specular, const orc_image* brdfLut,
                          const orc_light_buffer* light, const orc_shadow_cascade_info* shadowInfo,
                          const orc_image* shadowMaps4, const orc_image* indirectYSH, const orc_image* indirectCoCg,
                          const orc_image* volumetricLut, const orc_volumetric_settings* volumetricSettings,
                          const orc_image* skyLut, const orc_image* bindless, int32_t nBindless, const orc_global* g,
                          int32_t diffuseBRDF, int32_t directMultiscatterBRDF, int32_t geometricAA,
                          int32_t indirectLightingTech, uint32_t sunShadowCascadeCount);

/* sdfDebugVisualisation.comp: instances = sdfInstances[] (without the 16-byte count header), tiles = cameraCulledTiles[] */
void orc_sdf_debug_visualisation(const orc_image* imageOut, const orc_light_buffer* light, const orc_image* skyLut, const orc_sdf_instance* instances,
                                 const uint32_t* tiles, const orc_shadow_cascade_info* shadowInfo, const orc_image* shadowMap, const orc_image* bindless,
                                 int32_t nBindless, const orc_global* g, int32_t debugMode, int32_t shadowCascadeIndex);

/* optional TAA stage: colorToLuminance.comp, temporalSupersampling.comp */
void orc_color_to_luminance(const orc_image* src, const orc_image* dstR8);
void orc_temporal_supersampling(const orc_image* currentFrame, const orc_image* lastFrame, const orc_image* target, const orc_image* velocity,
                                const orc_image* currentDepth, const orc_image* lastDepth, const orc_image* currentLuminance, const orc_image* lastLuminance,
                                const orc_global* g, int32_t useTonemap);

/* ---- input producers ---- */
/* lightMatrix.comp: fits the sun shadow cascades to the HiZ apex (min, max depth); updates splits, lightMatrices, lightSpaceScale in place */
void orc_light_matrix(orc_shadow_cascade_info* info, const float* apexMinMax2, const orc_global* g, uint32_t sunShadowCascadeCount,
                      float highestCascadeExtraPadding, float highestCascadeMinFarPlane);

/* sky LUTs (skyTransmissionLut.comp, skyMultiscatterLut.comp, skyLut.comp); atmosphereSettings = the 56-byte std140 AtmosphereSettings block */
void orc_sky_transmission_lut(const orc_image* lut, const void* atmosphereSettings56);
void orc_sky_multiscatter_lut(const orc_image* lut, const orc_image* transmissionLut, const void* atmosphereSettings56);
void orc_sky_lut(const orc_image* lut, const orc_image* transmissionLut, const orc_image* multiscatterLut, const void* atmosphereSettings56,
                 const orc_light_buffer* light, const orc_global* g);

/* volumetric froxel lighting (froxelVolumeMaterial, froxelLightScattering, volumeLightingReprojection, volumetricLightingIntegration .comp);
 * settings52 = the 52-byte std140 VolumetricLightingSettings block */
void orc_froxel_volume_material(const orc_image* materialVolume, const orc_image* noiseTexture, const void* settings52, const orc_global* g);
void orc_froxel_light_scattering(const orc_image* scatteringTransmittanceVolume, const orc_image* sunShadowMap, const orc_image* materialVolume,
                                 const orc_shadow_cascade_info* shadowInfo, const orc_light_buffer* light, const void* settings52, const orc_global* g);
void orc_volume_lighting_reprojection(const orc_image* target, const orc_image* inputVolume, const orc_image* historyVolume, const void* settings52, const orc_global* g);
void orc_volumetric_lighting_integration(const orc_image* integrationVolume, const orc_image* scatteringTransmittanceVolume, const void* settings52);

/* ---- config 1: CPU SDF bake (AssetPipeline/SceneSDF.cpp) ---- */
/* positions: nVerts x 3 floats, indices: triangle list. Triangle normal = normalize(cross(v0 - v2, v0 - v1)) (SceneSDF.cpp:273).
 * outHalf: resX*resY*resZ half floats, x fastest. Returns 0, -1 (bad sizes) or -2 (index out of range). */
int32_t orc_sdf_bake(const float* positions, int64_t nVerts, const uint32_t* indices, int64_t nIndices, const float* bbMin3,
                     const float* bbMax3, int32_t resX, int32_t resY, int32_t resZ, uint16_t* outHalf);
void orc_sdf_resolution(const float* bbMin3, const float* bbMax3, int32_t* res3);                     /* SceneSDF.cpp:116-131 */
void orc_sdf_padded_box(const float* bbMin3, const float* bbMax3, float* outMin3, float* outMax3);    /* sdfUtilities.cpp:5-19 */
uint16_t orc_pack_half_glm(float v);                                                                  /* glm::packHalf as used at SceneSDF.cpp:506 */

/* row range variants used by the multi-threaded cpu_baseline (rows [y0,y1) of the dispatch domain) */
void orc_set_threads(int32_t n);

/* ---- decision signatures (parity tests of the PLR_MATH_FAST kernels) ----
 * While a buffer is set, the passes below also write one 32-bit word per OUTPUT pixel (index y * outputWidth + x) that records the
 * pixel's DISCRETE decisions - the places where a last-bit difference in float arithmetic selects a different branch, texel or
 * ray result. The HIP kernels emit the same words (plr_debug_set_decision_signature, include/plr.h). A parity test then demands the
 * per-channel tolerance from every pixel whose words agree and counts the pixels whose words differ against a hard cap.
 *   sdfDiffuseTrace:  bit 0 ray hit, bit 1 simpleShadow(hit point) == 1, bit 2 hit colour zeroed (outside the influence radius or
 *                     self intersection), bits 3-10 which of the 8 neighbours the 3x3 resolve accepted (loop order x outer, y inner,
 *                     centre skipped), bits 11-31 (index of the instance that owns the closest hit) + 1, 0 without a hit
 *   filterIndirectDiffuseSpatial: TWO words per pixel (index 2 * pixel, 2 * pixel + 1): bit i (sample i of 32) of the first = (texelX + offScreen) & 1,
 *                     of the second = (texelY + offScreen) & 1 of the nearest texel the sample reads (a step to a neighbouring texel in x / y or
 *                     across the off-screen test toggles a bit)
 *   indirectLightUpscale: bit 0 isEdge, bit 1 / bit 2 = x / y offset of the closest-depth texel
 *   deferred shade:   bits 0-1 shadow cascade, bits 2-5 number of lit PCF taps (0..12), bit 6 geometry pixel, bit 7 sky pixel */
void orc_set_decision_signature(uint32_t* words, int64_t count);

#ifdef __cplusplus
}
#endif
#endif
