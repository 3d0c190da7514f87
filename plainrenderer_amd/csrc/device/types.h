// GPU-visible structs shared by the kernels; layouts follow the reference's std140/std430 blocks.
#pragma once
#include <stdint.h>

namespace plr {

// `global` UBO, resources/shaders/global.inc:4-33 == GlobalShaderInfo (ResourceDescriptions.h:174-203), 340 bytes.
// cameraCut is a 4-byte bool on the GPU side (offset 320).
struct GlobalUbo {
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
};
static_assert(sizeof(GlobalUbo) == 340, "GlobalShaderInfo must be 340 bytes");

// resources/shaders/lightBuffer.inc:4-8 (std430; host allocates 20 bytes, RenderFrontend.cpp:1458-1468)
struct LightBuffer {
    float sunColor[3];
    float previousFrameExposure;
    float sunStrengthExposed;
};

// resources/shaders/SDF.inc:4-10, 96 bytes
struct SDFInstance {
    float localExtends[3];
    uint32_t sdfTextureIndex;
    float meanAlbedo[3];
    float padding;
    float worldToLocal[16];
};
static_assert(sizeof(SDFInstance) == 96, "SDFInstance must be 96 bytes");

// resources/shaders/sdfCulling.inc:7-15
constexpr uint32_t kCullingTileSize = 32;
constexpr uint32_t kMaxObjectsPerTile = 100;
struct CulledInstancesPerTile {
    uint32_t objectCount;
    uint32_t indices[kMaxObjectsPerTile];
};
struct BoundingBox {
    float bbMin[3]; float padding1;
    float bbMax[3]; float padding2;
};

// resources/shaders/sunShadowCascades.inc:7-11 (std430), 304 bytes
struct ShadowCascadeInfo {
    float splits[4];
    float lightMatrices[4][16];
    float lightSpaceScale[4][2];
};
static_assert(sizeof(ShadowCascadeInfo) == 304, "ShadowCascadeInfo must be 304 bytes");

// resources/shaders/volumetricFroxelLighting.inc:6-16 (std140)
struct VolumetricLightingSettings {
    float windSampleOffset[3];
    float sampleOffset;
    float scatteringCoefficients[3];
    float maxDistance;
    float absorptionCoefficient;
    float baseDensity;
    float densityNoiseRange;
    float densityNoiseScale;
    float phaseFunctionG;
};

} // namespace plr
