// The front of the frame is two independent chains: luminance histogram -> exposure, and depth pyramid (-> culling -> trace). Recorded back to
// back (histogramPerTile, histogramReset, histogramCombineTiles, preExposeLights, depthHiZPyramid, depthDownscale) they are six dependent
// launches, most of them at the ~5 us floor a dependent kernel costs on this chip. Pass fusion (backend.h) runs them as TWO launches whose
// blocks belong to either chain:
//   launch 1: per-tile histogram blocks (VALU / LDS bound) interleaved 4 : 1 with the pyramid's quad blocks (HBM bound)   [histogram_fast.hip]
//   launch 2: the exposure chain's blocks (combine + last-block exposure) beside the one block that finishes the pyramid   [exposure_tonemap.hip]
// and when sdfCameraFrustumCulling + sdfCameraTileCulling follow directly (eight passes), launch 2 hosts their blocks as well: the culling's only input
// from this frame is one pyramid texel per tile, which a culling block evaluates from the level launch 1 finished (a third dependent launch less).
// Every block runs exactly the code of its own kernel, so results are unchanged.
#pragma once
#include "../backend.h"
#include "../device/hiz_fast_device.h"
#include "../device/culling_device.h"

namespace plr {

// histogramReset + histogramCombineTiles + preExposeLights, validated and resolved (kernels/exposure_tonemap.hip)
struct ExposureChainPlan {
    const uint32_t* perTile = nullptr;
    uint32_t* histogram = nullptr;
    uint32_t nBins = 0, nTiles = 0, blocks = 0;
    void* scratch = nullptr;
    void* light = nullptr;
    ImgView transmissionLut;
    const GlobalUbo* global = nullptr;
    float minLuminanceLog = 0.f, maxLuminanceLog = 0.f;
};
int prepareExposureChain(const PassCtx* const* ctxs3, ExposureChainPlan* out);                                        // 0 / kUseGeneralKernel / < 0
// cull: the camera culling's two passes hosted by the same launch (validated by prepareFusedCulling), or null; cullLevel: the pyramid level its tiles sample,
// h.tailFirst or h.tailFirst + 1
int launchExposureChainAndPyramidTail(const ExposureChainPlan& e, const fasthiz::Plan& h, hipStream_t stream, const FusedCullParams* cull = nullptr, int cullLevel = 0); // 0 / < 0

// ---- a BAND's (or tile's) front, round 5. Recorded: histogramPerTile, histogramReset, histogramCombineTiles, [all-reduce callback], preExposeLights, depthHiZPyramid,
// depthDownscale, sdfCameraFrustumCulling, sdfCameraTileCulling. The callback names the histogram buffer, so the backend may sink it (and the exposure pass, which
// reads that buffer) behind the pyramid and the culling, which touch neither (backend.cpp gatherFusionGroups); the seven passes in front of it then run as two launches:
//   launch 1: the per-tile histogram's blocks beside the pyramid's quad blocks (histogramAndPyramidKernel, as for the whole frame)           [histogram_fast.hip]
//   launch 2: the tiles' levels 4 and 5, the culling's blocks, the reset + combine blocks (bandFrontSecondKernel)                             [exposure_tonemap.hip]
// instead of four (per-tile histogram | reset + combine | quad blocks | tile tail + culling): the two histogram launches sat at their launch floors.
struct ResetCombinePlan {
    const uint32_t* perTileBase = nullptr; // the per-tile histogram buffer
    const uint32_t* perTile = nullptr;     // the first tile of the dispatch in it
    uint32_t* histogram = nullptr;
    uint32_t nBins = 0, nTiles = 0, blocks = 0;
    void* scratch = nullptr;
};
int prepareResetCombine(const PassCtx* const* ctxs2, ResetCombinePlan* out); // histogramReset + histogramCombineTiles: 0 / kUseGeneralKernel / < 0 (kernels/exposure_tonemap.hip)
int launchBandFrontSecond(const fasthiz::Plan& h, const FusedCullParams& cull, const ResetCombinePlan& r, hipStream_t stream); // 0 / < 0
// the same launch for a frame with a per-tile pyramid that is not partitioned (8K on one GPU): the exposure runs in the block that takes the last combine ticket
int launchTileFrontSecondWithExposure(const fasthiz::Plan& h, const FusedCullParams& cull, const ExposureChainPlan& e, hipStream_t stream);

} // namespace plr
