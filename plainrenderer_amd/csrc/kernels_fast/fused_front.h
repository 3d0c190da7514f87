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

} // namespace plr
