// histogramPerTile.comp for the PLR_MATH_FAST set: the same integer bin counts as kernels/exposure_tonemap.hip (bit exact), without
// a software logarithm per pixel.
//
// bin(l) = uint(127 * clamp((log(l) - log(min)) / range, 0, 1)) is monotone in l, so it equals the number of thresholds t[1..127] with
// l >= t[b], where t[b] is the smallest float whose bin is >= b. The table is bisected once per pass with the exact function
// (launchHistogramThresholds, kernels/exposure_tonemap.hip) and lives in LDS; per pixel the hardware log2 gives a guess that is at most
// one bin off and two table compares settle it. tests/test_exposure_tonemap.py checks the table-driven bin against the exact one for every
// one of the 2^32 float bit patterns (plr_debug_verify_histogram_thresholds).
// The luminance itself (dot product, division by the previous exposure) keeps the shader's IEEE operation sequence, hence:
// PLR_BUILD_FLAGS: -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt
#include "../backend.h"
#include <map>
#include "../device/shading_common.h"
#include "../../../include/plr.h"
#include "fused_front.h"

namespace plr {

int launchHistogramThresholds(uint32_t* thresholds, uint32_t nBins, float minLuminance, float maxLuminance, hipStream_t stream);                            // kernels/exposure_tonemap.hip
int launchHistogramExactBins(uint8_t* out, uint32_t first, uint32_t count, uint32_t nBins, float minLuminance, float maxLuminance, hipStream_t stream);

namespace fasthist {

constexpr uint32_t kBins = 128; // the table-driven kernel is built for the reference's 128 bins (RenderFrontend.cpp:46); other counts take the general kernel

// thr: kBins floats, thr[0] = 0 (every non-negative value), thr[b] ascending; NaN entries (unreachable bins) compare false
PLR_DI uint32_t tableBin(float luminance, const float* thr, float guessScale, float guessBias) {
    // guess from the hardware log2: bin ~ (log2(l) * ln2 - logMin) / range * 127, at most one off before the compares
    const float g = __builtin_amdgcn_logf(luminance) * guessScale + guessBias; // -inf for l = 0, NaN for NaN / negative input
    int gi = (int)__builtin_amdgcn_fmed3f(g, 0.f, (float)(kBins - 2u));        // 0 .. 126; NaN -> 0
    // exact: count of thresholds <= l among {gi, gi + 1}, everything below gi is <=, everything above gi + 1 is >
    const float lo = thr[gi], hi = thr[gi + 1];
    gi += (luminance >= hi) ? 1 : 0;
    gi -= (luminance >= lo) ? 0 : 1;
    return (uint32_t)max(gi, 0);
}

struct HistParams {
    ImgView src;
    const LightBuffer* light;
    uint32_t* perTile;
    const uint32_t* thresholds;
    float guessScale, guessBias;
    uint32_t tilesX, tileY0, gridX, gridY;
    uint32_t tileX0; // first tile column of the dispatch (tile rendering: PassCtx::colSpan in tiles)
};

// one 32x32 tile; localHistogram / thr: kBins entries of LDS each
PLR_DI void histogramTileBlock(const HistParams& p, uint32_t tileXInGrid, uint32_t tileYIndex, uint32_t* localHistogram, float* thr) {
    const uint32_t tileXIndex = tileXInGrid + p.tileX0;
    const ImgView& src = p.src;
    const LightBuffer* __restrict__ light = p.light;
    uint32_t* __restrict__ perTile = p.perTile;
    const uint32_t* __restrict__ thresholds = p.thresholds;
    const float guessScale = p.guessScale, guessBias = p.guessBias;
    const uint32_t tilesX = p.tilesX;
    const uint32_t t = threadIdx.x;
    const uint32_t tileY = tileYIndex + p.tileY0;
    if (t < kBins) { localHistogram[t] = 0u; thr[t] = u2f(thresholds[t]); }
    __syncthreads();
    const int x0 = (int)tileXIndex * 32 + (int)(t & 7u) * 4;
    const int y = (int)tileY * 32 + (int)(t >> 3);
    const float prevExposure = light->previousFrameExposure;
    uint32_t texels[4] = {0u, 0u, 0u, 0u};
    int nValid = 0;
    if (y < src.h && x0 < src.w) {
        const uint32_t* row = (const uint32_t*)src.ptr + (size_t)y * (size_t)src.w;
        nValid = min(4, src.w - x0);
        if (nValid == 4 && ((src.w & 3) == 0)) {
            const uint4 v = *(const uint4*)(row + x0);
            texels[0] = v.x; texels[1] = v.y; texels[2] = v.z; texels[3] = v.w;
        } else {
            for (int i = 0; i < nValid; i++) texels[i] = row[x0 + i];
        }
    }
    // A lane's four neighbouring pixels usually fall into one or two bins: they are merged in registers, then every distinct bin is one LDS
    // atomic. (The exact kernel peels the wave's leading bins with ballots; on images with per-pixel variation the wave holds a dozen bins and
    // the votes cost more than the atomics they save. A constant image serialises 64 lanes on one counter: 64 cycles per wave instruction,
    // ~13 us for a 4K frame in the worst case.)
    uint32_t bins[4];
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const vec3 c = unpackR11G11B10(texels[i]);
        const float luminance = dot(c, vec3(0.2126f, 0.7152f, 0.0722f)) / prevExposure; // :28-30, :53 (IEEE, as in the shader)
        bins[i] = i < nValid ? tableBin(luminance, thr, guessScale, guessBias) : 0xffffffffu;
    }
#pragma unroll
    for (int i = 0; i < 4; i++) {
        if (bins[i] == 0xffffffffu) continue;
        uint32_t count = 1u;
#pragma unroll
        for (int j = i + 1; j < 4; j++)
            if (bins[j] == bins[i]) { count++; bins[j] = 0xffffffffu; }
        atomicAdd(&localHistogram[bins[i]], count);
    }
    __syncthreads();
    const uint32_t tileIndex = tileXIndex + tileY * tilesX;
    if (t < kBins) {
        // bin b of a tile is written back only if the reference invocation with localIndexFlat == b lies inside the image (see kernels/exposure_tonemap.hip)
        const int rx = (int)tileXIndex * 32 + (int)(t & 31u), ry = (int)tileY * 32 + (int)(t >> 5);
        if (rx < src.w && ry < src.h) perTile[(size_t)tileIndex * kBins + t] = localHistogram[t];
    }
}

__global__ __launch_bounds__(256) void histogramPerTileFastKernel(HistParams p) {
    __shared__ uint32_t localHistogram[kBins];
    __shared__ float thr[kBins];
    histogramTileBlock(p, blockIdx.x, blockIdx.y, localHistogram, thr);
}

// launch 1 of the fused frame front (fused_front.h): the per-tile histogram's blocks and the depth pyramid's quad blocks in one grid, four to one
// while both last (the first kind is bound by VALU / LDS, the second by HBM: interleaved, every CU hosts both)
// fillSlot: the frame's pending buffer fills (backend.h PassCtx::pendingFillSlot), applied by one more block - this kernel reads and writes none of their destinations
// (the launcher checks), and everything that does is launched behind it
template <bool DOWNSCALE>
__global__ __launch_bounds__(256) void histogramAndPyramidKernel(HistParams p, fasthiz::QuadParams q, uint32_t histBlocks, uint32_t hizBlocks, uint32_t hizGridX, uint8_t* fillSlot) {
    __shared__ uint32_t localHistogram[kBins];
    __shared__ float thr[kBins];
    const uint32_t b = blockIdx.x;
    if (b == histBlocks + hizBlocks) { applyFillsBlock(fillSlot, 0); return; } // (only launched when fillSlot is set)
    const uint32_t paired = min(hizBlocks, histBlocks / 4u), remHist = histBlocks - 4u * paired;
    bool isHiz;
    uint32_t index;
    if (b < 5u * paired) { const uint32_t g = b / 5u, r = b % 5u; isHiz = r == 4u; index = isHiz ? g : 4u * g + r; }
    else { const uint32_t rem = b - 5u * paired; isHiz = rem >= remHist; index = isHiz ? paired + (rem - remHist) : 4u * paired + rem; }
    if (isHiz) fasthiz::hizQuadBlock<4, DOWNSCALE>(q, (int)(index % hizGridX) + q.tileX0, (int)(index / hizGridX) + q.tileY0); // (the whole frame's launch starts at tile 0, 0)
    else histogramTileBlock(p, index % p.gridX, index / p.gridX, localHistogram, thr);
}

struct TableKey { float minL, maxL; uint32_t valid; };

// 0: *out filled (gridY == 0: nothing to do); kUseGeneralKernel; < 0
static int prepare(const PassCtx& c, HistParams* out) {
    if (!c.hasSampled(2) || c.sampled[2].fmt != F_R11G11B10 || !c.hasSbuf(3) || !c.hasSbuf(0)) return kUseGeneralKernel;
    const uint32_t nBins = c.specUint(0, 64u);
    const float minL = c.specFloat(1, 1.f), maxL = c.specFloat(2, 100.f);
    if (nBins != kBins || !(minL > 0.f) || !(maxL > minL)) return kUseGeneralKernel;
    const ImgView& src = c.sampled[2];
    const uint32_t tilesX = divUp((unsigned)src.w, 32u), tilesY = divUp((unsigned)src.h, 32u);
    if (c.sbuf[0].size < (size_t)tilesX * tilesY * nBins * 4u || c.sbuf[3].size < sizeof(LightBuffer)) return kUseGeneralKernel;
    const PassCtx::RowSpan rs = c.rowSpan((int)tilesY, 1);
    // per-pass scratch: the threshold table, keyed on (memory, minL, maxL) and marked valid only once its launch succeeded: a failed launch or a pass
    // re-created with another luminance range rebuilds it instead of binning against a zero / stale table. The "guess is at most one bin off"
    // property is verified for every float for the reference's range only (plr_debug_verify_histogram_thresholds): other ranges take the general kernel
    if (minL != 0.001f || maxL != 200000.f) return kUseGeneralKernel;
    const bool fresh = c.scratchSize && *c.scratchSize < kBins * 4u; // about to be allocated (zero-filled): whatever the map says about this address is history
    uint32_t* thresholds = (uint32_t*)c.scratch(kBins * 4u);
    if (!thresholds) return c.fail(-2, "histogramPerTile: cannot allocate scratch memory");
    static thread_local std::map<const void*, TableKey> built; // scratch memory -> what it holds (one backend per host thread)
    TableKey& key = built[(const void*)thresholds];
    if (fresh) key.valid = 0u;
    if (!key.valid || key.minL != minL || key.maxL != maxL) {
        key.valid = 0u;
        if (launchHistogramThresholds(thresholds, kBins, minL, maxL, c.stream)) return c.fail(-2, "histogramPerTile: threshold table launch failed");
        key.minL = minL; key.maxL = maxL; key.valid = 1u;
    }
    // guess = (log2(l) * ln2 - log(min)) / (log(max) - log(min)) * 127
    const double logMin = std::log((double)minL), range = std::log((double)maxL) - logMin;
    out->src = src; out->light = (const LightBuffer*)c.sbuf[3].ptr; out->perTile = (uint32_t*)c.sbuf[0].ptr; out->thresholds = thresholds;
    out->guessScale = (float)(0.6931471805599453 / range * (kBins - 1u)); out->guessBias = (float)(-logMin / range * (kBins - 1u));
    const PassCtx::ColSpan cs = c.colSpan((int)tilesX, 1); // tile columns [x0, x1) of the recorded dispatch (one workgroup per tile)
    out->tilesX = tilesX; out->tileY0 = (uint32_t)rs.y0; out->gridX = (uint32_t)(cs.x1 - cs.x0); out->gridY = rs.y1 > rs.y0 ? (uint32_t)(rs.y1 - rs.y0) : 0u;
    out->tileX0 = (uint32_t)cs.x0;
    return 0;
}
static int launch(const PassCtx& c) {
    HistParams p;
    if (int rc = prepare(c, &p)) return rc;
    if (p.gridY == 0 || p.gridX == 0) return 0;
    histogramPerTileFastKernel<<<dim3(p.gridX, p.gridY), 256, 0, c.stream>>>(p);
    PLR_CHECK_LAUNCH(c);
    return 0;
}

// The frame's pending fills (backend.h PassCtx::pendingFillSlot) ride in launch 1 of a front when it touches none of their destinations (uniform / storage buffers the
// host writes per frame: the frustum, the culled-instance count, filter weights, the rotating copy of the global buffer ... - launch 1 reads the colour and depth images
// and the light buffer): *fillSlot = the table for one more block of the launch, or null after the fills went out as a launch of their own (a dependent launch of
// ~ 10 us and its two gaps in front of every frame: profiles/r06c_band_timeline.txt, applyFillsKernel - the band front did not take them until round 6c)
static int takeFillsIfDisjoint(const PassCtx& c, const HistParams& hp, uint8_t** fillSlot) {
    *fillSlot = nullptr;
    if (!c.pendingFillSlot) return 0;
    const uint8_t* slot = c.pendingFillSlot;
    const uint32_t nFills = *(const uint32_t*)slot;
    const FillEntry* entries = (const FillEntry*)(slot + kFillTableHeader); // pinned host memory: readable here
    auto touches = [&](const void* base, size_t bytes) {
        for (uint32_t i = 0; i < nFills; i++)
            if (entries[i].dst < (uint64_t)(uintptr_t)base + bytes && (uint64_t)(uintptr_t)base < entries[i].dst + entries[i].size) return true;
        return false;
    };
    const bool disjoint = !touches(hp.light, sizeof(LightBuffer)) && !touches(hp.perTile, (size_t)hp.tilesX * (hp.tileY0 + hp.gridY) * kBins * 4u) &&
                          !touches(hp.thresholds, kBins * 4u);
    if (disjoint) { *fillSlot = c.pendingFillSlot; c.pendingFillsTaken = true; return 0; }
    return c.applyPendingFillsNow();
}

// the six passes of the frame front - eight with the camera culling's two behind them - as two launches (fused_front.h)
static int launchFusedFront(const PassCtx* const* ctxs, size_t count) {
    if (count != 6 && count != 8) return kUseGeneralKernel;
    HistParams hp;
    if (int rc = prepare(*ctxs[0], &hp)) return rc;
    if (hp.gridX == 0 || hp.gridY == 0 || hp.tileX0 != 0) return kUseGeneralKernel; // (the fused front is the whole frame's: a tile's passes fuse on their own)
    ExposureChainPlan ep;
    if (int rc = prepareExposureChain(ctxs + 1, &ep)) return rc;
    if (ep.perTile != hp.perTile) return kUseGeneralKernel; // the combine must read what the per-tile pass writes
    fasthiz::Plan zp;
    if (int rc = fasthiz::prepare(*ctxs[4], ctxs[5], &zp)) return rc;
    FusedCullParams cull;
    int cullLevel = -1;
    if (zp.perTile) {
        // a per-tile pyramid (a frame beyond the shader's 11 levels: 8K on one GPU) has no chain tail; with the culling behind it launch 2 is the band front's
        // (tile levels 4 and 5, culling, combine) with the exposure in its last combine block - fused_front.h
        if (count != 8) return kUseGeneralKernel;
        bool useHiZ = false;
        ImgView hiz;
        if (int rc = prepareFusedCulling(*ctxs[6], *ctxs[7], &cull, &useHiZ, &hiz)) return rc;
        const fasthiz::TileTailParams& t = zp.tileTail;
        if (!useHiZ || hiz.ptr != (const void*)t.level4 || hiz.w != t.w4 || hiz.h != t.h4) return kUseGeneralKernel;
        uint8_t* fillSlotT = nullptr;
        if (int rc = takeFillsIfDisjoint(*ctxs[0], hp, &fillSlotT)) return rc;
        const uint32_t histBlocksT = hp.gridX * hp.gridY, hizBlocksT = (uint32_t)(zp.gridX * zp.gridY);
        histogramAndPyramidKernel<true><<<histBlocksT + hizBlocksT + (fillSlotT ? 1u : 0u), 256, 0, ctxs[0]->stream>>>(hp, zp.quad, histBlocksT, hizBlocksT, (uint32_t)zp.gridX, fillSlotT);
        PLR_CHECK_LAUNCH(*ctxs[0]);
        return launchTileFrontSecondWithExposure(zp, cull, ep, ctxs[0]->stream);
    }
    if (count == 8) {
        // launch 2 hosts the culling when its tiles sample the pyramid level launch 2 starts with or the one above it (not a level the quad blocks make)
        bool useHiZ = false;
        ImgView hiz;
        if (int rc = prepareFusedCulling(*ctxs[6], *ctxs[7], &cull, &useHiZ, &hiz)) return rc;
        if (!useHiZ || zp.tailFirst < 1) return kUseGeneralKernel;
        for (int l = zp.tailFirst; l <= zp.tailFirst + 1 && l < zp.tail.count; l++)
            if (hiz.ptr == (const void*)zp.tail.level[l] && hiz.w == zp.tail.w[l] && hiz.h == zp.tail.h[l]) cullLevel = l;
        if (cullLevel < 0) return kUseGeneralKernel;
    }
    // the pyramid's inputs are not outputs of the exposure chain (and vice versa): nothing else orders the two chains
    const uint32_t histBlocks = hp.gridX * hp.gridY, hizBlocks = (uint32_t)(zp.gridX * zp.gridY);
    uint8_t* fillSlot = nullptr;
    if (int rc = takeFillsIfDisjoint(*ctxs[0], hp, &fillSlot)) return rc;
    histogramAndPyramidKernel<true><<<histBlocks + hizBlocks + (fillSlot ? 1u : 0u), 256, 0, ctxs[0]->stream>>>(hp, zp.quad, histBlocks, hizBlocks, (uint32_t)zp.gridX, fillSlot);
    PLR_CHECK_LAUNCH(*ctxs[0]);
    return launchExposureChainAndPyramidTail(ep, zp, ctxs[0]->stream, count == 8 ? &cull : nullptr, cullLevel);
}

// the seven passes in front of a band's histogram all-reduce - per-tile histogram, reset, combine, pyramid, depth downscale, the two culling passes - as two launches
// (fused_front.h; the backend has sunk the all-reduce callback and the exposure pass behind them)
static int launchBandFront(const PassCtx* const* ctxs, size_t count) {
    if (count != 7) return kUseGeneralKernel;
    HistParams hp;
    if (int rc = prepare(*ctxs[0], &hp)) return rc;
    if (hp.gridX == 0 || hp.gridY == 0) return kUseGeneralKernel;
    ResetCombinePlan rp;
    if (int rc = prepareResetCombine(ctxs + 1, &rp)) return rc;
    if (rp.perTileBase != hp.perTile) return kUseGeneralKernel; // the combine must read what the per-tile pass writes
    fasthiz::Plan zp;
    if (int rc = fasthiz::prepare(*ctxs[3], ctxs[4], &zp)) return rc;
    if (!zp.perTile) return kUseGeneralKernel; // the whole frame's chain has its own front (launchFusedFront)
    FusedCullParams cull;
    bool useHiZ = false;
    ImgView hiz;
    if (int rc = prepareFusedCulling(*ctxs[5], *ctxs[6], &cull, &useHiZ, &hiz)) return rc;
    const fasthiz::TileTailParams& t = zp.tileTail;
    if (!useHiZ || hiz.ptr != (const void*)t.level4 || hiz.w != t.w4 || hiz.h != t.h4) return kUseGeneralKernel; // the tiles must sample the level launch 2 finishes
    uint8_t* fillSlot = nullptr;
    if (int rc = takeFillsIfDisjoint(*ctxs[0], hp, &fillSlot)) return rc;
    const uint32_t histBlocks = hp.gridX * hp.gridY, hizBlocks = (uint32_t)(zp.gridX * zp.gridY);
    histogramAndPyramidKernel<true><<<histBlocks + hizBlocks + (fillSlot ? 1u : 0u), 256, 0, ctxs[0]->stream>>>(hp, zp.quad, histBlocks, hizBlocks, (uint32_t)zp.gridX, fillSlot);
    PLR_CHECK_LAUNCH(*ctxs[0]);
    return launchBandFrontSecond(zp, cull, rp, ctxs[0]->stream);
}

// ---- exhaustive verification (plr_debug_verify_histogram_thresholds)
__global__ void verifyKernel(const uint32_t* __restrict__ thresholds, const uint8_t* __restrict__ exact, uint32_t first, uint32_t count, float guessScale, float guessBias,
                             unsigned long long* __restrict__ mismatches) {
    __shared__ float thr[kBins];
    if (threadIdx.x < kBins) thr[threadIdx.x] = u2f(thresholds[threadIdx.x]);
    __syncthreads();
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    if (tableBin(u2f(first + i), thr, guessScale, guessBias) != (uint32_t)exact[i]) atomicAdd(mismatches, 1ull);
}

} // namespace fasthist

static int fasthist_launch(const PassCtx& c) { return fasthist::launch(c); }
PLR_REGISTER_SHADER_FAST("histogramPerTile.comp", fasthist_launch);
static int fused_frame_front(const PassCtx* const* ctxs, size_t count) { return fasthist::launchFusedFront(ctxs, count); }
static int fused_band_front(const PassCtx* const* ctxs, size_t count) { return fasthist::launchBandFront(ctxs, count); }
PLR_REGISTER_FUSION_TAKES_FILLS("band front: histogram + reset + combine || per-tile depth pyramid || camera culling", fused_band_front, "histogramPerTile.comp", "histogramReset.comp", "histogramCombineTiles.comp",
                    "depthHiZPyramid.comp", "depthDownscale.comp", "sdfCameraFrustumCulling.comp", "sdfCameraTileCulling.comp");
static int fused_frame_front_and_culling(const PassCtx* const* ctxs, size_t count) { return fasthist::launchFusedFront(ctxs, count); }
PLR_REGISTER_FUSION_TAKES_FILLS("frame front: histogram + exposure chain || depth pyramid || camera culling", fused_frame_front_and_culling, "histogramPerTile.comp", "histogramReset.comp", "histogramCombineTiles.comp",
                    "preExposeLights.comp", "depthHiZPyramid.comp", "depthDownscale.comp", "sdfCameraFrustumCulling.comp", "sdfCameraTileCulling.comp");
PLR_REGISTER_FUSION_TAKES_FILLS("frame front: histogram + exposure chain || depth pyramid", fused_frame_front, "histogramPerTile.comp", "histogramReset.comp", "histogramCombineTiles.comp",
                    "preExposeLights.comp", "depthHiZPyramid.comp", "depthDownscale.comp");
} // namespace plr

using namespace plr;

// compares the table-driven bin with the shader's formula for all 2^32 float bit patterns; *out_mismatches must come back 0
extern "C" int plr_debug_verify_histogram_thresholds(float min_luminance, float max_luminance, uint64_t* out_mismatches) {
    if (!out_mismatches || !(min_luminance > 0.f) || !(max_luminance > min_luminance)) return setLastError(PLR_ERR_INVALID_ARGUMENT, "plr_debug_verify_histogram_thresholds: invalid argument");
    uint32_t* thresholds = nullptr;
    uint8_t* exact = nullptr;
    unsigned long long* counter = nullptr;
    const uint32_t chunk = 1u << 26;
    if (hipMalloc((void**)&thresholds, fasthist::kBins * 4u) != hipSuccess || hipMalloc((void**)&exact, chunk) != hipSuccess || hipMalloc((void**)&counter, 8) != hipSuccess)
        return setLastError(PLR_ERR_HIP, "plr_debug_verify_histogram_thresholds: hipMalloc failed");
    hipMemset(counter, 0, 8);
    int rc = launchHistogramThresholds(thresholds, fasthist::kBins, min_luminance, max_luminance, nullptr);
    const double logMin = std::log((double)min_luminance), range = std::log((double)max_luminance) - logMin;
    const float guessScale = (float)(0.6931471805599453 / range * (fasthist::kBins - 1u)), guessBias = (float)(-logMin / range * (fasthist::kBins - 1u));
    for (uint64_t first = 0; first < (1ull << 32) && !rc; first += chunk) {
        rc = launchHistogramExactBins(exact, (uint32_t)first, chunk, fasthist::kBins, min_luminance, max_luminance, nullptr);
        fasthist::verifyKernel<<<chunk / 256u, 256>>>(thresholds, exact, (uint32_t)first, chunk, guessScale, guessBias, counter);
    }
    unsigned long long host = ~0ull;
    const hipError_t e = hipMemcpy(&host, counter, 8, hipMemcpyDeviceToHost);
    hipFree(thresholds); hipFree(exact); hipFree(counter);
    if (rc || e != hipSuccess) return setLastError(PLR_ERR_HIP, "plr_debug_verify_histogram_thresholds: launch failed");
    *out_mismatches = host;
    return PLR_OK;
}
