// Auto-exposure + tonemap passes for gfx950.
//   histogramPerTile.comp / histogramReset.comp / histogramCombineTiles.comp / preExposeLights.comp / tonemapping.comp
// (resources/shaders/, host side RenderFrontend.cpp:707-790,931-945).
//
// All five are HBM-streaming or tiny: no MFMA. Pixels are read 16 bytes per lane (4 packed R11G11B10 texels),
// histograms are built in LDS with wave-level vote aggregation, and the serial exposure shader is spread over
// one wave with the float accumulation kept in bin order so results stay bit-identical to a scalar evaluation.
#include "../backend.h"
#include "../device/shading_common.h"
#include "../kernels_fast/fused_front.h"

namespace plr {

// det_logf of device/detmath.h evaluated on the host for the two specialisation constants: the same IEEE operations
// in the same order (this file is built with -ffp-contract=off), hence the same bits as on the device.
static float hostDetLog(float x) {
    union { float f; uint32_t u; } cv; cv.f = x;
    uint32_t ix = cv.u; int e = 0;
    if (ix < 0x00800000u) { x = x * 8388608.0f; cv.f = x; ix = cv.u; e = -23; }
    e += (int)(ix >> 23) - 127;
    cv.u = (ix & 0x007fffffu) | 0x3f800000u;
    float m = cv.f;
    if (m > 1.41421354f) { m = m * 0.5f; e += 1; }
    const float f = m - 1.0f;
    const float s = f / (2.0f + f);
    const float z = s * s;
    float p = 0.222222224f;
    p = p * z + 0.285714298f; p = p * z + 0.400000006f; p = p * z + 0.666666687f;
    const float r = f - s * (f - z * p);
    const float fe = (float)e;
    return fe * PLR_LN2_HI + (fe * PLR_LN2_LO + r);
}

// bin of one luminance value (histogramPerTile.comp:53-57); the deterministic log makes it the same bits on host, device and in the oracle
PLR_DI uint32_t histogramBin(float luminance, uint32_t maxIndex, float minLuminanceLog, float range) {
    const float luminanceLog = det_logf(luminance);
    return (uint32_t)((float)maxIndex * gclamp((luminanceLog - minLuminanceLog) / range, 0.f, 1.f));
}

// ---- threshold table of the PLR_MATH_FAST per-tile kernel (kernels_fast/histogram_fast.hip) ----
// histogramBin is monotone in the luminance, so bin(l) = number of b in 1..maxIndex with l >= threshold[b], threshold[b] = the smallest
// non-negative float whose bin is >= b. One thread per threshold bisects over the float bit patterns with the exact function above;
// plr_debug_verify_histogram_thresholds checks the table-driven bin against the exact one for EVERY float (2^32 patterns).
__global__ void histogramThresholdKernel(uint32_t* __restrict__ thresholds, uint32_t nBins, float minLuminanceLog, float maxLuminanceLog) {
    const uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= nBins) return;
    const uint32_t maxIndex = nBins - 1u;
    const float range = maxLuminanceLog - minLuminanceLog;
    uint32_t lo = 0u, hi = 0x7f800000u; // bit patterns of +0 .. +inf: ordered like the values
    if (b == 0u) { thresholds[0] = 0u; return; }
    if (histogramBin(u2f(hi), maxIndex, minLuminanceLog, range) < b) { thresholds[b] = 0x7fc00000u; return; } // never reached: NaN compares false
    while (lo < hi) { // smallest pattern whose bin is >= b
        const uint32_t mid = lo + (hi - lo) / 2u;
        if (histogramBin(u2f(mid), maxIndex, minLuminanceLog, range) >= b) hi = mid; else lo = mid + 1u;
    }
    thresholds[b] = lo;
}
int launchHistogramThresholds(uint32_t* thresholds, uint32_t nBins, float minLuminance, float maxLuminance, hipStream_t stream) {
    histogramThresholdKernel<<<divUp(nBins, 64u), 64, 0, stream>>>(thresholds, nBins, hostDetLog(minLuminance), hostDetLog(maxLuminance));
    return hipGetLastError() == hipSuccess ? 0 : -2;
}
// exact bins of `count` consecutive float bit patterns starting at `first` (the verification's reference side)
__global__ void histogramExactBinsKernel(uint8_t* __restrict__ out, uint32_t first, uint32_t count, uint32_t nBins, float minLuminanceLog, float maxLuminanceLog) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < count) out[i] = (uint8_t)histogramBin(u2f(first + i), nBins - 1u, minLuminanceLog, maxLuminanceLog - minLuminanceLog);
}
int launchHistogramExactBins(uint8_t* out, uint32_t first, uint32_t count, uint32_t nBins, float minLuminance, float maxLuminance, hipStream_t stream) {
    histogramExactBinsKernel<<<divUp(count, 256u), 256, 0, stream>>>(out, first, count, nBins, hostDetLog(minLuminance), hostDetLog(maxLuminance));
    return hipGetLastError() == hipSuccess ? 0 : -2;
}

// ------------------------------------------------------------------------------------------------
// histogramPerTile.comp:32-65. One 256-thread block per 32x32 tile; thread t owns 4 consecutive pixels of
// row t/8. The bin index must be bit exact, so the log is the deterministic one.
// Unlike the reference, out-of-image invocations still reach the barriers; the observable rule is kept:
// bin b of a tile is written back only if the reference invocation with localIndexFlat == b lies inside the image.
__global__ __launch_bounds__(256) void histogramPerTileKernel(ImgView src, const LightBuffer* __restrict__ light, uint32_t* __restrict__ perTile,
                                                              uint32_t nBins, float minLuminanceLog, float maxLuminanceLog, uint32_t tilesX, uint32_t tileY0, uint32_t tileX0) {
    extern __shared__ uint32_t localHistogram[];
    const uint32_t t = threadIdx.x;
    const uint32_t tileY = blockIdx.y + tileY0, tileX = blockIdx.x + tileX0; // (tileX0: tile rendering, PassCtx::colSpan)
    for (uint32_t b = t; b < nBins; b += 256u) localHistogram[b] = 0u;
    __syncthreads();

    const int x0 = (int)tileX * 32 + (int)(t & 7u) * 4;
    const int y = (int)tileY * 32 + (int)(t >> 3);
    const float prevExposure = light->previousFrameExposure;
    const uint32_t maxIndex = nBins - 1u;
    const float range = maxLuminanceLog - minLuminanceLog;

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
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const bool valid = i < nValid;
        uint32_t bin = 0u;
        if (valid) {
            const vec3 c = unpackR11G11B10(texels[i]);
            const float luminance = dot(c, vec3(0.2126f, 0.7152f, 0.0722f)) / prevExposure; // :28-30, :53
            bin = histogramBin(luminance, maxIndex, minLuminanceLog, range);
        }
        // wave vote aggregation: neighbouring pixels mostly share a bin, so peel the leading bins with one LDS
        // atomic each, then let the stragglers add individually
        bool pending = valid;
        for (int round = 0; round < 3; round++) {
            const unsigned long long todo = __ballot(pending);
            if (todo == 0ull) break;
            const int leader = __ffsll((long long)todo) - 1;
            const uint32_t leaderBin = (uint32_t)__shfl((int)bin, leader);
            const unsigned long long same = __ballot(pending && bin == leaderBin);
            if ((int)(threadIdx.x & 63u) == leader) atomicAdd(&localHistogram[leaderBin], (uint32_t)__popcll(same));
            if (bin == leaderBin) pending = false;
        }
        if (pending) atomicAdd(&localHistogram[bin], 1u);
    }
    __syncthreads();

    const uint32_t tileIndex = tileX + tileY * tilesX;
    for (uint32_t b = t; b < nBins && b < 1024u; b += 256u) {
        const int rx = (int)tileX * 32 + (int)(b & 31u), ry = (int)tileY * 32 + (int)(b >> 5);
        if (rx < src.w && ry < src.h) perTile[(size_t)tileIndex * nBins + b] = localHistogram[b];
    }
}

static int launchHistogramPerTile(const PassCtx& c) {
    if (int rc = c.needSampled(2, F_R11G11B10, "histogramPerTile srcTexture")) return rc;
    if (int rc = c.needSbuf(3, sizeof(LightBuffer), "histogramPerTile lightBuffer")) return rc;
    const uint32_t nBins = c.specUint(0, 64u);
    const float minL = c.specFloat(1, 1.f), maxL = c.specFloat(2, 100.f);
    const ImgView& src = c.sampled[2];
    const uint32_t tilesX = divUp((unsigned)src.w, 32u), tilesY = divUp((unsigned)src.h, 32u);
    // the reference sizes this buffer for 1920x1080 only (RenderFrontend.cpp:1069-1070); demand the real tile count
    if (int rc = c.needSbuf(0, (size_t)tilesX * tilesY * nBins * 4u, "histogramPerTile per-tile buffer")) return rc;
    if (nBins == 0 || nBins > 1024u) return c.fail(-6, "histogramPerTile: nBins must be in 1..1024");
    // host-side log of the two specialisation constants with the same deterministic routine (exact same bits as device)
    const PassCtx::RowSpan rs = c.rowSpan((int)tilesY, 1); // tile rows [y0, y1) of the recorded dispatch (one workgroup per tile)
    if (rs.y1 <= rs.y0) return 0;
    const PassCtx::ColSpan cs = c.colSpan((int)tilesX, 1); // tile columns of the recorded dispatch
    if (cs.x1 <= cs.x0) return 0;
    const dim3 grid((unsigned)(cs.x1 - cs.x0), (unsigned)(rs.y1 - rs.y0));
    if (!(minL > 0.f) || !(maxL > 0.f)) return c.fail(-1, "histogramPerTile: luminance range must be positive");
    const float logs[2] = {hostDetLog(minL), hostDetLog(maxL)};
    histogramPerTileKernel<<<grid, 256, nBins * sizeof(uint32_t), c.stream>>>(src, (const LightBuffer*)c.sbuf[3].ptr, (uint32_t*)c.sbuf[0].ptr,
                                                                               nBins, logs[0], logs[1], tilesX, (uint32_t)rs.y0, (uint32_t)cs.x0);
    PLR_CHECK_LAUNCH(c);
    return 0;
}
PLR_REGISTER_SHADER("histogramPerTile.comp", launchHistogramPerTile);

// ------------------------------------------------------------------------------------------------
// histogramReset.comp:11-16
__global__ void histogramResetKernel(uint32_t* __restrict__ histogram, uint32_t nBins, uint32_t nThreads) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < nThreads && i < nBins) histogram[i] = 0u;
}
static int launchHistogramReset(const PassCtx& c) {
    const uint32_t nBins = c.specUint(0, 64u);
    if (int rc = c.needSbuf(1, (size_t)nBins * 4u, "histogramReset histogram")) return rc;
    const uint32_t nThreads = c.dispatch[0] * 64u;
    histogramResetKernel<<<divUp(nThreads, 64u), 64, 0, c.stream>>>((uint32_t*)c.sbuf[1].ptr, nBins, nThreads);
    PLR_CHECK_LAUNCH(c);
    return 0;
}
PLR_REGISTER_SHADER("histogramReset.comp", launchHistogramReset);

// ------------------------------------------------------------------------------------------------
// histogramCombineTiles.comp:27-34. The reference launches (tiles x 2) groups that each issue 64 global atomics;
// here one block sums a slab of tiles in registers (coalesced 4*nBins-byte rows) and issues one atomic per bin.
constexpr uint32_t kCombineTilesPerBlock = 32;
__global__ void histogramCombineKernel(const uint32_t* __restrict__ perTile, uint32_t* __restrict__ histogram, uint32_t nBins, uint32_t tile0, uint32_t nTiles, uint32_t binLimit) {
    const uint32_t bin = threadIdx.x + blockIdx.y * blockDim.x;
    if (bin >= nBins || bin >= binLimit) return;
    const uint32_t t0 = tile0 + blockIdx.x * kCombineTilesPerBlock;
    const uint32_t t1 = min(t0 + kCombineTilesPerBlock, tile0 + nTiles);
    uint32_t sum = 0u;
    // eight loads in flight per step: one load per iteration made the kernel a chain of 32 memory round trips (13 us for 4 MB)
    uint32_t t = t0;
    for (; t + 8u <= t1; t += 8u) {
        uint32_t v[8];
#pragma unroll
        for (uint32_t i = 0; i < 8u; i++) v[i] = perTile[(size_t)(t + i) * nBins + bin];
#pragma unroll
        for (uint32_t i = 0; i < 8u; i++) sum += v[i];
    }
    for (; t < t1; t++) sum += perTile[(size_t)t * nBins + bin];
    if (sum) atomicAdd(&histogram[bin], sum);
}
static int launchHistogramCombine(const PassCtx& c) {
    const uint32_t nBins = c.specUint(0, 64u);
    const uint32_t tile0 = c.base[0], nTiles = c.dispatch[0]; // tiles [tile0, tile0 + nTiles)
    if (int rc = c.needSbuf(0, (size_t)(tile0 + nTiles) * nBins * 4u, "histogramCombineTiles per-tile buffer")) return rc;
    if (int rc = c.needSbuf(1, (size_t)nBins * 4u, "histogramCombineTiles histogram")) return rc;
    const uint32_t binLimit = c.dispatch[1] * 64u; // bins covered by the recorded dispatch
    const dim3 grid(divUp(nTiles, kCombineTilesPerBlock), divUp(std::min(nBins, binLimit), 128u));
    if (nTiles == 0) return 0;
    histogramCombineKernel<<<grid, 128, 0, c.stream>>>((const uint32_t*)c.sbuf[0].ptr, (uint32_t*)c.sbuf[1].ptr, nBins, tile0, nTiles, binLimit);
    PLR_CHECK_LAUNCH(c);
    return 0;
}
PLR_REGISTER_SHADER("histogramCombineTiles.comp", launchHistogramCombine);

// ------------------------------------------------------------------------------------------------
// preExposeLights.comp:28-88. The reference runs one invocation over a 128-step loop; here one wave computes the
// cumulative histogram with a shuffle scan and the exp() terms in parallel, and lane 0 replays only the float
// accumulation in bin order (float addition is not associative, the order is part of the result).
PLR_DI float offsetFromSceneEV(float sceneEV100) {
    const float darkExp = 2.84f, lightExp = 12.81f, lightOffset = 1.47f, darkOffset = -3.17f;
    const float t = gclamp((sceneEV100 - darkExp) / (lightExp - darkOffset), 0.f, 1.f); // sic: lightExp - darkOffset (:35)
    return gmix(darkOffset, lightOffset, t);
}

constexpr int kMaxExposureBins = 1024;
// one wave; term / counted: LDS arrays of kMaxExposureBins entries. HIST(i) returns bin i's count.
template <class Hist>
PLR_DI void preExposeLightsWave(LightBuffer* __restrict__ light, Hist hist, const ImgView& transmissionLut, const GlobalUbo* __restrict__ g, int nBins,
                                float minLuminanceLog, float maxLuminanceLog, float* term, uint32_t* counted, int lane) {
    const uint32_t pixelCount = (uint32_t)(g->screenResolution[0] * g->screenResolution[1]);
    // inputs of the serial tail, fetched while the histogram scan runs (they do not depend on it)
    const vec4 sunTransmission = sampleLinear2D<F_R11G11B10, CLAMP>(transmissionLut, vec2(0.f, -g->sunDirection[1] * 0.5f + 0.5f));
    const float previousExposure = light->previousFrameExposure, exposureOffsetUser = g->exposureOffset, sunStrength = g->sunStrength;
    const float evMaxChange = g->exposureAdaptionSpeedEvPerSec * g->deltaTime;
    uint32_t carry = 0u;
    for (int base = 0; base < nBins; base += 64) {
        const int i = base + lane;
        const uint32_t h = i < nBins ? hist(i) : 0u;
        uint32_t incl = h;
        for (int d = 1; d < 64; d <<= 1) {
            const uint32_t up = (uint32_t)__shfl_up((int)incl, d);
            if (lane >= d) incl += up;
        }
        incl += carry;
        carry = (uint32_t)__shfl((int)incl, 63);
        if (i < nBins) {
            const float percentage = (float)incl / (float)pixelCount;
            const bool take = percentage < 0.95f && percentage >= 0.5f;
            float tv = 0.f;
            if (take) {
                const float binValueLog = minLuminanceLog + (maxLuminanceLog - minLuminanceLog) * (float)i / ((float)nBins - 1.f);
                tv = (float)h * det_expf(binValueLog);
            }
            term[i] = tv;
            counted[i] = take ? h : 0xffffffffu;
        }
    }
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); // the wave's own LDS writes before lane 0 reads them back
    if (lane != 0) return;
    float mean = 0.f;
    uint32_t countedPixels = 0u;
    // the float additions stay in bin order; a bin outside the percentile window has term 0 (mean + 0 = mean: the terms are non-negative, so
    // mean is never -0) - without the per-bin branch the LDS reads of eight bins are in flight together instead of one after the other
#pragma unroll 8
    for (int i = 0; i < nBins; i++) {
        const uint32_t cnt = counted[i];
        mean += term[i];
        countedPixels += cnt != 0xffffffffu ? cnt : 0u;
    }
    mean /= (float)countedPixels;
    const float sceneEV100 = det_log2f(mean * 100.f / 12.5f);
    float exposureOffset = offsetFromSceneEV(sceneEV100);
    exposureOffset += exposureOffsetUser;
    float targetEV100 = sceneEV100 - exposureOffset;
    targetEV100 = gmax(targetEV100, 10.f);
    const float previousEV100 = det_log2f(1.f / (gmax(previousExposure, 0.000001f) * 1.2f));
    const float evDelta = targetEV100 - previousEV100;
    const float evChange = gsign(evDelta) * gmin(fabsf(evDelta), fabsf(evMaxChange));
    const float currentEV100 = previousEV100 + evChange;
    const float exposure = 1.f / (det_powf(2.f, currentEV100) * 1.2f);
    light->sunStrengthExposed = sunStrength * exposure;
    light->previousFrameExposure = exposure;
    light->sunColor[0] = sunTransmission.x; light->sunColor[1] = sunTransmission.y; light->sunColor[2] = sunTransmission.z;
}

__global__ __launch_bounds__(64) void preExposeLightsKernel(LightBuffer* __restrict__ light, const uint32_t* __restrict__ histogram, ImgView transmissionLut,
                                                            const GlobalUbo* __restrict__ g, int nBins, float minLuminanceLog, float maxLuminanceLog) {
    __shared__ float term[kMaxExposureBins];
    __shared__ uint32_t counted[kMaxExposureBins];
    preExposeLightsWave(light, [&](int i) { return histogram[i]; }, transmissionLut, g, nBins, minLuminanceLog, maxLuminanceLog, term, counted, (int)threadIdx.x);
}

// ---- pass fusion (backend.h): histogramReset + histogramCombineTiles + preExposeLights recorded back to back, as one launch.
// Every block sums its slab of tiles like histogramCombineKernel and adds it to a zeroed accumulator with device-scope atomics; the block that
// takes the last ticket stores the totals into the histogram buffer (= reset + combine), zeroes the accumulator for the next frame and runs
// the exposure wave on the totals. All cross-block traffic goes through device-scope atomics (performed at the memory side, coherent across
// the eight XCDs' L2s), ordered by a release / acquire ticket (below), not by instruction timing.
constexpr int kFusedExposureMaxBins = 256;
struct ExposureScratch { uint32_t ticket; uint32_t pad[3]; uint32_t acc[kFusedExposureMaxBins]; };
// block `block` of `blocks`; term / counted / totals / isLast: the block's LDS. EXPOSE = false: reset + combine only (band rendering: the exposure
// follows the all-reduce over the bands, perTile then points at the band's first tile)
template <bool EXPOSE = true>
PLR_DI void histogramCombineExposeBlock(uint32_t block, uint32_t blocks, const uint32_t* __restrict__ perTile, uint32_t* __restrict__ histogram, uint32_t nBins, uint32_t nTiles,
                                        ExposureScratch* __restrict__ scratch, LightBuffer* __restrict__ light, const ImgView& transmissionLut, const GlobalUbo* __restrict__ g,
                                        float minLuminanceLog, float maxLuminanceLog, float* term, uint32_t* counted, uint32_t* totals, uint32_t* isLast) {
    const uint32_t t0 = block * kCombineTilesPerBlock, t1 = min(t0 + kCombineTilesPerBlock, nTiles);
    for (uint32_t bin = threadIdx.x; bin < nBins; bin += blockDim.x) {
        uint32_t sum = 0u, t = t0;
        for (; t + 8u <= t1; t += 8u) {
            uint32_t v[8];
#pragma unroll
            for (uint32_t i = 0; i < 8u; i++) v[i] = perTile[(size_t)(t + i) * nBins + bin];
#pragma unroll
            for (uint32_t i = 0; i < 8u; i++) sum += v[i];
        }
        for (; t < t1; t++) sum += perTile[(size_t)t * nBins + bin];
        if (sum) __hip_atomic_fetch_add(&scratch->acc[bin], sum, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    // publication protocol: the block's accumulator adds happen-before the barrier, thread 0 then takes its ticket with an agent-scope
    // RELEASE (orders the whole block's adds before the ticket, by cumulativity over the workgroup barrier) + ACQUIRE (the block that draws
    // the last ticket observes every other block's adds); the second barrier hands the acquire to the other threads of the last block
    __syncthreads();
    if (threadIdx.x == 0) *isLast = __hip_atomic_fetch_add(&scratch->ticket, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT) == blocks - 1u ? 1u : 0u;
    __syncthreads();
    if (!*isLast) return;
    for (uint32_t bin = threadIdx.x; bin < nBins; bin += blockDim.x) {
        const uint32_t v = __hip_atomic_exchange(&scratch->acc[bin], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); // read the total, leave zero for the next frame
        totals[bin] = v;
        histogram[bin] = v;
    }
    if (threadIdx.x == 0) __hip_atomic_store(&scratch->ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (!EXPOSE) return;
    __syncthreads();
    if (threadIdx.x < 64) preExposeLightsWave(light, [&](int i) { return totals[i]; }, transmissionLut, g, (int)nBins, minLuminanceLog, maxLuminanceLog, term, counted, (int)threadIdx.x);
}

__global__ __launch_bounds__(128) void histogramCombineExposeKernel(const uint32_t* __restrict__ perTile, uint32_t* __restrict__ histogram, uint32_t nBins, uint32_t nTiles,
                                                                    ExposureScratch* __restrict__ scratch, LightBuffer* __restrict__ light, ImgView transmissionLut,
                                                                    const GlobalUbo* __restrict__ g, float minLuminanceLog, float maxLuminanceLog) {
    __shared__ float term[kMaxExposureBins];
    __shared__ uint32_t counted[kMaxExposureBins];
    __shared__ uint32_t totals[kFusedExposureMaxBins];
    __shared__ uint32_t isLast;
    histogramCombineExposeBlock(blockIdx.x, gridDim.x, perTile, histogram, nBins, nTiles, scratch, light, transmissionLut, g, minLuminanceLog, maxLuminanceLog, term, counted, totals, &isLast);
}

// histogramReset + histogramCombineTiles of a band's tiles (the exposure pass follows the all-reduce callback)
__global__ __launch_bounds__(128) void histogramResetCombineKernel(const uint32_t* __restrict__ perTile, uint32_t* __restrict__ histogram, uint32_t nBins, uint32_t nTiles,
                                                                   ExposureScratch* __restrict__ scratch) {
    __shared__ uint32_t totals[kFusedExposureMaxBins];
    __shared__ uint32_t isLast;
    histogramCombineExposeBlock<false>(blockIdx.x, gridDim.x, perTile, histogram, nBins, nTiles, scratch, nullptr, ImgView{}, nullptr, 0.f, 0.f, nullptr, nullptr, totals, &isLast);
}

// launch 2 of the fused frame front (kernels_fast/fused_front.h): block 0 finishes the depth pyramid, the next chainBlocks are the exposure chain's, and -
// CULL - the rest are the camera culling's (device/culling_device.h), 16 tiles each. A culling tile needs one texel of the pyramid level block 0 is
// writing in this very launch (the first tail level): it evaluates that texel itself from the level below, which launch 1 finished - the same footprint,
// the same bits - instead of waiting for block 0.
template <bool CULL>
__global__ __launch_bounds__(1024) void exposureChainAndPyramidTailKernel(const uint32_t* __restrict__ perTile, uint32_t* __restrict__ histogram, uint32_t nBins, uint32_t nTiles,
                                                                          ExposureScratch* __restrict__ scratch, LightBuffer* __restrict__ light, ImgView transmissionLut,
                                                                          const GlobalUbo* __restrict__ g, float minLuminanceLog, float maxLuminanceLog, HizParams tail,
                                                                          int tailFirst, int tailTexelsA, uint32_t chainBlocks, FusedCullParams cull, uint32_t cullBlocks, int cullLevel) {
    extern __shared__ float2 pyramidTailLds[];
    __shared__ float term[kMaxExposureBins];
    __shared__ uint32_t counted[kMaxExposureBins];
    __shared__ uint32_t totals[kFusedExposureMaxBins];
    __shared__ uint32_t isLast;
    if (blockIdx.x == 0) { fasthiz::hizTailBlock<1024>(tail, tailFirst, tailTexelsA, pyramidTailLds); return; }
    const uint32_t b = blockIdx.x - 1u;
    if (CULL && b >= chainBlocks) {
        // the frustum-culled list lives in the dynamic LDS block 0 uses for the pyramid (the launcher sizes it for both uses).
        // The level the tiles sample is tailFirst (4K: the quad blocks made four levels) or tailFirst + 1 (1080p: three): the first is one footprint over the level
        // launch 1 finished, the second a footprint of such footprints - at most 81 texels of a level that is in memory, against waiting for block 0.
        const int w1 = tail.w[tailFirst], h1 = tail.h[tailFirst], sw = tail.w[tailFirst - 1], sh = tail.h[tailFirst - 1];
        const float2* __restrict__ below = tail.level[tailFirst - 1];
        auto firstTailLevelAt = [&](int x, int y) {
            const MinMax m = footprint<false>(2 * x, 2 * y, sw, sh, sh & 1, sw & 1, [&](int sx, int sy) { return below[(size_t)sy * (size_t)sw + (size_t)sx]; });
            return make_float2(m.mn, m.mx);
        };
        const bool second = cullLevel == tailFirst + 1;
        const int w = tail.w[cullLevel], h = tail.h[cullLevel];
        frustumAndTileCullingBlock<true, 1024>(cull, b - chainBlocks, cullBlocks, (uint32_t*)pyramidTailLds, counted, &isLast, [&](vec2 uv) {
            // sampleNearest2D<F_RG32F, CLAMP> of level cullLevel at uv (device/image.h), the texel evaluated instead of loaded (depthHiZPyramid.comp:52-124)
            const int x = clampi((int)floorf(saneCoord(uv.x * (float)w)), w), y = clampi((int)floorf(saneCoord(uv.y * (float)h)), h);
            if (!second) return firstTailLevelAt(x, y);
            const MinMax m = footprint<false>(2 * x, 2 * y, w1, h1, h1 & 1, w1 & 1, [&](int sx, int sy) { return firstTailLevelAt(sx, sy); });
            return make_float2(m.mn, m.mx);
        });
        return;
    }
    histogramCombineExposeBlock(b, chainBlocks, perTile, histogram, nBins, nTiles, scratch, light, transmissionLut, g, minLuminanceLog, maxLuminanceLog, term, counted, totals, &isLast);
}

static int launchPreExposeLights(const PassCtx& c) {
    if (int rc = c.needGlobal()) return rc;
    if (int rc = c.needSbuf(0, sizeof(LightBuffer), "preExposeLights lightBuffer")) return rc;
    const int nBins = c.specInt(0, 64);
    if (nBins < 1 || nBins > kMaxExposureBins) return c.fail(-6, "preExposeLights: nBins must be in 1..1024");
    if (int rc = c.needSbuf(1, (size_t)nBins * 4u, "preExposeLights histogram")) return rc;
    if (int rc = c.needSampled(2, F_R11G11B10, "preExposeLights transmissionLut")) return rc;
    const float minL = c.specFloat(1, 1.f), maxL = c.specFloat(2, 100.f);
    if (!(minL > 0.f) || !(maxL > 0.f)) return c.fail(-1, "preExposeLights: luminance range must be positive");
    preExposeLightsKernel<<<1, 64, 0, c.stream>>>((LightBuffer*)c.sbuf[0].ptr, (const uint32_t*)c.sbuf[1].ptr, c.sampled[2], c.global, nBins,
                                                   hostDetLog(minL), hostDetLog(maxL));
    PLR_CHECK_LAUNCH(c);
    return 0;
}
PLR_REGISTER_SHADER("preExposeLights.comp", launchPreExposeLights);
// Also the fast set's kernel: a single wave (preExposeLightsWave: lane-parallel histogram scan with wave reductions), the same device function the fused frame
// front runs in its last block. Stand-alone it is what band rendering launches behind the histogram all-reduce. Its output is two floats every later
// pass multiplies with: it keeps the exact set's arithmetic in both math modes on purpose.
PLR_REGISTER_SHADER_FAST("preExposeLights.comp", launchPreExposeLights);

int prepareExposureChain(const PassCtx* const* ctxs, ExposureChainPlan* out) {
    const PassCtx &reset = *ctxs[0], &comb = *ctxs[1], &expo = *ctxs[2];
    const uint32_t nBins = comb.specUint(0, 64u);
    if (nBins == 0 || nBins > (uint32_t)kFusedExposureMaxBins || reset.specUint(0, 64u) != nBins || (uint32_t)expo.specInt(0, 64) != nBins) return kUseGeneralKernel;
    if (!reset.hasSbuf(1) || !comb.hasSbuf(0) || !comb.hasSbuf(1) || !expo.hasSbuf(0) || !expo.hasSbuf(1) || !expo.hasSampled(2) || !expo.global) return kUseGeneralKernel;
    // one histogram buffer through the chain, whole-histogram dispatches, all tiles from tile 0 (not a band's partial combine)
    if (reset.sbuf[1].ptr != comb.sbuf[1].ptr || expo.sbuf[1].ptr != comb.sbuf[1].ptr || comb.sbuf[1].size < (size_t)nBins * 4u) return kUseGeneralKernel;
    if (reset.dispatch[0] * 64u < nBins || comb.dispatch[1] * 64u < nBins || comb.base[0] != 0 || comb.dispatch[0] == 0) return kUseGeneralKernel;
    const uint32_t nTiles = comb.dispatch[0];
    if (comb.sbuf[0].size < (size_t)nTiles * nBins * 4u || expo.sbuf[0].size < sizeof(LightBuffer) || expo.sampled[2].fmt != F_R11G11B10) return kUseGeneralKernel;
    const float minL = expo.specFloat(1, 1.f), maxL = expo.specFloat(2, 100.f);
    if (!(minL > 0.f) || !(maxL > 0.f)) return kUseGeneralKernel;
    ExposureScratch* scratch = (ExposureScratch*)comb.scratch(sizeof(ExposureScratch)); // zero-initialised by the backend, kept zero by the kernel
    if (!scratch) return comb.fail(-2, "histogramCombineTiles: cannot allocate scratch memory");
    out->perTile = (const uint32_t*)comb.sbuf[0].ptr; out->histogram = (uint32_t*)comb.sbuf[1].ptr; out->nBins = nBins; out->nTiles = nTiles;
    out->blocks = divUp(nTiles, kCombineTilesPerBlock); out->scratch = scratch; out->light = expo.sbuf[0].ptr; out->transmissionLut = expo.sampled[2]; out->global = expo.global;
    out->minLuminanceLog = hostDetLog(minL); out->maxLuminanceLog = hostDetLog(maxL);
    return 0;
}
static int launchFusedExposureChain(const PassCtx* const* ctxs, size_t count) {
    if (count != 3) return kUseGeneralKernel;
    ExposureChainPlan e;
    if (int rc = prepareExposureChain(ctxs, &e)) return rc;
    histogramCombineExposeKernel<<<e.blocks, 128, 0, ctxs[1]->stream>>>(e.perTile, e.histogram, e.nBins, e.nTiles, (ExposureScratch*)e.scratch, (LightBuffer*)e.light, e.transmissionLut,
                                                                        e.global, e.minLuminanceLog, e.maxLuminanceLog);
    PLR_CHECK_LAUNCH(*ctxs[1]);
    return 0;
}
int launchExposureChainAndPyramidTail(const ExposureChainPlan& e, const fasthiz::Plan& h, hipStream_t stream, const FusedCullParams* cull, int cullLevel) {
    // per device and per host thread's backend: set every time (a host call of a microsecond), not cached in a process-wide flag
    const void* kernel = cull ? (const void*)exposureChainAndPyramidTailKernel<true> : (const void*)exposureChainAndPyramidTailKernel<false>;
    if (hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 140 * 1024) != hipSuccess)
        return setLastError(-2, "exposure chain + pyramid tail: cannot raise the dynamic LDS limit");
    const uint32_t cullBlocks = cull ? divUp(cull->domainX * cull->domainY, 16u) : 0u;
    const size_t lds = cull ? std::max(h.tailLdsBytes, (size_t)kFusedCullMaxInstances * 4u) : h.tailLdsBytes;
    if (cull) exposureChainAndPyramidTailKernel<true><<<1u + e.blocks + cullBlocks, 1024, lds, stream>>>(e.perTile, e.histogram, e.nBins, e.nTiles, (ExposureScratch*)e.scratch, (LightBuffer*)e.light,
                                                                                                      e.transmissionLut, e.global, e.minLuminanceLog, e.maxLuminanceLog, h.tail, h.tailFirst,
                                                                                                      h.tailTexelsA, e.blocks, *cull, cullBlocks, cullLevel);
    else exposureChainAndPyramidTailKernel<false><<<1u + e.blocks, 1024, lds, stream>>>(e.perTile, e.histogram, e.nBins, e.nTiles, (ExposureScratch*)e.scratch, (LightBuffer*)e.light,
                                                                                       e.transmissionLut, e.global, e.minLuminanceLog, e.maxLuminanceLog, h.tail, h.tailFirst, h.tailTexelsA,
                                                                                       e.blocks, FusedCullParams{}, 0u, 0);
    const hipError_t err = hipGetLastError();
    return err == hipSuccess ? 0 : setLastError(-2, std::string("exposure chain + pyramid tail launch failed: ") + hipGetErrorString(err));
}
PLR_REGISTER_FUSION("histogramReset + histogramCombineTiles + preExposeLights", launchFusedExposureChain, "histogramReset.comp", "histogramCombineTiles.comp", "preExposeLights.comp");
// band rendering: reset + combine of the band's tiles (any first tile), one launch; the all-reduce callback and the exposure pass follow
int prepareResetCombine(const PassCtx* const* ctxs, ResetCombinePlan* out) {
    const PassCtx &reset = *ctxs[0], &comb = *ctxs[1];
    const uint32_t nBins = comb.specUint(0, 64u);
    if (nBins == 0 || nBins > (uint32_t)kFusedExposureMaxBins || reset.specUint(0, 64u) != nBins) return kUseGeneralKernel;
    if (!reset.hasSbuf(1) || !comb.hasSbuf(0) || !comb.hasSbuf(1) || reset.sbuf[1].ptr != comb.sbuf[1].ptr || comb.sbuf[1].size < (size_t)nBins * 4u) return kUseGeneralKernel;
    if (reset.dispatch[0] * 64u < nBins || comb.dispatch[1] * 64u < nBins || comb.dispatch[0] == 0) return kUseGeneralKernel;
    const uint32_t tile0 = comb.base[0], nTiles = comb.dispatch[0];
    if (comb.sbuf[0].size < (size_t)(tile0 + nTiles) * nBins * 4u) return kUseGeneralKernel;
    ExposureScratch* scratch = (ExposureScratch*)comb.scratch(sizeof(ExposureScratch)); // zero-initialised by the backend, kept zero by the kernel
    if (!scratch) return comb.fail(-2, "histogramCombineTiles: cannot allocate scratch memory");
    out->perTileBase = (const uint32_t*)comb.sbuf[0].ptr; out->perTile = out->perTileBase + (size_t)tile0 * nBins; out->histogram = (uint32_t*)comb.sbuf[1].ptr;
    out->nBins = nBins; out->nTiles = nTiles; out->blocks = divUp(nTiles, kCombineTilesPerBlock); out->scratch = scratch;
    return 0;
}
static int launchFusedResetCombine(const PassCtx* const* ctxs, size_t count) {
    if (count != 2) return kUseGeneralKernel;
    ResetCombinePlan r;
    if (int rc = prepareResetCombine(ctxs, &r)) return rc;
    histogramResetCombineKernel<<<r.blocks, 128, 0, ctxs[1]->stream>>>(r.perTile, r.histogram, r.nBins, r.nTiles, (ExposureScratch*)r.scratch);
    PLR_CHECK_LAUNCH(*ctxs[1]);
    return 0;
}
// launch 2 of a BAND's front (kernels_fast/fused_front.h, round 5): the blocks that finish the tiles' pyramid levels 4 and 5, the camera culling's blocks (a culling
// tile evaluates its one texel of level 4 from level 3, as in tileTailAndCullingKernel, kernels/sdfgi.hip) and the blocks of histogramReset + histogramCombineTiles over
// the band's tiles - three kinds of work that depend on launch 1 only, each block running the code of its own kernel.
// EXPOSE: a frame with a per-tile pyramid that is NOT partitioned (8K on one GPU: twelve levels, one more than the shader binds) - no all-reduce, the block that
// takes the last combine ticket runs the exposure as in the whole frame's launch 2.
template <bool EXPOSE>
__global__ __launch_bounds__(256) void bandFrontSecondKernel(fasthiz::TileTailParams t, uint32_t tailBlocks, FusedCullParams cull, uint32_t cullBlocks, const uint32_t* __restrict__ perTile,
                                                             uint32_t* __restrict__ histogram, uint32_t nBins, uint32_t nTiles, ExposureScratch* __restrict__ scratch, uint32_t combineBlocks,
                                                             LightBuffer* __restrict__ light, ImgView transmissionLut, const GlobalUbo* __restrict__ g, float minLuminanceLog, float maxLuminanceLog) {
    __shared__ uint32_t list[kFusedCullMaxInstances];
    __shared__ uint32_t waveTotals[4];
    __shared__ uint32_t base;
    __shared__ uint32_t totals[kFusedExposureMaxBins];
    __shared__ uint32_t isLast;
    __shared__ float term[EXPOSE ? kMaxExposureBins : 1];
    __shared__ uint32_t counted[EXPOSE ? kMaxExposureBins : 1];
    if (blockIdx.x < tailBlocks) { fasthiz::hizTileTailThread(t, (int)(blockIdx.x * 256u + threadIdx.x)); return; }
    if (blockIdx.x < tailBlocks + cullBlocks) {
        frustumAndTileCullingBlock<true, 256>(cull, blockIdx.x - tailBlocks, cullBlocks, list, waveTotals, &base, [&](vec2 uv) {
            const int x = clampi((int)floorf(saneCoord(uv.x * (float)t.w4)), t.w4), y = clampi((int)floorf(saneCoord(uv.y * (float)t.h4)), t.h4);
            const MinMax m = footprint<false>(2 * x, 2 * y, t.w3, t.h3, t.h3 & 1, t.w3 & 1, [&](int sx, int sy) { return t.level3[(size_t)sy * (size_t)t.w3 + (size_t)sx]; });
            return make_float2(m.mn, m.mx);
        });
        return;
    }
    histogramCombineExposeBlock<EXPOSE>(blockIdx.x - tailBlocks - cullBlocks, combineBlocks, perTile, histogram, nBins, nTiles, scratch, light, transmissionLut, g, minLuminanceLog,
                                        maxLuminanceLog, term, counted, totals, &isLast);
}
static int launchTileFrontSecond(const fasthiz::Plan& h, const FusedCullParams& cull, const uint32_t* perTile, uint32_t* histogram, uint32_t nBins, uint32_t nTiles, void* scratch,
                                 uint32_t combineBlocks, const ExposureChainPlan* expose, hipStream_t stream) {
    const fasthiz::TileTailParams& t = h.tileTail;
    const int n = (t.col4End - t.col4Begin) * (t.row4End - t.row4Begin) + (t.col5End - t.col5Begin) * (t.row5End - t.row5Begin);
    const uint32_t tailBlocks = n > 0 ? divUp((unsigned)n, 256u) : 0u, cullBlocks = divUp(cull.domainX * cull.domainY, 4u);
    const uint32_t grid = tailBlocks + cullBlocks + combineBlocks;
    if (expose) bandFrontSecondKernel<true><<<grid, 256, 0, stream>>>(t, tailBlocks, cull, cullBlocks, perTile, histogram, nBins, nTiles, (ExposureScratch*)scratch, combineBlocks,
                                                                       (LightBuffer*)expose->light, expose->transmissionLut, expose->global, expose->minLuminanceLog, expose->maxLuminanceLog);
    else bandFrontSecondKernel<false><<<grid, 256, 0, stream>>>(t, tailBlocks, cull, cullBlocks, perTile, histogram, nBins, nTiles, (ExposureScratch*)scratch, combineBlocks, nullptr, ImgView{},
                                                                 nullptr, 0.f, 0.f);
    const hipError_t err = hipGetLastError();
    return err == hipSuccess ? 0 : setLastError(-2, std::string("per-tile front, launch 2 failed: ") + hipGetErrorString(err));
}
int launchBandFrontSecond(const fasthiz::Plan& h, const FusedCullParams& cull, const ResetCombinePlan& r, hipStream_t stream) {
    return launchTileFrontSecond(h, cull, r.perTile, r.histogram, r.nBins, r.nTiles, r.scratch, r.blocks, nullptr, stream);
}
int launchTileFrontSecondWithExposure(const fasthiz::Plan& h, const FusedCullParams& cull, const ExposureChainPlan& e, hipStream_t stream) {
    return launchTileFrontSecond(h, cull, e.perTile, e.histogram, e.nBins, e.nTiles, e.scratch, e.blocks, &e, stream);
}
PLR_REGISTER_FUSION("histogramReset + histogramCombineTiles", launchFusedResetCombine, "histogramReset.comp", "histogramCombineTiles.comp");

// ------------------------------------------------------------------------------------------------
// tonemapping.comp:17-27 + tonemapping.inc:17-49 + colorConversion.inc:5-13 + dither.inc:6-12 + noise.inc:14-24.
// Streaming: 4 B in, 4 B out per pixel, 4 pixels (16 B) per lane. The output is 8-bit and the stated tolerance is
// +-1 LSB, so pow() uses the hardware v_log_f32 / v_exp_f32 pair (the det* routines would make this pass ALU bound).
PLR_DI float fastPow(float x, float y) { return __builtin_amdgcn_exp2f(y * __builtin_amdgcn_logf(x)); }

PLR_DI vec3 ACESFitted(vec3 color) {
    // the literal triples of ACESInputMat/ACESOutputMat act as rows after the transpose (tonemapping.inc:42,46)
    vec3 v(0.59719f * color.x + 0.35458f * color.y + 0.04823f * color.z, 0.07600f * color.x + 0.90834f * color.y + 0.01566f * color.z,
           0.02840f * color.x + 0.13383f * color.y + 0.83777f * color.z);
    const vec3 a = v * (v + 0.0245786f) - 0.000090537f;
    const vec3 b = v * (0.983729f * v + 0.4329510f) + 0.238081f;
    v = a / b;
    vec3 o(1.60475f * v.x + -0.53108f * v.y + -0.07367f * v.z, -0.10208f * v.x + 1.10813f * v.y + -0.00605f * v.z,
           -0.00327f * v.x + -0.07276f * v.y + 1.07602f * v.z);
    return vclamp(o, 0.f, 1.f);
}

PLR_DI float linearTosRGB1(float l) {
    const float lo = l * 12.92f;
    const float hi = fastPow(fabsf(l), 1.0f / 2.4f) * 1.055f - 0.055f;
    return l <= 0.0031308f ? lo : hi;
}

PLR_DI vec3 hash32(float qx, float qy) {
    const uint32_t UI0 = 1597334673u, UI1 = 3812015801u, UI2 = 2798796415u;
    uint32_t nx = (uint32_t)(int32_t)qx * UI0, ny = (uint32_t)(int32_t)qy * UI1, nz = (uint32_t)(int32_t)qx * UI2;
    const uint32_t m = nx ^ ny ^ nz;
    nx = m * UI0; ny = m * UI1; nz = m * UI2;
    const float UIF = 1.0f / (float)0xffffffffu;
    return vec3((float)nx, (float)ny, (float)nz) * UIF;
}

PLR_DI uint32_t tonemapPixel(uint32_t texel, int x, int y, float time) {
    const vec3 linearColor = unpackR11G11B10(texel);
    const vec3 t = ACESFitted(linearColor);
    vec3 s(linearTosRGB1(t.x), linearTosRGB1(t.y), linearTosRGB1(t.z));
    // ditherRGB8: hash32(uvec2(uv * g_time)) + hash32(uvec2((uv + vec2(165, 1292)) * g_time)) - 1, in 1/255 units
    vec3 noise = hash32((float)(uint32_t)((float)x * time), (float)(uint32_t)((float)y * time));
    noise += hash32((float)(uint32_t)(((float)x + 165.f) * time), (float)(uint32_t)(((float)y + 1292.f) * time));
    noise = noise - 1.f;
    noise = noise / 255.f;
    s = s + noise;
    // imageStore to the BGRA8 swapchain image: memory order B, G, R, A
    return encodeUnorm8(s.z) | (encodeUnorm8(s.y) << 8) | (encodeUnorm8(s.x) << 16) | (255u << 24);
}

template <bool BGRA>
__global__ __launch_bounds__(256) void tonemappingKernel(ImgView src, ImgView dst, const GlobalUbo* __restrict__ g, int coverW, int coverH, int yBase, int xBase) {
    const int x0 = xBase + (int)(blockIdx.x * 64u + (threadIdx.x & 63u)) * 4; // columns [xBase, coverW): xBase is a multiple of 8 (PassCtx::colSpan)
    const int y = yBase + (int)(blockIdx.y * 4u + (threadIdx.x >> 6));
    if (y >= coverH || x0 >= coverW) return;
    const float time = g->time;
    const uint32_t* srow = (const uint32_t*)src.ptr + (size_t)y * (size_t)src.w;
    uint32_t* drow = (uint32_t*)dst.ptr + (size_t)y * (size_t)dst.w;
    const int n = min(4, coverW - x0);
    uint32_t in[4], out[4];
    const bool vec = (n == 4) && ((src.w & 3) == 0) && ((dst.w & 3) == 0);
    if (vec) {
        const uint4 v = *(const uint4*)(srow + x0);
        in[0] = v.x; in[1] = v.y; in[2] = v.z; in[3] = v.w;
    } else {
        for (int i = 0; i < n; i++) in[i] = srow[x0 + i];
    }
#pragma unroll
    for (int i = 0; i < 4; i++) {
        if (i < n) {
            uint32_t p = tonemapPixel(in[i], x0 + i, y, time);
            if (!BGRA) p = (p & 0xff00ff00u) | ((p >> 16) & 0xffu) | ((p & 0xffu) << 16);
            out[i] = p;
        }
    }
    if (vec) *(uint4*)(drow + x0) = make_uint4(out[0], out[1], out[2], out[3]);
    else for (int i = 0; i < n; i++) drow[x0 + i] = out[i];
}

static int launchTonemapping(const PassCtx& c) {
    if (int rc = c.needGlobal()) return rc;
    if (int rc = c.needSampled(1, F_R11G11B10, "tonemapping imageIn")) return rc;
    if (int rc = c.needStorage(0, -1, "tonemapping imageOut")) return rc;
    const ImgView& src = c.sampled[1];
    const ImgView& dst = c.storage[0];
    if (dst.fmt != F_BGRA8 && dst.fmt != F_RGBA8) return c.fail(-4, "tonemapping imageOut must be BGRA8_uNorm or RGBA8");
    // invocations exist for dispatch*8 pixels; stores outside the target are dropped, fetches outside the source are
    // undefined in the reference, so the covered region is clipped to both images
    const PassCtx::ColSpan cs = c.colSpan(std::min(dst.w, src.w));
    const int coverW = cs.x1, xBase = cs.x0; // columns [xBase, coverW)
    const PassCtx::RowSpan rs = c.rowSpan(std::min(dst.h, src.h));
    const int coverH = rs.y1, y0 = rs.y0; // rows [y0, coverH)
    if (coverW <= xBase || coverH <= y0) return 0;
    const dim3 grid(divUp((unsigned)(coverW - xBase), 256u), divUp((unsigned)(coverH - y0), 4u));
    if (dst.fmt == F_BGRA8) tonemappingKernel<true><<<grid, 256, 0, c.stream>>>(src, dst, c.global, coverW, coverH, y0, xBase);
    else tonemappingKernel<false><<<grid, 256, 0, c.stream>>>(src, dst, c.global, coverW, coverH, y0, xBase);
    PLR_CHECK_LAUNCH(c);
    return 0;
}
PLR_REGISTER_SHADER("tonemapping.comp", launchTonemapping);

} // namespace plr
