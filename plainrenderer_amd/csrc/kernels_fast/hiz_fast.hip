// Min/max depth pyramid for depth buffers whose sides are multiples of 8 (1920x1080, 3840x2160, ...), PLR_MATH_FAST set.
// Results are the same bits as kernels/hiz.hip (min / max of exact values: nothing is rounded).
//
// kernels/hiz.hip handles any size with per-level loops over LDS regions (scalar 4-byte depth loads, runtime div / mod indexing): 36 us at
// 4K = 19 % of the HBM rate for what is a pure streaming reduction. Here:
//   hizQuadKernel: a lane owns a 4x4 patch of depth texels (four 16-byte row loads): the four level-0 texels and the level-1 texel of
//                  the patch are reduced in registers; level 2 is a reduction over the 2x2 lanes of a DPP quad (quad_perm), level 3 over the
//                  2x2 quads of a 16-lane DPP row (row_ror) - no LDS, no barrier. A wave covers 32x32 depth texels, stores are 16 B (level 0)
//                  and 8 B (level 1) per lane. With pass fusion the same lane also writes the half-resolution depth of depthDownscale.comp
//                  (texel (2x, 2y) of its patch as a half float) and that pass's launch disappears.
//   hizTailKernel: one 1024-thread block finishes levels 4.. out of LDS (level 4 of the 4K frame is 120 x 67 texels = 63 KB).
//   hizTileTailKernel: six-level per-tile pyramids (band rendering, 8K): levels 4 and 5 of the launch's tile rows straight from level 3 (device/hiz_fast_device.h).
// 55 MB of compulsory traffic in two launches.
// Every other size (odd sides, sides that are no multiple of 8: 322 x 182 after a window resize) takes the any-size block bodies of device/hiz_any_size.h, launched
// from here: the fast set covers every size itself, the exact set's launch path (kernels/hiz.hip) is reached through PLR_MATH_EXACT only.
#include "../backend.h"
#include "../device/shading_common.h"
#include "../device/hiz_common.h"
#include "../device/hiz_fast_device.h"
#include "../device/hiz_any_size.h"

namespace plr {
namespace fasthiz {

template <int LEVELS, bool DOWNSCALE>
__global__ __launch_bounds__(256) void hizQuadKernel(QuadParams p) { hizQuadBlock<LEVELS, DOWNSCALE>(p, (int)blockIdx.x + p.tileX0, (int)blockIdx.y + p.tileY0); }

__global__ __launch_bounds__(256) void hizTileTailKernel(TileTailParams p) { hizTileTailThread(p, (int)(blockIdx.x * 256u + threadIdx.x)); }

__global__ __launch_bounds__(1024) void hizTailKernel(HizParams p, int first, int texelsA) {
    extern __shared__ float2 hizTailLds[];
    hizTailBlock<1024>(p, first, texelsA, hizTailLds);
}

__global__ __launch_bounds__(256) void hizAnySizeBaseKernel(HizParams p) {
    extern __shared__ float2 hizAnySizeLds[];
    hizBaseBlock(p, hizAnySizeLds);
}

__global__ __launch_bounds__(1024) void hizAnySizeTailKernel(HizParams p) {
    __shared__ float2 bufA[32 * 32];
    __shared__ float2 bufB[32 * 32];
    hizTailAnyBlock(p, bufA, bufB);
}

static int launchAnySize(const PassCtx& c) {
    return hizLaunchAnySize(c, [&](dim3 grid, size_t ldsBytes, const HizParams& p) { hizAnySizeBaseKernel<<<grid, 256, ldsBytes, c.stream>>>(p); },
                            [&](const HizParams& p) { hizAnySizeTailKernel<<<1, 1024, 0, c.stream>>>(p); });
}

int prepare(const PassCtx& c, const PassCtx* down, Plan* out) {
    const int mipCount = c.specInt(0, 0);
    if (mipCount < 5 || mipCount > kHizMaxLevels || !c.hasSampled(13) || c.sampled[13].fmt != F_D32) return kUseGeneralKernel;
    const ImgView& depth = c.sampled[13];
    if (c.specInt(1, 0) != depth.w || c.specInt(2, 0) != depth.h || (depth.w & 7) || (depth.h & 7)) return kUseGeneralKernel;
    const int levels = ((depth.w | depth.h) & 15) ? 3 : 4; // levels made without LDS (every texel of them has an even-sized source)
    HizParams p{};
    p.count = mipCount;
    const int unused = kHizMaxLevels - mipCount;
    int sw = depth.w, sh = depth.h;
    for (int l = 0; l < mipCount; l++) {
        const int b = l + unused;
        if (!c.hasStorage(b) || c.storage[b].fmt != F_RG32F) return kUseGeneralKernel;
        const int w = std::max(sw / 2, 1), h = std::max(sh / 2, 1);
        if (c.storage[b].w != w || c.storage[b].h != h) return kUseGeneralKernel;
        p.level[l] = (float2*)c.storage[b].ptr; p.w[l] = w; p.h[l] = h;
        sw = w; sh = h;
    }
    // tile rows (64 depth rows each) of the recorded dispatch. The whole chain needs every tile; a SIX-level pyramid is per tile (band rendering, and
    // frames beyond the shader's 11 levels: frame_pipeline.cpp perTilePyramid) and can be built for any range of tile rows
    const int tileRows = (int)divUp((unsigned)p.h[0], 32u), tileCols = (int)divUp((unsigned)p.w[0], 32u);
    const bool wholeWidth = c.base[0] == 0 && (int)c.dispatch[0] >= tileCols; // (tile rendering: a range of tile columns too)
    const bool whole = c.base[1] == 0 && (int)c.dispatch[1] >= tileRows && wholeWidth;
    const bool perTile = mipCount == 6 && levels == 4;
    if (!whole && !perTile) return kUseGeneralKernel;
    int tile0 = 0, tile1 = tileRows, tileX0 = 0, tileX1 = tileCols;
    if (!whole) {
        const PassCtx::RowSpan rs = c.base[1] == 0 && (int)c.dispatch[1] >= tileRows ? PassCtx::RowSpan{0, tileRows} : c.rowSpan(tileRows, 1);
        tile0 = rs.y0; tile1 = rs.y1;
        if (!wholeWidth) { const PassCtx::ColSpan cs = c.colSpan(tileCols, 1); tileX0 = cs.x0; tileX1 = cs.x1; }
        if (tile1 <= tile0 || tileX1 <= tileX0) return kUseGeneralKernel; // nothing to do: let the general launcher say so
    }
    const int texA = p.w[levels] * p.h[levels], texB = mipCount > levels + 1 ? p.w[levels + 1] * p.h[levels + 1] : 1;
    const size_t tailLds = (size_t)(texA + texB) * sizeof(float2);
    if (!perTile && tailLds > 140 * 1024) return kUseGeneralKernel;
    QuadParams q{};
    q.depth = (const float*)depth.ptr; q.depthW = depth.w; q.depthH = depth.h;
    for (int l = 0; l < 4; l++) q.level[l] = p.level[l];
    q.levels = levels;
    // quad blocks: the tile rows of the dispatch; per tile, one more when a 3-row footprint of level 4 / 5 reaches into the next tile's level-3 rows
    int quad0 = tile0, quad1 = tile1, quadX0 = tileX0, quadX1 = tileX1;
    if (perTile && ((p.h[3] | p.h[4]) & 1)) quad1 = std::min(tile1 + 1, tileRows);
    if (perTile && ((p.w[3] | p.w[4]) & 1)) quadX1 = std::min(tileX1 + 1, tileCols);
    q.halfCol0 = 0; q.halfCol1 = 0x7fffffff;
    if (down) {
        if (!down->hasStorage(0) || !down->hasSampled(1) || down->storage[0].fmt != F_R16F || down->sampled[1].ptr != depth.ptr) return kUseGeneralKernel;
        const ImgView& dst = down->storage[0];
        const PassCtx::RowSpan rs = down->rowSpan(dst.h);
        const PassCtx::ColSpan dcs = down->colSpan(dst.w);
        if (dst.w * 2 != depth.w || dst.h * 2 != depth.h || dcs.x1 <= dcs.x0 || rs.y1 <= rs.y0) return kUseGeneralKernel;
        if (!perTile && (dcs.x0 != 0 || dcs.x1 != dst.w)) return kUseGeneralKernel;
        if (!perTile && (rs.y0 != 0 || rs.y1 != dst.h)) return kUseGeneralKernel;
        q.halfDepth = (uint16_t*)dst.ptr; q.halfW = dst.w;
        q.halfRow0 = rs.y0; q.halfRow1 = rs.y1;
        q.halfCol0 = dcs.x0; q.halfCol1 = dcs.x1;
        // the quad blocks also cover the tile rows of the half-resolution rows the downscale pass was asked for (32 half-res rows per tile), and its columns
        quad0 = std::min(quad0, rs.y0 / 32); quad1 = std::max(quad1, std::min((rs.y1 + 31) / 32, tileRows));
        quadX0 = std::min(quadX0, dcs.x0 / 32); quadX1 = std::max(quadX1, std::min((dcs.x1 + 31) / 32, tileCols));
    }
    q.tileY0 = quad0; q.tileX0 = quadX0;
    out->quad = q; out->tail = p;
    out->gridX = quadX1 - quadX0; out->gridY = quad1 - quad0;
    out->tailFirst = levels; out->tailTexelsA = texA; out->tailLdsBytes = tailLds; out->downscale = down != nullptr;
    out->perTile = perTile;
    if (perTile) {
        TileTailParams t{};
        t.level3 = p.level[3]; t.level4 = p.level[4]; t.level5 = p.level[5];
        t.w3 = p.w[3]; t.h3 = p.h[3]; t.w4 = p.w[4]; t.h4 = p.h[4]; t.w5 = p.w[5]; t.h5 = p.h[5];
        // a tile owns 2 rows of level 4 and 1 row of level 5 (kernels/hiz.hip: lo = tile * (32 >> level))
        t.row4Begin = std::min(2 * tile0, t.h4); t.row4End = std::min(2 * tile1, t.h4);
        t.row5Begin = std::min(tile0, t.h5); t.row5End = std::min(tile1, t.h5);
        t.col4Begin = std::min(2 * tileX0, t.w4); t.col4End = std::min(2 * tileX1, t.w4);
        t.col5Begin = std::min(tileX0, t.w5); t.col5End = std::min(tileX1, t.w5);
        out->tileTail = t;
    }
    return 0;
}

int launchQuadBlocks(const PassCtx& c, const Plan& plan) {
    const dim3 grid((unsigned)plan.gridX, (unsigned)plan.gridY);
    if (plan.downscale) hizQuadKernel<4, true><<<grid, 256, 0, c.stream>>>(plan.quad);
    else hizQuadKernel<4, false><<<grid, 256, 0, c.stream>>>(plan.quad);
    PLR_CHECK_LAUNCH(c);
    return 0;
}

static int launchImpl(const PassCtx& c, const PassCtx* down) {
    Plan plan;
    if (int rc = prepare(c, down, &plan)) {
        // not a size / configuration of the register-quad kernels: the any-size path (alone - a fused downscale member is launched by its own fast kernel then)
        if (rc == kUseGeneralKernel && !down) return launchAnySize(c);
        return rc;
    }
    // per device and per host thread's backend: set every time (a host call of a microsecond), not cached in a process-wide flag
    if (!plan.perTile && hipFuncSetAttribute((const void*)hizTailKernel, hipFuncAttributeMaxDynamicSharedMemorySize, 140 * 1024) != hipSuccess) return kUseGeneralKernel;
    if (int rc = launchQuadBlocks(c, plan)) return rc;
    if (plan.perTile) {
        const TileTailParams& t = plan.tileTail;
        const int n = (t.col4End - t.col4Begin) * (t.row4End - t.row4Begin) + (t.col5End - t.col5Begin) * (t.row5End - t.row5Begin);
        if (n > 0) hizTileTailKernel<<<divUp((unsigned)n, 256u), 256, 0, c.stream>>>(t);
    } else hizTailKernel<<<1, 1024, plan.tailLdsBytes, c.stream>>>(plan.tail, plan.tailFirst, plan.tailTexelsA);
    PLR_CHECK_LAUNCH(c);
    return 0;
}

} // namespace fasthiz

static int fasthiz_launch(const PassCtx& c) { return fasthiz::launchImpl(c, nullptr); }
PLR_REGISTER_SHADER_FAST("depthHiZPyramid.comp", fasthiz_launch);
static int fasthiz_fused_downscale(const PassCtx* const* ctxs, size_t count) { return count == 2 ? fasthiz::launchImpl(*ctxs[0], ctxs[1]) : kUseGeneralKernel; }
PLR_REGISTER_FUSION("depthHiZPyramid + depthDownscale", fasthiz_fused_downscale, "depthHiZPyramid.comp", "depthDownscale.comp");

} // namespace plr
