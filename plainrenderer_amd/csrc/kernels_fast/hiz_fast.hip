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
// 55 MB of compulsory traffic in two launches.
#include "../backend.h"
#include "../device/shading_common.h"
#include "../device/hiz_common.h"
#include "../device/hiz_fast_device.h"

namespace plr {
namespace fasthiz {

template <int LEVELS, bool DOWNSCALE>
__global__ __launch_bounds__(256) void hizQuadKernel(QuadParams p) { hizQuadBlock<LEVELS, DOWNSCALE>(p, (int)blockIdx.x, (int)blockIdx.y); }

__global__ __launch_bounds__(1024) void hizTailKernel(HizParams p, int first, int texelsA) {
    extern __shared__ float2 hizTailLds[];
    hizTailBlock<1024>(p, first, texelsA, hizTailLds);
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
    // the whole pyramid only: a dispatch that covers part of the tile rows (band rendering) takes the general kernel
    const int tileRows = (int)divUp((unsigned)p.h[0], 32u);
    if (c.base[1] != 0 || (int)c.dispatch[1] < tileRows) return kUseGeneralKernel;
    const int texA = p.w[levels] * p.h[levels], texB = mipCount > levels + 1 ? p.w[levels + 1] * p.h[levels + 1] : 1;
    const size_t tailLds = (size_t)(texA + texB) * sizeof(float2);
    if (tailLds > 140 * 1024) return kUseGeneralKernel;
    QuadParams q{};
    q.depth = (const float*)depth.ptr; q.depthW = depth.w; q.depthH = depth.h;
    for (int l = 0; l < 4; l++) q.level[l] = p.level[l];
    q.levels = levels;
    if (down) {
        if (!down->hasStorage(0) || !down->hasSampled(1) || down->storage[0].fmt != F_R16F || down->sampled[1].ptr != depth.ptr) return kUseGeneralKernel;
        const ImgView& dst = down->storage[0];
        const PassCtx::RowSpan rs = down->rowSpan(dst.h);
        if (dst.w * 2 != depth.w || dst.h * 2 != depth.h || rs.y0 != 0 || rs.y1 != dst.h || (int)(down->dispatch[0] * 8u) < dst.w) return kUseGeneralKernel;
        q.halfDepth = (uint16_t*)dst.ptr; q.halfW = dst.w;
    }
    out->quad = q; out->tail = p;
    out->gridX = (int)divUp((unsigned)depth.w, 64u); out->gridY = (int)divUp((unsigned)depth.h, 64u);
    out->tailFirst = levels; out->tailTexelsA = texA; out->tailLdsBytes = tailLds; out->downscale = down != nullptr;
    return 0;
}

static int launchImpl(const PassCtx& c, const PassCtx* down) {
    Plan plan;
    if (int rc = prepare(c, down, &plan)) return rc;
    static bool ldsRaised = false;
    if (!ldsRaised) {
        if (hipFuncSetAttribute((const void*)hizTailKernel, hipFuncAttributeMaxDynamicSharedMemorySize, 140 * 1024) != hipSuccess) return kUseGeneralKernel;
        ldsRaised = true;
    }
    const dim3 grid((unsigned)plan.gridX, (unsigned)plan.gridY);
    if (down) hizQuadKernel<4, true><<<grid, 256, 0, c.stream>>>(plan.quad);
    else hizQuadKernel<4, false><<<grid, 256, 0, c.stream>>>(plan.quad);
    PLR_CHECK_LAUNCH(c);
    hizTailKernel<<<1, 1024, plan.tailLdsBytes, c.stream>>>(plan.tail, plan.tailFirst, plan.tailTexelsA);
    PLR_CHECK_LAUNCH(c);
    return 0;
}

} // namespace fasthiz

static int fasthiz_launch(const PassCtx& c) { return fasthiz::launchImpl(c, nullptr); }
PLR_REGISTER_SHADER_FAST("depthHiZPyramid.comp", fasthiz_launch);
static int fasthiz_fused_downscale(const PassCtx* const* ctxs, size_t count) { return count == 2 ? fasthiz::launchImpl(*ctxs[0], ctxs[1]) : kUseGeneralKernel; }
PLR_REGISTER_FUSION("depthHiZPyramid + depthDownscale", fasthiz_fused_downscale, "depthHiZPyramid.comp", "depthDownscale.comp");

} // namespace plr
