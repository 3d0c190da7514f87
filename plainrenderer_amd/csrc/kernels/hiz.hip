// Min/max depth pyramid (HiZ) for gfx950: depthHiZPyramid.comp (resources/shaders/, host RenderFrontend.cpp:804-838,1770-1827), PLR_MATH_EXACT set.
// The block bodies and the launch plan serve any depth-buffer size and are shared with the fast set: device/hiz_any_size.h.
#include "../backend.h"
#include "../device/shading_common.h"
#include "../device/hiz_common.h"
#include "../device/hiz_any_size.h"

namespace plr {

__global__ __launch_bounds__(256) void hizBaseKernel(HizParams p) {
    extern __shared__ float2 hizLds[];
    hizBaseBlock(p, hizLds);
}

__global__ __launch_bounds__(1024) void hizTailKernel(HizParams p) {
    __shared__ float2 bufA[32 * 32];
    __shared__ float2 bufB[32 * 32];
    hizTailAnyBlock(p, bufA, bufB);
}

static int launchDepthHiZPyramid(const PassCtx& c) {
    return hizLaunchAnySize(c, [&](dim3 grid, size_t ldsBytes, const HizParams& p) { hizBaseKernel<<<grid, 256, ldsBytes, c.stream>>>(p); },
                            [&](const HizParams& p) { hizTailKernel<<<1, 1024, 0, c.stream>>>(p); });
}
PLR_REGISTER_SHADER("depthHiZPyramid.comp", launchDepthHiZPyramid);


// ------------------------------------------------------------------------------------------------ depthPyramidApex.comp (band rendering; no reference shader)
// A band builds per-tile pyramids (six levels): nobody computes the apex lightMatrix.comp:76-78 reads. This pass reduces the rows of a pyramid
// level that belong to the band - .r = min, .g = max, the pyramid's own rule (depthHiZPyramid.comp:52-124: the sky is already mapped out of .r
// by level 0) - into a 1 x 1 RG32F image; the bands' results are then combined by an all-reduce (min on .r, max on .g: SURVEY 8e, collective 2).
// min / max are associative and exact, so the combined value equals the apex of the unpartitioned chain bit for bit.
// Bindings: sampled 0 = pyramid level (RG32F), storage 1 = apex (RG32F, 1 x 1); the dispatch's rows are texel rows of that level; its x range (base, count) is
// the texel columns of a TILE (round 5), a count of one workgroup from column 0 means whole rows.
__global__ __launch_bounds__(256) void depthPyramidApexKernel(ImgView level, int row0, int row1, int col0, int col1, float2* __restrict__ apex) {
    __shared__ float smin[4], smax[4];
    const float2* t = (const float2*)level.ptr;
    const int cols = col1 - col0, n = (row1 - row0) * cols;
    float mn = __builtin_huge_valf(), mx = -__builtin_huge_valf();
    for (int i = (int)threadIdx.x; i < n; i += 256) {
        const float2 v = t[(size_t)(row0 + i / cols) * (size_t)level.w + (size_t)(col0 + i % cols)];
        mn = fminf(mn, v.x);
        mx = fmaxf(mx, v.y);
    }
    for (int off = 32; off > 0; off >>= 1) {
        mn = fminf(mn, __shfl_xor(mn, off));
        mx = fmaxf(mx, __shfl_xor(mx, off));
    }
    if ((threadIdx.x & 63u) == 0u) { smin[threadIdx.x >> 6] = mn; smax[threadIdx.x >> 6] = mx; }
    __syncthreads();
    if (threadIdx.x == 0) apex[0] = make_float2(fminf(fminf(smin[0], smin[1]), fminf(smin[2], smin[3])), fmaxf(fmaxf(smax[0], smax[1]), fmaxf(smax[2], smax[3])));
}
static int launchDepthPyramidApex(const PassCtx& c) {
    if (int rc = c.needSampled(0, F_RG32F, "depthPyramidApex pyramid level")) return rc;
    if (int rc = c.needStorage(1, F_RG32F, "depthPyramidApex apex")) return rc;
    const ImgView& level = c.sampled[0];
    const PassCtx::RowSpan rs = c.rowSpan(level.h, 1);
    if (rs.y1 <= rs.y0) return c.fail(-1, "depthPyramidApex: no rows to reduce");
    // columns: whole rows, unless the execution says that its x range is a range of texel columns of the level (tile rendering; push constant, 4 bytes, non-zero).
    // (Until round 5 "base 0, count <= 1" was read as whole rows: a tile one texel of the level wide - 64 pixels - at column 0 of a wider frame reduced texels the
    //  GPU never built. ADVICE r05)
    int columnRange = 0;
    if (c.push.size() >= 4) std::memcpy(&columnRange, c.push.data(), 4);
    const PassCtx::ColSpan cs = columnRange ? c.colSpan(level.w, 1) : PassCtx::ColSpan{0, level.w};
    if (cs.x1 <= cs.x0) return c.fail(-1, "depthPyramidApex: no columns to reduce");
    depthPyramidApexKernel<<<1, 256, 0, c.stream>>>(level, rs.y0, rs.y1, cs.x0, cs.x1, (float2*)c.storage[1].ptr);
    PLR_CHECK_LAUNCH(c);
    return 0;
}
PLR_REGISTER_SHADER("depthPyramidApex.comp", launchDepthPyramidApex);
PLR_REGISTER_SHADER_FAST("depthPyramidApex.comp", launchDepthPyramidApex); // comparisons only: one kernel serves both math modes

} // namespace plr
