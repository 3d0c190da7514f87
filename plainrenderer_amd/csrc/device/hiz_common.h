// Shared by the two HiZ pyramid builders (kernels/hiz.hip: any size; kernels_fast/hiz_fast.hip: sizes divisible by 8 / 16).
#pragma once
#include "shading_common.h"

namespace plr {

constexpr int kHizMaxLevels = 11; // shader limit (depthHiZPyramid.comp:16-20)
constexpr int kHizBaseLevels = 6;

struct HizParams {
    float2* level[kHizMaxLevels];
    int w[kHizMaxLevels], h[kHizMaxLevels];
    const float* depth;
    int depthW, depthH;
    int count;     // pyramid levels
    int tileY0;    // first 32x32 mip-0 tile row of the launch (dispatch base)
    int tileX0;    // first tile column of the launch (tile rendering: PassCtx::colSpan in tiles)
    int baseCount; // levels produced by hizBaseKernel
    int ldsA;      // texels of hizBaseKernel's first LDS buffer
};

struct MinMax { float mn, mx; };

// depthHiZPyramid.comp:52-124 for one destination texel; fetch(x, y) returns the clamped source texel as (min, max)
// (for the depth buffer both components are the depth value).
template <bool FROM_DEPTH, class Fetch>
PLR_DI MinMax footprint(int ulx, int uly, int srcW, int srcH, bool extraRow, bool extraColumn, Fetch fetch) {
    float depthMin = 1.f, depthMax = 0.f;
    auto acc = [&](int x, int y, bool corner) {
        const float2 t = fetch(min(x, srcW - 1), min(y, srcH - 1));
        if (FROM_DEPTH) {
            if (corner) depthMin = gmin(depthMin, t.x * (t.x == 0.f ? 1.f : 0.f)); // sic, :114
            else depthMin = gmin(depthMin, t.x + (t.x == 0.f ? 1.f : 0.f));
            depthMax = gmax(depthMax, t.x);
        } else {
            depthMin = gmin(depthMin, t.x + (t.y == 0.f ? 1.f : 0.f));
            depthMax = gmax(depthMax, t.y);
        }
    };
    acc(ulx, uly, false); acc(ulx + 1, uly, false); acc(ulx, uly + 1, false); acc(ulx + 1, uly + 1, false);
    if (extraRow) { acc(ulx, uly + 2, false); acc(ulx + 1, uly + 2, false); }
    if (extraColumn) { acc(ulx + 2, uly, false); acc(ulx + 2, uly + 1, false); }
    if (extraRow && extraColumn) acc(ulx + 2, uly + 2, true);
    return {depthMin, depthMax};
}

} // namespace plr
