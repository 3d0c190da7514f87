// Device code of the PLR_MATH_FAST depth pyramid (kernels_fast/hiz_fast.hip), as functions of a block index so that other launches can host
// these blocks beside their own (the exposure chain and the pyramid are independent: kernels_fast/histogram_fast.hip runs the per-tile
// histogram and the pyramid's quad blocks in one launch, kernels/exposure_tonemap.hip the exposure chain and the pyramid's tail).
#pragma once
#include "shading_common.h"
#include "hiz_common.h"

namespace plr {
namespace fasthiz {

PLR_DI float dppf(float v, int ctrl) {
    // all rows / banks enabled, bound_ctrl off: a lane whose source lane is invalid keeps `v` (never happens for the permutes used here)
    switch (ctrl) {
        case 0xB1: return u2f((uint32_t)__builtin_amdgcn_update_dpp((int)f2u(v), (int)f2u(v), 0xB1, 0xF, 0xF, false));   // quad_perm [1,0,3,2]
        case 0x4E: return u2f((uint32_t)__builtin_amdgcn_update_dpp((int)f2u(v), (int)f2u(v), 0x4E, 0xF, 0xF, false));   // quad_perm [2,3,0,1]
        case 0x124: return u2f((uint32_t)__builtin_amdgcn_update_dpp((int)f2u(v), (int)f2u(v), 0x124, 0xF, 0xF, false)); // row_ror:4
        default: return u2f((uint32_t)__builtin_amdgcn_update_dpp((int)f2u(v), (int)f2u(v), 0x128, 0xF, 0xF, false));   // row_ror:8
    }
}

struct QuadParams {
    const float* depth;
    int depthW, depthH;
    float2* level[4];
    uint16_t* halfDepth; // depthDownscale.comp's target (fused) or null
    int halfW;
    int levels;          // 4: both sides are multiples of 16; 3: multiples of 8 (1920 x 1080), level 3 is then the tail's first level (its 3-wide odd-size footprints need LDS)
    int tileY0;          // first 64-row tile row of the launch (band rendering / per-tile pyramids: a launch covers the tile rows it is asked for)
    int tileX0;          // first 64-column tile column of the launch (tile rendering)
    int halfRow0, halfRow1; // rows of halfDepth this launch may write (the depthDownscale execution's rows)
    int halfCol0, halfCol1; // and its columns (tile rendering: the downscale's column span; whole rows otherwise)
};

// min / max contribution of a (min, max) texel to the level above (depthHiZPyramid.comp:95-110): a texel whose max is 0 is all sky and must not
// pull the minimum down
PLR_DI float minTerm(float mn, float mx) { return mn + (mx == 0.f ? 1.f : 0.f); }

// one 256-thread block = 64x64 depth texels; (bx, by) = the block's tile
template <int LEVELS, bool DOWNSCALE>
PLR_DI void hizQuadBlock(const QuadParams& p, int bx, int by) {
    const int lane = (int)(threadIdx.x & 63u), wave = (int)(threadIdx.x >> 6);
    // lane bits: [1:0] position in a 2x2 quad of lanes, [3:2] position of the quad in a 16-lane row (2x2 quads), [5:4] position of the row in the wave
    const int x8 = (lane & 1) | ((lane >> 1) & 2) | ((lane >> 2) & 4), y8 = ((lane >> 1) & 1) | ((lane >> 2) & 2) | ((lane >> 3) & 4);
    const int px = bx * 64 + (wave & 1) * 32 + x8 * 4, py = by * 64 + (wave >> 1) * 32 + y8 * 4;
    const bool active = px < p.depthW && py < p.depthH; // sides are multiples of 8 << (levels - 3): a 4x4 patch (and every coarser texel made here) is inside or outside as a whole
    float mn1 = 1.f, mx1 = 0.f;
    if (active) {
        float4 r[4];
#pragma unroll
        for (int i = 0; i < 4; i++) r[i] = *(const float4*)(p.depth + (size_t)(py + i) * (size_t)p.depthW + (size_t)px);
        // level 0: 2x2 depth texels each; a depth of 0 is sky and counts as 1 for the minimum (:84-93)
        float mn0[2][2], mx0[2][2];
#pragma unroll
        for (int j = 0; j < 2; j++) {
            const float4 a = r[2 * j], b = r[2 * j + 1];
            const float d[2][4] = {{a.x, a.y, b.x, b.y}, {a.z, a.w, b.z, b.w}};
#pragma unroll
            for (int i = 0; i < 2; i++) {
                float mn = 1.f, mx = 0.f;
#pragma unroll
                for (int k = 0; k < 4; k++) { mn = __builtin_fminf(mn, d[i][k] + (d[i][k] == 0.f ? 1.f : 0.f)); mx = __builtin_fmaxf(mx, d[i][k]); }
                mn0[j][i] = mn; mx0[j][i] = mx;
            }
            // two level-0 texels of one row: 16 bytes
            *(float4*)(p.level[0] + (size_t)(py / 2 + j) * (size_t)(p.depthW / 2) + (size_t)(px / 2)) = make_float4(mn0[j][0], mx0[j][0], mn0[j][1], mx0[j][1]);
        }
#pragma unroll
        for (int j = 0; j < 2; j++)
#pragma unroll
            for (int i = 0; i < 2; i++) { mn1 = __builtin_fminf(mn1, minTerm(mn0[j][i], mx0[j][i])); mx1 = __builtin_fmaxf(mx1, mx0[j][i]); }
        p.level[1][(size_t)(py / 4) * (size_t)(p.depthW / 4) + (size_t)(px / 4)] = make_float2(mn1, mx1);
        if (DOWNSCALE) {
            // depthDownscale.comp:12-20: half-res texel (x, y) = depth texel (2x, 2y), stored as a half float
            uint16_t* h0 = p.halfDepth + (size_t)(py / 2) * (size_t)p.halfW + (size_t)(px / 2);
            const bool inCols = px / 2 >= p.halfCol0 && px / 2 < p.halfCol1; // (column spans are multiples of 8: the two texels of a lane are inside or outside together)
            if (inCols && py / 2 >= p.halfRow0 && py / 2 < p.halfRow1) *(uint32_t*)h0 = floatToHalfBits(r[0].x) | (floatToHalfBits(r[0].z) << 16);
            if (inCols && py / 2 + 1 >= p.halfRow0 && py / 2 + 1 < p.halfRow1) *(uint32_t*)(h0 + p.halfW) = floatToHalfBits(r[2].x) | (floatToHalfBits(r[2].z) << 16);
        }
    }
    // level 2: the four lanes of a quad hold the four level-1 texels of one level-2 texel. An inactive lane contributes the neutral pair
    // (1, 0) = the values the shader's accumulators start from.
    float a = active ? minTerm(mn1, mx1) : 1.f, b = active ? mx1 : 0.f;
    a = __builtin_fminf(a, dppf(a, 0xB1)); b = __builtin_fmaxf(b, dppf(b, 0xB1));
    a = __builtin_fminf(a, dppf(a, 0x4E)); b = __builtin_fmaxf(b, dppf(b, 0x4E));
    const float mn2 = __builtin_fminf(1.f, a), mx2 = b;
    if (active && (lane & 3) == 0) p.level[2][(size_t)(py / 8) * (size_t)(p.depthW / 8) + (size_t)(px / 8)] = make_float2(mn2, mx2);
    if (LEVELS >= 4 && p.levels >= 4) {
        // level 3: the four quads of a 16-lane row
        float c = active ? minTerm(mn2, mx2) : 1.f, d = active ? mx2 : 0.f;
        c = __builtin_fminf(c, dppf(c, 0x124)); d = __builtin_fmaxf(d, dppf(d, 0x124));
        c = __builtin_fminf(c, dppf(c, 0x128)); d = __builtin_fmaxf(d, dppf(d, 0x128));
        if (active && (lane & 15) == 0) p.level[3][(size_t)(py / 16) * (size_t)(p.depthW / 16) + (size_t)(px / 16)] = make_float2(__builtin_fminf(1.f, c), d);
    }
}

// levels [first, count) by one block of NT threads: the first from global memory (written by the quad blocks of an earlier launch), the others
// out of LDS (lds: texelsA + texels of level first + 1 float2 entries)
template <int NT>
PLR_DI void hizTailBlock(const HizParams& p, int first, int texelsA, float2* lds) {
    float2* bufA = lds;
    float2* bufB = lds + texelsA;
    const int t = threadIdx.x;
    {
        // the first tail level reads global memory: all of a thread's footprints are fetched before anything is stored, so the loads of its
        // (up to) eight texels are in flight together instead of one texel's after the other's
        const int l = first;
        const int sw = p.w[l - 1], sh = p.h[l - 1], w = p.w[l], h = p.h[l], n = w * h;
        const float2* __restrict__ srcG = p.level[l - 1];
        float2* __restrict__ dstG = p.level[l];
        for (int base = 0; base < n; base += NT * 8) {
            float2 v[8];
#pragma unroll
            for (int k = 0; k < 8; k++) {
                const int i = base + k * NT + t;
                v[k] = make_float2(1.f, 0.f);
                if (i < n) {
                    const int x = i % w, y = i / w;
                    const MinMax m = footprint<false>(2 * x, 2 * y, sw, sh, sh & 1, sw & 1, [&](int sx, int sy) { return srcG[(size_t)sy * sw + sx]; });
                    v[k] = make_float2(m.mn, m.mx);
                }
            }
#pragma unroll
            for (int k = 0; k < 8; k++) {
                const int i = base + k * NT + t;
                if (i < n) { bufA[i] = v[k]; dstG[i] = v[k]; }
            }
        }
        __syncthreads();
    }
    for (int l = first + 1; l < p.count; l++) {
        const int sw = p.w[l - 1], sh = p.h[l - 1];
        const int w = p.w[l], h = p.h[l];
        float2* dst = ((l - first) & 1) ? bufB : bufA;
        const float2* srcL = ((l - first) & 1) ? bufA : bufB;
        for (int i = t; i < w * h; i += NT) {
            const int x = i % w, y = i / w;
            const MinMax m = footprint<false>(2 * x, 2 * y, sw, sh, sh & 1, sw & 1, [&](int sx, int sy) { return srcL[sy * sw + sx]; });
            const float2 v = make_float2(m.mn, m.mx);
            dst[i] = v;
            p.level[l][i] = v;
        }
        __syncthreads();
    }
}


// ---- per-tile pyramids (band rendering; frames whose full chain exceeds the shader's 11 levels): six levels, no chain tail. Levels 4 and 5 of the
// tile rows [tileY0, tileY1) from level 3 in global memory (written by the quad blocks of an earlier launch), by as many blocks as it takes.
// A level-4 texel is the footprint of level 3, a level-5 texel the footprint of level-4 footprints - recomputed, not re-read: the launch has no
// grid-wide barrier, min / max are exact, so the bits are those of the level-by-level evaluation (kernels/hiz.hip). Footprints follow the
// reference's rule (depthHiZPyramid.comp:52-124): 3 rows / columns wherever the SOURCE level has an odd size.
struct TileTailParams {
    const float2* level3;
    float2* level4;
    float2* level5;
    int w3, h3, w4, h4, w5, h5;
    int row4Begin, row4End, row5Begin, row5End; // rows of levels 4 / 5 owned by the launch's tile rows
    int col4Begin, col4End, col5Begin, col5End; // and the columns owned by its tile columns (tile rendering; whole rows otherwise)
};
PLR_DI void hizTileTailThread(const TileTailParams& p, int i) {
    const int c4 = p.col4End - p.col4Begin, c5 = p.col5End - p.col5Begin;
    const int n4 = c4 * (p.row4End - p.row4Begin), n5 = c5 * (p.row5End - p.row5Begin);
    auto level4At = [&](int x, int y) {
        const MinMax m = footprint<false>(2 * x, 2 * y, p.w3, p.h3, p.h3 & 1, p.w3 & 1, [&](int sx, int sy) { return p.level3[(size_t)sy * p.w3 + sx]; });
        return make_float2(m.mn, m.mx);
    };
    if (i < n4) {
        const int x = p.col4Begin + i % c4, y = p.row4Begin + i / c4;
        p.level4[(size_t)y * p.w4 + x] = level4At(x, y);
    } else if (i < n4 + n5) {
        const int j = i - n4, x = p.col5Begin + j % c5, y = p.row5Begin + j / c5;
        const MinMax m = footprint<false>(2 * x, 2 * y, p.w4, p.h4, p.h4 & 1, p.w4 & 1, [&](int sx, int sy) { return level4At(sx, sy); });
        p.level5[(size_t)y * p.w5 + x] = make_float2(m.mn, m.mx);
    }
}

// everything a launch needs to host the pyramid's blocks (filled by fasthiz::prepare, kernels_fast/hiz_fast.hip)
struct Plan {
    QuadParams quad;
    HizParams tail;
    int gridX = 0, gridY = 0; // quad blocks
    int tailFirst = 4, tailTexelsA = 0;
    size_t tailLdsBytes = 0;
    bool downscale = false;
    bool perTile = false;     // six-level per-tile pyramid: tileTail replaces the chain tail, the quad blocks start at quad.tileY0
    TileTailParams tileTail;
};
// c: the depthHiZPyramid execution; down: the depthDownscale execution fused into it, or null. 0 / kUseGeneralKernel
int prepare(const PassCtx& c, const PassCtx* down, Plan* out);
// launch 1 of a plan: the quad blocks (levels 0..3 and, with a downscale pass fused in, the half-resolution depth); 0 / < 0 (kernels_fast/hiz_fast.hip)
int launchQuadBlocks(const PassCtx& c, const Plan& plan);

} // namespace fasthiz
} // namespace plr
