// The min/max depth pyramid for ANY depth-buffer size (odd sides, sides below one tile): block bodies and launch plan, instantiated by both kernel sets -
// kernels/hiz.hip (PLR_MATH_EXACT) and kernels_fast/hiz_fast.hip, whose register-quad kernels need sides that are multiples of 8 and which takes this path for every
// other size (a window resize away in the reference, RenderFrontend.cpp:229-275). min / max of exact values: the same bits whichever path builds a level.
//
// The reference is an SPD-style single dispatch: every 16x16 group reduces a 32x32 footprint of pyramid mip 0 through six
// levels, then the last group (atomic counter) finishes the chain. When an intermediate level has an odd size its 3-wide
// footprints reach into texels owned by a neighbouring group, which the reference reads with only a workgroup barrier.
// Here the pyramid is defined level by level on completed data (min/max are exact and associative, so any evaluation order
// gives the same bits) and computed in two launches:
//   base: one block per 32x32 tile of mip 0 produces up to six levels. Besides the texels it owns, a block computes
//         the few halo texels the odd-size footprints of its own next level need, in LDS, instead of racing for
//         its neighbour's stores. Depth is read exactly once (plus the halo), every level is written exactly once.
//   tail: one block finishes the (<= 64x64 texel) remainder of the chain out of LDS.
// HBM traffic = 4 B/px depth + 8 B per pyramid texel: the algorithmic 6.67 B/px.
#pragma once
#include "../backend.h"
#include "hiz_common.h"

namespace plr {

// 256 threads; hizLds: the launch's dynamic LDS (hizLaunchAnySize sizes it)
PLR_DI void hizBaseBlock(const HizParams& p, float2* hizLds) {
    // bufA holds the even levels' regions, bufB the odd ones; sized by the launcher for this pyramid's worst block (ldsA texels for bufA):
    // with all-even level sizes that is 32x32 + 16x16 texels (10 KB, every block of a 4K frame resident at once), with odd sizes upstream
    // up to 63x63 + 31x31
    float2* bufA = hizLds;
    float2* bufB = hizLds + p.ldsA;
    const int K = p.baseCount;
    // per level: owned range [lo, hi] and computed range [lo, need] (need >= hi: halo for the odd-size footprints above)
    int lox[kHizBaseLevels], loy[kHizBaseLevels], hix[kHizBaseLevels], hiy[kHizBaseLevels], needx[kHizBaseLevels], needy[kHizBaseLevels];
    for (int l = 0; l < K; l++) {
        const int t = 32 >> l;
        lox[l] = ((int)blockIdx.x + p.tileX0) * t; loy[l] = ((int)blockIdx.y + p.tileY0) * t;
        hix[l] = min(lox[l] + t, p.w[l]) - 1; hiy[l] = min(loy[l] + t, p.h[l]) - 1;
    }
    needx[K - 1] = hix[K - 1]; needy[K - 1] = hiy[K - 1];
    for (int l = K - 1; l >= 1; l--) {
        const int sw = p.w[l - 1], sh = p.h[l - 1];
        needx[l - 1] = hix[l - 1]; needy[l - 1] = hiy[l - 1];
        if (needx[l] >= lox[l] && needy[l] >= loy[l]) {
            needx[l - 1] = max(needx[l - 1], min(2 * needx[l] + 1 + (sw & 1), sw - 1));
            needy[l - 1] = max(needy[l - 1], min(2 * needy[l] + 1 + (sh & 1), sh - 1));
        }
    }
    const int t = threadIdx.x;
    // ---- level 0 from the depth buffer
    {
        const int rw = needx[0] - lox[0] + 1, rh = needy[0] - loy[0] + 1;
        const bool oddW = p.depthW & 1, oddH = p.depthH & 1;
        const float* depth = p.depth;
        const int dW = p.depthW, dH = p.depthH;
        if (rw > 0 && rh > 0) {
            for (int i = t; i < rw * rh; i += 256) {
                const int rx = i % rw, ry = i / rw;
                const int x = lox[0] + rx, y = loy[0] + ry;
                const MinMax m = footprint<true>(2 * x, 2 * y, dW, dH, oddH, oddW, [&](int sx, int sy) {
                    const float d = depth[(size_t)sy * dW + sx];
                    return make_float2(d, d);
                });
                const float2 v = make_float2(m.mn, m.mx);
                bufA[ry * rw + rx] = v;
                if (x <= hix[0] && y <= hiy[0]) p.level[0][(size_t)y * p.w[0] + x] = v;
            }
        }
    }
    __syncthreads();
    // ---- levels 1..K-1 out of LDS, ping-ponging between the two buffers
    for (int l = 1; l < K; l++) {
        const float2* src = (l & 1) ? bufA : bufB;
        float2* dst = (l & 1) ? bufB : bufA;
        const int srw = needx[l - 1] - lox[l - 1] + 1;
        const int sox = lox[l - 1], soy = loy[l - 1];
        const int sw = p.w[l - 1], sh = p.h[l - 1];
        const int rw = needx[l] - lox[l] + 1, rh = needy[l] - loy[l] + 1;
        if (rw > 0 && rh > 0) {
            for (int i = t; i < rw * rh; i += 256) {
                const int rx = i % rw, ry = i / rw;
                const int x = lox[l] + rx, y = loy[l] + ry;
                const MinMax m = footprint<false>(2 * x, 2 * y, sw, sh, sh & 1, sw & 1, [&](int sx, int sy) { return src[(sy - soy) * srw + (sx - sox)]; });
                const float2 v = make_float2(m.mn, m.mx);
                dst[ry * rw + rx] = v;
                if (x <= hix[l] && y <= hiy[l]) p.level[l][(size_t)y * p.w[l] + x] = v;
            }
        }
        __syncthreads();
    }
}

// 1024 threads; bufA / bufB: 32 x 32 texels of LDS each
PLR_DI void hizTailAnyBlock(const HizParams& p, float2* bufA, float2* bufB) {
    const int t = threadIdx.x;
    const int first = p.baseCount;
    for (int l = first; l < p.count; l++) {
        const int sw = p.w[l - 1], sh = p.h[l - 1];
        const int w = p.w[l], h = p.h[l];
        float2* dst = ((l - first) & 1) ? bufB : bufA;
        const float2* srcL = ((l - first) & 1) ? bufA : bufB;
        const float2* srcG = p.level[l - 1];
        for (int i = t; i < w * h; i += 1024) {
            const int x = i % w, y = i / w;
            MinMax m;
            if (l == first) m = footprint<false>(2 * x, 2 * y, sw, sh, sh & 1, sw & 1, [&](int sx, int sy) { return srcG[(size_t)sy * sw + sx]; });
            else m = footprint<false>(2 * x, 2 * y, sw, sh, sh & 1, sw & 1, [&](int sx, int sy) { return srcL[sy * sw + sx]; });
            const float2 v = make_float2(m.mn, m.mx);
            dst[i] = v;
            p.level[l][i] = v;
        }
        __syncthreads();
    }
}

// validates the execution, builds the parameters and launches through the caller's two kernels:
//   launchBase(dim3 grid, size_t ldsBytes, const HizParams&)  - 256 threads per block, dynamic LDS
//   launchTail(const HizParams&)                              - one block of 1024 threads
template <class LaunchBase, class LaunchTail>
inline int hizLaunchAnySize(const PassCtx& c, LaunchBase launchBase, LaunchTail launchTail) {
    const int mipCount = c.specInt(0, 0);
    const int resX = c.specInt(1, 0), resY = c.specInt(2, 0);
    if (mipCount < 1) return c.fail(-1, "depthHiZPyramid: mipCount specialisation constant must be >= 1");
    if (mipCount > kHizMaxLevels)
        return c.fail(-6, "depthHiZPyramid: more than 11 pyramid levels (base > 2048) is unsupported, as in the reference shader; build per-tile pyramids");
    if (int rc = c.needSampled(13, F_D32, "depthHiZPyramid depthBuffer")) return rc;
    const ImgView& depth = c.sampled[13];
    if (resX != depth.w || resY != depth.h) return c.fail(-1, "depthHiZPyramid: specialisation constants 1/2 must equal the depth buffer resolution");
    HizParams p{};
    p.depth = (const float*)depth.ptr;
    p.depthW = depth.w; p.depthH = depth.h;
    p.count = mipCount;
    p.baseCount = std::min(mipCount, kHizBaseLevels);
    // binding i is bound to pyramid mip max(i - unused, 0) (RenderFrontend.cpp:831-836)
    const int unused = kHizMaxLevels - mipCount;
    int sw = depth.w, sh = depth.h;
    for (int l = 0; l < mipCount; l++) {
        const int b = l + unused;
        if (int rc = c.needStorage(b, F_RG32F, "depthHiZPyramid pyramid mip")) return rc;
        const ImgView& v = c.storage[b];
        const int w = std::max(sw / 2, 1), h = std::max(sh / 2, 1);
        if (v.w != w || v.h != h)
            return c.fail(-4, "depthHiZPyramid: pyramid mip " + std::to_string(l) + " is " + std::to_string(v.w) + "x" + std::to_string(v.h) + ", expected " +
                                  std::to_string(w) + "x" + std::to_string(h));
        p.level[l] = (float2*)v.ptr; p.w[l] = w; p.h[l] = h;
        sw = w; sh = h;
    }
    if (p.baseCount < p.count && (p.w[p.baseCount] > 32 || p.h[p.baseCount] > 32)) return c.fail(-6, "depthHiZPyramid: tail level exceeds 32x32");
    // a dispatch base / count that covers only part of the tile rows (band rendering) builds the per-tile levels of those rows;
    // the tail of the chain needs every tile and is skipped for a partial dispatch
    const int tileRows = (int)divUp((unsigned)p.h[0], 32u);
    const PassCtx::RowSpan rs = c.base[1] == 0 && (int)c.dispatch[1] >= tileRows ? PassCtx::RowSpan{0, tileRows} : c.rowSpan(tileRows, 1);
    if (rs.y1 <= rs.y0) return 0;
    p.tileY0 = rs.y0;
    // tile columns of the dispatch (tile rendering); a dispatch from column 0 that covers every tile column is the whole width, whatever its count
    const int tileCols = (int)divUp((unsigned)p.w[0], 32u);
    const PassCtx::ColSpan cs = c.base[0] == 0 && (int)c.dispatch[0] >= tileCols ? PassCtx::ColSpan{0, tileCols} : c.colSpan(tileCols, 1);
    if (cs.x1 <= cs.x0) return 0;
    p.tileX0 = cs.x0;
    const bool wholePyramid = rs.y0 == 0 && rs.y1 == tileRows && cs.x0 == 0 && cs.x1 == tileCols;
    const dim3 grid((unsigned)(cs.x1 - cs.x0), (unsigned)(rs.y1 - rs.y0));
    // LDS regions of the worst block: per axis count[l-1] = max(tile, 2 * count[l] + (source size odd)), see the need[] recursion in the kernel
    int cx[kHizBaseLevels], cy[kHizBaseLevels];
    const int K = p.baseCount;
    cx[K - 1] = 32 >> (K - 1); cy[K - 1] = 32 >> (K - 1);
    for (int l = K - 1; l >= 1; l--) {
        cx[l - 1] = std::max(32 >> (l - 1), 2 * cx[l] + (p.w[l - 1] & 1));
        cy[l - 1] = std::max(32 >> (l - 1), 2 * cy[l] + (p.h[l - 1] & 1));
    }
    int texA = 1, texB = 1;
    for (int l = 0; l < K; l++) (l & 1 ? texB : texA) = std::max(l & 1 ? texB : texA, cx[l] * cy[l]);
    p.ldsA = texA;
    const size_t ldsBytes = (size_t)(texA + texB) * sizeof(float2);
    if (ldsBytes > 64 * 1024) return c.fail(-6, "depthHiZPyramid: LDS region of a tile exceeds 64 KB");
    launchBase(grid, ldsBytes, p);
    PLR_CHECK_LAUNCH(c);
    if (p.count > p.baseCount && wholePyramid) {
        launchTail(p);
        PLR_CHECK_LAUNCH(c);
    }
    return 0;
}

} // namespace plr
