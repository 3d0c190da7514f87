// Min/max depth pyramid for depth buffers whose sides are multiples of 16 (the 4K benchmark frame), PLR_MATH_FAST set.
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
};

// min / max contribution of a (min, max) texel to the level above (depthHiZPyramid.comp:95-110): a texel whose max is 0 is all sky and must not
// pull the minimum down
PLR_DI float minTerm(float mn, float mx) { return mn + (mx == 0.f ? 1.f : 0.f); }

template <int LEVELS, bool DOWNSCALE>
__global__ __launch_bounds__(256) void hizQuadKernel(QuadParams p) {
    const int lane = (int)(threadIdx.x & 63u), wave = (int)(threadIdx.x >> 6);
    // lane bits: [1:0] position in a 2x2 quad of lanes, [3:2] position of the quad in a 16-lane row (2x2 quads), [5:4] position of the row in the wave
    const int x8 = (lane & 1) | ((lane >> 1) & 2) | ((lane >> 2) & 4), y8 = ((lane >> 1) & 1) | ((lane >> 2) & 2) | ((lane >> 3) & 4);
    const int px = (int)blockIdx.x * 64 + (wave & 1) * 32 + x8 * 4, py = (int)blockIdx.y * 64 + (wave >> 1) * 32 + y8 * 4;
    const bool active = px < p.depthW && py < p.depthH; // sides are multiples of 16: a 4x4 patch (and every coarser texel) is inside or outside as a whole
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
            *(uint32_t*)h0 = floatToHalfBits(r[0].x) | (floatToHalfBits(r[0].z) << 16);
            *(uint32_t*)(h0 + p.halfW) = floatToHalfBits(r[2].x) | (floatToHalfBits(r[2].z) << 16);
        }
    }
    // level 2: the four lanes of a quad hold the four level-1 texels of one level-2 texel. An inactive lane contributes the neutral pair
    // (1, 0) = the values the shader's accumulators start from.
    float a = active ? minTerm(mn1, mx1) : 1.f, b = active ? mx1 : 0.f;
    a = __builtin_fminf(a, dppf(a, 0xB1)); b = __builtin_fmaxf(b, dppf(b, 0xB1));
    a = __builtin_fminf(a, dppf(a, 0x4E)); b = __builtin_fmaxf(b, dppf(b, 0x4E));
    const float mn2 = __builtin_fminf(1.f, a), mx2 = b;
    if (active && (lane & 3) == 0) p.level[2][(size_t)(py / 8) * (size_t)(p.depthW / 8) + (size_t)(px / 8)] = make_float2(mn2, mx2);
    if (LEVELS >= 4) {
        // level 3: the four quads of a 16-lane row
        float c = active ? minTerm(mn2, mx2) : 1.f, d = active ? mx2 : 0.f;
        c = __builtin_fminf(c, dppf(c, 0x124)); d = __builtin_fmaxf(d, dppf(d, 0x124));
        c = __builtin_fminf(c, dppf(c, 0x128)); d = __builtin_fmaxf(d, dppf(d, 0x128));
        if (active && (lane & 15) == 0) p.level[3][(size_t)(py / 16) * (size_t)(p.depthW / 16) + (size_t)(px / 16)] = make_float2(__builtin_fminf(1.f, c), d);
    }
}

// levels [first, count) by one block: the first from global memory (written by hizQuadKernel), the others out of LDS
__global__ __launch_bounds__(1024) void hizTailKernel(HizParams p, int first, int texelsA) {
    extern __shared__ float2 hizTailLds[];
    float2* bufA = hizTailLds;
    float2* bufB = hizTailLds + texelsA;
    const int t = threadIdx.x;
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

// c: the depthHiZPyramid execution; down: the depthDownscale execution fused into it, or null
static int launchImpl(const PassCtx& c, const PassCtx* down) {
    const int mipCount = c.specInt(0, 0);
    if (mipCount < 5 || mipCount > kHizMaxLevels || !c.hasSampled(13) || c.sampled[13].fmt != F_D32) return kUseGeneralKernel;
    const ImgView& depth = c.sampled[13];
    if (c.specInt(1, 0) != depth.w || c.specInt(2, 0) != depth.h || (depth.w & 15) || (depth.h & 15)) return kUseGeneralKernel;
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
    const int texA = p.w[4] * p.h[4], texB = mipCount > 5 ? p.w[5] * p.h[5] : 1;
    const size_t tailLds = (size_t)(texA + texB) * sizeof(float2);
    if (tailLds > 150 * 1024) return kUseGeneralKernel;
    QuadParams q{};
    q.depth = (const float*)depth.ptr; q.depthW = depth.w; q.depthH = depth.h;
    for (int l = 0; l < 4; l++) q.level[l] = p.level[l];
    if (down) {
        if (!down->hasStorage(0) || !down->hasSampled(1) || down->storage[0].fmt != F_R16F || down->sampled[1].ptr != depth.ptr) return kUseGeneralKernel;
        const ImgView& dst = down->storage[0];
        const PassCtx::RowSpan rs = down->rowSpan(dst.h);
        if (dst.w * 2 != depth.w || dst.h * 2 != depth.h || rs.y0 != 0 || rs.y1 != dst.h || (int)(down->dispatch[0] * 8u) < dst.w) return kUseGeneralKernel;
        q.halfDepth = (uint16_t*)dst.ptr; q.halfW = dst.w;
    }
    static bool ldsRaised = false;
    if (!ldsRaised) {
        if (hipFuncSetAttribute((const void*)hizTailKernel, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024) != hipSuccess) return kUseGeneralKernel;
        ldsRaised = true;
    }
    const dim3 grid(divUp((unsigned)depth.w, 64u), divUp((unsigned)depth.h, 64u));
    if (down) hizQuadKernel<4, true><<<grid, 256, 0, c.stream>>>(q);
    else hizQuadKernel<4, false><<<grid, 256, 0, c.stream>>>(q);
    PLR_CHECK_LAUNCH(c);
    hizTailKernel<<<1, 1024, tailLds, c.stream>>>(p, 4, texA);
    PLR_CHECK_LAUNCH(c);
    return 0;
}

} // namespace fasthiz

static int fasthiz_launch(const PassCtx& c) { return fasthiz::launchImpl(c, nullptr); }
PLR_REGISTER_SHADER_FAST("depthHiZPyramid.comp", fasthiz_launch);
static int fasthiz_fused_downscale(const PassCtx* const* ctxs, size_t count) { return count == 2 ? fasthiz::launchImpl(*ctxs[0], ctxs[1]) : kUseGeneralKernel; }
PLR_REGISTER_FUSION("depthHiZPyramid + depthDownscale", fasthiz_fused_downscale, "depthHiZPyramid.comp", "depthDownscale.comp");

} // namespace plr
