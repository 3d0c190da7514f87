// indirectLightUpscale.comp:17-71 for the 2x2 pixel quad of one half-resolution texel, when the full-resolution image is exactly twice the
// half-resolution one. Shared by the stand-alone upscale pass (stream_fast.hip) and the fused upscale + deferred shade (shading_fast.hip): one
// definition, compiled without contraction, so both produce the same bits.
//
// A full-res pixel X = 2k + p samples the half-res images at k + 0.25 + 0.5 p: the gather / bilinear footprints of the four pixels of a quad,
// and the texels their "closest depth" choice can select (uv + offset * halfResTexelSize lands on texel k + offset), all lie in the 3x3
// half-res neighbourhood (k-1 .. k+1) x (m-1 .. m+1). A thread loads that neighbourhood once (9 depths, 9 Y_SH, 9 CoCg texels in 9 wide loads)
// and linearises each half-res depth once instead of four times; the sub-texel weights are the constants 0.25 / 0.75.
#pragma once
#include "../backend.h"
#include "../device/shading_common.h"

namespace plr {
namespace fastquad {

PLR_DI vec4 halves4(uint2 t) { return vec4(halfBitsToFloat(t.x & 0xffffu), halfBitsToFloat(t.x >> 16), halfBitsToFloat(t.y & 0xffffu), halfBitsToFloat(t.y >> 16)); }
PLR_DI vec2 halves2(uint32_t t) { return vec2(halfBitsToFloat(t & 0xffffu), halfBitsToFloat(t >> 16)); }

// linearizeDepth (shading_common.h) with the product and the sum rounded separately, as the shader compiler evaluates them: at far depths
// far + (1 - d) * (near - far) cancels to ~far * d and the rounding of the product (half an ulp of far) moves the result by tenths of a
// metre - the same size as the 0.5 m edge threshold and the differences the closest-texel choice compares. A fused multiply-add is more
// accurate and therefore picks different texels; only the reciprocal stays approximate (1 ulp, 1e-4 m at 1 km).
PLR_DI float linearDepthRounded(float d, float nf, float nmf, float farP) {
#pragma clang fp contract(off)
    const float t = (1.f - d) * nmf;
    const float den = farP + t;
    return nf * __builtin_amdgcn_rcpf(den);
}

struct UpscaledQuad {
    uint2 ysh[2][2];      // [row][column] of the quad: the RGBA16F texel the pass stores
    uint32_t cocg[2][2];  // the RG16F texel
    uint32_t sig[2][2];   // decision signature (oracle/oracle.h): edge, chosen texel column / row
    float depth[2][2];    // the raw full-resolution depth values the quad read (column 1 repeats column 0 when X + 1 is outside the image)
};

// quad of half-res texel (k, m); the caller has checked that the screen resolution is the target size and the images are "regular" (launchUpscale)
PLR_DI void upscaleQuad(const ImgView& srcYSH, const ImgView& srcCoCg, const ImgView& fullResDepthT, const ImgView& halfResDepthT, const GlobalUbo* __restrict__ g,
                        int k, int m, UpscaledQuad* __restrict__ out) {
    // no fused multiply-adds in this pass: see linearDepthRounded; "linear depth - full-res depth" would also be fused for some of the four
    // candidates and not for others, which breaks exact ties (equal half-res depths) arbitrarily
#pragma clang fp contract(off)
    const int X = 2 * k, Y = 2 * m;
    const int hw = srcYSH.w, hh = srcYSH.h;
    const float nearP = g->nearPlane, farP = g->farPlane, nf = nearP * farP, nmf = nearP - farP;
    auto linearize = [&](float d) { return linearDepthRounded(d, nf, nmf, farP); };
    // 3x3 half-res neighbourhood with clamp-to-edge: N[r][c] = texel (clamp(k-1+c), clamp(m-1+r))
    float hd[3][3];
    uint2 ys[3][3];
    uint32_t cc[3][3];
    const bool interiorX = k >= 1 && k + 2 < hw; // a 4-texel wide load starting at k-1 stays inside the row
    const uint16_t* hdp = (const uint16_t*)halfResDepthT.ptr;
    const uint2* ysp = (const uint2*)srcYSH.ptr;
    const uint32_t* ccp = (const uint32_t*)srcCoCg.ptr;
#pragma unroll
    for (int r = 0; r < 3; r++) {
        const size_t row = (size_t)clampi(m - 1 + r, hh) * (size_t)hw;
        if (interiorX) {
            uint2 d4;  // four half depths (8 bytes) starting at k-1
            __builtin_memcpy(&d4, hdp + row + (k - 1), 8);
            hd[r][0] = linearize(halfBitsToFloat(d4.x & 0xffffu)); hd[r][1] = linearize(halfBitsToFloat(d4.x >> 16)); hd[r][2] = linearize(halfBitsToFloat(d4.y & 0xffffu));
            uint4 ya, cb;
            uint2 yb;
            __builtin_memcpy(&ya, ysp + row + (k - 1), 16);
            yb = ysp[row + (k + 1)];
            ys[r][0] = make_uint2(ya.x, ya.y); ys[r][1] = make_uint2(ya.z, ya.w); ys[r][2] = yb;
            __builtin_memcpy(&cb, ccp + row + (k - 1), 16);
            cc[r][0] = cb.x; cc[r][1] = cb.y; cc[r][2] = cb.z;
        } else {
#pragma unroll
            for (int c = 0; c < 3; c++) {
                const size_t i = row + (size_t)clampi(k - 1 + c, hw);
                hd[r][c] = linearize(halfBitsToFloat(hdp[i]));
                ys[r][c] = ysp[i];
                cc[r][c] = ccp[i];
            }
        }
    }
    // full-res depths of the quad (one 8-byte load per row when X + 1 exists)
    float fd[2][2];
#pragma unroll
    for (int py = 0; py < 2; py++) {
        const float* row = (const float*)fullResDepthT.ptr + (size_t)min(Y + py, fullResDepthT.h - 1) * (size_t)fullResDepthT.w;
        if (X + 1 < fullResDepthT.w) { float2 v; __builtin_memcpy(&v, row + X, 8); out->depth[py][0] = v.x; out->depth[py][1] = v.y; }
        else { out->depth[py][0] = row[X]; out->depth[py][1] = out->depth[py][0]; }
        fd[py][0] = linearize(out->depth[py][0]); fd[py][1] = linearize(out->depth[py][1]);
    }
#pragma unroll
    for (int py = 0; py < 2; py++) {
#pragma unroll
        for (int px = 0; px < 2; px++) {
            // footprint columns / rows inside the 3x3: parity 0 -> texels (k-1, k), a = 0.75; parity 1 -> (k, k+1), a = 0.25
            const int c0 = px, r0 = py;           // index of texel i0 / j0 inside the neighbourhood
            const float a = px ? 0.25f : 0.75f, b = py ? 0.25f : 0.75f;
            const float full = fd[py][px];
            // textureGather order: (i0, j1), (i1, j1), (i1, j0), (i0, j0)
            const float ds[4] = {hd[r0 + 1][c0], hd[r0 + 1][c0 + 1], hd[r0][c0 + 1], hd[r0][c0]};
            const int offx[4] = {0, 1, 1, 0}, offy[4] = {1, 1, 0, 0};
            float minDiff = 1000.f;
            int cx = 0, cy = 0;
            bool isEdge = false;
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const float diff = fabsf(ds[i] - full);
                isEdge = isEdge || diff > 0.5f;
                if (diff < minDiff) { minDiff = diff; cx = offx[i]; cy = offy[i]; }
            }
            vec4 ysh;
            vec2 co;
            if (isEdge) {
                // nearest texel at uv + offset * halfResTexelSize = half-res texel (k + cx, m + cy), clamped: neighbourhood index (1 + cx, 1 + cy)
                const int nc = min(k + cx, hw - 1) - (k - 1), nr = min(m + cy, hh - 1) - (m - 1);
                uint2 ty = ys[1][1];
                uint32_t tc = cc[1][1];
#pragma unroll
                for (int r = 1; r < 3; r++)
#pragma unroll
                    for (int c = 1; c < 3; c++)
                        if (r == nr && c == nc) { ty = ys[r][c]; tc = cc[r][c]; }
                ysh = halves4(ty);
                co = halves2(tc);
            } else {
                const float w00 = (1.f - a) * (1.f - b), w10 = a * (1.f - b), w01 = (1.f - a) * b, w11 = a * b;
                // explicit fused multiply-adds (not left to the contraction default of the including file): every includer computes the same bits
                auto bl = [&](float t00, float t10, float t01, float t11) { return __builtin_fmaf(t11, w11, __builtin_fmaf(t01, w01, __builtin_fmaf(t10, w10, t00 * w00))); };
                const vec4 y00 = halves4(ys[r0][c0]), y10 = halves4(ys[r0][c0 + 1]), y01 = halves4(ys[r0 + 1][c0]), y11 = halves4(ys[r0 + 1][c0 + 1]);
                const vec2 c00 = halves2(cc[r0][c0]), c10 = halves2(cc[r0][c0 + 1]), c01 = halves2(cc[r0 + 1][c0]), c11 = halves2(cc[r0 + 1][c0 + 1]);
                ysh = vec4(bl(y00.x, y10.x, y01.x, y11.x), bl(y00.y, y10.y, y01.y, y11.y), bl(y00.z, y10.z, y01.z, y11.z), bl(y00.w, y10.w, y01.w, y11.w));
                co = vec2(bl(c00.x, c10.x, c01.x, c11.x), bl(c00.y, c10.y, c01.y, c11.y));
            }
            out->ysh[py][px] = make_uint2(floatToHalfBits(ysh.x) | (floatToHalfBits(ysh.y) << 16), floatToHalfBits(ysh.z) | (floatToHalfBits(ysh.w) << 16));
            out->cocg[py][px] = floatToHalfBits(co.x) | (floatToHalfBits(co.y) << 16);
            out->sig[py][px] = (isEdge ? 1u : 0u) | (cx ? 2u : 0u) | (cy ? 4u : 0u);
        }
    }
}

} // namespace fastquad
} // namespace plr
