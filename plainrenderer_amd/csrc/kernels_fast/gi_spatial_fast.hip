// PLR_MATH_FAST variant of filterIndirectDiffuseSpatial.comp:30-135 (exact variant: kernels_exact/gi_filters.hip).
//
// Same samples, same weights, same accumulation order; what changes is how each quantity is computed:
//  * pixelToWorld(uv) = camPos + cameraToPixel / dot(cameraToPixel, forward) * depthLinear. cameraToPixel is the normalised
//    vector forward - tan*ndc.y*up + tan*aspect*ndc.x*right; with an orthonormal camera basis the normalisation cancels against
//    the dot product, so the position is camPos + (forward - tan*ndc.y*up + tan*aspect*ndc.x*right) * depthLinear: no sqrt, no
//    divide per sample.
//  * the sample's clip position is affine in the disc offset: clip = VP*pCenter + ox*(VP*R*tangent) + oy*(VP*R*bitangent); the three
//    projected vectors are computed once per pixel (only x, y, w rows are needed).
//  * distance to the tangent plane |dot(N, pixelWorld - pCenter)| = |dot(N, camPos - pCenter) + depthLinear * dot(N, ray)| with
//    dot(N, ray) affine in the sample's NDC.
//  * divisions are v_rcp_f32 multiplies, FMA contraction is on.
// Per sample this is ~45 VALU operations + 3 gathers instead of ~180 + 3. Results differ from the exact kernel by float
// rounding only (stated tolerance in tests/test_fast_kernels.py); nearest-texel selection and the off-screen test are the
// discontinuities where a last-bit difference can pick a neighbouring texel.
//
// The filter's tangent frame is the exception: tangent = normalize(pCenter - pixelToWorld(uv + one texel)) subtracts two world positions
// a pixel apart (0.012 m at 4K half-res) whose coordinates are tens of metres, so fp32 leaves the difference ~3e-4 relative precision
// and the 1.5 m disc inherits ~0.04 texel of position noise - FROM THE SHADER'S OWN ROUNDING. Any other way of computing the frame
// (including a more accurate one) moves 10 % of all samples onto a neighbouring texel of the 1-spp input. The per-pixel frame is therefore
// evaluated with the shader's exact operation sequence (IEEE divide / sqrt, no contraction: same bits as the oracle); only the 32
// samples per pixel use the restructured arithmetic.
// This file is therefore built like the exact set (no FMA contraction, IEEE divide / sqrt: pragmas are lexical and would not reach the shared
// helpers the frame is computed with); the sample loop switches contraction back on for its own statements and spells its arithmetic out
// as scalars, its reciprocals are explicit v_rcp_f32.
// PLR_BUILD_FLAGS: -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt
#include "../backend.h"
#include "../device/shading_common.h"
#include "../device/xcd.h"
#include "../device/buffer_fetch.h"
#include "fused_gi.h"
#include <cstdlib>
#include <map>
#include <utility>
#include <vector>
#include <type_traits>

namespace plr {

PLR_DI float rcpf(float x) { return __builtin_amdgcn_rcpf(x); }

// Work-groups are dispatched round-robin over the 8 XCDs (block b -> XCD b % 8), each with a private 4 MB L2. The filter gathers
// from a ~100-pixel disc around every pixel, so neighbouring tiles share most of their footprint: with the natural mapping the
// eight L2s each fetch their own copy (measured 364 MB of L2 fills for 62 MB of inputs). The remap hands every XCD one contiguous
// horizontal chunk of tile rows at a time and walks it column by column (device/xcd.h), so the tiles in flight on an XCD form a
// compact block whose footprint stays in that XCD's L2 at any frame width. Walking a band row by row instead (round 1) holds
// (rows in flight + 100) x the whole image width: 4 MB at 4K (half-res 1920 x 16-byte texels), 7 MB at 8K.
// The 32 disc samples are the same for every pixel: the shader seeds its RNG with wang_hash(frameIndexMod4 + filterIndex)
// (:53), so there are only five distinct tables. They are derived once (first launch of a pass) into the pass's scratch memory:
// table[key][0..31] = sqrt(r0), [32..63] = cos(2 pi r1), [64..95] = sin(2 pi r1), [96..127] / [128..159] = their products. The filter kernel reads them with uniform
// (scalar) loads instead of every block re-deriving them.
constexpr int kSampleKeys = 5, kSampleTableFloats = 160;
__global__ void spatialSampleTableKernel(float* __restrict__ table) {
    const int key = (int)threadIdx.x / 32, i = (int)threadIdx.x % 32;
    if (key >= kSampleKeys) return;
    uint32_t rngState = wang_hash((uint32_t)key);
    float r0 = 0.f, r1 = 0.f;
    for (int k = 0; k <= i; k++) { r0 = rand01(rngState); r1 = rand01(rngState); }
    float sn, cs;
    det_sincosf(2.f * PLR_GLSL_PI * r1, &sn, &cs);
    float* t = table + key * kSampleTableFloats;
    t[i] = sqrtf(r0); t[32 + i] = cs; t[64 + i] = sn;
    t[96 + i] = cs * t[i]; t[128 + i] = sn * t[i]; // the disc offsets themselves, for waves whose discs cannot shrink (the same single products the loop forms)
}

// Vector memory instructions, not bytes, are what this filter runs out of: a CU's texture addresser retires one wave-wide load
// per ~16-22 cycles whatever its width (measured: TA busy 75 % of the kernel with three 2/4/8-byte gathers per sample). When depth
// and GI images share the texel grid, a pre-pass packs {Y_SH (8 B), CoCg (4 B), linear-depth denominator (4 B float)} into one
// 16-byte texel, so a sample is ONE dwordx4 gather. The pre-pass also resolves the shader's NaN guard (:118): a texel with a NaN
// component is stored as zeros with a negative denominator (= "skip").
// up to four rectangles of the input in one launch (blockIdx.z): the halo rows above and below a band whose own rows the producer of the input has packed, or -
// tile rendering - the four strips of halo around a tile
struct PackRects { int n; int x0[4], y0[4], x1[4], y1[4]; };
template <int DEPTH_FMT>
__global__ __launch_bounds__(256) void spatialPackKernel(ImgView inYSH, ImgView inCoCg, ImgView depthTexture, const GlobalUbo* __restrict__ g, uint4* __restrict__ packed, PackRects r) {
    const int k = (int)blockIdx.z;
    const int x = r.x0[k] + (int)(blockIdx.x * 64u + (threadIdx.x & 63u));
    const int y = r.y0[k] + (int)(blockIdx.y * 4u + (threadIdx.x >> 6));
    if (x >= r.x1[k] || y >= r.y1[k]) return;
    const size_t idx = (size_t)y * (size_t)inYSH.w + (size_t)x;
    const uint2 yt = ((const uint2*)inYSH.ptr)[idx];
    const uint32_t ct = ((const uint32_t*)inCoCg.ptr)[idx];
    const float dep = Texel<DEPTH_FMT>::load(depthTexture.ptr, idx).x;
    packed[idx] = packGiTexel(yt, ct, dep, g->nearPlane, g->farPlane);
}

// the request-list exchange (MARK above): the texels a rank's samples land on outside its own rectangle arrive one by one (the owners send them, the exchange scatters
// them into the input images and the depth texture); this packs exactly those - one thread per word of the request bitmap
template <int DEPTH_FMT>
__global__ __launch_bounds__(256) void spatialSparsePackKernel(ImgView inYSH, ImgView inCoCg, ImgView depthTexture, const GlobalUbo* __restrict__ g, uint4* __restrict__ packed,
                                                               const uint32_t* __restrict__ bitmap, uint32_t rowWords, uint32_t words) {
    // a lane per TEXEL (bit): 8 words per block; a lane per word with a loop over its bits took 31 us per pass at 8K / 4 (divergent, serial), this is bandwidth of 1 MB
    const uint32_t wi = blockIdx.x * 8u + (threadIdx.x >> 5), b = threadIdx.x & 31u;
    if (wi >= words) return;
    if (!((bitmap[wi] >> b) & 1u)) return;
    const uint32_t y = wi / rowWords, x = (wi - y * rowWords) * 32u + b;
    if (x >= (uint32_t)inYSH.w) return;
    const size_t idx = (size_t)y * (size_t)inYSH.w + x;
    packed[idx] = packGiTexel(((const uint2*)inYSH.ptr)[idx], ((const uint32_t*)inCoCg.ptr)[idx], Texel<DEPTH_FMT>::load(depthTexture.ptr, idx).x, g->nearPlane, g->farPlane);
}
// the request pass's byte map (one byte per texel, row pitch `pitch` bytes) -> the bitmap: a lane per word, 32 bytes = two 16-byte loads. blockIdx.y: the map (two passes'
// requests in one launch). The bytes that were set are cleared again here - the map is all zero when the next frame's request pass starts, without a 2 x 2 MB memset
// launch in front of it every frame (r06c_band_timeline.txt: fillBufferAligned 5 - 10 us + the gap of a dependent launch)
__global__ __launch_bounds__(256) void requestBytesToBitsKernel(uint8_t* __restrict__ bytes0, uint32_t* __restrict__ bitmap0, uint8_t* __restrict__ bytes1, uint32_t* __restrict__ bitmap1,
                                                                uint32_t pitch, uint32_t rowWords, uint32_t words) {
    const uint32_t wi = blockIdx.x * 256u + threadIdx.x;
    if (wi >= words) return;
    uint8_t* bytes = blockIdx.y ? bytes1 : bytes0;
    uint32_t* bitmap = blockIdx.y ? bitmap1 : bitmap0;
    const uint32_t y = wi / rowWords, xw = wi - y * rowWords;
    uint4* src = (uint4*)(bytes + (size_t)y * pitch + (size_t)xw * 32u);
    const uint4 a = src[0], c = src[1];
    const uint32_t d[8] = {a.x, a.y, a.z, a.w, c.x, c.y, c.z, c.w};
    uint32_t bits = 0u;
#pragma unroll
    for (int k = 0; k < 8; k++) {
        // non-zero bytes of a dword -> 4 bits
        const uint32_t v = d[k];
        bits |= (((v & 0xffu) ? 1u : 0u) | ((v & 0xff00u) ? 2u : 0u) | ((v & 0xff0000u) ? 4u : 0u) | ((v & 0xff000000u) ? 8u : 0u)) << (4 * k);
    }
    bitmap[wi] = bits;
    if (bits) { src[0] = make_uint4(0u, 0u, 0u, 0u); src[1] = make_uint4(0u, 0u, 0u, 0u); }
}

// SIG: also write the decision signature: two words per pixel, bit i = x / y parity of sample i's nearest texel (both toggled when off screen; oracle/oracle.h)
// Per-launch constants the launcher works out on the host: this chip has no scalar float unit, so uniform float arithmetic (image sizes as floats, their
// correctly rounded reciprocals, the norms of three rows of viewProjection: three square roots) is otherwise repeated by every lane
struct SpatialFrameConsts {
    float tsx, tsy;    // 1 / output size: the IEEE quotient, as the shader's vec2(1) / textureSize
    float dW, dH;      // depth texture size
    float nW, nH;      // normal texture size
    float3 vpRowNorms; // |x row|, |y row|, |w row| of the xyz part of viewProjection
};
// MARK (round 6, the request-list GI exchange of a partitioned frame; plr_frame.h PLRF_HALO_REQUESTED): the kernel gathers nothing and writes no image - every
// sample that lands on screen but OUTSIDE the valid rectangle (the rank's own rows / columns) sets its texel's BYTE in requestBitmap (here: a byte map,
// bitmapRowWords = bytes per texel row). Same block -> pixel mapping, same per-pixel frame, same position statements as the filter launch that follows: the sample positions are
// evaluated in one place (below, contraction off, explicit fused multiply-adds) so that both instantiations land on the same texels, bit for bit.
template <int DEPTH_FMT, int TX, bool SAME_GRID, bool PACKED, bool SIG, bool MARK = false>
__global__ __launch_bounds__(256) void spatialFilterFastKernel(ImgView outYSH, ImgView outCoCg, ImgView inYSH, ImgView inCoCg, ImgView depthTexture, ImgView normalTexture,
                                                               const GlobalUbo* __restrict__ g, const float* __restrict__ sampleTables, const uint4* __restrict__ packed, int filterIndex,
                                                               int coverW, int coverH, int yBase, int xBase, int tilesX, int tilesY, int chunkRows, uint32_t* __restrict__ sig,
                                                               uint32_t validY0, uint32_t validRowCount, uint32_t validX0, uint32_t validColCount, int rowMissShrinks,
                                                               SpatialFrameConsts fc, uint32_t* __restrict__ requestBitmap = nullptr, uint32_t bitmapRowWords = 0, int requestPhase = 0,
                                                               uint32_t* __restrict__ requestBitmap2 = nullptr) {
    // (MARK with requestBitmap2: BOTH spatial filter passes' requests in one launch - filterIndex's pass into requestBitmap, then filter 1's into requestBitmap2; the
    //  per-pixel frame is the same for both, only the disc's radius and the sample table change: `samples`, `radiusWorld`, T, B, PT, PB and the safety flags are re-set)
    const float* samples = sampleTables + min(g->frameIndexMod4 + (uint32_t)filterIndex, (uint32_t)(kSampleKeys - 1)) * kSampleTableFloats;
    constexpr int TY = 256 / TX;
    int tileX, tileY;
    if (!xcdWalk2(tilesX, tilesY, chunkRows & 0xffff, chunkRows >> 16, tileX, tileY)) return; // device/xcd.h: the note on the XCDs' L2s above (chunkRows: rows | splitX << 16)
    const int px = xBase + tileX * TX + (int)(threadIdx.x % TX); // columns [xBase, coverW) (tile rendering: PassCtx::colSpan), rows [yBase, coverH)
    const int py = yBase + tileY * TY + (int)(threadIdx.x / TX);
    if (px >= coverW || py >= coverH) return;

    const float nearP = g->nearPlane, farP = g->farPlane;
    const float nf = nearP * farP, nmf = nearP - farP;
    const vec3 fwd = ld3(g->cameraForward), up = ld3(g->cameraUp), right = ld3(g->cameraRight), camPos = ld3(g->cameraPosition);
    const float tanH = g->cameraTanFovHalf, tanA = g->cameraTanFovHalf * g->cameraAspectRatio;
    const float dW = fc.dW, dH = fc.dH;
    const int dwi = depthTexture.w, dhi = depthTexture.h;

    float radiusWorld = filterIndex == 1 ? 1.f : 1.5f;
    // requestPhase (the filter with request lists, push constant of the execution): 1 = only the waves whose discs provably stay inside the dispatched rectangle (they
    // need nothing from another GPU: this launch runs WHILE the requested texels travel), 2 = the other waves (behind the exchange). The test is MARK's, below.
    if (MARK || requestPhase != 0) {
        // the rectangle: MARK - what is declared valid (the rank's own rows / columns); a filter launch - what it is dispatched over (the same rectangle)
        const uint32_t rY0 = MARK ? validY0 : (uint32_t)yBase, rYn = MARK ? validRowCount : (uint32_t)(coverH - yBase);
        const uint32_t rX0 = MARK ? validX0 : (uint32_t)xBase, rXn = MARK ? validColCount : (uint32_t)(coverW - xBase);
        // Most waves of a rank's rectangle request nothing: their discs stay inside it. Decided here from the centre texel's depth alone, with the bound `safe` /
        // `safeInside` use below (clip.w of the centre is its linear depth) and 5 % + two texels to spare - before the three exact world positions of the frame,
        // which are most of what such a wave would otherwise pay (the request passes took 55 us each at 8K / 4, profiles/r06_band_cost.txt)
        const int cx = clampTo(px, depthTexture.w - 1), cy = clampTo(py, depthTexture.h - 1);
        const float dq = Texel<DEPTH_FMT>::load(depthTexture.ptr, __umul24((uint32_t)cy, (uint32_t)depthTexture.w) + (uint32_t)cx).x;
        const float lin = nf * rcpf(farP + (1.f - dq) * nmf);
        const float dm = radiusWorld * 1.4143f * 1.01f * 1.05f;
        const float sx = fc.vpRowNorms.x, sy = fc.vpRowNorms.y, sw = fc.vpRowNorms.z;
        const float wMin = (lin - sw * dm) * 0.99f;
        const float su = fabsf(2.f * ((float)px + 0.5f) * fc.tsx - 1.f), sv = fabsf(2.f * ((float)py + 0.5f) * fc.tsy - 1.f);
        bool inside = wMin > 0.f && su * lin + sx * dm <= wMin && sv * lin + sy * dm <= wMin; // on screen: no mirrored sample
        const float inv = rcpf(__builtin_fmaxf(wMin, 1e-20f));
        const float reachY = 0.5f * (sy + sw) * dm * inv * (float)inYSH.h + 3.f, reachX = 0.5f * (sx + sw) * dm * inv * (float)inYSH.w + 3.f;
        inside = inside && (float)py - reachY >= (float)rY0 && (float)py + reachY < (float)(rY0 + rYn) &&
                 (float)px - reachX >= (float)rX0 && (float)px + reachX < (float)(rX0 + rXn);
        const bool allInside = __builtin_amdgcn_ballot_w64(!inside) == 0ull;
        if (MARK ? allInside : (requestPhase == 1 ? !allInside : allInside)) return;
    }
    float tsx, tsy, u0, v0;
    vec3 pCenter, T, B, uT, uB;
    {
        // the shader's own sequence (:33-41), bit for bit: see the note at the top of this file
#pragma clang fp contract(off)
        // IEEE quotient / square root by one Newton step on v_rcp_f32 / v_rsq_f32: the residual is exact (FMA), so the result is the correctly rounded
        // one except when the true value sits within ~2^-24 ulp of a rounding boundary - a third of the instructions of the full division
        // sequence, and the 19 divisions + 5 square roots of the frame are most of its cost. (A miss shows up as sample flips in the parity test.)
        auto quot = [](float a, float b, float rb) { const float q = a * rb; return __builtin_fmaf(__builtin_fmaf(-b, q, a), rb, q); };
        auto root = [](float x) { const float r = __builtin_amdgcn_rsqf(x), s0 = x * r; return __builtin_fmaf(__builtin_fmaf(-s0, s0, x), 0.5f * r, s0); };
        auto unit = [&](vec3 a) { // vecmath.h normalize: a * (1 / sqrt(dot(a, a)))
            const float len = root(dot(a, a));
            return a * quot(1.f, len, rcpf(len));
        };
        tsx = fc.tsx; tsy = fc.tsy;
        u0 = ((float)px + 0.5f) * tsx; v0 = ((float)py + 0.5f) * tsy;
        const float tanH = g->cameraTanFovHalf, aspect = g->cameraAspectRatio;
        auto pixelToWorldExact = [&](float u, float v) -> vec3 {
            // texel index: floor and conversion in one instruction, the clamp in one (device/image.h; same integers as min(max((int)floorf(.), 0), n - 1)),
            // a 32-bit texel index (image sides stay below 2^24)
            const int x = clampTo(floorToInt(u * dW), dwi - 1), y = clampTo(floorToInt(v * dH), dhi - 1);
            const float depth = Texel<DEPTH_FMT>::load(depthTexture.ptr, __umul24((uint32_t)y, (uint32_t)dwi) + (uint32_t)x).x;
            const float den = farP + (-depth + 1.f) * (nearP - farP);                 // linearizeDepth (linearDepth.inc:5-8)
            const float depthLinear = quot(nearP * farP, den, rcpf(den));
            const float ndx = u * 2.f - 1.f, ndy = v * 2.f - 1.f;
            vec3 V = -fwd;                                                            // calculateViewDirectionFromPixel (screenToWorld.inc:4-9)
            V += tanH * ndy * up;
            V -= tanH * aspect * ndx * right;
            const vec3 cameraToPixel = -unit(V);
            const float d = dot(cameraToPixel, fwd), rd = rcpf(d);
            return camPos + vec3(quot(cameraToPixel.x, d, rd), quot(cameraToPixel.y, d, rd), quot(cameraToPixel.z, d, rd)) * depthLinear;
        };
        pCenter = pixelToWorldExact(u0, v0);
        // uv + vec2(1, 0) * texelSize and uv + vec2(0, 1) * texelSize (:37-38): 1 * t = t and v + 0 * t = v exactly for the positive finite values here
        const vec3 pRight = pixelToWorldExact(u0 + tsx, v0);
        const vec3 pUp = pixelToWorldExact(u0, v0 + tsy);
        uT = unit(pCenter - pRight); uB = unit(pCenter - pUp);
        T = radiusWorld * uT;
        B = radiusWorld * uB;
    }
    const int nwi = normalTexture.w, nhi = normalTexture.h;
    vec3 N(0.f, 0.f, 1.f);
    if (!MARK) {
        const int x = clampTo(floorToInt(u0 * fc.nW), nwi - 1), y = clampTo(floorToInt(v0 * fc.nH), nhi - 1);
        const uint32_t t = ((const uint32_t*)normalTexture.ptr)[__umul24((uint32_t)y, (uint32_t)nwi) + (uint32_t)x];
        // c / 255 as the three-instruction form (this file is built with IEEE division: ten instructions per channel otherwise; the same values, device/image.h)
        N = 2.f * vec3(decodeUnorm8Newton(t & 0xffu), decodeUnorm8Newton((t >> 8) & 0xffu), decodeUnorm8Newton((t >> 16) & 0xffu)) - 1.f;
    }
    // clip.xyw = P0 + ox * PT + oy * PB
    const float* vp = g->viewProjection;
    auto projXYW = [&](vec3 p, float w) -> vec3 {
        return vec3(vp[0] * p.x + vp[4] * p.y + vp[8] * p.z + vp[12] * w, vp[1] * p.x + vp[5] * p.y + vp[9] * p.z + vp[13] * w,
                    vp[3] * p.x + vp[7] * p.y + vp[11] * p.z + vp[15] * w);
    };
    const vec3 P0 = projXYW(pCenter, 1.f);
    vec3 PT = projXYW(T, 0.f), PB = projXYW(B, 0.f);
    // dot(N, pixelWorld - pCenter) = c0 + lin * (nF + ndcY * nU + ndcX * nR)
    const float c0 = dot(N, camPos - pCenter);
    const float nF = dot(N, fwd), nU = -tanH * dot(N, up), nR = tanA * dot(N, right);

    // dist = |c0 + lin * (k0 + sv * k1 + su * k2)|
    const float k0 = nF - nU - nR, k1 = 2.f * nU, k2 = 2.f * nR;
    const float c0x4 = 4.f * c0;
    // Can a sample of this pixel leave the screen? Its world offset is ox * T + oy * B with |ox| + |oy| <= sqrt(2) and |T| = |B| = radiusWorld,
    // so |offset| <= dm; the clip coordinates move by at most (row norm) * dm, and |x'| <= w', |y'| <= w' is what "on screen" means.
    bool safe, safeInside = true; // safeInside: the disc provably stays inside the valid rows / columns too (band / tile rendering)
    auto computeSafety = [&]() {
        safeInside = true;
        const float dm = radiusWorld * 1.4143f * 1.01f;
        // norms of the x, y and w rows of viewProjection: per-frame constants, from the launcher (three square roots and a dozen multiply-adds per LANE otherwise -
        // this chip has no scalar float unit to take uniform arithmetic)
        const float sx = fc.vpRowNorms.x, sy = fc.vpRowNorms.y, sw = fc.vpRowNorms.z;
        const float wMin = (P0.z - sw * dm) * 0.999f;
        safe = wMin > 0.f && fabsf(P0.x) + sx * dm <= wMin && fabsf(P0.y) + sy * dm <= wMin;
        // band rendering: only part of the input rows is valid (PassCtx::validRows). A sample's v moves by at most 0.5 (|dy| + |y / w| |dw|) / w <=
        // 0.5 (sy + sw) dm / wMin from the centre's: a disc whose rows provably stay inside the valid rows needs no per-sample row test either
        if (validRowCount < (uint32_t)inYSH.h) {
            const float reach = 0.5f * (sy + sw) * dm * rcpf(__builtin_fmaxf(wMin, 1e-20f)) * (float)inYSH.h + 1.f; // rows
            safeInside = (float)py - reach >= (float)validY0 && (float)py + reach < (float)(validY0 + validRowCount);
        }
        // tile rendering: the same for the valid columns (PassCtx::validCols), with the x row's norm
        if (validColCount < (uint32_t)inYSH.w) {
            const float reach = 0.5f * (sx + sw) * dm * rcpf(__builtin_fmaxf(wMin, 1e-20f)) * (float)inYSH.w + 1.f; // columns
            safeInside = safeInside && (float)px - reach >= (float)validX0 && (float)px + reach < (float)(validX0 + validColCount);
        }
    };
    computeSafety();
    uint8_t* markBytes = (uint8_t*)requestBitmap;
    float resCo = 0.f, resCg = 0.f;
    float weightTotal = 0.f;
    float lengthModifier = 1.f;
    uint32_t sampleParityX = 0u, sampleParityY = 0u;
    const uint32_t ywi = (uint32_t)inYSH.w;
    const float yW = (float)inYSH.w, yH = (float)inYSH.h, yWm1 = yW - 1.f, yHm1 = yH - 1.f, dWm1 = dW - 1.f, dHm1 = dH - 1.f;
    // screen coordinates in the loop are centred and doubled: su = 2 u - 1 = clip.x / clip.w (no halving), "on screen" is |su| <= 1
    const float halfW = 0.5f * yW, halfH = 0.5f * yH, su0 = 2.f * u0 - 1.f, sv0 = 2.f * v0 - 1.f;
    // nf * q with q = k0 + v k1 + u k2 in those coordinates: the near * far factor of the plane distance is folded into the three constants
    const float k0n = nf * (k0 + 0.5f * (k1 + k2)), k1n = nf * 0.5f * k1, k2n = nf * 0.5f * k2;
    const float halfDW = 0.5f * dW, halfDH = 0.5f * dH;
    const uint2* yshTexels = (const uint2*)inYSH.ptr;
    const uint32_t* cocgTexels = (const uint32_t*)inCoCg.ptr;
    // Samples are processed four at a time, branch-free, so that the gathers of a group are in flight together (a per-sample branch
    // makes the compiler wait for each sample's loads in turn); only lengthModifier chains the samples, and it depends on coordinates
    // alone.
    // What bounds it (profiles/r04b_sq_counters.csv + profiles/r04b_isa_mix.txt, 3840x2160, 94.4 us per launch serialised): no single
    // unit. Per wave 1681 VALU instructions, of which half are full rate: priced with the measured issue rates (profiles/r04_valu_rates.txt)
    // they take 83 us of the 94 (0.88), at the 2-cycle peak 45 us (VALU roofline 0.47). The packed gathers (32 x 16 bytes per pixel) make
    // 137.5 k L1 cache-line accesses per CU in 227 k cycles (0.61 per cycle, L1 hit 62 %, L2 hit 93 %), the texture addresser is busy 65 - 67 %
    // of the time and a wave waits on some counter 62 % of its life: VALU issue and the gather path overlap, neither hides completely under
    // the other. Earlier readings for the record, each true of the kernel it was taken on: round 2, 1960 VALU per wave, time unchanged when
    // every gather read the pixel's own texel (issue bound); round 3 after the instruction diet, 1897 -> 1700 per wave with the time unchanged
    // at ~106 us (gather bound, before texel packing halved the gathers).
    // Measured and not kept: a wave-uniform "whole group on screen" shortcut (waves near discs that leave the screen pay for both paths, two
    // waves of occupancy lost); the first filter execution of a frame storing the per-pixel frame (36 bytes) for the second one to load -
    // 270 instructions fewer and slower (98 us loading, 105 us storing, against 94 us).
    // Two copies of the sample loop. A wave whose pixels' discs provably stay on screen (see `safe` above) runs the copy without the
    // mirroring, the off-screen test and the shrinking lengthModifier - a fifth of the per-sample instructions; those conditions could
    // not have fired, so a pixel's result is the same whichever copy its wave ran.
    float rY0 = 0.f, rY1 = 0.f, rY2 = 0.f, rY3 = 0.f;
    const BufferDesc packedTexels = texelBuffer(packed, 16u);
    // INSIDE (a SAFE copy only): the disc may reach rows / columns no neighbouring GPU has sent. Such a sample gets weight 0 and nothing else changes (the disc
    // does not shrink for it: rowMissShrinks == 0), so the on-screen copy serves with one range test per sample - in the ground partitions of the 8K frame most
    // waves' discs reach past the 128-row halo, and sending them all through the off-screen copy cost a band 25 us per pass (profiles/r05a_band_cost.txt)
    auto sampleLoop = [&](auto safeTag, auto insideTag) {
#pragma clang fp contract(fast)
        constexpr bool SAFE = decltype(safeTag)::value;
        constexpr bool INSIDE = decltype(insideTag)::value;
        for (int i0 = 0; i0 < 32; i0 += 4) {
            float su[4], sv[4];
            uint32_t ti[4], di[4];
            bool off[4];
    #pragma unroll
            for (int k = 0; k < 4; k++) {
                // ---- WHERE the sample lands: every operation spelled out (no contraction left to the compiler), so that the MARK instantiation of this kernel
                // and the filtering one compute the same texel from the same inputs
                float cu, cv;
                uint32_t tx, ty;
                bool offScreen = false, miss = false;
                {
#pragma clang fp contract(off)
                    float ox, oy;
                    if (SAFE) { ox = samples[96 + i0 + k]; oy = samples[128 + i0 + k]; }
                    else { const float d = samples[i0 + k] * lengthModifier; ox = samples[32 + i0 + k] * d; oy = samples[64 + i0 + k] * d; }
                    const float clipX = __builtin_fmaf(oy, PB.x, __builtin_fmaf(ox, PT.x, P0.x)), clipY = __builtin_fmaf(oy, PB.y, __builtin_fmaf(ox, PT.y, P0.y));
                    const float clipW = __builtin_fmaf(oy, PB.z, __builtin_fmaf(ox, PT.z, P0.z));
                    const float invW = rcpf(clipW);
                    cu = clipX * invW; cv = clipY * invW;
                    if (SAFE) {
                        // `safe` leaves a margin of 1e-3 of the screen: cu * halfW + halfW lies strictly inside (0, width) - no clamp, and trunc == floor
                        tx = (uint32_t)(int)__builtin_fmaf(cu, halfW, halfW); ty = (uint32_t)(int)__builtin_fmaf(cv, halfH, halfH);
                        if (!INSIDE) miss = (ty - validY0) >= validRowCount || (tx - validX0) >= validColCount;
                    } else {
                        // mirror at the borders (:86-89): a coordinate outside [0,1] is replaced by uv - offset
                        cu = fabsf(cu) > 1.f ? __builtin_fmaf(-2.f, ox, su0) : cu;
                        cv = fabsf(cv) > 1.f ? __builtin_fmaf(-2.f, oy, sv0) : cv;
                        offScreen = __builtin_fmaxf(fabsf(cu), fabsf(cv)) > 1.f; // still off-screen: weight 0, shrink the disc (:100-105)
                        // nearest texel: trunc == floor for non-negative coordinates, negative ones clamp to 0 either way
                        tx = (uint32_t)(int)__builtin_amdgcn_fmed3f(__builtin_fmaf(cu, halfW, halfW), 0.f, yWm1); ty = (uint32_t)(int)__builtin_amdgcn_fmed3f(__builtin_fmaf(cv, halfH, halfH), 0.f, yHm1);
                        // a row no neighbouring band has sent (band rendering) gets weight 0 like an off-screen sample; whether it also shrinks the disc for the
                        // samples after it, as an off-screen one does (:100-105), is the launcher's choice (rowMissShrinks; measured in profiles/r04_config5_series.txt)
                        miss = (ty - validY0) >= validRowCount || (tx - validX0) >= validColCount; // (or a column: tile rendering)
                        lengthModifier = (offScreen || (miss && rowMissShrinks)) ? lengthModifier * 0.98f : lengthModifier;
                    }
                }
                if (MARK) {
                    // one BYTE per texel, plain stores (every writer stores the same 1): millions of atomic ORs on words shared by 32 texels made this pass take 2.4 x
                    // the filter it precedes (profiles/r06_band_cost.txt); requestBytesToBitsKernel folds the bytes into the bitmap the exchange trades
                    if (miss && !offScreen) markBytes[(size_t)ty * bitmapRowWords + tx] = 1u;
                    continue;
                }
                off[k] = offScreen || miss;
                su[k] = cu; sv[k] = cv;
                ti[k] = __umul24(ty, ywi) + tx; // image sides stay below 2^24
                if (SIG) { const uint32_t o = off[k] ? 1u : 0u; sampleParityX |= ((tx + o) & 1u) << (i0 + k); sampleParityY |= ((ty + o) & 1u) << (i0 + k); }
                di[k] = SAME_GRID ? ti[k]
                                  : (uint32_t)(int)__builtin_amdgcn_fmed3f(cv * halfDH + halfDH, 0.f, dHm1) * (uint32_t)dwi + (uint32_t)(int)__builtin_amdgcn_fmed3f(cu * halfDW + halfDW, 0.f, dWm1);
            }
            if (MARK) continue;
            if (PACKED) {
                uint4 t4[4];
    #pragma unroll
                for (int k = 0; k < 4; k++) t4[k] = fetch128(packedTexels, ti[k]); // (device/buffer_fetch.h: no address arithmetic per gather)
    #pragma unroll
                for (int k = 0; k < 4; k++) {
                    const float qden = u2f(t4[k].w); // den / 4; <= 0: texel had a NaN component (skip)
                    const float num = fabsf(c0x4 * qden + (k0n + sv[k] * k1n + su[k] * k2n)); // |c0 * den + nf * q|
                    // num and den are finite here (den < 0 marks a texel to skip, masked below): the hardware maximum replaces the NaN-aware one
                    float weight = __builtin_amdgcn_fmed3f(qden * rcpf(__builtin_fmaxf(num, 0.0004f * qden)), 0.f, 1.f);
                    weight *= weight;
                    // a texel to skip has qden < 0: max(num, negative) = num >= 0, the quotient is <= 0 and the clamp has already made the weight 0
                    if (!SAFE || !INSIDE) weight = off[k] ? 0.f : weight;
                    rY0 += weight * halfBitsToFloat(t4[k].x & 0xffffu); rY1 += weight * halfBitsToFloat(t4[k].x >> 16);
                    rY2 += weight * halfBitsToFloat(t4[k].y & 0xffffu); rY3 += weight * halfBitsToFloat(t4[k].y >> 16);
                    resCo += weight * halfBitsToFloat(t4[k].z & 0xffffu);
                    resCg += weight * halfBitsToFloat(t4[k].z >> 16);
                    weightTotal += weight;
                }
            } else {
            float dep[4];
            uint2 yt[4];
            uint32_t ct[4];
    #pragma unroll
            for (int k = 0; k < 4; k++) {
                dep[k] = Texel<DEPTH_FMT>::load(depthTexture.ptr, di[k]).x;
                yt[k] = yshTexels[ti[k]];
                ct[k] = cocgTexels[ti[k]];
            }
    #pragma unroll
            for (int k = 0; k < 4; k++) {
                // depthLinear = nf / den with den = far + (1 - depth) * (near - far) > 0; the distance to the tangent plane is
                // |c0 + depthLinear * q| = |c0 * den + nf * q| / den, so weight = clamp(0.25 * den / max(|c0 * den + nf * q|, 1e-4 * den))^2: one reciprocal
                const float den = farP + (1.f - dep[k]) * nmf;
                const float num = fabsf(c0 * den + (k0n + sv[k] * k1n + su[k] * k2n));
                float weight = __builtin_amdgcn_fmed3f((0.25f * den) * rcpf(gmax(num, 0.0001f * den)), 0.f, 1.f);
                weight *= weight;
                vec4 sY(halfBitsToFloat(yt[k].x & 0xffffu), halfBitsToFloat(yt[k].x >> 16), halfBitsToFloat(yt[k].y & 0xffffu), halfBitsToFloat(yt[k].y >> 16));
                float co = halfBitsToFloat(ct[k] & 0xffffu), cg = halfBitsToFloat(ct[k] >> 16);
                // NaN guard (:118): finite half inputs cannot overflow this sum, so it is NaN exactly when a component is NaN
                const float nanProbe = ((sY.x + sY.y) + (sY.z + sY.w)) + (co + cg);
                const bool use = !off[k] && weight > 0.f && nanProbe == nanProbe;
                weight = use ? weight : 0.f;
                if (nanProbe != nanProbe) { sY = vec4(0.f); co = 0.f; cg = 0.f; } // 0 * NaN would poison the sums
                rY0 += weight * sY.x; rY1 += weight * sY.y; rY2 += weight * sY.z; rY3 += weight * sY.w;
                resCo += weight * co;
                resCg += weight * cg;
                weightTotal += weight;
            }
            }
        }
    };
    if (MARK) {
        // a wave whose discs provably stay inside the rank's own rectangle requests nothing; the others walk the samples with the copy the filter will take
        auto markPass = [&]() {
            if (__builtin_amdgcn_ballot_w64(!safe) != 0ull) sampleLoop(std::false_type{}, std::true_type{});
            else if (__builtin_amdgcn_ballot_w64(!safeInside) != 0ull) sampleLoop(std::true_type{}, std::false_type{});
        };
        markPass();
        if (requestBitmap2) { // the second spatial filter pass (filterIndex 1: its own sample table, a disc of 1 m): same frame, same statements
#pragma clang fp contract(off)
            samples = sampleTables + min(g->frameIndexMod4 + 1u, (uint32_t)(kSampleKeys - 1)) * kSampleTableFloats;
            radiusWorld = 1.f;
            T = radiusWorld * uT; B = radiusWorld * uB;
            PT = projXYW(T, 0.f); PB = projXYW(B, 0.f);
            computeSafety();
            lengthModifier = 1.f;
            markBytes = (uint8_t*)requestBitmap2;
            markPass();
        }
        return;
    }
    if (__builtin_amdgcn_ballot_w64(!safe) != 0ull) sampleLoop(std::false_type{}, std::true_type{});
    else if (__builtin_amdgcn_ballot_w64(!safeInside) == 0ull) sampleLoop(std::true_type{}, std::true_type{});
    else if (rowMissShrinks == 0) sampleLoop(std::true_type{}, std::false_type{});
    else sampleLoop(std::false_type{}, std::true_type{});
    const float inv = rcpf(gmax(weightTotal, 0.00001f));
    const size_t idx = (size_t)py * (size_t)outYSH.w + px;
    Texel<F_RGBA16F>::store(outYSH.ptr, idx, vec4(rY0 * inv, rY1 * inv, rY2 * inv, rY3 * inv));
    Texel<F_RG16F>::store(outCoCg.ptr, idx, vec4(resCo * inv, resCg * inv, 0.f, 0.f));
    if (SIG) { sig[2 * idx] = sampleParityX; sig[2 * idx + 1] = sampleParityY; }
}

// scratch of a filter pass: [sample tables | packed texels of the whole input image]
constexpr size_t kSpatialTableBytes = 4096;
static_assert(sizeof(float) * kSampleKeys * kSampleTableFloats <= kSpatialTableBytes, "sample tables");
static uint8_t* spatialScratch(const PassCtx& c, bool sameGrid) {
    const size_t packedBytes = sameGrid ? (size_t)c.sampled[2].w * (size_t)c.sampled[2].h * 16u : 0u;
    const bool freshScratch = c.scratchSize && *c.scratchSize < kSpatialTableBytes + packedBytes;
    uint8_t* scratch = (uint8_t*)c.scratch(kSpatialTableBytes + packedBytes);
    if (scratch && freshScratch) spatialSampleTableKernel<<<1, 256, 0, c.stream>>>((float*)scratch);
    return scratch;
}

int spatialFilterPackTarget(const PassCtx& c, SpatialPackTarget* out) {
    if (!c.hasSampled(2) || !c.hasSampled(3) || !c.hasSampled(4) || c.sampled[2].fmt != F_RGBA16F || c.sampled[3].fmt != F_RG16F) return kUseGeneralKernel;
    const bool sameGrid = c.sampled[4].w == c.sampled[2].w && c.sampled[4].h == c.sampled[2].h && c.sampled[3].w == c.sampled[2].w && c.sampled[3].h == c.sampled[2].h;
    if (!sameGrid || (c.sampled[4].fmt != F_R16F && c.sampled[4].fmt != F_D32)) return kUseGeneralKernel;
    // every reason for which THIS pass's launcher hands the execution to the general kernel must also be a "no" here: a producer that was promised a packed-texel
    // consumer may leave the unpacked image unwritten (fusion level 2), and the general kernel reads the image
    if (!c.global || !c.globalHost) return kUseGeneralKernel;
    uint8_t* scratch = spatialScratch(c, true);
    if (!scratch) return c.fail(-2, "filterIndirectDiffuseSpatial: cannot allocate scratch memory");
    out->packed = (uint4*)(scratch + kSpatialTableBytes);
    out->depth = c.sampled[4];
    return 0;
}

// ---- what of a filter pass's packed copy its producer filled this frame: rectangles (whole rows in band rendering, a tile's rectangle in tile rendering).
// Host-side bookkeeping, one entry per filter pass = per scratch slot; the backend is one instance per host thread
struct PackRect { int x0, y0, x1, y1; };
struct PackedRects {
    uint64_t frameSerial = 0;
    const void* source = nullptr;             // the Y_SH image the texels were packed from
    const void* packed = nullptr;             // the packed copy they were written to (a re-allocated scratch invalidates the entry)
    std::vector<PackRect> rects;
};
static thread_local std::map<const void*, PackedRects> g_packedRects;

int spatialPackTargetOfConsumer(const PassCtx& producer, int outY, int outC, SpatialPackTarget* out) {
    const PassCtx* f = producer.consumer;
    if (!f || !producer.hasStorage(outY) || !producer.hasStorage(outC) || !f->hasSampled(2) || !f->hasSampled(3) || !f->hasStorage(0)) return kUseGeneralKernel;
    if (producer.storage[outY].ptr != f->sampled[2].ptr || producer.storage[outC].ptr != f->sampled[3].ptr || producer.storage[outY].w != f->sampled[2].w ||
        producer.storage[outY].h != f->sampled[2].h)
        return kUseGeneralKernel;
    return spatialFilterPackTarget(*f, out);
}
void spatialNotePackedRect(const PassCtx& producer, int x0, int y0, int x1, int y1) {
    const PassCtx* f = producer.consumer;
    if (!f || !f->scratchSlot || y1 <= y0 || x1 <= x0) return;
    PackedRects& e = g_packedRects[(const void*)f->scratchSlot];
    const void* packed = *f->scratchSlot ? (const uint8_t*)*f->scratchSlot + kSpatialTableBytes : nullptr;
    if (e.frameSerial != producer.frameSerial || e.source != f->sampled[2].ptr || e.packed != packed) {
        e.frameSerial = producer.frameSerial; e.source = f->sampled[2].ptr; e.packed = packed; e.rects.clear();
    }
    e.rects.push_back({x0, y0, x1, y1});
    countFusedExecutions(1);
}
void spatialNotePackedRows(const PassCtx& producer, int y0, int y1) {
    const PassCtx* f = producer.consumer;
    if (f && f->hasSampled(2)) spatialNotePackedRect(producer, 0, y0, f->sampled[2].w, y1);
}
// the rectangle `all` minus what the producer packed this frame, as at most four rectangles (more: the caller packs `all` whole; the producer's texels
// are then written a second time with the same values)
static int unpackedRects(const PassCtx& c, const void* packed, PackRect all, PackRect out[4]) {
    std::vector<PackRect> rest{all};
    auto it = c.scratchSlot ? g_packedRects.find((const void*)c.scratchSlot) : g_packedRects.end();
    if (it != g_packedRects.end() && it->second.frameSerial == c.frameSerial && it->second.source == c.sampled[2].ptr && it->second.packed == packed) {
        for (const PackRect& r : it->second.rects) {
            std::vector<PackRect> next;
            for (const PackRect& q : rest) {
                if (r.x1 <= q.x0 || r.x0 >= q.x1 || r.y1 <= q.y0 || r.y0 >= q.y1) { next.push_back(q); continue; }
                if (q.y0 < r.y0) next.push_back({q.x0, q.y0, q.x1, r.y0});                                               // above
                if (r.y1 < q.y1) next.push_back({q.x0, r.y1, q.x1, q.y1});                                               // below
                const int m0 = std::max(q.y0, r.y0), m1 = std::min(q.y1, r.y1);
                if (q.x0 < r.x0) next.push_back({q.x0, m0, r.x0, m1});                                                   // left of it, between the two
                if (r.x1 < q.x1) next.push_back({r.x1, m0, q.x1, m1});                                                   // right of it
            }
            rest.swap(next);
        }
    }
    if (rest.size() > 4) { out[0] = all; return 1; }
    for (size_t i = 0; i < rest.size(); i++) out[i] = rest[i];
    return (int)rest.size();
}

static int launchSpatialFilterFast(const PassCtx& c) {
    if (int rc = c.needGlobal()) return rc;
    if (int rc = c.needStorage(0, F_RGBA16F, "filterIndirectDiffuseSpatial imageOut_Y_SH")) return rc;
    if (int rc = c.needStorage(1, F_RG16F, "filterIndirectDiffuseSpatial imageOut_CoCg")) return rc;
    if (int rc = c.needSampled(2, F_RGBA16F, "filterIndirectDiffuseSpatial texture_Y_SH")) return rc;
    if (int rc = c.needSampled(3, F_RG16F, "filterIndirectDiffuseSpatial texture_CoCg")) return rc;
    if (int rc = c.needSampled(4, -1, "filterIndirectDiffuseSpatial depthTexture")) return rc;
    if (int rc = c.needSampled(5, F_RGBA8, "filterIndirectDiffuseSpatial normalTexture")) return rc;
    if (c.sampled[3].w != c.sampled[2].w || c.sampled[3].h != c.sampled[2].h) return c.fail(-4, "filterIndirectDiffuseSpatial: Y_SH and CoCg inputs differ in size");
    const int filterIndex = c.specInt(0, 0);
    const ImgView& out = c.storage[0];
    const PassCtx::RowSpan rs = c.rowSpan(out.h);
    const PassCtx::ColSpan cs = c.colSpan(out.w);
    const int w = cs.x1, x0 = cs.x0, h = rs.y1, y0 = rs.y0; // columns [x0, w), rows [y0, h)
    if (w <= x0 || h <= y0) return 0;
    // 64x4 tiles measured best (64: 197 us, 32: 199 us, 16: 205 us per pass at 4K before the instruction diet)
#ifndef PLR_SPATIAL_TX
#define PLR_SPATIAL_TX 64
#endif
    constexpr int TXv = PLR_SPATIAL_TX, TYv = 256 / TXv;
    const int tilesX = (int)divUp((unsigned)(w - x0), (unsigned)TXv), tilesY = (int)divUp((unsigned)(h - y0), (unsigned)TYv);
    // XCD columns (device/xcd.h xcdWalk2): at the 4K frame's 1920 x 1080 filter image two columns of XCDs - chunks of 960 texels x ~90 rows, three per XCD - measure
    // 184 - 187 us for the two passes against 188 - 190 with full-width chunks of 68 rows (the frame 0.699 - 0.702 against 0.704 - 0.705 ms) and 1.77 x the
    // algorithmic HBM bytes against 1.87 x (four chunks of 68 rows: the same time, 1.94 x; two of 135: 1.60 x and the old time); at 8K (3840 x 2160) two columns
    // are neutral and four cost 3 %, at 1080p (960 x 540) two cost 4 %: one column there (profiles/r06_spatial_walk.txt).
    // PLR_SPATIAL_SPLIT_X / PLR_SPATIAL_CHUNKS: the hooks of tools/spatial_walk_ab.sh
    static const int splitXEnv = std::getenv("PLR_SPATIAL_SPLIT_X") ? std::atoi(std::getenv("PLR_SPATIAL_SPLIT_X")) : 0;
    static const int chunksEnv = std::getenv("PLR_SPATIAL_CHUNKS") ? std::atoi(std::getenv("PLR_SPATIAL_CHUNKS")) : 0;
    const int widthPx = tilesX * TXv, heightPx = tilesY * TYv;
    int splitX = splitXEnv ? splitXEnv : (widthPx >= 1536 && widthPx < 3072 ? 2 : 1);
    if ((splitX != 2 && splitX != 4 && splitX != 8) || tilesX < 2 * splitX) splitX = 1;
    // one column: chunks of ~68 pixel rows measured best at 4K (2 per XCD of full-width chunks: 222 -> 199 us for the two passes) and 8K (4 per XCD: 1472 -> 827 us)
    const int xcdRows = 8 / splitX, chunkPx = splitX == 1 ? 68 : 90;
    const int chunksPerXcd = chunksEnv > 0 ? chunksEnv : std::max(1, (heightPx + chunkPx * xcdRows / 2) / (chunkPx * xcdRows));
    const int chunkRows = xcdChunkRows2(tilesY, chunksPerXcd, splitX) | (splitX << 16);
    const dim3 grid = xcdWalkGrid2(tilesX, tilesY, chunksPerXcd, splitX);
    // half-res trace: depth and GI images share the texel grid (one texel index serves all gathers, and the packed path applies)
    const bool sameGrid = c.sampled[4].w == c.sampled[2].w && c.sampled[4].h == c.sampled[2].h;
    uint8_t* scratch = spatialScratch(c, sameGrid);
    if (!scratch) return c.fail(-2, "filterIndirectDiffuseSpatial: cannot allocate scratch memory");
    PLR_CHECK_LAUNCH(c);
    float* tables = (float*)scratch;
    uint4* packed = (uint4*)(scratch + kSpatialTableBytes);
    int validLo, validHi, validLoX, validHiX;
    c.validRowRange(c.sampled[2].h, &validLo, &validHi);
    c.validColRange(c.sampled[2].w, &validLoX, &validHiX);
    // the request-list exchange (plr_frame.h PLRF_HALO_REQUESTED): storage buffer 6 = the bitmap of the texels outside the dispatched rectangle this execution's samples land
    // on (written by giSampleRequests.comp over the same rectangle, filled in by the exchange). Everything is valid then; what has to be packed is the rectangle itself
    // and those texels
    const bool requested = c.hasSbuf(6);
    int requestPhase = 0; // push constant (4 bytes, optional): 0 = every wave, 1 = the waves that need no requested texel, 2 = the others (see the kernel)
    if (requested && c.push.size() >= 4) std::memcpy(&requestPhase, c.push.data(), 4);
    if (requestPhase < 0 || requestPhase > 2) return c.fail(-1, "filterIndirectDiffuseSpatial: request phase out of range");
    const uint32_t bitmapRowWords = ((uint32_t)c.sampled[2].w + 31u) / 32u;
    if (requested && (!sameGrid || c.sbuf[6].size < (size_t)bitmapRowWords * (size_t)c.sampled[2].h * 4u))
        return c.fail(-4, "filterIndirectDiffuseSpatial: the request bitmap (storage buffer 6) needs inputs on the depth texture's grid and ceil(width / 32) words per texel row");
    if (sameGrid) {
        // what the filter can read: the dispatched rows and a margin (samples further away read whatever an earlier frame packed there, exactly like
        // the stale image rows they would read unpacked), inside the rows declared valid (band rendering: a sample on any other row has weight 0);
        // minus what the producer of the input packed itself (fused_gi.h). Tile rendering: the same for the columns.
        // (the margin follows the declared valid rows: a band's GI halo grows with the frame height, 64 trace rows per 2160 - ADVICE r03: with a fixed 128
        //  a frame taller than 4320 rows read stale packed texels on valid halo rows beyond it)
        // (and the declared valid rows may be the whole image while the dispatch is a band: a GI halo as large as the image, the exact mode of tools/config5_series.sh)
        const int margin = requested ? 0 : std::max({128, y0 - validLo, validHi - h});
        const int p0 = std::max({y0 - margin, 0, validLo}), p1 = std::min({h + margin, (int)c.sampled[2].h, validHi});
        const bool tiled = requested || validLoX > 0 || validHiX < (int)c.sampled[2].w || x0 > 0 || w < (int)c.sampled[2].w;
        const int marginX = requested ? 0 : std::max({128, x0 - validLoX, validHiX - w});
        const int q0 = tiled ? std::max({x0 - marginX, 0, validLoX}) : 0, q1 = tiled ? std::min({w + marginX, (int)c.sampled[2].w, validHiX}) : (int)c.sampled[2].w;
        PackRect todo[4];
        const int nTodo = p1 > p0 && q1 > q0 && requestPhase != 2 ? unpackedRects(c, packed, PackRect{q0, p0, q1, p1}, todo) : 0; // (phase 2: phase 1 packed the rectangle)
        if (c.sampled[4].fmt != F_R16F && c.sampled[4].fmt != F_D32) return c.fail(-4, "filterIndirectDiffuseSpatial: depthTexture must be R16_sFloat or Depth32");
        if (nTodo) {
            PackRects pr{};
            pr.n = nTodo;
            int maxW = 0, maxH = 0;
            for (int i = 0; i < nTodo; i++) {
                pr.x0[i] = todo[i].x0; pr.y0[i] = todo[i].y0; pr.x1[i] = todo[i].x1; pr.y1[i] = todo[i].y1;
                maxW = std::max(maxW, todo[i].x1 - todo[i].x0); maxH = std::max(maxH, todo[i].y1 - todo[i].y0);
            }
            const dim3 pgrid(divUp((unsigned)maxW, 64u), divUp((unsigned)maxH, 4u), (unsigned)nTodo);
            if (c.sampled[4].fmt == F_R16F) spatialPackKernel<F_R16F><<<pgrid, 256, 0, c.stream>>>(c.sampled[2], c.sampled[3], c.sampled[4], c.global, packed, pr);
            else spatialPackKernel<F_D32><<<pgrid, 256, 0, c.stream>>>(c.sampled[2], c.sampled[3], c.sampled[4], c.global, packed, pr);
            PLR_CHECK_LAUNCH(c);
            c.splitTiming("texel packing");
        } else if (p1 > p0 && requestPhase != 2) countFusedExecutions(1); // the producer did all of it
        if (requested && requestPhase != 1) { // (phase 1 runs while the requested texels are still on their way)
            const uint32_t words = bitmapRowWords * (uint32_t)c.sampled[2].h;
            if (c.sampled[4].fmt == F_R16F) spatialSparsePackKernel<F_R16F><<<divUp(words, 8u), 256, 0, c.stream>>>(c.sampled[2], c.sampled[3], c.sampled[4], c.global, packed, (const uint32_t*)c.sbuf[6].ptr, bitmapRowWords, words);
            else spatialSparsePackKernel<F_D32><<<divUp(words, 8u), 256, 0, c.stream>>>(c.sampled[2], c.sampled[3], c.sampled[4], c.global, packed, (const uint32_t*)c.sbuf[6].ptr, bitmapRowWords, words);
            PLR_CHECK_LAUNCH(c);
            c.splitTiming("requested texels");
        }
    }
    uint32_t* sig = c.sigFor(2u * (size_t)out.w * (size_t)out.h); // two words per pixel
    // per-frame constants of the "can a sample of this pixel leave the screen" test, from the host's copy of the global buffer (a host the backend cannot read
    // the buffer of gets the general kernel, counted: plr_get_general_kernel_executions)
    if (!c.globalHost) return kUseGeneralKernel;
    const float* hvp = c.globalHost->viewProjection;
    auto rowNorm = [&](int r) { return (float)std::sqrt((double)hvp[r] * hvp[r] + (double)hvp[4 + r] * hvp[4 + r] + (double)hvp[8 + r] * hvp[8 + r]); };
    SpatialFrameConsts fc;
    fc.tsx = 1.f / (float)out.w; fc.tsy = 1.f / (float)out.h; // IEEE single-precision quotients (this file's host code is built without fast-math)
    fc.dW = (float)c.sampled[4].w; fc.dH = (float)c.sampled[4].h; fc.nW = (float)c.sampled[5].w; fc.nH = (float)c.sampled[5].h;
    fc.vpRowNorms = make_float3(rowNorm(0), rowNorm(1), rowNorm(3));
    static const int rowMissShrinks = std::getenv("PLR_BAND_ROW_MISS_SHRINKS") ? std::atoi(std::getenv("PLR_BAND_ROW_MISS_SHRINKS")) : 0; // experiment hook; 0 = the exact kernel's rule (kernels_exact/gi_filters.hip)
#define PLR_SPATIAL_ARGS out, c.storage[1], c.sampled[2], c.sampled[3], c.sampled[4], c.sampled[5], c.global, tables, packed, filterIndex, w, h, y0, x0, tilesX, tilesY, chunkRows, sig, \
                         (uint32_t)validLo, (uint32_t)std::max(validHi - validLo, 0), (uint32_t)validLoX, (uint32_t)std::max(validHiX - validLoX, 0), rowMissShrinks, fc, (uint32_t*)nullptr, 0u, requestPhase
#define PLR_SPATIAL_LAUNCH(FMT, SG, PK)                                                                             \
    do {                                                                                                            \
        if (sig) spatialFilterFastKernel<FMT, TXv, SG, PK, true><<<grid, 256, 0, c.stream>>>(PLR_SPATIAL_ARGS);     \
        else spatialFilterFastKernel<FMT, TXv, SG, PK, false><<<grid, 256, 0, c.stream>>>(PLR_SPATIAL_ARGS);        \
    } while (0)
    if (c.sampled[4].fmt == F_R16F) {
        if (sameGrid) PLR_SPATIAL_LAUNCH(F_R16F, true, true); else PLR_SPATIAL_LAUNCH(F_R16F, false, false);
    } else if (c.sampled[4].fmt == F_D32) {
        if (sameGrid) PLR_SPATIAL_LAUNCH(F_D32, true, true); else PLR_SPATIAL_LAUNCH(F_D32, false, false);
    } else return c.fail(-4, "filterIndirectDiffuseSpatial: depthTexture must be R16_sFloat or Depth32");
#undef PLR_SPATIAL_LAUNCH
#undef PLR_SPATIAL_ARGS
    PLR_CHECK_LAUNCH(c);
    return 0;
}
PLR_REGISTER_SHADER_FAST("filterIndirectDiffuseSpatial.comp", launchSpatialFilterFast);

// ---- giSampleRequests.comp (no reference counterpart; the request-list GI exchange of a partitioned frame, plr_frame.h PLRF_HALO_REQUESTED): for the rectangle of
// trace pixels it is dispatched over - recorded like the spatial filter execution it precedes: same dispatch, valid_rows / valid_cols = the rank's own rectangle - sets
// the bit of every texel OUTSIDE that rectangle a disc sample of filterIndirectDiffuseSpatial.comp (specialisation constant 0 = its filterIndex) lands on.
//   sampled image 4: the filter's depthTexture (the trace resolution); storage buffer 6: the bitmap, ceil(width / 32) words per texel row, cleared here.
// The sample positions depend on depth, camera and frame index only (filterIndirectDiffuseSpatial.comp:53-105), so this runs before the trace.
// `second`: the execution of the OTHER spatial filter pass over the same rectangle (pass fusion: both passes' requests in one launch), or null
static int launchGiSampleRequestsImpl(const PassCtx& c, const PassCtx* second) {
    if (int rc = c.needGlobal()) return rc;
    if (int rc = c.needSampled(4, -1, "giSampleRequests depthTexture")) return rc;
    if (int rc = c.needSbuf(6, 4, "giSampleRequests bitmap")) return rc;
    const ImgView& depth = c.sampled[4];
    if (depth.fmt != F_R16F && depth.fmt != F_D32) return c.fail(-4, "giSampleRequests: depthTexture must be R16_sFloat or Depth32");
    const uint32_t rowWords = ((uint32_t)depth.w + 31u) / 32u;
    const size_t bitmapBytes = (size_t)rowWords * (size_t)depth.h * 4u;
    if (c.sbuf[6].size < bitmapBytes || (second && second->sbuf[6].size < bitmapBytes)) return c.fail(-4, "giSampleRequests: the bitmap needs ceil(width / 32) words per texel row of the depth texture");
    if (!c.globalHost) return c.fail(-4, "giSampleRequests: the global uniform block must have been filled through plr_set_uniform_buffer_data");
    static const int rowMissShrinks = std::getenv("PLR_BAND_ROW_MISS_SHRINKS") ? std::atoi(std::getenv("PLR_BAND_ROW_MISS_SHRINKS")) : 0;
    if (rowMissShrinks) return c.fail(-4, "giSampleRequests: PLR_BAND_ROW_MISS_SHRINKS changes where samples land depending on what is valid; not with request lists");
    const int filterIndex = c.specInt(0, 0);
    const PassCtx::RowSpan rs = c.rowSpan(depth.h);
    const PassCtx::ColSpan cs = c.colSpan(depth.w);
    const int w = cs.x1, x0 = cs.x0, h = rs.y1, y0 = rs.y0;
    // scratch: [sample tables | byte map: one byte per texel, rows of rowWords * 32 bytes | a second byte map (two passes in one launch)]
    const uint32_t bytePitch = rowWords * 32u;
    const size_t byteMapBytes = (size_t)bytePitch * (size_t)depth.h;
    const size_t scratchBytes = kSpatialTableBytes + 2 * byteMapBytes;
    const bool freshScratch = c.scratchSize && *c.scratchSize < scratchBytes;
    uint8_t* scratch = (uint8_t*)c.scratch(scratchBytes);
    if (!scratch) return c.fail(-2, "giSampleRequests: cannot allocate scratch memory");
    if (freshScratch) spatialSampleTableKernel<<<1, 256, 0, c.stream>>>((float*)scratch);
    uint8_t* byteMap = scratch + kSpatialTableBytes;
    uint8_t* byteMap2 = second ? byteMap + byteMapBytes : nullptr;
    // (both maps are zero between frames: requestBytesToBitsKernel clears what a pass set)
    if (freshScratch && hipMemsetAsync(byteMap, 0, 2 * byteMapBytes, c.stream) != hipSuccess) return c.fail(-2, "giSampleRequests: hipMemsetAsync");
    auto clearBitmaps = [&]() {
        if (hipMemsetAsync(c.sbuf[6].ptr, 0, bitmapBytes, c.stream) != hipSuccess) return false;
        return !second || hipMemsetAsync(second->sbuf[6].ptr, 0, bitmapBytes, c.stream) == hipSuccess;
    };
    if (w <= x0 || h <= y0) return clearBitmaps() ? 0 : c.fail(-2, "giSampleRequests: hipMemsetAsync");
    constexpr int TXv = PLR_SPATIAL_TX, TYv = 256 / TXv;
    const int tilesX = (int)divUp((unsigned)(w - x0), (unsigned)TXv), tilesY = (int)divUp((unsigned)(h - y0), (unsigned)TYv);
    const int chunksPerXcd = std::max(1, (tilesY * TYv + 272) / 544);
    const int chunkRows = xcdChunkRows(tilesY, chunksPerXcd) | (1 << 16); // (rows | splitX << 16: the kernel's xcdWalk2)
    const dim3 grid = xcdWalkGrid(tilesX, tilesY, chunksPerXcd);
    PLR_CHECK_LAUNCH(c);
    int validLo, validHi, validLoX, validHiX;
    c.validRowRange(depth.h, &validLo, &validHi);
    c.validColRange(depth.w, &validLoX, &validHiX);
    if (validLo <= 0 && validHi >= depth.h && validLoX <= 0 && validHiX >= depth.w) // the rectangle is the image: nothing lies outside it
        return clearBitmaps() ? 0 : c.fail(-2, "giSampleRequests: hipMemsetAsync");
    const float* hvp = c.globalHost->viewProjection;
    auto rowNorm = [&](int r) { return (float)std::sqrt((double)hvp[r] * hvp[r] + (double)hvp[4 + r] * hvp[4 + r] + (double)hvp[8 + r] * hvp[8 + r]); };
    SpatialFrameConsts fc;
    fc.tsx = 1.f / (float)depth.w; fc.tsy = 1.f / (float)depth.h;
    fc.dW = (float)depth.w; fc.dH = (float)depth.h; fc.nW = fc.dW; fc.nH = fc.dH;
    fc.vpRowNorms = make_float3(rowNorm(0), rowNorm(1), rowNorm(3));
    ImgView gi = depth; // the GI images share the depth texture's grid: only their size is used
#define PLR_MARK_ARGS gi, gi, gi, gi, depth, depth, c.global, (const float*)scratch, (const uint4*)nullptr, filterIndex, w, h, y0, x0, tilesX, tilesY, chunkRows, (uint32_t*)nullptr, \
                      (uint32_t)validLo, (uint32_t)std::max(validHi - validLo, 0), (uint32_t)validLoX, (uint32_t)std::max(validHiX - validLoX, 0), 0, fc, (uint32_t*)byteMap, bytePitch, 0, (uint32_t*)byteMap2
    if (depth.fmt == F_R16F) spatialFilterFastKernel<F_R16F, TXv, true, true, false, true><<<grid, 256, 0, c.stream>>>(PLR_MARK_ARGS);
    else spatialFilterFastKernel<F_D32, TXv, true, true, false, true><<<grid, 256, 0, c.stream>>>(PLR_MARK_ARGS);
#undef PLR_MARK_ARGS
    PLR_CHECK_LAUNCH(c);
    const uint32_t words = rowWords * (uint32_t)depth.h;
    requestBytesToBitsKernel<<<dim3(divUp(words, 256u), second ? 2u : 1u), 256, 0, c.stream>>>(byteMap, (uint32_t*)c.sbuf[6].ptr, byteMap2, second ? (uint32_t*)second->sbuf[6].ptr : nullptr,
                                                                                              bytePitch, rowWords, words);
    PLR_CHECK_LAUNCH(c);
    return 0;
}
static int launchGiSampleRequests(const PassCtx& c) { return launchGiSampleRequestsImpl(c, nullptr); }
// pass fusion: the request passes of filter 0 and filter 1, recorded back to back over the same rectangle and depth texture, as ONE launch (they share the per-pixel
// frame - three exact world positions - which is most of what a wave pays before its samples)
static int launchGiSampleRequestsPair(const PassCtx* const* ctxs, size_t count) {
    if (count != 2) return kUseGeneralKernel;
    const PassCtx &a = *ctxs[0], &b = *ctxs[1];
    if (a.specInt(0, 0) != 0 || b.specInt(0, 0) != 1 || !a.hasSampled(4) || !b.hasSampled(4) || !a.hasSbuf(6) || !b.hasSbuf(6)) return kUseGeneralKernel;
    if (a.sampled[4].ptr != b.sampled[4].ptr || a.sbuf[6].ptr == b.sbuf[6].ptr || std::memcmp(a.base, b.base, sizeof(a.base)) || std::memcmp(a.dispatch, b.dispatch, sizeof(a.dispatch)) ||
        std::memcmp(a.validRows, b.validRows, sizeof(a.validRows)) || std::memcmp(a.validCols, b.validCols, sizeof(a.validCols)))
        return kUseGeneralKernel;
    return launchGiSampleRequestsImpl(a, &b);
}
PLR_REGISTER_FUSION("giSampleRequests x 2", launchGiSampleRequestsPair, "giSampleRequests.comp", "giSampleRequests.comp");
// PLR_MATH_FAST only: the marks are the texels THIS file's filter kernel lands on; the exact set's filter orders its arithmetic differently and a sample on a texel
// boundary may land next door (plr_set_math_mode(PLR_MATH_EXACT) + request lists fails loudly: no exact kernel for this shader)
PLR_REGISTER_SHADER_FAST("giSampleRequests.comp", launchGiSampleRequests);

// ---- the producers of the filter's input write the packed texels of the rows they produce (PassCtx::consumer; fused_gi.h)
PLR_REGISTER_CONSUMER_LINK(trace_spatial, "sdfDiffuseTrace.comp", "filterIndirectDiffuseSpatial.comp");
PLR_REGISTER_CONSUMER_LINK(temporal_spatial, "filterIndirectDiffuseTemporal.comp", "filterIndirectDiffuseSpatial.comp");

} // namespace plr
