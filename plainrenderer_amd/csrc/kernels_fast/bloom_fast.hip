// PLR_MATH_FAST variant of bloomUpsample.comp:19-57 (exact variant: kernels_exact/bloom.hip).
//
// The 9-tap tent (weights .25/.125/.0625 at 0, +-blurRadius texels) is the outer product of the 1D tent (.5, .25, .25), the
// 4-tap box of the previous mip (4 x .25 at +-0.5 texel) is the outer product of (.5, .5), and bilinear filtering is itself
// separable. So   target = sum_rows wy(r) * [ sum_cols wx(c) * source(c, r) ]  +  the same for the previous mip:
// a block first builds the horizontally filtered rows it needs in LDS (6 + 4 texel decodes per row entry), then every output
// combines 6 + 4 LDS rows vertically. 64x16 outputs per block read ~7 texels per output instead of 52 and the vector work drops
// from 52 to ~10 multiply-adds per output. Tap positions, 8-bit sub-texel weights and clamp-to-edge addressing are evaluated
// per tap exactly as in the sampler contract; only the association order of the weighted sum differs from the exact kernel.
#include "../backend.h"
#include "../device/shading_common.h"
#include <cstdlib>

namespace plr {
namespace fastbloom {

constexpr int TW = 64, TH = 16;
constexpr int ROWS_A = 16, ROWS_B = 12; // rows of the source / previous mip one 16-row output tile can touch (8 + 2*ceil(radius) + 2 / 8 + 3)

struct Tap { int i0, i1; float w0, w1; };

// bilinear footprint of one 1D tap at normalised coordinate u (size n), scaled by the tap's filter weight h
PLR_DI Tap tap1D(float u, int n, float h) {
    int i0; float a;
    linearCoord(u * (float)n, &i0, &a);
    Tap t;
    t.i0 = clampi(i0, n); t.i1 = clampi(i0 + 1, n);
    t.w0 = (1.f - a) * h; t.w1 = a * h;
    return t;
}

template <bool LOWEST>
PLR_DI void bloomUpsampleFastBlock(ImgView source, ImgView previous, ImgView target, float blurRadius, int coverW, int coverH, int yBase, int xBase, unsigned bx, unsigned by) {
    __shared__ float HA[ROWS_A][3][TW];
    __shared__ float HB[ROWS_B][3][TW];
    const int t = (int)threadIdx.x;
    const int lx = t & 63, lyBase = t >> 6;
    const int x = xBase + (int)bx * TW + lx; // columns [xBase, coverW) (tile rendering: PassCtx::colSpan; a multiple of 8)
    const int y0 = yBase + (int)by * TH;
    const float tsx = 1.f / (float)source.w, tsy = 1.f / (float)source.h;
    const float sx = blurRadius * tsx, sy = blurRadius * tsy;
    const float invTW = 1.f / (float)target.w, invTH = 1.f / (float)target.h;
    const int yLast = min(y0 + TH, coverH) - 1;

    // row ranges touched by this tile (tap positions are monotonic in y)
    const float vFirst = ((float)y0 + 0.5f) * invTH, vLast = ((float)yLast + 0.5f) * invTH;
    const int rowA0 = tap1D(vFirst + sy * -1.f, source.h, 1.f).i0;
    const int rowA1 = tap1D(vLast + sy * 1.f, source.h, 1.f).i1;
    const int nRowsA = min(rowA1 - rowA0 + 1, ROWS_A);
    int rowB0 = 0, nRowsB = 0;
    if (!LOWEST) {
        rowB0 = tap1D(vFirst + tsy * -0.5f, previous.h, 1.f).i0;
        const int rowB1 = tap1D(vLast + tsy * 0.5f, previous.h, 1.f).i1;
        nRowsB = min(rowB1 - rowB0 + 1, ROWS_B);
    }

    // ---- horizontal pass: this lane's column taps, then one LDS entry per (row, column)
    if (x < coverW) {
        const float u = ((float)x + 0.5f) * invTW;
        const Tap a0 = tap1D(u, source.w, 0.5f), a1 = tap1D(u + sx * 1.f, source.w, 0.25f), a2 = tap1D(u + sx * -1.f, source.w, 0.25f);
        const uint32_t* src = (const uint32_t*)source.ptr;
        for (int r = lyBase; r < nRowsA; r += 4) {
            const uint32_t* row = src + (size_t)(rowA0 + r) * (size_t)source.w;
            vec3 acc = unpackR11G11B10(row[a0.i0]) * a0.w0 + unpackR11G11B10(row[a0.i1]) * a0.w1;
            acc = acc + unpackR11G11B10(row[a1.i0]) * a1.w0 + unpackR11G11B10(row[a1.i1]) * a1.w1;
            acc = acc + unpackR11G11B10(row[a2.i0]) * a2.w0 + unpackR11G11B10(row[a2.i1]) * a2.w1;
            HA[r][0][lx] = acc.x; HA[r][1][lx] = acc.y; HA[r][2][lx] = acc.z;
        }
        if (!LOWEST) {
            const Tap b0 = tap1D(u + tsx * 0.5f, previous.w, 0.5f), b1 = tap1D(u + tsx * -0.5f, previous.w, 0.5f);
            const uint32_t* prv = (const uint32_t*)previous.ptr;
            for (int r = lyBase; r < nRowsB; r += 4) {
                const uint32_t* row = prv + (size_t)(rowB0 + r) * (size_t)previous.w;
                const vec3 acc = unpackR11G11B10(row[b0.i0]) * b0.w0 + unpackR11G11B10(row[b0.i1]) * b0.w1 + unpackR11G11B10(row[b1.i0]) * b1.w0 +
                                 unpackR11G11B10(row[b1.i1]) * b1.w1;
                HB[r][0][lx] = acc.x; HB[r][1][lx] = acc.y; HB[r][2][lx] = acc.z;
            }
        }
    }
    __syncthreads();
    if (x >= coverW) return;

    // ---- vertical pass
#pragma unroll
    for (int k = 0; k < TH / 4; k++) {
        const int y = y0 + lyBase + 4 * k;
        if (y >= coverH) break;
        const float v = ((float)y + 0.5f) * invTH;
        const Tap c0 = tap1D(v, source.h, 0.5f), c1 = tap1D(v + sy * 1.f, source.h, 0.25f), c2 = tap1D(v + sy * -1.f, source.h, 0.25f);
        vec3 color(0.f);
        auto rowA = [&](int r, float w) {
            r = min(max(r - rowA0, 0), ROWS_A - 1);
            color = color + vec3(HA[r][0][lx], HA[r][1][lx], HA[r][2][lx]) * w;
        };
        rowA(c0.i0, c0.w0); rowA(c0.i1, c0.w1); rowA(c1.i0, c1.w0); rowA(c1.i1, c1.w1); rowA(c2.i0, c2.w0); rowA(c2.i1, c2.w1);
        if (!LOWEST) {
            const Tap d0 = tap1D(v + tsy * 0.5f, previous.h, 0.5f), d1 = tap1D(v + tsy * -0.5f, previous.h, 0.5f);
            auto rowB = [&](int r, float w) {
                r = min(max(r - rowB0, 0), ROWS_B - 1);
                color = color + vec3(HB[r][0][lx], HB[r][1][lx], HB[r][2][lx]) * w;
            };
            rowB(d0.i0, d0.w0); rowB(d0.i1, d0.w1); rowB(d1.i0, d1.w0); rowB(d1.i1, d1.w1);
        }
        ((uint32_t*)target.ptr)[(size_t)y * (size_t)target.w + x] = packR11G11B10(color);
    }
}
template <bool LOWEST>
__global__ __launch_bounds__(256) void bloomUpsampleFastKernel(ImgView source, ImgView previous, ImgView target, float blurRadius, int coverW, int coverH, int yBase, int xBase) { bloomUpsampleFastBlock<LOWEST>(source, previous, target, blurRadius, coverW, coverH, yBase, xBase, blockIdx.x, blockIdx.y); }

// ---- 2x2 outputs per thread for the regular case (target = exactly twice the source and the previous mip; footprint within +-2)
// An output pixel X = 2k + p samples the half-resolution images at k + 0.25 + 0.5 p (texel units), so the three tent taps and the
// two box taps have the same sub-texel weights for every pixel of one parity: per axis the filter is a 5-entry (tent, texels
// k-2 .. k+2) and a 3-entry (box, texels k-1 .. k+1) weight vector per parity, computed once on the host with the sampler's 8-bit
// weight rule. A thread filters a 5x5 / 3x3 texel footprint horizontally for both x parities, then vertically for both y
// parities: 34 texel decodes and 13 loads per four outputs (the LDS kernel above: ~7 decodes per output plus the per-pixel tap
// set-up, ~400 VALU instructions per output; this one ~175).
struct ParityWeights { float tent[2][5]; float box[2][3]; };

// QY quad rows per thread (2x2 outputs each): consecutive quad rows share four of their five source rows (two of three previous-mip rows),
// so the decode and the horizontal filter of a source row are done once for both. Every output accumulates its rows in the same order
// whatever QY is: the result does not depend on it.
template <int QY>
PLR_DI void bloomUpsampleQuadBlock(ImgView source, ImgView previous, ImgView target, ParityWeights pw, bool lowest, int coverW, int coverH,
                                                               int yBase, int xBase, unsigned bx, unsigned by) {
    // thread -> QY vertically adjacent 2x2 output quads; a wave covers 128 x 2 QY outputs
    const int k = (xBase >> 1) + (int)(bx * 64u + (threadIdx.x & 63u));
    const int m = (yBase >> 1) + (int)(by * 4u + (threadIdx.x >> 6)) * QY;
    const int X = 2 * k, Y = 2 * m;
    if (X >= coverW || Y >= coverH) return;
    float wt[2][5], wb[2][3]; // in vector registers: see the strip kernel below
#pragma unroll
    for (int p = 0; p < 2; p++) {
#pragma unroll
        for (int j = 0; j < 5; j++) { wt[p][j] = pw.tent[p][j]; asm volatile("" : "+v"(wt[p][j])); }
#pragma unroll
        for (int j = 0; j < 3; j++) { wb[p][j] = pw.box[p][j]; asm volatile("" : "+v"(wb[p][j])); }
    }
    const int sw = source.w, sh = source.h;
    const uint32_t* src = (const uint32_t*)source.ptr;
    const bool interiorX = k >= 2 && k + 2 < sw;
    vec3 acc[QY][2][2]; // [quad row][y parity][x parity]
#pragma unroll
    for (int q = 0; q < QY; q++)
#pragma unroll
        for (int i = 0; i < 4; i++) acc[q][i >> 1][i & 1] = vec3(0.f);
    // horizontally filtered source rows m-2 .. m+2+(QY-1); row r is row r-q of quad row q
#pragma unroll
    for (int r = 0; r < 4 + QY; r++) {
        const uint32_t* row = src + (size_t)clampi(m - 2 + r, sh) * (size_t)sw;
        uint32_t t[5];
        if (interiorX) { uint4 v; __builtin_memcpy(&v, row + (k - 2), 16); t[0] = v.x; t[1] = v.y; t[2] = v.z; t[3] = v.w; t[4] = row[k + 2]; }
        else { for (int c = 0; c < 5; c++) t[c] = row[clampi(k - 2 + c, sw)]; }
        vec3 h0(0.f), h1(0.f);
#pragma unroll
        for (int c = 0; c < 5; c++) {
            const vec3 col = unpackR11G11B10(t[c]);
            h0 = h0 + col * wt[0][c];
            h1 = h1 + col * wt[1][c];
        }
#pragma unroll
        for (int q = 0; q < QY; q++) {
            const int j = r - q;
            if (j < 0 || j > 4) continue;
            acc[q][0][0] = acc[q][0][0] + h0 * wt[0][j]; acc[q][0][1] = acc[q][0][1] + h1 * wt[0][j];
            acc[q][1][0] = acc[q][1][0] + h0 * wt[1][j]; acc[q][1][1] = acc[q][1][1] + h1 * wt[1][j];
        }
    }
    if (!lowest) {
        const int pwid = previous.w, phei = previous.h;
        const uint32_t* prv = (const uint32_t*)previous.ptr;
        const bool interiorP = k >= 1 && k + 2 < pwid;
#pragma unroll
        for (int r = 0; r < 2 + QY; r++) {
            const uint32_t* row = prv + (size_t)clampi(m - 1 + r, phei) * (size_t)pwid;
            uint32_t t[3];
            if (interiorP) { uint4 v; __builtin_memcpy(&v, row + (k - 1), 16); t[0] = v.x; t[1] = v.y; t[2] = v.z; }
            else { for (int c = 0; c < 3; c++) t[c] = row[clampi(k - 1 + c, pwid)]; }
            vec3 h0(0.f), h1(0.f);
#pragma unroll
            for (int c = 0; c < 3; c++) {
                const vec3 col = unpackR11G11B10(t[c]);
                h0 = h0 + col * wb[0][c];
                h1 = h1 + col * wb[1][c];
            }
#pragma unroll
            for (int q = 0; q < QY; q++) {
                const int j = r - q;
                if (j < 0 || j > 2) continue;
                acc[q][0][0] = acc[q][0][0] + h0 * wb[0][j]; acc[q][0][1] = acc[q][0][1] + h1 * wb[0][j];
                acc[q][1][0] = acc[q][1][0] + h0 * wb[1][j]; acc[q][1][1] = acc[q][1][1] + h1 * wb[1][j];
            }
        }
    }
    uint32_t* out = (uint32_t*)target.ptr;
#pragma unroll
    for (int q = 0; q < QY; q++)
#pragma unroll
        for (int py = 0; py < 2; py++) {
            const int y = Y + 2 * q + py;
            if (y >= coverH || y < yBase) continue;
            uint32_t* orow = out + (size_t)y * (size_t)target.w + X;
            const uint32_t p0 = packR11G11B10(acc[q][py][0]), p1 = packR11G11B10(acc[q][py][1]);
            if (X + 1 < coverW) *(uint2*)orow = make_uint2(p0, p1); // X is even and the row pitch is even: 8-byte aligned
            else orow[0] = p0;
        }
}
template <int QY>
__global__ __launch_bounds__(256) void bloomUpsampleQuadKernel(ImgView source, ImgView previous, ImgView target, ParityWeights pw, bool lowest, int coverW, int coverH,
                                                               int yBase, int xBase) { bloomUpsampleQuadBlock<QY>(source, previous, target, pw, lowest, coverW, coverH, yBase, xBase, blockIdx.x, blockIdx.y); }

// ---- the same filter with the lanes of a wave holding neighbouring source columns (the quad kernel's 175 instructions per output are
// mostly texel decodes: each lane decodes the 5 x 6 + 3 x 4 texels of its own footprint although its neighbours decode four fifths of the
// same texels). Lane l of a wave holds source column k0 - 2 + l: it loads and decodes ONE texel per source row, the horizontal filter takes
// the neighbouring columns from the neighbouring lanes (DPP wave_shr / wave_shl, no LDS), lanes 2 .. 61 own two output columns each.
// A wave walks kStripRows quad rows down its strip, all its loads issued up front. Every sum is accumulated in the quad kernel's order:
// the results are the same bits.
constexpr int kStripCols = 60, kStripRows = 8;
PLR_DI float laneLeft(float v) { return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0x138, 0xf, 0xf, true)); }  // wave_shr:1
PLR_DI float laneRight(float v) { return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0x130, 0xf, 0xf, true)); } // wave_shl:1
PLR_DI vec3 laneLeft(const vec3& v) { return vec3(laneLeft(v.x), laneLeft(v.y), laneLeft(v.z)); }
PLR_DI vec3 laneRight(const vec3& v) { return vec3(laneRight(v.x), laneRight(v.y), laneRight(v.z)); }

template <bool LOWEST>
__global__ __launch_bounds__(256) void bloomUpsampleStripKernel(ImgView source, ImgView previous, ImgView target, ParityWeights pw, int coverW, int coverH, int yBase, int xBase) {
    const int lane = (int)(threadIdx.x & 63u), wave = (int)(threadIdx.x >> 6);
    const int kc = (xBase >> 1) + (int)blockIdx.x * kStripCols - 2 + lane;                          // the source column this lane holds
    const int m0 = (yBase >> 1) + ((int)blockIdx.y * 4 + wave) * kStripRows;         // first quad row (= source row) of the wave
    if (2 * m0 >= coverH) return; // wave-uniform
    // the filter weights live in vector registers: a multiply-add that reads a scalar register issues at half rate (tools/valu_rates.hip),
    // and nearly every instruction of this kernel is a multiply-add by one of these sixteen values
    float wt[2][5], wb[2][3];
#pragma unroll
    for (int p = 0; p < 2; p++) {
#pragma unroll
        for (int j = 0; j < 5; j++) { wt[p][j] = pw.tent[p][j]; asm volatile("" : "+v"(wt[p][j])); }
#pragma unroll
        for (int j = 0; j < 3; j++) { wb[p][j] = pw.box[p][j]; asm volatile("" : "+v"(wb[p][j])); }
    }
    const int sw = source.w, sh = source.h;
    const uint32_t* src = (const uint32_t*)source.ptr + clampi(kc, sw);
    uint32_t ta[kStripRows + 4], tb[kStripRows + 2];
#pragma unroll
    for (int r = 0; r < kStripRows + 4; r++) ta[r] = src[(size_t)clampi(m0 - 2 + r, sh) * (size_t)sw];
    if (!LOWEST) {
        const int pwid = previous.w, phei = previous.h;
        const uint32_t* prv = (const uint32_t*)previous.ptr + clampi(kc, pwid);
#pragma unroll
        for (int r = 0; r < kStripRows + 2; r++) tb[r] = prv[(size_t)clampi(m0 - 1 + r, phei) * (size_t)pwid];
    }
    // horizontally filtered rows for the two x parities: tent rows m0-2 .. , box rows m0-1 ..
    vec3 h0[kStripRows + 4], h1[kStripRows + 4], g0[kStripRows + 2], g1[kStripRows + 2];
    auto tentRow = [&](int r) {
        const vec3 c2 = unpackR11G11B10(ta[r]);
        const vec3 c1 = laneLeft(c2), c0 = laneLeft(c1), c3 = laneRight(c2), c4 = laneRight(c3);
        vec3 a(0.f), b(0.f);
        a = a + c0 * wt[0][0]; b = b + c0 * wt[1][0];
        a = a + c1 * wt[0][1]; b = b + c1 * wt[1][1];
        a = a + c2 * wt[0][2]; b = b + c2 * wt[1][2];
        a = a + c3 * wt[0][3]; b = b + c3 * wt[1][3];
        a = a + c4 * wt[0][4]; b = b + c4 * wt[1][4];
        h0[r] = a; h1[r] = b;
    };
    auto boxRow = [&](int r) {
        const vec3 c1 = unpackR11G11B10(tb[r]);
        const vec3 c0 = laneLeft(c1), c2 = laneRight(c1);
        vec3 a(0.f), b(0.f);
        a = a + c0 * wb[0][0]; b = b + c0 * wb[1][0];
        a = a + c1 * wb[0][1]; b = b + c1 * wb[1][1];
        a = a + c2 * wb[0][2]; b = b + c2 * wb[1][2];
        g0[r] = a; g1[r] = b;
    };
#pragma unroll
    for (int r = 0; r < 4; r++) tentRow(r);
    if (!LOWEST) {
#pragma unroll
        for (int r = 0; r < 2; r++) boxRow(r);
    }
    const bool outputLane = lane >= 2 && lane < 2 + kStripCols && 2 * kc < coverW;
    const bool both = 2 * kc + 1 < coverW;
    uint32_t* out = (uint32_t*)target.ptr + 2 * kc;
#pragma unroll
    for (int q = 0; q < kStripRows; q++) {
        tentRow(q + 4);
        if (!LOWEST) boxRow(q + 2);
#pragma unroll
        for (int py = 0; py < 2; py++) {
            vec3 a(0.f), b(0.f);
#pragma unroll
            for (int j = 0; j < 5; j++) { a = a + h0[q + j] * wt[py][j]; b = b + h1[q + j] * wt[py][j]; }
            if (!LOWEST) {
#pragma unroll
                for (int j = 0; j < 3; j++) { a = a + g0[q + j] * wb[py][j]; b = b + g1[q + j] * wb[py][j]; }
            }
            const int y = 2 * (m0 + q) + py;
            if (y < coverH && outputLane) { // y >= yBase: m0 starts at yBase / 2 and yBase is even
                uint32_t* orow = out + (size_t)y * (size_t)target.w;
                const uint32_t p0 = packR11G11B10(a), p1 = packR11G11B10(b);
                if (both) *(uint2*)orow = make_uint2(p0, p1); // 2 kc is even and the row pitch is even: 8-byte aligned
                else orow[0] = p0;
            }
        }
    }
}

// the 1D footprint of one tap (sampler rule of image.h:linearCoord) accumulated into weights over texels base .. base + n - 1
static bool accumulateTap(float position, float weight, int base, int n, float* w) {
    const float sane = std::min(std::max(position, -1.0e6f), 1.0e6f);
    const int ti = (int)std::floor((sane - 0.5f) * 256.0f + 0.5f);
    const int i0 = ti >> 8;
    const float a = (float)(ti & 255) * (1.0f / 256.0f);
    const int j0 = i0 - base, j1 = i0 + 1 - base;
    if ((1.f - a) != 0.f) { if (j0 < 0 || j0 >= n) return false; w[j0] += weight * (1.f - a); }
    if (a != 0.f) { if (j1 < 0 || j1 >= n) return false; w[j1] += weight * a; }
    return true;
}

static bool makeParityWeights(float blurRadius, ParityWeights* pw) {
    const int k = 8; // any interior texel: the weights do not depend on it
    for (int p = 0; p < 2; p++) {
        const float s = (float)k + 0.25f + 0.5f * (float)p;
        for (float& v : pw->tent[p]) v = 0.f;
        for (float& v : pw->box[p]) v = 0.f;
        bool ok = accumulateTap(s, 0.5f, k - 2, 5, pw->tent[p]) && accumulateTap(s + blurRadius, 0.25f, k - 2, 5, pw->tent[p]) &&
                  accumulateTap(s - blurRadius, 0.25f, k - 2, 5, pw->tent[p]) && accumulateTap(s + 0.5f, 0.5f, k - 1, 3, pw->box[p]) &&
                  accumulateTap(s - 0.5f, 0.5f, k - 1, 3, pw->box[p]);
        if (!ok) return false;
    }
    return true;
}

// what one bloomUpsample execution launches: the kernel kind, its grid and its arguments (shared by the single launch and the chain launch below)
enum UpKind { UP_NOTHING = 0, UP_STRIP, UP_QUAD, UP_GENERIC };
constexpr int kQuadQY = 2; // 4: mip 0 35.0 vs 36.6 us, but the small mips lose more (fewer, longer waves)
struct UpPlan {
    ImgView source, previous, target;
    ParityWeights pw;
    float blurRadius;
    int lowest, w, h, yBase, x0; // columns [x0, w), rows [yBase, h)
    int kind;
    unsigned gridX, gridY;
};
static int planUp(const PassCtx& c, UpPlan* out) {
    if (int rc = c.needStorage(0, F_R11G11B10, "bloomUpsample target")) return rc;
    if (int rc = c.needSampled(2, F_R11G11B10, "bloomUpsample source")) return rc;
    const bool lowest = c.specBool(0, false);
    if (!lowest) if (int rc = c.needSampled(1, F_R11G11B10, "bloomUpsample targetPreviousMip")) return rc;
    if (c.push.size() < 4) return c.fail(-1, "bloomUpsample: push constant blurRadius missing");
    float blurRadius;
    std::memcpy(&blurRadius, c.push.data(), 4);
    const ImgView& target = c.storage[0];
    const ImgView& source = c.sampled[2];
    const PassCtx::RowSpan rs = c.rowSpan(target.h);
    const PassCtx::ColSpan cs = c.colSpan(target.w);
    const int w = cs.x1, x0 = cs.x0, h = rs.y1, yBase = rs.y0;
    UpPlan& p = *out;
    p.source = source; p.previous = lowest ? source : c.sampled[1]; p.target = target;
    p.blurRadius = blurRadius; p.lowest = lowest; p.w = w; p.h = h; p.yBase = yBase; p.x0 = x0;
    p.kind = UP_NOTHING; p.gridX = p.gridY = 0;
    if (w <= x0 || h <= yBase) return 0;
    // the LDS row budget assumes the reference's configuration: source = next smaller mip (>= half the target height) and a
    // blur radius of at most 3 source texels; anything else takes the general (exact-order) kernel
    const bool fits = blurRadius >= 0.f && blurRadius <= 3.f && source.h * 2 + 1 >= target.h && (lowest || c.sampled[1].h * 2 + 1 >= target.h);
    if (!fits) return kUseGeneralKernel;
    // regular case: one thread per 2x2 outputs with per-parity weight vectors (needs even row base so quads do not straddle the dispatch)
    const bool regular = target.w == 2 * source.w && target.h == 2 * source.h && (lowest || (c.sampled[1].w == source.w && c.sampled[1].h == source.h)) &&
                         source.w >= 5 && (yBase & 1) == 0 && (target.w & 1) == 0 && makeParityWeights(blurRadius, &p.pw);
    if (regular) {
        // the strip kernel needs enough waves to fill the SIMDs (a wave is 60 x 8 quads): mip 0 of a 4K frame has 4320, mip 1 1088
        static const int stripMinWaves = std::getenv("PLR_BLOOM_STRIP_MIN_WAVES") ? atoi(std::getenv("PLR_BLOOM_STRIP_MIN_WAVES")) : 2048;
        const unsigned stripWaves = divUp((unsigned)divUp((unsigned)(w - x0), 2u), (unsigned)kStripCols) * divUp((unsigned)divUp((unsigned)(h - yBase), 2u), (unsigned)kStripRows);
        if ((int)stripWaves >= stripMinWaves) {
            p.kind = UP_STRIP;
            p.gridX = divUp((unsigned)divUp((unsigned)(w - x0), 2u), (unsigned)kStripCols); p.gridY = divUp((unsigned)divUp((unsigned)(h - yBase), 2u), 4u * kStripRows);
            return 0;
        }
        p.kind = UP_QUAD;
        p.gridX = divUp((unsigned)divUp((unsigned)(w - x0), 2u), 64u); p.gridY = divUp((unsigned)divUp((unsigned)(h - yBase), 2u), 4u * kQuadQY);
        return 0;
    }
    p.kind = UP_GENERIC;
    p.gridX = divUp((unsigned)(w - x0), (unsigned)TW); p.gridY = divUp((unsigned)(h - yBase), (unsigned)TH);
    return 0;
}

static int launch(const PassCtx& c) {
    UpPlan p;
    if (int rc = planUp(c, &p)) return rc;
    const dim3 grid(p.gridX, p.gridY);
    switch (p.kind) {
        case UP_NOTHING: return 0;
        case UP_STRIP:
            if (p.lowest) bloomUpsampleStripKernel<true><<<grid, 256, 0, c.stream>>>(p.source, p.source, p.target, p.pw, p.w, p.h, p.yBase, p.x0);
            else bloomUpsampleStripKernel<false><<<grid, 256, 0, c.stream>>>(p.source, p.previous, p.target, p.pw, p.w, p.h, p.yBase, p.x0);
            break;
        case UP_QUAD: bloomUpsampleQuadKernel<kQuadQY><<<grid, 256, 0, c.stream>>>(p.source, p.previous, p.target, p.pw, p.lowest != 0, p.w, p.h, p.yBase, p.x0); break;
        default:
            if (p.lowest) bloomUpsampleFastKernel<true><<<grid, 256, 0, c.stream>>>(p.source, p.source, p.target, p.blurRadius, p.w, p.h, p.yBase, p.x0);
            else bloomUpsampleFastKernel<false><<<grid, 256, 0, c.stream>>>(p.source, p.previous, p.target, p.blurRadius, p.w, p.h, p.yBase, p.x0);
    }
    PLR_CHECK_LAUNCH(c);
    return 0;
}

// ------------------------------------------------------------------------------------------------ bloomDownsample.comp:12-49
// When the source is exactly twice the target in both dimensions, every one of the 13 bilinear taps lands on a texel centre or
// half way between two / four texels, so the filter is a fixed 4x4 stencil over source texels (2x-1 .. 2x+2) x (2y-1 .. 2y+2):
//   the four centre texels weigh 0.125 (the +-0.5 taps) + 0.03125 (a quarter of the centre tap) = 0.15625, the other twelve 0.03125.
// Wide row loads replace up to 36 texel fetches. (The exact kernel's sub-texel weights are quantised to 1/256 from a
// float coordinate; where that rounding lands one step off the ideal 0.5 the two kernels differ by 1/256 of a texel difference.)
// One thread makes a 2x2 block of outputs from a 6x6 block of source texels (rows 4m-1 .. 4m+4, columns 4k-1 .. 4k+4): 9 texel decodes per
// output instead of 16 - the pass is bound by VALU issue. Per source row and output column A = the two inner texels, B = the two outer
// ones; a row is an inner row (centre += A, border += B) for one output row and an outer row (border += A + B) for the other.
PLR_DI void bloomDownsampleFastBlock(ImgView source, ImgView target, int coverW, int coverH, int yBase, int xBase, unsigned bx, unsigned by) {
    const int k = (xBase >> 1) + (int)(bx * 64u + (threadIdx.x & 63u)); // xBase: a multiple of 8 (PassCtx::colSpan)
    const int m = (yBase >> 1) + (int)(by * 4u + (threadIdx.x >> 6));
    const int X = 2 * k, Y = 2 * m;
    if (X >= coverW || Y >= coverH) return;
    const uint32_t* src = (const uint32_t*)source.ptr;
    const int sx = 4 * k - 1;
    const bool interior = sx >= 0 && sx + 5 < source.w;
    vec3 centre[2][2], border[2][2]; // [output row][output column]
#pragma unroll
    for (int i = 0; i < 4; i++) { centre[i >> 1][i & 1] = vec3(0.f); border[i >> 1][i & 1] = vec3(0.f); }
#pragma unroll
    for (int r = 0; r < 6; r++) {
        const uint32_t* row = src + (size_t)clampi(4 * m - 1 + r, source.h) * (size_t)source.w;
        uint32_t t[6];
        if (interior) {
            uint4 v; uint2 u;
            __builtin_memcpy(&v, row + sx, 16);
            __builtin_memcpy(&u, row + sx + 4, 8);
            t[0] = v.x; t[1] = v.y; t[2] = v.z; t[3] = v.w; t[4] = u.x; t[5] = u.y;
        } else { for (int c = 0; c < 6; c++) t[c] = row[clampi(sx + c, source.w)]; }
        vec3 col[6];
#pragma unroll
        for (int c = 0; c < 6; c++) col[c] = unpackR11G11B10(t[c]);
#pragma unroll
        for (int px = 0; px < 2; px++) {
            const vec3 A = col[2 * px + 1] + col[2 * px + 2], B = col[2 * px] + col[2 * px + 3];
#pragma unroll
            for (int q = 0; q < 2; q++) {
                const int j = r - 2 * q; // row j of output row q's 4x4 stencil
                if (j < 0 || j > 3) continue;
                if (j == 1 || j == 2) { centre[q][px] = centre[q][px] + A; border[q][px] = border[q][px] + B; }
                else border[q][px] = border[q][px] + (A + B);
            }
        }
    }
    uint32_t* out = (uint32_t*)target.ptr;
#pragma unroll
    for (int q = 0; q < 2; q++) {
        const int y = Y + q;
        if (y >= coverH || y < yBase) continue;
        const uint32_t p0 = packR11G11B10(centre[q][0] * 0.15625f + border[q][0] * 0.03125f), p1 = packR11G11B10(centre[q][1] * 0.15625f + border[q][1] * 0.03125f);
        uint32_t* orow = out + (size_t)y * (size_t)target.w + X;
        if (X + 1 < coverW && (target.w & 1) == 0) *(uint2*)orow = make_uint2(p0, p1);
        else { orow[0] = p0; if (X + 1 < coverW) orow[1] = p1; }
    }
}
__global__ __launch_bounds__(256) void bloomDownsampleFastKernel(ImgView source, ImgView target, int coverW, int coverH, int yBase, int xBase) { bloomDownsampleFastBlock(source, target, coverW, coverH, yBase, xBase, blockIdx.x, blockIdx.y); }

// ---- any source / target size (the one odd level of a 4K chain, 135 -> 68 rows: 8 k outputs, a fraction of a wave per CU, so the launch is one
// wave's latency). The exact kernel fetches a tap's texels only where the weight is not zero - thirteen taps of up to four dependent, branch-guarded
// loads one after the other (10.6 us for 8 k outputs). Here all 52 loads of a thread are issued first and the zero-weight rule is applied as a select
// on the decoded value (a texel with weight 0 must not contribute even if it is not finite).
struct TapFetch { uint32_t t00, t10, t01, t11; float w00, w10, w01, w11; };
PLR_DI TapFetch bloomTapFetch(const ImgView& im, float u, float v) {
    int i0, j0; float a, b;
    linearCoord(u * (float)im.w, &i0, &a);
    linearCoord(v * (float)im.h, &j0, &b);
    const uint32_t* base = (const uint32_t*)im.ptr;
    const int x0 = clampi(i0, im.w), x1 = clampi(i0 + 1, im.w);
    const uint32_t r0 = __umul24((uint32_t)clampi(j0, im.h), (uint32_t)im.w), r1 = __umul24((uint32_t)clampi(j0 + 1, im.h), (uint32_t)im.w);
    TapFetch f;
    f.t00 = base[r0 + x0]; f.t10 = base[r0 + x1]; f.t01 = base[r1 + x0]; f.t11 = base[r1 + x1];
    f.w00 = (1.f - a) * (1.f - b); f.w10 = a * (1.f - b); f.w01 = (1.f - a) * b; f.w11 = a * b;
    return f;
}
PLR_DI vec3 bloomTapValue(const TapFetch& f) {
    // the exact kernel's accumulation order: t00*w00 + t10*w10 + t01*w01 + t11*w11, zero-weight terms absent
    vec3 r = unpackR11G11B10(f.t00) * f.w00;
    const vec3 c10 = unpackR11G11B10(f.t10) * f.w10, c01 = unpackR11G11B10(f.t01) * f.w01, c11 = unpackR11G11B10(f.t11) * f.w11;
    r = f.w10 != 0.f ? r + c10 : r;
    r = f.w01 != 0.f ? r + c01 : r;
    r = f.w11 != 0.f ? r + c11 : r;
    return r;
}
PLR_DI void bloomDownsampleAnySizeBlock(ImgView source, ImgView target, int coverW, int coverH, int yBase, int xBase, unsigned bx, unsigned by) {
    const int x = xBase + (int)(bx * 64u + (threadIdx.x & 63u));
    const int y = yBase + (int)(by * 4u + (threadIdx.x >> 6));
    if (x >= coverW || y >= coverH) return;
    // correctly rounded quotients / reciprocals (Newton step on v_rcp_f32), as the shader's divisions: the taps sit on 1/256 sub-texel weight steps
    auto quot = [](float a, float b) { const float r = __builtin_amdgcn_rcpf(b), q = a * r; return __builtin_fmaf(__builtin_fmaf(-b, q, a), r, q); };
    const float uvx = quot((float)x + 0.5f, (float)target.w), uvy = quot((float)y + 0.5f, (float)target.h);
    const float tsx = quot(1.f, (float)source.w), tsy = quot(1.f, (float)source.h);
    // bloomDownsample.comp:12-49: tap offsets in source texels and their weights, in the shader's order
    const float ox[13] = {0.f, 0.5f, 0.5f, -0.5f, -0.5f, 1.5f, -1.5f, 0.f, 0.f, 1.5f, 1.5f, -1.5f, -1.5f};
    const float oy[13] = {0.f, 0.5f, -0.5f, 0.5f, -0.5f, 0.f, 0.f, 1.5f, -1.5f, 1.5f, -1.5f, 1.5f, -1.5f};
    const float wt[13] = {0.125f, 0.125f, 0.125f, 0.125f, 0.125f, 0.0625f, 0.0625f, 0.0625f, 0.0625f, 0.03125f, 0.03125f, 0.03125f, 0.03125f};
    TapFetch f[13];
#pragma unroll
    for (int i = 0; i < 13; i++) f[i] = bloomTapFetch(source, uvx + tsx * ox[i], uvy + tsy * oy[i]);
    vec3 color(0.f);
#pragma unroll
    for (int i = 0; i < 13; i++) color = color + bloomTapValue(f[i]) * wt[i];
    ((uint32_t*)target.ptr)[(size_t)y * (size_t)target.w + x] = packR11G11B10(color);
}
__global__ __launch_bounds__(256) void bloomDownsampleAnySizeKernel(ImgView source, ImgView target, int coverW, int coverH, int yBase, int xBase) { bloomDownsampleAnySizeBlock(source, target, coverW, coverH, yBase, xBase, blockIdx.x, blockIdx.y); }
enum DownKind { DOWN_NOTHING = 0, DOWN_REGULAR, DOWN_ANY_SIZE };
struct DownPlan { ImgView source, target; int w, h, y0, x0, kind; unsigned gridX, gridY; };
static int planDown(const PassCtx& c, DownPlan* out) {
    if (int rc = c.needStorage(0, F_R11G11B10, "bloomDownsample target")) return rc;
    if (int rc = c.needSampled(1, F_R11G11B10, "bloomDownsample source")) return rc;
    const ImgView& target = c.storage[0];
    const ImgView& source = c.sampled[1];
    const PassCtx::RowSpan rs = c.rowSpan(target.h);
    const PassCtx::ColSpan cs = c.colSpan(target.w);
    DownPlan& p = *out;
    p.source = source; p.target = target; p.w = cs.x1; p.x0 = cs.x0; p.h = rs.y1; p.y0 = rs.y0;
    p.kind = DOWN_NOTHING; p.gridX = p.gridY = 0;
    if (p.w <= p.x0 || p.h <= p.y0) return 0;
    if (source.w != 2 * target.w || source.h != 2 * target.h || source.w < 6) { // odd sizes: taps are not on texel centres
        if (source.w < 1 || source.h < 1 || source.w >= (1 << 12) || source.h >= (1 << 12)) return kUseGeneralKernel; // 24-bit texel index arithmetic
        p.kind = DOWN_ANY_SIZE;
        p.gridX = divUp((unsigned)(p.w - p.x0), 64u); p.gridY = divUp((unsigned)(p.h - p.y0), 4u);
        return 0;
    }
    if (p.y0 & 1) return kUseGeneralKernel; // 2x2 output blocks start on even rows
    p.kind = DOWN_REGULAR;
    p.gridX = divUp(divUp((unsigned)(p.w - p.x0), 2u), 64u); p.gridY = divUp(divUp((unsigned)(p.h - p.y0), 2u), 4u);
    return 0;
}

static int launchDown(const PassCtx& c) {
    DownPlan p;
    if (int rc = planDown(c, &p)) return rc;
    if (p.kind == DOWN_NOTHING) return 0;
    const dim3 grid(p.gridX, p.gridY);
    if (p.kind == DOWN_ANY_SIZE) bloomDownsampleAnySizeKernel<<<grid, 256, 0, c.stream>>>(p.source, p.target, p.w, p.h, p.y0, p.x0);
    else bloomDownsampleFastKernel<<<grid, 256, 0, c.stream>>>(p.source, p.target, p.w, p.h, p.y0, p.x0);
    PLR_CHECK_LAUNCH(c);
    return 0;
}

// ------------------------------------------------------------------------------------------------ the small levels of the chain as ONE launch per direction
// Bloom.cpp:56-143 records five downsamples and five upsamples. From the third level on a launch is a few hundred blocks or less and costs its launch boundary,
// not its work (profiles/r04_tail_cost.txt: ten launches at the launch floor). With pass fusion a run of such executions is one PERSISTENT launch: its blocks draw
// tickets - the tiles of level 0, then of level 1, ... - from one counter and run them with the very block bodies of the single launches above (same bits). A finished
// tile is counted on its level's arrival word (release, agent scope); a block holding a tile of level l first waits until level l - 1 is complete (acquire).
// Only blocks that are running hold tickets, so the waits end whatever else occupies the chip (this is the asynchronous tail: the next frame's front runs beside
// it) and however many of the launch's blocks are resident; the last block out zeroes the words for the next launch.
constexpr int kChainMaxLevels = 4;
struct ChainLevel {
    ImgView source, previous, target;
    ParityWeights pw;
    float blurRadius;
    int lowest, w, h, yBase, x0, kind;
    unsigned gridX, tiles;
};
struct ChainParams { ChainLevel level[kChainMaxLevels]; int n; uint32_t* counters; }; // counters: kChainMaxLevels arrival words, the ticket word, the exit word; zero between launches

template <bool UP>
__global__ __launch_bounds__(256) void bloomChainKernel(ChainParams p) {
    __shared__ uint32_t ticketLds;
    uint32_t* const next = p.counters + kChainMaxLevels;     // work tickets handed out
    uint32_t* const exited = p.counters + kChainMaxLevels + 1; // blocks that found no ticket left
    uint32_t total = 0;
    for (int l = 0; l < p.n; l++) total += p.level[l].tiles;
    for (;;) {
        __syncthreads(); // (ticketLds of the previous round has been read by everybody)
        if (threadIdx.x == 0) ticketLds = __hip_atomic_fetch_add(next, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __syncthreads();
        uint32_t t = ticketLds;
        if (t >= total) break;
        int l = 0;
        while (t >= p.level[l].tiles) { t -= p.level[l].tiles; l++; }
        const ChainLevel& L = p.level[l];
        if (l > 0) {
            // Tickets are handed out in order: every tile of level l - 1 is in the hands of a block that is RUNNING (it fetched the ticket), so this wait ends
            // whatever else occupies the chip - a block that has not started yet holds no work.
            if (threadIdx.x == 0) {
                const uint32_t need = p.level[l - 1].tiles;
                while (__hip_atomic_load(&p.counters[l - 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < need) __builtin_amdgcn_s_sleep(2);
            }
            __syncthreads();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent"); // the XCDs' L2s are not coherent with each other: see the level's texels, not stale lines
        }
        const unsigned bx = t % L.gridX, by = t / L.gridX;
        if (UP) {
            if (L.kind == UP_QUAD) bloomUpsampleQuadBlock<kQuadQY>(L.source, L.previous, L.target, L.pw, L.lowest != 0, L.w, L.h, L.yBase, L.x0, bx, by);
            else if (L.lowest) bloomUpsampleFastBlock<true>(L.source, L.source, L.target, L.blurRadius, L.w, L.h, L.yBase, L.x0, bx, by);
            else bloomUpsampleFastBlock<false>(L.source, L.previous, L.target, L.blurRadius, L.w, L.h, L.yBase, L.x0, bx, by);
        } else {
            if (L.kind == DOWN_ANY_SIZE) bloomDownsampleAnySizeBlock(L.source, L.target, L.w, L.h, L.yBase, L.x0, bx, by);
            else bloomDownsampleFastBlock(L.source, L.target, L.w, L.h, L.yBase, L.x0, bx, by);
        }
        if (l + 1 < p.n) { // (nobody waits for the last level)
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            __syncthreads();
            if (threadIdx.x == 0) __hip_atomic_fetch_add(&p.counters[l], 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    if (threadIdx.x == 0) {
        const uint32_t gone = __hip_atomic_fetch_add(exited, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
        if (gone == gridDim.x - 1) // the last block out: every ticket is done and nobody will ask again - zero the words for the next launch
            for (int i = 0; i < kChainMaxLevels + 2; i++) __hip_atomic_store(&p.counters[i], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// MEASURED AND NOT KEPT (profiles/r05_not_kept.txt (5)): OFF unless PLR_BLOOM_CHAIN=1. The 4K frame takes 0.857 ms with the two chain launches against 0.707 ms
// with the ten single ones: the agent-scope release / acquire between levels is a write-back and an invalidate of a whole XCD's L2 (buffer_wbl2 / buffer_inv sc1) per
// tile - that is what makes one XCD's texels visible to another inside a launch - and the L2 it flushes is the one the next frame's front is working in.
// The largest level a chain takes (PLR_BLOOM_CHAIN_MAX_TILES): above this a level has enough blocks to be worth a launch of its own.
static int chainMaxTiles() {
    static const int v = std::getenv("PLR_BLOOM_CHAIN_MAX_TILES") ? atoi(std::getenv("PLR_BLOOM_CHAIN_MAX_TILES")) : 640;
    static const bool on = std::getenv("PLR_BLOOM_CHAIN") && atoi(std::getenv("PLR_BLOOM_CHAIN")) != 0;
    return on ? v : 0;
}
template <bool UP>
static int chainGrid(const PassCtx& c, unsigned maxTiles, unsigned* grid) {
    // blocks resident at once on an empty chip, from the runtime's occupancy calculator (the kernel's registers and LDS), halved for margin
    // (asked once per host thread and device: a thread renders on one device, plr.h)
    static thread_local int cachedDev = -1;
    static thread_local unsigned cachedResident = 0;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return kUseGeneralKernel;
    if (dev != cachedDev) {
        int perCu = 0, cus = 0;
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus < 1 ||
            hipOccupancyMaxActiveBlocksPerMultiprocessor(&perCu, (const void*)bloomChainKernel<UP>, 256, 0) != hipSuccess || perCu < 1)
            return kUseGeneralKernel;
        cachedResident = (unsigned)cus * (unsigned)std::min(perCu, 2);
        cachedDev = dev;
    }
    (void)c;
    *grid = std::max(1u, std::min(maxTiles, cachedResident));
    return 0;
}
template <bool UP>
static int launchChain(const PassCtx* const* ctxs, size_t count) {
    if (count < 2 || count > (size_t)kChainMaxLevels || chainMaxTiles() <= 0) return kUseGeneralKernel;
    ChainParams cp{};
    cp.n = (int)count;
    unsigned maxTiles = 0;
    for (size_t i = 0; i < count; i++) {
        ChainLevel& L = cp.level[i];
        if (UP) {
            UpPlan u;
            if (int rc = planUp(*ctxs[i], &u)) return rc < 0 ? kUseGeneralKernel : rc; // (a broken execution reports its error from its own launch)
            if (u.kind != UP_QUAD && u.kind != UP_GENERIC) return kUseGeneralKernel;
            L.source = u.source; L.previous = u.previous; L.target = u.target; L.pw = u.pw; L.blurRadius = u.blurRadius; L.lowest = u.lowest;
            L.w = u.w; L.h = u.h; L.yBase = u.yBase; L.x0 = u.x0; L.kind = u.kind; L.gridX = u.gridX; L.tiles = u.gridX * u.gridY;
        } else {
            DownPlan d;
            if (int rc = planDown(*ctxs[i], &d)) return rc < 0 ? kUseGeneralKernel : rc;
            if (d.kind == DOWN_NOTHING) return kUseGeneralKernel;
            L.source = d.source; L.previous = d.source; L.target = d.target; L.blurRadius = 0.f; L.lowest = 0;
            L.w = d.w; L.h = d.h; L.yBase = d.y0; L.x0 = d.x0; L.kind = d.kind; L.gridX = d.gridX; L.tiles = d.gridX * d.gridY;
        }
        if (L.tiles == 0 || (int)L.tiles > chainMaxTiles()) return kUseGeneralKernel;
        // a chain: every level reads what the level before it wrote (anything else has no order to keep and is launched pass by pass)
        if (i > 0 && L.source.ptr != cp.level[i - 1].target.ptr) return kUseGeneralKernel;
        maxTiles = std::max(maxTiles, L.tiles);
    }
    const PassCtx& c = *ctxs[0];
    cp.counters = (uint32_t*)c.scratch(64); // zeroed when allocated, zeroed again by the last block of every launch
    if (!cp.counters) return kUseGeneralKernel;
    unsigned grid = 0;
    if (int rc = chainGrid<UP>(c, maxTiles, &grid)) return rc;
    bloomChainKernel<UP><<<grid, 256, 0, c.stream>>>(cp);
    PLR_CHECK_LAUNCH(c);
    return 0;
}

} // namespace fastbloom

static int fastbloom_up_launch(const PassCtx& c) { return fastbloom::launch(c); }
PLR_REGISTER_SHADER_FAST("bloomUpsample.comp", fastbloom_up_launch);
static int fastbloom_down_launch(const PassCtx& c) { return fastbloom::launchDown(c); }
PLR_REGISTER_SHADER_FAST("bloomDownsample.comp", fastbloom_down_launch);
// runs of small levels as one persistent launch (the longest run first; a run that starts on a level too large declines and the next execution starts one)
static int fastbloom_down_chain(const PassCtx* const* ctxs, size_t count) { return fastbloom::launchChain<false>(ctxs, count); }
static int fastbloom_up_chain(const PassCtx* const* ctxs, size_t count) { return fastbloom::launchChain<true>(ctxs, count); }
static int fastbloom_down_chain4(const PassCtx* const* ctxs, size_t count) { return fastbloom_down_chain(ctxs, count); }
static int fastbloom_down_chain3(const PassCtx* const* ctxs, size_t count) { return fastbloom_down_chain(ctxs, count); }
static int fastbloom_down_chain2(const PassCtx* const* ctxs, size_t count) { return fastbloom_down_chain(ctxs, count); }
static int fastbloom_up_chain4(const PassCtx* const* ctxs, size_t count) { return fastbloom_up_chain(ctxs, count); }
static int fastbloom_up_chain3(const PassCtx* const* ctxs, size_t count) { return fastbloom_up_chain(ctxs, count); }
static int fastbloom_up_chain2(const PassCtx* const* ctxs, size_t count) { return fastbloom_up_chain(ctxs, count); }
PLR_REGISTER_FUSION("bloomDownsample x4 (persistent chain)", fastbloom_down_chain4, "bloomDownsample.comp", "bloomDownsample.comp", "bloomDownsample.comp", "bloomDownsample.comp");
PLR_REGISTER_FUSION("bloomDownsample x3 (persistent chain)", fastbloom_down_chain3, "bloomDownsample.comp", "bloomDownsample.comp", "bloomDownsample.comp");
PLR_REGISTER_FUSION("bloomDownsample x2 (persistent chain)", fastbloom_down_chain2, "bloomDownsample.comp", "bloomDownsample.comp");
PLR_REGISTER_FUSION("bloomUpsample x4 (persistent chain)", fastbloom_up_chain4, "bloomUpsample.comp", "bloomUpsample.comp", "bloomUpsample.comp", "bloomUpsample.comp");
PLR_REGISTER_FUSION("bloomUpsample x3 (persistent chain)", fastbloom_up_chain3, "bloomUpsample.comp", "bloomUpsample.comp", "bloomUpsample.comp");
PLR_REGISTER_FUSION("bloomUpsample x2 (persistent chain)", fastbloom_up_chain2, "bloomUpsample.comp", "bloomUpsample.comp");
} // namespace plr
