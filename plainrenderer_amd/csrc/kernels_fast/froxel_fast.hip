// PLR_MATH_FAST variant of volumetricLightingIntegration.comp:15-45 (exact variant and the shader's quirks: kernels/producers.hip).
//
// The pass walks every froxel column front to back: inscattering accumulates, transmittance multiplies. The exact kernel spends its time in
// software exp(): two per slice for the exponential slice depths (volumetricFroxelLighting.inc:33-41) and four for the segment's
// transmittance - ~15 k dependent instructions per column on 225 waves. Here the slice boundaries are one v_exp_f32 each and carried from
// slice to slice, the four exponentials of a slice are ONE (the shader integrates with a grey extinction vec3(it.w): all three channels and
// the transmittance use exp(-it.w * segment)), and the column's loads do not depend on the running sums, so the unrolled loop keeps eight
// slices' texels in flight. One lane per column (x fastest: coalesced 8-byte texels), as many columns as the dispatch covers.
// No discrete decision in this pass: every value of the integrated volume within the half-float tolerance of tests/parity.py.
// (The three per-froxel passes in front of it - material, light scattering, reprojection - are one lane per froxel with coalesced texels in the
//  exact set already; their shadow-map comparison and sub-texel weights are discrete decisions, so they stay on the exact-order arithmetic.)
#include "../backend.h"
#include "../device/shading_common.h"

namespace plr {
namespace fastfroxel {

struct VolSettings { // volumetricFroxelLighting.inc:6-16, 52 bytes
    float windSampleOffset[3], sampleOffset;
    float scatteringCoefficients[3], maxDistance;
    float absorptionCoefficient, baseDensity, densityNoiseRange, densityNoiseScale, phaseFunctionG;
};

PLR_DI float expF(float x) { return __builtin_amdgcn_exp2f(x * 1.44269504089f); }

__global__ __launch_bounds__(64) void volumetricLightingIntegrationFastKernel(ImgView integrationVolume, ImgView inVolume, const VolSettings* __restrict__ sp, int coverW, int coverH) {
    const int i = (int)(blockIdx.x * 64u + threadIdx.x); // one wave per block: 14 k columns spread over all CUs
    if (i >= coverW * coverH) return;
    const int x = i % coverW, y = i / coverW;
    const float maxDistance = sp->maxDistance;
    const int resZ = integrationVolume.d;
    const float invZ = __builtin_amdgcn_rcpf((float)resZ);
    // froxelUVToDepth(uv) = (e^(3 uv) - 1) / (e^3 - 1) * maxDistance
    const float depthScale = maxDistance * (1.f / 19.0855369f);
    const bool columnInInput = x < inVolume.w && y < inVolume.h;
    const uint2* __restrict__ in = (const uint2*)inVolume.ptr + (size_t)y * (size_t)inVolume.w + (size_t)x;
    uint2* __restrict__ out = (uint2*)integrationVolume.ptr + (size_t)y * (size_t)integrationVolume.w + (size_t)x;
    const size_t inSlice = (size_t)inVolume.w * (size_t)inVolume.h, outSlice = (size_t)integrationVolume.w * (size_t)integrationVolume.h;
    float totalR = 0.f, totalG = 0.f, totalB = 0.f, transmittance = 1.f;
    float depthStart = 0.f; // froxelUVToDepth(0) = 0
    for (int z0 = 0; z0 < resZ; z0 += 8) { // (the shader's loop runs one slice past the volume: that slice reads zeros and its store is dropped - nothing observable)
        uint2 t[8];
#pragma unroll
        for (int k = 0; k < 8; k++) { // eight slices' texels in flight before the first is used
            const int z = z0 + k;
            t[k] = (columnInInput && z < inVolume.d && z < resZ) ? in[(size_t)z * inSlice] : make_uint2(0u, 0u);
        }
#pragma unroll
        for (int k = 0; k < 8; k++) {
            const int z = z0 + k;
            if (z >= resZ) break; // uniform
            const float sr = halfBitsToFloat(t[k].x & 0xffffu), sg = halfBitsToFloat(t[k].x >> 16), sb = halfBitsToFloat(t[k].y & 0xffffu), ext = halfBitsToFloat(t[k].y >> 16);
            const float depthEnd = (expF(3.f * ((float)(z + 1) * invZ)) - 1.f) * depthScale;
            const float segmentLength = depthEnd - depthStart;
            depthStart = depthEnd;
            const float e = expF(-ext * segmentLength);
            // integrateInscattering: (s - s e) / max(ext, 1e-5)
            const float kk = (1.f - e) * __builtin_amdgcn_rcpf(__builtin_fmaxf(ext, 0.00001f));
            totalR += sr * kk; totalG += sg * kk; totalB += sb * kk;
            transmittance *= e;
            out[(size_t)z * outSlice] = make_uint2(floatToHalfBits(totalR) | (floatToHalfBits(totalG) << 16), floatToHalfBits(totalB) | (floatToHalfBits(transmittance) << 16));
        }
    }
}

static int launchIntegration(const PassCtx& c) {
    if (!c.hasStorage(0) || c.storage[0].fmt != F_RGBA16F || !c.hasSampled(1) || c.sampled[1].fmt != F_RGBA16F || !c.hasUbuf(2) || c.ubuf[2].size < sizeof(VolSettings)) return kUseGeneralKernel;
    const ImgView& out = c.storage[0];
    const int w = std::min((int)(c.dispatch[0] * 8u), out.w), h = std::min((int)(c.dispatch[1] * 8u), out.h);
    if (w <= 0 || h <= 0 || c.dispatch[2] == 0) return 0;
    volumetricLightingIntegrationFastKernel<<<divUp((unsigned)(w * h), 64u), 64, 0, c.stream>>>(out, c.sampled[1], (const VolSettings*)c.ubuf[2].ptr, w, h);
    PLR_CHECK_LAUNCH(c);
    return 0;
}

} // namespace fastfroxel

static int fastfroxel_integration(const PassCtx& c) { return fastfroxel::launchIntegration(c); }
PLR_REGISTER_SHADER_FAST("volumetricLightingIntegration.comp", fastfroxel_integration);
} // namespace plr
