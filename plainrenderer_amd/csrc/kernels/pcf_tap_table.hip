// The PCF tap table of the fast deferred shade (kernels_fast/pcf_taps.h), tabulated with the EXACT set's arithmetic: this file is compiled with the exact
// set's flags (no contraction, IEEE divide / square root, the software sine / cosine of detmath.h) and is part of libplr.so, next to the fast shade that reads the table.
#include "../backend.h"
#include "../device/shading_common.h"
#include "../kernels_fast/pcf_taps.h"

namespace plr {

// the statements of calcShadow's tap loop (triangle.frag:104-110; kernels_exact/shading.hip) that depend on (noise, i) only, for the 256 values a UNORM8 noise
// texel decodes to
__global__ void pcfTapTableKernel(float2* __restrict__ table) {
    const int e = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    if (e >= kPcfNoiseValues * kPcfTaps) return;
    const int i = e % kPcfTaps;
    const float noise = decodeUnorm8((uint32_t)(e / kPcfTaps));
    const float sampleCount = 12.f;
    float d = ((float)i + 0.5f * noise) / sampleCount;
    d = sqrtf(d);
    const float angle = noise * 2.f * PLR_GLSL_PI + 2.f * PLR_GLSL_PI * (float)i / sampleCount;
    float sa, ca;
    det_sincosf(angle, &sa, &ca);
    table[e] = make_float2(ca * d, sa * d);
}
hipError_t buildPcfTapTable(float2* table, hipStream_t stream) {
    pcfTapTableKernel<<<(kPcfNoiseValues * kPcfTaps + 255) / 256, 256, 0, stream>>>(table);
    return hipGetLastError();
}

} // namespace plr
