// The twelve PCF taps of calcShadow (triangle.frag:104-110) depend on the pixel only through its blue-noise texel, an 8-bit value:
//   d = sqrt((i + noise / 2) / 12), angle = noise 2 pi + 2 pi i / 12, offset = (cos, sin)(angle) * (offsetScale * d).
// 256 noise values x 12 taps are tabulated once, with the exact set's arithmetic (software sin / cos of detmath.h, IEEE square root and division:
// the oracle's bits), as unit-disc offsets (cos(angle) d, sin(angle) d); the fast shade reads its pixel's row (96 bytes) instead of evaluating
// v_sin / v_cos, twelve v_sqrt and the rotation per pixel, and a tap direction no longer carries the hardware sine's 1e-5 error.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace plr {

constexpr int kPcfTaps = 12, kPcfNoiseValues = 256;
constexpr size_t kPcfTapTableBytes = (size_t)kPcfNoiseValues * kPcfTaps * 2 * sizeof(float); // [noise byte][tap] {x, y}: 24 KB

// fills table[noise * 12 + tap] on `stream` (kernels/shading.hip: compiled with the exact set's flags); hipSuccess or the launch error
hipError_t buildPcfTapTable(float2* table, hipStream_t stream);

} // namespace plr
