// Device probes of the detmath contract and the pixel codecs (plr_debug_math_eval / plr_debug_codec_eval).
// Function ids match oracle/probes.cpp so tests can demand bit identity between the two implementations.
#include "../backend.h"
#include "../device/shading_common.h"
#include "../../../include/plr.h"

namespace plr {

__global__ void mathEvalKernel(int fn, const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ out, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float x = a[i];
    const float y = b ? b[i] : 0.f;
    float r = 0.f;
    switch (fn) {
        case 0: r = det_logf(x); break;
        case 1: r = det_log2f(x); break;
        case 2: r = det_expf(x); break;
        case 3: r = det_exp2f(x); break;
        case 4: r = det_powf(x, y); break;
        case 5: r = det_sinf(x); break;
        case 6: r = det_cosf(x); break;
        case 7: r = det_acosf(x); break;
        case 8: r = det_atan2f(x, y); break;
        case 9: r = sqrtf(x); break;
        case 10: r = x / y; break;
        case 11: { const vec3 v = normalize(vec3(x, y, 1.f)); r = v.x; break; }
        case 12: r = gmin(x, y); break;
        case 13: r = gmax(x, y); break;
        default: break;
    }
    out[i] = r;
}

__global__ void codecEvalKernel(int fn, const void* __restrict__ in, void* __restrict__ out, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    switch (fn) {
        case 0: { const float* p = (const float*)in + 3 * i; ((uint32_t*)out)[i] = packR11G11B10(vec3(p[0], p[1], p[2])); break; }
        case 1: { const vec3 v = unpackR11G11B10(((const uint32_t*)in)[i]); float* o = (float*)out + 3 * i; o[0] = v.x; o[1] = v.y; o[2] = v.z; break; }
        case 2: ((uint16_t*)out)[i] = (uint16_t)floatToHalfBits(((const float*)in)[i]); break;
        case 3: ((float*)out)[i] = halfBitsToFloat(((const uint16_t*)in)[i]); break;
        case 4: ((uint8_t*)out)[i] = (uint8_t)encodeUnorm8(((const float*)in)[i]); break;
        case 6: ((float*)out)[i] = decodeSnorm16((int32_t)((const int16_t*)in)[i]); break;
        default: break;
    }
}

} // namespace plr

using namespace plr;

static thread_local std::string g_probeErr;
#define PROBE_TRY(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) return PLR_ERR_HIP; } while (0)

extern "C" int plr_debug_math_eval(int fn, const float* a, const float* b, float* out, int64_t n) {
    if (n <= 0) return PLR_OK;
    float *da = nullptr, *db = nullptr, *dout = nullptr;
    PROBE_TRY(hipMalloc((void**)&da, n * 4));
    PROBE_TRY(hipMalloc((void**)&dout, n * 4));
    PROBE_TRY(hipMemcpy(da, a, n * 4, hipMemcpyHostToDevice));
    if (b) { PROBE_TRY(hipMalloc((void**)&db, n * 4)); PROBE_TRY(hipMemcpy(db, b, n * 4, hipMemcpyHostToDevice)); }
    mathEvalKernel<<<(unsigned)((n + 255) / 256), 256>>>(fn, da, db, dout, n);
    PROBE_TRY(hipGetLastError());
    PROBE_TRY(hipMemcpy(out, dout, n * 4, hipMemcpyDeviceToHost));
    hipFree(da); hipFree(dout); if (db) hipFree(db);
    return PLR_OK;
}

extern "C" int plr_debug_codec_eval(int fn, const void* in, void* out, int64_t n) {
    if (n <= 0) return PLR_OK;
    size_t inBytes = 0, outBytes = 0;
    switch (fn) {
        case 0: inBytes = 12; outBytes = 4; break;
        case 1: inBytes = 4; outBytes = 12; break;
        case 2: inBytes = 4; outBytes = 2; break;
        case 3: inBytes = 2; outBytes = 4; break;
        case 4: inBytes = 4; outBytes = 1; break;
        case 6: inBytes = 2; outBytes = 4; break;
        default: return PLR_ERR_INVALID_ARGUMENT;
    }
    void *din = nullptr, *dout = nullptr;
    PROBE_TRY(hipMalloc(&din, n * inBytes));
    PROBE_TRY(hipMalloc(&dout, n * outBytes));
    PROBE_TRY(hipMemcpy(din, in, n * inBytes, hipMemcpyHostToDevice));
    codecEvalKernel<<<(unsigned)((n + 255) / 256), 256>>>(fn, din, dout, n);
    PROBE_TRY(hipGetLastError());
    PROBE_TRY(hipMemcpy(out, dout, n * outBytes, hipMemcpyDeviceToHost));
    hipFree(din); hipFree(dout);
    return PLR_OK;
}
