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
        case 14: r = decodeUnorm8Newton((uint32_t)x); break; // the oracle's fn 14 is the IEEE quotient x / 255
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

// ---- sampler probe (plr_debug_sampler_eval): the sampler definitions of device/image.h that the pass kernels are built from, on any image
template <int FMT, int ADDR>
__device__ vec4 sampleProbe(const ImgView& im, int filter, const float* c) {
    if (im.d > 1) {
        const vec3 uvw(c[0], c[1], c[2]);
        if (filter == 1) {
            if (ADDR == CLAMP || ADDR == REPEAT) return sampleLinear3D<FMT, ADDR>(im, uvw);
            return vec4(0.f); // no 3D image of the hot path is sampled with a border
        }
        return sampleNearest3D<FMT, ADDR>(im, uvw);
    }
    const vec2 uv(c[0], c[1]);
    if (filter == 2) return gatherR2D<FMT, ADDR>(im, uv);
    if (filter == 1) return sampleLinear2D<FMT, ADDR>(im, uv);
    return sampleNearest2D<FMT, ADDR>(im, uv);
}
template <int FMT>
__device__ vec4 sampleProbeAddr(const ImgView& im, int filter, int addr, const float* c) {
    switch (addr) {
        case CLAMP: return sampleProbe<FMT, CLAMP>(im, filter, c);
        case REPEAT: return sampleProbe<FMT, REPEAT>(im, filter, c);
        case BORDER_WHITE: return sampleProbe<FMT, BORDER_WHITE>(im, filter, c);
        default: return sampleProbe<FMT, BORDER_BLACK>(im, filter, c);
    }
}
__global__ void samplerEvalKernel(ImgView im, int filter, int addr, const float* __restrict__ coords, float* __restrict__ out, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float* c = coords + (im.d > 1 ? 3 : 2) * i;
    vec4 r(0.f);
    switch (im.fmt) {
        case F_RGBA16F: r = sampleProbeAddr<F_RGBA16F>(im, filter, addr, c); break;
        case F_RG16F: r = sampleProbeAddr<F_RG16F>(im, filter, addr, c); break;
        case F_R16F: r = sampleProbeAddr<F_R16F>(im, filter, addr, c); break;
        case F_R11G11B10: r = sampleProbeAddr<F_R11G11B10>(im, filter, addr, c); break;
        case F_D32: r = sampleProbeAddr<F_D32>(im, filter, addr, c); break;
        case F_D16: r = sampleProbeAddr<F_D16>(im, filter, addr, c); break;
        case F_RG16SN: r = sampleProbeAddr<F_RG16SN>(im, filter, addr, c); break;
        case F_RGBA8: r = sampleProbeAddr<F_RGBA8>(im, filter, addr, c); break;
        case F_RG8: r = sampleProbeAddr<F_RG8>(im, filter, addr, c); break;
        case F_RG32F: r = sampleProbeAddr<F_RG32F>(im, filter, addr, c); break;
        default: break;
    }
    out[4 * i] = r.x; out[4 * i + 1] = r.y; out[4 * i + 2] = r.z; out[4 * i + 3] = r.w;
}

// called by plr_debug_sampler_eval (backend.cpp resolves the image handle); coords / out are host memory
int launchSamplerProbe(const ImgView& view, int filter, int address, const float* coords, float* out, int64_t n) {
    switch (view.fmt) {
        case F_RGBA16F: case F_RG16F: case F_R16F: case F_R11G11B10: case F_D32: case F_D16: case F_RG16SN: case F_RGBA8: case F_RG8: case F_RG32F: break;
        default: return setLastError(PLR_ERR_UNSUPPORTED, "plr_debug_sampler_eval: image format has no sampler probe");
    }
    if (filter < 0 || filter > 2 || address < 0 || address > 3) return setLastError(PLR_ERR_INVALID_ARGUMENT, "plr_debug_sampler_eval: filter must be 0..2, address 0..3");
    const int dims = view.d > 1 ? 3 : 2;
    if (dims == 3 && (filter == 2 || (filter == 1 && address >= 2))) return setLastError(PLR_ERR_UNSUPPORTED, "plr_debug_sampler_eval: 3D images: nearest, or linear with clamp / repeat");
    float *dc = nullptr, *dout = nullptr;
    if (hipMalloc((void**)&dc, n * dims * 4) != hipSuccess || hipMalloc((void**)&dout, n * 16) != hipSuccess) return setLastError(PLR_ERR_HIP, "plr_debug_sampler_eval: hipMalloc failed");
    hipMemcpy(dc, coords, n * dims * 4, hipMemcpyHostToDevice);
    samplerEvalKernel<<<(unsigned)((n + 255) / 256), 256>>>(view, filter, address, dc, dout, n);
    const hipError_t e = hipGetLastError();
    hipMemcpy(out, dout, n * 16, hipMemcpyDeviceToHost);
    hipFree(dc); hipFree(dout);
    return e == hipSuccess ? PLR_OK : setLastError(PLR_ERR_HIP, std::string("plr_debug_sampler_eval: ") + hipGetErrorString(e));
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
