// ORACLE (test infrastructure, never shipped, never on the product path).
// Array probes of the detmath contract and the pixel-format codecs, used by tests to check the
// oracle against float64 libm / numpy and against the HIP side (plr_debug_math_eval / plr_debug_codec_eval
// use the same function ids).
#include "common.h"

using namespace orc;

extern "C" void orc_math_eval(int fn, const float* a, const float* b, float* out, int64_t n) {
    for (int64_t i = 0; i < n; i++) {
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
            case 9: r = std::sqrt(x); break;
            case 10: r = x / y; break;
            case 11: { const vec3 v = normalize(vec3(x, y, 1.f)); r = v.x; break; }
            case 12: r = gmin(x, y); break;
            case 13: r = gmax(x, y); break;
            default: break;
        }
        out[i] = r;
    }
}

// fn 0: float3 -> R11G11B10, 1: R11G11B10 -> float3, 2: float -> half, 3: half -> float,
//    4: float -> unorm8, 5: float -> snorm16, 6: snorm16 -> float
extern "C" void orc_codec_eval(int fn, const void* in, void* out, int64_t n) {
    for (int64_t i = 0; i < n; i++) {
        switch (fn) {
            case 0: { const float* p = (const float*)in + 3 * i; ((uint32_t*)out)[i] = packR11G11B10(vec3(p[0], p[1], p[2])); break; }
            case 1: { const vec3 v = unpackR11G11B10(((const uint32_t*)in)[i]); float* o = (float*)out + 3 * i; o[0] = v.x; o[1] = v.y; o[2] = v.z; break; }
            case 2: ((uint16_t*)out)[i] = floatToHalf(((const float*)in)[i]); break;
            case 3: ((float*)out)[i] = halfToFloat(((const uint16_t*)in)[i]); break;
            case 4: ((uint8_t*)out)[i] = encodeUnorm8(((const float*)in)[i]); break;
            case 5: ((int16_t*)out)[i] = encodeSnorm16(((const float*)in)[i]); break;
            case 6: ((float*)out)[i] = decodeSnorm16(((const int16_t*)in)[i]); break;
            default: break;
        }
    }
}
