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
            case 14: r = decodeUnorm8((uint8_t)x); break; // UNORM8 decode: the IEEE quotient c / 255 (the device's three-instruction form must equal it)
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

// ---- sampler probe: the 8 global samplers of resources/shaders/global.inc:35-42 (nearest / linear x clamp / repeat / border white / border black)
// on any image. coords = n x 2 (2D image) or n x 3 (3D image) normalised coordinates, out = n x 4. filter: 0 nearest, 1 linear, 2 textureGather
// component 0 (2D only; order (i0,j1), (i1,j1), (i1,j0), (i0,j0), the offsets table of indirectLightUpscale.comp:42-47).
// The HIP side is plr_debug_sampler_eval with the same arguments.
extern "C" void orc_sampler_eval(const orc_image* image, int32_t filter, int32_t address, const float* coords, float* out, int64_t n) {
    const Image& im = img(image);
    const bool is3d = im.d > 1;
    for (int64_t i = 0; i < n; i++) {
        vec4 r;
        if (is3d) r = texture3D(im, filter == 1 ? LINEAR : NEAREST, address, vec3(coords[3 * i], coords[3 * i + 1], coords[3 * i + 2]));
        else if (filter == 2) r = textureGatherR(im, address, vec2(coords[2 * i], coords[2 * i + 1]));
        else r = texture2D(im, filter == 1 ? LINEAR : NEAREST, address, vec2(coords[2 * i], coords[2 * i + 1]));
        out[4 * i] = r.x; out[4 * i + 1] = r.y; out[4 * i + 2] = r.z; out[4 * i + 3] = r.w;
    }
}

// fn 0: importanceSampleCosine(xi, N) (sampling.inc:25-45), in: 5 floats -> 3 floats
// fn 1: directionToSH_L1(v) (SphericalHarmonics.inc), in: 3 floats -> 4 floats
extern "C" void orc_kat_sampling(int fn, const float* in, float* out, int64_t n) {
    for (int64_t i = 0; i < n; i++) {
        if (fn == 0) {
            const float* p = in + 5 * i;
            const vec3 L = importanceSampleCosine(vec2(p[0], p[1]), vec3(p[2], p[3], p[4]));
            out[3 * i] = L.x; out[3 * i + 1] = L.y; out[3 * i + 2] = L.z;
        } else if (fn == 1) {
            const vec4 s = directionToSH_L1(vec3(in[3 * i], in[3 * i + 1], in[3 * i + 2]));
            out[4 * i] = s.x; out[4 * i + 1] = s.y; out[4 * i + 2] = s.z; out[4 * i + 3] = s.w;
        }
    }
}

// sampleSkyLut (sky.inc:85-116) for n directions (3 floats each) -> n x 3 floats
extern "C" void orc_kat_sky_lut(const orc_image* lut, const float* dirs, float* out, int64_t n) {
    for (int64_t i = 0; i < n; i++) {
        const vec3 c = sampleSkyLut(vec3(dirs[3 * i], dirs[3 * i + 1], dirs[3 * i + 2]), img(lut));
        out[3 * i] = c.x; out[3 * i + 1] = c.y; out[3 * i + 2] = c.z;
    }
}

// The twelve PCF taps of calcShadow (triangle.frag:100-110) as functions of the pixel's noise texel alone, for all 256 values of an R8 UNORM
// texel: out[(k * 12 + i) * 2] = (cos(angle) d, sin(angle) d) with d = sqrt((i + noise / 2) / 12), angle = noise 2 pi + 2 pi i / 12 - the
// shader's expressions in the shader's order (the same statements as shading.cpp's calcShadow, before the multiplication by offsetScale).
extern "C" void orc_kat_pcf_taps(float* out) {
    const float sampleCount = 12.f;
    for (int k = 0; k < 256; k++) {
        const float noise = decodeUnorm8((uint8_t)k);
        for (int i = 0; (float)i < sampleCount; i++) {
            float d = ((float)i + 0.5f * noise) / sampleCount;
            d = std::sqrt(d);
            const float angle = noise * 2.f * pi + 2.f * pi * (float)i / sampleCount;
            float sa, ca;
            det_sincosf(angle, &sa, &ca);
            out[(k * 12 + i) * 2] = ca * d;
            out[(k * 12 + i) * 2 + 1] = sa * d;
        }
    }
}
