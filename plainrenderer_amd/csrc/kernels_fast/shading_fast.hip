// PLR_MATH_FAST variant of the deferred shade (exact variant and the description of the re-expression: kernels/shading.hip).
//
// Same lighting model term by term; the arithmetic is restructured for the VALU:
//  * (1-x)^5 Fresnel / CoD terms are x2*x2*x instead of exp2(5*log2(x)); the remaining pow/log/exp use v_log_f32 / v_exp_f32
//  * world position = camPos + (forward - tan*ndc.y*up + tan*aspect*ndc.x*right) * depthLinear (the normalisation of the view ray
//    cancels against the division by dot(ray, forward)); normalisations use v_rsq_f32, divisions v_rcp_f32, FMA contraction is on
//  * the 12 PCF tap directions are one hardware sin/cos of the per-pixel noise angle rotated by a constant 12-entry table
//  * directionToSH_L1 of a unit vector has a constant norm, so its normalize() is two constants
// Output is R11G11B10 (6/5-bit mantissas); stated tolerance in tests/test_fast_kernels.py. A PCF tap whose depth comparison sits
// within rounding of equality can flip (1/12 of the sun term for that pixel).
#include "../backend.h"
#include "../device/shading_common.h"
#include "../device/fastmath.h"

namespace plr {

namespace fastshade {

#ifndef PLR_SHADE_WAVES
#define PLR_SHADE_WAVES 4 // 128 VGPRs, no scratch (5 waves = 96 VGPRs spills after the load-pairing rewrite)
#endif

PLR_DI float rcpf(float x) { return __builtin_amdgcn_rcpf(x); }
PLR_DI float rsqf(float x) { return __builtin_amdgcn_rsqf(x); }
PLR_DI float sqrtv(float x) { return __builtin_amdgcn_sqrtf(x); } // v_sqrt_f32 (1 ulp) without the denormal-range rescue sequence of sqrtf()
// v_min / v_max / v_med3: a NaN operand loses, as in the software forms of detmath.h, in one instruction
PLR_DI float fmin1(float a, float b) { return __builtin_fminf(a, b); }
PLR_DI float fmax1(float a, float b) { return __builtin_fmaxf(a, b); }
PLR_DI float fclamp(float x, float lo, float hi) { return __builtin_fminf(__builtin_fmaxf(x, lo), hi); }
PLR_DI float fmix(float a, float b, float t) { return a + (b - a) * t; }
PLR_DI float log2h(float x) { return __builtin_amdgcn_logf(x); }   // v_log_f32 (base 2)
PLR_DI float exp2h(float x) { return __builtin_amdgcn_exp2f(x); }  // v_exp_f32 (base 2)
PLR_DI vec3 nrm(vec3 v) { return v * rsqf(dot(v, v)); }
PLR_DI float pow5(float x) { x = fmax1(x, 0.f); const float x2 = x * x; return x2 * x2 * x; }
PLR_DI float fpow(float x, float y) { return x <= 0.f ? 0.f : exp2h(y * log2h(x)); }

PLR_DI float D_GGX(float NoH, float r) {
    const float a = NoH * r;
    const float k = r * rcpf(1.0f - NoH * NoH + a * a);
    return k * k * (1.0f / PLR_GLSL_PI);
}
PLR_DI float Visibility(float NoV, float NoL, float r) {
    const float r_2 = r * r;
    const float v1 = NoL * sqrtv(NoV * NoV * (1.f - r_2) + r_2);
    const float v2 = NoV * sqrtv(NoL * NoL * (1.f - r_2) + r_2);
    return 0.5f * rcpf(v1 + v2);
}
PLR_DI vec3 F_Schlick(vec3 f0, vec3 f90, float VoH) { return f0 + (f90 - f0) * pow5(1.f - VoH); }
PLR_DI vec3 DisneyDiffuse(vec3 diffuseColor, float NoL, float VoH, float NoV, float r) {
    const float energyBias = 0.5f * r;
    const float energyFactor = fmix(1.f, 1.f / 1.51f, r);
    const float f90 = energyBias + 2.f * VoH * VoH * r;
    const float fl = 1.f + (f90 - 1.f) * pow5(1.f - NoL), fv = 1.f + (f90 - 1.f) * pow5(1.f - NoV);
    return diffuseColor * ((1.f / PLR_GLSL_PI) * fl * fv * energyFactor);
}
PLR_DI vec3 CoDWWIIDiffuse(vec3 diffuseColor, float NoL, float VoH, float NoV, float NoH, float r) {
    const float f0Diffuse = VoH + pow5(1.f - VoH);
    const float f1 = (1.f - 0.75f * pow5(1.f - NoL)) * (1.f - 0.75f * pow5(1.f - NoV));
    const float g = log2h(2.f * rcpf(r * r) - 1.f) * (1.f / 18.f);
    const float t = fclamp(2.2f * g - 0.5f, 0.f, 1.f);
    const float fd = f0Diffuse + (f1 - f0Diffuse) * t;
    const float fb = (34.5f * g * g - 59.f * g + 24.5f) * VoH * exp2h(-fmax1(73.2f * g - 21.2f, 8.9f) * sqrtv(NoH));
    return diffuseColor * ((1.f / PLR_GLSL_PI) * (fd + fb));
}
PLR_DI float Titanfall2DiffuseSingleComponent(float NoL, float LoV, float NoV, float NoH, float r) {
    const float facing = 0.5f + 0.5f * LoV;
    const float rough = facing * (0.9f - 0.4f * facing) * (0.5f + NoH) * rcpf(fmax1(NoH, 0.03f));
    const float smoothDiffuse = 1.05f * (1.f - pow5(1.f - NoL)) * (1.f - pow5(1.f - NoV));
    return (1.f / PLR_GLSL_PI) * fmix(smoothDiffuse, rough, r);
}
PLR_DI vec3 GGXSingleScattering(float r, vec3 f0, float NoH, float NoV, float VoH, float NoL) {
    return (D_GGX(NoH, r) * Visibility(NoV, NoL, r)) * F_Schlick(f0, vec3(1.f), VoH);
}
PLR_DI float ReflectedEnergyAverage(float roughness) {
    const float smoothness = 1.f - sqrtv(roughness);
    float r = -0.0761947f - 0.383026f * smoothness;
    r = 1.04997f + smoothness * r;
    r = 0.409255f + smoothness * r;
    return fmin1(0.999f, r);
}
// both branches evaluated and selected: a divergent branch around the power costs more than the power (two quarter-rate instructions)
PLR_DI float sRGBToLinear1(float c) {
    const float lin = c * (1.f / 12.92f), pw = fpow((c + 0.055f) * (1.f / 1.055f), 2.4f);
    return c <= 0.004045f ? lin : pw;
}

// bilinear RGBA16F fetch with clamp-to-edge; the two texels of a row come from one 16-byte load (a load instruction costs the
// texture addresser the same whatever its width)
PLR_DI vec4 bilinearLut(const ImgView& im, float u, float v) {
    int i0, j0; float a, b;
    linearCoord(u * (float)im.w, &i0, &a);
    linearCoord(v * (float)im.h, &j0, &b);
    const int x0 = clampi(i0, im.w), x1 = clampi(i0 + 1, im.w), y0 = clampi(j0, im.h), y1 = clampi(j0 + 1, im.h);
    const int xb = clampTo(x0, im.w - 2);
    auto rowPair = [&](int y, vec4* t0, vec4* t1) {
        const uint2* row = (const uint2*)im.ptr + __umul24((uint32_t)y, (uint32_t)im.w);
        uint4 q; // the launcher sends images narrower than two texels to the general kernel
        __builtin_memcpy(&q, row + xb, 16);
        const uint2 lo = make_uint2(q.x, q.y), hi = make_uint2(q.z, q.w);
        const uint2 e0 = x0 == xb ? lo : hi, e1 = x1 == xb ? lo : hi;
        *t0 = vec4(halfBitsToFloat(e0.x & 0xffffu), halfBitsToFloat(e0.x >> 16), halfBitsToFloat(e0.y & 0xffffu), halfBitsToFloat(e0.y >> 16));
        *t1 = vec4(halfBitsToFloat(e1.x & 0xffffu), halfBitsToFloat(e1.x >> 16), halfBitsToFloat(e1.y & 0xffffu), halfBitsToFloat(e1.y >> 16));
    };
    vec4 t00, t10, t01, t11;
    rowPair(y0, &t00, &t10);
    rowPair(y1, &t01, &t11);
    const vec4 top = t00 + (t10 - t00) * a, bot = t01 + (t11 - t01) * a;
    return top + (bot - top) * b;
}

struct ShadeParams {
    ImgView color, depth, normal, albedo, specular, brdfLut, shadowMaps[4], ysh, cocg, volumetricLut, skyLut;
    const LightBuffer* light;
    const ShadowCascadeInfo* shadowInfo;
    const VolumetricLightingSettings* vol;
    const GlobalUbo* g;
    const ImgView* bindless;
    uint32_t bindlessCount;
    uint32_t cascadeCount;
    int coverW, coverH, yBase;
    uint32_t* sig; // decision signatures (plr_debug_set_decision_signature) or null
};

PLR_DI vec3 gbufferNormal(const ImgView& normalTexture, int x, int y) {
    x = clampi(x, normalTexture.w);
    y = clampi(y, normalTexture.h);
    const vec3 raw = fastm::unorm8x4(((const uint32_t*)normalTexture.ptr)[fastm::texelIndex((uint32_t)x, (uint32_t)y, (uint32_t)normalTexture.w)]).xyz() * 2.f - 1.f;
    const vec3 N = nrm(raw);
    return anyNan(N) ? raw : N;
}

// 12-tap rotated-disc PCF (lightingFunctions / shadow sampling of the shader). Per tap the shader evaluates
//   d = sqrt((i + noise/2) / 12), angle = 2 pi (i / 12 + noise), uv = base + (cos, sin)(angle) * 0.03 * lightSpaceScale * d,
//   nearest fetch with a black border, shadow += actualDepth >= texel.
// Restructured for issue count (the kernel is bound by VALU issue): the twelve directions are the noise rotation turned by
// multiples of 30 degrees, so they come from four products (c0, s0 times cos 30 / sin 30) and four sums; the texel-space scale is
// folded into the per-tap radius; the border test is two unsigned compares on the floored texel coordinate; and the depth compare
// runs on the raw 16-bit texel against floor(actualDepth * 65535) (for an integer texel t: t / 65535 <= a <=> t <= floor(65535 a)).
PLR_DI float calcShadow(vec3 pos, const ImgView& shadowMap, const float* lightMatrix, vec2 lightSpaceScale, float noise) {
    vec4 p = mulMat4(lightMatrix, vec4(pos, 1.f));
    const float iw = rcpf(p.w);
    const float fw = (float)shadowMap.w, fh = (float)shadowMap.h;
    const float bxw = (p.x * iw * 0.5f + 0.5f) * fw, byh = (p.y * iw * 0.5f + 0.5f) * fh; // tap centre in texels
    const float actualDepth = fclamp(p.z * iw, 0.f, 1.f);
    const uint32_t depthThreshold = (uint32_t)(actualDepth * 65535.f); // truncation = floor, the value is non-negative
    const float sxw = 0.03f * lightSpaceScale.x * fw, syh = 0.03f * lightSpaceScale.y * fh;
    const float s0 = __builtin_amdgcn_sinf(noise), c0 = __builtin_amdgcn_cosf(noise); // v_sin/v_cos take revolutions: angle = noise * 2 pi
    // directions i * 30 degrees + noise rotation
    const float k = 0.8660254f;
    const float ck = c0 * k, sh = s0 * 0.5f, ch = c0 * 0.5f, sk = s0 * k;
    const uint16_t* sm = (const uint16_t*)shadowMap.ptr;
    const uint32_t w = (uint32_t)shadowMap.w, h = (uint32_t)shadowMap.h;
    const int wm1 = shadowMap.w - 1, hm1 = shadowMap.h - 1;
    uint32_t lit = 0u;
    auto tap = [&](int i, float cx, float cy) {
        const float d = sqrtv(((float)i + 0.5f * noise) * (1.f / 12.f));
        const float tu = bxw + cx * (sxw * d), tv = byh + cy * (syh * d);
        const int xi = floorToInt(tu), yi = floorToInt(tv);
        const bool inside = (uint32_t)xi < w && (uint32_t)yi < h; // black border outside: depth 0, always "lit"
        const uint32_t x = (uint32_t)clampTo(xi, wm1), y = (uint32_t)clampTo(yi, hm1);
        const uint32_t texel = sm[fastm::texelIndex(x, y, w)];
        lit += ((inside ? texel : 0u) <= depthThreshold) ? 1u : 0u;
    };
    // taps i and i + 6 point in opposite directions
#pragma unroll
    for (int i = 0; i < 6; i++) {
        const float cx = i == 0 ? c0 : i == 1 ? ck - sh : i == 2 ? ch - sk : i == 3 ? -s0 : i == 4 ? -ch - sk : -ck - sh;
        const float cy = i == 0 ? s0 : i == 1 ? sk + ch : i == 2 ? sh + ck : i == 3 ? c0 : i == 4 ? ck - sh : ch - sk;
        tap(i, cx, cy);
        tap(i + 6, -cx, -cy);
    }
    return (float)lit * (1.f / 12.f);
}

template <int MULTISCATTER>
PLR_DI vec3 specularMultiscatteringLobe(const ImgView& brdfLutTex, float r, float NoL, vec3 f0, vec3 singleScatteringLobe, vec3 brdfLut) {
    const float energyOutgoing = brdfLut.y;
    const vec3 fresnelAverage = f0 + (1.f - f0) * (1.f / 21.f);
    if (MULTISCATTER == 0) {
        const float energyAverage = ReflectedEnergyAverage(r);
        const float energyIncoming = bilinearLut(brdfLutTex, r, NoL).y;
        const float unscaled = (1.f - energyIncoming) * (1.f - energyOutgoing) * rcpf(3.1415f * (1.f - energyAverage));
        const vec3 den = 1.f - fresnelAverage * (1.f - energyAverage);
        const vec3 scaling = (fresnelAverage * fresnelAverage * energyAverage) * vec3(rcpf(den.x), rcpf(den.y), rcpf(den.z));
        return unscaled * scaling;
    } else if (MULTISCATTER == 1) {
        const float lobe = (1.f - energyOutgoing) * (1.f / PLR_GLSL_PI);
        const vec3 den = 1.f - fresnelAverage * (1.f - energyOutgoing);
        return (fresnelAverage * fresnelAverage * (energyOutgoing * lobe)) * vec3(rcpf(den.x), rcpf(den.y), rcpf(den.z));
    } else if (MULTISCATTER == 2) {
        return f0 * (rcpf(energyOutgoing) - 1.f) * singleScatteringLobe;
    }
    return vec3(0.f);
}

template <int DIFFUSE_BRDF, int MULTISCATTER, bool GEOMETRIC_AA, int INDIRECT_TECH>
__global__ __launch_bounds__(256, PLR_SHADE_WAVES) void deferredShadingFastKernel(ShadeParams P) {
    const int px = (int)(blockIdx.x * 64u + (threadIdx.x & 63u));
    const int py = P.yBase + (int)(blockIdx.y * 4u + (threadIdx.x >> 6));
    if (px >= P.coverW || py >= P.coverH) return;
    const GlobalUbo* g = P.g;
    const float fx = (float)px + 0.5f, fy = (float)py + 0.5f;
    const float su = fx * rcpf((float)g->screenResolution[0]), sv = fy * rcpf((float)g->screenResolution[1]);
    const uint32_t idx = fastm::texelIndex((uint32_t)px, (uint32_t)py, (uint32_t)P.color.w);
    const float depth = ((const float*)P.depth.ptr)[idx];
    const vec3 camFwd = ld3(g->cameraForward), camPos = ld3(g->cameraPosition);
    const float ndx = su * 2.f - 1.f, ndy = sv * 2.f - 1.f;
    const vec3 ray = camFwd + (-g->cameraTanFovHalf * ndy) * ld3(g->cameraUp) + (g->cameraTanFovHalf * g->cameraAspectRatio * ndx) * ld3(g->cameraRight);
    if (depth == 0.f) {
        ((uint32_t*)P.color.ptr)[idx] = packR11G11B10(fastm::sampleSkyLut(nrm(ray), P.skyLut));
        if (P.sig) P.sig[idx] = 128u;
        return;
    }
    const float depthLinear = g->nearPlane * g->farPlane * rcpf(g->farPlane + (1.f - depth) * (g->nearPlane - g->farPlane));
    const vec3 toPixel = ray * depthLinear; // passPos - camPos
    const vec3 passPos = camPos + toPixel;

    // the launcher guarantees that the G-buffer images have the colour target's size: one texel index serves all of them
    const vec3 albedoTexel = fastm::unorm8x4(((const uint32_t*)P.albedo.ptr)[idx]).xyz();
    const vec3 specularTexel = fastm::unorm8x4(((const uint32_t*)P.specular.ptr)[idx]).xyz();
    const float metalic = specularTexel.z;
    float r = specularTexel.y;
    r = fmax1(r * r, 0.0045f);
    const vec3 albedo(sRGBToLinear1(albedoTexel.x), sRGBToLinear1(albedoTexel.y), sRGBToLinear1(albedoTexel.z));
    const vec3 diffuseColor = (1.f - metalic) * albedo;
    const vec3 N = gbufferNormal(P.normal, px, py);
    const vec3 L = nrm(ld3(g->sunDirection));
    const float pixelDepth = depthLinear; // dot(camPos - passPos, -forward) = depthLinear * dot(ray, forward) = depthLinear
    const vec3 V = -nrm(ray);
    const vec3 H = nrm(V + L);
    if (GEOMETRIC_AA) {
        const int xl = px & ~1, yl = py & ~1;
        const vec3 Nn = (px & 1) ? gbufferNormal(P.normal, xl, py) : gbufferNormal(P.normal, xl + 1, py);
        const vec3 Nm = (py & 1) ? gbufferNormal(P.normal, px, yl) : gbufferNormal(P.normal, px, yl + 1);
        const vec3 N_U = Nn - N, N_V = Nm - N; // sign is irrelevant: only squared lengths are used
        const float variance = 0.25f * (dot(N_V, N_V) + dot(N_U, N_U));
        const float kernelRoughness2 = fmin1(2.f * variance, 0.18f);
        r = fclamp(sqrtv(r * r + kernelRoughness2), 0.f, 1.f);
    }
    const float NoH = fmax1(dot(N, H), 0.f);
    const float NdotL = dot(N, L);
    const float NoL = fclamp(NdotL, 0.f, 1.f);
    const float VoH = fabsf(dot(V, H));
    const float LoV = fmax1(dot(L, V), 0.f);
    const float NoV = fmax1(fabsf(dot(N, V)), 0.0001f);
    const vec3 f0 = vmix(vec3(0.04f), albedo, metalic);

    const uint32_t noiseSlot = (uint32_t)g->noiseTextureIndices[g->frameIndexMod4 & 3u];
    const ImgView noiseTex = P.bindless[min(noiseSlot, P.bindlessCount - 1u)];
    const vec2 noiseTexel = fastm::unorm8x2(((const uint16_t*)noiseTex.ptr)[fastm::texelIndex((uint32_t)fastm::repeatIndex(px, noiseTex.w), (uint32_t)fastm::repeatIndex(py, noiseTex.h), (uint32_t)noiseTex.w)]);

    int cascadeIndex = 0;
#pragma unroll
    for (int cascade = 0; cascade < 3; cascade++) cascadeIndex += (cascade < (int)P.cascadeCount - 1 && pixelDepth >= P.shadowInfo->splits[cascade]) ? 1 : 0;
    const vec2 lss(P.shadowInfo->lightSpaceScale[cascadeIndex][0], P.shadowInfo->lightSpaceScale[cascadeIndex][1]);
    // one instance of the PCF loop: the cascade only selects which image view and matrix it reads
    ImgView shadowMap = P.shadowMaps[0]; // field-wise selects (the cascade differs between the pixels of a wave)
#pragma unroll
    for (int i = 1; i < 4; i++) {
        shadowMap.ptr = cascadeIndex == i ? P.shadowMaps[i].ptr : shadowMap.ptr;
        shadowMap.w = cascadeIndex == i ? P.shadowMaps[i].w : shadowMap.w;
        shadowMap.h = cascadeIndex == i ? P.shadowMaps[i].h : shadowMap.h;
    }
    const float sunShadow = calcShadow(passPos, shadowMap, P.shadowInfo->lightMatrices[cascadeIndex], lss, noiseTexel.x);
    if (P.sig) P.sig[idx] = (uint32_t)cascadeIndex | ((uint32_t)(sunShadow * 12.f + 0.5f) << 2) | 64u; // cascade, lit PCF taps, geometry (oracle/oracle.h)
    const vec3 directLighting = (fmax1(NdotL, 0.f) * sunShadow) * ld3(P.light->sunColor);
    const vec3 brdfLut = bilinearLut(P.brdfLut, r, NoV).xyz();

    vec3 diffuseDirect;
    vec3 diffuseBRDFIntegral(brdfLut.z);
    if (DIFFUSE_BRDF == 0) diffuseDirect = diffuseColor * (1.f / PLR_GLSL_PI) * directLighting;
    else if (DIFFUSE_BRDF == 1) diffuseDirect = DisneyDiffuse(diffuseColor, NoL, VoH, NoV, r) * directLighting;
    else if (DIFFUSE_BRDF == 2) diffuseDirect = CoDWWIIDiffuse(diffuseColor, NoL, VoH, NoV, NoH, r) * directLighting;
    else {
        const float single = Titanfall2DiffuseSingleComponent(NoL, LoV, NoV, NoH, r);
        diffuseDirect = diffuseColor * (single + diffuseColor * (0.1159f * r)) * directLighting;
        float multiIntegral = 0.1159f * r * PLR_GLSL_PI * 2.f;
        multiIntegral *= (1.f - (0.04f + 0.96f * pow5(1.f - NoV)));
        multiIntegral *= 0.94291f;
        diffuseBRDFIntegral = vmin(vec3(brdfLut.z) + diffuseColor * multiIntegral, vec3(1.f));
    }
    diffuseDirect = diffuseDirect * ((1.f - F_Schlick(f0, vec3(1.f), NoV)) * (1.f - F_Schlick(f0, vec3(1.f), NoL)));

    const vec3 singleScatteringLobe = GGXSingleScattering(r, f0, NoH, NoV, VoH, NoL);
    const vec3 multiScatteringLobe = specularMultiscatteringLobe<MULTISCATTER>(P.brdfLut, r, NoL, f0, singleScatteringLobe, brdfLut);
    const vec3 specularDirect = directLighting * (singleScatteringLobe + multiScatteringLobe);

    vec3 lightingIndirect;
    if (INDIRECT_TECH == 0) {
        const int ix = clampTo(floorToInt(su * (float)P.ysh.w), P.ysh.w - 1), iy = clampTo(floorToInt(sv * (float)P.ysh.h), P.ysh.h - 1);
        const vec4 irradiance_Y_SH = Texel<F_RGBA16F>::load(P.ysh.ptr, fastm::texelIndex((uint32_t)ix, (uint32_t)iy, (uint32_t)P.ysh.w));
        const vec4 cc = Texel<F_RG16F>::load(P.cocg.ptr, fastm::texelIndex((uint32_t)ix, (uint32_t)iy, (uint32_t)P.cocg.w));
        // directionToSH_L1(N) for unit N: normalize((0.28209, -0.48860 N.y, 0.48860 N.z, -0.48860 N.x)) = (0.5, -0.86603 N.y, ...)
        const vec4 shN(0.5f, -0.8660254f * N.y, 0.8660254f * N.z, -0.8660254f * N.x);
        const float irradiance_Y = dot(irradiance_Y_SH, shN);
        const vec3 irradiance = YCoCgToLinear(vec3(irradiance_Y, cc.x, cc.y));
        const vec3 diffuseIndirect = irradiance * diffuseColor * diffuseBRDFIntegral;
        const vec3 dominantDirection = dominantDirectionFromSH_L1(irradiance_Y_SH);
        const float dominantDirectionLength = fclamp(sqrtv(dot(dominantDirection, dominantDirection)), 0.01f, 1.f);
        const float r_indirect = fmix(1.f, r, sqrtv(dominantDirectionLength));
        const vec3 L_indirect = dominantDirection * rcpf(dominantDirectionLength);
        const vec3 H_indirect = nrm(L_indirect + V);
        const float NoH_indirect = fmax1(dot(N, H_indirect), 0.f);
        const float NoL_indirect = fmax1(dot(N, L_indirect), 0.f);
        const float VoH_indirect = fmax1(dot(V, H_indirect), 0.f);
        const vec3 single_i = GGXSingleScattering(r_indirect, f0, NoH_indirect, NoV, VoH_indirect, NoL_indirect);
        const vec3 multi_i = specularMultiscatteringLobe<MULTISCATTER>(P.brdfLut, r_indirect, NoL_indirect, f0, single_i, brdfLut);
        const vec3 specularIndirect = (single_i + multi_i) * YCoCgToLinear(vec3(irradiance_Y_SH.x, cc.x, cc.y));
        lightingIndirect = diffuseIndirect + specularIndirect;
    } else {
        const float amb = 0.003f * P.light->sunStrengthExposed;
        const vec3 singleScattering = vmix(vec3(brdfLut.x), vec3(brdfLut.y), f0);
        lightingIndirect = (amb * diffuseColor) * diffuseBRDFIntegral + singleScattering * amb;
    }
    vec3 outColor = (diffuseDirect + specularDirect) * P.light->sunStrengthExposed + lightingIndirect;
    {
        // applyVolumetricLighting: froxel z = log(linear * (e^3 - 1) + 1) / 3
        const float nu = su + (noiseTexel.x - 0.5f) * 0.013f, nv = sv + (noiseTexel.y - 0.5f) * 0.013f;
        const float linear = pixelDepth * rcpf(P.vol->maxDistance);
        const float z = log2h(linear * 19.0855369f + 1.f) * (0.693147181f / 3.f);
        const ImgView& vol = P.volumetricLut;
        int i0, j0, k0; float a, b, c;
        linearCoord(nu * (float)vol.w, &i0, &a);
        linearCoord(nv * (float)vol.h, &j0, &b);
        linearCoord(z * (float)vol.d, &k0, &c);
        const int x0 = clampi(i0, vol.w), x1 = clampi(i0 + 1, vol.w);
        const uint32_t y0 = __umul24((uint32_t)clampi(j0, vol.h), (uint32_t)vol.w), y1 = __umul24((uint32_t)clampi(j0 + 1, vol.h), (uint32_t)vol.w);
        const uint32_t sl = (uint32_t)vol.w * (uint32_t)vol.h; // uniform
        const uint32_t z0 = __umul24((uint32_t)clampi(k0, vol.d), sl), z1 = __umul24((uint32_t)clampi(k0 + 1, vol.d), sl); // slices below 2^24 texels
        const int xb = clampTo(x0, vol.w - 2);
        auto rowLerp = [&](uint32_t rowBase) { // texels x0, x1 of one row, lerped by a; one 16-byte load when the row has two texels
            const uint2* row = (const uint2*)vol.ptr + rowBase;
            uint4 q;
            __builtin_memcpy(&q, row + xb, 16);
            const uint2 tl = make_uint2(q.x, q.y), th = make_uint2(q.z, q.w);
            const uint2 e0 = x0 == xb ? tl : th, e1 = x1 == xb ? tl : th;
            const vec4 t0(halfBitsToFloat(e0.x & 0xffffu), halfBitsToFloat(e0.x >> 16), halfBitsToFloat(e0.y & 0xffffu), halfBitsToFloat(e0.y >> 16));
            const vec4 t1(halfBitsToFloat(e1.x & 0xffffu), halfBitsToFloat(e1.x >> 16), halfBitsToFloat(e1.y & 0xffffu), halfBitsToFloat(e1.y >> 16));
            return t0 + (t1 - t0) * a;
        };
        const vec4 l0 = rowLerp(z0 + y0), l1 = rowLerp(z0 + y1), h0 = rowLerp(z1 + y0), h1 = rowLerp(z1 + y1);
        const vec4 lo = l0 + (l1 - l0) * b, hi = h0 + (h1 - h0) * b;
        const vec4 it = lo + (hi - lo) * c;
        outColor = outColor * it.w + it.xyz();
    }
    ((uint32_t*)P.color.ptr)[idx] = packR11G11B10(outColor);
}

typedef void (*ShadeKernel)(ShadeParams);
template <int D, int M, bool G> static ShadeKernel pickIndirect(int tech) {
    return tech == 0 ? (ShadeKernel)deferredShadingFastKernel<D, M, G, 0> : (ShadeKernel)deferredShadingFastKernel<D, M, G, 1>;
}
template <int D, int M> static ShadeKernel pickAA(bool aa, int tech) { return aa ? pickIndirect<D, M, true>(tech) : pickIndirect<D, M, false>(tech); }
template <int D> static ShadeKernel pickMulti(int m, bool aa, int tech) {
    switch (m) {
        case 0: return pickAA<D, 0>(aa, tech);
        case 1: return pickAA<D, 1>(aa, tech);
        case 2: return pickAA<D, 2>(aa, tech);
        default: return pickAA<D, 3>(aa, tech);
    }
}

static int launchDeferredShadingFast(const PassCtx& c) {
    if (int rc = c.needGlobal()) return rc;
    if (int rc = c.needStorage(0, F_R11G11B10, "deferredShading colour target")) return rc;
    if (int rc = c.needSampled(3, F_RGBA16F, "deferredShading brdfLutTexture")) return rc;
    if (int rc = c.needSbuf(7, sizeof(LightBuffer), "deferredShading lightBuffer")) return rc;
    if (int rc = c.needSbuf(8, sizeof(ShadowCascadeInfo), "deferredShading sunShadowInfo")) return rc;
    for (int i = 0; i < 4; i++) if (int rc = c.needSampled(9 + i, F_D16, "deferredShading shadowMapCascade")) return rc;
    if (int rc = c.needSampled(15, F_RGBA16F, "deferredShading indirectDiffuse_Y_SH")) return rc;
    if (int rc = c.needSampled(16, F_RG16F, "deferredShading indirectDiffuse_CoCg")) return rc;
    if (int rc = c.needSampled(18, F_RGBA16F, "deferredShading volumetricLightingLUT")) return rc;
    if (int rc = c.needUbuf(19, 52, "deferredShading volumetric settings")) return rc;
    if (int rc = c.needSampled(20, F_D32, "deferredShading depth")) return rc;
    if (int rc = c.needSampled(21, F_RGBA8, "deferredShading world normals")) return rc;
    if (int rc = c.needSampled(22, F_RGBA8, "deferredShading albedo")) return rc;
    if (int rc = c.needSampled(23, F_RGBA8, "deferredShading specular")) return rc;
    if (int rc = c.needSampled(24, F_R11G11B10, "deferredShading skyLut")) return rc;
    if (!c.bindless || c.bindlessCount == 0) return c.fail(-4, "deferredShading: global texture array (set 2) is empty");
    if (c.sampled[16].w != c.sampled[15].w || c.sampled[16].h != c.sampled[15].h) return c.fail(-4, "deferredShading: Y_SH and CoCg differ in size");
    // this kernel addresses depth / albedo / specular with the colour target's texel index: other layouts take the general kernel
    for (int b : {20, 22, 23}) if (c.sampled[b].w != c.storage[0].w || c.sampled[b].h != c.storage[0].h) return kUseGeneralKernel;
    // the LUT / froxel fetches read the two texels of a row with one 16-byte load
    if (c.sampled[3].w < 2 || c.sampled[18].w < 2) return kUseGeneralKernel;
    const int diffuseBRDF = c.specInt(0, 0), multi = c.specInt(1, 0), tech = c.specInt(3, 0);
    const bool aa = c.specBool(2, false);
    const uint32_t cascades = c.specUint(4, 4u);
    if (diffuseBRDF < 0 || diffuseBRDF > 3 || multi < 0 || multi > 3 || cascades < 1 || cascades > 4) return c.fail(-1, "deferredShading: specialisation constant out of range");
    ShadeKernel k = nullptr;
    switch (diffuseBRDF) {
        case 0: k = pickMulti<0>(multi, aa, tech); break;
        case 1: k = pickMulti<1>(multi, aa, tech); break;
        case 2: k = pickMulti<2>(multi, aa, tech); break;
        default: k = pickMulti<3>(multi, aa, tech); break;
    }
    ShadeParams P{};
    P.color = c.storage[0]; P.depth = c.sampled[20]; P.normal = c.sampled[21]; P.albedo = c.sampled[22]; P.specular = c.sampled[23];
    P.brdfLut = c.sampled[3];
    for (int i = 0; i < 4; i++) P.shadowMaps[i] = c.sampled[9 + i];
    P.ysh = c.sampled[15]; P.cocg = c.sampled[16]; P.volumetricLut = c.sampled[18]; P.skyLut = c.sampled[24];
    P.light = (const LightBuffer*)c.sbuf[7].ptr; P.shadowInfo = (const ShadowCascadeInfo*)c.sbuf[8].ptr;
    P.vol = (const VolumetricLightingSettings*)c.ubuf[19].ptr; P.g = c.global;
    P.bindless = c.bindless; P.bindlessCount = c.bindlessCount; P.cascadeCount = cascades;
    const PassCtx::RowSpan rs = c.rowSpan(P.color.h);
    P.coverW = std::min((int)(c.dispatch[0] * 8u), P.color.w); P.coverH = rs.y1; P.yBase = rs.y0; // columns [0, coverW), rows [yBase, coverH)
    if (P.coverW <= 0 || P.coverH <= P.yBase) return 0;
    P.sig = c.sigFor((size_t)P.color.w * (size_t)P.color.h);
    k<<<dim3(divUp((unsigned)P.coverW, 64u), divUp((unsigned)(P.coverH - P.yBase), 4u)), 256, 0, c.stream>>>(P);
    PLR_CHECK_LAUNCH(c);
    return 0;
}

// ------------------------------------------------------------------------------------------------ brdfLut.comp:20-101, PLR_MATH_FAST
// One lane per texel, the shader's 1024 Hammersley samples accumulated in the shader's order (the exact kernel: 4.7 ms, almost all of it
// software sin / cos / pow). Everything that depends on the sample index only is the same for every texel and is tabulated once per block in
// LDS: the GGX azimuth's sine / cosine, the second Hammersley coordinate, and the whole cosine-weighted direction of the diffuse sample (N is
// +z: the tangent frame of sampling.inc:4-45 is tangent = -y, bitangent = +x, so a sample h becomes (h.y, -h.x, h.z)). The loop body is then
// hardware sqrt / rcp / exp2 / log2 on per-texel terms; per-texel invariants (r, NoV) are hoisted by the compiler.
constexpr int kLutSamples = 1024;
template <int DIFFUSE_BRDF>
__global__ __launch_bounds__(256) void brdfLutFastKernel(ImgView lut, int coverW, int coverH) {
    __shared__ float4 ggx[kLutSamples];     // sin(phi), cos(phi), xi.y, -
    __shared__ float4 cosine[kLutSamples];  // L of the cosine-weighted sample
    for (int i = (int)threadIdx.x; i < kLutSamples; i += 256) {
        const float x0 = (float)i * (1.f / (float)kLutSamples);
        uint32_t bits = (uint32_t)i;
        bits = __builtin_bitreverse32(bits); // radicalInverse_VdC (sampling.inc: the five swaps are a 32-bit bit reversal)
        const float x1 = (float)bits * 2.3283064365386963e-10f;
        // the table is 1024 entries per block: the shader's own operations (software sin / cos, IEEE sqrt), so that every texel integrates over the
        // oracle's sample directions - the grazing, low-roughness texels sum a handful of spikes and follow every bit of the directions
        float sg, cg, sp, cp;
        det_sincosf(2.f * PLR_GLSL_PI * x0, &sg, &cg);
        det_sincosf(2.f * PLR_GLSL_PI * x1, &sp, &cp);
        ggx[i] = make_float4(sg, cg, x1, 0.f);
        const float cosT = sqrtf(x0), sinT = sqrtf(1.f - x0);
        cosine[i] = make_float4(sp * sinT, -(cp * sinT), cosT, 0.f);
    }
    __syncthreads();
    const int ux = (int)(blockIdx.x * 16u + (threadIdx.x & 15u)), uy = (int)(blockIdx.y * 16u + (threadIdx.x >> 4));
    if (ux >= coverW || uy >= coverH) return;
    const float r = fmax1((float)ux / (float)lut.w, 0.0001f);
    const float NoV = fmax1((float)uy, 0.1f) / (float)lut.h;
    const vec3 V(sqrtv(1.0f - NoV * NoV), 0.f, NoV);
    float r4m1;
    {
#pragma clang fp contract(off)
        const float r2 = r * r;
        r4m1 = r2 * r2 - 1.f;
    }
    const float fresnelOutV = 1.f - F_Schlick(vec3(0.04f), vec3(1.f), NoV).x;
    vec3 result(0.f);
    for (int i = 0; i < kLutSamples; i++) {
        const float4 gq = ggx[i], cq = cosine[i];
        {
            // the polar angle cancels catastrophically for low roughness (1 - cos^2 with cos within an ulp of 1): its operations are the
            // shader's, separately rounded, with a correctly rounded quotient and roots (Newton step on v_rcp / v_rsq), or the grazing rows
            // of the LUT end up tens of per cent away from the oracle's
            float cosT, sinT;
            {
#pragma clang fp contract(off)
                const float den = 1.f + r4m1 * gq.z, num = 1.f - gq.z;
                const float rd = rcpf(den), q0 = num * rd;
                const float q = __builtin_fmaf(__builtin_fmaf(-den, q0, num), rd, q0);
                auto root = [](float x) { const float r = rsqf(x), s0 = x * r; return x > 0.f ? __builtin_fmaf(__builtin_fmaf(-s0, s0, x), 0.5f * r, s0) : 0.f; };
                cosT = root(q);
                const float c2 = cosT * cosT;
                sinT = root(1.f - c2);
            }
            const vec3 H(gq.x * sinT, -(gq.y * sinT), cosT);
            const float VdotH = dot(V, H);
            const float Lz = 2.f * VdotH * H.z - V.z;
            const float VoH = fmax1(VdotH, 0.f), NoH = fmax1(H.z, 0.f), NoL = fmax1(Lz, 0.f);
            const float F_c = pow5(1.f - VoH);
            const float k = Visibility(NoV, NoL, r) * VoH * NoL * rcpf(NoH);
            const bool lit = NoL > 0.f;
            result.x += lit ? F_c * k : 0.f;
            result.y += lit ? k : 0.f;
        }
        {
            const vec3 L(cq.x, cq.y, cq.z);
            const vec3 H = nrm(V + L);
            const float VoH = fclamp(dot(V, H), 0.f, 1.f), NoL = fmax1(L.z, 0.f), NoH = fmax1(H.z, 0.f);
            const float fresnelInOut = fresnelOutV * (1.f - F_Schlick(vec3(0.04f), vec3(1.f), NoL).x);
            if (DIFFUSE_BRDF == 0) result.z += (1.f / PLR_GLSL_PI) * fresnelInOut;
            else if (DIFFUSE_BRDF == 1) result.z += DisneyDiffuse(vec3(1.f), NoL, VoH, NoV, r).x * fresnelInOut;
            else if (DIFFUSE_BRDF == 2) result.z += CoDWWIIDiffuse(vec3(1.f), NoL, VoH, NoV, NoH, r).x * fresnelInOut;
            else result.z += Titanfall2DiffuseSingleComponent(NoL, fclamp(dot(L, V), 0.f, 1.f), NoV, NoH, r) * fresnelInOut;
        }
    }
    result = result * (1.f / (float)kLutSamples);
    result.x *= 4.f;
    result.y *= 4.f;
    Texel<F_RGBA16F>::store(lut.ptr, (size_t)uy * (size_t)lut.w + ux, vec4(result, 0.f));
}

static int launchBrdfLutFast(const PassCtx& c) {
    if (!c.hasStorage(0) || c.storage[0].fmt != F_RGBA16F) return kUseGeneralKernel;
    const ImgView& lut = c.storage[0];
    const int w = std::min((int)(c.dispatch[0] * 8u), lut.w), h = std::min((int)(c.dispatch[1] * 8u), lut.h);
    if (w <= 0 || h <= 0) return 0;
    const dim3 grid(divUp((unsigned)w, 16u), divUp((unsigned)h, 16u));
    switch (c.specInt(0, 0)) {
        case 0: brdfLutFastKernel<0><<<grid, 256, 0, c.stream>>>(lut, w, h); break;
        case 1: brdfLutFastKernel<1><<<grid, 256, 0, c.stream>>>(lut, w, h); break;
        case 2: brdfLutFastKernel<2><<<grid, 256, 0, c.stream>>>(lut, w, h); break;
        case 3: brdfLutFastKernel<3><<<grid, 256, 0, c.stream>>>(lut, w, h); break;
        default: return kUseGeneralKernel;
    }
    PLR_CHECK_LAUNCH(c);
    return 0;
}

} // namespace fastshade

static int fastshade_launch(const PassCtx& c) { return fastshade::launchDeferredShadingFast(c); }
PLR_REGISTER_SHADER_FAST("deferredShading.comp", fastshade_launch);
static int fastshade_brdf_lut(const PassCtx& c) { return fastshade::launchBrdfLutFast(c); }
PLR_REGISTER_SHADER_FAST("brdfLut.comp", fastshade_brdf_lut);
} // namespace plr
