// Shading for gfx950: brdfLut.comp and "deferredShading.comp", the deferred re-expression of the lighting in
// triangle.frag:84-341 (+ brdf.inc, GeometricAA.inc, sunShadowCascades.inc, volumetricFroxelLighting.inc); host side
// RenderFrontend.cpp:894-929,1031-1042,1093-1145.
//
// The reference shades forward in a raster pass (depth-equal test). Here the same per-pixel math runs as one compute pass
// over a G-buffer: gl_FragCoord.xy = iUV + 0.5, the world position is rebuilt from depth as sdfDiffuseTrace.comp:120-126
// does, albedo/specular come from two RGBA8 images, N = normalize(G-buffer normal), dFdxFine/dFdyFine(N) are the differences
// inside the 2x2 quad, and depth == 0 (sky) pixels receive sampleSkyLut(view direction) in place of the sky pass.
// Bindings keep triangle.frag's numbers (3, 7, 8, 9-12, 15, 16, 18, 19); the G-buffer inputs use 20-24, the colour target
// is storage image 0. One lane per pixel, 64-pixel row segments per wave; all LUT / shadow / froxel reads are gathers
// through L1/L2, the G-buffer and output are coalesced streams (28 B in, 4 B out per pixel).
#include "../backend.h"
#include "../device/shading_common.h"
#include "../kernels_fast/pcf_taps.h"

namespace plr {

// ---- brdf.inc
PLR_DI float D_GGX(float NoH, float r) {
    const float a = NoH * r;
    const float k = r / (1.0f - NoH * NoH + a * a);
    return k * k * (1.0f / PLR_GLSL_PI);
}
PLR_DI float Visibility(float NoV, float NoL, float r) {
    const float r_2 = r * r;
    const float v1 = NoL * sqrtf(NoV * NoV * (1.f - r_2) + r_2);
    const float v2 = NoV * sqrtf(NoL * NoL * (1.f - r_2) + r_2);
    return 0.5f / (v1 + v2);
}
PLR_DI vec3 F_Schlick(vec3 f0, vec3 f90, float VoH) { return f0 + (f90 - f0) * det_powf(1.f - VoH, 5.f); }
PLR_DI vec3 DisneyDiffuse(vec3 diffuseColor, float NoL, float VoH, float NoV, float r) {
    const float energyBias = gmix(0.f, 0.5f, r);
    const float energyFactor = gmix(1.f, 1.f / 1.51f, r);
    const float f90 = energyBias + 2.f * VoH * VoH * r;
    return diffuseColor / PLR_GLSL_PI * F_Schlick(vec3(1.f), vec3(f90), NoL) * F_Schlick(vec3(1.f), vec3(f90), NoV) * energyFactor;
}
PLR_DI vec3 CoDWWIIDiffuse(vec3 diffuseColor, float NoL, float VoH, float NoV, float NoH, float r) {
    const float f0Diffuse = VoH + det_powf(1.f - VoH, 5.f);
    const float f1 = (1.f - 0.75f * det_powf(1.f - NoL, 5.f)) * (1.f - 0.75f * det_powf(1.f - NoV, 5.f));
    const float g = det_log2f(2.f / (r * r) - 1.f) / 18.f;
    const float t = gclamp(2.2f * g - 0.5f, 0.f, 1.f);
    const float fd = f0Diffuse + (f1 - f0Diffuse) * t;
    const float fb = (34.5f * g * g - 59.f * g + 24.5f) * VoH * det_powf(2.f, -gmax(73.2f * g - 21.2f, 8.9f) * sqrtf(NoH));
    return diffuseColor / PLR_GLSL_PI * (fd + fb);
}
PLR_DI float Titanfall2DiffuseSingleComponent(float NoL, float LoV, float NoV, float NoH, float r) {
    const float facing = 0.5f + 0.5f * LoV;
    const float rough = facing * (0.9f - 0.4f * facing) * (0.5f + NoH) / gmax(NoH, 0.03f);
    const float smoothDiffuse = 1.05f * (1.f - det_powf(1.f - NoL, 5.f)) * (1.f - det_powf(1.f - NoV, 5.f));
    return 1.f / PLR_GLSL_PI * gmix(smoothDiffuse, rough, r);
}
PLR_DI vec3 Titanfall2Diffuse(vec3 diffuseColor, float NoL, float LoV, float NoV, float NoH, float r) {
    const float single = Titanfall2DiffuseSingleComponent(NoL, LoV, NoV, NoH, r);
    const float multi = 0.1159f * r;
    return diffuseColor * (single + diffuseColor * multi);
}
PLR_DI vec3 GGXSingleScattering(float r, vec3 f0, float NoH, float NoV, float VoH, float NoL) {
    const float D = D_GGX(NoH, r);
    const float Vis = Visibility(NoV, NoL, r);
    const vec3 F = F_Schlick(f0, vec3(1.f), VoH);
    return D * Vis * F;
}
PLR_DI float radicalInverse_VdC(uint32_t bits) {
    bits = (bits << 16u) | (bits >> 16u);
    bits = ((bits & 0x55555555u) << 1u) | ((bits & 0xAAAAAAAAu) >> 1u);
    bits = ((bits & 0x33333333u) << 2u) | ((bits & 0xCCCCCCCCu) >> 2u);
    bits = ((bits & 0x0F0F0F0Fu) << 4u) | ((bits & 0xF0F0F0F0u) >> 4u);
    bits = ((bits & 0x00FF00FFu) << 8u) | ((bits & 0xFF00FF00u) >> 8u);
    return (float)bits * 2.3283064365386963e-10f;
}

// ------------------------------------------------------------------------------------------------
// brdfLut.comp:20-101: one lane per texel, 1024 Hammersley samples each (run once, not per frame)
template <int DIFFUSE_BRDF>
__global__ __launch_bounds__(64) void brdfLutKernel(ImgView lut, int coverW, int coverH) {
    const int ux = (int)(blockIdx.x * 8u + (threadIdx.x & 7u)), uy = (int)(blockIdx.y * 8u + (threadIdx.x >> 3));
    if (ux >= coverW || uy >= coverH) return;
    float r = (float)ux / (float)lut.w;
    r = gmax(r, 0.0001f);
    const float NoV = gmax((float)uy, 0.1f) / (float)lut.h;
    const vec3 V(sqrtf(1.0f - NoV * NoV), 0.f, NoV);
    const vec3 N(0.f, 0.f, 1.f);
    const int samples = 1024;
    vec3 result(0.f);
    for (int i = 0; i < samples; i++) {
        const vec2 xi((float)i / (float)samples, radicalInverse_VdC((uint32_t)i));
        {
            const vec3 H = importanceSampleGGX(xi, r, N);
            const vec3 L = 2.f * dot(V, H) * H - V;
            const float VoH = gmax(dot(V, H), 0.f);
            const float NoH = gmax(H.z, 0.f);
            const float NoL = gmax(L.z, 0.f);
            if (NoL > 0.f) {
                const float F_c = det_powf(1.f - VoH, 5.f);
                const float Vis = Visibility(NoV, NoL, r);
                const float k = Vis * VoH * NoL / NoH;
                result.x += F_c * k;
                result.y += k;
            }
        }
        {
            const vec3 L = importanceSampleCosine(xi, N);
            const vec3 H = normalize(V + L);
            const float VoH = gclamp(dot(V, H), 0.f, 1.f);
            const float NoL = gmax(L.z, 0.f);
            const float NoH = gmax(H.z, 0.f);
            const vec3 F0Diffuse(0.04f);
            const float fresnelInOut = (1.f - F_Schlick(F0Diffuse, vec3(1.f), NoV).x) * (1.f - F_Schlick(F0Diffuse, vec3(1.f), NoL).x);
            if (DIFFUSE_BRDF == 0) result.z += (1.f / PLR_GLSL_PI) * fresnelInOut;
            else if (DIFFUSE_BRDF == 1) result.z += DisneyDiffuse(vec3(1.f), NoL, VoH, NoV, r).x * fresnelInOut;
            else if (DIFFUSE_BRDF == 2) result.z += CoDWWIIDiffuse(vec3(1.f), NoL, VoH, NoV, NoH, r).x * fresnelInOut;
            else {
                const float LoV = gclamp(dot(L, V), 0.f, 1.f);
                result.z += Titanfall2DiffuseSingleComponent(NoL, LoV, NoV, NoH, r) * fresnelInOut;
            }
        }
    }
    result /= (float)samples;
    result.x *= 4.f;
    result.y *= 4.f;
    Texel<F_RGBA16F>::store(lut.ptr, (size_t)uy * (size_t)lut.w + ux, vec4(result, 0.f));
}

static int launchBrdfLut(const PassCtx& c) {
    if (int rc = c.needStorage(0, F_RGBA16F, "brdfLut lut")) return rc;
    const ImgView& lut = c.storage[0];
    const int w = std::min((int)(c.dispatch[0] * 8u), lut.w), h = std::min((int)(c.dispatch[1] * 8u), lut.h);
    if (w <= 0 || h <= 0) return 0;
    const dim3 grid(divUp((unsigned)w, 8u), divUp((unsigned)h, 8u));
    switch (c.specInt(0, 0)) {
        case 0: brdfLutKernel<0><<<grid, 64, 0, c.stream>>>(lut, w, h); break;
        case 1: brdfLutKernel<1><<<grid, 64, 0, c.stream>>>(lut, w, h); break;
        case 2: brdfLutKernel<2><<<grid, 64, 0, c.stream>>>(lut, w, h); break;
        case 3: brdfLutKernel<3><<<grid, 64, 0, c.stream>>>(lut, w, h); break;
        default: return c.fail(-1, "brdfLut: diffuseBRDF must be 0..3");
    }
    PLR_CHECK_LAUNCH(c);
    return 0;
}
PLR_REGISTER_SHADER("brdfLut.comp", launchBrdfLut);

// ------------------------------------------------------------------------------------------------ deferred shading
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
    int xBase; // columns [xBase, coverW) (tile rendering: PassCtx::colSpan), rows [yBase, coverH)
};

PLR_DI vec3 gbufferNormal(const ImgView& normalTexture, int x, int y) {
    x = clampi(x, normalTexture.w);
    y = clampi(y, normalTexture.h);
    const vec3 raw = Texel<F_RGBA8>::load(normalTexture.ptr, (size_t)y * (size_t)normalTexture.w + x).xyz() * 2.f - 1.f;
    vec3 N = normalize(raw);
    if (anyNan(N)) N = raw;
    return N;
}

// triangle.frag:92-120
PLR_DI float calcShadow(vec3 pos, const ImgView& shadowMap, const float* lightMatrix, vec2 lightSpaceScale, float noise) {
    vec4 p = mulMat4(lightMatrix, vec4(pos, 1.f));
    p = p / p.w;
    const vec2 xy(p.x * 0.5f + 0.5f, p.y * 0.5f + 0.5f);
    const float actualDepth = gclamp(p.z, 0.f, 1.f);
    const vec2 offsetScale = 0.03f * lightSpaceScale;
    float shadow = 0.f;
    const float sampleCount = 12.f;
    for (int i = 0; i < 12; i++) {
        float d = ((float)i + 0.5f * noise) / sampleCount;
        d = sqrtf(d);
        const float angle = noise * 2.f * PLR_GLSL_PI + 2.f * PLR_GLSL_PI * (float)i / sampleCount;
        float sa, ca;
        det_sincosf(angle, &sa, &ca);
        const vec2 offset = vec2(ca, sa) * (offsetScale * d);
        const vec2 samplePosition = xy + offset;
        const float depthTexel = sampleNearest2D<F_D16, BORDER_BLACK>(shadowMap, samplePosition).x;
        shadow += (actualDepth >= depthTexel) ? 1.f : 0.f;
    }
    return shadow / sampleCount;
}

PLR_DI float ReflectedEnergyAverage(float roughness) {
    const float smoothness = 1.f - sqrtf(roughness);
    float r = -0.0761947f - 0.383026f * smoothness;
    r = 1.04997f + smoothness * r;
    r = 0.409255f + smoothness * r;
    return gmin(0.999f, r);
}

template <int MULTISCATTER>
PLR_DI vec3 specularMultiscatteringLobe(const ImgView& brdfLutTex, float r, float NoL, vec3 f0, vec3 singleScatteringLobe, vec3 brdfLut) {
    const float energyOutgoing = brdfLut.y;
    const vec3 fresnelAverage = f0 + (1.f - f0) / 21.f;
    if (MULTISCATTER == 0) {
        const float energyAverage = ReflectedEnergyAverage(r);
        const float energyIncoming = sampleLinear2D<F_RGBA16F, CLAMP>(brdfLutTex, vec2(r, NoL)).y;
        const float unscaled = (1.f - energyIncoming) * (1.f - energyOutgoing) / (3.1415f * (1.f - energyAverage));
        const vec3 scaling = (fresnelAverage * fresnelAverage * energyAverage) / (1.f - fresnelAverage * (1.f - energyAverage));
        return unscaled * scaling;
    } else if (MULTISCATTER == 1) {
        const vec3 lobe((1.f - energyOutgoing) / PLR_GLSL_PI);
        const vec3 scaling = (fresnelAverage * fresnelAverage * energyOutgoing) / (1.f - fresnelAverage * (1.f - energyOutgoing));
        return lobe * scaling;
    } else if (MULTISCATTER == 2) {
        return f0 * (1.f / energyOutgoing - 1.f) * singleScatteringLobe;
    }
    return vec3(0.f);
}

template <int DIFFUSE_BRDF, int MULTISCATTER, bool GEOMETRIC_AA, int INDIRECT_TECH>
__global__ __launch_bounds__(256) void deferredShadingKernel(ShadeParams P) {
    const int px = P.xBase + (int)(blockIdx.x * 64u + (threadIdx.x & 63u));
    const int py = P.yBase + (int)(blockIdx.y * 4u + (threadIdx.x >> 6));
    if (px >= P.coverW || py >= P.coverH) return;
    const GlobalUbo* g = P.g;
    const vec2 screenRes((float)g->screenResolution[0], (float)g->screenResolution[1]);
    const vec2 fragCoord((float)px + 0.5f, (float)py + 0.5f);
    const vec2 screenUV = fragCoord / screenRes;
    const size_t idx = (size_t)py * (size_t)P.color.w + px;
    const float depth = texelFetch2D<F_D32>(P.depth, px, py).x;
    const vec3 camFwd = ld3(g->cameraForward), camPos = ld3(g->cameraPosition);
    const vec2 pixelNDC(screenUV.x * 2.f - 1.f, screenUV.y * 2.f - 1.f);
    const vec3 Vcam = -calculateViewDirectionFromPixel(pixelNDC, camFwd, ld3(g->cameraUp), ld3(g->cameraRight), g->cameraTanFovHalf, g->cameraAspectRatio);
    if (depth == 0.f) {
        ((uint32_t*)P.color.ptr)[idx] = packR11G11B10(sampleSkyLut(Vcam, P.skyLut));
        return;
    }
    const float depthLinear = linearizeDepth(depth, g->nearPlane, g->farPlane);
    const vec3 passPos = camPos + Vcam / dot(Vcam, camFwd) * depthLinear;

    const vec3 albedoTexel = texelFetch2D<F_RGBA8>(P.albedo, px, py).xyz();
    const vec3 specularTexel = texelFetch2D<F_RGBA8>(P.specular, px, py).xyz();
    const float metalic = specularTexel.z;
    float r = specularTexel.y;
    r = gmax(r * r, 0.0045f);
    const vec3 albedo = sRGBToLinear(albedoTexel);
    const vec3 diffuseColor = (1.f - metalic) * albedo;
    const vec3 N = gbufferNormal(P.normal, px, py);
    const vec3 L = normalize(ld3(g->sunDirection));
    vec3 V = camPos - passPos;
    const float pixelDepth = dot(V, -camFwd);
    V = normalize(V);
    const vec3 H = normalize(V + L);
    if (GEOMETRIC_AA) {
        // GeometricAA.inc:4-19 with the quad differences standing in for dFdxFine / dFdyFine
        const int xl = px & ~1, yl = py & ~1;
        const vec3 N_U = gbufferNormal(P.normal, xl + 1, py) - gbufferNormal(P.normal, xl, py);
        const vec3 N_V = gbufferNormal(P.normal, px, yl + 1) - gbufferNormal(P.normal, px, yl);
        const float variance = (0.5f * 0.5f) * (dot(N_V, N_V) + dot(N_U, N_U));
        const float kernelRoughness2 = gmin(2.f * variance, 0.18f);
        r = gclamp(sqrtf(r * r + kernelRoughness2), 0.f, 1.f);
    }
    const float NoH = gmax(dot(N, H), 0.f);
    const float NoL = gclamp(dot(N, L), 0.f, 1.f);
    const float VoH = fabsf(dot(V, H));
    const float LoV = gmax(dot(L, V), 0.f);
    float NoV = fabsf(dot(N, V));
    NoV = gmax(NoV, 0.0001f);
    const vec3 f0 = vmix(vec3(0.04f), albedo, metalic);

    const uint32_t noiseSlot = (uint32_t)g->noiseTextureIndices[g->frameIndexMod4 & 3u];
    const ImgView noiseTex = P.bindless[min(noiseSlot, P.bindlessCount - 1u)];
    const vec4 noiseTexel = sampleNearest2D<F_RG8, REPEAT>(noiseTex, vec2(fragCoord.x / (float)noiseTex.w, fragCoord.y / (float)noiseTex.h));

    int cascadeIndex = 0;
    for (int cascade = 0; cascade < (int)P.cascadeCount - 1; cascade++) cascadeIndex += (pixelDepth >= P.shadowInfo->splits[cascade]) ? 1 : 0;
    cascadeIndex = min(cascadeIndex, 3);
    const vec2 lss(P.shadowInfo->lightSpaceScale[cascadeIndex][0], P.shadowInfo->lightSpaceScale[cascadeIndex][1]);
    float sunShadow;
    // the four images are separate bindings in the shader; select without dynamically indexing the by-value parameter block
    if (cascadeIndex == 0) sunShadow = calcShadow(passPos, P.shadowMaps[0], P.shadowInfo->lightMatrices[0], lss, noiseTexel.x);
    else if (cascadeIndex == 1) sunShadow = calcShadow(passPos, P.shadowMaps[1], P.shadowInfo->lightMatrices[1], lss, noiseTexel.x);
    else if (cascadeIndex == 2) sunShadow = calcShadow(passPos, P.shadowMaps[2], P.shadowInfo->lightMatrices[2], lss, noiseTexel.x);
    else sunShadow = calcShadow(passPos, P.shadowMaps[3], P.shadowInfo->lightMatrices[3], lss, noiseTexel.x);
    const vec3 directLighting = gmax(dot(N, L), 0.f) * sunShadow * ld3(P.light->sunColor);
    const vec3 brdfLut = sampleLinear2D<F_RGBA16F, CLAMP>(P.brdfLut, vec2(r, NoV)).xyz();

    vec3 diffuseDirect;
    vec3 diffuseBRDFIntegral(1.f);
    if (DIFFUSE_BRDF == 0) {
        diffuseDirect = diffuseColor / PLR_GLSL_PI * directLighting;
        diffuseBRDFIntegral = vec3(brdfLut.z);
    } else if (DIFFUSE_BRDF == 1) {
        diffuseDirect = DisneyDiffuse(diffuseColor, NoL, VoH, NoV, r) * directLighting;
        diffuseBRDFIntegral = vec3(brdfLut.z);
    } else if (DIFFUSE_BRDF == 2) {
        diffuseDirect = CoDWWIIDiffuse(diffuseColor, NoL, VoH, NoV, NoH, r) * directLighting;
        diffuseBRDFIntegral = vec3(brdfLut.z);
    } else {
        diffuseDirect = Titanfall2Diffuse(diffuseColor, NoL, LoV, NoV, NoH, r) * directLighting;
        float multiIntegral = 0.1159f * r * PLR_GLSL_PI * 2.f;
        multiIntegral *= (1.f - F_Schlick(vec3(0.04f), vec3(1.f), NoV).x);
        multiIntegral *= 0.94291f;
        diffuseBRDFIntegral = vmin(vec3(brdfLut.z) + diffuseColor * multiIntegral, vec3(1.f));
    }
    diffuseDirect = diffuseDirect * ((1.f - F_Schlick(f0, vec3(1.f), NoV)) * (1.f - F_Schlick(f0, vec3(1.f), NoL)));

    const vec3 singleScatteringLobe = GGXSingleScattering(r, f0, NoH, NoV, VoH, NoL);
    const vec3 multiScatteringLobe = specularMultiscatteringLobe<MULTISCATTER>(P.brdfLut, r, NoL, f0, singleScatteringLobe, brdfLut);
    const vec3 specularDirect = directLighting * (singleScatteringLobe + multiScatteringLobe);

    vec3 lightingIndirect;
    if (INDIRECT_TECH == 0) {
        const vec4 irradiance_Y_SH = sampleNearest2D<F_RGBA16F, CLAMP>(P.ysh, screenUV);
        const float irradiance_Y = dot(irradiance_Y_SH, directionToSH_L1(N));
        const vec4 cc = sampleNearest2D<F_RG16F, CLAMP>(P.cocg, screenUV);
        const vec3 irradiance = YCoCgToLinear(vec3(irradiance_Y, cc.x, cc.y));
        const vec3 diffuseIndirect = irradiance * diffuseColor * diffuseBRDFIntegral;
        const vec3 dominantDirection = dominantDirectionFromSH_L1(irradiance_Y_SH);
        float dominantDirectionLength = length(dominantDirection);
        dominantDirectionLength = gclamp(dominantDirectionLength, 0.01f, 1.f);
        const float r_indirect = gmix(1.f, r, sqrtf(dominantDirectionLength));
        const vec3 L_indirect = dominantDirection / dominantDirectionLength;
        const vec3 H_indirect = normalize(L_indirect + V);
        const float NoH_indirect = gmax(dot(N, H_indirect), 0.f);
        const float NoL_indirect = gmax(dot(N, L_indirect), 0.f);
        const float VoH_indirect = gmax(dot(V, H_indirect), 0.f);
        const vec3 single_i = GGXSingleScattering(r_indirect, f0, NoH_indirect, NoV, VoH_indirect, NoL_indirect);
        const vec3 multi_i = specularMultiscatteringLobe<MULTISCATTER>(P.brdfLut, r_indirect, NoL_indirect, f0, single_i, brdfLut);
        const vec3 specularIndirect = (single_i + multi_i) * YCoCgToLinear(vec3(irradiance_Y_SH.x, cc.x, cc.y));
        lightingIndirect = diffuseIndirect + specularIndirect;
    } else {
        const float ambientStrength = 0.003f;
        const vec3 irradiance = vec3(ambientStrength) * P.light->sunStrengthExposed;
        const vec3 reflection = vec3(ambientStrength) * P.light->sunStrengthExposed;
        const vec3 singleScattering = vmix(vec3(brdfLut.x), vec3(brdfLut.y), f0);
        const vec3 diffuseIndirect = irradiance * diffuseColor * diffuseBRDFIntegral;
        const vec3 specularIndirect = singleScattering * reflection;
        lightingIndirect = diffuseIndirect + specularIndirect;
    }
    vec3 outColor = (diffuseDirect + specularDirect) * P.light->sunStrengthExposed + lightingIndirect;

    // applyVolumetricLighting, triangle.frag:133-144 + volumetricFroxelLighting.inc:33-53 (exponential slices, k = 3)
    {
        vec2 noise(noiseTexel.x, noiseTexel.y);
        noise = noise - 0.5f;
        noise = noise * 0.013f;
        vec2 suv = fragCoord / screenRes;
        suv += noise;
        const float k = 3.f;
        const float linear = pixelDepth / P.vol->maxDistance;
        const float z = det_logf(linear * (det_expf(k) - 1.f) + 1.f) / k;
        // trilinear RGBA16F lookup, clamp to edge
        const ImgView& vol = P.volumetricLut;
        int i0, j0, k0; float a, b, c;
        linearCoord(suv.x * (float)vol.w, &i0, &a);
        linearCoord(suv.y * (float)vol.h, &j0, &b);
        linearCoord(z * (float)vol.d, &k0, &c);
        vec4 it(0.f);
        for (int dz = 0; dz < 2; dz++)
            for (int dy = 0; dy < 2; dy++)
                for (int dx = 0; dx < 2; dx++) {
                    const float wx = dx ? a : (1.f - a), wy = dy ? b : (1.f - b), wz = dz ? c : (1.f - c);
                    const size_t ti = ((size_t)clampi(k0 + dz, vol.d) * (size_t)vol.h + (size_t)clampi(j0 + dy, vol.h)) * (size_t)vol.w + (size_t)clampi(i0 + dx, vol.w);
                    it = it + Texel<F_RGBA16F>::load(vol.ptr, ti) * ((wx * wy) * wz);
                }
        outColor = outColor * it.w + it.xyz();
    }
    ((uint32_t*)P.color.ptr)[idx] = packR11G11B10(outColor);
}

typedef void (*ShadeKernel)(ShadeParams);
template <int D, int M, bool G> static ShadeKernel pickIndirect(int tech) {
    return tech == 0 ? (ShadeKernel)deferredShadingKernel<D, M, G, 0> : (ShadeKernel)deferredShadingKernel<D, M, G, 1>;
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

static int launchDeferredShading(const PassCtx& c) {
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
    const PassCtx::ColSpan cs = c.colSpan(P.color.w);
    P.coverW = cs.x1; P.xBase = cs.x0; P.coverH = rs.y1; P.yBase = rs.y0; // columns [xBase, coverW), rows [yBase, coverH)
    if (P.coverW <= P.xBase || P.coverH <= P.yBase) return 0;
    k<<<dim3(divUp((unsigned)(P.coverW - P.xBase), 64u), divUp((unsigned)(P.coverH - P.yBase), 4u)), 256, 0, c.stream>>>(P);
    PLR_CHECK_LAUNCH(c);
    return 0;
}
PLR_REGISTER_SHADER("deferredShading.comp", launchDeferredShading);

} // namespace plr
