// ORACLE (test infrastructure, never shipped, never on the product path).
// Restates: resources/shaders/brdfLut.comp and the lighting of resources/shaders/triangle.frag:84-341 (+ brdf.inc,
// GeometricAA.inc, sunShadowCascades.inc, volumetricFroxelLighting.inc, SphericalHarmonics.inc, colorConversion.inc).
//
// The reference shades forward inside a raster pass; north_star asks for a deferred pass over a synthetic G-buffer.
// The re-expression ("deferredShading.comp") feeds main()'s interpolants from the G-buffer:
//   gl_FragCoord.xy = iUV + 0.5; passPos = world position rebuilt from depth exactly as sdfDiffuseTrace.comp:120-126 does
//   (uv = gl_FragCoord / resolution); albedo/specular texels come from two RGBA8 images; N = normalize(G-buffer normal)
//   (the prepass stores the vertex normal, depthPrepass.frag:46-48); dFdxFine/dFdyFine(N) are the differences inside the
//   pixel's 2x2 quad; depth == 0 (sky) pixels receive sampleSkyLut(view direction) as a stand-in for the sky pass.
#include "common.h"

using namespace orc;

namespace {

// ---- brdf.inc
float D_GGX(float NoH, float r) {
    const float a = NoH * r;
    const float k = r / (1.0f - NoH * NoH + a * a);
    return k * k * (1.0f / pi);
}
float Visibility(float NoV, float NoL, float r) {
    const float r_2 = r * r;
    const float v1 = NoL * std::sqrt(NoV * NoV * (1.f - r_2) + r_2);
    const float v2 = NoV * std::sqrt(NoL * NoL * (1.f - r_2) + r_2);
    return 0.5f / (v1 + v2);
}
vec3 F_Schlick(vec3 f0, vec3 f90, float VoH) { return f0 + (f90 - f0) * det_powf(1.f - VoH, 5.f); }

vec3 DisneyDiffuse(vec3 diffuseColor, float NoL, float VoH, float NoV, float r) {
    const float energyBias = gmix(0.f, 0.5f, r);
    const float energyFactor = gmix(1.f, 1.f / 1.51f, r);
    const float fresnelDiffuse90Biased = energyBias + 2.f * VoH * VoH * r;
    return diffuseColor / pi * F_Schlick(vec3(1.f), vec3(fresnelDiffuse90Biased), NoL) * F_Schlick(vec3(1.f), vec3(fresnelDiffuse90Biased), NoV) * energyFactor;
}
vec3 CoDWWIIDiffuse(vec3 diffuseColor, float NoL, float VoH, float NoV, float NoH, float r) {
    const float f0Diffuse = VoH + det_powf(1.f - VoH, 5.f);
    const float f1 = (1.f - 0.75f * det_powf(1.f - NoL, 5.f)) * (1.f - 0.75f * det_powf(1.f - NoV, 5.f));
    const float g = det_log2f(2.f / (r * r) - 1.f) / 18.f;
    const float t = gclamp(2.2f * g - 0.5f, 0.f, 1.f);
    const float fd = f0Diffuse + (f1 - f0Diffuse) * t;
    const float fb = (34.5f * g * g - 59.f * g + 24.5f) * VoH * det_powf(2.f, -gmax(73.2f * g - 21.2f, 8.9f) * std::sqrt(NoH));
    return diffuseColor / pi * (fd + fb);
}
float Titanfall2DiffuseSingleComponent(float NoL, float LoV, float NoV, float NoH, float r) {
    const float facing = 0.5f + 0.5f * LoV;
    const float rough = facing * (0.9f - 0.4f * facing) * (0.5f + NoH) / gmax(NoH, 0.03f);
    const float smoothDiffuse = 1.05f * (1.f - det_powf(1.f - NoL, 5.f)) * (1.f - det_powf(1.f - NoV, 5.f));
    return 1.f / pi * gmix(smoothDiffuse, rough, r);
}
vec3 Titanfall2Diffuse(vec3 diffuseColor, float NoL, float LoV, float NoV, float NoH, float r) {
    const float single = Titanfall2DiffuseSingleComponent(NoL, LoV, NoV, NoH, r);
    const float multi = 0.1159f * r;
    return diffuseColor * (single + diffuseColor * multi);
}
vec3 GGXSingleScattering(float r, vec3 f0, float NoH, float NoV, float VoH, float NoL) {
    const float D = D_GGX(NoH, r);
    const float Vis = Visibility(NoV, NoL, r);
    const vec3 F = F_Schlick(f0, vec3(1.f), VoH);
    return D * Vis * F;
}

// sampling.inc:47-58
float radicalInverse_VdC(uint32_t bits) {
    bits = (bits << 16u) | (bits >> 16u);
    bits = ((bits & 0x55555555u) << 1u) | ((bits & 0xAAAAAAAAu) >> 1u);
    bits = ((bits & 0x33333333u) << 2u) | ((bits & 0xCCCCCCCCu) >> 2u);
    bits = ((bits & 0x0F0F0F0Fu) << 4u) | ((bits & 0xF0F0F0F0u) >> 4u);
    bits = ((bits & 0x00FF00FFu) << 8u) | ((bits & 0xFF00FF00u) >> 8u);
    return (float)bits * 2.3283064365386963e-10f;
}

} // namespace

// brdfLut.comp:20-101
extern "C" void orc_brdf_lut(const orc_image* lutP, int32_t diffuseBRDF) {
    const Image& lut = img(lutP);
    parallelFor(lut.h, [&](int y0, int y1) {
        for (int uy = y0; uy < y1; uy++)
            for (int ux = 0; ux < lut.w; ux++) {
                float r = (float)ux / (float)lut.w;
                r = gmax(r, 0.0001f);
                const float NoV = gmax((float)uy, 0.1f) / (float)lut.h;
                const vec3 V(std::sqrt(1.0f - NoV * NoV), 0.f, NoV);
                const vec3 N(0.f, 0.f, 1.f);
                const int samples = 1024;
                vec3 result(0.f);
                for (int i = 0; i < samples; i++) {
                    const vec2 xi((float)i / (float)samples, radicalInverse_VdC((uint32_t)i));
                    {
                        const vec3 Hh = importanceSampleGGX(xi, r, N);
                        const vec3 L = 2.f * dot(V, Hh) * Hh - V;
                        const float VoH = gmax(dot(V, Hh), 0.f);
                        const float NoH = gmax(Hh.z, 0.f);
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
                        const vec3 Hh = normalize(V + L);
                        const float VoH = gclamp(dot(V, Hh), 0.f, 1.f);
                        const float NoL = gmax(L.z, 0.f);
                        const float NoH = gmax(Hh.z, 0.f);
                        const vec3 F0Diffuse(0.04f);
                        const float fresnelInOut = (1.f - F_Schlick(F0Diffuse, vec3(1.f), NoV).x) * (1.f - F_Schlick(F0Diffuse, vec3(1.f), NoL).x);
                        if (diffuseBRDF == 0) result.z += (1.f / pi) * fresnelInOut;
                        else if (diffuseBRDF == 1) result.z += DisneyDiffuse(vec3(1.f), NoL, VoH, NoV, r).x * fresnelInOut;
                        else if (diffuseBRDF == 2) result.z += CoDWWIIDiffuse(vec3(1.f), NoL, VoH, NoV, NoH, r).x * fresnelInOut;
                        else if (diffuseBRDF == 3) {
                            const float LoV = gclamp(dot(L, V), 0.f, 1.f);
                            result.z += Titanfall2DiffuseSingleComponent(NoL, LoV, NoV, NoH, r) * fresnelInOut;
                        }
                    }
                }
                result /= (float)samples;
                result.x *= 4.f;
                result.y *= 4.f;
                imageStore(lut, ivec2(ux, uy), vec4(result, 0.f));
            }
    });
}

namespace {

struct ShadeCtx {
    const Image *brdfLut, *shadowMaps, *ysh, *cocg, *volumetricLut, *noiseTex;
    const orc_light_buffer* light;
    const orc_shadow_cascade_info* shadowInfo;
    const orc_volumetric_settings* vol;
    const orc_global* g;
    int diffuseBRDF, multiscatter, indirectTech;
};

// triangle.frag:84-120
float calcShadow(const ShadeCtx& c, vec3 pos, const Image& shadowMap, const mat4& lightMatrix, int cascade, vec2 fragCoord) {
    vec4 p = lightMatrix * vec4(pos, 1.f);
    p = p / p.w;
    const vec2 xy = vec2(p.x, p.y) * 0.5f + 0.5f;
    const float actualDepth = gclamp(p.z, 0.f, 1.f);
    const vec2 noiseUV = fragCoord / vec2((float)c.noiseTex->w, (float)c.noiseTex->h);
    const float noise = texture2D(*c.noiseTex, NEAREST, REPEAT, noiseUV).x;
    const vec2 offsetScale = 0.03f * vec2(c.shadowInfo->lightSpaceScale[cascade][0], c.shadowInfo->lightSpaceScale[cascade][1]);
    float shadow = 0.f;
    const float sampleCount = 12.f;
    for (int i = 0; (float)i < sampleCount; i++) {
        float d = ((float)i + 0.5f * noise) / sampleCount;
        d = std::sqrt(d);
        const float angle = noise * 2.f * pi + 2.f * pi * (float)i / sampleCount;
        float sa, ca;
        det_sincosf(angle, &sa, &ca);
        vec2 offset(ca, sa);
        offset *= offsetScale * d;
        const vec2 samplePosition = xy + offset;
        const float depthTexel = texture2D(shadowMap, NEAREST, BORDER_BLACK, samplePosition).x;
        shadow += (actualDepth >= depthTexel) ? 1.f : 0.f;
    }
    return shadow / sampleCount;
}

// triangle.frag:123-131
float ReflectedEnergyAverage(float roughness) {
    const float smoothness = 1.f - std::sqrt(roughness);
    float r = -0.0761947f - 0.383026f * smoothness;
    r = 1.04997f + smoothness * r;
    r = 0.409255f + smoothness * r;
    return gmin(0.999f, r);
}

// triangle.frag:146-175
vec3 computeSpecularMultiscatteringLobe(const ShadeCtx& c, float r, float NoL, vec3 f0, vec3 singleScatteringLobe, vec3 brdfLut) {
    vec3 multiScatteringLobe;
    const float energyOutgoing = brdfLut.y;
    const vec3 fresnelAverage = f0 + (1.f - f0) / 21.f;
    if (c.multiscatter == 0) {
        const float energyAverage = ReflectedEnergyAverage(r);
        const float energyIncoming = texture2D(*c.brdfLut, LINEAR, CLAMP, vec2(r, NoL)).y;
        const float multiScatteringLobeUnscaled = (1.f - energyIncoming) * (1.f - energyOutgoing) / (3.1415f * (1.f - energyAverage));
        const vec3 multiScatteringScaling = (fresnelAverage * fresnelAverage * energyAverage) / (1.f - fresnelAverage * (1.f - energyAverage));
        multiScatteringLobe = multiScatteringLobeUnscaled * multiScatteringScaling;
    } else if (c.multiscatter == 1) {
        multiScatteringLobe = vec3((1.f - energyOutgoing) / pi);
        const vec3 multiScatteringScaling = (fresnelAverage * fresnelAverage * energyOutgoing) / (1.f - fresnelAverage * (1.f - energyOutgoing));
        multiScatteringLobe *= multiScatteringScaling;
    } else if (c.multiscatter == 2) {
        multiScatteringLobe = f0 * (1.f / energyOutgoing - 1.f) * singleScatteringLobe;
    } else {
        multiScatteringLobe = vec3(0.f);
    }
    return multiScatteringLobe;
}

// GeometricAA.inc:4-19 with explicit quad derivatives
float modifiedRoughnessGeometricAA(vec3 N_U, vec3 N_V, float r) {
    const float kappa = 0.18f;
    const float pixelVariance = 0.5f;
    const float pxVar2 = pixelVariance * pixelVariance;
    const float lengthN_U2 = dot(N_U, N_U);
    const float lengthN_V2 = dot(N_V, N_V);
    const float variance = pxVar2 * (lengthN_V2 + lengthN_U2);
    const float kernelRoughness2 = gmin(2.f * variance, kappa);
    return gclamp(std::sqrt(r * r + kernelRoughness2), 0.f, 1.f);
}

// volumetricFroxelLighting.inc:33-53 (exponentialDepthDistribution = true, k = 3)
vec4 volumeTextureLookup(vec2 screenUV, float depth, const Image& froxelTexture, float maxDistance) {
    const float k = 3.f;
    const float linear = depth / maxDistance;
    const float z = det_logf(linear * (det_expf(k) - 1.f) + 1.f) / k;
    return texture3D(froxelTexture, LINEAR, CLAMP, vec3(screenUV.x, screenUV.y, z));
}

vec3 gbufferNormal(const Image& normalTexture, int x, int y) {
    x = std::min(std::max(x, 0), normalTexture.w - 1);
    y = std::min(std::max(y, 0), normalTexture.h - 1);
    const vec3 raw = loadTexel(normalTexture, x, y).xyz() * 2.f - 1.f;
    vec3 N = normalize(raw);
    if (isnan3(N)) N = raw;
    return N;
}

} // namespace

extern "C" void orc_deferred_shading(const orc_image* colorP, const orc_image* depthP, const orc_image* normalP, const orc_image* albedoP,
                                     const orc_image* specularP, const orc_image* brdfLutP, const orc_light_buffer* light,
                                     const orc_shadow_cascade_info* shadowInfo, const orc_image* shadowMaps4, const orc_image* yshP, const orc_image* cocgP,
                                     const orc_image* volumetricLutP, const orc_volumetric_settings* volSettings, const orc_image* skyLutP,
                                     const orc_image* bindless, int32_t nBindless, const orc_global* g, int32_t diffuseBRDF, int32_t directMultiscatterBRDF,
                                     int32_t geometricAA, int32_t indirectLightingTech, uint32_t sunShadowCascadeCount) {
    const Image &color = img(colorP), &depthTexture = img(depthP), &normalTexture = img(normalP), &albedoTexture = img(albedoP), &specularTexture = img(specularP),
                &skyLut = img(skyLutP);
    ShadeCtx c;
    c.brdfLut = &img(brdfLutP);
    c.shadowMaps = reinterpret_cast<const Image*>(shadowMaps4);
    c.ysh = &img(yshP); c.cocg = &img(cocgP); c.volumetricLut = &img(volumetricLutP);
    c.noiseTex = &img(&bindless[g->noiseTextureIndices[g->frameIndexMod4]]);
    c.light = light; c.shadowInfo = shadowInfo; c.vol = volSettings; c.g = g;
    c.diffuseBRDF = diffuseBRDF; c.multiscatter = directMultiscatterBRDF; c.indirectTech = indirectLightingTech;
    const vec2 screenRes((float)g->screenResolution[0], (float)g->screenResolution[1]);
    const vec3 camFwd = v3(g->cameraForward), camPos = v3(g->cameraPosition);
    parallelFor(color.h, [&](int y0, int y1) {
        for (int py = y0; py < y1; py++)
            for (int px = 0; px < color.w; px++) {
                const ivec2 iUV(px, py);
                const vec2 fragCoord = toVec2(iUV) + 0.5f;
                const vec2 screenUV = fragCoord / screenRes;
                const float depth = texelFetch(depthTexture, iUV).x;
                const vec2 pixelNDC = screenUV * 2.f - 1.f;
                const vec3 Vcam = -calculateViewDirectionFromPixel(pixelNDC, camFwd, v3(g->cameraUp), v3(g->cameraRight), g->cameraTanFovHalf, g->cameraAspectRatio);
                if (depth == 0.f) { // sky stand-in
                    imageStore(color, iUV, vec4(sampleSkyLut(Vcam, skyLut), 1.f));
                    writeSig((int64_t)py * color.w + px, 128u);
                    continue;
                }
                const float depthLinear = linearizeDepth(depth, g->nearPlane, g->farPlane);
                const vec3 passPos = camPos + Vcam / dot(Vcam, camFwd) * depthLinear;

                // ---- triangle.frag main()
                const vec3 albedoTexel = texelFetch(albedoTexture, iUV).xyz();
                const vec3 specularTexel = texelFetch(specularTexture, iUV).xyz();
                const float metalic = specularTexel.z;
                float r = specularTexel.y;
                r = gmax(r * r, 0.0045f);
                const vec3 albedo = sRGBToLinear(albedoTexel);
                const vec3 diffuseColor = (1.f - metalic) * albedo;
                const vec3 N = gbufferNormal(normalTexture, px, py);
                const vec3 L = normalize(v3(g->sunDirection));
                vec3 V = camPos - passPos;
                const float pixelDepth = dot(V, -camFwd);
                V = normalize(V);
                const vec3 Hh = normalize(V + L);
                if (geometricAA) {
                    const int xl = px & ~1, yl = py & ~1;
                    const vec3 N_U = gbufferNormal(normalTexture, xl + 1, py) - gbufferNormal(normalTexture, xl, py);
                    const vec3 N_V = gbufferNormal(normalTexture, px, yl + 1) - gbufferNormal(normalTexture, px, yl);
                    r = modifiedRoughnessGeometricAA(N_U, N_V, r);
                }
                const float NoH = gmax(dot(N, Hh), 0.f);
                const float NoL = gclamp(dot(N, L), 0.f, 1.f);
                const float VoH = std::fabs(dot(V, Hh));
                const float LoV = gmax(dot(L, V), 0.f);
                float NoV = std::fabs(dot(N, V));
                NoV = gmax(NoV, 0.0001f);
                const vec3 f0 = mix(vec3(0.04f), albedo, metalic);

                int cascadeIndex = 0;
                for (int cascade = 0; cascade < (int)sunShadowCascadeCount - 1; cascade++) cascadeIndex += (pixelDepth >= shadowInfo->splits[cascade]) ? 1 : 0;
                const float sunShadow = calcShadow(c, passPos, c.shadowMaps[cascadeIndex], toMat4(shadowInfo->lightMatrices[cascadeIndex]), cascadeIndex, fragCoord);
                // decision signature: cascade, number of lit PCF taps (sunShadow = count / 12 exactly representable counts), geometry pixel
                writeSig((int64_t)py * color.w + px, (uint32_t)cascadeIndex | ((uint32_t)(sunShadow * 12.f + 0.5f) << 2) | 64u);
                const vec3 directLighting = gmax(dot(N, L), 0.f) * sunShadow * v3(light->sunColor);
                const vec3 brdfLut = texture2D(*c.brdfLut, LINEAR, CLAMP, vec2(r, NoV)).xyz();

                vec3 diffuseDirect;
                vec3 diffuseBRDFIntegral(1.f);
                if (diffuseBRDF == 0) {
                    diffuseDirect = diffuseColor / pi * directLighting;
                    diffuseBRDFIntegral = vec3(brdfLut.z);
                } else if (diffuseBRDF == 1) {
                    diffuseDirect = DisneyDiffuse(diffuseColor, NoL, VoH, NoV, r) * directLighting;
                    diffuseBRDFIntegral = vec3(brdfLut.z);
                } else if (diffuseBRDF == 2) {
                    diffuseDirect = CoDWWIIDiffuse(diffuseColor, NoL, VoH, NoV, NoH, r) * directLighting;
                    diffuseBRDFIntegral = vec3(brdfLut.z);
                } else {
                    diffuseDirect = Titanfall2Diffuse(diffuseColor, NoL, LoV, NoV, NoH, r) * directLighting;
                    float multiIntegral = 0.1159f * r * pi * 2.f;
                    multiIntegral *= (1.f - F_Schlick(vec3(0.04f), vec3(1.f), NoV).x);
                    multiIntegral *= 0.94291f;
                    diffuseBRDFIntegral = min(vec3(brdfLut.z) + diffuseColor * multiIntegral, vec3(1.f));
                }
                diffuseDirect *= (1.f - F_Schlick(f0, vec3(1.f), NoV)) * (1.f - F_Schlick(f0, vec3(1.f), NoL));

                const vec3 singleScatteringLobe = GGXSingleScattering(r, f0, NoH, NoV, VoH, NoL);
                const vec3 multiScatteringLobe = computeSpecularMultiscatteringLobe(c, r, NoL, f0, singleScatteringLobe, brdfLut);
                const vec3 specularDirect = directLighting * (singleScatteringLobe + multiScatteringLobe);

                vec3 lightingIndirect;
                if (indirectLightingTech == 0) {
                    const vec4 irradiance_Y_SH = texture2D(*c.ysh, NEAREST, CLAMP, screenUV);
                    const float irradiance_Y = dot(irradiance_Y_SH, directionToSH_L1(N));
                    const vec4 cc = texture2D(*c.cocg, NEAREST, CLAMP, screenUV);
                    const vec3 irradiance = YCoCgToLinear(vec3(irradiance_Y, cc.x, cc.y));
                    const vec3 diffuseIndirect = irradiance * diffuseColor * diffuseBRDFIntegral;
                    const vec3 dominantDirection = dominantDirectionFromSH_L1(irradiance_Y_SH);
                    float dominantDirectionLength = length(dominantDirection);
                    dominantDirectionLength = gclamp(dominantDirectionLength, 0.01f, 1.f);
                    const float r_indirect = gmix(1.f, r, std::sqrt(dominantDirectionLength));
                    const vec3 L_indirect = dominantDirection / dominantDirectionLength;
                    const vec3 H_indirect = normalize(L_indirect + V);
                    const float NoH_indirect = gmax(dot(N, H_indirect), 0.f);
                    const float NoL_indirect = gmax(dot(N, L_indirect), 0.f);
                    const float VoH_indirect = gmax(dot(V, H_indirect), 0.f);
                    const vec3 singleScattering_indirect = GGXSingleScattering(r_indirect, f0, NoH_indirect, NoV, VoH_indirect, NoL_indirect);
                    const vec3 multiScattering_indirect = computeSpecularMultiscatteringLobe(c, r_indirect, NoL_indirect, f0, singleScattering_indirect, brdfLut);
                    const vec3 specularIndirect = (singleScattering_indirect + multiScattering_indirect) * YCoCgToLinear(vec3(irradiance_Y_SH.x, cc.x, cc.y));
                    lightingIndirect = diffuseIndirect + specularIndirect;
                } else {
                    const float ambientStrength = 0.003f;
                    const vec3 irradiance = vec3(ambientStrength) * light->sunStrengthExposed;
                    const vec3 reflection = vec3(ambientStrength) * light->sunStrengthExposed;
                    const vec3 singleScattering = mix(vec3(brdfLut.x), vec3(brdfLut.y), f0);
                    const vec3 diffuseIndirect = irradiance * diffuseColor * diffuseBRDFIntegral;
                    const vec3 specularIndirect = singleScattering * reflection;
                    lightingIndirect = diffuseIndirect + specularIndirect;
                }
                vec3 outColor = (diffuseDirect + specularDirect) * light->sunStrengthExposed + lightingIndirect;

                // applyVolumetricLighting (:133-144)
                {
                    const vec2 noiseUV = fragCoord / vec2((float)c.noiseTex->w, (float)c.noiseTex->h);
                    const vec4 nz = texture2D(*c.noiseTex, NEAREST, REPEAT, noiseUV);
                    vec2 noise(nz.x, nz.y);
                    noise -= 0.5f;
                    noise *= 0.013f;
                    vec2 suv = fragCoord / screenRes;
                    suv += noise;
                    const vec4 it = volumeTextureLookup(suv, pixelDepth, *c.volumetricLut, volSettings->maxDistance);
                    outColor = outColor * it.w + it.xyz();
                }
                imageStore(color, iUV, vec4(outColor, 1.f));
            }
    });
}
