// Device functions of the SDF ray march in the reference's operation order (SDF.inc, sunShadowCascades.inc): shared by sdfDebugVisualisation.comp
// (kernels/sdfgi.hip, one kernel for both math modes) and the exact-set sdfDiffuseTrace.comp (kernels_exact/sdf_trace_exact.hip, libplr_exact.so).
#pragma once
#include "shading_common.h"
#include "culling_device.h"

namespace plr {

// trilinear, clamp-to-edge sample of an R16F volume (sampler contract: 8-bit sub-texel weights)
PLR_DI float sampleSDF(const ImgView& v, vec3 uvw) {
    int i0, j0, k0; float a, b, c;
    linearCoord(uvw.x * (float)v.w, &i0, &a);
    linearCoord(uvw.y * (float)v.h, &j0, &b);
    linearCoord(uvw.z * (float)v.d, &k0, &c);
    const int x0 = clampi(i0, v.w), x1 = clampi(i0 + 1, v.w);
    const int y0 = clampi(j0, v.h) * v.w, y1 = clampi(j0 + 1, v.h) * v.w;
    const int sl = v.w * v.h;
    const int z0 = clampi(k0, v.d) * sl, z1 = clampi(k0 + 1, v.d) * sl;
    const uint16_t* p = (const uint16_t*)v.ptr;
    const float t000 = halfBitsToFloat(p[z0 + y0 + x0]), t100 = halfBitsToFloat(p[z0 + y0 + x1]);
    const float t010 = halfBitsToFloat(p[z0 + y1 + x0]), t110 = halfBitsToFloat(p[z0 + y1 + x1]);
    const float t001 = halfBitsToFloat(p[z1 + y0 + x0]), t101 = halfBitsToFloat(p[z1 + y0 + x1]);
    const float t011 = halfBitsToFloat(p[z1 + y1 + x0]), t111 = halfBitsToFloat(p[z1 + y1 + x1]);
    const float a0 = 1.f - a, b0 = 1.f - b, c0 = 1.f - c;
    float r = t000 * ((a0 * b0) * c0);
    r = r + t100 * ((a * b0) * c0);
    r = r + t010 * ((a0 * b) * c0);
    r = r + t110 * ((a * b) * c0);
    r = r + t001 * ((a0 * b0) * c);
    r = r + t101 * ((a * b0) * c);
    r = r + t011 * ((a0 * b) * c);
    r = r + t111 * ((a * b) * c);
    return r;
}

// SDF.inc:16-25
PLR_DI vec3 normalFromSDF(vec3 uv, vec3 extends, const ImgView& sdf) {
    const float extendsMax = gmax(extends.x, gmax(extends.y, extends.z));
    const vec3 extendsNormalized = extends / extendsMax;
    const vec3 epsilon = vec3(0.15f) / vec3((float)sdf.w, (float)sdf.h, (float)sdf.d) / extendsNormalized;
    return normalize(vec3(sampleSDF(sdf, uv + vec3(epsilon.x, 0.f, 0.f)) - sampleSDF(sdf, uv - vec3(epsilon.x, 0.f, 0.f)),
                          sampleSDF(sdf, uv + vec3(0.f, epsilon.y, 0.f)) - sampleSDF(sdf, uv - vec3(0.f, epsilon.y, 0.f)),
                          sampleSDF(sdf, uv + vec3(0.f, 0.f, epsilon.z)) - sampleSDF(sdf, uv - vec3(0.f, 0.f, epsilon.z))));
}

struct TraceResult {
    bool hit;
    float closestHitDistance;
    vec3 hitPos;
    vec3 albedo;
    vec3 N;       // only written by the WITH_NORMAL instantiation (sdfDebugVisualisation.comp)
    int hitCount; // "
};

// SDF.inc:42-86
PLR_DI bool rayAABBIntersection(vec3 o, vec3 dir, vec3 mn, vec3 mx, float* tOut) {
    bool hit = false;
    float t = 100000.f;
    float intersection = o.x < 0.f ? mn.x : mx.x;
    const float tx = (intersection - o.x) / dir.x;
    vec3 p = o + tx * dir;
    if (tx > 0.f && p.y >= mn.y && p.y <= mx.y && p.z >= mn.z && p.z <= mx.z) { t = gmin(t, tx); hit = true; }
    intersection = o.y < 0.f ? mn.y : mx.y;
    const float ty = (intersection - o.y) / dir.y;
    p = o + ty * dir;
    if (ty > 0.f && p.x >= mn.x && p.x <= mx.x && p.z >= mn.z && p.z <= mx.z) { t = gmin(t, ty); hit = true; }
    intersection = o.z < 0.f ? mn.z : mx.z;
    const float tz = (intersection - o.z) / dir.z;
    p = o + tz * dir;
    if (tz > 0.f && p.x >= mn.x && p.x <= mx.x && p.y >= mn.y && p.y <= mx.y) { t = gmin(t, tz); hit = true; }
    *tOut = t;
    return hit;
}

// SDF.inc:101-184. `inst` and `sdf` are wave uniform (diffuse trace) or per lane (debug visualisation).
template <bool WITH_NORMAL = false>
PLR_DI void traceRayTroughSDFInstance(const SDFInstance& inst, vec3 rayStartWorld, const ImgView& sdf, vec3 rayDirectionWorld, TraceResult& tr) {
    const float* m = inst.worldToLocal;
    const vec3 localExtends = ld3(inst.localExtends);
    vec3 rayStartLocal = mulMat4(m, vec4(rayStartWorld, 1.f)).xyz();
    const vec3 rayEndLocal = mulMat4(m, vec4(rayStartWorld + rayDirectionWorld, 1.f)).xyz();
    vec3 rayDirection = rayEndLocal - rayStartLocal;
    rayDirection /= length(rayDirection);
    const vec3 sdfMaxLocal = localExtends * 0.5f;
    const vec3 sdfMinLocal = -sdfMaxLocal;
    float hitDistanceLocal = 0.f;
    const bool inside = rayStartLocal.x >= sdfMinLocal.x && rayStartLocal.y >= sdfMinLocal.y && rayStartLocal.z >= sdfMinLocal.z &&
                        rayStartLocal.x <= sdfMaxLocal.x && rayStartLocal.y <= sdfMaxLocal.y && rayStartLocal.z <= sdfMaxLocal.z;
    if (!inside) {
        float t;
        if (rayAABBIntersection(rayStartLocal, rayDirection, sdfMinLocal, sdfMaxLocal, &t)) {
            rayStartLocal += t * rayDirection;
            hitDistanceLocal = t;
        } else return;
    }
    vec3 localSamplePos = rayStartLocal;
    const float distanceThreshold = length(localExtends / vec3((float)sdf.w, (float)sdf.h, (float)sdf.d)) * 0.25f;
    float dLast = 0.f, d = 0.f;
    const float localToGlobalScale = 1.f / length(vec3(m[0], m[1], m[2]));
    if (localToGlobalScale * hitDistanceLocal > tr.closestHitDistance) return;
    vec3 localExtendsHalf = localExtends * 0.5f;
    localExtendsHalf = localExtendsHalf + 0.01f;
    for (int i = 0; i < 128; i++) {
        if (localSamplePos.x > localExtendsHalf.x || localSamplePos.y > localExtendsHalf.y || localSamplePos.z > localExtendsHalf.z ||
            localSamplePos.x < -localExtendsHalf.x || localSamplePos.y < -localExtendsHalf.y || localSamplePos.z < -localExtendsHalf.z)
            break;
        vec3 sampleUV = localSamplePos / localExtends + 0.5f;
        dLast = d;
        d = sampleSDF(sdf, sampleUV);
        if (d < distanceThreshold) {
            tr.hit = true;
            const float distanceGlobal = hitDistanceLocal * localToGlobalScale;
            if (distanceGlobal < tr.closestHitDistance) {
                tr.closestHitDistance = distanceGlobal;
                const float lastStepSizeLocal = d / (1.f - (d - dLast));
                localSamplePos += rayDirection * lastStepSizeLocal;
                // the reference also evaluates normalFromSDF and the transformed normal here; neither reaches an output of
                // sdfDiffuseTrace.comp (only the debug visualisation reads traceResult.N and hitCount), so they are computed on request
                if (WITH_NORMAL) {
                    tr.hitCount = i;
                    const vec3 nUV = localSamplePos / localExtends + 0.5f;
                    const vec3 nl = normalFromSDF(nUV, localExtends, sdf);
                    // transpose(mat3(worldToLocal)) * N
                    tr.N = vec3(m[0] * nl.x + m[1] * nl.y + m[2] * nl.z, m[4] * nl.x + m[5] * nl.y + m[6] * nl.z, m[8] * nl.x + m[9] * nl.y + m[10] * nl.z);
                }
                tr.albedo = vpow(ld3(inst.meanAlbedo), 2.2f);
                const float lastStepSizeGlobal = lastStepSizeLocal * localToGlobalScale;
                tr.hitPos = rayStartWorld + rayDirectionWorld * (distanceGlobal + lastStepSizeGlobal);
            }
            break;
        }
        localSamplePos += rayDirection * fabsf(d);
        hitDistanceLocal += fabsf(d);
    }
}

// sunShadowCascades.inc:13-20 with the nearest / white-border sampler and a D16 map
PLR_DI float simpleShadow(vec3 posWorld, const float* lightMatrix, const ImgView& shadowMap) {
    vec4 p = mulMat4(lightMatrix, vec4(posWorld, 1.f));
    p = p / p.w;
    const vec2 xy(p.x * 0.5f + 0.5f, p.y * 0.5f + 0.5f);
    const float actualDepth = gclamp(p.z, 0.f, 1.f);
    const float shadowMapDepth = sampleNearest2D<F_D16, BORDER_WHITE>(shadowMap, xy).x;
    return actualDepth > shadowMapDepth ? 1.f : 0.f;
}

struct SdfInstanceBuffer { uint32_t instanceCount, pad1, pad2, pad3; SDFInstance instances[1]; };
struct RayInfo { float nx, ny, nz, depth, cr, cg, cb; };

} // namespace plr
