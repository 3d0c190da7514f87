// Input-producing compute passes: lightMatrix.comp. They are tiny fixed-size dispatches; the point of having them
// behind the same pass API is that the frame is closed under compute. Exact kernel set: operation order = the shader's (oracle/producers.cpp).
#include "../backend.h"
#include "../device/shading_common.h"

namespace plr {

struct M4 { float c[4][4]; }; // column major: c[col][row]
PLR_DI M4 mulM4(const M4& a, const M4& b) {
    M4 r;
    for (int col = 0; col < 4; col++)
        for (int row = 0; row < 4; row++) r.c[col][row] = a.c[0][row] * b.c[col][0] + a.c[1][row] * b.c[col][1] + a.c[2][row] * b.c[col][2] + a.c[3][row] * b.c[col][3];
    return r;
}

// lightMatrix.comp:57-137, local_size 1x1x1: one thread fits the cascades to the depth range of the frame (HiZ apex)
__global__ void lightMatrixKernel(ShadowCascadeInfo* __restrict__ info, ImgView apex, const GlobalUbo* __restrict__ g, int count, float highestCascadeExtraPadding,
                                  float highestCascadeMinFarPlane) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    const float FLOAT_MAX = 3.402823466e+38f, FLOAT_MIN = 1.175494351e-38f;
    const float shadowSampleRadius = 0.03f; // sunShadowCascades.inc:5
    M4 corr{};
    corr.c[0][0] = 1.f; corr.c[1][1] = 1.f; corr.c[2][2] = -0.5f; corr.c[3][2] = 0.5f; corr.c[3][3] = 1.f;
    const vec3 forward = -ld3(g->sunDirection);
    vec3 up = fabsf(forward.y) < 0.9999f ? vec3(0.f, -1.f, 0.f) : vec3(0.f, 0.f, -1.f);
    const vec3 right = cross(forward, up);
    up = cross(right, forward);
    const vec3 nr = normalize(right), nu = normalize(up);
    M4 V{};
    V.c[0][0] = nr.x; V.c[1][0] = nr.y; V.c[2][0] = nr.z;
    V.c[0][1] = nu.x; V.c[1][1] = nu.y; V.c[2][1] = nu.z;
    V.c[0][2] = forward.x; V.c[1][2] = forward.y; V.c[2][2] = forward.z;
    V.c[3][3] = 1.f;
    const float2 depthMinMax = ((const float2*)apex.ptr)[0]; // imageLoad(depthMinMaxLowestMip, ivec2(0)).rg
    const float depthMaxLinear = linearizeDepth(depthMinMax.x, g->nearPlane, g->farPlane);
    const float depthMinLinear = linearizeDepth(depthMinMax.y, g->nearPlane, g->farPlane);
    for (int i = 0; i < count - 1; i++) info->splits[i] = depthMinLinear + ((depthMaxLinear - depthMinLinear) * (float)(i + 1) / (float)count);
    const vec3 camPos = ld3(g->cameraPosition), camFwd = ld3(g->cameraForward), camUp = ld3(g->cameraUp), camRight = ld3(g->cameraRight);
    for (int i = 0; i < count; i++) {
        vec3 minP(FLOAT_MAX), maxP(FLOAT_MIN); // sic (:88-89)
        float cascadeMinDepth = i > 0 ? info->splits[i - 1] : 0.f;
        float cascadeMaxDepth = i < 4 ? info->splits[i] : 0.f;
        if (i == 0) cascadeMinDepth = depthMinLinear;
        if (i == count - 1) {
            cascadeMinDepth = g->nearPlane;
            cascadeMaxDepth = gmax(depthMaxLinear, highestCascadeMinFarPlane);
        }
        vec3 pts[8];
        const vec3 nearC = camPos + camFwd * cascadeMinDepth, farC = camPos + camFwd * cascadeMaxDepth;
        const float hN = g->cameraTanFovHalf * cascadeMinDepth, hF = g->cameraTanFovHalf * cascadeMaxDepth;
        const float wN = hN * g->cameraAspectRatio, wF = hF * g->cameraAspectRatio;
        pts[0] = farC + camUp * hF + camRight * wF; pts[1] = farC + camUp * hF - camRight * wF;
        pts[2] = farC - camUp * hF + camRight * wF; pts[3] = farC - camUp * hF - camRight * wF;
        pts[4] = nearC + camUp * hN + camRight * wN; pts[5] = nearC + camUp * hN - camRight * wN;
        pts[6] = nearC - camUp * hN + camRight * wN; pts[7] = nearC - camUp * hN - camRight * wN;
        for (int k = 0; k < 8; k++) {
            const vec3 p = pts[k];
            const vec3 t(V.c[0][0] * p.x + V.c[1][0] * p.y + V.c[2][0] * p.z + V.c[3][0] * 1.f, V.c[0][1] * p.x + V.c[1][1] * p.y + V.c[2][1] * p.z + V.c[3][1] * 1.f,
                         V.c[0][2] * p.x + V.c[1][2] * p.y + V.c[2][2] * p.z + V.c[3][2] * 1.f);
            minP = vec3(gmin(minP.x, t.x), gmin(minP.y, t.y), gmin(minP.z, t.z));
            maxP = vec3(gmax(maxP.x, t.x), gmax(maxP.y, t.y), gmax(maxP.z, t.z));
        }
        if (i == count - 1) { minP = minP - highestCascadeExtraPadding; maxP = maxP + highestCascadeExtraPadding; }
        minP = minP - shadowSampleRadius * 2.f;
        maxP = maxP + shadowSampleRadius * 2.f;
        const vec3 d = maxP - minP;
        const vec3 scale(2.f / d.x, 2.f / d.y, 2.f / d.z);
        const vec3 s = maxP + minP;
        const vec3 offset(-0.5f * s.x * scale.x, -0.5f * s.y * scale.y, -0.5f * s.z * scale.z);
        M4 P{};
        P.c[0][0] = scale.x; P.c[1][1] = scale.y; P.c[2][2] = scale.z;
        P.c[3][0] = offset.x; P.c[3][1] = offset.y; P.c[3][2] = offset.z; P.c[3][3] = 1.f;
        const M4 L = mulM4(mulM4(corr, P), V);
        for (int col = 0; col < 4; col++)
            for (int row = 0; row < 4; row++) info->lightMatrices[i][col * 4 + row] = L.c[col][row];
        info->lightSpaceScale[i][0] = scale.x;
        info->lightSpaceScale[i][1] = scale.y;
    }
}

static int launchLightMatrix(const PassCtx& c) {
    if (int rc = c.needGlobal()) return rc;
    if (int rc = c.needSbuf(0, sizeof(ShadowCascadeInfo), "lightMatrix sunShadowInfo")) return rc;
    if (int rc = c.needStorage(1, F_RG32F, "lightMatrix depthMinMaxLowestMip")) return rc;
    if (c.sbuf[0].readOnly) return c.fail(-4, "lightMatrix: sunShadowInfo is bound read-only");
    if (c.push.size() < 8) return c.fail(-1, "lightMatrix: push constants (padding, min far plane) missing");
    const int count = c.specInt(0, 4);
    if (count < 1 || count > 4) return c.fail(-1, "lightMatrix: sunShadowCascadeCount must be 1..4");
    float pc[2];
    std::memcpy(pc, c.push.data(), 8);
    if (c.dispatch[0] == 0 || c.dispatch[1] == 0 || c.dispatch[2] == 0) return 0;
    lightMatrixKernel<<<1, 64, 0, c.stream>>>((ShadowCascadeInfo*)c.sbuf[0].ptr, c.storage[1], c.global, count, pc[0], pc[1]);
    PLR_CHECK_LAUNCH(c);
    return 0;
}
PLR_REGISTER_SHADER("lightMatrix.comp", launchLightMatrix);

} // namespace plr
