// TEST INFRASTRUCTURE ONLY (see oracle.h). CPU restatement of the input-producing compute passes.
// PARITY UNPINNED by the reference (no tests / golden data); lightMatrix is cross-checked against the independent host-side
// cascade fit of plainrenderer_amd/synth.py in tests/test_producers.py.
#include <cmath>
#include <cstring>

#include "common.h"
#include "oracle.h"

using namespace orc;

namespace {
struct M4 { float c[4][4]; }; // column major: c[col][row]
// GLSL matrix product, term order k = 0..3 (the kernel uses the same order)
M4 mul(const M4& a, const M4& b) {
    M4 r;
    for (int col = 0; col < 4; col++)
        for (int row = 0; row < 4; row++) r.c[col][row] = a.c[0][row] * b.c[col][0] + a.c[1][row] * b.c[col][1] + a.c[2][row] * b.c[col][2] + a.c[3][row] * b.c[col][3];
    return r;
}
} // namespace

// resources/shaders/lightMatrix.comp:57-137 (one invocation). apexMinMax = texel (0,0) of the lowest HiZ mip: .x = min depth
// (farthest, reverse Z), .y = max depth.
extern "C" void orc_light_matrix(orc_shadow_cascade_info* info, const float* apexMinMax, const orc_global* g, uint32_t sunShadowCascadeCount,
                                 float highestCascadeExtraPadding, float highestCascadeMinFarPlane) {
    const float FLOAT_MAX = 3.402823466e+38f, FLOAT_MIN = 1.175494351e-38f;
    const float shadowSampleRadius = 0.03f; // sunShadowCascades.inc:5
    M4 corr{};
    corr.c[0][0] = 1.f; corr.c[1][1] = 1.f; corr.c[2][2] = -0.5f; corr.c[3][2] = 0.5f; corr.c[3][3] = 1.f; // :59-63 (initialiser lists are columns)
    const vec3 forward = -vec3(g->sunDirection[0], g->sunDirection[1], g->sunDirection[2]);
    vec3 up = std::fabs(forward.y) < 0.9999f ? vec3(0.f, -1.f, 0.f) : vec3(0.f, 0.f, -1.f);
    const vec3 right = cross(forward, up);
    up = cross(right, forward);
    const vec3 nr = normalize(right), nu = normalize(up);
    // V[0].xyz = right, V[1].xyz = up, V[2].xyz = forward, then transposed: rows of V are the basis vectors
    M4 V{};
    V.c[0][0] = nr.x; V.c[1][0] = nr.y; V.c[2][0] = nr.z;
    V.c[0][1] = nu.x; V.c[1][1] = nu.y; V.c[2][1] = nu.z;
    V.c[0][2] = forward.x; V.c[1][2] = forward.y; V.c[2][2] = forward.z;
    V.c[3][3] = 1.f;

    const float depthMaxLinear = linearizeDepth(apexMinMax[0], g->nearPlane, g->farPlane);
    const float depthMinLinear = linearizeDepth(apexMinMax[1], g->nearPlane, g->farPlane);
    const int count = (int)sunShadowCascadeCount;
    for (int i = 0; i < count - 1; i++) info->splits[i] = depthMinLinear + ((depthMaxLinear - depthMinLinear) * (float)(i + 1) / (float)count);

    const vec3 camPos(g->cameraPosition[0], g->cameraPosition[1], g->cameraPosition[2]), camFwd(g->cameraForward[0], g->cameraForward[1], g->cameraForward[2]);
    const vec3 camUp(g->cameraUp[0], g->cameraUp[1], g->cameraUp[2]), camRight(g->cameraRight[0], g->cameraRight[1], g->cameraRight[2]);
    for (int i = 0; i < count; i++) {
        vec3 minP(FLOAT_MAX), maxP(FLOAT_MIN); // sic: FLOAT_MIN is the smallest positive float (:88-89)
        float cascadeMinDepth = i > 0 ? info->splits[i - 1] : 0.f; // the shader reads splits[-1] for i == 0 and overwrites it below
        float cascadeMaxDepth = i < 4 ? info->splits[i] : 0.f;
        if (i == 0) cascadeMinDepth = depthMinLinear;
        if (i == count - 1) {
            cascadeMinDepth = g->nearPlane;
            cascadeMaxDepth = gmax(depthMaxLinear, highestCascadeMinFarPlane);
        }
        // computeFrustumPoints (:29-49)
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
        const M4 L = mul(mul(corr, P), V);
        std::memcpy(info->lightMatrices[i], L.c, 64);
        info->lightSpaceScale[i][0] = scale.x;
        info->lightSpaceScale[i][1] = scale.y;
    }
}
