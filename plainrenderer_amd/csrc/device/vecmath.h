// Small vector types for the HIP kernels. Arithmetic is component-wise binary32, evaluated in
// source order; the library is built with -ffp-contract=off so a*b+c is two roundings.
#pragma once
#include "detmath.h"

namespace plr {

struct vec2 {
    float x, y;
    PLR_DI vec2() : x(0.f), y(0.f) {}
    PLR_DI explicit vec2(float a) : x(a), y(a) {}
    PLR_DI vec2(float a, float b) : x(a), y(b) {}
};
struct vec3 {
    float x, y, z;
    PLR_DI vec3() : x(0.f), y(0.f), z(0.f) {}
    PLR_DI explicit vec3(float a) : x(a), y(a), z(a) {}
    PLR_DI vec3(float a, float b, float c) : x(a), y(b), z(c) {}
};
struct vec4 {
    float x, y, z, w;
    PLR_DI vec4() : x(0.f), y(0.f), z(0.f), w(0.f) {}
    PLR_DI explicit vec4(float a) : x(a), y(a), z(a), w(a) {}
    PLR_DI vec4(float a, float b, float c, float d) : x(a), y(b), z(c), w(d) {}
    PLR_DI vec4(vec3 v, float d) : x(v.x), y(v.y), z(v.z), w(d) {}
    PLR_DI vec3 xyz() const { return vec3(x, y, z); }
};
struct ivec2 {
    int x, y;
    PLR_DI ivec2() : x(0), y(0) {}
    PLR_DI ivec2(int a, int b) : x(a), y(b) {}
};

#define PLR_V2(OP) \
    PLR_DI vec2 operator OP(vec2 a, vec2 b) { return vec2(a.x OP b.x, a.y OP b.y); } \
    PLR_DI vec2 operator OP(vec2 a, float b) { return vec2(a.x OP b, a.y OP b); } \
    PLR_DI vec2 operator OP(float a, vec2 b) { return vec2(a OP b.x, a OP b.y); }
#define PLR_V3(OP) \
    PLR_DI vec3 operator OP(vec3 a, vec3 b) { return vec3(a.x OP b.x, a.y OP b.y, a.z OP b.z); } \
    PLR_DI vec3 operator OP(vec3 a, float b) { return vec3(a.x OP b, a.y OP b, a.z OP b); } \
    PLR_DI vec3 operator OP(float a, vec3 b) { return vec3(a OP b.x, a OP b.y, a OP b.z); }
#define PLR_V4(OP) \
    PLR_DI vec4 operator OP(vec4 a, vec4 b) { return vec4(a.x OP b.x, a.y OP b.y, a.z OP b.z, a.w OP b.w); } \
    PLR_DI vec4 operator OP(vec4 a, float b) { return vec4(a.x OP b, a.y OP b, a.z OP b, a.w OP b); } \
    PLR_DI vec4 operator OP(float a, vec4 b) { return vec4(a OP b.x, a OP b.y, a OP b.z, a OP b.w); }
PLR_V2(+) PLR_V2(-) PLR_V2(*) PLR_V2(/)
PLR_V3(+) PLR_V3(-) PLR_V3(*) PLR_V3(/)
PLR_V4(+) PLR_V4(-) PLR_V4(*) PLR_V4(/)

PLR_DI vec2 operator-(vec2 a) { return vec2(-a.x, -a.y); }
PLR_DI vec3 operator-(vec3 a) { return vec3(-a.x, -a.y, -a.z); }
PLR_DI vec3& operator+=(vec3& a, vec3 b) { a = a + b; return a; }
PLR_DI vec3& operator-=(vec3& a, vec3 b) { a = a - b; return a; }
PLR_DI vec3& operator*=(vec3& a, float b) { a = a * b; return a; }
PLR_DI vec3& operator/=(vec3& a, float b) { a = a / b; return a; }
PLR_DI vec4& operator+=(vec4& a, vec4 b) { a = a + b; return a; }
PLR_DI vec2& operator+=(vec2& a, vec2 b) { a = a + b; return a; }

PLR_DI float dot(vec2 a, vec2 b) { return a.x * b.x + a.y * b.y; }
PLR_DI float dot(vec3 a, vec3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
PLR_DI float dot(vec4 a, vec4 b) { return a.x * b.x + a.y * b.y + a.z * b.z + a.w * b.w; }
PLR_DI float length(vec2 a) { return sqrtf(dot(a, a)); }
PLR_DI float length(vec3 a) { return sqrtf(dot(a, a)); }
PLR_DI float length(vec4 a) { return sqrtf(dot(a, a)); }
PLR_DI float distance(vec3 a, vec3 b) { return length(a - b); }
// normalize(v) := v * (1/sqrt(dot(v,v)))
PLR_DI vec3 normalize(vec3 a) { const float inv = 1.0f / sqrtf(dot(a, a)); return a * inv; }
PLR_DI vec4 normalize(vec4 a) { const float inv = 1.0f / sqrtf(dot(a, a)); return a * inv; }
PLR_DI vec3 cross(vec3 a, vec3 b) { return vec3(a.y * b.z - b.y * a.z, a.z * b.x - b.z * a.x, a.x * b.y - b.x * a.y); }
PLR_DI vec3 reflect(vec3 I, vec3 N) { return I - 2.0f * dot(N, I) * N; }

PLR_DI vec3 vmin(vec3 a, vec3 b) { return vec3(gmin(a.x, b.x), gmin(a.y, b.y), gmin(a.z, b.z)); }
PLR_DI vec3 vmax(vec3 a, vec3 b) { return vec3(gmax(a.x, b.x), gmax(a.y, b.y), gmax(a.z, b.z)); }
PLR_DI vec3 vclamp(vec3 a, float lo, float hi) { return vec3(gclamp(a.x, lo, hi), gclamp(a.y, lo, hi), gclamp(a.z, lo, hi)); }
PLR_DI vec3 vclamp(vec3 a, vec3 lo, vec3 hi) { return vmin(vmax(a, lo), hi); }
PLR_DI vec3 vabs(vec3 a) { return vec3(fabsf(a.x), fabsf(a.y), fabsf(a.z)); }
PLR_DI vec3 vmix(vec3 a, vec3 b, float t) { return a * (1.f - t) + b * t; }
PLR_DI vec4 vmix(vec4 a, vec4 b, float t) { return a * (1.f - t) + b * t; }
PLR_DI vec2 vmix(vec2 a, vec2 b, float t) { return a * (1.f - t) + b * t; }
PLR_DI vec3 vmix(vec3 a, vec3 b, vec3 t) { return a * (vec3(1.f) - t) + b * t; }
PLR_DI vec3 vpow(vec3 a, float e) { return vec3(det_powf(a.x, e), det_powf(a.y, e), det_powf(a.z, e)); }
PLR_DI bool anyNan(vec3 a) { return a.x != a.x || a.y != a.y || a.z != a.z; }
PLR_DI bool anyNan(vec4 a) { return a.x != a.x || a.y != a.y || a.z != a.z || a.w != a.w; }
PLR_DI bool anyNan(vec2 a) { return a.x != a.x || a.y != a.y; }

// column-major mat4 as 16 floats: element (col c, row r) = m[c*4+r]
PLR_DI vec4 mulMat4(const float* m, vec4 v) {
    return vec4(m[0] * v.x + m[4] * v.y + m[8] * v.z + m[12] * v.w, m[1] * v.x + m[5] * v.y + m[9] * v.z + m[13] * v.w,
                m[2] * v.x + m[6] * v.y + m[10] * v.z + m[14] * v.w, m[3] * v.x + m[7] * v.y + m[11] * v.z + m[15] * v.w);
}

} // namespace plr
