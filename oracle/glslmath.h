// ORACLE (test infrastructure, never shipped, never on the product path).
// Minimal GLSL-flavoured vector/matrix types so the CPU restatements can follow the
// reference shaders (resources/shaders/*.comp|.inc|.frag) statement by statement with
// the same operation order. All arithmetic is plain binary32, evaluated left to right,
// no FMA (build with -ffp-contract=off).
#pragma once
#include "detmath.h"

namespace orc {

struct vec2 { float x, y; vec2() : x(0), y(0) {} vec2(float a) : x(a), y(a) {} vec2(float a, float b) : x(a), y(b) {} };
struct vec3 {
    float x, y, z;
    vec3() : x(0), y(0), z(0) {}
    vec3(float a) : x(a), y(a), z(a) {}
    vec3(float a, float b, float c) : x(a), y(b), z(c) {}
    float& operator[](int i) { return (&x)[i]; }
    float operator[](int i) const { return (&x)[i]; }
};
struct vec4 {
    float x, y, z, w;
    vec4() : x(0), y(0), z(0), w(0) {}
    vec4(float a) : x(a), y(a), z(a), w(a) {}
    vec4(float a, float b, float c, float d) : x(a), y(b), z(c), w(d) {}
    vec4(const vec3& v, float d) : x(v.x), y(v.y), z(v.z), w(d) {}
    float& operator[](int i) { return (&x)[i]; }
    float operator[](int i) const { return (&x)[i]; }
    vec3 xyz() const { return vec3(x, y, z); }
};
struct ivec2 { int x, y; ivec2() : x(0), y(0) {} ivec2(int a) : x(a), y(a) {} ivec2(int a, int b) : x(a), y(b) {} };
struct ivec3 { int x, y, z; ivec3() : x(0), y(0), z(0) {} ivec3(int a, int b, int c) : x(a), y(b), z(c) {} };

static inline vec2 toVec2(ivec2 v) { return vec2((float)v.x, (float)v.y); }

#define ORC_VEC_OPS(V, ...) \
    static inline V operator+(V a, V b) { return V##_zip(a, b, [](float p, float q) { return p + q; }); } \
    static inline V operator-(V a, V b) { return V##_zip(a, b, [](float p, float q) { return p - q; }); } \
    static inline V operator*(V a, V b) { return V##_zip(a, b, [](float p, float q) { return p * q; }); } \
    static inline V operator/(V a, V b) { return V##_zip(a, b, [](float p, float q) { return p / q; }); } \
    static inline V operator+(V a, float b) { return a + V(b); } \
    static inline V operator-(V a, float b) { return a - V(b); } \
    static inline V operator*(V a, float b) { return a * V(b); } \
    static inline V operator/(V a, float b) { return a / V(b); } \
    static inline V operator+(float a, V b) { return V(a) + b; } \
    static inline V operator-(float a, V b) { return V(a) - b; } \
    static inline V operator*(float a, V b) { return V(a) * b; } \
    static inline V operator/(float a, V b) { return V(a) / b; } \
    static inline V& operator+=(V& a, V b) { a = a + b; return a; } \
    static inline V& operator-=(V& a, V b) { a = a - b; return a; } \
    static inline V& operator*=(V& a, V b) { a = a * b; return a; } \
    static inline V& operator/=(V& a, V b) { a = a / b; return a; } \
    static inline V& operator*=(V& a, float b) { a = a * b; return a; } \
    static inline V& operator/=(V& a, float b) { a = a / b; return a; } \
    static inline V& operator+=(V& a, float b) { a = a + b; return a; } \
    static inline V& operator-=(V& a, float b) { a = a - b; return a; } \
    static inline V min(V a, V b) { return V##_zip(a, b, gmin); } \
    static inline V max(V a, V b) { return V##_zip(a, b, gmax); } \
    static inline V clamp(V a, float lo, float hi) { return V##_zip(a, a, [lo, hi](float p, float) { return gclamp(p, lo, hi); }); } \
    static inline V clamp(V a, V lo, V hi) { return min(max(a, lo), hi); } \
    static inline V abs(V a) { return V##_zip(a, a, [](float p, float) { return std::fabs(p); }); } \
    static inline V mix(V a, V b, float t) { return a * (1.f - t) + b * t; }

template <class F> static inline vec2 vec2_zip(vec2 a, vec2 b, F f) { return vec2(f(a.x, b.x), f(a.y, b.y)); }
template <class F> static inline vec3 vec3_zip(vec3 a, vec3 b, F f) { return vec3(f(a.x, b.x), f(a.y, b.y), f(a.z, b.z)); }
template <class F> static inline vec4 vec4_zip(vec4 a, vec4 b, F f) { return vec4(f(a.x, b.x), f(a.y, b.y), f(a.z, b.z), f(a.w, b.w)); }
ORC_VEC_OPS(vec2)
ORC_VEC_OPS(vec3)
ORC_VEC_OPS(vec4)

static inline vec2 operator-(vec2 a) { return vec2(-a.x, -a.y); }
static inline vec3 operator-(vec3 a) { return vec3(-a.x, -a.y, -a.z); }
static inline vec4 operator-(vec4 a) { return vec4(-a.x, -a.y, -a.z, -a.w); }

static inline float dot(vec2 a, vec2 b) { return a.x * b.x + a.y * b.y; }
static inline float dot(vec3 a, vec3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
static inline float dot(vec4 a, vec4 b) { return a.x * b.x + a.y * b.y + a.z * b.z + a.w * b.w; }
static inline float length(vec2 a) { return std::sqrt(dot(a, a)); }
static inline float length(vec3 a) { return std::sqrt(dot(a, a)); }
static inline float length(vec4 a) { return std::sqrt(dot(a, a)); }
static inline float distance(vec3 a, vec3 b) { return length(a - b); }
// normalize(v) is defined as v * (1/sqrt(dot(v,v))) (one IEEE division, one IEEE sqrt)
static inline vec3 normalize(vec3 a) { const float inv = 1.0f / std::sqrt(dot(a, a)); return a * inv; }
static inline vec4 normalize(vec4 a) { const float inv = 1.0f / std::sqrt(dot(a, a)); return a * inv; }
static inline vec3 cross(vec3 a, vec3 b) {
    return vec3(a.y * b.z - b.y * a.z, a.z * b.x - b.z * a.x, a.x * b.y - b.x * a.y);
}
static inline vec3 reflect(vec3 I, vec3 N) { return I - 2.0f * dot(N, I) * N; }
static inline float clamp(float x, float lo, float hi) { return gclamp(x, lo, hi); }
static inline float mix(float a, float b, float t) { return gmix(a, b, t); }
static inline vec4 mix(vec4 a, vec4 b, vec4 t) { return a * (vec4(1.f) - t) + b * t; }
static inline vec3 mix(vec3 a, vec3 b, vec3 t) { return a * (vec3(1.f) - t) + b * t; }
static inline vec3 pow(vec3 a, vec3 b) { return vec3(det_powf(a.x, b.x), det_powf(a.y, b.y), det_powf(a.z, b.z)); }
static inline vec2 floor(vec2 a) { return vec2(std::floor(a.x), std::floor(a.y)); }
static inline bool isnan3(vec3 a) { return a.x != a.x || a.y != a.y || a.z != a.z; }
static inline bool isnan4(vec4 a) { return a.x != a.x || a.y != a.y || a.z != a.z || a.w != a.w; }
static inline bool isnan2(vec2 a) { return a.x != a.x || a.y != a.y; }

// column-major 4x4 like GLSL/glm: m.c[col][row]
struct mat4 {
    float c[4][4];
    vec4 col(int i) const { return vec4(c[i][0], c[i][1], c[i][2], c[i][3]); }
};
// GLSL: M * v = sum_i column_i * v[i]; component r = m[0][r]*v.x + m[1][r]*v.y + m[2][r]*v.z + m[3][r]*v.w
static inline vec4 operator*(const mat4& m, vec4 v) {
    vec4 r;
    for (int i = 0; i < 4; i++) r[i] = m.c[0][i] * v.x + m.c[1][i] * v.y + m.c[2][i] * v.z + m.c[3][i] * v.w;
    return r;
}

} // namespace orc
