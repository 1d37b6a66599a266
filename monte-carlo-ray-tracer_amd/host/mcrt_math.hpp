// mcrt_math.hpp - the few vector types the host scene API needs.
//
// The reference uses glm (vec2/vec3/dvec3, include/glm).  Scene construction
// must reproduce its vertices bit for bit (object geometry feeds the hit
// parity tests), so the operations below keep glm 0.9.9.2's operation order:
//   dot       (x*x' + y*y') + z*z'        include/glm/detail/func_geometric.inl:48-55
//   cross                                  :68-79
//   normalize v * (1 / sqrt(dot(v,v)))     :82-90, func_exponential.inl:44-50
// Build with -ffp-contract=off.
#pragma once
#include <cfloat>
#include <cmath>
#include <cstdint>

#ifndef M_TWO_PI
#define M_TWO_PI 6.28318530717958647692528676655900576
#endif
#define FLOAT_INFINITY FLT_MAX                          // source/core/math.h:32
#define INTERSECTION_ERROR_MARGIN (FLT_EPSILON * 20.0f) // source/core/math.h:34

struct vec2 {
    float x = 0.0f, y = 0.0f;
    vec2() = default;
    vec2(float a, float b) : x(a), y(b) {}
};

struct vec3 {
    float x = 0.0f, y = 0.0f, z = 0.0f;
    vec3() = default;
    explicit vec3(float s) : x(s), y(s), z(s) {}
    template <class A, class B, class C> vec3(A a, B b, C c) : x(static_cast<float>(a)), y(static_cast<float>(b)), z(static_cast<float>(c)) {}
    vec3& operator+=(const vec3& o) { x += o.x; y += o.y; z += o.z; return *this; }
    vec3& operator*=(float s) { x *= s; y *= s; z *= s; return *this; }
};
inline vec3 operator+(const vec3& a, const vec3& b) { return vec3(a.x + b.x, a.y + b.y, a.z + b.z); }
inline vec3 operator-(const vec3& a, const vec3& b) { return vec3(a.x - b.x, a.y - b.y, a.z - b.z); }
inline vec3 operator*(const vec3& a, float s) { return vec3(a.x * s, a.y * s, a.z * s); }
inline vec3 operator/(const vec3& a, float s) { return vec3(a.x / s, a.y / s, a.z / s); }
inline float dot(const vec3& a, const vec3& b)
{
    const float tx = a.x * b.x, ty = a.y * b.y, tz = a.z * b.z;
    return tx + ty + tz;
}
inline vec3 cross(const vec3& p, const vec3& q) { return vec3(p.y * q.z - q.y * p.z, p.z * q.x - q.z * p.x, p.x * q.y - q.x * p.y); }
inline vec3 normalize(const vec3& v)
{
    const float inv = 1.0f / std::sqrt(dot(v, v));
    return v * inv;
}

// ColorDbl = glm::dvec3, source/core/math.h:41
struct ColorDbl {
    double r = 0.0, g = 0.0, b = 0.0;
    ColorDbl() = default;
    explicit ColorDbl(double s) : r(s), g(s), b(s) {}
    ColorDbl(double x, double y, double z) : r(x), g(y), b(z) {}
};
inline ColorDbl operator+(const ColorDbl& a, const ColorDbl& c) { return ColorDbl(a.r + c.r, a.g + c.g, a.b + c.b); }
inline ColorDbl operator*(const ColorDbl& a, double s) { return ColorDbl(a.r * s, a.g * s, a.b * s); }
inline ColorDbl operator/(const ColorDbl& a, double s) { return ColorDbl(a.r / s, a.g / s, a.b / s); }
