// Minimal stand-in for the parts of glm the reference's kernels use (vec2/vec3/vec4/mat3, column-major,
// products summed k = 0,1,2 like glm's own operator*).  Written for oracle/ref_build only — test infrastructure.
#pragma once
#include <hip/hip_runtime.h>
#include <cmath>
#define GLM_FN __host__ __device__ inline
namespace glm {
struct vec2 {
    float x, y;
    GLM_FN vec2() : x(0), y(0) {}
    template <typename A, typename B> GLM_FN vec2(A a, B b) : x((float)a), y((float)b) {}
};
struct vec3 {
    float x, y, z;
    GLM_FN vec3() : x(0), y(0), z(0) {}
    GLM_FN explicit vec3(float s) : x(s), y(s), z(s) {}
    template <typename A, typename B, typename C> GLM_FN vec3(A a, B b, C c) : x((float)a), y((float)b), z((float)c) {}
    GLM_FN float& operator[](int i) { return (&x)[i]; }
    GLM_FN const float& operator[](int i) const { return (&x)[i]; }
    GLM_FN vec3& operator+=(const vec3& o) { x += o.x; y += o.y; z += o.z; return *this; }
    GLM_FN vec3& operator+=(float s) { x += s; y += s; z += s; return *this; }
    GLM_FN vec3& operator*=(float s) { x *= s; y *= s; z *= s; return *this; }
};
struct vec4 {
    float x, y, z, w;
    GLM_FN vec4() : x(0), y(0), z(0), w(0) {}
    template <typename A, typename B, typename C, typename D> GLM_FN vec4(A a, B b, C c, D d) : x((float)a), y((float)b), z((float)c), w((float)d) {}
};
GLM_FN vec3 operator+(const vec3& a, const vec3& b) { return vec3(a.x + b.x, a.y + b.y, a.z + b.z); }
GLM_FN vec3 operator-(const vec3& a, const vec3& b) { return vec3(a.x - b.x, a.y - b.y, a.z - b.z); }
GLM_FN vec3 operator-(const vec3& a) { return vec3(-a.x, -a.y, -a.z); }
GLM_FN vec3 operator*(float s, const vec3& a) { return vec3(s * a.x, s * a.y, s * a.z); }
GLM_FN vec3 operator*(const vec3& a, float s) { return vec3(a.x * s, a.y * s, a.z * s); }
GLM_FN vec3 operator/(const vec3& a, float s) { return vec3(a.x / s, a.y / s, a.z / s); }
GLM_FN float dot(const vec3& a, const vec3& b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
GLM_FN float length(const vec3& a) { return sqrtf(dot(a, a)); }
GLM_FN vec3 max(const vec3& a, float s) { return vec3(a.x > s ? a.x : s, a.y > s ? a.y : s, a.z > s ? a.z : s); }
struct mat3 {
    vec3 c[3];  // columns
    GLM_FN mat3() { c[0] = vec3(1, 0, 0); c[1] = vec3(0, 1, 0); c[2] = vec3(0, 0, 1); }
    GLM_FN explicit mat3(float d) { c[0] = vec3(d, 0, 0); c[1] = vec3(0, d, 0); c[2] = vec3(0, 0, d); }
    template <typename A0, typename A1, typename A2, typename A3, typename A4, typename A5, typename A6, typename A7, typename A8>
    GLM_FN mat3(A0 a0, A1 a1, A2 a2, A3 a3, A4 a4, A5 a5, A6 a6, A7 a7, A8 a8)
    {
        c[0] = vec3(a0, a1, a2); c[1] = vec3(a3, a4, a5); c[2] = vec3(a6, a7, a8);
    }
    GLM_FN vec3& operator[](int i) { return c[i]; }
    GLM_FN const vec3& operator[](int i) const { return c[i]; }
};
GLM_FN mat3 operator*(const mat3& a, const mat3& b)
{
    mat3 r(0.0f);
    for (int col = 0; col < 3; col++)
        for (int row = 0; row < 3; row++) r[col][row] = a[0][row] * b[col][0] + a[1][row] * b[col][1] + a[2][row] * b[col][2];
    return r;
}
GLM_FN mat3 operator*(float s, const mat3& a)
{
    mat3 r(0.0f);
    for (int col = 0; col < 3; col++) r[col] = s * a[col];
    return r;
}
GLM_FN mat3 transpose(const mat3& a)
{
    mat3 r(0.0f);
    for (int col = 0; col < 3; col++)
        for (int row = 0; row < 3; row++) r[col][row] = a[row][col];
    return r;
}
}  // namespace glm
