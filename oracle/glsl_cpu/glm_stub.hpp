// glm_stub.hpp -- the handful of glm types and functions the reference's HOST arithmetic uses, so that its own text can run here.
//
// TEST INFRASTRUCTURE ONLY (oracle/_ref).  glm is not in this container (the reference fetches it with CMake; Linux builds take the
// system package).  oracle/build_ref.py compiles, verbatim,
//     GSScene.cpp:17-24 + :38-54     struct VertexStorage and the body of GSScene::load's conversion loop
//     GSScene.h:41-46                struct Vertex
//     Renderer.h:21-29, :40-50       struct UniformBuffer, struct Camera (with Camera::translate)
//     Renderer.cpp:719-754           Renderer::updateUniforms, whole
// against this header: the ORDER OF CALLS, the constants, the sign flips and the double / float mix of that text are then the
// reference's own, executed.  What this header restates is what is inside the glm calls, following glm 0.9.9.8 (Ubuntu 22.04's
// libglm-dev: the reference's Linux build uses find_package(glm), CMakeLists.txt:43; its pinned 1.0.0 for Windows / Apple evaluates
// the same expressions in the same order for the scalar, non-SIMD configuration), file and function named at each definition.
// Scalar code paths only (GLM_FORCE_PURE semantics: the default for an unaligned build); column-major matrices, m[column][row];
// quaternion storage order x, y, z, w with the constructor taking (w, x, y, z).
#pragma once
#include <cmath>
#include <cstdint>
#include <cstring>

namespace glm {

struct vec3 {
    float x, y, z;
    vec3() : x(0), y(0), z(0) {}
    explicit vec3(float s) : x(s), y(s), z(s) {}
    vec3(float a, float b, float c) : x(a), y(b), z(c) {}
    float& operator[](int i) { return (&x)[i]; }
    const float& operator[](int i) const { return (&x)[i]; }
    vec3& operator+=(const vec3& o) { x += o.x; y += o.y; z += o.z; return *this; }
};
struct vec4 {
    float x, y, z, w;
    vec4() : x(0), y(0), z(0), w(0) {}
    explicit vec4(float s) : x(s), y(s), z(s), w(s) {}
    vec4(float a, float b, float c, float d) : x(a), y(b), z(c), w(d) {}
    vec4(const vec3& v, float d) : x(v.x), y(v.y), z(v.z), w(d) {}
    float& operator[](int i) { return (&x)[i]; }
    const float& operator[](int i) const { return (&x)[i]; }
};
inline vec3 operator+(const vec3& a, const vec3& b) { return vec3(a.x + b.x, a.y + b.y, a.z + b.z); }
inline vec3 operator*(const vec3& a, float s) { return vec3(a.x * s, a.y * s, a.z * s); }
inline vec4 operator+(const vec4& a, const vec4& b) { return vec4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }
inline vec4 operator-(const vec4& a, const vec4& b) { return vec4(a.x - b.x, a.y - b.y, a.z - b.z, a.w - b.w); }
inline vec4 operator*(const vec4& a, const vec4& b) { return vec4(a.x * b.x, a.y * b.y, a.z * b.z, a.w * b.w); }
inline vec4 operator*(const vec4& a, float s) { return vec4(a.x * s, a.y * s, a.z * s, a.w * s); }

// detail/func_exponential.inl: exp(vec) applies std::exp per component; inversesqrt(x) = 1 / sqrt(x)
inline vec3 exp(const vec3& v) { return vec3(std::exp(v.x), std::exp(v.y), std::exp(v.z)); }
inline float inversesqrt(float x) { return 1.0f / std::sqrt(x); }
// detail/func_geometric.inl: compute_dot<vec<4>>: tmp = a * b; (tmp.x + tmp.y) + (tmp.z + tmp.w);  compute_dot<vec<3>>: tmp.x + tmp.y + tmp.z
inline float dot(const vec4& a, const vec4& b) {
    const vec4 tmp(a * b);
    return (tmp.x + tmp.y) + (tmp.z + tmp.w);
}
// compute_normalize: v * inversesqrt(dot(v, v))
inline vec4 normalize(const vec4& v) { return v * inversesqrt(dot(v, v)); }
// compute_cross: (x.y * y.z - y.y * x.z, x.z * y.x - y.z * x.x, x.x * y.y - y.x * x.y)
inline vec3 cross(const vec3& x, const vec3& y) { return vec3(x.y * y.z - y.y * x.z, x.z * y.x - y.z * x.x, x.x * y.y - y.x * x.y); }
// detail/func_trigonometric.inl: radians(degrees) = degrees * 0.01745329251994329576923690768489 (narrowed to the argument's type)
inline float radians(float degrees) { return degrees * static_cast<float>(0.01745329251994329576923690768489); }

struct mat3 {
    vec3 c[3];
    mat3() {}
    explicit mat3(float d) { c[0] = vec3(d, 0, 0); c[1] = vec3(0, d, 0); c[2] = vec3(0, 0, d); }
    vec3& operator[](int i) { return c[i]; }
    const vec3& operator[](int i) const { return c[i]; }
};
struct mat4 {
    vec4 c[4];
    mat4() {}
    explicit mat4(float d) { c[0] = vec4(d, 0, 0, 0); c[1] = vec4(0, d, 0, 0); c[2] = vec4(0, 0, d, 0); c[3] = vec4(0, 0, 0, d); }
    mat4(const vec4& a, const vec4& b, const vec4& cc, const vec4& d) { c[0] = a; c[1] = b; c[2] = cc; c[3] = d; }
    // detail/type_mat4x4.inl, mat<4,4>(mat<3,3>): columns (m[i], 0), last column (0, 0, 0, 1)
    explicit mat4(const mat3& m) { c[0] = vec4(m[0], 0); c[1] = vec4(m[1], 0); c[2] = vec4(m[2], 0); c[3] = vec4(0, 0, 0, 1); }
    vec4& operator[](int i) { return c[i]; }
    const vec4& operator[](int i) const { return c[i]; }
};
static_assert(sizeof(vec4) == 16 && sizeof(mat4) == 64 && sizeof(vec3) == 12, "glm layouts");
// detail/type_mat4x4.inl, operator*(mat4, mat4): Result[j] = SrcA0 * SrcBj[0] + SrcA1 * SrcBj[1] + SrcA2 * SrcBj[2] + SrcA3 * SrcBj[3]
inline mat4 operator*(const mat4& m1, const mat4& m2) {
    mat4 r;
    for (int j = 0; j < 4; j++) r[j] = m1[0] * m2[j][0] + m1[1] * m2[j][1] + m1[2] * m2[j][2] + m1[3] * m2[j][3];
    return r;
}
inline mat4 operator*(const mat4& m, float s) { return mat4(m[0] * s, m[1] * s, m[2] * s, m[3] * s); }

struct quat {  // detail/type_quat.hpp: data x, y, z, w; qua(w, x, y, z)
    float x, y, z, w;
    quat() : x(0), y(0), z(0), w(1) {}
    quat(float w_, float x_, float y_, float z_) : x(x_), y(y_), z(z_), w(w_) {}
};
// detail/type_quat.inl, operator*(qua, vec3): QuatVector(q.x, q.y, q.z); uv = cross(QuatVector, v); uuv = cross(QuatVector, uv);
//                                             v + ((uv * q.w) + uuv) * 2
inline vec3 operator*(const quat& q, const vec3& v) {
    const vec3 QuatVector(q.x, q.y, q.z);
    const vec3 uv(cross(QuatVector, v));
    const vec3 uuv(cross(QuatVector, uv));
    return v + ((uv * q.w) + uuv) * 2.0f;
}
// gtc/quaternion.inl, mat3_cast
inline mat3 mat3_cast(const quat& q) {
    mat3 Result(1.0f);
    const float qxx(q.x * q.x), qyy(q.y * q.y), qzz(q.z * q.z);
    const float qxz(q.x * q.z), qxy(q.x * q.y), qyz(q.y * q.z);
    const float qwx(q.w * q.x), qwy(q.w * q.y), qwz(q.w * q.z);
    Result[0][0] = 1.0f - 2.0f * (qyy + qzz);
    Result[0][1] = 2.0f * (qxy + qwz);
    Result[0][2] = 2.0f * (qxz - qwy);
    Result[1][0] = 2.0f * (qxy - qwz);
    Result[1][1] = 1.0f - 2.0f * (qxx + qzz);
    Result[1][2] = 2.0f * (qyz + qwx);
    Result[2][0] = 2.0f * (qxz + qwy);
    Result[2][1] = 2.0f * (qyz - qwx);
    Result[2][2] = 1.0f - 2.0f * (qxx + qyy);
    return Result;
}
inline mat4 mat4_cast(const quat& q) { return mat4(mat3_cast(q)); }
// ext/matrix_transform.inl, translate(m, v): Result = m; Result[3] = m[0] * v[0] + m[1] * v[1] + m[2] * v[2] + m[3]
inline mat4 translate(const mat4& m, const vec3& v) {
    mat4 Result(m);
    Result[3] = m[0] * v[0] + m[1] * v[1] + m[2] * v[2] + m[3];
    return Result;
}
// detail/func_matrix.inl, compute_inverse<4, 4>
inline mat4 inverse(const mat4& m) {
    const float Coef00 = m[2][2] * m[3][3] - m[3][2] * m[2][3];
    const float Coef02 = m[1][2] * m[3][3] - m[3][2] * m[1][3];
    const float Coef03 = m[1][2] * m[2][3] - m[2][2] * m[1][3];
    const float Coef04 = m[2][1] * m[3][3] - m[3][1] * m[2][3];
    const float Coef06 = m[1][1] * m[3][3] - m[3][1] * m[1][3];
    const float Coef07 = m[1][1] * m[2][3] - m[2][1] * m[1][3];
    const float Coef08 = m[2][1] * m[3][2] - m[3][1] * m[2][2];
    const float Coef10 = m[1][1] * m[3][2] - m[3][1] * m[1][2];
    const float Coef11 = m[1][1] * m[2][2] - m[2][1] * m[1][2];
    const float Coef12 = m[2][0] * m[3][3] - m[3][0] * m[2][3];
    const float Coef14 = m[1][0] * m[3][3] - m[3][0] * m[1][3];
    const float Coef15 = m[1][0] * m[2][3] - m[2][0] * m[1][3];
    const float Coef16 = m[2][0] * m[3][2] - m[3][0] * m[2][2];
    const float Coef18 = m[1][0] * m[3][2] - m[3][0] * m[1][2];
    const float Coef19 = m[1][0] * m[2][2] - m[2][0] * m[1][2];
    const float Coef20 = m[2][0] * m[3][1] - m[3][0] * m[2][1];
    const float Coef22 = m[1][0] * m[3][1] - m[3][0] * m[1][1];
    const float Coef23 = m[1][0] * m[2][1] - m[2][0] * m[1][1];
    const vec4 Fac0(Coef00, Coef00, Coef02, Coef03);
    const vec4 Fac1(Coef04, Coef04, Coef06, Coef07);
    const vec4 Fac2(Coef08, Coef08, Coef10, Coef11);
    const vec4 Fac3(Coef12, Coef12, Coef14, Coef15);
    const vec4 Fac4(Coef16, Coef16, Coef18, Coef19);
    const vec4 Fac5(Coef20, Coef20, Coef22, Coef23);
    const vec4 Vec0(m[1][0], m[0][0], m[0][0], m[0][0]);
    const vec4 Vec1(m[1][1], m[0][1], m[0][1], m[0][1]);
    const vec4 Vec2(m[1][2], m[0][2], m[0][2], m[0][2]);
    const vec4 Vec3(m[1][3], m[0][3], m[0][3], m[0][3]);
    const vec4 Inv0(Vec1 * Fac0 - Vec2 * Fac1 + Vec3 * Fac2);
    const vec4 Inv1(Vec0 * Fac0 - Vec2 * Fac3 + Vec3 * Fac4);
    const vec4 Inv2(Vec0 * Fac1 - Vec1 * Fac3 + Vec3 * Fac5);
    const vec4 Inv3(Vec0 * Fac2 - Vec1 * Fac4 + Vec2 * Fac5);
    const vec4 SignA(+1, -1, +1, -1);
    const vec4 SignB(-1, +1, -1, +1);
    const mat4 Inverse(Inv0 * SignA, Inv1 * SignB, Inv2 * SignA, Inv3 * SignB);
    const vec4 Row0(Inverse[0][0], Inverse[1][0], Inverse[2][0], Inverse[3][0]);
    const vec4 Dot0(m[0] * Row0);
    const float Dot1 = (Dot0.x + Dot0.y) + (Dot0.z + Dot0.w);
    const float OneOverDeterminant = 1.0f / Dot1;
    return Inverse * OneOverDeterminant;
}
// ext/matrix_clip_space.inl, perspectiveRH_NO (glm::perspective without GLM_FORCE_LEFT_HANDED / GLM_FORCE_DEPTH_ZERO_TO_ONE: the
// reference defines neither, Renderer.h:4 only sets GLM_SWIZZLE)
inline mat4 perspective(float fovy, float aspect, float zNear, float zFar) {
    const float tanHalfFovy = std::tan(fovy / 2.0f);
    mat4 Result(0.0f);
    Result[0][0] = 1.0f / (aspect * tanHalfFovy);
    Result[1][1] = 1.0f / (tanHalfFovy);
    Result[2][2] = -(zFar + zNear) / (zFar - zNear);
    Result[2][3] = -1.0f;
    Result[3][2] = -(2.0f * zFar * zNear) / (zFar - zNear);
    return Result;
}

}  // namespace glm
