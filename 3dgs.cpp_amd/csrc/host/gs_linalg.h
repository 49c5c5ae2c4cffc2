// gs_linalg.h -- the few glm operations the camera controls use (Renderer.cpp:33-83, Renderer.h:47-49),
// restated from glm 1.0.0's published formulas: quat*quat, quat*vec3, angleAxis/rotate, normalize.
#pragma once
#include <cmath>

namespace gs {
struct vec3 {
    float x = 0, y = 0, z = 0;
};
struct quat {  // w first, like glm::quat's constructor
    float w = 1, x = 0, y = 0, z = 0;
};
inline vec3 operator+(vec3 a, vec3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
inline vec3 operator*(vec3 a, float s) { return {a.x * s, a.y * s, a.z * s}; }
inline vec3 cross(vec3 a, vec3 b) { return {a.y * b.z - b.y * a.z, a.z * b.x - b.z * a.x, a.x * b.y - b.x * a.y}; }
inline float length(vec3 a) { return std::sqrt(a.x * a.x + a.y * a.y + a.z * a.z); }
inline vec3 normalize(vec3 a) {
    const float inv = 1.0f / length(a);
    return a * inv;
}
// glm operator*(quat, quat)
inline quat operator*(quat p, quat q) {
    return {p.w * q.w - p.x * q.x - p.y * q.y - p.z * q.z, p.w * q.x + p.x * q.w + p.y * q.z - p.z * q.y,
            p.w * q.y + p.y * q.w + p.z * q.x - p.x * q.z, p.w * q.z + p.z * q.w + p.x * q.y - p.y * q.x};
}
// glm operator*(quat, vec3): v + 2 * (cross(qv, v) * w + cross(qv, cross(qv, v)))
inline vec3 operator*(quat q, vec3 v) {
    const vec3 qv{q.x, q.y, q.z};
    const vec3 uv = cross(qv, v);
    const vec3 uuv = cross(qv, uv);
    return v + (uv * q.w + uuv) * 2.0f;
}
// glm::rotate(quat, angle, axis): q * angleAxis(angle, normalize(axis))
inline quat rotate(quat q, float angle, vec3 axis) {
    const vec3 a = normalize(axis);
    const float s = std::sin(angle * 0.5f);
    return q * quat{std::cos(angle * 0.5f), a.x * s, a.y * s, a.z * s};
}
}  // namespace gs
