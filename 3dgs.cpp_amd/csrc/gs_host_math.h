// gs_host_math.h -- host-side arithmetic of the hot path's callers: PLY record activation
// (GSScene::load, src/GSScene.cpp:36-59) and the camera uniform block
// (Renderer::updateUniforms, src/Renderer.cpp:719-754).  fp32, one rounding per operation,
// glm 1.0.0's operation order (the reference pins glm at CMakeLists.txt:31-35; glm itself is
// not vendored, so its published formulas are restated here).  Column-major 4x4: m[c*4+r].
#pragma once
#include <cmath>
#include <cstdint>
#include <cstring>

#include "../../include/gs3d_hip.h"

namespace gs {
namespace host {

struct Mat4 {
    float m[16];
    float& at(int c, int r) { return m[c * 4 + r]; }
    float at(int c, int r) const { return m[c * 4 + r]; }
    static Mat4 identity() {
        Mat4 o{};
        o.m[0] = o.m[5] = o.m[10] = o.m[15] = 1.0f;
        return o;
    }
};

// glm operator*(mat4, mat4): column c of the result = A.col0*B[c][0] + A.col1*B[c][1] + ...
inline Mat4 mul(const Mat4& a, const Mat4& b) {
    Mat4 o{};
    for (int c = 0; c < 4; ++c)
        for (int r = 0; r < 4; ++r) {
            float s = a.at(0, r) * b.at(c, 0);
            s = s + a.at(1, r) * b.at(c, 1);
            s = s + a.at(2, r) * b.at(c, 2);
            s = s + a.at(3, r) * b.at(c, 3);
            o.at(c, r) = s;
        }
    return o;
}

// glm::mat4_cast(quat{w,x,y,z})
inline Mat4 from_quat(const float q[4]) {
    const float w = q[0], x = q[1], y = q[2], z = q[3];
    const float qxx = x * x, qyy = y * y, qzz = z * z;
    const float qxz = x * z, qxy = x * y, qyz = y * z;
    const float qwx = w * x, qwy = w * y, qwz = w * z;
    Mat4 o = Mat4::identity();
    o.at(0, 0) = 1.0f - 2.0f * (qyy + qzz);
    o.at(0, 1) = 2.0f * (qxy + qwz);
    o.at(0, 2) = 2.0f * (qxz - qwy);
    o.at(1, 0) = 2.0f * (qxy - qwz);
    o.at(1, 1) = 1.0f - 2.0f * (qxx + qzz);
    o.at(1, 2) = 2.0f * (qyz + qwx);
    o.at(2, 0) = 2.0f * (qxz + qwy);
    o.at(2, 1) = 2.0f * (qyz - qwx);
    o.at(2, 2) = 1.0f - 2.0f * (qxx + qyy);
    return o;
}

// glm::inverse(mat4): cofactor expansion, glm/detail/func_matrix.inl compute_inverse<4,4>.
inline Mat4 inverse(const Mat4& m) {
    auto M = [&](int c, int r) { return m.at(c, r); };
    const float c00 = M(2, 2) * M(3, 3) - M(3, 2) * M(2, 3);
    const float c02 = M(1, 2) * M(3, 3) - M(3, 2) * M(1, 3);
    const float c03 = M(1, 2) * M(2, 3) - M(2, 2) * M(1, 3);
    const float c04 = M(2, 1) * M(3, 3) - M(3, 1) * M(2, 3);
    const float c06 = M(1, 1) * M(3, 3) - M(3, 1) * M(1, 3);
    const float c07 = M(1, 1) * M(2, 3) - M(2, 1) * M(1, 3);
    const float c08 = M(2, 1) * M(3, 2) - M(3, 1) * M(2, 2);
    const float c10 = M(1, 1) * M(3, 2) - M(3, 1) * M(1, 2);
    const float c11 = M(1, 1) * M(2, 2) - M(2, 1) * M(1, 2);
    const float c12 = M(2, 0) * M(3, 3) - M(3, 0) * M(2, 3);
    const float c14 = M(1, 0) * M(3, 3) - M(3, 0) * M(1, 3);
    const float c15 = M(1, 0) * M(2, 3) - M(2, 0) * M(1, 3);
    const float c16 = M(2, 0) * M(3, 2) - M(3, 0) * M(2, 2);
    const float c18 = M(1, 0) * M(3, 2) - M(3, 0) * M(1, 2);
    const float c19 = M(1, 0) * M(2, 2) - M(2, 0) * M(1, 2);
    const float c20 = M(2, 0) * M(3, 1) - M(3, 0) * M(2, 1);
    const float c22 = M(1, 0) * M(3, 1) - M(3, 0) * M(1, 1);
    const float c23 = M(1, 0) * M(2, 1) - M(2, 0) * M(1, 1);
    const float fac[6][4] = {{c00, c00, c02, c03}, {c04, c04, c06, c07}, {c08, c08, c10, c11},
                             {c12, c12, c14, c15}, {c16, c16, c18, c19}, {c20, c20, c22, c23}};
    const float vec[4][4] = {{M(1, 0), M(0, 0), M(0, 0), M(0, 0)},
                             {M(1, 1), M(0, 1), M(0, 1), M(0, 1)},
                             {M(1, 2), M(0, 2), M(0, 2), M(0, 2)},
                             {M(1, 3), M(0, 3), M(0, 3), M(0, 3)}};
    Mat4 inv{};
    for (int i = 0; i < 4; ++i) {
        const float sign_a = (i & 1) ? -1.0f : 1.0f, sign_b = -sign_a;
        inv.at(0, i) = ((vec[1][i] * fac[0][i] - vec[2][i] * fac[1][i]) + vec[3][i] * fac[2][i]) * sign_a;
        inv.at(1, i) = ((vec[0][i] * fac[0][i] - vec[2][i] * fac[3][i]) + vec[3][i] * fac[4][i]) * sign_b;
        inv.at(2, i) = ((vec[0][i] * fac[1][i] - vec[1][i] * fac[3][i]) + vec[3][i] * fac[5][i]) * sign_a;
        inv.at(3, i) = ((vec[0][i] * fac[2][i] - vec[1][i] * fac[4][i]) + vec[2][i] * fac[5][i]) * sign_b;
    }
    const float d0 = M(0, 0) * inv.at(0, 0), d1 = M(0, 1) * inv.at(1, 0);
    const float d2 = M(0, 2) * inv.at(2, 0), d3 = M(0, 3) * inv.at(3, 0);
    const float one_over_det = 1.0f / ((d0 + d1) + (d2 + d3));
    for (float& v : inv.m) v = v * one_over_det;
    return inv;
}

// Renderer::updateUniforms, src/Renderer.cpp:719-754.
inline void camera_uniforms(const gs_camera& cam, uint32_t width, uint32_t height, gs_uniforms* out) {
    std::memset(out, 0, sizeof *out);
    out->width = width;
    out->height = height;
    out->camera_position[0] = cam.position[0];
    out->camera_position[1] = cam.position[1];
    out->camera_position[2] = cam.position[2];
    out->camera_position[3] = 1.0f;

    const Mat4 rotation = from_quat(cam.rotation);
    Mat4 translation = Mat4::identity();  // glm::translate(mat4(1), position)
    translation.at(3, 0) = cam.position[0];
    translation.at(3, 1) = cam.position[1];
    translation.at(3, 2) = cam.position[2];
    const Mat4 view = inverse(mul(translation, rotation));

    // :730 std::tan(glm::radians(fov) / 2.0) is evaluated in double and narrowed
    const float radians = cam.fov * 0.01745329251994329576923690768489f;
    const float tan_fovx = static_cast<float>(std::tan(static_cast<double>(radians) / 2.0));
    const float tan_fovy = tan_fovx * static_cast<float>(height) / static_cast<float>(width);

    // glm::perspectiveRH_NO(atan(tan_fovy) * 2, w / h, near, far)
    const float fovy = std::atan(tan_fovy) * 2.0f;
    const float aspect = static_cast<float>(width) / static_cast<float>(height);
    const float tan_half = std::tan(fovy / 2.0f);
    Mat4 persp{};
    persp.at(0, 0) = 1.0f / (aspect * tan_half);
    persp.at(1, 1) = 1.0f / tan_half;
    persp.at(2, 2) = -(cam.far_plane + cam.near_plane) / (cam.far_plane - cam.near_plane);
    persp.at(2, 3) = -1.0f;
    persp.at(3, 2) = -(2.0f * cam.far_plane * cam.near_plane) / (cam.far_plane - cam.near_plane);
    Mat4 proj = mul(persp, view);
    Mat4 v = view;
    for (int c = 0; c < 4; ++c) {  // :738-750 shader space is x right, y down, z forward
        v.at(c, 1) *= -1.0f;
        v.at(c, 2) *= -1.0f;
        proj.at(c, 1) *= -1.0f;
    }
    std::memcpy(out->proj_mat, proj.m, sizeof proj.m);
    std::memcpy(out->view_mat, v.m, sizeof v.m);
    out->tan_fovx = tan_fovx;
    out->tan_fovy = tan_fovy;
}

constexpr int kRecordFloats = 62;  // VertexStorage, GSScene.cpp:17-24
constexpr int kVertexFloats = 60;  // GSScene::Vertex, GSScene.h:41-46

// One PLY record -> one activated vertex (position4, scale_opacity4, rotation4, shs48).
inline void activate_record(const float* r, float* v) {
    const float* shs = r + 6;
    const float opacity = r[54];
    const float* scale = r + 55;
    const float* rot = r + 58;
    v[0] = r[0];
    v[1] = r[1];
    v[2] = r[2];
    v[3] = 1.0f;
    v[4] = std::exp(scale[0]);  // glm::exp(vec3) is std::exp per component
    v[5] = std::exp(scale[1]);
    v[6] = std::exp(scale[2]);
    v[7] = 1.0f / (1.0f + std::exp(-opacity));
    // glm::normalize(vec4): v * inversesqrt(dot(v, v)), dot<4> = (x*x + y*y) + (z*z + w*w)
    const float dot = (rot[0] * rot[0] + rot[1] * rot[1]) + (rot[2] * rot[2] + rot[3] * rot[3]);
    const float inv = 1.0f / std::sqrt(dot);
    for (int k = 0; k < 4; ++k) v[8 + k] = rot[k] * inv;
    float* out = v + 12;
    out[0] = shs[0];
    out[1] = shs[1];
    out[2] = shs[2];
    constexpr int SH_N = 16;
    for (int j = 1; j < SH_N; ++j) {  // planar RRR..GGG..BBB -> interleaved RGB per coefficient
        out[j * 3 + 0] = shs[(j - 1) + 3];
        out[j * 3 + 1] = shs[(j - 1) + SH_N + 2];
        out[j * 3 + 2] = shs[(j - 1) + SH_N * 2 + 1];
    }
}

}  // namespace host
}  // namespace gs
