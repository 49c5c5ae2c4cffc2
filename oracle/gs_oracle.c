/*
 * gs_oracle.c -- CPU restatement of the shg8/3DGS.cpp splat pipeline.
 *
 * TEST INFRASTRUCTURE ONLY (parity oracle + bench.py cpu_baseline).  See
 * gs_oracle.h for the contract.  PARITY PINNED TO THE REFERENCE'S OWN SHADER TEXT compiled for the CPU
 * (oracle/build_ref.py -> oracle/_ref; tests/test_oracle_vs_ref.py): the reference ships no golden vectors and cannot
 * be built here as a Vulkan program.
 *
 * Third-party arithmetic restated here because the dependency is not vendored
 * in /root/reference: glm 1.0.0 (CMakeLists.txt:31-35): mat4_cast, translate,
 * operator*(mat4,mat4), inverse(mat4), perspectiveRH_NO, radians.
 *
 * Build: gcc -O2 -ffp-contract=off -fno-fast-math -mavx2 -mfma -fopenmp
 * (-mfma only so that the explicit fmaf() calls in gso_exp are inlined).
 */
#define _POSIX_C_SOURCE 200809L
#include "gs_oracle.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include <immintrin.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* ------------------------------------------------------------------------- */
/* small GLSL-like helpers, column-major: m.c[col][row]                        */
/* ------------------------------------------------------------------------- */
typedef struct { float c[3][3]; } mat3;

/* GLSL mat3*mat3: (A*B)[c][r] = sum_k A[k][r]*B[c][k], k ascending,
 * every product and sum rounded separately. */
static mat3 mat3_mul(mat3 a, mat3 b) {
    mat3 o;
    for (int c = 0; c < 3; ++c)
        for (int r = 0; r < 3; ++r) {
            float s = a.c[0][r] * b.c[c][0];
            s = s + a.c[1][r] * b.c[c][1];
            s = s + a.c[2][r] * b.c[c][2];
            o.c[c][r] = s;
        }
    return o;
}
static mat3 mat3_transpose(mat3 a) {
    mat3 o;
    for (int c = 0; c < 3; ++c)
        for (int r = 0; r < 3; ++r) o.c[c][r] = a.c[r][c];
    return o;
}
/* mat4 (column-major float[16]) * vec4, terms added in column order. */
static void mat4_mul_vec4(const float* m, const float* v, float* out) {
    for (int r = 0; r < 4; ++r) {
        float s = m[0 * 4 + r] * v[0];
        s = s + m[1 * 4 + r] * v[1];
        s = s + m[2 * 4 + r] * v[2];
        s = s + m[3 * 4 + r] * v[3];
        out[r] = s;
    }
}
/* float -> int, truncate toward zero, saturate, NaN -> 0 (GLSL int(): out of
 * range undefined; we define it the way v_cvt_i32_f32 behaves). */
static int32_t f2i_sat(float v) {
    if (!(v == v)) return 0;
    if (v >= 2147483648.0f) return INT32_MAX;
    if (v <= -2147483648.0f) return INT32_MIN;
    return (int32_t)v;
}
static int32_t clampi(int32_t v, int32_t lo, int32_t hi) { return v < lo ? lo : (v > hi ? hi : v); }

/* ------------------------------------------------------------------------- */
/* exp(): render.comp:77.  2^(x*log2 e) by round-to-nearest split and a       */
/* degree-6 minimax polynomial on [-0.5, 0.5]; max error < 3 ULP on [-87, 0]. */
/* Defined operation by operation so a GPU can reproduce it bit for bit.      */
/* ------------------------------------------------------------------------- */
float gso_exp(float x) {
    const float L2E = 1.44269502162933349609375f; /* fl(log2 e) = 0x1.715476p+0 */
    const float MAGIC = 12582912.0f;              /* 1.5 * 2^23 */
    x = fmaxf(x, -87.0f);
    x = fminf(x, 88.0f);
    float tm = fmaf(x, L2E, MAGIC); /* integer part lands in the low mantissa bits */
    float n = tm - MAGIC;
    float f = fmaf(x, L2E, -n); /* |f| <= 0.5 (+ rounding) */
    float p = 0x1.41d332p-13f;
    p = fmaf(p, f, 0x1.5f456ap-10f);
    p = fmaf(p, f, 0x1.3b2dbcp-7f);
    p = fmaf(p, f, 0x1.c6aed4p-5f);
    p = fmaf(p, f, 0x1.ebfbdap-3f);
    p = fmaf(p, f, 0x1.62e430p-1f);
    p = fmaf(p, f, 1.0f);
    uint32_t pb, tb;
    memcpy(&pb, &p, 4);
    memcpy(&tb, &tm, 4);
    pb = pb + (tb << 23); /* scale by 2^n through the exponent field */
    memcpy(&p, &pb, 4);
    return p;
}

/* ------------------------------------------------------------------------- */
/* exp() as libm evaluates it.  The reference's shader text compiled for the   */
/* CPU (oracle/_ref) calls libm's expf for render.comp:77; glibc >= 2.27       */
/* (sysdeps/ieee754/flt-32/e_expf.c, from ARM's optimized-routines, 2017) is   */
/* NOT in /root/reference, so its published algorithm is restated here:        */
/*   x*32/ln2 = k + r in binary64 (round-to-nearest by the 0x1.8p52 shift),    */
/*   2^(k/32) from a 32-entry table with the exponent added to the bit         */
/*   pattern, 2^(r/32) ~ C0 r^3 + C1 r^2 + C2 r + 1, one rounding to binary32. */
/* The fused operations are those of glibc's x86-64 FMA build (the ifunc       */
/* variant this container and the GPU boxes select).  The table is generated   */
/* (2^(i/32) correctly rounded, minus i << 47).  tests/test_expf_libm.py pins  */
/* it: bit-equal to this machine's expf on every binary32 in [-87, 0].         */
/* The HIP kernels evaluate the same ten binary64 operations (gs_expf_libm).   */
/* Domain: x <= 0 (the blend never asks for more: power <= 0); below          */
/* -0x1.9fe368p6 (-103.97) the result is 0 like glibc's underflow branch.      */
/* ------------------------------------------------------------------------- */
const uint64_t k_expf_tab_export[32] = {
    0x3ff0000000000000ull, 0x3fefd9b0d3158574ull, 0x3fefb5586cf9890full, 0x3fef9301d0125b51ull, 0x3fef72b83c7d517bull,
    0x3fef54873168b9aaull, 0x3fef387a6e756238ull, 0x3fef1e9df51fdee1ull, 0x3fef06fe0a31b715ull, 0x3feef1a7373aa9cbull,
    0x3feedea64c123422ull, 0x3feece086061892dull, 0x3feebfdad5362a27ull, 0x3feeb42b569d4f82ull, 0x3feeab07dd485429ull,
    0x3feea47eb03a5585ull, 0x3feea09e667f3bcdull, 0x3fee9f75e8ec5f74ull, 0x3feea11473eb0187ull, 0x3feea589994cce13ull,
    0x3feeace5422aa0dbull, 0x3feeb737b0cdc5e5ull, 0x3feec49182a3f090ull, 0x3feed503b23e255dull, 0x3feee89f995ad3adull,
    0x3feeff76f2fb5e47ull, 0x3fef199bdd85529cull, 0x3fef3720dcef9069ull, 0x3fef5818dcfba487ull, 0x3fef7c97337b9b5full,
    0x3fefa4afa2a490daull, 0x3fefd0765b6e4540ull};
/* the table, readable from the tests (tests/test_expf_libm.py regenerates it) */
#define k_expf_tab k_expf_tab_export
float gso_expf_libm(float x) {
    const double InvLn2N = 0x1.71547652b82fep+0 * 32.0, SHIFT = 0x1.8p+52;
    const double C0 = 0x1.c6af84b912394p-5 / 32.0 / 32.0 / 32.0, C1 = 0x1.ebfce50fac4f3p-3 / 32.0 / 32.0,
                 C2 = 0x1.62e42ff0c52d6p-1 / 32.0;
    if (x < -0x1.9fe368p6f) return 0.0f; /* glibc: __math_uflowf */
    const double xd = (double)x;
    double kd = fma(InvLn2N, xd, SHIFT);
    uint64_t ki;
    memcpy(&ki, &kd, 8);
    kd = kd - SHIFT;
    const double r = fma(InvLn2N, xd, -kd);
    uint64_t t = k_expf_tab[ki & 31u] + (ki << 47);
    double sc;
    memcpy(&sc, &t, 8);
    const double z = fma(C0, r, C1);
    const double r2 = r * r;
    double y = fma(C2, r, 1.0);
    y = fma(z, r2, y);
    y = y * sc;
    return (float)y;
}
/* The same function as the HIP kernels evaluate it (gs_expf_libm in csrc/gs_device.h): the cubic and the scale in FOUR
 * binary64 operations -- q = (C0 r + C1) r + C2;  y = q (r s) + s -- instead of glibc's five.  The binary64 values differ in
 * their last bits, their roundings to binary32 never do: gso_expf_device_mismatches counts the binary32 inputs on which
 * this sequence and this machine's libm differ, and tests/test_expf_libm.py requires 0 over every binary32 <= 0. */
float gso_expf_device(float x) {
    const double InvLn2N = 0x1.71547652b82fep+0 * 32.0, SHIFT = 0x1.8p+52;
    const double C0 = 0x1.c6af84b912394p-5 / 32.0 / 32.0 / 32.0, C1 = 0x1.ebfce50fac4f3p-3 / 32.0 / 32.0,
                 C2 = 0x1.62e42ff0c52d6p-1 / 32.0;
    if (x < -0x1.9fe368p6f) return 0.0f; /* (the kernels never ask: power >= -lim there) */
    const double xd = (double)x;
    double kd = fma(InvLn2N, xd, SHIFT);
    uint64_t ki;
    memcpy(&ki, &kd, 8);
    kd = kd - SHIFT;
    const double r = fma(InvLn2N, xd, -kd);
    uint64_t t = k_expf_tab[ki & 31u] + (ki << 47);
    double sc;
    memcpy(&sc, &t, 8);
    double q = fma(C0, r, C1);
    q = fma(q, r, C2);
    const double rs = r * sc;
    const double y = fma(q, rs, sc);
    return (float)y;
}
/* ---- render.comp:78 as a bound on `power`: the ALPHA CUT of an opacity ---------------------------------------------------
 * kept(o, p) = !(min(0.99, o * expf(p)) < 1/255), the shader's :77-79 for one entry and one pixel.  libm's expf is monotone on
 * p <= 0 (gso_expf_monotone_violations counts the adjacent binary32 pairs on which it is not: 0), multiplying by o > 0 and
 * rounding are monotone, so { p <= 0 : kept } = [cut, 0]: the HIP blend decides :78 by `power >= cut` in every exp mode
 * (csrc/gs_device.h: alpha_cut; k_alpha_cut computes the plane at load).  This is the checker's version: the same bisection
 * over the bit patterns of the negative floats, with gso_expf_libm (pinned to libm).  +inf: nothing kept; -inf: everything. */
static int gso_alpha_kept(float o, float p) {
    const float alpha = fminf(0.99f, o * gso_expf_libm(p)); /* fminf(0.99, NaN) = 0.99: the pipeline's definition */
    return !(alpha < 1.0f / 255.0f);
}
float gso_alpha_cut(float o) {
    uint32_t lo = 0x80000000u, hi = 0xFF800000u; /* -0 .. -inf: a more negative value has the larger pattern */
    float x;
    memcpy(&x, &lo, 4);
    if (!gso_alpha_kept(o, x)) return INFINITY;
    memcpy(&x, &hi, 4);
    if (gso_alpha_kept(o, x)) return -INFINITY;
    while (hi - lo > 1u) {
        const uint32_t mid = lo + (hi - lo) / 2u;
        memcpy(&x, &mid, 4);
        if (gso_alpha_kept(o, x)) lo = mid; else hi = mid;
    }
    memcpy(&x, &lo, 4);
    return x;
}
void gso_alpha_cut_array(const float* o, uint64_t n, float* out) {
#pragma omp parallel for
    for (int64_t i = 0; i < (int64_t)n; ++i) out[i] = gso_alpha_cut(o[i]);
}
/* adjacent binary32 pairs (bits b, b + 1: the second is the more negative) in [first_bits, first_bits + count) on which
 * THIS MACHINE'S expf increases as x decreases: the premise of the alpha cut.  tests/test_expf_libm.py requires 0. */
uint64_t gso_expf_monotone_violations(uint32_t first_bits, uint64_t count) {
    uint64_t bad = 0;
#pragma omp parallel for reduction(+ : bad)
    for (int64_t i = 0; i < (int64_t)count; ++i) {
        uint32_t b0 = first_bits + (uint32_t)i, b1 = b0 + 1u;
        float x0, x1;
        memcpy(&x0, &b0, 4);
        memcpy(&x1, &b1, 4);
        if (expf(x1) > expf(x0)) ++bad;
    }
    return bad;
}
/* Checksums of THIS MACHINE'S expf over blocks of 2^20 consecutive bit patterns, the way the device hook gs_debug_expf_scan
 * forms them of the kernels' gs_expf_libm: sums[j] = sum of bits(expf(x)) * ((bits(x) * 0x9E3779B1) | 1) mod 2^64. */
void gso_libm_expf_block_sums(uint32_t first_bits, uint64_t count, uint64_t* sums) {
    const int64_t blocks = (int64_t)((count + (1u << 20) - 1) >> 20);
#pragma omp parallel for schedule(dynamic, 4)
    for (int64_t j = 0; j < blocks; ++j) {
        uint64_t sum = 0;
        for (uint64_t at = (uint64_t)j << 20; at < (((uint64_t)j + 1) << 20) && at < count; ++at) {
            uint32_t xb = first_bits + (uint32_t)at, yb;
            float x, y;
            memcpy(&x, &xb, 4);
            y = expf(x);
            memcpy(&yb, &y, 4);
            sum += (uint64_t)yb * (uint64_t)((xb * 0x9E3779B1u) | 1u);
        }
        sums[j] = sum;
    }
}
/* bulk form for the exhaustive pin: out[i] = gso_expf_libm(bits -> float of first + i) */
void gso_expf_libm_range(uint32_t first_bits, uint64_t count, float* out) {
#pragma omp parallel for
    for (int64_t i = 0; i < (int64_t)count; ++i) {
        uint32_t b = first_bits + (uint32_t)i;
        float x;
        memcpy(&x, &b, 4);
        out[i] = gso_expf_libm(x);
    }
}
/* count of binary32 values with bit patterns first_bits .. first_bits + count - 1 on which gso_expf_libm and this
 * machine's libm expf differ (the pin itself, without moving 4 GB through Python) */
uint64_t gso_expf_device_mismatches(uint32_t first_bits, uint64_t count, uint32_t* first_bad_bits) {
    uint64_t bad = 0;
    uint32_t first_bad = 0xFFFFFFFFu;
#pragma omp parallel for reduction(+ : bad) reduction(min : first_bad)
    for (int64_t i = 0; i < (int64_t)count; ++i) {
        uint32_t b = first_bits + (uint32_t)i, ua, ub;
        float x, a, c;
        memcpy(&x, &b, 4);
        a = gso_expf_device(x);
        c = expf(x);
        memcpy(&ua, &a, 4);
        memcpy(&ub, &c, 4);
        if (ua != ub) {
            ++bad;
            if (b < first_bad) first_bad = b;
        }
    }
    if (first_bad_bits) *first_bad_bits = first_bad;
    return bad;
}
uint64_t gso_expf_libm_mismatches(uint32_t first_bits, uint64_t count, uint32_t* first_bad_bits) {
    uint64_t bad = 0;
    uint32_t first_bad = 0xFFFFFFFFu;
#pragma omp parallel for reduction(+ : bad) reduction(min : first_bad)
    for (int64_t i = 0; i < (int64_t)count; ++i) {
        uint32_t b = first_bits + (uint32_t)i, ua, ub;
        float x, a, c;
        memcpy(&x, &b, 4);
        a = gso_expf_libm(x);
        c = expf(x);
        memcpy(&ua, &a, 4);
        memcpy(&ub, &c, 4);
        if (ua != ub) {
            ++bad;
            if (b < first_bad) first_bad = b;
        }
    }
    if (first_bad_bits) *first_bad_bits = first_bad;
    return bad;
}

/* ------------------------------------------------------------------------- */
/* Camera: Renderer::updateUniforms, src/Renderer.cpp:719-754 (+ glm 1.0.0)   */
/* ------------------------------------------------------------------------- */
static void mat4_identity(float* m) {
    memset(m, 0, 16 * sizeof(float));
    m[0] = m[5] = m[10] = m[15] = 1.0f;
}
/* glm operator*(mat4, mat4): each result column = A0*b0 + A1*b1 + A2*b2 + A3*b3 */
static void mat4_mul(const float* a, const float* b, float* out) {
    float t[16];
    for (int c = 0; c < 4; ++c)
        for (int r = 0; r < 4; ++r) {
            float s = a[0 * 4 + r] * b[c * 4 + 0];
            s = s + a[1 * 4 + r] * b[c * 4 + 1];
            s = s + a[2 * 4 + r] * b[c * 4 + 2];
            s = s + a[3 * 4 + r] * b[c * 4 + 3];
            t[c * 4 + r] = s;
        }
    memcpy(out, t, sizeof t);
}
/* glm::mat4_cast(quat) (gtc/quaternion.inl mat3_cast) */
static void quat_to_mat4(const float* q /* w,x,y,z */, float* m) {
    float w = q[0], x = q[1], y = q[2], z = q[3];
    float qxx = x * x, qyy = y * y, qzz = z * z;
    float qxz = x * z, qxy = x * y, qyz = y * z;
    float qwx = w * x, qwy = w * y, qwz = w * z;
    mat4_identity(m);
    m[0 * 4 + 0] = 1.0f - 2.0f * (qyy + qzz);
    m[0 * 4 + 1] = 2.0f * (qxy + qwz);
    m[0 * 4 + 2] = 2.0f * (qxz - qwy);
    m[1 * 4 + 0] = 2.0f * (qxy - qwz);
    m[1 * 4 + 1] = 1.0f - 2.0f * (qxx + qzz);
    m[1 * 4 + 2] = 2.0f * (qyz + qwx);
    m[2 * 4 + 0] = 2.0f * (qxz + qwy);
    m[2 * 4 + 1] = 2.0f * (qyz - qwx);
    m[2 * 4 + 2] = 1.0f - 2.0f * (qxx + qyy);
}
/* glm::inverse(mat4) (detail/func_matrix.inl compute_inverse<4,4>) */
static void mat4_inverse(const float* m_, float* out) {
#define M(c, r) m_[(c) * 4 + (r)]
    float c00 = M(2, 2) * M(3, 3) - M(3, 2) * M(2, 3);
    float c02 = M(1, 2) * M(3, 3) - M(3, 2) * M(1, 3);
    float c03 = M(1, 2) * M(2, 3) - M(2, 2) * M(1, 3);
    float c04 = M(2, 1) * M(3, 3) - M(3, 1) * M(2, 3);
    float c06 = M(1, 1) * M(3, 3) - M(3, 1) * M(1, 3);
    float c07 = M(1, 1) * M(2, 3) - M(2, 1) * M(1, 3);
    float c08 = M(2, 1) * M(3, 2) - M(3, 1) * M(2, 2);
    float c10 = M(1, 1) * M(3, 2) - M(3, 1) * M(1, 2);
    float c11 = M(1, 1) * M(2, 2) - M(2, 1) * M(1, 2);
    float c12 = M(2, 0) * M(3, 3) - M(3, 0) * M(2, 3);
    float c14 = M(1, 0) * M(3, 3) - M(3, 0) * M(1, 3);
    float c15 = M(1, 0) * M(2, 3) - M(2, 0) * M(1, 3);
    float c16 = M(2, 0) * M(3, 2) - M(3, 0) * M(2, 2);
    float c18 = M(1, 0) * M(3, 2) - M(3, 0) * M(1, 2);
    float c19 = M(1, 0) * M(2, 2) - M(2, 0) * M(1, 2);
    float c20 = M(2, 0) * M(3, 1) - M(3, 0) * M(2, 1);
    float c22 = M(1, 0) * M(3, 1) - M(3, 0) * M(1, 1);
    float c23 = M(1, 0) * M(2, 1) - M(2, 0) * M(1, 1);
    float f0[4] = {c00, c00, c02, c03}, f1[4] = {c04, c04, c06, c07}, f2[4] = {c08, c08, c10, c11};
    float f3[4] = {c12, c12, c14, c15}, f4[4] = {c16, c16, c18, c19}, f5[4] = {c20, c20, c22, c23};
    float v0[4] = {M(1, 0), M(0, 0), M(0, 0), M(0, 0)};
    float v1[4] = {M(1, 1), M(0, 1), M(0, 1), M(0, 1)};
    float v2[4] = {M(1, 2), M(0, 2), M(0, 2), M(0, 2)};
    float v3[4] = {M(1, 3), M(0, 3), M(0, 3), M(0, 3)};
    static const float sa[4] = {+1, -1, +1, -1}, sb[4] = {-1, +1, -1, +1};
    float inv[16];
    for (int i = 0; i < 4; ++i) {
        float i0 = (v1[i] * f0[i] - v2[i] * f1[i]) + v3[i] * f2[i];
        float i1 = (v0[i] * f0[i] - v2[i] * f3[i]) + v3[i] * f4[i];
        float i2 = (v0[i] * f1[i] - v1[i] * f3[i]) + v3[i] * f5[i];
        float i3 = (v0[i] * f2[i] - v1[i] * f4[i]) + v2[i] * f5[i];
        inv[0 * 4 + i] = i0 * sa[i];
        inv[1 * 4 + i] = i1 * sb[i];
        inv[2 * 4 + i] = i2 * sa[i];
        inv[3 * 4 + i] = i3 * sb[i];
    }
    float d0 = M(0, 0) * inv[0 * 4 + 0], d1 = M(0, 1) * inv[1 * 4 + 0];
    float d2 = M(0, 2) * inv[2 * 4 + 0], d3 = M(0, 3) * inv[3 * 4 + 0];
    float det = (d0 + d1) + (d2 + d3);
    float ood = 1.0f / det;
    for (int i = 0; i < 16; ++i) out[i] = inv[i] * ood;
#undef M
}

void gso_camera_uniforms(const gso_camera* cam, uint32_t width, uint32_t height, gso_uniforms* out) {
    memset(out, 0, sizeof *out);
    out->width = width;
    out->height = height;
    out->camera_position[0] = cam->position[0];
    out->camera_position[1] = cam->position[1];
    out->camera_position[2] = cam->position[2];
    out->camera_position[3] = 1.0f;

    float rotation[16], translation[16], tr[16], view[16];
    quat_to_mat4(cam->rotation, rotation);
    /* glm::translate(mat4(1), p): column 3 = m0*x + m1*y + m2*z + m3 */
    mat4_identity(translation);
    translation[12] = cam->position[0];
    translation[13] = cam->position[1];
    translation[14] = cam->position[2];
    mat4_mul(translation, rotation, tr);
    mat4_inverse(tr, view);

    /* Renderer.cpp:730-731: tan(double(radians(fov)) / 2.0) narrowed to float */
    float rad = cam->fov * 0.01745329251994329576923690768489f;
    float tan_fovx = (float)tan((double)rad / 2.0);
    float tan_fovy = tan_fovx * (float)height / (float)width;

    /* glm::perspectiveRH_NO(fovy, aspect, near, far) */
    float fovy = atanf(tan_fovy) * 2.0f;
    float aspect = (float)width / (float)height;
    float zn = cam->near_plane, zf = cam->far_plane;
    float thf = tanf(fovy / 2.0f);
    float persp[16];
    memset(persp, 0, sizeof persp);
    persp[0 * 4 + 0] = 1.0f / (aspect * thf);
    persp[1 * 4 + 1] = 1.0f / thf;
    persp[2 * 4 + 2] = -(zf + zn) / (zf - zn);
    persp[2 * 4 + 3] = -1.0f;
    persp[3 * 4 + 2] = -(2.0f * zf * zn) / (zf - zn);
    mat4_mul(persp, view, out->proj_mat);
    memcpy(out->view_mat, view, sizeof view);

    /* Renderer.cpp:738-750: flip row 1 and row 2 of view, row 1 of proj */
    for (int c = 0; c < 4; ++c) {
        out->view_mat[c * 4 + 1] *= -1.0f;
        out->view_mat[c * 4 + 2] *= -1.0f;
        out->proj_mat[c * 4 + 1] *= -1.0f;
    }
    out->tan_fovx = tan_fovx;
    out->tan_fovy = tan_fovy;
}

/* ------------------------------------------------------------------------- */
/* Scene ingest: GSScene.cpp:17-24 (VertexStorage), :36-59                    */
/* ------------------------------------------------------------------------- */
void gso_activate_records(const float* records, uint64_t n, gso_vertex* out) {
    for (uint64_t i = 0; i < n; ++i) {
        const float* r = records + i * 62; /* pos3 normal3 shs48 opacity scale3 rot4 */
        const float* shs = r + 6;
        float opacity = r[54];
        const float* scale = r + 55;
        const float* rot = r + 58;
        gso_vertex* v = out + i;
        v->position[0] = r[0];
        v->position[1] = r[1];
        v->position[2] = r[2];
        v->position[3] = 1.0f;
        /* :44 glm::exp(vec3) -> std::exp(float); 1/(1+exp(-o)) in float */
        v->scale_opacity[0] = expf(scale[0]);
        v->scale_opacity[1] = expf(scale[1]);
        v->scale_opacity[2] = expf(scale[2]);
        v->scale_opacity[3] = 1.0f / (1.0f + expf(-opacity));
        /* :45 glm::normalize(vec4) = v * inversesqrt(dot(v,v)); glm dot<4> = (x+y)+(z+w) */
        float d0 = rot[0] * rot[0], d1 = rot[1] * rot[1], d2 = rot[2] * rot[2], d3 = rot[3] * rot[3];
        float dot = (d0 + d1) + (d2 + d3);
        float is = 1.0f / sqrtf(dot);
        v->rotation[0] = rot[0] * is;
        v->rotation[1] = rot[1] * is;
        v->rotation[2] = rot[2] * is;
        v->rotation[3] = rot[3] * is;
        /* :47-55 planar f_rest (15 R, 15 G, 15 B) -> interleaved triples */
        v->sh[0] = shs[0];
        v->sh[1] = shs[1];
        v->sh[2] = shs[2];
        const int SH_N = 16;
        for (int j = 1; j < SH_N; ++j) {
            v->sh[j * 3 + 0] = shs[(j - 1) + 3];
            v->sh[j * 3 + 1] = shs[(j - 1) + SH_N + 2];
            v->sh[j * 3 + 2] = shs[(j - 1) + SH_N * 2 + 1];
        }
    }
}

/* GSScene::loadPlyHeader, GSScene.cpp:99-149: only "element vertex N" and
 * "end_header" matter; property names are not validated. */
int gso_load_ply(const char* path, gso_vertex** out, uint64_t* n_out) {
    FILE* f = fopen(path, "rb");
    if (!f) return -1;
    char line[1024];
    long long n = -1;
    int header_end = 0;
    while (fgets(line, sizeof line, f)) {
        char tok[64] = {0}, tok2[64] = {0};
        long long val = 0;
        int k = sscanf(line, "%63s %63s %lld", tok, tok2, &val);
        if (k >= 1 && strcmp(tok, "end_header") == 0) {
            header_end = 1;
            break;
        }
        if (k == 3 && strcmp(tok, "element") == 0 && strcmp(tok2, "vertex") == 0) n = val;
    }
    if (!header_end || n < 0) {
        fclose(f);
        return -2;
    }
    float* rec = (float*)malloc((size_t)n * 62 * sizeof(float));
    gso_vertex* v = (gso_vertex*)malloc((size_t)(n ? n : 1) * sizeof(gso_vertex));
    if (!rec || !v) {
        free(rec);
        free(v);
        fclose(f);
        return -3;
    }
    size_t got = fread(rec, 62 * sizeof(float), (size_t)n, f);
    fclose(f);
    if (got != (size_t)n) {
        free(rec);
        free(v);
        return -4;
    }
    gso_activate_records(rec, (uint64_t)n, v);
    free(rec);
    *out = v;
    *n_out = (uint64_t)n;
    return 0;
}
void gso_free(void* p) { free(p); }

/* ------------------------------------------------------------------------- */
/* common.glsl:51-75 rotationFromQuaternion; precomp_cov3d.comp:25-47         */
/* ------------------------------------------------------------------------- */
static mat3 rotation_from_quaternion(const float* q /* .x=w .y=x .z=y .w=z */) {
    float qx = q[1], qy = q[2], qz = q[3], qw = q[0];
    float qx2 = qx * qx, qy2 = qy * qy, qz2 = qz * qz;
    mat3 m;
    m.c[0][0] = 1 - 2 * qy2 - 2 * qz2;
    m.c[0][1] = 2 * qx * qy - 2 * qz * qw;
    m.c[0][2] = 2 * qx * qz + 2 * qy * qw;
    m.c[1][0] = 2 * qx * qy + 2 * qz * qw;
    m.c[1][1] = 1 - 2 * qx2 - 2 * qz2;
    m.c[1][2] = 2 * qy * qz - 2 * qx * qw;
    m.c[2][0] = 2 * qx * qz - 2 * qy * qw;
    m.c[2][1] = 2 * qy * qz + 2 * qx * qw;
    m.c[2][2] = 1 - 2 * qx2 - 2 * qy2;
    return m;
}

void gso_cov3d(const gso_vertex* v, uint64_t n, float* cov3d) {
    const float scale_factor = 1.0f; /* GSScene.cpp:176 */
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < (int64_t)n; ++i) {
        mat3 S;
        memset(&S, 0, sizeof S);
        S.c[0][0] = v[i].scale_opacity[0] * scale_factor;
        S.c[1][1] = v[i].scale_opacity[1] * scale_factor;
        S.c[2][2] = v[i].scale_opacity[2] * scale_factor;
        mat3 R = rotation_from_quaternion(v[i].rotation);
        mat3 M = mat3_mul(S, R);
        mat3 C = mat3_mul(mat3_transpose(M), M);
        cov3d[i * 6 + 0] = C.c[0][0];
        cov3d[i * 6 + 1] = C.c[0][1];
        cov3d[i * 6 + 2] = C.c[0][2];
        cov3d[i * 6 + 3] = C.c[1][1];
        cov3d[i * 6 + 4] = C.c[1][2];
        cov3d[i * 6 + 5] = C.c[2][2];
    }
}

/* ------------------------------------------------------------------------- */
/* preprocess.comp                                                            */
/* ------------------------------------------------------------------------- */
static const float SH_C0 = 0.28209479177387814f;
static const float SH_C1 = 0.4886025119029199f;
static const float SH_C2[5] = {1.0925484305920792f, -1.0925484305920792f, 0.31539156525252005f,
                               -1.0925484305920792f, 0.5462742152960396f};
static const float SH_C3[7] = {-0.5900435899266435f, 2.890611442640554f, -0.4570457994644658f,
                               0.3731763325901154f,  -0.4570457994644658f, 1.445305721320277f,
                               -0.5900435899266435f};

/* preprocess.comp:34-52 */
static mat3 projection_jacobian_approx(float tx, float ty, float tz, const gso_uniforms* u) {
    float limx = 1.3f * u->tan_fovx;
    float limy = 1.3f * u->tan_fovy;
    float txtz = tx / tz;
    float tytz = ty / tz;
    tx = fminf(limx, fmaxf(-limx, txtz)) * tz;
    ty = fminf(limy, fmaxf(-limy, tytz)) * tz;
    float focal_x = (float)u->width / (2 * u->tan_fovx);
    float focal_y = (float)u->height / (2 * u->tan_fovy);
    mat3 J;
    J.c[0][0] = focal_x / tz;
    J.c[0][1] = 0;
    J.c[0][2] = -(focal_x * tx) / (tz * tz);
    J.c[1][0] = 0;
    J.c[1][1] = focal_y / tz;
    J.c[1][2] = -(focal_y * ty) / (tz * tz);
    J.c[2][0] = 0;
    J.c[2][1] = 0;
    J.c[2][2] = 0;
    return J;
}

/* preprocess.comp:54-66 -> (cov[0][0], cov[0][1], cov[1][0], cov[1][1]) */
static void compute_cov2d(float tx, float ty, float tz, const float* cov3d, const gso_uniforms* u,
                          float* m00, float* m01, float* m10, float* m11) {
    mat3 J = projection_jacobian_approx(tx, ty, tz, u);
    mat3 V3; /* mat3(view_mat): upper-left 3x3 */
    for (int c = 0; c < 3; ++c)
        for (int r = 0; r < 3; ++r) V3.c[c][r] = u->view_mat[c * 4 + r];
    mat3 W = mat3_transpose(V3);
    mat3 Sigma;
    Sigma.c[0][0] = cov3d[0];
    Sigma.c[0][1] = cov3d[1];
    Sigma.c[0][2] = cov3d[2];
    Sigma.c[1][0] = cov3d[1];
    Sigma.c[1][1] = cov3d[3];
    Sigma.c[1][2] = cov3d[4];
    Sigma.c[2][0] = cov3d[2];
    Sigma.c[2][1] = cov3d[4];
    Sigma.c[2][2] = cov3d[5];
    mat3 T = mat3_mul(W, J);
    mat3 cov = mat3_mul(mat3_mul(mat3_transpose(T), Sigma), T);
    cov.c[0][0] += 0.3f;
    cov.c[1][1] += 0.3f;
    *m00 = cov.c[0][0];
    *m01 = cov.c[0][1];
    *m10 = cov.c[1][0];
    *m11 = cov.c[1][1];
}

/* preprocess.comp:73-108 */
static void compute_sh(const gso_vertex* v, const gso_uniforms* u, float* rgb) {
    float dx = v->position[0] - u->camera_position[0];
    float dy = v->position[1] - u->camera_position[1];
    float dz = v->position[2] - u->camera_position[2];
    float len = sqrtf(dx * dx + dy * dy + dz * dz);
    float x = dx / len, y = dy / len, z = dz / len;
    const float* sh = v->sh;
    for (int k = 0; k < 3; ++k) {
#define S(i) sh[(i) * 3 + k]
        float c = SH_C0 * S(0);
        c -= SH_C1 * S(1) * y;
        c += SH_C1 * S(2) * z;
        c -= SH_C1 * S(3) * x;
        c += SH_C2[0] * S(4) * x * y;
        c += SH_C2[1] * S(5) * y * z;
        c += SH_C2[2] * S(6) * (2.0f * z * z - x * x - y * y);
        c += SH_C2[3] * S(7) * z * x;
        c += SH_C2[4] * S(8) * (x * x - y * y);
        c += SH_C3[0] * S(9) * (3.0f * x * x - y * y) * y;
        c += SH_C3[1] * S(10) * x * y * z;
        c += SH_C3[2] * S(11) * (4.0f * z * z - x * x - y * y) * y;
        c += SH_C3[3] * S(12) * z * (2.0f * z * z - 3.0f * x * x - 3.0f * y * y);
        c += SH_C3[4] * S(13) * x * (4.0f * z * z - x * x - y * y);
        c += SH_C3[5] * S(14) * (x * x - y * y) * z;
        c += SH_C3[6] * S(15) * x * (x * x - 3.0f * y * y);
        c += 0.5f;
#undef S
        rgb[k] = c;
    }
    if (rgb[0] < 0.0f) rgb[0] = 0.0f; /* :102-104: only .x is clamped */
}

/* preprocess.comp:110-113 */
static float ndc2pix(float v, int S) { return ((v + 1.0f) * (float)S - 1.0f) * 0.5f; }

void gso_preprocess(const gso_vertex* v, const float* cov3d, uint64_t n, const gso_uniforms* u,
                    gso_vertex_attr* attr, uint32_t* tiles_overlap) {
    const int32_t tile_w = (int32_t)((u->width + 16 - 1) / 16);
    const int32_t tile_h = (int32_t)((u->height + 16 - 1) / 16);
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < (int64_t)n; ++i) {
        gso_vertex_attr* a = attr + i;
        memset(a, 0, sizeof *a); /* :127 radius = 0 (rest: deterministic instead of stale) */
        tiles_overlap[i] = 0;    /* :128 */

        float p_hom[4], p_view[4];
        mat4_mul_vec4(u->proj_mat, v[i].position, p_hom); /* :130 */
        float p_w = 1.0f / p_hom[3];
        float ndc_x = p_hom[0] * p_w, ndc_y = p_hom[1] * p_w;
        mat4_mul_vec4(u->view_mat, v[i].position, p_view); /* :134 */
        if (p_view[2] <= 0.2f) continue;                   /* :135 */

        float m00, m01, m10, m11;
        compute_cov2d(p_view[0], p_view[1], p_view[2], cov3d + i * 6, u, &m00, &m01, &m10, &m11);
        float det = m00 * m11 - m10 * m01; /* determinant(mat2) :140 */
        if (det <= 0.0f) continue;
        float inv_det = 1.0f / det; /* inverse(mat2) :144 */
        a->conic_opacity[0] = m11 * inv_det;
        a->conic_opacity[1] = -m01 * inv_det;
        a->conic_opacity[2] = m00 * inv_det;
        a->conic_opacity[3] = v[i].scale_opacity[3];

        float mid = 0.5f * (m00 + m11); /* :148-152 */
        float sq = sqrtf(fmaxf(0.1f, mid * mid - det));
        float lambda1 = mid + sq;
        float lambda2 = mid - sq;
        float lambda = fmaxf(lambda1, lambda2);
        float radii = ceilf(3.0f * sqrtf(lambda));

        float uvx = ndc2pix(ndc_x, (int)u->width); /* :158 */
        float uvy = ndc2pix(ndc_y, (int)u->height);

        /* :160-165 */
        uint32_t bx0 = (uint32_t)clampi(f2i_sat((uvx - radii) / 16), 0, tile_w);
        uint32_t by0 = (uint32_t)clampi(f2i_sat((uvy - radii) / 16), 0, tile_h);
        uint32_t bx1 = (uint32_t)clampi(f2i_sat((uvx + radii + 16 - 1) / 16), 0, tile_w);
        uint32_t by1 = (uint32_t)clampi(f2i_sat((uvy + radii + 16 - 1) / 16), 0, tile_h);
        uint32_t num = (bx1 - bx0) * (by1 - by0); /* :169 */
        if (num == 0) continue;

        a->aabb[0] = bx0;
        a->aabb[1] = by0;
        a->aabb[2] = bx1;
        a->aabb[3] = by1;
        tiles_overlap[i] = num;
        a->depth = p_view[2];
        a->color_radii[3] = radii;
        compute_sh(v + i, u, a->color_radii);
        a->uv[0] = uvx;
        a->uv[1] = uvy;
        a->magic = 0x4d415449u;
    }
    /* entries culled after the conic was written keep radius 0 and are never
     * read downstream; zero them so the output is a pure function of input. */
    for (uint64_t i = 0; i < n; ++i)
        if (tiles_overlap[i] == 0) memset(attr + i, 0, sizeof attr[i]);
}

/* ------------------------------------------------------------------------- */
void gso_inclusive_scan(const uint32_t* in, uint64_t n, uint32_t* out) {
    uint32_t s = 0;
    for (uint64_t i = 0; i < n; ++i) {
        s += in[i];
        out[i] = s;
    }
}

/* preprocess_sort.comp:31-61 */
void gso_duplicate(const gso_vertex_attr* attr, const uint32_t* prefix, uint64_t n, uint32_t tile_x,
                   uint64_t* keys, uint32_t* payload) {
#pragma omp parallel for schedule(dynamic, 1024)
    for (int64_t index = 0; index < (int64_t)n; ++index) {
        const gso_vertex_attr* a = attr + index;
        if (a->color_radii[3] == 0) continue; /* :37 */
        uint32_t ind = index == 0 ? 0 : prefix[index - 1];
        uint32_t depth_bits;
        memcpy(&depth_bits, &a->depth, 4);
        for (uint32_t i = a->aabb[0]; i < a->aabb[2]; ++i)
            for (uint32_t j = a->aabb[1]; j < a->aabb[3]; ++j) {
                uint64_t tile_index = (uint64_t)i + (uint64_t)j * tile_x;
                keys[ind] = (tile_index << 32) | (uint64_t)depth_bits;
                payload[ind] = (uint32_t)index;
                ind++;
            }
    }
}

/* Stable LSD radix sort, 8 x 8-bit digits over the whole 64-bit key -- the
 * same pass structure as Renderer.cpp:598-629 (hist + scatter per digit),
 * restated sequentially.  Passes whose digit is constant are skipped (the
 * result is unchanged). */
void gso_sort_pairs(uint64_t* keys, uint32_t* payload, uint64_t d) {
    if (d < 2) return;
    uint64_t* k2 = (uint64_t*)malloc(d * sizeof(uint64_t));
    uint32_t* p2 = (uint32_t*)malloc(d * sizeof(uint32_t));
    if (!k2 || !p2) abort();
    uint64_t *ks = keys, *kd = k2;
    uint32_t *ps = payload, *pd = p2;
    /* stable LSD radix, 8 bits per pass.  Each thread owns one contiguous slice of the input: its histogram, then
     * (digit-major, slice-minor) exclusive offsets, then an in-order scatter of its slice -- stable, and the result
     * does not depend on the number of threads. */
    int nt = 1;
#ifdef _OPENMP
    nt = omp_get_max_threads();
    if (nt > 64) nt = 64;
    if ((uint64_t)nt > d / 65536 + 1) nt = (int)(d / 65536 + 1);
#endif
    uint64_t* hist = (uint64_t*)malloc((size_t)nt * 256 * sizeof(uint64_t));
    if (!hist) abort();
    for (int pass = 0; pass < 8; ++pass) {
        int shift = pass * 8;
        memset(hist, 0, (size_t)nt * 256 * sizeof(uint64_t));
#ifdef _OPENMP
#pragma omp parallel for num_threads(nt) schedule(static, 1)
#endif
        for (int t = 0; t < nt; ++t) {
            uint64_t lo = d * (uint64_t)t / (uint64_t)nt, hi = d * (uint64_t)(t + 1) / (uint64_t)nt;
            uint64_t* h = hist + (size_t)t * 256;
            for (uint64_t i = lo; i < hi; ++i) h[(ks[i] >> shift) & 255]++;
        }
        int constant = 0;
        uint64_t sum = 0;
        for (int b = 0; b < 256; ++b) {
            uint64_t digit_total = 0;
            for (int t = 0; t < nt; ++t) {
                uint64_t c = hist[(size_t)t * 256 + b];
                hist[(size_t)t * 256 + b] = sum;
                sum += c;
                digit_total += c;
            }
            if (digit_total == d) constant = 1;
        }
        if (constant) continue; /* every key has the same digit: the pass is the identity */
#ifdef _OPENMP
#pragma omp parallel for num_threads(nt) schedule(static, 1)
#endif
        for (int t = 0; t < nt; ++t) {
            uint64_t lo = d * (uint64_t)t / (uint64_t)nt, hi = d * (uint64_t)(t + 1) / (uint64_t)nt;
            uint64_t* h = hist + (size_t)t * 256;
            for (uint64_t i = lo; i < hi; ++i) {
                uint64_t pos = h[(ks[i] >> shift) & 255]++;
                kd[pos] = ks[i];
                pd[pos] = ps[i];
            }
        }
        uint64_t* tk = ks;
        ks = kd;
        kd = tk;
        uint32_t* tp = ps;
        ps = pd;
        pd = tp;
    }
    if (ks != keys) {
        memcpy(keys, ks, d * sizeof(uint64_t));
        memcpy(payload, ps, d * sizeof(uint32_t));
    }
    free(hist);
    free(k2);
    free(p2);
}

/* Renderer.cpp:633 (fill 0) + tile_boundary.comp:22-50 */
void gso_tile_boundary(const uint64_t* keys, uint64_t d, uint32_t* boundaries, uint64_t num_tiles) {
    memset(boundaries, 0, num_tiles * 2 * sizeof(uint32_t));
    for (uint64_t index = 0; index < d; ++index) {
        uint32_t key = (uint32_t)(keys[index] >> 32);
        if (index == 0) {
            boundaries[key * 2] = (uint32_t)index;
        } else {
            uint32_t prev = (uint32_t)(keys[index - 1] >> 32);
            if (key != prev) {
                boundaries[key * 2] = (uint32_t)index;
                boundaries[prev * 2 + 1] = (uint32_t)index;
            }
        }
        if (index == d - 1) boundaries[key * 2 + 1] = (uint32_t)d;
    }
}

/* render.comp:30-99.
 * GLSL lets a compiler contract a multiply and a dependent add into one FMA (no `precise` in the shader).  The
 * pipeline's definition makes three such contractions (marked below; the HIP kernel makes the same ones), which is
 * what gso_render evaluates.  gso_set_contraction(0) switches gso_render to the UNCONTRACTED reading -- one rounding
 * per operation exactly as the text is written -- which is what the reference's shader text compiled for the CPU
 * (oracle/_ref, -ffp-contract=off) evaluates: with it, oracle and reference text differ in exp() alone, for any scene.
 * The two readings agree to ULP noise unless `power` is a difference of much larger terms (thin, long splats far from
 * their centre), where one rounding more or less moves alpha by ~2^-24 x |terms| (tests/test_oracle_vs_ref.py). */
static int g_contract = 0; /* default: the reference text's reading */
void gso_set_contraction(int on) { g_contract = on != 0; }
/* which exp() render.comp:77 gets: 0 the pipeline polynomial gso_exp (the product's exp mode 0), 2 libm's expf restated
 * (gso_expf_libm: the product's exp mode 2 and, by the pin, what oracle/_ref evaluates) */
static int g_exp_mode = 2; /* default: what the reference text compiled for the CPU calls */
void gso_set_exp_mode(int mode) { g_exp_mode = mode == 2 ? 2 : 0; }

void gso_render(const gso_vertex_attr* attr, const uint32_t* boundaries, const uint32_t* payload,
                uint32_t width, uint32_t height, float* rgba) {
    const int contract = g_contract, exp_libm = g_exp_mode == 2;
    const uint32_t tiles_width = (width + 16 - 1) / 16;
    const uint32_t tiles_height = (height + 16 - 1) / 16;
#pragma omp parallel for schedule(dynamic, 1) collapse(2)
    for (int64_t ty = 0; ty < (int64_t)tiles_height; ++ty)
        for (int64_t tx = 0; tx < (int64_t)tiles_width; ++tx) {
            uint32_t start = boundaries[(tx + ty * tiles_width) * 2];
            uint32_t end = boundaries[(tx + ty * tiles_width) * 2 + 1];
            float* cuts = NULL; /* fast exp readings: the alpha cut of the tile's entries */
            if (!exp_libm && end > start) {
                cuts = (float*)malloc((size_t)(end - start) * sizeof(float));
                for (uint32_t i = start; i < end; ++i) cuts[i - start] = gso_alpha_cut(attr[payload[i]].conic_opacity[3]);
            }
            for (uint32_t ly = 0; ly < 16; ++ly)
                for (uint32_t lx = 0; lx < 16; ++lx) {
                    uint32_t px = (uint32_t)tx * 16 + lx, py = (uint32_t)ty * 16 + ly;
                    if (px >= width || py >= height) continue; /* :36-39 */
                    float T = 1.0f;
                    float c0 = 0.0f, c1 = 0.0f, c2 = 0.0f;
                    for (uint32_t i = start; i < end; ++i) {
                        const gso_vertex_attr* a = attr + payload[i];
                        float dx = a->uv[0] - (float)px;
                        float dy = a->uv[1] - (float)py;
                        const float* co = a->conic_opacity;
                        /* :66  -0.5*(co.x*dx*dx + co.z*dy*dy) - co.y*dx*dy with the two
                         * multiply-adds GLSL allows a compiler to contract written as FMAs */
                        float power;
                        if (contract) {
                            float s = fmaf(co[2] * dy, dy, co[0] * dx * dx);
                            power = fmaf(-(co[1] * dx), dy, -0.5f * s);
                        } else { /* as written, left to right (built with -ffp-contract=off) */
                            power = -0.5f * (co[0] * dx * dx + co[2] * dy * dy) - co[1] * dx * dy;
                        }
                        /* :68.  A NaN power (non-finite conic or centre) also skips: GLSL leaves
                         * the comparison and the following exp(NaN) undefined; the pipeline
                         * defines "no contribution". */
                        if (power > 0.0f || power != power) continue;
                        float alpha = fminf(0.99f, co[3] * (exp_libm ? gso_expf_libm(power) : gso_exp(power))); /* :77 */
                        /* :78.  The default reading evaluates the text.  The product's fast exp modes take this decision
                         * from the reference's arithmetic -- power against the entry's alpha cut (gso_alpha_cut), i.e. whether
                         * the REFERENCE's alpha is >= 1/255 -- and so does the oracle's reading of them. */
                        if (exp_libm ? alpha < 1.0f / 255.0f : power < cuts[i - start]) continue;
                        float test_T = T * (1 - alpha);
                        if (test_T < 0.0001f) break; /* :82-85 */
                        if (contract) {
                            c0 = fmaf(a->color_radii[0] * alpha, T, c0); /* :87, contracted */
                            c1 = fmaf(a->color_radii[1] * alpha, T, c1);
                            c2 = fmaf(a->color_radii[2] * alpha, T, c2);
                        } else { /* c += color * alpha * T */
                            c0 = c0 + a->color_radii[0] * alpha * T;
                            c1 = c1 + a->color_radii[1] * alpha * T;
                            c2 = c2 + a->color_radii[2] * alpha * T;
                        }
                        T = test_T;
                    }
                    float* o = rgba + ((uint64_t)py * width + px) * 4;
                    o[0] = c0;
                    o[1] = c1;
                    o[2] = c2;
                    o[3] = 1.0f; /* :98 */
                }
            free(cuts);
        }
}

static double now_ms(void) {
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return ts.tv_sec * 1e3 + ts.tv_nsec * 1e-6;
}

/* ---- eight pixels per step (AVX2 + FMA): the CPU BASELINE's blend ------------------------------------------------
 * Same operations in the same order as gso_render, lane by lane (explicit FMAs where gso_render has them, separate
 * multiplies and adds elsewhere: -ffp-contract=off holds for intrinsics too), with masks where the scalar code has
 * `continue` / `break`.  gso_render stays the parity checker; tests require the two to agree bit for bit. */
static inline __m256 gso_exp8(__m256 x) {
    const __m256 L2E = _mm256_set1_ps(1.44269502162933349609375f), MAGIC = _mm256_set1_ps(12582912.0f);
    x = _mm256_max_ps(x, _mm256_set1_ps(-87.0f)); /* second operand on NaN, like fmaxf(NaN, -87) */
    x = _mm256_min_ps(x, _mm256_set1_ps(88.0f));
    __m256 tm = _mm256_fmadd_ps(x, L2E, MAGIC);
    __m256 n = _mm256_sub_ps(tm, MAGIC);
    __m256 f = _mm256_fmsub_ps(x, L2E, n); /* fma(x, L2E, -n) */
    __m256 p = _mm256_set1_ps(0x1.41d332p-13f);
    p = _mm256_fmadd_ps(p, f, _mm256_set1_ps(0x1.5f456ap-10f));
    p = _mm256_fmadd_ps(p, f, _mm256_set1_ps(0x1.3b2dbcp-7f));
    p = _mm256_fmadd_ps(p, f, _mm256_set1_ps(0x1.c6aed4p-5f));
    p = _mm256_fmadd_ps(p, f, _mm256_set1_ps(0x1.ebfbdap-3f));
    p = _mm256_fmadd_ps(p, f, _mm256_set1_ps(0x1.62e430p-1f));
    p = _mm256_fmadd_ps(p, f, _mm256_set1_ps(1.0f));
    __m256i pb = _mm256_add_epi32(_mm256_castps_si256(p), _mm256_slli_epi32(_mm256_castps_si256(tm), 23));
    return _mm256_castsi256_ps(pb);
}

/* gso_expf_libm, eight lanes: the same ten binary64 operations per lane (two __m256d halves) */
static inline __m128 gso_expf_libm4(__m128 x) {
    const __m256d InvLn2N = _mm256_set1_pd(0x1.71547652b82fep+0 * 32.0), SHIFT = _mm256_set1_pd(0x1.8p+52);
    const __m256d C0 = _mm256_set1_pd(0x1.c6af84b912394p-5 / 32.0 / 32.0 / 32.0), C1 = _mm256_set1_pd(0x1.ebfce50fac4f3p-3 / 32.0 / 32.0),
                  C2 = _mm256_set1_pd(0x1.62e42ff0c52d6p-1 / 32.0);
    const __m256d xd = _mm256_cvtps_pd(x);
    __m256d kd = _mm256_fmadd_pd(InvLn2N, xd, SHIFT);
    const __m256i ki = _mm256_castpd_si256(kd);
    kd = _mm256_sub_pd(kd, SHIFT);
    const __m256d r = _mm256_fmsub_pd(InvLn2N, xd, kd); /* fma(InvLn2N, xd, -kd) */
    const __m256i idx = _mm256_and_si256(ki, _mm256_set1_epi64x(31));
    __m256i t = _mm256_i64gather_epi64((const long long*)k_expf_tab, idx, 8);
    t = _mm256_add_epi64(t, _mm256_slli_epi64(ki, 47));
    const __m256d sc = _mm256_castsi256_pd(t);
    const __m256d z = _mm256_fmadd_pd(C0, r, C1);
    const __m256d r2 = _mm256_mul_pd(r, r);
    __m256d y = _mm256_fmadd_pd(C2, r, _mm256_set1_pd(1.0));
    y = _mm256_fmadd_pd(z, r2, y);
    y = _mm256_mul_pd(y, sc);
    __m128 res = _mm256_cvtpd_ps(y);
    /* x < -0x1.9fe368p6: 0 (glibc's underflow branch); false for NaN */
    return _mm_andnot_ps(_mm_cmp_ps(x, _mm_set1_ps(-0x1.9fe368p6f), _CMP_LT_OQ), res);
}
static inline __m256 gso_expf_libm8(__m256 x) {
    return _mm256_set_m128(gso_expf_libm4(_mm256_extractf128_ps(x, 1)), gso_expf_libm4(_mm256_castps256_ps128(x)));
}

void gso_render_simd(const gso_vertex_attr* attr, const uint32_t* boundaries, const uint32_t* payload,
                     uint32_t width, uint32_t height, float* rgba) {
    const uint32_t tiles_width = (width + 16 - 1) / 16;
    const uint32_t tiles_height = (height + 16 - 1) / 16;
    const __m256 lane = _mm256_setr_ps(0, 1, 2, 3, 4, 5, 6, 7);
    const int contract = g_contract, exp_libm = g_exp_mode == 2;
#pragma omp parallel for schedule(dynamic, 1) collapse(2)
    for (int64_t ty = 0; ty < (int64_t)tiles_height; ++ty)
        for (int64_t tx = 0; tx < (int64_t)tiles_width; ++tx) {
            const uint32_t start = boundaries[(tx + ty * tiles_width) * 2];
            const uint32_t end = boundaries[(tx + ty * tiles_width) * 2 + 1];
            float* cuts = NULL; /* fast exp readings: the alpha cut of the tile's entries (gso_render) */
            if (!exp_libm && end > start) {
                cuts = (float*)malloc((size_t)(end - start) * sizeof(float));
                for (uint32_t i = start; i < end; ++i) cuts[i - start] = gso_alpha_cut(attr[payload[i]].conic_opacity[3]);
            }
            for (uint32_t ly = 0; ly < 16; ++ly) {
                const uint32_t py = (uint32_t)ty * 16 + ly;
                if (py >= height) continue;
                for (uint32_t half = 0; half < 2; ++half) {
                    const uint32_t px0 = (uint32_t)tx * 16 + half * 8;
                    if (px0 >= width) continue;
                    const __m256 fpx = _mm256_add_ps(_mm256_set1_ps((float)px0), lane); /* exact: small integers */
                    const __m256 fpy = _mm256_set1_ps((float)py);
                    /* lanes inside the image and not yet past their `break` */
                    __m256 active = _mm256_cmp_ps(fpx, _mm256_set1_ps((float)width), _CMP_LT_OQ);
                    __m256 T = _mm256_set1_ps(1.0f), c0 = _mm256_setzero_ps(), c1 = c0, c2 = c0;
                    for (uint32_t i = start; i < end && _mm256_movemask_ps(active); ++i) {
                        const gso_vertex_attr* a = attr + payload[i];
                        const float* co = a->conic_opacity;
                        const __m256 dx = _mm256_sub_ps(_mm256_set1_ps(a->uv[0]), fpx);
                        const __m256 dy = _mm256_sub_ps(_mm256_set1_ps(a->uv[1]), fpy);
                        __m256 power;
                        if (contract) {
                            const __m256 s = _mm256_fmadd_ps(_mm256_mul_ps(_mm256_set1_ps(co[2]), dy), dy,
                                                             _mm256_mul_ps(_mm256_mul_ps(_mm256_set1_ps(co[0]), dx), dx));
                            const __m256 nb = _mm256_xor_ps(_mm256_mul_ps(_mm256_set1_ps(co[1]), dx), _mm256_set1_ps(-0.0f));
                            power = _mm256_fmadd_ps(nb, dy, _mm256_mul_ps(_mm256_set1_ps(-0.5f), s));
                        } else { /* -0.5f * (co.x*dx*dx + co.z*dy*dy) - co.y*dx*dy, as written */
                            const __m256 s = _mm256_add_ps(_mm256_mul_ps(_mm256_mul_ps(_mm256_set1_ps(co[0]), dx), dx),
                                                           _mm256_mul_ps(_mm256_mul_ps(_mm256_set1_ps(co[2]), dy), dy));
                            power = _mm256_sub_ps(_mm256_mul_ps(_mm256_set1_ps(-0.5f), s),
                                                  _mm256_mul_ps(_mm256_mul_ps(_mm256_set1_ps(co[1]), dx), dy));
                        }
                        /* !(power > 0 || power != power)  ==  power <= 0, ordered */
                        __m256 m = _mm256_and_ps(active, _mm256_cmp_ps(power, _mm256_setzero_ps(), _CMP_LE_OQ));
                        if (!_mm256_movemask_ps(m)) continue;
                        /* fminf(0.99f, x): the other operand when x is NaN */
                        const __m256 alpha = _mm256_min_ps(_mm256_mul_ps(_mm256_set1_ps(co[3]), exp_libm ? gso_expf_libm8(power) : gso_exp8(power)),
                                                           _mm256_set1_ps(0.99f));
                        /* :78 -- the text in the default reading, power against the alpha cut in the fast exp readings (gso_render) */
                        m = exp_libm ? _mm256_andnot_ps(_mm256_cmp_ps(alpha, _mm256_set1_ps(1.0f / 255.0f), _CMP_LT_OQ), m)
                                     : _mm256_andnot_ps(_mm256_cmp_ps(power, _mm256_set1_ps(cuts[i - start]), _CMP_LT_OQ), m);
                        const __m256 test_T = _mm256_mul_ps(T, _mm256_sub_ps(_mm256_set1_ps(1.0f), alpha));
                        const __m256 brk = _mm256_and_ps(m, _mm256_cmp_ps(test_T, _mm256_set1_ps(0.0001f), _CMP_LT_OQ));
                        const __m256 upd = _mm256_andnot_ps(brk, m);
                        __m256 n0, n1, n2;
                        if (contract) {
                            n0 = _mm256_fmadd_ps(_mm256_mul_ps(_mm256_set1_ps(a->color_radii[0]), alpha), T, c0);
                            n1 = _mm256_fmadd_ps(_mm256_mul_ps(_mm256_set1_ps(a->color_radii[1]), alpha), T, c1);
                            n2 = _mm256_fmadd_ps(_mm256_mul_ps(_mm256_set1_ps(a->color_radii[2]), alpha), T, c2);
                        } else { /* c += color * alpha * T */
                            n0 = _mm256_add_ps(c0, _mm256_mul_ps(_mm256_mul_ps(_mm256_set1_ps(a->color_radii[0]), alpha), T));
                            n1 = _mm256_add_ps(c1, _mm256_mul_ps(_mm256_mul_ps(_mm256_set1_ps(a->color_radii[1]), alpha), T));
                            n2 = _mm256_add_ps(c2, _mm256_mul_ps(_mm256_mul_ps(_mm256_set1_ps(a->color_radii[2]), alpha), T));
                        }
                        c0 = _mm256_blendv_ps(c0, n0, upd);
                        c1 = _mm256_blendv_ps(c1, n1, upd);
                        c2 = _mm256_blendv_ps(c2, n2, upd);
                        T = _mm256_blendv_ps(T, test_T, upd);
                        active = _mm256_andnot_ps(brk, active);
                    }
                    float r[8], g[8], b[8];
                    _mm256_storeu_ps(r, c0);
                    _mm256_storeu_ps(g, c1);
                    _mm256_storeu_ps(b, c2);
                    for (uint32_t k = 0; k < 8 && px0 + k < width; ++k) {
                        float* o = rgba + ((uint64_t)py * width + px0 + k) * 4;
                        o[0] = r[k];
                        o[1] = g[k];
                        o[2] = b[k];
                        o[3] = 1.0f;
                    }
                }
            }
            free(cuts);
        }
}

static int g_simd_blend = 0;
void gso_set_simd_blend(int on) { g_simd_blend = on != 0; }

int gso_render_frame(const gso_vertex* v, const float* cov3d, uint64_t n, const gso_uniforms* u,
                     float* rgba, gso_stats* stats) {
    const uint32_t tile_x = (u->width + 15) / 16, tile_y = (u->height + 15) / 16;
    const uint64_t T = (uint64_t)tile_x * tile_y;
    gso_stats st;
    memset(&st, 0, sizeof st);
    st.num_gaussians = n;
    gso_vertex_attr* attr = (gso_vertex_attr*)malloc((n ? n : 1) * sizeof *attr);
    uint32_t* tiles = (uint32_t*)malloc((n ? n : 1) * sizeof(uint32_t));
    uint32_t* prefix = (uint32_t*)malloc((n ? n : 1) * sizeof(uint32_t));
    uint32_t* bounds = (uint32_t*)malloc((T ? T : 1) * 2 * sizeof(uint32_t));
    if (!attr || !tiles || !prefix || !bounds) return -1;

    double t0 = now_ms();
    gso_preprocess(v, cov3d, n, u, attr, tiles);
    double t1 = now_ms();
    gso_inclusive_scan(tiles, n, prefix);
    uint64_t d = n ? prefix[n - 1] : 0;
    for (uint64_t i = 0; i < n; ++i) st.num_visible += tiles[i] != 0;
    st.num_instances = d;
    double t2 = now_ms();
    uint64_t* keys = (uint64_t*)malloc((d ? d : 1) * sizeof(uint64_t));
    uint32_t* payload = (uint32_t*)malloc((d ? d : 1) * sizeof(uint32_t));
    if (!keys || !payload) return -1;
    gso_duplicate(attr, prefix, n, tile_x, keys, payload);
    double t3 = now_ms();
    gso_sort_pairs(keys, payload, d);
    double t4 = now_ms();
    gso_tile_boundary(keys, d, bounds, T);
    double t5 = now_ms();
    if (rgba) (g_simd_blend ? gso_render_simd : gso_render)(attr, bounds, payload, u->width, u->height, rgba);
    double t6 = now_ms();
    st.ms[0] = t1 - t0;
    st.ms[1] = t2 - t1;
    st.ms[2] = t3 - t2;
    st.ms[3] = t4 - t3;
    st.ms[4] = t5 - t4;
    st.ms[5] = t6 - t5;
    if (stats) *stats = st;
    free(attr);
    free(tiles);
    free(prefix);
    free(bounds);
    free(keys);
    free(payload);
    return 0;
}

void gso_pack_bgra8(const float* rgba, uint64_t pixels, uint8_t* bgra) {
    for (uint64_t i = 0; i < pixels; ++i) {
        for (int k = 0; k < 4; ++k) {
            float x = rgba[i * 4 + k];
            x = x < 0.0f ? 0.0f : (x > 1.0f ? 1.0f : x);
            if (!(x == x)) x = 0.0f;
            uint8_t q = (uint8_t)(int)rintf(x * 255.0f);
            int dst = k == 0 ? 2 : (k == 2 ? 0 : k);
            bgra[i * 4 + dst] = q;
        }
    }
}

int gso_num_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
void gso_set_num_threads(int n) {
#ifdef _OPENMP
    if (n > 0) omp_set_num_threads(n);
#else
    (void)n;
#endif
}
