// glsl_compat.hpp -- just enough of GLSL 4.50 in C++ to compile the REFERENCE'S OWN compute-shader text for the CPU.
//
// TEST INFRASTRUCTURE ONLY (oracle/_ref).  oracle/build_ref.py reads
//   /root/reference/src/shaders/{common.glsl, precomp_cov3d, preprocess, prefix_sum, preprocess_sort,
//                                tile_boundary, render}.comp
// at build time, applies four mechanical rewrites (documented there), wraps each shader in
// `namespace glsl { namespace cs_<name> { ... } }` and compiles it against this header.  No reference source is
// copied into the repository: only the built oracle/_ref/libgs_ref.so (git-ignored) contains it.
//
// What this header has to define is what the GLSL specification leaves to the implementation:
//   * every operator is one IEEE binary32 operation (the TU is built with -ffp-contract=off, SSE2 scalar math);
//   * matrices are column-major (m[c][r]); matrix products sum over k = 0, 1, 2(, 3) left to right;
//   * min/max/clamp are the specification's definitions ("y if y < x, otherwise x", ...);
//   * determinant / inverse of a mat2 are the cofactor formulas (1/det, then one multiply per element);
//   * length() = sqrt(x*x + y*y + z*z); sqrt, ceil and exp are libm's correctly rounded / < 1 ULP functions;
//   * float -> int conversion is the C++ cast (GLSL leaves out-of-range conversions undefined).
// Everything sits in `namespace glsl` and the shader text is compiled INSIDE that namespace so that glsl::sqrt,
// glsl::min, ... hide the <cmath> overloads instead of competing with them.
#pragma once
#include <cmath>
#include <cstddef>
#include <cstdint>
#include <cstring>
#include <type_traits>

namespace glsl {

typedef unsigned int uint;
typedef ::uint64_t uint64_t;
#define GLSL_SCALAR(A) class = typename std::enable_if<std::is_arithmetic<A>::value>::type

// ---------------------------------------------------------------- vectors
template <class T> struct tvec2 {
    T x, y;
    tvec2() : x(0), y(0) {}
    template <class A, GLSL_SCALAR(A)> explicit tvec2(A s) : x(T(s)), y(T(s)) {}
    template <class A, class B> tvec2(A a, B b) : x(T(a)), y(T(b)) {}
    template <class U> explicit tvec2(const tvec2<U>& o) : x(T(o.x)), y(T(o.y)) {}
    T& operator[](int i) { return (&x)[i]; }
    const T& operator[](int i) const { return (&x)[i]; }
};
template <class T> struct tvec3 {
    T x, y, z;
    tvec3() : x(0), y(0), z(0) {}
    template <class A, GLSL_SCALAR(A)> explicit tvec3(A s) : x(T(s)), y(T(s)), z(T(s)) {}
    template <class A, class B, class C> tvec3(A a, B b, C c) : x(T(a)), y(T(b)), z(T(c)) {}
    template <class U> explicit tvec3(const tvec3<U>& o) : x(T(o.x)), y(T(o.y)), z(T(o.z)) {}
    T& operator[](int i) { return (&x)[i]; }
    const T& operator[](int i) const { return (&x)[i]; }
    tvec3& operator+=(const tvec3& o) { x = x + o.x; y = y + o.y; z = z + o.z; return *this; }
    tvec3& operator-=(const tvec3& o) { x = x - o.x; y = y - o.y; z = z - o.z; return *this; }
    tvec3& operator+=(T s) { x = x + s; y = y + s; z = z + s; return *this; }
    tvec3& operator/=(T s) { x = x / s; y = y / s; z = z / s; return *this; }
};

template <class T> struct tvec4;
// `.xyz` of a 4-vector, readable and assignable (the only multi-component swizzle the shaders use).
template <class T> struct swizzle_xyz {
    T d[4];
    operator tvec3<T>() const { return tvec3<T>(d[0], d[1], d[2]); }
    swizzle_xyz& operator=(const tvec3<T>& v) { d[0] = v.x; d[1] = v.y; d[2] = v.z; return *this; }
};
template <class T> struct tvec4 {
    union {
        struct { T x, y, z, w; };
        swizzle_xyz<T> xyz;
    };
    tvec4() { x = 0; y = 0; z = 0; w = 0; }
    template <class A, GLSL_SCALAR(A)> explicit tvec4(A s) { x = T(s); y = T(s); z = T(s); w = T(s); }
    template <class A, class B, class C, class D> tvec4(A a, B b, C c, D d) { x = T(a); y = T(b); z = T(c); w = T(d); }
    template <class D> tvec4(const tvec3<T>& v, D d) { x = v.x; y = v.y; z = v.z; w = T(d); }
    template <class U> explicit tvec4(const tvec4<U>& o) { x = T(o.x); y = T(o.y); z = T(o.z); w = T(o.w); }
    tvec4(const tvec4& o) { x = o.x; y = o.y; z = o.z; w = o.w; }
    tvec4& operator=(const tvec4& o) { x = o.x; y = o.y; z = o.z; w = o.w; return *this; }
    T& operator[](int i) { return (&x)[i]; }
    const T& operator[](int i) const { return (&x)[i]; }
};

typedef tvec2<float> vec2;
typedef tvec3<float> vec3;
typedef tvec4<float> vec4;
typedef tvec2<int> ivec2;
typedef tvec3<int> ivec3;
typedef tvec4<int> ivec4;
typedef tvec2<uint> uvec2;
typedef tvec3<uint> uvec3;
typedef tvec4<uint> uvec4;
static_assert(sizeof(vec4) == 16 && sizeof(uvec4) == 16 && sizeof(vec2) == 8 && sizeof(vec3) == 12, "std430 sizes");

// non-template operators: the swizzle proxy converts to vec3 implicitly
inline vec2 operator-(vec2 a, vec2 b) { return vec2(a.x - b.x, a.y - b.y); }
inline vec3 operator+(vec3 a, vec3 b) { return vec3(a.x + b.x, a.y + b.y, a.z + b.z); }
inline vec3 operator-(vec3 a, vec3 b) { return vec3(a.x - b.x, a.y - b.y, a.z - b.z); }
inline vec3 operator*(vec3 a, float s) { return vec3(a.x * s, a.y * s, a.z * s); }
inline vec3 operator*(float s, vec3 a) { return vec3(s * a.x, s * a.y, s * a.z); }
inline vec3 operator/(vec3 a, float s) { return vec3(a.x / s, a.y / s, a.z / s); }
inline vec4 operator*(vec4 a, float s) { return vec4(a.x * s, a.y * s, a.z * s, a.w * s); }

// ---------------------------------------------------------------- matrices (column-major, m[c][r])
struct mat3;
struct mat4 {
    vec4 c[4];
    mat4() {}
    vec4& operator[](int i) { return c[i]; }
    const vec4& operator[](int i) const { return c[i]; }
};
struct mat3 {
    vec3 c[3];
    mat3() {}
    explicit mat3(float d) { c[0][0] = d; c[1][1] = d; c[2][2] = d; }
    mat3(float a0, float a1, float a2, float b0, float b1, float b2, float c0, float c1, float c2) {
        c[0] = vec3(a0, a1, a2); c[1] = vec3(b0, b1, b2); c[2] = vec3(c0, c1, c2);
    }
    explicit mat3(const mat4& m) {  // upper-left 3x3
        for (int i = 0; i < 3; i++) c[i] = vec3(m[i].x, m[i].y, m[i].z);
    }
    vec3& operator[](int i) { return c[i]; }
    const vec3& operator[](int i) const { return c[i]; }
};
struct mat2 {
    vec2 c[2];
    mat2() {}
    mat2(float a0, float a1, float b0, float b1) { c[0] = vec2(a0, a1); c[1] = vec2(b0, b1); }
    explicit mat2(const mat3& m) { c[0] = vec2(m[0].x, m[0].y); c[1] = vec2(m[1].x, m[1].y); }
    vec2& operator[](int i) { return c[i]; }
    const vec2& operator[](int i) const { return c[i]; }
};
static_assert(sizeof(mat4) == 64, "std140 mat4");

inline mat3 operator*(const mat3& a, const mat3& b) {
    mat3 r;
    for (int c = 0; c < 3; c++)
        for (int row = 0; row < 3; row++) r[c][row] = a[0][row] * b[c][0] + a[1][row] * b[c][1] + a[2][row] * b[c][2];
    return r;
}
inline vec4 operator*(const mat4& m, const vec4& v) {
    vec4 r;
    for (int row = 0; row < 4; row++) r[row] = m[0][row] * v[0] + m[1][row] * v[1] + m[2][row] * v[2] + m[3][row] * v[3];
    return r;
}
inline mat3 transpose(const mat3& m) {
    mat3 r;
    for (int c = 0; c < 3; c++)
        for (int row = 0; row < 3; row++) r[c][row] = m[row][c];
    return r;
}
inline float determinant(const mat2& m) { return m[0][0] * m[1][1] - m[1][0] * m[0][1]; }
inline mat2 inverse(const mat2& m) {
    float ood = 1.0f / (m[0][0] * m[1][1] - m[1][0] * m[0][1]);
    return mat2(m[1][1] * ood, -m[0][1] * ood, -m[1][0] * ood, m[0][0] * ood);
}

// ---------------------------------------------------------------- built-in functions
inline float sqrt(float x) { return ::sqrtf(x); }
inline float ceil(float x) { return ::ceilf(x); }
inline float exp(float x) { return ::expf(x); }
inline float pow(float x, float y) { return ::powf(x, y); }
inline float min(float x, float y) { return y < x ? y : x; }
inline float max(float x, float y) { return x < y ? y : x; }
inline int min(int x, int y) { return y < x ? y : x; }
inline int max(int x, int y) { return x < y ? y : x; }
inline int clamp(int x, int lo, int hi) { return min(max(x, lo), hi); }
inline float clamp(float x, float lo, float hi) { return min(max(x, lo), hi); }
inline float length(vec3 v) { return sqrt(v.x * v.x + v.y * v.y + v.z * v.z); }
inline uint floatBitsToUint(float f) { uint u; std::memcpy(&u, &f, 4); return u; }

// ---------------------------------------------------------------- resources
template <class T> struct buffer {  // `buffer Block { T name[]; };`
    T* p = nullptr;
    size_t n = 0;
    int length() const { return int(n); }
    T& operator[](size_t i) { return p[i]; }
    void bind(const void* ptr, size_t count) { p = const_cast<T*>(static_cast<const T*>(ptr)); n = count; }
};
struct image2D {  // rgba32f texels, row-major
    float* texels = nullptr;
    int width = 0, height = 0;
};
inline void imageStore(image2D& img, ivec2 p, vec4 v) {
    float* t = img.texels + (size_t(p.y) * img.width + p.x) * 4;
    t[0] = v.x; t[1] = v.y; t[2] = v.z; t[3] = v.w;
}

// ---------------------------------------------------------------- built-in variables (one invocation per thread)
inline thread_local uvec3 gl_GlobalInvocationID, gl_WorkGroupID, gl_LocalInvocationID;
inline thread_local uint gl_LocalInvocationIndex;

// Runs `body` for every invocation of a (gx, gy, 1) grid of (lx, ly, 1) workgroups; workgroups in parallel.
template <class F> inline void dispatch(uint gx, uint gy, uint lx, uint ly, F body) {
#pragma omp parallel for collapse(2) schedule(dynamic, 4)
    for (long wy = 0; wy < long(gy); wy++)
        for (long wx = 0; wx < long(gx); wx++)
            for (uint y = 0; y < ly; y++)
                for (uint x = 0; x < lx; x++) {
                    gl_WorkGroupID = uvec3(uint(wx), uint(wy), 0u);
                    gl_LocalInvocationID = uvec3(x, y, 0u);
                    gl_LocalInvocationIndex = y * lx + x;
                    gl_GlobalInvocationID = uvec3(uint(wx) * lx + x, uint(wy) * ly + y, 0u);
                    body();
                }
}

}  // namespace glsl
