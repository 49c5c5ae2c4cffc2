// gs_device.h -- device-side helpers shared by the kernel files (wave64, gfx950).
#pragma once
#include "gs_kernels.h"

#include <hip/hip_fp16.h>

#include <algorithm>
#include <cstdio>

namespace gs {

// the frame's timeline (gs_kernels.h: FrameStamp): a pass's first kernel stamps its start -- thread 0 of workgroup 0, which the
// dispatcher starts first
__device__ __forceinline__ void frame_stamp(uint64_t* stamps, int which) {
    if (stamps && blockIdx.x == 0 && threadIdx.x == 0) stamps[which] = wall_clock64();
}


#define WAVE 64
#define BLOCK 256

// ---------------------------------------------------------------------------------------
// small column-major 3x3 helpers (GLSL conventions: c[col][row])
// ---------------------------------------------------------------------------------------
struct M3 {
    float c[3][3];
};

// GLSL mat3 * mat3: (A*B)[c][r] = sum_k A[k][r] * B[c][k], k ascending, no fusion.
__device__ __forceinline__ M3 m3_mul(const M3& a, const M3& b) {
    M3 o;
#pragma unroll
    for (int c = 0; c < 3; ++c)
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            float s = a.c[0][r] * b.c[c][0];
            s = s + a.c[1][r] * b.c[c][1];
            s = s + a.c[2][r] * b.c[c][2];
            o.c[c][r] = s;
        }
    return o;
}
__device__ __forceinline__ M3 m3_transpose(const M3& a) {
    M3 o;
#pragma unroll
    for (int c = 0; c < 3; ++c)
#pragma unroll
        for (int r = 0; r < 3; ++r) o.c[c][r] = a.c[r][c];
    return o;
}

// float -> int, toward zero, saturating; after the clamp to [0, tiles] the result equals the
// reference's int() for every in-range input (out-of-range int() is undefined in GLSL).
__device__ __forceinline__ int f2i_sat(float v) {
    v = fminf(fmaxf(v, -2147483648.0f), 2147483520.0f);
    return (int)v;
}
__device__ __forceinline__ int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

// ---------------------------------------------------------------------------------------
// block-wide exclusive scan of one uint per thread (256 threads = 4 waves).
// ---------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t wave_incl_scan(uint32_t v) {
    const int lane = threadIdx.x & (WAVE - 1);
#pragma unroll
    for (int d = 1; d < WAVE; d <<= 1) {
        uint32_t t = __shfl_up(v, d, WAVE);
        if (lane >= d) v += t;
    }
    return v;
}

// Returns the exclusive prefix of v over the block; *total = block sum.  scratch: >= 8 uints of LDS.
template <int THREADS>
__device__ __forceinline__ uint32_t block_excl_scan(uint32_t v, uint32_t* scratch, uint32_t* total) {
    constexpr int NW = THREADS / WAVE;
    const int lane = threadIdx.x & (WAVE - 1), w = threadIdx.x / WAVE;
    uint32_t incl = wave_incl_scan(v);
    __syncthreads();  // scratch reuse
    if (lane == WAVE - 1) scratch[w] = incl;
    __syncthreads();
    uint32_t base = 0, tot = 0;
#pragma unroll
    for (int k = 0; k < NW; ++k) {
        uint32_t s = scratch[k];
        if (k < w) base += s;
        tot += s;
    }
    *total = tot;
    return base + incl - v;
}
// gs_exp without the lower clamp: every lane whose result is used has power in [-7, 0].
__device__ __forceinline__ float gs_exp_blend(float x) {
    const float L2E = 1.44269502162933349609375f;
    const float MAGIC = 12582912.0f;
    float tm = __builtin_fmaf(x, L2E, MAGIC);
    float n = tm - MAGIC;
    float f = __builtin_fmaf(x, L2E, -n);
    float p = 0x1.41d332p-13f;
    p = __builtin_fmaf(p, f, 0x1.5f456ap-10f);
    p = __builtin_fmaf(p, f, 0x1.3b2dbcp-7f);
    p = __builtin_fmaf(p, f, 0x1.c6aed4p-5f);
    p = __builtin_fmaf(p, f, 0x1.ebfbdap-3f);
    p = __builtin_fmaf(p, f, 0x1.62e430p-1f);
    p = __builtin_fmaf(p, f, 1.0f);
    return __uint_as_float(__float_as_uint(p) + (__float_as_uint(tm) << 23));
}

// exp() as glibc's expf evaluates it (glibc >= 2.27, sysdeps/ieee754/flt-32/e_expf.c = ARM optimized-routines expf: x 32/ln2 split
// into k + r in binary64, 2^(k/32) from a 32-entry table, a cubic in r, ONE rounding to binary32 at the end), operation by
// operation in binary64 with the fused operations of the x86-64 FMA build -- so that the blend can be bit-identical to the
// reference's shader text compiled for the CPU (the test suite's checker), whose exp() is libm's.  Restated from the published algorithm,
// the table generated (2^(i/32) correctly rounded, exponent pre-subtracted), and PINNED by tests/test_expf_libm.py: equal to
// this container's libm expf on every binary32 <= 0 (2.1e9 values).  9 binary64 operations (half rate on gfx950) + one
// LDS read: ~23 issue slots against 10 for the polynomial.  DOMAIN: -103.97 <= x <= 0 (glibc's general path; there is no
// underflow branch here: gs_expf_libm_full adds it).  The blend only asks for power >= the entry's alpha cut, which is >= -5.55
// for every opacity <= 1 and >= -104 for every finite one; a NaN or infinite opacity yields alpha = 0.99 whatever exp returns.
static __device__ const uint64_t kExpfTab[32] = {
    0x3ff0000000000000ull, 0x3fefd9b0d3158574ull, 0x3fefb5586cf9890full, 0x3fef9301d0125b51ull, 0x3fef72b83c7d517bull,
    0x3fef54873168b9aaull, 0x3fef387a6e756238ull, 0x3fef1e9df51fdee1ull, 0x3fef06fe0a31b715ull, 0x3feef1a7373aa9cbull,
    0x3feedea64c123422ull, 0x3feece086061892dull, 0x3feebfdad5362a27ull, 0x3feeb42b569d4f82ull, 0x3feeab07dd485429ull,
    0x3feea47eb03a5585ull, 0x3feea09e667f3bcdull, 0x3fee9f75e8ec5f74ull, 0x3feea11473eb0187ull, 0x3feea589994cce13ull,
    0x3feeace5422aa0dbull, 0x3feeb737b0cdc5e5ull, 0x3feec49182a3f090ull, 0x3feed503b23e255dull, 0x3feee89f995ad3adull,
    0x3feeff76f2fb5e47ull, 0x3fef199bdd85529cull, 0x3fef3720dcef9069ull, 0x3fef5818dcfba487ull, 0x3fef7c97337b9b5full,
    0x3fefa4afa2a490daull, 0x3fefd0765b6e4540ull};
__device__ __forceinline__ float gs_expf_libm(float x, const uint2* __restrict__ tab /* kExpfTab, or a copy of it in LDS */) {
    const double InvLn2N = 0x1.71547652b82fep+0 * 32.0, SHIFT = 0x1.8p+52;
    const double C0 = 0x1.c6af84b912394p-5 / 32.0 / 32.0 / 32.0, C1 = 0x1.ebfce50fac4f3p-3 / 32.0 / 32.0, C2 = 0x1.62e42ff0c52d6p-1 / 32.0;
    const double xd = (double)x;
    double kd = __builtin_fma(InvLn2N, xd, SHIFT);          // k = round(x 32/ln2) in the low mantissa bits
    const uint32_t ki = (uint32_t)__double_as_longlong(kd);
    kd = kd - SHIFT;
    const double r = __builtin_fma(InvLn2N, xd, -kd);
    uint2 t = tab[ki & 31u];
    t.y += ki << 15;                                        // t += ki << 47: the exponent of 2^(k/32)
    const double sc = __longlong_as_double((long long)(((uint64_t)t.y << 32) | t.x));
    // glibc evaluates  z = C0 r + C1;  y = C2 r + 1;  y = z r^2 + y;  y = y s  (five operations).  Here the same cubic times the
    // same s in four:  q = (C0 r + C1) r + C2;  y = q (r s) + s.  The two differ in the last bits of the binary64 value, never
    // in its rounding to binary32: tests/test_expf_libm.py runs this very sequence against libm's expf on every binary32 <= 0
    // (IEEE binary64 operations give the same bits on the host as on the device).  q's first fma is written as one VOP3
    // v_fma_f64 (left to itself the compiler copies C1 and uses the two-address v_fmac_f64).
    double q0, q;
    asm("v_fma_f64 %0, %1, %2, %3" : "=v"(q0) : "s"(C0), "v"(r), "v"(C1));
    asm("v_fma_f64 %0, %1, %2, %3" : "=v"(q) : "v"(q0), "v"(r), "v"(C2));
    const double rs = r * sc;
    const double y = __builtin_fma(q, rs, sc);
    return (float)y;
}


// exp() of render.comp:77 for ANY x <= 0, as libm's expf returns it: gs_expf_libm on the algorithm's general path, 0 below
// glibc's underflow bound (e_expf.c: x < -0x1.9fe368p6f -> __math_uflowf) and for -inf.
__device__ __forceinline__ float gs_expf_libm_full(float x, const uint2* __restrict__ tab) {
    return x < -0x1.9fe368p6f ? 0.0f : gs_expf_libm(x, tab);
}

// render.comp:77-79 as a predicate on `power`:  alpha = min(0.99, o * exp(power));  if (alpha < 1/255) continue;
__device__ __forceinline__ bool alpha_kept(float o, float power, const uint2* __restrict__ tab) {
    const float alpha = fminf(0.99f, o * gs_expf_libm_full(power, tab));  // min(0.99, NaN) = 0.99: the pipeline's definition
    return !(alpha < 1.0f / 255.0f);
}

// The ALPHA CUT of a Gaussian: the most negative power at which render.comp:78 still keeps an entry of opacity o.
// libm's expf is monotone on x <= 0 (tests/test_expf_libm.py checks every adjacent pair of binary32 values), a product with
// o > 0 and its rounding are monotone, so  { power <= 0 : alpha(power) >= 1/255 }  is an interval [cut, 0] and
//     power >= cut   <=>   !(min(0.99, o * expf(power)) < 1/255)          bit for bit, for every binary32 power <= 0.
// The blend tests `power` against the cut instead of evaluating the inequality on alpha: the decision of render.comp:78 then
// is the reference's in EVERY exp mode (a fast exp cannot flip it), and lanes below the cut never reach exp at all.
// +inf: no power <= 0 is kept (o <= 1/255 or so, o <= 0); -inf: every power is (o = NaN or +inf).  A bisection over the
// bit patterns of the negative floats, 31 evaluations: computed ONCE per Gaussian at load time (k_alpha_cut).
__device__ __forceinline__ float alpha_cut(float o, const uint2* __restrict__ tab) {
    if (!alpha_kept(o, -0.0f, tab)) return __uint_as_float(0x7F800000u);
    if (alpha_kept(o, __uint_as_float(0xFF800000u), tab)) return __uint_as_float(0xFF800000u);
    uint32_t lo = 0x80000000u, hi = 0xFF800000u;  // kept at lo (-0), not kept at hi (-inf); more negative = larger pattern
    while (hi - lo > 1u) {
        const uint32_t mid = lo + (hi - lo) / 2u;
        if (alpha_kept(o, __uint_as_float(mid), tab)) lo = mid; else hi = mid;
    }
    return __uint_as_float(lo);
}

}  // namespace gs
