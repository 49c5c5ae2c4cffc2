// gs_device.h -- device-side helpers shared by the kernel files (wave64, gfx950).
#pragma once
#include "gs_kernels.h"

#include <hip/hip_fp16.h>

#include <algorithm>
#include <cstdio>

namespace gs {

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
}  // namespace gs
