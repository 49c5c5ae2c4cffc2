// gs_scene.hip -- load-time kernels: cov3D precompute, SH quantisation, the per-Gaussian alpha cut of render.comp:78.
//
// Part of libgs3d_hip.so (gfx950 only).  Built with -ffp-contract=off: the floating-point contract of this path is "IEEE
// binary32, one rounding per operation, in the order the reference shader writes it" (DESIGN.md section 3); fused
// multiply-adds appear only where written explicitly.
// Reference restated (paths relative to /root/reference/src/shaders): precomp_cov3d.comp:25-47, common.glsl:51-75
#include "gs_device.h"

namespace gs {

// ---------------------------------------------------------------------------------------
// cov3D precompute (load time).  precomp_cov3d.comp:25-47.
// ---------------------------------------------------------------------------------------
__device__ __forceinline__ M3 rotation_from_quaternion(float qw, float qx, float qy, float qz) {
    float qx2 = qx * qx, qy2 = qy * qy, qz2 = qz * qz;
    M3 m;
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

__global__ __launch_bounds__(BLOCK) void k_cov3d(const float* __restrict__ blob, float* __restrict__ cov3d,
                                                 uint32_t n, uint32_t stride) {
    uint32_t i = blockIdx.x * BLOCK + threadIdx.x;
    if (i >= n) return;
    const size_t N = stride, NC = n;
    const float scale_factor = 1.0f;  // GSScene.cpp:176
    M3 S = {};
    S.c[0][0] = blob[(P_SCALE + 0) * N + i] * scale_factor;
    S.c[1][1] = blob[(P_SCALE + 1) * N + i] * scale_factor;
    S.c[2][2] = blob[(P_SCALE + 2) * N + i] * scale_factor;
    M3 R = rotation_from_quaternion(blob[(P_ROT + 0) * N + i], blob[(P_ROT + 1) * N + i],
                                    blob[(P_ROT + 2) * N + i], blob[(P_ROT + 3) * N + i]);
    M3 M = m3_mul(S, R);
    M3 C = m3_mul(m3_transpose(M), M);
    cov3d[0 * NC + i] = C.c[0][0];
    cov3d[1 * NC + i] = C.c[0][1];
    cov3d[2 * NC + i] = C.c[0][2];
    cov3d[3 * NC + i] = C.c[1][1];
    cov3d[4 * NC + i] = C.c[1][2];
    cov3d[5 * NC + i] = C.c[2][2];
}

// Opt-in SH quantisation: the fp32 SH block -> binary16, round to nearest even.
__global__ __launch_bounds__(BLOCK) void k_sh_to_half(const float* __restrict__ sh, uint16_t* __restrict__ out, uint64_t count) {
    const uint64_t i = (uint64_t)blockIdx.x * BLOCK + threadIdx.x;
    if (i < count) out[i] = __half_as_ushort(__float2half_rn(sh[i]));
}
void launch_sh_to_half(const float* blob, uint16_t* sh16, uint32_t n, uint32_t stride, hipStream_t s) {
    if (n == 0) return;
    const uint64_t count = 48ull * n;
    hipLaunchKernelGGL(k_sh_to_half, dim3((uint32_t)((count + BLOCK - 1) / BLOCK)), dim3(BLOCK), 0, s,
                       blob + (size_t)P_SH * stride, sh16, count);
}

// The alpha cut of every Gaussian (gs_device.h: alpha_cut), a function of its opacity alone: one plane of n floats beside cov3D.
// *beyond_unit (nullable, zeroed by the caller) is set when an opacity exceeds 1: the guarded blend's bound assumes the sigmoid's range.
__global__ __launch_bounds__(BLOCK) void k_alpha_cut(const float* __restrict__ opacity, float* __restrict__ cut, uint32_t n,
                                                     uint32_t* __restrict__ beyond_unit) {
    const uint32_t i = blockIdx.x * BLOCK + threadIdx.x;
    if (i >= n) return;
    const float o = opacity[i];
    cut[i] = alpha_cut(o, reinterpret_cast<const uint2*>(kExpfTab));
    if (beyond_unit && o > 1.0f) *beyond_unit = 1u;  // (racing stores of the same value)
}
void launch_alpha_cut(const float* blob, float* cut, uint32_t n, uint32_t stride, uint32_t* beyond_unit, hipStream_t s) {
    if (n == 0) return;
    hipLaunchKernelGGL(k_alpha_cut, dim3((n + BLOCK - 1) / BLOCK), dim3(BLOCK), 0, s, blob + (size_t)P_OPACITY * stride, cut, n, beyond_unit);
}

// The scene's second copy in spatial order (gs_scene::make_spatial_copy): one thread per (Gaussian, 16-byte chunk of its 59 floats).
__global__ __launch_bounds__(BLOCK) void k_permute_blob(const float* __restrict__ src, const uint32_t* __restrict__ perm,
                                                        float* __restrict__ dst, uint32_t n, uint32_t stride) {
    const uint64_t t = (uint64_t)blockIdx.x * BLOCK + threadIdx.x;
    const uint32_t j = (uint32_t)(t / 15u), c = (uint32_t)(t % 15u);  // chunks 0..11: the SH block; 12..14: the 11 planes, four at a time
    if (j >= n) return;
    const uint32_t g = perm[j];
    if (c < 12u) {
        reinterpret_cast<float4*>(dst + (size_t)P_SH * stride)[(size_t)j * 12 + c] =
            reinterpret_cast<const float4*>(src + (size_t)P_SH * stride)[(size_t)g * 12 + c];
    } else {
        for (uint32_t p = (c - 12u) * 4u; p < min((c - 11u) * 4u, (uint32_t)P_SH); ++p) dst[(size_t)p * stride + j] = src[(size_t)p * stride + g];
    }
}
void launch_permute_blob(const float* src, const uint32_t* perm, float* dst, uint32_t n, uint32_t stride, hipStream_t s) {
    if (n == 0) return;
    const uint64_t threads = 15ull * n;
    hipLaunchKernelGGL(k_permute_blob, dim3((uint32_t)((threads + BLOCK - 1) / BLOCK)), dim3(BLOCK), 0, s, src, perm, dst, n, stride);
}

// Checksum of a scene replica (gs_dist_verify): the sum of the blob's bit patterns.
__global__ __launch_bounds__(BLOCK) void k_blob_checksum(const uint32_t* __restrict__ words, uint64_t count, unsigned long long* __restrict__ out) {
    unsigned long long sum = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * BLOCK + threadIdx.x; i < count; i += (uint64_t)gridDim.x * BLOCK) sum += words[i];
    for (int d = 32; d >= 1; d >>= 1) sum += __shfl_xor(sum, d, WAVE);
    if ((threadIdx.x & (WAVE - 1)) == 0 && sum != 0) atomicAdd(out, sum);
}
void launch_blob_checksum(const float* blob, uint64_t floats, uint64_t* out, hipStream_t s) {
    (void)hipMemsetAsync(out, 0, sizeof(uint64_t), s);
    if (floats == 0) return;
    hipLaunchKernelGGL(k_blob_checksum, dim3(2048), dim3(BLOCK), 0, s, reinterpret_cast<const uint32_t*>(blob), floats,
                       reinterpret_cast<unsigned long long*>(out));
}

void launch_cov3d(const float* blob, float* cov3d, uint32_t n, uint32_t stride, hipStream_t s) {
    if (n == 0) return;
    hipLaunchKernelGGL(k_cov3d, dim3((n + BLOCK - 1) / BLOCK), dim3(BLOCK), 0, s, blob, cov3d, n, stride);
}
}  // namespace gs
