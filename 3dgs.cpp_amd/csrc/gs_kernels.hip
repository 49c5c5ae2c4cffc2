// gs_kernels.hip -- hand-written gfx950 (CDNA4, wave64) kernels of the splat pipeline.
//
// Build with -ffp-contract=off: the floating-point contract of this path is "IEEE binary32,
// one rounding per operation, in the order the reference shader writes it" (DESIGN.md §3), so
// that pixel output does not depend on a compiler's fusion choices.  Fused multiply-adds
// appear only where written explicitly (gs_exp).
//
// Reference shaders restated (paths relative to /root/reference/src/shaders):
//   k_cov3d        precomp_cov3d.comp:25-47, common.glsl:51-75
//   k_preprocess   preprocess.comp:34-183
//   k_radix_*      sort/hist.comp + sort/sort.comp (result: stable ascending order)
//   k_scan_*       prefix_sum.comp:32-59 (result: prefix sums)
//   k_duplicate    preprocess_sort.comp:31-61
//   k_bin_*, k_tile_scan   preprocess_sort.comp + the tile part of the sort + tile_boundary.comp:22-50
//   k_blend        render.comp:30-99
#include "gs_kernels.h"

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

// exp() of render.comp:77 -- the pipeline's definition: 2^(x log2 e) with a
// round-to-nearest integer split done in the mantissa (magic-number add), a degree-6 minimax
// polynomial in explicit FMAs and an exponent-field add.  < 2 ULP on [-6, 0]; inside GLSL's
// 3+2|x| ULP allowance everywhere.  x must be <= 88; values below -87 are clamped.
__device__ __forceinline__ float gs_exp(float x) {
    const float L2E = 1.44269502162933349609375f;
    const float MAGIC = 12582912.0f;
    x = fmaxf(x, -87.0f);
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

void launch_cov3d(const float* blob, float* cov3d, uint32_t n, uint32_t stride, hipStream_t s) {
    if (n == 0) return;
    hipLaunchKernelGGL(k_cov3d, dim3((n + BLOCK - 1) / BLOCK), dim3(BLOCK), 0, s, blob, cov3d, n, stride);
}

// ---------------------------------------------------------------------------------------
// preprocess.  One thread per Gaussian; position / cov3D / opacity are SoA planes (coalesced 256 B per
// wave and plane); the SH block is AoS (48 contiguous floats) and is read only by lanes that survive
// every cull.
// ---------------------------------------------------------------------------------------
constexpr float SH_C0 = 0.28209479177387814f;  // common.glsl:16-33
constexpr float SH_C1 = 0.4886025119029199f;

__device__ __forceinline__ float ndc2pix(float v, int S) { return ((v + 1.0f) * (float)S - 1.0f) * 0.5f; }

struct PreUniforms {
    gs_uniforms u;
    Counters* counters;  // nullable
    uint32_t* nbins;     // nullable: [N] number of S x S-tile bins the tile box touches (0 when culled), index order
    int bin_shift;
};

__global__ __launch_bounds__(BLOCK) void k_preprocess(SceneView sv, PreUniforms pu, AttrView av) {
    const gs_uniforms& u = pu.u;
    uint32_t i = blockIdx.x * BLOCK + threadIdx.x;
    if (i >= sv.n) return;
    if (i == 0 && pu.counters) {  // first kernel of the frame: the counters that are not plainly overwritten
        pu.counters->overflow = 0;
        pu.counters->max_bin = 0;
    }
    const size_t N = sv.stride, NC = sv.n;
    const float* __restrict__ blob = sv.blob;

    const int tile_w = (int)((u.width + kTile - 1) / kTile);
    const int tile_h = (int)((u.height + kTile - 1) / kTile);

    const float px = blob[(P_POS + 0) * N + i];
    const float py = blob[(P_POS + 1) * N + i];
    const float pz = blob[(P_POS + 2) * N + i];

    uint32_t num_tiles = 0, num_bins = 0;
    do {
        // preprocess.comp:130-135 (position.w == 1)
        float p_hom[4], p_view[3];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float s = u.proj_mat[0 * 4 + r] * px;
            s = s + u.proj_mat[1 * 4 + r] * py;
            s = s + u.proj_mat[2 * 4 + r] * pz;
            s = s + u.proj_mat[3 * 4 + r] * 1.0f;
            p_hom[r] = s;
        }
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            float s = u.view_mat[0 * 4 + r] * px;
            s = s + u.view_mat[1 * 4 + r] * py;
            s = s + u.view_mat[2 * 4 + r] * pz;
            s = s + u.view_mat[3 * 4 + r] * 1.0f;
            p_view[r] = s;
        }
        const float p_w = 1.0f / p_hom[3];
        const float ndc_x = p_hom[0] * p_w, ndc_y = p_hom[1] * p_w;
        if (p_view[2] <= 0.2f) break;

        // preprocess.comp:34-52 get_projection_jacobian_approx
        float tx = p_view[0], ty = p_view[1];
        const float tz = p_view[2];
        const float limx = 1.3f * u.tan_fovx;
        const float limy = 1.3f * u.tan_fovy;
        const float txtz = tx / tz;
        const float tytz = ty / tz;
        tx = fminf(limx, fmaxf(-limx, txtz)) * tz;
        ty = fminf(limy, fmaxf(-limy, tytz)) * tz;
        const float focal_x = (float)u.width / (2 * u.tan_fovx);
        const float focal_y = (float)u.height / (2 * u.tan_fovy);
        M3 J;
        J.c[0][0] = focal_x / tz;
        J.c[0][1] = 0;
        J.c[0][2] = -(focal_x * tx) / (tz * tz);
        J.c[1][0] = 0;
        J.c[1][1] = focal_y / tz;
        J.c[1][2] = -(focal_y * ty) / (tz * tz);
        J.c[2][0] = 0;
        J.c[2][1] = 0;
        J.c[2][2] = 0;

        // preprocess.comp:54-66 compute_cov2d
        M3 W;  // transpose(mat3(view_mat))
#pragma unroll
        for (int c = 0; c < 3; ++c)
#pragma unroll
            for (int r = 0; r < 3; ++r) W.c[c][r] = u.view_mat[r * 4 + c];
        const float* __restrict__ cv = sv.cov3d;
        const float s0 = cv[0 * NC + i], s1 = cv[1 * NC + i], s2 = cv[2 * NC + i];
        const float s3 = cv[3 * NC + i], s4 = cv[4 * NC + i], s5 = cv[5 * NC + i];
        M3 Sigma;
        Sigma.c[0][0] = s0;
        Sigma.c[0][1] = s1;
        Sigma.c[0][2] = s2;
        Sigma.c[1][0] = s1;
        Sigma.c[1][1] = s3;
        Sigma.c[1][2] = s4;
        Sigma.c[2][0] = s2;
        Sigma.c[2][1] = s4;
        Sigma.c[2][2] = s5;
        M3 T = m3_mul(W, J);
        M3 cov = m3_mul(m3_mul(m3_transpose(T), Sigma), T);
        const float m00 = cov.c[0][0] + 0.3f;
        const float m11 = cov.c[1][1] + 0.3f;
        const float m01 = cov.c[0][1], m10 = cov.c[1][0];

        const float det = m00 * m11 - m10 * m01;  // :140
        if (det <= 0.0f) break;
        const float inv_det = 1.0f / det;  // inverse(mat2) :144
        const float c00 = m11 * inv_det;
        const float c01 = -m01 * inv_det;
        const float c11 = m00 * inv_det;

        const float mid = 0.5f * (m00 + m11);  // :148-152
        const float sq = sqrtf(fmaxf(0.1f, mid * mid - det));
        const float lambda1 = mid + sq;
        const float lambda2 = mid - sq;
        const float lambda = fmaxf(lambda1, lambda2);
        const float radii = ceilf(3.0f * sqrtf(lambda));

        const float uvx = ndc2pix(ndc_x, (int)u.width);  // :158
        const float uvy = ndc2pix(ndc_y, (int)u.height);

        // :160-165 tile box
        const int bx0 = clampi(f2i_sat((uvx - radii) / kTile), 0, tile_w);
        const int by0 = clampi(f2i_sat((uvy - radii) / kTile), 0, tile_h);
        const int bx1 = clampi(f2i_sat((uvx + radii + kTile - 1) / kTile), 0, tile_w);
        const int by1 = clampi(f2i_sat((uvy + radii + kTile - 1) / kTile), 0, tile_h);
        const uint32_t nt = (uint32_t)(bx1 - bx0) * (uint32_t)(by1 - by0);
        if (nt == 0) break;

        // :73-108 compute_sh (degree 3 always; only .x clamped)
        const float opacity = blob[(size_t)P_OPACITY * N + i];
        float dx = px - u.camera_position[0];
        float dy = py - u.camera_position[1];
        float dz = pz - u.camera_position[2];
        const float len = sqrtf(dx * dx + dy * dy + dz * dz);
        const float x = dx / len, y = dy / len, z = dz / len;
        // the 48 SH floats of a Gaussian are contiguous (192 B = three 64-byte lines): only lanes that
        // survived every cull fetch them, so SH traffic is 192 B per VISIBLE Gaussian
        float sh[48];
        {
            const float4* __restrict__ shv = reinterpret_cast<const float4*>(blob + (size_t)P_SH * N) + (size_t)i * 12;
#pragma unroll
            for (int q = 0; q < 12; ++q) {
                const float4 t = shv[q];
                sh[4 * q + 0] = t.x;
                sh[4 * q + 1] = t.y;
                sh[4 * q + 2] = t.z;
                sh[4 * q + 3] = t.w;
            }
        }
        float rgb[3];
        const float C2_0 = 1.0925484305920792f, C2_1 = -1.0925484305920792f, C2_2 = 0.31539156525252005f,
                    C2_3 = -1.0925484305920792f, C2_4 = 0.5462742152960396f;
        const float C3_0 = -0.5900435899266435f, C3_1 = 2.890611442640554f, C3_2 = -0.4570457994644658f,
                    C3_3 = 0.3731763325901154f, C3_4 = -0.4570457994644658f, C3_5 = 1.445305721320277f,
                    C3_6 = -0.5900435899266435f;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
#define S(j) sh[(j) * 3 + k]
            float c = SH_C0 * S(0);
            c -= SH_C1 * S(1) * y;
            c += SH_C1 * S(2) * z;
            c -= SH_C1 * S(3) * x;
            c += C2_0 * S(4) * x * y;
            c += C2_1 * S(5) * y * z;
            c += C2_2 * S(6) * (2.0f * z * z - x * x - y * y);
            c += C2_3 * S(7) * z * x;
            c += C2_4 * S(8) * (x * x - y * y);
            c += C3_0 * S(9) * (3.0f * x * x - y * y) * y;
            c += C3_1 * S(10) * x * y * z;
            c += C3_2 * S(11) * (4.0f * z * z - x * x - y * y) * y;
            c += C3_3 * S(12) * z * (2.0f * z * z - 3.0f * x * x - 3.0f * y * y);
            c += C3_4 * S(13) * x * (4.0f * z * z - x * x - y * y);
            c += C3_5 * S(14) * (x * x - y * y) * z;
            c += C3_6 * S(15) * x * (x * x - 3.0f * y * y);
            c += 0.5f;
#undef S
            rgb[k] = c;
        }
        if (rgb[0] < 0.0f) rgb[0] = 0.0f;

        num_tiles = nt;
        num_bins = (uint32_t)(((bx1 - 1) >> pu.bin_shift) - (bx0 >> pu.bin_shift) + 1) *
                   (uint32_t)(((by1 - 1) >> pu.bin_shift) - (by0 >> pu.bin_shift) + 1);
        av.depth[i] = p_view[2];
        av.radius[i] = radii;
        av.aabb[i] = make_ushort4((unsigned short)bx0, (unsigned short)by0, (unsigned short)bx1,
                                  (unsigned short)by1);
        av.conic_op[i] = make_float4(c00, c01, c11, opacity);
        av.uv_rg[i] = make_float4(uvx, uvy, rgb[0], rgb[1]);
        av.b[i] = rgb[2];
    } while (false);
    av.tiles[i] = num_tiles;  // :128 / :176
    if (pu.nbins) pu.nbins[i] = num_bins;
}

void launch_preprocess(const SceneView& sv, const gs_uniforms& u, const AttrView& av, Counters* counters,
                       uint32_t* nbins, int bin_shift, hipStream_t s) {
    if (sv.n == 0) return;
    PreUniforms pu;
    pu.u = u;
    pu.counters = counters;
    pu.nbins = nbins;
    pu.bin_shift = bin_shift;
    hipLaunchKernelGGL(k_preprocess, dim3((sv.n + BLOCK - 1) / BLOCK), dim3(BLOCK), 0, s, sv, pu, av);
}

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

// ---------------------------------------------------------------------------------------
// Stable LSD radix pass (8-bit digit), fixed grid, element count resident on the device.
// Element e of a 2048-key tile belongs to wave (e / 512), round ((e % 512) / 64), lane (e % 64):
// every load is a coalesced 256-byte row and tile order == (wave, round, lane) order.
// ---------------------------------------------------------------------------------------
struct RadixArgs {
    const uint32_t* keys_in;
    const uint32_t* vals_in;
    uint32_t* keys_out;
    uint32_t* vals_out;
    const uint32_t* n_in;
    uint32_t n_static;
    const uint32_t* tiles;
    uint32_t* n_out;
    uint32_t* block_hist;
    uint32_t* digit_total;
    const ushort4* gather_aabb;
    uint32_t* tiles_sorted;
    int bin_shift;
    int shift;
    uint32_t mask;
    int blocks;
};

template <bool FIRST>
__device__ __forceinline__ uint32_t radix_count(const RadixArgs& a) {
    if (FIRST) return a.n_static;
    uint32_t n = *a.n_in;
    return n < a.n_static ? n : a.n_static;
}

template <bool FIRST>
__device__ __forceinline__ bool radix_load(const RadixArgs& a, uint32_t e, uint32_t n, uint32_t& key,
                                           uint32_t& val) {
    if (e >= n) return false;
    if (FIRST) {
        if (a.tiles[e] == 0) return false;
        key = a.keys_in[e];  // bits of depth[e]
        val = e;
    } else {
        key = a.keys_in[e];
        val = a.vals_in[e];
    }
    return true;
}

template <bool FIRST>
__global__ __launch_bounds__(BLOCK) void k_radix_hist(RadixArgs a) {
    __shared__ uint32_t hist[256];
    const uint32_t n = radix_count<FIRST>(a);
    const uint32_t ntiles = (n + kSortTileKeys - 1) / kSortTileKeys;
    const uint32_t t0 = (uint32_t)((uint64_t)blockIdx.x * ntiles / a.blocks);
    const uint32_t t1 = (uint32_t)((uint64_t)(blockIdx.x + 1) * ntiles / a.blocks);
    hist[threadIdx.x] = 0;
    __syncthreads();
    for (uint32_t t = t0; t < t1; ++t) {
        const uint32_t base = t * kSortTileKeys;
#pragma unroll
        for (int r = 0; r < kSortTileKeys / BLOCK; ++r) {
            uint32_t key = 0, val = 0;
            if (radix_load<FIRST>(a, base + r * BLOCK + threadIdx.x, n, key, val))
                atomicAdd(&hist[(key >> a.shift) & a.mask], 1u);
        }
    }
    __syncthreads();
    a.block_hist[threadIdx.x * a.blocks + blockIdx.x] = hist[threadIdx.x];
}

// One block per digit: exclusive scan of that digit's row of block counts, row total out.
__global__ __launch_bounds__(BLOCK) void k_radix_scan(uint32_t* block_hist, uint32_t* digit_total, int blocks) {
    __shared__ uint32_t scratch[8];
    uint32_t* row = block_hist + (size_t)blockIdx.x * blocks;
    const int per = (blocks + BLOCK - 1) / BLOCK;  // <= 4
    uint32_t v[4] = {0, 0, 0, 0};
    uint32_t sum = 0;
    for (int k = 0; k < per; ++k) {
        int idx = threadIdx.x * per + k;
        v[k] = idx < blocks ? row[idx] : 0;
        sum += v[k];
    }
    uint32_t total;
    uint32_t excl = block_excl_scan<BLOCK>(sum, scratch, &total);
    for (int k = 0; k < per; ++k) {
        int idx = threadIdx.x * per + k;
        if (idx < blocks) row[idx] = excl;
        excl += v[k];
    }
    if (threadIdx.x == 0) digit_total[blockIdx.x] = total;
}

template <bool FIRST>
__global__ __launch_bounds__(BLOCK) void k_radix_scatter(RadixArgs a) {
    __shared__ uint32_t s_keys[kSortTileKeys];
    __shared__ uint32_t s_vals[kSortTileKeys];
    __shared__ uint32_t s_wcnt[4][256];   // per-wave digit counters, then local positions
    __shared__ uint32_t s_base[256];      // global write cursor of this block per digit
    __shared__ uint32_t s_tcnt[256];      // digit counts of the current tile
    __shared__ uint32_t s_texcl[256];     // exclusive scan of s_tcnt
    __shared__ uint32_t scratch[8];

    const int tid = threadIdx.x, lane = tid & (WAVE - 1), w = tid / WAVE;
    const uint32_t n = radix_count<FIRST>(a);
    const uint32_t ntiles = (n + kSortTileKeys - 1) / kSortTileKeys;
    const uint32_t t0 = (uint32_t)((uint64_t)blockIdx.x * ntiles / a.blocks);
    const uint32_t t1 = (uint32_t)((uint64_t)(blockIdx.x + 1) * ntiles / a.blocks);

    {   // global digit base + this block's prefix inside the digit
        uint32_t tot = a.digit_total[tid], all;
        uint32_t excl = block_excl_scan<BLOCK>(tot, scratch, &all);
        s_base[tid] = excl + a.block_hist[tid * a.blocks + blockIdx.x];
        if (FIRST && blockIdx.x == 0 && tid == 0) *a.n_out = all;
    }
    __syncthreads();

    const uint64_t lt_mask = (1ull << lane) - 1ull;
    for (uint32_t t = t0; t < t1; ++t) {
        const uint32_t base = t * kSortTileKeys + w * 512;
#pragma unroll
        for (int k = 0; k < 4; ++k) s_wcnt[k][tid] = 0;
        __syncthreads();

        uint32_t key[8], val[8], rank[8];
        bool ok[8];
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            key[r] = 0;
            val[r] = 0;
            ok[r] = radix_load<FIRST>(a, base + r * WAVE + lane, n, key[r], val[r]);
        }
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            const uint32_t d = (key[r] >> a.shift) & a.mask;
            uint64_t m = __ballot(ok[r]);
#pragma unroll
            for (int bit = 0; bit < 8; ++bit) {
                const bool set = (d >> bit) & 1u;
                const uint64_t b = __ballot(ok[r] && set);
                m &= set ? b : ~b;
            }
            // m: lanes holding a valid key with my digit (meaningful where ok[r])
            uint32_t old = 0;
            const int leader = m ? (__ffsll((unsigned long long)m) - 1) : 0;
            if (ok[r] && lane == leader) {
                old = s_wcnt[w][d];
                s_wcnt[w][d] = old + (uint32_t)__popcll(m);
            }
            old = __shfl(old, leader, WAVE);
            rank[r] = old + (uint32_t)__popcll(m & lt_mask);
        }
        __syncthreads();
        {   // per-digit: prefix over waves, tile count, exclusive scan over digits
            const uint32_t c0 = s_wcnt[0][tid], c1 = s_wcnt[1][tid], c2 = s_wcnt[2][tid], c3 = s_wcnt[3][tid];
            const uint32_t cnt = c0 + c1 + c2 + c3;
            uint32_t all;
            const uint32_t excl = block_excl_scan<BLOCK>(cnt, scratch, &all);
            s_tcnt[tid] = cnt;
            s_texcl[tid] = excl;
            s_wcnt[0][tid] = excl;
            s_wcnt[1][tid] = excl + c0;
            s_wcnt[2][tid] = excl + c0 + c1;
            s_wcnt[3][tid] = excl + c0 + c1 + c2;
        }
        __syncthreads();
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            if (ok[r]) {
                const uint32_t d = (key[r] >> a.shift) & a.mask;
                const uint32_t pos = s_wcnt[w][d] + rank[r];
                s_keys[pos] = key[r];
                s_vals[pos] = val[r];
            }
        }
        __syncthreads();
        const uint32_t tile_valid = s_texcl[255] + s_tcnt[255];
#pragma unroll
        for (int j = 0; j < kSortTileKeys / BLOCK; ++j) {
            const uint32_t slot = j * BLOCK + tid;
            if (slot < tile_valid) {
                const uint32_t k2 = s_keys[slot], v2 = s_vals[slot];
                const uint32_t d = (k2 >> a.shift) & a.mask;
                const uint32_t dst = s_base[d] + (slot - s_texcl[d]);
                a.keys_out[dst] = k2;
                a.vals_out[dst] = v2;
                if (a.tiles_sorted) {  // number of S x S-tile bins the Gaussian's tile box touches
                    const ushort4 bx = a.gather_aabb[v2];
                    const uint32_t nx = ((bx.z - 1u) >> a.bin_shift) - (bx.x >> a.bin_shift) + 1u;
                    const uint32_t ny = ((bx.w - 1u) >> a.bin_shift) - (bx.y >> a.bin_shift) + 1u;
                    a.tiles_sorted[dst] = nx * ny;
                }
            }
        }
        __syncthreads();
        s_base[tid] += s_tcnt[tid];
        // the barrier after zeroing s_wcnt at the top of the next tile orders this update
    }
}

void launch_radix_pass(const RadixPass& p, hipStream_t s) {
    RadixArgs a;
    a.keys_in = p.keys_in;
    a.vals_in = p.vals_in;
    a.keys_out = p.keys_out;
    a.vals_out = p.vals_out;
    a.n_in = p.n_in;
    a.n_static = p.n_static;
    a.tiles = p.tiles;
    a.n_out = p.n_out;
    a.block_hist = p.block_hist;
    a.digit_total = p.digit_total;
    a.gather_aabb = p.gather_aabb;
    a.bin_shift = p.bin_shift;
    a.tiles_sorted = p.tiles_sorted;
    a.shift = p.shift;
    a.mask = (1u << p.bits) - 1u;
    a.blocks = p.blocks;
    if (p.first) {
        hipLaunchKernelGGL(k_radix_hist<true>, dim3(p.blocks), dim3(BLOCK), 0, s, a);
        hipLaunchKernelGGL(k_radix_scan, dim3(256), dim3(BLOCK), 0, s, p.block_hist, p.digit_total, p.blocks);
        hipLaunchKernelGGL(k_radix_scatter<true>, dim3(p.blocks), dim3(BLOCK), 0, s, a);
    } else {
        hipLaunchKernelGGL(k_radix_hist<false>, dim3(p.blocks), dim3(BLOCK), 0, s, a);
        hipLaunchKernelGGL(k_radix_scan, dim3(256), dim3(BLOCK), 0, s, p.block_hist, p.digit_total, p.blocks);
        hipLaunchKernelGGL(k_radix_scatter<false>, dim3(p.blocks), dim3(BLOCK), 0, s, a);
    }
}

// ---------------------------------------------------------------------------------------
// Exclusive scan (reduce, then downsweep with the spine folded in), fixed grid of kScanBlocks.
// ---------------------------------------------------------------------------------------
__device__ __forceinline__ void scan_range(uint32_t n, uint32_t& lo, uint32_t& hi) {
    // contiguous, 1024-aligned slices
    const uint32_t chunks = (n + 1023) / 1024;
    lo = (uint32_t)((uint64_t)blockIdx.x * chunks / kScanBlocks) * 1024u;
    hi = (uint32_t)((uint64_t)(blockIdx.x + 1) * chunks / kScanBlocks) * 1024u;
    if (hi > n) hi = n;
    if (lo > n) lo = n;
}

__global__ __launch_bounds__(BLOCK) void k_scan_reduce(const uint32_t* __restrict__ cnt, const uint32_t* n_ptr,
                                                       uint32_t n_bound, uint32_t* partial) {
    __shared__ uint32_t scratch[8];
    uint32_t n = n_ptr ? *n_ptr : n_bound;
    if (n > n_bound) n = n_bound;
    uint32_t lo, hi;
    scan_range(n, lo, hi);
    uint32_t sum = 0, nonzero = 0;
    for (uint32_t i = lo + threadIdx.x; i < hi; i += BLOCK) {
        const uint32_t c = cnt[i];
        sum += c;
        nonzero += c != 0 ? 1u : 0u;
    }
    uint32_t total, total_nz;
    block_excl_scan<BLOCK>(sum, scratch, &total);
    block_excl_scan<BLOCK>(nonzero, scratch, &total_nz);
    if (threadIdx.x == 0) {
        partial[blockIdx.x] = total;
        partial[kScanBlocks + blockIdx.x] = total_nz;  // second half: how many entries of the slice are non-zero
    }
}

__global__ __launch_bounds__(BLOCK) void k_scan_down(const uint32_t* __restrict__ cnt, uint32_t* __restrict__ off,
                                                     const uint32_t* n_ptr, uint32_t n_bound,
                                                     const uint32_t* partial, uint32_t* total_out,
                                                     uint32_t* nonzero_out) {
    __shared__ uint32_t scratch[8];
    uint32_t n = n_ptr ? *n_ptr : n_bound;
    if (n > n_bound) n = n_bound;
    uint32_t lo, hi;
    scan_range(n, lo, hi);
    static_assert(kScanBlocks == 2 * BLOCK, "the folded spine reads two partials per thread");
    // spine folded in: every block sums the kScanBlocks (= 2 x 256) block partials before its own (2 KB, L2)
    uint32_t running, grand;
    {
        const uint32_t p0 = partial[threadIdx.x], p1 = partial[threadIdx.x + BLOCK];
        const uint32_t mine = (threadIdx.x < blockIdx.x ? p0 : 0u) + (threadIdx.x + BLOCK < blockIdx.x ? p1 : 0u);
        block_excl_scan<BLOCK>(mine, scratch, &running);
        block_excl_scan<BLOCK>(p0 + p1, scratch, &grand);
        if (blockIdx.x == 0 && threadIdx.x == 0) *total_out = grand;
        if (nonzero_out && blockIdx.x == 0) {  // uniform branch
            uint32_t nz;
            block_excl_scan<BLOCK>(partial[kScanBlocks + threadIdx.x] + partial[kScanBlocks + BLOCK + threadIdx.x], scratch, &nz);
            if (threadIdx.x == 0) *nonzero_out = nz;
        }
    }
    for (uint32_t base = lo; base < hi; base += 1024) {
        const uint32_t i0 = base + threadIdx.x * 4;  // base is 1024-aligned -> 16-byte aligned
        uint32_t v[4] = {0, 0, 0, 0};
        if (i0 + 3 < hi) {
            const uint4 q = *reinterpret_cast<const uint4*>(cnt + i0);
            v[0] = q.x;
            v[1] = q.y;
            v[2] = q.z;
            v[3] = q.w;
        } else {
#pragma unroll
            for (int k = 0; k < 4; ++k)
                if (i0 + k < hi) v[k] = cnt[i0 + k];
        }
        const uint32_t sum = v[0] + v[1] + v[2] + v[3];
        uint32_t total;
        uint32_t excl = running + block_excl_scan<BLOCK>(sum, scratch, &total);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if (i0 + k < hi) off[i0 + k] = excl;
            excl += v[k];
        }
        running += total;
    }
}

void launch_exclusive_scan(const uint32_t* cnt, uint32_t* off, const uint32_t* n, uint32_t n_bound,
                           uint32_t* partial, uint32_t* total_out, uint32_t* nonzero_out, hipStream_t s) {
    hipLaunchKernelGGL(k_scan_reduce, dim3(kScanBlocks), dim3(BLOCK), 0, s, cnt, n, n_bound, partial);
    hipLaunchKernelGGL(k_scan_down, dim3(kScanBlocks), dim3(BLOCK), 0, s, cnt, off, n, n_bound, partial, total_out,
                       nonzero_out);
}

// ---------------------------------------------------------------------------------------
// duplicate: one lane per visible Gaussian (in depth order); boxes of >= 64 tiles are written
// by the whole wave so that the widest splats do not serialise one lane.
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(BLOCK) void k_duplicate(const uint32_t* __restrict__ order,
                                                     const uint32_t* __restrict__ off,
                                                     const uint32_t* __restrict__ tiles_sorted,
                                                     const ushort4* __restrict__ aabb, const uint32_t* n_visible,
                                                     uint32_t n_bound, uint32_t tiles_x, int shift, uint32_t capacity,
                                                     uint32_t* __restrict__ inst_tile,
                                                     uint32_t* __restrict__ inst_gid, Counters* counters) {
    uint32_t n = n_visible ? *n_visible : n_bound;  // null: every Gaussian, in index order (bin-local path)
    if (n > n_bound) n = n_bound;
    const uint32_t j = blockIdx.x * BLOCK + threadIdx.x;
    const int lane = threadIdx.x & (WAVE - 1);
    if (blockIdx.x == 0 && threadIdx.x == 0 && counters->bin_entries > capacity) counters->overflow |= 1u;
    if (blockIdx.x * BLOCK >= n) return;

    bool valid = j < n;
    uint32_t g = 0, o = 0, cnt = 0;
    ushort4 box = make_ushort4(0, 0, 0, 0);
    if (valid) {
        g = order ? order[j] : j;
        o = off[j];
        cnt = tiles_sorted[j];
        if (cnt) box = aabb[g];
        if (shift) {  // tile box -> box of (tile >> shift) bins, upper bounds exclusive
            box.z = (unsigned short)(((box.z - 1u) >> shift) + 1u);
            box.w = (unsigned short)(((box.w - 1u) >> shift) + 1u);
            box.x = (unsigned short)(box.x >> shift);
            box.y = (unsigned short)(box.y >> shift);
        }
        if ((uint64_t)o + cnt > capacity) valid = false;  // overflow: frame is re-run after growing
    }
    const bool big = valid && cnt >= WAVE;
    uint64_t bm = __ballot(big);
    while (bm) {
        const int src = __ffsll((unsigned long long)bm) - 1;
        bm &= bm - 1;
        const uint32_t x0 = __shfl((uint32_t)box.x, src, WAVE), y0 = __shfl((uint32_t)box.y, src, WAVE);
        const uint32_t y1 = __shfl((uint32_t)box.w, src, WAVE);
        const uint32_t c = __shfl(cnt, src, WAVE), ob = __shfl(o, src, WAVE), gs_ = __shfl(g, src, WAVE);
        const uint32_t h = y1 - y0;
        for (uint32_t k = lane; k < c; k += WAVE) {
            const uint32_t xi = k / h, yi = k - xi * h;  // x outer, y inner (preprocess_sort.comp:47-48)
            inst_tile[ob + k] = (x0 + xi) + (y0 + yi) * tiles_x;
            inst_gid[ob + k] = gs_;
        }
    }
    if (valid && !big) {
        uint32_t ind = o;
        for (uint32_t x = box.x; x < box.z; ++x)
            for (uint32_t y = box.y; y < box.w; ++y) {
                inst_tile[ind] = x + y * tiles_x;
                inst_gid[ind] = g;
                ++ind;
            }
    }
}

void launch_duplicate(const uint32_t* order, const uint32_t* off, const uint32_t* tiles_sorted,
                      const ushort4* aabb, const uint32_t* n_visible, uint32_t n_bound, uint32_t tiles_x,
                      int shift, uint32_t capacity, uint32_t* inst_tile, uint32_t* inst_gid, Counters* counters,
                      hipStream_t s) {
    if (n_bound == 0) return;
    hipLaunchKernelGGL(k_duplicate, dim3((n_bound + BLOCK - 1) / BLOCK), dim3(BLOCK), 0, s, order, off,
                       tiles_sorted, aabb, n_visible, n_bound, tiles_x, shift, capacity, inst_tile, inst_gid, counters);
}

// ---------------------------------------------------------------------------------------
// Hierarchical binning: per-tile lists without sorting the D instances.
//
// The screen is cut into <= 256 bins of S x S tiles.  Level 1 (existing kernels) lists, per bin, the
// Gaussians whose tile box touches it, in depth order: a scan, k_duplicate on bin boxes and ONE stable
// 8-bit radix pass over E1 ~ 1.2 V entries -- instead of two passes over D ~ 10 V.  Level 2 (below)
// expands each bin's list into its S*S tiles: 256-candidate chunks count per tile (k_bin_count), the
// per-tile counts are scanned over chunks and over tiles (k_bin_scan, k_tile_scan -> the ranges and D),
// and k_bin_fill writes every Gaussian id at  range.start + chunk prefix + rank-in-chunk,  the rank
// coming from a wave64 ballot per tile.  Depth order is preserved at every step (stable radix pass,
// chunks in order, lanes in order), so each tile's list equals the reference's stably sorted payload.
// The instance data moved through HBM drops from ~48 B to ~4 B per instance.
// ---------------------------------------------------------------------------------------
struct BinArgs {
    const uint32_t* cand;         // bin-major, depth-ordered Gaussian ids (level-1 output)
    const uint32_t* bin_count;    // [256] candidates per bin (the radix pass's digit totals)
    const ushort4* aabb;
    uint32_t* chunk_hist;         // [chunk][S*S]: count, then exclusive prefix over the bin's chunks
    uint32_t* tile_total;         // [T]
    uint32_t* ranges;             // [T][2]
    uint32_t* sorted_gid;         // [capacity]
    Counters* counters;
    uint32_t capacity;
    uint32_t tiles_x, tiles_y, bins_x;
    int shift;                    // log2(S)
    uint32_t max_chunks;
};

constexpr int kBinChunk = 64;     // candidates per chunk (= one wave; the four waves of a workgroup are independent)
constexpr int kBinGrid = 2048;    // persistent grid (x4 waves) striding over the device-resident chunk count

// Offsets and chunk prefixes of all bins (once per block); returns the total number of chunks.
__device__ __forceinline__ uint32_t bin_prepare(const BinArgs& a, uint32_t* s_off, uint32_t* s_cpre, uint32_t* scratch) {
    const int tid = threadIdx.x;
    const uint32_t c = a.bin_count[tid];
    uint32_t total, total_chunks;
    const uint32_t off = block_excl_scan<BLOCK>(c, scratch, &total);
    const uint32_t nch = (c + kBinChunk - 1) / kBinChunk;
    const uint32_t cpre = block_excl_scan<BLOCK>(nch, scratch, &total_chunks);
    s_off[tid] = off;
    s_cpre[tid] = cpre;
    __syncthreads();
    return __builtin_amdgcn_readfirstlane(total_chunks);  // same in every lane: keep it scalar
}

// chunk id -> (bin, first candidate, candidate count); uniform binary search over the 256 chunk prefixes
__device__ __forceinline__ void bin_locate(const BinArgs& a, const uint32_t* s_off, const uint32_t* s_cpre, uint32_t chunk,
                                           uint32_t& bin, uint32_t& first, uint32_t& count) {
    uint32_t lo = 0, hi = 256;  // last bin whose chunk prefix is <= chunk and that owns chunks
#pragma unroll
    for (int it = 0; it < 8; ++it) {
        const uint32_t mid = (lo + hi) >> 1;
        if (s_cpre[mid] <= chunk) lo = mid; else hi = mid;
    }
    // empty bins share the prefix of their successor: step forward to the owner of this chunk
    while (a.bin_count[lo] == 0) ++lo;
    bin = lo;
    const uint32_t q = chunk - s_cpre[bin];
    first = s_off[bin] + q * kBinChunk;
    const uint32_t end = s_off[bin] + a.bin_count[bin];
    count = min((uint32_t)kBinChunk, end - first);
}

// candidate's tile box clipped to the bin, in bin-local tile coordinates (upper bounds exclusive)
__device__ __forceinline__ void bin_local_box(const BinArgs& a, uint32_t bin, ushort4 box, int& lx0, int& ly0, int& lx1,
                                              int& ly1) {
    const int S = 1 << a.shift;
    const int ox = (int)(bin % a.bins_x) << a.shift, oy = (int)(bin / a.bins_x) << a.shift;
    lx0 = max((int)box.x, ox) - ox;
    ly0 = max((int)box.y, oy) - oy;
    lx1 = min((int)box.z, ox + S) - ox;
    ly1 = min((int)box.w, oy + S) - oy;
}

// Coverage of a candidate's bin-local box as bit masks over the bin's S*S tiles: tile t = y*S + x lives in
// bit (t % 64) of word (t / 64), i.e. lane (t % 64) "owns" tile t in register slot t / 64.  R = S*S/64 words.
template <int R>
__device__ __forceinline__ void bin_cover_masks(int shift, int lx0, int ly0, int lx1, int ly1, uint64_t (&m)[R]) {
    const int S = 1 << shift;
    const int rows_per_word = 64 >> shift;  // 8, 4, 2 for S = 8, 16, 32
    const uint64_t rowbits = lx1 > lx0 ? ((lx1 - lx0 >= 64 ? ~0ull : ((1ull << (lx1 - lx0)) - 1ull)) << lx0) : 0ull;
#pragma unroll
    for (int r = 0; r < R; ++r) {
        uint64_t w = 0;
        for (int yy = 0; yy < rows_per_word; ++yy) {
            const int y = r * rows_per_word + yy;
            if (y >= ly0 && y < ly1) w |= rowbits << (yy * S);
        }
        m[r] = w;
    }
}

// 64 x 64 bit-matrix transpose across a wave: lane k enters with row k, lane t leaves with column t
// (bit k of the result = bit t of lane k's input).  Six butterfly steps; step j swaps the off-diagonal j x j
// blocks between lanes l and l ^ j.  ds_swizzle is a lane permutation inside 32-lane halves (no LDS memory).
template <int J>
__device__ __forceinline__ uint32_t transpose_step(uint32_t x, bool up) {
    constexpr uint32_t MASK = J == 16 ? 0x0000FFFFu : J == 8 ? 0x00FF00FFu : J == 4 ? 0x0F0F0F0Fu
                            : J == 2 ? 0x33333333u : 0x55555555u;
    const uint32_t p = (uint32_t)__builtin_amdgcn_ds_swizzle((int)x, (J << 10) | 0x1F);  // lane ^ J
    return up ? (((p >> J) & MASK) | (x & ~MASK)) : ((x & MASK) | ((p & MASK) << J));
}
__device__ __forceinline__ uint64_t wave_transpose64(uint64_t x, uint32_t lane) {
    uint32_t lo = (uint32_t)x, hi = (uint32_t)(x >> 32);
    {   // j = 32: the low lanes' high words and the high lanes' low words change places
        const bool up = (lane & 32u) != 0;
        const uint32_t recv = (uint32_t)__shfl_xor((int)(up ? lo : hi), 32);
        if (up) lo = recv; else hi = recv;
    }
    { const bool up = (lane & 16u) != 0; lo = transpose_step<16>(lo, up); hi = transpose_step<16>(hi, up); }
    { const bool up = (lane & 8u) != 0;  lo = transpose_step<8>(lo, up);  hi = transpose_step<8>(hi, up); }
    { const bool up = (lane & 4u) != 0;  lo = transpose_step<4>(lo, up);  hi = transpose_step<4>(hi, up); }
    { const bool up = (lane & 2u) != 0;  lo = transpose_step<2>(lo, up);  hi = transpose_step<2>(hi, up); }
    { const bool up = (lane & 1u) != 0;  lo = transpose_step<1>(lo, up);  hi = transpose_step<1>(hi, up); }
    return ((uint64_t)hi << 32) | lo;
}

// Per-wave expansion of its 64 candidates into the tiles of the bin.  Lane k enters with candidate k's coverage
// words; a bit-matrix transpose per word puts lane t in possession of tile t's column: which of the 64
// candidates cover it, in candidate (= depth) order.  Counting is a popcount.  Filling walks each lane's own
// set bits: "k = lowest set bit, store ids[k] at cursor, cursor++", for as many rounds as the fullest tile of
// the chunk needs (typically 8-16, where a candidate-major walk issues 64 mostly-empty scattered stores: the
// scattered-store issue rate is what bounds this kernel).  The store is a raw buffer store whose offset is
// forced out of range on lanes that have run dry, so the hardware's bounds check drops it (and enforces the
// list capacity) without a branch.
template <int R, bool FILL>
__device__ __forceinline__ void bin_walk(const uint64_t (&m)[R], uint32_t (&cursor)[R], __amdgpu_buffer_rsrc_t out,
                                         const uint32_t* __restrict__ ids /* wave-private [64]: the candidates' ids */) {
    const uint32_t lane = threadIdx.x & (WAVE - 1);
#pragma unroll
    for (int r = 0; r < R; ++r) {
        if (R > 1 && __builtin_amdgcn_ballot_w64(m[r] != 0) == 0) {  // uniform: no candidate touches this word
            if (!FILL) cursor[r] = 0;
            continue;
        }
        uint64_t col = wave_transpose64(m[r], lane);
        if (!FILL) {
            cursor[r] = (uint32_t)__popcll(col);
            continue;
        }
        // up to four ids per lane and round, written with ONE store of 4..16 bytes: the lists are written at ~1 TB/s
        // because every lane's store is its own request to the L2 (a different line per lane), so what counts is the
        // number of requests, not of bytes.  One store instruction per size; lanes of another size are out of range.
        typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
        typedef uint32_t u32x3 __attribute__((ext_vector_type(3)));
        typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
        uint32_t cur = cursor[r];
        while (__builtin_amdgcn_ballot_w64(col != 0) != 0) {
            const uint32_t pc = (uint32_t)__popcll(col);
            const uint32_t cnt = pc < 4u ? pc : 4u;
            uint32_t id[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {  // an exhausted column reads slot 63 and drops the value below
                id[j] = ids[(uint32_t)(__ffsll((unsigned long long)col) - 1) & 63u];
                col &= col - 1;
            }
            const uint32_t off = cur * 4u;
            __builtin_amdgcn_raw_buffer_store_b32(id[0], out, cnt == 1u ? off : 0xFFFFFFFFu, 0, 0);
            u32x2 v2 = {id[0], id[1]};
            __builtin_amdgcn_raw_buffer_store_b64(v2, out, cnt == 2u ? off : 0xFFFFFFFFu, 0, 0);
            u32x3 v3 = {id[0], id[1], id[2]};
            __builtin_amdgcn_raw_buffer_store_b96(v3, out, cnt == 3u ? off : 0xFFFFFFFFu, 0, 0);
            u32x4 v4 = {id[0], id[1], id[2], id[3]};
            __builtin_amdgcn_raw_buffer_store_b128(v4, out, cnt == 4u ? off : 0xFFFFFFFFu, 0, 0);
            cur += cnt;
        }
    }
}

template <int R>
__global__ __launch_bounds__(BLOCK) void k_bin_count(BinArgs a) {
    __shared__ uint32_t s_off[256], s_cpre[256], scratch[8];
    uint32_t total_chunks = bin_prepare(a, s_off, s_cpre, scratch);
    const int tid = threadIdx.x, lane = tid & (WAVE - 1), w = tid / WAVE;
    constexpr int SS = 64 * R;
    if (total_chunks > a.max_chunks) {  // chunk table too small: flag, stay in bounds; the frame is re-run
        if (blockIdx.x == 0 && tid == 0) a.counters->overflow |= 1u;
        total_chunks = a.max_chunks;
    }
    // the chunk loop is per wave: tell the compiler its control values are wave-uniform (scalar registers)
    const uint32_t wu = __builtin_amdgcn_readfirstlane((uint32_t)w);
    // chunk -> wave: groups of 16 consecutive chunks (1024 candidates, nearly always one bin) stay on one XCD
    // (workgroup b runs on XCD b % 8), so that the partial lines of that bin's tile lists are completed in one L2
    const uint32_t xcd = blockIdx.x % 8, slot = (blockIdx.x / 8) * 4 + wu, slots = (gridDim.x / 8) * 4;
    const uint32_t q_end = (((total_chunks + 15) / 16 + 7) / 8) * 16;
    for (uint32_t q = slot; q < q_end; q += slots) {
        const uint32_t chunk = ((q / 16) * 8 + xcd) * 16 + (q % 16);
        if (chunk >= total_chunks) continue;
        uint32_t bin, first, count;
        bin_locate(a, s_off, s_cpre, chunk, bin, first, count);
        bin = __builtin_amdgcn_readfirstlane(bin);
        first = __builtin_amdgcn_readfirstlane(first);
        count = __builtin_amdgcn_readfirstlane(count);
        uint64_t m[R];
        uint32_t g = 0;
        int lx0 = 0, ly0 = 0, lx1 = 0, ly1 = 0;
        if ((uint32_t)lane < count) {
            g = a.cand[first + lane];
            bin_local_box(a, bin, a.aabb[g], lx0, ly0, lx1, ly1);
        }
        bin_cover_masks<R>(a.shift, lx0, ly0, lx1, ly1, m);
        uint32_t cursor[R];
        bin_walk<R, false>(m, cursor, __builtin_amdgcn_make_buffer_rsrc(a.sorted_gid, 0, 0, 0x27000), nullptr);
        uint32_t* out = a.chunk_hist + (size_t)chunk * SS;
#pragma unroll
        for (int r = 0; r < R; ++r) out[r * 64 + lane] = cursor[r];
    }
}

// One block per bin: for each of its tiles, exclusive prefix of the chunk counts (in place) and the tile total.
__global__ __launch_bounds__(BLOCK) void k_bin_scan(BinArgs a) {
    __shared__ uint32_t scratch[8];
    __shared__ uint32_t s_seg[4][64];
    const int tid = threadIdx.x, S = 1 << a.shift, SS = S * S;
    const uint32_t bin = blockIdx.x;
    const uint32_t c = a.bin_count[tid];
    const uint32_t nch = (c + kBinChunk - 1) / kBinChunk;
    uint32_t total_chunks;
    const uint32_t cpre = block_excl_scan<BLOCK>(nch, scratch, &total_chunks);
    __shared__ uint32_t s_first, s_n;
    if ((uint32_t)tid == bin) {
        s_first = cpre;
        s_n = nch;
        atomicMax(&a.counters->max_bin, c);  // the fullest bin: the host picks the depth-order path of later frames by it
    }
    __syncthreads();
    const uint32_t first = min(s_first, a.max_chunks), n = min(s_n, a.max_chunks - first);
    const uint32_t ox = (bin % a.bins_x) << a.shift, oy = (bin / a.bins_x) << a.shift;
    // 64-tile bins: four threads per tile, each sweeping a quarter of the bin's chunks (sum, then prefix);
    // larger bins: one thread per tile (strided), one sweep
    const int P = SS == 64 ? 4 : 1;
    for (int t0 = 0; t0 < SS; t0 += BLOCK / P) {
        const int t = t0 + (P == 4 ? (tid & 63) : tid);
        const int seg = P == 4 ? (tid >> 6) : 0;
        const bool on = t < SS;
        const uint32_t q0 = (uint32_t)((uint64_t)n * seg / P), q1 = (uint32_t)((uint64_t)n * (seg + 1) / P);
        uint32_t* base = a.chunk_hist + (size_t)first * SS + (on ? t : 0);
        uint32_t seg_sum = 0;
        if (P == 4) {
            if (on) {
                uint32_t q = q0;
                for (; q + 8 <= q1; q += 8) {
                    uint32_t v[8];
#pragma unroll
                    for (int k = 0; k < 8; ++k) v[k] = base[(size_t)(q + k) * SS];
#pragma unroll
                    for (int k = 0; k < 8; ++k) seg_sum += v[k];
                }
                for (; q < q1; ++q) seg_sum += base[(size_t)q * SS];
            }
            s_seg[seg][tid & 63] = seg_sum;
            __syncthreads();
        }
        uint32_t running = 0;
        if (P == 4)
            for (int k = 0; k < seg; ++k) running += s_seg[k][tid & 63];
        if (on) {
            uint32_t q = q0;
            for (; q + 8 <= q1; q += 8) {  // 8 independent loads in flight per step
                uint32_t v[8];
#pragma unroll
                for (int k = 0; k < 8; ++k) v[k] = base[(size_t)(q + k) * SS];
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    base[(size_t)(q + k) * SS] = running;
                    running += v[k];
                }
            }
            for (; q < q1; ++q) {
                const uint32_t v = base[(size_t)q * SS];
                base[(size_t)q * SS] = running;
                running += v;
            }
            if (seg == P - 1) {
                const uint32_t x = ox + (t & (S - 1)), y = oy + (t >> a.shift);
                if (x < a.tiles_x && y < a.tiles_y) a.tile_total[y * a.tiles_x + x] = running;
            }
        }
        if (P == 4) __syncthreads();
    }
}

// Single block: exclusive scan of the per-tile totals -> ranges (absent tiles stay (0,0) like the
// reference's zero-filled tileBoundaryBuffer), D -> counters.
__global__ __launch_bounds__(1024) void k_tile_scan(BinArgs a) {
    __shared__ uint32_t scratch[16];
    const uint32_t T = a.tiles_x * a.tiles_y;
    uint32_t running = 0;
    for (uint32_t base = 0; base < T; base += 4096) {
        const uint32_t i0 = base + threadIdx.x * 4;
        uint32_t v[4], sum = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            v[k] = i0 + k < T ? a.tile_total[i0 + k] : 0;
            sum += v[k];
        }
        uint32_t total;
        uint32_t excl = running + block_excl_scan<1024>(sum, scratch, &total);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if (i0 + k < T) {
                // clamped to the list's capacity: an overflowing frame is re-run, but must not read past it
                a.ranges[2 * (i0 + k)] = v[k] ? min(excl, a.capacity) : 0u;
                a.ranges[2 * (i0 + k) + 1] = v[k] ? min(excl + v[k], a.capacity) : 0u;
            }
            excl += v[k];
        }
        running += total;
    }
    if (threadIdx.x == 0) {
        a.counters->instances = running;
        if (running > a.capacity) a.counters->overflow |= 1u;
    }
}

template <int R>
__global__ __launch_bounds__(BLOCK) void k_bin_fill(BinArgs a) {
    __shared__ uint32_t s_off[256], s_cpre[256], scratch[8];
    __shared__ uint32_t s_ids[4][WAVE];
    const uint32_t total_chunks = min(bin_prepare(a, s_off, s_cpre, scratch), a.max_chunks);
    const int tid = threadIdx.x, lane = tid & (WAVE - 1), w = tid / WAVE;
    constexpr int SS = 64 * R;
    const int S = 1 << a.shift;
    // raw buffer over the list: byte offsets >= 4 * capacity are dropped by the hardware bounds check
    const __amdgpu_buffer_rsrc_t out = __builtin_amdgcn_make_buffer_rsrc(a.sorted_gid, 0, a.capacity * 4u, 0x27000);
    // the chunk loop is per wave: tell the compiler its control values are wave-uniform (scalar registers)
    const uint32_t wu = __builtin_amdgcn_readfirstlane((uint32_t)w);
    // chunk -> wave: groups of 16 consecutive chunks (1024 candidates, nearly always one bin) stay on one XCD
    // (workgroup b runs on XCD b % 8), so that the partial lines of that bin's tile lists are completed in one L2
    const uint32_t xcd = blockIdx.x % 8, slot = (blockIdx.x / 8) * 4 + wu, slots = (gridDim.x / 8) * 4;
    const uint32_t q_end = (((total_chunks + 15) / 16 + 7) / 8) * 16;
    for (uint32_t q = slot; q < q_end; q += slots) {
        const uint32_t chunk = ((q / 16) * 8 + xcd) * 16 + (q % 16);
        if (chunk >= total_chunks) continue;
        uint32_t bin, first, count;
        bin_locate(a, s_off, s_cpre, chunk, bin, first, count);
        bin = __builtin_amdgcn_readfirstlane(bin);
        first = __builtin_amdgcn_readfirstlane(first);
        count = __builtin_amdgcn_readfirstlane(count);
        uint64_t m[R];
        uint32_t g = 0;
        int lx0 = 0, ly0 = 0, lx1 = 0, ly1 = 0;
        if ((uint32_t)lane < count) {
            g = a.cand[first + lane];
            bin_local_box(a, bin, a.aabb[g], lx0, ly0, lx1, ly1);
        }
        bin_cover_masks<R>(a.shift, lx0, ly0, lx1, ly1, m);
        // where this chunk's run starts in each tile's list: range start + earlier chunks of the bin
        const uint32_t ox = (bin % a.bins_x) << a.shift, oy = (bin / a.bins_x) << a.shift;
        const uint32_t* prefix = a.chunk_hist + (size_t)chunk * SS;
        uint32_t cursor[R];
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int t = r * 64 + lane;
            const uint32_t x = ox + (t & (S - 1)), y = oy + (t >> a.shift);
            cursor[r] = (x < a.tiles_x && y < a.tiles_y) ? a.ranges[2 * (y * a.tiles_x + x)] + prefix[t] : 0u;
        }
        s_ids[w][lane] = g;
        bin_walk<R, true>(m, cursor, out, s_ids[w]);  // append, in candidate order
    }
}

// ---------------------------------------------------------------------------------------
// Depth order INSIDE the bins (the bin-local path).  The candidates of a bin arrive in Gaussian-index order (stable
// radix pass by bin over candidates emitted in index order); one workgroup per bin orders them by (depth bits, id)
// entirely in LDS: four stable 8-bit LSD passes over (key, id) pairs, no global traffic and no kernel boundary
// between the passes.  This replaces the twelve kernels of the global depth order.
// A bin with more than kBinSortMax candidates does not fit: the kernel raises overflow bit 2 and the host re-runs
// the frame on the global-depth-order path.
//
// (wave, round, lane) order is list order; the stable rank inside (wave, digit) comes from wave64 ballot matching as
// in k_radix_scatter.
// ---------------------------------------------------------------------------------------
constexpr int kBinSortThreads = 1024;
__global__ __launch_bounds__(kBinSortThreads) void k_bin_sort(const uint32_t* __restrict__ bin_count,
                                                              const uint32_t* __restrict__ ids_in,
                                                              const float* __restrict__ depth,
                                                              uint32_t* __restrict__ ids_out, Counters* counters) {
    // ONE (key, id) buffer: between the ranking and the scatter of a pass every element lives in registers (two
    // barriers apart), so a pass scatters back into the buffer it read -- 16 384 entries in 128 KiB
    __shared__ uint32_t s_key[kBinSortMax];
    __shared__ uint32_t s_id[kBinSortMax];
    __shared__ uint32_t s_wcnt[16][256];  // per-wave digit counters, then per-wave write cursors
    __shared__ uint32_t scratch[16];
    const int tid = threadIdx.x, lane = tid & (WAVE - 1), w = tid / WAVE;
    const uint32_t bin = blockIdx.x;
    uint32_t c, off;
    {
        const uint32_t v = tid < 256 ? bin_count[tid] : 0u;
        uint32_t all;
        const uint32_t excl = block_excl_scan<kBinSortThreads>(v, scratch, &all);
        __shared__ uint32_t s_c, s_o;
        if ((uint32_t)tid == bin) {
            s_c = v;
            s_o = excl;
        }
        __syncthreads();
        c = s_c;
        off = s_o;
    }
    if (c == 0) return;
    if (c > (uint32_t)kBinSortMax) {
        if (tid == 0) atomicOr(&counters->overflow, 2u);
        return;
    }
    for (uint32_t e = tid; e < c; e += kBinSortThreads) {
        const uint32_t g = ids_in[off + e];
        s_id[e] = g;
        s_key[e] = __float_as_uint(depth[g]);
    }
    constexpr int kRounds = kBinSortMax / kBinSortThreads;
    const uint64_t lt_mask = (1ull << lane) - 1ull;
    // list element e belongs to wave e / (64 * rounds), round (e / 64) % rounds, lane e % 64: all 16 waves share the work
    const int rounds = (int)((c + kBinSortThreads - 1) / kBinSortThreads);  // block-uniform, <= kRounds
    const uint32_t wbase = (uint32_t)w * (uint32_t)rounds * WAVE;
#pragma unroll 1
    for (int pass = 0; pass < 4; ++pass) {
        const int shift = pass * 8;
#pragma unroll
        for (int k = 0; k < 4; ++k) s_wcnt[(tid >> 8) * 4 + k][tid & 255] = 0;
        __syncthreads();  // also orders the previous pass's (or the load's) LDS writes before this pass's reads
        uint32_t key[kRounds], id[kRounds], rank[kRounds];
#pragma unroll
        for (int r = 0; r < kRounds; ++r) {
            key[r] = 0;
            id[r] = 0;
            rank[r] = 0;
            if (r < rounds) {
                const uint32_t e = wbase + r * WAVE + lane;
                const bool ok = e < c;
                if (ok) {
                    key[r] = s_key[e];
                    id[r] = s_id[e];
                }
                const uint32_t d = (key[r] >> shift) & 255u;
                // lanes holding a valid key with my digit: AND over the bits of (ballot(bit) XNOR my bit), in 32-bit halves
                const uint64_t okm = __ballot(ok);
                uint32_t mlo = (uint32_t)okm, mhi = (uint32_t)(okm >> 32);
#pragma unroll
                for (int bit = 0; bit < 8; ++bit) {
                    const uint32_t mine = (d >> bit) & 1u;
                    const uint64_t b = __builtin_amdgcn_ballot_w64(mine != 0);
                    const uint32_t splat = 0u - mine;
                    mlo &= ~((uint32_t)b ^ splat);
                    mhi &= ~((uint32_t)(b >> 32) ^ splat);
                }
                const uint64_t m = ((uint64_t)mhi << 32) | mlo;
                uint32_t old = 0;
                const int leader = m ? (__ffsll((unsigned long long)m) - 1) : 0;
                if (ok && lane == leader) {
                    old = s_wcnt[w][d];
                    s_wcnt[w][d] = old + (uint32_t)__popcll(m);
                }
                old = __shfl(old, leader, WAVE);
                rank[r] = old + (uint32_t)__popcll(m & lt_mask);
            }
        }
        __syncthreads();
        {   // per digit: prefix over the 16 waves, then exclusive scan over the digits -> per-wave write cursors
            uint32_t cw[16], cnt = 0;
            if (tid < 256) {
#pragma unroll
                for (int k = 0; k < 16; ++k) {
                    cw[k] = s_wcnt[k][tid];
                    cnt += cw[k];
                }
            }
            uint32_t all;
            uint32_t excl = block_excl_scan<kBinSortThreads>(cnt, scratch, &all);
            if (tid < 256) {
#pragma unroll
                for (int k = 0; k < 16; ++k) {
                    s_wcnt[k][tid] = excl;
                    excl += cw[k];
                }
            }
        }
        __syncthreads();
#pragma unroll
        for (int r = 0; r < kRounds; ++r) {
            if (r < rounds) {
                const uint32_t e = wbase + r * WAVE + lane;
                if (e < c) {
                    const uint32_t d = (key[r] >> shift) & 255u;
                    const uint32_t pos = s_wcnt[w][d] + rank[r];
                    s_key[pos] = key[r];
                    s_id[pos] = id[r];
                }
            }
        }
        __syncthreads();  // the cursors in s_wcnt are re-zeroed at the top of the next pass
    }
    for (uint32_t e = tid; e < c; e += kBinSortThreads) ids_out[off + e] = s_id[e];
}

void launch_bin_sort(const uint32_t* bin_count, const uint32_t* ids_in, const float* depth, uint32_t* ids_out,
                     Counters* counters, uint32_t bins, hipStream_t s) {
    hipLaunchKernelGGL(k_bin_sort, dim3(bins), dim3(kBinSortThreads), 0, s, bin_count, ids_in, depth, ids_out, counters);
}

static BinArgs bin_args(const BinLaunch& b) {
    BinArgs a;
    a.cand = b.cand;
    a.bin_count = b.bin_count;
    a.aabb = b.aabb;
    a.chunk_hist = b.chunk_hist;
    a.tile_total = b.tile_total;
    a.ranges = b.ranges;
    a.sorted_gid = b.sorted_gid;
    a.counters = b.counters;
    a.capacity = b.capacity;
    a.tiles_x = b.tiles_x;
    a.tiles_y = b.tiles_y;
    a.bins_x = b.bins_x;
    a.shift = b.shift;
    a.max_chunks = b.max_chunks;
    return a;
}

void launch_bin_ranges(const BinLaunch& b, hipStream_t s) {
    const BinArgs a = bin_args(b);
    if (b.shift == 3) hipLaunchKernelGGL(k_bin_count<1>, dim3(kBinGrid), dim3(BLOCK), 0, s, a);
    else if (b.shift == 4) hipLaunchKernelGGL(k_bin_count<4>, dim3(kBinGrid), dim3(BLOCK), 0, s, a);
    else hipLaunchKernelGGL(k_bin_count<16>, dim3(kBinGrid), dim3(BLOCK), 0, s, a);
    hipLaunchKernelGGL(k_bin_scan, dim3(b.bins), dim3(BLOCK), 0, s, a);
    hipLaunchKernelGGL(k_tile_scan, dim3(1), dim3(1024), 0, s, a);
}

void launch_bin_fill(const BinLaunch& b, hipStream_t s) {
    const BinArgs a = bin_args(b);
    if (b.shift == 3) hipLaunchKernelGGL(k_bin_fill<1>, dim3(kBinGrid), dim3(BLOCK), 0, s, a);
    else if (b.shift == 4) hipLaunchKernelGGL(k_bin_fill<4>, dim3(kBinGrid), dim3(BLOCK), 0, s, a);
    else hipLaunchKernelGGL(k_bin_fill<16>, dim3(kBinGrid), dim3(BLOCK), 0, s, a);
}

// ---------------------------------------------------------------------------------------
// blend.  One wave per 8x8 pixel quadrant of a 16x16 tile (4 waves = one workgroup per tile), and the
// four waves are fully independent: no workgroup barrier anywhere, so a quadrant whose pixels have
// saturated retires at once and a slow quadrant never stalls its neighbours.
//
// Each wave walks its tile's depth-sorted list in chunks of 64 entries: lane l fetches entry l's record
// (9 floats, gathered through the sorted Gaussian id; the next chunk is prefetched while the current one
// is blended), tests it against the wave's quadrant, parks it in a wave-private LDS slab, and a 64-bit
// ballot of the survivors drives a scalar loop that evaluates only those entries, in list order, with
// broadcast LDS reads.  The per-pixel body is predicated (selects) instead of branched: nested divergent
// branches cost ~40 scalar exec-mask instructions per entry and saturate the CU's scalar unit.
//
// render.comp:61-98 semantics.  Floating-point contract: the shader's expressions with the three
// multiply-adds that GLSL lets a compiler contract written as explicit FMAs (marked FMA below; the
// oracle makes the same choice), nothing reassociated.
//
// Exactness of the culling: an entry contributes to a pixel only if alpha = min(0.99, o*exp(power))
// >= 1/255, i.e. power >= -tau with tau = ln(255*o), and -power = q(d) = 0.5 d^T C d (C = conic) is
// a convex quadratic of d = uv - pixel.  If the minimum of q over the quadrant's pixel rectangle exceeds
// tau (with 0.1 % + 1e-3 slack against the rounding of exp/log, plus 8 ULP of the quadratic's largest TERMS over the
// quadrant against the cancellation error of `power` for thin diagonal splats), every pixel of the
// quadrant executes `continue` in the shader, so skipping the entry for that wave changes nothing.
// The same bound gives a per-entry lower limit on power below which exp() need not be evaluated.
// ---------------------------------------------------------------------------------------

// min over the pixel rectangle [xa,xb] x [ya,yb] of q(d) = 0.5 (c00 dx^2 + c11 dy^2) + c01 dx dy,
// d = uv - pixel.  q is convex with its minimum 0 at d = 0: inside the rectangle the answer is 0,
// otherwise the minimum lies on an edge facing the centre, where q is a 1-D parabola.
__device__ __forceinline__ float min_q_rect(float c00, float c01, float c11, float u, float v, float xa,
                                            float xb, float ya, float yb) {
    const float dx_lo = u - xb, dx_hi = u - xa, dy_lo = v - yb, dy_hi = v - ya;
    const bool in_x = !(dx_lo > 0.0f) && !(dx_hi < 0.0f), in_y = !(dy_lo > 0.0f) && !(dy_hi < 0.0f);  // NaN -> inside
    // This is a bound, not part of the pipeline's arithmetic: FMAs are welcome (the caller's slack covers rounding).
    // q(a, t) = h00 a^2 + t (h11 t + c01 a) with h = c / 2.  Only the edges that face the centre can hold the minimum
    // (a segment from the centre to a point of a far edge crosses a near edge, where the convex q is smaller): at most
    // one vertical and one horizontal edge; on an edge one coordinate is fixed and the other is the parabola's
    // minimiser r * fixed, clamped to the edge.
    const float h00 = 0.5f * c00, h11 = 0.5f * c11;
    const float r11 = -c01 * __builtin_amdgcn_rcpf(c11), r00 = -c01 * __builtin_amdgcn_rcpf(c00);
    const float a = dx_lo > 0.0f ? dx_lo : dx_hi;  // the vertical edge nearer to the centre: dx fixed, parabola in dy
    const float t = fminf(fmaxf(r11 * a, dy_lo), dy_hi);
    const float qv = __builtin_fmaf(t, __builtin_fmaf(h11, t, c01 * a), h00 * a * a);
    const float b = dy_lo > 0.0f ? dy_lo : dy_hi;  // the horizontal edge nearer to the centre
    const float s = fminf(fmaxf(r00 * b, dx_lo), dx_hi);
    const float qh = __builtin_fmaf(s, __builtin_fmaf(h00, s, c01 * b), h11 * b * b);
    // any point of an edge bounds the true minimum from above: an inexact rcp only loosens the test
    return in_x ? (in_y ? 0.0f : qh) : (in_y ? qv : fminf(qv, qh));
}

#ifdef GS_BLEND_STATS
// debug instrumentation (separate build, never the shipped library)
__device__ unsigned long long g_blend_stats[8];
#define STAT_ADD(i, v) do { const unsigned long long v_ = (unsigned long long)(v); const bool first_ = (__ffsll((unsigned long long)__ballot(true)) - 1) == lane; if (first_) atomicAdd(&g_blend_stats[i], v_); } while (0)
#else
#define STAT_ADD(i, v) do { } while (0)
#endif

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

struct BlendEntry {
    float4 co;  // c00 c01 c11 opacity
    float4 uv;  // u v r g
    float b;
};

__device__ __forceinline__ void blend_fetch(BlendEntry& e, uint32_t g, const float4* __restrict__ conic_op,
                                            const float4* __restrict__ uv_rg, const float* __restrict__ bch) {
    e.co = conic_op[g];
    e.uv = uv_rg[g];
    e.b = bch[g];
}

__global__ __launch_bounds__(BLOCK) void k_blend(const uint2* __restrict__ ranges,
                                                 const uint32_t* __restrict__ sorted_gid,
                                                 const uint32_t* __restrict__ tile_order,
                                                 const float4* __restrict__ conic_op,
                                                 const float4* __restrict__ uv_rg, const float* __restrict__ bch,
                                                 uint32_t width, uint32_t height, uint32_t tiles_x,
                                                 float4* __restrict__ rgba, uchar4* __restrict__ bgra,
                                                 const Counters* __restrict__ counters, Counters* host_counters) {
    // wave-private slabs (no cross-wave sharing, no barriers), three planes of 64 float4 per wave: {c00 c01 c11 o} {u v r g} {b, pmin, -, -}.  Plane-major keeps the staging
    // ds_write_b128 conflict-free (lane stride 16 B); one scalar-derived address + constant offsets serve the reads
    __shared__ float4 s_rec[4][3][WAVE];

    const int tid = threadIdx.x, lane = tid & (WAVE - 1), w = tid / WAVE;
    // last kernel of the frame: hand V, D, E1 and the overflow flag to the host (pinned memory; visible to it once
    // the frame's completion event, which carries the system-scope release, has fired) -- no copy node in the stream
    if (host_counters && blockIdx.x == 0 && tid == 0) *host_counters = *counters;
    // XCD-aware, load-balanced tile order: a host-built table (gs_capi.cpp, ensure_tile_order)
    const uint32_t tile = tile_order[blockIdx.x];
    const uint32_t tile_x = tile % tiles_x, tile_y = tile / tiles_x;
    const uint32_t qx0 = tile_x * kTile + (w & 1) * 8, qy0 = tile_y * kTile + (w >> 1) * 8;
    const uint32_t px = qx0 + (lane & 7), py = qy0 + (lane >> 3);
    const bool inside = px < width && py < height;  // render.comp:36-39
    const float fx = (float)px, fy = (float)py;
    const float rx0 = (float)qx0, ry0 = (float)qy0;

    const uint2 range = ranges[tile];
    float T = 1.0f, c0 = 0.0f, c1 = 0.0f, c2 = 0.0f;
    // Pixel predicates live in 64-bit scalar masks (one bit per lane): combining them is scalar-unit work
    // and testing "any lane" is one s_cmp, where bool-typed code would spend VALU instructions on it.
    uint64_t alive = __builtin_amdgcn_ballot_w64(inside);  // pixels still accumulating

    if (alive != 0 && range.x < range.y) {
        // software pipeline over 64-entry chunks: ids two chunks ahead, records one chunk ahead
        BlendEntry nxt;
        nxt.co = make_float4(0, 0, 0, 0);
        nxt.uv = make_float4(0, 0, 0, 0);
        nxt.b = 0;
        uint32_t g_next = 0;
        {
            const uint32_t i0 = range.x + lane;
            if (i0 < range.y) blend_fetch(nxt, sorted_gid[i0], conic_op, uv_rg, bch);
            const uint32_t i1 = i0 + WAVE;
            if (i1 < range.y) g_next = sorted_gid[i1];
        }
        for (uint32_t base = range.x; base < range.y; base += WAVE) {
            const BlendEntry cur = nxt;
            const bool have = base + lane < range.y;
            {   // prefetch: records of chunk +1 (ids already here), ids of chunk +2
                const uint32_t i1 = base + WAVE + lane;
                if (i1 < range.y) blend_fetch(nxt, g_next, conic_op, uv_rg, bch);
                const uint32_t i2 = i1 + WAVE;
                if (i2 < range.y) g_next = sorted_gid[i2];
            }
            // classify entry `lane` of this chunk against the wave's quadrant
            const float tau = __logf(255.0f * cur.co.w);
            // NaN opacity: min(0.99, NaN) is 0.99 in the pipeline's definition -> the entry is never culled
            const float lim = tau == tau ? fmaxf(tau, 0.0f) * 1.001f + 1e-3f : 3.0e38f;
            bool keep = have && !(tau <= -1e-3f);  // tau <= 0: o*exp(p) < 1/255 for every p <= 0
            if (keep) {
                const float mq = min_q_rect(cur.co.x, cur.co.y, cur.co.z, cur.uv.x, cur.uv.y, rx0, rx0 + 7.0f,
                                            ry0, ry0 + 7.0f);
                // The rounding error of the shader's `power` (and of mq) is relative to the TERMS c00 dx^2, c11 dy^2,
                // c01 dx dy, not to their sum: a thin diagonal splat far from its centre has terms ~1e5 cancelling to
                // q ~ 5.  The slack therefore grows with the terms at the quadrant's corner farthest from the centre
                // (8 roundings of 2^-24 each, generously).
                const float ax = fmaxf(fabsf(cur.uv.x - rx0), fabsf(cur.uv.x - (rx0 + 7.0f)));
                const float ay = fmaxf(fabsf(cur.uv.y - ry0), fabsf(cur.uv.y - (ry0 + 7.0f)));
                const float mag = __builtin_fmaf(0.5f * fabsf(cur.co.x) * ax, ax,
                                                 __builtin_fmaf(0.5f * fabsf(cur.co.z) * ay, ay, fabsf(cur.co.y) * ax * ay));
                keep = !(mq > __builtin_fmaf(mag, 4.8e-7f, lim));  // NaN -> keep
            }
            uint64_t bm = __ballot(keep);
            STAT_ADD(0, 1);
            STAT_ADD(6, __popcll(__ballot(have)));
            STAT_ADD(1, __popcll(bm));
            if (bm == 0) continue;
            s_rec[w][0][lane] = cur.co;
            s_rec[w][1][lane] = cur.uv;
            s_rec[w][2][lane] = make_float4(cur.b, -lim, 0.0f, 0.0f);

            while (bm) {
                const int k = __ffsll((unsigned long long)bm) - 1;
                bm &= bm - 1;
                STAT_ADD(2, 1);                       // (entry, wave) pairs evaluated
                STAT_ADD(3, __popcll(alive));         // lanes alive
                float4 co = s_rec[w][0][k];
                float4 uv = s_rec[w][1][k];
                float4 bp = s_rec[w][2][k];
                // all ten floats in one LDS round trip: without this the compiler sinks the loads of o, r, g, b
                // behind the exp() branch, where 94 % of the pairs then pay a second LDS latency
                asm volatile("" : "+v"(co.w), "+v"(uv.z), "+v"(uv.w), "+v"(bp.x));
                const float dx = uv.x - fx;
                const float dy = uv.y - fy;
                // :66  -0.5 * (co.x*dx*dx + co.z*dy*dy) - co.y*dx*dy
                const float s = __builtin_fmaf(co.z * dy, dy, co.x * dx * dx);        // FMA
                const float power = __builtin_fmaf(-(co.y * dx), dy, -0.5f * s);      // FMA
                // power <= 0 is false for NaN: a NaN power skips the entry (the pipeline's definition)
                const uint64_t m1 = alive & __builtin_amdgcn_ballot_w64(power <= 0.0f) &
                                    __builtin_amdgcn_ballot_w64(!(power < bp.y));
                if (m1 != 0) {
                    STAT_ADD(4, 1);                   // pairs reaching exp
                    STAT_ADD(5, __popcll(m1));        // lanes needing exp
                    const float alpha = fminf(0.99f, co.w * gs_exp_blend(power));  // :77
                    const uint64_t m2 = m1 & __builtin_amdgcn_ballot_w64(!(alpha < 1.0f / 255.0f));
                    const float test_T = T * (1 - alpha);
                    const uint64_t mk = m2 & __builtin_amdgcn_ballot_w64(test_T < 0.0001f);  // :82-85 break
                    const bool upd = __builtin_amdgcn_inverse_ballot_w64(m2 & ~mk);
                    // lanes that do not take this entry add rgb * 0 * T: exactly nothing
                    const float a_eff = upd ? alpha : 0.0f;
                    c0 = __builtin_fmaf(uv.z * a_eff, T, c0);  // :87  FMA
                    c1 = __builtin_fmaf(uv.w * a_eff, T, c1);
                    c2 = __builtin_fmaf(bp.x * a_eff, T, c2);
                    T = upd ? test_T : T;
                    alive &= ~mk;
                    if (alive == 0) bm = 0;
                }
            }
            if (alive == 0) break;
        }
    }
    if (inside) {
        const size_t p = (size_t)py * width + px;
        if (rgba) rgba[p] = make_float4(c0, c1, c2, 1.0f);  // :98
        if (bgra) {
            // imageStore to B8G8R8A8_UNORM: clamp to [0,1], round to nearest
            const float r = fminf(fmaxf(c0, 0.0f), 1.0f), g = fminf(fmaxf(c1, 0.0f), 1.0f),
                        b = fminf(fmaxf(c2, 0.0f), 1.0f);
            bgra[p] = make_uchar4((unsigned char)(int)__builtin_rintf(b * 255.0f),
                                  (unsigned char)(int)__builtin_rintf(g * 255.0f),
                                  (unsigned char)(int)__builtin_rintf(r * 255.0f), 255);
        }
    }
}

void launch_blend(const uint32_t* ranges, const uint32_t* sorted_gid, const uint32_t* tile_order, const AttrView& av,
                  uint32_t width,
                  uint32_t height, float* rgba, uint8_t* bgra, const Counters* counters,
                  Counters* host_counters, hipStream_t s) {
    if (width == 0 || height == 0) return;
    const uint32_t tx = (width + kTile - 1) / kTile, ty = (height + kTile - 1) / kTile;
    hipLaunchKernelGGL(k_blend, dim3(tx * ty), dim3(BLOCK), 0, s, reinterpret_cast<const uint2*>(ranges),
                       sorted_gid, tile_order, av.conic_op, av.uv_rg, av.b, width, height, tx,
                       reinterpret_cast<float4*>(rgba), reinterpret_cast<uchar4*>(bgra), counters, host_counters);
}

#ifdef GS_BLEND_STATS
extern "C" int gs_debug_blend_stats(unsigned long long* out, int reset) {
    unsigned long long z[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_blend_stats), sizeof z) != hipSuccess) return -1;
    if (reset && hipMemcpyToSymbol(HIP_SYMBOL(g_blend_stats), z, sizeof z) != hipSuccess) return -1;
    return 0;
}
#endif

}  // namespace gs
