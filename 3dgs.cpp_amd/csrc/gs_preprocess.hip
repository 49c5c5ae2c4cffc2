// gs_preprocess.hip -- k_preprocess: cull, EWA conic, radius, tile box, SH -> RGB; one lane per Gaussian.
//
// Part of libgs3d_hip.so (gfx950 only).  Built with -ffp-contract=off: the floating-point contract of this path is "IEEE
// binary32, one rounding per operation, in the order the reference shader writes it" (DESIGN.md section 3); fused
// multiply-adds appear only where written explicitly.
// Reference restated (paths relative to /root/reference/src/shaders): preprocess.comp:34-183
#include "gs_device.h"

namespace gs {

#ifndef GS_PRE_SH_LDS
#define GS_PRE_SH_LDS 1  // k_preprocess fetches the SH blocks of a wave's visible Gaussians with LDS-DMA, whole lines at a time
#endif

// ---------------------------------------------------------------------------------------
// preprocess.  One thread per Gaussian; position / cov3D / opacity are SoA planes (coalesced 256 B per
// wave and plane); the SH block is AoS (48 contiguous floats) and is read only by lanes that survive
// every cull; the 64-byte attribute records leave the wave through LDS, four lanes writing one whole line.
// ---------------------------------------------------------------------------------------
constexpr float SH_C0 = 0.28209479177387814f;  // common.glsl:16-33
constexpr float SH_C1 = 0.4886025119029199f;

__device__ __forceinline__ float ndc2pix(float v, int S) { return ((v + 1.0f) * (float)S - 1.0f) * 0.5f; }

struct PreUniforms {
    gs_uniforms u;
    Counters* counters;      // nullable
    const FrameParams* fp;   // nullable: the uniforms live there (graph replay)
    uint64_t* stamps;        // nullable: the frame's timeline
};

// Wave-private LDS of k_preprocess: the attribute records of a wave's 64 Gaussians on their way to HBM, three planes of
// 64 float4 with a plane stride of 68 (272 dwords = 16 mod 64: the cooperative reads of 16 consecutive lanes cover all
// 64 banks once).
constexpr int kPrePlane = 68, kPreStage = 3 * kPrePlane + 16;  // + 64 scene ids (spatial order: where the records go)
#if GS_PRE_SH_LDS
// ... and, ahead of that (the two uses alias: the SH blocks are consumed before the records are staged), the SH blocks of up
// to 32 of the wave's visible Gaussians, fetched by LDS-DMA: 32 x 192 B = 384 float4 (+ 64 source-lane bytes)
#ifndef GS_PRE_SH_HALF
#define GS_PRE_SH_HALF 32
#endif
constexpr int kPreShHalf = GS_PRE_SH_HALF;
constexpr int kPreWaveLds = kPreShHalf * 12 + 4;  // float4 units; >= kPreStage
static_assert(kPreWaveLds >= kPreStage, "the record stage must fit the wave's LDS slab");
#else
constexpr int kPreWaveLds = kPreStage;
#endif

// preprocess.comp:73-108 compute_sh (degree 3 always; only .x clamped), the channel's terms accumulated in the shader's order.
// SH(j, k) = coefficient j of channel k.  Written coefficient-major (all three channels take term j before any takes term
// j + 1): per channel the sequence of operations is the shader's, and a source that lives in LDS can be consumed as it is read
// (FENCE: a compiler barrier every few terms, so that the 48 reads are not all hoisted into registers at once).
template <bool FENCE, class SH>
__device__ __forceinline__ void sh_to_rgb(const SH& S, float x, float y, float z, float (&rgb)[3]) {
    const float C2_0 = 1.0925484305920792f, C2_1 = -1.0925484305920792f, C2_2 = 0.31539156525252005f,
                C2_3 = -1.0925484305920792f, C2_4 = 0.5462742152960396f;
    const float C3_0 = -0.5900435899266435f, C3_1 = 2.890611442640554f, C3_2 = -0.4570457994644658f,
                C3_3 = 0.3731763325901154f, C3_4 = -0.4570457994644658f, C3_5 = 1.445305721320277f,
                C3_6 = -0.5900435899266435f;
    float c[3];
#define GS_SH_FENCE() do { if (FENCE) asm volatile("" : "+v"(c[0]), "+v"(c[1]), "+v"(c[2]) :: "memory"); } while (0)
#pragma unroll
    for (int k = 0; k < 3; ++k) c[k] = SH_C0 * S(0, k);
#pragma unroll
    for (int k = 0; k < 3; ++k) c[k] -= SH_C1 * S(1, k) * y;
#pragma unroll
    for (int k = 0; k < 3; ++k) c[k] += SH_C1 * S(2, k) * z;
#pragma unroll
    for (int k = 0; k < 3; ++k) c[k] -= SH_C1 * S(3, k) * x;
    GS_SH_FENCE();
#pragma unroll
    for (int k = 0; k < 3; ++k) c[k] += C2_0 * S(4, k) * x * y;
#pragma unroll
    for (int k = 0; k < 3; ++k) c[k] += C2_1 * S(5, k) * y * z;
#pragma unroll
    for (int k = 0; k < 3; ++k) c[k] += C2_2 * S(6, k) * (2.0f * z * z - x * x - y * y);
#pragma unroll
    for (int k = 0; k < 3; ++k) c[k] += C2_3 * S(7, k) * z * x;
    GS_SH_FENCE();
#pragma unroll
    for (int k = 0; k < 3; ++k) c[k] += C2_4 * S(8, k) * (x * x - y * y);
#pragma unroll
    for (int k = 0; k < 3; ++k) c[k] += C3_0 * S(9, k) * (3.0f * x * x - y * y) * y;
#pragma unroll
    for (int k = 0; k < 3; ++k) c[k] += C3_1 * S(10, k) * x * y * z;
#pragma unroll
    for (int k = 0; k < 3; ++k) c[k] += C3_2 * S(11, k) * (4.0f * z * z - x * x - y * y) * y;
    GS_SH_FENCE();
#pragma unroll
    for (int k = 0; k < 3; ++k) c[k] += C3_3 * S(12, k) * z * (2.0f * z * z - 3.0f * x * x - 3.0f * y * y);
#pragma unroll
    for (int k = 0; k < 3; ++k) c[k] += C3_4 * S(13, k) * x * (4.0f * z * z - x * x - y * y);
#pragma unroll
    for (int k = 0; k < 3; ++k) c[k] += C3_5 * S(14, k) * (x * x - y * y) * z;
#pragma unroll
    for (int k = 0; k < 3; ++k) c[k] += C3_6 * S(15, k) * x * (x * x - 3.0f * y * y);
#pragma unroll
    for (int k = 0; k < 3; ++k) rgb[k] = c[k] + 0.5f;
#undef GS_SH_FENCE
    if (rgb[0] < 0.0f) rgb[0] = 0.0f;
}
struct ShFromRegs {
    float v[48];
    __device__ __forceinline__ float operator()(int j, int k) const { return v[j * 3 + k]; }
};
struct ShFromLds {
    const float* p;
    __device__ __forceinline__ float operator()(int j, int k) const { return p[j * 3 + k]; }
};
struct ShFromLds16 {
    const uint16_t* p;
    __device__ __forceinline__ float operator()(int j, int k) const { return __half2float(__ushort_as_half(p[j * 3 + k])); }
};

// One Gaussian per lane; `valid` = the lane has one (the last wave of the grid is ragged: every lane takes part in the
// wave-cooperative parts).  stage: this wave's kPreStage float4 of LDS.
__device__ __forceinline__ void preprocess_one(const SceneView& sv, const gs_uniforms& u, const AttrView& av, uint32_t i,
                                               bool valid, float4* __restrict__ stage) {
    const size_t N = sv.stride, NC = sv.n;
    const float* __restrict__ blob = sv.blob;
    const uint32_t lane = threadIdx.x & (WAVE - 1);
    // i: the Gaussian's place in the arrays this kernel READS; oid: its id in the scene, under which everything is WRITTEN
    const uint32_t oid = sv.perm ? (valid ? sv.perm[i] : 0u) : i;

    const int tile_w = (int)((u.width + kTile - 1) / kTile);
    const int tile_h = (int)((u.height + kTile - 1) / kTile);

    // what a visible lane carries from the culls to the stores
    uint32_t num_tiles = 0;
    float px = 0, py = 0, pz = 0, depth = 0, c00 = 0, c01 = 0, c11 = 0, opacity = 0, acut = 0, radii = 0, uvx = 0, uvy = 0;
    int bx0 = 0, by0 = 0, bx1 = 0, by1 = 0;
    if (valid) do {
        px = blob[(P_POS + 0) * N + i];
        py = blob[(P_POS + 1) * N + i];
        pz = blob[(P_POS + 2) * N + i];
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
        c00 = m11 * inv_det;
        c01 = -m01 * inv_det;
        c11 = m00 * inv_det;

        const float mid = 0.5f * (m00 + m11);  // :148-152
        const float sq = sqrtf(fmaxf(0.1f, mid * mid - det));
        const float lambda1 = mid + sq;
        const float lambda2 = mid - sq;
        const float lambda = fmaxf(lambda1, lambda2);
        radii = ceilf(3.0f * sqrtf(lambda));

        uvx = ndc2pix(ndc_x, (int)u.width);  // :158
        uvy = ndc2pix(ndc_y, (int)u.height);

        // :160-165 tile box
        bx0 = clampi(f2i_sat((uvx - radii) / kTile), 0, tile_w);
        by0 = clampi(f2i_sat((uvy - radii) / kTile), 0, tile_h);
        bx1 = clampi(f2i_sat((uvx + radii + kTile - 1) / kTile), 0, tile_w);
        by1 = clampi(f2i_sat((uvy + radii + kTile - 1) / kTile), 0, tile_h);
        const uint32_t nt = (uint32_t)(bx1 - bx0) * (uint32_t)(by1 - by0);
        if (nt == 0) break;
        depth = p_view[2];
        opacity = blob[(size_t)P_OPACITY * N + i];
        acut = sv.acut[i];  // render.comp:78 as a bound on `power` (computed at load from the opacity; the blend's cut)
        num_tiles = nt;
    } while (false);
    const bool vis = num_tiles != 0;

    // ---- the wave's run of slots in its workgroup's dense list of visible Gaussians (AttrView::vis): one atomic per wave,
    // issued here, its result first needed after the SH work below -- the round trip rides behind the SH fetch
    uint32_t vis_base = 0;
    if (av.vis) {
        const uint64_t m = __ballot(vis);
        const uint32_t region = (i / BLOCK) % kVisRegions;  // (i / BLOCK = the workgroup)
        if (lane == 0 && m != 0)
            vis_base = region * av.vis_region_slots + atomicAdd(av.vis_count + region * kVisCounterStride, (uint32_t)__popcll(m));
    }

    // ---- the SH block of the visible Gaussians: 48 contiguous floats each (192 B = three 64-byte lines); only lanes
    // that survived every cull need them, so SH traffic is 192 B per VISIBLE Gaussian.
    float rgb[3] = {0.0f, 0.0f, 0.0f};
#if GS_PRE_SH_LDS
    // Fetched wave-cooperatively with LDS-DMA (global_load_lds_dwordx4: global -> LDS without passing through VGPRs): the
    // visible lanes publish their lane numbers by rank; then lane l = 12 s + q of each instruction reads 16-byte chunk q of
    // the (5 b + s)-th visible Gaussian -- twelve adjacent lanes cover one Gaussian's three whole lines, where a lane reading
    // its own block issues twelve quarter-line requests -- and the DMA lays the chunks down lane-linearly, i.e. as the
    // blocks, back to back, in rank order.  Each visible lane then reads its own block from LDS.  Up to 32 Gaussians per
    // round (6 KiB per wave, aliased with the record stage below); a denser wave takes a second round.
    const uint64_t vm_sh = __ballot(vis);
    const uint32_t n_vis = (uint32_t)__popcll(vm_sh);
    const uint32_t my_rank = (uint32_t)__popcll(vm_sh & ((1ull << lane) - 1ull));
    uint8_t* const s_src = reinterpret_cast<uint8_t*>(stage + kPreShHalf * 12);  // [64] lane number of the r-th visible Gaussian
    if (vis) s_src[my_rank] = (uint8_t)lane;
    __builtin_amdgcn_wave_barrier();
    const int chunks = sv.sh16 ? 6 : 12;                 // 16-byte chunks per Gaussian (binary16 storage: 96 B)
    const uint32_t per_inst = sv.sh16 ? 10u : 5u;        // Gaussians per DMA instruction (60 of the 64 lanes)
    const uint32_t slot = sv.sh16 ? lane / 6u : lane / 12u, q = sv.sh16 ? lane % 6u : lane % 12u;
    const char* const sh_bytes = sv.sh16 ? reinterpret_cast<const char*>(sv.sh16) : reinterpret_cast<const char*>(blob + (size_t)P_SH * N);
    const size_t sh_stride = sv.sh16 ? 96 : 192;
    // LDS byte address of the wave's slab (the low half of a flat LDS pointer is the LDS offset), as a scalar: M0 takes it
    const uint32_t lds_base = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(uintptr_t)stage);
    for (uint32_t half = 0; half * kPreShHalf < n_vis; ++half) {
        const uint32_t first = half * kPreShHalf, last = min(n_vis, first + (uint32_t)kPreShHalf);
        uint32_t dst = lds_base;
        for (uint32_t g0 = first; g0 < last; g0 += per_inst, dst += per_inst * (uint32_t)sh_stride) {
            const uint32_t g = g0 + slot;
            if (lane < 60u && g < last) {
                const uint32_t src_lane = s_src[g];
                const char* gsrc = sh_bytes + (size_t)(i - lane + src_lane) * sh_stride + q * 16u;
                unsigned keep;
                asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                             : "=&s"(keep) : "v"(gsrc), "s"(dst) : "memory");
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_wave_barrier();
        if (vis && my_rank >= first && my_rank < last) {
            float dx = px - u.camera_position[0];
            float dy = py - u.camera_position[1];
            float dz = pz - u.camera_position[2];
            const float len = sqrtf(dx * dx + dy * dy + dz * dz);
            const float x = dx / len, y = dy / len, z = dz / len;
            const float4* blk = stage + (size_t)(my_rank - first) * chunks;
            if (sv.sh16) {  // binary16 storage, widened exactly
                sh_to_rgb<true>(ShFromLds16{reinterpret_cast<const uint16_t*>(blk)}, x, y, z, rgb);
            } else {
                sh_to_rgb<true>(ShFromLds{reinterpret_cast<const float*>(blk)}, x, y, z, rgb);
            }
            if (!av.vis) {  // (with the dense lists the entry below carries depth and box: the N-wide planes are not written at all)
                av.depth[oid] = depth;
                av.aabb[oid] = make_ushort4((unsigned short)bx0, (unsigned short)by0, (unsigned short)bx1, (unsigned short)by1);
            }
        }
        // every lane of this round has consumed its block (the values above depend on the reads): the next round's DMA, or
        // the record stage, may overwrite the slab
        __builtin_amdgcn_wave_barrier();
    }
#else
    // ---- the SH block of the visible Gaussians: 48 contiguous floats each (192 B = three 64-byte lines); only lanes
    // that survived every cull need them, so SH traffic is 192 B per VISIBLE Gaussian
    // (A wave-cooperative fetch -- twelve lanes reading the twelve 16-byte chunks of one Gaussian, three full-line requests
    // instead of twelve quarter-line ones, the chunks handed to their owner through LDS -- measured 5 us SLOWER: 106 VGPRs
    // instead of 73 while chunks and coefficients are live together, four waves per SIMD instead of six.)
    if (vis) {
        ShFromRegs sh;
        if (sv.sh16) {  // opt-in binary16 storage (gs_scene_quantize_sh): 96 B per visible Gaussian, widened exactly
            const uint4* __restrict__ shv = reinterpret_cast<const uint4*>(sv.sh16) + (size_t)i * 6;
#pragma unroll
            for (int q = 0; q < 6; ++q) {
                const uint4 t = shv[q];
                const uint32_t wds[4] = {t.x, t.y, t.z, t.w};
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    sh.v[8 * q + 2 * k + 0] = __half2float(__ushort_as_half((unsigned short)(wds[k] & 0xFFFFu)));
                    sh.v[8 * q + 2 * k + 1] = __half2float(__ushort_as_half((unsigned short)(wds[k] >> 16)));
                }
            }
        } else {
            const float4* __restrict__ shv = reinterpret_cast<const float4*>(blob + (size_t)P_SH * N) + (size_t)i * 12;
#pragma unroll
            for (int q = 0; q < 12; ++q) {
                const float4 t = shv[q];
                sh.v[4 * q + 0] = t.x;
                sh.v[4 * q + 1] = t.y;
                sh.v[4 * q + 2] = t.z;
                sh.v[4 * q + 3] = t.w;
            }
        }
        float dx = px - u.camera_position[0];
        float dy = py - u.camera_position[1];
        float dz = pz - u.camera_position[2];
        const float len = sqrtf(dx * dx + dy * dy + dz * dz);
        sh_to_rgb<false>(sh, dx / len, dy / len, dz / len, rgb);
        if (!av.vis) {
            av.depth[oid] = depth;
            av.aabb[oid] = make_ushort4((unsigned short)bx0, (unsigned short)by0, (unsigned short)bx1, (unsigned short)by1);
        }
    }
#endif
    // :128 / :176.  With the dense lists of visible Gaussians (av.vis) nothing downstream reads the N-wide planes tiles / depth /
    // aabb -- level 1 streams the lists, level 2 the candidate records -- so they are not written: a culled Gaussian writes
    // nothing at all (the stage taps rebuild the planes from the lists: launch_vis_to_planes)
    if (valid && !av.vis) av.tiles[oid] = num_tiles;

    // ---- the 64-byte-strided record of every visible Gaussian.  Wave-cooperative: the records pass through LDS and four
    // lanes write one record -- ONE 64-byte request per visible Gaussian (the last quarter as zeros) instead of three
    // 16-byte ones from its own lane: k_preprocess 39 -> 37 us (without any record store it takes 30).
    const uint64_t vm = __ballot(vis);
    if (av.vis) {  // 16 bytes per visible Gaussian, the wave's entries back to back
        const uint32_t base = (uint32_t)__builtin_amdgcn_readfirstlane((int)vis_base);
        if (vis)
            av.vis[base + (uint32_t)__popcll(vm & ((1ull << lane) - 1ull))] =
                make_uint4(oid, __float_as_uint(depth), (uint32_t)bx0 | ((uint32_t)by0 << 16), (uint32_t)bx1 | ((uint32_t)by1 << 16));
    }
    if (vis) {
        stage[0 * kPrePlane + lane] = make_float4(c00, c01, c11, opacity);
        stage[1 * kPrePlane + lane] = make_float4(uvx, uvy, rgb[0], rgb[1]);
        stage[2 * kPrePlane + lane] = make_float4(rgb[2], depth, radii, acut);
    }
    uint32_t* const s_oid = reinterpret_cast<uint32_t*>(stage + 3 * kPrePlane);  // [64] (behind the three planes; spatial order only)
    if (sv.perm) s_oid[lane] = oid;
    __builtin_amdgcn_wave_barrier();
    {
        float4* const rec0 = reinterpret_cast<float4*>(av.rec + (i - lane));  // the wave's first record (never dereferenced past n)
        const uint32_t c = lane & 3u;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const uint32_t r = (uint32_t)t * 16u + (lane >> 2);
            if ((vm >> r) & 1ull) {  // the whole line: leaving the unused quarter out (three lanes per record) measured 2 us slower
                const float4 val = c < 3u ? stage[c * kPrePlane + r] : make_float4(0, 0, 0, 0);
                float4* const dst = sv.perm ? reinterpret_cast<float4*>(av.rec + s_oid[r]) + c : rec0 + (size_t)r * 4 + c;
                *dst = val;
            }
        }
    }
}

__global__ __launch_bounds__(BLOCK) void k_preprocess(SceneView sv, PreUniforms pu, AttrView av) {
    __shared__ float4 s_stage[BLOCK / WAVE][kPreWaveLds];
    const gs_uniforms& u = pu.fp ? pu.fp->u : pu.u;  // uniform either way: scalar loads
    const uint32_t i = blockIdx.x * BLOCK + threadIdx.x;
    frame_stamp(pu.stamps, ST_PREPROCESS);
    if (i == 0 && pu.counters) {  // first kernel of the frame: the counters the later kernels accumulate into
        pu.counters->visible = 0;
        pu.counters->instances = 0;
        pu.counters->overflow = 0;
        pu.counters->bin_entries = 0;
        pu.counters->max_bin = 0;
        pu.counters->slabs = 0;
        pu.counters->blend_resolved = 0;
        pu.counters->blend_redo = 0;
        pu.counters->q_head = 0;
        pu.counters->q_bins_done = 0;
    }
    preprocess_one(sv, u, av, i, i < sv.n, s_stage[threadIdx.x / WAVE]);
}

void launch_preprocess(const SceneView& sv, const gs_uniforms& u, const AttrView& av, Counters* counters,
                       const FrameParams* fp, uint64_t* stamps, hipStream_t s) {
    if (sv.n == 0) return;
    PreUniforms pu;
    pu.u = u;
    pu.counters = counters;
    pu.fp = fp;
    pu.stamps = stamps;
    hipLaunchKernelGGL(k_preprocess, dim3((sv.n + BLOCK - 1) / BLOCK), dim3(BLOCK), 0, s, sv, pu, av);
}
}  // namespace gs
