// Micro-benchmark: the ACCESS PATTERN of k_preprocess without its arithmetic -- how fast can gfx950 move exactly
// the bytes that kernel moves, the way it moves them?
//   per Gaussian (N):        pos 3 + cov3D 6 planar floats, coalesced                                               36 B in
//   per VISIBLE Gaussian (V): opacity + alpha cut (planar, 8 B in); the 192-byte SH block, fetched wave-cooperatively with
//                            LDS-DMA (global_load_lds_dwordx4, twelve adjacent lanes per Gaussian's three lines) and read
//                            back from LDS; one 64-byte record written as a whole line by four lanes, AT THE GAUSSIAN'S SCENE
//                            ID (the kernel reads the Morton-ordered copy and writes by scene id: scattered lines); one 16-byte
//                            entry appended to one of 256 dense lists (one atomic per wave)                          200 B in, 80 B out
// Visibility comes from a precomputed byte per Gaussian, constant over runs of R consecutive Gaussians (R = 1: every lane on
// its own; 64: whole waves; 4096: what a frustum does to a Morton-ordered scene).  No projection, no SH evaluation: the loaded
// values are summed into the record so that nothing is dead.
//   hipcc --offload-arch=gfx950 -O3 preprocess_pattern.hip -o preprocess_pattern && ./preprocess_pattern [N] [visible fraction]
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <numeric>
#include <random>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); std::exit(1); } } while (0)

constexpr int BLOCK = 256, WAVE = 64, kRegions = 256, kCounterStride = 32, kHalf = 32;  // (32 Gaussians' SH per LDS round: 6 KiB per wave)

struct Args {
    const float* planes;     // [9][N]: pos, cov3D
    const float* opac_cut;   // [2][N]
    const float* sh;         // [N][48]
    const uint32_t* perm;    // [N]: scene id of Gaussian i of the (Morton-ordered) copy
    const uint8_t* visible;  // [N]
    float4* rec;             // [N] 64-byte records (4 x float4)
    uint4* list;             // [kRegions][slots]
    uint32_t* list_count;    // [kRegions * kCounterStride]
    uint32_t n, slots;
};

__global__ __launch_bounds__(BLOCK) void k_pattern(Args a) {
    __shared__ float4 s_stage[BLOCK / WAVE][kHalf * 12 + 8];  // SH blocks of up to 32 Gaussians, or the three record planes (aliased)
    const uint32_t i = blockIdx.x * BLOCK + threadIdx.x;
    const int lane = threadIdx.x & (WAVE - 1);
    float4* const stage = s_stage[threadIdx.x / WAVE];
    const bool valid = i < a.n;
    float acc = 0.0f;
    if (valid) {
#pragma unroll
        for (int p = 0; p < 9; ++p) acc += a.planes[(size_t)p * a.n + i];
    }
    const bool vis = valid && a.visible[i] != 0;
    const uint64_t vm = __ballot(vis);
    const uint32_t n_vis = (uint32_t)__popcll(vm);
    const uint32_t rank = (uint32_t)__popcll(vm & ((1ull << lane) - 1ull));
    uint32_t base = 0;
    const uint32_t region = blockIdx.x % kRegions;
    if (lane == 0 && vm != 0) base = region * a.slots + atomicAdd(a.list_count + region * kCounterStride, n_vis);
    float opacity = 0.0f, cut = 0.0f;
    if (vis) {
        opacity = a.opac_cut[i];
        cut = a.opac_cut[(size_t)a.n + i];
    }
    // ---- the SH blocks of the visible Gaussians through LDS-DMA, as gs_preprocess.hip does it
    uint8_t* const s_src = reinterpret_cast<uint8_t*>(stage + kHalf * 12);
    if (vis) s_src[rank] = (uint8_t)lane;
    __builtin_amdgcn_wave_barrier();
    const uint32_t slot = lane / 12u, q = lane % 12u;
    const char* const sh_bytes = reinterpret_cast<const char*>(a.sh);
    const uint32_t lds_base = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(uintptr_t)stage);
    float shsum = 0.0f;
    for (uint32_t half = 0; half * kHalf < n_vis; ++half) {
        const uint32_t first = half * kHalf, last = min(n_vis, first + (uint32_t)kHalf);
        uint32_t dst = lds_base;
        for (uint32_t g0 = first; g0 < last; g0 += 5u, dst += 5u * 192u) {
            const uint32_t g = g0 + slot;
            if (lane < 60 && g < last) {
                const uint32_t src_lane = s_src[g];
                const char* gsrc = sh_bytes + (size_t)(i - lane + src_lane) * 192u + q * 16u;
                unsigned keep;
                asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                             : "=&s"(keep) : "v"(gsrc), "s"(dst) : "memory");
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_wave_barrier();
        if (vis && rank >= first && rank < last) {
            const float4* blk = stage + (size_t)(rank - first) * 12;
#pragma unroll
            for (int k = 0; k < 12; ++k) {
                const float4 t = blk[k];
                shsum += (t.x + t.y) + (t.z + t.w);
            }
        }
        __builtin_amdgcn_wave_barrier();
    }
    // ---- the list entry and the record (whole 64-byte lines, four lanes each, at the scene id)
    const uint32_t oid = valid ? a.perm[i] : 0u;
    const uint32_t b = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
    if (vis) a.list[b + rank] = make_uint4(oid, __float_as_uint(acc), (uint32_t)lane, rank);
    constexpr int kPlane = 64;
    if (vis) {
        stage[0 * kPlane + lane] = make_float4(acc, shsum, opacity, cut);
        stage[1 * kPlane + lane] = make_float4(acc, acc, shsum, shsum);
        stage[2 * kPlane + lane] = make_float4(shsum, acc, opacity, cut);
    }
    uint32_t* const s_oid = reinterpret_cast<uint32_t*>(stage + 3 * kPlane);
    s_oid[lane] = oid;
    __builtin_amdgcn_wave_barrier();
    const uint32_t c = lane & 3u;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const uint32_t r = (uint32_t)t * 16u + (lane >> 2);
        if ((vm >> r) & 1ull) {
            const float4 val = c < 3u ? stage[c * kPlane + r] : make_float4(0, 0, 0, 0);
            a.rec[(size_t)s_oid[r] * 4 + c] = val;
        }
    }
}

// plain streaming copy of the same number of bytes, for the "achievable" figure on this box
__global__ __launch_bounds__(BLOCK) void k_copy(const float4* __restrict__ in, float4* __restrict__ out, size_t n4) {
    for (size_t i = (size_t)blockIdx.x * BLOCK + threadIdx.x; i < n4; i += (size_t)gridDim.x * BLOCK) out[i] = in[i];
}

int main(int argc, char** argv) {
    const uint32_t n = argc > 1 ? (uint32_t)std::atoll(argv[1]) : 6000000u;
    const double frac = argc > 2 ? std::atof(argv[2]) : 0.4764;  // S(6e6) at 1080p: V / N
    CHECK(hipSetDevice(0));
    float *planes, *opac, *sh;
    uint32_t *perm, *counts;
    uint8_t* visible;
    float4* rec;
    uint4* list;
    const uint32_t slots = ((n / kRegions) * 2 + 4096) & ~1023u;
    CHECK(hipMalloc(&planes, (size_t)9 * n * 4));
    CHECK(hipMalloc(&opac, (size_t)2 * n * 4));
    CHECK(hipMalloc(&sh, (size_t)48 * n * 4));
    CHECK(hipMalloc(&perm, (size_t)n * 4));
    CHECK(hipMalloc(&visible, n));
    CHECK(hipMalloc(&rec, (size_t)n * 64));
    CHECK(hipMalloc(&list, (size_t)kRegions * slots * 16));
    CHECK(hipMalloc(&counts, kRegions * kCounterStride * 4));
    CHECK(hipMemset(planes, 0, (size_t)9 * n * 4));
    CHECK(hipMemset(opac, 0, (size_t)2 * n * 4));
    CHECK(hipMemset(sh, 0, (size_t)48 * n * 4));
    std::mt19937 rng(1);
    std::vector<uint32_t> h_perm(n);
    std::iota(h_perm.begin(), h_perm.end(), 0u);
    std::shuffle(h_perm.begin(), h_perm.end(), rng);  // the Morton order of a random scene is a random permutation of its ids
    CHECK(hipMemcpy(perm, h_perm.data(), (size_t)n * 4, hipMemcpyHostToDevice));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    Args a{planes, opac, sh, perm, visible, rec, list, counts, n, slots};
    std::printf("preprocess access pattern, N = %u, visible fraction %.4f (gfx950; bytes = 36 N + 248 V: bench.py's algorithmic bytes of k_preprocess without the N-wide planes; moved: 36 N + 280 V)\n", n, frac);
    for (uint32_t run : {1u, 64u, 4096u, 65536u}) {
        std::vector<uint8_t> h_vis(n);
        std::uniform_real_distribution<double> u01(0.0, 1.0);
        uint64_t v = 0;
        for (uint32_t s = 0; s < n; s += run) {
            const uint8_t on = u01(rng) < frac ? 1 : 0;
            for (uint32_t k = s; k < std::min(n, s + run); ++k) h_vis[k] = on, v += on;
        }
        CHECK(hipMemcpy(visible, h_vis.data(), n, hipMemcpyHostToDevice));
        float best = 1e30f, sum = 0;
        const int reps = 12;
        for (int it = 0; it < reps + 2; ++it) {
            CHECK(hipMemsetAsync(counts, 0, kRegions * kCounterStride * 4, nullptr));
            CHECK(hipEventRecord(e0, nullptr));
            hipLaunchKernelGGL(k_pattern, dim3((n + BLOCK - 1) / BLOCK), dim3(BLOCK), 0, nullptr, a);
            CHECK(hipEventRecord(e1, nullptr));
            CHECK(hipEventSynchronize(e1));
            float ms;
            CHECK(hipEventElapsedTime(&ms, e0, e1));
            if (it >= 2) best = std::min(best, ms), sum += ms;
        }
        const double bytes = 36.0 * n + 248.0 * (double)v;  // bench.py's accounting of k_preprocess: pos 12 + cov3D 24 per Gaussian; opacity 4 + SH 192 + 52 B of attributes out per visible one
        std::printf("  visibility in runs of %6u: V = %9llu  mean %.1f us  best %.1f us  -> %.2f TB/s (mean), %.2f TB/s (best)\n", run, (unsigned long long)v,
                    1e3 * sum / reps, 1e3 * best, bytes / (1e-3 * sum / reps) / 1e12, bytes / (1e-3 * best) / 1e12);
    }
    {   // a streaming copy of 1 GB for scale
        const size_t n4 = (size_t)1 << 26;
        float4 *in, *out;
        CHECK(hipMalloc(&in, n4 * 16));
        CHECK(hipMalloc(&out, n4 * 16));
        CHECK(hipMemset(in, 0, n4 * 16));
        float best = 1e30f;
        for (int it = 0; it < 8; ++it) {
            CHECK(hipEventRecord(e0, nullptr));
            hipLaunchKernelGGL(k_copy, dim3(256 * 16), dim3(BLOCK), 0, nullptr, in, out, n4);
            CHECK(hipEventRecord(e1, nullptr));
            CHECK(hipEventSynchronize(e1));
            float ms;
            CHECK(hipEventElapsedTime(&ms, e0, e1));
            if (it >= 2) best = std::min(best, ms);
        }
        std::printf("  streaming copy, 1 GiB in + 1 GiB out: best %.1f us -> %.2f TB/s\n", 1e3 * best, 2.0 * n4 * 16 / (1e-3 * best) / 1e12);
    }
    return 0;
}
