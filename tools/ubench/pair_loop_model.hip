// Micro-benchmark (round 6): what bounds the blend's pair loop?  The loop issues ~23 VALU instructions and three BROADCAST LDS reads
// (ds_read_b128 x 2 + ds_read_b64: 40 bytes x 64 lanes of identical data) per (entry, wave) pair, eight waves per SIMD.  Measured k_blend:
// ~86 cycles per pair and SIMD at 2.33 GHz, where 23 VALU x 2 cycles = 46.  This kernel times synthetic loops of the same shape:
//   0  23 VALU, no LDS                                     3  23 VALU + the three reads under exec = lanes 0-15 (one lane group)
//   1  23 VALU + the three broadcast reads (the shipped)   4  33 VALU (10 v_readfirstlane) + one-lane-group reads: the entry in SGPRs
//   2  the three reads alone                               5  23 VALU + s_load_dwordx8 + s_load_dwordx2 of the NEXT entry (scalar cache)
// hipcc --offload-arch=gfx950 -O3 pair_loop_model.hip -o pair_loop_model && ./pair_loop_model
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

#define VALU23                                                                                                        \
    "v_sub_f32 v58, v58, %[fx]\n v_sub_f32 v59, v59, %[fy]\n v_mul_f32 v54, v54, v58\n v_mul_f32 v56, v56, v59\n"     \
    "v_mul_f32 v54, v58, v54\n v_mul_f32 v56, v59, v56\n v_mul_f32 v55, v55, v58\n v_add_f32 v54, v54, v56\n"          \
    "v_mul_f32 v55, v55, v59\n v_add_f32 v54, v55, v54\n v_cmp_le_u32 vcc, v54, v63\n v_mul_f32 v56, 0xbfb8aa3b, v54\n" \
    "v_exp_f32 v56, v56\n s_nop 0\n v_mul_f32 v56, v57, v56\n v_min_f32 v56, 0x3f7d70a4, v56\n v_sub_f32 v55, 1.0, v56\n" \
    "v_mul_f32 %[w], v56, %[T]\n v_mul_f32 %[T], %[T], v55\n v_cmp_gt_f32 vcc, %[k], %[T]\n v_cmp_eq_u32 vcc, %[T], %[k]\n" \
    "v_fmac_f32 %[c0], v60, %[w]\n v_fmac_f32 %[c1], v61, %[w]\n v_fmac_f32 %[c2], v62, %[w]\n"

// the shipped loop's body (gs_blend.hip: GS_PAIR_GUARDED), one pair per iteration; ABL selects an ablation:
//   0 as shipped   1 v_exp_f32 -> v_mul_f32   2 the exec-writing compare (v_cmpx) -> v_cmp   3 no scalar tail (exec / alive / rem bookkeeping)   4 no s_nop
#define REAL_HEAD                                                                                                                  \
    "ds_read_b128 v[54:57], %[a]\n ds_read_b128 v[58:61], %[a] offset:1024\n ds_read_b64 v[62:63], %[a] offset:2048\n s_waitcnt lgkmcnt(0)\n" \
    "v_sub_f32 v58, v58, %[fx]\n v_sub_f32 v59, v59, %[fy]\n v_mul_f32 v54, v54, v58\n v_mul_f32 v56, v56, v59\n"                  \
    "v_mul_f32 v54, v58, v54\n v_mul_f32 v56, v59, v56\n v_mul_f32 v55, v55, v58\n v_add_f32 v54, v54, v56\n"                       \
    "v_mul_f32 v55, v55, v59\n v_add_f32 v54, v55, v54\n"
template <int ABL>
__device__ __forceinline__ void real_pair(uint32_t addr, uint64_t& alive, uint32_t& rem, float fx, float fy, float& T, float& c0, float& c1, float& c2) {
    float w;
    uint64_t mk;
    const float k1e4 = 1e-4f;
    const uint32_t slice = 0x38D1u;
    asm volatile(REAL_HEAD
                 "%=:\n"
                 : : [a] "v"(addr), [fx] "v"(fx), [fy] "v"(fy) : "memory", "v54", "v55", "v56", "v57", "v58", "v59", "v60", "v61", "v62", "v63");
    if (ABL == 2) asm volatile("v_cmp_le_u32 vcc, v54, v63" ::: "vcc");
    else asm volatile("v_cmpx_le_u32 vcc, v54, v63" ::: "vcc", "exec");
    if (ABL == 1) asm volatile("v_mul_f32 v56, 0xbfb8aa3b, v54\n v_mul_f32 v56, v56, v56" ::: "v56");
    else if (ABL == 4) asm volatile("v_mul_f32 v56, 0xbfb8aa3b, v54\n v_exp_f32 v56, v56\n v_mov_b32 v55, v57" ::: "v56", "v55");
    else asm volatile("v_mul_f32 v56, 0xbfb8aa3b, v54\n v_exp_f32 v56, v56\n s_nop 0" ::: "v56");
    asm volatile("v_mul_f32 v56, v57, v56\n v_min_f32 v56, 0x3f7d70a4, v56\n v_sub_f32 v55, 1.0, v56\n v_mul_f32 %[w], v56, %[T]\n v_mul_f32 %[T], %[T], v55\n"
                 "v_cmp_gt_f32 %[mk], %[k1e4], %[T]\n v_cmp_eq_u32_sdwa vcc, %[T], %[slice] src0_sel:WORD_1 src1_sel:DWORD\n"
                 : [T] "+v"(T), [w] "=&v"(w), [mk] "=&s"(mk) : [k1e4] "s"(k1e4), [slice] "s"(slice) : "vcc", "v55", "v56");
    if (ABL == 3) {
        asm volatile("v_fmac_f32 %[c0], v60, %[w]\n v_fmac_f32 %[c1], v61, %[w]\n v_fmac_f32 %[c2], v62, %[w]\n"
                     : [c0] "+v"(c0), [c1] "+v"(c1), [c2] "+v"(c2) : [w] "v"(w));
    } else {
        asm volatile("s_andn2_b64 exec, exec, %[mk]\n v_fmac_f32 %[c0], v60, %[w]\n v_fmac_f32 %[c1], v61, %[w]\n v_fmac_f32 %[c2], v62, %[w]\n"
                     "s_andn2_b64 %[alive], %[alive], %[mk]\n s_cselect_b32 %[rem], %[rem], 0\n s_mov_b64 exec, %[alive]\n s_add_u32 %[rem], %[rem], -1\n"
                     : [c0] "+v"(c0), [c1] "+v"(c1), [c2] "+v"(c2), [alive] "+s"(alive), [rem] "+s"(rem) : [w] "v"(w), [mk] "s"(mk) : "scc", "exec");
    }
}

template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, const float* entries, int iters) {
    __shared__ float4 slab[4][3][64];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    for (int p = 0; p < 3; ++p) slab[w][p][lane] = MODE >= 6 ? (p == 0 ? make_float4(0.001f, 0.0f, 0.001f, 0.3f) : p == 1 ? make_float4(3.5f, 3.5f, 0.5f, 0.5f) : make_float4(0.5f, 100.0f, 0.f, 0.f))
                                                              : make_float4(0.001f * lane, 0.002f, 0.003f, 0.5f);
    __syncthreads();
    float c0 = 0, c1 = 0, c2 = 0, T = 1.0f, ww;
    const float fx = (float)(lane & 7), fy = (float)(lane >> 3), kk = 1e-4f;
    uint32_t addr = (uint32_t)(uintptr_t)&slab[w][0][0];
    const uint32_t wu = (uint32_t)__builtin_amdgcn_readfirstlane((int)(blockIdx.x * 4 + w));  // wave-uniform, and known to be
    const float* e = entries + (size_t)(wu & 1023) * 16;
    for (int i = 0; i < iters; ++i) {
        if (MODE == 0) {
            asm volatile(VALU23 : [c0] "+v"(c0), [c1] "+v"(c1), [c2] "+v"(c2), [T] "+v"(T), [w] "=&v"(ww) : [fx] "v"(fx), [fy] "v"(fy), [k] "s"(kk)
                         : "vcc", "v54", "v55", "v56", "v57", "v58", "v59", "v60", "v61", "v62", "v63");
        } else if (MODE == 1) {
            asm volatile("ds_read_b128 v[54:57], %[a]\n ds_read_b128 v[58:61], %[a] offset:1024\n ds_read_b64 v[62:63], %[a] offset:2048\n s_waitcnt lgkmcnt(0)\n" VALU23
                         : [c0] "+v"(c0), [c1] "+v"(c1), [c2] "+v"(c2), [T] "+v"(T), [w] "=&v"(ww) : [fx] "v"(fx), [fy] "v"(fy), [k] "s"(kk), [a] "v"(addr)
                         : "vcc", "memory", "v54", "v55", "v56", "v57", "v58", "v59", "v60", "v61", "v62", "v63");
        } else if (MODE == 2) {
            asm volatile("ds_read_b128 v[54:57], %[a]\n ds_read_b128 v[58:61], %[a] offset:1024\n ds_read_b64 v[62:63], %[a] offset:2048\n s_waitcnt lgkmcnt(0)\n"
                         "v_add_f32 %[c0], v54, v58\n"
                         : [c0] "+v"(c0) : [a] "v"(addr) : "memory", "v54", "v55", "v56", "v57", "v58", "v59", "v60", "v61", "v62", "v63");
        } else if (MODE == 3) {
            uint64_t sv;
            asm volatile("s_mov_b64 %[sv], exec\n s_mov_b64 exec, 0xffff\n"
                         "ds_read_b128 v[54:57], %[a]\n ds_read_b128 v[58:61], %[a] offset:1024\n ds_read_b64 v[62:63], %[a] offset:2048\n s_mov_b64 exec, %[sv]\n s_waitcnt lgkmcnt(0)\n" VALU23
                         : [c0] "+v"(c0), [c1] "+v"(c1), [c2] "+v"(c2), [T] "+v"(T), [w] "=&v"(ww), [sv] "=&s"(sv) : [fx] "v"(fx), [fy] "v"(fy), [k] "s"(kk), [a] "v"(addr)
                         : "vcc", "memory", "v54", "v55", "v56", "v57", "v58", "v59", "v60", "v61", "v62", "v63");
        } else if (MODE == 4) {
            uint64_t sv;
            asm volatile("s_mov_b64 %[sv], exec\n s_mov_b64 exec, 0xffff\n"
                         "ds_read_b128 v[54:57], %[a]\n ds_read_b128 v[58:61], %[a] offset:1024\n ds_read_b64 v[62:63], %[a] offset:2048\n s_mov_b64 exec, %[sv]\n s_waitcnt lgkmcnt(0)\n"
                         "v_readfirstlane_b32 s40, v54\n v_readfirstlane_b32 s41, v55\n v_readfirstlane_b32 s42, v56\n v_readfirstlane_b32 s43, v57\n v_readfirstlane_b32 s44, v58\n"
                         "v_readfirstlane_b32 s45, v59\n v_readfirstlane_b32 s46, v60\n v_readfirstlane_b32 s47, v61\n v_readfirstlane_b32 s48, v62\n v_readfirstlane_b32 s49, v63\n"
                         "v_sub_f32 v58, s44, %[fx]\n v_sub_f32 v59, s45, %[fy]\n v_mul_f32 v54, s40, v58\n v_mul_f32 v56, s42, v59\n"
                         "v_mul_f32 v54, v58, v54\n v_mul_f32 v56, v59, v56\n v_mul_f32 v55, s41, v58\n v_add_f32 v54, v54, v56\n"
                         "v_mul_f32 v55, v55, v59\n v_add_f32 v54, v55, v54\n v_cmp_ge_u32 vcc, s49, v54\n v_mul_f32 v56, 0xbfb8aa3b, v54\n"
                         "v_exp_f32 v56, v56\n s_nop 0\n v_mul_f32 v56, s43, v56\n v_min_f32 v56, 0x3f7d70a4, v56\n v_sub_f32 v55, 1.0, v56\n"
                         "v_mul_f32 %[w], v56, %[T]\n v_mul_f32 %[T], %[T], v55\n v_cmp_gt_f32 vcc, %[k], %[T]\n v_cmp_eq_u32 vcc, %[T], %[k]\n"
                         "v_mul_f32 v60, s46, %[w]\n v_fmac_f32 %[c1], s47, %[w]\n v_fmac_f32 %[c2], s48, %[w]\n v_add_f32 %[c0], %[c0], v60\n"
                         : [c0] "+v"(c0), [c1] "+v"(c1), [c2] "+v"(c2), [T] "+v"(T), [w] "=&v"(ww), [sv] "=&s"(sv) : [fx] "v"(fx), [fy] "v"(fy), [k] "s"(kk), [a] "v"(addr)
                         : "vcc", "memory", "v54", "v55", "v56", "v57", "v58", "v59", "v60", "v61", "v62", "v63", "s40", "s41", "s42", "s43", "s44", "s45", "s46", "s47", "s48", "s49");
        } else if (MODE == 5) {
            // the entry of THIS iteration is in s[40:49] (loaded during the previous one); the next one's load is issued first
            asm volatile("s_waitcnt lgkmcnt(0)\n"
                         "s_mov_b64 s[60:61], s[40:41]\n s_mov_b64 s[62:63], s[42:43]\n s_mov_b64 s[64:65], s[44:45]\n s_mov_b64 s[66:67], s[46:47]\n s_mov_b64 s[68:69], s[48:49]\n"
                         "s_load_dwordx8 s[40:47], %[e], 0x0\n s_load_dwordx2 s[48:49], %[e], 0x20\n"
                         "v_sub_f32 v58, s64, %[fx]\n v_sub_f32 v59, s65, %[fy]\n v_mul_f32 v54, s60, v58\n v_mul_f32 v56, s62, v59\n"
                         "v_mul_f32 v54, v58, v54\n v_mul_f32 v56, v59, v56\n v_mul_f32 v55, s61, v58\n v_add_f32 v54, v54, v56\n"
                         "v_mul_f32 v55, v55, v59\n v_add_f32 v54, v55, v54\n v_cmp_ge_u32 vcc, s69, v54\n v_mul_f32 v56, 0xbfb8aa3b, v54\n"
                         "v_exp_f32 v56, v56\n s_nop 0\n v_mul_f32 v56, s63, v56\n v_min_f32 v56, 0x3f7d70a4, v56\n v_sub_f32 v55, 1.0, v56\n"
                         "v_mul_f32 %[w], v56, %[T]\n v_mul_f32 %[T], %[T], v55\n v_cmp_gt_f32 vcc, %[k], %[T]\n v_cmp_eq_u32 vcc, %[T], %[k]\n"
                         "v_fmac_f32 %[c0], s66, %[w]\n v_fmac_f32 %[c1], s67, %[w]\n v_fmac_f32 %[c2], s68, %[w]\n"
                         : [c0] "+v"(c0), [c1] "+v"(c1), [c2] "+v"(c2), [T] "+v"(T), [w] "=&v"(ww) : [fx] "v"(fx), [fy] "v"(fy), [k] "s"(kk), [e] "s"(e)
                         : "vcc", "memory", "v54", "v55", "v56", "v57", "v58", "v59", "s40", "s41", "s42", "s43", "s44", "s45", "s46", "s47", "s48", "s49", "s60", "s61", "s62",
                           "s63", "s64", "s65", "s66", "s67", "s68", "s69");
            e = entries + (size_t)((wu + (uint32_t)i) & 1023) * 16;
        } else if (MODE >= 6 && MODE <= 10) {
            uint64_t alive = ~0ull;
            uint32_t rem = 1000;
            real_pair<MODE - 6>(addr, alive, rem, fx, fy, T, c0, c1, c2);
            T = T * 0.5f + 0.5f;  // (keeps T away from the thresholds: one more VALU, in every ablation alike)
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = c0 + c1 + c2 + T;
}

int main() {
    float *out, *ent;
    const int blocks = 256 * 8, iters = 4096;
    hipMalloc(&out, blocks * 256 * 4);
    hipMalloc(&ent, 1024 * 64);
    hipMemset(ent, 0, 1024 * 64);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const char* names[] = {"23 VALU, no LDS", "23 VALU + 3 broadcast reads (shipped shape)", "3 broadcast reads alone (+1 VALU)", "23 VALU + 3 reads under exec = 16 lanes",
                           "33 VALU (10 readfirstlane) + 16-lane reads, entry in SGPRs", "23 VALU + s_load x8 + x2 of the next entry",
                           "the shipped body (+1 VALU to keep T in range)", "  ... v_exp_f32 -> v_mul_f32", "  ... v_cmpx (writes exec) -> v_cmp", "  ... without the scalar tail (exec / alive / rem)",
                           "  ... the s_nop behind v_exp replaced by a v_mov"};
    for (int mode = 0; mode < 11; ++mode) {
        float best = 1e9f;
        for (int rep = 0; rep < 4; ++rep) {
            hipEventRecord(e0);
            switch (mode) {
                case 0: hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(256), 0, 0, out, ent, iters); break;
                case 1: hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(256), 0, 0, out, ent, iters); break;
                case 2: hipLaunchKernelGGL(k<2>, dim3(blocks), dim3(256), 0, 0, out, ent, iters); break;
                case 3: hipLaunchKernelGGL(k<3>, dim3(blocks), dim3(256), 0, 0, out, ent, iters); break;
                case 4: hipLaunchKernelGGL(k<4>, dim3(blocks), dim3(256), 0, 0, out, ent, iters); break;
                case 5: hipLaunchKernelGGL(k<5>, dim3(blocks), dim3(256), 0, 0, out, ent, iters); break;
                case 6: hipLaunchKernelGGL(k<6>, dim3(blocks), dim3(256), 0, 0, out, ent, iters); break;
                case 7: hipLaunchKernelGGL(k<7>, dim3(blocks), dim3(256), 0, 0, out, ent, iters); break;
                case 8: hipLaunchKernelGGL(k<8>, dim3(blocks), dim3(256), 0, 0, out, ent, iters); break;
                case 9: hipLaunchKernelGGL(k<9>, dim3(blocks), dim3(256), 0, 0, out, ent, iters); break;
                case 10: hipLaunchKernelGGL(k<10>, dim3(blocks), dim3(256), 0, 0, out, ent, iters); break;
            }
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            if (ms < best) best = ms;
        }
        // 8 waves per SIMD, each `iters` iterations: ns per iteration and SIMD
        const double ns = best * 1e6 / ((double)iters * 8);
        printf("mode %d  %-62s %8.3f ms   %6.1f ns per pair and SIMD  (%5.1f cycles at 2.33 GHz)\n", mode, names[mode], best, ns, ns * 2.33);
    }
    return 0;
}
