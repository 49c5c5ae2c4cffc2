// Micro-benchmark: wave64 VALU issue rate on gfx950 -- scalar fp32 ops vs packed (v_pk_*) ops.
// hipcc --offload-arch=gfx950 -O3 valu_rate.hip -o valu_rate && ./valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v2f __attribute__((ext_vector_type(2)));

template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, int iters) {
    float a0 = threadIdx.x * 1e-3f, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    v2f p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7};
    const float m = 1.0000001f, c = 1e-9f;
    const v2f pm = {m, m}, pc = {c, c};
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            if (MODE == 0) {  // 8 independent v_mul_f32
                a0 *= m; a1 *= m; a2 *= m; a3 *= m; a4 *= m; a5 *= m; a6 *= m; a7 *= m;
            } else if (MODE == 1) {  // 8 independent v_fma_f32
                a0 = __builtin_fmaf(a0, m, c); a1 = __builtin_fmaf(a1, m, c); a2 = __builtin_fmaf(a2, m, c); a3 = __builtin_fmaf(a3, m, c);
                a4 = __builtin_fmaf(a4, m, c); a5 = __builtin_fmaf(a5, m, c); a6 = __builtin_fmaf(a6, m, c); a7 = __builtin_fmaf(a7, m, c);
            } else if (MODE == 2) {  // 4 independent v_pk_mul_f32 (8 floats)
                p0 *= pm; p1 *= pm; p2 *= pm; p3 *= pm;
            } else if (MODE == 3) {  // 4 independent v_pk_fma_f32
                p0 = __builtin_elementwise_fma(p0, pm, pc); p1 = __builtin_elementwise_fma(p1, pm, pc);
                p2 = __builtin_elementwise_fma(p2, pm, pc); p3 = __builtin_elementwise_fma(p3, pm, pc);
            } else if (MODE == 6) {  // 8 independent v_mul_f32, inline asm (no SLP packing)
                asm volatile("v_mul_f32 %0, %0, %8\n v_mul_f32 %1, %1, %8\n v_mul_f32 %2, %2, %8\n v_mul_f32 %3, %3, %8\n"
                             "v_mul_f32 %4, %4, %8\n v_mul_f32 %5, %5, %8\n v_mul_f32 %6, %6, %8\n v_mul_f32 %7, %7, %8"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m));
            } else if (MODE == 7) {  // 8 independent v_fma_f32, inline asm
                asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
                             "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m), "v"(c));
            } else if (MODE == 8) {  // 8 v_exp_f32 (transcendental)
                asm volatile("v_exp_f32 %0, %0\n v_exp_f32 %1, %1\n v_exp_f32 %2, %2\n v_exp_f32 %3, %3\n"
                             "v_exp_f32 %4, %4\n v_exp_f32 %5, %5\n v_exp_f32 %6, %6\n v_exp_f32 %7, %7"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
            } else if (MODE >= 9 && MODE <= 12) {
                // 8 independent v_fma_f32 under a partial exec mask: does the hardware skip an inactive 32-lane half?
                // 9: lanes 0-31 only; 10: lanes 32-63 only; 11: even lanes (both halves half full); 12: one lane
                const unsigned lane = threadIdx.x & 63u;
                const bool on = MODE == 9 ? lane < 32u : MODE == 10 ? lane >= 32u : MODE == 11 ? (lane & 1u) == 0u : lane == 0u;
                if (on)
                    asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
                                 "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9"
                                 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m), "v"(c));
            } else if (MODE >= 20 && MODE <= 24) {  // binary64 and conversions: what gs_expf_libm is made of
                double d0 = a0, d1 = a1, d2 = a2, d3 = a3;
                if (MODE == 20)
                    asm volatile("v_fma_f64 %0, %0, %4, %5\n v_fma_f64 %1, %1, %4, %5\n v_fma_f64 %2, %2, %4, %5\n v_fma_f64 %3, %3, %4, %5\n"
                                 "v_fma_f64 %0, %0, %4, %5\n v_fma_f64 %1, %1, %4, %5\n v_fma_f64 %2, %2, %4, %5\n v_fma_f64 %3, %3, %4, %5"
                                 : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3) : "v"((double)m), "v"((double)c));
                if (MODE == 21)
                    asm volatile("v_mul_f64 %0, %0, %4\n v_mul_f64 %1, %1, %4\n v_mul_f64 %2, %2, %4\n v_mul_f64 %3, %3, %4\n"
                                 "v_mul_f64 %0, %0, %4\n v_mul_f64 %1, %1, %4\n v_mul_f64 %2, %2, %4\n v_mul_f64 %3, %3, %4"
                                 : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3) : "v"((double)m));
                if (MODE == 22)
                    asm volatile("v_add_f64 %0, %0, %4\n v_add_f64 %1, %1, %4\n v_add_f64 %2, %2, %4\n v_add_f64 %3, %3, %4\n"
                                 "v_add_f64 %0, %0, %4\n v_add_f64 %1, %1, %4\n v_add_f64 %2, %2, %4\n v_add_f64 %3, %3, %4"
                                 : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3) : "v"((double)c));
                if (MODE == 23)  // f32 -> f64
                    asm volatile("v_cvt_f64_f32 %0, %4\n v_cvt_f64_f32 %1, %5\n v_cvt_f64_f32 %2, %6\n v_cvt_f64_f32 %3, %7\n"
                                 "v_cvt_f64_f32 %0, %4\n v_cvt_f64_f32 %1, %5\n v_cvt_f64_f32 %2, %6\n v_cvt_f64_f32 %3, %7"
                                 : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3) : "v"(a4), "v"(a5), "v"(a6), "v"(a7));
                if (MODE == 24)  // f64 -> f32
                    asm volatile("v_cvt_f32_f64 %0, %4\n v_cvt_f32_f64 %1, %5\n v_cvt_f32_f64 %2, %6\n v_cvt_f32_f64 %3, %7\n"
                                 "v_cvt_f32_f64 %0, %4\n v_cvt_f32_f64 %1, %5\n v_cvt_f32_f64 %2, %6\n v_cvt_f32_f64 %3, %7"
                                 : "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(d0), "v"(d1), "v"(d2), "v"(d3));
                a0 = (float)d0; a1 = (float)d1; a2 = (float)d2; a3 = (float)d3;
            } else if (MODE == 4) {  // 8 v_add_f32
                a0 += c; a1 += c; a2 += c; a3 += c; a4 += c; a5 += c; a6 += c; a7 += c;
            } else if (MODE == 5) {  // 8 v_cndmask (select on compare) + cmp
                a0 = a0 > a1 ? a2 : a0; a1 = a1 > a2 ? a3 : a1; a2 = a2 > a3 ? a4 : a2; a3 = a3 > a4 ? a5 : a3;
                a4 = a4 > a5 ? a6 : a4; a5 = a5 > a6 ? a7 : a5; a6 = a6 > a7 ? a0 : a6; a7 = a7 > a0 ? a1 : a7;
            }
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + p0.x + p0.y + p1.x + p1.y + p2.x + p2.y + p3.x + p3.y;
}

template <int MODE>
void run(const char* name, int floats_per_inst, int insts_per_u) {
    float* d;
    const int blocks = 256 * 8, iters = 2000;
    hipMalloc(&d, blocks * 256 * 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, 10);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double wave_insts = (double)blocks * 4 * iters * 16 * insts_per_u;
    double per_simd = wave_insts / 1024.0;
    printf("%-14s %8.3f ms  %6.2f ns/inst/SIMD  -> %.2f cycles@2.4GHz  %.1f Gflop-lanes/s\n", name, ms,
           ms * 1e6 / per_simd, ms * 1e6 / per_simd * 2.4, wave_insts * 64 * floats_per_inst / (ms * 1e-3) / 1e9);
    hipFree(d);
}
int main() {
    run<0>("v_mul_f32", 1, 8);
    run<1>("v_fma_f32", 1, 8);
    run<2>("v_pk_mul_f32", 2, 4);
    run<3>("v_pk_fma_f32", 2, 4);
    run<4>("v_add_f32", 1, 8);
    run<6>("asm v_mul_f32", 1, 8);
    run<7>("asm v_fma_f32", 1, 8);
    run<8>("asm v_exp_f32", 1, 8);
    run<5>("cmp+cndmask", 1, 16);
    // binary64: 8 instructions of the kind per step, plus 4 cvt in + 4 cvt out of the harness (subtract the cvt rows)
    run<20>("v_fma_f64 (+8cvt)", 1, 8);
    run<21>("v_mul_f64 (+8cvt)", 1, 8);
    run<22>("v_add_f64 (+8cvt)", 1, 8);
    run<23>("v_cvt_f64_f32 (+8cvt)", 1, 8);
    run<24>("v_cvt_f32_f64 (+8cvt)", 1, 8);
    run<9>("fma, lanes 0-31", 1, 8);
    run<10>("fma, lanes 32-63", 1, 8);
    run<11>("fma, even lanes", 1, 8);
    run<12>("fma, one lane", 1, 8);
    return 0;
}
