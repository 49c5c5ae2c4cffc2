// occ_test.hip -- do two 1024-thread workgroups share a CU?  512 blocks on 256 CUs; every block records the constant-rate
// clock at its start and spins ~20 us.  "start spread" ~ 0: all resident at once; ~ 20 us: one workgroup per CU.
//   hipcc --offload-arch=gfx950 -O2 tools/ubench/occ_test.hip -o tools/ubench/occ_test && tools/ubench/occ_test
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <vector>

template <int NSGPR_DUMMY>
__global__ __launch_bounds__(1024, 8) void k_spin(unsigned long long* t0, unsigned long long* t1, int spin_ticks, const int* junk) {
    extern __shared__ unsigned int lds[];
    const unsigned long long a = wall_clock64();
    if (threadIdx.x == 0) t0[blockIdx.x] = a;
    // optional scalar register pressure: NSGPR_DUMMY uniform values kept live across the spin
    int acc[NSGPR_DUMMY > 0 ? NSGPR_DUMMY : 1];
#pragma unroll
    for (int k = 0; k < NSGPR_DUMMY; ++k) acc[k] = junk[k];
    lds[threadIdx.x] = threadIdx.x;
    __syncthreads();
    while (wall_clock64() - a < (unsigned long long)spin_ticks) {
#pragma unroll
        for (int k = 0; k < NSGPR_DUMMY; ++k) acc[k] = acc[k] * 3 + k;
    }
    int s = 0;
#pragma unroll
    for (int k = 0; k < NSGPR_DUMMY; ++k) s += acc[k];
    if (threadIdx.x == 0) t1[blockIdx.x] = wall_clock64() + (s == 12345 ? 1 : 0) + lds[5] * 0;
}

template <int N>
static void run(const char* name, size_t lds_bytes, unsigned long long* d0, unsigned long long* d1, const int* junk) {
    const int blocks = 512;
    if (lds_bytes > 65536)
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_spin<N>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
    for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL(k_spin<N>, dim3(blocks), dim3(1024), lds_bytes, 0, d0, d1, 2000, junk);
    (void)hipDeviceSynchronize();
    std::vector<unsigned long long> t0(blocks), t1(blocks);
    (void)hipMemcpy(t0.data(), d0, blocks * 8, hipMemcpyDeviceToHost);
    (void)hipMemcpy(t1.data(), d1, blocks * 8, hipMemcpyDeviceToHost);
    const auto lo = *std::min_element(t0.begin(), t0.end());
    const auto hi = *std::max_element(t0.begin(), t0.end());
    const auto end = *std::max_element(t1.begin(), t1.end());
    int occ = -1;
    (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, reinterpret_cast<const void*>(&k_spin<N>), 1024, lds_bytes);
    hipFuncAttributes a{};
    (void)hipFuncGetAttributes(&a, reinterpret_cast<const void*>(&k_spin<N>));
    std::printf("%-28s lds %6zu B  regs %3d  api says %d blocks/CU   start spread %6.1f us   kernel %6.1f us\n", name, lds_bytes,
                a.numRegs, occ, (hi - lo) / 100.0, (end - lo) / 100.0);
}

int main() {
    unsigned long long *d0, *d1;
    int* junk;
    (void)hipMalloc(&d0, 512 * 8);
    (void)hipMalloc(&d1, 512 * 8);
    (void)hipMalloc(&junk, 256 * 4);
    (void)hipMemset(junk, 1, 256 * 4);
    run<0>("plain, 4 KiB LDS", 4096, d0, d1, junk);
    run<0>("plain, 40 KiB LDS", 40960, d0, d1, junk);
    run<0>("plain, 72 KiB LDS", 73728, d0, d1, junk);
    run<0>("plain, 81 KiB LDS", 82944, d0, d1, junk);
    run<40>("40 live uniforms, 40 KiB", 40960, d0, d1, junk);
    run<70>("70 live uniforms, 40 KiB", 40960, d0, d1, junk);
    return 0;
}
