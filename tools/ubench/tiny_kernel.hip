// Micro-benchmark (round 6): what does the one-wave k_frame_end cost, and how much of it is its 64-byte write to pinned host memory?
// 2000 back-to-back launches of a one-wave kernel on one stream: (a) stamping device memory, (b) stamping pinned host memory,
// (c) both (the shipped form), (d) empty.  hipcc --offload-arch=gfx950 -O3 tiny_kernel.hip -o tiny_kernel && ./tiny_kernel
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
template <int MODE>
__global__ __launch_bounds__(64) void k(uint64_t* dev, uint64_t* host) {
    const uint64_t now = wall_clock64();
    if (MODE == 0 && threadIdx.x == 6) dev[6] = now;
    if (MODE == 1 && threadIdx.x < 8) host[threadIdx.x] = now;
    if (MODE == 2 && threadIdx.x < 8) {
        const uint64_t v = threadIdx.x == 6 ? now : dev[threadIdx.x];
        if (threadIdx.x == 6) dev[6] = now;
        host[threadIdx.x] = v;
    }
}
int main() {
    uint64_t *dev, *host;
    hipMalloc(&dev, 64);
    hipMemset(dev, 0, 64);
    hipHostMalloc(&host, 64, hipHostMallocDefault);
    hipStream_t s;
    hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const char* names[] = {"stamp to device memory", "8 stamps to pinned host memory", "both (k_frame_end as shipped)", "empty"};
    for (int mode = 0; mode < 4; ++mode) {
        float best = 1e9f;
        for (int rep = 0; rep < 5; ++rep) {
            hipEventRecord(e0, s);
            for (int i = 0; i < 2000; ++i) {
                if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(1), dim3(64), 0, s, dev, host);
                if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(1), dim3(64), 0, s, dev, host);
                if (mode == 2) hipLaunchKernelGGL(k<2>, dim3(1), dim3(64), 0, s, dev, host);
                if (mode == 3) hipLaunchKernelGGL(k<3>, dim3(1), dim3(64), 0, s, dev, host);
            }
            hipEventRecord(e1, s);
            hipEventSynchronize(e1);
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            if (ms < best) best = ms;
        }
        printf("%-36s %6.2f us per launch (2000 back to back)\n", names[mode], best * 1e3 / 2000);
    }
    return 0;
}
