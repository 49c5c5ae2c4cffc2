// Micro-benchmark (round 6): how fast does gfx950 take device-scope atomicAdd on a FEW hot addresses from every CU?
// The question behind it: could level 1 size its bins' candidate lists with one global atomic per (workgroup, bin) instead of a
// count kernel + a scan kernel?  hipcc --offload-arch=gfx950 -O3 atomic_rate.hip -o atomic_rate && ./atomic_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

// every thread issues `per` atomics; address = counter[(hash) % naddr * stride]; ret: use the returned value
template <bool RET>
__global__ __launch_bounds__(256) void k(uint32_t* c, uint32_t naddr, uint32_t stride, int per, uint32_t* sink) {
    uint32_t h = (blockIdx.x * 256u + threadIdx.x) * 2654435761u;
    uint32_t acc = 0;
    for (int i = 0; i < per; ++i) {
        h = h * 1664525u + 1013904223u;
        uint32_t* p = c + (size_t)((h >> 8) % naddr) * stride;
        if (RET) acc += atomicAdd(p, 1u);
        else __hip_atomic_fetch_add(p, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (RET && acc == 0xFFFFFFFFu) *sink = acc;
}

int main() {
    uint32_t *c, *sink;
    hipMalloc(&c, 1 << 24);
    hipMalloc(&sink, 4);
    hipMemset(c, 0, 1 << 24);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    struct Case { const char* name; uint32_t naddr, stride; int blocks, per; bool ret; };
    const Case cases[] = {
        {"135 addr, 128 B apart, no return, 3906 WG x 256 thr x 1 (1.0 M atomics)", 135, 32, 3906, 1, false},
        {"135 addr, adjacent words, no return, 1.0 M", 135, 1, 3906, 1, false},
        {"135 addr, 128 B apart, WITH return, 1.0 M", 135, 32, 3906, 1, true},
        {"510 addr, 128 B apart, no return, 1.0 M", 510, 32, 3906, 1, false},
        {"510 addr, 128 B apart, no return, 5.0 M", 510, 32, 3906, 5, false},
        {"135 addr, 128 B apart, no return, 977 WG x 135 thr-equivalent (0.25 M)", 135, 32, 977, 1, false},
        {"135 addr, 128 B apart, WITH return, 0.25 M", 135, 32, 977, 1, true},
        {"1 addr, no return, 0.25 M", 1, 32, 977, 1, false},
        {"65536 addr, 128 B apart, no return, 1.0 M", 65536, 32, 3906, 1, false},
    };
    for (const Case& cs : cases) {
        float best = 1e9f;
        for (int rep = 0; rep < 5; ++rep) {
            hipEventRecord(e0);
            if (cs.ret) hipLaunchKernelGGL(k<true>, dim3(cs.blocks), dim3(256), 0, 0, c, cs.naddr, cs.stride, cs.per, sink);
            else hipLaunchKernelGGL(k<false>, dim3(cs.blocks), dim3(256), 0, 0, c, cs.naddr, cs.stride, cs.per, sink);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            if (ms < best) best = ms;
        }
        const double n = (double)cs.blocks * 256 * cs.per;
        printf("%-78s %8.1f us  %6.2f ns/atomic  %7.1f ns per atomic and address\n", cs.name, best * 1e3, best * 1e6 / n, best * 1e6 / (n / cs.naddr));
    }
    return 0;
}
