// gs_radix.hip -- stable 8-bit LSD radix passes over the visible Gaussians' depth bits (global depth-order path only).
//
// Part of libgs3d_hip.so (gfx950 only).  Built with -ffp-contract=off: the floating-point contract of this path is "IEEE
// binary32, one rounding per operation, in the order the reference shader writes it" (DESIGN.md section 3); fused
// multiply-adds appear only where written explicitly.
// Reference restated (paths relative to /root/reference/src/shaders): sort/hist.comp + sort/sort.comp (result: stable ascending order)
#include "gs_device.h"

namespace gs {

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
    int shift;
    uint32_t mask;
    int blocks;
    uint64_t* stamps;  // nullable: the frame's timeline (the first pass's histogram kernel stamps ST_ORDER)
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
    frame_stamp(a.stamps, ST_ORDER);
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
    a.shift = p.shift;
    a.mask = (1u << p.bits) - 1u;
    a.blocks = p.blocks;
    a.stamps = p.stamps;
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
}  // namespace gs
