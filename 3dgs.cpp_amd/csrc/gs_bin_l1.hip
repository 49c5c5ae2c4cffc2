// gs_bin_l1.hip -- level 1 of the tile binning: which Gaussian touches which bin (count, scan, scatter of candidate records).
//
// Part of libgs3d_hip.so (gfx950 only).  Built with -ffp-contract=off: the floating-point contract of this path is "IEEE
// binary32, one rounding per operation, in the order the reference shader writes it" (DESIGN.md section 3); fused
// multiply-adds appear only where written explicitly.
// Reference restated (paths relative to /root/reference/src/shaders): prefix_sum.comp:32-59 + preprocess_sort.comp:31-61
#include "gs_bin.h"

namespace gs {

#ifndef GS_L1_WORDLOOP
#define GS_L1_WORDLOOP 1
#endif

// Wave w of the block takes chunks 4w .. 4w + 3 (consecutive items): all loads of its four chunks are issued before
// the first is used, and 8 such blocks are resident per CU -- the kernels are a handful of dependent memory round
// trips each, so what matters is how many of them are in flight.
template <int R1>
__global__ __launch_bounds__(BLOCK) void k_l1_hist(L1Args a) {
    constexpr int NB = 64 * R1;
    __shared__ uint32_t s_hist[NB];
    __shared__ uint32_t s_vis;
    const int tid = threadIdx.x, lane = tid & (WAVE - 1), w = tid / WAVE;
    frame_stamp(a.stamps, ST_L1_COUNT);
    const uint32_t blk = l1_block();
    if (blk >= a.nblk) return;
    uint32_t n = a.n_items ? *a.n_items : a.n_bound;
    if (n > a.n_bound) n = a.n_bound;
    uint32_t vis_first = 0;
    const uint32_t vis_here = a.vis ? l1_vis_block(a, blk, vis_first) : 0u;
    if (a.vis && vis_here == 0) return;
    for (int b = tid; b < NB; b += BLOCK) s_hist[b] = 0;
    if (tid == 0) s_vis = 0;
    __syncthreads();
    // all four items' loads first (independent round trips), then one chunk at a time: the chunk body holds up to 16
    // inlined transposes and must not be unrolled four times over (the instruction cache is 64 KiB)
    __shared__ uint32_t s_box[kL1Chunks][WAVE];
    {
        uint32_t box[kL1PerWave];
        if (a.vis) {  // dense items: one 16-byte load each, no culled lanes but in the list's last block
            uint4 r[kL1PerWave];
#pragma unroll
            for (int j = 0; j < kL1PerWave; ++j) {
                const uint32_t q = (w * kL1PerWave + j) * WAVE + lane;
                r[j] = q < vis_here ? a.vis[vis_first + q] : make_uint4(0, 0, 0, 0);
            }
#pragma unroll
            for (int j = 0; j < kL1PerWave; ++j) {
                const uint32_t q = (w * kL1PerWave + j) * WAVE + lane;
                box[j] = q < vis_here ? l1_bin_box(a, r[j]) : 0u;
            }
        } else {
#pragma unroll
            for (int j = 0; j < kL1PerWave; ++j)
                l1_item(a, blk * kL1Items + (w * kL1PerWave + j) * WAVE + lane, n, box[j]);
        }
#pragma unroll
        for (int j = 0; j < kL1PerWave; ++j) s_box[w * kL1PerWave + j][lane] = box[j];
    }
    // counting needs no order: every lane adds one to each bin of its box (LDS atomics; typically 1-4 bins, and the rare
    // screen-filling splat only slows its own wave)
    uint32_t vis = 0;
#pragma unroll 1
    for (int j = 0; j < kL1PerWave; ++j) {
        const uint32_t box = s_box[w * kL1PerWave + j][lane];  // wave-private row: no barrier needed
        vis += (uint32_t)__popcll(__ballot(box != 0));
        const uint32_t x0 = box & 255u, y0 = (box >> 8) & 255u, x1 = (box >> 16) & 255u, y1 = box >> 24;
        // a splat that touches many bins (a screen-filling one touches all of them) would keep its lane in this loop for
        // hundreds of rounds while the other 63 wait: boxes of more than kL1BigBox bins are spread over the whole wave below
        const bool big = (x1 - x0) * (y1 - y0) > kL1BigBox;
        if (!big)
            for (uint32_t y = y0; y < y1; ++y)
                for (uint32_t x = x0; x < x1; ++x) atomicAdd(&s_hist[(y << a.g.grid_shift) | x], 1u);
        for (uint64_t bm = __ballot(big); bm != 0; bm &= bm - 1) {
            const uint32_t bb = (uint32_t)__builtin_amdgcn_readlane((int)box, __ffsll((unsigned long long)bm) - 1);
            const uint32_t bx0 = bb & 255u, by0 = (bb >> 8) & 255u, bw = ((bb >> 16) & 255u) - bx0, cells = bw * ((bb >> 24) - by0);
            for (uint32_t c = lane; c < cells; c += WAVE) atomicAdd(&s_hist[((by0 + c / bw) << a.g.grid_shift) | (bx0 + c % bw)], 1u);
        }
    }
    if (lane == 0 && vis) atomicAdd(&s_vis, vis);
    __syncthreads();
    for (int b = tid; b < NB; b += BLOCK)
        if (bin_on_screen(a.g, b)) a.hist[(size_t)b * a.nblk + blk] = s_hist[b];
    // V on the bin-local path (on the global path the first depth pass counts it): one more row of the table, summed
    // by k_l1_scan -- a thousand atomics on one counter would cost more than the rest of this kernel
    if (tid == 0 && !a.order && !a.vis) a.hist[(size_t)NB * a.nblk + blk] = s_vis;
}

// One workgroup per bin: exclusive prefix of the bin's row of block counts (in place), row total -> bin_count.
__global__ __launch_bounds__(BLOCK) void k_l1_scan(L1Args a) {
    __shared__ uint32_t scratch[8];
    const uint32_t bin = blockIdx.x;
    const uint32_t nb = 1u << (2 * a.g.grid_shift);
    if (bin == nb) {  // the row of per-block visible counts (bin-local path): V
        if (a.order) return;
        if (a.vis) {  // the dense lists' lengths add up to V
            uint32_t c = threadIdx.x < kVisRegions ? a.vis_count[threadIdx.x * kVisCounterStride] : 0u;
            if (c > a.vis_region_slots) c = a.vis_region_slots;
            uint32_t total;
            block_excl_scan<BLOCK>(c, scratch, &total);
            if (threadIdx.x == 0) a.counters->visible = total;
            return;
        }
        uint32_t sum = 0;
        for (uint32_t i = threadIdx.x; i < a.nblk; i += BLOCK) sum += a.hist[(size_t)nb * a.nblk + i];
        uint32_t total;
        block_excl_scan<BLOCK>(sum, scratch, &total);
        if (threadIdx.x == 0) a.counters->visible = total;
        return;
    }
    if (!bin_on_screen(a.g, bin)) {
        if (threadIdx.x == 0) a.bin_count[bin] = 0;
        return;
    }
    uint32_t* row = a.hist + (size_t)bin * a.nblk;
    // dense lists: only the blocks that had entries wrote their cell (the first ceil(count / kL1Items) of every list)
    __shared__ uint32_t s_used[kVisRegions];  // per list: blocks with entries
    const uint32_t per = a.vis ? a.vis_region_slots / kL1Items : 1u;
    if (a.vis) {
        if (threadIdx.x < kVisRegions) {
            uint32_t c = a.vis_count[threadIdx.x * kVisCounterStride];
            if (c > a.vis_region_slots) c = a.vis_region_slots;
            s_used[threadIdx.x] = (c + kL1Items - 1) / kL1Items;
        }
        __syncthreads();
    }
    uint32_t running = 0;
    for (uint32_t base = 0; base < a.nblk; base += 4 * BLOCK) {
        const uint32_t i0 = base + threadIdx.x * 4;
        uint32_t v[4], sum = 0;
        bool live[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            live[k] = i0 + k < a.nblk && (!a.vis || (i0 + k) % per < s_used[(i0 + k) / per]);
            v[k] = live[k] ? row[i0 + k] : 0u;
            sum += v[k];
        }
        uint32_t total;
        uint32_t excl = running + block_excl_scan<BLOCK>(sum, scratch, &total);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if (live[k]) row[i0 + k] = excl;
            excl += v[k];
        }
        running += total;
    }
    if (threadIdx.x == 0) a.bin_count[bin] = running;
}

template <int R1>
__global__ __launch_bounds__(BLOCK) void k_l1_scatter(L1Args a) {
    constexpr int NB = 64 * R1;
    __shared__ uint32_t s_start[NB];            // where this block's run starts in each bin's list
    __shared__ uint16_t s_cnt[kL1Chunks][NB];   // per chunk and bin: count, then exclusive prefix over the chunks
    __shared__ uint32_t s_ids[kL1Chunks][WAVE];
    __shared__ uint32_t scratch[8];
    const int tid = threadIdx.x, lane = tid & (WAVE - 1), w = tid / WAVE;
    frame_stamp(a.stamps, ST_L1_SCATTER);
    const uint32_t blk = l1_block();
    if (blk >= a.nblk) return;
    {   // bin offsets = exclusive scan of the bin totals (<= 1024 values: every block redoes it, no extra launch)
        uint32_t c[NB / BLOCK], sum = 0;
#pragma unroll
        for (int k = 0; k < NB / BLOCK; ++k) {
            c[k] = a.bin_count[tid * (NB / BLOCK) + k];
            sum += c[k];
        }
        uint32_t total;
        uint32_t off = block_excl_scan<BLOCK>(sum, scratch, &total);
#pragma unroll
        for (int k = 0; k < NB / BLOCK; ++k) {
            const uint32_t b = tid * (NB / BLOCK) + k;
            s_start[b] = off + (bin_on_screen(a.g, b) ? a.hist[(size_t)b * a.nblk + blk] : 0u);
            if (blk == 0) a.bin_count[kBinOffsets + b] = off;  // where bin b's run starts: level 2 reads it instead of redoing this scan per bin
            off += c[k];
            if (blk == 0 && c[k]) atomicMax(&a.counters->max_bin, c[k]);  // the fullest bin
        }
        if (blk == 0 && tid == 0) {  // E1, candidate overflow
            a.counters->bin_entries = total;
            if (total > a.capacity) atomicOr(&a.counters->overflow, 1u);
        }
    }
    uint32_t n = a.n_items ? *a.n_items : a.n_bound;
    if (n > a.n_bound) n = a.n_bound;
    __shared__ uint32_t s_box[kL1Chunks][WAVE];
    {   // all four items' loads first (independent round trips)
        uint32_t box[kL1PerWave], gid[kL1PerWave];
#pragma unroll
        for (int j = 0; j < kL1PerWave; ++j)
            gid[j] = l1_item(a, blk * kL1Items + (w * kL1PerWave + j) * WAVE + lane, n, box[j]);
#pragma unroll
        for (int j = 0; j < kL1PerWave; ++j) {
            s_ids[w * kL1PerWave + j][lane] = gid[j];
            s_box[w * kL1PerWave + j][lane] = box[j];
        }
    }
    // per chunk and bin: how many of the chunk's items touch the bin.  Counting needs no order: LDS atomics on the
    // 16-bit counters, two to a word (a chunk contributes at most 64 to a counter: no carry between the halves)
    {
        uint32_t* words = reinterpret_cast<uint32_t*>(&s_cnt[0][0]);
        for (int k = tid; k < kL1Chunks * NB / 2; k += BLOCK) words[k] = 0;
        __syncthreads();
#pragma unroll 1
        for (int j = 0; j < kL1PerWave; ++j) {
            const int ch = w * kL1PerWave + j;
            const uint32_t box = s_box[ch][lane];
            const uint32_t x0 = box & 255u, y0 = (box >> 8) & 255u, x1 = (box >> 16) & 255u, y1 = box >> 24;
            for (uint32_t y = y0; y < y1; ++y)
                for (uint32_t x = x0; x < x1; ++x) {
                    const uint32_t e = (uint32_t)ch * NB + ((y << a.g.grid_shift) | x);
                    atomicAdd(&words[e >> 1], 1u << (16u * (e & 1u)));
                }
        }
    }
    __syncthreads();
    for (int b = tid; b < NB; b += BLOCK) {
        uint32_t run = 0;
#pragma unroll
        for (int k = 0; k < kL1Chunks; ++k) {
            const uint32_t v = s_cnt[k][b];
            s_cnt[k][b] = (uint16_t)run;
            run += v;
        }
    }
    __syncthreads();
    // raw buffer over the candidate list: byte offsets >= 4 * capacity are dropped by the hardware bounds check
    const __amdgpu_buffer_rsrc_t out = __builtin_amdgcn_make_buffer_rsrc(a.cand, 0, a.capacity * 4u, 0x27000);
#pragma unroll 1
    for (int j = 0; j < kL1PerWave; ++j) {
        const int ch = w * kL1PerWave + j;
#if GS_L1_WORDLOOP
        const uint32_t box = s_box[ch][lane];
#pragma unroll 1
        for (int r = 0; r < R1; ++r) {
            const uint64_t mr = cover_word(a.g.grid_shift, box, r);
            if (__builtin_amdgcn_ballot_w64(mr != 0) == 0) continue;
            walk_column(wave_transpose64(mr, lane), s_start[r * 64 + lane] + s_cnt[ch][r * 64 + lane], out, s_ids[ch]);
        }
#else
        uint64_t m[R1];
        packed_cover_masks<R1>(a.g.grid_shift, s_box[ch][lane], m);
#pragma unroll
        for (int r = 0; r < R1; ++r) {
            if (__builtin_amdgcn_ballot_w64(m[r] != 0) == 0) continue;
            walk_column(wave_transpose64(m[r], lane), s_start[r * 64 + lane] + s_cnt[ch][r * 64 + lane], out, s_ids[ch]);
        }
#endif
    }
}

template <int R1>
__global__ __launch_bounds__(BLOCK) void k_l1_scatter_any_order(L1Args a) {
    constexpr int NB = 64 * R1;
    __shared__ uint32_t s_cur[NB];  // next free slot of this block's run in each bin's list
    __shared__ uint32_t scratch[8];
    const int tid = threadIdx.x, lane = tid & (WAVE - 1), w = tid / WAVE;
    frame_stamp(a.stamps, ST_L1_SCATTER);
    const uint32_t blk = l1_block();
    if (blk >= a.nblk) return;
    uint32_t vis_first = 0;
    const uint32_t vis_here = a.vis ? l1_vis_block(a, blk, vis_first) : 0u;
    if (a.vis && blk != 0 && vis_here == 0) return;  // (block 0 always reports E1 and the fullest bin)
    {   // bin offsets = exclusive scan of the bin totals (<= 1024 values: every block redoes it, no extra launch)
        uint32_t c[NB / BLOCK], sum = 0;
#pragma unroll
        for (int k = 0; k < NB / BLOCK; ++k) {
            c[k] = a.bin_count[tid * (NB / BLOCK) + k];
            sum += c[k];
        }
        uint32_t total;
        uint32_t off = block_excl_scan<BLOCK>(sum, scratch, &total);
#pragma unroll
        for (int k = 0; k < NB / BLOCK; ++k) {
            const uint32_t b = tid * (NB / BLOCK) + k;
            s_cur[b] = off + (bin_on_screen(a.g, b) ? a.hist[(size_t)b * a.nblk + blk] : 0u);
            if (blk == 0) a.bin_count[kBinOffsets + b] = off;  // where bin b's run starts: level 2 reads it instead of redoing this scan per bin
            off += c[k];
            if (blk == 0 && c[k]) atomicMax(&a.counters->max_bin, c[k]);  // the fullest bin
        }
        if (blk == 0 && tid == 0) {  // E1, candidate overflow
            a.counters->bin_entries = total;
            if (total > a.capacity) atomicOr(&a.counters->overflow, 1u);
        }
    }
    // all four items' loads first.  Dense list: one 16-byte load per item; else two dependent round trips over the N-wide
    // planes: tiles, then box + depth of the visible ones
    uint32_t nt[kL1PerWave], key[kL1PerWave], ids[kL1PerWave];
    ushort4 tb[kL1PerWave];
    if (a.vis) {
        uint4 r[kL1PerWave];
#pragma unroll
        for (int j = 0; j < kL1PerWave; ++j) {
            const uint32_t q = (w * kL1PerWave + j) * WAVE + lane;
            r[j] = q < vis_here ? a.vis[vis_first + q] : make_uint4(0, 0, 0, 0);
        }
#pragma unroll
        for (int j = 0; j < kL1PerWave; ++j) {
            const uint32_t q = (w * kL1PerWave + j) * WAVE + lane;
            nt[j] = q < vis_here ? 1u : 0u;
            ids[j] = r[j].x;
            key[j] = r[j].y;
            tb[j] = make_ushort4((unsigned short)(r[j].z & 0xFFFFu), (unsigned short)(r[j].z >> 16), (unsigned short)(r[j].w & 0xFFFFu),
                                 (unsigned short)(r[j].w >> 16));
        }
    } else {
#pragma unroll
        for (int j = 0; j < kL1PerWave; ++j) {
            const uint32_t p = blk * kL1Items + (w * kL1PerWave + j) * WAVE + lane;
            nt[j] = p < a.n_bound ? a.tiles[p] : 0u;
            ids[j] = p;
        }
#pragma unroll
        for (int j = 0; j < kL1PerWave; ++j) {
            const uint32_t p = blk * kL1Items + (w * kL1PerWave + j) * WAVE + lane;
            tb[j] = make_ushort4(0, 0, 0, 0);
            key[j] = 0;
            if (nt[j] != 0) {
                tb[j] = a.aabb[p];
                key[j] = __float_as_uint(a.depth[p]);
            }
        }
    }
    __syncthreads();
    auto emit = [&](uint32_t bin, uint32_t k, uint32_t gid, ushort4 box) {
        const uint32_t pos = atomicAdd(&s_cur[bin], 1u);
        if (pos < a.capacity) {
            uint32_t* const rec = a.cand + (size_t)kCandWords * pos;  // three adjacent dwords: one 12-byte store
            rec[0] = k;
            rec[1] = gid;
            rec[2] = bin_local_box16(a.g, bin, box);
        }
    };
#pragma unroll
    for (int j = 0; j < kL1PerWave; ++j) {
        const uint32_t gid = ids[j];
        uint32_t x0 = 0, y0 = 0, x1 = 0, y1 = 0;
        if (nt[j] != 0) {
            x0 = tb[j].x >> a.g.bin_shift, y0 = tb[j].y >> a.g.bin_shift;
            x1 = ((tb[j].z - 1u) >> a.g.bin_shift) + 1u, y1 = ((tb[j].w - 1u) >> a.g.bin_shift) + 1u;
        }
        const bool big = (x1 - x0) * (y1 - y0) > kL1BigBox;  // see k_l1_hist
        if (!big)
            for (uint32_t y = y0; y < y1; ++y)
                for (uint32_t x = x0; x < x1; ++x) emit((y << a.g.grid_shift) | x, key[j], gid, tb[j]);
        for (uint64_t bm = __ballot(big); bm != 0; bm &= bm - 1) {  // one Gaussian at a time, one bin per lane
            const int src = __ffsll((unsigned long long)bm) - 1;
            const uint32_t bx0 = (uint32_t)__builtin_amdgcn_readlane((int)x0, src), by0 = (uint32_t)__builtin_amdgcn_readlane((int)y0, src);
            const uint32_t bw = (uint32_t)__builtin_amdgcn_readlane((int)x1, src) - bx0;
            const uint32_t cells = bw * ((uint32_t)__builtin_amdgcn_readlane((int)y1, src) - by0);
            const uint32_t bkey = (uint32_t)__builtin_amdgcn_readlane((int)key[j], src);
            const uint32_t bgid = (uint32_t)__builtin_amdgcn_readlane((int)gid, src);
            const uint32_t blo = (uint32_t)__builtin_amdgcn_readlane((int)((uint32_t)tb[j].x | ((uint32_t)tb[j].y << 16)), src);
            const uint32_t bhi = (uint32_t)__builtin_amdgcn_readlane((int)((uint32_t)tb[j].z | ((uint32_t)tb[j].w << 16)), src);
            const ushort4 bbox = make_ushort4((unsigned short)(blo & 0xFFFFu), (unsigned short)(blo >> 16), (unsigned short)(bhi & 0xFFFFu),
                                              (unsigned short)(bhi >> 16));
            for (uint32_t c = lane; c < cells; c += WAVE) emit(((by0 + c / bw) << a.g.grid_shift) | (bx0 + c % bw), bkey, bgid, bbox);
        }
    }
}

// Stage taps after a frame whose level 1 streamed the dense lists (k_preprocess wrote no planes): tiles_overlap, depth and the
// tile box of every visible Gaussian rebuilt from the lists' entries (the caller zeroes `tiles` first; the frame's list lengths
// are the words the blend left behind the counters).
__global__ __launch_bounds__(BLOCK) void k_vis_to_planes(const uint4* __restrict__ vis, const uint32_t* __restrict__ vis_count,
                                                         uint32_t region_slots, uint32_t* __restrict__ tiles, float* __restrict__ depth,
                                                         ushort4* __restrict__ aabb) {
    const uint32_t region = blockIdx.y, slot = blockIdx.x * BLOCK + threadIdx.x;
    const uint32_t count = min(vis_count[region * kVisCounterStride + 1], region_slots);
    if (slot >= count) return;
    const uint4 e = vis[(size_t)region * region_slots + slot];  // {id, depth bits, x0 | y0 << 16, x1 | y1 << 16}
    const uint32_t x0 = e.z & 0xFFFFu, y0 = e.z >> 16, x1 = e.w & 0xFFFFu, y1 = e.w >> 16;
    tiles[e.x] = (x1 - x0) * (y1 - y0);
    depth[e.x] = __uint_as_float(e.y);
    aabb[e.x] = make_ushort4((unsigned short)x0, (unsigned short)y0, (unsigned short)x1, (unsigned short)y1);
}
void launch_vis_to_planes(const AttrView& av, uint32_t n, hipStream_t s) {
    if (!av.vis || n == 0) return;
    (void)hipMemsetAsync(av.tiles, 0, (size_t)n * sizeof(uint32_t), s);
    hipLaunchKernelGGL(k_vis_to_planes, dim3((av.vis_region_slots + BLOCK - 1) / BLOCK, kVisRegions), dim3(BLOCK), 0, s, av.vis, av.vis_count,
                       av.vis_region_slots, av.tiles, av.depth, av.aabb);
}

static L1Args l1_args(const BinLaunch& b) {
    L1Args a;
    a.g = BinGrid{b.tiles_x, b.tiles_y, b.bins_x, b.bins_y, b.bin_shift, b.grid_shift};
    a.order = b.order;
    a.n_items = b.n_items;
    a.n_bound = b.n_bound;
    a.tiles = b.tiles;
    a.aabb = b.aabb;
    a.depth = b.depth;
    a.vis = b.vis;
    a.vis_count = b.vis_count;
    a.vis_region_slots = b.vis_region_slots;
    a.hist = b.hist;
    a.bin_count = b.bin_count;
    a.cand = b.cand;
    a.counters = b.counters;
    a.capacity = b.cand_capacity;  // (level 1 only ever writes candidates)
    // level-1 blocks: over the N items, or over the slots of the dense lists
    a.nblk = b.vis ? kVisRegions * (b.vis_region_slots / kL1Items) : bin_level1_blocks(b.n_bound);
    a.stamps = b.stamps;
    return a;
}

uint32_t bin_level1_blocks(uint32_t n_items) { return (n_items + kL1Items - 1) / kL1Items; }
uint32_t vis_region_slots(uint32_t n) {
    const uint32_t groups = (n + BLOCK - 1) / BLOCK, per_region = (groups + kVisRegions - 1) / kVisRegions;  // k_preprocess workgroups per list
    const uint32_t slots = per_region * BLOCK;
    return slots == 0 ? kL1Items : (slots + kL1Items - 1) / kL1Items * kL1Items;
}
uint32_t bin_level1_columns(uint32_t n_items) {
    const uint32_t dense = kVisRegions * (vis_region_slots(n_items) / kL1Items), planes = bin_level1_blocks(n_items);
    return dense > planes ? dense : planes;
}
static_assert(BLOCK >= (int)kVisRegions, "k_blend's first workgroup zeroes the list counters, k_l1_scan's sums them: a thread each");

void launch_bin_level1_count(const BinLaunch& b, hipStream_t s) {
    const L1Args a = l1_args(b);
    if (a.nblk == 0) return;
    if (b.grid_shift == 4) hipLaunchKernelGGL(k_l1_hist<4>, dim3(l1_grid(a.nblk)), dim3(BLOCK), 0, s, a);
    else hipLaunchKernelGGL(k_l1_hist<16>, dim3(l1_grid(a.nblk)), dim3(BLOCK), 0, s, a);
    hipLaunchKernelGGL(k_l1_scan, dim3((1u << (2 * b.grid_shift)) + 1u), dim3(BLOCK), 0, s, a);
}

void launch_bin_level1_scatter(const BinLaunch& b, bool any_order, hipStream_t s) {
    const L1Args a = l1_args(b);
    if (a.nblk == 0) return;
    if (any_order) {
        if (b.grid_shift == 4) hipLaunchKernelGGL(k_l1_scatter_any_order<4>, dim3(l1_grid(a.nblk)), dim3(BLOCK), 0, s, a);
        else hipLaunchKernelGGL(k_l1_scatter_any_order<16>, dim3(l1_grid(a.nblk)), dim3(BLOCK), 0, s, a);
    } else {
        if (b.grid_shift == 4) hipLaunchKernelGGL(k_l1_scatter<4>, dim3(l1_grid(a.nblk)), dim3(BLOCK), 0, s, a);
        else hipLaunchKernelGGL(k_l1_scatter<16>, dim3(l1_grid(a.nblk)), dim3(BLOCK), 0, s, a);
    }
}

}  // namespace gs
