// gs_bin.h -- device-side pieces shared by the two levels of the tile binning (gs_bin_l1.hip, gs_bin_l2.hip).
#pragma once
#include "gs_device.h"

namespace gs {

#ifndef GS_DPP_TRANSPOSE
#define GS_DPP_TRANSPOSE 1
#endif

// ---------------------------------------------------------------------------------------
// Two-level binning: per-tile depth-ordered lists without sorting the D instances.
//
// The screen is cut into a grid of at most 32 x 32 bins of S x S tiles (bin id = by * GW + bx on a padded grid of
// width GW = 16 or 32).  LEVEL 1 lists, per bin, the items whose tile box touches it, in item order (items are the
// Gaussians in index order on the bin-local path, or the visible Gaussians in depth order on the global path):
//     k_l1_hist     per block of 1024 items, how many touch each bin           hist[bin][block]
//     k_l1_scan     per bin, exclusive prefix over the blocks + the bin total  (one workgroup per bin)
//     k_l1_scatter  each block appends its items to the bins' lists at  bin offset + block prefix + rank in block
// LEVEL 2 (k_bin_build, one 1024-thread workgroup per bin) puts the bin's candidates into (depth bits, id) order in
// LDS (bin-local path; on the global path they already are), counts how many cover each of the bin's S*S tiles,
// takes a segment of the list buffer for the bin (one atomic add: the lists are bin-major, the tiles of a bin
// consecutive), writes the tile ranges, and appends every candidate to the lists of the tiles it covers, in order.
//
// Both levels use the same primitive.  A wave takes 64 consecutive items; lane k turns item k's box into coverage
// words over the cells (bins at level 1, tiles at level 2); a 64 x 64 bit-matrix transpose across the wave gives
// lane c the column of cell c: which of the 64 items cover it, in item order.  Counting is a popcount; appending
// walks the set bits.  Order is preserved at every step (blocks in order, waves in order, lanes in order), so each
// tile's list equals the reference's stably sorted payload (preprocess_sort.comp + the 8 radix passes) and the
// ranges equal tile_boundary.comp's up to the position of the lists in the buffer -- while the instance data moved
// through HBM drops from ~ 8 passes x 24 B x D to ~ 4 B x D.
// ---------------------------------------------------------------------------------------

// Coverage of a box [x0,x1) x [y0,y1) as bit masks over a (1 << shift)-wide grid of cells: cell c = y * W + x lives
// in bit (c % 64) of word (c / 64), i.e. lane (c % 64) "owns" cell c in register slot c / 64.
template <int R>
__device__ __forceinline__ void cover_masks(int shift, int x0, int y0, int x1, int y1, uint64_t (&m)[R]) {
    const int W = 1 << shift;
    const int rows_per_word = 64 >> shift;  // 16, 8, 4, 2 for W = 4, 8, 16, 32
    const uint64_t rowbits = x1 > x0 ? ((x1 - x0 >= 64 ? ~0ull : ((1ull << (x1 - x0)) - 1ull)) << x0) : 0ull;
#pragma unroll
    for (int r = 0; r < R; ++r) {
        uint64_t w = 0;
        for (int yy = 0; yy < rows_per_word; ++yy) {
            const int y = r * rows_per_word + yy;
            if (y >= y0 && y < y1) w |= rowbits << (yy * W);
        }
        m[r] = w;
    }
}

// 64 x 64 bit-matrix transpose across a wave: lane k enters with row k, lane t leaves with column t
// (bit k of the result = bit t of lane k's input).  Six butterfly steps; step j swaps the off-diagonal j x j
// blocks between lanes l and l ^ j.  ds_swizzle is a lane permutation inside 32-lane halves (no LDS memory).
// lane ^ J exchange.  J = 1, 2: one DPP quad permutation; J = 4: half-row mirror then quad reversal; J = 8: row mirror
// then half-row mirror (DPP modifiers ride on VALU moves: full rate, no LDS pipe); J = 16: ds_swizzle, a lane
// permutation inside 32-lane halves that goes through the LDS pipe (no memory) -- with ten of them per transpose
// that pipe was what bounded every kernel built on the transpose, hence the DPP forms for the four short strides.
template <int J>
__device__ __forceinline__ uint32_t lane_xor(uint32_t x) {
#if GS_DPP_TRANSPOSE
    if (J == 1) return (uint32_t)__builtin_amdgcn_mov_dpp((int)x, 0xB1, 0xF, 0xF, true);   // quad_perm [1,0,3,2]
    if (J == 2) return (uint32_t)__builtin_amdgcn_mov_dpp((int)x, 0x4E, 0xF, 0xF, true);   // quad_perm [2,3,0,1]
    if (J == 4) {
        const int t = __builtin_amdgcn_mov_dpp((int)x, 0x141, 0xF, 0xF, true);             // row_half_mirror: i -> 7 - i
        return (uint32_t)__builtin_amdgcn_mov_dpp(t, 0x1B, 0xF, 0xF, true);                 // quad_perm [3,2,1,0]
    }
    if (J == 8) {
        const int t = __builtin_amdgcn_mov_dpp((int)x, 0x140, 0xF, 0xF, true);             // row_mirror: i -> 15 - i
        return (uint32_t)__builtin_amdgcn_mov_dpp(t, 0x141, 0xF, 0xF, true);                // row_half_mirror
    }
#endif
    return (uint32_t)__builtin_amdgcn_ds_swizzle((int)x, (J << 10) | 0x1F);  // lane ^ J
}
template <int J>
__device__ __forceinline__ uint32_t transpose_step(uint32_t x, bool up) {
    constexpr uint32_t MASK = J == 16 ? 0x0000FFFFu : J == 8 ? 0x00FF00FFu : J == 4 ? 0x0F0F0F0Fu
                            : J == 2 ? 0x33333333u : 0x55555555u;
    const uint32_t p = lane_xor<J>(x);
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

// Lane c owns a cell's column `col` (which of the wave's 64 items cover the cell, bit k = item k) and the cell's
// list cursor `cur` (in entries).  Appends ids[k] for every set bit, in order: up to four ids per lane and round,
// written with ONE store of 4..16 bytes -- every lane's store is its own request to the L2 (a different line per
// lane), so what bounds this is the number of requests, not of bytes.  One store instruction per size; lanes of
// another size are given an out-of-range offset, which the hardware's bounds check drops (the same check enforces
// the list capacity) without a branch.
__device__ __forceinline__ void walk_column(uint64_t col, uint32_t cur, __amdgpu_buffer_rsrc_t out,
                                            const uint32_t* __restrict__ ids /* wave-private [64] */) {
    typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
    typedef uint32_t u32x3 __attribute__((ext_vector_type(3)));
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
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

constexpr int kL1Items = 1024;   // items per level-1 block: 16 chunks of 64, four per wave of a 256-thread workgroup
constexpr int kL1Chunks = kL1Items / WAVE;
constexpr int kL1PerWave = kL1Chunks / (BLOCK / WAVE);
constexpr uint32_t kL1BigBox = 12;  // bins: beyond this a Gaussian's bin box is emitted by the whole wave, one bin per lane

struct BinGrid {
    uint32_t tiles_x, tiles_y;  // tiles of the screen
    uint32_t bins_x, bins_y;    // bins of the screen (<= 32 each)
    int bin_shift;              // log2 S: a bin is S x S tiles
    int grid_shift;             // log2 GW: padded bin id = by << grid_shift | bx
};

// bin_count[0 .. 1024): candidates per (padded) bin, by k_l1_scan; bin_count[kBinOffsets + b]: the exclusive prefix of those, by block 0 of
// the scatter kernels (round 6: every level-2 workgroup used to redo the 1024-wide scan -- two block scans, 1.7 us of a bin's 35)
constexpr uint32_t kBinOffsets = 1024;

struct L1Args {
    BinGrid g;
    const uint32_t* order;      // null: item p is Gaussian p; else item p is Gaussian order[p] (depth order)
    const uint32_t* n_items;    // device-resident item count; null: n_bound
    uint32_t n_bound;
    const uint32_t* tiles;      // [N] tiles_overlap (0 = culled)
    const ushort4* aabb;        // [N] tile boxes
    const float* depth;         // [N] (the record-emitting scatter only)
    const uint4* vis;           // null, or the dense lists of visible Gaussians (AttrView::vis): then the items are their entries
    const uint32_t* vis_count;
    uint32_t vis_region_slots;  // slots per list = kL1Items x (level-1 blocks per list)
    uint32_t* hist;             // [bins (padded)][nblk]
    uint32_t* bin_count;        // [bins (padded)]
    uint32_t* cand;             // [capacity] bin-major candidate Gaussian ids
    Counters* counters;
    uint32_t capacity;
    uint32_t nblk;
    uint64_t* stamps;           // nullable: the frame's timeline (k_l1_hist stamps ST_L1_COUNT, the scatter kernels ST_L1_SCATTER)
};

__device__ __forceinline__ bool bin_on_screen(const BinGrid& g, uint32_t bin) {
    return (bin & ((1u << g.grid_shift) - 1u)) < g.bins_x && (bin >> g.grid_shift) < g.bins_y;
}

// item p -> Gaussian id and its box in bin coordinates packed x0 | y0 << 8 | x1 << 16 | y1 << 24 (upper bounds
// exclusive; 0 = culled or absent: covers nothing)
__device__ __forceinline__ uint32_t l1_item(const L1Args& a, uint32_t p, uint32_t n, uint32_t& box_out) {
    uint32_t gid = 0;
    box_out = 0;
    if (p < n) {
        gid = a.order ? a.order[p] : p;
        if (a.tiles[gid] != 0) {
            const ushort4 box = a.aabb[gid];
            const uint32_t x0 = box.x >> a.g.bin_shift, y0 = box.y >> a.g.bin_shift;
            const uint32_t x1 = ((box.z - 1u) >> a.g.bin_shift) + 1u, y1 = ((box.w - 1u) >> a.g.bin_shift) + 1u;
            box_out = x0 | (y0 << 8) | (x1 << 16) | (y1 << 24);
        }
    }
    return gid;
}
// Dense lists: level-1 block `blk` covers slots [first, first + kL1Items) of list blk / (blocks per list); returns how many of
// them hold an entry (0: the block has nothing to do, and its cells of the table are never read).
__device__ __forceinline__ uint32_t l1_vis_block(const L1Args& a, uint32_t blk, uint32_t& first) {
    const uint32_t per = a.vis_region_slots / kL1Items, region = blk / per, at = (blk % per) * kL1Items;
    uint32_t cnt = (uint32_t)__builtin_amdgcn_readfirstlane((int)a.vis_count[region * kVisCounterStride]);
    if (cnt > a.vis_region_slots) cnt = a.vis_region_slots;  // (cannot happen: a list holds what its workgroups can append)
    first = region * a.vis_region_slots + at;
    return cnt > at ? (cnt - at < (uint32_t)kL1Items ? cnt - at : (uint32_t)kL1Items) : 0u;
}
// tile box of a dense-list entry {id, depth bits, x0 | y0 << 16, x1 | y1 << 16} -> box in bin coordinates, packed like l1_item's
__device__ __forceinline__ uint32_t l1_bin_box(const L1Args& a, uint4 r) {
    const uint32_t x0 = (r.z & 0xFFFFu) >> a.g.bin_shift, y0 = (r.z >> 16) >> a.g.bin_shift;
    const uint32_t x1 = (((r.w & 0xFFFFu) - 1u) >> a.g.bin_shift) + 1u, y1 = (((r.w >> 16) - 1u) >> a.g.bin_shift) + 1u;
    return x0 | (y0 << 8) | (x1 << 16) | (y1 << 24);
}
template <int R>
__device__ __forceinline__ void packed_cover_masks(int shift, uint32_t box, uint64_t (&m)[R]) {
    cover_masks<R>(shift, (int)(box & 255u), (int)((box >> 8) & 255u), (int)((box >> 16) & 255u), (int)(box >> 24), m);
}

// one coverage word (cells 64 r .. 64 r + 63) of a packed box: what cover_masks computes, for a run-time r
__device__ __forceinline__ uint64_t cover_word(int shift, uint32_t box, int r) {
    const int x0 = (int)(box & 255u), y0 = (int)((box >> 8) & 255u), x1 = (int)((box >> 16) & 255u), y1 = (int)(box >> 24);
    const int W = 1 << shift, rows_per_word = 64 >> shift;
    const uint64_t rowbits = x1 > x0 ? (((1ull << (x1 - x0)) - 1ull) << x0) : 0ull;  // x1 - x0 <= 32 on these grids
    uint64_t w = 0;
    for (int yy = 0; yy < rows_per_word; ++yy) {
        const int y = r * rows_per_word + yy;
        if (y >= y0 && y < y1) w |= rowbits << (yy * W);
    }
    return w;
}

// Which block of 1024 items a workgroup takes.  Workgroup b runs on XCD b % 8 (the dispatch rule the blend's tile order
// relies on too) and every XCD has its own L2.  What the level-1 kernels write is fine-grained and block-major: a block's
// 4-byte cell in each bin's row of the table (16 consecutive blocks to a line) and its run of a record or two in each bin's
// list (a line holds 5 records).  Dealt out in launch order, the blocks that share a line sit on eight different XCDs and
// every L2 evicts its own partial copy of it: WRITE_SIZE was 4 x the bytes stored at 6 M Gaussians.  So runs of
// kL1XcdRun consecutive blocks go to the SAME XCD (the lines are completed in one L2), and the XCDs still advance through
// the scene side by side.  The grid is rounded up to whole rounds of 8 runs; the blocks past the end return at once.
#ifndef GS_L1_XCD_RUN
#define GS_L1_XCD_RUN 32
#endif
constexpr uint32_t kL1XcdRun = GS_L1_XCD_RUN;  // 0: workgroup b takes block b
__host__ __device__ constexpr uint32_t l1_grid(uint32_t nblk) {
    return kL1XcdRun == 0 ? nblk : (nblk + 8u * kL1XcdRun - 1u) / (8u * kL1XcdRun) * (8u * kL1XcdRun);
}
__device__ __forceinline__ uint32_t l1_block() {
    if (kL1XcdRun == 0) return blockIdx.x;
    const uint32_t xcd = blockIdx.x & 7u, j = blockIdx.x >> 3;
    return ((j / kL1XcdRun) * 8u + xcd) * kL1XcdRun + j % kL1XcdRun;
}

// Bin-local path: the order of a bin's candidates does not matter (k_bin_fast orders them by (depth bits, id), a total
// order), so a block's items take their slots in its run of each bin's list with LDS atomics -- no ranking at all.
// What lands in the list is a 12-byte RECORD per candidate -- {depth bits, Gaussian id, tile box clipped to the bin} --
// everything level 2 needs, so that k_bin_fast STREAMS its bin's candidates with one coalesced read instead of gathering
// depth[id] (a 64-byte line per 4-byte value) and, after the sort, aabb[id] again (round 2: 2.3x the algorithmic bytes).
constexpr int kCandWords = 3;  // {key, id, box16}
// a tile box clipped to a bin of S <= 8 tiles, bin-local, inclusive upper bounds, 4 bits each: x0 | y0 << 4 | x1 << 8 | y1 << 12
__device__ __forceinline__ uint32_t bin_local_box16(const BinGrid& g, uint32_t bin, ushort4 box) {
    const int S = 1 << g.bin_shift;
    const int ox = (int)(bin & ((1u << g.grid_shift) - 1u)) << g.bin_shift, oy = (int)(bin >> g.grid_shift) << g.bin_shift;
    const int lx0 = max((int)box.x, ox) - ox, ly0 = max((int)box.y, oy) - oy;
    const int lx1 = min((int)box.z, ox + S) - ox - 1, ly1 = min((int)box.w, oy + S) - oy - 1;
    return (uint32_t)lx0 | ((uint32_t)ly0 << 4) | ((uint32_t)lx1 << 8) | ((uint32_t)ly1 << 12);
}

}  // namespace gs
