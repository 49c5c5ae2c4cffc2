// gs_kernels.h -- launch wrappers of the gfx950 kernels (gs_kernels.hip).
// Host-side declarations only; everything takes raw device pointers + a stream.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/gs3d_hip.h"

namespace gs {

constexpr int kTile = 16;             // TILE_WIDTH == TILE_HEIGHT, common.glsl:1-2
constexpr int kSortTileKeys = 2048;   // keys per radix tile (256 threads x 8)
constexpr int kSortMaxBlocks = 1024;  // fixed sort grid (data-dependent sizes stay on the device)
constexpr int kScanBlocks = 512;      // fixed scan grid

// Packed scene blob (59 floats per Gaussian + padding): planes 0..10 are SoA with a plane stride of
// blob_stride(n) = n rounded up to 16 floats -- plane p of Gaussian i is blob[p*stride + i] -- followed by the SH
// block as AoS: the 48 SH floats of Gaussian i are blob[11*stride + 48*i .. +48).  With the blob itself 64-byte
// aligned, every plane and every Gaussian's 192-byte SH block start on a 64-byte line for any n.
enum ScenePlane { P_POS = 0, P_SCALE = 3, P_ROT = 6, P_OPACITY = 10, P_SH = 11, P_COUNT = 59 };
constexpr uint64_t blob_stride(uint64_t n) { return (n + 15) & ~uint64_t(15); }
constexpr uint64_t blob_floats(uint64_t n) { return P_SH * blob_stride(n) + 48 * n; }

struct SceneView {
    const float* blob;   // 11 planes of `stride` floats, then n x 48 SH floats
    const float* cov3d;  // 6 planes of n floats
    uint32_t n;
    uint32_t stride;     // blob_stride(n)
};

// Per-frame buffers indexed by Gaussian id.
struct AttrView {
    uint32_t* tiles;     // tiles_overlap (0 = culled)
    float* depth;
    float* radius;
    ushort4* aabb;
    float4* conic_op;    // c00 c01 c11 opacity
    float4* uv_rg;       // u v r g
    float* b;
};

// Device-resident frame counters.
struct Counters {
    uint32_t visible;    // V
    uint32_t instances;  // D (may exceed capacity: then `overflow` is set and nothing past capacity is written)
    uint32_t overflow;   // bit 0: a capacity (lists, candidates, chunk table) was exceeded; bit 1: a bin outgrew k_bin_sort
    uint32_t bin_entries;  // E1: (bin, Gaussian) candidates of the level-1 binning
    uint32_t max_bin;    // candidates in the fullest bin
    uint32_t pad;
};
constexpr int kBinSortMax = 16384;  // candidates per bin that k_bin_sort can order in LDS (128 KiB of (key, id))

void launch_cov3d(const float* blob, float* cov3d, uint32_t n, uint32_t stride, hipStream_t s);
// counters (nullable): the kernel clears counters->overflow, so that a frame needs no memset node.
// nbins (nullable): [N] bins of (1 << bin_shift)^2 tiles touched by each Gaussian's tile box, 0 when culled.
void launch_preprocess(const SceneView& sv, const gs_uniforms& u, const AttrView& av, Counters* counters,
                       uint32_t* nbins, int bin_shift, hipStream_t s);
// Bin-local depth order: bin-major candidate ids (index order inside a bin) -> (depth bits, id) order inside each bin.
void launch_bin_sort(const uint32_t* bin_count, const uint32_t* ids_in, const float* depth, uint32_t* ids_out,
                     Counters* counters, uint32_t bins, hipStream_t s);

// Stable LSD radix pass on (u32 key, u32 value) pairs, 8-bit digit at `shift`.
//   first != 0: the input is (key = bits(depth[i]), value = i) for every i < n_static with tiles[i] != 0
//               (compaction folded into the first pass); the element count is written to *n_out.
//   first == 0: the element count is read from *n_in (device memory).
// scratch: block histograms, 256 * (blocks + 1) uint32.
struct RadixPass {
    const uint32_t* keys_in;
    const uint32_t* vals_in;
    uint32_t* keys_out;
    uint32_t* vals_out;
    const uint32_t* n_in;     // device count (first == 0)
    uint32_t n_static;        // N (first != 0) or the capacity bound used to size the grid
    const uint32_t* tiles;    // first != 0
    uint32_t* n_out;          // first != 0
    uint32_t* block_hist;     // [256][blocks]
    uint32_t* digit_total;    // [256]
    int shift;
    int bits;                 // significant bits in this digit (<= 8)
    int blocks;
    int first;
    const ushort4* gather_aabb;    // last depth pass: per sorted Gaussian, the number of (tile >> bin_shift) bins
    int bin_shift;                 // its tile box touches ...
    uint32_t* tiles_sorted;        // ... is written here (may be null)
};
void launch_radix_pass(const RadixPass& p, hipStream_t s);

// Exclusive scan of cnt[0..*n) -> off, total -> *total_out (2 kernels, fixed grid).  n == nullptr: n_bound entries.
// partial: 2 * kScanBlocks uints.  nonzero_out (nullable): number of non-zero entries.
void launch_exclusive_scan(const uint32_t* cnt, uint32_t* off, const uint32_t* n, uint32_t n_bound,
                           uint32_t* partial, uint32_t* total_out, uint32_t* nonzero_out, hipStream_t s);

// preprocess_sort.comp counterpart, in depth order: for j < *n_visible, g = order[j], writes
// tile ids (x outer, y inner) and g at off[j]...  Sets counters->overflow when D > capacity.
// order == nullptr: g = j; n_visible == nullptr: n_bound entries (entries with a zero count emit nothing).
void launch_duplicate(const uint32_t* order, const uint32_t* off, const uint32_t* tiles_sorted,
                      const ushort4* aabb, const uint32_t* n_visible, uint32_t n_bound, uint32_t tiles_x,
                      int shift, uint32_t capacity, uint32_t* inst_tile, uint32_t* inst_gid, Counters* counters,
                      hipStream_t s);

// Level 2 of the hierarchical binning (k_bin_count, k_bin_scan, k_tile_scan, k_bin_fill): expands the
// bin-major candidate list into per-tile lists, the tile ranges and D.
struct BinLaunch {
    const uint32_t* cand;
    const uint32_t* bin_count;  // [256]
    const ushort4* aabb;
    uint32_t* chunk_hist;       // [max_chunks][S*S]
    uint32_t* tile_total;       // [T], zero-filled by the caller
    uint32_t* ranges;           // [T][2]
    uint32_t* sorted_gid;
    Counters* counters;
    uint32_t capacity;
    uint32_t tiles_x, tiles_y, bins_x, bins;
    int shift;
    uint32_t max_chunks;
};
void launch_bin_ranges(const BinLaunch& b, hipStream_t s);  // k_bin_count, k_bin_scan, k_tile_scan
void launch_bin_fill(const BinLaunch& b, hipStream_t s);    // k_bin_fill

// render.comp counterpart.
// tile_order[b] = the tile workgroup b renders (a permutation of the tiles)
void launch_blend(const uint32_t* ranges, const uint32_t* sorted_gid, const uint32_t* tile_order, const AttrView& av,
                  uint32_t width,
                  uint32_t height, float* rgba, uint8_t* bgra, const Counters* counters,
                  Counters* host_counters /* pinned, nullable: *host_counters = *counters */, hipStream_t s);

}  // namespace gs
