// gs_kernels.h -- launch wrappers of the gfx950 kernels (gs_scene / gs_preprocess / gs_radix / gs_bin_l1 / gs_bin_l2 / gs_blend .hip).
// Host-side declarations only; everything takes raw device pointers + a stream.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/gs3d_hip.h"

namespace gs {

constexpr int kTile = 16;             // TILE_WIDTH == TILE_HEIGHT, common.glsl:1-2
constexpr int kSortTileKeys = 2048;   // keys per radix tile (256 threads x 8)
constexpr int kSortMaxBlocks = 1024;  // fixed sort grid (data-dependent sizes stay on the device)

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
    const uint16_t* sh16;  // nullable: the SH block as binary16, n x 48 (gs_scene_quantize_sh); read instead of the fp32 one
    const float* acut;   // n floats: the alpha cut of every Gaussian (launch_alpha_cut), a function of its opacity, computed at load
    // Nullable.  The arrays above may hold the Gaussians in ANOTHER ORDER than the scene's (gs_scene: spatial / Morton order, so
    // that a wave's 64 Gaussians are neighbours in space: culled together, fetched together, binned together): Gaussian j of them
    // is Gaussian perm[j] of the scene, and everything a frame writes (tiles, depth, boxes, records, list entries) goes by THAT id.
    const uint32_t* perm;
};

// Per-frame buffers indexed by Gaussian id.
// What the blend gathers per list entry -- conic + opacity, uv + r g, b: 36 bytes -- lives in ONE 64-byte, 64-byte
// aligned record per Gaussian, so that an entry costs the memory system one line instead of three (three separate
// arrays measured 10 % slower in the blend and 9 % in the frame rate: the gathers are request-bound, not byte-bound).
// The record is the reference's VertexAttribute (common.glsl:42-49) in spirit; what the binning kernels stream or
// gather on their own (tiles, tile box, depth) additionally lives in dense arrays.
struct AttrRecord {
    float4 conic_op;   // c00 c01 c11 opacity
    float4 uv_rg;      // u v r g
    float4 b_depth_r;  // b, depth, radius, alpha cut (the most negative `power` at which render.comp:78 keeps the entry)
    uint4 pad_;        // zeros (k_preprocess writes a record as one full 64-byte line); the blend never reads it
};
static_assert(sizeof(AttrRecord) == 64, "one line per Gaussian");

struct AttrView {
    uint32_t* tiles;     // tiles_overlap (0 = culled)      } written by k_preprocess only when `vis` is null (the plane-input
    float* depth;        //                                 } level 1, the global path, bins of >= 16 x 16 tiles read them);
    ushort4* aabb;       //                                 } with the dense lists: rebuilt on demand for the stage taps
    AttrRecord* rec;     // written for visible Gaussians only
    // Optional (null: not written): the frame's visible Gaussians as DENSE lists in any order, 16 bytes each --
    // {id, depth bits, tile box x0 | y0 << 16, x1 | y1 << 16} -- which the level-1 kernels of the bin-local path stream
    // instead of walking the N-wide planes above (half of whose lanes are culled) with a dependent gather behind them.
    // kVisRegions lists of vis_region_slots slots each: workgroup b of k_preprocess appends to list b % kVisRegions, a wave
    // taking its run of slots with ONE atomic add on the list's counter, vis_count[region * kVisCounterStride].  (One list
    // with one counter was measured first: 94 k same-address atomics per frame at 6 M Gaussians retire at ~10 ns each and
    // k_preprocess took 1.1 ms instead of 0.27.)  The frame's last kernel (k_blend) zeroes the counters for the next frame.
    uint4* vis;
    uint32_t* vis_count;
    uint32_t vis_region_slots;
};
constexpr uint32_t kVisRegions = 256, kVisCounterStride = 32;  // a counter per 128 bytes
// slots per list for a scene of n Gaussians: what the workgroups (of 256) that append to it can hold, a whole number of level-1 blocks
uint32_t vis_region_slots(uint32_t n);

// Device-resident frame counters.
struct Counters {
    uint32_t visible;    // V
    uint32_t instances;  // D (may exceed capacity: then `overflow` is set and nothing past capacity is written)
    uint32_t overflow;   // bit 0: the capacity (candidates or lists) was exceeded; bit 1: a bin outgrew the in-LDS order of its level
    uint32_t bin_entries;  // E1: (bin, Gaussian) candidates of the level-1 binning
    uint32_t max_bin;    // candidates in the fullest bin
    uint32_t slabs;      // depth-slab descriptors written by k_bin_slabs for k_slab_work (level 4)
    // written BY the guarded blend (the host copy, published when the blend starts, does not hold them; gs_get_stats reads the device words):
    uint32_t blend_resolved;  // break decisions (render.comp:83) inside the guard's window, resolved by replaying the pixel exactly
    uint32_t blend_redo;      // quadrants (8 x 8 px) abandoned and re-rendered whole with the reference's arithmetic
    // the work queue of k_bin_queue (level 4 in one launch): items handed out so far (bins first, then depth slabs); bins beyond a slab planned so far
    uint32_t q_head, q_bins_done;
};
// The frame's timeline, stamped by the kernels themselves (round 6).  Every pass's first kernel writes the constant-rate clock
// (wall_clock64: s_memrealtime, hipDeviceAttributeWallClockRate kHz) at its start -- thread 0 of workgroup 0 -- and a one-wave kernel
// behind the blend stamps the frame's end and copies the stamps to pinned host memory.  Round 5 bracketed the passes with hipEvents
// (the reference's vkCmdWriteTimestamp, Renderer.cpp:484-526): each record is a barrier packet the next dispatch waits for, 4.5 us
// of idle GPU apiece, 22 us of a 290-us frame rendered alone (profiles/r06_seams.txt); two kernels with nothing between them start
// back to back.  A span therefore runs from a pass's first kernel's start to the next pass's: the seam belongs to the pass before it.
enum FrameStamp { ST_PREPROCESS = 0, ST_ORDER = 1, ST_L1_COUNT = 2, ST_L1_SCATTER = 3, ST_L2 = 4, ST_BLEND = 5, ST_END = 6, ST_COUNT = 8 };

// What changes from one frame to the next.  Normally these travel as kernel arguments; when a frame is replayed
// as a captured HIP graph (gs_set_graph_mode) they are read from this block in device memory instead, which the
// host refreshes with one small copy ahead of the graph launch -- the graph itself never needs re-recording.
struct FrameParams {
    gs_uniforms u;
    float* rgba;
    uint8_t* bgra;
    Counters* host_counters;
    uint64_t* host_stamps;  // pinned, [ST_COUNT]
};

// candidates per bin that k_bin_fast orders in LDS at depth-order level 0 .. 3 (8 bytes of LDS each); level 4: k_bin_slabs,
// bins of up to 65535 taken in depth slabs of <= 12288; level 5 = the global path
constexpr int kBinSortLevels = 5;
constexpr int kBinSlabLevel = 4;
constexpr uint32_t kBinSortLimit[kBinSortLevels] = {4096, 8192, 12288, 16384, 65535};
constexpr int kBinSortMax = 16384;
constexpr uint32_t kSlabDescBytes = 288, kSlabCapacity = 8192, kSlabWorkGroups = 512, kQueueWorkGroups = 256;

void launch_cov3d(const float* blob, float* cov3d, uint32_t n, uint32_t stride, hipStream_t s);
// cut[i] = the most negative power <= 0 with !(min(0.99, opacity[i] * expf(power)) < 1/255)  (render.comp:77-78; +inf: none,
// -inf: all), exact for libm's expf: the blend compares `power` with it instead of alpha with 1/255 (gs_device.h: alpha_cut)
// *beyond_unit (device memory, nullable, zeroed by the caller): set to 1 if any opacity exceeds 1
void launch_alpha_cut(const float* blob, float* cut, uint32_t n, uint32_t stride, uint32_t* beyond_unit, hipStream_t s);
// dst = the blob with its Gaussians in the order perm (dst Gaussian j = src Gaussian perm[j]): 11 planes + the SH block
void launch_permute_blob(const float* src, const uint32_t* perm, float* dst, uint32_t n, uint32_t stride, hipStream_t s);
// *out (device) = sum of the blob's 32-bit patterns, as 64-bit integers (wrapping): what gs_dist_verify compares across ranks
void launch_blob_checksum(const float* blob, uint64_t floats, uint64_t* out, hipStream_t s);
// fp32 SH block of the blob -> binary16 (round to nearest even), n x 48 values
void launch_sh_to_half(const float* blob, uint16_t* sh16, uint32_t n, uint32_t stride, hipStream_t s);
// counters (nullable): the kernel clears the frame's counters, so that a frame needs no memset node.
// fp (nullable, device memory): read the uniforms / output pointers from it instead of the arguments (graph replay)
// stamps (nullable, device memory, [ST_COUNT]): the frame's timeline, see FrameStamp
void launch_preprocess(const SceneView& sv, const gs_uniforms& u, const AttrView& av, Counters* counters,
                       const FrameParams* fp, uint64_t* stamps, hipStream_t s);
// behind the frame's last kernel: stamps[ST_END] = now, the stamps copied to host_stamps (pinned; fp non-null: fp->host_stamps)
void launch_frame_end(uint64_t* stamps, uint64_t* host_stamps, const FrameParams* fp, hipStream_t s);

// Stable LSD radix pass on (u32 key, u32 value) pairs, 8-bit digit at `shift` (the global depth order).
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
    uint64_t* stamps;         // nullable: the pass's first kernel stamps [ST_ORDER] (the caller sets it on the first pass only)
};
void launch_radix_pass(const RadixPass& p, hipStream_t s);

// Two-level binning (gs_bin_l1.hip, gs_bin_l2.hip): level 1 lists the items (Gaussians in index order, or the visible Gaussians in
// depth order when `order` is given) per bin of S x S tiles; level 2 turns each bin's list into the per-tile lists
// (ordering it by (depth bits, id) in LDS first when sort is set), the tile ranges and D.
struct BinLaunch {
    const uint32_t* order;      // null: item p = Gaussian p; else Gaussian order[p]
    const uint32_t* n_items;    // device-resident item count (null: n_bound)
    uint32_t n_bound;           // N: sizes the grid and the hist table
    const uint32_t* tiles;      // [N]
    const ushort4* aabb;        // [N]
    const float* depth;         // [N]
    const uint4* vis;           // null, or (any-order scatter only) AttrView::vis / vis_count / vis_region_slots: the items are the lists' entries
    uint32_t* vis_count;
    uint32_t vis_region_slots;
    uint32_t* hist;             // [padded bins + 1][bin_level1_columns(n_bound)] (the last row: visible items per block)
    uint32_t* bin_count;        // [2048]: per padded bin its candidate count, then (from 1024 on) its offset in the candidate buffer
    uint32_t* cand;             // [cand_capacity] records (3 words each) or ids
    uint32_t* ranges;           // [T][2]
    uint32_t* sorted_gid;       // [capacity (+4)]
    Counters* counters;
    uint32_t capacity;          // tile instances the lists hold
    uint32_t cand_capacity;     // level-1 candidates the candidate buffer holds
    void* slabs;                // [slab_capacity] 288-byte depth-slab descriptors (level 4)
    uint32_t slab_capacity;
    uint32_t slab_epoch;        // != 0: level 4 as ONE launch (k_bin_queue); a value no earlier launch on these descriptors carried
                                // (a descriptor is ready when it holds this launch's epoch).  0: k_bin_slabs + k_slab_work
    uint32_t tiles_x, tiles_y, bins_x, bins_y;
    int bin_shift;              // log2 S
    int grid_shift;             // 4 or 5: padded bin id = by << grid_shift | bx
    uint64_t* stamps;           // nullable: the frame's timeline (FrameStamp); level 1 and level 2 stamp their starts
};
// tiles / depth / aabb of the visible Gaussians rebuilt from the dense lists of the frame that just ran (stage taps: k_preprocess
// writes no planes when level 1 streams the lists); av.vis must be the lists of that frame
void launch_vis_to_planes(const AttrView& av, uint32_t n, hipStream_t s);
uint32_t bin_level1_blocks(uint32_t n_items);
uint32_t bin_level1_columns(uint32_t n_items);  // columns of the hist table: level-1 blocks over the planes or over the dense lists, whichever is more
void bin_debug_occupancy();
hipError_t bin_prepare_device();                                    // once per device, before the first launch_bin_level2
void launch_bin_level1_count(const BinLaunch& b, hipStream_t s);    // k_l1_hist, k_l1_scan
// any_order: the candidates of a bin may land in any order inside each block's run (the bin-local path with bins of
// <= 8 x 8 tiles orders them by (depth, id) anyway); otherwise item order is kept
void launch_bin_level1_scatter(const BinLaunch& b, bool any_order, hipStream_t s);  // k_l1_scatter / k_l1_scatter_any_order
// level 0 .. 4: the bin's candidates are ordered by (depth bits, id) in LDS, up to kBinSortLimit[level] per bin (level 4 in
// several slabs); level 5: no ordering (the candidates arrive in depth order: global path), any bin size
void launch_bin_level2(const BinLaunch& b, int level, hipStream_t s);  // k_bin_build

// render.comp counterpart.
// tile_order[b] = the tile workgroup b renders (a permutation of the tiles)
void launch_blend(const uint32_t* ranges, const uint32_t* sorted_gid, const uint32_t* tile_order, const AttrView& av,
                  uint32_t width,
                  uint32_t height, float* rgba, uint8_t* bgra, Counters* counters,
                  Counters* host_counters /* pinned, nullable: *host_counters = *counters */,
                  int exp_mode /* 0 pipeline polynomial, 1 v_exp_f32, 2 libm's expf restated in binary64, 3 v_exp_f32 under the guard
                                  of render.comp:82 (break decisions near the cut taken from mode 2's arithmetic; needs every
                                  opacity <= 1: the caller passes 2 for a scene that holds a larger one) */,
                  bool contract /* the three FMA contractions GLSL permits in render.comp:66,87, or (default) none */,
                  const FrameParams* fp,
                  bool lockstep /* the four waves of a tile take every chunk of its list together (s_barrier): gs_blend.hip */,
                  uint64_t* stamps /* nullable: [ST_BLEND] = the kernel's start */,
                  hipStream_t s);

}  // namespace gs
