// gs_bin_l2.hip -- level 2 of the tile binning: (depth bits, id) order inside a bin, tile ranges, per-tile lists.
//
// Part of libgs3d_hip.so (gfx950 only).  Built with -ffp-contract=off: the floating-point contract of this path is "IEEE
// binary32, one rounding per operation, in the order the reference shader writes it" (DESIGN.md section 3); fused
// multiply-adds appear only where written explicitly.
// Reference restated (paths relative to /root/reference/src/shaders): the result of sort/hist.comp + sort/sort.comp x 8 inside a bin, tile_boundary.comp:22-50, the sorted payload
#include "gs_bin.h"

namespace gs {

// ---------------------------------------------------------------------------------------
// Level 2.  One workgroup per bin.
//   SORT: the bin's candidates arrive in Gaussian-index order and are put into (depth bits, id) order entirely in
//   LDS: four stable 8-bit LSD passes over (key, id) pairs, sorted in place because between the ranking and the
//   scatter of a pass every element lives in registers.  (wave, round, lane) order is list order; the stable rank
//   inside (wave, digit) comes from wave64 ballot matching.  Two sizes: 256 threads order up to 4096 candidates
//   (40 KiB of LDS: four workgroups per CU, every bin of a 1080p frame resident at once), 1024 threads up to 16384.
//   A bin beyond the size in use raises overflow bit 2 and the host re-runs the frame with the next size, and
//   beyond 16384 on the global depth-order path.
//   !SORT: the candidates already are in depth order (global path) and are streamed from memory, any number.
// ---------------------------------------------------------------------------------------
// One depth slab of a dense bin, as k_bin_slabs (planning) hands it to k_slab_work: which records to select from the bin's
// run and where each tile's list continues.
struct SlabDesc {
    uint32_t off, c_total;     // the bin's record run in the candidate buffer
    uint32_t kmin;             // bucket of a key = (key - kmin) >> sh
    int32_t sh;
    uint32_t b_lo, b_hi;       // the slab's buckets
    uint32_t ready;            // k_bin_queue: the epoch of the launch that wrote this descriptor (published with an agent-scope release)
    uint32_t pad;
    uint32_t cur[64];          // per tile of the bin: where this slab's entries go in the list buffer
};
static_assert(sizeof(SlabDesc) == 288, "descriptor layout");

struct BuildArgs {
    BinGrid g;
    uint32_t* cand;        // k_bin_build: [capacity] ids; k_bin_fast: [capacity] 12-byte records (rewritten in place in the degenerate-tie case)
    const uint32_t* bin_count;
    const float* depth;
    const ushort4* aabb;
    uint32_t* ranges;      // [T][2]
    uint32_t* sorted_gid;  // [capacity]
    Counters* counters;
    uint32_t capacity;       // tile instances the lists hold
    uint32_t cand_capacity;  // level-1 candidates the candidate buffer holds
    SlabDesc* slabs;       // k_bin_slabs -> k_slab_work (level 4)
    uint32_t slab_capacity;
    uint32_t epoch;        // k_bin_queue: see BinLaunch::slab_epoch
    uint64_t* stamps;      // nullable: the frame's timeline (the level's first kernel stamps ST_L2)
};

// candidate's tile box clipped to the bin, in bin-local tile coordinates (upper bounds exclusive), packed like l1_item's
__device__ __forceinline__ uint32_t bin_local_box(const BinGrid& g, uint32_t bin, ushort4 box) {
    const int S = 1 << g.bin_shift;
    const int ox = (int)(bin & ((1u << g.grid_shift) - 1u)) << g.bin_shift, oy = (int)(bin >> g.grid_shift) << g.bin_shift;
    const int lx0 = max((int)box.x, ox) - ox, ly0 = max((int)box.y, oy) - oy;
    const int lx1 = min((int)box.z, ox + S) - ox, ly1 = min((int)box.w, oy + S) - oy;
    return (uint32_t)lx0 | ((uint32_t)ly0 << 8) | ((uint32_t)lx1 << 16) | ((uint32_t)ly1 << 24);
}

constexpr int kBuildSlots = 16;  // chunks per fill round (one or four per wave)

#ifdef GS_BUILD_TIMING
// debug instrumentation (separate build, never the shipped library: make variant TAG=tm DEFS=-DGS_BUILD_TIMING, read by
// tools/build_timing.py and tools/slab_timing.py): the constant-rate clock at phase ends, one row per bin (rows 0 .. 511) or per
// depth slab (rows 512 + descriptor % 512; k_slab_work's prologue stamps row 1023)
__device__ unsigned long long g_build_t[1024][10];
#define BUILD_ROW(v) uint32_t build_row = (v)
#define BUILD_ROW_SET(v) build_row = (v)
#define BUILD_T(i) do { if (threadIdx.x == 0) g_build_t[build_row][i] = wall_clock64(); } while (0)
#else
#define BUILD_ROW(v) do { } while (0)
#define BUILD_ROW_SET(v) do { } while (0)
#define BUILD_T(i) do { } while (0)
#endif

template <int R2, int THREADS, bool SORT>
struct BuildLayout {
    static constexpr int NW = THREADS / WAVE;
    static constexpr int MAXC = SORT ? THREADS * 16 : 0;   // 4096 or 16384 candidates in LDS
    static constexpr int SS = 64 * R2;                     // tile slots of a bin (S = 4: 16 of the 64 are real)
    static constexpr bool CACHE_BOX = SORT && R2 <= 4;     // bin-local boxes kept in the key area once the order is final
    static constexpr int TABLES = 3 * SS + kBuildSlots * SS / 2 + (SORT ? 0 : kBuildSlots * WAVE);  // u32 words
    static constexpr int WCNT = SORT ? NW * 256 : 0;
    // [keys / boxes MAXC][ids MAXC][wcnt]; once the order is final the tables go behind the ids (over wcnt), or into
    // the key area when that is free (boxes not cached) and large enough
    static constexpr int T_OFF = (SORT && (CACHE_BOX || TABLES > MAXC)) ? 2 * MAXC : 0;
    static constexpr int WORDS = (T_OFF + TABLES > 2 * MAXC + WCNT) ? T_OFF + TABLES : 2 * MAXC + WCNT;
};

template <int R2, int THREADS, bool SORT>
__global__ __launch_bounds__(THREADS) void k_bin_build(BuildArgs a) {
    frame_stamp(a.stamps, ST_L2);
    BUILD_ROW(blockIdx.x);
    using L = BuildLayout<R2, THREADS, SORT>;
    constexpr int NW = L::NW, MAXC = L::MAXC, SS = L::SS, PER = kBuildSlots / NW;  // PER chunks per wave and round
    extern __shared__ uint32_t smem[];
    uint32_t* const s_key = smem;            // SORT; later the packed bin-local boxes (CACHE_BOX)
    uint32_t* const s_id = smem + MAXC;      // SORT
    uint32_t (*const s_wcnt)[256] = reinterpret_cast<uint32_t(*)[256]>(smem + 2 * MAXC);
    uint32_t* const t_cnt = smem + L::T_OFF;                                               // [SS] instances per tile
    uint32_t* const t_cur = t_cnt + SS;                                                    // [2][SS] list cursors (ping-pong)
    uint16_t (*const r_cnt)[SS] = reinterpret_cast<uint16_t(*)[SS]>(t_cnt + 3 * SS);      // [16][SS] per round
    uint32_t* const w_ids = t_cnt + 3 * SS + kBuildSlots * SS / 2;                         // [16][64] (!SORT)
    __shared__ uint32_t scratch[NW];
    __shared__ uint32_t s_seg;

    const int tid = threadIdx.x, lane = tid & (WAVE - 1), w = tid / WAVE;
    const uint32_t bin = ((blockIdx.x / a.g.bins_x) << a.g.grid_shift) | (blockIdx.x % a.g.bins_x);  // on-screen bins only
    BUILD_T(0);
    uint32_t c, off;
    // this bin's count and offset (k_l1_scan, the scatter's block 0); block-uniform: scalar loads
    c = a.bin_count[bin];
    off = a.bin_count[kBinOffsets + bin];
    if ((uint64_t)off + c > a.cand_capacity) c = 0;  // candidate overflow (flagged by k_l1_scatter): the frame is re-run
    if (SORT && c > (uint32_t)MAXC) {
        if (tid == 0) atomicOr(&a.counters->overflow, 2u);
        c = 0;
    }
    BUILD_T(1);
    if (SORT && c != 0) {
        constexpr int kRounds = 16;  // MAXC / THREADS
        {   // ids, then their depths: all of a thread's loads of one kind are in flight together (two round trips in all,
            // where a loop over the elements would chain two per element)
            uint32_t g[kRounds], k[kRounds];
#pragma unroll
            for (int r = 0; r < kRounds; ++r) {
                const uint32_t e = r * THREADS + tid;
                g[r] = e < c ? a.cand[off + e] : 0u;
            }
#pragma unroll
            for (int r = 0; r < kRounds; ++r) {
                const uint32_t e = r * THREADS + tid;
                k[r] = e < c ? __float_as_uint(a.depth[g[r]]) : 0u;
            }
#pragma unroll
            for (int r = 0; r < kRounds; ++r) {
                const uint32_t e = r * THREADS + tid;
                if (e < c) {
                    s_id[e] = g[r];
                    s_key[e] = k[r];
                }
            }
        }
        BUILD_T(2);
        const uint64_t lt_mask = (1ull << lane) - 1ull;
        // list element e belongs to wave e / (64 * rounds), round (e / 64) % rounds, lane e % 64
        const int rounds = (int)((c + THREADS - 1) / THREADS);  // block-uniform, <= kRounds
        const uint32_t wbase = (uint32_t)w * (uint32_t)rounds * WAVE;
#pragma unroll 1
        for (int pass = 0; pass < 4; ++pass) {
            const int shift = pass * 8;
            for (int k = tid; k < NW * 256; k += THREADS) s_wcnt[k >> 8][k & 255] = 0;
            __syncthreads();  // also orders the previous pass's (or the load's) LDS writes before this pass's reads
            uint32_t key[kRounds], id[kRounds], rank[kRounds];
#pragma unroll
            for (int r = 0; r < kRounds; ++r) {
                key[r] = 0;
                id[r] = 0;
                rank[r] = 0;
                if (r < rounds) {
                    const uint32_t e = wbase + r * WAVE + lane;
                    const bool ok = e < c;
                    if (ok) {
                        key[r] = s_key[e];
                        id[r] = s_id[e];
                    }
                    const uint32_t d = (key[r] >> shift) & 255u;
                    // lanes holding a valid key with my digit: AND over the bits of (ballot(bit) XNOR my bit)
                    const uint64_t okm = __ballot(ok);
                    uint32_t mlo = (uint32_t)okm, mhi = (uint32_t)(okm >> 32);
#pragma unroll
                    for (int bit = 0; bit < 8; ++bit) {
                        const uint32_t mine = (d >> bit) & 1u;
                        const uint64_t b = __builtin_amdgcn_ballot_w64(mine != 0);
                        const uint32_t splat = 0u - mine;
                        mlo &= ~((uint32_t)b ^ splat);
                        mhi &= ~((uint32_t)(b >> 32) ^ splat);
                    }
                    const uint64_t m = ((uint64_t)mhi << 32) | mlo;
                    uint32_t old = 0;
                    const int leader = m ? (__ffsll((unsigned long long)m) - 1) : 0;
                    if (ok && lane == leader) {
                        old = s_wcnt[w][d];
                        s_wcnt[w][d] = old + (uint32_t)__popcll(m);
                    }
                    old = __shfl(old, leader, WAVE);
                    rank[r] = old + (uint32_t)__popcll(m & lt_mask);
                }
            }
            __syncthreads();
            {   // per digit: prefix over the waves, then exclusive scan over the digits -> per-wave write cursors
                uint32_t cw[NW], cnt = 0;
                if (tid < 256) {
#pragma unroll
                    for (int k = 0; k < NW; ++k) {
                        cw[k] = s_wcnt[k][tid];
                        cnt += cw[k];
                    }
                }
                uint32_t all;
                uint32_t excl = block_excl_scan<THREADS>(cnt, scratch, &all);
                if (tid < 256) {
#pragma unroll
                    for (int k = 0; k < NW; ++k) {
                        s_wcnt[k][tid] = excl;
                        excl += cw[k];
                    }
                }
            }
            __syncthreads();
#pragma unroll
            for (int r = 0; r < kRounds; ++r) {
                if (r < rounds) {
                    const uint32_t e = wbase + r * WAVE + lane;
                    if (e < c) {
                        const uint32_t d = (key[r] >> shift) & 255u;
                        const uint32_t pos = s_wcnt[w][d] + rank[r];
                        s_key[pos] = key[r];
                        s_id[pos] = id[r];
                    }
                }
            }
            __syncthreads();  // the cursors in s_wcnt are re-zeroed at the top of the next pass
        }
        BUILD_T(3);
        if (L::CACHE_BOX) {  // the order is final: the key area now holds every candidate's bin-local tile box
            ushort4 box[kRounds];
#pragma unroll
            for (int r = 0; r < kRounds; ++r) {  // all gathers in flight together
                const uint32_t e = r * THREADS + tid;
                box[r] = e < c ? a.aabb[s_id[e]] : make_ushort4(0, 0, 0, 0);
            }
#pragma unroll
            for (int r = 0; r < kRounds; ++r) {
                const uint32_t e = r * THREADS + tid;
                if (e < c) s_key[e] = bin_local_box(a.g, bin, box[r]);
            }
        }
    }
    BUILD_T(4);
    // ---- per-tile counts (the tables live behind the ids, or in the key area when the boxes are not cached)
    for (int t = tid; t < SS; t += THREADS) t_cnt[t] = 0;
    __syncthreads();
    const uint32_t nch = (c + WAVE - 1) / WAVE;
    // chunk ch of the bin's list -> this lane's candidate and the columns of the bin's tiles
    auto chunk_columns = [&](uint32_t ch, uint64_t (&col)[R2], uint32_t& gid) {
        const uint32_t e = ch * WAVE + lane;
        uint32_t box = 0;
        gid = 0;
        if (e < c) {
            gid = SORT ? s_id[e] : a.cand[off + e];
            box = L::CACHE_BOX ? s_key[e] : bin_local_box(a.g, bin, a.aabb[gid]);
        }
        packed_cover_masks<R2>(a.g.bin_shift, box, col);
#pragma unroll
        for (int r = 0; r < R2; ++r)
            col[r] = (R2 == 1 || __builtin_amdgcn_ballot_w64(col[r] != 0) != 0) ? wave_transpose64(col[r], lane) : 0ull;
    };
    {
        uint32_t acc[R2];
#pragma unroll
        for (int r = 0; r < R2; ++r) acc[r] = 0;
        for (uint32_t ch = w; ch < nch; ch += NW) {
            uint64_t col[R2];
            uint32_t gid;
            chunk_columns(ch, col, gid);
#pragma unroll
            for (int r = 0; r < R2; ++r) acc[r] += (uint32_t)__popcll(col[r]);
        }
#pragma unroll
        for (int r = 0; r < R2; ++r)
            if (acc[r]) atomicAdd(&t_cnt[r * 64 + lane], acc[r]);
    }
    __syncthreads();
    BUILD_T(5);
    // ---- tile ranges: a segment of the list buffer for the bin (tiles consecutive inside it)
    {
        uint32_t v[(SS + THREADS - 1) / THREADS], sum = 0;
#pragma unroll
        for (int k = 0; k < (SS + THREADS - 1) / THREADS; ++k) {  // thread t owns tiles t * K .. t * K + K - 1
            const int t = tid * ((SS + THREADS - 1) / THREADS) + k;
            v[k] = t < SS ? t_cnt[t] : 0u;
            sum += v[k];
        }
        uint32_t d_bin;
        uint32_t excl = block_excl_scan<THREADS>(sum, scratch, &d_bin);
        if (tid == 0) {
            const uint32_t seg = d_bin ? atomicAdd(&a.counters->instances, d_bin) : 0u;
            s_seg = seg;
            if ((uint64_t)seg + d_bin > a.capacity) atomicOr(&a.counters->overflow, 1u);
        }
        __syncthreads();
        const uint32_t S = 1u << a.g.bin_shift;
        const uint32_t ox = (bin & ((1u << a.g.grid_shift) - 1u)) << a.g.bin_shift, oy = (bin >> a.g.grid_shift) << a.g.bin_shift;
#pragma unroll
        for (int k = 0; k < (SS + THREADS - 1) / THREADS; ++k) {
            const int t = tid * ((SS + THREADS - 1) / THREADS) + k;
            if (t < SS) {
                const uint32_t lx = (uint32_t)t & (S - 1), ly = (uint32_t)t >> a.g.bin_shift;
                const uint32_t x = ox + lx, y = oy + ly;
                // saturating: an overflowing frame is re-run, but its ranges must stay inside the list
                const uint64_t start64 = (uint64_t)s_seg + excl;
                const uint32_t start = start64 > a.capacity ? a.capacity : (uint32_t)start64;
                const uint32_t end = start64 + v[k] > a.capacity ? a.capacity : (uint32_t)(start64 + v[k]);
                if (ly < S && x < a.g.tiles_x && y < a.g.tiles_y) {
                    // absent tiles stay (0, 0) like the reference's zero-filled tileBoundaryBuffer
                    a.ranges[2 * (y * a.g.tiles_x + x)] = v[k] ? start : 0u;
                    a.ranges[2 * (y * a.g.tiles_x + x) + 1] = v[k] ? end : 0u;
                }
                t_cur[t] = start;
            }
            excl += v[k];
        }
    }
    __syncthreads();
    BUILD_T(6);
    // ---- fill, 16 chunks per round (PER per wave, consecutive) so that the lists keep the candidates' order
    const __amdgpu_buffer_rsrc_t out = __builtin_amdgcn_make_buffer_rsrc(a.sorted_gid, 0, a.capacity * 4u, 0x27000);
    int par = 0;
    for (uint32_t rb = 0; rb < nch; rb += kBuildSlots, par ^= 1) {
        uint64_t col[PER][R2];
#pragma unroll
        for (int j = 0; j < PER; ++j) {
            const int slot = w * PER + j;
            uint32_t gid;
            chunk_columns(rb + slot, col[j], gid);  // chunks past the end have empty columns
#pragma unroll
            for (int r = 0; r < R2; ++r) r_cnt[slot][r * 64 + lane] = (uint16_t)__popcll(col[j][r]);
            if (!SORT) w_ids[slot * WAVE + lane] = gid;
        }
        __syncthreads();
        for (int t = tid; t < SS; t += THREADS) {  // per tile: where each slot's run starts, and the next round's cursor
            uint32_t run = 0;
#pragma unroll
            for (int k = 0; k < kBuildSlots; ++k) {
                const uint32_t v = r_cnt[k][t];
                r_cnt[k][t] = (uint16_t)run;
                run += v;
            }
            t_cur[(par ^ 1) * SS + t] = t_cur[par * SS + t] + run;
        }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < PER; ++j) {
            const int slot = w * PER + j;
            const uint32_t ch = rb + slot;
            const uint32_t* ids = SORT ? s_id + (ch < nch ? ch : 0u) * WAVE : w_ids + slot * WAVE;
#pragma unroll
            for (int r = 0; r < R2; ++r) {
                if (__builtin_amdgcn_ballot_w64(col[j][r] != 0) == 0) continue;
                walk_column(col[j][r], t_cur[par * SS + r * 64 + lane] + r_cnt[slot][r * 64 + lane], out, ids);
            }
        }
    }
    BUILD_T(7);
}

#ifdef GS_BUILD_TIMING
extern "C" int gs_debug_build_timing(unsigned long long* out /* [1024][10] */) {
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_build_t), sizeof(unsigned long long) * 1024 * 10) == hipSuccess ? 0 : -1;
}
#endif

// ---------------------------------------------------------------------------------------
// Level 2, bin-local path, bins of 4 x 4 or 8 x 8 tiles (every resolution up to 4096 x 4096 tiles... i.e. 8K): one
// 1024-thread workgroup per bin, everything in LDS.
//   1. the bin's candidates (any order) and their depth bits -> LDS;
//   2. four stable 8-bit LSD passes on the depth bits (ballot-matched ranks, as above);
//   3. ties: candidates of equal depth must follow each other by Gaussian id (what the reference's stable sort of
//      (tile, depth) keys over index-ordered input gives).  Equal keys are adjacent now; every element of a run of
//      equal keys counts the smaller ids of its run and moves there.  Runs are short (two or three) unless the scene
//      is degenerate; a run longer than 64 sends the whole bin through id passes followed by the depth passes again;
//   4. per candidate: its tile box inside the bin (16 bits); per 64-candidate chunk and tile, how many candidates of the
//      chunk cover the tile (16 tiles: one counted ballot per tile; 64 tiles: a bit-matrix transpose); prefix over the
//      chunks per tile; tile totals -> a segment of the list buffer (one atomic add), the tile ranges;
//   5. fill: a wave takes a chunk; for each tile of the bin, a ballot of the lanes whose box covers it ranks them in
//      list order, and they store their ids at  tile start + chunk prefix + rank  -- consecutive addresses.
// ROUNDS = candidates per thread: 4, 8, 12 or 16 (4096 / 8192 / 12288 / 16384 per bin; 48 / 80 / 112 / 144 KiB of LDS).
// ---------------------------------------------------------------------------------------
// A raw-buffer descriptor over [p, p + bytes) whose four words are provably scalar: the compiler "waterfalls" a buffer access
// whose descriptor it cannot prove wave-uniform (a readfirstlane loop around the instruction with a full s_waitcnt: every
// load serialised), and values that passed through LDS or a block scan look divergent to it however uniform they are.
__device__ __forceinline__ __amdgpu_buffer_rsrc_t uniform_rsrc(const void* p, uint32_t bytes) {
    const uint64_t addr = reinterpret_cast<uint64_t>(p);
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)addr);
    const uint32_t hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(addr >> 32));
    return __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void*>(((uint64_t)hi << 32) | lo), 0,
                                             __builtin_amdgcn_readfirstlane((int)bytes), 0x27000);
}

// the 4 x 4 tiles of a bin that a bin-local box (x0 | y0 << 4 | x1 << 8 | y1 << 12, inclusive) covers: bit y * 4 + x
__device__ __forceinline__ uint32_t cover16(uint32_t pb) {
    const uint32_t lx0 = pb & 15u, ly0 = (pb >> 4) & 15u, lx1 = (pb >> 8) & 15u, ly1 = (pb >> 12) & 15u;
    const uint32_t xm = ((2u << lx1) - (1u << lx0)) & 15u;                   // columns lx0 .. lx1
    const uint32_t rows = (16u << (4u * ly1)) - (1u << (4u * ly0));          // every bit of rows ly0 .. ly1
    return (xm * 0x1111u) & rows;
}

template <int ROUNDS>
struct FastLayout {
    static constexpr int THREADS = 1024, NW = THREADS / WAVE, MAXC = THREADS * ROUNDS;
    static constexpr int WCNT_WORDS = NW * 512 / 2;                       // u16 [16][512]: digits of up to nine bits
    static constexpr int MISC_WORDS = 64 + 64 + NW * 64;                  // t_cnt, t_cur, s_seg
    static constexpr int TAIL = WCNT_WORDS > MISC_WORDS ? WCNT_WORDS : MISC_WORDS;
    static constexpr int WORDS = 2 * MAXC + TAIL;
};

// Depth slabs (MODE 1 and 2): a bin of more than MAXC candidates (up to 65535) is cut along its depth range into slabs of
// buckets holding <= MAXC candidates each.  MODE 1 (k_bin_slabs, one workgroup per bin) PLANS such a bin -- one pass over its
// records for the depth range, one for the bucket histogram, the slab bounds, one for the per-slab per-tile counts, which give
// the bin its list segment, the tile ranges and every slab its place in every tile's list -- and writes one descriptor per
// slab; bins of <= MAXC it processes itself like MODE 0.  MODE 2 (k_slab_work, workgroups striding over the descriptors)
// streams the bin's records once more, compacts the slab's members into LDS, orders them like a small bin and appends them at
// the descriptor's cursors.  The slabs are depth-ordered and each is (depth, id)-ordered inside: so is every tile's list.
// MODE 3 (k_bin_queue, round 5): both in ONE launch of persistent workgroups that take items from a queue -- first the bins, fullest
// first (a bin beyond MAXC is planned, its descriptors published with an agent-scope release; a smaller one is processed whole), then
// the slabs as they become ready.  510 bins and 297 slabs of uneven size on 256 one-per-CU workgroups left the sum of the workgroup
// times at 0.55 (bins) and 0.45 (slabs) of 256 CUs x span with a launch each (profiles/r04_level2_counts_ab.txt); the queue packs them.
// A workgroup waits for a slab only once every bin has been CLAIMED, i.e. is being worked on by a resident workgroup: no deadlock
// whatever the residency; the wait ends when the descriptor is ready or every bin is done and it is not.
template <int ROUNDS, int MODE = 0>
__device__ __forceinline__ void bin_fast_body(const BuildArgs& a) {
    constexpr bool SLABS = MODE != 0;
    using L = FastLayout<ROUNDS>;
    constexpr int THREADS = L::THREADS, NW = L::NW, MAXC = L::MAXC;
    extern __shared__ uint32_t smem[];
    // while the order is being made: the sort's payload (slot in the bin's record run | box16 << 14) and the depth bits
    uint32_t* const s_pay = smem;
    uint32_t* const s_key = smem + MAXC;
    uint16_t (*const s_wcnt)[512] = reinterpret_cast<uint16_t(*)[512]>(smem + 2 * MAXC);
    // once the order is final: the key area holds the ids, the payload area the 16-bit boxes and the chunk table, the
    // counter area the tile tables
    uint32_t* const s_id = smem + MAXC;                                                          // [MAXC]
    uint16_t* const s_box = reinterpret_cast<uint16_t*>(smem);                                   // [MAXC]
    uint16_t (*const s_tbl)[64] = reinterpret_cast<uint16_t(*)[64]>(smem + MAXC / 2);           // [MAXC / 64][64]
    uint32_t* const t_cnt = smem + 2 * MAXC;                                                     // [64]
    uint32_t* const t_cur = t_cnt + 64;                                                          // [64]
    uint32_t (*const s_seg)[64] = reinterpret_cast<uint32_t(*)[64]>(t_cnt + 128);                // [16][64]
    __shared__ uint32_t scratch[NW], scratch_hi[NW];
    __shared__ uint32_t s_seg0, s_flag;
    constexpr int kSlotBits = SLABS ? 16 : 14;  // a candidate's slot in the bin's record run: < MAXC <= 16384, or < 65536
    constexpr uint32_t kSlotMask = (1u << kSlotBits) - 1u;
    constexpr uint32_t kMaxInBin = SLABS ? 65535u : (uint32_t)MAXC;
    constexpr int kMaxSlabs = 16;
    __shared__ uint32_t g_cur[SLABS ? 64 : 1], t_tot[SLABS ? 64 : 1];  // a slab's list cursors; per-tile totals of the bin
    __shared__ uint32_t slab_first[SLABS ? kMaxSlabs + 1 : 1];         // first bucket of each slab
    __shared__ uint32_t cnt2[(MODE == 1 || MODE == 3) ? kMaxSlabs : 1][64];           // instances per slab and tile
    __shared__ uint32_t s_fill;
    __shared__ uint16_t q_order[MODE == 3 ? 1024 : 1];  // MODE 3: the on-screen bins, fullest first
    __shared__ uint32_t q_item, q_planned;              // the claimed item; how many bins are beyond MAXC (they are planned, and first)

    BUILD_ROW(MODE == 2 ? 1023u : blockIdx.x);
    if constexpr (MODE != 2) frame_stamp(a.stamps, ST_L2);  // (k_slab_work is the level's second launch)
    int tid = threadIdx.x, lane = tid & (WAVE - 1), w = tid / WAVE;  // (MODE 3 re-derives them per item: see there)
    // one workgroup per ON-SCREEN bin (the grid is bins_x * bins_y: with the padding of the bin grid in it, workgroups
    // that exit at once upset the dispatcher's placement and a few CUs end up with three of the real ones)
    auto screen_bin = [&](uint32_t i) { return ((i / a.g.bins_x) << a.g.grid_shift) | (i % a.g.bins_x); };
    uint32_t bin = screen_bin(MODE == 3 ? 0u : blockIdx.x);
    bool slab_item = MODE == 2;  // MODE 3: the item at hand is a depth slab (block-uniform)
    BUILD_T(0);
    uint32_t c_total = 0, off = 0;
    auto bin_extent = [&]() {  // this bin's count and offset (k_l1_scan and the scatter's block 0 wrote them): block-uniform, two scalar loads
        // -- telling the compiler so keeps everything derived from them (loop bounds, the record buffer's descriptor) in scalar
        // registers; a descriptor it believes divergent is "waterfalled": every load wrapped in a readfirstlane loop with a full s_waitcnt
        c_total = (uint32_t)__builtin_amdgcn_readfirstlane((int)a.bin_count[bin]);
        off = (uint32_t)__builtin_amdgcn_readfirstlane((int)a.bin_count[kBinOffsets + bin]);
        if ((uint64_t)off + c_total > a.cand_capacity) c_total = 0;  // candidate overflow (flagged by the scatter): the frame is re-run
        if (c_total > kMaxInBin) {
            if (tid == 0) atomicOr(&a.counters->overflow, 2u);
            c_total = 0;
        }
    };
    if constexpr (MODE == 0 || MODE == 1) bin_extent();
    if (tid == 0) s_flag = 0;
    BUILD_T(1);
#define multi (MODE == 2 ? true : (MODE == 1 ? c_total > (uint32_t)MAXC : (MODE == 3 ? (slab_item || c_total > (uint32_t)MAXC) : false)))  /* block-uniform */
    // c: the candidates in LDS (the whole bin, or the current slab of it); rounds / wbase follow it
    uint32_t c = multi ? 0u : c_total;
    int rounds = (int)((c + THREADS - 1) / THREADS);  // block-uniform, <= ROUNDS
    // this bin's run of 12-byte records {key, id, box16} as a raw buffer: 32-bit offsets (one address register per load instead
    // of two) and the hardware's bounds check in place of branches (reads past the run return 0)
    typedef uint32_t u32x3 __attribute__((ext_vector_type(3)));
#define recs uniform_rsrc(a.cand + (size_t)kCandWords * off, c_total * 12u)  /* rebuilt from scalars at every use: see uniform_rsrc */
    uint64_t lt_mask = (1ull << lane) - 1ull;
    // list element e belongs to wave e / (64 * rounds), round (e / 64) % rounds, lane e % 64
    uint32_t wbase = (uint32_t)w * (uint32_t)rounds * WAVE;
    // The thread's coordinates made opaque again (round 5): every phase below is a fully unrolled loop over ROUNDS elements whose
    // offsets all derive from tid; left to itself the compiler computes the offsets of ALL phases up front, keeps them alive across
    // the body and spills them (k_slab_work: 108 bytes of scratch per lane -> 0 with one refresh per slab; k_bin_queue 372 -> 140).
    auto refresh = [&]() {
        int t = threadIdx.x;
        asm volatile("" : "+v"(t));
        tid = t;
        lane = t & (WAVE - 1);
        w = t / WAVE;
        lt_mask = (1ull << lane) - 1ull;
        wbase = (uint32_t)w * (uint32_t)rounds * WAVE;
    };
    // One stable LSD pass over (s_key, s_pay) in place: digit = `db` (<= 9) bits of (s_key - sub) at `shift`; the pass
    // writes s_key - sub back (the first pass of a sort normalises the keys to the bin's smallest, the others pass sub = 0).
    auto radix_pass = [&](int shift, int db, uint32_t sub) {
        const uint32_t dmask = (1u << db) - 1u;
        for (int k = tid; k < NW * 512 / 2; k += THREADS) reinterpret_cast<uint32_t*>(&s_wcnt[0][0])[k] = 0;
        __syncthreads();  // also orders the previous pass's (or the load's) LDS writes before this pass's reads
        uint32_t key[ROUNDS], pay[ROUNDS], rank[ROUNDS];
#pragma unroll
        for (int r = 0; r < ROUNDS; ++r) {
            key[r] = 0;
            pay[r] = 0;
            rank[r] = 0;
            if (r < rounds) {
                const uint32_t e = wbase + r * WAVE + lane;
                const bool ok = e < c;
                if (ok) {
                    key[r] = s_key[e] - sub;
                    pay[r] = s_pay[e];
                }
                const uint32_t d = (key[r] >> shift) & dmask;
                // lanes holding a valid element with my digit: AND over the bits of (ballot(bit) XNOR my bit)
                const uint64_t okm = __ballot(ok);
                uint32_t mlo = (uint32_t)okm, mhi = (uint32_t)(okm >> 32);
#pragma unroll
                for (int bit = 0; bit < 9; ++bit) {
                    if (bit < db) {  // wave-uniform
                        const uint32_t mine = (d >> bit) & 1u;
                        const uint64_t b = __builtin_amdgcn_ballot_w64(mine != 0);
                        const uint32_t splat = 0u - mine;
                        mlo &= ~((uint32_t)b ^ splat);
                        mhi &= ~((uint32_t)(b >> 32) ^ splat);
                    }
                }
                const uint64_t m = ((uint64_t)mhi << 32) | mlo;
                uint32_t old = 0;
                const int leader = m ? (__ffsll((unsigned long long)m) - 1) : 0;
                if (ok && lane == leader) {
                    old = s_wcnt[w][d];
                    s_wcnt[w][d] = (uint16_t)(old + (uint32_t)__popcll(m));
                }
                old = __shfl(old, leader, WAVE);
                rank[r] = old + (uint32_t)__popcll(m & lt_mask);
            }
        }
        __syncthreads();
        {   // per digit: prefix over the waves, then exclusive scan over the digits -> per-wave write cursors
            uint32_t cw[NW], cnt = 0;
            if (tid < 512) {
#pragma unroll
                for (int k = 0; k < NW; ++k) {
                    cw[k] = s_wcnt[k][tid];
                    cnt += cw[k];
                }
            }
            uint32_t all;
            uint32_t excl = block_excl_scan<THREADS>(cnt, scratch, &all);
            if (tid < 512) {
#pragma unroll
                for (int k = 0; k < NW; ++k) {
                    s_wcnt[k][tid] = (uint16_t)excl;
                    excl += cw[k];
                }
            }
        }
        __syncthreads();
#pragma unroll
        for (int r = 0; r < ROUNDS; ++r) {
            if (r < rounds) {
                const uint32_t e = wbase + r * WAVE + lane;
                if (e < c) {
                    const uint32_t d = (key[r] >> shift) & dmask;
                    const uint32_t pos = (uint32_t)s_wcnt[w][d] + rank[r];
                    s_key[pos] = key[r];
                    s_pay[pos] = pay[r];
                }
            }
        }
        __syncthreads();
    };
    // smallest key and the span of the keys in LDS (block-uniform, scalar)
    auto block_minmax = [&](uint32_t lo, uint32_t hi, uint32_t& kmin, uint32_t& span) {
#pragma unroll
        for (int d = 1; d < WAVE; d <<= 1) {
            lo = min(lo, (uint32_t)__shfl_xor((int)lo, d, WAVE));
            hi = max(hi, (uint32_t)__shfl_xor((int)hi, d, WAVE));
        }
        __syncthreads();  // scratch reuse; also: the LDS writes of the load are visible
        if (lane == 0) {
            scratch[w] = lo;
            scratch_hi[w] = hi;
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < NW; ++k) {
            lo = min(lo, scratch[k]);
            hi = max(hi, scratch_hi[k]);
        }
        kmin = (uint32_t)__builtin_amdgcn_readfirstlane((int)lo);
        span = (uint32_t)__builtin_amdgcn_readfirstlane((int)hi) - kmin;
    };
    auto key_range = [&](uint32_t& kmin, uint32_t& span) {
        uint32_t lo = 0xFFFFFFFFu, hi = 0u;
#pragma unroll
        for (int r = 0; r < ROUNDS; ++r) {
            const uint32_t e = r * THREADS + tid;
            if (r < rounds && e < c) {
                const uint32_t k = s_key[e];
                lo = min(lo, k);
                hi = max(hi, k);
            }
        }
        block_minmax(lo, hi, kmin, span);
    };
    // Order (s_key, s_pay) by s_key with stable LSD passes: the keys are normalised to the smallest of them, which leaves
    // `bits` significant bits (25 or so for the depths of one bin of a frame: three passes of nine bits)
    auto sort_by_key_lsd = [&](uint32_t kmin, uint32_t span) {
        const int bits = span ? 32 - __builtin_clz(span) : 0;
        const int npass = (bits + 8) / 9;
        const int db = npass ? (bits + npass - 1) / npass : 0;
#pragma unroll 1
        for (int pass = 0; pass < npass; ++pass) radix_pass(pass * db, db, pass == 0 ? kmin : 0u);
        return npass ? kmin : 0u;  // what the stored keys are short of the originals
    };
    // The usual case, in two steps instead of three or four passes: (1) the keys' top twelve significant bits cut the bin into
    // 4096 buckets (a counting scatter with LDS atomics: the order inside a bucket does not matter yet), (2) every element
    // counts the elements of its bucket that precede it -- smaller key, or equal key and earlier position -- and moves to
    // bucket start + that count.  Buckets hold two or three elements when the depths are spread over the bin's range; when
    // they are not (a bucket of more than kMsdBucketMax: the candidates of a wall seen face on) the stable passes take over.
    // Equal keys end up adjacent in an arbitrary order either way: the tie step below puts them in id order.
    constexpr uint32_t kMsdBuckets = 4096, kMsdBucketMax = 64;
    uint32_t* const m_cnt = smem + 2 * MAXC;                                                  // [4096] u16, two to a word
    uint16_t* const m_start = reinterpret_cast<uint16_t*>(smem + 2 * MAXC + kMsdBuckets / 2);  // [4096] u16 bucket starts
    static_assert(kMsdBuckets * 4 <= L::TAIL * 4, "the bucket tables live in the counters' area");
    auto sort_by_key_msd = [&](uint32_t kmin, uint32_t span) -> bool {
        const int bits = 32 - __builtin_clz(span);  // span != 0
        const int sh = bits > 12 ? bits - 12 : 0;
        for (uint32_t k = tid; k < kMsdBuckets / 2; k += THREADS) m_cnt[k] = 0;
        __syncthreads();
        uint32_t key[ROUNDS], pay[ROUNDS];
#pragma unroll
        for (int r = 0; r < ROUNDS; ++r) {
            const uint32_t e = r * THREADS + tid;
            key[r] = 0;
            pay[r] = 0;
            if (r < rounds && e < c) {
                key[r] = s_key[e];
                pay[r] = s_pay[e];
                const uint32_t b = (key[r] - kmin) >> sh;
                atomicAdd(&m_cnt[b >> 1], 1u << (16u * (b & 1u)));  // a bucket holds <= 16384: no carry between the halves
            }
        }
        __syncthreads();
        {   // thread t owns buckets 4 t .. 4 t + 3: their starts, and the counters back to zero (they become cursors)
            const uint2 w2 = reinterpret_cast<const uint2*>(m_cnt)[tid];
            const uint32_t v0 = w2.x & 0xFFFFu, v1 = w2.x >> 16, v2 = w2.y & 0xFFFFu, v3 = w2.y >> 16;
            uint32_t all;
            uint32_t excl = block_excl_scan<THREADS>(v0 + v1 + v2 + v3, scratch, &all);
            if (max(max(v0, v1), max(v2, v3)) > kMsdBucketMax) s_flag = 2;  // (every writer stores the same value)
            reinterpret_cast<uint2*>(m_cnt)[tid] = make_uint2(0u, 0u);
            ushort4 st;
            st.x = (unsigned short)excl;
            st.y = (unsigned short)(excl + v0);
            st.z = (unsigned short)(excl + v0 + v1);
            st.w = (unsigned short)(excl + v0 + v1 + v2);
            reinterpret_cast<ushort4*>(m_start)[tid] = st;
        }
        __syncthreads();
        if (__builtin_amdgcn_readfirstlane((int)s_flag) == 2) {  // a crowded bucket: nothing has moved yet
            __syncthreads();
            if (tid == 0) s_flag = 0;
            return false;
        }
#pragma unroll
        for (int r = 0; r < ROUNDS; ++r) {
            const uint32_t e = r * THREADS + tid;
            if (r < rounds && e < c) {
                const uint32_t b = (key[r] - kmin) >> sh;
                const uint32_t old = atomicAdd(&m_cnt[b >> 1], 1u << (16u * (b & 1u)));
                const uint32_t pos = (uint32_t)m_start[b] + ((old >> (16u * (b & 1u))) & 0xFFFFu);
                s_key[pos] = key[r];
                s_pay[pos] = pay[r];
            }
        }
        __syncthreads();
        uint32_t dst[ROUNDS];
#pragma unroll
        for (int r = 0; r < ROUNDS; ++r) {
            const uint32_t e = r * THREADS + tid;
            dst[r] = 0xFFFFFFFFu;
            if (r < rounds && e < c) {
                const uint32_t k = s_key[e];
                pay[r] = s_pay[e];
                key[r] = k;
                const uint32_t b = (k - kmin) >> sh;
                const uint32_t j0 = m_start[b], j1 = b + 1 < kMsdBuckets ? (uint32_t)m_start[b + 1] : c;
                uint32_t before = 0;
                for (uint32_t j = j0; j < j1; ++j) {
                    const uint32_t kj = s_key[j];
                    before += (kj < k || (kj == k && j < e)) ? 1u : 0u;
                }
                dst[r] = j0 + before;
            }
        }
        __syncthreads();  // every read of the bucketed order is done
#pragma unroll
        for (int r = 0; r < ROUNDS; ++r)
            if (dst[r] != 0xFFFFFFFFu) {
                s_key[dst[r]] = key[r];
                s_pay[dst[r]] = pay[r];
            }
        __syncthreads();
        return true;
    };
    // returns what the stored keys are short of the originals (the LSD passes normalise them; equality is what matters later)
    auto sort_by_key = [&](bool try_msd) -> uint32_t {
        uint32_t kmin, span;
        key_range(kmin, span);
        if (span == 0) return 0u;  // all keys equal: nothing to order
        if constexpr (ROUNDS <= 12) {  // (sixteen elements per thread do not leave the registers for the second step)
            if (try_msd && sort_by_key_msd(kmin, span)) return 0u;
        }
        return sort_by_key_lsd(kmin, span);
    };
    const int S = 1 << a.g.bin_shift;
    // the bin's list segment and its tiles' ranges, from the per-tile instance counts (lane t < 64: tile t of the bin)
    auto place_lists = [&](uint32_t v, uint32_t* cursors) {
        uint32_t d_bin;
        const uint32_t excl = block_excl_scan<THREADS>(v, scratch, &d_bin);
        if (tid == 0) {
            const uint32_t seg = d_bin ? atomicAdd(&a.counters->instances, d_bin) : 0u;
            s_seg0 = seg;
            if ((uint64_t)seg + d_bin > a.capacity) atomicOr(&a.counters->overflow, 1u);
        }
        __syncthreads();
        if (tid < 64) {
            const uint32_t ox = (bin & ((1u << a.g.grid_shift) - 1u)) << a.g.bin_shift, oy = (bin >> a.g.grid_shift) << a.g.bin_shift;
            const uint32_t lx = (uint32_t)tid & (uint32_t)(S - 1), ly = (uint32_t)tid >> a.g.bin_shift;
            const uint32_t x = ox + lx, y = oy + ly;
            // saturating: an overflowing frame is re-run, but its ranges must stay inside the list
            const uint64_t start64 = (uint64_t)s_seg0 + excl;
            const uint32_t start = start64 > a.capacity ? a.capacity : (uint32_t)start64;
            const uint32_t end = start64 + v > a.capacity ? a.capacity : (uint32_t)(start64 + v);
            if (ly < (uint32_t)S && x < a.g.tiles_x && y < a.g.tiles_y) {
                // absent tiles stay (0, 0) like the reference's zero-filled tileBoundaryBuffer
                a.ranges[2 * (y * a.g.tiles_x + x)] = v ? start : 0u;
                a.ranges[2 * (y * a.g.tiles_x + x) + 1] = v ? end : 0u;
            }
            cursors[tid] = start;
        }
    };
    // ---- a bin beyond MAXC: its depth range, the bucket histogram, the per-tile totals, the slabs
    uint32_t n_slabs = 1, g_kmin = 0;
    int g_sh = 0;
    auto plan_bin = [&]() {
    if constexpr (MODE == 1 || MODE == 3) if (!slab_item && multi) {
        uint32_t lo = 0xFFFFFFFFu, hi = 0u;
        for (uint32_t base = 0; base < c_total; base += 8 * THREADS) {  // pass 0: the keys' range (eight loads in flight)
            uint32_t k[8];
#pragma unroll
            for (int r = 0; r < 8; ++r) k[r] = __builtin_amdgcn_raw_buffer_load_b32(recs, (base + r * THREADS + tid) * 12u, 0, 0);
#pragma unroll
            for (int r = 0; r < 8; ++r)
                if (base + r * THREADS + tid < c_total) {
                    lo = min(lo, k[r]);
                    hi = max(hi, k[r]);
                }
        }
        uint32_t span;
        block_minmax(lo, hi, g_kmin, span);
        const int bits = span ? 32 - __builtin_clz(span) : 0;
        g_sh = bits > 12 ? bits - 12 : 0;
        for (uint32_t k = tid; k < kMsdBuckets / 2; k += THREADS) m_cnt[k] = 0;
        for (uint32_t k = tid; k < (uint32_t)kMaxSlabs * 64u; k += THREADS) cnt2[k >> 6][k & 63u] = 0;
        __syncthreads();
        for (uint32_t base = 0; base < c_total; base += 8 * THREADS) {  // pass 1: the bucket histogram
            uint32_t k[8];
#pragma unroll
            for (int r = 0; r < 8; ++r) k[r] = __builtin_amdgcn_raw_buffer_load_b32(recs, (base + r * THREADS + tid) * 12u, 0, 0);
#pragma unroll
            for (int r = 0; r < 8; ++r)
                if (base + r * THREADS + tid < c_total) {
                    const uint32_t bk = (k[r] - g_kmin) >> g_sh;
                    atomicAdd(&m_cnt[bk >> 1], 1u << (16u * (bk & 1u)));  // <= 65535 per bucket: no carry
                }
        }
        __syncthreads();
        {   // exclusive prefix over the buckets (thread t: buckets 4 t .. 4 t + 3), kept as u16 (the bin holds < 65536)
            const uint2 w2 = reinterpret_cast<const uint2*>(m_cnt)[tid];
            const uint32_t v0 = w2.x & 0xFFFFu, v1 = w2.x >> 16, v2 = w2.y & 0xFFFFu, v3 = w2.y >> 16;
            uint32_t all;
            const uint32_t excl = block_excl_scan<THREADS>(v0 + v1 + v2 + v3, scratch, &all);
            if (max(max(v0, v1), max(v2, v3)) > (uint32_t)MAXC) s_flag = 3;  // one bucket beyond a slab: the global path
            ushort4 st;
            st.x = (unsigned short)excl;
            st.y = (unsigned short)(excl + v0);
            st.z = (unsigned short)(excl + v0 + v1);
            st.w = (unsigned short)(excl + v0 + v1 + v2);
            reinterpret_cast<ushort4*>(m_start)[tid] = st;
            if (tid == 0) {
                slab_first[0] = 0;
                s_fill = 1;  // slabs so far
            }
        }
        __syncthreads();
        // slab k + 1 starts at the first bucket that no longer fits behind slab k's first (every thread looks at its four)
        for (int k = 0; k < kMaxSlabs; ++k) {
            const uint32_t base = m_start[slab_first[k]];
            if (c_total - base <= (uint32_t)MAXC) break;  // block-uniform: the rest fits
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const uint32_t bk = 4u * tid + j;
                const uint32_t p0 = m_start[bk], p1 = bk + 1 < kMsdBuckets ? (uint32_t)m_start[bk + 1] : c_total;
                if (p0 - base <= (uint32_t)MAXC && p1 - base > (uint32_t)MAXC && bk > slab_first[k]) {
                    slab_first[k + 1] = bk;
                    s_fill = k + 2;
                }
            }
            __syncthreads();
            if (s_fill != (uint32_t)k + 2) {  // (cannot happen while no bucket exceeds MAXC; keeps the loop finite)
                if (tid == 0) s_flag = 3;
                break;
            }
        }
        __syncthreads();
        n_slabs = (uint32_t)__builtin_amdgcn_readfirstlane((int)s_fill);
        if (tid == 0) slab_first[n_slabs] = kMsdBuckets;
        const uint32_t remaining = c_total - m_start[slab_first[n_slabs - 1]];
        if (__builtin_amdgcn_readfirstlane((int)s_flag) == 3 || remaining > (uint32_t)MAXC) {
            if (tid == 0) atomicOr(&a.counters->overflow, 2u);
            n_slabs = 0;
        }
        __syncthreads();
        if (tid == 0) s_flag = 0;
        // pass 2: instances per slab and tile
        for (uint32_t base = 0; n_slabs != 0 && base < c_total; base += 8 * THREADS) {
            uint32_t k[8], bx[8];
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                const u32x3 rec = __builtin_amdgcn_raw_buffer_load_b96(recs, (base + r * THREADS + tid) * 12u, 0, 0);
                k[r] = rec.x;
                bx[r] = rec.z;
            }
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                const bool valid = base + r * THREADS + tid < c_total;
                const uint32_t bk = (k[r] - g_kmin) >> g_sh;
                uint32_t sl = 0;
                for (uint32_t j = 1; j < n_slabs; ++j) sl += bk >= slab_first[j] ? 1u : 0u;
                const uint32_t pb = bx[r];
                if (S == 4) {
                    // 16 tiles: per tile the ballot of the covering lanes, per slab the lanes that belong to it -- lane t adds
                    // the count of (slab, tile t) with ONE atomic per wave, slab and 64 records (an atomic per record and
                    // tile put all 1024 threads on the same 16 n_slabs counters)
                    const uint32_t cm = valid ? cover16(pb) : 0u;
                    if (__builtin_amdgcn_ballot_w64(cm != 0) == 0) continue;
                    uint64_t bal[16];
#pragma unroll
                    for (int t = 0; t < 16; ++t) bal[t] = __builtin_amdgcn_ballot_w64(((cm >> t) & 1u) != 0);
                    for (uint32_t j = 0; j < n_slabs; ++j) {
                        const uint64_t mk = __builtin_amdgcn_ballot_w64(valid && sl == j);
                        if (mk == 0) continue;
                        uint32_t mine = 0;
#pragma unroll
                        for (int t = 0; t < 16; ++t) mine = lane == t ? (uint32_t)__popcll(bal[t] & mk) : mine;
                        if (lane < 16 && mine != 0) atomicAdd(&cnt2[j][lane], mine);
                    }
                } else if (valid) {
                    const uint32_t lx0 = pb & 15u, ly0 = (pb >> 4) & 15u, lx1 = (pb >> 8) & 15u, ly1 = (pb >> 12) & 15u;
                    for (uint32_t y = ly0; y <= ly1; ++y)
                        for (uint32_t x = lx0; x <= lx1; ++x) atomicAdd(&cnt2[sl][(y << a.g.bin_shift) + x], 1u);
                }
            }
        }
        __syncthreads();
        if (tid < 64) {  // per tile: the bin's total; cnt2 becomes the exclusive prefix over the slabs
            uint32_t run = 0;
            for (uint32_t k = 0; k < n_slabs; ++k) {
                const uint32_t v = cnt2[k][tid];
                cnt2[k][tid] = run;
                run += v;
            }
            t_tot[tid] = run;
        }
        __syncthreads();
        place_lists(tid < 64 ? t_tot[tid] : 0u, g_cur);
        __syncthreads();
        // one descriptor per slab; k_slab_work does the rest
        if (tid == 0) {
            const uint32_t base = n_slabs ? atomicAdd(&a.counters->slabs, n_slabs) : 0u;
            s_seg0 = base;
            if (base + n_slabs > a.slab_capacity) atomicOr(&a.counters->overflow, 2u);  // -> the global path
        }
        __syncthreads();
        const uint32_t dbase = (uint32_t)__builtin_amdgcn_readfirstlane((int)s_seg0);
        if (dbase + n_slabs <= a.slab_capacity) {
            // MODE 3 hands the descriptors to workgroups of the SAME launch, possibly on another XCD (whose L2 is not coherent with this
            // one): every word goes out as a relaxed agent-scope atomic store (written through, sc1) and is read the same way on the
            // other side; then every wave's stores retired -> barrier -> the ready words, the same way.  No L2 write-back, no fence:
            // an agent-scope release here (buffer_wbl2) flushes the whole XCD's dirty lines -- the other frames' records, lists and
            // pixels included -- once per planned bin, and cost 6 % of the frame rate with three frames in flight.
            auto put = [&](uint32_t* p, uint32_t v) {
                if constexpr (MODE == 3) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                else *p = v;
            };
            for (uint32_t k = 0; k < n_slabs; ++k) {
                SlabDesc* d = a.slabs + dbase + k;
                if (tid < 64) put(&d->cur[tid], g_cur[tid] + cnt2[k][tid]);
                if (tid == 64) {
                    put(&d->off, off);
                    put(&d->c_total, c_total);
                    put(&d->kmin, g_kmin);
                    put(reinterpret_cast<uint32_t*>(&d->sh), (uint32_t)g_sh);
                    put(&d->b_lo, slab_first[k]);
                    put(&d->b_hi, slab_first[k + 1]);
                }
            }
            if constexpr (MODE == 3) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __syncthreads();
                if (tid == 0) {
                    for (uint32_t k = 0; k < n_slabs; ++k)
                        __hip_atomic_store(&a.slabs[dbase + k].ready, a.epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // ... and they have left before this bin counts as planned
                }
            }
        }
        n_slabs = 0;  // nothing more to do for this bin here
    }
    };
    if constexpr (MODE == 1) plan_bin();
    // (a lambda, not a loop body: with a loop around it -- even one of a constant single trip -- the register allocator
    // of hipcc 7.2 spills 45 instead of 19 registers in k_bin_fast<12>)
    auto slab_body = [&](const uint32_t slab) {
    if constexpr (SLABS) if (multi) {  // ---- this slab's members, compacted into LDS (any order: they are ordered next)
        const uint32_t b_lo = slab_first[slab], b_hi = slab_first[slab + 1];
        __syncthreads();  // the previous slab is done with LDS
        if (tid == 0) s_fill = 0;
        __syncthreads();
        for (uint32_t base = 0; base < c_total; base += 8 * THREADS) {
            uint32_t k[8], bx[8];
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                const u32x3 rec = __builtin_amdgcn_raw_buffer_load_b96(recs, (base + r * THREADS + tid) * 12u, 0, 0);
                k[r] = rec.x;
                bx[r] = rec.z;
            }
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                const uint32_t e = base + r * THREADS + tid;
                const uint32_t bk = (k[r] - g_kmin) >> g_sh;
                const bool mine = e < c_total && bk >= b_lo && bk < b_hi;
                const uint64_t mm = __builtin_amdgcn_ballot_w64(mine);  // one atomic per wave and round
                if (mm != 0) {
                    uint32_t at = 0;
                    if (lane == __ffsll((unsigned long long)mm) - 1) at = atomicAdd(&s_fill, (uint32_t)__popcll(mm));
                    at = (uint32_t)__builtin_amdgcn_readlane((int)at, __ffsll((unsigned long long)mm) - 1) + (uint32_t)__popcll(mm & lt_mask);
                    if (mine && at < (uint32_t)MAXC) {
                        s_key[at] = k[r];
                        s_pay[at] = e | (bx[r] << kSlotBits);
                    }
                }
            }
        }
        __syncthreads();
        c = min((uint32_t)__builtin_amdgcn_readfirstlane((int)s_fill), (uint32_t)MAXC);
        rounds = (int)((c + THREADS - 1) / THREADS);
        wbase = (uint32_t)w * (uint32_t)rounds * WAVE;
    }
    // attempt 0: the records as level 1 left them (any order inside a block's run).  attempt 1 (only after a run of more
    // than 64 equal depths, i.e. a degenerate scene): the records rewritten in id order, so that the stable passes alone
    // leave equal depths in id order
    for (int attempt = 0; c != 0 && attempt < 2; ++attempt) {
        // ---- the bin's records, streamed: eight loads per thread in flight together (sixteen would not leave the registers
        // for it: a spilled address register is reloaded through the same counter the loads use, which serialises them)
        uint32_t tid12 = (uint32_t)tid * 12u;
        asm volatile("" : "+v"(tid12));  // opaque: or the sixteen offsets are hoisted out of the attempt loop, kept, and spilled
        constexpr int HR = ROUNDS <= 8 ? ROUNDS : ROUNDS / 2;  // 4, 8, 6, 8
        const int h_end = multi ? 0 : ROUNDS;  // (a slab's members are in LDS already)
#pragma unroll
        for (int h = 0; h < h_end; h += HR) {
            uint32_t k[HR], b[HR];
#pragma unroll
            for (int r = 0; r < HR; ++r) {
                k[r] = 0;
                b[r] = 0;
                if (h + r < rounds) {
                    const u32x3 rec = __builtin_amdgcn_raw_buffer_load_b96(recs, tid12 + (uint32_t)(h + r) * (THREADS * 12u), 0, 0);
                    k[r] = rec.x;
                    b[r] = rec.z;
                }
            }
#pragma unroll
            for (int r = 0; r < HR; ++r) {
                const uint32_t e = (h + r) * THREADS + tid;
                if (h + r < rounds && e < c) {
                    s_key[e] = k[r];
                    s_pay[e] = e | (b[r] << kSlotBits);
                }
            }
        }
        if (attempt == 0) BUILD_T(2);
        sort_by_key(attempt == 0);
        if (attempt == 0) BUILD_T(8);
        refresh();
        // ---- ties.  Candidates of equal depth must follow each other by Gaussian id (what the reference's stable sort of
        // (tile, depth) keys over index-ordered input gives).  Equal keys are adjacent now.  The element that STARTS a run of
        // equal keys measures the run (at most 64 more); all elements fetch their ids (a gather inside the bin's own record
        // run, which this workgroup has just streamed) and the ids replace the keys; then each run's first element alone
        // puts its run into id order, in place (an insertion sort over (id, payload): runs are two or three long unless the
        // scene is degenerate, and no two runs share a position, so there is nothing to synchronise).
        uint32_t my_id[ROUNDS], run_len[(ROUNDS + 3) / 4];  // run lengths - 1, eight bits each
        bool too_long = false;
#pragma unroll
        for (int q = 0; q < (ROUNDS + 3) / 4; ++q) run_len[q] = 0;
#pragma unroll
        for (int r = 0; r < ROUNDS; ++r) {
            const uint32_t e = r * THREADS + tid;
            my_id[r] = 0;
            if (r < rounds && e < c) {
                my_id[r] = __builtin_amdgcn_raw_buffer_load_b32(recs, (s_pay[e] & kSlotMask) * 12u + 4u, 0, 0);
                const uint32_t k = s_key[e];
                if (attempt == 0 && (e == 0 || s_key[e - 1] != k) && e + 1 < c && s_key[e + 1] == k) {
                    uint32_t more = 1;
                    while (e + more + 1 < c && more < 64 && s_key[e + more + 1] == k) ++more;
                    if (e + more + 1 < c && s_key[e + more + 1] == k) too_long = true;
                    run_len[r / 4] |= more << (8 * (r % 4));
                }
            }
        }
        if (too_long) s_flag = 1;
        __syncthreads();  // every read of the keys is done: the ids take their place
        const bool redo = __builtin_amdgcn_readfirstlane((int)s_flag) != 0;  // block-uniform (an LDS read is not, to the compiler)
#pragma unroll
        for (int r = 0; r < ROUNDS; ++r) {
            const uint32_t e = r * THREADS + tid;
            if (r < rounds && e < c) s_id[e] = my_id[r];
        }
        __syncthreads();
        if (attempt == 0) BUILD_T(9);
        if (multi && redo) {  // a run of > 64 equal depths inside a slab: not handled here -- the global path takes the frame
            if (tid == 0) atomicOr(&a.counters->overflow, 2u);
            __syncthreads();
            if (tid == 0) s_flag = 0;
            break;
        }
        if (!redo) {
#pragma unroll
            for (int r = 0; r < ROUNDS; ++r) {
                const uint32_t more = (run_len[r / 4] >> (8 * (r % 4))) & 255u;
                if (more != 0) {
                    const uint32_t e = r * THREADS + tid;
                    for (uint32_t i = e + 1; i <= e + more; ++i) {  // insertion sort of [e, e + more] by id
                        const uint32_t vi = s_id[i], vp = s_pay[i];
                        uint32_t j = i;
                        while (j > e && s_id[j - 1] > vi) {
                            s_id[j] = s_id[j - 1];
                            s_pay[j] = s_pay[j - 1];
                            --j;
                        }
                        s_id[j] = vi;
                        s_pay[j] = vp;
                    }
                }
            }
            __syncthreads();
            break;
        }
        // ---- a long run of equal depths: order the bin by id (the ids sit in the key area: four more passes), rewrite its
        // records in that order and start over; the second attempt needs no tie handling
        const uint32_t id_base = sort_by_key(false);
        uint32_t nk[ROUNDS], ni[ROUNDS], nb[ROUNDS];
#pragma unroll
        for (int r = 0; r < ROUNDS; ++r) {
            const uint32_t e = r * THREADS + tid;
            nk[r] = ni[r] = nb[r] = 0;
            if (r < rounds && e < c) {
                const uint32_t pw = s_pay[e];
                nk[r] = __builtin_amdgcn_raw_buffer_load_b32(recs, (pw & kSlotMask) * 12u, 0, 0);
                ni[r] = s_id[e] + id_base;
                nb[r] = pw >> kSlotBits;
            }
        }
        __syncthreads();  // (workgroup-scope fence included) every record has been read before any is overwritten
#pragma unroll
        for (int r = 0; r < ROUNDS; ++r) {
            const uint32_t e = r * THREADS + tid;
            if (r < rounds && e < c) {
                u32x3 rec = {nk[r], ni[r], nb[r]};
                __builtin_amdgcn_raw_buffer_store_b96(rec, recs, e * 12u, 0, 0);
            }
        }
        __threadfence_block();
        if (tid == 0) s_flag = 0;
        __syncthreads();
    }
#undef recs
    BUILD_T(3);
    refresh();
    // ---- the candidates' tile boxes inside the bin (they rode along in the payload), and the (chunk, tile) counts
    const uint32_t nch = (c + WAVE - 1) / WAVE;
    {
        uint32_t pb16[ROUNDS];
#pragma unroll
        for (int r = 0; r < ROUNDS; ++r) {
            const uint32_t e = r * THREADS + tid;
            pb16[r] = e < c ? s_pay[e] >> kSlotBits : 0u;
        }
        __syncthreads();  // the payloads are dead from here on: their area becomes boxes + chunk table
        for (uint32_t k = tid; k < (uint32_t)MAXC / 2; k += THREADS) smem[MAXC / 2 + k] = 0;  // the table
        if (tid < 64) t_cnt[tid] = 0;
        __syncthreads();
#pragma unroll
        for (int r = 0; r < ROUNDS; ++r) {
            const uint32_t e = r * THREADS + tid;
            const uint32_t pb = pb16[r];
            if (e < c) s_box[e] = (uint16_t)pb;  // x0 | y0 << 4 | x1 << 8 | y1 << 12, inclusive upper bounds
            if (S == 4) {
                // 16 tiles: this wave holds chunk r * NW + w; the ballot of the lanes whose box covers tile t, counted, is the
                // chunk's entry for tile t (LDS atomics on the 16 counters of a chunk -- 64 lanes, four tiles each, on eight
                // words -- took 19 us of a 12288-candidate slab's 82: profiles/r04_slab_phases_T.txt)
                const uint32_t ch = (uint32_t)r * NW + (uint32_t)__builtin_amdgcn_readfirstlane(w);
                if (ch < nch) {
                    const uint32_t cm = e < c ? cover16(pb) : 0u;
                    uint32_t mine = 0;
#pragma unroll
                    for (int t = 0; t < 16; ++t) {
                        const uint64_t bal = __builtin_amdgcn_ballot_w64(((cm >> t) & 1u) != 0);
                        mine = lane == t ? (uint32_t)__popcll(bal) : mine;
                    }
                    s_tbl[ch][lane] = (uint16_t)mine;
                }
            }
        }
        if (S != 4) {  // 64 tiles: one bit-matrix transpose per chunk hands lane t the column of tile t
            __syncthreads();
            for (uint32_t ch = w; ch < nch; ch += NW) {
                const uint32_t e = ch * WAVE + lane;
                const uint32_t pb = e < c ? (uint32_t)s_box[e] : 0xFFFFu;  // 0xFFFF: x0 = 15 > x1: covers nothing
                uint64_t m[1];
                cover_masks<1>(3, (int)(pb & 15u), (int)((pb >> 4) & 15u), (int)((pb >> 8) & 15u) + 1, (int)(pb >> 12) + 1, m);
                if (e >= c) m[0] = 0;
                s_tbl[ch][lane] = (uint16_t)__popcll(wave_transpose64(m[0], lane));
            }
        }
    }
    __syncthreads();
    BUILD_T(4);
    // ---- per tile: exclusive prefix of the chunk counts (16 segments of chunks per tile), tile total
    {
        const int t = tid & 63, g = tid >> 6;
        const uint32_t per = (nch + NW - 1) / NW;
        const uint32_t q0 = min(nch, (uint32_t)g * per), q1 = min(nch, q0 + per);
        uint32_t sum = 0;
        for (uint32_t q = q0; q < q1; ++q) sum += s_tbl[q][t];
        s_seg[g][t] = sum;
        __syncthreads();
        if (tid < 64) {
            uint32_t run = 0;
#pragma unroll
            for (int k = 0; k < NW; ++k) {
                const uint32_t v = s_seg[k][tid];
                s_seg[k][tid] = run;
                run += v;
            }
            t_cnt[tid] = run;
        }
        __syncthreads();
        uint32_t run = s_seg[g][t];
        for (uint32_t q = q0; q < q1; ++q) {
            const uint32_t v = s_tbl[q][t];
            s_tbl[q][t] = (uint16_t)run;  // a tile's list in a bin is at most MAXC long: 16 bits
            run += v;
        }
    }
    __syncthreads();
    BUILD_T(5);
    // ---- tile ranges: a segment of the list buffer for the bin (tiles consecutive inside it) -- or, for a slab of a bin
    // that was placed up front, where the slabs before this one left each tile's list
    if (!multi) {
        place_lists(tid < 64 ? t_cnt[tid] : 0u, t_cur);
    } else if (tid < 64) {
        t_cur[tid] = g_cur[tid];
    }
    __syncthreads();
    BUILD_T(6);
    refresh();
    // ---- fill.  Chunks are independent now: the table holds where each chunk's run starts in every tile's list.
    const __amdgpu_buffer_rsrc_t out = __builtin_amdgcn_make_buffer_rsrc(a.sorted_gid, 0, a.capacity * 4u, 0x27000);
    for (uint32_t ch = w; ch < nch; ch += NW) {
        const uint32_t e = ch * WAVE + lane;
        const bool valid = e < c;
        const uint32_t id = valid ? s_id[e] : 0u;
        const uint32_t pb = valid ? (uint32_t)s_box[e] : 0u;
        const int lx0 = (int)(pb & 15u), ly0 = (int)((pb >> 4) & 15u), lx1 = (int)((pb >> 8) & 15u), ly1 = (int)(pb >> 12);
        // lane t: where this chunk's run starts in tile t's list
        const uint32_t base = t_cur[lane] + (uint32_t)s_tbl[ch][lane];
        if (S == 4) {
            // 16 tiles: for each, the ballot of the covering lanes ranks them in list order and they store their ids at
            // consecutive addresses
            const uint32_t cm = valid ? cover16(pb) : 0u;
#pragma unroll
            for (int t = 0; t < 16; ++t) {
                const bool covered = ((cm >> t) & 1u) != 0;
                const uint64_t bal = __builtin_amdgcn_ballot_w64(covered);
                if (bal == 0) continue;
                const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(bal >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)bal, 0u));
                const uint32_t pos = (uint32_t)__builtin_amdgcn_readlane((int)base, t) + rank;
                __builtin_amdgcn_raw_buffer_store_b32(id, out, covered ? pos * 4u : 0xFFFFFFFFu, 0, 0);
            }
        } else {
            // 64 tiles: transpose, then every lane walks its tile's column (up to four ids per store)
            uint64_t m[1];
            cover_masks<1>(3, lx0, ly0, lx1 + 1, ly1 + 1, m);
            if (!valid) m[0] = 0;
            walk_column(wave_transpose64(m[0], lane), base, out, s_id + ch * WAVE);
        }
    }
    if (multi && tid < 64) g_cur[tid] += t_cnt[tid];  // (t_cnt: this slab's per-tile counts; the next slab starts with a barrier)
    };
    if constexpr (MODE == 0) {
        slab_body(0u);
    } else if constexpr (MODE == 1) {
        if (n_slabs != 0) slab_body(0u);  // a bin of <= MAXC: one slab, the whole of it
    } else if constexpr (MODE == 3) {
        const uint32_t n_bins = a.g.bins_x * a.g.bins_y;  // <= 1024
        {   // the on-screen bins, fullest first: rank by counting (<= 1024 broadcast reads per thread, once per workgroup)
            uint32_t* const cnt = smem;  // (LDS is free until the first item starts)
            const uint32_t mine = (uint32_t)tid < n_bins ? a.bin_count[screen_bin((uint32_t)tid)] : 0u;
            cnt[tid] = mine;
            if (tid == 0) q_planned = 0;
            __syncthreads();
            if (mine > (uint32_t)MAXC) atomicAdd(&q_planned, 1u);
            if ((uint32_t)tid < n_bins) {
                uint32_t before = 0;
                for (uint32_t j = 0; j < n_bins; ++j) {
                    const uint32_t v = cnt[j];
                    before += (v > mine || (v == mine && j < (uint32_t)tid)) ? 1u : 0u;
                }
                q_order[before] = (uint16_t)tid;
            }
            __syncthreads();
        }
        const uint32_t n_planned = (uint32_t)__builtin_amdgcn_readfirstlane((int)q_planned);  // the first n_planned items of the queue
        for (;;) {
            __syncthreads();  // the previous item is done with LDS and the shared variables
            refresh();  // (per item)
            if (tid == 0) {
                q_item = atomicAdd(&a.counters->q_head, 1u);
                s_flag = 0;
            }
            __syncthreads();
            const uint32_t item = (uint32_t)__builtin_amdgcn_readfirstlane((int)q_item);
            if (item < n_bins) {  // ---- a bin
                slab_item = false;
                bin = screen_bin((uint32_t)q_order[item]);
                bin_extent();
                c = multi ? 0u : c_total;
                rounds = (int)((c + THREADS - 1) / THREADS);
                wbase = (uint32_t)w * (uint32_t)rounds * WAVE;
                n_slabs = 1;
                g_kmin = 0;
                g_sh = 0;
                plan_bin();
            } else {
            // ---- a depth slab: wait until its descriptor is ready, or until every bin is done and it is not (then there is none)
            const uint32_t d = item - n_bins;
            if (tid == 0) {
                uint32_t have = 0;
                if (d < a.slab_capacity) {
                    for (uint32_t spin = 0;; ++spin) {
                        if (__hip_atomic_load(&a.slabs[d].ready, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == a.epoch) {
                            have = 1;
                            break;
                        }
                        if (__hip_atomic_load(&a.counters->q_bins_done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= n_planned) {
                            // every bin beyond MAXC has been planned (they head the queue, so this is early in the launch: nobody
                            // waits for the slowest ordinary bin), and every descriptor was released before its bin counted: one
                            // more look decides
                            // The final look must not be ordered ahead of the q_bins_done load above (two relaxed loads have no
                            // order of their own; round-5 advisor finding): the load has returned (vmcnt(0)) and the compiler may not
                            // move the next one across this point.  Both are sc1 loads past this CU's L1: nothing to invalidate.
                            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                            have = __hip_atomic_load(&a.slabs[d].ready, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == a.epoch ? 1u : 0u;
                            // belt and braces: every planned bin has counted its slabs by now, so a claimed index below that count
                            // whose descriptor is NOT ready means a slab would be dropped silently -- re-run the frame instead
                            if (!have && d < min(__hip_atomic_load(&a.counters->slabs, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), a.slab_capacity) &&
                                !(__hip_atomic_load(&a.counters->overflow, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & 2u))
                                atomicOr(&a.counters->overflow, 2u);
                            break;
                        }
                        if (spin > (1u << 20)) {  // (~ seconds: never expected; rather the global path than a hung queue)
                            atomicOr(&a.counters->overflow, 2u);
                            break;
                        }
                        __builtin_amdgcn_s_sleep(64);
                    }
                }
                q_item = have;
            }
            __syncthreads();
            if (__builtin_amdgcn_readfirstlane((int)q_item) == 0) break;  // no such slab: the queue is drained
            SlabDesc* desc = a.slabs + d;
            slab_item = true;
            // (written through by another workgroup of this launch: read past this CU's L1, word by word -- see plan_bin)
            auto get = [&](uint32_t* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); };
            off = (uint32_t)__builtin_amdgcn_readfirstlane((int)get(&desc->off));
            c_total = (uint32_t)__builtin_amdgcn_readfirstlane((int)get(&desc->c_total));
            g_kmin = (uint32_t)__builtin_amdgcn_readfirstlane((int)get(&desc->kmin));
            g_sh = __builtin_amdgcn_readfirstlane((int)get(reinterpret_cast<uint32_t*>(&desc->sh)));
            if (tid < 64) g_cur[tid] = get(&desc->cur[tid]);
            if (tid == 0) {
                slab_first[0] = get(&desc->b_lo);
                slab_first[1] = get(&desc->b_hi);
            }
            __syncthreads();
            c = 0;
            n_slabs = 1;
            }
            // a bin beyond MAXC counts as planned whatever became of it (descriptors written, or an overflow flagged): its ready
            // words have left (plan_bin waits for them) before the count
            if (!slab_item && item < n_planned) {
                __syncthreads();
                if (tid == 0) __hip_atomic_fetch_add(&a.counters->q_bins_done, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            if (n_slabs != 0) slab_body(0u);  // (one call site for both kinds of item: the body is large)
        }
    } else {
        const uint32_t n_desc = min(a.counters->slabs, a.slab_capacity);
        for (uint32_t d = blockIdx.x; d < n_desc; d += gridDim.x) {
            const SlabDesc* desc = a.slabs + d;
            __syncthreads();  // the previous slab is done with LDS and the shared variables
            refresh();  // (per slab)
            off = (uint32_t)__builtin_amdgcn_readfirstlane((int)desc->off);
            c_total = (uint32_t)__builtin_amdgcn_readfirstlane((int)desc->c_total);
            g_kmin = (uint32_t)__builtin_amdgcn_readfirstlane((int)desc->kmin);
            g_sh = __builtin_amdgcn_readfirstlane(desc->sh);
            if (tid < 64) g_cur[tid] = desc->cur[tid];
            if (tid == 0) {
                slab_first[0] = desc->b_lo;
                slab_first[1] = desc->b_hi;
                s_flag = 0;
            }
            __syncthreads();
            BUILD_ROW_SET(512u + (d & 511u));
            BUILD_T(1);
            slab_body(0u);
            BUILD_T(7);
        }
        BUILD_ROW_SET(1023u);
    }
    BUILD_T(7);
}
#undef multi
// 8 waves per SIMD for the two smaller sizes, i.e. two workgroups per CU: needs <= 64 VGPRs and <= 80 SGPRs (a SIMD has
// 800 SGPRs, allocated in sixteens plus sixteen per wave); the attribute wants a literal, hence three kernels
template <int ROUNDS> __global__ void k_bin_fast(BuildArgs a);
template <> __global__ __launch_bounds__(1024, 4) void k_bin_fast<4>(BuildArgs a) { bin_fast_body<4>(a); }
template <> __global__ __launch_bounds__(1024, 4) void k_bin_fast<8>(BuildArgs a) { bin_fast_body<8>(a); }
template <> __global__ __launch_bounds__(1024, 4) void k_bin_fast<12>(BuildArgs a) { bin_fast_body<12>(a); }
template <> __global__ __launch_bounds__(1024, 4) void k_bin_fast<16>(BuildArgs a) { bin_fast_body<16>(a); }
// bins of up to 65535 candidates, taken in depth slabs of <= 12288 (depth-order level 4): plan, then one workgroup per slab
#ifndef GS_SLAB_ROUNDS
#define GS_SLAB_ROUNDS 12  // candidates per thread of the slab kernels: slabs (and the bins k_bin_slabs takes itself) of <= 1024 x this
#endif
__global__ __launch_bounds__(1024, 4) void k_bin_slabs(BuildArgs a) { bin_fast_body<GS_SLAB_ROUNDS, 1>(a); }
__global__ __launch_bounds__(1024, 4) void k_slab_work(BuildArgs a) { bin_fast_body<GS_SLAB_ROUNDS, 2>(a); }
// the two of them as one launch of persistent workgroups over a queue of bins, then slabs
__global__ __launch_bounds__(1024, 4) void k_bin_queue(BuildArgs a) { bin_fast_body<GS_SLAB_ROUNDS, 3>(a); }

template <int R2, int THREADS, bool SORT>
static hipError_t build_prepare() {  // > 64 KiB of dynamic LDS needs the attribute
    return hipFuncSetAttribute(reinterpret_cast<const void*>(&k_bin_build<R2, THREADS, SORT>),
                               hipFuncAttributeMaxDynamicSharedMemorySize,
                               (int)(BuildLayout<R2, THREADS, SORT>::WORDS * sizeof(uint32_t)));
}
template <int ROUNDS>
static hipError_t fast_prepare() {
    return hipFuncSetAttribute(reinterpret_cast<const void*>(&k_bin_fast<ROUNDS>), hipFuncAttributeMaxDynamicSharedMemorySize,
                               (int)(FastLayout<ROUNDS>::WORDS * sizeof(uint32_t)));
}
// debug: what the runtime says about residency of the per-bin kernels (GS_DEBUG_OCCUPANCY=1 at renderer creation)
void bin_debug_occupancy() {
    int n4 = -1, n8 = -1, n16 = -1;
    (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&n4, reinterpret_cast<const void*>(&k_bin_fast<4>), 1024,
                                                       FastLayout<4>::WORDS * sizeof(uint32_t));
    (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&n8, reinterpret_cast<const void*>(&k_bin_fast<8>), 1024,
                                                       FastLayout<8>::WORDS * sizeof(uint32_t));
    (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&n16, reinterpret_cast<const void*>(&k_bin_fast<16>), 1024,
                                                       FastLayout<16>::WORDS * sizeof(uint32_t));
    hipFuncAttributes a4{}, a8{};
    (void)hipFuncGetAttributes(&a4, reinterpret_cast<const void*>(&k_bin_fast<4>));
    (void)hipFuncGetAttributes(&a8, reinterpret_cast<const void*>(&k_bin_fast<8>));
    std::fprintf(stderr, "[occupancy] k_bin_fast<4>: %d blocks/CU (regs %d, static lds %zu, dyn %zu); <8>: %d (regs %d); <16>: %d\n", n4,
                 a4.numRegs, a4.sharedSizeBytes, FastLayout<4>::WORDS * sizeof(uint32_t), n8, a8.numRegs, n16);
}

hipError_t bin_prepare_device() {  // once per device (gs_renderer::init)
    hipError_t e = build_prepare<4, 1024, true>();
    if (e == hipSuccess) e = build_prepare<16, 1024, true>();
    if (e == hipSuccess) e = fast_prepare<8>();
    if (e == hipSuccess) e = fast_prepare<12>();
    if (e == hipSuccess) e = fast_prepare<16>();
    if (e == hipSuccess)
        e = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_bin_slabs), hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)(FastLayout<GS_SLAB_ROUNDS>::WORDS * sizeof(uint32_t)));
    if (e == hipSuccess)
        e = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_slab_work), hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)(FastLayout<GS_SLAB_ROUNDS>::WORDS * sizeof(uint32_t)));
    if (e == hipSuccess)
        e = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_bin_queue), hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)(FastLayout<GS_SLAB_ROUNDS>::WORDS * sizeof(uint32_t)));
    return e;
}

template <int R2>
static void launch_build(const BuildArgs& a, bool sort, uint32_t bins, hipStream_t s) {
    constexpr size_t lds1 = BuildLayout<R2, 1024, true>::WORDS * sizeof(uint32_t);
    constexpr size_t lds2 = BuildLayout<R2, 1024, false>::WORDS * sizeof(uint32_t);
    if (sort) hipLaunchKernelGGL((k_bin_build<R2, 1024, true>), dim3(bins), dim3(1024), lds1, s, a);
    else hipLaunchKernelGGL((k_bin_build<R2, 1024, false>), dim3(bins), dim3(1024), lds2, s, a);
}

void launch_bin_level2(const BinLaunch& b, int level, hipStream_t s) {
    BuildArgs a;
    a.g = BinGrid{b.tiles_x, b.tiles_y, b.bins_x, b.bins_y, b.bin_shift, b.grid_shift};
    a.cand = b.cand;
    a.bin_count = b.bin_count;
    a.depth = b.depth;
    a.aabb = b.aabb;
    a.ranges = b.ranges;
    a.sorted_gid = b.sorted_gid;
    a.counters = b.counters;
    a.capacity = b.capacity;
    a.cand_capacity = b.cand_capacity;
    a.slabs = reinterpret_cast<SlabDesc*>(b.slabs);
    a.slab_capacity = b.slab_capacity;
    a.epoch = b.slab_epoch;
    a.stamps = b.stamps;
    const uint32_t bins = b.bins_x * b.bins_y;  // on-screen bins: the kernels map the block index onto the padded grid
    const bool sort = level < kBinSortLevels;
    if (sort && b.bin_shift <= 3) {  // bins of 4 x 4 or 8 x 8 tiles: the all-in-LDS kernel, sized by the level
        if (level == 0) hipLaunchKernelGGL(k_bin_fast<4>, dim3(bins), dim3(1024), FastLayout<4>::WORDS * sizeof(uint32_t), s, a);
        else if (level == 1) hipLaunchKernelGGL(k_bin_fast<8>, dim3(bins), dim3(1024), FastLayout<8>::WORDS * sizeof(uint32_t), s, a);
        else if (level == 2) hipLaunchKernelGGL(k_bin_fast<12>, dim3(bins), dim3(1024), FastLayout<12>::WORDS * sizeof(uint32_t), s, a);
        else if (level == 3) hipLaunchKernelGGL(k_bin_fast<16>, dim3(bins), dim3(1024), FastLayout<16>::WORDS * sizeof(uint32_t), s, a);
        else if (b.slab_epoch != 0) {
            // one launch: persistent workgroups (one per CU's worth of LDS; more than fit simply find the queue drained) over bins, then slabs
            hipLaunchKernelGGL(k_bin_queue, dim3(bins < kQueueWorkGroups ? bins : kQueueWorkGroups), dim3(1024), FastLayout<GS_SLAB_ROUNDS>::WORDS * sizeof(uint32_t), s, a);
        } else {
            hipLaunchKernelGGL(k_bin_slabs, dim3(bins), dim3(1024), FastLayout<GS_SLAB_ROUNDS>::WORDS * sizeof(uint32_t), s, a);
            hipLaunchKernelGGL(k_slab_work, dim3(kSlabWorkGroups), dim3(1024), FastLayout<GS_SLAB_ROUNDS>::WORDS * sizeof(uint32_t), s, a);
        }
    } else if (b.bin_shift <= 3) {
        launch_build<1>(a, sort, bins, s);
    } else if (b.bin_shift == 4) {
        launch_build<4>(a, sort, bins, s);
    } else {
        launch_build<16>(a, sort, bins, s);
    }
}
}  // namespace gs
