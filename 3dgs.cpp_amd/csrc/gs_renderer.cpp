// gs_renderer.cpp -- the renderer object and the render entry points of include/gs3d_hip.h.
//
// Frame orchestration replaces Renderer::recordPreprocessCommandBuffer / recordRenderCommandBuffer /
// draw (src/Renderer.cpp:468-529, 532-717, 366-426): every pass is enqueued on one HIP stream with
// grids that do not depend on the data-dependent counts V (visible) and D (instances); the counts
// live in device memory, so the reference's mid-frame fence wait + 4-byte readback + command-buffer
// re-record (Renderer.cpp:391-399, 538) disappears.  The frame's last kernel publishes the counters to
// pinned memory only to detect instance-buffer overflow (Renderer.cpp:541-563 grows and retries too)
// or a bin that outgrew the bin-local depth order (then the frame is re-run on the global path).
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <memory>
#include <string>
#include <vector>

#include "gs_blend_tuner.h"
#include "gs_internal.h"

using namespace gs_host;

// ------------------------------------------------------------------------------------------
// gs_renderer
// ------------------------------------------------------------------------------------------
#ifndef GS_L1_DENSE
#define GS_L1_DENSE 1  // 0 (A/B builds): level 1 walks the N-wide planes as in round 2
#endif
// One complete set of per-frame device buffers + the stream its passes run on.  Frames alternate between
// sets, so with >= 2 sets the small launch-bound passes of frame i+1 (scans, binning) overlap the
// VALU-bound blend of frame i on the same GPU.
struct FrameBuffers {
    hipStream_t stream = nullptr;
    // CU-partition experiment (GS_CU_MASK_PREP / GS_CU_MASK_BLEND): the blend runs on its own CU-masked stream
    hipStream_t blend_stream = nullptr;
    hipEvent_t prep_done = nullptr, blend_done = nullptr;
    bool blend_recorded = false;
    // per-Gaussian attributes
    DevBuf<uint32_t> tiles;
    DevBuf<float> depth;
    DevBuf<ushort4> aabb;
    DevBuf<gs::AttrRecord> rec;  // one 64-byte record per Gaussian: what the blend gathers
    DevBuf<uint4> vis;           // the frame's visible Gaussians as dense lists (gs::AttrView::vis): level 1's input on the bin-local path
    DevBuf<uint32_t> vis_count;  // the lists' counters, one per 128 bytes (the word behind each: the finished frame's count)
    bool planes_stale = false;   // the last frame on this set streamed the dense lists: tiles / depth / aabb were not written (taps rebuild them)
    uint32_t vis_region_slots = 0;
    // global depth order (only allocated when that path is taken)
    DevBuf<uint32_t> dkeys[2], dvals[2];
    DevBuf<uint32_t> block_hist, digit_total;
    // two-level binning
    DevBuf<uint32_t> l1_hist, bin_count;  // [padded bins][level-1 blocks], [1024 counts + 1024 offsets]
    DevBuf<uint32_t> cand;                // [3 x cand_capacity] bin-major candidates: 12-byte records {depth bits, id, box} on the bin-local path, plain ids otherwise
    DevBuf<uint32_t> sorted;              // [capacity + 4] per-tile lists, bin-major
    DevBuf<uint32_t> ranges;              // [T][2]
    DevBuf<uint8_t> slabs;                // depth-slab descriptors (level 4; allocated on first use)
    uint32_t slab_epoch = 0;              // k_bin_queue: one value per launch on these descriptors (BinLaunch::slab_epoch)
    DevBuf<gs::Counters> counters;
    DevBuf<uint64_t> stamps;              // the frame's timeline, written by its kernels (gs_kernels.h: FrameStamp)
    // HIP-graph replay (gs_set_graph_mode): the frame's fixed-shape launches captured once per configuration
    DevBuf<gs::FrameParams> params;
    // ... one per setting of the blend's lockstep: the tuner flips it several times per measurement, and re-capturing the frame on every
    // flip would make the measurement weigh the capture (round-5 advisor finding)
    hipGraphExec_t graph_execs[2] = {nullptr, nullptr};
    struct GraphKey {
        int level = -1, hw_exp = 0, contract = 1, bin_shift = -1;
        uint32_t width = 0, height = 0, capacity = 0, cand_capacity = 0;
        const void *tile_order = nullptr, *ranges = nullptr, *sh16 = nullptr;
        bool lockstep = false;
        bool operator==(const GraphKey& o) const {
            return lockstep == o.lockstep && level == o.level && hw_exp == o.hw_exp && contract == o.contract && bin_shift == o.bin_shift && width == o.width && height == o.height &&
                   capacity == o.capacity && cand_capacity == o.cand_capacity && tile_order == o.tile_order && ranges == o.ranges && sh16 == o.sh16;
        }
    } graph_keys[2];
    void drop_graph(int which) {
        if (graph_execs[which]) (void)hipGraphExecDestroy(graph_execs[which]);
        graph_execs[which] = nullptr;
        graph_keys[which] = GraphKey{};
    }
    void drop_graph() {
        drop_graph(0);
        drop_graph(1);
    }
    size_t n = 0;
    bool ready = false;

    void init(size_t n_, uint32_t capacity, uint32_t cand_capacity, const std::vector<uint32_t>& mask_prep, const std::vector<uint32_t>& mask_blend, bool dense_lists) {
        n = n_;
        if (!mask_prep.empty() && !mask_blend.empty()) {
            HIP_CHECK(hipExtStreamCreateWithCUMask(&stream, static_cast<uint32_t>(mask_prep.size()), mask_prep.data()));
            HIP_CHECK(hipExtStreamCreateWithCUMask(&blend_stream, static_cast<uint32_t>(mask_blend.size()), mask_blend.data()));
            HIP_CHECK(hipEventCreateWithFlags(&prep_done, hipEventDisableTiming));
            HIP_CHECK(hipEventCreateWithFlags(&blend_done, hipEventDisableTiming));
        } else {
            HIP_CHECK(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking));
        }
        tiles.alloc(n);
        depth.alloc(n);
        aabb.alloc(n);
        rec.alloc(n);
        vis_region_slots = gs::vis_region_slots(static_cast<uint32_t>(n));
        if (dense_lists) ensure_dense_lists();  // only scenes of >= dense_min Gaussians ever use them (16 B x N)
        l1_hist.alloc(1025 * static_cast<size_t>(gs::bin_level1_columns(static_cast<uint32_t>(n))));  // + the row of visible counts
        bin_count.alloc(2048);  // counts, then offsets (gs_bin.h: kBinOffsets)
        counters.alloc(1);
        stamps.alloc(gs::ST_COUNT);
        HIP_CHECK(hipMemset(stamps.p, 0, gs::ST_COUNT * sizeof(uint64_t)));
        params.alloc(1);
        set_capacity(capacity);
        set_cand_capacity(cand_capacity);
        ready = true;
    }
    void ensure_dense_lists() {  // the dense lists of visible Gaussians: only scenes of >= dense_min Gaussians ever use them (16 B x N)
        if (vis.p) return;
        vis.alloc(static_cast<size_t>(gs::kVisRegions) * vis_region_slots);
        vis_count.alloc(gs::kVisRegions * gs::kVisCounterStride);
        HIP_CHECK(hipMemset(vis_count.p, 0, vis_count.n * sizeof(uint32_t)));  // every frame's LAST kernel zeroes them again
    }
    void ensure_depth_order() {  // the global depth-order path's buffers
        if (dkeys[0].p) return;
        for (int k = 0; k < 2; ++k) {
            dkeys[k].alloc(n);
            dvals[k].alloc(n);
        }
        block_hist.alloc(256 * static_cast<size_t>(gs::kSortMaxBlocks));
        digit_total.alloc(256);
    }
    void set_capacity(uint32_t cap) {  // D: the per-tile lists
        drop_graph();  // the captured launches hold the old buffers
        sorted.alloc(static_cast<size_t>(cap) + 4);  // + 4: a 16-byte list store that starts inside the capacity may end past it
    }
    // E1: the level-1 candidates, sized on their own (round 5; they used to share the instance capacity: 3 words x 8 N, six times what
    // config B's 0.74 M records need).  A frame whose candidates overflow leaves a gap of unwritten entries that k_bin_build still
    // gathers through before the frame is re-run: the gap must hold valid Gaussian ids (0), never whatever hipMalloc handed back.
    void set_cand_capacity(uint32_t ccap) {
        drop_graph();
        cand.alloc(3 * static_cast<size_t>(ccap));
        HIP_CHECK(hipMemset(cand.p, 0, 3 * static_cast<size_t>(ccap) * sizeof(uint32_t)));
    }
    void sync() const {
        HIP_CHECK(hipStreamSynchronize(stream));
        if (blend_stream) HIP_CHECK(hipStreamSynchronize(blend_stream));
    }
    ~FrameBuffers() {
        drop_graph();
        if (stream) (void)hipStreamDestroy(stream);
        if (blend_stream) (void)hipStreamDestroy(blend_stream);
        if (prep_done) (void)hipEventDestroy(prep_done);
        if (blend_done) (void)hipEventDestroy(blend_done);
    }
};

struct FrameSlot {
    gs_uniforms u{};
    float* rgba = nullptr;
    uint8_t* bgra = nullptr;
    hipEvent_t done = nullptr;           // the frame's one event: completion (round 5 bracketed every pass with one: 4.5 us of idle GPU each)
    gs::Counters* h_counters = nullptr;  // pinned
    uint64_t* h_stamps = nullptr;        // pinned [ST_COUNT]: the frame's timeline, written by its kernels and copied here by k_frame_end
    gs::FrameParams* h_params = nullptr;  // pinned staging of the frame's parameter block (graph replay)
    bool timed = false;
    int level = 0;  // the depth-order level this frame ran at (gs_renderer::level)
    int bin_shift = 3;
    bool lockstep = false;      // the blend's lockstep setting this frame ran with
    uint32_t tune_round = 0;    // ... and the tuner's round it belongs to (samples of an earlier round are ignored)
    int tune_shape = 0;         // ... and which of the bank's tuners (one per frame shape) that was
};

struct gs_renderer {
    static constexpr int kMaxInFlight = 8;
    // twice what can be in flight: the frame-interval statistic reads the events of `latest_done`, which with frames
    // completing out of order across the sets may lie up to kMaxInFlight - 1 frames behind the oldest pending one
    static constexpr int kSlots = 2 * kMaxInFlight;

    gs_scene* scene = nullptr;
    bool timing = true;          // gs_set_timing: per-pass spans in the statistics (free since round 6: the kernels stamp them)
    double tick_ms = 1e-5;       // one tick of the kernels' clock in ms (100 MHz unless the device says otherwise)

    FrameBuffers sets[kMaxInFlight];
    int num_sets = 1;
    uint32_t capacity = 0;       // tile instances (D) the list buffers hold
    uint32_t cand_capacity = 0;  // level-1 candidates (E1) the candidate buffers hold
    FrameBuffers* last_set = nullptr;  // buffers of the most recently enqueued frame (stage taps)

    // frames in flight: a ring of descriptors, all enqueued on `stream` (so device buffers are
    // reused in stream order); the host only waits when the ring is full or on gs_synchronize.
    FrameSlot slots[kSlots];
    int in_flight_limit = 1;  // the reference has FRAMES_IN_FLIGHT 1 (VulkanContext.h:6)
    uint64_t frames_enqueued = 0;
    int pending = 0;

    gs_frame_stats last{};  // stats of the most recently retired frame

    // How a frame's per-tile lists get their depth order.  level 0 .. 4: bin-local -- the
    // workgroup that builds a bin's lists orders its candidates in LDS first (up to 4096 / 8192 / 12288 / 16384 per bin, or,
    // level 4, up to 65535 in depth slabs of <= 12288; 6 kernels per frame); level 5: global -- the V visible Gaussians are ordered first (12 more kernels; any bin size).
    // sort_mode 0 = automatic: start at level 0; a bin that does not fit re-runs the frame at the level its size asks
    // for; after 32 frames that would have fitted the level below, go back down.
    int sort_mode = 0;           // 0 auto, 1 global depth order, 2 bin-local (forced: a bin beyond 16384 is an error)
    int level = 0;
    uint32_t frames_since_fallback = 0;
    // Depth slabs (level 4) can fail for reasons that have nothing to do with the bin's size -- one depth bucket beyond a slab, a
    // run of more than 64 exactly equal depths inside one, more slabs than descriptors: the frame then goes to the global path,
    // and since `max_bin` still fits level 4 the step-down below would send it straight back into the same failure every 32
    // frames, for ever (round-3 advisor finding).  Each such failure doubles the frames the renderer stays on the global path
    // before it tries the slabs again (32 .. 8192); 64 clean frames at level 4 reset it.
    uint32_t slab_hold = 32, slab_clean_frames = 0;
    static constexpr int kGlobalLevel = gs::kBinSortLevels;
    static uint32_t level_limit(int lv) { return gs::kBinSortLimit[lv]; }
    int frame_level() const { return sort_mode == 1 ? kGlobalLevel : level; }
    bool graph_mode = false;     // replay each frame as one captured HIP graph (gs_set_graph_mode)
    // the blend's exp() (gs_set_exp_mode): 3 (default) the hardware's v_exp_f32 under the guard of render.comp:82 -- the reference's
    // decisions, its pixels to rounding noise; 2 libm's expf restated in binary64 -- the reference's bits; 0 pipeline polynomial, 1 v_exp_f32
    int exp_mode = 3;
    // a scene that holds an opacity > 1 is outside the guard's premises: blended with mode 2's arithmetic instead
    // (with the contractions on, mode 3 runs as mode 1 whatever the scene holds: there is nothing to guard -- gs3d_hip.h)
    int blend_exp_mode() const { return exp_mode == 3 && !contract && !scene->unit_opacity ? 2 : exp_mode; }
    bool contract = false;       // the three FMA contractions GLSL permits in render.comp:66,87 (gs_set_blend_contraction); default: as written
    // The blend's LOCKSTEP (gs_blend.hip): the four waves of a tile take every chunk of its list together, so that their gathers of the
    // same records meet in L1.  Worth +25 % of the blend on trained-like scenes (L1-miss-bound: T(6e6) 505 -> 378 us; T(1e6) with three
    // frames in flight 2 675 -> 3 575 frames/s), -9 % on the S scenes (pair-loop-bound).  Nothing the renderer knows up front tells the
    // two apart, so it MEASURES (gs_blend_tuner.h: the rate at which frames complete over OFF - ON - OFF windows; the frames are bit-identical either way), keeps
    // lockstep where it wins by 3 %, and looks again every 4096 frames or when the frame's shape changes.  GS_BLEND_LOCKSTEP=0 / 1 (or gs_set_blend_lockstep)
    // pins it; the tuner then rests.  One tuner per frame shape (BlendTunerBank): alternating resolutions do not restart each other.
    BlendTunerBank tuners;
    int min_bin_shift = 3;       // GS_BIN_SHIFT: log2 of the default bin edge in tiles (8 x 8 tiles)
    // GS_L1_DENSE_MIN: scenes of at least this many Gaussians hand level 1 the dense lists of visible Gaussians (measured
    // A/B, profiles/r03_l1_dense_lists_ab.txt: 6 M Gaussians +2 % one frame at a time, +2..7 % with three in flight --
    // level 1 is several rounds of workgroups there; 1 M: -0.5 %, level 1 is one round of workgroups bound by its round
    // trips and k_preprocess pays 2 us for the lists; 2 M and 3.5 M, profiles/r03_l1_dense_threshold.txt: -2.2 % / -0.5 %
    // one frame at a time, +0.5 % with three in flight).  The GPU tests set it to 0 for small scenes.
    uint64_t dense_min = 4u << 20;
    bool debug_levels = std::getenv("GS_DEBUG_LEVELS") != nullptr;
    bool level2_queue = !(std::getenv("GS_L2_QUEUE") && std::atoi(std::getenv("GS_L2_QUEUE")) == 0);
    // GS_DEBUG_STALLS=<ms>: a gs_render call that keeps the host longer than this is reported on stderr with the time each of
    // its parts took (wait for a free frame slot; the launches of each pass; the closing event records) -- how the runtime's
    // own hiccups (profiles/r05_stall_*.txt) are told from the renderer's
    double stall_ms = std::getenv("GS_DEBUG_STALLS") ? std::atof(std::getenv("GS_DEBUG_STALLS")) : 0.0;
    static constexpr int kLaps = 10;
    std::chrono::steady_clock::time_point laps[kLaps];
    void lap(int k) {
        if (stall_ms > 0.0) laps[k] = std::chrono::steady_clock::now();
    }
    void report_stall() {
        if (stall_ms <= 0.0) return;
        auto ms = [&](int a, int b) { return std::chrono::duration<double, std::milli>(laps[b] - laps[a]).count(); };
        if (ms(0, 9) < stall_ms) return;
        std::fprintf(stderr, "[gs3d] stall: frame %llu held the host %.3f ms: wait-for-slot %.3f, setup %.3f, preprocess %.3f, order %.3f, level1 %.3f, "
                             "level2 %.3f, blend %.3f, closing events %.3f\n", (unsigned long long)(frames_enqueued - 1), ms(0, 9), ms(0, 1), ms(1, 2), ms(2, 3),
                     ms(3, 4), ms(4, 5), ms(5, 6), ms(6, 7), ms(7, 9));
    }
    bool refined = false;        // bins of half that edge: taken when a bin outgrows the largest in-LDS order
    bool settle_level = false;   // the next clean frame at the level a refinement jumped to tells which level its bins really need
    bool have_frame = false;
    uint32_t retries = 0;        // lifetime count of re-run frames (statistics only)
    uint32_t redo_chain = 0;     // consecutive re-runs since a frame last retired cleanly: the runaway guard
    double total_ms[7] = {0, 0, 0, 0, 0, 0, 0};
    uint64_t total_frames = 0;
    uint64_t lifetime_frames = 0;  // frames retired since creation (gs_poll_stats)
    // completion-to-completion intervals of consecutive frames (the frame time a consumer sees with frames in flight)
    static constexpr size_t kIntervalRing = 8192;
    std::vector<float> intervals;
    bool prev_retired = false;  // the frame before the one being retired completed normally (its events are valid)
    uint64_t latest_done = 0;   // index of the retired frame whose blend ended last (at most sets - 1 frames back)

    uint32_t* sorted_gid = nullptr;  // result buffers of the last enqueued frame
    uint32_t* depth_order = nullptr;
    uint64_t num_tiles = 0;

    ~gs_renderer() {
        for (auto& sl : slots) {
            if (sl.done) (void)hipEventDestroy(sl.done);
            if (sl.h_counters) (void)hipHostFree(sl.h_counters);
            if (sl.h_stamps) (void)hipHostFree(sl.h_stamps);
            if (sl.h_params) (void)hipHostFree(sl.h_params);
        }
    }

    void set_capacity(uint32_t cap) {
        capacity = cap;
        for (auto& fb : sets)
            if (fb.ready) fb.set_capacity(cap);
    }
    void set_cand_capacity(uint32_t ccap) {
        cand_capacity = ccap;
        for (auto& fb : sets)
            if (fb.ready) fb.set_cand_capacity(ccap);
    }

    void init() {
        if (const char* e = std::getenv("GS_BLEND_LOCKSTEP")) tuners.pin(std::atoi(e) < 0 ? -1 : (std::atoi(e) != 0 ? 1 : 0));
        HIP_CHECK(hipSetDevice(scene->device));
        {   // the clock the kernels stamp the frame's timeline with (wall_clock64): constant rate, in kHz
            int khz = 0;
            if (hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, scene->device) == hipSuccess && khz > 0) tick_ms = 1.0 / khz;
        }
        HIP_CHECK(gs::bin_prepare_device());
        if (std::getenv("GS_DEBUG_OCCUPANCY")) gs::bin_debug_occupancy();
        for (auto& sl : slots) {
            HIP_CHECK(hipEventCreateWithFlags(&sl.done, hipEventDisableTiming));
            HIP_CHECK(hipHostMalloc(reinterpret_cast<void**>(&sl.h_stamps), gs::ST_COUNT * sizeof(uint64_t), hipHostMallocDefault));
            std::memset(sl.h_stamps, 0, gs::ST_COUNT * sizeof(uint64_t));
            HIP_CHECK(hipHostMalloc(reinterpret_cast<void**>(&sl.h_counters), sizeof(gs::Counters), hipHostMallocDefault));
            *sl.h_counters = gs::Counters{};
            HIP_CHECK(hipHostMalloc(reinterpret_cast<void**>(&sl.h_params), sizeof(gs::FrameParams), hipHostMallocDefault));
        }
        uint64_t want = std::max<uint64_t>(1u << 20, 8 * static_cast<uint64_t>(scene->n));
        // test knob: start small so that the overflow / grow / re-run machinery is exercised by small scenes
        if (const char* e = std::getenv("GS_INITIAL_CAPACITY")) want = std::max<uint64_t>(256, std::strtoull(e, nullptr, 10));
        capacity = static_cast<uint32_t>(std::min<uint64_t>(want, kMaxInstances));
        // candidates: E1 <= D always; ~1.5 per visible Gaussian with bins of 8 x 8 tiles, ~2 with 4 x 4 (config B 0.74 M, E 4.6 M, T 5.3 M)
        uint64_t want_cand = std::max<uint64_t>(1u << 18, 2 * static_cast<uint64_t>(scene->n));
        if (const char* e = std::getenv("GS_INITIAL_CAND_CAPACITY")) want_cand = std::max<uint64_t>(256, std::strtoull(e, nullptr, 10));
        else if (std::getenv("GS_INITIAL_CAPACITY")) want_cand = std::min<uint64_t>(want_cand, want);  // (the tests' small start applies to both)
        cand_capacity = static_cast<uint32_t>(std::min<uint64_t>(want_cand, capacity));
        parse_cu_masks();
        sets[0].init(scene->n, capacity, cand_capacity, mask_prep, mask_blend, scene->n >= dense_min);
    }

    // Experiment: GS_CU_MASK_PREP / GS_CU_MASK_BLEND = hex strings, most significant CU first,
    // 256 bits each.  With both set, a frame's passes before the blend run on a stream restricted to the first mask and
    // the blend on a stream restricted to the second, chained by events.
    std::vector<uint32_t> mask_prep, mask_blend;
    static std::vector<uint32_t> parse_mask(const char* hex) {
        std::vector<uint32_t> words;
        if (!hex) return words;
        std::string h(hex);
        while (h.size() % 8) h.insert(h.begin(), '0');
        for (size_t i = h.size(); i >= 8; i -= 8) words.push_back(static_cast<uint32_t>(std::stoul(h.substr(i - 8, 8), nullptr, 16)));
        return words;
    }
    void parse_cu_masks() {
        mask_prep = parse_mask(std::getenv("GS_CU_MASK_PREP"));
        mask_blend = parse_mask(std::getenv("GS_CU_MASK_BLEND"));
    }

    void set_num_sets(int k) {
        for (int i = 0; i < k; ++i)
            if (!sets[i].ready) sets[i].init(scene->n, capacity, cand_capacity, mask_prep, mask_blend, scene->n >= dense_min);
        num_sets = k;
    }

    // Blend workgroup -> tile table (shared by all buffer sets, rebuilt when the tile grid changes).  Workgroup b
    // runs on XCD b % 8 (observed dispatch rule), each XCD has a private L2, and the dispatcher hands out workgroups
    // in order, so a heavily loaded XCD holds the others back.  The screen is cut into blocks of B x B tiles and the
    // blocks are dealt to the XCDs like a skewed checkerboard: every XCD gets blocks from all over the image (balanced
    // for any scene) and the tiles of a block, which share most of their splat records, meet in one L2.
    DevBuf<uint32_t> tile_order;
    uint32_t order_tx = 0, order_ty = 0;
    void ensure_tile_order(uint32_t tx, uint32_t ty) {
        if (tx == order_tx && ty == order_ty && tile_order.p) return;
        drain();
        // B = 4 (64 x 64 px): on a clustered scene the blend takes 0.218 ms against 0.241 ms with one contiguous band
        // of tiles per XCD (max/mean XCD load 1.01 against 1.54); B = 2, 6, 8 and the bands all measured equal or slower
        constexpr uint32_t B = 4;
        const uint64_t nt = static_cast<uint64_t>(tx) * ty;
        std::vector<std::vector<uint32_t>> per_xcd(8);
        const uint32_t nbx = (tx + B - 1) / B, nby = (ty + B - 1) / B;
        for (uint32_t by = 0; by < nby; ++by)
            for (uint32_t bx = 0; bx < nbx; ++bx) {
                auto& list = per_xcd[(bx + 3 * by) % 8];
                for (uint32_t y = by * B; y < std::min(ty, (by + 1) * B); ++y)
                    for (uint32_t x = bx * B; x < std::min(tx, (bx + 1) * B); ++x) list.push_back(y * tx + x);
            }
        // workgroup b takes the next tile of XCD b % 8's list; lists that run dry borrow from the longest one
        std::vector<uint32_t> order(nt);
        size_t cursor[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        for (uint64_t b = 0; b < nt; ++b) {
            int x = static_cast<int>(b % 8);
            if (cursor[x] >= per_xcd[x].size()) {
                size_t best = 0;
                for (int k = 0; k < 8; ++k)
                    if (per_xcd[k].size() - cursor[k] > best) best = per_xcd[k].size() - cursor[k], x = k;
            }
            order[b] = per_xcd[x][cursor[x]++];
        }
        tile_order.ensure(nt);
        if (nt) HIP_CHECK(hipMemcpy(tile_order.p, order.data(), nt * sizeof(uint32_t), hipMemcpyHostToDevice));
        order_tx = tx;
        order_ty = ty;
    }

    // The bin grid of a frame: bins of S x S tiles, at most 32 x 32 of them, padded to a 16- or 32-wide grid.
    struct BinGeometry {
        int bin_shift, grid_shift;
        uint32_t bins_x, bins_y;
    };
    static bool grid_fits(uint32_t tx, uint32_t ty, int s) { return (((tx - 1) >> s) + 1) <= 32 && (((ty - 1) >> s) + 1) <= 32; }
    // the coarsest-allowed choice: bins of 8 x 8 tiles (or GS_BIN_SHIFT), larger only to keep the grid within 32 x 32
    static int base_shift(uint32_t tx, uint32_t ty, int min_shift) {
        int s = std::max(2, min_shift);
        while (!grid_fits(tx, ty, s)) ++s;
        return s;
    }
    BinGeometry bin_geometry(uint32_t tx, uint32_t ty) const {
        int s = base_shift(tx, ty, min_bin_shift);
        if (s > 5) throw Error(GS_ERR_INVALID, "resolution too large for the tile binning (max 16384 x 16384)");
        // `refined`: a bin outgrew the largest in-LDS order -> bins of half the edge (a quarter of the candidates or so)
        if (refined && s > 2 && grid_fits(tx, ty, s - 1)) --s;
        BinGeometry g;
        g.bin_shift = s;
        g.bins_x = ((tx - 1) >> s) + 1;
        g.bins_y = ((ty - 1) >> s) + 1;
        g.grid_shift = (g.bins_x <= 16 && g.bins_y <= 16) ? 4 : 5;
        return g;
    }
    bool can_refine(const gs_uniforms& u) const {
        const uint32_t tx = (u.width + gs::kTile - 1) / gs::kTile, ty = (u.height + gs::kTile - 1) / gs::kTile;
        const int s = base_shift(tx, ty, min_bin_shift);
        return !refined && s > 2 && s <= 5 && grid_fits(tx, ty, s - 1);
    }

    void enqueue(const gs_uniforms& u, float* d_rgba, uint8_t* d_bgra) {
        HIP_CHECK(hipSetDevice(scene->device));
        FrameSlot& sl = slots[frames_enqueued % kSlots];
        FrameBuffers& fb = sets[frames_enqueued % num_sets];
        hipStream_t stream = fb.stream;
        const uint32_t n = static_cast<uint32_t>(scene->n);
        const uint32_t tx = (u.width + gs::kTile - 1) / gs::kTile, ty = (u.height + gs::kTile - 1) / gs::kTile;
        if (tx > 65535 || ty > 65535) throw Error(GS_ERR_INVALID, "resolution too large (tile box is 16-bit)");
        const uint64_t nt = static_cast<uint64_t>(tx) * ty;
        auto lacks_buffers = [&](int at_level) {
            return 2 * nt > fb.ranges.n || (at_level >= kGlobalLevel && !fb.dkeys[0].p) || (at_level == gs::kBinSlabLevel && !fb.slabs.p);
        };
        if (lacks_buffers(frame_level())) {
            // (re)allocation: wait for queued frames that still use the old buffers.  Retiring them may re-run a frame at another
            // depth-order level or bin size (retire_oldest): what THIS frame runs with is decided after the wait, not before
            // -- a frame queued with the level of before the wait fails at once and, being judged as a failure of the new
            // level, used to push the renderer onto the global path for slab_hold frames
            drain();
            const int at_level = frame_level();
            fb.ranges.ensure(2 * nt);
            if (at_level >= kGlobalLevel) fb.ensure_depth_order();
            if (at_level == gs::kBinSlabLevel && !fb.slabs.p) {
                fb.slabs.alloc(static_cast<size_t>(gs::kSlabCapacity) * gs::kSlabDescBytes);
                HIP_CHECK(hipMemset(fb.slabs.p, 0, fb.slabs.n));  // no descriptor carries a launch's epoch yet
            }
        }
        // (after any drain above: retiring may re-run frames through enqueue, which would leave another set's buffers here)
        last_set = &fb;
        const int tune_shape = tuners.select(u.width, u.height);  // this frame shape's own tuner (a new shape measures afresh)
        const BlendTuner& tuner = tuners.current();
        const bool lockstep = tuner.current();
        const BinGeometry geo = bin_geometry(tx, ty);
        const int lv = frame_level();
        const bool bin_local = lv < kGlobalLevel;
        num_tiles = nt;
        ensure_tile_order(tx, ty);
        lap(2);

        gs::SceneView sv{scene->render_blob(), scene->cov3d.p, n, static_cast<uint32_t>(gs::blob_stride(n)),
                          scene->sh_half ? scene->sh16.p : nullptr, scene->acut.p, scene->perm.p};
        gs::Counters* cnt = fb.counters.p;
        // the level-1 kernels that take their items in any order (bin-local path, bins of <= 8 x 8 tiles) stream the dense
        // list of visible Gaussians, which k_preprocess then writes beside the planes
        const bool l1_any_order = bin_local && geo.bin_shift <= 3;
        // (the lists exist only for scenes of >= dense_min Gaussians: FrameBuffers::init; the blend zeroes their counters)
        const bool dense_list = GS_L1_DENSE && l1_any_order && n != 0 && n >= dense_min && fb.vis.p && u.width != 0 && u.height != 0;
        gs::AttrView av{fb.tiles.p, fb.depth.p, fb.aabb.p, fb.rec.p, dense_list ? fb.vis.p : nullptr, dense_list ? fb.vis_count.p : nullptr,
                        fb.vis_region_slots};
        fb.planes_stale = dense_list;

        // the first and the last kernel of the frame clear / publish the counters themselves; the blit nodes (and
        // their fences) are only needed when one of the two is not launched
        const bool fused_counters = n != 0 && u.width != 0 && u.height != 0;
        if (!fused_counters) {
            HIP_CHECK(hipMemsetAsync(cnt, 0, sizeof(gs::Counters), stream));
            HIP_CHECK(hipMemsetAsync(fb.ranges.p, 0, 2 * nt * sizeof(uint32_t), stream));  // no Gaussians: every tile (0, 0)
        }
        if (fb.blend_stream && fb.blend_recorded) HIP_CHECK(hipStreamWaitEvent(stream, fb.blend_done, 0));  // the set's previous blend
        // the frame's launches; `fp` non-null = replayable form (per-frame values read from fb.params), no span events
        uint64_t* const stamps = fb.stamps.p;
        auto passes = [&](const gs::FrameParams* fp, hipStream_t bstream) {
            gs::launch_preprocess(sv, u, av, cnt, fp, stamps, stream);
            lap(3);
            if (!bin_local && n != 0) {
                // ---- global depth order of the visible Gaussians: 4 x 8-bit stable passes on bits(depth) ----
                const int blocks = std::max(1, std::min<int>(gs::kSortMaxBlocks, (n + gs::kSortTileKeys - 1) / gs::kSortTileKeys));
                const uint32_t* kin = reinterpret_cast<const uint32_t*>(fb.depth.p);
                const uint32_t* vin = nullptr;
                for (int pass = 0; pass < 4; ++pass) {
                    gs::RadixPass p{};
                    const int dst = pass & 1;
                    p.keys_in = kin;
                    p.vals_in = vin;
                    p.keys_out = fb.dkeys[dst].p;
                    p.vals_out = fb.dvals[dst].p;
                    p.n_in = &cnt->visible;
                    p.n_static = n;
                    p.tiles = fb.tiles.p;
                    p.n_out = &cnt->visible;
                    p.block_hist = fb.block_hist.p;
                    p.digit_total = fb.digit_total.p;
                    p.shift = pass * 8;
                    p.bits = 8;
                    p.blocks = blocks;
                    p.first = pass == 0;
                    p.stamps = pass == 0 ? stamps : nullptr;
                    gs::launch_radix_pass(p, stream);
                    kin = fb.dkeys[dst].p;
                    vin = fb.dvals[dst].p;
                }
            }
            lap(4);
            if (n != 0) {
                gs::BinLaunch b{};
                b.order = bin_local ? nullptr : fb.dvals[1].p;
                b.n_items = bin_local ? nullptr : &cnt->visible;
                b.n_bound = n;
                b.tiles = fb.tiles.p;
                b.aabb = fb.aabb.p;
                b.depth = fb.depth.p;
                b.vis = av.vis;
                b.vis_count = av.vis_count;
                b.vis_region_slots = av.vis_region_slots;
                b.hist = fb.l1_hist.p;
                b.bin_count = fb.bin_count.p;
                b.cand = fb.cand.p;
                b.ranges = fb.ranges.p;
                b.sorted_gid = fb.sorted.p;
                b.counters = cnt;
                b.capacity = capacity;
                b.cand_capacity = cand_capacity;
                b.slabs = fb.slabs.p;
                b.slab_capacity = fb.slabs.p ? gs::kSlabCapacity : 0u;
                // level 4 as one launch over a queue of bins and slabs -- not in a captured frame (a replay repeats its arguments,
                // and a descriptor is ready when it holds THIS launch's epoch); GS_L2_QUEUE=0: the two launches of round 4
                b.slab_epoch = 0;
                if (level2_queue && !fp && lv == gs::kBinSlabLevel) {
                    if (++fb.slab_epoch == 0) ++fb.slab_epoch;
                    b.slab_epoch = fb.slab_epoch;
                }
                b.tiles_x = tx;
                b.tiles_y = ty;
                b.bins_x = geo.bins_x;
                b.bins_y = geo.bins_y;
                b.bin_shift = geo.bin_shift;
                b.grid_shift = geo.grid_shift;
                b.stamps = stamps;
                // ---- level 1: which Gaussian touches which bin (count + scan, then the per-bin candidate lists) ----
                gs::launch_bin_level1_count(b, stream);
                gs::launch_bin_level1_scatter(b, l1_any_order, stream);
                lap(5);
                // ---- level 2: order inside the bin (bin-local path), tile ranges, per-tile lists ----
                gs::launch_bin_level2(b, lv, stream);
            }
            lap(6);
            // ---- blend ----
            if (bstream != stream) {
                HIP_CHECK(hipEventRecord(fb.prep_done, stream));
                HIP_CHECK(hipStreamWaitEvent(bstream, fb.prep_done, 0));
            }
            gs::launch_blend(fb.ranges.p, fb.sorted.p, tile_order.p, av, u.width, u.height, d_rgba, d_bgra, cnt,
                             fused_counters ? sl.h_counters : nullptr, blend_exp_mode(), contract, fp, lockstep, stamps, bstream);
            gs::launch_frame_end(stamps, sl.h_stamps, fp, bstream);
            lap(7);
        };
        depth_order = bin_local ? nullptr : fb.dvals[1].p;
        sorted_gid = fb.sorted.p;
        hipStream_t bstream = fb.blend_stream ? fb.blend_stream : stream;
        const bool replay = graph_mode && fused_counters && !fb.blend_stream;
        if (replay) {
            *sl.h_params = gs::FrameParams{u, d_rgba, d_bgra, sl.h_counters, sl.h_stamps};
            HIP_CHECK(hipMemcpyAsync(fb.params.p, sl.h_params, sizeof(gs::FrameParams), hipMemcpyHostToDevice, stream));
            FrameBuffers::GraphKey key;
            key.level = lv;
            key.bin_shift = geo.bin_shift;
            key.hw_exp = blend_exp_mode();
            key.contract = contract ? 1 : 0;
            key.width = u.width;
            key.height = u.height;
            key.capacity = capacity;
            key.cand_capacity = cand_capacity;
            key.tile_order = tile_order.p;
            key.ranges = fb.ranges.p;
            key.sh16 = sv.sh16;
            key.lockstep = lockstep;
            const int gi = lockstep ? 1 : 0;
            if (!fb.graph_execs[gi] || !(key == fb.graph_keys[gi])) {  // first frame of this configuration: capture its launches
                fb.drop_graph(gi);
                hipGraph_t graph = nullptr;
                HIP_CHECK(hipStreamBeginCapture(stream, hipStreamCaptureModeThreadLocal));
                try {
                    passes(fb.params.p, stream);
                } catch (...) {
                    (void)hipStreamEndCapture(stream, &graph);
                    if (graph) (void)hipGraphDestroy(graph);
                    throw;
                }
                HIP_CHECK(hipStreamEndCapture(stream, &graph));
                const hipError_t e = hipGraphInstantiate(&fb.graph_execs[gi], graph, nullptr, nullptr, 0);
                (void)hipGraphDestroy(graph);
                HIP_CHECK(e);
                fb.graph_keys[gi] = key;
            }
            HIP_CHECK(hipGraphLaunch(fb.graph_execs[gi], stream));
        } else {
            passes(nullptr, bstream);
        }
        if (!fused_counters) HIP_CHECK(hipMemcpyAsync(sl.h_counters, cnt, sizeof(gs::Counters), hipMemcpyDeviceToHost, bstream));
        HIP_CHECK(hipEventRecord(sl.done, bstream));
        if (fb.blend_stream) {
            HIP_CHECK(hipEventRecord(fb.blend_done, bstream));
            fb.blend_recorded = true;
        }
        HIP_CHECK(hipGetLastError());

        sl.level = lv;
        sl.bin_shift = geo.bin_shift;
        sl.lockstep = lockstep;
        sl.tune_round = tuner.round;
        sl.tune_shape = tune_shape;
        sl.u = u;
        sl.rgba = d_rgba;
        sl.bgra = d_bgra;
        sl.timed = fused_counters;  // (a frame without Gaussians or pixels launches nothing that stamps)
        ++frames_enqueued;
        ++pending;
        lap(9);
    }

    FrameSlot& oldest() { return slots[(frames_enqueued - pending) % kSlots]; }

    // Wait for the oldest queued frame; record its stats; on instance-buffer overflow grow the
    // buffers and re-run it and every frame queued behind it (Renderer.cpp:541-563 retries too).
    void retire_oldest() {
        FrameSlot& sl = oldest();
        HIP_CHECK(hipEventSynchronize(sl.done));
        if (sl.h_counters->overflow) {
            for (auto& fb : sets)
                if (fb.ready) fb.sync();
            struct Redo {
                gs_uniforms u;
                float* rgba;
                uint8_t* bgra;
            };
            std::vector<Redo> redo;
            uint64_t need = 0, need_cand = 0;
            uint32_t fullest = 0;
            bool grow = false, bin_too_big = false;
            for (int k = 0; k < pending; ++k) {
                FrameSlot& q = slots[(frames_enqueued - pending + k) % kSlots];
                redo.push_back({q.u, q.rgba, q.bgra});
                if (q.h_counters->overflow & 1u) {
                    grow = true;
                    // which of the two ran over: the level-1 candidates (E1: the frame's instance count then means nothing -- its
                    // bins were skipped) or the per-tile lists (D)
                    if (q.h_counters->bin_entries > cand_capacity) need_cand = std::max<uint64_t>(need_cand, q.h_counters->bin_entries);
                    else need = std::max<uint64_t>(need, q.h_counters->instances);
                }
                // (a frame that ran with another level or bin size than the renderer's current ones says nothing about those)
                const uint32_t qtx = (q.u.width + gs::kTile - 1) / gs::kTile, qty = (q.u.height + gs::kTile - 1) / gs::kTile;
                const bool current = q.level == frame_level() && q.bin_shift == bin_geometry(qtx, qty).bin_shift;
                if (q.level < kGlobalLevel && (q.h_counters->overflow & 2u) && current) bin_too_big = true;
                if (current) fullest = std::max(fullest, q.h_counters->max_bin);
            }
            const int failed_level = sl.level;
            if (debug_levels) {  // GS_DEBUG_LEVELS: what made the renderer change its depth-order level
                std::fprintf(stderr, "[gs3d] frame %llu overflowed at level %d (refined %d, hold %u):", (unsigned long long)(frames_enqueued - pending),
                             failed_level, (int)refined, slab_hold);
                for (int k = 0; k < pending; ++k) {
                    const FrameSlot& q = slots[(frames_enqueued - pending + k) % kSlots];
                    std::fprintf(stderr, " [lvl %d bin 2^%d ovf %u max_bin %u E1 %u D %u slabs %u]", q.level, q.bin_shift, q.h_counters->overflow,
                                 q.h_counters->max_bin, q.h_counters->bin_entries, q.h_counters->instances, q.h_counters->slabs);
                }
                std::fprintf(stderr, " capacity %u candidates %u\n", capacity, cand_capacity);
            }
            // the queued frames are dropped from the ring first: whatever is thrown below, the renderer stays usable
            frames_enqueued -= pending;
            pending = 0;
            prev_retired = false;
            if (bin_too_big) {  // a bin outgrew the in-LDS order of this level: one level up from here on
                const bool slabs_unsuitable = failed_level == gs::kBinSlabLevel && fullest <= level_limit(gs::kBinSlabLevel);
                if (slabs_unsuitable) {  // not the bin's size: equal or crowded depths (see slab_hold)
                    slab_hold = std::min<uint32_t>(slab_hold * 2, 8192);
                    slab_clean_frames = 0;
                }
                int wanted = failed_level + 1;
                while (wanted < kGlobalLevel && fullest > level_limit(wanted)) ++wanted;
                if (wanted >= gs::kBinSlabLevel && can_refine(sl.u)) {  // smaller bins before slabs or the global path
                    refined = true;
                    wanted = gs::kBinSlabLevel - 1;  // (what the smaller bins hold is not known yet: the largest in-LDS order)
                    settle_level = true;
                } else {
                    if (sort_mode == 2 && wanted >= kGlobalLevel)
                        throw Error(GS_ERR_OVERFLOW, slabs_unsuitable
                                        ? "a bin's depths are too crowded for the bin-local order (one depth bucket beyond a slab, or more than 64 equal depths in one): needs the global depth-order path"
                                        : "a bin holds more candidates than the bin-local sort can order");
                    level = std::max(level, wanted);
                }
                if (refined) level = std::max(level, wanted);
                frames_since_fallback = 0;
            }
            // runaway guard: one frame may need a path fall-back and a few grow steps (each grow is sized from the counts
            // the overflowing frame reported, so it converges at once unless the chunk table and the lists take turns)
            if (++redo_chain > 8) throw Error(GS_ERR_OVERFLOW, "instance buffers overflowed repeatedly");
            if (grow && need_cand > cand_capacity) {
                need_cand = need_cand + need_cand / 2 + 4096;  // 1.5x head-room: a moving camera should not re-grow every few frames
                if (need_cand > kMaxInstances) throw Error(GS_ERR_OVERFLOW, "more than 2^30 level-1 candidates");
                set_cand_capacity(static_cast<uint32_t>(need_cand));
            }
            if (grow && need > capacity) {
                need = need + need / 2 + 4096;
                if (need > kMaxInstances) throw Error(GS_ERR_OVERFLOW, "more than 2^30 tile instances");
                set_capacity(static_cast<uint32_t>(need));
            }
            ++retries;
            for (const Redo& f : redo) enqueue(f.u, f.rgba, f.bgra);
            return;
        }
        redo_chain = 0;
        if (sl.level == gs::kBinSlabLevel && ++slab_clean_frames >= 64) slab_hold = 32;  // the slabs work on this scene (again)
        if (settle_level && sort_mode != 1 && sl.level == level && level < kGlobalLevel) {
            // the first clean frame after the bins were refined: its fullest bin says which order the smaller bins need -- straight
            // there instead of 32 frames at the largest one per step down (config C: level 3 -> 2, k_bin_fast<16> -> <12>)
            while (level > 0 && sl.h_counters->max_bin <= level_limit(level - 1) * 7 / 8) --level;
            frames_since_fallback = 0;
            settle_level = false;
        }
        if (sort_mode != 1 && level > 0) {  // one level down once the bins have fitted it for a while
            if (sl.h_counters->max_bin <= level_limit(level - 1) * 7 / 8) {
                // (from the global path back to the slabs: only after slab_hold frames, see there)
                if (++frames_since_fallback >= (level == kGlobalLevel ? slab_hold : 32u)) {
                    --level;
                    frames_since_fallback = 0;
                }
            } else {
                frames_since_fallback = 0;
            }
        } else if (sort_mode != 1 && refined) {  // at the smallest order with the small bins: try the default bins again
            if (sl.h_counters->max_bin <= level_limit(0) / 2) {  // four times the tiles per bin should still fit level 3 (<= 16384)
                if (++frames_since_fallback >= 32) {
                    refined = false;
                    level = gs::kBinSlabLevel - 1;
                    frames_since_fallback = 0;
                }
            } else {
                frames_since_fallback = 0;
            }
        }
        gs_frame_stats st{};
        st.num_gaussians = scene->n;
        st.num_visible = sl.h_counters->visible;
        st.num_instances = sl.h_counters->instances;
        st.num_bin_entries = sl.h_counters->bin_entries;
        st.max_bin_entries = sl.h_counters->max_bin;
        st.sort_path = sl.level < kGlobalLevel ? 2u : 1u;
        st.sort_level = static_cast<uint32_t>(sl.level);
        st.bin_tiles = 1u << sl.bin_shift;
        st.instance_capacity = capacity;
        // The frame's timeline as its kernels stamped it (gs_kernels.h: FrameStamp), in ticks of the device's constant-rate clock: a
        // span runs from the start of a pass's first kernel to the start of the next pass's -- the seam behind a pass belongs to it.
        const uint64_t* const ts = sl.h_stamps;
        auto span = [&](int a, int b) {
            const int64_t d = static_cast<int64_t>(ts[b] - ts[a]);
            return d > 0 ? static_cast<float>(static_cast<double>(d) * tick_ms) : 0.0f;
        };
        if (sl.timed) st.ms_total = span(gs::ST_PREPROCESS, gs::ST_END);
        if (sl.timed && timing) {
            // The reference's six span names (Renderer.cpp:484-526, 580-699).  prefix_sum = the level-1 count + scan,
            // preprocess_sort = the level-1 scatter (what lands where), sort = the global depth order (when taken) +
            // k_bin_build.  k_bin_build also produces the tile ranges: tile_boundary.comp's work has no kernel of
            // its own any more, so that span is 0 by construction.
            const bool global = sl.level >= kGlobalLevel;  // (only then is ST_ORDER this frame's)
            st.ms_preprocess = span(gs::ST_PREPROCESS, global ? gs::ST_ORDER : gs::ST_L1_COUNT);
            st.ms_prefix_sum = span(gs::ST_L1_COUNT, gs::ST_L1_SCATTER);
            st.ms_preprocess_sort = span(gs::ST_L1_SCATTER, gs::ST_L2);
            st.ms_sort = (global ? span(gs::ST_ORDER, gs::ST_L1_COUNT) : 0.0f) + span(gs::ST_L2, gs::ST_BLEND);
            st.ms_tile_boundary = 0.0f;
            st.ms_render = span(gs::ST_BLEND, gs::ST_END);
        }
        st.retries = retries;
        last = st;
        have_frame = true;
        const float v[7] = {st.ms_preprocess, st.ms_prefix_sum, st.ms_preprocess_sort, st.ms_sort,
                            st.ms_tile_boundary, st.ms_render, st.ms_total};
        for (int k = 0; k < 7; ++k) total_ms[k] += v[k];
        ++total_frames;
        ++lifetime_frames;
        {   // frames on different streams may finish out of order: measure against the latest completion so far
            const uint64_t idx = frames_enqueued - pending;  // this frame
            if (prev_retired) {
                // (the frames' END stamps: one clock for every stream)
                const int64_t dticks = static_cast<int64_t>(ts[gs::ST_END] - slots[latest_done % kSlots].h_stamps[gs::ST_END]);
                const float dt = dticks > 0 ? static_cast<float>(static_cast<double>(dticks) * tick_ms) : 0.0f;
                {
                    if (intervals.size() >= kIntervalRing) intervals.erase(intervals.begin(), intervals.begin() + kIntervalRing / 2);
                    intervals.push_back(dt > 0.0f ? dt : 0.0f);  // 0: it had already finished when its predecessor did
                    // the blend tuner compares completion rates (or, for a host-paced consumer, the frames' own spans: BlendTuner::cost)
                    tuners.sample(sl.tune_shape, sl.u.width, sl.u.height, dt > 0.0f ? dt : 0.0f, st.ms_total, sl.lockstep, sl.tune_round);
                    if (dt > 0.0f) latest_done = idx;
                }
            } else {
                latest_done = idx;
            }
            prev_retired = true;
        }
        --pending;
    }

    void make_room() {
        while (pending >= in_flight_limit) retire_oldest();
    }
    void drain() {
        while (pending > 0) retire_oldest();
    }
};

// ------------------------------------------------------------------------------------------
// C ABI
// ------------------------------------------------------------------------------------------
extern "C" {

int gs_renderer_create(gs_scene* scene, gs_renderer** out) {
    return guarded([&] {
        if (!scene || !out) throw Error(GS_ERR_INVALID, "null argument");
        auto r = std::make_unique<gs_renderer>();
        r->scene = scene;
        if (const char* e = std::getenv("GS_L1_DENSE_MIN")) r->dense_min = std::strtoull(e, nullptr, 10);  // (before init: sizes the buffer sets)
        r->init();
        if (const char* e = std::getenv("GS_GRAPH")) r->graph_mode = std::atoi(e) != 0;  // initial gs_set_graph_mode
        if (const char* e = std::getenv("GS_EXP_MODE")) r->exp_mode = std::min(3, std::max(0, std::atoi(e)));  // initial gs_set_exp_mode
        if (const char* e = std::getenv("GS_BLEND_CONTRACTION")) r->contract = std::atoi(e) != 0;  // initial gs_set_blend_contraction
        if (const char* e = std::getenv("GS_BIN_SHIFT")) r->min_bin_shift = std::min(5, std::max(2, std::atoi(e)));  // default bin edge
        if (const char* e = std::getenv("GS_SORT_PATH")) {  // initial gs_set_sort_path, for hosts that cannot call it (the viewer)
            const int mode = std::atoi(e);
            if (mode < 0 || mode > 2) throw Error(GS_ERR_INVALID, "GS_SORT_PATH must be 0 (auto), 1 (global) or 2 (bin-local)");
            r->sort_mode = mode;
        }
        *out = r.release();
    });
}

void gs_renderer_destroy(gs_renderer* r) {
    if (r)
        for (auto& fb : r->sets)
            if (fb.ready) {
                (void)hipStreamSynchronize(fb.stream);
                if (fb.blend_stream) (void)hipStreamSynchronize(fb.blend_stream);
            }
    delete r;
}

int gs_render(gs_renderer* r, const gs_uniforms* u, float* d_rgba, uint8_t* d_bgra) {
    return guarded([&] {
        if (!r || !u) throw Error(GS_ERR_INVALID, "null argument");
        if (u->width == 0 || u->height == 0) throw Error(GS_ERR_INVALID, "empty framebuffer");
        r->lap(0);
        r->make_room();  // at most in_flight_limit frames queued; resolves pending overflows first
        r->lap(1);
        if (r->stall_ms > 0.0)
            for (int k = 2; k < gs_renderer::kLaps; ++k) r->laps[k] = r->laps[1];
        r->enqueue(*u, d_rgba, d_bgra);
        r->report_stall();
    });
}

int gs_render_host(gs_renderer* r, const gs_uniforms* u, float* h_rgba, uint8_t* h_bgra) {
    return guarded([&] {
        if (!r || !u) throw Error(GS_ERR_INVALID, "null argument");
        if (u->width == 0 || u->height == 0) throw Error(GS_ERR_INVALID, "empty framebuffer");
        HIP_CHECK(hipSetDevice(r->scene->device));
        const size_t px = static_cast<size_t>(u->width) * u->height;
        DevBuf<float> d_rgba;
        DevBuf<uint8_t> d_bgra;
        if (h_rgba) d_rgba.alloc(px * 4);
        if (h_bgra) d_bgra.alloc(px * 4);
        r->drain();
        r->enqueue(*u, h_rgba ? d_rgba.p : nullptr, h_bgra ? d_bgra.p : nullptr);
        r->drain();
        if (h_rgba) HIP_CHECK(hipMemcpy(h_rgba, d_rgba.p, px * 4 * sizeof(float), hipMemcpyDeviceToHost));
        if (h_bgra) HIP_CHECK(hipMemcpy(h_bgra, d_bgra.p, px * 4, hipMemcpyDeviceToHost));
    });
}

int gs_synchronize(gs_renderer* r) {
    return guarded([&] {
        if (!r) throw Error(GS_ERR_INVALID, "null argument");
        r->drain();
    });
}

int gs_set_timing(gs_renderer* r, int enabled) {
    return guarded([&] {
        if (!r) throw Error(GS_ERR_INVALID, "null argument");
        r->drain();
        r->timing = enabled != 0;
    });
}

int gs_set_frames_in_flight(gs_renderer* r, int frames) {
    return guarded([&] {
        if (!r) throw Error(GS_ERR_INVALID, "null argument");
        if (frames < 1 || frames > gs_renderer::kMaxInFlight) throw Error(GS_ERR_INVALID, "frames in flight must be 1..8");
        r->drain();
        r->in_flight_limit = frames;
        r->set_num_sets(frames);
    });
}

int gs_get_stats(gs_renderer* r, gs_frame_stats* out) {
    return guarded([&] {
        if (!r || !out) throw Error(GS_ERR_INVALID, "null argument");
        r->drain();
        *out = r->last;
        out->num_gaussians = r->scene->n;
        out->instance_capacity = r->capacity;
        out->retries = r->retries;
        // written by the blend itself, after it published the other counters: read from the device (everything has retired)
        out->blend_redo = out->blend_resolved = 0;
        if (r->have_frame && r->last_set && r->last_set->counters.p) {
            gs::Counters c{};
            HIP_CHECK(hipMemcpy(&c, r->last_set->counters.p, sizeof c, hipMemcpyDeviceToHost));
            out->blend_redo = c.blend_redo;
            out->blend_resolved = c.blend_resolved;
        }
    });
}

int gs_poll_stats(gs_renderer* r, gs_frame_stats* out, uint64_t* frames_retired) {
    return guarded([&] {
        if (!r || !out) throw Error(GS_ERR_INVALID, "null argument");
        HIP_CHECK(hipSetDevice(r->scene->device));
        while (r->pending > 0) {  // retire what has completed, without waiting for what has not
            const hipError_t e = hipEventQuery(r->oldest().done);
            if (e == hipErrorNotReady) break;
            HIP_CHECK(e);
            r->retire_oldest();
        }
        *out = r->last;
        out->num_gaussians = r->scene->n;
        out->instance_capacity = r->capacity;
        out->retries = r->retries;
        out->blend_redo = out->blend_resolved = 0;
        if (frames_retired) *frames_retired = r->lifetime_frames;
    });
}

int gs_get_timing_totals(gs_renderer* r, gs_frame_stats* sum, uint64_t* frames, int reset) {
    return guarded([&] {
        if (!r || !sum || !frames) throw Error(GS_ERR_INVALID, "null argument");
        r->drain();
        *sum = r->last;
        sum->ms_preprocess = static_cast<float>(r->total_ms[0]);
        sum->ms_prefix_sum = static_cast<float>(r->total_ms[1]);
        sum->ms_preprocess_sort = static_cast<float>(r->total_ms[2]);
        sum->ms_sort = static_cast<float>(r->total_ms[3]);
        sum->ms_tile_boundary = static_cast<float>(r->total_ms[4]);
        sum->ms_render = static_cast<float>(r->total_ms[5]);
        sum->ms_total = static_cast<float>(r->total_ms[6]);
        *frames = r->total_frames;
        if (reset) {
            for (double& v : r->total_ms) v = 0.0;
            r->total_frames = 0;
        }
    });
}

int gs_get_frame_intervals(gs_renderer* r, float* out_ms, uint64_t capacity, uint64_t* n_out, int reset) {
    return guarded([&] {
        if (!r || !n_out || (!out_ms && capacity)) throw Error(GS_ERR_INVALID, "null argument");
        r->drain();
        const uint64_t n = std::min<uint64_t>(capacity, r->intervals.size());
        if (n) std::memcpy(out_ms, r->intervals.data() + (r->intervals.size() - n), n * sizeof(float));
        *n_out = r->intervals.size();
        if (reset) {
            r->intervals.clear();
            r->prev_retired = false;
        }
    });
}

int gs_set_sort_path(gs_renderer* r, int mode) {
    return guarded([&] {
        if (!r) throw Error(GS_ERR_INVALID, "renderer is null");
        if (mode < 0 || mode > 2) throw Error(GS_ERR_INVALID, "sort path must be 0 (auto), 1 (global) or 2 (bin-local)");
        r->drain();
        r->sort_mode = mode;
        r->level = 0;
        r->refined = false;
        r->frames_since_fallback = 0;
    });
}

int gs_set_exp_mode(gs_renderer* r, int mode) {
    return guarded([&] {
        if (!r) throw Error(GS_ERR_INVALID, "renderer is null");
        if (mode < 0 || mode > 3)
            throw Error(GS_ERR_INVALID, "exp mode must be 0 (pipeline polynomial), 1 (hardware v_exp_f32), 2 (libm's expf in binary64) or 3 (guarded v_exp_f32)");
        r->drain();
        r->exp_mode = mode;
    });
}

int gs_set_blend_contraction(gs_renderer* r, int enabled) {
    return guarded([&] {
        if (!r) throw Error(GS_ERR_INVALID, "renderer is null");
        r->drain();
        r->contract = enabled != 0;
    });
}

int gs_set_blend_lockstep(gs_renderer* r, int mode) {
    return guarded([&] {
        if (!r) throw Error(GS_ERR_INVALID, "null argument");
        if (mode < -1 || mode > 1) throw Error(GS_ERR_INVALID, "blend lockstep: -1 automatic, 0 off, 1 on");
        r->drain();
        r->tuners.pin(mode);
    });
}

int gs_get_blend_lockstep(gs_renderer* r, int* settled) {
    int now = 0;
    const int rc = guarded([&] {
        if (!r) throw Error(GS_ERR_INVALID, "null argument");
        const BlendTuner& t = r->tuners.current();  // (the shape of the most recently enqueued frame)
        now = t.current() ? 1 : 0;
        if (settled) *settled = (t.forced >= 0 || t.phase == 3) ? 1 : 0;
    });
    return rc != 0 ? rc : now;
}

int gs_set_graph_mode(gs_renderer* r, int enabled) {
    return guarded([&] {
        if (!r) throw Error(GS_ERR_INVALID, "renderer is null");
        r->drain();
        r->graph_mode = enabled != 0;
        if (!r->graph_mode)
            for (auto& fb : r->sets) fb.drop_graph();
    });
}

int gs_debug_download(gs_renderer* r, int stage, void* dst, uint64_t bytes) {
    return guarded([&] {
        if (!r || !dst) throw Error(GS_ERR_INVALID, "null argument");
        r->drain();
        if (!r->have_frame) throw Error(GS_ERR_INVALID, "no frame rendered yet");
        if (stage == GS_STAGE_DEPTH_ORDER && !r->depth_order)
            throw Error(GS_ERR_INVALID, "the depth-order tap exists only on the global depth-order path (gs_set_sort_path(r, 1))");
        const uint64_t n = r->scene->n, v = r->last.num_visible;
        const uint64_t d = std::min<uint64_t>(r->last.num_instances, r->capacity);
        const void* src = nullptr;
        uint64_t size = 0;
        if (r->last_set->planes_stale && (stage == GS_STAGE_TILES || stage == GS_STAGE_DEPTH || stage == GS_STAGE_AABB)) {
            // the frame streamed the dense lists of visible Gaussians and wrote no per-Gaussian planes: rebuild them from the lists
            FrameBuffers& fb = *r->last_set;
            gs::launch_vis_to_planes(gs::AttrView{fb.tiles.p, fb.depth.p, fb.aabb.p, fb.rec.p, fb.vis.p, fb.vis_count.p, fb.vis_region_slots},
                                     static_cast<uint32_t>(n), fb.stream);
            HIP_CHECK(hipGetLastError());
            HIP_CHECK(hipStreamSynchronize(fb.stream));
            fb.planes_stale = false;
        }
        switch (stage) {
            case GS_STAGE_TILES: src = r->last_set->tiles.p; size = n * 4; break;
            case GS_STAGE_DEPTH: src = r->last_set->depth.p; size = n * 4; break;
            case GS_STAGE_RADIUS:
            case GS_STAGE_CONIC_OPACITY:
            case GS_STAGE_UV_RG:
            case GS_STAGE_B:
            case GS_STAGE_ALPHA_CUT:
                {   // fields of the 64-byte attribute records (only the visible Gaussians' records are written: the rest
                    // of the tap is whatever an earlier frame left, like the reference's VertexAttribute buffer)
                    const size_t width = stage == GS_STAGE_RADIUS || stage == GS_STAGE_B || stage == GS_STAGE_ALPHA_CUT ? 1 : 4;
                    if (bytes < n * width * 4) throw Error(GS_ERR_INVALID, "destination too small for stage buffer");
                    std::vector<gs::AttrRecord> recs(n);
                    if (n) HIP_CHECK(hipMemcpy(recs.data(), r->last_set->rec.p, n * sizeof(gs::AttrRecord), hipMemcpyDeviceToHost));
                    float* out = static_cast<float*>(dst);
                    for (uint64_t i = 0; i < n; ++i) {
                        const gs::AttrRecord& a = recs[i];
                        if (stage == GS_STAGE_RADIUS) out[i] = a.b_depth_r.z;
                        else if (stage == GS_STAGE_B) out[i] = a.b_depth_r.x;
                        else if (stage == GS_STAGE_ALPHA_CUT) out[i] = a.b_depth_r.w;
                        else std::memcpy(out + 4 * i, stage == GS_STAGE_CONIC_OPACITY ? &a.conic_op : &a.uv_rg, 16);
                    }
                    return;
                }
            case GS_STAGE_AABB: src = r->last_set->aabb.p; size = n * 8; break;
            case GS_STAGE_DEPTH_ORDER: src = r->depth_order; size = v * 4; break;
            case GS_STAGE_SORTED_TILE:
            case GS_STAGE_SORTED_GID:
            case GS_STAGE_RANGES:
                {   // The per-tile lists are stored bin-major (the tiles of a bin consecutive, the bins wherever their
                    // workgroup's atomic add put them); `ranges` holds each tile's (start, end) in that buffer.  The
                    // reference's buffers are the same lists laid end to end in tile order: the taps present them so.
                    const uint64_t nt = r->num_tiles;
                    std::vector<uint32_t> rg(2 * nt);
                    if (nt) HIP_CHECK(hipMemcpy(rg.data(), r->last_set->ranges.p, rg.size() * 4, hipMemcpyDeviceToHost));
                    if (stage == GS_STAGE_RANGES) {
                        if (bytes < nt * 8) throw Error(GS_ERR_INVALID, "destination too small for stage buffer");
                        uint32_t* out = static_cast<uint32_t*>(dst);
                        uint64_t pos = 0;
                        for (uint64_t t = 0; t < nt; ++t) {
                            const uint32_t len = rg[2 * t + 1] - rg[2 * t];
                            out[2 * t] = len ? static_cast<uint32_t>(pos) : 0u;  // tile_boundary.comp leaves absent tiles (0, 0)
                            out[2 * t + 1] = len ? static_cast<uint32_t>(pos + len) : 0u;
                            pos += len;
                        }
                        return;
                    }
                    if (bytes < d * 4) throw Error(GS_ERR_INVALID, "destination too small for stage buffer");
                    std::vector<uint32_t> lists;
                    if (stage == GS_STAGE_SORTED_GID) {
                        lists.resize(r->capacity);
                        HIP_CHECK(hipMemcpy(lists.data(), r->sorted_gid, lists.size() * 4, hipMemcpyDeviceToHost));
                    }
                    uint32_t* out = static_cast<uint32_t*>(dst);
                    uint64_t pos = 0;
                    for (uint64_t t = 0; t < nt; ++t)
                        for (uint64_t i = rg[2 * t]; i < rg[2 * t + 1] && pos < d; ++i)
                            out[pos++] = stage == GS_STAGE_SORTED_GID ? lists[i] : static_cast<uint32_t>(t);
                    return;
                }
            case GS_STAGE_LISTS_RAW: src = r->sorted_gid; size = d * 4; break;
            case GS_STAGE_RANGES_RAW: src = r->last_set->ranges.p; size = r->num_tiles * 8; break;
            default: throw Error(GS_ERR_INVALID, "unknown stage");
        }
        if (bytes < size) throw Error(GS_ERR_INVALID, "destination too small for stage buffer");
        if (size) HIP_CHECK(hipMemcpy(dst, src, size, hipMemcpyDeviceToHost));
    });
}

void* gs_renderer_stream(gs_renderer* r) { return r ? r->sets[0].stream : nullptr; }

}  // extern "C"
