/*
 * gs3d_hip.h -- C ABI of libgs3d_hip.so, the MI355X (gfx950) splat rasterizer.
 *
 * This is the drop-in boundary for the per-frame hot path of shg8/3DGS.cpp
 * (preprocess -> scan -> duplicate -> sort -> tile ranges -> alpha blend).
 * Plain pointers and sizes only; no C++/torch types.  Each entry point names
 * the reference interface it replaces (paths relative to /root/reference).
 * The reference-API C++ mirror (VulkanSplatting / Renderer / GSScene) in
 * include/3dgs/ and 3dgs.cpp_amd/csrc/host/ is a thin layer over these calls.
 *
 * All functions return 0 on success and a negative gs_status on failure;
 * gs_last_error() returns the message for the calling thread (the reference
 * throws std::runtime_error with the same text, e.g. GSScene.h:29).
 * Handles are not thread-safe.  By default a renderer has one HIP stream and one
 * frame in flight (VulkanContext.h:6 FRAMES_IN_FLIGHT 1); gs_set_frames_in_flight(k)
 * gives it k buffer sets on k streams.
 */
#ifndef GS3D_HIP_H
#define GS3D_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct gs_scene gs_scene;       /* replaces class GSScene,  src/GSScene.h:23-65   */
typedef struct gs_renderer gs_renderer; /* replaces class Renderer, src/Renderer.h:19-168 */

enum gs_status {
    GS_OK = 0,
    GS_ERR_INVALID = -1,   /* bad argument */
    GS_ERR_IO = -2,        /* file missing / malformed PLY (GSScene.h:29, GSScene.cpp:101,147) */
    GS_ERR_DEVICE = -3,    /* HIP error, or no gfx950 device: there is NO CPU fallback */
    GS_ERR_NOMEM = -4,
    GS_ERR_OVERFLOW = -5   /* instance buffers could not be grown (Renderer.cpp:541-563) */
};

/* Renderer::Camera, src/Renderer.h:40-50 (defaults :79-85). */
typedef struct {
    float position[3];
    float rotation[4]; /* quaternion w, x, y, z */
    float fov;         /* degrees, horizontal */
    float near_plane;
    float far_plane;
} gs_camera;

/* Renderer::UniformBuffer, src/Renderer.h:21-29; std140, 160 bytes, column-major. */
typedef struct {
    float camera_position[4];
    float proj_mat[16];
    float view_mat[16];
    uint32_t width;
    uint32_t height;
    float tan_fovx;
    float tan_fovy;
} gs_uniforms;

/* Timestamps of Renderer::retrieveTimestamps (Renderer.cpp:85-100): the six span
 * names registered at Renderer.cpp:484-526,580-699, plus the "instances" metric
 * (Renderer.cpp:540). */
typedef struct {
    uint64_t num_gaussians; /* N */
    uint64_t num_visible;   /* V = #Gaussians with tiles_overlap != 0 */
    uint64_t num_instances; /* D = "instances" */
    uint64_t instance_capacity;
    float ms_preprocess;
    float ms_prefix_sum;
    float ms_preprocess_sort;
    float ms_sort;
    float ms_tile_boundary;
    float ms_render;
    float ms_total;
    uint32_t retries; /* frames re-run after growing the instance buffers */
    uint32_t num_bin_entries; /* E1: (bin, Gaussian) candidates of the tile binning's first level */
    uint32_t max_bin_entries; /* candidates in the fullest bin */
    uint32_t sort_path;       /* depth-order path the frame took: 1 = global, 2 = bin-local (gs_set_sort_path) */
    uint32_t bin_tiles;       /* edge of the frame's bins in tiles (4, 8, 16 or 32) */
    uint32_t sort_level;      /* 0 .. 3: in-LDS order of up to 4096 / 8192 / 12288 / 16384 candidates per bin; 4: up to 65535 in depth slabs; 5: global path */
    uint32_t blend_redo;      /* gs_get_stats only, exp mode 3: quadrants (8 x 8 px) of the last frame abandoned by the fast pass and re-rendered with mode 2's arithmetic */
    uint32_t blend_resolved;  /* gs_get_stats only, exp mode 3: break decisions (render.comp:83) that fell inside the guard's window and were resolved by an exact replay of that pixel */
    uint32_t pad_;
} gs_frame_stats;

/* Stage taps for parity tests (the role of Buffer::download / assertEquals,
 * src/vulkan/Buffer.cpp:176-226).  Layouts are documented in DESIGN.md. */
enum gs_stage {
    GS_STAGE_TILES = 0,          /* uint32[N]  tiles_overlap                                  */
    GS_STAGE_DEPTH = 1,          /* float[N]   view-space depth (valid where tiles != 0)      */
    GS_STAGE_RADIUS = 2,         /* float[N]                                                  */
    GS_STAGE_AABB = 3,           /* uint16[4N] x0 y0 x1 y1                                    */
    GS_STAGE_CONIC_OPACITY = 4,  /* float[4N]                                                 */
    GS_STAGE_UV_RG = 5,          /* float[4N]  u v r g                                        */
    GS_STAGE_B = 6,              /* float[N]   b                                              */
    GS_STAGE_DEPTH_ORDER = 7,    /* uint32[V]  visible Gaussian ids, ascending (depth, id)    */
    GS_STAGE_ALPHA_CUT = 8,      /* float[N]   the most negative `power` at which render.comp:78 keeps an entry of this Gaussian
                                               (from its opacity, exact for libm's expf; +inf: never kept, -inf: always)  */
    GS_STAGE_SORTED_TILE = 11,   /* uint32[D]  == sorted key >> 32 of the reference (expanded from the ranges) */
    GS_STAGE_SORTED_GID = 12,    /* uint32[D]  == sorted payload of the reference             */
    GS_STAGE_RANGES = 13,        /* uint32[2T] == tileBoundaryBuffer                          */
    GS_STAGE_LISTS_RAW = 14,     /* uint32[D]  the per-tile lists exactly as they lie in HBM (bin-major)            */
    GS_STAGE_RANGES_RAW = 15     /* uint32[2T] each tile's (start, end) in GS_STAGE_LISTS_RAW; (0, 0) when empty    */
};

const char* gs_last_error(void);
int gs_device_count(int* count);

/* ---- scene: GSScene ------------------------------------------------------- */

/* GSScene::GSScene(filename) + GSScene::load (GSScene.h:25-31, GSScene.cpp:26-68):
 * parse the binary PLY, apply exp/sigmoid/normalize + SH reorder, upload as SoA,
 * run the cov3D precompute (GSScene.cpp:157-184). */
int gs_scene_load_ply(const char* path, int device, gs_scene** out);

/* Host-only halves of GSScene::load, usable without a GPU:
 *   gs_read_ply: loadPlyHeader + payload read (GSScene.cpp:99-149, :36-41); records may be NULL to
 *                query *n_out first; capacity in records.
 *   gs_activate_records: the per-record conversion (GSScene.cpp:42-55): exp(scale), sigmoid(opacity),
 *                normalize(rot), planar->interleaved SH; n x 62 floats -> n x 60 floats (GSScene::Vertex). */
int gs_read_ply(const char* path, float* records, uint64_t capacity, uint64_t* n_out);
int gs_activate_records(const float* records, uint64_t n, float* vertices);

/* Same, from n PLY-domain records (62 floats each, GSScene.cpp:17-24) in host memory. */
int gs_scene_from_records(const float* records, uint64_t n, int device, gs_scene** out);

/* From n activated GSScene::Vertex structs (60 floats each, GSScene.h:41-46) in host
 * memory: what vertexBuffer->uploadFrom(staging) transfers (GSScene.cpp:61). */
int gs_scene_from_vertices(const float* vertices, uint64_t n, int device, gs_scene** out);

/* Adopt a packed SoA blob already resident in HBM (gs_scene_blob_floats(n) floats:
 * 11 SoA planes pos[3] scale[3] rot[4] opacity[1], each padded to st = n rounded up to 16 floats, then the SH
 * block as AoS sh[n][48]; d_blob must be 64-byte aligned, which makes every plane and every Gaussian's 192-byte
 * SH block start on a 64-byte line); the caller keeps it alive.
 * This is the multi-GPU path: rank 0 builds the blob, RCCL broadcasts it, every rank
 * adopts its copy.  No reference counterpart (the reference is single-GPU). */
int gs_scene_from_device_blob(float* d_blob, uint64_t n, int device, gs_scene** out);
uint64_t gs_scene_blob_floats(uint64_t n);
int gs_scene_blob(const gs_scene* s, float** d_blob, uint64_t* floats);

/* GSScene::getNumVertices (GSScene.h:37-39). */
uint64_t gs_scene_num_vertices(const gs_scene* s);
/* Opt-in storage quantisation (no reference counterpart; the reference keeps fp32 SH, GSScene.h:41-46): the 48 SH
 * coefficients of every Gaussian are rounded to binary16 (nearest even) once, and preprocess reads 96 B instead of
 * 192 B per visible Gaussian.  It CHANGES the scene: the result equals the reference pipeline run on the rounded
 * coefficients (what the parity tests feed the oracle), not on the original ones.  gs_scene_sh_bits: 32 or 16.
 * The switch is read when a frame is enqueued: frames already in flight on the scene's renderers finish on the fp32 block
 * (which stays allocated), later ones read the binary16 block -- call gs_synchronize first if no frame may straddle it.
 * gs_dist_broadcast_scene replicates the setting. */
int gs_scene_quantize_sh(gs_scene* s);
int gs_scene_sh_bits(const gs_scene* s);
/* Vertices [first, first + count) as GSScene::Vertex (60 floats each): spot checks of scenes too large to read back whole. */
int gs_scene_download_vertex_range(const gs_scene* s, uint64_t first, uint64_t count, float* vertices);
/* Read back the activated vertices as GSScene::Vertex[n] / cov3DBuffer as float[6n]. */
int gs_scene_download_vertices(const gs_scene* s, float* vertices);
int gs_scene_download_cov3d(const gs_scene* s, float* cov3d);
void gs_scene_destroy(gs_scene* s);

/* ---- renderer: Renderer --------------------------------------------------- */

/* Renderer::initialize (Renderer.cpp:19-31) minus Vulkan/swapchain/GUI: stream,
 * per-frame buffers, the completion events.  The scene must outlive the renderer. */
int gs_renderer_create(gs_scene* scene, gs_renderer** out);
void gs_renderer_destroy(gs_renderer* r);

/* Renderer::updateUniforms (Renderer.cpp:719-754), host arithmetic only. */
int gs_camera_uniforms(const gs_camera* cam, uint32_t width, uint32_t height, gs_uniforms* out);

/* Renderer::draw (Renderer.cpp:366-426): enqueue one frame on the renderer's stream.
 * d_rgba: width*height*4 floats in HBM (RGB + alpha 1, render.comp:98) or NULL;
 * d_bgra: width*height*4 bytes B8G8R8A8_UNORM (Swapchain.cpp:22-28) or NULL.
 * Asynchronous: returns after enqueueing; gs_synchronize() or gs_get_stats() waits.
 * The instance count stays on the device (no mid-frame readback, cf. Renderer.cpp:391-399);
 * an overflow of the instance buffers is detected at the next gs_synchronize/gs_get_stats,
 * which grows them and re-runs the frame (Renderer.cpp:541-563). */
int gs_render(gs_renderer* r, const gs_uniforms* u, float* d_rgba, uint8_t* d_bgra);

/* Convenience: gs_render into host buffers, synchronous. */
int gs_render_host(gs_renderer* r, const gs_uniforms* u, float* h_rgba, uint8_t* h_bgra);

int gs_synchronize(gs_renderer* r);
/* Enable/disable the six per-pass spans of gs_frame_stats (off: the total only).  The reference writes timestamps around every
 * pass (Renderer.cpp:484-526, 580-699; QueryManager.cpp:22-41); here the kernels stamp the device's constant-rate clock at
 * their start themselves, so the spans cost nothing (round 5 recorded a hipEvent per pass: 4.5 us of idle GPU each).  A
 * span runs from the start of a pass's first kernel to the start of the next pass's. */
int gs_set_timing(gs_renderer* r, int enabled);
/* Frames that may be queued on the stream before gs_render blocks (1..8, default 1 like
 * FRAMES_IN_FLIGHT, VulkanContext.h:6).  With k > 1 the renderer keeps k sets of per-frame buffers
 * on k streams: the host enqueues frame i+1 while frame i runs, and the two frames' passes overlap
 * on the GPU.  Frames that may be in flight together must be given distinct output buffers.  An
 * overflowed frame and everything queued behind it are re-run after growing. */
int gs_set_frames_in_flight(gs_renderer* r, int frames);
/* How the per-tile lists get their depth order (same lists either way; no reference counterpart -- the reference
 * sorts all D instances, sort/hist.comp + sort/sort.comp):
 *   1  global:    the V visible Gaussians are ordered by depth first (12 small kernels), then binned;
 *   2  bin-local: Gaussians are binned in index order (bins of S x S tiles, up to 32 x 32 of them) and the workgroup
 *                 that builds a bin's tile lists first orders its candidates in LDS (6 kernels per frame);
 *                 a bin with more than 16384 candidates does not fit -> GS_ERR_OVERFLOW at the next synchronisation;
 *   0  automatic (default): bin-local, with a transparent re-run on the global path when a bin does not fit (and back
 *                 once the bins have fitted again for 32 frames).
 * The GS_STAGE_DEPTH_ORDER tap exists on path 1 only. */
int gs_set_sort_path(gs_renderer* r, int mode);
/* The blend's arithmetic (render.comp:61-98).  The reference for every mode is the shader's text read literally and
 * evaluated the way its CPU compilation (the checker the tests pin this library to) evaluates it: every product
 * and sum of :66 and :87 rounded on its own, exp() of :77 = glibc's expf (x86-64 FMA build, glibc >= 2.27).
 *
 * In EVERY mode render.comp:78's cut `alpha < 1/255` is decided exactly as the reference decides it: on `power`, against the
 * Gaussian's alpha cut (GS_STAGE_ALPHA_CUT: computed at load from the opacity with libm's expf; expf is monotone, so
 * power >= cut <=> alpha >= 1/255 bit for bit).
 *
 * gs_set_exp_mode -- exp() of render.comp:77, which GLSL leaves to the implementation (3 + 2|x| ULP):
 *   3  (default) the hardware's v_exp_f32 UNDER A GUARD: render.comp:82's break `T (1 - alpha) < 1e-4` is the one decision a
 *      fast exp can still flip (T drifts by a few ULP per blended entry).  A pixel whose T (1 - alpha) comes within a proven
 *      window of 1e-4 (gs_blend.hip: kGuard*; a few 1e-5 relative) gets the reference's decision COMPUTED: that one pixel's
 *      list is replayed with mode 2's arithmetic (about one pixel in 1300 at config B; gs_frame_stats::blend_resolved counts
 *      them).  A quadrant that would need more than eight such replays is re-rendered whole with mode 2's arithmetic
 *      (blend_redo), and a scene that holds an opacity > 1 (outside the sigmoid's range, where the bound does not apply) is
 *      blended in mode 2 altogether.  Everything else has taken exactly the reference's decisions: the frame is within
 *      ROUNDING NOISE of the reference text on any scene (<= 1e-5 asserted by the GPU tests on every configuration, measured
 *      <= 4.2e-7), with no threshold-flip pixels -- BASELINE.json's bar is 1e-4 -- at nearly the unguarded loop's speed.
 *      Not reproducible bit for bit on a CPU (v_exp_f32 is not).
 *   2  glibc's expf algorithm restated in binary64 (x 32/ln2 = k + r, 2^(k/32) from a 32-entry table, a cubic, one rounding
 *      to binary32): bit-equal to that libm on every binary32 <= 0 (checked exhaustively on the device and on the host).
 *      The frame is BIT-IDENTICAL to render.comp compiled for a CPU (tests/test_gpu_blend_modes.py, test_gpu_full_size.py);
 *      ~17 % fewer frames/s than mode 3 at config B.  Domain note: valid for opacity <= 1 and power >= -104 (no underflow branch;
 *      the alpha cut keeps smaller powers away from it for every finite opacity).
 *   0  the pipeline-defined binary32 polynomial (< 2 ULP): reproducible bit for bit on a CPU (the oracle's fast reading);
 *      unguarded, so a pixel may break one entry early or late where T (1 - alpha) sits within rounding of 1e-4;
 *   1  the hardware's v_exp_f32 without the guard.
 * With gs_set_blend_contraction(1) mode 3 runs as mode 1: the contracted `power` already differs from the reference's.
 * GS_EXP_MODE sets the initial mode for hosts that cannot call this (the viewer). */
int gs_set_exp_mode(gs_renderer* r, int mode);
/* gs_set_blend_contraction -- render.comp:66 and :87 hold three multiply-adds that GLSL lets a compiler contract into FMAs
 * (the shader has no `precise`).  0 (default): as written, one rounding per operation; 1: the three contractions (5 VALU
 * instructions per blended pair fewer).  The two differ by ULP noise on benign scenes and by up to 3e-3 on scenes of thin,
 * long splats, where `power` is a difference of much larger terms.
 * GS_BLEND_CONTRACTION sets the initial mode. */
int gs_set_blend_contraction(gs_renderer* r, int enabled);
/* Replay frames as ONE captured HIP graph each (answers VulkanContext.h:6 / Renderer.cpp:391-395, 532-717: the
 * reference re-records its render command buffer every frame because dispatch sizes depend on D; here every grid is
 * data-independent, so a frame's launches are captured once per configuration -- resolution, depth-order level,
 * capacity -- and what changes per frame, the uniforms and the output pointers, is read by the kernels from a small
 * parameter block refreshed by one copy ahead of the launch).  Per-pass spans are not recorded in this mode
 * (ms_total still is).  Default off: it lowers the host's cost per frame, not the GPU's.  GS_GRAPH=1 sets the initial mode. */
int gs_set_graph_mode(gs_renderer* r, int enabled);
/* The blend's LOCKSTEP: the four waves of a 16 x 16 tile take every 64-entry chunk of its list together (one workgroup barrier per
 * chunk), so that their gathers of the same splat records meet in the CU's L1.  It changes no pixel (the frames are bit-identical
 * either way), only where the time goes: +25 % of the blend on trained-like scenes (long lists of which an 8 x 8 quadrant keeps one
 * entry in seven: the kernel is bound by its L1 misses), -9 % on scenes whose blend is bound by the pair loop.  mode -1 (default):
 * the renderer measures the rate at which frames complete under both settings over its first 60 to 130 frames, switches lockstep
 * on only where it wins by 3 % in two passes running, and looks again every 4096 frames or when the frame's size changes;
 * 0 / 1: pinned off / on (also GS_BLEND_LOCKSTEP=0 / 1 at renderer creation). */
int gs_set_blend_lockstep(gs_renderer* r, int mode);
/* The setting the next frame will run with (0 / 1; negative: error); *settled (nullable) = 1 once the measurement has decided
 * (or the mode is pinned). */
int gs_get_blend_lockstep(gs_renderer* r, int* settled);
/* Sums of the per-pass spans over all frames retired since the last reset (ms fields are sums,
 * counts are those of the last frame); *frames = number of frames summed.  Synchronizes. */
int gs_get_timing_totals(gs_renderer* r, gs_frame_stats* sum, uint64_t* frames, int reset);
/* Completion-to-completion intervals (ms, GPU timestamps of the blend's end) of consecutive retired frames: the
 * frame time a consumer sees with frames in flight -- what the reference's FPS counter samples once a second
 * (Renderer.cpp:436-444), per frame.  Copies the most recent min(capacity, *n_out) values, oldest first; *n_out =
 * number available (at most 8192 are kept).  Synchronizes. */
int gs_get_frame_intervals(gs_renderer* r, float* out_ms, uint64_t capacity, uint64_t* n_out, int reset);
/* Renderer::retrieveTimestamps (Renderer.cpp:85-100) for the last frame; synchronizes. */
int gs_get_stats(gs_renderer* r, gs_frame_stats* out);
/* The same WITHOUT waiting, for a frame loop that keeps frames in flight (gs_set_frames_in_flight): retires the queued frames
 * that have completed (growing the buffers and re-running on overflow, like gs_synchronize) and returns the statistics of the
 * most recently retired one (all zero before the first).  *frames_retired (nullable) = frames retired since the renderer was
 * created.  blend_redo / blend_resolved are not read here (0). */
int gs_poll_stats(gs_renderer* r, gs_frame_stats* out, uint64_t* frames_retired);
/* Copy a stage buffer of the last frame to host memory; synchronizes.  The per-tile lists live bin-major in HBM;
 * GS_STAGE_SORTED_GID / _SORTED_TILE / _RANGES present them laid end to end in tile order, i.e. as the reference's
 * sorted payload, the tile half of its sorted keys and its tileBoundaryBuffer. */
int gs_debug_download(gs_renderer* r, int stage, void* dst, uint64_t bytes);
/* Test hook (tests/test_gpu_expf.py): evaluate the blend's exp() implementations ON THE DEVICE over the binary32 values with
 * bit patterns [first_bits, first_bits + count) (the negative floats run from 0x80000000 = -0 to 0xFF800000 = -inf).
 * block_sums[j] = sum over block j of 2^20 consecutive patterns of  bits(expf(x)) * ((bits(x) * 0x9E3779B1) | 1)  mod 2^64,
 * expf = the kernels' restatement of glibc's expf (exp mode 2, with libm's underflow to 0 below -103.97): the host computes
 * the same sums with its libm and compares.  guard (nullable, 4 doubles): the measured premise of exp mode 3's guard --
 * [0] max over x in [-16, 0] of |v_exp_f32(fl(x log2e)) - expf(x)| / expf(x) - E1 |x|, [1] the same ratio's max over [-1, 0],
 * [2] the guard's E0 (must exceed [0] + 2^-23), [3] its E1. */
int gs_debug_expf_scan(int device, uint32_t first_bits, uint64_t count, uint64_t* block_sums, uint64_t blocks_capacity, double* guard);
/* The hipStream_t the renderer enqueues on (for HIP-event timing by the caller). */
void* gs_renderer_stream(gs_renderer* r);

/* ---- multi-GPU: replicate the scene once, shard the camera poses (no per-frame collective) --------------------
 * One process per GPU.  Rank 0 calls gs_dist_unique_id and hands the 128 bytes to the other ranks out of band (a
 * file, a socket, MPI, torch.distributed ...); every rank calls gs_dist_create; the root loads the scene
 * (gs_scene_load_ply) and every rank calls gs_dist_broadcast_scene, which sends the Gaussian count and then the packed
 * scene blob with ONE ncclBroadcast over xGMI (RCCL, loaded on first use) and returns the rank's resident scene
 * (the root gets its own handle back; the others own a new scene with cov3D recomputed locally).  Pose i belongs
 * to rank i mod world.  No reference counterpart: the reference renders on one physical device
 * (VulkanContext.cpp:134-178, `-d`). */
#define GS_DIST_ID_BYTES 128
typedef struct gs_dist gs_dist;
int gs_dist_unique_id(uint8_t id[GS_DIST_ID_BYTES]);
int gs_dist_create(const uint8_t id[GS_DIST_ID_BYTES], int rank, int world, int device, gs_dist** out);
int gs_dist_rank(const gs_dist* d);
int gs_dist_world(const gs_dist* d);
uint64_t gs_dist_pose_count(const gs_dist* d, uint64_t poses);  /* how many of `poses` poses this rank renders */
int gs_dist_broadcast_scene(gs_dist* d, gs_scene* mine /* root only */, int root, gs_scene** out);
/* The same with flags.  The header message carries, beside the count, the scene's storage flags: a root scene quantised
 * with gs_scene_quantize_sh arrives quantised on every rank (rounded locally from the broadcast fp32 block), so that all
 * replicas render from the same coefficients.  GS_DIST_COPY_ON_ROOT: the root, too, receives into a NEW scene (out-of-place
 * broadcast from `mine`), which it owns beside `mine` -- every rank then runs the same receiving code. */
#define GS_DIST_COPY_ON_ROOT 1u
int gs_dist_broadcast_scene_ex(gs_dist* d, gs_scene* mine /* root only */, int root, unsigned flags, gs_scene** out);
/* Evidence that the collective saw every rank and that every rank holds the root's scene (collective: every rank calls it
 * after gs_dist_broadcast_scene): `ranks` = an ncclAllReduce(sum) of one per rank, `checksums_equal` = the ncclAllReduce min
 * and max of each rank's checksum of its replica (the sum of the blob's 32-bit patterns) coincide with this rank's,
 * `broadcast_ms` / `broadcast_bytes` = this rank's wall time and payload of its last scene broadcast. */
typedef struct {
    uint64_t ranks;            /* ranks that took part in the all-reduce (must equal world) */
    uint64_t world;
    uint64_t checksum;         /* of this rank's replica */
    uint64_t broadcast_bytes;
    double broadcast_ms;
    uint32_t checksums_equal;  /* 1: every rank's replica has this checksum */
    uint32_t rccl_version;     /* ncclGetVersion */
} gs_dist_report;
int gs_dist_verify(gs_dist* d, const gs_scene* scene, gs_dist_report* out);
void gs_dist_destroy(gs_dist* d);

#ifdef __cplusplus
}
#endif
#endif
