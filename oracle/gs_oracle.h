/*
 * gs_oracle.h -- CPU restatement of the shg8/3DGS.cpp per-frame splat pipeline.
 *
 * TEST INFRASTRUCTURE ONLY.  This is the parity oracle and the "cpu_baseline"
 * of bench.py.  Nothing under 3dgs.cpp_amd/ may include, link or call it.
 *
 * PARITY UNPINNED BY THE REFERENCE: the reference ships no tests, golden
 * images or sample scenes, and it cannot be built here (Vulkan, glslang, glm
 * are absent).  The oracle is pinned by hand-derived known-answer tests and an
 * independent numpy restatement (tests/test_oracle_*.py) instead.
 *
 * Every function cites the reference file:line it restates (paths relative to
 * /root/reference).  Matrices are column-major exactly like GLSL / glm:
 * m[c*R + r] is column c, row r.
 *
 * Floating-point contract (shared with the HIP kernels by specification, not
 * by shared code): IEEE-754 binary32, every operation rounded separately in
 * the order written in the shader (compile with -ffp-contract=off; the only
 * fused operations are the explicit fmaf() calls in gso_exp and the three
 * multiply-adds of render.comp:66,87 that GLSL permits a compiler to contract
 * -- contraction, never reassociation), IEEE
 * division and sqrt, float->int conversion truncates toward zero and
 * saturates, and exp() in render.comp is the function gso_exp() below
 * (GLSL leaves exp() precision to the driver: 3+2|x| ULP; gso_exp is < 3 ULP).
 */
#ifndef GS_ORACLE_H
#define GS_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* GSScene::Vertex, src/GSScene.h:41-46 == common.glsl:35-40.  60 floats. */
typedef struct {
    float position[4];      /* xyz, w = 1                                  */
    float scale_opacity[4]; /* exp(scale) xyz, sigmoid(opacity)            */
    float rotation[4];      /* normalised quaternion stored (w, x, y, z)   */
    float sh[48];           /* 16 RGB triples, interleaved                 */
} gso_vertex;

/* VertexAttribute, common.glsl:42-49 == Renderer.h:31-38.  64 bytes. */
typedef struct {
    float conic_opacity[4];
    float color_radii[4];
    uint32_t aabb[4];
    float uv[2];
    float depth;
    uint32_t magic;
} gso_vertex_attr;

/* Renderer::UniformBuffer, src/Renderer.h:21-29 (std140, 160 bytes). */
typedef struct {
    float camera_position[4];
    float proj_mat[16];
    float view_mat[16];
    uint32_t width;
    uint32_t height;
    float tan_fovx;
    float tan_fovy;
} gso_uniforms;

/* Renderer::Camera, src/Renderer.h:40-50 / defaults :79-85. */
typedef struct {
    float position[3];
    float rotation[4]; /* w, x, y, z (glm::quat ctor order) */
    float fov;         /* degrees, horizontal */
    float near_plane;
    float far_plane;
} gso_camera;

typedef struct {
    uint64_t num_gaussians; /* N */
    uint64_t num_visible;   /* V: tiles_overlap != 0 */
    uint64_t num_instances; /* D */
    double ms[6];           /* preprocess, prefix_sum, preprocess_sort, sort, tile_boundary, render */
} gso_stats;

/* exp() used by render.comp:77 -- the pipeline-wide definition. */
float gso_exp(float x);

/* Renderer::updateUniforms, src/Renderer.cpp:719-754. */
void gso_camera_uniforms(const gso_camera* cam, uint32_t width, uint32_t height, gso_uniforms* out);

/* GSScene::load record conversion, src/GSScene.cpp:17-24,36-59.
 * records: n x 62 floats (PLY payload), out: n gso_vertex. */
void gso_activate_records(const float* records, uint64_t n, gso_vertex* out);

/* GSScene::load + loadPlyHeader, src/GSScene.cpp:26-68,99-149.
 * Returns 0 on success; *out is malloc'ed (free with gso_free). */
int gso_load_ply(const char* path, gso_vertex** out, uint64_t* n);
void gso_free(void* p);

/* precomp_cov3d.comp:25-47 with scale_factor = 1 (GSScene.cpp:176). */
void gso_cov3d(const gso_vertex* v, uint64_t n, float* cov3d /* 6n */);

/* preprocess.comp:115-183.  attr/tiles_overlap are fully overwritten
 * (culled entries: zeroed attr, tiles_overlap 0). */
void gso_preprocess(const gso_vertex* v, const float* cov3d, uint64_t n, const gso_uniforms* u,
                    gso_vertex_attr* attr, uint32_t* tiles_overlap);

/* prefix_sum.comp:32-59 (result only: inclusive scan). */
void gso_inclusive_scan(const uint32_t* in, uint64_t n, uint32_t* out);

/* preprocess_sort.comp:31-61. */
void gso_duplicate(const gso_vertex_attr* attr, const uint32_t* prefix, uint64_t n, uint32_t tile_x,
                   uint64_t* keys, uint32_t* payload);

/* sort/hist.comp + sort/sort.comp x8 (Renderer.cpp:598-629): result only --
 * a stable ascending sort on the 64-bit key. */
void gso_sort_pairs(uint64_t* keys, uint32_t* payload, uint64_t d);

/* vkCmdFillBuffer(0) + tile_boundary.comp:22-50.  boundaries: 2*T uints. */
void gso_tile_boundary(const uint64_t* keys, uint64_t d, uint32_t* boundaries, uint64_t num_tiles);

/* render.comp:30-99.  rgba: width*height*4 floats (alpha = 1). */
void gso_render(const gso_vertex_attr* attr, const uint32_t* boundaries, const uint32_t* payload,
                uint32_t width, uint32_t height, float* rgba);

/* The same blend, eight pixels per step with AVX2 + FMA intrinsics: every lane performs gso_render's operations in
 * gso_render's order, so the two agree bit for bit (tests/test_oracle_kat.py).  It exists for the CPU BASELINE only
 * (bench.py): gso_render stays the parity checker.  gso_set_simd_blend(1) makes gso_render_frame use it. */
void gso_render_simd(const gso_vertex_attr* attr, const uint32_t* boundaries, const uint32_t* payload,
                     uint32_t width, uint32_t height, float* rgba);
void gso_set_simd_blend(int on);

/* Whole frame, Renderer::draw order (Renderer.cpp:366-426).  rgba may be
 * NULL.  Uses OpenMP threads when built with -fopenmp (gso_num_threads()).
 * Returns 0, or -1 on allocation failure. */
int gso_render_frame(const gso_vertex* v, const float* cov3d, uint64_t n, const gso_uniforms* u,
                     float* rgba, gso_stats* stats);

/* imageStore to B8G8R8A8_UNORM (Swapchain.cpp:22-28): clamp, round-to-nearest. */
void gso_pack_bgra8(const float* rgba, uint64_t pixels, uint8_t* bgra);

int gso_num_threads(void);
void gso_set_num_threads(int n);

#ifdef __cplusplus
}
#endif
#endif
