// harness.hpp -- binds host arrays to a shader's resources and runs its main() over the reference's dispatch grid.
// TEST INFRASTRUCTURE ONLY.  Included by the translation units oracle/build_ref.py generates, INSIDE
// `namespace glsl { namespace cs_<name> {` and after the shader text, so the names below (vertices, attr, width ...)
// are the shader's own declarations.  Grid sizes and push constants follow the reference host code cited per entry.
// One section per shader, selected by the CS_* macro the generator defines.

#if defined(CS_PRECOMP_COV3D)
// GSScene::precomputeCov3D, src/GSScene.cpp:157-184: ceil(N/256) groups, push constant scale_factor = 1.0.
extern "C" void gsr_precomp_cov3d(const void* verts, uint64_t n, float sf, float* out) {
    vertices.bind(verts, n);
    cov3ds.bind(out, 6 * n);
    scale_factor = sf;
    dispatch(uint((n + 255) / 256), 1, local_size[0], local_size[1], [] { main(); });
}

#elif defined(CS_PREPROCESS)
// Renderer::recordPreprocessCommandBuffer, src/Renderer.cpp:477-486.  params: the 160-byte std140 block
// (Renderer.h:21-29): vec4, mat4, mat4, uint, uint, float, float.
extern "C" void gsr_preprocess(const void* verts, const float* cov, uint64_t n, const void* params, void* attr_out,
                               uint32_t* tiles_out) {
    const char* p = static_cast<const char*>(params);
    std::memcpy(&camera_position, p, 16);
    std::memcpy(&proj_mat, p + 16, 64);
    std::memcpy(&view_mat, p + 80, 64);
    std::memcpy(&width, p + 144, 4);
    std::memcpy(&height, p + 148, 4);
    std::memcpy(&tan_fovx, p + 152, 4);
    std::memcpy(&tan_fovy, p + 156, 4);
    vertices.bind(verts, n);
    cov3ds.bind(cov, 6 * n);
    attr.bind(attr_out, n);
    tiles_overlap.bind(tiles_out, n);
    dispatch(uint((n + 255) / 256), 1, local_size[0], local_size[1], [] { main(); });
}

#elif defined(CS_PREFIX_SUM)
// Renderer.cpp:497-523: timestep = 0 .. ceil(log2(float(N))) inclusive, src = ping, dst = pong, a barrier between
// passes.  Returns the buffer the reference reads afterwards (iters even: ping = 0, odd: pong = 1).
extern "C" int gsr_prefix_sum(uint32_t* ping, uint32_t* pong, uint64_t n) {
    src.bind(ping, n);
    dst.bind(pong, n);
    const uint32_t iters = static_cast<uint32_t>(std::ceil(std::log2(static_cast<float>(n))));
    for (uint32_t t = 0; t <= iters; t++) {
        timestep = t;
        dispatch(uint((n + 255) / 256), 1, local_size[0], local_size[1], [] { main(); });
    }
    return iters % 2 == 0 ? 0 : 1;
}

#elif defined(CS_PREPROCESS_SORT)
// Renderer.cpp:575-587: ceil(N/256) groups, push constant tileX = ceil(width/16).
extern "C" void gsr_preprocess_sort(const void* attr_in, const uint32_t* prefix, uint64_t n, uint32_t tile_x,
                                    uint64_t* keys_out, uint32_t* payloads_out, uint64_t d) {
    attr.bind(attr_in, n);
    prefixSum.bind(prefix, n);
    keys.bind(keys_out, d);
    payloads.bind(payloads_out, d);
    tileX = tile_x;
    dispatch(uint((n + 255) / 256), 1, local_size[0], local_size[1], [] { main(); });
}

#elif defined(CS_TILE_BOUNDARY)
// Renderer.cpp:633-648: vkCmdFillBuffer(0), then ceil(D/256) groups with push constant numInstances = D.
extern "C" void gsr_tile_boundary(const uint64_t* sorted_keys, uint64_t d, uint32_t* out, uint64_t num_tiles) {
    std::memset(out, 0, num_tiles * 2 * sizeof(uint32_t));
    keys.bind(sorted_keys, d);
    boundaries.bind(out, 2 * num_tiles);
    numInstances = uint(d);
    dispatch(uint((d + 255) / 256), 1, local_size[0], local_size[1], [] { main(); });
}

#elif defined(CS_RENDER)
// Renderer.cpp:664-677: ceil(W/16) x ceil(H/16) groups of 16 x 16, push constants {width, height}.
extern "C" void gsr_render(const void* attr_in, uint64_t n, const uint32_t* bounds, uint64_t num_tiles,
                           const uint32_t* sorted, uint64_t d, uint32_t w, uint32_t h, float* rgba) {
    attr.bind(attr_in, n);
    boundaries.bind(bounds, 2 * num_tiles);
    sorted_vertices.bind(sorted, d);
    output_image.texels = rgba;
    output_image.width = int(w);
    output_image.height = int(h);
    width = w;
    height = h;
    dispatch((w + 15) / 16, (h + 15) / 16, local_size[0], local_size[1], [] { main(); });
}

#elif defined(CS_SORT_HIST)
// One histogram pass of the radix sort, Renderer.cpp:598-612: `num_workgroups` groups of 256, push constants
// {g_num_elements, g_shift, g_num_workgroups, g_num_blocks_per_workgroup} (Renderer.h:52-57).
extern "C" void gsr_radix_hist(const uint64_t* keys_in, uint32_t* hist, uint32_t num_elements, uint32_t shift, uint32_t num_workgroups,
                               uint32_t blocks_per_workgroup, uint32_t subgroup_size) {
    g_elements_in.bind(keys_in, num_elements);
    g_histograms.bind(hist, size_t(256) * num_workgroups);
    g_num_elements = num_elements;
    g_shift = shift;
    g_num_workgroups = num_workgroups;
    g_num_blocks_per_workgroup = blocks_per_workgroup;
    dispatch_workgroups(num_workgroups, local_size[0], subgroup_size, [] { main(); });
}

#elif defined(CS_SORT_SORT)
// One scatter pass of the radix sort, Renderer.cpp:616-620: the same grid and push constants as the histogram pass before it.
extern "C" void gsr_radix_scatter(const uint64_t* keys_in, uint64_t* keys_out, const uint32_t* payload_in, uint32_t* payload_out,
                                  const uint32_t* hist, uint32_t num_elements, uint32_t shift, uint32_t num_workgroups,
                                  uint32_t blocks_per_workgroup, uint32_t subgroup_size) {
    g_elements_in.bind(keys_in, num_elements);
    g_elements_out.bind(keys_out, num_elements);
    g_payload_in.bind(payload_in, num_elements);
    g_payload_out.bind(payload_out, num_elements);
    g_histograms.bind(hist, size_t(256) * num_workgroups);
    g_num_elements = num_elements;
    g_shift = shift;
    g_num_workgroups = num_workgroups;
    g_num_blocks_per_workgroup = blocks_per_workgroup;
    dispatch_workgroups(num_workgroups, local_size[0], subgroup_size, [] { main(); });
}
#endif
