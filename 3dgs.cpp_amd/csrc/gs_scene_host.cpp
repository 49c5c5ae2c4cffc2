// gs_scene_host.cpp -- the scene object behind the gs_scene_* entry points of include/gs3d_hip.h: upload of
// GSScene::Vertex records into the blob (11 SoA planes + the SH block), the load-time passes (cov3D, alpha cuts, the
// optional copy in spatial order), downloads and SH quantisation.  Replaces GSScene::load's buffer creation and
// GSScene::precomputeCov3D (GSScene.cpp:26-97, 157-184).
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <memory>

#include "gs_internal.h"

using namespace gs_host;

// GSScene::precomputeCov3D, GSScene.cpp:157-184
void gs_scene::finish_load() {
    cov3d.alloc(6 * n);
    if (reinterpret_cast<uintptr_t>(blob) % 64 != 0)
        throw Error(GS_ERR_INVALID, "the scene blob must be 64-byte aligned (SH blocks are read as 16-byte vectors)");
    // best effort: the copy is an optimisation (a second blob in HBM, a host-side sort); if any of its allocations fails the
    // scene simply renders from the blob as loaded (advisor, round 4)
    try {
        make_spatial_copy();
    } catch (const std::bad_alloc&) {
        drop_spatial_copy();
    } catch (const Error& e) {
        if (e.code != GS_ERR_NOMEM) throw;
        drop_spatial_copy();
    }
    gs::launch_cov3d(render_blob(), cov3d.p, static_cast<uint32_t>(n), static_cast<uint32_t>(gs::blob_stride(n)), nullptr);
    acut.alloc(n);
    DevBuf<uint32_t> beyond;
    beyond.alloc(1);
    HIP_CHECK(hipMemset(beyond.p, 0, sizeof(uint32_t)));
    gs::launch_alpha_cut(render_blob(), acut.p, static_cast<uint32_t>(n), static_cast<uint32_t>(gs::blob_stride(n)), beyond.p, nullptr);
    HIP_CHECK(hipGetLastError());
    uint32_t flag = 0;
    HIP_CHECK(hipMemcpy(&flag, beyond.p, sizeof flag, hipMemcpyDeviceToHost));  // (synchronises)
    unit_opacity = flag == 0;
}

void gs_scene::drop_spatial_copy() {
    (void)hipGetLastError();  // (a failed hipMalloc leaves its error behind)
    perm.release();
    spatial_blob.release();
    std::fprintf(stderr, "[gs3d] no memory for the scene's copy in spatial order: rendering from the blob as loaded\n");
}
// Spatial order: Morton code of the position (21 bits per axis over the scene's bounding box), ties by id.
void gs_scene::make_spatial_copy() {
    uint64_t min_n = 4ull << 20;
    if (const char* e = std::getenv("GS_SPATIAL_MIN")) min_n = std::strtoull(e, nullptr, 10);
    if (n == 0 || n < min_n) return;
    const size_t st = gs::blob_stride(n);
    std::vector<float> pos(3 * n);
    for (int k = 0; k < 3; ++k)
        HIP_CHECK(hipMemcpy(pos.data() + static_cast<size_t>(k) * n, blob + static_cast<size_t>(gs::P_POS + k) * st, n * sizeof(float), hipMemcpyDeviceToHost));
    float lo[3], hi[3];
    for (int k = 0; k < 3; ++k) {
        lo[k] = std::numeric_limits<float>::infinity();
        hi[k] = -lo[k];
        for (uint64_t i = 0; i < n; ++i) {
            const float v = pos[static_cast<size_t>(k) * n + i];
            if (std::isfinite(v)) lo[k] = std::min(lo[k], v), hi[k] = std::max(hi[k], v);
        }
        if (!(hi[k] > lo[k])) hi[k] = lo[k] + 1.0f;
    }
    auto spread = [](uint64_t v) {  // 21 bits -> every third bit
        v &= 0x1FFFFFull;
        v = (v | v << 32) & 0x1F00000000FFFFull;
        v = (v | v << 16) & 0x1F0000FF0000FFull;
        v = (v | v << 8) & 0x100F00F00F00F00Full;
        v = (v | v << 4) & 0x10C30C30C30C30C3ull;
        v = (v | v << 2) & 0x1249249249249249ull;
        return v;
    };
    std::vector<std::pair<uint64_t, uint32_t>> keyed(n);
    for (uint64_t i = 0; i < n; ++i) {
        uint64_t code = 0;
        for (int k = 0; k < 3; ++k) {
            const float v = pos[static_cast<size_t>(k) * n + i];
            const double t = std::isfinite(v) ? (static_cast<double>(v) - lo[k]) / (static_cast<double>(hi[k]) - lo[k]) : 0.0;
            code |= spread(static_cast<uint64_t>(std::min(2097151.0, std::max(0.0, t * 2097152.0)))) << k;
        }
        keyed[i] = {code, static_cast<uint32_t>(i)};
    }
    std::sort(keyed.begin(), keyed.end());
    std::vector<uint32_t> order(n);
    for (uint64_t i = 0; i < n; ++i) order[i] = keyed[i].second;
    perm.alloc(n);
    HIP_CHECK(hipMemcpy(perm.p, order.data(), n * sizeof(uint32_t), hipMemcpyHostToDevice));
    spatial_blob.alloc(gs::blob_floats(n));
    gs::launch_permute_blob(blob, perm.p, spatial_blob.p, static_cast<uint32_t>(n), static_cast<uint32_t>(st), nullptr);
    HIP_CHECK(hipGetLastError());
    HIP_CHECK(hipStreamSynchronize(nullptr));
}

namespace gs_host {

void quantize_sh(gs_scene* s) {  // gs_scene_quantize_sh; also run on the receiving ranks of a quantised scene's broadcast
    if (s->sh_half) return;
    HIP_CHECK(hipSetDevice(s->device));
    s->sh16.alloc(48 * static_cast<size_t>(s->n));
    gs::launch_sh_to_half(s->render_blob(), s->sh16.p, static_cast<uint32_t>(s->n), static_cast<uint32_t>(gs::blob_stride(s->n)), nullptr);
    HIP_CHECK(hipGetLastError());
    HIP_CHECK(hipStreamSynchronize(nullptr));
    s->sh_half = true;  // read when a frame is enqueued: frames already in flight keep reading the fp32 block, which stays
}

void upload_vertices(gs_scene* s, const float* vertices, uint64_t n) {
    // AoS GSScene::Vertex[n] -> blob: 11 SoA planes (pos3, scale3, rot4, opacity) + AoS SH block (48 per Gaussian)
    if (n >= kMaxGaussians) throw Error(GS_ERR_INVALID, "too many Gaussians (limit 2^31)");
    s->n = n;
    const size_t st = gs::blob_stride(n);
    std::vector<float> planes(gs::blob_floats(n));
    parallel_for(n, [&](uint64_t lo, uint64_t hi) {
        for (uint64_t i = lo; i < hi; ++i) {
            const float* v = vertices + i * gs::host::kVertexFloats;
            for (int k = 0; k < 3; ++k) planes[(gs::P_POS + k) * st + i] = v[k];
            for (int k = 0; k < 3; ++k) planes[(gs::P_SCALE + k) * st + i] = v[4 + k];
            for (int k = 0; k < 4; ++k) planes[(gs::P_ROT + k) * st + i] = v[8 + k];
            planes[static_cast<size_t>(gs::P_OPACITY) * st + i] = v[7];
            std::memcpy(&planes[static_cast<size_t>(gs::P_SH) * st + i * 48], v + 12, 48 * sizeof(float));
        }
    });
    s->owned_blob.alloc(planes.size());
    s->blob = s->owned_blob.p;
    if (n) HIP_CHECK(hipMemcpy(s->blob, planes.data(), planes.size() * sizeof(float), hipMemcpyHostToDevice));
    s->finish_load();
}

void activate_and_upload(gs_scene* s, const float* records, uint64_t n) {
    std::vector<float> verts(static_cast<size_t>(n) * gs::host::kVertexFloats);
    parallel_for(n, [&](uint64_t lo, uint64_t hi) {
        for (uint64_t i = lo; i < hi; ++i)
            gs::host::activate_record(records + i * gs::host::kRecordFloats, verts.data() + i * gs::host::kVertexFloats);
    });
    upload_vertices(s, verts.data(), n);
}

}  // namespace gs_host

extern "C" {

int gs_activate_records(const float* records, uint64_t n, float* vertices) {
    return guarded([&] {
        if ((!records || !vertices) && n) throw Error(GS_ERR_INVALID, "null argument");
        parallel_for(n, [&](uint64_t lo, uint64_t hi) {
            for (uint64_t i = lo; i < hi; ++i)
                gs::host::activate_record(records + i * gs::host::kRecordFloats, vertices + i * gs::host::kVertexFloats);
        });
    });
}

int gs_scene_from_records(const float* records, uint64_t n, int device, gs_scene** out) {
    return guarded([&] {
        if ((!records && n) || !out) throw Error(GS_ERR_INVALID, "null argument");
        select_device(device);
        auto s = std::make_unique<gs_scene>();
        s->device = device;
        activate_and_upload(s.get(), records, n);
        *out = s.release();
    });
}

int gs_scene_from_vertices(const float* vertices, uint64_t n, int device, gs_scene** out) {
    return guarded([&] {
        if ((!vertices && n) || !out) throw Error(GS_ERR_INVALID, "null argument");
        select_device(device);
        auto s = std::make_unique<gs_scene>();
        s->device = device;
        upload_vertices(s.get(), vertices, n);
        *out = s.release();
    });
}

uint64_t gs_scene_blob_floats(uint64_t n) { return gs::blob_floats(n); }

int gs_scene_from_device_blob(float* d_blob, uint64_t n, int device, gs_scene** out) {
    return guarded([&] {
        if ((!d_blob && n) || !out) throw Error(GS_ERR_INVALID, "null argument");
        if (n >= kMaxGaussians) throw Error(GS_ERR_INVALID, "too many Gaussians (limit 2^31)");
        select_device(device);
        auto s = std::make_unique<gs_scene>();
        s->device = device;
        s->n = n;
        s->blob = d_blob;
        s->finish_load();
        *out = s.release();
    });
}

int gs_scene_blob(const gs_scene* s, float** d_blob, uint64_t* floats) {
    return guarded([&] {
        if (!s || !d_blob || !floats) throw Error(GS_ERR_INVALID, "null argument");
        *d_blob = s->blob;
        *floats = gs_scene_blob_floats(s->n);
    });
}

uint64_t gs_scene_num_vertices(const gs_scene* s) { return s ? s->n : 0; }

int gs_scene_quantize_sh(gs_scene* s) {
    return guarded([&] {
        if (!s) throw Error(GS_ERR_INVALID, "null argument");
        quantize_sh(s);
    });
}

int gs_scene_sh_bits(const gs_scene* s) { return s ? (s->sh_half ? 16 : 32) : 0; }

int gs_scene_download_vertex_range(const gs_scene* s, uint64_t first, uint64_t count, float* vertices) {
    return guarded([&] {
        if (!s || (!vertices && count)) throw Error(GS_ERR_INVALID, "null argument");
        if (first > s->n || count > s->n - first) throw Error(GS_ERR_INVALID, "vertex range out of bounds");
        HIP_CHECK(hipSetDevice(s->device));
        const size_t st = gs::blob_stride(s->n);
        std::vector<float> plane(count), sh(48 * static_cast<size_t>(count));
        auto fetch = [&](int p, int dst_slot) {
            if (count) HIP_CHECK(hipMemcpy(plane.data(), s->blob + static_cast<size_t>(p) * st + first, count * sizeof(float), hipMemcpyDeviceToHost));
            for (uint64_t i = 0; i < count; ++i) vertices[i * gs::host::kVertexFloats + dst_slot] = plane[i];
        };
        for (int k = 0; k < 3; ++k) fetch(gs::P_POS + k, k);
        for (uint64_t i = 0; i < count; ++i) vertices[i * gs::host::kVertexFloats + 3] = 1.0f;
        for (int k = 0; k < 3; ++k) fetch(gs::P_SCALE + k, 4 + k);
        fetch(gs::P_OPACITY, 7);
        for (int k = 0; k < 4; ++k) fetch(gs::P_ROT + k, 8 + k);
        if (count) HIP_CHECK(hipMemcpy(sh.data(), s->blob + static_cast<size_t>(gs::P_SH) * st + first * 48, sh.size() * sizeof(float), hipMemcpyDeviceToHost));
        for (uint64_t i = 0; i < count; ++i) std::memcpy(vertices + i * gs::host::kVertexFloats + 12, sh.data() + i * 48, 48 * sizeof(float));
    });
}

int gs_scene_download_vertices(const gs_scene* s, float* vertices) {
    return guarded([&] {
        if (!s || (!vertices && s->n)) throw Error(GS_ERR_INVALID, "null argument");
        HIP_CHECK(hipSetDevice(s->device));
        const uint64_t n = s->n;
        const size_t st = gs::blob_stride(n);
        std::vector<float> planes(gs::blob_floats(n));
        if (n) HIP_CHECK(hipMemcpy(planes.data(), s->blob, planes.size() * sizeof(float), hipMemcpyDeviceToHost));
        for (uint64_t i = 0; i < n; ++i) {
            float* v = vertices + i * gs::host::kVertexFloats;
            for (int k = 0; k < 3; ++k) v[k] = planes[(gs::P_POS + k) * st + i];
            v[3] = 1.0f;
            for (int k = 0; k < 3; ++k) v[4 + k] = planes[(gs::P_SCALE + k) * st + i];
            v[7] = planes[static_cast<size_t>(gs::P_OPACITY) * st + i];
            for (int k = 0; k < 4; ++k) v[8 + k] = planes[(gs::P_ROT + k) * st + i];
            for (int k = 0; k < 48; ++k) v[12 + k] = planes[static_cast<size_t>(gs::P_SH) * st + i * 48 + k];
        }
    });
}

int gs_scene_download_cov3d(const gs_scene* s, float* cov3d) {
    return guarded([&] {
        if (!s || (!cov3d && s->n)) throw Error(GS_ERR_INVALID, "null argument");
        HIP_CHECK(hipSetDevice(s->device));
        const uint64_t n = s->n;
        std::vector<float> planes(6 * static_cast<size_t>(n));
        if (n) HIP_CHECK(hipMemcpy(planes.data(), s->cov3d.p, planes.size() * sizeof(float), hipMemcpyDeviceToHost));
        std::vector<uint32_t> order;  // cov3D lives in the order the frame kernels read the scene in: back to the scene's own
        if (s->perm.p) {
            order.resize(n);
            HIP_CHECK(hipMemcpy(order.data(), s->perm.p, n * sizeof(uint32_t), hipMemcpyDeviceToHost));
        }
        for (uint64_t i = 0; i < n; ++i)
            for (int k = 0; k < 6; ++k) cov3d[(order.empty() ? i : order[i]) * 6 + k] = planes[k * n + i];
    });
}

void gs_scene_destroy(gs_scene* s) { delete s; }

}  // extern "C"
