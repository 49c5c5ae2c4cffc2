// gs_internal.h -- what the host-side translation units of libgs3d_hip.so share: the error type behind
// gs_last_error, the device-buffer owner, the scene object, and the load-time helpers.  Not installed;
// the public surface is include/gs3d_hip.h.
//
//   gs_capi.cpp      error plumbing, device selection, the entry points that own no object
//   gs_scene_host.cpp  gs_scene: load-time passes, upload, download, SH quantisation
//   gs_ply.cpp       PLY ingest (header parse, mmap, streamed upload)
//   gs_renderer.cpp  per-frame buffer sets, the frame state machine, the render entry points
//   gs_dist.cpp      multi-GPU: RCCL scene broadcast + pose sharding
#pragma once

#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>
#include <new>
#include <stdexcept>
#include <string>
#include <thread>
#include <vector>

#include "gs_host_math.h"
#include "gs_kernels.h"

namespace gs_host {

extern thread_local std::string g_last_error;  // gs_capi.cpp

struct Error : std::runtime_error {
    int code;
    Error(int c, const std::string& m) : std::runtime_error(m), code(c) {}
};

#define HIP_CHECK(expr)                                                                             \
    do {                                                                                            \
        hipError_t e_ = (expr);                                                                     \
        if (e_ != hipSuccess)                                                                       \
            throw ::gs_host::Error(GS_ERR_DEVICE, std::string(#expr) + ": " + hipGetErrorString(e_)); \
    } while (0)

template <class F>
int guarded(F&& f) {
    try {
        f();
        return GS_OK;
    } catch (const Error& e) {
        g_last_error = e.what();
        return e.code;
    } catch (const std::bad_alloc&) {
        g_last_error = "out of host memory";
        return GS_ERR_NOMEM;
    } catch (const std::exception& e) {
        g_last_error = e.what();
        return GS_ERR_INVALID;
    }
}

void select_device(int device);  // gs_capi.cpp

template <class T>
struct DevBuf {
    T* p = nullptr;
    size_t n = 0;
    DevBuf() = default;
    DevBuf(const DevBuf&) = delete;
    DevBuf& operator=(const DevBuf&) = delete;
    ~DevBuf() { release(); }
    void release() {
        if (p) (void)hipFree(p);
        p = nullptr;
        n = 0;
    }
    void alloc(size_t count) {
        release();
        if (count == 0) count = 1;
        void* q = nullptr;
        hipError_t e = hipMalloc(&q, count * sizeof(T));
        if (e != hipSuccess) throw Error(GS_ERR_NOMEM, std::string("hipMalloc: ") + hipGetErrorString(e));
        p = static_cast<T*>(q);
        n = count;
    }
    void ensure(size_t count) {
        if (count > n) alloc(count);
    }
};

constexpr uint64_t kMaxGaussians = 1ull << 31;  // ids and counts are 32-bit on the device
constexpr uint64_t kMaxInstances = (1ull << 30) - 4096;  // the per-tile lists live in one 4 GiB raw buffer

// Load-time host work (activation, AoS -> blob, PLY remapping) is embarrassingly parallel over Gaussians; the
// reference does it on one thread, one 248-byte ifstream::read per Gaussian (GSScene.cpp:36-59).
template <class F>
void parallel_for(uint64_t n, F&& body) {
    const unsigned hw = std::max(1u, std::thread::hardware_concurrency());
    const uint64_t workers = std::min<uint64_t>(std::min(16u, hw), std::max<uint64_t>(1, n / 16384));
    if (workers <= 1) {
        body(uint64_t{0}, n);
        return;
    }
    std::vector<std::thread> pool;
    for (uint64_t w = 0; w < workers; ++w)
        pool.emplace_back([&, w] { body(n * w / workers, n * (w + 1) / workers); });
    for (auto& t : pool) t.join();
}

}  // namespace gs_host

// ------------------------------------------------------------------------------------------
// gs_scene: SoA scene in HBM (59 planes) + cov3D (6 planes).  Replaces GSScene's AoS
// vertexBuffer (240 B / Gaussian) and cov3DBuffer.
// ------------------------------------------------------------------------------------------
struct gs_scene {
    int device = 0;
    uint64_t n = 0;
    gs_host::DevBuf<float> owned_blob;
    float* blob = nullptr;  // owned_blob.p or adopted
    gs_host::DevBuf<float> cov3d;
    // The per-frame kernels may read the scene from a SECOND copy in spatial (Morton) order -- a wave's 64 Gaussians neighbours
    // in space -- while ids, taps, the broadcast blob and the downloads keep the scene's own order (gs::SceneView::perm).
    // Scenes of >= GS_SPATIAL_MIN Gaussians get the copy (default 4 M, where level 1 streams the dense lists: measured A/B on one
    // box, frames bit-identical, profiles/r04_spatial_order_ab.txt: S(6e6) at 1080p +5.2 % frames/s, at 2160p +2.4 %, T(6e6) +1.8 %
    // -- the level-1 scatter 53 -> 36 us, its records leaving a block in long runs per bin; config B -8 % with three frames in
    // flight and +-0 one at a time: off there).  Costs a second scene (236 B / Gaussian) and a host-side sort at load.
    gs_host::DevBuf<float> spatial_blob;
    gs_host::DevBuf<uint32_t> perm;  // Gaussian j of the spatial copy = Gaussian perm[j] of the scene
    const float* render_blob() const { return spatial_blob.p ? spatial_blob.p : blob; }
    bool unit_opacity = true;  // no opacity exceeds 1 (the sigmoid's range): what the guarded blend's bound assumes
    gs_host::DevBuf<float> acut;     // the alpha cut of every Gaussian: render.comp:78 as a bound on `power`, from the opacity (gs::launch_alpha_cut)
    gs_host::DevBuf<uint16_t> sh16;  // gs_scene_quantize_sh: the SH block as binary16 (preprocess reads it instead)
    bool sh_half = false;

    void finish_load();  // GSScene::precomputeCov3D (GSScene.cpp:157-184) + the spatial copy + the alpha cuts
    void drop_spatial_copy();
    void make_spatial_copy();
};

namespace gs_host {

// gs_scene_host.cpp
void quantize_sh(gs_scene* s);  // gs_scene_quantize_sh; also run on the receiving ranks of a quantised scene's broadcast
void upload_vertices(gs_scene* s, const float* vertices, uint64_t n);
void activate_and_upload(gs_scene* s, const float* records, uint64_t n);

// gs_ply.cpp
std::vector<float> read_ply(const std::string& path, uint64_t* n_out);
void load_ply_streamed(gs_scene* s, const std::string& path);

}  // namespace gs_host
