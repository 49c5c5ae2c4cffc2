// gs_capi.cpp -- the C ABI's error plumbing and the entry points that own no object (include/gs3d_hip.h).
// The objects live next door: gs_scene_host.cpp / gs_ply.cpp (scene, PLY ingest), gs_renderer.cpp (frames),
// gs_dist.cpp (multi-GPU); gs_internal.h is what they share.
#include "gs_internal.h"

namespace gs_host {

thread_local std::string g_last_error;

void select_device(int device) {
    int count = 0;
    hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess || count == 0)
        throw Error(GS_ERR_DEVICE, "no HIP device available (this library has no CPU fallback)");
    if (device < 0 || device >= count) throw Error(GS_ERR_INVALID, "device index out of range");
    HIP_CHECK(hipSetDevice(device));
}

}  // namespace gs_host

using namespace gs_host;

extern "C" {

const char* gs_last_error(void) { return g_last_error.c_str(); }

int gs_device_count(int* count) {
    return guarded([&] {
        if (!count) throw Error(GS_ERR_INVALID, "count is null");
        int c = 0;
        if (hipGetDeviceCount(&c) != hipSuccess) c = 0;
        *count = c;
    });
}

int gs_camera_uniforms(const gs_camera* cam, uint32_t width, uint32_t height, gs_uniforms* out) {
    return guarded([&] {
        if (!cam || !out) throw Error(GS_ERR_INVALID, "null argument");
        if (width == 0 || height == 0) throw Error(GS_ERR_INVALID, "empty framebuffer");
        gs::host::camera_uniforms(*cam, width, height, out);
    });
}

}  // extern "C"
