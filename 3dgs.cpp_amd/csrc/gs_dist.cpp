// gs_dist.cpp -- the gs_dist_* entry points of include/gs3d_hip.h.
//
// Multi-GPU: replicate the scene, shard the poses.  The path shards by frame, so the only collective is
// one ncclBroadcast of the packed scene blob at load time -- issued here, natively, so that a C++ host of this ABI
// (the viewer, INTEGRATION.md's RendererHip.cpp) can run one process per GPU without Python.  No reference
// counterpart: the reference picks one physical device (VulkanContext.cpp:134-178).
// RCCL is dlopen'ed on first use: single-GPU consumers never load it.
#include <dlfcn.h>
#include <rccl/rccl.h>  // types only: the library is loaded on first use

#include <chrono>
#include <cstring>
#include <memory>

#include "gs_internal.h"

using namespace gs_host;

namespace {
struct Rccl {
    void* lib = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*Broadcast)(const void*, void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GetVersion)(int*) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
};
Rccl& rccl() {
    static Rccl r = [] {
        Rccl x;
        for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
            x.lib = ::dlopen(name, RTLD_NOW | RTLD_LOCAL);
            if (x.lib) break;
        }
        if (!x.lib) return x;
        x.GetUniqueId = reinterpret_cast<decltype(x.GetUniqueId)>(::dlsym(x.lib, "ncclGetUniqueId"));
        x.CommInitRank = reinterpret_cast<decltype(x.CommInitRank)>(::dlsym(x.lib, "ncclCommInitRank"));
        x.CommDestroy = reinterpret_cast<decltype(x.CommDestroy)>(::dlsym(x.lib, "ncclCommDestroy"));
        x.Broadcast = reinterpret_cast<decltype(x.Broadcast)>(::dlsym(x.lib, "ncclBroadcast"));
        x.AllReduce = reinterpret_cast<decltype(x.AllReduce)>(::dlsym(x.lib, "ncclAllReduce"));
        x.GetVersion = reinterpret_cast<decltype(x.GetVersion)>(::dlsym(x.lib, "ncclGetVersion"));
        x.GetErrorString = reinterpret_cast<decltype(x.GetErrorString)>(::dlsym(x.lib, "ncclGetErrorString"));
        return x;
    }();
    if (!r.lib || !r.GetUniqueId || !r.CommInitRank || !r.CommDestroy || !r.Broadcast || !r.AllReduce || !r.GetVersion || !r.GetErrorString)
        throw Error(GS_ERR_DEVICE, "librccl.so could not be loaded (multi-GPU entry points need RCCL)");
    return r;
}
void nccl_check(ncclResult_t e, const char* what) {
    if (e != ncclSuccess) throw Error(GS_ERR_DEVICE, std::string(what) + ": " + rccl().GetErrorString(e));
}
}  // namespace

struct gs_dist {
    ncclComm_t comm = nullptr;
    int rank = 0, world = 1, device = 0;
    hipStream_t stream = nullptr;
    double broadcast_ms = 0.0;   // wall time of the last scene broadcast on this rank (header + blob, to completion)
    uint64_t broadcast_bytes = 0;
    ~gs_dist() {
        if (comm) (void)rccl().CommDestroy(comm);
        if (stream) (void)hipStreamDestroy(stream);
    }
};

extern "C" {

int gs_dist_unique_id(uint8_t id[GS_DIST_ID_BYTES]) {
    return guarded([&] {
        if (!id) throw Error(GS_ERR_INVALID, "null argument");
        static_assert(GS_DIST_ID_BYTES == NCCL_UNIQUE_ID_BYTES, "id size");
        ncclUniqueId u;
        nccl_check(rccl().GetUniqueId(&u), "ncclGetUniqueId");
        std::memcpy(id, u.internal, GS_DIST_ID_BYTES);
    });
}

int gs_dist_create(const uint8_t id[GS_DIST_ID_BYTES], int rank, int world, int device, gs_dist** out) {
    return guarded([&] {
        if (!id || !out) throw Error(GS_ERR_INVALID, "null argument");
        if (world < 1 || rank < 0 || rank >= world) throw Error(GS_ERR_INVALID, "rank / world out of range");
        select_device(device);
        auto d = std::make_unique<gs_dist>();
        d->rank = rank;
        d->world = world;
        d->device = device;
        HIP_CHECK(hipStreamCreateWithFlags(&d->stream, hipStreamNonBlocking));
        ncclUniqueId u;
        std::memcpy(u.internal, id, GS_DIST_ID_BYTES);
        nccl_check(rccl().CommInitRank(&d->comm, world, u, rank), "ncclCommInitRank");
        *out = d.release();
    });
}

int gs_dist_rank(const gs_dist* d) { return d ? d->rank : -1; }
int gs_dist_world(const gs_dist* d) { return d ? d->world : 0; }

// pose i is rendered by rank i mod world; the k-th pose of a rank is rank + k * world
uint64_t gs_dist_pose_count(const gs_dist* d, uint64_t poses) {
    if (!d || poses <= static_cast<uint64_t>(d->rank)) return 0;
    return (poses - d->rank + d->world - 1) / d->world;
}

int gs_dist_broadcast_scene_ex(gs_dist* d, gs_scene* mine, int root, unsigned flags, gs_scene** out) {
    return guarded([&] {
        if (!d || !out) throw Error(GS_ERR_INVALID, "null argument");
        if (root < 0 || root >= d->world) throw Error(GS_ERR_INVALID, "root out of range");
        if (d->rank == root && !mine) throw Error(GS_ERR_INVALID, "the root rank must pass its scene");
        HIP_CHECK(hipSetDevice(d->device));
        const auto t_start = std::chrono::steady_clock::now();
        auto stamp = [&](uint64_t floats) {
            d->broadcast_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_start).count();
            d->broadcast_bytes = floats * sizeof(float);
        };
        // (1) a two-word header: the Gaussian count and the scene's storage flags (bit 0: SH kept as binary16 -- every replica
        // must render from the same coefficients as the root), (2) the packed blob: 11 padded SoA planes + the SH block, one message
        DevBuf<uint64_t> d_hdr;
        d_hdr.alloc(2);
        uint64_t hdr[2] = {d->rank == root ? mine->n : 0, d->rank == root && mine->sh_half ? 1ull : 0ull};
        HIP_CHECK(hipMemcpyAsync(d_hdr.p, hdr, sizeof hdr, hipMemcpyHostToDevice, d->stream));
        nccl_check(rccl().Broadcast(d_hdr.p, d_hdr.p, 2, ncclUint64, root, d->comm, d->stream), "ncclBroadcast(header)");
        HIP_CHECK(hipMemcpyAsync(hdr, d_hdr.p, sizeof hdr, hipMemcpyDeviceToHost, d->stream));
        HIP_CHECK(hipStreamSynchronize(d->stream));
        const uint64_t n = hdr[0];
        if (n >= kMaxGaussians) throw Error(GS_ERR_INVALID, "too many Gaussians (limit 2^31)");
        // the root normally keeps its own scene (in-place broadcast); with GS_DIST_COPY_ON_ROOT it receives into a fresh
        // scene like every other rank (out-of-place broadcast: send buffer = its scene), so that the caller may release or
        // keep editing the original -- and so that the receiving path can be exercised on a single GPU
        const bool receive = d->rank != root || (flags & GS_DIST_COPY_ON_ROOT) != 0;
        if (!receive) {
            nccl_check(rccl().Broadcast(mine->blob, mine->blob, gs::blob_floats(n), ncclFloat32, root, d->comm, d->stream),
                       "ncclBroadcast(scene)");
            HIP_CHECK(hipStreamSynchronize(d->stream));
            stamp(gs::blob_floats(n));
            *out = mine;
            return;
        }
        auto s = std::make_unique<gs_scene>();
        s->device = d->device;
        s->n = n;
        s->owned_blob.alloc(gs::blob_floats(n));
        s->blob = s->owned_blob.p;
        const float* send = d->rank == root ? mine->blob : s->blob;
        nccl_check(rccl().Broadcast(send, s->blob, gs::blob_floats(n), ncclFloat32, root, d->comm, d->stream),
                   "ncclBroadcast(scene)");
        HIP_CHECK(hipStreamSynchronize(d->stream));
        stamp(gs::blob_floats(n));
        s->finish_load();  // cov3D is recomputed locally: 24 B / Gaussian of arithmetic instead of 24 B over xGMI
        if (hdr[1] & 1ull) quantize_sh(s.get());  // 96 B / Gaussian of local rounding instead of 96 B over xGMI
        *out = s.release();
    });
}

int gs_dist_broadcast_scene(gs_dist* d, gs_scene* mine, int root, gs_scene** out) {
    return gs_dist_broadcast_scene_ex(d, mine, root, 0u, out);
}

int gs_dist_verify(gs_dist* d, const gs_scene* scene, gs_dist_report* out) {
    return guarded([&] {
        if (!d || !scene || !out) throw Error(GS_ERR_INVALID, "null argument");
        HIP_CHECK(hipSetDevice(d->device));
        // three words through the collective: a one from every rank (sum), this rank's checksum of its replica (min, max)
        DevBuf<uint64_t> w;
        w.alloc(4);
        gs::launch_blob_checksum(scene->blob, gs::blob_floats(scene->n), w.p + 3, d->stream);
        HIP_CHECK(hipGetLastError());
        uint64_t h[4] = {1, 0, 0, 0};
        HIP_CHECK(hipMemcpyAsync(w.p, h, sizeof(uint64_t), hipMemcpyHostToDevice, d->stream));
        HIP_CHECK(hipMemcpyAsync(w.p + 1, w.p + 3, sizeof(uint64_t), hipMemcpyDeviceToDevice, d->stream));
        HIP_CHECK(hipMemcpyAsync(w.p + 2, w.p + 3, sizeof(uint64_t), hipMemcpyDeviceToDevice, d->stream));
        nccl_check(rccl().AllReduce(w.p, w.p, 1, ncclUint64, ncclSum, d->comm, d->stream), "ncclAllReduce(ranks)");
        nccl_check(rccl().AllReduce(w.p + 1, w.p + 1, 1, ncclUint64, ncclMin, d->comm, d->stream), "ncclAllReduce(checksum min)");
        nccl_check(rccl().AllReduce(w.p + 2, w.p + 2, 1, ncclUint64, ncclMax, d->comm, d->stream), "ncclAllReduce(checksum max)");
        HIP_CHECK(hipMemcpyAsync(h, w.p, sizeof h, hipMemcpyDeviceToHost, d->stream));
        HIP_CHECK(hipStreamSynchronize(d->stream));
        int version = 0;
        nccl_check(rccl().GetVersion(&version), "ncclGetVersion");
        *out = gs_dist_report{};
        out->ranks = h[0];
        out->world = static_cast<uint64_t>(d->world);
        out->checksum = h[3];
        out->checksums_equal = h[1] == h[2] && h[1] == h[3] ? 1u : 0u;
        out->rccl_version = static_cast<uint32_t>(version);
        out->broadcast_ms = d->broadcast_ms;
        out->broadcast_bytes = d->broadcast_bytes;
    });
}

void gs_dist_destroy(gs_dist* d) { delete d; }

}  // extern "C"
