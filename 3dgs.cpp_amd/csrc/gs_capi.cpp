// gs_capi.cpp -- scene / renderer objects and the C ABI of include/gs3d_hip.h.
//
// Frame orchestration replaces Renderer::recordPreprocessCommandBuffer / recordRenderCommandBuffer /
// draw (src/Renderer.cpp:468-529, 532-717, 366-426): every pass is enqueued on one HIP stream with
// grids that do not depend on the data-dependent counts V (visible) and D (instances); the counts
// live in device memory, so the reference's mid-frame fence wait + 4-byte readback + command-buffer
// re-record (Renderer.cpp:391-399, 538) disappears.  The frame's last kernel publishes the counters to
// pinned memory only to detect instance-buffer overflow (Renderer.cpp:541-563 grows and retries too)
// or a bin that outgrew the bin-local depth order (then the frame is re-run on the global path).
#include <hip/hip_runtime.h>
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <memory>
#include <sstream>
#include <stdexcept>
#include <string>
#include <thread>
#include <vector>

#include "gs_host_math.h"
#include "gs_kernels.h"

namespace {

thread_local std::string g_last_error;

struct Error : std::runtime_error {
    int code;
    Error(int c, const std::string& m) : std::runtime_error(m), code(c) {}
};

#define HIP_CHECK(expr)                                                                             \
    do {                                                                                            \
        hipError_t e_ = (expr);                                                                     \
        if (e_ != hipSuccess)                                                                       \
            throw Error(GS_ERR_DEVICE, std::string(#expr) + ": " + hipGetErrorString(e_));          \
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

void select_device(int device) {
    int count = 0;
    hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess || count == 0)
        throw Error(GS_ERR_DEVICE, "no HIP device available (this library has no CPU fallback)");
    if (device < 0 || device >= count) throw Error(GS_ERR_INVALID, "device index out of range");
    HIP_CHECK(hipSetDevice(device));
}

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

}  // namespace

// ------------------------------------------------------------------------------------------
// gs_scene: SoA scene in HBM (59 planes) + cov3D (6 planes).  Replaces GSScene's AoS
// vertexBuffer (240 B / Gaussian) and cov3DBuffer.
// ------------------------------------------------------------------------------------------
struct gs_scene {
    int device = 0;
    uint64_t n = 0;
    DevBuf<float> owned_blob;
    float* blob = nullptr;  // owned_blob.p or adopted
    DevBuf<float> cov3d;

    void finish_load() {  // GSScene::precomputeCov3D, GSScene.cpp:157-184
        cov3d.alloc(6 * n);
        if (reinterpret_cast<uintptr_t>(blob) % 64 != 0)
            throw Error(GS_ERR_INVALID, "the scene blob must be 64-byte aligned (SH blocks are read as 16-byte vectors)");
        gs::launch_cov3d(blob, cov3d.p, static_cast<uint32_t>(n), static_cast<uint32_t>(gs::blob_stride(n)), nullptr);
        HIP_CHECK(hipGetLastError());
        HIP_CHECK(hipStreamSynchronize(nullptr));
    }
};

namespace {

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

// PLY ingest.  The reference (GSScene::loadPlyHeader, GSScene.cpp:99-149) reads only `format`, `element vertex N`
// and `end_header`, never checks property names and assumes 62 floats per vertex in the INRIA order.  That exact
// layout takes the same path here (bit-identical records).  Any other binary-little-endian layout is mapped BY
// NAME instead of being silently mis-read: properties may come in any order, extra ones are skipped, normals are
// optional, and a lower SH degree (3*K f_rest values, K = 0, 3, 8 or 15 per channel, planar) is zero-extended.
struct PlyProperty {
    std::string type, name;
    size_t offset = 0, size = 0;
};

size_t ply_type_size(const std::string& t) {
    if (t == "char" || t == "uchar" || t == "int8" || t == "uint8") return 1;
    if (t == "short" || t == "ushort" || t == "int16" || t == "uint16") return 2;
    if (t == "int" || t == "uint" || t == "float" || t == "int32" || t == "uint32" || t == "float32") return 4;
    if (t == "double" || t == "float64") return 8;
    return 0;
}

// What the header says about the payload: record count and stride, where the payload starts, and for each of the
// 62 record slots the byte offset of its source property inside a file record (-1 = absent -> 0).
struct PlyLayout {
    uint64_t n = 0;
    size_t stride = 0;
    uint64_t data_offset = 0;
    bool standard = false;  // exactly the reference's 62-float layout: records are used as they lie in the file
    std::vector<long> src;
};

PlyLayout parse_ply_header(const std::string& path) {
    std::ifstream f(path, std::ios::binary);
    if (!f.is_open()) throw Error(GS_ERR_IO, "File does not exist: " + path);
    std::string line, format;
    long long n = -1;
    bool header_end = false, in_vertex = false;
    std::vector<PlyProperty> props;
    size_t stride = 0;
    while (std::getline(f, line)) {
        std::istringstream iss(line);
        std::string token;
        iss >> token;
        if (token == "format") {
            iss >> format;
        } else if (token == "element") {
            iss >> token;
            in_vertex = token == "vertex";
            if (in_vertex) iss >> n;
        } else if (token == "property" && in_vertex) {
            PlyProperty p;
            iss >> p.type >> p.name;
            if (p.type == "list") throw Error(GS_ERR_IO, "PLY vertex element has a list property: " + path);
            p.size = ply_type_size(p.type);
            if (!p.size) throw Error(GS_ERR_IO, "PLY property '" + p.name + "' has unknown type '" + p.type + "'");
            p.offset = stride;
            stride += p.size;
            props.push_back(p);
        } else if (token == "end_header") {
            header_end = true;
            break;
        }
    }
    if (!header_end) throw Error(GS_ERR_IO, "Could not find end of header");
    if (n < 0) throw Error(GS_ERR_IO, "PLY header has no 'element vertex'");
    if (!format.empty() && format != "binary_little_endian")
        throw Error(GS_ERR_IO, "unsupported PLY format '" + format + "' (binary_little_endian only): " + path);

    PlyLayout L;
    L.n = static_cast<uint64_t>(n);
    L.data_offset = static_cast<uint64_t>(f.tellg());
    static const char* const kStandard[] = {"x", "y", "z", "nx", "ny", "nz", "f_dc_0", "f_dc_1", "f_dc_2"};
    bool standard = props.size() == gs::host::kRecordFloats || props.empty();
    for (size_t k = 0; standard && k < props.size(); ++k) {
        std::string want;
        if (k < 9) want = kStandard[k];
        else if (k < 54) want = "f_rest_" + std::to_string(k - 9);
        else if (k == 54) want = "opacity";
        else if (k < 58) want = "scale_" + std::to_string(k - 55);
        else want = "rot_" + std::to_string(k - 58);
        standard = props[k].size == 4 && props[k].name == want && (props[k].type == "float" || props[k].type == "float32");
    }
    L.standard = standard;
    if (standard) {  // the reference's layout (or a header without property lines, which the reference also accepts)
        L.stride = gs::host::kRecordFloats * sizeof(float);
        return L;
    }

    // name-mapped path: slot k of the 62-float record <- byte offset in the file's vertex record (or absent)
    L.stride = stride;
    std::vector<long>& src = L.src;
    src.assign(gs::host::kRecordFloats, -1);
    auto find = [&](const std::string& name) -> long {
        for (const auto& p : props)
            if (p.name == name) {
                if (p.size != 4 || !(p.type == "float" || p.type == "float32"))
                    throw Error(GS_ERR_IO, "PLY property '" + name + "' must be a 32-bit float");
                return static_cast<long>(p.offset);
            }
        return -1;
    };
    auto require = [&](int slot, const std::string& name) {
        src[slot] = find(name);
        if (src[slot] < 0) throw Error(GS_ERR_IO, "PLY is missing property '" + name + "': " + path);
    };
    require(0, "x");
    require(1, "y");
    require(2, "z");
    for (int k = 0; k < 3; ++k) require(6 + k, "f_dc_" + std::to_string(k));
    require(54, "opacity");
    for (int k = 0; k < 3; ++k) require(55 + k, "scale_" + std::to_string(k));
    for (int k = 0; k < 4; ++k) require(58 + k, "rot_" + std::to_string(k));
    int rest = 0;
    while (rest < 45 && find("f_rest_" + std::to_string(rest)) >= 0) ++rest;
    if (rest % 3 != 0) throw Error(GS_ERR_IO, "PLY has " + std::to_string(rest) + " f_rest properties (must be a multiple of 3)");
    const int per_channel = rest / 3;  // planar: all R, then all G, then all B
    for (int c = 0; c < 3; ++c)
        for (int j = 0; j < per_channel; ++j) src[9 + c * 15 + j] = find("f_rest_" + std::to_string(c * per_channel + j));
    return L;
}

// one file record -> the 62-float PLY-domain record
inline void ply_gather_record(const PlyLayout& L, const char* in, float* out) {
    if (L.standard) {
        std::memcpy(out, in, gs::host::kRecordFloats * sizeof(float));
        return;
    }
    for (int k = 0; k < gs::host::kRecordFloats; ++k) {
        float v = 0.0f;  // absent: normals, higher-degree SH
        if (L.src[k] >= 0) std::memcpy(&v, in + L.src[k], sizeof v);
        out[k] = v;
    }
}

// The payload of a PLY, mapped read-only (files larger than RAM are paged through; offsets are 64-bit).
struct MappedPly {
    PlyLayout layout;
    const char* base = nullptr;
    size_t length = 0;
    const char* payload = nullptr;
    explicit MappedPly(const std::string& path) : layout(parse_ply_header(path)) {
        const int fd = ::open(path.c_str(), O_RDONLY);
        if (fd < 0) throw Error(GS_ERR_IO, "File does not exist: " + path);
        struct stat st {};
        if (::fstat(fd, &st) != 0) {
            ::close(fd);
            throw Error(GS_ERR_IO, "cannot stat " + path);
        }
        length = static_cast<size_t>(st.st_size);
        const uint64_t need = layout.n * static_cast<uint64_t>(layout.stride);
        if (length < layout.data_offset || length - layout.data_offset < need) {
            ::close(fd);
            throw Error(GS_ERR_IO, std::string("PLY payload is shorter than 'element vertex' x ") +
                                       (layout.standard ? "62 floats: " : "record size: ") + path);
        }
        if (length) {
            void* m = ::mmap(nullptr, length, PROT_READ, MAP_PRIVATE, fd, 0);
            ::close(fd);
            if (m == MAP_FAILED) throw Error(GS_ERR_IO, "cannot map " + path);
            base = static_cast<const char*>(m);
            (void)::madvise(m, length, MADV_SEQUENTIAL);
        } else {
            ::close(fd);
        }
        payload = base + layout.data_offset;
    }
    ~MappedPly() {
        if (base) ::munmap(const_cast<char*>(base), length);
    }
    MappedPly(const MappedPly&) = delete;
    MappedPly& operator=(const MappedPly&) = delete;
};

std::vector<float> read_ply(const std::string& path, uint64_t* n_out) {
    MappedPly m(path);
    const PlyLayout& L = m.layout;
    std::vector<float> rec(static_cast<size_t>(L.n) * gs::host::kRecordFloats);
    *n_out = L.n;
    parallel_for(L.n, [&](uint64_t lo, uint64_t hi) {
        if (L.standard) {  // the file records ARE the records
            if (hi > lo) std::memcpy(rec.data() + lo * gs::host::kRecordFloats, m.payload + lo * L.stride, (hi - lo) * L.stride);
            return;
        }
        for (uint64_t i = lo; i < hi; ++i) ply_gather_record(L, m.payload + i * L.stride, rec.data() + i * gs::host::kRecordFloats);
    });
    return rec;
}

// GSScene::load without the host-side copies of the scene: the mapped payload is converted chunk by chunk on the
// load-time worker threads (gather by name -> activation -> the blob's planes) into two pinned staging buffers and
// streamed to HBM while the next chunk is being converted.  Host memory stays at ~120 MB whatever the file size.
void load_ply_streamed(gs_scene* s, const std::string& path) {
    MappedPly m(path);
    const PlyLayout& L = m.layout;
    const uint64_t n = L.n;
    if (n >= kMaxGaussians) throw Error(GS_ERR_INVALID, "too many Gaussians (limit 2^31)");
    s->n = n;
    s->owned_blob.alloc(gs::blob_floats(n));
    s->blob = s->owned_blob.p;
    const size_t st = gs::blob_stride(n);
    if (n) {
        constexpr uint64_t kChunk = 1ull << 18;
        const uint64_t chunk = std::min(kChunk, n);
        float* stage[2] = {nullptr, nullptr};
        hipEvent_t freed[2] = {nullptr, nullptr};
        hipStream_t up = nullptr;
        auto cleanup = [&] {
            for (int k = 0; k < 2; ++k) {
                if (stage[k]) (void)hipHostFree(stage[k]);
                if (freed[k]) (void)hipEventDestroy(freed[k]);
            }
            if (up) (void)hipStreamDestroy(up);
        };
        try {
            HIP_CHECK(hipStreamCreateWithFlags(&up, hipStreamNonBlocking));
            for (int k = 0; k < 2; ++k) {
                HIP_CHECK(hipHostMalloc(reinterpret_cast<void**>(&stage[k]), static_cast<size_t>(gs::P_COUNT) * chunk * sizeof(float),
                                        hipHostMallocDefault));
                HIP_CHECK(hipEventCreateWithFlags(&freed[k], hipEventDisableTiming));
            }
            uint64_t c = 0;
            for (uint64_t c0 = 0; c0 < n; c0 += chunk, ++c) {
                const uint64_t cnt = std::min(chunk, n - c0);
                float* buf = stage[c & 1];
                if (c >= 2) HIP_CHECK(hipEventSynchronize(freed[c & 1]));  // the upload that last read this buffer
                parallel_for(cnt, [&](uint64_t lo, uint64_t hi) {
                    float rec[gs::host::kRecordFloats], v[gs::host::kVertexFloats];
                    for (uint64_t j = lo; j < hi; ++j) {
                        ply_gather_record(L, m.payload + (c0 + j) * L.stride, rec);
                        gs::host::activate_record(rec, v);
                        for (int k = 0; k < 3; ++k) buf[(gs::P_POS + k) * chunk + j] = v[k];
                        for (int k = 0; k < 3; ++k) buf[(gs::P_SCALE + k) * chunk + j] = v[4 + k];
                        for (int k = 0; k < 4; ++k) buf[(gs::P_ROT + k) * chunk + j] = v[8 + k];
                        buf[static_cast<size_t>(gs::P_OPACITY) * chunk + j] = v[7];
                        std::memcpy(buf + static_cast<size_t>(gs::P_SH) * chunk + j * 48, v + 12, 48 * sizeof(float));
                    }
                });
                for (int p = 0; p < gs::P_SH; ++p)
                    HIP_CHECK(hipMemcpyAsync(s->blob + static_cast<size_t>(p) * st + c0, buf + static_cast<size_t>(p) * chunk,
                                             cnt * sizeof(float), hipMemcpyHostToDevice, up));
                HIP_CHECK(hipMemcpyAsync(s->blob + static_cast<size_t>(gs::P_SH) * st + c0 * 48,
                                         buf + static_cast<size_t>(gs::P_SH) * chunk, cnt * 48 * sizeof(float),
                                         hipMemcpyHostToDevice, up));
                HIP_CHECK(hipEventRecord(freed[c & 1], up));
            }
            HIP_CHECK(hipStreamSynchronize(up));
        } catch (...) {
            cleanup();
            throw;
        }
        cleanup();
    }
    s->finish_load();
}

}  // namespace

// ------------------------------------------------------------------------------------------
// gs_renderer
// ------------------------------------------------------------------------------------------
// One complete set of per-frame device buffers + the stream its passes run on.  Frames alternate between
// sets, so with >= 2 sets the small launch-bound passes of frame i+1 (scans, binning) overlap the
// VALU-bound blend of frame i on the same GPU.
struct FrameBuffers {
    hipStream_t stream = nullptr;
    // per-Gaussian attributes
    DevBuf<uint32_t> tiles;
    DevBuf<float> depth, radius, bch;
    DevBuf<ushort4> aabb;
    DevBuf<float4> conic_op, uv_rg;
    // depth sort
    DevBuf<uint32_t> dkeys[2], dvals[2], tiles_sorted, offsets;
    DevBuf<uint32_t> block_hist, digit_total, scan_partial;
    // instances
    DevBuf<uint32_t> ikeys[2], ivals[2];
    DevBuf<uint32_t> ranges;
    DevBuf<uint32_t> sorted, chunk_hist, tile_total;  // hierarchical binning
    DevBuf<gs::Counters> counters;
    bool ready = false;

    void init(size_t n, uint32_t capacity) {
        HIP_CHECK(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking));
        tiles.alloc(n);
        depth.alloc(n);
        radius.alloc(n);
        bch.alloc(n);
        aabb.alloc(n);
        conic_op.alloc(n);
        uv_rg.alloc(n);
        for (int k = 0; k < 2; ++k) {
            dkeys[k].alloc(n);
            dvals[k].alloc(n);
        }
        tiles_sorted.alloc(n);
        offsets.alloc(n);
        block_hist.alloc(256 * static_cast<size_t>(gs::kSortMaxBlocks));
        digit_total.alloc(256);
        scan_partial.alloc(2 * gs::kScanBlocks);
        counters.alloc(1);
        set_capacity(capacity);
        ready = true;
    }
    void set_capacity(uint32_t cap) {
        for (int k = 0; k < 2; ++k) {
            ikeys[k].alloc(cap);
            ivals[k].alloc(cap);
            // a frame whose candidates overflow the capacity leaves a gap of unwritten entries that the bin kernels
            // still gather through before the frame is re-run: the gap must hold valid Gaussian ids (0), never
            // whatever hipMalloc handed back
            HIP_CHECK(hipMemset(ikeys[k].p, 0, static_cast<size_t>(cap) * sizeof(uint32_t)));
            HIP_CHECK(hipMemset(ivals[k].p, 0, static_cast<size_t>(cap) * sizeof(uint32_t)));
        }
        sorted.alloc(static_cast<size_t>(cap) + 4);  // + 4: a 16-byte list store that starts inside the capacity may end past it
    }
    ~FrameBuffers() {
        if (stream) (void)hipStreamDestroy(stream);
    }
};

struct FrameSlot {
    gs_uniforms u{};
    float* rgba = nullptr;
    uint8_t* bgra = nullptr;
    hipEvent_t ev[9] = {};
    hipEvent_t done = nullptr;
    gs::Counters* h_counters = nullptr;  // pinned
    bool timed = false;
    bool bin_local = false;  // the depth-order path this frame took
};

struct gs_renderer {
    static constexpr int kMaxInFlight = 8;
    static constexpr int kSlots = kMaxInFlight + 1;  // one more than can be in flight: the previous frame's events stay readable

    gs_scene* scene = nullptr;
    bool timing = true;

    FrameBuffers sets[kMaxInFlight];
    int num_sets = 1;
    uint32_t capacity = 0;
    FrameBuffers* last_set = nullptr;  // buffers of the most recently enqueued frame (stage taps)

    // frames in flight: a ring of descriptors, all enqueued on `stream` (so device buffers are
    // reused in stream order); the host only waits when the ring is full or on gs_synchronize.
    FrameSlot slots[kSlots];
    int in_flight_limit = 1;  // the reference has FRAMES_IN_FLIGHT 1 (VulkanContext.h:6)
    uint64_t frames_enqueued = 0;
    int pending = 0;

    gs_frame_stats last{};  // stats of the most recently retired frame

    // Which depth-order path a frame takes: bin-local = one in-LDS sort per bin after the binning
    // (14 kernels per frame), global = the V visible Gaussians ordered first (26 kernels; any bin size).
    // sort_mode 0 = automatic: bin-local unless the fullest bin of a recent frame does not fit k_bin_sort.
    int sort_mode = 0;           // 0 auto, 1 global depth order, 2 bin-local (forced: a bin that does not fit is an error)
    bool bin_local_ok = true;    // automatic mode: no recent frame had a bin beyond kBinSortMax
    uint32_t frames_since_fallback = 0;
    bool use_bin_local() const { return sort_mode == 2 || (sort_mode == 0 && bin_local_ok); }
    bool have_frame = false;
    uint32_t retries = 0;        // lifetime count of re-run frames (statistics only)
    uint32_t redo_chain = 0;     // consecutive re-runs since a frame last retired cleanly: the runaway guard
    double total_ms[7] = {0, 0, 0, 0, 0, 0, 0};
    uint64_t total_frames = 0;
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
            for (auto& e : sl.ev)
                if (e) (void)hipEventDestroy(e);
            if (sl.done) (void)hipEventDestroy(sl.done);
            if (sl.h_counters) (void)hipHostFree(sl.h_counters);
        }
    }

    void set_capacity(uint32_t cap) {
        capacity = cap;
        for (auto& fb : sets)
            if (fb.ready) fb.set_capacity(cap);
    }

    void init() {
        HIP_CHECK(hipSetDevice(scene->device));
        for (auto& sl : slots) {
            // span timestamps only: no system-scope fence (L2 write-back) between the passes; `done` keeps the fence
            for (auto& e : sl.ev) HIP_CHECK(hipEventCreateWithFlags(&e, hipEventDisableSystemFence));
            HIP_CHECK(hipEventCreateWithFlags(&sl.done, hipEventDisableTiming));
            HIP_CHECK(hipHostMalloc(reinterpret_cast<void**>(&sl.h_counters), sizeof(gs::Counters), hipHostMallocDefault));
            *sl.h_counters = gs::Counters{};
        }
        uint64_t want = std::max<uint64_t>(1u << 20, 8 * static_cast<uint64_t>(scene->n));
        // test knob: start small so that the overflow / grow / re-run machinery is exercised by small scenes
        if (const char* e = std::getenv("GS_INITIAL_CAPACITY")) want = std::max<uint64_t>(256, std::strtoull(e, nullptr, 10));
        capacity = static_cast<uint32_t>(std::min<uint64_t>(want, kMaxInstances));
        sets[0].init(scene->n, capacity);
    }

    void set_num_sets(int k) {
        for (int i = 0; i < k; ++i)
            if (!sets[i].ready) sets[i].init(scene->n, capacity);
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

    void enqueue(const gs_uniforms& u, float* d_rgba, uint8_t* d_bgra) {
        HIP_CHECK(hipSetDevice(scene->device));
        FrameSlot& sl = slots[frames_enqueued % kSlots];
        FrameBuffers& fb = sets[frames_enqueued % num_sets];
        hipStream_t stream = fb.stream;
        auto &tiles = fb.tiles, &tiles_sorted = fb.tiles_sorted, &offsets = fb.offsets, &block_hist = fb.block_hist,
             &digit_total = fb.digit_total, &scan_partial = fb.scan_partial, &ranges = fb.ranges,
             &sorted = fb.sorted, &chunk_hist = fb.chunk_hist, &tile_total = fb.tile_total;
        auto &depth = fb.depth, &radius = fb.radius, &bch = fb.bch;
        auto &aabb = fb.aabb;
        auto &conic_op = fb.conic_op, &uv_rg = fb.uv_rg;
        auto &dkeys = fb.dkeys, &dvals = fb.dvals, &ikeys = fb.ikeys, &ivals = fb.ivals;
        auto &counters = fb.counters;
        last_set = &fb;
        hipEvent_t* ev = sl.ev;
        const uint32_t n = static_cast<uint32_t>(scene->n);
        const uint32_t tx = (u.width + gs::kTile - 1) / gs::kTile, ty = (u.height + gs::kTile - 1) / gs::kTile;
        if (tx > 65535 || ty > 65535) throw Error(GS_ERR_INVALID, "resolution too large (tile box is 16-bit)");
        const uint64_t nt = static_cast<uint64_t>(tx) * ty;
        // bins of S x S tiles, at most 256 of them
        int bin_shift = 3;
        while ((((tx - 1) >> bin_shift) + 1) * (((ty - 1) >> bin_shift) + 1) > 256) ++bin_shift;
        const uint32_t bins_x = ((tx - 1) >> bin_shift) + 1, bins_y = ((ty - 1) >> bin_shift) + 1;
        const uint32_t bin_tiles = 1u << (2 * bin_shift);
        if (bin_tiles > 1024) throw Error(GS_ERR_INVALID, "resolution too large for the tile binning (max 8192 x 8192)");
        const uint32_t max_chunks = capacity / 256 + 256;  // 64-candidate chunks; sized for E1 <= capacity / 4 (flagged otherwise)
        if (2 * nt > ranges.n || static_cast<size_t>(max_chunks) * bin_tiles > chunk_hist.n) {
            drain();  // resize: wait for queued frames that still use the old buffers
            ranges.ensure(2 * nt);
            tile_total.ensure(nt);
            chunk_hist.ensure(static_cast<size_t>(max_chunks) * bin_tiles);
        }
        num_tiles = nt;
        ensure_tile_order(tx, ty);

        gs::SceneView sv{scene->blob, scene->cov3d.p, n, static_cast<uint32_t>(gs::blob_stride(n))};
        gs::AttrView av{tiles.p, depth.p, radius.p, aabb.p, conic_op.p, uv_rg.p, bch.p};
        gs::Counters* cnt = counters.p;

        // the first and the last kernel of the frame clear / publish the counters themselves; the blit nodes (and
        // their fences) are only needed when one of the two is not launched
        const bool fused_counters = n != 0 && u.width != 0 && u.height != 0;
        if (!fused_counters) HIP_CHECK(hipMemsetAsync(cnt, 0, sizeof(gs::Counters), stream));
        HIP_CHECK(hipEventRecord(ev[0], stream));
        const bool bin_local = use_bin_local();
        // bin-local path: preprocess also emits the number of bins each Gaussian touches (index order)
        gs::launch_preprocess(sv, u, av, cnt, bin_local ? tiles_sorted.p : nullptr, bin_shift, stream);
        if (timing) HIP_CHECK(hipEventRecord(ev[1], stream));

        const uint32_t* cand = nullptr;  // bin-major candidate ids, (depth, id) order inside a bin
        if (!bin_local) {
            // ---- depth order of the visible Gaussians: 4 x 8-bit stable passes on bits(depth) ----
            const int blocks = std::max(1, std::min<int>(gs::kSortMaxBlocks, (n + gs::kSortTileKeys - 1) / gs::kSortTileKeys));
            const uint32_t* kin = reinterpret_cast<const uint32_t*>(depth.p);
            const uint32_t* vin = nullptr;
            for (int pass = 0; pass < 4; ++pass) {
                gs::RadixPass p{};
                const int dst = pass & 1;
                p.keys_in = kin;
                p.vals_in = vin;
                p.keys_out = dkeys[dst].p;
                p.vals_out = dvals[dst].p;
                p.n_in = &cnt->visible;
                p.n_static = n;
                p.tiles = tiles.p;
                p.n_out = &cnt->visible;
                p.block_hist = block_hist.p;
                p.digit_total = digit_total.p;
                p.shift = pass * 8;
                p.bits = 8;
                p.blocks = blocks;
                p.first = pass == 0;
                if (pass == 3) {
                    p.gather_aabb = aabb.p;  // per Gaussian: how many bins its tile box touches
                    p.bin_shift = bin_shift;
                    p.tiles_sorted = tiles_sorted.p;
                }
                gs::launch_radix_pass(p, stream);
                kin = dkeys[dst].p;
                vin = dvals[dst].p;
            }
            depth_order = dvals[1].p;
        } else {
            depth_order = nullptr;
        }
        if (timing) HIP_CHECK(hipEventRecord(ev[2], stream));

        {
            // ---- level 1: (bin, Gaussian) candidates, one stable pass by bin ----
            // global path: over the V visible Gaussians in depth order; bin-local path: over all N in index order
            // (culled ones count 0 bins), and the scan also counts the visible ones
            gs::launch_exclusive_scan(tiles_sorted.p, offsets.p, bin_local ? nullptr : &cnt->visible, n, scan_partial.p,
                                      &cnt->bin_entries, bin_local ? &cnt->visible : nullptr, stream);
            if (timing) HIP_CHECK(hipEventRecord(ev[3], stream));
            gs::launch_duplicate(bin_local ? nullptr : depth_order, offsets.p, tiles_sorted.p, aabb.p,
                                 bin_local ? nullptr : &cnt->visible, n, bins_x, bin_shift, capacity, ikeys[0].p,
                                 ivals[0].p, cnt, stream);
            if (timing) HIP_CHECK(hipEventRecord(ev[4], stream));
            {
                gs::RadixPass p{};
                p.keys_in = ikeys[0].p;
                p.vals_in = ivals[0].p;
                p.keys_out = ikeys[1].p;
                p.vals_out = ivals[1].p;
                p.n_in = &cnt->bin_entries;
                p.n_static = capacity;
                p.block_hist = block_hist.p;
                p.digit_total = digit_total.p;
                p.shift = 0;
                p.bits = 8;
                p.blocks = std::max(1, std::min<int>(gs::kSortMaxBlocks, (capacity + gs::kSortTileKeys - 1) / gs::kSortTileKeys));
                gs::launch_radix_pass(p, stream);
            }
            cand = ivals[1].p;
            if (bin_local) {  // inside each bin: index order -> (depth bits, id) order, in LDS
                gs::launch_bin_sort(digit_total.p, ivals[1].p, depth.p, ivals[0].p, cnt, bins_x * bins_y, stream);
                cand = ivals[0].p;
            }
            if (timing) HIP_CHECK(hipEventRecord(ev[5], stream));
            // ---- level 2: per-tile counts -> ranges and D (tile_boundary), then the per-tile lists ----
            gs::BinLaunch b{};
            b.cand = cand;
            b.bin_count = digit_total.p;
            b.aabb = aabb.p;
            b.chunk_hist = chunk_hist.p;
            b.tile_total = tile_total.p;
            b.ranges = ranges.p;
            b.sorted_gid = sorted.p;
            b.counters = cnt;
            b.capacity = capacity;
            b.tiles_x = tx;
            b.tiles_y = ty;
            b.bins_x = bins_x;
            b.bins = bins_x * bins_y;
            b.shift = bin_shift;
            b.max_chunks = max_chunks;
            gs::launch_bin_ranges(b, stream);
            if (timing) HIP_CHECK(hipEventRecord(ev[6], stream));
            gs::launch_bin_fill(b, stream);
            if (timing) HIP_CHECK(hipEventRecord(ev[8], stream));
            sorted_gid = sorted.p;
        }

        // ---- blend ----
        gs::launch_blend(ranges.p, sorted_gid, tile_order.p, av, u.width, u.height, d_rgba, d_bgra, cnt,
                         fused_counters ? sl.h_counters : nullptr, stream);
        HIP_CHECK(hipEventRecord(ev[7], stream));
        if (!fused_counters) HIP_CHECK(hipMemcpyAsync(sl.h_counters, cnt, sizeof(gs::Counters), hipMemcpyDeviceToHost, stream));
        HIP_CHECK(hipEventRecord(sl.done, stream));
        HIP_CHECK(hipGetLastError());

        sl.bin_local = bin_local;
        sl.u = u;
        sl.rgba = d_rgba;
        sl.bgra = d_bgra;
        sl.timed = timing;
        ++frames_enqueued;
        ++pending;
    }

    FrameSlot& oldest() { return slots[(frames_enqueued - pending) % kSlots]; }

    // Wait for the oldest queued frame; record its stats; on instance-buffer overflow grow the
    // buffers and re-run it and every frame queued behind it (Renderer.cpp:541-563 retries too).
    void retire_oldest() {
        FrameSlot& sl = oldest();
        HIP_CHECK(hipEventSynchronize(sl.done));
        if (sl.h_counters->overflow) {
            for (auto& fb : sets)
                if (fb.ready) HIP_CHECK(hipStreamSynchronize(fb.stream));
            struct Redo {
                gs_uniforms u;
                float* rgba;
                uint8_t* bgra;
            };
            std::vector<Redo> redo;
            uint64_t need = 0;
            bool grow = false, bin_too_big = false;
            for (int k = 0; k < pending; ++k) {
                FrameSlot& q = slots[(frames_enqueued - pending + k) % kSlots];
                redo.push_back({q.u, q.rgba, q.bgra});
                if (q.h_counters->overflow & 1u) {
                    grow = true;
                    // D instances, or 4 x E1 level-1 candidates (the chunk table is sized from the capacity)
                    need = std::max<uint64_t>(need, std::max<uint64_t>(q.h_counters->instances, 4ull * q.h_counters->bin_entries));
                }
                if (q.bin_local && (q.h_counters->overflow & 2u)) bin_too_big = true;
            }
            // the queued frames are dropped from the ring first: whatever is thrown below, the renderer stays usable
            frames_enqueued -= pending;
            pending = 0;
            prev_retired = false;
            if (bin_too_big) {  // a bin outgrew the in-LDS sort: take the global depth order from here on
                if (sort_mode == 2) throw Error(GS_ERR_OVERFLOW, "a bin holds more candidates than the bin-local sort can order");
                bin_local_ok = false;
                frames_since_fallback = 0;
            }
            // runaway guard: one frame may need a path fall-back and a few grow steps (each grow is sized from the counts
            // the overflowing frame reported, so it converges at once unless the chunk table and the lists take turns)
            if (++redo_chain > 8) throw Error(GS_ERR_OVERFLOW, "instance buffers overflowed repeatedly");
            if (grow) {
                need = need + need / 2 + 4096;  // 1.5x head-room: a moving camera should not re-grow every few frames
                if (need > kMaxInstances) throw Error(GS_ERR_OVERFLOW, "more than 2^30 tile instances");
                set_capacity(static_cast<uint32_t>(need));
            }
            ++retries;
            for (const Redo& f : redo) enqueue(f.u, f.rgba, f.bgra);
            return;
        }
        redo_chain = 0;
        if (sort_mode == 0 && !bin_local_ok) {  // back to the bin-local path once the bins have fitted for a while
            if (sl.h_counters->max_bin <= static_cast<uint32_t>(gs::kBinSortMax) * 7 / 8) {
                if (++frames_since_fallback >= 32) bin_local_ok = true;
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
        st.sort_path = sl.bin_local ? 2u : 1u;
        st.instance_capacity = capacity;
        auto span = [&](int a, int b) {
            float ms = 0.0f;
            HIP_CHECK(hipEventElapsedTime(&ms, sl.ev[a], sl.ev[b]));
            return ms;
        };
        st.ms_total = span(0, 7);
        if (sl.timed) {
            st.ms_preprocess = span(0, 1);
            st.ms_sort = span(1, 2) + span(4, 5) + span(6, 8);
            st.ms_prefix_sum = span(2, 3);
            st.ms_preprocess_sort = span(3, 4);
            st.ms_tile_boundary = span(5, 6);
            st.ms_render = span(8, 7);
        }
        st.retries = retries;
        last = st;
        have_frame = true;
        const float v[7] = {st.ms_preprocess, st.ms_prefix_sum, st.ms_preprocess_sort, st.ms_sort,
                            st.ms_tile_boundary, st.ms_render, st.ms_total};
        for (int k = 0; k < 7; ++k) total_ms[k] += v[k];
        ++total_frames;
        {   // frames on different streams may finish out of order: measure against the latest completion so far
            const uint64_t idx = frames_enqueued - pending;  // this frame
            if (prev_retired) {
                float dt = 0.0f;
                if (hipEventElapsedTime(&dt, slots[latest_done % kSlots].ev[7], sl.ev[7]) == hipSuccess) {
                    if (intervals.size() >= kIntervalRing) intervals.erase(intervals.begin(), intervals.begin() + kIntervalRing / 2);
                    intervals.push_back(dt > 0.0f ? dt : 0.0f);  // 0: it had already finished when its predecessor did
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

const char* gs_last_error(void) { return g_last_error.c_str(); }

int gs_device_count(int* count) {
    return guarded([&] {
        if (!count) throw Error(GS_ERR_INVALID, "count is null");
        int c = 0;
        if (hipGetDeviceCount(&c) != hipSuccess) c = 0;
        *count = c;
    });
}

int gs_activate_records(const float* records, uint64_t n, float* vertices) {
    return guarded([&] {
        if ((!records || !vertices) && n) throw Error(GS_ERR_INVALID, "null argument");
        parallel_for(n, [&](uint64_t lo, uint64_t hi) {
            for (uint64_t i = lo; i < hi; ++i)
                gs::host::activate_record(records + i * gs::host::kRecordFloats, vertices + i * gs::host::kVertexFloats);
        });
    });
}

int gs_read_ply(const char* path, float* records, uint64_t capacity, uint64_t* n_out) {
    return guarded([&] {
        if (!path || !n_out) throw Error(GS_ERR_INVALID, "null argument");
        uint64_t n = 0;
        std::vector<float> rec = read_ply(path, &n);
        *n_out = n;
        if (records) {
            if (capacity < n) throw Error(GS_ERR_INVALID, "record buffer too small");
            std::memcpy(records, rec.data(), rec.size() * sizeof(float));
        }
    });
}

int gs_scene_load_ply(const char* path, int device, gs_scene** out) {
    return guarded([&] {
        if (!path || !out) throw Error(GS_ERR_INVALID, "null argument");
        (void)parse_ply_header(path);  // IO / format errors first, like GSScene's ctor, before any device is touched
        select_device(device);
        auto s = std::make_unique<gs_scene>();
        s->device = device;
        load_ply_streamed(s.get(), path);
        *out = s.release();
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
        for (uint64_t i = 0; i < n; ++i)
            for (int k = 0; k < 6; ++k) cov3d[i * 6 + k] = planes[k * n + i];
    });
}

void gs_scene_destroy(gs_scene* s) { delete s; }

int gs_renderer_create(gs_scene* scene, gs_renderer** out) {
    return guarded([&] {
        if (!scene || !out) throw Error(GS_ERR_INVALID, "null argument");
        auto r = std::make_unique<gs_renderer>();
        r->scene = scene;
        r->init();
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
            if (fb.ready) (void)hipStreamSynchronize(fb.stream);
    delete r;
}

int gs_camera_uniforms(const gs_camera* cam, uint32_t width, uint32_t height, gs_uniforms* out) {
    return guarded([&] {
        if (!cam || !out) throw Error(GS_ERR_INVALID, "null argument");
        if (width == 0 || height == 0) throw Error(GS_ERR_INVALID, "empty framebuffer");
        gs::host::camera_uniforms(*cam, width, height, out);
    });
}

int gs_render(gs_renderer* r, const gs_uniforms* u, float* d_rgba, uint8_t* d_bgra) {
    return guarded([&] {
        if (!r || !u) throw Error(GS_ERR_INVALID, "null argument");
        if (u->width == 0 || u->height == 0) throw Error(GS_ERR_INVALID, "empty framebuffer");
        r->make_room();  // at most in_flight_limit frames queued; resolves pending overflows first
        r->enqueue(*u, d_rgba, d_bgra);
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
        r->bin_local_ok = true;
        r->frames_since_fallback = 0;
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
        switch (stage) {
            case GS_STAGE_TILES: src = r->last_set->tiles.p; size = n * 4; break;
            case GS_STAGE_DEPTH: src = r->last_set->depth.p; size = n * 4; break;
            case GS_STAGE_RADIUS: src = r->last_set->radius.p; size = n * 4; break;
            case GS_STAGE_AABB: src = r->last_set->aabb.p; size = n * 8; break;
            case GS_STAGE_CONIC_OPACITY: src = r->last_set->conic_op.p; size = n * 16; break;
            case GS_STAGE_UV_RG: src = r->last_set->uv_rg.p; size = n * 16; break;
            case GS_STAGE_B: src = r->last_set->bch.p; size = n * 4; break;
            case GS_STAGE_DEPTH_ORDER: src = r->depth_order; size = v * 4; break;
            case GS_STAGE_SORTED_TILE:
                {   // the tile id of list position i follows from the ranges
                    if (bytes < d * 4) throw Error(GS_ERR_INVALID, "destination too small for stage buffer");
                    std::vector<uint32_t> rg(2 * r->num_tiles);
                    if (!rg.empty()) HIP_CHECK(hipMemcpy(rg.data(), r->last_set->ranges.p, rg.size() * 4, hipMemcpyDeviceToHost));
                    uint32_t* out = static_cast<uint32_t*>(dst);
                    for (uint64_t t = 0; t < r->num_tiles; ++t)
                        for (uint64_t i = rg[2 * t]; i < std::min<uint64_t>(rg[2 * t + 1], d); ++i) out[i] = static_cast<uint32_t>(t);
                    return;
                }
            case GS_STAGE_SORTED_GID: src = r->sorted_gid; size = d * 4; break;
            case GS_STAGE_RANGES: src = r->last_set->ranges.p; size = r->num_tiles * 8; break;
            default: throw Error(GS_ERR_INVALID, "unknown stage");
        }
        if (bytes < size) throw Error(GS_ERR_INVALID, "destination too small for stage buffer");
        if (size) HIP_CHECK(hipMemcpy(dst, src, size, hipMemcpyDeviceToHost));
    });
}

void* gs_renderer_stream(gs_renderer* r) { return r ? r->sets[0].stream : nullptr; }

}  // extern "C"
