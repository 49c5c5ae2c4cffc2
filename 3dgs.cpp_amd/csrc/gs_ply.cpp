// gs_ply.cpp -- PLY ingest: header parse, the mapped payload, records by name, the streamed upload.
// Replaces GSScene::load / loadPlyHeader (GSScene.cpp:26-68, 99-149).
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <cstring>
#include <fstream>
#include <memory>
#include <sstream>

#include "gs_internal.h"

using namespace gs_host;

namespace {

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


}  // namespace

namespace gs_host {

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

}  // namespace gs_host

extern "C" {

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

}  // extern "C"
