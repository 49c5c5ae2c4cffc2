// GSScene.h -- scene ingest behind the reference's GSScene interface (src/GSScene.h:23-65): same
// constructor contract (throws std::runtime_error("File does not exist: ...")), load(), getNumVertices(),
// the Vertex / Cov3DUpperRight records and the two public device buffers.  Storage is the SoA blob of
// libgs3d_hip (59 planes) instead of the AoS 240-byte Vertex array; `DeviceBuffer` says where it lives.
#pragma once
#include <cstdint>
#include <memory>
#include <string>

struct gs_scene;

struct DeviceBuffer {  // stands in for the reference's Buffer (a device allocation the renderer binds)
    const void* ptr = nullptr;
    uint64_t bytes = 0;
};

class GSScene {
public:
    explicit GSScene(const std::string& filename);
    ~GSScene();
    GSScene(const GSScene&) = delete;
    GSScene& operator=(const GSScene&) = delete;

    // GSScene::load(context): the "context" is a HIP device ordinal here.
    void load(int device);
    // One fixed test Gaussian (the reference's loadTestScene is unseeded random, GSScene.cpp:70-97).
    void loadTestScene(int device);

    uint64_t getNumVertices() const;

    struct Vertex {  // GSScene.h:41-46
        float position[4];
        float scale_opacity[4];
        float rotation[4];
        float shs[48];
    };
    struct Cov3DUpperRight {
        float mat[6];
    };

    std::shared_ptr<DeviceBuffer> vertexBuffer;  // packed SoA blob (59 planes of n floats)
    std::shared_ptr<DeviceBuffer> cov3DBuffer;   // 6 planes of n floats

    gs_scene* handle() const { return scene; }

private:
    std::string filename;
    gs_scene* scene = nullptr;
    void publishBuffers();
};
