// Renderer.h -- frame orchestrator behind the reference's Renderer interface (src/Renderer.h:19-168):
// Renderer(config), initialize(), draw(), run(), stop(), handleInput(), retrieveTimestamps(), and the
// public `camera` with translate().  Internals are HIP stream launches through include/gs3d_hip.h.
#pragma once
#include <atomic>
#include <chrono>
#include <map>
#include <memory>
#include <string>
#include <vector>

#include "3dgs/3dgs.h"
#include "GSScene.h"
#include "Window.h"
#include "gs_linalg.h"

struct gs_renderer;

class Renderer {
public:
    struct Camera {  // Renderer.h:40-50, defaults :79-85
        gs::vec3 position;
        gs::quat rotation;
        float fov = 45.0f;
        float nearPlane = 0.1f;
        float farPlane = 1000.0f;
        void translate(gs::vec3 translation) { position = position + rotation * translation; }
    };

    explicit Renderer(VulkanSplatting::RendererConfiguration configuration);
    ~Renderer();

    void initialize();
    void handleInput();
    void retrieveTimestamps();
    void draw();
    void run();
    void stop();

    Camera camera{};

    // last retrieveTimestamps(): pass name -> milliseconds, the reference's six spans (Renderer.cpp:484-699)
    std::map<std::string, double> metrics;
    uint64_t instances = 0;  // the "instances" text metric (Renderer.cpp:540)

private:
    VulkanSplatting::RendererConfiguration configuration;
    std::shared_ptr<Window> window;
    std::shared_ptr<GSScene> scene;
    gs_renderer* renderer = nullptr;
    std::atomic<bool> running{true};
    bool mouseCaptured = false;
    // the "swapchain": one B8G8R8A8 image per frame in flight (a swapchain has two or three; Swapchain.cpp:22-28)
    static constexpr int kMaxImages = 3;
    void* d_bgra[kMaxImages] = {nullptr, nullptr, nullptr};
    uint64_t bgraBytes = 0;
    int framesInFlight = 1;  // run(): GS_FRAMES_IN_FLIGHT (default 3) unless frames are dumped or metrics logged per frame
    std::vector<uint8_t> h_bgra;
    int fpsCounter = 0;
    std::chrono::high_resolution_clock::time_point lastFpsTime = std::chrono::high_resolution_clock::now();
    std::string metricsCsv;
    uint64_t frameIndex = 0;
};
