// 3dgs.h -- public facade of the MI355X splat rasterizer, API- and layout-compatible with
// shg8/3DGS.cpp's include/3dgs/3dgs.h:11-51 so that applications written against the reference
// (apps/viewer/main.cpp:56-98, the Apple bridging header :11-37) compile and link unchanged.
//
// The class keeps the reference's name.  Nothing behind it is Vulkan: `Renderer` enqueues HIP
// kernels through the C ABI in include/gs3d_hip.h, and `Window` is the (opaque here) presentation /
// input interface, implemented headless on a GPU server.
//
// ABI notes: the constructor is inline, so the member order and types of RendererConfiguration and of
// the class itself are part of the contract (the viewer aggregate-initialises the first four fields
// positionally, main.cpp:56-63).
#ifndef VULKANSPLATTING_H
#define VULKANSPLATTING_H

#include <memory>
#include <optional>
#include <string>

class Renderer;  // frame orchestrator (3dgs.cpp_amd/csrc/host/Renderer.h)
class Window;    // presentation + input interface (3dgs.cpp_amd/csrc/host/Window.h)

class VulkanSplatting {
public:
    struct RendererConfiguration {
        bool enableVulkanValidationLayers = false;  // accepted and ignored: there is no Vulkan
        std::optional<uint8_t> physicalDeviceId = std::nullopt;  // HIP device ordinal
        bool immediateSwapchain = false;            // accepted and ignored: no swapchain
        std::string scene;                          // path of the binary PLY

        float fov = 45.0f;   // like the reference, these three are not read: the camera has its own
        float near = 0.2f;   // defaults (Renderer.h:79-85)
        float far = 1000.0f;
        bool enableGui = false;  // no GUI overlay; per-pass timings go to the log / GS_METRICS_CSV

        std::shared_ptr<Window> window;
    };

    explicit VulkanSplatting(RendererConfiguration configuration) : configuration(configuration) {}

#ifdef VKGS_ENABLE_GLFW
    // Returns the headless window: fixed framebuffer size, scripted camera, GS_FRAMES ticks.
    static std::shared_ptr<Window> createGlfwWindow(std::string name, int width, int height);
#endif

#ifdef VKGS_ENABLE_METAL
    static std::shared_ptr<Window> createMetalWindow(void* caMetalLayer, int width, int height);
#endif

    void start();       // initialize + blocking frame loop until the window stops ticking
    void initialize();  // embedded-host mode: the caller drives draw()
    void draw();
    void logTranslation(float x, float y);
    void logMovement(float x, float y, float z);
    void stop();

private:
    RendererConfiguration configuration;
    std::shared_ptr<Renderer> renderer;
};

#endif  // VULKANSPLATTING_H
