// 3dgs.h -- public facade of the MI355X splat rasterizer.
//
// API- and layout-compatible with the facade of shg8/3DGS.cpp (reference include/3dgs/3dgs.h:11-51), so that
// applications written against the reference -- apps/viewer/main.cpp:56-98, the Apple bridging header :11-37 --
// compile and link unchanged.  The class keeps the reference's name; nothing behind it is Vulkan: `Renderer`
// enqueues HIP kernels through the C ABI of include/gs3d_hip.h, and `Window` (opaque here) is the presentation /
// input interface, implemented headless on a GPU server.
//
// What is contractual: the data members of RendererConfiguration (order, types, defaults: the viewer
// aggregate-initialises the first four positionally, main.cpp:56-63, and the constructor is inline) and the two
// data members of the class.  Member functions may be listed in any order.
#ifndef GS3D_VULKANSPLATTING_FACADE_H
#define GS3D_VULKANSPLATTING_FACADE_H

#include <cstdint>
#include <memory>
#include <optional>
#include <string>
#include <utility>

class Window;    // presentation + polled input (3dgs.cpp_amd/csrc/host/Window.h)
class Renderer;  // frame orchestrator        (3dgs.cpp_amd/csrc/host/Renderer.h)

class VulkanSplatting {
public:
    struct RendererConfiguration {
        // Accepted for source compatibility and ignored: there are no Vulkan validation layers here.
        bool enableVulkanValidationLayers = false;
        // HIP device ordinal (the reference: index into the Vulkan physical devices).
        std::optional<uint8_t> physicalDeviceId = std::nullopt;
        // Accepted and ignored: there is no swapchain.
        bool immediateSwapchain = false;
        // Path of the binary little-endian PLY.
        std::string scene;

        // As in the reference these three are never read; the camera carries its own defaults (Renderer.h:79-85).
        float fov = 45.0f;
        float near = 0.2f;
        float far = 1000.0f;
        // No GUI overlay exists; per-pass timings go to the log and to $GS_METRICS_CSV.
        bool enableGui = false;

        std::shared_ptr<Window> window;
    };

    explicit VulkanSplatting(RendererConfiguration cfg) : configuration(std::move(cfg)) {}

    // Embedded-host mode: initialize() once, then draw() per display tick; input is pushed with the log* calls.
    void initialize();
    void draw();
    void logMovement(float dx, float dy, float dz);  // camera.translate in the camera frame
    void logTranslation(float dx, float dy);         // pan deltas, consumed by the next draw()

    // Stand-alone mode: initialize + blocking frame loop until the window stops ticking; stop() ends it.
    void start();
    void stop();

#ifdef VKGS_ENABLE_GLFW
    // Returns the headless window: fixed framebuffer size, $GS_FRAMES ticks, optional $GS_DUMP_DIR frame dump.
    static std::shared_ptr<Window> createGlfwWindow(std::string title, int width, int height);
#endif
#ifdef VKGS_ENABLE_METAL
    static std::shared_ptr<Window> createMetalWindow(void* caMetalLayer, int width, int height);
#endif

private:
    RendererConfiguration configuration;
    std::shared_ptr<Renderer> renderer;
};

#endif  // GS3D_VULKANSPLATTING_FACADE_H
