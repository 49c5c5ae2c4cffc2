// embedded_host_test.cpp -- drives the facade the way the reference's second consumer does (the Apple
// bridging header, apps/apple/VulkanSplatting/VulkanSplatting-Bridging-Header.h:11-37): initialize() once,
// then draw() per display tick with input pushed through logTranslation / logMovement, then stop().
// Usage: embedded_host_test scene.ply width height   (frames are dumped to $GS_DUMP_DIR as PPM)
#include <cstdio>
#include <cstdlib>
#include <exception>

#include "3dgs.h"

static VulkanSplatting* instance = nullptr;

static void vkgs_initialize(VulkanSplatting::RendererConfiguration config) {
    instance = new VulkanSplatting(config);
    instance->initialize();
}
static void vkgs_draw() { instance->draw(); }
static void vkgs_pan_translation(float x, float y) { instance->logTranslation(x, y); }
static void vkgs_movement(float x, float y, float z) { instance->logMovement(x, y, z); }
static void vkgs_cleanup() {
    instance->stop();
    delete instance;
    instance = nullptr;
}

int main(int argc, char** argv) {
    if (argc < 4) {
        std::fprintf(stderr, "usage: %s scene.ply width height\n", argv[0]);
        return 2;
    }
    try {
        VulkanSplatting::RendererConfiguration config{};
        config.scene = argv[1];
        config.window = VulkanSplatting::createGlfwWindow("embedded", std::atoi(argv[2]), std::atoi(argv[3]));
        vkgs_initialize(config);
        vkgs_draw();                       // frame 0: default camera
        vkgs_movement(0.25f, -0.5f, 1.0f);  // camera.translate
        vkgs_draw();                       // frame 1: translated
        vkgs_pan_translation(40.0f, -20.0f);
        vkgs_draw();                       // frame 2: yaw/pitch by 0.005 rad per unit
        vkgs_cleanup();
    } catch (const std::exception& e) {
        std::fprintf(stderr, "error: %s\n", e.what());
        return 1;
    }
    return 0;
}
