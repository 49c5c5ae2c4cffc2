// host.cpp -- VulkanSplatting / Renderer / GSScene / HeadlessWindow over the C ABI of libgs3d_hip.
// Control flow follows src/3dgs.cpp:6-44, src/Renderer.cpp:19-31,33-83,85-100,366-450 and
// src/GSScene.cpp:26-68; errors are std::runtime_error with the reference's messages.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <filesystem>
#include <fstream>
#include <stdexcept>

#include "../../../include/gs3d_hip.h"
#include "Renderer.h"

namespace {
void check(int rc) {
    if (rc != GS_OK) throw std::runtime_error(gs_last_error());
}
const char* env_or(const char* k, const char* dflt) {
    const char* v = std::getenv(k);
    return v ? v : dflt;
}
}  // namespace

// ---------------------------------------------------------------- HeadlessWindow
HeadlessWindow::HeadlessWindow(std::string name_, int w, int h)
    : name(std::move(name_)), width(static_cast<uint32_t>(w)), height(static_cast<uint32_t>(h)) {
    frameBudget = std::atoll(env_or("GS_FRAMES", "1"));
    dumpDir = env_or("GS_DUMP_DIR", "");
    const std::string path = env_or("GS_CAMERA_PATH", "");
    if (!path.empty()) {
        std::ifstream f(path);
        if (!f) throw std::runtime_error("File does not exist: " + path);
        double dx, dy;
        std::string k;
        while (f >> dx >> dy >> k) {
            ScriptedInput in{dx, dy, {}};
            const char* names = "WASD_^x";
            for (char c : k)
                for (int i = 0; i < 7; ++i)
                    if (c == names[i]) in.keys[i] = true;
            script.push_back(in);
        }
    }
}
bool HeadlessWindow::tick() {
    if (frameBudget > 0 && ticks >= frameBudget) return false;
    keys = {};
    buttonDown = false;  // polled state, like glfwGetMouseButton: held only while this tick's input says so
    if (static_cast<size_t>(ticks) < script.size()) {  // this tick's polled input
        const ScriptedInput& in = script[static_cast<size_t>(ticks)];
        keys = in.keys;
        if (in.dx != 0.0 || in.dy != 0.0) {
            logTranslation(static_cast<float>(in.dx), static_cast<float>(in.dy));
            buttonDown = true;  // a scripted drag: button 0 is down on the ticks that carry a cursor delta
        }
    }
    ++ticks;
    return true;
}
std::array<double, 2> HeadlessWindow::getCursorTranslation() {
    const std::array<double, 2> r{accumulatedX, accumulatedY};
    accumulatedX = accumulatedY = 0;
    return r;
}
void HeadlessWindow::logTranslation(float x, float y) {
    accumulatedX += x;
    accumulatedY += y;  // windowing/MetalWindow.cpp:41-45: deltas only, no button
}
void HeadlessWindow::present(const uint8_t* bgra, uint32_t w, uint32_t h) {
    if (!dumpDir.empty()) {
        char path[4096];
        std::snprintf(path, sizeof path, "%s/frame_%05llu.ppm", dumpDir.c_str(), static_cast<unsigned long long>(framesPresented));
        std::ofstream f(path, std::ios::binary);
        f << "P6\n" << w << " " << h << "\n255\n";
        std::vector<uint8_t> rgb(static_cast<size_t>(w) * h * 3);
        for (size_t i = 0; i < static_cast<size_t>(w) * h; ++i) {
            rgb[3 * i + 0] = bgra[4 * i + 2];
            rgb[3 * i + 1] = bgra[4 * i + 1];
            rgb[3 * i + 2] = bgra[4 * i + 0];
        }
        f.write(reinterpret_cast<const char*>(rgb.data()), static_cast<std::streamsize>(rgb.size()));
    }
    ++framesPresented;
}

// ---------------------------------------------------------------- GSScene
GSScene::GSScene(const std::string& filename_) : filename(filename_) {
    if (!std::filesystem::exists(filename)) throw std::runtime_error("File does not exist: " + filename);
}
GSScene::~GSScene() { gs_scene_destroy(scene); }
void GSScene::publishBuffers() {
    float* blob = nullptr;
    uint64_t floats = 0;
    check(gs_scene_blob(scene, &blob, &floats));
    vertexBuffer = std::make_shared<DeviceBuffer>(DeviceBuffer{blob, floats * sizeof(float)});
    cov3DBuffer = std::make_shared<DeviceBuffer>(DeviceBuffer{nullptr, gs_scene_num_vertices(scene) * 6 * sizeof(float)});
}
void GSScene::load(int device) {
    const auto t0 = std::chrono::high_resolution_clock::now();
    check(gs_scene_load_ply(filename.c_str(), device, &scene));
    publishBuffers();
    const auto ms = std::chrono::duration_cast<std::chrono::milliseconds>(std::chrono::high_resolution_clock::now() - t0).count();
    std::fprintf(stderr, "[info] Loaded %s in %lldms (%llu Gaussians)\n", filename.c_str(), static_cast<long long>(ms),
                 static_cast<unsigned long long>(getNumVertices()));
}
void GSScene::loadTestScene(int device) {
    Vertex v{};
    v.position[2] = -3.0f;
    v.position[3] = 1.0f;
    v.scale_opacity[0] = v.scale_opacity[1] = v.scale_opacity[2] = 0.1f;
    v.scale_opacity[3] = 0.5f;
    v.rotation[0] = 1.0f;
    v.shs[0] = v.shs[1] = v.shs[2] = 1.0f;
    check(gs_scene_from_vertices(reinterpret_cast<const float*>(&v), 1, device, &scene));
    publishBuffers();
}
uint64_t GSScene::getNumVertices() const { return gs_scene_num_vertices(scene); }

// ---------------------------------------------------------------- Renderer
Renderer::Renderer(VulkanSplatting::RendererConfiguration configuration_)
    : configuration(std::move(configuration_)), window(configuration.window) {}

Renderer::~Renderer() {
    gs_renderer_destroy(renderer);
    for (void* p : d_bgra)
        if (p) (void)hipFree(p);
}

void Renderer::initialize() {  // Renderer.cpp:19-31 without Vulkan / swapchain / GUI / pipelines
    if (!window) throw std::runtime_error("RendererConfiguration.window is null");
    const int device = configuration.physicalDeviceId ? static_cast<int>(*configuration.physicalDeviceId) : 0;
    scene = std::make_shared<GSScene>(configuration.scene);  // loadSceneToGPU, Renderer.cpp:157-164
    scene->load(device);
    check(gs_renderer_create(scene->handle(), &renderer));
    metricsCsv = env_or("GS_METRICS_CSV", "");
    // The reference has one frame in flight (VulkanContext.h:6) and reads the frame's timestamps back before the next one
    // (Renderer.cpp:428-450).  Here a loop that neither dumps frames nor logs per-frame metrics keeps up to three frames queued
    // (each into its own image, presented in order): GS_FRAMES_IN_FLIGHT, default 3; 1 = the reference's behaviour.
    framesInFlight = 1;
    if (!window->wantsFrame() && metricsCsv.empty()) {
        const std::string v = env_or("GS_FRAMES_IN_FLIGHT", "3");
        framesInFlight = std::max(1, std::min(kMaxImages, std::atoi(v.c_str())));
    }
    check(gs_set_frames_in_flight(renderer, framesInFlight));
}

void Renderer::handleInput() {  // Renderer.cpp:33-83 (GUI capture checks drop out: there is no GUI)
    const auto translation = window->getCursorTranslation();
    const auto keys = window->getKeys();
    // There is no ImGui here: wantCaptureMouse / wantCaptureKeyboard are false; guiManager.mouseCapture is mouseCaptured.
    if ((!configuration.enableGui || !mouseCaptured) && window->getMouseButton()[0]) {
        window->mouseCapture(true);
        mouseCaptured = true;
    }
    if ((!configuration.enableGui || mouseCaptured) && (translation[0] != 0.0 || translation[1] != 0.0)) {
        camera.rotation = gs::rotate(camera.rotation, static_cast<float>(translation[0]) * 0.005f, gs::vec3{0.0f, -1.0f, 0.0f});
        camera.rotation = gs::rotate(camera.rotation, static_cast<float>(translation[1]) * 0.005f, gs::vec3{-1.0f, 0.0f, 0.0f});
    }
    gs::vec3 direction{};
    if (keys[0]) direction = direction + gs::vec3{0.0f, 0.0f, -1.0f};
    if (keys[1]) direction = direction + gs::vec3{-1.0f, 0.0f, 0.0f};
    if (keys[2]) direction = direction + gs::vec3{0.0f, 0.0f, 1.0f};
    if (keys[3]) direction = direction + gs::vec3{1.0f, 0.0f, 0.0f};
    if (keys[4]) direction = direction + gs::vec3{0.0f, 1.0f, 0.0f};
    if (keys[5]) direction = direction + gs::vec3{0.0f, -1.0f, 0.0f};
    if (keys[6]) {
        window->mouseCapture(false);
        mouseCaptured = false;
    }
    if (direction.x != 0.0f || direction.y != 0.0f || direction.z != 0.0f)
        camera.position = camera.position + (camera.rotation * gs::normalize(direction)) * 0.3f;
}

void Renderer::draw() {  // Renderer.cpp:366-426: handleInput, updateUniforms, the passes, present
    handleInput();
    const auto [width, height] = window->getFramebufferSize();
    gs_camera cam{};
    cam.position[0] = camera.position.x;
    cam.position[1] = camera.position.y;
    cam.position[2] = camera.position.z;
    cam.rotation[0] = camera.rotation.w;
    cam.rotation[1] = camera.rotation.x;
    cam.rotation[2] = camera.rotation.y;
    cam.rotation[3] = camera.rotation.z;
    cam.fov = camera.fov;
    cam.near_plane = camera.nearPlane;
    cam.far_plane = camera.farPlane;
    gs_uniforms u{};
    check(gs_camera_uniforms(&cam, width, height, &u));
    const uint64_t need = static_cast<uint64_t>(width) * height * 4;
    if (need > bgraBytes) {  // the "swapchain images": B8G8R8A8_UNORM (Swapchain.cpp:22-28), one per frame in flight
        check(gs_synchronize(renderer));
        for (int k = 0; k < framesInFlight; ++k) {
            if (d_bgra[k]) (void)hipFree(d_bgra[k]);
            d_bgra[k] = nullptr;
            if (hipMalloc(&d_bgra[k], need) != hipSuccess) throw std::runtime_error("Failed to allocate the frame image");
        }
        bgraBytes = need;
    }
    void* const image = d_bgra[frameIndex % static_cast<uint64_t>(framesInFlight)];
    check(gs_render(renderer, &u, nullptr, static_cast<uint8_t*>(image)));
    if (window->wantsFrame()) {
        check(gs_synchronize(renderer));
        h_bgra.resize(need);
        if (hipMemcpy(h_bgra.data(), image, need, hipMemcpyDeviceToHost) != hipSuccess)
            throw std::runtime_error("Failed to read back the frame image");
        window->present(h_bgra.data(), width, height);
    } else {
        window->present(nullptr, width, height);
    }
    ++frameIndex;
}

void Renderer::retrieveTimestamps() {  // Renderer.cpp:85-100; QueryManager::parseResults names
    gs_frame_stats st{};
    if (framesInFlight > 1) check(gs_poll_stats(renderer, &st, nullptr));  // the newest FINISHED frame: nothing waits
    else check(gs_get_stats(renderer, &st));
    metrics = {{"preprocess", st.ms_preprocess}, {"prefix_sum", st.ms_prefix_sum}, {"preprocess_sort", st.ms_preprocess_sort},
               {"sort", st.ms_sort},             {"tile_boundary", st.ms_tile_boundary}, {"render", st.ms_render}};
    instances = st.num_instances;
    if (!metricsCsv.empty()) {
        std::ofstream f(metricsCsv, std::ios::app);
        if (frameIndex <= 1) f << "frame,instances,preprocess,prefix_sum,preprocess_sort,sort,tile_boundary,render\n";
        f << frameIndex << ',' << instances << ',' << st.ms_preprocess << ',' << st.ms_prefix_sum << ',' << st.ms_preprocess_sort
          << ',' << st.ms_sort << ',' << st.ms_tile_boundary << ',' << st.ms_render << '\n';
    }
}

void Renderer::run() {  // Renderer.cpp:428-450
    while (running) {
        if (!window->tick()) break;
        draw();
        const auto now = std::chrono::high_resolution_clock::now();
        const auto diff = std::chrono::duration_cast<std::chrono::milliseconds>(now - lastFpsTime).count();
        if (diff > 1000) {
            std::fprintf(stderr, "[debug] FPS: %d\n", fpsCounter);
            fpsCounter = 0;
            lastFpsTime = now;
        } else {
            fpsCounter++;
        }
        retrieveTimestamps();
    }
    check(gs_synchronize(renderer));
}

void Renderer::stop() {
    running = false;
    if (renderer) check(gs_synchronize(renderer));
}

// ---------------------------------------------------------------- VulkanSplatting (src/3dgs.cpp:6-44)
#ifdef VKGS_ENABLE_GLFW
std::shared_ptr<Window> VulkanSplatting::createGlfwWindow(std::string name, int width, int height) {
    return std::make_shared<HeadlessWindow>(std::move(name), width, height);
}
#endif

void VulkanSplatting::start() {
    renderer = std::make_shared<Renderer>(configuration);
    renderer->initialize();
    renderer->run();
}
void VulkanSplatting::initialize() {
    renderer = std::make_shared<Renderer>(configuration);
    renderer->initialize();
}
void VulkanSplatting::draw() { renderer->draw(); }
void VulkanSplatting::logTranslation(float x, float y) { configuration.window->logTranslation(x, y); }
void VulkanSplatting::logMovement(float x, float y, float z) { renderer->camera.translate(gs::vec3{x, y, z}); }
void VulkanSplatting::stop() { renderer->stop(); }
