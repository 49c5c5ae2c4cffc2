// Window.h -- what the renderer needs from a window: framebuffer size, polled input, a frame tick,
// and a place to hand the finished frame.  Same role and method names as the reference's abstract
// Window (src/vulkan/Window.h:10-34) minus the Vulkan surface/extension hooks (there is no surface).
#pragma once
#include <array>
#include <cstdint>
#include <string>
#include <utility>
#include <vector>

class Window {
public:
    virtual std::array<bool, 3> getMouseButton() { return {false, false, false}; }
    [[nodiscard]] virtual std::pair<uint32_t, uint32_t> getFramebufferSize() const = 0;
    virtual std::array<double, 2> getCursorTranslation() { return {0, 0}; }
    // W, A, S, D, space, shift, escape
    virtual std::array<bool, 7> getKeys() { return {false, false, false, false, false, false, false}; }
    virtual void mouseCapture(bool) {}
    virtual bool tick() { return false; }
    virtual void logTranslation(float, float) {}
    virtual void logMovement(float, float) {}
    // Presentation hook (the swapchain's role): true if the window wants the B8G8R8A8 frame on the host.
    virtual bool wantsFrame() const { return false; }
    virtual void present(const uint8_t* /*bgra*/, uint32_t /*width*/, uint32_t /*height*/) {}
    virtual ~Window() = default;
};

// Headless window for GPU servers: fixed size; tick() returns true GS_FRAMES times (default 1,
// 0 = forever); pan deltas pushed through logTranslation accumulate like the reference's MetalWindow
// (windowing/MetalWindow.cpp:24-45); GS_DUMP_DIR=<dir> writes each presented frame as frame_%05d.ppm.
// GS_CAMERA_PATH=<file> scripts the polled input the GLFW window would report, one line per tick:
//     <cursor dx> <cursor dy> <keys>
// with keys a string over W A S D ' '->'_' shift->'^' escape->'x' ('-' = none); a line with a cursor delta also
// reports mouse button 0 (the reference captures the mouse on that button, Renderer.cpp:36-44).  Lines are consumed
// in order; after the last one the input is idle.  This makes a fly-through reproducible frame for frame.
class HeadlessWindow final : public Window {
public:
    HeadlessWindow(std::string name, int width, int height);
    [[nodiscard]] std::pair<uint32_t, uint32_t> getFramebufferSize() const override { return {width, height}; }
    std::array<double, 2> getCursorTranslation() override;
    std::array<bool, 3> getMouseButton() override { return {buttonDown, false, false}; }
    std::array<bool, 7> getKeys() override { return keys; }
    bool tick() override;
    void logTranslation(float x, float y) override;
    bool wantsFrame() const override { return !dumpDir.empty(); }
    void present(const uint8_t* bgra, uint32_t w, uint32_t h) override;

    uint64_t framesPresented = 0;

private:
    std::string name;
    uint32_t width, height;
    long long frameBudget = 1;
    long long ticks = 0;
    double accumulatedX = 0, accumulatedY = 0;
    bool buttonDown = false;
    std::string dumpDir;
    struct ScriptedInput {
        double dx, dy;
        std::array<bool, 7> keys;
    };
    std::vector<ScriptedInput> script;  // GS_CAMERA_PATH
    std::array<bool, 7> keys{};
};
