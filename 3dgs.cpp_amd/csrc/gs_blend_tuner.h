// gs_blend_tuner.h -- which of the blend's two schedules a renderer runs (gs_blend.hip: the four waves of a tile in
// LOCKSTEP over the chunks of its list, or each at its own pace), decided by measurement.  Plain C++ without a device in
// sight: tests/test_blend_tuner.py drives it on the CPU with synthetic completion times.
#pragma once

#include <cstdint>

namespace gs_host {

struct BlendTuner {
    // What is compared is the rate at which frames COMPLETE under each setting -- the interval between the completions of consecutive
    // frames (GPU timestamps), summed over a window -- not the blend's own span: with frames in flight a kernel's span says how much
    // of the chip it was given, not what it cost (measured: on the S scenes the lockstepped blend's span is the shorter one with
    // three frames in flight while the frame rate is 5 % lower).  Per setting: kSkip frames ignored after the switch (frames of the
    // other setting are still in flight beside them), kSamples intervals summed.  Lockstep has to win by 3 %: off is the safe side
    // (-9 % at worst against +25 %).
    // The windows run OFF (12) - ON (24) - OFF (12) (round 5: 8 - 16 - 8): a drift of the clocks over the measurement -- a renderer's first frames run on a chip
    // that is still coming up -- then weighs on both settings alike (measured: with a plain off-then-on order config B picked the
    // lockstep it loses 5 % with, three times out of four).  A win for lockstep has to be CONFIRMED by a second pass (won_once).
    // Round 6: a renderer's first kHold frames are not measured at all (a fresh process's clocks ramp for ~100 frames -- convexly, which the
    // symmetric windows do not cancel -- and its first frames may be re-run at another depth-order level), and the windows are 12 - 24 - 12:
    // one process in four of an A/B had called a 2 % LOSS for lockstep twice in a row inside its first hundred frames
    // (profiles/r06_blend_pair_loop.txt).
    static constexpr int kSkip = 6, kWindow = 12, kHold = 40;
    static constexpr uint32_t kPeriod = 4096;       // settled frames between two looks
    int forced = -1;        // -1 automatic, 0 / 1 pinned
    int phase = 0;          // 0: off, 1: on (two windows), 2: off again, 3: settled
    bool choice = false;    // the settled setting
    uint32_t round = 1, seen = 0, settled_frames = 0, held = 0;
    double sum[2] = {0, 0};
    int count = 0;
    bool measuring_on() const { return phase == 1; }
    bool current() const { return forced >= 0 ? forced != 0 : (phase == 3 ? choice : measuring_on()); }
    bool won_once = false;  // lockstep won the pass before this one: a second pass has to agree before it is switched on
    void restart() {
        begin_pass();
        won_once = false;
    }
    void begin_pass() {
        phase = 0;
        seen = settled_frames = 0;
        sum[0] = sum[1] = 0;
        count = 0;
        ++round;
    }
    // What one retired frame contributes: the time since the previous completion -- unless the frame's OWN span (first kernel's
    // start to the blend's end) is shorter, which is the mark of a host-paced consumer (one frame in flight and a wait per frame,
    // a vsync'ed viewer): the interval then holds the host's idle time and measures the application, not the blend (round-5
    // advisor finding), while a frame that had the chip to itself cost exactly its span.  With frames in flight it is the other
    // way round (spans stretch over the other frames' kernels, intervals are what the chip delivers), so the smaller of the two
    // is the right figure in both regimes.  span_ms <= 0: unknown.
    static float cost(float interval_ms, float span_ms) { return span_ms > 0.0f && span_ms < interval_ms ? span_ms : interval_ms; }
    void sample(float interval_ms, float span_ms, bool lockstep, uint32_t frame_round) { sample(cost(interval_ms, span_ms), lockstep, frame_round); }
    // a retired frame: the time since the previous completion (ms), the setting and the round it ran with
    void sample(float interval_ms, bool lockstep, uint32_t frame_round) {
        if (forced >= 0) return;
        if (phase == 3) {
            if (++settled_frames >= kPeriod) restart();
            return;
        }
        if (held < (uint32_t)kHold) {  // the renderer's first frames: not measured
            ++held;
            return;
        }
        if (frame_round != round || lockstep != measuring_on()) return;  // a frame of before the switch
        if (++seen <= (uint32_t)kSkip) return;
        sum[measuring_on() ? 1 : 0] += interval_ms;
        if (++count < (phase == 1 ? 2 * kWindow : kWindow)) return;
        count = 0;
        seen = 0;
        ++round;
        if (++phase == 3) {
            // ON needs two passes that agree (windows of 8 to 16 frames are a few milliseconds on the 6 M scenes: one pass in five called
            // a tie for lockstep there, at 3 % of the frame rate); OFF, the safe side, is taken at once
            const bool wins = sum[1] < 0.97 * sum[0];
            if (wins && !won_once) {
                won_once = true;
                begin_pass();  // (++round again: harmless, frames of the old round are ignored either way)
                return;
            }
            choice = wins;
            won_once = false;
            settled_frames = 0;
        }
    }
};

// One tuner per frame shape (round-5 advisor finding: a caller alternating two resolutions restarted the one tuner on every
// change and never settled).  A handful of shapes, least recently used one replaced; `select` returns the shape's index, which a
// frame carries to its retirement so that its sample reaches the tuner it ran under (and no other).
struct BlendTunerBank {
    static constexpr int kShapes = 4;
    struct Entry {
        uint32_t w = 0, h = 0;
        uint64_t used = 0;
        bool live = false;
        BlendTuner tuner;
    } e[kShapes];
    int forced = -1;
    uint64_t clock = 0;
    int active = 0;
    BlendTuner& current() { return e[active].tuner; }
    const BlendTuner& current() const { return e[active].tuner; }
    int select(uint32_t w, uint32_t h) {
        int victim = 0;
        for (int i = 0; i < kShapes; ++i) {
            if (e[i].live && e[i].w == w && e[i].h == h) {
                e[i].used = ++clock;
                return active = i;
            }
            if (!e[i].live) victim = i;
            else if (e[victim].live && e[i].used < e[victim].used) victim = i;
        }
        e[victim] = Entry{};
        e[victim].w = w;
        e[victim].h = h;
        e[victim].live = true;
        e[victim].used = ++clock;
        e[victim].tuner.forced = forced;
        return active = victim;
    }
    void pin(int mode) {  // -1 automatic (every shape measures afresh), 0 / 1 pinned
        forced = mode;
        for (auto& x : e) {
            x.tuner.forced = mode;
            if (mode < 0) x.tuner.restart();
        }
    }
    void sample(int shape, uint32_t w, uint32_t h, float interval_ms, float span_ms, bool lockstep, uint32_t frame_round) {
        if (shape < 0 || shape >= kShapes || !e[shape].live || e[shape].w != w || e[shape].h != h) return;  // its tuner was replaced
        e[shape].tuner.sample(interval_ms, span_ms, lockstep, frame_round);
    }
};

}  // namespace gs_host
