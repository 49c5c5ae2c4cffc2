// A renderer's frame loop around gs_host::BlendTuner with the device replaced by a clock model: frames are enqueued with the
// tuner's current setting and round, `in_flight` of them are pending, and each retires with the interval its setting costs
// (times a clock factor that ramps from `ramp_from` down to 1 over `ramp_frames` frames, times a seeded jitter).
// Built by tests/test_blend_tuner.py with g++; nothing here touches a GPU.
#include <cstdint>
#include <deque>

#include "gs_blend_tuner.h"

namespace {
struct Lcg {
    uint64_t s;
    double next() {  // uniform in [-1, 1)
        s = s * 6364136223846793005ull + 1442695040888963407ull;
        return static_cast<double>(s >> 11) / 4503599627370496.0 - 1.0;
    }
};
struct Pending {
    bool lockstep;
    uint32_t round;
};
}  // namespace

extern "C" {

// returns the settled choice (0 / 1), or -1 if the tuner never settled; *settled_at = frames enqueued when it did;
// *on_frames = frames that ran in lockstep before that
int tuner_sim(double off_ms, double on_ms, double ramp_from, int ramp_frames, double jitter, int in_flight, int frames, uint64_t seed,
              int forced, int* settled_at, int* on_frames) {
    gs_host::BlendTuner t;
    t.forced = forced;
    Lcg rng{seed};
    std::deque<Pending> q;
    int on = 0, retired = 0;
    *settled_at = -1;
    for (int f = 0; f < frames; ++f) {
        while (static_cast<int>(q.size()) >= in_flight) {
            const Pending p = q.front();
            q.pop_front();
            const double clock = retired < ramp_frames ? ramp_from + (1.0 - ramp_from) * retired / ramp_frames : 1.0;
            t.sample(static_cast<float>((p.lockstep ? on_ms : off_ms) * clock * (1.0 + jitter * rng.next())), p.lockstep, p.round);
            ++retired;
        }
        if (*settled_at < 0 && (t.forced >= 0 || t.phase == 3)) *settled_at = f;
        const bool ls = t.current();
        if (*settled_at < 0 && ls) ++on;
        q.push_back({ls, t.round});
    }
    *on_frames = on;
    if (t.forced >= 0) return t.current() ? 1 : 0;
    return t.phase == 3 ? (t.choice ? 1 : 0) : -1;
}

// the periodic second look: frames until the round after a settled one begins, and what the second look chose
int tuner_relook(double off_ms, double on_first_ms, double on_later_ms, int in_flight, int* first_choice, int* relook_frame) {
    gs_host::BlendTuner t;
    std::deque<Pending> q;
    *first_choice = -1;
    *relook_frame = -1;
    bool settled_once = false;
    for (int f = 0; f < 3 * static_cast<int>(gs_host::BlendTuner::kPeriod); ++f) {
        while (static_cast<int>(q.size()) >= in_flight) {
            const Pending p = q.front();
            q.pop_front();
            t.sample(static_cast<float>(p.lockstep ? (settled_once ? on_later_ms : on_first_ms) : off_ms), p.lockstep, p.round);
        }
        if (!settled_once && t.phase == 3) {
            settled_once = true;
            *first_choice = t.choice ? 1 : 0;
        } else if (settled_once && *relook_frame < 0 && t.phase != 3) {
            *relook_frame = f;
        } else if (*relook_frame >= 0 && t.phase == 3) {
            return t.choice ? 1 : 0;
        }
        q.push_back({t.current(), t.round});
    }
    return -1;
}

// frames of an earlier round (still in flight when restart() was called) must not be counted
int tuner_ignores_stale(void) {
    gs_host::BlendTuner t;
    const uint32_t old_round = t.round;
    t.restart();
    for (int i = 0; i < 100; ++i) t.sample(1.0f, false, old_round);
    return t.seen == 0 && t.count == 0 && t.phase == 0 ? 1 : 0;
}

}  // extern "C"
