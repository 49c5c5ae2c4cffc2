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

// A HOST-PACED consumer (one frame in flight, a wait per frame, then `idle_ms` of application time that varies wildly): the
// completion interval is span + idle, the frame's own span is what the blend cost.  Returns the settled choice or -1.
int tuner_host_paced(double off_span_ms, double on_span_ms, double idle_ms, double idle_jitter, uint64_t seed, int with_spans, int frames) {
    gs_host::BlendTuner t;
    Lcg rng{seed};
    for (int f = 0; f < frames; ++f) {
        const bool ls = t.current();
        const uint32_t round = t.round;
        const double span = ls ? on_span_ms : off_span_ms;
        const double idle = idle_ms * (1.0 + idle_jitter * rng.next());
        if (with_spans) t.sample(static_cast<float>(span + idle), static_cast<float>(span), ls, round);
        else t.sample(static_cast<float>(span + idle), ls, round);
        if (t.phase == 3) return t.choice ? 1 : 0;
    }
    return -1;
}

// A caller alternating two frame shapes (a stereo pair, a thumbnail beside the main view): each shape's tuner must settle on its
// own costs.  Returns settled choices as bits (bit 0: shape A chose lockstep, bit 1: shape B), or -1 if either never settled;
// *frames_used = frames until both had settled.
int tuner_bank_alternating(double a_off, double a_on, double b_off, double b_on, int in_flight, int frames, int* frames_used) {
    gs_host::BlendTunerBank bank;
    struct P {
        int shape;
        uint32_t w, h;
        bool ls;
        uint32_t round;
    };
    std::deque<P> q;
    *frames_used = -1;
    for (int f = 0; f < frames; ++f) {
        while (static_cast<int>(q.size()) >= in_flight) {
            const P p = q.front();
            q.pop_front();
            const bool a = p.w == 1920;
            bank.sample(p.shape, p.w, p.h, static_cast<float>(p.ls ? (a ? a_on : b_on) : (a ? a_off : b_off)), 0.0f, p.ls, p.round);
        }
        const bool a = (f & 1) == 0;
        const uint32_t w = a ? 1920 : 640, h = a ? 1080 : 360;
        const int shape = bank.select(w, h);
        q.push_back({shape, w, h, bank.current().current(), bank.current().round});
        const int ia = bank.select(1920, 1080), ib = bank.select(640, 360);
        if (bank.e[ia].tuner.phase == 3 && bank.e[ib].tuner.phase == 3) {
            *frames_used = f + 1;
            return (bank.e[ia].tuner.choice ? 1 : 0) | (bank.e[ib].tuner.choice ? 2 : 0);
        }
    }
    return -1;
}

// more shapes than the bank holds: the least recently used one is replaced, a replaced shape's late samples go nowhere, pinning
// reaches every tuner (the ones created later included)
int tuner_bank_replaces_lru(void) {
    gs_host::BlendTunerBank bank;
    int idx[6];
    for (int i = 0; i < 4; ++i) idx[i] = bank.select(100 + i, 100);
    for (int i = 0; i < 4; ++i)
        for (int j = i + 1; j < 4; ++j)
            if (idx[i] == idx[j]) return 0;
    bank.select(100, 100);                 // shape 0 is now the most recently used; shape 1 the least
    idx[4] = bank.select(500, 500);
    if (idx[4] != idx[1]) return 0;
    const uint32_t round = bank.e[idx[4]].tuner.round;
    for (int i = 0; i < 100; ++i) bank.sample(idx[1], 101, 100, 1.0f, 0.0f, false, round);  // the replaced shape's frames retire late
    if (bank.e[idx[4]].tuner.seen != 0 || bank.e[idx[4]].tuner.count != 0) return 0;
    if (bank.select(100, 100) != idx[0]) return 0;
    bank.pin(1);
    idx[5] = bank.select(900, 900);
    if (!bank.current().current() || !bank.e[idx[0]].tuner.current()) return 0;
    bank.pin(-1);
    return bank.current().forced == -1 && bank.current().phase == 0 ? 1 : 0;
}

}  // extern "C"
