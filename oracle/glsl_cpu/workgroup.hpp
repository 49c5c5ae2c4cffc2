// workgroup.hpp -- a compute WORKGROUP on the CPU: barrier(), shared memory, subgroup operations, atomics.
//
// TEST INFRASTRUCTURE ONLY (oracle/_ref).  The reference's radix sort (src/shaders/sort/hist.comp, sort.comp) cannot be run as
// one-invocation-after-the-other like the other shaders: its invocations meet at barrier()s inside a loop, exchange values
// through subgroupAdd / subgroupExclusiveAdd / subgroupBroadcast and through `shared` arrays updated with atomicAdd.  Here every
// invocation of a workgroup is a FIBER (ucontext) with its own stack, the shader's main() runs in it exactly as written, and
//   * barrier()              parks the fiber until every live invocation of the workgroup has arrived;
//   * a subgroup operation   deposits the lane's operand, parks the fiber until all lanes of its subgroup have deposited theirs,
//                            and returns this lane's result -- the semantics of GL_KHR_shader_subgroup for a fully active
//                            subgroup (the sort's subgroup operations sit under `if (lID < 256)` with 256 invocations: every lane
//                            is active; a lane that reaches a barrier or ends while its subgroup waits is reported as an error);
//   * atomicAdd              is a plain read-modify-write: the fibers of one workgroup run on one thread, one at a time.
// Workgroups are independent (they only meet in global memory between dispatches) and are spread over OpenMP threads.
// The SUBGROUP SIZE is a parameter of the dispatch: sort.comp hard-codes `#define SUBGROUP_SIZE 32` for the size of its `sums`
// array, the tests run it with 32 (what the text assumes) and with 64 (what an AMD device would give it).
#pragma once
#include <ucontext.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#include "glsl_compat.hpp"

namespace glsl {

inline thread_local uint gl_SubgroupID, gl_SubgroupInvocationID, gl_SubgroupSize, gl_NumSubgroups;

struct WorkgroupRun {
    static constexpr size_t kStack = 64 * 1024;
    struct Fiber {
        ucontext_t ctx;
        bool done = false;
    };
    struct Subgroup {
        uint arrived = 0, generation = 0;
        uint operand[64], other[64], result[64];
    };
    uint size = 0, subgroup_size = 32;
    uvec3 group_id;
    ucontext_t scheduler;
    std::vector<Fiber> fibers;
    std::vector<Subgroup> subgroups;
    std::vector<char> stacks;
    uint current = 0, alive = 0, at_barrier = 0, barrier_generation = 0;
    unsigned long long progress = 0;
    void (*body)() = nullptr;

    void enter(uint i) {  // the built-in variables of invocation i
        gl_WorkGroupID = group_id;
        gl_LocalInvocationID = uvec3(i, 0u, 0u);
        gl_LocalInvocationIndex = i;
        gl_GlobalInvocationID = uvec3(group_id.x * size + i, 0u, 0u);
        gl_SubgroupSize = subgroup_size;
        gl_NumSubgroups = (size + subgroup_size - 1) / subgroup_size;
        gl_SubgroupID = i / subgroup_size;
        gl_SubgroupInvocationID = i % subgroup_size;
    }
    void yield() {
        const uint me = current;
        swapcontext(&fibers[me].ctx, &scheduler);
        enter(me);
    }
    [[noreturn]] static void fail(const char* what) {
        std::fprintf(stderr, "glsl_cpu workgroup: %s\n", what);
        std::abort();
    }
};
inline thread_local WorkgroupRun* wg_run = nullptr;

inline void wg_trampoline() {
    WorkgroupRun* w = wg_run;
    w->body();
    w->fibers[w->current].done = true;
    --w->alive;
    ++w->progress;
    // (a barrier the others wait at is released by the last live invocation that arrives; an invocation that ends early simply
    //  stops counting -- GLSL requires barrier() in uniform control flow, the sort's are)
    swapcontext(&w->fibers[w->current].ctx, &w->scheduler);
}

inline void barrier() {
    WorkgroupRun* w = wg_run;
    const uint generation = w->barrier_generation;
    ++w->progress;  // (an arrival is progress: a round of the scheduler in which no invocation moved is a deadlock)
    if (++w->at_barrier == w->alive) {
        w->at_barrier = 0;
        ++w->barrier_generation;
        return;
    }
    while (w->barrier_generation == generation) w->yield();
}

// One collective step of a subgroup: every lane deposits (a, b), the last one to arrive evaluates `f(lane, operands a, operands b,
// lanes)` for all lanes.
template <class F> inline uint subgroup_collective(uint a, uint b, F f) {
    WorkgroupRun* w = wg_run;
    const uint me = w->current;
    WorkgroupRun::Subgroup& s = w->subgroups[me / w->subgroup_size];
    const uint lane = me % w->subgroup_size;
    const uint first = (me / w->subgroup_size) * w->subgroup_size;
    const uint lanes = (first + w->subgroup_size <= w->size) ? w->subgroup_size : w->size - first;
    s.operand[lane] = a;
    s.other[lane] = b;
    const uint generation = s.generation;
    ++w->progress;
    if (++s.arrived == lanes) {
        for (uint l = 0; l < lanes; l++) s.result[l] = f(l, s.operand, s.other, lanes);
        s.arrived = 0;
        ++s.generation;
    } else {
        while (s.generation == generation) w->yield();
    }
    return s.result[lane];
}
inline uint subgroupAdd(uint v) {
    return subgroup_collective(v, 0u, [](uint, const uint* a, const uint*, uint lanes) {
        uint sum = 0;
        for (uint l = 0; l < lanes; l++) sum += a[l];
        return sum;
    });
}
inline uint subgroupExclusiveAdd(uint v) {
    return subgroup_collective(v, 0u, [](uint lane, const uint* a, const uint*, uint) {
        uint sum = 0;
        for (uint l = 0; l < lane; l++) sum += a[l];
        return sum;
    });
}
inline uint subgroupBroadcast(uint v, uint id) {  // `id` is the same in every lane of the subgroup (the specification demands it)
    return subgroup_collective(v, id, [](uint lane, const uint* a, const uint* ids, uint lanes) { return ids[lane] < lanes ? a[ids[lane]] : 0u; });
}
inline bool subgroupElect() { return gl_SubgroupInvocationID == 0u; }  // the lowest active lane of a fully active subgroup

template <class T, class V> inline T atomicAdd(T& mem, V value) {
    const T old = mem;
    mem = T(old + T(value));
    return old;
}
inline uint bitCount(uint v) { return uint(__builtin_popcount(v)); }

// Runs `body` (a shader's main) for `groups` workgroups of `local_size` invocations with subgroups of `subgroup_size` lanes.
inline void dispatch_workgroups(uint groups, uint local_size, uint subgroup_size, void (*body)()) {
    if (subgroup_size == 0 || subgroup_size > 64) WorkgroupRun::fail("subgroup size must be 1..64");
#pragma omp parallel
    {
        WorkgroupRun w;
        w.size = local_size;
        w.subgroup_size = subgroup_size;
        w.body = body;
        w.fibers.resize(local_size);
        w.subgroups.resize((local_size + subgroup_size - 1) / subgroup_size);
        w.stacks.resize(size_t(local_size) * WorkgroupRun::kStack);
        wg_run = &w;
#pragma omp for schedule(dynamic, 1)
        for (long g = 0; g < long(groups); g++) {
            w.group_id = uvec3(uint(g), 0u, 0u);
            w.alive = local_size;
            w.at_barrier = 0;
            for (auto& s : w.subgroups) s.arrived = 0;
            for (uint i = 0; i < local_size; i++) {
                WorkgroupRun::Fiber& f = w.fibers[i];
                f.done = false;
                getcontext(&f.ctx);
                f.ctx.uc_stack.ss_sp = w.stacks.data() + size_t(i) * WorkgroupRun::kStack;
                f.ctx.uc_stack.ss_size = WorkgroupRun::kStack;
                f.ctx.uc_link = nullptr;
                makecontext(&f.ctx, wg_trampoline, 0);
            }
            while (w.alive) {
                const unsigned long long before = w.progress;
                for (uint i = 0; i < local_size; i++) {
                    if (w.fibers[i].done) continue;
                    w.current = i;
                    w.enter(i);
                    swapcontext(&w.scheduler, &w.fibers[i].ctx);
                }
                if (w.alive && w.progress == before)
                    WorkgroupRun::fail("deadlock: invocations wait at a barrier or a subgroup operation that the others never reach "
                                       "(barrier / subgroup operation in non-uniform control flow)");
            }
        }
        wg_run = nullptr;
    }
}

}  // namespace glsl
