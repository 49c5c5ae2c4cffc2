// sort_pairs.cpp -- the sort of the (key, payload) instances between preprocess_sort.comp and tile_boundary.comp.
// TEST INFRASTRUCTURE ONLY (part of oracle/_ref/libgs_ref.so).
//
// gsr_radix_sort_pairs RUNS THE REFERENCE'S OWN SORT: src/shaders/sort/hist.comp + sort/sort.comp, eight times, with the grid,
// the push constants and the buffer ping-pong of Renderer.cpp:598-629 (the shader text is compiled by oracle/build_ref.py like
// every other shader; its workgroups -- barriers, shared arrays, subgroup operations, atomics -- run on glsl_cpu/workgroup.hpp).
// gsr_sort_pairs is what that must equal: an LSD radix sort over all 64 key bits is a stable ascending sort of the keys, which
// std::stable_sort does in a fraction of the time; the per-frame checks use it, tests/test_oracle_vs_ref.py pins the two to each
// other (config A's instances, tie-heavy keys, ragged sizes, a million random 64-bit keys, subgroups of 32 and of 64).
#include <algorithm>
#include <cstdint>
#include <numeric>
#include <vector>

extern "C" void gsr_radix_hist(const uint64_t* keys_in, uint32_t* hist, uint32_t num_elements, uint32_t shift, uint32_t num_workgroups,
                               uint32_t blocks_per_workgroup, uint32_t subgroup_size);
extern "C" void gsr_radix_scatter(const uint64_t* keys_in, uint64_t* keys_out, const uint32_t* payload_in, uint32_t* payload_out,
                                  const uint32_t* hist, uint32_t num_elements, uint32_t shift, uint32_t num_workgroups,
                                  uint32_t blocks_per_workgroup, uint32_t subgroup_size);

extern "C" void gsr_sort_pairs(uint64_t* keys, uint32_t* payloads, uint64_t d) {
    std::vector<uint64_t> order(d);
    std::iota(order.begin(), order.end(), uint64_t(0));
    std::stable_sort(order.begin(), order.end(), [&](uint64_t a, uint64_t b) { return keys[a] < keys[b]; });
    std::vector<uint64_t> k(d);
    std::vector<uint32_t> p(d);
    for (uint64_t i = 0; i < d; i++) {
        k[i] = keys[order[i]];
        p[i] = payloads[order[i]];
    }
    std::copy(k.begin(), k.end(), keys);
    std::copy(p.begin(), p.end(), payloads);
}

// Renderer.cpp:598-629.  blocks_per_workgroup: Renderer.h:134-138 (32; 256 on Apple).  Returns 0, or -1 when d does not fit the
// shaders' 32-bit element count.
extern "C" int gsr_radix_sort_pairs(uint64_t* keys, uint32_t* payloads, uint64_t d, uint32_t blocks_per_workgroup, uint32_t subgroup_size) {
    if (d > 0xffffffffull || blocks_per_workgroup == 0) return -1;
    if (d == 0) return 0;  // (a dispatch of zero groups: nothing runs)
    const uint32_t n = static_cast<uint32_t>(d);
    uint32_t invocation_size = (n + blocks_per_workgroup - 1) / blocks_per_workgroup;  // :600
    invocation_size = (invocation_size + 255) / 256;                                   // :601
    std::vector<uint64_t> k_even(keys, keys + d), k_odd(d);                             // sortKBufferEven / Odd
    std::vector<uint32_t> v_even(payloads, payloads + d), v_odd(d);                    // sortVBufferEven / Odd
    std::vector<uint32_t> hist(size_t(256) * invocation_size);                         // sortHistBuffer
    for (int i = 0; i < 8; i++) {
        const bool even = i % 2 == 0;  // descriptor set 0: Even -> Odd; set 1: Odd -> Even (Renderer.cpp:256-291)
        const uint64_t* kin = even ? k_even.data() : k_odd.data();
        uint64_t* kout = even ? k_odd.data() : k_even.data();
        const uint32_t* vin = even ? v_even.data() : v_odd.data();
        uint32_t* vout = even ? v_odd.data() : v_even.data();
        gsr_radix_hist(kin, hist.data(), n, uint32_t(i * 8), invocation_size, blocks_per_workgroup, subgroup_size);
        gsr_radix_scatter(kin, kout, vin, vout, hist.data(), n, uint32_t(i * 8), invocation_size, blocks_per_workgroup, subgroup_size);
    }
    std::copy(k_even.begin(), k_even.end(), keys);  // after eight passes the result is back in the Even buffers (what tile_boundary reads)
    std::copy(v_even.begin(), v_even.end(), payloads);
    return 0;
}
