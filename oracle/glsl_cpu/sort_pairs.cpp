// sort_pairs.cpp -- stands in for the reference's eight hist.comp + sort.comp passes (Renderer.cpp:598-629):
// an LSD radix sort over all 64 key bits is a stable ascending sort of the keys, nothing more.
// TEST INFRASTRUCTURE ONLY (part of oracle/_ref/libgs_ref.so).
#include <algorithm>
#include <cstdint>
#include <numeric>
#include <vector>

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
