// par.hpp -- contiguous ranges of [0, n) on a few host threads (the front end's O(N) passes over a 10M-node tree).
#pragma once
#include <algorithm>
#include <cstdint>
#include <cstdlib>
#include <thread>
#include <vector>

namespace uh {

inline unsigned host_threads() {
    if (const char *e = getenv("USHER_AMD_THREADS")) return (unsigned)std::max(1, atoi(e));
    return std::min(32u, std::max(1u, std::thread::hardware_concurrency()));
}

// fn(begin, end, thread index); fewer than min_per_thread items per thread -> fewer threads (one: inline).
// USHER_AMD_GRAIN overrides min_per_thread (the tests set 1 so that small trees are split too).
template <class F>
void parallel_for(uint64_t n, F fn, uint64_t min_per_thread = 8192) {
    const uint64_t grain = getenv("USHER_AMD_GRAIN") ? (uint64_t)std::max(1, atoi(getenv("USHER_AMD_GRAIN"))) : 0;
    if (grain) min_per_thread = grain;
    const unsigned t = (unsigned)std::min<uint64_t>(host_threads(), std::max<uint64_t>(1, n / std::max<uint64_t>(1, min_per_thread)));
    if (t <= 1) { fn((uint64_t)0, n, 0u); return; }
    std::vector<std::thread> th;
    th.reserve(t - 1);
    for (unsigned i = 1; i < t; i++) th.emplace_back([&fn, n, t, i] { fn(n * i / t, n * (i + 1) / t, i); });
    fn((uint64_t)0, n / t, 0u);
    for (auto &x : th) x.join();
}

// in place: a[i] <- sum of a[0..i); returns the total
template <class V>
uint64_t exclusive_scan(V *a, uint64_t n) {
    const unsigned T = host_threads();
    std::vector<uint64_t> part(T + 1, 0);
    parallel_for(n, [&](uint64_t b, uint64_t e, unsigned tid) { uint64_t s = 0; for (uint64_t i = b; i < e; i++) s += a[i]; part[tid + 1] = s; }, 1u << 16);
    for (unsigned i = 0; i < T; i++) part[i + 1] += part[i];
    parallel_for(n, [&](uint64_t b, uint64_t e, unsigned tid) { uint64_t s = part[tid]; for (uint64_t i = b; i < e; i++) { const V v = a[i]; a[i] = (V)s; s += v; } }, 1u << 16);
    return part[T];
}

}  // namespace uh
