// Minimal fork-join helper for the embarrassingly parallel per-bubble host phases (the role `-t`
// plays in the reference, src/CDBG.cpp:1723-1780 -- but results are always assembled in task
// order, so the output stays the `-t 1` one).
#pragma once
#include <atomic>
#include <cstddef>
#include <thread>
#include <vector>

namespace pfh {

// fn(chunk_index, begin, end) for consecutive chunks of `chunk` items, dynamically scheduled
template <class F>
void parallel_chunks(size_t n, size_t chunk, unsigned threads, F &&fn) {
    if (n == 0) return;
    if (chunk == 0) chunk = 1;
    const size_t n_chunks = (n + chunk - 1) / chunk;
    if (threads <= 1 || n_chunks == 1) {
        for (size_t c = 0; c < n_chunks; ++c) fn(c, c * chunk, std::min(n, (c + 1) * chunk));
        return;
    }
    std::atomic<size_t> next{0};
    auto worker = [&]() {
        for (;;) {
            const size_t c = next.fetch_add(1, std::memory_order_relaxed);
            if (c >= n_chunks) return;
            fn(c, c * chunk, std::min(n, (c + 1) * chunk));
        }
    };
    const unsigned T = (unsigned)std::min<size_t>(threads, n_chunks);
    std::vector<std::thread> pool;
    pool.reserve(T - 1);
    for (unsigned t = 1; t < T; ++t) pool.emplace_back(worker);
    worker();
    for (auto &th : pool) th.join();
}

inline size_t n_chunks_of(size_t n, size_t chunk) { return chunk ? (n + chunk - 1) / chunk : 0; }

}  // namespace pfh
