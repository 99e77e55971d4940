// Host-only check of pfh::parallel_chunks / pfh::WorkPool (ploidyfrost_amd/csrc/host/pf_parallel.hpp): every chunk runs
// exactly once, results do not depend on the thread count, and regions issued concurrently from several threads (the
// pipeline stages of PloidyEstimation do that) complete without losing work.  Built and run by tests/test_host_logic_cpu.py.
#include <atomic>
#include <cstdio>
#include <numeric>
#include <thread>
#include <vector>

#include "pf_parallel.hpp"

int main() {
    using namespace pfh;
    int bad = 0;
    for (unsigned threads : {1u, 2u, 7u, 32u}) {
        for (size_t n : {0ul, 1ul, 5ul, 1000ul, 100003ul}) {
            std::vector<int> hit(n, 0);
            std::atomic<size_t> chunks{0};
            parallel_chunks(n, 97, threads, [&](size_t ci, size_t b, size_t e) {
                if (b != ci * 97 || e > n || e <= b) ++bad;
                for (size_t i = b; i < e; ++i) hit[i]++;
                chunks++;
            });
            for (int h : hit) bad += h != 1;
            bad += chunks != n_chunks_of(n, 97);
        }
    }
    // four threads, each issuing many regions of its own while the others do the same
    std::vector<std::thread> users;
    std::atomic<long> total{0};
    for (int t = 0; t < 4; ++t)
        users.emplace_back([&, t] {
            for (int rep = 0; rep < 200; ++rep) {
                std::vector<long> part(64, 0);
                parallel_chunks(6400, 100, 8, [&](size_t ci, size_t b, size_t e) {
                    long s = 0;
                    for (size_t i = b; i < e; ++i) s += (long)i * (t + 1);
                    part[ci] = s;
                });
                total += std::accumulate(part.begin(), part.end(), 0L);
            }
        });
    for (auto &u : users) u.join();
    const long one = 6399L * 6400 / 2;
    bad += total != 200 * one * (1 + 2 + 3 + 4);
    printf(bad ? "FAILED %d\n" : "ok\n", bad);
    return bad != 0;
}
