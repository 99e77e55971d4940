// PF_TRACE_LOAD=1: where loading and construction spend their time, one line per step on stderr.
#pragma once
#include <chrono>
#include <cstdio>
#include <cstdlib>

namespace pfh {

struct LoadTrace {
    bool on = getenv("PF_TRACE_LOAD") != nullptr;
    std::chrono::steady_clock::time_point t = std::chrono::steady_clock::now();
    void mark(const char *what) {
        if (!on) return;
        const auto now = std::chrono::steady_clock::now();
        fprintf(stderr, "[load] %-28s %.3fs\n", what, std::chrono::duration<double>(now - t).count());
        t = now;
    }
};

}  // namespace pfh
