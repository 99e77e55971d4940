// Where loading and construction spend their time.  Every step is kept in a process-wide log that the C facade hands out
// (pfh_load_trace: bench.py's `load_breakdown`); PF_TRACE_LOAD=1 also prints one line per step on stderr.
#pragma once
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <mutex>
#include <string>
#include <utility>
#include <vector>

namespace pfh {

struct LoadLog {
    std::mutex mu;
    std::vector<std::pair<std::string, double>> steps;
    static constexpr size_t kMaxSteps = 4096;   // the log of a process that never asks for it (pfh_load_trace(reset = 1)) stays this small
    static LoadLog &get() {
        static LoadLog l;
        return l;
    }
};

struct LoadTrace {
    bool on = getenv("PF_TRACE_LOAD") != nullptr;
    std::chrono::steady_clock::time_point t = std::chrono::steady_clock::now();
    void mark(const char *what) {
        const auto now = std::chrono::steady_clock::now();
        const double s = std::chrono::duration<double>(now - t).count();
        {
            LoadLog &l = LoadLog::get();
            std::lock_guard<std::mutex> lk(l.mu);
            if (l.steps.size() >= LoadLog::kMaxSteps) l.steps.erase(l.steps.begin(), l.steps.begin() + LoadLog::kMaxSteps / 2);   // (a long-lived embedder: the newest half stays)
            l.steps.emplace_back(what, s);
        }
        if (on) fprintf(stderr, "[load] %-28s %.3fs\n", what, s);
        t = now;
    }
};

}  // namespace pfh
