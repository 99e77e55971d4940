// The commit replay of findSuperBubble spread over host threads.
//
// The commits are order-dependent (reference src/CDBG.cpp:206-214: every candidate entrance in unitig order, behind the
// `partner == NULL` gate), but two traversal records whose footprints are disjoint commute.  Footprints are known before any
// commit runs (pf_state_ops.hpp lists what a record can touch), so the records fall into connected components of the graph
// {unitig sides, "touched by one record"}; a component is replayed in record order by one thread, different components side by
// side.  On the bench graphs the largest component holds 1 % of the work (tools/exp_replay_components.py).
//
//   SideComponents   the components on the host (sequential union-find): the device-less API, the CPU tests and the footprint
//                    check; the product path gets them from the device (pf_side_components, pf_replay_order in pf_cc.hip)
//   ParallelReplay   the executor: per-side flag bytes during the replay, merged into MyUnitig's one byte at the end
//   check_footprints runs the sequential replay with an accessor that verifies every access against the components
#pragma once
#include <algorithm>
#include <atomic>
#include <cstdint>
#include <cstring>
#include <mutex>
#include <vector>

#include "pf_parallel.hpp"
#include "pf_state_ops.hpp"

namespace pfh {

constexpr uint32_t kReplayClasses = 256;   // components are dealt into this many work units (>= threads, dynamically scheduled)

// A component's label is its smallest side, and records come in entrance order: runs of 2048 consecutive labels go to one class, so
// that a class reads (mostly) runs of neighbouring records -- streams the hardware prefetcher follows -- instead of every n-th one.
constexpr uint32_t kReplayClassShift = 11;
inline uint32_t replay_class_of(uint32_t label, uint32_t n_classes) { return (label >> kReplayClassShift) % n_classes; }

class SideComponents {
public:
    void reset(uint32_t n_unitigs);
    // the records of one slice (cumulative: components only ever merge); list_of(r) = the record's vertex list
    template <class ListOf>
    void add(const pf_bfs_record *rec, uint64_t n, ListOf list_of) {
        for (uint64_t i = 0; i < n; ++i)
            if (record_effective(rec[i])) add_record(rec[i], list_of(rec[i]));
    }
    uint32_t label(uint32_t side) { return find(side); }
    // colored path: the accept commit may mark an endpoint with an incomplete colour set NON_SUPER (pf_state_ops.hpp)
    void set_colour_gate(const ColourGate *g) { gate_ = g; }
    // order[] = record indices grouped by class (ascending inside a class), class_off[n_classes + 1]
    void order(const pf_bfs_record *rec, uint64_t n, uint32_t n_classes, std::vector<uint32_t> &order, std::vector<uint32_t> &class_off);

private:
    void add_record(const pf_bfs_record &r, const uint32_t *list);
    uint32_t find(uint32_t x) {
        while (parent_[x] != x) {
            parent_[x] = parent_[parent_[x]];
            x = parent_[x];
        }
        return x;
    }
    void unite(uint32_t a, uint32_t b) {
        a = find(a);
        b = find(b);
        if (a == b) return;
        if (a < b) parent_[b] = a; else parent_[a] = b;
    }
    void partner(uint32_t side, uint32_t other_side);
    std::vector<uint32_t> parent_, first_;
    const ColourGate *gate_ = nullptr;
};

struct ReplayStats {
    uint64_t large = 0, large_seen = 0, max_seen = 0, large_used = 0, large_used_max = 0;
    void merge(const ReplayStats &o) {
        large += o.large; large_seen += o.large_seen; large_used += o.large_used;
        max_seen = std::max(max_seen, o.max_seen);
        large_used_max = std::max(large_used_max, o.large_used_max);
    }
};

class ParallelReplay {
public:
    // state of a pass: plus / minus are the caller's arrays (zeroed by the caller), the per-side flag bytes live here
    void begin(uint32_t n_unitigs, uint32_t *plus, uint32_t *minus, size_t complex_size, unsigned threads) {
        n_ = n_unitigs;
        plus_ = plus;
        minus_ = minus;
        z_ = complex_size;
        if (f2_.size() != 2 * (size_t)n_unitigs) f2_.assign(2 * (size_t)n_unitigs, 0);
        else parallel_chunks(f2_.size(), 1u << 20, threads, [&](size_t, size_t b, size_t e) { memset(f2_.data() + b, 0, e - b); });
    }

    // the records of one slice in the device's (or SideComponents') class order
    template <class ListOf>
    void run(const pf_bfs_record *rec, ListOf list_of, const uint32_t *order, const uint32_t *class_off, uint32_t n_classes, unsigned threads,
             ReplayStats &stats) {
        std::mutex mu;
        const std::function<void(size_t)> body = [&](size_t c) {
            Commits<FlagsPerSide> cm{FlagsPerSide{f2_.data(), plus_, minus_}, z_, NoColours{}};
            ReplayStats st;
            const uint32_t *o = order + class_off[c];
            const size_t n = class_off[c + 1] - class_off[c];
            for (size_t j = 0; j < n; ++j) {
                // the commits chase per-unitig state at random: pull the state of a record a few iterations ahead into cache
                if (j + 32 < n) __builtin_prefetch(&rec[o[j + 32]]);
                if (j + 12 < n) {
                    const pf_bfs_record &nx = rec[o[j + 12]];
                    __builtin_prefetch(list_of(nx));
                    __builtin_prefetch(&plus_[nx.entrance >> 1]);
                    __builtin_prefetch(&minus_[nx.entrance >> 1]);
                }
                if (j + 6 < n) {
                    const pf_bfs_record &nx = rec[o[j + 6]];
                    const uint32_t *l = list_of(nx);
                    const uint32_t nl = nx.n_list < 6 ? nx.n_list : 6;
                    for (uint32_t q = 0; q < nl; ++q) {
                        const uint32_t w = l[q] >> 1;
                        __builtin_prefetch(&f2_[2 * (size_t)w]);
                        __builtin_prefetch(&plus_[w]);
                        __builtin_prefetch(&minus_[w]);
                    }
                }
                const pf_bfs_record &r = rec[o[j]];
                if (r.n_seen > 4096) { st.large++; st.large_seen += r.n_seen; }
                if (r.n_seen > st.max_seen) st.max_seen = r.n_seen;
                if (!cm.gate_open(r.entrance)) continue;
                if (r.n_seen > 4096) { st.large_used++; st.large_used_max = std::max<uint64_t>(st.large_used_max, r.n_seen); }
                cm.replay(r, list_of(r));
            }
            std::lock_guard<std::mutex> lk(mu);
            stats.merge(st);
        };
        WorkPool::instance().run(n_classes, threads, body);
    }

    // MyUnitig's one flag byte per unitig from the two side bytes
    void finish(uint8_t *flags, unsigned threads) const {
        parallel_chunks(n_, 1u << 18, threads, [&](size_t, size_t b, size_t e) {
            for (size_t u = b; u < e; ++u) flags[u] = FlagsPerSide::merged(f2_[2 * u], f2_[2 * u + 1]);
        });
    }

private:
    uint32_t n_ = 0;
    uint32_t *plus_ = nullptr, *minus_ = nullptr;
    size_t z_ = 8;
    std::vector<uint8_t> f2_;
};

// Sequential replay of the records in slices of `slice` records (0 = all at once) with every access checked against the
// components known when the slice starts to run (cumulative, as the executor sees them): returns the number of accesses to a side
// outside the running record's component; `first_bad` = index of the first such record (or UINT64_MAX).
uint64_t check_footprints(const pf_bfs_record *rec, uint64_t n, const uint32_t *pool, uint32_t n_unitigs, size_t complex_size, uint64_t slice,
                          uint64_t *first_bad, const ColourGate *gate = nullptr);

}  // namespace pfh
