// pfh::CDBG::findSuperBubble_multithread_ptr and the coverage launch it may start for the next phase.
#include <fcntl.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <cmath>
#include <condition_variable>
#include <cstdio>
#include <cstring>
#include <ctime>
#include <deque>
#include <mutex>
#include <set>
#include <stdexcept>
#include <thread>
#include <unordered_map>

#include "pf_cdbg_impl.hpp"
#include "pf_parallel.hpp"

namespace pfh {

// Threads of the parallel commit replay: the single-sample path only (the colour gates of CCDBG's accept commit write NON_SUPER
// on endpoints, outside the footprint model), PF_REPLAY=seq or one thread keep the sequential loop.
unsigned CDBG::replay_threads(size_t thr) const {
    if (col_ != nullptr) return 0;
    static const char *env = getenv("PF_REPLAY");
    if (env && !strcmp(env, "seq")) return 0;
    static const int env_t = [] { const char *e = getenv("PF_REPLAY_THREADS"); return e ? atoi(e) : -1; }();
    int t = replay_threads_ >= 0 ? replay_threads_ : env_t >= 0 ? env_t : (int)std::min<size_t>(threads_ ? threads_ : std::max<size_t>(thr, 1), 32);
    return t >= 2 ? (unsigned)t : 0;
}

// The commits run on the device, one thread per component (pf_cc.hip), on the single-sample path unless the caller asked for
// host threads (set_replay_threads >= 0) or PF_REPLAY says host (components on host threads) or seq (the sequential loop).
bool CDBG::commits_on_device(size_t thr) const {
    (void)thr;
    if ((col_ != nullptr && !colours_on_device_) || !third_tier_on_host_ || replay_threads_ >= 0) return false;
    static const bool off = [] { const char *e = getenv("PF_REPLAY"); return e && (!strcmp(e, "host") || !strcmp(e, "seq")); }();
    static const bool env_threads = getenv("PF_REPLAY_THREADS") != nullptr;
    return !off && !env_threads;
}

int CDBG::sync_state_to_host() {
    if (!state_host_stale_) return 0;
    const int st = pf_call_get_state(ctx_, flags_.data(), plus_.data(), minus_.data());
    if (st != PF_OK) return fail(st, std::string(tag_) + "::findSuperBubble(): " + pf_last_error(ctx_));
    state_host_stale_ = false;
    return 0;
}

// findSuperBubble with nothing but the long traversals on the host: K-BFS leaves its records in HBM, K-CC finds the components,
// pf_replay_device commits every small component with one thread; the few large ones (and those of the traversals walked here)
// are committed on this side and their sides patched into the device state.
int CDBG::find_superbubbles_device(const std::string &outpre, const size_t &thr) {
    const auto t_all = clk::now();
    clock_t c0 = clock();
    const uint32_t N = g_.n();
    const bool trace_find = getenv("PF_TRACE_FIND") != nullptr;
    auto tf = [&](const char *what) { if (trace_find) fprintf(stderr, "[find] %-28s %.2f ms\n", what, since(t_all) * 1e3); };
    out_bytes_ = 0;
    state_on_device_ = false;
    cov_ready_ = false;
    for (auto &hl : huge_lists_) hl.clear();
    std::vector<uint32_t> &deferred = deferred_;
    std::vector<uint32_t> &deferred_ent = deferred_ent_;
    if (deferred.size() < 4096) deferred.resize(4096);
    if (deferred_ent.size() < deferred.size()) deferred_ent.resize(deferred.size());
    uint64_t n_rec = 0, pool_used = 0, n_deferred = 0;
    int st;
    // The long traversals are walked on host cores, side by side -- and from the moment the device gives each of them up: the wave
    // tier reports its give-ups into pinned host memory as they happen (pf_bfs_live_deferred), the walkers poll it while
    // pf_bfs_candidates_resident is still running.  The longest walk bounds this phase (136 k vertices = 2.7 ms at 5 M unitigs); it
    // no longer waits for the device to finish with the other candidates first.  A walk is a function of its entrance alone, so the
    // results are keyed by candidate index: whatever the live list missed (more give-ups than it holds, a repeated call after a pool
    // overflow) is walked afterwards.
    const unsigned walk_threads = (unsigned)std::max<uint64_t>(1, std::min<uint64_t>(threads_ ? threads_ : std::max<size_t>(thr, 1), (4ull << 30) / (4ull * std::max<uint32_t>(N, 1))));
    constexpr uint64_t LIVE_CAP = 4096;
    constexpr bool live_on = true;
    volatile uint64_t *live = nullptr;
    if (pf_bfs_live_deferred(ctx_, live_on ? LIVE_CAP : 0, &live) != PF_OK) live = nullptr;
    struct Walked {
        pf_bfs_record rec;
        std::vector<uint32_t> list;
        uint32_t cand = 0;   // candidate index + 1; 0 = slot unused
    };
    std::vector<Walked> early(live ? LIVE_CAP : 0);
    std::atomic<uint64_t> next_slot{0};
    std::atomic<int64_t> final_n{-1};   // how many entries the live list ends with; -1 = the device is still at it
    std::string walk_err;
    auto walk_one = [&](uint32_t entrance, pf_bfs_record &r, std::vector<uint32_t> &out) {
        std::unique_ptr<HugeWalker> w;
        {
            std::lock_guard<std::mutex> lk(walkers_mu_);
            if (!walkers_.empty()) { w = std::move(walkers_.back()); walkers_.pop_back(); }
        }
        if (!w) w = std::make_unique<HugeWalker>();
        memset(&r, 0, sizeof r);
        r.entrance = entrance;
        r.exit = 0xFFFFFFFFu;
        const auto tw = clk::now();
        const std::vector<uint32_t> &list = w->walk(succ_.data(), pred_.data(), N, r.entrance, r);
        if (getenv("PF_TRACE_BFS")) fprintf(stderr, "[bfs] host walk from %u: %u vertices, outcome %d, %.3f ms (started %.3f ms into findSuperBubble)\n", r.entrance, r.n_seen, (int)r.outcome, since(tw) * 1e3, (std::chrono::duration<double>(tw - t_all).count()) * 1e3);
        out.assign(list.begin(), list.begin() + r.n_list);
        std::lock_guard<std::mutex> lk(walkers_mu_);
        walkers_.push_back(std::move(w));
    };
    std::thread early_walk;
    if (live)
        early_walk = std::thread([&] {
            try {
                parallel_chunks(walk_threads, 1, walk_threads, [&](size_t, size_t, size_t) {
                    for (;;) {
                        const uint64_t k = next_slot.fetch_add(1);
                        if (k >= LIVE_CAP) return;
                        uint64_t e = 0;
                        for (;;) {   // entry k, or the end of the list
                            e = __atomic_load_n(const_cast<const uint64_t *>(&live[k]), __ATOMIC_ACQUIRE);
                            if (e) break;
                            const int64_t fn = final_n.load(std::memory_order_acquire);
                            if (fn >= 0 && (int64_t)k >= fn) return;
                            __builtin_ia32_pause();
                        }
                        Walked &wk = early[(size_t)k];
                        walk_one((uint32_t)(e >> 32), wk.rec, wk.list);
                        wk.cand = (uint32_t)e;   // (index + 1)
                    }
                });
            } catch (const std::exception &e) { walk_err = e.what(); }
        });
    struct EarlyGuard {   // every way out of this function: the walkers learn that the list has ended, and are waited for
        std::thread &t;
        std::atomic<int64_t> &fin;
        ~EarlyGuard() {
            if (fin.load() < 0) fin.store(0);
            if (t.joinable()) t.join();
        }
    } early_guard{early_walk, final_n};
    for (;;) {
        st = pf_bfs_candidates_resident(ctx_, 0, N, &n_rec, &pool_used, deferred.data(), deferred_ent.data(), deferred.size(), &n_deferred);
        if (st == PF_ERR_OVERFLOW && n_deferred > deferred.size()) {
            deferred.resize(n_deferred + n_deferred / 4);
            deferred_ent.resize(deferred.size());
            continue;
        }
        break;
    }
    {
        // (the live list holds notices: of every traversal that reached 48 vertices, whether the device gave it up afterwards or not)
        uint64_t n_live = 0;
        if (st == PF_OK && live) (void)pf_bfs_live_count(ctx_, &n_live);
        final_n.store(st == PF_OK ? (int64_t)std::min<uint64_t>(n_live, LIVE_CAP) : 0, std::memory_order_release);
        if (trace_find) fprintf(stderr, "[find]   %llu notices in the live list, %llu traversals given up\n", (unsigned long long)n_live, (unsigned long long)n_deferred);
    }
    if (st != PF_OK) return fail(st, std::string("CDBG::findSuperBubble(): ") + pf_last_error(ctx_));
    tf("traversed on the device");
    std::vector<pf_bfs_record> walked((size_t)n_deferred);
    std::vector<std::vector<uint32_t>> lists((size_t)n_deferred);
    std::thread walk([&] {
        try {
            if (early_walk.joinable()) early_walk.join();
            // pair the early walks with the device's list by candidate index; walk what is left
            std::unordered_map<uint32_t, size_t> by_cand;
            for (size_t k = 0; k < early.size(); ++k)
                if (early[k].cand) by_cand.emplace(early[k].cand - 1, k);
            std::vector<size_t> todo;
            for (size_t d = 0; d < (size_t)n_deferred; ++d) {
                auto it = by_cand.find(deferred[d]);
                if (it != by_cand.end() && early[it->second].rec.entrance == deferred_ent[d]) {
                    walked[d] = early[it->second].rec;
                    lists[d].swap(early[it->second].list);
                    by_cand.erase(it);
                } else {
                    todo.push_back(d);
                }
            }
            if (!todo.empty())
                parallel_chunks(todo.size(), 1, walk_threads, [&](size_t x, size_t, size_t) { walk_one(deferred_ent[todo[x]], walked[todo[x]], lists[todo[x]]); });
        } catch (const std::exception &e) { walk_err = e.what(); }
    });
    // K-CC over the records that are on the device, beside the walkers
    st = pf_side_components(ctx_, 1, nullptr, n_rec, nullptr, 0, nullptr, 0, nullptr, 0);
    tf("  components of the device's records");
    // ... and the coverage kernel PloidyEstimation starts with (it depends on nothing this phase computes)
    if (st == PF_OK && overlap_output_) cov_ready_ = launch_coverage() == PF_OK;
    tf("  coverage launched");
    walk.join();
    if (!walk_err.empty()) return fail(PF_ERR_HIP, "CDBG::findSuperBubble(): walk of a long traversal: " + walk_err);
    if (st != PF_OK) return fail(st, std::string("CDBG::findSuperBubble(): ") + pf_last_error(ctx_));
    tf("long traversals walked");
    // in candidate order (the device hands the deferred candidates over in the order their wavefronts gave up)
    std::vector<uint32_t> xpool;
    {
        std::vector<size_t> by_index((size_t)n_deferred);
        for (size_t d = 0; d < by_index.size(); ++d) by_index[d] = d;
        std::sort(by_index.begin(), by_index.end(), [&](size_t a, size_t b) { return deferred[a] < deferred[b]; });
        std::vector<pf_bfs_record> w2((size_t)n_deferred);
        std::vector<uint32_t> d2((size_t)n_deferred);
        for (size_t i = 0; i < by_index.size(); ++i) {
            const size_t d = by_index[i];
            w2[i] = walked[d];
            w2[i].list_off = xpool.size();
            w2[i].pad_ = 0;
            xpool.insert(xpool.end(), lists[d].begin(), lists[d].end());
            d2[i] = deferred[d];
        }
        walked.swap(w2);
        std::copy(d2.begin(), d2.end(), deferred.begin());
    }
    tf("  walked records in candidate order");
    if (!walked.empty()) st = pf_side_components(ctx_, 0, nullptr, n_rec, nullptr, 0, walked.data(), walked.size(), xpool.data(), xpool.size());
    tf("  their components");
    uint64_t n_big = 0, big_entries = 0;
    // (PF_REPLAY_SMALL_LIMIT: tests push more components -- all of them with 0 -- through the host half and its patch)
    static const uint32_t small_limit = [] { const char *e = getenv("PF_REPLAY_SMALL_LIMIT"); return e ? (uint32_t)atoi(e) : 256u; }();
    if (st == PF_OK) st = pf_replay_device(ctx_, (uint32_t)std::min<size_t>(complex_size_, 0xFFFFFFFFu), small_limit, &n_big, &big_entries);
    if (st != PF_OK) return fail(st, std::string("CDBG::findSuperBubble(): ") + pf_last_error(ctx_));
    tf("components + commits on the device");
    // what is left for this side: the records of the large components and of the components of the walked traversals, merged
    // in record order
    std::vector<uint32_t> big_idx((size_t)n_big);
    std::vector<pf_bfs_record> big_rec((size_t)n_big);
    std::vector<uint32_t> big_pool((size_t)big_entries + 1);
    st = pf_replay_big_fetch(ctx_, big_idx.data(), big_rec.data(), big_pool.data());
    if (st != PF_OK) return fail(st, std::string("CDBG::findSuperBubble(): ") + pf_last_error(ctx_));
    tf("large components fetched");
    if (big_f2_.size() != 2 * (size_t)N) big_f2_.assign(2 * (size_t)N, 0);
    // (plus_ / minus_ / big_f2_ are all-zero here: every pass undoes what it touched, see below; a stale host copy is re-zeroed)
    if (!state_host_stale_) {
        const unsigned zt = (unsigned)std::min<size_t>(threads_ ? threads_ : std::max<size_t>(thr, 1), 8);
        parallel_chunks(N, 1u << 19, zt, [&](size_t, size_t b, size_t e) {
            memset(plus_.data() + b, 0, (e - b) * 4);
            memset(minus_.data() + b, 0, (e - b) * 4);
        });
    }
    times_.bfs_large = times_.bfs_large_seen = times_.bfs_max_seen = times_.bfs_large_used = times_.bfs_large_used_max = 0;
    std::vector<uint32_t> p_sides, p_links;
    std::vector<uint8_t> p_bytes;
    {
        // (every side written is noted: exactly those go to the device afterwards)
        const FlagsPerSideLogged acc{FlagsPerSide{big_f2_.data(), plus_.data(), minus_.data()}, &p_sides};
        Commits<FlagsPerSideLogged> cm{acc, complex_size_, NoColours{}};
        Commits<FlagsPerSideLogged, ColourGate> cmc{acc, complex_size_, col_ ? st_.colour_gate() : ColourGate{}};
        size_t a = 0, b = 0;   // a over big_idx, b over the walked records (ascending candidate index both)
        auto commit = [&](const pf_bfs_record &r, const uint32_t *list) {
            if (r.n_seen > 4096) { times_.bfs_large++; times_.bfs_large_seen += r.n_seen; }
            if (r.n_seen > times_.bfs_max_seen) times_.bfs_max_seen = r.n_seen;
            if (!record_effective(r)) return;
            if (!cm.gate_open(r.entrance)) return;
            if (r.n_seen > 4096) { times_.bfs_large_used++; times_.bfs_large_used_max = std::max<uint64_t>(times_.bfs_large_used_max, r.n_seen); }
            if (col_) cmc.replay(r, list); else cm.replay(r, list);
        };
        double t_walked = 0, t_big = 0;
        uint64_t l_walked = 0, l_big = 0, e_walked = 0, e_big = 0;
        while (a < (size_t)n_big || b < (size_t)n_deferred) {
            const bool take_walked = a >= (size_t)n_big || (b < (size_t)n_deferred && deferred[b] < big_idx[a]);
            const auto tc = trace_find ? clk::now() : clk::time_point();
            if (take_walked) {
                if (trace_find && record_effective(walked[b]) && cm.gate_open(walked[b].entrance)) { l_walked += walked[b].n_list; ++e_walked; }
                commit(walked[b], xpool.data() + walked[b].list_off); ++b;
                if (trace_find) t_walked += since(tc);
            } else {
                if (trace_find && record_effective(big_rec[a]) && cm.gate_open(big_rec[a].entrance)) { l_big += big_rec[a].n_list; ++e_big; }
                commit(big_rec[a], big_pool.data() + big_rec[a].list_off); ++a;
                if (trace_find) t_big += since(tc);
            }
        }
        if (trace_find)
            fprintf(stderr, "[find]   replayed here: %zu walked records (%llu take effect, %llu list entries) %.3f ms, %llu records of large components (%llu, %llu) %.3f ms\n",
                    (size_t)n_deferred, (unsigned long long)e_walked, (unsigned long long)l_walked, t_walked * 1e3, (unsigned long long)n_big, (unsigned long long)e_big, (unsigned long long)l_big, t_big * 1e3);
    }
    tf("large components replayed");
    p_links.resize(p_sides.size());
    p_bytes.resize(p_sides.size());
    const unsigned pt = (unsigned)std::min<size_t>(threads_ ? threads_ : std::max<size_t>(thr, 1), 8);
    parallel_chunks(p_sides.size(), 1u << 14, pt, [&](size_t, size_t b, size_t e) {
        for (size_t i = b; i < e; ++i) {
            const uint32_t s = p_sides[i];
            p_links[i] = (s & 1) ? minus_[s >> 1] : plus_[s >> 1];
            p_bytes[i] = big_f2_[s];
        }
    });
    tf("patch gathered");
    // (the host arrays go back to all-zero for the next pass beside the device's patch: that reads the gathered copies only)
    std::thread undo([&] {
        parallel_chunks(p_sides.size(), 1u << 14, pt, [&](size_t, size_t b, size_t e) {
            for (size_t i = b; i < e; ++i) {   // (a side logged twice may be zeroed from two threads: relaxed atomic stores of the same value)
                const uint32_t s = p_sides[i];
                __atomic_store_n(&((s & 1) ? minus_ : plus_)[s >> 1], 0u, __ATOMIC_RELAXED);
                __atomic_store_n(&big_f2_[s], (uint8_t)0, __ATOMIC_RELAXED);
            }
        });
    });
    st = pf_replay_finish(ctx_, p_sides.data(), p_links.data(), p_bytes.data(), p_sides.size());
    undo.join();
    if (st != PF_OK) return fail(st, std::string("CDBG::findSuperBubble(): ") + pf_last_error(ctx_));
    if (trace_find) fprintf(stderr, "[find] %zu sides patched on the device %.2f ms\n", p_sides.size(), since(t_all) * 1e3);
    state_host_stale_ = true;
    state_on_device_ = true;
    tf("large components committed here");
    times_.bfs_device_s = since(t_all);
    times_.candidates = n_rec;
    times_.bfs_deferred = n_deferred;
    times_.host_commit_records = n_big + n_deferred;
    times_.host_walk_vertices = 0;
    for (const auto &w : walked) times_.host_walk_vertices += w.n_seen;
    times_.replay_s = 0;
    if (!quiet_) {
        printf(mt_format_ ? "%s::findSuperBubble(): Finding superbubbles Cpu time : %gs\n" : "%s::findSuperBubble():  Cpu time : %gs\n", tag_,
               (double)(clock() - c0) / CLOCKS_PER_SEC);
        printf(mt_format_ ? "%s::findSuperBubble(): Finding superbubbles Real time : %gs\n" : "%s::findSuperBubble():  Real time : %gs\n", tag_,
               since(t_all));
    }
    return finish_find(outpre, thr, t_all, write_sb_);
}

// ---- findSuperBubble (reference src/CDBG.cpp:178-252) -------------------------------------
int CDBG::findSuperBubble_multithread_ptr(const std::string &outpre, const size_t &thr) {
    if (status_) return status_;
    join_prealloc();
    if (join_pending_write()) return status_;
    if (!quiet_) printf("%s::findSuperBubble(): Finding superbubbles\n", tag_);
    if (write_files_ && ensure_dir()) return status_;
    if (commits_on_device(thr)) {
        if (!quiet_) printf("%s::findSuperBubble(): There are %u unitigs \n", tag_, g_.n());
        return find_superbubbles_device(outpre, thr);
    }
    state_host_stale_ = false;
    const auto t_all = clk::now();
    clock_t c0 = clock();
    const uint32_t N = g_.n();
    if (!quiet_) printf("%s::findSuperBubble(): There are %u unitigs \n", tag_, N);
    out_bytes_ = 0;
    state_on_device_ = false;
    const unsigned rt = replay_threads(thr);
    const unsigned zt = std::max<unsigned>(rt, (unsigned)std::min<size_t>(threads_ ? threads_ : std::max<size_t>(thr, 1), 8));
    parallel_chunks(N, 1u << 19, zt, [&](size_t, size_t b, size_t e) {
        memset(flags_.data() + b, 0, e - b);
        memset(plus_.data() + b, 0, (e - b) * 4);
        memset(minus_.data() + b, 0, (e - b) * 4);
    });
    if (rt) par_.begin(N, plus_.data(), minus_.data(), complex_size_, rt);

    // Every candidate entrance is traversed on the device, one wavefront each.  The unitig range is cut into slices:
    // a helper thread (the only one issuing device calls here) runs K-BFS slice by slice and, after the last one, the
    // coverage kernel PloidyEstimation starts with -- while this thread replays the records of the finished slices in
    // the reference's visiting order, with its `partner == NULL` gate (src/CDBG.cpp:206, 211): records come in
    // ascending oriented-vertex order = unitig order, '+' before '-'.
    // The unitig range is cut into slices so that the replay of slice i overlaps the traversal of slice i + 1.  With the long
    // traversals on host cores a slice's own are walked in a few milliseconds, side by side; only with the device's third tier
    // (one wavefront per giant traversal, run slice after slice) is the first pass over a graph kept in one piece.
    constexpr int kMaxSlices = 4;
    // (with the commits spread over host threads a slice's replay takes a millisecond: slicing only pays for the sequential replay)
    const int kSlices = replay_threads(thr) ? 1 : (third_tier_on_host_ || (find_passes_ > 0 && times_.bfs_large == 0)) ? kMaxSlices : 1;
    ++find_passes_;
    uint32_t s_u0[kMaxSlices + 1];
    uint64_t s_cand[kMaxSlices], s_rec0[kMaxSlices + 1], s_pool0[kMaxSlices + 1], s_nrec[kMaxSlices], s_used[kMaxSlices];
    s_rec0[0] = s_pool0[0] = 0;
    for (int i = 0; i <= kSlices; ++i) s_u0[i] = (uint32_t)((uint64_t)N * i / kSlices);
    for (int i = 0; i < kSlices; ++i) {
        int st0 = pf_count_candidates(ctx_, s_u0[i], s_u0[i + 1], &s_cand[i]);
        if (st0 != PF_OK) return fail(st0, pf_last_error(ctx_));
        s_rec0[i + 1] = s_rec0[i] + std::max<uint64_t>(s_cand[i], 1);
        // the pool guess leaves room for the per-wave chunk slack
        s_pool0[i + 1] = s_pool0[i] + s_cand[i] * 6 + (1u << 20);
        s_nrec[i] = s_used[i] = 0;
    }
    // pinned, reused from pass to pass
    const bool trace_find = getenv("PF_TRACE_FIND") != nullptr;
    auto tf = [&](const char *what) { if (trace_find) fprintf(stderr, "[find] %-28s %.2f ms\n", what, since(t_all) * 1e3); };
    tf("candidates counted");
    bx_.bfs_rec.ensure(ctx_, s_rec0[kSlices]);
    bx_.bfs_pool.ensure(ctx_, s_pool0[kSlices]);
    if (rt) bx_.bfs_order.ensure(ctx_, s_rec0[kSlices]);
    uint32_t class_off[kMaxSlices][kReplayClasses + 1];
    tf("pinned record buffers");
    pf_bfs_record *rec = bx_.bfs_rec.p;
    // a slice whose pool guess was too small gets a buffer of its own
    std::vector<std::unique_ptr<PinnedBuf<uint32_t>>> own_pool(kSlices);
    const uint32_t *slice_pool[kMaxSlices];
    std::mutex mu;
    std::condition_variable cv;
    int done = 0, dev_st = PF_OK;
    std::string dev_err, walk_err;
    double bfs_s = 0;
    const bool prefetch_cov = overlap_output_;  // the same switch: work of the next call started behind the caller's back
    cov_ready_ = false;
    static_assert(kMaxSlices <= 4, "huge_lists_ holds four slices");
    for (auto &hl : huge_lists_) hl.clear();
    // host walkers of the long traversals: each keeps 4 bytes of state per unitig, at most ~4 GiB of it in total
    const unsigned walk_threads = (unsigned)std::max<uint64_t>(1, std::min<uint64_t>(threads_ ? threads_ : std::max<size_t>(thr, 1), (4ull << 30) / (4ull * std::max<uint32_t>(N, 1))));
    std::atomic<uint64_t> n_deferred_total{0};
    std::thread device([&] {
        try {
        std::vector<uint32_t> &deferred = deferred_;   // keeps its size from pass to pass
        for (int i = 0; i < kSlices; ++i) {
            const auto tb = clk::now();
            uint32_t *pl = bx_.bfs_pool.p + s_pool0[i];
            uint64_t cap = s_pool0[i + 1] - s_pool0[i];
            uint64_t n_deferred = 0;
            if (third_tier_on_host_ && deferred.size() < 4096) deferred.resize(4096);
            std::vector<uint32_t> &deferred_ent = deferred_ent_;
            if (deferred_ent.size() < deferred.size()) deferred_ent.resize(deferred.size());
            int st1;
            for (;;) {
                if (third_tier_on_host_) {
                    // the records travel to the host while the long traversals are walked (pf_bfs_candidates_end below)
                    st1 = pf_bfs_candidates_begin(ctx_, s_u0[i], s_u0[i + 1], rec + s_rec0[i], s_rec0[i + 1] - s_rec0[i], pl, cap, &s_nrec[i], &s_used[i],
                                                  deferred.data(), deferred_ent.data(), deferred.size(), &n_deferred);
                    if (st1 == PF_ERR_OVERFLOW && n_deferred > deferred.size()) {
                        deferred.resize(n_deferred + n_deferred / 4);
                        deferred_ent.resize(deferred.size());
                        continue;
                    }
                } else
                st1 = pf_bfs_candidates(ctx_, s_u0[i], s_u0[i + 1], rec + s_rec0[i], s_rec0[i + 1] - s_rec0[i], pl, cap, &s_nrec[i], &s_used[i]);
                if (st1 == PF_ERR_OVERFLOW && s_used[i] > cap) {
                    own_pool[i] = std::make_unique<PinnedBuf<uint32_t>>();
                    own_pool[i]->ensure(ctx_, s_used[i] + s_used[i] / 8);
                    pl = own_pool[i]->p;
                    cap = own_pool[i]->cap;
                    continue;
                }
                break;
            }
            n_deferred_total += n_deferred;
            if (getenv("PF_TRACE_BFS")) fprintf(stderr, "[bfs] device tiers of slice %d: %.2f ms, %llu candidates left for the third tier\n", i, since(tb) * 1e3, (unsigned long long)n_deferred);
            std::vector<pf_bfs_record> walked((size_t)n_deferred);
            std::thread walk;
            if (st1 == PF_OK && n_deferred) {
                // third tier: one host thread per giant traversal, side by side; lists go to huge_lists_
                huge_lists_[i].assign((size_t)n_deferred, std::vector<uint32_t>());
                walk = std::thread([&, i] {
                    try {
                    parallel_chunks((size_t)n_deferred, 1, walk_threads, [&](size_t d, size_t, size_t) {
                        std::unique_ptr<HugeWalker> w;
                        {
                            std::lock_guard<std::mutex> lk(walkers_mu_);
                            if (!walkers_.empty()) { w = std::move(walkers_.back()); walkers_.pop_back(); }
                        }
                        if (!w) w = std::make_unique<HugeWalker>();
                        pf_bfs_record &r = walked[d];
                        memset(&r, 0, sizeof r);
                        const uint32_t s = deferred_ent[d];
                        r.entrance = s;
                        r.exit = 0xFFFFFFFFu;
                        const auto tw = clk::now();
                        const std::vector<uint32_t> &list = w->walk(succ_.data(), pred_.data(), N, s, r);
                        if (getenv("PF_TRACE_BFS")) fprintf(stderr, "[bfs] host walk from %u: %u vertices, outcome %d, %.2f ms\n", s, r.n_seen, (int)r.outcome, since(tw) * 1e3);
                        huge_lists_[i][d].assign(list.begin(), list.begin() + r.n_list);
                        r.list_off = d;
                        r.pad_ = 1;
                        {
                            std::lock_guard<std::mutex> lk(walkers_mu_);
                            walkers_.push_back(std::move(w));
                        }
                    });
                    } catch (const std::exception &e) {
                        std::lock_guard<std::mutex> lk(mu);
                        walk_err = e.what();
                    }
                });
            }
            // K-CC for the records that are on the device, beside the walkers and the copy
            if (st1 == PF_OK && rt) st1 = pf_side_components(ctx_, i == 0, nullptr, s_nrec[i], nullptr, 0, nullptr, 0, nullptr, 0);
            if (walk.joinable()) walk.join();
            {
                std::lock_guard<std::mutex> lk(mu);
                if (!walk_err.empty()) throw std::runtime_error("walk of a long traversal: " + walk_err);
            }
            if (third_tier_on_host_) {
                const int ste = pf_bfs_candidates_end(ctx_);
                if (st1 == PF_OK) st1 = ste;
            }
            if (st1 == PF_OK)
                for (uint64_t d = 0; d < n_deferred; ++d) rec[s_rec0[i] + deferred[d]] = walked[(size_t)d];
            const auto t_cc = clk::now();
            if (trace_find) fprintf(stderr, "[find]   slice %d traversed (%llu records, %llu walked on host cores) %.2f ms (+%.2f)\n", i, (unsigned long long)s_nrec[i],
                                    (unsigned long long)n_deferred, since(t_all) * 1e3, since(tb) * 1e3);
            if (st1 == PF_OK && rt) {
                // K-CC: the slice's records join the components (they are still on the device; the traversals walked on the host
                // add their footprints from here), then the slice's commit order by component class
                std::vector<pf_bfs_record> xrec;
                std::vector<uint32_t> xpool;
                const pf_bfs_record *srec = rec + s_rec0[i];
                for (uint64_t d = 0; d < n_deferred; ++d) {
                    pf_bfs_record r = srec[deferred[d]];
                    const std::vector<uint32_t> &l = huge_lists_[i][(size_t)d];
                    r.list_off = xpool.size();
                    r.pad_ = 0;
                    xpool.insert(xpool.end(), l.begin(), l.begin() + r.n_list);
                    xrec.push_back(r);
                }
                if (!xrec.empty()) st1 = pf_side_components(ctx_, 0, nullptr, s_nrec[i], nullptr, 0, xrec.data(), xrec.size(), xpool.data(), xpool.size());
                if (st1 == PF_OK) st1 = pf_replay_order(ctx_, kReplayClasses, bx_.bfs_order.p + s_rec0[i], class_off[i], nullptr);
                if (trace_find) fprintf(stderr, "[find]   slice %d components + order %.2f ms (+%.2f)\n", i, since(t_all) * 1e3, since(t_cc) * 1e3);
            }
            bfs_s += since(tb);
            {
                std::lock_guard<std::mutex> lk(mu);
                slice_pool[i] = pl;
                if (st1 != PF_OK) { dev_st = st1; dev_err = pf_last_error(ctx_); }
                done = i + 1;
            }
            cv.notify_all();
            if (st1 != PF_OK) return;
        }
        if (prefetch_cov) cov_ready_ = launch_coverage() == PF_OK;
        } catch (const std::exception &e) {   // (bad_alloc of a list or a buffer: reported like a device error, not std::terminate)
            (void)pf_bfs_candidates_end(ctx_);   // (a copy of records may still be in flight)
            {
                std::lock_guard<std::mutex> lk(mu);
                dev_st = PF_ERR_HIP;
                dev_err = std::string("host layer: ") + e.what();
            }
            cv.notify_all();
        }
    });
    times_.bfs_large = times_.bfs_large_seen = times_.bfs_max_seen = 0;  // (kSlices above looked at the previous pass)
    times_.bfs_deferred = 0;
    times_.bfs_large_used = times_.bfs_large_used_max = 0;
    uint64_t n_rec_total = 0;
    double replay_s = 0;
    int st = PF_OK;
    for (int sl = 0; sl < kSlices; ++sl) {
        {
            std::unique_lock<std::mutex> lk(mu);
            cv.wait(lk, [&] { return done > sl || dev_st != PF_OK; });
            if (done <= sl || (dev_st != PF_OK && done == sl + 1)) { st = dev_st; break; }
        }
        const auto tr = clk::now();
        const pf_bfs_record *srec = rec + s_rec0[sl];
        const uint64_t n_rec = s_nrec[sl];
        const uint32_t *pool = slice_pool[sl];
        n_rec_total += n_rec;
        if (rt) {
            ReplayStats rs;
            auto list_of = [&](const pf_bfs_record &r) { return r.pad_ ? huge_lists_[sl][r.list_off].data() : pool + r.list_off; };
            par_.run(srec, list_of, bx_.bfs_order.p + s_rec0[sl], class_off[sl], kReplayClasses, rt, rs);
            if (trace_find) fprintf(stderr, "[find]   slice %d replayed on %u threads %.2f ms (+%.2f)\n", sl, rt, since(t_all) * 1e3, since(tr) * 1e3);
            times_.bfs_large += rs.large; times_.bfs_large_seen += rs.large_seen; times_.bfs_large_used += rs.large_used;
            times_.bfs_max_seen = std::max<uint64_t>(times_.bfs_max_seen, rs.max_seen);
            times_.bfs_large_used_max = std::max<uint64_t>(times_.bfs_large_used_max, rs.large_used_max);
        } else
        for (uint64_t i = 0; i < n_rec; ++i) {
            // the commits chase per-unitig state at random: pull the state of a record a few iterations ahead into cache
            if (i + 12 < n_rec) {
                const pf_bfs_record &nx = srec[i + 12];
                if (!nx.pad_) __builtin_prefetch(pool + nx.list_off);
                __builtin_prefetch(&plus_[nx.entrance >> 1]);
                __builtin_prefetch(&minus_[nx.entrance >> 1]);
            }
            if (i + 6 < n_rec) {
                const pf_bfs_record &nx = srec[i + 6];
                const uint32_t *l = nx.pad_ ? huge_lists_[sl][nx.list_off].data() : pool + nx.list_off;
                const uint32_t nl = nx.n_list < 6 ? nx.n_list : 6;
                for (uint32_t q = 0; q < nl; ++q) {
                    const uint32_t w = l[q] >> 1;
                    __builtin_prefetch(&flags_[w]);
                    __builtin_prefetch(&plus_[w]);
                    __builtin_prefetch(&minus_[w]);
                }
            }
            const pf_bfs_record &r = srec[i];
            if (r.n_seen > 4096) { times_.bfs_large++; times_.bfs_large_seen += r.n_seen; }
            if (r.n_seen > times_.bfs_max_seen) times_.bfs_max_seen = r.n_seen;
            const uint32_t u = r.entrance >> 1;
            if ((plus_of(r.entrance) ? plus_[u] : minus_[u]) != 0) continue;
            if (r.n_seen > 4096) { times_.bfs_large_used++; if (r.n_seen > times_.bfs_large_used_max) times_.bfs_large_used_max = r.n_seen; }
            replay(r, r.pad_ ? huge_lists_[sl][r.list_off].data() : pool + r.list_off);
        }
        replay_s += since(tr);
    }
    if (rt && st == PF_OK) {
        const auto tr = clk::now();
        par_.finish(flags_.data(), rt);
        replay_s += since(tr);
    }
    tf("replay done");
    device.join();
    tf("device thread joined");
    if (st != PF_OK || dev_st != PF_OK) return fail(dev_st != PF_OK ? dev_st : st, std::string("CDBG::findSuperBubble(): ") + dev_err);
    times_.bfs_device_s = bfs_s;
    times_.candidates = n_rec_total;
    times_.bfs_deferred = n_deferred_total.load();
    times_.replay_s = replay_s;
    if (!quiet_) {
        // (the reference words these two lines differently in its threaded function, src/CDBG.cpp:1783-1786)
        printf(mt_format_ ? "%s::findSuperBubble(): Finding superbubbles Cpu time : %gs\n" : "%s::findSuperBubble():  Cpu time : %gs\n", tag_,
               (double)(clock() - c0) / CLOCKS_PER_SEC);
        printf(mt_format_ ? "%s::findSuperBubble(): Finding superbubbles Real time : %gs\n" : "%s::findSuperBubble():  Real time : %gs\n", tag_,
               since(t_all));
    }
    return finish_find(outpre, thr, t_all, write_sb_);
}

// second half of findSuperBubble (reference src/CDBG.cpp:222-252): the rows of <outpre>_super_bubble.txt from the final state
int CDBG::finish_find(const std::string &outpre, const size_t &thr, clk_time t_all, bool write_file) {
    (void)thr;
    const bool trace_find = getenv("PF_TRACE_FIND") != nullptr;
    auto tf = [&](const char *what) { if (trace_find) fprintf(stderr, "[find] %-28s %.2f ms\n", what, since(t_all) * 1e3); };
    auto t0 = clk::now();
    // The state goes to the device once -- PloidyEstimation's scan reads it there as well -- and the rows of super_bubble.txt
    // (one per open endpoint side in unitig order, numbered by a prefix count) are formatted there; the text comes back and is
    // written by a helper thread, behind the caller's back when overlap_output is on.
    int st = state_on_device_ ? PF_OK : pf_call_set_state(ctx_, flags_.data(), plus_.data(), minus_.data());
    uint64_t nb = 0, len = 0;
    if (st == PF_OK) st = pf_superbubble_rows(ctx_, col_ != nullptr ? 1 : 0, &nb, &len);
    if (st != PF_OK) return fail(st, std::string(tag_) + "::findSuperBubble(): " + pf_last_error(ctx_));
    state_on_device_ = true;
    tf("super_bubble rows on the device");
    static const char kHeader[] = "BubbleId\tEntrance\tStrand\tExit\tisSimple\tisComplex\n";
    n_super_bubble_ = nb;
    times_.bubbles_out = nb;
    out_bytes_ += len + (sizeof(kHeader) - 1);
    if (write_file && write_files_) {
        join_pending_write();
        sb_text_.ensure(ctx_, std::max<uint64_t>(len, 1));
        const unsigned T = threads_ ? threads_ : 1;
        auto job = [this, name = outpre + "_super_bubble.txt", len, T, trace_find]() -> int {
            const auto tj = clk::now();
            if (pf_superbubble_fetch(ctx_, sb_text_.p, len) != PF_OK) return 1;
            const double t_fetch = since(tj);
            MappedOut &mo = out_maps_[PF_CALL_STREAMS];   // (the slot after the ten streams of PloidyEstimation)
            if (mo.open_for(outdir_ + "/" + name)) return 1;
            const uint64_t hl = sizeof(kHeader) - 1;
            int rc = mo.write(0, kHeader, hl, 1);
            rc |= mo.write(hl, sb_text_.p, len, T);
            rc |= mo.finish(hl + len);
            if (trace_find) fprintf(stderr, "[find]   super_bubble.txt: fetched after %.2f ms, in the file after %.2f ms (%.1f MB)\n", t_fetch * 1e3, since(tj) * 1e3, len / 1e6);
            return rc;
        };
        if (overlap_output_) {
            // written behind the caller's back while PloidyEstimation starts; joined there (or by the next use of the file)
            pending_write_ = std::thread([this, job] { pending_rc_ = job(); });
        } else if (job()) {
            return fail(PF_ERR_ARG, "CDBG:: Open " + outpre + "_super_bubble.txt file error");
        }
    }
    times_.bubble_write_s = since(t0);
    times_.find_total_s = since(t_all);
    tf("super_bubble rows done");
    if (!quiet_) printf("%s::findSuperBubble(): %llu  SuperBubbles Found\n", tag_, (unsigned long long)nb);
    return 0;
}

// ---- one graph over several GPUs: a rank traverses the candidate entrances of its unitig range only ----
// K-BFS (LDS tier on the device, the long traversals on host cores) for the entrances on unitigs [u0, u1); the records and their
// vertex lists are left self-contained in shard_rec_ / shard_pool_ (list_off relative to shard_pool_) for the exchange.
int CDBG::find_shard(uint32_t u0, uint32_t u1) {
    if (status_) return status_;
    join_prealloc();
    const uint32_t N = g_.n();
    if (u0 > u1 || u1 > N) return fail(PF_ERR_ARG, "CDBG::find_shard(): range outside the graph");
    const auto tb = clk::now();
    uint64_t n_cand = 0;
    int st = pf_count_candidates(ctx_, u0, u1, &n_cand);
    if (st != PF_OK) return fail(st, pf_last_error(ctx_));
    bx_.bfs_rec.ensure(ctx_, std::max<uint64_t>(n_cand, 1));
    bx_.bfs_pool.ensure(ctx_, n_cand * 6 + (1u << 20));
    std::vector<uint32_t> &deferred = deferred_;
    if (deferred.size() < 4096) deferred.resize(4096);
    uint64_t n_rec = 0, used = 0, n_deferred = 0;
    for (;;) {
        st = pf_bfs_candidates_split(ctx_, u0, u1, bx_.bfs_rec.p, bx_.bfs_rec.cap, bx_.bfs_pool.p, bx_.bfs_pool.cap, &n_rec, &used, deferred.data(),
                                     deferred.size(), &n_deferred);
        if (st == PF_ERR_OVERFLOW && n_deferred > deferred.size()) { deferred.resize(n_deferred + n_deferred / 4); continue; }
        if (st == PF_ERR_OVERFLOW && used > bx_.bfs_pool.cap) { bx_.bfs_pool.ensure(ctx_, used + used / 8); continue; }
        break;
    }
    if (st != PF_OK) return fail(st, std::string("CDBG::findSuperBubble(): ") + pf_last_error(ctx_));
    shard_rec_.assign(bx_.bfs_rec.p, bx_.bfs_rec.p + n_rec);
    shard_pool_.assign(bx_.bfs_pool.p, bx_.bfs_pool.p + used);
    if (n_deferred) {
        const unsigned walk_threads = (unsigned)std::max<uint64_t>(1, std::min<uint64_t>(threads_ ? threads_ : 1, (4ull << 30) / (4ull * std::max<uint32_t>(N, 1))));
        std::vector<std::vector<uint32_t>> lists((size_t)n_deferred);
        parallel_chunks((size_t)n_deferred, 1, walk_threads, [&](size_t d, size_t, size_t) {
            std::unique_ptr<HugeWalker> w;
            {
                std::lock_guard<std::mutex> lk(walkers_mu_);
                if (!walkers_.empty()) { w = std::move(walkers_.back()); walkers_.pop_back(); }
            }
            if (!w) w = std::make_unique<HugeWalker>();
            pf_bfs_record &r = shard_rec_[deferred[d]];
            const std::vector<uint32_t> &list = w->walk(succ_.data(), pred_.data(), N, r.entrance, r);
            lists[d].assign(list.begin(), list.begin() + r.n_list);
            std::lock_guard<std::mutex> lk(walkers_mu_);
            walkers_.push_back(std::move(w));
        });
        for (size_t d = 0; d < (size_t)n_deferred; ++d) {
            pf_bfs_record &r = shard_rec_[deferred[d]];
            r.list_off = shard_pool_.size();
            r.pad_ = 0;
            shard_pool_.insert(shard_pool_.end(), lists[d].begin(), lists[d].end());
        }
    }
    times_.bfs_device_s = since(tb);
    times_.bfs_deferred = n_deferred;
    return 0;
}

// The commit replay over the records of all shards, in shard order = entrance order (reference visiting order, src/CDBG.cpp:
// 206-214), then the rows of super_bubble.txt; every rank ends with the same state.
int CDBG::find_replay(const std::string &outpre, uint32_t n_shards, const pf_bfs_record *const *records, const uint64_t *n_records,
                      const uint32_t *const *pools, bool write_file, const uint64_t *pool_lens, const pf_bfs_record *const *dev_records,
                      const uint32_t *const *dev_pools) {
    if (status_) return status_;
    if (join_pending_write()) return status_;
    if (write_file && write_files_ && ensure_dir()) return status_;
    const auto t_all = clk::now();
    out_bytes_ = 0;
    state_on_device_ = false;
    state_host_stale_ = false;
    std::fill(flags_.begin(), flags_.end(), 0);
    std::fill(plus_.begin(), plus_.end(), 0);
    std::fill(minus_.begin(), minus_.end(), 0);
    cov_ready_ = false;
    times_.bfs_large = times_.bfs_large_seen = times_.bfs_max_seen = times_.bfs_large_used = times_.bfs_large_used_max = 0;
    const auto tr = clk::now();
    uint64_t total = 0;
    uint32_t last = 0;
    // the parallel replay needs the lengths of the pools (the device refuses lists outside them)
    const unsigned rt = pool_lens ? replay_threads(threads_ ? threads_ : 1) : 0;
    if (rt) par_.begin(g_.n(), plus_.data(), minus_.data(), complex_size_, rt);
    std::vector<uint32_t> order;
    bool seen_any = false;
    // shards given as device memory only (the all-gather's output): their host copies come down into the pinned exchange buffers
    std::vector<const pf_bfs_record *> h_rec(n_shards);
    std::vector<const uint32_t *> h_pool(n_shards);
    {
        uint64_t need_rec = 0, need_pool = 0;
        for (uint32_t sh = 0; sh < n_shards; ++sh) {
            h_rec[sh] = records ? records[sh] : nullptr;
            h_pool[sh] = pools ? pools[sh] : nullptr;
            if (!h_rec[sh] || !h_pool[sh]) {
                if (!pool_lens || !dev_records || !dev_pools || !dev_records[sh] || !dev_pools[sh])
                    return fail(PF_ERR_ARG, "CDBG::find_replay(): a shard has neither host nor device arrays");
                need_rec += n_records[sh];
                need_pool += pool_lens[sh];
            }
        }
        if (need_rec) {
            bx_.bfs_rec.ensure(ctx_, need_rec + 1);
            bx_.bfs_pool.ensure(ctx_, need_pool + 1);
            uint64_t ar = 0, ap = 0;
            for (uint32_t sh = 0; sh < n_shards; ++sh) {
                if (h_rec[sh] && h_pool[sh]) continue;
                int st = pf_fetch(ctx_, bx_.bfs_rec.p + ar, dev_records[sh], n_records[sh] * sizeof(pf_bfs_record));
                if (st == PF_OK) st = pf_fetch(ctx_, bx_.bfs_pool.p + ap, dev_pools[sh], pool_lens[sh] * 4);
                if (st != PF_OK) return fail(st, std::string("CDBG::find_replay(): ") + pf_last_error(ctx_));
                h_rec[sh] = bx_.bfs_rec.p + ar;
                h_pool[sh] = bx_.bfs_pool.p + ap;
                ar += n_records[sh];
                ap += pool_lens[sh];
            }
        }
    }
    for (uint32_t sh = 0; sh < n_shards; ++sh) {
        const pf_bfs_record *rec = h_rec[sh];
        const uint32_t *pool = h_pool[sh];
        const uint64_t n = n_records[sh];
        total += n;
        if (rt) {
            for (uint64_t i = 0; i < n; ++i) {
                const pf_bfs_record &r = rec[i];
                if ((r.entrance >> 1) >= g_.n() || (seen_any && r.entrance < last)) return fail(PF_ERR_ARG, "CDBG::find_replay(): records out of order");
                last = r.entrance;
                seen_any = true;
            }
            const bool on_dev = dev_records && dev_pools && dev_records[sh] && dev_pools[sh];
            int st = pf_side_components(ctx_, sh == 0, on_dev ? dev_records[sh] : rec, n, on_dev ? dev_pools[sh] : pool, pool_lens[sh], nullptr, 0, nullptr, 0);
            uint32_t class_off[kReplayClasses + 1];
            order.resize(std::max<uint64_t>(n, 1));
            if (st == PF_OK) st = pf_replay_order(ctx_, kReplayClasses, order.data(), class_off, nullptr);
            if (st != PF_OK) return fail(st, std::string("CDBG::find_replay(): ") + pf_last_error(ctx_));
            ReplayStats rs;
            par_.run(rec, [&](const pf_bfs_record &r) { return pool + r.list_off; }, order.data(), class_off, kReplayClasses, rt, rs);
            times_.bfs_large += rs.large; times_.bfs_large_seen += rs.large_seen;
            times_.bfs_max_seen = std::max<uint64_t>(times_.bfs_max_seen, rs.max_seen);
            continue;
        }
        for (uint64_t i = 0; i < n; ++i) {
            if (i + 8 < n) {
                const pf_bfs_record &nx = rec[i + 8];
                __builtin_prefetch(pool + nx.list_off);
                __builtin_prefetch(&plus_[nx.entrance >> 1]);
                __builtin_prefetch(&minus_[nx.entrance >> 1]);
            }
            const pf_bfs_record &r = rec[i];
            if ((r.entrance >> 1) >= g_.n() || (total > 1 && r.entrance < last)) return fail(PF_ERR_ARG, "CDBG::find_replay(): records out of order");
            last = r.entrance;
            if (r.n_seen > 4096) { times_.bfs_large++; times_.bfs_large_seen += r.n_seen; }
            if (r.n_seen > times_.bfs_max_seen) times_.bfs_max_seen = r.n_seen;
            if (!st_.gate_open(r.entrance)) continue;
            replay(r, pool + r.list_off);
        }
    }
    if (rt) par_.finish(flags_.data(), rt);
    times_.replay_s = since(tr);
    times_.candidates = total;
    return finish_find(outpre, 1, t_all, write_file);
}

// K-COV (colored: K-COV-C) for all unitigs into the pinned result buffers.  A missing k-mer is not an error here: the
// single-sample path raises it only for the unitigs it really uses, the colored path never (src/CCDBG.cpp:113-117).
int CDBG::launch_coverage() {
    if (resident_path()) {  // the results stay on the device, where the scan of the calling pipeline reads them
        const int st = pf_call_coverage(ctx_);
        if (st != PF_OK) cov_err_ = pf_last_error(ctx_);
        return st;
    }
    const uint32_t N = g_.n();
    // a database without canonical counting (single-sample only) is read per orientation: [0, N) the unitigs as stored,
    // [N, 2N) their reverse complements (readCov(UnitigMap), src/CDBG.cpp:94-117)
    const uint32_t C = col_ ? col_->n_colors : (both_strands_ ? 1 : 2);
    bx_.cov_sum.ensure(ctx_, (size_t)N * C);
    bx_.cov_min.ensure(ctx_, (size_t)N * C);
    bx_.cov_miss.ensure(ctx_, (size_t)N * C);
    if (col_) bx_.cov_max.ensure(ctx_, (size_t)N * C);
    int st;
    if (col_) st = pf_unitig_cov_colored(ctx_, 0, N, bx_.cov_sum.p, bx_.cov_min.p, bx_.cov_max.p, bx_.cov_miss.p);
    else if (both_strands_) st = pf_unitig_cov(ctx_, 0, N, bx_.cov_sum.p, bx_.cov_min.p, bx_.cov_miss.p);
    else {
        st = pf_unitig_cov_exact(ctx_, 0, N, 0, bx_.cov_sum.p, bx_.cov_min.p, bx_.cov_miss.p);
        if (st == PF_OK || st == PF_ERR_MISSING_KMER) st = pf_unitig_cov_exact(ctx_, 0, N, 1, bx_.cov_sum.p + N, bx_.cov_min.p + N, bx_.cov_miss.p + N);
    }
    if (st != PF_OK && st != PF_ERR_MISSING_KMER) { cov_err_ = pf_last_error(ctx_); return st; }
    return PF_OK;
}

}  // namespace pfh
