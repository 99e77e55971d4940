#include <chrono>
#include <cstring>
#include <exception>
#include <memory>
#include <string>
#include <vector>

#include "pf_bfs_host.hpp"
#include "pf_cdbg.hpp"
#include "pf_trace.hpp"
#include "pf_filter.hpp"
#include "pf_gmm_model.hpp"
#include "pf_host_colors.hpp"
#include "pf_replay_par.hpp"
#include "ploidyfrost_host.h"

struct pfh_run {
    pfh::UnitigSet graph;
    std::unique_ptr<pfh::ColoredUnitigSet> cgraph;  // colored runs: graph + colour sets (then `graph` stays empty)
    std::unique_ptr<pfh::CDBG> cdbg;
    pfh::CCDBG *ccdbg = nullptr;                     // == cdbg.get() for colored runs
    const pfh::UnitigSet &g() const { return cgraph ? cgraph->graph : graph; }
    double z_M = 2, z_D = -1, z_G = -3;
    size_t z = 8;
    double load_s = 0, upload_s = 0;
    std::string err;
};

static std::string g_open_err;

// no C++ exception may cross the C boundary
template <class F>
static int guarded(pfh_run *r, F &&f) {
    r->err.clear();   // (a message of an earlier call must not outlive it: pfh_last_error prefers this string)
    try {
        return f();
    } catch (const std::exception &e) {
        r->err = std::string("ploidyfrost host layer: ") + e.what();
        return PF_ERR_ARG;
    } catch (...) {
        r->err = "ploidyfrost host layer: unknown exception";
        return PF_ERR_ARG;
    }
}


struct pfh_gmm {
    pfh::GmmModel model;
    std::string err;
    explicit pfh_gmm(int device) : model(device) {}
};
template <class F>
static int gmm_guard(pfh_gmm *m, F &&fn) {
    if (!m) return 1;
    m->err.clear();
    try {
        return fn();
    } catch (const std::exception &e) {
        m->err = std::string("ploidyfrost host layer: ") + e.what();
        return 1;
    }
}

extern "C" {

pfh_run *pfh_open(const char *gfa_path, const char *kmc_prefix, uint32_t complex_size, double match, double mismatch,
                  double gap, int device) {
    using clk = std::chrono::steady_clock;
    try {
    auto r = std::make_unique<pfh_run>();
    auto t0 = clk::now();
    // K-GFA: parsed and packed on the device by the CDBG constructor, where the numbering is finished as well (K-MINZ)
    static const bool host_gfa = [] { const char *e = getenv("PF_GFA"); return e && !strcmp(e, "host"); }();
    if (!(host_gfa ? r->graph.load_gfa(gfa_path, g_open_err, true) : r->graph.open_gfa(gfa_path, g_open_err))) return nullptr;
    r->load_s = std::chrono::duration<double>(clk::now() - t0).count();
    r->z = complex_size;
    r->z_M = match; r->z_D = mismatch; r->z_G = gap;
    t0 = clk::now();
    r->cdbg = std::make_unique<pfh::CDBG>(r->graph, r->z, r->z_M, r->z_D, r->z_G, kmc_prefix ? kmc_prefix : "", device, true);
    r->upload_s = std::chrono::duration<double>(clk::now() - t0).count();
    if (!r->cdbg->good()) { g_open_err = r->cdbg->error(); return nullptr; }
    return r.release();
    } catch (const std::exception &e) {
        g_open_err = std::string("ploidyfrost host layer: ") + e.what();
        return nullptr;
    }
}

pfh_run *pfh_open_colored(const char *gfa_path, const char *colors_path, const char *kmc_list_file, uint32_t complex_size,
                          double match, double mismatch, double gap, uint32_t threads, int device) {
    using clk = std::chrono::steady_clock;
    try {
        auto r = std::make_unique<pfh_run>();
        auto t0 = clk::now();
        r->cgraph = std::make_unique<pfh::ColoredUnitigSet>();
        if (!r->cgraph->read(gfa_path, colors_path, threads ? threads : 1, false)) { g_open_err = r->cgraph->err; return nullptr; }
        r->load_s = std::chrono::duration<double>(clk::now() - t0).count();
        r->z = complex_size;
        r->z_M = match; r->z_D = mismatch; r->z_G = gap;
        t0 = clk::now();
        auto cc = std::make_unique<pfh::CCDBG>(*r->cgraph, r->z, r->z_M, r->z_D, r->z_G, kmc_list_file ? kmc_list_file : "",
                                               (size_t)(threads ? threads : 1), device, true);
        r->upload_s = std::chrono::duration<double>(clk::now() - t0).count();
        if (!cc->good()) { g_open_err = cc->error(); return nullptr; }
        r->ccdbg = cc.get();
        r->cdbg = std::move(cc);
        return r.release();
    } catch (const std::exception &e) {
        g_open_err = std::string("ploidyfrost host layer: ") + e.what();
        return nullptr;
    }
}

uint32_t pfh_num_colors(const pfh_run *r) { return r->cgraph ? (uint32_t)r->cgraph->getNbColors() : 0; }

int pfh_ploidy_estimation_colored(pfh_run *r, const char *outpre, const int *lower, const int *upper, uint32_t n_colors) {
    if (!r->ccdbg) { r->err = "pfh_ploidy_estimation_colored: the run was not opened with pfh_open_colored"; return PF_ERR_ARG; }
    return guarded(r, [&] {
        std::vector<std::pair<int, int>> cut(n_colors);
        for (uint32_t c = 0; c < n_colors; ++c) cut[c] = {lower[c], upper[c]};
        return r->ccdbg->ploidyEstimation_multithread_ptr(outpre, cut, 1);
    });
}

void pfh_close(pfh_run *r) { delete r; }
const char *pfh_last_error(const pfh_run *r) {
    if (!r) return g_open_err.c_str();
    return !r->err.empty() ? r->err.c_str() : r->cdbg->error().c_str();
}
void pfh_set_output_dir(pfh_run *r, const char *dir) { r->cdbg->set_output_dir(dir); }
void pfh_set_write_files(pfh_run *r, int on) { r->cdbg->set_write_files(on != 0); }
void pfh_set_threads(pfh_run *r, uint32_t threads) { r->cdbg->set_threads(threads); }
void pfh_set_overlap_output(pfh_run *r, int on) { r->cdbg->set_overlap_output(on != 0); }
void pfh_set_third_tier_on_host(pfh_run *r, int on) { r->cdbg->set_third_tier_on_host(on != 0); }
int pfh_set_reference_threads(pfh_run *r, uint32_t n) {
    return guarded(r, [&] { return r->cdbg->set_reference_threads(n); });
}
void pfh_set_batch_bubbles(pfh_run *r, uint64_t n) { r->cdbg->set_batch_bubbles((size_t)n); }
void pfh_set_align_pieces(pfh_run *r, uint64_t n) { r->cdbg->set_align_pieces((size_t)n); }
int pfh_set_unitig_id(pfh_run *r, const char *outpre) {
    return guarded(r, [&] { return r->cdbg->setUnitigId(outpre, "", 1); });
}
int pfh_find_superbubbles(pfh_run *r, const char *outpre) {
    return guarded(r, [&] { return r->cdbg->findSuperBubble_multithread_ptr(outpre, 1); });
}
int pfh_ploidy_estimation(pfh_run *r, const char *outpre, int lower, int upper) {
    return guarded(r, [&] { return r->cdbg->ploidyEstimation_multithread_ptr(outpre, lower, upper, 1); });
}
void *pfh_device_ctx(pfh_run *r) { return r->cdbg->device(); }

// ---- one graph over several GPUs -------------------------------------------------------------------
int pfh_find_shard(pfh_run *r, uint32_t u0, uint32_t u1) {
    return guarded(r, [&] { return r->cdbg->find_shard(u0, u1); });
}
const pf_bfs_record *pfh_shard_records(const pfh_run *r, uint64_t *n) {
    if (n) *n = r->cdbg->shard_records().size();
    return r->cdbg->shard_records().data();
}
const uint32_t *pfh_shard_pool(const pfh_run *r, uint64_t *n) {
    if (n) *n = r->cdbg->shard_pool().size();
    return r->cdbg->shard_pool().data();
}
int pfh_find_replay(pfh_run *r, const char *outpre, uint32_t n_shards, const pf_bfs_record *const *records, const uint64_t *n_records,
                    const uint32_t *const *pools, int write_file, const uint64_t *pool_lens, const pf_bfs_record *const *dev_records,
                    const uint32_t *const *dev_pools) {
    return guarded(r, [&] { return r->cdbg->find_replay(outpre, n_shards, records, n_records, pools, write_file != 0, pool_lens, dev_records, dev_pools); });
}
void pfh_set_replay_threads(pfh_run *r, int threads) { r->cdbg->set_replay_threads(threads); }
void pfh_set_write_super_bubble(pfh_run *r, int on) { r->cdbg->set_write_super_bubble(on != 0); }
int pfh_ploidy_select(pfh_run *r, int lower, int upper, uint64_t *n_bubbles) {
    return guarded(r, [&] { uint64_t n = 0; const int rc = r->cdbg->ploidy_select(lower, upper, n); if (n_bubbles) *n_bubbles = n; return rc; });
}
int pfh_ploidy_select_colored(pfh_run *r, const int *lower, const int *upper, int n_cutoffs, uint64_t *n_bubbles) {
    return guarded(r, [&] {
        std::vector<std::pair<int, int>> cut;
        for (int c = 0; c < n_cutoffs; ++c) cut.push_back({lower[c], upper[c]});
        uint64_t n = 0;
        const int rc = r->cdbg->ploidy_select(cut, n);
        if (n_bubbles) *n_bubbles = n;
        return rc;
    });
}
int pfh_ploidy_align(pfh_run *r, uint64_t t0, uint64_t t1, uint64_t *n_called) {
    return guarded(r, [&] { uint64_t n = 0; const int rc = r->cdbg->ploidy_align(t0, t1, n); if (n_called) *n_called = n; return rc; });
}
int pfh_ploidy_text(pfh_run *r, uint64_t var_count_base, uint64_t sizes[10], uint64_t counters[8]) {
    return guarded(r, [&] { return r->cdbg->ploidy_text(var_count_base, sizes, counters); });
}
int pfh_ploidy_write(pfh_run *r, const char *outpre, const uint64_t offsets[10], const uint64_t totals[10], int truncate) {
    return guarded(r, [&] { return r->cdbg->ploidy_write(outpre, offsets, totals, truncate != 0); });
}

int pfh_filter(int argc, char **argv, int multi) { return pfh::filter_main(argc, argv, multi != 0); }

uint64_t pfh_r_format_double(double x, char *out, uint64_t cap) {
    const std::string s = pfh::r_format_double(x);
    if (out && cap) {
        const uint64_t n = std::min<uint64_t>(cap - 1, s.size());
        memcpy(out, s.data(), n);
        out[n] = 0;
    }
    return s.size();
}

uint64_t pfh_load_trace(char *out, uint64_t cap, int reset) {
    pfh::LoadLog &l = pfh::LoadLog::get();
    std::lock_guard<std::mutex> lk(l.mu);
    std::string text;
    char num[32];
    for (auto &st : l.steps) {
        snprintf(num, sizeof num, "\t%.6f\n", st.second);
        text += st.first;
        text += num;
    }
    if (out && cap) {
        const uint64_t n = std::min<uint64_t>(cap - 1, text.size());
        memcpy(out, text.data(), n);
        out[n] = 0;
    }
    if (reset) l.steps.clear();
    return text.size();
}

void pfh_get_times(const pfh_run *r, pfh_times *o) {
    memset(o, 0, sizeof(*o));
    const pfh::PhaseTimes &t = r->cdbg->times();
    o->load_s = r->load_s; o->upload_s = r->upload_s;
    o->bfs_device_s = t.bfs_device_s; o->replay_s = t.replay_s; o->bubble_write_s = t.bubble_write_s; o->find_total_s = t.find_total_s;
    o->cov_device_s = t.cov_device_s; o->tasks_s = t.tasks_s; o->align_s = t.align_s; o->sites_s = t.sites_s;
    o->format_s = t.format_s; o->write_s = t.write_s; o->ploidy_total_s = t.ploidy_total_s;
    o->unitigs = r->g().n(); o->kmers = r->g().n_kmers; o->candidates = t.candidates; o->superbubbles = r->cdbg->n_superbubbles();
    o->tasks = t.tasks; o->align_jobs = t.align_jobs; o->site_strings = t.site_strings; o->output_bytes = r->cdbg->output_bytes();
    for (int a = 0; a < 4; ++a) o->allele[a] = r->cdbg->allele_sites(a + 2);
    o->core_cov = r->cdbg->core_cov(); o->core_num = r->cdbg->core_num();
    o->scan_s = t.scan_s; o->scan_serial_s = t.scan_serial_s;
    o->bfs_large = t.bfs_large; o->bfs_max_seen = t.bfs_max_seen;
    o->bfs_deferred = t.bfs_deferred;
    o->host_commit_records = t.host_commit_records; o->host_walk_vertices = t.host_walk_vertices;
    o->snp_jobs = t.snp_jobs; o->pair_jobs = t.pair_jobs; o->wave_jobs = t.wave_jobs; o->stack_jobs = t.stack_jobs;
}

const char *pfh_last_allele_frequency(const pfh_run *r, uint64_t *len) {
    const std::string &s = r->cdbg->last_allele_frequency();
    if (len) *len = s.size();
    return s.data();
}

void pfh_state(const pfh_run *r, uint8_t *flags, uint32_t *plus, uint32_t *minus) {
    const size_t N = r->g().n();
    if (flags) memcpy(flags, r->cdbg->state_flags().data(), N);
    if (plus) memcpy(plus, r->cdbg->state_plus().data(), N * 4);
    if (minus) memcpy(minus, r->cdbg->state_minus().data(), N * 4);
}

// ---- the commit replay on a bare state (no device) ----
}  // extern "C"
struct pfh_replay {
    pfh::UnitigState st;
    uint32_t last = 0;
    bool any = false;
    // pfh_replay_apply_parallel
    bool par = false;
    pfh::SideComponents cc;
    pfh::ParallelReplay pr;
};
extern "C" {
pfh_replay *pfh_replay_open(uint32_t n_unitigs, uint32_t complex_size) {
    try {
        auto h = std::make_unique<pfh_replay>();
        h->st.reset(n_unitigs);
        h->st.complex_size = complex_size;
        return h.release();
    } catch (const std::exception &e) {
        g_open_err = std::string("ploidyfrost host layer: ") + e.what();
        return nullptr;
    }
}
void pfh_replay_close(pfh_replay *h) { delete h; }
int pfh_replay_apply(pfh_replay *h, const pf_bfs_record *records, uint64_t n_records, const uint32_t *pool) {
    if (!h || (n_records && (!records || !pool))) return 1;
    for (uint64_t i = 0; i < n_records; ++i) {
        const pf_bfs_record &r = records[i];
        if ((r.entrance >> 1) >= h->st.flags.size() || (h->any && r.entrance < h->last)) return 2;  // shards must arrive in order
        h->last = r.entrance;
        h->any = true;
        if (!h->st.gate_open(r.entrance)) continue;
        h->st.replay(r, pool + r.list_off);
    }
    return 0;
}
// The same shard through the parallel replay (pf_replay_par.hpp): components on the host, `threads` workers.  One handle takes
// either this call or pfh_replay_apply, not both.
int pfh_replay_apply_parallel(pfh_replay *h, const pf_bfs_record *records, uint64_t n_records, const uint32_t *pool, uint32_t threads) {
    if (!h || (n_records && (!records || !pool))) return 1;
    if (h->any && !h->par) return 3;
    const uint32_t N = (uint32_t)h->st.flags.size();
    for (uint64_t i = 0; i < n_records; ++i) {
        const pf_bfs_record &r = records[i];
        if ((r.entrance >> 1) >= N || ((h->any || i) && r.entrance < h->last)) return 2;
        h->last = r.entrance;
    }
    if (!h->par) {
        h->par = true;
        h->cc.reset(N);
        h->pr.begin(N, h->st.plus.data(), h->st.minus.data(), h->st.complex_size, threads ? threads : 1);
    }
    h->any = true;
    auto list_of = [&](const pf_bfs_record &r) { return pool + r.list_off; };
    h->cc.add(records, n_records, list_of);
    std::vector<uint32_t> order, class_off;
    h->cc.order(records, n_records, pfh::kReplayClasses, order, class_off);
    pfh::ReplayStats st;
    h->pr.run(records, list_of, order.data(), class_off.data(), pfh::kReplayClasses, threads ? threads : 1, st);
    h->pr.finish(h->st.flags.data(), threads ? threads : 1);
    return 0;
}
uint64_t pfh_replay_check_footprints(const pf_bfs_record *records, uint64_t n_records, const uint32_t *pool, uint32_t n_unitigs,
                                     uint32_t complex_size, uint64_t slice, uint64_t *first_bad) {
    return pfh::check_footprints(records, n_records, pool, n_unitigs, complex_size, slice, first_bad);
}
// component label of every record's entrance side after the records were added (test hook for the device's labels)
void pfh_side_components(const pf_bfs_record *records, uint64_t n_records, const uint32_t *pool, uint32_t n_unitigs, uint32_t *labels) {
    pfh::SideComponents cc;
    cc.reset(n_unitigs);
    cc.add(records, n_records, [&](const pf_bfs_record &r) { return pool + r.list_off; });
    for (uint64_t i = 0; i < n_records; ++i) labels[i] = cc.label(pfh::entrance_side(records[i].entrance));
}
void pfh_replay_state(const pfh_replay *h, uint8_t *flags, uint32_t *plus, uint32_t *minus) {
    const size_t N = h->st.flags.size();
    if (flags) memcpy(flags, h->st.flags.data(), N);
    if (plus) memcpy(plus, h->st.plus.data(), N * 4);
    if (minus) memcpy(minus, h->st.minus.data(), N * 4);
}

// ---- colour sets (host only) ---------------------------------------------------------------------
}  // extern "C"

struct pfh_colors {
    pfh::UnitigSet graph;
    pfh::ColorSets sets;
};

extern "C" {

pfh_colors *pfh_colors_open(const char *gfa_path, const char *colors_path, uint32_t threads) {
    try {
        auto c = std::make_unique<pfh_colors>();
        if (!c->graph.load_gfa(gfa_path, g_open_err)) return nullptr;
        if (!c->sets.load(colors_path, c->graph, threads ? threads : 1, g_open_err)) return nullptr;
        return c.release();
    } catch (const std::exception &e) {
        g_open_err = std::string("ploidyfrost host layer: ") + e.what();
        return nullptr;
    }
}

// the footprint check with the colored commits: colour sets of an opened (graph, colours) pair, CSR rows from the caller
uint64_t pfh_colors_check_footprints(const pfh_colors *c, const uint32_t *succ, const pf_bfs_record *records, uint64_t n_records,
                                     const uint32_t *pool, uint32_t complex_size, uint64_t slice, uint64_t *first_bad) {
    pfh::ColourGate g;
    g.n_colors = c->sets.n_colors;
    g.k = c->graph.k;
    g.len_bp = c->graph.len_bp.data();
    g.words = c->sets.words;
    g.full_mask = c->sets.full_mask.data();
    g.size_total = c->sets.size_total.data();
    g.n_full_enc = c->sets.n_full_enc.data();
    g.succ = succ;
    return pfh::check_footprints(records, n_records, pool, c->graph.n(), complex_size, slice, first_bad, &g);
}
void pfh_colors_close(pfh_colors *c) { delete c; }
uint32_t pfh_colors_count(const pfh_colors *c) { return c->sets.n_colors; }
uint32_t pfh_colors_unitigs(const pfh_colors *c) { return c->graph.n(); }
const char *pfh_colors_name(const pfh_colors *c, uint32_t colour) {
    return colour < c->sets.names.size() ? c->sets.names[colour].c_str() : "";
}
uint64_t pfh_colors_unitig(const pfh_colors *c, uint32_t u, uint8_t *presence, uint32_t *n_kmers, uint32_t *n_full_enc) {
    const uint32_t km = c->graph.len_km(u);
    if (n_kmers) *n_kmers = km;
    if (n_full_enc) *n_full_enc = c->sets.n_full_enc[u];
    if (presence)
        for (uint32_t ci = 0; ci < c->sets.n_colors; ++ci)
            for (uint32_t i = 0; i < km; ++i) presence[(size_t)ci * km + i] = c->sets.contains(u, ci, i, 1);
    return c->sets.size_total[u];
}
uint64_t pfh_gfa_abundant_kmers(const char *gfa_path) {
    try {
        pfh::UnitigSet g;
        if (!g.load_gfa(gfa_path, g_open_err)) return ~0ull;
        return g.n_abundant;
    } catch (const std::exception &e) {
        g_open_err = std::string("ploidyfrost host layer: ") + e.what();
        return ~0ull;
    }
}
uint32_t pfh_gfa_numbering_replays(const char *gfa_path) {
    try {
        pfh::UnitigSet g;
        if (!g.load_gfa(gfa_path, g_open_err)) return ~0u;
        return g.numbering_replays;
    } catch (const std::exception &e) {
        g_open_err = std::string("ploidyfrost host layer: ") + e.what();
        return ~0u;
    }
}
int pfh_host_walk(const uint32_t *succ, const uint32_t *pred, uint32_t n_unitigs, uint32_t entrance, pf_bfs_record *record, uint32_t *list,
                  uint64_t list_cap) {
    if (!succ || !pred || !record || n_unitigs == 0 || (entrance >> 1) >= n_unitigs) return 1;
    try {
        static thread_local pfh::HugeWalker walker;
        const std::vector<uint32_t> &l = walker.walk(succ, pred, n_unitigs, entrance, *record);
        record->list_off = 0;
        if (record->n_list > list_cap) return 2;
        if (list) std::copy(l.begin(), l.begin() + record->n_list, list);
        return 0;
    } catch (const std::exception &e) {
        g_open_err = std::string("ploidyfrost host layer: ") + e.what();
        return 1;
    }
}
uint64_t pfh_host_walk_range(const uint32_t *succ, const uint32_t *pred, uint32_t n_unitigs, uint32_t u0, uint32_t u1,
                             pf_bfs_record *records, uint64_t rec_cap, uint32_t *pool, uint64_t pool_cap, uint64_t *pool_used) {
    if (!succ || !pred || u0 > u1 || u1 > n_unitigs) return ~0ull;
    try {
        pfh::HugeWalker walker;
        uint64_t n = 0, used = 0;
        bool fits = true;
        for (uint32_t ov = 2 * u0; ov < 2 * u1; ++ov) {
            int deg = 0;
            for (int b = 0; b < 4; ++b) deg += succ[(size_t)ov * 4 + b] != 0xFFFFFFFFu;
            if (deg < 2) continue;
            pf_bfs_record r;
            memset(&r, 0, sizeof(r));
            const std::vector<uint32_t> &l = walker.walk(succ, pred, n_unitigs, ov, r);
            r.list_off = used;
            if (n < rec_cap && used + r.n_list <= pool_cap && records && pool) {
                records[n] = r;
                std::copy(l.begin(), l.begin() + r.n_list, pool + used);
            } else {
                fits = false;
            }
            ++n;
            used += r.n_list;
        }
        if (pool_used) *pool_used = used;
        return fits ? n : ~0ull;
    } catch (const std::exception &e) {
        g_open_err = std::string("ploidyfrost host layer: ") + e.what();
        return ~0ull;
    }
}
uint64_t pfh_gfa_minimizer_counts(const char *gfa_path, uint8_t *counters, uint64_t slots) {
    try {
        pfh::UnitigSet g;
        if (!g.load_gfa(gfa_path, g_open_err, true)) return ~0ull;
        std::vector<uint8_t> c;
        g.finish_numbering(&c);
        if (counters && slots >= c.size()) std::copy(c.begin(), c.end(), counters);
        return c.size();
    } catch (const std::exception &e) {
        g_open_err = std::string("ploidyfrost host layer: ") + e.what();
        return ~0ull;
    }
}
int pfh_gfa_write_unitig_ids(const char *gfa_path, const char *out_path) {
    try {
        pfh::UnitigSet g;
        if (!g.load_gfa(gfa_path, g_open_err)) return 1;
        FILE *f = fopen(out_path, "w");
        if (!f) { g_open_err = std::string("cannot write ") + out_path; return 1; }
        for (uint32_t u = 0; u < g.n(); ++u) {
            const std::string_view s = g.seq(u);
            fprintf(f, "%u\t%.*s\n", u + 1, (int)s.size(), s.data());
        }
        fclose(f);
        return 0;
    } catch (const std::exception &e) {
        g_open_err = std::string("ploidyfrost host layer: ") + e.what();
        return 1;
    }
}
// The numbering with the replay's inputs handed in, as the device layer hands them over (pf_minimizer_replay_inputs): `bump` is added to
// every counter of the host's own pass (saturating: upper bounds, as K-MINZ's are) and every unitig is flagged.  Must write the same
// file as pfh_gfa_write_unitig_ids.  [tests]
int pfh_gfa_write_unitig_ids_given_inputs(const char *gfa_path, const char *out_path, int bump) {
    try {
        std::vector<uint8_t> c;
        {
            pfh::UnitigSet probe;
            if (!probe.load_gfa(gfa_path, g_open_err, true)) return 1;
            probe.finish_numbering(&c);
        }
        for (uint8_t &x : c) x = (uint8_t)std::min<int>(255, (int)x + bump);
        pfh::UnitigSet g;
        if (!g.load_gfa(gfa_path, g_open_err, true)) return 1;
        const std::vector<uint8_t> flags(g.n(), 1);
        g.finish_numbering(nullptr, c.empty() ? nullptr : c.data(), c.empty() ? nullptr : flags.data(), c.size());
        FILE *f = fopen(out_path, "w");
        if (!f) { g_open_err = std::string("cannot write ") + out_path; return 1; }
        for (uint32_t u = 0; u < g.n(); ++u) {
            const std::string_view s = g.seq(u);
            fprintf(f, "%u\t%.*s\n", u + 1, (int)s.size(), s.data());
        }
        fclose(f);
        return 0;
    } catch (const std::exception &e) {
        g_open_err = std::string("ploidyfrost host layer: ") + e.what();
        return 1;
    }
}
// ---- `PloidyFrost model` ---------------------------------------------------------------------------------------------
pfh_gmm *pfh_gmm_open(int device) {
    try {
        return new pfh_gmm(device);
    } catch (const std::exception &e) {
        g_open_err = std::string("ploidyfrost host layer: ") + e.what();
        return nullptr;
    }
}
void pfh_gmm_close(pfh_gmm *m) { delete m; }
const char *pfh_gmm_last_error(const pfh_gmm *m) { return m ? (m->err.empty() ? m->model.error().c_str() : m->err.c_str()) : g_open_err.c_str(); }
int pfh_gmm_read_fre(pfh_gmm *m, const char *file, double min_frequency) {
    return gmm_guard(m, [&] { return m->model.readFreFile(file, min_frequency); });
}
int pfh_gmm_read_cov(pfh_gmm *m, const char *prefix, double min_frequency) {
    return gmm_guard(m, [&] { return m->model.readCovFile(prefix, min_frequency); });
}
int pfh_gmm_set_values(pfh_gmm *m, const double *values, uint64_t n) {
    return gmm_guard(m, [&] { m->model.readData(std::vector<double>(values, values + n)); return 0; });
}
uint64_t pfh_gmm_size(const pfh_gmm *m) { return m ? m->model.values().size() : 0; }
int pfh_gmm_values(const pfh_gmm *m, double *out) {
    if (!m || !out) return 1;
    std::copy(m->model.values().begin(), m->model.values().end(), out);
    return 0;
}
int pfh_gmm_fit(pfh_gmm *m, uint32_t gauss, double m_thre, double n_thre, int32_t max_iter, double max_delta, double *weights,
                double *means, double *vars, double *loglik, double *aic, uint32_t *iterations) {
    return gmm_guard(m, [&] {
        m->model.setMThreshold(m_thre);
        m->model.setNThreshold(n_thre);
        m->model.setMaxIterNum(max_iter);
        m->model.setMaxDeltaNum(max_delta);
        m->model.resize(gauss);
        if (m->model.emIterate()) return 1;
        for (uint32_t i = 0; i < gauss; ++i) {
            if (weights) weights[i] = m->model.getWeights()[i];
            if (means) means[i] = m->model.getMeans()[i];
            if (vars) vars[i] = m->model.getVars()[i];
        }
        if (loglik) *loglik = m->model.getLogLikelihood();
        if (aic) *aic = m->model.getAIC();
        if (iterations) *iterations = m->model.iterations();
        return 0;
    });
}
int pfh_gmm_run(pfh_gmm *m, int min_gauss, int max_gauss, double m_thre, double n_thre, int32_t max_iter, double max_delta,
                const char *outprefix) {
    return gmm_guard(m, [&] {
        m->model.setMThreshold(m_thre);
        m->model.setNThreshold(n_thre);
        m->model.setMaxIterNum(max_iter);
        m->model.setMaxDeltaNum(max_delta);
        return pfh::run_model(m->model, min_gauss, max_gauss, outprefix, m->err);
    });
}
int pfh_gmm_kernel_time(pfh_gmm *m, int enable, double *total_ms, uint64_t *launches) {
    if (!m) return 1;
    pf_ctx *ctx = m->model.device_context();
    if (!ctx) return 1;
    if (enable >= 0) return pf_enable_timing(ctx, enable);
    return pf_kernel_time(ctx, PF_K_GMM, total_ms, launches);
}

uint64_t pfh_bifrost_kmer_hash(uint64_t left_aligned_kmer, uint64_t seed) { return pfh::bifrost_kmer_hash(left_aligned_kmer, seed); }

}  // extern "C"
