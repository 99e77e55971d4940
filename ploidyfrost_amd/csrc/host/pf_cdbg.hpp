// pfh::CDBG -- the product's host-side mirror of the reference class CDBG
// (reference src/CDBG.hpp:20-41, driven from src/Main.cpp:829-849):
//
//     CDBG g(graph, z, M, D, G, kmc_prefix);
//     g.setUnitigId(outpre, graphfile, threads);
//     g.printInfo(verbose, outpre);                              // only with -i
//     g.findSuperBubble_multithread_ptr(outpre, threads);
//     g.ploidyEstimation_multithread_ptr(outpre, lower, upper, threads);
//
// Same method names, argument meaning, output files (./PloidyFrost_output/<outpre>_*.txt) and
// stdout lines.  Differences, all deliberate:
//   * every compute-heavy step runs on the MI355X through include/ploidyfrost_hip.h
//     (adjacency join, superbubble traversals, per-unitig and per-string k-mer coverage,
//     Needleman-Wunsch fill + traceback); the host keeps only the order-dependent state
//     replay, the progressive-alignment bookkeeping and the text formatting;
//   * methods return an int status instead of calling exit(); main() turns a failure into the
//     reference's "message + exit(EXIT_FAILURE)";
//   * output is always the deterministic `-t 1` ordering, whatever `threads` says.
#pragma once
#include <chrono>
#include <memory>
#include <condition_variable>
#include <mutex>

#include "pf_bfs_host.hpp"
#include <cstdint>
#include <memory>
#include <string>
#include <thread>
#include <utility>
#include <vector>

#include "pf_host_colors.hpp"
#include "pf_host_graph.hpp"
#include "pf_pinned.hpp"
#include "pf_replay_par.hpp"
#include "pf_state.hpp"
#include "ploidyfrost_hip.h"

namespace pfh {

struct Scoring {
    double match = 2, mismatch = -1, gap = -3;
};

struct PhaseTimes {
    double bfs_device_s = 0, replay_s = 0, bubble_write_s = 0;
    double cov_device_s = 0, tasks_s = 0, align_s = 0, sites_s = 0, format_s = 0, write_s = 0;
    double find_total_s = 0, ploidy_total_s = 0;
    double scan_s = 0, scan_serial_s = 0, align_build_s = 0, align_device_s = 0, align_post_s = 0, align_choose_s = 0;
    uint64_t bfs_large = 0, bfs_large_seen = 0, bfs_max_seen = 0;  // traversals beyond 4096 unitigs
    uint64_t bfs_large_used = 0, bfs_large_used_max = 0;            // ... that the replay's gate let through, and the longest of those
    uint64_t candidates = 0, bfs_deferred = 0, bubbles_out = 0, tasks = 0, align_jobs = 0, site_strings = 0;
    uint64_t host_commit_records = 0, host_walk_vertices = 0;   // findSuperBubble's host share: records committed on a host thread (large components + walked traversals), vertices the walkers visited
    uint64_t snp_jobs = 0, pair_jobs = 0, wave_jobs = 0, stack_jobs = 0;   // of align_jobs: by K-SNP, by K-PAIR, by K-BUBBLE, by K-STACK
};

// The count database on its way to HBM while the caller is still reading the graph file: context creation, header / prefix
// table parse, K-KMC decode and the table build need nothing from the graph, so a front end starts them first
// (CountsLoader::start) and hands the loader to the CDBG constructor, which waits for it and adopts its context.
struct CountsLoader {
    ~CountsLoader();
    void start(int device, const std::string &kmc_prefix);
    // valid after wait()
    pf_ctx *ctx = nullptr;
    int status = 0;
    std::string error;
    bool both_strands = true;
    int k = 0;
    void wait() { if (th_.joinable()) th_.join(); }
    pf_ctx *release() { pf_ctx *c = ctx; ctx = nullptr; return c; }
    // the context as soon as it exists (the count table is still on its way): nullptr when its creation failed.  Only calls
    // that are safe beside the table's ingest may use it before wait() (pf_gfa_parse).
    pf_ctx *wait_context() {
        std::unique_lock<std::mutex> lk(mu_);
        cv_.wait(lk, [&] { return ctx_known_; });
        return ctx;
    }

private:
    std::thread th_;
    std::mutex mu_;
    std::condition_variable cv_;
    bool ctx_known_ = false;
};

class CDBG {
public:
    CDBG(UnitigSet &graph, const size_t &complexsize, double &m, double &d, double &g, std::string kmc_db = "",
         int device = 0, bool quiet = false, CountsLoader *counts = nullptr);
    virtual ~CDBG();
    CDBG(const CDBG &) = delete;
    CDBG &operator=(const CDBG &) = delete;

    bool good() const { return status_ == 0; }
    int status() const { return status_; }
    const std::string &error() const { return err_; }

    int setUnitigId(const std::string &outpre, const std::string &graphfile, const size_t &thr);
    int printInfo(const bool &verbose, const std::string &outpre);
    int findSuperBubble_multithread_ptr(const std::string &outpre, const size_t &thread);
    int ploidyEstimation_multithread_ptr(const std::string &outpre, const int &lower, const int &upper, const size_t &thr);

    // extras (not in the reference interface)
    void set_output_dir(const std::string &dir) { outdir_ = dir; }  // default "PloidyFrost_output"
    void set_quiet(bool q) { quiet_ = q; }
    void set_threads(unsigned t) { threads_ = t; }                  // host threads; 0 = use the `thr` argument
    void set_third_tier_on_host(bool on) { third_tier_on_host_ = on; }   // default on; off = one wavefront per giant traversal
    void set_write_files(bool w) { write_files_ = w; }              // bench: format but do not touch the disk
    void set_batch_bubbles(size_t n) { batch_bubbles_ = n; }
    // resident pipeline: text pieces (4 x batch_bubbles bubbles each) per alignment launch; default 64: a whole pass of up
    // to 2^24 bubbles is aligned in one launch sequence
    void set_align_pieces(size_t n) { align_pieces_ = n ? n : 1; }
    // write <outpre>_super_bubble.txt in the background while PloidyEstimation runs (complete when that call, the next
    // findSuperBubble or the destructor returns); off by default: the file is complete when findSuperBubble returns
    void set_overlap_output(bool on) { overlap_output_ = on; }
    // PloidyEstimation through the device's resident calling pipeline (pf_call_*, the default for the single-sample path) or,
    // off, through the host-threaded pipeline around pf_align_bubbles / pf_string_cov (what the colored path uses)
    void set_resident_calling(bool on) { resident_ = on; }
    // Text format of the reference's `-t N` run with N > 1 (single-sample path): BubbleId / var_count from 0, the allele_frequency
    // rows of a bubble grouped by arity (src/CDBG.cpp:1829, 2056, 2158-2162, 2550) and its stdout wording; the order of rows
    // stays the deterministic `-t 1` one (the reference's own depends on thread timing).
    int set_reference_threads(size_t n);
    // ---- one graph over several GPUs (reference owner rule src/CDBG.cpp:1190, 1352, 1656-1679) -------------
    // findSuperBubble: every rank traverses the entrances of its unitig range (find_shard), the ranks exchange the records,
    // every rank replays all of them (find_replay; rank 0 writes the file).  PloidyEstimation: the scan and the sequential pass
    // run on every rank (ploidy_select), each rank aligns its slice of the bubble list (ploidy_align), the ranks exchange how
    // many of their bubbles were called, format with that base (ploidy_text) and, knowing every rank's slab sizes, write
    // their slabs at their offsets of the shared result files (ploidy_write).
    int find_shard(uint32_t u0, uint32_t u1);
    const std::vector<pf_bfs_record> &shard_records() const { return shard_rec_; }
    const std::vector<uint32_t> &shard_pool() const { return shard_pool_; }
    // dev_records / dev_pools (optional): the same shards where they already lie in device memory (after an all-gather), so that
    // the components of the parallel replay are found without another upload; pool_lens = entries of each pool
    int find_replay(const std::string &outpre, uint32_t n_shards, const pf_bfs_record *const *records, const uint64_t *n_records,
                    const uint32_t *const *pools, bool write_file, const uint64_t *pool_lens = nullptr,
                    const pf_bfs_record *const *dev_records = nullptr, const uint32_t *const *dev_pools = nullptr);
    // the commit replay on host threads (pf_replay_par.hpp): 0 = sequential; default min(threads, 32) for the single-sample path
    void set_replay_threads(int t) { replay_threads_ = t; }
    // one graph on several ranks with findSuperBubble run by each of them: only one rank writes <outpre>_super_bubble.txt
    void set_write_super_bubble(bool on) { write_sb_ = on; }
    int ploidy_select(int lower, int upper, uint64_t &n_bubbles);
    int ploidy_select(const std::vector<std::pair<int, int>> &cutoff, uint64_t &n_bubbles);   // colored: one pair per colour
    int ploidy_align(uint64_t t0, uint64_t t1, uint64_t &n_called);
    int ploidy_text(uint64_t var_count_base, uint64_t sizes[PF_CALL_STREAMS], uint64_t counters[8]);
    int ploidy_write(const std::string &outpre, const uint64_t offsets[PF_CALL_STREAMS], const uint64_t totals[PF_CALL_STREAMS], bool truncate);
    pf_ctx *device() { return ctx_; }
    const PhaseTimes &times() const { return times_; }
    uint64_t allele_sites(int arity) const { return allele_[arity - 2]; }
    uint64_t core_cov() const { return core_cov_; }
    uint64_t core_num() const { return core_num_; }
    uint64_t n_superbubbles() const { return n_super_bubble_; }
    // (after commits on the device the host copy is fetched when somebody asks)
    const std::vector<uint8_t> &state_flags() { sync_state_to_host(); return flags_; }
    const std::vector<uint32_t> &state_plus() { sync_state_to_host(); return plus_; }
    const std::vector<uint32_t> &state_minus() { sync_state_to_host(); return minus_; }
    int sync_state_to_host();
    uint64_t output_bytes() const { return out_bytes_; }
    // text of <outpre>_allele_frequency.txt of the last run (the record slab a multi-GPU job gathers)
    // (after a pass that wrote its files the rows are read back from <outpre>_allele_frequency.txt when somebody asks, instead of
    // being copied aside piece by piece under the writer of every pass)
    const std::string &last_allele_frequency() const;

protected:
    // graph + adjacency on the device, no count database yet (the colored subclass brings its own)
    struct NoCounts {};
    CDBG(UnitigSet &graph, const size_t &complexsize, double &m, double &d, double &g, int device, bool quiet, NoCounts);
    int init_device(int device, pf_ctx *adopt = nullptr, bool colored = false);
    void start_prealloc();  // the first launches' device buffers, on a helper thread (single-sample, resident pipeline)
    // the path proper; cutoff holds one (lower, upper) pair (single sample) or one per colour
    int ploidy_estimation(const std::string &outpre, const std::vector<std::pair<int, int>> &cutoff, const size_t &thr);
    // the same through pf_call_* (pf_cdbg_call.cpp); call_select = scan + the sequential pass of the driver loop
    int ploidy_estimation_resident(const std::string &outpre, const std::vector<std::pair<int, int>> &cutoff, const size_t &thr);
    int call_select(const std::vector<std::pair<int, int>> &cutoff, uint64_t &n_tasks);
    bool resident_path() const { return resident_ && (col_ == nullptr || colored_resident_); }
    // set by CCDBG: the colour sets of the graph's unitigs (reference src/CCDBG.cpp path) and the stdout tag
    const ColorSets *col_ = nullptr;
    const char *tag_ = "CDBG";

    struct Task;
    using clk_time = std::chrono::steady_clock::time_point;
    int finish_find(const std::string &outpre, const size_t &thr, clk_time t_all, bool write_file);
    std::vector<pf_bfs_record> shard_rec_;
    std::vector<uint32_t> shard_pool_;
    pf_call_result slice_res_ = {};
    uint64_t slice_nb_ = 0, slice_var_base_ = 0;   // ploidy_align's range and ploidy_text's numbering base, for ploidy_write's pieces
    int fail(int st, const std::string &msg);
    int join_pending_write();
    int join_pending_ids();
    // pinned exchange buffers of the first pass, allocated on a helper thread while the graph goes to the device (page-locking
    // a few hundred MB costs tens of milliseconds; sizes are estimates from the unitig count, the phases grow them if needed)
    std::thread prealloc_, prealloc_walkers_, prealloc_call_, prealloc_align_;
    void join_prealloc() {
        if (prealloc_.joinable()) prealloc_.join();
        if (prealloc_walkers_.joinable()) prealloc_walkers_.join();
        if (prealloc_call_.joinable()) prealloc_call_.join();
        if (prealloc_align_.joinable()) prealloc_align_.join();
    }
    std::thread pending_ids_;
    int pending_ids_rc_ = 0;
    uint64_t ids_bytes_ = 0;
    int launch_coverage();
    uint64_t find_passes_ = 0;
    bool cov_ready_ = false;  // the coverage arrays in bx_ are those of the current graph and count table
    std::string cov_err_;
    bool overlap_output_ = false;
    std::thread pending_write_;
    int pending_rc_ = 0;
    int ensure_dir();
    int write_file(const std::string &name, const std::string &data);
    int write_pieces(const std::string &name, const std::vector<const std::string *> &pieces, uint64_t &bytes) const;
    int write_many(const std::vector<std::pair<std::string, std::vector<const std::string *>>> &files, unsigned threads);
    void replay(const pf_bfs_record &r, const uint32_t *list) { st_.replay(r, list); }

    UnitigSet &g_;
    size_t complex_size_;
    Scoring sc_;
    std::string outdir_ = "PloidyFrost_output";
    pf_ctx *ctx_ = nullptr;
    int status_ = 0;
    std::string err_;
    bool quiet_ = false, write_files_ = true;
    bool both_strands_ = true;
    unsigned threads_ = 0;
    int replay_threads_ = -1;   // -1: default
    bool write_sb_ = true;
    // commits on the device (pf_replay_device): the host arrays are stale until sync_state_to_host; big_f2_ = the per-side flag
    // bytes of the few components committed on the host (kept all-zero between passes)
    bool state_host_stale_ = false;
    bool colours_on_device_ = false;   // colored path: pf_replay_set_colours succeeded
    bool colored_resident_ = false;    // colored path: pf_call_set_colours succeeded -- the calling phase runs on the resident pipeline
    std::vector<uint8_t> big_f2_;
    bool commits_on_device(size_t thr) const;
    int find_superbubbles_device(const std::string &outpre, const size_t &thr);
    ParallelReplay par_;
    unsigned replay_threads(size_t thr) const;
    // third K-BFS tier (traversals beyond 4096 vertices) on host cores, pf_bfs_host.hpp; false = the device's k_bfs_huge
    bool third_tier_on_host_ = true;
    std::vector<std::unique_ptr<HugeWalker>> walkers_;
    std::mutex walkers_mu_;
    std::vector<uint32_t> deferred_;   // record indices pf_bfs_candidates_split leaves to the host walkers
    std::vector<uint32_t> deferred_ent_;   // their entrances (pf_bfs_candidates_begin)
    // vertex lists of the records walked on the host (pf_bfs_record::pad_ == 1, list_off = index), per K-BFS slice: a slice's
    // lists are complete before its records are handed to the replay
    std::vector<std::vector<uint32_t>> huge_lists_[4];

    std::vector<uint32_t> succ_, pred_;  // host copy of the CSR, [2N][4]
    // MyUnitig state (reference src/MyUnitig.hpp), array-indexed, with the commits that mutate it (pf_state.hpp)
    UnitigState st_;
    std::vector<uint8_t> &flags_ = st_.flags;
    std::vector<uint32_t> &plus_ = st_.plus, &minus_ = st_.minus;  // 0 = NULL, id = u + 1

    // pinned exchange buffers of pf_align_bubbles, three sets: one per batch in flight (reused from pass to pass)
    struct AlignExchange {
        PinnedBuf<char> text, otext;
        PinnedBuf<pf_bubble_path> paths;
        PinnedBuf<pf_bubble_task> tasks;
        PinnedBuf<pf_bubble_result> res;
        PinnedBuf<pf_bubble_site> osites;
        PinnedBuf<uint8_t> ogroups;
        PinnedBuf<uint32_t> oilen;
        void release_all() {
            text.release(); otext.release(); paths.release(); tasks.release(); res.release(); osites.release();
            ogroups.release(); oilen.release();
        }
    } ax_[3];
    // (a text piece = four batches: 131 072 bubbles, 18 MB over PCIe at k = 25 -- pieces short enough that the last one's PCIe copy and file
    // copy, which nothing overlaps, stay small)
    size_t batch_bubbles_ = 32768, align_pieces_ = 64;   // (text pieces of 4 x 32768 bubbles: tools/ab_pass.py BATCH=n, profiles/r16_experiments.txt)
    // pinned buffers of the whole-graph device calls
    struct BubbleExchange {
        PinnedBuf<pf_bfs_record> bfs_rec;
        PinnedBuf<uint32_t> bfs_pool;
        PinnedBuf<uint32_t> bfs_order;   // parallel replay: record indices grouped by class (pf_replay_order)
        PinnedBuf<uint64_t> cov_sum;
        PinnedBuf<uint32_t> cov_min;
        PinnedBuf<uint8_t> cov_miss;
        PinnedBuf<uint32_t> cov_max;  // colored path: per (colour, unitig) arrays, colour-major
        void release_all() {
            cov_max.release();
            bfs_rec.release(); bfs_pool.release(); bfs_order.release(); cov_sum.release(); cov_min.release();
            cov_miss.release();
        }
    } bx_;
    // exchange buffers of the resident calling pipeline: side records down, selection up, text slabs down
    struct CallExchange {
        PinnedBuf<pf_call_side> sides;
        PinnedBuf<uint32_t> kept;
        PinnedBuf<char> slab[2];
        void release_all() { sides.release(); kept.release(); slab[0].release(); slab[1].release(); }
    } cx_;
    bool resident_ = true;
    bool mt_format_ = false;
    // the result files of the resident pipeline and super_bubble.txt, kept open and mapped between passes (pf_cdbg_impl.hpp)
    std::unique_ptr<struct MappedOut[]> out_maps_;
    bool state_on_device_ = false;   // pf_call_set_state holds the state findSuperBubble left (finish_find uploads it)
    PinnedBuf<char> sb_text_;        // text of super_bubble.txt on its way from the device to the file
    PhaseTimes times_;
    mutable std::string last_allfre_, last_allfre_file_;
    mutable uint64_t last_allfre_bytes_ = 0;
    uint64_t allele_[4] = {0, 0, 0, 0};
    uint64_t core_cov_ = 0, core_num_ = 0, n_super_bubble_ = 0, out_bytes_ = 0;
};

// pfh::ColoredUnitigSet -- what the colored path needs of the reference's ColoredCDBG<MyUnitig>
// (bifrost/src/ColoredCDBG.hpp): the unitigs of the GFA file and their colour sets from the .bfg_colors file.
struct ColoredUnitigSet {
    UnitigSet graph;
    ColorSets colors;
    std::string err;
    // ColoredCDBG::read(graphfile, colorfile, nb_threads, verbose) (bifrost/src/ColoredCDBG.tcc:428-600)
    bool read(const std::string &graphfile, const std::string &colorfile, size_t nb_threads = 1, bool verbose = false);
    size_t getNbColors() const { return colors.n_colors; }
    int getK() const { return graph.k; }
    size_t size() const { return graph.n(); }
    bool reload_colors();   // after the unitig order changed (abundant k-mers found by the device census)

private:
    std::string colorfile_;
    unsigned color_threads_ = 1;
};

// pfh::CCDBG -- mirror of the reference class CCDBG (src/CCDBG.hpp:18-40, driven from src/Main.cpp:775-810):
//
//     CCDBG g(cdbg, z, M, D, G, kmc_list_file, threads);
//     g.setUnitigId(outpre, graphfile, threads);
//     g.findSuperBubble_multithread_ptr(outpre, threads);
//     g.ploidyEstimation_multithread_ptr(outpre, cutoffs, threads);      // one (lower, upper) per colour
//
// Output files are byte-identical to the reference's `-t 1` run.  All colours' count databases live in one
// HBM table (include/ploidyfrost_hip.h, pf_upload_counts_colored).
class CCDBG : public CDBG {
public:
    CCDBG(ColoredUnitigSet &graph, const size_t &complexsize, double &m, double &d, double &g, std::string kmc_db_list = "",
          const size_t &thread = 1, int device = 0, bool quiet = false);
    int ploidyEstimation_multithread_ptr(const std::string &outpre, const std::vector<std::pair<int, int>> &cutoff, const size_t &thr);

private:
    ColoredUnitigSet &cg_;
};

}  // namespace pfh
