// Context behind the C ABI (include/ploidyfrost_hip.h): everything resident in HBM for one GPU.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <mutex>
#include <string>
#include <vector>

#include "pf_device_common.hpp"
#include "ploidyfrost_hip.h"

namespace pf {
// device allocation that lives until the end of the enclosing call, freed on every return path
template <typename T>
struct DevTmp {
    T *p = nullptr;
    DevTmp() = default;
    DevTmp(const DevTmp &) = delete;
    DevTmp &operator=(const DevTmp &) = delete;
    ~DevTmp() { if (p) (void)hipFree(p); }
    hipError_t alloc(size_t bytes) { return hipMalloc(reinterpret_cast<void **>(&p), bytes ? bytes : 1); }
};

struct TimedLaunch {
    int kernel;
    hipEvent_t a, b;
    bool closed;   // b has been recorded
};
struct CallState;  // pf_call.hip: buffers of the resident calling pipeline
}  // namespace pf

struct pf_ctx {
    int device = 0;
    int n_cu = 256;
    std::string name;
    std::string err;
    hipStream_t own_stream = nullptr, stream = nullptr;

    // graph SoA (HBM): 2-bit packed unitigs, word offsets, lengths; CSR adjacency
    uint64_t *d_seq = nullptr;
    uint64_t *d_off = nullptr;
    uint32_t *d_len = nullptr;
    uint32_t N = 0;
    int k = 0;
    uint64_t n_words = 0, n_kmers = 0;
    // k-mer numbering for the k-mer-parallel coverage kernels: d_kpre[u] = k-mers of unitigs < u (N + 1 entries),
    // d_kwin[w] = the unitig holding global k-mer 256 * w
    uint64_t *d_kpre = nullptr;
    uint32_t *d_kwin = nullptr;
    uint64_t n_kwin = 0;
    uint32_t *d_succ = nullptr, *d_pred = nullptr;  // [2N][4]
    uint32_t *d_pred16 = nullptr;                   // [2N][4][4] two-hop rows, built on first use by the huge BFS tier
    bool has_adj = false;
    uint32_t *d_cand = nullptr;    // oriented vertices with out-degree > 1, ascending
    std::vector<uint32_t> h_cand;  // host copy (shard range queries)
    std::vector<uint32_t> h_len;   // host copy of the unitig lengths (argument validation)

    // k-mer count table (HBM): open addressing, 16-B slots, capacity = power of two >= 2n
    pf::CountLine *d_tab = nullptr;   // count table: tab_cap lines of ten keys (pf_device_common.hpp)
    uint64_t tab_cap = 0, tab_n = 0;
    // K-COV-JOIN between pf_join_counts_begin and the first reader of d_gcov (join_finish)
    hipStream_t join_stream = nullptr;
    hipEvent_t join_done = nullptr;
    bool join_inflight = false;
    hipEvent_t join_c_done = nullptr;   // the same for the colored join (pf_colored.hip)
    bool join_c_inflight = false;
    int tab_k = 0;                // k of the database: a count table is addressed by the minimizers of its keys (pf_device_common.hpp)
    bool tab_one_strand = false;  // no k-mer is stored in both orientations (checked when the table is built)
    bool tab_exact = false;       // database built without canonical counting: no composite lookups
    uint64_t tab_max_count = 0;   // upper count filter of the last upload (decides the width of K-COV's row sums)

    // per-k-mer coverage SoA in graph order (K-COV-JOIN at load, streamed by K-COV every pass):
    // d_gcov[g] = count of graph k-mer g (unitigs laid end to end, numbering of d_kpre), PF_GCOV_MISSING = not in the table;
    // bit g % 64 of d_khead[g / 64] = k-mer g is the first of its unitig (bit n_kmers is set as well);
    // d_krow[r] = the unitig holding k-mer 64 * r
    uint32_t *d_gcov = nullptr;
    uint64_t *d_khead = nullptr;
    uint32_t *d_krow = nullptr;
    uint64_t n_krow = 0;
    bool gcov_valid = false;

    // colored path: the count databases of all colours in one table, slot = { u64 key, u32 count[n_colors] }
    // padded to 1 << ctab_shift bytes (pf_colored.hip)
    uint8_t *d_ctab = nullptr;
    uint64_t ctab_cap = 0;
    uint32_t ctab_line_bytes = 128, n_colors = 0;   // ctab_cap lines of ctab_line_bytes (pf_colored_dev.hpp)
    bool ctab_one_strand = false;
    uint64_t ctab_unread = 0;   // colours whose database was written without canonical counting: never looked up (the first 64 colours: the resident pipeline's)
    uint8_t *d_unread = nullptr;   // the same per colour, one byte each (any number of colours)
    uint64_t ctab_max_count = 0;  // largest upper count filter among the colours that are looked up
    // colored per-k-mer coverage SoA: colour c's count of graph k-mer g at d_gcov_c[c * gcov_c_stride + g] (K-COV-C-JOIN)
    uint32_t *d_gcov_c = nullptr;
    uint64_t gcov_c_stride = 0;
    bool gcov_c_valid = false;

    // reusable result staging for host-pointer callers
    uint64_t *d_cov_sum = nullptr;
    uint32_t *d_cov_min = nullptr;
    uint8_t *d_cov_miss = nullptr;
    uint32_t cov_cap = 0;

    // `model` sub-command (pf_gmm.hip): allele frequencies resident in workspace WS_GMM_X
    uint64_t gmm_n = 0;
    bool gmm_loaded = false;

    pf::CallState *call = nullptr;
    // K-BUBBLE: one stream per LDS size class (pf_bubble.hip), [classes] = the event the class streams wait for; a set per lane of
    // the calling pipeline (PF_CALL_LANES)
    hipStream_t bub_streams[PF_CALL_LANES][12] = {};   // (pf_bubble_launch.hpp: kBubMaxClasses)
    hipEvent_t bub_events[PF_CALL_LANES][13] = {};

    // K-CC (pf_cc.hip): union-find over unitig sides for the parallel commit replay; the records and vertex pool of the last
    // K-BFS call as they lie in the workspace
    void *cc = nullptr;
    void *gfa = nullptr;   // K-GFA (pf_gfa.hip): segment table of the last pf_gfa_ingest until pf_gfa_segments fetches it
    void *comm = nullptr;  // pf_gather.hip: this rank's RCCL communicator and its two small device buffers
    const void *cc_rec = nullptr;
    const uint32_t *cc_pool = nullptr;
    const pf_bfs_record *bfs_last_rec = nullptr;
    const uint32_t *bfs_last_pool = nullptr;
    uint64_t bfs_last_n = 0, bfs_last_pool_len = 0;
    uint64_t bfs_call_id = 0;
    uint64_t bfs_res_pool_cap = 0;   // pf_bfs_candidates_resident: pool size that sufficed last time
    // pf_bfs_candidates_begin .. _end: the copy of records and pool to the host in flight on copy_stream
    hipStream_t copy_stream = nullptr;
    struct {
        bool active = false;
        pf_bfs_record *records = nullptr;
        uint64_t c0 = 0;
        std::vector<uint32_t> deferred;
    } bfs_pending;

    unsigned int bfs_deferred = 0;  // candidates of the last pf_bfs_candidates that needed the big tier
    unsigned int bfs_live_n = 0;    // entries the last pf_bfs_candidates call put into the live list (pf_bfs_live_count)
    unsigned long long *h_live = nullptr;   // pf_bfs_live_deferred: pinned, coherent host memory the wave tier reports its give-ups into
    uint64_t live_cap = 0;

    // grow-only device workspaces reused across calls (slot ids: enum pf::WsSlot)
    std::vector<std::pair<void *, size_t>> ws;

    bool timing = false;
    uint64_t timing_mask = ~0ull;          // pf_timing_select: bit k = launches of kernel k are timed
    std::vector<hipEvent_t> event_pool;    // events of launches pf_reset_timing has dropped, for the launches to come
    std::vector<pf::TimedLaunch> launches;
    size_t side_launch = (size_t)-1;   // ctx_begin_on .. ctx_end_on
    size_t main_launch = (size_t)-1;   // ctx_begin .. ctx_end
    std::mutex launch_mu;              // `launches` is appended to by the thread that formats text as well (ctx_begin_at)
    uint64_t units[PF_K_COUNT_] = {};  // work items handed to the timed launches of each kernel (pf_kernel_units)
};

namespace pf {
// pf_ctx::err written from a call that may run beside another one on the same context (pf_call_align_lane on one host thread,
// pf_call_text_range_lane on another): `CtxErr{ctx} = "..."` assigns under the context's lock; pf_last_error copies under it
struct CtxErr {
    pf_ctx *c;
    void operator=(std::string s) const {
        std::lock_guard<std::mutex> lk(c->launch_mu);
        c->err = std::move(s);
    }
};
struct Kc4Args;
int launch_cov_stream(pf_ctx *ctx, Kc4Args a, uint32_t n_colors, bool wide, bool colored);  // pf_device.hip
int ctx_begin(pf_ctx *ctx, int kernel);
void ctx_end(pf_ctx *ctx);
int ctx_begin_on(pf_ctx *ctx, int kernel, hipStream_t stream);   // the same for a launch on another stream
void ctx_end_on(pf_ctx *ctx, hipStream_t stream);
// the same for a caller that keeps the place of its launch itself: any thread, any stream
int ctx_begin_at(pf_ctx *ctx, int kernel, hipStream_t stream, size_t *at);
void ctx_end_at(pf_ctx *ctx, size_t at, hipStream_t stream);
inline void ctx_units(pf_ctx *ctx, int kernel, uint64_t n) { if (ctx->timing) __atomic_fetch_add(&ctx->units[kernel], n, __ATOMIC_RELAXED); }
int ctx_grid(const pf_ctx *ctx, uint64_t work_items, int block, int per_cu);
int join_graph_counts(pf_ctx *ctx);
int join_graph_counts_begin(pf_ctx *ctx);   // pf_device.hip: the kernels on a stream of their own; join_finish() before d_gcov is read
int join_finish(pf_ctx *ctx);
int join_graph_counts_colored_begin(pf_ctx *ctx);
int join_colored_finish(pf_ctx *ctx);
hipStream_t join_stream(pf_ctx *ctx);   // pf_device.hip: the stream both joins run on
int join_graph_counts_colored(pf_ctx *ctx);  // pf_colored.hip: the same for the joined table of all colours (pf_ctx::d_gcov_c)  // K-COV-JOIN (pf_device.hip): fills pf_ctx::d_gcov when graph and canonical count table are both resident
// device workspace `slot`, at least `bytes` large (contents undefined); nullptr on allocation failure
void *ctx_ws(pf_ctx *ctx, int slot, size_t bytes);
// a stream for the kernels of lane `lane` of the calling pipeline: lane 0's at the default priority, the other lanes' at the lowest --
// when ranges are aligned side by side (PF_ALIGN_THREADS) the earlier range, whose text the pass waits for, goes first and the
// later one fills what it leaves idle.  PF_LANE_PRIORITY=0: all at the default priority (measurements).
hipError_t lane_stream_create(hipStream_t *s, int lane);
// PF_TRACE_LOAD: where a load call of this library spends its time (the host layer's LoadTrace prints the calls themselves)
struct DevLoadTrace {
    bool on = getenv("PF_TRACE_LOAD") != nullptr;
    std::chrono::steady_clock::time_point t = std::chrono::steady_clock::now();
    void mark(const char *what) {
        if (!on) return;
        const auto now = std::chrono::steady_clock::now();
        fprintf(stderr, "[load]   . %-42s %.3fs\n", what, std::chrono::duration<double>(now - t).count());
        t = now;
    }
};
void call_state_create(pf_ctx *ctx);   // pf_call.hip: the (empty) state of the calling pipeline, made with the context
void call_destroy(pf_ctx *ctx);     // pf_call.hip
void cc_destroy(pf_ctx *ctx);       // pf_cc.hip
void gfa_destroy(pf_ctx *ctx);      // pf_gfa.hip
void comm_destroy(pf_ctx *ctx);     // pf_gather.hip
void call_invalidate(pf_ctx *ctx);  // graph or count table replaced
int call_state_arrays(pf_ctx *ctx, uint8_t **flags, uint32_t **plus, uint32_t **minus);   // pf_call.hip: T1 state arrays for a device-side writer
void call_state_resident(pf_ctx *ctx);
enum WsSlot {
    WS_ALN_TEXT = 0, WS_ALN_JOBS, WS_ALN_SMALL, WS_ALN_RETRY, WS_ALN_IDX, WS_ALN_OFIRST, WS_ALN_OCOUNT, WS_ALN_OHITS, WS_ALN_OTEXT,
    WS_ALN_OGAPS, WS_ALN_STTEXT, WS_ALN_STGAPS, WS_ALN_STHITS, WS_ALN_WORK, WS_BFS_REC, WS_BFS_POOL, WS_BFS_SMALL, WS_BFS_DEF, WS_BFS_WLIST,
    WS_BFS_RES_REC, WS_BFS_RES_POOL,
    WS_STR_TEXT, WS_STR_OFF, WS_STR_SUM, WS_STR_OK, WS_STR_MISS,
    WS_BUB_TEXT, WS_BUB_PATHS, WS_BUB_TASKS, WS_BUB_SMALL, WS_BUB_RETRY, WS_BUB_IDX, WS_BUB_RES, WS_BUB_OTEXT, WS_BUB_OSITES,
    WS_BUB_OGROUPS, WS_BUB_OILEN, WS_BUB_SCRATCH, WS_BUB_WORK, WS_BUB_IDX2, WS_CCOV_SUM, WS_CCOV_MIN, WS_CCOV_MAX, WS_CCOV_MISS, WS_GMM_X, WS_GMM_STATE, WS_GMM_PART,
    WS_JOIN_REST, WS_JOIN_REST_N,   // K-COV-JOIN: the look-ups its pipeline hands on (pf_device.hip)
    WS_BUB_LANES,   // K-BUBBLE's five launch workspaces (small, retry, scratch, work, idx2) of lanes 1 .. PF_CALL_LANES - 1 (bub_ws)
    WS_COUNT_ = WS_BUB_LANES + 5 * (PF_CALL_LANES - 1)
};
// K-BUBBLE's launch workspaces: lane 0 uses the slots of old, the other lanes their own
inline int bub_ws(int slot, int lane) {
    if (lane == 0) return slot;
    const int j = slot == WS_BUB_SMALL ? 0 : slot == WS_BUB_RETRY ? 1 : slot == WS_BUB_SCRATCH ? 2 : slot == WS_BUB_WORK ? 3 : 4;
    return WS_BUB_LANES + 5 * (lane - 1) + j;
}
}  // namespace pf
