// K-CC: which traversal records may be committed side by side.
//
// The commits of findSuperBubble (reference src/CDBG.cpp:206-214, 552-846) are applied in unitig order because each one reads
// state earlier ones wrote.  What a record can touch is known before any commit runs (csrc/host/pf_state_ops.hpp): the side of
// its entrance it leaves through, the side of its exit it enters through, both sides of every other vertex of its list, and --
// through partner links, which only ever join two sides one accepted record touched together -- nothing outside that set, with
// two corrections for the reference's release rule ("if (ex->get_plus() == me) set_plus_self(); else set_minus_self();"):
// a unitig whose side is linked to two different sides by accepted records counts as one unit, and so does the exit of a
// rejected traversal.  Records of different connected components of {sides, "touched by one record"} commute; the host layer
// replays every component in record order on its own thread (csrc/host/pf_replay_par.hpp).
//
// The records and vertex lists are in HBM already when K-BFS returns, so the components are found here: a lock-free
// union-find over the 2N sides (roots point downwards: parent[x] <= x, hooked with atomicCAS, path halving), one thread per
// record (a block per record for the few lists the caller walked itself), then one find per record for its label and a stable
// radix sort of the record indices by class = hash(label) % n_classes.  Cumulative over the slices of a pass: components only
// ever merge, and a slice is ordered by the components as they stand once its own records are in.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstring>
#include <rocprim/rocprim.hpp>

#include "host/pf_state_ops.hpp"
#include "pf_ctx.hpp"
#include "pf_device_common.hpp"
#include "pf_scan.hpp"
#include "ploidyfrost_hip.h"

using namespace pf;

#define PF_HIP(call)                                                                         \
    do {                                                                                     \
        hipError_t e_ = (call);                                                              \
        if (e_ != hipSuccess) {                                                              \
            pf::CtxErr{ctx} = std::string(#call) + ": " + hipGetErrorString(e_);                   \
            return PF_ERR_HIP;                                                               \
        }                                                                                    \
    } while (0)

namespace {

constexpr uint32_t CC_NONE = 0xFFFFFFFFu;

__device__ inline uint32_t cc_load(const uint32_t *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

__device__ inline uint32_t cc_find(uint32_t *parent, uint32_t x) {
    for (;;) {
        const uint32_t p = cc_load(parent + x);
        if (p == x) return x;
        const uint32_t g = cc_load(parent + p);
        if (g != p) __hip_atomic_store(parent + x, g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // path halving: g is an ancestor of x whatever others do
        x = p;
    }
}

__device__ inline void cc_unite(uint32_t *parent, uint32_t a, uint32_t b) {
    for (;;) {
        a = cc_find(parent, a);
        b = cc_find(parent, b);
        if (a == b) return;
        const uint32_t hi = a > b ? a : b, lo = a > b ? b : a;
        // only a root is ever hooked, and only under a smaller index: no cycles, and a lost race just retries from the new roots
        if (atomicCAS(parent + hi, hi, lo) == hi) return;
    }
}

__device__ inline bool cc_effective(const pf_bfs_record &r) {
    if (r.outcome > PF_BFS_ACCEPT) return false;   // K-BFS's own marks (deferred to the caller's walkers): the caller passes those as `extra`
    if (r.outcome == PF_BFS_NONE) return r.flag_cycle != 0;
    if (r.outcome == PF_BFS_ACCEPT) return r.n_seen >= 4;
    return true;
}

// an accepted record links `side` to the side `other`
__device__ inline void cc_partner(uint32_t *first, uint8_t *multi, uint32_t side, uint32_t other) {
    const uint32_t old = atomicCAS(first + side, CC_NONE, other);
    if (old != CC_NONE && old != other) multi[side >> 1] = 1;
}

// the part of a record's footprint that does not depend on the list position (one thread of the record's group does it)
// gate: the colored path's (nullptr otherwise) -- an accepted record may mark an endpoint with an incomplete colour set
// NON_SUPER, a write to the whole unitig: such endpoints count as one unit
__device__ inline void cc_endpoints(const pf_bfs_record &r, uint32_t *parent, uint32_t *first, uint8_t *multi, const pfh::ColourGate *gate) {
    const uint32_t s = r.entrance, t = r.exit;
    if (r.outcome != PF_BFS_NONE) cc_unite(parent, s, t ^ 1u);
    if (r.outcome == PF_BFS_ACCEPT) {
        cc_partner(first, multi, s, t ^ 1u);
        cc_partner(first, multi, t ^ 1u, s);
        if (gate) {
            if (gate->incomplete_entrance(s >> 1)) multi[s >> 1] = 1;
            if (gate->incomplete_exit(t >> 1, s >> 1)) multi[t >> 1] = 1;
        }
    }
    if (r.outcome == PF_BFS_REJECT) multi[t >> 1] = 1;
}

__device__ inline void cc_entry(const pf_bfs_record &r, uint32_t w, uint32_t *parent, uint32_t n_sides, uint32_t *bad) {
    if (w >= n_sides) { *bad = 1; return; }
    const bool all_interior = r.outcome == PF_BFS_NONE || r.outcome == PF_BFS_CYCLE_EXIT;   // cycle commits poison every entry
    if (!all_interior && (w == r.entrance || w == r.exit)) return;
    cc_unite(parent, r.entrance, 2 * (w >> 1));
    cc_unite(parent, r.entrance, 2 * (w >> 1) + 1);
}

__global__ void k_cc_init(uint32_t *parent, uint32_t *first, uint8_t *multi, uint32_t n_sides) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n_sides) {
        parent[i] = i;
        first[i] = CC_NONE;
        if ((i & 1) == 0) multi[i >> 1] = 0;
    }
}

// a record whose vertices or list lie outside the graph / the pool is refused (the caller's arrays are not ours to trust)
__device__ inline bool cc_valid(const pf_bfs_record &r, uint32_t n_sides, uint64_t pool_len) {
    return r.entrance < n_sides && (r.outcome == PF_BFS_NONE || r.exit < n_sides) && r.list_off <= pool_len && r.n_list <= pool_len - r.list_off;
}

// one thread per record (lists of the device tiers: at most 128 entries, four on average)
__global__ __launch_bounds__(256) void k_cc_edges(const pf_bfs_record *__restrict__ rec, uint64_t n, const uint32_t *__restrict__ pool, uint64_t pool_len,
                                                  uint32_t n_sides, uint32_t *parent, uint32_t *first, uint8_t *multi, uint32_t *bad,
                                                  pfh::ColourGate gate) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const pf_bfs_record r = rec[i];
    if (r.entrance >= n_sides) { *bad = 1; return; }
    if (!cc_effective(r)) return;
    if (!cc_valid(r, n_sides, pool_len)) { *bad = 1; return; }
    cc_endpoints(r, parent, first, multi, gate.n_colors ? &gate : nullptr);
    const uint32_t *l = pool + r.list_off;
    for (uint32_t q = 0; q < r.n_list; ++q) cc_entry(r, l[q], parent, n_sides, bad);
}

// up to 64 blocks per record (grid y): the long lists of the traversals the caller walked itself
__global__ __launch_bounds__(256) void k_cc_edges_long(const pf_bfs_record *__restrict__ rec, uint64_t n, const uint32_t *__restrict__ pool,
                                                       uint64_t pool_len, uint32_t n_sides, uint32_t *parent, uint32_t *first, uint8_t *multi,
                                                       uint32_t *bad, pfh::ColourGate gate) {
    const uint64_t i = blockIdx.x;
    if (i >= n) return;
    const pf_bfs_record r = rec[i];
    if (!cc_effective(r)) return;
    if (!cc_valid(r, n_sides, pool_len)) { *bad = 1; return; }   // (block-uniform)
    if (threadIdx.x == 0 && blockIdx.y == 0) cc_endpoints(r, parent, first, multi, gate.n_colors ? &gate : nullptr);
    const uint32_t *l = pool + r.list_off;
    // Every entry joins the entry at half its index (a binary tree over the list positions) instead of the entrance: the same
    // component, but the hooks spread over the list instead of all landing on one root (a 36 000-entry list cost 8 ms that way).
    const bool all_interior = r.outcome == PF_BFS_NONE || r.outcome == PF_BFS_CYCLE_EXIT;
    for (uint32_t q = blockIdx.y * blockDim.x + threadIdx.x; q < r.n_list; q += gridDim.y * blockDim.x) {
        const uint32_t w = l[q];
        if (w >= n_sides) { *bad = 1; continue; }
        if (!all_interior && (w == r.entrance || w == r.exit)) continue;   // an endpoint: its side is joined by cc_endpoints
        const uint32_t a = 2 * (w >> 1);
        cc_unite(parent, a, a + 1);
        uint32_t up = r.entrance;
        if (q) {
            const uint32_t pw = l[(q - 1) >> 1];
            if (pw < n_sides) up = (!all_interior && pw == r.entrance) ? r.entrance : (!all_interior && pw == r.exit) ? (r.exit ^ 1u) : 2 * (pw >> 1);
        }
        cc_unite(parent, a, up);
    }
}

__global__ void k_cc_multi(const uint8_t *__restrict__ multi, uint32_t n_unitigs, uint32_t *parent) {
    const uint32_t u = blockIdx.x * blockDim.x + threadIdx.x;
    if (u < n_unitigs && multi[u]) cc_unite(parent, 2 * u, 2 * u + 1);
}

__global__ void k_cc_labels(const pf_bfs_record *__restrict__ rec, uint64_t n, uint32_t *parent, uint32_t n_sides, uint32_t n_classes,
                            uint32_t *__restrict__ labels, uint32_t *__restrict__ cls, uint32_t *__restrict__ idx, uint32_t *hist) {
    __shared__ uint32_t s_hist[1024];
    for (uint32_t c = threadIdx.x; c < n_classes; c += blockDim.x) s_hist[c] = 0;
    __syncthreads();
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
    const uint32_t e = rec[i].entrance;
    const uint32_t lab = e < n_sides ? cc_find(parent, e) : 0u;   // (refused by pf_side_components already)
    labels[i] = lab;
    const uint32_t c = (lab >> 11) % n_classes;   // = pfh::replay_class_of: runs of neighbouring components share a class
    cls[i] = c;
    idx[i] = (uint32_t)i;
    atomicAdd(s_hist + c, 1u);
    }
    __syncthreads();
    for (uint32_t c = threadIdx.x; c < n_classes; c += blockDim.x)
        if (s_hist[c]) atomicAdd(hist + c, s_hist[c]);
}

// ---- the commits on the device ------------------------------------------------------------------------------------------
// With the components known, a component is a unit of sequential work: one THREAD commits the records of one component in
// record order (csrc/host/pf_state_ops.hpp: the same text of the commits as on the host), a million and a half components side
// by side.  A component too large for one thread (the neighbourhood of a chromosome-long traversal: 130 000 list entries on
// the 5 M-unitig graph) is left to the caller, and so is every component that holds a record the caller walked itself.
struct FlagsDevice {
    uint8_t *f2;   // [2u] plus side, [2u + 1] minus side
    uint32_t *plus, *minus;
    __device__ uint32_t link(uint32_t u, bool ps) const { return __hip_atomic_load(ps ? &plus[u] : &minus[u], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
    // (the plus slot of a unitig whose plus side may belong to another thread's component: cannot come out true then)
    __device__ bool plus_points_to(uint32_t ex, uint32_t me) const { return __hip_atomic_load(&plus[ex], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == me + 1; }
    __device__ void set_link(uint32_t u, bool ps, uint32_t v, bool real) {
        __hip_atomic_store(ps ? &plus[u] : &minus[u], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        uint8_t *b = f2 + 2 * (size_t)u + (ps ? 0 : 1);
        *b = real ? (uint8_t)(*b | pfh::state_bits::S_LINK) : (uint8_t)(*b & ~pfh::state_bits::S_LINK);
    }
    __device__ void mark_strict(uint32_t u, bool ps) { f2[2 * (size_t)u + (ps ? 0 : 1)] |= pfh::state_bits::S_STRICT; }
    __device__ void mark_complex(uint32_t u, bool ps) { f2[2 * (size_t)u + (ps ? 0 : 1)] |= pfh::state_bits::S_COMPLEX; }
    __device__ bool non_super(uint32_t u, bool ps) const { return (f2[2 * (size_t)u + (ps ? 0 : 1)] & pfh::state_bits::S_NON_SUPER) != 0; }
    __device__ void set_non_super(uint32_t u) {
        f2[2 * (size_t)u] |= pfh::state_bits::S_NON_SUPER;
        f2[2 * (size_t)u + 1] |= pfh::state_bits::S_NON_SUPER;
    }
    __device__ void begin_record(const pf_bfs_record &) {}
};

__global__ void k_replay_label(const pf_bfs_record *__restrict__ rec, uint64_t n, uint32_t *parent, uint32_t n_sides, uint32_t *__restrict__ labels,
                               uint32_t *__restrict__ idx, uint32_t *work) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const pf_bfs_record r = rec[i];
    const uint32_t lab = r.entrance < n_sides ? cc_find(parent, r.entrance) : 0u;
    labels[i] = lab;
    idx[i] = (uint32_t)i;
    if (cc_effective(r)) atomicAdd(work + lab, r.n_list + 1u);
}

__global__ void k_replay_force_big(const pf_bfs_record *__restrict__ xrec, uint64_t n, uint32_t *parent, uint32_t n_sides, uint8_t *big) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n && xrec[i].entrance < n_sides) big[cc_find(parent, xrec[i].entrance)] = 1;
}

// keys / vals: the records sorted by component label (stable: ascending record index inside a component)
template <class Col>
__global__ __launch_bounds__(256) void k_replay_small(const uint32_t *__restrict__ keys, const uint32_t *__restrict__ vals, uint64_t n,
                                                      const pf_bfs_record *__restrict__ rec, const uint32_t *__restrict__ pool, const uint8_t *__restrict__ big,
                                                      const uint32_t *__restrict__ work, uint32_t limit, FlagsDevice acc, uint32_t complex_size, Col col) {
    const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    const uint32_t key = keys[t];
    if (t && keys[t - 1] == key) return;   // not the head of its component
    if (big[key] || work[key] > limit) return;
    pfh::Commits<FlagsDevice, Col> cm{acc, complex_size, col};
    for (uint64_t j = t; j < n && keys[j] == key; ++j) {
        const pf_bfs_record r = rec[vals[j]];
        if (!cm.gate_open(r.entrance)) continue;
        cm.replay(r, pool + r.list_off);
    }
}

__global__ void k_replay_flag_big(const pf_bfs_record *__restrict__ rec, uint64_t n, const uint32_t *__restrict__ labels, const uint8_t *__restrict__ big,
                                  const uint32_t *__restrict__ work, uint32_t limit, uint8_t *__restrict__ flag) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t lab = labels[i];
    flag[i] = (cc_effective(rec[i]) && (big[lab] || work[lab] > limit)) ? 1 : 0;
}

__global__ void k_replay_big_sizes(const uint32_t *__restrict__ big_idx, uint64_t n_big, const pf_bfs_record *__restrict__ rec, uint64_t *__restrict__ sizes) {
    const uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j <= n_big) sizes[j] = j < n_big ? (uint64_t)rec[big_idx[j]].n_list : 0;
}

__global__ void k_replay_big_gather(const uint32_t *__restrict__ big_idx, uint64_t n_big, const pf_bfs_record *__restrict__ rec, const uint32_t *__restrict__ pool,
                                    const uint64_t *__restrict__ off, pf_bfs_record *__restrict__ out_rec, uint32_t *__restrict__ out_pool) {
    const uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n_big) return;
    pf_bfs_record r = rec[big_idx[j]];
    const uint32_t *l = pool + r.list_off;
    r.list_off = off[j];
    out_rec[j] = r;
    for (uint32_t q = 0; q < r.n_list; ++q) out_pool[off[j] + q] = l[q];
}

__global__ void k_replay_patch(const uint32_t *__restrict__ sides, const uint32_t *__restrict__ links, const uint8_t *__restrict__ bytes, uint64_t n,
                               uint32_t n_sides, uint32_t *plus, uint32_t *minus, uint8_t *f2) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t s = sides[i];
    if (s >= n_sides) return;
    ((s & 1) ? minus : plus)[s >> 1] = links[i];
    f2[s] = bytes[i];
}

__global__ void k_replay_merge(const uint8_t *__restrict__ f2, uint32_t n_unitigs, uint8_t *__restrict__ flags) {
    const uint32_t u = blockIdx.x * blockDim.x + threadIdx.x;
    if (u < n_unitigs) flags[u] = pfh::FlagsPerSide::merged(f2[2 * (size_t)u], f2[2 * (size_t)u + 1]);
}

struct CcState {
    uint32_t *parent = nullptr, *first = nullptr;
    uint8_t *multi = nullptr;
    uint32_t *bad = nullptr;
    uint32_t n_unitigs = 0;
    // per call
    uint32_t *labels = nullptr, *cls = nullptr, *idx = nullptr, *cls2 = nullptr, *idx2 = nullptr, *hist = nullptr;
    uint64_t cap = 0;
    void *sort_tmp = nullptr;
    size_t sort_tmp_bytes = 0;
    pf_bfs_record *up_rec = nullptr;
    uint64_t up_rec_cap = 0;
    uint32_t *up_pool = nullptr;
    uint64_t up_pool_cap = 0;
    uint64_t n_last = 0;
    // the commits on the device (pf_replay_device)
    uint8_t *big = nullptr;        // [2N] by root: the component is left to the caller
    uint32_t *work = nullptr;      // [2N] by root: list entries + records of the component
    uint8_t *f2 = nullptr;         // [2N] per-side flag bytes during the commits
    uint32_t *keys = nullptr, *keys2 = nullptr, *vals = nullptr, *vals2 = nullptr;   // records sorted by component
    uint64_t sort_cap = 0;
    bool ready = false;            // parent / first / multi have been initialised (pf_find_reserve only takes them)
    uint32_t *big_idx = nullptr;   // indices of the records of the components left to the caller, ascending
    pf_bfs_record *big_rec = nullptr;
    uint32_t *big_pool = nullptr;
    uint64_t big_cap = 0, big_pool_cap = 0, n_big = 0, big_entries = 0;
    // colored path (pf_replay_set_colours): what the colour gate of the accept commit reads, per unitig
    uint64_t *full_mask = nullptr, *size_total = nullptr;
    uint32_t *n_full_enc = nullptr;
    uint32_t n_colors = 0;
    pf_bfs_record *xrec = nullptr;   // (device copy of the last call's extra records, kept for pf_replay_device: t_xrec below)
    uint64_t n_xrec = 0;
    uint64_t added_call = 0;   // the K-BFS call whose device-resident records are already in the union-find
    bool labelled = false;
    // scratch of the calls below, kept from pass to pass (a hipMalloc / hipFree pair per temporary is a device-wide
    // synchronisation and a tenth of a millisecond each: a dozen of them were a fifth of findSuperBubble's time)
    pf_bfs_record *t_xrec = nullptr;
    uint32_t *t_xpool = nullptr, *t_sides = nullptr, *t_links = nullptr;
    uint8_t *t_flag = nullptr, *t_sel = nullptr, *t_scan = nullptr, *t_bytes = nullptr;
    uint64_t *t_cnt = nullptr, *t_sz = nullptr, *t_off = nullptr;
    uint64_t t_xrec_cap = 0, t_xpool_cap = 0, t_sides_cap = 0, t_links_cap = 0, t_flag_cap = 0, t_sel_cap = 0, t_scan_cap = 0, t_bytes_cap = 0, t_cnt_cap = 0,
             t_sz_cap = 0, t_off_cap = 0;
    void release() {
        for (void *p : {(void *)t_xrec, (void *)t_xpool, (void *)t_sides, (void *)t_links, (void *)t_flag, (void *)t_sel, (void *)t_scan, (void *)t_bytes, (void *)t_cnt,
                        (void *)t_sz, (void *)t_off})
            if (p) (void)hipFree(p);
        for (void *p : {(void *)big, (void *)work, (void *)f2, (void *)keys, (void *)keys2, (void *)vals, (void *)vals2, (void *)big_idx, (void *)big_rec,
                        (void *)big_pool, (void *)full_mask, (void *)size_total, (void *)n_full_enc})
            if (p) (void)hipFree(p);
        for (void *p : {(void *)parent, (void *)first, (void *)multi, (void *)bad, (void *)labels, (void *)cls, (void *)idx, (void *)cls2, (void *)idx2, (void *)hist,
                        sort_tmp, (void *)up_rec, (void *)up_pool})
            if (p) (void)hipFree(p);
        *this = CcState();
    }
};

template <class T>
bool grow(T *&p, uint64_t &cap, uint64_t want) {
    if (p && cap >= want) return true;
    if (p) (void)hipFree(p);
    p = nullptr;
    cap = want + want / 4 + 64;
    return hipMalloc(reinterpret_cast<void **>(&p), cap * sizeof(T)) == hipSuccess;
}

struct CcState;
pfh::ColourGate gate_of(const pf_ctx *ctx, const CcState *S);

bool is_device(const void *p) {
    hipPointerAttribute_t at;
    const bool dev = hipPointerGetAttributes(&at, p) == hipSuccess && at.type == hipMemoryTypeDevice;
    (void)hipGetLastError();
    return dev;
}

pfh::ColourGate gate_of(const pf_ctx *ctx, const CcState *S) {
    pfh::ColourGate g;
    if (!S->n_colors) return g;   // (n_colors == 0: no gate)
    g.n_colors = S->n_colors;
    g.words = (S->n_colors + 63) / 64;
    g.k = ctx->k;
    g.len_bp = ctx->d_len;
    g.full_mask = S->full_mask;
    g.size_total = S->size_total;
    g.n_full_enc = S->n_full_enc;
    g.succ = ctx->d_succ;
    return g;
}

}  // namespace

namespace pf {
void cc_destroy(pf_ctx *ctx) {
    if (!ctx->cc) return;
    static_cast<CcState *>(ctx->cc)->release();
    delete static_cast<CcState *>(ctx->cc);
    ctx->cc = nullptr;
}
}  // namespace pf

// the per-graph arrays of K-CC
static int cc_graph_arrays(pf_ctx *ctx, CcState *S, uint32_t N) {
    const uint32_t n_sides = 2 * N;
    // (the colour gate's arrays belong to the graph, not to a pass: they survive)
    uint64_t *fm = S->full_mask, *stt = S->size_total;
    uint32_t *nf = S->n_full_enc;
    const uint32_t nc = S->n_colors;
    S->full_mask = S->size_total = nullptr;
    S->n_full_enc = nullptr;
    S->release();   // (resets the fields, the object stays)
    S->full_mask = fm; S->size_total = stt; S->n_full_enc = nf; S->n_colors = nc;
    if (hipMalloc(reinterpret_cast<void **>(&S->bad), 16) != hipSuccess ||
        hipMalloc(reinterpret_cast<void **>(&S->parent), (size_t)n_sides * 4 + 4) != hipSuccess ||
        hipMalloc(reinterpret_cast<void **>(&S->first), (size_t)n_sides * 4 + 4) != hipSuccess ||
        hipMalloc(reinterpret_cast<void **>(&S->multi), (size_t)N + 4) != hipSuccess) {
        pf::CtxErr{ctx} = "pf_side_components: out of device memory";
        return PF_ERR_HIP;
    }
    S->n_unitigs = N;
    S->ready = false;
    return PF_OK;
}

// the buffers of pf_replay_device for n records
static int cc_replay_arrays(pf_ctx *ctx, CcState *S, uint64_t n) {
    const uint32_t n_sides = 2 * S->n_unitigs;
    if (!S->big) {
        PF_HIP(hipMalloc(reinterpret_cast<void **>(&S->big), (size_t)n_sides + 4));
        PF_HIP(hipMalloc(reinterpret_cast<void **>(&S->work), (size_t)n_sides * 4 + 4));
        PF_HIP(hipMalloc(reinterpret_cast<void **>(&S->f2), (size_t)n_sides + 4));
    }
    if (S->sort_cap < n + 1) {
        for (uint32_t **p : {&S->keys, &S->keys2, &S->vals, &S->vals2, &S->big_idx})
            if (*p) { (void)hipFree(*p); *p = nullptr; }
        S->sort_cap = n + n / 4 + 64;
        for (uint32_t **p : {&S->keys, &S->keys2, &S->vals, &S->vals2, &S->big_idx})
            PF_HIP(hipMalloc(reinterpret_cast<void **>(p), S->sort_cap * 4));
    }
    unsigned bits = 1;
    while ((1ull << bits) < n_sides) ++bits;
    size_t need = 0;
    PF_HIP(rocprim::radix_sort_pairs(nullptr, need, S->keys, S->keys2, S->vals, S->vals2, (size_t)n, 0, bits, ctx->stream));
    if (need > S->sort_tmp_bytes) {
        if (S->sort_tmp) (void)hipFree(S->sort_tmp);
        S->sort_tmp = nullptr;
        PF_HIP(hipMalloc(&S->sort_tmp, need + 256));
        S->sort_tmp_bytes = need + 256;
    }
    return PF_OK;
}

extern "C" {

int pf_find_reserve(pf_ctx *ctx, uint64_t n_records) {
    if (!ctx || !ctx->has_adj || (n_records >> 32)) return PF_ERR_ARG;
    PF_HIP(hipSetDevice(ctx->device));
    if (!ctx->cc) ctx->cc = new CcState();
    CcState *S = static_cast<CcState *>(ctx->cc);
    if (S->n_unitigs != ctx->N || !S->parent) {
        const int ga = cc_graph_arrays(ctx, S, ctx->N);
        if (ga != PF_OK) return ga;
    }
    const int ra = cc_replay_arrays(ctx, S, std::max<uint64_t>(n_records, 1));
    if (ra != PF_OK) return ra;
    if (!grow(S->t_flag, S->t_flag_cap, n_records + 8) || !grow(S->t_cnt, S->t_cnt_cap, 2)) { pf::CtxErr{ctx} = "pf_find_reserve: out of device memory"; return PF_ERR_HIP; }
    uint8_t *flags = nullptr;
    uint32_t *plus = nullptr, *minus = nullptr;
    return call_state_arrays(ctx, &flags, &plus, &minus);
}

int pf_side_components(pf_ctx *ctx, int reset, const pf_bfs_record *records, uint64_t n_records, const uint32_t *pool, uint64_t pool_len,
                       const pf_bfs_record *extra, uint64_t n_extra, const uint32_t *extra_pool, uint64_t extra_pool_len) {
    if (!ctx || !ctx->has_adj) return PF_ERR_ARG;
    if ((records == nullptr) != (pool == nullptr) || (n_extra && (!extra || (!extra_pool && extra_pool_len)))) return PF_ERR_ARG;
    PF_HIP(hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    if (!ctx->cc) ctx->cc = new CcState();
    CcState *S = static_cast<CcState *>(ctx->cc);
    const uint32_t N = ctx->N;
    const uint32_t n_sides = 2 * N;
    if (S->n_unitigs != N || !S->parent) {
        const int ga = cc_graph_arrays(ctx, S, N);
        if (ga != PF_OK) return ga;
        reset = 1;
    }
    if (!S->ready) reset = 1;   // (arrays taken by pf_find_reserve: never initialised)
    S->ready = true;
    if (reset) {
        k_cc_init<<<(n_sides + 255) / 256, 256, 0, st>>>(S->parent, S->first, S->multi, n_sides);
        S->xrec = nullptr;   // (it points into t_xrec)
        S->n_xrec = 0;
    }
    S->labelled = false;
    PF_HIP(hipMemsetAsync(S->bad, 0, 4, st));
    // the records: the ones K-BFS left in its workspace, or the caller's
    const pf_bfs_record *d_rec = nullptr;
    const uint32_t *d_pool = nullptr;
    if (!records && n_records == 0) {
        // a graph without a candidate entrance: nothing to add
    } else if (!records) {
        if (ctx->bfs_last_n != n_records || !ctx->bfs_last_rec) { pf::CtxErr{ctx} = "pf_side_components: no records of that length from the last K-BFS call"; return PF_ERR_ARG; }
        d_rec = ctx->bfs_last_rec;
        d_pool = ctx->bfs_last_pool;
        pool_len = ctx->bfs_last_pool_len;
    } else if (is_device(records)) {
        d_rec = records;
        d_pool = pool;
    } else {
        if (!grow(S->up_rec, S->up_rec_cap, n_records + n_extra + 1) || !grow(S->up_pool, S->up_pool_cap, pool_len + extra_pool_len + 1)) {
            pf::CtxErr{ctx} = "pf_side_components: out of device memory";
            return PF_ERR_HIP;
        }
        PF_HIP(hipMemcpyAsync(S->up_rec, records, n_records * sizeof(pf_bfs_record), hipMemcpyHostToDevice, st));
        PF_HIP(hipMemcpyAsync(S->up_pool, pool, pool_len * 4, hipMemcpyHostToDevice, st));
        d_rec = S->up_rec;
        d_pool = S->up_pool;
    }
    const pfh::ColourGate gate = gate_of(ctx, S);
    // a second call for the same K-BFS call (the caller adds the traversals it walked itself once they are done) adds only those
    const bool again = !records && !reset && S->added_call == ctx->bfs_call_id;
    if (!records) S->added_call = ctx->bfs_call_id;
    if (n_records && !again) {
        k_cc_edges<<<(unsigned)((n_records + 255) / 256), 256, 0, st>>>(d_rec, n_records, d_pool, pool_len, n_sides, S->parent, S->first, S->multi, S->bad, gate);
        PF_HIP(hipGetLastError());
    }
    if (n_extra) {
        // traversals the caller walked itself: their records (list_off relative to extra_pool) contribute edges only
        if (!grow(S->t_xrec, S->t_xrec_cap, n_extra) || !grow(S->t_xpool, S->t_xpool_cap, extra_pool_len + 1)) { pf::CtxErr{ctx} = "pf_side_components: out of device memory"; return PF_ERR_HIP; }
        pf_bfs_record *d_xrec = S->t_xrec;
        uint32_t *d_xpool = S->t_xpool;
        PF_HIP(hipMemcpyAsync(d_xrec, extra, n_extra * sizeof(pf_bfs_record), hipMemcpyDefault, st));
        if (extra_pool_len) PF_HIP(hipMemcpyAsync(d_xpool, extra_pool, extra_pool_len * 4, hipMemcpyDefault, st));
        k_cc_edges_long<<<dim3((unsigned)n_extra, 64), 256, 0, st>>>(d_xrec, n_extra, d_xpool, extra_pool_len, n_sides, S->parent, S->first, S->multi, S->bad, gate);
        PF_HIP(hipGetLastError());
        PF_HIP(hipStreamSynchronize(st));   // (the caller's buffers may go)
        // the records themselves stay where they are: pf_replay_device leaves their components to the caller, who holds their lists
        S->xrec = S->t_xrec;
        S->n_xrec = n_extra;
    }
    k_cc_multi<<<(N + 255) / 256, 256, 0, st>>>(S->multi, N, S->parent);
    PF_HIP(hipGetLastError());
    uint32_t bad = 0;
    PF_HIP(hipMemcpyAsync(&bad, S->bad, 4, hipMemcpyDeviceToHost, st));
    PF_HIP(hipStreamSynchronize(st));
    if (bad) { pf::CtxErr{ctx} = "pf_side_components: a record names a vertex outside the graph or a list outside the pool"; S->n_last = 0; return PF_ERR_ARG; }
    S->n_last = n_records;
    ctx->cc_rec = d_rec;
    ctx->cc_pool = d_pool;
    return PF_OK;
}

int pf_replay_order(pf_ctx *ctx, uint32_t n_classes, uint32_t *order, uint32_t *class_off, uint32_t *labels) {
    if (!ctx || !ctx->cc || !order || !class_off || n_classes == 0 || n_classes > 1024) return PF_ERR_ARG;
    CcState *S = static_cast<CcState *>(ctx->cc);
    PF_HIP(hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    const uint64_t n = S->n_last;
    if (n >> 32) { pf::CtxErr{ctx} = "pf_replay_order: more than 2^32 records"; return PF_ERR_ARG; }
    uint64_t cap = S->cap;
    if (!S->labels || S->cap < n + 1) {
        for (uint32_t **p : {&S->labels, &S->cls, &S->idx, &S->cls2, &S->idx2})
            if (*p) { (void)hipFree(*p); *p = nullptr; }
        cap = n + n / 4 + 64;
        for (uint32_t **p : {&S->labels, &S->cls, &S->idx, &S->cls2, &S->idx2})
            if (hipMalloc(reinterpret_cast<void **>(p), cap * 4) != hipSuccess) { pf::CtxErr{ctx} = "pf_replay_order: out of device memory"; return PF_ERR_HIP; }
        S->cap = cap;
    }
    if (!S->hist) PF_HIP(hipMalloc(reinterpret_cast<void **>(&S->hist), 1025 * 4));
    PF_HIP(hipMemsetAsync(S->hist, 0, 1025 * 4, st));
    if (n) {
        k_cc_labels<<<(unsigned)((n + 255) / 256), 256, 0, st>>>(static_cast<const pf_bfs_record *>(ctx->cc_rec), n, S->parent, 2 * S->n_unitigs, n_classes,
                                                                 S->labels, S->cls, S->idx, S->hist);
        PF_HIP(hipGetLastError());
        unsigned bits = 1;
        while ((1u << bits) < n_classes) ++bits;
        size_t need = 0;
        PF_HIP(rocprim::radix_sort_pairs(nullptr, need, S->cls, S->cls2, S->idx, S->idx2, (size_t)n, 0, bits, st));
        if (need > S->sort_tmp_bytes) {
            if (S->sort_tmp) (void)hipFree(S->sort_tmp);
            S->sort_tmp = nullptr;
            PF_HIP(hipMalloc(&S->sort_tmp, need + 256));
            S->sort_tmp_bytes = need + 256;
        }
        size_t have = S->sort_tmp_bytes;
        PF_HIP(rocprim::radix_sort_pairs(S->sort_tmp, have, S->cls, S->cls2, S->idx, S->idx2, (size_t)n, 0, bits, st));
        PF_HIP(hipMemcpyAsync(order, S->idx2, n * 4, hipMemcpyDefault, st));
        if (labels) PF_HIP(hipMemcpyAsync(labels, S->labels, n * 4, hipMemcpyDefault, st));
    }
    uint32_t hist[1025];
    PF_HIP(hipMemcpyAsync(hist, S->hist, (size_t)n_classes * 4, hipMemcpyDeviceToHost, st));
    PF_HIP(hipStreamSynchronize(st));
    class_off[0] = 0;
    for (uint32_t c = 0; c < n_classes; ++c) class_off[c + 1] = class_off[c] + hist[c];
    S->labelled = true;
    return PF_OK;
}

// colored path: what the colour gate of the accept commit reads (src/CCDBG.cpp:2530-2621), per unitig; n_colors == 0 removes it
int pf_replay_set_colours(pf_ctx *ctx, uint32_t n_colors, const uint64_t *full_mask, const uint64_t *size_total, const uint32_t *n_full_enc) {
    if (!ctx || !ctx->has_adj || n_colors > PF_MAX_COLORS || (n_colors && (!full_mask || !size_total || !n_full_enc))) return PF_ERR_ARG;
    PF_HIP(hipSetDevice(ctx->device));
    if (!ctx->cc) ctx->cc = new CcState();
    CcState *S = static_cast<CcState *>(ctx->cc);
    for (void **p : {(void **)&S->full_mask, (void **)&S->size_total, (void **)&S->n_full_enc})
        if (*p) { (void)hipFree(*p); *p = nullptr; }
    S->n_colors = 0;
    if (!n_colors) return PF_OK;
    const size_t N = ctx->N;
    const size_t CW = (n_colors + 63) / 64;   // 64-bit words of a colour set
    PF_HIP(hipMalloc(reinterpret_cast<void **>(&S->full_mask), N * CW * 8));
    PF_HIP(hipMalloc(reinterpret_cast<void **>(&S->size_total), N * 8));
    PF_HIP(hipMalloc(reinterpret_cast<void **>(&S->n_full_enc), N * 4));
    PF_HIP(hipMemcpy(S->full_mask, full_mask, N * CW * 8, hipMemcpyDefault));
    PF_HIP(hipMemcpy(S->size_total, size_total, N * 8, hipMemcpyDefault));
    PF_HIP(hipMemcpy(S->n_full_enc, n_full_enc, N * 4, hipMemcpyDefault));
    S->n_colors = n_colors;
    return PF_OK;
}

// ---- the commits on the device: see the kernels above ----
int pf_replay_device(pf_ctx *ctx, uint32_t complex_size, uint32_t small_limit, uint64_t *n_big, uint64_t *big_entries) {
    if (!ctx || !ctx->cc || !n_big || !big_entries) return PF_ERR_ARG;
    CcState *S = static_cast<CcState *>(ctx->cc);
    if (!S->parent || (!ctx->cc_rec && S->n_last)) { pf::CtxErr{ctx} = "pf_replay_device: pf_side_components first"; return PF_ERR_ARG; }
    PF_HIP(hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    const uint32_t N = S->n_unitigs, n_sides = 2 * N;
    const uint64_t n = S->n_last;
    if (n >> 32) { pf::CtxErr{ctx} = "pf_replay_device: more than 2^32 records"; return PF_ERR_ARG; }
    const pf_bfs_record *rec = static_cast<const pf_bfs_record *>(ctx->cc_rec);
    const uint32_t *pool = ctx->cc_pool;
    {
        const int ra = cc_replay_arrays(ctx, S, std::max<uint64_t>(n, 1));
        if (ra != PF_OK) return ra;
    }
    uint8_t *flags = nullptr;
    uint32_t *plus = nullptr, *minus = nullptr;
    const int sa = call_state_arrays(ctx, &flags, &plus, &minus);
    if (sa != PF_OK) return sa;
    PF_HIP(hipMemsetAsync(S->big, 0, n_sides, st));
    PF_HIP(hipMemsetAsync(S->work, 0, (size_t)n_sides * 4, st));
    PF_HIP(hipMemsetAsync(S->f2, 0, n_sides, st));
    PF_HIP(hipMemsetAsync(plus, 0, (size_t)N * 4, st));
    PF_HIP(hipMemsetAsync(minus, 0, (size_t)N * 4, st));
    *n_big = *big_entries = 0;
    S->n_big = S->big_entries = 0;
    if (n == 0) return PF_OK;
    const unsigned grid = (unsigned)((n + 255) / 256);
    k_replay_label<<<grid, 256, 0, st>>>(rec, n, S->parent, n_sides, S->keys, S->vals, S->work);
    if (S->n_xrec) k_replay_force_big<<<(unsigned)((S->n_xrec + 255) / 256), 256, 0, st>>>(S->xrec, S->n_xrec, S->parent, n_sides, S->big);
    PF_HIP(hipGetLastError());
    unsigned bits = 1;
    while ((1ull << bits) < n_sides) ++bits;
    size_t have = S->sort_tmp_bytes;   // (sized by cc_replay_arrays)
    PF_HIP(rocprim::radix_sort_pairs(S->sort_tmp, have, S->keys, S->keys2, S->vals, S->vals2, (size_t)n, 0, bits, st));
    // the records left to the caller, ascending (gathered first: the caller commits them while the device commits the rest)
    const char *oom = "pf_replay_device: out of device memory";
    if (!grow(S->t_flag, S->t_flag_cap, n + 8) || !grow(S->t_cnt, S->t_cnt_cap, 2)) { pf::CtxErr{ctx} = oom; return PF_ERR_HIP; }
    k_replay_flag_big<<<grid, 256, 0, st>>>(rec, n, S->keys, S->big, S->work, small_limit, S->t_flag);
    if (!grow(S->t_sel, S->t_sel_cap, scan_scratch_bytes(n))) { pf::CtxErr{ctx} = oom; return PF_ERR_HIP; }
    PF_HIP(select_flagged_u8(S->t_flag, S->big_idx, nullptr, S->t_cnt, n, S->t_sel, st));
    uint64_t nb = 0;
    PF_HIP(hipMemcpyAsync(&nb, S->t_cnt, 8, hipMemcpyDeviceToHost, st));
    PF_HIP(hipStreamSynchronize(st));
    nb &= 0xFFFFFFFFull;   // (the selector counts in 32 bits)
    uint64_t entries = 0;
    if (nb) {
        if (!grow(S->t_sz, S->t_sz_cap, nb + 1) || !grow(S->t_off, S->t_off_cap, nb + 1)) { pf::CtxErr{ctx} = oom; return PF_ERR_HIP; }
        k_replay_big_sizes<<<(unsigned)((nb + 1 + 255) / 256), 256, 0, st>>>(S->big_idx, nb, rec, S->t_sz);
        if (!grow(S->t_scan, S->t_scan_cap, scan_scratch_bytes(nb + 1))) { pf::CtxErr{ctx} = oom; return PF_ERR_HIP; }
        PF_HIP(scan_exclusive_u64(S->t_sz, S->t_off, nb + 1, S->t_scan, st));
        PF_HIP(hipMemcpyAsync(&entries, S->t_off + nb, 8, hipMemcpyDeviceToHost, st));
        PF_HIP(hipStreamSynchronize(st));
        if (S->big_cap < nb) {
            if (S->big_rec) (void)hipFree(S->big_rec);
            S->big_rec = nullptr;
            S->big_cap = nb + nb / 4 + 64;
            PF_HIP(hipMalloc(reinterpret_cast<void **>(&S->big_rec), S->big_cap * sizeof(pf_bfs_record)));
        }
        if (S->big_pool_cap < entries + 1) {
            if (S->big_pool) (void)hipFree(S->big_pool);
            S->big_pool = nullptr;
            S->big_pool_cap = entries + entries / 4 + 64;
            PF_HIP(hipMalloc(reinterpret_cast<void **>(&S->big_pool), S->big_pool_cap * 4));
        }
        k_replay_big_gather<<<(unsigned)((nb + 255) / 256), 256, 0, st>>>(S->big_idx, nb, rec, pool, S->t_off, S->big_rec, S->big_pool);
        PF_HIP(hipGetLastError());
        PF_HIP(hipStreamSynchronize(st));
    }
    S->n_big = nb;
    S->big_entries = entries;
    *n_big = nb;
    *big_entries = entries;
    // the small components: left in flight on the context's stream -- disjoint from what the caller commits meanwhile; the
    // patch of pf_replay_finish queues behind them and waits
    FlagsDevice acc{S->f2, plus, minus};
    if (S->n_colors) k_replay_small<pfh::ColourGate><<<grid, 256, 0, st>>>(S->keys2, S->vals2, n, rec, pool, S->big, S->work, small_limit, acc, complex_size, gate_of(ctx, S));
    else k_replay_small<pfh::NoColours><<<grid, 256, 0, st>>>(S->keys2, S->vals2, n, rec, pool, S->big, S->work, small_limit, acc, complex_size, pfh::NoColours{});
    PF_HIP(hipGetLastError());
    return PF_OK;
}

int pf_replay_big_fetch(pf_ctx *ctx, uint32_t *index, pf_bfs_record *records, uint32_t *pool) {
    if (!ctx || !ctx->cc) return PF_ERR_ARG;
    CcState *S = static_cast<CcState *>(ctx->cc);
    PF_HIP(hipSetDevice(ctx->device));
    if (S->n_big) {
        if (index) PF_HIP(hipMemcpy(index, S->big_idx, S->n_big * 4, hipMemcpyDeviceToHost));
        if (records) PF_HIP(hipMemcpy(records, S->big_rec, S->n_big * sizeof(pf_bfs_record), hipMemcpyDeviceToHost));
        if (pool && S->big_entries) PF_HIP(hipMemcpy(pool, S->big_pool, S->big_entries * 4, hipMemcpyDeviceToHost));
    }
    return PF_OK;
}

int pf_replay_finish(pf_ctx *ctx, const uint32_t *sides, const uint32_t *links, const uint8_t *side_flags, uint64_t n_patch) {
    if (!ctx || !ctx->cc || (n_patch && (!sides || !links || !side_flags))) return PF_ERR_ARG;
    CcState *S = static_cast<CcState *>(ctx->cc);
    if (!S->f2) { pf::CtxErr{ctx} = "pf_replay_finish: pf_replay_device first"; return PF_ERR_ARG; }
    PF_HIP(hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    const uint32_t N = S->n_unitigs;
    uint8_t *flags = nullptr;
    uint32_t *plus = nullptr, *minus = nullptr;
    const int sa = call_state_arrays(ctx, &flags, &plus, &minus);   // (the same buffers: contents are kept)
    if (sa != PF_OK) return sa;
    if (n_patch) {
        if (!grow(S->t_sides, S->t_sides_cap, n_patch) || !grow(S->t_links, S->t_links_cap, n_patch) || !grow(S->t_bytes, S->t_bytes_cap, n_patch)) {
            pf::CtxErr{ctx} = "pf_replay_finish: out of device memory";
            return PF_ERR_HIP;
        }
        PF_HIP(hipMemcpyAsync(S->t_sides, sides, n_patch * 4, hipMemcpyDefault, st));
        PF_HIP(hipMemcpyAsync(S->t_links, links, n_patch * 4, hipMemcpyDefault, st));
        PF_HIP(hipMemcpyAsync(S->t_bytes, side_flags, n_patch, hipMemcpyDefault, st));
        k_replay_patch<<<(unsigned)((n_patch + 255) / 256), 256, 0, st>>>(S->t_sides, S->t_links, S->t_bytes, n_patch, 2 * N, plus, minus, S->f2);
        PF_HIP(hipGetLastError());
    }
    k_replay_merge<<<(N + 255) / 256, 256, 0, st>>>(S->f2, N, flags);
    PF_HIP(hipGetLastError());
    PF_HIP(hipStreamSynchronize(st));
    call_state_resident(ctx);
    return PF_OK;
}

}  // extern "C"
