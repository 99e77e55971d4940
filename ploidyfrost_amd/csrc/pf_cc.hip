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

#include "pf_ctx.hpp"
#include "pf_device_common.hpp"
#include "ploidyfrost_hip.h"

using namespace pf;

#define PF_HIP(call)                                                                         \
    do {                                                                                     \
        hipError_t e_ = (call);                                                              \
        if (e_ != hipSuccess) {                                                              \
            ctx->err = std::string(#call) + ": " + hipGetErrorString(e_);                   \
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
__device__ inline void cc_endpoints(const pf_bfs_record &r, uint32_t *parent, uint32_t *first, uint8_t *multi) {
    const uint32_t s = r.entrance, t = r.exit;
    if (r.outcome != PF_BFS_NONE) cc_unite(parent, s, t ^ 1u);
    if (r.outcome == PF_BFS_ACCEPT) {
        cc_partner(first, multi, s, t ^ 1u);
        cc_partner(first, multi, t ^ 1u, s);
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
                                                  uint32_t n_sides, uint32_t *parent, uint32_t *first, uint8_t *multi, uint32_t *bad) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const pf_bfs_record r = rec[i];
    if (r.entrance >= n_sides) { *bad = 1; return; }
    if (!cc_effective(r)) return;
    if (!cc_valid(r, n_sides, pool_len)) { *bad = 1; return; }
    cc_endpoints(r, parent, first, multi);
    const uint32_t *l = pool + r.list_off;
    for (uint32_t q = 0; q < r.n_list; ++q) cc_entry(r, l[q], parent, n_sides, bad);
}

// up to 64 blocks per record (grid y): the long lists of the traversals the caller walked itself
__global__ __launch_bounds__(256) void k_cc_edges_long(const pf_bfs_record *__restrict__ rec, uint64_t n, const uint32_t *__restrict__ pool,
                                                       uint64_t pool_len, uint32_t n_sides, uint32_t *parent, uint32_t *first, uint8_t *multi,
                                                       uint32_t *bad) {
    const uint64_t i = blockIdx.x;
    if (i >= n) return;
    const pf_bfs_record r = rec[i];
    if (!cc_effective(r)) return;
    if (!cc_valid(r, n_sides, pool_len)) { *bad = 1; return; }   // (block-uniform)
    if (threadIdx.x == 0 && blockIdx.y == 0) cc_endpoints(r, parent, first, multi);
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
    uint64_t added_call = 0;   // the K-BFS call whose device-resident records are already in the union-find
    bool labelled = false;
    void release() {
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

bool is_device(const void *p) {
    hipPointerAttribute_t at;
    const bool dev = hipPointerGetAttributes(&at, p) == hipSuccess && at.type == hipMemoryTypeDevice;
    (void)hipGetLastError();
    return dev;
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

extern "C" {

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
        S->release();   // (resets the fields, the object stays)
        if (hipMalloc(reinterpret_cast<void **>(&S->bad), 16) != hipSuccess ||
            hipMalloc(reinterpret_cast<void **>(&S->parent), (size_t)n_sides * 4 + 4) != hipSuccess ||
            hipMalloc(reinterpret_cast<void **>(&S->first), (size_t)n_sides * 4 + 4) != hipSuccess ||
            hipMalloc(reinterpret_cast<void **>(&S->multi), (size_t)N + 4) != hipSuccess) {
            ctx->err = "pf_side_components: out of device memory";
            return PF_ERR_HIP;
        }
        S->n_unitigs = N;
        reset = 1;
    }
    if (reset) k_cc_init<<<(n_sides + 255) / 256, 256, 0, st>>>(S->parent, S->first, S->multi, n_sides);
    S->labelled = false;
    PF_HIP(hipMemsetAsync(S->bad, 0, 4, st));
    // the records: the ones K-BFS left in its workspace, or the caller's
    const pf_bfs_record *d_rec = nullptr;
    const uint32_t *d_pool = nullptr;
    if (!records) {
        if (ctx->bfs_last_n != n_records || !ctx->bfs_last_rec) { ctx->err = "pf_side_components: no records of that length from the last K-BFS call"; return PF_ERR_ARG; }
        d_rec = ctx->bfs_last_rec;
        d_pool = ctx->bfs_last_pool;
        pool_len = ctx->bfs_last_pool_len;
    } else if (is_device(records)) {
        d_rec = records;
        d_pool = pool;
    } else {
        if (!grow(S->up_rec, S->up_rec_cap, n_records + n_extra + 1) || !grow(S->up_pool, S->up_pool_cap, pool_len + extra_pool_len + 1)) {
            ctx->err = "pf_side_components: out of device memory";
            return PF_ERR_HIP;
        }
        PF_HIP(hipMemcpyAsync(S->up_rec, records, n_records * sizeof(pf_bfs_record), hipMemcpyHostToDevice, st));
        PF_HIP(hipMemcpyAsync(S->up_pool, pool, pool_len * 4, hipMemcpyHostToDevice, st));
        d_rec = S->up_rec;
        d_pool = S->up_pool;
    }
    // a second call for the same K-BFS call (the caller adds the traversals it walked itself once they are done) adds only those
    const bool again = !records && !reset && S->added_call == ctx->bfs_call_id;
    if (!records) S->added_call = ctx->bfs_call_id;
    if (n_records && !again) {
        k_cc_edges<<<(unsigned)((n_records + 255) / 256), 256, 0, st>>>(d_rec, n_records, d_pool, pool_len, n_sides, S->parent, S->first, S->multi, S->bad);
        PF_HIP(hipGetLastError());
    }
    if (n_extra) {
        // traversals the caller walked itself: their records (list_off relative to extra_pool) contribute edges only
        pf_bfs_record *d_xrec = nullptr;
        uint32_t *d_xpool = nullptr;
        DevTmp<pf_bfs_record> xr;
        DevTmp<uint32_t> xp;
        PF_HIP(xr.alloc(n_extra * sizeof(pf_bfs_record)));
        PF_HIP(xp.alloc((extra_pool_len + 1) * 4));
        d_xrec = xr.p;
        d_xpool = xp.p;
        PF_HIP(hipMemcpyAsync(d_xrec, extra, n_extra * sizeof(pf_bfs_record), hipMemcpyDefault, st));
        if (extra_pool_len) PF_HIP(hipMemcpyAsync(d_xpool, extra_pool, extra_pool_len * 4, hipMemcpyDefault, st));
        k_cc_edges_long<<<dim3((unsigned)n_extra, 64), 256, 0, st>>>(d_xrec, n_extra, d_xpool, extra_pool_len, n_sides, S->parent, S->first, S->multi, S->bad);
        PF_HIP(hipGetLastError());
        PF_HIP(hipStreamSynchronize(st));   // (the temporaries are freed on return)
    }
    k_cc_multi<<<(N + 255) / 256, 256, 0, st>>>(S->multi, N, S->parent);
    PF_HIP(hipGetLastError());
    uint32_t bad = 0;
    PF_HIP(hipMemcpyAsync(&bad, S->bad, 4, hipMemcpyDeviceToHost, st));
    PF_HIP(hipStreamSynchronize(st));
    if (bad) { ctx->err = "pf_side_components: a record names a vertex outside the graph or a list outside the pool"; S->n_last = 0; return PF_ERR_ARG; }
    S->n_last = n_records;
    ctx->cc_rec = d_rec;
    return PF_OK;
}

int pf_replay_order(pf_ctx *ctx, uint32_t n_classes, uint32_t *order, uint32_t *class_off, uint32_t *labels) {
    if (!ctx || !ctx->cc || !order || !class_off || n_classes == 0 || n_classes > 1024) return PF_ERR_ARG;
    CcState *S = static_cast<CcState *>(ctx->cc);
    PF_HIP(hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    const uint64_t n = S->n_last;
    if (n >> 32) { ctx->err = "pf_replay_order: more than 2^32 records"; return PF_ERR_ARG; }
    uint64_t cap = S->cap;
    if (!S->labels || S->cap < n + 1) {
        for (uint32_t **p : {&S->labels, &S->cls, &S->idx, &S->cls2, &S->idx2})
            if (*p) { (void)hipFree(*p); *p = nullptr; }
        cap = n + n / 4 + 64;
        for (uint32_t **p : {&S->labels, &S->cls, &S->idx, &S->cls2, &S->idx2})
            if (hipMalloc(reinterpret_cast<void **>(p), cap * 4) != hipSuccess) { ctx->err = "pf_replay_order: out of device memory"; return PF_ERR_HIP; }
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

}  // extern "C"
