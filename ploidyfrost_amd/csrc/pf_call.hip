// The resident calling pipeline: CDBG::ploidyEstimation_ptr (reference src/CDBG.cpp:1101-1705) with everything between
// the commit replay and the file system on the MI355X.  The host uploads the MyUnitig state, runs the light sequential
// part of the driver loop on compact side records, and appends the text slabs it gets back to the ten result files;
// bubbles never exist on the host as objects.
//
//   K-SCAN    k_call_sides   one thread per unitig: for each open endpoint side what the driver loop (:1146-1222, 1347-1363)
//                            would do there -- exit by first successors, ownership (reference-string compare on the
//                            2-bit words), coverage gate, sortSeq_simple -- from static state only
//   K-PREP    k_call_prep    one thread per selected bubble: strict bubbles become K-BUBBLE tasks whose paths are oriented
//                            unitigs; branching ones are queued for K-PATHS; every task lands in the work queue of its
//                            LDS size class
//   K-PATHS   k_call_paths   one wavefront per branching bubble: the two-stack enumeration of all s->t walks (:1364-1412),
//                            path strings decoded lane-parallel from the 2-bit graph, sortSeq_branching as a rank sort
//   K-BUBBLE  (pf_bubble.hip) SeqAlign::SequenceAlignment for every task
//   K-SITES   k_call_sites   one wavefront per branching bubble: per-site k-length strings (:1448-1600), de-duplicated per
//                            allele group in std::set order, looked up in the count table (readCov(string), :29-60),
//                            group coverages accumulated in the reference's order
//   K-TEXT    k_call_format  one thread per bubble, twice: measure, (scan,) write -- the rows of alignseq.txt,
//                            allele_frequency.txt and the eight {bi,tri,tetra,penta}{cov,fre}.txt files, numbers printed
//                            as `ostream << double` does (pf_format_dev.hpp); bubble numbering (var_count) by a scan
//
// All integer / byte work with data-dependent control flow; HBM traffic is the text itself.  No MFMA.
#include <hip/hip_runtime.h>

#include <hipcub/hipcub.hpp>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstddef>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "pf_bubble_launch.hpp"
#include "pf_alnpack.hpp"
#include "pf_call_dev.hpp"
#include "pf_colored_dev.hpp"
#include "pf_cov_stream.hpp"
#include "pf_ctx.hpp"
#include "pf_device_common.hpp"
#include "pf_format_dev.hpp"
#include "pf_pair_dev.hpp"
#include "pf_stack_dev.hpp"
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

constexpr uint8_t B_PLUS = 0x01, B_MINUS = 0x02, B_STRICT_M = 0x08, B_STRICT_P = 0x10, B_COMPLEX_M = 0x20, B_COMPLEX_P = 0x40;
constexpr int N_STREAMS = PF_CALL_STREAMS;
constexpr int N_INT = N_STREAMS + 1;   // size / offset tables: the ten streams + the packed form of alignseq (pf_alnpack.hpp)
constexpr int S_PACK = N_STREAMS;
// first-pass pool sizes per bubble of a range (pf_call_align_lane, pf_call_reserve_lanes): bytes of aligned rows, sites, group bytes,
// indel lengths, bytes of path text
constexpr uint32_t FIRST_ROW_TEXT = 384, FIRST_SITES = 4, FIRST_GROUPS = 12, FIRST_ILEN = 2, FIRST_PATH_TEXT = 64;
// work lists of a batch: K-BUBBLE's queues (heavy and light per size class), then the three lists of the other kernels
constexpr int NQ = 2 * (kBubLdsClasses + 1);
constexpr int KEY_BRANCHING = NQ, KEY_SNP = NQ + 1, KEY_PAIR = NQ + 2, KEY_PAIR2 = NQ + 3, KEY_STACK = NQ + 4, KEY_TRIO = NQ + 5, KEY_TRIO4 = NQ + 6, KEY_NONE = NQ + 7;

struct DevBuf {
    void *p = nullptr;
    size_t cap = 0;
    template <typename T>
    T *as() const { return reinterpret_cast<T *>(p); }
    // contents are not preserved
    bool ensure(size_t bytes) {
        if (bytes <= cap && p) return true;
        if (p) (void)hipFree(p);
        p = nullptr;
        cap = 0;
        const size_t want = bytes + bytes / 4 + 256;
        if (hipMalloc(&p, want) != hipSuccess) { (void)hipGetLastError(); p = nullptr; return false; }
        cap = want;
        return true;
    }
    void release() {
        if (p) (void)hipFree(p);
        p = nullptr;
        cap = 0;
    }
};

// a capacity learnt by whichever lane met the larger batch
template <typename T, typename V>
inline void amax(std::atomic<T> &a, V v) {
    T cur = a.load(std::memory_order_relaxed);
    while (cur < (T)v && !a.compare_exchange_weak(cur, (T)v, std::memory_order_relaxed)) {}
}

// counters of one batch, device side (zeroed per batch)
struct CallCounters {
    unsigned int q_n[NQ];           // work queues: class c heavy = q_n[2c], light = q_n[2c + 1]
    unsigned int n_branching;
    unsigned int n_snp, n_snp_done;        // two-path bubbles of equal length: candidates for / takers of the single-SNP shortcut
    unsigned int n_pair, n_pair_done;      // two short paths: K-PAIR's list / the bubbles it finished
    unsigned int n_pair2, n_pair2_done;    // two paths of up to 128 bases: the list of K-PAIR's second tier
    unsigned int n_stack, n_stack_b, n_stack_done;   // K-STACK's lists (strict bubbles; branching ones, filled by K-PATHS) / the bubbles whose alignment it certified
    unsigned int n_trio, n_trio4, n_trio_done;   // three / four short paths: K-TRIO's lists / the bubbles it finished
    unsigned int paths_next, sites_next;   // queue heads of K-PATHS / K-SITES
    unsigned int n_many, max_rows;         // bubbles of more than 255 walks (K-PATHS' second launch takes them) / the most walks of any bubble
    unsigned int ks_need;                  // K-SITES: the longest site string a wavefront had no room for (err bit 4: the launch is repeated with room)
    unsigned int err;               // bit 0: > 65535 paths, 1: missing k-mer in a site string, 2: site string outside its row,
                                    // 3: a path pool overflowed (sizes below tell how much is needed), 4: site string too long
    unsigned int err_entrance, err_exit;   // the bubble bits 0 / 5 speak of (oriented vertices; whichever wavefront wrote last)
    unsigned long long path_head, text_head, sv_head, walk_head;
    unsigned long long max_need, retry_need;
    unsigned long long allele[4], core_cov, core_num, n_called, site_strings;
};

static_assert(offsetof(CallCounters, core_cov) == offsetof(CallCounters, allele) + 32 && offsetof(CallCounters, core_num) == offsetof(CallCounters, allele) + 40, "allele[4], core_cov, core_num are contiguous");

}  // namespace

namespace pf {

struct CallState {
    // T1 state
    DevBuf flags, plus, minus;
    bool have_state = false;
    // C1 results, one slot per unitig (two per unitig for a database without canonical counting)
    DevBuf cov_sum, cov_min, cov_miss;
    bool have_cov = false, per_strand = false;
    // colored path (pf_call_set_colours): cutoffs per colour, the colour sets the calling phase asks about -- per unitig the mask of
    // colours on every k-mer and UnitigColors::size(); for a colour on part of a unitig one bit per k-mer (reference orientation):
    // entries part_first[u] .. part_first[u + 1] = {colour, first word in part_bits} -- and K-COV-C's results, colour-major
    uint32_t n_colors = 0;
    DevBuf col_low, col_up, col_full, col_size, part_first, part_colour, part_word, part_bits;
    DevBuf ccov_sum, ccov_min, ccov_max, ccov_miss;
    std::atomic<uint32_t> sites_ks{0};   // K-SITES: room per site string once a launch asked for more than 2k + 64
    bool pack_alignseq = false;   // pf_call_set_alignseq_packed
    // scan
    DevBuf side_cnt, side_base, sides, ctask, scan_tmp, target, pending, killed, rstate, rflag, rsmall;
    uint64_t n_sides = 0;
    uint32_t low = 0, up = 0;
    // selection
    DevBuf kept;
    // super_bubble.txt rows
    DevBuf sb_cnt, sb_base, sb_sizes, sb_offs, sb_out;
    uint64_t sb_len = 0;
    uint64_t n_tasks = 0;
    // what pf_call_align leaves resident for pf_call_text_range: PF_CALL_LANES sets ("lanes"), so that the rows of one range of
    // bubbles can be formatted, fetched and written while the next ranges are aligned into the other sets
    struct AlignOut {
        DevBuf res, otext, osites, ogroups, oilen, sv_off, sv, vc;
        uint64_t t0 = 0;
        uint32_t nb = 0;
        pf_call_result cur = {};
        uint64_t used[5] = {};   // pool fill after the last pf_call_align_lane: row text, sites, group bytes, indel lengths, site values
        hipEvent_t read_ev[2] = {nullptr, nullptr};   // the last write passes of K-TEXT over this lane (one per text stream) have finished: the next alignment into it waits for them on its stream
    } lane[PF_CALL_LANES];
    // the working set of one pf_call_align_lane call, one per lane as well: lists, queues, path pools, per-wavefront scratch,
    // counters, streams -- calls on different lanes run side by side from different host threads (every kernel of a range ends in
    // a tail of a few slow bubbles: the next range's kernels fill the device meanwhile)
    struct AlignWork {
        DevBuf counters, btask, bpath, ptext, queues, blist, slist, plist, plist2, klist, klist_b, stack_scr, tlist, tlist4, trio_scr, trio_rows, trio_ok, pair_scr, pair_scr2, has, scan_tmp, scan_tmp2, paths_scr, sites_scr;
        DevBuf mlist, paths_big_scr;  // K-PATHS: bubbles of more than 255 walks, and the scratch of the launch that takes them
        uint32_t mlist_cap = 0;
        DevBuf walk_off, walk_pool;   // per batch: the oriented unitigs each branching bubble's walks visit (findUnitig of its site strings)
        uint64_t walk_cap = 0;
        hipStream_t stream = nullptr;   // lanes 1 ..: their own (lane 0 runs on the context's stream)
        bool own_stream = false;
        // K-PATHS runs beside K-SNP / K-PAIR (disjoint bubbles, shared atomic counters) on a stream of its own
        hipStream_t side_stream = nullptr;
        hipEvent_t ev_prep = nullptr, ev_paths = nullptr;
    } work[PF_CALL_LANES];
    // K-TEXT's own scratch, counters and streams: it may run from another host thread beside pf_call_align (other lanes).  Two sets,
    // taken in turn by the parity of the slab: a piece's count pass ends in a wait of the host (the slab is laid out from the
    // totals), its write pass does not -- the count pass of the next piece, on the other stream, runs beside it
    struct TextWork {
        DevBuf sizes, offs, totals, tcounters, tscan;
        hipStream_t stream = nullptr;
    } text[2];
    // capacities learnt from earlier batches (any lane)
    std::atomic<uint64_t> path_pool{0}, text_pool{0}, sv_pool{0};
    std::atomic<uint64_t> otext_cap{0}, osites_cap{0}, ogroups_cap{0}, oilen_cap{0};
    // output slabs: two sets, so that one can be fetched while the next batch is formatted
    // (the ten streams of a slab lie one after the other in one buffer, as the host wants them: one copy fetches a slab)
    DevBuf out[PF_CALL_SLABS];
    uint64_t out_len[PF_CALL_SLABS][N_STREAMS] = {}, out_off[PF_CALL_SLABS][N_STREAMS] = {};
    hipStream_t copy_stream = nullptr;
    hipEvent_t fetch_ev[2] = {nullptr, nullptr};   // pf_call_fetch_range / pf_call_fetch_wait
    hipEvent_t text_ev[PF_CALL_SLABS] = {};        // the write pass of the piece in a slab has finished (the fetches wait for it on their stream)
    bool mt_format = false;   // pf_call_set_format
    void release_all() {
        DevBuf *all[] = {&col_low, &col_up, &col_full, &col_size, &part_first, &part_colour, &part_word, &part_bits, &ccov_sum, &ccov_min, &ccov_max, &ccov_miss,
                         &flags, &plus, &minus, &cov_sum, &cov_min, &cov_miss, &side_cnt, &side_base, &sides, &ctask, &scan_tmp, &target, &pending, &killed, &rstate, &rflag, &rsmall, &kept, &sb_cnt, &sb_base, &sb_sizes, &sb_offs, &sb_out,
                         };
        for (DevBuf *b : all) b->release();
        for (TextWork &t : text) {
            for (DevBuf *b : {&t.sizes, &t.offs, &t.totals, &t.tcounters, &t.tscan}) b->release();
            if (t.stream) { (void)hipStreamDestroy(t.stream); t.stream = nullptr; }
        }
        for (AlignOut &o : lane) {
            for (DevBuf *b : {&o.res, &o.otext, &o.osites, &o.ogroups, &o.oilen, &o.sv_off, &o.sv, &o.vc}) b->release();
            for (hipEvent_t &e : o.read_ev) if (e) { (void)hipEventDestroy(e); e = nullptr; }
        }
        for (AlignWork &w : work) {
            for (DevBuf *b : {&w.counters, &w.btask, &w.bpath, &w.ptext, &w.queues, &w.blist, &w.slist, &w.plist, &w.plist2, &w.klist, &w.klist_b, &w.stack_scr, &w.tlist, &w.tlist4, &w.trio_scr, &w.trio_rows,
                              &w.trio_ok, &w.pair_scr, &w.pair_scr2, &w.has, &w.scan_tmp, &w.scan_tmp2, &w.paths_scr, &w.sites_scr, &w.mlist, &w.paths_big_scr, &w.walk_off, &w.walk_pool})
                b->release();
            if (w.own_stream && w.stream) (void)hipStreamDestroy(w.stream);
            w.stream = nullptr; w.own_stream = false;
            if (w.side_stream) { (void)hipStreamDestroy(w.side_stream); w.side_stream = nullptr; }
            if (w.ev_prep) { (void)hipEventDestroy(w.ev_prep); w.ev_prep = nullptr; }
            if (w.ev_paths) { (void)hipEventDestroy(w.ev_paths); w.ev_paths = nullptr; }
        }
        for (DevBuf &b : out) b.release();
        if (copy_stream) { (void)hipStreamDestroy(copy_stream); copy_stream = nullptr; }
        for (hipEvent_t &e : fetch_ev)
            if (e) { (void)hipEventDestroy(e); e = nullptr; }
        for (hipEvent_t &e : text_ev)
            if (e) { (void)hipEventDestroy(e); e = nullptr; }
    }
};

void call_destroy(pf_ctx *ctx) {
    if (!ctx->call) return;
    ctx->call->release_all();
    delete ctx->call;
    ctx->call = nullptr;
}
void call_invalidate(pf_ctx *ctx) {  // a new graph or count table: resident scan results no longer apply
    if (!ctx->call) return;
    ctx->call->have_cov = false;
    ctx->call->have_state = false;
    ctx->call->n_sides = ctx->call->n_tasks = 0;
}

}  // namespace pf

#define NEED_TEXT(buf, bytes) do { if (!(buf).ensure(bytes)) { pf::CtxErr{ctx} = "pf_call_text: out of device memory"; return PF_ERR_HIP; } } while (0)

namespace {

CallState *state_of(pf_ctx *ctx) {
    if (!ctx->call) {
        ctx->call = new CallState();
        // copies of finished text slabs run beside the next batch's kernels
        if (hipStreamCreateWithFlags(&ctx->call->copy_stream, hipStreamNonBlocking) != hipSuccess) { (void)hipGetLastError(); ctx->call->copy_stream = nullptr; }
    }
    return ctx->call;
}

// ---------------------------------------------------------------------------------------------------------------------
// K-SCAN
struct ScanArgs {
    const uint8_t *flags;
    const uint32_t *plus, *minus, *succ, *pred;
    const uint64_t *seq, *off;
    const uint32_t *len;
    uint32_t N;
    int k;
    const uint64_t *cov_sum;
    const uint32_t *cov_min;
    const uint8_t *cov_miss;
    int per_strand;
    uint32_t low, up;
    const uint32_t *side_base;  // exclusive scan of the per-unitig side counts
    pf_call_side *sides;
    CallTask *tasks;
    uint32_t *target;           // per side: the record of the side the bubble's exit faces (closed when this side is handled), or NONE
    // colored (CCDBG): K-COV-C's results colour-major ([c * N + u]), one (low, up) per colour, the colour sets
    uint32_t n_colors;
    const uint64_t *ccov_sum;
    const uint32_t *ccov_min, *ccov_max;
    const uint8_t *ccov_miss;
    const uint32_t *clow, *cup;
    const uint64_t *full, *size_total;
};

// readCovUni(u, low, up, c) of src/CCDBG.cpp:123-156 from K-COV-C's resident results: (sum / len, true) iff every k-mer is in colour
// c's database with low < count < up, else (0, false)
struct ColourCov {
    const uint64_t *sum;
    const uint32_t *mn, *mx;
    const uint8_t *miss;
    const uint32_t *low, *up;
    uint32_t N;
    __device__ inline bool ok(uint32_t c, uint32_t u) const {
        const size_t o = (size_t)c * N + u;
        return !miss[o] && mn[o] > low[c] && mx[o] < up[c];
    }
    __device__ inline double mean(uint32_t c, uint32_t u, uint32_t len_km) const { return (double)sum[(size_t)c * N + u] / (double)len_km; }
};

// colored sortSeq_simple (src/CCDBG.cpp:368-480) with its exact partition scheme: descending number of colours, then descending
// length, then descending reference string.  n <= 4.
__device__ inline void sort_inner_colored_dev(const uint64_t *__restrict__ seq, const uint64_t *__restrict__ off, const uint32_t *__restrict__ len,
                                              uint32_t *pc, uint32_t *ov, int n) {
    int stack_lo[8], stack_hi[8];
    int sp = 1;
    stack_lo[0] = 0;
    stack_hi[0] = n - 1;
    auto ref_cmp = [&](int x, int y) -> int {   // length first, then the strings (equal lengths: strcmp)
        const uint32_t lx = len[ov[x] >> 1], ly = len[ov[y] >> 1];
        if (lx != ly) return lx > ly ? 1 : -1;
        return unitig_cmp(seq, off, len, ov[x] >> 1, ov[y] >> 1);
    };
    while (sp > 0) {
        --sp;
        const int low = stack_lo[sp], high = stack_hi[sp];
        if (high <= low) continue;
        int i = low, j = high;
        for (;;) {
            while (pc[i] >= pc[low]) {
                if (pc[i] > pc[low] || ref_cmp(i, low) > 0) i++;
                else break;
                if (i == high) break;
            }
            while (pc[j] <= pc[low]) {
                if (pc[j] < pc[low] || ref_cmp(j, low) < 0) j--;
                else break;
                if (j == low) break;
            }
            if (i >= j) break;
            const uint32_t tp = pc[i]; pc[i] = pc[j]; pc[j] = tp;
            const uint32_t to = ov[i]; ov[i] = ov[j]; ov[j] = to;
        }
        {
            const uint32_t tp = pc[low]; pc[low] = pc[j]; pc[j] = tp;
            const uint32_t to = ov[low]; ov[low] = ov[j]; ov[j] = to;
        }
        stack_lo[sp] = low; stack_hi[sp] = j - 1; ++sp;
        stack_lo[sp] = j + 1; stack_hi[sp] = high; ++sp;
    }
}

__global__ void k_call_count_sides(const uint8_t *__restrict__ flags, uint32_t N, uint32_t *__restrict__ cnt) {
    const uint32_t u = blockIdx.x * blockDim.x + threadIdx.x;
    if (u <= N) cnt[u] = u < N ? (uint32_t)__popc(flags[u] & 3u) : 0u;
}

__device__ inline uint32_t first_succ(const uint32_t *__restrict__ succ, uint32_t ov) {
    const uint4 r = *reinterpret_cast<const uint4 *>(succ + (size_t)ov * 4);
    if (r.x != NONE) return r.x;
    if (r.y != NONE) return r.y;
    if (r.z != NONE) return r.z;
    return r.w;
}

template <bool COLORED>
__global__ __launch_bounds__(256) void k_call_sides(ScanArgs a) {
    const uint32_t u = blockIdx.x * blockDim.x + threadIdx.x;
    if (u >= a.N) return;
    const uint8_t f = a.flags[u];
    if ((f & 3) == 0) return;
    uint32_t slot = a.side_base[u];
    const uint32_t N = a.N;
    auto cslot = [&](uint32_t ov) -> size_t { return (a.per_strand && (ov & 1)) ? (size_t)N + (ov >> 1) : (size_t)(ov >> 1); };
    auto len_km = [&](uint32_t x) { return a.len[x] - (uint32_t)a.k + 1; };
    auto mean_ov = [&](uint32_t ov) { return (double)a.cov_sum[cslot(ov)] / (double)len_km(ov >> 1); };
    for (int side = 0; side < 2; ++side) {
        const bool ps = side == 0;
        if (!(f & (ps ? B_PLUS : B_MINUS))) continue;
        pf_call_side r;
        r.u = u;
        r.exit_ov = NONE;
        r.err_unitig = 0;
        r.plus_side = ps;
        r.kind = 0;
        r.aligned = 0;
        r.err = 0;
        CallTask t;
        t.u = u;
        t.entrance_ov = t.exit_ov = 0;
        t.strict = t.n_inner = t.n_cov = t.pad_ = 0;
        for (int q = 0; q < 4; ++q) { t.inner[q] = 0; t.cov[q] = 0; }
        t.core_mean = t.cov_sum = 0;
        const uint32_t my = slot++;
        do {
            if (f & (ps ? B_COMPLEX_P : B_COMPLEX_M)) { r.kind = 1; break; }
            const uint32_t uo = 2 * u + (ps ? 0 : 1);
            const bool strict = (f & (ps ? B_STRICT_P : B_STRICT_M)) != 0;
            if (!COLORED && a.cov_miss[cslot(uo)]) { r.err = 1; r.err_unitig = u; break; }  // core = readCov(u), u oriented
            uint32_t exit_ov;
            if (strict) {
                exit_ov = first_succ(a.succ, uo);
                if (exit_ov != NONE) exit_ov = first_succ(a.succ, exit_ov);
            } else {
                const uint32_t want = ps ? a.plus[u] : a.minus[u];
                exit_ov = first_succ(a.succ, uo);
                // (bounded: a walk longer than the graph means the partner is not on the first-successor chain)
                for (uint32_t steps = 0; exit_ov != NONE && (exit_ov >> 1) + 1 != want; ++steps) {
                    if (steps > N) { exit_ov = NONE; break; }
                    exit_ov = first_succ(a.succ, exit_ov);
                }
            }
            if (exit_ov == NONE) { r.err = 2; break; }
            const uint32_t eu = exit_ov >> 1;
            r.exit_ov = exit_ov;
            t.entrance_ov = uo;
            t.exit_ov = exit_ov;
            t.strict = strict;
            if (unitig_cmp(a.seq, a.off, a.len, u, eu) < 0) { r.kind = 2; break; }  // the other endpoint owns this bubble
            r.kind = 3;
            if (COLORED) {
                // src/CCDBG.cpp:2838-2853: the per-colour means are summed until a colour fails its range test (the `flag == false;`
                // there is a no-op, so the bubble is processed regardless)
                const ColourCov cc{a.ccov_sum, a.ccov_min, a.ccov_max, a.ccov_miss, a.clow, a.cup, N};
                const uint32_t C = a.n_colors;
                double core = 0;
                for (uint32_t c = 0; c < C; ++c) {
                    if (!cc.ok(c, u)) break;
                    core += cc.mean(c, u, len_km(u));
                }
                t.core_mean = core;
                bool flag = true;
                if (strict) {   // :2867-2931: the [colour][path] matrix of mean coverages, its gates, the colored sortSeq_simple
                    uint32_t pc[4] = {0, 0, 0, 0};
                    uint32_t path = 0;
                    const uint32_t *row = a.succ + (size_t)uo * 4;
                    for (int b = 0; b < 4 && flag; ++b) {
                        const uint32_t w = row[b];
                        if (w == NONE) continue;
                        const uint32_t wu = w >> 1;
                        t.inner[t.n_inner++] = w;
                        const uint64_t fm = a.full[wu];
                        uint32_t jn = 0;
                        for (uint32_t c = 0; c < C; ++c) {
                            if (!((fm >> c) & 1)) continue;
                            ++jn;
                            if (!cc.ok(c, wu)) { flag = false; break; }
                        }
                        if (!flag) break;
                        if (a.size_total[wu] != (uint64_t)jn * len_km(wu)) { flag = false; break; }  // a colour on part of it
                        pc[path++] = jn;
                    }
                    if (flag) {   // some colour must see more than one of the paths (an entry of the matrix is its mean, 0 if absent)
                        flag = false;
                        for (uint32_t c = 0; c < C && !flag; ++c) {
                            int nz = 0;
                            for (uint32_t q = 0; q < path; ++q) {
                                const uint32_t wu = t.inner[q] >> 1;
                                nz += ((a.full[wu] >> c) & 1) && cc.mean(c, wu, len_km(wu)) != 0.0;
                            }
                            flag = nz > 1;
                        }
                    }
                    if (flag) {
                        sort_inner_colored_dev(a.seq, a.off, a.len, pc, t.inner, (int)path);
                        t.n_cov = (uint8_t)path;
                    }
                }
                r.aligned = flag;
                break;
            }
            t.core_mean = mean_ov(uo);
            bool aligned = true;
            if (strict) {
                const uint32_t *row = a.succ + (size_t)uo * 4;
                for (int b = 0; b < 4 && aligned && !r.err; ++b) {
                    const uint32_t w = row[b];
                    if (w == NONE) continue;
                    t.inner[t.n_inner++] = w;
                    if (a.cov_miss[cslot(w)]) { r.err = 1; r.err_unitig = w >> 1; break; }
                    const uint32_t mn = a.cov_min[cslot(w)];
                    if (mn > a.low && mn < a.up) {
                        const double mcov = mean_ov(w);
                        t.cov[t.n_cov++] = mcov;
                        t.cov_sum += mcov;
                    } else {
                        aligned = false;
                    }
                }
                if (aligned && !r.err) {
                    // the reference also reads the predecessors' coverage and drops it (src/CDBG.cpp:1224-1239)
                    const uint32_t *prow = a.pred + (size_t)uo * 4;
                    for (int b = 0; b < 4; ++b) {
                        const uint32_t w = prow[b];
                        if (w != NONE && a.cov_miss[cslot(w)]) { r.err = 1; r.err_unitig = w >> 1; break; }
                    }
                    if (!r.err) sort_inner_dev(a.seq, a.off, a.len, t.cov, t.inner, (int)t.n_cov);
                }
            }
            r.aligned = aligned;
        } while (false);
        a.sides[my] = r;
        a.tasks[my] = t;
        // handling this side as the owner clears the facing side of the exit (src/CDBG.cpp:1656-1679): which record is that?
        uint32_t tg = NONE;
        if (r.kind == 3) {
            const uint32_t eu = r.exit_ov >> 1;
            const uint8_t ef = a.flags[eu];
            const bool facing_minus = (r.exit_ov & 1) == 0;   // '+' exit: its minus side faces the bubble
            if (ef & (facing_minus ? B_MINUS : B_PLUS)) tg = a.side_base[eu] + ((facing_minus && (ef & B_PLUS)) ? 1u : 0u);
        }
        a.target[my] = tg;
    }
}

// ---- part B of the driver loop, exactly, without walking the sides one after the other ------------------------------------
// Sequentially (src/CDBG.cpp:1146-1186, 1656-1679): a side is handled only if its bit is still set when its unitig comes up, and
// an owner that is handled clears the side its exit faces.  So side j is alive iff no owner i < j with target(i) = j is alive --
// a recursion over strictly smaller indices.  Rounds of a monotone propagation settle it: `pending[j]` counts the potential
// killers of j not yet known to be dead; a side with no pending killer is alive and kills its target, a killed side releases
// its own target.  Symmetric bubbles settle in two rounds; chains through asymmetric state take one round per link.
struct ResolveArgs {
    const pf_call_side *sides;
    const uint32_t *target;
    uint32_t n;
    int *pending;        // potential killers not yet dead
    uint8_t *killed;     // some killer is alive
    uint8_t *state;      // 0 undecided, 1 alive, 2 dead
    uint32_t *flag;      // 1: called (alive owner that passes the gate)
    unsigned int *undecided;
    unsigned int *first_err;   // smallest index of an alive side with err != 0
};

__global__ void k_call_pending(ResolveArgs a) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= a.n) return;
    const uint32_t t = a.target[i];
    if (a.sides[i].kind == 3 && t != NONE && t > i) atomicAdd(&a.pending[t], 1);
}

__global__ void k_call_resolve(ResolveArgs a) {
    const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= a.n || a.state[j]) return;
    // (plain loads of values other threads update with atomics in this very launch: a stale value only postpones the decision)
    const bool dead = __hip_atomic_load(&a.killed[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0;
    const int pend = __hip_atomic_load(&a.pending[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (!dead && pend > 0) { atomicAdd(a.undecided, 1u); return; }
    const pf_call_side r = a.sides[j];
    const uint32_t t = a.target[j];
    const bool kills = r.kind == 3 && t != NONE && t > j;
    if (dead) {
        a.state[j] = 2;
        a.flag[j] = 0;
        if (kills) atomicSub(&a.pending[t], 1);
    } else {
        a.state[j] = 1;
        if (r.kind != 1 && r.err) atomicMin(a.first_err, j);
        a.flag[j] = (r.kind == 3 && !r.err && r.aligned) ? 1u : 0u;
        if (kills) __hip_atomic_store(&a.killed[t], (uint8_t)1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// appends `val` to one of the lists chosen by key (0 .. NQ-1 K-BUBBLE's queues, KEY_BRANCHING, KEY_SNP, KEY_PAIR; KEY_NONE =
// nowhere): one atomic per key and wave
struct CallLists {
    uint32_t *queues;   // NQ lists of nb entries
    uint32_t *blist, *slist, *plist, *plist2, *klist, *klist_b, *tlist, *tlist4;
    uint32_t nb;
};
__device__ inline void wave_append(int key, uint32_t val, const CallLists &L, CallCounters *cnt) {
    unsigned long long todo = __ballot(key != KEY_NONE);
    while (todo) {
        // the key of the first lane still waiting, and every lane with the same key
        const int x = __shfl(key, __ffsll((long long)todo) - 1, WAVE);
        const unsigned long long m = __ballot(key == x);
        todo &= ~m;
        const int leader = __ffsll((long long)m) - 1;
        uint32_t base = 0;
        // (the counter's place by one integer select after the other; see paths_flush for why not a nested choice of pointers)
        uint32_t c_off = (uint32_t)offsetof(CallCounters, n_trio4);
        c_off = x == KEY_TRIO ? (uint32_t)offsetof(CallCounters, n_trio) : c_off;
        c_off = x == KEY_STACK ? (uint32_t)offsetof(CallCounters, n_stack) : c_off;
        c_off = x == KEY_PAIR2 ? (uint32_t)offsetof(CallCounters, n_pair2) : c_off;
        c_off = x == KEY_PAIR ? (uint32_t)offsetof(CallCounters, n_pair) : c_off;
        c_off = x == KEY_SNP ? (uint32_t)offsetof(CallCounters, n_snp) : c_off;
        c_off = x == KEY_BRANCHING ? (uint32_t)offsetof(CallCounters, n_branching) : c_off;
        c_off = x < NQ ? (uint32_t)offsetof(CallCounters, q_n) + 4u * (uint32_t)x : c_off;
        if (lane_id() == leader)
            base = atomicAdd(reinterpret_cast<unsigned int *>(reinterpret_cast<char *>(cnt) + c_off), (unsigned int)__popcll(m));
        base = __shfl(base, leader, WAVE);
        if (key == x) {
            const uint32_t at = base + (uint32_t)__popcll(m & ((1ull << lane_id()) - 1));
            if (x < NQ) L.queues[(size_t)x * L.nb + at] = val;
            else if (x == KEY_BRANCHING) L.blist[at] = val;
            else if (x == KEY_SNP) L.slist[at] = val;
            else if (x == KEY_PAIR) L.plist[at] = val;
            else if (x == KEY_PAIR2) L.plist2[at] = val;
            else if (x == KEY_STACK) L.klist[at] = val;
            else if (x == KEY_TRIO) L.tlist[at] = val;
            else L.tlist4[at] = val;
        }
    }
}

// The same for a block of four wavefronts (every thread of the block must call it): the counters of all lists share a cache line or two,
// and atomics on one line queue one behind the other -- a wavefront's three or four were most of K-PREP's launch.  The wavefronts leave
// their counts per key in LDS, one thread per key adds the block's total, every lane takes its place behind the wavefronts before its own.
constexpr int N_KEYS = NQ + 7;
__device__ inline void block_append(int key, uint32_t val, const CallLists &L, CallCounters *cnt) {
    __shared__ uint32_t s_n[4][N_KEYS];
    __shared__ uint32_t s_base[N_KEYS];
    const int lane = lane_id(), wv = (int)(threadIdx.x >> 6);
    if (lane < N_KEYS) s_n[wv][lane] = 0;
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
    __builtin_amdgcn_wave_barrier();
    uint32_t rank = 0;
    unsigned long long todo = __ballot(key != KEY_NONE);
    while (todo) {
        const int x = __shfl(key, __ffsll((long long)todo) - 1, WAVE);
        const unsigned long long m = __ballot(key == x);
        todo &= ~m;
        if (key == x) rank = (uint32_t)__popcll(m & ((1ull << lane) - 1));
        if (lane == __ffsll((long long)m) - 1) s_n[wv][x] = (uint32_t)__popcll(m);
    }
    __syncthreads();
    if (threadIdx.x < (unsigned)N_KEYS) {
        const int x = (int)threadIdx.x;
        const uint32_t total = s_n[0][x] + s_n[1][x] + s_n[2][x] + s_n[3][x];
        uint32_t base = 0;
        if (total) {
            uint32_t c_off = (uint32_t)offsetof(CallCounters, n_trio4);
            c_off = x == KEY_TRIO ? (uint32_t)offsetof(CallCounters, n_trio) : c_off;
            c_off = x == KEY_STACK ? (uint32_t)offsetof(CallCounters, n_stack) : c_off;
            c_off = x == KEY_PAIR2 ? (uint32_t)offsetof(CallCounters, n_pair2) : c_off;
            c_off = x == KEY_PAIR ? (uint32_t)offsetof(CallCounters, n_pair) : c_off;
            c_off = x == KEY_SNP ? (uint32_t)offsetof(CallCounters, n_snp) : c_off;
            c_off = x == KEY_BRANCHING ? (uint32_t)offsetof(CallCounters, n_branching) : c_off;
            c_off = x < NQ ? (uint32_t)offsetof(CallCounters, q_n) + 4u * (uint32_t)x : c_off;
            base = atomicAdd(reinterpret_cast<unsigned int *>(reinterpret_cast<char *>(cnt) + c_off), total);
        }
        s_base[x] = base;
    }
    __syncthreads();
    if (key != KEY_NONE) {
        uint32_t at = s_base[key] + rank;
        for (int w = 0; w < wv; ++w) at += s_n[w][key];
        if (key < NQ) L.queues[(size_t)key * L.nb + at] = val;
        else if (key == KEY_BRANCHING) L.blist[at] = val;
        else if (key == KEY_SNP) L.slist[at] = val;
        else if (key == KEY_PAIR) L.plist[at] = val;
        else if (key == KEY_PAIR2) L.plist2[at] = val;
        else if (key == KEY_STACK) L.klist[at] = val;
        else if (key == KEY_TRIO) L.tlist[at] = val;
        else L.tlist4[at] = val;
    }
    __syncthreads();   // (the tables may be used again by the caller's next call)
}

// ---------------------------------------------------------------------------------------------------------------------
// K-PREP
struct PrepArgs {
    const CallTask *ct;
    const uint32_t *kept;
    uint64_t t0;
    uint32_t nb;
    const uint32_t *len;
    pf_bubble_task *btask;
    pf_bubble_path *bpath;
    pf_bubble_result *res;
    CallLists lists;   // work queues; branching bubbles; single-SNP candidates; two short paths (batch-local indices)
    int snp_ok;        // the scores allow the single-SNP shortcut
    int pair_ok;       // K-PAIR runs
    int stack_ok;      // K-STACK runs
    int trio_ok;       // K-TRIO runs
    CallCounters *cnt;
};

__global__ __launch_bounds__(256) void k_call_prep(PrepArgs a) {
    const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    int key = KEY_NONE;
    unsigned long long need3 = 0, retry = 0;
    if (j < a.nb) {
        const CallTask &t = a.ct[a.kept[a.t0 + j]];
        pf_bubble_result z;
        z.rows_off = z.site_off = z.group_off = z.ilen_off = 0;
        z.n_rows = z.n_cols = z.n_sites = z.n_indel_len = 0;
        a.res[j] = z;
        if (t.strict) {
            uint32_t l0 = 0, lmax = 0, lmin = 0xFFFFFFFFu, sum = 0;
            for (int q = 0; q < t.n_inner; ++q) {
                const uint32_t L = a.len[t.inner[q] >> 1];
                a.bpath[(size_t)4 * j + q] = pf_bubble_path{0, L, t.inner[q]};
                if (q == 0) l0 = L;
                lmax = L > lmax ? L : lmax;
                lmin = L < lmin ? L : lmin;
                sum += L;
            }
            a.btask[j] = pf_bubble_task{(uint64_t)4 * j, t.n_inner, 0};
            if (t.n_inner >= 2) {  // fewer than two paths: the reference indexes str[1] blindly; skipped
                const int c = bubble_class(l0, lmax);
                key = 2 * c + ((t.n_inner > 2 || lmax > 64) ? 0 : 1);
                if (c == kBubLdsClasses) need3 = bubble_need(l0, lmax);
                retry = job_bytes(sum < 60000u ? sum : 60000u, lmax);
                // two paths of one length: K-SNP looks at them first (thread per bubble) and hands on what is not a single SNP;
                // two short paths of any kind: K-PAIR (thread per bubble)
                // K-STACK first (thread per bubble, a certificate instead of the dynamic programming) for whatever it can hold
                if (t.n_inner > 2 && a.stack_ok && lmax <= STACK_MAX && (a.stack_ok >= 2 || sum == t.n_inner * l0)) key = KEY_STACK;
                else if (t.n_inner > 2 && a.trio_ok && lmax <= TRIO_MAX && lmax - lmin <= (uint32_t)PairGeom<TRIO_MAX>::MAX_SKEW) key = t.n_inner == 3 ? KEY_TRIO : KEY_TRIO4;
                if (t.n_inner == 2) {
                    const uint32_t l1 = sum - l0;
                    if (a.snp_ok && sum == 2 * l0) key = KEY_SNP;
                    else if (a.stack_ok >= 3 && lmax <= STACK_MAX) key = KEY_STACK;
                    else if (a.pair_ok && pair_fits<PAIR_MAX>(l0, l1)) key = KEY_PAIR;
                    else if (a.pair_ok && pair_fits<PAIR_MAX2>(l0, l1)) key = KEY_PAIR2;
                }
            }
        } else {
            a.btask[j] = pf_bubble_task{0, 0, 0};
            key = KEY_BRANCHING;
        }
    }
    block_append(key, j, a.lists, a.cnt);
    // class 3 / retry sizing: rare, one atomic per wave that has any
    unsigned long long m3 = need3, mr = retry;
    for (int o = 32; o > 0; o >>= 1) {
        const unsigned long long x3 = ((unsigned long long)__shfl_down((uint32_t)(m3 >> 32), o, WAVE) << 32) | __shfl_down((uint32_t)m3, o, WAVE);
        const unsigned long long xr = ((unsigned long long)__shfl_down((uint32_t)(mr >> 32), o, WAVE) << 32) | __shfl_down((uint32_t)mr, o, WAVE);
        m3 = x3 > m3 ? x3 : m3;
        mr = xr > mr ? xr : mr;
    }
    if (lane_id() == 0) {
        if (m3) atomicMax(&a.cnt->max_need, m3);
        if (mr) atomicMax(&a.cnt->retry_need, mr);
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// K-SNP: the bi-allelic SNP bubble -- two equally long inner unitigs that differ in one base, most of all bubbles -- needs no
// dynamic programming (the single-SNP shortcut of K-BUBBLE, proof in pf_bubble.hip) and no wavefront either: its cost is a
// chain of dependent loads (task -> unitig offsets -> 2-bit words), so one THREAD per bubble keeps 64 of them in flight per
// wavefront instead of one.  The two rows, the SNP column and the groups {1, 2} go to the same pools K-BUBBLE publishes to
// (one atomic per wavefront and pool); anything that is not exactly one mismatch goes to K-BUBBLE's queue of its size class.
struct SnpArgs {
    const CallTask *ct;
    const uint32_t *kept;
    uint64_t t0;
    uint32_t nb;
    const uint32_t *slist;
    const uint64_t *seq, *off;
    const uint32_t *len;
    pf_bubble_result *res;
    char *otext;
    uint64_t text_cap;
    pf_bubble_site *osites;
    uint64_t site_cap;
    uint8_t *ogroups;
    uint64_t group_cap;
    unsigned long long *heads;  // K-BUBBLE's pool heads: [0] text, [1] sites, [2] groups
    CallLists lists;
    int pair_ok, stack_ok;
    CallCounters *cnt;
};

constexpr uint32_t SNP_STAGE = 8192;   // bytes of LDS per wavefront for its rows (64 bubbles of two 64-base paths)

__global__ __launch_bounds__(256) void k_call_snp(SnpArgs a) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    const int lane = lane_id();
    const uint32_t n = a.cnt->n_snp;
    const bool active = i < n;
    uint32_t j = 0, m = 0, ov0 = 0, ov1 = 0, col = 0, diff = 0;
    const uint64_t *w0 = nullptr, *w1 = nullptr;
    if (active) {
        j = a.slist[i];
        const CallTask &t = a.ct[a.kept[a.t0 + j]];
        ov0 = t.inner[0];
        ov1 = t.inner[1];
        m = a.len[ov0 >> 1];
        w0 = a.seq + a.off[ov0 >> 1];
        w1 = a.seq + a.off[ov1 >> 1];
        // 32 bases per step from two packed words (the paths are equally long: K-PREP's condition for this list)
        for (uint32_t c = 0; 32 * c < m; ++c) {
            const uint64_t x = oriented_chunk(w0, m, (ov0 & 1) != 0, c) ^ oriented_chunk(w1, m, (ov1 & 1) != 0, c);
            const uint64_t d = (x | (x >> 1)) & 0x5555555555555555ull;   // one bit per differing base
            if (d) {
                diff += (uint32_t)__popcll(d);
                col = 32 * c + (uint32_t)(__clzll((long long)d) >> 1);
            }
        }
    }
    const bool take = active && diff == 1;
    // pool space: one set of atomics per BLOCK.  The pool heads share a cache line and every taker of the batch adds to them: atomics on
    // one line queue one behind the other (~11 ns each), and a wavefront's four were what this launch took -- 40 000 of them, 0.45 of its
    // 0.6 ms (SQ: 81 % of the wave-cycles waiting, 1.5 % issuing VALU).  The four wavefronts of a block pool their totals through LDS.
    const unsigned long long tm = __ballot(take);
    uint32_t my_excl = 0, wave_total = 0;
    {
        // exclusive prefix of 2 m over the taking lanes
        uint32_t mine = take ? 2 * m : 0, incl = mine;
        for (int o = 1; o < WAVE; o <<= 1) {
            const uint32_t x = __shfl_up(incl, o, WAVE);
            if (lane >= o) incl += x;
        }
        my_excl = incl - mine;
        wave_total = __shfl(incl, WAVE - 1, WAVE);
    }
    __shared__ uint32_t s_wtot[4], s_wcnt[4];
    __shared__ unsigned long long s_tb, s_sb;
    const int wv = (int)(threadIdx.x >> 6);
    if (lane == 0) {
        s_wtot[wv] = wave_total;
        s_wcnt[wv] = (uint32_t)__popcll(tm);
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        const uint32_t total = s_wtot[0] + s_wtot[1] + s_wtot[2] + s_wtot[3], cnt = s_wcnt[0] + s_wcnt[1] + s_wcnt[2] + s_wcnt[3];
        unsigned long long tb = 0, sb = 0;
        if (cnt) {
            tb = atomicAdd(&a.heads[0], (unsigned long long)total);
            sb = atomicAdd(&a.heads[1], (unsigned long long)cnt);
            atomicAdd(&a.heads[2], 2ull * cnt);   // groups: two bytes per site, at 2 * (site index)
            atomicAdd(&a.cnt->n_snp_done, cnt);
        }
        s_tb = tb;
        s_sb = sb;
    }
    __syncthreads();
    uint32_t before_t = 0, before_c = 0;
    for (int w = 0; w < wv; ++w) { before_t += s_wtot[w]; before_c += s_wcnt[w]; }
    const unsigned long long wb = s_tb + before_t;   // where this wavefront's rows start
    const unsigned long long t_off = wb + my_excl;
    const unsigned long long s_off = s_sb + before_c + (unsigned long long)__popcll(tm & ((1ull << lane) - 1));
    // The rows leave through LDS: a wavefront's rows are one contiguous span of the text pool (lane order), so they are staged
    // per lane and copied out by consecutive lanes -- whole 64-byte segments per store instead of 64 scattered single bytes
    // (which cost 39 bytes of HBM write traffic per byte written, by the PMC counters).
    __shared__ __attribute__((aligned(16))) char s_rows[4][SNP_STAGE];
    char *stage = s_rows[wv];
    const bool staged = wave_total <= SNP_STAGE;
    if (take) {
        // K-SNP is the first taker of a batch (heads zeroed before it, K-BUBBLE launched behind it on the stream): the group
        // head moves two bytes for every site it takes, so its group offset is twice its site offset
        pf_bubble_result r;
        r.rows_off = t_off;
        r.site_off = s_off;
        r.group_off = 2 * s_off;
        r.ilen_off = 0;
        r.n_rows = 2;
        r.n_cols = m;
        r.n_sites = 1;
        r.n_indel_len = 0;
        a.res[j] = r;
        if (t_off + 2ull * m <= a.text_cap && s_off + 1 <= a.site_cap && 2 * s_off + 2 <= a.group_cap) {
            // (two loops, so that the staged one stores through an LDS pointer: one pointer for both would be a generic one, flat stores)
            if (staged) {
                char *o = stage + my_excl;
                for (uint32_t c = 0; 32 * c < m; ++c) {
                    const uint64_t x0 = oriented_chunk(w0, m, (ov0 & 1) != 0, c), x1 = oriented_chunk(w1, m, (ov1 & 1) != 0, c);
                    const uint32_t e = m - 32 * c < 32 ? m - 32 * c : 32;
                    for (uint32_t q = 0; q < e; ++q) {
                        o[32 * c + q] = pf::base_char((uint32_t)((x0 >> (62 - 2 * q)) & 3));
                        o[m + 32 * c + q] = pf::base_char((uint32_t)((x1 >> (62 - 2 * q)) & 3));
                    }
                }
            } else {
                char *o = a.otext + t_off;
                for (uint32_t c = 0; 32 * c < m; ++c) {
                    const uint64_t x0 = oriented_chunk(w0, m, (ov0 & 1) != 0, c), x1 = oriented_chunk(w1, m, (ov1 & 1) != 0, c);
                    const uint32_t e = m - 32 * c < 32 ? m - 32 * c : 32;
                    for (uint32_t q = 0; q < e; ++q) {
                        o[32 * c + q] = pf::base_char((uint32_t)((x0 >> (62 - 2 * q)) & 3));
                        o[m + 32 * c + q] = pf::base_char((uint32_t)((x1 >> (62 - 2 * q)) & 3));
                    }
                }
            }
            a.ogroups[2 * s_off] = 1;
            a.ogroups[2 * s_off + 1] = 2;
            pf_bubble_site sr;
            sr.col = col;
            sr.is_indel = 0;
            sr.maxnum = 2;
            sr.pad_ = 0;
            a.osites[s_off] = sr;
        }
    }
    if (tm && staged && wb + wave_total <= a.text_cap) {
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
        __builtin_amdgcn_wave_barrier();
        // (in words: the stage is word-aligned, global memory takes the unaligned word; the last one to three bytes singly)
        const uint32_t n_words = wave_total >> 2;
        char *dst = a.otext + wb;
        for (uint32_t x = lane; x < n_words; x += WAVE) {
            const uint32_t w = reinterpret_cast<const uint32_t *>(stage)[x];
            __builtin_memcpy(dst + 4 * (size_t)x, &w, 4);
        }
        for (uint32_t x = (n_words << 2) + lane; x < wave_total; x += WAVE) dst[x] = stage[x];
    }
    // the rest: K-PAIR when short, else K-BUBBLE's queue of their size class
    int key = KEY_NONE;
    if (active && !take) key = (a.stack_ok >= 3 && m <= STACK_MAX) ? KEY_STACK : (a.pair_ok && m <= PAIR_MAX) ? KEY_PAIR : (a.pair_ok && m <= PAIR_MAX2) ? KEY_PAIR2 : 2 * bubble_class(m, m) + (m > 64 ? 0 : 1);
    block_append(key, j, a.lists, a.cnt);
}

// ---------------------------------------------------------------------------------------------------------------------
// K-PAIR (pf_pair_dev.hpp): SequenceAlignment of two paths of at most 64 (tier 1) / 128 (tier 2) bases, one thread per bubble
struct PairArgs {
    const CallTask *ct;
    const uint32_t *kept;
    uint64_t t0;
    const uint64_t *seq, *off;
    const uint32_t *len;
    double M, D, G;
    int Mi, Di, Gi;               // the same as ints (integral scores)
    const uint32_t *list;         // this tier's bubbles (batch-local indices) ...
    const unsigned int *n_list;   // ... and how many (on the device: K-PREP and K-SNP fill the list)
    unsigned int *n_done;
    unsigned long long *prof;   // diagnostic (PF_PAIR_STATS): ticks of lane 0 in decode, fill, traceback, classify, publish; or nullptr
    uint8_t *scratch;           // PairGeom<NMAX>::scratch_bytes per wavefront of the grid
    pf_bubble_result *res;
    char *otext;
    uint64_t text_cap;
    pf_bubble_site *osites;
    uint64_t site_cap;
    uint8_t *ogroups;
    uint64_t group_cap;
    uint32_t *oilen;
    uint64_t ilen_cap;
    unsigned long long *heads;  // [0] text, [1] sites, [2] groups, [3] ilen
    CallLists lists;
    CallCounters *cnt;
};

__device__ inline unsigned long long wave_take(unsigned long long *head, uint32_t mine, uint32_t &excl) {
    // exclusive prefix of `mine` over the wavefront and one atomic for the total; returns the wavefront's base
    const int lane = lane_id();
    uint32_t incl = mine;
    for (int o = 1; o < WAVE; o <<= 1) {
        const uint32_t x = __shfl_up(incl, o, WAVE);
        if (lane >= o) incl += x;
    }
    excl = incl - mine;
    const uint32_t total = __shfl(incl, WAVE - 1, WAVE);
    unsigned long long base = 0;
    if (total) {
        if (lane == 0) base = atomicAdd(head, (unsigned long long)total);
        base = ((unsigned long long)__shfl((uint32_t)(base >> 32), 0, WAVE) << 32) | __shfl((uint32_t)base, 0, WAVE);
    }
    return base;
}

// too few bubbles for the second tier to fill the device (a launch lasts as long as one wavefront's 64 bubbles whatever their
// number): they join K-BUBBLE's queues of their size classes instead
__global__ __launch_bounds__(256) void k_call_pair2_reroute(PairArgs a) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    int key = KEY_NONE;
    uint32_t j = 0;
    if (i < *a.n_list) {
        j = a.list[i];
        const CallTask &t = a.ct[a.kept[a.t0 + j]];
        const uint32_t m = a.len[t.inner[0] >> 1], n = a.len[t.inner[1] >> 1];
        key = 2 * bubble_class(m, m > n ? m : n) + (m > 64 || n > 64 ? 0 : 1);
    }
    wave_append(key, j, a.lists, a.cnt);
}

template <int NMAX, bool INTEGRAL>
__global__ __launch_bounds__(64, NMAX == 64 ? 3 : 2) void k_call_pair(PairArgs a) {
    using Gm = PairGeom<NMAX>;
    const int lane = lane_id();
    PairMem mem;
    uint8_t *g = a.scratch + (uint64_t)blockIdx.x * Gm::scratch_bytes;
    mem.dir = reinterpret_cast<uint32_t *>(g) + lane;
    mem.ra = reinterpret_cast<char *>(g + Gm::dir_bytes) + lane;
    mem.rb = mem.ra + 64ull * Gm::LEN;
    mem.fa = mem.rb + 64ull * Gm::LEN;
    mem.fb = mem.fa + 64ull * Gm::LEN;
    const uint32_t n_list = *a.n_list;
    for (uint32_t base = blockIdx.x * 64; base < n_list; base += gridDim.x * 64) {
        const uint32_t i = base + lane;
        const bool active = i < n_list;
        uint32_t j = 0, L = 0, n_sites = 0, n_ilen = 0;
        int defer_key = KEY_NONE;
        bool defer = false;
        if (active) {
            j = a.list[i];
            const CallTask &t = a.ct[a.kept[a.t0 + j]];
            const uint32_t ov0 = t.inner[0], ov1 = t.inner[1];
            const uint32_t m = a.len[ov0 >> 1], n = a.len[ov1 >> 1];
            unsigned long long tq = a.prof ? wall_clock64() : 0;
            auto mark = [&](int slot) {
                if (!a.prof) return;
                const unsigned long long now = wall_clock64();
                if (lane == 0) atomicAdd(&a.prof[slot], now - tq);
                tq = now;
            };
            uint64_t Aw[Gm::NA], Bw[Gm::NA];
            uint32_t b0[Gm::NA], b1[Gm::NA];
            const uint64_t *w0 = a.seq + a.off[ov0 >> 1], *w1 = a.seq + a.off[ov1 >> 1];
#pragma unroll
            for (int c = 0; c < Gm::NA; ++c) {
                Aw[c] = 32u * c < m ? oriented_chunk(w0, m, (ov0 & 1) != 0, (uint32_t)c) : 0;
                Bw[c] = 32u * c < n ? oriented_chunk(w1, n, (ov1 & 1) != 0, (uint32_t)c) : 0;
                pair_planes(Bw[c], b0[c], b1[c]);
            }
            const int dmin = n < m ? (int)n - (int)m : 0;
            mark(0);
            pair_fill<NMAX, INTEGRAL>(mem.dir, Aw, b0, b1, m, dmin, a.M, a.D, a.G, a.Mi, a.Di, a.Gi);
            mark(1);
            L = pair_traceback<NMAX>(mem, Aw, Bw, m, n, dmin);
            mark(2);
            if (L == 0) {   // several optimal paths, a gap-open budget in the way, or a walk outside the band: K-BUBBLE's queue of the bubble's size class
                defer = true;
                defer_key = 2 * bubble_class(m, m > n ? m : n) + (m > 64 || n > 64 ? 0 : 1);
            } else {
                const PairCounts pc = pair_classify<false>(mem.fa, mem.fb, L, nullptr, nullptr);
                n_sites = pc.n_sites;
                n_ilen = pc.n_indel_len;
            }
            mark(3);
        }
        const unsigned long long tp0 = a.prof ? wall_clock64() : 0;
        const bool take = active && !defer;
        // pool space for the whole wavefront: one atomic per pool
        uint32_t e_text, e_sites, e_groups, e_ilen;
        const unsigned long long b_text = wave_take(&a.heads[0], take ? 2 * L : 0, e_text);
        const unsigned long long b_sites = wave_take(&a.heads[1], take ? n_sites : 0, e_sites);
        const unsigned long long b_groups = wave_take(&a.heads[2], take ? 2 * n_sites : 0, e_groups);
        const unsigned long long b_ilen = wave_take(&a.heads[3], take ? n_ilen : 0, e_ilen);
        if (take) {
            const unsigned long long t_off = b_text + e_text, s_off = b_sites + e_sites, g_off = b_groups + e_groups, l_off = b_ilen + e_ilen;
            pf_bubble_result r;
            r.rows_off = t_off;
            r.site_off = s_off;
            r.group_off = g_off;
            r.ilen_off = l_off;
            r.n_rows = 2;
            r.n_cols = L;
            r.n_sites = n_sites;
            r.n_indel_len = n_ilen;
            a.res[j] = r;
            if (t_off + 2ull * L <= a.text_cap && s_off + n_sites <= a.site_cap && g_off + 2ull * n_sites <= a.group_cap && l_off + n_ilen <= a.ilen_cap) {
                char *o = a.otext + t_off;
                for (uint32_t c = 0; c < L; ++c) { o[c] = PF_AT(mem.fa, c); o[L + c] = PF_AT(mem.fb, c); }
                (void)pair_classify<true>(mem.fa, mem.fb, L, a.osites + s_off, a.oilen + l_off);
                for (uint32_t q = 0; q < n_sites; ++q) { a.ogroups[g_off + 2 * q] = 1; a.ogroups[g_off + 2 * q + 1] = 2; }
            }
        }
        const unsigned long long done_m = __ballot(take);
        if (lane == 0 && done_m) atomicAdd(a.n_done, (unsigned int)__popcll(done_m));
        wave_append(defer_key, j, a.lists, a.cnt);
        if (a.prof && lane == 0) { atomicAdd(&a.prof[4], wall_clock64() - tp0); atomicAdd(&a.prof[5], 1ull); }
    }
}

// The column pass over R rows of length L (character j of row r at rows[r * row_stride + j * col_stride]), src/SeqAlign.cpp:56-157 as K-BUBBLE's classify + publish
// restate it: which columns are sites, which of them open an indel, the allele groups by first appearance over the rows, the
// indel lengths.  Counted, or with EMIT written out.
struct TrioCounts {
    uint32_t n_sites, n_ilen;
};
template <bool EMIT>
__device__ inline TrioCounts trio_classify(const char *rows, size_t row_stride, size_t col_stride, uint32_t R, uint32_t L, pf_bubble_site *sites, uint8_t *groups,
                                           uint32_t *ilen) {
    uint32_t ns = 0, nl = 0, last_indel_pos = 0;
    bool open = false;
    uint32_t prev_gap = 0;   // bit r: row r had a gap in the previous column
    for (uint32_t j = 0; j < L; ++j) {
        uint32_t seen = 0, n_seen = 0, gap = 0;   // `seen`: one bit per character class (A C G T -)
        for (uint32_t r = 0; r < R; ++r) {
            const char c = rows[(size_t)r * row_stride + (size_t)j * col_stride];
            const uint32_t cls = c == '-' ? 4u : (((uint32_t)(unsigned char)c >> 1) & 3u);
            if (!((seen >> cls) & 1u)) { seen |= 1u << cls; ++n_seen; }
            gap |= (c == '-' ? 1u : 0u) << r;
        }
        const bool same_status = j > 0 && gap == prev_gap;
        const int t = n_seen > 1 ? (gap ? 2 : 1) : 0;
        bool site = false, opens = false;
        if (t != 2) {
            if (open) { if (EMIT) ilen[nl] = j - last_indel_pos; nl++; open = false; }
            if (t == 1) site = true;
        } else {
            const bool same_run = open && same_status;
            if (open && !same_run) { if (EMIT) ilen[nl] = j - last_indel_pos; nl++; }
            if (!same_run) { last_indel_pos = j; open = true; site = true; opens = true; }
            else if (n_seen > 2) site = true;
        }
        if (site) {
            if (EMIT) {
                uint8_t *grp = groups + (size_t)ns * R;
                uint32_t tab = 0, next = 0;   // group of character class c in nibble c
                for (uint32_t r = 0; r < R; ++r) {
                    const char c = rows[(size_t)r * row_stride + (size_t)j * col_stride];
                    const uint32_t cls = c == '-' ? 4u : (((uint32_t)(unsigned char)c >> 1) & 3u);
                    uint32_t gq = (tab >> (4 * cls)) & 15u;
                    if (!gq) { gq = ++next; tab |= gq << (4 * cls); }
                    grp[r] = (uint8_t)gq;
                }
                pf_bubble_site sr;
                sr.col = j;
                sr.is_indel = opens ? 1 : 0;
                sr.maxnum = (uint8_t)next;
                sr.pad_ = 0;
                sites[ns] = sr;
            }
            ns++;
        }
        prev_gap = gap;
    }
    return TrioCounts{ns, nl};
}

// ---------------------------------------------------------------------------------------------------------------------
// K-STACK (pf_stack_dev.hpp): bubbles whose paths are all of one length -- the alignment is the paths stacked once every
// needlemanWunch(path 0, path p) is certified to have the diagonal as its single optimal path; one thread per bubble
constexpr uint32_t STACK_GAP_ROWS = 8;   // most rows of a bubble whose paths differ in length (its rows go through the wavefront's scratch)
__host__ __device__ inline uint64_t stack_scratch_bytes() { return (uint64_t)STACK_GAP_ROWS * STACK_MAX * 64; }

struct StackArgs {
    const uint32_t *list;
    const unsigned int *n_list;   // how many (on the device)
    uint8_t *scratch;             // stack_scratch_bytes() per wavefront of the grid
    uint32_t *oilen;
    uint64_t ilen_cap;
    int pair_ok;                  // two-path bubbles of the strict list that are not certified go to K-PAIR when they fit it (it runs behind this launch)
    const pf_bubble_task *btask;
    const pf_bubble_path *bpath;
    const char *ptext;          // path text of the branching bubbles (K-PATHS)
    const uint64_t *seq, *off;
    const uint32_t *len;
    int M, D, G;
    pf_bubble_result *res;
    char *otext;
    uint64_t text_cap;
    pf_bubble_site *osites;
    uint64_t site_cap;
    uint8_t *ogroups;
    uint64_t group_cap;
    unsigned long long *heads;  // [0] text, [1] sites, [2] groups, [3] ilen
    CallLists lists;
    int trio_ok;                // what is not certified goes to K-TRIO when it qualifies
    CallCounters *cnt;
};

__device__ inline void stack_load(const StackArgs &a, const pf_bubble_path &pp, StackPlanes &P) {
#pragma unroll
    for (int w = 0; w < 4; ++w) P.lo[w] = P.hi[w] = 0;
    if (pp.ov != NONE) {
        const uint64_t *w = a.seq + a.off[pp.ov >> 1];
#pragma unroll
        for (int c = 0; c < 4; ++c)
            if (32u * c < pp.len) pair_planes(oriented_chunk(w, pp.len, (pp.ov & 1) != 0, (uint32_t)c), P.lo[c], P.hi[c]);
    } else {
        const char *s = a.ptext + pp.text_off;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            uint32_t lo = 0, hi = 0;
            const uint32_t e = 32u * c < pp.len ? (pp.len - 32u * c < 32u ? pp.len - 32u * c : 32u) : 0u;
            for (uint32_t q = 0; q < e; ++q) {
                const uint32_t x = ((uint32_t)(unsigned char)s[32 * c + q] >> 1) & 3u;   // A 0, C 1, T 2, G 3
                const uint32_t code = x ^ (x >> 1);                                       // A 0, C 1, G 2, T 3
                lo |= (code & 1u) << q;
                hi |= (code >> 1) << q;
            }
            P.lo[c] = lo;
            P.hi[c] = hi;
        }
    }
}

__device__ inline uint32_t stack_code(const StackPlanes &P, uint32_t c) {
    uint32_t lo = P.lo[0], hi = P.hi[0];
#pragma unroll
    for (int w = 1; w < 4; ++w) { lo = (c >> 5) == (uint32_t)w ? P.lo[w] : lo; hi = (c >> 5) == (uint32_t)w ? P.hi[w] : hi; }
    return ((lo >> (c & 31)) & 1u) | (((hi >> (c & 31)) & 1u) << 1);
}

__global__ __launch_bounds__(64) void k_call_stack(StackArgs a) {
    const int lane = lane_id();
    const uint32_t n_list = *a.n_list;
    char *grows = reinterpret_cast<char *>(a.scratch + (uint64_t)blockIdx.x * stack_scratch_bytes()) + lane;   // rows of a gapped alignment, lane-interleaved
    constexpr size_t RS = (size_t)STACK_MAX * 64, CS = 64;
    for (uint32_t base = blockIdx.x * 64; base < n_list; base += gridDim.x * 64) {
        const uint32_t i = base + lane;
        const bool active = i < n_list;
        uint32_t j = 0, L = 0, n = 0, n_sites = 0, n_ilen = 0, l0 = 0, l1 = 0, lmax = 0;
        uint64_t first = 0;
        uint32_t U[4] = {0, 0, 0, 0};   // columns in which some path differs from path 0 (paths of one length)
        bool ok = false, gapped = false;
        if (active) {
            j = a.list[i];
            const pf_bubble_task bt = a.btask[j];
            n = bt.n_paths;
            first = bt.path_first;
            const pf_bubble_path p0 = a.bpath[first];
            l0 = L = lmax = p0.len;
            StackPlanes X, Y;
            stack_load(a, p0, X);
            // which kind: all paths as long as the first, or some shorter (a gap run in their rows); two paths may also have the
            // longer one second (strict bubbles are sorted by coverage): then row 0 takes the gaps
            for (uint32_t p = 1; p < n; ++p) {
                const uint32_t lp = a.bpath[first + p].len;
                if (p == 1) l1 = lp;
                if (lp != L) gapped = true;
                lmax = lp > lmax ? lp : lmax;
            }
            ok = true;
            if (!gapped) {
                for (uint32_t p = 1; p < n && ok; ++p) {
                    stack_load(a, a.bpath[first + p], Y);
                    ok = stack_certify(X, Y, L, a.M, a.D, a.G);
#pragma unroll
                    for (int w = 0; w < 4; ++w) U[w] |= (X.lo[w] ^ Y.lo[w]) | (X.hi[w] ^ Y.hi[w]);
                }
                n_sites = __popc(U[0]) + __popc(U[1]) + __popc(U[2]) + __popc(U[3]);
            } else if (n > STACK_GAP_ROWS) {
                ok = false;
            } else if (n == 2 && l1 > l0) {
                // the second path is the longer: the same certificate with the roles swapped (the recurrence is symmetric in its
                // two strings: UP and LEFT change places), and the gap run lies in row 0
                stack_load(a, a.bpath[first + 1], Y);
                const uint32_t d = l1 - l0, at = indel_place(Y, X, l0, d);
                ok = at != 0xFFFFFFFFu && indel_certify(Y, X, l1, l0, at, a.M, a.D, a.G);
                if (ok) {
                    L = l1;
                    for (uint32_t c = 0; c < L; ++c) {
                        grows[(size_t)c * CS] = (c < at) ? pf::base_char((uint32_t)(stack_code(X, c))) : (c < at + d ? '-' : pf::base_char((uint32_t)(stack_code(X, c - d))));
                        grows[RS + (size_t)c * CS] = pf::base_char((uint32_t)(stack_code(Y, c)));
                    }
                }
            } else {
                for (uint32_t c = 0; c < L; ++c) grows[(size_t)c * CS] = pf::base_char((uint32_t)(stack_code(X, c)));
                for (uint32_t p = 1; p < n && ok; ++p) {
                    const pf_bubble_path pp = a.bpath[first + p];
                    stack_load(a, pp, Y);
                    char *row = grows + (size_t)p * RS;
                    if (pp.len == L) {
                        ok = stack_certify(X, Y, L, a.M, a.D, a.G);
                        if (ok) for (uint32_t c = 0; c < L; ++c) row[(size_t)c * CS] = pf::base_char((uint32_t)(stack_code(Y, c)));
                    } else if (pp.len < L) {
                        const uint32_t d = L - pp.len, at = indel_place(X, Y, pp.len, d);
                        ok = at != 0xFFFFFFFFu && indel_certify(X, Y, L, pp.len, at, a.M, a.D, a.G);
                        if (ok) for (uint32_t c = 0; c < L; ++c) row[(size_t)c * CS] = (c < at) ? pf::base_char((uint32_t)(stack_code(Y, c))) : (c < at + d ? '-' : pf::base_char((uint32_t)(stack_code(Y, c - d))));
                    } else {
                        ok = false;   // a later path longer than the first: row 0 would take a gap
                    }
                }
            }
            if (ok && gapped) {
                const TrioCounts tc = trio_classify<false>(grows, RS, CS, n, L, nullptr, nullptr, nullptr);
                n_sites = tc.n_sites;
                n_ilen = tc.n_ilen;
            }
        }
        const bool take = active && ok;
        uint32_t e_text, e_sites, e_groups, e_ilen;
        const unsigned long long b_text = wave_take(&a.heads[0], take ? n * L : 0, e_text);
        const unsigned long long b_sites = wave_take(&a.heads[1], take ? n_sites : 0, e_sites);
        const unsigned long long b_groups = wave_take(&a.heads[2], take ? n * n_sites : 0, e_groups);
        const unsigned long long b_ilen = wave_take(&a.heads[3], take ? n_ilen : 0, e_ilen);
        if (take) {
            const unsigned long long t_off = b_text + e_text, s_off = b_sites + e_sites, g_off = b_groups + e_groups, l_off = b_ilen + e_ilen;
            pf_bubble_result r;
            r.rows_off = t_off;
            r.site_off = s_off;
            r.group_off = g_off;
            r.ilen_off = l_off;
            r.n_rows = n;
            r.n_cols = L;
            r.n_sites = n_sites;
            r.n_indel_len = n_ilen;
            a.res[j] = r;
            const bool room = t_off + (uint64_t)n * L <= a.text_cap && s_off + n_sites <= a.site_cap && g_off + (uint64_t)n * n_sites <= a.group_cap &&
                              l_off + n_ilen <= a.ilen_cap;
            if (room && gapped) {
                char *o = a.otext + t_off;
                for (uint32_t p = 0; p < n; ++p)
                    for (uint32_t c = 0; c < L; ++c) o[(size_t)p * L + c] = grows[(size_t)p * RS + (size_t)c * CS];
                (void)trio_classify<true>(grows, RS, CS, n, L, a.osites + s_off, a.ogroups + g_off, a.oilen + l_off);
            } else if (room) {
                // the rows, and per variant column the bases of all rows
                for (uint32_t p = 0; p < n; ++p) {
                    const pf_bubble_path pp = a.bpath[first + p];
                    StackPlanes Y;
                    stack_load(a, pp, Y);
                    char *o = a.otext + t_off + (uint64_t)p * L;
                    for (uint32_t c = 0; c < L; ++c) o[c] = pf::base_char((uint32_t)(stack_code(Y, c)));
                    // this row's base in every variant column, kept in the group bytes for now
                    uint32_t q = 0;
#pragma unroll
                    for (int w = 0; w < 4; ++w) {
                        uint32_t m = U[w];
                        while (m) {
                            const uint32_t c = 32u * w + (uint32_t)__ffs((int)m) - 1;
                            m &= m - 1;
                            a.ogroups[g_off + (uint64_t)q * n + p] = (uint8_t)stack_code(Y, c);
                            ++q;
                        }
                    }
                }
                // groups numbered by first appearance over the rows (src/SeqAlign.cpp:59-120), the site records
                uint32_t q = 0;
#pragma unroll
                for (int w = 0; w < 4; ++w) {
                    uint32_t m = U[w];
                    while (m) {
                        const uint32_t c = 32u * w + (uint32_t)__ffs((int)m) - 1;
                        m &= m - 1;
                        uint32_t tab = 0, maxnum = 0;   // group of base b in byte b
                        uint8_t *gp = a.ogroups + g_off + (uint64_t)q * n;
                        for (uint32_t p = 0; p < n; ++p) {
                            const uint32_t b = gp[p];
                            uint32_t gr = (tab >> (8 * b)) & 0xFFu;
                            if (!gr) { gr = ++maxnum; tab |= gr << (8 * b); }
                            gp[p] = (uint8_t)gr;
                        }
                        pf_bubble_site sr;
                        sr.col = c;
                        sr.is_indel = 0;
                        sr.maxnum = (uint8_t)maxnum;
                        sr.pad_ = 0;
                        a.osites[s_off + q] = sr;
                        ++q;
                    }
                }
            }
        }
        const unsigned long long done_m = __ballot(take);
        if (lane == 0 && done_m) atomicAdd(&a.cnt->n_stack_done, (unsigned int)__popcll(done_m));
        // not certified: two paths to K-PAIR (its fill decides, or finds the tie), the others to K-TRIO when that runs, else K-BUBBLE's
        // queue of the bubble's size class
        int key = KEY_NONE;
        if (active && !ok) {
            if (n == 2 && a.pair_ok && pair_fits<PAIR_MAX>(l0, l1)) key = KEY_PAIR;
            else if (n == 2 && a.pair_ok && pair_fits<PAIR_MAX2>(l0, l1)) key = KEY_PAIR2;
            else if (n == 2) key = 2 * bubble_class(l0, lmax) + (lmax > 64 ? 0 : 1);
            else if (a.trio_ok && (n == 3 || n == 4) && lmax <= TRIO_MAX && !gapped) key = n == 3 ? KEY_TRIO : KEY_TRIO4;
            else key = 2 * bubble_class(l0, lmax);
        }
        wave_append(key, j, a.lists, a.cnt);
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// K-TRIO: SeqAlign::SequenceAlignment (src/SeqAlign.cpp:550-640) for a bubble of three or four short paths -- when every round of
// the progressive alignment keeps ONE alignment that leaves row 0 as it is.  The paths of a branching bubble are sorted by
// descending length, so path 0 is the longest; a later path that differs from it by substitutions and a deletion aligns to it
// with gaps in its own row only.  Then round p is needlemanWunch(path 0, path p) of two gap-free strings, its one candidate
// re-opens no gap in the older rows, and the last round's one alignment is what compareStrPair (:8-236) receives: row 0 = path 0,
// row p = path p with its gaps.  Two kernels: K-TRIO-ALIGN, one THREAD per (bubble, path p >= 1) -- K-STACK's certificate when the
// two are equally long, else K-PAIR's fill and single-path traceback (pf_pair_dev.hpp); every lane of a wavefront has one fill
// to do, where a thread per bubble ran its rounds one after the other with the wavefront waiting for its longest lane --, and
// K-TRIO-FINISH, one thread per bubble: the column pass (:56-157) and the allele groups over the rows the first kernel left, or,
// when a round had several optimal paths, a gap in row 0, or a walk outside K-PAIR's band, the bubble's place in K-BUBBLE's queues.
constexpr uint32_t TRIO_ROW = TRIO_MAX;   // bytes of one row in the rows buffer

struct TrioArgs {
    const uint32_t *list;       // bubbles of NP paths ...
    uint32_t n_list;            // ... and how many (the host read the counter)
    const pf_bubble_task *btask;
    const pf_bubble_path *bpath;
    const char *ptext;
    const uint64_t *seq, *off;
    const uint32_t *len;
    int M, D, G;
    uint8_t *scratch;           // PairGeom<TRIO_MAX>::scratch_bytes per wavefront of the align grid
    char *rows;                 // n_list * NP rows of TRIO_ROW bytes: slot i of the list, row p
    uint8_t *okflag;            // n_list flags, preset to 1: a round that this tier cannot decide clears its bubble's
    pf_bubble_result *res;
    char *otext;
    uint64_t text_cap;
    pf_bubble_site *osites;
    uint64_t site_cap;
    uint8_t *ogroups;
    uint64_t group_cap;
    uint32_t *oilen;
    uint64_t ilen_cap;
    unsigned long long *heads;
    CallLists lists;
    CallCounters *cnt;
};

template <int NA>
__device__ inline void trio_load(const TrioArgs &a, const pf_bubble_path &pp, uint64_t (&W)[NA]) {
    if (pp.ov != NONE) {
        const uint64_t *w = a.seq + a.off[pp.ov >> 1];
#pragma unroll
        for (int c = 0; c < NA; ++c) W[c] = 32u * c < pp.len ? oriented_chunk(w, pp.len, (pp.ov & 1) != 0, (uint32_t)c) : 0;
    } else {
        const char *s = a.ptext + pp.text_off;
#pragma unroll
        for (int c = 0; c < NA; ++c) {
            uint64_t x = 0;
            const uint32_t e = 32u * c < pp.len ? (pp.len - 32u * c < 32u ? pp.len - 32u * c : 32u) : 0u;
            for (uint32_t q = 0; q < e; ++q) {
                const uint32_t y = ((uint32_t)(unsigned char)s[32 * c + q] >> 1) & 3u;   // A 0, C 1, T 2, G 3
                x |= (uint64_t)(y ^ (y >> 1)) << (62 - 2 * q);                            // A 0, C 1, G 2, T 3
            }
            W[c] = x;
        }
    }
}

template <int NP>
__global__ __launch_bounds__(64, 2) void k_call_trio_align(TrioArgs a) {
    using Gm = PairGeom<TRIO_MAX>;
    const int lane = lane_id();
    PairMem mem;
    uint8_t *g = a.scratch + (uint64_t)blockIdx.x * Gm::scratch_bytes;
    mem.dir = reinterpret_cast<uint32_t *>(g) + lane;
    mem.ra = reinterpret_cast<char *>(g + Gm::dir_bytes) + lane;
    mem.rb = mem.ra + 64ull * Gm::LEN;
    mem.fa = mem.rb + 64ull * Gm::LEN;
    mem.fb = mem.fa + 64ull * Gm::LEN;
    const uint32_t n_pairs = a.n_list * (NP - 1);
    for (uint32_t base = blockIdx.x * 64; base < n_pairs; base += gridDim.x * 64) {
        const uint32_t t = base + lane;
        if (t >= n_pairs) continue;
        const uint32_t slot = t / (NP - 1), p = 1 + t % (NP - 1);
        const uint32_t j = a.list[slot];
        const uint64_t first = a.btask[j].path_first;
        const pf_bubble_path p0 = a.bpath[first], pp = a.bpath[first + p];
        const uint32_t m = p0.len, n = pp.len;
        uint64_t Aw[Gm::NA], Bw[Gm::NA];
        trio_load<Gm::NA>(a, p0, Aw);
        trio_load<Gm::NA>(a, pp, Bw);
        StackPlanes X, Y;
#pragma unroll
        for (int w = 0; w < 4; ++w) X.lo[w] = X.hi[w] = Y.lo[w] = Y.hi[w] = 0;
#pragma unroll
        for (int c = 0; c < Gm::NA; ++c) { pair_planes(Aw[c], X.lo[c], X.hi[c]); pair_planes(Bw[c], Y.lo[c], Y.hi[c]); }
        char *row = a.rows + ((size_t)slot * NP + p) * TRIO_ROW;
        if (p == 1) {   // row 0 is path 0 itself
            char *r0 = a.rows + (size_t)slot * NP * TRIO_ROW;
            for (uint32_t c = 0; c < m; ++c) r0[c] = pair_base<TRIO_MAX>(Aw, c);
        }
        if (n == m && stack_certify(X, Y, m, a.M, a.D, a.G)) {
            for (uint32_t c = 0; c < m; ++c) row[c] = pair_base<TRIO_MAX>(Bw, c);
            continue;
        }
        const int dmin = n < m ? (int)n - (int)m : 0;
        uint32_t b0[Gm::NA], b1[Gm::NA];
#pragma unroll
        for (int c = 0; c < Gm::NA; ++c) { b0[c] = Y.lo[c]; b1[c] = Y.hi[c]; }
        pair_fill<TRIO_MAX, true>(mem.dir, Aw, b0, b1, m, dmin, 0.0, 0.0, 0.0, a.M, a.D, a.G);
        const uint32_t L = pair_traceback<TRIO_MAX>(mem, Aw, Bw, m, n, dmin);
        if (L != m) { a.okflag[slot] = 0; continue; }   // several optimal paths / outside the band (0), or a gap in row 0 (longer than m)
        for (uint32_t c = 0; c < m; ++c) row[c] = PF_AT(mem.fb, c);
    }
}

template <int NP>
__global__ __launch_bounds__(256) void k_call_trio_finish(TrioArgs a) {
    const int lane = lane_id();
    const uint32_t slot = blockIdx.x * blockDim.x + threadIdx.x;
    const bool active = slot < a.n_list;
    uint32_t j = 0, m = 0, n_sites = 0, n_ilen = 0;
    bool ok = false;
    const char *rows = a.rows + (size_t)slot * NP * TRIO_ROW;
    if (active) {
        j = a.list[slot];
        m = a.bpath[a.btask[j].path_first].len;
        ok = a.okflag[slot] != 0;
        if (ok) {
            const TrioCounts tc = trio_classify<false>(rows, TRIO_ROW, 1, NP, m, nullptr, nullptr, nullptr);
            n_sites = tc.n_sites;
            n_ilen = tc.n_ilen;
        }
    }
    const bool take = active && ok;
    uint32_t e_text, e_sites, e_groups, e_ilen;
    const unsigned long long b_text = wave_take(&a.heads[0], take ? NP * m : 0, e_text);
    const unsigned long long b_sites = wave_take(&a.heads[1], take ? n_sites : 0, e_sites);
    const unsigned long long b_groups = wave_take(&a.heads[2], take ? NP * n_sites : 0, e_groups);
    const unsigned long long b_ilen = wave_take(&a.heads[3], take ? n_ilen : 0, e_ilen);
    if (take) {
        const unsigned long long t_off = b_text + e_text, s_off = b_sites + e_sites, g_off = b_groups + e_groups, l_off = b_ilen + e_ilen;
        pf_bubble_result r;
        r.rows_off = t_off;
        r.site_off = s_off;
        r.group_off = g_off;
        r.ilen_off = l_off;
        r.n_rows = NP;
        r.n_cols = m;
        r.n_sites = n_sites;
        r.n_indel_len = n_ilen;
        a.res[j] = r;
        if (t_off + (uint64_t)NP * m <= a.text_cap && s_off + n_sites <= a.site_cap && g_off + (uint64_t)NP * n_sites <= a.group_cap &&
            l_off + n_ilen <= a.ilen_cap) {
            char *o = a.otext + t_off;
            for (uint32_t p = 0; p < NP; ++p)
                for (uint32_t c = 0; c < m; ++c) o[(size_t)p * m + c] = rows[(size_t)p * TRIO_ROW + c];
            (void)trio_classify<true>(rows, TRIO_ROW, 1, NP, m, a.osites + s_off, a.ogroups + g_off, a.oilen + l_off);
        }
    }
    const unsigned long long done_m = __ballot(take);
    if (lane == 0 && done_m) atomicAdd(&a.cnt->n_trio_done, (unsigned int)__popcll(done_m));
    int key = KEY_NONE;
    if (active && !ok) key = 2 * bubble_class(m, m);   // (three paths or more: the class's heavy queue)
    wave_append(key, j, a.lists, a.cnt);
}

// ---------------------------------------------------------------------------------------------------------------------
// K-PATHS
struct PathArgs {
    const CallTask *ct;
    const uint32_t *kept;
    uint64_t t0;
    uint32_t nb;
    const uint32_t *blist;
    const uint32_t *succ;
    const uint64_t *seq, *off;
    const uint32_t *len;
    int k;
    uint32_t depth_cap;     // entries of the major stack (complex size + slack); minor holds 4x
    uint8_t *scratch;       // per wave, for the bubbles whose stacks outgrow the registers: major[depth_cap], minor[4 depth_cap], seg_start
    uint64_t scratch_per_wave;
    int force_scratch;      // (tests) every bubble walks with the stacks in scratch
    // A bubble of more than 255 walks leaves the first launch for a list (mlist) and is walked again by a second launch of a few
    // wavefronts whose path tables lie in global scratch (max_paths entries) instead of LDS.
    uint32_t max_paths;     // 255 (tables in LDS) or PATHS_BIG
    const unsigned int *n_list;   // how many entries of blist count
    uint32_t *mlist;
    uint32_t mlist_cap;
    // colored: the distinct oriented unitigs the walks of a bubble visit, in the order of their first visit (what CCDBG searches for
    // the first k-mer of a site string, src/CCDBG.cpp:3251, 3390): walk_off[j] = first entry in walk_pool | count << 40
    uint32_t *walk_pool;    // nullptr: single-sample
    uint64_t *walk_off;
    uint64_t walk_cap;
    pf_bubble_task *btask;
    pf_bubble_path *bpath;  // strict region [0, 4 nb), then the pool
    uint64_t path_cap;      // entries available behind the strict region
    char *text;
    uint64_t text_cap;
    uint32_t *queues;
    uint32_t *klist;        // K-STACK's list of branching bubbles
    int stack_ok;
    uint32_t *tlist, *tlist4;   // K-TRIO's lists (three / four paths)
    int trio_ok;
    CallCounters *cnt;
};

constexpr uint32_t MAX_PATHS = 255;        // walks of one bubble whose tables fit LDS
constexpr uint32_t PATHS_BIG = 65535;      // ... in the second launch's global tables

// what the walk of one bubble leaves behind; everything wave-uniform
struct WalkOut {
    uint32_t n_paths;
    bool too_many, too_deep, text_ok;
    uint32_t lmax, lmin;
    uint64_t sum;
    uint32_t n_seen, seen_reg;   // colored: distinct vertices visited (seen_reg: entry x in lane x; after the scratch walk they lie in its scratch)
};
struct TextChunk { unsigned long long cur, end; };   // a wave's piece of the text pool (one atomic per ~40 paths)

__device__ inline void walk_reset(WalkOut &o) { o = WalkOut{0, false, false, true, 0, 0xFFFFFFFFu, 0, 0, 0}; }

__device__ inline unsigned long long take_text(const PathArgs &a, TextChunk &tx, uint32_t total, int lane) {
    if (total > tx.end - tx.cur) {
        const unsigned long long want = total > 4096u ? total : 4096u;
        unsigned long long got = 0;
        if (lane == 0) got = atomicAdd(&a.cnt->text_head, want);
        got = ((unsigned long long)read_lane((uint32_t)(got >> 32), 0) << 32) | read_lane((uint32_t)got, 0);
        tx.cur = got;
        tx.end = got + want;
    }
    const unsigned long long at = tx.cur;
    tx.cur += total;
    return at;
}

__device__ inline void note_path(WalkOut &o, const PathArgs &a, unsigned long long *poff, uint32_t *plen, unsigned long long at,
                                 uint32_t total, int lane) {
    if (lane == 0) { poff[o.n_paths] = at; plen[o.n_paths] = total; }
    o.lmax = total > o.lmax ? total : o.lmax;
    o.lmin = total < o.lmin ? total : o.lmin;
    o.sum += total;
    if (at + total > a.text_cap) o.text_ok = false;
    ++o.n_paths;
}

__device__ inline char path_char(const PathArgs &a, const uint32_t *major, const uint32_t *seg_start, uint32_t n_seg, uint32_t pos,
                                 uint32_t first_idx) {
    // segment holding character `pos` of the path string: linear search, segments are few
    uint32_t x = 0;
    while (x + 1 < n_seg && seg_start[x + 1] <= pos) ++x;
    const uint32_t idx = pos - seg_start[x] + (x == 0 ? first_idx : 0);
    return pf::base_char((uint32_t)(oriented_base(a.seq, a.off, a.len, major[x], idx)));
}

// Two-stack enumeration of every s -> t walk (src/CDBG.cpp:1364-1412) with both stacks in REGISTERS: entry x of the major stack
// lives in lane x -- the oriented unitig with its length, its word offset and its four successors beside it -- and entry x of the
// minor stack in lane x & 63 of register x >> 6.  A push is one predicated move, a read one v_readlane; the only memory the
// walk itself touches is the successor row of a vertex when it is entered.  The string of a walk (one character of s, the
// first len - k + 1 of every inner unitig, the first k of t) is cut with one scan over the lanes and written 64 characters at
// a time, every lane fetching the one packed word that holds its base.  Returns false when a stack outgrows the registers
// (64 / 256 entries): the caller repeats the bubble with the stacks in global scratch (walk_in_scratch).
__device__ inline bool walk_in_registers(const PathArgs &a, const CallTask &t, TextChunk &tx, unsigned long long *poff, uint32_t *plen,
                                         WalkOut &o, const int lane) {
    const uint32_t eu = t.exit_ov >> 1;
    const uint32_t K = (uint32_t)a.k;
    const uint32_t first_idx = a.len[t.u] - K;   // s gives the first character of its last k-mer
    uint32_t mj = 0, ml = 0, s0 = NONE, s1 = NONE, s2 = NONE, s3 = NONE;
    unsigned long long mo = 0;
    uint32_t mn0 = 0, mn1 = 0, mn2 = 0, mn3 = 0;
    uint32_t n_major = 0, n_minor = 1;
    uint32_t seen = NONE;   // colored: distinct vertices in the order of their first visit, entry x in lane x
    o.n_seen = 0;
    if (lane == 0) mn0 = t.entrance_ov;
    auto minor_top = [&]() {
        const uint32_t x = n_minor - 1, r = x >> 6;
        const uint32_t v = r == 0 ? mn0 : r == 1 ? mn1 : r == 2 ? mn2 : mn3;
        return read_lane(v, (int)(x & 63));
    };
    while (n_minor) {
        const uint32_t w = minor_top();
        --n_minor;
        if (n_major >= a.depth_cap) { o.too_deep = true; return true; }
        if (n_major >= 64) return false;
        if (a.walk_pool && !__ballot((uint32_t)lane < o.n_seen && seen == w)) {
            if (o.n_seen >= WAVE) return false;
            if ((uint32_t)lane == o.n_seen) seen = w;
            ++o.n_seen;
        }
        const uint32_t u = w >> 1;
        const bool at_exit = u == eu;
        uint32_t r0 = NONE, r1 = NONE, r2 = NONE, r3 = NONE;
        if (!at_exit) {
            const uint32_t *r = a.succ + (size_t)w * 4;
            r0 = r[0]; r1 = r[1]; r2 = r[2]; r3 = r[3];
        }
        if ((uint32_t)lane == n_major) { mj = w; ml = a.len[u]; mo = a.off[u]; s0 = r0; s1 = r1; s2 = r2; s3 = r3; }
        ++n_major;
        if (at_exit) {
            if (o.n_paths >= a.max_paths) { o.too_many = true; return true; }
            const uint32_t cnt = (uint32_t)lane < n_major ? (lane == 0 ? 1u : ((uint32_t)lane + 1 == n_major ? K : ml - K + 1)) : 0u;
            const uint32_t incl = scan_u32_dpp<0>(cnt);
            const uint32_t start = incl - cnt;
            uint32_t total = read_lane(incl, (int)n_major - 1);
            if (n_major == 1) total = 0;  // s == t cannot be a bubble; keep the arithmetic sane
            const unsigned long long at = take_text(a, tx, total, lane);
            if (at + total <= a.text_cap)
                for (uint32_t p0 = 0; p0 < total; p0 += WAVE) {
                    const uint32_t p = p0 + (uint32_t)lane;
                    uint32_t mine = 0;
                    for (uint32_t x = 1; x < n_major; ++x)
                        if (p >= read_lane(start, (int)x)) mine = x;
                    const uint32_t sw = (uint32_t)__shfl((int)mj, (int)mine), sl = (uint32_t)__shfl((int)ml, (int)mine);
                    const uint32_t ss = (uint32_t)__shfl((int)start, (int)mine);
                    const uint32_t so_lo = (uint32_t)__shfl((int)(uint32_t)mo, (int)mine), so_hi = (uint32_t)__shfl((int)(uint32_t)(mo >> 32), (int)mine);
                    if (p < total) {
                        const uint32_t idx = p - ss + (mine == 0 ? first_idx : 0u);
                        const bool rev = (sw & 1) != 0;
                        const uint32_t j = rev ? sl - 1 - idx : idx;
                        const uint64_t word = a.seq[(((uint64_t)so_hi << 32) | so_lo) + (j >> 5)];
                        uint32_t b = (uint32_t)(word >> (62 - 2 * (j & 31))) & 3u;
                        b = rev ? 3 - b : b;
                        a.text[at + p] = (char)((0x54474341u >> (8 * b)) & 0xFFu);   // "ACGT"[b]
                    }
                }
            note_path(o, a, poff, plen, at, total, lane);
            --n_major;
            while (n_major && n_minor) {
                const int top = (int)n_major - 1;
                const uint32_t nx = minor_top();
                if (read_lane(s0, top) == nx || read_lane(s1, top) == nx || read_lane(s2, top) == nx || read_lane(s3, top) == nx) break;
                --n_major;
            }
        } else {
            const uint32_t rr[4] = {r0, r1, r2, r3};
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                const uint32_t x = rr[b];
                if (x == NONE) continue;
                if (n_minor >= 4 * a.depth_cap) { o.too_deep = true; return true; }
                if (n_minor >= 256) return false;
                const uint32_t r = n_minor >> 6;
                if ((uint32_t)lane == (n_minor & 63)) {
                    if (r == 0) mn0 = x;
                    else if (r == 1) mn1 = x;
                    else if (r == 2) mn2 = x;
                    else mn3 = x;
                }
                ++n_minor;
            }
        }
    }
    o.seen_reg = seen;
    return true;
}

// The same walk with the stacks in the wave's global scratch: any depth the complex size allows.
// (its arguments by value: a reference to the kernel's argument block would move the whole block into private memory for every use)
__device__ __noinline__ void walk_in_scratch(const PathArgs a, const CallTask &t, TextChunk &tx, unsigned long long *poff, uint32_t *plen,
                                             WalkOut &o, uint32_t *major, uint32_t *minor, uint32_t *seg_start, uint32_t *seen, const int lane) {
    const uint32_t eu = t.exit_ov >> 1;
    const uint32_t ulen = a.len[t.u] - (uint32_t)a.k + 1;
    uint32_t n_major = 0, n_minor = 0;
    o.n_seen = 0;
    if (lane == 0) minor[0] = t.entrance_ov;
    n_minor = 1;
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
    __builtin_amdgcn_wave_barrier();
    while (n_minor && !o.too_many && !o.too_deep) {
        const uint32_t w = minor[n_minor - 1];
        --n_minor;
        if (n_major >= a.depth_cap) { o.too_deep = true; break; }
        if (lane == 0) major[n_major] = w;
        ++n_major;
        if (a.walk_pool) {
            bool known = false;
            for (uint32_t x0 = 0; x0 < o.n_seen && !known; x0 += WAVE) known = __ballot(x0 + lane < o.n_seen && seen[x0 + lane] == w) != 0;
            if (!known) {
                if (o.n_seen >= 4 * a.depth_cap) { o.too_deep = true; break; }
                if (lane == 0) seen[o.n_seen] = w;
                ++o.n_seen;
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
        __builtin_amdgcn_wave_barrier();
        if ((w >> 1) == eu) {
            if (o.n_paths >= a.max_paths) { o.too_many = true; break; }
            uint32_t total = 0;
            for (uint32_t x = 0; x < n_major; ++x) {
                if (lane == 0) seg_start[x] = total;
                const uint32_t wl = a.len[major[x] >> 1] - (uint32_t)a.k + 1;
                total += x == 0 ? 1u : (x + 1 == n_major ? (uint32_t)a.k : wl);
            }
            if (n_major == 1) total = 0;
            const unsigned long long at = take_text(a, tx, total, lane);
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
            __builtin_amdgcn_wave_barrier();
            if (at + total <= a.text_cap)
                for (uint32_t p = lane; p < total; p += WAVE) a.text[at + p] = path_char(a, major, seg_start, n_major, p, ulen - 1);
            note_path(o, a, poff, plen, at, total, lane);
            --n_major;
            while (n_major && n_minor) {
                const uint32_t *r = a.succ + (size_t)major[n_major - 1] * 4;
                const uint32_t nx = minor[n_minor - 1];
                if (r[0] == nx || r[1] == nx || r[2] == nx || r[3] == nx) break;
                --n_major;
            }
        } else {
            const uint32_t *r = a.succ + (size_t)w * 4;
            for (int b = 0; b < 4; ++b) {
                const uint32_t x = r[b];
                if (x == NONE) continue;
                if (n_minor >= 4 * a.depth_cap) { o.too_deep = true; break; }
                if (lane == 0) minor[n_minor] = x;
                ++n_minor;
            }
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
            __builtin_amdgcn_wave_barrier();
        }
    }
}

// K-PATHS' list keys: the alignment queues 0 .. NQ-1, then K-STACK's and K-TRIO's lists
constexpr uint32_t PK_STACK = NQ, PK_TRIO3 = NQ + 1, PK_TRIO4 = NQ + 2, PK_NONE = 0xFFFFFFFFu;

// appends the wave's pending entries (entry x in lane x, n of them) to their lists: one atomic per list that occurs
__device__ inline void paths_flush(const PathArgs &a, uint32_t pend_key, uint32_t pend_j, uint32_t n, int lane) {
    const bool have = (uint32_t)lane < n;
    unsigned long long todo = __ballot(have);
    while (todo) {
        const uint32_t key = read_lane(pend_key, __ffsll((long long)todo) - 1);
        const bool mine = have && pend_key == key;
        const unsigned long long m = __ballot(mine);
        // (one select after the other on integers, no nested choice of pointers: hipcc 7.2 turned the nested form into branches that
        // left the counter's address unset for the last key -- found with rocgdb on a bubble list that had all four kinds)
        uint32_t c_off = (uint32_t)offsetof(CallCounters, n_trio4);
        c_off = key == PK_TRIO3 ? (uint32_t)offsetof(CallCounters, n_trio) : c_off;
        c_off = key == PK_STACK ? (uint32_t)offsetof(CallCounters, n_stack_b) : c_off;
        c_off = key < (uint32_t)NQ ? (uint32_t)offsetof(CallCounters, q_n) + 4u * key : c_off;
        unsigned int *counter = reinterpret_cast<unsigned int *>(reinterpret_cast<char *>(a.cnt) + c_off);
        uint64_t l_at = (uint64_t)(uintptr_t)a.tlist4;
        l_at = key == PK_TRIO3 ? (uint64_t)(uintptr_t)a.tlist : l_at;
        l_at = key == PK_STACK ? (uint64_t)(uintptr_t)a.klist : l_at;
        l_at = key < (uint32_t)NQ ? (uint64_t)(uintptr_t)(a.queues + (size_t)key * a.nb) : l_at;
        uint32_t *list = reinterpret_cast<uint32_t *>((uintptr_t)l_at);
        uint32_t base = 0;
        if (lane == 0) base = atomicAdd(counter, (unsigned int)__popcll(m));
        base = read_lane(base, 0);
        if (mine) list[base + (uint32_t)__popcll(m & ((1ull << lane) - 1))] = pend_j;
        todo &= ~m;
    }
}

constexpr uint32_t PATHS_PAIRWISE = 8;   // up to this many walks the wave orders them pair by pair, 64 characters per step

template <bool BIG>
__global__ __launch_bounds__(64) void k_call_paths(PathArgs a) {
    const int lane = lane_id();
    __shared__ unsigned long long poff_lds[BIG ? 1 : 256];
    __shared__ uint32_t plen_lds[BIG ? 1 : 256];
    uint8_t *scr = a.scratch + (uint64_t)blockIdx.x * a.scratch_per_wave;
    uint32_t *major = reinterpret_cast<uint32_t *>(scr);                        // depth_cap
    uint32_t *minor = major + a.depth_cap;                                      // 4 depth_cap
    uint32_t *seg_start = minor + 4 * a.depth_cap;                              // depth_cap + 1
    uint32_t *seen_scr = seg_start + a.depth_cap + 1;                           // 4 depth_cap (colored)
    // (BIG: the path tables behind them, 8-aligned)
    unsigned long long *poff = BIG ? reinterpret_cast<unsigned long long *>(scr + ((((uint64_t)10 * a.depth_cap + 4) * 4 + 7) & ~7ull)) : poff_lds;
    uint32_t *plen = BIG ? reinterpret_cast<uint32_t *>(poff + a.max_paths + 1) : plen_lds;
    const uint32_t n_branching = *a.n_list;
    TextChunk tx{0, 0}, px{0, 0}, wx{0, 0};
    uint32_t pend_key = PK_NONE, pend_j = 0, n_pend = 0;   // list entries not yet appended: entry x in lane x
    unsigned long long need_retry = 0, need_max = 0;
    for (uint32_t q = blockIdx.x; q < n_branching; q += gridDim.x) {   // (bubbles cost about the same: no queue head to fight over)
        const uint32_t j = a.blist[q];
        const CallTask &t = a.ct[a.kept[a.t0 + j]];
        WalkOut wo;
        walk_reset(wo);
        __builtin_amdgcn_wave_barrier();   // (the loop before may still be reading poff / plen)
        bool seen_in_scratch = false;
        if (a.force_scratch || !walk_in_registers(a, t, tx, poff, plen, wo, lane)) {
            walk_reset(wo);
            walk_in_scratch(a, t, tx, poff, plen, wo, major, minor, seg_start, seen_scr, lane);
            seen_in_scratch = true;
        }
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
        __builtin_amdgcn_wave_barrier();
        const uint32_t n_paths = wo.n_paths;
        if (wo.too_many || wo.too_deep) {
            if (lane == 0) {
                if (!BIG && !wo.too_deep && a.mlist) {   // more walks than the LDS tables hold: the second launch's
                    const uint32_t at = atomicAdd(&a.cnt->n_many, 1u);
                    if (at < a.mlist_cap) a.mlist[at] = j;   // (beyond: the host sees n_many > mlist_cap, grows the list and repeats the attempt)
                } else {
                    atomicOr(&a.cnt->err, wo.too_many ? 1u : 32u);
                    a.cnt->err_entrance = t.entrance_ov;
                    a.cnt->err_exit = t.exit_ov;
                }
                a.btask[j] = pf_bubble_task{0, 0, 0};
            }
            continue;
        }
        // ---- sortSeq_branching (src/CDBG.cpp:417-480): descending length, ties by descending strcmp.  Distinct walks spell
        //      distinct strings, so the order is total and a rank sort gives what the reference's quicksort gives ----
        if (a.walk_pool) {   // colored: the vertices visited, for K-SITES
            const uint32_t ns = wo.n_seen;
            if (ns > wx.end - wx.cur) {
                const unsigned long long want = ns > 256u ? ns : 256u;
                unsigned long long got = 0;
                if (lane == 0) got = atomicAdd(&a.cnt->walk_head, want);
                got = ((unsigned long long)read_lane((uint32_t)(got >> 32), 0) << 32) | read_lane((uint32_t)got, 0);
                wx.cur = got;
                wx.end = got + want;
            }
            const unsigned long long at = wx.cur;
            wx.cur += ns;
            if (at + ns <= a.walk_cap) {
                if (seen_in_scratch) { for (uint32_t x = lane; x < ns; x += WAVE) a.walk_pool[at + x] = seen_scr[x]; }
                else if ((uint32_t)lane < ns) a.walk_pool[at + lane] = wo.seen_reg;
            }
            if (lane == 0) a.walk_off[j] = at | ((unsigned long long)ns << 40);
        }
        if (n_paths > px.end - px.cur) {   // path entries come in pieces of 128 like the text (holes in the pool are harmless)
            const unsigned long long want = n_paths > 128u ? n_paths : 128u;
            unsigned long long got = 0;
            if (lane == 0) got = atomicAdd(&a.cnt->path_head, want);
            got = ((unsigned long long)read_lane((uint32_t)(got >> 32), 0) << 32) | read_lane((uint32_t)got, 0);
            px.cur = got;
            px.end = got + want;
        }
        const unsigned long long first = px.cur;
        px.cur += n_paths;
        const bool fits = first + n_paths <= a.path_cap;
        const bool text_ok = wo.text_ok;
        const uint32_t lmax = wo.lmax, lmin = wo.lmin;
        const uint64_t sum = wo.sum;
        if (fits && text_ok && n_paths <= PATHS_PAIRWISE) {
            // a few walks (nearly every bubble): the wave takes the pairs one by one, 64 characters of both strings per step;
            // lane i counts the walks that come before walk i
            uint32_t rank = 0;
            for (uint32_t i = 0; i + 1 < n_paths; ++i) {
                const uint32_t li = read_lane(plen[i], 0);
                const unsigned long long oi = poff[i];
                for (uint32_t o = i + 1; o < n_paths; ++o) {
                    const uint32_t lo = read_lane(plen[o], 0);
                    bool o_first = lo > li;   // (identical strings cannot occur; if they did, the earlier walk goes first)
                    if (lo == li) {
                        const unsigned long long oo = poff[o];
                        for (uint32_t c0 = 0; c0 < li; c0 += WAVE) {
                            const uint32_t c = c0 + (uint32_t)lane;
                            const uint32_t ci = c < li ? (unsigned char)a.text[oi + c] : 0u, co = c < li ? (unsigned char)a.text[oo + c] : 0u;
                            const unsigned long long diff = __ballot(ci != co);
                            if (diff) {
                                const int at = __ffsll((long long)diff) - 1;
                                o_first = read_lane(co, at) > read_lane(ci, at);
                                break;
                            }
                        }
                    }
                    if ((uint32_t)lane == (o_first ? i : o)) ++rank;
                }
            }
            if ((uint32_t)lane < n_paths) a.bpath[(size_t)4 * a.nb + first + rank] = pf_bubble_path{poff[lane], plen[lane], PF_NONE};
        } else if (fits && text_ok) {
            for (uint32_t i = lane; i < n_paths; i += WAVE) {
                const char *si = a.text + poff[i];
                const uint32_t li = plen[i];
                uint32_t rank = 0;
                for (uint32_t o = 0; o < n_paths; ++o) {
                    if (o == i) continue;
                    const uint32_t lo = plen[o];
                    bool before;
                    if (lo != li) before = lo > li;
                    else {
                        const char *so = a.text + poff[o];
                        uint32_t p = 0;
                        while (p < li && so[p] == si[p]) ++p;
                        before = p < li ? (unsigned char)so[p] > (unsigned char)si[p] : o < i;
                    }
                    rank += before;
                }
                a.bpath[(size_t)4 * a.nb + first + rank] = pf_bubble_path{poff[i], li, PF_NONE};
            }
        } else if (lane == 0) {
            atomicOr(&a.cnt->err, 8u);
        }
        if (lane == 0) a.btask[j] = pf_bubble_task{(uint64_t)4 * a.nb + first, n_paths, 0};
        if (BIG && lane == 0) atomicMax(&a.cnt->max_rows, n_paths);
        // where the bubble goes next (all of this is wave-uniform): the list entry waits in the wave's registers, the two sizes in
        // its running maxima -- one atomic per list and 64 bubbles instead of three per bubble on one cache line, which is what
        // bounded this kernel (profiles/r08_experiments.txt)
        uint32_t key = PK_NONE;
        if (n_paths >= 2 && fits && text_ok) {
            const unsigned long long jb = (unsigned long long)job_bytes((uint32_t)(sum < 60000 ? sum : 60000), lmax);
            need_retry = jb > need_retry ? jb : need_retry;
            if (a.stack_ok && lmax <= STACK_MAX &&
                (sum == (uint64_t)n_paths * lmax ? n_paths <= STACK_PATHS : (a.stack_ok >= 2 && n_paths <= STACK_GAP_ROWS))) {
                // K-STACK looks at them first (thread per bubble: paths of one length, or shorter than the first by one gap run) and
                // hands on what it cannot certify
                key = PK_STACK;
            } else if (n_paths >= 3 && a.trio_ok && n_paths <= 4 && lmax <= TRIO_MAX && lmax - lmin <= (uint32_t)PairGeom<TRIO_MAX>::MAX_SKEW) {
                // a few short paths: K-TRIO (thread per bubble) aligns each to the first and hands on what needs more than that
                key = n_paths == 3 ? PK_TRIO3 : PK_TRIO4;
                if (bubble_class(lmax, lmax) == kBubLdsClasses) {
                    const unsigned long long bn = (unsigned long long)bubble_need(lmax, lmax);
                    need_max = bn > need_max ? bn : need_max;
                }
            } else {
                const int c = bubble_class(lmax, lmax);  // sorted by length: the first path is the longest
                key = (uint32_t)(2 * c + ((n_paths > 2 || lmax > 64) ? 0 : 1));
                if (c == kBubLdsClasses) {
                    const unsigned long long bn = (unsigned long long)bubble_need(lmax, lmax);
                    need_max = bn > need_max ? bn : need_max;
                }
            }
        }
        if (key != PK_NONE) {
            if ((uint32_t)lane == n_pend) { pend_key = key; pend_j = j; }
            if (++n_pend == WAVE) { paths_flush(a, pend_key, pend_j, n_pend, lane); n_pend = 0; }
        }
    }
    if (n_pend) paths_flush(a, pend_key, pend_j, n_pend, lane);
    if (lane == 0) {
        if (need_retry) atomicMax(&a.cnt->retry_need, need_retry);
        if (need_max) atomicMax(&a.cnt->max_need, need_max);
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// K-SITES
struct SiteArgs {
    const CallTask *ct;
    const uint32_t *kept;
    uint64_t t0;
    const uint32_t *blist;
    const pf_bubble_result *res;
    const char *otext;
    pf_bubble_site *osites;     // pad_ receives the site's ok flag
    const uint8_t *ogroups;
    int k;
    const CountLine *tab;
    uint64_t mask;
    int one_strand, tab_exact;
    uint32_t low, up;
    uint32_t ks;                // capacity of one site string
    uint32_t rows_cap;          // rows the per-wave tables hold (a multiple of 64, >= the most walks of any bubble)
    uint8_t *scratch;
    uint64_t scratch_per_wave;
    uint64_t *sv_off;           // per bubble: first value in sv
    double *sv;                 // per site: maxnum group coverages, then their sum
    uint64_t sv_cap;
    CallCounters *cnt;
    unsigned long long *prof;   // PF_SITES_STATS: per wavefront {total, pop + load, strings, ranks + probes, groups, bubbles}
    // colored (CCDBG): the joined table of all colours with one (low, up) per colour, the colour sets, the graph and the vertices
    // each bubble's walks visit (K-PATHS).  A site then has n_colors * maxnum values -- group coverage [colour][group] -- and its
    // verdict: pad_ = 1 iff no string failed a colour's range test and every colour covers some string in full.
    uint32_t n_colors;
    CTab ctab;
    int c_one_strand;
    uint64_t c_unread;
    const uint32_t *clow, *cup;
    const uint64_t *full;
    const uint32_t *part_first, *part_colour;
    const uint64_t *part_word, *part_bits;
    const uint32_t *walk_pool;
    const uint64_t *walk_off;
    const uint64_t *seq, *off;
    const uint32_t *len;
};

// UnitigColors::contains(um, colour) (bifrost/src/ColorSet.cpp:776-823) for the mapping [dist, dist + n_km) of unitig u: the colour
// on every one of those k-mers
__device__ inline bool colour_contains(const SiteArgs &a, uint32_t u, uint32_t c, uint32_t dist, uint32_t n_km) {
    if ((a.full[u] >> c) & 1) return true;
    for (uint32_t e = a.part_first[u]; e < a.part_first[u + 1]; ++e) {
        if (a.part_colour[e] != c) continue;
        const uint64_t *bits = a.part_bits + a.part_word[e];
        for (uint32_t i = dist; i < dist + n_km; ++i)
            if (!((bits[i >> 6] >> (i & 63)) & 1)) return false;
        return true;
    }
    return false;
}

// cdbg.findUnitig(s, 0, len) of src/CCDBG.cpp:3251, 3390 followed by UnitigColors::contains on that mapping, for one site string:
// its first k-mer lies on one of the bubble's unitigs (a k-mer occurs once in the graph, in one orientation); the mapping is extended
// along that unitig while the characters agree (CompactedDBG.tcc:3815-3837, CompressedSequence.cpp:497-520).  Returns the mask of
// colours present on every k-mer of the mapping; found = false when no unitig of the bubble holds the first k-mer.
__device__ inline uint64_t colours_of_string(const SiteArgs &a, const char *sp, uint32_t lp, const uint32_t *walk, uint32_t n_walk, bool &found) {
    const int k = a.k;
    const uint64_t kmask = (1ull << (2 * k)) - 1;
    found = false;
    if (lp < (uint32_t)k) return 0;
    auto code = [](char ch) -> int { return ch == 'A' ? 0 : ch == 'C' ? 1 : ch == 'G' ? 2 : ch == 'T' ? 3 : -1; };
    uint64_t head = 0;
    for (int i = 0; i < k; ++i) {
        const int b = code(sp[i]);
        if (b < 0) return 0;   // (a gap character never matches a unitig)
        head = (head << 2) | (uint64_t)b;
    }
    const uint64_t rhead = rc_kmer(head, k);
    for (uint32_t q = 0; q < n_walk; ++q) {
        const uint32_t u = walk[q] >> 1;
        const uint32_t Lu = a.len[u];
        const uint64_t *w = a.seq + a.off[u];
        auto base_at = [&](uint32_t x) -> int { return (int)((w[x >> 5] >> (62 - 2 * (x & 31))) & 3u); };
        uint64_t y = 0, word = 0;
        int hit = 0;   // 1 forward, 2 reverse complement
        uint32_t p0 = 0;
        // (std::string::find(head) over the whole unitig first, then find(rhead): a unitig holds at most one of the two)
        for (uint32_t pos = 0; pos < Lu && hit != 1; ++pos) {
            if ((pos & 31) == 0) word = w[pos >> 5];
            y = ((y << 2) | ((word >> (62 - 2 * (pos & 31))) & 3u)) & kmask;
            if (pos + 1 < (uint32_t)k) continue;
            if (y == head) { hit = 1; p0 = pos + 1 - (uint32_t)k; }
            else if (y == rhead && !hit) { hit = 2; p0 = pos + 1 - (uint32_t)k; }
        }
        if (!hit) continue;
        uint32_t dist, n_km;
        if (hit == 1) {
            uint32_t jn = (uint32_t)k;
            while (jn < lp && p0 + jn < Lu && code(sp[jn]) == base_at(p0 + jn)) ++jn;
            n_km = jn - (uint32_t)k + 1;
            dist = p0;
        } else {
            long ps = (long)p0 + k - 1;
            uint32_t jn = 0;
            while (jn < lp && ps >= 0 && code(sp[jn]) == 3 - base_at((uint32_t)ps)) { ++jn; --ps; }
            n_km = jn - (uint32_t)k + 1;
            dist = p0 - (n_km - 1);
        }
        uint64_t m = 0;
        for (uint32_t c = 0; c < a.n_colors; ++c)
            if (colour_contains(a, u, c, dist, n_km)) m |= 1ull << c;
        found = true;
        return m;
    }
    return 0;
}

template <bool COLORED>
__global__ __launch_bounds__(64) void k_call_sites(SiteArgs a) {
    const int lane = lane_id();
    const uint32_t KS = a.ks;
    const uint32_t C = COLORED ? a.n_colors : 1;
    const int k = a.k;
    uint8_t *scr = a.scratch + (uint64_t)blockIdx.x * a.scratch_per_wave;
    // per row (rows_cap rows: 256, more when a bubble of the batch has more walks): appended characters, final string, and the per-row scalars
    const size_t RC = a.rows_cap;
    char *app = reinterpret_cast<char *>(scr);
    char *fin = app + RC * KS;
    uint32_t *flen = reinterpret_cast<uint32_t *>(fin + RC * KS);
    uint32_t *at = flen + RC;
    uint32_t *rank = at + RC;
    uint8_t *dup = reinterpret_cast<uint8_t *>(rank + RC);
    uint8_t *sok = dup + RC;
    double *mean = reinterpret_cast<double *>(sok + RC);
    // colored, per row: the colours its string's mapping carries in full, the colours whose range test it passed, its mean per colour
    uint64_t *cmask = reinterpret_cast<uint64_t *>(mean + RC);
    uint64_t *cokm = cmask + RC;
    double *cmean = reinterpret_cast<double *>(cokm + RC);   // [rows_cap][C]
    const uint64_t kmask = (1ull << (2 * k)) - 1;
    const uint32_t n_branching = a.cnt->n_branching;
    unsigned long long my_strings = 0;   // (added to the batch's count once, at the end)
    unsigned long long pk[6] = {0, 0, 0, 0, 0, 0};
    const unsigned long long pk0 = a.prof ? wall_clock64() : 0;
    auto sync = [] {
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
        __builtin_amdgcn_wave_barrier();
    };
    // No shared queue head and one bump of the value pool per SV_CHUNK values instead of one per bubble: two atomics per bubble on
    // two addresses -- 60 000 per launch, served one after the other by the L2 -- were what the launch took, whatever its
    // wavefronts did in between.  Wavefront w takes bubbles w, w + grid, ...
    constexpr uint32_t SV_CHUNK = 1024;
    unsigned long long chunk_at = 0;
    uint32_t chunk_left = 0;
    // (... for all but the last two rounds, which are handed out one bubble at a time: the wavefronts finish together)
    const uint32_t rounds = n_branching / gridDim.x;
    const uint32_t n_static = rounds > 2 ? (rounds - 2) * gridDim.x : 0;
    uint32_t q_static = blockIdx.x;
    for (;;) {
        uint32_t q;
        if (q_static < n_static) {
            q = q_static;
            q_static += gridDim.x;
        } else {
            q = 0;
            if (lane == 0) q = atomicAdd(&a.cnt->sites_next, 1u);
            q = n_static + read_lane(q, 0);
            if (q >= n_branching) break;
        }
        const unsigned long long pa = a.prof ? wall_clock64() : 0;
        const uint32_t j = a.blist[q];
        const pf_bubble_result r = a.res[j];
        if (r.n_rows == 0 || r.n_rows == 0xFFFFFFFFu) continue;
        ++pk[5];
        const uint32_t R = r.n_rows, L = r.n_cols;
        const char *rows = a.otext + r.rows_off;
        // values: one slot per allele group and site, plus the site's sum
        uint32_t n_val = 0;
        for (uint32_t si = 0; si < r.n_sites; ++si) n_val += C * (uint32_t)a.osites[r.site_off + si].maxnum + 1;
        unsigned long long v0 = 0;
        if (n_val > chunk_left) {   // (wave-uniform)
            const uint32_t take = n_val > SV_CHUNK ? n_val : SV_CHUNK;
            if (lane == 0) v0 = atomicAdd(&a.cnt->sv_head, (unsigned long long)take);
            chunk_at = ((unsigned long long)read_lane((uint32_t)(v0 >> 32), 0) << 32) | read_lane((uint32_t)v0, 0);
            chunk_left = take;
        }
        v0 = chunk_at;
        chunk_at += n_val;
        chunk_left -= n_val;
        if (lane == 0) a.sv_off[j] = v0;
        const bool room = v0 + n_val <= a.sv_cap;
        uint32_t err = 0, n_strings = 0;
        uint32_t indel = 0;
        unsigned long long vcur = v0;
        if (a.prof) pk[1] += wall_clock64() - pa;
        for (uint32_t si = 0; si < r.n_sites && !err; ++si) {
            const unsigned long long pb = a.prof ? wall_clock64() : 0;
            const pf_bubble_site sr = a.osites[r.site_off + si];
            const uint8_t *grp = a.ogroups + r.group_off + (uint64_t)si * R;
            const uint32_t site = sr.col;
            const uint32_t maxnum = sr.maxnum;
            // A site that is no indel in a bubble that has met none so far -- most sites -- takes k raw columns of every row
            // (src/CDBG.cpp:1559-1596): equally long strings over {-, A, C, G, T}, k <= 31.  Lane p holds row p's string as
            // order-preserving 3-bit codes in two registers (comparing them = comparing the strings) next to the 2-bit k-mer
            // the probe wants; ranks, duplicates and the groups' sums go through lane reads instead of the scratch arrays,
            // whose every access is a step in a chain of dependent loads.  Same decisions, same order of the additions.
            const long plain_from = (long)site - k + 1;
            if (!COLORED && !sr.is_indel && indel == 0 && plain_from >= 0 && (uint64_t)plain_from + (uint64_t)k <= L && k <= 31 && R <= WAVE) {
                const uint32_t p = (uint32_t)lane;
                const bool mine = p < R;
                uint64_t hi = 0, lo = 0, km = 0;
                uint32_t g = 0;
                if (mine) {
                    // (all 32 bytes asked for at once -- a loop of loads would wait for each in turn; the row pool is allocated
                    // with slack, so the bytes past the k-th exist)
                    unsigned char cs[32];
                    __builtin_memcpy(cs, rows + (size_t)p * L + plain_from, 32);
#pragma unroll
                    for (int x = 0; x < 31; ++x) {   // (constant indices: cs stays in registers)
                        if (x >= k) continue;
                        const char ch = (char)cs[x];
                        const uint64_t c3 = ch == '-' ? 0 : ch == 'A' ? 1 : ch == 'C' ? 2 : ch == 'G' ? 3 : 4;
                        hi = (hi << 3) | (lo >> 61);
                        lo = (lo << 3) | c3;
                        km = (km << 2) | (ch == 'A' ? 0 : ch == 'C' ? 1 : ch == 'G' ? 2 : 3);
                    }
                    km &= kmask;
                    g = grp[p];
                }
                bool d = false;
                uint32_t rk = 0;
                for (uint32_t o = 0; o < R; ++o) {
                    const uint64_t ohi = ((uint64_t)read_lane((uint32_t)(hi >> 32), (int)o) << 32) | read_lane((uint32_t)hi, (int)o);
                    const uint64_t olo = ((uint64_t)read_lane((uint32_t)(lo >> 32), (int)o) << 32) | read_lane((uint32_t)lo, (int)o);
                    const uint32_t og = read_lane(g, (int)o);
                    if (!mine || o == p || og != g) continue;
                    if (ohi == hi && olo == lo) { if (o < p) d = true; }
                    else if (ohi < hi || (ohi == hi && olo < lo)) ++rk;
                }
                const unsigned long long pc = a.prof ? wall_clock64() : 0;
                pk[2] += pc - pb;
                bool miss = false, okp = true;
                double mn = 0.0;
                if (mine && !d) {
                    uint64_t sum = 0;
                    if (!a.tab_exact) {
                        uint32_t cnt;
                        if (!canonical_count(a.tab, a.mask, km, k, cnt, a.one_strand != 0)) miss = true;
                        else if (cnt > a.low && cnt < a.up) sum = cnt;
                        else okp = false;   // src/CDBG.cpp:45-50
                    }
                    mn = (double)sum;   // (one k-mer: the mean is its count)
                }
                n_strings += (uint32_t)__popcll(__ballot(mine && !d));
                const unsigned long long pd = a.prof ? wall_clock64() : 0;
                pk[3] += pd - pc;
                // group coverages in set order; the first string out of range drops the site (src/CDBG.cpp:1527-1551) -- and ends the
                // loop: a string BEHIND it is never looked up, so a k-mer of such a string that is in no database ends nothing
                // (the reference's exit sits inside readCov, :52-56).  Only a missing k-mer the walk reaches is the run's end.
                bool ok = true, fatal = false;
                double total = 0.0;
                for (uint32_t gi = 0; gi < maxnum; ++gi) {
                    double tc = 0.0;
                    if (ok) {
                        uint32_t last = 0;
                        bool have_last = false;
                        for (;;) {
                            const bool cand = mine && g == gi + 1 && !d && (!have_last || rk > last);
                            const unsigned long long m = __ballot(cand);
                            if (!m) break;
                            const uint32_t best = read_lane(wave_min_u32(cand ? rk : 0xFFFFFFFFu), 0);
                            const int bp = __ffsll((long long)__ballot(cand && rk == best)) - 1;   // (distinct strings of a group: distinct ranks)
                            const uint32_t verdict = read_lane(miss ? 2u : (okp ? 1u : 0u), bp);
                            if (verdict == 2) { fatal = true; ok = false; break; }
                            if (!verdict) { ok = false; break; }
                            const uint64_t mb = (uint64_t)__double_as_longlong(mn);
                            tc += __longlong_as_double((long long)(((uint64_t)read_lane((uint32_t)(mb >> 32), bp) << 32) | read_lane((uint32_t)mb, bp)));
                            last = best;
                            have_last = true;
                        }
                        if (ok) total += tc;
                    }
                    if (lane == 0 && room) a.sv[vcur + gi] = tc;
                }
                if (fatal) { err = 2; break; }
                if (lane == 0 && room) a.sv[vcur + maxnum] = total;
                if (lane == 0) a.osites[r.site_off + si].pad_ = ok ? 1 : 0;
                vcur += maxnum + 1;
                if (a.prof) pk[4] += wall_clock64() - pd;
                continue;
            }
            uint32_t napp = 0;
            if (sr.is_indel) {
                // every path: the next non-gap character at / after the site, again and again until the characters just appended
                // are not all equal (src/CDBG.cpp:1474-1493)
                for (uint32_t p = lane; p < R; p += WAVE) at[p] = site;
                sync();
                // A row that ends in gaps runs out here: the reference's `substr(pos, 1)` at pos == size() is the EMPTY string --
                // nothing is appended to that row, '\0' joins the set of characters (so the loop ends unless every row ran out at
                // once) and pos moves on to size() + 1, where the next substr throws and the reference terminates.  A row that ran
                // out keeps at[p] = L + 1 and has one character fewer than the others.
                for (;;) {
                    uint32_t bad = 0;
                    for (uint32_t base = 0; base < R; base += WAVE) {
                        const uint32_t p = base + lane;
                        bool e = false;
                        if (p < R) {
                            uint32_t x = at[p];
                            if (x > L) e = true;   // (every row ran out in the round before: std::out_of_range in the reference)
                            else {
                                while (x < L && rows[(size_t)p * L + x] == '-') ++x;
                                if (napp < KS) app[(size_t)p * KS + napp] = x < L ? rows[(size_t)p * L + x] : '\0';
                                at[p] = x + 1;
                            }
                        }
                        if (__ballot(e)) bad = 1;
                    }
                    if (napp >= KS) {   // (still growing: ask for the most a string can reach -- a row's characters and its raw columns)
                        bad = 16;
                        if (lane == 0) atomicMax(&a.cnt->ks_need, 2 * L + (uint32_t)k + 2);
                    }
                    sync();
                    if (bad) { err = bad == 16 ? 16 : 4; break; }
                    const char first = app[napp];
                    bool differ = false;
                    for (uint32_t base = 0; base < R; base += WAVE) {
                        const uint32_t p = base + lane;
                        if (__ballot(p < R && app[(size_t)p * KS + napp] != first)) differ = true;
                    }
                    ++napp;
                    if (differ) break;
                }
                if (err) break;
            }
            // the k-length string of every path around the site (src/CDBG.cpp:1494-1525, 1559-1596)
            uint32_t row_err = 0;
            for (uint32_t base = 0; base < R; base += WAVE) {
                const uint32_t p = base + lane;
                uint32_t e = 0;
                if (p < R) {
                    const char *row = rows + (size_t)p * L;
                    char *out = fin + (size_t)p * KS;
                    uint32_t n = 0;
                    auto push = [&](char c) { if (n < KS) out[n] = c; ++n; };
                    if (sr.is_indel) {
                        const uint32_t napp_p = at[p] > L ? napp - 1 : napp;   // (a row that ran out in the last round)
                        const long need = (long)k - (long)napp_p;
                        if (indel == 0) {
                            // substr(site - k + n, k - n): a start below zero or past the row throws; a negative count is npos
                            const long from = (long)site - k + (long)napp_p;
                            if (from < 0 || (uint64_t)from > L) e = 4;
                            else {
                                const uint32_t take = need < 0 ? (uint32_t)(L - (uint64_t)from) : (uint32_t)std::min<uint64_t>((uint64_t)need, L - (uint64_t)from);
                                for (uint32_t x = 0; x < take; ++x) push(row[from + x]);
                                for (uint32_t x = 0; x < napp_p; ++x) push(app[(size_t)p * KS + x]);
                            }
                        } else {
                            uint32_t c = 0;
                            for (uint32_t x = 0; x < site && x < L; ++x) c += row[x] != '-';
                            if (need < 0 || (long)c < need) {
                                for (uint32_t x = 0; x < site && x < L; ++x)
                                    if (row[x] != '-') push(row[x]);
                                for (uint32_t x = 0; x < napp_p; ++x) push(app[(size_t)p * KS + x]);
                                for (uint32_t x = at[p]; n < (uint32_t)k; ++x) {
                                    if (x >= L) { e = 4; break; }
                                    if (row[x] != '-') push(row[x]);
                                }
                            } else {
                                // the last `need` non-gap characters before the site
                                uint32_t skip = c - (uint32_t)need;
                                for (uint32_t x = 0; x < site && x < L; ++x) {
                                    if (row[x] == '-') continue;
                                    if (skip) { --skip; continue; }
                                    push(row[x]);
                                }
                                for (uint32_t x = 0; x < napp_p; ++x) push(app[(size_t)p * KS + x]);
                            }
                        }
                    } else if (indel > 0) {
                        uint32_t c = 0;
                        for (uint32_t x = 0; x <= site && x < L; ++x) c += row[x] != '-';
                        if (c < (uint32_t)k) {
                            for (uint32_t x = 0; x <= site && x < L; ++x)
                                if (row[x] != '-') push(row[x]);
                            for (uint32_t x = site + 1; n < (uint32_t)k; ++x) {
                                if (x >= L) { e = 4; break; }
                                if (row[x] != '-') push(row[x]);
                            }
                        } else {
                            uint32_t skip = c - (uint32_t)k;
                            for (uint32_t x = 0; x <= site && x < L; ++x) {
                                if (row[x] == '-') continue;
                                if (skip) { --skip; continue; }
                                push(row[x]);
                            }
                        }
                    } else {
                        const long from = (long)site - k + 1;
                        if (from < 0 || (uint64_t)from > L) e = 4;
                        else {
                            const uint32_t take = (uint32_t)std::min<uint64_t>((uint64_t)k, L - (uint64_t)from);
                            for (uint32_t x = 0; x < take; ++x) push(row[from + x]);
                        }
                    }
                    if (n > KS) { e = 16; atomicMax(&a.cnt->ks_need, n); }
                    flen[p] = n;
                }
                if (__ballot(e == 4)) row_err |= 4;
                if (__ballot(e == 16)) row_err |= 16;
            }
            if (sr.is_indel) ++indel;
            sync();
            const unsigned long long pc = a.prof ? wall_clock64() : 0;
            pk[2] += pc - pb;
            if (row_err) { err = (row_err & 4) ? 4 : 16; break; }
            // distinct strings per allele group in std::set order, their coverage (readCov(string), src/CDBG.cpp:29-60)
            for (uint32_t base = 0; base < R; base += WAVE) {
                const uint32_t p = base + lane;
                bool miss = false;
                if (p < R) {
                    const char *sp = fin + (size_t)p * KS;
                    const uint32_t lp = flen[p];
                    const uint8_t g = grp[p];
                    bool d = false;
                    uint32_t rk = 0;
                    for (uint32_t o = 0; o < R; ++o) {
                        if (o == p || grp[o] != g) continue;
                        const char *so = fin + (size_t)o * KS;
                        const uint32_t lo = flen[o], lm = lo < lp ? lo : lp;
                        uint32_t x = 0;
                        while (x < lm && so[x] == sp[x]) ++x;
                        int cmp;  // so <=> sp
                        if (x < lm) cmp = (unsigned char)so[x] < (unsigned char)sp[x] ? -1 : 1;
                        else cmp = lo < lp ? -1 : (lo > lp ? 1 : 0);
                        if (cmp == 0) { if (o < p) d = true; }
                        else if (cmp < 0) ++rk;
                    }
                    dup[p] = d;
                    rank[p] = rk;   // counts duplicates of smaller strings as well: only the order of the ranks matters
                    uint8_t ok = 1;
                    double mn = 0.0;
                    if (COLORED) {
                        if (!d) {
                            // readCov(string, low, up, c) (src/CCDBG.cpp:89-122) for every colour: one look at the joined table per k-mer
                            uint64_t okm = C >= 64 ? ~0ull : ((1ull << C) - 1);
                            uint64_t *cs = reinterpret_cast<uint64_t *>(cmean + (size_t)p * C);
                            for (uint32_t c = 0; c < C; ++c) cs[c] = 0;
                            StringWindow win;
                            for (uint32_t c0 = 0; c0 < lp; ++c0) {
                                const uint64_t x = win.push(sp[c0], kmask, (uint32_t)k);
                                if (c0 + 1 < (uint32_t)k) continue;
                                const uint32_t *sa, *sb;
                                colored_slots(a.ctab, x, k, a.c_one_strand != 0, sa, sb);
                                for (uint32_t c = 0; c < C; ++c) {
                                    if (!((okm >> c) & 1) || ((a.c_unread >> c) & 1)) continue;   // (a colour never looked up: (0, true))
                                    const uint32_t cnt = ctab_count(sa, sb, c);
                                    if (cnt != CTAB_MISSING && cnt > a.clow[c] && cnt < a.cup[c]) cs[c] += cnt;
                                    else { cs[c] = 0; okm &= ~(1ull << c); }   // missing or outside (low, up): (0, false), :105-117
                                }
                            }
                            for (uint32_t c = 0; c < C; ++c) cmean[(size_t)p * C + c] = (double)cs[c] / (double)((uint64_t)lp - (uint64_t)k + 1);
                            cokm[p] = okm;
                            const uint64_t wf = a.walk_off[j];
                            bool found;
                            cmask[p] = colours_of_string(a, sp, lp, a.walk_pool + (wf & ((1ull << 40) - 1)), (uint32_t)(wf >> 40), found);
                            if (!found) cmask[p] = 1ull << 63;   // (findUnitig finds nothing: fatal in the reference IF the walk below reaches this string)
                        }
                    } else if (!d) {
                        uint64_t sum = 0;
                        if (!a.tab_exact) {
                            StringWindow win;
                            for (uint32_t c = 0; c < lp; ++c) {
                                const uint64_t x = win.push(sp[c], kmask, (uint32_t)k);
                                if (c + 1 >= (uint32_t)k) {
                                    uint32_t cnt;
                                    if (!canonical_count(a.tab, a.mask, x, k, cnt, a.one_strand != 0)) { miss = true; break; }
                                    if (cnt > a.low && cnt < a.up) sum += cnt;
                                    else { sum = 0; ok = 0; break; }  // src/CDBG.cpp:45-50
                                }
                            }
                        }
                        mn = (double)sum / (double)((uint64_t)lp - (uint64_t)k + 1);
                    }
                    sok[p] = miss ? 2 : ok;   // (2: a k-mer in no database -- the reference's exit, if the walk below reaches this string)
                    mean[p] = mn;
                }
                n_strings += (uint32_t)__popcll(__ballot(p < R && !dup[p]));
            }
            sync();
            const unsigned long long pd = a.prof ? wall_clock64() : 0;
            pk[3] += pd - pc;
            bool fatal = false;
            if (COLORED) {
                // src/CCDBG.cpp:3236-3339, 3374-3475: per allele group its strings in set order; a colour the string's mapping carries
                // in full adds the string's mean to [colour][group]; a failed range test of such a colour, or a colour no string
                // carries, drops the site.  Lane c keeps colour c's sums.
                bool ok = true;
                uint64_t seen = 0;
                for (uint32_t gi = 0; gi < maxnum; ++gi) {
                    double tc = 0.0;
                    uint32_t last = 0;
                    bool have_last = false;
                    while (ok) {
                        uint32_t best = 0xFFFFFFFFu, bp = 0;
                        for (uint32_t p = 0; p < R; ++p) {
                            if (grp[p] != gi + 1 || dup[p]) continue;
                            const uint32_t rk = rank[p];
                            if (have_last && rk <= last) continue;
                            if (rk < best) { best = rk; bp = p; }
                        }
                        if (best == 0xFFFFFFFFu) break;
                        const uint64_t m = cmask[bp];
                        if (m >> 63) { fatal = true; ok = false; break; }
                        seen |= m;
                        if (m & ~cokm[bp]) { ok = false; break; }
                        if ((uint32_t)lane < C && ((m >> lane) & 1)) tc += cmean[(size_t)bp * C + lane];
                        last = best;
                        have_last = true;
                    }
                    if ((uint32_t)lane < C && room) a.sv[vcur + (uint64_t)lane * maxnum + gi] = tc;
                }
                if (fatal) { err = 64; break; }
                const uint64_t all_colours = C >= 64 ? ~0ull : ((1ull << C) - 1);
                const bool valid = ok && seen == all_colours;
                if (lane == 0 && room) a.sv[vcur + (uint64_t)C * maxnum] = valid ? 1.0 : 0.0;
                if (lane == 0) a.osites[r.site_off + si].pad_ = valid ? 1 : 0;
                vcur += (uint64_t)C * maxnum + 1;
                sync();
                if (a.prof) pk[4] += wall_clock64() - pd;
                continue;
            }
            // group coverages in set order; the first string out of range drops the site (src/CDBG.cpp:1527-1551)
            bool ok = true;
            double total = 0.0;
            for (uint32_t gi = 0; gi < maxnum; ++gi) {
                double tc = 0.0;
                if (ok) {
                    // rows of this group, ascending rank: wave-uniform selection of the next smallest rank
                    uint32_t last = 0;
                    bool have_last = false;
                    for (;;) {
                        uint32_t best = 0xFFFFFFFFu, bp = 0;
                        for (uint32_t p = 0; p < R; ++p) {
                            if (grp[p] != gi + 1 || dup[p]) continue;
                            const uint32_t rk = rank[p];
                            if (have_last && rk <= last) continue;
                            if (rk < best) { best = rk; bp = p; }
                        }
                        if (best == 0xFFFFFFFFu) break;
                        if (sok[bp] == 2) { fatal = true; ok = false; break; }
                        if (!sok[bp]) { ok = false; break; }
                        tc += mean[bp];
                        last = best;
                        have_last = true;
                    }
                    if (ok) total += tc;
                }
                if (lane == 0 && room) a.sv[vcur + gi] = tc;
            }
            if (fatal) { err = 2; break; }
            if (lane == 0 && room) a.sv[vcur + maxnum] = total;
            if (lane == 0) a.osites[r.site_off + si].pad_ = ok ? 1 : 0;
            vcur += maxnum + 1;
            sync();
            if (a.prof) pk[4] += wall_clock64() - pd;
        }
        if (lane == 0 && err) atomicOr(&a.cnt->err, err);
        my_strings += n_strings;
    }
    if (lane == 0 && my_strings) atomicAdd(&a.cnt->site_strings, my_strings);
    if (a.prof && lane == 0) {
        pk[0] = wall_clock64() - pk0;
        for (int x = 0; x < 6; ++x) a.prof[(size_t)blockIdx.x * 6 + x] = pk[x];
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// K-TEXT
// indel_len[indel - 1] as the callers print it (src/CDBG.cpp:1310, 1565; src/CCDBG.cpp:3034, 3315, 3450).  An indel run still open
// at the last column never has its length pushed (compareStrPair closes a run only on a later column, src/SeqAlign.cpp:56-157), so
// for that last indel site the reference reads one element past the vector: heap garbage that changes from run to run, or a null
// dereference when the vector is empty -- rows ending in gaps, i.e. gap-friendly scores only.  Undefined there; defined here (and in
// the oracle, pfo::indel_len_at) as the length of the open run: columns - the site's column.
__device__ __forceinline__ uint32_t open_run_len(const uint32_t *ilen, uint32_t i, uint32_t n_ilen, uint32_t n_cols, uint32_t col) {
    return i < n_ilen ? ilen[i] : n_cols - col;
}

struct FmtArgs {
    const CallTask *ct;
    const uint32_t *kept;
    uint64_t t0;              // selection index of the aligned batch's first bubble
    uint32_t j0;              // first bubble of this text batch inside the aligned batch
    uint32_t nb;              // bubbles of this text batch
    const pf_bubble_result *res;
    const char *otext;
    const pf_bubble_site *osites;
    const uint8_t *ogroups;
    const uint32_t *oilen;
    const uint64_t *sv_off;
    const double *sv;
    const uint32_t *vc;       // inclusive count of called bubbles inside the batch
    uint64_t vc_base;
    int mt;                   // the reference's -t > 1 format: var_count from 0, allele_frequency rows grouped by arity per bubble
    const uint32_t *len;
    uint32_t *sizes;          // [N_INT][nb + 1]
    const uint64_t *offs;     // exclusive scan of sizes, one run over all streams
    char *out[N_INT];
    int packed;               // alignseq leaves as out[S_PACK] = index + records (pf_alnpack.hpp); out[PF_OUT_ALIGNSEQ] is not written
    CallCounters *cnt;
    // colored (CCDBG): a site gives one row per colour that sees two allele groups or more (src/CCDBG.cpp:2971-3059, 3236-3339)
    uint32_t n_colors, N;
    int k;
    const uint64_t *full, *ccov_sum;
};

// computeCramerVCoefficient (src/CCDBG.cpp:330-366) on rows ca and cb of a [colour][allele] coverage matrix given as val(colour, allele)
template <class Val>
__device__ inline double cramer_v_dev(const Val &val, uint32_t ca, uint32_t cb, uint32_t n_alleles) {
#pragma clang fp contract(off)
    double n = 0, nA = 0, nB = 0, chi = 0;
    uint32_t count = 0;
    for (uint32_t i = 0; i < n_alleles; ++i) {
        const double A = val(ca, i), B = val(cb, i), p = A + B;
        nA += A;
        nB += B;
        n = n + p;
        if (p != 0) ++count;
    }
    if ((count & 255u) < 2) return 0;   // (the reference counts in a uint8_t)
    for (uint32_t i = 0; i < n_alleles; ++i) {
        const double A = val(ca, i), B = val(cb, i), p = A + B;
        if (p == 0) continue;
        const double exA = nA * p / n, exB = nB * p / n;
        const double dA = A - exA, dB = B - exB;
        chi += dA * dA / exA;   // pow(x, 2) is x * x, correctly rounded, in glibc as here
        chi += dB * dB / exB;
    }
    return sqrt(chi / n);
}
// the largest over all colour pairs (:2964-2970, 3285-3291); std::max keeps its first argument when the second is NaN
template <class Val>
__device__ inline double max_cramer_v_dev(const Val &val, uint32_t n_colors, uint32_t n_alleles) {
    double c = 0;
    for (uint32_t ci = 0; ci + 1 < n_colors; ++ci)
        for (uint32_t cj = ci + 1; cj < n_colors; ++cj) {
            const double v = cramer_v_dev(val, ci, cj, n_alleles);
            c = c < v ? v : c;
        }
    return c;
}

template <bool W>
struct Row {  // one output stream position: a pointer when writing, a byte count when measuring
    char *p;
    uint32_t n;
    __device__ inline void put(char c) { if (W) *p++ = c; else ++n; }
};

// two streams that receive the same characters (a frequency row goes to its arity's file and to allele_frequency.txt): formatted
// once, stored twice -- copying the first stream's bytes back out of memory made every character wait for the store before it
template <bool W>
struct Tee {
    Row<W> &a, &b;
    __device__ inline void put(char c) { a.put(c); b.put(c); }
};

__global__ void k_call_has(const pf_bubble_result *__restrict__ res, uint32_t nb, uint32_t *__restrict__ has) {
    const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j < nb) has[j] = (res[j].n_rows != 0 && res[j].n_rows != 0xFFFFFFFFu) ? 1u : 0u;
}

// The write pass stages the four large streams in LDS: consecutive lanes hold consecutive bubbles, so a wavefront's text in a
// stream is one contiguous span of the output; it is formatted into LDS and copied out by consecutive lanes (whole 64-byte
// segments per store).  A span that does not fit its stage (long rows) is written directly, byte by byte, as before.
constexpr uint32_t FMT_BLOCK = 128;
constexpr uint32_t FMT_STAGE[4] = {12288, 2048, 2048, 3072};   // alignseq, allele_frequency, bifre, bicov
constexpr int FMT_STAGED_STREAM[4] = {1, 0, 2, 6};

template <bool W, bool COLORED>
__global__ __launch_bounds__(FMT_BLOCK) void k_call_format(FmtArgs a) {
    const uint32_t jj = blockIdx.x * blockDim.x + threadIdx.x;   // index inside the text batch (sizes / offsets)
    const uint32_t j = a.j0 + jj;                                // ... inside the aligned batch (results, numbering, site values)
    unsigned long long allele[4] = {0, 0, 0, 0}, core_cov = 0, core_num = 0;
    __shared__ __attribute__((aligned(16))) char s_stage[W ? (FMT_BLOCK / 64) * (12288 + 2048 + 2048 + 3072) : 4];
    bool staged[4] = {false, false, false, false};
    uint64_t span0[4] = {0, 0, 0, 0};
    uint32_t span_len[4] = {0, 0, 0, 0};
    char *stage[4] = {nullptr, nullptr, nullptr, nullptr};
    char *cp_dst = nullptr;          // this lane's bubble: where row 0's characters go, where the rows lie, their length and number,
    const char *cp_src = nullptr;    // and the distance from one row's characters to the next row's in the output
    uint32_t cp_L = 0, cp_R = 0, cp_step = 0;
    if (W) {
        const size_t stride = (size_t)a.nb + 1;
        const uint32_t w_first = jj & ~63u;
        if (w_first < a.nb) {   // (wave-uniform)
            const uint32_t w_end = w_first + 64 < a.nb ? w_first + 64 : a.nb;
            char *base = s_stage + (threadIdx.x >> 6) * (12288 + 2048 + 2048 + 3072);
            uint32_t acc = 0;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                // (alignseq's stage holds the wavefront's packed records instead of its text when the stream leaves packed)
                const int st_ = (q == 0 && a.packed) ? S_PACK : FMT_STAGED_STREAM[q];
                span0[q] = a.offs[st_ * stride + w_first];
                span_len[q] = (uint32_t)(a.offs[st_ * stride + w_end] - span0[q]);
                staged[q] = span_len[q] <= FMT_STAGE[q];
                stage[q] = base + acc;
                acc += FMT_STAGE[q];
            }
        }
    }
    if (jj < a.nb) {
        const size_t stride = (size_t)a.nb + 1;
        Row<W> s_all{nullptr, 0}, s_aln{nullptr, 0}, s_fre[4], s_cov[4];
        for (int x = 0; x < 4; ++x) { s_fre[x] = Row<W>{nullptr, 0}; s_cov[x] = Row<W>{nullptr, 0}; }
        if (W) {
            s_all.p = a.out[0] + (a.offs[0 * stride + jj] - a.offs[0 * stride]);
            if (!a.packed) s_aln.p = a.out[1] + (a.offs[1 * stride + jj] - a.offs[1 * stride]);
#pragma unroll
            for (int x = 0; x < 4; ++x) {
                s_fre[x].p = a.out[2 + x] + (a.offs[(2 + x) * stride + jj] - a.offs[(2 + x) * stride]);
                s_cov[x].p = a.out[6 + x] + (a.offs[(6 + x) * stride + jj] - a.offs[(6 + x) * stride]);
            }
            if (staged[0] && !a.packed) s_aln.p = stage[0] + (a.offs[1 * stride + jj] - span0[0]);
            if (staged[1]) s_all.p = stage[1] + (a.offs[0 * stride + jj] - span0[1]);
            if (staged[2]) s_fre[0].p = stage[2] + (a.offs[2 * stride + jj] - span0[2]);
            if (staged[3]) s_cov[0].p = stage[3] + (a.offs[6 * stride + jj] - span0[3]);
        }
        const pf_bubble_result r = a.res[j];
        if (r.n_rows != 0 && r.n_rows != 0xFFFFFFFFu) {
            const CallTask &t = a.ct[a.kept[a.t0 + j]];
            const uint32_t R = r.n_rows, L = r.n_cols;
            const char *rows = a.otext + r.rows_off;
            const uint64_t my_vc = a.vc_base + a.vc[j] - (a.mt ? 1 : 0);   // fetch_add(1) returns the old value (src/CDBG.cpp:2056)
            char *fre_start[4] = {s_fre[0].p, s_fre[1].p, s_fre[2].p, s_fre[3].p};
            const uint32_t fre_n0[4] = {s_fre[0].n, s_fre[1].n, s_fre[2].n, s_fre[3].n};
            if (W && a.packed) {
                // alignseq, packed: this lane writes its bubble's header; the rows are packed by the whole wavefront further down
                char *rec = staged[0] ? stage[0] + (a.offs[(size_t)S_PACK * stride + jj] - span0[0])
                                      : a.out[S_PACK] + alnpack_index_bytes(a.nb) + (a.offs[(size_t)S_PACK * stride + jj] - a.offs[(size_t)S_PACK * stride]);
                const uint64_t vc64 = my_vc;
                const uint32_t h[4] = {t.u + 1, (t.exit_ov >> 1) + 1, L, R | (t.strict ? 0x80000000u : 0u)};
                __builtin_memcpy(rec, &vc64, 8);
                __builtin_memcpy(rec + 8, h, 16);
                cp_dst = rec + ALNPACK_HEADER; cp_src = rows; cp_L = L; cp_R = R;
            }
            // alignseq: var_count, strict flag, entrance id, exit id, aligned row (src/CDBG.cpp:1259, 1428)
            for (uint32_t p = 0; p < R && !(W && a.packed); ++p) {
                char *const row_start = s_aln.p;
                put_uint(s_aln, my_vc);
                s_aln.put('\t'); s_aln.put(t.strict ? '1' : '0'); s_aln.put('\t');
                put_uint(s_aln, (uint64_t)t.u + 1);
                s_aln.put('\t');
                put_uint(s_aln, (uint64_t)(t.exit_ov >> 1) + 1);
                s_aln.put('\t');
                // the aligned row itself is copied by the whole wavefront further down (every row of a bubble has the same prefix, so
                // row p's characters start p * (prefix + L + 1) behind row 0's): a lane copying its own rows byte by byte touches 64
                // different lines per load instruction -- the copies were what the write pass waited for
                if (W) { if (p == 0) { cp_dst = s_aln.p; cp_src = rows; cp_L = L; cp_R = R; } s_aln.p += L; }
                else s_aln.n += L;
                s_aln.put('\n');
                if (W && p == 0) cp_step = (uint32_t)(s_aln.p - row_start);
            }
            core_cov = (unsigned long long)t.core_mean;
            core_num = 1;
            const pf_bubble_site *sites = a.osites + r.site_off;
            const uint32_t *ilen = a.oilen + r.ilen_off;
            const size_t usize = a.len[t.u], esize = a.len[t.exit_ov >> 1];
            const uint32_t ns = r.n_sites;
            uint32_t indel = 0;
            uint64_t vcur = t.strict ? 0 : a.sv_off[j];
            for (uint32_t i = 0; i < ns; ++i) {
                const pf_bubble_site sr = sites[i];
                const uint8_t *grp = a.ogroups + r.group_off + (uint64_t)i * R;
                // distance to the neighbouring sites / unitig ends (src/CDBG.cpp:1279-1298)
                uint32_t vd;
                if (i == 0) {
                    if (ns != 1) vd = (uint32_t)std::min((size_t)(uint32_t)(sites[1].col - sites[0].col - 1), usize);
                    else vd = (uint32_t)std::min(usize, esize);
                } else if (i == ns - 1) {
                    vd = (uint32_t)std::min((size_t)(uint32_t)(sites[i].col - sites[i - 1].col - 1), esize);
                } else {
                    vd = std::min((uint32_t)(sites[i].col - sites[i - 1].col - 1), (uint32_t)(sites[i + 1].col - sites[i].col - 1));
                }
                const uint32_t maxnum = sr.maxnum;
                if (sr.is_indel) ++indel;  // counted even when the site is dropped below (src/CDBG.cpp:1526)
                if (COLORED) {
                    const uint32_t C = a.n_colors;
                    const double *cvals = nullptr;
                    if (!t.strict) {
                        cvals = a.sv + vcur;
                        vcur += (uint64_t)C * maxnum + 1;
                        if (!sr.pad_) continue;   // a string failed a colour's range test, or some colour covers no string (:3292-3300)
                    }
                    // strict: the [colour][path] matrix of the scan, again from K-COV-C's results (an entry = the mean coverage of the
                    // path's unitig in a colour that has it in full, else 0), paths as sorted there
                    uint64_t fm[4] = {0, 0, 0, 0};
                    uint32_t wu[4] = {0, 0, 0, 0}, lk[4] = {1, 1, 1, 1};
                    if (t.strict)
                        for (uint32_t p = 0; p < R && p < 4; ++p) {
                            wu[p] = t.inner[p] >> 1;
                            fm[p] = a.full[wu[p]];
                            lk[p] = a.len[wu[p]] - (uint32_t)a.k + 1;
                        }
                    auto m_at = [&](uint32_t c, uint32_t p) -> double {
                        const uint32_t q = p < 4 ? p : 3;
                        return ((fm[q] >> c) & 1) ? (double)a.ccov_sum[(size_t)c * a.N + wu[q]] / (double)lk[q] : 0.0;
                    };
                    auto gc_at = [&](uint32_t c, uint32_t x) -> double {   // coverage of allele group x in colour c
                        if (!t.strict) return cvals[(size_t)c * maxnum + x];
                        double tc = 0.0;
                        for (uint32_t p = 0; p < R; ++p)
                            if ((uint32_t)grp[p] - 1 == x) tc += m_at(c, p);
                        return tc;
                    };
                    const double coefficient = t.strict ? max_cramer_v_dev(m_at, C, R) : max_cramer_v_dev(gc_at, C, maxnum);
                    for (uint32_t c = 0; c < C; ++c) {
                        uint32_t n_res = 0;
                        double sum = 0;
                        for (uint32_t x = 0; x < maxnum; ++x) {
                            const double v = gc_at(c, x);
                            if (v > 0.0) { ++n_res; sum += v; }
                        }
                        if (n_res < 2) continue;
                        const int ar = (int)n_res - 2;
                        Row<W> cov = ar == 0 ? s_cov[0] : ar == 1 ? s_cov[1] : ar == 2 ? s_cov[2] : s_cov[3];
                        Row<W> fre = ar == 0 ? s_fre[0] : ar == 1 ? s_fre[1] : ar == 2 ? s_fre[2] : s_fre[3];
                        const bool filed = ar <= 3;
                        for (uint32_t x = 0; x < maxnum; ++x) {
                            const double v = gc_at(c, x);
                            if (!(v > 0.0)) continue;
                            if (filed) { put_double(cov, v); cov.put('\t'); }
                            const double fr = v / sum;
                            if (filed) {
                                Tee<W> both{fre, s_all};
                                put_double(both, fr);
                                both.put('\n');
                            } else {
                                put_double(s_all, fr);
                                s_all.put('\n');
                            }
                        }
                        if (filed) {
                            put_uint(cov, c);
                            cov.put('\t');
                            cov.put(t.strict ? '1' : '0'); cov.put('\t');
                            if (sr.is_indel) put_uint(cov, open_run_len(ilen, indel - 1, r.n_indel_len, r.n_cols, sr.col));
                            else cov.put('0');
                            cov.put('\t');
                            put_uint(cov, my_vc);
                            cov.put('\t');
                            put_uint(cov, ns);
                            cov.put('\t');
                            put_double(cov, coefficient);
                            cov.put('\t');
                            put_uint(cov, vd);
                            cov.put('\t'); cov.put('\n');
                            ++allele[ar];
                            if (ar == 0) { s_cov[0] = cov; s_fre[0] = fre; }
                            else if (ar == 1) { s_cov[1] = cov; s_fre[1] = fre; }
                            else if (ar == 2) { s_cov[2] = cov; s_fre[2] = fre; }
                            else { s_cov[3] = cov; s_fre[3] = fre; }
                        }
                    }
                    continue;
                }
                double denom;
                const double *vals = nullptr;
                if (t.strict) {
                    denom = t.cov_sum;
                } else {
                    vals = a.sv + vcur;
                    vcur += maxnum + 1;
                    if (!sr.pad_) continue;
                    denom = vals[maxnum];
                }
                const int ar = (int)maxnum - 2;  // file of this arity, if 0..3
                Row<W> cov = ar == 0 ? s_cov[0] : ar == 1 ? s_cov[1] : ar == 2 ? s_cov[2] : s_cov[3];
                Row<W> fre = ar == 0 ? s_fre[0] : ar == 1 ? s_fre[1] : ar == 2 ? s_fre[2] : s_fre[3];
                const bool filed = ar >= 0 && ar <= 3;
                for (uint32_t x = 0; x < maxnum; ++x) {
                    double tc;
                    if (t.strict) {
                        tc = 0.0;
                        for (uint32_t p = 0; p < R; ++p)
                            if ((uint32_t)grp[p] - 1 == x) tc += t.cov[p < 4 ? p : 3];
                    } else {
                        tc = vals[x];
                    }
                    if (filed) { put_double(cov, tc); cov.put('\t'); }
                    // the frequency row: the arity's fre file (2..5 alleles) and allele_frequency.txt -- there in site order, or,
                    // in the -t > 1 format, grouped by arity at the end of the bubble (src/CDBG.cpp:2158-2162)
                    const double fr = tc / denom;
                    if (filed) {
                        if (!a.mt) {
                            Tee<W> both{fre, s_all};
                            put_double(both, fr);
                            both.put('\n');
                        } else {
                            put_double(fre, fr);
                            fre.put('\n');
                        }
                    } else if (!a.mt) {
                        put_double(s_all, fr);
                        s_all.put('\n');
                    }
                }
                if (filed) {
                    cov.put(t.strict ? '1' : '0'); cov.put('\t');
                    if (sr.is_indel) put_uint(cov, open_run_len(ilen, indel - 1, r.n_indel_len, r.n_cols, sr.col));
                    else cov.put('0');
                    cov.put('\t');
                    put_uint(cov, my_vc);
                    cov.put('\t');
                    put_uint(cov, ns);
                    cov.put('\t');
                    put_uint(cov, vd);
                    cov.put('\t'); cov.put('\n');
                    ++allele[ar];
                    if (ar == 0) { s_cov[0] = cov; s_fre[0] = fre; }
                    else if (ar == 1) { s_cov[1] = cov; s_fre[1] = fre; }
                    else if (ar == 2) { s_cov[2] = cov; s_fre[2] = fre; }
                    else { s_cov[3] = cov; s_fre[3] = fre; }
                }
            }
            if (a.mt) {
                // allfre << bifre_info << trifre_info << tetrafre_info (<< pentafre_info only in the strict branch, :2162 vs :2550)
                const int n_ar = t.strict ? 4 : 3;
                for (int x = 0; x < n_ar; ++x) {
                    if (W) { for (char *c = fre_start[x]; c < s_fre[x].p; ++c) *s_all.p++ = *c; }
                    else s_all.n += s_fre[x].n - fre_n0[x];
                }
            }
        }
        if (!W) {
            a.sizes[0 * stride + jj] = s_all.n;
            a.sizes[1 * stride + jj] = s_aln.n;
            a.sizes[(size_t)S_PACK * stride + jj] = (a.packed && s_aln.n) ? ALNPACK_HEADER + r.n_rows * alnpack_row_bytes(r.n_cols) : 0u;
#pragma unroll
            for (int x = 0; x < 4; ++x) {
                a.sizes[(2 + x) * stride + jj] = s_fre[x].n;
                a.sizes[(6 + x) * stride + jj] = s_cov[x].n;
            }
        }
    }
    if (W) {
        {   // the aligned rows, bubble after bubble, by all lanes
            unsigned long long todo = __ballot(cp_R != 0);
            const int lane = lane_id();
            while (todo) {
                const int b = __ffsll((long long)todo) - 1;
                todo &= todo - 1;
                const uint32_t Lb = read_lane(cp_L, b), Rb = read_lane(cp_R, b), step = read_lane(cp_step, b);
                const uint64_t d64 = (uint64_t)(uintptr_t)cp_dst, s64 = (uint64_t)(uintptr_t)cp_src;
                char *dst = reinterpret_cast<char *>((uintptr_t)(((uint64_t)read_lane((uint32_t)(d64 >> 32), b) << 32) | read_lane((uint32_t)d64, b)));
                const char *src = reinterpret_cast<const char *>((uintptr_t)(((uint64_t)read_lane((uint32_t)(s64 >> 32), b) << 32) | read_lane((uint32_t)s64, b)));
                if (a.packed) {
                    // eight characters = three bytes (pf_alnpack.hpp).  A lane per character: consecutive lanes read consecutive bytes
                    // of the row, the eight lanes of a group OR their 3-bit codes together (three exchanges), and the first three
                    // lanes of the group store one byte each -- into the LDS stage when the wavefront's records fit it
                    const uint32_t rb = alnpack_row_bytes(Lb);
                    for (uint32_t p = 0; p < Rb; ++p)
                        for (uint32_t x0 = 0; x0 < Lb; x0 += WAVE) {
                            const uint32_t x = x0 + (uint32_t)lane;
                            uint32_t v = x < Lb ? alnpack_code(src[(size_t)p * Lb + x]) << (3 * (lane & 7)) : 0u;
                            v |= (uint32_t)__shfl_xor((int)v, 1, WAVE);
                            v |= (uint32_t)__shfl_xor((int)v, 2, WAVE);
                            v |= (uint32_t)__shfl_xor((int)v, 4, WAVE);
                            const uint32_t g = x >> 3, byte = (uint32_t)lane & 7;
                            if (byte < 3 && (g << 3) < Lb) dst[(size_t)p * rb + 3 * (size_t)g + byte] = (char)(v >> (8 * byte));
                        }
                    continue;
                }
                for (uint32_t p = 0; p < Rb; ++p)
                    for (uint32_t x = (uint32_t)lane; x < Lb; x += WAVE) dst[(size_t)p * step + x] = src[(size_t)p * Lb + x];
            }
        }
        if (a.packed && jj < a.nb && (jj % ALNPACK_GROUP == 0 || jj + 1 == a.nb)) {
            // the index of the piece: where the text and the records of every 256th bubble begin, and where both end
            const size_t stride = (size_t)a.nb + 1;
            char *idx = a.out[S_PACK];
            auto entry = [&](uint64_t g, uint32_t at) {
                const uint64_t e[2] = {a.offs[1 * stride + at] - a.offs[1 * stride], a.offs[(size_t)S_PACK * stride + at] - a.offs[(size_t)S_PACK * stride]};
                __builtin_memcpy(idx + 16 + 16 * g, e, 16);
            };
            if (jj % ALNPACK_GROUP == 0) entry(jj / ALNPACK_GROUP, jj);
            if (jj + 1 == a.nb) {
                const uint64_t n_groups = ((uint64_t)a.nb + ALNPACK_GROUP - 1) / ALNPACK_GROUP;
                const uint64_t head[2] = {n_groups, ALNPACK_GROUP};
                __builtin_memcpy(idx, head, 16);
                entry(n_groups, a.nb);
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
        __builtin_amdgcn_wave_barrier();
        const size_t stride = (size_t)a.nb + 1;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            if (!staged[q]) continue;
            const bool pk = q == 0 && a.packed;
            const int st_ = pk ? S_PACK : FMT_STAGED_STREAM[q];
            char *dst = a.out[st_] + (pk ? alnpack_index_bytes(a.nb) : 0) + (span0[q] - a.offs[st_ * stride]);
            // four bytes per lane and step (the stage is word-aligned in LDS; the span lies where it lies in the stream: global
            // memory takes the unaligned word), the last one to three bytes singly
            const uint32_t n_words = span_len[q] >> 2;
            for (uint32_t x = lane_id(); x < n_words; x += WAVE) {
                const uint32_t w = reinterpret_cast<const uint32_t *>(stage[q])[x];
                __builtin_memcpy(dst + 4 * (size_t)x, &w, 4);
            }
            for (uint32_t x = (n_words << 2) + lane_id(); x < span_len[q]; x += WAVE) dst[x] = stage[q][x];
        }
    }
    if (!W) {
        // counters: one atomic per wave and counter
        unsigned long long v[7] = {allele[0], allele[1], allele[2], allele[3], core_cov, core_num, 0};
        for (int x = 0; x < 6; ++x) {
            const unsigned long long s = wave_sum_u64(v[x]);
            // (allele[4], core_cov, core_num lie one after the other: the host zeroes them as six words as well)
            if (lane_id() == 0 && s) atomicAdd(reinterpret_cast<unsigned long long *>(reinterpret_cast<char *>(a.cnt) + offsetof(CallCounters, allele)) + x, s);
        }
        if (jj == a.nb) {
            const size_t stride = (size_t)a.nb + 1;
            for (int s = 0; s < N_INT; ++s) a.sizes[s * stride + a.nb] = 0;
        }
    }
}

__global__ void k_call_totals(const uint64_t *__restrict__ offs, const uint32_t *__restrict__ sizes, uint32_t nb, uint64_t *__restrict__ totals) {
    const int s = threadIdx.x;
    const size_t stride = (size_t)nb + 1;
    if (s < N_INT) totals[s] = offs[s * stride + nb] - offs[s * stride];
    (void)sizes;
}

// ---------------------------------------------------------------------------------------------------------------------
// O1, first file: the rows of <outpre>_super_bubble.txt (src/CDBG.cpp:222-252; colored rule src/CCDBG.cpp:2106-2132) from the
// state on the device: one row per open endpoint side in unitig order, numbered from 1.
struct SbArgs {
    const uint8_t *flags;
    const uint32_t *plus, *minus;
    uint32_t N;
    int colored;
    uint32_t first_id;          // 1; 0 in the reference's -t > 1 format (fetch_add, src/CDBG.cpp:1829)
    const uint32_t *row_base;   // exclusive scan of rows per unitig
    uint32_t *sizes;            // bytes per unitig (N + 1 entries, the last 0)
    const uint64_t *offs;
    char *out;
};

__device__ inline uint32_t sb_rows_of(const SbArgs &a, uint32_t u, bool &p_row, bool &m_row) {
    const uint8_t f = a.flags[u];
    p_row = m_row = false;
    if ((f & 3) == 0) return 0;
    if (a.colored) { p_row = a.plus[u] != 0; m_row = a.minus[u] != 0; }   // an open unitig lists every side whose partner is set, self included
    else { p_row = (f & B_PLUS) != 0; m_row = (f & B_MINUS) != 0; }
    return (uint32_t)p_row + (uint32_t)m_row;
}

__global__ void k_sb_count(SbArgs a, uint32_t *cnt) {
    const uint32_t u = blockIdx.x * blockDim.x + threadIdx.x;
    if (u > a.N) return;
    bool p, m;
    cnt[u] = u < a.N ? sb_rows_of(a, u, p, m) : 0u;
}

constexpr uint32_t SB_STAGE = 4096;   // bytes of LDS per wavefront for its rows (64 unitigs, at most two rows of < 50 bytes each)

template <bool W>
__global__ __launch_bounds__(256) void k_sb_format(SbArgs a) {
    const uint32_t u = blockIdx.x * blockDim.x + threadIdx.x;
    // write pass: a wavefront's rows are one contiguous span of the file; staged in LDS and copied out by consecutive lanes
    __shared__ __attribute__((aligned(16))) char s_stage[W ? 4 * SB_STAGE : 4];
    char *stage = s_stage + (threadIdx.x >> 6) * SB_STAGE;
    uint64_t span0 = 0;
    uint32_t span_len = 0;
    bool staged = false;
    if (W) {
        const uint32_t w_first = u & ~63u;
        if (w_first < a.N) {
            const uint32_t w_end = w_first + 64 < a.N ? w_first + 64 : a.N;
            span0 = a.offs[w_first];
            span_len = (uint32_t)(a.offs[w_end] - span0);
            staged = span_len <= SB_STAGE;
        }
    }
    if (u < a.N) {
        bool rows[2];
        const uint32_t n = sb_rows_of(a, u, rows[0], rows[1]);
        Row<W> o{W ? (staged ? stage + (a.offs[u] - span0) : a.out + a.offs[u]) : nullptr, 0};
        if (n) {
            const uint8_t f = a.flags[u];
            uint64_t nb = (uint64_t)a.row_base[u] + a.first_id;
            for (int side = 0; side < 2; ++side) {
                if (!rows[side]) continue;
                const bool ps = side == 0;
                put_uint(o, nb++);
                o.put('\t');
                put_uint(o, (uint64_t)u + 1);
                o.put('\t'); o.put(ps ? '+' : '-'); o.put('\t');
                put_uint(o, ps ? a.plus[u] : a.minus[u]);
                o.put('\t'); o.put((f & (ps ? B_STRICT_P : B_STRICT_M)) ? '1' : '0');
                o.put('\t'); o.put((f & (ps ? B_COMPLEX_P : B_COMPLEX_M)) ? '1' : '0');
                o.put('\n');
            }
        }
        if (!W) a.sizes[u] = o.n;
    } else if (u == a.N && !W) {
        a.sizes[u] = 0;
    }
    if (W && staged) {
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
        __builtin_amdgcn_wave_barrier();
        char *dst = a.out + span0;
        const uint32_t n_words = span_len >> 2;   // (in words, as K-TEXT's stages leave)
        for (uint32_t x = lane_id(); x < n_words; x += WAVE) {
            const uint32_t w = reinterpret_cast<const uint32_t *>(stage)[x];
            __builtin_memcpy(dst + 4 * (size_t)x, &w, 4);
        }
        for (uint32_t x = (n_words << 2) + lane_id(); x < span_len; x += WAVE) dst[x] = stage[x];
    }
}

struct Widen {
    __host__ __device__ uint64_t operator()(uint32_t x) const { return (uint64_t)x; }
};

__global__ void k_format_doubles(const double *__restrict__ x, uint64_t n, char *__restrict__ out, uint8_t *__restrict__ len) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    BufSink s{out + i * 32};
    put_double(s, x[i]);
    len[i] = (uint8_t)(s.p - (out + i * 32));
}

}  // namespace

// =====================================================================================================================
// T1 state written on the device (pf_replay_device, pf_cc.hip): the arrays pf_call_set_state would fill, for their writer
namespace pf {
int call_state_arrays(pf_ctx *ctx, uint8_t **flags, uint32_t **plus, uint32_t **minus) {
    if (!ctx->d_seq || !ctx->has_adj) { pf::CtxErr{ctx} = "T1 state: graph and adjacency first"; return PF_ERR_ARG; }
    CallState *S = state_of(ctx);
    const size_t N = ctx->N;
    if (!S->flags.ensure(N + 1) || !S->plus.ensure(N * 4) || !S->minus.ensure(N * 4)) { pf::CtxErr{ctx} = "T1 state: out of device memory"; return PF_ERR_HIP; }
    *flags = S->flags.as<uint8_t>();
    *plus = S->plus.as<uint32_t>();
    *minus = S->minus.as<uint32_t>();
    S->have_state = false;
    return PF_OK;
}
void call_state_resident(pf_ctx *ctx) {
    CallState *S = state_of(ctx);
    S->have_state = true;
    S->n_sides = S->n_tasks = 0;
}
}  // namespace pf

extern "C" {

int pf_format_doubles(pf_ctx *ctx, const double *values, uint64_t n, char *text, uint8_t *len) {
    if (!ctx || (n && (!values || !text || !len))) return PF_ERR_ARG;
    if (n == 0) return PF_OK;
    PF_HIP(hipSetDevice(ctx->device));
    DevTmp<double> dx;
    DevTmp<char> dt;
    DevTmp<uint8_t> dl;
    PF_HIP(dx.alloc(n * 8));
    PF_HIP(dt.alloc(n * 32));
    PF_HIP(dl.alloc(n));
    PF_HIP(hipMemcpyAsync(dx.p, values, n * 8, hipMemcpyDefault, ctx->stream));
    k_format_doubles<<<(unsigned)((n + 255) / 256), 256, 0, ctx->stream>>>(dx.p, n, dt.p, dl.p);
    PF_HIP(hipGetLastError());
    PF_HIP(hipMemcpyAsync(text, dt.p, n * 32, hipMemcpyDefault, ctx->stream));
    PF_HIP(hipMemcpyAsync(len, dl.p, n, hipMemcpyDefault, ctx->stream));
    PF_HIP(hipStreamSynchronize(ctx->stream));
    return PF_OK;
}

int pf_call_set_state(pf_ctx *ctx, const uint8_t *flags, const uint32_t *plus, const uint32_t *minus) {
    if (!ctx || !flags || !plus || !minus) return PF_ERR_ARG;
    if (!ctx->d_seq || !ctx->has_adj) { pf::CtxErr{ctx} = "pf_call_set_state: graph and adjacency first"; return PF_ERR_ARG; }
    PF_HIP(hipSetDevice(ctx->device));
    CallState *S = state_of(ctx);
    const size_t N = ctx->N;
    if (!S->flags.ensure(N + 1) || !S->plus.ensure(N * 4) || !S->minus.ensure(N * 4)) { pf::CtxErr{ctx} = "pf_call_set_state: out of device memory"; return PF_ERR_HIP; }
    PF_HIP(hipMemcpyAsync(S->flags.p, flags, N, hipMemcpyDefault, ctx->stream));
    PF_HIP(hipMemcpyAsync(S->plus.p, plus, N * 4, hipMemcpyDefault, ctx->stream));
    PF_HIP(hipMemcpyAsync(S->minus.p, minus, N * 4, hipMemcpyDefault, ctx->stream));
    PF_HIP(hipStreamSynchronize(ctx->stream));
    S->have_state = true;
    S->n_sides = S->n_tasks = 0;
    return PF_OK;
}

int pf_call_get_state(pf_ctx *ctx, uint8_t *flags, uint32_t *plus, uint32_t *minus) {
    if (!ctx || !ctx->call) return PF_ERR_ARG;
    CallState *S = ctx->call;
    if (!S->have_state) { pf::CtxErr{ctx} = "pf_call_get_state: no T1 state on the device"; return PF_ERR_ARG; }
    PF_HIP(hipSetDevice(ctx->device));
    const size_t N = ctx->N;
    PF_HIP(hipStreamSynchronize(ctx->stream));
    if (flags) PF_HIP(hipMemcpy(flags, S->flags.p, N, hipMemcpyDeviceToHost));
    if (plus) PF_HIP(hipMemcpy(plus, S->plus.p, N * 4, hipMemcpyDeviceToHost));
    if (minus) PF_HIP(hipMemcpy(minus, S->minus.p, N * 4, hipMemcpyDeviceToHost));
    return PF_OK;
}

int pf_call_set_format(pf_ctx *ctx, int reference_mt) {
    if (!ctx) return PF_ERR_ARG;
    state_of(ctx)->mt_format = reference_mt != 0;
    return PF_OK;
}

int pf_superbubble_rows(pf_ctx *ctx, int colored_rule, uint64_t *n_rows, uint64_t *text_len) {
    if (!ctx || !n_rows || !text_len) return PF_ERR_ARG;
    CallState *S = ctx->call;
    if (!S || !S->have_state) { pf::CtxErr{ctx} = "pf_superbubble_rows: pf_call_set_state first"; return PF_ERR_ARG; }
    PF_HIP(hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    const uint32_t N = ctx->N;
    const size_t n1 = (size_t)N + 1;
    if (!S->sb_cnt.ensure(n1 * 4) || !S->sb_base.ensure(n1 * 4) || !S->sb_sizes.ensure(n1 * 4) || !S->sb_offs.ensure(n1 * 8)) {
        pf::CtxErr{ctx} = "pf_superbubble_rows: out of device memory";
        return PF_ERR_HIP;
    }
    SbArgs a;
    a.flags = S->flags.as<uint8_t>(); a.plus = S->plus.as<uint32_t>(); a.minus = S->minus.as<uint32_t>(); a.N = N; a.colored = colored_rule;
    a.first_id = S->mt_format ? 0u : 1u;
    a.row_base = S->sb_base.as<uint32_t>(); a.sizes = S->sb_sizes.as<uint32_t>(); a.offs = S->sb_offs.as<uint64_t>(); a.out = nullptr;
    const unsigned grid = (unsigned)((n1 + 255) / 256);
    k_sb_count<<<grid, 256, 0, st>>>(a, S->sb_cnt.as<uint32_t>());
    size_t t1 = 0, t2 = 0;
    hipcub::TransformInputIterator<uint64_t, Widen, const uint32_t *> wide(S->sb_sizes.as<uint32_t>(), Widen());
    PF_HIP(hipcub::DeviceScan::ExclusiveSum(nullptr, t1, S->sb_cnt.as<uint32_t>(), S->sb_base.as<uint32_t>(), (int)n1, st));
    PF_HIP(hipcub::DeviceScan::ExclusiveSum(nullptr, t2, wide, S->sb_offs.as<uint64_t>(), (int)n1, st));
    if (!S->scan_tmp.ensure(std::max(t1, t2))) { pf::CtxErr{ctx} = "pf_superbubble_rows: out of device memory"; return PF_ERR_HIP; }
    PF_HIP(hipcub::DeviceScan::ExclusiveSum(S->scan_tmp.p, t1, S->sb_cnt.as<uint32_t>(), S->sb_base.as<uint32_t>(), (int)n1, st));
    ctx_begin(ctx, PF_K_CALL_FORMAT);
    k_sb_format<false><<<grid, 256, 0, st>>>(a);
    ctx_end(ctx);
    PF_HIP(hipcub::DeviceScan::ExclusiveSum(S->scan_tmp.p, t2, wide, S->sb_offs.as<uint64_t>(), (int)n1, st));
    uint32_t rows = 0;
    uint64_t len = 0;
    PF_HIP(hipMemcpyAsync(&rows, S->sb_base.as<uint32_t>() + N, 4, hipMemcpyDeviceToHost, st));
    PF_HIP(hipMemcpyAsync(&len, S->sb_offs.as<uint64_t>() + N, 8, hipMemcpyDeviceToHost, st));
    PF_HIP(hipStreamSynchronize(st));
    if (!S->sb_out.ensure(std::max<uint64_t>(len, 16))) { pf::CtxErr{ctx} = "pf_superbubble_rows: out of device memory"; return PF_ERR_HIP; }
    a.out = S->sb_out.as<char>();
    // the write pass runs on the copy stream, in front of pf_superbubble_fetch's copy and beside whatever the caller launches next
    // (PloidyEstimation's coverage and scan only read the state): the caller has the counts and does not wait for the text
    hipStream_t wst = S->copy_stream ? S->copy_stream : st;
    ctx_begin_on(ctx, PF_K_CALL_FORMAT, wst);
    k_sb_format<true><<<grid, 256, 0, wst>>>(a);
    ctx_end_on(ctx, wst);
    PF_HIP(hipGetLastError());
    if (wst == st) PF_HIP(hipStreamSynchronize(st));
    S->sb_len = len;
    *n_rows = rows;
    *text_len = len;
    return PF_OK;
}

int pf_superbubble_fetch(pf_ctx *ctx, char *dst, uint64_t len) {
    if (!ctx || !ctx->call || (len && !dst)) return PF_ERR_ARG;
    CallState *S = ctx->call;
    if (len > S->sb_len) return PF_ERR_ARG;
    if (len == 0) return PF_OK;
    if (hipSetDevice(ctx->device) != hipSuccess || !S->copy_stream) return PF_ERR_HIP;
    size_t tl_at = (size_t)-1;
    (void)ctx_begin_at(ctx, PF_K_COPY_TEXT, S->copy_stream, &tl_at);
    if (hipMemcpyAsync(dst, S->sb_out.p, (size_t)len, hipMemcpyDeviceToHost, S->copy_stream) != hipSuccess) return PF_ERR_HIP;
    ctx_end_at(ctx, tl_at, S->copy_stream);
    if (hipStreamSynchronize(S->copy_stream) != hipSuccess) return PF_ERR_HIP;
    return PF_OK;
}

int pf_call_set_colours(pf_ctx *ctx, uint32_t n_colors, const uint64_t *full_mask, const uint64_t *size_total, const uint32_t *part_first,
                        const uint32_t *part_colour, const uint64_t *part_word, const uint64_t *part_bits, uint64_t n_part, uint64_t n_words) {
    if (!ctx || !ctx->d_seq) { if (ctx) pf::CtxErr{ctx} = "pf_call_set_colours: graph first"; return PF_ERR_ARG; }
    PF_HIP(hipSetDevice(ctx->device));
    CallState *S = state_of(ctx);
    S->n_colors = 0;
    S->have_cov = false;
    if (!n_colors) return PF_OK;
    if (n_colors > PF_MAX_COLORS || n_colors != ctx->n_colors || !full_mask || !size_total || !part_first || (n_part && (!part_colour || !part_word)) ||
        (n_words && !part_bits)) {
        pf::CtxErr{ctx} = "pf_call_set_colours: the colour sets of every unitig, for the colours of the uploaded databases, are required";
        return PF_ERR_ARG;
    }
    const size_t N = ctx->N;
    if (!S->col_low.ensure((size_t)n_colors * 4) || !S->col_up.ensure((size_t)n_colors * 4) || !S->col_full.ensure(N * 8) || !S->col_size.ensure(N * 8) ||
        !S->part_first.ensure((N + 1) * 4) || !S->part_colour.ensure((n_part + 1) * 4) || !S->part_word.ensure((n_part + 1) * 8) ||
        !S->part_bits.ensure((n_words + 1) * 8)) {
        pf::CtxErr{ctx} = "pf_call_set_colours: out of device memory";
        return PF_ERR_HIP;
    }
    hipStream_t st = ctx->stream;
    PF_HIP(hipMemsetAsync(S->col_low.p, 0, (size_t)n_colors * 4, st));
    PF_HIP(hipMemsetAsync(S->col_up.p, 0xFF, (size_t)n_colors * 4, st));
    PF_HIP(hipMemcpyAsync(S->col_full.p, full_mask, N * 8, hipMemcpyDefault, st));
    PF_HIP(hipMemcpyAsync(S->col_size.p, size_total, N * 8, hipMemcpyDefault, st));
    PF_HIP(hipMemcpyAsync(S->part_first.p, part_first, (N + 1) * 4, hipMemcpyDefault, st));
    if (n_part) {
        PF_HIP(hipMemcpyAsync(S->part_colour.p, part_colour, n_part * 4, hipMemcpyDefault, st));
        PF_HIP(hipMemcpyAsync(S->part_word.p, part_word, n_part * 8, hipMemcpyDefault, st));
    }
    if (n_words) PF_HIP(hipMemcpyAsync(S->part_bits.p, part_bits, n_words * 8, hipMemcpyDefault, st));
    PF_HIP(hipStreamSynchronize(st));
    S->n_colors = n_colors;
    return PF_OK;
}

int pf_call_set_cutoffs(pf_ctx *ctx, uint32_t n_colors, const uint32_t *lower, const uint32_t *upper) {
    if (!ctx || !ctx->call || !ctx->call->n_colors || n_colors != ctx->call->n_colors || !lower || !upper) {
        if (ctx) pf::CtxErr{ctx} = "pf_call_set_cutoffs: pf_call_set_colours first; one (lower, upper) per colour";
        return PF_ERR_ARG;
    }
    PF_HIP(hipSetDevice(ctx->device));
    CallState *S = ctx->call;
    PF_HIP(hipMemcpyAsync(S->col_low.p, lower, (size_t)n_colors * 4, hipMemcpyDefault, ctx->stream));
    PF_HIP(hipMemcpyAsync(S->col_up.p, upper, (size_t)n_colors * 4, hipMemcpyDefault, ctx->stream));
    PF_HIP(hipStreamSynchronize(ctx->stream));
    return PF_OK;
}

int pf_call_coverage(pf_ctx *ctx) {
    if (ctx && ctx->call && ctx->call->n_colors) {
        // colored: CCDBG::readCovUni for every (colour, unitig), left where the scan, K-SITES and K-TEXT read it
        if (!ctx->d_seq || !ctx->d_ctab) { pf::CtxErr{ctx} = "pf_call_coverage: graph and count tables first"; return PF_ERR_ARG; }
        PF_HIP(hipSetDevice(ctx->device));
        CallState *S = ctx->call;
        const size_t n = (size_t)ctx->N * S->n_colors;
        if (!S->ccov_sum.ensure(n * 8) || !S->ccov_min.ensure(n * 4) || !S->ccov_max.ensure(n * 4) || !S->ccov_miss.ensure(n)) {
            pf::CtxErr{ctx} = "pf_call_coverage: out of device memory";
            return PF_ERR_HIP;
        }
        const int st = pf_unitig_cov_colored(ctx, 0, ctx->N, S->ccov_sum.as<uint64_t>(), S->ccov_min.as<uint32_t>(), S->ccov_max.as<uint32_t>(), S->ccov_miss.as<uint8_t>());
        if (st != PF_OK) return st;
        S->per_strand = false;
        S->have_cov = true;
        return PF_OK;
    }
    if (!ctx || !ctx->d_seq || !ctx->d_tab) { if (ctx) pf::CtxErr{ctx} = "pf_call_coverage: graph and count table first"; return PF_ERR_ARG; }
    PF_HIP(hipSetDevice(ctx->device));
    CallState *S = state_of(ctx);
    const size_t N = ctx->N;
    const int slots = ctx->tab_exact ? 2 : 1;
    if (!S->cov_sum.ensure(N * slots * 8) || !S->cov_min.ensure(N * slots * 4) || !S->cov_miss.ensure(N * slots)) {
        pf::CtxErr{ctx} = "pf_call_coverage: out of device memory";
        return PF_ERR_HIP;
    }
    int st;
    if (!ctx->tab_exact) {
        st = pf_unitig_cov(ctx, 0, (uint32_t)N, S->cov_sum.as<uint64_t>(), S->cov_min.as<uint32_t>(), S->cov_miss.as<uint8_t>());
    } else {
        st = pf_unitig_cov_exact(ctx, 0, (uint32_t)N, 0, S->cov_sum.as<uint64_t>(), S->cov_min.as<uint32_t>(), S->cov_miss.as<uint8_t>());
        if (st == PF_OK || st == PF_ERR_MISSING_KMER)
            st = pf_unitig_cov_exact(ctx, 0, (uint32_t)N, 1, S->cov_sum.as<uint64_t>() + N, S->cov_min.as<uint32_t>() + N, S->cov_miss.as<uint8_t>() + N);
    }
    if (st != PF_OK && st != PF_ERR_MISSING_KMER) return st;  // a missing k-mer matters only where the driver loop reads it
    S->per_strand = ctx->tab_exact;
    S->have_cov = true;
    return PF_OK;
}

int pf_call_scan(pf_ctx *ctx, uint32_t lower, uint32_t upper, uint64_t *n_sides) {
    if (!ctx || !n_sides) return PF_ERR_ARG;
    CallState *S = ctx->call;
    if (!S || !S->have_state || !S->have_cov) { pf::CtxErr{ctx} = "pf_call_scan: pf_call_set_state and pf_call_coverage first"; return PF_ERR_ARG; }
    PF_HIP(hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    const uint32_t N = ctx->N;
    if (!S->side_cnt.ensure(((size_t)N + 1) * 4) || !S->side_base.ensure(((size_t)N + 1) * 4)) { pf::CtxErr{ctx} = "pf_call_scan: out of device memory"; return PF_ERR_HIP; }
    k_call_count_sides<<<(N + 1 + 255) / 256, 256, 0, st>>>(S->flags.as<uint8_t>(), N, S->side_cnt.as<uint32_t>());
    size_t tmp = 0;
    PF_HIP(hipcub::DeviceScan::ExclusiveSum(nullptr, tmp, S->side_cnt.as<uint32_t>(), S->side_base.as<uint32_t>(), (int)(N + 1), st));
    if (!S->scan_tmp.ensure(tmp)) { pf::CtxErr{ctx} = "pf_call_scan: out of device memory"; return PF_ERR_HIP; }
    PF_HIP(hipcub::DeviceScan::ExclusiveSum(S->scan_tmp.p, tmp, S->side_cnt.as<uint32_t>(), S->side_base.as<uint32_t>(), (int)(N + 1), st));
    uint32_t total = 0;
    PF_HIP(hipMemcpyAsync(&total, S->side_base.as<uint32_t>() + N, 4, hipMemcpyDeviceToHost, st));
    PF_HIP(hipStreamSynchronize(st));
    S->n_sides = total;
    S->n_tasks = 0;
    S->low = lower;
    S->up = upper;
    *n_sides = total;
    if (total == 0) return PF_OK;
    if (!S->sides.ensure((size_t)total * sizeof(pf_call_side)) || !S->ctask.ensure((size_t)total * sizeof(CallTask)) ||
        !S->target.ensure((size_t)total * 4)) {
        pf::CtxErr{ctx} = "pf_call_scan: out of device memory";
        return PF_ERR_HIP;
    }
    ScanArgs a;
    a.flags = S->flags.as<uint8_t>(); a.plus = S->plus.as<uint32_t>(); a.minus = S->minus.as<uint32_t>();
    a.succ = ctx->d_succ; a.pred = ctx->d_pred; a.seq = ctx->d_seq; a.off = ctx->d_off; a.len = ctx->d_len; a.N = N; a.k = ctx->k;
    a.cov_sum = S->cov_sum.as<uint64_t>(); a.cov_min = S->cov_min.as<uint32_t>(); a.cov_miss = S->cov_miss.as<uint8_t>();
    a.per_strand = S->per_strand; a.low = lower; a.up = upper;
    a.side_base = S->side_base.as<uint32_t>(); a.sides = S->sides.as<pf_call_side>(); a.tasks = S->ctask.as<CallTask>();
    a.target = S->target.as<uint32_t>();
    a.n_colors = S->n_colors;
    a.ccov_sum = S->ccov_sum.as<uint64_t>(); a.ccov_min = S->ccov_min.as<uint32_t>(); a.ccov_max = S->ccov_max.as<uint32_t>(); a.ccov_miss = S->ccov_miss.as<uint8_t>();
    a.clow = S->col_low.as<uint32_t>(); a.cup = S->col_up.as<uint32_t>(); a.full = S->col_full.as<uint64_t>(); a.size_total = S->col_size.as<uint64_t>();
    ctx_begin(ctx, PF_K_CALL_SCAN);
    if (S->n_colors) k_call_sides<true><<<(N + 255) / 256, 256, 0, st>>>(a);
    else k_call_sides<false><<<(N + 255) / 256, 256, 0, st>>>(a);
    ctx_end(ctx);
    ctx_units(ctx, PF_K_CALL_SCAN, N);
    PF_HIP(hipGetLastError());
    return PF_OK;
}

int pf_call_resolve(pf_ctx *ctx, uint64_t *n_bubbles, uint32_t *err, uint32_t *err_unitig) {
    if (!ctx || !ctx->call || !n_bubbles || !err || !err_unitig) return PF_ERR_ARG;
    CallState *S = ctx->call;
    *n_bubbles = 0;
    *err = 0;
    *err_unitig = 0;
    S->n_tasks = 0;
    const uint64_t n = S->n_sides;
    if (n == 0) return PF_OK;
    PF_HIP(hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    if (!S->pending.ensure(n * 4) || !S->killed.ensure(n) || !S->rstate.ensure(n) || !S->rflag.ensure(n * 4) || !S->rsmall.ensure(64) ||
        !S->kept.ensure(n * 4)) {
        pf::CtxErr{ctx} = "pf_call_resolve: out of device memory";
        return PF_ERR_HIP;
    }
    PF_HIP(hipMemsetAsync(S->pending.p, 0, n * 4, st));
    PF_HIP(hipMemsetAsync(S->killed.p, 0, n, st));
    PF_HIP(hipMemsetAsync(S->rstate.p, 0, n, st));
    unsigned int *small = S->rsmall.as<unsigned int>();   // [0] undecided, [1] first_err, [2] n selected
    ResolveArgs a;
    a.sides = S->sides.as<pf_call_side>(); a.target = S->target.as<uint32_t>(); a.n = (uint32_t)n; a.pending = S->pending.as<int>();
    a.killed = S->killed.as<uint8_t>(); a.state = S->rstate.as<uint8_t>(); a.flag = S->rflag.as<uint32_t>();
    a.undecided = small; a.first_err = small + 1;
    const unsigned int init[3] = {0, 0xFFFFFFFFu, 0};
    PF_HIP(hipMemcpyAsync(small, init, 12, hipMemcpyHostToDevice, st));
    const unsigned grid = (unsigned)((n + 255) / 256);
    k_call_pending<<<grid, 256, 0, st>>>(a);
    for (uint64_t round = 0;; ++round) {
        if (round > n + 2) { pf::CtxErr{ctx} = "pf_call_resolve: the driver pass does not settle"; return PF_ERR_ARG; }
        k_call_resolve<<<grid, 256, 0, st>>>(a);
        PF_HIP(hipMemsetAsync(small, 0, 4, st));
        k_call_resolve<<<grid, 256, 0, st>>>(a);   // two rounds per look at the counter: the common case needs exactly two
        unsigned int h[2];
        PF_HIP(hipMemcpyAsync(h, small, 8, hipMemcpyDeviceToHost, st));
        PF_HIP(hipStreamSynchronize(st));
        if (h[1] != 0xFFFFFFFFu && h[0] == 0) {   // every side settled and an alive one carries an error: the first in order counts
            pf_call_side bad;
            PF_HIP(hipMemcpy(&bad, S->sides.as<pf_call_side>() + h[1], sizeof(bad), hipMemcpyDeviceToHost));
            *err = bad.err;
            *err_unitig = bad.err_unitig;
            return PF_OK;
        }
        if (h[0] == 0) break;
    }
    // the called sides, ascending
    size_t tmp = 0;
    hipcub::CountingInputIterator<uint32_t> ids(0);
    PF_HIP(hipcub::DeviceSelect::Flagged(nullptr, tmp, ids, S->rflag.as<uint32_t>(), S->kept.as<uint32_t>(), small + 2, (int)n, st));
    if (!S->scan_tmp.ensure(tmp)) { pf::CtxErr{ctx} = "pf_call_resolve: out of device memory"; return PF_ERR_HIP; }
    PF_HIP(hipcub::DeviceSelect::Flagged(S->scan_tmp.p, tmp, ids, S->rflag.as<uint32_t>(), S->kept.as<uint32_t>(), small + 2, (int)n, st));
    unsigned int sel = 0;
    PF_HIP(hipMemcpyAsync(&sel, small + 2, 4, hipMemcpyDeviceToHost, st));
    PF_HIP(hipStreamSynchronize(st));
    S->n_tasks = sel;
    *n_bubbles = sel;
    return PF_OK;
}

int pf_call_sides(pf_ctx *ctx, pf_call_side *out, uint64_t cap) {
    if (!ctx || !ctx->call) return PF_ERR_ARG;
    CallState *S = ctx->call;
    if (cap < S->n_sides || (S->n_sides && !out)) return PF_ERR_OVERFLOW;
    if (S->n_sides == 0) return PF_OK;
    PF_HIP(hipSetDevice(ctx->device));
    PF_HIP(hipMemcpyAsync(out, S->sides.p, (size_t)S->n_sides * sizeof(pf_call_side), hipMemcpyDefault, ctx->stream));
    PF_HIP(hipStreamSynchronize(ctx->stream));
    return PF_OK;
}

int pf_call_select(pf_ctx *ctx, const uint32_t *side_index, uint64_t n_tasks) {
    if (!ctx || !ctx->call || (n_tasks && !side_index)) return PF_ERR_ARG;
    CallState *S = ctx->call;
    if (n_tasks > S->n_sides) { pf::CtxErr{ctx} = "pf_call_select: more bubbles than open sides"; return PF_ERR_ARG; }
    PF_HIP(hipSetDevice(ctx->device));
    if (!S->kept.ensure(std::max<size_t>(n_tasks, 1) * 4)) { pf::CtxErr{ctx} = "pf_call_select: out of device memory"; return PF_ERR_HIP; }
    if (n_tasks) PF_HIP(hipMemcpyAsync(S->kept.p, side_index, (size_t)n_tasks * 4, hipMemcpyDefault, ctx->stream));
    PF_HIP(hipStreamSynchronize(ctx->stream));
    S->n_tasks = n_tasks;
    return PF_OK;
}

// one batch, first half: bubbles [t0, t1) of the selection up to the site coverages; out->n_called tells how far var_count advances
int pf_call_align_lane(pf_ctx *ctx, int lane, uint64_t t0, uint64_t t1, uint32_t complex_size, double match, double mismatch, double gap,
                       pf_call_result *out) {
    if (!ctx || !out || lane < 0 || lane >= PF_CALL_LANES) return PF_ERR_ARG;
    CallState *S = ctx->call;
    if (!S || t0 > t1 || t1 > S->n_tasks) { pf::CtxErr{ctx} = "pf_call_align: range outside the selection"; return PF_ERR_ARG; }
    memset(out, 0, sizeof(*out));
    CallState::AlignOut &O = S->lane[lane];
    O.nb = 0;
    O.t0 = t0;
    O.cur = pf_call_result{};   // (an empty slice -- fewer bubbles than ranks -- must not hand pf_call_text the previous pass's counters)
    if (t1 == t0) return PF_OK;
    if (t1 - t0 > (1u << 24)) { pf::CtxErr{ctx} = "pf_call_align: at most 2^24 bubbles per batch"; return PF_ERR_ARG; }
    PF_HIP(hipSetDevice(ctx->device));
    CallState::AlignWork &W = S->work[lane];
    if (lane != 0 && !W.stream) { PF_HIP(lane_stream_create(&W.stream, lane)); W.own_stream = true; }
    hipStream_t st = lane == 0 ? ctx->stream : W.stream;   // (lane 0: the context's stream, whatever pf_set_stream made it since)
    // the write pass of K-TEXT over what this lane held may still be running (pf_call_text_range_lane does not wait for it)
    for (hipEvent_t e : O.read_ev) if (e) PF_HIP(hipStreamWaitEvent(st, e, 0));
    // (another lane's stream needs no event to wait for: the scan and the selection on the context's stream ended in host waits --
    // pf_call_resolve / pf_call_select hand the host the number of bubbles this call's range is cut from)
    // launch timing by place: calls on other lanes time their launches at the same time
    size_t tl_at = (size_t)-1;
    auto tbegin = [&](int kernel, hipStream_t s) { (void)ctx_begin_at(ctx, kernel, s, &tl_at); };
    auto tend = [&](hipStream_t s) { ctx_end_at(ctx, tl_at, s); };
    const uint32_t nb = (uint32_t)(t1 - t0);
    const int k = ctx->k;
    const bool trace_stages = getenv("PF_TRACE_ALIGN") != nullptr;   // where a call's time goes (a first call above all)
    const auto t_enter = std::chrono::steady_clock::now();
    auto ta = [&](const char *what) {
        if (trace_stages) fprintf(stderr, "[pf_call_align]   %-34s %.2f ms\n", what, std::chrono::duration<double>(std::chrono::steady_clock::now() - t_enter).count() * 1e3);
    };
    const char *oom = "pf_call_run: out of device memory";
#define NEED(buf, bytes) do { if (!(buf).ensure(bytes)) { pf::CtxErr{ctx} = oom; return PF_ERR_HIP; } } while (0)
    NEED(W.counters, sizeof(CallCounters));
    NEED(W.btask, (size_t)nb * sizeof(pf_bubble_task));
    NEED(W.queues, (size_t)NQ * nb * 4);
    NEED(W.blist, (size_t)nb * 4);
    NEED(O.res, (size_t)nb * sizeof(pf_bubble_result));
    NEED(O.sv_off, (size_t)nb * 8);
    NEED(W.has, (size_t)nb * 4);
    NEED(O.vc, (size_t)nb * 4);
    CallCounters *d_cnt = W.counters.as<CallCounters>();
    CallCounters hc;
    // K-PATHS scratch: stacks sized by the complex size (a non-complex bubble has at most that many vertices)
    const uint32_t depth_cap = std::max<uint32_t>(complex_size + 4, 16);
    const uint64_t paths_per_wave = ((256 * 8 + 256 * 4 + (10ull * depth_cap + 4) * 4) + 255) & ~255ull;
    constexpr int paths_per_cu = 16;
    const int paths_grid = ctx->n_cu * paths_per_cu;
    NEED(W.paths_scr, paths_per_wave * paths_grid);
    // bubbles of more than 255 walks in one range: the list grows to what an attempt asked for (advisor, round 3: an overflow used
    // to be reported as "more than 65535 paths", a refusal of a run no bubble of which had that many)
    if (W.mlist_cap < 4096) W.mlist_cap = 4096;
    NEED(W.mlist, (size_t)W.mlist_cap * 4);
    PathArgs ph_keep = {};
    const int paths_force_scratch = [] { const char *e = getenv("PF_PATHS_SCRATCH"); return e && atoi(e) ? 1 : 0; }();   // (read per call: tests)

    // ---- K-PREP, K-SNP, K-PATHS, K-BUBBLE; every pool grows until the batch fits (first batches of a run only) ----
    NEED(W.slist, (size_t)nb * 4);
    NEED(W.plist, (size_t)nb * 4);
    NEED(W.plist2, (size_t)nb * 4);
    NEED(W.klist, (size_t)nb * 4);
    NEED(W.klist_b, (size_t)nb * 4);
    NEED(W.tlist, (size_t)nb * 4);
    NEED(W.tlist4, (size_t)nb * 4);
    const int snp_ok = snp_shortcut_scores(match, mismatch, gap) ? 1 : 0;
    // K-PAIR: register-bound (the score row of the fill is 65 / 129 registers): 3 / 2 wavefronts per SIMD, the grid loops over its list
    // (scores of sane magnitude only: the fill adds them in ints)
    const bool pair_tier = std::fabs(match) < 1e5 && std::fabs(mismatch) < 1e5 && std::fabs(gap) < 1e5;
    const bool stack_tier = stack_scores(match, mismatch, gap);
    // what K-STACK is given (read per call: tools/ab_pass.py): 1 = bubbles of three and more paths of one length, 2 = also those whose
    // later paths are shorter than the first (one gap run each), 3 = also the two-path bubbles ahead of K-PAIR
    const int stack_level = [] { const char *e = getenv("PF_STACK_LEVEL"); return e ? std::max(1, std::min(3, atoi(e))) : 1; }();
    const int stack_grid = ctx->n_cu * 8;
    if (stack_tier) NEED(W.stack_scr, stack_scratch_bytes() * stack_grid);
    // K-TRIO is OFF unless asked for: measured at BASELINE.json's configs[2] (profiles/r07_experiments.txt) it takes 24 k of K-BUBBLE's 34 k
    // bubbles and saves K-BUBBLE 1.0 ms per pass, but costs 3.2 ms itself -- a whole 96 x 70 fill per THREAD is 0.9 ms of dependent
    // instructions on a wavefront that has the SIMD to itself, whatever the number of bubbles; K-BUBBLE's wavefront per bubble
    // spreads the same cells over 64 lanes.  Kept (and held to the oracle by tests/test_gpu_call.py) for graphs with enough such
    // bubbles to fill the device several times over.  PF_TRIO_TIER=1 switches it on (read per call).
    const bool trio_env = [] { const char *e = getenv("PF_TRIO_TIER"); return e && e[0] == '1'; }();
    const bool trio_tier = trio_env && stack_tier && pair_tier;
    const int trio_grid = ctx->n_cu * 8;
    const bool pair_integral = match == std::floor(match) && mismatch == std::floor(mismatch) && gap == std::floor(gap);
    const int pair_grid = ctx->n_cu * 12, pair_grid2 = ctx->n_cu * 4;
    if (pair_tier) NEED(W.pair_scr, PairGeom<PAIR_MAX>::scratch_bytes * pair_grid);
    unsigned long long heads[4] = {0, 0, 0, 0};
    uint64_t n_jobs = 0;
    ta("lists and scratch");
    for (int attempt = 0;; ++attempt) {
        if (attempt > 5) { pf::CtxErr{ctx} = "pf_call_align: pools do not converge"; return PF_ERR_OVERFLOW; }
        const uint64_t path_cap = std::max<uint64_t>(S->path_pool, (uint64_t)nb / 2 + 128ull * paths_grid + 1024);   // (a started piece per wavefront)
        // (first-pass sizes, learnt afterwards; a pool that turns out too small costs a repeated attempt -- at configs[4]'s parameters,
        // k = 31 and insertions to 50 bp, a whole K-BUBBLE run thrown away: 334 B of rows, 2.4 sites, 8 group bytes, 1.1 indel lengths
        // and 55 B of path text per bubble there; 216 B / 1.2 / 2.7 / 0.03 / 17 B at configs[2]'s)
        const uint64_t text_cap = std::max<uint64_t>(S->text_pool, (uint64_t)nb * FIRST_PATH_TEXT + (1u << 16));
        const uint64_t cap_text = std::max<uint64_t>(S->otext_cap, (uint64_t)FIRST_ROW_TEXT * nb + (1u << 16));
        const uint64_t cap_sites = std::max<uint64_t>(S->osites_cap, (uint64_t)FIRST_SITES * nb + 64);
        const uint64_t cap_groups = std::max<uint64_t>(S->ogroups_cap, (uint64_t)FIRST_GROUPS * nb + 64);
        const uint64_t cap_ilen = std::max<uint64_t>(S->oilen_cap, (uint64_t)FIRST_ILEN * nb + 64);
        NEED(W.bpath, ((size_t)4 * nb + path_cap) * sizeof(pf_bubble_path));
        NEED(W.ptext, text_cap);
        const uint64_t walk_cap = S->n_colors ? std::max<uint64_t>(W.walk_cap, (uint64_t)nb * 2 + 256ull * paths_grid + 1024) : 0;
        if (S->n_colors) {
            NEED(W.walk_pool, walk_cap * 4);
            NEED(W.walk_off, (size_t)nb * 8);
        }
        NEED(O.otext, cap_text);
        NEED(O.osites, cap_sites * sizeof(pf_bubble_site));
        NEED(O.ogroups, cap_groups);
        NEED(O.oilen, cap_ilen * 4);
        unsigned long long *d_heads = bubble_pool_heads(ctx, lane);
        if (!d_heads) return PF_ERR_HIP;
        PF_HIP(hipMemsetAsync(d_cnt, 0, sizeof(CallCounters), st));
        PF_HIP(hipMemsetAsync(d_heads, 0, 32, st));
        PrepArgs pa;
        pa.ct = S->ctask.as<CallTask>(); pa.kept = S->kept.as<uint32_t>(); pa.t0 = t0; pa.nb = nb; pa.len = ctx->d_len;
        pa.btask = W.btask.as<pf_bubble_task>(); pa.bpath = W.bpath.as<pf_bubble_path>(); pa.res = O.res.as<pf_bubble_result>();
        pa.lists = CallLists{W.queues.as<uint32_t>(), W.blist.as<uint32_t>(), W.slist.as<uint32_t>(), W.plist.as<uint32_t>(), W.plist2.as<uint32_t>(), W.klist.as<uint32_t>(), W.klist_b.as<uint32_t>(), W.tlist.as<uint32_t>(), W.tlist4.as<uint32_t>(), nb};
        pa.snp_ok = snp_ok;
        pa.pair_ok = pair_tier ? 1 : 0;
        pa.stack_ok = stack_tier ? stack_level : 0;
        pa.trio_ok = trio_tier ? 1 : 0;
        pa.cnt = d_cnt;
        tbegin(PF_K_CALL_PREP, st);
        k_call_prep<<<(nb + 255) / 256, 256, 0, st>>>(pa);
        tend(st);
        // K-PATHS (branching bubbles) beside K-SNP and K-PAIR (two-path bubbles): latency-bound walks next to an issue-bound fill
        constexpr bool fork_paths = true;
        hipStream_t pst = st;
        if (fork_paths) {
            if (!W.side_stream) {
                PF_HIP(lane_stream_create(&W.side_stream, lane));
                PF_HIP(hipEventCreateWithFlags(&W.ev_prep, hipEventDisableTiming));
                PF_HIP(hipEventCreateWithFlags(&W.ev_paths, hipEventDisableTiming));
            }
            pst = W.side_stream;
            PF_HIP(hipEventRecord(W.ev_prep, st));
            PF_HIP(hipStreamWaitEvent(pst, W.ev_prep, 0));
        }
        {
            PathArgs ph;
            ph.ct = pa.ct; ph.kept = pa.kept; ph.t0 = t0; ph.nb = nb; ph.blist = pa.lists.blist; ph.succ = ctx->d_succ; ph.seq = ctx->d_seq;
            ph.off = ctx->d_off; ph.len = ctx->d_len; ph.k = k; ph.depth_cap = depth_cap; ph.scratch = W.paths_scr.as<uint8_t>();
            ph.scratch_per_wave = paths_per_wave; ph.force_scratch = paths_force_scratch; ph.btask = pa.btask; ph.bpath = pa.bpath; ph.path_cap = path_cap;
            ph.text = W.ptext.as<char>(); ph.text_cap = text_cap; ph.queues = pa.lists.queues; ph.cnt = d_cnt;
            ph.klist = pa.lists.klist_b; ph.stack_ok = pa.stack_ok;
            ph.tlist = pa.lists.tlist; ph.tlist4 = pa.lists.tlist4; ph.trio_ok = pa.trio_ok;
            ph.walk_pool = S->n_colors ? W.walk_pool.as<uint32_t>() : nullptr; ph.walk_off = W.walk_off.as<uint64_t>(); ph.walk_cap = walk_cap;
            ph.max_paths = MAX_PATHS; ph.n_list = &d_cnt->n_branching; ph.mlist = W.mlist.as<uint32_t>(); ph.mlist_cap = W.mlist_cap;
            ph_keep = ph;
            tbegin(PF_K_CALL_PATHS, pst);
            k_call_paths<false><<<paths_grid, 64, 0, pst>>>(ph);
            tend(pst);
            if (fork_paths) PF_HIP(hipEventRecord(W.ev_paths, pst));
        }
        if (snp_ok) {
            SnpArgs sn;
            sn.ct = pa.ct; sn.kept = pa.kept; sn.t0 = t0; sn.nb = nb; sn.slist = pa.lists.slist; sn.seq = ctx->d_seq; sn.off = ctx->d_off;
            sn.len = ctx->d_len; sn.res = pa.res; sn.otext = O.otext.as<char>(); sn.text_cap = cap_text;
            sn.osites = O.osites.as<pf_bubble_site>(); sn.site_cap = cap_sites; sn.ogroups = O.ogroups.as<uint8_t>(); sn.group_cap = cap_groups;
            sn.heads = d_heads; sn.lists = pa.lists; sn.pair_ok = pa.pair_ok; sn.stack_ok = pa.stack_ok; sn.cnt = d_cnt;
            tbegin(PF_K_CALL_SNP, st);
            k_call_snp<<<(nb + 255) / 256, 256, 0, st>>>(sn);   // (the list length is on the device: surplus threads leave at once)
            tend(st);
        }
        StackArgs sk;
        if (stack_tier) {
            // K-STACK, first launch: the strict bubbles K-PREP and K-SNP listed (two paths that are not a single mismatch, three and
            // four paths); what it cannot certify is K-PAIR's (launched behind it), K-TRIO's or K-BUBBLE's
            sk.list = pa.lists.klist; sk.n_list = &d_cnt->n_stack; sk.scratch = W.stack_scr.as<uint8_t>();
            sk.btask = pa.btask; sk.bpath = pa.bpath; sk.ptext = W.ptext.as<char>();
            sk.seq = ctx->d_seq; sk.off = ctx->d_off; sk.len = ctx->d_len;
            sk.M = (int)match; sk.D = (int)mismatch; sk.G = (int)gap;
            sk.res = pa.res; sk.otext = O.otext.as<char>(); sk.text_cap = cap_text; sk.osites = O.osites.as<pf_bubble_site>(); sk.site_cap = cap_sites;
            sk.ogroups = O.ogroups.as<uint8_t>(); sk.group_cap = cap_groups; sk.oilen = O.oilen.as<uint32_t>(); sk.ilen_cap = cap_ilen;
            sk.heads = d_heads; sk.lists = pa.lists; sk.cnt = d_cnt;
            sk.trio_ok = pa.trio_ok; sk.pair_ok = pa.pair_ok;
            tbegin(PF_K_CALL_STACK, st);
            k_call_stack<<<stack_grid, 64, 0, st>>>(sk);
            tend(st);
        }
        PairArgs pr;
        static const bool pair_stats = getenv("PF_PAIR_STATS") != nullptr;
        DevTmp<unsigned long long> prof_;
        if (pair_tier) {
            pr.ct = pa.ct; pr.kept = pa.kept; pr.t0 = t0; pr.seq = ctx->d_seq; pr.off = ctx->d_off; pr.len = ctx->d_len;
            pr.M = match; pr.D = mismatch; pr.G = gap;
            pr.Mi = (int)match; pr.Di = (int)mismatch; pr.Gi = (int)gap;
            pr.list = pa.lists.plist; pr.n_list = &d_cnt->n_pair; pr.n_done = &d_cnt->n_pair_done;
            pr.scratch = W.pair_scr.as<uint8_t>(); pr.res = pa.res; pr.otext = O.otext.as<char>(); pr.text_cap = cap_text;
            pr.osites = O.osites.as<pf_bubble_site>(); pr.site_cap = cap_sites; pr.ogroups = O.ogroups.as<uint8_t>(); pr.group_cap = cap_groups;
            pr.oilen = O.oilen.as<uint32_t>(); pr.ilen_cap = cap_ilen; pr.heads = d_heads; pr.lists = pa.lists; pr.cnt = d_cnt;
            pr.prof = nullptr;
            if (pair_stats) {
                PF_HIP(prof_.alloc(64));
                PF_HIP(hipMemsetAsync(prof_.p, 0, 64, st));
                pr.prof = prof_.p;
            }
            tbegin(PF_K_CALL_PAIR, st);
            if (pair_integral) k_call_pair<PAIR_MAX, true><<<pair_grid, 64, 0, st>>>(pr);
            else k_call_pair<PAIR_MAX, false><<<pair_grid, 64, 0, st>>>(pr);
            tend(st);
            if (pair_stats) {
                unsigned long long h[8];
                PF_HIP(hipMemcpy(h, prof_.p, 64, hipMemcpyDeviceToHost));
                fprintf(stderr, "[k_call_pair] %llu wavefront rounds; lane-0 ticks (10 ns): decode %llu fill %llu traceback %llu classify %llu publish %llu\n", h[5],
                        h[0], h[1], h[2], h[3], h[4]);
            }
        }
        if (fork_paths) PF_HIP(hipStreamWaitEvent(st, W.ev_paths, 0));
        if (stack_tier) {
            // K-STACK, second launch: the branching bubbles K-PATHS listed (their two-path rejects cannot go to K-PAIR, which reads
            // the inner unitigs of a strict bubble: K-BUBBLE's)
            sk.list = pa.lists.klist_b; sk.n_list = &d_cnt->n_stack_b; sk.pair_ok = 0;
            tbegin(PF_K_CALL_STACK, st);
            k_call_stack<<<stack_grid, 64, 0, st>>>(sk);
            tend(st);
        }
        PF_HIP(hipGetLastError());
        PF_HIP(hipMemcpyAsync(&hc, d_cnt, sizeof(hc), hipMemcpyDeviceToHost, st));
        PF_HIP(hipStreamSynchronize(st));
        ta("K-PREP .. K-STACK done");
        bool again = false;   // kernels launched once the host knows their lists' lengths: the counters are read once more behind them
        if (pair_tier && hc.n_pair2) {
            // second tier (paths of 65 .. 128 bases, or a longer indel than the first tier's band follows): few on most graphs, so
            // its scratch and its launch wait until the host knows there are any; its rejects join K-BUBBLE's queues
            const int g2 = (int)std::min<uint32_t>((hc.n_pair2 + 63) / 64, (uint32_t)pair_grid2);
            pr.list = pa.lists.plist2; pr.n_list = &d_cnt->n_pair2; pr.n_done = &d_cnt->n_pair2_done;
            pr.prof = nullptr;
            const uint32_t tier2_min = [] { const char *e = getenv("PF_PAIR2_MIN"); return e ? (uint32_t)atoi(e) : 0u; }();   // (read per call: tests run the tier on a few bubbles)
            if (hc.n_pair2 < (tier2_min ? tier2_min : (uint32_t)ctx->n_cu * 32u)) {
                k_call_pair2_reroute<<<(hc.n_pair2 + 255) / 256, 256, 0, st>>>(pr);
            } else {
                NEED(W.pair_scr2, PairGeom<PAIR_MAX2>::scratch_bytes * g2);
                pr.scratch = W.pair_scr2.as<uint8_t>();
                tbegin(PF_K_CALL_PAIR, st);
                if (pair_integral) k_call_pair<PAIR_MAX2, true><<<g2, 64, 0, st>>>(pr);
                else k_call_pair<PAIR_MAX2, false><<<g2, 64, 0, st>>>(pr);
                tend(st);
            }
            again = true;
        }
        if (trio_tier && (hc.n_trio || hc.n_trio4)) {
            // K-TRIO: behind K-PATHS (its lists), K-PREP (strict bubbles of three and four paths) and K-STACK (what that could not
            // certify): one thread per (bubble, later path) aligns the path to path 0, one thread per bubble finishes
            const uint64_t n3 = hc.n_trio, n4 = hc.n_trio4, pairs = 2 * n3 + 3 * n4;
            const int ga = (int)std::min<uint64_t>((pairs + 63) / 64, (uint64_t)trio_grid);
            NEED(W.trio_scr, PairGeom<TRIO_MAX>::scratch_bytes * ga);
            NEED(W.trio_rows, (3 * n3 + 4 * n4) * TRIO_ROW);
            NEED(W.trio_ok, n3 + n4);
            PF_HIP(hipMemsetAsync(W.trio_ok.p, 1, n3 + n4, st));
            TrioArgs tr;
            tr.btask = pa.btask; tr.bpath = pa.bpath; tr.ptext = W.ptext.as<char>();
            tr.seq = ctx->d_seq; tr.off = ctx->d_off; tr.len = ctx->d_len;
            tr.M = (int)match; tr.D = (int)mismatch; tr.G = (int)gap;
            tr.scratch = W.trio_scr.as<uint8_t>();
            tr.res = pa.res; tr.otext = O.otext.as<char>(); tr.text_cap = cap_text; tr.osites = O.osites.as<pf_bubble_site>(); tr.site_cap = cap_sites;
            tr.ogroups = O.ogroups.as<uint8_t>(); tr.group_cap = cap_groups; tr.oilen = O.oilen.as<uint32_t>(); tr.ilen_cap = cap_ilen;
            tr.heads = d_heads; tr.lists = pa.lists; tr.cnt = d_cnt;
            tbegin(PF_K_CALL_TRIO, st);
            if (n3) {
                tr.list = pa.lists.tlist; tr.n_list = (uint32_t)n3; tr.rows = W.trio_rows.as<char>(); tr.okflag = W.trio_ok.as<uint8_t>();
                k_call_trio_align<3><<<(int)std::min<uint64_t>((2 * n3 + 63) / 64, (uint64_t)ga), 64, 0, st>>>(tr);
                k_call_trio_finish<3><<<(unsigned)((n3 + 255) / 256), 256, 0, st>>>(tr);
            }
            if (n4) {
                tr.list = pa.lists.tlist4; tr.n_list = (uint32_t)n4; tr.rows = W.trio_rows.as<char>() + 3 * n3 * TRIO_ROW; tr.okflag = W.trio_ok.as<uint8_t>() + n3;
                k_call_trio_align<4><<<(int)std::min<uint64_t>((3 * n4 + 63) / 64, (uint64_t)ga), 64, 0, st>>>(tr);
                k_call_trio_finish<4><<<(unsigned)((n4 + 255) / 256), 256, 0, st>>>(tr);
            }
            tend(st);
            again = true;
        }
        if (again) {
            PF_HIP(hipGetLastError());
            PF_HIP(hipMemcpyAsync(&hc, d_cnt, sizeof(hc), hipMemcpyDeviceToHost, st));
            PF_HIP(hipStreamSynchronize(st));
        }
        ctx_units(ctx, PF_K_CALL_PREP, nb);
        if (snp_ok) ctx_units(ctx, PF_K_CALL_SNP, hc.n_snp);
        if (pair_tier) ctx_units(ctx, PF_K_CALL_PAIR, hc.n_pair + hc.n_pair2);
        if (stack_tier) ctx_units(ctx, PF_K_CALL_STACK, hc.n_stack + hc.n_stack_b);
        if (trio_tier) ctx_units(ctx, PF_K_CALL_TRIO, hc.n_trio + hc.n_trio4);
        ctx_units(ctx, PF_K_CALL_PATHS, hc.n_branching);
        if (getenv("PF_TRACE_ALIGN")) fprintf(stderr, "[pf_call_align] bubbles of more than 255 walks: %u, err %u\n", hc.n_many, hc.err);
        if (hc.n_many > W.mlist_cap && !(hc.err & 33u)) {
            W.mlist_cap = hc.n_many + hc.n_many / 8 + 64;
            NEED(W.mlist, (size_t)W.mlist_cap * 4);
            continue;
        }
        if (hc.n_many && !(hc.err & 33u)) {
            // bubbles of more than 255 walks: walked again by a few wavefronts with room for PATHS_BIG walks each
            const uint32_t n_many = hc.n_many;
            const int big_grid = (int)std::min<uint32_t>(n_many, 32);
            const uint64_t big_per_wave = (((((uint64_t)10 * depth_cap + 4) * 4 + 7) & ~7ull) + ((uint64_t)PATHS_BIG + 1) * 12 + 255) & ~255ull;
            NEED(W.paths_big_scr, big_per_wave * big_grid);
            PathArgs pb = ph_keep;
            pb.blist = W.mlist.as<uint32_t>(); pb.n_list = &d_cnt->n_many; pb.max_paths = PATHS_BIG; pb.mlist = nullptr; pb.mlist_cap = 0;
            pb.scratch = W.paths_big_scr.as<uint8_t>(); pb.scratch_per_wave = big_per_wave;
            tbegin(PF_K_CALL_PATHS, st);
            k_call_paths<true><<<big_grid, 64, 0, st>>>(pb);
            tend(st);
            PF_HIP(hipGetLastError());
            PF_HIP(hipMemcpyAsync(&hc, d_cnt, sizeof(hc), hipMemcpyDeviceToHost, st));
            PF_HIP(hipStreamSynchronize(st));
        }
        if (hc.err & 33u) {
            char where[96];
            snprintf(where, sizeof where, " (superbubble from unitig %u%c to unitig %u%c)", (hc.err_entrance >> 1) + 1, (hc.err_entrance & 1) ? '-' : '+',
                     (hc.err_exit >> 1) + 1, (hc.err_exit & 1) ? '-' : '+');
            pf::CtxErr{ctx} = std::string(hc.err & 1u ? "pf_call_run: a bubble has more than 65535 paths" : "pf_call_run: a bubble is deeper than the complex size allows") + where;
            return PF_ERR_ARG;
        }
        amax(S->otext_cap, cap_text); amax(S->osites_cap, cap_sites);
        amax(S->ogroups_cap, cap_groups); amax(S->oilen_cap, cap_ilen);
        amax(S->path_pool, path_cap);
        amax(S->text_pool, text_cap);
        static const bool trace_retry = getenv("PF_TRACE_ALIGN") != nullptr;
        if (trace_retry)
            fprintf(stderr, "[pf_call_align] attempt %d: %u bubbles, path pool %llu of %llu, path text %llu of %llu, rows text cap %llu (needs %llu)\n", attempt, nb,
                    (unsigned long long)hc.path_head, (unsigned long long)path_cap, (unsigned long long)hc.text_head, (unsigned long long)text_cap,
                    (unsigned long long)cap_text, (unsigned long long)(3 * hc.text_head + 128ull * hc.n_branching + 160ull * nb));
        if (hc.path_head > path_cap || hc.text_head > text_cap || hc.walk_head > walk_cap) {
            amax(S->path_pool, hc.path_head + hc.path_head / 8 + 1024);
            amax(S->text_pool, hc.text_head + hc.text_head / 8 + 4096);
            if (S->n_colors) W.walk_cap = std::max<uint64_t>(W.walk_cap, hc.walk_head + hc.walk_head / 8 + 1024);
            continue;
        }
        if (S->n_colors) W.walk_cap = std::max(W.walk_cap, walk_cap);
        // the aligned rows of the branching bubbles come on top of what K-SNP took: make room before K-BUBBLE runs
        // (the path text is handed out in per-wavefront chunks: its size varies by a few per mille from pass to pass, hence the margin)
        if (cap_text < 3 * hc.text_head + 128ull * hc.n_branching + 160ull * nb) {
            const uint64_t need = 3 * hc.text_head + 128ull * hc.n_branching + 160ull * nb;
            amax(S->otext_cap, need + need / 16 + 4096);
            continue;
        }
        n_jobs = 0;
        for (int x = 0; x < NQ; ++x) n_jobs += hc.q_n[x];
        // K-BUBBLE: the queues are heavy-then-light per class; compact them into one index array
        NEED(W.scan_tmp2, std::max<size_t>((size_t)n_jobs, 1) * 4);
        {
            uint32_t *d_idx = W.scan_tmp2.as<uint32_t>();
            size_t at = 0;
            for (int x = 0; x < NQ; ++x) {
                if (!hc.q_n[x]) continue;
                PF_HIP(hipMemcpyAsync(d_idx + at, W.queues.as<uint32_t>() + (size_t)x * nb, (size_t)hc.q_n[x] * 4, hipMemcpyDeviceToDevice, st));
                at += hc.q_n[x];
            }
        }
        BubbleLaunch BL;
        BL.text = W.ptext.as<char>(); BL.paths = W.bpath.as<pf_bubble_path>(); BL.tasks = W.btask.as<pf_bubble_task>();
        BL.n_tasks = nb; BL.idx = W.scan_tmp2.as<uint32_t>();
        for (int c = 0; c <= kBubLdsClasses; ++c) BL.n_cls[c] = hc.q_n[2 * c] + hc.q_n[2 * c + 1];
        BL.max_need = hc.max_need; BL.retry_need = hc.retry_need;
        BL.match = match; BL.mismatch = mismatch; BL.gap = gap;
        BL.res = O.res.as<pf_bubble_result>(); BL.otext = O.otext.as<char>(); BL.osites = O.osites.as<pf_bubble_site>();
        BL.ogroups = O.ogroups.as<uint8_t>(); BL.oilen = O.oilen.as<uint32_t>();
        BL.text_cap = cap_text; BL.site_cap = cap_sites; BL.group_cap = cap_groups; BL.ilen_cap = cap_ilen;
        BL.keep_heads = true;
        BL.lane = lane; BL.stream = st;
        ta("K-BUBBLE launching");
        const int bst = bubble_launch(ctx, BL, heads);
        ta("K-BUBBLE done");
        if (trace_retry)
            fprintf(stderr, "[pf_call_align] attempt %d: K-BUBBLE status %d, pools text %llu of %llu, sites %llu of %llu, groups %llu of %llu, indel lengths %llu of %llu\n", attempt, bst,
                    heads[0], (unsigned long long)cap_text, heads[1], (unsigned long long)cap_sites, heads[2], (unsigned long long)cap_groups, heads[3], (unsigned long long)cap_ilen);
        if (bst == PF_ERR_OVERFLOW && (heads[0] > cap_text || heads[1] > cap_sites || heads[2] > cap_groups || heads[3] > cap_ilen)) {
            amax(S->otext_cap, heads[0] + heads[0] / 8);
            amax(S->osites_cap, heads[1] + heads[1] / 8);
            amax(S->ogroups_cap, heads[2] + heads[2] / 8);
            amax(S->oilen_cap, heads[3] + heads[3] / 8);
            continue;
        }
        if (bst != PF_OK) return bst;
        break;
    }
    out->n_branching = hc.n_branching;
    out->align_jobs = n_jobs + hc.n_snp_done + hc.n_pair_done + hc.n_pair2_done + hc.n_stack_done + hc.n_trio_done;
    out->snp_jobs = hc.n_snp_done; out->pair_jobs = hc.n_pair_done + hc.n_pair2_done; out->wave_jobs = n_jobs; out->stack_jobs = hc.n_stack_done;
    out->trio_jobs = hc.n_trio_done;

    // ---- bubble numbering inside the batch (launched ahead of K-SITES, read with its counters: one wait for both) ----
    k_call_has<<<(nb + 255) / 256, 256, 0, st>>>(O.res.as<pf_bubble_result>(), nb, W.has.as<uint32_t>());
    size_t tmp1 = 0;
    PF_HIP(hipcub::DeviceScan::InclusiveSum(nullptr, tmp1, W.has.as<uint32_t>(), O.vc.as<uint32_t>(), (int)nb, st));
    NEED(W.scan_tmp, tmp1);
    PF_HIP(hipcub::DeviceScan::InclusiveSum(W.scan_tmp.p, tmp1, W.has.as<uint32_t>(), O.vc.as<uint32_t>(), (int)nb, st));
    uint32_t n_called = 0;
    PF_HIP(hipMemcpyAsync(&n_called, O.vc.as<uint32_t>() + (nb - 1), 4, hipMemcpyDeviceToHost, st));

    // ---- K-SITES ----
    // (a site string is k characters long unless it takes the raw columns up to its row's end -- substr with a negative count,
    // src/CDBG.cpp:1499 -- or more than k characters agree behind an indel: a launch that meets one longer than its room says how
    // long, ks_need, and is repeated with that much)
    uint32_t KS = std::max<uint32_t>((uint32_t)(2 * k + 64), S->sites_ks.load());
    if (hc.n_branching) {
        const uint32_t C = S->n_colors;
        const uint64_t rows_cap = std::max<uint64_t>(256, ((uint64_t)hc.max_rows + 63) & ~63ull);
        constexpr int sites_per_cu = 16;
        for (int attempt = 0, ks_attempt = 0;; ++attempt) {
            const uint64_t sites_per_wave = ((2 * rows_cap * KS + rows_cap * (4 + 4 + 4 + 1 + 1 + 8) + (C ? rows_cap * (8 + 8 + 8ull * C) : 0)) + 255) & ~255ull;
            // (tables for thousands of rows: fewer wavefronts, at most 2 GB of them)
            const int sites_grid = (int)std::max<uint64_t>(1, std::min<uint64_t>(std::min<uint32_t>(hc.n_branching, (uint32_t)(ctx->n_cu * sites_per_cu)), (2ull << 30) / sites_per_wave));
            NEED(W.sites_scr, sites_per_wave * sites_grid);
            const uint64_t sv_cap = std::max<uint64_t>(S->sv_pool, 8ull * std::max<uint32_t>(C, 1) * hc.n_branching + 1024ull * sites_grid + 1024);   // (a started chunk per wavefront)
            NEED(O.sv, sv_cap * 8);
            SiteArgs sa;
            sa.ct = S->ctask.as<CallTask>(); sa.kept = S->kept.as<uint32_t>(); sa.t0 = t0; sa.blist = W.blist.as<uint32_t>();
            sa.res = O.res.as<pf_bubble_result>(); sa.otext = O.otext.as<char>(); sa.osites = O.osites.as<pf_bubble_site>();
            sa.ogroups = O.ogroups.as<uint8_t>(); sa.k = k; sa.tab = ctx->d_tab; sa.mask = ctx->tab_cap - 1;
            sa.one_strand = ctx->tab_one_strand; sa.tab_exact = ctx->tab_exact; sa.low = S->low; sa.up = S->up; sa.ks = KS; sa.rows_cap = (uint32_t)rows_cap;
            sa.scratch = W.sites_scr.as<uint8_t>(); sa.scratch_per_wave = sites_per_wave; sa.sv_off = O.sv_off.as<uint64_t>();
            sa.sv = O.sv.as<double>(); sa.sv_cap = sv_cap; sa.cnt = d_cnt;
            sa.n_colors = C;
            sa.ctab = CTab{ctx->d_ctab, ctx->ctab_cap - 1, ctx->ctab_line_bytes}; sa.c_one_strand = ctx->ctab_one_strand; sa.c_unread = ctx->ctab_unread;
            sa.clow = S->col_low.as<uint32_t>(); sa.cup = S->col_up.as<uint32_t>(); sa.full = S->col_full.as<uint64_t>();
            sa.part_first = S->part_first.as<uint32_t>(); sa.part_colour = S->part_colour.as<uint32_t>(); sa.part_word = S->part_word.as<uint64_t>();
            sa.part_bits = S->part_bits.as<uint64_t>(); sa.walk_pool = W.walk_pool.as<uint32_t>(); sa.walk_off = W.walk_off.as<uint64_t>();
            sa.seq = ctx->d_seq; sa.off = ctx->d_off; sa.len = ctx->d_len;
            static const bool sites_stats = getenv("PF_SITES_STATS") != nullptr;   // measurements: where a wavefront's time goes
            DevTmp<unsigned long long> sprof_;
            sa.prof = nullptr;
            if (sites_stats) {
                PF_HIP(sprof_.alloc((size_t)sites_grid * 48));
                PF_HIP(hipMemsetAsync(sprof_.p, 0, (size_t)sites_grid * 48, st));
                sa.prof = sprof_.p;
            }
            PF_HIP(hipMemsetAsync(&d_cnt->sites_next, 0, 4, st));
            PF_HIP(hipMemsetAsync(&d_cnt->sv_head, 0, 8, st));
            PF_HIP(hipMemsetAsync(&d_cnt->site_strings, 0, 8, st));
            tbegin(PF_K_CALL_SITES, st);
            if (C) k_call_sites<true><<<sites_grid, 64, 0, st>>>(sa);
            else k_call_sites<false><<<sites_grid, 64, 0, st>>>(sa);
            tend(st);
            ctx_units(ctx, PF_K_CALL_SITES, hc.n_branching);
            PF_HIP(hipGetLastError());
            if (sa.prof) {
                PF_HIP(hipStreamSynchronize(st));
                std::vector<unsigned long long> h((size_t)sites_grid * 6);
                PF_HIP(hipMemcpy(h.data(), sprof_.p, h.size() * 8, hipMemcpyDeviceToHost));
                unsigned long long sum[6] = {0, 0, 0, 0, 0, 0}, mx = 0;
                for (int w = 0; w < sites_grid; ++w) {
                    for (int x = 0; x < 6; ++x) sum[x] += h[(size_t)w * 6 + x];
                    mx = std::max(mx, h[(size_t)w * 6]);
                }
                fprintf(stderr, "[k_call_sites] %d wavefronts, %llu bubbles; ticks (10 ns) per wavefront: total %.0f (max %llu) = pop + load %.0f, strings %.0f, ranks + probes %.0f, groups %.0f\n",
                        sites_grid, sum[5], (double)sum[0] / sites_grid, mx, (double)sum[1] / sites_grid, (double)sum[2] / sites_grid, (double)sum[3] / sites_grid,
                        (double)sum[4] / sites_grid);
            }
            PF_HIP(hipMemcpyAsync(&hc, d_cnt, sizeof(hc), hipMemcpyDeviceToHost, st));
            PF_HIP(hipStreamSynchronize(st));
            if (hc.err & 16u) {
                if (++ks_attempt > 3 || hc.ks_need <= KS) { pf::CtxErr{ctx} = "pf_call_run: the room for a site string does not converge"; return PF_ERR_OVERFLOW; }
                KS = (hc.ks_need + 63u) & ~63u;
                amax(S->sites_ks, KS);
                hc.err = 0; hc.ks_need = 0;   // (the other bits are looked at when every string had room: a string cut short has no verdict)
                PF_HIP(hipMemsetAsync(&d_cnt->err, 0, 4, st));
                PF_HIP(hipMemsetAsync(&d_cnt->ks_need, 0, 4, st));
                --attempt;
                continue;
            }
            if (hc.err & 2u) { pf::CtxErr{ctx} = "CDBG::readCov(): a kmer of a site string can not found ."; return PF_ERR_MISSING_KMER; }
            if (hc.err & 4u) { pf::CtxErr{ctx} = "CDBG::PloidyEstimation(): a site string runs past the end of an aligned row (the reference terminates here: std::out_of_range from substr, src/CDBG.cpp:1478-1590)"; return PF_ERR_ARG; }
            if (hc.err & 64u) { pf::CtxErr{ctx} = "CCDBG::PloidyEstimation(): a site string does not start on a unitig of its bubble"; return PF_ERR_ARG; }
            if (hc.sv_head > sv_cap) {
                if (attempt >= 2) { pf::CtxErr{ctx} = "pf_call_run: site value pool does not converge"; return PF_ERR_OVERFLOW; }
                amax(S->sv_pool, hc.sv_head + hc.sv_head / 8 + 1024);
                continue;
            }
            amax(S->sv_pool, sv_cap);
            break;
        }
    } else {
        NEED(O.sv, 16);
        PF_HIP(hipStreamSynchronize(st));   // (n_called)
    }
    out->site_strings = hc.site_strings;
    ta("K-SITES done");

    out->n_called = n_called;
    O.nb = nb;
    O.cur = *out;
    for (int x = 0; x < 4; ++x) O.used[x] = heads[x];
    O.used[4] = hc.n_branching ? hc.sv_head : 0;
#undef NEED
    return PF_OK;
}

int pf_call_peek(pf_ctx *ctx, int lane, pf_call_bubble *bubbles, pf_bubble_result *results, uint64_t *sv_off, uint64_t bubble_cap, char *text,
                 pf_bubble_site *sites, uint8_t *groups, uint32_t *ilen, double *sv, const uint64_t cap[5], uint64_t used[6]) {
    if (!ctx || lane < 0 || lane >= PF_CALL_LANES || !used) return PF_ERR_ARG;
    CallState *S = ctx->call;
    if (!S) { pf::CtxErr{ctx} = "pf_call_peek: pf_call_align first"; return PF_ERR_ARG; }
    if (S->n_colors) { pf::CtxErr{ctx} = "pf_call_peek: the single-sample path's view (a colored site holds one value per colour and group)"; return PF_ERR_ARG; }
    const CallState::AlignOut &O = S->lane[lane];
    used[0] = O.nb;
    for (int x = 0; x < 5; ++x) used[x + 1] = O.used[x];
    if (!O.nb) return PF_OK;
    if (!bubbles && !results && !text && !sites && !groups && !ilen && !sv && !sv_off) return PF_OK;   // sizes only
    if (bubble_cap < O.nb || !cap) return PF_ERR_ARG;
    for (int x = 0; x < 5; ++x)
        if (cap[x] < O.used[x]) return PF_ERR_ARG;
    PF_HIP(hipSetDevice(ctx->device));
    if (bubbles) {
        std::vector<uint32_t> kept(O.nb);
        PF_HIP(hipMemcpy(kept.data(), S->kept.as<uint32_t>() + O.t0, (size_t)O.nb * 4, hipMemcpyDeviceToHost));
        std::vector<CallTask> ct(S->n_sides);
        if (S->n_sides) PF_HIP(hipMemcpy(ct.data(), S->ctask.p, (size_t)S->n_sides * sizeof(CallTask), hipMemcpyDeviceToHost));
        for (uint32_t j = 0; j < O.nb; ++j) {
            const CallTask &t = ct[kept[j]];
            pf_call_bubble &b = bubbles[j];
            b.entrance_ov = t.entrance_ov; b.exit_ov = t.exit_ov; b.strict = t.strict; b.n_inner = t.n_inner;
            for (int x = 0; x < 4; ++x) { b.inner[x] = t.inner[x]; b.cov[x] = t.cov[x]; }
            b.core_mean = t.core_mean; b.cov_sum = t.cov_sum;
        }
    }
    if (results) PF_HIP(hipMemcpy(results, O.res.p, (size_t)O.nb * sizeof(pf_bubble_result), hipMemcpyDeviceToHost));
    if (sv_off) PF_HIP(hipMemcpy(sv_off, O.sv_off.p, (size_t)O.nb * 8, hipMemcpyDeviceToHost));
    if (text && O.used[0]) PF_HIP(hipMemcpy(text, O.otext.p, O.used[0], hipMemcpyDeviceToHost));
    if (sites && O.used[1]) PF_HIP(hipMemcpy(sites, O.osites.p, O.used[1] * sizeof(pf_bubble_site), hipMemcpyDeviceToHost));
    if (groups && O.used[2]) PF_HIP(hipMemcpy(groups, O.ogroups.p, O.used[2], hipMemcpyDeviceToHost));
    if (ilen && O.used[3]) PF_HIP(hipMemcpy(ilen, O.oilen.p, O.used[3] * 4, hipMemcpyDeviceToHost));
    if (sv && O.used[4]) PF_HIP(hipMemcpy(sv, O.sv.p, O.used[4] * 8, hipMemcpyDeviceToHost));
    return PF_OK;
}

// The buffers pf_call_align_lane asks for on its first call for ranges of up to nb bubbles, taken NOW (a caller does this beside the
// load): a first pass then starts with its pools in place instead of two dozen hipMallocs, 20 ms at 5 M unitigs.  Sizes are the
// first-call formulas of pf_call_align_lane; whatever turns out too small there grows as before.
int pf_call_reserve_lanes(pf_ctx *ctx, uint64_t nb64, uint32_t complex_size, int n_lanes) {
    if (!ctx || nb64 == 0 || n_lanes < 1 || n_lanes > PF_CALL_LANES) return PF_ERR_ARG;
    CallState *S = state_of(ctx);
    PF_HIP(hipSetDevice(ctx->device));
    const uint32_t nb = (uint32_t)std::min<uint64_t>(nb64, 1u << 24);
    const char *oom = "pf_call_reserve: out of device memory";
#define NEED(buf, bytes) do { if (!(buf).ensure(bytes)) { pf::CtxErr{ctx} = oom; return PF_ERR_HIP; } } while (0)
    for (int lane = 0; lane < n_lanes; ++lane) {
        CallState::AlignWork &W = S->work[lane];
        NEED(W.counters, sizeof(CallCounters));
        NEED(W.btask, (size_t)nb * sizeof(pf_bubble_task));
        NEED(W.queues, (size_t)NQ * nb * 4);
        for (DevBuf *b : {&W.blist, &W.slist, &W.plist, &W.plist2, &W.klist, &W.klist_b, &W.tlist, &W.tlist4, &W.has}) NEED(*b, (size_t)nb * 4);
        NEED(W.stack_scr, stack_scratch_bytes() * (uint64_t)(ctx->n_cu * 8));
        const uint32_t depth_cap = std::max<uint32_t>(complex_size + 4, 16);
        const uint64_t paths_per_wave = ((256 * 8 + 256 * 4 + (10ull * depth_cap + 4) * 4) + 255) & ~255ull;
        NEED(W.paths_scr, paths_per_wave * (uint64_t)(ctx->n_cu * 16));
        NEED(W.pair_scr, PairGeom<PAIR_MAX>::scratch_bytes * (uint64_t)(ctx->n_cu * 12));
        NEED(W.bpath, ((size_t)4 * nb + (uint64_t)nb / 2 + 128ull * (ctx->n_cu * 16) + 1024) * sizeof(pf_bubble_path));
        NEED(W.ptext, (uint64_t)nb * FIRST_PATH_TEXT + (1u << 16));
        NEED(W.scan_tmp2, (size_t)nb / 4 * 4 + 4096);
        {   // K-SITES' tables for bubbles of up to 256 walks (single-sample)
            const uint64_t KS = (uint64_t)(2 * ctx->k + 64), rows_cap = 256;
            const uint64_t sites_per_wave = ((2 * rows_cap * KS + rows_cap * (4 + 4 + 4 + 1 + 1 + 8)) + 255) & ~255ull;
            NEED(W.sites_scr, sites_per_wave * (uint64_t)(ctx->n_cu * 16));
        }
        CallState::AlignOut &O = S->lane[lane];
        NEED(O.res, (size_t)nb * sizeof(pf_bubble_result));
        NEED(O.sv_off, (size_t)nb * 8);
        NEED(O.vc, (size_t)nb * 4);
        NEED(O.otext, (uint64_t)FIRST_ROW_TEXT * nb + (1u << 16));
        NEED(O.osites, ((uint64_t)FIRST_SITES * nb + 64) * sizeof(pf_bubble_site));
        NEED(O.ogroups, (uint64_t)FIRST_GROUPS * nb + 64);
        NEED(O.oilen, ((uint64_t)FIRST_ILEN * nb + 64) * 4);
        NEED(O.sv, ((uint64_t)nb / 4 + 1024ull * ctx->n_cu * 16 + 1024) * 8);
        if (lane != 0 && !W.stream) { PF_HIP(lane_stream_create(&W.stream, lane)); W.own_stream = true; }
        if (!W.side_stream) {
            PF_HIP(lane_stream_create(&W.side_stream, lane));
            PF_HIP(hipEventCreateWithFlags(&W.ev_prep, hipEventDisableTiming));
            PF_HIP(hipEventCreateWithFlags(&W.ev_paths, hipEventDisableTiming));
        }
        const int st = bubble_reserve(ctx, nb, lane);
        if (st != PF_OK) return st;
    }
#undef NEED
    return PF_OK;
}

int pf_call_reserve(pf_ctx *ctx, uint64_t nb64, uint32_t complex_size) { return pf_call_reserve_lanes(ctx, nb64, complex_size, 2); }

int pf_call_align(pf_ctx *ctx, uint64_t t0, uint64_t t1, uint32_t complex_size, double match, double mismatch, double gap,
                  pf_call_result *out) {
    return pf_call_align_lane(ctx, 0, t0, t1, complex_size, match, mismatch, gap, out);
}

// one batch, second half: K-TEXT of the bubbles pf_call_align left resident, into slab 0 or 1
// (any host thread: a stream, scratch and counters of its own, launch timing by place -- one pf_call_text_range at a time,
// beside at most one pf_call_align_lane on the OTHER lane)
static int text_work_of(pf_ctx *ctx, pf::CallState *S, int which, uint32_t nb) {
    pf::CallState::TextWork &T = S->text[which];
    if (!T.stream) {
        // highest priority: its short kernels go ahead of the alignment kernels of the other lanes, whose grids fill the device
        // for milliseconds -- the text has a PCIe copy and a file copy still before it
        int least = 0, greatest = 0;
        PF_HIP(hipDeviceGetStreamPriorityRange(&least, &greatest));
        PF_HIP(hipStreamCreateWithPriority(&T.stream, hipStreamNonBlocking, greatest));
    }
    NEED_TEXT(T.sizes, (size_t)N_INT * (nb + 1) * 4);
    NEED_TEXT(T.offs, ((size_t)N_INT * (nb + 1) + 1) * 8);
    NEED_TEXT(T.totals, 16 * 8);
    NEED_TEXT(T.tcounters, sizeof(CallCounters));
    size_t tmp2 = 0;
    hipcub::TransformInputIterator<uint64_t, Widen, const uint32_t *> wide(T.sizes.as<uint32_t>(), Widen());
    PF_HIP(hipcub::DeviceScan::ExclusiveSum(nullptr, tmp2, wide, T.offs.as<uint64_t>(), (int)((size_t)N_INT * (nb + 1)), T.stream));
    NEED_TEXT(T.tscan, tmp2);
    return PF_OK;
}

// what the first pf_call_text_range of a run would take (see pf_call_reserve): the text streams, the size tables of a piece of
// piece_bubbles bubbles and the text slabs (an estimate from k: rows of two to four aligned paths; grows as before when short)
int pf_call_reserve_text(pf_ctx *ctx, uint64_t piece_bubbles) {
    if (!ctx || piece_bubbles == 0) return PF_ERR_ARG;
    pf::CallState *S = state_of(ctx);
    PF_HIP(hipSetDevice(ctx->device));
    const uint32_t nb = (uint32_t)std::min<uint64_t>(piece_bubbles, 1u << 24);
    for (int which = 0; which < 2; ++which) { const int ts = text_work_of(ctx, S, which, nb); if (ts != PF_OK) return ts; }
    for (int slab = 0; slab < PF_CALL_SLABS; ++slab) NEED_TEXT(S->out[slab], (11ull * (uint64_t)ctx->k + 40) * nb);
    // K-TEXT's two passes once over no bubbles on each of its streams: what the first launch of a kernel this size pays on a stream
    // (scratch for its spills: 2.3 ms of a first piece's count pass) is paid here, beside the load
    for (int which = 0; which < 2; ++which) {
        pf::CallState::TextWork &T = S->text[which];
        FmtArgs fa;
        memset(&fa, 0, sizeof(fa));
        fa.sizes = T.sizes.as<uint32_t>(); fa.offs = T.offs.as<uint64_t>(); fa.cnt = T.tcounters.as<CallCounters>();
        if (S->n_colors) { k_call_format<false, true><<<1, FMT_BLOCK, 0, T.stream>>>(fa); k_call_format<true, true><<<1, FMT_BLOCK, 0, T.stream>>>(fa); }
        else { k_call_format<false, false><<<1, FMT_BLOCK, 0, T.stream>>>(fa); k_call_format<true, false><<<1, FMT_BLOCK, 0, T.stream>>>(fa); }
        PF_HIP(hipGetLastError());
        PF_HIP(hipStreamSynchronize(T.stream));
    }
    return PF_OK;
}

static int call_text_impl(pf_ctx *ctx, int lane, int slab, uint64_t first, uint64_t count, uint64_t var_count_base, pf_call_result *out, bool sizes_only) {
    if (!ctx || !out || slab < 0 || slab >= PF_CALL_SLABS || lane < 0 || lane >= PF_CALL_LANES) return PF_ERR_ARG;
    CallState *S = ctx->call;
    if (!S) return PF_ERR_ARG;
    CallState::AlignOut &O = S->lane[lane];
    if (first + count > O.nb) { pf::CtxErr{ctx} = "pf_call_text_range: range outside the aligned batch"; return PF_ERR_ARG; }
    if (S->n_colors && S->mt_format) { pf::CtxErr{ctx} = "pf_call_text_range: the -t > 1 format is the single-sample path's"; return PF_ERR_ARG; }
    if (!sizes_only)
        for (int s = 0; s < N_STREAMS; ++s) S->out_len[slab][s] = 0;
    *out = O.cur;
    out->n_called = 0;
    if (count == 0) return PF_OK;
    PF_HIP(hipSetDevice(ctx->device));
    const bool trace_stages = getenv("PF_TRACE_ALIGN") != nullptr;
    const auto t_enter = std::chrono::steady_clock::now();
    auto ta = [&](const char *what) {
        if (trace_stages) fprintf(stderr, "[pf_call_text]    %-34s %.2f ms\n", what, std::chrono::duration<double>(std::chrono::steady_clock::now() - t_enter).count() * 1e3);
    };
    const uint32_t nb = (uint32_t)count;
    const int which = sizes_only ? 0 : (slab & 1);
    { const int ts = text_work_of(ctx, S, which, nb); if (ts != PF_OK) return ts; }
    CallState::TextWork &T = S->text[which];
    hipStream_t st = T.stream;
    const uint64_t t0 = O.t0;
    // bubbles called inside the range: difference of the batch-wide running count (read with the sizes below: one wait)
    uint32_t vc_edge[2] = {0, 0};
    PF_HIP(hipMemcpyAsync(&vc_edge[1], O.vc.as<uint32_t>() + (first + count - 1), 4, hipMemcpyDeviceToHost, st));
    if (first) PF_HIP(hipMemcpyAsync(&vc_edge[0], O.vc.as<uint32_t>() + (first - 1), 4, hipMemcpyDeviceToHost, st));
    const char *oom = "pf_call_text: out of device memory";
#define NEED(buf, bytes) do { if (!(buf).ensure(bytes)) { pf::CtxErr{ctx} = oom; return PF_ERR_HIP; } } while (0)
    CallCounters *d_cnt = T.tcounters.as<CallCounters>();
    CallCounters hc;
    size_t tmp2 = 0;
    const size_t n_sizes = (size_t)N_INT * (nb + 1);
    hipcub::TransformInputIterator<uint64_t, Widen, const uint32_t *> wide(T.sizes.as<uint32_t>(), Widen());
    PF_HIP(hipcub::DeviceScan::ExclusiveSum(nullptr, tmp2, wide, T.offs.as<uint64_t>(), (int)n_sizes, st));
    ta("stream and size tables");
    PF_HIP(hipMemsetAsync(&d_cnt->allele[0], 0, 6 * 8, st));  // allele[4], core_cov, core_num
    FmtArgs fa;
    fa.ct = S->ctask.as<CallTask>(); fa.kept = S->kept.as<uint32_t>(); fa.t0 = t0; fa.j0 = (uint32_t)first; fa.nb = nb;
    fa.res = O.res.as<pf_bubble_result>();
    fa.otext = O.otext.as<char>(); fa.osites = O.osites.as<pf_bubble_site>(); fa.ogroups = O.ogroups.as<uint8_t>();
    fa.oilen = O.oilen.as<uint32_t>(); fa.sv_off = O.sv_off.as<uint64_t>(); fa.sv = O.sv.as<double>(); fa.vc = O.vc.as<uint32_t>();
    fa.vc_base = var_count_base; fa.mt = S->mt_format ? 1 : 0; fa.len = ctx->d_len; fa.sizes = T.sizes.as<uint32_t>(); fa.offs = T.offs.as<uint64_t>(); fa.cnt = d_cnt;
    for (int s = 0; s < N_INT; ++s) fa.out[s] = nullptr;
    fa.packed = S->pack_alignseq ? 1 : 0;
    size_t at = 0;
    ctx_begin_at(ctx, PF_K_CALL_FORMAT, st, &at);
    fa.n_colors = S->n_colors; fa.N = ctx->N; fa.k = ctx->k; fa.full = S->col_full.as<uint64_t>(); fa.ccov_sum = S->ccov_sum.as<uint64_t>();
    if (S->n_colors) k_call_format<false, true><<<(nb + 1 + FMT_BLOCK - 1) / FMT_BLOCK, FMT_BLOCK, 0, st>>>(fa);
    else k_call_format<false, false><<<(nb + 1 + FMT_BLOCK - 1) / FMT_BLOCK, FMT_BLOCK, 0, st>>>(fa);
    ctx_end_at(ctx, at, st);
    PF_HIP(hipcub::DeviceScan::ExclusiveSum(T.tscan.p, tmp2, wide, T.offs.as<uint64_t>(), (int)n_sizes, st));
    k_call_totals<<<1, 64, 0, st>>>(T.offs.as<uint64_t>(), T.sizes.as<uint32_t>(), nb, T.totals.as<uint64_t>());
    uint64_t totals[N_INT + 1] = {};
    PF_HIP(hipMemcpyAsync(totals, T.totals.p, N_INT * 8, hipMemcpyDeviceToHost, st));
    PF_HIP(hipMemcpyAsync(&hc, d_cnt, sizeof(hc), hipMemcpyDeviceToHost, st));
    PF_HIP(hipStreamSynchronize(st));
    out->n_called = vc_edge[1] - vc_edge[0];
    // what a stream takes in the slab: its text -- alignseq, when packed, its index and records instead
    const uint64_t packed_len = (fa.packed && totals[PF_OUT_ALIGNSEQ]) ? alnpack_index_bytes(nb) + totals[S_PACK] : 0;
    auto slab_len = [&](int s) { return (s == PF_OUT_ALIGNSEQ && fa.packed) ? packed_len : totals[s]; };
    out->alignseq_packed_len = packed_len;
    uint64_t all = 0;
    for (int s = 0; s < N_STREAMS; ++s) all += slab_len(s);
    if (sizes_only) {   // pf_call_text_sizes: the count pass alone
        for (int s = 0; s < N_STREAMS; ++s) out->text_len[s] = totals[s];
        for (int x = 0; x < 4; ++x) out->allele[x] = hc.allele[x];
        out->core_cov = hc.core_cov;
        out->core_num = hc.core_num;
        return PF_OK;
    }
    ta("sizes counted");
    NEED(S->out[slab], std::max<uint64_t>(all, 16));
    ta("slab taken");
    all = 0;
    for (int s = 0; s < N_STREAMS; ++s) {
        fa.out[s] = S->out[slab].as<char>() + all;
        S->out_off[slab][s] = all;
        S->out_len[slab][s] = slab_len(s);
        out->text_len[s] = totals[s];
        all += slab_len(s);
    }
    if (fa.packed) { fa.out[S_PACK] = fa.out[PF_OUT_ALIGNSEQ]; fa.out[PF_OUT_ALIGNSEQ] = nullptr; }
    ctx_begin_at(ctx, PF_K_CALL_FORMAT, st, &at);
    if (S->n_colors) k_call_format<true, true><<<(nb + FMT_BLOCK - 1) / FMT_BLOCK, FMT_BLOCK, 0, st>>>(fa);
    else k_call_format<true, false><<<(nb + FMT_BLOCK - 1) / FMT_BLOCK, FMT_BLOCK, 0, st>>>(fa);
    ctx_end_at(ctx, at, st);
    ctx_units(ctx, PF_K_CALL_FORMAT, nb);
    PF_HIP(hipGetLastError());
    // no wait for the write pass: the fetches of this slab wait for it on their own stream (text_ev), and the next piece's count
    // pass queues behind it on this one
    if (!S->text_ev[slab]) PF_HIP(hipEventCreateWithFlags(&S->text_ev[slab], hipEventDisableTiming));
    PF_HIP(hipEventRecord(S->text_ev[slab], st));
    // ... and neither does the caller, who may hand the lane to the next pf_call_align_lane at once: that call's stream waits here
    if (!O.read_ev[which]) PF_HIP(hipEventCreateWithFlags(&O.read_ev[which], hipEventDisableTiming));
    PF_HIP(hipEventRecord(O.read_ev[which], st));
    for (int x = 0; x < 4; ++x) out->allele[x] = hc.allele[x];
    out->core_cov = hc.core_cov;
    out->core_num = hc.core_num;
#undef NEED
    return PF_OK;
}

int pf_call_set_alignseq_packed(pf_ctx *ctx, int on) {
    if (!ctx) return PF_ERR_ARG;
    CallState *S = state_of(ctx);
    if (!S) return PF_ERR_HIP;
    S->pack_alignseq = on != 0;
    return PF_OK;
}

int pf_call_text_range_lane(pf_ctx *ctx, int lane, int slab, uint64_t first, uint64_t count, uint64_t var_count_base, pf_call_result *out) {
    return call_text_impl(ctx, lane, slab, first, count, var_count_base, out, false);
}

int pf_call_text_sizes(pf_ctx *ctx, int lane, uint64_t first, uint64_t count, uint64_t var_count_base, pf_call_result *out) {
    return call_text_impl(ctx, lane, 0, first, count, var_count_base, out, true);
}

int pf_call_text_range(pf_ctx *ctx, int slab, uint64_t first, uint64_t count, uint64_t var_count_base, pf_call_result *out) {
    return pf_call_text_range_lane(ctx, 0, slab, first, count, var_count_base, out);
}

int pf_call_text(pf_ctx *ctx, int slab, uint64_t var_count_base, pf_call_result *out) {
    if (!ctx || !ctx->call) return PF_ERR_ARG;
    const int st = pf_call_text_range(ctx, slab, 0, ctx->call->lane[0].nb, var_count_base, out);
    if (st == PF_OK) out->n_called = ctx->call->lane[0].cur.n_called;
    return st;
}

int pf_call_run(pf_ctx *ctx, int slab, uint64_t t0, uint64_t t1, uint64_t var_count_base, uint32_t complex_size, double match,
                double mismatch, double gap, pf_call_result *out) {
    const int st = pf_call_align(ctx, t0, t1, complex_size, match, mismatch, gap, out);
    if (st != PF_OK) return st;
    return pf_call_text(ctx, slab, var_count_base, out);
}

int pf_call_fetch(pf_ctx *ctx, int slab, int stream, char *dst, uint64_t len) {
    if (!ctx || !ctx->call || slab < 0 || slab >= PF_CALL_SLABS || stream < 0 || stream >= N_STREAMS) return PF_ERR_ARG;
    CallState *S = ctx->call;
    if (len > S->out_len[slab][stream] || (len && !dst)) return PF_ERR_ARG;
    if (len == 0) return PF_OK;
    // its own stream, and no context state written: safe beside a pf_call_run on the other slab
    if (hipSetDevice(ctx->device) != hipSuccess) return PF_ERR_HIP;
    if (!S->copy_stream) return PF_ERR_HIP;
    if (S->text_ev[slab] && hipStreamWaitEvent(S->copy_stream, S->text_ev[slab], 0) != hipSuccess) return PF_ERR_HIP;
    if (hipMemcpyAsync(dst, S->out[slab].as<char>() + S->out_off[slab][stream], (size_t)len, hipMemcpyDeviceToHost, S->copy_stream) != hipSuccess) return PF_ERR_HIP;
    if (hipStreamSynchronize(S->copy_stream) != hipSuccess) return PF_ERR_HIP;
    return PF_OK;
}

// all ten streams of a slab, one after the other into dst (stream s at off[s] = sum of the lengths before it): ten copies in
// flight, one wait
int pf_call_fetch_slab(pf_ctx *ctx, int slab, char *dst, const uint64_t *len) {
    if (!ctx || !ctx->call || slab < 0 || slab >= PF_CALL_SLABS || !len) return PF_ERR_ARG;
    CallState *S = ctx->call;
    if (hipSetDevice(ctx->device) != hipSuccess) return PF_ERR_HIP;
    if (!S->copy_stream) return PF_ERR_HIP;
    if (S->text_ev[slab] && hipStreamWaitEvent(S->copy_stream, S->text_ev[slab], 0) != hipSuccess) return PF_ERR_HIP;
    uint64_t at = 0;
    bool whole = true;
    for (int s = 0; s < N_STREAMS; ++s) {
        if (len[s] > S->out_len[slab][s] || (len[s] && !dst)) return PF_ERR_ARG;
        whole = whole && len[s] == S->out_len[slab][s];
        at += len[s];
    }
    size_t tl_at = (size_t)-1;
    (void)ctx_begin_at(ctx, PF_K_COPY_TEXT, S->copy_stream, &tl_at);
    if (whole) {   // the slab as it lies: one copy
        if (at && hipMemcpyAsync(dst, S->out[slab].p, (size_t)at, hipMemcpyDeviceToHost, S->copy_stream) != hipSuccess) return PF_ERR_HIP;
    } else {
        at = 0;
        for (int s = 0; s < N_STREAMS; ++s) {
            if (len[s] && hipMemcpyAsync(dst + at, S->out[slab].as<char>() + S->out_off[slab][s], (size_t)len[s], hipMemcpyDeviceToHost, S->copy_stream) != hipSuccess)
                return PF_ERR_HIP;
            at += len[s];
        }
    }
    ctx_end_at(ctx, tl_at, S->copy_stream);
    if (hipStreamSynchronize(S->copy_stream) != hipSuccess) return PF_ERR_HIP;
    return PF_OK;
}

// A slab fetched in byte ranges of its packed layout (stream after stream, as pf_call_fetch_slab delivers it), without waiting:
// range number `slot` (0 or 1, alternating) is complete when pf_call_fetch_wait(slot) returns -- the caller copies range i into its
// files while range i + 1 crosses PCIe.
int pf_call_fetch_range(pf_ctx *ctx, int slab, uint64_t first_byte, char *dst, uint64_t len, int slot) {
    if (!ctx || !ctx->call || slab < 0 || slab >= PF_CALL_SLABS || slot < 0 || slot > 1 || (len && !dst)) return PF_ERR_ARG;
    CallState *S = ctx->call;
    uint64_t all = 0;
    for (int s = 0; s < N_STREAMS; ++s) all += S->out_len[slab][s];
    if (first_byte + len > all) return PF_ERR_ARG;
    if (hipSetDevice(ctx->device) != hipSuccess || !S->copy_stream) return PF_ERR_HIP;
    if (!S->fetch_ev[slot] && hipEventCreateWithFlags(&S->fetch_ev[slot], hipEventDisableTiming) != hipSuccess) return PF_ERR_HIP;
    if (S->text_ev[slab] && hipStreamWaitEvent(S->copy_stream, S->text_ev[slab], 0) != hipSuccess) return PF_ERR_HIP;
    size_t tl_at = (size_t)-1;
    (void)ctx_begin_at(ctx, PF_K_COPY_TEXT, S->copy_stream, &tl_at);
    if (len && hipMemcpyAsync(dst, S->out[slab].as<char>() + first_byte, (size_t)len, hipMemcpyDeviceToHost, S->copy_stream) != hipSuccess) return PF_ERR_HIP;
    ctx_end_at(ctx, tl_at, S->copy_stream);
    if (hipEventRecord(S->fetch_ev[slot], S->copy_stream) != hipSuccess) return PF_ERR_HIP;
    return PF_OK;
}

int pf_call_fetch_wait(pf_ctx *ctx, int slot) {
    if (!ctx || !ctx->call || slot < 0 || slot > 1 || !ctx->call->fetch_ev[slot]) return PF_ERR_ARG;
    if (hipSetDevice(ctx->device) != hipSuccess) return PF_ERR_HIP;
    return hipEventSynchronize(ctx->call->fetch_ev[slot]) == hipSuccess ? PF_OK : PF_ERR_HIP;
}

}  // extern "C"
