// What the translation units of the resident calling pipeline share (pf_call.hip, the host side and the C-ABI; pf_call_scan.hip,
// pf_call_tiers.hip, pf_call_paths.hip, pf_call_sites.hip, pf_call_text.hip, one per stage, the kernels): the per-context state, the
// argument blocks the host fills and the kernels read, the work-list append every stage uses, and the kernels' declarations -- a
// kernel is defined (and, a template, instantiated) in the file of its stage and launched from pf_call.hip through its host stub.
// Internal to those six files.
#pragma once

#include <hip/hip_runtime.h>

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

namespace pf_call {

constexpr uint8_t B_PLUS = 0x01, B_MINUS = 0x02, B_STRICT_M = 0x08, B_STRICT_P = 0x10, B_COMPLEX_M = 0x20, B_COMPLEX_P = 0x40;
constexpr int N_STREAMS = PF_CALL_STREAMS;
constexpr int N_INT = N_STREAMS + 1;   // size / offset tables: the ten streams + the packed form of alignseq (pf_alnpack.hpp)
constexpr int S_PACK = N_STREAMS;
// first-pass pool sizes per bubble of a range (pf_call_align_lane, pf_call_reserve_lanes): bytes of aligned rows, sites, group bytes,
// indel lengths, bytes of path text
constexpr uint32_t FIRST_ROW_TEXT = 384, FIRST_SITES = 4, FIRST_GROUPS = 12, FIRST_ILEN = 2, FIRST_PATH_TEXT = 64;
// work lists of a batch: K-BUBBLE's queues (heavy and light per size class), then the three lists of the other kernels
constexpr int NQ = 2 * (kBubLdsClasses + 1);
constexpr int KEY_BRANCHING = NQ, KEY_SNP = NQ + 1, KEY_PAIR = NQ + 2, KEY_PAIR2 = NQ + 3, KEY_STACK = NQ + 4, KEY_NONE = NQ + 5;

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

}  // namespace pf_call

namespace pf {
using namespace pf_call;

struct CallState {
    // T1 state
    DevBuf flags, plus, minus;
    bool have_state = false;
    // C1 results, one slot per unitig (two per unitig for a database without canonical counting)
    DevBuf cov_sum, cov_min, cov_miss;
    bool have_cov = false, per_strand = false;
    // colored path (pf_call_set_colours): cutoffs per colour, the colour sets the calling phase asks about -- per unitig the set of
    // colours on every k-mer (col_words 64-bit words) and UnitigColors::size(); for a colour on part of a unitig one bit per k-mer (reference orientation):
    // entries part_first[u] .. part_first[u + 1] = {colour, first word in part_bits} -- and K-COV-C's results, colour-major
    uint32_t n_colors = 0, col_words = 1;
    DevBuf col_low, col_up, col_full, col_size, part_first, part_colour, part_word, part_bits;
    DevBuf ccov_sum, ccov_min, ccov_max, ccov_miss;
    std::atomic<uint32_t> sites_ks{0};   // K-SITES: room per site string once a launch asked for more than 2k + 64
    bool pack_alignseq = false;   // pf_call_set_alignseq_packed
    bool pack_numeric = false;    // pf_call_set_numeric_packed
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
        DevBuf counters, btask, bpath, ptext, queues, blist, slist, plist, plist2, klist, klist_b, stack_scr, pair_scr, pair_scr2, has, scan_tmp, scan_tmp2, paths_scr, sites_scr;
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
    // pf_call_set_numeric_packed: what the fetches copy is then a second buffer per slab -- the numeric streams at four bits a
    // character, alignseq as it lies in `out`, a 16-byte tail with the flag word -- and out_len / out_off speak of that one; the text
    // stays where K-TEXT wrote it (txt_off / txt_len: pf_call_fetch_text, the way out for a stream with other characters)
    DevBuf outp[PF_CALL_SLABS];
    const char *fetch_base[PF_CALL_SLABS] = {};
    bool nib[PF_CALL_SLABS] = {};
    uint64_t txt_len[PF_CALL_SLABS][N_STREAMS] = {}, txt_off[PF_CALL_SLABS][N_STREAMS] = {};
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
            for (DevBuf *b : {&w.counters, &w.btask, &w.bpath, &w.ptext, &w.queues, &w.blist, &w.slist, &w.plist, &w.plist2, &w.klist, &w.klist_b, &w.stack_scr,
                              &w.pair_scr, &w.pair_scr2, &w.has, &w.scan_tmp, &w.scan_tmp2, &w.paths_scr, &w.sites_scr, &w.mlist, &w.paths_big_scr, &w.walk_off, &w.walk_pool})
                b->release();
            if (w.own_stream && w.stream) (void)hipStreamDestroy(w.stream);
            w.stream = nullptr; w.own_stream = false;
            if (w.side_stream) { (void)hipStreamDestroy(w.side_stream); w.side_stream = nullptr; }
            if (w.ev_prep) { (void)hipEventDestroy(w.ev_prep); w.ev_prep = nullptr; }
            if (w.ev_paths) { (void)hipEventDestroy(w.ev_paths); w.ev_paths = nullptr; }
        }
        for (DevBuf &b : out) b.release();
        for (DevBuf &b : outp) b.release();
        if (copy_stream) { (void)hipStreamDestroy(copy_stream); copy_stream = nullptr; }
        for (hipEvent_t &e : fetch_ev)
            if (e) { (void)hipEventDestroy(e); e = nullptr; }
        for (hipEvent_t &e : text_ev)
            if (e) { (void)hipEventDestroy(e); e = nullptr; }
    }
};
}  // namespace pf

namespace pf_call {

// colour c in the set of unitig u: sets of `words` 64-bit words per unitig, bit c % 64 of word c / 64
__device__ inline bool colour_in(const uint64_t *__restrict__ full, uint32_t words, uint32_t u, uint32_t c) {
    return (full[(size_t)u * words + (c >> 6)] >> (c & 63)) & 1;
}

inline CallState *state_of(pf_ctx *ctx) {
    if (!ctx->call) {
        ctx->call = new CallState();
        // copies of finished text slabs run beside the next batch's kernels
        if (hipStreamCreateWithFlags(&ctx->call->copy_stream, hipStreamNonBlocking) != hipSuccess) { (void)hipGetLastError(); ctx->call->copy_stream = nullptr; }
    }
    return ctx->call;
}

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
    const uint64_t *full, *size_total;   // full: cwords 64-bit words per unitig (colour_in)
    uint32_t cwords;
};

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

// appends `val` to one of the lists chosen by key (0 .. NQ-1 K-BUBBLE's queues, KEY_BRANCHING, KEY_SNP, KEY_PAIR; KEY_NONE =
// nowhere): one atomic per key and wave
struct CallLists {
    uint32_t *queues;   // NQ lists of nb entries
    uint32_t *blist, *slist, *plist, *plist2, *klist, *klist_b;
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
        uint32_t c_off = (uint32_t)offsetof(CallCounters, n_stack);
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
            else L.klist[at] = val;
        }
    }
}

// The same for a block of four wavefronts (every thread of the block must call it): the counters of all lists share a cache line or two,
// and atomics on one line queue one behind the other -- a wavefront's three or four were most of K-PREP's launch.  The wavefronts leave
// their counts per key in LDS, one thread per key adds the block's total, every lane takes its place behind the wavefronts before its own.
constexpr int N_KEYS = NQ + 5;
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
            uint32_t c_off = (uint32_t)offsetof(CallCounters, n_stack);
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
        else L.klist[at] = val;
    }
    __syncthreads();   // (the tables may be used again by the caller's next call)
}

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
    CallCounters *cnt;
};

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
    CallCounters *cnt;
};

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
    CallCounters *cnt;
};

constexpr uint32_t MAX_PATHS = 255;        // walks of one bubble whose tables fit LDS
constexpr uint32_t PATHS_BIG = 65535;      // ... in the second launch's global tables

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
    const uint8_t *unread;    // per colour: its database is never looked up (written without canonical counting)
    const uint32_t *clow, *cup;
    const uint64_t *full;
    uint32_t cwords;          // 64-bit words of a colour set: (n_colors + 63) / 64
    const uint32_t *part_first, *part_colour;
    const uint64_t *part_word, *part_bits;
    const uint32_t *walk_pool;
    const uint64_t *walk_off;
    const uint64_t *seq, *off;
    const uint32_t *len;
};

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
    uint32_t n_colors, N, cwords;
    int k;
    const uint64_t *full, *ccov_sum;
};

// (a wavefront a block: K-TEXT's wavefronts share nothing, and a block of two holds its LDS stage until the slower one is done:
// a launch 0.177 -> 0.170 ms, a step 20.3 -> 19.95 ms at configs[2])
constexpr uint32_t FMT_BLOCK = 64;

// K-NIB: the numeric streams of a text piece at four bits a character (pf_call_set_numeric_packed)
constexpr int NIB_STREAMS = N_STREAMS - 1;   // all but alignseq
struct NibArgs {
    const char *src[NIB_STREAMS];
    uint8_t *dst[NIB_STREAMS];
    uint64_t len[NIB_STREAMS];
    uint64_t unit0[NIB_STREAMS + 1];   // units of 16 characters before each stream's
    uint32_t stream[NIB_STREAMS];      // which stream (bit of the flag word)
    uint32_t *flag;                    // bit s: stream s holds a character outside the sixteen
};

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

// the kernels, by stage
__global__ void k_call_count_sides(const uint8_t *, uint32_t, uint32_t *);   // pf_call_scan.hip
template <bool COLORED> __global__ void k_call_sides(ScanArgs);   // pf_call_scan.hip
__global__ void k_call_pending(ResolveArgs);   // pf_call_scan.hip
__global__ void k_call_resolve(ResolveArgs);   // pf_call_scan.hip
__global__ void k_call_prep(PrepArgs);   // pf_call_tiers.hip
__global__ void k_call_snp(SnpArgs);   // pf_call_tiers.hip
__global__ void k_call_pair2_reroute(PairArgs);   // pf_call_tiers.hip
template <int NMAX, bool INTEGRAL> __global__ void k_call_pair(PairArgs);   // pf_call_tiers.hip
__global__ void k_call_stack(StackArgs);   // pf_call_tiers.hip
template <bool BIG> __global__ void k_call_paths(PathArgs);   // pf_call_paths.hip
template <bool COLORED> __global__ void k_call_sites(SiteArgs);   // pf_call_sites.hip
__global__ void k_call_has(const pf_bubble_result *, uint32_t, uint32_t *);   // pf_call_text.hip
template <bool W, bool COLORED> __global__ void k_call_format(FmtArgs);   // pf_call_text.hip
__global__ void k_call_totals(const uint64_t *, const uint32_t *, uint32_t, uint64_t *);   // pf_call_text.hip
__global__ void k_sb_count(SbArgs, uint32_t *);   // pf_call_text.hip
template <bool W> __global__ void k_sb_format(SbArgs);   // pf_call_text.hip
__global__ void k_format_doubles(const double *, uint64_t, char *, uint8_t *);   // pf_call_text.hip
__global__ void k_text_nibbles(NibArgs);   // pf_call_text.hip

}  // namespace pf_call
