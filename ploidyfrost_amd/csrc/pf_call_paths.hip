// The resident calling pipeline (pf_call.hip has the overview), device side: K-PATHS: k_call_paths.
#include "pf_call_kernels.hpp"

namespace pf_call {

// ---------------------------------------------------------------------------------------------------------------------
// K-PATHS


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

// K-PATHS' list keys: the alignment queues 0 .. NQ-1, then K-STACK's list
constexpr uint32_t PK_STACK = NQ, PK_NONE = 0xFFFFFFFFu;

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
        uint32_t c_off = (uint32_t)offsetof(CallCounters, n_stack_b);
        c_off = key < (uint32_t)NQ ? (uint32_t)offsetof(CallCounters, q_n) + 4u * key : c_off;
        unsigned int *counter = reinterpret_cast<unsigned int *>(reinterpret_cast<char *>(a.cnt) + c_off);
        uint64_t l_at = (uint64_t)(uintptr_t)a.klist;
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
__global__ __launch_bounds__(64, 4) void k_call_paths(PathArgs a) {
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
        const uint32_t lmax = wo.lmax;
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

// the forms pf_call.hip launches
template __global__ void k_call_paths<false>(PathArgs);
template __global__ void k_call_paths<true>(PathArgs);

}  // namespace pf_call
