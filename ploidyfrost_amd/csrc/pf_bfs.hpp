// K-BFS: one wavefront per candidate entrance.  Device restatement of the pure (topology + id
// only) part of CDBG::extractSuperBubble_ptr (reference src/CDBG.cpp:253-372) plus the
// structural "strict" test of setNoBubble_ptr (src/CDBG.cpp:765-782).
//
// The traversal itself is sequential (a LIFO of vertices whose predecessors are all
// visited); what the 64 lanes parallelise is
//   * the CSR loads: lanes 0-3 fetch the successor row of the popped vertex, lanes 0-15 then
//     fetch the predecessor rows of those (up to) four successors in one gather,
//   * every membership test on the `seen` table (keyed by unitig id, as the reference's
//     state_map / strand_map are) by compare + ballot,
//   * the "only the exit is left" test over the whole table, the strict test and the
//     record write-out.
// The tables live in LDS (Store = per-wave LDS slices, CAP entries); a candidate that outgrows
// them is deferred to the same code running over per-wave global scratch.
#pragma once
#include "pf_device_common.hpp"
#include "ploidyfrost_hip.h"

namespace pf {

constexpr uint8_t BFS_DEFERRED = 0xFF;   // record placeholder: rerun in the big tier
constexpr uint8_t BFS_TOO_LARGE = 0xFE;  // outgrew the big tier as well

struct BfsStore {
    uint32_t *ent;   // seen table: oriented vertex, first-seen order
    uint8_t *meta;   // per entry: bits 0-1 state (1 visited, 2 seen), bit 2 recorded strand (1 = '+')
    uint32_t *todo;  // LIFO
    uint32_t *cyc;   // cycle set (insertion order, deduplicated)
    uint32_t cap;
    // early notice to the caller's walkers (pf_bfs_live_deferred): a traversal that reaches hint_at vertices is entered into the
    // live list at once -- entrance << 32 | hint_tag -- while it goes on here; nullptr = no notices
    unsigned long long *live = nullptr;
    unsigned int *n_live = nullptr;
    uint32_t live_cap = 0, hint_at = 0, hint_tag = 0;
};

struct BfsResult {
    uint32_t exit_ov;
    uint32_t n_seen, n_cyc;
    uint8_t outcome, flag_cycle, flag_tip, strict;
    bool overflow;
    bool hinted;   // the live list has this traversal already
};

// lane 0 writes the tables, every lane reads them: order the wave's memory traffic
__device__ inline void wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
    __builtin_amdgcn_wave_barrier();
}

// index of the entry whose unitig id equals id(ov), or -1; all lanes return the same value
__device__ inline int bfs_find(const BfsStore &st, uint32_t n, uint32_t ov) {
    const int lane = lane_id();
    const uint32_t want = ov >> 1;
    for (uint32_t base = 0; base < n; base += WAVE) {
        const uint32_t i = base + lane;
        const bool hit = i < n && (st.ent[i] >> 1) == want;
        const unsigned long long m = __ballot(hit);
        if (m) return (int)(base + __ffsll((long long)m) - 1);
    }
    return -1;
}

__device__ inline bool bfs_cyc_add(const BfsStore &st, uint32_t &n_cyc, uint32_t ov) {
    const int lane = lane_id();
    for (uint32_t base = 0; base < n_cyc; base += WAVE) {
        const uint32_t i = base + lane;
        if (__ballot(i < n_cyc && st.cyc[i] == ov)) return true;
    }
    if (n_cyc >= st.cap) return false;
    if (lane == 0) st.cyc[n_cyc] = ov;
    n_cyc++;
    return true;
}

// All lanes of the wave call this with the same arguments; the result is wave-uniform.
__device__ inline BfsResult bfs_traverse(const uint32_t *__restrict__ succ, const uint32_t *__restrict__ pred,
                                         const BfsStore &st, uint32_t s) {
    const int lane = lane_id();
    BfsResult r;
    r.exit_ov = NONE;
    r.n_seen = 0;
    r.n_cyc = 0;
    r.outcome = PF_BFS_NONE;
    r.flag_cycle = r.flag_tip = r.strict = 0;
    r.overflow = false;
    r.hinted = false;

    uint32_t n = 0, top = 0, n_cyc = 0;
    bool cyc_flag = false, tip_flag = false;
    if (lane == 0) {
        st.ent[0] = s;
        st.meta[0] = 0;  // no state yet (src/CDBG.cpp:265-266: pushed, not yet in state_map)
        st.todo[0] = s;
    }
    n = 1;
    top = 1;
    wave_sync();

    while (top > 0) {
        const uint32_t v = st.todo[top - 1];
        top--;
        // state_map[id(v)] = visited; strand_map[id(v)] = v.strand      (:271-272)
        {
            const int e = bfs_find(st, n, v);  // always present
            if (lane == 0) st.meta[e] = (uint8_t)(1 | (((v & 1) == 0) ? 4 : 0));
        }
        wave_sync();
        // CSR: successor row of v, then predecessor rows of its successors
        const uint32_t my_succ = lane < 4 ? succ[(size_t)v * 4 + lane] : NONE;
        const uint32_t sb = __shfl(my_succ, lane >> 2, WAVE);
        const uint32_t my_pred = (lane < 16 && sb != NONE) ? pred[(size_t)sb * 4 + (lane & 3)] : NONE;
        if (__ballot(my_succ != NONE) == 0) {
            tip_flag = true;  // :273-276
        } else {
            for (int b = 0; b < 4; ++b) {
                const uint32_t u = read_lane(my_succ, b);
                if (u == NONE) continue;
                if (u == s) {  // :281-287
                    cyc_flag = true;
                    if (!bfs_cyc_add(st, n_cyc, s) || !bfs_cyc_add(st, n_cyc, v)) { r.overflow = true; return r; }
                    continue;
                }
                // state_map and the seen table hold the same unitig ids once the entrance has been
                // popped (it is popped first), so "in state_map" == "has an entry".
                int e = bfs_find(st, n, u);
                const uint8_t um = e >= 0 ? st.meta[e] : 0;
                if (e < 0 || (um & 3) != 1) {
                    uint8_t new_meta;
                    if (e < 0) {  // :290-294  first sighting: append, record strand
                        if (n >= st.cap) { r.overflow = true; return r; }
                        e = (int)n;
                        if (lane == 0) st.ent[n] = u;
                        n++;
                        new_meta = (uint8_t)(2 | (((u & 1) == 0) ? 4 : 0));
                        if (st.live && n == st.hint_at) {
                            r.hinted = true;
                            if (lane == 0) {
                                const unsigned int d = atomicAdd(st.n_live, 1u);
                                if (d < st.live_cap)
                                    __hip_atomic_store(&st.live[d], ((unsigned long long)s << 32) | (unsigned long long)st.hint_tag, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
                            }
                        }
                    } else {  // :295-303  seen before: strand must agree
                        if (((um >> 2) & 1) != ((u & 1) == 0 ? 1u : 0u)) {
                            cyc_flag = true;
                            if (!bfs_cyc_add(st, n_cyc, u) || !bfs_cyc_add(st, n_cyc, v)) { r.overflow = true; return r; }
                        }
                        new_meta = (uint8_t)(2 | (um & 4));
                    }
                    if (lane == 0) st.meta[e] = new_meta;  // :304
                    wave_sync();
                    bool all_pred = true;  // :305-325
                    for (int j = 0; j < 4; ++j) {
                        const uint32_t p = read_lane(my_pred, b * 4 + j);
                        if (p == NONE) continue;
                        const int pe = bfs_find(st, n, p);
                        const uint8_t pm = pe >= 0 ? st.meta[pe] : 0;
                        if (pe >= 0 && (pm & 3) != 0) {
                            if ((pm & 3) != 1) all_pred = false;
                            if (((pm >> 2) & 1) != ((p & 1) == 0 ? 1u : 0u)) {
                                cyc_flag = true;
                                if (!bfs_cyc_add(st, n_cyc, u) || !bfs_cyc_add(st, n_cyc, p)) { r.overflow = true; return r; }
                            }
                        } else {
                            all_pred = false;
                        }
                    }
                    if (all_pred) {  // :326-327
                        if (top >= st.cap) { r.overflow = true; return r; }
                        if (lane == 0) st.todo[top] = u;
                        top++;
                    }
                } else {  // :329-334
                    cyc_flag = true;
                    if (!bfs_cyc_add(st, n_cyc, v) || !bfs_cyc_add(st, n_cyc, u)) { r.overflow = true; return r; }
                }
                wave_sync();
            }
        }
        if (top == 1) {  // :337-371
            const uint32_t t0 = st.todo[0];
            bool pending = false;
            for (uint32_t base = 0; base < n; base += WAVE) {
                const uint32_t i = base + lane;
                const bool bad = i < n && st.ent[i] != t0 && (st.meta[i] & 3) == 2;
                if (__ballot(bad)) { pending = true; break; }
            }
            if (!pending) {
                r.exit_ov = t0;
                const uint32_t ts = lane < 4 ? succ[(size_t)t0 * 4 + lane] : NONE;
                const bool back = __ballot(ts == s) != 0;
                if (back) r.outcome = PF_BFS_CYCLE_EXIT;
                else if (cyc_flag || tip_flag) r.outcome = PF_BFS_REJECT;
                else r.outcome = PF_BFS_ACCEPT;
                break;
            }
        }
    }
    r.n_seen = n;
    r.n_cyc = n_cyc;
    r.flag_cycle = cyc_flag;
    r.flag_tip = tip_flag;
    // strict structural test (src/CDBG.cpp:765-782), evaluated lane-parallel over the table
    if (r.outcome == PF_BFS_ACCEPT && n >= 4 && n <= 6) {
        const uint32_t t = r.exit_ov;
        bool ok = true;
        if ((uint32_t)lane < n) {
            const uint32_t w = st.ent[lane];
            if (w != s && w != t) {
                const uint4 pr = *reinterpret_cast<const uint4 *>(pred + (size_t)w * 4);
                const uint4 sr = *reinterpret_cast<const uint4 *>(succ + (size_t)w * 4);
                const uint32_t pa[4] = {pr.x, pr.y, pr.z, pr.w}, sa[4] = {sr.x, sr.y, sr.z, sr.w};
                int din = 0, dout = 0;
                uint32_t fp = NONE, fs = NONE;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    if (pa[j] != NONE) { if (din == 0) fp = pa[j]; din++; }
                    if (sa[j] != NONE) { if (dout == 0) fs = sa[j]; dout++; }
                }
                ok = din == 1 && dout == 1 && (fp >> 1) == (s >> 1) && (fs >> 1) == (t >> 1);
            }
        }
        r.strict = __ballot(!ok) == 0 ? 1 : 0;
    }
    return r;
}

// ---- thread tier: one THREAD per candidate ------------------------------------------------------------------------------------
// Most traversals see four to six vertices (a bi-allelic bubble: s, two inner unitigs, t).  Their cost is not arithmetic but
// a chain of dependent CSR loads, so a wavefront per candidate keeps one chain in flight where it could keep 64.  This tier
// runs the same traversal with one thread per candidate over tables of BFS_THREAD_CAP entries in LDS (entry-major, thread-minor:
// lane i always hits bank i); whatever outgrows them is handed to the wavefront tier above, unchanged.
constexpr uint32_t BFS_THREAD_CAP = 8;

struct BfsThreadStore {   // strided views into the block's LDS arrays
    uint32_t *ent, *todo, *cyc;
    uint8_t *meta;
    uint32_t stride;      // threads per block
    __device__ inline uint32_t &E(uint32_t i) const { return ent[i * stride]; }
    __device__ inline uint32_t &T(uint32_t i) const { return todo[i * stride]; }
    __device__ inline uint32_t &C(uint32_t i) const { return cyc[i * stride]; }
    __device__ inline uint8_t &M(uint32_t i) const { return meta[i * stride]; }
};

__device__ inline int bfs_thread_find(const BfsThreadStore &st, uint32_t n, uint32_t ov) {
    const uint32_t want = ov >> 1;
    for (uint32_t i = 0; i < n; ++i)
        if ((st.E(i) >> 1) == want) return (int)i;
    return -1;
}
__device__ inline bool bfs_thread_cyc_add(const BfsThreadStore &st, uint32_t &n_cyc, uint32_t ov) {
    for (uint32_t i = 0; i < n_cyc; ++i)
        if (st.C(i) == ov) return true;
    if (n_cyc >= BFS_THREAD_CAP) return false;
    st.C(n_cyc++) = ov;
    return true;
}

__device__ inline BfsResult bfs_traverse_thread(const uint32_t *__restrict__ succ, const uint32_t *__restrict__ pred,
                                                const BfsThreadStore &st, uint32_t s) {
    BfsResult r;
    r.exit_ov = NONE;
    r.n_seen = 0;
    r.n_cyc = 0;
    r.outcome = PF_BFS_NONE;
    r.flag_cycle = r.flag_tip = r.strict = 0;
    r.overflow = false;
    uint32_t n = 1, top = 1, n_cyc = 0;
    bool cyc_flag = false, tip_flag = false;
    st.E(0) = s;
    st.M(0) = 0;  // no state yet (src/CDBG.cpp:265-266: pushed, not yet in state_map)
    st.T(0) = s;
#define PF_BFS_OVER() do { r.overflow = true; return r; } while (0)
    while (top > 0) {
        const uint32_t v = st.T(--top);
        {   // state_map[id(v)] = visited; strand_map[id(v)] = v.strand      (:271-272)
            const int e = bfs_thread_find(st, n, v);
            st.M((uint32_t)e) = (uint8_t)(1 | (((v & 1) == 0) ? 4 : 0));
        }
        const uint4 sr = *reinterpret_cast<const uint4 *>(succ + (size_t)v * 4);
        const uint32_t row[4] = {sr.x, sr.y, sr.z, sr.w};
        if (sr.x == NONE && sr.y == NONE && sr.z == NONE && sr.w == NONE) {
            tip_flag = true;  // :273-276
        } else {
            for (int b = 0; b < 4; ++b) {
                const uint32_t u = row[b];
                if (u == NONE) continue;
                if (u == s) {  // :281-287
                    cyc_flag = true;
                    if (!bfs_thread_cyc_add(st, n_cyc, s) || !bfs_thread_cyc_add(st, n_cyc, v)) PF_BFS_OVER();
                    continue;
                }
                int e = bfs_thread_find(st, n, u);
                const uint8_t um = e >= 0 ? st.M((uint32_t)e) : 0;
                if (e < 0 || (um & 3) != 1) {
                    uint8_t new_meta;
                    if (e < 0) {  // :290-294  first sighting: append, record strand
                        if (n >= BFS_THREAD_CAP) PF_BFS_OVER();
                        e = (int)n;
                        st.E(n++) = u;
                        new_meta = (uint8_t)(2 | (((u & 1) == 0) ? 4 : 0));
                    } else {  // :295-303  seen before: strand must agree
                        if (((um >> 2) & 1) != ((u & 1) == 0 ? 1u : 0u)) {
                            cyc_flag = true;
                            if (!bfs_thread_cyc_add(st, n_cyc, u) || !bfs_thread_cyc_add(st, n_cyc, v)) PF_BFS_OVER();
                        }
                        new_meta = (uint8_t)(2 | (um & 4));
                    }
                    st.M((uint32_t)e) = new_meta;  // :304
                    const uint4 pr = *reinterpret_cast<const uint4 *>(pred + (size_t)u * 4);
                    const uint32_t prow[4] = {pr.x, pr.y, pr.z, pr.w};
                    bool all_pred = true;  // :305-325
                    for (int j = 0; j < 4; ++j) {
                        const uint32_t p = prow[j];
                        if (p == NONE) continue;
                        const int pe = bfs_thread_find(st, n, p);
                        const uint8_t pm = pe >= 0 ? st.M((uint32_t)pe) : 0;
                        if (pe >= 0 && (pm & 3) != 0) {
                            if ((pm & 3) != 1) all_pred = false;
                            if (((pm >> 2) & 1) != ((p & 1) == 0 ? 1u : 0u)) {
                                cyc_flag = true;
                                if (!bfs_thread_cyc_add(st, n_cyc, u) || !bfs_thread_cyc_add(st, n_cyc, p)) PF_BFS_OVER();
                            }
                        } else {
                            all_pred = false;
                        }
                    }
                    if (all_pred) {  // :326-327
                        if (top >= BFS_THREAD_CAP) PF_BFS_OVER();
                        st.T(top++) = u;
                    }
                } else {  // :329-334
                    cyc_flag = true;
                    if (!bfs_thread_cyc_add(st, n_cyc, v) || !bfs_thread_cyc_add(st, n_cyc, u)) PF_BFS_OVER();
                }
            }
        }
        if (top == 1) {  // :337-371
            const uint32_t t0 = st.T(0);
            bool pending = false;
            for (uint32_t i = 0; i < n; ++i)
                if (st.E(i) != t0 && (st.M(i) & 3) == 2) { pending = true; break; }
            if (!pending) {
                r.exit_ov = t0;
                const uint4 ts = *reinterpret_cast<const uint4 *>(succ + (size_t)t0 * 4);
                const bool back = ts.x == s || ts.y == s || ts.z == s || ts.w == s;
                if (back) r.outcome = PF_BFS_CYCLE_EXIT;
                else if (cyc_flag || tip_flag) r.outcome = PF_BFS_REJECT;
                else r.outcome = PF_BFS_ACCEPT;
                break;
            }
        }
    }
#undef PF_BFS_OVER
    r.n_seen = n;
    r.n_cyc = n_cyc;
    r.flag_cycle = cyc_flag;
    r.flag_tip = tip_flag;
    // strict structural test (src/CDBG.cpp:765-782)
    if (r.outcome == PF_BFS_ACCEPT && n >= 4 && n <= 6) {
        const uint32_t t = r.exit_ov;
        bool ok = true;
        for (uint32_t i = 0; i < n && ok; ++i) {
            const uint32_t w = st.E(i);
            if (w == s || w == t) continue;
            const uint4 pr = *reinterpret_cast<const uint4 *>(pred + (size_t)w * 4);
            const uint4 sr = *reinterpret_cast<const uint4 *>(succ + (size_t)w * 4);
            const uint32_t pa[4] = {pr.x, pr.y, pr.z, pr.w}, sa[4] = {sr.x, sr.y, sr.z, sr.w};
            int din = 0, dout = 0;
            uint32_t fp = NONE, fs = NONE;
            for (int j = 0; j < 4; ++j) {
                if (pa[j] != NONE) { if (din == 0) fp = pa[j]; din++; }
                if (sa[j] != NONE) { if (dout == 0) fs = sa[j]; dout++; }
            }
            ok = din == 1 && dout == 1 && (fp >> 1) == (s >> 1) && (fs >> 1) == (t >> 1);
        }
        r.strict = ok ? 1 : 0;
    }
    return r;
}

}  // namespace pf
