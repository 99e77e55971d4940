// K-BFS, last tier: traversals of any size (up to the whole graph).  Same algorithm as pf_bfs.hpp
// (reference src/CDBG.cpp:253-372) but the per-unitig state lives in direct-indexed arrays of
// N entries in global memory, stamped with a per-candidate epoch so that nothing is cleared between
// candidates:
//   info[u]   = (epoch << 4) | (recorded strand << 2) | state      state_map + strand_map
//   first[u]  = oriented vertex as first seen                       the entry of vec_km_seen
//   seen[]    = first-seen order (the list the host replay needs), todo[] = LIFO, cyc[] = cycle set
//   cycstamp[ov] = epoch when ov is in the cycle set (deduplication)
// The "only the exit is left" test (src/CDBG.cpp:337-351) is O(1): a running count of entries in state
// `seen`, minus the entry of the single stacked vertex if it is one of them.
// A traversal that swallows a chromosome is inherently sequential (LIFO order fixes the seen order
// and the cycle flags); the wave only parallelises the CSR loads (successor row, then the four
// predecessor rows) -- one candidate per wave, several waves side by side.
#pragma once
#include "pf_bfs.hpp"

namespace pf {

struct HugeStore {
    uint32_t *info;      // [N]
    uint32_t *first;     // [N]
    uint32_t *cycstamp;  // [2N]
    uint32_t *seen;      // [N + 1]
    uint32_t *todo;      // [2N + 8]
    uint32_t *cyc;       // [2N]
    uint32_t n_unitigs;
};

__device__ inline bool huge_cyc_add(const HugeStore &st, uint32_t epoch, uint32_t &n_cyc, uint32_t ov) {
    if (st.cycstamp[ov] == epoch) return true;
    if (n_cyc >= 2 * st.n_unitigs) return false;
    if (lane_id() == 0) {
        st.cycstamp[ov] = epoch;
        st.cyc[n_cyc] = ov;
    }
    n_cyc++;
    __builtin_amdgcn_wave_barrier();
    return true;
}

// Per popped vertex the memory round trips are: the vertex itself (skipped when it was pushed in the
// previous step), {its state word, its successor row}, {the four predecessor rows, the successors'
// state words}, {the predecessors' state words} -- each group one wave-wide load.  State words
// changed while the successors are resolved in order are patched in the lanes that prefetched them,
// so no load is repeated and a single fence per step orders the stores.
__device__ inline BfsResult bfs_traverse_huge(const uint32_t *__restrict__ succ, const uint32_t *__restrict__ pred,
                                              const HugeStore &st, uint32_t epoch, uint32_t s) {
    const int lane = lane_id();
    BfsResult r;
    r.exit_ov = NONE;
    r.n_seen = 0;
    r.n_cyc = 0;
    r.outcome = PF_BFS_NONE;
    r.flag_cycle = r.flag_tip = r.strict = 0;
    r.overflow = false;
    const uint32_t tag = epoch << 4;
    auto decode = [&](uint32_t x) -> uint32_t { return (x >> 4) == epoch ? (x & 15) : 0; };  // 0 = not in state_map
    uint32_t n = 1, top = 1, n_cyc = 0, n_pending = 0;  // n_pending = entries in state `seen` (2)
    bool cyc_flag = false, tip_flag = false;
    if (lane == 0) {
        st.seen[0] = s;
        st.first[s >> 1] = s;
        st.todo[0] = s;
    }
    uint32_t cached_top = s;  // value of todo[top-1] when known without a load
    bool have_top = true;
    wave_sync();
    while (top > 0) {
        const uint32_t v = have_top ? cached_top : st.todo[top - 1];
        have_top = false;
        top--;
        // {state word of v, successor row of v}
        const uint32_t ld1 = lane < 4 ? succ[(size_t)v * 4 + lane] : (lane == 4 ? st.info[v >> 1] : 0u);
        const uint32_t my_succ = lane < 4 ? ld1 : NONE;
        const uint32_t v_old = decode(__shfl(ld1, 4, WAVE));
        if ((v_old & 3) == 2) n_pending--;
        const uint32_t v_info = tag | 1u | (((v & 1) == 0) ? 4u : 0u);  // visited; strand_map[id(v)] = v.strand
        if (lane == 0) st.info[v >> 1] = v_info;
        // {predecessor rows of the successors, state words of the successors}
        const uint32_t sb = __shfl(my_succ, lane >> 2, WAVE);
        const uint32_t my_pred = (lane < 16 && sb != NONE) ? pred[(size_t)sb * 4 + (lane & 3)] : NONE;
        uint32_t my_sinfo = (lane < 4 && my_succ != NONE) ? ((my_succ >> 1) == (v >> 1) ? v_info : st.info[my_succ >> 1]) : 0u;
        // {state words of the predecessors}
        uint32_t my_pinfo = (lane < 16 && my_pred != NONE) ? ((my_pred >> 1) == (v >> 1) ? v_info : st.info[my_pred >> 1]) : 0u;
        auto patch = [&](uint32_t unitig, uint32_t word) {
            if (lane < 4 && my_succ != NONE && (my_succ >> 1) == unitig) my_sinfo = word;
            if (lane < 16 && my_pred != NONE && (my_pred >> 1) == unitig) my_pinfo = word;
        };
        if (__ballot(my_succ != NONE) == 0) {
            tip_flag = true;
        } else {
            for (int b = 0; b < 4; ++b) {
                const uint32_t u = __shfl(my_succ, b, WAVE);
                if (u == NONE) continue;
                if (u == s) {
                    cyc_flag = true;
                    if (!huge_cyc_add(st, epoch, n_cyc, s) || !huge_cyc_add(st, epoch, n_cyc, v)) { r.overflow = true; return r; }
                    continue;
                }
                const uint32_t um = decode(__shfl(my_sinfo, b, WAVE));
                // the entrance sits in `seen` before it has a state; it is popped first, so afterwards
                // "in state_map" and "in seen" coincide
                if (um == 0 || (um & 3) != 1) {
                    uint32_t new_info;
                    if (um == 0) {
                        if (n > st.n_unitigs) { r.overflow = true; return r; }
                        if (lane == 0) { st.seen[n] = u; st.first[u >> 1] = u; }
                        n++;
                        new_info = tag | 2u | (((u & 1) == 0) ? 4u : 0u);
                        n_pending++;
                    } else {
                        if (((um >> 2) & 1) != ((u & 1) == 0 ? 1u : 0u)) {
                            cyc_flag = true;
                            if (!huge_cyc_add(st, epoch, n_cyc, u) || !huge_cyc_add(st, epoch, n_cyc, v)) { r.overflow = true; return r; }
                        }
                        new_info = tag | 2u | (um & 4);  // state was already `seen`
                    }
                    if (lane == 0) st.info[u >> 1] = new_info;
                    patch(u >> 1, new_info);
                    bool all_pred = true;
                    for (int j = 0; j < 4; ++j) {
                        const uint32_t p = __shfl(my_pred, b * 4 + j, WAVE);
                        const uint32_t pm = decode(__shfl(my_pinfo, b * 4 + j, WAVE));
                        if (p == NONE) continue;
                        if (pm != 0) {
                            if ((pm & 3) != 1) all_pred = false;
                            if (((pm >> 2) & 1) != ((p & 1) == 0 ? 1u : 0u)) {
                                cyc_flag = true;
                                if (!huge_cyc_add(st, epoch, n_cyc, u) || !huge_cyc_add(st, epoch, n_cyc, p)) { r.overflow = true; return r; }
                            }
                        } else {
                            all_pred = false;
                        }
                    }
                    if (all_pred) {
                        if (top >= 2 * st.n_unitigs + 8) { r.overflow = true; return r; }
                        if (lane == 0) st.todo[top] = u;
                        top++;
                        cached_top = u;
                        have_top = true;
                    }
                } else {
                    cyc_flag = true;
                    if (!huge_cyc_add(st, epoch, n_cyc, v) || !huge_cyc_add(st, epoch, n_cyc, u)) { r.overflow = true; return r; }
                }
            }
        }
        // No fence between steps: a wave's vector-memory instructions execute in program order and the L1
        // is write-through, so the next step's loads observe this step's stores (same-wave RAW); the
        // barrier only pins the compiler's ordering.
        __builtin_amdgcn_wave_barrier();
        if (top == 1) {
            const uint32_t t0 = have_top ? cached_top : st.todo[0];
            const uint32_t tm = decode(st.info[t0 >> 1]);
            const uint32_t mine = ((tm & 3) == 2 && st.first[t0 >> 1] == t0) ? 1u : 0u;
            if (n_pending == mine) {
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
    return r;  // n > 6 here by construction: the strict test (n_seen <= 6) cannot apply
}

}  // namespace pf
