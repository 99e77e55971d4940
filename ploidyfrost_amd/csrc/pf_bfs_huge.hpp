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

// Memory round trips decide the speed of a chromosome-long traversal, so each step needs only one in
// the common case:
//   * `pred16[ov]` holds, for each of the four successor slots of ov, the predecessor row of that
//     successor (a two-hop row built once per graph), so the neighbourhood of v is two loads off v;
//   * the next vertex popped is almost always the successor pushed last, so while the state words of
//     v's successors and predecessors are in flight the rows (succ + pred16) of all four successors
//     are fetched speculatively; the chosen one is then already in registers;
//   * state words changed while the successors are resolved in order are patched in the lanes that
//     prefetched them; no load is repeated and no fence is needed inside a step.
__device__ inline BfsResult bfs_traverse_huge(const uint32_t *__restrict__ succ, const uint32_t *__restrict__ pred16,
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
    // speculative rows of the previous step's successors: spec_S lanes 4c..4c+3 = succ row of
    // candidate c, spec_P lanes 16c..16c+15 = its pred16 row; spec_v[c] = the candidate itself
    uint32_t spec_S = NONE, spec_P = NONE, spec_cand = NONE;  // spec_cand: lanes 0..3 hold the candidates
    uint32_t cached_top = s, cached_word = 0, cached_first = s;
    bool have_top = true;
    __builtin_amdgcn_wave_barrier();
    while (top > 0) {
        const bool popped_cached = have_top;
        const uint32_t v = have_top ? cached_top : st.todo[top - 1];
        have_top = false;
        top--;
        // rows of v: from the speculation of the previous step when v was one of its successors
        uint32_t my_succ, my_pred;
        {
            const unsigned long long hit = __ballot(lane < 4 && spec_cand == v);
            if (hit) {
                const int c = __ffsll((long long)hit) - 1;
                my_succ = __shfl(spec_S, 4 * c + (lane & 3), WAVE);
                my_pred = __shfl(spec_P, 16 * c + (lane & 15), WAVE);
                if (lane >= 4) my_succ = NONE;
                if (lane >= 16) my_pred = NONE;
            } else {
                my_succ = lane < 4 ? succ[(size_t)v * 4 + lane] : NONE;
                my_pred = lane < 16 ? pred16[(size_t)v * 16 + lane] : NONE;
            }
        }
        // one round trip: state words of v, of its successors and of their predecessors, plus the
        // rows of the four successors for the next step
        // the word of a vertex pushed in the previous step is still in registers
        const uint32_t v_word = popped_cached ? cached_word : st.info[v >> 1];
        uint32_t my_sinfo = (lane < 4 && my_succ != NONE) ? st.info[my_succ >> 1] : 0u;
        uint32_t my_pinfo = (lane < 16 && my_pred != NONE) ? st.info[my_pred >> 1] : 0u;
        {
            const uint32_t cs = __shfl(my_succ, lane >> 2, WAVE);   // candidate of my 4-lane group (lanes 0..15)
            const uint32_t cp = __shfl(my_succ, lane >> 4, WAVE);   // candidate of my 16-lane group
            spec_S = (lane < 16 && cs != NONE) ? succ[(size_t)cs * 4 + (lane & 3)] : NONE;
            spec_P = cp != NONE ? pred16[(size_t)cp * 16 + (lane & 15)] : NONE;
            spec_cand = my_succ;
        }
        const uint32_t v_old = decode(v_word);
        if ((v_old & 3) == 2) n_pending--;
        const uint32_t v_info = tag | 1u | (((v & 1) == 0) ? 4u : 0u);  // visited; strand_map[id(v)] = v.strand
        if (lane == 0) st.info[v >> 1] = v_info;
        // v's own word may be among the prefetched ones (self loops, hairpins)
        if (lane < 4 && my_succ != NONE && (my_succ >> 1) == (v >> 1)) my_sinfo = v_info;
        if (lane < 16 && my_pred != NONE && (my_pred >> 1) == (v >> 1)) my_pinfo = v_info;
        auto patch = [&](uint32_t unitig, uint32_t word) {
            if (lane < 4 && my_succ != NONE && (my_succ >> 1) == unitig) my_sinfo = word;
            if (lane < 16 && my_pred != NONE && (my_pred >> 1) == unitig) my_pinfo = word;
            if (have_top && (cached_top >> 1) == unitig) cached_word = word;
        };
        if (__ballot(my_succ != NONE) == 0) {
            tip_flag = true;
        } else {
            for (int b = 0; b < 4; ++b) {
                const uint32_t u = read_lane(my_succ, b);
                if (u == NONE) continue;
                if (u == s) {
                    cyc_flag = true;
                    if (!huge_cyc_add(st, epoch, n_cyc, s) || !huge_cyc_add(st, epoch, n_cyc, v)) { r.overflow = true; return r; }
                    continue;
                }
                const uint32_t um = decode(read_lane(my_sinfo, b));
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
                    // the four predecessors of u sit in lanes 4b..4b+3 (with their state words, patched above):
                    // judged by those lanes at once, two ballots instead of a scalar walk over them
                    const uint32_t pm_l = decode(my_pinfo);
                    const bool p_valid = lane < 16 && my_pred != NONE;
                    const bool p_blocks = p_valid && (pm_l == 0 || (pm_l & 3) != 1);
                    const bool p_strand = p_valid && pm_l != 0 && (((pm_l >> 2) & 1) != ((my_pred & 1) == 0 ? 1u : 0u));
                    const unsigned long long group = 0xFull << (4 * b);
                    const bool all_pred = (__ballot(p_blocks) & group) == 0;
                    unsigned long long bad = __ballot(p_strand) & group;
                    while (bad) {  // ascending j, as the reference's loop over the predecessors
                        const int l = __ffsll((long long)bad) - 1;
                        bad &= bad - 1;
                        cyc_flag = true;
                        if (!huge_cyc_add(st, epoch, n_cyc, u) || !huge_cyc_add(st, epoch, n_cyc, read_lane(my_pred, l))) { r.overflow = true; return r; }
                    }
                    if (all_pred) {
                        if (top >= 2 * st.n_unitigs + 8) { r.overflow = true; return r; }
                        if (lane == 0) st.todo[top] = u;
                        top++;
                        cached_top = u;
                        cached_word = new_info;
                        cached_first = um == 0 ? u : NONE;  // first[] of a vertex seen earlier has to be loaded
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
            const uint32_t tm = decode(have_top ? cached_word : st.info[t0 >> 1]);
            uint32_t mine = 0;
            if ((tm & 3) == 2) mine = ((have_top && cached_first != NONE) ? cached_first : st.first[t0 >> 1]) == t0 ? 1u : 0u;
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
