#include "pf_bfs_host.hpp"

#include <emmintrin.h>
#include <sys/mman.h>

#include <algorithm>
#include <cstdlib>
#include <unordered_set>

namespace pfh {

namespace {
constexpr uint32_t NONE = 0xFFFFFFFFu;
}

void advise_huge_pages(const void *p, size_t bytes) {
    if (!p) return;
    constexpr uintptr_t H = (uintptr_t)2 << 20;
    const uintptr_t a = ((uintptr_t)p + H - 1) & ~(H - 1), e = ((uintptr_t)p + bytes) & ~(H - 1);
    if (e > a) (void)madvise(reinterpret_cast<void *>(a), e - a, MADV_HUGEPAGE);
}

const std::vector<uint32_t> &HugeWalker::walk(const uint32_t *succ, const uint32_t *pred, uint32_t N, uint32_t s, pf_bfs_record &r) {
    if (info.size() != N) {
        info.clear();
        info.reserve(N);
        advise_huge_pages(info.data(), (size_t)N * 4);
        info.assign(N, 0); epoch = 0;
    }
    if (++epoch >= (1u << 28)) { std::fill(info.begin(), info.end(), 0); epoch = 1; }
    const uint32_t tag = epoch << 4;
    auto state_of = [&](uint32_t unitig) -> uint32_t { const uint32_t x = info[unitig]; return (x >> 4) == epoch ? (x & 15) : 0; };  // 0 = not in state_map
    auto strand_bit = [](uint32_t ov) -> uint32_t { return (ov & 1) == 0 ? 4u : 0u; };
    seen.clear();
    todo.clear();
    cyc.clear();
    bool flag_cycle = false, flag_tip = false;
    std::unordered_set<uint32_t> in_cyc;
    auto cyc_add = [&](uint32_t ov) {  // unordered_set<UnitigMap>: kept in insertion order, the commits do not depend on it
        if (in_cyc.insert(ov).second) cyc.push_back(ov);
    };
    uint64_t n_pending = 0;  // unitigs in state `seen` (0x02)
    seen.push_back(s);
    todo.push_back(s);
    r.entrance = s;
    r.exit = NONE;
    r.outcome = PF_BFS_NONE;
    r.strict = 0;
    r.pad_ = 0;
    // The slots of a CSR row are indexed by base: which of the four hold a vertex is as good as random, so a branch per slot
    // mispredicts every other time (three to five times per popped vertex; the walk is bound by exactly that, not by memory: 26 ns
    // per vertex with everything in L1).  The rows are therefore turned into bit masks of their occupied slots -- one vector
    // compare -- and the loops run over the set bits.
    auto occupied = [](const uint32_t *row) -> unsigned {
        const __m128i x = _mm_loadu_si128(reinterpret_cast<const __m128i *>(row));
        return ~(unsigned)_mm_movemask_ps(_mm_castsi128_ps(_mm_cmpeq_epi32(x, _mm_set1_epi32(-1)))) & 15u;
    };
    while (!todo.empty()) {
        const uint32_t v = todo.back();
        todo.pop_back();
        {
            const uint32_t x = info[v >> 1];
            n_pending -= (uint64_t)(((x >> 4) == epoch) & ((x & 3) == 2));
        }
        info[v >> 1] = tag | 1u | strand_bit(v);  // state_map[id] = 0x01; strand_map[id] = v.strand
        const uint32_t *row = succ + (size_t)v * 4;
        const unsigned sm = occupied(row);
        for (unsigned m = sm; m; m &= m - 1) {
            const uint32_t u = row[__builtin_ctz(m)];
            __builtin_prefetch(pred + (size_t)u * 4);
            __builtin_prefetch(&info[u >> 1]);
            __builtin_prefetch(succ + (size_t)u * 4);   // u is popped soon (LIFO) when all its predecessors are in
        }
        for (unsigned m = sm; m; m &= m - 1) {
            const uint32_t u = row[__builtin_ctz(m)];
            if (u == s) {
                flag_cycle = true;
                cyc_add(s);
                cyc_add(v);
                continue;
            }
            const uint32_t um = state_of(u >> 1);
            if ((um & 3) != 1) {   // not in the map, or seen and not yet visited
                if (um == 0) {
                    seen.push_back(u);
                    info[u >> 1] = tag | 2u | strand_bit(u);
                    ++n_pending;
                } else {
                    if ((um & 4) != strand_bit(u)) {
                        flag_cycle = true;
                        cyc_add(u);
                        cyc_add(v);
                    }
                    info[u >> 1] = tag | 2u | (um & 4);  // it was `seen` already
                }
                bool all_pred = true;
                const uint32_t *prow = pred + (size_t)u * 4;
                for (unsigned pm4 = occupied(prow); pm4; pm4 &= pm4 - 1) {
                    const uint32_t p = prow[__builtin_ctz(pm4)];
                    const uint32_t pm = state_of(p >> 1);
                    all_pred &= (pm & 3) == 1;   // (0 = not in the map, 2 = seen only)
                    if (pm != 0 && (pm & 4) != strand_bit(p)) {
                        flag_cycle = true;
                        cyc_add(u);
                        cyc_add(p);
                    }
                }
                if (all_pred) todo.push_back(u);
            } else {
                flag_cycle = true;
                cyc_add(v);
                cyc_add(u);
            }
        }
        if (!sm) flag_tip = true;
        if (todo.size() == 1) {
            // "no other entry of vec_km_seen is in state 0x02" (src/CDBG.cpp:337-351) by counting
            const uint32_t t0 = todo[0];
            const uint32_t tm = state_of(t0 >> 1);
            // (is t0 itself the pending entry of its unitig?  A `seen` entry keeps the strand of its first sighting.)
            const uint64_t mine = ((tm & 3) == 2 && (tm & 4) == strand_bit(t0)) ? 1 : 0;
            if (n_pending == mine) {
                r.exit = t0;
                bool back = false;
                const uint32_t *trow = succ + (size_t)t0 * 4;
                for (int b = 0; b < 4; ++b) back |= trow[b] == s;
                r.outcome = back ? PF_BFS_CYCLE_EXIT : (flag_cycle || flag_tip) ? PF_BFS_REJECT : PF_BFS_ACCEPT;
                break;
            }
        }
    }
    r.n_seen = (uint32_t)seen.size();
    // the structural "strict" test of the accept commit (src/CDBG.cpp:765-782): at most 6 vertices and every inner one has exactly
    // one predecessor, on s's unitig, and one successor, on t's (what K-BFS evaluates for the traversals it keeps)
    if (r.outcome == PF_BFS_ACCEPT && seen.size() >= 4 && seen.size() <= 6) {
        bool ok = true;
        for (uint32_t w : seen) {
            if (w == s || w == r.exit) continue;
            int din = 0, dout = 0;
            uint32_t fp = NONE, fs = NONE;
            for (int b = 0; b < 4; ++b) {
                const uint32_t p = pred[(size_t)w * 4 + b], q = succ[(size_t)w * 4 + b];
                if (p != NONE) { if (!din) fp = p; ++din; }
                if (q != NONE) { if (!dout) fs = q; ++dout; }
            }
            if (!(din == 1 && dout == 1 && (fp >> 1) == (s >> 1) && (fs >> 1) == (r.exit >> 1))) { ok = false; break; }
        }
        r.strict = ok;
    }
    r.flag_cycle = flag_cycle;
    r.flag_tip = flag_tip;
    const std::vector<uint32_t> &list = r.outcome != PF_BFS_NONE ? seen : cyc;
    r.n_list = r.outcome != PF_BFS_NONE ? (uint32_t)seen.size() : (flag_cycle ? (uint32_t)cyc.size() : 0);
    return list;
}

}  // namespace pfh
