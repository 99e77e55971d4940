// MyUnitig state (reference src/MyUnitig.hpp:5-136) as three arrays and the order-dependent commits of
// extractSuperBubble_ptr (src/CDBG.cpp:373-413, 552-846; colored twin src/CCDBG.cpp:2349-2660) applied to traversal records.
// Needs no device: the records may come from K-BFS, from the host walkers, or from another rank.
#include "pf_state.hpp"

#include "pf_state_ops.hpp"

namespace pfh {

using namespace state_bits;
constexpr uint32_t NONE = 0xFFFFFFFFu;

void UnitigState::reset(uint32_t n) {
    flags.assign(n, 0);
    plus.assign(n, 0);
    minus.assign(n, 0);
}

namespace {
// the colored accept commit's gate as the hook Commits<> calls (UnitigState::colours_allow below)
struct ColourGate {
    static constexpr bool colored = true;
    UnitigState *st;
    template <class A> bool allow(A &, const pf_bfs_record &r, const uint32_t *list) const { return st->colours_allow(r, list); }
};
}  // namespace

// One traversal record in the reference's visiting order (the text of the commits: pf_state_ops.hpp).
void UnitigState::replay(const pf_bfs_record &r, const uint32_t *list) {
    if (col) {
        Commits<FlagsPerUnitig, ColourGate> c{FlagsPerUnitig{flags.data(), plus.data(), minus.data()}, complex_size, ColourGate{this}};
        c.replay(r, list);
    } else {
        Commits<FlagsPerUnitig> c{FlagsPerUnitig{flags.data(), plus.data(), minus.data()}, complex_size, NoColours{}};
        c.replay(r, list);
    }
}

// The colored accept commit's extra gates (src/CCDBG.cpp:2530-2621): both endpoints carry every colour on every
// k-mer, and every colour a vertex of the bubble carries in full continues, in full, on one of its successors.
bool UnitigState::colours_allow(const pf_bfs_record &r, const uint32_t *list) {
    const ColorSets &cs = *col;
    const uint32_t C = cs.n_colors;
    const uint32_t s = r.entrance, su = s >> 1, t = r.exit, tu = t >> 1;
    const uint64_t km_s = g->len_km(su), km_t = g->len_km(tu);
    bool f = true;
    if (cs.size_with(su, km_s, km_s) != km_s * C) { f = false; flags[su] |= B_NON_SUPER; }
    // the exit's set is sized with the entrance's mapping (:2552): only the pair encoding notices
    if (cs.size_with(tu, km_t, km_s) != km_t * C) { f = false; flags[tu] |= B_NON_SUPER; }
    if (!f) return false;
    const uint64_t all = C == 64 ? ~0ull : ((1ull << C) - 1);
    for (uint32_t i = 0; i < r.n_list; ++i) {
        const uint32_t w = list[i];
        if (w == t) continue;
        // the reference keys its colour lists by unitig id and pre-loads both endpoints with every colour
        const uint64_t mine = ((w >> 1) == su || (w >> 1) == tu) ? all : cs.full_mask[w >> 1];
        uint64_t cont = 0;
        const uint32_t *row = &succ[(size_t)w * 4];
        for (int b = 0; b < 4; ++b)
            if (row[b] != NONE) cont |= cs.full_mask[row[b] >> 1];
        if ((cont & mine) != mine) return false;
    }
    return true;
}

}  // namespace pfh
