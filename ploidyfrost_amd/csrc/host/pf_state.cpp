// MyUnitig state (reference src/MyUnitig.hpp:5-136) as three arrays and the order-dependent commits of
// extractSuperBubble_ptr (src/CDBG.cpp:373-413, 552-846; colored twin src/CCDBG.cpp:2349-2660) applied to traversal records.
// Needs no device: the records may come from K-BFS, from the host walkers, or from another rank.
#include "pf_state.hpp"

namespace pfh {

namespace {
constexpr uint8_t B_PLUS = 0x01, B_MINUS = 0x02, B_NON_SUPER = 0x04, B_STRICT_M = 0x08, B_STRICT_P = 0x10, B_COMPLEX_M = 0x20,
                  B_COMPLEX_P = 0x40;
constexpr uint32_t NONE = 0xFFFFFFFFu;
inline bool plus_side_of(uint32_t ov) { return (ov & 1) == 0; }
}  // namespace

void UnitigState::reset(uint32_t n) {
    flags.assign(n, 0);
    plus.assign(n, 0);
    minus.assign(n, 0);
}

// ---- MyUnitig state -------------------------------------------------------------------------
void UnitigState::side_self(uint32_t u, bool plus_side) {
    if (plus_side) { plus[u] = u + 1; flags[u] &= (uint8_t)~B_PLUS; }
    else { minus[u] = u + 1; flags[u] &= (uint8_t)~B_MINUS; }
}
// "if (ex->get_plus() == me) ex->set_plus_self(); else ex->set_minus_self();"
void UnitigState::release(uint32_t ex, uint32_t me) { side_self(ex, plus[ex] == me + 1); }
// interior vertex of any committed traversal (e.g. src/CDBG.cpp:800-826)
void UnitigState::poison(uint32_t u) {
    uint32_t p = plus[u];
    if (p != 0 && p != u + 1) release(p - 1, u);
    side_self(u, true);
    p = minus[u];
    if (p != 0 && p != u + 1) release(p - 1, u);
    side_self(u, false);
    flags[u] |= B_NON_SUPER;
}

// Order-dependent part of extractSuperBubble_ptr: the three setNoBubble commits
// (src/CDBG.cpp:552-846) and the no-exit tail (:373-413), applied to one device record.
void UnitigState::replay(const pf_bfs_record &r, const uint32_t *list) {
    const uint32_t s = r.entrance, su = s >> 1;
    if (r.outcome == PF_BFS_NONE) {
        if (!r.flag_cycle) return;
        for (uint32_t i = 0; i < r.n_list; ++i) poison(list[i] >> 1);
        side_self(su, plus_side_of(s));
        return;
    }
    const uint32_t t = r.exit, tu = t >> 1;
    if (r.outcome == PF_BFS_CYCLE_EXIT) {  // setNoBubble_ptr_cycle
        if (col) {
            // src/CCDBG.cpp:2351-2384: a side is self-marked only if it held a real partner
            for (uint32_t i = 0; i < r.n_list; ++i) {
                const uint32_t w = list[i] >> 1;
                uint32_t p = plus[w];
                if (p != 0 && p != w + 1) { release(p - 1, w); side_self(w, true); }
                p = minus[w];
                if (p != 0 && p != w + 1) { release(p - 1, w); side_self(w, false); }
                flags[w] |= B_NON_SUPER;
            }
        } else
        for (uint32_t i = 0; i < r.n_list; ++i) poison(list[i] >> 1);
        side_self(su, plus_side_of(s));
        side_self(tu, !plus_side_of(t));
        return;
    }
    if (r.outcome == PF_BFS_REJECT) {  // setNoBubble_ptr(seen, p)
        uint32_t p = plus_side_of(s) ? plus[su] : minus[su];
        if (p != 0) release(p - 1, su);
        side_self(su, plus_side_of(s));
        p = !plus_side_of(t) ? plus[tu] : minus[tu];
        if (p != 0) release(p - 1, tu);
        side_self(tu, !plus_side_of(t));
        for (uint32_t i = 0; i < r.n_list; ++i)
            if (list[i] != s && list[i] != t) poison(list[i] >> 1);
        return;
    }
    // PF_BFS_ACCEPT: setNoBubble_ptr(p, seen)
    if (r.n_seen < 4) return;
    if ((flags[tu] & B_NON_SUPER) || (flags[su] & B_NON_SUPER)) {
        for (uint32_t i = 0; i < r.n_list; ++i) {
            const uint32_t w = list[i];
            if (w == s) side_self(su, plus_side_of(s));
            else if (w == t) side_self(tu, !plus_side_of(t));
            else poison(w >> 1);
        }
        return;
    }
    if (r.strict) {  // n_seen <= 6 and the structural test, evaluated on the device
        flags[su] |= plus_side_of(s) ? B_STRICT_P : B_STRICT_M;
        flags[tu] |= !plus_side_of(t) ? B_STRICT_P : B_STRICT_M;
    }
    if (r.n_seen > complex_size) {
        flags[su] |= plus_side_of(s) ? B_COMPLEX_P : B_COMPLEX_M;
        flags[tu] |= !plus_side_of(t) ? B_COMPLEX_P : B_COMPLEX_M;
    }
    for (uint32_t i = 0; i < r.n_list; ++i)
        if (list[i] != s && list[i] != t) poison(list[i] >> 1);
    if (col && !colours_allow(r, list)) {
        side_self(su, plus_side_of(s));
        side_self(tu, !plus_side_of(t));
        return;
    }
    if (plus_side_of(s)) { plus[su] = tu + 1; flags[su] |= B_PLUS; }
    else { minus[su] = tu + 1; flags[su] |= B_MINUS; }
    if (plus_side_of(t)) { minus[tu] = su + 1; flags[tu] |= B_MINUS; }
    else { plus[tu] = su + 1; flags[tu] |= B_PLUS; }
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
