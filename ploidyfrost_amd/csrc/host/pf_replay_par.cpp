// Components of the commit replay on the host, and the footprint check of the model behind them (pf_replay_par.hpp).
#include "pf_replay_par.hpp"

#include <cstdlib>

namespace pfh {

namespace {
constexpr uint32_t NONE = 0xFFFFFFFFu;
// test hook: PF_CC_WITHOUT=2 drops the two-partner rule, =4 the rejected-exit rule, =8 the colored path's incomplete-endpoint rule (the footprint check must then find accesses
// outside the components: tests/test_replay_parallel_cpu.py holds the check itself to that)
int rules_dropped() {
    const char *e = getenv("PF_CC_WITHOUT");
    return e ? atoi(e) : 0;
}
}  // namespace

void SideComponents::reset(uint32_t n_unitigs) {
    parent_.resize(2 * (size_t)n_unitigs);
    for (size_t i = 0; i < parent_.size(); ++i) parent_[i] = (uint32_t)i;
    first_.assign(2 * (size_t)n_unitigs, NONE);
}

// An accepted record links `side` to `other_side`.  A side that accepted records link to two different sides (of two unitigs, or
// the two sides of one) can be left with a partner whose link back has been overwritten or released; releasing that partner
// later falls through to the unitig's minus side (pf_state_ops.hpp): both sides of such a unitig are one unit.  (A traversal and
// its twin from the other end link the same two sides.)
void SideComponents::partner(uint32_t side, uint32_t other_side) {
    if (first_[side] == NONE) first_[side] = other_side;
    else if (first_[side] != other_side && !(rules_dropped() & 2)) unite(side, side ^ 1u);
}

void SideComponents::add_record(const pf_bfs_record &r, const uint32_t *list) {
    const uint32_t s = r.entrance;
    const uint32_t anchor = entrance_side(s);
    const bool has_exit = r.outcome != PF_BFS_NONE;
    const uint32_t t = r.exit;
    // cycle commits poison every list entry; the others spare the two endpoints
    const bool all_interior = r.outcome == PF_BFS_NONE || r.outcome == PF_BFS_CYCLE_EXIT;
    for (uint32_t i = 0; i < r.n_list; ++i) {
        const uint32_t w = list[i];
        if (!all_interior && (w == s || w == t)) continue;
        unite(anchor, 2 * (w >> 1));
        unite(anchor, 2 * (w >> 1) + 1);
    }
    if (has_exit) unite(anchor, exit_side(t));
    if (r.outcome == PF_BFS_ACCEPT) {
        partner(entrance_side(s), exit_side(t));
        partner(exit_side(t), entrance_side(s));
        if (gate_ && !(rules_dropped() & 8)) {   // NON_SUPER on an endpoint with an incomplete colour set: a write to the whole unitig
            if (gate_->incomplete_entrance(s >> 1)) unite(2 * (s >> 1), 2 * (s >> 1) + 1);
            if (gate_->incomplete_exit(t >> 1, s >> 1)) unite(2 * (t >> 1), 2 * (t >> 1) + 1);
        }
    }
    // setNoBubble_ptr(seen, p) releases whatever the exit's side holds, a self-mark included ("release(t, t)" resolves to the plus
    // side when that is self-marked too): the exit counts as one unit
    if (r.outcome == PF_BFS_REJECT && !(rules_dropped() & 4)) unite(exit_side(t), exit_side(t) ^ 1u);
}

void SideComponents::order(const pf_bfs_record *rec, uint64_t n, uint32_t n_classes, std::vector<uint32_t> &order, std::vector<uint32_t> &class_off) {
    class_off.assign(n_classes + 1, 0);
    std::vector<uint32_t> cls(n);
    for (uint64_t i = 0; i < n; ++i) {
        cls[i] = replay_class_of(find(entrance_side(rec[i].entrance)), n_classes);
        class_off[cls[i] + 1]++;
    }
    for (uint32_t c = 0; c < n_classes; ++c) class_off[c + 1] += class_off[c];
    order.resize(n);
    std::vector<uint32_t> at(class_off.begin(), class_off.end() - 1);
    for (uint64_t i = 0; i < n; ++i) order[at[cls[i]]++] = (uint32_t)i;
}

// ---- footprint check ---------------------------------------------------------------------------------------------------
namespace {
struct FlagsChecked {
    FlagsPerUnitig base;
    SideComponents *cc;
    uint64_t *bad;
    uint32_t cur = 0;
    void own(uint32_t u, bool ps) const {
        if (cc->label(2 * u + (ps ? 0 : 1)) != cur) ++*bad;
    }
    uint32_t link(uint32_t u, bool ps) const { own(u, ps); return base.link(u, ps); }
    // looked at without owning it: must not come out true for a plus side outside the component
    bool plus_points_to(uint32_t ex, uint32_t me) const {
        const bool yes = base.plus_points_to(ex, me);
        if (yes && cc->label(2 * ex) != cur) ++*bad;
        return yes;
    }
    void set_link(uint32_t u, bool ps, uint32_t v, bool real) { own(u, ps); base.set_link(u, ps, v, real); }
    void mark_strict(uint32_t u, bool ps) { own(u, ps); base.mark_strict(u, ps); }
    void mark_complex(uint32_t u, bool ps) { own(u, ps); base.mark_complex(u, ps); }
    bool non_super(uint32_t u, bool ps) const { own(u, ps); return base.non_super(u, ps); }
    void set_non_super(uint32_t u) { own(u, true); own(u, false); base.set_non_super(u); }
    void begin_record(const pf_bfs_record &r) { cur = cc->label(entrance_side(r.entrance)); }
};
}  // namespace

namespace {
template <class Col>
uint64_t check_with(const pf_bfs_record *rec, uint64_t n, const uint32_t *pool, uint32_t n_unitigs, size_t complex_size, uint64_t slice,
                    uint64_t *first_bad, Col col, const ColourGate *gate) {
    std::vector<uint8_t> flags(n_unitigs, 0);
    std::vector<uint32_t> plus(n_unitigs, 0), minus(n_unitigs, 0);
    SideComponents cc;
    cc.reset(n_unitigs);
    cc.set_colour_gate(gate);
    uint64_t bad = 0;
    if (first_bad) *first_bad = UINT64_MAX;
    if (slice == 0) slice = n ? n : 1;
    Commits<FlagsChecked, Col> cm{FlagsChecked{FlagsPerUnitig{flags.data(), plus.data(), minus.data()}, &cc, &bad, 0}, complex_size, col};
    auto list_of = [&](const pf_bfs_record &r) { return pool + r.list_off; };
    for (uint64_t a = 0; a < n; a += slice) {
        const uint64_t b = std::min(n, a + slice);
        cc.add(rec + a, b - a, list_of);
        for (uint64_t i = a; i < b; ++i) {
            const pf_bfs_record &r = rec[i];
            cm.a.begin_record(r);
            const uint64_t before = bad;
            if (cm.gate_open(r.entrance)) cm.replay(r, list_of(r));
            if (bad != before && first_bad && *first_bad == UINT64_MAX) *first_bad = i;
        }
    }
    return bad;
}
}  // namespace

uint64_t check_footprints(const pf_bfs_record *rec, uint64_t n, const uint32_t *pool, uint32_t n_unitigs, size_t complex_size, uint64_t slice,
                          uint64_t *first_bad, const ColourGate *gate) {
    if (gate) return check_with<ColourGate>(rec, n, pool, n_unitigs, complex_size, slice, first_bad, *gate, gate);
    return check_with<NoColours>(rec, n, pool, n_unitigs, complex_size, slice, first_bad, NoColours{}, nullptr);
}

}  // namespace pfh
