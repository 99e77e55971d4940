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

// One traversal record in the reference's visiting order (the text of the commits and of the colour gate: pf_state_ops.hpp).
ColourGate UnitigState::colour_gate() const {
    ColourGate cg;
    cg.n_colors = col->n_colors;
    cg.k = g->k;
    cg.len_bp = g->len_bp.data();
    cg.words = col->words;
    cg.full_mask = col->full_mask.data();
    cg.size_total = col->size_total.data();
    cg.n_full_enc = col->n_full_enc.data();
    cg.succ = succ;
    return cg;
}

void UnitigState::replay(const pf_bfs_record &r, const uint32_t *list) {
    if (col) {
        Commits<FlagsPerUnitig, ColourGate> c{FlagsPerUnitig{flags.data(), plus.data(), minus.data()}, complex_size, colour_gate()};
        c.replay(r, list);
    } else {
        Commits<FlagsPerUnitig> c{FlagsPerUnitig{flags.data(), plus.data(), minus.data()}, complex_size, NoColours{}};
        c.replay(r, list);
    }
}

}  // namespace pfh
