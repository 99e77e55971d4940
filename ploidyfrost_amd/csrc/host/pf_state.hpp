// pfh::UnitigState -- the per-unitig state the reference keeps in its MyUnitig payload (src/MyUnitig.hpp:5-136: `b` bits,
// plus / minus partner pointers) as three arrays, and the commits that mutate it.
#pragma once
#include <cstdint>
#include <vector>

#include "pf_host_colors.hpp"
#include "pf_host_graph.hpp"
#include "ploidyfrost_hip.h"

namespace pfh {

struct UnitigState {
    std::vector<uint8_t> flags;
    std::vector<uint32_t> plus, minus;  // 0 = NULL, id = u + 1
    size_t complex_size = 8;
    // colored path only (src/CCDBG.cpp:2530-2660): colour sets, unitig lengths and the CSR the colour-flow gate walks
    const ColorSets *col = nullptr;
    const UnitigSet *g = nullptr;
    const uint32_t *succ = nullptr;

    void reset(uint32_t n_unitigs);
    // one traversal record in the reference's visiting order; the caller applies the `partner == NULL` gate (src/CDBG.cpp:206, 211)
    void replay(const pf_bfs_record &r, const uint32_t *list);
    bool gate_open(uint32_t entrance_ov) const { return ((entrance_ov & 1) == 0 ? plus[entrance_ov >> 1] : minus[entrance_ov >> 1]) == 0; }

    // the colored accept commit's extra gates (src/CCDBG.cpp:2530-2621) over this state's colour sets, graph and CSR
    struct ColourGate colour_gate() const;
};

}  // namespace pfh
