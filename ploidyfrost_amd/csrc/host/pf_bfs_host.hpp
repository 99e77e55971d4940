// Third tier of K-BFS on host cores: the traversals that outgrow the device's 4096-entry tables.
//
// extractSuperBubble_ptr (reference src/CDBG.cpp:253-372) pops a LIFO; which vertices it has seen when it stops, and
// whether it stops at an exit at all, depend on that order, so one traversal cannot be spread over lanes or waves -- it is a
// chain of dependent memory accesses, a few per popped vertex.  A lone wavefront pays ~0.7-1 us of HBM latency plus ~1.5 us
// of instruction issue per vertex (pf_bfs_huge.hpp: 2.4 us measured); a host core walks the same CSR rows (already on the
// host for the commit replay) with ~100 ns of memory latency and out-of-order execution.  Such traversals are rare in count
// (a handful per million unitigs: wherever a repeated (k-1)-mer links two loci the walk runs on to the end of the chromosome)
// but each visits 10^4 - 10^6 vertices, so they decide the length of a pass.  The small and the 4096-entry tiers stay on the
// device (one wavefront per candidate, hundreds of thousands side by side); pf_bfs_candidates_split hands the rest over.
//
// State per walker: info[u] = (epoch << 4) | (recorded strand << 2) | state, direct-indexed by unitig and stamped with a
// per-traversal epoch (the strand of a `seen` entry is that of its first sighting: which oriented vertex the entry stands for).
#pragma once
#include <cstdint>
#include <vector>

#include "ploidyfrost_hip.h"

namespace pfh {

// advice only (Linux transparent huge pages in "madvise" mode): the 2 MB-aligned part of [p, p + bytes), before its first touch
void advise_huge_pages(const void *p, size_t bytes);

struct HugeWalker {
    std::vector<uint32_t> info, seen, todo, cyc;
    uint32_t epoch = 0;
    // walks from the oriented vertex s over the CSR rows (4 slots per oriented vertex, PF_NONE = empty); fills every field
    // of `r` except list_off and returns the list the replay needs (seen[] when an exit was found, the cycle set otherwise)
    const std::vector<uint32_t> &walk(const uint32_t *succ, const uint32_t *pred, uint32_t n_unitigs, uint32_t s, pf_bfs_record &r);
};

}  // namespace pfh
