// Which k-length unitigs Bifrost files as "abundant" k-mers while it loads a GFA file, and in which order it then
// iterates them -- the last piece of the reference's unitig numbering.
//
// `CompactedDBG::readGFA` (bifrost/src/CompactedDBG.tcc:7888-7908, the `-t 1` branch) hands every S-line to
// `addUnitig` (tcc:3928-4080), which files the unitig's minimizers in a hash map of buckets.  A k-length unitig whose
// minimizer bucket already holds >= 15 entries is not kept in `km_unitigs` but inserted into the k-mer hash table
// `h_kmers_ccov` (bifrost/src/KmerHashTable.hpp:326-354); the unitig iterator (UnitigIterator.tcc:32-58) walks the long
// unitigs, then `km_unitigs`, then that table slot by slot.  Bucket sizes depend on everything filed before, including
// the entries long unitigs redirect to their next-best minimizer once a bucket is crowded (tcc:3983-4008), so the
// decision is reproduced by replaying that bookkeeping -- but only for the buckets that can ever reach 15 entries:
//   1. parallel: every minimizer occurrence of every unitig is counted in a table of saturating counters (an upper
//      bound per bucket).  No counter at 15 => no bucket is ever crowded => nothing is abundant (the usual case; done).
//   2. otherwise the unitigs touching a bucket whose counter reached 15 ("tracked") are replayed in file order with the
//      reference's exact rules; a redirect into an untracked bucket makes it tracked and the replay starts over.
#pragma once
#include <cstdint>
#include <vector>

namespace pfh {

struct SegRef {
    const char *s;   // bases (A/C/G/T); k-length unitigs already canonical (tcc:3945-3954)
    uint32_t len;
};

struct UnitigNumbering {
    std::vector<uint32_t> abundant;   // indices into segs of the abundant k-length unitigs, in the reference's iteration order
    uint64_t tracked_buckets = 0;
    uint32_t replays = 0;
    uint64_t replayed_unitigs = 0;
};

// segs in file order; g <= k - 2 (bifrost/src/CompactedDBG.tcc:8383).  counters (optional): the saturating occurrence
// counters of step 1, slot = mix(canonical minimizer) & (size - 1) -- the table the device pass (K-MINZ) bounds from above.
// counters_in / touches_in (optional, both or neither): what the device pass already knows (pf_minimizer_replay_inputs) -- the counter
// table (upper bounds of step 1's, same geometry) and per seg whether it meets a slot that reached the limit: steps 1 and 2 of this
// function, each a pass over every unitig, are then skipped (a later round of the replay, after a redirect, still flags on its own).
void bifrost_numbering(int k, int g, const std::vector<SegRef> &segs, unsigned threads, UnitigNumbering &out,
                       std::vector<uint8_t> *counters = nullptr, const uint8_t *counters_in = nullptr, const uint8_t *touches_in = nullptr, uint64_t counters_in_len = 0);

}  // namespace pfh
