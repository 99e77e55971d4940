// Batched SeqAlign::SequenceAlignment (reference src/SeqAlign.cpp:550-640) for many bubbles at
// once.  Every O(m*n) fill and every traceback runs on the GPU (pf_align_batch, one wavefront per
// pairwise job); the host only sequences the rounds of the progressive alignment (row 3, 4, ...
// against row 0 of every alignment kept so far), re-opens gaps in the older rows, and applies
// the selection ladder of compareStrPair (src/SeqAlign.cpp:8-236).
#pragma once
#include <cstdint>
#include <string>
#include <vector>

#include "ploidyfrost_hip.h"

namespace pfh {

struct Scoring {
    double match = 2, mismatch = -1, gap = -3;
};

struct Msa {
    std::vector<std::string> rows;  // empty: no alignment survived, the bubble is skipped
    std::vector<uint32_t> snp_pos, indel_pos, indel_len;
    std::vector<uint16_t> group;  // [col * rows.size() + row]: 1-based allele group, 0 = not a site
    uint32_t n_cols = 0;
    uint16_t group_at(uint32_t col, uint32_t row) const { return group[(size_t)col * rows.size() + row]; }
};

struct AlignStats {
    uint64_t jobs = 0, rounds = 0, hits = 0;
    double build_s = 0, device_s = 0, post_s = 0, choose_s = 0;
};

// Owns the pinned exchange buffers that are reused from call to call.
class Aligner {
public:
    explicit Aligner(pf_ctx *ctx);
    ~Aligner();
    Aligner(const Aligner &) = delete;
    Aligner &operator=(const Aligner &) = delete;
    // paths[t] = the (already sorted) path strings of bubble t, at least two each.
    // Returns PF_OK or the device status (message in err).
    int align(const Scoring &sc, const std::vector<std::vector<std::string>> &paths, std::vector<Msa> &out, AlignStats *stats,
              std::string &err, unsigned threads = 1);

private:
    struct Impl;
    Impl *impl_;
    pf_ctx *ctx_;
};

}  // namespace pfh
