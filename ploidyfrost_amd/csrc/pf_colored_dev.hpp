// The joined count table of the colored path on the device (pf_colored.hip builds it): the single-sample table's lines
// (pf_device_common.hpp: ten keys, chosen by the key's minimizer; buddy line, second pair, then the lines from mix64(key) on) with
// one count per colour and slot behind the keys -- count of colour c, slot i at byte 80 + 40 c + 4 i of the line; an absent
// (k-mer, colour) pair is the all-ones count.  A line is 80 + 40 C bytes rounded up to 128 (C = 3: 256 B).  Shared by K-COV-C /
// K-STRCOV-C (pf_colored.hip) and the colored K-SITES of the resident calling pipeline (pf_call.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "pf_device_common.hpp"

namespace pf {

constexpr uint32_t CTAB_MISSING = 0xFFFFFFFFu;
constexpr int CPP = 4;  // colours per pass of K-COV-C

struct CTab {
    const uint8_t *base;
    uint64_t mask;         // lines - 1
    uint32_t line_bytes;   // a multiple of 128
};
__host__ __device__ inline uint32_t ctab_line_bytes(uint32_t n_colors) { return (80u + 40u * n_colors + 127u) & ~127u; }

__device__ inline const uint8_t *ctab_line(const CTab &t, uint64_t line) { return t.base + line * t.line_bytes; }
__device__ inline LineKeys ctab_keys(const CTab &t, uint64_t line) {
    LineKeys k;
    const uint4 *p = reinterpret_cast<const uint4 *>(ctab_line(t, line));
#pragma unroll
    for (int i = 0; i < LINE_KEYS / 2; ++i) k.q[i] = p[i];
    return k;
}
// the count of colour 0 of slot `at` of a line; colour c's is LINE_KEYS * c words further
__device__ inline const uint32_t *ctab_counts(const CTab &t, uint64_t line, int at) {
    return reinterpret_cast<const uint32_t *>(ctab_line(t, line) + 8 * LINE_KEYS) + at;
}

// the counts of an exact key (its colour-0 count; see ctab_counts), or nullptr
__device__ inline const uint32_t *ctab_find(const CTab &t, uint64_t key, const LineSeq &sq, int first_try = 0) {
    for (int i = first_try; i < LINE_TRIES; ++i) {
        const uint64_t line = seq_line(sq, i, t.mask);
        bool open;
        const int at = line_slot(ctab_keys(t, line), key, open);
        if (at >= 0) return ctab_counts(t, line, at);
        if (open) return nullptr;
    }
    for (uint64_t line = mix64(key) & t.mask;; line = (line + 1) & t.mask) {
        bool open;
        const int at = line_slot(ctab_keys(t, line), key, open);
        if (at >= 0) return ctab_counts(t, line, at);
        if (open) return nullptr;
    }
}

// The composite lookup of CCDBG.cpp:98-103 per colour -- "if (!IsKmer(fwd)) reverse(); CheckKmer()" -- for CPP
// colours starting at c0: the count the colour's database holds for the forward k-mer, else for its reverse
// complement, else CTAB_MISSING.  When no k-mer is a key of the table in both orientations (`one_strand`) the canonical
// form is probed first and the second probe only happens for k-mers absent from the table.
__device__ inline void colored_counts(const CTab &t, uint64_t fwd, int k, bool one_strand, uint32_t c0, uint32_t n_here,
                                      uint32_t out[CPP]) {
    const uint64_t rc = rc_kmer(fwd, k);
    const uint64_t first = (one_strand && rc < fwd) ? rc : fwd;
    const LineSeq sq = kmer_lines(fwd, rc, k, t.mask);
    const uint32_t *a = ctab_find(t, first, sq);
    const uint32_t *b = (a && one_strand) ? nullptr : ctab_find(t, first == fwd ? rc : fwd, sq);
#pragma unroll
    for (int j = 0; j < CPP; ++j) {
        uint32_t v = CTAB_MISSING;
        if ((uint32_t)j < n_here) {
            if (a) v = a[LINE_KEYS * (c0 + j)];
            if (v == CTAB_MISSING && b) v = b[LINE_KEYS * (c0 + j)];
        }
        out[j] = v;
    }
}

// both slots of the composite lookup at once, for callers that read many colours of one k-mer: count of colour c =
// ctab_count(a, b, c)
__device__ inline void colored_slots(const CTab &t, uint64_t fwd, int k, bool one_strand, const uint32_t *&a, const uint32_t *&b) {
    const uint64_t rc = rc_kmer(fwd, k);
    const uint64_t first = (one_strand && rc < fwd) ? rc : fwd;
    const LineSeq sq = kmer_lines(fwd, rc, k, t.mask);
    a = ctab_find(t, first, sq);
    b = (a && one_strand) ? nullptr : ctab_find(t, first == fwd ? rc : fwd, sq);
}
__device__ inline uint32_t ctab_count(const uint32_t *a, const uint32_t *b, uint32_t c) {
    uint32_t v = CTAB_MISSING;
    if (a) v = a[LINE_KEYS * c];
    if (v == CTAB_MISSING && b) v = b[LINE_KEYS * c];
    return v;
}

}  // namespace pf
