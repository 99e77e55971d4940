// The joined count table of the colored path on the device (pf_colored.hip builds it): slot = { u64 key, u32 count[C] } padded to a
// power-of-two stride, open addressing; an absent (k-mer, colour) pair is the all-ones count.  Shared by K-COV-C / K-STRCOV-C
// (pf_colored.hip) and the colored K-SITES of the resident calling pipeline (pf_call.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "pf_device_common.hpp"

namespace pf {

constexpr uint32_t CTAB_MISSING = 0xFFFFFFFFu;
constexpr int CPP = 4;  // colours per pass of K-COV-C

struct CTab {
    const uint8_t *base;
    uint64_t mask;
    uint32_t shift;  // log2(stride bytes)
};

__device__ inline const uint8_t *ctab_slot(const CTab &t, uint64_t i) { return t.base + (i << t.shift); }
__device__ inline uint32_t ctab_line_shift(uint32_t shift) { return shift < 7 ? 7 - shift : 0; }   // log2(slots of a 128-B line)

// The table is addressed like the single-sample one (pf_device_common.hpp, "lines ... addressed by the key's minimizer"): the
// slots of a 128-B line (four at three colours) form a bucket, LINE_TRIES buckets by double hashing of the minimizer, then the
// slots from mix64(key) on.  `mask` counts slots; buckets are aligned groups of them.
struct CSeq {
    LineSeq q;       // in buckets
    uint32_t ls;     // log2(slots of a bucket)
};
__device__ inline CSeq ctab_seq(const CTab &t, uint64_t fwd, uint64_t rc, int k) {
    const uint32_t ls = ctab_line_shift(t.shift);
    return CSeq{kmer_lines(fwd, rc, k, t.mask >> ls), ls};
}
__device__ inline uint64_t ctab_bucket(const CTab &t, const CSeq &sq, int i) { return seq_line(sq.q, i, t.mask >> sq.ls) << sq.ls; }   // its first slot

// slot of an exact key, or nullptr
__device__ inline const uint8_t *ctab_find(const CTab &t, uint64_t key, const CSeq &sq) {
    const uint32_t per = 1u << sq.ls;
    for (int i = 0; i < LINE_TRIES; ++i) {
        const uint64_t b = ctab_bucket(t, sq, i);
        for (uint32_t j = 0; j < per; ++j) {
            const uint8_t *s = ctab_slot(t, b + j);
            const uint64_t kx = *reinterpret_cast<const uint64_t *>(s);
            if (kx == key) return s;
            if (kx == EMPTY_KEY) return nullptr;
        }
    }
    for (uint64_t i = mix64(key) & t.mask;; i = (i + 1) & t.mask) {
        const uint8_t *s = ctab_slot(t, i);
        const uint64_t kx = *reinterpret_cast<const uint64_t *>(s);
        if (kx == key) return s;
        if (kx == EMPTY_KEY) return nullptr;
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
    const CSeq home = ctab_seq(t, fwd, rc, k);
    const uint8_t *a = ctab_find(t, first, home);
    const uint8_t *b = (a && one_strand) ? nullptr : ctab_find(t, first == fwd ? rc : fwd, home);
#pragma unroll
    for (int j = 0; j < CPP; ++j) {
        uint32_t v = CTAB_MISSING;
        if ((uint32_t)j < n_here) {
            if (a) v = *reinterpret_cast<const uint32_t *>(a + 8 + 4 * (c0 + j));
            if (v == CTAB_MISSING && b) v = *reinterpret_cast<const uint32_t *>(b + 8 + 4 * (c0 + j));
        }
        out[j] = v;
    }
}


// both slots of the composite lookup at once, for callers that read many colours of one k-mer: count of colour c =
// ctab_count(a, b, c)
__device__ inline void colored_slots(const CTab &t, uint64_t fwd, int k, bool one_strand, const uint8_t *&a, const uint8_t *&b) {
    const uint64_t rc = rc_kmer(fwd, k);
    const uint64_t first = (one_strand && rc < fwd) ? rc : fwd;
    const CSeq home = ctab_seq(t, fwd, rc, k);
    a = ctab_find(t, first, home);
    b = (a && one_strand) ? nullptr : ctab_find(t, first == fwd ? rc : fwd, home);
}
__device__ inline uint32_t ctab_count(const uint8_t *a, const uint8_t *b, uint32_t c) {
    uint32_t v = CTAB_MISSING;
    if (a) v = *reinterpret_cast<const uint32_t *>(a + 8 + 4 * c);
    if (v == CTAB_MISSING && b) v = *reinterpret_cast<const uint32_t *>(b + 8 + 4 * c);
    return v;
}

}  // namespace pf
