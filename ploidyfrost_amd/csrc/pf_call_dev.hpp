// Device-side pieces of the resident calling pipeline (pf_call.hip): records shared by its kernels, the packed-sequence
// helpers (oriented base, lexicographic compare of two stored unitigs) and the path quicksort of the strict branch.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "pf_device_common.hpp"

namespace pf {

// one bubble to call (reference: the locals of CDBG::ploidyEstimation_ptr's strict / branching branches,
// src/CDBG.cpp:1187-1344, 1345-1655), produced by K-SCAN for every endpoint side that would own its bubble
struct CallTask {
    uint32_t u;            // owner endpoint (unitig index)
    uint32_t entrance_ov, exit_ov;
    uint8_t strict, n_inner, n_cov, pad_;
    uint32_t inner[4];     // strict: inner oriented unitigs, sorted (sortSeq_simple)
    double core_mean, cov_sum;
    double cov[4];         // strict: their mean coverages, same order
};

// base `idx` of oriented unitig ov (mappedSequenceToString()[idx]): 0..3 = A C G T
__device__ inline uint32_t oriented_base(const uint64_t *__restrict__ seq, const uint64_t *__restrict__ off,
                                         const uint32_t *__restrict__ len, uint32_t ov, uint32_t idx) {
    const uint32_t u = ov >> 1;
    const uint64_t *w = seq + off[u];
    const bool rev = (ov & 1) != 0;
    const uint32_t j = rev ? len[u] - 1 - idx : idx;
    const uint32_t b = (uint32_t)(w[j >> 5] >> (62 - 2 * (j & 31))) & 3u;
    return rev ? 3 - b : b;
}

// reverse complement of 32 packed bases (first base most significant)
__device__ inline uint64_t revcomp32(uint64_t x) {
    uint64_t r = __brevll(x);                                                        // bit order reversed: pairs reversed and swapped inside
    r = ((r & 0xAAAAAAAAAAAAAAAAull) >> 1) | ((r & 0x5555555555555555ull) << 1);     // swap back inside every pair
    return ~r;
}

// bases [32 c, 32 c + 32) of an oriented unitig as one packed word (first base most significant, zero beyond the end):
// w = its stored words, L = its length.  Two loads at most -- against three per base through oriented_base.
__device__ inline uint64_t oriented_chunk(const uint64_t *__restrict__ w, uint32_t L, bool rev, uint32_t c) {
    if (!rev) return w[c];
    const int32_t hi = (int32_t)L - 32 * (int32_t)c;   // stored bases [hi - 32, hi) in reverse
    if (hi <= 0) return 0;
    if (hi < 32) {
        const uint64_t x = w[0] >> (2 * (32 - hi));    // bases 0 .. hi-1, right-aligned
        return revcomp32(x) & (~0ull << (2 * (32 - hi)));
    }
    const uint32_t lo = (uint32_t)hi - 32, sh = 2 * (lo & 31);
    uint64_t x = w[lo >> 5] << sh;
    if (sh) x |= w[(lo >> 5) + 1] >> (64 - sh);        // (lo & 31 != 0 and hi <= L: the next word belongs to the unitig)
    return revcomp32(x);
}

// strcmp(referenceUnitigToString(a), referenceUnitigToString(b)): words hold the first base most significant with zero
// padding, so whole words compare like the strings; the last word is masked to the shorter length and a proper prefix is
// the smaller string.
__device__ inline int unitig_cmp(const uint64_t *__restrict__ seq, const uint64_t *__restrict__ off, const uint32_t *__restrict__ len,
                                 uint32_t a, uint32_t b) {
    if (a == b) return 0;
    const uint32_t la = len[a], lb = len[b], L = la < lb ? la : lb;
    const uint64_t *wa = seq + off[a], *wb = seq + off[b];
    const uint32_t nw = (L + 31) >> 5;
    for (uint32_t i = 0; i < nw; ++i) {
        uint64_t x = wa[i], y = wb[i];
        if (i == nw - 1 && (L & 31)) {
            const uint64_t m = ~0ull << (64 - 2 * (L & 31));
            x &= m;
            y &= m;
        }
        if (x != y) return x < y ? -1 : 1;
    }
    return la < lb ? -1 : (la > lb ? 1 : 0);
}

// sortSeq_simple (reference src/CDBG.cpp:482-551): the reference's own non-stable quicksort -- descending mean coverage, ties
// by descending reference string -- with its exact partition scheme (the swap sequence decides how full ties come out).
// n <= 4, so the recursion is an explicit stack of ranges.
__device__ inline void sort_inner_dev(const uint64_t *__restrict__ seq, const uint64_t *__restrict__ off, const uint32_t *__restrict__ len,
                                      double *cov, uint32_t *ov, int n) {
    int stack_lo[8], stack_hi[8];
    int sp = 0;
    stack_lo[0] = 0;
    stack_hi[0] = n - 1;
    sp = 1;
    while (sp > 0) {
        --sp;
        const int low = stack_lo[sp], high = stack_hi[sp];
        if (high <= low) continue;
        int i = low, j = high;
        for (;;) {
            while (cov[i] >= cov[low]) {
                if (cov[i] > cov[low] || unitig_cmp(seq, off, len, ov[i] >> 1, ov[low] >> 1) > 0) i++;
                else break;
                if (i == high) break;
            }
            while (cov[j] <= cov[low]) {
                if (cov[j] < cov[low] || unitig_cmp(seq, off, len, ov[j] >> 1, ov[low] >> 1) < 0) j--;
                else break;
                if (j == low) break;
            }
            if (i >= j) break;
            const double tc = cov[i]; cov[i] = cov[j]; cov[j] = tc;
            const uint32_t to = ov[i]; ov[i] = ov[j]; ov[j] = to;
        }
        {
            const double tc = cov[low]; cov[low] = cov[j]; cov[j] = tc;
            const uint32_t to = ov[low]; ov[low] = ov[j]; ov[j] = to;
        }
        // (at most n - 1 ranges of two or more elements can ever be pending; empty and single ones pop at once)
        stack_lo[sp] = low; stack_hi[sp] = j - 1; ++sp;
        stack_lo[sp] = j + 1; stack_hi[sp] = high; ++sp;
    }
}

}  // namespace pf
