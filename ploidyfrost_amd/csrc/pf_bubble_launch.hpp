// K-BUBBLE launch interface shared by pf_bubble.hip (pf_align_bubbles: batches described by the host) and pf_call.hip (the
// resident calling pipeline: paths, tasks and per-class work queues produced on the device).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "pf_align_dev.hpp"
#include "ploidyfrost_hip.h"

struct pf_ctx;

namespace pf {

// LDS size classes of the Needleman-Wunsch working storage (bytes per wavefront): the finer the classes, the more wavefronts of
// the mid-sized bubbles fit a CU's 160 KB (a 100 x 100 matrix needs 16 KB: ten per CU, where a 20 KB class gave seven).
// Class kBubLdsClasses = global memory.
// (round 4: eight classes instead of six; and, with the direction matrix at four bits a cell -- pf_align_dev.hpp -- every boundary at
// half of what it was, so that a bubble sits in the class it sat in before with twice as many wavefronts of that class fitting a CU:
// a 100 x 100 bubble needs 7.4 KB where it needed 12.5.  Halving the matrix WITHOUT moving the boundaries put the heavy bubbles
// into the classes of the many medium ones and made the metric's config slower: profiles/r16_experiments.txt.)
// A ninth class, 64 KB, takes what used to leave for the global-memory tier (matrices of 64 - 128 KB at a byte a cell).
constexpr int kBubLdsClasses = 9;
constexpr uint64_t kBubClassBytes[kBubLdsClasses] = {3072, 4608, 6 * 1024, 8 * 1024, 10 * 1024, 12 * 1024, 20 * 1024, 32 * 1024, 64 * 1024};
constexpr int kBubMaxClasses = 12;   // (sizes of the per-class streams and events of a context)
__host__ __device__ inline int bubble_class_waves_per_cu(int c) {
    const int fit = (int)((160 * 1024) / kBubClassBytes[c]);
    return fit > 24 ? 24 : fit;
}

// working storage a bubble whose first path has l0 characters and whose longest has lmax may need: row 0 can grow by the gaps
// opened in later rounds, so there is headroom; the device re-checks every job against its tier
__host__ __device__ inline uint64_t bubble_need(uint32_t l0, uint32_t lmax) {
    return job_bytes((l0 > lmax ? l0 : lmax) + 32, lmax);
}
__host__ __device__ inline int bubble_class(uint32_t l0, uint32_t lmax) {
    const uint64_t need = bubble_need(l0, lmax);
    for (int x = 0; x < kBubLdsClasses; ++x)
        if (need <= kBubClassBytes[x]) return x;
    return kBubLdsClasses;
}

struct BubbleLaunch {
    // inputs, device memory
    const char *text = nullptr;
    const pf_bubble_path *paths = nullptr;
    const pf_bubble_task *tasks = nullptr;
    uint32_t n_tasks = 0;             // size of tasks / res (queues may list fewer)
    const uint32_t *idx = nullptr;    // class queues back to back
    uint32_t n_cls[kBubLdsClasses + 1] = {};
    uint64_t max_need = 0;            // largest bubble_need in the global-memory class
    uint64_t retry_need = 0;          // working storage bound for the retry tier
    double match = 2, mismatch = -1, gap = -3;
    // outputs, device memory
    pf_bubble_result *res = nullptr;
    char *otext = nullptr;
    pf_bubble_site *osites = nullptr;
    uint8_t *ogroups = nullptr;
    uint32_t *oilen = nullptr;
    uint64_t text_cap = 0, site_cap = 0, group_cap = 0, ilen_cap = 0;
    // the pool heads (bubble_pool_heads) already hold what an earlier kernel of the caller took from the pools: do not reset them
    bool keep_heads = false;
    // which set of K-BUBBLE's workspaces, class streams and events (the lane of the resident pipeline's pf_call_align_lane: calls on
    // different lanes run side by side), and the stream the launch is ordered on (nullptr: the context's)
    int lane = 0;
    hipStream_t stream = nullptr;
};

// device address of the four pool heads {text bytes, sites, group bytes, ilen entries} that K-BUBBLE bumps (first 32 bytes of a
// 128-byte block the launch otherwise zeroes): a caller that publishes some bubbles itself allocates from the same heads
unsigned long long *bubble_pool_heads(pf_ctx *ctx, int lane = 0);

// The single-SNP shortcut of K-BUBBLE as a predicate on the scores (proof in pf_bubble.hip): two equally long paths that differ
// in exactly one base align as themselves, one SNP column, groups {1, 2}, whenever this holds.
inline bool snp_shortcut_scores(double M, double D, double G) {
    if (!(M < 1e6 && M > -1e6 && D < 1e6 && D > -1e6 && G < 1e6 && G > -1e6)) return false;
    const bool integral = M == (double)(long long)M && D == (double)(long long)D && G == (double)(long long)G;
    return integral && M >= D && M + 1 >= 2 * (G + 1) && D - 2 * G - 2 > 0;
}

int bubble_launch(pf_ctx *ctx, const BubbleLaunch &L, unsigned long long heads[4]);
int bubble_reserve(pf_ctx *ctx, uint32_t n_tasks, int lane = 0);

}  // namespace pf
