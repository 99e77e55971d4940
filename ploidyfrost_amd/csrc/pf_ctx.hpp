// Context behind the C ABI (include/ploidyfrost_hip.h): everything resident in HBM for one GPU.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <string>
#include <vector>

#include "pf_device_common.hpp"

namespace pf {
struct TimedLaunch {
    int kernel;
    hipEvent_t a, b;
};
}  // namespace pf

struct pf_ctx {
    int device = 0;
    int n_cu = 256;
    std::string name;
    std::string err;
    hipStream_t own_stream = nullptr, stream = nullptr;

    // graph SoA (HBM): 2-bit packed unitigs, word offsets, lengths; CSR adjacency
    uint64_t *d_seq = nullptr;
    uint64_t *d_off = nullptr;
    uint32_t *d_len = nullptr;
    uint32_t N = 0;
    int k = 0;
    uint64_t n_words = 0, n_kmers = 0;
    uint32_t *d_succ = nullptr, *d_pred = nullptr;  // [2N][4]
    bool has_adj = false;
    uint32_t *d_cand = nullptr;    // oriented vertices with out-degree > 1, ascending
    std::vector<uint32_t> h_cand;  // host copy (shard range queries)

    // k-mer count table (HBM): open addressing, 16-B slots, capacity = power of two >= 2n
    pf::Slot *d_tab = nullptr;
    uint64_t tab_cap = 0, tab_n = 0;

    // reusable result staging for host-pointer callers
    uint64_t *d_cov_sum = nullptr;
    uint32_t *d_cov_min = nullptr;
    uint8_t *d_cov_miss = nullptr;
    uint32_t cov_cap = 0;

    unsigned int bfs_deferred = 0;  // candidates of the last pf_bfs_candidates that needed the big tier

    bool timing = false;
    std::vector<pf::TimedLaunch> launches;
};

namespace pf {
int ctx_begin(pf_ctx *ctx, int kernel);
void ctx_end(pf_ctx *ctx);
int ctx_grid(const pf_ctx *ctx, uint64_t work_items, int block, int per_cu);
}  // namespace pf
