// One graph over the GPUs of a node, from the C++ side (the reference boundary it sits behind: src/Main.cpp:829-849).
//
// `ploidyfrost ... --gpus N` forks N - 1 processes BEFORE anything touches the GPU: one process per GPU, rank r on device r.  Every
// rank loads the graph and the count table (replicated in its GPU's HBM), runs findSuperBubble and the owner scan -- they give every
// rank the same bubble list in output order -- and aligns, formats and writes its contiguous slice of that list straight into the
// shared result files.  Two small all-gathers are all the ranks exchange (pf_gather: RCCL over xGMI): the bubbles each rank called,
// then slab sizes + allele histograms + coverage counters.  Rank 0 writes Unitig_Id.txt and super_bubble.txt and prints the summary.
// The same protocol as ploidyfrost_amd/dist.py plays over torch.distributed for bench.py.
//
// PF_SHARE_GPU=1 (tests on a one-GPU box): every rank uses device 0 and the words travel over the socket pairs that exist for the
// communicator's id anyway, because RCCL refuses two ranks on one device.
#pragma once
#include <cstdint>
#include <string>
#include <vector>

struct pf_ctx;

namespace pfh {

struct RankGroup {
    int rank = 0, world = 1;
    bool share_gpu = false;
    std::vector<int> peers;   // rank 0: one socket per other rank (index r - 1); other ranks: peers[0] = the socket to rank 0
    std::vector<int> children;   // rank 0: pids
    std::string err;

    int device() const { return share_gpu ? 0 : rank; }
    // forks world - 1 children (call before any HIP call of the process); returns false in no process on failure paths that matter:
    // a failed fork leaves rank 0 alone with err set
    bool start(int world_size);
    // collective over the socket pairs: true iff every rank passed ok = true (a rank that is gone counts as false).  Called before
    // every step that ends in an RCCL call, which would wait for ever for a rank that left.  `stage` names the step in messages.
    bool agree(bool ok, const char *stage);
    // RCCL communicator on this rank's context (or nothing, when the ranks share a GPU)
    bool connect(pf_ctx *ctx);
    // all[r * n + i] = word i of rank r; collective
    bool gather(pf_ctx *ctx, const uint64_t *mine, uint32_t n, uint64_t *all);
    // rank 0: waits for the other ranks; returns the first non-zero exit status (0 = all well).  Other ranks: 0.
    int finish();
};

}  // namespace pfh
