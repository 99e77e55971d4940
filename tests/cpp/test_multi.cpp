// `ploidyfrost --gpus N` around RCCL, without RCCL and without a GPU: csrc/host/pf_multi.cpp linked against stand-ins for the four
// device-layer calls it makes (pf_comm_unique_id, pf_comm_init, pf_gather, pf_last_error) -- a link-time seam in this test, not a
// switch in the product.  The stand-in all-gather runs over a shared mapping made before the fork.  Checks:
//   ok   <N>                 rank r on device r; every rank gets rank 0's communicator id and its own (rank, world); the gathered words
//   fail <N> <rank> <stage>  PF_FAIL_RANK: one rank reports a failure of its own at a stage -> every rank leaves non-zero BEFORE the
//                            next collective (no rank enters the stand-in gather after it), nothing hangs
//   die  <N> <rank>          a rank that exits without a word -> the others see its socket close at the next agreement and leave
// Built and run by tests/test_dist_cpu.py.
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <sys/mman.h>
#include <thread>
#include <unistd.h>

#include "pf_multi.hpp"
#include "ploidyfrost_hip.h"

namespace {
constexpr int MAX_RANKS = 16, MAX_ROUNDS = 8, MAX_WORDS = 32;
struct Shared {
    std::atomic<int> arrived[MAX_ROUNDS];
    std::atomic<int> gathers_entered;
    uint64_t words[MAX_ROUNDS][MAX_RANKS][MAX_WORDS];
    unsigned char id_seen[MAX_RANKS][PF_COMM_ID_BYTES];
    int rank_seen[MAX_RANKS], world_seen[MAX_RANKS];
};
Shared *shm = nullptr;
int my_rank = -1, my_world = 0, my_round = 0;
}  // namespace

extern "C" {
int pf_comm_unique_id(unsigned char *id) {
    for (int i = 0; i < PF_COMM_ID_BYTES; ++i) id[i] = (unsigned char)(0x5A ^ (i * 7 + 1));
    return PF_OK;
}
int pf_comm_init(pf_ctx *, const unsigned char *id, int rank, int world) {
    my_rank = rank;
    my_world = world;
    memcpy(shm->id_seen[rank], id, PF_COMM_ID_BYTES);
    shm->rank_seen[rank] = rank;
    shm->world_seen[rank] = world;
    return PF_OK;
}
int pf_gather(pf_ctx *, const uint64_t *mine, uint32_t n, uint64_t *all) {
    shm->gathers_entered.fetch_add(1);
    const int r = my_round++;
    if (r >= MAX_ROUNDS || n > MAX_WORDS) return PF_ERR_ARG;
    memcpy(shm->words[r][my_rank], mine, (size_t)n * 8);
    shm->arrived[r].fetch_add(1);
    const auto t0 = std::chrono::steady_clock::now();
    while (shm->arrived[r].load() < my_world) {   // what RCCL does for ever; here: ten seconds, then the test fails
        if (std::chrono::steady_clock::now() - t0 > std::chrono::seconds(10)) return PF_ERR_HIP;
        std::this_thread::yield();
    }
    for (int q = 0; q < my_world; ++q) memcpy(all + (size_t)q * n, shm->words[r][q], (size_t)n * 8);
    return PF_OK;
}
const char *pf_last_error(const pf_ctx *) { return "stand-in: a rank never arrived at the all-gather"; }
}

int main(int argc, char **argv) {
    if (argc < 3) return 2;
    const std::string mode = argv[1];
    const int world = atoi(argv[2]);
    const int who = argc > 3 ? atoi(argv[3]) : -1;
    shm = static_cast<Shared *>(mmap(nullptr, sizeof(Shared), PROT_READ | PROT_WRITE, MAP_SHARED | MAP_ANONYMOUS, -1, 0));
    if (shm == MAP_FAILED) return 2;
    memset(shm, 0, sizeof(Shared));
    unsetenv("PF_SHARE_GPU");
    if (mode == "fail") setenv("PF_FAIL_RANK", (std::string(argv[3]) + ":" + argv[4]).c_str(), 1);
    pfh::RankGroup ranks;
    if (!ranks.start(world)) { fprintf(stderr, "start: %s\n", ranks.err.c_str()); return 2; }
    pf_ctx *ctx = reinterpret_cast<pf_ctx *>(0x1);   // never looked into by the stand-ins
    auto leave = [&](const char *why) {
        fprintf(stderr, "rank %d: %s (%s)\n", ranks.rank, why, ranks.err.c_str());
        if (ranks.rank == 0) {
            (void)ranks.finish();
            fprintf(stderr, "gathers entered: %d\n", shm->gathers_entered.load());
        }
        _exit(1);
    };
    if (ranks.device() != ranks.rank) leave("rank r is not on device r");
    if (!ranks.agree(true, "load")) leave("load");
    if (!ranks.connect(ctx)) leave("connect");
    if (mode == "die" && ranks.rank == who) _exit(3);   // gone without a word
    if (!ranks.agree(true, "findSuperBubble")) leave("findSuperBubble");
    uint64_t mine[3] = {(uint64_t)ranks.rank * 100, (uint64_t)ranks.rank * 100 + 1, 7}, all[MAX_RANKS * 3];
    if (!ranks.agree(true, "align")) leave("align");
    if (!ranks.gather(ctx, mine, 1, all)) leave("gather 1");
    for (int r = 0; r < world; ++r)
        if (all[r] != (uint64_t)r * 100) leave("gathered words (1)");
    if (!ranks.agree(true, "text")) leave("text");
    if (!ranks.gather(ctx, mine, 3, all)) leave("gather 2");
    for (int r = 0; r < world; ++r)
        if (all[r * 3] != (uint64_t)r * 100 || all[r * 3 + 1] != (uint64_t)r * 100 + 1 || all[r * 3 + 2] != 7) leave("gathered words (2)");
    if (!ranks.agree(true, "write")) leave("write");
    if (ranks.rank != 0) _exit(0);
    const int rc = ranks.finish();
    if (rc) { fprintf(stderr, "a rank ended with status %d\n", rc); return rc; }
    unsigned char want[PF_COMM_ID_BYTES];
    pf_comm_unique_id(want);
    for (int r = 0; r < world; ++r) {
        if (memcmp(shm->id_seen[r], want, PF_COMM_ID_BYTES)) { fprintf(stderr, "rank %d did not get rank 0's id\n", r); return 1; }
        if (shm->rank_seen[r] != r || shm->world_seen[r] != world) { fprintf(stderr, "rank %d initialised as (%d, %d)\n", r, shm->rank_seen[r], shm->world_seen[r]); return 1; }
    }
    printf("OK %d ranks, %d gathers entered\n", world, shm->gathers_entered.load());
    return 0;
}
