// See pf_multi.hpp.
#include "pf_multi.hpp"

#include <cerrno>
#include <csignal>
#include <cstdlib>
#include <cstring>
#include <sys/prctl.h>
#include <sys/socket.h>
#include <sys/wait.h>
#include <unistd.h>

#include "ploidyfrost_hip.h"

namespace pfh {

namespace {
bool write_all(int fd, const void *p, size_t n) {
    const char *c = static_cast<const char *>(p);
    while (n) {
        const ssize_t w = send(fd, c, n, MSG_NOSIGNAL);   // a peer that is gone is an error to report, not a SIGPIPE
        if (w < 0) { if (errno == EINTR) continue; return false; }
        c += w;
        n -= (size_t)w;
    }
    return true;
}
bool read_all(int fd, void *p, size_t n) {
    char *c = static_cast<char *>(p);
    while (n) {
        const ssize_t r = read(fd, c, n);
        if (r < 0) { if (errno == EINTR) continue; return false; }
        if (r == 0) return false;   // the peer is gone
        c += r;
        n -= (size_t)r;
    }
    return true;
}
}  // namespace

bool RankGroup::start(int world_size) {
    world = world_size;
    rank = 0;
    share_gpu = getenv("PF_SHARE_GPU") != nullptr;
    if (world <= 1) return true;
    std::vector<int> mine;
    for (int r = 1; r < world; ++r) {
        int sv[2];
        if (socketpair(AF_UNIX, SOCK_STREAM, 0, sv) != 0) { err = "socketpair failed"; world = r; break; }
        const pid_t pid = fork();
        if (pid < 0) { close(sv[0]); close(sv[1]); err = "fork failed"; world = r; break; }
        if (pid == 0) {
            prctl(PR_SET_PDEATHSIG, SIGTERM);   // a rank 0 that dies takes the others with it
            close(sv[0]);
            for (int fd : mine) close(fd);
            rank = r;
            peers.assign(1, sv[1]);
            children.clear();
            return true;
        }
        close(sv[1]);
        mine.push_back(sv[0]);
        children.push_back((int)pid);
    }
    peers = mine;
    if (!err.empty()) {   // not every rank exists: the ones that do are told to leave (their sockets close)
        for (int fd : peers) close(fd);
        peers.clear();
        return false;
    }
    return true;
}

bool RankGroup::connect(pf_ctx *ctx) {
    if (world <= 1 || share_gpu) return true;
    unsigned char id[PF_COMM_ID_BYTES];
    memset(id, 0, sizeof id);
    bool ok = true;
    if (rank == 0 && pf_comm_unique_id(id) != PF_OK) { err = "pf_comm_unique_id failed (librccl)"; ok = false; }
    if (!agree(ok, "communicator id")) return false;   // (nobody waits in ncclCommInitRank for an id that does not exist)
    if (rank == 0) {
        for (int fd : peers)
            if (!write_all(fd, id, sizeof id)) { err = "a rank left before the communicator was made"; ok = false; }
    } else if (!read_all(peers[0], id, sizeof id)) {
        err = "rank 0 left before the communicator was made";
        ok = false;
    }
    if (!agree(ok, "communicator")) return false;
    if (pf_comm_init(ctx, id, rank, world) != PF_OK) { err = pf_last_error(ctx); ok = false; }
    return agree(ok, "communicator ready");
}

// Everybody or nobody goes on.  RCCL has no timeout: a rank that entered ncclCommInitRank or an all-gather waits for ever for a peer
// that left after an error of its own (a k-mer of its slice in no database, no memory on its device ...).  So before every step that
// ends in a collective the ranks tell rank 0 over the socket pairs whether they are still well, and rank 0 tells them whether
// everybody is; a socket that closed counts as a rank that is not.
bool RankGroup::agree(bool ok, const char *stage) {
    if (const char *e = getenv("PF_FAIL_RANK")) {   // test seam: "<rank>:<stage>" makes that rank report a failure of its own there
        const std::string want = std::to_string(rank) + ":" + stage;
        if (want == e) { ok = false; err = std::string("injected failure at ") + stage; }
    }
    if (world <= 1) return ok;
    unsigned char mine = ok ? 1 : 0, all = mine;
    if (rank == 0) {
        for (int fd : peers) {
            unsigned char theirs = 0;
            if (!read_all(fd, &theirs, 1)) theirs = 0;
            all &= theirs;
        }
        for (int fd : peers) (void)write_all(fd, &all, 1);
    } else if (!write_all(peers[0], &mine, 1) || !read_all(peers[0], &all, 1)) {
        all = 0;
    }
    if (ok && !all && err.empty()) err = std::string("another rank failed before ") + stage;
    return all != 0;
}

bool RankGroup::gather(pf_ctx *ctx, const uint64_t *mine, uint32_t n, uint64_t *all) {
    if (world <= 1) { memcpy(all, mine, (size_t)n * 8); return true; }
    if (!share_gpu) {
        if (pf_gather(ctx, mine, n, all) != PF_OK) { err = pf_last_error(ctx); return false; }
        return true;
    }
    // ranks on one device: through rank 0, over the socket pairs
    if (rank == 0) {
        memcpy(all, mine, (size_t)n * 8);
        for (int r = 1; r < world; ++r)
            if (!read_all(peers[(size_t)r - 1], all + (size_t)r * n, (size_t)n * 8)) { err = "rank " + std::to_string(r) + " left the run"; return false; }
        for (int fd : peers)
            if (!write_all(fd, all, (size_t)n * 8 * (size_t)world)) { err = "a rank left the run"; return false; }
        return true;
    }
    if (!write_all(peers[0], mine, (size_t)n * 8) || !read_all(peers[0], all, (size_t)n * 8 * (size_t)world)) { err = "rank 0 left the run"; return false; }
    return true;
}

int RankGroup::finish() {
    if (rank != 0) return 0;
    int worst = 0;
    for (int pid : children) {
        int st = 0;
        while (waitpid((pid_t)pid, &st, 0) < 0 && errno == EINTR) {}
        const int code = WIFEXITED(st) ? WEXITSTATUS(st) : 128 + (WIFSIGNALED(st) ? WTERMSIG(st) : 0);
        if (code && !worst) worst = code;
    }
    children.clear();
    for (int fd : peers) close(fd);
    peers.clear();
    return worst;
}

}  // namespace pfh
