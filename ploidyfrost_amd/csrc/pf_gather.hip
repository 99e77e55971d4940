// pf_gather: the one exchange step of the path when a graph is cut over the GPUs of a node: an all-gather of a few
// 64-bit words per rank -- how many bubbles each rank called (var_count numbers bubbles across the whole run, reference
// src/CDBG.cpp:1254-1258), then every rank's slab sizes with its allele histograms and coverage counters -- over RCCL (xGMI between
// the GPUs of a node), one rank per GPU, one communicator per context.  No payload crosses ranks: with the sizes every rank writes its
// slabs straight into the shared result files (host/pf_multi.cpp, ploidyfrost_amd/dist.py).
//
// librccl is loaded when the first communicator is made (dlopen), not at start-up: a one-GPU run never pays for it.
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <cstring>
#include <mutex>
#include <string>

#include "pf_ctx.hpp"
#include "ploidyfrost_hip.h"

namespace {

struct Rccl {
    void *lib = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*AllGather)(const void *, void *, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
    std::string err;
    bool load() {
        if (lib) return true;
        for (const char *name : {"librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so"}) {
            lib = dlopen(name, RTLD_NOW | RTLD_LOCAL);
            if (lib) break;
        }
        if (!lib) { err = std::string("librccl.so cannot be loaded: ") + dlerror(); return false; }
        GetUniqueId = reinterpret_cast<decltype(GetUniqueId)>(dlsym(lib, "ncclGetUniqueId"));
        CommInitRank = reinterpret_cast<decltype(CommInitRank)>(dlsym(lib, "ncclCommInitRank"));
        AllGather = reinterpret_cast<decltype(AllGather)>(dlsym(lib, "ncclAllGather"));
        CommDestroy = reinterpret_cast<decltype(CommDestroy)>(dlsym(lib, "ncclCommDestroy"));
        GetErrorString = reinterpret_cast<decltype(GetErrorString)>(dlsym(lib, "ncclGetErrorString"));
        if (!GetUniqueId || !CommInitRank || !AllGather || !CommDestroy) { err = "librccl.so lacks the collectives this layer calls"; return false; }
        return true;
    }
};
Rccl &rccl() {
    static Rccl r;
    return r;
}
std::mutex g_mu;

struct Comm {
    ncclComm_t comm = nullptr;
    int rank = 0, world = 1;
    uint64_t *d_in = nullptr, *d_out = nullptr;
};
constexpr uint32_t GATHER_MAX = 64;   // words per rank and call

}  // namespace

namespace pf {
void comm_destroy(pf_ctx *ctx) {
    Comm *c = static_cast<Comm *>(ctx->comm);
    if (!c) return;
    if (c->comm) rccl().CommDestroy(c->comm);
    (void)hipFree(c->d_in);
    (void)hipFree(c->d_out);
    delete c;
    ctx->comm = nullptr;
}
}  // namespace pf

extern "C" {

int pf_comm_unique_id(unsigned char id[PF_COMM_ID_BYTES]) {
    if (!id) return PF_ERR_ARG;
    std::lock_guard<std::mutex> lk(g_mu);
    static_assert(PF_COMM_ID_BYTES == NCCL_UNIQUE_ID_BYTES, "id size");
    if (!rccl().load()) return PF_ERR_HIP;
    ncclUniqueId u;
    if (rccl().GetUniqueId(&u) != ncclSuccess) { rccl().err = "ncclGetUniqueId failed"; return PF_ERR_HIP; }
    memcpy(id, u.internal, NCCL_UNIQUE_ID_BYTES);
    return PF_OK;
}

int pf_comm_init(pf_ctx *ctx, const unsigned char id[PF_COMM_ID_BYTES], int rank, int world) {
    if (!ctx || !id || world < 1 || rank < 0 || rank >= world) return PF_ERR_ARG;
    pf::comm_destroy(ctx);
    {
        std::lock_guard<std::mutex> lk(g_mu);
        if (!rccl().load()) { pf::CtxErr{ctx} = "pf_comm_init: " + rccl().err; return PF_ERR_HIP; }
    }
    if (hipSetDevice(ctx->device) != hipSuccess) { pf::CtxErr{ctx} = "pf_comm_init: hipSetDevice failed"; return PF_ERR_HIP; }
    Comm *c = new Comm;
    c->rank = rank;
    c->world = world;
    ncclUniqueId u;
    memcpy(u.internal, id, NCCL_UNIQUE_ID_BYTES);
    const ncclResult_t r = rccl().CommInitRank(&c->comm, world, u, rank);
    if (r != ncclSuccess) {
        pf::CtxErr{ctx} = std::string("pf_comm_init: ncclCommInitRank: ") + (rccl().GetErrorString ? rccl().GetErrorString(r) : "failed") +
                          " (one rank per GPU: RCCL refuses two ranks on one device)";
        delete c;
        return PF_ERR_HIP;
    }
    if (hipMalloc(reinterpret_cast<void **>(&c->d_in), GATHER_MAX * 8) != hipSuccess ||
        hipMalloc(reinterpret_cast<void **>(&c->d_out), (size_t)GATHER_MAX * 8 * (size_t)world) != hipSuccess) {
        pf::CtxErr{ctx} = "pf_comm_init: out of device memory";
        ctx->comm = c;
        pf::comm_destroy(ctx);
        return PF_ERR_HIP;
    }
    ctx->comm = c;
    return PF_OK;
}

int pf_gather(pf_ctx *ctx, const uint64_t *mine, uint32_t n, uint64_t *all) {
    if (!ctx || !mine || !all || n == 0 || n > GATHER_MAX) return PF_ERR_ARG;
    Comm *c = static_cast<Comm *>(ctx->comm);
    if (!c) { pf::CtxErr{ctx} = "pf_gather: pf_comm_init first"; return PF_ERR_ARG; }
    if (hipSetDevice(ctx->device) != hipSuccess) return PF_ERR_HIP;
    hipStream_t st = ctx->stream;
    if (hipMemcpyAsync(c->d_in, mine, (size_t)n * 8, hipMemcpyHostToDevice, st) != hipSuccess) { pf::CtxErr{ctx} = "pf_gather: upload failed"; return PF_ERR_HIP; }
    const ncclResult_t r = rccl().AllGather(c->d_in, c->d_out, n, ncclUint64, c->comm, st);
    if (r != ncclSuccess) { pf::CtxErr{ctx} = std::string("pf_gather: ncclAllGather: ") + (rccl().GetErrorString ? rccl().GetErrorString(r) : "failed"); return PF_ERR_HIP; }
    if (hipMemcpyAsync(all, c->d_out, (size_t)n * 8 * (size_t)c->world, hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess) {
        pf::CtxErr{ctx} = "pf_gather: the gathered words did not come back";
        return PF_ERR_HIP;
    }
    return PF_OK;
}

void pf_comm_destroy(pf_ctx *ctx) {
    if (ctx) pf::comm_destroy(ctx);
}

}  // extern "C"
