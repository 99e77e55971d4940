// K-MINZ: can any bucket of Bifrost's minimizer index get crowded while this graph is read?  (GFA ingest, unitig numbering.)
//
// Bifrost numbers a k-length unitig last, as an "abundant" k-mer, when the bucket of its minimizer already holds 15 entries
// (bifrost/src/CompactedDBG.tcc:3928-4080); host/pf_host_minz.cpp replays that bookkeeping exactly -- but only has to when
// some bucket can reach 15 entries at all, which no ordinary graph does.  Deciding that needs every minimizer occurrence
// of every unitig counted: one rolling hash per k-mer on the host (0.8 s for 233 M k-mers on 64 threads), a millisecond here.
//
// A g-mer position q of a unitig is counted when its hash equals the minimum over the window of some k-mer that contains
// it (positions p+1 .. p+k-g-1 of k-mer p: a minimizer may not sit at either end, minHashIterator.hpp:63-119 with nh = true).
// That is every position the reference's iterator reports and, where hashes tie inside a window, possibly a few more: the
// counts are upper bounds of the host's, so "no slot reaches 15" here implies the same there, never the other way round.
// Slot = mix64(canonical minimizer) & (slots - 1), the same table geometry as the host pass.
//
// One wavefront per unitig, 64 consecutive g-mer positions per step: each lane hashes one g-mer from the 2-bit sequence
// (RepHash: two rolling words, wyhash of the ordered pair), window minima and the membership test run on wave shuffles,
// counted positions go to the table with one atomic each.
#include <hip/hip_runtime.h>

#include <string>

#include "../../include/ploidyfrost_hip.h"
#include "pf_ctx.hpp"

#define PF_HIP(call)                                                                         \
    do {                                                                                     \
        hipError_t e_ = (call);                                                              \
        if (e_ != hipSuccess) {                                                              \
            pf::CtxErr{ctx} = std::string(#call) + ": " + hipGetErrorString(e_);                   \
            return PF_ERR_HIP;                                                               \
        }                                                                                    \
    } while (0)

namespace pf {

__device__ inline uint64_t rotl64(uint64_t x, int r) { r &= 63; return r ? (x << r) | (x >> (64 - r)) : x; }
__device__ inline uint64_t wymix(uint64_t a, uint64_t b) { return (a * b) ^ __umul64hi(a, b); }
__device__ inline uint64_t shfl64(uint64_t v, int src) {
    const uint32_t lo = __shfl((uint32_t)v, src, 64), hi = __shfl((uint32_t)(v >> 32), src, 64);
    return ((uint64_t)hi << 32) | lo;
}
__device__ inline uint64_t mix64_minz(uint64_t x) {  // pf_host_minz.cpp mix64
    x ^= x >> 33; x *= 0xff51afd7ed558ccdull;
    x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ull;
    x ^= x >> 33;
    return x;
}

// hash (bifrost/src/RepHash.hpp:28-95) and canonical value of the g-mer whose 2g bits are `fw` (first base highest)
__device__ inline uint64_t gmer_hash_rep(uint64_t fw, int g, uint64_t &rep) {
    const uint64_t hv[4] = {2053695854357871005ULL, 5073395517033431291ULL, 10060236952204337488ULL, 7783083932390163561ULL};
    uint64_t h = 0, ht = 0, rc = 0;
    for (int i = 0; i < g; ++i) {
        const uint32_t c = (uint32_t)(fw >> (2 * (g - 1 - i))) & 3;   // A0 C1 G2 T3
        const uint32_t cm = c ^ (c >> 1);                             // (ascii & 6) >> 1: A0 C1 G3 T2
        const uint32_t tm = cm ^ 2;                                   // ((ascii ^ 4) & 6) >> 1: A2 C3 G1 T0
        h ^= rotl64(hv[cm], g - 1 - i);
        ht ^= rotl64(hv[tm], i);
        rc |= (uint64_t)(3 - c) << (2 * i);
    }
    rep = fw < rc ? fw : rc;
    const uint64_t lo = h < ht ? h : ht, hi = h < ht ? ht : h;
    const uint64_t wyp0 = 0xa0761d6478bd642full, wyp1 = 0xe7037ed1a0b428dbull;
    const uint64_t a = ((lo & 0xFFFFFFFFull) << 32) | (hi & 0xFFFFFFFFull);
    const uint64_t b = ((hi >> 32) << 32) | (lo >> 32);
    return wymix(wyp1 ^ 16, wymix(a ^ wyp1, b ^ wyp0));
}

// FLAG = false: the census (every counted position adds one to its slot).  FLAG = true, behind the census: a unitig is flagged when
// one of its counted positions falls into a slot that reached `limit` -- the unitigs the host replay has to run through
// Bifrost's addUnitig (host/pf_host_minz.cpp: its own pass over all unitigs, 1.3 s at 10 M unitigs, for the same answer).
template <bool FLAG>
__global__ __launch_bounds__(256) void k_minz_count(const uint64_t *__restrict__ seq, const uint64_t *__restrict__ off,
                                                    const uint32_t *__restrict__ len, uint32_t N, int k, int g,
                                                    uint32_t *__restrict__ table, uint64_t mask, uint32_t limit,
                                                    unsigned int *__restrict__ out /* [0] max count, [1] slots that reached limit */,
                                                    uint8_t *__restrict__ flags /* FLAG: one byte per unitig */) {
    const int lane = threadIdx.x & 63;
    const uint32_t wave = (uint32_t)(((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6);
    const uint32_t n_waves = (uint32_t)(((uint64_t)gridDim.x * blockDim.x) >> 6);
    const int W = k - g - 1;            // g-mer positions per k-mer window
    const int TQ = 64 - 2 * (W - 1);    // positions decided per step
    uint32_t my_max = 0, my_crowded = 0;
    for (uint32_t u = wave; u < N; u += n_waves) {
        const uint64_t *w = seq + off[u];
        const int L = (int)len[u];
        const int n_km = L - k + 1, pmax = L - g - 1;   // positions 1 .. pmax carry minimizers
        bool hit = false;
        for (int q0 = 1; q0 <= pmax; q0 += TQ) {
            const int ts = q0 - (W - 1);
            const int q = ts + lane;
            const bool q_ok = q >= 1 && q <= pmax;
            uint64_t h = ~0ull, rep = 0;
            if (q_ok) {
                const uint64_t hi = w[q >> 5], lo = w[(q >> 5) + 1];   // the device copy carries two padding words
                const int sh = 2 * (q & 31);
                const uint64_t v = sh ? (hi << sh) | (lo >> (64 - sh)) : hi;
                h = gmer_hash_rep(v >> (64 - 2 * g), g, rep);
            }
            // lane j: minimum over the window of k-mer p = ts - 1 + j, i.e. over lanes j .. j + W - 1
            const int p = ts - 1 + lane;
            const bool w_ok = p >= 0 && p < n_km && lane + W - 1 < 64;
            uint64_t wm = ~0ull;
            for (int i = 0; i < W; ++i) {
                const uint64_t x = shfl64(h, (lane + i) & 63);
                if (x < wm) wm = x;
            }
            const unsigned long long wvalid = __ballot(w_ok);
            // position lane l (interior only) is counted when some window j in l - W + 1 .. l has its minimum here
            bool counted = false;
            const bool interior = lane >= W - 1 && lane <= 64 - W && q_ok;
            for (int i = 0; i < W; ++i) {
                const int j = lane - i;
                const uint64_t m = shfl64(wm, j & 63);
                if (interior && j >= 0 && ((wvalid >> j) & 1) && m == h) counted = true;
            }
            if (counted) {
                if (FLAG) {
                    if (table[mix64_minz(rep) & mask] >= limit) hit = true;
                } else {
                    const uint32_t c = atomicAdd(&table[mix64_minz(rep) & mask], 1u) + 1;
                    if (c > my_max) my_max = c;
                    if (c == limit) ++my_crowded;
                }
            }
        }
        if (FLAG) {
            const unsigned long long any = __ballot(hit);
            if (lane == 0) flags[u] = any ? 1 : 0;
        }
    }
    if (FLAG) return;
    for (int o = 32; o > 0; o >>= 1) {
        const uint32_t m = __shfl_down(my_max, o, 64), c = __shfl_down(my_crowded, o, 64);
        if (m > my_max) my_max = m;
        my_crowded += c;
    }
    if (lane == 0) {
        if (my_max) atomicMax(&out[0], my_max);
        if (my_crowded) atomicAdd(&out[1], my_crowded);
    }
}

__global__ void k_minz_narrow(const uint32_t *__restrict__ table, uint64_t slots, uint8_t *__restrict__ out) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < slots) out[i] = (uint8_t)(table[i] < 255u ? table[i] : 255u);
}

}  // namespace pf

using namespace pf;

extern "C" {

uint64_t pf_minimizer_table_slots(uint64_t n_kmers) {
    uint64_t cap = 1ull << 16;
    while (cap < n_kmers / 2 && cap < (1ull << 30)) cap <<= 1;
    return cap;
}

int pf_minimizer_crowding(pf_ctx *ctx, int g, uint32_t limit, uint32_t *max_occurrences, uint64_t *crowded_slots, uint32_t *table_out) {
    if (!ctx || !ctx->d_seq || !max_occurrences || g < 1 || g > 31 || g > ctx->k - 2 || limit == 0) {
        if (ctx) pf::CtxErr{ctx} = "pf_minimizer_crowding: no graph uploaded, or g outside 1 .. min(31, k - 2)";
        return PF_ERR_ARG;
    }
    PF_HIP(hipSetDevice(ctx->device));
    const uint64_t slots = pf_minimizer_table_slots(ctx->n_kmers);
    DevTmp<uint32_t> table;
    DevTmp<unsigned int> out;
    PF_HIP(table.alloc(slots * 4));
    PF_HIP(out.alloc(8));
    PF_HIP(hipMemsetAsync(table.p, 0, slots * 4, ctx->stream));
    PF_HIP(hipMemsetAsync(out.p, 0, 8, ctx->stream));
    ctx_begin(ctx, PF_K_MINZ);
    k_minz_count<false><<<ctx_grid(ctx, (uint64_t)ctx->N * 64, 256, 16), 256, 0, ctx->stream>>>(ctx->d_seq, ctx->d_off, ctx->d_len, ctx->N, ctx->k, g, table.p,
                                                                                                slots - 1, limit, out.p, nullptr);
    ctx_end(ctx);
    unsigned int h[2] = {0, 0};
    PF_HIP(hipMemcpyAsync(h, out.p, 8, hipMemcpyDeviceToHost, ctx->stream));
    if (table_out) PF_HIP(hipMemcpyAsync(table_out, table.p, slots * 4, hipMemcpyDefault, ctx->stream));
    PF_HIP(hipStreamSynchronize(ctx->stream));
    *max_occurrences = h[0];
    if (crowded_slots) *crowded_slots = h[1];
    return PF_OK;
}

int pf_minimizer_replay_inputs(pf_ctx *ctx, int g, uint32_t limit, uint8_t *counters8_out, uint8_t *unitig_flags_out) {
    if (!ctx || !ctx->d_seq || !counters8_out || !unitig_flags_out || g < 1 || g > 31 || g > ctx->k - 2 || limit == 0) {
        if (ctx) pf::CtxErr{ctx} = "pf_minimizer_replay_inputs: no graph uploaded, or g outside 1 .. min(31, k - 2)";
        return PF_ERR_ARG;
    }
    PF_HIP(hipSetDevice(ctx->device));
    const uint64_t slots = pf_minimizer_table_slots(ctx->n_kmers);
    DevTmp<uint32_t> table;
    DevTmp<unsigned int> out;
    DevTmp<uint8_t> narrow, flags;
    PF_HIP(table.alloc(slots * 4));
    PF_HIP(out.alloc(8));
    PF_HIP(narrow.alloc(slots));
    PF_HIP(flags.alloc(ctx->N));
    PF_HIP(hipMemsetAsync(table.p, 0, slots * 4, ctx->stream));
    PF_HIP(hipMemsetAsync(out.p, 0, 8, ctx->stream));
    const int grid = ctx_grid(ctx, (uint64_t)ctx->N * 64, 256, 16);
    ctx_begin(ctx, PF_K_MINZ);
    k_minz_count<false><<<grid, 256, 0, ctx->stream>>>(ctx->d_seq, ctx->d_off, ctx->d_len, ctx->N, ctx->k, g, table.p, slots - 1, limit, out.p, nullptr);
    k_minz_count<true><<<grid, 256, 0, ctx->stream>>>(ctx->d_seq, ctx->d_off, ctx->d_len, ctx->N, ctx->k, g, table.p, slots - 1, limit, out.p, flags.p);
    k_minz_narrow<<<(unsigned)((slots + 255) / 256), 256, 0, ctx->stream>>>(table.p, slots, narrow.p);
    ctx_end(ctx);
    PF_HIP(hipGetLastError());
    PF_HIP(hipMemcpyAsync(counters8_out, narrow.p, slots, hipMemcpyDefault, ctx->stream));
    PF_HIP(hipMemcpyAsync(unitig_flags_out, flags.p, ctx->N, hipMemcpyDefault, ctx->stream));
    PF_HIP(hipStreamSynchronize(ctx->stream));
    return PF_OK;
}

}  // extern "C"
