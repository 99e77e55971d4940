// Colored (multi-sample) coverage on gfx950: the device side of reference src/CCDBG.cpp's
// readCovUni (:123-156) and readCov(string, low, up, colour) (:89-122) over *one* count database per colour
// (CCDBG::CCDBG, :13-43).
//
// MI355X layout: the C databases are joined into ONE open-addressing table keyed by the stored k-mer,
// each slot = { u64 key, u32 count[C] } padded to a power-of-two stride (16 B for C <= 2, 32 B for C <= 6,
// 64 B for C <= 14, ...), so that a k-mer costs one random HBM access for all colours instead of C.  An absent
// (k-mer, colour) pair is the all-ones count.  K-COV-C walks the unitigs exactly like K-COV (one wavefront per
// unitig, lanes over its k-mers) and reduces sum / min / max / missing per colour, four colours per pass
// in registers; the range test of readCovUni ("low < count < up" for every k-mer) is min > low && max < up on
// the host, which keeps the cutoffs out of the resident result.  Integer / index work: no MFMA.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "pf_colored_dev.hpp"
#include "pf_cov_stream.hpp"
#include "pf_ctx.hpp"
#include "pf_device_common.hpp"
#include "ploidyfrost_hip.h"

using namespace pf;

#define PF_HIP(call)                                                                         \
    do {                                                                                     \
        hipError_t e_ = (call);                                                              \
        if (e_ != hipSuccess) {                                                              \
            pf::CtxErr{ctx} = std::string(#call) + ": " + hipGetErrorString(e_);                   \
            return PF_ERR_HIP;                                                               \
        }                                                                                    \
    } while (0)

namespace {

constexpr uint32_t MISSING = pf::CTAB_MISSING;

// K-TABLE-C: one thread per record of colour `colour`
__global__ void k_ctab_build(uint8_t *base, uint64_t mask, uint32_t shift, int k, const uint64_t *__restrict__ kmers,
                             const uint32_t *__restrict__ counts, uint64_t n, uint64_t min_count, uint64_t max_count, uint32_t colour,
                             unsigned int *noncanon) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    bool nc = false;
    for (; i < n; i += stride) {
        const uint32_t c = counts[i];
        if (c < min_count || c > max_count) continue;  // not retrievable (kmc_file.cpp:1459)
        const uint64_t key = kmers[i];
        const uint64_t rc = rc_kmer(key, k);
        nc |= rc < key;
        const CTab t{base, mask, shift};
        const CSeq sq = ctab_seq(t, key, rc, k);
        const uint32_t per = 1u << sq.ls;
        bool done = false;
        for (int tr = 0; tr < LINE_TRIES && !done; ++tr) {
            const uint64_t b = ctab_bucket(t, sq, tr);
            for (uint32_t j = 0; j < per && !done; ++j) {
                uint8_t *slot = base + ((b + j) << shift);
                unsigned long long old = *reinterpret_cast<volatile unsigned long long *>(slot);
                if (old == EMPTY_KEY) old = atomicCAS(reinterpret_cast<unsigned long long *>(slot), EMPTY_KEY, key);
                if (old == EMPTY_KEY || old == key) { *reinterpret_cast<uint32_t *>(slot + 8 + 4 * colour) = c; done = true; }
            }
        }
        for (uint64_t s = mix64(key) & mask; !done; s = (s + 1) & mask) {
            uint8_t *slot = base + (s << shift);
            unsigned long long old = atomicCAS(reinterpret_cast<unsigned long long *>(slot), EMPTY_KEY, key);
            if (old == EMPTY_KEY || old == key) { *reinterpret_cast<uint32_t *>(slot + 8 + 4 * colour) = c; done = true; }
        }
    }
    if (__any(nc) && lane_id() == 0) atomicOr(noncanon, 1u);
}

// is some k-mer a key in both orientations (whatever the colours)?  one thread per slot
__global__ void k_ctab_two_strands(const uint8_t *base, uint64_t cap, uint32_t shift, int k, unsigned int *flag) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    CTab t{base, cap - 1, shift};
    for (; i < cap; i += stride) {
        const uint8_t *s = ctab_slot(t, i);
        const uint64_t key = *reinterpret_cast<const uint64_t *>(s);
        if (key == EMPTY_KEY) continue;
        const uint64_t r = rc_kmer(key, k);
        if (r == key) continue;
        if (ctab_find(t, r, ctab_seq(t, key, r, k))) atomicOr(flag, 1u);
    }
}

// K-COV-C: one wavefront per unitig, lanes over its k-mers, CPP colours per pass.
// out_*[c * n + (u - u0)], n = u1 - u0.
__global__ __launch_bounds__(256) void k_cov_colored(CTab t, const uint64_t *__restrict__ seq, const uint64_t *__restrict__ off,
                                                     const uint32_t *__restrict__ len, int k, bool one_strand, uint32_t n_colors,
                                                     uint32_t u0, uint32_t u1, const uint8_t *__restrict__ unread, uint64_t *__restrict__ out_sum,
                                                     uint32_t *__restrict__ out_min, uint32_t *__restrict__ out_max,
                                                     uint8_t *__restrict__ out_miss, const uint64_t *__restrict__ kpre,
                                                     uint32_t *__restrict__ gcov, uint64_t g_stride) {
    const int lane = lane_id();
    const uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const uint32_t n_waves = (gridDim.x * blockDim.x) >> 6;
    const uint32_t n = u1 - u0;
    for (uint32_t u = u0 + wave; u < u1; u += n_waves) {
        const uint64_t *w = seq + off[u];
        const uint32_t nk = len[u] - k + 1;
        for (uint32_t c0 = 0; c0 < n_colors; c0 += CPP) {
            const uint32_t n_here = min((uint32_t)CPP, n_colors - c0);
            uint64_t sum[CPP];
            uint32_t mn[CPP], mx[CPP];
            bool miss[CPP];
#pragma unroll
            for (int j = 0; j < CPP; ++j) { sum[j] = 0; mn[j] = MISSING; mx[j] = 0; miss[j] = false; }
            for (uint32_t p = lane; p < nk; p += WAVE) {
                uint32_t cnt[CPP];
                colored_counts(t, kmer_at(w, p, k), k, one_strand, c0, n_here, cnt);
                if (gcov) {  // K-COV-C-JOIN: every colour's count goes to the k-mer's place in graph order, nothing is reduced
#pragma unroll
                    for (int j = 0; j < CPP; ++j)
                        if ((uint32_t)j < n_here) gcov[(uint64_t)(c0 + j) * g_stride + kpre[u] + p] = cnt[j];
                    continue;
                }
#pragma unroll
                for (int j = 0; j < CPP; ++j) {
                    if (cnt[j] == MISSING) { miss[j] = true; continue; }
                    sum[j] += cnt[j];
                    mn[j] = cnt[j] < mn[j] ? cnt[j] : mn[j];
                    mx[j] = cnt[j] > mx[j] ? cnt[j] : mx[j];
                }
            }
            if (gcov) continue;
#pragma unroll
            for (int j = 0; j < CPP; ++j) {
                if ((uint32_t)j >= n_here) break;
                const uint64_t s = wave_sum_u64(sum[j]);
                const uint32_t lo = wave_min_u32(mn[j]);
                const uint32_t hi = ~wave_min_u32(~mx[j]);
                const bool any_miss = __ballot(miss[j]) != 0;
                if (lane == 0) {
                    const size_t o = (size_t)(c0 + j) * n + (u - u0);
                    // a colour whose database was written without canonical counting is never looked up: readCovUni returns
                    // (0, true) for it (src/CCDBG.cpp:128, 155) -- sum 0, nothing missing, every count "inside" any cutoffs
                    const bool skip = unread[c0 + j] != 0;
                    out_sum[o] = skip ? 0 : s;
                    out_min[o] = skip ? MISSING : lo;
                    out_max[o] = skip ? 0 : hi;
                    out_miss[o] = skip ? 0 : any_miss;
                }
            }
        }
    }
}

// results of the streaming K-COV-C before the kernel: nothing seen (also the final answer for colours that are never looked up)
__global__ void k_ccov_init(uint64_t n, uint64_t *__restrict__ out_sum, uint32_t *__restrict__ out_min, uint32_t *__restrict__ out_max,
                            uint8_t *__restrict__ out_miss) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) { out_sum[i] = 0; out_min[i] = MISSING; out_max[i] = 0; out_miss[i] = 0; }
}

// K-STRCOV-C: one thread per (string, colour); out_*[i * n_colors + c]
__global__ void k_strcov_colored(CTab t, int k, bool one_strand, uint32_t n_colors, const uint8_t *__restrict__ unread, const char *__restrict__ text,
                                 const uint64_t *__restrict__ str_off, uint32_t n_str, const uint32_t *__restrict__ low,
                                 const uint32_t *__restrict__ up, uint64_t *__restrict__ out_sum, uint8_t *__restrict__ out_ok) {
    uint64_t id = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    const uint64_t kmask = (1ull << (2 * k)) - 1;
    const uint64_t total = (uint64_t)n_str * n_colors;
    for (; id < total; id += stride) {
        const uint32_t i = (uint32_t)(id / n_colors), c = (uint32_t)(id % n_colors);
        const char *s = text + str_off[i];
        const uint32_t L = (uint32_t)(str_off[i + 1] - str_off[i]);
        const uint32_t lo = low[c], hi = up[c];
        uint64_t sum = 0;
        uint8_t ok = 1;
        StringWindow win;
        if (unread[c]) {  // readCov(s, low, up, c) without canonical counting: (0, true), no lookup (src/CCDBG.cpp:94, 121)
            out_sum[id] = 0;
            out_ok[id] = 1;
            continue;
        }
        for (uint32_t j = 0; j < L; ++j) {
            const uint64_t x = win.push(s[j], kmask, (uint32_t)k);
            if (j + 1 >= (uint32_t)k) {
                uint32_t cnt[CPP];
                colored_counts(t, x, k, one_strand, c, 1, cnt);
                // a missing k-mer and a count outside (low, up) both give (0, false): CCDBG.cpp:105-117
                if (cnt[0] != MISSING && cnt[0] > lo && cnt[0] < hi) sum += cnt[0];
                else { sum = 0; ok = 0; break; }
            }
        }
        out_sum[id] = sum;
        out_ok[id] = ok;
    }
}

template <typename T>
bool on_device(const T *p) {
    hipPointerAttribute_t at;
    const bool dev = p && hipPointerGetAttributes(&at, p) == hipSuccess && at.type == hipMemoryTypeDevice;
    (void)hipGetLastError();
    return dev;
}

}  // namespace

// K-COV-C-JOIN: the colored counterpart of pf::join_graph_counts (pf_device.hip): one probe of the joined table per graph
// k-mer leaves every colour's count (all ones = the colour's database lacks the k-mer) at the k-mer's position in graph
// order, colour-major.  Once per (graph, set of databases); pf_unitig_cov_colored then streams it.  A count equal to the
// marker cannot be represented: such a set of databases keeps the probing K-COV-C.
namespace pf {
int join_graph_counts_colored(pf_ctx *ctx) {
    ctx->gcov_c_valid = false;
    if (!ctx->d_seq || !ctx->d_ctab || !ctx->n_colors || !ctx->d_kpre || ctx->n_kmers == 0 || ctx->ctab_max_count >= MISSING) return PF_OK;
    PF_HIP(hipSetDevice(ctx->device));
    const uint64_t stride = (ctx->n_krow + 4) * 64;  // whole super-rows of 256 k-mers, 16-byte aligned slices
    if (!ctx->d_gcov_c || ctx->gcov_c_stride != stride) {
        if (ctx->d_gcov_c) { (void)hipFree(ctx->d_gcov_c); ctx->d_gcov_c = nullptr; }
        if (hipMalloc(reinterpret_cast<void **>(&ctx->d_gcov_c), stride * ctx->n_colors * sizeof(uint32_t)) != hipSuccess) {
            (void)hipGetLastError();  // no room for the SoA: pf_unitig_cov_colored keeps probing the table every pass (same results)
            ctx->d_gcov_c = nullptr;
            return PF_OK;
        }
        ctx->gcov_c_stride = stride;
    }
    const CTab t{ctx->d_ctab, ctx->ctab_cap - 1, ctx->ctab_shift};
    ctx_begin(ctx, PF_K_COV_JOIN);
    k_cov_colored<<<ctx_grid(ctx, (uint64_t)ctx->N * 64, 256, 16), 256, 0, ctx->stream>>>(t, ctx->d_seq, ctx->d_off, ctx->d_len, ctx->k, ctx->ctab_one_strand,
                                                                                        ctx->n_colors, 0, ctx->N, ctx->d_unread, nullptr, nullptr,
                                                                                        nullptr, nullptr, ctx->d_kpre, ctx->d_gcov_c, stride);
    ctx_end(ctx);
    PF_HIP(hipGetLastError());
    PF_HIP(hipStreamSynchronize(ctx->stream));
    ctx->gcov_c_valid = true;
    return PF_OK;
}
}  // namespace pf

extern "C" {

uint32_t pf_num_colors(const pf_ctx *ctx) { return ctx ? ctx->n_colors : 0; }

int pf_upload_counts_colored(pf_ctx *ctx, uint32_t n_colors, const uint64_t *const *kmers, const uint32_t *const *counts,
                             const uint64_t *n, const uint64_t *min_count, const uint64_t *max_count, const int *both_strands) {
    if (!ctx || n_colors == 0 || !kmers || !counts || !n || !min_count || !max_count || !both_strands) return PF_ERR_ARG;
    if (n_colors > PF_MAX_COLORS_TABLE) { pf::CtxErr{ctx} = "more colours than the device table holds (PF_MAX_COLORS_TABLE)"; return PF_ERR_ARG; }
    if (!ctx->d_seq || !ctx->k) { pf::CtxErr{ctx} = "pf_upload_counts_colored: the graph comes first (the table is addressed by minimizers of length-k keys)"; return PF_ERR_ARG; }
    uint64_t total = 0, biggest = 0;
    for (uint32_t c = 0; c < n_colors; ++c) {
        if (n[c] && (!kmers[c] || !counts[c])) return PF_ERR_ARG;
        if (!both_strands[c]) continue;   // never looked up (src/CCDBG.cpp:94, 128): its records stay out of the table
        total += n[c];
        biggest = std::max(biggest, n[c]);
    }
    ctx->ctab_unread = 0;
    ctx->ctab_max_count = 0;
    std::vector<uint8_t> unread(n_colors, 0);
    for (uint32_t c = 0; c < n_colors; ++c) {
        if (!both_strands[c]) { unread[c] = 1; if (c < 64) ctx->ctab_unread |= 1ull << c; }
        else ctx->ctab_max_count = std::max<uint64_t>(ctx->ctab_max_count, max_count[c]);
    }
    if (ctx->d_gcov_c) { (void)hipFree(ctx->d_gcov_c); ctx->d_gcov_c = nullptr; }
    ctx->gcov_c_valid = false;
    PF_HIP(hipSetDevice(ctx->device));
    if (ctx->d_ctab) { (void)hipFree(ctx->d_ctab); ctx->d_ctab = nullptr; }
    if (ctx->d_unread) { (void)hipFree(ctx->d_unread); ctx->d_unread = nullptr; }
    PF_HIP(hipMalloc(reinterpret_cast<void **>(&ctx->d_unread), n_colors));
    PF_HIP(hipMemcpy(ctx->d_unread, unread.data(), n_colors, hipMemcpyHostToDevice));
    ctx->n_colors = 0;
    // distinct keys <= total: a capacity above 1.25 * total always leaves empty slots; colours that share most
    // k-mers (samples of one species) end at a load factor near max(n) / cap <= 0.5
    uint64_t cap = 1024;
    while (cap < 2 * biggest || cap < total + total / 4 + 1) cap <<= 1;
    uint32_t shift = 4;
    while ((1u << shift) < 8 + 4 * n_colors) ++shift;
    PF_HIP(hipMalloc(reinterpret_cast<void **>(&ctx->d_ctab), cap << shift));
    PF_HIP(hipMemsetAsync(ctx->d_ctab, 0xFF, cap << shift, ctx->stream));  // EMPTY_KEY keys, MISSING counts
    ctx->ctab_cap = cap;
    ctx->ctab_shift = shift;
    DevTmp<unsigned int> noncanon_;
    PF_HIP(noncanon_.alloc(4));
    PF_HIP(hipMemsetAsync(noncanon_.p, 0, 4, ctx->stream));
    for (uint32_t c = 0; c < n_colors; ++c) {
        if (!n[c] || !both_strands[c]) continue;
        DevTmp<uint64_t> dk_;
        DevTmp<uint32_t> dc_;
        const uint64_t *pk = kmers[c];
        const uint32_t *pc = counts[c];
        if (!on_device(kmers[c])) {
            PF_HIP(dk_.alloc(n[c] * 8));
            PF_HIP(dc_.alloc(n[c] * 4));
            PF_HIP(hipMemcpyAsync(dk_.p, kmers[c], n[c] * 8, hipMemcpyDefault, ctx->stream));
            PF_HIP(hipMemcpyAsync(dc_.p, counts[c], n[c] * 4, hipMemcpyDefault, ctx->stream));
            pk = dk_.p;
            pc = dc_.p;
        }
        ctx_begin(ctx, PF_K_TABLE_BUILD);
        k_ctab_build<<<ctx_grid(ctx, n[c], 256, 8), 256, 0, ctx->stream>>>(ctx->d_ctab, cap - 1, shift, ctx->k, pk, pc, n[c], min_count[c],
                                                                           max_count[c], c, noncanon_.p);
        ctx_end(ctx);
        PF_HIP(hipStreamSynchronize(ctx->stream));  // the staging buffers die at the end of this iteration
    }
    {
        // a table whose keys are all canonical holds no k-mer in both orientations: only one with other keys is searched for a pair
        unsigned int h_noncanon = 0, h_flag = 0;
        PF_HIP(hipMemcpyAsync(&h_noncanon, noncanon_.p, 4, hipMemcpyDeviceToHost, ctx->stream));
        PF_HIP(hipStreamSynchronize(ctx->stream));
        if (total && h_noncanon) {
            DevTmp<unsigned int> flag_;
            PF_HIP(flag_.alloc(4));
            PF_HIP(hipMemsetAsync(flag_.p, 0, 4, ctx->stream));
            k_ctab_two_strands<<<ctx_grid(ctx, cap, 256, 8), 256, 0, ctx->stream>>>(ctx->d_ctab, cap, shift, ctx->k, flag_.p);
            PF_HIP(hipMemcpyAsync(&h_flag, flag_.p, 4, hipMemcpyDeviceToHost, ctx->stream));
            PF_HIP(hipStreamSynchronize(ctx->stream));
        }
        ctx->ctab_one_strand = total ? h_flag == 0 : false;
    }
    ctx->n_colors = n_colors;
    return join_graph_counts_colored(ctx);
}

static int unitig_cov_colored_impl(pf_ctx *ctx, uint32_t u0, uint32_t u1, uint64_t *sum, uint32_t *mn, uint32_t *mx, uint8_t *miss, bool probe) {
    if (!ctx || !ctx->d_seq || !ctx->d_ctab || !ctx->n_colors || u0 > u1 || u1 > ctx->N || !sum || !mn || !mx || !miss) return PF_ERR_ARG;
    if (u0 == u1) return PF_OK;
    PF_HIP(hipSetDevice(ctx->device));
    const size_t n = (size_t)(u1 - u0) * ctx->n_colors;
    const bool dev_out = on_device(sum);
    uint64_t *ds = sum;
    uint32_t *dlo = mn, *dhi = mx;
    uint8_t *dx = miss;
    if (!dev_out) {
        ds = (uint64_t *)ctx_ws(ctx, WS_CCOV_SUM, n * 8);
        dlo = (uint32_t *)ctx_ws(ctx, WS_CCOV_MIN, n * 4);
        dhi = (uint32_t *)ctx_ws(ctx, WS_CCOV_MAX, n * 4);
        dx = (uint8_t *)ctx_ws(ctx, WS_CCOV_MISS, n);
        if (!ds || !dlo || !dhi || !dx) return PF_ERR_HIP;
    }
    constexpr bool env_probe = false;
    probe = probe || env_probe;
    if (!probe && !ctx->gcov_c_valid) {  // the graph was replaced under the table
        const int rc = join_graph_counts_colored(ctx);
        if (rc) return rc;
    }
    const bool stream = !probe && ctx->gcov_c_valid;
    uint64_t g_range[2] = {0, 0};
    if (stream) {
        PF_HIP(hipMemcpyAsync(&g_range[0], ctx->d_kpre + u0, 8, hipMemcpyDeviceToHost, ctx->stream));
        PF_HIP(hipMemcpyAsync(&g_range[1], ctx->d_kpre + u1, 8, hipMemcpyDeviceToHost, ctx->stream));
        PF_HIP(hipStreamSynchronize(ctx->stream));
    }
    ctx_begin(ctx, PF_K_COV_COLORED);
    if (stream) {
        // streaming form (pf_cov_stream.hpp): one grid row per colour over that colour's slice of the coverage SoA
        k_ccov_init<<<ctx_grid(ctx, n, 256, 8), 256, 0, ctx->stream>>>(n, ds, dlo, dhi, dx);
        Kc4Args a{ctx->d_gcov_c, ctx->gcov_c_stride, ctx->d_unread, ctx->d_khead, ctx->d_krow, u0, u1 - u0, g_range[0], g_range[1],
                  g_range[0] / 256, (g_range[1] + 255) / 256, ds, dlo, dhi, dx};
        const int rc = launch_cov_stream(ctx, a, ctx->n_colors, ctx->ctab_max_count >= (1ull << 20), true);
        if (rc) return rc;
    } else {
        const CTab t{ctx->d_ctab, ctx->ctab_cap - 1, ctx->ctab_shift};
        const int grid = ctx_grid(ctx, (uint64_t)(u1 - u0) * 64, 256, 16);
        k_cov_colored<<<grid, 256, 0, ctx->stream>>>(t, ctx->d_seq, ctx->d_off, ctx->d_len, ctx->k, ctx->ctab_one_strand, ctx->n_colors, u0, u1,
                                                     ctx->d_unread, ds, dlo, dhi, dx, nullptr, nullptr, 0);
    }
    ctx_end(ctx);
    if (!dev_out) {
        PF_HIP(hipMemcpyAsync(sum, ds, n * 8, hipMemcpyDeviceToHost, ctx->stream));
        PF_HIP(hipMemcpyAsync(mn, dlo, n * 4, hipMemcpyDeviceToHost, ctx->stream));
        PF_HIP(hipMemcpyAsync(mx, dhi, n * 4, hipMemcpyDeviceToHost, ctx->stream));
        PF_HIP(hipMemcpyAsync(miss, dx, n, hipMemcpyDeviceToHost, ctx->stream));
        PF_HIP(hipStreamSynchronize(ctx->stream));
    }
    return PF_OK;
}

int pf_unitig_cov_colored(pf_ctx *ctx, uint32_t u0, uint32_t u1, uint64_t *sum, uint32_t *mn, uint32_t *mx, uint8_t *miss) {
    return unitig_cov_colored_impl(ctx, u0, u1, sum, mn, mx, miss, false);
}

int pf_unitig_cov_colored_probe(pf_ctx *ctx, uint32_t u0, uint32_t u1, uint64_t *sum, uint32_t *mn, uint32_t *mx, uint8_t *miss) {
    return unitig_cov_colored_impl(ctx, u0, u1, sum, mn, mx, miss, true);
}

int pf_string_cov_colored(pf_ctx *ctx, const char *text, const uint64_t *str_off, uint32_t n_str, const uint32_t *low,
                          const uint32_t *up, uint64_t *sum, uint8_t *ok) {
    if (!ctx || !ctx->d_ctab || !ctx->n_colors || !low || !up || (n_str && (!text || !str_off || !sum || !ok))) return PF_ERR_ARG;
    if (n_str == 0) return PF_OK;
    PF_HIP(hipSetDevice(ctx->device));
    const uint32_t C = ctx->n_colors;
    uint64_t total = 0;
    PF_HIP(hipMemcpy(&total, str_off + n_str, 8, hipMemcpyDefault));
    char *dt = (char *)ctx_ws(ctx, WS_STR_TEXT, (size_t)total + 1);
    uint64_t *doff = (uint64_t *)ctx_ws(ctx, WS_STR_OFF, ((size_t)n_str + 1) * 8);
    uint64_t *ds = (uint64_t *)ctx_ws(ctx, WS_STR_SUM, (size_t)n_str * C * 8);
    uint8_t *dk = (uint8_t *)ctx_ws(ctx, WS_STR_OK, (size_t)n_str * C);
    uint32_t *dcut = (uint32_t *)ctx_ws(ctx, WS_STR_MISS, (size_t)C * 8);
    if (!dt || !doff || !ds || !dk || !dcut) return PF_ERR_HIP;
    PF_HIP(hipMemcpyAsync(dt, text, (size_t)total, hipMemcpyDefault, ctx->stream));
    PF_HIP(hipMemcpyAsync(doff, str_off, ((size_t)n_str + 1) * 8, hipMemcpyDefault, ctx->stream));
    PF_HIP(hipMemcpyAsync(dcut, low, (size_t)C * 4, hipMemcpyDefault, ctx->stream));
    PF_HIP(hipMemcpyAsync(dcut + C, up, (size_t)C * 4, hipMemcpyDefault, ctx->stream));
    const CTab t{ctx->d_ctab, ctx->ctab_cap - 1, ctx->ctab_shift};
    ctx_begin(ctx, PF_K_STRCOV_COLORED);
    k_strcov_colored<<<ctx_grid(ctx, (uint64_t)n_str * C, 256, 8), 256, 0, ctx->stream>>>(t, ctx->k, ctx->ctab_one_strand, C, ctx->d_unread, dt, doff, n_str,
                                                                                          dcut, dcut + C, ds, dk);
    ctx_end(ctx);
    PF_HIP(hipMemcpyAsync(sum, ds, (size_t)n_str * C * 8, hipMemcpyDefault, ctx->stream));
    PF_HIP(hipMemcpyAsync(ok, dk, (size_t)n_str * C, hipMemcpyDefault, ctx->stream));
    PF_HIP(hipStreamSynchronize(ctx->stream));
    return PF_OK;
}

}  // extern "C"
