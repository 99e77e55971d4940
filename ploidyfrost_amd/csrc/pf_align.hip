// K-ALN: SeqAlign::needlemanWunch + SeqAlign::traceback (reference src/SeqAlign.cpp:480-549,
// 306-478) on gfx950 -- one wavefront per pairwise job.
//
//   fill      anti-diagonal sweep, lanes over the cells of a diagonal.  Only the 3 direction
//             flags per cell are kept (1 byte: low nibble = the by-value `matrix`, high nibble =
//             `matrix_temp`); scores live in three rolling diagonals.  Everything sits in LDS
//             (dynamic LDS sized per size class); jobs too large for LDS run the same code over
//             per-wave global scratch.  Score arithmetic is the reference's `int = long + double`
//             (truncation), +1 when the move continues the predecessor's own direction, with the
//             `A[i] == '-'` look-ahead rule; every direction that ties the maximum is flagged.
//   traceback data-dependent DFS over all co-optimal paths (Left, then Up, then LeftUp) on lane 0,
//             with the gap-open budgets that start at 5 and shrink to the best alignment found,
//             permanent flag clearing on refused moves, and the reference's asymmetric gap-open
//             bookkeeping for row B.  Each completed path is scored as variantAnalyze does
//             (src/SeqAlign.cpp:237-305) and kept / replaces / is dropped per AlignUnit::operator-
//             (src/SeqAlign.hpp:43-67).
// No MFMA: integer compare/select on byte flags.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstring>
#include <string>
#include <vector>

#include "pf_align_dev.hpp"
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

// publish the staged alignments of one job into the caller's pools (whole wave)
__device__ void publish(const AlnOut &o, uint32_t job, const AlnScratch &sc, uint32_t n_hits, uint32_t text_used,
                        uint32_t gaps_used) {
    const int lane = lane_id();
    unsigned long long h0 = 0, t0 = 0, g0 = 0;
    if (lane == 0) {
        h0 = atomicAdd(&o.heads[0], (unsigned long long)n_hits);
        t0 = atomicAdd(&o.heads[1], (unsigned long long)text_used);
        g0 = atomicAdd(&o.heads[2], (unsigned long long)gaps_used);
        o.hit_first[job] = h0;
        o.hit_count[job] = n_hits;
    }
    h0 = ((unsigned long long)__shfl((uint32_t)(h0 >> 32), 0, WAVE) << 32) | __shfl((uint32_t)h0, 0, WAVE);
    t0 = ((unsigned long long)__shfl((uint32_t)(t0 >> 32), 0, WAVE) << 32) | __shfl((uint32_t)t0, 0, WAVE);
    g0 = ((unsigned long long)__shfl((uint32_t)(g0 >> 32), 0, WAVE) << 32) | __shfl((uint32_t)g0, 0, WAVE);
    if (h0 + n_hits > o.hit_cap || t0 + text_used > o.text_cap || g0 + gaps_used > o.gap_cap) return;  // host sees heads
    for (uint32_t i = lane; i < n_hits; i += WAVE) {
        pf_align_hit h = sc.hits[i];
        h.text_off += t0;
        h.gap_off += g0;
        o.hits[h0 + i] = h;
    }
    for (uint32_t i = lane; i < text_used; i += WAVE) o.text[t0 + i] = sc.text[i];
    for (uint32_t i = lane; i < gaps_used; i += WAVE) o.gaps[g0 + i] = sc.gaps[i];
}

struct AlnParams {
    const char *text;
    const pf_align_job *jobs;
    const uint32_t *idx;  // job indices of this launch
    uint32_t n;
    double M, D, G;
    int integral;
    // per-wave staging
    char *st_text;
    uint32_t *st_gaps;
    pf_align_hit *st_hits;
    uint32_t st_text_cap, st_gap_cap, st_hit_cap;
    // global working storage (global tier only)
    uint8_t *work;
    uint64_t work_per_wave;
    int final_tier;  // overflow here is an error, not a retry
};

template <bool LDS>
__global__ __launch_bounds__(64) void k_align(AlnParams p, AlnOut o) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    uint8_t *base;
    if constexpr (LDS) base = smem;
    else base = p.work + (uint64_t)blockIdx.x * p.work_per_wave;
    AlnScratch sc;
    sc.text = p.st_text + (uint64_t)blockIdx.x * p.st_text_cap;
    sc.gaps = p.st_gaps + (uint64_t)blockIdx.x * p.st_gap_cap;
    sc.hits = p.st_hits + (uint64_t)blockIdx.x * p.st_hit_cap;
    sc.text_cap = p.st_text_cap;
    sc.gap_cap = p.st_gap_cap;
    sc.hit_cap = p.st_hit_cap;
    for (uint32_t q = blockIdx.x; q < p.n; q += gridDim.x) {
        const uint32_t job = p.idx[q];
        const pf_align_job jb = p.jobs[job];
        uint32_t nh, tu, gu;
        const bool ok = align_job(base, p.text + jb.a_off, p.text + jb.b_off, jb.a_len, jb.b_len, p.M, p.D, p.G, p.integral, sc, nh, tu, gu);
        if (ok) {
            publish(o, job, sc, nh, tu, gu);
        } else if (lane_id() == 0) {
            if (p.final_tier) {
                o.hit_first[job] = 0;
                o.hit_count[job] = 0xFFFFFFFFu;  // error marker
            } else {
                const unsigned int r = atomicAdd(o.n_retry, 1u);
                o.retry[r] = job;
            }
        }
        aln_sync();
    }
}

}  // namespace

extern "C" int pf_align_batch(pf_ctx *ctx, const char *text, uint64_t text_len, const pf_align_job *jobs, uint32_t n_jobs,
                              double match, double mismatch, double gap, uint64_t *hit_first, uint32_t *hit_count,
                              pf_align_hit *hits, uint64_t hit_cap, char *out_text, uint64_t text_cap, uint32_t *out_gaps,
                              uint64_t gap_cap, uint64_t used[3]) {
    if (!ctx || !used || (n_jobs && (!text || !jobs || !hit_first || !hit_count || !hits || !out_text || !out_gaps)))
        return PF_ERR_ARG;
    used[0] = used[1] = used[2] = 0;
    if (n_jobs == 0) return PF_OK;
    PF_HIP(hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    // host copy of the job table (size classes, argument validation)
    std::vector<pf_align_job> hj(n_jobs);
    PF_HIP(hipMemcpy(hj.data(), jobs, (size_t)n_jobs * sizeof(pf_align_job), hipMemcpyDefault));
    constexpr uint32_t LIM = 60000;
    const uint64_t cls_bytes[3] = {5 * 1024, 20 * 1024, 64 * 1024};
    std::vector<uint32_t> cls[4];
    uint64_t max_need = 0;
    for (uint32_t j = 0; j < n_jobs; ++j) {
        const pf_align_job &jb = hj[j];
        if (jb.a_len == 0 || jb.b_len == 0 || jb.a_len > LIM || jb.b_len > LIM || jb.a_off + jb.a_len > text_len ||
            jb.b_off + jb.b_len > text_len || (uint64_t)(jb.a_len + 1) * (jb.b_len + 1) > (1ull << 31)) {
            pf::CtxErr{ctx} = "pf_align_batch: job outside the text buffer or longer than 60000 bases";
            return PF_ERR_ARG;
        }
        const uint64_t need = job_bytes(jb.a_len, jb.b_len);
        int c = 3;
        for (int t = 0; t < 3; ++t)
            if (need <= cls_bytes[t]) { c = t; break; }
        cls[c].push_back(j);
        if (c == 3) max_need = std::max(max_need, need);
    }
    // device copies of inputs
    auto is_dev = [](const void *p) {
        hipPointerAttribute_t at;
        bool d = hipPointerGetAttributes(&at, p) == hipSuccess && at.type == hipMemoryTypeDevice;
        (void)hipGetLastError();
        return d;
    };
    char *d_text = const_cast<char *>(text);
    pf_align_job *d_jobs = const_cast<pf_align_job *>(jobs);
    const bool own_text = !is_dev(text), own_jobs = !is_dev(jobs);
    if (own_text) {
        d_text = (char *)ctx_ws(ctx, WS_ALN_TEXT, text_len + 1);
        if (!d_text) return PF_ERR_HIP;
        PF_HIP(hipMemcpyAsync(d_text, text, text_len, hipMemcpyHostToDevice, st));
    }
    if (own_jobs) {
        d_jobs = (pf_align_job *)ctx_ws(ctx, WS_ALN_JOBS, (size_t)n_jobs * sizeof(pf_align_job));
        if (!d_jobs) return PF_ERR_HIP;
        PF_HIP(hipMemcpyAsync(d_jobs, hj.data(), (size_t)n_jobs * sizeof(pf_align_job), hipMemcpyHostToDevice, st));
    }
    const bool dev_out = is_dev(hits);
    AlnOut o;
    uint8_t *small = (uint8_t *)ctx_ws(ctx, WS_ALN_SMALL, 64);
    uint32_t *d_retry = (uint32_t *)ctx_ws(ctx, WS_ALN_RETRY, (size_t)n_jobs * 4);
    uint32_t *d_idx = (uint32_t *)ctx_ws(ctx, WS_ALN_IDX, (size_t)n_jobs * 4);
    if (!small || !d_retry || !d_idx) return PF_ERR_HIP;
    unsigned long long *d_heads = reinterpret_cast<unsigned long long *>(small);
    unsigned int *d_nretry = reinterpret_cast<unsigned int *>(small + 32);
    PF_HIP(hipMemsetAsync(small, 0, 64, st));
    if (dev_out) {
        o.hit_first = hit_first; o.hit_count = hit_count; o.hits = hits; o.text = out_text; o.gaps = out_gaps;
    } else {
        o.hit_first = (uint64_t *)ctx_ws(ctx, WS_ALN_OFIRST, (size_t)n_jobs * 8);
        o.hit_count = (uint32_t *)ctx_ws(ctx, WS_ALN_OCOUNT, (size_t)n_jobs * 4);
        o.hits = (pf_align_hit *)ctx_ws(ctx, WS_ALN_OHITS, std::max<uint64_t>(hit_cap, 1) * sizeof(pf_align_hit));
        o.text = (char *)ctx_ws(ctx, WS_ALN_OTEXT, std::max<uint64_t>(text_cap, 1));
        o.gaps = (uint32_t *)ctx_ws(ctx, WS_ALN_OGAPS, std::max<uint64_t>(gap_cap, 1) * 4);
        if (!o.hit_first || !o.hit_count || !o.hits || !o.text || !o.gaps) return PF_ERR_HIP;
    }
    o.hit_cap = hit_cap; o.text_cap = text_cap; o.gap_cap = gap_cap;
    o.heads = d_heads; o.retry = d_retry; o.n_retry = d_nretry;

    // per-wave staging for the LDS tiers
    const uint32_t ST_TEXT = 64 * 1024, ST_GAPS = 8 * 1024, ST_HITS = 512;
    const int max_waves = ctx->n_cu * 8;
    char *st_text = (char *)ctx_ws(ctx, WS_ALN_STTEXT, (size_t)max_waves * ST_TEXT);
    uint32_t *st_gaps = (uint32_t *)ctx_ws(ctx, WS_ALN_STGAPS, (size_t)max_waves * ST_GAPS * 4);
    pf_align_hit *st_hits = (pf_align_hit *)ctx_ws(ctx, WS_ALN_STHITS, (size_t)max_waves * ST_HITS * sizeof(pf_align_hit));
    if (!st_text || !st_gaps || !st_hits) return PF_ERR_HIP;

    AlnParams p;
    p.text = d_text; p.jobs = d_jobs; p.idx = d_idx; p.M = match; p.D = mismatch; p.G = gap;
    p.integral = (match == std::floor(match) && mismatch == std::floor(mismatch) && gap == std::floor(gap) &&
                  std::fabs(match) < 1e6 && std::fabs(mismatch) < 1e6 && std::fabs(gap) < 1e6) ? 1 : 0;
    p.st_text = st_text; p.st_gaps = st_gaps; p.st_hits = st_hits;
    p.st_text_cap = ST_TEXT; p.st_gap_cap = ST_GAPS; p.st_hit_cap = ST_HITS;
    p.work = nullptr; p.work_per_wave = 0; p.final_tier = 0;

    static bool attr_set = false;
    if (!attr_set) {
        PF_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_align<true>), hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024));
        attr_set = true;
    }
    uint32_t idx_off = 0;
    for (int c = 0; c < 3; ++c) {
        if (cls[c].empty()) continue;
        const uint32_t nc = (uint32_t)cls[c].size();
        PF_HIP(hipMemcpyAsync(d_idx + idx_off, cls[c].data(), (size_t)nc * 4, hipMemcpyHostToDevice, st));
        p.idx = d_idx + idx_off;
        p.n = nc;
        const int grid = (int)std::min<uint32_t>(nc, (uint32_t)max_waves);
        ctx_begin(ctx, PF_K_ALIGN);
        k_align<true><<<grid, 64, cls_bytes[c], st>>>(p, o);
        ctx_end(ctx);
        idx_off += nc;
    }
    // global tier: jobs too large for LDS
    uint8_t *work = nullptr;
    if (!cls[3].empty()) {
        const uint32_t nc = (uint32_t)cls[3].size();
        const int grid = (int)std::min<uint32_t>(nc, 256);
        const uint64_t per = (max_need + 255) & ~255ull;
        work = (uint8_t *)ctx_ws(ctx, WS_ALN_WORK, per * grid);
        if (!work) return PF_ERR_HIP;
        PF_HIP(hipMemcpyAsync(d_idx + idx_off, cls[3].data(), (size_t)nc * 4, hipMemcpyHostToDevice, st));
        p.idx = d_idx + idx_off;
        p.n = nc;
        p.work = work;
        p.work_per_wave = per;
        ctx_begin(ctx, PF_K_ALIGN_BIG);
        k_align<false><<<grid, 64, 0, st>>>(p, o);
        ctx_end(ctx);
    }
    unsigned int n_retry = 0;
    PF_HIP(hipMemcpyAsync(&n_retry, d_nretry, 4, hipMemcpyDeviceToHost, st));
    PF_HIP(hipStreamSynchronize(st));
    int status = PF_OK;
    DevTmp<uint8_t> work2_;
    DevTmp<char> bt_;
    DevTmp<uint32_t> bg_;
    DevTmp<pf_align_hit> bh_;
    if (n_retry) {
        // staging overflow: rerun those jobs in the global tier with 256x the staging
        std::vector<uint32_t> rj(n_retry);
        PF_HIP(hipMemcpy(rj.data(), d_retry, (size_t)n_retry * 4, hipMemcpyDeviceToHost));
        uint64_t need = 0;
        for (uint32_t j : rj) need = std::max(need, job_bytes(hj[j].a_len, hj[j].b_len));
        const int grid = (int)std::min<uint32_t>(n_retry, 16);
        const uint64_t per = (need + 255) & ~255ull;
        const uint32_t BT = 16u << 20, BG = 2u << 20, BH = 128u << 10;
        PF_HIP(work2_.alloc(per * grid));
        PF_HIP(bt_.alloc((size_t)grid * BT));
        PF_HIP(bg_.alloc((size_t)grid * BG * 4));
        PF_HIP(bh_.alloc((size_t)grid * BH * sizeof(pf_align_hit)));
        uint8_t *work2 = work2_.p;
        char *bt = bt_.p;
        uint32_t *bg = bg_.p;
        pf_align_hit *bh = bh_.p;
        PF_HIP(hipMemcpyAsync(d_idx, rj.data(), (size_t)n_retry * 4, hipMemcpyHostToDevice, st));
        p.idx = d_idx; p.n = n_retry; p.work = work2; p.work_per_wave = per; p.final_tier = 1;
        p.st_text = bt; p.st_gaps = bg; p.st_hits = bh;
        p.st_text_cap = BT; p.st_gap_cap = BG; p.st_hit_cap = BH;
        ctx_begin(ctx, PF_K_ALIGN_BIG);
        k_align<false><<<grid, 64, 0, st>>>(p, o);
        ctx_end(ctx);
        PF_HIP(hipStreamSynchronize(st));
    }
    unsigned long long heads[3];
    PF_HIP(hipMemcpy(heads, d_heads, 24, hipMemcpyDeviceToHost));
    used[0] = heads[0]; used[1] = heads[1]; used[2] = heads[2];
    if (heads[0] > hit_cap || heads[1] > text_cap || heads[2] > gap_cap) {
        pf::CtxErr{ctx} = "pf_align_batch: output buffers too small";
        status = PF_ERR_OVERFLOW;
    }
    if (!dev_out) {
        if (status == PF_OK) {
            PF_HIP(hipMemcpy(hit_first, o.hit_first, (size_t)n_jobs * 8, hipMemcpyDeviceToHost));
            PF_HIP(hipMemcpy(hit_count, o.hit_count, (size_t)n_jobs * 4, hipMemcpyDeviceToHost));
            PF_HIP(hipMemcpy(hits, o.hits, (size_t)heads[0] * sizeof(pf_align_hit), hipMemcpyDeviceToHost));
            PF_HIP(hipMemcpy(out_text, o.text, (size_t)heads[1], hipMemcpyDeviceToHost));
            PF_HIP(hipMemcpy(out_gaps, o.gaps, (size_t)heads[2] * 4, hipMemcpyDeviceToHost));
            if (n_retry)
                for (uint32_t j = 0; j < n_jobs; ++j)
                    if (hit_count[j] == 0xFFFFFFFFu) {
                        pf::CtxErr{ctx} = "pf_align_batch: a job has more co-optimal alignments than the staging area holds";
                        status = PF_ERR_OVERFLOW;
                        break;
                    }
        }
    }
    return status;
}
