// K-GMM: the EM fit behind `PloidyFrost model` (reference src/GmmModel.cpp:259-394, driven by src/Main.cpp:636-692).
//
// The reference fits, for every candidate ploidy p, a mixture of g = p-1 Gaussians with FIXED means i/(g+1) to the allele
// frequencies of all variant sites: each EM iteration is one sequential pass for the log-likelihood (computeLogLikelihood,
// :259-276) and one for the responsibilities (emStep, :277-334), up to 1000 iterations x 9 models over ~1 M values on one
// core.  Here the values stay in HBM; one pass over them yields both the log-likelihood of the current parameters and the
// sums emStep needs (k_gmm_pass: grid-stride, fp64, fixed-order block reduction into per-block partials), a one-block
// kernel folds the partials in a fixed order, applies emStep's update rules and the loop test of emIterate (:371-385) and
// leaves everything in a device-resident state record (k_gmm_update).  The iteration loop is launch-bound (8 B per value
// and pass), so 16 pass/update pairs are captured once per fit in a hipGraph and replayed until the state says "done";
// kernels of a finished fit return at once.  fp64 sums are tree-shaped here and sequential in the reference: results agree to
// rounding (tests state the tolerance), the summation order is fixed so that runs are reproducible bit for bit.
#include <hip/hip_runtime.h>

#include <cfloat>
#include <cmath>
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

constexpr int GMM_MAXG = PF_GMM_MAX_GAUSS;
constexpr int GMM_BLOCK = 256;
constexpr int GMM_NVAL = 2 * GMM_MAXG + 2;  // per block: gaussSum[g], varSum[g], sum, log-likelihood
constexpr int GMM_PAIRS = 16;                // pass/update pairs per graph launch

struct GmmState {
    double w[GMM_MAXG], mean[GMM_MAXG], var[GMM_MAXG];
    double ll, last, delta;
    double m_thre, n_thre, max_delta;
    int32_t max_iter, gauss;
    uint32_t count, done, passes;
};

__device__ inline double wave_sum_f64(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    return v;
}

// one pass over the values with the current parameters: partial[block][...] in a fixed order
__global__ __launch_bounds__(GMM_BLOCK) void k_gmm_pass(const double *__restrict__ x, uint64_t n, const GmmState *__restrict__ st,
                                                        double *__restrict__ partial) {
    if (st->done) return;
    const int G = st->gauss;
    double w[GMM_MAXG], mean[GMM_MAXG], var2[GMM_MAXG], inv[GMM_MAXG], gs[GMM_MAXG], vs[GMM_MAXG];
#pragma unroll
    for (int i = 0; i < GMM_MAXG; ++i) {
        const bool on = i < G;
        w[i] = on ? st->w[i] : 0.0;
        mean[i] = on ? st->mean[i] : 0.0;
        const double v = on ? st->var[i] : 1.0;
        var2[i] = 2 * v;
        inv[i] = 1 / sqrt(2 * M_PI * v);   // getProbability (src/GmmModel.hpp:14-17), the factor that does not depend on x
        gs[i] = vs[i] = 0.0;
    }
    double sum = 0.0, ll = 0.0;
    const uint64_t stride = (uint64_t)gridDim.x * GMM_BLOCK;
    for (uint64_t j = (uint64_t)blockIdx.x * GMM_BLOCK + threadIdx.x; j < n; j += stride) {
        const double af = x[j];
        double p[GMM_MAXG], plain = 0.0, rowsum = 0.0;
#pragma unroll
        for (int i = 0; i < GMM_MAXG; ++i) {
            if (i < G) {
                const double d = af - mean[i];
                double q = w[i] * (inv[i] * exp(-(d * d / var2[i])));
                plain += q;                       // computeLogLikelihood: only the row sum is guarded
                if (q == 0.0) q = DBL_MIN;        // emStep: every term is
                p[i] = q;
                rowsum += q;
            }
        }
        if (plain == 0.0) plain = DBL_MIN;
        ll += log(plain);
#pragma unroll
        for (int i = 0; i < GMM_MAXG; ++i) {
            if (i < G) {
                const double r = p[i] / rowsum;
                const double d = af - mean[i];
                gs[i] += r;
                vs[i] += r * (d * d);
                sum += r;
            }
        }
    }
    __shared__ double red[GMM_BLOCK / 64][GMM_NVAL];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
    for (int i = 0; i < GMM_MAXG; ++i) {
        if (i < G) {
            const double a = wave_sum_f64(gs[i]), b = wave_sum_f64(vs[i]);
            if (lane == 0) { red[wv][i] = a; red[wv][GMM_MAXG + i] = b; }
        }
    }
    {
        const double a = wave_sum_f64(sum), b = wave_sum_f64(ll);
        if (lane == 0) { red[wv][2 * GMM_MAXG] = a; red[wv][2 * GMM_MAXG + 1] = b; }
    }
    __syncthreads();
    if (threadIdx.x < GMM_NVAL) {
        const int v = threadIdx.x;
        const bool used = v >= 2 * GMM_MAXG || (v % GMM_MAXG) < G;
        double t = 0.0;
        if (used)
            for (int q = 0; q < GMM_BLOCK / 64; ++q) t += red[q][v];
        partial[(size_t)blockIdx.x * GMM_NVAL + v] = t;
    }
}

// folds the partials, then emIterate's bookkeeping and emStep's update (src/GmmModel.cpp:277-334, 371-385)
__global__ __launch_bounds__(GMM_BLOCK) void k_gmm_update(const double *__restrict__ partial, int n_blocks, GmmState *st) {
    if (st->done) return;
    __shared__ double red[GMM_BLOCK / 64][GMM_NVAL];
    __shared__ double tot[GMM_NVAL];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    for (int v = 0; v < GMM_NVAL; ++v) {
        double t = 0.0;
        for (int b = threadIdx.x; b < n_blocks; b += GMM_BLOCK) t += partial[(size_t)b * GMM_NVAL + v];
        t = wave_sum_f64(t);
        if (lane == 0) red[wv][v] = t;
    }
    __syncthreads();
    if (threadIdx.x < GMM_NVAL) {
        double t = 0.0;
        for (int q = 0; q < GMM_BLOCK / 64; ++q) t += red[q][threadIdx.x];
        tot[threadIdx.x] = t;
    }
    __syncthreads();
    if (threadIdx.x != 0) return;
    const int G = st->gauss;
    const double ll_now = tot[2 * GMM_MAXG + 1];
    if (st->passes == 0) {           // logLikelihood = computeLogLikelihood(); deltaLogl = DBL_MAX
        st->ll = st->last = ll_now;
        st->delta = DBL_MAX;
    } else {                          // last = logLikelihood; logLikelihood = computeLogLikelihood(); ++count
        st->last = st->ll;
        st->ll = ll_now;
        st->delta = st->ll - st->last;
        st->count++;
    }
    st->passes++;
    if (!(st->delta > st->max_delta && st->count < (uint32_t)st->max_iter)) { st->done = 1; return; }
    // emStep with the sums of this pass
    const double sum = tot[2 * GMM_MAXG];
    double nw[GMM_MAXG], nv[GMM_MAXG];
    double max_w = -DBL_MAX, min_w = DBL_MAX;
    for (int i = 0; i < G; ++i) {
        double var = 1 / tot[i] * tot[GMM_MAXG + i];
        const double weight = tot[i] / sum;
        if (var == 0.0) var = DBL_MIN;
        nv[i] = var;
        nw[i] = weight;
        if (weight > max_w) max_w = weight;
        if (weight < min_w) min_w = weight;
    }
    if (max_w != nw[0] && max_w != nw[G - 1]) {
        if (min_w < (double)1 / G / st->m_thre) return;
        if (min_w < max_w / G / st->n_thre) return;
    }
    for (int i = 0; i < G; ++i) { st->var[i] = nv[i]; st->w[i] = nw[i]; }
}

}  // namespace pf

using namespace pf;

extern "C" {

int pf_gmm_upload(pf_ctx *ctx, const double *x, uint64_t n) {
    if (!ctx || (n && !x)) return PF_ERR_ARG;
    PF_HIP(hipSetDevice(ctx->device));
    double *dx = (double *)ctx_ws(ctx, WS_GMM_X, (size_t)n * 8);
    if (!dx) return PF_ERR_HIP;
    if (n) PF_HIP(hipMemcpyAsync(dx, x, (size_t)n * 8, hipMemcpyDefault, ctx->stream));
    PF_HIP(hipStreamSynchronize(ctx->stream));
    ctx->gmm_n = n;
    ctx->gmm_loaded = true;
    return PF_OK;
}

uint64_t pf_gmm_count(const pf_ctx *ctx) { return ctx && ctx->gmm_loaded ? ctx->gmm_n : 0; }

int pf_gmm_fit(pf_ctx *ctx, uint32_t gauss, double m_thre, double n_thre, int32_t max_iter, double max_delta, double *weights,
               double *means, double *vars, double *loglik, uint32_t *iterations) {
    if (!ctx || !ctx->gmm_loaded || gauss < 1 || gauss > (uint32_t)GMM_MAXG || max_iter < 0 || !weights || !means || !vars || !loglik) {
        if (ctx) pf::CtxErr{ctx} = "pf_gmm_fit: values not uploaded, or gauss outside 1..PF_GMM_MAX_GAUSS";
        return PF_ERR_ARG;
    }
    PF_HIP(hipSetDevice(ctx->device));
    const uint64_t n = ctx->gmm_n;
    const double *dx = (const double *)ctx_ws(ctx, WS_GMM_X, (size_t)n * 8);
    int n_blocks = (int)std::min<uint64_t>((n + (uint64_t)GMM_BLOCK * 4 - 1) / ((uint64_t)GMM_BLOCK * 4), (uint64_t)ctx->n_cu * 4);
    if (n_blocks < 1) n_blocks = 1;
    GmmState *dst = (GmmState *)ctx_ws(ctx, WS_GMM_STATE, sizeof(GmmState));
    double *dpart = (double *)ctx_ws(ctx, WS_GMM_PART, (size_t)n_blocks * GMM_NVAL * 8);
    if (!dx || !dst || !dpart) return PF_ERR_HIP;
    GmmState h{};
    for (uint32_t i = 1; i <= gauss; ++i) {  // GmmModel::resize, src/GmmModel.cpp:8-20
        h.mean[i - 1] = (double)i / (gauss + 1);
        h.w[i - 1] = (double)1 / gauss;
        h.var[i - 1] = 0.01;
    }
    h.m_thre = m_thre;
    h.n_thre = n_thre;
    h.max_delta = max_delta;
    h.max_iter = max_iter;
    h.gauss = (int32_t)gauss;
    PF_HIP(hipMemcpyAsync(dst, &h, sizeof h, hipMemcpyHostToDevice, ctx->stream));
    PF_HIP(hipStreamSynchronize(ctx->stream));
    // the loop body, captured once (a stream that cannot be captured -- the legacy default stream -- gets plain launches)
    hipGraph_t graph = nullptr;
    hipGraphExec_t exec = nullptr;
    auto enqueue_pairs = [&]() {
        for (int it = 0; it < GMM_PAIRS; ++it) {
            k_gmm_pass<<<n_blocks, GMM_BLOCK, 0, ctx->stream>>>(dx, n, dst, dpart);
            k_gmm_update<<<1, GMM_BLOCK, 0, ctx->stream>>>(dpart, n_blocks, dst);
        }
    };
    if (hipStreamBeginCapture(ctx->stream, hipStreamCaptureModeThreadLocal) == hipSuccess) {
        enqueue_pairs();
        hipError_t ce = hipStreamEndCapture(ctx->stream, &graph);
        if (ce == hipSuccess && graph) ce = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
        if (ce != hipSuccess || !exec) {
            if (graph) hipGraphDestroy(graph);
            pf::CtxErr{ctx} = std::string("K-GMM graph capture: ") + hipGetErrorString(ce);
            return PF_ERR_HIP;
        }
    } else {
        (void)hipGetLastError();
    }
    int rc = PF_OK;
    // at most max_iter + 1 passes are needed; every round runs GMM_PAIRS of them
    for (int64_t launched = 0; launched <= (int64_t)max_iter + GMM_PAIRS; launched += GMM_PAIRS) {
        hipError_t ce = hipSuccess;
        ctx_begin(ctx, PF_K_GMM);
        if (exec) ce = hipGraphLaunch(exec, ctx->stream);
        else { enqueue_pairs(); ce = hipGetLastError(); }
        ctx_end(ctx);
        if (ce == hipSuccess) ce = hipMemcpyAsync(&h, dst, sizeof h, hipMemcpyDeviceToHost, ctx->stream);
        if (ce == hipSuccess) ce = hipStreamSynchronize(ctx->stream);
        if (ce != hipSuccess) { pf::CtxErr{ctx} = std::string("K-GMM launch: ") + hipGetErrorString(ce); rc = PF_ERR_HIP; break; }
        if (h.done) break;
    }
    if (exec) hipGraphExecDestroy(exec);
    if (graph) hipGraphDestroy(graph);
    if (rc != PF_OK) return rc;
    if (!h.done) { pf::CtxErr{ctx} = "K-GMM did not finish within its iteration bound"; return PF_ERR_HIP; }
    for (uint32_t i = 0; i < gauss; ++i) { weights[i] = h.w[i]; means[i] = h.mean[i]; vars[i] = h.var[i]; }
    *loglik = h.ll;
    if (iterations) *iterations = h.count;
    return PF_OK;
}

}  // extern "C"
