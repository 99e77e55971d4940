// The resident calling pipeline: CDBG::ploidyEstimation_ptr (reference src/CDBG.cpp:1101-1705) with everything between
// the commit replay and the file system on the MI355X.  The host uploads the MyUnitig state, runs the light sequential
// part of the driver loop on compact side records, and appends the text slabs it gets back to the ten result files;
// bubbles never exist on the host as objects.
//
//   K-SCAN    k_call_sides   one thread per unitig: for each open endpoint side what the driver loop (:1146-1222, 1347-1363)
//                            would do there -- exit by first successors, ownership (reference-string compare on the
//                            2-bit words), coverage gate, sortSeq_simple -- from static state only
//   K-PREP    k_call_prep    one thread per selected bubble: strict bubbles become K-BUBBLE tasks whose paths are oriented
//                            unitigs; branching ones are queued for K-PATHS; every task lands in the work queue of its
//                            LDS size class
//   K-PATHS   k_call_paths   one wavefront per branching bubble: the two-stack enumeration of all s->t walks (:1364-1412),
//                            path strings decoded lane-parallel from the 2-bit graph, sortSeq_branching as a rank sort
//   K-BUBBLE  (pf_bubble.hip) SeqAlign::SequenceAlignment for every task
//   K-SITES   k_call_sites   one wavefront per branching bubble: per-site k-length strings (:1448-1600), de-duplicated per
//                            allele group in std::set order, looked up in the count table (readCov(string), :29-60),
//                            group coverages accumulated in the reference's order
//   K-TEXT    k_call_format  one thread per bubble, twice: measure, (scan,) write -- the rows of alignseq.txt,
//                            allele_frequency.txt and the eight {bi,tri,tetra,penta}{cov,fre}.txt files, numbers printed
//                            as `ostream << double` does (pf_format_dev.hpp); bubble numbering (var_count) by a scan
//
// All integer / byte work with data-dependent control flow; HBM traffic is the text itself.  No MFMA.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstddef>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "pf_bubble_launch.hpp"
#include "pf_alnpack.hpp"
#include "pf_call_dev.hpp"
#include "pf_colored_dev.hpp"
#include "pf_cov_stream.hpp"
#include "pf_ctx.hpp"
#include "pf_device_common.hpp"
#include "pf_format_dev.hpp"
#include "pf_pair_dev.hpp"
#include "pf_scan.hpp"
#include "pf_stack_dev.hpp"
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

#define NEED_TEXT(buf, bytes) do { if (!(buf).ensure(bytes)) { pf::CtxErr{ctx} = "pf_call_text: out of device memory"; return PF_ERR_HIP; } } while (0)

#include "pf_call_kernels.hpp"

using namespace pf_call;

namespace pf {

void call_state_create(pf_ctx *ctx) { (void)state_of(ctx); }
void call_destroy(pf_ctx *ctx) {
    if (!ctx->call) return;
    ctx->call->release_all();
    delete ctx->call;
    ctx->call = nullptr;
}
void call_invalidate(pf_ctx *ctx) {  // a new graph or count table: resident scan results no longer apply
    if (!ctx->call) return;
    ctx->call->have_cov = false;
    ctx->call->have_state = false;
    ctx->call->n_sides = ctx->call->n_tasks = 0;
}

}  // namespace pf

// =====================================================================================================================
// T1 state written on the device (pf_replay_device, pf_cc.hip): the arrays pf_call_set_state would fill, for their writer
namespace pf {
int call_state_arrays(pf_ctx *ctx, uint8_t **flags, uint32_t **plus, uint32_t **minus) {
    if (!ctx->d_seq || !ctx->has_adj) { pf::CtxErr{ctx} = "T1 state: graph and adjacency first"; return PF_ERR_ARG; }
    CallState *S = state_of(ctx);
    const size_t N = ctx->N;
    if (!S->flags.ensure(N + 1) || !S->plus.ensure(N * 4) || !S->minus.ensure(N * 4)) { pf::CtxErr{ctx} = "T1 state: out of device memory"; return PF_ERR_HIP; }
    *flags = S->flags.as<uint8_t>();
    *plus = S->plus.as<uint32_t>();
    *minus = S->minus.as<uint32_t>();
    S->have_state = false;
    return PF_OK;
}
void call_state_resident(pf_ctx *ctx) {
    CallState *S = state_of(ctx);
    S->have_state = true;
    S->n_sides = S->n_tasks = 0;
}
}  // namespace pf

extern "C" {

int pf_format_doubles(pf_ctx *ctx, const double *values, uint64_t n, char *text, uint8_t *len) {
    if (!ctx || (n && (!values || !text || !len))) return PF_ERR_ARG;
    if (n == 0) return PF_OK;
    PF_HIP(hipSetDevice(ctx->device));
    DevTmp<double> dx;
    DevTmp<char> dt;
    DevTmp<uint8_t> dl;
    PF_HIP(dx.alloc(n * 8));
    PF_HIP(dt.alloc(n * 32));
    PF_HIP(dl.alloc(n));
    PF_HIP(hipMemcpyAsync(dx.p, values, n * 8, hipMemcpyDefault, ctx->stream));
    k_format_doubles<<<(unsigned)((n + 255) / 256), 256, 0, ctx->stream>>>(dx.p, n, dt.p, dl.p);
    PF_HIP(hipGetLastError());
    PF_HIP(hipMemcpyAsync(text, dt.p, n * 32, hipMemcpyDefault, ctx->stream));
    PF_HIP(hipMemcpyAsync(len, dl.p, n, hipMemcpyDefault, ctx->stream));
    PF_HIP(hipStreamSynchronize(ctx->stream));
    return PF_OK;
}

int pf_call_set_state(pf_ctx *ctx, const uint8_t *flags, const uint32_t *plus, const uint32_t *minus) {
    if (!ctx || !flags || !plus || !minus) return PF_ERR_ARG;
    if (!ctx->d_seq || !ctx->has_adj) { pf::CtxErr{ctx} = "pf_call_set_state: graph and adjacency first"; return PF_ERR_ARG; }
    PF_HIP(hipSetDevice(ctx->device));
    CallState *S = state_of(ctx);
    const size_t N = ctx->N;
    if (!S->flags.ensure(N + 1) || !S->plus.ensure(N * 4) || !S->minus.ensure(N * 4)) { pf::CtxErr{ctx} = "pf_call_set_state: out of device memory"; return PF_ERR_HIP; }
    PF_HIP(hipMemcpyAsync(S->flags.p, flags, N, hipMemcpyDefault, ctx->stream));
    PF_HIP(hipMemcpyAsync(S->plus.p, plus, N * 4, hipMemcpyDefault, ctx->stream));
    PF_HIP(hipMemcpyAsync(S->minus.p, minus, N * 4, hipMemcpyDefault, ctx->stream));
    PF_HIP(hipStreamSynchronize(ctx->stream));
    S->have_state = true;
    S->n_sides = S->n_tasks = 0;
    return PF_OK;
}

int pf_call_get_state(pf_ctx *ctx, uint8_t *flags, uint32_t *plus, uint32_t *minus) {
    if (!ctx || !ctx->call) return PF_ERR_ARG;
    CallState *S = ctx->call;
    if (!S->have_state) { pf::CtxErr{ctx} = "pf_call_get_state: no T1 state on the device"; return PF_ERR_ARG; }
    PF_HIP(hipSetDevice(ctx->device));
    const size_t N = ctx->N;
    PF_HIP(hipStreamSynchronize(ctx->stream));
    if (flags) PF_HIP(hipMemcpy(flags, S->flags.p, N, hipMemcpyDeviceToHost));
    if (plus) PF_HIP(hipMemcpy(plus, S->plus.p, N * 4, hipMemcpyDeviceToHost));
    if (minus) PF_HIP(hipMemcpy(minus, S->minus.p, N * 4, hipMemcpyDeviceToHost));
    return PF_OK;
}

int pf_call_set_format(pf_ctx *ctx, int reference_mt) {
    if (!ctx) return PF_ERR_ARG;
    state_of(ctx)->mt_format = reference_mt != 0;
    return PF_OK;
}

int pf_superbubble_rows(pf_ctx *ctx, int colored_rule, uint64_t *n_rows, uint64_t *text_len) {
    if (!ctx || !n_rows || !text_len) return PF_ERR_ARG;
    CallState *S = ctx->call;
    if (!S || !S->have_state) { pf::CtxErr{ctx} = "pf_superbubble_rows: pf_call_set_state first"; return PF_ERR_ARG; }
    PF_HIP(hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    const uint32_t N = ctx->N;
    const size_t n1 = (size_t)N + 1;
    if (!S->sb_cnt.ensure(n1 * 4) || !S->sb_base.ensure(n1 * 4) || !S->sb_sizes.ensure(n1 * 4) || !S->sb_offs.ensure(n1 * 8)) {
        pf::CtxErr{ctx} = "pf_superbubble_rows: out of device memory";
        return PF_ERR_HIP;
    }
    SbArgs a;
    a.flags = S->flags.as<uint8_t>(); a.plus = S->plus.as<uint32_t>(); a.minus = S->minus.as<uint32_t>(); a.N = N; a.colored = colored_rule;
    a.first_id = S->mt_format ? 0u : 1u;
    a.row_base = S->sb_base.as<uint32_t>(); a.sizes = S->sb_sizes.as<uint32_t>(); a.offs = S->sb_offs.as<uint64_t>(); a.out = nullptr;
    const unsigned grid = (unsigned)((n1 + 255) / 256);
    k_sb_count<<<grid, 256, 0, st>>>(a, S->sb_cnt.as<uint32_t>());
    if (!S->scan_tmp.ensure(scan_scratch_bytes(n1))) { pf::CtxErr{ctx} = "pf_superbubble_rows: out of device memory"; return PF_ERR_HIP; }
    PF_HIP(scan_exclusive_u32(S->sb_cnt.as<uint32_t>(), S->sb_base.as<uint32_t>(), n1, S->scan_tmp.p, st));
    ctx_begin(ctx, PF_K_CALL_FORMAT);
    k_sb_format<false><<<grid, 256, 0, st>>>(a);
    ctx_end(ctx);
    PF_HIP(scan_exclusive_u32_u64(S->sb_sizes.as<uint32_t>(), S->sb_offs.as<uint64_t>(), n1, S->scan_tmp.p, st));
    uint32_t rows = 0;
    uint64_t len = 0;
    PF_HIP(hipMemcpyAsync(&rows, S->sb_base.as<uint32_t>() + N, 4, hipMemcpyDeviceToHost, st));
    PF_HIP(hipMemcpyAsync(&len, S->sb_offs.as<uint64_t>() + N, 8, hipMemcpyDeviceToHost, st));
    PF_HIP(hipStreamSynchronize(st));
    if (!S->sb_out.ensure(std::max<uint64_t>(len, 16))) { pf::CtxErr{ctx} = "pf_superbubble_rows: out of device memory"; return PF_ERR_HIP; }
    a.out = S->sb_out.as<char>();
    // the write pass runs on the copy stream, in front of pf_superbubble_fetch's copy and beside whatever the caller launches next
    // (PloidyEstimation's coverage and scan only read the state): the caller has the counts and does not wait for the text
    hipStream_t wst = S->copy_stream ? S->copy_stream : st;
    ctx_begin_on(ctx, PF_K_CALL_FORMAT, wst);
    k_sb_format<true><<<grid, 256, 0, wst>>>(a);
    ctx_end_on(ctx, wst);
    PF_HIP(hipGetLastError());
    if (wst == st) PF_HIP(hipStreamSynchronize(st));
    S->sb_len = len;
    *n_rows = rows;
    *text_len = len;
    return PF_OK;
}

int pf_superbubble_fetch(pf_ctx *ctx, char *dst, uint64_t len) {
    if (!ctx || !ctx->call || (len && !dst)) return PF_ERR_ARG;
    CallState *S = ctx->call;
    if (len > S->sb_len) return PF_ERR_ARG;
    if (len == 0) return PF_OK;
    if (hipSetDevice(ctx->device) != hipSuccess || !S->copy_stream) return PF_ERR_HIP;
    size_t tl_at = (size_t)-1;
    (void)ctx_begin_at(ctx, PF_K_COPY_TEXT, S->copy_stream, &tl_at);
    if (hipMemcpyAsync(dst, S->sb_out.p, (size_t)len, hipMemcpyDeviceToHost, S->copy_stream) != hipSuccess) return PF_ERR_HIP;
    ctx_end_at(ctx, tl_at, S->copy_stream);
    if (hipStreamSynchronize(S->copy_stream) != hipSuccess) return PF_ERR_HIP;
    return PF_OK;
}

int pf_call_set_colours(pf_ctx *ctx, uint32_t n_colors, const uint64_t *full_mask, const uint64_t *size_total, const uint32_t *part_first,
                        const uint32_t *part_colour, const uint64_t *part_word, const uint64_t *part_bits, uint64_t n_part, uint64_t n_words) {
    if (!ctx || !ctx->d_seq) { if (ctx) pf::CtxErr{ctx} = "pf_call_set_colours: graph first"; return PF_ERR_ARG; }
    PF_HIP(hipSetDevice(ctx->device));
    CallState *S = state_of(ctx);
    S->n_colors = 0;
    S->have_cov = false;
    if (!n_colors) return PF_OK;
    if (n_colors > PF_MAX_COLORS || n_colors != ctx->n_colors || !full_mask || !size_total || !part_first || (n_part && (!part_colour || !part_word)) ||
        (n_words && !part_bits)) {
        pf::CtxErr{ctx} = "pf_call_set_colours: the colour sets of every unitig, for the colours of the uploaded databases, are required";
        return PF_ERR_ARG;
    }
    const size_t N = ctx->N, CW = (n_colors + 63) / 64;
    if (!S->col_low.ensure((size_t)n_colors * 4) || !S->col_up.ensure((size_t)n_colors * 4) || !S->col_full.ensure(N * CW * 8) || !S->col_size.ensure(N * 8) ||
        !S->part_first.ensure((N + 1) * 4) || !S->part_colour.ensure((n_part + 1) * 4) || !S->part_word.ensure((n_part + 1) * 8) ||
        !S->part_bits.ensure((n_words + 1) * 8)) {
        pf::CtxErr{ctx} = "pf_call_set_colours: out of device memory";
        return PF_ERR_HIP;
    }
    hipStream_t st = ctx->stream;
    PF_HIP(hipMemsetAsync(S->col_low.p, 0, (size_t)n_colors * 4, st));
    PF_HIP(hipMemsetAsync(S->col_up.p, 0xFF, (size_t)n_colors * 4, st));
    PF_HIP(hipMemcpyAsync(S->col_full.p, full_mask, N * CW * 8, hipMemcpyDefault, st));
    PF_HIP(hipMemcpyAsync(S->col_size.p, size_total, N * 8, hipMemcpyDefault, st));
    PF_HIP(hipMemcpyAsync(S->part_first.p, part_first, (N + 1) * 4, hipMemcpyDefault, st));
    if (n_part) {
        PF_HIP(hipMemcpyAsync(S->part_colour.p, part_colour, n_part * 4, hipMemcpyDefault, st));
        PF_HIP(hipMemcpyAsync(S->part_word.p, part_word, n_part * 8, hipMemcpyDefault, st));
    }
    if (n_words) PF_HIP(hipMemcpyAsync(S->part_bits.p, part_bits, n_words * 8, hipMemcpyDefault, st));
    PF_HIP(hipStreamSynchronize(st));
    S->n_colors = n_colors;
    S->col_words = (uint32_t)CW;
    return PF_OK;
}

int pf_call_set_cutoffs(pf_ctx *ctx, uint32_t n_colors, const uint32_t *lower, const uint32_t *upper) {
    if (!ctx || !ctx->call || !ctx->call->n_colors || n_colors != ctx->call->n_colors || !lower || !upper) {
        if (ctx) pf::CtxErr{ctx} = "pf_call_set_cutoffs: pf_call_set_colours first; one (lower, upper) per colour";
        return PF_ERR_ARG;
    }
    PF_HIP(hipSetDevice(ctx->device));
    CallState *S = ctx->call;
    PF_HIP(hipMemcpyAsync(S->col_low.p, lower, (size_t)n_colors * 4, hipMemcpyDefault, ctx->stream));
    PF_HIP(hipMemcpyAsync(S->col_up.p, upper, (size_t)n_colors * 4, hipMemcpyDefault, ctx->stream));
    PF_HIP(hipStreamSynchronize(ctx->stream));
    return PF_OK;
}

int pf_call_coverage(pf_ctx *ctx) {
    if (ctx && ctx->call && ctx->call->n_colors) {
        // colored: CCDBG::readCovUni for every (colour, unitig), left where the scan, K-SITES and K-TEXT read it
        if (!ctx->d_seq || !ctx->d_ctab) { pf::CtxErr{ctx} = "pf_call_coverage: graph and count tables first"; return PF_ERR_ARG; }
        PF_HIP(hipSetDevice(ctx->device));
        CallState *S = ctx->call;
        const size_t n = (size_t)ctx->N * S->n_colors;
        if (!S->ccov_sum.ensure(n * 8) || !S->ccov_min.ensure(n * 4) || !S->ccov_max.ensure(n * 4) || !S->ccov_miss.ensure(n)) {
            pf::CtxErr{ctx} = "pf_call_coverage: out of device memory";
            return PF_ERR_HIP;
        }
        const int st = pf_unitig_cov_colored(ctx, 0, ctx->N, S->ccov_sum.as<uint64_t>(), S->ccov_min.as<uint32_t>(), S->ccov_max.as<uint32_t>(), S->ccov_miss.as<uint8_t>());
        if (st != PF_OK) return st;
        S->per_strand = false;
        S->have_cov = true;
        return PF_OK;
    }
    if (!ctx || !ctx->d_seq || !ctx->d_tab) { if (ctx) pf::CtxErr{ctx} = "pf_call_coverage: graph and count table first"; return PF_ERR_ARG; }
    PF_HIP(hipSetDevice(ctx->device));
    CallState *S = state_of(ctx);
    const size_t N = ctx->N;
    const int slots = ctx->tab_exact ? 2 : 1;
    if (!S->cov_sum.ensure(N * slots * 8) || !S->cov_min.ensure(N * slots * 4) || !S->cov_miss.ensure(N * slots)) {
        pf::CtxErr{ctx} = "pf_call_coverage: out of device memory";
        return PF_ERR_HIP;
    }
    int st;
    if (!ctx->tab_exact) {
        st = pf_unitig_cov(ctx, 0, (uint32_t)N, S->cov_sum.as<uint64_t>(), S->cov_min.as<uint32_t>(), S->cov_miss.as<uint8_t>());
    } else {
        st = pf_unitig_cov_exact(ctx, 0, (uint32_t)N, 0, S->cov_sum.as<uint64_t>(), S->cov_min.as<uint32_t>(), S->cov_miss.as<uint8_t>());
        if (st == PF_OK || st == PF_ERR_MISSING_KMER)
            st = pf_unitig_cov_exact(ctx, 0, (uint32_t)N, 1, S->cov_sum.as<uint64_t>() + N, S->cov_min.as<uint32_t>() + N, S->cov_miss.as<uint8_t>() + N);
    }
    if (st != PF_OK && st != PF_ERR_MISSING_KMER) return st;  // a missing k-mer matters only where the driver loop reads it
    S->per_strand = ctx->tab_exact;
    S->have_cov = true;
    return PF_OK;
}

int pf_call_scan(pf_ctx *ctx, uint32_t lower, uint32_t upper, uint64_t *n_sides) {
    if (!ctx || !n_sides) return PF_ERR_ARG;
    CallState *S = ctx->call;
    if (!S || !S->have_state || !S->have_cov) { pf::CtxErr{ctx} = "pf_call_scan: pf_call_set_state and pf_call_coverage first"; return PF_ERR_ARG; }
    PF_HIP(hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    const uint32_t N = ctx->N;
    if (!S->side_cnt.ensure(((size_t)N + 1) * 4) || !S->side_base.ensure(((size_t)N + 1) * 4)) { pf::CtxErr{ctx} = "pf_call_scan: out of device memory"; return PF_ERR_HIP; }
    k_call_count_sides<<<(N + 1 + 255) / 256, 256, 0, st>>>(S->flags.as<uint8_t>(), N, S->side_cnt.as<uint32_t>());
    if (!S->scan_tmp.ensure(scan_scratch_bytes((uint64_t)N + 1))) { pf::CtxErr{ctx} = "pf_call_scan: out of device memory"; return PF_ERR_HIP; }
    PF_HIP(scan_exclusive_u32(S->side_cnt.as<uint32_t>(), S->side_base.as<uint32_t>(), (uint64_t)N + 1, S->scan_tmp.p, st));
    uint32_t total = 0;
    PF_HIP(hipMemcpyAsync(&total, S->side_base.as<uint32_t>() + N, 4, hipMemcpyDeviceToHost, st));
    PF_HIP(hipStreamSynchronize(st));
    S->n_sides = total;
    S->n_tasks = 0;
    S->low = lower;
    S->up = upper;
    *n_sides = total;
    if (total == 0) return PF_OK;
    if (!S->sides.ensure((size_t)total * sizeof(pf_call_side)) || !S->ctask.ensure((size_t)total * sizeof(CallTask)) ||
        !S->target.ensure((size_t)total * 4)) {
        pf::CtxErr{ctx} = "pf_call_scan: out of device memory";
        return PF_ERR_HIP;
    }
    ScanArgs a;
    a.flags = S->flags.as<uint8_t>(); a.plus = S->plus.as<uint32_t>(); a.minus = S->minus.as<uint32_t>();
    a.succ = ctx->d_succ; a.pred = ctx->d_pred; a.seq = ctx->d_seq; a.off = ctx->d_off; a.len = ctx->d_len; a.N = N; a.k = ctx->k;
    a.cov_sum = S->cov_sum.as<uint64_t>(); a.cov_min = S->cov_min.as<uint32_t>(); a.cov_miss = S->cov_miss.as<uint8_t>();
    a.per_strand = S->per_strand; a.low = lower; a.up = upper;
    a.side_base = S->side_base.as<uint32_t>(); a.sides = S->sides.as<pf_call_side>(); a.tasks = S->ctask.as<CallTask>();
    a.target = S->target.as<uint32_t>();
    a.n_colors = S->n_colors;
    a.ccov_sum = S->ccov_sum.as<uint64_t>(); a.ccov_min = S->ccov_min.as<uint32_t>(); a.ccov_max = S->ccov_max.as<uint32_t>(); a.ccov_miss = S->ccov_miss.as<uint8_t>();
    a.clow = S->col_low.as<uint32_t>(); a.cup = S->col_up.as<uint32_t>(); a.full = S->col_full.as<uint64_t>(); a.cwords = S->col_words; a.size_total = S->col_size.as<uint64_t>();
    ctx_begin(ctx, PF_K_CALL_SCAN);
    if (S->n_colors) k_call_sides<true><<<(N + 255) / 256, 256, 0, st>>>(a);
    else k_call_sides<false><<<(N + 255) / 256, 256, 0, st>>>(a);
    ctx_end(ctx);
    ctx_units(ctx, PF_K_CALL_SCAN, N);
    PF_HIP(hipGetLastError());
    return PF_OK;
}

int pf_call_resolve(pf_ctx *ctx, uint64_t *n_bubbles, uint32_t *err, uint32_t *err_unitig) {
    if (!ctx || !ctx->call || !n_bubbles || !err || !err_unitig) return PF_ERR_ARG;
    CallState *S = ctx->call;
    *n_bubbles = 0;
    *err = 0;
    *err_unitig = 0;
    S->n_tasks = 0;
    const uint64_t n = S->n_sides;
    if (n == 0) return PF_OK;
    PF_HIP(hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    if (!S->pending.ensure(n * 4) || !S->killed.ensure(n) || !S->rstate.ensure(n) || !S->rflag.ensure(n * 4) || !S->rsmall.ensure(64) ||
        !S->kept.ensure(n * 4)) {
        pf::CtxErr{ctx} = "pf_call_resolve: out of device memory";
        return PF_ERR_HIP;
    }
    PF_HIP(hipMemsetAsync(S->pending.p, 0, n * 4, st));
    PF_HIP(hipMemsetAsync(S->killed.p, 0, n, st));
    PF_HIP(hipMemsetAsync(S->rstate.p, 0, n, st));
    unsigned int *small = S->rsmall.as<unsigned int>();   // [0] undecided, [1] first_err, [2] n selected
    ResolveArgs a;
    a.sides = S->sides.as<pf_call_side>(); a.target = S->target.as<uint32_t>(); a.n = (uint32_t)n; a.pending = S->pending.as<int>();
    a.killed = S->killed.as<uint8_t>(); a.state = S->rstate.as<uint8_t>(); a.flag = S->rflag.as<uint32_t>();
    a.undecided = small; a.first_err = small + 1;
    const unsigned int init[3] = {0, 0xFFFFFFFFu, 0};
    PF_HIP(hipMemcpyAsync(small, init, 12, hipMemcpyHostToDevice, st));
    const unsigned grid = (unsigned)((n + 255) / 256);
    k_call_pending<<<grid, 256, 0, st>>>(a);
    for (uint64_t round = 0;; ++round) {
        if (round > n + 2) { pf::CtxErr{ctx} = "pf_call_resolve: the driver pass does not settle"; return PF_ERR_ARG; }
        k_call_resolve<<<grid, 256, 0, st>>>(a);
        PF_HIP(hipMemsetAsync(small, 0, 4, st));
        k_call_resolve<<<grid, 256, 0, st>>>(a);   // two rounds per look at the counter: the common case needs exactly two
        unsigned int h[2];
        PF_HIP(hipMemcpyAsync(h, small, 8, hipMemcpyDeviceToHost, st));
        PF_HIP(hipStreamSynchronize(st));
        if (h[1] != 0xFFFFFFFFu && h[0] == 0) {   // every side settled and an alive one carries an error: the first in order counts
            pf_call_side bad;
            PF_HIP(hipMemcpy(&bad, S->sides.as<pf_call_side>() + h[1], sizeof(bad), hipMemcpyDeviceToHost));
            *err = bad.err;
            *err_unitig = bad.err_unitig;
            return PF_OK;
        }
        if (h[0] == 0) break;
    }
    // the called sides, ascending
    if (!S->scan_tmp.ensure(scan_scratch_bytes(n))) { pf::CtxErr{ctx} = "pf_call_resolve: out of device memory"; return PF_ERR_HIP; }
    PF_HIP(select_flagged_u32(S->rflag.as<uint32_t>(), S->kept.as<uint32_t>(), small + 2, nullptr, n, S->scan_tmp.p, st));
    unsigned int sel = 0;
    PF_HIP(hipMemcpyAsync(&sel, small + 2, 4, hipMemcpyDeviceToHost, st));
    PF_HIP(hipStreamSynchronize(st));
    S->n_tasks = sel;
    *n_bubbles = sel;
    return PF_OK;
}

int pf_call_sides(pf_ctx *ctx, pf_call_side *out, uint64_t cap) {
    if (!ctx || !ctx->call) return PF_ERR_ARG;
    CallState *S = ctx->call;
    if (cap < S->n_sides || (S->n_sides && !out)) return PF_ERR_OVERFLOW;
    if (S->n_sides == 0) return PF_OK;
    PF_HIP(hipSetDevice(ctx->device));
    PF_HIP(hipMemcpyAsync(out, S->sides.p, (size_t)S->n_sides * sizeof(pf_call_side), hipMemcpyDefault, ctx->stream));
    PF_HIP(hipStreamSynchronize(ctx->stream));
    return PF_OK;
}

int pf_call_select(pf_ctx *ctx, const uint32_t *side_index, uint64_t n_tasks) {
    if (!ctx || !ctx->call || (n_tasks && !side_index)) return PF_ERR_ARG;
    CallState *S = ctx->call;
    if (n_tasks > S->n_sides) { pf::CtxErr{ctx} = "pf_call_select: more bubbles than open sides"; return PF_ERR_ARG; }
    PF_HIP(hipSetDevice(ctx->device));
    if (!S->kept.ensure(std::max<size_t>(n_tasks, 1) * 4)) { pf::CtxErr{ctx} = "pf_call_select: out of device memory"; return PF_ERR_HIP; }
    if (n_tasks) PF_HIP(hipMemcpyAsync(S->kept.p, side_index, (size_t)n_tasks * 4, hipMemcpyDefault, ctx->stream));
    PF_HIP(hipStreamSynchronize(ctx->stream));
    S->n_tasks = n_tasks;
    return PF_OK;
}

// one batch, first half: bubbles [t0, t1) of the selection up to the site coverages; out->n_called tells how far var_count advances
int pf_call_align_lane(pf_ctx *ctx, int lane, uint64_t t0, uint64_t t1, uint32_t complex_size, double match, double mismatch, double gap,
                       pf_call_result *out) {
    if (!ctx || !out || lane < 0 || lane >= PF_CALL_LANES) return PF_ERR_ARG;
    CallState *S = ctx->call;
    if (!S || t0 > t1 || t1 > S->n_tasks) { pf::CtxErr{ctx} = "pf_call_align: range outside the selection"; return PF_ERR_ARG; }
    memset(out, 0, sizeof(*out));
    CallState::AlignOut &O = S->lane[lane];
    O.nb = 0;
    O.t0 = t0;
    O.cur = pf_call_result{};   // (an empty slice -- fewer bubbles than ranks -- must not hand pf_call_text the previous pass's counters)
    if (t1 == t0) return PF_OK;
    if (t1 - t0 > (1u << 24)) { pf::CtxErr{ctx} = "pf_call_align: at most 2^24 bubbles per batch"; return PF_ERR_ARG; }
    PF_HIP(hipSetDevice(ctx->device));
    CallState::AlignWork &W = S->work[lane];
    if (lane != 0 && !W.stream) { PF_HIP(lane_stream_create(&W.stream, lane)); W.own_stream = true; }
    hipStream_t st = lane == 0 ? ctx->stream : W.stream;   // (lane 0: the context's stream, whatever pf_set_stream made it since)
    // the write pass of K-TEXT over what this lane held may still be running (pf_call_text_range_lane does not wait for it)
    for (hipEvent_t e : O.read_ev) if (e) PF_HIP(hipStreamWaitEvent(st, e, 0));
    // (another lane's stream needs no event to wait for: the scan and the selection on the context's stream ended in host waits --
    // pf_call_resolve / pf_call_select hand the host the number of bubbles this call's range is cut from)
    // launch timing by place: calls on other lanes time their launches at the same time
    size_t tl_at = (size_t)-1;
    auto tbegin = [&](int kernel, hipStream_t s) { (void)ctx_begin_at(ctx, kernel, s, &tl_at); };
    auto tend = [&](hipStream_t s) { ctx_end_at(ctx, tl_at, s); };
    const uint32_t nb = (uint32_t)(t1 - t0);
    const int k = ctx->k;
    const bool trace_stages = getenv("PF_TRACE_ALIGN") != nullptr;   // where a call's time goes (a first call above all)
    const auto t_enter = std::chrono::steady_clock::now();
    auto ta = [&](const char *what) {
        if (trace_stages) fprintf(stderr, "[pf_call_align]   %-34s %.2f ms\n", what, std::chrono::duration<double>(std::chrono::steady_clock::now() - t_enter).count() * 1e3);
    };
    const char *oom = "pf_call_run: out of device memory";
#define NEED(buf, bytes) do { if (!(buf).ensure(bytes)) { pf::CtxErr{ctx} = oom; return PF_ERR_HIP; } } while (0)
    NEED(W.counters, sizeof(CallCounters));
    NEED(W.btask, (size_t)nb * sizeof(pf_bubble_task));
    NEED(W.queues, (size_t)NQ * nb * 4);
    NEED(W.blist, (size_t)nb * 4);
    NEED(O.res, (size_t)nb * sizeof(pf_bubble_result));
    NEED(O.sv_off, (size_t)nb * 8);
    NEED(W.has, (size_t)nb * 4);
    NEED(O.vc, (size_t)nb * 4);
    CallCounters *d_cnt = W.counters.as<CallCounters>();
    CallCounters hc;
    // K-PATHS scratch: stacks sized by the complex size (a non-complex bubble has at most that many vertices)
    const uint32_t depth_cap = std::max<uint32_t>(complex_size + 4, 16);
    const uint64_t paths_per_wave = ((256 * 8 + 256 * 4 + (10ull * depth_cap + 4) * 4) + 255) & ~255ull;
    constexpr int paths_per_cu = 16;
    const int paths_grid = ctx->n_cu * paths_per_cu;
    NEED(W.paths_scr, paths_per_wave * paths_grid);
    // bubbles of more than 255 walks in one range: the list grows to what an attempt asked for (advisor, round 3: an overflow used
    // to be reported as "more than 65535 paths", a refusal of a run no bubble of which had that many)
    if (W.mlist_cap < 4096) W.mlist_cap = 4096;
    NEED(W.mlist, (size_t)W.mlist_cap * 4);
    PathArgs ph_keep = {};
    const int paths_force_scratch = [] { const char *e = getenv("PF_PATHS_SCRATCH"); return e && atoi(e) ? 1 : 0; }();   // (read per call: tests)

    // ---- K-PREP, K-SNP, K-PATHS, K-BUBBLE; every pool grows until the batch fits (first batches of a run only) ----
    NEED(W.slist, (size_t)nb * 4);
    NEED(W.plist, (size_t)nb * 4);
    NEED(W.plist2, (size_t)nb * 4);
    NEED(W.klist, (size_t)nb * 4);
    NEED(W.klist_b, (size_t)nb * 4);
    const int snp_ok = snp_shortcut_scores(match, mismatch, gap) ? 1 : 0;
    // K-PAIR: register-bound (the score row of the fill is 65 / 129 registers): 3 / 2 wavefronts per SIMD, the grid loops over its list
    // (scores of sane magnitude only: the fill adds them in ints)
    const bool pair_tier = std::fabs(match) < 1e5 && std::fabs(mismatch) < 1e5 && std::fabs(gap) < 1e5;
    const bool stack_tier = stack_scores(match, mismatch, gap);
    // what K-STACK is given (read per call: tools/ab_pass.py): 1 = bubbles of three and more paths of one length, 2 = also those whose
    // later paths are shorter than the first (one gap run each), 3 = also the two-path bubbles ahead of K-PAIR
    const int stack_level = [] { const char *e = getenv("PF_STACK_LEVEL"); return e ? std::max(1, std::min(3, atoi(e))) : 1; }();
    const int stack_grid = ctx->n_cu * 8;
    if (stack_tier) NEED(W.stack_scr, stack_scratch_bytes() * stack_grid);
    const bool pair_integral = match == std::floor(match) && mismatch == std::floor(mismatch) && gap == std::floor(gap);
    const int pair_grid = ctx->n_cu * 12, pair_grid2 = ctx->n_cu * 4;
    if (pair_tier) NEED(W.pair_scr, PairGeom<PAIR_MAX>::scratch_bytes * pair_grid);
    unsigned long long heads[4] = {0, 0, 0, 0};
    uint64_t n_jobs = 0;
    ta("lists and scratch");
    for (int attempt = 0;; ++attempt) {
        if (attempt > 5) { pf::CtxErr{ctx} = "pf_call_align: pools do not converge"; return PF_ERR_OVERFLOW; }
        const uint64_t path_cap = std::max<uint64_t>(S->path_pool, (uint64_t)nb / 2 + 128ull * paths_grid + 1024);   // (a started piece per wavefront)
        // (first-pass sizes, learnt afterwards; a pool that turns out too small costs a repeated attempt -- at configs[4]'s parameters,
        // k = 31 and insertions to 50 bp, a whole K-BUBBLE run thrown away: 334 B of rows, 2.4 sites, 8 group bytes, 1.1 indel lengths
        // and 55 B of path text per bubble there; 216 B / 1.2 / 2.7 / 0.03 / 17 B at configs[2]'s)
        const uint64_t text_cap = std::max<uint64_t>(S->text_pool, (uint64_t)nb * FIRST_PATH_TEXT + (1u << 16));
        const uint64_t cap_text = std::max<uint64_t>(S->otext_cap, (uint64_t)FIRST_ROW_TEXT * nb + (1u << 16));
        const uint64_t cap_sites = std::max<uint64_t>(S->osites_cap, (uint64_t)FIRST_SITES * nb + 64);
        const uint64_t cap_groups = std::max<uint64_t>(S->ogroups_cap, (uint64_t)FIRST_GROUPS * nb + 64);
        const uint64_t cap_ilen = std::max<uint64_t>(S->oilen_cap, (uint64_t)FIRST_ILEN * nb + 64);
        NEED(W.bpath, ((size_t)4 * nb + path_cap) * sizeof(pf_bubble_path));
        NEED(W.ptext, text_cap);
        const uint64_t walk_cap = S->n_colors ? std::max<uint64_t>(W.walk_cap, (uint64_t)nb * 2 + 256ull * paths_grid + 1024) : 0;
        if (S->n_colors) {
            NEED(W.walk_pool, walk_cap * 4);
            NEED(W.walk_off, (size_t)nb * 8);
        }
        NEED(O.otext, cap_text);
        NEED(O.osites, cap_sites * sizeof(pf_bubble_site));
        NEED(O.ogroups, cap_groups);
        NEED(O.oilen, cap_ilen * 4);
        unsigned long long *d_heads = bubble_pool_heads(ctx, lane);
        if (!d_heads) return PF_ERR_HIP;
        PF_HIP(hipMemsetAsync(d_cnt, 0, sizeof(CallCounters), st));
        PF_HIP(hipMemsetAsync(d_heads, 0, 32, st));
        PrepArgs pa;
        pa.ct = S->ctask.as<CallTask>(); pa.kept = S->kept.as<uint32_t>(); pa.t0 = t0; pa.nb = nb; pa.len = ctx->d_len;
        pa.btask = W.btask.as<pf_bubble_task>(); pa.bpath = W.bpath.as<pf_bubble_path>(); pa.res = O.res.as<pf_bubble_result>();
        pa.lists = CallLists{W.queues.as<uint32_t>(), W.blist.as<uint32_t>(), W.slist.as<uint32_t>(), W.plist.as<uint32_t>(), W.plist2.as<uint32_t>(), W.klist.as<uint32_t>(), W.klist_b.as<uint32_t>(), nb};
        pa.snp_ok = snp_ok;
        pa.pair_ok = pair_tier ? 1 : 0;
        pa.stack_ok = stack_tier ? stack_level : 0;
        pa.cnt = d_cnt;
        tbegin(PF_K_CALL_PREP, st);
        k_call_prep<<<(nb + 255) / 256, 256, 0, st>>>(pa);
        tend(st);
        // K-PATHS (branching bubbles) beside K-SNP and K-PAIR (two-path bubbles): latency-bound walks next to an issue-bound fill
        constexpr bool fork_paths = true;
        hipStream_t pst = st;
        if (fork_paths) {
            if (!W.side_stream) {
                PF_HIP(lane_stream_create(&W.side_stream, lane));
                PF_HIP(hipEventCreateWithFlags(&W.ev_prep, hipEventDisableTiming));
                PF_HIP(hipEventCreateWithFlags(&W.ev_paths, hipEventDisableTiming));
            }
            pst = W.side_stream;
            PF_HIP(hipEventRecord(W.ev_prep, st));
            PF_HIP(hipStreamWaitEvent(pst, W.ev_prep, 0));
        }
        {
            PathArgs ph;
            ph.ct = pa.ct; ph.kept = pa.kept; ph.t0 = t0; ph.nb = nb; ph.blist = pa.lists.blist; ph.succ = ctx->d_succ; ph.seq = ctx->d_seq;
            ph.off = ctx->d_off; ph.len = ctx->d_len; ph.k = k; ph.depth_cap = depth_cap; ph.scratch = W.paths_scr.as<uint8_t>();
            ph.scratch_per_wave = paths_per_wave; ph.force_scratch = paths_force_scratch; ph.btask = pa.btask; ph.bpath = pa.bpath; ph.path_cap = path_cap;
            ph.text = W.ptext.as<char>(); ph.text_cap = text_cap; ph.queues = pa.lists.queues; ph.cnt = d_cnt;
            ph.klist = pa.lists.klist_b; ph.stack_ok = pa.stack_ok;
           
            ph.walk_pool = S->n_colors ? W.walk_pool.as<uint32_t>() : nullptr; ph.walk_off = W.walk_off.as<uint64_t>(); ph.walk_cap = walk_cap;
            ph.max_paths = MAX_PATHS; ph.n_list = &d_cnt->n_branching; ph.mlist = W.mlist.as<uint32_t>(); ph.mlist_cap = W.mlist_cap;
            ph_keep = ph;
            tbegin(PF_K_CALL_PATHS, pst);
            k_call_paths<false><<<paths_grid, 64, 0, pst>>>(ph);
            tend(pst);
            if (fork_paths) PF_HIP(hipEventRecord(W.ev_paths, pst));
        }
        if (snp_ok) {
            SnpArgs sn;
            sn.ct = pa.ct; sn.kept = pa.kept; sn.t0 = t0; sn.nb = nb; sn.slist = pa.lists.slist; sn.seq = ctx->d_seq; sn.off = ctx->d_off;
            sn.len = ctx->d_len; sn.res = pa.res; sn.otext = O.otext.as<char>(); sn.text_cap = cap_text;
            sn.osites = O.osites.as<pf_bubble_site>(); sn.site_cap = cap_sites; sn.ogroups = O.ogroups.as<uint8_t>(); sn.group_cap = cap_groups;
            sn.heads = d_heads; sn.lists = pa.lists; sn.pair_ok = pa.pair_ok; sn.stack_ok = pa.stack_ok; sn.cnt = d_cnt;
            tbegin(PF_K_CALL_SNP, st);
            k_call_snp<<<(nb + 255) / 256, 256, 0, st>>>(sn);   // (the list length is on the device: surplus threads leave at once)
            tend(st);
        }
        StackArgs sk;
        if (stack_tier) {
            // K-STACK, first launch: the strict bubbles K-PREP and K-SNP listed (two paths that are not a single mismatch, three and
            // four paths); what it cannot certify is K-PAIR's (launched behind it) or K-BUBBLE's
            sk.list = pa.lists.klist; sk.n_list = &d_cnt->n_stack; sk.scratch = W.stack_scr.as<uint8_t>();
            sk.btask = pa.btask; sk.bpath = pa.bpath; sk.ptext = W.ptext.as<char>();
            sk.seq = ctx->d_seq; sk.off = ctx->d_off; sk.len = ctx->d_len;
            sk.M = (int)match; sk.D = (int)mismatch; sk.G = (int)gap;
            sk.res = pa.res; sk.otext = O.otext.as<char>(); sk.text_cap = cap_text; sk.osites = O.osites.as<pf_bubble_site>(); sk.site_cap = cap_sites;
            sk.ogroups = O.ogroups.as<uint8_t>(); sk.group_cap = cap_groups; sk.oilen = O.oilen.as<uint32_t>(); sk.ilen_cap = cap_ilen;
            sk.heads = d_heads; sk.lists = pa.lists; sk.cnt = d_cnt;
            sk.pair_ok = pa.pair_ok;
            tbegin(PF_K_CALL_STACK, st);
            k_call_stack<<<stack_grid, 64, 0, st>>>(sk);
            tend(st);
        }
        PairArgs pr;
        static const bool pair_stats = getenv("PF_PAIR_STATS") != nullptr;
        DevTmp<unsigned long long> prof_;
        if (pair_tier) {
            pr.ct = pa.ct; pr.kept = pa.kept; pr.t0 = t0; pr.seq = ctx->d_seq; pr.off = ctx->d_off; pr.len = ctx->d_len;
            pr.M = match; pr.D = mismatch; pr.G = gap;
            pr.Mi = (int)match; pr.Di = (int)mismatch; pr.Gi = (int)gap;
            pr.list = pa.lists.plist; pr.n_list = &d_cnt->n_pair; pr.n_done = &d_cnt->n_pair_done;
            pr.scratch = W.pair_scr.as<uint8_t>(); pr.res = pa.res; pr.otext = O.otext.as<char>(); pr.text_cap = cap_text;
            pr.osites = O.osites.as<pf_bubble_site>(); pr.site_cap = cap_sites; pr.ogroups = O.ogroups.as<uint8_t>(); pr.group_cap = cap_groups;
            pr.oilen = O.oilen.as<uint32_t>(); pr.ilen_cap = cap_ilen; pr.heads = d_heads; pr.lists = pa.lists; pr.cnt = d_cnt;
            pr.prof = nullptr;
            if (pair_stats) {
                PF_HIP(prof_.alloc(64));
                PF_HIP(hipMemsetAsync(prof_.p, 0, 64, st));
                pr.prof = prof_.p;
            }
            tbegin(PF_K_CALL_PAIR, st);
            if (pair_integral) k_call_pair<PAIR_MAX, true><<<pair_grid, 64, 0, st>>>(pr);
            else k_call_pair<PAIR_MAX, false><<<pair_grid, 64, 0, st>>>(pr);
            tend(st);
            if (pair_stats) {
                unsigned long long h[8];
                PF_HIP(hipMemcpy(h, prof_.p, 64, hipMemcpyDeviceToHost));
                fprintf(stderr, "[k_call_pair] %llu wavefront rounds; lane-0 ticks (10 ns): decode %llu fill %llu traceback %llu classify %llu publish %llu\n", h[5],
                        h[0], h[1], h[2], h[3], h[4]);
            }
        }
        if (fork_paths) PF_HIP(hipStreamWaitEvent(st, W.ev_paths, 0));
        if (stack_tier) {
            // K-STACK, second launch: the branching bubbles K-PATHS listed (their two-path rejects cannot go to K-PAIR, which reads
            // the inner unitigs of a strict bubble: K-BUBBLE's)
            sk.list = pa.lists.klist_b; sk.n_list = &d_cnt->n_stack_b; sk.pair_ok = 0;
            tbegin(PF_K_CALL_STACK, st);
            k_call_stack<<<stack_grid, 64, 0, st>>>(sk);
            tend(st);
        }
        PF_HIP(hipGetLastError());
        PF_HIP(hipMemcpyAsync(&hc, d_cnt, sizeof(hc), hipMemcpyDeviceToHost, st));
        PF_HIP(hipStreamSynchronize(st));
        ta("K-PREP .. K-STACK done");
        bool again = false;   // kernels launched once the host knows their lists' lengths: the counters are read once more behind them
        if (pair_tier && hc.n_pair2) {
            // second tier (paths of 65 .. 128 bases, or a longer indel than the first tier's band follows): few on most graphs, so
            // its scratch and its launch wait until the host knows there are any; its rejects join K-BUBBLE's queues
            const int g2 = (int)std::min<uint32_t>((hc.n_pair2 + 63) / 64, (uint32_t)pair_grid2);
            pr.list = pa.lists.plist2; pr.n_list = &d_cnt->n_pair2; pr.n_done = &d_cnt->n_pair2_done;
            pr.prof = nullptr;
            const uint32_t tier2_min = [] { const char *e = getenv("PF_PAIR2_MIN"); return e ? (uint32_t)atoi(e) : 0u; }();   // (read per call: tests run the tier on a few bubbles)
            if (hc.n_pair2 < (tier2_min ? tier2_min : (uint32_t)ctx->n_cu * 32u)) {
                k_call_pair2_reroute<<<(hc.n_pair2 + 255) / 256, 256, 0, st>>>(pr);
            } else {
                NEED(W.pair_scr2, PairGeom<PAIR_MAX2>::scratch_bytes * g2);
                pr.scratch = W.pair_scr2.as<uint8_t>();
                tbegin(PF_K_CALL_PAIR, st);
                if (pair_integral) k_call_pair<PAIR_MAX2, true><<<g2, 64, 0, st>>>(pr);
                else k_call_pair<PAIR_MAX2, false><<<g2, 64, 0, st>>>(pr);
                tend(st);
            }
            again = true;
        }
        if (again) {
            PF_HIP(hipGetLastError());
            PF_HIP(hipMemcpyAsync(&hc, d_cnt, sizeof(hc), hipMemcpyDeviceToHost, st));
            PF_HIP(hipStreamSynchronize(st));
        }
        ctx_units(ctx, PF_K_CALL_PREP, nb);
        if (snp_ok) ctx_units(ctx, PF_K_CALL_SNP, hc.n_snp);
        if (pair_tier) ctx_units(ctx, PF_K_CALL_PAIR, hc.n_pair + hc.n_pair2);
        if (stack_tier) ctx_units(ctx, PF_K_CALL_STACK, hc.n_stack + hc.n_stack_b);
        ctx_units(ctx, PF_K_CALL_PATHS, hc.n_branching);
        if (getenv("PF_TRACE_ALIGN")) fprintf(stderr, "[pf_call_align] bubbles of more than 255 walks: %u, err %u\n", hc.n_many, hc.err);
        if (hc.n_many > W.mlist_cap && !(hc.err & 33u)) {
            W.mlist_cap = hc.n_many + hc.n_many / 8 + 64;
            NEED(W.mlist, (size_t)W.mlist_cap * 4);
            continue;
        }
        if (hc.n_many && !(hc.err & 33u)) {
            // bubbles of more than 255 walks: walked again by a few wavefronts with room for PATHS_BIG walks each
            const uint32_t n_many = hc.n_many;
            const int big_grid = (int)std::min<uint32_t>(n_many, 32);
            const uint64_t big_per_wave = (((((uint64_t)10 * depth_cap + 4) * 4 + 7) & ~7ull) + ((uint64_t)PATHS_BIG + 1) * 12 + 255) & ~255ull;
            NEED(W.paths_big_scr, big_per_wave * big_grid);
            PathArgs pb = ph_keep;
            pb.blist = W.mlist.as<uint32_t>(); pb.n_list = &d_cnt->n_many; pb.max_paths = PATHS_BIG; pb.mlist = nullptr; pb.mlist_cap = 0;
            pb.scratch = W.paths_big_scr.as<uint8_t>(); pb.scratch_per_wave = big_per_wave;
            tbegin(PF_K_CALL_PATHS, st);
            k_call_paths<true><<<big_grid, 64, 0, st>>>(pb);
            tend(st);
            PF_HIP(hipGetLastError());
            PF_HIP(hipMemcpyAsync(&hc, d_cnt, sizeof(hc), hipMemcpyDeviceToHost, st));
            PF_HIP(hipStreamSynchronize(st));
        }
        if (hc.err & 33u) {
            char where[96];
            snprintf(where, sizeof where, " (superbubble from unitig %u%c to unitig %u%c)", (hc.err_entrance >> 1) + 1, (hc.err_entrance & 1) ? '-' : '+',
                     (hc.err_exit >> 1) + 1, (hc.err_exit & 1) ? '-' : '+');
            pf::CtxErr{ctx} = std::string(hc.err & 1u ? "pf_call_run: a bubble has more than 65535 paths" : "pf_call_run: a bubble is deeper than the complex size allows") + where;
            return PF_ERR_ARG;
        }
        amax(S->otext_cap, cap_text); amax(S->osites_cap, cap_sites);
        amax(S->ogroups_cap, cap_groups); amax(S->oilen_cap, cap_ilen);
        amax(S->path_pool, path_cap);
        amax(S->text_pool, text_cap);
        static const bool trace_retry = getenv("PF_TRACE_ALIGN") != nullptr;
        if (trace_retry)
            fprintf(stderr, "[pf_call_align] attempt %d: %u bubbles, path pool %llu of %llu, path text %llu of %llu, rows text cap %llu (needs %llu)\n", attempt, nb,
                    (unsigned long long)hc.path_head, (unsigned long long)path_cap, (unsigned long long)hc.text_head, (unsigned long long)text_cap,
                    (unsigned long long)cap_text, (unsigned long long)(3 * hc.text_head + 128ull * hc.n_branching + 160ull * nb));
        if (hc.path_head > path_cap || hc.text_head > text_cap || hc.walk_head > walk_cap) {
            amax(S->path_pool, hc.path_head + hc.path_head / 8 + 1024);
            amax(S->text_pool, hc.text_head + hc.text_head / 8 + 4096);
            if (S->n_colors) W.walk_cap = std::max<uint64_t>(W.walk_cap, hc.walk_head + hc.walk_head / 8 + 1024);
            continue;
        }
        if (S->n_colors) W.walk_cap = std::max(W.walk_cap, walk_cap);
        // the aligned rows of the branching bubbles come on top of what K-SNP took: make room before K-BUBBLE runs
        // (the path text is handed out in per-wavefront chunks: its size varies by a few per mille from pass to pass, hence the margin)
        if (cap_text < 3 * hc.text_head + 128ull * hc.n_branching + 160ull * nb) {
            const uint64_t need = 3 * hc.text_head + 128ull * hc.n_branching + 160ull * nb;
            amax(S->otext_cap, need + need / 16 + 4096);
            continue;
        }
        n_jobs = 0;
        for (int x = 0; x < NQ; ++x) n_jobs += hc.q_n[x];
        // K-BUBBLE: the queues are heavy-then-light per class; compact them into one index array
        NEED(W.scan_tmp2, std::max<size_t>((size_t)n_jobs, 1) * 4);
        {
            uint32_t *d_idx = W.scan_tmp2.as<uint32_t>();
            size_t at = 0;
            for (int x = 0; x < NQ; ++x) {
                if (!hc.q_n[x]) continue;
                PF_HIP(hipMemcpyAsync(d_idx + at, W.queues.as<uint32_t>() + (size_t)x * nb, (size_t)hc.q_n[x] * 4, hipMemcpyDeviceToDevice, st));
                at += hc.q_n[x];
            }
        }
        BubbleLaunch BL;
        BL.text = W.ptext.as<char>(); BL.paths = W.bpath.as<pf_bubble_path>(); BL.tasks = W.btask.as<pf_bubble_task>();
        BL.n_tasks = nb; BL.idx = W.scan_tmp2.as<uint32_t>();
        for (int c = 0; c <= kBubLdsClasses; ++c) BL.n_cls[c] = hc.q_n[2 * c] + hc.q_n[2 * c + 1];
        BL.max_need = hc.max_need; BL.retry_need = hc.retry_need;
        BL.match = match; BL.mismatch = mismatch; BL.gap = gap;
        BL.res = O.res.as<pf_bubble_result>(); BL.otext = O.otext.as<char>(); BL.osites = O.osites.as<pf_bubble_site>();
        BL.ogroups = O.ogroups.as<uint8_t>(); BL.oilen = O.oilen.as<uint32_t>();
        BL.text_cap = cap_text; BL.site_cap = cap_sites; BL.group_cap = cap_groups; BL.ilen_cap = cap_ilen;
        BL.keep_heads = true;
        BL.lane = lane; BL.stream = st;
        ta("K-BUBBLE launching");
        const int bst = bubble_launch(ctx, BL, heads);
        ta("K-BUBBLE done");
        if (trace_retry)
            fprintf(stderr, "[pf_call_align] attempt %d: K-BUBBLE status %d, pools text %llu of %llu, sites %llu of %llu, groups %llu of %llu, indel lengths %llu of %llu\n", attempt, bst,
                    heads[0], (unsigned long long)cap_text, heads[1], (unsigned long long)cap_sites, heads[2], (unsigned long long)cap_groups, heads[3], (unsigned long long)cap_ilen);
        if (bst == PF_ERR_OVERFLOW && (heads[0] > cap_text || heads[1] > cap_sites || heads[2] > cap_groups || heads[3] > cap_ilen)) {
            amax(S->otext_cap, heads[0] + heads[0] / 8);
            amax(S->osites_cap, heads[1] + heads[1] / 8);
            amax(S->ogroups_cap, heads[2] + heads[2] / 8);
            amax(S->oilen_cap, heads[3] + heads[3] / 8);
            continue;
        }
        if (bst != PF_OK) return bst;
        break;
    }
    out->n_branching = hc.n_branching;
    out->align_jobs = n_jobs + hc.n_snp_done + hc.n_pair_done + hc.n_pair2_done + hc.n_stack_done;
    out->snp_jobs = hc.n_snp_done; out->pair_jobs = hc.n_pair_done + hc.n_pair2_done; out->wave_jobs = n_jobs; out->stack_jobs = hc.n_stack_done;

    // ---- bubble numbering inside the batch (launched ahead of K-SITES, read with its counters: one wait for both) ----
    k_call_has<<<(nb + 255) / 256, 256, 0, st>>>(O.res.as<pf_bubble_result>(), nb, W.has.as<uint32_t>());
    NEED(W.scan_tmp, scan_scratch_bytes(nb));
    PF_HIP(scan_inclusive_u32(W.has.as<uint32_t>(), O.vc.as<uint32_t>(), nb, W.scan_tmp.p, st));
    uint32_t n_called = 0;
    PF_HIP(hipMemcpyAsync(&n_called, O.vc.as<uint32_t>() + (nb - 1), 4, hipMemcpyDeviceToHost, st));

    // ---- K-SITES ----
    // (a site string is k characters long unless it takes the raw columns up to its row's end -- substr with a negative count,
    // src/CDBG.cpp:1499 -- or more than k characters agree behind an indel: a launch that meets one longer than its room says how
    // long, ks_need, and is repeated with that much)
    uint32_t KS = std::max<uint32_t>((uint32_t)(2 * k + 64), S->sites_ks.load());
    if (hc.n_branching) {
        const uint32_t C = S->n_colors;
        const uint64_t rows_cap = std::max<uint64_t>(256, ((uint64_t)hc.max_rows + 63) & ~63ull);
        constexpr int sites_per_cu = 16;
        for (int attempt = 0, ks_attempt = 0;; ++attempt) {
            const uint64_t sites_per_wave = ((2 * rows_cap * KS + rows_cap * (4 + 4 + 4 + 1 + 1 + 8) + (C ? rows_cap * (16ull * S->col_words + 8ull * C + 1) : 0)) + 255) & ~255ull;
            // (tables for thousands of rows: fewer wavefronts, at most 2 GB of them)
            const int sites_grid = (int)std::max<uint64_t>(1, std::min<uint64_t>(std::min<uint32_t>(hc.n_branching, (uint32_t)(ctx->n_cu * sites_per_cu)), (2ull << 30) / sites_per_wave));
            NEED(W.sites_scr, sites_per_wave * sites_grid);
            const uint64_t sv_cap = std::max<uint64_t>(S->sv_pool, 8ull * std::max<uint32_t>(C, 1) * hc.n_branching + 1024ull * sites_grid + 1024);   // (a started chunk per wavefront)
            NEED(O.sv, sv_cap * 8);
            SiteArgs sa;
            sa.ct = S->ctask.as<CallTask>(); sa.kept = S->kept.as<uint32_t>(); sa.t0 = t0; sa.blist = W.blist.as<uint32_t>();
            sa.res = O.res.as<pf_bubble_result>(); sa.otext = O.otext.as<char>(); sa.osites = O.osites.as<pf_bubble_site>();
            sa.ogroups = O.ogroups.as<uint8_t>(); sa.k = k; sa.tab = ctx->d_tab; sa.mask = ctx->tab_cap - 1;
            sa.one_strand = ctx->tab_one_strand; sa.tab_exact = ctx->tab_exact; sa.low = S->low; sa.up = S->up; sa.ks = KS; sa.rows_cap = (uint32_t)rows_cap;
            sa.scratch = W.sites_scr.as<uint8_t>(); sa.scratch_per_wave = sites_per_wave; sa.sv_off = O.sv_off.as<uint64_t>();
            sa.sv = O.sv.as<double>(); sa.sv_cap = sv_cap; sa.cnt = d_cnt;
            sa.n_colors = C;
            sa.ctab = CTab{ctx->d_ctab, ctx->ctab_cap - 1, ctx->ctab_line_bytes}; sa.c_one_strand = ctx->ctab_one_strand; sa.unread = ctx->d_unread;
            sa.clow = S->col_low.as<uint32_t>(); sa.cup = S->col_up.as<uint32_t>(); sa.full = S->col_full.as<uint64_t>(); sa.cwords = S->col_words;
            sa.part_first = S->part_first.as<uint32_t>(); sa.part_colour = S->part_colour.as<uint32_t>(); sa.part_word = S->part_word.as<uint64_t>();
            sa.part_bits = S->part_bits.as<uint64_t>(); sa.walk_pool = W.walk_pool.as<uint32_t>(); sa.walk_off = W.walk_off.as<uint64_t>();
            sa.seq = ctx->d_seq; sa.off = ctx->d_off; sa.len = ctx->d_len;
            static const bool sites_stats = getenv("PF_SITES_STATS") != nullptr;   // measurements: where a wavefront's time goes
            DevTmp<unsigned long long> sprof_;
            sa.prof = nullptr;
            if (sites_stats) {
                PF_HIP(sprof_.alloc((size_t)sites_grid * 48));
                PF_HIP(hipMemsetAsync(sprof_.p, 0, (size_t)sites_grid * 48, st));
                sa.prof = sprof_.p;
            }
            PF_HIP(hipMemsetAsync(&d_cnt->sites_next, 0, 4, st));
            PF_HIP(hipMemsetAsync(&d_cnt->sv_head, 0, 8, st));
            PF_HIP(hipMemsetAsync(&d_cnt->site_strings, 0, 8, st));
            tbegin(PF_K_CALL_SITES, st);
            if (C) k_call_sites<true><<<sites_grid, 64, 0, st>>>(sa);
            else k_call_sites<false><<<sites_grid, 64, 0, st>>>(sa);
            tend(st);
            ctx_units(ctx, PF_K_CALL_SITES, hc.n_branching);
            PF_HIP(hipGetLastError());
            if (sa.prof) {
                PF_HIP(hipStreamSynchronize(st));
                std::vector<unsigned long long> h((size_t)sites_grid * 6);
                PF_HIP(hipMemcpy(h.data(), sprof_.p, h.size() * 8, hipMemcpyDeviceToHost));
                unsigned long long sum[6] = {0, 0, 0, 0, 0, 0}, mx = 0;
                for (int w = 0; w < sites_grid; ++w) {
                    for (int x = 0; x < 6; ++x) sum[x] += h[(size_t)w * 6 + x];
                    mx = std::max(mx, h[(size_t)w * 6]);
                }
                fprintf(stderr, "[k_call_sites] %d wavefronts, %llu bubbles; ticks (10 ns) per wavefront: total %.0f (max %llu) = pop + load %.0f, strings %.0f, ranks + probes %.0f, groups %.0f\n",
                        sites_grid, sum[5], (double)sum[0] / sites_grid, mx, (double)sum[1] / sites_grid, (double)sum[2] / sites_grid, (double)sum[3] / sites_grid,
                        (double)sum[4] / sites_grid);
            }
            PF_HIP(hipMemcpyAsync(&hc, d_cnt, sizeof(hc), hipMemcpyDeviceToHost, st));
            PF_HIP(hipStreamSynchronize(st));
            if (hc.err & 16u) {
                if (++ks_attempt > 3 || hc.ks_need <= KS) { pf::CtxErr{ctx} = "pf_call_run: the room for a site string does not converge"; return PF_ERR_OVERFLOW; }
                KS = (hc.ks_need + 63u) & ~63u;
                amax(S->sites_ks, KS);
                hc.err = 0; hc.ks_need = 0;   // (the other bits are looked at when every string had room: a string cut short has no verdict)
                PF_HIP(hipMemsetAsync(&d_cnt->err, 0, 4, st));
                PF_HIP(hipMemsetAsync(&d_cnt->ks_need, 0, 4, st));
                --attempt;
                continue;
            }
            if (hc.err & 2u) { pf::CtxErr{ctx} = "CDBG::readCov(): a kmer of a site string can not found ."; return PF_ERR_MISSING_KMER; }
            if (hc.err & 4u) { pf::CtxErr{ctx} = "CDBG::PloidyEstimation(): a site string runs past the end of an aligned row (the reference terminates here: std::out_of_range from substr, src/CDBG.cpp:1478-1590)"; return PF_ERR_ARG; }
            if (hc.err & 64u) { pf::CtxErr{ctx} = "CCDBG::PloidyEstimation(): a site string does not start on a unitig of its bubble"; return PF_ERR_ARG; }
            if (hc.sv_head > sv_cap) {
                if (attempt >= 2) { pf::CtxErr{ctx} = "pf_call_run: site value pool does not converge"; return PF_ERR_OVERFLOW; }
                amax(S->sv_pool, hc.sv_head + hc.sv_head / 8 + 1024);
                continue;
            }
            amax(S->sv_pool, sv_cap);
            break;
        }
    } else {
        NEED(O.sv, 16);
        PF_HIP(hipStreamSynchronize(st));   // (n_called)
    }
    out->site_strings = hc.site_strings;
    ta("K-SITES done");

    out->n_called = n_called;
    O.nb = nb;
    O.cur = *out;
    for (int x = 0; x < 4; ++x) O.used[x] = heads[x];
    O.used[4] = hc.n_branching ? hc.sv_head : 0;
#undef NEED
    return PF_OK;
}

int pf_call_peek(pf_ctx *ctx, int lane, pf_call_bubble *bubbles, pf_bubble_result *results, uint64_t *sv_off, uint64_t bubble_cap, char *text,
                 pf_bubble_site *sites, uint8_t *groups, uint32_t *ilen, double *sv, const uint64_t cap[5], uint64_t used[6]) {
    if (!ctx || lane < 0 || lane >= PF_CALL_LANES || !used) return PF_ERR_ARG;
    CallState *S = ctx->call;
    if (!S) { pf::CtxErr{ctx} = "pf_call_peek: pf_call_align first"; return PF_ERR_ARG; }
    if (S->n_colors) { pf::CtxErr{ctx} = "pf_call_peek: the single-sample path's view (a colored site holds one value per colour and group)"; return PF_ERR_ARG; }
    const CallState::AlignOut &O = S->lane[lane];
    used[0] = O.nb;
    for (int x = 0; x < 5; ++x) used[x + 1] = O.used[x];
    if (!O.nb) return PF_OK;
    if (!bubbles && !results && !text && !sites && !groups && !ilen && !sv && !sv_off) return PF_OK;   // sizes only
    if (bubble_cap < O.nb || !cap) return PF_ERR_ARG;
    for (int x = 0; x < 5; ++x)
        if (cap[x] < O.used[x]) return PF_ERR_ARG;
    PF_HIP(hipSetDevice(ctx->device));
    if (bubbles) {
        std::vector<uint32_t> kept(O.nb);
        PF_HIP(hipMemcpy(kept.data(), S->kept.as<uint32_t>() + O.t0, (size_t)O.nb * 4, hipMemcpyDeviceToHost));
        std::vector<CallTask> ct(S->n_sides);
        if (S->n_sides) PF_HIP(hipMemcpy(ct.data(), S->ctask.p, (size_t)S->n_sides * sizeof(CallTask), hipMemcpyDeviceToHost));
        for (uint32_t j = 0; j < O.nb; ++j) {
            const CallTask &t = ct[kept[j]];
            pf_call_bubble &b = bubbles[j];
            b.entrance_ov = t.entrance_ov; b.exit_ov = t.exit_ov; b.strict = t.strict; b.n_inner = t.n_inner;
            for (int x = 0; x < 4; ++x) { b.inner[x] = t.inner[x]; b.cov[x] = t.cov[x]; }
            b.core_mean = t.core_mean; b.cov_sum = t.cov_sum;
        }
    }
    if (results) PF_HIP(hipMemcpy(results, O.res.p, (size_t)O.nb * sizeof(pf_bubble_result), hipMemcpyDeviceToHost));
    if (sv_off) PF_HIP(hipMemcpy(sv_off, O.sv_off.p, (size_t)O.nb * 8, hipMemcpyDeviceToHost));
    if (text && O.used[0]) PF_HIP(hipMemcpy(text, O.otext.p, O.used[0], hipMemcpyDeviceToHost));
    if (sites && O.used[1]) PF_HIP(hipMemcpy(sites, O.osites.p, O.used[1] * sizeof(pf_bubble_site), hipMemcpyDeviceToHost));
    if (groups && O.used[2]) PF_HIP(hipMemcpy(groups, O.ogroups.p, O.used[2], hipMemcpyDeviceToHost));
    if (ilen && O.used[3]) PF_HIP(hipMemcpy(ilen, O.oilen.p, O.used[3] * 4, hipMemcpyDeviceToHost));
    if (sv && O.used[4]) PF_HIP(hipMemcpy(sv, O.sv.p, O.used[4] * 8, hipMemcpyDeviceToHost));
    return PF_OK;
}

// The buffers pf_call_align_lane asks for on its first call for ranges of up to nb bubbles, taken NOW (a caller does this beside the
// load): a first pass then starts with its pools in place instead of two dozen hipMallocs, 20 ms at 5 M unitigs.  Sizes are the
// first-call formulas of pf_call_align_lane; whatever turns out too small there grows as before.
int pf_call_reserve_lanes(pf_ctx *ctx, uint64_t nb64, uint32_t complex_size, int n_lanes) {
    if (!ctx || nb64 == 0 || n_lanes < 1 || n_lanes > PF_CALL_LANES) return PF_ERR_ARG;
    CallState *S = state_of(ctx);
    PF_HIP(hipSetDevice(ctx->device));
    const uint32_t nb = (uint32_t)std::min<uint64_t>(nb64, 1u << 24);
    const char *oom = "pf_call_reserve: out of device memory";
#define NEED(buf, bytes) do { if (!(buf).ensure(bytes)) { pf::CtxErr{ctx} = oom; return PF_ERR_HIP; } } while (0)
    DevLoadTrace trace;
    for (int lane = 0; lane < n_lanes; ++lane) {
        CallState::AlignWork &W = S->work[lane];
        NEED(W.counters, sizeof(CallCounters));
        NEED(W.btask, (size_t)nb * sizeof(pf_bubble_task));
        NEED(W.queues, (size_t)NQ * nb * 4);
        for (DevBuf *b : {&W.blist, &W.slist, &W.plist, &W.plist2, &W.klist, &W.klist_b, &W.has}) NEED(*b, (size_t)nb * 4);
        NEED(W.stack_scr, stack_scratch_bytes() * (uint64_t)(ctx->n_cu * 8));
        const uint32_t depth_cap = std::max<uint32_t>(complex_size + 4, 16);
        const uint64_t paths_per_wave = ((256 * 8 + 256 * 4 + (10ull * depth_cap + 4) * 4) + 255) & ~255ull;
        NEED(W.paths_scr, paths_per_wave * (uint64_t)(ctx->n_cu * 16));
        NEED(W.pair_scr, PairGeom<PAIR_MAX>::scratch_bytes * (uint64_t)(ctx->n_cu * 12));
        NEED(W.bpath, ((size_t)4 * nb + (uint64_t)nb / 2 + 128ull * (ctx->n_cu * 16) + 1024) * sizeof(pf_bubble_path));
        NEED(W.ptext, (uint64_t)nb * FIRST_PATH_TEXT + (1u << 16));
        NEED(W.scan_tmp2, (size_t)nb / 4 * 4 + 4096);
        {   // K-SITES' tables for bubbles of up to 256 walks (single-sample; for the longest k: the graph may still be on its way)
            const uint64_t KS = (uint64_t)(2 * 31 + 64), rows_cap = 256;
            const uint64_t sites_per_wave = ((2 * rows_cap * KS + rows_cap * (4 + 4 + 4 + 1 + 1 + 8)) + 255) & ~255ull;
            NEED(W.sites_scr, sites_per_wave * (uint64_t)(ctx->n_cu * 16));
        }
        CallState::AlignOut &O = S->lane[lane];
        NEED(O.res, (size_t)nb * sizeof(pf_bubble_result));
        NEED(O.sv_off, (size_t)nb * 8);
        NEED(O.vc, (size_t)nb * 4);
        NEED(O.otext, (uint64_t)FIRST_ROW_TEXT * nb + (1u << 16));
        NEED(O.osites, ((uint64_t)FIRST_SITES * nb + 64) * sizeof(pf_bubble_site));
        NEED(O.ogroups, (uint64_t)FIRST_GROUPS * nb + 64);
        NEED(O.oilen, ((uint64_t)FIRST_ILEN * nb + 64) * 4);
        NEED(O.sv, ((uint64_t)nb / 4 + 1024ull * ctx->n_cu * 16 + 1024) * 8);
        trace.mark("reserve: a lane's buffers");
        if (lane != 0 && !W.stream) { PF_HIP(lane_stream_create(&W.stream, lane)); W.own_stream = true; }
        if (!W.side_stream) {
            PF_HIP(lane_stream_create(&W.side_stream, lane));
            PF_HIP(hipEventCreateWithFlags(&W.ev_prep, hipEventDisableTiming));
            PF_HIP(hipEventCreateWithFlags(&W.ev_paths, hipEventDisableTiming));
        }
        trace.mark("reserve: a lane's streams and events");
        const int st = bubble_reserve(ctx, nb, lane);
        if (st != PF_OK) return st;
        trace.mark("reserve: K-BUBBLE's workspaces and first launches");
    }
#undef NEED
    return PF_OK;
}

int pf_call_reserve(pf_ctx *ctx, uint64_t nb64, uint32_t complex_size) { return pf_call_reserve_lanes(ctx, nb64, complex_size, 2); }

int pf_call_align(pf_ctx *ctx, uint64_t t0, uint64_t t1, uint32_t complex_size, double match, double mismatch, double gap,
                  pf_call_result *out) {
    return pf_call_align_lane(ctx, 0, t0, t1, complex_size, match, mismatch, gap, out);
}

// one batch, second half: K-TEXT of the bubbles pf_call_align left resident, into slab 0 or 1
// (any host thread: a stream, scratch and counters of its own, launch timing by place -- one pf_call_text_range at a time,
// beside at most one pf_call_align_lane on the OTHER lane)
static int text_work_of(pf_ctx *ctx, pf::CallState *S, int which, uint32_t nb) {
    pf::CallState::TextWork &T = S->text[which];
    if (!T.stream) {
        // highest priority: its short kernels go ahead of the alignment kernels of the other lanes, whose grids fill the device
        // for milliseconds -- the text has a PCIe copy and a file copy still before it
        int least = 0, greatest = 0;
        PF_HIP(hipDeviceGetStreamPriorityRange(&least, &greatest));
        PF_HIP(hipStreamCreateWithPriority(&T.stream, hipStreamNonBlocking, greatest));
    }
    NEED_TEXT(T.sizes, (size_t)N_INT * (nb + 1) * 4);
    NEED_TEXT(T.offs, ((size_t)N_INT * (nb + 1) + 1) * 8);
    NEED_TEXT(T.totals, 16 * 8);
    NEED_TEXT(T.tcounters, sizeof(CallCounters));
    NEED_TEXT(T.tscan, scan_scratch_bytes((uint64_t)N_INT * (nb + 1)));
    return PF_OK;
}

// what the first pf_call_text_range of a run would take (see pf_call_reserve): the text streams, the size tables of a piece of
// piece_bubbles bubbles and the text slabs (an estimate from k: rows of two to four aligned paths; grows as before when short)
int pf_call_reserve_text(pf_ctx *ctx, uint64_t piece_bubbles) {
    if (!ctx || piece_bubbles == 0) return PF_ERR_ARG;
    pf::CallState *S = state_of(ctx);
    PF_HIP(hipSetDevice(ctx->device));
    const uint32_t nb = (uint32_t)std::min<uint64_t>(piece_bubbles, 1u << 24);
    for (int which = 0; which < 2; ++which) { const int ts = text_work_of(ctx, S, which, nb); if (ts != PF_OK) return ts; }
    for (int slab = 0; slab < PF_CALL_SLABS; ++slab) NEED_TEXT(S->out[slab], (11ull * 31 + 40) * nb);   // (the longest k: the graph may still be on its way)
    // K-TEXT's two passes once over no bubbles on each of its streams: what the first launch of a kernel this size pays on a stream
    // (scratch for its spills: 2.3 ms of a first piece's count pass) is paid here, beside the load
    for (int which = 0; which < 2; ++which) {
        pf::CallState::TextWork &T = S->text[which];
        FmtArgs fa;
        memset(&fa, 0, sizeof(fa));
        fa.sizes = T.sizes.as<uint32_t>(); fa.offs = T.offs.as<uint64_t>(); fa.cnt = T.tcounters.as<CallCounters>();
        if (S->n_colors) { k_call_format<false, true><<<1, FMT_BLOCK, 0, T.stream>>>(fa); k_call_format<true, true><<<1, FMT_BLOCK, 0, T.stream>>>(fa); }
        else { k_call_format<false, false><<<1, FMT_BLOCK, 0, T.stream>>>(fa); k_call_format<true, false><<<1, FMT_BLOCK, 0, T.stream>>>(fa); }
        PF_HIP(hipGetLastError());
        PF_HIP(hipStreamSynchronize(T.stream));
    }
    return PF_OK;
}

static int call_text_impl(pf_ctx *ctx, int lane, int slab, uint64_t first, uint64_t count, uint64_t var_count_base, pf_call_result *out, bool sizes_only) {
    if (!ctx || !out || slab < 0 || slab >= PF_CALL_SLABS || lane < 0 || lane >= PF_CALL_LANES) return PF_ERR_ARG;
    CallState *S = ctx->call;
    if (!S) return PF_ERR_ARG;
    CallState::AlignOut &O = S->lane[lane];
    if (first + count > O.nb) { pf::CtxErr{ctx} = "pf_call_text_range: range outside the aligned batch"; return PF_ERR_ARG; }
    if (S->n_colors && S->mt_format) { pf::CtxErr{ctx} = "pf_call_text_range: the -t > 1 format is the single-sample path's"; return PF_ERR_ARG; }
    if (!sizes_only)
        for (int s = 0; s < N_STREAMS; ++s) S->out_len[slab][s] = 0;
    *out = O.cur;
    out->n_called = 0;
    if (count == 0) return PF_OK;
    PF_HIP(hipSetDevice(ctx->device));
    const bool trace_stages = getenv("PF_TRACE_ALIGN") != nullptr;
    const auto t_enter = std::chrono::steady_clock::now();
    auto ta = [&](const char *what) {
        if (trace_stages) fprintf(stderr, "[pf_call_text]    %-34s %.2f ms\n", what, std::chrono::duration<double>(std::chrono::steady_clock::now() - t_enter).count() * 1e3);
    };
    const uint32_t nb = (uint32_t)count;
    const int which = sizes_only ? 0 : (slab & 1);
    { const int ts = text_work_of(ctx, S, which, nb); if (ts != PF_OK) return ts; }
    CallState::TextWork &T = S->text[which];
    hipStream_t st = T.stream;
    const uint64_t t0 = O.t0;
    // bubbles called inside the range: difference of the batch-wide running count (read with the sizes below: one wait)
    uint32_t vc_edge[2] = {0, 0};
    PF_HIP(hipMemcpyAsync(&vc_edge[1], O.vc.as<uint32_t>() + (first + count - 1), 4, hipMemcpyDeviceToHost, st));
    if (first) PF_HIP(hipMemcpyAsync(&vc_edge[0], O.vc.as<uint32_t>() + (first - 1), 4, hipMemcpyDeviceToHost, st));
    const char *oom = "pf_call_text: out of device memory";
#define NEED(buf, bytes) do { if (!(buf).ensure(bytes)) { pf::CtxErr{ctx} = oom; return PF_ERR_HIP; } } while (0)
    CallCounters *d_cnt = T.tcounters.as<CallCounters>();
    CallCounters hc;
    const size_t n_sizes = (size_t)N_INT * (nb + 1);
    ta("stream and size tables");
    PF_HIP(hipMemsetAsync(&d_cnt->allele[0], 0, 6 * 8, st));  // allele[4], core_cov, core_num
    FmtArgs fa;
    fa.ct = S->ctask.as<CallTask>(); fa.kept = S->kept.as<uint32_t>(); fa.t0 = t0; fa.j0 = (uint32_t)first; fa.nb = nb;
    fa.res = O.res.as<pf_bubble_result>();
    fa.otext = O.otext.as<char>(); fa.osites = O.osites.as<pf_bubble_site>(); fa.ogroups = O.ogroups.as<uint8_t>();
    fa.oilen = O.oilen.as<uint32_t>(); fa.sv_off = O.sv_off.as<uint64_t>(); fa.sv = O.sv.as<double>(); fa.vc = O.vc.as<uint32_t>();
    fa.vc_base = var_count_base; fa.mt = S->mt_format ? 1 : 0; fa.len = ctx->d_len; fa.sizes = T.sizes.as<uint32_t>(); fa.offs = T.offs.as<uint64_t>(); fa.cnt = d_cnt;
    for (int s = 0; s < N_INT; ++s) fa.out[s] = nullptr;
    fa.packed = S->pack_alignseq ? 1 : 0;
    size_t at = 0;
    ctx_begin_at(ctx, PF_K_CALL_FORMAT, st, &at);
    fa.n_colors = S->n_colors; fa.N = ctx->N; fa.k = ctx->k; fa.full = S->col_full.as<uint64_t>(); fa.cwords = S->col_words; fa.ccov_sum = S->ccov_sum.as<uint64_t>();
    if (S->n_colors) k_call_format<false, true><<<(nb + 1 + FMT_BLOCK - 1) / FMT_BLOCK, FMT_BLOCK, 0, st>>>(fa);
    else k_call_format<false, false><<<(nb + 1 + FMT_BLOCK - 1) / FMT_BLOCK, FMT_BLOCK, 0, st>>>(fa);
    ctx_end_at(ctx, at, st);
    PF_HIP(scan_exclusive_u32_u64(T.sizes.as<uint32_t>(), T.offs.as<uint64_t>(), n_sizes, T.tscan.p, st));
    k_call_totals<<<1, 64, 0, st>>>(T.offs.as<uint64_t>(), T.sizes.as<uint32_t>(), nb, T.totals.as<uint64_t>());
    uint64_t totals[N_INT + 1] = {};
    PF_HIP(hipMemcpyAsync(totals, T.totals.p, N_INT * 8, hipMemcpyDeviceToHost, st));
    PF_HIP(hipMemcpyAsync(&hc, d_cnt, sizeof(hc), hipMemcpyDeviceToHost, st));
    PF_HIP(hipStreamSynchronize(st));
    out->n_called = vc_edge[1] - vc_edge[0];
    // what a stream takes in the slab: its text -- alignseq, when packed, its index and records instead
    const uint64_t packed_len = (fa.packed && totals[PF_OUT_ALIGNSEQ]) ? alnpack_index_bytes(nb) + totals[S_PACK] : 0;
    auto slab_len = [&](int s) { return (s == PF_OUT_ALIGNSEQ && fa.packed) ? packed_len : totals[s]; };
    out->alignseq_packed_len = packed_len;
    uint64_t all = 0;
    for (int s = 0; s < N_STREAMS; ++s) all += slab_len(s);
    if (sizes_only) {   // pf_call_text_sizes: the count pass alone
        for (int s = 0; s < N_STREAMS; ++s) out->text_len[s] = totals[s];
        for (int x = 0; x < 4; ++x) out->allele[x] = hc.allele[x];
        out->core_cov = hc.core_cov;
        out->core_num = hc.core_num;
        return PF_OK;
    }
    ta("sizes counted");
    NEED(S->out[slab], std::max<uint64_t>(all, 16));
    ta("slab taken");
    all = 0;
    for (int s = 0; s < N_STREAMS; ++s) {
        fa.out[s] = S->out[slab].as<char>() + all;
        S->out_off[slab][s] = all;
        S->out_len[slab][s] = slab_len(s);
        out->text_len[s] = totals[s];
        all += slab_len(s);
    }
    if (fa.packed) { fa.out[S_PACK] = fa.out[PF_OUT_ALIGNSEQ]; fa.out[PF_OUT_ALIGNSEQ] = nullptr; }
    ctx_begin_at(ctx, PF_K_CALL_FORMAT, st, &at);
    if (S->n_colors) k_call_format<true, true><<<(nb + FMT_BLOCK - 1) / FMT_BLOCK, FMT_BLOCK, 0, st>>>(fa);
    else k_call_format<true, false><<<(nb + FMT_BLOCK - 1) / FMT_BLOCK, FMT_BLOCK, 0, st>>>(fa);
    ctx_end_at(ctx, at, st);
    ctx_units(ctx, PF_K_CALL_FORMAT, nb);
    PF_HIP(hipGetLastError());
    S->fetch_base[slab] = S->out[slab].as<char>();
    S->nib[slab] = false;
    out->numeric_packed = 0;
    for (int s = 0; s < N_STREAMS; ++s) { S->txt_off[slab][s] = S->out_off[slab][s]; S->txt_len[slab][s] = S->out_len[slab][s]; }
    if (S->pack_numeric) {
        // K-NIB behind the write pass: what the fetches copy is a second buffer -- the numeric streams at four bits a character,
        // alignseq as it lies in the slab, every region a multiple of 16 bytes, and a 16-byte tail whose first word says which
        // streams held a character outside the sixteen (those are fetched as text: pf_call_fetch_text)
        uint64_t foff[N_STREAMS], flen[N_STREAMS], fall = 0;
        for (int s = 0; s < N_STREAMS; ++s) {
            flen[s] = s == PF_OUT_ALIGNSEQ ? ((slab_len(s) + 15) & ~15ull) : PF_NUMERIC_PACKED_LEN(totals[s]);
            foff[s] = fall;
            fall += flen[s];
        }
        NEED(S->outp[slab], fall + 16);
        char *pb = S->outp[slab].as<char>();
        PF_HIP(hipMemsetAsync(pb + fall, 0, 16, st));
        if (slab_len(PF_OUT_ALIGNSEQ))
            PF_HIP(hipMemcpyAsync(pb + foff[PF_OUT_ALIGNSEQ], S->out[slab].as<char>() + S->txt_off[slab][PF_OUT_ALIGNSEQ], (size_t)slab_len(PF_OUT_ALIGNSEQ), hipMemcpyDeviceToDevice, st));
        NibArgs na;
        int x = 0;
        uint64_t units = 0;
        for (int s = 0; s < N_STREAMS; ++s) {
            if (s == PF_OUT_ALIGNSEQ) continue;
            na.src[x] = S->out[slab].as<char>() + S->txt_off[slab][s];
            na.dst[x] = reinterpret_cast<uint8_t *>(pb + foff[s]);
            na.len[x] = totals[s];
            na.unit0[x] = units;
            na.stream[x] = (uint32_t)s;
            units += (totals[s] + 15) / 16;
            ++x;
        }
        na.unit0[NIB_STREAMS] = units;
        na.flag = reinterpret_cast<uint32_t *>(pb + fall);
        if (units) k_text_nibbles<<<(unsigned)((units + 255) / 256), 256, 0, st>>>(na);
        PF_HIP(hipGetLastError());
        for (int s = 0; s < N_STREAMS; ++s) { S->out_off[slab][s] = foff[s]; S->out_len[slab][s] = flen[s]; }
        S->fetch_base[slab] = pb;
        S->nib[slab] = true;
        out->numeric_packed = 1;
    }
    // no wait for the write pass: the fetches of this slab wait for it on their own stream (text_ev), and the next piece's count
    // pass queues behind it on this one
    if (!S->text_ev[slab]) PF_HIP(hipEventCreateWithFlags(&S->text_ev[slab], hipEventDisableTiming));
    PF_HIP(hipEventRecord(S->text_ev[slab], st));
    // ... and neither does the caller, who may hand the lane to the next pf_call_align_lane at once: that call's stream waits here
    if (!O.read_ev[which]) PF_HIP(hipEventCreateWithFlags(&O.read_ev[which], hipEventDisableTiming));
    PF_HIP(hipEventRecord(O.read_ev[which], st));
    for (int x = 0; x < 4; ++x) out->allele[x] = hc.allele[x];
    out->core_cov = hc.core_cov;
    out->core_num = hc.core_num;
#undef NEED
    return PF_OK;
}

int pf_call_set_numeric_packed(pf_ctx *ctx, int on) {
    if (!ctx) return PF_ERR_ARG;
    CallState *S = state_of(ctx);
    if (!S) return PF_ERR_HIP;
    S->pack_numeric = on != 0;
    return PF_OK;
}

// the text of one stream of a slab as K-TEXT wrote it, whatever the fetches otherwise copy (the way out for a numeric stream whose
// bit is set in the flag word)
int pf_call_fetch_text(pf_ctx *ctx, int slab, int stream, char *dst, uint64_t len) {
    if (!ctx || !ctx->call || slab < 0 || slab >= PF_CALL_SLABS || stream < 0 || stream >= N_STREAMS) return PF_ERR_ARG;
    CallState *S = ctx->call;
    if (len > S->txt_len[slab][stream] || (len && !dst)) return PF_ERR_ARG;
    if (len == 0) return PF_OK;
    if (hipSetDevice(ctx->device) != hipSuccess || !S->copy_stream) return PF_ERR_HIP;
    if (S->text_ev[slab] && hipStreamWaitEvent(S->copy_stream, S->text_ev[slab], 0) != hipSuccess) return PF_ERR_HIP;
    if (hipMemcpyAsync(dst, S->out[slab].as<char>() + S->txt_off[slab][stream], (size_t)len, hipMemcpyDeviceToHost, S->copy_stream) != hipSuccess) return PF_ERR_HIP;
    if (hipStreamSynchronize(S->copy_stream) != hipSuccess) return PF_ERR_HIP;
    return PF_OK;
}

int pf_call_set_alignseq_packed(pf_ctx *ctx, int on) {
    if (!ctx) return PF_ERR_ARG;
    CallState *S = state_of(ctx);
    if (!S) return PF_ERR_HIP;
    S->pack_alignseq = on != 0;
    return PF_OK;
}

int pf_call_text_range_lane(pf_ctx *ctx, int lane, int slab, uint64_t first, uint64_t count, uint64_t var_count_base, pf_call_result *out) {
    return call_text_impl(ctx, lane, slab, first, count, var_count_base, out, false);
}

int pf_call_text_sizes(pf_ctx *ctx, int lane, uint64_t first, uint64_t count, uint64_t var_count_base, pf_call_result *out) {
    return call_text_impl(ctx, lane, 0, first, count, var_count_base, out, true);
}

int pf_call_text_range(pf_ctx *ctx, int slab, uint64_t first, uint64_t count, uint64_t var_count_base, pf_call_result *out) {
    return pf_call_text_range_lane(ctx, 0, slab, first, count, var_count_base, out);
}

int pf_call_text(pf_ctx *ctx, int slab, uint64_t var_count_base, pf_call_result *out) {
    if (!ctx || !ctx->call) return PF_ERR_ARG;
    const int st = pf_call_text_range(ctx, slab, 0, ctx->call->lane[0].nb, var_count_base, out);
    if (st == PF_OK) out->n_called = ctx->call->lane[0].cur.n_called;
    return st;
}

int pf_call_run(pf_ctx *ctx, int slab, uint64_t t0, uint64_t t1, uint64_t var_count_base, uint32_t complex_size, double match,
                double mismatch, double gap, pf_call_result *out) {
    const int st = pf_call_align(ctx, t0, t1, complex_size, match, mismatch, gap, out);
    if (st != PF_OK) return st;
    return pf_call_text(ctx, slab, var_count_base, out);
}

int pf_call_fetch(pf_ctx *ctx, int slab, int stream, char *dst, uint64_t len) {
    if (!ctx || !ctx->call || slab < 0 || slab >= PF_CALL_SLABS || stream < 0 || stream >= N_STREAMS) return PF_ERR_ARG;
    CallState *S = ctx->call;
    if (len > S->out_len[slab][stream] || (len && !dst)) return PF_ERR_ARG;
    if (len == 0) return PF_OK;
    // its own stream, and no context state written: safe beside a pf_call_run on the other slab
    if (hipSetDevice(ctx->device) != hipSuccess) return PF_ERR_HIP;
    if (!S->copy_stream) return PF_ERR_HIP;
    if (S->text_ev[slab] && hipStreamWaitEvent(S->copy_stream, S->text_ev[slab], 0) != hipSuccess) return PF_ERR_HIP;
    if (hipMemcpyAsync(dst, S->fetch_base[slab] + S->out_off[slab][stream], (size_t)len, hipMemcpyDeviceToHost, S->copy_stream) != hipSuccess) return PF_ERR_HIP;
    if (hipStreamSynchronize(S->copy_stream) != hipSuccess) return PF_ERR_HIP;
    return PF_OK;
}

// all ten streams of a slab, one after the other into dst (stream s at off[s] = sum of the lengths before it): ten copies in
// flight, one wait
int pf_call_fetch_slab(pf_ctx *ctx, int slab, char *dst, const uint64_t *len) {
    if (!ctx || !ctx->call || slab < 0 || slab >= PF_CALL_SLABS || !len) return PF_ERR_ARG;
    CallState *S = ctx->call;
    if (hipSetDevice(ctx->device) != hipSuccess) return PF_ERR_HIP;
    if (!S->copy_stream) return PF_ERR_HIP;
    if (S->text_ev[slab] && hipStreamWaitEvent(S->copy_stream, S->text_ev[slab], 0) != hipSuccess) return PF_ERR_HIP;
    uint64_t at = 0;
    bool whole = true;
    for (int s = 0; s < N_STREAMS; ++s) {
        if (len[s] > S->out_len[slab][s] || (len[s] && !dst)) return PF_ERR_ARG;
        whole = whole && len[s] == S->out_len[slab][s];
        at += len[s];
    }
    size_t tl_at = (size_t)-1;
    (void)ctx_begin_at(ctx, PF_K_COPY_TEXT, S->copy_stream, &tl_at);
    if (whole) {   // the slab as it lies: one copy (with the 16-byte tail behind the streams when the numeric streams are packed)
        const uint64_t tail = S->nib[slab] ? 16 : 0;
        if (at + tail && hipMemcpyAsync(dst, S->fetch_base[slab], (size_t)(at + tail), hipMemcpyDeviceToHost, S->copy_stream) != hipSuccess) return PF_ERR_HIP;
    } else {
        at = 0;
        for (int s = 0; s < N_STREAMS; ++s) {
            if (len[s] && hipMemcpyAsync(dst + at, S->fetch_base[slab] + S->out_off[slab][s], (size_t)len[s], hipMemcpyDeviceToHost, S->copy_stream) != hipSuccess)
                return PF_ERR_HIP;
            at += len[s];
        }
    }
    ctx_end_at(ctx, tl_at, S->copy_stream);
    if (hipStreamSynchronize(S->copy_stream) != hipSuccess) return PF_ERR_HIP;
    return PF_OK;
}

// A slab fetched in byte ranges of its packed layout (stream after stream, as pf_call_fetch_slab delivers it), without waiting:
// range number `slot` (0 or 1, alternating) is complete when pf_call_fetch_wait(slot) returns -- the caller copies range i into its
// files while range i + 1 crosses PCIe.
int pf_call_fetch_range(pf_ctx *ctx, int slab, uint64_t first_byte, char *dst, uint64_t len, int slot) {
    if (!ctx || !ctx->call || slab < 0 || slab >= PF_CALL_SLABS || slot < 0 || slot > 1 || (len && !dst)) return PF_ERR_ARG;
    CallState *S = ctx->call;
    uint64_t all = 0;
    for (int s = 0; s < N_STREAMS; ++s) all += S->out_len[slab][s];
    if (first_byte + len > all) return PF_ERR_ARG;
    if (hipSetDevice(ctx->device) != hipSuccess || !S->copy_stream) return PF_ERR_HIP;
    if (!S->fetch_ev[slot] && hipEventCreateWithFlags(&S->fetch_ev[slot], hipEventDisableTiming) != hipSuccess) return PF_ERR_HIP;
    if (S->text_ev[slab] && hipStreamWaitEvent(S->copy_stream, S->text_ev[slab], 0) != hipSuccess) return PF_ERR_HIP;
    size_t tl_at = (size_t)-1;
    (void)ctx_begin_at(ctx, PF_K_COPY_TEXT, S->copy_stream, &tl_at);
    if (len && hipMemcpyAsync(dst, S->fetch_base[slab] + first_byte, (size_t)len, hipMemcpyDeviceToHost, S->copy_stream) != hipSuccess) return PF_ERR_HIP;
    ctx_end_at(ctx, tl_at, S->copy_stream);
    if (hipEventRecord(S->fetch_ev[slot], S->copy_stream) != hipSuccess) return PF_ERR_HIP;
    return PF_OK;
}

int pf_call_fetch_wait(pf_ctx *ctx, int slot) {
    if (!ctx || !ctx->call || slot < 0 || slot > 1 || !ctx->call->fetch_ev[slot]) return PF_ERR_ARG;
    if (hipSetDevice(ctx->device) != hipSuccess) return PF_ERR_HIP;
    return hipEventSynchronize(ctx->call->fetch_ev[slot]) == hipSuccess ? PF_OK : PF_ERR_HIP;
}

}  // extern "C"
