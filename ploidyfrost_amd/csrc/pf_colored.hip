// Colored (multi-sample) coverage on gfx950: the device side of reference src/CCDBG.cpp's
// readCovUni (:123-156) and readCov(string, low, up, colour) (:89-122) over *one* count database per colour
// (CCDBG::CCDBG, :13-43).
//
// MI355X layout: the C databases are joined into ONE table keyed by the stored k-mer -- the single-sample table's lines of ten
// keys chosen by the key's minimizer (pf_device_common.hpp), with one count per colour and slot behind the keys (pf_colored_dev.hpp:
// a line is 80 + 40 C bytes rounded up to 128), so that a k-mer costs one random HBM access for all colours instead of C, and the
// k-mers of a unitig share their lines.  An absent (k-mer, colour) pair is the all-ones count.  K-COV-C walks the unitigs (one
// wavefront per unitig, lanes over its k-mers) and reduces sum / min / max / missing per colour, four colours per pass in registers;
// the range test of readCovUni ("low < count < up" for every k-mer) is min > low && max < up on the host, which keeps the cutoffs out
// of the resident result.  Integer / index work: no MFMA.
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
__global__ void k_ctab_build(uint8_t *base, uint64_t mask, uint32_t line_bytes, int k, const uint64_t *__restrict__ kmers,
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
        const LineSeq sq = kmer_lines(key, rc, k, mask);
        uint64_t line = 0;
        bool done = false;
        for (int tr = 0; !done; ++tr) {   // the single-sample table's probe sequence (count_claim), the count at this colour's place
            line = tr < LINE_TRIES ? seq_line(sq, tr, mask) : tr == LINE_TRIES ? (mix64(key) & mask) : ((line + 1) & mask);
            uint8_t *L = base + line * line_bytes;
            for (int s = 0; s < LINE_KEYS && !done; ++s) {
                unsigned long long *slot = reinterpret_cast<unsigned long long *>(L) + s;
                unsigned long long old = __hip_atomic_load(slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (old == EMPTY_KEY) old = atomicCAS(slot, EMPTY_KEY, key);
                if (old == EMPTY_KEY || old == key) {
                    reinterpret_cast<uint32_t *>(L + 8 * LINE_KEYS)[LINE_KEYS * colour + s] = c;
                    done = true;
                }
            }
        }
    }
    if (__any(nc) && lane_id() == 0) atomicOr(noncanon, 1u);
}

// is some k-mer a key in both orientations (whatever the colours)?  one thread per slot
__global__ void k_ctab_two_strands(const uint8_t *base, uint64_t n_lines, uint32_t line_bytes, int k, unsigned int *flag) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    CTab t{base, n_lines - 1, line_bytes};
    for (; i < n_lines * LINE_KEYS; i += stride) {
        const uint64_t key = reinterpret_cast<const uint64_t *>(ctab_line(t, i / LINE_KEYS))[i % LINE_KEYS];
        if (key == EMPTY_KEY) continue;
        const uint64_t r = rc_kmer(key, k);
        if (r == key) continue;
        if (ctab_find(t, r, kmer_lines(key, r, k, t.mask))) atomicOr(flag, 1u);
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
// The kernel is pf_device.hip's k_cov_join over this table's lines: the same software pipeline (five rows of 64 k-mers in flight a
// wavefront), the last load stage fetching the matching slot's count of every colour; a key that is not in its first line goes
// to the wavefront's slice of a list that k_cov_join_colored_rest looks up.  It takes tables without a k-mer in both orientations
// (every canonically counted set of databases) of up to JOINC_MAX colours; the others keep the loop of k_cov_colored.
namespace {

constexpr int JOINC_MAX = 8;
struct JoinRestC {
    uint64_t g, fwd;
};

template <int K>
__global__ __launch_bounds__(256) void k_cov_join_colored(CTab t, int k_rt, const uint64_t *__restrict__ seq, const uint64_t *__restrict__ off,
                                                          const uint64_t *__restrict__ kpre, const uint64_t *__restrict__ khead,
                                                          const uint32_t *__restrict__ krow, uint32_t n_colors, uint64_t n_kmers,
                                                          uint32_t *__restrict__ gcov, uint64_t g_stride, JoinRestC *__restrict__ rest,
                                                          uint32_t *__restrict__ rest_n, uint32_t rest_cap, uint32_t rows_per_wave) {
    const int k = K ? K : k_rt;
    const int lane = lane_id();
    const uint64_t wave = (((uint64_t)blockIdx.x * blockDim.x) >> 6) + (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const uint64_t n_rows = (n_kmers + 63) >> 6;
    const uint64_t row0 = wave * rows_per_wave;
    const long n_it = row0 < n_rows ? (long)(n_rows - row0 < rows_per_wave ? n_rows - row0 : rows_per_wave) : 0;
    const uint64_t below = lane == 63 ? ~1ull : (((2ull << lane) - 1) & ~1ull);   // bits 1 .. lane
    JoinRestC *my_rest = rest + wave * (rest_cap + 1);
    uint32_t n_rest = 0;
    if (n_it == 0) {
        if (lane == 0) rest_n[wave] = 0;
        return;
    }
    // (stage by stage as in k_cov_join: every stage in every iteration on a clamped row, every lane stores)
    uint64_t s1_hd, s2_pre = 0, s2_wo = 0, s3_w0 = 0, s3_w1 = 0, s4_first = 0, s4_fwd = 0, s4_line = 0, s5_fwd = 0;
    uint32_t s1_kr;
    int s3_s = 0;
    LineKeys s4_keys;
    uint32_t s5_val[JOINC_MAX];
    bool s5_miss = false;
#pragma unroll
    for (int i = 0; i < LINE_KEYS / 2; ++i) s4_keys.q[i] = make_uint4(0, 0, 0, 0);
#pragma unroll
    for (int c = 0; c < JOINC_MAX; ++c) s5_val[c] = 0;
    const long last = n_it - 1;
    auto row_of = [&](long i) { return row0 + (uint64_t)(i < 0 ? 0 : i > last ? last : i); };
    s1_hd = khead[row0];
    s1_kr = krow[row0];
    uint64_t s0_hd = khead[row_of(1)];
    uint32_t s0_kr = krow[row_of(1)];
    for (long it = 0; it < n_it + 4; ++it) {
        {   // row it - 4: its counts have arrived
            const uint64_t g = row_of(it - 4) * 64 + lane;
            const bool miss = s5_miss && it >= 4 && g < n_kmers;
            const uint64_t mm = __ballot(miss);
            const uint32_t at = n_rest + (uint32_t)__popcll(mm & ((1ull << lane) - 1));
            my_rest[miss && at < rest_cap ? at : rest_cap] = JoinRestC{g, s5_fwd};
            n_rest += (uint32_t)__popcll(mm);
#pragma unroll
            for (int c = 0; c < JOINC_MAX; ++c)
                if ((uint32_t)c < n_colors) gcov[(uint64_t)c * g_stride + g] = s5_miss ? CTAB_MISSING : s5_val[c];
        }
        {   // row it - 3: its line's keys have arrived
            bool open;
            const int at = line_slot(s4_keys, s4_first, open);
            s5_miss = at < 0;
            s5_fwd = s4_fwd;
            const uint32_t *cnt = ctab_counts(t, s4_line, at < 0 ? 0 : at);
#pragma unroll
            for (int c = 0; c < JOINC_MAX; ++c) s5_val[c] = (uint32_t)c < n_colors ? cnt[LINE_KEYS * c] : 0;
        }
        {   // row it - 2: its sequence words have arrived
            uint64_t x = s3_w0 << s3_s;
            x |= s3_s ? s3_w1 >> (64 - s3_s) : 0;
            const uint64_t fwd = x >> (64 - 2 * k);
            const uint64_t rc = rc_kmer(fwd, k);
            const LineSeq sq = kmer_lines(fwd, rc, k, t.mask);
            s4_first = rc < fwd ? rc : fwd;   // (one orientation per k-mer in this table: the canonical form is the one to look for)
            s4_fwd = fwd;
            s4_line = sq.line;
            s4_keys = ctab_keys(t, sq.line);
        }
        {   // row it - 1: kpre / off of its lanes' unitigs have arrived
            uint64_t g = row_of(it - 1) * 64 + lane;
            g = g < n_kmers ? g : n_kmers - 1;
            const uint32_t p = (uint32_t)(g - s2_pre);
            const uint64_t *w = seq + s2_wo + (p >> 5);
            s3_s = (int)(p & 31) * 2;
            s3_w0 = w[0];
            s3_w1 = w[1];
        }
        {   // row it: its khead / krow have arrived
            const uint64_t r = row_of(it);
            uint64_t hd = s1_hd & below;
            hd &= r == n_rows - 1 ? (2ull << ((n_kmers - 1) & 63)) - 1 : ~0ull;
            const uint32_t u = s1_kr + (uint32_t)__popcll(hd);
            s2_pre = kpre[u];
            s2_wo = off[u];
        }
        {   // rows it + 1, it + 2
            s1_hd = s0_hd;
            s1_kr = s0_kr;
            const uint64_t r = row_of(it + 2);
            s0_hd = khead[r];
            s0_kr = krow[r];
        }
    }
    if (lane == 0) rest_n[wave] = n_rest;
}

__global__ __launch_bounds__(256) void k_cov_join_colored_rest(CTab t, int k, uint32_t n_colors, const JoinRestC *__restrict__ rest,
                                                               const uint32_t *__restrict__ rest_n, uint32_t rest_cap, uint32_t *__restrict__ gcov,
                                                               uint64_t g_stride, const uint64_t *__restrict__ seq, const uint64_t *__restrict__ off,
                                                               const uint64_t *__restrict__ kpre, const uint64_t *__restrict__ khead,
                                                               const uint32_t *__restrict__ krow, uint64_t n_kmers, uint32_t rows_per_wave) {
    const uint64_t wave = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int lane = lane_id();
    const uint32_t n = rest_n[wave];
    auto look_up = [&](uint64_t g, uint64_t fwd) {
        for (uint32_t c0 = 0; c0 < n_colors; c0 += CPP) {
            const uint32_t n_here = min((uint32_t)CPP, n_colors - c0);
            uint32_t cnt[CPP];
            colored_counts(t, fwd, k, true, c0, n_here, cnt);
#pragma unroll
            for (int j = 0; j < CPP; ++j)
                if ((uint32_t)j < n_here) gcov[(uint64_t)(c0 + j) * g_stride + g] = cnt[j];
        }
    };
    if (n > rest_cap) {   // the slice was full: this wavefront's rows again, from the graph
        const uint64_t n_rows = (n_kmers + 63) >> 6;
        const uint64_t below = lane == 63 ? ~1ull : (((2ull << lane) - 1) & ~1ull);
        for (uint64_t r = wave * rows_per_wave; r < (wave + 1) * rows_per_wave && r < n_rows; ++r) {
            const uint64_t g = r * 64 + lane;
            if (g >= n_kmers) break;
            const uint32_t u = krow[r] + (uint32_t)__popcll(khead[r] & below);
            look_up(g, kmer_at(seq + off[u], (uint32_t)(g - kpre[u]), k));
        }
        return;
    }
    const JoinRestC *mine = rest + wave * (rest_cap + 1);
    for (uint32_t e = lane; e < n; e += WAVE) look_up(mine[e].g, mine[e].fwd);
}

}  // namespace

namespace pf {
// begin / finish like the single-sample join's (pf_device.hip): the kernels on the join's own stream, join_colored_finish() before
// the joined array is read
int join_graph_counts_colored_begin(pf_ctx *ctx) {
    ctx->gcov_c_valid = false;
    ctx->join_c_inflight = false;
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
    const CTab t{ctx->d_ctab, ctx->ctab_cap - 1, ctx->ctab_line_bytes};
    if (!ctx->join_c_done) PF_HIP(hipEventCreateWithFlags(&ctx->join_c_done, hipEventDisableTiming));
    const hipStream_t st = join_stream(ctx);
    size_t at = 0;
    if (ctx->ctab_one_strand && ctx->n_colors <= (uint32_t)JOINC_MAX && ctx->ctab_unread == 0) {
        constexpr uint32_t ROWS = 64;
        const uint64_t n_rows = (ctx->n_kmers + 63) / 64;
        const uint64_t n_waves = ((n_rows + ROWS - 1) / ROWS + 3) / 4 * 4;
        const int blocks = (int)(n_waves / 4);
        const uint32_t rest_cap = ROWS * 64 / 2;   // (half of the k-mers of the wavefront's rows: join_graph_counts_begin says why)
        JoinRestC *rest = (JoinRestC *)ctx_ws(ctx, WS_JOIN_REST, n_waves * (rest_cap + 1) * sizeof(JoinRestC));
        uint32_t *rest_n = (uint32_t *)ctx_ws(ctx, WS_JOIN_REST_N, n_waves * 4);
        if (!rest || !rest_n) { pf::CtxErr{ctx} = "no room for K-COV-C-JOIN's hand-over list"; return PF_ERR_HIP; }
        (void)ctx_begin_at(ctx, PF_K_COV_JOIN, st, &at);
#define PF_JOINC_LAUNCH(KK) k_cov_join_colored<KK><<<blocks, 256, 0, st>>>(t, ctx->k, ctx->d_seq, ctx->d_off, ctx->d_kpre, ctx->d_khead, ctx->d_krow, \
        ctx->n_colors, ctx->n_kmers, ctx->d_gcov_c, stride, rest, rest_n, rest_cap, ROWS)
        if (ctx->k == 25) PF_JOINC_LAUNCH(25);
        else if (ctx->k == 31) PF_JOINC_LAUNCH(31);
        else PF_JOINC_LAUNCH(0);
#undef PF_JOINC_LAUNCH
        ctx_end_at(ctx, at, st);
        (void)ctx_begin_at(ctx, PF_K_COV_JOIN_REST, st, &at);
        k_cov_join_colored_rest<<<blocks, 256, 0, st>>>(t, ctx->k, ctx->n_colors, rest, rest_n, rest_cap, ctx->d_gcov_c, stride, ctx->d_seq, ctx->d_off,
                                                                  ctx->d_kpre, ctx->d_khead, ctx->d_krow, ctx->n_kmers, ROWS);
        ctx_end_at(ctx, at, st);
    } else {
        (void)ctx_begin_at(ctx, PF_K_COV_JOIN, st, &at);
        k_cov_colored<<<ctx_grid(ctx, (uint64_t)ctx->N * 64, 256, 16), 256, 0, st>>>(t, ctx->d_seq, ctx->d_off, ctx->d_len, ctx->k, ctx->ctab_one_strand,
                                                                                            ctx->n_colors, 0, ctx->N, ctx->d_unread, nullptr, nullptr,
                                                                                            nullptr, nullptr, ctx->d_kpre, ctx->d_gcov_c, stride);
        ctx_end_at(ctx, at, st);
    }
    PF_HIP(hipGetLastError());
    PF_HIP(hipEventRecord(ctx->join_c_done, st));
    ctx->join_c_inflight = true;
    return PF_OK;
}

int join_colored_finish(pf_ctx *ctx) {
    if (!ctx->join_c_inflight) return PF_OK;
    ctx->join_c_inflight = false;
    PF_HIP(hipSetDevice(ctx->device));
    PF_HIP(hipEventSynchronize(ctx->join_c_done));
    ctx->gcov_c_valid = true;
    return PF_OK;
}

int join_graph_counts_colored(pf_ctx *ctx) {
    const int rc = join_graph_counts_colored_begin(ctx);
    return rc ? rc : join_colored_finish(ctx);
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
    (void)join_colored_finish(ctx);
    if (ctx->d_gcov_c) { (void)hipFree(ctx->d_gcov_c); ctx->d_gcov_c = nullptr; }
    ctx->gcov_c_valid = false;
    PF_HIP(hipSetDevice(ctx->device));
    if (ctx->d_ctab) { (void)hipFree(ctx->d_ctab); ctx->d_ctab = nullptr; }
    if (ctx->d_unread) { (void)hipFree(ctx->d_unread); ctx->d_unread = nullptr; }
    PF_HIP(hipMalloc(reinterpret_cast<void **>(&ctx->d_unread), n_colors));
    PF_HIP(hipMemcpy(ctx->d_unread, unread.data(), n_colors, hipMemcpyHostToDevice));
    ctx->n_colors = 0;
    // lines of ten keys: distinct keys <= total, and colours that share most k-mers (samples of one species) end near max(n) keys:
    // ten slots for every four keys of the largest database, and never fewer slots than 1.25 * total
    uint64_t cap = 128;
    while (cap * 4 < biggest || cap * LINE_KEYS < total + total / 4 + 1) cap <<= 1;
    const uint32_t line_bytes = ctab_line_bytes(n_colors);
    PF_HIP(hipMalloc(reinterpret_cast<void **>(&ctx->d_ctab), cap * line_bytes));
    PF_HIP(hipMemsetAsync(ctx->d_ctab, 0xFF, cap * line_bytes, ctx->stream));  // EMPTY_KEY keys, MISSING counts
    ctx->ctab_cap = cap;
    ctx->ctab_line_bytes = line_bytes;
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
        k_ctab_build<<<ctx_grid(ctx, n[c], 256, 8), 256, 0, ctx->stream>>>(ctx->d_ctab, cap - 1, line_bytes, ctx->k, pk, pc, n[c], min_count[c],
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
            k_ctab_two_strands<<<ctx_grid(ctx, cap * LINE_KEYS, 256, 8), 256, 0, ctx->stream>>>(ctx->d_ctab, cap, line_bytes, ctx->k, flag_.p);
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
    { const int rc = join_colored_finish(ctx); if (rc) return rc; }   // look-ups of pf_join_counts_begin still on their way
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
        const CTab t{ctx->d_ctab, ctx->ctab_cap - 1, ctx->ctab_line_bytes};
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
    const CTab t{ctx->d_ctab, ctx->ctab_cap - 1, ctx->ctab_line_bytes};
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
