// gfx950 device layer behind include/ploidyfrost_hip.h: context, tables, and the kernels
//   K-TABLE  count-table build           (CKMCFile::OpenForRA data -> device hash table)
//   K-ADJ    end-k-mer join -> CSR       (Bifrost neighbour discovery, NeighborIterator.tcc:25-47)
//   K-COV    per-unitig coverage         (CDBG::readCov(UnitigMap), src/CDBG.cpp:66-120)
//   K-BFS    superbubble traversal       (CDBG::extractSuperBubble_ptr, src/CDBG.cpp:253-372)
//   K-STRCOV site-string coverage        (CDBG::readCov(string), src/CDBG.cpp:29-60)
// K-ALN lives in pf_align.hip.  Integer / index-bound work: no MFMA anywhere.
#include <hip/hip_runtime.h>
#include <sys/mman.h>

#include <algorithm>
#include <cstdio>
#include <cstring>
#include <string>
#include <type_traits>
#include <vector>

#include "pf_bfs.hpp"
#include "pf_bfs_huge.hpp"
#include "pf_cov_stream.hpp"
#include "pf_ctx.hpp"
#include "pf_device_common.hpp"
#include "pf_scan.hpp"
#include "ploidyfrost_hip.h"

using namespace pf;

static std::string g_create_err;

#define PF_HIP(call)                                                                         \
    do {                                                                                     \
        hipError_t e_ = (call);                                                              \
        if (e_ != hipSuccess) {                                                              \
            pf::CtxErr{ctx} = std::string(#call) + ": " + hipGetErrorString(e_);                   \
            return PF_ERR_HIP;                                                               \
        }                                                                                    \
    } while (0)

// ------------------------------------------------------------------------------------------
// kernels
// ------------------------------------------------------------------------------------------

__global__ void k_fill_slots(Slot *t, uint64_t n) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) {
        uint4 v;
        v.x = 0xFFFFFFFFu; v.y = 0xFFFFFFFFu; v.z = 0xFFFFFFFFu; v.w = 0;
        *reinterpret_cast<uint4 *>(t + i) = v;
    }
}

// K-TABLE: one thread per database record.  `noncanon` is set when some key is larger than its reverse complement: only then can
// the table hold a k-mer in both orientations (check_table_strands).
__global__ void k_table_build(CountLine *t, uint64_t mask, int k, const uint64_t *__restrict__ kmers,
                              const uint32_t *__restrict__ counts, uint64_t n, uint64_t min_count, uint64_t max_count, unsigned int *noncanon) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    bool nc = false;
    for (; i < n; i += stride) {
        const uint32_t c = counts[i];
        if (c < min_count || c > max_count) continue;  // not retrievable (kmc_file.cpp:1459)
        const uint64_t key = kmers[i];
        const uint64_t rc = rc_kmer(key, k);
        nc |= rc < key;
        *count_claim(t, mask, key, kmer_lines(key, rc, k, mask)) = c;
    }
    if (__any(nc) && lane_id() == 0) atomicOr(noncanon, 1u);
}

// K-KMC: the records of a KMC database as they lie in <db>.kmc_suf -- (k-p)/4 suffix bytes, most significant first, then the
// counter, least significant byte first (KMC/kmc_api/kmc_file.cpp:775-782, 1383-1462) -- decoded into (k-mer, count) pairs.
// lut[e] = first record of prefix-table entry e (KMC1: e = prefix; KMC2: e = bin * 4^p + prefix), lut[n_lut] = n.  One thread
// decodes KMC_PER_THREAD consecutive records: one binary search, then a forward walk over the entries.
constexpr int KMC_PER_THREAD = 4;
__global__ void k_kmc_decode(const uint8_t *__restrict__ rec, uint64_t n, uint32_t sb, uint32_t cb, const uint64_t *__restrict__ lut,
                             uint64_t n_lut, uint32_t pref_mask, uint32_t suffix_bits, uint64_t *__restrict__ kmers,
                             uint32_t *__restrict__ counts) {
    const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    const uint32_t rb = sb + cb;
    for (uint64_t i0 = t * KMC_PER_THREAD; i0 < n; i0 += stride * KMC_PER_THREAD) {
        uint64_t lo = 0, hi = n_lut;  // the last entry e with lut[e] <= i0
        while (hi - lo > 1) {
            const uint64_t mid = (lo + hi) >> 1;
            if (lut[mid] <= i0) lo = mid; else hi = mid;
        }
        uint64_t e = lo;
        const uint64_t i1 = i0 + KMC_PER_THREAD < n ? i0 + KMC_PER_THREAD : n;
        for (uint64_t i = i0; i < i1; ++i) {
            while (e + 1 < n_lut && lut[e + 1] <= i) ++e;
            const uint8_t *q = rec + i * rb;
            uint64_t s = 0, c = 0;
            for (uint32_t b = 0; b < sb; ++b) s = (s << 8) | q[b];
            for (uint32_t b = 0; b < cb; ++b) c |= (uint64_t)q[sb + b] << (8 * b);
            kmers[i] = ((uint64_t)(e & pref_mask) << suffix_bits) | s;
            counts[i] = (uint32_t)c;
        }
    }
}

// Does the table hold both orientations of some k-mer?  One thread per slot.  (Run only for a table with keys that are not
// canonical: when every key is at most its reverse complement, the reverse complement of a key is not a key.)
__global__ void k_table_two_strands(const CountLine *__restrict__ t, uint64_t n_lines, int k, unsigned int *flag) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (; i < n_lines * LINE_KEYS; i += stride) {
        const uint64_t key = t[i / LINE_KEYS].key[i % LINE_KEYS];
        if (key == EMPTY_KEY) continue;
        const uint64_t r = rc_kmer(key, k);
        uint32_t c;
        if (r != key && count_find(t, n_lines - 1, r, kmer_lines(key, r, k, n_lines - 1), c)) atomicOr(flag, 1u);
    }
}

__global__ void k_lookup(const CountLine *__restrict__ t, uint64_t mask, int k, bool one_strand, const uint64_t *__restrict__ kmers,
                         uint64_t n, uint32_t *__restrict__ counts, uint8_t *__restrict__ found) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) {
        uint32_t c = 0;
        bool ok = canonical_count(t, mask, kmers[i], k, c, one_strand);
        counts[i] = ok ? c : 0;
        found[i] = ok;
    }
}

// K-ADJ insert: the canonical k-mer at each extremity of each unitig.
// value = (u << 2) | (is_tail << 1) | stored_is_canonical; the minimum wins, i.e. the lowest
// unitig and its head first -- the order a sequential insert-if-absent would produce.
__global__ void k_adj_insert(Slot *t, uint64_t mask, const uint64_t *__restrict__ seq, const uint64_t *__restrict__ off,
                             const uint32_t *__restrict__ len, uint32_t N, int k) {
    uint32_t u = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t stride = gridDim.x * blockDim.x;
    for (; u < N; u += stride) {
        const uint64_t *w = seq + off[u];
        const uint32_t L = len[u];
        const int ends = L > (uint32_t)k ? 2 : 1;
        for (int e = 0; e < ends; ++e) {
            const uint64_t x = kmer_at(w, e ? L - k : 0, k);
            const uint64_t r = rc_kmer(x, k);
            const uint64_t c = x < r ? x : r;
            const uint32_t v = (u << 2) | ((uint32_t)e << 1) | (x == c ? 1u : 0u);
            uint64_t s = mix64(c) & mask;
            for (;;) {
                unsigned long long old = atomicCAS(reinterpret_cast<unsigned long long *>(&t[s].key), EMPTY_KEY, c);
                if (old == EMPTY_KEY || old == c) { atomicMin(&t[s].val, v); break; }
                s = (s + 1) & mask;
            }
        }
    }
}

__device__ inline uint32_t adj_find(const Slot *__restrict__ t, uint64_t mask, uint64_t q, int k) {
    const uint64_t r = rc_kmer(q, k);
    const uint64_t c = q < r ? q : r;
    uint32_t v;
    if (!table_find(t, mask, c, v)) return NONE;
    // strand = (the query itself, not its twin, is what is stored): CompactedDBG.tcc:1476-1499
    const bool strand = ((q == c) == ((v & 1u) != 0));
    return ((v >> 2) << 1) | (strand ? 0u : 1u);
}

// K-ADJ probe: one thread per oriented vertex, 4 successor + 4 predecessor probes, two 16-B rows out.
__global__ void k_adj_probe(const Slot *__restrict__ t, uint64_t mask, const uint64_t *__restrict__ seq,
                            const uint64_t *__restrict__ off, const uint32_t *__restrict__ len, uint32_t N, int k,
                            uint32_t *__restrict__ succ, uint32_t *__restrict__ pred) {
    uint32_t ov = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t stride = gridDim.x * blockDim.x;
    const uint64_t kmask = (1ull << (2 * k)) - 1;
    for (; ov < 2 * N; ov += stride) {
        const uint32_t u = ov >> 1;
        const uint64_t *w = seq + off[u];
        const uint32_t L = len[u];
        const uint64_t head = kmer_at(w, 0, k);
        const uint64_t tail = L > (uint32_t)k ? kmer_at(w, L - k, k) : head;
        // NeighborIterator.tcc:16-17
        const uint64_t km_head = (ov & 1) ? rc_kmer(tail, k) : head;
        const uint64_t km_tail = (ov & 1) ? rc_kmer(head, k) : tail;
        uint4 so, po;
        so.x = adj_find(t, mask, ((km_tail << 2) | 0) & kmask, k);
        so.y = adj_find(t, mask, ((km_tail << 2) | 1) & kmask, k);
        so.z = adj_find(t, mask, ((km_tail << 2) | 2) & kmask, k);
        so.w = adj_find(t, mask, ((km_tail << 2) | 3) & kmask, k);
        po.x = adj_find(t, mask, (km_head >> 2) | (0ull << (2 * (k - 1))), k);
        po.y = adj_find(t, mask, (km_head >> 2) | (1ull << (2 * (k - 1))), k);
        po.z = adj_find(t, mask, (km_head >> 2) | (2ull << (2 * (k - 1))), k);
        po.w = adj_find(t, mask, (km_head >> 2) | (3ull << (2 * (k - 1))), k);
        *reinterpret_cast<uint4 *>(succ + (size_t)ov * 4) = so;
        *reinterpret_cast<uint4 *>(pred + (size_t)ov * 4) = po;
    }
}

// graph upload: per-unitig checks and k-mer counts; then, from the scanned counts, the window / row -> unitig maps and the
// unitig-start bits the k-mer-parallel kernels use
__global__ void k_graph_check(const uint64_t *__restrict__ off, const uint32_t *__restrict__ len, uint32_t N, int k, uint64_t total_words,
                              uint64_t *__restrict__ cnt, unsigned int *bad) {
    const uint32_t u = blockIdx.x * blockDim.x + threadIdx.x;
    if (u > N) return;
    if (u == N) { cnt[u] = 0; return; }
    const uint32_t L = len[u];
    const uint64_t a = off[u], b = off[u + 1];
    unsigned int e = 0;
    if (L < (uint32_t)k) e |= 1;
    if (b < a || (b - a) * 32 < L || b > total_words) e |= 2;
    if (e) atomicOr(bad, e);
    cnt[u] = L >= (uint32_t)k ? (uint64_t)(L - (uint32_t)k + 1) : 0;
}
__global__ void k_fill_u32(uint32_t *p, uint64_t n, uint32_t v) {
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) p[i] = v;
}
__global__ void k_graph_index(const uint64_t *__restrict__ kpre, uint32_t N, uint64_t n_win, uint64_t n_row, uint32_t *__restrict__ kwin,
                              uint32_t *__restrict__ krow, uint64_t *__restrict__ khead) {
    const uint32_t u = blockIdx.x * blockDim.x + threadIdx.x;
    if (u > N) return;
    const uint64_t g0 = kpre[u];
    atomicOr((unsigned long long *)&khead[g0 >> 6], 1ull << (g0 & 63));   // u == N: the bit behind the last k-mer
    if (u == N) return;
    const uint64_t g1 = kpre[u + 1];
    for (uint64_t w = (g0 + 255) >> 8; w < n_win && (w << 8) < g1; ++w) kwin[w] = u;
    for (uint64_t r = (g0 + 63) >> 6; r < n_row && (r << 6) < g1; ++r) krow[r] = u;
}

__global__ void k_mark_candidates(const uint32_t *__restrict__ succ, uint32_t n_ov, uint8_t *__restrict__ flag) {
    uint32_t ov = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t stride = gridDim.x * blockDim.x;
    for (; ov < n_ov; ov += stride) {
        const uint4 s = *reinterpret_cast<const uint4 *>(succ + (size_t)ov * 4);
        const int d = (s.x != NONE) + (s.y != NONE) + (s.z != NONE) + (s.w != NONE);
        flag[ov] = d > 1;
    }
}

// K-COV: k-mer-parallel.  Global k-mer g (unitigs laid end to end) belongs to unitig u with kpre[u] <= g < kpre[u+1].
// A wavefront takes a window of 256 consecutive k-mers, four per lane (lane, lane + 64, ...), so that every lane
// keeps four table probes in flight -- the kernel is bound by the latency of dependent random accesses, not by
// bytes -- and the work per wave does not depend on unitig lengths.  The window's slice of the prefix array sits in
// LDS; each lane finds its unitig by binary search there.  Per-unitig sum / min / missing are combined with a
// segmented wave scan (lanes of one unitig are contiguous) and one atomic per (window row, unitig) into outputs
// initialised by k_cov_init (sum 0, min 10000 as src/CDBG.cpp:71, missing 0).
constexpr int KCOV_PER_LANE = 4;
constexpr int KCOV_WIN = WAVE * KCOV_PER_LANE;  // 256, also the granularity of pf_ctx::d_kwin

__global__ void k_cov_init(uint32_t n, uint64_t *__restrict__ out_sum, uint32_t *__restrict__ out_min, uint8_t *__restrict__ out_miss) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t stride = gridDim.x * blockDim.x;
    for (; i < n; i += stride) { out_sum[i] = 0; out_min[i] = 10000; out_miss[i] = 0; }
}

__global__ __launch_bounds__(256) void k_cov(const CountLine *__restrict__ t, uint64_t mask, int k, const uint64_t *__restrict__ seq,
                                             const uint64_t *__restrict__ off, const uint64_t *__restrict__ kpre, const uint32_t *__restrict__ kwin, uint32_t N,
                                             bool one_strand, int exact, uint32_t u0, uint64_t g_begin, uint64_t g_end, uint64_t w_begin,
                                             uint64_t n_win, uint64_t *__restrict__ out_sum, uint32_t *__restrict__ out_min,
                                             uint8_t *__restrict__ out_miss, uint32_t *__restrict__ gcov) {
    __shared__ uint64_t s_pre[4][KCOV_WIN + 2];
    const int lane = lane_id();
    const int wi = threadIdx.x >> 6;
    uint64_t *P = s_pre[wi];
    const uint64_t wave = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const uint64_t n_waves = ((uint64_t)gridDim.x * blockDim.x) >> 6;
    for (uint64_t wx = wave; wx < n_win; wx += n_waves) {
        const uint64_t w = w_begin + wx;
        const uint64_t g0 = w * KCOV_WIN;
        const uint32_t ub = kwin[w];
        // P[x] = kpre[ub + x], x = 0..256: the window cannot reach beyond unitig ub + 256
        for (int x = lane; x <= KCOV_WIN; x += WAVE) {
            const uint64_t ux = (uint64_t)ub + x;
            P[x] = kpre[ux < N ? ux : N];
        }
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");  // lanes read what other lanes of the wave wrote
        __builtin_amdgcn_wave_barrier();
        CountProbe pr[KCOV_PER_LANE];
        uint32_t uu[KCOV_PER_LANE];
        bool act[KCOV_PER_LANE];
#pragma unroll
        for (int j = 0; j < KCOV_PER_LANE; ++j) {
            const uint64_t g = g0 + (uint64_t)j * WAVE + lane;
            act[j] = g >= g_begin && g < g_end;
            uu[j] = NONE;
            if (act[j]) {
                int lo = 0, hi = KCOV_WIN;  // largest x with P[x] <= g (P[0] <= g0 by construction)
                while (lo < hi) {
                    const int mid = (lo + hi + 1) >> 1;
                    if (P[mid] <= g) lo = mid; else hi = mid - 1;
                }
                uu[j] = ub + (uint32_t)lo;
                const uint64_t fwd = kmer_at(seq + off[uu[j]], (uint32_t)(g - P[lo]), k);
                if (exact) {  // database without canonical counting: the k-mer as it reads in the wanted orientation, nothing else
                    const uint64_t key = exact == 2 ? rc_kmer(fwd, k) : fwd;
                    count_probe_at(t, key, kmer_lines(key, exact == 2 ? fwd : rc_kmer(fwd, k), k, mask), pr[j]);
                } else {
                    count_probe(t, mask, fwd, k, one_strand, pr[j]);
                }
            }
        }
#pragma unroll
        for (int j = 0; j < KCOV_PER_LANE; ++j) {
            uint64_t sum = 0;
            uint32_t mn = 0xFFFFFFFFu;
            uint32_t miss = 0;
            if (act[j]) {
                uint32_t c;
                if (count_finish(t, mask, pr[j], c)) { sum = c; mn = c; }
                else miss = 1;
            }
            if (gcov) {  // K-COV-JOIN: the count goes to the k-mer's place in graph order (coalesced), nothing is reduced
                if (act[j]) gcov[g0 + (uint64_t)j * WAVE + lane] = miss ? GCOV_MISSING : (uint32_t)sum;
                continue;
            }
            // segmented inclusive scan over lanes of the same unitig (contiguous runs, inactive lanes carry NONE)
            const uint32_t u = uu[j];
#pragma unroll
            for (int o = 1; o < WAVE; o <<= 1) {
                const uint32_t uo = __shfl_up(u, o, WAVE);
                const uint32_t slo = __shfl_up((uint32_t)sum, o, WAVE), shi = __shfl_up((uint32_t)(sum >> 32), o, WAVE);
                const uint32_t mo = __shfl_up(mn, o, WAVE), xo = __shfl_up(miss, o, WAVE);
                if (lane >= o && uo == u) {
                    sum += ((uint64_t)shi << 32) | slo;
                    mn = mo < mn ? mo : mn;
                    miss |= xo;
                }
            }
            const uint32_t un = __shfl_down(u, 1, WAVE);
            if (u != NONE && (lane == WAVE - 1 || un != u)) {  // last lane of its unitig in this row
                const uint32_t o = u - u0;
                atomicAdd(reinterpret_cast<unsigned long long *>(out_sum + o), (unsigned long long)sum);
                if (mn != 0xFFFFFFFFu) atomicMin(out_min + o, mn);
                if (miss) out_miss[o] = 1;
            }
        }
        __builtin_amdgcn_wave_barrier();  // P is rewritten by the next window
    }
}

// K-COV-JOIN: every graph k-mer looked up in the count table, its count left at the k-mer's place in graph order (gcov).
// A wavefront takes rows of 64 consecutive k-mers, a k-mer per lane.  A look-up is a chain of five dependent loads --
//   S1 the row's khead word (a bit per k-mer, set at unitig starts) and krow (unitig of the row's first k-mer): the lane's unitig
//      is a population count away, no search;
//   S2 kpre / off of that unitig;   S3 the two sequence words that hold the k-mer;
//   S4 the ten keys of the line its minimizer names (the lanes of a run of k-mers fetch the same line);   S5 the count of the slot
//      that matched
// -- and a wavefront that walks the chain row by row spends its time waiting (measured: 8 us a row at eight wavefronts a SIMD).  So
// the loop is software-pipelined: iteration i issues S1 of row i + 1, S2 of row i, S3 of row i - 1, S4 of row i - 2, S5 of row
// i - 3 and stores row i - 4, each stage consuming what the iteration before issued: five rows in flight per wavefront, the loads
// retired in the order they were issued.
// A key that is not in the first line of its sequence (one in six: the line was full) would stall that pipeline for the three
// further loads of its buddy line: the lane writes (k-mer index, k-mer) to its wavefront's slice of `rest` instead, and
// k_cov_join_rest looks those up afterwards (a slice that is full -- every second k-mer of the wavefront's rows missed its first
// line: a stretch of repeats -- makes k_cov_join_rest look all of those rows up again).
struct JoinRest {
    uint64_t g, fwd;
};
template <int K>
__global__ __launch_bounds__(256) void k_cov_join(const CountLine *__restrict__ t, uint64_t mask, int k_rt, const uint64_t *__restrict__ seq,
                                                  const uint64_t *__restrict__ off, const uint64_t *__restrict__ kpre,
                                                  const uint64_t *__restrict__ khead, const uint32_t *__restrict__ krow, bool one_strand,
                                                  uint64_t n_kmers, uint32_t *__restrict__ gcov, JoinRest *__restrict__ rest,
                                                  uint32_t *__restrict__ rest_n, uint32_t rest_cap, uint32_t rows_per_wave) {
    const int k = K ? K : k_rt;
    const int lane = lane_id();
    // (wave-uniform by construction; said so, the row's khead / krow come through the scalar cache, beside the vector loads)
    const uint64_t wave = (((uint64_t)blockIdx.x * blockDim.x) >> 6) + (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const uint64_t n_rows = (n_kmers + 63) >> 6;
    // rows wave * rows_per_wave ...: a wavefront is short-lived (64 rows: a tenth of a millisecond), so that the kernels of streams
    // of higher priority -- the join runs beside findSuperBubble's -- get the CUs it gives back
    const uint64_t row0 = wave * rows_per_wave;
    const long n_it = row0 < n_rows ? (long)(n_rows - row0 < rows_per_wave ? n_rows - row0 : rows_per_wave) : 0;
    const uint64_t below = lane == 63 ? ~1ull : (((2ull << lane) - 1) & ~1ull);   // bits 1 .. lane
    JoinRest *my_rest = rest + wave * (rest_cap + 1);
    uint32_t n_rest = 0;
    // what a stage leaves for the next iteration
    uint64_t s1_hd = 0;  uint32_t s1_kr = 0;
    uint64_t s2_pre = 0, s2_wo = 0;
    uint64_t s3_w0 = 0, s3_w1 = 0;  int s3_s = 0;
    LineKeys s4_keys;  uint64_t s4_first = 0, s4_fwd = 0, s4_line = 0;
    uint32_t s5_val = 0;  bool s5_miss = false;  uint64_t s5_fwd = 0;
#pragma unroll
    for (int i = 0; i < LINE_KEYS / 2; ++i) s4_keys.q[i] = make_uint4(0, 0, 0, 0);
    if (n_it == 0) {
        if (lane == 0) rest_n[wave] = 0;
        return;
    }
    // Every stage runs in every iteration, on a row clamped to the wavefront's own rows, and every lane stores: no branch stands
    // between the loads of one iteration and their use in the next, so the compiler waits for exactly the load a stage needs
    // (s_waitcnt vmcnt(n), n > 0) instead of for all of them.  What the first four and the last four iterations compute for rows
    // that are not there is overwritten (a row's real store comes after, from the same lane) or never used.
    const long last = n_it - 1;
    auto row_of = [&](long i) { return row0 + (uint64_t)(i < 0 ? 0 : i > last ? last : i); };
    s1_hd = khead[row0];
    s1_kr = krow[row0];
    uint64_t s0_hd = khead[row_of(1)];   // two rows ahead: what an iteration loads is not asked for before the next one
    uint32_t s0_kr = krow[row_of(1)];
    for (long it = 0; it < n_it + 4; ++it) {
        // row it - 4: its count has arrived
        {
            const uint64_t g = row_of(it - 4) * 64 + lane;
            const bool miss = s5_miss && it >= 4 && g < n_kmers;
            const uint64_t mm = __ballot(miss);
            const uint32_t at = n_rest + (uint32_t)__popcll(mm & ((1ull << lane) - 1));
            my_rest[miss && at < rest_cap ? at : rest_cap] = JoinRest{g, s5_fwd};   // slot rest_cap of a slice: nobody reads it
            n_rest += (uint32_t)__popcll(mm);
            gcov[g] = s5_miss ? GCOV_MISSING : s5_val;   // (a row's padding lanes included: gcov holds whole rows)
        }
        // row it - 3: its line's keys have arrived
        {
            bool open;
            const int at = line_slot(s4_keys, s4_first, open);
            s5_miss = at < 0;
            s5_fwd = s4_fwd;
            s5_val = t[s4_line].val[at < 0 ? 0 : at];
        }
        // row it - 2: its sequence words have arrived
        {
            uint64_t x = s3_w0 << s3_s;
            x |= s3_s ? s3_w1 >> (64 - s3_s) : 0;
            const uint64_t fwd = x >> (64 - 2 * k);
            const uint64_t rc = rc_kmer(fwd, k);
            const LineSeq sq = kmer_lines(fwd, rc, k, mask);
            s4_first = (one_strand && rc < fwd) ? rc : fwd;
            s4_fwd = fwd;
            s4_line = sq.line;
            s4_keys = load_line_keys(t, sq.line);
        }
        // row it - 1: kpre / off of its lanes' unitigs have arrived
        {
            uint64_t g = row_of(it - 1) * 64 + lane;
            g = g < n_kmers ? g : n_kmers - 1;
            const uint32_t p = (uint32_t)(g - s2_pre);
            const uint64_t *w = seq + s2_wo + (p >> 5);
            s3_s = (int)(p & 31) * 2;
            s3_w0 = w[0];
            s3_w1 = w[1];   // (the sequence array is padded by two words)
        }
        // row it: its khead / krow have arrived
        {
            const uint64_t r = row_of(it);
            uint64_t hd = s1_hd & below;
            hd &= r == n_rows - 1 ? (2ull << ((n_kmers - 1) & 63)) - 1 : ~0ull;   // lanes beyond the last k-mer: its unitig (the bit behind it is set)
            const uint32_t u = s1_kr + (uint32_t)__popcll(hd);
            s2_pre = kpre[u];
            s2_wo = off[u];
        }
        // rows it + 1, it + 2
        {
            s1_hd = s0_hd;
            s1_kr = s0_kr;
            const uint64_t r = row_of(it + 2);
            s0_hd = khead[r];
            s0_kr = krow[r];
        }
    }
    if (lane == 0) rest_n[wave] = n_rest;   // above rest_cap: the slice was full, k_cov_join_rest takes the wavefront's rows again
}

// the look-ups K-COV-JOIN's pipeline handed on: wavefront w takes slice w.  A slice that was full (more than half of the k-mers of
// the wavefront's rows missed their first line: a stretch of repeats) holds only some of them: that wavefront's rows are looked up
// again from the graph, every look-up walked to its end where it stands.
__global__ __launch_bounds__(256) void k_cov_join_rest(const CountLine *__restrict__ t, uint64_t mask, int k, bool one_strand,
                                                       const JoinRest *__restrict__ rest, const uint32_t *__restrict__ rest_n, uint32_t rest_cap,
                                                       uint32_t *__restrict__ gcov, const uint64_t *__restrict__ seq, const uint64_t *__restrict__ off,
                                                       const uint64_t *__restrict__ kpre, const uint64_t *__restrict__ khead,
                                                       const uint32_t *__restrict__ krow, uint64_t n_kmers, uint32_t rows_per_wave) {
    const uint64_t wave = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int lane = lane_id();
    const uint32_t n = rest_n[wave];
    if (n > rest_cap) {
        const uint64_t n_rows = (n_kmers + 63) >> 6;
        const uint64_t below = lane == 63 ? ~1ull : (((2ull << lane) - 1) & ~1ull);
        for (uint64_t r = wave * rows_per_wave; r < (wave + 1) * rows_per_wave && r < n_rows; ++r) {
            const uint64_t g = r * 64 + lane;
            if (g >= n_kmers) break;
            const uint32_t u = krow[r] + (uint32_t)__popcll(khead[r] & below);
            uint32_t c;
            gcov[g] = canonical_count(t, mask, kmer_at(seq + off[u], (uint32_t)(g - kpre[u]), k), k, c, one_strand) ? c : GCOV_MISSING;
        }
        return;
    }
    const JoinRest *mine = rest + wave * (rest_cap + 1);
    for (uint32_t e = lane; e < n; e += WAVE) {
        const JoinRest x = mine[e];
        uint32_t c;
        bool found;
        if (one_strand) {   // one form to look for, and its first line is known to be full of other keys
            const uint64_t rc = rc_kmer(x.fwd, k);
            found = count_find(t, mask, rc < x.fwd ? rc : x.fwd, kmer_lines(x.fwd, rc, k, mask), c, 1);
        } else {
            found = canonical_count(t, mask, x.fwd, k, c, false);
        }
        gcov[x.g] = found ? c : GCOV_MISSING;
    }
}

// K-STRCOV: one thread per string (strings are k .. k+few bases long).
__global__ void k_strcov(const CountLine *__restrict__ t, uint64_t mask, int k, bool one_strand, const char *__restrict__ text,
                         const uint64_t *__restrict__ str_off, uint32_t n_str, uint32_t low, uint32_t up,
                         uint64_t *__restrict__ out_sum, uint8_t *__restrict__ out_ok, uint8_t *__restrict__ out_miss) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t stride = gridDim.x * blockDim.x;
    const uint64_t kmask = (1ull << (2 * k)) - 1;
    for (; i < n_str; i += stride) {
        const char *s = text + str_off[i];
        const uint32_t L = (uint32_t)(str_off[i + 1] - str_off[i]);
        uint64_t sum = 0;
        uint8_t ok = 1, miss = 0;
        StringWindow win;
        for (uint32_t j = 0; j < L; ++j) {
            const uint64_t x = win.push(s[j], kmask, (uint32_t)k);
            if (j + 1 >= (uint32_t)k) {
                uint32_t c;
                if (!canonical_count(t, mask, x, k, c, one_strand)) { miss = 1; break; }
                if (c > low && c < up) sum += c;
                else { sum = 0; ok = 0; break; }  // src/CDBG.cpp:45-50
            }
        }
        out_sum[i] = sum;
        out_ok[i] = ok;
        out_miss[i] = miss;
    }
}

// K-BFS, LDS tier: 4 waves per block, each with its own LDS slice of CAP entries.
constexpr uint32_t BFS_LDS_CAP = 128;
constexpr uint32_t BFS_BIG_CAP = 1u << 12;  // linear-scan tables: beyond this the direct-indexed tier takes over

struct BfsOut {
    pf_bfs_record *rec;
    uint32_t *pool;
    uint64_t pool_cap;
    unsigned long long *pool_head;  // running total (may exceed pool_cap: tells the size needed)
    uint32_t *deferred;             // candidate indices for the big tier
    unsigned int *n_deferred;
    uint32_t *deferred2;            // ... and for the direct-indexed tier
    unsigned int *n_deferred2;
    // pf_bfs_live_deferred: the same hand-over, at once, in host memory the caller polls while the kernel runs: entry d =
    // entrance << 32 | (candidate index + 1)
    unsigned long long *live;
    uint32_t live_cap;
    unsigned int *n_live;   // entries of the live list (notices of traversals still running on the device included)
    uint32_t hint_at;       // a traversal is entered when it reaches this many vertices
};

// Per-wave bump allocation in the vertex pool: a wave reserves BFS_POOL_CHUNK entries with one
// atomic and hands them out locally, so the single pool head is touched once per ~50 candidates instead
// of once per candidate (one hot word saturates at ~90 atomics/us chip-wide).
constexpr uint32_t BFS_POOL_CHUNK = 256;
struct BfsAlloc {
    unsigned long long cur = 0, end = 0;
};

__device__ inline void bfs_emit(const BfsOut &o, BfsAlloc &al, uint64_t ci, uint32_t s, const BfsResult &r, const BfsStore &st) {
    const int lane = lane_id();
    // what the host replay needs: seen[] when an exit was found, the cycle set otherwise
    const bool want_seen = r.outcome != PF_BFS_NONE;
    const uint32_t n_list = want_seen ? r.n_seen : (r.flag_cycle ? r.n_cyc : 0);
    const uint32_t *src = want_seen ? st.ent : st.cyc;
    unsigned long long off = 0;
    if (n_list) {
        if (n_list > al.end - al.cur) {
            const uint32_t want = n_list > BFS_POOL_CHUNK ? n_list : BFS_POOL_CHUNK;
            unsigned long long got = 0;
            if (lane == 0) got = atomicAdd(o.pool_head, (unsigned long long)want);
            got = ((unsigned long long)__shfl((uint32_t)(got >> 32), 0, WAVE) << 32) | __shfl((uint32_t)got, 0, WAVE);
            al.cur = got;
            al.end = got + want;
        }
        off = al.cur;
        al.cur += n_list;
    }
    if (off + n_list <= o.pool_cap)
        for (uint32_t i = lane; i < n_list; i += WAVE) o.pool[off + i] = src[i];
    if (lane == 0) {
        pf_bfs_record rec;
        rec.entrance = s;
        rec.exit = r.exit_ov;
        rec.n_seen = r.n_seen;
        rec.n_list = n_list;
        rec.list_off = off;
        rec.outcome = r.outcome;
        rec.flag_cycle = r.flag_cycle;
        rec.flag_tip = r.flag_tip;
        rec.strict = r.strict;
        rec.pad_ = 0;
        o.rec[ci] = rec;
    }
}

// thread tier (pf_bfs.hpp): one thread per candidate; what outgrows its 8-entry tables is listed for the wavefront tier
__global__ __launch_bounds__(256) void k_bfs_thread(const uint32_t *__restrict__ succ, const uint32_t *__restrict__ pred,
                                                    const uint32_t *__restrict__ cand, uint64_t c0, uint64_t c1, BfsOut o, uint32_t *wave_list,
                                                    unsigned int *n_wave_list) {
    __shared__ uint32_t s_ent[BFS_THREAD_CAP * 256];
    __shared__ uint32_t s_todo[BFS_THREAD_CAP * 256];
    __shared__ uint32_t s_cyc[BFS_THREAD_CAP * 256];
    __shared__ uint8_t s_meta[BFS_THREAD_CAP * 256];
    const uint32_t tid = threadIdx.x;
    const BfsThreadStore st{s_ent + tid, s_todo + tid, s_cyc + tid, s_meta + tid, 256};
    const uint64_t c = c0 + (uint64_t)blockIdx.x * 256 + tid;
    const int lane = lane_id();
    const bool active = c < c1;
    BfsResult r;
    r.overflow = false;
    r.outcome = PF_BFS_NONE;
    r.n_seen = r.n_cyc = 0;
    r.flag_cycle = 0;
    uint32_t s = 0;
    if (active) {
        s = cand[c];
        r = bfs_traverse_thread(succ, pred, st, s);
    }
    const bool done = active && !r.overflow;
    // vertex lists: one atomic per wavefront for the space of all its lists
    const bool want_seen = r.outcome != PF_BFS_NONE;
    const uint32_t n_list = done ? (want_seen ? r.n_seen : (r.flag_cycle ? r.n_cyc : 0)) : 0;
    uint32_t incl = n_list;
    for (int d = 1; d < WAVE; d <<= 1) {
        const uint32_t x = __shfl_up(incl, d, WAVE);
        if (lane >= d) incl += x;
    }
    const uint32_t total = __shfl(incl, WAVE - 1, WAVE);
    unsigned long long base = 0;
    if (total) {
        if (lane == 0) base = atomicAdd(o.pool_head, (unsigned long long)total);
        base = ((unsigned long long)__shfl((uint32_t)(base >> 32), 0, WAVE) << 32) | __shfl((uint32_t)base, 0, WAVE);
    }
    const unsigned long long off = base + (incl - n_list);
    if (done) {
        if (off + n_list <= o.pool_cap)
            for (uint32_t i = 0; i < n_list; ++i) o.pool[off + i] = want_seen ? st.E(i) : st.C(i);
        pf_bfs_record rec;
        rec.entrance = s;
        rec.exit = r.exit_ov;
        rec.n_seen = r.n_seen;
        rec.n_list = n_list;
        rec.list_off = n_list ? off : 0;
        rec.outcome = r.outcome;
        rec.flag_cycle = r.flag_cycle;
        rec.flag_tip = r.flag_tip;
        rec.strict = r.strict;
        rec.pad_ = 0;
        o.rec[c - c0] = rec;
    }
    // the rest goes to the wavefront tier, in candidate order within the wavefront
    const bool over = active && r.overflow;
    const unsigned long long m = __ballot(over);
    if (m) {
        unsigned int b = 0;
        if (lane == 0) b = atomicAdd(n_wave_list, (unsigned int)__popcll(m));
        b = __shfl(b, 0, WAVE);
        if (over) wave_list[b + (unsigned int)__popcll(m & ((1ull << lane) - 1))] = (uint32_t)(c - c0);
    }
}

// wavefront tier: candidates c0 + [0, c1 - c0), or, with a list, the candidates c0 + list[0 .. *n_list)
__global__ __launch_bounds__(256) void k_bfs(const uint32_t *__restrict__ succ, const uint32_t *__restrict__ pred,
                                             const uint32_t *__restrict__ cand, uint64_t c0, uint64_t c1, BfsOut o,
                                             const uint32_t *__restrict__ list, const unsigned int *__restrict__ n_list, uint32_t cap) {
    __shared__ uint32_t s_ent[4][BFS_LDS_CAP];
    __shared__ uint32_t s_todo[4][BFS_LDS_CAP];
    __shared__ uint32_t s_cyc[4][BFS_LDS_CAP];
    __shared__ uint8_t s_meta[4][BFS_LDS_CAP];
    const int wv = threadIdx.x >> 6;
    BfsStore st{s_ent[wv], s_meta[wv], s_todo[wv], s_cyc[wv], cap};   // cap <= BFS_LDS_CAP: where this tier gives a traversal up
    st.live = o.live; st.n_live = o.n_live; st.live_cap = o.live_cap; st.hint_at = o.hint_at;
    const uint64_t wave = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const uint64_t n_waves = ((uint64_t)gridDim.x * blockDim.x) >> 6;
    BfsAlloc al;
    const uint64_t n_items = list ? (uint64_t)*n_list : c1 - c0;
    for (uint64_t it = wave; it < n_items; it += n_waves) {
        const uint64_t c = c0 + (list ? (uint64_t)list[it] : it);
        const uint32_t s = cand[c];
        st.hint_tag = (uint32_t)(c - c0 + 1);
        BfsResult r = bfs_traverse(succ, pred, st, s);
        if (r.overflow) {
            if (lane_id() == 0) {
                pf_bfs_record rec;
                memset(&rec, 0, sizeof(rec));
                rec.entrance = s;
                rec.exit = NONE;
                rec.outcome = BFS_DEFERRED;
                o.rec[c - c0] = rec;
                const unsigned int d = atomicAdd(o.n_deferred, 1u);
                o.deferred[d] = (uint32_t)(c - c0);
                if (o.live && !r.hinted) {   // (a table other than the seen list ran over before the notice went out)
                    const unsigned int l = atomicAdd(o.n_live, 1u);
                    if (l < o.live_cap)
                        __hip_atomic_store(&o.live[l], ((unsigned long long)s << 32) | (unsigned long long)(c - c0 + 1), __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
                }
            }
        } else {
            bfs_emit(o, al, c - c0, s, r, st);
        }
        wave_sync();
    }
}

// K-BFS, big tier: same traversal over per-wave global scratch (BFS_BIG_CAP entries per table).
__global__ __launch_bounds__(64) void k_bfs_big(const uint32_t *__restrict__ succ, const uint32_t *__restrict__ pred,
                                                const uint32_t *__restrict__ cand, uint64_t c0, unsigned int n_deferred,
                                                uint32_t *scratch32, uint8_t *scratch8, BfsOut o) {
    const uint32_t wave = blockIdx.x;
    uint32_t *base = scratch32 + (size_t)wave * 3 * BFS_BIG_CAP;
    BfsStore st{base, scratch8 + (size_t)wave * BFS_BIG_CAP, base + BFS_BIG_CAP, base + 2 * BFS_BIG_CAP, BFS_BIG_CAP};
    BfsAlloc al;
    for (unsigned int d = wave; d < n_deferred; d += gridDim.x) {
        const uint32_t ci = o.deferred[d];
        const uint32_t s = cand[c0 + ci];
        BfsResult r = bfs_traverse(succ, pred, st, s);
        if (r.overflow) {
            if (lane_id() == 0) {
                const unsigned int d2 = atomicAdd(o.n_deferred2, 1u);
                o.deferred2[d2] = ci;
            }
        } else {
            bfs_emit(o, al, ci, s, r, st);
        }
        wave_sync();
    }
}

// K-BFS, last tier: direct-indexed state (pf_bfs_huge.hpp), one candidate per wave.
// two-hop rows for the huge tier: the predecessor rows of the four successors of every oriented vertex
__global__ void k_pred16(const uint32_t *__restrict__ succ, const uint32_t *__restrict__ pred, uint32_t n_ov,
                         uint32_t *__restrict__ pred16) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (; i < (uint64_t)n_ov * 4; i += stride) {
        const uint32_t sv = succ[i];
        uint4 row;
        row.x = row.y = row.z = row.w = NONE;
        if (sv != NONE) row = *reinterpret_cast<const uint4 *>(pred + (size_t)sv * 4);
        *reinterpret_cast<uint4 *>(pred16 + i * 4) = row;
    }
}

__global__ __launch_bounds__(64) void k_bfs_huge(const uint32_t *__restrict__ succ, const uint32_t *__restrict__ pred,
                                                 const uint32_t *__restrict__ cand, uint64_t c0, unsigned int n_deferred2,
                                                 uint32_t *scratch, uint32_t n_unitigs, BfsOut o) {
    const uint32_t wave = blockIdx.x;
    const size_t N = n_unitigs;
    uint32_t *b = scratch + (size_t)wave * (9 * N + 16);
    HugeStore st{b, b + N, b + 2 * N, b + 4 * N, b + 5 * N + 4, b + 7 * N + 12, n_unitigs};
    BfsStore view{st.seen, nullptr, st.todo, st.cyc, n_unitigs};
    BfsAlloc al;
    uint32_t epoch = 0;
    for (unsigned int d = wave; d < n_deferred2; d += gridDim.x) {
        const uint32_t ci = o.deferred2[d];
        const uint32_t s = cand[c0 + ci];
        ++epoch;
        BfsResult r = bfs_traverse_huge(succ, pred, st, epoch, s);
        if (r.overflow) {
            if (lane_id() == 0) o.rec[ci].outcome = BFS_TOO_LARGE;
        } else {
            bfs_emit(o, al, ci, s, r, view);
        }
        wave_sync();
    }
}

// ------------------------------------------------------------------------------------------
// host side of the C ABI
// ------------------------------------------------------------------------------------------
namespace pf {

static const char *kKernelNames[PF_K_COUNT_] = {"k_table_build", "k_adj_insert", "k_adj_probe", "k_cov", "k_bfs",
                                                "k_bfs_big",     "k_align",      "k_align_big", "k_strcov",    "k_bubble",
                                                "k_bubble_big",  "k_cov_colored", "k_strcov_colored", "k_gmm", "k_kmc_decode", "k_minz_count", "k_cov_join",
                                                "k_call_sides", "k_call_prep", "k_call_paths", "k_call_sites", "k_call_format", "k_call_snp", "k_bfs_thread", "k_call_pair", "k_call_stack", "k_cov_join_rest", "copy_text_to_host"};

// (events come from a pool that pf_reset_timing refills: creating two per launch cost more than the launch)
static size_t launch_push(pf_ctx *ctx, int kernel, hipStream_t stream) {
    if (!((ctx->timing_mask >> kernel) & 1ull)) return (size_t)-1;
    TimedLaunch tl;
    tl.kernel = kernel;
    tl.a = tl.b = nullptr;
    {
        std::lock_guard<std::mutex> lk(ctx->launch_mu);
        if (ctx->event_pool.size() >= 2) {
            tl.a = ctx->event_pool.back(); ctx->event_pool.pop_back();
            tl.b = ctx->event_pool.back(); ctx->event_pool.pop_back();
        }
    }
    // (no system-scope fence when an event completes: a launch's timestamps need none, and the fence of twenty events a pass held the
    // kernels on the other streams up -- K-BUBBLE's launches timed cost the pass 1 ms.  PF_EVENT_FENCE=1: plain events, measurements)
    constexpr unsigned ev_flags = hipEventDisableSystemFence;   // (a system-scope fence at every event cost a pass 1.5 ms: profiles/r16_experiments.txt)
    if (!tl.a && (hipEventCreateWithFlags(&tl.a, ev_flags) != hipSuccess || hipEventCreateWithFlags(&tl.b, ev_flags) != hipSuccess)) return (size_t)-1;
    hipEventRecord(tl.a, stream);
    tl.closed = false;
    std::lock_guard<std::mutex> lk(ctx->launch_mu);
    ctx->launches.push_back(tl);
    return ctx->launches.size() - 1;
}
static void launch_close(pf_ctx *ctx, size_t at, hipStream_t stream) {
    hipEvent_t b;
    {
        std::lock_guard<std::mutex> lk(ctx->launch_mu);
        if (at >= ctx->launches.size()) return;
        b = ctx->launches[at].b;
        ctx->launches[at].closed = true;
    }
    hipEventRecord(b, stream);
}
int ctx_begin(pf_ctx *ctx, int kernel) {
    if (!ctx->timing) return 0;
    ctx->main_launch = launch_push(ctx, kernel, ctx->stream);
    return ctx->main_launch == (size_t)-1;
}
void ctx_end(pf_ctx *ctx) {
    if (!ctx->timing) return;
    launch_close(ctx, ctx->main_launch, ctx->stream);
}
// a launch on a side stream: its pair of events is kept aside until ctx_end_on, so that launches on the main stream may be
// bracketed in between
int ctx_begin_on(pf_ctx *ctx, int kernel, hipStream_t stream) {
    if (!ctx->timing) return 0;
    ctx->side_launch = launch_push(ctx, kernel, stream);
    return ctx->side_launch == (size_t)-1;
}
void ctx_end_on(pf_ctx *ctx, hipStream_t stream) {
    if (!ctx->timing) return;
    launch_close(ctx, ctx->side_launch, stream);
    ctx->side_launch = (size_t)-1;
}
int ctx_begin_at(pf_ctx *ctx, int kernel, hipStream_t stream, size_t *at) {
    *at = (size_t)-1;
    if (!ctx->timing) return 0;
    *at = launch_push(ctx, kernel, stream);
    return *at == (size_t)-1;
}
void ctx_end_at(pf_ctx *ctx, size_t at, hipStream_t stream) {
    if (!ctx->timing) return;
    launch_close(ctx, at, stream);
}

void *ctx_ws(pf_ctx *ctx, int slot, size_t bytes) {
    {
        std::lock_guard<std::mutex> lk(ctx->launch_mu);   // (pf_call_align_lane on two lanes may ask for their first workspace at once)
        if (ctx->ws.size() < (size_t)WS_COUNT_) ctx->ws.resize(WS_COUNT_, {nullptr, 0});
    }
    auto &w = ctx->ws[slot];
    if (w.second >= bytes && w.first) return w.first;
    if (w.first) { hipFree(w.first); w.first = nullptr; w.second = 0; }
    const size_t want = bytes + bytes / 4 + 256;
    void *p = nullptr;
    if (hipMalloc(&p, want) != hipSuccess) { pf::CtxErr{ctx} = "hipMalloc of a workspace failed"; return nullptr; }
    w.first = p;
    w.second = want;
    return p;
}

hipError_t lane_stream_create(hipStream_t *s, int lane) {
    if (lane == 0) return hipStreamCreateWithFlags(s, hipStreamNonBlocking);
    int least = 0, greatest = 0;
    const hipError_t e = hipDeviceGetStreamPriorityRange(&least, &greatest);
    if (e != hipSuccess) return e;
    return hipStreamCreateWithPriority(s, hipStreamNonBlocking, least);
}

int ctx_grid(const pf_ctx *ctx, uint64_t work_items, int block, int per_cu) {
    uint64_t want = (work_items + block - 1) / block;
    uint64_t cap = (uint64_t)ctx->n_cu * per_cu;
    if (want < 1) want = 1;
    return (int)std::min<uint64_t>(want, cap);
}

}  // namespace pf

// Launch of the streaming coverage kernel (pf_cov_stream.hpp) for K-COV and K-COV-C: one wavefront per window.
namespace pf {
int launch_cov_stream(pf_ctx *ctx, Kc4Args a, uint32_t n_colors, bool wide, bool colored) {
    const uint64_t n_sr = a.sr_end - a.sr_begin;
    // four wavefronts per block: the hardware hands out blocks as wavefronts retire (a capped grid-stride launch leaves 3.7
    // windows per wavefront at 1 M unitigs: a quarter of the chip idle in the last round)
    const dim3 grid((unsigned)(((n_sr + KC4_SR - 1) / KC4_SR + 3) / 4), n_colors);
    if (colored) {
        if (wide) k_cov_stream4<true, true><<<grid, 256, 0, ctx->stream>>>(a);
        else k_cov_stream4<false, true><<<grid, 256, 0, ctx->stream>>>(a);
    } else {
        if (wide) k_cov_stream4<true, false><<<grid, 256, 0, ctx->stream>>>(a);
        else k_cov_stream4<false, false><<<grid, 256, 0, ctx->stream>>>(a);
    }
    PF_HIP(hipGetLastError());
    return PF_OK;
}
}  // namespace pf

// K-COV-JOIN: hash join of the graph's k-mers with the count table, once per (graph, database): every graph k-mer's canonical
// count (K2 + K3 composite, filter [min_count, max_count] applied at table build) lands at its position in graph order.
// This is the load-time counterpart of K-ADJ: what CDBG::readCov looks up k-mer by k-mer (src/CDBG.cpp:66-120) becomes a
// per-k-mer coverage SoA next to the 2-bit sequence SoA.  A count equal to the marker value cannot be represented: such a
// database (max_count = 2^32 - 1) keeps the probing K-COV.
// begin / finish: the kernels go to a stream of their own (lowest priority: what else the context launches goes first), so that a
// caller with other work for the device -- findSuperBubble does not read a coverage -- runs it beside the look-ups;
// join_finish() is what every reader of the joined array calls first.
namespace pf {
hipStream_t join_stream(pf_ctx *ctx) {
    if (!ctx->join_stream) {
        int least = 0, greatest = 0;
        (void)hipDeviceGetStreamPriorityRange(&least, &greatest);
        if (hipStreamCreateWithPriority(&ctx->join_stream, hipStreamNonBlocking, least) != hipSuccess) { (void)hipGetLastError(); ctx->join_stream = nullptr; }
    }
    return ctx->join_stream ? ctx->join_stream : ctx->stream;
}

int join_graph_counts_begin(pf_ctx *ctx) {
    ctx->gcov_valid = false;
    ctx->join_inflight = false;
    if (!ctx->d_seq || !ctx->d_tab || ctx->tab_exact || ctx->tab_max_count >= GCOV_MISSING || ctx->n_kmers == 0) return PF_OK;
    PF_HIP(hipSetDevice(ctx->device));
    if (!ctx->d_gcov && hipMalloc(&ctx->d_gcov, ((ctx->n_krow + 4) * 64) * sizeof(uint32_t)) != hipSuccess) {  // whole super-rows of 256
        (void)hipGetLastError();  // no room for the SoA: pf_unitig_cov keeps probing the table every pass (same results)
        ctx->d_gcov = nullptr;
        return PF_OK;
    }
    // a wavefront takes JOIN_ROWS_PER_WAVE consecutive rows (the pipeline's four iterations of filling and draining: 6 %) and has its
    // slice of the hand-over list -- three in ten of its k-mers (one in six is expected)
    // a wavefront takes JOIN_ROWS_PER_WAVE consecutive rows (the pipeline's four iterations of filling and draining: 6 %) -- a tenth
    // of a millisecond, so that kernels of other streams get the CUs it gives back: the join runs beside findSuperBubble's -- and
    // has its slice of the hand-over list: four in ten of its k-mers (one in six is expected)
    constexpr uint32_t JOIN_ROWS_PER_WAVE = 64;
    const uint64_t n_rows = (ctx->n_kmers + 63) / 64;
    const uint64_t n_waves = ((n_rows + JOIN_ROWS_PER_WAVE - 1) / JOIN_ROWS_PER_WAVE + 3) / 4 * 4;
    const int blocks = (int)(n_waves / 4);
    // a slice holds half of the k-mers of its wavefront's rows (a sixth is what the average wavefront hands on).  Slices of four
    // tenths made one wavefront in some dozens look its rows up a second time -- 0.3 of the join's 4.4 ms; slices of five, six, eight,
    // ten tenths measure alike (4.05 - 4.1 ms), and a list of every k-mer (3.7 GB) is a tenth of a second of hipMalloc and process
    // exit that a run of one pass does not get back.
    const uint32_t rest_cap = JOIN_ROWS_PER_WAVE * 64 / 2;
    JoinRest *rest = (JoinRest *)ctx_ws(ctx, WS_JOIN_REST, n_waves * (rest_cap + 1) * sizeof(JoinRest));   // (+ 1: the slot the lanes with nothing to hand on write to)
    uint32_t *rest_n = (uint32_t *)ctx_ws(ctx, WS_JOIN_REST_N, n_waves * 4);
    if (!rest || !rest_n) { pf::CtxErr{ctx} = "no room for K-COV-JOIN's hand-over list"; return PF_ERR_HIP; }
    if (!ctx->join_done) PF_HIP(hipEventCreateWithFlags(&ctx->join_done, hipEventDisableTiming));
    const hipStream_t st = join_stream(ctx);
    // (the join's stream does not wait for the context's by itself: what is resident was complete when its upload returned)
    size_t at = 0;
    (void)ctx_begin_at(ctx, PF_K_COV_JOIN, st, &at);
#define PF_JOIN_LAUNCH(KK) k_cov_join<KK><<<blocks, 256, 0, st>>>(ctx->d_tab, ctx->tab_cap - 1, ctx->k, ctx->d_seq, ctx->d_off, ctx->d_kpre, ctx->d_khead, \
        ctx->d_krow, ctx->tab_one_strand, ctx->n_kmers, ctx->d_gcov, rest, rest_n, rest_cap, JOIN_ROWS_PER_WAVE)
    if (ctx->k == 25) PF_JOIN_LAUNCH(25);
    else if (ctx->k == 31) PF_JOIN_LAUNCH(31);
    else PF_JOIN_LAUNCH(0);
#undef PF_JOIN_LAUNCH
    ctx_end_at(ctx, at, st);
    (void)ctx_begin_at(ctx, PF_K_COV_JOIN_REST, st, &at);
    k_cov_join_rest<<<blocks, 256, 0, st>>>(ctx->d_tab, ctx->tab_cap - 1, ctx->k, ctx->tab_one_strand, rest, rest_n, rest_cap, ctx->d_gcov, ctx->d_seq, ctx->d_off,
                                            ctx->d_kpre, ctx->d_khead, ctx->d_krow, ctx->n_kmers, JOIN_ROWS_PER_WAVE);
    ctx_end_at(ctx, at, st);
    PF_HIP(hipGetLastError());
    PF_HIP(hipEventRecord(ctx->join_done, st));
    ctx->join_inflight = true;
    return PF_OK;
}

int join_finish(pf_ctx *ctx) {
    if (!ctx->join_inflight) return PF_OK;
    ctx->join_inflight = false;
    PF_HIP(hipSetDevice(ctx->device));
    PF_HIP(hipEventSynchronize(ctx->join_done));
    ctx->gcov_valid = true;
    return PF_OK;
}

int join_graph_counts(pf_ctx *ctx) {
    const int rc = join_graph_counts_begin(ctx);
    return rc ? rc : join_finish(ctx);
}
}  // namespace pf

// helper: device staging of an input that may live on the host
template <typename T>
static int stage_in(pf_ctx *ctx, const T *src, size_t n, T **dev, bool *owned) {
    hipPointerAttribute_t at;
    *owned = false;
    if (n == 0) { *dev = nullptr; return PF_OK; }
    if (hipPointerGetAttributes(&at, src) == hipSuccess && at.type == hipMemoryTypeDevice) {
        *dev = const_cast<T *>(src);
        return PF_OK;
    }
    (void)hipGetLastError();
    PF_HIP(hipMalloc(dev, n * sizeof(T)));
    *owned = true;
    PF_HIP(hipMemcpyAsync(*dev, src, n * sizeof(T), hipMemcpyDefault, ctx->stream));
    return PF_OK;
}


extern "C" {

const char *pf_kernel_name(int kernel) { return (kernel >= 0 && kernel < PF_K_COUNT_) ? kKernelNames[kernel] : "?"; }

const char *pf_last_error(const pf_ctx *ctx) {
    if (!ctx) return g_create_err.c_str();
    // (a copy per calling thread: two calls of the calling pipeline may run side by side on one context, see pf::CtxErr)
    static thread_local std::string mine;
    {
        std::lock_guard<std::mutex> lk(const_cast<pf_ctx *>(ctx)->launch_mu);
        mine = ctx->err;
    }
    return mine.c_str();
}

int pf_warmup(int device) {
    // brings up the HIP runtime and the device's primary context (a tenth of a second): a caller with other work to do first --
    // reading its input files -- runs this on a helper thread and finds pf_create instant later
    if (hipSetDevice(device) != hipSuccess) { (void)hipGetLastError(); return PF_ERR_NO_DEVICE; }
    if (hipFree(nullptr) != hipSuccess) { (void)hipGetLastError(); return PF_ERR_HIP; }
    return PF_OK;
}

int pf_create(int device, pf_ctx **out) {
    if (!out) return PF_ERR_ARG;
    *out = nullptr;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) {
        g_create_err = "no HIP device visible: the PloidyFrost device layer has no CPU fallback";
        return PF_ERR_NO_DEVICE;
    }
    if (device < 0 || device >= n) { g_create_err = "device index out of range"; return PF_ERR_ARG; }
    hipDeviceProp_t prop;
    if (hipSetDevice(device) != hipSuccess || hipGetDeviceProperties(&prop, device) != hipSuccess) {
        g_create_err = "hipSetDevice / hipGetDeviceProperties failed";
        return PF_ERR_HIP;
    }
    if (std::string(prop.gcnArchName).rfind("gfx950", 0) != 0) {
        g_create_err = std::string("device is ") + prop.gcnArchName + ", this library is built for gfx950 only";
        return PF_ERR_NO_DEVICE;
    }
    pf_ctx *ctx = new pf_ctx();
    ctx->device = device;
    ctx->n_cu = prop.multiProcessorCount;
    ctx->name = prop.name;
    if (hipStreamCreateWithFlags(&ctx->own_stream, hipStreamNonBlocking) != hipSuccess) {
        g_create_err = "hipStreamCreate failed";
        delete ctx;
        return PF_ERR_HIP;
    }
    ctx->stream = ctx->own_stream;
    // what a helper thread that takes buffers beside the load (pf_call_reserve*) would otherwise create on its first call, while the
    // loading thread looks at the same fields: the workspace table at its final size and the calling pipeline's state
    ctx->ws.resize(WS_COUNT_, {nullptr, 0});
    call_state_create(ctx);
    *out = ctx;
    return PF_OK;
}

static void free_graph(pf_ctx *ctx) {
    call_invalidate(ctx);
    hipFree(ctx->d_seq); hipFree(ctx->d_off); hipFree(ctx->d_len); hipFree(ctx->d_succ); hipFree(ctx->d_pred);
    hipFree(ctx->d_cand);
    hipFree(ctx->d_pred16);
    hipFree(ctx->d_kpre);
    hipFree(ctx->d_kwin);
    hipFree(ctx->d_gcov); hipFree(ctx->d_khead); hipFree(ctx->d_krow); hipFree(ctx->d_gcov_c);
    ctx->d_gcov = nullptr;
    ctx->d_gcov_c = nullptr;
    ctx->gcov_c_valid = false;
    ctx->d_khead = nullptr;
    ctx->d_krow = nullptr;
    ctx->gcov_valid = false;
    ctx->d_kpre = nullptr;
    ctx->d_kwin = nullptr;
    ctx->d_pred16 = nullptr;
    ctx->d_seq = ctx->d_off = nullptr;
    ctx->d_len = ctx->d_succ = ctx->d_pred = ctx->d_cand = nullptr;
    ctx->h_cand.clear();
    ctx->has_adj = false;
}

void pf_destroy(pf_ctx *ctx) {
    if (!ctx) return;
    hipSetDevice(ctx->device);
    hipStreamSynchronize(ctx->stream);
    if (ctx->join_stream) { hipStreamSynchronize(ctx->join_stream); hipStreamDestroy(ctx->join_stream); }
    if (ctx->join_done) hipEventDestroy(ctx->join_done);
    if (ctx->join_c_done) hipEventDestroy(ctx->join_c_done);
    for (auto &tl : ctx->launches) { hipEventDestroy(tl.a); hipEventDestroy(tl.b); }
    for (auto &e : ctx->event_pool) hipEventDestroy(e);
    call_destroy(ctx);
    cc_destroy(ctx);
    gfa_destroy(ctx);
    comm_destroy(ctx);
    free_graph(ctx);
    hipFree(ctx->d_tab);
    hipFree(ctx->d_ctab);
    hipFree(ctx->d_unread);
    hipFree(ctx->d_cov_sum); hipFree(ctx->d_cov_min); hipFree(ctx->d_cov_miss);
    for (auto &w : ctx->ws) hipFree(w.first);
    if (ctx->h_live) hipHostFree(ctx->h_live);
    if (ctx->copy_stream) { hipStreamSynchronize(ctx->copy_stream); hipStreamDestroy(ctx->copy_stream); }
    for (auto &lane : ctx->bub_streams) for (auto &s : lane) if (s) hipStreamDestroy(s);
    for (auto &lane : ctx->bub_events) for (auto &e : lane) if (e) hipEventDestroy(e);
    hipStreamDestroy(ctx->own_stream);
    delete ctx;
}

int pf_set_stream(pf_ctx *ctx, void *s) {
    if (!ctx) return PF_ERR_ARG;
    ctx->stream = s ? (hipStream_t)s : ctx->own_stream;
    return PF_OK;
}

int pf_synchronize(pf_ctx *ctx) {
    if (!ctx) return PF_ERR_ARG;
    PF_HIP(hipStreamSynchronize(ctx->stream));
    return PF_OK;
}

int pf_enable_timing(pf_ctx *ctx, int on) {
    if (!ctx) return PF_ERR_ARG;
    ctx->timing = on != 0;
    return PF_OK;
}

int pf_timing_select(pf_ctx *ctx, uint64_t kernel_mask) {
    if (!ctx) return PF_ERR_ARG;
    ctx->timing_mask = kernel_mask;
    return PF_OK;
}

int pf_reset_timing(pf_ctx *ctx) {
    if (!ctx) return PF_ERR_ARG;
    PF_HIP(hipSetDevice(ctx->device));
    PF_HIP(hipDeviceSynchronize());   // (launches on every stream of the pipeline)
    std::lock_guard<std::mutex> lk(ctx->launch_mu);
    for (auto &tl : ctx->launches) { ctx->event_pool.push_back(tl.a); ctx->event_pool.push_back(tl.b); }
    ctx->launches.clear();
    memset(ctx->units, 0, sizeof(ctx->units));
    return PF_OK;
}

int pf_kernel_units(pf_ctx *ctx, int kernel, uint64_t *units) {
    if (!ctx || kernel < 0 || kernel >= PF_K_COUNT_ || !units) return PF_ERR_ARG;
    *units = ctx->units[kernel];
    return PF_OK;
}

int pf_kernel_time(pf_ctx *ctx, int kernel, double *total_ms, uint64_t *launches) {
    if (!ctx || kernel < 0 || kernel >= PF_K_COUNT_) return PF_ERR_ARG;
    PF_HIP(hipStreamSynchronize(ctx->stream));
    double tot = 0;
    uint64_t n = 0;
    for (auto &tl : ctx->launches) {
        if (tl.kernel != kernel || !tl.closed) continue;
        float ms = 0;
        PF_HIP(hipEventElapsedTime(&ms, tl.a, tl.b));
        tot += ms;
        n++;
    }
    if (total_ms) *total_ms = tot;
    if (launches) *launches = n;
    return PF_OK;
}

// union of the timed launches' intervals: of every kernel (kernel_mask all ones), or of the kernels of the mask
static int busy_union(pf_ctx *ctx, uint64_t kernel_mask, double *busy_ms, double *span_ms) {
    if (!ctx) return PF_ERR_ARG;
    PF_HIP(hipSetDevice(ctx->device));
    PF_HIP(hipDeviceSynchronize());
    std::vector<std::pair<float, float>> iv;
    size_t unread = 0;
    {
        std::lock_guard<std::mutex> lk(ctx->launch_mu);
        iv.reserve(ctx->launches.size());
        for (auto &tl : ctx->launches) {
            float a = 0, b = 0;
            if (!((kernel_mask >> tl.kernel) & 1ull)) continue;
            if (!tl.closed) { ++unread; continue; }
            if (hipEventElapsedTime(&a, ctx->launches[0].a, tl.a) != hipSuccess || hipEventElapsedTime(&b, ctx->launches[0].a, tl.b) != hipSuccess) {
                // (the device was synchronized above: an event that cannot be read was never recorded -- a launch that failed)
                (void)hipGetLastError();
                ++unread;
                continue;
            }
            if (b > a) iv.emplace_back(a, b);
        }
    }
    std::sort(iv.begin(), iv.end());
    double busy = 0, lo = 0, hi = 0;
    bool open = false;
    for (auto &x : iv) {
        if (!open) { lo = x.first; hi = x.second; open = true; }
        else if (x.first <= hi) hi = std::max<double>(hi, x.second);
        else { busy += hi - lo; lo = x.first; hi = x.second; }
    }
    if (open) busy += hi - lo;
    if (busy_ms) *busy_ms = busy;
    if (span_ms) {
        float mx = 0;
        for (auto &x : iv) mx = std::max(mx, x.second);
        *span_ms = iv.empty() ? 0.0 : (double)mx - (double)iv.front().first;
    }
    if (unread) {
        pf::CtxErr{ctx} = "pf_device_busy: " + std::to_string(unread) + " launch intervals could not be read (busy_ms leaves them out)";
        return PF_ERR_HIP;
    }
    return PF_OK;
}

int pf_device_busy(pf_ctx *ctx, double *busy_ms, double *span_ms) { return busy_union(ctx, ~0ull, busy_ms, span_ms); }
int pf_kernel_busy(pf_ctx *ctx, uint64_t kernel_mask, double *busy_ms) { return busy_union(ctx, kernel_mask, busy_ms, nullptr); }

// Page-locked host memory.  The pages come from the kernel, populated (mmap: no runtime in it), and are then registered with the
// runtime: hipHostMalloc of the few hundred MB a load takes held the runtime's allocation lock for 8 ms per 64 MB while it pinned
// them -- a tenth of a second in which the loading thread's own hipMalloc / hipMemcpy calls stood still -- where populating takes
// half that time outside any lock and registering 0.4 ms per 64 MB inside it (tools/exp/pin_cost.cpp).  Ordinary cached memory on
// the CPU side (the host layer reads these buffers element by element after each copy).
namespace {
std::mutex g_host_mu;
std::vector<std::pair<void *, size_t>> g_host_maps;   // what pf_host_free has to unmap
}  // namespace
int pf_host_alloc(pf_ctx *ctx, size_t bytes, void **out) {
    if (!ctx || !out) return PF_ERR_ARG;
    PF_HIP(hipSetDevice(ctx->device));
    const size_t page = 1u << 21;
    const size_t size = ((bytes ? bytes : 1) + page - 1) & ~(page - 1);
    void *p = mmap(nullptr, size, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_POPULATE, -1, 0);
    if (p != MAP_FAILED) {
        if (hipHostRegister(p, size, hipHostRegisterDefault) == hipSuccess) {
            std::lock_guard<std::mutex> lk(g_host_mu);
            g_host_maps.emplace_back(p, size);
            *out = p;
            return PF_OK;
        }
        (void)hipGetLastError();
        munmap(p, size);
    }
    PF_HIP(hipHostMalloc(out, bytes ? bytes : 1, hipHostMallocNonCoherent));
    return PF_OK;
}

int pf_fetch(pf_ctx *ctx, void *dst_host, const void *src_dev, uint64_t bytes) {
    if (!ctx || (bytes && (!dst_host || !src_dev))) return PF_ERR_ARG;
    PF_HIP(hipSetDevice(ctx->device));
    if (bytes) PF_HIP(hipMemcpy(dst_host, src_dev, bytes, hipMemcpyDeviceToHost));
    return PF_OK;
}

void pf_host_free(pf_ctx *ctx, void *p) {
    (void)ctx;
    if (!p) return;
    size_t size = 0;
    {
        std::lock_guard<std::mutex> lk(g_host_mu);
        for (size_t i = 0; i < g_host_maps.size(); ++i)
            if (g_host_maps[i].first == p) { size = g_host_maps[i].second; g_host_maps.erase(g_host_maps.begin() + (long)i); break; }
    }
    if (size) { (void)hipHostUnregister(p); munmap(p, size); }
    else hipHostFree(p);
}

int pf_device_name(pf_ctx *ctx, char *buf, size_t cap) {
    if (!ctx || !buf || !cap) return PF_ERR_ARG;
    snprintf(buf, cap, "%s", ctx->name.c_str());
    return PF_OK;
}

int pf_device_pci_bus_id(pf_ctx *ctx, char *buf, size_t cap) {
    if (!ctx || !buf || cap < 16) return PF_ERR_ARG;
    PF_HIP(hipDeviceGetPCIBusId(buf, (int)cap, ctx->device));
    return PF_OK;
}

uint64_t pf_table_capacity(const pf_ctx *ctx) { return ctx ? ctx->tab_cap * pf::LINE_KEYS : 0; }
uint64_t pf_num_kmers(const pf_ctx *ctx) { return ctx ? ctx->n_kmers : 0; }

int pf_upload_graph(pf_ctx *ctx, const uint64_t *seq_words, const uint64_t *seq_off, const uint32_t *len_bp,
                    uint32_t N, int k) {
    if (!ctx || !seq_words || !seq_off || !len_bp || N == 0 || k < 3 || k > 31) return PF_ERR_ARG;
    if (N >= (1u << 30)) { pf::CtxErr{ctx} = "more than 2^30 unitigs"; return PF_ERR_ARG; }
    if (ctx->d_tab && ctx->tab_k != k) { pf::CtxErr{ctx} = "pf_upload_graph: k differs from the k of the resident count table"; return PF_ERR_ARG; }
    PF_HIP(hipSetDevice(ctx->device));
    DevLoadTrace trace;
    (void)join_finish(ctx);
    (void)join_colored_finish(ctx);
    free_graph(ctx);
    trace.mark("graph: previous graph released");
    // seq_off[N] = total words; fetch it (host or device pointer)
    uint64_t total_words = 0;
    PF_HIP(hipMemcpy(&total_words, seq_off + N, 8, hipMemcpyDefault));
    trace.mark("graph: size of the sequence array fetched");
    ctx->N = N;
    ctx->k = k;
    ctx->n_words = total_words;
    PF_HIP(hipMalloc(&ctx->d_seq, (total_words + 2) * 8));
    PF_HIP(hipMemsetAsync(ctx->d_seq, 0, (total_words + 2) * 8, ctx->stream));
    PF_HIP(hipMalloc(&ctx->d_off, ((size_t)N + 1) * 8));
    PF_HIP(hipMalloc(&ctx->d_len, (size_t)N * 4));
    trace.mark("graph: arrays allocated");
    PF_HIP(hipMemcpyAsync(ctx->d_seq, seq_words, total_words * 8, hipMemcpyDefault, ctx->stream));
    PF_HIP(hipMemcpyAsync(ctx->d_off, seq_off, ((size_t)N + 1) * 8, hipMemcpyDefault, ctx->stream));
    PF_HIP(hipMemcpyAsync(ctx->d_len, len_bp, (size_t)N * 4, hipMemcpyDefault, ctx->stream));
    PF_HIP(hipStreamSynchronize(ctx->stream));
    trace.mark("graph: arrays copied");
    // validation of what the kernels assume (lengths >= k, offsets cover lengths) and the k-mer numbering of the k-mer-parallel
    // kernels, all on the device: d_kpre = exclusive scan of the k-mers per unitig; d_kwin[w] / d_krow[r] = the unitig holding
    // k-mer 256 w / 64 r; d_khead = one bit per k-mer (unitig starts, plus the end of the last one)
    {
        DevTmp<uint64_t> cnt_;
        DevTmp<unsigned int> bad_;
        PF_HIP(cnt_.alloc(((size_t)N + 1) * 8));
        PF_HIP(bad_.alloc(4));
        PF_HIP(hipMemsetAsync(bad_.p, 0, 4, ctx->stream));
        k_graph_check<<<(N + 1 + 255) / 256, 256, 0, ctx->stream>>>(ctx->d_off, ctx->d_len, N, k, total_words, cnt_.p, bad_.p);
        PF_HIP(hipMalloc(&ctx->d_kpre, ((size_t)N + 1) * 8));
        DevTmp<uint8_t> tmp_;
        PF_HIP(tmp_.alloc(scan_scratch_bytes((uint64_t)N + 1)));
        PF_HIP(scan_exclusive_u64(cnt_.p, ctx->d_kpre, (uint64_t)N + 1, tmp_.p, ctx->stream));
        unsigned int bad = 0;
        uint64_t nk = 0;
        PF_HIP(hipMemcpyAsync(&bad, bad_.p, 4, hipMemcpyDeviceToHost, ctx->stream));
        PF_HIP(hipMemcpyAsync(&nk, ctx->d_kpre + N, 8, hipMemcpyDeviceToHost, ctx->stream));
        PF_HIP(hipStreamSynchronize(ctx->stream));
        if (bad) {
            pf::CtxErr{ctx} = (bad & 1) ? "unitig shorter than k" : "seq_off does not cover len_bp";
            free_graph(ctx);
            return PF_ERR_ARG;
        }
        trace.mark("graph: checked, k-mers numbered");
        ctx->n_kmers = nk;
        const uint64_t n_win = nk / 256 + 1;
        const uint64_t n_row = nk / 64 + 1, n_row_pad = n_row + 8;  // padded: the four-per-lane kernel reads whole super-rows (4 words)
        PF_HIP(hipMalloc(&ctx->d_kwin, n_win * 4));
        PF_HIP(hipMalloc(&ctx->d_khead, n_row_pad * 8));
        PF_HIP(hipMalloc(&ctx->d_krow, n_row_pad * 4));
        PF_HIP(hipMemsetAsync(ctx->d_khead, 0, n_row_pad * 8, ctx->stream));
        k_fill_u32<<<ctx_grid(ctx, n_win, 256, 8), 256, 0, ctx->stream>>>(ctx->d_kwin, n_win, N - 1);
        k_fill_u32<<<ctx_grid(ctx, n_row_pad, 256, 8), 256, 0, ctx->stream>>>(ctx->d_krow, n_row_pad, N - 1);
        k_graph_index<<<(N + 1 + 255) / 256, 256, 0, ctx->stream>>>(ctx->d_kpre, N, n_win, n_row, ctx->d_kwin, ctx->d_krow, ctx->d_khead);
        PF_HIP(hipGetLastError());
        ctx->n_kwin = n_win;
        ctx->n_krow = n_row;
        // host copy of the lengths (argument validation of later calls)
        ctx->h_len.resize(N);
        PF_HIP(hipMemcpyAsync(ctx->h_len.data(), ctx->d_len, (size_t)N * 4, hipMemcpyDeviceToHost, ctx->stream));
        PF_HIP(hipStreamSynchronize(ctx->stream));
        trace.mark("graph: row index, lengths to the host");
    }
    if (ctx->d_tab && !ctx->tab_exact) {
        const int js = pf::join_graph_counts(ctx);
        trace.mark("graph: first join with the count table");
        return js;
    }
    return PF_OK;
}

int pf_build_adjacency(pf_ctx *ctx, uint32_t *succ, uint32_t *pred) {
    if (!ctx || !ctx->d_seq) return PF_ERR_ARG;
    PF_HIP(hipSetDevice(ctx->device));
    const uint32_t N = ctx->N;
    uint64_t cap = 1;
    while (cap < (uint64_t)N * 4) cap <<= 1;
    DevTmp<Slot> ends_;
    PF_HIP(ends_.alloc(cap * sizeof(Slot)));
    Slot *ends = ends_.p;
    if (!ctx->d_succ) PF_HIP(hipMalloc(&ctx->d_succ, (size_t)N * 8 * 4));
    if (!ctx->d_pred) PF_HIP(hipMalloc(&ctx->d_pred, (size_t)N * 8 * 4));
    k_fill_slots<<<ctx_grid(ctx, cap, 256, 8), 256, 0, ctx->stream>>>(ends, cap);
    ctx_begin(ctx, PF_K_ADJ_INSERT);
    k_adj_insert<<<ctx_grid(ctx, N, 256, 8), 256, 0, ctx->stream>>>(ends, cap - 1, ctx->d_seq, ctx->d_off, ctx->d_len, N, ctx->k);
    ctx_end(ctx);
    ctx_begin(ctx, PF_K_ADJ_PROBE);
    k_adj_probe<<<ctx_grid(ctx, (uint64_t)N * 2, 256, 8), 256, 0, ctx->stream>>>(ends, cap - 1, ctx->d_seq, ctx->d_off,
                                                                               ctx->d_len, N, ctx->k, ctx->d_succ, ctx->d_pred);
    ctx_end(ctx);
    // candidate list: oriented vertices with out-degree > 1, ascending
    DevTmp<uint8_t> flag_;
    DevTmp<uint32_t> num_;
    PF_HIP(flag_.alloc((size_t)N * 2));
    PF_HIP(num_.alloc(4));
    uint8_t *flag = flag_.p;
    uint32_t *d_num = num_.p;
    hipFree(ctx->d_cand);
    ctx->d_cand = nullptr;
    hipFree(ctx->d_pred16);  // stale two-hop rows of a previous adjacency
    ctx->d_pred16 = nullptr;
    PF_HIP(hipMalloc(&ctx->d_cand, (size_t)N * 2 * 4));
    k_mark_candidates<<<ctx_grid(ctx, (uint64_t)N * 2, 256, 8), 256, 0, ctx->stream>>>(ctx->d_succ, N * 2, flag);
    {
        DevTmp<uint8_t> tmp_;
        PF_HIP(tmp_.alloc(scan_scratch_bytes((uint64_t)N * 2)));
        PF_HIP(select_flagged_u8(flag, ctx->d_cand, d_num, nullptr, (uint64_t)N * 2, tmp_.p, ctx->stream));
        PF_HIP(hipStreamSynchronize(ctx->stream));
    }
    uint32_t n_cand = 0;
    PF_HIP(hipMemcpy(&n_cand, d_num, 4, hipMemcpyDeviceToHost));
    ctx->h_cand.resize(n_cand);
    if (n_cand) PF_HIP(hipMemcpy(ctx->h_cand.data(), ctx->d_cand, (size_t)n_cand * 4, hipMemcpyDeviceToHost));
    ctx->has_adj = true;
    if (succ) PF_HIP(hipMemcpyAsync(succ, ctx->d_succ, (size_t)N * 8 * 4, hipMemcpyDefault, ctx->stream));
    if (pred) PF_HIP(hipMemcpyAsync(pred, ctx->d_pred, (size_t)N * 8 * 4, hipMemcpyDefault, ctx->stream));
    PF_HIP(hipStreamSynchronize(ctx->stream));
    return PF_OK;
}

int pf_kmc_decode(pf_ctx *ctx, const uint8_t *records, uint64_t n_records, uint32_t suffix_bytes, uint32_t counter_bytes,
                  const uint64_t *lut, uint64_t n_lut, uint32_t lut_prefix_len, uint32_t k, uint64_t **kmers_dev, uint32_t **counts_dev) {
    if (!ctx || !kmers_dev || !counts_dev || (n_records && (!records || !lut)) || n_lut == 0 || lut_prefix_len == 0 || lut_prefix_len > 15 ||
        k > 31 || lut_prefix_len >= k || (k - lut_prefix_len) != suffix_bytes * 4 || counter_bytes == 0 || counter_bytes > 8) {
        if (ctx) pf::CtxErr{ctx} = "pf_kmc_decode: inconsistent k / lut_prefix_len / suffix_bytes / counter_bytes";
        return PF_ERR_ARG;
    }
    *kmers_dev = nullptr;
    *counts_dev = nullptr;
    PF_HIP(hipSetDevice(ctx->device));
    if (lut[0] != 0 || lut[n_lut] != n_records) { pf::CtxErr{ctx} = "pf_kmc_decode: lut must start at 0 and end (entry n_lut) at n_records"; return PF_ERR_ARG; }
    for (uint64_t e = 0; e < n_lut; ++e)
        if (lut[e] > lut[e + 1]) { pf::CtxErr{ctx} = "pf_kmc_decode: lut is not non-decreasing"; return PF_ERR_ARG; }
    DevTmp<uint8_t> drec;
    DevTmp<uint64_t> dlut;
    uint64_t *dk = nullptr;
    uint32_t *dc = nullptr;
    const size_t rec_bytes = (size_t)n_records * (suffix_bytes + counter_bytes);
    PF_HIP(drec.alloc(rec_bytes));
    PF_HIP(dlut.alloc((n_lut + 1) * 8));
    PF_HIP(hipMalloc(&dk, n_records ? n_records * 8 : 8));
    if (hipMalloc(&dc, n_records ? n_records * 4 : 4) != hipSuccess) { hipFree(dk); pf::CtxErr{ctx} = "hipMalloc of the decoded counts failed"; return PF_ERR_HIP; }
    hipError_t e1 = hipMemcpyAsync(drec.p, records, rec_bytes, hipMemcpyDefault, ctx->stream);
    hipError_t e2 = hipMemcpyAsync(dlut.p, lut, (n_lut + 1) * 8, hipMemcpyDefault, ctx->stream);
    if (e1 == hipSuccess && e2 == hipSuccess && n_records) {
        ctx_begin(ctx, PF_K_KMC_DECODE);
        k_kmc_decode<<<ctx_grid(ctx, (n_records + KMC_PER_THREAD - 1) / KMC_PER_THREAD, 256, 8), 256, 0, ctx->stream>>>(
            drec.p, n_records, suffix_bytes, counter_bytes, dlut.p, n_lut, (1u << (2 * lut_prefix_len)) - 1, 8 * suffix_bytes, dk, dc);
        ctx_end(ctx);
        e1 = hipGetLastError();
    }
    if (e1 == hipSuccess) e1 = e2;
    if (e1 == hipSuccess) e1 = hipStreamSynchronize(ctx->stream);
    if (e1 != hipSuccess) {
        hipFree(dk);
        hipFree(dc);
        pf::CtxErr{ctx} = std::string("pf_kmc_decode: ") + hipGetErrorString(e1);
        return PF_ERR_HIP;
    }
    *kmers_dev = dk;
    *counts_dev = dc;
    return PF_OK;
}

void pf_device_free(pf_ctx *ctx, void *p) {
    if (!p) return;
    if (ctx) (void)hipSetDevice(ctx->device);
    (void)hipFree(p);
}

int pf_copy_to_host(pf_ctx *ctx, void *dst, const void *src_dev, size_t bytes) {
    if (!ctx || (bytes && (!dst || !src_dev))) return PF_ERR_ARG;
    PF_HIP(hipSetDevice(ctx->device));
    PF_HIP(hipMemcpyAsync(dst, src_dev, bytes, hipMemcpyDefault, ctx->stream));
    PF_HIP(hipStreamSynchronize(ctx->stream));
    return PF_OK;
}

// one orientation per k-mer?  (decides the probe order of the composite lookup, never its result)  A table whose keys are all
// canonical -- what K-TABLE saw while it hashed them -- cannot hold a key's reverse complement; only a table with other keys is
// searched for such a pair (233 M probes saved for every database written with canonical counting).
static int check_table_strands(pf_ctx *ctx, bool keys_all_canonical) {
    ctx->tab_one_strand = false;
    if (!ctx->d_tab || !ctx->tab_n) return PF_OK;
    if (keys_all_canonical) { ctx->tab_one_strand = true; return PF_OK; }
    DevTmp<unsigned int> flag_;
    unsigned int h_flag = 0;
    PF_HIP(flag_.alloc(4));
    PF_HIP(hipMemsetAsync(flag_.p, 0, 4, ctx->stream));
    k_table_two_strands<<<ctx_grid(ctx, ctx->tab_cap * LINE_KEYS, 256, 8), 256, 0, ctx->stream>>>(ctx->d_tab, ctx->tab_cap, ctx->tab_k, flag_.p);
    PF_HIP(hipMemcpyAsync(&h_flag, flag_.p, 4, hipMemcpyDeviceToHost, ctx->stream));
    PF_HIP(hipStreamSynchronize(ctx->stream));
    ctx->tab_one_strand = h_flag == 0;
    return PF_OK;
}

int pf_upload_counts(pf_ctx *ctx, const uint64_t *kmers, const uint32_t *counts, uint64_t n, uint32_t k, uint64_t min_count,
                     uint64_t max_count, int both_strands) {
    if (!ctx || (n && (!kmers || !counts)) || k < 3 || k > 31) return PF_ERR_ARG;
    if (ctx->d_seq && (int)k != ctx->k) { pf::CtxErr{ctx} = "pf_upload_counts: k differs from the k of the resident graph"; return PF_ERR_ARG; }
    (void)join_finish(ctx);
    ctx->tab_exact = !both_strands;   // GetBothStrands() == false: lookups take the k-mer as it reads (src/CDBG.cpp:94-117)
    ctx->tab_max_count = max_count;
    call_invalidate(ctx);
    PF_HIP(hipSetDevice(ctx->device));
    hipFree(ctx->d_tab);
    ctx->d_tab = nullptr;
    hipFree(ctx->d_gcov);
    ctx->d_gcov = nullptr;
    ctx->gcov_valid = false;
    // lines of ten keys, 2 to 4 of them taken on average: a line overflows for one key in ten
    uint64_t cap = 128;
    while (cap * 4 < n) cap <<= 1;
    PF_HIP(hipMalloc(&ctx->d_tab, cap * sizeof(CountLine)));
    ctx->tab_cap = cap;
    ctx->tab_n = n;
    ctx->tab_k = (int)k;
    PF_HIP(hipMemsetAsync(ctx->d_tab, 0xFF, cap * sizeof(CountLine), ctx->stream));   // every key free
    // stage the records on the device if the caller passed host memory
    hipPointerAttribute_t at;
    const bool on_dev = n && hipPointerGetAttributes(&at, kmers) == hipSuccess && at.type == hipMemoryTypeDevice;
    (void)hipGetLastError();
    DevTmp<uint64_t> dk_;
    DevTmp<uint32_t> dc_;
    const uint64_t *pk = kmers;
    const uint32_t *pc = counts;
    if (!on_dev && n) {
        PF_HIP(dk_.alloc(n * 8));
        PF_HIP(dc_.alloc(n * 4));
        PF_HIP(hipMemcpyAsync(dk_.p, kmers, n * 8, hipMemcpyDefault, ctx->stream));
        PF_HIP(hipMemcpyAsync(dc_.p, counts, n * 4, hipMemcpyDefault, ctx->stream));
        pk = dk_.p;
        pc = dc_.p;
    }
    DevTmp<unsigned int> noncanon_;
    unsigned int h_noncanon = 0;
    PF_HIP(noncanon_.alloc(4));
    PF_HIP(hipMemsetAsync(noncanon_.p, 0, 4, ctx->stream));
    if (n) {
        ctx_begin(ctx, PF_K_TABLE_BUILD);
        k_table_build<<<ctx_grid(ctx, n, 256, 8), 256, 0, ctx->stream>>>(ctx->d_tab, cap - 1, (int)k, pk, pc, n, min_count, max_count, noncanon_.p);
        ctx_end(ctx);
    }
    PF_HIP(hipMemcpyAsync(&h_noncanon, noncanon_.p, 4, hipMemcpyDeviceToHost, ctx->stream));
    PF_HIP(hipStreamSynchronize(ctx->stream));
    const int rc = check_table_strands(ctx, h_noncanon == 0);
    if (rc) return rc;
    if (ctx->d_seq && !ctx->tab_exact) return pf::join_graph_counts(ctx);
    return PF_OK;
}

int pf_join_counts_begin(pf_ctx *ctx) {
    if (!ctx || !ctx->d_seq) return PF_ERR_ARG;
    const bool single = ctx->d_tab && !ctx->tab_exact, colored = ctx->d_ctab && ctx->n_colors;
    if (!single && !colored) { pf::CtxErr{ctx} = "pf_join_counts: no canonical count table resident"; return PF_ERR_ARG; }
    { const int rc = pf::join_finish(ctx); if (rc) return rc; }
    { const int rc = pf::join_colored_finish(ctx); if (rc) return rc; }
    if (single) {
        const int rc = pf::join_graph_counts_begin(ctx);
        if (rc) return rc;
        if (!ctx->join_inflight) { pf::CtxErr{ctx} = "pf_join_counts: no joined coverage array for this table (max_count = 2^32 - 1, or no room)"; return PF_ERR_ARG; }
    }
    if (colored) {
        const int rc = pf::join_graph_counts_colored_begin(ctx);
        if (rc) return rc;
        if (!ctx->join_c_inflight) { pf::CtxErr{ctx} = "pf_join_counts: no joined coverage array for these databases (max_count = 2^32 - 1, or no room)"; return PF_ERR_ARG; }
    }
    return PF_OK;
}

int pf_join_counts_end(pf_ctx *ctx) {
    if (!ctx) return PF_ERR_ARG;
    const int rc = pf::join_finish(ctx);
    return rc ? rc : pf::join_colored_finish(ctx);
}

int pf_join_counts(pf_ctx *ctx) {
    const int rc = pf_join_counts_begin(ctx);
    return rc ? rc : pf_join_counts_end(ctx);
}

int pf_lookup_kmers(pf_ctx *ctx, const uint64_t *kmers, uint64_t n, uint32_t *counts, uint8_t *found) {
    if (!ctx || !ctx->d_tab || !kmers || !counts || !found) return PF_ERR_ARG;
    if (n == 0) return PF_OK;
    PF_HIP(hipSetDevice(ctx->device));
    uint64_t *dk;
    bool own;
    int rc = stage_in(ctx, kmers, n, &dk, &own);
    if (rc) return rc;
    DevTmp<uint64_t> own_;
    if (own) own_.p = dk;  // freed on every return path
    DevTmp<uint32_t> dc_;
    DevTmp<uint8_t> df_;
    PF_HIP(dc_.alloc(n * 4));
    PF_HIP(df_.alloc(n));
    uint32_t *dc = dc_.p;
    uint8_t *df = df_.p;
    k_lookup<<<ctx_grid(ctx, n, 256, 8), 256, 0, ctx->stream>>>(ctx->d_tab, ctx->tab_cap - 1, ctx->tab_k, ctx->tab_one_strand, dk, n, dc, df);
    PF_HIP(hipMemcpyAsync(counts, dc, n * 4, hipMemcpyDefault, ctx->stream));
    PF_HIP(hipMemcpyAsync(found, df, n, hipMemcpyDefault, ctx->stream));
    PF_HIP(hipStreamSynchronize(ctx->stream));
    return PF_OK;
}

static int unitig_cov_impl(pf_ctx *ctx, uint32_t u0, uint32_t u1, int exact, uint64_t *sum, uint32_t *mn, uint8_t *miss, bool probe = false) {
    if (!ctx || !ctx->d_seq || !ctx->d_tab || u0 > u1 || u1 > ctx->N || !sum || !mn || !miss) return PF_ERR_ARG;
    if (u0 == u1) return PF_OK;
    PF_HIP(hipSetDevice(ctx->device));
    const uint32_t n = u1 - u0;
    // outputs: write straight into device memory when the caller gave device pointers
    hipPointerAttribute_t at;
    const bool dev_out = hipPointerGetAttributes(&at, sum) == hipSuccess && at.type == hipMemoryTypeDevice;
    (void)hipGetLastError();
    uint64_t *ds = sum;
    uint32_t *dm = mn;
    uint8_t *dx = miss;
    if (!dev_out) {
        if (ctx->cov_cap < n) {
            hipFree(ctx->d_cov_sum); hipFree(ctx->d_cov_min); hipFree(ctx->d_cov_miss);
            ctx->d_cov_sum = nullptr; ctx->d_cov_min = nullptr; ctx->d_cov_miss = nullptr;
            PF_HIP(hipMalloc(&ctx->d_cov_sum, (size_t)n * 8));
            PF_HIP(hipMalloc(&ctx->d_cov_min, (size_t)n * 4));
            PF_HIP(hipMalloc(&ctx->d_cov_miss, (size_t)n));
            ctx->cov_cap = n;
        }
        ds = ctx->d_cov_sum; dm = ctx->d_cov_min; dx = ctx->d_cov_miss;
    }
    // 256 k-mers per wavefront, 4 wavefronts per block
    uint64_t g_range[2];
    PF_HIP(hipMemcpyAsync(&g_range[0], ctx->d_kpre + u0, 8, hipMemcpyDeviceToHost, ctx->stream));
    PF_HIP(hipMemcpyAsync(&g_range[1], ctx->d_kpre + u1, 8, hipMemcpyDeviceToHost, ctx->stream));
    PF_HIP(hipStreamSynchronize(ctx->stream));
    const uint64_t w_begin = g_range[0] / KCOV_WIN, w_end = (g_range[1] + KCOV_WIN - 1) / KCOV_WIN;
    static const bool env_probe = [] { const char *e = getenv("PF_KCOV_SCAN"); return e && !strcmp(e, "probe"); }();  // measurements: the probing form in whole runs
    probe = probe || env_probe;
    { const int rc = join_finish(ctx); if (rc) return rc; }   // look-ups of pf_join_counts_begin still on their way
    if (!exact && !probe && !ctx->gcov_valid) {  // graph and table were uploaded before the join existed for them
        const int rc = join_graph_counts(ctx);
        if (rc) return rc;
    }
    ctx_begin(ctx, PF_K_COV);
    k_cov_init<<<ctx_grid(ctx, n, 256, 8), 256, 0, ctx->stream>>>(n, ds, dm, dx);
    if (!exact && !probe && ctx->gcov_valid) {
        // streaming form (pf_cov_stream.hpp): four k-mers per lane, windows of KC4_SR super-rows of 256 k-mers.  Earlier forms of
        // this round -- one k-mer per lane with the scan on DPP (0.111 ms) or ds_bpermute (0.123 ms) -- are in profiles/history/r01j_kcov_stream.txt
        Kc4Args a{ctx->d_gcov, 0, nullptr, ctx->d_khead, ctx->d_krow, u0, n, g_range[0], g_range[1], g_range[0] / 256, (g_range[1] + 255) / 256,
                  ds, dm, nullptr, dx};
        const bool wide = ctx->tab_max_count >= (1ull << 20);  // a window's carry sums up to KC4_SR * 256 counts in the narrow type
        const int rc = launch_cov_stream(ctx, a, 1, wide, false);
        if (rc) return rc;
    } else {
        k_cov<<<ctx_grid(ctx, (w_end - w_begin) * 64, 256, 16), 256, 0, ctx->stream>>>(ctx->d_tab, ctx->tab_cap - 1, ctx->k, ctx->d_seq, ctx->d_off, ctx->d_kpre,
                                                                                     ctx->d_kwin, ctx->N, ctx->tab_one_strand, exact, u0, g_range[0],
                                                                                     g_range[1], w_begin, w_end - w_begin, ds, dm, dx, nullptr);
    }
    ctx_end(ctx);
    if (!dev_out) {
        PF_HIP(hipMemcpyAsync(sum, ds, (size_t)n * 8, hipMemcpyDeviceToHost, ctx->stream));
        PF_HIP(hipMemcpyAsync(mn, dm, (size_t)n * 4, hipMemcpyDeviceToHost, ctx->stream));
        PF_HIP(hipMemcpyAsync(miss, dx, (size_t)n, hipMemcpyDeviceToHost, ctx->stream));
        PF_HIP(hipStreamSynchronize(ctx->stream));
        for (uint32_t i = 0; i < n; ++i)
            if (miss[i]) { pf::CtxErr{ctx} = "a k-mer of the graph is missing from the count table"; return PF_ERR_MISSING_KMER; }
    }
    return PF_OK;
}

int pf_unitig_cov(pf_ctx *ctx, uint32_t u0, uint32_t u1, uint64_t *sum, uint32_t *mn, uint8_t *miss) {
    if (ctx && ctx->tab_exact) {
        pf::CtxErr{ctx} = "the count database was built without canonical counting: pf_unitig_cov_exact gives its per-orientation coverage";
        return PF_ERR_ARG;
    }
    return unitig_cov_impl(ctx, u0, u1, 0, sum, mn, miss);
}

int pf_unitig_cov_probe(pf_ctx *ctx, uint32_t u0, uint32_t u1, uint64_t *sum, uint32_t *mn, uint8_t *miss) {
    if (ctx && ctx->tab_exact) {
        pf::CtxErr{ctx} = "the count database was built without canonical counting: pf_unitig_cov_exact gives its per-orientation coverage";
        return PF_ERR_ARG;
    }
    return unitig_cov_impl(ctx, u0, u1, 0, sum, mn, miss, true);
}

int pf_unitig_cov_exact(pf_ctx *ctx, uint32_t u0, uint32_t u1, int reverse, uint64_t *sum, uint32_t *mn, uint8_t *miss) {
    return unitig_cov_impl(ctx, u0, u1, reverse ? 2 : 1, sum, mn, miss);
}

int pf_string_cov(pf_ctx *ctx, const char *text, const uint64_t *str_off, uint32_t n_str, uint32_t low, uint32_t up,
                  uint64_t *sum, uint8_t *ok, uint8_t *miss) {
    if (!ctx || !ctx->d_tab || (n_str && (!text || !str_off || !sum || !ok || !miss))) return PF_ERR_ARG;
    if (n_str == 0) return PF_OK;
    PF_HIP(hipSetDevice(ctx->device));
    if (ctx->tab_exact) {
        // readCov(string) looks nothing up in a database without canonical counting and returns (0, true) (src/CDBG.cpp:34, 59)
        hipPointerAttribute_t at;
        const bool dev_out = hipPointerGetAttributes(&at, sum) == hipSuccess && at.type == hipMemoryTypeDevice;
        (void)hipGetLastError();
        if (dev_out) {
            PF_HIP(hipMemsetAsync(sum, 0, (size_t)n_str * 8, ctx->stream));
            PF_HIP(hipMemsetAsync(ok, 1, n_str, ctx->stream));
            PF_HIP(hipMemsetAsync(miss, 0, n_str, ctx->stream));
            PF_HIP(hipStreamSynchronize(ctx->stream));
        } else {
            memset(sum, 0, (size_t)n_str * 8);
            memset(ok, 1, n_str);
            memset(miss, 0, n_str);
        }
        return PF_OK;
    }
    uint64_t total = 0;
    PF_HIP(hipMemcpy(&total, str_off + n_str, 8, hipMemcpyDefault));
    char *dt = (char *)ctx_ws(ctx, WS_STR_TEXT, (size_t)total + 1);
    uint64_t *doff = (uint64_t *)ctx_ws(ctx, WS_STR_OFF, ((size_t)n_str + 1) * 8);
    uint64_t *ds = (uint64_t *)ctx_ws(ctx, WS_STR_SUM, (size_t)n_str * 8);
    uint8_t *dk = (uint8_t *)ctx_ws(ctx, WS_STR_OK, n_str);
    uint8_t *dm = (uint8_t *)ctx_ws(ctx, WS_STR_MISS, n_str);
    if (!dt || !doff || !ds || !dk || !dm) return PF_ERR_HIP;
    PF_HIP(hipMemcpyAsync(dt, text, (size_t)total, hipMemcpyDefault, ctx->stream));
    PF_HIP(hipMemcpyAsync(doff, str_off, ((size_t)n_str + 1) * 8, hipMemcpyDefault, ctx->stream));
    ctx_begin(ctx, PF_K_STRCOV);
    k_strcov<<<ctx_grid(ctx, n_str, 256, 8), 256, 0, ctx->stream>>>(ctx->d_tab, ctx->tab_cap - 1, ctx->tab_k, ctx->tab_one_strand, dt, doff, n_str, low, up, ds, dk, dm);
    ctx_end(ctx);
    PF_HIP(hipMemcpyAsync(sum, ds, (size_t)n_str * 8, hipMemcpyDefault, ctx->stream));
    PF_HIP(hipMemcpyAsync(ok, dk, n_str, hipMemcpyDefault, ctx->stream));
    PF_HIP(hipMemcpyAsync(miss, dm, n_str, hipMemcpyDefault, ctx->stream));
    PF_HIP(hipStreamSynchronize(ctx->stream));
    return PF_OK;
}

static void cand_range(const pf_ctx *ctx, uint32_t u0, uint32_t u1, uint64_t *c0, uint64_t *c1) {
    const auto &v = ctx->h_cand;
    *c0 = std::lower_bound(v.begin(), v.end(), u0 * 2) - v.begin();
    *c1 = std::lower_bound(v.begin(), v.end(), u1 * 2) - v.begin();
}

int pf_count_candidates(pf_ctx *ctx, uint32_t u0, uint32_t u1, uint64_t *n) {
    if (!ctx || !ctx->has_adj || u0 > u1 || u1 > ctx->N || !n) return PF_ERR_ARG;
    uint64_t c0, c1;
    cand_range(ctx, u0, u1, &c0, &c1);
    *n = c1 - c0;
    return PF_OK;
}

// deferred != nullptr: the third tier is left to the caller (pf_bfs_candidates_split)
// async_copy (pf_bfs_candidates_begin): the records and the pool are sent to the host on the copy stream and the call returns;
// pf_bfs_candidates_end waits for them and marks the deferred records.
static int bfs_candidates_impl(pf_ctx *ctx, uint32_t u0, uint32_t u1, pf_bfs_record *records, uint64_t rec_cap, uint32_t *pool,
                               uint64_t pool_cap, uint64_t *n_records, uint64_t *pool_used, uint32_t *deferred, uint64_t deferred_cap,
                               uint64_t *n_deferred, bool async_copy = false, uint32_t *deferred_entrance = nullptr) {
    if (!ctx || !ctx->has_adj || u0 > u1 || u1 > ctx->N || !records || !pool || !n_records || !pool_used) return PF_ERR_ARG;
    if (ctx->bfs_pending.active) { pf::CtxErr{ctx} = "pf_bfs_candidates_end first"; return PF_ERR_ARG; }
    if (n_deferred) *n_deferred = 0;
    PF_HIP(hipSetDevice(ctx->device));
    uint64_t c0, c1;
    cand_range(ctx, u0, u1, &c0, &c1);
    const uint64_t n = c1 - c0;
    *n_records = n;
    *pool_used = 0;
    if (n > rec_cap) { pf::CtxErr{ctx} = "record buffer too small"; return PF_ERR_OVERFLOW; }
    if (n == 0) return PF_OK;
    hipPointerAttribute_t at;
    const bool dev_out = hipPointerGetAttributes(&at, records) == hipSuccess && at.type == hipMemoryTypeDevice;
    (void)hipGetLastError();
    pf_bfs_record *d_rec = records;
    uint32_t *d_pool = pool;
    if (!dev_out) {
        d_rec = (pf_bfs_record *)ctx_ws(ctx, WS_BFS_REC, n * sizeof(pf_bfs_record));
        d_pool = (uint32_t *)ctx_ws(ctx, WS_BFS_POOL, (pool_cap ? pool_cap : 1) * 4);
        if (!d_rec || !d_pool) return PF_ERR_HIP;
    }
    uint8_t *small = (uint8_t *)ctx_ws(ctx, WS_BFS_SMALL, 64);
    uint32_t *d_def = (uint32_t *)ctx_ws(ctx, WS_BFS_DEF, (n * 2 + 8) * 4);
    if (!small || !d_def) return PF_ERR_HIP;
    unsigned long long *d_head = reinterpret_cast<unsigned long long *>(small);
    unsigned int *d_ndef = reinterpret_cast<unsigned int *>(small + 16);
    unsigned int *d_ndef2 = reinterpret_cast<unsigned int *>(small + 32);
    PF_HIP(hipMemsetAsync(small, 0, 64, ctx->stream));
    unsigned int *d_nlive = reinterpret_cast<unsigned int *>(small + 56);
    BfsOut o{d_rec, d_pool, pool_cap, d_head, d_def, d_ndef, d_def + n + 4, d_ndef2, nullptr, 0, d_nlive, 0};
    if (ctx->h_live && deferred) {   // (armed by pf_bfs_live_deferred: the caller polls the list while the kernels below run)
        // The list is zeroed by pf_bfs_live_deferred -- once per arming, BEFORE the caller starts the threads that poll it.  Zeroing it
        // here raced with them (they saw the pass before's entries first; advisor, round 3) and a retry after a pool overflow wiped
        // what they were reading.  A retry writes the list again from slot 0: an entry a poller took before and the one that replaces
        // it are both real candidates of this graph, walks are keyed by (candidate, entrance), what no poller saw is walked afterwards.
        o.live = reinterpret_cast<unsigned long long *>(ctx->h_live);
        o.live_cap = (uint32_t)ctx->live_cap;
        // (PF_BFS_HINT_AT, read per call: measurements; beyond the tier's tables = notice only when it gives up)
        o.hint_at = [] { const char *e = getenv("PF_BFS_HINT_AT"); return e ? (uint32_t)std::max(9, atoi(e)) : 48u; }();
    }
    ctx->bfs_live_n = 0;
    // thread tier first (one thread per candidate, 8-entry tables), then the wavefront tier for what outgrew it
    uint32_t *d_wlist = (uint32_t *)ctx_ws(ctx, WS_BFS_WLIST, (n + 8) * 4);
    if (!d_wlist) return PF_ERR_HIP;
    unsigned int *d_nwlist = reinterpret_cast<unsigned int *>(small + 48);
    constexpr bool thread_tier = true;
    // (PF_BFS_WAVE_CAP, read per call: measurements of where the wavefront tier should give up)
    const uint32_t wave_cap = [] { const char *e = getenv("PF_BFS_WAVE_CAP"); return e ? (uint32_t)std::max(16, std::min((int)BFS_LDS_CAP, atoi(e))) : BFS_LDS_CAP; }();
    if (thread_tier) {
        ctx_begin(ctx, PF_K_BFS_THREAD);
        k_bfs_thread<<<(unsigned)((n + 255) / 256), 256, 0, ctx->stream>>>(ctx->d_succ, ctx->d_pred, ctx->d_cand, c0, c1, o, d_wlist, d_nwlist);
        ctx_end(ctx);
        const int grid = ctx_grid(ctx, (n / 4 + 64) * 64, 256, 8);
        ctx_begin(ctx, PF_K_BFS);
        k_bfs<<<grid, 256, 0, ctx->stream>>>(ctx->d_succ, ctx->d_pred, ctx->d_cand, c0, c1, o, d_wlist, d_nwlist, wave_cap);
        ctx_end(ctx);
    } else {
        const int grid = ctx_grid(ctx, n * 64, 256, 8);
        ctx_begin(ctx, PF_K_BFS);
        k_bfs<<<grid, 256, 0, ctx->stream>>>(ctx->d_succ, ctx->d_pred, ctx->d_cand, c0, c1, o, nullptr, nullptr, wave_cap);
        ctx_end(ctx);
    }
    unsigned int n_def = 0, n_wl = 0, n_live = 0;
    PF_HIP(hipMemcpyAsync(&n_def, d_ndef, 4, hipMemcpyDeviceToHost, ctx->stream));
    if (o.live) PF_HIP(hipMemcpyAsync(&n_live, d_nlive, 4, hipMemcpyDeviceToHost, ctx->stream));
    if (ctx->timing && thread_tier) PF_HIP(hipMemcpyAsync(&n_wl, d_nwlist, 4, hipMemcpyDeviceToHost, ctx->stream));
    PF_HIP(hipStreamSynchronize(ctx->stream));
    ctx->bfs_live_n = n_live;
    ctx_units(ctx, PF_K_BFS, thread_tier ? n_wl : n);
    if (thread_tier) ctx_units(ctx, PF_K_BFS_THREAD, n);
    int status = PF_OK;
    if (n_def && deferred) {
        // the caller walks everything that outgrew the LDS tier itself (a host core needs ~20 ns per vertex; the 4096-entry
        // tier below searches its tables linearly and is quadratic in the traversal's size)
        if (n_deferred) *n_deferred = n_def;
        if (n_def > deferred_cap) { pf::CtxErr{ctx} = "deferred-candidate buffer too small"; status = PF_ERR_OVERFLOW; }
        else PF_HIP(hipMemcpy(deferred, d_def, (size_t)n_def * 4, hipMemcpyDeviceToHost));
    } else if (n_def) {
        const unsigned int waves = std::min<unsigned int>(n_def, 256);
        DevTmp<uint32_t> sc32_;
        DevTmp<uint8_t> sc8_;
        PF_HIP(sc32_.alloc((size_t)waves * 3 * BFS_BIG_CAP * 4));
        PF_HIP(sc8_.alloc((size_t)waves * BFS_BIG_CAP));
        uint32_t *sc32 = sc32_.p;
        uint8_t *sc8 = sc8_.p;
        ctx_begin(ctx, PF_K_BFS_BIG);
        k_bfs_big<<<waves, 64, 0, ctx->stream>>>(ctx->d_succ, ctx->d_pred, ctx->d_cand, c0, n_def, sc32, sc8, o);
        ctx_end(ctx);
        unsigned int n_def2 = 0;
        PF_HIP(hipMemcpyAsync(&n_def2, d_ndef2, 4, hipMemcpyDeviceToHost, ctx->stream));
        PF_HIP(hipStreamSynchronize(ctx->stream));
        if (n_def2) {
            // traversals beyond the linear tables: direct-indexed state sized by the graph, a few waves
            // one wave per traversal, as many side by side as ~16 GiB of state allow
            const unsigned int hw = (unsigned int)std::max<size_t>(1, std::min<size_t>(std::min<size_t>(n_def2, 64), (16ull << 30) / (36 * (size_t)ctx->N + 64)));
            const size_t per = 9 * (size_t)ctx->N + 16;
            DevTmp<uint32_t> hs_;
            PF_HIP(hs_.alloc(per * hw * 4));
            uint32_t *hs = hs_.p;
            PF_HIP(hipMemsetAsync(hs, 0, per * hw * 4, ctx->stream));
            ctx_begin(ctx, PF_K_BFS_BIG);
            if (!ctx->d_pred16) {
                PF_HIP(hipMalloc(&ctx->d_pred16, (size_t)ctx->N * 2 * 16 * 4));
                k_pred16<<<ctx_grid(ctx, (uint64_t)ctx->N * 8, 256, 8), 256, 0, ctx->stream>>>(ctx->d_succ, ctx->d_pred, ctx->N * 2, ctx->d_pred16);
            }
            k_bfs_huge<<<hw, 64, 0, ctx->stream>>>(ctx->d_succ, ctx->d_pred16, ctx->d_cand, c0, n_def2, hs, ctx->N, o);
            ctx_end(ctx);
            PF_HIP(hipStreamSynchronize(ctx->stream));
        }
    }
    ctx->bfs_deferred = n_def;
    ctx->bfs_last_rec = d_rec; ctx->bfs_last_pool = d_pool; ctx->bfs_last_n = n; ctx->bfs_last_pool_len = pool_cap;
    ctx->bfs_call_id++;
    unsigned long long head = 0;
    PF_HIP(hipMemcpy(&head, d_head, 8, hipMemcpyDeviceToHost));
    *pool_used = head;
    if (head > pool_cap) {
        pf::CtxErr{ctx} = "vertex pool too small";
        status = PF_ERR_OVERFLOW;
    }
    if (!dev_out && status == PF_OK && async_copy) {
        if (!ctx->copy_stream) PF_HIP(hipStreamCreateWithFlags(&ctx->copy_stream, hipStreamNonBlocking));
        PF_HIP(hipMemcpyAsync(records, d_rec, n * sizeof(pf_bfs_record), hipMemcpyDeviceToHost, ctx->copy_stream));
        PF_HIP(hipMemcpyAsync(pool, d_pool, (size_t)head * 4, hipMemcpyDeviceToHost, ctx->copy_stream));
        ctx->bfs_pending.active = true;
        ctx->bfs_pending.records = records;
        ctx->bfs_pending.c0 = c0;
        ctx->bfs_pending.deferred.assign(deferred, deferred + (n_deferred ? *n_deferred : 0));
        if (deferred_entrance && n_deferred)
            for (uint64_t d = 0; d < *n_deferred; ++d) deferred_entrance[d] = ctx->h_cand[c0 + deferred[d]];
        return PF_OK;
    }
    if (!dev_out) {
        if (status == PF_OK) {
            PF_HIP(hipMemcpy(records, d_rec, n * sizeof(pf_bfs_record), hipMemcpyDeviceToHost));
            PF_HIP(hipMemcpy(pool, d_pool, (size_t)head * 4, hipMemcpyDeviceToHost));
            if (deferred && n_deferred)
                for (uint64_t d = 0; d < *n_deferred; ++d) {  // records the caller fills: entrance set, everything else empty
                    pf_bfs_record &r = records[deferred[d]];
                    memset(&r, 0, sizeof r);
                    r.entrance = ctx->h_cand[c0 + deferred[d]];
                    r.exit = NONE;
                }
            for (uint64_t i = 0; i < n; ++i)
                if (records[i].outcome == BFS_TOO_LARGE) {
                    pf::CtxErr{ctx} = "a traversal exceeded the direct-indexed tier (internal limit)";
                    status = PF_ERR_OVERFLOW;
                    break;
                }
        }
    }
    return status;
}

int pf_bfs_candidates(pf_ctx *ctx, uint32_t u0, uint32_t u1, pf_bfs_record *records, uint64_t rec_cap, uint32_t *pool,
                      uint64_t pool_cap, uint64_t *n_records, uint64_t *pool_used) {
    return bfs_candidates_impl(ctx, u0, u1, records, rec_cap, pool, pool_cap, n_records, pool_used, nullptr, 0, nullptr);
}

int pf_bfs_candidates_split(pf_ctx *ctx, uint32_t u0, uint32_t u1, pf_bfs_record *records, uint64_t rec_cap, uint32_t *pool,
                            uint64_t pool_cap, uint64_t *n_records, uint64_t *pool_used, uint32_t *deferred, uint64_t deferred_cap,
                            uint64_t *n_deferred) {
    if (!deferred || !n_deferred) return PF_ERR_ARG;
    if (ctx) {
        hipPointerAttribute_t at;
        if (records && hipPointerGetAttributes(&at, records) == hipSuccess && at.type == hipMemoryTypeDevice) {
            pf::CtxErr{ctx} = "pf_bfs_candidates_split fills host records";
            return PF_ERR_ARG;
        }
        (void)hipGetLastError();
    }
    return bfs_candidates_impl(ctx, u0, u1, records, rec_cap, pool, pool_cap, n_records, pool_used, deferred, deferred_cap, n_deferred);
}

int pf_bfs_candidates_begin(pf_ctx *ctx, uint32_t u0, uint32_t u1, pf_bfs_record *records, uint64_t rec_cap, uint32_t *pool,
                            uint64_t pool_cap, uint64_t *n_records, uint64_t *pool_used, uint32_t *deferred, uint32_t *deferred_entrance,
                            uint64_t deferred_cap, uint64_t *n_deferred) {
    if (!deferred || !deferred_entrance || !n_deferred) return PF_ERR_ARG;
    if (ctx) {
        hipPointerAttribute_t at;
        if (records && hipPointerGetAttributes(&at, records) == hipSuccess && at.type == hipMemoryTypeDevice) {
            pf::CtxErr{ctx} = "pf_bfs_candidates_begin fills host records";
            return PF_ERR_ARG;
        }
        (void)hipGetLastError();
    }
    return bfs_candidates_impl(ctx, u0, u1, records, rec_cap, pool, pool_cap, n_records, pool_used, deferred, deferred_cap, n_deferred, true,
                               deferred_entrance);
}

// K-BFS with records and vertex pool left in the context's own device buffers (for K-CC and the device-side commits): nothing
// travels to the host but the deferred candidates' indices and entrances.
int pf_bfs_candidates_resident(pf_ctx *ctx, uint32_t u0, uint32_t u1, uint64_t *n_records, uint64_t *pool_used, uint32_t *deferred,
                               uint32_t *deferred_entrance, uint64_t deferred_cap, uint64_t *n_deferred) {
    if (!ctx || !n_records || !pool_used || !deferred || !deferred_entrance || !n_deferred) return PF_ERR_ARG;
    if (!ctx->has_adj || u0 > u1 || u1 > ctx->N) return PF_ERR_ARG;
    uint64_t c0, c1;
    cand_range(ctx, u0, u1, &c0, &c1);
    const uint64_t n = c1 - c0;
    uint64_t cap = std::max<uint64_t>(ctx->bfs_res_pool_cap, n * 6 + (1u << 20));
    for (int attempt = 0; attempt < 4; ++attempt) {
        pf_bfs_record *d_rec = (pf_bfs_record *)ctx_ws(ctx, WS_BFS_RES_REC, (n + 1) * sizeof(pf_bfs_record));
        uint32_t *d_pool = (uint32_t *)ctx_ws(ctx, WS_BFS_RES_POOL, (cap + 1) * 4);
        if (!d_rec || !d_pool) return PF_ERR_HIP;
        const int st = bfs_candidates_impl(ctx, u0, u1, d_rec, n + 1, d_pool, cap, n_records, pool_used, deferred, deferred_cap, n_deferred);
        if (st == PF_ERR_OVERFLOW && *pool_used > cap) { cap = *pool_used + *pool_used / 8 + 1024; continue; }
        if (st == PF_OK) {
            ctx->bfs_res_pool_cap = cap;
            for (uint64_t d = 0; d < *n_deferred; ++d) deferred_entrance[d] = ctx->h_cand[c0 + deferred[d]];
        }
        return st;
    }
    pf::CtxErr{ctx} = "pf_bfs_candidates_resident: the vertex pool does not converge";
    return PF_ERR_OVERFLOW;
}

int pf_bfs_live_count(pf_ctx *ctx, uint64_t *n) {
    if (!ctx || !n) return PF_ERR_ARG;
    *n = ctx->bfs_live_n;
    return PF_OK;
}

int pf_bfs_live_deferred(pf_ctx *ctx, uint64_t cap, volatile uint64_t **list) {
    if (!ctx || !list) return PF_ERR_ARG;
    *list = nullptr;
    PF_HIP(hipSetDevice(ctx->device));
    if (cap == 0) {   // off
        if (ctx->h_live) (void)hipHostFree(ctx->h_live);
        ctx->h_live = nullptr;
        ctx->live_cap = 0;
        return PF_OK;
    }
    if (ctx->h_live && ctx->live_cap != cap) { (void)hipHostFree(ctx->h_live); ctx->h_live = nullptr; ctx->live_cap = 0; }
    if (!ctx->h_live) {
        void *p = nullptr;
        if (hipHostMalloc(&p, cap * 8, hipHostMallocCoherent | hipHostMallocMapped) != hipSuccess) { (void)hipGetLastError(); pf::CtxErr{ctx} = "pf_bfs_live_deferred: no pinned host memory"; return PF_ERR_HIP; }
        ctx->h_live = static_cast<unsigned long long *>(p);
        ctx->live_cap = cap;
    }
    for (uint64_t x = 0; x < cap; ++x) __atomic_store_n(&ctx->h_live[x], 0ull, __ATOMIC_RELAXED);   // (per arming: see bfs_candidates_impl)
    __atomic_thread_fence(__ATOMIC_SEQ_CST);
    *list = reinterpret_cast<volatile uint64_t *>(ctx->h_live);
    return PF_OK;
}

int pf_bfs_candidates_end(pf_ctx *ctx) {
    if (!ctx) return PF_ERR_ARG;
    if (!ctx->bfs_pending.active) return PF_OK;
    ctx->bfs_pending.active = false;
    PF_HIP(hipSetDevice(ctx->device));
    PF_HIP(hipStreamSynchronize(ctx->copy_stream));
    for (uint32_t idx : ctx->bfs_pending.deferred) {  // records the caller fills: entrance set, everything else empty
        pf_bfs_record &r = ctx->bfs_pending.records[idx];
        memset(&r, 0, sizeof r);
        r.entrance = ctx->h_cand[ctx->bfs_pending.c0 + idx];
        r.exit = NONE;
    }
    return PF_OK;
}

}  // extern "C"
