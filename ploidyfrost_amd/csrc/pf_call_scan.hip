// The resident calling pipeline (pf_call.hip has the overview), device side: K-SCAN and part B of the driver loop: k_call_count_sides, k_call_sides, k_call_pending, k_call_resolve.
#include "pf_call_kernels.hpp"

namespace pf_call {

// ---------------------------------------------------------------------------------------------------------------------
// K-SCAN

// readCovUni(u, low, up, c) of src/CCDBG.cpp:123-156 from K-COV-C's resident results: (sum / len, true) iff every k-mer is in colour
// c's database with low < count < up, else (0, false)
struct ColourCov {
    const uint64_t *sum;
    const uint32_t *mn, *mx;
    const uint8_t *miss;
    const uint32_t *low, *up;
    uint32_t N;
    __device__ inline bool ok(uint32_t c, uint32_t u) const {
        const size_t o = (size_t)c * N + u;
        return !miss[o] && mn[o] > low[c] && mx[o] < up[c];
    }
    __device__ inline double mean(uint32_t c, uint32_t u, uint32_t len_km) const { return (double)sum[(size_t)c * N + u] / (double)len_km; }
};

// colored sortSeq_simple (src/CCDBG.cpp:368-480) with its exact partition scheme: descending number of colours, then descending
// length, then descending reference string.  n <= 4.
__device__ inline void sort_inner_colored_dev(const uint64_t *__restrict__ seq, const uint64_t *__restrict__ off, const uint32_t *__restrict__ len,
                                              uint32_t *pc, uint32_t *ov, int n) {
    int stack_lo[8], stack_hi[8];
    int sp = 1;
    stack_lo[0] = 0;
    stack_hi[0] = n - 1;
    auto ref_cmp = [&](int x, int y) -> int {   // length first, then the strings (equal lengths: strcmp)
        const uint32_t lx = len[ov[x] >> 1], ly = len[ov[y] >> 1];
        if (lx != ly) return lx > ly ? 1 : -1;
        return unitig_cmp(seq, off, len, ov[x] >> 1, ov[y] >> 1);
    };
    while (sp > 0) {
        --sp;
        const int low = stack_lo[sp], high = stack_hi[sp];
        if (high <= low) continue;
        int i = low, j = high;
        for (;;) {
            while (pc[i] >= pc[low]) {
                if (pc[i] > pc[low] || ref_cmp(i, low) > 0) i++;
                else break;
                if (i == high) break;
            }
            while (pc[j] <= pc[low]) {
                if (pc[j] < pc[low] || ref_cmp(j, low) < 0) j--;
                else break;
                if (j == low) break;
            }
            if (i >= j) break;
            const uint32_t tp = pc[i]; pc[i] = pc[j]; pc[j] = tp;
            const uint32_t to = ov[i]; ov[i] = ov[j]; ov[j] = to;
        }
        {
            const uint32_t tp = pc[low]; pc[low] = pc[j]; pc[j] = tp;
            const uint32_t to = ov[low]; ov[low] = ov[j]; ov[j] = to;
        }
        stack_lo[sp] = low; stack_hi[sp] = j - 1; ++sp;
        stack_lo[sp] = j + 1; stack_hi[sp] = high; ++sp;
    }
}

__global__ void k_call_count_sides(const uint8_t *__restrict__ flags, uint32_t N, uint32_t *__restrict__ cnt) {
    const uint32_t u = blockIdx.x * blockDim.x + threadIdx.x;
    if (u <= N) cnt[u] = u < N ? (uint32_t)__popc(flags[u] & 3u) : 0u;
}

__device__ inline uint32_t first_succ(const uint32_t *__restrict__ succ, uint32_t ov) {
    const uint4 r = *reinterpret_cast<const uint4 *>(succ + (size_t)ov * 4);
    if (r.x != NONE) return r.x;
    if (r.y != NONE) return r.y;
    if (r.z != NONE) return r.z;
    return r.w;
}

template <bool COLORED>
__global__ __launch_bounds__(256) void k_call_sides(ScanArgs a) {
    const uint32_t u = blockIdx.x * blockDim.x + threadIdx.x;
    if (u >= a.N) return;
    const uint8_t f = a.flags[u];
    if ((f & 3) == 0) return;
    uint32_t slot = a.side_base[u];
    const uint32_t N = a.N;
    auto cslot = [&](uint32_t ov) -> size_t { return (a.per_strand && (ov & 1)) ? (size_t)N + (ov >> 1) : (size_t)(ov >> 1); };
    auto len_km = [&](uint32_t x) { return a.len[x] - (uint32_t)a.k + 1; };
    auto mean_ov = [&](uint32_t ov) { return (double)a.cov_sum[cslot(ov)] / (double)len_km(ov >> 1); };
    for (int side = 0; side < 2; ++side) {
        const bool ps = side == 0;
        if (!(f & (ps ? B_PLUS : B_MINUS))) continue;
        pf_call_side r;
        r.u = u;
        r.exit_ov = NONE;
        r.err_unitig = 0;
        r.plus_side = ps;
        r.kind = 0;
        r.aligned = 0;
        r.err = 0;
        CallTask t;
        t.u = u;
        t.entrance_ov = t.exit_ov = 0;
        t.strict = t.n_inner = t.n_cov = t.pad_ = 0;
        for (int q = 0; q < 4; ++q) { t.inner[q] = 0; t.cov[q] = 0; }
        t.core_mean = t.cov_sum = 0;
        const uint32_t my = slot++;
        do {
            if (f & (ps ? B_COMPLEX_P : B_COMPLEX_M)) { r.kind = 1; break; }
            const uint32_t uo = 2 * u + (ps ? 0 : 1);
            const bool strict = (f & (ps ? B_STRICT_P : B_STRICT_M)) != 0;
            if (!COLORED && a.cov_miss[cslot(uo)]) { r.err = 1; r.err_unitig = u; break; }  // core = readCov(u), u oriented
            uint32_t exit_ov;
            if (strict) {
                exit_ov = first_succ(a.succ, uo);
                if (exit_ov != NONE) exit_ov = first_succ(a.succ, exit_ov);
            } else {
                const uint32_t want = ps ? a.plus[u] : a.minus[u];
                exit_ov = first_succ(a.succ, uo);
                // (bounded: a walk longer than the graph means the partner is not on the first-successor chain)
                for (uint32_t steps = 0; exit_ov != NONE && (exit_ov >> 1) + 1 != want; ++steps) {
                    if (steps > N) { exit_ov = NONE; break; }
                    exit_ov = first_succ(a.succ, exit_ov);
                }
            }
            if (exit_ov == NONE) { r.err = 2; break; }
            const uint32_t eu = exit_ov >> 1;
            r.exit_ov = exit_ov;
            t.entrance_ov = uo;
            t.exit_ov = exit_ov;
            t.strict = strict;
            if (unitig_cmp(a.seq, a.off, a.len, u, eu) < 0) { r.kind = 2; break; }  // the other endpoint owns this bubble
            r.kind = 3;
            if (COLORED) {
                // src/CCDBG.cpp:2838-2853: the per-colour means are summed until a colour fails its range test (the `flag == false;`
                // there is a no-op, so the bubble is processed regardless)
                const ColourCov cc{a.ccov_sum, a.ccov_min, a.ccov_max, a.ccov_miss, a.clow, a.cup, N};
                const uint32_t C = a.n_colors;
                double core = 0;
                for (uint32_t c = 0; c < C; ++c) {
                    if (!cc.ok(c, u)) break;
                    core += cc.mean(c, u, len_km(u));
                }
                t.core_mean = core;
                bool flag = true;
                if (strict) {   // :2867-2931: the [colour][path] matrix of mean coverages, its gates, the colored sortSeq_simple
                    uint32_t pc[4] = {0, 0, 0, 0};
                    uint32_t path = 0;
                    const uint32_t *row = a.succ + (size_t)uo * 4;
                    for (int b = 0; b < 4 && flag; ++b) {
                        const uint32_t w = row[b];
                        if (w == NONE) continue;
                        const uint32_t wu = w >> 1;
                        t.inner[t.n_inner++] = w;
                        uint32_t jn = 0;
                        for (uint32_t c = 0; c < C; ++c) {
                            if (!colour_in(a.full, a.cwords, wu, c)) continue;
                            ++jn;
                            if (!cc.ok(c, wu)) { flag = false; break; }
                        }
                        if (!flag) break;
                        if (a.size_total[wu] != (uint64_t)jn * len_km(wu)) { flag = false; break; }  // a colour on part of it
                        pc[path++] = jn;
                    }
                    if (flag) {   // some colour must see more than one of the paths (an entry of the matrix is its mean, 0 if absent)
                        flag = false;
                        for (uint32_t c = 0; c < C && !flag; ++c) {
                            int nz = 0;
                            for (uint32_t q = 0; q < path; ++q) {
                                const uint32_t wu = t.inner[q] >> 1;
                                nz += colour_in(a.full, a.cwords, wu, c) && cc.mean(c, wu, len_km(wu)) != 0.0;
                            }
                            flag = nz > 1;
                        }
                    }
                    if (flag) {
                        sort_inner_colored_dev(a.seq, a.off, a.len, pc, t.inner, (int)path);
                        t.n_cov = (uint8_t)path;
                    }
                }
                r.aligned = flag;
                break;
            }
            t.core_mean = mean_ov(uo);
            bool aligned = true;
            if (strict) {
                const uint32_t *row = a.succ + (size_t)uo * 4;
                for (int b = 0; b < 4 && aligned && !r.err; ++b) {
                    const uint32_t w = row[b];
                    if (w == NONE) continue;
                    t.inner[t.n_inner++] = w;
                    if (a.cov_miss[cslot(w)]) { r.err = 1; r.err_unitig = w >> 1; break; }
                    const uint32_t mn = a.cov_min[cslot(w)];
                    if (mn > a.low && mn < a.up) {
                        const double mcov = mean_ov(w);
                        t.cov[t.n_cov++] = mcov;
                        t.cov_sum += mcov;
                    } else {
                        aligned = false;
                    }
                }
                if (aligned && !r.err) {
                    // the reference also reads the predecessors' coverage and drops it (src/CDBG.cpp:1224-1239)
                    const uint32_t *prow = a.pred + (size_t)uo * 4;
                    for (int b = 0; b < 4; ++b) {
                        const uint32_t w = prow[b];
                        if (w != NONE && a.cov_miss[cslot(w)]) { r.err = 1; r.err_unitig = w >> 1; break; }
                    }
                    if (!r.err) sort_inner_dev(a.seq, a.off, a.len, t.cov, t.inner, (int)t.n_cov);
                }
            }
            r.aligned = aligned;
        } while (false);
        a.sides[my] = r;
        a.tasks[my] = t;
        // handling this side as the owner clears the facing side of the exit (src/CDBG.cpp:1656-1679): which record is that?
        uint32_t tg = NONE;
        if (r.kind == 3) {
            const uint32_t eu = r.exit_ov >> 1;
            const uint8_t ef = a.flags[eu];
            const bool facing_minus = (r.exit_ov & 1) == 0;   // '+' exit: its minus side faces the bubble
            if (ef & (facing_minus ? B_MINUS : B_PLUS)) tg = a.side_base[eu] + ((facing_minus && (ef & B_PLUS)) ? 1u : 0u);
        }
        a.target[my] = tg;
    }
}

// ---- part B of the driver loop, exactly, without walking the sides one after the other ------------------------------------
// Sequentially (src/CDBG.cpp:1146-1186, 1656-1679): a side is handled only if its bit is still set when its unitig comes up, and
// an owner that is handled clears the side its exit faces.  So side j is alive iff no owner i < j with target(i) = j is alive --
// a recursion over strictly smaller indices.  Rounds of a monotone propagation settle it: `pending[j]` counts the potential
// killers of j not yet known to be dead; a side with no pending killer is alive and kills its target, a killed side releases
// its own target.  Symmetric bubbles settle in two rounds; chains through asymmetric state take one round per link.

__global__ void k_call_pending(ResolveArgs a) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= a.n) return;
    const uint32_t t = a.target[i];
    if (a.sides[i].kind == 3 && t != NONE && t > i) atomicAdd(&a.pending[t], 1);
}

__global__ void k_call_resolve(ResolveArgs a) {
    const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= a.n || a.state[j]) return;
    // (plain loads of values other threads update with atomics in this very launch: a stale value only postpones the decision)
    const bool dead = __hip_atomic_load(&a.killed[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0;
    const int pend = __hip_atomic_load(&a.pending[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (!dead && pend > 0) { atomicAdd(a.undecided, 1u); return; }
    const pf_call_side r = a.sides[j];
    const uint32_t t = a.target[j];
    const bool kills = r.kind == 3 && t != NONE && t > j;
    if (dead) {
        a.state[j] = 2;
        a.flag[j] = 0;
        if (kills) atomicSub(&a.pending[t], 1);
    } else {
        a.state[j] = 1;
        if (r.kind != 1 && r.err) atomicMin(a.first_err, j);
        a.flag[j] = (r.kind == 3 && !r.err && r.aligned) ? 1u : 0u;
        if (kills) __hip_atomic_store(&a.killed[t], (uint8_t)1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// the forms pf_call.hip launches
template __global__ void k_call_sides<true>(ScanArgs);
template __global__ void k_call_sides<false>(ScanArgs);

}  // namespace pf_call
