// The resident calling pipeline (pf_call.hip has the overview), device side: K-PREP and the thread-per-bubble alignment tiers: k_call_prep, k_call_snp, k_call_pair (+ k_call_pair2_reroute), k_call_stack.
#include "pf_call_kernels.hpp"

namespace pf_call {

// ---------------------------------------------------------------------------------------------------------------------
// K-PREP

__global__ __launch_bounds__(256) void k_call_prep(PrepArgs a) {
    const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    int key = KEY_NONE;
    unsigned long long need3 = 0, retry = 0;
    if (j < a.nb) {
        const CallTask &t = a.ct[a.kept[a.t0 + j]];
        pf_bubble_result z;
        z.rows_off = z.site_off = z.group_off = z.ilen_off = 0;
        z.n_rows = z.n_cols = z.n_sites = z.n_indel_len = 0;
        a.res[j] = z;
        if (t.strict) {
            uint32_t l0 = 0, lmax = 0, lmin = 0xFFFFFFFFu, sum = 0;
            for (int q = 0; q < t.n_inner; ++q) {
                const uint32_t L = a.len[t.inner[q] >> 1];
                a.bpath[(size_t)4 * j + q] = pf_bubble_path{0, L, t.inner[q]};
                if (q == 0) l0 = L;
                lmax = L > lmax ? L : lmax;
                lmin = L < lmin ? L : lmin;
                sum += L;
            }
            a.btask[j] = pf_bubble_task{(uint64_t)4 * j, t.n_inner, 0};
            if (t.n_inner >= 2) {  // fewer than two paths: the reference indexes str[1] blindly; skipped
                const int c = bubble_class(l0, lmax);
                key = 2 * c + ((t.n_inner > 2 || lmax > 64) ? 0 : 1);
                if (c == kBubLdsClasses) need3 = bubble_need(l0, lmax);
                retry = job_bytes(sum < 60000u ? sum : 60000u, lmax);
                // two paths of one length: K-SNP looks at them first (thread per bubble) and hands on what is not a single SNP;
                // two short paths of any kind: K-PAIR (thread per bubble)
                // K-STACK first (thread per bubble, a certificate instead of the dynamic programming) for whatever it can hold
                if (t.n_inner > 2 && a.stack_ok && lmax <= STACK_MAX && (a.stack_ok >= 2 || sum == t.n_inner * l0)) key = KEY_STACK;
                if (t.n_inner == 2) {
                    const uint32_t l1 = sum - l0;
                    if (a.snp_ok && sum == 2 * l0) key = KEY_SNP;
                    else if (a.stack_ok >= 3 && lmax <= STACK_MAX) key = KEY_STACK;
                    else if (a.pair_ok && pair_fits<PAIR_MAX>(l0, l1)) key = KEY_PAIR;
                    else if (a.pair_ok && pair_fits<PAIR_MAX2>(l0, l1)) key = KEY_PAIR2;
                }
            }
        } else {
            a.btask[j] = pf_bubble_task{0, 0, 0};
            key = KEY_BRANCHING;
        }
    }
    block_append(key, j, a.lists, a.cnt);
    // class 3 / retry sizing: rare, one atomic per wave that has any
    unsigned long long m3 = need3, mr = retry;
    for (int o = 32; o > 0; o >>= 1) {
        const unsigned long long x3 = ((unsigned long long)__shfl_down((uint32_t)(m3 >> 32), o, WAVE) << 32) | __shfl_down((uint32_t)m3, o, WAVE);
        const unsigned long long xr = ((unsigned long long)__shfl_down((uint32_t)(mr >> 32), o, WAVE) << 32) | __shfl_down((uint32_t)mr, o, WAVE);
        m3 = x3 > m3 ? x3 : m3;
        mr = xr > mr ? xr : mr;
    }
    if (lane_id() == 0) {
        if (m3) atomicMax(&a.cnt->max_need, m3);
        if (mr) atomicMax(&a.cnt->retry_need, mr);
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// K-SNP: the bi-allelic SNP bubble -- two equally long inner unitigs that differ in one base, most of all bubbles -- needs no
// dynamic programming (the single-SNP shortcut of K-BUBBLE, proof in pf_bubble.hip) and no wavefront either: its cost is a
// chain of dependent loads (task -> unitig offsets -> 2-bit words), so one THREAD per bubble keeps 64 of them in flight per
// wavefront instead of one.  The two rows, the SNP column and the groups {1, 2} go to the same pools K-BUBBLE publishes to
// (one atomic per wavefront and pool); anything that is not exactly one mismatch goes to K-BUBBLE's queue of its size class.

constexpr uint32_t SNP_STAGE = 8192;   // bytes of LDS per wavefront for its rows (64 bubbles of two 64-base paths)

__global__ __launch_bounds__(256) void k_call_snp(SnpArgs a) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    const int lane = lane_id();
    const uint32_t n = a.cnt->n_snp;
    const bool active = i < n;
    uint32_t j = 0, m = 0, ov0 = 0, ov1 = 0, col = 0, diff = 0;
    const uint64_t *w0 = nullptr, *w1 = nullptr;
    if (active) {
        j = a.slist[i];
        const CallTask &t = a.ct[a.kept[a.t0 + j]];
        ov0 = t.inner[0];
        ov1 = t.inner[1];
        m = a.len[ov0 >> 1];
        w0 = a.seq + a.off[ov0 >> 1];
        w1 = a.seq + a.off[ov1 >> 1];
        // 32 bases per step from two packed words (the paths are equally long: K-PREP's condition for this list)
        for (uint32_t c = 0; 32 * c < m; ++c) {
            const uint64_t x = oriented_chunk(w0, m, (ov0 & 1) != 0, c) ^ oriented_chunk(w1, m, (ov1 & 1) != 0, c);
            const uint64_t d = (x | (x >> 1)) & 0x5555555555555555ull;   // one bit per differing base
            if (d) {
                diff += (uint32_t)__popcll(d);
                col = 32 * c + (uint32_t)(__clzll((long long)d) >> 1);
            }
        }
    }
    const bool take = active && diff == 1;
    // pool space: one set of atomics per BLOCK.  The pool heads share a cache line and every taker of the batch adds to them: atomics on
    // one line queue one behind the other (~11 ns each), and a wavefront's four were what this launch took -- 40 000 of them, 0.45 of its
    // 0.6 ms (SQ: 81 % of the wave-cycles waiting, 1.5 % issuing VALU).  The four wavefronts of a block pool their totals through LDS.
    const unsigned long long tm = __ballot(take);
    uint32_t my_excl = 0, wave_total = 0;
    {
        // exclusive prefix of 2 m over the taking lanes
        uint32_t mine = take ? 2 * m : 0, incl = mine;
        for (int o = 1; o < WAVE; o <<= 1) {
            const uint32_t x = __shfl_up(incl, o, WAVE);
            if (lane >= o) incl += x;
        }
        my_excl = incl - mine;
        wave_total = __shfl(incl, WAVE - 1, WAVE);
    }
    __shared__ uint32_t s_wtot[4], s_wcnt[4];
    __shared__ unsigned long long s_tb, s_sb;
    const int wv = (int)(threadIdx.x >> 6);
    if (lane == 0) {
        s_wtot[wv] = wave_total;
        s_wcnt[wv] = (uint32_t)__popcll(tm);
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        const uint32_t total = s_wtot[0] + s_wtot[1] + s_wtot[2] + s_wtot[3], cnt = s_wcnt[0] + s_wcnt[1] + s_wcnt[2] + s_wcnt[3];
        unsigned long long tb = 0, sb = 0;
        if (cnt) {
            tb = atomicAdd(&a.heads[0], (unsigned long long)total);
            sb = atomicAdd(&a.heads[1], (unsigned long long)cnt);
            atomicAdd(&a.heads[2], 2ull * cnt);   // groups: two bytes per site, at 2 * (site index)
            atomicAdd(&a.cnt->n_snp_done, cnt);
        }
        s_tb = tb;
        s_sb = sb;
    }
    __syncthreads();
    uint32_t before_t = 0, before_c = 0;
    for (int w = 0; w < wv; ++w) { before_t += s_wtot[w]; before_c += s_wcnt[w]; }
    const unsigned long long wb = s_tb + before_t;   // where this wavefront's rows start
    const unsigned long long t_off = wb + my_excl;
    const unsigned long long s_off = s_sb + before_c + (unsigned long long)__popcll(tm & ((1ull << lane) - 1));
    // The rows leave through LDS: a wavefront's rows are one contiguous span of the text pool (lane order), so they are staged
    // per lane and copied out by consecutive lanes -- whole 64-byte segments per store instead of 64 scattered single bytes
    // (which cost 39 bytes of HBM write traffic per byte written, by the PMC counters).
    __shared__ __attribute__((aligned(16))) char s_rows[4][SNP_STAGE];
    char *stage = s_rows[wv];
    const bool staged = wave_total <= SNP_STAGE;
    if (take) {
        // K-SNP is the first taker of a batch (heads zeroed before it, K-BUBBLE launched behind it on the stream): the group
        // head moves two bytes for every site it takes, so its group offset is twice its site offset
        pf_bubble_result r;
        r.rows_off = t_off;
        r.site_off = s_off;
        r.group_off = 2 * s_off;
        r.ilen_off = 0;
        r.n_rows = 2;
        r.n_cols = m;
        r.n_sites = 1;
        r.n_indel_len = 0;
        a.res[j] = r;
        if (t_off + 2ull * m <= a.text_cap && s_off + 1 <= a.site_cap && 2 * s_off + 2 <= a.group_cap) {
            // (two loops, so that the staged one stores through an LDS pointer: one pointer for both would be a generic one, flat stores)
            if (staged) {
                char *o = stage + my_excl;
                for (uint32_t c = 0; 32 * c < m; ++c) {
                    const uint64_t x0 = oriented_chunk(w0, m, (ov0 & 1) != 0, c), x1 = oriented_chunk(w1, m, (ov1 & 1) != 0, c);
                    const uint32_t e = m - 32 * c < 32 ? m - 32 * c : 32;
                    for (uint32_t q = 0; q < e; ++q) {
                        o[32 * c + q] = pf::base_char((uint32_t)((x0 >> (62 - 2 * q)) & 3));
                        o[m + 32 * c + q] = pf::base_char((uint32_t)((x1 >> (62 - 2 * q)) & 3));
                    }
                }
            } else {
                char *o = a.otext + t_off;
                for (uint32_t c = 0; 32 * c < m; ++c) {
                    const uint64_t x0 = oriented_chunk(w0, m, (ov0 & 1) != 0, c), x1 = oriented_chunk(w1, m, (ov1 & 1) != 0, c);
                    const uint32_t e = m - 32 * c < 32 ? m - 32 * c : 32;
                    for (uint32_t q = 0; q < e; ++q) {
                        o[32 * c + q] = pf::base_char((uint32_t)((x0 >> (62 - 2 * q)) & 3));
                        o[m + 32 * c + q] = pf::base_char((uint32_t)((x1 >> (62 - 2 * q)) & 3));
                    }
                }
            }
            a.ogroups[2 * s_off] = 1;
            a.ogroups[2 * s_off + 1] = 2;
            pf_bubble_site sr;
            sr.col = col;
            sr.is_indel = 0;
            sr.maxnum = 2;
            sr.pad_ = 0;
            a.osites[s_off] = sr;
        }
    }
    if (tm && staged && wb + wave_total <= a.text_cap) {
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
        __builtin_amdgcn_wave_barrier();
        // (in words: the stage is word-aligned, global memory takes the unaligned word; the last one to three bytes singly)
        const uint32_t n_words = wave_total >> 2;
        char *dst = a.otext + wb;
        for (uint32_t x = lane; x < n_words; x += WAVE) {
            const uint32_t w = reinterpret_cast<const uint32_t *>(stage)[x];
            __builtin_memcpy(dst + 4 * (size_t)x, &w, 4);
        }
        for (uint32_t x = (n_words << 2) + lane; x < wave_total; x += WAVE) dst[x] = stage[x];
    }
    // the rest: K-PAIR when short, else K-BUBBLE's queue of their size class
    int key = KEY_NONE;
    if (active && !take) key = (a.stack_ok >= 3 && m <= STACK_MAX) ? KEY_STACK : (a.pair_ok && m <= PAIR_MAX) ? KEY_PAIR : (a.pair_ok && m <= PAIR_MAX2) ? KEY_PAIR2 : 2 * bubble_class(m, m) + (m > 64 ? 0 : 1);
    block_append(key, j, a.lists, a.cnt);
}

// ---------------------------------------------------------------------------------------------------------------------
// K-PAIR (pf_pair_dev.hpp): SequenceAlignment of two paths of at most 64 (tier 1) / 128 (tier 2) bases, one thread per bubble

__device__ inline unsigned long long wave_take(unsigned long long *head, uint32_t mine, uint32_t &excl) {
    // exclusive prefix of `mine` over the wavefront and one atomic for the total; returns the wavefront's base
    const int lane = lane_id();
    uint32_t incl = mine;
    for (int o = 1; o < WAVE; o <<= 1) {
        const uint32_t x = __shfl_up(incl, o, WAVE);
        if (lane >= o) incl += x;
    }
    excl = incl - mine;
    const uint32_t total = __shfl(incl, WAVE - 1, WAVE);
    unsigned long long base = 0;
    if (total) {
        if (lane == 0) base = atomicAdd(head, (unsigned long long)total);
        base = ((unsigned long long)__shfl((uint32_t)(base >> 32), 0, WAVE) << 32) | __shfl((uint32_t)base, 0, WAVE);
    }
    return base;
}

// too few bubbles for the second tier to fill the device (a launch lasts as long as one wavefront's 64 bubbles whatever their
// number): they join K-BUBBLE's queues of their size classes instead
__global__ __launch_bounds__(256) void k_call_pair2_reroute(PairArgs a) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    int key = KEY_NONE;
    uint32_t j = 0;
    if (i < *a.n_list) {
        j = a.list[i];
        const CallTask &t = a.ct[a.kept[a.t0 + j]];
        const uint32_t m = a.len[t.inner[0] >> 1], n = a.len[t.inner[1] >> 1];
        key = 2 * bubble_class(m, m > n ? m : n) + (m > 64 || n > 64 ? 0 : 1);
    }
    wave_append(key, j, a.lists, a.cnt);
}

template <int NMAX, bool INTEGRAL>
__global__ __launch_bounds__(64, NMAX == 64 ? 3 : 2) void k_call_pair(PairArgs a) {
    using Gm = PairGeom<NMAX>;
    const int lane = lane_id();
    PairMem mem;
    uint8_t *g = a.scratch + (uint64_t)blockIdx.x * Gm::scratch_bytes;
    mem.dir = reinterpret_cast<uint32_t *>(g) + lane;
    mem.ra = reinterpret_cast<char *>(g + Gm::dir_bytes) + lane;
    mem.rb = mem.ra + 64ull * Gm::LEN;
    mem.fa = mem.rb + 64ull * Gm::LEN;
    mem.fb = mem.fa + 64ull * Gm::LEN;
    const uint32_t n_list = *a.n_list;
    for (uint32_t base = blockIdx.x * 64; base < n_list; base += gridDim.x * 64) {
        const uint32_t i = base + lane;
        const bool active = i < n_list;
        uint32_t j = 0, L = 0, n_sites = 0, n_ilen = 0;
        int defer_key = KEY_NONE;
        bool defer = false;
        if (active) {
            j = a.list[i];
            const CallTask &t = a.ct[a.kept[a.t0 + j]];
            const uint32_t ov0 = t.inner[0], ov1 = t.inner[1];
            const uint32_t m = a.len[ov0 >> 1], n = a.len[ov1 >> 1];
            unsigned long long tq = a.prof ? wall_clock64() : 0;
            auto mark = [&](int slot) {
                if (!a.prof) return;
                const unsigned long long now = wall_clock64();
                if (lane == 0) atomicAdd(&a.prof[slot], now - tq);
                tq = now;
            };
            uint64_t Aw[Gm::NA], Bw[Gm::NA];
            uint32_t b0[Gm::NA], b1[Gm::NA];
            const uint64_t *w0 = a.seq + a.off[ov0 >> 1], *w1 = a.seq + a.off[ov1 >> 1];
#pragma unroll
            for (int c = 0; c < Gm::NA; ++c) {
                Aw[c] = 32u * c < m ? oriented_chunk(w0, m, (ov0 & 1) != 0, (uint32_t)c) : 0;
                Bw[c] = 32u * c < n ? oriented_chunk(w1, n, (ov1 & 1) != 0, (uint32_t)c) : 0;
                pair_planes(Bw[c], b0[c], b1[c]);
            }
            const int dmin = n < m ? (int)n - (int)m : 0;
            mark(0);
            pair_fill<NMAX, INTEGRAL>(mem.dir, Aw, b0, b1, m, dmin, a.M, a.D, a.G, a.Mi, a.Di, a.Gi);
            mark(1);
            L = pair_traceback<NMAX>(mem, Aw, Bw, m, n, dmin);
            mark(2);
            if (L == 0) {   // several optimal paths, a gap-open budget in the way, or a walk outside the band: K-BUBBLE's queue of the bubble's size class
                defer = true;
                defer_key = 2 * bubble_class(m, m > n ? m : n) + (m > 64 || n > 64 ? 0 : 1);
            } else {
                const PairCounts pc = pair_classify<false>(mem.fa, mem.fb, L, nullptr, nullptr);
                n_sites = pc.n_sites;
                n_ilen = pc.n_indel_len;
            }
            mark(3);
        }
        const unsigned long long tp0 = a.prof ? wall_clock64() : 0;
        const bool take = active && !defer;
        // pool space for the whole wavefront: one atomic per pool
        uint32_t e_text, e_sites, e_groups, e_ilen;
        const unsigned long long b_text = wave_take(&a.heads[0], take ? 2 * L : 0, e_text);
        const unsigned long long b_sites = wave_take(&a.heads[1], take ? n_sites : 0, e_sites);
        const unsigned long long b_groups = wave_take(&a.heads[2], take ? 2 * n_sites : 0, e_groups);
        const unsigned long long b_ilen = wave_take(&a.heads[3], take ? n_ilen : 0, e_ilen);
        if (take) {
            const unsigned long long t_off = b_text + e_text, s_off = b_sites + e_sites, g_off = b_groups + e_groups, l_off = b_ilen + e_ilen;
            pf_bubble_result r;
            r.rows_off = t_off;
            r.site_off = s_off;
            r.group_off = g_off;
            r.ilen_off = l_off;
            r.n_rows = 2;
            r.n_cols = L;
            r.n_sites = n_sites;
            r.n_indel_len = n_ilen;
            a.res[j] = r;
            if (t_off + 2ull * L <= a.text_cap && s_off + n_sites <= a.site_cap && g_off + 2ull * n_sites <= a.group_cap && l_off + n_ilen <= a.ilen_cap) {
                char *o = a.otext + t_off;
                for (uint32_t c = 0; c < L; ++c) { o[c] = PF_AT(mem.fa, c); o[L + c] = PF_AT(mem.fb, c); }
                (void)pair_classify<true>(mem.fa, mem.fb, L, a.osites + s_off, a.oilen + l_off);
                for (uint32_t q = 0; q < n_sites; ++q) { a.ogroups[g_off + 2 * q] = 1; a.ogroups[g_off + 2 * q + 1] = 2; }
            }
        }
        const unsigned long long done_m = __ballot(take);
        if (lane == 0 && done_m) atomicAdd(a.n_done, (unsigned int)__popcll(done_m));
        wave_append(defer_key, j, a.lists, a.cnt);
        if (a.prof && lane == 0) { atomicAdd(&a.prof[4], wall_clock64() - tp0); atomicAdd(&a.prof[5], 1ull); }
    }
}

// The column pass over R rows of length L (character j of row r at rows[r * row_stride + j * col_stride]), src/SeqAlign.cpp:56-157 as K-BUBBLE's classify + publish
// restate it: which columns are sites, which of them open an indel, the allele groups by first appearance over the rows, the
// indel lengths.  Counted, or with EMIT written out.
struct ColumnCounts {
    uint32_t n_sites, n_ilen;
};
template <bool EMIT>
__device__ inline ColumnCounts classify_columns(const char *rows, size_t row_stride, size_t col_stride, uint32_t R, uint32_t L, pf_bubble_site *sites, uint8_t *groups,
                                           uint32_t *ilen) {
    uint32_t ns = 0, nl = 0, last_indel_pos = 0;
    bool open = false;
    uint32_t prev_gap = 0;   // bit r: row r had a gap in the previous column
    for (uint32_t j = 0; j < L; ++j) {
        uint32_t seen = 0, n_seen = 0, gap = 0;   // `seen`: one bit per character class (A C G T -)
        for (uint32_t r = 0; r < R; ++r) {
            const char c = rows[(size_t)r * row_stride + (size_t)j * col_stride];
            const uint32_t cls = c == '-' ? 4u : (((uint32_t)(unsigned char)c >> 1) & 3u);
            if (!((seen >> cls) & 1u)) { seen |= 1u << cls; ++n_seen; }
            gap |= (c == '-' ? 1u : 0u) << r;
        }
        const bool same_status = j > 0 && gap == prev_gap;
        const int t = n_seen > 1 ? (gap ? 2 : 1) : 0;
        bool site = false, opens = false;
        if (t != 2) {
            if (open) { if (EMIT) ilen[nl] = j - last_indel_pos; nl++; open = false; }
            if (t == 1) site = true;
        } else {
            const bool same_run = open && same_status;
            if (open && !same_run) { if (EMIT) ilen[nl] = j - last_indel_pos; nl++; }
            if (!same_run) { last_indel_pos = j; open = true; site = true; opens = true; }
            else if (n_seen > 2) site = true;
        }
        if (site) {
            if (EMIT) {
                uint8_t *grp = groups + (size_t)ns * R;
                uint32_t tab = 0, next = 0;   // group of character class c in nibble c
                for (uint32_t r = 0; r < R; ++r) {
                    const char c = rows[(size_t)r * row_stride + (size_t)j * col_stride];
                    const uint32_t cls = c == '-' ? 4u : (((uint32_t)(unsigned char)c >> 1) & 3u);
                    uint32_t gq = (tab >> (4 * cls)) & 15u;
                    if (!gq) { gq = ++next; tab |= gq << (4 * cls); }
                    grp[r] = (uint8_t)gq;
                }
                pf_bubble_site sr;
                sr.col = j;
                sr.is_indel = opens ? 1 : 0;
                sr.maxnum = (uint8_t)next;
                sr.pad_ = 0;
                sites[ns] = sr;
            }
            ns++;
        }
        prev_gap = gap;
    }
    return ColumnCounts{ns, nl};
}

// ---------------------------------------------------------------------------------------------------------------------
// K-STACK (pf_stack_dev.hpp): bubbles whose paths are all of one length -- the alignment is the paths stacked once every
// needlemanWunch(path 0, path p) is certified to have the diagonal as its single optimal path; one thread per bubble


__device__ inline void stack_load(const StackArgs &a, const pf_bubble_path &pp, StackPlanes &P) {
#pragma unroll
    for (int w = 0; w < 4; ++w) P.lo[w] = P.hi[w] = 0;
    if (pp.ov != NONE) {
        const uint64_t *w = a.seq + a.off[pp.ov >> 1];
#pragma unroll
        for (int c = 0; c < 4; ++c)
            if (32u * c < pp.len) pair_planes(oriented_chunk(w, pp.len, (pp.ov & 1) != 0, (uint32_t)c), P.lo[c], P.hi[c]);
    } else {
        const char *s = a.ptext + pp.text_off;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            uint32_t lo = 0, hi = 0;
            const uint32_t e = 32u * c < pp.len ? (pp.len - 32u * c < 32u ? pp.len - 32u * c : 32u) : 0u;
            for (uint32_t q = 0; q < e; ++q) {
                const uint32_t x = ((uint32_t)(unsigned char)s[32 * c + q] >> 1) & 3u;   // A 0, C 1, T 2, G 3
                const uint32_t code = x ^ (x >> 1);                                       // A 0, C 1, G 2, T 3
                lo |= (code & 1u) << q;
                hi |= (code >> 1) << q;
            }
            P.lo[c] = lo;
            P.hi[c] = hi;
        }
    }
}

__device__ inline uint32_t stack_code(const StackPlanes &P, uint32_t c) {
    uint32_t lo = P.lo[0], hi = P.hi[0];
#pragma unroll
    for (int w = 1; w < 4; ++w) { lo = (c >> 5) == (uint32_t)w ? P.lo[w] : lo; hi = (c >> 5) == (uint32_t)w ? P.hi[w] : hi; }
    return ((lo >> (c & 31)) & 1u) | (((hi >> (c & 31)) & 1u) << 1);
}

__global__ __launch_bounds__(64) void k_call_stack(StackArgs a) {
    const int lane = lane_id();
    const uint32_t n_list = *a.n_list;
    char *grows = reinterpret_cast<char *>(a.scratch + (uint64_t)blockIdx.x * stack_scratch_bytes()) + lane;   // rows of a gapped alignment, lane-interleaved
    constexpr size_t RS = (size_t)STACK_MAX * 64, CS = 64;
    for (uint32_t base = blockIdx.x * 64; base < n_list; base += gridDim.x * 64) {
        const uint32_t i = base + lane;
        const bool active = i < n_list;
        uint32_t j = 0, L = 0, n = 0, n_sites = 0, n_ilen = 0, l0 = 0, l1 = 0, lmax = 0;
        uint64_t first = 0;
        uint32_t U[4] = {0, 0, 0, 0};   // columns in which some path differs from path 0 (paths of one length)
        bool ok = false, gapped = false;
        if (active) {
            j = a.list[i];
            const pf_bubble_task bt = a.btask[j];
            n = bt.n_paths;
            first = bt.path_first;
            const pf_bubble_path p0 = a.bpath[first];
            l0 = L = lmax = p0.len;
            StackPlanes X, Y;
            stack_load(a, p0, X);
            // which kind: all paths as long as the first, or some shorter (a gap run in their rows); two paths may also have the
            // longer one second (strict bubbles are sorted by coverage): then row 0 takes the gaps
            for (uint32_t p = 1; p < n; ++p) {
                const uint32_t lp = a.bpath[first + p].len;
                if (p == 1) l1 = lp;
                if (lp != L) gapped = true;
                lmax = lp > lmax ? lp : lmax;
            }
            ok = true;
            if (!gapped) {
                for (uint32_t p = 1; p < n && ok; ++p) {
                    stack_load(a, a.bpath[first + p], Y);
                    ok = stack_certify(X, Y, L, a.M, a.D, a.G);
#pragma unroll
                    for (int w = 0; w < 4; ++w) U[w] |= (X.lo[w] ^ Y.lo[w]) | (X.hi[w] ^ Y.hi[w]);
                }
                n_sites = __popc(U[0]) + __popc(U[1]) + __popc(U[2]) + __popc(U[3]);
            } else if (n > STACK_GAP_ROWS) {
                ok = false;
            } else if (n == 2 && l1 > l0) {
                // the second path is the longer: the same certificate with the roles swapped (the recurrence is symmetric in its
                // two strings: UP and LEFT change places), and the gap run lies in row 0
                stack_load(a, a.bpath[first + 1], Y);
                const uint32_t d = l1 - l0, at = indel_place(Y, X, l0, d);
                ok = at != 0xFFFFFFFFu && indel_certify(Y, X, l1, l0, at, a.M, a.D, a.G);
                if (ok) {
                    L = l1;
                    for (uint32_t c = 0; c < L; ++c) {
                        grows[(size_t)c * CS] = (c < at) ? pf::base_char((uint32_t)(stack_code(X, c))) : (c < at + d ? '-' : pf::base_char((uint32_t)(stack_code(X, c - d))));
                        grows[RS + (size_t)c * CS] = pf::base_char((uint32_t)(stack_code(Y, c)));
                    }
                }
            } else {
                for (uint32_t c = 0; c < L; ++c) grows[(size_t)c * CS] = pf::base_char((uint32_t)(stack_code(X, c)));
                for (uint32_t p = 1; p < n && ok; ++p) {
                    const pf_bubble_path pp = a.bpath[first + p];
                    stack_load(a, pp, Y);
                    char *row = grows + (size_t)p * RS;
                    if (pp.len == L) {
                        ok = stack_certify(X, Y, L, a.M, a.D, a.G);
                        if (ok) for (uint32_t c = 0; c < L; ++c) row[(size_t)c * CS] = pf::base_char((uint32_t)(stack_code(Y, c)));
                    } else if (pp.len < L) {
                        const uint32_t d = L - pp.len, at = indel_place(X, Y, pp.len, d);
                        ok = at != 0xFFFFFFFFu && indel_certify(X, Y, L, pp.len, at, a.M, a.D, a.G);
                        if (ok) for (uint32_t c = 0; c < L; ++c) row[(size_t)c * CS] = (c < at) ? pf::base_char((uint32_t)(stack_code(Y, c))) : (c < at + d ? '-' : pf::base_char((uint32_t)(stack_code(Y, c - d))));
                    } else {
                        ok = false;   // a later path longer than the first: row 0 would take a gap
                    }
                }
            }
            if (ok && gapped) {
                const ColumnCounts tc = classify_columns<false>(grows, RS, CS, n, L, nullptr, nullptr, nullptr);
                n_sites = tc.n_sites;
                n_ilen = tc.n_ilen;
            }
        }
        const bool take = active && ok;
        uint32_t e_text, e_sites, e_groups, e_ilen;
        const unsigned long long b_text = wave_take(&a.heads[0], take ? n * L : 0, e_text);
        const unsigned long long b_sites = wave_take(&a.heads[1], take ? n_sites : 0, e_sites);
        const unsigned long long b_groups = wave_take(&a.heads[2], take ? n * n_sites : 0, e_groups);
        const unsigned long long b_ilen = wave_take(&a.heads[3], take ? n_ilen : 0, e_ilen);
        if (take) {
            const unsigned long long t_off = b_text + e_text, s_off = b_sites + e_sites, g_off = b_groups + e_groups, l_off = b_ilen + e_ilen;
            pf_bubble_result r;
            r.rows_off = t_off;
            r.site_off = s_off;
            r.group_off = g_off;
            r.ilen_off = l_off;
            r.n_rows = n;
            r.n_cols = L;
            r.n_sites = n_sites;
            r.n_indel_len = n_ilen;
            a.res[j] = r;
            const bool room = t_off + (uint64_t)n * L <= a.text_cap && s_off + n_sites <= a.site_cap && g_off + (uint64_t)n * n_sites <= a.group_cap &&
                              l_off + n_ilen <= a.ilen_cap;
            if (room && gapped) {
                char *o = a.otext + t_off;
                for (uint32_t p = 0; p < n; ++p)
                    for (uint32_t c = 0; c < L; ++c) o[(size_t)p * L + c] = grows[(size_t)p * RS + (size_t)c * CS];
                (void)classify_columns<true>(grows, RS, CS, n, L, a.osites + s_off, a.ogroups + g_off, a.oilen + l_off);
            } else if (room) {
                // the rows, and per variant column the bases of all rows
                for (uint32_t p = 0; p < n; ++p) {
                    const pf_bubble_path pp = a.bpath[first + p];
                    StackPlanes Y;
                    stack_load(a, pp, Y);
                    char *o = a.otext + t_off + (uint64_t)p * L;
                    for (uint32_t c = 0; c < L; ++c) o[c] = pf::base_char((uint32_t)(stack_code(Y, c)));
                    // this row's base in every variant column, kept in the group bytes for now
                    uint32_t q = 0;
#pragma unroll
                    for (int w = 0; w < 4; ++w) {
                        uint32_t m = U[w];
                        while (m) {
                            const uint32_t c = 32u * w + (uint32_t)__ffs((int)m) - 1;
                            m &= m - 1;
                            a.ogroups[g_off + (uint64_t)q * n + p] = (uint8_t)stack_code(Y, c);
                            ++q;
                        }
                    }
                }
                // groups numbered by first appearance over the rows (src/SeqAlign.cpp:59-120), the site records
                uint32_t q = 0;
#pragma unroll
                for (int w = 0; w < 4; ++w) {
                    uint32_t m = U[w];
                    while (m) {
                        const uint32_t c = 32u * w + (uint32_t)__ffs((int)m) - 1;
                        m &= m - 1;
                        uint32_t tab = 0, maxnum = 0;   // group of base b in byte b
                        uint8_t *gp = a.ogroups + g_off + (uint64_t)q * n;
                        for (uint32_t p = 0; p < n; ++p) {
                            const uint32_t b = gp[p];
                            uint32_t gr = (tab >> (8 * b)) & 0xFFu;
                            if (!gr) { gr = ++maxnum; tab |= gr << (8 * b); }
                            gp[p] = (uint8_t)gr;
                        }
                        pf_bubble_site sr;
                        sr.col = c;
                        sr.is_indel = 0;
                        sr.maxnum = (uint8_t)maxnum;
                        sr.pad_ = 0;
                        a.osites[s_off + q] = sr;
                        ++q;
                    }
                }
            }
        }
        const unsigned long long done_m = __ballot(take);
        if (lane == 0 && done_m) atomicAdd(&a.cnt->n_stack_done, (unsigned int)__popcll(done_m));
        // not certified: two paths to K-PAIR (its fill decides, or finds the tie), the others to K-BUBBLE's queue of the bubble's
        // size class
        int key = KEY_NONE;
        if (active && !ok) {
            if (n == 2 && a.pair_ok && pair_fits<PAIR_MAX>(l0, l1)) key = KEY_PAIR;
            else if (n == 2 && a.pair_ok && pair_fits<PAIR_MAX2>(l0, l1)) key = KEY_PAIR2;
            else if (n == 2) key = 2 * bubble_class(l0, lmax) + (lmax > 64 ? 0 : 1);
            else key = 2 * bubble_class(l0, lmax);
        }
        wave_append(key, j, a.lists, a.cnt);
    }
}

// the forms pf_call.hip launches
template __global__ void k_call_pair<PAIR_MAX, true>(PairArgs);
template __global__ void k_call_pair<PAIR_MAX, false>(PairArgs);
template __global__ void k_call_pair<PAIR_MAX2, true>(PairArgs);
template __global__ void k_call_pair<PAIR_MAX2, false>(PairArgs);

}  // namespace pf_call
