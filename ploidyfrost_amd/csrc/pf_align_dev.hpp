// Device side of K-ALN shared by k_align (pf_align.hip) and k_bubble (pf_bubble.hip): the
// Needleman-Wunsch fill + all-co-optimal traceback of one pairwise job on one wavefront
// (reference src/SeqAlign.cpp:480-549, 306-478).  See pf_align.hip for the description.
#pragma once
#include <hip/hip_runtime.h>
#include <limits.h>

#include <algorithm>
#include <stdint.h>

#include "pf_device_common.hpp"
#include "ploidyfrost_hip.h"

namespace pf {

enum : uint8_t { UP = 1, DIAG = 2, LEFT = 4 };

__host__ __device__ inline uint32_t al4(uint32_t x) { return (x + 3u) & ~3u; }

// bytes of working storage one job needs (must match the carve-up in align_job)
// The direction matrix takes four bits a cell (the three flags of src/SeqAlign.cpp:497-549), two cells a byte, every row on a byte
// boundary: row i begins at byte i * dir_row_bytes(n), cell (i, j) is nibble j & 1 of byte j >> 1 of its row.  (Rounds 1-3 kept a
// byte a cell: the flags, and above them the flags the depth-first traceback had not tried yet; those now live on the walk's own
// stack.  Half the matrix = twice the wavefronts of a size class in a CU's LDS.)
__host__ __device__ inline uint32_t dir_row_bytes(uint32_t n) { return (n + 2u) >> 1; }
__host__ __device__ inline uint64_t job_bytes(uint32_t m, uint32_t n) {
    const uint64_t cells = (uint64_t)(m + 1) * dir_row_bytes(n);
    const uint64_t mx = m > n ? m : n;
    return ((cells + 3) & ~3ull) + 12ull * (mx + 1) + al4(m) + al4(n) + 3ull * al4(m + n) + al4(2 * (m + n)) + 16;
}

struct AlnScratch {  // per-wave staging of the kept alignments (global memory)
    char *text;
    uint32_t *gaps;
    pf_align_hit *hits;
    uint32_t text_cap, gap_cap, hit_cap;
};

struct AlnOut {
    uint64_t *hit_first;
    uint32_t *hit_count;
    pf_align_hit *hits;
    uint64_t hit_cap;
    char *text;
    uint64_t text_cap;
    uint32_t *gaps;
    uint64_t gap_cap;
    unsigned long long *heads;  // [0] hits, [1] text bytes, [2] gap entries (running totals)
    uint32_t *retry;            // jobs whose staging overflowed
    unsigned int *n_retry;
};

__device__ inline void aln_sync() {
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
    __builtin_amdgcn_wave_barrier();
}

// The fill of align_job (reference src/SeqAlign.cpp:497-549), compiled once for integral scores -- the default 2 / -1 / -3: the
// reference's `int = long + double` is plain integer arithmetic -- and once for the fp64 form with its truncation: the choice is made
// per job, outside the loop over the anti-diagonals (inside it the branch, and the fp64 instructions of the road not taken, were
// issued at every step).
template <bool INTEGRAL>
__device__ inline void nw_fill(uint8_t *dir, int *s0, const char *A, const char *B, uint32_t m, uint32_t n, double M, double D, double G) {
    const int lane = lane_id();
    const uint32_t RB = dir_row_bytes(n);
    // Rows are processed in blocks of 64, one row per lane; inside a block the anti-diagonal
    // wavefront lives in registers: at step t lane l owns cell (row, t - l + 1), its left neighbour is
    // its own previous cell, the cells above / above-left are what lane l-1 produced one / two steps
    // earlier (one DPP wave shift each).  LDS sees only the flag byte of each finished cell and, for
    // m > 64, the last row of a block (read back by lane 0 of the next block).
    int *brow_s = s0;
    uint8_t *brow_f = reinterpret_cast<uint8_t *>(s0 + (n + 1));
    const uint32_t n_blocks = (m + 63) / 64;
    for (uint32_t blk = 0; blk < n_blocks; ++blk) {
        const uint32_t row0 = blk * 64;  // the row above this block
        const uint32_t r = row0 + lane + 1;
        int pend = UP;   // the flags of the cell before the one being computed: column 0 of a row below the first is the border's UP
        const bool row_ok = r <= m;
        const char a = row_ok ? A[r - 1] : '\0';
        const bool next_is_gap = row_ok && r != m && A[r] == '-';  // the look-ahead of :528-532
        const uint32_t rows_here = m - row0 < 64 ? m - row0 : 64;
        const bool feeds_next = blk + 1 < n_blocks && lane == 63;
        // integral scores (the default 2 / -1 / -3): the reference's `int = long + double` is plain
        // integer arithmetic; otherwise the fp64 form with its truncation is kept
        const int Mi = (int)M, Di = (int)D, Gi = (int)G;
        int last_s = INTEGRAL ? Gi * (int)r : (int)(long)(G * (double)r), last2_s = 0;  // (r, 0): border, flag Up
        int last_f = UP, last2_f = 0;
        int carry_s = blk == 0 ? 0 : (INTEGRAL ? Gi * (int)row0 : (int)(long)(G * (double)row0));  // (row0, 0)
        int carry_f = blk == 0 ? 0 : UP;
        int b_reg = 0;
        const uint32_t steps = n + rows_here - 1;
        for (uint32_t t = 0; t < steps; ++t) {
            const int b_in = t < n ? (int)B[t] : 0;
            b_reg = __builtin_amdgcn_update_dpp(0, b_reg, 0x138, 0xf, 0xf, false);  // wave_shr:1
            int up_s = __builtin_amdgcn_update_dpp(0, last_s, 0x138, 0xf, 0xf, false);
            int up_f = __builtin_amdgcn_update_dpp(0, last_f, 0x138, 0xf, 0xf, false);
            int dg_s = __builtin_amdgcn_update_dpp(0, last2_s, 0x138, 0xf, 0xf, false);
            int dg_f = __builtin_amdgcn_update_dpp(0, last2_f, 0x138, 0xf, 0xf, false);
            const int j = (int)t - lane + 1;
            if (lane == 0) {
                b_reg = b_in;
                dg_s = carry_s;
                dg_f = carry_f;
                if (t < n) {
                    if (blk == 0) { up_s = INTEGRAL ? Gi * j : (int)(long)(G * (double)j); up_f = LEFT; }
                    else { up_s = brow_s[j]; up_f = brow_f[j]; }
                }
                carry_s = up_s;
                carry_f = up_f;
            }
            if (row_ok && j >= 1 && j <= (int)n) {
                const char b = (char)b_reg;
                int up, dg, lf;
                if (INTEGRAL) {
                    up = up_s + Gi;
                    dg = dg_s + (a == b ? Mi : ((a == '-' || b == '-') ? Gi : Di));
                    lf = last_s + Gi;
                } else {
                    const double sub = a == b ? M : ((a == '-' || b == '-') ? G : D);
                    up = (int)((double)up_s + G);
                    dg = (int)((double)dg_s + sub);
                    lf = (int)((double)last_s + G);
                }
                if (up_f & UP) up += 1;
                if (dg_f & DIAG) dg += 1;
                if (last_f & LEFT) lf += 1;
                int best = up > dg ? up : dg;
                best = best > lf ? best : lf;
                if (best == lf && next_is_gap) {
                    lf = INT_MIN;
                    best = up > dg ? up : dg;
                }
                int f = 0;
                if (up == best) f |= UP;
                if (dg == best) f |= DIAG;
                if (lf == best) f |= LEFT;
                last2_s = last_s;
                last2_f = last_f;
                last_s = best;
                last_f = f;
                // one store a step, no branch: an even column writes its byte with the upper half empty, the odd column after it writes
                // the same byte again with both halves (cells j - 1 and j of this lane's row)
                dir[r * RB + ((uint32_t)j >> 1)] = (uint8_t)((j & 1) ? (pend | (f << 4)) : f);
                pend = f;
                if (feeds_next) { brow_s[j] = best; brow_f[j] = (uint8_t)f; }
            }
        }
        aln_sync();
    }

}

// The same fill for integral scores small enough that a cell's score and its three direction flags share one register
// (score << 3 | flags; |score| <= (max |M|, |D|, |G| + 1) * (i + j) < 2^27 is the caller's condition): three DPP shifts per
// anti-diagonal step instead of five, half the register moves around them.  Same cells, same values, same flags.
__device__ inline void nw_fill_packed(uint8_t *dir, int *s0, const char *A, const char *B, uint32_t m, uint32_t n, int Mi, int Di, int Gi) {
    const int lane = lane_id();
    const uint32_t RB = dir_row_bytes(n);
    int *brow_s = s0;
    uint8_t *brow_f = reinterpret_cast<uint8_t *>(s0 + (n + 1));
    const uint32_t n_blocks = (m + 63) / 64;
    for (uint32_t blk = 0; blk < n_blocks; ++blk) {
        const uint32_t row0 = blk * 64;
        const uint32_t r = row0 + lane + 1;
        int pend = UP;   // the flags of the cell before the one being computed: column 0 of a row below the first is the border's UP
        const bool row_ok = r <= m;
        const char a = row_ok ? A[r - 1] : '\0';
        const bool next_is_gap = row_ok && r != m && A[r] == '-';
        const uint32_t rows_here = m - row0 < 64 ? m - row0 : 64;
        const bool feeds_next = blk + 1 < n_blocks && lane == 63;
        int last = ((Gi * (int)r) << 3) | UP, last2 = 0;                   // (r, 0): border, flag Up
        int carry = blk == 0 ? 0 : (((Gi * (int)row0) << 3) | UP);          // (row0, 0)
        int b_reg = 0;
        const uint32_t steps = n + rows_here - 1;
        for (uint32_t t = 0; t < steps; ++t) {
            // What lane 0 takes in at this step -- B's next character, and the cell above its own (row row0: the border, or the last
            // row of the block before, from LDS) -- is handed to the shifts as their `old` operand: a wave_shr leaves lane 0, which has
            // no lane to read from, with exactly that.  (It was a branch of lane 0's own at every step.)
            const uint32_t j0 = t < n ? t + 1 : n;   // lane 0's column (t >= n: lane 0 has left its row; any value will do)
            const int b_in = (int)B[j0 - 1];
            const int up_in = blk == 0 ? (((Gi * (int)j0) << 3) | LEFT) : ((brow_s[j0] << 3) | (int)brow_f[j0]);
            b_reg = __builtin_amdgcn_update_dpp(b_in, b_reg, 0x138, 0xf, 0xf, false);  // wave_shr:1
            const int up_sf = __builtin_amdgcn_update_dpp(up_in, last, 0x138, 0xf, 0xf, false);
            const int dg_sf = __builtin_amdgcn_update_dpp(carry, last2, 0x138, 0xf, 0xf, false);
            carry = up_in;   // (the cell above this step's is the cell above-left of the next step's)
            const int j = (int)t - lane + 1;
            if (row_ok && j >= 1 && j <= (int)n) {
                const char b = (char)b_reg;
                const int up = (up_sf >> 3) + Gi + (up_sf & 1);                                   // UP = 1
                const int dg = (dg_sf >> 3) + (a == b ? Mi : ((a == '-' || b == '-') ? Gi : Di)) + ((dg_sf >> 1) & 1);   // DIAG = 2
                int lf = (last >> 3) + Gi + ((last >> 2) & 1);                                   // LEFT = 4
                const int ud = up > dg ? up : dg;
                int best = ud > lf ? ud : lf;
                if (best == lf && next_is_gap) {
                    lf = INT_MIN;
                    best = ud;
                }
                int f = 0;
                if (up == best) f |= UP;
                if (dg == best) f |= DIAG;
                if (lf == best) f |= LEFT;
                last2 = last;
                last = (best << 3) | f;
                // one store a step, no branch: an even column writes its byte with the upper half empty, the odd column after it writes
                // the same byte again with both halves (cells j - 1 and j of this lane's row)
                dir[r * RB + ((uint32_t)j >> 1)] = (uint8_t)((j & 1) ? (pend | (f << 4)) : f);
                pend = f;
                if (feeds_next) { brow_s[j] = best; brow_f[j] = (uint8_t)f; }
            }
        }
        aln_sync();
    }
}

// One job on one wavefront.  `base` = working storage (LDS or global), job_bytes(m, n) bytes.
// Returns false when the staging area overflowed.
__device__ inline bool align_job(uint8_t *base, const char *__restrict__ ga, const char *__restrict__ gb, uint32_t m, uint32_t n,
                          double M, double D, double G, int integral, const AlnScratch &sc, uint32_t &n_hits_out,
                          uint32_t &text_out, uint32_t &gaps_out, unsigned long long *prof = nullptr) {
    const int lane = lane_id();
    const uint32_t RB = dir_row_bytes(n);
    const uint32_t cells = (m + 1) * RB;
    uint8_t *dir = base;
    // the three flags of cell (i, j)
    auto flags_at = [&](uint32_t i, uint32_t j) -> uint8_t { return (uint8_t)((dir[i * RB + (j >> 1)] >> ((j & 1u) << 2)) & 7u); };
    int *s0 = reinterpret_cast<int *>(base + ((cells + 3) & ~3u));
    const uint32_t mx = m > n ? m : n;
    char *A = reinterpret_cast<char *>(s0 + 3 * (mx + 1));
    char *B = A + al4(m);
    char *ra = B + al4(n);
    char *rb = ra + al4(m + n);
    uint8_t *mv = reinterpret_cast<uint8_t *>(rb + al4(m + n));
    uint16_t *gp = reinterpret_cast<uint16_t *>(mv + al4(m + n));

    const unsigned long long pc0 = prof ? wall_clock64() : 0;
    for (uint32_t i = lane; i < m; i += WAVE) A[i] = ga[i];
    for (uint32_t j = lane; j < n; j += WAVE) B[j] = gb[j];
    // borders (src/SeqAlign.cpp:486-496)
    // (row 0: LEFT from column 1 on; column 0 of the rows below, UP, is written by the fill together with their column 1)
    for (uint32_t q = lane; q < RB; q += WAVE)
        dir[q] = (uint8_t)((q != 0 && 2 * q <= n ? LEFT : 0) | ((2 * q + 1 <= n ? LEFT : 0) << 4));
    if (n == 0)
        for (uint32_t i = 1 + lane; i <= m; i += WAVE) dir[i * RB] = UP;
    aln_sync();
    // ---- fill ---------------------------------------------------------------------------------
    if (integral) {
        const long Ml = (long)M, Dl = (long)D, Gl = (long)G;
        const long mag = std::max(std::max(Ml < 0 ? -Ml : Ml, Dl < 0 ? -Dl : Dl), Gl < 0 ? -Gl : Gl) + 1;   // (+ 1: the bonus for going on in a direction)
        if (mag * (long)(m + n + 2) < (1L << 27)) nw_fill_packed(dir, s0, A, B, m, n, (int)M, (int)D, (int)G);
        else nw_fill<true>(dir, s0, A, B, m, n, M, D, G);
    } else {
        nw_fill<false>(dir, s0, A, B, m, n, M, D, G);
    }

    const unsigned long long pc1 = prof ? wall_clock64() : 0;
    // ---- traceback ------------------------------------------------------------------------------
    uint32_t n_hits = 0, text_used = 0, gaps_used = 0;
    bool overflow = false;
    // Fast path (all lanes, wave-uniform): most matrices have exactly one optimal path -- every cell on
    // it carries a single direction flag.  Then the reference's DFS descends it once, emits that one
    // alignment and unwinds without finding an alternative, so walking the path is the whole
    // traceback.  Any cell with two flags, or a gap move at an exhausted budget, falls back to the full
    // DFS below on the untouched matrix.
    bool unique = true;
    uint32_t f_len = 0, f_ng = 0;
    {
        uint32_t i = m, j = n, oa = 0, ob = 0;
        char fa = '\0', fb = '\0';
        while (i > 0 || j > 0) {
            const uint8_t f = flags_at(i, j);
            char ca, cb;
            if (f == DIAG) {
                ca = A[i - 1];
                cb = B[j - 1];
                i--;
                j--;
            } else if (f == LEFT) {
                if (oa >= 5) { unique = false; break; }
                if (f_len == 0 || fa != '+') ++oa;
                ca = '+';
                cb = B[j - 1];
                if (lane == 0) gp[f_ng] = (uint16_t)i;
                f_ng++;
                j--;
            } else if (f == UP) {
                if (ob >= 5) { unique = false; break; }
                if (f_len == 0 || fb == '-') ++ob;
                ca = A[i - 1];
                cb = '-';
                i--;
            } else {
                unique = false;
                break;
            }
            if (lane == 0) { ra[f_len] = ca; rb[f_len] = cb; }
            fa = ca;
            fb = cb;
            f_len++;
        }
    }
    // One complete path on the stack (ra / rb / gp, back to front), the only alignment so far: variantAnalyze (src/SeqAlign.cpp:237-305)
    // lane-parallel over its columns, the rows and the gap positions copied out by all lanes, hit 0 written.
    auto keep_only_path = [&](uint32_t f_len, uint32_t f_ng) {
        long long isum = 0, score = 0;
        uint32_t npos = 0, nind = 0;
        const int Mi = (int)M, Di = (int)D, Gi = (int)G;
        for (uint32_t base0 = 0; base0 < f_len; base0 += WAVE) {
            const uint32_t t = base0 + lane;
            if (t < f_len) {
                const char ax = ra[f_len - 1 - t], a = ax == '+' ? '-' : ax, b = rb[f_len - 1 - t];
                if (integral) isum += (a == '-' || b == '-') ? Gi : (a == b ? Mi : Di);
                if (a != b) {
                    const int side = a == '-' ? 1 : (b == '-' ? 2 : 3);
                    if (side == 3) {
                        npos++;
                    } else {
                        int prev_run = 0;
                        if (t > 0) {
                            const char px = ra[f_len - t], pa = px == '+' ? '-' : px, pb = rb[f_len - t];
                            if (pa != pb) prev_run = pa == '-' ? 1 : (pb == '-' ? 2 : 0);
                        }
                        if (prev_run != side) { npos++; nind++; }
                    }
                }
            }
        }
        npos = __shfl((uint32_t)wave_sum_u64(npos), 0, WAVE);
        nind = __shfl((uint32_t)wave_sum_u64(nind), 0, WAVE);
        if (integral) {
            score = (long long)wave_sum_u64((uint64_t)isum);
        } else if (lane == 0) {
            for (uint32_t t = f_len; t-- > 0;) {
                const char a = ra[t] == '+' ? '-' : ra[t], b = rb[t];
                const double sx = (a == '-' || b == '-') ? G : (a == b ? M : D);
                score = (long long)((double)score + sx);
            }
        }
        if (2 * f_len > sc.text_cap || f_ng > sc.gap_cap || sc.hit_cap < 1) {
            overflow = true;
        } else {
            char *ta = sc.text, *tb = ta + f_len;
            for (uint32_t t = lane; t < f_len; t += WAVE) {
                const char ax = ra[f_len - 1 - t];
                ta[t] = ax == '+' ? '-' : ax;
                tb[t] = rb[f_len - 1 - t];
            }
            for (uint32_t t = lane; t < f_ng; t += WAVE) sc.gaps[t] = gp[t];
            if (lane == 0) {
                pf_align_hit h;
                h.text_off = 0;
                h.gap_off = 0;
                h.len = f_len;
                h.n_gaps = f_ng;
                h.score = score;
                h.n_pos = npos;
                h.n_indel = nind;
                sc.hits[0] = h;
            }
            n_hits = 1;
            text_used = 2 * f_len;
            gaps_used = f_ng;
        }
    };
    // the depth-first walk's first complete path, when the walk ended without touching its stack again (below): kept by all lanes
    uint32_t pending = 0, pend_len = 0, pend_ng = 0;
    if (unique) {
        aln_sync();
        keep_only_path(f_len, f_ng);
    } else if (lane == 0) {
        // one lane, a chain of dependent steps: ahead of the other wavefronts of the SIMD at issue (their fills are throughput work)
        __builtin_amdgcn_s_setprio(3);
        unsigned long long st_steps = 0, st_leaves = 0, st_takes = 0, st_leaf_ticks = 0;   // diagnostic (prof)
        uint64_t open_a = 0, open_b = 0, lim_a = 5, lim_b = 5;  // size_t in the reference
        uint32_t len = 0, ng = 0;
        uint32_t i = m, j = n;
        bool have = false;
        long long last_score = 0;
        uint32_t last_npos = 0, last_indel = 0;
        char top_a = '\0', top_b = '\0';   // ra[len - 1], rb[len - 1] ('\0' on an empty stack)
        // cells below the current one on the stack that still have a direction to try.  None, and nothing left at the current cell:
        // all that remains of the walk is its unwinding, which finds nothing and whose restored flags nobody will read -- the walk
        // ends there (a hundred steps of the two hundred a job takes: one complete path, kept, and the way back)
        uint32_t alts = 0;
        // The directions still to be tried at the current cell (the reference's matrix_temp entry, src/SeqAlign.cpp:306-478): the
        // cell's flags when the walk arrives, less what it has tried from there.  For the cells below on the stack they are kept with
        // the move that left them (mv: move in the low nibble, what was left to try in the high one) -- the matrix itself is only
        // written when a gap move is refused for lack of budget, which removes that direction from the cell for good.
        uint8_t work = flags_at(i, j);
        auto drop_flag = [&](uint32_t ci, uint32_t cj, uint8_t f) {
            uint8_t &b = dir[ci * RB + (cj >> 1)];
            b = (uint8_t)(b & ~(f << ((cj & 1u) << 2)));
        };
        for (;;) {
            if (prof) ++st_steps;
            if (i == 0 && j == 0 && open_a <= lim_a && open_b <= lim_b && !have) {
                // The first complete path is kept whatever it scores.  Nineteen walks of twenty find no second one and end a few steps
                // later without touching the stack again (alts below): its analysis and its copy -- two loops over its columns on this
                // one lane, a quarter of the walk's time -- are left to all lanes behind the walk (keep_only_path).  A walk that goes
                // on does them itself, as before, before it takes the first entry off the stack (pend_flush).
                if (prof) { ++st_leaves; ++st_takes; }
                if (sc.hit_cap < 1 || 2 * len > sc.text_cap || ng > sc.gap_cap) {
                    overflow = true;
                    break;
                }
                pending = 1; pend_len = len; pend_ng = ng;
                n_hits = 1;
                text_used = 2 * len;
                gaps_used = ng;
                lim_a = open_a;
                lim_b = open_b;
                have = true;
            } else if (i == 0 && j == 0 && open_a <= lim_a && open_b <= lim_b) {
                const unsigned long long tl0 = prof ? wall_clock64() : 0;
                if (prof) ++st_leaves;
                // variantAnalyze over the forward strings (stored back to front)
                long long score = 0;
                uint32_t npos = 0, indel = 0;
                uint8_t run = 0;
                for (uint32_t t = len; t-- > 0;) {
                    const char a = ra[t] == '+' ? '-' : ra[t];
                    const char b = rb[t];
                    const double s = (a == '-' || b == '-') ? G : (a == b ? M : D);
                    score = (long long)((double)score + s);
                    if (a != b) {
                        if (a == '-') { if (run != 1) { run = 1; indel++; npos++; } }
                        else if (b == '-') { if (run != 2) { run = 2; indel++; npos++; } }
                        else { run = 0; npos++; }
                    } else {
                        run = 0;
                    }
                }
                bool take = true;
                if (have) {
                    long long diff;  // last - this (src/SeqAlign.hpp:43-67)
                    if (last_score == score) {
                        if (last_npos == npos) diff = last_indel == indel ? 0 : (long long)indel - (long long)last_indel;
                        else diff = (long long)npos - (long long)last_npos;
                    } else {
                        diff = last_score > score ? 1 : -1;
                    }
                    const int d32 = (int)diff;
                    if (d32 < 0) { n_hits = 0; text_used = 0; gaps_used = 0; }
                    else if (d32 > 0) take = false;
                }
                if (take) {
                    if (n_hits >= sc.hit_cap || text_used + 2 * len > sc.text_cap || gaps_used + ng > sc.gap_cap) {
                        overflow = true;
                        break;
                    }
                    pf_align_hit h;
                    h.text_off = text_used;
                    h.gap_off = gaps_used;
                    h.len = len;
                    h.n_gaps = ng;
                    h.score = score;
                    h.n_pos = npos;
                    h.n_indel = indel;
                    sc.hits[n_hits++] = h;
                    char *ta = sc.text + text_used, *tb = ta + len;
                    for (uint32_t t = 0; t < len; ++t) {
                        const char a = ra[len - 1 - t];
                        ta[t] = a == '+' ? '-' : a;
                        tb[t] = rb[len - 1 - t];
                    }
                    for (uint32_t t = 0; t < ng; ++t) sc.gaps[gaps_used + t] = gp[t];
                    text_used += 2 * len;
                    gaps_used += ng;
                    lim_a = open_a;
                    lim_b = open_b;
                    have = true;
                    last_score = score;
                    last_npos = npos;
                    last_indel = indel;
                    if (prof) ++st_takes;
                }
                if (prof) st_leaf_ticks += wall_clock64() - tl0;
            }
            if (work == DIAG) {
                // The commonest step by far, in a loop of its own: a run of cells whose only open direction is the diagonal (the
                // general step below does exactly this for each of them, behind its tests for the leaf and the two gap moves)
                char na2, nb2;
                for (;;) {
                    na2 = A[i - 1];
                    nb2 = B[j - 1];
                    ra[len] = na2;
                    rb[len] = nb2;
                    mv[len] = DIAG;   // (nothing left to try at the cell that is left)
                    ++len;
                    --i;
                    --j;
                    work = flags_at(i, j);
                    if (i == 0 || j == 0) break;   // (a border cell: the general step)
                    if (work != DIAG) break;
                    if (prof) ++st_steps;
                }
                top_a = na2; top_b = nb2;
                continue;
            }
            const char na = i ? A[i - 1] : '\0', nb = j ? B[j - 1] : '\0';
            const char fa = top_a;
            const char fb = top_b;
            if (work & LEFT) {
                bool go;
                if (open_a < lim_a) {
                    if (len == 0 || fa != '+') ++open_a;
                    go = true;
                } else if (open_a == lim_a) {
                    go = fa == '+';
                } else {
                    go = false;
                }
                work &= (uint8_t) ~LEFT;
                if (!go) {
                    drop_flag(i, j, LEFT);
                    continue;
                }
                alts += work != 0;
                ra[len] = '+';
                rb[len] = nb;
                top_a = '+';
                top_b = nb;
                mv[len] = (uint8_t)(LEFT | (work << 4));
                gp[ng++] = (uint16_t)i;
                len++;
                j -= 1;
                work = flags_at(i, j);
            } else if (work & UP) {
                bool go;
                if (open_b < lim_b) {
                    if (len == 0 || fb == '-') ++open_b;
                    go = true;
                } else if (open_b == lim_b) {
                    go = fb == '-';
                } else {
                    go = false;
                }
                work &= (uint8_t) ~UP;
                if (!go) {
                    drop_flag(i, j, UP);
                    continue;
                }
                alts += work != 0;
                ra[len] = na;
                rb[len] = '-';
                top_a = na;
                top_b = '-';
                mv[len] = (uint8_t)(UP | (work << 4));
                len++;
                i -= 1;
                work = flags_at(i, j);
            } else if (work & DIAG) {
                ra[len] = na;
                rb[len] = nb;
                top_a = na;
                top_b = nb;
                mv[len] = DIAG;
                len++;
                i -= 1;
                j -= 1;
                work = flags_at(i, j);
            } else {
                // nothing left to try at this cell: back up -- and on, in this loop, through every cell that has nothing left either
                // (the unwinding behind a complete path: a hundred cells with one direction each)
                char ta = fa, tb = fb;
                bool out = false;
                for (;;) {
                    if (len == 0 || alts == 0) { out = true; break; }
                    if (pending) {
                        // the walk goes on: the first complete path, still whole on the stack, is analysed and copied out here
                        const unsigned long long tl0 = prof ? wall_clock64() : 0;
                        long long score = 0;
                        uint32_t npos = 0, indel = 0;
                        uint8_t run = 0;
                        for (uint32_t t = pend_len; t-- > 0;) {
                            const char a = ra[t] == '+' ? '-' : ra[t];
                            const char b = rb[t];
                            const double s = (a == '-' || b == '-') ? G : (a == b ? M : D);
                            score = (long long)((double)score + s);
                            if (a != b) {
                                if (a == '-') { if (run != 1) { run = 1; indel++; npos++; } }
                                else if (b == '-') { if (run != 2) { run = 2; indel++; npos++; } }
                                else { run = 0; npos++; }
                            } else {
                                run = 0;
                            }
                        }
                        pf_align_hit h;
                        h.text_off = 0;
                        h.gap_off = 0;
                        h.len = pend_len;
                        h.n_gaps = pend_ng;
                        h.score = score;
                        h.n_pos = npos;
                        h.n_indel = indel;
                        sc.hits[0] = h;
                        char *ta = sc.text, *tb = ta + pend_len;
                        for (uint32_t t = 0; t < pend_len; ++t) {
                            const char a = ra[pend_len - 1 - t];
                            ta[t] = a == '+' ? '-' : a;
                            tb[t] = rb[pend_len - 1 - t];
                        }
                        for (uint32_t t = 0; t < pend_ng; ++t) sc.gaps[t] = gp[t];
                        last_score = score;
                        last_npos = npos;
                        last_indel = indel;
                        pending = 0;
                        if (prof) st_leaf_ticks += wall_clock64() - tl0;
                    }
                    const char pa = len >= 2 ? ra[len - 2] : '\0', pb = len >= 2 ? rb[len - 2] : '\0';   // the new top
                    const uint8_t mvv = mv[len - 1];
                    if (ta == '+') {
                        if (len >= 2) { if (pa != '+') --open_a; }
                        else --open_a;
                    }
                    if (tb == '-') {
                        if (len >= 2) { if (pb != '-') --open_b; }
                        else --open_b;
                    }
                    if (ta == '+') ng--;
                    ta = pa;
                    tb = pb;
                    const uint8_t went = mvv & 7;
                    if (went == LEFT) j += 1;
                    else if (went == UP) i += 1;
                    else { i += 1; j += 1; }
                    len--;
                    work = mvv >> 4;   // what was left to try at the cell the walk is back at
                    if (work) { --alts; break; }   // (the cell was counted when it was left): the general step
                    if (prof) ++st_steps;
                }
                top_a = ta;
                top_b = tb;
                if (out) break;
            }
        }
        __builtin_amdgcn_s_setprio(0);
        if (prof) { atomicAdd(&prof[8], st_steps); atomicAdd(&prof[9], st_leaves); atomicAdd(&prof[10], st_takes); atomicAdd(&prof[11], st_leaf_ticks); atomicAdd(&prof[12], 1ull); }
    }
    aln_sync();
    if (!unique) {
        pending = __shfl(pending, 0, WAVE);
        if (pending && __shfl((int)overflow, 0, WAVE) == 0) {
            n_hits = text_used = gaps_used = 0;   // (lane 0's 1, 2 * len, ng: set again by keep_only_path, on every lane)
            keep_only_path(__shfl(pend_len, 0, WAVE), __shfl(pend_ng, 0, WAVE));
            aln_sync();
        }
    }
    if (prof && lane == 0) {
        atomicAdd(&prof[0], pc1 - pc0);
        atomicAdd(&prof[1], wall_clock64() - pc1);
        if (unique) atomicAdd(&prof[13], 1ull);
    }
    n_hits_out = __shfl(n_hits, 0, WAVE);
    text_out = __shfl(text_used, 0, WAVE);
    gaps_out = __shfl(gaps_used, 0, WAVE);
    return __shfl((int)overflow, 0, WAVE) == 0;
}

}  // namespace pf
