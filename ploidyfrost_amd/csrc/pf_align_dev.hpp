// Device side of K-ALN shared by k_align (pf_align.hip) and k_bubble (pf_bubble.hip): the
// Needleman-Wunsch fill + all-co-optimal traceback of one pairwise job on one wavefront
// (reference src/SeqAlign.cpp:480-549, 306-478).  See pf_align.hip for the description.
#pragma once
#include <hip/hip_runtime.h>
#include <limits.h>
#include <stdint.h>

#include "pf_device_common.hpp"
#include "ploidyfrost_hip.h"

namespace pf {

enum : uint8_t { UP = 1, DIAG = 2, LEFT = 4 };

__host__ __device__ inline uint32_t al4(uint32_t x) { return (x + 3u) & ~3u; }

// bytes of working storage one job needs (must match the carve-up in align_job)
__host__ __device__ inline uint64_t job_bytes(uint32_t m, uint32_t n) {
    const uint64_t cells = (uint64_t)(m + 1) * (n + 1);
    return ((cells + 3) & ~3ull) + 12ull * (m + 1) + al4(m) + al4(n) + 3ull * al4(m + n) + al4(2 * (m + n)) + 16;
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

// One job on one wavefront.  `base` = working storage (LDS or global), job_bytes(m, n) bytes.
// Returns false when the staging area overflowed.
__device__ inline bool align_job(uint8_t *base, const char *__restrict__ ga, const char *__restrict__ gb, uint32_t m, uint32_t n,
                          double M, double D, double G, const AlnScratch &sc, uint32_t &n_hits_out, uint32_t &text_out,
                          uint32_t &gaps_out) {
    const int lane = lane_id();
    const uint32_t W = n + 1;
    const uint32_t cells = (m + 1) * W;
    uint8_t *dir = base;
    int *s0 = reinterpret_cast<int *>(base + ((cells + 3) & ~3u));
    int *s1 = s0 + (m + 1);
    int *s2 = s1 + (m + 1);
    char *A = reinterpret_cast<char *>(s2 + (m + 1));
    char *B = A + al4(m);
    char *ra = B + al4(n);
    char *rb = ra + al4(m + n);
    uint8_t *mv = reinterpret_cast<uint8_t *>(rb + al4(m + n));
    uint16_t *gp = reinterpret_cast<uint16_t *>(mv + al4(m + n));

    for (uint32_t i = lane; i < m; i += WAVE) A[i] = ga[i];
    for (uint32_t j = lane; j < n; j += WAVE) B[j] = gb[j];
    // borders (src/SeqAlign.cpp:486-496)
    for (uint32_t i = lane; i <= m; i += WAVE) dir[i * W] = i ? (uint8_t)(UP | (UP << 4)) : 0;
    for (uint32_t j = 1 + lane; j <= n; j += WAVE) dir[j] = (uint8_t)(LEFT | (LEFT << 4));
    // rolling diagonals, indexed by row: p2 = diagonal d-2, p1 = d-1, cur = d
    int *p2 = s0, *p1 = s1, *cur = s2;
    if (lane == 0) {
        p2[0] = 0;                                  // (0,0)
        p1[0] = n >= 1 ? (int)(long)(G * 1) : 0;    // (0,1)
        if (m >= 1) p1[1] = (int)(long)(G * 1);     // (1,0)
    }
    aln_sync();
    for (uint32_t d = 2; d <= m + n; ++d) {
        const uint32_t lo = d > n ? d - n : 1;
        const uint32_t hi = d - 1 < m ? d - 1 : m;
        for (uint32_t i = lo + lane; i <= hi; i += WAVE) {
            const uint32_t j = d - i;
            const uint8_t f_up = dir[(i - 1) * W + j], f_dg = dir[(i - 1) * W + j - 1], f_lf = dir[i * W + j - 1];
            int up = (int)((double)(long)p1[i - 1] + G);
            if (f_up & UP) up += 1;
            const char a = A[i - 1], b = B[j - 1];
            const double sub = a == b ? M : ((a == '-' || b == '-') ? G : D);
            int dg = (int)((double)(long)p2[i - 1] + sub);
            if (f_dg & DIAG) dg += 1;
            int lf = (int)((double)(long)p1[i] + G);
            if (f_lf & LEFT) lf += 1;
            int best = up > dg ? up : dg;
            best = best > lf ? best : lf;
            if (best == lf && i != m && A[i] == '-') {  // :528-532
                lf = INT_MIN;
                best = up > dg ? up : dg;
            }
            uint8_t f = 0;
            if (up == best) f |= UP;
            if (dg == best) f |= DIAG;
            if (lf == best) f |= LEFT;
            cur[i] = best;
            dir[i * W + j] = (uint8_t)(f | (f << 4));
        }
        if (lane == 0) {
            if (d <= n) cur[0] = (int)(long)(G * (double)d);  // (0,d)
            if (d <= m) cur[d] = (int)(long)(G * (double)d);  // (d,0)
        }
        aln_sync();
        int *t = p2;
        p2 = p1;
        p1 = cur;
        cur = t;
    }

    // ---- traceback (lane 0) --------------------------------------------------------------
    uint32_t n_hits = 0, text_used = 0, gaps_used = 0;
    bool overflow = false;
    if (lane == 0) {
        uint64_t open_a = 0, open_b = 0, lim_a = 5, lim_b = 5;  // size_t in the reference
        uint32_t len = 0, ng = 0;
        uint32_t i = m, j = n;
        bool have = false;
        long long last_score = 0;
        uint32_t last_npos = 0, last_indel = 0;
        for (;;) {
            const uint32_t c = i * W + j;
            if (i == 0 && j == 0 && open_a <= lim_a && open_b <= lim_b) {
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
                }
            }
            const uint8_t dc = dir[c];
            const uint8_t work = dc >> 4;
            const char fa = len ? ra[len - 1] : '\0';
            const char fb = len ? rb[len - 1] : '\0';
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
                if (!go) {
                    dir[c] = dc & (uint8_t) ~(LEFT | (LEFT << 4));
                    continue;
                }
                dir[c] = dc & (uint8_t) ~(LEFT << 4);
                ra[len] = '+';
                rb[len] = B[j - 1];
                mv[len] = LEFT;
                gp[ng++] = (uint16_t)i;
                len++;
                j -= 1;
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
                if (!go) {
                    dir[c] = dc & (uint8_t) ~(UP | (UP << 4));
                    continue;
                }
                dir[c] = dc & (uint8_t) ~(UP << 4);
                ra[len] = A[i - 1];
                rb[len] = '-';
                mv[len] = UP;
                len++;
                i -= 1;
            } else if (work & DIAG) {
                dir[c] = dc & (uint8_t) ~(DIAG << 4);
                ra[len] = A[i - 1];
                rb[len] = B[j - 1];
                mv[len] = DIAG;
                len++;
                i -= 1;
                j -= 1;
            } else {
                if (len == 0) break;
                dir[c] = (uint8_t)((dc & 0x0F) | ((dc & 0x0F) << 4));  // matrix_temp[p] = matrix[p]
                if (fa == '+') {
                    if (len >= 2) { if (ra[len - 2] != '+') --open_a; }
                    else --open_a;
                }
                if (fb == '-') {
                    if (len >= 2) { if (rb[len - 2] != '-') --open_b; }
                    else --open_b;
                }
                if (fa == '+') ng--;
                const uint8_t mvv = mv[len - 1];
                if (mvv == LEFT) j += 1;
                else if (mvv == UP) i += 1;
                else { i += 1; j += 1; }
                len--;
            }
        }
    }
    aln_sync();
    n_hits_out = __shfl(n_hits, 0, WAVE);
    text_out = __shfl(text_used, 0, WAVE);
    gaps_out = __shfl(gaps_used, 0, WAVE);
    return __shfl((int)overflow, 0, WAVE) == 0;
}

}  // namespace pf
