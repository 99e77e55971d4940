// K-PAIR, device side: SeqAlign::SequenceAlignment (reference src/SeqAlign.cpp:550-640) for a bubble with exactly TWO paths of
// at most 64 bases, one THREAD per bubble.
//
// Two short paths are what nearly every bubble that needs dynamic programming looks like (a bi-allelic indel, or two SNPs
// closer than k).  A wavefront per bubble (K-BUBBLE) spends most of its instructions on the parts of the algorithm that are
// sequential per bubble -- the traceback walk, the column state machine, the ladder -- with 63 lanes watching; here every lane
// runs the whole algorithm for its own bubble, so a wavefront retires 64 bubbles for about the instruction count of a few.
// With two paths SequenceAlignment is its first round only: needlemanWunch + traceback of the pair (:480-549, 306-478) and
// compareStrPair over the kept alignments (:8-236).
//
// Memory: the score row of the fill (ONE row: up / diagonal neighbours are the row's old values) and the two paths as 2-bit
// codes live in registers; the direction matrix, the traceback strings and the kept alignments live in per-wavefront global
// scratch, entry-major and lane-minor, so that lanes working in step touch one cache line per access.
#pragma once
#include <hip/hip_runtime.h>
#include <limits.h>
#include <stdint.h>

#include "pf_align_dev.hpp"

namespace pf {

constexpr uint32_t PAIR_MAX = 64;            // longest path
constexpr uint32_t PAIR_W = 68;                // cells per matrix row in memory (65 used): a row is 17 dwords
constexpr uint32_t PAIR_CELLS = (PAIR_MAX + 1) * PAIR_W;
constexpr uint32_t PAIR_LEN = 2 * PAIR_MAX;  // longest alignment
constexpr uint32_t PAIR_HITS = 8;            // kept alignments; more: the bubble goes to K-BUBBLE

// bytes of global scratch one wavefront needs
__host__ __device__ inline uint64_t pair_scratch_bytes() {
    return 64ull * (PAIR_CELLS + 3ull * PAIR_LEN + (uint64_t)PAIR_HITS * (2 * PAIR_LEN + 32));
}

struct PairHit {
    long long score;
    uint32_t len, n_pos, n_indel, pad_;
};

struct PairMem {   // this lane's view: element e of an array at base[e * 64]
    uint64_t Aw[2], Bw[2];   // the two paths, 2 bits per base, first base most significant
    uint8_t *dir;         // global, PAIR_CELLS, interleaved by DWORD (four consecutive cells of a lane are one aligned word): pair_dir()
    char *ra, *rb;        // global, PAIR_LEN each
    uint8_t *mv;          // global, PAIR_LEN
    char *htext;          // global, PAIR_HITS * 2 * PAIR_LEN
    PairHit *hits;        // global, PAIR_HITS (struct stride 64)
};

#define PF_AT(p, e) (p)[(size_t)(e) * 64]
// cell e of this lane's direction matrix (mem.dir already points at the lane's first dword)
__device__ inline uint8_t &pair_dir(uint8_t *dir, size_t e) { return dir[(e >> 2) * 256 + (e & 3)]; }
__device__ inline char pair_base(const uint64_t (&w)[2], uint32_t idx) { return "ACGT"[(uint32_t)(w[idx >> 5] >> (62 - 2 * (idx & 31))) & 3u]; }

// what compareStrPair's column pass (src/SeqAlign.cpp:56-157) yields for two rows, computed in one sweep
struct PairMetrics {
    int snp, indel;                   // the reference's 8-bit counters
    uint32_t n_sites, n_indel_len;
    uint64_t d_snp, d_indel, d_all;   // compute_dis of snp_pos, indel_pos and their merge (:10-38)
    int first_site, last_site;
    bool any_site;
};

struct PairSpread {   // compute_dis over an ascending list fed one element at a time
    uint32_t n = 0, first = 0, prev = 0;
    uint64_t d = 0;
    __device__ inline void add(uint32_t v) {
        if (n == 0) { first = v; d = v; }
        else {
            const int gap = (int)(v - prev - 1);
            d = (uint64_t)(gap < (int)d ? gap : (int)d);
        }
        prev = v;
        ++n;
    }
    __device__ inline uint64_t value(uint64_t L) const {
        if (n == 0) return 0;
        if (n == 1) {
            const int left = (int)first, right = (int)(L - first) - 1;
            return left > right ? (uint64_t)(left + 1) : (uint64_t)right;
        }
        const uint64_t tail = L - prev - 1;
        return d < tail ? d : tail;
    }
};

// One sweep over the columns of the alignment (x, y: this lane's interleaved rows of length L).  With EMIT the variant columns
// and the indel lengths are written out (sites[] : column and "opens an indel"; ilen[]).
template <bool EMIT>
__device__ inline PairMetrics pair_classify(const char *x, const char *y, uint32_t L, uint64_t Lref, pf_bubble_site *sites, uint32_t *ilen) {
    PairMetrics m;
    uint8_t snp = 0, indel = 0;
    uint32_t ns = 0, nl = 0, last_indel_pos = 0;
    bool open = false;
    PairSpread s_snp, s_ind, s_all;
    char pa = 0, pb = 0;
    for (uint32_t j = 0; j < L; ++j) {
        const char a = PF_AT(x, j), b = PF_AT(y, j);
        const int t = a == b ? 0 : ((a == '-' || b == '-') ? 2 : 1);
        if (t != 2) {
            if (open) { if (EMIT) ilen[nl] = j - last_indel_pos; nl++; open = false; }
            if (t == 1) {
                snp++;
                s_snp.add(j);
                s_all.add(j);
                if (EMIT) { pf_bubble_site sr; sr.col = j; sr.is_indel = 0; sr.maxnum = 2; sr.pad_ = 0; sites[ns] = sr; }
                ns++;
            }
        } else {
            const bool same_status = j > 0 && ((a == '-') == (pa == '-')) && ((b == '-') == (pb == '-'));
            const bool same_run = open && same_status;
            if (open && !same_run) { if (EMIT) ilen[nl] = j - last_indel_pos; nl++; }
            if (!same_run) {
                ++indel;
                last_indel_pos = j;
                s_ind.add(j);
                s_all.add(j);
                open = true;
                if (EMIT) { pf_bubble_site sr; sr.col = j; sr.is_indel = 1; sr.maxnum = 2; sr.pad_ = 0; sites[ns] = sr; }
                ns++;
            }
        }
        pa = a;
        pb = b;
    }
    m.snp = snp;
    m.indel = indel;
    m.n_sites = ns;
    m.n_indel_len = nl;
    m.d_snp = s_snp.value(Lref);
    m.d_indel = s_ind.value(Lref);
    m.d_all = s_all.value(Lref);
    m.any_site = s_all.n > 0;
    m.first_site = s_all.n ? (int)s_all.first : 0;
    m.last_site = s_all.n ? (int)s_all.prev : 0;
    return m;
}

// needlemanWunch (src/SeqAlign.cpp:480-549) of A (m) x B (n): direction flags of every cell to mem.dir.
// The score row never leaves the register file: the loop over the columns is unrolled over all PAIR_MAX of them (lanes whose B is
// shorter are predicated off), so every row[j] is a named register and a cell is a dozen dependent-free vector instructions --
// no LDS, no scratch.  An entry holds score * 8 + direction flags.  A and B are the paths as 2-bit codes, first base most
// significant: A is indexed by the (runtime) row, B by the (compile-time) column.
__device__ inline void pair_fill(const PairMem &mem, const uint64_t (&Aw)[2], const uint64_t (&Bw)[2], uint32_t m, uint32_t n, double M, double D,
                                 double G, int integral) {
    (void)n;   // every row is computed over all PAIR_MAX columns, branch-free: the cells behind a lane's own n are never read, and
               // without a per-cell branch the scheduler interleaves the independent parts of neighbouring cells (only `left` chains)
    const int Mi = (int)M, Di = (int)D, Gi = (int)G;
    int row[PAIR_MAX + 1];
    row[0] = 0;
    uint32_t *d32 = reinterpret_cast<uint32_t *>(mem.dir);   // this lane's dwords: word w of row i at d32[(i * 17 + w) * 64]
    constexpr uint32_t LL = LEFT | (LEFT << 4);
#pragma unroll
    for (uint32_t j = 1; j <= PAIR_MAX; ++j) row[j] = (integral ? Gi * (int)j : (int)(long)(G * (double)j)) * 8 + LEFT;
    d32[0] = LL << 8 | LL << 16 | LL << 24;                   // (0, 0) carries no flag
#pragma unroll
    for (uint32_t w = 1; w < PAIR_W / 4; ++w) d32[(size_t)w * 64] = LL | LL << 8 | LL << 16 | LL << 24;
    if (integral) {
        for (uint32_t i = 1; i <= m; ++i) {
            const uint32_t a = (uint32_t)(Aw[(i - 1) >> 5] >> (62 - 2 * ((i - 1) & 31))) & 3u;
            int dg_v = row[0];
            int last_s = Gi * (int)i;   // (i, 0): border, flag Up
            int last_f = UP;
            row[0] = last_s * 8 + UP;
            uint32_t *drow = d32 + (size_t)i * (PAIR_W / 4) * 64;
            uint32_t pack = UP | (UP << 4);   // column 0: border
#pragma unroll
            for (uint32_t j = 1; j <= PAIR_MAX; ++j) {
                const uint32_t b = (uint32_t)(Bw[(j - 1) >> 5] >> (62 - 2 * ((j - 1) & 31))) & 3u;
                const int up_v = row[j];
                const int up = (up_v >> 3) + Gi + (up_v & 1);                  // UP = 1: +1 for continuing a direction (:512-526)
                const int dg = (dg_v >> 3) + (a == b ? Mi : Di) + ((dg_v >> 1) & 1);   // DIAG = 2   (paths carry no '-')
                const int lf = last_s + Gi + ((last_f >> 2) & 1);              // LEFT = 4
                int best = up > dg ? up : dg;
                best = best > lf ? best : lf;
                const int f = (up == best ? UP : 0) | (dg == best ? DIAG : 0) | (lf == best ? LEFT : 0);
                dg_v = up_v;
                last_s = best;
                last_f = f;
                row[j] = best * 8 + f;
                pack |= (uint32_t)(f | (f << 4)) << (8 * (j & 3));
                if ((j & 3) == 3 || j == PAIR_MAX) { drow[(size_t)(j >> 2) * 64] = pack; pack = 0; }
            }
        }
    } else {
        for (uint32_t i = 1; i <= m; ++i) {
            const uint32_t a = (uint32_t)(Aw[(i - 1) >> 5] >> (62 - 2 * ((i - 1) & 31))) & 3u;
            int dg_v = row[0];
            int last_s = (int)(long)(G * (double)i);
            int last_f = UP;
            row[0] = last_s * 8 + UP;
            uint32_t *drow = d32 + (size_t)i * (PAIR_W / 4) * 64;
            uint32_t pack = UP | (UP << 4);   // column 0: border
#pragma unroll
            for (uint32_t j = 1; j <= PAIR_MAX; ++j) {
                const uint32_t b = (uint32_t)(Bw[(j - 1) >> 5] >> (62 - 2 * ((j - 1) & 31))) & 3u;
                const int up_v = row[j];
                // the reference's `int = long + double`: truncation at every cell
                const int up = (int)((double)(up_v >> 3) + G) + (up_v & 1);
                const int dg = (int)((double)(dg_v >> 3) + (a == b ? M : D)) + ((dg_v >> 1) & 1);
                const int lf = (int)((double)last_s + G) + ((last_f >> 2) & 1);
                int best = up > dg ? up : dg;
                best = best > lf ? best : lf;
                const int f = (up == best ? UP : 0) | (dg == best ? DIAG : 0) | (lf == best ? LEFT : 0);
                dg_v = up_v;
                last_s = best;
                last_f = f;
                row[j] = best * 8 + f;
                pack |= (uint32_t)(f | (f << 4)) << (8 * (j & 3));
                if ((j & 3) == 3 || j == PAIR_MAX) { drow[(size_t)(j >> 2) * 64] = pack; pack = 0; }
            }
        }
    }
}

// variantAnalyze (src/SeqAlign.cpp:237-305) over the strings as the traceback holds them (back to front, '+' = gap in A)
__device__ inline void pair_score(const PairMem &mem, uint32_t len, double M, double D, double G, long long &score, uint32_t &npos, uint32_t &indel) {
    score = 0;
    npos = indel = 0;
    uint8_t run = 0;
    for (uint32_t t = len; t-- > 0;) {
        const char ax = PF_AT(mem.ra, t), a = ax == '+' ? '-' : ax, b = PF_AT(mem.rb, t);
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
}

__device__ inline void pair_keep(const PairMem &mem, uint32_t h, uint32_t len, long long score, uint32_t npos, uint32_t indel) {
    PairHit ph;
    ph.score = score;
    ph.len = len;
    ph.n_pos = npos;
    ph.n_indel = indel;
    ph.pad_ = 0;
    PF_AT(mem.hits, h) = ph;
    char *ta = mem.htext + (size_t)h * 2 * PAIR_LEN * 64, *tb = ta + (size_t)PAIR_LEN * 64;
    for (uint32_t t = 0; t < len; ++t) {
        const char a = PF_AT(mem.ra, len - 1 - t);
        PF_AT(ta, t) = a == '+' ? '-' : a;
        PF_AT(tb, t) = PF_AT(mem.rb, len - 1 - t);
    }
}

// traceback (src/SeqAlign.cpp:306-478): all co-optimal alignments within the shrinking 5-gap-open budgets, kept in traceback
// order.  Returns the number kept, or 0xFFFFFFFF when they exceed PAIR_HITS -- or, with unique_only, when the matrix has more
// than one optimal path: the depth-first search over alternatives is a chain of dependent read-modify-writes of the direction
// matrix, which a thread pays with an L2 round trip each and a wavefront with an LDS access (K-BUBBLE keeps the matrix in LDS).
__device__ inline uint32_t pair_traceback(const PairMem &mem, uint32_t m, uint32_t n, double M, double D, double G, bool unique_only) {
    constexpr uint32_t W = PAIR_W;
    // most matrices have exactly one optimal path -- every cell on it carries a single flag -- and then the reference's DFS
    // descends it once, emits that alignment and unwinds without an alternative: walking it is the whole traceback
    {
        bool unique = true;
        uint32_t i = m, j = n, oa = 0, ob = 0, len = 0;
        char fa = '\0', fb = '\0';
        while (i > 0 || j > 0) {
            const uint8_t f = pair_dir(mem.dir, (size_t)i * W + j) & 7;
            char ca, cb;
            if (f == DIAG) { ca = pair_base(mem.Aw, i - 1); cb = pair_base(mem.Bw, j - 1); i--; j--; }
            else if (f == LEFT) {
                if (oa >= 5) { unique = false; break; }
                if (len == 0 || fa != '+') ++oa;
                ca = '+';
                cb = pair_base(mem.Bw, j - 1);
                j--;
            } else if (f == UP) {
                if (ob >= 5) { unique = false; break; }
                if (len == 0 || fb == '-') ++ob;
                ca = pair_base(mem.Aw, i - 1);
                cb = '-';
                i--;
            } else { unique = false; break; }
            PF_AT(mem.ra, len) = ca;
            PF_AT(mem.rb, len) = cb;
            fa = ca;
            fb = cb;
            len++;
        }
        if (unique) {
            long long score;
            uint32_t npos, indel;
            pair_score(mem, len, M, D, G, score, npos, indel);
            pair_keep(mem, 0, len, score, npos, indel);
            return 1;
        }
    }
    if (unique_only) return 0xFFFFFFFFu;
    uint64_t open_a = 0, open_b = 0, lim_a = 5, lim_b = 5;  // size_t in the reference
    uint32_t len = 0, n_hits = 0;
    uint32_t i = m, j = n;
    bool have = false;
    long long last_score = 0;
    uint32_t last_npos = 0, last_indel = 0;
    for (;;) {
        const size_t c = (size_t)i * W + j;
        if (i == 0 && j == 0 && open_a <= lim_a && open_b <= lim_b) {
            long long score;
            uint32_t npos, indel;
            pair_score(mem, len, M, D, G, score, npos, indel);
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
                if (d32 < 0) n_hits = 0;
                else if (d32 > 0) take = false;
            }
            if (take) {
                if (n_hits >= PAIR_HITS) return 0xFFFFFFFFu;
                pair_keep(mem, n_hits++, len, score, npos, indel);
                lim_a = open_a;
                lim_b = open_b;
                have = true;
                last_score = score;
                last_npos = npos;
                last_indel = indel;
            }
        }
        const uint8_t dc = pair_dir(mem.dir, c);
        const uint8_t work = dc >> 4;
        const char fa = len ? PF_AT(mem.ra, len - 1) : '\0';
        const char fb = len ? PF_AT(mem.rb, len - 1) : '\0';
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
            if (!go) { pair_dir(mem.dir, c) = dc & (uint8_t) ~(LEFT | (LEFT << 4)); continue; }
            pair_dir(mem.dir, c) = dc & (uint8_t) ~(LEFT << 4);
            PF_AT(mem.ra, len) = '+';
            PF_AT(mem.rb, len) = pair_base(mem.Bw, j - 1);
            PF_AT(mem.mv, len) = LEFT;
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
            if (!go) { pair_dir(mem.dir, c) = dc & (uint8_t) ~(UP | (UP << 4)); continue; }
            pair_dir(mem.dir, c) = dc & (uint8_t) ~(UP << 4);
            PF_AT(mem.ra, len) = pair_base(mem.Aw, i - 1);
            PF_AT(mem.rb, len) = '-';
            PF_AT(mem.mv, len) = UP;
            len++;
            i -= 1;
        } else if (work & DIAG) {
            pair_dir(mem.dir, c) = dc & (uint8_t) ~(DIAG << 4);
            PF_AT(mem.ra, len) = pair_base(mem.Aw, i - 1);
            PF_AT(mem.rb, len) = pair_base(mem.Bw, j - 1);
            PF_AT(mem.mv, len) = DIAG;
            len++;
            i -= 1;
            j -= 1;
        } else {
            if (len == 0) break;
            pair_dir(mem.dir, c) = (uint8_t)((dc & 0x0F) | ((dc & 0x0F) << 4));  // matrix_temp[p] = matrix[p]
            if (fa == '+') {
                if (len >= 2) { if (PF_AT(mem.ra, len - 2) != '+') --open_a; }
                else --open_a;
            }
            if (fb == '-') {
                if (len >= 2) { if (PF_AT(mem.rb, len - 2) != '-') --open_b; }
                else --open_b;
            }
            const uint8_t mvv = PF_AT(mem.mv, len - 1);
            if (mvv == LEFT) j += 1;
            else if (mvv == UP) i += 1;
            else { i += 1; j += 1; }
            len--;
        }
    }
    return n_hits;
}

// compareStrPair's selection ladder (src/SeqAlign.cpp:158-233) over the kept alignments; -1: none
__device__ inline int pair_choose(const PairMem &mem, uint32_t n_hits) {
    if (n_hits == 0) return -1;
    if (n_hits == 1) return 0;   // a single candidate beats the initial INT_MAX / 2 counts whatever its metrics
    const uint64_t Lref = PF_AT(mem.hits, n_hits - 1).len;
    int best = -1, best_snp = INT_MAX / 2, best_indel = INT_MAX / 2;
    int d_snp = INT_MAX, d_indel = INT_MAX, d_all = INT_MAX, left = -1, right = -1;
    for (uint32_t c = 0; c < n_hits; ++c) {
        const uint32_t L = PF_AT(mem.hits, c).len;
        const char *x = mem.htext + (size_t)c * 2 * PAIR_LEN * 64, *y = x + (size_t)PAIR_LEN * 64;
        const PairMetrics m = pair_classify<false>(x, y, L, Lref, nullptr, nullptr);
        int verdict = 0;  // 1 take, 2 take on the strcmp tie-break
        uint64_t c_indel = 0, c_snp = 0, c_all = 0;
        const int total = m.snp + m.indel, btotal = best_snp + best_indel;
        if (total < btotal) verdict = 1;
        else if (total == btotal) {
            if (m.indel < best_indel) verdict = 1;
            else if (m.indel == best_indel) {
                c_indel = m.d_indel;
                if (c_indel > (uint64_t)d_indel) verdict = 1;
                else if (c_indel == (uint64_t)d_indel) {
                    c_snp = m.d_snp;
                    if (c_snp > (uint64_t)d_snp) verdict = 1;
                    else if (c_snp == (uint64_t)d_snp) {
                        c_all = m.d_all;
                        if (c_all > (uint64_t)d_all) verdict = 1;
                        else if (c_all == (uint64_t)d_all) {
                            if (m.first_site > left || m.last_site > right) verdict = 1;
                            else if (m.first_site == left && m.last_site == right && best >= 0) {
                                const uint32_t LB = PF_AT(mem.hits, best).len;
                                const char *bx = mem.htext + (size_t)best * 2 * PAIR_LEN * 64;
                                for (uint32_t r = 0; r < 2 && verdict == 0; ++r) {
                                    const char *cr = r ? y : x, *br = bx + (size_t)r * PAIR_LEN * 64;
                                    const uint32_t lm = L < LB ? L : LB;
                                    uint32_t p = 0;
                                    while (p < lm && PF_AT(cr, p) == PF_AT(br, p)) ++p;
                                    const bool greater = p < lm ? (unsigned char)PF_AT(cr, p) > (unsigned char)PF_AT(br, p) : L > LB;
                                    if (greater) verdict = 2;
                                }
                                if (verdict == 2) { left = m.first_site; right = m.last_site; }
                            }
                        }
                    }
                }
            }
        }
        if (verdict == 0) continue;
        if (verdict == 1) {
            const int f = m.any_site ? m.first_site : -1, l = m.any_site ? m.last_site : -1;
            left = left > f ? left : f;
            right = right > l ? right : l;
            c_all = m.d_all;
            c_snp = m.d_snp;
            c_indel = m.d_indel;
        }
        d_all = (int)c_all;
        d_snp = (int)c_snp;
        d_indel = (int)c_indel;
        best_snp = m.snp;
        best_indel = m.indel;
        best = (int)c;
    }
    return best;
}

}  // namespace pf
