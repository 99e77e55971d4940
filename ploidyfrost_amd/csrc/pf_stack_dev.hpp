// K-STACK, device side: SeqAlign::SequenceAlignment (reference src/SeqAlign.cpp:550-640) for a bubble whose paths are ALL OF ONE
// LENGTH and whose alignment turns out to be the paths themselves, stacked -- one THREAD per bubble, no dynamic programming.
//
// Which bubbles: two or three substitutions closer than k on a polyploid genome give three to eight paths of one length (the
// branching bubbles K-BUBBLE spent most of its time on at BASELINE.json's configs[2]: three paths, 78 % of its ticks), a site with
// three or four alleles gives a strict bubble of three or four equally long inner unitigs.  For such paths every round of the
// progressive alignment is needlemanWunch(path 0, path p) of two gap-free strings of one length, and when that matrix has a single
// optimal path -- the main diagonal -- the round keeps one alignment, "both unchanged", re-opens no gap in the older rows, and
// compareStrPair has one candidate: the result is the stacked paths, every column with more than one base a SNP column.
//
// The certificate.  That the diagonal is the single optimum is PROVED per pair, not assumed: with b = 1 the reference's bonus for
// continuing a direction (:512-526), the exact diagonal values Dg(r) = Dg(r-1) + s(x_r, y_r) + [r >= 2] b are compared with UPPER
// BOUNDS of the two neighbours a diagonal cell could also be reached from.  The bounds come from the same recurrence run over a
// band of STACK_W cells on either side of the diagonal (with the sequences' real matches and mismatches -- a shifted alignment
// of unrelated sequence matches a quarter of its bases, a repeat all of them, and the band tells the two apart), every move
// credited with the bonus, and cells beyond the band replaced by the bound S(i, j) <= min(i, j) (M + b) + |i - j| (G + b), which
// holds for any cell when M >= D and M + b >= 2 (G + b) (K-BUBBLE's single-mismatch shortcut rests on the same bound with a band
// of zero, which carries one mismatch; the band carries about 4 (W + 1) |G + b| / (M - D) of them).  If at every row the
// diagonal move beats both bounds STRICTLY, every diagonal cell carries the single flag DIAG, the traceback finds exactly one
// alignment, and nothing else needs computing.  A pair that is not certified sends its bubble to K-BUBBLE: nothing is ever
// decided on a bound that does not hold.
#pragma once
#include <hip/hip_runtime.h>
#include <limits.h>
#include <stdint.h>

#include "pf_call_dev.hpp"
#include "pf_pair_dev.hpp"

namespace pf {

constexpr uint32_t STACK_MAX = 128;    // longest path
constexpr uint32_t STACK_PATHS = 16;   // most paths
constexpr int STACK_W = 3;             // cells of the band on either side of the diagonal

// scores under which the bound for the cells beyond the band holds (and the arithmetic is the reference's integer arithmetic)
inline bool stack_scores(double M, double D, double G) {
    if (!(M < 1e5 && M > -1e5 && D < 1e5 && D > -1e5 && G < 1e5 && G > -1e5)) return false;
    const bool integral = M == (double)(long long)M && D == (double)(long long)D && G == (double)(long long)G;
    return integral && M >= D && M + 1 >= 2 * (G + 1);
}

struct StackPlanes {   // a path as two bit planes, base c at bit (c & 31) of word (c >> 5); zero beyond its length
    uint32_t lo[4], hi[4];
};

// bit t of the result = plane bit (32 wb + t + d), zero outside 0 .. 127
template <int WB, int D>
__device__ inline uint32_t stack_shifted(const uint32_t (&y)[4]) {
    const uint32_t cur = y[WB];
    if constexpr (D == 0) return cur;
    else if constexpr (D > 0) {
        const uint32_t nxt = WB + 1 < 4 ? y[WB + 1 < 4 ? WB + 1 : 3] : 0u;
        return (cur >> D) | (nxt << (32 - D));
    } else {
        const uint32_t prv = WB >= 1 ? y[WB >= 1 ? WB - 1 : 0] : 0u;
        return (cur << (-D)) | (prv >> (32 + D));
    }
}

template <int WB, int D>
__device__ inline uint32_t stack_eq(const StackPlanes &x, const StackPlanes &y) {
    return ~((x.lo[WB] ^ stack_shifted<WB, D>(y.lo)) | (x.hi[WB] ^ stack_shifted<WB, D>(y.hi)));
}

// rows 32 WB + 1 .. of the certificate; V = the band of the row above (index d + W), updated in place.  false: not certified.
template <int WB>
__device__ inline bool stack_rows(const StackPlanes &x, const StackPlanes &y, uint32_t L, int M, int D, int G, int (&V)[2 * STACK_W + 1]) {
    constexpr int W = STACK_W;
    constexpr int NEG = INT_MIN / 4;
    if (32u * WB >= L) return true;
    const int Gb = G + 1, Mb = M + 1;
    const uint32_t e_m3 = stack_eq<WB, -3>(x, y), e_m2 = stack_eq<WB, -2>(x, y), e_m1 = stack_eq<WB, -1>(x, y), e_0 = stack_eq<WB, 0>(x, y),
                   e_p1 = stack_eq<WB, 1>(x, y), e_p2 = stack_eq<WB, 2>(x, y), e_p3 = stack_eq<WB, 3>(x, y);
    const uint32_t rows = L - 32u * WB < 32u ? L - 32u * WB : 32u;
    for (uint32_t t = 0; t < rows; ++t) {
        const int r = (int)(32u * WB + t) + 1;   // row of the matrix; x_r = base r - 1
        const uint32_t eq[2 * W + 1] = {(e_m3 >> t) & 1u, (e_m2 >> t) & 1u, (e_m1 >> t) & 1u, (e_0 >> t) & 1u, (e_p1 >> t) & 1u, (e_p2 >> t) & 1u,
                                        (e_p3 >> t) & 1u};
        int N[2 * W + 1];
#pragma unroll
        for (int d = -W; d <= W; ++d) {
            const int c = r + d;   // the cell (r, c)
            // neighbours: up (r-1, c) = band index d+1 of the row above; diagonal (r-1, c-1) = index d; left (r, c-1) = index d-1 of this row
            const int up = d + 1 <= W ? V[d + 1 + W] : (r - 1) * Mb + (W + 1) * Gb;
            const int dg = V[d + W];
            const int lf = d - 1 >= -W ? N[d - 1 + W] : (c - 1 > 0 ? (c - 1) * Mb + (W + 1) * Gb : (c - 1 == 0 ? G * r : NEG));
            const int s = eq[d + W] ? M : D;
            int v;
            if (d == 0) {
                const int dgc = dg + s + (r >= 2 ? 1 : 0);   // (r-1, r-1) carries DIAG (certified) unless it is (0, 0)
                if (!(dgc > up + Gb && dgc > lf + Gb)) return false;
                v = dgc;
            } else {
                // a move out of a diagonal cell continues nothing (that cell's only flag is DIAG); any other move may
                const int upv = up + G + (d == -1 ? 0 : 1);
                const int lfv = lf + G + (d == 1 ? 0 : 1);
                const int dgv = dg + s + 1;
                v = upv > dgv ? upv : dgv;
                v = v > lfv ? v : lfv;
            }
            if (c < 0 || c > (int)L) v = NEG;   // no such cell
            else if (c == 0) v = G * r;       // the border (:486-496), exact
            N[d + W] = v;
        }
#pragma unroll
        for (int q = 0; q < 2 * W + 1; ++q) V[q] = N[q];
    }
    return true;
}

// true: needlemanWunch(x, y) of two gap-free strings of length L has the main diagonal as its single optimal path
__device__ inline bool stack_certify(const StackPlanes &x, const StackPlanes &y, uint32_t L, int M, int D, int G) {
    constexpr int W = STACK_W;
    int V[2 * W + 1];
#pragma unroll
    for (int d = -W; d <= W; ++d) V[d + W] = d >= 0 ? G * d : INT_MIN / 4;   // row 0: S(0, d) = G d
    return stack_rows<0>(x, y, L, M, D, G, V) && stack_rows<1>(x, y, L, M, D, G, V) && stack_rows<2>(x, y, L, M, D, G, V) &&
           stack_rows<3>(x, y, L, M, D, G, V);
}

// ---- one gap run -------------------------------------------------------------------------------------------------------------
// The same certificate for two gap-free strings of DIFFERENT length, A (m bases) the longer: the candidate path runs down the
// diagonal for `a` bases, then UP for d = m - n cells (d bases of A against gaps in B's row), then down the diagonal to (m, n) --
// what needlemanWunch finds for a deletion of d bases after base a.  Exact values along the path (the bonus for continuing a
// direction is known there: every path cell carries the one flag of the move that entered it), upper bounds in a band of STACK_W
// cells on either side of it, the cells beyond bounded as above; certified when at every path cell the path's move beats the
// bounds of the two others strictly.  Then the matrix has this one optimal path, the traceback keeps this one alignment (one gap
// opened: inside the budget of five), and B's row is B with d gaps behind its first a bases.  The caller picks `a` (indel_place:
// the split with the most matching bases); a wrong pick cannot be certified, so nothing rests on how it was picked.

__device__ inline uint32_t plane_bit(const uint32_t (&p)[4], int idx) {   // 0 outside 0 .. 127
    if (idx < 0 || idx > 127) return 0;
    uint32_t w = p[0];
#pragma unroll
    for (int x = 1; x < 4; ++x) w = (idx >> 5) == x ? p[x] : w;
    return (w >> (idx & 31)) & 1u;
}

// seven plane bits from index i0 (any of them outside 0 .. 127: 0), bit t of the result = plane bit i0 + t
__device__ inline uint32_t plane_window7(const uint32_t (&p)[4], int i0) {
    const int w = i0 >> 5;   // (arithmetic shift: floor)
    auto word = [&](int x) -> uint32_t {
        uint32_t v = 0;
#pragma unroll
        for (int q = 0; q < 4; ++q) v = x == q ? p[q] : v;
        return v;
    };
    const uint64_t two = ((uint64_t)word(w + 1) << 32) | word(w);
    return (uint32_t)(two >> (i0 & 31)) & 0x7Fu;
}

// the split with the most matching bases: bases [0, a) of B against A's, bases [a, n) against A's d further on; 0xFFFFFFFF when
// two splits tie for it (a repeat: the matrix has several optimal paths)
__device__ inline uint32_t indel_place(const StackPlanes &A, const StackPlanes &B, uint32_t n, uint32_t d) {
    int best = -1, cur = 0;
    uint32_t best_a = 0;
    bool tie = false;
    for (uint32_t i = 0; i < n; ++i)   // a = 0: every base of B against A's base d further on
        cur += (plane_bit(A.lo, (int)(i + d)) == plane_bit(B.lo, (int)i) && plane_bit(A.hi, (int)(i + d)) == plane_bit(B.hi, (int)i)) ? 1 : 0;
    best = cur;
    for (uint32_t a = 1; a <= n; ++a) {   // base a - 1 moves from the shifted part to the unshifted one
        const uint32_t i = a - 1;
        const uint32_t bl = plane_bit(B.lo, (int)i), bh = plane_bit(B.hi, (int)i);
        cur += ((plane_bit(A.lo, (int)i) == bl && plane_bit(A.hi, (int)i) == bh) ? 1 : 0) -
               ((plane_bit(A.lo, (int)(i + d)) == bl && plane_bit(A.hi, (int)(i + d)) == bh) ? 1 : 0);
        if (cur > best) { best = cur; best_a = a; tie = false; }
        else if (cur == best) tie = true;
    }
    return tie ? 0xFFFFFFFFu : best_a;
}

__device__ inline bool indel_certify(const StackPlanes &A, const StackPlanes &B, uint32_t m, uint32_t n, uint32_t a, int M, int D, int G) {
    constexpr int W = STACK_W;
    constexpr int NEG = INT_MIN / 4;
    const int Gb = G + 1, Mb = M + 1;
    const uint32_t d = m - n;
    auto ugen = [&](int p, int q) -> int {   // bound of the cell (p, q) beyond the band; exact on the borders, NEG where there is no cell
        if (p < 0 || q < 0 || q > (int)n) return NEG;
        if (q == 0) return G * p;
        if (p == 0) return G * q;
        const int mn = p < q ? p : q, df = p < q ? q - p : p - q;
        return mn * Mb + df * Gb;
    };
    int V[2 * W + 1];
#pragma unroll
    for (int q = -W; q <= W; ++q) V[q + W] = q >= 0 && q <= (int)n ? G * q : NEG;   // row 0, centred on column 0
    int fprev = 0;   // the move that entered the path cell of the row above: 0 none ((0, 0) or a border cell), 1 DIAG, 2 UP
    for (uint32_t r = 1; r <= m; ++r) {
        const bool diag_row = r <= a || r > a + d;
        const int sg = diag_row ? 1 : 0;
        const int c = r <= a ? (int)r : (r <= a + d ? (int)a : (int)(r - d));   // the path's column in this row
        // does A's base r - 1 equal B's base at the columns c - W .. c + W (base index column - 1)?
        const uint32_t al = plane_bit(A.lo, (int)r - 1), ah = plane_bit(A.hi, (int)r - 1);
        const uint32_t wl = plane_window7(B.lo, c - W - 1), wh = plane_window7(B.hi, c - W - 1);
        const uint32_t eq = ~((wl ^ (0u - al)) | (wh ^ (0u - ah))) & 0x7Fu;
        int N[2 * W + 1];
#pragma unroll
        for (int q = -W; q <= W; ++q) {
            const int col = c + q;
            const int ui = q + sg, di = q + sg - 1, li = q - 1;   // band indices of the up / diagonal neighbour in the row above, of the left one in this row
            const int upv = (ui >= -W && ui <= W) ? V[(ui < -W ? -W : (ui > W ? W : ui)) + W] : ugen((int)r - 1, col);
            const int dgv = (di >= -W && di <= W) ? V[(di < -W ? -W : (di > W ? W : di)) + W] : ugen((int)r - 1, col - 1);
            const int lfv = li >= -W ? N[(li < -W ? -W : li) + W] : ugen((int)r, col - 1);
            const int s = ((eq >> (q + W)) & 1u) ? M : D;
            // a move out of a path cell continues that cell's one flag or nothing; any other move is credited with the bonus
            const int cu = upv + G + (ui == 0 ? (fprev == 2 ? 1 : 0) : 1);
            const int cd = dgv + s + (di == 0 ? (fprev == 1 ? 1 : 0) : 1);
            const int cl = lfv + G + (li == 0 ? 0 : 1);
            int v;
            if (q == 0) {
                if (diag_row) { if (!(cd > cu && cd > cl)) { if (col > 0) return false; } v = cd; }
                else { if (!(cu > cd && cu > cl)) { if (col > 0) return false; } v = cu; }
            } else {
                v = cu > cd ? cu : cd;
                v = v > cl ? v : cl;
            }
            if (col < 0 || col > (int)n) v = NEG;   // no such cell
            else if (col == 0) v = G * (int)r;    // the border (:486-496): exact, flag UP, nothing to certify
            N[q + W] = v;
        }
#pragma unroll
        for (int q = 0; q < 2 * W + 1; ++q) V[q] = N[q];
        fprev = c == 0 ? 0 : (diag_row ? 1 : 2);   // (a border cell's UP flag continues no interior move out of it but UP -- which leaves through column 0 itself)
    }
    return true;
}

}  // namespace pf
