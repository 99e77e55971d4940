// K-PAIR, device side: SeqAlign::SequenceAlignment (reference src/SeqAlign.cpp:550-640) for a bubble with exactly TWO paths of
// at most 64 (tier 1) or 128 (tier 2) bases, one THREAD per bubble.
//
// Two short paths are what nearly every bubble that needs dynamic programming looks like (a bi-allelic indel, or two SNPs
// closer than k).  A wavefront per bubble (K-BUBBLE) spends most of its instructions on the parts of the algorithm that are
// sequential per bubble -- the traceback walk, the column state machine, the ladder -- with 63 lanes watching; here every lane
// runs the whole algorithm for its own bubble, so a wavefront retires 64 bubbles for about the instruction count of a few.
// With two paths SequenceAlignment is its first round only: needlemanWunch + traceback of the pair (:480-549, 306-478) and
// compareStrPair over the kept alignments (:8-236).  This tier finishes the bubbles whose matrix has ONE optimal path (every
// cell on it carries a single direction flag: the reference's depth-first search then descends it once, keeps that alignment and
// unwinds without an alternative, and compareStrPair has nothing to choose); the others go on to K-BUBBLE.
//
// Memory.  The score row of the fill (ONE row: up / diagonal neighbours are the row's old values) lives in registers, the loop
// over the columns fully unrolled so that every row[j] is a named register; the second path as two bit planes, so that "does
// B[j] equal this row's base" is one bit-field extract.  Direction flags are 4 bits per cell, eight cells per register; the
// flags of the row above (their UP / DIAG bits are the reference's +1 for continuing a direction, :512-526) are the previous
// row's registers.  What leaves the registers is a BAND of each finished row: the optimal path of two paths that share their
// first and last k - 1 bases runs between the main diagonal and the diagonal through (m, n), so only the cells within PAIR_BAND
// of that corridor are kept -- STRIDE dwords per row and lane in per-wavefront global scratch, dword-interleaved across the lanes
// (lanes working in step touch one line per store).  A traceback that steps outside the band sends the bubble to K-BUBBLE like
// a tie does.  Round 2 kept a byte per cell of the whole matrix: 68 x 65 bytes per bubble, 745 MB of HBM writes per launch on
// the 5 M-unitig graph, for 56 MB of algorithmic traffic.
#pragma once
#include <hip/hip_runtime.h>
#include <limits.h>
#include <stdint.h>

#include "pf_align_dev.hpp"

namespace pf {

constexpr uint32_t PAIR_MAX = 64;     // longest path of tier 1
constexpr uint32_t PAIR_MAX2 = 128;   // ... of tier 2
constexpr int PAIR_BAND = 4;          // cells kept on either side of the corridor

// direction codes inside K-PAIR (4 bits per cell, bit 3 unused)
enum : uint32_t { PC_LEFT = 1, PC_DIAG = 2, PC_UP = 4 };

template <int NMAX>
struct PairGeom {
    static constexpr int NW = NMAX / 8;                    // dwords of direction codes per matrix row (columns 1 .. NMAX)
    static constexpr int STRIDE = NMAX == 128 ? 12 : 4;    // dwords of a row kept in scratch (NMAX = 64, 96: an indel of up to 8 bases)
    static constexpr int NA = NMAX / 32;                   // 64-bit words of a packed path / 32-bit words of a bit plane
    static constexpr uint32_t LEN = 2 * NMAX;              // longest alignment
    // |m - n| the band can follow whatever the alignment of its window to dword boundaries
    static constexpr int MAX_SKEW = 8 * (STRIDE - 2) - 2 * PAIR_BAND;
    static constexpr uint64_t dir_bytes = (uint64_t)(NMAX + 1) * STRIDE * 4 * 64;
    static constexpr uint64_t scratch_bytes = dir_bytes + 4ull * LEN * 64;   // + ra, rb (traceback order), fa, fb (forward)
};

template <int NMAX>
__host__ __device__ inline bool pair_fits(uint32_t m, uint32_t n) {
    const uint32_t mx = m > n ? m : n, d = m > n ? m - n : n - m;
    return mx <= (uint32_t)NMAX && d <= (uint32_t)PairGeom<NMAX>::MAX_SKEW;
}

#define PF_AT(p, e) (p)[(size_t)(e) * 64]

// first dword of row i's window: the corridor's leftmost column of that row, less the band
template <int NMAX>
__device__ inline int pair_window(int i, int dmin) {
    int w0 = (i + dmin - PAIR_BAND - 1) >> 3;   // (column j sits at nibble j - 1)
    w0 = w0 < 0 ? 0 : w0;
    const int last = PairGeom<NMAX>::NW - PairGeom<NMAX>::STRIDE;
    return w0 > last ? last : w0;
}

// 32 bases of a packed word (first base most significant) as two bit planes, base c at bit c
__device__ inline void pair_planes(uint64_t x, uint32_t &lo, uint32_t &hi) {
    // the odd bits (high bit of every base) and the even bits (low bit), each squeezed into 32 bits
    auto squeeze = [](uint64_t v) -> uint32_t {   // bits 0, 2, 4, ... of v -> bits 0 .. 31
        v &= 0x5555555555555555ull;
        v = (v | (v >> 1)) & 0x3333333333333333ull;
        v = (v | (v >> 2)) & 0x0F0F0F0F0F0F0F0Full;
        v = (v | (v >> 4)) & 0x00FF00FF00FF00FFull;
        v = (v | (v >> 8)) & 0x0000FFFF0000FFFFull;
        v = (v | (v >> 16)) & 0x00000000FFFFFFFFull;
        return (uint32_t)v;
    };
    // base c occupies bits 63 - 2c (high) and 62 - 2c (low): after the squeeze base c is at bit 31 - c; reversed: bit c
    lo = __brev(squeeze(x));
    hi = __brev(squeeze(x >> 1));
}

// needlemanWunch (src/SeqAlign.cpp:480-549) of A (m bases, packed) x B (n bases, bit planes): the band of every row's direction
// codes to `dir` (this lane's first dword).  Cells right of a lane's own n are computed and never read.
template <int NMAX, bool INTEGRAL>
__device__ inline void pair_fill(uint32_t *dir, const uint64_t (&Aw)[PairGeom<NMAX>::NA], const uint32_t (&b0)[PairGeom<NMAX>::NA],
                                 const uint32_t (&b1)[PairGeom<NMAX>::NA], uint32_t m, int dmin, double M, double D, double G, int Mi, int Di,
                                 int Gi) {
    using Gm = PairGeom<NMAX>;
    const int MDi = Mi - Di;   // (the scores as ints come from the host: converted in the loop, a select between the doubles and a
                               // v_cvt_i32_f64 per cell is what the compiler makes of `match ? (int)M : (int)D`)
    int row[NMAX + 1];
    uint32_t prev[Gm::NW], cur[Gm::NW];
    row[0] = 0;
#pragma unroll
    for (int j = 1; j <= NMAX; ++j) row[j] = INTEGRAL ? Gi * j : (int)(long)(G * (double)j);
#pragma unroll
    for (int w = 0; w < Gm::NW; ++w) { prev[w] = 0; cur[w] = 0; }   // row 0 carries LEFT only: no UP / DIAG bit to continue
    for (uint32_t i = 1; i <= m; ++i) {
        // this row's base of A, and where B equals it
        uint64_t aw = Aw[0];
#pragma unroll
        for (int x = 1; x < Gm::NA; ++x) aw = ((i - 1) >> 5) == (uint32_t)x ? Aw[x] : aw;
        const uint32_t a = (uint32_t)(aw >> (62 - 2 * ((i - 1) & 31))) & 3u;
        uint32_t eq[Gm::NA];
#pragma unroll
        for (int x = 0; x < Gm::NA; ++x) eq[x] = ((a & 1) ? b0[x] : ~b0[x]) & ((a & 2) ? b1[x] : ~b1[x]);
        int dgs = row[0];                                                     // (i - 1, 0): a border cell, no DIAG flag
        int last = INTEGRAL ? Gi * (int)i : (int)(long)(G * (double)i);       // (i, 0): border, flag UP
        row[0] = last;
        uint32_t lfbit = 0;
#pragma unroll
        for (int j = 1; j <= NMAX; ++j) {
            const uint32_t upbit = (prev[(j - 1) >> 3] >> (4 * ((j - 1) & 7) + 2)) & 1u;
            const uint32_t dgbit = j >= 2 ? (prev[(j - 2) >> 3] >> (4 * ((j - 2) & 7) + 1)) & 1u : 0u;
            const uint32_t match = (eq[(j - 1) >> 5] >> ((j - 1) & 31)) & 1u;
            const int ups = row[j];
            int up, dg, lf;
            if (INTEGRAL) {
                up = ups + Gi + (int)upbit;
                dg = dgs + ((int)match * MDi + Di) + (int)dgbit;
                lf = last + Gi + (int)lfbit;
            } else {   // the reference's `int = long + double`: truncation at every cell
                up = (int)((double)ups + G) + (int)upbit;
                dg = (int)((double)dgs + (match ? M : D)) + (int)dgbit;
                lf = (int)((double)last + G) + (int)lfbit;
            }
            int best = up > dg ? up : dg;
            best = best > lf ? best : lf;
            lfbit = lf == best ? 1u : 0u;
            const uint32_t code = ((up == best ? 1u : 0u) << 2) | ((dg == best ? 1u : 0u) << 1) | lfbit;
            cur[(j - 1) >> 3] |= code << (4 * ((j - 1) & 7));
            dgs = ups;
            row[j] = best;
            last = best;
        }
        const int w0 = pair_window<NMAX>((int)i, dmin);
        uint32_t *drow = dir + (size_t)i * Gm::STRIDE * 64;
#pragma unroll
        for (int w = 0; w < Gm::NW; ++w) {
            const int rel = w - w0;
            if (rel >= 0 && rel < Gm::STRIDE) drow[(size_t)rel * 64] = cur[w];
            prev[w] = cur[w];
            cur[w] = 0;
        }
    }
}

struct PairMem {   // this lane's view of the wavefront's scratch: element e of an array at base[e * 64]
    uint32_t *dir;        // (NMAX + 1) * STRIDE dwords
    char *ra, *rb;        // the alignment in traceback order ('+' = gap in A), LEN each
    char *fa, *fb;        // ... front to back, LEN each
};

template <int NMAX>
__device__ inline char pair_base(const uint64_t (&w)[PairGeom<NMAX>::NA], uint32_t idx) {
    uint64_t x = w[0];
#pragma unroll
    for (int q = 1; q < PairGeom<NMAX>::NA; ++q) x = (idx >> 5) == (uint32_t)q ? w[q] : x;
    return pf::base_char((uint32_t)((uint32_t)(x >> (62 - 2 * (idx & 31))) & 3u));
}

// traceback (src/SeqAlign.cpp:306-478) when the matrix has one optimal path within the 5-gap-open budgets: the reference's
// search descends it, keeps it and finds no alternative.  Returns the alignment's length with the rows in mem.fa / mem.fb, or 0
// when a cell on the way carries several flags, a budget is hit, or the walk leaves the stored band (K-BUBBLE decides those).
template <int NMAX>
__device__ inline uint32_t pair_traceback(const PairMem &mem, const uint64_t (&Aw)[PairGeom<NMAX>::NA], const uint64_t (&Bw)[PairGeom<NMAX>::NA],
                                          uint32_t m, uint32_t n, int dmin) {
    using Gm = PairGeom<NMAX>;
    uint32_t i = m, j = n, oa = 0, ob = 0, len = 0;
    char fa = '\0', fb = '\0';
    while (i > 0 || j > 0) {
        uint32_t f;
        if (i == 0) f = PC_LEFT;          // the borders (:486-496)
        else if (j == 0) f = PC_UP;
        else {
            const int rel = (int)((j - 1) >> 3) - pair_window<NMAX>((int)i, dmin);
            if (rel < 0 || rel >= Gm::STRIDE) return 0;
            f = (mem.dir[((size_t)i * Gm::STRIDE + (uint32_t)rel) * 64] >> (4 * ((j - 1) & 7))) & 7u;
        }
        char ca, cb;
        if (f == PC_DIAG) { ca = pair_base<NMAX>(Aw, i - 1); cb = pair_base<NMAX>(Bw, j - 1); i--; j--; }
        else if (f == PC_LEFT) {
            if (oa >= 5) return 0;
            if (len == 0 || fa != '+') ++oa;
            ca = '+';
            cb = pair_base<NMAX>(Bw, j - 1);
            j--;
        } else if (f == PC_UP) {
            if (ob >= 5) return 0;
            if (len == 0 || fb == '-') ++ob;
            ca = pair_base<NMAX>(Aw, i - 1);
            cb = '-';
            i--;
        } else return 0;
        PF_AT(mem.ra, len) = ca;
        PF_AT(mem.rb, len) = cb;
        fa = ca;
        fb = cb;
        len++;
    }
    for (uint32_t t = 0; t < len; ++t) {
        const char x = PF_AT(mem.ra, len - 1 - t);
        PF_AT(mem.fa, t) = x == '+' ? '-' : x;
        PF_AT(mem.fb, t) = PF_AT(mem.rb, len - 1 - t);
    }
    return len;
}

// what compareStrPair's column pass (src/SeqAlign.cpp:56-157) yields for two rows: the variant columns (column, "opens an
// indel") and the indel lengths.  With EMIT they are written out, otherwise counted.
struct PairCounts {
    uint32_t n_sites, n_indel_len;
};
template <bool EMIT>
__device__ inline PairCounts pair_classify(const char *x, const char *y, uint32_t L, pf_bubble_site *sites, uint32_t *ilen) {
    uint32_t ns = 0, nl = 0, last_indel_pos = 0;
    bool open = false;
    char pa = 0, pb = 0;
    for (uint32_t j = 0; j < L; ++j) {
        const char a = PF_AT(x, j), b = PF_AT(y, j);
        const int t = a == b ? 0 : ((a == '-' || b == '-') ? 2 : 1);
        if (t != 2) {
            if (open) { if (EMIT) ilen[nl] = j - last_indel_pos; nl++; open = false; }
            if (t == 1) {
                if (EMIT) { pf_bubble_site sr; sr.col = j; sr.is_indel = 0; sr.maxnum = 2; sr.pad_ = 0; sites[ns] = sr; }
                ns++;
            }
        } else {
            const bool same_status = j > 0 && ((a == '-') == (pa == '-')) && ((b == '-') == (pb == '-'));
            const bool same_run = open && same_status;
            if (open && !same_run) { if (EMIT) ilen[nl] = j - last_indel_pos; nl++; }
            if (!same_run) {
                last_indel_pos = j;
                open = true;
                if (EMIT) { pf_bubble_site sr; sr.col = j; sr.is_indel = 1; sr.maxnum = 2; sr.pad_ = 0; sites[ns] = sr; }
                ns++;
            }
        }
        pa = a;
        pb = b;
    }
    return PairCounts{ns, nl};
}

}  // namespace pf
