// K-BUBBLE: the whole of SeqAlign::SequenceAlignment (reference src/SeqAlign.cpp:550-640) for one
// bubble on one wavefront.
//
//   round 0     needlemanWunch + traceback of paths 0 and 1 (pf_align_dev.hpp) -> kept alignments
//   round i>=2  for every kept alignment: NW + traceback of its row 0 against path i; for every
//               older row j the new gaps are re-opened (src/SeqAlign.cpp:583-597), the row is
//               re-scored against the new row as variantAnalyze does (:237-305) and only the
//               best-ranked candidates (AlignUnit::operator-, SeqAlign.hpp:43-67) stay alive; the
//               per-alignment totals decide which groups of candidates survive (:619-636)
//   choose      compareStrPair (:8-236): column classification of every surviving alignment and
//               the seven-step selection ladder; the winner's rows, variant columns, allele groups
//               and indel lengths are published
//
// Work split inside the wave: the sequential decisions (candidate lists, ladder) are executed
// redundantly by all 64 lanes on wave-uniform values; everything that touches characters is
// lane-parallel over columns -- path decoding from the 2-bit graph, gap re-opening (each lane finds
// its column's source by counting gaps), row scoring (per-column score / new-indel flags, wave
// reductions; a lane-0 chain only for non-integral scores, where the reference's `long += double`
// truncates at every column), column classification, row copies.  Alignments live in two per-wave
// arenas in global memory (ping-pong between rounds); the NW matrices are in LDS as in K-ALN.
// No MFMA: byte compares and integer reductions.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "pf_align_dev.hpp"
#include "pf_bubble_launch.hpp"
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

struct MsaRef {
    uint32_t off;   // rows * len chars, row-major, in the arena
    uint32_t len;
    uint32_t rows;
};

struct BubCaps {  // per-wave scratch capacities (all multiples of 8)
    uint32_t st_text, st_gaps, st_hits;  // traceback staging
    uint32_t pbuf;                       // decoded path strings
    uint32_t arena;                      // each of the two alignment arenas
    uint32_t list;                       // alignments per list
    uint32_t row;                        // longest aligned row
};

__host__ __device__ inline uint64_t bub_scratch_bytes(const BubCaps &c) {
    return (uint64_t)c.st_text + 4ull * c.st_gaps + sizeof(pf_align_hit) * (uint64_t)c.st_hits + c.pbuf + 2ull * c.arena +
           2ull * sizeof(MsaRef) * c.list + 2ull * 2 * c.st_hits + 2ull * c.row + 3ull * 4 * c.row + 64;
}

struct BubParams {
    const char *text;
    const pf_bubble_path *paths;
    const pf_bubble_task *tasks;
    const uint32_t *idx;
    uint32_t n;
    double M, D, G;
    int integral;
    const uint64_t *seq;
    const uint64_t *off;
    const uint32_t *len;
    uint32_t n_unitigs;
    uint8_t *scratch;
    uint64_t scratch_per_wave;
    BubCaps caps;
    uint8_t *work;          // NW working storage when not in LDS
    uint64_t work_per_wave;
    uint32_t work_bytes;    // bytes available per wave for one NW job (LDS or global)
    int final_tier;
    unsigned long long *task_clk;  // diagnostic: per-task wall clock ticks (100 MHz), or nullptr
    unsigned int *next;            // work queue head of this launch
    unsigned long long *prof;      // diagnostic: [0] fill [1] traceback [2] decode [3] round0 copy [4] rounds [5] choose [6] publish
    unsigned int *bad;             // final tier: bubbles that exceed even its capacities
};

struct BubOut {
    pf_bubble_result *res;
    char *text;
    uint64_t text_cap;
    pf_bubble_site *sites;
    uint64_t site_cap;
    uint8_t *groups;
    uint64_t group_cap;
    uint32_t *ilen;
    uint64_t ilen_cap;
    unsigned long long *heads;  // [0] text, [1] sites, [2] groups, [3] ilen
    uint32_t *retry;
    unsigned int *n_retry;
};

// Per-wave bump allocation in the four output pools (one atomic per chunk instead of four per bubble).
struct BubAlloc {
    unsigned long long cur[4] = {0, 0, 0, 0}, end[4] = {0, 0, 0, 0};
};
__device__ inline unsigned long long bub_take(const BubOut &o, BubAlloc &al, int pool, unsigned long long n, unsigned long long chunk) {
    if (n == 0) return al.cur[pool];
    if (n > al.end[pool] - al.cur[pool]) {
        const unsigned long long want = n > chunk ? n : chunk;
        unsigned long long got = 0;
        if (lane_id() == 0) got = atomicAdd(&o.heads[pool], want);
        got = ((unsigned long long)__shfl((uint32_t)(got >> 32), 0, WAVE) << 32) | __shfl((uint32_t)got, 0, WAVE);
        al.cur[pool] = got;
        al.end[pool] = got + want;
    }
    const unsigned long long off = al.cur[pool];
    al.cur[pool] += n;
    return off;
}

struct RowScore {
    long long score;
    uint32_t n_pos, indel;
};

__device__ inline long long bcast_i64(long long v) {
    const uint32_t lo = __shfl((uint32_t)v, 0, WAVE), hi = __shfl((uint32_t)((unsigned long long)v >> 32), 0, WAVE);
    return (long long)(((unsigned long long)hi << 32) | lo);
}

// AlignUnit::operator- truncated to int (src/SeqAlign.hpp:43-67)
__device__ inline int rank_diff(const RowScore &l, const RowScore &r) {
    long long d;
    if (l.score != r.score) d = l.score > r.score ? 1 : -1;
    else if (l.n_pos != r.n_pos) d = (long long)r.n_pos - (long long)l.n_pos;
    else d = (long long)r.indel - (long long)l.indel;
    return (int)d;
}

// column p of `row` after re-opening gaps (gaps in traceback order = descending; src/SeqAlign.cpp:583-597)
__device__ inline char regap_char(const char *row, const uint32_t *gaps, uint32_t ng, uint32_t p) {
    uint32_t c = 0;
    while (c < ng && gaps[ng - 1 - c] + c < p) ++c;
    if (c < ng && gaps[ng - 1 - c] + c == p) return '-';
    return row[p - c];
}

// variantAnalyze (src/SeqAlign.cpp:237-305) of two equal-length rows; wave-uniform result
__device__ inline RowScore score_rows(const char *x, const char *y, uint32_t L, double M, double D, double G, int integral) {
    const int lane = lane_id();
    long long isum = 0;
    uint32_t npos = 0, nind = 0;
    for (uint32_t base = 0; base < L; base += WAVE) {
        const uint32_t p = base + lane;
        if (p < L) {
            const char a = x[p], b = y[p];
            if (integral) isum += (long long)((a == '-' || b == '-') ? G : (a == b ? M : D));
            if (a != b) {
                const int side = a == '-' ? 1 : (b == '-' ? 2 : 3);
                if (side == 3) {
                    npos++;
                } else {
                    int prev_run = 0;  // gap side of the previous column, 0 if it was equal or a snp
                    if (p > 0) {
                        const char pa = x[p - 1], pb = y[p - 1];
                        if (pa != pb) prev_run = pa == '-' ? 1 : (pb == '-' ? 2 : 0);
                    }
                    if (prev_run != side) { npos++; nind++; }
                }
            }
        }
    }
    RowScore r;
    r.n_pos = (uint32_t)wave_sum_u64(npos);
    r.indel = (uint32_t)wave_sum_u64(nind);
    r.n_pos = __shfl(r.n_pos, 0, WAVE);
    r.indel = __shfl(r.indel, 0, WAVE);
    if (integral) {
        r.score = bcast_i64((long long)wave_sum_u64((uint64_t)isum));
    } else {
        long long sc = 0;
        if (lane == 0)
            for (uint32_t p = 0; p < L; ++p) {
                const char a = x[p], b = y[p];
                const double s = (a == '-' || b == '-') ? G : (a == b ? M : D);
                sc = (long long)((double)sc + s);
            }
        r.score = bcast_i64(sc);
    }
    return r;
}

// strcmp(a, b) > 0 for strings of length la / lb without NULs; wave-uniform
__device__ inline bool str_greater(const char *a, uint32_t la, const char *b, uint32_t lb) {
    const int lane = lane_id();
    const uint32_t L = la < lb ? la : lb;
    for (uint32_t base = 0; base < L; base += WAVE) {
        const uint32_t p = base + lane;
        const bool diff = p < L && a[p] != b[p];
        const unsigned long long m = __ballot(diff);
        if (m) {
            const uint32_t q = base + (uint32_t)__ffsll((long long)m) - 1;
            return (unsigned char)a[q] > (unsigned char)b[q];
        }
    }
    return la > lb;
}

// compute_dis (src/SeqAlign.cpp:10-38) over the merge of two ascending lists (either may be empty)
__device__ inline uint64_t spread_merged(const uint32_t *a, uint32_t na, const uint32_t *b, uint32_t nb, uint64_t L) {
    const uint32_t n = na + nb;
    if (n == 0) return 0;
    uint32_t ia = 0, ib = 0;
    auto next = [&]() -> uint32_t {
        // std::merge takes from the second range only when it is strictly smaller
        if (ib < nb && (ia >= na || b[ib] < a[ia])) return b[ib++];
        return a[ia++];
    };
    const uint32_t v0 = next();
    if (n == 1) {
        const int left = (int)v0, right = (int)(L - v0) - 1;
        return left > right ? (uint64_t)(left + 1) : (uint64_t)right;
    }
    uint64_t d = v0;
    uint32_t prev = v0;
    for (uint32_t i = 1; i < n; ++i) {
        const uint32_t v = next();
        const int gap = (int)(v - prev - 1);
        d = (uint64_t)(gap < (int)d ? gap : (int)d);
        prev = v;
    }
    const uint64_t tail = L - prev - 1;
    return d < tail ? d : tail;
}

struct Metrics {
    int snp, indel;  // the reference's 8-bit counters, widened after wrap
    uint32_t n_snp_pos, n_indel_pos, n_indel_len;
    int first_site, last_site;  // of the merged list (0 when empty, as the reference's UB reads do here)
    bool any_site;
};

// Column classification of one alignment (src/SeqAlign.cpp:56-157): per-column pass lane-parallel,
// list building sequential over one flag byte per column.  Fills snp_pos / indel_pos / indel_len.
__device__ inline Metrics classify(const char *rows, uint32_t R, uint32_t L, uint8_t *colinfo, uint32_t *snp_pos,
                                   uint32_t *indel_pos, uint32_t *indel_len) {
    const int lane = lane_id();
    for (uint32_t base = 0; base < L; base += WAVE) {
        const uint32_t j = base + lane;
        if (j < L) {
            char seen[5];
            int n_seen = 0;
            bool has_gap = false, same_status = j > 0;
            for (uint32_t r = 0; r < R; ++r) {
                const char c = rows[(size_t)r * L + j];
                int q = 0;
                while (q < n_seen && seen[q] != c) ++q;
                if (q == n_seen && n_seen < 5) seen[n_seen++] = c;
                has_gap |= c == '-';
                if (j > 0 && ((c == '-') != (rows[(size_t)r * L + j - 1] == '-'))) same_status = false;
            }
            uint8_t t = 0;
            if (n_seen > 1) t = has_gap ? 2 : 1;
            colinfo[j] = (uint8_t)(t | (same_status ? 4 : 0) | (n_seen > 2 ? 8 : 0));
        }
    }
    aln_sync();
    Metrics m;
    uint8_t snp = 0, indel = 0;
    uint32_t ns = 0, ni = 0, nl = 0;
    bool open = false;
    // List building (src/SeqAlign.cpp:56-157) is a state machine over the columns, but only two kinds of column can change its
    // state or emit anything: variant columns, and the column right after a gap column (where an open indel closes).  Those are
    // found 64 columns at a time by ballot and visited in order; the long runs of identical columns between them are skipped.
    uint32_t prev_gap_carry = 0;  // was the last column of the previous 64-block a gap column?
    for (uint32_t base = 0; base < L; base += WAVE) {
        const uint32_t jj = base + lane;
        const uint8_t cj = jj < L ? colinfo[jj] : 0;
        const unsigned long long m_var = __ballot((cj & 3) != 0);
        const unsigned long long m_gap = __ballot((cj & 3) == 2);
        unsigned long long ev = m_var | (m_gap << 1) | (unsigned long long)prev_gap_carry;
        if (base + WAVE > L && L - base < 64) ev &= (1ull << (L - base)) - 1;
        prev_gap_carry = (uint32_t)(m_gap >> 63);
        while (ev) {
            const int b = __ffsll((long long)ev) - 1;
            ev &= ev - 1;
            const uint32_t j = base + (uint32_t)b;
            const uint8_t ci = (uint8_t)read_lane((uint32_t)cj, b);
            const uint8_t t = ci & 3;
            uint8_t lab = 0;  // bit 4: labelled column, bit 5: opens an indel
            if (t != 2) {
                if (open) { if (lane == 0) indel_len[nl] = j - indel_pos[(uint8_t)(indel - 1)]; nl++; open = false; }
                if (t == 1) { if (lane == 0) snp_pos[ns] = j; ns++; snp++; lab = 16; }
            } else {
                const bool same_run = open && (ci & 4);
                if (open && !same_run) { if (lane == 0) indel_len[nl] = j - indel_pos[(uint8_t)(indel - 1)]; nl++; }
                if (!same_run) {
                    ++indel;
                    if (lane == 0) indel_pos[ni] = j;
                    ni++;
                    open = true;
                    lab = 16 | 32;  // (indel_pos is written and read back by lane 0 only)
                } else if (ci & 8) {
                    lab = 16;
                }
            }
            if (lane == 0 && lab) colinfo[j] = (uint8_t)(ci | lab);
        }
    }
    aln_sync();
    m.snp = snp;
    m.indel = indel;
    m.n_snp_pos = ns;
    m.n_indel_pos = ni;
    m.n_indel_len = nl;
    m.any_site = ns + ni > 0;
    int f = 0, l = 0;
    if (ns + ni) {
        const uint32_t s0 = ns ? snp_pos[0] : 0xFFFFFFFFu, i0 = ni ? indel_pos[0] : 0xFFFFFFFFu;
        f = (int)(s0 < i0 ? s0 : i0);
        const uint32_t s1 = ns ? snp_pos[ns - 1] : 0, i1 = ni ? indel_pos[ni - 1] : 0;
        l = (int)(s1 > i1 ? s1 : i1);
    }
    m.first_site = f;
    m.last_site = l;
    return m;
}

// returns 0 ok, 1 = a capacity of this tier was exceeded (rerun in the next tier)
template <bool LDS>
__device__ int bubble_task(const BubParams &p, const BubOut &o, BubAlloc &al, uint32_t job, uint8_t *nw_base, uint8_t *scr) {
    const int lane = lane_id();
    const BubCaps &C = p.caps;
    // carve the per-wave scratch
    AlnScratch st;
    st.text = reinterpret_cast<char *>(scr);
    st.gaps = reinterpret_cast<uint32_t *>(scr + C.st_text);
    st.hits = reinterpret_cast<pf_align_hit *>(scr + C.st_text + 4ull * C.st_gaps);
    st.text_cap = C.st_text;
    st.gap_cap = C.st_gaps;
    st.hit_cap = C.st_hits;
    uint8_t *q = scr + C.st_text + 4ull * C.st_gaps + sizeof(pf_align_hit) * (uint64_t)C.st_hits;
    char *pbuf = reinterpret_cast<char *>(q);
    q += C.pbuf;
    char *arena[2] = {reinterpret_cast<char *>(q), reinterpret_cast<char *>(q + C.arena)};
    q += 2ull * C.arena;
    MsaRef *list[2] = {reinterpret_cast<MsaRef *>(q), reinterpret_cast<MsaRef *>(q + sizeof(MsaRef) * C.list)};
    q += 2ull * sizeof(MsaRef) * C.list;
    uint16_t *alive[2] = {reinterpret_cast<uint16_t *>(q), reinterpret_cast<uint16_t *>(q + 2ull * C.st_hits)};
    q += 2ull * 2 * C.st_hits;
    char *tmp_row = reinterpret_cast<char *>(q);
    q += C.row;
    uint8_t *colinfo = q;
    q += C.row;
    uint32_t *snp_pos = reinterpret_cast<uint32_t *>(q);
    uint32_t *indel_pos = snp_pos + C.row;
    uint32_t *indel_len = indel_pos + C.row;

    const pf_bubble_task tk = p.tasks[job];
    const uint32_t N = tk.n_paths;
    const pf_bubble_path *paths = p.paths + tk.path_first;

    unsigned long long tq = p.prof ? wall_clock64() : 0;
    auto mark = [&](int slot) {
        if (!p.prof) return;
        const unsigned long long now = wall_clock64();
        if (lane == 0) atomicAdd(&p.prof[slot], now - tq);
        tq = now;
    };
    // ---- path strings into pbuf (decode oriented unitigs from the 2-bit graph) ------------------
    uint32_t poff_total = 0;
    for (uint32_t i = 0; i < N; ++i) poff_total += paths[i].len;
    if (poff_total > C.pbuf) return 1;
    {
        uint32_t at = 0;
        for (uint32_t i = 0; i < N; ++i) {
            const pf_bubble_path pp = paths[i];
            if (pp.ov == PF_NONE) {
                const char *src = p.text + pp.text_off;
                for (uint32_t t = lane; t < pp.len; t += WAVE) pbuf[at + t] = src[t];
            } else {
                const uint32_t u = pp.ov >> 1;
                const uint64_t *w = p.seq + p.off[u];
                const bool rev = (pp.ov & 1) != 0;
                for (uint32_t t = lane; t < pp.len; t += WAVE) {
                    const uint32_t j = rev ? pp.len - 1 - t : t;
                    const uint32_t b = (uint32_t)(w[j >> 5] >> (62 - 2 * (j & 31))) & 3u;
                    pbuf[at + t] = pf::base_char((uint32_t)(rev ? 3 - b : b));
                }
            }
            at += pp.len;
        }
    }
    aln_sync();
    auto path_ptr = [&](uint32_t i) -> const char * {
        uint32_t at = 0;
        for (uint32_t x = 0; x < i; ++x) at += paths[x].len;
        return pbuf + at;
    };

    mark(2);
    // ---- round 0 ---------------------------------------------------------------------------------
    int cur = 0;
    uint32_t n_kept = 0, used[2] = {0, 0};
    bool shortcut_scores = false;   // the scores make a single mismatch on the diagonal the strict optimum (proof below)
    bool gapless = false;           // the one kept alignment is the paths themselves, stacked (every round so far took the shortcut)
    {
        const uint32_t m = paths[0].len, n = paths[1].len;
        if (job_bytes(m, n) > p.work_bytes) return 1;
        // Two equally long paths that differ in exactly one base -- the bi-allelic SNP bubble, most of all bubbles --
        // need no dynamic programming when the scores make the diagonal the strict optimum of every diagonal cell.
        // With b = 1 the bonus for continuing a direction (src/SeqAlign.cpp:512-526) and h <= 1 mismatches so far:
        //   S(i,i) = sum of substitution scores + (i-1) b, reached by the diagonal move alone, as long as it beats
        //   the moves out of the neighbours (i-1,i) and (i,i-1);
        //   any cell obeys S(i,j) <= min(i,j) (M+b) + |i-j| (G+b) when M >= D and M + b >= 2 (G + b);
        //   so diag - up >= M - h (M-D) - 2G - 2b >= D - 2G - 2b.
        // If D - 2G - 2b > 0 every diagonal cell carries the single flag DIAG, the traceback finds exactly one
        // alignment -- the two strings unchanged -- and all the tie-breaking machinery has nothing to decide.
        bool snp_only = false;
        shortcut_scores = p.integral && p.M >= p.D && p.M + 1 >= 2 * (p.G + 1) && p.D - 2 * p.G - 2 > 0;
        if (shortcut_scores && m == n) {
            const char *x = path_ptr(0), *y = path_ptr(1);
            uint32_t diff = 0;
            for (uint32_t t = lane; t < m; t += WAVE) diff += x[t] != y[t];
            snp_only = read_lane((uint32_t)wave_sum_u64(diff), 0) == 1;
        }
        uint32_t nh, tu, gu;
        if (snp_only && N == 2) {
            // ... and with only these two paths there is nothing left to choose or classify: the answer is the two rows,
            // one SNP column, groups {1, 2} (what classify + publish below produce for such rows)
            const char *x = path_ptr(0), *y = path_ptr(1);
            uint32_t my_pos = 0;
            bool mine = false;
            for (uint32_t t = lane; t < m; t += WAVE)
                if (x[t] != y[t]) { my_pos = t; mine = true; }
            const uint32_t col = read_lane(my_pos, __ffsll((long long)__ballot(mine)) - 1);
            const unsigned long long t0 = bub_take(o, al, 0, 2ull * m, 4096);
            const unsigned long long s0 = bub_take(o, al, 1, 1, 64);
            const unsigned long long g0 = bub_take(o, al, 2, 2, 256);
            const unsigned long long l0 = bub_take(o, al, 3, 0, 32);
            pf_bubble_result res;
            memset(&res, 0, sizeof(res));
            res.rows_off = t0;
            res.site_off = s0;
            res.group_off = g0;
            res.ilen_off = l0;
            res.n_rows = 2;
            res.n_cols = m;
            res.n_sites = 1;
            res.n_indel_len = 0;
            if (lane == 0) o.res[job] = res;
            if (t0 + 2ull * m > o.text_cap || s0 + 1 > o.site_cap || g0 + 2 > o.group_cap || l0 > o.ilen_cap) return 0;
            for (uint32_t t = lane; t < 2 * m; t += WAVE) o.text[t0 + t] = x[t];  // paths 0 and 1 are adjacent in pbuf
            if (lane == 0) {
                o.groups[g0] = 1;
                o.groups[g0 + 1] = 2;
                pf_bubble_site sr;
                sr.col = col;
                sr.is_indel = 0;
                sr.maxnum = 2;
                sr.pad_ = 0;
                o.sites[s0] = sr;
            }
            mark(6);
            return 0;
        }
        if (snp_only) {
            if (m > C.row || 2ull * m > C.arena || C.list < 1) return 1;
            const char *src = path_ptr(0);  // paths 0 and 1 are adjacent in pbuf: the two rows, as traceback would emit them
            char *dst = arena[cur];
            for (uint32_t t = lane; t < 2 * m; t += WAVE) dst[t] = src[t];
            if (lane == 0) list[cur][0] = MsaRef{0, m, 2};
            used[cur] = 2 * m;
            nh = 0;
            n_kept = 1;
            gapless = true;
            aln_sync();
        } else {
        if (!align_job(nw_base, path_ptr(0), path_ptr(1), m, n, p.M, p.D, p.G, p.integral, st, nh, tu, gu, p.prof)) return 1;
        if (nh > C.list) return 1;
        for (uint32_t h = 0; h < nh; ++h) {
            const pf_align_hit hh = st.hits[h];
            if (hh.len > C.row || used[cur] + 2ull * hh.len > C.arena) return 1;
            const char *src = st.text + hh.text_off;
            char *dst = arena[cur] + used[cur];
            for (uint32_t t = lane; t < 2 * hh.len; t += WAVE) dst[t] = src[t];
            if (lane == 0) list[cur][h] = MsaRef{used[cur], hh.len, 2};
            used[cur] += 2 * hh.len;
        }
        n_kept = nh;
        aln_sync();
        }
    }

    mark(3);
    // ---- progressive rounds (src/SeqAlign.cpp:559-638) ------------------------------------------
    for (uint32_t i = 2; i < N; ++i) {
        const int nxt = cur ^ 1;
        uint32_t n_new = 0;
        used[nxt] = 0;
        int best_total = INT_MIN;
        const char *pi = path_ptr(i);
        const uint32_t li = paths[i].len;
        if (gapless && n_kept == 1 && shortcut_scores && list[cur][0].len == li) {
            // The shortcut of round 0 again: row 0 is path 0 itself (no gap so far) and path i is as long and differs from it in
            // exactly one base -- needlemanWunch(row 0, path i) then has the single optimal alignment "both unchanged" (the
            // same proof: h = 1), the one candidate re-opens no gap in the older rows and survives every ranking alone.
            // The new alignment is the old one with path i underneath.
            const MsaRef M = list[cur][0];
            const char *mrows = arena[cur] + M.off;
            uint32_t diff = 0;
            for (uint32_t t = lane; t < li; t += WAVE) diff += mrows[t] != pi[t];
            if (read_lane((uint32_t)wave_sum_u64(diff), 0) == 1) {
                const uint32_t rows = i + 1;
                if ((uint64_t)rows * li > C.arena || C.list < 1) return 1;
                char *dst = arena[nxt];
                for (uint32_t t = lane; t < i * li; t += WAVE) dst[t] = mrows[t];
                for (uint32_t t = lane; t < li; t += WAVE) dst[(size_t)i * li + t] = pi[t];
                if (lane == 0) list[nxt][0] = MsaRef{0, li, rows};
                used[nxt] = rows * li;
                aln_sync();
                cur = nxt;
                n_kept = 1;
                continue;
            }
        }
        gapless = false;
        for (uint32_t kk = 0; kk < n_kept; ++kk) {
            const MsaRef M = list[cur][kk];
            const char *mrows = arena[cur] + M.off;
            if (job_bytes(M.len, li) > p.work_bytes) return 1;
            uint32_t nh, tu, gu;
            if (!align_job(nw_base, mrows, pi, M.len, li, p.M, p.D, p.G, p.integral, st, nh, tu, gu, p.prof)) return 1;
            // candidates alive, in traceback order
            uint32_t n_alive = nh;
            int ab = 0;
            for (uint32_t c = lane; c < nh; c += WAVE) alive[ab][c] = (uint16_t)c;
            aln_sync();
            uint32_t total = 0;  // `int` in the reference; sums of INT_MIN wrap
            for (uint32_t j = 1; j < i; ++j) {
                RowScore top;
                top.score = INT_MIN;
                top.n_pos = 0;
                top.indel = 0;
                int best_j = INT_MIN;
                uint32_t n_next = 0;
                for (uint32_t a = 0; a < n_alive; ++a) {
                    const uint32_t c = alive[ab][a];
                    const pf_align_hit hh = st.hits[c];
                    if (hh.len > C.row) return 1;
                    const char *old = mrows + (size_t)j * M.len;
                    const uint32_t *gp = st.gaps + hh.gap_off;
                    for (uint32_t t = lane; t < hh.len; t += WAVE) tmp_row[t] = regap_char(old, gp, hh.n_gaps, t);
                    aln_sync();
                    const RowScore rs = score_rows(tmp_row, st.text + hh.text_off + hh.len, hh.len, p.M, p.D, p.G, p.integral);
                    const int diff = rank_diff(rs, top);
                    if (diff > 0) { top = rs; n_next = 0; }
                    if (diff >= 0) {
                        best_j = (int)top.score;
                        if (lane == 0) alive[ab ^ 1][n_next] = (uint16_t)c;
                        n_next++;
                    }
                    aln_sync();
                }
                ab ^= 1;
                n_alive = n_next;
                total += (uint32_t)best_j;
            }
            const int tkk = (int)total;
            if (tkk > best_total) { best_total = tkk; n_new = 0; used[nxt] = 0; }
            if (tkk >= best_total) {
                for (uint32_t a = 0; a < n_alive; ++a) {
                    const uint32_t c = alive[ab][a];
                    const pf_align_hit hh = st.hits[c];
                    const uint32_t rows = i + 1;
                    if (n_new >= C.list || hh.len > C.row || used[nxt] + (uint64_t)rows * hh.len > C.arena) return 1;
                    char *dst = arena[nxt] + used[nxt];
                    const char *ha = st.text + hh.text_off;
                    const uint32_t *gp = st.gaps + hh.gap_off;
                    for (uint32_t t = lane; t < hh.len; t += WAVE) {
                        dst[t] = ha[t];                                      // new row 0
                        dst[(size_t)i * hh.len + t] = ha[hh.len + t];         // the new row
                        for (uint32_t j = 1; j < i; ++j)
                            dst[(size_t)j * hh.len + t] = regap_char(mrows + (size_t)j * M.len, gp, hh.n_gaps, t);
                    }
                    if (lane == 0) list[nxt][n_new] = MsaRef{used[nxt], hh.len, rows};
                    n_new++;
                    used[nxt] += rows * hh.len;
                }
            }
            aln_sync();
        }
        cur = nxt;
        n_kept = n_new;
    }

    mark(4);
    // ---- compareStrPair (src/SeqAlign.cpp:8-236) ---------------------------------------------------
    int best = -1;
    {
        const uint64_t Lref = n_kept ? list[cur][n_kept - 1].len : 0;
        int best_snp = INT_MAX / 2, best_indel = INT_MAX / 2;
        int d_snp = INT_MAX, d_indel = INT_MAX, d_all = INT_MAX, left = -1, right = -1;
        // a single surviving alignment wins whatever its metrics (the first candidate always beats the
        // initial INT_MAX/2 counts), so the ladder is only run when there is a choice
        if (n_kept == 1) best = 0;
        for (uint32_t c = 0; n_kept > 1 && c < n_kept; ++c) {
            const MsaRef M = list[cur][c];
            const char *rows = arena[cur] + M.off;
            // the NW working storage is idle now: keep the per-column flags there (LDS in the LDS tiers)
            uint8_t *ci = M.len <= p.work_bytes ? nw_base : colinfo;
            const Metrics m = classify(rows, M.rows, M.len, ci, snp_pos, indel_pos, indel_len);
            int verdict = 0;  // 1 take, 2 take on the strcmp tie-break
            uint64_t c_indel = 0, c_snp = 0, c_all = 0;
            const int total = m.snp + m.indel, btotal = best_snp + best_indel;
            if (total < btotal) verdict = 1;
            else if (total == btotal) {
                if (m.indel < best_indel) verdict = 1;
                else if (m.indel == best_indel) {
                    c_indel = spread_merged(indel_pos, m.n_indel_pos, nullptr, 0, Lref);
                    if (c_indel > (uint64_t)d_indel) verdict = 1;
                    else if (c_indel == (uint64_t)d_indel) {
                        c_snp = spread_merged(snp_pos, m.n_snp_pos, nullptr, 0, Lref);
                        if (c_snp > (uint64_t)d_snp) verdict = 1;
                        else if (c_snp == (uint64_t)d_snp) {
                            c_all = spread_merged(snp_pos, m.n_snp_pos, indel_pos, m.n_indel_pos, Lref);
                            if (c_all > (uint64_t)d_all) verdict = 1;
                            else if (c_all == (uint64_t)d_all) {
                                if (m.first_site > left || m.last_site > right) verdict = 1;
                                else if (m.first_site == left && m.last_site == right && best >= 0) {
                                    const MsaRef B = list[cur][best];
                                    const char *brows = arena[cur] + B.off;
                                    for (uint32_t r = 0; r < M.rows; ++r)
                                        if (str_greater(rows + (size_t)r * M.len, M.len, brows + (size_t)r * B.len, B.len)) {
                                            verdict = 2;
                                            break;
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
                c_all = spread_merged(snp_pos, m.n_snp_pos, indel_pos, m.n_indel_pos, Lref);
                c_snp = spread_merged(snp_pos, m.n_snp_pos, nullptr, 0, Lref);
                c_indel = spread_merged(indel_pos, m.n_indel_pos, nullptr, 0, Lref);
            }
            d_all = (int)c_all;
            d_snp = (int)c_snp;
            d_indel = (int)c_indel;
            best_snp = m.snp;
            best_indel = m.indel;
            best = (int)c;
        }
    }

    mark(5);
    // ---- publish ---------------------------------------------------------------------------------
    pf_bubble_result res;
    memset(&res, 0, sizeof(res));
    if (best < 0) {
        if (lane == 0) o.res[job] = res;
        return 0;
    }
    const MsaRef B = list[cur][best];
    const char *rows = arena[cur] + B.off;
    if (B.len <= p.work_bytes) colinfo = nw_base;
    const Metrics m = classify(rows, B.rows, B.len, colinfo, snp_pos, indel_pos, indel_len);
    // variant columns = labelled columns (every row gets a group >= 1 there)
    uint32_t n_sites = 0;
    for (uint32_t base = 0; base < B.len; base += WAVE) {
        const uint32_t j = base + lane;
        n_sites += (uint32_t)__popcll(__ballot(j < B.len && (colinfo[j] & 16)));
    }
    const unsigned long long t0 = bub_take(o, al, 0, (unsigned long long)B.rows * B.len, 4096);
    const unsigned long long s0 = bub_take(o, al, 1, n_sites, 64);
    const unsigned long long g0 = bub_take(o, al, 2, (unsigned long long)n_sites * B.rows, 256);
    const unsigned long long l0 = bub_take(o, al, 3, m.n_indel_len, 32);
    res.rows_off = t0;
    res.site_off = s0;
    res.group_off = g0;
    res.ilen_off = l0;
    res.n_rows = B.rows;
    res.n_cols = B.len;
    res.n_sites = n_sites;
    res.n_indel_len = m.n_indel_len;
    if (lane == 0) o.res[job] = res;
    if (t0 + (uint64_t)B.rows * B.len > o.text_cap || s0 + n_sites > o.site_cap || g0 + (uint64_t)n_sites * B.rows > o.group_cap ||
        l0 + m.n_indel_len > o.ilen_cap)
        return 0;  // the host sees the heads and asks again with larger pools
    for (uint32_t t = lane; t < B.rows * B.len; t += WAVE) o.text[t0 + t] = rows[t];
    for (uint32_t t = lane; t < m.n_indel_len; t += WAVE) o.ilen[l0 + t] = indel_len[t];
    uint32_t rank_base = 0;
    for (uint32_t base = 0; base < B.len; base += WAVE) {
        const uint32_t j = base + lane;
        const bool is_site = j < B.len && (colinfo[j] & 16);
        const unsigned long long mask = __ballot(is_site);
        if (is_site) {
            const uint32_t rk = rank_base + (uint32_t)__popcll(mask & ((1ull << lane) - 1));
            uint8_t *grp = o.groups + g0 + (uint64_t)rk * B.rows;
            uint8_t next = 0;
            for (uint32_t r = 0; r < B.rows; ++r) {
                const char c = rows[(size_t)r * B.len + j];
                uint32_t e = 0;
                while (e < r && rows[(size_t)e * B.len + j] != c) ++e;
                grp[r] = e < r ? grp[e] : ++next;
            }
            pf_bubble_site sr;
            sr.col = j;
            sr.is_indel = (colinfo[j] & 32) ? 1 : 0;
            sr.maxnum = next;
            sr.pad_ = 0;
            o.sites[s0 + rk] = sr;
        }
        rank_base += (uint32_t)__popcll(mask);
    }
    mark(6);
    return 0;
}

// Three wavefronts a SIMD: left to itself the compiler takes 212 registers (two wavefronts a SIMD, eight a CU -- a third of what
// the class grids ask for), and a kernel that spends its time on one lane's dependent steps gains more from a third wavefront
// than it loses to the 45 registers that go to scratch: K-BUBBLE's share of a pass 56 -> 45 ms at configs[4] (10 M unitigs,
// PloidyEstimation 93 -> 84 ms), 7.3 -> 6.7 ms at 1 M, 2.7 -> 2.5 ms at configs[2]; four (128 registers, 112 spilled) measures the same.
template <bool LDS>
__global__ __launch_bounds__(64, 3) void k_bubble(BubParams p, BubOut o) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    uint8_t *nw_base;
    if constexpr (LDS) nw_base = smem;
    else nw_base = p.work + (uint64_t)blockIdx.x * p.work_per_wave;
    uint8_t *scr = p.scratch + (uint64_t)blockIdx.x * p.scratch_per_wave;
    BubAlloc al;
    // Tasks are handed out through one counter per launch (the host lists the multi-path ones first): a bubble
    // costs anything between 10 us and 2 ms, so a static split leaves most waves idle behind the unlucky ones.
    for (;;) {
        uint32_t q = 0;
        if (lane_id() == 0) q = atomicAdd(p.next, 1u);
        q = read_lane(q, 0);
        if (q >= p.n) break;
        const uint32_t job = p.idx[q];
        const unsigned long long c0 = p.task_clk ? wall_clock64() : 0;
        const int rc = bubble_task<LDS>(p, o, al, job, nw_base, scr);
        if (p.task_clk && lane_id() == 0) p.task_clk[job] = wall_clock64() - c0;
        if (rc != 0 && lane_id() == 0) {
            if (p.final_tier) {
                pf_bubble_result r;
                memset(&r, 0, sizeof(r));
                r.n_rows = 0xFFFFFFFFu;  // error marker
                o.res[job] = r;
                if (p.bad) atomicAdd(p.bad, 1u);
            } else {
                const unsigned int x = atomicAdd(o.n_retry, 1u);
                o.retry[x] = job;
            }
        }
        aln_sync();
    }
}

}  // namespace

namespace pf {

// Launch part of K-BUBBLE shared by pf_align_bubbles (host-described batches) and the resident calling pipeline
// (pf_call.hip: paths, tasks and the per-class work queues are produced on the device).  d_idx holds the four class
// queues back to back (class c: n_cls[c] task indices, heavy ones first); results / pools are device memory.
// Leaves the pool heads in heads[4]; returns PF_ERR_OVERFLOW when a bubble exceeds the largest scratch tier
// (its result then carries n_rows = 0xFFFFFFFF).
unsigned long long *bubble_pool_heads(pf_ctx *ctx, int lane) { return (unsigned long long *)ctx_ws(ctx, bub_ws(WS_BUB_SMALL, lane), 128); }

// the class launches of a lane go on three streams ("trains", bubble_launch): the device runs three kernels of a process at a time
constexpr int kBubTrains = 3;
static int bubble_streams(pf_ctx *ctx, int lane) {
    if (ctx->bub_streams[lane][0]) return PF_OK;
    for (int c = 0; c < kBubTrains; ++c) {   // (a stream is milliseconds to create: no more of them than are used)
        PF_HIP(lane_stream_create(&ctx->bub_streams[lane][c], lane));
        PF_HIP(hipEventCreateWithFlags(&ctx->bub_events[lane][c], hipEventDisableTiming));
    }
    PF_HIP(hipEventCreateWithFlags(&ctx->bub_events[lane][kBubLdsClasses], hipEventDisableTiming));
    return PF_OK;
}
static int bubble_func_attr(pf_ctx *ctx) {   // (once per process, whichever lane comes first)
    static const hipError_t once = hipFuncSetAttribute(reinterpret_cast<const void *>(&k_bubble<true>), hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
    PF_HIP(once);
    return PF_OK;
}

// K-BUBBLE's workspaces for batches of up to n_tasks bubbles, ahead of the first launch (pf_call_reserve)
int bubble_reserve(pf_ctx *ctx, uint32_t n_tasks, int lane) {
    const BubCaps std_caps{64 * 1024, 8 * 1024, 512, 16 * 1024, 128 * 1024, 256, 8 * 1024};
    const uint64_t per = (bub_scratch_bytes(std_caps) + 255) & ~255ull;
    if (!ctx_ws(ctx, bub_ws(WS_BUB_SMALL, lane), 128) || !ctx_ws(ctx, bub_ws(WS_BUB_RETRY, lane), (size_t)std::max<uint32_t>(n_tasks, 1) * 4) ||
        !ctx_ws(ctx, bub_ws(WS_BUB_SCRATCH, lane), per * (uint64_t)ctx->n_cu * 24))
        return PF_ERR_HIP;
    DevLoadTrace trace;
    { const int e = bubble_streams(ctx, lane); if (e != PF_OK) return e; }
    trace.mark("bubble: class streams");
    { const int e = bubble_func_attr(ctx); if (e != PF_OK) return e; }
    trace.mark("bubble: function attributes");
    // One empty launch per class, over the train streams, now, beside the load: the first K-BUBBLE launch of a stream pays for the
    // queue's scratch (the kernel spills) and its first use of that much LDS -- 13 ms for the nine launches of a first
    // PloidyEstimation when they were paid there.  (n = 0: every wavefront leaves at its first look at the queue.)
    uint8_t *small = (uint8_t *)ctx_ws(ctx, bub_ws(WS_BUB_SMALL, lane), 128);
    PF_HIP(hipMemsetAsync(small, 0, 128, ctx->bub_streams[lane][0]));
    PF_HIP(hipStreamSynchronize(ctx->bub_streams[lane][0]));
    BubParams p;
    memset(&p, 0, sizeof(p));
    BubOut o;
    memset(&o, 0, sizeof(o));
    p.next = reinterpret_cast<unsigned int *>(small + 64);
    for (int c = 0; c < kBubLdsClasses; ++c) k_bubble<true><<<1, 64, kBubClassBytes[c], ctx->bub_streams[lane][c % kBubTrains]>>>(p, o);
    PF_HIP(hipGetLastError());
    for (int c = 0; c < kBubTrains; ++c) PF_HIP(hipStreamSynchronize(ctx->bub_streams[lane][c]));
    trace.mark("bubble: an empty launch per class stream");
    return PF_OK;
}

int bubble_launch(pf_ctx *ctx, const BubbleLaunch &L, unsigned long long heads[4]) {
    const int lane = L.lane;
    hipStream_t st = L.stream ? L.stream : ctx->stream;
    // launch timing by place (a call on another lane may time its launches at the same time)
    size_t tl_at = (size_t)-1;
    auto tbegin = [&](int kernel, hipStream_t s) { (void)ctx_begin_at(ctx, kernel, s, &tl_at); };
    auto tend = [&](hipStream_t s) { ctx_end_at(ctx, tl_at, s); };
    uint8_t *small = (uint8_t *)ctx_ws(ctx, bub_ws(WS_BUB_SMALL, lane), 128);  // pool heads, retry count, one queue head per launch
    uint32_t *d_retry = (uint32_t *)ctx_ws(ctx, bub_ws(WS_BUB_RETRY, lane), (size_t)std::max<uint32_t>(L.n_tasks, 1) * 4);
    if (!small || !d_retry) return PF_ERR_HIP;
    if (L.keep_heads) PF_HIP(hipMemsetAsync(small + 32, 0, 96, st));
    else PF_HIP(hipMemsetAsync(small, 0, 128, st));
    BubOut o;
    o.res = L.res; o.text = L.otext; o.sites = L.osites; o.groups = L.ogroups; o.ilen = L.oilen;
    o.text_cap = L.text_cap; o.site_cap = L.site_cap; o.group_cap = L.group_cap; o.ilen_cap = L.ilen_cap;
    o.heads = reinterpret_cast<unsigned long long *>(small);
    o.n_retry = reinterpret_cast<unsigned int *>(small + 40);
    unsigned int *queue_heads = reinterpret_cast<unsigned int *>(small + 64);
    int n_launch = 0;
    o.retry = d_retry;

    BubParams p;
    memset(&p, 0, sizeof(p));
    p.text = L.text; p.paths = L.paths; p.tasks = L.tasks; p.M = L.match; p.D = L.mismatch; p.G = L.gap;
    p.integral = (L.match == std::floor(L.match) && L.mismatch == std::floor(L.mismatch) && L.gap == std::floor(L.gap) &&
                  std::fabs(L.match) < 1e6 && std::fabs(L.mismatch) < 1e6 && std::fabs(L.gap) < 1e6) ? 1 : 0;
    p.seq = ctx->d_seq; p.off = ctx->d_off; p.len = ctx->d_len; p.n_unitigs = ctx->N;
    const BubCaps std_caps{64 * 1024, 8 * 1024, 512, 16 * 1024, 128 * 1024, 256, 8 * 1024};
    // waves in flight: the kernel is a chain of dependent LDS / global accesses per bubble, so it wants
    // every wave slot the LDS budget allows (5 KiB class: 32 per CU by LDS, capped at 24)
    // The size classes run side by side on streams of their own, each with its slice of the scratch: every launch lasts at least
    // as long as its slowest bubble (an 8-path bubble: 0.8 ms of dependent steps on one wavefront), and the three tails overlap.
    // Measured with the passes of both settings alternating in one process (tools/ab_pass.py, 5 M unitigs, two align ranges):
    // 28.1 -> 26.0 ms per pass.  PF_BUBBLE_STREAMS=0: one after the other.
    constexpr bool class_streams = true;   // a launch per size class, side by side on streams of their own
    // Every wavefront has its slice of the scratch (0.5 MB) and loops over its class's queue; bubble_reserve took scratch for 24
    // wavefronts a CU beside the load.  How the classes share them: the device runs the kernels of at most three or four streams of a
    // process at a time (its hardware queues; measured at configs[4]'s parameters: of seven class launches on seven streams three ran,
    // the fourth began when the first had ended -- with grids sized as if all seven ran together the device held ten wavefronts a CU
    // where it could hold twenty-four).  So the class launches go on THREE streams ("trains"), each class on the train that has the
    // least work so far (work = bubbles x the class's bytes: a bubble's time grows with its matrix -- 13 k ticks in the 6 KB class, 165 k
    // in the 64 KB one), heaviest class first; a launch gets a third of the wavefronts (never more than it has bubbles, or than its LDS
    // lets a CU hold) and the launches of a train, one after the other, use the same third of the scratch.
    // (measured, passes of 17.3 ms / 17.9 ms at configs[4]'s parameters, 1 M unitigs / configs[2]: one train 23.5 / 19.4, two 18.8 / 19.0,
    // four 22.3 / 17.9: profiles/r5_experiments.txt)
    constexpr int kTrains = kBubTrains;
    int grids[kBubLdsClasses];
    int train_of[kBubLdsClasses];
    int order[kBubLdsClasses];
    int n_order = 0;
    const uint64_t wave_budget = (uint64_t)ctx->n_cu * 24;
    const uint64_t train_waves = wave_budget / kTrains;
    {
        double load[kTrains] = {0, 0, 0};
        for (int c = 0; c < kBubLdsClasses; ++c) {
            grids[c] = 0;
            train_of[c] = 0;
            if (L.n_cls[c]) order[n_order++] = c;
        }
        auto work_of = [&](int c) { return (double)L.n_cls[c] * (double)kBubClassBytes[c]; };
        std::sort(order, order + n_order, [&](int x, int y) { return work_of(x) > work_of(y) || (work_of(x) == work_of(y) && x > y); });
        for (int i = 0; i < n_order; ++i) {
            const int c = order[i];
            int tr = 0;
            for (int q = 1; q < kTrains; ++q)
                if (load[q] < load[tr]) tr = q;
            train_of[c] = tr;
            load[tr] += work_of(c);
            grids[c] = (int)std::min<uint64_t>(std::min<uint64_t>(L.n_cls[c], (uint64_t)ctx->n_cu * bubble_class_waves_per_cu(c)), class_streams ? train_waves : wave_budget);
        }
    }
    const uint64_t max_waves = wave_budget;
    p.caps = std_caps;
    p.scratch_per_wave = (bub_scratch_bytes(std_caps) + 255) & ~255ull;
    const bool trace_ws = getenv("PF_TRACE_ALIGN") != nullptr;
    const auto t_ws = std::chrono::steady_clock::now();
    uint8_t *const scratch0 = (uint8_t *)ctx_ws(ctx, bub_ws(WS_BUB_SCRATCH, lane), p.scratch_per_wave * max_waves);
    if (trace_ws)
        fprintf(stderr, "[bubble_launch] scratch for %llu wavefronts x %llu bytes = %.1f MB taken in %.2f ms\n", (unsigned long long)max_waves, (unsigned long long)p.scratch_per_wave,
                (double)(p.scratch_per_wave * max_waves) / 1e6, std::chrono::duration<double>(std::chrono::steady_clock::now() - t_ws).count() * 1e3);
    p.scratch = scratch0;
    if (!p.scratch) return PF_ERR_HIP;
    if (class_streams) { const int e = bubble_streams(ctx, lane); if (e != PF_OK) return e; }

    DevTmp<unsigned long long> clk_;
    unsigned long long *d_clk = nullptr;
    const bool want_clk = getenv("PF_BUBBLE_STATS") != nullptr;
    if (want_clk) {
        PF_HIP(clk_.alloc(((size_t)L.n_tasks + 16) * 8));
        d_clk = clk_.p;
        PF_HIP(hipMemsetAsync(d_clk, 0, ((size_t)L.n_tasks + 16) * 8, st));
        p.task_clk = d_clk;
        p.prof = d_clk + L.n_tasks;
    }
    { const int e = bubble_func_attr(ctx); if (e != PF_OK) return e; }
    uint32_t idx_off = 0;
    for (int c = 0; c < kBubLdsClasses; ++c) idx_off += L.n_cls[c];   // (the global-memory class's queue lies behind the LDS classes')
    if (class_streams) PF_HIP(hipEventRecord(ctx->bub_events[lane][kBubLdsClasses], st));
    bool train_used[kTrains] = {false, false, false};
    for (int i = 0; i < n_order; ++i) {   // heaviest class first: its tail is the longest
        const int c = order[i];
        const uint32_t nc = L.n_cls[c];
        uint32_t first = 0;   // (the queues lie class after class in L.idx)
        for (int q = 0; q < c; ++q) first += L.n_cls[q];
        p.idx = L.idx + first;
        p.n = nc;
        p.work_bytes = (uint32_t)kBubClassBytes[c];
        const int grid = grids[c];
        p.next = queue_heads + n_launch++;
        hipStream_t cst = st;
        if (class_streams) {
            const int tr = train_of[c];
            cst = ctx->bub_streams[lane][tr];
            if (!train_used[tr]) PF_HIP(hipStreamWaitEvent(cst, ctx->bub_events[lane][kBubLdsClasses], 0));
            train_used[tr] = true;
            p.scratch = scratch0 + (uint64_t)tr * train_waves * p.scratch_per_wave;
        }
        tbegin(PF_K_BUBBLE, cst);
        k_bubble<true><<<grid, 64, kBubClassBytes[c], cst>>>(p, o);
        tend(cst);
        ctx_units(ctx, PF_K_BUBBLE, nc);
    }
    if (class_streams)
        for (int tr = 0; tr < kTrains; ++tr) {
            if (!train_used[tr]) continue;
            PF_HIP(hipEventRecord(ctx->bub_events[lane][tr], ctx->bub_streams[lane][tr]));
            PF_HIP(hipStreamWaitEvent(st, ctx->bub_events[lane][tr], 0));
        }
    p.scratch = scratch0;
    if (L.n_cls[kBubLdsClasses]) {
        const uint32_t nc = L.n_cls[kBubLdsClasses];
        const int grid = (int)std::min<uint32_t>(nc, 256);
        const uint64_t per = (std::min<uint64_t>(L.max_need * 2, 1ull << 31) + 255) & ~255ull;
        uint8_t *work = (uint8_t *)ctx_ws(ctx, bub_ws(WS_BUB_WORK, lane), per * grid);
        if (!work) return PF_ERR_HIP;
        p.idx = L.idx + idx_off;
        p.n = nc;
        p.work = work;
        p.work_per_wave = per;
        p.work_bytes = (uint32_t)std::min<uint64_t>(per, 0xFFFFFFFFu);
        p.next = queue_heads + n_launch++;
        tbegin(PF_K_BUBBLE_BIG, st);
        k_bubble<false><<<grid, 64, 0, st>>>(p, o);
        tend(st);
        ctx_units(ctx, PF_K_BUBBLE_BIG, nc);
    }
    unsigned int n_retry = 0;
    PF_HIP(hipMemcpyAsync(&n_retry, o.n_retry, 4, hipMemcpyDeviceToHost, st));
    PF_HIP(hipStreamSynchronize(st));
    int status = PF_OK;
    DevTmp<uint8_t> big_scratch_, big_work_;
    DevTmp<unsigned int> bad_;
    if (n_retry) {
        // some capacity of the standard tier was exceeded: rerun with 32x the scratch on a few waves.  The working storage a
        // retried bubble needs is bounded by the longest rows it can produce: the sum of its path lengths (the host knows the
        // exact figure when it described the batch; the resident caller passes the bound of its batch).
        const uint64_t need = std::max<uint64_t>(1 << 20, L.retry_need);
        const BubCaps big{2u << 20, 256 * 1024, 16 * 1024, 512 * 1024, 4u << 20, 8192, 256 * 1024};
        const int grid = (int)std::min<uint32_t>(n_retry, 8);
        const uint64_t per_s = (bub_scratch_bytes(big) + 255) & ~255ull;
        const uint64_t per_w = (std::min<uint64_t>(need, 0xFFFFFF00u) + 255) & ~255ull;
        PF_HIP(big_scratch_.alloc(per_s * grid));
        PF_HIP(big_work_.alloc(per_w * grid));
        PF_HIP(bad_.alloc(4));
        PF_HIP(hipMemsetAsync(bad_.p, 0, 4, st));
        // the retry list becomes the queue of the last launch (copied: the kernel appends to o.retry again on failure paths only
        // in non-final tiers)
        uint32_t *d_q = (uint32_t *)ctx_ws(ctx, bub_ws(WS_BUB_IDX2, lane), (size_t)n_retry * 4);
        if (!d_q) return PF_ERR_HIP;
        PF_HIP(hipMemcpyAsync(d_q, d_retry, (size_t)n_retry * 4, hipMemcpyDeviceToDevice, st));
        p.idx = d_q; p.n = n_retry; p.caps = big; p.scratch = big_scratch_.p; p.scratch_per_wave = per_s;
        p.work = big_work_.p; p.work_per_wave = per_w; p.work_bytes = (uint32_t)per_w; p.final_tier = 1;
        p.bad = bad_.p;
        p.next = queue_heads + n_launch++;
        tbegin(PF_K_BUBBLE_BIG, st);
        k_bubble<false><<<grid, 64, 0, st>>>(p, o);
        tend(st);
        unsigned int bad = 0;
        PF_HIP(hipMemcpyAsync(&bad, bad_.p, 4, hipMemcpyDeviceToHost, st));
        PF_HIP(hipStreamSynchronize(st));
        if (bad) {
            pf::CtxErr{ctx} = "pf_align_bubbles: a bubble exceeds the largest scratch tier";
            status = PF_ERR_OVERFLOW;
        }
    }
    if (want_clk) {
        const uint32_t n_tasks = L.n_tasks;
        std::vector<unsigned long long> clk(n_tasks + 16);
        PF_HIP(hipMemcpy(clk.data(), d_clk, ((size_t)n_tasks + 16) * 8, hipMemcpyDeviceToHost));
        fprintf(stderr, "[pf_align_bubbles] traceback: %llu jobs on the one-path fast road, %llu through the depth-first walk: %llu steps, %llu complete paths (%llu ticks in their analysis and copy), %llu kept\n",
                clk[n_tasks + 13], clk[n_tasks + 12], clk[n_tasks + 8], clk[n_tasks + 9], clk[n_tasks + 11], clk[n_tasks + 10]);
        fprintf(stderr, "[pf_align_bubbles] ticks by phase: fill %llu traceback %llu | decode %llu round0(incl. NW) %llu rounds %llu choose %llu publish %llu\n",
                clk[n_tasks], clk[n_tasks + 1], clk[n_tasks + 2], clk[n_tasks + 3], clk[n_tasks + 4], clk[n_tasks + 5], clk[n_tasks + 6]);
        clk.resize(n_tasks);
        {   // per size class: bubbles, wavefronts of its launch, ticks, and the longest bubble -- ticks / wavefronts is what the launch takes at best
            uint32_t tot_q = 0;
            for (int c = 0; c <= kBubLdsClasses; ++c) tot_q += L.n_cls[c];
            std::vector<uint32_t> hq(tot_q);
            if (tot_q) PF_HIP(hipMemcpy(hq.data(), L.idx, (size_t)tot_q * 4, hipMemcpyDeviceToHost));
            uint32_t at = 0;
            for (int c = 0; c <= kBubLdsClasses; ++c) {
                unsigned long long sum = 0, mx = 0;
                for (uint32_t q = 0; q < L.n_cls[c]; ++q) { const unsigned long long t = hq[at + q] < n_tasks ? clk[hq[at + q]] : 0; sum += t; mx = std::max(mx, t); }
                at += L.n_cls[c];
                if (L.n_cls[c])
                    fprintf(stderr, "   class %d (%s): %u bubbles on %d wavefronts, %llu ticks (%.2f ms if spread evenly), longest %llu ticks\n", c,
                            c == kBubLdsClasses ? "global" : "LDS", L.n_cls[c], c < kBubLdsClasses ? grids[c] : 0, sum,
                            c < kBubLdsClasses && grids[c] ? (double)sum / grids[c] * 1e-5 : 0.0, mx);
            }
        }
        std::vector<unsigned long long> srt(clk);
        std::sort(srt.begin(), srt.end());
        unsigned long long tot = 0;
        for (auto c : srt) tot += c;
        fprintf(stderr, "[pf_align_bubbles] %u tasks: ticks(10ns) sum %llu  median %llu  p90 %llu  p99 %llu  p99.9 %llu  max %llu\n", n_tasks, tot,
                srt[n_tasks / 2], srt[(size_t)(n_tasks * 0.9)], srt[(size_t)(n_tasks * 0.99)], srt[(size_t)(n_tasks * 0.999)], srt[n_tasks - 1]);
        {   // where the time goes by kind of bubble: paths (2, 3, 4, 5+) x (strict: paths are unitigs / branching: text paths)
            std::vector<pf_bubble_task> ht(n_tasks);
            PF_HIP(hipMemcpy(ht.data(), L.tasks, (size_t)n_tasks * sizeof(pf_bubble_task), hipMemcpyDeviceToHost));
            unsigned long long tk[2][4] = {{0, 0, 0, 0}, {0, 0, 0, 0}}, nk[2][4] = {{0, 0, 0, 0}, {0, 0, 0, 0}};
            for (uint32_t t = 0; t < n_tasks; ++t) {
                if (!clk[t] || ht[t].n_paths < 2) continue;
                pf_bubble_path p0;
                if (hipMemcpy(&p0, L.paths + ht[t].path_first, sizeof(p0), hipMemcpyDeviceToHost) != hipSuccess) break;
                const int a = p0.ov == PF_NONE ? 1 : 0, b = ht[t].n_paths >= 5 ? 3 : (int)ht[t].n_paths - 2;
                tk[a][b] += clk[t];
                nk[a][b]++;
                if (t > 60000) break;   // (one small copy per task: a sample is enough)
            }
            for (int a = 0; a < 2; ++a)
                fprintf(stderr, "   %s: 2 paths %llu tasks %llu ticks | 3: %llu / %llu | 4: %llu / %llu | 5+: %llu / %llu\n", a ? "branching" : "strict   ",
                        nk[a][0], tk[a][0], nk[a][1], tk[a][1], nk[a][2], tk[a][2], nk[a][3], tk[a][3]);
        }
        unsigned long long over[4] = {0, 0, 0, 0};
        for (auto c : srt) { over[0] += c > 10000; over[1] += c > 30000; over[2] += c > 100000; if (c > 30000) over[3] += c; }
        fprintf(stderr, "   tasks over 0.1 ms: %llu, over 0.3 ms: %llu (their sum %llu ticks), over 1 ms: %llu\n", over[0], over[1], over[3], over[2]);
        for (int top = 0; top < 8 && top < (int)n_tasks; ++top) {
            const uint32_t t = (uint32_t)(std::max_element(clk.begin(), clk.end()) - clk.begin());
            pf_bubble_task tk;
            PF_HIP(hipMemcpy(&tk, L.tasks + t, sizeof(tk), hipMemcpyDeviceToHost));
            std::vector<pf_bubble_path> pp(std::min<uint32_t>(tk.n_paths, 12));
            if (!pp.empty()) PF_HIP(hipMemcpy(pp.data(), L.paths + tk.path_first, pp.size() * sizeof(pf_bubble_path), hipMemcpyDeviceToHost));
            pf_bubble_result rr;
            PF_HIP(hipMemcpy(&rr, L.res + t, sizeof(rr), hipMemcpyDeviceToHost));
            fprintf(stderr, "   slowest: task %u ticks %llu paths %u -> rows %u cols %u sites %u; lens", t, clk[t], tk.n_paths, rr.n_rows, rr.n_cols, rr.n_sites);
            for (auto &x : pp) fprintf(stderr, " %u", x.len);
            fprintf(stderr, "\n");
            clk[t] = 0;
        }
    }
    PF_HIP(hipMemcpy(heads, o.heads, 32, hipMemcpyDeviceToHost));
    if (status == PF_OK && (heads[0] > L.text_cap || heads[1] > L.site_cap || heads[2] > L.group_cap || heads[3] > L.ilen_cap)) {
        pf::CtxErr{ctx} = "pf_align_bubbles: output pools too small";
        status = PF_ERR_OVERFLOW;
    }
    return status;
}

}  // namespace pf

extern "C" int pf_align_bubbles(pf_ctx *ctx, const char *text, uint64_t text_len, const pf_bubble_path *paths, uint64_t n_paths,
                                const pf_bubble_task *tasks, uint32_t n_tasks, double match, double mismatch, double gap,
                                pf_bubble_result *results, char *out_text, uint64_t text_cap, pf_bubble_site *out_sites,
                                uint64_t site_cap, uint8_t *out_groups, uint64_t group_cap, uint32_t *out_ilen,
                                uint64_t ilen_cap, uint64_t used[4]) {
    if (!ctx || !used || (n_tasks && (!paths || !tasks || !results || !out_text || !out_sites || !out_groups || !out_ilen)))
        return PF_ERR_ARG;
    used[0] = used[1] = used[2] = used[3] = 0;
    if (n_tasks == 0) return PF_OK;
    if (!ctx->d_seq) { pf::CtxErr{ctx} = "pf_align_bubbles: no graph uploaded"; return PF_ERR_ARG; }
    PF_HIP(hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    auto is_dev = [](const void *p) {
        hipPointerAttribute_t at;
        bool d = p && hipPointerGetAttributes(&at, p) == hipSuccess && at.type == hipMemoryTypeDevice;
        (void)hipGetLastError();
        return d;
    };
    // host view of the arguments for validation and size classes (copied only when they live on the device)
    std::vector<pf_bubble_task> ht_own;
    std::vector<pf_bubble_path> hp_own;
    const pf_bubble_task *ht = tasks;
    const pf_bubble_path *hp = paths;
    if (is_dev(tasks)) {
        ht_own.resize(n_tasks);
        PF_HIP(hipMemcpy(ht_own.data(), tasks, (size_t)n_tasks * sizeof(pf_bubble_task), hipMemcpyDeviceToHost));
        ht = ht_own.data();
    }
    if (n_paths && is_dev(paths)) {
        hp_own.resize(n_paths);
        PF_HIP(hipMemcpy(hp_own.data(), paths, (size_t)n_paths * sizeof(pf_bubble_path), hipMemcpyDeviceToHost));
        hp = hp_own.data();
    }
    const std::vector<uint32_t> &hlen = ctx->h_len;  // unitig lengths for ov paths
    std::vector<uint32_t> cls[kBubLdsClasses + 1];
    std::vector<char> heavy;
    heavy.reserve(n_tasks);
    uint64_t max_need = 0, retry_need = 0;
    for (uint32_t t = 0; t < n_tasks; ++t) {
        const pf_bubble_task &tk = ht[t];
        if (tk.n_paths < 2 || tk.n_paths > 65535 || tk.path_first + tk.n_paths > n_paths) {
            pf::CtxErr{ctx} = "pf_align_bubbles: a bubble needs 2..65535 paths inside the path array";
            return PF_ERR_ARG;
        }
        uint32_t l0 = 0, lmax = 0;
        uint64_t sum = 0;
        for (uint32_t i = 0; i < tk.n_paths; ++i) {
            const pf_bubble_path &pp = hp[tk.path_first + i];
            if (pp.ov != PF_NONE) {
                if ((pp.ov >> 1) >= ctx->N) { pf::CtxErr{ctx} = "pf_align_bubbles: path unitig out of range"; return PF_ERR_ARG; }
                if (pp.len != hlen[pp.ov >> 1]) { pf::CtxErr{ctx} = "pf_align_bubbles: path length differs from the unitig's"; return PF_ERR_ARG; }
            } else if (pp.text_off + pp.len > text_len) {
                pf::CtxErr{ctx} = "pf_align_bubbles: path outside the text buffer";
                return PF_ERR_ARG;
            }
            if (pp.len == 0 || pp.len > 60000) { pf::CtxErr{ctx} = "pf_align_bubbles: empty path or longer than 60000"; return PF_ERR_ARG; }
            if (i == 0) l0 = pp.len;
            lmax = std::max(lmax, pp.len);
            sum += pp.len;
        }
        const int c = bubble_class(l0, lmax);
        cls[c].push_back(t);
        heavy.push_back(tk.n_paths > 2 || lmax > 64);
        if (c == kBubLdsClasses) max_need = std::max(max_need, bubble_need(l0, lmax));
        retry_need = std::max(retry_need, job_bytes((uint32_t)std::min<uint64_t>(sum, 60000), lmax));
    }
    for (auto &v : cls)  // multi-path / long bubbles first (stable: the order inside each half is kept)
        std::stable_partition(v.begin(), v.end(), [&](uint32_t t) { return heavy[t]; });
    // device inputs
    const char *d_text = text;
    if (text_len && !is_dev(text)) {
        char *b = (char *)ctx_ws(ctx, WS_BUB_TEXT, text_len + 1);
        if (!b) return PF_ERR_HIP;
        PF_HIP(hipMemcpyAsync(b, text, text_len, hipMemcpyHostToDevice, st));
        d_text = b;
    }
    pf_bubble_path *d_paths = (pf_bubble_path *)ctx_ws(ctx, WS_BUB_PATHS, std::max<uint64_t>(n_paths, 1) * sizeof(pf_bubble_path));
    pf_bubble_task *d_tasks = (pf_bubble_task *)ctx_ws(ctx, WS_BUB_TASKS, (size_t)n_tasks * sizeof(pf_bubble_task));
    uint32_t *d_idx = (uint32_t *)ctx_ws(ctx, WS_BUB_IDX, (size_t)n_tasks * 4);
    if (!d_paths || !d_tasks || !d_idx) return PF_ERR_HIP;
    PF_HIP(hipMemcpyAsync(d_paths, hp, (size_t)n_paths * sizeof(pf_bubble_path), hipMemcpyHostToDevice, st));
    PF_HIP(hipMemcpyAsync(d_tasks, ht, (size_t)n_tasks * sizeof(pf_bubble_task), hipMemcpyHostToDevice, st));
    const bool dev_out = is_dev(results);
    BubbleLaunch L;
    if (dev_out) {
        L.res = results; L.otext = out_text; L.osites = out_sites; L.ogroups = out_groups; L.oilen = out_ilen;
    } else {
        L.res = (pf_bubble_result *)ctx_ws(ctx, WS_BUB_RES, (size_t)n_tasks * sizeof(pf_bubble_result));
        L.otext = (char *)ctx_ws(ctx, WS_BUB_OTEXT, std::max<uint64_t>(text_cap, 1));
        L.osites = (pf_bubble_site *)ctx_ws(ctx, WS_BUB_OSITES, std::max<uint64_t>(site_cap, 1) * sizeof(pf_bubble_site));
        L.ogroups = (uint8_t *)ctx_ws(ctx, WS_BUB_OGROUPS, std::max<uint64_t>(group_cap, 1));
        L.oilen = (uint32_t *)ctx_ws(ctx, WS_BUB_OILEN, std::max<uint64_t>(ilen_cap, 1) * 4);
        if (!L.res || !L.otext || !L.osites || !L.ogroups || !L.oilen) return PF_ERR_HIP;
    }
    L.text_cap = text_cap; L.site_cap = site_cap; L.group_cap = group_cap; L.ilen_cap = ilen_cap;
    L.text = d_text; L.paths = d_paths; L.tasks = d_tasks; L.n_tasks = n_tasks;
    L.match = match; L.mismatch = mismatch; L.gap = gap;
    uint32_t idx_off = 0;
    for (int c = 0; c <= kBubLdsClasses; ++c) {
        L.n_cls[c] = (uint32_t)cls[c].size();
        if (cls[c].empty()) continue;
        PF_HIP(hipMemcpyAsync(d_idx + idx_off, cls[c].data(), cls[c].size() * 4, hipMemcpyHostToDevice, st));
        idx_off += (uint32_t)cls[c].size();
    }
    L.idx = d_idx;
    L.max_need = max_need;
    L.retry_need = retry_need;
    unsigned long long heads[4] = {0, 0, 0, 0};
    int status = bubble_launch(ctx, L, heads);
    for (int x = 0; x < 4; ++x) used[x] = heads[x];
    if (!dev_out && status == PF_OK) {
        PF_HIP(hipMemcpyAsync(results, L.res, (size_t)n_tasks * sizeof(pf_bubble_result), hipMemcpyDeviceToHost, st));
        PF_HIP(hipMemcpyAsync(out_text, L.otext, (size_t)heads[0], hipMemcpyDeviceToHost, st));
        PF_HIP(hipMemcpyAsync(out_sites, L.osites, (size_t)heads[1] * sizeof(pf_bubble_site), hipMemcpyDeviceToHost, st));
        PF_HIP(hipMemcpyAsync(out_groups, L.ogroups, (size_t)heads[2], hipMemcpyDeviceToHost, st));
        PF_HIP(hipMemcpyAsync(out_ilen, L.oilen, (size_t)heads[3] * 4, hipMemcpyDeviceToHost, st));
        PF_HIP(hipStreamSynchronize(st));
    }
    return status;
}
