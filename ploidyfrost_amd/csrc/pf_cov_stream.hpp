// Streaming per-unitig coverage reduction over the per-k-mer coverage SoA (K-COV, K-COV-C): shared by pf_device.hip and
// pf_colored.hip.  Reference functions: CDBG::readCov(const UnitigMap&) (src/CDBG.cpp:66-120), CCDBG::readCovUni
// (src/CCDBG.cpp:123-156).
// Inputs (pf_ctx.hpp): gcov[g] = count of graph k-mer g in graph order (GCOV_MISSING = absent), written by the load-time join;
// khead = one bit per k-mer, set where a unitig begins (and after the last k-mer); krow[r] = the unitig of k-mer 64 r.
// The scans use DPP row shifts / broadcasts: they are called from wave-uniform control flow only (all 64 lanes active), the
// per-lane conditions are applied to the moved values afterwards.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>

#include "pf_device_common.hpp"

namespace pf {

// one DPP move: lanes the control leaves without a source (or outside row_mask) keep `old`
template <int CTRL, int ROW_MASK>
__device__ inline uint32_t dpp_u32(uint32_t old, uint32_t v) {
    return (uint32_t)__builtin_amdgcn_update_dpp((int)old, (int)v, CTRL, ROW_MASK, 0xF, false);
}
template <int CTRL, int ROW_MASK>
__device__ inline unsigned long long dpp_u64(unsigned long long old, unsigned long long v) {
    const uint32_t lo = dpp_u32<CTRL, ROW_MASK>((uint32_t)old, (uint32_t)v), hi = dpp_u32<CTRL, ROW_MASK>((uint32_t)(old >> 32), (uint32_t)(v >> 32));
    return ((unsigned long long)hi << 32) | lo;
}
template <int CTRL, int ROW_MASK>
__device__ inline uint32_t dpp_any(uint32_t old, uint32_t v) { return dpp_u32<CTRL, ROW_MASK>(old, v); }
template <int CTRL, int ROW_MASK>
__device__ inline unsigned long long dpp_any(unsigned long long old, unsigned long long v) { return dpp_u64<CTRL, ROW_MASK>(old, v); }

// K-COV, streaming form with four k-mers per lane (the default): a lane takes 16 contiguous bytes of gcov (one dwordx4
// load), reduces its four k-mers serially -- `pre` = the k-mers before the lane's first unitig start, `suf` = those from its
// last unitig start on, both = all four when no unitig starts in the lane -- and the wavefront scans 64 lane aggregates
// instead of 64 k-mers: a quarter of the scan work per k-mer, which is what bounded the one-k-mer-per-lane form (VALU issue,
// not bytes).  A window is KC4_SR super-rows of 256 k-mers handled in order with a wave-uniform carry, so a unitig whose
// first and last k-mer lie in one window (2048 k-mers) is reduced completely in registers and written with plain stores;
// only what crosses a window border goes through atomics (about two per window instead of two per row and unitig).
//   lane with a unitig start at its k-mer i0: closes the unitig running up to i0 - 1 (value = scan of the lane before, or
//   the carry for lane 0, + pre), plain store if that unitig began inside the window, atomics otherwise;
//   further starts inside the same lane delimit unitigs of <= 3 k-mers, stored directly;
//   the open unitig at the end of the window is flushed with atomics.
constexpr int KC4_SR = 8;   // measured at 1 M unitigs: 4 -> 0.093 ms, 8 -> 0.072 ms, 16 -> 0.081 ms (profiles/history/r01j_kcov_stream.txt)

// Segmented inclusive scan (sum, min[, max]) over the 64 lanes; segments begin at the lanes set in `heads`.  `d` = distance
// from the lane to the last segment start at or below it (>= 64: none) turns every "same segment?" test into a 32-bit compare.
__device__ inline int seg_distance(uint64_t heads, int lane) {
    const uint64_t le_mask = lane == 63 ? ~0ull : ((2ull << lane) - 1);
    const uint64_t m = heads & le_mask;
    return m ? lane - (63 - __clzll((long long)m)) : 127;
}

template <typename S, bool MAXV>
__device__ inline void seg_scan_dpp(S &sum, uint32_t &mn, uint32_t &mx, int d, int lane) {
    const int li = lane & 15;
#define PF_SCAN_STEP(CTRL, MASK, COND)                                                                  \
    {                                                                                                   \
        const S so = dpp_any<CTRL, MASK>((S)0, sum);                                                    \
        const uint32_t mo = dpp_any<CTRL, MASK>(0xFFFFFFFFu, mn);                                       \
        const uint32_t xo = MAXV ? dpp_any<CTRL, MASK>(0u, mx) : 0u;                                    \
        if (COND) {                                                                                     \
            sum += so;                                                                                  \
            mn = mo < mn ? mo : mn;                                                                     \
            if (MAXV) mx = xo > mx ? xo : mx;                                                           \
        }                                                                                               \
    }
    // row_shr:O inside rows of 16 lanes: no segment starts at lanes lane - O + 1 .. lane
    PF_SCAN_STEP(0x111, 0xF, li >= 1 && d >= 1)
    PF_SCAN_STEP(0x112, 0xF, li >= 2 && d >= 2)
    PF_SCAN_STEP(0x114, 0xF, li >= 4 && d >= 4)
    PF_SCAN_STEP(0x118, 0xF, li >= 8 && d >= 8)
    PF_SCAN_STEP(0x142, 0xA, (lane & 16) && d > li)          // row_bcast:15 -> rows 1, 3: nothing starts from the row's first lane to this one
    PF_SCAN_STEP(0x143, 0xC, lane >= 32 && d > lane - 32)    // row_bcast:31 -> rows 2, 3
#undef PF_SCAN_STEP
}

// The same scan for narrow counts (every value below 2^26, sums below 2^32) without a single per-step condition: the sum is
// an ordinary inclusive prefix sum minus the prefix at the lane before the segment's first lane (one gather); min and max
// are ordinary scans of keys that carry the segment's first lane S above the value -- (63 - S) << 26 | v for the min, so
// that lanes of earlier segments can never win, S << 26 | v for the max.  6 fused DPP operations per quantity instead of
// 6 x (two moves, a test, two selects, the operation).
template <int OP>  // 0 add, 1 min, 2 max
__device__ inline uint32_t scan_u32_dpp(uint32_t x) {
    const uint32_t ident = OP == 1 ? 0xFFFFFFFFu : 0u;
#define PF_USCAN_STEP(CTRL, MASK)                                            \
    {                                                                        \
        const uint32_t y = dpp_u32<CTRL, MASK>(ident, x);                    \
        x = OP == 0 ? x + y : OP == 1 ? (y < x ? y : x) : (y > x ? y : x);   \
    }
    PF_USCAN_STEP(0x111, 0xF)
    PF_USCAN_STEP(0x112, 0xF)
    PF_USCAN_STEP(0x114, 0xF)
    PF_USCAN_STEP(0x118, 0xF)
    PF_USCAN_STEP(0x142, 0xA)
    PF_USCAN_STEP(0x143, 0xC)
#undef PF_USCAN_STEP
    return x;
}

template <bool MAXV>
__device__ inline void seg_scan_narrow(uint32_t &sum, uint32_t &mn, uint32_t &mx, int d, int lane) {
    constexpr uint32_t K26 = (1u << 26) - 1;
    const int S = d < 64 ? lane - d : 0;  // first lane of this lane's segment (0: the segment carried in)
    const uint32_t P = scan_u32_dpp<0>(sum);
    const uint32_t before = __shfl(P, S > 0 ? S - 1 : 0, WAVE);
    sum = P - (S > 0 ? before : 0u);
    uint32_t key = ((uint32_t)(63 - S) << 26) | (mn > K26 ? K26 : mn);
    key = scan_u32_dpp<1>(key) & K26;
    mn = key == K26 ? 0xFFFFFFFFu : key;
    if (MAXV) mx = scan_u32_dpp<2>(((uint32_t)S << 26) | mx) & K26;
}

// one finished (complete) or partial piece of a unitig.  CAP10K: C1's "min initialised 10000" (src/CDBG.cpp:71); the colored
// twin has no such cap and keeps a max (src/CCDBG.cpp:123-156)
template <typename S, bool COLORED>
__device__ inline void kc4_emit(uint32_t o, uint32_t n_out, S sum, uint32_t mn, uint32_t mx, bool complete, uint64_t *__restrict__ out_sum,
                                uint32_t *__restrict__ out_min, uint32_t *__restrict__ out_max) {
    if (o >= n_out) return;
    if (complete) {
        out_sum[o] = (uint64_t)sum;
        out_min[o] = COLORED ? mn : (mn < 10000u ? mn : 10000u);
        if (COLORED) out_max[o] = mx;
    } else {
        atomicAdd(reinterpret_cast<unsigned long long *>(out_sum + o), (unsigned long long)sum);
        if (mn != 0xFFFFFFFFu) atomicMin(out_min + o, mn);
        if (COLORED && mx) atomicMax(out_max + o, mx);
    }
}

template <typename S>
__device__ inline S readlane63(S v);
template <>
__device__ inline uint32_t readlane63<uint32_t>(uint32_t v) { return (uint32_t)__builtin_amdgcn_readlane((int)v, 63); }
template <>
__device__ inline unsigned long long readlane63<unsigned long long>(unsigned long long v) {
    return ((unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(v >> 32), 63) << 32) |
           (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)v, 63);
}

// what a launch of the streaming kernels works on
struct Kc4Args {
    const uint32_t *gcov;
    uint64_t g_stride;              // colored: slice stride of gcov
    const uint8_t *unread;          // colored: one byte per colour, 1 = never looked up
    const uint64_t *khead;
    const uint32_t *krow;
    uint32_t u0, n_out;
    uint64_t g_begin, g_end, sr_begin, sr_end;
    uint64_t *out_sum;
    uint32_t *out_min, *out_max;
    uint8_t *out_miss;
};

// one window: NSR super-rows starting at sr0, their counts already requested into c
template <bool WIDE, bool COLORED, int NSR>
__device__ inline void kc4_window(const uint4 (&c)[NSR], uint64_t sr0, int lane, const uint64_t *__restrict__ khead,
                                  const uint32_t *__restrict__ krow, uint32_t u0, uint32_t n_out, uint64_t g_begin, uint64_t g_end,
                                  uint64_t sr_end, uint64_t *__restrict__ out_sum, uint32_t *__restrict__ out_min,
                                  uint32_t *__restrict__ out_max, uint8_t *__restrict__ out_miss) {
    typedef typename std::conditional<WIDE, unsigned long long, uint32_t>::type sum_t;
    const int w = lane >> 4, sh = (lane & 15) * 4;
    const uint64_t lt_mask = (1ull << lane) - 1;                         // lanes below
    {
        sum_t csum = 0;           // carry: the unitig open at the end of the super-row before (wave-uniform)
        uint32_t cmin = 0xFFFFFFFFu, cmax = 0;
        bool cstarted = false;    // ... began inside this window
        uint32_t ulast = 0;
#pragma unroll
        for (int j = 0; j < NSR; ++j) {
            const uint64_t sr = sr0 + j;
            if (sr >= sr_end) break;
            const uint64_t H0 = khead[sr * 4], H1 = khead[sr * 4 + 1], H2 = khead[sr * 4 + 2], H3 = khead[sr * 4 + 3];
            const uint32_t kb = krow[sr * 4];
            const uint64_t g0 = sr * 256;
            const bool edge = g0 < g_begin || g0 + 256 > g_end;
            const uint64_t word = w == 0 ? H0 : w == 1 ? H1 : w == 2 ? H2 : H3;
            const uint32_t h = (uint32_t)(word >> sh) & 0xFu;
            const uint32_t below = (w > 0 ? __popcll(H0) : 0) + (w > 1 ? __popcll(H1) : 0) + (w > 2 ? __popcll(H2) : 0) +
                                   __popcll(word & ((2ull << sh) - 1));
            const uint32_t ub = kb + below - (uint32_t)(H0 & 1);  // unitig of the lane's first k-mer
            uint32_t v[4] = {c[j].x, c[j].y, c[j].z, c[j].w};
            if (edge) {  // k-mers outside [g_begin, g_end) (other unitigs, or the padding past the last k-mer) count as absent
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const uint64_t g = g0 + 4 * lane + i;
                    v[i] = (g >= g_begin && g < g_end) ? v[i] : GCOV_MISSING;
                }
            }
            uint32_t s[4];  // the count, 0 when absent (sum and max); v itself serves the min (absent = all ones)
            bool anymiss = false;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const bool x = v[i] == GCOV_MISSING;
                anymiss |= x;
                s[i] = x ? 0u : v[i];
            }
            if (anymiss) {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const uint32_t o = ub + __popc((h >> 1) & ((1u << i) - 1)) - u0;
                    if (v[i] == GCOV_MISSING && o < n_out) out_miss[o] = 1;
                }
            }
            // pre: k-mers with no unitig start at or before them; suf: k-mers with no unitig start after them
            sum_t ps = 0, ss = 0;
            uint32_t pm = 0xFFFFFFFFu, sm = 0xFFFFFFFFu, px = 0, sx = 0;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                if ((h & ((2u << i) - 1)) == 0) {
                    ps += s[i];
                    pm = v[i] < pm ? v[i] : pm;
                    if (COLORED) px = s[i] > px ? s[i] : px;
                }
                if ((h >> (i + 1)) == 0) {
                    ss += s[i];
                    sm = v[i] < sm ? v[i] : sm;
                    if (COLORED) sx = s[i] > sx ? s[i] : sx;
                }
            }
            if (h & (h - 1)) {  // two or more starts in the lane: the unitigs between them are complete here
#pragma unroll
                for (int i = 0; i < 3; ++i) {
                    if (((h >> i) & 1) && (h >> (i + 1))) {
                        sum_t es = 0;
                        uint32_t em = 0xFFFFFFFFu, ex = 0;
                        bool in = true;
#pragma unroll
                        for (int q = i; q < 4; ++q) {
                            if (q > i && ((h >> q) & 1)) in = false;
                            if (in) {
                                es += s[q];
                                em = v[q] < em ? v[q] : em;
                                ex = s[q] > ex ? s[q] : ex;
                            }
                        }
                        kc4_emit<sum_t, COLORED>(ub + __popc((h >> 1) & ((1u << i) - 1)) - u0, n_out, es, em, ex, true, out_sum, out_min, out_max);
                    }
                }
            }
            const uint64_t F = __ballot(h != 0);
            sum_t vs = ss;
            uint32_t vm = sm, vx = sx;
            const int d = seg_distance(F, lane);
            if constexpr (WIDE) seg_scan_dpp<sum_t, COLORED>(vs, vm, vx, d, lane);
            else seg_scan_narrow<COLORED>(vs, vm, vx, d, lane);   // counts < 2^20 (the launch condition): keys and sums fit
            if (d > lane) {  // no start at or below this lane: still inside the unitig carried in
                vs += csum;
                vm = cmin < vm ? cmin : vm;
                if (COLORED) vx = cmax > vx ? cmax : vx;
            }
            sum_t prev_s = __shfl_up(vs, 1, WAVE);
            uint32_t prev_m = __shfl_up(vm, 1, WAVE);
            uint32_t prev_x = COLORED ? __shfl_up(vx, 1, WAVE) : 0u;
            if (lane == 0) { prev_s = csum; prev_m = cmin; prev_x = cmax; }
            if (h != 0 && !(j == 0 && lane == 0 && (h & 1))) {  // (a start on the window's first k-mer closes nothing of this window)
                const sum_t es = prev_s + ps;
                const uint32_t em = prev_m < pm ? prev_m : pm;
                const uint32_t ex = prev_x > px ? prev_x : px;
                const bool started = (F & lt_mask) != 0 || cstarted;
                kc4_emit<sum_t, COLORED>(ub - (h & 1u) - u0, n_out, es, em, ex, started, out_sum, out_min, out_max);
            }
            csum = readlane63<sum_t>(vs);
            cmin = readlane63<uint32_t>(vm);
            if (COLORED) cmax = readlane63<uint32_t>(vx);
            cstarted = cstarted || F != 0;
            ulast = readlane63<uint32_t>(ub + __popc(h >> 1));
        }
        if (lane == 0) kc4_emit<sum_t, COLORED>(ulast - u0, n_out, csum, cmin, cmax, false, out_sum, out_min, out_max);
    }
}

template <int NSR>
__device__ inline void kc4_load(uint4 (&c)[NSR], const uint32_t *__restrict__ gcov, uint64_t sr0, uint64_t sr_end, int lane) {
#pragma unroll
    for (int j = 0; j < NSR; ++j)
        c[j] = sr0 + j < sr_end ? *reinterpret_cast<const uint4 *>(gcov + (sr0 + j) * 256 + 4 * lane) : make_uint4(0, 0, 0, 0);
}

// COLORED: blockIdx.y = colour; its counts at gcov + colour * g_stride, its results at out_* + colour * n_out
// (colour-major, the layout of pf_unitig_cov_colored); colours in `unread` are left as the init kernel wrote them.
// One wavefront per window of KC4_SR super-rows, uncapped grid.  Measured alternatives at 1 M unitigs (0.072 ms for this form,
// profiles/history/r01o_kcov_forms.txt): a persistent grid of resident wavefronts with round-robin windows of 4 super-rows and the
// next window's loads in flight during the reduction: 0.106-0.115 ms (4-7 blocks per CU); the same with windows handed out
// through one atomic counter: 0.59 ms (45 000 returning atomics on one address).
template <bool WIDE, bool COLORED>
__global__ __launch_bounds__(256) void k_cov_stream4(Kc4Args a) {
    if (COLORED) {
        const uint32_t colour = blockIdx.y;
        if (a.unread[colour]) return;
        a.gcov += (uint64_t)colour * a.g_stride;
        a.out_sum += (uint64_t)colour * a.n_out;
        a.out_min += (uint64_t)colour * a.n_out;
        a.out_max += (uint64_t)colour * a.n_out;
        a.out_miss += (uint64_t)colour * a.n_out;
    }
    const int lane = lane_id();
    const uint32_t wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);  // wave-uniform: row metadata comes through scalar loads
    const uint64_t wave = (uint64_t)blockIdx.x * (blockDim.x >> 6) + wv;
    const uint64_t n_waves = (uint64_t)gridDim.x * (blockDim.x >> 6);
    const uint64_t n_win = (a.sr_end - a.sr_begin + KC4_SR - 1) / KC4_SR;
    for (uint64_t wx = wave; wx < n_win; wx += n_waves) {
        const uint64_t sr0 = a.sr_begin + wx * KC4_SR;
        uint4 c[KC4_SR];
        kc4_load<KC4_SR>(c, a.gcov, sr0, a.sr_end, lane);
        kc4_window<WIDE, COLORED, KC4_SR>(c, sr0, lane, a.khead, a.krow, a.u0, a.n_out, a.g_begin, a.g_end, a.sr_end, a.out_sum, a.out_min,
                                          a.out_max, a.out_miss);
    }
}

}  // namespace pf
