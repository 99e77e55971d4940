// Prefix sums and flag selection (pf_scan.hpp): tile sums, then the tiles again, each block adding up the sums before its own
// (arrays beyond 33 M elements: one block scans the sums in between).  HBM-bound, two passes over the input (12 - 24 bytes an element); the arrays are a few million elements -- tens of
// microseconds a call.
#include "pf_scan.hpp"

namespace pf {
namespace {

constexpr int SCAN_BLOCK = 256, SCAN_ITEMS = 8, SCAN_TILE = SCAN_BLOCK * SCAN_ITEMS;
constexpr int OFFS_BLOCK = 1024;

__device__ inline uint64_t shfl_up_u64(uint64_t v, int d) {
    const uint32_t lo = __shfl_up((uint32_t)v, d, 64), hi = __shfl_up((uint32_t)(v >> 32), d, 64);
    return ((uint64_t)hi << 32) | lo;
}
// inclusive scan over the 64 lanes of a wavefront
__device__ inline uint64_t wave_scan(uint64_t v) {
    const int lane = (int)(threadIdx.x & 63);
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint64_t o = shfl_up_u64(v, d);
        if (lane >= d) v += o;
    }
    return v;
}
// inclusive scan over the threads of a block (blockDim.x / 64 <= 16 wavefronts); total = the block's sum
__device__ inline uint64_t block_scan(uint64_t v, uint64_t &total) {
    __shared__ uint64_t s_wave[16];
    const int wv = (int)(threadIdx.x >> 6), lane = (int)(threadIdx.x & 63), n_wv = (int)(blockDim.x >> 6);
    const uint64_t incl = wave_scan(v);
    __syncthreads();   // (s_wave of a previous call has been read)
    if (lane == 63) s_wave[wv] = incl;
    __syncthreads();
    uint64_t before = 0, all = 0;
    for (int w = 0; w < n_wv; ++w) {
        const uint64_t x = s_wave[w];
        if (w < wv) before += x;
        all += x;
    }
    total = all;
    return incl + before;
}

struct AsIs {
    template <class T> __device__ uint64_t operator()(T x) const { return (uint64_t)x; }
};
struct NonZero {
    template <class T> __device__ uint64_t operator()(T x) const { return x != 0 ? 1u : 0u; }
};

// a thread's SCAN_ITEMS consecutive elements of tile blockIdx.x (zero beyond n)
template <class In, class Map>
__device__ inline void load_items(const In *__restrict__ in, uint64_t n, uint64_t (&v)[SCAN_ITEMS], Map map) {
    const uint64_t base = (uint64_t)blockIdx.x * SCAN_TILE + (uint64_t)threadIdx.x * SCAN_ITEMS;
    if ((uint64_t)(blockIdx.x + 1) * SCAN_TILE <= n) {   // a whole tile: plain (vector) loads
#pragma unroll
        for (int i = 0; i < SCAN_ITEMS; ++i) v[i] = map(in[base + i]);
    } else {
#pragma unroll
        for (int i = 0; i < SCAN_ITEMS; ++i) v[i] = base + i < n ? map(in[base + i]) : 0;
    }
}

template <class In, class Map>
__global__ __launch_bounds__(SCAN_BLOCK) void k_scan_tile_sums(const In *__restrict__ in, uint64_t n, uint64_t *__restrict__ tile_sum, Map map) {
    uint64_t v[SCAN_ITEMS];
    load_items(in, n, v, map);
    uint64_t s = 0;
#pragma unroll
    for (int i = 0; i < SCAN_ITEMS; ++i) s += v[i];
    uint64_t total;
    (void)block_scan(s, total);
    if (threadIdx.x == 0) tile_sum[blockIdx.x] = total;
}

// tile sums -> tile offsets in place (one block); the total behind them and, where asked for, as a count
__global__ __launch_bounds__(OFFS_BLOCK) void k_scan_tile_offsets(uint64_t *__restrict__ tile, uint64_t n_tiles, uint32_t *__restrict__ count32,
                                                                 uint64_t *__restrict__ count64) {
    uint64_t carry = 0;
    for (uint64_t t0 = 0; t0 < n_tiles; t0 += OFFS_BLOCK) {
        const uint64_t t = t0 + threadIdx.x;
        const uint64_t x = t < n_tiles ? tile[t] : 0;
        uint64_t total;
        const uint64_t incl = block_scan(x, total);
        if (t < n_tiles) tile[t] = carry + incl - x;
        carry += total;
    }
    if (threadIdx.x == 0) {
        tile[n_tiles] = carry;
        if (count32) *count32 = (uint32_t)carry;
        if (count64) *count64 = carry;
    }
}

// what lies before tile blockIdx.x.  SUMS: `tile` holds the tiles' sums and the block adds up those before its own (a few
// thousand tiles: some dozens of loads a thread from the L2 -- cheaper than a kernel of one block between the two passes); else it
// holds the offsets k_scan_tile_offsets made of them.
template <bool SUMS>
__device__ inline uint64_t tile_offset(const uint64_t *__restrict__ tile) {
    if (!SUMS) return tile[blockIdx.x];
    uint64_t s = 0;
    for (unsigned t = threadIdx.x; t < blockIdx.x; t += SCAN_BLOCK) s += tile[t];
    uint64_t total;
    (void)block_scan(s, total);
    return total;
}

template <class In, class Out, bool INCLUSIVE, bool SUMS>
__global__ __launch_bounds__(SCAN_BLOCK) void k_scan_apply(const In *__restrict__ in, Out *__restrict__ out, uint64_t n, const uint64_t *__restrict__ tile) {
    uint64_t v[SCAN_ITEMS];
    load_items(in, n, v, AsIs());
    uint64_t s = 0;
#pragma unroll
    for (int i = 0; i < SCAN_ITEMS; ++i) s += v[i];
    const uint64_t before = tile_offset<SUMS>(tile);
    uint64_t total;
    uint64_t run = block_scan(s, total) - s + before;   // what lies before this thread's first element
    const uint64_t base = (uint64_t)blockIdx.x * SCAN_TILE + (uint64_t)threadIdx.x * SCAN_ITEMS;
#pragma unroll
    for (int i = 0; i < SCAN_ITEMS; ++i) {
        if (INCLUSIVE) run += v[i];
        if (base + i < n) out[base + i] = (Out)run;
        if (!INCLUSIVE) run += v[i];
    }
}

template <class Flag, bool SUMS>
__global__ __launch_bounds__(SCAN_BLOCK) void k_select_apply(const Flag *__restrict__ flags, uint32_t *__restrict__ ids, uint64_t n, const uint64_t *__restrict__ tile,
                                                             uint32_t *__restrict__ count32, uint64_t *__restrict__ count64) {
    uint64_t v[SCAN_ITEMS];
    load_items(flags, n, v, NonZero());
    uint64_t s = 0;
#pragma unroll
    for (int i = 0; i < SCAN_ITEMS; ++i) s += v[i];
    const uint64_t before = tile_offset<SUMS>(tile);
    uint64_t total;
    uint64_t at = block_scan(s, total) - s + before;
    if (SUMS && blockIdx.x + 1 == gridDim.x && threadIdx.x == 0) {   // (the last tile knows the count)
        if (count32) *count32 = (uint32_t)(before + total);
        if (count64) *count64 = before + total;
    }
    const uint64_t base = (uint64_t)blockIdx.x * SCAN_TILE + (uint64_t)threadIdx.x * SCAN_ITEMS;
#pragma unroll
    for (int i = 0; i < SCAN_ITEMS; ++i)
        if (v[i]) ids[at++] = (uint32_t)(base + i);
}

inline uint64_t tiles_of(uint64_t n) { return (n + SCAN_TILE - 1) / SCAN_TILE; }
// up to this many tiles (33 M elements) a block of the second pass adds up the sums of the tiles before its own; beyond, one
// block turns the sums into offsets in between (the work of the adding grows with the square of the tiles)
constexpr unsigned kSumsInApply = 16384;

template <class In, class Out, bool INCLUSIVE>
hipError_t scan_impl(const In *in, Out *out, uint64_t n, void *scratch, hipStream_t st) {
    if (n == 0) return hipSuccess;
    if (!in || !out || !scratch || tiles_of(n) > 0x7FFFFFFFull) return hipErrorInvalidValue;
    uint64_t *tile = static_cast<uint64_t *>(scratch);
    const unsigned grid = (unsigned)tiles_of(n);
    k_scan_tile_sums<In, AsIs><<<grid, SCAN_BLOCK, 0, st>>>(in, n, tile, AsIs());
    if (grid <= kSumsInApply) {
        k_scan_apply<In, Out, INCLUSIVE, true><<<grid, SCAN_BLOCK, 0, st>>>(in, out, n, tile);
    } else {
        k_scan_tile_offsets<<<1, OFFS_BLOCK, 0, st>>>(tile, grid, nullptr, nullptr);
        k_scan_apply<In, Out, INCLUSIVE, false><<<grid, SCAN_BLOCK, 0, st>>>(in, out, n, tile);
    }
    return hipGetLastError();
}

template <class Flag>
hipError_t select_impl(const Flag *flags, uint32_t *ids, uint32_t *count32, uint64_t *count64, uint64_t n, void *scratch, hipStream_t st) {
    if (!scratch || (n && (!flags || !ids)) || n > 0xFFFFFFFFull) return hipErrorInvalidValue;
    uint64_t *tile = static_cast<uint64_t *>(scratch);
    const unsigned grid = (unsigned)tiles_of(n);
    if (n == 0) {   // nothing to look at: the counts still say so
        k_scan_tile_offsets<<<1, OFFS_BLOCK, 0, st>>>(tile, 0, count32, count64);
        return hipGetLastError();
    }
    k_scan_tile_sums<Flag, NonZero><<<grid, SCAN_BLOCK, 0, st>>>(flags, n, tile, NonZero());
    if (grid <= kSumsInApply) {
        k_select_apply<Flag, true><<<grid, SCAN_BLOCK, 0, st>>>(flags, ids, n, tile, count32, count64);
    } else {
        k_scan_tile_offsets<<<1, OFFS_BLOCK, 0, st>>>(tile, grid, count32, count64);
        k_select_apply<Flag, false><<<grid, SCAN_BLOCK, 0, st>>>(flags, ids, n, tile, nullptr, nullptr);
    }
    return hipGetLastError();
}

}  // namespace

size_t scan_scratch_bytes(uint64_t n) { return (size_t)(tiles_of(n) + 2) * 8; }

hipError_t scan_exclusive_u32(const uint32_t *in, uint32_t *out, uint64_t n, void *scratch, hipStream_t st) { return scan_impl<uint32_t, uint32_t, false>(in, out, n, scratch, st); }
hipError_t scan_inclusive_u32(const uint32_t *in, uint32_t *out, uint64_t n, void *scratch, hipStream_t st) { return scan_impl<uint32_t, uint32_t, true>(in, out, n, scratch, st); }
hipError_t scan_exclusive_u32_u64(const uint32_t *in, uint64_t *out, uint64_t n, void *scratch, hipStream_t st) { return scan_impl<uint32_t, uint64_t, false>(in, out, n, scratch, st); }
hipError_t scan_exclusive_u64(const uint64_t *in, uint64_t *out, uint64_t n, void *scratch, hipStream_t st) { return scan_impl<uint64_t, uint64_t, false>(in, out, n, scratch, st); }
hipError_t select_flagged_u8(const uint8_t *flags, uint32_t *ids, uint32_t *count32, uint64_t *count64, uint64_t n, void *scratch, hipStream_t st) {
    return select_impl<uint8_t>(flags, ids, count32, count64, n, scratch, st);
}
hipError_t select_flagged_u32(const uint32_t *flags, uint32_t *ids, uint32_t *count32, uint64_t *count64, uint64_t n, void *scratch, hipStream_t st) {
    return select_impl<uint32_t>(flags, ids, count32, count64, n, scratch, st);
}

}  // namespace pf

// ---- self test (tests/test_gpu_kernels.py) ---------------------------------------------------------------------------------------
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

#include "pf_ctx.hpp"
#include "ploidyfrost_hip.h"

extern "C" int pf_selftest_scan(pf_ctx *ctx, uint64_t n, uint32_t seed) {
    if (!ctx || n == 0 || n > (1ull << 28)) return PF_ERR_ARG;
    if (hipSetDevice(ctx->device) != hipSuccess) return PF_ERR_HIP;
    // values small enough that no 32-bit sum wraps (the 32-bit scans are then comparable with 64-bit host arithmetic)
    std::vector<uint32_t> in(n);
    std::vector<uint8_t> f8(n);
    uint64_t x = 0x9E3779B97F4A7C15ull ^ seed;
    for (uint64_t i = 0; i < n; ++i) {
        x ^= x << 13; x ^= x >> 7; x ^= x << 17;
        in[i] = (uint32_t)(x % 13);
        f8[i] = (uint8_t)(((x >> 20) & 7) == 0 ? (1 + ((x >> 40) & 3)) : 0);   // one in eight flagged, with values other than 1
    }
    std::vector<uint64_t> ex(n), in64(n);
    std::vector<uint32_t> want_ids;
    uint64_t run = 0;
    for (uint64_t i = 0; i < n; ++i) { ex[i] = run; run += in[i]; in64[i] = (uint64_t)in[i] << 20; if (f8[i]) want_ids.push_back((uint32_t)i); }
    pf::DevTmp<uint32_t> d_in, d_o32, d_ids, d_f32, d_c32;
    pf::DevTmp<uint64_t> d_in64, d_o64, d_c64;
    pf::DevTmp<uint8_t> d_f8, d_tmp;
    if (d_in.alloc(n * 4) != hipSuccess || d_o32.alloc(n * 4) != hipSuccess || d_ids.alloc(n * 4) != hipSuccess || d_f32.alloc(n * 4) != hipSuccess ||
        d_c32.alloc(4) != hipSuccess || d_in64.alloc(n * 8) != hipSuccess || d_o64.alloc(n * 8) != hipSuccess || d_c64.alloc(8) != hipSuccess || d_f8.alloc(n) != hipSuccess ||
        d_tmp.alloc(pf::scan_scratch_bytes(n)) != hipSuccess) {
        pf::CtxErr{ctx} = "pf_selftest_scan: out of device memory";
        return PF_ERR_HIP;
    }
    hipStream_t st = ctx->stream;
    std::vector<uint32_t> f32(n), o32(n), ids(n);
    std::vector<uint64_t> o64(n);
    for (uint64_t i = 0; i < n; ++i) f32[i] = f8[i] ? 0x80000000u >> (i & 7) : 0;
    int bad = 0;
    auto fail = [&](const char *what, uint64_t i) { if (!bad) pf::CtxErr{ctx} = std::string("pf_selftest_scan: ") + what + " differs at " + std::to_string(i); ++bad; };
#define PF_T(call) do { if ((call) != hipSuccess) { pf::CtxErr{ctx} = std::string(#call) + " failed"; return PF_ERR_HIP; } } while (0)
    PF_T(hipMemcpyAsync(d_in.p, in.data(), n * 4, hipMemcpyHostToDevice, st));
    PF_T(hipMemcpyAsync(d_in64.p, in64.data(), n * 8, hipMemcpyHostToDevice, st));
    PF_T(hipMemcpyAsync(d_f8.p, f8.data(), n, hipMemcpyHostToDevice, st));
    PF_T(hipMemcpyAsync(d_f32.p, f32.data(), n * 4, hipMemcpyHostToDevice, st));
    if (getenv("PF_SCAN_TIMES")) {   // diag: the four scans and a selection over n elements, alone on the device
        hipEvent_t a, b;
        PF_T(hipEventCreate(&a));
        PF_T(hipEventCreate(&b));
        PF_T(hipStreamSynchronize(st));
        auto timed = [&](const char *what, auto &&call) {
            (void)call();   // (first launch)
            (void)hipEventRecord(a, st);
            for (int r = 0; r < 10; ++r) (void)call();
            (void)hipEventRecord(b, st);
            (void)hipEventSynchronize(b);
            float ms = 0;
            (void)hipEventElapsedTime(&ms, a, b);
            fprintf(stderr, "[scan] %-22s n = %llu: %.1f us\n", what, (unsigned long long)n, ms * 100.0);
        };
        timed("exclusive u32", [&] { return pf::scan_exclusive_u32(d_in.p, d_o32.p, n, d_tmp.p, st); });
        timed("inclusive u32", [&] { return pf::scan_inclusive_u32(d_in.p, d_o32.p, n, d_tmp.p, st); });
        timed("exclusive u32 -> u64", [&] { return pf::scan_exclusive_u32_u64(d_in.p, d_o64.p, n, d_tmp.p, st); });
        timed("exclusive u64", [&] { return pf::scan_exclusive_u64(d_in64.p, d_o64.p, n, d_tmp.p, st); });
        timed("select u8", [&] { return pf::select_flagged_u8(d_f8.p, d_ids.p, d_c32.p, d_c64.p, n, d_tmp.p, st); });
        (void)hipEventDestroy(a);
        (void)hipEventDestroy(b);
    }
    PF_T(pf::scan_exclusive_u32(d_in.p, d_o32.p, n, d_tmp.p, st));
    PF_T(hipMemcpyAsync(o32.data(), d_o32.p, n * 4, hipMemcpyDeviceToHost, st));
    PF_T(hipStreamSynchronize(st));
    for (uint64_t i = 0; i < n && bad < 4; ++i) if (o32[i] != (uint32_t)ex[i]) fail("exclusive u32", i);
    PF_T(pf::scan_inclusive_u32(d_in.p, d_o32.p, n, d_tmp.p, st));
    PF_T(hipMemcpyAsync(o32.data(), d_o32.p, n * 4, hipMemcpyDeviceToHost, st));
    PF_T(hipStreamSynchronize(st));
    for (uint64_t i = 0; i < n && bad < 4; ++i) if (o32[i] != (uint32_t)(ex[i] + in[i])) fail("inclusive u32", i);
    PF_T(pf::scan_exclusive_u32_u64(d_in.p, d_o64.p, n, d_tmp.p, st));
    PF_T(hipMemcpyAsync(o64.data(), d_o64.p, n * 8, hipMemcpyDeviceToHost, st));
    PF_T(hipStreamSynchronize(st));
    for (uint64_t i = 0; i < n && bad < 4; ++i) if (o64[i] != ex[i]) fail("exclusive u32 -> u64", i);
    PF_T(pf::scan_exclusive_u64(d_in64.p, d_o64.p, n, d_tmp.p, st));   // (sums beyond 32 bits)
    PF_T(hipMemcpyAsync(o64.data(), d_o64.p, n * 8, hipMemcpyDeviceToHost, st));
    PF_T(hipStreamSynchronize(st));
    for (uint64_t i = 0; i < n && bad < 4; ++i) if (o64[i] != ex[i] << 20) fail("exclusive u64", i);
    for (int kind = 0; kind < 2; ++kind) {
        uint32_t c32 = 0;
        uint64_t c64 = 0;
        PF_T(hipMemsetAsync(d_ids.p, 0xFF, n * 4, st));
        if (kind == 0) PF_T(pf::select_flagged_u8(d_f8.p, d_ids.p, d_c32.p, d_c64.p, n, d_tmp.p, st));
        else PF_T(pf::select_flagged_u32(d_f32.p, d_ids.p, d_c32.p, d_c64.p, n, d_tmp.p, st));
        PF_T(hipMemcpyAsync(ids.data(), d_ids.p, n * 4, hipMemcpyDeviceToHost, st));
        PF_T(hipMemcpyAsync(&c32, d_c32.p, 4, hipMemcpyDeviceToHost, st));
        PF_T(hipMemcpyAsync(&c64, d_c64.p, 8, hipMemcpyDeviceToHost, st));
        PF_T(hipStreamSynchronize(st));
        if (c32 != want_ids.size() || c64 != want_ids.size()) fail(kind ? "select u32: count" : "select u8: count", c64);
        for (uint64_t i = 0; i < want_ids.size() && bad < 4; ++i) if (ids[i] != want_ids[i]) fail(kind ? "select u32" : "select u8", i);
        if (want_ids.size() < n && ids[want_ids.size()] != 0xFFFFFFFFu) fail("select: wrote past its count", want_ids.size());
    }
#undef PF_T
    return bad ? PF_ERR_HIP : PF_OK;
}
