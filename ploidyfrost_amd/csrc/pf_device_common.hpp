// Device-side helpers shared by the gfx950 kernels: 2-bit k-mer arithmetic, the open-addressing
// tables, wave64 reductions.  gfx950 only: wavefront = 64 lanes, no dual paths.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace pf {

constexpr int WAVE = 64;
constexpr uint64_t EMPTY_KEY = 0xFFFFFFFFFFFFFFFFull;
constexpr uint32_t NONE = 0xFFFFFFFFu;
constexpr uint32_t GCOV_MISSING = 0xFFFFFFFFu;  // pf_ctx::d_gcov: the k-mer is not in the count table

// 16-byte slot: one probe = one 16-B load inside one 64-B sector.
struct __attribute__((aligned(16))) Slot {
    uint64_t key;
    uint32_t val;
    uint32_t pad;
};

__host__ __device__ inline uint64_t mix64(uint64_t x) {
    x ^= x >> 30;
    x *= 0xBF58476D1CE4E5B9ull;
    x ^= x >> 27;
    x *= 0x94D049BB133111EBull;
    x ^= x >> 31;
    return x;
}

// k-mer starting at base p of a unitig whose packed words start at `w`
// (first base most significant inside each word).
__device__ inline uint64_t kmer_at(const uint64_t *__restrict__ w, uint32_t p, int k) {
    const uint32_t wi = p >> 5;
    const int s = (int)(p & 31) * 2;
    uint64_t hi = w[wi];
    uint64_t x = hi << s;
    if (s + 2 * k > 64) x |= w[wi + 1] >> (64 - s);  // s > 0 here because 2k <= 62
    return x >> (64 - 2 * k);
}

// reverse complement of a right-aligned 2-bit k-mer
__device__ inline uint64_t rc_kmer(uint64_t x, int k) {
    x = ~x;
    x = __brevll(x);
    x = ((x & 0x5555555555555555ull) << 1) | ((x >> 1) & 0x5555555555555555ull);
    return x >> (64 - 2 * k);
}

__device__ inline Slot load_slot(const Slot *t, uint64_t i) {
    // one 16-B vector load
    const uint4 v = *reinterpret_cast<const uint4 *>(t + i);
    Slot s;
    s.key = ((uint64_t)v.y << 32) | v.x;
    s.val = v.z;
    s.pad = v.w;
    return s;
}

// exact-key probe; returns true and the value when present
__device__ inline bool table_find(const Slot *__restrict__ t, uint64_t mask, uint64_t key, uint32_t &val) {
    uint64_t i = mix64(key) & mask;
    for (;;) {
        Slot s = load_slot(t, i);
        if (s.key == key) { val = s.val; return true; }
        if (s.key == EMPTY_KEY) return false;
        i = (i + 1) & mask;
    }
}

// The hot path's composite lookup (reference src/CDBG.cpp:38-56):
// "if (!IsKmer(fwd)) reverse(); CheckKmer(...)", i.e. the count of whichever orientation the database
// holds, the forward one first.  When the table is known to hold at most one orientation of every
// k-mer (`one_strand`, verified on the device at upload time -- every database written with canonical
// counting is like that) the order of the two probes cannot change the answer, so the canonical
// form, the one such databases store, is probed first: one probe per k-mer instead of ~1.5 plus the
// miss chain.
__device__ inline bool canonical_count(const Slot *__restrict__ t, uint64_t mask, uint64_t fwd, int k, uint32_t &cnt,
                                       bool one_strand) {
    const uint64_t rc = rc_kmer(fwd, k);
    const uint64_t first = (one_strand && rc < fwd) ? rc : fwd;
    if (table_find(t, mask, first, cnt)) return true;
    return table_find(t, mask, first == fwd ? rc : fwd, cnt);
}

// The rolling window of readCov(string) (src/CDBG.cpp:36-43, src/CCDBG.cpp:96-103).  The reference keeps ONE CKmerAPI object per
// call, created as k times 'A', and CKmerAPI::from_string (KMC/kmc_api/kmer_api.h:502-510) leaves it UNTOUCHED when the k characters
// hold anything but ACGT -- the '-' of an aligned row, which reaches a site string that takes raw columns up to its row's end
// (substr with a negative count, src/CDBG.cpp:1499).  Such a window is looked up with what the object held before: the previous
// clean window's k-mer in the form that was found (the composite look-up of that k-mer gives the same count again), or poly-A before
// any.  push() returns the k-mer to look up for the window that ends at this character.
struct StringWindow {
    uint64_t x = 0, held = 0;
    uint32_t clean = 0;   // ACGT characters in a row up to here
    __device__ __forceinline__ uint64_t push(char ch, uint64_t kmask, uint32_t k) {
        const bool acgt = ch == 'A' || ch == 'C' || ch == 'G' || ch == 'T';
        x = ((x << 2) | (uint64_t)(ch == 'A' ? 0 : ch == 'C' ? 1 : ch == 'G' ? 2 : 3)) & kmask;
        clean = acgt ? clean + 1 : 0;
        if (clean >= k) held = x;
        return held;
    }
};

// The same lookup split in two so that a lane can keep several k-mers in flight: count_probe() issues the first
// 16-B probe of the form that is tried first, count_finish() consumes it and walks on only if it has to.
struct CountProbe {
    uint64_t first, second;
    Slot s;
};
__device__ inline void count_probe(const Slot *__restrict__ t, uint64_t mask, uint64_t fwd, int k, bool one_strand, CountProbe &p) {
    const uint64_t rc = rc_kmer(fwd, k);
    p.first = (one_strand && rc < fwd) ? rc : fwd;
    p.second = p.first == fwd ? rc : fwd;
    p.s = load_slot(t, mix64(p.first) & mask);
}
__device__ inline bool count_finish(const Slot *__restrict__ t, uint64_t mask, const CountProbe &p, uint32_t &cnt) {
    if (p.s.key == p.first) { cnt = p.s.val; return true; }
    if (p.s.key != EMPTY_KEY) {  // occupied by another key: continue the linear probe behind it
        uint64_t i = (mix64(p.first) + 1) & mask;
        for (;;) {
            const Slot s = load_slot(t, i);
            if (s.key == p.first) { cnt = s.val; return true; }
            if (s.key == EMPTY_KEY) break;
            i = (i + 1) & mask;
        }
    }
    return table_find(t, mask, p.second, cnt);
}

__device__ inline uint64_t wave_sum_u64(uint64_t v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        uint32_t lo = __shfl_down((uint32_t)v, o, WAVE);
        uint32_t hi = __shfl_down((uint32_t)(v >> 32), o, WAVE);
        v += ((uint64_t)hi << 32) | lo;
    }
    return v;
}

__device__ inline uint32_t wave_min_u32(uint32_t v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        uint32_t x = __shfl_down(v, o, WAVE);
        v = x < v ? x : v;
    }
    return v;
}

__device__ inline int lane_id() { return (int)(threadIdx.x & (WAVE - 1)); }

// value of `x` in lane `i` (wave-uniform i): v_readlane_b32 into an SGPR, a few cycles, where a
// general __shfl is an LDS-crossbar ds_bpermute with LDS latency
__device__ inline uint32_t read_lane(uint32_t x, int i) { return (uint32_t)__builtin_amdgcn_readlane((int)x, i); }

// "ACGT"[b] without the table: a string literal indexed by a lane-dependent value is a byte load from constant memory per character
// (K-SNP's rows cost two of them per column); the four characters fit one register
__device__ inline char base_char(uint32_t b) { return (char)((0x54474341u >> (8u * (b & 3u))) & 0xFFu); }

}  // namespace pf
