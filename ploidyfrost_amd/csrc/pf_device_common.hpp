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

// exact-key probe of a table addressed by mix64 alone (K-ADJ's end-k-mer table); returns true and the value when present
__device__ inline bool table_find(const Slot *__restrict__ t, uint64_t mask, uint64_t key, uint32_t &val) {
    uint64_t i = mix64(key) & mask;
    for (;;) {
        Slot s = load_slot(t, i);
        if (s.key == key) { val = s.val; return true; }
        if (s.key == EMPTY_KEY) return false;
        i = (i + 1) & mask;
    }
}

// ---- count tables: lines of ten keys, addressed by the key's minimizer ---------------------------------------------------------
// CKMCFile::CheckKmer (KMC/kmc_api/kmc_file.cpp:330-366) is asked for the k-mers of a unitig one after the other
// (src/CDBG.cpp:66-120): neighbours that overlap in k - 1 bases.  A table addressed by a hash of the whole k-mer sends each of
// them to a line of its own -- 233 M random lines for the 5 M-unitig graph, which is what the row-activation rate of the HBM
// stacks gives (measured: 7.8 ms).  Here a k-mer's line is chosen by its MINIMIZER: the canonical 16-mer of smallest hash among
// the k - 15 it holds.  The k-mers of a unitig share a minimizer in runs of up to k - 15 (ten at k = 25: a quarter of the runs,
// 44 % of the k-mers; 5.8 on average), and the lanes of a wavefront that hold a run fetch one 128-B line between them.  The
// minimizer of a k-mer and of its reverse complement are the same (the set of canonical m-mers is), so the second probe of the
// composite look-up reads the same line.
//   line: 128 B = ten keys (all ones = free) + their ten counts: a look-up loads the 80 bytes of keys, then one count.
//   probe sequence of key x: the minimizer's line, its buddy (line ^ 1: the same DRAM page; takes what a run of more than ten keys
//   or two minimizers on one line leave over -- one key in six at 3.3 keys a line), a second pair of lines elsewhere (double
//   hashing: full lines do not grow into clusters), and then, only when all four are full (a minimizer shared by dozens of
//   k-mers: a repeat family), the lines from mix64(x) on, one after the other.  Nothing is ever removed: "a free slot in a line of
//   the sequence ends the search" holds throughout.
// k < HOME_M + 2: the line is chosen by mix64 of the canonical k-mer.
constexpr int HOME_M = 16;
constexpr int LINE_KEYS = 10;
constexpr int LINE_TRIES = 4;

struct __attribute__((aligned(128))) CountLine {
    uint64_t key[LINE_KEYS];
    uint32_t val[LINE_KEYS];
    uint32_t pad[2];
};

struct LineSeq {
    uint64_t line, step;   // step: even, so that a second pair is another pair
};
__device__ inline uint64_t seq_line(const LineSeq &sq, int i, uint64_t mask) { return ((sq.line + (uint64_t)(i >> 1) * sq.step) & mask) ^ (uint64_t)(i & 1); }

__device__ inline LineSeq kmer_lines(uint64_t fwd, uint64_t rc, int k, uint64_t mask) {
    uint64_t h;
    if (k < HOME_M + 2) {
        h = mix64(fwd < rc ? fwd : rc);
    } else {
        static_assert(HOME_M == 16, "the m-mers are taken as whole 32-bit words");
        uint64_t a = fwd;                  // low word of a >> 2j: the m-mer that ends j bases before the k-mer's end
        uint64_t b = rc << (64 - 2 * k);   // high word of b << 2j: the reverse complement of that m-mer
        uint32_t best = 0xFFFFFFFFu;
        for (int j = k - HOME_M; j >= 0; --j) {
            const uint32_t f = (uint32_t)a;
            const uint32_t r = (uint32_t)(b >> 32);
            const uint32_t c = f < r ? f : r;
            const uint32_t x = c * 0x9E3779B1u;   // odd multiplier: a bijection of the m-mers, so the smallest product names one m-mer
            best = x < best ? x : best;
            a >>= 2;
            b <<= 2;
        }
        // the smallest of ten products is small: mixed again before it names a line (murmur3's finalizer, twice for line and step)
        uint32_t x = best;
        x ^= x >> 16; x *= 0x85EBCA6Bu; x ^= x >> 13; x *= 0xC2B2AE35u; x ^= x >> 16;
        uint32_t y = x * 0x9E3779B1u;
        y ^= y >> 15;
        h = ((uint64_t)y << 32) | x;
    }
    return LineSeq{(h ^ (h >> 32)) & mask, ((h >> 32) | 1) << 1};
}

// the ten keys of a line in registers
struct LineKeys {
    uint4 q[LINE_KEYS / 2];
    __device__ inline uint64_t key(int i) const { return (i & 1) ? (((uint64_t)q[i >> 1].w << 32) | q[i >> 1].z) : (((uint64_t)q[i >> 1].y << 32) | q[i >> 1].x); }
};
__device__ inline LineKeys load_line_keys(const CountLine *__restrict__ t, uint64_t line) {
    LineKeys k;
    const uint4 *p = reinterpret_cast<const uint4 *>(t[line].key);
#pragma unroll
    for (int i = 0; i < LINE_KEYS / 2; ++i) k.q[i] = p[i];
    return k;
}
// slot of `key` among the loaded keys (-1: not there); open = a free slot lies before the end of the line
__device__ inline int line_slot(const LineKeys &k, uint64_t key, bool &open) {
    int at = -1;
    open = false;
#pragma unroll
    for (int i = 0; i < LINE_KEYS; ++i) {
        const uint64_t x = k.key(i);
        if (x == key) at = i;
        if (x == EMPTY_KEY) open = true;
    }
    return at;
}

// exact-key probe of a count table along the key's line sequence, from try `first_try` on
__device__ inline bool count_find(const CountLine *__restrict__ t, uint64_t mask, uint64_t key, const LineSeq &sq, uint32_t &val, int first_try = 0) {
    for (int i = first_try; i < LINE_TRIES; ++i) {
        const uint64_t line = seq_line(sq, i, mask);
        bool open;
        const int at = line_slot(load_line_keys(t, line), key, open);
        if (at >= 0) { val = t[line].val[at]; return true; }
        if (open) return false;
    }
    for (uint64_t line = mix64(key) & mask;; line = (line + 1) & mask) {
        bool open;
        const int at = line_slot(load_line_keys(t, line), key, open);
        if (at >= 0) { val = t[line].val[at]; return true; }
        if (open) return false;
    }
}

// the place of a key's count, claiming the first free slot of its probe sequence (K-TABLE).  The ten keys of a line are read at
// once (ten independent loads that bypass the XCD's L2, which is not coherent with the other XCDs' atomics): what they show as taken
// stays taken, what they show as free is claimed by compare-and-swap, which tells when another insert was quicker.
__device__ inline uint32_t *count_claim(CountLine *t, uint64_t mask, uint64_t key, const LineSeq &sq) {
    uint64_t line = 0;
    for (int i = 0;; ++i) {
        line = i < LINE_TRIES ? seq_line(sq, i, mask) : i == LINE_TRIES ? (mix64(key) & mask) : ((line + 1) & mask);
        unsigned long long seen[LINE_KEYS];
#pragma unroll
        for (int s = 0; s < LINE_KEYS; ++s)
            seen[s] = __hip_atomic_load(reinterpret_cast<unsigned long long *>(&t[line].key[s]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
        for (int s = 0; s < LINE_KEYS; ++s) {
            unsigned long long old = seen[s];
            if (old == EMPTY_KEY) old = atomicCAS(reinterpret_cast<unsigned long long *>(&t[line].key[s]), EMPTY_KEY, key);
            if (old == EMPTY_KEY || old == key) return &t[line].val[s];
        }
    }
}

// The hot path's composite lookup (reference src/CDBG.cpp:38-56):
// "if (!IsKmer(fwd)) reverse(); CheckKmer(...)", i.e. the count of whichever orientation the database
// holds, the forward one first.  When the table is known to hold at most one orientation of every
// k-mer (`one_strand`, verified on the device at upload time -- every database written with canonical
// counting is like that) the order of the two probes cannot change the answer, so the canonical
// form, the one such databases store, is probed first: one probe per k-mer instead of ~1.5 plus the
// miss chain.
__device__ inline bool canonical_count(const CountLine *__restrict__ t, uint64_t mask, uint64_t fwd, int k, uint32_t &cnt,
                                       bool one_strand) {
    const uint64_t rc = rc_kmer(fwd, k);
    const LineSeq sq = kmer_lines(fwd, rc, k, mask);
    const uint64_t first = (one_strand && rc < fwd) ? rc : fwd;
    if (count_find(t, mask, first, sq, cnt)) return true;
    return count_find(t, mask, first == fwd ? rc : fwd, sq, cnt);
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

// The same lookup split in two so that a lane can keep several k-mers in flight: count_probe() issues the loads of the keys of
// the first line of the sequence (both forms of the k-mer have the same), count_finish() consumes them and goes on only if it has to.
struct CountProbe {
    uint64_t first, second;
    LineSeq sq;
    LineKeys keys;
};
__device__ inline void count_probe_at(const CountLine *__restrict__ t, uint64_t key, const LineSeq &sq, CountProbe &p) {
    p.first = p.second = key;
    p.sq = sq;
    p.keys = load_line_keys(t, sq.line);
}
__device__ inline void count_probe(const CountLine *__restrict__ t, uint64_t mask, uint64_t fwd, int k, bool one_strand, CountProbe &p) {
    const uint64_t rc = rc_kmer(fwd, k);
    const uint64_t first = (one_strand && rc < fwd) ? rc : fwd;
    count_probe_at(t, first, kmer_lines(fwd, rc, k, mask), p);
    p.second = first == fwd ? rc : fwd;
}
__device__ inline bool count_finish(const CountLine *__restrict__ t, uint64_t mask, const CountProbe &p, uint32_t &cnt) {
    bool open;
    int at = line_slot(p.keys, p.first, open);
    if (at >= 0) { cnt = t[p.sq.line].val[at]; return true; }
    if (!open && count_find(t, mask, p.first, p.sq, cnt, 1)) return true;   // the line is full of other keys: the rest of the sequence
    if (p.second == p.first) return false;
    at = line_slot(p.keys, p.second, open);
    if (at >= 0) { cnt = t[p.sq.line].val[at]; return true; }
    return !open && count_find(t, mask, p.second, p.sq, cnt, 1);
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
