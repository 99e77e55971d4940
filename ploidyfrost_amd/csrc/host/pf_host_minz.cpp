#include "pf_host_minz.hpp"

#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <unordered_map>
#include <unordered_set>

#include "pf_host_colors.hpp"  // bifrost_kmer_hash
#include "pf_parallel.hpp"

namespace pfh {
namespace {

const uint64_t kHvals[4] = {2053695854357871005ULL, 5073395517033431291ULL, 10060236952204337488ULL, 7783083932390163561ULL};
const uint64_t kMaskId = 0xffffffff00000000ull;    // MASK_CONTIG_ID   (bifrost/src/CompactedDBG.hpp:47)
const uint64_t kMaskType = 0x80000000ull;          // MASK_CONTIG_TYPE (CompactedDBG.hpp:48)
const uint32_t kAbundanceLim = 15;                 // min_abundance_lim == max_abundance_lim (CompactedDBG.hpp:742-743)

inline uint64_t rotl(uint64_t x, int r) { return (x << r) | (x >> (64 - r)); }
inline uint64_t rotr1(uint64_t x) { return (x >> 1) | (x << 63); }
inline uint64_t wymix64(uint64_t a, uint64_t b) {
    __uint128_t r = (__uint128_t)a * b;
    return (uint64_t)r ^ (uint64_t)(r >> 64);
}
inline unsigned charmask(unsigned char c) { return (c & 6) >> 1; }
inline unsigned twinmask(unsigned char c) { return ((c ^ 4) & 6) >> 1; }
inline uint64_t code2(char c) { return (uint64_t)(((unsigned char)c >> 1) ^ ((unsigned char)c >> 2)) & 3; }  // A0 C1 G2 T3

// bifrost/src/RepHash.hpp:24-103: two rolling words (one per strand), hashed strand-symmetrically with wyhash
struct RollHash {
    int g = 0;
    uint64_t h = 0, ht = 0;
    void init(const char *s) {
        h = ht = 0;
        for (int i = 0; i < g; ++i) {
            h = rotl(h, 1) ^ kHvals[charmask((unsigned char)s[i])];
            ht = rotl(ht, 1) ^ kHvals[twinmask((unsigned char)s[g - 1 - i])];
        }
    }
    void update(unsigned char out, unsigned char in) {  // updateFW, RepHash.hpp:47-63
        h = rotl(h, 1) ^ rotl(kHvals[charmask(out)], g) ^ kHvals[charmask(in)];
        ht = rotr1(ht ^ rotl(kHvals[twinmask(in)], g) ^ kHvals[twinmask(out)]);
    }
    uint64_t hash() const {  // wyhash (final version 3) of the 16 bytes {min, max}, seed 0
        const uint64_t lo = h < ht ? h : ht, hi = h < ht ? ht : h;
        const uint64_t wyp0 = 0xa0761d6478bd642full, wyp1 = 0xe7037ed1a0b428dbull;
        const uint64_t a = ((lo & 0xFFFFFFFFull) << 32) | (hi & 0xFFFFFFFFull);
        const uint64_t b = ((hi >> 32) << 32) | (lo >> 32);
        return wymix64(wyp1 ^ 16, wymix64(a ^ wyp1, b ^ wyp0));
    }
};

inline uint64_t gmer_hash(const char *s, int g) {
    RollHash r;
    r.g = g;
    r.init(s);
    return r.hash();
}

// Minimizer(s).rep() as a number (same length on both sides of every comparison made with it)
inline uint64_t minz_rep(const char *s, int g) {
    uint64_t fw = 0, rc = 0;
    for (int i = 0; i < g; ++i) {
        const uint64_t c = code2(s[i]);
        fw = (fw << 2) | c;
        rc |= (3 - c) << (2 * i);
    }
    return fw < rc ? fw : rc;
}

struct MinRes {
    uint64_t hash;
    int pos;
};

// minHashIterator<RepHash>(s, n, k, g, RepHash(), nh = true), bifrost/src/minHashIterator.hpp:28-262: the monotone
// queue of candidate positions of the current k-mer; a minimizer may not start at offset 0 or k-g of its k-mer
struct MinIt {
    const char *s;
    int n, k, g;
    int p = -1;
    bool invalid = false;
    std::vector<MinRes> &v;
    size_t a = 0, b = 0;  // the queue is v[a .. b)
    RollHash hf;

    MinIt(const char *s_, int n_, int k_, int g_, std::vector<MinRes> &scratch) : s(s_), n(n_), k(k_), g(g_), v(scratch) {
        hf.g = g;
        if ((int)v.size() < n + 2) v.resize((size_t)n + 2);
        if (n < k) invalid = true;
        else next();
    }
    void push(uint64_t h, int pos) {
        while (b > a && v[b - 1].hash > h) --b;
        v[b++] = MinRes{h, pos};
    }
    void next() {  // operator++, minHashIterator.hpp:63-119
        if (invalid) return;
        ++p;
        if (p >= n - k + 1) { invalid = true; return; }
        if (p == 0) {
            hf.init(s + 1);
            v[b++] = MinRes{hf.hash(), 1};
            for (int j = 1; j < k - g - 1;) {
                hf.update((unsigned char)s[j], (unsigned char)s[j + g]);
                ++j;
                push(hf.hash(), j);
            }
        } else {
            if (v[a].pos < p + 1) ++a;
            hf.update((unsigned char)s[p + k - g - 2], (unsigned char)s[p + k - 2]);
            push(hf.hash(), p + k - g - 1);
        }
    }
    int front_pos() const { return v[a].pos; }
    // getNewMin, minHashIterator.hpp:145-186: the best minimizer of the current k-mer whose hash is above the discarded one
    MinRes new_min(const MinRes &discard) const {
        const int end = p + k - g - 1;
        int j = p + 1;
        uint64_t h = gmer_hash(s + j, g);
        while (h <= discard.hash && j < end) { ++j; h = gmer_hash(s + j, g); }
        if (j == end && h <= discard.hash) return discard;
        MinRes m{h, j};
        while (j < end) {
            ++j;
            h = gmer_hash(s + j, g);
            if (h <= m.hash && h > discard.hash) {
                if ((h == m.hash && minz_rep(s + j, g) < minz_rep(s + m.pos, g)) || h != m.hash) m = MinRes{h, j};
            }
        }
        return m;
    }
};

// the minimizer positions addUnitig files when no bucket is crowded (tcc:3959-3969): fn(position) per occurrence
template <class F>
inline void plain_events(const char *s, int n, int k, int g, std::vector<MinRes> &scratch, F &&fn) {
    MinIt it(s, n, k, g, scratch);
    for (int64_t last = -1; !it.invalid; it.next()) {
        if (last < it.front_pos()) {
            for (size_t t = it.a;; ++t) {
                fn(it.v[t].pos);
                last = it.v[t].pos;
                if (t + 1 >= it.b || it.v[t + 1].hash != it.v[t].hash) break;
            }
        }
    }
}

inline uint64_t mix64(uint64_t x) {
    x ^= x >> 33; x *= 0xff51afd7ed558ccdull;
    x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ull;
    x ^= x >> 33;
    return x;
}

// packed_tiny_vector as far as addUnitig looks into it: its size and its last two entries
struct Bucket {
    uint32_t size = 0;
    uint64_t last = 0, prev = 0;
    void push_back(uint64_t x) { prev = last; last = x; ++size; }
    void insert_before_last(uint64_t x) { prev = x; ++size; }
};

// KmerHashTable<T> (bifrost/src/KmerHashTable.hpp:84, 223-255, 326-354): 1024 slots, linear probing from
// Kmer::hash() & (size-1), doubled (and refilled in slot order) when fewer than a fifth of the slots are empty
struct KmerTable {
    std::vector<uint64_t> key;
    std::vector<uint32_t> val;   // UINT32_MAX = empty
    size_t num_empty = 0;
    KmerTable() { init(1024); }
    void init(size_t sz) { key.assign(sz, 0); val.assign(sz, UINT32_MAX); num_empty = sz; }
    void place(uint64_t km, uint32_t v) {
        const size_t m = key.size() - 1;
        size_t h = bifrost_kmer_hash(km, 0) & m;
        while (val[h] != UINT32_MAX) h = (h + 1) & m;
        key[h] = km;
        val[h] = v;
        --num_empty;
    }
    void insert(uint64_t km, uint32_t v) {
        if (5 * num_empty < key.size()) {
            std::vector<uint64_t> ok;
            std::vector<uint32_t> ov;
            ok.swap(key);
            ov.swap(val);
            init(2 * ok.size());
            for (size_t i = 0; i < ok.size(); ++i)
                if (ov[i] != UINT32_MAX) place(ok[i], ov[i]);
        }
        place(km, v);
    }
};

struct Replay {
    int k, g;
    const std::atomic<uint8_t> *cnt;
    uint64_t cnt_mask;
    const std::unordered_set<uint64_t> *extra;
    std::unordered_map<uint64_t, Bucket> buckets;
    std::unordered_set<uint64_t> newly;
    std::vector<MinRes> scratch;
    std::vector<std::pair<uint64_t, Bucket>> undo;

    bool tracked(uint64_t m) const {
        return cnt[mix64(m) & cnt_mask].load(std::memory_order_relaxed) >= kAbundanceLim || extra->count(m) != 0;
    }
    Bucket *bucket(uint64_t m, bool via_redirect) {
        if (tracked(m)) return &buckets[m];
        if (via_redirect) newly.insert(m);   // an entry this bucket's counter never saw: follow it exactly from now on
        return nullptr;                       // fewer than 15 entries for certain; its content decides nothing
    }

    // addUnitig (tcc:3928-4080) for one S-line; true when the k-length unitig is filed as abundant
    bool add(const char *s, int len, uint64_t id) {
        const uint64_t mask = kMaskId | kMaskType;
        bool is_short = len == k, is_abundant = false, forbidden = false;
        uint64_t pid = id << 32;
        if (is_short) pid |= kMaskType;
        undo.clear();
        {
            MinIt it(s, len, k, g, scratch);
            for (int64_t last = -1; !it.invalid && !is_abundant; it.next()) {
                if (!(last < it.front_pos() || forbidden)) continue;
                forbidden = false;
                for (size_t t = it.a;; ++t) {
                    const MinRes res = it.v[t];
                    Bucket *b = bucket(minz_rep(s + res.pos, g), false);
                    pid = (pid & mask) | (uint64_t)res.pos;
                    if (!is_short) {
                        MinRes mhr = res;
                        while (b && (b->size >= kAbundanceLim || (b->size > 0 && (b->last & mask) == mask))) {
                            const MinRes alt = it.new_min(mhr);
                            forbidden = true;
                            if (alt.hash == mhr.hash) break;
                            if ((b->last & mask) != mask) {  // first time this bucket is found crowded
                                if ((b->last & kMaskId) == kMaskId) b->last |= kMaskType;
                                else b->push_back(mask);
                            }
                            mhr = alt;
                            b = bucket(minz_rep(s + mhr.pos, g), true);
                        }
                    }
                    if (b) {
                        if (is_short && b->size >= kAbundanceLim) {
                            is_abundant = true;
                            break;
                        }
                        if (is_short) undo.emplace_back(minz_rep(s + res.pos, g), *b);
                        if (b->size == 0) b->push_back(pid);
                        else if ((b->last & kMaskId) == kMaskId) {  // an abundant counter or the crowded mark stays last
                            if (b->size == 1 || b->prev != pid) b->insert_before_last(pid);
                        } else if (b->last != pid) b->push_back(pid);
                    }
                    last = res.pos;
                    if (t + 1 >= it.b || it.v[t + 1].hash != it.v[t].hash) break;
                }
            }
        }
        if (!is_abundant) return false;
        // deleteUnitig_(true, false, id, false) (tcc:5291-5384) takes out what this call filed before it met the crowded bucket
        for (size_t i = undo.size(); i-- > 0;) buckets[undo[i].first] = undo[i].second;
        MinIt it(s, len, k, g, scratch);
        for (int64_t last = -1; !it.invalid; it.next()) {  // tcc:4039-4065
            if (!(last < it.front_pos())) continue;
            for (size_t t = it.a;; ++t) {
                const MinRes res = it.v[t];
                if (Bucket *b = bucket(minz_rep(s + res.pos, g), false)) {
                    if (b->size > 0 && (b->last & kMaskId) == kMaskId) b->last++;
                    else b->push_back(kMaskId + 1);
                }
                last = res.pos;
                if (t + 1 >= it.b || it.v[t + 1].hash != it.v[t].hash) break;
            }
        }
        return true;
    }
};

}  // namespace

void bifrost_numbering(int k, int g, const std::vector<SegRef> &segs, unsigned threads, UnitigNumbering &out, std::vector<uint8_t> *counters,
                       const uint8_t *counters_in, const uint8_t *touches_in, uint64_t counters_in_len) {
    const bool trace = getenv("PF_TRACE_LOAD") != nullptr;
    auto t_last = std::chrono::steady_clock::now();
    auto mark = [&](const char *what) {
        if (!trace) return;
        const auto now = std::chrono::steady_clock::now();
        fprintf(stderr, "[load]   numbering: %-18s %.3fs\n", what, std::chrono::duration<double>(now - t_last).count());
        t_last = now;
    };
    out = UnitigNumbering();
    const size_t S = segs.size();
    if (S == 0 || g < 1 || g > k - 2 || g > 31) return;
    uint64_t n_kmers = 0;
    for (const SegRef &sg : segs) n_kmers += sg.len - (uint32_t)k + 1;
    uint64_t cap = 1ull << 16;
    while (cap < n_kmers / 2 && cap < (1ull << 30)) cap <<= 1;
    std::unique_ptr<std::atomic<uint8_t>[]> cnt(new std::atomic<uint8_t>[cap]);
    const size_t kChunk = 4096;
    const bool from_device = counters_in != nullptr && touches_in != nullptr && counters_in_len == cap;   // (same table geometry, or not used)
    parallel_chunks((size_t)cap, (size_t)1 << 20, threads, [&](size_t, size_t b0, size_t b1) {
        for (size_t i = b0; i < b1; ++i) cnt[i].store(from_device ? counters_in[i] : (uint8_t)0, std::memory_order_relaxed);
    });
    mark("counter table");
    std::atomic<bool> any_crowded{from_device};
    if (!from_device) parallel_chunks(S, kChunk, threads, [&](size_t, size_t s0, size_t s1) {
        std::vector<MinRes> scratch;
        bool crowded = false;
        for (size_t i = s0; i < s1; ++i) {
            const SegRef &sg = segs[i];
            plain_events(sg.s, (int)sg.len, k, g, scratch, [&](int pos) {
                std::atomic<uint8_t> &c = cnt[mix64(minz_rep(sg.s + pos, g)) & (cap - 1)];
                if (c.load(std::memory_order_relaxed) < 255 && c.fetch_add(1, std::memory_order_relaxed) + 1 >= kAbundanceLim) crowded = true;
            });
        }
        if (crowded) any_crowded.store(true, std::memory_order_relaxed);
    });
    mark("count occurrences");
    if (counters) {
        counters->resize(cap);
        for (uint64_t i = 0; i < cap; ++i) (*counters)[i] = cnt[i].load(std::memory_order_relaxed);
    }
    if (!any_crowded.load()) return;

    std::unordered_set<uint64_t> extra;
    std::vector<uint8_t> touches(S);
    for (;;) {
        Replay rp;
        rp.k = k;
        rp.g = g;
        rp.cnt = cnt.get();
        rp.cnt_mask = cap - 1;
        rp.extra = &extra;
        if (from_device && extra.empty()) memcpy(touches.data(), touches_in, S);   // (first round: the device's flags)
        else parallel_chunks(S, kChunk, threads, [&](size_t, size_t s0, size_t s1) {
            std::vector<MinRes> scratch;
            for (size_t i = s0; i < s1; ++i) {
                const SegRef &sg = segs[i];
                bool hit = false;
                plain_events(sg.s, (int)sg.len, k, g, scratch, [&](int pos) {
                    if (!hit && rp.tracked(minz_rep(sg.s + pos, g))) hit = true;
                });
                touches[i] = hit;
            }
        });
        mark("flag unitigs");
        out.replays++;
        out.replayed_unitigs = 0;
        KmerTable table;
        uint64_t n_long = 0, n_short = 0;
        for (size_t i = 0; i < S; ++i) {
            const SegRef &sg = segs[i];
            const bool is_short = (int)sg.len == k;
            bool abundant = false;
            if (touches[i]) {
                abundant = rp.add(sg.s, (int)sg.len, is_short ? n_short : n_long);
                out.replayed_unitigs++;
            }
            if (abundant) {
                uint64_t km = 0;
                for (int j = 0; j < k; ++j) km |= code2(sg.s[j]) << (62 - 2 * j);
                table.insert(km, (uint32_t)i);
            } else if (is_short) ++n_short;
            else ++n_long;
        }
        mark("replay");
        if (!rp.newly.empty()) {
            extra.insert(rp.newly.begin(), rp.newly.end());
            continue;
        }
        out.tracked_buckets = rp.buckets.size();
        out.abundant.clear();
        for (size_t h = 0; h < table.val.size(); ++h)
            if (table.val[h] != UINT32_MAX) out.abundant.push_back(table.val[h]);
        return;
    }
}

}  // namespace pfh
