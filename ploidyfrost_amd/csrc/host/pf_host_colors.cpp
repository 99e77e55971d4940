#include "pf_host_colors.hpp"

#include <algorithm>
#include <atomic>
#include <cstdio>
#include <cstring>
#include <mutex>

#include "pf_parallel.hpp"

namespace pfh {

// ---- wyhash final version 3 (public domain, github.com/wangyi-fudan/wyhash) for an 8-byte key ----------
namespace {
inline uint64_t wymix(uint64_t a, uint64_t b) {
    __uint128_t r = (__uint128_t)a * b;
    return (uint64_t)r ^ (uint64_t)(r >> 64);
}
const uint64_t kWyp0 = 0xa0761d6478bd642full, kWyp1 = 0xe7037ed1a0b428dbull;
}  // namespace

uint64_t bifrost_kmer_hash(uint64_t x, uint64_t seed) {
    // len = 8: a = (first 4 bytes << 32) | last 4 bytes, b = (last 4 bytes << 32) | first 4 bytes, little endian
    const uint64_t lo = x & 0xFFFFFFFFull, hi = x >> 32;
    const uint64_t a = (lo << 32) | hi, b = (hi << 32) | lo;
    seed ^= kWyp0;
    return wymix(kWyp1 ^ 8, wymix(a ^ kWyp1, b ^ seed));
}

namespace {

typedef std::pair<uint64_t, uint64_t> Run;  // inclusive [first, second]

struct Reader {
    const uint8_t *p, *end;
    bool ok = true;
    template <class T>
    T get() {
        T v{};
        if ((size_t)(end - p) < sizeof(T)) { ok = false; p = end; return v; }
        memcpy(&v, p, sizeof(T));
        p += sizeof(T);
        return v;
    }
    const uint8_t *take(size_t n) {
        if ((size_t)(end - p) < n) { ok = false; p = end; return nullptr; }
        const uint8_t *r = p;
        p += n;
        return r;
    }
};

void push_value(std::vector<Run> &runs, uint64_t v) {
    if (!runs.empty() && runs.back().second + 1 == v) runs.back().second = v;
    else runs.emplace_back(v, v);
}

// TinyBitmap::read layout (bifrost/src/TinyBitmap.cpp:851-880; modes :1407-1412, iteration :1309-1394)
bool decode_tiny(Reader &r, std::vector<Run> *runs, std::string &err) {
    const uint16_t header = r.get<uint16_t>();
    const uint16_t sz = header >> 3, mode = header & 0x6;
    if (!r.ok) { err = "truncated TinyBitmap"; return false; }
    if (sz == 0) return true;
    const uint8_t *raw = r.take((size_t)(sz - 1) * 2);
    if (!raw || sz < 3) { err = "truncated TinyBitmap"; return false; }
    if (header & 1) { err = "TinyBitmap in 32-bit mode is not a format Bifrost writes"; return false; }
    if (!runs) return true;
    auto w = [&](uint32_t i) -> uint16_t { uint16_t v; memcpy(&v, raw + (size_t)(i - 1) * 2, 2); return v; };
    const uint32_t card = w(1);
    const uint64_t offset = (uint64_t)w(2) << 16;
    if (mode == 0x0) {
        for (uint32_t i = 3; i < sz; ++i) {
            uint16_t e = w(i);
            for (uint32_t j = 0; e; e >>= 1, ++j)
                if (e & 1) push_value(*runs, offset | (((uint64_t)(i - 3) << 4) + j));
        }
    } else if (mode == 0x2) {
        if (card + 3 > sz) { err = "TinyBitmap list longer than its block"; return false; }
        for (uint32_t i = 3; i < card + 3; ++i) push_value(*runs, offset | w(i));
    } else if (mode == 0x4) {
        if (card + 3 > sz || (card & 1)) { err = "TinyBitmap run list longer than its block"; return false; }
        for (uint32_t i = 3; i + 1 < card + 3; i += 2) {
            const uint64_t a = offset | w(i), b = offset | w(i + 1);
            if (b < a) { err = "TinyBitmap run with end before start"; return false; }
            if (!runs->empty() && runs->back().second + 1 == a) runs->back().second = b;
            else runs->emplace_back(a, b);
        }
    } else {
        err = "unknown TinyBitmap mode";
        return false;
    }
    return true;
}

// Roaring "portable" serialisation (RoaringFormatSpec; written by Roaring::write in ColorSet.cpp:1199-1212)
bool decode_roaring(const uint8_t *buf, size_t n, std::vector<Run> &runs, std::string &err) {
    Reader r{buf, buf + n};
    const uint32_t cookie = r.get<uint32_t>();
    uint32_t size;
    const bool hasrun = (cookie & 0xFFFF) == 12347;
    if (hasrun) size = (cookie >> 16) + 1;
    else if (cookie == 12346) size = r.get<uint32_t>();
    else { err = "Roaring bitmap cookie not recognised"; return false; }
    if (!r.ok || size > 65536) { err = "Roaring bitmap header damaged"; return false; }
    const uint8_t *run_flags = hasrun ? r.take((size + 7) / 8) : nullptr;
    const uint8_t *keycard = r.take((size_t)size * 4);
    if (!r.ok) { err = "Roaring bitmap truncated"; return false; }
    if (!hasrun || size >= 4) r.take((size_t)size * 4);  // container offsets, not needed for a sequential read
    for (uint32_t k = 0; k < size; ++k) {
        uint16_t key, cm1;
        memcpy(&key, keycard + (size_t)k * 4, 2);
        memcpy(&cm1, keycard + (size_t)k * 4 + 2, 2);
        const uint32_t card = (uint32_t)cm1 + 1;
        const uint64_t hi = (uint64_t)key << 16;
        const bool isrun = hasrun && (run_flags[k / 8] & (1u << (k % 8)));
        if (isrun) {
            const uint16_t n_runs = r.get<uint16_t>();
            const uint8_t *d = r.take((size_t)n_runs * 4);
            if (!d) { err = "Roaring run container truncated"; return false; }
            for (uint32_t i = 0; i < n_runs; ++i) {
                uint16_t s, l;
                memcpy(&s, d + (size_t)i * 4, 2);
                memcpy(&l, d + (size_t)i * 4 + 2, 2);
                const uint64_t a = hi | s, b = hi | (uint64_t)(s + l);
                if (!runs.empty() && runs.back().second + 1 == a) runs.back().second = b;
                else runs.emplace_back(a, b);
            }
        } else if (card > 4096) {
            const uint8_t *d = r.take(8192);
            if (!d) { err = "Roaring bitset container truncated"; return false; }
            for (uint32_t wi = 0; wi < 1024; ++wi) {
                uint64_t word;
                memcpy(&word, d + (size_t)wi * 8, 8);
                while (word) {
                    const int b = __builtin_ctzll(word);
                    // extend over the consecutive ones starting at b
                    uint64_t rest = word >> b;
                    const int len = (~rest == 0) ? 64 - b : __builtin_ctzll(~rest);
                    const uint64_t a = hi | ((uint64_t)wi * 64 + b);
                    if (!runs.empty() && runs.back().second + 1 == a) runs.back().second = a + len - 1;
                    else runs.emplace_back(a, a + len - 1);
                    if (b + len >= 64) word = 0;
                    else word &= ~(((1ull << len) - 1) << b);
                }
            }
        } else {
            const uint8_t *d = r.take((size_t)card * 2);
            if (!d) { err = "Roaring array container truncated"; return false; }
            for (uint32_t i = 0; i < card; ++i) {
                uint16_t v;
                memcpy(&v, d + (size_t)i * 2, 2);
                push_value(runs, hi | v);
            }
        }
    }
    return true;
}

struct Decoded {
    std::vector<Run> ids;   // (colour, k-mer) pair ids: colour * n_kmers + position
    std::vector<Run> full;  // colour ids of the pair form's first member
    bool pair_form = false;
};

// UnitigColors::read (ColorSet.cpp:1228-1283).  runs == nullptr: only skip over the set.
bool decode_set(Reader &r, std::vector<Run> *runs, Decoded *top, int depth, std::string &err) {
    const uint64_t bits = r.get<uint64_t>();
    if (!r.ok) { err = "colour set truncated"; return false; }
    switch (bits & 0x7) {
        case 0x0: return decode_tiny(r, runs, err);
        case 0x1:
            if (runs)
                for (uint64_t v = bits >> 3, i = 0; v; v >>= 1, ++i)
                    if (v & 1) push_value(*runs, i);
            return true;
        case 0x2:
            if (runs) push_value(*runs, bits >> 3);
            return true;
        case 0x3: {
            const uint32_t n = (uint32_t)(bits >> 3);
            const uint8_t *d = r.take(n);
            if (!d) { err = "Roaring bitmap truncated"; return false; }
            return runs ? decode_roaring(d, n, *runs, err) : true;
        }
        case 0x4:
            if (depth > 0) { err = "nested pair-form colour set"; return false; }
            if (top) top->pair_form = true;
            if (!decode_set(r, top ? &top->full : nullptr, nullptr, depth + 1, err)) return false;
            return decode_set(r, runs, nullptr, depth + 1, err);
        case 0x5:
            // A reference to a shared colour set is written as this flag word alone (UnitigColors::write, ColorSet.cpp:1190-1194:
            // `copy_UnitigColors` is false in DataStorage::write) and UnitigColors::read has no case for it (:1228-1283): the file
            // does not say WHICH shared set, the reference is left with flag 5 on a null pointer and dies at the unitig's first
            // colour query.  Skipped over like the reference does; an error once a unitig's colours are asked of it.
            if (runs || top) { err = "colour set is a reference to a shared colour set, which the file format does not resolve (the reference's reader leaves a null pointer there: ColorSet.cpp:1190-1194, 1228-1283)"; return false; }
            return true;
        default:
            err = "unknown colour set encoding";
            return false;
    }
}

bool read_file(const std::string &path, std::vector<uint8_t> &buf) {
    FILE *f = fopen(path.c_str(), "rb");
    if (!f) return false;
    fseek(f, 0, SEEK_END);
    long n = ftell(f);
    fseek(f, 0, SEEK_SET);
    buf.resize((size_t)n);
    bool ok = n == 0 || fread(buf.data(), 1, (size_t)n, f) == (size_t)n;
    fclose(f);
    return ok;
}

struct PairHash {
    size_t operator()(const std::pair<uint64_t, uint64_t> &p) const { return (size_t)(p.first * 0x9E3779B97F4A7C15ull ^ p.second); }
};

}  // namespace

bool ColorSets::contains(uint32_t u, uint32_t colour, uint32_t dist, uint32_t len) const {
    if (full(u, colour)) return true;
    if (!any(u, colour)) return false;
    auto it = partial.find(u);
    if (it == partial.end()) return false;
    for (const Partial &p : it->second) {
        if (p.colour != colour) continue;
        for (uint32_t i = dist; i < dist + len; ++i)
            if (!((p.bits[i >> 6] >> (i & 63)) & 1)) return false;
        return true;
    }
    return false;
}

bool ColorSets::load(const std::string &path, const UnitigSet &g, unsigned threads, std::string &err) {
    std::vector<uint8_t> buf;
    if (!read_file(path, buf)) { err = "DataStorage::read(): Could not open file " + path + " for reading color sets"; return false; }
    Reader r{buf.data(), buf.data() + buf.size()};
    // header, DataStorage.tcc:831-862
    const uint64_t format_version = r.get<uint64_t>(), nb_seeds = r.get<uint64_t>(), nb_colors = r.get<uint64_t>(),
                   nb_cs = r.get<uint64_t>(), sz_cs = r.get<uint64_t>(), sz_shared_cs = r.get<uint64_t>(),
                   overflow_sz = r.get<uint64_t>();
    if (!r.ok) { err = "colour file header truncated"; return false; }
    if (format_version != 1 && format_version != 2) { err = "unsupported colour file format version"; return false; }
    if (nb_seeds >= 256) { err = "DataStorage::read(): Does not support more than 255 hash seeds"; return false; }
    if (nb_colors == 0 || nb_colors > kMaxColors) { err = "number of colours outside 1.." + std::to_string(kMaxColors); return false; }
    if (nb_cs > sz_cs || sz_cs > buf.size() || sz_shared_cs > buf.size()) { err = "colour file header damaged"; return false; }
    std::vector<uint64_t> seeds(nb_seeds);
    for (auto &s : seeds) s = r.get<uint64_t>();
    if (format_version == 2) {
        const uint64_t block_sz = r.get<uint64_t>();
        if (!r.ok || block_sz == 0) { err = "colour file header damaged"; return false; }
        // std::streampos of every block of shared sets, then of colour sets (DataStorage.tcc:887-897): only used by the reference's threaded reader
        const uint64_t n_pos = (sz_shared_cs / block_sz) + ((sz_shared_cs % block_sz) != 0) + (sz_cs / block_sz) + ((sz_cs % block_sz) != 0);
        r.take((size_t)n_pos * 16);
    }
    n_colors = (uint32_t)nb_colors;
    names.clear();
    for (uint32_t i = 0; i < n_colors && r.ok; ++i) {
        const uint8_t *nl = (const uint8_t *)memchr(r.p, '\n', (size_t)(r.end - r.p));
        if (!nl) { err = "colour names truncated"; return false; }
        names.emplace_back((const char *)r.p, (size_t)(nl - r.p));
        r.p = nl + 1;
    }
    r.take((size_t)((sz_cs >> 6) + ((sz_cs & 0x3F) != 0)) * 8);  // unitig_cs_link occupancy bits
    if (!r.ok) { err = "colour file truncated before the colour sets"; return false; }
    // the shared colour sets, each with its reference count (DataStorage.tcc:811-820): read and, like in the reference, linked to nothing
    for (uint64_t i = 0; i < sz_shared_cs; ++i) {
        if (!decode_set(r, nullptr, nullptr, 0, err)) { err += " (shared colour set " + std::to_string(i) + ")"; return false; }
        (void)r.get<uint64_t>();
        if (!r.ok) { err = "shared colour sets truncated"; return false; }
    }
    // offsets of the sz_cs colour sets (variable length)
    std::vector<const uint8_t *> at(sz_cs);
    for (uint64_t i = 0; i < sz_cs; ++i) {
        at[i] = r.p;
        if (!decode_set(r, nullptr, nullptr, 0, err)) { err += " (colour set " + std::to_string(i) + ")"; return false; }
    }
    // overflow table: (head k-mer, unitig length in bases) -> slot, DataStorage.tcc:1027-1035
    std::unordered_map<std::pair<uint64_t, uint64_t>, uint64_t, PairHash> overflow;
    overflow.reserve(overflow_sz * 2);
    for (uint64_t i = 0; i < overflow_sz; ++i) {
        const uint64_t km = r.get<uint64_t>(), sz = r.get<uint64_t>(), pos = r.get<uint64_t>();
        if (!r.ok || pos >= sz_cs) { err = "colour file overflow table damaged"; return false; }
        overflow[{km, sz}] = pos;
    }
    const uint32_t N = g.n();
    if (g.da_tag.size() != N) { err = "ColoredCDBG::read(): One sequence line in GFA file has no DataAccessor tag. Operation aborted."; return false; }
    words = (n_colors + 63) / 64;
    full_mask.assign((size_t)N * words, 0);
    any_mask.assign((size_t)N * words, 0);
    size_total.assign(N, 0);
    n_full_enc.assign(N, 0);
    partial.clear();
    std::mutex mu;
    std::string first_err;
    std::atomic<bool> failed{false};
    const uint8_t *file_end = buf.data() + buf.size();
    const int k = g.k;
    parallel_chunks(N, 4096, threads, [&](size_t, size_t b, size_t e) {
        Decoded d;
        std::vector<uint64_t> cnt(n_colors);
        for (size_t u = b; u < e && !failed.load(std::memory_order_relaxed); ++u) {
            auto fail = [&](const std::string &m) {
                std::lock_guard<std::mutex> lk(mu);
                if (!failed.exchange(true)) first_err = m + " (unitig " + std::to_string(u + 1) + ")";
            };
            const int da = g.da_tag[u];
            if (da < 0) { fail("ColoredCDBG::read(): One sequence line in GFA file has no DataAccessor tag. Operation aborted."); return; }
            // head k-mer, left aligned as Bifrost's Kmer keeps it (bifrost/src/Kmer.cpp:92-107)
            const uint64_t head = (g.words[g.word_off[u]] >> (64 - 2 * k)) << (64 - 2 * k);
            uint64_t slot;
            if (da == 0) {
                auto it = overflow.find({head, (uint64_t)g.len_bp[u]});
                if (it == overflow.end()) { fail("colour set of a unitig not found in the overflow table"); return; }
                slot = it->second;
            } else {
                if ((uint64_t)da > nb_seeds || nb_cs == 0) { fail("DataAccessor tag beyond the file's hash seeds"); return; }
                slot = bifrost_kmer_hash(head, seeds[da - 1]) % nb_cs;
            }
            d.ids.clear();
            d.full.clear();
            d.pair_form = false;
            Reader rr{at[slot], file_end};
            std::string derr;
            if (!decode_set(rr, &d.ids, &d, 0, derr)) { fail(derr); return; }
            const uint64_t km = g.len_km((uint32_t)u);
            std::fill(cnt.begin(), cnt.end(), 0);
            uint64_t total = 0, nf = 0;
            bool bad = false;
            for (const Run &run : d.full)
                for (uint64_t c = run.first; c <= run.second; ++c) {
                    if (c >= n_colors) { bad = true; break; }
                    cnt[c] += km;
                    ++nf;
                }
            for (const Run &run : d.ids) {
                if (run.second >= km * n_colors) { bad = true; break; }
                for (uint64_t c = run.first / km; c <= run.second / km; ++c) {
                    const uint64_t lo = std::max(run.first, c * km), hi = std::min(run.second, c * km + km - 1);
                    cnt[c] += hi - lo + 1;
                }
            }
            if (bad) { fail("colour set refers to a (colour, k-mer) pair outside the unitig: wrong slot or damaged file"); return; }
            uint64_t *fm = &full_mask[u * words], *am = &any_mask[u * words];   // (this thread's unitigs only)
            std::vector<Partial> parts;
            for (uint32_t c = 0; c < n_colors; ++c) {
                total += cnt[c];
                if (cnt[c] > km) { bad = true; break; }
                if (cnt[c] == km) { fm[c >> 6] |= 1ull << (c & 63); am[c >> 6] |= 1ull << (c & 63); }
                else if (cnt[c]) {
                    am[c >> 6] |= 1ull << (c & 63);
                    Partial p;
                    p.colour = c;
                    p.bits.assign((km + 63) / 64, 0);
                    for (const Run &run : d.ids) {
                        const uint64_t lo = std::max(run.first, c * km), hi = std::min(run.second, c * km + km - 1);
                        for (uint64_t v = lo; v <= hi && lo <= hi; ++v) p.bits[(v - c * km) >> 6] |= 1ull << ((v - c * km) & 63);
                    }
                    parts.push_back(std::move(p));
                }
            }
            if (bad) { fail("colour set counts a colour more than once per k-mer"); return; }
            size_total[u] = total;
            n_full_enc[u] = d.pair_form ? (uint32_t)nf : 0;
            if (!parts.empty()) {
                std::lock_guard<std::mutex> lk(mu);
                partial.emplace((uint32_t)u, std::move(parts));
            }
        }
    });
    if (failed) { err = first_err; return false; }
    return true;
}

}  // namespace pfh
