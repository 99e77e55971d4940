// ORACLE (test infrastructure only -- see pf_oracle.h).
#include "pf_oracle_graph.hpp"

#include <algorithm>
#include <cstdio>
#include <cstring>
#include <fstream>

namespace pfo {

std::string revcomp(const std::string &s) {
    std::string r(s.size(), 'N');
    for (size_t i = 0; i < s.size(); ++i) r[s.size() - 1 - i] = comp(s[i]);
    return r;
}

uint64_t pack_kmer(const char *s, int k) {
    uint64_t x = 0;
    for (int i = 0; i < k; ++i) x = (x << 2) | (uint64_t)(base_code(s[i]) & 3);
    return x;
}

uint64_t rc_kmer(uint64_t x, int k) {
    uint64_t r = 0;
    for (int i = 0; i < k; ++i) {
        r = (r << 2) | (3 - (x & 3));
        x >>= 2;
    }
    return r;
}

// ---------------------------------------------------------------------------------------
// KMC reader.  Follows the layout parsed by KMC/kmc_api/kmc_file.cpp:140-179 (markers),
// :185-302 (ReadParamsFrom_prefix_file_buf).  Only what CheckKmer needs is kept.
// ---------------------------------------------------------------------------------------
static bool slurp(const std::string &path, std::vector<uint8_t> &buf) {
    FILE *f = fopen(path.c_str(), "rb");
    if (!f) return false;
    fseek(f, 0, SEEK_END);
    long sz = ftell(f);
    fseek(f, 0, SEEK_SET);
    buf.resize((size_t)sz);
    size_t got = sz ? fread(buf.data(), 1, (size_t)sz, f) : 0;
    fclose(f);
    return got == (size_t)sz;
}

static uint64_t rd64(const uint8_t *p) { uint64_t v; memcpy(&v, p, 8); return v; }
static uint32_t rd32(const uint8_t *p) { uint32_t v; memcpy(&v, p, 4); return v; }

bool KmcDb::load(const std::string &prefix) {
    std::vector<uint8_t> pre, suf;
    if (!slurp(prefix + ".kmc_pre", pre) || !slurp(prefix + ".kmc_suf", suf)) {
        err = "cannot read " + prefix + ".kmc_pre/.kmc_suf";
        return false;
    }
    if (pre.size() < 24 || memcmp(pre.data(), "KMCP", 4) || memcmp(pre.data() + pre.size() - 4, "KMCP", 4) ||
        suf.size() < 8 || memcmp(suf.data(), "KMCS", 4) || memcmp(suf.data() + suf.size() - 4, "KMCS", 4)) {
        err = "bad KMC markers";
        return false;
    }
    version = rd32(pre.data() + pre.size() - 12);  // kmc_file.cpp:188-192
    uint64_t size = pre.size() - 8;                          // without both markers
    const uint8_t *body = pre.data() + 4;
    if (version == 0) {
        // KMC1 (kmc_file.cpp:246-299)
        lut_words = (size - 4) / 8;
        uint64_t header_offset = pre[pre.size() - 8];
        size -= 4;
        uint64_t hi = (size - header_offset) / 8;
        if (hi + 5 > lut_words) { err = "bad KMC1 header offset"; return false; }
        lut.resize(lut_words);
        for (uint64_t i = 0; i < lut_words; ++i) lut[i] = rd64(body + 8 * i);
        uint64_t d = lut[hi];
        k = (uint32_t)d;
        mode = (uint32_t)(d >> 32);
        counter_size = (uint32_t)lut[hi + 1];
        p = (uint32_t)(lut[hi + 1] >> 32);
        min_count = (uint32_t)lut[hi + 2];
        max_count = lut[hi + 2] >> 32;
        total = lut[hi + 3];
        both_strands = !((lut[hi + 4] & 0xF) == 1);
        max_count += lut[hi + 4] & 0xFFFFFFFF00000000ull;
        lut[hi] = total + 1;  // :292
    } else if (version == 0x200) {
        // KMC2 (kmc_file.cpp:196-245): LUT[n_bins*4^p + 1] | signature_map[4^sig_len + 1] | header | offset
        const uint64_t header_offset = pre[pre.size() - 8];
        size -= 4;
        if (header_offset + 8 > pre.size() || header_offset < 37) { err = "bad KMC2 header offset"; return false; }
        const uint8_t *h = pre.data() + pre.size() - (header_offset + 8);
        k = rd32(h);
        mode = rd32(h + 4);
        counter_size = rd32(h + 8);
        p = rd32(h + 12);
        sig_len = rd32(h + 16);
        min_count = rd32(h + 20);
        max_count = rd32(h + 24);
        total = rd64(h + 28);
        both_strands = !h[36];
        if (sig_len < 5 || sig_len > 11) { err = "unsupported KMC2 signature length"; return false; }
        const uint64_t sig_entries = (1ull << (2 * sig_len)) + 1;
        if (size < sig_entries * 4 + header_offset + 8) { err = "kmc_pre too short for the KMC2 layout"; return false; }
        const uint64_t lut_area = size - (sig_entries * 4 + header_offset + 8);
        single_lut = 1ull << (2 * p);
        lut_words = (lut_area + 8) / 8;
        lut.resize(lut_words);
        for (uint64_t i = 0; i < lut_words; ++i) lut[i] = rd64(body + 8 * i);
        lut[lut_area / 8] = total + 1;
        sig_map.resize(sig_entries);
        for (uint64_t i = 0; i < sig_entries; ++i) sig_map[i] = rd32(body + lut_area + 8 + 4 * i);
        // norm table of CMmer (mmer.h:34-87)
        const uint32_t n = 1u << (2 * sig_len);
        auto allowed = [&](uint32_t m) {
            if ((m & 0x3f) == 0x3f || (m & 0x3f) == 0x3b || (m & 0x3c) == 0x3c) return false;
            for (uint32_t j = 0; j + 3 < sig_len; ++j) {
                if ((m & 0xf) == 0) return false;
                m >>= 2;
            }
            if (m == 0 || m == 0x04 || (m & 0xf) == 0) return false;
            return true;
        };
        norm.resize(n);
        for (uint32_t i = 0; i < n; ++i) {
            uint32_t rev = 0, t = i;
            for (uint32_t j = 0; j < sig_len; ++j) { rev |= (3 - (t & 3)) << (2 * (sig_len - 1 - j)); t >>= 2; }
            const uint32_t a = allowed(i) ? i : n, b = allowed(rev) ? rev : n;
            norm[i] = a < b ? a : b;
        }
    } else {
        err = "unsupported KMC version";
        return false;
    }
    if (mode != 0) { err = "KMC quake-mode (float) counters are not supported"; return false; }
    if (k == 0 || k > 31 || p >= k || (k - p) % 4) { err = "unsupported k / lut_prefix_length"; return false; }
    uint32_t sb = (k - p) / 4;
    uint32_t rec = sb + counter_size;
    uint64_t body_sz = suf.size() - 8;
    if (body_sz < total * rec) { err = "kmc_suf shorter than total_kmers records"; return false; }
    suffix.resize(total);
    count.resize(total);
    const uint8_t *r = suf.data() + 4;
    for (uint64_t i = 0; i < total; ++i, r += rec) {
        uint64_t s = 0;
        for (uint32_t b = 0; b < sb; ++b) s = (s << 8) | r[b];
        uint64_t c = 0;
        for (uint32_t b = 0; b < counter_size; ++b) c |= (uint64_t)r[sb + b] << (8 * b);
        suffix[i] = s;
        count[i] = (uint32_t)c;
    }
    loaded = true;
    return true;
}

uint32_t KmcDb::signature(uint64_t kmer) const {
    const uint64_t mask = (1ull << (2 * sig_len)) - 1;
    uint32_t best = 0xFFFFFFFFu;
    for (uint32_t i = 0; i + sig_len <= k; ++i) {
        const uint32_t v = norm[(kmer >> (2 * (k - sig_len - i))) & mask];
        if (v < best) best = v;
    }
    return best;
}

bool KmcDb::check(uint64_t kmer, uint32_t &cnt) const {
    uint64_t prefix = kmer >> (2 * (k - p));
    if (prefix >= lut_words) return false;  // :345-346
    if (version == 0x200) prefix += (uint64_t)sig_map[signature(kmer)] * single_lut;  // :348-356
    if (prefix + 1 >= lut_words) return false;  // (keeps lut[prefix+1] in range)
    int64_t lo = (int64_t)lut[prefix], hi = (int64_t)lut[prefix + 1] - 1;
    if (lo >= (int64_t)total) return false;  // :1385
    // The reference would read one record past the buffer when lut[prefix+1]-1 == total
    // (undefined there); the oracle clamps.
    if (hi >= (int64_t)total) hi = (int64_t)total - 1;
    uint64_t want = kmer & ((1ull << (2 * (k - p))) - 1);
    while (lo <= hi) {
        int64_t mid = (lo + hi) / 2;
        uint64_t s = suffix[mid];
        if (s == want) {
            uint64_t c = count[mid];
            cnt = (uint32_t)c;
            return c >= min_count && c <= max_count;
        }
        if (s < want) lo = mid + 1; else hi = mid - 1;
    }
    return false;
}

bool KmcDb::canonical_count(uint64_t fwd, uint32_t &cnt) const {
    uint32_t c;
    uint64_t q = fwd;
    if (!check(q, c)) q = rc_kmer(fwd, (int)k);
    return check(q, cnt);
}

// ---------------------------------------------------------------------------------------
// GFA loader: unitig order = id order: long unitigs (length > k) in S-line
// order, then short ones (length == k) in S-line order, each stored canonical
// (bifrost/src/CompactedDBG.tcc:3945-3954, 7900-7907; UnitigIterator.tcc:42-56).
// ---------------------------------------------------------------------------------------
bool Graph::load_gfa(const std::string &path) {
    std::ifstream in(path, std::ios::binary);
    if (!in) { err = "cannot open " + path; return false; }
    std::string data((std::istreambuf_iterator<char>(in)), std::istreambuf_iterator<char>());
    size_t pos = 0;
    auto next_line = [&](std::string &line) -> bool {
        // a final line without '\n' is dropped (GFA_Parser.cpp:486)
        size_t e = data.find('\n', pos);
        if (e == std::string::npos) return false;
        line.assign(data, pos, e - pos);
        pos = e + 1;
        return true;
    };
    std::string line;
    if (!next_line(line) || line.empty() || line[0] != 'H') { err = "missing GFA header"; return false; }
    int version = 1;
    if (line.compare(0, 10, "H\tVN:Z:2.0") == 0) version = 2;
    {
        size_t a = 2;
        while (a <= line.size()) {
            size_t b = line.find('\t', a);
            if (b == std::string::npos) b = line.size();
            std::string sub = line.substr(a, b - a);
            if (sub.compare(0, 5, "KL:Z:") == 0) k = atoi(sub.c_str() + 5);
            else if (sub.compare(0, 5, "ML:Z:") == 0) g = atoi(sub.c_str() + 5);
            a = b + 1;
        }
    }
    if (k < 3 || k > 31) { err = "unsupported k"; return false; }
    std::vector<std::string> longs, shorts;
    while (next_line(line)) {
        if (line.size() < 2 || line[0] != 'S' || line[1] != '\t') continue;
        std::vector<std::string> f;
        size_t a = 2;
        while (a <= line.size()) {
            size_t b = line.find('\t', a);
            if (b == std::string::npos) b = line.size();
            f.push_back(line.substr(a, b - a));
            a = b + 1;
        }
        size_t si = (version == 1) ? 1 : 2;
        if (f.size() <= si) { err = "missing fields in segment line"; return false; }
        std::string s = f[si];
        // the field is taken up to the tab / line feed (GFA_Parser.cpp:497-520): the '\r' of a CRLF line that ends with the
        // sequence is its last byte -- stored as A in a segment longer than k (CompressedSequence.cpp:597-614: bits['\r'] = 0),
        // as T in a k-length one (Kmer::set_kmer, Kmer.cpp:92-107: ((c & 4) >> 1) + ... = 3)
        if (!s.empty() && s.back() == '\r') s.back() = (int)s.size() == k ? 'T' : 'A';
        for (auto &c : s) {
            int bc = base_code(c);
            if (bc < 0) { err = "non-ACGT base in segment"; return false; }
            c = "ACGT"[bc];
        }
        if ((int)s.size() < k) { err = "segment shorter than k"; return false; }
        if ((int)s.size() == k) {
            std::string r = revcomp(s);
            shorts.push_back(r < s ? r : s);
        } else {
            longs.push_back(std::move(s));
        }
    }
    seq.clear();
    seq.reserve(longs.size() + shorts.size());
    for (auto &s : longs) seq.push_back(std::move(s));
    for (auto &s : shorts) seq.push_back(std::move(s));
    n_kmers = 0;
    for (auto &s : seq) n_kmers += s.size() - k + 1;
    return true;
}

std::string Graph::mapped(uint32_t ov) const {
    const std::string &s = seq[ov >> 1];
    return (ov & 1) ? revcomp(s) : s;
}

// Neighbours (G2).  find(km, extremities_only=true) (bifrost/src/CompactedDBG.tcc:1403-1523)
// returns the unitig holding km's canonical form at its first or last k-mer position, with
// strand = (km itself, not its twin, is what is stored there).  A table keyed by the
// canonical k-mer of every unitig extremity reproduces it; when two extremities carry the
// same canonical k-mer (invalid compaction) the lowest unitig id wins.
void Graph::build_adjacency() {
    const uint32_t N = n();
    struct Ext { uint32_t u; bool stored_is_canon; };
    std::unordered_map<uint64_t, Ext> tab;
    tab.reserve((size_t)N * 2 + 16);
    auto add = [&](uint32_t u, const char *p) {
        uint64_t x = pack_kmer(p, k), r = rc_kmer(x, k);
        uint64_t c = x < r ? x : r;
        tab.emplace(c, Ext{u, x == c});
    };
    for (uint32_t u = 0; u < N; ++u) {
        const std::string &s = seq[u];
        add(u, s.data());
        if ((int)s.size() > k) add(u, s.data() + s.size() - k);
    }
    auto find = [&](uint64_t q) -> uint32_t {
        uint64_t r = rc_kmer(q, k);
        uint64_t c = q < r ? q : r;
        auto it = tab.find(c);
        if (it == tab.end()) return NONE;
        bool strand = ((q == c) == it->second.stored_is_canon);
        return it->second.u * 2 + (strand ? 0 : 1);
    };
    const uint64_t mask = (k == 32) ? ~0ull : ((1ull << (2 * k)) - 1);
    succ.assign((size_t)N * 8, NONE);
    pred.assign((size_t)N * 8, NONE);
    for (uint32_t u = 0; u < N; ++u) {
        const std::string &s = seq[u];
        uint64_t head = pack_kmer(s.data(), k);
        uint64_t tail = pack_kmer(s.data() + s.size() - k, k);
        for (int st = 0; st < 2; ++st) {
            uint32_t ov = u * 2 + st;
            // NeighborIterator.tcc:16-17
            uint64_t km_head = st == 0 ? head : rc_kmer(tail, k);
            uint64_t km_tail = st == 0 ? tail : rc_kmer(head, k);
            for (uint64_t b = 0; b < 4; ++b) {
                succ[(size_t)ov * 4 + b] = find(((km_tail << 2) | b) & mask);              // forwardBase
                pred[(size_t)ov * 4 + b] = find((km_head >> 2) | (b << (2 * (k - 1))));  // backwardBase
            }
        }
    }
}

int Graph::out_degree(uint32_t ov) const {
    int d = 0;
    for (int b = 0; b < 4; ++b) d += succ[(size_t)ov * 4 + b] != NONE;
    return d;
}
int Graph::in_degree(uint32_t ov) const {
    int d = 0;
    for (int b = 0; b < 4; ++b) d += pred[(size_t)ov * 4 + b] != NONE;
    return d;
}
uint32_t Graph::first_succ(uint32_t ov) const {
    for (int b = 0; b < 4; ++b)
        if (succ[(size_t)ov * 4 + b] != NONE) return succ[(size_t)ov * 4 + b];
    return NONE;
}
uint32_t Graph::first_pred(uint32_t ov) const {
    for (int b = 0; b < 4; ++b)
        if (pred[(size_t)ov * 4 + b] != NONE) return pred[(size_t)ov * 4 + b];
    return NONE;
}

}  // namespace pfo
