#include "pf_host_graph.hpp"
#include "pf_host_minz.hpp"
#include "pf_parallel.hpp"
#include "pf_trace.hpp"

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <cstdio>
#include <atomic>
#include "ploidyfrost_hip.h"

#include <chrono>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <thread>

namespace pfh {

namespace {
inline int code_of(char c) {
    switch (c) {
        case 'A': case 'a': return 0;
        case 'C': case 'c': return 1;
        case 'G': case 'g': return 2;
        case 'T': case 't': return 3;
        default: return -1;
    }
}
const char kBase[4] = {'A', 'C', 'G', 'T'};
// The byte at the end of a sequence field.  GFA_Parser.cpp:497-520 takes the field up to the tab / line feed, so the '\r' of a
// CRLF file whose segment line ends with the sequence is the sequence's last byte; Bifrost stores it as A in a segment longer
// than k (CompressedSequence::bits, CompressedSequence.cpp:597-614) and as T in a k-length one (Kmer::set_kmer, Kmer.cpp:92-107),
// and no minimizer ever covers it (minHashIterator.hpp:63-119).  Every other byte that is no base: -1 (refused, csrc/pf_gfa.hip).
inline int code_of_last(char c, bool k_length) { return c == '\r' ? (k_length ? 3 : 0) : code_of(c); }

struct Mapped {
    const char *p = nullptr;
    size_t n = 0;
    int fd = -1;
    bool open(const std::string &path) {
        fd = ::open(path.c_str(), O_RDONLY);
        if (fd < 0) return false;
        struct stat st;
        if (fstat(fd, &st) != 0) return false;
        n = (size_t)st.st_size;
        if (n == 0) { p = ""; return true; }
        void *m = mmap(nullptr, n, PROT_READ, MAP_PRIVATE, fd, 0);
        if (m == MAP_FAILED) return false;
        p = (const char *)m;
        return true;
    }
    ~Mapped() {
        if (p && n) munmap((void *)p, n);
        if (fd >= 0) close(fd);
    }
};
}  // namespace

void UnitigSet::append_mapped(uint32_t ov, std::string &dst) const {
    std::string_view s = seq(ov >> 1);
    if ((ov & 1) == 0) {
        dst.append(s.data(), s.size());
    } else {
        size_t at = dst.size();
        dst.resize(at + s.size());
        for (size_t i = 0; i < s.size(); ++i) {
            char c = s[s.size() - 1 - i];
            dst[at + i] = c == 'A' ? 'T' : c == 'C' ? 'G' : c == 'G' ? 'C' : 'A';
        }
    }
}

bool UnitigSet::load_gfa(const std::string &path, std::string &err, bool defer_numbering) {
    LoadTrace trace;
    Mapped f;
    if (!f.open(path)) { err = "cannot open " + path; return false; }
    const char *p = f.p, *end = f.p + f.n;
    auto line_end = [&](const char *q) -> const char * { return (const char *)memchr(q, '\n', (size_t)(end - q)); };
    const char *le = line_end(p);
    if (!le || p == le || *p != 'H') { err = "GFA header line missing"; return false; }
    int version = 1;
    if (le - p >= 10 && memcmp(p, "H\tVN:Z:2.0", 10) == 0) version = 2;
    {
        // tags of the header line (CompactedDBG.tcc:868-884)
        const char *q = p + 2;
        while (q < le) {
            const char *t = (const char *)memchr(q, '\t', (size_t)(le - q));
            if (!t) t = le;
            if (t - q > 5 && memcmp(q, "KL:Z:", 5) == 0) k = atoi(std::string(q + 5, t).c_str());
            else if (t - q > 5 && memcmp(q, "ML:Z:", 5) == 0) g = atoi(std::string(q + 5, t).c_str());
            q = t + 1;
        }
    }
    if (k < 3 || k > 31) { err = "k outside 3..31"; return false; }
    // pass 1: locate the sequence field of every complete S-line -- the file is cut into pieces at line starts, the pieces
    // are scanned side by side and their segment lists joined in file order
    struct Seg { const char *s; uint32_t len; int16_t da; uint32_t rank; };
    const unsigned T = std::min(64u, std::max(1u, std::thread::hardware_concurrency()));
    const char *body = le + 1;
    const size_t body_n = (size_t)(end - body);
    const size_t n_pieces = std::max<size_t>(1, std::min<size_t>((size_t)T * 4, body_n / (1u << 20)));
    std::vector<const char *> cut(n_pieces + 1, end);
    cut[0] = body;
    for (size_t i = 1; i < n_pieces; ++i) {
        const char *c = body + body_n / n_pieces * i;
        const char *nl = (const char *)memchr(c, '\n', (size_t)(end - c));
        cut[i] = nl ? nl + 1 : end;
    }
    struct Piece { std::vector<Seg> segs; std::string err; bool any_da = false; };
    std::vector<Piece> pieces(n_pieces);
    parallel_chunks(n_pieces, 1, T, [&](size_t pi, size_t, size_t) {
        Piece &pc = pieces[pi];
        const char *q = cut[pi];
        const char *stop = cut[pi + 1];
        while (q < stop) {
            const char *e = line_end(q);
            if (!e) break;  // unterminated last line is dropped
            if (e - q >= 2 && q[0] == 'S' && q[1] == '\t') {
                const char *fld = q + 2;
                for (int skip = (version == 1 ? 1 : 2); skip > 0 && fld; --skip) {
                    const char *t = (const char *)memchr(fld, '\t', (size_t)(e - fld));
                    fld = t ? t + 1 : nullptr;
                }
                if (!fld) { pc.err = "missing fields in a segment line"; return; }
                const char *t = (const char *)memchr(fld, '\t', (size_t)(e - fld));
                if (!t) t = e;
                const uint32_t len = (uint32_t)(t - fld);   // (a '\r' before the line feed is part of the field)
                if ((int)len < k) { pc.err = "segment shorter than k"; return; }
                int16_t da = -1;
                for (const char *tag = t; tag < e;) {  // optional tags after the sequence
                    const char *nt = (const char *)memchr(tag + 1, '\t', (size_t)(e - tag - 1));
                    if (!nt) nt = e;
                    if (nt - tag > 6 && memcmp(tag + 1, "DA:Z:", 5) == 0) { da = (int16_t)atoi(std::string(tag + 6, nt).c_str()); pc.any_da = true; }
                    tag = nt;
                }
                pc.segs.push_back({fld, len, da, 0});
            }
            q = e + 1;
        }
    });
    std::vector<Seg> longs, shorts;
    uint64_t long_bp = 0;
    bool any_da = false;
    {
        size_t nl = 0, ns = 0;
        for (const Piece &pc : pieces) {
            if (!pc.err.empty()) { err = pc.err; return false; }
            any_da |= pc.any_da;
            for (const Seg &sg : pc.segs) ((int)sg.len == k ? ns : nl)++;
        }
        longs.reserve(nl);
        shorts.reserve(ns);
        uint32_t rank = 0;
        for (Piece &pc : pieces) {
            for (Seg sg : pc.segs) {
                sg.rank = rank++;
                if ((int)sg.len == k) shorts.push_back(sg);
                else { longs.push_back(sg); long_bp += sg.len; }
            }
            std::vector<Seg>().swap(pc.segs);
        }
    }
    trace.mark("gfa: locate segments");
    const size_t N = longs.size() + shorts.size();
    if (N == 0) { err = "no segments in the GFA file"; return false; }
    if (N >= (1u << 30)) { err = "more than 2^30 unitigs"; return false; }
    n_short = shorts.size();
    text.resize(long_bp + (uint64_t)shorts.size() * k);
    off.resize(N + 1);
    len_bp.resize(N);
    {
        uint64_t at = 0;
        for (size_t i = 0; i < longs.size(); ++i) { off[i] = at; len_bp[i] = longs[i].len; at += longs[i].len; }
        for (size_t i = 0; i < shorts.size(); ++i) { off[longs.size() + i] = at; len_bp[longs.size() + i] = (uint32_t)k; at += (uint64_t)k; }
        off[N] = at;
    }
    std::atomic<bool> bad_base{false};
    parallel_chunks(longs.size(), 4096, T, [&](size_t, size_t b0, size_t b1) {
        for (size_t i = b0; i < b1; ++i) {
            const Seg &sg = longs[i];
            char *dst = text.data() + off[i];
            for (uint32_t j = 0; j < sg.len; ++j) {
                const int c = j + 1 == sg.len ? code_of_last(sg.s[j], false) : code_of(sg.s[j]);
                if (c < 0) { bad_base.store(true, std::memory_order_relaxed); return; }
                dst[j] = kBase[c];
            }
        }
    });
    parallel_chunks(shorts.size(), 16384, T, [&](size_t, size_t b0, size_t b1) {
        char fw[32], rc[32];
        for (size_t i = b0; i < b1; ++i) {
            const Seg &sg = shorts[i];
            for (int j = 0; j < k; ++j) {
                const int c = j + 1 == k ? code_of_last(sg.s[j], true) : code_of(sg.s[j]);
                if (c < 0) { bad_base.store(true, std::memory_order_relaxed); return; }
                fw[j] = kBase[c];
                rc[k - 1 - j] = kBase[3 - c];
            }
            memcpy(text.data() + off[longs.size() + i], memcmp(rc, fw, (size_t)k) < 0 ? rc : fw, (size_t)k);  // km_rep (CompactedDBG.tcc:3945-3954)
        }
    });
    if (bad_base.load()) { err = "non-ACGT base in a segment"; return false; }
    trace.mark("gfa: copy sequences");
    da_tag.clear();
    if (any_da) {
        da_tag.reserve(N);
        for (const Seg &s : longs) da_tag.push_back(s.da);
        for (const Seg &s : shorts) da_tag.push_back(s.da);
    }
    // k-length unitigs Bifrost keeps in its table of abundant k-mers are iterated last, in that table's slot order
    n_abundant = 0;
    numbering_replays = 0;
    file_rank.resize(N);
    for (size_t i = 0; i < longs.size(); ++i) file_rank[i] = longs[i].rank;
    for (size_t i = 0; i < shorts.size(); ++i) file_rank[longs.size() + i] = shorts[i].rank;
    numbering_deferred = true;
    if (!defer_numbering) finish_numbering();
    trace.mark(defer_numbering ? "gfa: numbering deferred" : "gfa: unitig numbering");
    if (numbering_deferred || n_abundant == 0)
    pack();
    trace.mark("gfa: 2-bit pack");
    return true;
}

// ---- device ingest ----------------------------------------------------------------------------------------------------
struct GfaSource {
    Mapped file;
    const char *body = nullptr;
    uint64_t body_n = 0;
    int version = 1;
    std::vector<uint64_t> file_off;   // per unitig: its sequence field inside body
    std::mutex mu;
    std::atomic<bool> text_ready{false};   // set once `text` is complete (copy_seq reads it without the lock)
};

static bool parse_gfa_header(const char *p, const char *end, int &version, int &k, int &g, const char *&body, std::string &err) {
    const char *le = (const char *)memchr(p, '\n', (size_t)(end - p));
    if (!le || p == le || *p != 'H') { err = "GFA header line missing"; return false; }
    version = 1;
    if (le - p >= 10 && memcmp(p, "H\tVN:Z:2.0", 10) == 0) version = 2;
    const char *q = p + 2;   // tags of the header line (CompactedDBG.tcc:868-884)
    while (q < le) {
        const char *t = (const char *)memchr(q, '\t', (size_t)(le - q));
        if (!t) t = le;
        if (t - q > 5 && memcmp(q, "KL:Z:", 5) == 0) k = atoi(std::string(q + 5, t).c_str());
        else if (t - q > 5 && memcmp(q, "ML:Z:", 5) == 0) g = atoi(std::string(q + 5, t).c_str());
        q = t + 1;
    }
    if (k < 3 || k > 31) { err = "k outside 3..31"; return false; }
    body = le + 1;
    return true;
}

bool UnitigSet::open_gfa(const std::string &path, std::string &err) {
    auto s = std::make_shared<GfaSource>();
    if (!s->file.open(path)) { err = "cannot open " + path; return false; }
    const char *body = nullptr;
    if (!parse_gfa_header(s->file.p, s->file.p + s->file.n, s->version, k, g, body, err)) return false;
    s->body = body;
    s->body_n = (uint64_t)(s->file.p + s->file.n - body);
    src_ = s;
    ingested_ = false;
    return true;
}

uint64_t UnitigSet::estimated_unitigs() const {
    if (parsed_on_ && parse_status_ == 0) return parsed_n_;   // (the device has parsed the file already: the count itself)
    return src_ ? src_->body_n / 75 + 1024 : n();
}

int UnitigSet::parse_on_device(pf_ctx *ctx) {
    LoadTrace trace;
    parsed_on_ = ctx;
    parse_status_ = pf_gfa_parse(ctx, src_->body, src_->body_n, src_->version, k, &parsed_n_, &parsed_short_);
    trace.mark("gfa: parse + 2-bit pack on the device");
    return parse_status_;
}

int UnitigSet::ingest_on_device(pf_ctx *ctx, std::string &err) {
    if (parsed_on_ != ctx) parse_on_device(ctx);
    LoadTrace trace;
    if (parse_status_ != PF_OK) { err = pf_gfa_error(ctx); return parse_status_; }
    uint32_t N = parsed_n_, ns = parsed_short_;
    int st = pf_gfa_upload(ctx);
    if (st != PF_OK) { err = pf_last_error(ctx); return st; }
    trace.mark("gfa: packed graph adopted by the context");
    n_short = ns;
    len_bp.resize(N);
    src_->file_off.resize(N);
    file_rank.resize(N);
    std::vector<int16_t> da(N);
    int any_da = 0;
    st = pf_gfa_segments(ctx, len_bp.data(), src_->file_off.data(), file_rank.data(), da.data(), nullptr, &any_da);
    if (st != PF_OK) { err = pf_last_error(ctx); return st; }
    if (any_da) da_tag.swap(da); else da_tag.clear();
    off.resize((size_t)N + 1);
    n_kmers = 0;
    uint64_t at = 0;
    for (size_t u = 0; u < N; ++u) { off[u] = at; at += len_bp[u]; n_kmers += len_bp[u] - (uint32_t)k + 1; }
    off[N] = at;
    text.clear();
    words.clear();
    word_off.clear();
    n_abundant = 0;
    numbering_replays = 0;
    numbering_deferred = true;
    ingested_ = true;
    trace.mark("gfa: segment table on the host");
    return PF_OK;
}

// one sequence as load_gfa leaves it (upper case, a k-length one canonical), from the mapped file
void UnitigSet::text_from_file(uint32_t u, char *dst) const {
    const char *s = src_->body + src_->file_off[u];
    const uint32_t L = len_bp[u];
    if ((int)L == k) {
        char fw[32], rc[32];
        for (int j = 0; j < k; ++j) {
            const int c = j + 1 == k ? code_of_last(s[j], true) : code_of(s[j]);
            fw[j] = kBase[c & 3];
            rc[k - 1 - j] = kBase[3 - (c & 3)];
        }
        memcpy(dst, memcmp(rc, fw, (size_t)k) < 0 ? rc : fw, (size_t)k);
    } else {
        for (uint32_t j = 0; j + 1 < L; ++j) dst[j] = kBase[code_of(s[j]) & 3];
        dst[L - 1] = kBase[code_of_last(s[L - 1], false) & 3];
    }
}

void UnitigSet::copy_seq(uint32_t u, char *dst) const {
    if (src_ && ingested_ && !src_->text_ready.load(std::memory_order_acquire)) text_from_file(u, dst);
    else memcpy(dst, text.data() + off[u], len_bp[u]);
}

// the sequences as load_gfa leaves them (upper case, k-length ones canonical), from the mapped file
void UnitigSet::ensure_text() {
    if (!src_ || !ingested_) return;
    std::lock_guard<std::mutex> lk(src_->mu);
    if (src_->text_ready) return;
    const size_t N = len_bp.size();
    text.resize(off[N]);
    const unsigned T = std::min(64u, std::max(1u, std::thread::hardware_concurrency()));
    parallel_chunks(N, 8192, T, [&](size_t, size_t b0, size_t b1) {
        for (size_t u = b0; u < b1; ++u) text_from_file((uint32_t)u, text.data() + off[u]);
    });
    src_->text_ready.store(true, std::memory_order_release);
}

void UnitigSet::finish_numbering(std::vector<uint8_t> *counters, const uint8_t *dev_counters, const uint8_t *dev_flags, uint64_t dev_counters_len) {
    ensure_text();
    const size_t N = len_bp.size(), n_long = N - (size_t)n_short;
    if (numbering_deferred && n_short && g >= 1 && g <= k - 2 && g <= 31 && file_rank.size() == N) {
        const unsigned T = std::min(64u, std::max(1u, std::thread::hardware_concurrency()));
        std::vector<SegRef> refs(N);
        std::vector<uint32_t> short_of_rank(N, UINT32_MAX);
        for (size_t u = 0; u < N; ++u) {
            refs[file_rank[u]] = SegRef{text.data() + off[u], len_bp[u]};
            if (u >= n_long) short_of_rank[file_rank[u]] = (uint32_t)(u - n_long);
        }
        UnitigNumbering num;
        std::vector<uint8_t> touches_in;
        if (dev_counters && dev_flags) {   // (the device's flags are per unitig in the current order; the replay goes by file order)
            touches_in.resize(N);
            for (size_t u = 0; u < N; ++u) touches_in[file_rank[u]] = dev_flags[u];
        }
        bifrost_numbering(k, g, refs, T, num, counters, touches_in.empty() ? nullptr : dev_counters, touches_in.empty() ? nullptr : touches_in.data(), dev_counters_len);
        numbering_replays = num.replays;
        if (!num.abundant.empty()) {
            n_abundant = num.abundant.size();
            std::vector<uint8_t> moved((size_t)n_short, 0);
            std::vector<uint32_t> order;   // old index (among the k-length unitigs) of the unitig that takes each place
            order.reserve((size_t)n_short);
            for (uint32_t r : num.abundant) moved[short_of_rank[r]] = 1;
            for (uint32_t i = 0; i < (uint32_t)n_short; ++i)
                if (!moved[i]) order.push_back(i);
            for (uint32_t r : num.abundant) order.push_back(short_of_rank[r]);
            const uint64_t base = off[n_long];
            std::vector<char> old(text.begin() + (ptrdiff_t)base, text.end());
            for (size_t i = 0; i < order.size(); ++i) memcpy(text.data() + base + i * (size_t)k, old.data() + (size_t)order[i] * (size_t)k, (size_t)k);
            if (!da_tag.empty()) {
                std::vector<int16_t> od(da_tag.begin() + (ptrdiff_t)n_long, da_tag.end());
                for (size_t i = 0; i < order.size(); ++i) da_tag[n_long + i] = od[order[i]];
            }
            pack();
        }
    }
    numbering_settled();
}

void UnitigSet::from_sequences(const std::vector<std::string> &seqs, int k_) {
    k = k_;
    const size_t N = seqs.size();
    off.assign(N + 1, 0);
    len_bp.resize(N);
    uint64_t tot = 0;
    for (size_t u = 0; u < N; ++u) { off[u] = tot; len_bp[u] = (uint32_t)seqs[u].size(); tot += seqs[u].size(); }
    off[N] = tot;
    text.resize(tot);
    for (size_t u = 0; u < N; ++u) memcpy(text.data() + off[u], seqs[u].data(), seqs[u].size());
    pack();
}

void UnitigSet::pack() {
    const size_t N = len_bp.size();
    word_off.assign(N + 1, 0);
    n_kmers = 0;
    for (size_t u = 0; u < N; ++u) {
        word_off[u + 1] = word_off[u] + (len_bp[u] + 31) / 32;
        n_kmers += len_bp[u] - (uint32_t)k + 1;
    }
    words.resize(word_off[N] + 1);
    words[word_off[N]] = 0;
    const unsigned T = std::min(64u, std::max(1u, std::thread::hardware_concurrency()));
    parallel_chunks(N, 8192, T, [&](size_t, size_t u0, size_t u1) {
        for (size_t u = u0; u < u1; ++u) {
            const char *s = text.data() + off[u];
            uint64_t *w = words.data() + word_off[u];
            const uint32_t L = len_bp[u];
            for (uint32_t i = 0; i < L; i += 32) {
                uint64_t x = 0;
                const uint32_t m = std::min<uint32_t>(32, L - i);
                for (uint32_t j = 0; j < m; ++j) x |= (uint64_t)code_of(s[i + j]) << (62 - 2 * j);
                w[i >> 5] = x;
            }
        }
    });
}

// ------------------------------------------------------------------------------------------
static bool read_all(const std::string &path, std::vector<uint8_t> &buf) {
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

KmcRecords::~KmcRecords() {
    if (map_ && map_n_) munmap(map_, map_n_);
}

bool KmcRecords::load(const std::string &prefix, std::string &err) {
    std::vector<uint8_t> pre;
    if (!read_all(prefix + ".kmc_pre", pre)) { err = "cannot read " + prefix + ".kmc_pre / .kmc_suf"; return false; }
    {
        const std::string path = prefix + ".kmc_suf";
        const int fd = ::open(path.c_str(), O_RDONLY);
        struct stat st;
        if (fd < 0 || fstat(fd, &st) != 0) {
            if (fd >= 0) close(fd);
            err = "cannot read " + prefix + ".kmc_pre / .kmc_suf";
            return false;
        }
        map_n_ = (size_t)st.st_size;
        if (map_n_) {
            map_ = mmap(nullptr, map_n_, PROT_READ, MAP_PRIVATE | MAP_POPULATE, fd, 0);
            if (map_ == MAP_FAILED) { map_ = nullptr; map_n_ = 0; close(fd); err = "cannot map " + path; return false; }
        }
        close(fd);
    }
    const uint8_t *suf = (const uint8_t *)map_;
    const size_t suf_n = map_n_;
    if (pre.size() < 24 || memcmp(pre.data(), "KMCP", 4) != 0 || memcmp(pre.data() + pre.size() - 4, "KMCP", 4) != 0 || suf_n < 8 ||
        memcmp(suf, "KMCS", 4) != 0 || memcmp(suf + suf_n - 4, "KMCS", 4) != 0) {
        err = "not a KMC database (markers)";
        return false;
    }
    uint32_t version;
    memcpy(&version, pre.data() + pre.size() - 12, 4);
    const uint8_t *body = pre.data() + 4;
    auto word = [&](uint64_t i) { uint64_t v; memcpy(&v, body + 8 * i, 8); return v; };
    uint64_t n_entries = 0;
    if (version == 0x200) {
        // KMC2 (kmc_file.cpp:196-245): 'KMCP' | u64 LUT[n_bins*4^p + 1] | u32 signature_map[4^sig + 1] | header | u32 off | 'KMCP'.
        // The lookup of the reference goes signature -> bin -> LUT row -> binary search; to enumerate the
        // records only the LUT is needed: entry b*4^p + x covers the records of bin b whose prefix is x.
        const uint64_t header_offset = pre[pre.size() - 8];
        if (header_offset + 8 > pre.size() || header_offset < 37) { err = "bad KMC2 header offset"; return false; }
        const uint8_t *h = pre.data() + pre.size() - (header_offset + 8);
        auto u32at = [&](size_t o) { uint32_t v; memcpy(&v, h + o, 4); return v; };
        k = u32at(0);
        mode = u32at(4);
        counter_size = u32at(8);
        lut_prefix_len = u32at(12);
        const uint32_t sig_len = u32at(16);
        min_count = u32at(20);
        max_count = u32at(24);
        memcpy(&total, h + 28, 8);
        both_strands = !h[36];
        if (mode != 0) { err = "KMC databases with float counters (mode 1) are not supported"; return false; }
        const uint32_t p = lut_prefix_len;
        if (k < 3 || k > 31 || p == 0 || p >= k || (k - p) % 4 || counter_size == 0 || counter_size > 8 || sig_len < 5 || sig_len > 11) {
            err = "unsupported k / lut_prefix_length / counter_size / signature length in the KMC2 header";
            return false;
        }
        const uint64_t size = pre.size() - 12;  // without markers and the header_offset word
        const uint64_t sig_bytes = ((1ull << (2 * sig_len)) + 1) * 4;
        if (size < sig_bytes + header_offset + 8) { err = "kmc_pre too short for the KMC2 layout"; return false; }
        const uint64_t n_words = (size - sig_bytes - header_offset) / 8;  // n_bins * 4^p + 1
        const uint64_t n_pref = 1ull << (2 * p);
        if (n_words < n_pref + 1 || (n_words - 1) % n_pref) { err = "KMC2 prefix table is not a whole number of bins"; return false; }
        n_entries = n_words - 1;  // the reader patches the last word to total + 1
    } else if (version == 0) {
        // KMC1: 'KMCP' | u64 LUT[...] | header | u32 header_offset | 'KMCP'
        const uint64_t body_sz = pre.size() - 8 - 4;  // without markers and the header_offset word
        const uint64_t header_offset = pre[pre.size() - 8];
        if (header_offset > body_sz || header_offset < 40) { err = "bad KMC header offset"; return false; }
        const uint64_t hi = (body_sz - header_offset) / 8;
        k = (uint32_t)word(hi);
        mode = (uint32_t)(word(hi) >> 32);
        counter_size = (uint32_t)word(hi + 1);
        lut_prefix_len = (uint32_t)(word(hi + 1) >> 32);
        min_count = (uint32_t)word(hi + 2);
        max_count = word(hi + 2) >> 32;
        total = word(hi + 3);
        both_strands = (word(hi + 4) & 0xF) != 1;
        max_count += word(hi + 4) & 0xFFFFFFFF00000000ull;
        if (mode != 0) { err = "KMC databases with float counters (mode 1) are not supported"; return false; }
        const uint32_t p = lut_prefix_len;
        if (k < 3 || k > 31 || p == 0 || p >= k || (k - p) % 4 || counter_size == 0 || counter_size > 8) {
            err = "unsupported k / lut_prefix_length / counter_size in the KMC header";
            return false;
        }
        n_entries = 1ull << (2 * p);
        if (hi < n_entries) { err = "KMC prefix table shorter than 4^p"; return false; }
    } else {
        err = "unknown KMC database version";
        return false;
    }
    suffix_bytes = (k - lut_prefix_len) / 4;
    if (suf_n - 8 < total * (uint64_t)(suffix_bytes + counter_size)) { err = "kmc_suf holds fewer records than total_kmers"; return false; }
    lut.resize(n_entries + 1);
    uint64_t prev = 0;
    for (uint64_t e = 0; e < n_entries; ++e) {
        uint64_t v = word(e);
        if (v > total) v = total;
        if (v < prev || (e == 0 && v != 0)) { err = "KMC prefix table is not ascending from 0"; return false; }
        lut[e] = prev = v;
    }
    lut[n_entries] = total;
    records = suf + 4;
    return true;
}

}  // namespace pfh
