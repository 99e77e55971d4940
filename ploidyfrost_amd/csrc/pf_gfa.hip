// K-GFA: the S-lines of a Bifrost GFA file -> the 2-bit packed unitig set, on the device.
//
// Replaces, for this path, the parse half of CompactedDBG<U>::read (bifrost/src/CompactedDBG.tcc:823-960, 7888-7908 with
// GFA_Parser.cpp:380-520): segment lines of GFA 1 ("S\t<name>\t<sequence>[\ttags]") or GFA 2 ("S\t<name>\t<length>\t<sequence>
// [\ttags]"), any other line skipped, a last line without '\n' dropped (GFA_Parser.cpp:486), "DA:Z:<n>" kept for the colored
// path.  The sequence field is every byte up to the next tab or the line feed, as the reference's getline + strchr take it: the
// carriage return of a CRLF file whose segment line ends with the sequence IS the last byte of that sequence.  Bifrost stores it as
// a base -- A through CompressedSequence::bits in a segment longer than k (CompressedSequence.cpp:231-239, 597-614), T through
// Kmer::set_kmer's bit arithmetic in a k-length one (Kmer.cpp:92-107) -- and never hashes it (a minimizer may not end on the last
// base of a k-mer, minHashIterator.hpp:63-119), so the same graph is built here.  Any other byte that is no base is refused:
// Bifrost would file such a unitig under minimizers hashed from the raw bytes (RepHash.hpp:109-117) that its stored bases do not
// have, and what its look-ups then find is an accident of that index.  Unitig order = Bifrost's iteration order before the abundant
// k-mers are moved (host/pf_host_graph.hpp): segments longer than k in file order, then the k-length ones, each stored as
// min(sequence, reverse complement).
//
//   k_gfa_lines<count / write>  one thread per 2 KiB tile of the file walks the lines that START in its tile (a line start is
//                               byte 0 or the byte after a '\n'), finds the sequence field of every S-line, and counts /
//                               writes (offset, length, DA) -- two passes around an exclusive scan of the per-tile counts,
//                               which is what keeps file order without atomics
//   k_gfa_words                 words per unitig for the scan that gives seq_off
//   k_gfa_pack                  one thread per 64-bit word of the packed sequence (its unitig found by binary search in
//                               seq_off): 32 characters -> 2 bits each, first base most significant; k-length unitigs are
//                               canonicalised in the same thread (one word)
// The packed arrays then go through pf_upload_graph (device pointers), which validates them like any caller's.
#include <hip/hip_runtime.h>

#include <string>

#include "pf_ctx.hpp"
#include "pf_device_common.hpp"
#include "pf_scan.hpp"
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

constexpr uint32_t GFA_TILE = 2048;
constexpr uint32_t ERR_FIELDS = 1, ERR_SHORT = 2, ERR_BASE = 4;

struct GfaState {
    char *text = nullptr;          // the file body (after the header line), padded
    uint64_t n_bytes = 0;
    uint64_t *seg_off = nullptr;   // per unitig: offset of its sequence field in the body
    uint32_t *seg_len = nullptr, *rank = nullptr;
    int32_t *da = nullptr;
    uint8_t *stored_rc = nullptr;
    uint64_t *words = nullptr, *word_off = nullptr;   // the packed graph between pf_gfa_parse and pf_gfa_upload
    uint32_t n = 0, n_short = 0;
    int k = 0;
    bool any_da = false;
    hipStream_t stream = nullptr;   // its own: pf_gfa_parse may run beside other calls on the context (the count table's ingest)
    std::string err;
    void release() {
        for (void *p : {(void *)text, (void *)seg_off, (void *)seg_len, (void *)rank, (void *)da, (void *)stored_rc, (void *)words, (void *)word_off})
            if (p) (void)hipFree(p);
        hipStream_t keep = stream;
        *this = GfaState();
        stream = keep;
    }
};

__device__ inline uint64_t find_byte(const char *__restrict__ t, uint64_t from, uint64_t to, char c) {
    // first position in [from, to) holding c, or `to`
    uint64_t i = from;
    // byte steps up to an 8-byte boundary, then whole words (the body is padded by 64 bytes: reading past `to` is harmless)
    while (i < to && (i & 7)) {
        if (t[i] == c) return i;
        ++i;
    }
    const uint64_t pat = 0x0101010101010101ull * (uint8_t)c;
    while (i < to) {
        const uint64_t x = *reinterpret_cast<const uint64_t *>(t + i) ^ pat;
        const uint64_t z = (x - 0x0101010101010101ull) & ~x & 0x8080808080808080ull;   // a zero byte of x = a match
        if (z) {
            const uint64_t p = i + ((uint64_t)__ffsll((long long)z) - 1) / 8;
            return p < to ? p : to;
        }
        i += 8;
    }
    return to;
}

struct LineSeg {
    uint64_t off;
    uint32_t len;
    int32_t da;
    bool ok, has_da;
};

// the sequence field of the S-line in [q, e)
__device__ inline LineSeg parse_segment(const char *__restrict__ t, uint64_t q, uint64_t e, int version, int k, uint32_t *err) {
    LineSeg s{0, 0, -1, false, false};
    uint64_t fld = q + 2;
    for (int skip = version == 1 ? 1 : 2; skip > 0; --skip) {
        const uint64_t tab = find_byte(t, fld, e, '\t');
        if (tab >= e) { atomicOr(err, ERR_FIELDS); return s; }
        fld = tab + 1;
    }
    const uint64_t end = find_byte(t, fld, e, '\t');
    const uint32_t len = (uint32_t)(end - fld);   // (a '\r' before the line end stays: GFA_Parser.cpp:497-503)
    if ((int)len < k) { atomicOr(err, ERR_SHORT); return s; }
    // optional tags after the sequence: "DA:Z:<n>"
    for (uint64_t tag = end; tag < e;) {
        uint64_t nt = find_byte(t, tag + 1, e, '\t');
        if (nt - tag > 6 && t[tag + 1] == 'D' && t[tag + 2] == 'A' && t[tag + 3] == ':' && t[tag + 4] == 'Z' && t[tag + 5] == ':') {
            // atoi: optional blanks and sign, then digits
            uint64_t p = tag + 6;
            while (p < nt && (t[p] == ' ' || (t[p] >= 9 && t[p] <= 13))) ++p;
            bool neg = false;
            if (p < nt && (t[p] == '-' || t[p] == '+')) { neg = t[p] == '-'; ++p; }
            int v = 0;
            while (p < nt && t[p] >= '0' && t[p] <= '9') { v = v * 10 + (t[p] - '0'); ++p; }
            s.da = (int32_t)(int16_t)(neg ? -v : v);
            s.has_da = true;
        }
        tag = nt;
    }
    s.off = fld;
    s.len = len;
    s.ok = true;
    return s;
}

template <bool WRITE>
__global__ __launch_bounds__(256) void k_gfa_lines(const char *__restrict__ t, uint64_t n, int version, int k, uint64_t n_tiles,
                                                   uint32_t *__restrict__ cnt_long, uint32_t *__restrict__ cnt_short,
                                                   const uint32_t *__restrict__ base_long, const uint32_t *__restrict__ base_short, uint32_t n_long_total,
                                                   uint64_t *__restrict__ seg_off, uint32_t *__restrict__ seg_len, uint32_t *__restrict__ rank,
                                                   int32_t *__restrict__ da, uint32_t *err, uint32_t *any_da) {
    const uint64_t tile = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (tile >= n_tiles) return;
    const uint64_t lo = tile * GFA_TILE, hi = lo + GFA_TILE < n ? lo + GFA_TILE : n;
    // first line start inside the tile
    uint64_t q = lo;
    if (lo > 0) {
        const uint64_t nl = find_byte(t, lo - 1, n, '\n');
        q = nl + 1;   // (n + 1 when there is none: beyond every tile)
    }
    uint32_t nl_ = 0, ns_ = 0;
    while (q < hi) {
        const uint64_t e = find_byte(t, q, n, '\n');
        if (e >= n) break;   // unterminated last line: dropped
        if (e - q >= 2 && t[q] == 'S' && t[q + 1] == '\t') {
            const LineSeg s = parse_segment(t, q, e, version, k, err);
            if (s.ok) {
                const bool is_short = (int)s.len == k;
                if (WRITE) {
                    const uint32_t u = is_short ? n_long_total + base_short[tile] + ns_ : base_long[tile] + nl_;
                    seg_off[u] = s.off;
                    seg_len[u] = s.len;
                    rank[u] = base_long[tile] + base_short[tile] + nl_ + ns_;   // file order among all segments
                    da[u] = s.da;
                    if (s.has_da) *any_da = 1;
                }
                if (is_short) ++ns_; else ++nl_;
            }
        }
        q = e + 1;
    }
    if (!WRITE) {
        cnt_long[tile] = nl_;
        cnt_short[tile] = ns_;
    }
}

__global__ void k_gfa_words(const uint32_t *__restrict__ seg_len, uint32_t n, uint64_t *__restrict__ words) {
    const uint32_t u = blockIdx.x * blockDim.x + threadIdx.x;
    if (u <= n) words[u] = u < n ? (uint64_t)((seg_len[u] + 31) / 32) : 0;
}

__device__ inline int base_code(char c) {
    switch (c) {
        case 'A': case 'a': return 0;
        case 'C': case 'c': return 1;
        case 'G': case 'g': return 2;
        case 'T': case 't': return 3;
        default: return -1;
    }
}

__device__ inline uint64_t revcomp_words(uint64_t x) {
    uint64_t r = __brevll(x);
    r = ((r & 0xAAAAAAAAAAAAAAAAull) >> 1) | ((r & 0x5555555555555555ull) << 1);
    return ~r;
}

__global__ __launch_bounds__(256) void k_gfa_pack(const char *__restrict__ t, const uint64_t *__restrict__ seg_off, const uint32_t *__restrict__ seg_len,
                                                  const uint64_t *__restrict__ word_off, uint32_t n, uint64_t n_words, int k, uint64_t *__restrict__ out,
                                                  uint8_t *__restrict__ stored_rc, uint32_t *err) {
    const uint64_t w = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (w >= n_words) return;
    // the unitig holding word w: the last u with word_off[u] <= w
    uint32_t a = 0, b = n;
    while (b - a > 1) {
        const uint32_t m = a + (b - a) / 2;
        if (word_off[m] <= w) a = m; else b = m;
    }
    const uint32_t u = a, L = seg_len[u];
    const uint32_t j0 = (uint32_t)(w - word_off[u]) * 32;
    const uint32_t m = L - j0 < 32 ? L - j0 : 32;
    const char *s = t + seg_off[u] + j0;
    uint64_t x = 0;
    bool bad = false;
    const bool is_short = (int)L == k;
    for (uint32_t j = 0; j < m; ++j) {
        int c = base_code(s[j]);
        if (c < 0) {
            // the '\r' that ends a CRLF line's sequence: CompressedSequence::bits['\r'] = 0, Kmer::set_kmer('\r') = 3
            if (s[j] == '\r' && j0 + j + 1 == L) c = is_short ? 3 : 0;
            else bad = true;
        }
        x |= (uint64_t)(c & 3) << (62 - 2 * j);
    }
    if (bad) { atomicOr(err, ERR_BASE); return; }
    if (is_short) {   // km.rep() (CompactedDBG.tcc:3945-3954): the smaller of the k-mer and its twin
        const uint64_t rc = revcomp_words(x) << (2 * (32 - k));
        const bool take_rc = rc < x;
        stored_rc[u] = take_rc ? 1 : 0;
        if (take_rc) x = rc;
    } else if (j0 == 0) {
        stored_rc[u] = 0;
    }
    out[w] = x;
}

GfaState *state_of(pf_ctx *ctx) {
    if (!ctx->gfa) ctx->gfa = new GfaState();
    return static_cast<GfaState *>(ctx->gfa);
}

}  // namespace

namespace pf {
void gfa_destroy(pf_ctx *ctx) {
    if (!ctx->gfa) return;
    static_cast<GfaState *>(ctx->gfa)->release();
    if (static_cast<GfaState *>(ctx->gfa)->stream) (void)hipStreamDestroy(static_cast<GfaState *>(ctx->gfa)->stream);
    delete static_cast<GfaState *>(ctx->gfa);
    ctx->gfa = nullptr;
}
}  // namespace pf

extern "C" {

// Errors of pf_gfa_parse are kept in the K-GFA state (pf_gfa_upload reports them): the call may run beside another one on the
// same context, whose error string it must not touch.
#undef PF_HIP
#define PF_HIP(call)                                                                         \
    do {                                                                                     \
        hipError_t e_ = (call);                                                              \
        if (e_ != hipSuccess) {                                                              \
            S->err = std::string(#call) + ": " + hipGetErrorString(e_);                     \
            return PF_ERR_HIP;                                                               \
        }                                                                                    \
    } while (0)

int pf_gfa_parse(pf_ctx *ctx, const char *body, uint64_t n_bytes, int gfa_version, int k, uint32_t *n_unitigs, uint32_t *n_short) {
    if (!ctx || (!body && n_bytes) || !n_unitigs || !n_short || (gfa_version != 1 && gfa_version != 2) || k < 3 || k > 31) return PF_ERR_ARG;
    GfaState *S = state_of(ctx);
    S->release();
    S->err.clear();
    PF_HIP(hipSetDevice(ctx->device));
    if (!S->stream) PF_HIP(hipStreamCreateWithFlags(&S->stream, hipStreamNonBlocking));
    hipStream_t st = S->stream;
    S->n_bytes = n_bytes;
    PF_HIP(hipMalloc(reinterpret_cast<void **>(&S->text), n_bytes + 128));
    PF_HIP(hipMemsetAsync(S->text + n_bytes, 0, 128, st));
    if (n_bytes) PF_HIP(hipMemcpyAsync(S->text, body, n_bytes, hipMemcpyDefault, st));
    const uint64_t n_tiles = (n_bytes + GFA_TILE - 1) / GFA_TILE;
    if (n_tiles >= (1ull << 31)) { S->err = "GFA file larger than 4 TiB"; S->release(); return PF_ERR_ARG; }
    DevTmp<uint32_t> cl_, cs_, bl_, bs_, small_;
    PF_HIP(cl_.alloc((n_tiles + 1) * 4));
    PF_HIP(cs_.alloc((n_tiles + 1) * 4));
    PF_HIP(bl_.alloc((n_tiles + 1) * 4));
    PF_HIP(bs_.alloc((n_tiles + 1) * 4));
    PF_HIP(small_.alloc(16));
    PF_HIP(hipMemsetAsync(small_.p, 0, 16, st));
    PF_HIP(hipMemsetAsync(cl_.p + n_tiles, 0, 4, st));
    PF_HIP(hipMemsetAsync(cs_.p + n_tiles, 0, 4, st));
    uint32_t *d_err = small_.p, *d_any_da = small_.p + 1;
    const unsigned grid = (unsigned)((n_tiles + 255) / 256);
    if (n_tiles)
        k_gfa_lines<false><<<grid, 256, 0, st>>>(S->text, n_bytes, gfa_version, k, n_tiles, cl_.p, cs_.p, nullptr, nullptr, 0, nullptr, nullptr, nullptr, nullptr,
                                                 d_err, d_any_da);
    PF_HIP(hipGetLastError());
    DevTmp<uint8_t> tmp_;
    PF_HIP(tmp_.alloc(scan_scratch_bytes(n_tiles + 1)));
    PF_HIP(scan_exclusive_u32(cl_.p, bl_.p, n_tiles + 1, tmp_.p, st));
    PF_HIP(scan_exclusive_u32(cs_.p, bs_.p, n_tiles + 1, tmp_.p, st));
    uint32_t n_long = 0, n_sh = 0, err = 0;
    PF_HIP(hipMemcpyAsync(&n_long, bl_.p + n_tiles, 4, hipMemcpyDeviceToHost, st));
    PF_HIP(hipMemcpyAsync(&n_sh, bs_.p + n_tiles, 4, hipMemcpyDeviceToHost, st));
    PF_HIP(hipMemcpyAsync(&err, d_err, 4, hipMemcpyDeviceToHost, st));
    PF_HIP(hipStreamSynchronize(st));
    auto refuse = [&](const char *msg) { const std::string m = msg; S->release(); S->err = m; return PF_ERR_ARG; };
    if (err & ERR_FIELDS) return refuse("missing fields in a segment line");
    if (err & ERR_SHORT) return refuse("segment shorter than k");
    const uint64_t N = (uint64_t)n_long + n_sh;
    if (N == 0) return refuse("no segments in the GFA file");
    if (N >= (1u << 30)) return refuse("more than 2^30 unitigs");
    PF_HIP(hipMalloc(reinterpret_cast<void **>(&S->seg_off), N * 8));
    PF_HIP(hipMalloc(reinterpret_cast<void **>(&S->seg_len), (N + 1) * 4));
    PF_HIP(hipMalloc(reinterpret_cast<void **>(&S->rank), N * 4));
    PF_HIP(hipMalloc(reinterpret_cast<void **>(&S->da), N * 4));
    PF_HIP(hipMalloc(reinterpret_cast<void **>(&S->stored_rc), N));
    k_gfa_lines<true><<<grid, 256, 0, st>>>(S->text, n_bytes, gfa_version, k, n_tiles, nullptr, nullptr, bl_.p, bs_.p, n_long, S->seg_off, S->seg_len, S->rank,
                                            S->da, d_err, d_any_da);
    PF_HIP(hipGetLastError());
    // word offsets, then the packed words
    DevTmp<uint64_t> wc_;
    PF_HIP(wc_.alloc((N + 1) * 8));
    PF_HIP(hipMalloc(reinterpret_cast<void **>(&S->word_off), (N + 1) * 8));
    k_gfa_words<<<(unsigned)((N + 1 + 255) / 256), 256, 0, st>>>(S->seg_len, (uint32_t)N, wc_.p);
    DevTmp<uint8_t> tmp2_;
    PF_HIP(tmp2_.alloc(scan_scratch_bytes(N + 1)));
    PF_HIP(scan_exclusive_u64(wc_.p, S->word_off, N + 1, tmp2_.p, st));
    uint64_t n_words = 0;
    uint32_t any_da = 0;
    PF_HIP(hipMemcpyAsync(&n_words, S->word_off + N, 8, hipMemcpyDeviceToHost, st));
    PF_HIP(hipMemcpyAsync(&any_da, d_any_da, 4, hipMemcpyDeviceToHost, st));
    PF_HIP(hipStreamSynchronize(st));
    PF_HIP(hipMalloc(reinterpret_cast<void **>(&S->words), (n_words + 2) * 8));
    PF_HIP(hipMemsetAsync(S->words + n_words, 0, 16, st));
    k_gfa_pack<<<(unsigned)((n_words + 255) / 256), 256, 0, st>>>(S->text, S->seg_off, S->seg_len, S->word_off, (uint32_t)N, n_words, k, S->words, S->stored_rc, d_err);
    PF_HIP(hipGetLastError());
    PF_HIP(hipMemcpyAsync(&err, d_err, 4, hipMemcpyDeviceToHost, st));
    PF_HIP(hipStreamSynchronize(st));
    if (err & ERR_BASE) return refuse("non-ACGT base in a segment");
    S->n = (uint32_t)N;
    S->n_short = n_sh;
    S->k = k;
    S->any_da = any_da != 0;
    // the file itself is no longer needed on the device
    (void)hipFree(S->text);
    S->text = nullptr;
    *n_unitigs = (uint32_t)N;
    *n_short = n_sh;
    return PF_OK;
}

#undef PF_HIP
#define PF_HIP(call)                                                                         \
    do {                                                                                     \
        hipError_t e_ = (call);                                                              \
        if (e_ != hipSuccess) {                                                              \
            pf::CtxErr{ctx} = std::string(#call) + ": " + hipGetErrorString(e_);                   \
            return PF_ERR_HIP;                                                               \
        }                                                                                    \
    } while (0)

const char *pf_gfa_error(const pf_ctx *ctx) {
    return ctx && ctx->gfa ? static_cast<const GfaState *>(ctx->gfa)->err.c_str() : "";
}

int pf_gfa_upload(pf_ctx *ctx) {
    if (!ctx || !ctx->gfa) return PF_ERR_ARG;
    GfaState *S = static_cast<GfaState *>(ctx->gfa);
    if (!S->n || !S->words) { pf::CtxErr{ctx} = S->err.empty() ? "pf_gfa_upload: pf_gfa_parse first" : S->err; return PF_ERR_ARG; }
    const int up = pf_upload_graph(ctx, S->words, S->word_off, S->seg_len, S->n, S->k);
    (void)hipFree(S->words);
    (void)hipFree(S->word_off);
    S->words = S->word_off = nullptr;
    return up;
}

int pf_gfa_ingest(pf_ctx *ctx, const char *body, uint64_t n_bytes, int gfa_version, int k, uint32_t *n_unitigs, uint32_t *n_short) {
    const int st = pf_gfa_parse(ctx, body, n_bytes, gfa_version, k, n_unitigs, n_short);
    if (st != PF_OK) {
        if (ctx && ctx->gfa) pf::CtxErr{ctx} = static_cast<GfaState *>(ctx->gfa)->err;
        return st;
    }
    return pf_gfa_upload(ctx);
}

int pf_gfa_segments(pf_ctx *ctx, uint32_t *len_bp, uint64_t *seq_off, uint32_t *file_rank, int16_t *da_tag, uint8_t *stored_rc, int *any_da) {
    if (!ctx || !ctx->gfa) return PF_ERR_ARG;
    GfaState *S = static_cast<GfaState *>(ctx->gfa);
    if (!S->n) { pf::CtxErr{ctx} = "pf_gfa_segments: pf_gfa_ingest first"; return PF_ERR_ARG; }
    PF_HIP(hipSetDevice(ctx->device));
    const size_t N = S->n;
    if (len_bp) PF_HIP(hipMemcpy(len_bp, S->seg_len, N * 4, hipMemcpyDeviceToHost));
    if (seq_off) PF_HIP(hipMemcpy(seq_off, S->seg_off, N * 8, hipMemcpyDeviceToHost));
    if (file_rank) PF_HIP(hipMemcpy(file_rank, S->rank, N * 4, hipMemcpyDeviceToHost));
    if (stored_rc) PF_HIP(hipMemcpy(stored_rc, S->stored_rc, N, hipMemcpyDeviceToHost));
    if (da_tag) {
        std::vector<int32_t> tmp(N);
        PF_HIP(hipMemcpy(tmp.data(), S->da, N * 4, hipMemcpyDeviceToHost));
        for (size_t i = 0; i < N; ++i) da_tag[i] = (int16_t)tmp[i];
    }
    if (any_da) *any_da = S->any_da ? 1 : 0;
    S->release();   // one fetch per ingest
    return PF_OK;
}

}  // extern "C"
