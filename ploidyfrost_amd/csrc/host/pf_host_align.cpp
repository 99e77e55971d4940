#include "pf_host_align.hpp"

#include "pf_parallel.hpp"

#include <algorithm>
#include <chrono>
#include <climits>
#include <cstdlib>
#include <cstring>

namespace pfh {

namespace {

// grow-only pinned host buffer (falls back to pageable memory if pinning fails)
template <typename T>
struct PinnedBuf {
    pf_ctx *ctx = nullptr;
    T *p = nullptr;
    size_t cap = 0;
    bool pinned = false;
    PinnedBuf() = default;
    PinnedBuf(const PinnedBuf &) = delete;
    PinnedBuf &operator=(const PinnedBuf &) = delete;
    ~PinnedBuf() { release(); }
    void release() {
        if (!p) return;
        if (pinned) pf_host_free(ctx, p); else free(p);
        p = nullptr;
        cap = 0;
    }
    void ensure(pf_ctx *c, size_t n) {  // contents are NOT preserved
        if (n <= cap) return;
        release();
        ctx = c;
        const size_t want = n + n / 4 + 64;
        void *q = nullptr;
        if (pf_host_alloc(c, want * sizeof(T), &q) == PF_OK) { p = (T *)q; pinned = true; }
        else { p = (T *)malloc(want * sizeof(T)); pinned = false; }
        cap = want;
    }
    T *data() { return p; }
    const T *data() const { return p; }
    size_t size() const { return cap; }
};

// Raw output of pf_align_batch: kept alignments per job, read in place (no per-hit copies).
struct RawHits {
    PinnedBuf<uint64_t> first;
    PinnedBuf<uint32_t> count;
    PinnedBuf<pf_align_hit> hits;
    PinnedBuf<char> text;
    PinnedBuf<uint32_t> gaps;
    uint32_t n(uint32_t job) const { return count.p[job]; }
    const pf_align_hit &hit(uint32_t job, uint32_t h) const { return hits.p[first.p[job] + h]; }
    std::string a(const pf_align_hit &x) const { return std::string(text.p + x.text_off, x.len); }
    std::string b(const pf_align_hit &x) const { return std::string(text.p + x.text_off + x.len, x.len); }
    const uint32_t *gap_ptr(const pf_align_hit &x) const { return gaps.p + x.gap_off; }
};

class JobBatch {
public:
    void clear() { used_ = 0; jobs_.clear(); }
    void reserve(pf_ctx *ctx, size_t jobs, size_t bytes) {
        jobs_.reserve(jobs);
        grow(ctx, bytes);
    }
    uint32_t add(pf_ctx *ctx, const std::string &a, const std::string &b) {
        if (used_ + a.size() + b.size() > text_.cap) grow(ctx, (used_ + a.size() + b.size()) * 2);
        pf_align_job j;
        j.a_off = used_;
        j.a_len = (uint32_t)a.size();
        memcpy(text_.p + used_, a.data(), a.size());
        used_ += a.size();
        j.b_off = used_;
        j.b_len = (uint32_t)b.size();
        memcpy(text_.p + used_, b.data(), b.size());
        used_ += b.size();
        jobs_.push_back(j);
        return (uint32_t)jobs_.size() - 1;
    }
    size_t size() const { return jobs_.size(); }

    int run(pf_ctx *ctx, const Scoring &sc, RawHits &out, std::string &err) {
        const uint32_t n = (uint32_t)jobs_.size();
        if (n == 0) return PF_OK;
        out.first.ensure(ctx, n);
        out.count.ensure(ctx, n);
        uint64_t hit_cap = (uint64_t)n * 2 + 64, text_cap = used_ * 3 + 4096, gap_cap = (uint64_t)n * 8 + 1024;
        for (;;) {
            out.hits.ensure(ctx, hit_cap);
            out.text.ensure(ctx, text_cap);
            out.gaps.ensure(ctx, gap_cap);
            uint64_t used[3] = {0, 0, 0};
            int st = pf_align_batch(ctx, text_.p, used_, jobs_.data(), n, sc.match, sc.mismatch, sc.gap, out.first.p, out.count.p,
                                    out.hits.p, out.hits.cap, out.text.p, out.text.cap, out.gaps.p, out.gaps.cap, used);
            if (st == PF_ERR_OVERFLOW && (used[0] > out.hits.cap || used[1] > out.text.cap || used[2] > out.gaps.cap)) {
                hit_cap = std::max<uint64_t>(out.hits.cap, used[0] + used[0] / 8);
                text_cap = std::max<uint64_t>(out.text.cap, used[1] + used[1] / 8);
                gap_cap = std::max<uint64_t>(out.gaps.cap, used[2] + used[2] / 8);
                continue;
            }
            if (st != PF_OK) { err = pf_last_error(ctx); return st; }
            break;
        }
        return PF_OK;
    }

private:
    void grow(pf_ctx *ctx, size_t bytes) {
        if (bytes <= text_.cap) return;
        PinnedBuf<char> bigger;
        bigger.ensure(ctx, bytes);
        if (used_) memcpy(bigger.p, text_.p, used_);
        std::swap(text_.p, bigger.p);
        std::swap(text_.cap, bigger.cap);
        std::swap(text_.pinned, bigger.pinned);
        std::swap(text_.ctx, bigger.ctx);
    }
    PinnedBuf<char> text_;
    size_t used_ = 0;
    std::vector<pf_align_job> jobs_;
};

// ---- row scoring: SeqAlign::variantAnalyze (src/SeqAlign.cpp:237-305) --------------------
struct RowScore {
    long score = 0;
    uint32_t n_pos = 0, indel = 0;
};

RowScore score_pair(const Scoring &sc, const std::string &x, const std::string &y) {
    RowScore r;
    int gap_side = 0;
    const size_t L = x.size();
    for (size_t i = 0; i < L; ++i) {
        const char a = x[i], b = y[i];
        const double s = (a == '-' || b == '-') ? sc.gap : (a == b ? sc.match : sc.mismatch);
        r.score = (long)(r.score + s);
        if (a == b) { gap_side = 0; continue; }
        const int side = a == '-' ? 1 : (b == '-' ? 2 : 0);
        if (side == 0) { r.n_pos++; gap_side = 0; }
        else if (gap_side != side) { gap_side = side; r.indel++; r.n_pos++; }
    }
    return r;
}

// AlignUnit::operator- (src/SeqAlign.hpp:43-67), truncated to int as its callers do
int rank_diff(const RowScore &l, const RowScore &r) {
    long d;
    if (l.score != r.score) d = l.score > r.score ? 1 : -1;
    else if (l.n_pos != r.n_pos) d = (long)r.n_pos - (long)l.n_pos;
    else d = (long)r.indel - (long)l.indel;
    return (int)d;
}

// an older row with the gaps of a new pairwise alignment re-opened (src/SeqAlign.cpp:583-597)
std::string reopen_gaps(const std::string &row, const uint32_t *gaps, uint32_t n_gaps) {
    if (n_gaps == 0) return row;
    std::string out;
    out.reserve(row.size() + n_gaps);
    uint32_t from = 0;
    for (size_t s = n_gaps; s-- > 0;) {
        out.append(row, from, gaps[s] - from);
        out.push_back('-');
        from = gaps[s];
    }
    out.append(row, from, std::string::npos);
    return out;
}

// ---- compareStrPair (src/SeqAlign.cpp:8-236) ----------------------------------------------
struct Sites {
    std::vector<uint32_t> snp_pos, indel_pos, indel_len, merged;
    std::vector<uint16_t> group;  // cols x R
    uint8_t snp = 0, indel = 0;   // 8-bit counters, as in the reference
};

void classify_columns(const std::vector<std::string> &rows, Sites &s) {
    const size_t R = rows.size(), cols = rows.back().size();
    s.group.assign(cols * R, 0);
    bool open = false;
    auto label = [&](size_t j) {
        uint16_t next = 0;
        uint16_t *g = &s.group[j * R];
        for (size_t a = 0; a < R; ++a) {
            size_t b = 0;
            while (b < a && rows[b][j] != rows[a][j]) ++b;
            g[a] = b < a ? g[b] : ++next;
        }
    };
    for (size_t j = 0; j < cols; ++j) {
        // distinct characters of the column, and whether a gap is among them
        char seen[8];
        int n_seen = 0;
        bool has_gap = false;
        for (size_t a = 0; a < R; ++a) {
            const char c = rows[a][j];
            int q = 0;
            while (q < n_seen && seen[q] != c) ++q;
            if (q == n_seen && n_seen < 8) seen[n_seen++] = c;
            has_gap |= c == '-';
        }
        if (n_seen <= 1) {
            if (open) { s.indel_len.push_back((uint32_t)j - s.indel_pos[s.indel - 1]); open = false; }
            continue;
        }
        if (!has_gap) {
            if (open) { s.indel_len.push_back((uint32_t)j - s.indel_pos[s.indel - 1]); open = false; }
            s.snp_pos.push_back((uint32_t)j);
            s.snp++;
            label(j);
            continue;
        }
        bool same_run = open;
        if (open) {
            for (size_t a = 0; a < R && same_run; ++a) same_run = (rows[a][j] == '-') == (rows[a][j - 1] == '-');
            if (!same_run) s.indel_len.push_back((uint32_t)j - s.indel_pos[s.indel - 1]);
        }
        if (!same_run) {
            ++s.indel;
            s.indel_pos.push_back((uint32_t)j);
            open = true;
        }
        if (!same_run || n_seen > 2) label(j);
    }
    s.merged.resize(s.snp_pos.size() + s.indel_pos.size());
    std::merge(s.snp_pos.begin(), s.snp_pos.end(), s.indel_pos.begin(), s.indel_pos.end(), s.merged.begin());
}

// compute_dis (src/SeqAlign.cpp:10-38); L = length of the last row of the last candidate
size_t site_spread(const std::vector<uint32_t> &v, size_t L) {
    if (v.empty()) return 0;
    if (v.size() == 1) {
        const int left = (int)v[0], right = (int)(L - v[0]) - 1;
        return left > right ? (size_t)(left + 1) : (size_t)right;
    }
    size_t d = v[0];
    for (size_t i = 1; i < v.size(); ++i) d = (size_t)std::min((int)(v[i] - v[i - 1] - 1), (int)d);
    return std::min(d, L - v.back() - 1);
}

void choose_alignment(const std::vector<std::vector<std::string>> &cands, Msa &out) {
    out = Msa();
    if (cands.empty()) return;
    const size_t L = cands.back().back().size();
    // running best (src/SeqAlign.cpp:39-46)
    int best_snp = INT_MAX / 2, best_indel = INT_MAX / 2;
    int d_snp = INT_MAX, d_indel = INT_MAX, d_all = INT_MAX;
    int left = -1, right = -1;
    const std::vector<std::string> *best_rows = nullptr;
    Sites best_sites;
    for (const auto &rows : cands) {
        Sites s;
        classify_columns(rows, s);
        enum { KEEP, TAKE, TAKE_TIE } verdict = KEEP;
        size_t c_indel = 0, c_snp = 0, c_all = 0;
        const int total = s.snp + s.indel, best_total = best_snp + best_indel;
        if (total < best_total) verdict = TAKE;
        else if (total == best_total) {
            if (s.indel < best_indel) verdict = TAKE;
            else if (s.indel == best_indel) {
                c_indel = site_spread(s.indel_pos, L);
                if (c_indel > (size_t)d_indel) verdict = TAKE;
                else if (c_indel == (size_t)d_indel) {
                    c_snp = site_spread(s.snp_pos, L);
                    if (c_snp > (size_t)d_snp) verdict = TAKE;
                    else if (c_snp == (size_t)d_snp) {
                        c_all = site_spread(s.merged, L);
                        if (c_all > (size_t)d_all) verdict = TAKE;
                        else if (c_all == (size_t)d_all) {
                            const int l = s.merged.empty() ? 0 : (int)s.merged.front();
                            const int r = s.merged.empty() ? 0 : (int)s.merged.back();
                            if (l > left || r > right) verdict = TAKE;
                            else if (l == left && r == right && best_rows) {
                                for (size_t m = 0; m < rows.size(); ++m)
                                    if (strcmp(rows[m].c_str(), (*best_rows)[m].c_str()) > 0) { verdict = TAKE_TIE; break; }
                                if (verdict == TAKE_TIE) { left = l; right = r; }
                            }
                        }
                    }
                }
            }
        }
        if (verdict == KEEP) continue;
        if (verdict == TAKE) {
            left = std::max(left, s.merged.empty() ? -1 : (int)s.merged.front());
            right = std::max(right, s.merged.empty() ? -1 : (int)s.merged.back());
            c_all = site_spread(s.merged, L);
            c_snp = site_spread(s.snp_pos, L);
            c_indel = site_spread(s.indel_pos, L);
        }
        d_all = (int)c_all;
        d_snp = (int)c_snp;
        d_indel = (int)c_indel;
        best_snp = s.snp;
        best_indel = s.indel;
        best_rows = &rows;
        best_sites = std::move(s);
    }
    if (!best_rows) return;
    out.rows = *best_rows;
    out.n_cols = (uint32_t)out.rows.back().size();
    out.snp_pos = std::move(best_sites.snp_pos);
    out.indel_pos = std::move(best_sites.indel_pos);
    out.indel_len = std::move(best_sites.indel_len);
    out.group = std::move(best_sites.group);
}

}  // namespace

struct Aligner::Impl {
    RawHits raw;
    JobBatch jb;
};

Aligner::Aligner(pf_ctx *ctx) : impl_(new Impl()), ctx_(ctx) {}
Aligner::~Aligner() { delete impl_; }

int Aligner::align(const Scoring &sc, const std::vector<std::vector<std::string>> &paths, std::vector<Msa> &out,
                   AlignStats *stats, std::string &err, unsigned threads) {
    pf_ctx *ctx = ctx_;
    const size_t T = paths.size();
    constexpr size_t CH = 512;
    if (out.size() != T) {
        parallel_chunks(out.size(), CH, threads, [&](size_t, size_t b, size_t e) {
            for (size_t t = b; t < e; ++t) out[t] = Msa();
        });
        out.resize(T);
    }
    // kept[t] = the alignments (lists of rows) still in play for bubble t
    std::vector<std::vector<std::vector<std::string>>> kept(T);
    std::vector<uint32_t> first_job(T, 0);
    size_t max_rows = 0;
    RawHits &raw = impl_->raw;
    using clk = std::chrono::steady_clock;
    auto lap = [](clk::time_point &t) { auto n = clk::now(); double d = std::chrono::duration<double>(n - t).count(); t = n; return d; };
    AlignStats dummy;
    AlignStats &S = stats ? *stats : dummy;
    clk::time_point tp = clk::now();
    {
        JobBatch &jb = impl_->jb;
        jb.clear();
        size_t bytes = 0;
        for (size_t t = 0; t < T; ++t)
            if (paths[t].size() >= 2) bytes += paths[t][0].size() + paths[t][1].size();
        jb.reserve(ctx, T, bytes);
        for (size_t t = 0; t < T; ++t) {
            max_rows = std::max(max_rows, paths[t].size());
            if (paths[t].size() >= 2) first_job[t] = jb.add(ctx, paths[t][0], paths[t][1]);
        }
        S.jobs += jb.size();
        S.rounds++;
        S.build_s += lap(tp);
        int st = jb.run(ctx, sc, raw, err);
        if (st != PF_OK) return st;
        S.device_s += lap(tp);
        parallel_chunks(T, CH, threads, [&](size_t, size_t b, size_t e) {
            for (size_t t = b; t < e; ++t) {
                if (paths[t].size() < 2) continue;
                const uint32_t j = first_job[t];
                kept[t].reserve(raw.n(j));
                for (uint32_t h = 0; h < raw.n(j); ++h) {
                    const pf_align_hit &x = raw.hit(j, h);
                    kept[t].push_back({raw.a(x), raw.b(x)});
                }
            }
        });
        S.post_s += lap(tp);
    }
    // progressive rounds: row i against row 0 of every kept alignment (src/SeqAlign.cpp:559-638)
    for (size_t i = 2; i < max_rows; ++i) {
        JobBatch &jb = impl_->jb;
        jb.clear();
        for (size_t t = 0; t < T; ++t) {
            if (paths[t].size() <= i || kept[t].empty()) continue;
            first_job[t] = (uint32_t)jb.size();
            for (const auto &rows : kept[t]) jb.add(ctx, rows[0], paths[t][i]);
        }
        if (jb.size() == 0) continue;
        S.jobs += jb.size();
        S.rounds++;
        S.build_s += lap(tp);
        int st = jb.run(ctx, sc, raw, err);
        if (st != PF_OK) return st;
        S.device_s += lap(tp);
        parallel_chunks(T, CH, threads, [&](size_t, size_t tb, size_t te) {
            for (size_t t = tb; t < te; ++t) {
                if (paths[t].size() <= i || kept[t].empty()) continue;
                std::vector<std::vector<std::string>> prev;
                prev.swap(kept[t]);
                int best_total = INT_MIN;
                for (size_t kk = 0; kk < prev.size(); ++kk) {
                    const uint32_t job = first_job[t] + (uint32_t)kk;
                    const uint32_t nc = raw.n(job);
                    std::vector<std::vector<std::string>> built(nc);
                    std::vector<std::string> cand_b(nc);
                    std::vector<int> alive(nc);
                    for (uint32_t c = 0; c < nc; ++c) {
                        const pf_align_hit &x = raw.hit(job, c);
                        alive[c] = (int)c;
                        built[c].push_back(raw.a(x));
                        cand_b[c] = raw.b(x);
                    }
                    uint32_t total = 0;  // int in the reference; sums of INT_MIN wrap
                    for (size_t j = 1; j < i; ++j) {
                        RowScore top;
                        top.score = INT_MIN;
                        int best_j = INT_MIN;
                        std::vector<int> next;
                        for (int c : alive) {
                            const pf_align_hit &x = raw.hit(job, (uint32_t)c);
                            std::string re = reopen_gaps(prev[kk][j], raw.gap_ptr(x), x.n_gaps);
                            RowScore rs = score_pair(sc, re, cand_b[c]);
                            const int diff = rank_diff(rs, top);
                            if (diff > 0) { top = rs; next.clear(); }
                            if (diff >= 0) {
                                best_j = (int)top.score;
                                next.push_back(c);
                                built[c].push_back(std::move(re));
                            }
                        }
                        alive.swap(next);
                        total += (uint32_t)best_j;
                    }
                    const int tk = (int)total;
                    if (tk > best_total) { best_total = tk; kept[t].clear(); }
                    if (tk >= best_total)
                        for (int c : alive) {
                            built[c].push_back(std::move(cand_b[c]));
                            kept[t].push_back(std::move(built[c]));
                        }
                }
            }
        });
    }
    S.post_s += lap(tp);
    parallel_chunks(T, CH, threads, [&](size_t, size_t b, size_t e) {
        for (size_t t = b; t < e; ++t) {
            choose_alignment(kept[t], out[t]);
            std::vector<std::vector<std::string>>().swap(kept[t]);  // free here, in parallel
        }
    });
    S.choose_s += lap(tp);
    return PF_OK;
}

}  // namespace pfh
