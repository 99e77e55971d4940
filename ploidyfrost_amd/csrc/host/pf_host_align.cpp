#include "pf_host_align.hpp"

#include <algorithm>
#include <climits>
#include <cstring>

namespace pfh {

namespace {

struct Hit {
    std::string a, b;
    std::vector<uint32_t> gaps;  // rows of A where gaps were opened, traceback order
    long score;
    uint32_t n_pos, n_indel;
};

// ---- one device call --------------------------------------------------------------------
class JobBatch {
public:
    uint32_t add(const std::string &a, const std::string &b) {
        pf_align_job j;
        j.a_off = text_.size();
        j.a_len = (uint32_t)a.size();
        text_ += a;
        j.b_off = text_.size();
        j.b_len = (uint32_t)b.size();
        text_ += b;
        jobs_.push_back(j);
        return (uint32_t)jobs_.size() - 1;
    }
    size_t size() const { return jobs_.size(); }

    int run(pf_ctx *ctx, const Scoring &sc, std::vector<std::vector<Hit>> &per_job, std::string &err) {
        const uint32_t n = (uint32_t)jobs_.size();
        per_job.assign(n, {});
        if (n == 0) return PF_OK;
        uint64_t hit_cap = (uint64_t)n * 2 + 64, text_cap = text_.size() * 3 + 4096, gap_cap = (uint64_t)n * 8 + 1024;
        std::vector<uint64_t> first(n);
        std::vector<uint32_t> count(n);
        for (;;) {
            hits_.resize(hit_cap);
            otext_.resize(text_cap);
            ogaps_.resize(gap_cap);
            uint64_t used[3] = {0, 0, 0};
            int st = pf_align_batch(ctx, text_.data(), text_.size(), jobs_.data(), n, sc.match, sc.mismatch, sc.gap, first.data(),
                                    count.data(), hits_.data(), hit_cap, otext_.data(), text_cap, ogaps_.data(), gap_cap, used);
            if (st == PF_ERR_OVERFLOW && (used[0] > hit_cap || used[1] > text_cap || used[2] > gap_cap)) {
                hit_cap = std::max(hit_cap, used[0] + used[0] / 8);
                text_cap = std::max(text_cap, used[1] + used[1] / 8);
                gap_cap = std::max(gap_cap, used[2] + used[2] / 8);
                continue;
            }
            if (st != PF_OK) { err = pf_last_error(ctx); return st; }
            break;
        }
        for (uint32_t j = 0; j < n; ++j) {
            auto &dst = per_job[j];
            dst.resize(count[j]);
            for (uint32_t h = 0; h < count[j]; ++h) {
                const pf_align_hit &src = hits_[first[j] + h];
                Hit &d = dst[h];
                d.a.assign(otext_.data() + src.text_off, src.len);
                d.b.assign(otext_.data() + src.text_off + src.len, src.len);
                d.gaps.assign(ogaps_.begin() + src.gap_off, ogaps_.begin() + src.gap_off + src.n_gaps);
                d.score = (long)src.score;
                d.n_pos = src.n_pos;
                d.n_indel = src.n_indel;
            }
        }
        return PF_OK;
    }

private:
    std::string text_;
    std::vector<pf_align_job> jobs_;
    std::vector<pf_align_hit> hits_;
    std::vector<char> otext_;
    std::vector<uint32_t> ogaps_;
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
std::string reopen_gaps(const std::string &row, const std::vector<uint32_t> &gaps) {
    if (gaps.empty()) return row;
    std::string out;
    out.reserve(row.size() + gaps.size());
    uint32_t from = 0;
    for (size_t s = gaps.size(); s-- > 0;) {
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

int align_bubbles(pf_ctx *ctx, const Scoring &sc, const std::vector<std::vector<std::string>> &paths, std::vector<Msa> &out,
                  AlignStats *stats, std::string &err) {
    const size_t T = paths.size();
    out.assign(T, Msa());
    // kept[t] = the alignments (lists of rows) still in play for bubble t
    std::vector<std::vector<std::vector<std::string>>> kept(T);
    size_t max_rows = 0;
    std::vector<std::vector<Hit>> hits;
    {
        JobBatch jb;
        for (size_t t = 0; t < T; ++t) {
            max_rows = std::max(max_rows, paths[t].size());
            if (paths[t].size() >= 2) jb.add(paths[t][0], paths[t][1]);
        }
        if (stats) { stats->jobs += jb.size(); stats->rounds++; }
        int st = jb.run(ctx, sc, hits, err);
        if (st != PF_OK) return st;
        size_t j = 0;
        for (size_t t = 0; t < T; ++t) {
            if (paths[t].size() < 2) continue;
            for (Hit &h : hits[j]) kept[t].push_back({std::move(h.a), std::move(h.b)});
            if (stats) stats->hits += hits[j].size();
            ++j;
        }
    }
    // progressive rounds: row i against row 0 of every kept alignment (src/SeqAlign.cpp:559-638)
    std::vector<uint32_t> first_job(T);
    for (size_t i = 2; i < max_rows; ++i) {
        JobBatch jb;
        for (size_t t = 0; t < T; ++t) {
            if (paths[t].size() <= i || kept[t].empty()) continue;
            first_job[t] = (uint32_t)jb.size();
            for (const auto &rows : kept[t]) jb.add(rows[0], paths[t][i]);
        }
        if (jb.size() == 0) continue;
        if (stats) { stats->jobs += jb.size(); stats->rounds++; }
        int st = jb.run(ctx, sc, hits, err);
        if (st != PF_OK) return st;
        for (size_t t = 0; t < T; ++t) {
            if (paths[t].size() <= i || kept[t].empty()) continue;
            std::vector<std::vector<std::string>> prev;
            prev.swap(kept[t]);
            int best_total = INT_MIN;
            for (size_t kk = 0; kk < prev.size(); ++kk) {
                std::vector<Hit> &cand = hits[first_job[t] + kk];
                if (stats) stats->hits += cand.size();
                std::vector<std::vector<std::string>> built(cand.size());
                std::vector<int> alive(cand.size());
                for (size_t c = 0; c < cand.size(); ++c) { alive[c] = (int)c; built[c].push_back(cand[c].a); }
                uint32_t total = 0;  // int in the reference; sums of INT_MIN wrap
                for (size_t j = 1; j < i; ++j) {
                    RowScore top;
                    top.score = INT_MIN;
                    int best_j = INT_MIN;
                    std::vector<int> next;
                    for (int c : alive) {
                        std::string re = reopen_gaps(prev[kk][j], cand[c].gaps);
                        RowScore rs = score_pair(sc, re, cand[c].b);
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
                        built[c].push_back(cand[c].b);
                        kept[t].push_back(std::move(built[c]));
                    }
            }
        }
    }
    for (size_t t = 0; t < T; ++t) choose_alignment(kept[t], out[t]);
    return PF_OK;
}

}  // namespace pfh
