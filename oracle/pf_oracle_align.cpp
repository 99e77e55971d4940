// ORACLE (test infrastructure only -- see pf_oracle.h).  CPU restatement of SeqAlign.
//
// The quirks that decide byte parity are kept on purpose:
//   * every score is `int = long + double` (truncation) and a direction that continues the
//     predecessor's own direction earns +1                      (SeqAlign.cpp:512-526)
//   * the look-ahead `A[i] == '-'` that forbids a Left move      (:528-532)
//   * all directions equal to the maximum are flagged            (:534-545)
//   * traceback enumerates every co-optimal path, Left then Up then LeftUp, with gap-open
//     budgets that start at 5 and shrink to the best found; a refused move clears the flag in
//     the pristine matrix too; the gap-open bookkeeping of row B is asymmetric (:356-474)
//   * progressive alignment of rows 3..N against row 0 of every kept alignment (:559-638)
//   * the seven-step selection ladder of compareStrPair          (:158-233)
#include "pf_oracle_align.hpp"

#include <algorithm>
#include <climits>
#include <cstring>
#include <set>

namespace pfo {

namespace {
enum : uint8_t { UP = 1, DIAG = 2, LEFT = 4 };

struct Fill {
    size_t m, n;
    std::vector<long> score;
    std::vector<uint8_t> dir;
    size_t at(size_t i, size_t j) const { return i * (n + 1) + j; }
};

// SeqAlign.cpp:480-547
Fill nw_fill(const Scoring &sc, const std::string &A, const std::string &B) {
    Fill f;
    f.m = A.size();
    f.n = B.size();
    f.score.assign((f.m + 1) * (f.n + 1), 0);
    f.dir.assign((f.m + 1) * (f.n + 1), 0);
    for (size_t i = 1; i <= f.m; ++i) {
        f.score[f.at(i, 0)] = (long)(sc.gap * i);
        f.dir[f.at(i, 0)] = UP;
    }
    for (size_t j = 1; j <= f.n; ++j) {
        f.score[f.at(0, j)] = (long)(sc.gap * j);
        f.dir[f.at(0, j)] = LEFT;
    }
    auto sub = [&](char a, char b) -> double {
        if (a == b) return sc.match;
        if (a == '-' || b == '-') return sc.gap;
        return sc.mismatch;
    };
    for (size_t i = 1; i <= f.m; ++i) {
        for (size_t j = 1; j <= f.n; ++j) {
            int up = (int)(f.score[f.at(i - 1, j)] + sc.gap);
            if (f.dir[f.at(i - 1, j)] & UP) up += 1;
            int dg = (int)(f.score[f.at(i - 1, j - 1)] + sub(A[i - 1], B[j - 1]));
            if (f.dir[f.at(i - 1, j - 1)] & DIAG) dg += 1;
            int lf = (int)(f.score[f.at(i, j - 1)] + sc.gap);
            if (f.dir[f.at(i, j - 1)] & LEFT) lf += 1;
            int best = std::max(std::max(up, dg), lf);
            if (best == lf && i != f.m && A[i] == '-') {
                lf = INT_MIN;
                best = up > dg ? up : dg;
            }
            uint8_t d = 0;
            if (up == best) d |= UP;
            if (dg == best) d |= DIAG;
            if (lf == best) d |= LEFT;
            f.score[f.at(i, j)] = best;
            f.dir[f.at(i, j)] = d;
        }
    }
    return f;
}
}  // namespace

Aln score_rows(const Scoring &sc, const std::string &A, const std::string &B) {
    Aln r;
    r.a = A;
    r.b = B;
    uint8_t run = 0;  // 1: inside a gap of A, 2: inside a gap of B
    for (size_t i = 0; i < A.size(); ++i) {
        double s;
        if (A[i] == '-' || B[i] == '-') s = sc.gap;
        else if (A[i] == B[i]) s = sc.match;
        else s = sc.mismatch;
        r.score = (long)(r.score + s);
        if (A[i] != B[i]) {
            if (A[i] == '-') {
                if (run != 1) { run = 1; r.indel++; r.n_pos++; }
            } else if (B[i] == '-') {
                if (run != 2) { run = 2; r.indel++; r.n_pos++; }
            } else {
                run = 0;
                r.n_pos++;
            }
        } else {
            run = 0;
        }
    }
    return r;
}

long aln_minus(const Aln &l, const Aln &r) {
    if (l.score == r.score) {
        if (l.n_pos == r.n_pos) {
            if (l.indel == r.indel) return 0;
            return (long)r.indel - (long)l.indel;
        }
        return (long)((unsigned long)r.n_pos - (unsigned long)l.n_pos);
    }
    return l.score > r.score ? 1 : -1;
}

std::vector<Aln> pairwise_all_optimal(const Scoring &sc, const std::string &A, const std::string &B) {
    Fill f = nw_fill(sc, A, B);
    std::vector<uint8_t> keep = f.dir;  // the by-value `matrix`
    std::vector<uint8_t> work = f.dir;  // `matrix_temp`
    std::vector<Aln> out;
    std::vector<std::pair<size_t, size_t>> stack;
    // resA / resB are built back to front in the reference; here they are kept reversed,
    // so "resX[0]" is the last element.
    std::string ra, rb;
    auto front = [](const std::string &s) -> char { return s.empty() ? '\0' : s.back(); };
    size_t open_a = 0, open_b = 0, lim_a = 5, lim_b = 5;
    std::vector<uint32_t> gap_pos;
    stack.emplace_back(f.m, f.n);
    while (!stack.empty()) {
        const size_t i = stack.back().first, j = stack.back().second;
        const size_t c = f.at(i, j);
        if (i == 0 && j == 0 && open_a <= lim_a && open_b <= lim_b) {
            std::string ta(ra.rbegin(), ra.rend()), tb(rb.rbegin(), rb.rend());
            for (char &ch : ta)
                if (ch == '+') ch = '-';
            Aln au = score_rows(sc, ta, tb);
            au.gap_pos = gap_pos;
            bool take = false;
            if (!out.empty()) {
                int diff = (int)aln_minus(out.back(), au);
                if (diff == 0) take = true;
                else if (diff < 0) { out.clear(); take = true; }
            } else {
                take = true;
            }
            if (take) {
                out.push_back(std::move(au));
                lim_a = open_a;
                lim_b = open_b;
            }
        }
        if (work[c] & LEFT) {
            bool go;
            if (open_a < lim_a) {
                if (ra.empty() || front(ra) != '+') ++open_a;
                go = true;
            } else if (open_a == lim_a) {
                go = front(ra) == '+';
            } else {
                go = false;
            }
            if (!go) {
                keep[c] &= (uint8_t)~LEFT;
                work[c] &= (uint8_t)~LEFT;
                continue;
            }
            stack.emplace_back(i, j - 1);
            ra.push_back('+');
            gap_pos.push_back((uint32_t)i);
            rb.push_back(B[j - 1]);
            work[c] &= (uint8_t)~LEFT;
        } else if (work[c] & UP) {
            bool go;
            if (open_b < lim_b) {
                if (rb.empty() || front(rb) == '-') ++open_b;
                go = true;
            } else if (open_b == lim_b) {
                go = front(rb) == '-';
            } else {
                go = false;
            }
            if (!go) {
                keep[c] &= (uint8_t)~UP;
                work[c] &= (uint8_t)~UP;
                continue;
            }
            stack.emplace_back(i - 1, j);
            ra.push_back(A[i - 1]);
            rb.push_back('-');
            work[c] &= (uint8_t)~UP;
        } else if (work[c] & DIAG) {
            stack.emplace_back(i - 1, j - 1);
            ra.push_back(A[i - 1]);
            rb.push_back(B[j - 1]);
            work[c] &= (uint8_t)~DIAG;
        } else {
            if (ra.empty()) break;
            stack.pop_back();
            work[c] = keep[c];
            const size_t L = ra.size();
            if (ra[L - 1] == '+') {
                if (L >= 2) { if (ra[L - 2] != '+') --open_a; }
                else --open_a;
            }
            if (rb[L - 1] == '-') {
                if (L >= 2) { if (rb[L - 2] != '-') --open_b; }
                else --open_b;
            }
            if (ra[L - 1] == '+') gap_pos.pop_back();
            ra.pop_back();
            rb.pop_back();
        }
    }
    return out;
}

namespace {

struct Picked {
    std::vector<std::string> rows;
    std::vector<uint32_t> snp_pos, indel_pos, indel_len;
    std::vector<std::vector<uint16_t>> num;
};

// compareStrPair, SeqAlign.cpp:8-236
Picked pick_alignment(const std::vector<std::vector<std::string>> &cands) {
    Picked best;
    if (cands.empty()) return best;
    const size_t ref_len = cands.back().back().size();
    auto spread = [&](const std::vector<uint32_t> &v) -> size_t {
        size_t d = 0;
        if (v.empty()) return d;
        if (v.size() == 1) {
            int left = (int)v[0];
            int right = (int)(ref_len - v[0]) - 1;
            d = left > right ? (size_t)(left + 1) : (size_t)right;
        } else {
            d = v[0];
            for (size_t i = 1; i < v.size(); ++i) d = (size_t)std::min((int)(v[i] - v[i - 1] - 1), (int)d);
            d = std::min(d, ref_len - v.back() - 1);
        }
        return d;
    };
    int snp_dis = INT_MAX, indel_dis = INT_MAX, all_dis = INT_MAX;
    int snp_count = INT_MAX / 2, indel_count = INT_MAX / 2;
    int site_l = -1, site_r = -1;
    for (const auto &rows : cands) {
        const size_t R = rows.size();
        const size_t cols = rows.back().size();
        std::vector<uint32_t> snp_pos, indel_pos, indel_len;
        std::vector<std::vector<uint16_t>> num_all;
        bool in_indel = false;
        uint8_t indel = 0, snp = 0;
        auto number_groups = [&](size_t j, std::vector<uint16_t> &num) {
            int next = 0;
            for (size_t a = 0; a < R; ++a) {
                bool same = false;
                for (size_t b = 0; b < a; ++b)
                    if (rows[b][j] == rows[a][j]) { same = true; num[a] = num[b]; break; }
                if (!same) num[a] = (uint16_t)++next;
            }
        };
        for (size_t j = 0; j < cols; ++j) {
            std::set<char> chars;
            std::vector<uint16_t> num(R, 0);
            for (size_t a = 0; a < R; ++a) chars.insert(rows[a][j]);
            if (chars.size() > 1) {
                if (!chars.count('-')) {
                    if (in_indel) { indel_len.push_back((uint32_t)j - indel_pos[indel - 1]); in_indel = false; }
                    snp_pos.push_back((uint32_t)j);
                    snp++;
                    number_groups(j, num);
                } else {
                    bool continues = true;
                    if (in_indel) {
                        for (size_t a = 0; a < R; ++a) {
                            bool g = rows[a][j] == '-', gp = rows[a][j - 1] == '-';
                            if (g != gp) { continues = false; break; }
                        }
                        if (!continues) {
                            indel_len.push_back((uint32_t)j - indel_pos[indel - 1]);
                            ++indel;
                            indel_pos.push_back((uint32_t)j);
                        }
                    } else {
                        continues = false;
                        ++indel;
                        indel_pos.push_back((uint32_t)j);
                        in_indel = true;
                    }
                    if (!continues || chars.size() > 2) number_groups(j, num);
                }
            } else if (in_indel) {
                indel_len.push_back((uint32_t)j - indel_pos[indel - 1]);
                in_indel = false;
            }
            num_all.push_back(std::move(num));
        }
        std::vector<uint32_t> merged(snp_pos.size() + indel_pos.size());
        std::merge(snp_pos.begin(), snp_pos.end(), indel_pos.begin(), indel_pos.end(), merged.begin());
        bool better = false;
        auto adopt = [&]() {
            best.rows = rows;
            best.snp_pos = snp_pos;
            best.indel_pos = indel_pos;
            best.num = num_all;
            best.indel_len = indel_len;
        };
        if (snp + indel < snp_count + indel_count) {
            better = true;
        } else if (snp + indel == snp_count + indel_count) {
            if (indel < indel_count) {
                better = true;
            } else if (indel == indel_count) {
                size_t d_indel = spread(indel_pos);
                if (d_indel > (size_t)indel_dis) {
                    better = true;
                } else if (d_indel == (size_t)indel_dis) {
                    size_t d_snp = spread(snp_pos);
                    if (d_snp > (size_t)snp_dis) {
                        better = true;
                    } else if (d_snp == (size_t)snp_dis) {
                        size_t d_all = spread(merged);
                        if (d_all > (size_t)all_dis) {
                            better = true;
                        } else if (d_all == (size_t)all_dis) {
                            int l = merged.empty() ? 0 : (int)merged[0];
                            int r = merged.empty() ? 0 : (int)merged.back();
                            if (l > site_l || r > site_r) {
                                better = true;
                            } else if (l == site_l && r == site_r) {
                                for (size_t a = 0; a < R; ++a) {
                                    if (strcmp(rows[a].c_str(), best.rows[a].c_str()) > 0) {
                                        all_dis = (int)d_all;
                                        site_l = l;
                                        site_r = r;
                                        snp_count = snp;
                                        indel_count = indel;
                                        snp_dis = (int)d_snp;
                                        indel_dis = (int)d_indel;
                                        adopt();
                                        break;
                                    }
                                }
                            }
                        }
                    }
                }
            }
        }
        if (better) {
            all_dis = (int)spread(merged);
            site_l = std::max(site_l, merged.empty() ? -1 : (int)merged[0]);
            site_r = std::max(site_r, merged.empty() ? -1 : (int)merged.back());
            snp_count = snp;
            indel_count = indel;
            snp_dis = (int)spread(snp_pos);
            indel_dis = (int)spread(indel_pos);
            adopt();
        }
    }
    return best;
}

}  // namespace

AlignResult align_paths(const Scoring &sc, const std::vector<std::string> &strs) {
    AlignResult res;
    if (strs.size() < 2) return res;
    std::vector<std::vector<std::string>> kept;
    for (auto &au : pairwise_all_optimal(sc, strs[0], strs[1])) kept.push_back({au.a, au.b});
    for (size_t i = 2; i < strs.size(); ++i) {
        std::vector<std::vector<std::string>> prev;
        prev.swap(kept);
        int best_total = INT_MIN;
        for (size_t kk = 0; kk < prev.size(); ++kk) {
            uint32_t total_k = 0;  // int in the reference; unsigned here so INT_MIN sums wrap
            std::vector<Aln> cand = pairwise_all_optimal(sc, prev[kk][0], strs[i]);
            std::vector<std::vector<std::string>> built(cand.size());
            std::vector<int> alive;
            for (size_t c = 0; c < cand.size(); ++c) {
                alive.push_back((int)c);
                built[c].push_back(cand[c].a);
            }
            for (size_t j = 1; j < i; ++j) {
                int best_j = INT_MIN;
                Aln top;
                top.score = INT_MIN;
                std::vector<int> alive_j;
                for (int c : alive) {
                    // re-open in row j the gaps that aligning row 0 against strs[i] introduced
                    const std::vector<uint32_t> &gp = cand[c].gap_pos;
                    const std::string &old = prev[kk][j];
                    std::string re;
                    if (!gp.empty()) {
                        uint32_t from = 0;
                        for (size_t s = gp.size(); s-- > 0;) {
                            re += old.substr(from, gp[s] - from);
                            re += '-';
                            from = gp[s];
                        }
                        re += old.substr(from);
                    } else {
                        re = old;
                    }
                    Aln au = score_rows(sc, re, cand[c].b);
                    int diff = (int)aln_minus(au, top);
                    if (diff > 0) {
                        top = au;
                        best_j = (int)top.score;
                        alive_j.clear();
                        alive_j.push_back(c);
                        built[c].push_back(re);
                    } else if (diff == 0) {
                        best_j = (int)top.score;
                        alive_j.push_back(c);
                        built[c].push_back(re);
                    }
                }
                alive = alive_j;
                total_k += (uint32_t)best_j;
            }
            int tk = (int)total_k;
            if (tk > best_total) {
                best_total = tk;
                kept.clear();
            }
            if (tk >= best_total) {
                for (int c : alive) {
                    built[c].push_back(cand[c].b);
                    kept.push_back(built[c]);
                }
            }
        }
    }
    Picked p = pick_alignment(kept);
    res.rows = std::move(p.rows);
    res.snp_pos = std::move(p.snp_pos);
    res.indel_pos = std::move(p.indel_pos);
    res.indel_len = std::move(p.indel_len);
    res.partition = std::move(p.num);
    return res;
}

}  // namespace pfo
