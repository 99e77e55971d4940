// ORACLE (test infrastructure only -- see pf_oracle.h).  CPU restatement of the CDBG hot
// path: candidate scan + traversal + state commits (src/CDBG.cpp:178-415, 552-846) and
// variant calling (src/CDBG.cpp:29-120, 417-551, 1101-1705).
#include "pf_oracle.h"

#include <sys/stat.h>

#include <algorithm>
#include <cstring>
#include <fstream>
#include <map>
#include <set>
#include <sstream>
#include <stack>

#include "pf_oracle_align.hpp"
#include "pf_oracle_graph.hpp"

using namespace pfo;

#include "pf_oracle_ctx.hpp"

using namespace pfo_int;

namespace pfo_int {
std::string g_err;

// extractSuperBubble_ptr, CDBG.cpp:253-372 (pure part: topology + ids only)
Traversal traverse(const pfo_ctx &c, uint32_t s) {
    const Graph &g = c.g;
    Traversal t;
    std::vector<uint32_t> todo;
    std::map<uint32_t, uint8_t> state;    // keyed by unitig id: 1 visited, 2 seen
    std::map<uint32_t, bool> strand_of;   // keyed by unitig id
    auto cyc_add = [&](uint32_t ov) {
        if (std::find(t.cyc.begin(), t.cyc.end(), ov) == t.cyc.end()) t.cyc.push_back(ov);
    };
    todo.push_back(s);
    t.seen.push_back(s);
    while (!todo.empty()) {
        uint32_t v = todo.back();
        todo.pop_back();
        state[c.id(v)] = 1;
        strand_of[c.id(v)] = c.strand(v);
        if (g.out_degree(v) == 0) {
            t.flag_tip = true;
        } else {
            for (int b = 0; b < 4; ++b) {
                uint32_t u = g.succ_row(v)[b];
                if (u == NONE) continue;
                if (u == s) {
                    t.flag_cycle = true;
                    cyc_add(s);
                    cyc_add(v);
                    continue;
                }
                auto it = state.find(c.id(u));
                if (it == state.end() || it->second != 1) {
                    if (it == state.end()) {
                        t.seen.push_back(u);
                        strand_of[c.id(u)] = c.strand(u);
                    } else if (strand_of[c.id(u)] != c.strand(u)) {
                        t.flag_cycle = true;
                        cyc_add(u);
                        cyc_add(v);
                    }
                    state[c.id(u)] = 2;
                    bool all_pred = true;
                    for (int pb = 0; pb < 4; ++pb) {
                        uint32_t p = g.pred_row(u)[pb];
                        if (p == NONE) continue;
                        auto ip = state.find(c.id(p));
                        if (ip != state.end()) {
                            if (ip->second != 1) all_pred = false;
                            if (strand_of[c.id(p)] != c.strand(p)) {
                                t.flag_cycle = true;
                                cyc_add(u);
                                cyc_add(p);
                            }
                        } else {
                            all_pred = false;
                        }
                    }
                    if (all_pred) todo.push_back(u);
                } else {
                    t.flag_cycle = true;
                    cyc_add(v);
                    cyc_add(u);
                }
            }
        }
        if (todo.size() == 1) {
            bool clean = true;
            for (uint32_t w : t.seen) {
                if (w != todo[0] && state[c.id(w)] == 2) { clean = false; break; }
            }
            if (clean) {
                t.exit_ov = todo[0];
                bool back_to_s = false;
                for (int b = 0; b < 4; ++b)
                    if (g.succ_row(t.exit_ov)[b] == s) back_to_s = true;
                if (back_to_s) t.outcome = PFO_BFS_CYCLE_EXIT;
                else if (t.flag_cycle || t.flag_tip) t.outcome = PFO_BFS_REJECT;
                else t.outcome = PFO_BFS_ACCEPT;
                return t;
            }
        }
    }
    return t;
}

// setNoBubble_ptr_cycle, CDBG.cpp:552-602
void commit_cycle_exit(pfo_ctx &c, const Traversal &t, uint32_t s) {
    for (uint32_t w : t.seen) c.poison(w >> 1);
    c.set_side_self(s >> 1, c.strand(s));
    c.set_side_self(t.exit_ov >> 1, !c.strand(t.exit_ov));
}

// setNoBubble_ptr(vec, p), CDBG.cpp:603-699
void commit_reject(pfo_ctx &c, const Traversal &t, uint32_t s) {
    auto endpoint = [&](uint32_t d, bool plus_side) {
        uint32_t p = c.side(d, plus_side);
        if (p != 0) c.release_partner(p - 1, d);
        c.set_side_self(d, plus_side);
    };
    endpoint(s >> 1, c.strand(s));
    endpoint(t.exit_ov >> 1, !c.strand(t.exit_ov));
    for (uint32_t w : t.seen) {
        if (w == s || w == t.exit_ov) continue;
        c.poison(w >> 1);
    }
}

// setNoBubble_ptr(p, vec), CDBG.cpp:700-846
void commit_accept(pfo_ctx &c, const Traversal &t, uint32_t s) {
    const Graph &g = c.g;
    const uint32_t e = t.exit_ov;
    const uint32_t ds = s >> 1, de = e >> 1;
    if (t.seen.size() < 4) return;
    if ((c.flags[de] & F_NON_SUPER) || (c.flags[ds] & F_NON_SUPER)) {
        for (uint32_t w : t.seen) {
            if (w == s) { c.set_side_self(ds, c.strand(s)); continue; }
            if (w == e) { c.set_side_self(de, !c.strand(e)); continue; }
            c.poison(w >> 1);
        }
        return;
    }
    if (t.seen.size() <= 6) {
        bool strict = true;
        for (uint32_t w : t.seen) {
            if (w == s || w == e) continue;
            bool ok = g.in_degree(w) == 1 && (g.first_pred(w) >> 1) == ds && g.out_degree(w) == 1 &&
                      (g.first_succ(w) >> 1) == de;
            if (!ok) { strict = false; break; }
        }
        if (strict) {
            c.flags[ds] |= c.strand(s) ? F_STRICT_PLUS : F_STRICT_MINUS;
            c.flags[de] |= (!c.strand(e)) ? F_STRICT_PLUS : F_STRICT_MINUS;
        }
    }
    if (t.seen.size() > c.complex_size) {
        c.flags[ds] |= c.strand(s) ? F_COMPLEX_PLUS : F_COMPLEX_MINUS;
        c.flags[de] |= (!c.strand(e)) ? F_COMPLEX_PLUS : F_COMPLEX_MINUS;
    }
    for (uint32_t w : t.seen) {
        if (w == s || w == e) continue;
        c.poison(w >> 1);
    }
    if (c.strand(s)) { c.plus[ds] = de + 1; c.flags[ds] |= F_PLUS_OPEN; }
    else { c.minus[ds] = de + 1; c.flags[ds] |= F_MINUS_OPEN; }
    if (c.strand(e)) { c.minus[de] = ds + 1; c.flags[de] |= F_MINUS_OPEN; }
    else { c.plus[de] = ds + 1; c.flags[de] |= F_PLUS_OPEN; }
}

// tail of extractSuperBubble_ptr, CDBG.cpp:373-413
void commit_no_exit(pfo_ctx &c, const Traversal &t, uint32_t s) {
    if (!t.flag_cycle) return;
    for (uint32_t w : t.cyc) c.poison(w >> 1);
    c.set_side_self(s >> 1, c.strand(s));
}

void run_candidate(pfo_ctx &c, uint32_t s) {
    Traversal t = traverse(c, s);
    switch (t.outcome) {
        case PFO_BFS_CYCLE_EXIT: c.colored ? commit_cycle_exit_colored(c, t, s) : commit_cycle_exit(c, t, s); break;
        case PFO_BFS_REJECT: commit_reject(c, t, s); break;  // CCDBG.cpp:2662-2758 is CDBG.cpp:603-699 verbatim
        case PFO_BFS_ACCEPT: c.colored ? commit_accept_colored(c, t, s) : commit_accept(c, t, s); break;
        default: commit_no_exit(c, t, s); break;
    }
}

// readCov(const UnitigMap&), CDBG.cpp:66-120.  ov only matters for non-both-strands DBs.
int unitig_cov(const pfo_ctx &c, uint32_t ov, uint64_t &sum, uint32_t &mn) {
    const Graph &g = c.g;
    const int k = g.k;
    sum = 0;
    mn = 10000;
    std::string s = c.db.both_strands ? g.seq[ov >> 1] : g.mapped(ov);
    uint32_t L = g.len_km(ov >> 1);
    for (uint32_t i = 0; i < L; ++i) {
        uint64_t x = pack_kmer(s.data() + i, k);
        uint32_t cnt = 0;
        bool ok = c.db.both_strands ? c.db.canonical_count(x, cnt) : c.db.check(x, cnt);
        if (!ok) return 1;
        sum += cnt;
        if (cnt < mn) mn = cnt;
    }
    return 0;
}

// readCov(const string&, low, up), CDBG.cpp:29-60
int string_cov(const pfo_ctx &c, const std::string &s, uint32_t low, uint32_t up, uint64_t &sum, bool &ok, uint64_t *lost = nullptr) {
    const int k = c.g.k;
    sum = 0;
    ok = true;
    if (!c.db.both_strands) return 0;
    StringProbe probe(c.db, k);
    for (size_t i = 0; i + k <= s.size(); ++i) {
        uint32_t cnt = 0;
        if (!probe.count(s.data() + i, cnt)) { if (lost) *lost = probe.held; return 1; }
        if (cnt > low && cnt < up) sum += cnt;
        else { sum = 0; ok = false; return 0; }
    }
    return 0;
}

// sortSeq_simple, CDBG.cpp:482-551 (descending mean coverage, ties by descending reference string)
void sort_simple(const Graph &g, std::vector<double> &cov, std::vector<uint32_t> &uv, int low, int high) {
    if (high <= low) return;
    int i = low, j = high;
    auto ref = [&](int x) -> const char * { return g.seq[uv[x] >> 1].c_str(); };
    while (true) {
        while (cov[i] >= cov[low]) {
            if (cov[i] > cov[low]) i++;
            else if (strcmp(ref(i), ref(low)) > 0) i++;
            else break;
            if (i == high) break;
        }
        while (cov[j] <= cov[low]) {
            if (cov[j] < cov[low]) j--;
            else if (strcmp(ref(j), ref(low)) < 0) j--;
            else break;
            if (j == low) break;
        }
        if (i >= j) break;
        std::swap(cov[i], cov[j]);
        std::swap(uv[i], uv[j]);
    }
    std::swap(cov[low], cov[j]);
    std::swap(uv[low], uv[j]);
    sort_simple(g, cov, uv, low, j - 1);
    sort_simple(g, cov, uv, j + 1, high);
}

// sortSeq_branching, CDBG.cpp:417-480 (descending length, ties by descending strcmp)
void sort_branching(std::vector<std::string> &v, int low, int high) {
    if (high <= low) return;
    int i = low, j = high;
    while (true) {
        while (v[i].length() >= v[low].length()) {
            if (v[i].length() > v[low].length()) i++;
            else if (strcmp(v[i].c_str(), v[low].c_str()) > 0) i++;
            else break;
            if (i == high) break;
        }
        while (v[j].length() <= v[low].length()) {
            if (v[j].length() < v[low].length()) j--;
            else if (strcmp(v[j].c_str(), v[low].c_str()) < 0) j--;
            else break;
            if (j == low) break;
        }
        if (i >= j) break;
        std::swap(v[i], v[j]);
    }
    std::swap(v[low], v[j]);
    sort_branching(v, low, j - 1);
    sort_branching(v, j + 1, high);
}

std::string strip_gaps(const std::string &s) {
    std::string r;
    for (char ch : s)
        if (ch != '-') r.push_back(ch);
    return r;
}

bool ensure_dir(const std::string &d) {
    struct stat st;
    if (stat(d.c_str(), &st) == 0) return S_ISDIR(st.st_mode);
    return mkdir(d.c_str(), 0777) == 0;
}

}  // namespace pfo_int

// ------------------------------------------------------------------------------------------
// C interface
// ------------------------------------------------------------------------------------------
extern "C" {

const char *pfo_last_error(void) { return g_err.c_str(); }

pfo_ctx *pfo_open(const char *gfa_path, const char *kmc_prefix) {
    pfo_ctx *c = new pfo_ctx();
    if (!c->g.load_gfa(gfa_path)) {
        g_err = c->g.err;
        delete c;
        return nullptr;
    }
    c->g.build_adjacency();
    if (kmc_prefix && *kmc_prefix) {
        if (!c->db.load(kmc_prefix)) {
            g_err = c->db.err;
            delete c;
            return nullptr;
        }
        if ((int)c->db.k != c->g.k) {
            g_err = "k of the KMC database differs from the graph's";
            delete c;
            return nullptr;
        }
    }
    uint32_t N = c->g.n();
    c->flags.assign(N, 0);
    c->plus.assign(N, 0);
    c->minus.assign(N, 0);
    return c;
}

void pfo_close(pfo_ctx *c) { delete c; }
int pfo_k(const pfo_ctx *c) { return c->g.k; }
uint32_t pfo_num_unitigs(const pfo_ctx *c) { return c->g.n(); }
uint64_t pfo_num_kmers(const pfo_ctx *c) { return c->g.n_kmers; }

uint32_t pfo_unitig_seq(const pfo_ctx *c, uint32_t u, char *buf, uint32_t cap) {
    const std::string &s = c->g.seq[u];
    uint32_t n = std::min<uint32_t>(cap, (uint32_t)s.size());
    if (buf && n) memcpy(buf, s.data(), n);
    return (uint32_t)s.size();
}

void pfo_adjacency(const pfo_ctx *c, uint32_t *succ, uint32_t *pred) {
    memcpy(succ, c->g.succ.data(), c->g.succ.size() * 4);
    memcpy(pred, c->g.pred.data(), c->g.pred.size() * 4);
}

int pfo_unitig_cov(const pfo_ctx *c, uint32_t u, uint64_t *sum, uint32_t *min_count) {
    return unitig_cov(*c, u * 2, *sum, *min_count);
}

int pfo_unitig_cov_oriented(const pfo_ctx *c, uint32_t ov, uint64_t *sum, uint32_t *min_count) {
    return unitig_cov(*c, ov, *sum, *min_count);
}

int pfo_string_cov(const pfo_ctx *c, const char *s, uint32_t len, uint32_t low, uint32_t up, uint64_t *sum,
                   int *ok) {
    bool o;
    int r = string_cov(*c, std::string(s, len), low, up, *sum, o);
    *ok = o;
    return r;
}

int pfo_kmer_count(const pfo_ctx *c, const char *kmer, uint32_t *count) {
    return c->db.canonical_count(pack_kmer(kmer, c->g.k), *count) ? 1 : 0;
}

int pfo_extract(const pfo_ctx *c, uint32_t s_ov, uint32_t *exit_ov, uint32_t *n_seen, uint32_t *seen,
                uint32_t *n_cyc, uint32_t *cyc, uint32_t cap, int *flag_cycle, int *flag_tip) {
    Traversal t = traverse(*c, s_ov);
    *exit_ov = t.exit_ov;
    *n_seen = (uint32_t)t.seen.size();
    *n_cyc = (uint32_t)t.cyc.size();
    for (uint32_t i = 0; i < t.seen.size() && i < cap; ++i) seen[i] = t.seen[i];
    for (uint32_t i = 0; i < t.cyc.size() && i < cap; ++i) cyc[i] = t.cyc[i];
    *flag_cycle = t.flag_cycle;
    *flag_tip = t.flag_tip;
    return t.outcome;
}

int pfo_seq_align(double M, double D, double G, const char *const *strs, int n, char *out, uint32_t cap,
                  uint32_t *n_snp, uint32_t *snp_pos, uint32_t *n_indel, uint32_t *indel_pos,
                  uint32_t *n_indel_len, uint32_t *indel_len, uint32_t *n_cols, uint16_t *partition,
                  uint32_t pos_cap, uint32_t part_cap) {
    Scoring sc{M, D, G};
    std::vector<std::string> v;
    for (int i = 0; i < n; ++i) v.emplace_back(strs[i]);
    AlignResult r = align_paths(sc, v);
    std::string joined;
    for (auto &s : r.rows) { joined += s; joined += '\n'; }
    if (out && cap) {
        uint32_t m = std::min<uint32_t>(cap - 1, (uint32_t)joined.size());
        memcpy(out, joined.data(), m);
        out[m] = 0;
    }
    auto put = [&](const std::vector<uint32_t> &src, uint32_t *cnt, uint32_t *dst) {
        if (cnt) *cnt = (uint32_t)src.size();
        if (dst) for (uint32_t i = 0; i < src.size() && i < pos_cap; ++i) dst[i] = src[i];
    };
    put(r.snp_pos, n_snp, snp_pos);
    put(r.indel_pos, n_indel, indel_pos);
    put(r.indel_len, n_indel_len, indel_len);
    if (n_cols) *n_cols = (uint32_t)r.partition.size();
    if (partition) {
        uint32_t w = 0;
        for (auto &col : r.partition)
            for (uint16_t x : col)
                if (w < part_cap) partition[w++] = x;
    }
    return (int)r.rows.size();
}

// needlemanWunch + traceback for one pair, serialised one alignment per line:
//   a_row \t b_row \t score \t n_pos \t indel \t gap_pos,gap_pos,...
int pfo_pairwise(double M, double D, double G, const char *a, const char *b, char *out, uint32_t cap) {
    Scoring sc{M, D, G};
    std::vector<Aln> v = pairwise_all_optimal(sc, a, b);
    std::string joined;
    for (auto &al : v) {
        joined += al.a + "\t" + al.b + "\t" + std::to_string(al.score) + "\t" + std::to_string(al.n_pos) + "\t" +
                  std::to_string(al.indel) + "\t";
        for (size_t i = 0; i < al.gap_pos.size(); ++i) joined += (i ? "," : "") + std::to_string(al.gap_pos[i]);
        joined += "\n";
    }
    if (out && cap) {
        uint32_t m = std::min<uint32_t>(cap - 1, (uint32_t)joined.size());
        memcpy(out, joined.data(), m);
        out[m] = 0;
    }
    return joined.size() + 1 > cap ? -(int)v.size() : (int)v.size();
}

// setUnitigId, CDBG.cpp:121-143
int pfo_set_unitig_id(pfo_ctx *c, const char *outdir, const char *prefix) {
    if (!ensure_dir(outdir)) { g_err = "cannot create output directory"; return 1; }
    std::ofstream o(std::string(outdir) + "/" + prefix + "_Unitig_Id.txt");
    if (!o) { g_err = "cannot open Unitig_Id file"; return 1; }
    for (uint32_t u = 0; u < c->g.n(); ++u) o << (u + 1) << "\t" << c->g.seq[u] << "\n";
    return 0;
}

// findSuperBubble_ptr, CDBG.cpp:178-252
int pfo_find_superbubbles(pfo_ctx *c, const char *outdir, const char *prefix, uint32_t complex_size,
                          uint64_t *n_bubbles) {
    c->complex_size = complex_size;
    const uint32_t N = c->g.n();
    c->flags.assign(N, 0);
    c->plus.assign(N, 0);
    c->minus.assign(N, 0);
    for (uint32_t u = 0; u < N; ++u) {
        if (c->g.out_degree(2 * u) > 1 && c->plus[u] == 0) run_candidate(*c, 2 * u);
        if (c->g.out_degree(2 * u + 1) > 1 && c->minus[u] == 0) run_candidate(*c, 2 * u + 1);
    }
    uint64_t nb = 0;
    if (outdir) {
        if (!ensure_dir(outdir)) { g_err = "cannot create output directory"; return 1; }
        std::ofstream o(std::string(outdir) + "/" + prefix + "_super_bubble.txt", std::ios::out | std::ios::trunc);
        if (!o) { g_err = "cannot open super_bubble file"; return 1; }
        o << "BubbleId\tEntrance\tStrand\tExit\tisSimple\tisComplex\n";
        for (uint32_t u = 0; u < N; ++u) {
            uint8_t f = c->flags[u];
            if ((f & 3) == 0) continue;
            if (c->colored) {
                // CCDBG.cpp:2106-2132 prints a side whenever its partner pointer is non-NULL -- self included
                if (c->plus[u] != 0)
                    o << ++nb << "\t" << (u + 1) << "\t+\t" << c->plus[u] << "\t" << ((f & F_STRICT_PLUS) ? "1" : "0")
                      << "\t" << ((f & F_COMPLEX_PLUS) ? "1" : "0") << "\n";
                if (c->minus[u] != 0)
                    o << ++nb << "\t" << (u + 1) << "\t-\t" << c->minus[u] << "\t" << ((f & F_STRICT_MINUS) ? "1" : "0")
                      << "\t" << ((f & F_COMPLEX_MINUS) ? "1" : "0") << "\n";
                continue;
            }
            if (f & F_PLUS_OPEN) {
                o << ++nb << "\t" << (u + 1) << "\t+\t" << c->plus[u] << "\t" << ((f & F_STRICT_PLUS) ? "1" : "0")
                  << "\t" << ((f & F_COMPLEX_PLUS) ? "1" : "0") << "\n";
            }
            if (f & F_MINUS_OPEN) {
                o << ++nb << "\t" << (u + 1) << "\t-\t" << c->minus[u] << "\t" << ((f & F_STRICT_MINUS) ? "1" : "0")
                  << "\t" << ((f & F_COMPLEX_MINUS) ? "1" : "0") << "\n";
            }
        }
    } else {
        for (uint32_t u = 0; u < N; ++u) {
            if (c->colored) { if (c->flags[u] & 3) nb += (c->plus[u] != 0) + (c->minus[u] != 0); }
            else nb += ((c->flags[u] & 1) != 0) + ((c->flags[u] & 2) != 0);
        }
    }
    if (n_bubbles) *n_bubbles = nb;
    return 0;
}

void pfo_state(const pfo_ctx *c, uint8_t *flags, uint32_t *plus, uint32_t *minus) {
    const uint32_t N = c->g.n();
    if (flags) memcpy(flags, c->flags.data(), N);
    if (plus) memcpy(plus, c->plus.data(), (size_t)N * 4);
    if (minus) memcpy(minus, c->minus.data(), (size_t)N * 4);
}

// ploidyEstimation_ptr, CDBG.cpp:1101-1705
int pfo_ploidy_estimation(pfo_ctx *c, const char *outdir, const char *prefix, int lower, int upper, double M,
                          double D, double G, uint64_t allele_out[4], uint64_t *core_cov, uint64_t *core_num) {
    const Graph &g = c->g;
    const int k = g.k;
    const uint32_t N = g.n();
    Scoring sc{M, D, G};
    if (!ensure_dir(outdir)) { g_err = "cannot create output directory"; return 1; }
    const std::string base = std::string(outdir) + "/" + prefix;
    std::ofstream allfre(base + "_allele_frequency.txt", std::ios::trunc), bifre(base + "_bifre.txt", std::ios::trunc),
        trifre(base + "_trifre.txt", std::ios::trunc), tetrafre(base + "_tetrafre.txt", std::ios::trunc),
        pentafre(base + "_pentafre.txt", std::ios::trunc), pentacov(base + "_pentacov.txt", std::ios::trunc),
        bicov(base + "_bicov.txt", std::ios::trunc), tricov(base + "_tricov.txt", std::ios::trunc),
        tetracov(base + "_tetracov.txt", std::ios::trunc), s_var(base + "_alignseq.txt", std::ios::trunc);
    if (!allfre || !bifre || !trifre || !tetrafre || !pentafre || !pentacov || !bicov || !tricov || !tetracov || !s_var) {
        g_err = "cannot open output files";
        return 1;
    }
    std::ofstream *fre_by[4] = {&bifre, &trifre, &tetrafre, &pentafre};
    std::ofstream *cov_by[4] = {&bicov, &tricov, &tetracov, &pentacov};
    uint64_t allele[4] = {0, 0, 0, 0};
    size_t coreNum = 0, coreCov = 0, var_count = 0;
    UbLog ub_log;  // cells whose value is undefined in the reference (indel_len_at), for the tests' masks
    uint64_t cov_lines[4] = {0, 0, 0, 0};
    auto emit = [&](unsigned short maxnum, const std::string &fre, const std::string &cov, bool ub) {
        allfre << fre;
        if (maxnum >= 2 && maxnum <= 5) {
            ++allele[maxnum - 2];
            *fre_by[maxnum - 2] << fre;
            *cov_by[maxnum - 2] << cov;
            ++cov_lines[maxnum - 2];
            if (ub) ub_log.cell(maxnum - 2, cov_lines[maxnum - 2]);
        }
    };
    auto var_dist = [&](const std::vector<uint32_t> &vs, uint32_t i, size_t usize, size_t esize) -> uint32_t {
        uint32_t d;
        if (i == 0) {
            if (i != vs.size() - 1) d = (uint32_t)std::min((size_t)(uint32_t)(vs[i + 1] - vs[i] - 1), usize);
            else d = (uint32_t)std::min(usize, esize);
        } else if (i == vs.size() - 1) {
            d = (uint32_t)std::min((size_t)(uint32_t)(vs[i] - vs[i - 1] - 1), esize);
        } else {
            d = std::min((uint32_t)(vs[i] - vs[i - 1] - 1), (uint32_t)(vs[i + 1] - vs[i] - 1));
        }
        return d;
    };
    for (uint32_t u = 0; u < N; ++u) {
        uint8_t &f = c->flags[u];
        while ((f & 3) != 0) {
            bool st;
            if (f & F_PLUS_OPEN) {
                st = true;
                if (f & F_COMPLEX_PLUS) { f &= 0xFE; continue; }
            } else {  // minus open
                st = false;
                if (f & F_COMPLEX_MINUS) { f &= 0xFD; break; }
            }
            const uint32_t uo = 2 * u + (st ? 0 : 1);
            const bool strict = (f & (st ? F_STRICT_PLUS : F_STRICT_MINUS)) != 0;
            uint64_t core_sum;
            uint32_t core_min;
            if (unitig_cov(*c, uo, core_sum, core_min)) { g_err = "k-mer of the graph missing from the database"; return 2; }
            const double core_mean = (double)core_sum / g.len_km(u);
            uint32_t exit_ov;
            if (strict) {
                exit_ov = g.first_succ(g.first_succ(uo));
            } else {
                const uint32_t want = st ? c->plus[u] : c->minus[u];
                exit_ov = g.first_succ(uo);
                while (exit_ov != NONE && c->id(exit_ov) != want) exit_ov = g.first_succ(exit_ov);
                if (exit_ov == NONE) { g_err = "exit not reachable along first successors"; return 3; }
            }
            if (g.seq[u].compare(g.seq[exit_ov >> 1]) < 0) {
                f &= st ? 0xFE : 0xFD;
                continue;
            }
            const size_t usize = g.size_bp(u), esize = g.size_bp(exit_ov >> 1);
            if (strict) {
                double sum = 0;
                std::vector<uint32_t> uv;
                std::vector<double> cov;
                bool pass = true;
                for (int b = 0; b < 4; ++b) {
                    uint32_t w = g.succ_row(uo)[b];
                    if (w == NONE) continue;
                    uv.push_back(w);
                    uint64_t s;
                    uint32_t mn;
                    if (unitig_cov(*c, w, s, mn)) { g_err = "k-mer of the graph missing from the database"; return 2; }
                    if (mn > (uint32_t)lower && mn < (uint32_t)upper) {
                        double mean = (double)s / g.len_km(w >> 1);
                        cov.push_back(mean);
                        sum += mean;
                    } else {
                        pass = false;
                        break;
                    }
                }
                if (pass) {
                    // predecessor coverages are computed and dropped (CDBG.cpp:1224-1239); the
                    // only observable effect is the exit on a missing k-mer
                    for (int b = 0; b < 4; ++b) {
                        uint32_t w = g.pred_row(uo)[b];
                        if (w == NONE) continue;
                        uint64_t s;
                        uint32_t mn;
                        if (unitig_cov(*c, w, s, mn)) { g_err = "k-mer of the graph missing from the database"; return 2; }
                    }
                    sort_simple(g, cov, uv, 0, (int)cov.size() - 1);
                    std::vector<std::string> strs;
                    for (uint32_t w : uv) strs.push_back(g.mapped(w));
                    AlignResult ar = align_paths(sc, strs);
                    if (!ar.rows.empty()) {
                        ++var_count;
                        for (auto &s : ar.rows)
                            s_var << var_count << "\t" << 1 << "\t" << (u + 1) << "\t" << c->id(exit_ov) << "\t" << s << "\n";
                        coreCov += (size_t)core_mean;
                        coreNum++;
                        std::vector<uint32_t> var_site;
                        for (uint32_t i = 0; i < ar.partition.size(); ++i)
                            if (ar.partition[i].back() > 0) var_site.push_back(i);
                        uint32_t indel = 0;
                        for (uint32_t i = 0; i < var_site.size(); ++i) {
                            std::stringstream cov_info, fre_info;
                            const std::vector<uint16_t> &part = ar.partition[var_site[i]];
                            unsigned short maxnum = *std::max_element(part.begin(), part.end());
                            std::vector<double> tc(maxnum, 0);
                            uint32_t vd = var_dist(var_site, i, usize, esize);
                            for (size_t j = 0; j < part.size(); ++j) tc[part[j] - 1] += cov[j];
                            for (double x : tc) {
                                cov_info << x << "\t";
                                fre_info << (double)x / sum << "\n";
                            }
                            bool ub = false;
                            if (std::find(ar.indel_pos.begin(), ar.indel_pos.end(), var_site[i]) != ar.indel_pos.end()) {
                                ++indel;
                                cov_info << 1 << "\t" << indel_len_at(ar, indel - 1, &ub) << "\t" << var_count << "\t" << var_site.size()
                                         << "\t" << vd << "\t" << "\n";
                            } else {
                                cov_info << 1 << "\t" << "0\t" << var_count << "\t" << var_site.size() << "\t" << vd << "\t" << "\n";
                            }
                            emit(maxnum, fre_info.str(), cov_info.str(), ub);
                        }
                    }
                }
            } else {
                // all s->t paths (CDBG.cpp:1364-1412)
                std::vector<std::string> strs;
                std::vector<uint32_t> major, minor;
                std::string bubble;
                const uint32_t ulen = g.len_km(u);
                minor.push_back(uo);
                while (!minor.empty()) {
                    uint32_t w = minor.back();
                    minor.pop_back();
                    major.push_back(w);
                    std::string str = g.mapped(w);
                    const uint32_t wlen = g.len_km(w >> 1);
                    bubble += str.substr(0, wlen);
                    if ((w >> 1) == (exit_ov >> 1)) {
                        bubble += str.substr(wlen);
                        strs.push_back(bubble.substr(ulen - 1, bubble.length() - ulen + 1 - wlen + 1));
                        bubble = bubble.substr(0, bubble.length() - str.length());
                        major.pop_back();
                        while (!major.empty() && !minor.empty()) {
                            bool linked = false;
                            for (int b = 0; b < 4; ++b)
                                if (g.succ_row(major.back())[b] == minor.back()) { linked = true; break; }
                            if (linked) break;
                            bubble = bubble.substr(0, bubble.length() - g.len_km(major.back() >> 1));
                            major.pop_back();
                        }
                    } else {
                        for (int b = 0; b < 4; ++b) {
                            uint32_t x = g.succ_row(w)[b];
                            if (x != NONE) minor.push_back(x);
                        }
                    }
                }
                sort_branching(strs, 0, (int)strs.size() - 1);
                AlignResult ar = align_paths(sc, strs);
                if (!ar.rows.empty()) {
                    const std::vector<std::string> &rows = ar.rows;
                    ++var_count;
                    for (auto &s : rows)
                        s_var << var_count << "\t" << 0 << "\t" << (u + 1) << "\t" << c->id(exit_ov) << "\t" << s << "\n";
                    coreCov += (size_t)core_mean;
                    coreNum++;
                    std::vector<uint32_t> var_site;
                    for (uint32_t i = 0; i < ar.partition.size(); ++i)
                        if (ar.partition[i].back() > 0) var_site.push_back(i);
                    uint32_t indel = 0;
                    for (uint32_t i = 0; i < var_site.size(); ++i) {
                        std::stringstream cov_info, fre_info;
                        const uint32_t site = var_site[i];
                        const std::vector<uint16_t> &part = ar.partition[site];
                        unsigned short maxnum = *std::max_element(part.begin(), part.end());
                        std::vector<std::string> kstr(rows.size());
                        std::vector<std::set<std::string>> groups(maxnum);
                        uint32_t vd = var_dist(var_site, i, usize, esize);
                        const bool is_indel =
                            std::find(ar.indel_pos.begin(), ar.indel_pos.end(), site) != ar.indel_pos.end();
                        if (is_indel) {
                            std::vector<int> at(rows.size(), (int)site);
                            while (true) {
                                std::set<char> cs;
                                for (size_t p = 0; p < rows.size(); ++p) {
                                    // std::string::substr(pos, 1): one character, the EMPTY string at pos == size() (a row
                                    // that ends in gaps: nothing is appended and c[0] == '\0' joins the set), out_of_range
                                    // beyond -- the reference then terminates (CDBG.cpp:1478-1488, CCDBG.cpp:3188-3198)
                                    std::string ch = rows[p].substr(at[p], 1);
                                    while (ch.compare("-") == 0) {
                                        at[p] += 1;
                                        ch = rows[p].substr(at[p], 1);
                                    }
                                    at[p] += 1;
                                    kstr[p] += ch;
                                    cs.insert(ch[0]);
                                }
                                if (cs.size() > 1) break;
                            }
                            if (indel == 0) {
                                for (size_t p = 0; p < rows.size(); ++p) {
                                    int n = (int)kstr[p].length();
                                    kstr[p] = rows[p].substr(site - k + n, k - n) + kstr[p];
                                }
                            } else {
                                for (size_t p = 0; p < rows.size(); ++p) {
                                    int n = (int)kstr[p].length();
                                    std::string tmp = strip_gaps(rows[p].substr(0, site));
                                    if (tmp.length() < (size_t)(k - n)) {
                                        kstr[p] = tmp + kstr[p];
                                        for (int x = at[p]; kstr[p].length() < (size_t)k; ++x) {
                                            std::string ch = rows[p].substr(x, 1);
                                            if (ch.compare("-") != 0) kstr[p] += ch;
                                        }
                                    } else {
                                        kstr[p] = tmp.substr(tmp.length() - k + n, k - n) + kstr[p];
                                    }
                                }
                            }
                            ++indel;
                        } else if (indel > 0) {
                            for (size_t p = 0; p < rows.size(); ++p) {
                                std::string tmp = strip_gaps(rows[p].substr(0, site + 1));
                                if (tmp.length() < (size_t)k) {
                                    kstr[p] = tmp;
                                    for (int x = site + 1; kstr[p].length() < (size_t)k; ++x) {
                                        std::string ch = rows[p].substr(x, 1);
                                        if (ch.compare("-") != 0) kstr[p] += ch;
                                    }
                                } else {
                                    kstr[p] = tmp.substr(tmp.length() - k, k);
                                }
                            }
                        } else {
                            for (size_t p = 0; p < rows.size(); ++p) kstr[p] = rows[p].substr(site - k + 1, k);
                        }
                        for (size_t p = 0; p < part.size(); ++p) groups[part[p] - 1].insert(kstr[p]);
                        std::vector<double> tc(maxnum, 0);
                        double sum = 0;
                        bool site_ok = true;
                        for (size_t gi = 0; gi < groups.size() && site_ok; ++gi) {
                            for (const auto &s : groups[gi]) {
                                uint64_t ssum;
                                bool ok;
                                uint64_t lost = 0;
                                if (string_cov(*c, s, (uint32_t)lower, (uint32_t)upper, ssum, ok, &lost)) {
                                    // the reference's own line (src/CDBG.cpp:54): the object as it was looked up
                                    std::string km((size_t)k, 'A');
                                    for (int x = 0; x < k; ++x) km[x] = "ACGT"[(lost >> (2 * (k - 1 - x))) & 3];
                                    g_err = "CDBG::readCov():" + km + " kmer can not found .";
                                    return 2;
                                }
                                if (!ok) { site_ok = false; break; }
                                tc[gi] += (double)ssum / (s.length() - k + 1);
                            }
                            sum += tc[gi];
                        }
                        if (!site_ok) continue;
                        for (double x : tc) {
                            cov_info << x << "\t";
                            fre_info << x / sum << "\n";
                        }
                        bool ub = false;
                        if (is_indel)
                            cov_info << 0 << "\t" << indel_len_at(ar, indel - 1, &ub) << "\t" << var_count << "\t" << var_site.size()
                                     << "\t" << vd << "\t" << "\n";
                        else
                            cov_info << 0 << "\t" << "0\t" << var_count << "\t" << var_site.size() << "\t" << vd << "\t" << "\n";
                        emit(maxnum, fre_info.str(), cov_info.str(), ub);
                    }
                }
            }
            // CDBG.cpp:1656-1679
            f &= st ? 0xFE : 0xFD;
            uint8_t &fe = c->flags[exit_ov >> 1];
            if (c->strand(exit_ov)) fe &= 0xFD; else fe &= 0xFE;
        }
    }
    for (int i = 0; i < 4; ++i) allele_out[i] = allele[i];
    *core_cov = coreCov;
    *core_num = coreNum;
    return 0;
}

}  // extern "C"
