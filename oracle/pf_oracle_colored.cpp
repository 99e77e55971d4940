// ORACLE (test infrastructure only -- see pf_oracle.h).  CPU restatement of the colored (multi-sample) twin of
// the hot path, reference src/CCDBG.cpp: the state commits that differ from CDBG's (:2349-2401, :2402-2661),
// readCov/readCovUni (:89-156), computeCramerVCoefficient (:330-366), the colored sortSeq_simple (:368-480) and
// ploidyEstimation_ptr (:2759-3531).  Colour sets are not re-parsed from .bfg_colors here: they are read from
// the dump the real Bifrost produced for the fixture (oracle/ref_colors_dump.cpp), which is what pins the
// colour semantics (UnitigColors::contains / ::size, bifrost/src/ColorSet.cpp:776-823, 898-927).
#include <algorithm>
#include <cmath>
#include <cstring>
#include <fstream>
#include <map>
#include <set>
#include <sstream>

#include "pf_oracle_align.hpp"
#include "pf_oracle_ctx.hpp"

using namespace pfo;
using namespace pfo_int;

namespace pfo_int {

bool ColorData::load_dump(const std::string &path, const Graph &g) {
    std::ifstream in(path);
    if (!in) { err = "cannot open colour dump " + path; return false; }
    std::string line;
    const uint32_t N = g.n();
    bits.assign(N, {});
    size_total.assign(N, 0);
    n_full_enc.assign(N, 0);
    uint32_t seen = 0;
    while (std::getline(in, line)) {
        if (line.empty()) continue;
        std::vector<std::string> f;
        size_t a = 0;
        while (true) {
            size_t b = line.find('\t', a);
            f.push_back(line.substr(a, b == std::string::npos ? b : b - a));
            if (b == std::string::npos) break;
            a = b + 1;
        }
        if (f[0] == "#colors") {
            n_colors = (uint32_t)std::stoul(f[1]);
            if (std::stoi(f[3]) != g.k || std::stoul(f[5]) != N) { err = "colour dump does not belong to this graph"; return false; }
            continue;
        }
        if (f[0] == "#name") { names.push_back(f[1]); continue; }
        if (f.size() < 6) { err = "malformed colour dump line"; return false; }
        const uint32_t u = (uint32_t)std::stoul(f[0]) - 1;
        if (u >= N || g.seq[u].compare(0, g.k, f[1]) != 0 || std::stoul(f[2]) != g.len_km(u)) {
            err = "colour dump and graph disagree on unitig " + f[0];
            return false;
        }
        size_total[u] = std::stoull(f[3]);
        n_full_enc[u] = (uint32_t)std::stoul(f[4]);
        const uint32_t km = g.len_km(u);
        size_t p = 0;
        for (uint32_t c = 0; c < n_colors; ++c) {
            size_t q = f[5].find(',', p);
            std::string tok = f[5].substr(p, q == std::string::npos ? q : q - p);
            p = q == std::string::npos ? f[5].size() : q + 1;
            if (tok == "F") tok.assign(km, '1');
            else if (tok == "E") tok.assign(km, '0');
            if (tok.size() != km) { err = "malformed colour row for unitig " + f[0]; return false; }
            bits[u].push_back(tok);
        }
        ++seen;
    }
    if (seen != N || n_colors == 0) { err = "colour dump incomplete"; return false; }
    return true;
}

void KmerIndex::build(const Graph &g) {
    where.clear();
    where.reserve(g.n_kmers * 2);
    for (uint32_t u = 0; u < g.n(); ++u) {
        const std::string &s = g.seq[u];
        for (uint32_t i = 0; i + g.k <= s.size(); ++i) {
            uint64_t x = pack_kmer(s.data() + i, g.k), r = rc_kmer(x, g.k);
            where[x < r ? x : r] = ((uint64_t)u << 32) | i;
        }
    }
}

// CompactedDBG::findUnitig(const char*, pos=0, len) (CompactedDBG.tcc:3815-3837) with CompressedSequence::jump
// (CompressedSequence.cpp:497-520): the match is extended along the unitig while the characters agree
bool KmerIndex::find_unitig(const Graph &g, const std::string &s, uint32_t &u, uint32_t &dist, uint32_t &len) const {
    const int k = g.k;
    if ((int)s.size() < k) return false;
    for (int i = 0; i < k; ++i)
        if (base_code(s[i]) < 0) return false;
    uint64_t x = pack_kmer(s.data(), k), r = rc_kmer(x, k);
    auto it = where.find(x < r ? x : r);
    if (it == where.end()) return false;
    u = (uint32_t)(it->second >> 32);
    const uint32_t p = (uint32_t)it->second;
    const std::string &seq = g.seq[u];
    const bool fwd = pack_kmer(seq.data() + p, k) == x;
    size_t j = 0;
    if (fwd) {
        while (j < s.size() && p + j < seq.size() && s[j] == seq[p + j]) ++j;
        len = (uint32_t)(j - k + 1);
        dist = p;
    } else {
        int pos = (int)p + k - 1;
        while (j < s.size() && pos >= 0 && s[j] == comp(seq[pos])) { ++j; --pos; }
        len = (uint32_t)(j - k + 1);
        dist = p - (len - 1);
    }
    return true;
}

// setNoBubble_ptr_cycle, CCDBG.cpp:2349-2401
void commit_cycle_exit_colored(pfo_ctx &c, const Traversal &t, uint32_t s) {
    for (uint32_t w : t.seen) c.poison_linked_only(w >> 1);
    c.set_side_self(s >> 1, c.strand(s));
    c.set_side_self(t.exit_ov >> 1, !c.strand(t.exit_ov));
}

// setNoBubble_ptr(p, vec), CCDBG.cpp:2402-2661
void commit_accept_colored(pfo_ctx &c, const Traversal &t, uint32_t s) {
    const Graph &g = c.g;
    const ColorData &col = c.col;
    const uint32_t C = col.n_colors;
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
    auto both_self = [&] {
        c.set_side_self(ds, c.strand(s));
        c.set_side_self(de, !c.strand(e));
    };
    // colour completeness of both endpoints (:2530-2572); the exit's set is sized with the entrance's mapping
    bool f = true;
    const uint64_t km_s = g.len_km(ds), km_e = g.len_km(de);
    if (col.size_with(ds, km_s, km_s) != km_s * C) {
        f = false;
        c.flags[ds] |= F_NON_SUPER;
        both_self();
    }
    if (col.size_with(de, km_e, km_s) != km_e * C) {
        f = false;
        c.flags[de] |= F_NON_SUPER;
        both_self();
    }
    if (!f) return;
    // colour flow (:2573-2621): every colour a vertex carries in full must continue, in full, on some successor
    std::map<uint32_t, std::vector<int>> cmap;  // keyed by unitig id
    cmap[c.id(s)].resize(C);
    cmap[c.id(e)].resize(C);
    for (uint32_t i = 0; i < C; ++i) { cmap[c.id(s)][i] = (int)i; cmap[c.id(e)][i] = (int)i; }
    for (uint32_t w : t.seen) {
        if (w == e) continue;
        if (cmap.find(c.id(w)) == cmap.end()) {
            std::vector<int> colour;
            for (uint32_t i = 0; i < C; ++i)
                if (col.full(w >> 1, i)) colour.push_back((int)i);
            cmap[c.id(w)] = colour;
        }
        const std::vector<int> &mine = cmap[c.id(w)];
        std::set<int> suc_colour;
        for (int b = 0; b < 4; ++b) {
            uint32_t x = g.succ_row(w)[b];
            if (x == NONE) continue;
            for (int ci : mine)
                if (col.full(x >> 1, (uint32_t)ci)) suc_colour.insert(ci);
        }
        if (suc_colour.size() != mine.size()) { f = false; break; }
    }
    if (f) {
        if (c.strand(s)) { c.plus[ds] = de + 1; c.flags[ds] |= F_PLUS_OPEN; }
        else { c.minus[ds] = de + 1; c.flags[ds] |= F_MINUS_OPEN; }
        if (c.strand(e)) { c.minus[de] = ds + 1; c.flags[de] |= F_MINUS_OPEN; }
        else { c.plus[de] = ds + 1; c.flags[de] |= F_PLUS_OPEN; }
    } else {
        both_self();
    }
}

}  // namespace pfo_int

namespace {

struct Cov { double mean; bool ok; };

// readCovUni, CCDBG.cpp:123-156 (a missing k-mer is (0,false) here, not an exit)
Cov cov_uni(const pfo_ctx &c, uint32_t colour, uint32_t u, uint32_t low, uint32_t up) {
    const KmcDb &db = c.dbs[colour];
    const Graph &g = c.g;
    double sum = 0;
    const uint32_t L = g.len_km(u);
    if (db.both_strands) {
        const std::string &s = g.seq[u];
        for (uint32_t i = 0; i < L; ++i) {
            uint32_t cnt = 0;
            if (!db.canonical_count(pack_kmer(s.data() + i, g.k), cnt)) return {0, false};
            if (cnt > low && cnt < up) sum += cnt;
            else return {0, false};
        }
    }
    return {sum / L, true};
}

// readCov(string, low, up, colour), CCDBG.cpp:89-122
Cov cov_str(const pfo_ctx &c, uint32_t colour, const std::string &s, uint32_t low, uint32_t up) {
    const KmcDb &db = c.dbs[colour];
    const int k = c.g.k;
    double sum = 0;
    if (db.both_strands) {
        StringProbe probe(db, k);
        for (size_t i = 0; i + k <= s.size(); ++i) {
            uint32_t cnt = 0;
            if (!probe.count(s.data() + i, cnt)) return {0, false};
            if (cnt > low && cnt < up) sum += cnt;
            else return {0, false};
        }
    }
    return {sum / (double)(s.length() - k + 1), true};
}

// computeCramerVCoefficient, CCDBG.cpp:330-366
double cramer_v(const std::vector<double> &A, const std::vector<double> &B) {
    double n = 0, nA = 0, nB = 0;
    uint8_t count = 0;
    std::vector<double> p(A.size(), 0);
    double chi = 0;
    for (size_t i = 0; i < A.size(); ++i) {
        nA += A[i];
        nB += B[i];
        p[i] = A[i] + B[i];
        n = n + p[i];
        if (p[i] != 0) ++count;
    }
    if (count < 2) return 0;
    for (size_t i = 0; i < A.size(); ++i) {
        if (p[i] == 0) continue;
        double exA = nA * p[i] / n;
        double exB = nB * p[i] / n;
        chi += pow(A[i] - exA, 2) / exA;
        chi += pow(B[i] - exB, 2) / exB;
    }
    return sqrt(chi / n);
}

double max_cramer_v(const std::vector<std::vector<double>> &cov) {
    double coefficient = 0;
    for (size_t ci = 0; ci + 1 < cov.size(); ++ci)
        for (size_t cj = ci + 1; cj < cov.size(); ++cj) coefficient = std::max(coefficient, cramer_v(cov[ci], cov[cj]));
    return coefficient;
}

// sortSeq_simple, CCDBG.cpp:368-480: descending #colours, then descending length, then descending strcmp
void sort_simple_colored(const Graph &g, std::vector<size_t> &pc, std::vector<uint32_t> &uv,
                         std::vector<std::vector<double>> &cov, int low, int high) {
    if (high <= low) return;
    int i = low, j = high;
    auto ref = [&](int x) -> const std::string & { return g.seq[uv[x] >> 1]; };
    auto swap_at = [&](int a, int b) {
        std::swap(pc[a], pc[b]);
        std::swap(uv[a], uv[b]);
        for (auto &row : cov) std::swap(row[a], row[b]);
    };
    while (true) {
        while (pc[i] >= pc[low]) {
            if (pc[i] > pc[low]) i++;
            else if (ref(i).length() > ref(low).length()) i++;
            else if (ref(i).length() == ref(low).length()) {
                if (strcmp(ref(i).c_str(), ref(low).c_str()) > 0) i++;
                else break;
            } else break;
            if (i == high) break;
        }
        while (pc[j] <= pc[low]) {
            if (pc[j] < pc[low]) j--;
            else if (ref(j).length() < ref(low).length()) j--;
            else if (ref(j).length() == ref(low).length()) {
                if (strcmp(ref(j).c_str(), ref(low).c_str()) < 0) j--;
                else break;
            } else break;
            if (j == low) break;
        }
        if (i >= j) break;
        swap_at(i, j);
    }
    swap_at(low, j);
    sort_simple_colored(g, pc, uv, cov, low, j - 1);
    sort_simple_colored(g, pc, uv, cov, j + 1, high);
}

}  // namespace

extern "C" {

pfo_ctx *pfo_open_colored(const char *gfa_path, const char *colors_dump, const char *db_list_file) {
    pfo_ctx *c = new pfo_ctx();
    auto fail = [&](const std::string &m) -> pfo_ctx * { g_err = m; delete c; return nullptr; };
    if (!c->g.load_gfa(gfa_path)) return fail(c->g.err);
    c->g.build_adjacency();
    if (!c->col.load_dump(colors_dump, c->g)) return fail(c->col.err);
    c->colored = true;
    if (db_list_file && *db_list_file) {
        // CCDBG::CCDBG, CCDBG.cpp:13-43: one database name per line, one per colour
        std::ifstream in(db_list_file);
        if (!in) return fail("CCDBG::CCDBG():Error: Open kmc database name file error");
        c->dbs.resize(c->col.n_colors);
        for (uint32_t i = 0; i < c->col.n_colors; ++i) {
            std::string name;
            std::getline(in, name, '\n');
            if (!c->dbs[i].load(name)) return fail("CCDBG::CCDBG():Error: Open kmc database " + name + " error");
            if ((int)c->dbs[i].k != c->g.k) return fail("k of a KMC database differs from the graph's");
        }
    }
    c->kidx.build(c->g);
    const uint32_t N = c->g.n();
    c->flags.assign(N, 0);
    c->plus.assign(N, 0);
    c->minus.assign(N, 0);
    return c;
}

uint32_t pfo_num_colors(const pfo_ctx *c) { return c->colored ? c->col.n_colors : 0; }

// colour presence of unitig u: out[colour * n_kmers + i] = 0/1; returns UnitigColors::size(um); *n_full_enc as dumped
uint64_t pfo_unitig_colors(const pfo_ctx *c, uint32_t u, uint8_t *out, uint32_t *n_full_enc) {
    const uint32_t km = c->g.len_km(u);
    if (out)
        for (uint32_t ci = 0; ci < c->col.n_colors; ++ci)
            for (uint32_t i = 0; i < km; ++i) out[(size_t)ci * km + i] = c->col.bits[u][ci][i] == '1';
    if (n_full_enc) *n_full_enc = c->col.n_full_enc[u];
    return c->col.size_total[u];
}

// readCovUni for one colour: *ok = 0 when a k-mer is missing or a count is outside (low, up)
void pfo_unitig_cov_color(const pfo_ctx *c, uint32_t colour, uint32_t u, uint32_t low, uint32_t up, double *mean, int *ok) {
    Cov r = cov_uni(*c, colour, u, low, up);
    *mean = r.mean;
    *ok = r.ok;
}

void pfo_string_cov_color(const pfo_ctx *c, uint32_t colour, const char *s, uint32_t len, uint32_t low, uint32_t up, double *mean,
                          int *ok) {
    Cov r = cov_str(*c, colour, std::string(s, len), low, up);
    *mean = r.mean;
    *ok = r.ok;
}

int pfo_find_unitig(const pfo_ctx *c, const char *s, uint32_t len, uint32_t *u, uint32_t *dist, uint32_t *n) {
    return c->kidx.find_unitig(c->g, std::string(s, len), *u, *dist, *n) ? 1 : 0;
}

// ploidyEstimation_ptr, CCDBG.cpp:2759-3531
int pfo_ploidy_estimation_colored(pfo_ctx *c, const char *outdir, const char *prefix, const int *lower, const int *upper,
                                  double M, double D, double G, uint64_t allele_out[4], uint64_t *core_cov, uint64_t *core_num) {
    const Graph &g = c->g;
    const ColorData &col = c->col;
    const uint32_t C = col.n_colors;
    const int k = g.k;
    const uint32_t N = g.n();
    Scoring sc{M, D, G};
    if (!c->colored || c->dbs.size() != C) { g_err = "colored context with one database per colour required"; return 1; }
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
    auto emit = [&](size_t arity, const std::string &fre, const std::string &cov, bool ub) {
        allfre << fre;
        if (arity >= 2 && arity <= 5) {
            ++allele[arity - 2];
            *fre_by[arity - 2] << fre;
            *cov_by[arity - 2] << cov;
            ++cov_lines[arity - 2];
            if (ub) ub_log.cell((int)arity - 2, cov_lines[arity - 2]);
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
    // one row per colour with at least two non-zero allele groups (:3006-3058, :3292-3339, :3427-3475)
    auto emit_rows = [&](const std::vector<std::vector<double>> &group_cov, const std::string &tail, bool ub) {
        for (uint32_t ci = 0; ci < C; ++ci) {
            std::vector<double> res;
            double sum = 0;
            for (double x : group_cov[ci])
                if (x > 0.0) { res.push_back(x); sum += x; }
            if (res.size() < 2) continue;
            std::stringstream cov_info, fre_info;
            for (double x : res) {
                cov_info << x << "\t";
                fre_info << (double)x / sum << "\n";
            }
            cov_info << ci << "\t" << tail;
            emit(res.size(), fre_info.str(), cov_info.str(), ub);
        }
    };
    for (uint32_t u = 0; u < N; ++u) {
        uint8_t &f = c->flags[u];
        while ((f & 3) != 0) {
            bool st;
            if (f & F_PLUS_OPEN) {
                st = true;
                if (f & F_COMPLEX_PLUS) { f &= 0xFE; continue; }
            } else {
                st = false;
                if (f & F_COMPLEX_MINUS) { f &= 0xFD; break; }
            }
            const uint32_t uo = 2 * u + (st ? 0 : 1);
            const bool strict = (f & (st ? F_STRICT_PLUS : F_STRICT_MINUS)) != 0;
            // core (:2838-2853): the "flag == false;" there is a no-op, so a failing colour only stops the sum
            double core_first = 0;
            for (uint32_t i = 0; i < C; ++i) {
                Cov t = cov_uni(*c, i, u, (uint32_t)lower[i], (uint32_t)upper[i]);
                if (t.ok) core_first += t.mean;
                else break;
            }
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
                const int nsucc = g.out_degree(uo);
                std::vector<std::vector<double>> cov(C, std::vector<double>(nsucc, 0));
                std::vector<size_t> path_color;
                std::vector<uint32_t> uv;
                bool flag = true;
                uint8_t path = 0;
                for (int b = 0; b < 4 && flag; ++b) {
                    uint32_t w = g.succ_row(uo)[b];
                    if (w == NONE) continue;
                    uv.push_back(w);
                    size_t j = 0;
                    for (uint32_t i = 0; i < C; ++i) {
                        if (col.full(w >> 1, i)) {
                            j++;
                            Cov inside = cov_uni(*c, i, w >> 1, (uint32_t)lower[i], (uint32_t)upper[i]);
                            if (inside.ok) cov[i][path] = inside.mean;
                            else { flag = false; break; }
                        }
                    }
                    if (!flag) break;
                    if (col.size_total[w >> 1] != j * g.len_km(w >> 1)) { flag = false; break; }
                    ++path;
                    path_color.push_back(j);
                }
                if (flag) {
                    flag = false;
                    for (const auto &row : cov) {
                        int nz = 0;
                        for (double d : row)
                            if (d != 0.0) ++nz;
                        if (nz > 1) { flag = true; break; }
                    }
                }
                if (flag) {
                    sort_simple_colored(g, path_color, uv, cov, 0, (int)path_color.size() - 1);
                    std::vector<std::string> strs;
                    for (uint32_t w : uv) strs.push_back(g.mapped(w));
                    AlignResult ar = align_paths(sc, strs);
                    if (!ar.rows.empty()) {
                        ++var_count;
                        for (auto &s : ar.rows)
                            s_var << var_count << "\t" << 1 << "\t" << (u + 1) << "\t" << c->id(exit_ov) << "\t" << s << "\n";
                        coreCov += (size_t)core_first;
                        coreNum++;
                        std::vector<uint32_t> var_site;
                        for (uint32_t i = 0; i < ar.partition.size(); ++i)
                            if (ar.partition[i].back() > 0) var_site.push_back(i);
                        uint32_t indel = 0;
                        const double coefficient = max_cramer_v(cov);
                        for (uint32_t i = 0; i < var_site.size(); ++i) {
                            std::stringstream tail;
                            const std::vector<uint16_t> &part = ar.partition[var_site[i]];
                            unsigned short maxnum = *std::max_element(part.begin(), part.end());
                            uint32_t vd = var_dist(var_site, i, usize, esize);
                            bool ub = false;
                            if (std::find(ar.indel_pos.begin(), ar.indel_pos.end(), var_site[i]) != ar.indel_pos.end()) {
                                ++indel;
                                tail << 1 << "\t" << indel_len_at(ar, indel - 1, &ub) << "\t" << var_count << "\t" << var_site.size() << "\t"
                                     << coefficient << "\t" << vd << "\t" << "\n";
                            } else {
                                tail << 1 << "\t" << "0\t" << var_count << "\t" << var_site.size() << "\t" << coefficient << "\t" << vd
                                     << "\t" << "\n";
                            }
                            std::vector<std::vector<double>> group_cov(C, std::vector<double>(maxnum, 0.0));
                            for (uint32_t ci = 0; ci < C; ++ci)
                                for (size_t j = 0; j < part.size(); ++j) group_cov[ci][part[j] - 1] += cov[ci][j];
                            emit_rows(group_cov, tail.str(), ub);
                        }
                    }
                }
            } else {
                // all s->t paths (:3079-3127, the same walk as CDBG.cpp:1364-1412)
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
                    ++coreNum;
                    coreCov += (size_t)core_first;
                    ++var_count;
                    for (auto &s : rows)
                        s_var << var_count << "\t" << 0 << "\t" << (u + 1) << "\t" << c->id(exit_ov) << "\t" << s << "\n";
                    std::vector<uint32_t> var_site;
                    for (uint32_t i = 0; i < ar.partition.size(); ++i)
                        if (ar.partition[i].back() > 0) var_site.push_back(i);
                    uint32_t indel = 0;
                    for (uint32_t i = 0; i < var_site.size(); ++i) {
                        const uint32_t site = var_site[i];
                        const std::vector<uint16_t> &part = ar.partition[site];
                        unsigned short maxnum = *std::max_element(part.begin(), part.end());
                        std::vector<std::string> kstr(rows.size());
                        std::vector<std::set<std::string>> groups(maxnum);
                        uint32_t vd = var_dist(var_site, i, usize, esize);
                        const bool is_indel = std::find(ar.indel_pos.begin(), ar.indel_pos.end(), site) != ar.indel_pos.end();
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
                        // per-colour group coverage (:3240-3283, :3378-3418)
                        std::vector<std::vector<double>> group_cov(C, std::vector<double>(maxnum, 0.0));
                        std::set<size_t> colour_set;
                        bool site_ok = true;
                        for (size_t gi = 0; gi < groups.size() && site_ok; ++gi) {
                            for (const auto &s : groups[gi]) {
                                uint32_t pu, pdist, plen;
                                if (!c->kidx.find_unitig(g, s, pu, pdist, plen)) { g_err = "site string not found in the graph"; return 3; }
                                for (uint32_t ci = 0; ci < C; ++ci) {
                                    if (col.contains(pu, ci, pdist, plen)) {
                                        colour_set.insert(ci);
                                        Cov r = cov_str(*c, ci, s, (uint32_t)lower[ci], (uint32_t)upper[ci]);
                                        if (!r.ok) { site_ok = false; break; }
                                        group_cov[ci][gi] += r.mean;
                                    }
                                }
                                if (!site_ok) break;
                            }
                        }
                        if (colour_set.size() != C) continue;
                        if (!site_ok) continue;
                        const double coefficient = max_cramer_v(group_cov);
                        std::stringstream tail;
                        bool ub = false;
                        if (is_indel)
                            tail << 0 << "\t" << indel_len_at(ar, indel - 1, &ub) << "\t" << var_count << "\t" << var_site.size() << "\t" << coefficient
                                 << "\t" << vd << "\t" << "\n";
                        else
                            tail << 0 << "\t" << "0\t" << var_count << "\t" << var_site.size() << "\t" << coefficient << "\t" << vd << "\t"
                                 << "\n";
                        emit_rows(group_cov, tail.str(), ub);
                    }
                }
            }
            // :3481-3504
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
