// Row predicates of script/Filter.R and script/Filter-multi.R -- see pf_filter.hpp.  PARITY UNPINNED (no R in the build image).
#include "pf_filter.hpp"

#include <climits>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iostream>
#include <sstream>

namespace pfh {

// ---- R's number rendering ---------------------------------------------------------------------------------------------------------
// write.table sets R_print.digits = DBL_DIG (15) and encodes every cell on its own: formatReal() finds the fewest significant digits
// nsig <= 15 that reproduce the value rounded to 15 digits and its decimal exponent kp, then compares the widths of the two notations
// (R src/main/format.c: scientific(), formatReal(); src/main/printutils.c: EncodeReal0): fixed needs neg + max(kp + 1, 1) digits left of
// the point and rgt = max(0, nsig - kp - 1) right of it; scientific needs neg + (mF > 0) + mF + 4 + e with mF = nsig - 1 mantissa
// decimals and e = 1 exponent digit beyond the two always printed (2 when |kp| >= 100).  Fixed wins ties (scipen = 0).
std::string r_format_double(double x) {
    if (std::isnan(x)) return "NA";
    if (std::isinf(x)) return x > 0 ? "Inf" : "-Inf";
    if (x == 0.0) return "0";
    char buf[64];
    snprintf(buf, sizeof buf, "%.14e", std::fabs(x));   // 15 significant digits, correctly rounded
    // d.dddddddddddddde[+-]XX
    const char *e = strchr(buf, 'e');
    const int kp = atoi(e + 1);
    int nsig = 15;
    for (const char *p = e - 1; p > buf && nsig > 1; --p) {
        if (*p == '.') continue;
        if (*p != '0') break;
        --nsig;
    }
    const int neg = x < 0 ? 1 : 0;
    const int left = kp + 1;
    const int rgt = std::max(0, nsig - kp - 1);
    const int w_fixed = neg + (left <= 0 ? 1 : left) + (rgt ? rgt + 1 : 0);
    const int mF = nsig - 1;
    const int w_sci = neg + (mF > 0 ? 1 : 0) + mF + 4 + ((kp >= 100 || kp <= -100) ? 2 : 1);
    char out[400];
    if (w_fixed <= w_sci) snprintf(out, sizeof out, "%.*f", rgt, x);
    else snprintf(out, sizeof out, "%.*e", mF, x);   // (C prints at least two exponent digits, as R does)
    return out;
}

// round(x, 7): R >= 4.0.0 picks, of the two decimal neighbours with 7 places, the one nearer to x (ties to even); in long double
// the scaled value's nearest integer (ties to even) gives the same neighbour for every x this script can meet (0 <= x <= 1)
double r_round7(double x) {
    const long double s = 1e7L;
    return (double)(nearbyintl((long double)x * s) / s);
}

namespace {

struct Column {
    std::vector<double> v;
    bool is_int = true;   // type.convert: every entry an integer inside int's range -> integer column, else double
};

struct Table {
    size_t ncol = 0, nrow = 0;
    std::vector<Column> col;
    bool loaded = false;
};

bool parse_int32(const std::string &t, double &out) {
    if (t.empty()) return false;
    size_t i = (t[0] == '-' || t[0] == '+') ? 1 : 0;
    if (i == t.size() || t.size() - i > 10) return false;
    for (size_t j = i; j < t.size(); ++j)
        if (t[j] < '0' || t[j] > '9') return false;
    const long long x = atoll(t.c_str());
    if (x > INT_MAX || x < -INT_MAX) return false;   // (R keeps INT_MIN for NA)
    out = (double)x;
    return true;
}

// read.table(file, col.names = <ncol names>): fields separated by white space (a row's trailing tab adds none), blank lines and
// everything after '#' skipped, an empty file = a table of no rows (col.names given: R src/library/utils/R/readtable.R); a line
// with another number of fields is scan()'s error.  numeric_cols: the colClasses the script forces to "numeric".
int read_table(const std::string &path, size_t ncol, const std::vector<size_t> &numeric_cols, Table &t, std::string &err) {
    std::ifstream in(path);
    t = Table();
    t.ncol = ncol;
    t.col.assign(ncol, Column());
    for (size_t c : numeric_cols) t.col[c].is_int = false;
    std::string line;
    size_t lineno = 0;
    while (std::getline(in, line)) {
        ++lineno;
        const size_t hash = line.find('#');
        if (hash != std::string::npos) line.resize(hash);
        std::istringstream ss(line);
        std::vector<std::string> f;
        std::string tok;
        while (ss >> tok) f.push_back(tok);
        if (f.empty()) continue;
        if (f.size() != ncol) {
            err = "Error in scan(file = file, what = what, sep = sep, quote = quote, dec = dec,  : \n  line " + std::to_string(lineno) + " did not have " +
                  std::to_string(ncol) + " elements (" + path + ")";
            return 1;
        }
        for (size_t c = 0; c < ncol; ++c) {
            double x;
            if (t.col[c].is_int && parse_int32(f[c], x)) { t.col[c].v.push_back(x); continue; }
            t.col[c].is_int = false;
            char *end = nullptr;
            x = strtod(f[c].c_str(), &end);
            // strtod takes more than R's scan() does ("nan", "-nan", "inf", hex floats), and what R does take of those (NA, NaN, Inf)
            // turns a row predicate into NA, for which `x[cond, ]` emits a row of NAs instead of dropping the row.  The path's own
            // files can hold such a cell (a Cramer's V of 0 / 0 prints as "-nan"); with nothing to pin R's answer to, refuse it.
            if (f[c] == "NA" || (end != f[c].c_str() && !*end && (!std::isfinite(x) || f[c].find_first_of("xX") != std::string::npos))) {
                err = "pf_filter: cell '" + f[c] + "' in line " + std::to_string(lineno) + " of " + path +
                      " is not a finite decimal number; R reads it as NA/NaN/Inf (or not at all) and the scripts' row predicates then give NA rows -- refused (parity unpinned)";
                return 1;
            }
            if (end == f[c].c_str() || *end) {
                err = "Error in scan(file = file, what = what, sep = sep, quote = quote, dec = dec,  : \n  scan() expected 'a real', got '" + f[c] + "' (" + path + ")";
                return 1;
            }
            t.col[c].v.push_back(x);
        }
        ++t.nrow;
    }
    t.loaded = true;
    return 0;
}

void keep_rows(Table &t, const std::vector<char> &keep) {
    size_t n = 0;
    for (size_t r = 0; r < t.nrow; ++r)
        if (keep[r]) {
            for (auto &c : t.col) c.v[n] = c.v[r];
            ++n;
        }
    for (auto &c : t.col) c.v.resize(n);
    t.nrow = n;
}

// write.table(x, file, col.names = FALSE, sep = "\t", quote = FALSE, row.names = FALSE)
int write_table(const std::string &path, const Table &t) {
    std::ofstream out(path, std::ios::trunc | std::ios::binary);
    if (!out) return 1;
    std::string s;
    char num[32];
    for (size_t r = 0; r < t.nrow; ++r) {
        s.clear();
        for (size_t c = 0; c < t.ncol; ++c) {
            if (c) s += '\t';
            if (t.col[c].is_int) { snprintf(num, sizeof num, "%d", (int)t.col[c].v[r]); s += num; }
            else s += r_format_double(t.col[c].v[r]);
        }
        s += '\n';
        out << s;
    }
    return out ? 0 : 1;
}

}  // namespace

int run_filter(const FilterOptions &opt, std::string &messages, std::string &err) {
    if (opt.frequency > 0.5) {   // Filter.R:30-33
        messages += "frequency should < 0.5 \n";
        return 0;
    }
    // columns (Filter.R:41-46, 51-53, ...; Filter-multi.R:45-48, ...): A coverages, [color], isStrict, VarType, VarId, VarNum, [Cramer], VarDis
    const char *names[4] = {"_bicov.txt", "_tricov.txt", "_tetracov.txt", "_pentacov.txt"};
    Table tab[4];
    for (int a = 0; a < 4; ++a) {
        const size_t A = (size_t)a + 2;
        const std::string path = opt.inprefix + names[a];
        std::ifstream probe(path);
        if (!probe.good()) {
            messages += "This file ( " + path + " ) does not exists !\n";
            return 0;
        }
        std::vector<size_t> numeric;
        if (a == 0) {   // only the bi table has colClasses
            numeric = {0, 1};
            if (opt.multi) { numeric.push_back(2); numeric.push_back(A + 5); }
        }
        if (read_table(path, A + (opt.multi ? 7 : 5), numeric, tab[a], err)) return 1;
    }
    for (int a = 0; a < 4; ++a) {
        Table &t = tab[a];
        const size_t A = (size_t)a + 2;
        const size_t c_color = A, c_strict = A + (opt.multi ? 1 : 0), c_type = c_strict + 1, c_num = c_strict + 3,
                     c_cramer = c_strict + 4, c_dis = c_strict + (opt.multi ? 5 : 4);
        std::vector<char> keep(t.nrow, 1);
        for (size_t r = 0; r < t.nrow; ++r) {
            bool k = true;
            if (opt.simple) k = k && t.col[c_strict].v[r] == 1;                 // :82-87
            if (opt.indel) k = k && t.col[c_type].v[r] == 0;                    // :88-94
            if (opt.snp) k = k && t.col[c_type].v[r] > 0;                       // :95-101
            double first4 = 0;
            for (size_t c = 0; c < A; ++c) {
                const double x = t.col[c].v[r];
                k = k && x > (double)opt.low && x < (double)opt.up;
                if (c < 4) first4 += x;
            }
            // Filter.R:106-113: the tetra and penta tables also ask CovA + CovB + CovC + CovD < up (the penta row's fifth coverage is not
            // in the sum); Filter-multi.R has no such clause
            if (!opt.multi && A >= 4) k = k && first4 < (double)opt.up;
            k = k && t.col[c_num].v[r] < (double)opt.num && t.col[c_dis].v[r] > (double)opt.distance && t.col[c_type].v[r] < (double)opt.size;
            if (opt.multi) {
                k = k && t.col[c_cramer].v[r] > opt.cramer;                     // Filter-multi.R:106-137
                if (opt.color >= 0) k = k && t.col[c_color].v[r] == (double)opt.color;
            }
            keep[r] = k;
        }
        keep_rows(t, keep);
    }
    for (int a = 0; a < 4; ++a)
        if (write_table(opt.outprefix + names[a], tab[a])) { err = "cannot open file '" + opt.outprefix + names[a] + "'"; return 1; }
    // allele frequencies of the kept rows, allele by allele inside a table (c(bifre[1,], bifre[2,]), ...), tables in arity order
    std::vector<double> fre;
    bool any = false;
    for (int a = 0; a < 4; ++a) {
        const Table &t = tab[a];
        if (!t.nrow) continue;
        any = true;
        const size_t A = (size_t)a + 2;
        for (size_t c = 0; c < A; ++c)
            for (size_t r = 0; r < t.nrow; ++r) {
                double sum = 0;
                for (size_t x = 0; x < A; ++x) sum += t.col[x].v[r];   // (left to right, as the script adds them)
                fre.push_back(t.col[c].v[r] / sum);
            }
    }
    if (!any) {   // fre_all is still NULL: round(NULL[...], 7) is an R error; the four tables are on disk, the frequency file is not made
        err = "Error in round(fre_all[fre_all > opt$frequency & fre_all < (1 - opt$frequency)],  : \n  non-numeric argument to mathematical function";
        return 1;
    }
    std::ofstream out(opt.outprefix + "_allele_frequency.txt", std::ios::trunc | std::ios::binary);
    if (!out) { err = "cannot open file '" + opt.outprefix + "_allele_frequency.txt'"; return 1; }
    for (double x : fre)
        if (x > opt.frequency && x < 1 - opt.frequency) out << r_format_double(r_round7(x)) << '\n';
    return out ? 0 : 1;
}

namespace {
void filter_usage(bool multi) {
    std::cout << "Usage: PloidyFrost " << (multi ? "filter-multi" : "filter") << " [options]   -- Process the coverage file (script/" << (multi ? "Filter-multi.R" : "Filter.R") << ")\n\n"
              << "  -S, --simple        only simple bubble\n"
              << "  -o, --outprefix     output prefix (default : 'filtered')\n"
              << "  -i, --inprefix      input prefix (default : 'input')\n";
    if (multi) std::cout << "  -c, --color         sample color (default : -1 = all)\n";
    std::cout << "  -l, --low           lower coverage cutoff value (default : 0)\n"
              << "  -u, --up            upper coverage cutoff value (default : 10000)\n"
              << "  -I, --indel         filter indel\n"
              << "  -P, --snp           filter snp\n"
              << "  -n, --num           VarNum cutoff value (default : 10000)\n"
              << "  -d, --distance      VarDistance cutoff value (default : -1)\n"
              << "  -s, --size          VarSize cutoff value (default : 10000)\n"
              << "  -q, --frequency     frequency range(frequency,1-frequency) (default : 0.05)\n";
    if (multi) std::cout << "  -v, --cramer        Cramer'V (default : 0)\n";
}
}  // namespace

int filter_main(int argc, char **argv, bool multi) {
    FilterOptions o;
    o.multi = multi;
    struct Opt { char s; const char *l; bool flag; };
    static const Opt table[] = {{'S', "simple", true}, {'o', "outprefix", false}, {'i', "inprefix", false}, {'l', "low", false}, {'u', "up", false},
                                {'I', "indel", true}, {'P', "snp", true}, {'n', "num", false}, {'d', "distance", false}, {'s', "size", false},
                                {'q', "frequency", false}, {'c', "color", false}, {'v', "cramer", false}, {'h', "help", true}};
    for (int i = 2; i < argc; ++i) {
        std::string a = argv[i], val;
        const Opt *hit = nullptr;
        bool have_val = false;
        if (a.rfind("--", 0) == 0) {
            const size_t eq = a.find('=');
            const std::string name = a.substr(2, eq == std::string::npos ? std::string::npos : eq - 2);
            for (const Opt &t : table)
                if (name == t.l) hit = &t;
            if (eq != std::string::npos) { val = a.substr(eq + 1); have_val = true; }
        } else if (a.size() >= 2 && a[0] == '-') {
            for (const Opt &t : table)
                if (a[1] == t.s) hit = &t;
            if (a.size() > 2) { val = a.substr(2); have_val = true; }
        }
        if (!hit || (!multi && (hit->s == 'c' || hit->s == 'v'))) { std::cerr << "Error: unknown option " << a << std::endl; filter_usage(multi); return 1; }
        if (hit->s == 'h') { filter_usage(multi); return 0; }
        if (!hit->flag && !have_val) {
            if (i + 1 >= argc) { std::cerr << "Error: option " << a << " needs a value" << std::endl; return 1; }
            val = argv[++i];
        }
        switch (hit->s) {
            case 'S': o.simple = true; break;
            case 'I': o.indel = true; break;
            case 'P': o.snp = true; break;
            case 'o': o.outprefix = val; break;
            case 'i': o.inprefix = val; break;
            case 'l': o.low = atol(val.c_str()); break;
            case 'u': o.up = atol(val.c_str()); break;
            case 'n': o.num = atol(val.c_str()); break;
            case 'd': o.distance = atol(val.c_str()); break;
            case 's': o.size = atol(val.c_str()); break;
            case 'c': o.color = atol(val.c_str()); break;
            case 'q': o.frequency = atof(val.c_str()); break;
            case 'v': o.cramer = atof(val.c_str()); break;
        }
    }
    std::string messages, err;
    const int rc = run_filter(o, messages, err);
    if (!messages.empty()) std::cerr << messages;
    if (rc) std::cerr << err << std::endl << "Execution halted" << std::endl;
    return rc;
}

}  // namespace pfh
