// TEST INFRASTRUCTURE ONLY -- CPU restatement of `PloidyFrost model` (reference src/GmmModel.cpp, src/Main.cpp:636-692):
// the readers, the EM loop and the result file, sequential fp64 exactly as the reference sums them.  Pinned against the
// reference binary's own <prefix>_model_result.txt on tests/golden/model/* (tests/test_model_cpu.py).
#include <cfloat>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <sstream>
#include <string>
#include <vector>

#include "pf_oracle.h"

namespace {

struct Gmm {
    std::vector<double> x, w, mean, var;
    size_t g = 0;
    double m_thre = 5, n_thre = 2, max_delta = 0.01, ll = 0, aic = 0;
    int max_iter = 1000;
    unsigned iterations = 0;
    std::string err;

    // src/GmmModel.hpp:14-17
    static double pdf(double mean, double var, double v) { return 1 / (sqrt(2 * M_PI * var)) * exp(-(pow(v - mean, 2) / (2 * var))); }

    void resize(size_t n) {  // src/GmmModel.cpp:8-20
        g = n;
        w.assign(n, 0);
        mean.assign(n, 0);
        var.assign(n, 0);
        for (size_t i = 1; i <= n; ++i) {
            mean[i - 1] = (double)i / (n + 1);
            w[i - 1] = (double)1 / n;
            var[i - 1] = 0.01;
        }
    }
    double loglik() const {  // :259-276
        double sum = 0;
        for (double af : x) {
            double s = 0;
            for (size_t i = 0; i < g; ++i) s += w[i] * pdf(mean[i], var[i], af);
            if (s == 0.0) s = DBL_MIN;
            sum += log(s);
        }
        return sum;
    }
    void em_step() {  // :277-334 (the means are computed there and then dropped: they never move)
        std::vector<double> part(g, 0), gsum(g, 0), vsum(g, 0);
        double sum = 0;
        for (double af : x) {
            double row = 0;
            for (size_t i = 0; i < g; ++i) {
                part[i] = w[i] * pdf(mean[i], var[i], af);
                if (part[i] == 0.0) part[i] = DBL_MIN;
                row += part[i];
            }
            for (size_t i = 0; i < g; ++i) {
                part[i] /= row;
                gsum[i] += part[i];
                vsum[i] += part[i] * pow(af - mean[i], 2);
                sum += part[i];
            }
        }
        std::vector<double> nv(g), nw(g);
        for (size_t i = 0; i < g; ++i) {
            double v = 1 / gsum[i] * vsum[i];
            if (v == 0.0) v = DBL_MIN;
            nv[i] = v;
            nw[i] = gsum[i] / sum;
        }
        double mx = nw[0], mn = nw[0];
        for (size_t i = 1; i < g; ++i) {  // std::max_element / min_element: first of equals
            if (mx < nw[i]) mx = nw[i];
            if (nw[i] < mn) mn = nw[i];
        }
        if (mx != nw[0] && mx != nw[g - 1]) {
            if (mn < (double)1 / g / m_thre) return;
            if (mn < mx / g / n_thre) return;
        }
        var = nv;
        w = nw;
    }
    void iterate() {  // :371-385
        ll = loglik();
        double last = ll, delta = DBL_MAX;
        size_t count = 0;
        while (delta > max_delta && count < (size_t)max_iter) {
            em_step();
            last = ll;
            ll = loglik();
            delta = ll - last;
            ++count;
        }
        iterations = (unsigned)count;
        aic = (2 * ((double)g * 2 - 1) - 2 * ll) / x.size();
    }
    void output(std::ostream &os) const {  // :350-369
        os << "ploidy : " << g + 1 << "\tgauss : " << g << std::endl;
        os << "avg loglikelihood : " << ll / x.size() << std::endl;
        os << "AIC : " << aic << std::endl;
        os << "means :\t" << std::endl << "\t";
        for (size_t i = 0; i < g; i++) os << mean[i] << "\t";
        os << std::endl << "weights :\t" << std::endl << "\t";
        for (size_t i = 0; i < g; i++) os << w[i] << "\t";
        os << std::endl << "variances :\t" << std::endl << "\t";
        for (size_t i = 0; i < g; i++) os << var[i] << "\t";
        os << std::endl << "-----------------------------------" << std::endl;
    }
    bool read_fre(const char *file, double freq) {  // :240-257
        std::ifstream f(file);
        if (!f.is_open()) { err = "cannot open frequency file"; return false; }
        double a;
        while (!f.eof()) {
            f >> a;
            if (f.fail() && !f.eof()) { err = "not a number in the frequency file (the reference never returns from it)"; return false; }
            if (a >= freq && a <= 1 - freq) x.push_back(a);
        }
        return true;
    }
    bool read_cov(const char *prefix, double freq) {  // :21-239
        x.clear();
        const std::string name(prefix);
        std::ifstream bi(name + "_bicov.txt"), tri(name + "_tricov.txt"), tet(name + "_tetracov.txt"), pen(name + "_pentacov.txt");
        if (!bi.is_open() || !tri.is_open() || !tet.is_open() || !pen.is_open()) { err = "cannot open the coverage files"; return false; }
        std::string s;
        auto fields = [&](int n, int *c) {
            size_t at = 0;
            for (int i = 0; i < n; ++i) {
                size_t t = s.find("\t", at);
                if (t == std::string::npos) return false;
                c[i] = atoi(s.c_str() + at);
                at = t + 1;
            }
            return true;
        };
        int c[4];
        while (std::getline(bi, s)) {
            if (!fields(2, c)) continue;
            const int sum = c[0] + c[1];
            if (sum >= 10000) continue;
            if (sum == 0) { err = "coverage row sums to 0"; return false; }
            if (c[0] / sum >= freq && c[0] / sum <= 1 - freq) { x.push_back(double(c[0]) / sum); x.push_back(double(c[1]) / sum); }
        }
        while (std::getline(tri, s)) {
            if (!fields(3, c)) continue;
            const int sum = c[0] + c[1] + c[2];
            if (sum >= 10000) continue;
            if (sum == 0) { err = "coverage row sums to 0"; return false; }
            int mn = c[0];
            if (c[1] < c[0]) mn = c[1];
            if (c[2] < c[1]) mn = c[2];
            if (mn / sum >= freq && mn / sum <= 1 - freq)
                for (int i = 0; i < 3; ++i) x.push_back(double(c[i]) / sum);
        }
        while (std::getline(tet, s)) {
            if (!fields(4, c)) continue;
            const int sum = c[0] + c[1] + c[2] + c[3];
            if (sum >= 10000) continue;
            if (sum == 0) { err = "coverage row sums to 0"; return false; }
            int mn = c[0];
            if (c[1] < c[0]) mn = c[1];
            if (c[2] < c[1]) mn = c[2];
            if (c[3] < c[2]) mn = c[3];
            if (mn / sum >= freq && mn / sum <= 1 - freq)
                for (int i = 0; i < 4; ++i) x.push_back(double(c[i]) / sum);
        }
        // the reference closes the penta stream before its loop (:172): no penta row is ever read
        return true;
    }
};

std::string g_err;

}  // namespace

extern "C" {

struct pfo_gmm { Gmm m; };

pfo_gmm *pfo_gmm_open(void) { return new pfo_gmm(); }
void pfo_gmm_close(pfo_gmm *p) { delete p; }
const char *pfo_gmm_error(const pfo_gmm *p) { return p->m.err.c_str(); }
int pfo_gmm_read_fre(pfo_gmm *p, const char *file, double freq) { return p->m.read_fre(file, freq) ? 0 : 1; }
int pfo_gmm_read_cov(pfo_gmm *p, const char *prefix, double freq) { return p->m.read_cov(prefix, freq) ? 0 : 1; }
void pfo_gmm_set_values(pfo_gmm *p, const double *v, uint64_t n) { p->m.x.assign(v, v + n); }
uint64_t pfo_gmm_size(const pfo_gmm *p) { return p->m.x.size(); }
void pfo_gmm_values(const pfo_gmm *p, double *out) { for (size_t i = 0; i < p->m.x.size(); ++i) out[i] = p->m.x[i]; }
void pfo_gmm_fit(pfo_gmm *p, uint32_t gauss, double m_thre, double n_thre, int max_iter, double max_delta, double *w, double *mean,
                 double *var, double *loglik, double *aic, uint32_t *iterations) {
    Gmm &m = p->m;
    m.m_thre = m_thre;
    m.n_thre = n_thre;
    m.max_iter = max_iter;
    m.max_delta = max_delta;
    m.resize(gauss);
    m.iterate();
    for (uint32_t i = 0; i < gauss; ++i) { w[i] = m.w[i]; mean[i] = m.mean[i]; var[i] = m.var[i]; }
    *loglik = m.ll;
    *aic = m.aic;
    *iterations = m.iterations;
}
// src/Main.cpp:659-690
int pfo_gmm_run(pfo_gmm *p, int lo, int hi, double m_thre, double n_thre, int max_iter, double max_delta, const char *outprefix) {
    Gmm &m = p->m;
    m.m_thre = m_thre;
    m.n_thre = n_thre;
    m.max_iter = max_iter;
    m.max_delta = max_delta;
    std::ofstream out(std::string(outprefix) + "_model_result.txt", std::ios::out | std::ios::trunc);
    if (!out.is_open()) { m.err = "cannot write the result file"; return 1; }
    double maxll = DBL_MIN, minaic = DBL_MAX, ll_p = 0, aic_p = 0;
    for (int i = lo; i <= hi; ++i) {
        m.resize((size_t)i);
        m.iterate();
        m.output(out);
        if (m.ll > maxll) { maxll = m.ll; ll_p = i + 1; }
        if (m.aic < minaic) { minaic = m.aic; aic_p = i + 1; }
    }
    out << "max loglikelihood : " << maxll << "\tploidy : " << ll_p << std::endl;
    out << "min AIC : " << minaic << "\tploidy : " << aic_p << std::endl;
    out << "estimated ploidy level is : " << aic_p << std::endl;
    return 0;
}

}  // extern "C"
