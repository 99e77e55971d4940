#include "pf_gmm_model.hpp"

#include <cfloat>
#include <cstdlib>
#include <fstream>
#include <iostream>

#include "ploidyfrost_hip.h"

namespace pfh {

GmmModel::~GmmModel() {
    if (ctx_) pf_destroy(ctx_);
}

// src/GmmModel.cpp:8-20
void GmmModel::resize(size_t g) {
    gauss = g;
    vars.assign(g, 0.01);
    means.resize(g);
    weights.assign(g, (double)1 / (double)g);
    for (size_t i = 1; i <= g; ++i) means[i - 1] = (double)i / (double)(g + 1);
}

// src/GmmModel.cpp:371-385, the loop itself runs on the device
int GmmModel::emIterate() {
    if (gauss < 1 || gauss > PF_GMM_MAX_GAUSS) return fail("GmmModel: the device fit takes 1.." + std::to_string(PF_GMM_MAX_GAUSS) + " Gaussians");
    if (!ctx_) {
        if (pf_create(device_, &ctx_) != PF_OK) {
            const char *e = pf_last_error(nullptr);
            ctx_ = nullptr;
            return fail(std::string("GmmModel: no device context (") + (e ? e : "?") + "); the fit runs on the GPU only");
        }
    }
    if (!uploaded_) {
        if (pf_gmm_upload(ctx_, allele_fre.data(), allele_fre.size()) != PF_OK) return fail(std::string("GmmModel: ") + pf_last_error(ctx_));
        uploaded_ = true;
    }
    uint32_t it = 0;
    if (pf_gmm_fit(ctx_, (uint32_t)gauss, m_thre, n_thre, emMaxIter, emMaxDelta, weights.data(), means.data(), vars.data(), &logLikelihood, &it) != PF_OK)
        return fail(std::string("GmmModel: ") + pf_last_error(ctx_));
    iterations_ = it;
    computeAIC();
    return 0;
}

// src/GmmModel.cpp:240-257.  The read that runs into the end of a file ending in white space extracts nothing and leaves
// `a` as it was, so the last value of such a file counts twice -- as in the reference.  A token that is not a number makes
// the reference loop forever; here it is an error.
int GmmModel::readFreFile(const std::string &filename, const double &frequency) {
    std::ifstream fileReader(filename, std::ios::in);
    if (!fileReader.is_open()) return fail("ERROR: open frequency file error!");
    uploaded_ = false;
    double a;
    while (!fileReader.eof()) {
        fileReader >> a;
        if (fileReader.fail() && !fileReader.eof()) return fail("ERROR: " + filename + " holds something that is not a number");
        if (a >= frequency && a <= 1 - frequency) allele_fre.push_back(a);
    }
    return 0;
}

namespace {
// the first n tab-terminated integer fields of a line (src/GmmModel.cpp:41-52 and its copies): false when the line has
// fewer than n tabs
bool leading_ints(const std::string &s, int n, int *out) {
    size_t from = 0;
    for (int i = 0; i < n; ++i) {
        const size_t t = s.find('\t', from);
        if (t == std::string::npos) return false;
        out[i] = atoi(s.c_str() + from);
        from = t + 1;
    }
    return true;
}
}  // namespace

// src/GmmModel.cpp:21-239.  Kept as written there: the frequency test divides INTEGERS (cov / cov_sum is 0 unless one
// allele holds every read), "min" of three or more alleles only compares neighbours, and the penta file is closed before
// it is read, so its rows never count.  A row whose coverages sum to 0 stops the reference with a division fault; here it
// is an error.
int GmmModel::readCovFile(const std::string &name, const double &frequency) {
    allele_fre.clear();
    uploaded_ = false;
    std::ifstream in[4];
    const char *suffix[4] = {"_bicov.txt", "_tricov.txt", "_tetracov.txt", "_pentacov.txt"};
    for (int f = 0; f < 4; ++f) in[f].open(name + suffix[f], std::ios::in);
    for (int f = 0; f < 4; ++f)
        if (!in[f].is_open()) return fail("Model::readCovFile() : Open cov file error");
    std::string s;
    for (int f = 0; f < 3; ++f) {  // bi, tri, tetra
        const int n = f + 2;
        while (std::getline(in[f], s, '\n')) {
            int cov[5];
            if (!leading_ints(s, n, cov)) continue;
            int cov_sum = 0;
            for (int i = 0; i < n; ++i) cov_sum += cov[i];
            if (cov_sum >= 10000) continue;
            if (cov_sum == 0) return fail("Model::readCovFile() : a row of " + name + suffix[f] + " sums to 0 (the reference divides by it)");
            int lead = cov[0];  // "min": bi rows test the first allele, longer rows walk neighbour pairs
            for (int i = 1; i < n && n > 2; ++i)
                if (cov[i] < cov[i - 1]) lead = cov[i];
            const int q = lead / cov_sum;
            if (q >= frequency && q <= 1 - frequency)
                for (int i = 0; i < n; ++i) allele_fre.push_back(double(cov[i]) / cov_sum);
        }
    }
    return 0;
}

// src/GmmModel.cpp:350-369
void GmmModel::output(std::ostream &os) const {
    os << "ploidy : " << gauss + 1 << "\tgauss : " << gauss << std::endl;
    os << "avg loglikelihood : " << getLogLikelihood() / allele_fre.size() << std::endl;
    os << "AIC : " << getAIC() << std::endl;
    os << "means :\t" << std::endl << "\t";
    for (size_t i = 0; i < gauss; i++) os << means[i] << "\t";
    os << std::endl;
    os << "weights :\t" << std::endl << "\t";
    for (size_t i = 0; i < gauss; i++) os << weights[i] << "\t";
    os << std::endl;
    os << "variances :\t" << std::endl << "\t";
    for (size_t i = 0; i < gauss; i++) os << vars[i] << "\t";
    os << std::endl << "-----------------------------------" << std::endl;
}

// src/GmmModel.cpp:335-349 (computeAIC() there, the same value)
void GmmModel::print() const {
    std::ostream &os = std::cout;
    os << "ploidy:\t" << gauss + 1 << "\tgauss:\t" << gauss << std::endl;
    os << "avg loglikelihood:\t" << getLogLikelihood() / allele_fre.size() << std::endl;
    os << "AIC:\t" << getAIC() << std::endl;
    os << "means:\t" << std::endl << "\t";
    for (size_t i = 0; i < gauss; i++) os << means[i] << "\t";
    os << std::endl;
    os << "weights:\t" << std::endl << "\t";
    for (size_t i = 0; i < gauss; i++) os << weights[i] << "\t";
    os << std::endl;
    os << "variances:\t" << std::endl << "\t";
    for (size_t i = 0; i < gauss; i++) os << vars[i] << "\t";
    os << std::endl << std::endl;
}

// src/Main.cpp:659-690
int run_model(GmmModel &model, int lo, int hi, const std::string &outprefix, std::string &err) {
    std::ofstream outfile(outprefix + "_model_result.txt", std::ios::out | std::ios::trunc);
    if (!outfile.is_open()) { err = "ERROR: open output file " + outprefix + "_model_result.txt error!"; return 1; }
    double maxll = DBL_MIN, minaic = DBL_MAX, ll_p = 0, aic_p = 0;
    for (int i = lo; i <= hi; i++) {
        model.resize((size_t)i);
        if (model.emIterate()) { err = model.error(); return 1; }
        model.output(outfile);
        if (model.getLogLikelihood() > maxll) { maxll = model.getLogLikelihood(); ll_p = i + 1; }
        if (model.getAIC() < minaic) { minaic = model.getAIC(); aic_p = i + 1; }
    }
    outfile << "max loglikelihood : " << maxll << "\tploidy : " << ll_p << std::endl;
    outfile << "min AIC : " << minaic << "\tploidy : " << aic_p << std::endl;
    outfile << "estimated ploidy level is : " << aic_p << std::endl;
    return 0;
}

}  // namespace pfh
