// pfh::GmmModel: the reference's GmmModel (src/GmmModel.hpp:5-49, src/GmmModel.cpp) with the EM iterations on the device
// (pf_gmm_upload / pf_gmm_fit, ../pf_gmm.hip).  Same method names and argument meaning; errors come back as a status plus
// error() instead of exit().  The readers run on the host -- they are the reference's text parsing, quirks included -- and
// need no device; emIterate() creates the device context on first use and fails without a gfx950 device (no CPU fit).
#pragma once
#include <cstddef>
#include <ostream>
#include <string>
#include <vector>

struct pf_ctx;

namespace pfh {

class GmmModel {
public:
    explicit GmmModel(int device = 0) : device_(device) {}
    ~GmmModel();
    GmmModel(const GmmModel &) = delete;
    GmmModel &operator=(const GmmModel &) = delete;

    void resize(size_t g);
    void setMThreshold(double m) { m_thre = m; }
    void setNThreshold(double n) { n_thre = n; }
    void setMaxIterNum(int i) { emMaxIter = i; }
    void setMaxDeltaNum(double i) { emMaxDelta = i; }
    int emIterate();   // 0 = ok
    double getLogLikelihood() const { return logLikelihood; }
    double computeAIC() {
        aic = (2 * ((double)gauss * 2 - 1) - 2 * logLikelihood) / (double)allele_fre.size();
        return aic;
    }
    double getAIC() const { return aic; }
    void readData(const std::vector<double> &v) { allele_fre = v; uploaded_ = false; }
    int readFreFile(const std::string &filename, const double &freq);
    int readCovFile(const std::string &prefix, const double &freq);
    void output(std::ostream &os) const;
    void print() const;

    const std::vector<double> &values() const { return allele_fre; }
    const std::vector<double> &getWeights() const { return weights; }
    const std::vector<double> &getMeans() const { return means; }
    const std::vector<double> &getVars() const { return vars; }
    unsigned iterations() const { return iterations_; }
    const std::string &error() const { return err_; }
    pf_ctx *device_context() const { return ctx_; }

private:
    int fail(const std::string &m) { err_ = m; return 1; }
    std::vector<double> allele_fre;
    size_t gauss = 0;
    std::vector<double> weights, means, vars;
    double m_thre = 5.0, n_thre = 2.0;
    int emMaxIter = 1000;
    double emMaxDelta = 0.01;
    double logLikelihood = 0, aic = 0;
    unsigned iterations_ = 0;
    int device_;
    pf_ctx *ctx_ = nullptr;
    bool uploaded_ = false;
    std::string err_;
};

// `PloidyFrost model` after option parsing (src/Main.cpp:644-692): fits gauss = lo .. hi, writes <outprefix>_model_result.txt
int run_model(GmmModel &model, int lo, int hi, const std::string &outprefix, std::string &err);

}  // namespace pfh
