// ORACLE (test infrastructure only -- see pf_oracle.h).  Command-line front end used by the
// tests and by bench.py's cpu_baseline leg when the real reference binary (oracle/_ref) is
// not available:   pf_oracle_cli -g graph.gfa -d kmc_prefix -o prefix [-O outdir] [-l L] [-u U]
//                                [-z Z] [-M m] [-D d] [-G g]
// colored twin:    pf_oracle_cli -g graph.gfa -f colors_dump.txt -d db_list.txt [-C cutoffs.txt] -o prefix ...
// Prints per-phase wall times in the reference's format (src/CDBG.cpp:217-220, 1683-1686).
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <string>
#include <vector>

#include "pf_oracle.h"

int main(int argc, char **argv) {
    std::string gfa, db, colors, cutoffs, prefix = "output", outdir = "PloidyFrost_output";
    int lower = 10, upper = 1000;
    unsigned z = 8;
    double M = 2, D = -1, G = -3;
    for (int i = 1; i + 1 < argc; i += 2) {
        std::string a = argv[i];
        const char *v = argv[i + 1];
        if (a == "-g") gfa = v;
        else if (a == "-d") db = v;
        else if (a == "-f") colors = v;
        else if (a == "-C") cutoffs = v;
        else if (a == "-o") prefix = v;
        else if (a == "-O") outdir = v;
        else if (a == "-l") lower = atoi(v);
        else if (a == "-u") upper = atoi(v);
        else if (a == "-z") z = (unsigned)atoi(v);
        else if (a == "-M") M = atof(v);
        else if (a == "-D") D = atof(v);
        else if (a == "-G") G = atof(v);
        else if (a == "-t") {}
        else { fprintf(stderr, "unknown option %s\n", a.c_str()); return 2; }
    }
    if (gfa.empty() || db.empty()) { fprintf(stderr, "need -g and -d\n"); return 2; }
    using clk = std::chrono::steady_clock;
    auto t0 = clk::now();
    pfo_ctx *c = colors.empty() ? pfo_open(gfa.c_str(), db.c_str()) : pfo_open_colored(gfa.c_str(), colors.c_str(), db.c_str());
    if (!c) { fprintf(stderr, "pf_oracle: %s\n", pfo_last_error()); return 1; }
    auto t1 = clk::now();
    printf("ORACLE: load time : %gs (%u unitigs)\n", std::chrono::duration<double>(t1 - t0).count(), pfo_num_unitigs(c));
    if (pfo_set_unitig_id(c, outdir.c_str(), prefix.c_str())) { fprintf(stderr, "pf_oracle: %s\n", pfo_last_error()); return 1; }
    auto t2 = clk::now();
    uint64_t nb = 0;
    if (pfo_find_superbubbles(c, outdir.c_str(), prefix.c_str(), z, &nb)) { fprintf(stderr, "pf_oracle: %s\n", pfo_last_error()); return 1; }
    auto t3 = clk::now();
    printf("CDBG::findSuperBubble():  Real time : %gs\n", std::chrono::duration<double>(t3 - t2).count());
    printf("CDBG::findSuperBubble(): %llu  SuperBubbles Found\n", (unsigned long long)nb);
    uint64_t allele[4], cc, cn;
    int rc;
    if (colors.empty()) {
        rc = pfo_ploidy_estimation(c, outdir.c_str(), prefix.c_str(), lower, upper, M, D, G, allele, &cc, &cn);
    } else {
        // Main.cpp:398-455: "lower\tupper" per colour, default (10, 1000)
        std::vector<int> lo(pfo_num_colors(c), 10), up(pfo_num_colors(c), 1000);
        if (!cutoffs.empty()) {
            std::ifstream in(cutoffs);
            std::string line;
            for (size_t i = 0; i < lo.size() && std::getline(in, line); ++i) {
                size_t t = line.find('\t');
                if (t == std::string::npos) { fprintf(stderr, "Error: Coverage File is badly Formatted.\n"); return 1; }
                lo[i] = (int)atoll(line.substr(0, t).c_str());
                up[i] = (int)atoll(line.substr(t + 1).c_str());
            }
        }
        rc = pfo_ploidy_estimation_colored(c, outdir.c_str(), prefix.c_str(), lo.data(), up.data(), M, D, G, allele, &cc, &cn);
    }
    if (rc) { fprintf(stderr, "pf_oracle: %s\n", pfo_last_error()); return 1; }
    auto t4 = clk::now();
    printf("CDBG::PloidyEstimation():  Real time : %gs\n", std::chrono::duration<double>(t4 - t3).count());
    printf("CDBG::PloidyEstimation(): Alleles in SuperBubbles  :\t2 :%llu\t3 :%llu\t4 :%llu\t5 :%llu\n",
           (unsigned long long)allele[0], (unsigned long long)allele[1], (unsigned long long)allele[2],
           (unsigned long long)allele[3]);
    if (cn) printf("CDBG::PloidyEstimation(): Sites' Average Coverage:%d\n", (int)(cc / cn));
    pfo_close(c);
    return 0;
}
