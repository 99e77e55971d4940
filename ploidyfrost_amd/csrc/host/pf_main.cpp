// `ploidyfrost` -- command line of the MI355X build, same option surface as the reference's
// hot path (reference src/Main.cpp:124-198 getopt string and cases, :278-541 checks, :770-849 dispatch):
//     ploidyfrost -g <BifrostGraph.gfa> -d <KMCDatabase> -o <prefix> [-t T -l L -u U -z Z -M m -D d -G g -v -i]
//     ploidyfrost -g <BifrostGraph.gfa> -f <BifrostGraph.bfg_colors> -d <KMCDatabaseList> [-C <cutoffs>] -o <prefix> ...
// Output: ./PloidyFrost_output/<prefix>_*.txt, byte-identical to the reference run with -t 1.
// The coverage thresholds of the path can be derived from k-mer histograms exactly as in the reference: the
// `cutoffL` / `cutoffU` sub-commands and `-h <histogram>` (with -f: a list of histograms) with `-q <quantile>`
// (src/Main.cpp:200-277, 354-396, 721-762).  `model` (GMM ploidy inference, src/Main.cpp:636-719; the EM on the device) and
// `filter` / `filter-multi` (the row predicates of script/Filter.R, script/Filter-multi.R; host/pf_filter.cpp) are the steps behind
// the path.
#include <sys/prctl.h>
#include <sys/wait.h>
#include <signal.h>
#include <cerrno>
#include <getopt.h>
#include <sys/stat.h>
#include <unistd.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <cstring>
#include <iostream>
#include <string>
#include <fstream>
#include <thread>
#include <vector>

#include "pf_cdbg.hpp"
#include "pf_filter.hpp"
#include "pf_multi.hpp"
#include "pf_gmm_model.hpp"
#include "ploidyfrost_hip.h"

using namespace std;

namespace {
void PrintUsage() {
    cout << "Usage: PloidyFrost -g <BifrostGraph> -d <KMCDatabase> -o <outfile_prefix> ..." << endl << endl;
    cout << "parameters with required argument:" << endl << endl
         << "  -g,             Input Bifrost Graph file (GFA format)" << endl
         << "  -o,             Prefix for Output files (default : 'output')" << endl
         << "  -t,             Number of Threads (default is 1; output always follows the -t 1 order)" << endl
         << "  -d,             Load KMC Database (with -f: a file listing one database per colour)" << endl
         << "  -f,             Input Bifrost color file (BFG_COLORS format): colored, multi-sample analysis" << endl
         << "  -C,             Coverage thresholds file, one \"lower<TAB>upper\" line per colour (default : 10 1000)" << endl
         << "  -l,             Lower coverage threshold (default : 10 )" << endl
         << "  -u,             Upper coverage threshold (default : 1000 )" << endl
         << "  -z,             Maximum number of unitigs in superbubble (default : 8 )" << endl
         << "  -M,             Match score (default : 2 )" << endl
         << "  -D,             Mismatch penalty (default : -1 )" << endl
         << "  -G,             Gap penalty (default : -3 )" << endl << endl
         << "parameters with no argument:" << endl << endl
         << "  -v,             Print information messages during construction" << endl
         << "  -i,             Output Information about Bifrost graph" << endl << endl
         << "  -h,             k-mer histogram file (with -f: a list of them): thresholds = cutoffL / cutoffU of it" << endl
         << "  -q,             quantile for the upper threshold derived from -h (default : 0.998 )" << endl
         << "  --ref-threads N N > 1: text format of the reference's `-t N` run (ids from 0, allele_frequency grouped by arity)" << endl
         << "  --gpus N        one graph over N GPUs of this node (single-sample path): one process per GPU, the bubble list cut into N slices," << endl
         << "                  two small all-gathers over RCCL, every rank writes its slabs into the shared result files" << endl
         << "  --detach-teardown  return as soon as the result files are complete; a child process gives the device memory back" << endl << endl
         << "Usage: PloidyFrost cutoffL kmer_histogram_file" << endl
         << "Usage: PloidyFrost cutoffU kmer_histogram_file (quantile[<1 ,default:0.998])" << endl << endl
         << "Usage: PloidyFrost model ...          (GMM ploidy inference from the coverage / frequency files; `PloidyFrost model` prints its options)" << endl
         << "Usage: PloidyFrost filter ...         (the row predicates of script/Filter.R over <prefix>_*cov.txt; -h prints its options)" << endl
         << "Usage: PloidyFrost filter-multi ...   (script/Filter-multi.R: the colored tables, with -c colour and -v Cramer's V)" << endl;
}

// lower threshold: 1.25 x the position of the first local minimum of the histogram (src/Main.cpp:200-235)
int cutoffL(const string &file) {
    ifstream f(file);
    if (!f.is_open()) { cout << "ERROR:Open Histogram File " << file << " error!" << endl; exit(EXIT_FAILURE); }
    vector<size_t> v;
    string s;
    while (getline(f, s, '\n')) {
        const size_t pos1 = s.find("\t");
        if (pos1 == string::npos) { cerr << "Error: Histogram File is badly Formatted." << endl; exit(EXIT_FAILURE); }
        v.emplace_back((size_t)atoll(s.substr(pos1 + 1).c_str()));
    }
    size_t peak;
    for (peak = 1; peak < v.size(); peak++)
        if (v[peak - 1] < v[peak]) break;
    return int(round(1.25 * ((double)peak - 1)));
}

// upper threshold: the multiplicity below which `frequency` of the k-mers beyond the first bin lie (src/Main.cpp:236-277)
int cutoffH(const string &file, double frequency = 0.998) {
    ifstream f(file);
    if (!f.is_open()) { cout << "ERROR:Open Histogram File " << file << " error!" << endl; exit(EXIT_FAILURE); }
    vector<size_t> v;
    v.emplace_back(0);
    string s;
    while (getline(f, s, '\n')) {
        const size_t pos1 = s.find("\t");
        if (pos1 == string::npos) { cerr << "Error: Histogram File is badly Formatted." << endl; exit(EXIT_FAILURE); }
        v.emplace_back((size_t)atoll(s.substr(pos1 + 1).c_str()) + v.back());
    }
    if (v.size() <= 2) { cerr << "Error: Histogram File is badly Formatted." << endl; exit(EXIT_FAILURE); }
    const size_t cf = (size_t)(frequency * (double)(v.back() - v[1]) + (double)v[1]);
    size_t peak;
    for (peak = 2; peak < v.size(); peak++)
        if (v[peak] > cf) break;
    return (int)peak;
}

struct Options {
    string graphfile, colorfile, outprefix = "output", db, coveragefile, hist;
    size_t nb_threads = 1, complex_size = 8, ref_threads = 1, gpus = 1;
    bool verbose = false, info = false, detach_teardown = false;
    int coverage_lower = 10, coverage_upper = 1000, k = 25;
    vector<pair<int, int>> coverage_vec;
    double match = 2, mismatch = -1, gap = -3, frequency = 0.998;
};

bool file_exists(const string &p) {
    struct stat sb;
    return stat(p.c_str(), &sb) == 0;
}

void PrintModelUsage() {  // src/Main.cpp:694-718
    cout << "Usage: PloidyFrost model" << endl
         << "GMM model" << endl
         << "  -f,             Prefix of coverage files" << endl
         << "  -g,             Allele frequency file" << endl
         << "  -l,             Minimum ploidy level" << endl
         << "  -u,             Maximum ploidy level" << endl
         << "  -q,             Minimum allele frequency" << endl
         << "  -m,             Weigth minimum threshold ( each_weight > 1/(p-1)/m , default value of m is 5)" << endl
         << "  -n,             Weigth minimum threshold ( first_weight > maximum_weight/(p-1)/n , default value of n is 2)" << endl
         << "  -k,             Maximum iterations" << endl
         << "  -a,             Maximum delta" << endl
         << "  -o,             Output prefix" << endl
         << endl;
}

// `PloidyFrost model` (src/Main.cpp:636-719): option defaults, check_Model_ProgramOptions (:542-617), the fits on the device
int model_main(int argc, char **argv) {
    string graphfile, colorfile, outprefix = "output";
    int lower = 1, upper = 9, iters = 1000;
    double frequency = 0, delta = 0.01, mthreshold = 5.0, nthreshold = 2.0;
    int oc;
    while ((oc = getopt(argc, argv, "M:D:G:z:a:l:q:u:e:C:R:o:t:g:f:k:d:m:n:h:ibvpNSc")) != -1) {
        switch (oc) {
            case 'q': frequency = atof(optarg); break;
            case 'm': mthreshold = atof(optarg); break;
            case 'n': nthreshold = atof(optarg); break;
            case 'l': lower = atoi(optarg); break;
            case 'u': upper = atoi(optarg); break;
            case 'a': delta = atof(optarg); break;
            case 'o': outprefix = optarg; break;
            case 'g': graphfile = optarg; break;
            case 'f': colorfile = optarg; break;
            case 'k': iters = atoi(optarg); break;
            default: break;
        }
    }
    bool ok = true;
    if (lower > upper) { cerr << "Error:  min gauss <= max gauss  " << endl; ok = false; }
    if (lower < 1 || upper < 1) { cerr << "Error: gauss > 0  " << endl; ok = false; }
    if (frequency >= 0.5) { cerr << "Error: frequency cutoff value should < 0.5  " << endl; ok = false; }
    if (iters < 0) { cerr << "Error: iterate count should > 0 " << endl; ok = false; }
    if (delta < 0) { cerr << "Error: iterate delta should > 0 " << endl; ok = false; }
    if (mthreshold < 0) { cerr << "Error: minimum threshold should > 0 " << endl; ok = false; }
    if (nthreshold < 0) { cerr << "Error: minimum threshold should > 0 " << endl; ok = false; }
    if (colorfile.empty() && graphfile.empty()) { cout << "ERROR: input a frequency or coverage file " << endl; ok = false; }
    if (!graphfile.empty() && !file_exists(graphfile)) { cout << "ERROR: open frequency file " << graphfile << " error!" << endl; ok = false; }
    if (!colorfile.empty())
        for (const char *suf : {"_bicov.txt", "_tricov.txt", "_tetracov.txt"})
            if (!file_exists(colorfile + suf)) { cout << "ERROR: open coverage file " << colorfile + suf << " error!" << endl; ok = false; }
    if (ok && upper > PF_GMM_MAX_GAUSS) { cerr << "Error: this build fits at most " << PF_GMM_MAX_GAUSS << " Gaussians (-u)" << endl; ok = false; }
    if (!ok) { PrintModelUsage(); return 0; }
    pfh::GmmModel model;
    model.setMThreshold(mthreshold);
    model.setNThreshold(nthreshold);
    model.setMaxIterNum(iters);
    model.setMaxDeltaNum(delta);
    if (!colorfile.empty() ? model.readCovFile(colorfile, frequency) : model.readFreFile(graphfile, frequency)) {
        cout << model.error() << endl;
        exit(EXIT_FAILURE);
    }
    string err;
    if (pfh::run_model(model, lower, upper, outprefix, err)) {
        cout << err << endl;
        exit(EXIT_FAILURE);
    }
    return 0;
}
}  // namespace

int main(int argc, char **argv) {
    if (argc < 2) { PrintUsage(); return 0; }
    if (!strcmp(argv[1], "model")) return model_main(argc, argv);
    if (!strcmp(argv[1], "filter")) return pfh::filter_main(argc, argv, false);         // script/Filter.R
    if (!strcmp(argv[1], "filter-multi")) return pfh::filter_main(argc, argv, true);    // script/Filter-multi.R
    if (!strcmp(argv[1], "cutoffL")) {  // src/Main.cpp:721-730
        if (argc != 3) { cout << "Usage:PloidyFrost cutoffL kmer_histogram_file" << endl; exit(EXIT_FAILURE); }
        cout << max(10, cutoffL(argv[2])) << endl;
        return 0;
    }
    if (!strcmp(argv[1], "cutoffU")) {  // src/Main.cpp:731-762 (no newline after the value when a quantile is given)
        const char *usage = "Usage:PloidyFrost cutoffU kmer_histogram_file (quantile[<1 ,default:0.998]) ";
        if (argc == 3) {
            cout << cutoffH(argv[2]) << endl;
        } else if (argc == 4) {
            double y;
            try { y = stod(argv[3]); } catch (const exception &) { cout << usage << endl; exit(EXIT_FAILURE); }
            if (y >= 1) { cout << usage << endl; exit(EXIT_FAILURE); }
            cout << cutoffH(argv[2], y);
        } else {
            cout << usage << endl;
            exit(EXIT_FAILURE);
        }
        return 0;
    }
    Options opt;
    // --ref-threads N (this build's own switch, taken out of argv before getopt sees it): write the text format of the reference's
    // `-t N` functions -- with N > 1: ids and var_count from 0, allele_frequency rows grouped by arity per bubble -- whatever -t says
    for (int i = 1; i < argc; ++i) {
        if (strcmp(argv[i], "--ref-threads") == 0 && i + 1 < argc) {
            opt.ref_threads = (size_t)atoi(argv[i + 1]);
            for (int j = i; j + 2 <= argc; ++j) argv[j] = j + 2 < argc ? argv[j + 2] : nullptr;
            argc -= 2;
            --i;
        } else if (strcmp(argv[i], "--gpus") == 0 && i + 1 < argc) {
            opt.gpus = (size_t)std::max(1, atoi(argv[i + 1]));
            for (int j = i; j + 2 <= argc; ++j) argv[j] = j + 2 < argc ? argv[j + 2] : nullptr;
            argc -= 2;
            --i;
        } else if (strcmp(argv[i], "--detach-teardown") == 0) {
            opt.detach_teardown = true;
            for (int j = i; j + 1 <= argc; ++j) argv[j] = j + 1 < argc ? argv[j + 1] : nullptr;
            argc -= 1;
            --i;
        }
    }
    int oc;
    while ((oc = getopt(argc, argv, "M:D:G:z:a:l:q:u:e:C:R:o:t:g:f:k:d:m:n:h:ibvpNSc")) != -1) {
        switch (oc) {
            case 'z': opt.complex_size = (size_t)atoi(optarg); break;
            case 'M': opt.match = atof(optarg); break;
            case 'D': opt.mismatch = atof(optarg); break;
            case 'G': opt.gap = atof(optarg); break;
            case 'u': opt.coverage_upper = atoi(optarg);  // falls through, as in the reference (:149-153)
            case 'C': opt.coveragefile = optarg; break;
            case 'h': opt.hist = optarg; break;
            case 'g': opt.graphfile = optarg; break;
            case 'f': opt.colorfile = optarg; break;
            case 'o': opt.outprefix = optarg; break;
            case 'l': opt.coverage_lower = atoi(optarg); break;
            case 't': opt.nb_threads = (size_t)atoi(optarg); break;
            case 'k': opt.k = atoi(optarg); break;
            case 'v': opt.verbose = true; break;
            case 'd': opt.db = optarg; break;
            case 'i': opt.info = true; break;
            case 'q': opt.frequency = atof(optarg); break;
            case 'm': case 'n': case 'a': case 'b': case 'p': break;  // accepted, no effect on this path
            default:
                cout << "Invalid option" << endl;
                PrintUsage();
                exit(EXIT_FAILURE);
        }
    }
    // check_ProgramOptions (:278-541), single-sample subset
    bool ok = true;
    const size_t max_threads = std::thread::hardware_concurrency();
    if ((long)opt.nb_threads <= 0) { cerr << "Error: Number of threads cannot be less than or equal to 0." << endl; ok = false; }
    if (opt.nb_threads > max_threads) { cerr << "Error: Number of threads cannot be greater than or equal to " << max_threads << "." << endl; ok = false; }
    if (opt.frequency < 0 || opt.frequency > 1) { cerr << "Error: frequency cutoff value should be between 0 and 1 " << endl; ok = false; }
    size_t kmc_db_num = 0;
    if (opt.db.empty()) { cerr << "Error: Need input a kmc database prefix!\n"; ok = false; }
    else if (opt.colorfile.empty()) {
        if (!file_exists(opt.db + ".kmc_pre") || !file_exists(opt.db + ".kmc_suf")) {
            cerr << "Error: Could not read the input kmc database " << opt.db << "." << endl;
            ok = false;
        }
        if (!opt.hist.empty()) {  // :354-358
            opt.coverage_lower = max(10, cutoffL(opt.hist));
            opt.coverage_upper = cutoffH(opt.hist, opt.frequency);
        }
    } else {
        // :326-353: the -d file lists one database per line
        ifstream in(opt.db);
        if (!file_exists(opt.db) || in.fail()) { ok = false; }
        else {
            string name;
            while (getline(in, name, '\n')) {
                ++kmc_db_num;
                if (!file_exists(name + ".kmc_pre") || !file_exists(name + ".kmc_suf")) {
                    cerr << "Error: Could not read the input kmc database " << name << "." << endl;
                    ok = false;
                    break;
                }
            }
        }
        if (!opt.hist.empty()) {  // :359-396: one histogram file per line, one per database
            ifstream hin(opt.hist);
            if (!file_exists(opt.hist) || hin.fail()) { ok = false; }
            else {
                string name;
                size_t i = 0;
                while (getline(hin, name, '\n')) {
                    opt.coverage_vec.push_back({max(10, cutoffL(name)), cutoffH(name, opt.frequency)});
                    if (opt.coverage_vec[i].first > opt.coverage_vec[i].second) { cerr << "Error: lower cutoff need be smaller than upper cutoff " << endl; ok = false; }
                    i++;
                }
                if (i != kmc_db_num) { cerr << "ERROR: the numbers of kmc databases and hist files are not equal! " << endl; exit(EXIT_FAILURE); }
            }
        } else
        // :398-455: -C "lower\tupper" per database, default (10, 1000)
        if (!opt.coveragefile.empty()) {
            ifstream cin_(opt.coveragefile);
            if (!file_exists(opt.coveragefile) || cin_.fail()) { ok = false; }
            else {
                string line;
                size_t i = 0;
                while (getline(cin_, line, '\n')) {
                    const size_t pos1 = line.find("\t");
                    if (pos1 == string::npos) { cerr << "Error: Coverage File is badly Formatted." << endl; exit(EXIT_FAILURE); }
                    opt.coverage_vec.push_back({(int)atoll(line.substr(0, pos1).c_str()), (int)atoll(line.substr(pos1 + 1).c_str())});
                    if (opt.coverage_vec[i].first < 0 || opt.coverage_vec[i].second < 0) { cerr << "Error: Filter coverage need a positive number." << endl; ok = false; }
                    if (opt.coverage_vec[i].first > opt.coverage_vec[i].second) { cerr << "Error: lower cutoff need be smaller than upper cutoff " << endl; ok = false; }
                    i++;
                }
                if (i != kmc_db_num) { cerr << "ERROR: the numbers of kmc databases and coverages are not equal! " << endl; exit(EXIT_FAILURE); }
            }
        } else {
            opt.coverage_vec.insert(opt.coverage_vec.end(), kmc_db_num, pair<int, int>(10, 1000));
        }
        if (!file_exists(opt.colorfile)) { cerr << "Error: The input color file does not exist." << endl; ok = false; }
    }
    if (opt.complex_size < 4) { cerr << "Error: Maximum number of unitigs in superbubble is at least 4 !" << endl; ok = false; }
    if (opt.mismatch > opt.match) { cerr << "Error: Mismatch penalty should be smaller than match score !" << endl; ok = false; }
    if (opt.gap > opt.match) { cerr << "Error: Gap penalty should be smaller than match score !" << endl; ok = false; }
    if (opt.outprefix.empty()) { cerr << "Error: No output filename prefix given." << endl; ok = false; }
    if (opt.coverage_lower < 0 || opt.coverage_upper < 0) { cerr << "Error: Filter coverage need a positive number." << endl; ok = false; }
    if (opt.coverage_lower > opt.coverage_upper) { cerr << "Error: lower cutoff need be smaller than upper cutoff " << endl; ok = false; }
    if (opt.graphfile.empty()) { cerr << "Error: No graph file was provided in input." << endl; ok = false; }
    else if (!file_exists(opt.graphfile)) { cerr << "Error: The graph file does not exist." << endl; ok = false; }
    if (!ok) { PrintUsage(); return 0; }

    if (!opt.colorfile.empty() && opt.gpus > 1) { cerr << "Error: --gpus cuts the single-sample path (-g / -d); the colored one runs on one GPU" << endl; return 1; }
    if (!opt.colorfile.empty()) {  // src/Main.cpp:775-810
        pfh::ColoredUnitigSet cdbg;
        auto t0 = std::chrono::steady_clock::now();
        if (!cdbg.read(opt.graphfile, opt.colorfile, opt.nb_threads, opt.verbose)) {
            cout << "ColoredCDBG::read(): Graph could not be loaded! Exit. (" << cdbg.err << ")" << endl;
            exit(EXIT_FAILURE);
        }
        cout << "ColoredCDBG::read(): Graph loading successful" << endl;

        cout << "CCDBG: Graph loading Real time : " << std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() << "s" << endl;
        if (cdbg.getNbColors() != kmc_db_num) {
            cerr << "CCDBG::CCDBG():Error: " << kmc_db_num << " kmc databases listed for " << cdbg.getNbColors() << " colors" << endl;
            exit(EXIT_FAILURE);
        }
        pfh::CCDBG g(cdbg, opt.complex_size, opt.match, opt.mismatch, opt.gap, opt.db, opt.nb_threads);
        if (opt.verbose && cdbg.graph.n_abundant)
            cout << "ColoredCDBG::read(): " << cdbg.graph.n_abundant << " k-length unitigs are abundant k-mers (numbered last, in Bifrost's hash table order)" << endl;
        auto die = [&]() {
            cerr << g.error() << endl;
            exit(EXIT_FAILURE);
        };
        if (!g.good()) die();
        g.set_threads((unsigned)opt.nb_threads);
        g.set_overlap_output(true);
        if (g.setUnitigId(opt.outprefix, opt.graphfile, opt.nb_threads)) die();
        if (opt.info && g.printInfo(opt.verbose, opt.outprefix)) die();
        if (g.findSuperBubble_multithread_ptr(opt.outprefix, opt.nb_threads)) die();
        for (size_t i = 0; i < opt.coverage_vec.size(); i++) {
            cout << "CCDBG:: Database " << i << " Minimum Coverage:" << opt.coverage_vec[i].first << endl;
            cout << "CCDBG:: Maximum Coverage:" << opt.coverage_vec[i].second << endl;
        }
        if (g.ploidyEstimation_multithread_ptr(opt.outprefix, opt.coverage_vec, opt.nb_threads)) die();
        if (opt.verbose) {
            const pfh::PhaseTimes &t = g.times();
            printf("[device] candidates %llu  bfs %.3fs replay %.3fs | cov %.3fs tasks %.3fs (%llu) align %.3fs (%llu jobs) "
                   "sites %.3fs (%llu strings) format %.3fs write %.3fs\n",
                   (unsigned long long)t.candidates, t.bfs_device_s, t.replay_s, t.cov_device_s, t.tasks_s, (unsigned long long)t.tasks,
                   t.align_s, (unsigned long long)t.align_jobs, t.sites_s, (unsigned long long)t.site_strings, t.format_s, t.write_s);
        }
        return 0;
    }

    // Giving 12 GB of device memory and the pinned buffers back is 0.35-0.4 s of driver work at process exit (5 M unitigs) --
    // a third of the whole run.  On request (--detach-teardown) that work is done by a child process:
    // the parent returns the moment the child reports that the last result file is complete, the child finishes its exit with
    // nobody waiting for it -- and still holds its device memory for those 0.4 s, which a scheduler that starts the next job on
    // the parent's return has to know; hence not the default.  (Forked here, before the first thread and the first device call.)
    int done_fd = -1;
    if (opt.detach_teardown) {
        int fds[2];
        if (pipe(fds) == 0) {
            cout.flush();
            fflush(nullptr);
            const pid_t pid = fork();
            if (pid > 0) {
                close(fds[1]);
                unsigned char code = 0;
                ssize_t n;
                do n = read(fds[0], &code, 1); while (n < 0 && errno == EINTR);
                if (n == 1) _exit(code);
                int st = 0;   // the child left without reporting: its exit status is the run's
                while (waitpid(pid, &st, 0) < 0 && errno == EINTR) {}
                _exit(WIFEXITED(st) ? WEXITSTATUS(st) : 128 + (WIFSIGNALED(st) ? WTERMSIG(st) : 0));
            }
            if (pid == 0) {
                close(fds[0]);
                done_fd = fds[1];
                prctl(PR_SET_PDEATHSIG, SIGTERM);   // an interrupted parent takes the run with it
            } else {
                close(fds[0]);
                close(fds[1]);
            }
        }
    }
    // --gpus N: the other ranks are forked here, before anything has touched the GPU (pf_multi.hpp); from here on every rank runs the
    // same program on its own device, rank 0 speaks
    pfh::RankGroup ranks;
    if (opt.gpus > 1) {
        if (opt.ref_threads > 1) { cerr << "Error: --gpus and --ref-threads do not go together" << endl; return 1; }
        cout.flush();
        fflush(nullptr);
        if (!ranks.start((int)opt.gpus)) { cerr << "Error: --gpus " << opt.gpus << ": " << ranks.err << endl; return 1; }
        if (ranks.rank != 0) {   // the other ranks say nothing
            if (!freopen("/dev/null", "w", stdout)) {}
        }
    }
    // the device context and the count table are built on a helper thread while the graph file is read
    pfh::CountsLoader counts;
    counts.start(ranks.device(), opt.db);
    pfh::UnitigSet graph;
    std::string err;
    auto t0 = std::chrono::steady_clock::now();
    const bool trace_main = getenv("PF_TRACE_LOAD") != nullptr;
    auto mark = [&](const char *what) {
        if (trace_main) fprintf(stderr, "[main] %-28s %.3fs since start\n", what, std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count());
    };
    // K-GFA: the file is mapped and its header read here; the segments are parsed and packed on the device by the CDBG constructor
    // (PF_GFA=host: the host loader).  Abundant k-mers are decided there as well (K-MINZ).
    static const bool host_gfa = [] { const char *e = getenv("PF_GFA"); return e && !strcmp(e, "host"); }();
    if (!(host_gfa ? graph.load_gfa(opt.graphfile, err, true) : graph.open_gfa(opt.graphfile, err))) {
        cout << "CompactedDBG::read(): Graph could not be loaded! Exit. (" << err << ")" << endl;
        exit(EXIT_FAILURE);
    }
    if (host_gfa) {
        cout << "CompactedDBG::read(): Graph loading successful" << endl;
        cout << "CDBG: Graph loading Real time : " << std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() << "s" << endl;
    } else if (pf_ctx *c = counts.wait_context()) {
        graph.parse_on_device(c);   // beside the count table's ingest; a refusal is reported by the constructor
    }
    mark("graph file read");

    // (on the heap and never destroyed: at the end of main the process leaves through quick_exit -- giving 12 GB of device and
    // pinned memory back piece by piece takes longer than some of the phases)
    pfh::CDBG &g = *new pfh::CDBG(graph, opt.complex_size, opt.match, opt.mismatch, opt.gap, opt.db, ranks.device(), false, &counts);
    mark("graph + counts on the device");
    auto die = [&]() {
        cerr << g.error() << endl;
        exit(EXIT_FAILURE);
    };
    if (!g.good() && ranks.world <= 1) {
        if (g.error().rfind("CompactedDBG::read()", 0) == 0) { cout << g.error() << endl; exit(EXIT_FAILURE); }   // (the ingest's word for it)
        die();
    }
    if (ranks.world > 1) {   // a rank without its graph or table says so to the others before they wait for it in the communicator
        auto say0 = [&](const std::string &what) {   // (one write per line: the ranks share the terminal)
            const std::string line = "rank " + std::to_string(ranks.rank) + ": " + what + "\n";
            if (write(2, line.data(), line.size()) < 0) {}
        };
        if (!g.good()) say0(g.error());
        if (!ranks.agree(g.good(), "load")) {
            if (g.good()) say0(ranks.err);
            if (ranks.rank == 0) (void)ranks.finish();
            _exit(EXIT_FAILURE);
        }
    }
    if (!host_gfa) {
        cout << "CompactedDBG::read(): Graph loading successful" << endl;
        cout << "CDBG: Graph loading Real time : " << std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() << "s" << endl;
    }
    if (opt.verbose && graph.n_abundant)
        cout << "CompactedDBG::read(): " << graph.n_abundant << " k-length unitigs are abundant k-mers (numbered last, in Bifrost's hash table order)" << endl;
    g.set_threads((unsigned)opt.nb_threads);
    g.set_overlap_output(true);
    if (opt.ref_threads > 1 && g.set_reference_threads(opt.ref_threads)) die();
    if (getenv("PF_BFS_HUGE_ON_DEVICE")) g.set_third_tier_on_host(false);   // experiments: giant traversals on one wavefront each
    if (ranks.world > 1) {
        // ---- one graph over the GPUs of the node (the protocol of ploidyfrost_amd/dist.py from the C++ side) ----
        // every step that ends in a collective: the rank's own outcome first, then whether everybody is still there (pf_multi.hpp,
        // RankGroup::agree) -- a rank that failed alone must not leave the others waiting in RCCL
        // (one write per line: the ranks share the terminal)
        auto say = [&](const std::string &what) {
            const std::string line = "rank " + std::to_string(ranks.rank) + ": " + what + "\n";
            if (write(2, line.data(), line.size()) < 0) {}
        };
        auto leave = [&](const std::string &why) {
            say(why);
            if (ranks.rank == 0) (void)ranks.finish();
            _exit(EXIT_FAILURE);
        };
        auto together = [&](bool failed, const std::string &why, const char *stage) {
            if (failed) say(why);
            if (!ranks.agree(!failed, stage)) leave(failed ? std::string("leaving") : ranks.err);
        };
        if (!ranks.connect(g.device())) leave("--gpus: " + ranks.err);
        mark("communicator");
        g.set_write_super_bubble(ranks.rank == 0);
        int bad = 0;
        if (ranks.rank == 0) {
            bad = g.setUnitigId(opt.outprefix, opt.graphfile, opt.nb_threads);
            if (!bad && opt.info) bad = g.printInfo(opt.verbose, opt.outprefix);
        }
        if (!bad) bad = g.findSuperBubble_multithread_ptr(opt.outprefix, opt.nb_threads);   // every rank: the same state everywhere
        together(bad != 0, g.error(), "findSuperBubble");
        mark("findSuperBubble");
        cout << "CDBG:: Minimum Coverage:" << opt.coverage_lower << endl;
        cout << "CDBG:: Maximum Coverage:" << opt.coverage_upper << endl;
        const auto tp0 = std::chrono::steady_clock::now();
        cout << "CDBG::PloidyEstimation():  Analyzing superbubbles to generate sites' information" << endl;
        uint64_t nb = 0;
        bad = g.ploidy_select(opt.coverage_lower, opt.coverage_upper, nb);
        const uint64_t W = (uint64_t)ranks.world, R = (uint64_t)ranks.rank;
        const uint64_t t0b = nb / W * R + std::min<uint64_t>(R, nb % W), t1b = t0b + nb / W + (R < nb % W ? 1 : 0);
        uint64_t called = 0;
        if (!bad) bad = g.ploidy_align(t0b, t1b, called);
        together(bad != 0, g.error(), "align");
        std::vector<uint64_t> all((size_t)W * 18);
        if (!ranks.gather(g.device(), &called, 1, all.data())) leave("--gpus: " + ranks.err);
        uint64_t base = 0;
        for (uint64_t r = 0; r < R; ++r) base += all[r];
        uint64_t mine[18];
        bad = g.ploidy_text(base, mine, mine + PF_CALL_STREAMS);
        together(bad != 0, g.error(), "text");
        if (!ranks.gather(g.device(), mine, 18, all.data())) leave("--gpus: " + ranks.err);
        uint64_t offsets[PF_CALL_STREAMS] = {}, totals[PF_CALL_STREAMS] = {}, sums[8] = {};
        for (uint64_t r = 0; r < W; ++r) {
            for (int s_ = 0; s_ < PF_CALL_STREAMS; ++s_) {
                if (r < R) offsets[s_] += all[r * 18 + (uint64_t)s_];
                totals[s_] += all[r * 18 + (uint64_t)s_];
            }
            for (int c = 0; c < 8; ++c) sums[c] += all[r * 18 + PF_CALL_STREAMS + (uint64_t)c];
        }
        bad = g.ploidy_write(opt.outprefix, offsets, totals, true);
        together(bad != 0, g.error(), "write");
        uint64_t done = 1;   // nobody returns before everybody's slabs are in the files
        if (!ranks.gather(g.device(), &done, 1, all.data())) leave("--gpus: " + ranks.err);
        mark("PloidyEstimation");
        if (ranks.rank != 0) {
            fflush(nullptr);
            _exit(0);
        }
        printf("CDBG::PloidyEstimation():  Real time : %gs\n", std::chrono::duration<double>(std::chrono::steady_clock::now() - tp0).count());
        printf("CDBG::PloidyEstimation(): Alleles in SuperBubbles  :\t2 :%llu\t3 :%llu\t4 :%llu\t5 :%llu\n", (unsigned long long)sums[0],
               (unsigned long long)sums[1], (unsigned long long)sums[2], (unsigned long long)sums[3]);
        if (sums[5]) printf("CDBG::PloidyEstimation(): Sites' Average Coverage:%d\n", (int)(sums[4] / sums[5]));
        if (opt.verbose) printf("[ranks]  %d ranks, %llu bubbles in the list, %llu called; this rank's slice %llu .. %llu\n", ranks.world, (unsigned long long)nb,
                                (unsigned long long)sums[6], (unsigned long long)t0b, (unsigned long long)t1b);
        cout.flush();
        fflush(nullptr);
        const int rc = ranks.finish();
        if (rc) { cerr << "a rank ended with status " << rc << endl; _exit(rc); }
        _exit(0);
    }
    if (g.setUnitigId(opt.outprefix, opt.graphfile, opt.nb_threads)) die();
    if (opt.info && g.printInfo(opt.verbose, opt.outprefix)) die();
    mark("setUnitigId");
    if (g.findSuperBubble_multithread_ptr(opt.outprefix, opt.nb_threads)) die();
    mark("findSuperBubble");
    cout << "CDBG:: Minimum Coverage:" << opt.coverage_lower << endl;
    cout << "CDBG:: Maximum Coverage:" << opt.coverage_upper << endl;
    if (g.ploidyEstimation_multithread_ptr(opt.outprefix, opt.coverage_lower, opt.coverage_upper, opt.nb_threads)) die();
    mark("PloidyEstimation");
    if (opt.verbose) {
        const pfh::PhaseTimes &t = g.times();
        printf("[device] candidates %llu (big tier %llu)  bfs %.3fs replay %.3fs | cov %.3fs tasks %.3fs (%llu) align %.3fs (%llu jobs) "
               "sites %.3fs (%llu strings) format %.3fs write %.3fs\n",
               (unsigned long long)t.candidates, (unsigned long long)t.bfs_deferred, t.bfs_device_s, t.replay_s, t.cov_device_s,
               t.tasks_s, (unsigned long long)t.tasks, t.align_s, (unsigned long long)t.align_jobs, t.sites_s,
               (unsigned long long)t.site_strings, t.format_s, t.write_s);
        printf("[bfs]    traversals > 4096 unitigs: %llu (sum of seen %llu, largest %llu); committed by the replay: %llu (largest %llu)\n",
               (unsigned long long)t.bfs_large, (unsigned long long)t.bfs_large_seen, (unsigned long long)t.bfs_max_seen,
               (unsigned long long)t.bfs_large_used, (unsigned long long)t.bfs_large_used_max);
        printf("[host]   scan %.3fs | align: build %.3fs device-call %.3fs post %.3fs choose %.3fs\n", t.scan_s, t.align_build_s,
               t.align_device_s, t.align_post_s, t.align_choose_s);
    }
    // every result file is complete (PloidyEstimation joined the background writers): leave without walking the destructors
    cout.flush();
    cerr.flush();
    fflush(nullptr);
    mark("done");
    if (done_fd >= 0) {
        prctl(PR_SET_PDEATHSIG, 0);
        const unsigned char ok = 0;
        if (write(done_fd, &ok, 1) != 1) {}
        close(done_fd);
    }
    // (PF_ORDERLY_EXIT: leave through exit() -- a profiler that writes its report from an exit handler needs that)
    if (getenv("PF_ORDERLY_EXIT")) exit(0);
    _exit(0);
}
