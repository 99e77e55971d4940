// pfh::CDBG::ploidy_estimation: variant calling over the bubbles findSuperBubble left open (single-sample and colored).
#include <fcntl.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <cmath>
#include <condition_variable>
#include <cstdio>
#include <cstring>
#include <ctime>
#include <deque>
#include <mutex>
#include <set>
#include <stdexcept>
#include <thread>

#include "pf_cdbg_impl.hpp"
#include "pf_parallel.hpp"

namespace pfh {

// ---- sorting of the paths -----------------------------------------------------------------
namespace {
// sortSeq_simple (reference src/CDBG.cpp:482-551): the reference's own non-stable quicksort --
// descending mean coverage, ties by descending reference string.  The exact swap sequence
// matters for ties, so it is the same partition scheme.
void sort_inner(const UnitigSet &g, double *cov, uint32_t *ov, int low, int high) {
    if (high <= low) return;
    auto gt = [&](int a, int b) {  // strcmp(ref(a), ref(b)) > 0
        return g.seq(ov[a] >> 1).compare(g.seq(ov[b] >> 1)) > 0;
    };
    auto lt = [&](int a, int b) { return g.seq(ov[a] >> 1).compare(g.seq(ov[b] >> 1)) < 0; };
    int i = low, j = high;
    for (;;) {
        while (cov[i] >= cov[low]) {
            if (cov[i] > cov[low] || gt(i, low)) i++;
            else break;
            if (i == high) break;
        }
        while (cov[j] <= cov[low]) {
            if (cov[j] < cov[low] || lt(j, low)) j--;
            else break;
            if (j == low) break;
        }
        if (i >= j) break;
        std::swap(cov[i], cov[j]);
        std::swap(ov[i], ov[j]);
    }
    std::swap(cov[low], cov[j]);
    std::swap(ov[low], ov[j]);
    sort_inner(g, cov, ov, low, j - 1);
    sort_inner(g, cov, ov, j + 1, high);
}

// sortSeq_branching (reference src/CDBG.cpp:417-480): descending length, ties by descending strcmp
void sort_paths(std::vector<std::string> &v, int low, int high) {
    if (high <= low) return;
    auto before = [&](int a, int b) {  // a sorts strictly before b
        return v[a].size() > v[b].size() || (v[a].size() == v[b].size() && v[a].compare(v[b]) > 0);
    };
    int i = low, j = high;
    for (;;) {
        while (v[i].size() >= v[low].size()) {
            if (before(i, low)) i++;
            else break;
            if (i == high) break;
        }
        while (v[j].size() <= v[low].size()) {
            if (before(low, j)) j--;
            else break;
            if (j == low) break;
        }
        if (i >= j) break;
        std::swap(v[i], v[j]);
    }
    std::swap(v[low], v[j]);
    sort_paths(v, low, j - 1);
    sort_paths(v, j + 1, high);
}

}  // namespace

// ---- ploidyEstimation (reference src/CDBG.cpp:1101-1705) -----------------------------------
// The reference handles one bubble at a time; here the same work is laid out in phases so that
// each phase is either one batched device call or an embarrassingly parallel loop over bubbles:
//   scan     (sequential, light)  replay of the driver loop: which endpoint owns which bubble,
//                                 in which order -- depends only on state bits, CSR, reference strings
//   paths    (parallel)           path strings: oriented inner unitigs / two-stack DFS + quicksorts
//   align    (GPU rounds)         SeqAlign::SequenceAlignment for all bubbles at once
//   sites    (parallel + GPU)     per-site k-length strings -> one K-STRCOV launch
//   format   (parallel)           text of the eleven files, concatenated in bubble order
int CDBG::ploidyEstimation_multithread_ptr(const std::string &outpre, const int &lower, const int &upper, const size_t &thr) {
    return ploidy_estimation(outpre, {{lower, upper}}, thr);
}

namespace {
// computeCramerVCoefficient (reference src/CCDBG.cpp:330-366) on two rows of a [colour][allele] coverage matrix
double cramer_v(const double *A, const double *B, size_t n_alleles) {
    double n = 0, nA = 0, nB = 0, chi = 0;
    uint8_t count = 0;
    double p[256];
    for (size_t i = 0; i < n_alleles; ++i) {
        nA += A[i];
        nB += B[i];
        p[i] = A[i] + B[i];
        n = n + p[i];
        if (p[i] != 0) ++count;
    }
    if (count < 2) return 0;
    for (size_t i = 0; i < n_alleles; ++i) {
        if (p[i] == 0) continue;
        const double exA = nA * p[i] / n, exB = nB * p[i] / n;
        chi += std::pow(A[i] - exA, 2) / exA;
        chi += std::pow(B[i] - exB, 2) / exB;
    }
    return std::sqrt(chi / n);
}
// the maximum over all colour pairs (src/CCDBG.cpp:2964-2970, 3285-3291); m is [n_colors][stride]
double max_cramer_v(const double *m, size_t n_colors, size_t stride, size_t n_alleles) {
    double c = 0;
    for (size_t ci = 0; ci + 1 < n_colors; ++ci)
        for (size_t cj = ci + 1; cj < n_colors; ++cj) c = std::max(c, cramer_v(m + ci * stride, m + cj * stride, n_alleles));
    return c;
}

// colored sortSeq_simple (reference src/CCDBG.cpp:368-480): descending number of colours, then descending length,
// then descending reference string; cov is [n_colors][stride], permuted along with the unitigs
void sort_inner_colored(const UnitigSet &g, size_t *pc, uint32_t *ov, double *cov, size_t n_colors, size_t stride, int low, int high) {
    if (high <= low) return;
    auto ref = [&](int x) { return g.seq(ov[x] >> 1); };
    auto swap_at = [&](int a, int b) {
        std::swap(pc[a], pc[b]);
        std::swap(ov[a], ov[b]);
        for (size_t c = 0; c < n_colors; ++c) std::swap(cov[c * stride + a], cov[c * stride + b]);
    };
    int i = low, j = high;
    for (;;) {
        while (pc[i] >= pc[low]) {
            if (pc[i] > pc[low]) i++;
            else if (ref(i).size() > ref(low).size()) i++;
            else if (ref(i).size() == ref(low).size() && ref(i).compare(ref(low)) > 0) i++;
            else break;
            if (i == high) break;
        }
        while (pc[j] <= pc[low]) {
            if (pc[j] < pc[low]) j--;
            else if (ref(j).size() < ref(low).size()) j--;
            else if (ref(j).size() == ref(low).size() && ref(j).compare(ref(low)) < 0) j--;
            else break;
            if (j == low) break;
        }
        if (i >= j) break;
        swap_at(i, j);
    }
    swap_at(low, j);
    sort_inner_colored(g, pc, ov, cov, n_colors, stride, low, j - 1);
    sort_inner_colored(g, pc, ov, cov, n_colors, stride, j + 1, high);
}
}  // namespace

int CDBG::ploidy_estimation(const std::string &outpre, const std::vector<std::pair<int, int>> &cutoff, const size_t &thr) {
    if (status_) return status_;
    if (mt_format_ && !resident_path())
        return fail(PF_ERR_ARG, std::string(tag_) + "::PloidyEstimation(): the -t > 1 output format needs the resident calling pipeline");
    if (mt_format_ && col_)
        return fail(PF_ERR_ARG, std::string(tag_) + "::PloidyEstimation(): the -t > 1 output format is the single-sample path's");
    if (resident_path()) {
        if (cutoff.size() != (col_ ? col_->n_colors : 1u))
            return fail(PF_ERR_ARG, std::string(tag_) + "::PloidyEstimation(): one (lower, upper) cutoff" + (col_ ? " per colour" : "") + " is required");
        return ploidy_estimation_resident(outpre, cutoff, thr);
    }
    const auto t_all = clk::now();
    clock_t c0 = clock();
    if (!quiet_) printf("%s::PloidyEstimation():  Analyzing superbubbles to generate sites' information\n", tag_);
    if (write_files_ && ensure_dir()) return status_;
    g_.ensure_text();   // this pipeline compares and copies sequences on the host
    if (sync_state_to_host()) return status_;   // ... and reads the state findSuperBubble left (committed on the device)
    const uint32_t N = g_.n();
    const int k = g_.k;
    const bool colored = col_ != nullptr;
    const uint32_t C = colored ? col_->n_colors : 1;
    if (cutoff.size() != C) return fail(PF_ERR_ARG, std::string(tag_) + "::PloidyEstimation(): one (lower, upper) cutoff per colour is required");
    const uint32_t low = (uint32_t)cutoff[0].first, up = (uint32_t)cutoff[0].second;
    std::vector<uint32_t> lows(C), ups(C);
    for (uint32_t c = 0; c < C; ++c) { lows[c] = (uint32_t)cutoff[c].first; ups[c] = (uint32_t)cutoff[c].second; }
    const unsigned T = threads_ ? threads_ : (unsigned)std::max<size_t>(thr, 1);
    times_.cov_device_s = times_.tasks_s = times_.align_s = times_.sites_s = times_.format_s = times_.write_s = 0;
    times_.tasks = times_.align_jobs = times_.site_strings = 0;
    times_.align_build_s = times_.align_device_s = times_.align_post_s = times_.align_choose_s = times_.scan_s = 0;

    // the ten result files are appended to batch by batch (stage 2 of the pipeline below), one writer per file; they are
    // opened -- truncating what an earlier pass left there costs milliseconds -- by a helper thread while coverage and
    // the scan run
    static const char *kArity[4] = {"bi", "tri", "tetra", "penta"};
    struct OutFile {
        std::string name;
        FILE *f = nullptr;
        uint64_t bytes = 0;
        int rc = 0;
    };
    std::vector<OutFile> files(10);
    files[0].name = outpre + "_allele_frequency.txt";
    files[1].name = outpre + "_alignseq.txt";
    for (int a = 0; a < 4; ++a) {
        files[2 + a].name = outpre + "_" + kArity[a] + "fre.txt";
        files[6 + a].name = outpre + "_" + kArity[a] + "cov.txt";
    }
    auto close_files = [&] {
        for (OutFile &of : files)
            if (of.f) { fclose(of.f); of.f = nullptr; }
    };
    int open_failed = -1;
    std::thread opener;
    if (write_files_)
        opener = std::thread([&] {
            for (size_t i = 0; i < files.size(); ++i) {
                files[i].f = fopen((outdir_ + "/" + files[i].name).c_str(), "wb");
                if (!files[i].f) { open_failed = (int)i; return; }
            }
        });
    struct OpenerGuard {  // every early return below must not leave the helper running or files open
        std::thread &t;
        std::vector<OutFile> &f;
        ~OpenerGuard() {
            if (t.joinable()) t.join();
            for (OutFile &of : f)
                if (of.f) { fclose(of.f); of.f = nullptr; }
        }
    } opener_guard{opener, files};

    const bool trace = getenv("PF_TRACE_PLOIDY") != nullptr;
    auto tp = [&](const char *what) { if (trace) fprintf(stderr, "[ploidy] %-28s %.2f ms\n", what, since(t_all) * 1e3); };
    // C1 for every unitig (and, colored, every colour) in one launch (the reference calls readCov per use); already
    // there when findSuperBubble started it behind its replay
    auto t0 = clk::now();
    int st = PF_OK;
    if (!cov_ready_) st = launch_coverage();
    cov_ready_ = false;
    if (st != PF_OK) return fail(st, std::string(tag_) + "::PloidyEstimation(): " + cov_err_);
    const uint64_t *cov_sum = bx_.cov_sum.p;
    const uint32_t *cov_min = bx_.cov_min.p;
    const uint32_t *cov_max = bx_.cov_max.p;
    const uint8_t *cov_miss = bx_.cov_miss.p;
    times_.cov_device_s = since(t0);
    tp("coverage done");
    auto mean_of = [&](uint32_t u) { return (double)cov_sum[u] / (double)g_.len_km(u); };
    // coverage slot of an oriented unitig: one per unitig, or (database without canonical counting) one per orientation
    const bool per_strand = !colored && !both_strands_;
    auto cslot = [&](uint32_t ov) -> size_t { return per_strand && (ov & 1) ? (size_t)N + (ov >> 1) : (size_t)(ov >> 1); };
    auto mean_of_ov = [&](uint32_t ov) { return (double)cov_sum[cslot(ov)] / (double)g_.len_km(ov >> 1); };
    // readCovUni(u, low, up, c) of src/CCDBG.cpp:123-156: (sum / len, true) iff every k-mer is in colour c's database
    // with low < count < up, else (0, false)
    auto cov_ok_c = [&](uint32_t c, uint32_t u) {
        const size_t o = (size_t)c * N + u;
        return !cov_miss[o] && cov_min[o] > lows[c] && cov_max[o] < ups[c];
    };
    auto mean_of_c = [&](uint32_t c, uint32_t u) { return (double)cov_sum[(size_t)c * N + u] / (double)g_.len_km(u); };
    auto missing = [&](uint32_t u) -> int {
        if (!cov_miss[u]) return 0;
        return fail(PF_ERR_MISSING_KMER, "CDBG::readCov(): a kmer of unitig " + std::to_string(u + 1) + " can not found .");
    };
    auto succ_row = [&](uint32_t ov) { return &succ_[(size_t)ov * 4]; };
    auto first_succ = [&](uint32_t ov) -> uint32_t {
        const uint32_t *r = succ_row(ov);
        for (int b = 0; b < 4; ++b)
            if (r[b] != NONE) return r[b];
        return NONE;
    };

    // the ten per-site streams + alignseq
    uint64_t var_count = 0;
    allele_[0] = allele_[1] = allele_[2] = allele_[3] = 0;
    core_cov_ = core_num_ = 0;

    // ---- scan, part A (parallel): what each open endpoint side would do if it is still open when the
    //      driver loop of src/CDBG.cpp:1146-1186 reaches it -- exit, ownership, coverage gate, sorted
    //      inner unitigs.  Depends only on static state (partners, strict/complex bits, CSR, strings).
    t0 = clk::now();
    struct SideRec {
        Task t;
        uint8_t plus_side;
        uint8_t kind;  // 1 complex, 2 the other endpoint owns the bubble, 3 processed here
        uint8_t aligned;
        uint8_t err;   // 1 missing k-mer, 2 exit unreachable (raised only if the side is still open)
        uint32_t err_unitig;
    };
    constexpr size_t UCH = 4096;
    const size_t n_uch = n_chunks_of(N, UCH);
    std::vector<std::vector<SideRec>> side_chunks(n_uch);
    // colored strict bubbles: [colour][4] mean-coverage matrices, one pool per chunk (Task::cov_ref points into them)
    std::vector<std::vector<double>> cov_pools(colored ? n_uch : 0);
    parallel_chunks(N, UCH, T, [&](size_t ci, size_t ub, size_t ue) {
        std::vector<SideRec> &out = side_chunks[ci];
        for (uint32_t u = (uint32_t)ub; u < (uint32_t)ue; ++u) {
            const uint8_t f = flags_[u];
            if ((f & 3) == 0) continue;
            for (int side = 0; side < 2; ++side) {
                const bool ps = side == 0;
                if (!(f & (ps ? B_PLUS : B_MINUS))) continue;
                SideRec r;
                r.plus_side = ps;
                r.kind = 0;
                r.aligned = 0;
                r.err = 0;
                r.err_unitig = 0;
                r.t.u = u;
                if (f & (ps ? B_COMPLEX_P : B_COMPLEX_M)) {
                    r.kind = 1;
                    out.push_back(r);
                    continue;
                }
                const uint32_t uo = 2 * u + (ps ? 0 : 1);
                const bool strict = (f & (ps ? B_STRICT_P : B_STRICT_M)) != 0;
                if (!colored && cov_miss[cslot(uo)]) { r.err = 1; r.err_unitig = u; out.push_back(r); continue; }  // core = readCov(u), u oriented
                uint32_t exit_ov;
                if (strict) {
                    exit_ov = first_succ(first_succ(uo));
                } else {
                    const uint32_t want = ps ? plus_[u] : minus_[u];
                    exit_ov = first_succ(uo);
                    // (bounded: a walk longer than the graph means the partner is not on the first-successor chain)
                    for (uint32_t steps = 0; exit_ov != NONE && (exit_ov >> 1) + 1 != want; ++steps) {
                        if (steps > N) { exit_ov = NONE; break; }
                        exit_ov = first_succ(exit_ov);
                    }
                }
                if (exit_ov == NONE) { r.err = 2; out.push_back(r); continue; }
                const uint32_t eu = exit_ov >> 1;
                r.t.entrance_ov = uo;
                r.t.exit_ov = exit_ov;
                r.t.strict = strict;
                if (g_.seq(u).compare(g_.seq(eu)) < 0) {  // the other endpoint owns this bubble
                    r.kind = 2;
                    out.push_back(r);
                    continue;
                }
                r.kind = 3;
                if (colored) {
                    // src/CCDBG.cpp:2838-2853: the per-colour means are summed until a colour fails its range test
                    // (the `flag == false;` there is a no-op, so the bubble is processed regardless)
                    double core = 0;
                    for (uint32_t c = 0; c < C; ++c) {
                        if (!cov_ok_c(c, u)) break;
                        core += mean_of_c(c, u);
                    }
                    r.t.core_mean = core;
                    bool flag = true;
                    if (strict) {  // :2867-2931
                        Task &t = r.t;
                        std::vector<double> &pool = cov_pools[ci];
                        const size_t base = pool.size();
                        pool.resize(base + (size_t)C * 4, 0.0);
                        double *m = pool.data() + base;
                        size_t pc[4] = {0, 0, 0, 0};
                        uint32_t path = 0;
                        for (int b = 0; b < 4 && flag; ++b) {
                            const uint32_t w = succ_row(uo)[b];
                            if (w == NONE) continue;
                            const uint32_t wu = w >> 1;
                            t.inner[t.n_inner++] = w;
                            size_t j = 0;
                            for (uint32_t c = 0; c < C; ++c) {
                                if (!col_->full(wu, c)) continue;
                                j++;
                                if (cov_ok_c(c, wu)) m[(size_t)c * 4 + path] = mean_of_c(c, wu);
                                else { flag = false; break; }
                            }
                            if (!flag) break;
                            if (col_->size_total[wu] != j * (uint64_t)g_.len_km(wu)) { flag = false; break; }  // a colour on part of it
                            pc[path++] = j;
                        }
                        if (flag) {  // some colour must see more than one of the paths
                            flag = false;
                            for (uint32_t c = 0; c < C && !flag; ++c) {
                                int nz = 0;
                                for (uint32_t q = 0; q < path; ++q) nz += m[(size_t)c * 4 + q] != 0.0;
                                flag = nz > 1;
                            }
                        }
                        if (flag) {
                            sort_inner_colored(g_, pc, t.inner, m, C, 4, 0, (int)path - 1);
                            t.n_cov = (uint8_t)path;
                            t.cov_ref = ((uint64_t)ci << 32) | base;
                        } else {
                            pool.resize(base);
                        }
                    }
                    r.aligned = flag;
                    out.push_back(r);
                    continue;
                }
                r.t.core_mean = per_strand ? mean_of_ov(uo) : mean_of(u);
                bool aligned = true;
                if (strict) {
                    Task &t = r.t;
                    for (int b = 0; b < 4 && aligned && !r.err; ++b) {
                        const uint32_t w = succ_row(uo)[b];
                        if (w == NONE) continue;
                        t.inner[t.n_inner++] = w;
                        if (cov_miss[cslot(w)]) { r.err = 1; r.err_unitig = w >> 1; break; }
                        if (cov_min[cslot(w)] > low && cov_min[cslot(w)] < up) {
                            const double mcov = mean_of_ov(w);
                            t.cov[t.n_cov++] = mcov;
                            t.cov_sum += mcov;
                        } else {
                            aligned = false;
                        }
                    }
                    if (aligned && !r.err) {
                        // the reference also reads the predecessors' coverage and drops it (:1224-1239)
                        for (int b = 0; b < 4; ++b) {
                            const uint32_t w = pred_[(size_t)uo * 4 + b];
                            if (w != NONE && cov_miss[cslot(w)]) { r.err = 1; r.err_unitig = w >> 1; break; }
                        }
                        if (!r.err) sort_inner(g_, t.cov, t.inner, 0, (int)t.n_cov - 1);
                    }
                }
                r.aligned = aligned;
                out.push_back(r);
            }
        }
    });
    // ---- scan, part B (sequential, light): the driver loop itself -- a side is handled only if its bit
    //      is still set when its unitig comes up; handling a bubble closes both endpoint sides (:1656-1679)
    std::vector<Task> all_tasks;
    const auto t_serial = clk::now();
    // the sequential pass only decides; the kept tasks are gathered afterwards, chunks in parallel
    std::vector<std::vector<uint32_t>> kept(n_uch);
    for (size_t ci = 0; ci < n_uch; ++ci) {
        const std::vector<SideRec> &chunk = side_chunks[ci];
        std::vector<uint32_t> &keep = kept[ci];
        for (uint32_t ri = 0; ri < chunk.size(); ++ri) {
            const SideRec &r = chunk[ri];
            const uint32_t u = r.t.u;
            const uint8_t own = r.plus_side ? B_PLUS : B_MINUS;
            if (!(flags_[u] & own)) continue;
            if (r.kind == 1) { flags_[u] &= (uint8_t)~own; continue; }
            if (r.err == 1) { missing(r.err_unitig); return status_; }
            if (r.err == 2) return fail(PF_ERR_ARG, "CDBG::PloidyEstimation(): exit of a bubble is not reachable");
            flags_[u] &= (uint8_t)~own;
            if (r.kind == 2) continue;
            if (r.aligned) keep.push_back(ri);
            flags_[r.t.exit_ov >> 1] &= (uint8_t) ~(plus_of(r.t.exit_ov) ? B_MINUS : B_PLUS);
        }
    }
    times_.scan_serial_s = since(t_serial);
    {
        std::vector<size_t> base(n_uch + 1, 0);
        for (size_t ci = 0; ci < n_uch; ++ci) base[ci + 1] = base[ci] + kept[ci].size();
        all_tasks.resize(base[n_uch]);
        parallel_chunks(n_uch, 1, T, [&](size_t ci, size_t, size_t) {
            Task *dst = all_tasks.data() + base[ci];
            for (uint32_t ri : kept[ci]) *dst++ = side_chunks[ci][ri].t;
            std::vector<SideRec>().swap(side_chunks[ci]);
        });
    }
    const size_t skip_before = 0;
    times_.scan_s += since(t0);
    tp("scan done");

    const size_t CHUNK = std::max<size_t>(batch_bubbles_, 1);  // bubbles per batch
    constexpr size_t PCH = 256;     // bubbles per parallel work item
    struct ChunkOut {
        std::string s_var, allfre, fre[4], cov[4];
        uint64_t allele[4] = {0, 0, 0, 0}, core_cov = 0, core_num = 0;
    };
    std::vector<std::vector<ChunkOut>> all_outs;  // per batch, per work item: concatenated at the end
    struct PathChunk {
        std::vector<pf_bubble_path> paths;  // text_off relative to this chunk's text
        std::vector<uint32_t> count;        // paths per bubble
        std::string text;
        // colored: the oriented unitigs the walks of each branching bubble visit (findUnitig of the site strings)
        std::vector<uint32_t> walk_ovs, walk_first;
    };
    // The bubbles are processed in batches through a three-stage software pipeline: stage 1 (a host thread) builds the
    // path strings of batch b+2, stage 2 (another thread, the only one issuing device calls in this phase -- a pf_ctx is
    // not re-entrant) runs K-BUBBLE, the site strings and K-STRCOV of batch b+1, and the calling thread turns batch b
    // into text and appends it to the result files (stage 3).  Three sets of pinned exchange buffers go round.  Batches
    // are consumed in order, so var_count and the output order are those of one sequential pass.
    struct GroupRef { uint32_t first, count; };           // range of chunk-local string indices
    struct TaskSites {
        uint32_t group_first = 0;                          // index into the chunk's `groups`
    };
    struct SiteChunk {
        std::vector<std::string> strings;
        std::vector<GroupRef> groups;      // per (branching bubble, site, allele group), in order
        std::vector<uint32_t> first_group;  // per bubble of the chunk: index of its first GroupRef
        std::vector<uint64_t> mask;         // colored, per string: colours its findUnitig() mapping carries in full (CW words per string)
        std::vector<uint32_t> nowhere;      // colored: strings (chunk-local index, ascending) whose first k-mer is on no unitig of the bubble
        int err = 0;
    };
    struct Batch {
        size_t batch0 = 0, NT = 0;
        std::vector<SiteChunk> schunks;
        std::vector<uint64_t> chunk_base, str_sum, str_len;
        std::vector<uint8_t> str_ok, str_miss;
        uint64_t site_strings = 0;
        double sites_s = 0;
        AlignExchange *ax = nullptr;
        std::vector<PathChunk> pchunks;
        std::vector<uint32_t> dev_index;
        uint32_t n_dev = 0;
        double tasks_s = 0, align_s = 0;
        uint64_t text_len = 0, n_paths = 0;
        bool count_only = false;   // a batch ahead of this rank's slice: aligned for var_count only
        int st = PF_OK;
        std::string err;
    };
    static const pf_bubble_result kNoResult = {0, 0, 0, 0, 0, 0, 0, 0};
    // stage 1 of a batch (its own thread): path strings into the exchange buffers
    auto prepare = [&](Batch &B) -> int {
        auto t0 = clk::now();
        AlignExchange &X = *B.ax;
        const Task *tasks = all_tasks.data() + B.batch0;
        const size_t NT = B.NT;
        const size_t n_pch = n_chunks_of(NT, PCH);
        std::vector<PathChunk> &pchunks = B.pchunks;
        std::vector<uint32_t> &dev_index = B.dev_index;
        uint32_t &n_dev = B.n_dev;

        // ---- paths: oriented inner unitigs are decoded on the device; the s->t walks of the branching
        //      bubbles are enumerated here (two-stack DFS of src/CDBG.cpp:1364-1412) ---------------
        pchunks.assign(n_pch, PathChunk());
        parallel_chunks(NT, PCH, T, [&](size_t ci, size_t tb, size_t te) {
            PathChunk &pc = pchunks[ci];
            std::vector<uint32_t> major, minor;
            std::string walk;
            std::vector<std::string> strs;
            for (size_t ti = tb; ti < te; ++ti) {
                const Task &t = tasks[ti];
                if (colored) pc.walk_first.push_back((uint32_t)pc.walk_ovs.size());
                if (t.strict) {
                    for (int q = 0; q < t.n_inner; ++q) pc.paths.push_back({0, g_.size_bp(t.inner[q] >> 1), t.inner[q]});
                    pc.count.push_back(t.n_inner);
                    continue;
                }
                const uint32_t eu = t.exit_ov >> 1;
                const uint32_t ulen = g_.len_km(t.u);
                major.clear();
                minor.clear();
                walk.clear();
                strs.clear();
                minor.push_back(t.entrance_ov);
                while (!minor.empty()) {
                    const uint32_t w = minor.back();
                    minor.pop_back();
                    major.push_back(w);
                    if (colored && std::find(pc.walk_ovs.begin() + pc.walk_first.back(), pc.walk_ovs.end(), w) == pc.walk_ovs.end())
                        pc.walk_ovs.push_back(w);
                    const uint32_t wlen = g_.len_km(w >> 1);
                    const size_t before = walk.size();
                    g_.append_mapped(w, walk);
                    if ((w >> 1) == eu) {
                        const size_t total = walk.size();
                        strs.push_back(walk.substr(ulen - 1, total - ulen + 1 - wlen + 1));
                        walk.resize(before);
                        major.pop_back();
                        while (!major.empty() && !minor.empty()) {
                            const uint32_t *r = succ_row(major.back());
                            if (r[0] == minor.back() || r[1] == minor.back() || r[2] == minor.back() || r[3] == minor.back()) break;
                            walk.resize(walk.size() - g_.len_km(major.back() >> 1));
                            major.pop_back();
                        }
                    } else {
                        walk.resize(before + wlen);  // keep only the first len characters
                        const uint32_t *r = succ_row(w);
                        for (int b = 0; b < 4; ++b)
                            if (r[b] != NONE) minor.push_back(r[b]);
                    }
                }
                sort_paths(strs, 0, (int)strs.size() - 1);
                for (const std::string &sx : strs) {
                    pc.paths.push_back({(uint64_t)pc.text.size(), (uint32_t)sx.size(), NONE});
                    pc.text += sx;
                }
                pc.count.push_back((uint32_t)strs.size());
            }
            if (colored) pc.walk_first.push_back((uint32_t)pc.walk_ovs.size());
        });
        // one flat batch: bubbles with fewer than two paths (the reference indexes str[1] blindly) are skipped
        std::vector<uint64_t> text_base(n_pch + 1, 0), path_base(n_pch + 1, 0);
        for (size_t c = 0; c < n_pch; ++c) {
            text_base[c + 1] = text_base[c] + pchunks[c].text.size();
            path_base[c + 1] = path_base[c] + pchunks[c].paths.size();
        }
        X.text.ensure(ctx_, text_base[n_pch] + 1);
        X.paths.ensure(ctx_, path_base[n_pch] + 1);
        X.tasks.ensure(ctx_, NT);
        dev_index.assign(NT, NONE);  // bubble -> index in the device batch
        n_dev = 0;
        {
            size_t ti = 0;
            for (size_t c = 0; c < n_pch; ++c) {
                uint64_t pf = path_base[c];
                for (uint32_t cnt : pchunks[c].count) {
                    if (cnt >= 2) {
                        X.tasks.p[n_dev] = pf_bubble_task{pf, cnt, 0};
                        dev_index[ti] = n_dev++;
                    }
                    pf += cnt;
                    ++ti;
                }
            }
        }
        parallel_chunks(n_pch, 1, T, [&](size_t c, size_t, size_t) {
            PathChunk &pc = pchunks[c];
            if (!pc.text.empty()) memcpy(X.text.p + text_base[c], pc.text.data(), pc.text.size());
            pf_bubble_path *dst = X.paths.p + path_base[c];
            for (size_t i = 0; i < pc.paths.size(); ++i) {
                dst[i] = pc.paths[i];
                if (dst[i].ov == NONE) dst[i].text_off += text_base[c];
            }
            PathChunk().text.swap(pc.text);
            std::vector<pf_bubble_path>().swap(pc.paths);
        });
        B.tasks_s = since(t0);
        B.text_len = text_base[n_pch];
        B.n_paths = path_base[n_pch];
        return PF_OK;
    };
    // stage 2 of a batch (its own thread, the only one that talks to the device in this phase): SeqAlign on the
    // device, site strings and their coverage
    auto produce = [&](Batch &B) -> int {
        AlignExchange &X = *B.ax;
        const Task *tasks = all_tasks.data() + B.batch0;
        const size_t NT = B.NT;
        const size_t n_pch = n_chunks_of(NT, PCH);
        std::vector<PathChunk> &pchunks = B.pchunks;
        std::vector<uint32_t> &dev_index = B.dev_index;
        const uint32_t n_dev = B.n_dev;
        int st = PF_OK;
        // ---- align: SeqAlign::SequenceAlignment of every bubble, one wavefront each ------------------
        auto t0 = clk::now();
        X.res.ensure(ctx_, std::max<uint32_t>(n_dev, 1));
        uint64_t cap_text = std::max<uint64_t>(X.otext.cap, (B.text_len + 128ull * n_dev) * 2 + 4096);
        uint64_t cap_sites = std::max<uint64_t>(X.osites.cap, 4ull * n_dev + 64);
        uint64_t cap_groups = std::max<uint64_t>(X.ogroups.cap, 16ull * n_dev + 64);
        uint64_t cap_ilen = std::max<uint64_t>(X.oilen.cap, 2ull * n_dev + 64);
        for (;;) {
            X.otext.ensure(ctx_, cap_text);
            X.osites.ensure(ctx_, cap_sites);
            X.ogroups.ensure(ctx_, cap_groups);
            X.oilen.ensure(ctx_, cap_ilen);
            uint64_t used[4];
            st = pf_align_bubbles(ctx_, X.text.p, B.text_len, X.paths.p, B.n_paths, X.tasks.p, n_dev, sc_.match,
                                  sc_.mismatch, sc_.gap, X.res.p, X.otext.p, X.otext.cap, X.osites.p, X.osites.cap,
                                  X.ogroups.p, X.ogroups.cap, X.oilen.p, X.oilen.cap, used);
            if (st == PF_ERR_OVERFLOW && (used[0] > X.otext.cap || used[1] > X.osites.cap || used[2] > X.ogroups.cap ||
                                          used[3] > X.oilen.cap)) {
                cap_text = std::max<uint64_t>(X.otext.cap, used[0] + used[0] / 8);
                cap_sites = std::max<uint64_t>(X.osites.cap, used[1] + used[1] / 8);
                cap_groups = std::max<uint64_t>(X.ogroups.cap, used[2] + used[2] / 8);
                cap_ilen = std::max<uint64_t>(X.oilen.cap, used[3] + used[3] / 8);
                continue;
            }
            break;
        }
        if (st != PF_OK) { B.err = std::string(tag_) + "::PloidyEstimation(): alignment: " + pf_last_error(ctx_); return st; }
        B.align_s = since(t0);
        if (B.count_only) return PF_OK;  // ahead of this rank's slice: only whether each alignment has rows matters
        auto result_of = [&](size_t ti) -> const pf_bubble_result & {
            return dev_index[ti] == NONE ? kNoResult : X.res.p[dev_index[ti]];
        };

        // ---- sites: strings of the branching bubbles (src/CDBG.cpp:1448-1600) -> one C2 batch ----
        t0 = clk::now();
        std::vector<SiteChunk> &schunks = B.schunks;
        schunks.assign(n_pch, SiteChunk());
        // cdbg.findUnitig(s, 0, len) of src/CCDBG.cpp:3251, 3390 followed by UnitigColors::contains on that mapping: the
        // first k-mer of a site string lies on one of the bubble's unitigs (each k-mer occurs once in the graph); the
        // mapping is extended along that unitig while the characters agree (CompactedDBG.tcc:3815-3837,
        // CompressedSequence.cpp:497-520).  Returns the mask of colours present on every k-mer of the mapping.
        const uint32_t CW = (C + 63) / 64;   // 64-bit words of a colour set
        auto colours_of_string = [&](const std::string &sx, const uint32_t *ovs, size_t n_ovs, int &err, std::vector<uint64_t> &into, std::vector<uint32_t> &nowhere) {
            const size_t at = into.size();
            into.resize(at + CW, 0);
            std::string rc((size_t)k, 'A');
            for (int i = 0; i < k; ++i) {
                const char ch = sx[(size_t)k - 1 - i];
                rc[i] = ch == 'A' ? 'T' : ch == 'C' ? 'G' : ch == 'G' ? 'C' : 'A';
            }
            const std::string_view head(sx.data(), (size_t)k), rhead(rc);
            for (size_t q = 0; q < n_ovs; ++q) {
                const uint32_t u = ovs[q] >> 1;
                const std::string_view seq = g_.seq(u);
                uint32_t dist, len;
                size_t p = seq.find(head);
                if (p != std::string_view::npos) {
                    size_t j = 0;
                    while (j < sx.size() && p + j < seq.size() && sx[j] == seq[p + j]) ++j;
                    len = (uint32_t)(j - (size_t)k + 1);
                    dist = (uint32_t)p;
                } else if ((p = seq.find(rhead)) != std::string_view::npos) {
                    long pos = (long)p + k - 1;
                    size_t j = 0;
                    auto comp = [](char ch) { return ch == 'A' ? 'T' : ch == 'C' ? 'G' : ch == 'G' ? 'C' : 'A'; };
                    while (j < sx.size() && pos >= 0 && sx[j] == comp(seq[(size_t)pos])) { ++j; --pos; }
                    len = (uint32_t)(j - (size_t)k + 1);
                    dist = (uint32_t)p - (len - 1);
                } else {
                    continue;
                }
                for (uint32_t c = 0; c < C; ++c)
                    if (col_->contains(u, c, dist, len)) into[at + (c >> 6)] |= 1ull << (c & 63);
                return;
            }
            // findUnitig finds nothing: the reference dereferences the empty mapping -- IF it gets to this string.  The walk over a
            // site's strings ends at the first one out of range (src/CCDBG.cpp:3262-3276), so the verdict is the walker's (consume).
            (void)err;
            nowhere.push_back((uint32_t)(at / CW));
        };
        parallel_chunks(NT, PCH, T, [&](size_t ci, size_t tb, size_t te) {
            SiteChunk &sc = schunks[ci];
            sc.first_group.reserve(te - tb);
            std::vector<std::string> kstr;
            std::vector<int> at;
            std::vector<std::set<std::string>> groups;
            for (size_t ti = tb; ti < te; ++ti) {
                sc.first_group.push_back((uint32_t)sc.groups.size());
                const pf_bubble_result &r = result_of(ti);
                if (r.n_rows == 0 || tasks[ti].strict) continue;
                const size_t R = r.n_rows, L = r.n_cols;
                const char *rows = X.otext.p + r.rows_off;
                auto row_at = [&](size_t p, size_t x) -> char {  // std::string::at of the reference: out of range is fatal
                    if (x >= L) { sc.err = 1; return 'A'; }
                    return rows[p * L + x];
                };
                auto sub = [&](size_t p, long from, long n) -> std::string {  // substr(from, n): a negative count is npos
                    if (from < 0 || (size_t)from > L) { sc.err = 1; return std::string(); }
                    return std::string(rows + p * L + from, n < 0 ? L - (size_t)from : std::min<size_t>((size_t)n, L - (size_t)from));
                };
                // substr(x, 1): one character, NOTHING at x == size() (a row that ends in gaps), fatal beyond
                auto sub1 = [&](size_t p, size_t x, bool &none) -> char {
                    none = x >= L;
                    if (x > L) sc.err = 1;
                    return none ? '\0' : rows[p * L + x];
                };
                auto ungapped_prefix = [&](size_t p, size_t n) {
                    std::string o;
                    for (size_t x = 0; x < n && x < L; ++x)
                        if (rows[p * L + x] != '-') o.push_back(rows[p * L + x]);
                    return o;
                };
                kstr.assign(R, std::string());
                at.assign(R, 0);
                uint32_t indel = 0;
                for (uint32_t si = 0; si < r.n_sites && !sc.err; ++si) {
                    const pf_bubble_site &sr = X.osites.p[r.site_off + si];
                    const uint8_t *grp = X.ogroups.p + r.group_off + (uint64_t)si * R;
                    const uint32_t site = sr.col;
                    for (auto &x : kstr) x.clear();
                    if (sr.is_indel) {
                        std::fill(at.begin(), at.end(), (int)site);
                        for (;;) {
                            char first = 0;
                            bool differ = false;
                            for (size_t p = 0; p < R; ++p) {
                                bool none;
                                char ch = sub1(p, (size_t)at[p], none);
                                while (!none && ch == '-') ch = sub1(p, (size_t)++at[p], none);
                                at[p]++;
                                if (!none) kstr[p].push_back(ch);   // ('\0' joins the reference's set of characters all the same)
                                if (p == 0) first = ch;
                                else if (ch != first) differ = true;
                            }
                            if (differ || sc.err) break;
                        }
                        for (size_t p = 0; p < R; ++p) {
                            const int n = (int)kstr[p].size();
                            if (indel == 0) {
                                kstr[p] = sub(p, (long)site - k + n, k - n) + kstr[p];
                            } else {
                                std::string tmp = ungapped_prefix(p, site);
                                if (tmp.size() < (size_t)(k - n)) {
                                    kstr[p] = tmp + kstr[p];
                                    for (int x = at[p]; kstr[p].size() < (size_t)k && !sc.err; ++x) {
                                        const char ch = row_at(p, (size_t)x);
                                        if (ch != '-') kstr[p].push_back(ch);
                                    }
                                } else {
                                    kstr[p] = tmp.substr(tmp.size() - (size_t)k + n, (size_t)(k - n)) + kstr[p];
                                }
                            }
                        }
                        ++indel;
                    } else if (indel > 0) {
                        for (size_t p = 0; p < R; ++p) {
                            std::string tmp = ungapped_prefix(p, site + 1);
                            if (tmp.size() < (size_t)k) {
                                kstr[p] = tmp;
                                for (int x = (int)site + 1; kstr[p].size() < (size_t)k && !sc.err; ++x) {
                                    const char ch = row_at(p, (size_t)x);
                                    if (ch != '-') kstr[p].push_back(ch);
                                }
                            } else {
                                kstr[p] = tmp.substr(tmp.size() - (size_t)k, (size_t)k);
                            }
                        }
                    } else {
                        for (size_t p = 0; p < R; ++p) kstr[p] = sub(p, (long)site - k + 1, k);
                    }
                    // distinct strings per allele group, in std::set order
                    groups.assign(sr.maxnum, std::set<std::string>());
                    for (size_t p = 0; p < R; ++p) groups[grp[p] - 1].insert(kstr[p]);
                    for (auto &gs : groups) {
                        sc.groups.push_back({(uint32_t)sc.strings.size(), (uint32_t)gs.size()});
                        for (auto &sx : gs) {
                            sc.strings.push_back(sx);
                            if (colored) {
                                const PathChunk &pc = pchunks[ci];
                                const uint32_t w0 = pc.walk_first[ti - tb], w1 = pc.walk_first[ti - tb + 1];
                                colours_of_string(sx, pc.walk_ovs.data() + w0, w1 - w0, sc.err, sc.mask, sc.nowhere);
                            }
                        }
                    }
                }
            }
        });
        for (auto &scx : schunks) {
            if (scx.err == 2) { B.err = "CCDBG::PloidyEstimation(): a site string does not start on a unitig of its bubble"; return PF_ERR_ARG; }
            if (scx.err) { B.err = "CDBG::PloidyEstimation(): a site string runs past the end of an aligned row (the reference terminates here: std::out_of_range from substr, src/CDBG.cpp:1478-1590)"; return PF_ERR_ARG; }
        }
        std::vector<uint64_t> &chunk_base = B.chunk_base;
        chunk_base.assign(n_pch + 1, 0);
        for (size_t c = 0; c < n_pch; ++c) chunk_base[c + 1] = chunk_base[c] + schunks[c].strings.size();
        const size_t n_strings = chunk_base[n_pch];
        std::vector<uint64_t> &str_sum = B.str_sum, &str_len = B.str_len;  // colored: [string][colour]
        std::vector<uint8_t> &str_ok = B.str_ok;
        str_sum.assign(n_strings * C, 0);
        str_len.assign(n_strings, 0);
        str_ok.assign(n_strings * C, 0);
        std::vector<uint8_t> str_miss(colored ? 0 : n_strings);
        if (n_strings) {
            std::string text;
            std::vector<uint64_t> soff(n_strings + 1, 0);
            size_t q = 0;
            for (auto &scx : schunks)
                for (auto &sx : scx.strings) { soff[q] = text.size(); str_len[q] = sx.size(); text += sx; ++q; }
            soff[n_strings] = text.size();
            if (colored)  // a missing k-mer is (0, false) on this path, not an exit (src/CCDBG.cpp:113-117)
                st = pf_string_cov_colored(ctx_, text.data(), soff.data(), (uint32_t)n_strings, lows.data(), ups.data(), str_sum.data(),
                                           str_ok.data());
            else
                st = pf_string_cov(ctx_, text.data(), soff.data(), (uint32_t)n_strings, low, up, str_sum.data(), str_ok.data(),
                                   str_miss.data());
            if (st != PF_OK) { B.err = std::string(tag_) + "::PloidyEstimation(): " + pf_last_error(ctx_); return st; }
            // (a k-mer that is in no database ends the reference's run inside readCov -- if the walk over the site's strings gets to
            // that string; it ends at the first string out of range, src/CDBG.cpp:1527-1551: judged in order by consume)
            B.str_miss.swap(str_miss);
            B.site_strings = n_strings;
        }
        B.sites_s = since(t0);

        return PF_OK;
    };

    // stage 3 of a batch (the calling thread): formatting and appending to the result files
    auto consume = [&](Batch &B) -> int {
        AlignExchange &X = *B.ax;
        const Task *tasks = all_tasks.data() + B.batch0;
        const size_t NT = B.NT;
        const size_t n_pch = n_chunks_of(NT, PCH);
        const std::vector<uint32_t> &dev_index = B.dev_index;
        const std::vector<SiteChunk> &schunks = B.schunks;
        const std::vector<uint64_t> &chunk_base = B.chunk_base, &str_sum = B.str_sum, &str_len = B.str_len;
        const std::vector<uint8_t> &str_ok = B.str_ok, &str_miss = B.str_miss;
        std::atomic<int> fatal{0};   // 1: a k-mer in no database, 2: a site string on no unitig -- reached by the reference's walk
        auto result_of = [&](size_t ti) -> const pf_bubble_result & {
            return dev_index[ti] == NONE ? kNoResult : X.res.p[dev_index[ti]];
        };
        // ---- format ------------------------------------------------------------------------------
        auto t0 = clk::now();
        std::vector<uint64_t> vc(NT);  // var_count of each bubble (1-based over non-empty alignments)
        for (size_t ti = 0; ti < NT; ++ti) {
            if (result_of(ti).n_rows) ++var_count;
            vc[ti] = var_count;
        }
        if (B.count_only) return PF_OK;
        all_outs.emplace_back(n_pch);
        std::vector<ChunkOut> &outs = all_outs.back();
        parallel_chunks(NT, PCH, T, [&](size_t ci, size_t tb, size_t te) {
            ChunkOut &o = outs[ci];
            const SiteChunk &sc = schunks[ci];
            std::string cov_info, fre_info, tail;
            double tc[256];
            std::vector<double> gc;  // colored: [colour][allele group] coverage of the site
            const uint32_t CW = (C + 63) / 64;
            std::vector<uint64_t> seen_colours(CW);
            for (size_t ti = tb; ti < te; ++ti) {
                const Task &t = tasks[ti];
                const pf_bubble_result &r = result_of(ti);
                if (r.n_rows == 0) continue;
                const size_t R = r.n_rows, L = r.n_cols;
                const char *rows = X.otext.p + r.rows_off;
                const uint64_t my_vc = vc[ti];
                for (size_t p = 0; p < R; ++p) {
                    put_uint(o.s_var, my_vc);
                    o.s_var += t.strict ? "\t1\t" : "\t0\t";
                    put_uint(o.s_var, t.u + 1);
                    o.s_var.push_back('\t');
                    put_uint(o.s_var, (t.exit_ov >> 1) + 1);
                    o.s_var.push_back('\t');
                    o.s_var.append(rows + p * L, L);
                    o.s_var.push_back('\n');
                }
                o.core_cov += (uint64_t)t.core_mean;
                o.core_num++;
                const pf_bubble_site *sites = X.osites.p + r.site_off;
                const uint32_t *ilen = X.oilen.p + r.ilen_off;
                const size_t usize = g_.size_bp(t.u), esize = g_.size_bp(t.exit_ov >> 1);
                const uint32_t ns = r.n_sites;
                uint32_t indel = 0;
                uint32_t gcur = sc.first_group[ti - tb];  // walks this bubble's GroupRefs (branching only)
                for (uint32_t i = 0; i < ns; ++i) {
                    const pf_bubble_site &sr = sites[i];
                    const uint8_t *grp = X.ogroups.p + r.group_off + (uint64_t)i * R;
                    // distance to the neighbouring sites / unitig ends (src/CDBG.cpp:1279-1298)
                    uint32_t vd;
                    if (i == 0) {
                        if (ns != 1) vd = (uint32_t)std::min((size_t)(uint32_t)(sites[1].col - sites[0].col - 1), usize);
                        else vd = (uint32_t)std::min(usize, esize);
                    } else if (i == ns - 1) {
                        vd = (uint32_t)std::min((size_t)(uint32_t)(sites[i].col - sites[i - 1].col - 1), esize);
                    } else {
                        vd = std::min((uint32_t)(sites[i].col - sites[i - 1].col - 1), (uint32_t)(sites[i + 1].col - sites[i].col - 1));
                    }
                    const uint32_t maxnum = sr.maxnum;
                    for (uint32_t x = 0; x < maxnum; ++x) tc[x] = 0.0;
                    double denom;
                    if (sr.is_indel) ++indel;  // counted even when the site is dropped below (src/CDBG.cpp:1526)
                    if (colored) {
                        // src/CCDBG.cpp:2971-3059 (strict), :3236-3339 and :3374-3475 (branching): one row per colour that
                        // sees at least two allele groups, each with the colour id and the largest Cramer's V over colour pairs
                        gc.assign((size_t)C * maxnum, 0.0);
                        double coefficient;
                        if (t.strict) {
                            const double *m = cov_pools[t.cov_ref >> 32].data() + (uint32_t)t.cov_ref;  // [colour][4], sorted paths
                            for (uint32_t c = 0; c < C; ++c)
                                for (size_t p = 0; p < R; ++p) gc[(size_t)c * maxnum + grp[p] - 1] += m[(size_t)c * 4 + p];
                            coefficient = max_cramer_v(m, C, 4, R);
                        } else {
                            const uint64_t base = chunk_base[ci];
                            std::fill(seen_colours.begin(), seen_colours.end(), 0);
                            bool ok = true;
                            for (uint32_t gi = 0; gi < maxnum && ok; ++gi) {
                                const GroupRef &gr = sc.groups[gcur + gi];
                                for (uint32_t qi = gr.first; qi < gr.first + gr.count && ok; ++qi) {
                                    const uint64_t q = base + qi;
                                    if (std::binary_search(sc.nowhere.begin(), sc.nowhere.end(), qi)) { fatal = 2; ok = false; break; }
                                    const uint64_t *mask = &sc.mask[(size_t)qi * CW];
                                    for (uint32_t c = 0; c < C; ++c) {
                                        if (!((mask[c >> 6] >> (c & 63)) & 1)) continue;
                                        seen_colours[c >> 6] |= 1ull << (c & 63);
                                        if (!str_ok[q * C + c]) { ok = false; break; }
                                        gc[(size_t)c * maxnum + gi] += (double)str_sum[q * C + c] / (double)(str_len[q] - (size_t)k + 1);
                                    }
                                }
                            }
                            gcur += maxnum;
                            bool every_colour = true;   // (colour_set.size() == C, src/CCDBG.cpp:3292, 3427)
                            for (uint32_t x = 0; x < CW; ++x) {
                                const uint32_t left = C - 64 * x;
                                every_colour = every_colour && seen_colours[x] == (left >= 64 ? ~0ull : ((1ull << left) - 1));
                            }
                            if (!every_colour || !ok) continue;
                            coefficient = max_cramer_v(gc.data(), C, maxnum, maxnum);
                        }
                        tail.clear();
                        tail += t.strict ? "1\t" : "0\t";
                        if (sr.is_indel) put_uint(tail, indel - 1 < r.n_indel_len ? ilen[indel - 1] : r.n_cols - sr.col);  // (open run: pf_call.hip open_run_len)
                        else tail.push_back('0');
                        tail.push_back('\t');
                        put_uint(tail, my_vc);
                        tail.push_back('\t');
                        put_uint(tail, ns);
                        tail.push_back('\t');
                        put_double(tail, coefficient);
                        tail.push_back('\t');
                        put_uint(tail, vd);
                        tail += "\t\n";
                        for (uint32_t c = 0; c < C; ++c) {
                            const double *row = gc.data() + (size_t)c * maxnum;
                            uint32_t n_res = 0;
                            double sum = 0;
                            for (uint32_t x = 0; x < maxnum; ++x)
                                if (row[x] > 0.0) { ++n_res; sum += row[x]; }
                            if (n_res < 2) continue;
                            cov_info.clear();
                            fre_info.clear();
                            for (uint32_t x = 0; x < maxnum; ++x) {
                                if (!(row[x] > 0.0)) continue;
                                put_double(cov_info, row[x]);
                                cov_info.push_back('\t');
                                put_double(fre_info, row[x] / sum);
                                fre_info.push_back('\n');
                            }
                            put_uint(cov_info, c);
                            cov_info.push_back('\t');
                            cov_info += tail;
                            o.allfre += fre_info;
                            if (n_res >= 2 && n_res <= 5) {
                                ++o.allele[n_res - 2];
                                o.fre[n_res - 2] += fre_info;
                                o.cov[n_res - 2] += cov_info;
                            }
                        }
                        continue;
                    }
                    if (t.strict) {
                        for (size_t p = 0; p < R; ++p) tc[grp[p] - 1] += t.cov[p];
                        denom = t.cov_sum;
                    } else {
                        const uint64_t base = chunk_base[ci];
                        bool ok = true;
                        double sum = 0;
                        for (uint32_t gi = 0; gi < maxnum; ++gi) {
                            const GroupRef &gr = sc.groups[gcur + gi];
                            if (ok) {
                                for (uint64_t q = base + gr.first; q < base + gr.first + gr.count; ++q) {
                                    if (!str_miss.empty() && str_miss[q]) { fatal = 1; ok = false; break; }
                                    if (!str_ok[q]) { ok = false; break; }
                                    tc[gi] += (double)str_sum[q] / (double)(str_len[q] - (size_t)k + 1);
                                }
                                if (ok) sum += tc[gi];
                            }
                        }
                        gcur += maxnum;
                        if (!ok) continue;
                        denom = sum;
                    }
                    cov_info.clear();
                    fre_info.clear();
                    for (uint32_t x = 0; x < maxnum; ++x) {
                        put_double(cov_info, tc[x]);
                        cov_info.push_back('\t');
                        put_double(fre_info, tc[x] / denom);
                        fre_info.push_back('\n');
                    }
                    cov_info += t.strict ? "1\t" : "0\t";
                    if (sr.is_indel) put_uint(cov_info, indel - 1 < r.n_indel_len ? ilen[indel - 1] : r.n_cols - sr.col);  // (open run: pf_call.hip open_run_len)
                    else cov_info.push_back('0');
                    cov_info.push_back('\t');
                    put_uint(cov_info, my_vc);
                    cov_info.push_back('\t');
                    put_uint(cov_info, ns);
                    cov_info.push_back('\t');
                    put_uint(cov_info, vd);
                    cov_info += "\t\n";
                    o.allfre += fre_info;
                    if (maxnum >= 2 && maxnum <= 5) {
                        ++o.allele[maxnum - 2];
                        o.fre[maxnum - 2] += fre_info;
                        o.cov[maxnum - 2] += cov_info;
                    }
                }
            }
        });
        if (fatal == 1) { B.err = "CDBG::readCov(): a kmer of a site string can not found ."; return PF_ERR_MISSING_KMER; }
        if (fatal == 2) { B.err = "CCDBG::PloidyEstimation(): a site string does not start on a unitig of its bubble"; return PF_ERR_ARG; }
        for (ChunkOut &o : outs) {
            for (int a = 0; a < 4; ++a) allele_[a] += o.allele[a];
            core_cov_ += o.core_cov;
            core_num_ += o.core_num;
        }
        times_.format_s += since(t0);
        return PF_OK;
    };

    if (opener.joinable()) opener.join();
    if (open_failed >= 0) { close_files(); return fail(PF_ERR_ARG, "CDBG:: Open " + files[(size_t)open_failed].name + " file error"); }
    last_allfre_.clear();
    last_allfre_file_.clear();
    tp("files open");
    double write_s = 0;
    auto write_batch = [&](std::vector<ChunkOut> &outs) {
        const auto tw = clk::now();
        parallel_chunks(files.size(), 1, T, [&](size_t fi, size_t, size_t) {
            OutFile &of = files[fi];
            for (ChunkOut &o : outs) {
                const std::string &piece = fi == 0 ? o.allfre : fi == 1 ? o.s_var : fi < 6 ? o.fre[fi - 2] : o.cov[fi - 6];
                of.bytes += piece.size();
                if (of.f && !piece.empty() && fwrite(piece.data(), 1, piece.size(), of.f) != piece.size()) of.rc = 1;
                if (fi == 0) last_allfre_ += piece;
            }
        });
        std::vector<ChunkOut>().swap(outs);
        write_s += since(tw);
    };
    // stage 4: a writer thread appends the finished batches, in order, while the next ones are formatted
    std::mutex wmu;
    std::condition_variable wcv;
    std::deque<std::vector<ChunkOut> *> wqueue;
    bool wdone = false;
    std::thread writer([&] {
        for (;;) {
            std::vector<ChunkOut> *outs = nullptr;
            {
                std::unique_lock<std::mutex> lk(wmu);
                wcv.wait(lk, [&] { return wdone || !wqueue.empty(); });
                if (wqueue.empty()) return;
                outs = wqueue.front();
                wqueue.pop_front();
            }
            write_batch(*outs);
            tp("  batch written");
        }
    });
    struct WriterGuard {  // joined on every way out
        std::thread &t;
        std::mutex &mu;
        std::condition_variable &cv;
        bool &done;
        ~WriterGuard() {
            if (!t.joinable()) return;
            { std::lock_guard<std::mutex> lk(mu); done = true; }
            cv.notify_all();
            t.join();
        }
    } writer_guard{writer, wmu, wcv, wdone};

    {
        constexpr size_t kRing = 3;  // sets of exchange buffers = batches in flight
        std::vector<std::pair<size_t, size_t>> cuts;  // (first bubble, bubbles); the slice boundary is a batch boundary
        for (size_t at = 0; at < skip_before; at += CHUNK) cuts.emplace_back(at, std::min(CHUNK, skip_before - at));
        const size_t n_counted = cuts.size();
        for (size_t at = skip_before; at < all_tasks.size(); at += CHUNK) cuts.emplace_back(at, std::min(CHUNK, all_tasks.size() - at));
        const size_t n_batches = cuts.size();
        all_outs.reserve(n_batches);  // the writer holds pointers to its elements
        std::vector<Batch> batches(n_batches);
        for (size_t b = 0; b < n_batches; ++b) {
            batches[b].batch0 = cuts[b].first;
            batches[b].NT = cuts[b].second;
            batches[b].count_only = b < n_counted;
            batches[b].ax = &ax_[b % kRing];
        }
        std::mutex mu;
        std::condition_variable cv;
        size_t prepared = 0, produced = 0, consumed = 0;  // batches finished by stage 1 / 2 / 3
        bool stop = false;
        std::thread stage1, stage2;
        if (n_batches > 1) {
            stage1 = std::thread([&] {
                for (size_t b = 0; b < n_batches; ++b) {
                    {  // the exchange buffers of batch b are those of batch b - 3
                        std::unique_lock<std::mutex> lk(mu);
                        cv.wait(lk, [&] { return stop || b < consumed + kRing; });
                        if (stop) return;
                    }
                    batches[b].st = prepare(batches[b]);
                    tp("  batch prepared (paths)");
                    {
                        std::lock_guard<std::mutex> lk(mu);
                        prepared = b + 1;
                        if (batches[b].st != PF_OK) stop = true;
                    }
                    cv.notify_all();
                    if (batches[b].st != PF_OK) return;
                }
            });
            stage2 = std::thread([&] {
                for (size_t b = 0; b < n_batches; ++b) {
                    {
                        std::unique_lock<std::mutex> lk(mu);
                        cv.wait(lk, [&] { return stop || prepared > b; });
                        if (prepared <= b) return;  // stopped before this batch was prepared
                        if (batches[b].st != PF_OK) { produced = b + 1; cv.notify_all(); return; }
                    }
                    const int st2 = produce(batches[b]);
                    tp("  batch produced (device)");
                    {
                        std::lock_guard<std::mutex> lk(mu);
                        batches[b].st = st2;
                        produced = b + 1;
                        if (st2 != PF_OK) stop = true;
                    }
                    cv.notify_all();
                    if (st2 != PF_OK) return;
                }
            });
        }
        int rc = PF_OK;
        std::string rc_err;
        for (size_t b = 0; b < n_batches && rc == PF_OK; ++b) {
            if (n_batches > 1) {
                std::unique_lock<std::mutex> lk(mu);
                cv.wait(lk, [&] { return produced > b; });
            } else {
                batches[b].st = prepare(batches[b]);
                if (batches[b].st == PF_OK) batches[b].st = produce(batches[b]);
            }
            Batch &B = batches[b];
            if (!B.count_only) {  // counters are this rank's share: they add up over the ranks of a partitioned run
                times_.tasks += B.NT;
                times_.align_jobs += B.n_dev;
                times_.site_strings += B.site_strings;
            }
            times_.tasks_s += B.tasks_s;
            times_.align_s += B.align_s;
            times_.sites_s += B.sites_s;
            if (B.st != PF_OK) { rc = B.st; rc_err = B.err; break; }
            rc = consume(B);
            if (rc != PF_OK && rc_err.empty()) rc_err = B.err;
            tp("  batch consumed (format)");
            if (rc == PF_OK && !B.count_only) {
                { std::lock_guard<std::mutex> lk(wmu); wqueue.push_back(&all_outs.back()); }
                wcv.notify_all();
            }
            std::vector<PathChunk>().swap(B.pchunks);
            std::vector<uint32_t>().swap(B.dev_index);
            std::vector<SiteChunk>().swap(B.schunks);
            std::vector<uint64_t>().swap(B.str_sum);
            {
                std::lock_guard<std::mutex> lk(mu);
                consumed = b + 1;
                if (rc != PF_OK) stop = true;
            }
            cv.notify_all();
        }
        if (stage1.joinable()) {
            { std::lock_guard<std::mutex> lk(mu); stop = true; }
            cv.notify_all();
            stage1.join();
            stage2.join();
        }
        if (rc != PF_OK) {
            { std::lock_guard<std::mutex> lk(wmu); wdone = true; }
            wcv.notify_all();
            writer.join();
            close_files();
            return rc_err.empty() ? status_ : fail(rc, rc_err);
        }
    }


    {
        { std::lock_guard<std::mutex> lk(wmu); wdone = true; }
        wcv.notify_all();
        writer.join();
    }
    tp("pipeline done");
    t0 = clk::now();
    if (join_pending_write() || join_pending_ids()) { close_files(); return status_; }
    close_files();
    for (OutFile &of : files) {
        out_bytes_ += of.bytes;
        if (of.rc) return fail(PF_ERR_ARG, "CDBG:: write error on " + of.name);
    }
    write_s += since(t0);
    times_.write_s = write_s;
    tp("files closed");
    times_.ploidy_total_s = since(t_all);
    if (!quiet_) {
        printf("%s::PloidyEstimation():  Cpu time : %gs\n", tag_, (double)(clock() - c0) / CLOCKS_PER_SEC);
        printf("%s::PloidyEstimation():  Real time : %gs\n", tag_, times_.ploidy_total_s);
        printf("%s::PloidyEstimation(): Alleles in SuperBubbles  :\t2 :%llu\t3 :%llu\t4 :%llu\t5 :%llu\n", tag_,
               (unsigned long long)allele_[0], (unsigned long long)allele_[1], (unsigned long long)allele_[2],
               (unsigned long long)allele_[3]);
        // the reference divides unguarded (src/CDBG.cpp:1703) and dies with SIGFPE when no site exists
        if (core_num_) printf("%s::PloidyEstimation(): Sites' Average Coverage:%d\n", tag_, (int)(core_cov_ / core_num_));
    }
    return 0;
}

}  // namespace pfh
