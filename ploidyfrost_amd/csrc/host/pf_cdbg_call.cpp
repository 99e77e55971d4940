// pfh::CDBG::ploidy_estimation_resident: CDBG::ploidyEstimation_ptr (reference src/CDBG.cpp:1101-1705) with the device's
// resident calling pipeline (include/ploidyfrost_hip.h, pf_call_*).  What stays on the host is what is sequential by
// construction or belongs to the operating system:
//   * the light pass of the driver loop (:1146-1186, 1656-1679) -- a side is handled only if its bit is still set when its
//     unitig comes up, and handling a bubble closes both endpoint sides -- over 16-byte side records the device scanned;
//   * appending the text slabs the device formatted to the ten result files.
// Bubbles are processed in batches; while the device works on batch b the slabs of batch b-1 are copied back and written.
#include <fcntl.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <cstdio>
#include <cstring>
#include <ctime>
#include <deque>
#include <memory>
#include <mutex>
#include <thread>

#include "../pf_alnpack.hpp"
#include "../pf_nibble.hpp"
#include "pf_cdbg_impl.hpp"
#include "pf_parallel.hpp"

namespace pfh {

namespace {
const char *kStreamSuffix[PF_CALL_STREAMS] = {"_allele_frequency.txt", "_alignseq.txt", "_bifre.txt",    "_trifre.txt",   "_tetrafre.txt",
                                              "_pentafre.txt",         "_bicov.txt",    "_tricov.txt",   "_tetracov.txt", "_pentacov.txt"};
}

// ---- first half: everything up to the list of bubbles to call (device scan + the sequential pass) ----
int CDBG::call_select(const std::vector<std::pair<int, int>> &cutoff, uint64_t &n_tasks) {
    const uint32_t low = (uint32_t)cutoff[0].first, up = (uint32_t)cutoff[0].second;
    auto t0 = clk::now();
    int st = PF_OK;
    if (col_) {   // colored: one (lower, upper) per colour
        std::vector<uint32_t> lows(cutoff.size()), ups(cutoff.size());
        for (size_t c = 0; c < cutoff.size(); ++c) { lows[c] = (uint32_t)cutoff[c].first; ups[c] = (uint32_t)cutoff[c].second; }
        st = pf_call_set_cutoffs(ctx_, (uint32_t)cutoff.size(), lows.data(), ups.data());
        if (st != PF_OK) return fail(st, std::string(tag_) + "::PloidyEstimation(): " + pf_last_error(ctx_));
    }
    if (!cov_ready_) st = launch_coverage();
    cov_ready_ = false;
    if (st != PF_OK) return fail(st, std::string(tag_) + "::PloidyEstimation(): " + cov_err_);
    times_.cov_device_s = since(t0);
    t0 = clk::now();
    if (!state_on_device_) {
        if (sync_state_to_host()) return status_;   // (committed on the device and already consumed once: back through the host copy)
        st = pf_call_set_state(ctx_, flags_.data(), plus_.data(), minus_.data());
    }
    state_on_device_ = false;   // (one PloidyEstimation per findSuperBubble, as in the reference's main(); a second one uploads again)
    uint64_t n_sides = 0;
    if (st == PF_OK) st = pf_call_scan(ctx_, low, up, &n_sides);
    if (st != PF_OK) return fail(st, std::string(tag_) + "::PloidyEstimation(): scan: " + pf_last_error(ctx_));
    uint64_t nk = 0;
    static const bool host_pass = [] { const char *e = getenv("PF_SCAN"); return e && !strcmp(e, "host"); }();
    if (!host_pass) {
        // part B on the device (pf_call_resolve): no side record leaves the GPU
        const auto t_serial = clk::now();
        uint32_t err = 0, err_unitig = 0;
        st = pf_call_resolve(ctx_, &nk, &err, &err_unitig);
        if (st != PF_OK) return fail(st, std::string(tag_) + "::PloidyEstimation(): scan: " + pf_last_error(ctx_));
        if (err == 1) return fail(PF_ERR_MISSING_KMER, "CDBG::readCov(): a kmer of unitig " + std::to_string(err_unitig + 1) + " can not found .");
        if (err == 2) return fail(PF_ERR_ARG, "CDBG::PloidyEstimation(): exit of a bubble is not reachable");
        times_.scan_serial_s = since(t_serial);
    } else {
    cx_.sides.ensure(ctx_, std::max<uint64_t>(n_sides, 1));
    cx_.kept.ensure(ctx_, std::max<uint64_t>(n_sides, 1));
    st = pf_call_sides(ctx_, cx_.sides.p, cx_.sides.cap);
    if (st != PF_OK) return fail(st, std::string(tag_) + "::PloidyEstimation(): scan: " + pf_last_error(ctx_));
    // the driver loop itself, walked on the host (PF_SCAN=host: the independent check of pf_call_resolve)
    const auto t_serial = clk::now();
    const pf_call_side *sides = cx_.sides.p;
    uint32_t *kept = cx_.kept.p;
    if (sync_state_to_host()) return status_;
    std::vector<uint8_t> fl(flags_);
    for (uint64_t ri = 0; ri < n_sides; ++ri) {
        const pf_call_side &r = sides[ri];
        const uint32_t u = r.u;
        const uint8_t own = r.plus_side ? B_PLUS : B_MINUS;
        if (!(fl[u] & own)) continue;
        if (r.kind == 1) { fl[u] &= (uint8_t)~own; continue; }
        if (r.err == 1)
            return fail(PF_ERR_MISSING_KMER, "CDBG::readCov(): a kmer of unitig " + std::to_string(r.err_unitig + 1) + " can not found .");
        if (r.err == 2) return fail(PF_ERR_ARG, "CDBG::PloidyEstimation(): exit of a bubble is not reachable");
        fl[u] &= (uint8_t)~own;
        if (r.kind == 2) continue;
        if (r.aligned) kept[nk++] = (uint32_t)ri;
        fl[r.exit_ov >> 1] &= (uint8_t) ~(plus_of(r.exit_ov) ? B_MINUS : B_PLUS);
    }
    times_.scan_serial_s = since(t_serial);
    st = pf_call_select(ctx_, kept, nk);
    if (st != PF_OK) return fail(st, std::string(tag_) + "::PloidyEstimation(): " + pf_last_error(ctx_));
    }
    times_.scan_s = since(t0);
    n_tasks = nk;
    return 0;
}

// ---- the same in steps, for a bubble list cut over several ranks ----
int CDBG::ploidy_select(int lower, int upper, uint64_t &n_bubbles) { return ploidy_select(std::vector<std::pair<int, int>>{{lower, upper}}, n_bubbles); }

int CDBG::ploidy_select(const std::vector<std::pair<int, int>> &cutoff, uint64_t &n_bubbles) {
    if (status_) return status_;
    if (col_ && !colored_resident_) return fail(PF_ERR_ARG, "CCDBG::ploidy_select(): the colour sets are not on the device (PF_CALL=host?)");
    if (cutoff.size() != (col_ ? col_->n_colors : 1u))
        return fail(PF_ERR_ARG, std::string(tag_) + "::ploidy_select(): one (lower, upper) cutoff" + (col_ ? " per colour" : "") + " is required");
    times_.tasks = times_.align_jobs = times_.site_strings = 0;
    times_.snp_jobs = times_.pair_jobs = times_.wave_jobs = times_.stack_jobs = 0;
    allele_[0] = allele_[1] = allele_[2] = allele_[3] = 0;
    core_cov_ = core_num_ = 0;
    const bool keep = resident_;
    resident_ = true;   // coverage must land in the device arrays the scan reads
    const int rc = call_select(cutoff, n_bubbles);
    resident_ = keep;
    return rc;
}

int CDBG::ploidy_align(uint64_t t0, uint64_t t1, uint64_t &n_called) {
    if (status_) return status_;
    const auto t = clk::now();
    const int st = pf_call_align(ctx_, t0, t1, (uint32_t)std::min<size_t>(complex_size_, 1u << 20), sc_.match, sc_.mismatch, sc_.gap, &slice_res_);
    if (st != PF_OK) {
        const std::string e = pf_last_error(ctx_);
        return fail(st, e.rfind("CDBG::", 0) == 0 ? e : std::string(tag_) + "::PloidyEstimation(): " + e);
    }
    n_called = slice_res_.n_called;
    slice_nb_ = t1 - t0;
    times_.tasks = t1 - t0;
    times_.align_jobs = slice_res_.align_jobs;
    times_.snp_jobs = slice_res_.snp_jobs; times_.pair_jobs = slice_res_.pair_jobs; times_.wave_jobs = slice_res_.wave_jobs; times_.stack_jobs = slice_res_.stack_jobs;
    times_.site_strings = slice_res_.site_strings;
    times_.align_s = since(t);
    return 0;
}

int CDBG::ploidy_text(uint64_t var_count_base, uint64_t sizes[PF_CALL_STREAMS], uint64_t counters[8]) {
    if (status_) return status_;
    const auto t = clk::now();
    // the count pass over the whole slice: the sizes the ranks exchange; the text itself is made piece by piece in ploidy_write, each
    // piece formatted while the one before crosses PCIe and the one before that is copied into the files
    slice_var_base_ = var_count_base;
    const uint64_t called = slice_res_.n_called;
    const int st = pf_call_text_sizes(ctx_, 0, 0, slice_nb_, var_count_base, &slice_res_);
    if (st != PF_OK) return fail(st, std::string(tag_) + "::PloidyEstimation(): " + pf_last_error(ctx_));
    slice_res_.n_called = called;
    for (int s = 0; s < PF_CALL_STREAMS; ++s) sizes[s] = slice_nb_ ? slice_res_.text_len[s] : 0;
    for (int a = 0; a < 4; ++a) { counters[a] = slice_res_.allele[a]; allele_[a] = slice_res_.allele[a]; }
    counters[4] = core_cov_ = slice_res_.core_cov;
    counters[5] = core_num_ = slice_res_.core_num;
    counters[6] = slice_res_.n_called;
    counters[7] = times_.tasks;
    times_.format_s = since(t);
    return 0;
}

// this rank's slabs into the shared result files, each at the offset the ranks before it leave; the last writer of a run (or any
// one rank, after a barrier) passes truncate = true to cut the files to their total length
int CDBG::ploidy_write(const std::string &outpre, const uint64_t offsets[PF_CALL_STREAMS], const uint64_t totals[PF_CALL_STREAMS], bool truncate) {
    if (status_) return status_;
    if (ensure_dir()) return status_;
    const auto t = clk::now();
    const unsigned T = threads_ ? threads_ : 1;
    MappedOut *maps = out_maps_.get();
    for (int s = 0; s < PF_CALL_STREAMS; ++s)
        if (maps[s].open_for(outdir_ + "/" + outpre + kStreamSuffix[s])) return fail(PF_ERR_ARG, "CDBG:: Open " + outpre + kStreamSuffix[s] + " file error");
    // every rank first gives the files their final length -- the same value from all of them, so the order does not matter and no
    // rank ever cuts off what another has written -- and then stores its slab at its offsets
    int rc = 0;
    if (truncate)
        for (int s = 0; s < PF_CALL_STREAMS; ++s) rc |= maps[s].finish(totals[s]);
    // The slice in pieces of whole text batches: piece i + 1 is formatted on the device (K-TEXT's write pass is not waited for) while
    // piece i crosses PCIe into one of two pinned buffers and piece i - 1 is copied from the other into the mapped files.
    const uint64_t CHUNK = std::min<uint64_t>(std::max<uint64_t>((uint64_t)batch_bubbles_ * 4, 1), (uint64_t)1 << 24);
    const uint64_t n_pieces = (slice_nb_ + CHUNK - 1) / CHUNK;
    std::vector<pf_call_result> pres((size_t)n_pieces);
    uint64_t running[PF_CALL_STREAMS];
    for (int s = 0; s < PF_CALL_STREAMS; ++s) running[s] = offsets[s];
    last_allfre_.clear();
    last_allfre_file_.clear();
    int st = PF_OK;
    auto format = [&](uint64_t i) {
        const uint64_t first = i * CHUNK, count = std::min<uint64_t>(CHUNK, slice_nb_ - first);
        return pf_call_text_range_lane(ctx_, 0, (int)(i % PF_CALL_SLABS), first, count, slice_var_base_, &pres[(size_t)i]);
    };
    auto piece_bytes = [&](uint64_t i) {
        uint64_t all = 0;
        for (int s = 0; s < PF_CALL_STREAMS; ++s) all += pres[(size_t)i].text_len[s];
        return all;
    };
    auto fetch = [&](uint64_t i) {
        PinnedBuf<char> &hb = cx_.slab[i & 1];
        const uint64_t all = piece_bytes(i);
        hb.ensure(ctx_, std::max<uint64_t>(all, 1));
        return pf_call_fetch_range(ctx_, (int)(i % PF_CALL_SLABS), 0, hb.p, all, (int)(i & 1));
    };
    if (n_pieces) st = format(0);
    if (st == PF_OK && n_pieces) st = fetch(0);
    for (uint64_t i = 0; i < n_pieces && st == PF_OK; ++i) {
        if (i + 1 < n_pieces) st = format(i + 1);
        if (st == PF_OK) st = pf_call_fetch_wait(ctx_, (int)(i & 1));
        if (st == PF_OK && i + 1 < n_pieces) st = fetch(i + 1);   // (its buffer was copied out two pieces ago)
        if (st != PF_OK) break;
        const pf_call_result &r = pres[(size_t)i];
        const char *src = cx_.slab[i & 1].p;
        CopySpan spans[PF_CALL_STREAMS];
        size_t n_spans = 0;
        uint64_t at = 0;
        for (int s = 0; s < PF_CALL_STREAMS; ++s) {
            const uint64_t len = r.text_len[s];
            if (len) {
                char *dst = maps[s].prepare(running[s], len);
                if (dst) spans[n_spans++] = CopySpan{dst, src + at, len};
                else rc |= maps[s].write(running[s], src + at, len, T);   // (a file that cannot be mapped: through the descriptor)
            }
            if (s == 0) last_allfre_.append(src + at, len);
            running[s] += len;
            at += len;
            out_bytes_ += len;
        }
        copy_spans(spans, n_spans, T);
    }
    if (st != PF_OK) return fail(st, std::string(tag_) + "::PloidyEstimation(): " + pf_last_error(ctx_));
    for (int s = 0; s < PF_CALL_STREAMS; ++s)
        if (running[s] - offsets[s] != (slice_nb_ ? slice_res_.text_len[s] : 0)) return fail(PF_ERR_ARG, std::string(tag_) + "::PloidyEstimation(): the pieces of a slice do not add up to its size");
    times_.write_s = since(t);
    if (rc) return fail(PF_ERR_ARG, "CDBG:: write error on the result files");
    return 0;
}

int CDBG::ploidy_estimation_resident(const std::string &outpre, const std::vector<std::pair<int, int>> &cutoff, const size_t &thr) {
    const auto t_all = clk::now();
    clock_t c0 = clock();
    if (!quiet_) printf("%s::PloidyEstimation():  Analyzing superbubbles to generate sites' information\n", tag_);
    if (write_files_ && ensure_dir()) return status_;
    const unsigned T = threads_ ? threads_ : (unsigned)std::max<size_t>(thr, 1);
    times_.cov_device_s = times_.tasks_s = times_.align_s = times_.sites_s = times_.format_s = times_.write_s = 0;
    times_.tasks = times_.align_jobs = times_.site_strings = 0;
    times_.snp_jobs = times_.pair_jobs = times_.wave_jobs = times_.stack_jobs = 0;
    times_.align_build_s = times_.align_device_s = times_.align_post_s = times_.align_choose_s = times_.scan_s = 0;
    allele_[0] = allele_[1] = allele_[2] = allele_[3] = 0;
    core_cov_ = core_num_ = 0;
    const bool trace = getenv("PF_TRACE_PLOIDY") != nullptr;
    auto tp = [&](const char *what) { if (trace) fprintf(stderr, "[ploidy] %-28s %.2f ms\n", what, since(t_all) * 1e3); };

    // The ten result files stay open and mapped between passes (MappedOut); nothing is truncated up front (freeing the pages of
    // an earlier run costs milliseconds): each file is cut to its final length at the end.  A helper thread (re)opens them
    // while the scan runs.
    struct OutFile {
        std::string name;
        uint64_t bytes = 0;
        int rc = 0;
    };
    std::vector<OutFile> files(PF_CALL_STREAMS);
    for (int s = 0; s < PF_CALL_STREAMS; ++s) files[(size_t)s].name = outpre + kStreamSuffix[s];
    MappedOut *maps = out_maps_.get();
    int open_failed = -1;
    std::thread opener;
    if (write_files_)
        opener = std::thread([&] {
            for (size_t i = 0; i < files.size(); ++i)
                if (maps[i].open_for(outdir_ + "/" + files[i].name)) { open_failed = (int)i; return; }
        });
    struct OpenerGuard {  // every early return below must not leave the helper running
        std::thread &t;
        ~OpenerGuard() { if (t.joinable()) t.join(); }
    } opener_guard{opener};

    uint64_t n_tasks = 0;
    if (call_select(cutoff, n_tasks)) return status_;
    tp("scan + selection done");
    if (opener.joinable()) opener.join();
    if (open_failed >= 0) return fail(PF_ERR_ARG, "CDBG:: Open " + files[(size_t)open_failed].name + " file error");
    last_allfre_.clear();
    last_allfre_file_.clear();
    // Fresh result files (a one-shot run: every run of the CLI) get their pages NOW, on helper threads, while the device aligns the
    // first range -- 438 MB of first-touch faults under the writer were half of a first PloidyEstimation at 5 M unitigs.  The sizes
    // are estimates from the number of bubbles and k (a bubble of two paths of 2k - 1 bases, one site); what they miss is extended
    // when it is written, what they overshoot is cut off at the end (MappedOut::finish).  Files that already have their pages
    // (a later pass) are left alone.
    std::thread prefault0;
    std::atomic<bool> prefault0_stop{false};   // set once the last piece is in the files: what the estimates overshoot need not be touched
    struct Prefault0Guard {
        std::thread &t;
        std::atomic<bool> &stop;
        ~Prefault0Guard() { stop.store(true); if (t.joinable()) t.join(); }
    } prefault0_guard{prefault0, prefault0_stop};
    if (write_files_ && n_tasks) {
        const double nbub = 1.1 * (double)n_tasks;
        // (order: pf_call_stream.  Measured at k = 25, tetraploid: 18.4, 157, 18.0, 0.34, 0.07, 0, 30.2, 0.45, 0.08, 0 bytes per bubble;
        // what is reserved beyond the final size is touched here and cut off again at the end -- 8 ms for 60 MB too many)
        const double per_bubble[PF_CALL_STREAMS] = {20, 2.0 * (2 * g_.k + 26), 20, 0.5, 0.25, 0, 34, 0.75, 0.25, 0};
        std::vector<std::pair<int, uint64_t>> fresh;
        for (int s = 0; s < PF_CALL_STREAMS; ++s) {
            const uint64_t est = (uint64_t)(per_bubble[s] * nbub);
            if (est >= (1u << 20) && maps[s].reserve(est) == 0) fresh.push_back({s, est});
        }
        if (!fresh.empty())
            prefault0 = std::thread([maps, fresh, T, &prefault0_stop] {
                constexpr uint64_t STEP = 4u << 20;
                std::vector<std::pair<int, uint64_t>> spans;   // (stream, offset) in 4 MB steps, files interleaved
                uint64_t longest = 0;
                for (auto &f : fresh) longest = std::max(longest, f.second);
                for (uint64_t at = 0; at < longest; at += STEP)
                    for (auto &f : fresh)
                        if (at < f.second) spans.push_back({f.first, at});
                parallel_chunks(spans.size(), 1, std::max(2u, T), [&](size_t i, size_t, size_t) {
                    if (prefault0_stop.load(std::memory_order_relaxed)) return;
                    const int s = spans[i].first;
                    uint64_t end = spans[i].second + STEP;
                    for (auto &f : fresh) if (f.first == s) end = std::min(end, f.second);
                    maps[s].populate(spans[i].second, end);
                });
            });
    }

    // alignseq leaves the device packed (pf_alnpack.hpp: a header per bubble, rows at 3 bits per character) and becomes text in the
    // writer below; PF_ALIGNSEQ_ASCII=1 keeps the device writing the text itself (measurements, and the check that both give one file)
    const bool aln_packed = !getenv("PF_ALIGNSEQ_ASCII");
    if (pf_call_set_alignseq_packed(ctx_, aln_packed ? 1 : 0) != PF_OK) return fail(PF_ERR_HIP, std::string(tag_) + "::PloidyEstimation(): " + pf_last_error(ctx_));
    // the other nine streams -- numbers: sixteen characters -- leave it at four bits a character (K-NIB) and become text in the
    // writer as well; PF_NUMERIC_ASCII=1 keeps them as text on the way (measurements, and the check that both give the same files).
    // The copy to the host is what a pass ends with: 17.7 MB a piece at 55 GB/s, 0.33 ms each, twelve pieces behind the alignment.
    const bool num_packed = !getenv("PF_NUMERIC_ASCII");
    if (pf_call_set_numeric_packed(ctx_, num_packed ? 1 : 0) != PF_OK) return fail(PF_ERR_HIP, std::string(tag_) + "::PloidyEstimation(): " + pf_last_error(ctx_));
    struct PackGuard {   // the sliced calls (ploidy_text, one graph over several ranks) read the text as the device writes it
        pf_ctx *c;
        ~PackGuard() { (void)pf_call_set_alignseq_packed(c, 0); (void)pf_call_set_numeric_packed(c, 0); }
    } pack_guard{ctx_};
    // what stream s of a piece takes in a slab
    auto slab_len = [](const pf_call_result &r, int s) -> uint64_t {
        const uint64_t aln = (s == PF_OUT_ALIGNSEQ && r.alignseq_packed_len) ? r.alignseq_packed_len : r.text_len[s];
        if (!r.numeric_packed) return aln;
        return s == PF_OUT_ALIGNSEQ ? ((aln + 15) & ~15ull) : PF_NUMERIC_PACKED_LEN(r.text_len[s]);
    };

    // ---- pieces: device (this thread) | copy back (fetcher thread) | append to the files (writer thread) ----
    // (PF_BATCH_BUBBLES: tools/fuzz_parity.py drives the CLI through many small pieces and ranges with it)
    const size_t batch_env = [] { const char *e = getenv("PF_BATCH_BUBBLES"); return e ? (size_t)std::max(1, atoi(e)) : (size_t)0; }();
    const size_t batch_now = batch_env ? batch_env : batch_bubbles_;
    const size_t CHUNK = std::min<size_t>(std::max<size_t>(batch_now ? batch_now * 4 : 1, 1), (size_t)1 << 24);
    struct Done {
        pf_call_result res;
        int slab;    // on the device (PF_CALL_SLABS of them: a whole range can be formatted before the alignment kernels of
                     // the next range fill the device); the pinned host slabs alternate, piece b -> b % 2
        int hslab;
        // numeric streams whose nibbles are not their text (a character outside the sixteen): fetched as text by the fetcher
        std::shared_ptr<std::vector<std::vector<char>>> plain;
    };
    std::mutex mu;
    std::condition_variable cv;
    std::deque<Done> ready, towrite;
    size_t fetched = 0;  // pieces whose slabs have been copied off the device
    size_t written = 0;  // pieces whose text is in the files (their host slab is free again)
    bool fetcher_done = false;
    bool stop = false, producer_done = false;
    int wst = PF_OK;
    std::string werr;
    double write_s = 0;
    // piece b travels device slab b % PF_CALL_SLABS -> host slab b % 2 -> files; the PCIe copy of piece b + 1 runs beside the file copy of piece b
    auto thread_failed = [&](const char *who, const std::exception &e) {   // an exception in a helper thread ends the pipeline, not the process
        {
            std::lock_guard<std::mutex> lk(mu);
            wst = PF_ERR_HIP;
            werr = std::string(who) + ": " + e.what();
            stop = true;
            fetcher_done = true;
        }
        cv.notify_all();
    };
    std::thread fetcher([&] {
        try {
        size_t b = 0;
        for (;;) {
            Done d;
            {
                std::unique_lock<std::mutex> lk(mu);
                cv.wait(lk, [&] { return stop || !ready.empty() || producer_done; });
                if (stop || ready.empty()) { fetcher_done = true; lk.unlock(); cv.notify_all(); return; }
                d = ready.front();
                ready.pop_front();
                cv.wait(lk, [&] { return stop || b < written + 2; });   // host slab b % 2 was last used by piece b - 2
                if (stop) { fetcher_done = true; lk.unlock(); cv.notify_all(); return; }
            }
            uint64_t total = 0, fetch_len[PF_CALL_STREAMS];
            for (int s = 0; s < PF_CALL_STREAMS; ++s) { fetch_len[s] = slab_len(d.res, s); total += fetch_len[s]; }
            PinnedBuf<char> &hb = cx_.slab[d.hslab];
            hb.ensure(ctx_, std::max<uint64_t>(total, 1) + 16);
            int st = pf_call_fetch_slab(ctx_, d.slab, hb.p, fetch_len);
            if (st == PF_OK && d.res.numeric_packed) {
                uint32_t flagged = 0;   // (the tail behind the streams: bit s = stream s holds something else than numbers)
                memcpy(&flagged, hb.p + total, 4);
                if (flagged) {
                    d.plain = std::make_shared<std::vector<std::vector<char>>>((size_t)PF_CALL_STREAMS);
                    for (int s = 0; s < PF_CALL_STREAMS && st == PF_OK; ++s)
                        if (((flagged >> s) & 1) && d.res.text_len[s]) {
                            (*d.plain)[(size_t)s].resize(d.res.text_len[s]);
                            st = pf_call_fetch_text(ctx_, d.slab, s, (*d.plain)[(size_t)s].data(), d.res.text_len[s]);
                        }
                }
            }
            {
                std::lock_guard<std::mutex> lk(mu);
                fetched = b + 1;
                if (st != PF_OK) { wst = st; werr = "copy of a text slab failed"; stop = true; fetcher_done = true; }
                else towrite.push_back(d);
            }
            cv.notify_all();
            if (st != PF_OK) return;
            if (trace) fprintf(stderr, "[ploidy]   batch %zu fetched (%.1f MB) %.2f ms\n", b, total / 1e6, since(t_all) * 1e3);
            ++b;
        }
        } catch (const std::exception &e) { thread_failed("copy of a text slab", e); }
    });
    std::thread writer([&] {
        try {
        size_t b = 0;
        for (;;) {
            Done d;
            {
                std::unique_lock<std::mutex> lk(mu);
                cv.wait(lk, [&] { return stop || !towrite.empty() || fetcher_done; });
                if (stop || towrite.empty()) return;
                d = towrite.front();
                towrite.pop_front();
            }
            const auto tw = clk::now();
            uint64_t total = 0, off[PF_CALL_STREAMS + 1];
            for (int s = 0; s < PF_CALL_STREAMS; ++s) { off[s] = total; total += slab_len(d.res, s); }
            PinnedBuf<char> &hb = cx_.slab[d.hslab];
            // append: every stream at its running offset through a shared mapping, copied by all threads side by side
            if (write_files_) {
                CopySpan spans[PF_CALL_STREAMS];
                bool nib_span[PF_CALL_STREAMS] = {};   // the span's source is nibbles: expanded instead of copied
                size_t n_spans = 0;
                pf::AlnPackPiece packed;   // alignseq of this piece, when it came packed: written out of its records, group by group
                char *packed_dst = nullptr;
                std::vector<char> packed_tmp;
                for (int s = 0; s < PF_CALL_STREAMS; ++s) {
                    if (!d.res.text_len[s]) continue;
                    char *dst = maps[s].prepare(files[(size_t)s].bytes, d.res.text_len[s]);
                    if (s == PF_OUT_ALIGNSEQ && d.res.alignseq_packed_len) {
                        if (!packed.parse(reinterpret_cast<const uint8_t *>(hb.p + off[s]), d.res.alignseq_packed_len)) { files[(size_t)s].rc = 1; continue; }
                        if (!dst) { packed_tmp.resize(d.res.text_len[s]); dst = packed_tmp.data(); }   // (no mapping: text first, then pwrite)
                        packed_dst = dst;
                        continue;
                    }
                    const char *src = hb.p + off[s];
                    bool nib = d.res.numeric_packed != 0 && s != PF_OUT_ALIGNSEQ;   // (alignseq as text, PF_ALIGNSEQ_ASCII: copied)
                    if (nib && d.plain && !(*d.plain)[(size_t)s].empty()) { src = (*d.plain)[(size_t)s].data(); nib = false; }   // (fetched as text)
                    if (dst) {
                        nib_span[n_spans] = nib;
                        spans[n_spans++] = CopySpan{dst, src, d.res.text_len[s]};
                    } else if (nib) {   // (no mapping: text first, then pwrite)
                        std::vector<char> tmp(d.res.text_len[s]);
                        pf::nibble_expand(tmp.data(), reinterpret_cast<const uint8_t *>(src), d.res.text_len[s]);
                        if (maps[s].write(files[(size_t)s].bytes, tmp.data(), tmp.size(), T)) files[(size_t)s].rc = 1;
                    } else if (maps[s].write(files[(size_t)s].bytes, src, d.res.text_len[s], T)) files[(size_t)s].rc = 1;   // (no mapping: pwrite)
                }
                // the other streams' copies and alignseq's groups in one dispatch of the pool
                constexpr uint64_t PIECE = 1u << 20;
                std::vector<CopySpan> cut;
                std::vector<char> cut_nib;
                constexpr uint64_t NIB_PIECE = 128u << 10;   // (expanding is a table look-up a byte: an eighth of what a thread copies in the same time)
                for (size_t i = 0; i < n_spans; ++i) {
                    const uint64_t step = nib_span[i] ? NIB_PIECE : PIECE;   // (both even: a piece of text starts on a byte of nibbles)
                    for (uint64_t at = 0; at < spans[i].len; at += step) {
                        cut.push_back(CopySpan{spans[i].dst + at, spans[i].src + (nib_span[i] ? at / 2 : at), std::min<uint64_t>(step, spans[i].len - at)});
                        cut_nib.push_back(nib_span[i] ? 1 : 0);
                    }
                }
                constexpr uint64_t GROUPS_PER_TASK = 16;   // 4096 bubbles, about 600 KB of text
                const size_t n_aln = packed_dst ? (size_t)((packed.n_groups + GROUPS_PER_TASK - 1) / GROUPS_PER_TASK) : 0;
                std::atomic<int> bad{0};
                parallel_chunks(n_aln + cut.size(), 1, T, [&](size_t i, size_t, size_t) {
                    if (i >= n_aln) {   // (alignseq first: the larger tasks)
                        const CopySpan &c = cut[i - n_aln];
                        if (cut_nib[i - n_aln]) pf::nibble_expand(c.dst, reinterpret_cast<const uint8_t *>(c.src), c.len);
                        else memcpy(c.dst, c.src, (size_t)c.len);
                        return;
                    }
                    const uint64_t g0 = (uint64_t)i * GROUPS_PER_TASK, g1 = std::min<uint64_t>(packed.n_groups, g0 + GROUPS_PER_TASK);
                    uint64_t t0_, r0_, t1_, r1_;
                    packed.entry(g0, t0_, r0_);
                    packed.entry(g1, t1_, r1_);
                    if (t1_ > d.res.text_len[PF_OUT_ALIGNSEQ] || t0_ > t1_ || r0_ > r1_ ||
                        (uint64_t)(packed.records - packed.base) + r1_ > d.res.alignseq_packed_len) { bad = 1; return; }
                    char *end = pf::alnpack_expand(packed.records + r0_, packed.records + r1_, packed_dst + t0_, packed_dst + t1_);
                    if (end != packed_dst + t1_) bad = 1;   // (nullptr: a record outside its group's bytes, or rows beyond the group's text)
                });
                if (bad) files[PF_OUT_ALIGNSEQ].rc = 1;
                if (!packed_tmp.empty() && maps[PF_OUT_ALIGNSEQ].write(files[PF_OUT_ALIGNSEQ].bytes, packed_tmp.data(), packed_tmp.size(), T)) files[PF_OUT_ALIGNSEQ].rc = 1;
            }
            if (!write_files_) {   // (else: read back from the file on demand)
                if (d.plain && !(*d.plain)[0].empty()) last_allfre_.append((*d.plain)[0].data(), d.res.text_len[0]);
                else if (d.res.numeric_packed) {
                    const size_t at0 = last_allfre_.size();
                    last_allfre_.resize(at0 + d.res.text_len[0]);
                    pf::nibble_expand(&last_allfre_[at0], reinterpret_cast<const uint8_t *>(hb.p + off[0]), d.res.text_len[0]);
                } else last_allfre_.append(hb.p + off[0], d.res.text_len[0]);
            }
            for (int s = 0; s < PF_CALL_STREAMS; ++s) files[(size_t)s].bytes += d.res.text_len[s];
            write_s += since(tw);
            if (trace) fprintf(stderr, "[ploidy]   batch %zu written %.2f ms\n", b, since(t_all) * 1e3);
            ++b;
            { std::lock_guard<std::mutex> lk(mu); written = b; }
            cv.notify_all();
        }
        } catch (const std::exception &e) { thread_failed("writing the result files", e); }
    });
    struct WriterGuard {  // joined on every way out
        std::thread &t, &t2;
        std::mutex &mu;
        std::condition_variable &cv;
        bool &stop;
        ~WriterGuard() {
            if (!t.joinable() && !t2.joinable()) return;
            { std::lock_guard<std::mutex> lk(mu); stop = true; }
            cv.notify_all();
            if (t.joinable()) t.join();
            if (t2.joinable()) t2.join();
        }
    } writer_guard{fetcher, writer, mu, cv, stop};

    int rc = PF_OK;
    std::string rc_err;
    std::thread prefault;
    struct PrefaultGuard {
        std::thread &t;
        ~PrefaultGuard() { if (t.joinable()) t.join(); }
    } prefault_guard{prefault};
    const auto t_dev = clk::now();
    // Alignment in large launches (every kernel's tail -- one wavefront finishing the heaviest bubble -- is paid once per
    // launch), text in pieces of CHUNK bubbles that are fetched and written while the next piece is formatted.  A long list is
    // aligned in a few ranges, in turn into the result lanes of the context: the formatter thread turns range r into
    // text -- and the fetcher moves it over PCIe, the slowest stage -- while this thread aligns range r + 1.
    uint64_t ALIGN = std::min<uint64_t>((uint64_t)CHUNK * std::max<size_t>(align_pieces_, 1), (uint64_t)1 << 24);
    {
        const int ranges_env = [] { const char *e = getenv("PF_ALIGN_RANGES"); return e ? atoi(e) : 0; }();   // measurements (read per pass: tools/ab_pass.py)
        const uint64_t pieces = (n_tasks + CHUNK - 1) / CHUNK;
        const uint64_t n_ranges = ranges_env > 0 ? (uint64_t)ranges_env : (pieces >= 4 ? 2 : 1);
        if (n_ranges > 1) ALIGN = std::min<uint64_t>(ALIGN, std::max<uint64_t>(1, (pieces + n_ranges - 1) / n_ranges) * CHUNK);
    }
    int first_env = 0;   // pieces in the first range
    constexpr int LANES = PF_CALL_LANES;
    // aligner threads: pf_call_align_lane on that many lanes side by side (below)
    const int aligners_wanted = [&] { const char *e = getenv("PF_ALIGN_THREADS"); return e ? std::max(1, std::min(LANES, atoi(e))) : std::min(2, LANES); }();
    {
        // Where two ranges are cut.  One aligner: what a pass ends with is the text of the LAST range, formatted and copied with nothing
        // beside it, so the first range takes five eighths of the pieces (measured with a knob since removed: 5 + 3 of eight pieces
        // 20.08 ms, 4 + 4 20.28, 6 + 2 20.50, 3 + 5 20.90).  Two aligners: both ranges are on the device from the start and end
        // within half a millisecond of each other; in halves (4 + 4 17.19 ms, 5 + 3 17.66, 3 + 5 17.45, 2 + 6 17.89).
        const uint64_t pieces = (n_tasks + CHUNK - 1) / CHUNK;
        if (first_env <= 0 && !getenv("PF_ALIGN_RANGES") && pieces >= 4)
            first_env = (int)std::min<uint64_t>(aligners_wanted >= 2 ? (pieces + 1) / 2 : (pieces * 5 + 7) / 8, ((uint64_t)1 << 24) / CHUNK);
    }
    // The ranges of this pass, and where each lies: range r is aligned into lane r % LANES by whichever aligner thread takes it
    // next, and formatted -- in range order -- once it is there.  Two aligners (PF_ALIGN_THREADS) call pf_call_align_lane on two
    // lanes side by side.  A range's chain of a dozen dependent launches costs 1.7 ms whatever it holds (196 k bubbles 2.3 ms,
    // 983 k 4.5 ms, 1.55 M 6.6 ms: every kernel is at least one round of its slowest work item long); two ranges side by side on
    // streams of equal priority only time-slice (their kernels are sized to fill the device) and the pass got SLOWER, 18.2 -> 19.3
    // ms; with the later lanes' streams at the lowest priority (pf::lane_stream_create) the first range is not held up, the second
    // fills what it leaves idle, and both are aligned 5.8 ms after the scan instead of 8.8: 18.45 -> 17.19 ms per pass together with
    // K-TEXT's two streams (profiles/r16_experiments.txt).
    struct Range {
        uint64_t a0 = 0, a1 = 0;
        int lane = 0;
        bool aligned = false;
        uint64_t n_called = 0;
    };
    std::vector<Range> ranges;
    for (uint64_t a0 = 0, a1 = 0; a0 < n_tasks; a0 = a1) {
        a1 = std::min<uint64_t>(n_tasks, a0 + (first_env > 0 && ranges.empty() ? (uint64_t)first_env * CHUNK : ALIGN));
        Range r;
        r.a0 = a0; r.a1 = a1; r.lane = (int)(ranges.size() % (size_t)LANES);
        ranges.push_back(r);
    }
    const int aligners = (int)std::min<size_t>(ranges.size(), (size_t)aligners_wanted);
    size_t next_range = 0;         // the next range an aligner takes
    size_t ranges_formatted = 0;   // ranges whose last piece has been formatted: their lane may be aligned into again
    std::thread formatter([&] {
        struct Last {   // on every way out: the fetcher learns that no further piece will come
            std::mutex &mu;
            std::condition_variable &cv;
            bool &producer_done;
            ~Last() {
                { std::lock_guard<std::mutex> lk(mu); producer_done = true; }
                cv.notify_all();
            }
        } last{mu, cv, producer_done};
        try {
        size_t b = 0;   // text piece number: slabs alternate
        uint64_t var_base = 0;
        for (size_t ri = 0; ri < ranges.size(); ++ri) {
            Range r;
            {
                std::unique_lock<std::mutex> lk(mu);
                cv.wait(lk, [&] { return stop || ranges[ri].aligned; });
                if (stop) return;
                r = ranges[ri];
            }
            const uint64_t rn = r.a1 - r.a0;
            for (uint64_t p0 = 0; p0 < rn; p0 += CHUNK, ++b) {
                const uint64_t count = std::min<uint64_t>(CHUNK, rn - p0);
                {   // its device slab was last used by piece b - PF_CALL_SLABS
                    std::unique_lock<std::mutex> lk(mu);
                    cv.wait(lk, [&] { return stop || b < fetched + PF_CALL_SLABS; });
                    if (stop) return;
                }
                Done d;
                d.slab = (int)(b % PF_CALL_SLABS);
                d.hslab = (int)(b & 1);
                const int st = pf_call_text_range_lane(ctx_, r.lane, d.slab, p0, count, var_base, &d.res);
                if (st != PF_OK) {
                    { std::lock_guard<std::mutex> lk(mu); if (rc == PF_OK) { rc = st; rc_err = pf_last_error(ctx_); } stop = true; }
                    cv.notify_all();
                    return;
                }
                if (trace) fprintf(stderr, "[ploidy]   piece %zu formatted on the device %.2f ms\n", b, since(t_all) * 1e3);
                if (b == 0 && write_files_) {
                    // first piece of a pass: fresh result files get their final size (extrapolated from this piece, cut to the true
                    // one at the end) and their pages now, on helper threads, instead of fault by fault under the writer
                    const double scale = 1.05 * (double)n_tasks / (double)std::max<uint64_t>(1, count);
                    std::vector<std::pair<int, uint64_t>> fresh;
                    for (int s = 0; s < PF_CALL_STREAMS; ++s) {
                        const uint64_t est = (uint64_t)((double)d.res.text_len[s] * scale) + 4096;
                        if (d.res.text_len[s] && maps[s].reserve(est) == 0) fresh.push_back({s, est});
                    }
                    if (!fresh.empty())
                        prefault = std::thread([maps, fresh, T] {
                            constexpr uint64_t STEP = 4u << 20;
                            std::vector<std::pair<int, uint64_t>> spans;   // (stream, offset) in 4 MB steps, files interleaved
                            uint64_t longest = 0;
                            for (auto &f : fresh) longest = std::max(longest, f.second);
                            for (uint64_t at = 0; at < longest; at += STEP)
                                for (auto &f : fresh)
                                    if (at < f.second) spans.push_back({f.first, at});
                            parallel_chunks(spans.size(), 1, std::max(2u, T / 2), [&](size_t i, size_t, size_t) {
                                const int s = spans[i].first;
                                uint64_t end = spans[i].second + STEP;
                                for (auto &f : fresh) if (f.first == s) end = std::min(end, f.second);
                                maps[s].populate(spans[i].second, end);
                            });
                        });
                }
                for (int a = 0; a < 4; ++a) allele_[a] += d.res.allele[a];
                core_cov_ += d.res.core_cov;
                core_num_ += d.res.core_num;
                { std::lock_guard<std::mutex> lk(mu); ready.push_back(d); }
                cv.notify_all();
            }
            var_base += r.n_called;
            { std::lock_guard<std::mutex> lk(mu); ++ranges_formatted; }
            cv.notify_all();
        }
        } catch (const std::exception &e) { thread_failed("formatting the result rows", e); }
    });
    struct FormatterGuard {  // joined on every way out, before the prefault thread it may have started
        std::thread &t;
        std::mutex &mu;
        std::condition_variable &cv;
        bool &stop;
        ~FormatterGuard() {
            if (!t.joinable()) return;
            { std::lock_guard<std::mutex> lk(mu); stop = true; }
            cv.notify_all();
            t.join();
        }
    } formatter_guard{formatter, mu, cv, stop};
    auto aligner = [&] {
        try {
        for (;;) {
            size_t ri;
            {   // the next range; its lane was last read by the text of range ri - LANES
                std::unique_lock<std::mutex> lk(mu);
                if (stop || next_range >= ranges.size()) return;
                ri = next_range++;
                cv.wait(lk, [&] { return stop || ri < ranges_formatted + (size_t)LANES; });
                if (stop) return;
            }
            const Range r = ranges[ri];
            pf_call_result ar;
            const int st = pf_call_align_lane(ctx_, r.lane, r.a0, r.a1, (uint32_t)std::min<size_t>(complex_size_, 1u << 20), sc_.match, sc_.mismatch, sc_.gap, &ar);
            if (st != PF_OK) {
                { std::lock_guard<std::mutex> lk(mu); if (rc == PF_OK) { rc = st; rc_err = pf_last_error(ctx_); } stop = true; }
                cv.notify_all();
                return;
            }
            if (trace) fprintf(stderr, "[ploidy]   range %zu: %llu bubbles aligned on the device %.2f ms\n", ri, (unsigned long long)(r.a1 - r.a0), since(t_all) * 1e3);
            {
                std::lock_guard<std::mutex> lk(mu);
                times_.tasks += r.a1 - r.a0;
                times_.align_jobs += ar.align_jobs;
                times_.snp_jobs += ar.snp_jobs; times_.pair_jobs += ar.pair_jobs; times_.wave_jobs += ar.wave_jobs; times_.stack_jobs += ar.stack_jobs;
                times_.site_strings += ar.site_strings;
                ranges[ri].n_called = ar.n_called;
                ranges[ri].aligned = true;
            }
            cv.notify_all();
        }
        } catch (const std::exception &e) { thread_failed("aligning the bubbles", e); }
    };
    {
        std::vector<std::thread> extra;
        struct ExtraGuard {
            std::vector<std::thread> &v;
            ~ExtraGuard() { for (auto &t : v) if (t.joinable()) t.join(); }
        } extra_guard{extra};
        for (int a = 1; a < aligners; ++a) extra.emplace_back(aligner);
        aligner();
    }
    formatter.join();
    times_.align_s = since(t_dev);
    fetcher.join();
    writer.join();
    if (prefault.joinable()) prefault.join();
    prefault0_stop.store(true);
    if (prefault0.joinable()) prefault0.join();
    tp("pipeline done");
    if (rc != PF_OK) {
        // the device layer words the reference's own messages (missing k-mer, site string outside its row)
        return fail(rc, rc_err.rfind("CDBG::", 0) == 0 ? rc_err : std::string(tag_) + "::PloidyEstimation(): " + rc_err);
    }
    if (wst != PF_OK) return fail(wst, std::string(tag_) + "::PloidyEstimation(): " + werr);
    auto t0 = clk::now();
    if (join_pending_write() || join_pending_ids()) return status_;
    tp("super_bubble.txt joined");
    for (size_t i = 0; i < files.size(); ++i) {
        OutFile &of = files[i];
        out_bytes_ += of.bytes;
        const auto tfin = clk::now();
        if (write_files_ && maps[i].finish(of.bytes)) of.rc = 1;
        if (trace) fprintf(stderr, "[ploidy]   %s finished at %llu bytes in %.3f ms\n", of.name.c_str(), (unsigned long long)of.bytes, since(tfin) * 1e3);
        if (of.rc) return fail(PF_ERR_ARG, "CDBG:: write error on " + of.name);
    }
    if (write_files_) { last_allfre_file_ = outdir_ + "/" + files[0].name; last_allfre_bytes_ = files[0].bytes; }
    write_s += since(t0);
    times_.write_s = write_s;
    tp("files closed");
    times_.ploidy_total_s = since(t_all);
    if (!quiet_) {
        printf(mt_format_ ? "%s::PloidyEstimation(): Cpu time : %gs\n" : "%s::PloidyEstimation():  Cpu time : %gs\n", tag_,
               (double)(clock() - c0) / CLOCKS_PER_SEC);   // (src/CDBG.cpp:2612-2615 vs 1683-1686)
        printf(mt_format_ ? "%s::PloidyEstimation(): Real time : %gs\n" : "%s::PloidyEstimation():  Real time : %gs\n", tag_, times_.ploidy_total_s);
        printf("%s::PloidyEstimation(): Alleles in SuperBubbles  :\t2 :%llu\t3 :%llu\t4 :%llu\t5 :%llu\n", tag_,
               (unsigned long long)allele_[0], (unsigned long long)allele_[1], (unsigned long long)allele_[2],
               (unsigned long long)allele_[3]);
        // the reference divides unguarded (src/CDBG.cpp:1703) and dies with SIGFPE when no site exists
        if (core_num_) printf("%s::PloidyEstimation(): Sites' Average Coverage:%d\n", tag_, (int)(core_cov_ / core_num_));
    }
    return 0;
}

const std::string &CDBG::last_allele_frequency() const {
    if (!last_allfre_file_.empty()) {
        last_allfre_.clear();
        if (FILE *f = fopen(last_allfre_file_.c_str(), "rb")) {
            last_allfre_.resize(last_allfre_bytes_);
            const size_t got = fread(&last_allfre_[0], 1, last_allfre_bytes_, f);
            last_allfre_.resize(got);
            fclose(f);
        }
        last_allfre_file_.clear();
    }
    return last_allfre_;
}

}  // namespace pfh
