// pfh::CDBG: construction, output plumbing, MyUnitig state and the order-dependent commit replay.
#include <fcntl.h>
#include <sched.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <chrono>
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
#include "pf_trace.hpp"

namespace pfh {

// The host side of a pass is bound by memory latency (the commit replay chases per-unitig state at random) and feeds the GPU
// over PCIe: on a multi-socket host both want the threads and their memory on the socket the GPU hangs off.  Restricts the
// calling thread -- and with it every thread created from here on: the work pool, the pipeline stages -- to the CPUs of the
// GPU's NUMA node.  Does nothing on single-node hosts, when sysfs does not tell, or with PF_NO_NUMA_BIND set.
static void bind_to_device_node(pf_ctx *ctx) {
    if (getenv("PF_NO_NUMA_BIND")) return;
    char bdf[64] = {0};
    if (pf_device_pci_bus_id(ctx, bdf, sizeof bdf) != PF_OK) return;
    for (char *c = bdf; *c; ++c) *c = (char)tolower(*c);
    int node = -1;
    if (FILE *f = fopen((std::string("/sys/bus/pci/devices/") + bdf + "/numa_node").c_str(), "r")) {
        if (fscanf(f, "%d", &node) != 1) node = -1;
        fclose(f);
    }
    if (node < 0) return;
    std::string list;
    if (FILE *f = fopen(("/sys/devices/system/node/node" + std::to_string(node) + "/cpulist").c_str(), "r")) {
        char buf[4096];
        if (fgets(buf, sizeof buf, f)) list = buf;
        fclose(f);
    }
    cpu_set_t now, want;
    CPU_ZERO(&want);
    if (sched_getaffinity(0, sizeof now, &now) != 0) return;
    int n_want = 0;
    for (size_t i = 0; i < list.size();) {   // "0-63,128-191"
        char *end = nullptr;
        const long a = strtol(list.c_str() + i, &end, 10);
        if (end == list.c_str() + i) break;
        long b = a;
        i = (size_t)(end - list.c_str());
        if (i < list.size() && list[i] == '-') {
            b = strtol(list.c_str() + i + 1, &end, 10);
            i = (size_t)(end - list.c_str());
        }
        for (long c = a; c <= b && c < CPU_SETSIZE; ++c)
            if (CPU_ISSET((int)c, &now)) { CPU_SET((int)c, &want); ++n_want; }
        if (i < list.size() && list[i] == ',') ++i;
        else break;
    }
    if (n_want >= 4 && n_want < CPU_COUNT(&now)) (void)sched_setaffinity(0, sizeof want, &want);   // never widen, never starve
    if (getenv("PF_TRACE_LOAD")) fprintf(stderr, "[load] GPU %s on NUMA node %d: host threads on %d of %d CPUs\n", bdf, node, n_want, CPU_COUNT(&now));
}

CountsLoader::~CountsLoader() {
    wait();
    if (ctx) pf_destroy(ctx);
}

void CountsLoader::start(int device, const std::string &kmc_prefix) {
    th_ = std::thread([this, device, kmc_prefix] {
        LoadTrace trace;
        status = pf_create(device, &ctx);
        {
            std::lock_guard<std::mutex> lk(mu_);
            ctx_known_ = true;
        }
        cv_.notify_all();
        if (status != PF_OK) { error = std::string("CDBG::CDBG():Error: ") + pf_last_error(nullptr); return; }
        trace.mark("device: context (beside the graph read)");
        KmcRecords db;
        std::string e;
        if (!db.load(kmc_prefix, e)) { status = PF_ERR_ARG; error = "CDBG::CDBG():Error: Open kmc database error . (" + e + ")"; return; }
        both_strands = db.both_strands;
        k = (int)db.k;
        uint64_t *dk = nullptr;
        uint32_t *dc = nullptr;
        status = pf_kmc_decode(ctx, db.records, db.total, db.suffix_bytes, db.counter_size, db.lut.data(), db.n_lut(), db.lut_prefix_len, db.k, &dk, &dc);
        if (status == PF_OK) status = pf_upload_counts(ctx, dk, dc, db.total, db.k, db.min_count, db.max_count, db.both_strands);
        pf_device_free(ctx, dk);
        pf_device_free(ctx, dc);
        if (status != PF_OK) error = std::string("CDBG::CDBG():Error: ") + pf_last_error(ctx);
        trace.mark("kmc: device decode + table (beside the graph read)");
    });
}

int CDBG::init_device(int device, pf_ctx *adopt, bool colored) {
    LoadTrace trace;
    int st = PF_OK;
    if (adopt) ctx_ = adopt;
    else st = pf_create(device, &ctx_);
    trace.mark("device: context");
    if (st != PF_OK) return fail(st, std::string(tag_) + "::" + tag_ + "():Error: " + pf_last_error(nullptr));
    bind_to_device_node(ctx_);
    out_maps_.reset(new MappedOut[PF_CALL_STREAMS + 1]);   // ten streams of PloidyEstimation + super_bubble.txt
    if (!getenv("PF_NO_PREALLOC")) {
        const uint64_t n_est = g_.ingest_pending() ? g_.estimated_unitigs() : g_.n();
        prealloc_ = std::thread([this, n_est, colored] {
            bx_.bfs_rec.ensure(ctx_, n_est * 7 / 10 + 4096);            // ~0.67 candidates per unitig
            bx_.bfs_pool.ensure(ctx_, n_est * 7 / 10 * 6 + (5u << 20));
            sb_text_.ensure(ctx_, n_est * 19 + 4096);                      // ~0.63 rows of ~29 bytes per unitig
            if (!colored) {   // text slabs of the resident calling pipeline
                cx_.slab[0].ensure(ctx_, 64u << 20);
                cx_.slab[1].ensure(ctx_, 64u << 20);
            }
        });
    }
    if (!getenv("PF_NO_PREALLOC") && adopt && resident_ && !colored) {
        // the device buffers of the first alignment launch and of the first text piece, K-BUBBLE's and K-TEXT's first launches on
        // their streams: 0.13 s at 5 M unitigs, on a helper thread beside the rest of the load (graph upload, numbering,
        // adjacency: 0.2 s) -- they need the context and nothing that is loaded (pf_call_reserve in ploidyfrost_hip.h).  Only when
        // the count table came with the context (the CLI: it was built while the graph file was read): a constructor that still
        // has the database to decode and the table to build keeps the runtime busy from this thread, two threads inside the runtime
        // mostly wait for one another, and start_prealloc() at its end does the same beside the caller's next steps.  About a third
        // of the unitigs end up as called bubbles, aligned in two ranges of whole text pieces.
        const uint64_t n_est = g_.ingest_pending() ? g_.estimated_unitigs() : g_.n();
        const uint64_t piece = (uint64_t)std::max<size_t>(batch_bubbles_, 1) * 4;
        const uint64_t est = std::min<uint64_t>(1u << 24, ((uint64_t)(0.17 * (double)n_est) + piece - 1) / piece * piece);
        prealloc_align_ = std::thread([this, est, piece] {
            LoadTrace trace;
            (void)pf_call_reserve(ctx_, est, (uint32_t)complex_size_);
            trace.mark("reserve: alignment buffers of two lanes (beside the load)");
            (void)pf_call_reserve_text(ctx_, piece);
            trace.mark("reserve: text buffers (beside the load)");
        });
    }
    if (g_.ingest_pending()) {
        // K-GFA: the S-lines are parsed and packed on the device; the host keeps the segment table
        std::string e;
        st = g_.ingest_on_device(ctx_, e);
        if (st != PF_OK) return fail(st, "CompactedDBG::read(): Graph could not be loaded! Exit. (" + e + ")");
    } else {
        st = pf_upload_graph(ctx_, g_.words.data(), g_.word_off.data(), g_.len_bp.data(), g_.n(), g_.k);
        if (st != PF_OK) return fail(st, std::string(tag_) + "::" + tag_ + "():Error: graph upload: " + pf_last_error(ctx_));
    }
    trace.mark("device: graph upload");
    if (!getenv("PF_NO_PREALLOC") && !colored) {
        // the walkers of the long traversals keep 4 bytes of state per unitig each: zero-filled here, beside the rest of the load,
        // instead of inside the first findSuperBubble (23 ms at 5 M unitigs)
        const uint32_t n_now = g_.n();
        prealloc_walkers_ = std::thread([this, n_now] {
            const unsigned want = std::min(32u, std::max(1u, std::thread::hardware_concurrency() / 4));
            std::vector<std::unique_ptr<HugeWalker>> made(want);
            parallel_chunks(want, 1, want, [&](size_t i, size_t, size_t) {
                made[i] = std::make_unique<HugeWalker>();
                made[i]->info.reserve(n_now);
                advise_huge_pages(made[i]->info.data(), (size_t)n_now * 4);
                made[i]->info.assign(n_now, 0);
                made[i]->seen.reserve(1u << 18);
                made[i]->todo.reserve(1u << 12);
            });
            std::lock_guard<std::mutex> lk(walkers_mu_);
            for (auto &w : made) walkers_.push_back(std::move(w));
        });
    }
    if (g_.numbering_deferred) {
        // unitig numbering, last part: K-MINZ bounds the fill of every minimizer bucket; only a graph that can crowd one
        // (15 entries, bifrost/src/CompactedDBG.tcc:4013) needs the host replay of Bifrost's bookkeeping and a second upload
        uint32_t most = 0;
        if (g_.n_short && g_.g >= 1 && g_.g <= g_.k - 2 && g_.g <= 31) {
            st = pf_minimizer_crowding(ctx_, g_.g, 15, &most, nullptr, nullptr);
            if (st != PF_OK) return fail(st, std::string(tag_) + "::" + tag_ + "():Error: minimizer census: " + pf_last_error(ctx_));
        }
        if (most >= 15) {
            // (the census table and the unitigs that meet a crowded slot come from the device: the host replay's own two passes over
            // every unitig were 2.5 of a 3.9 s CLI run at BASELINE.json configs[4])
            std::vector<uint8_t> cnt8, flg;
            {
                cnt8.resize(pf_minimizer_table_slots(g_.n_kmers));
                flg.resize(g_.n());
                st = pf_minimizer_replay_inputs(ctx_, g_.g, 15, cnt8.data(), flg.data());
                if (st != PF_OK) return fail(st, std::string(tag_) + "::" + tag_ + "():Error: minimizer census: " + pf_last_error(ctx_));
            }
            g_.finish_numbering(nullptr, cnt8.empty() ? nullptr : cnt8.data(), flg.empty() ? nullptr : flg.data(), cnt8.size());
            if (g_.n_abundant) {
                st = pf_upload_graph(ctx_, g_.words.data(), g_.word_off.data(), g_.len_bp.data(), g_.n(), g_.k);
                if (st != PF_OK) return fail(st, std::string(tag_) + "::" + tag_ + "():Error: graph upload: " + pf_last_error(ctx_));
            }
        } else {
            g_.numbering_settled();
        }
        trace.mark(most >= 15 ? "device: numbering (replayed)" : "device: numbering (K-MINZ)");
    }
    const uint32_t N = g_.n();
    // (the host walkers chase these rows at random: 2 MB pages where the kernel hands them out on advice, asked for before the
    // first touch)
    succ_.reserve((size_t)N * 8);
    pred_.reserve((size_t)N * 8);
    advise_huge_pages(succ_.data(), (size_t)N * 8 * 4);
    advise_huge_pages(pred_.data(), (size_t)N * 8 * 4);
    succ_.resize((size_t)N * 8);
    pred_.resize((size_t)N * 8);
    st = pf_build_adjacency(ctx_, succ_.data(), pred_.data());
    if (st != PF_OK) return fail(st, std::string(tag_) + "::" + tag_ + "():Error: adjacency: " + pf_last_error(ctx_));
    trace.mark("device: adjacency");
    st_.reset(N);
    st_.complex_size = complex_size_;
    st_.g = &g_;
    st_.succ = succ_.data();
    return 0;
}

void CDBG::start_prealloc() {
    if (!getenv("PF_NO_PREALLOC") && resident_ && ctx_ && status_ == PF_OK) {
        // findSuperBubble's device buffers, which are sized by the number of candidates (the adjacency must be resident), taken
        // beside what follows the load on the host (the unitig id file); findSuperBubble waits for this thread.  The calling
        // pipeline's buffers are taken here as well when init_device's helper did not take them beside the load (colored graphs:
        // K-TEXT's colored form is chosen by pf_call_set_colours).
        const uint32_t N = g_.n();
        const uint64_t piece = (uint64_t)std::max<size_t>(batch_bubbles_, 1) * 4;
        const uint64_t est = std::min<uint64_t>(1u << 24, ((uint64_t)(0.17 * (double)N) + piece - 1) / piece * piece);
        const bool call_buffers = !prealloc_align_.joinable();
        prealloc_call_ = std::thread([this, est, piece, N, call_buffers] {
            LoadTrace trace;
            uint64_t n_cand = 0;   // findSuperBubble's buffers first: it is the first to run
            if (commits_on_device(1) && pf_count_candidates(ctx_, 0, N, &n_cand) == PF_OK) {
                trace.mark("reserve: candidates counted");
                (void)pf_find_reserve(ctx_, n_cand);
                trace.mark("reserve: findSuperBubble's buffers");
                // (K-CC once over no records: what its first call of a run pays beyond its 0.6 ms is paid here)
                (void)pf_side_components(ctx_, 1, nullptr, 0, nullptr, 0, nullptr, 0, nullptr, 0);
                trace.mark("reserve: K-CC's first call");
            }
            if (!call_buffers) return;
            (void)pf_call_reserve(ctx_, est, (uint32_t)complex_size_);
            trace.mark("reserve: alignment buffers of two lanes");
            (void)pf_call_reserve_text(ctx_, piece);
            trace.mark("reserve: text buffers");
        });
    }
}

CDBG::CDBG(UnitigSet &graph, const size_t &complexsize, double &m, double &d, double &g, std::string kmc_db, int device,
           bool quiet, CountsLoader *counts)
    : g_(graph), complex_size_(complexsize), quiet_(quiet) {
    sc_.match = m;
    sc_.mismatch = d;
    sc_.gap = g;
    if (const char *e = getenv("PF_CALL")) resident_ = strcmp(e, "host") != 0;  // measurements: the host-threaded pipeline
    if (counts) {
        // the count table was built while the graph file was read: adopt its context, then the graph goes in (the join of graph
        // k-mers with the table runs as soon as both are there, whichever comes second)
        counts->wait();
        if (counts->status != PF_OK) { fail(counts->status, counts->error); return; }
        if (counts->k != g_.k) { fail(PF_ERR_ARG, "CDBG::CDBG():Error: k of the kmc database differs from the graph's"); return; }
        both_strands_ = counts->both_strands;
        if (init_device(device, counts->release())) return;
        start_prealloc();
        if (!quiet_) printf("CDBG::CDBG():CDBG initialized!\n");
        return;
    }
    if (init_device(device)) return;
    if (!kmc_db.empty()) {
        LoadTrace trace;
        KmcRecords db;
        std::string e;
        if (!db.load(kmc_db, e)) { fail(PF_ERR_ARG, "CDBG::CDBG():Error: Open kmc database error . (" + e + ")"); return; }
        if ((int)db.k != g_.k) { fail(PF_ERR_ARG, "CDBG::CDBG():Error: k of the kmc database differs from the graph's"); return; }
        trace.mark("kmc: map + prefix table");
        both_strands_ = db.both_strands;
        // the records are decoded on the device (K-KMC) straight from the mapped file, then hashed into the count table
        uint64_t *dk = nullptr;
        uint32_t *dc = nullptr;
        int st = pf_kmc_decode(ctx_, db.records, db.total, db.suffix_bytes, db.counter_size, db.lut.data(), db.n_lut(), db.lut_prefix_len, db.k, &dk, &dc);
        if (st == PF_OK) st = pf_upload_counts(ctx_, dk, dc, db.total, db.k, db.min_count, db.max_count, db.both_strands);
        pf_device_free(ctx_, dk);
        pf_device_free(ctx_, dc);
        trace.mark("kmc: device decode + table");
        if (st != PF_OK) { fail(st, std::string("CDBG::CDBG():Error: ") + pf_last_error(ctx_)); return; }
    }
    start_prealloc();
    if (!quiet_) printf("CDBG::CDBG():CDBG initialized!\n");
}

CDBG::CDBG(UnitigSet &graph, const size_t &complexsize, double &m, double &d, double &g, int device, bool quiet, NoCounts)
    : g_(graph), complex_size_(complexsize), quiet_(quiet) {
    sc_.match = m;
    sc_.mismatch = d;
    sc_.gap = g;
    tag_ = "CCDBG";
    init_device(device, nullptr, true);
}

int write_span_parallel(int fd, uint64_t file_off, const char *src, uint64_t len, unsigned threads) {
    if (len == 0) return 0;
    struct stat sb;
    if (fstat(fd, &sb) != 0) return 1;
    if ((uint64_t)sb.st_size < file_off + len && ftruncate(fd, (off_t)(file_off + len)) != 0) return 1;
    const uint64_t page = (uint64_t)sysconf(_SC_PAGESIZE);
    const uint64_t map_off = file_off & ~(page - 1), lead = file_off - map_off;
    void *m = mmap(nullptr, (size_t)(len + lead), PROT_READ | PROT_WRITE, MAP_SHARED, fd, (off_t)map_off);
    if (m == MAP_FAILED) {
        // (a file system without shared writable mappings: plain positioned writes)
        uint64_t at = 0;
        while (at < len) {
            const ssize_t w = pwrite(fd, src + at, len - at, (off_t)(file_off + at));
            if (w <= 0) return 1;
            at += (uint64_t)w;
        }
        return 0;
    }
    char *dst = static_cast<char *>(m) + lead;
    constexpr uint64_t PIECE = 1u << 20;
    parallel_chunks((size_t)((len + PIECE - 1) / PIECE), 1, threads, [&](size_t i, size_t, size_t) {
        const uint64_t at = (uint64_t)i * PIECE;
        memcpy(dst + at, src + at, (size_t)std::min<uint64_t>(PIECE, len - at));
    });
    return munmap(m, (size_t)(len + lead)) != 0;
}

void MappedOut::close_file() {
    std::lock_guard<std::mutex> lk(remap_mu);
    if (base) munmap(base, map_len);
    if (fd >= 0) close(fd);
    base = nullptr;
    map_len = 0;
    backed_lo = backed_hi = 0;
    finished_once = false;
    no_fallocate = false;
    fd = -1;
    path.clear();
}

int MappedOut::open_for(const std::string &p) {
    if (fd >= 0 && p == path) {
        struct stat a, b;
        if (fstat(fd, &a) == 0 && stat(p.c_str(), &b) == 0 && a.st_ino == b.st_ino && a.st_dev == b.st_dev && a.st_nlink > 0) return 0;
    }
    close_file();
    fd = open(p.c_str(), O_RDWR | O_CREAT, 0666);
    if (fd < 0) return 1;
    path = p;
    return 0;
}

// Blocks for [off, off + len): posix_fallocate extends the file and reserves what it covers -- ENOSPC / EDQUOT come back here as
// a return value instead of as a SIGBUS under a memcpy into the mapping.  Ranges already backed cost nothing (a pass over a
// resident graph rewrites the same ranges every time).
bool MappedOut::back(uint64_t off, uint64_t len) {
    if (len == 0) return true;
    if (off >= backed_lo && off + len <= backed_hi) return true;
    // fallocate(2) itself, not posix_fallocate: where the file system cannot do it (EOPNOTSUPP: some NFS / FUSE mounts) glibc's
    // stand-in writes a byte per block -- beside a writer that stores through the mapping that can zero a byte just stored (advisor,
    // round 3).  Such a file is only extended (ftruncate) and a full disk shows when the pages are written back.
    if (!no_fallocate && fallocate(fd, 0, (off_t)off, (off_t)len) != 0) {
        if (errno != EOPNOTSUPP && errno != ENOSYS && errno != EINVAL) return false;
        no_fallocate = true;
    }
    if (no_fallocate) {
        struct stat sb;
        if (fstat(fd, &sb) != 0) return false;
        if ((uint64_t)sb.st_size < off + len && ftruncate(fd, (off_t)(off + len)) != 0) return false;
    }
    if (backed_hi > backed_lo && off <= backed_hi && off + len >= backed_lo) {   // touches what is known: one interval
        backed_lo = std::min(backed_lo, off);
        backed_hi = std::max(backed_hi, off + len);
    } else {
        backed_lo = off;
        backed_hi = off + len;
    }
    return true;
}

bool MappedOut::map_at_least(uint64_t bytes) {
    if (bytes <= map_len) return true;
    std::lock_guard<std::mutex> lk(remap_mu);
    if (base) munmap(base, map_len);
    base = nullptr;
    const size_t want = (size_t)(bytes + bytes / 4 + (8u << 20));   // (mapping past the end of the file is fine: never touched)
    void *m = mmap(nullptr, want, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    if (m == MAP_FAILED) { map_len = 0; return false; }
    base = static_cast<char *>(m);
    map_len = want;
    return true;
}

int MappedOut::write(uint64_t off, const char *src, uint64_t len, unsigned threads) {
    if (len == 0) return 0;
    if (fd < 0) return 1;
    if (!back(off, len)) return 1;
    if (!map_at_least(off + len)) return write_span_parallel(fd, off, src, len, threads);
    char *dst = base + off;
    constexpr uint64_t PIECE = 1u << 20;
    parallel_chunks((size_t)((len + PIECE - 1) / PIECE), 1, threads, [&](size_t i, size_t, size_t) {
        const uint64_t at = (uint64_t)i * PIECE;
        memcpy(dst + at, src + at, (size_t)std::min<uint64_t>(PIECE, len - at));
    });
    return 0;
}

char *MappedOut::prepare(uint64_t off, uint64_t len) {
    if (fd < 0) return nullptr;
    if (!back(off, len)) return nullptr;   // (the caller falls back to write(), which reports the error)
    if (!map_at_least(off + len)) return nullptr;
    return base + off;
}

void copy_spans(const CopySpan *spans, size_t n, unsigned threads) {
    constexpr uint64_t PIECE = 1u << 20;
    std::vector<CopySpan> cut;
    for (size_t i = 0; i < n; ++i)
        for (uint64_t at = 0; at < spans[i].len; at += PIECE)
            cut.push_back(CopySpan{spans[i].dst + at, spans[i].src + at, std::min<uint64_t>(PIECE, spans[i].len - at)});
    parallel_chunks(cut.size(), 1, threads, [&](size_t i, size_t, size_t) { memcpy(cut[i].dst, cut[i].src, (size_t)cut[i].len); });
}

int MappedOut::reserve(uint64_t bytes) {
    if (fd < 0) return 1;
    struct stat sb;
    if (fstat(fd, &sb) != 0) return 1;
    if (finished_once || (uint64_t)sb.st_size >= bytes / 2) return 1;   // (it has been written before: its pages exist)
    // sized now, backed piece by piece: by populate() on the helper threads, or by prepare() where the writer gets there first
    if (ftruncate(fd, (off_t)bytes) != 0) return 1;
    return map_at_least(bytes) ? 0 : 1;
}

// [from, to) of a reserved file: its blocks (posix_fallocate: a full disk shows here and leaves the range to prepare(), which
// reports it), then its page-table entries.  Holds the mapping in place while it works.
void MappedOut::populate(uint64_t from, uint64_t to) {
    if (to <= from || fd < 0) return;
    if (no_fallocate) return;   // (no early prefault for a file whose blocks cannot be reserved)
    if (fallocate(fd, 0, (off_t)from, (off_t)(to - from)) != 0) {
        if (errno == EOPNOTSUPP || errno == ENOSYS || errno == EINVAL) no_fallocate = true;
        return;
    }
#ifdef MADV_POPULATE_WRITE
    std::lock_guard<std::mutex> lk(remap_mu);
    if (base && to <= map_len) (void)madvise(base + (from & ~4095ull), (size_t)(to - (from & ~4095ull)), MADV_POPULATE_WRITE);
#endif
}

int MappedOut::finish(uint64_t final_len) {
    if (fd < 0) return 1;
    struct stat sb;
    if (fstat(fd, &sb) != 0) return 1;
    if ((uint64_t)sb.st_size != final_len) {
        if (getenv("PF_TRACE_PLOIDY")) fprintf(stderr, "[ploidy]   %s cut from %llu to %llu bytes\n", path.c_str(), (unsigned long long)sb.st_size, (unsigned long long)final_len);
        if (ftruncate(fd, (off_t)final_len) != 0) return 1;
    }
    // what lay beyond is gone (whoever cut it: every rank of a sharded run passes the same length here before it stores anything)
    backed_hi = std::min(backed_hi, final_len);
    if (backed_hi <= backed_lo) backed_lo = backed_hi = 0;
    finished_once = true;   // a later pass over the same graph writes about as much: no estimate sizes the file again
    return 0;
}

int CDBG::set_reference_threads(size_t n) {
    if (n > 1 && col_) return fail(PF_ERR_ARG, "CCDBG:: the -t > 1 output format is built for the single-sample path only");
    mt_format_ = n > 1;
    const int st = pf_call_set_format(ctx_, mt_format_ ? 1 : 0);
    return st == PF_OK ? 0 : fail(st, pf_last_error(ctx_));
}

int CDBG::join_pending_ids() {
    if (pending_ids_.joinable()) {
        pending_ids_.join();
        if (pending_ids_rc_) { pending_ids_rc_ = 0; return fail(PF_ERR_ARG, "CDBG:: Open Unitig_Id file error"); }
    }
    return 0;
}

int CDBG::join_pending_write() {
    if (pending_write_.joinable()) {
        pending_write_.join();
        if (pending_rc_) { pending_rc_ = 0; return fail(PF_ERR_ARG, "CDBG:: Open super_bubble file error"); }
    }
    return 0;
}

CDBG::~CDBG() {
    join_prealloc();
    join_pending_ids();
    join_pending_write();
    bx_.release_all();  // pinned buffers go before the context
    for (auto &a : ax_) a.release_all();
    cx_.release_all();
    sb_text_.release();
    pf_destroy(ctx_);
}

int CDBG::fail(int st, const std::string &msg) {
    status_ = st ? st : PF_ERR_ARG;
    err_ = msg;
    return status_;
}

int CDBG::ensure_dir() {
    struct stat sb;
    if (stat(outdir_.c_str(), &sb) == 0) return 0;
    if (mkdir(outdir_.c_str(), 0777) != 0 && access(outdir_.c_str(), 0)) return fail(PF_ERR_ARG, "cannot create " + outdir_);
    return 0;
}

int CDBG::write_file(const std::string &name, const std::string &data) {
    out_bytes_ += data.size();
    if (!write_files_) return 0;
    FILE *f = fopen((outdir_ + "/" + name).c_str(), "wb");
    if (!f) return fail(PF_ERR_ARG, "CDBG:: Open " + name + " file error");
    if (!data.empty() && fwrite(data.data(), 1, data.size(), f) != data.size()) {
        fclose(f);
        return fail(PF_ERR_ARG, "CDBG:: write error on " + name);
    }
    fclose(f);
    return 0;
}

// thread-safe: touches no member state (the caller accounts the bytes)
int CDBG::write_pieces(const std::string &name, const std::vector<const std::string *> &pieces, uint64_t &bytes) const {
    bytes = 0;
    for (const std::string *p : pieces) bytes += p->size();
    if (!write_files_) return 0;
    FILE *f = fopen((outdir_ + "/" + name).c_str(), "wb");
    if (!f) return 1;
    int rc = 0;
    for (const std::string *p : pieces)
        if (!p->empty() && fwrite(p->data(), 1, p->size(), f) != p->size()) { rc = 1; break; }
    fclose(f);
    return rc;
}

// Several files, each the concatenation of its pieces; one writer thread per file (concurrent
// writers into one tmpfs/ext4 inode only contend on its lock).
int CDBG::write_many(const std::vector<std::pair<std::string, std::vector<const std::string *>>> &files, unsigned threads) {
    std::vector<int> rc(files.size(), 0);
    std::vector<uint64_t> bytes(files.size(), 0);
    parallel_chunks(files.size(), 1, threads, [&](size_t i, size_t, size_t) { rc[i] = write_pieces(files[i].first, files[i].second, bytes[i]); });
    for (size_t i = 0; i < files.size(); ++i) {
        out_bytes_ += bytes[i];
        if (rc[i]) return fail(PF_ERR_ARG, "CDBG:: Open " + files[i].first + " file error");
    }
    return 0;
}

// ---- setUnitigId (reference src/CDBG.cpp:121-143) -----------------------------------------
int CDBG::setUnitigId(const std::string &outpre, const std::string &, const size_t &) {
    if (status_) return status_;
    if (write_files_ && ensure_dir()) return status_;
    if (!quiet_) printf("%s::setUnitigId(): Setting Unitig Id\n", tag_);
    clock_t c0 = clock();
    time_t w0 = time(nullptr);
    // `id<TAB>sequence` rows: every unitig range knows where its rows start (digits of the ids + lengths), formats them and writes
    // them at that offset, ranges side by side.  With overlap_output the whole file is written behind the caller's back (it depends
    // on nothing the phases compute) and is complete when PloidyEstimation, the next setUnitigId or the destructor returns.
    const unsigned T = threads_ ? threads_ : std::min(32u, std::max(1u, std::thread::hardware_concurrency()));
    if (join_pending_ids()) return status_;
    auto job = [this, name = outpre + "_Unitig_Id.txt", T]() -> int {
        const bool trace_ids = getenv("PF_TRACE_PLOIDY") != nullptr;
        const auto t_job = clk::now();
        auto ti = [&](const char *what) { if (trace_ids) fprintf(stderr, "[unitig ids] %-28s %.2f ms\n", what, since(t_job) * 1e3); };
        // (after a device ingest the sequences are still in the mapped file only: the rows are made from there)
        constexpr size_t UCH = 32768;
        const size_t N = g_.n(), n_ch = n_chunks_of(N, UCH);
        std::vector<uint64_t> base(n_ch + 1, 0);
        parallel_chunks(N, UCH, T, [&](size_t ci, size_t u0, size_t u1) {
            uint64_t b = (uint64_t)(g_.off[u1] - g_.off[u0]) + 2 * (u1 - u0);
            for (size_t u = u0; u < u1; ++u) {
                uint64_t id = u + 1;
                do { ++b; id /= 10; } while (id);
            }
            base[ci + 1] = b;
        });
        for (size_t c = 0; c < n_ch; ++c) base[c + 1] += base[c];
        ids_bytes_ = base[n_ch];
        ti("row offsets");
        if (!write_files_) return 0;
        const int fd = open((outdir_ + "/" + name).c_str(), O_WRONLY | O_CREAT, 0666);
        if (fd < 0) return 1;
        std::vector<int> rc(n_ch, 0);
        parallel_chunks(N, UCH, T, [&](size_t ci, size_t u0, size_t u1) {
            std::string out;
            out.reserve((size_t)(base[ci + 1] - base[ci]));
            for (size_t u = u0; u < u1; ++u) {
                put_uint(out, u + 1);
                out.push_back('\t');
                const size_t at0 = out.size(), L = g_.size_bp((uint32_t)u);
                out.resize(at0 + L);
                g_.copy_seq((uint32_t)u, &out[at0]);
                out.push_back('\n');
            }
            uint64_t at = base[ci], left = out.size();
            const char *src = out.data();
            while (left) {
                const ssize_t w = pwrite(fd, src, left, (off_t)at);
                if (w <= 0) { rc[ci] = 1; return; }
                left -= (uint64_t)w;
                src += w;
                at += (uint64_t)w;
            }
        });
        ti("rows written");
        int bad = ftruncate(fd, (off_t)base[n_ch]) != 0;
        close(fd);
        for (int x : rc) bad |= x;
        return bad;
    };
    if (overlap_output_) {
        pending_ids_ = std::thread([this, job] { pending_ids_rc_ = job(); });
    } else {
        if (job()) return fail(PF_ERR_ARG, "CDBG:: Open " + outpre + "_Unitig_Id.txt file error");
    }
    if (!quiet_) {
        printf("%s::setUnitigId(): Cpu time : %gs\n", tag_, (double)(clock() - c0) / CLOCKS_PER_SEC);
        printf("%s::setUnitigId(): Real time : %gs\n", tag_, difftime(time(nullptr), w0));
    }
    return 0;
}

// ---- printInfo (reference src/CDBG.cpp:144-162): <outpre>_graph_info.txt in the CWD --------
int CDBG::printInfo(const bool &verbose, const std::string &outpre) {
    if (status_) return status_;
    uint64_t length = 0;
    for (uint32_t u = 0; u < g_.n(); ++u) length += g_.size_bp(u);
    char line[256];
    snprintf(line, sizeof line, "k:%d\tg:%d\tnbKmer:%llu\tnbUnitig:%u\tlength:%llu\t", g_.k, g_.g,
             (unsigned long long)g_.n_kmers, g_.n(), (unsigned long long)length);
    if (verbose) printf(">>>>>>>>>Bifrost Graph Information>>>>>>>>>\n%s\n", line);
    FILE *f = fopen((outpre + "_graph_info.txt").c_str(), "w");
    if (f) { fputs(line, f); fclose(f); }
    return 0;
}

}  // namespace pfh
