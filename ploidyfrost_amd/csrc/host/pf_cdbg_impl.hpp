// Private to the host layer: what the translation units of pfh::CDBG / pfh::CCDBG share (pf_cdbg.cpp: construction,
// output plumbing, MyUnitig state and the commit replay; pf_cdbg_find.cpp: findSuperBubble; pf_cdbg_ploidy.cpp:
// PloidyEstimation; pf_ccdbg.cpp: the colored front end).
#pragma once
#include <mutex>
#include <charconv>
#include <chrono>
#include <cstdint>
#include <string>

#include "pf_cdbg.hpp"

namespace pfh {

namespace {
using clk = std::chrono::steady_clock;
inline double since(clk::time_point t0) { return std::chrono::duration<double>(clk::now() - t0).count(); }

// MyUnitig::b bit layout (reference src/MyUnitig.hpp:37-46, 52-85, 97-130)
constexpr uint8_t B_PLUS = 0x01, B_MINUS = 0x02, B_NON_SUPER = 0x04, B_STRICT_M = 0x08, B_STRICT_P = 0x10,
                  B_COMPLEX_M = 0x20, B_COMPLEX_P = 0x40;
constexpr uint32_t NONE = 0xFFFFFFFFu;

inline bool plus_of(uint32_t ov) { return (ov & 1) == 0; }

// `ostream << double` with default flags == printf("%g") (precision 6)
// (std::to_chars with chars_format::general and a precision is specified as that printf conversion)
inline void put_double(std::string &s, double x) {
    char buf[48];
    auto r = std::to_chars(buf, buf + sizeof buf, x, std::chars_format::general, 6);
    s.append(buf, (size_t)(r.ptr - buf));
}
inline void put_uint(std::string &s, uint64_t x) {
    char buf[24];
    auto r = std::to_chars(buf, buf + sizeof buf, x);
    s.append(buf, (size_t)(r.ptr - buf));
}
}  // namespace

// len bytes of src into the file at file_off, by `threads` threads side by side.  Concurrent write()s to ONE file serialise on
// the inode lock (tmpfs and ext4 alike), so a 400 MB result file written in pieces by 32 threads moves at the speed of one;
// stores into a shared mapping of the same range do not.  The file is extended first when it is too short.  0 = ok.
int write_span_parallel(int fd, uint64_t file_off, const char *src, uint64_t len, unsigned threads);

// A result file that stays open and mapped from pass to pass.  A service that runs pass after pass over a resident graph
// rewrites files of (nearly) the same size every time; with the mapping kept, a pass's text goes into pages whose table
// entries already exist -- plain memory copies by all threads -- instead of faulting every page in again.  The file is
// re-opened when the path changes or the file was replaced behind our back.
struct MappedOut {
    std::string path;
    int fd = -1;
    char *base = nullptr;
    size_t map_len = 0;
    // [backed_lo, backed_hi): bytes of the file whose blocks are known to exist (posix_fallocate by this object).  A store through
    // the mapping into a hole of a sparse file is the first to learn that the disk (or /dev/shm) is full -- as a SIGBUS that takes
    // the process, and the Python interpreter hosting this library, with it; so nothing is stored outside this interval.
    uint64_t backed_lo = 0, backed_hi = 0;
    bool finished_once = false;   // finish() has given this file a length: reserve() leaves it alone from then on
    bool no_fallocate = false;    // the file system cannot reserve blocks (EOPNOTSUPP): extended by ftruncate, no early prefault
    std::mutex remap_mu;   // populate() on a helper thread against a remap by prepare() / write() / reserve()
    MappedOut() = default;
    MappedOut(const MappedOut &) = delete;
    MappedOut &operator=(const MappedOut &) = delete;
    ~MappedOut() { close_file(); }
    int open_for(const std::string &p);                                            // 0 = ok
    int write(uint64_t off, const char *src, uint64_t len, unsigned threads);      // grows file and mapping as needed
    // the first half of write() alone: file and mapping hold [off, off + len) afterwards; where to copy to (nullptr: no mapping,
    // use write()).  copy_spans() below then copies the spans of several files in ONE dispatch of the thread pool -- a
    // dispatch per file costs more than the copy of a small stream.
    char *prepare(uint64_t off, uint64_t len);
    int finish(uint64_t final_len);                                                // the file's length after this pass
    // A file about to receive ~bytes for the first time: sized and mapped now (before anybody writes), its pages are then
    // faulted in by populate() on helper threads while the first pieces are still on their way -- a fresh page costs the kernel
    // more than the copy into it (57 MB took 10 ms through first-touch faults).  No-op for a file that already has its pages.
    int reserve(uint64_t bytes);
    void populate(uint64_t from, uint64_t to);
    void close_file();
private:
    bool back(uint64_t off, uint64_t len);   // blocks for [off, off + len) (extends the file); false: no space / not supported
    bool map_at_least(uint64_t bytes);
};

struct CopySpan {
    char *dst;
    const char *src;
    uint64_t len;
};
void copy_spans(const CopySpan *spans, size_t n, unsigned threads);

// one bubble to call, in output order
struct CDBG::Task {
    uint32_t u = 0;        // owner endpoint (unitig index)
    uint32_t entrance_ov = 0, exit_ov = 0;
    bool strict = false;
    double core_mean = 0;
    // strict: inner unitigs sorted by (mean coverage desc, reference string desc) and their means
    uint32_t inner[4] = {0, 0, 0, 0};
    double cov[4] = {0, 0, 0, 0};
    uint8_t n_inner = 0, n_cov = 0;
    double cov_sum = 0;
    // colored strict bubble: where its [colour][inner] coverage matrix sits (chunk << 32 | offset into the chunk's pool)
    uint64_t cov_ref = 0;
};


}  // namespace pfh
