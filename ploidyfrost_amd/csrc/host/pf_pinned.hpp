// Grow-only pinned host buffer for the exchange with the device layer (falls back to pageable
// memory if pinning fails).  Contents are not preserved across ensure().
#pragma once
#include <new>
#include <cstddef>
#include <cstdlib>

#include "ploidyfrost_hip.h"

namespace pfh {

template <typename T>
struct PinnedBuf {
    pf_ctx *ctx = nullptr;
    T *p = nullptr;
    size_t cap = 0;
    bool pinned = false;
    PinnedBuf() = default;
    PinnedBuf(const PinnedBuf &) = delete;
    PinnedBuf &operator=(const PinnedBuf &) = delete;
    ~PinnedBuf() { release(); }
    void release() {
        if (!p) return;
        if (pinned) pf_host_free(ctx, p); else free(p);
        p = nullptr;
        cap = 0;
    }
    void ensure(pf_ctx *c, size_t n) {
        if (n <= cap) return;
        release();
        ctx = c;
        const size_t want = n + n / 4 + 64;
        void *q = nullptr;
        if (pf_host_alloc(c, want * sizeof(T), &q) == PF_OK) { p = (T *)q; pinned = true; }
        else {
            p = (T *)malloc(want * sizeof(T));
            pinned = false;
            if (!p) { cap = 0; throw std::bad_alloc(); }
        }
        cap = want;
    }
};

}  // namespace pfh
