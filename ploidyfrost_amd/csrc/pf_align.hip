// K-ALN placeholder (replaced by the real kernel).
#include "pf_ctx.hpp"
#include "ploidyfrost_hip.h"
extern "C" int pf_align_batch(pf_ctx *ctx, const char *, uint64_t, const pf_align_job *, uint32_t, double, double, double,
                              uint64_t *, pf_align_hit *, uint64_t, char *, uint64_t, uint32_t *, uint64_t, uint64_t[3]) {
    if (ctx) ctx->err = "pf_align_batch: not built yet";
    return PF_ERR_ARG;
}
