/*
 * ploidyfrost_host.h -- C facade over the C++ host layer (pfh::UnitigSet + pfh::CDBG,
 * ploidyfrost_amd/csrc/host/pf_cdbg.hpp), so that non-C++ callers (bench.py, the tests) can
 * drive exactly the call sequence of the reference's main() (src/Main.cpp:829-849):
 *
 *   pfh_open(gfa, kmc_prefix, z, M, D, G, device)      CompactedDBG::read + CDBG::CDBG
 *   pfh_set_unitig_id(h, outpre)                       CDBG::setUnitigId
 *   pfh_find_superbubbles(h, outpre)                   CDBG::findSuperBubble_multithread_ptr
 *   pfh_ploidy_estimation(h, outpre, lower, upper)     CDBG::ploidyEstimation_multithread_ptr
 *
 * Everything compute-heavy inside runs on the GPU through ploidyfrost_hip.h.
 */
#ifndef PLOIDYFROST_HOST_H_
#define PLOIDYFROST_HOST_H_
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct pfh_run pfh_run;

typedef struct pfh_times {
    double load_s, upload_s;
    double bfs_device_s, replay_s, bubble_write_s, find_total_s;
    double cov_device_s, tasks_s, align_s, sites_s, format_s, write_s, ploidy_total_s;
    uint64_t unitigs, kmers, candidates, superbubbles, tasks, align_jobs, site_strings, output_bytes;
    uint64_t allele[4], core_cov, core_num;
} pfh_times;

/* NULL on failure: message via pfh_last_error(NULL) */
pfh_run *pfh_open(const char *gfa_path, const char *kmc_prefix, uint32_t complex_size, double match, double mismatch,
                  double gap, int device);
void pfh_close(pfh_run *);
const char *pfh_last_error(const pfh_run *);
void pfh_set_output_dir(pfh_run *, const char *dir); /* default ./PloidyFrost_output */
void pfh_set_write_files(pfh_run *, int on);         /* 0: format everything, write nothing */
void pfh_set_threads(pfh_run *, uint32_t threads);   /* host threads for the per-bubble phases (the reference's -t);
                                                         output order is always the -t 1 one */
int pfh_set_unitig_id(pfh_run *, const char *outpre);
int pfh_find_superbubbles(pfh_run *, const char *outpre);
int pfh_ploidy_estimation(pfh_run *, const char *outpre, int lower, int upper);
void pfh_get_times(const pfh_run *, pfh_times *out);
/* the pf_ctx of include/ploidyfrost_hip.h that this run drives (timing, stream control) */
void *pfh_device_ctx(pfh_run *);
/* <outpre>_allele_frequency.txt of the last pfh_ploidy_estimation, in memory (valid until the next
 * run): the per-rank record slab that a multi-GPU job all-gathers. */
const char *pfh_last_allele_frequency(const pfh_run *, uint64_t *len);
/* per-unitig state after pfh_find_superbubbles (MyUnitig flag byte, partner ids; 0 = NULL) */
void pfh_state(const pfh_run *, uint8_t *flags, uint32_t *plus, uint32_t *minus);

#ifdef __cplusplus
}
#endif
#endif
