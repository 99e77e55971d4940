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

#include "ploidyfrost_hip.h" /* pf_bfs_record */
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
    double scan_s, scan_serial_s; /* owner/order scan of PloidyEstimation: whole, and its sequential part */
    uint64_t bfs_large, bfs_max_seen; /* K-BFS traversals of more than 4096 vertices in the last findSuperBubble; the longest one */
    uint64_t bfs_deferred;            /* candidates that left the device's LDS tier (128 entries) for the host walkers */
    uint64_t snp_jobs, pair_jobs, wave_jobs; /* of align_jobs: bubbles finished by K-SNP, by K-PAIR, sent to K-BUBBLE (resident pipeline) */
    uint64_t stack_jobs;                     /* ... finished by K-STACK (paths of one length; the alignment certified to be the paths stacked) */
    uint64_t host_commit_records;            /* findSuperBubble: records committed on a host thread (large components, walked traversals) */
    uint64_t host_walk_vertices;             /* ... vertices the host walkers visited for the traversals the device gave up on */
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
void pfh_set_overlap_output(pfh_run *, int on);      /* 1: <outpre>_super_bubble.txt is written in the background and is complete
                                                         when pfh_ploidy_estimation (or pfh_close) returns; default 0 */
/* K-BFS traversals beyond 4096 vertices: on host cores (default, pf_bfs_candidates_split + host/pf_bfs_host.cpp: one thread per
 * traversal) or, with on = 0, on the device (k_bfs_huge: one wavefront per traversal).  Same records either way. */
void pfh_set_third_tier_on_host(pfh_run *, int on);
/* n > 1: write the text format of the reference's `-t n` functions (ids and var_count from 0, allele_frequency rows of a bubble
 * grouped by arity; src/CDBG.cpp:1829, 2056, 2158-2162, 2550) -- in the deterministic `-t 1` row order; n <= 1: the `-t 1` format.
 * Single-sample path. */
int pfh_set_reference_threads(pfh_run *, uint32_t n);
void pfh_set_batch_bubbles(pfh_run *, uint64_t n);
/* single-sample path: text pieces (4 x batch_bubbles bubbles, formatted / fetched / written one after the other) per alignment
 * launch; default 64 = the whole pass is aligned at once up to 2^24 bubbles.  Tests set small values to cross the boundaries. */
void pfh_set_align_pieces(pfh_run *, uint64_t n);   /* bubbles per batch of the align/format pipeline (default 65536) */
int pfh_set_unitig_id(pfh_run *, const char *outpre);
int pfh_find_superbubbles(pfh_run *, const char *outpre);
int pfh_ploidy_estimation(pfh_run *, const char *outpre, int lower, int upper);
void pfh_get_times(const pfh_run *, pfh_times *out);
/* Where the loads of this process spent their time (GFA map / parse / upload, count database, join, adjacency, numbering, ...):
 * "step\tseconds\n" per step since the last reset, in the order the steps ended (steps of helper threads overlap those of the
 * caller's).  Returns the length of the text; at most cap - 1 bytes and a NUL are written.  reset != 0 empties the log afterwards. */
uint64_t pfh_load_trace(char *out, uint64_t cap, int reset);
/* The row filters behind the path (reference script/Filter.R, script/Filter-multi.R; csrc/host/pf_filter.hpp; PARITY UNPINNED --
 * no R in the build image): argv as the scripts take it (argv[0], argv[1] are skipped like "ploidyfrost filter"), multi != 0 = the
 * colored tables.  pfh_r_format_double: one number as R's write.table renders it (15 significant digits, fixed or scientific by
 * width); returns the length, writes at most cap - 1 bytes and a NUL. */
int pfh_filter(int argc, char **argv, int multi);
uint64_t pfh_r_format_double(double x, char *out, uint64_t cap);
/* the pf_ctx of include/ploidyfrost_hip.h that this run drives (timing, stream control) */
void *pfh_device_ctx(pfh_run *);
/* <outpre>_allele_frequency.txt of the last pfh_ploidy_estimation, in memory (valid until the next
 * run): the per-rank record slab that a multi-GPU job all-gathers. */
const char *pfh_last_allele_frequency(const pfh_run *, uint64_t *len);
/* per-unitig state after pfh_find_superbubbles (MyUnitig flag byte, partner ids; 0 = NULL) */
void pfh_state(const pfh_run *, uint8_t *flags, uint32_t *plus, uint32_t *minus);

/* ---- colored (multi-sample) runs: the call sequence of src/Main.cpp:775-810 --------------------------
 *   pfh_open_colored(gfa, bfg_colors, kmc_list, ...)          ColoredCDBG::read + CCDBG::CCDBG
 *   pfh_set_unitig_id / pfh_find_superbubbles                 CCDBG::setUnitigId / findSuperBubble_multithread_ptr
 *   pfh_ploidy_estimation_colored(h, outpre, lower, upper, n) CCDBG::ploidyEstimation_multithread_ptr, one
 *                                                             (lower, upper) cutoff per colour (the -C file)
 * kmc_list_file: one KMC database prefix per line, one line per colour (src/CCDBG.cpp:13-43). */
pfh_run *pfh_open_colored(const char *gfa_path, const char *colors_path, const char *kmc_list_file, uint32_t complex_size,
                          double match, double mismatch, double gap, uint32_t threads, int device);
uint32_t pfh_num_colors(const pfh_run *);
int pfh_ploidy_estimation_colored(pfh_run *, const char *outpre, const int *lower, const int *upper, uint32_t n_colors);

/* ---- colour sets of a colored graph (no GPU involved) -----------------------------------------
 * The product's own reader of Bifrost's .bfg_colors (ploidyfrost_amd/csrc/host/pf_host_colors.hpp), which
 * replaces ColoredCDBG::read -> DataStorage::read -> UnitigColors::read (bifrost/src/ColoredCDBG.tcc:428,
 * DataStorage.tcc:790, ColorSet.cpp:1228) on the CCDBG path.  Exposed so that tests can hold it against
 * the real Bifrost's reading of the same file. */
typedef struct pfh_colors pfh_colors;
pfh_colors *pfh_colors_open(const char *gfa_path, const char *colors_path, uint32_t threads); /* NULL: pfh_last_error(NULL) */
void pfh_colors_close(pfh_colors *);
uint32_t pfh_colors_count(const pfh_colors *);                 /* ColoredCDBG::getNbColors */
uint32_t pfh_colors_unitigs(const pfh_colors *);
const char *pfh_colors_name(const pfh_colors *, uint32_t colour);
/* presence[colour * n_kmers + i] = 1 when k-mer i (reference orientation) of unitig u carries the colour
 * (UnitigColors::contains, ColorSet.cpp:776); returns UnitigColors::size(um) (ColorSet.cpp:898);
 * *n_full_enc = colours the file stores in the {full colours, rest} pair form, 0 otherwise */
uint64_t pfh_colors_unitig(const pfh_colors *, uint32_t u, uint8_t *presence, uint32_t *n_kmers, uint32_t *n_full_enc);
/* The unitig numbering is the reference's `-t 1` one: long unitigs in S-line order, then k-length unitigs in S-line
 * order, then the k-length unitigs Bifrost files as "abundant" k-mers (their minimizer's bucket already held 15 entries
 * when they were read, CompactedDBG.tcc:4013-4021, 4031-4068) in the slot order of its k-mer hash table
 * (KmerHashTable.hpp:326-354; UnitigIterator.tcc:32-58).
 * pfh_gfa_abundant_kmers: how many unitigs of the file end up in that last group (UINT64_MAX = unreadable file).
 * pfh_gfa_write_unitig_ids: writes `id<TAB>sequence` per unitig, the content of <prefix>_Unitig_Id.txt (src/CDBG.cpp:121-143),
 * from the GFA file alone (no device involved); 0 = ok.
 * pfh_gfa_numbering_replays: 0 when no minimizer bucket can reach 15 entries (nothing to decide), else how many times the
 * loader replayed Bifrost's bucket bookkeeping (> 1: long unitigs were redirected into buckets it had not followed). */
uint64_t pfh_gfa_abundant_kmers(const char *gfa_path);
uint32_t pfh_gfa_numbering_replays(const char *gfa_path);
/* the host pass's saturating minimizer-occurrence counters (what K-MINZ, pf_minimizer_crowding, bounds from above): returns
 * the number of slots (0 when the graph has no k-length unitig to decide about) and fills counters when slots suffices */
uint64_t pfh_gfa_minimizer_counts(const char *gfa_path, uint8_t *counters, uint64_t slots);
int pfh_gfa_write_unitig_ids(const char *gfa_path, const char *out_path);
/* [tests] the same file through the numbering replay with its inputs handed in (as pf_minimizer_replay_inputs hands them over): the host's
 * own counters + bump (saturating upper bounds), every unitig flagged */
int pfh_gfa_write_unitig_ids_given_inputs(const char *gfa_path, const char *out_path, int bump);
/* ---- `PloidyFrost model`: class GmmModel (src/GmmModel.hpp:5-49) and the driver of src/Main.cpp:636-692 ------------------
 * pfh_gmm_open needs no device; the readers are the reference's text parsers (readFreFile src/GmmModel.cpp:240-257,
 * readCovFile :21-239); pfh_gmm_fit = setMThreshold/setNThreshold/setMaxIterNum/setMaxDeltaNum + resize(gauss) +
 * emIterate() on the GPU (fails without one); pfh_gmm_run fits gauss = min..max and writes <outprefix>_model_result.txt
 * (Main.cpp:659-690).  Every call: 0 = ok, else pfh_gmm_last_error(). */
typedef struct pfh_gmm pfh_gmm;
pfh_gmm *pfh_gmm_open(int device);
void pfh_gmm_close(pfh_gmm *);
const char *pfh_gmm_last_error(const pfh_gmm *);
int pfh_gmm_read_fre(pfh_gmm *, const char *allele_frequency_file, double min_frequency);
int pfh_gmm_read_cov(pfh_gmm *, const char *coverage_file_prefix, double min_frequency);
int pfh_gmm_set_values(pfh_gmm *, const double *values, uint64_t n);   /* GmmModel::readData */
uint64_t pfh_gmm_size(const pfh_gmm *);
int pfh_gmm_values(const pfh_gmm *, double *out);
int pfh_gmm_fit(pfh_gmm *, uint32_t gauss, double m_thre, double n_thre, int32_t max_iter, double max_delta, double *weights,
                double *means, double *vars, double *loglik, double *aic, uint32_t *iterations);
int pfh_gmm_run(pfh_gmm *, int min_gauss, int max_gauss, double m_thre, double n_thre, int32_t max_iter, double max_delta,
                const char *outprefix);
/* enable = 1 / 0 switches the HIP-event timing of the K-GMM launches on / off; enable < 0 reads the totals */
int pfh_gmm_kernel_time(pfh_gmm *, int enable, double *total_ms, uint64_t *launches);

/* The host tier of K-BFS on its own (host/pf_bfs_host.hpp; no device involved): extractSuperBubble_ptr's traversal
 * (src/CDBG.cpp:253-372) from one oriented vertex over CSR rows laid out as pf_build_adjacency returns them.  Fills *record
 * (list_off = 0) and copies its list -- seen[] when an exit was found, the cycle set otherwise -- to `list`.
 * 0 = ok, 2 = list_cap too small (record->n_list says how much is needed). */
int pfh_host_walk(const uint32_t *succ, const uint32_t *pred, uint32_t n_unitigs, uint32_t entrance, pf_bfs_record *record, uint32_t *list,
                  uint64_t list_cap);

/* ---- one graph over several GPUs: every rank opens the same graph and database -------------------------------
 * findSuperBubble:   pfh_find_shard(u0, u1) traverses the candidate entrances on this rank's unitig range (K-BFS + host walkers);
 *                    pfh_shard_records / pfh_shard_pool expose the records (pf_bfs_record, list_off relative to the pool) for the
 *                    exchange (RCCL all-gather); pfh_find_replay applies the records of all shards in shard order -- the
 *                    reference's visiting order with its `partner == NULL` gate (src/CDBG.cpp:206-214) -- so that every rank ends
 *                    with the same MyUnitig state; write_file = 1 on the rank that writes <outpre>_super_bubble.txt.
 * PloidyEstimation:  pfh_ploidy_select (scan + sequential pass, on every rank) -> n_bubbles, the run's bubble list in output
 *                    order; the rank aligns its contiguous slice [t0, t1) (pfh_ploidy_align -> how many of them are called:
 *                    var_count, src/CDBG.cpp:1254-1258); after exchanging those counts it formats with the number of bubbles
 *                    called by the ranks before it (pfh_ploidy_text -> the byte sizes of its ten slabs and counters[8] =
 *                    {2,3,4,5-allele sites, coreCov, coreNum, called, bubbles}); after exchanging the sizes it writes its slabs at
 *                    its offsets of the shared result files (pfh_ploidy_write; truncate = 1 cuts them to `totals`: pass it on
 *                    every rank, the length is the same).  Rank-order concatenation = the reference's `-t 1` files, because a
 *                    bubble's position is fixed by its owner endpoint (:1190, 1352, 1656-1679). */
int pfh_find_shard(pfh_run *, uint32_t u0, uint32_t u1);
const pf_bfs_record *pfh_shard_records(const pfh_run *, uint64_t *n_records);
const uint32_t *pfh_shard_pool(const pfh_run *, uint64_t *n_entries);
/* pool_lens (entries of each pool; NULL = sequential replay): the commits run on host threads, component by component
 * (pfh_set_replay_threads; csrc/host/pf_replay_par.hpp), the components found on the device -- from dev_records / dev_pools when the
 * shards already lie in device memory (the all-gather's output), else from an upload of records / pools. */
int pfh_find_replay(pfh_run *, const char *outpre, uint32_t n_shards, const pf_bfs_record *const *records, const uint64_t *n_records,
                    const uint32_t *const *pools, int write_file, const uint64_t *pool_lens, const pf_bfs_record *const *dev_records,
                    const uint32_t *const *dev_pools);
/* where the commits of findSuperBubble run: -1 = default (single-sample path: on the device, one thread per component,
 * pf_replay_device; pfh_find_replay and the colored path: host threads / sequential), 0 = the sequential loop on the host,
 * n >= 2 = n host threads, component by component */
void pfh_set_replay_threads(pfh_run *, int threads);
/* several ranks running findSuperBubble on the same graph into one output directory: 0 on all but the rank that writes
 * <outpre>_super_bubble.txt (the rows are computed everywhere; PloidyEstimation does not need the file) */
void pfh_set_write_super_bubble(pfh_run *, int on);
int pfh_ploidy_select(pfh_run *, int lower, int upper, uint64_t *n_bubbles);
/* the same for a colored run (pfh_open_colored): one (lower, upper) per colour, the reference's -C file (src/Main.cpp:775-810) */
int pfh_ploidy_select_colored(pfh_run *, const int *lower, const int *upper, int n_cutoffs, uint64_t *n_bubbles);
int pfh_ploidy_align(pfh_run *, uint64_t t0, uint64_t t1, uint64_t *n_called);
int pfh_ploidy_text(pfh_run *, uint64_t var_count_base, uint64_t sizes[10], uint64_t counters[8]);
int pfh_ploidy_write(pfh_run *, const char *outpre, const uint64_t offsets[10], const uint64_t totals[10], int truncate);

/* The same exchange without a device (CPU tests of the N > 1 path): the records of a unitig range from the host walker alone
 * (pfh_host_walk_range: every oriented vertex with out-degree > 1 on unitigs [u0, u1), ascending; returns the number of records,
 * *pool_used the list entries; UINT64_MAX when a capacity is too small) and the commit replay on a bare state
 * (pfh_replay_open: n_unitigs, -z; pfh_replay_apply: one shard's records, in shard order; pfh_replay_state as pfh_state). */
uint64_t pfh_host_walk_range(const uint32_t *succ, const uint32_t *pred, uint32_t n_unitigs, uint32_t u0, uint32_t u1,
                             pf_bfs_record *records, uint64_t rec_cap, uint32_t *pool, uint64_t pool_cap, uint64_t *pool_used);
typedef struct pfh_replay pfh_replay;
pfh_replay *pfh_replay_open(uint32_t n_unitigs, uint32_t complex_size);
void pfh_replay_close(pfh_replay *);
int pfh_replay_apply(pfh_replay *, const pf_bfs_record *records, uint64_t n_records, const uint32_t *pool);
void pfh_replay_state(const pfh_replay *, uint8_t *flags, uint32_t *plus, uint32_t *minus);
/* The same replay spread over `threads` host threads: records whose footprints (the unitig sides they can touch) are disjoint
 * commute, so the connected components of {sides, "touched by one record"} are replayed side by side, each in record order
 * (csrc/host/pf_replay_par.hpp).  A handle takes either this call or pfh_replay_apply.  pfh_side_components: the component
 * label of every record's entrance side (what pf_side_components computes on the device); pfh_replay_check_footprints: the
 * sequential replay with every state access checked against those components, `slice` records at a time (0 = all at once)
 * -> number of accesses outside the running record's component (0 = the model holds), *first_bad = the first such record. */
int pfh_replay_apply_parallel(pfh_replay *, const pf_bfs_record *records, uint64_t n_records, const uint32_t *pool, uint32_t threads);
void pfh_side_components(const pf_bfs_record *records, uint64_t n_records, const uint32_t *pool, uint32_t n_unitigs, uint32_t *labels);
uint64_t pfh_replay_check_footprints(const pf_bfs_record *records, uint64_t n_records, const uint32_t *pool, uint32_t n_unitigs,
                                     uint32_t complex_size, uint64_t slice, uint64_t *first_bad);
/* the same with the colored commits (CCDBG): colour sets of an opened (graph, colours) pair, succ = the graph's CSR rows [2N][4] */
struct pfh_colors;
uint64_t pfh_colors_check_footprints(const struct pfh_colors *, const uint32_t *succ, const pf_bfs_record *records, uint64_t n_records,
                                     const uint32_t *pool, uint32_t complex_size, uint64_t slice, uint64_t *first_bad);

/* Kmer::hash(seed) of the reference's Bifrost build (wyhash over the 8-byte left-aligned k-mer) */
uint64_t pfh_bifrost_kmer_hash(uint64_t left_aligned_kmer, uint64_t seed);

#ifdef __cplusplus
}
#endif
#endif
