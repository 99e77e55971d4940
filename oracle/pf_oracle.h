/*
 * pf_oracle.h -- C interface of the CPU ORACLE (test infrastructure, NOT product code).
 *
 * oracle/ is a plain, single-threaded CPU restatement of PloidyFrost's
 * superbubble + variant-calling hot path (reference: src/CDBG.cpp,
 * src/SeqAlign.cpp, plus the semantics of the vendored Bifrost graph and KMC
 * reader that the path relies on -- see SURVEY.md section 3.1 / 5.8).  It exists
 * only to CHECK the HIP path: tests/, __graft_entry__.smoke() and the
 * cpu_baseline leg of bench.py may use it; nothing under ploidyfrost_amd/ links,
 * imports or executes it.
 *
 * Parity pin: the oracle is pinned byte-for-byte against the twelve output
 * files of the real reference binary (oracle/_ref/PloidyFrost, built by
 * oracle/Makefile.ref from /root/reference) on the fixtures under
 * tests/golden/ (tests/test_oracle_golden.py).
 *
 * Conventions shared with the product:
 *   unitig index u   : 0-based, id = u+1 is the reference's MyUnitig id (CDBG.cpp:131-136)
 *   oriented vertex  : ov = 2*u + (strand ? 0 : 1)   ('+' even, '-' odd); ov^1 flips
 *   PFO_NONE         : empty CSR slot
 */
#ifndef PF_ORACLE_H_
#define PF_ORACLE_H_
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define PFO_NONE 0xFFFFFFFFu

typedef struct pfo_ctx pfo_ctx;

/* outcomes of one traversal (CDBG.cpp:253-415) */
enum { PFO_BFS_NONE = 0, PFO_BFS_CYCLE_EXIT = 1, PFO_BFS_REJECT = 2, PFO_BFS_ACCEPT = 3 };

pfo_ctx *pfo_open(const char *gfa_path, const char *kmc_prefix); /* NULL on error (message on stderr) */
void pfo_close(pfo_ctx *);
const char *pfo_last_error(void);

int pfo_k(const pfo_ctx *);
uint32_t pfo_num_unitigs(const pfo_ctx *);
uint64_t pfo_num_kmers(const pfo_ctx *);
/* unitig u in reference orientation; returns length, copies at most cap bytes (no NUL) */
uint32_t pfo_unitig_seq(const pfo_ctx *, uint32_t u, char *buf, uint32_t cap);

/* adjacency (G2): succ/pred arrays of 2N*4 entries, A,C,G,T slot order */
void pfo_adjacency(const pfo_ctx *, uint32_t *succ, uint32_t *pred);

/* C1: per-unitig coverage.  Returns 0 ok, 1 if some k-mer is missing (reference would exit). */
int pfo_unitig_cov(const pfo_ctx *, uint32_t u, uint64_t *sum, uint32_t *min_count);
/* the same for an oriented unitig ov = 2u + (strand ? 0 : 1): the orientation matters only for a database without canonical counting */
int pfo_unitig_cov_oriented(const pfo_ctx *, uint32_t ov, uint64_t *sum, uint32_t *min_count);
/* C2: coverage of a >=k-length string; *ok = 0 when a count is outside (low,up);
 * returns 1 if a k-mer is missing. */
int pfo_string_cov(const pfo_ctx *, const char *s, uint32_t len, uint32_t low, uint32_t up,
                   uint64_t *sum, int *ok);
/* single canonical k-mer count (K2/K3 composite); returns 1 = found */
int pfo_kmer_count(const pfo_ctx *, const char *kmer, uint32_t *count);

/* S2: one traversal from oriented vertex s.  seen/cyc must hold cap entries each.
 * Returns outcome; *exit_ov = PFO_NONE when outcome is NONE. */
int pfo_extract(const pfo_ctx *, uint32_t s_ov, uint32_t *exit_ov, uint32_t *n_seen, uint32_t *seen,
                uint32_t *n_cyc, uint32_t *cyc, uint32_t cap, int *flag_cycle, int *flag_tip);

/* A1: SeqAlign::SequenceAlignment over n strings.  Output: number of rows (0 = no alignment),
 * rows written '\n'-separated into out (cap bytes); snp/indel positions and the partition
 * matrix (cols x rows, uint16) written to the int arrays when non-NULL. */
int pfo_seq_align(double M, double D, double G, const char *const *strs, int n, char *out, uint32_t cap,
                  uint32_t *n_snp, uint32_t *snp_pos, uint32_t *n_indel, uint32_t *indel_pos,
                  uint32_t *n_indel_len, uint32_t *indel_len, uint32_t *n_cols, uint16_t *partition,
                  uint32_t pos_cap, uint32_t part_cap);

/* needlemanWunch + traceback for one pair (SeqAlign.cpp:480-549, 306-478): the kept alignments,
 * one per line "a_row\tb_row\tscore\tn_pos\tindel\tgap,gap,..."; returns their number (negative
 * when out was too small). */
int pfo_pairwise(double M, double D, double G, const char *a, const char *b, char *out, uint32_t cap);

/* S1 + P1..P3 + O1: the whole path; writes the twelve <outdir>/<prefix>_*.txt files.
 * Per-unitig state after findSuperBubble is kept in the context.  0 = ok. */
int pfo_set_unitig_id(pfo_ctx *, const char *outdir, const char *prefix);
int pfo_find_superbubbles(pfo_ctx *, const char *outdir, const char *prefix, uint32_t complex_size,
                          uint64_t *n_bubbles);
int pfo_ploidy_estimation(pfo_ctx *, const char *outdir, const char *prefix, int lower, int upper,
                          double M, double D, double G, uint64_t allele[4], uint64_t *core_cov,
                          uint64_t *core_num);
/* per-unitig state after pfo_find_superbubbles: flags byte (MyUnitig.hpp bit layout),
 * plus / minus partner ids (0 = NULL, own id = self) */
void pfo_state(const pfo_ctx *, uint8_t *flags, uint32_t *plus, uint32_t *minus);

/* ---- colored (multi-sample) twin, reference src/CCDBG.cpp ------------------------------------
 * colors_dump: the text oracle/_ref/colors_dump (oracle/ref_colors_dump.cpp) wrote for the graph, i.e.
 * the real Bifrost's reading of its .bfg_colors; db_list_file: one KMC prefix per line, one per colour
 * (CCDBG.cpp:13-43).  pfo_set_unitig_id / pfo_find_superbubbles / pfo_state work on such a context
 * with the colored commits (CCDBG.cpp:2349-2661) and the colored super_bubble listing (:2105-2132). */
pfo_ctx *pfo_open_colored(const char *gfa_path, const char *colors_dump, const char *db_list_file);
uint32_t pfo_num_colors(const pfo_ctx *);
/* out[colour * n_kmers + i] = 1 when k-mer i of unitig u carries the colour; returns UnitigColors::size(um) */
uint64_t pfo_unitig_colors(const pfo_ctx *, uint32_t u, uint8_t *out, uint32_t *n_full_enc);
/* readCovUni / readCov(string) for one colour (CCDBG.cpp:123-156, 89-122): *ok = 0 when a k-mer is
 * missing or a count lies outside (low, up) */
void pfo_unitig_cov_color(const pfo_ctx *, uint32_t colour, uint32_t u, uint32_t low, uint32_t up, double *mean, int *ok);
void pfo_string_cov_color(const pfo_ctx *, uint32_t colour, const char *s, uint32_t len, uint32_t low, uint32_t up,
                          double *mean, int *ok);
/* CompactedDBG::findUnitig(s, 0, len): 1 = found; unitig index, forward dist, matched k-mers */
int pfo_find_unitig(const pfo_ctx *, const char *s, uint32_t len, uint32_t *u, uint32_t *dist, uint32_t *n);
/* lower/upper: one cutoff pair per colour (Main.cpp:398-455) */
int pfo_ploidy_estimation_colored(pfo_ctx *, const char *outdir, const char *prefix, const int *lower, const int *upper,
                                  double M, double D, double G, uint64_t allele[4], uint64_t *core_cov, uint64_t *core_num);

/* ---- `PloidyFrost model` (src/GmmModel.cpp, src/Main.cpp:636-692), pf_oracle_gmm.cpp ---------------------------------- */
typedef struct pfo_gmm pfo_gmm;
pfo_gmm *pfo_gmm_open(void);
void pfo_gmm_close(pfo_gmm *);
const char *pfo_gmm_error(const pfo_gmm *);
int pfo_gmm_read_fre(pfo_gmm *, const char *file, double min_frequency);      /* GmmModel::readFreFile */
int pfo_gmm_read_cov(pfo_gmm *, const char *prefix, double min_frequency);    /* GmmModel::readCovFile */
void pfo_gmm_set_values(pfo_gmm *, const double *values, uint64_t n);
uint64_t pfo_gmm_size(const pfo_gmm *);
void pfo_gmm_values(const pfo_gmm *, double *out);
void pfo_gmm_fit(pfo_gmm *, uint32_t gauss, double m_thre, double n_thre, int max_iter, double max_delta, double *weights,
                 double *means, double *vars, double *loglik, double *aic, uint32_t *iterations);
int pfo_gmm_run(pfo_gmm *, int min_gauss, int max_gauss, double m_thre, double n_thre, int max_iter, double max_delta,
                const char *outprefix);

#ifdef __cplusplus
}
#endif
#endif
