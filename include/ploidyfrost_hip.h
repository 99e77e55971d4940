/*
 * ploidyfrost_hip.h -- C ABI of the MI355X (gfx950) device layer for PloidyFrost's
 * superbubble + variant-calling hot path.
 *
 * PloidyFrost has no plugin/FFI mechanism: its hot path sits behind the C++ classes CDBG
 * (reference src/CDBG.hpp:20-41) and CCDBG, called only from main() (src/Main.cpp:829-849).
 * This header is the boundary a maintainer would bind instead: every entry point names the
 * reference function(s) whose compute it replaces.  Plain C types only; every buffer is
 * caller-allocated; every call returns an int status (PF_OK = 0) instead of exit();
 * one context per host thread / per GPU.
 *
 * Pointers marked [host|dev] may be host or device addresses (copied with
 * hipMemcpyDefault), which lets a PyTorch caller hand in tensor storage directly.
 *
 * Conventions
 *   unitig index u : 0-based, in the reference's iteration order (id = u + 1,
 *                    src/CDBG.cpp:131-136: long unitigs in S-line order, then short ones)
 *   oriented vertex: ov = 2*u + (strand ? 0 : 1); ov ^ 1 is the reverse complement
 *   k-mer          : uint64, 2 bits/base (A0 C1 G2 T3), first base most significant,
 *                    right aligned; 3 <= k <= 31
 *   PF_NONE        : empty adjacency slot / no vertex
 */
#ifndef PLOIDYFROST_HIP_H_
#define PLOIDYFROST_HIP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PF_NONE 0xFFFFFFFFu

enum pf_status {
    PF_OK = 0,
    PF_ERR_ARG = 1,         /* bad argument / call order */
    PF_ERR_HIP = 2,         /* HIP runtime error, see pf_last_error() */
    PF_ERR_NO_DEVICE = 3,   /* no gfx950 device visible: there is no CPU fallback */
    PF_ERR_OVERFLOW = 4,    /* a caller-provided output buffer is too small */
    PF_ERR_MISSING_KMER = 5 /* a graph k-mer is absent from the count table
                               (the reference exit()s: src/CDBG.cpp:52-56, 92-96) */
};

typedef struct pf_ctx pf_ctx;

/* ---- context ------------------------------------------------------------------------- */
int pf_create(int device, pf_ctx **out);
/* Optional: initialise the HIP runtime and the device ahead of pf_create, e.g. on a helper thread while input files are read. */
int pf_warmup(int device);
void pf_destroy(pf_ctx *);
const char *pf_last_error(const pf_ctx *); /* ctx may be NULL: last creation error */
/* Launch on this hipStream_t (e.g. torch's current stream) instead of the context's own. */
int pf_set_stream(pf_ctx *, void *hip_stream);
int pf_synchronize(pf_ctx *);
/* Per-kernel HIP-event timing: when enabled every launch is bracketed by events on the
 * launch stream; pf_kernel_time() synchronises and returns the accumulated time / count. */
enum pf_kernel {
    PF_K_TABLE_BUILD = 0, PF_K_ADJ_INSERT, PF_K_ADJ_PROBE, PF_K_COV, PF_K_BFS, PF_K_BFS_BIG,
    PF_K_ALIGN, PF_K_ALIGN_BIG, PF_K_STRCOV, PF_K_BUBBLE, PF_K_BUBBLE_BIG, PF_K_COV_COLORED, PF_K_STRCOV_COLORED, PF_K_GMM, PF_K_KMC_DECODE, PF_K_MINZ, PF_K_COV_JOIN,
    PF_K_CALL_SCAN, PF_K_CALL_PREP, PF_K_CALL_PATHS, PF_K_CALL_SITES, PF_K_CALL_FORMAT, PF_K_CALL_SNP, PF_K_BFS_THREAD, PF_K_CALL_PAIR, PF_K_CALL_STACK,
    PF_K_COV_JOIN_REST, /* second kernel of K-COV-JOIN: the look-ups whose first line was full */
    PF_K_COPY_TEXT, /* not a kernel of this library: the copies of result text to the host (the runtime moves them with a kernel of its own) */
    PF_K_COUNT_
};
int pf_enable_timing(pf_ctx *, int on);
int pf_kernel_time(pf_ctx *, int kernel, double *total_ms, uint64_t *launches);
int pf_reset_timing(pf_ctx *);
/* Which kernels pf_enable_timing times: bit k of the mask = launches of kernel k (default: all).  Two events per launch are not
 * free on a pass of a hundred launches from four host threads; a caller that wants one kernel's durations inside a region it
 * also times as a whole (bench.py: K-BUBBLE for the roofline) selects that kernel alone. */
int pf_timing_select(pf_ctx *, uint64_t kernel_mask);
/* The union of the timed launches' intervals of the kernels of the mask, in ms: what a kernel that is launched several times side by
 * side (K-BUBBLE: a launch per size class and align range) really occupies of a pass -- the sum pf_kernel_time gives counts the
 * overlap as often as there are launches. */
int pf_kernel_busy(pf_ctx *, uint64_t kernel_mask, double *busy_ms);
/* Device-busy time of everything timed since the last reset: the UNION of the launches' [start, end] intervals (HIP events on the
 * streams they were launched on), in ms -- the launches of the calling pipeline overlap (two align ranges, K-BUBBLE's classes on
 * streams of their own, K-TEXT beside the next range's alignment), so the sum pf_kernel_time gives counts overlapped time twice;
 * *span_ms = first start to last end.  Library launches between them (rocPRIM scans / sorts, copies, fills) are not timed. */
int pf_device_busy(pf_ctx *, double *busy_ms, double *span_ms);
/* Work items the timed launches of a kernel were given since the last reset, in the unit its algorithmic bytes are quoted per
 * (DESIGN.md 3): candidates for K-BFS's tiers, bubbles for K-SNP / K-PAIR / K-BUBBLE / K-PATHS / K-SITES / K-TEXT, sides for K-SCAN. */
int pf_kernel_units(pf_ctx *, int kernel, uint64_t *units);
const char *pf_kernel_name(int kernel);

/* ---- graph (replaces Bifrost's CompactedDBG storage for this path) --------------------- */
/* Packed unitigs: unitig u occupies words [seq_off[u], seq_off[u] + ceil(len_bp[u]/32)) of
 * seq_words; base j sits in word j/32 at bits [62 - 2*(j%32), 63 - 2*(j%32)] (first base
 * most significant), padding bits zero.  seq_off has n_unitigs + 1 entries. [host|dev] */
int pf_upload_graph(pf_ctx *, const uint64_t *seq_words, const uint64_t *seq_off, const uint32_t *len_bp,
                    uint32_t n_unitigs, int k);
/* K-GFA (pf_gfa.hip): the same graph straight from the bytes of a Bifrost GFA file -- the parse half of CompactedDBG::read
 * (bifrost/src/CompactedDBG.tcc:823-960, 7888-7908; GFA_Parser.cpp:380-520) on the device.  body = the file after its header
 * line [host|dev], gfa_version 1 or 2 (the header's VN:Z), k from its KL:Z.  Segment lines are located, their sequences packed two
 * bits per base, the k-length ones canonicalised; the graph is resident afterwards as after pf_upload_graph, in the unitig order
 * of pf_upload_graph's callers before abundant k-mers are moved (segments longer than k in file order, then the k-length ones).
 * PF_ERR_ARG with the loader's messages: "missing fields in a segment line", "segment shorter than k", "non-ACGT base in a
 * segment", "no segments in the GFA file".  A sequence field is every byte between its tabs / the line feed, as GFA_Parser.cpp:497-520
 * takes it: the '\r' that ends the sequence of a CRLF file is its last base (A in a segment longer than k, CompressedSequence.cpp:597-614;
 * T in a k-length one, Kmer.cpp:92-107), as in the reference; any other byte that is no base is refused.  pf_gfa_segments hands the host what it keeps per unitig (once per ingest; any
 * pointer may be NULL): length, offset of the sequence field inside body, file rank among the segments, the DA:Z tag (-1 =
 * none; *any_da = some segment had one), and whether a k-length unitig is stored as its reverse complement. */
int pf_gfa_ingest(pf_ctx *, const char *body, uint64_t n_bytes, int gfa_version, int k, uint32_t *n_unitigs, uint32_t *n_short);
/* The same in two steps: pf_gfa_parse (locate + pack, on a stream of its own, touching nothing else of the context: it may run
 * on one thread while another feeds the count table to the same context; its error text: pf_gfa_error) and pf_gfa_upload (the
 * packed arrays become the context's graph, as pf_upload_graph). */
int pf_gfa_parse(pf_ctx *, const char *body, uint64_t n_bytes, int gfa_version, int k, uint32_t *n_unitigs, uint32_t *n_short);
int pf_gfa_upload(pf_ctx *);
const char *pf_gfa_error(const pf_ctx *);
int pf_gfa_segments(pf_ctx *, uint32_t *len_bp, uint64_t *seq_off, uint32_t *file_rank, int16_t *da_tag, uint8_t *stored_rc, int *any_da);
/* G2: neighbour discovery (bifrost/src/NeighborIterator.tcc:25-47 via
 * CompactedDBG::find(km, extremities_only=true), CompactedDBG.tcc:1403-1523): joins the
 * end k-mers of all unitigs on the device and fills the CSR kept in the context.
 * succ/pred (2*n_unitigs*4 entries, A,C,G,T slot order) may be NULL. [host|dev] */
int pf_build_adjacency(pf_ctx *, uint32_t *succ, uint32_t *pred);

/* ---- k-mer count table (replaces CKMCFile, KMC/kmc_api/kmc_file.cpp) ------------------- */
/* K-MINZ (GFA ingest, unitig numbering): how often does the most frequent minimizer occur in the uploaded graph?  Bifrost numbers
 * a k-length unitig last when the bucket of its minimizer already holds 15 entries (CompactedDBG.tcc:3928-4080); the loader has to
 * replay that bookkeeping on the host only when some bucket can get that full.  Counts every g-mer position that is the minimum
 * (bifrost/src/RepHash.hpp hash; not at either end of the k-mer, minHashIterator.hpp:63-119) of the window of some k-mer containing it,
 * in a table of pf_minimizer_table_slots(n_kmers) slots indexed by a mix of the canonical minimizer -- per slot an upper bound of
 * the entries Bifrost can file there.  *max_occurrences < 15  =>  no abundant k-mer: ids are long unitigs, then k-length ones, in
 * file order.  crowded_slots (optional): slots that reached `limit`; table_out (optional, host or device): the u32 counters. */
uint64_t pf_minimizer_table_slots(uint64_t n_kmers);
int pf_minimizer_crowding(pf_ctx *, int g, uint32_t limit, uint32_t *max_occurrences, uint64_t *crowded_slots, uint32_t *table_out);
/* For a graph that CAN crowd a bucket (max_occurrences >= limit), what the host replay of Bifrost's bookkeeping
 * (csrc/host/pf_host_minz.cpp) otherwise computes in two passes of its own over every unitig: the census table narrowed to
 * saturating bytes (pf_minimizer_table_slots(n_kmers) of them) and, per unitig in upload order, whether one of its counted positions
 * falls into a slot that reached `limit` -- the unitigs that have to go through addUnitig's replay.  Both are upper bounds of the
 * host's own (ties inside a window count twice here), which only makes the replay follow a few more buckets and unitigs.
 * counters8_out, unitig_flags_out: [host|dev] */
int pf_minimizer_replay_inputs(pf_ctx *, int g, uint32_t limit, uint8_t *counters8_out, uint8_t *unitig_flags_out);

/* K-KMC: KMC database ingest on the device.  `records` = the record area of <db>.kmc_suf (after its 4-byte marker): n_records
 * records of suffix_bytes = (k - lut_prefix_len) / 4 suffix bytes (most significant first) + counter_bytes counter bytes (least
 * significant first), ordered as the prefix table says (KMC/kmc_api/kmc_file.cpp:185-302, 775-782): lut[e] = index of the first
 * record of table entry e, e = prefix for the KMC1 layout and bin * 4^p + prefix for the KMC2 (signature-binned) layout, with
 * n_lut entries plus lut[n_lut] = n_records.  Output: two device arrays (exact k-mers as stored, counts truncated to 32 bits)
 * to hand to pf_upload_counts / pf_upload_counts_colored, released with pf_device_free.  records: [host|dev]; lut: host */
int pf_kmc_decode(pf_ctx *, const uint8_t *records, uint64_t n_records, uint32_t suffix_bytes, uint32_t counter_bytes,
                  const uint64_t *lut, uint64_t n_lut, uint32_t lut_prefix_len, uint32_t k, uint64_t **kmers_dev, uint32_t **counts_dev);
void pf_device_free(pf_ctx *, void *device_pointer);
int pf_copy_to_host(pf_ctx *, void *dst, const void *src_dev, size_t bytes);

/* K1/K2: builds the device hash table from the database records (exact k-mers as stored, any
 * order).  Records with count outside [min_count, max_count] are not retrievable, as in
 * CKMCFile::BinarySearch (kmc_file.cpp:1459).  both_strands mirrors GetBothStrands().
 * k = the database's k-mer length (CKMCFileInfo::kmer_length; it must equal the graph's, whichever of the two is uploaded
 * first): the table is addressed by the MINIMIZER of a key, not by a hash of the whole key, so that the k-mers of a unitig --
 * which CDBG::readCov asks for one after the other (src/CDBG.cpp:66-120) -- lie in the same few 128-B lines
 * (csrc/pf_device_common.hpp, "minimizer-local addressing").  [host|dev] */
int pf_upload_counts(pf_ctx *, const uint64_t *kmers, const uint32_t *counts, uint64_t n, uint32_t k, uint64_t min_count,
                     uint64_t max_count, int both_strands);
/* K-COV-JOIN again (it runs by itself when graph and table are both resident): every look-up CDBG::readCov(UnitigMap) makes in one
 * run of the reference -- one CheckKmer per graph k-mer (src/CDBG.cpp:66-120 inside :1187-1220) -- as one kernel launch.  For
 * callers that account for those look-ups per pass (bench.py's value_incl_join).  PF_ERR_ARG without a graph and a canonical table. */
int pf_join_counts(pf_ctx *);
/* The same in two halves: _begin launches the look-ups on a stream of their own (lowest priority) and returns; whatever the caller
 * runs next on the context that reads no coverage -- pf_bfs_candidates ... pf_replay_device: findSuperBubble -- shares the device
 * with them; _end waits.  Every reader of the joined array (pf_unitig_cov, pf_call_coverage) waits by itself, so _end may be left out. */
int pf_join_counts_begin(pf_ctx *);
int pf_join_counts_end(pf_ctx *);
/* K2+K3 composite of src/CDBG.cpp:38-56 for a batch of k-mers:
 * "if (!IsKmer(fwd)) reverse(); CheckKmer()".  found[i] = 0 when absent. [host|dev] */
int pf_lookup_kmers(pf_ctx *, const uint64_t *kmers, uint64_t n, uint32_t *counts, uint8_t *found);

/* ---- C1: CDBG::readCov(const UnitigMap&) (src/CDBG.cpp:66-120) for unitigs [u0, u1) ------ */
/* sum = sum of canonical counts over the unitig's k-mers, min = min(10000, counts),
 * miss = 1 if some k-mer is absent.  Arrays are indexed from u0.  The host does the one
 * division (mean = sum / len). Returns PF_ERR_MISSING_KMER after filling the arrays if
 * any miss flag is set. [host|dev] */
int pf_unitig_cov(pf_ctx *, uint32_t u0, uint32_t u1, uint64_t *sum, uint32_t *min, uint8_t *miss);
/* How pf_unitig_cov gets there: as soon as a graph and a canonical count table are both resident (whichever of
 * pf_upload_graph / pf_upload_counts comes second) the device joins them once -- every graph k-mer's count is written
 * to its position in graph order, a per-k-mer coverage SoA beside the sequence SoA (kernel PF_K_COV_JOIN; the
 * load-time counterpart of pf_build_adjacency).  pf_unitig_cov then streams that array: a segmented sum / min over
 * 4 contiguous bytes per k-mer instead of one random table access per k-mer.  pf_unitig_cov_probe is the same
 * function computed the other way -- every k-mer looked up in the hash table at call time -- kept for databases the
 * SoA cannot hold (max_count = 2^32 - 1) and as the independent check of the join (tests, measurements).
 * Identical results by construction. [host|dev] */
int pf_unitig_cov_probe(pf_ctx *, uint32_t u0, uint32_t u1, uint64_t *sum, uint32_t *min, uint8_t *miss);
/* The branch for a database built without canonical counting (GetBothStrands() == false, src/CDBG.cpp:94-117): every k-mer of
 * mappedSequenceToString() is looked up as it reads.  reverse = 0: the unitig as stored; 1: its reverse complement (the '-'
 * orientation).  pf_unitig_cov refuses such a table; pf_string_cov returns sum 0 / ok 1 for it without looking anything up, as
 * readCov(string) does (:34, :59). [host|dev] */
int pf_unitig_cov_exact(pf_ctx *, uint32_t u0, uint32_t u1, int reverse, uint64_t *sum, uint32_t *min, uint8_t *miss);

/* ---- S2: CDBG::extractSuperBubble_ptr (src/CDBG.cpp:253-415), pure part ------------------ */
typedef struct pf_bfs_record {
    uint32_t entrance;  /* oriented vertex s */
    uint32_t exit;      /* oriented vertex t, PF_NONE when outcome == PF_BFS_NONE */
    uint32_t n_seen;    /* |vec_km_seen| (full length, even when list not stored) */
    uint32_t n_list;    /* entries stored at list_off: seen[] for outcomes 1..3, the
                           cycle set for outcome 0 with flag_cycle */
    uint64_t list_off;  /* offset into the vertex pool */
    uint8_t outcome;    /* enum pf_bfs_outcome */
    uint8_t flag_cycle;
    uint8_t flag_tip;
    uint8_t strict;     /* accept only: the structural test of src/CDBG.cpp:765-782 passed */
    uint32_t pad_;
} pf_bfs_record;

enum pf_bfs_outcome {
    PF_BFS_NONE = 0,       /* stack ran empty: no exit (src/CDBG.cpp:373-414 applies) */
    PF_BFS_CYCLE_EXIT = 1, /* exit found, s is a successor of t  -> setNoBubble_ptr_cycle */
    PF_BFS_REJECT = 2,     /* exit found, cycle or tip inside     -> setNoBubble_ptr(seen, p) */
    PF_BFS_ACCEPT = 3      /* exit found, clean                   -> setNoBubble_ptr(p, seen) */
};

/* Candidates: every oriented vertex with out-degree > 1 (the order-dependent
 * `partner == NULL` gate of src/CDBG.cpp:206,211 is applied by the host during replay).
 * pf_count_candidates returns how many lie in unitigs [u0, u1); pf_bfs_candidates traverses
 * them, one wavefront each, and writes one record per candidate in ascending entrance
 * order plus the variable-length vertex lists into pool (pool_cap entries).
 * PF_ERR_OVERFLOW: *pool_used tells the size needed. [host|dev] */
int pf_count_candidates(pf_ctx *, uint32_t u0, uint32_t u1, uint64_t *n_candidates);
int pf_bfs_candidates(pf_ctx *, uint32_t u0, uint32_t u1, pf_bfs_record *records, uint64_t rec_cap,
                      uint32_t *pool, uint64_t pool_cap, uint64_t *n_records, uint64_t *pool_used);
/* The same with everything beyond the first (LDS, 128-entry) tier left to the caller: a long traversal is one chain of
 * dependent memory accesses (LIFO order fixes its result), which a host core walks two orders of magnitude faster than a lone
 * wavefront (~20 ns against ~2.4 us per vertex).  records / pool must be host memory; deferred[0 .. *n_deferred) = indices (into records) of those candidates,
 * whose records come back with only `entrance` set (exit = PF_NONE, outcome = PF_BFS_NONE, no list) for the caller to fill. */
int pf_bfs_candidates_split(pf_ctx *, uint32_t u0, uint32_t u1, pf_bfs_record *records, uint64_t rec_cap, uint32_t *pool,
                            uint64_t pool_cap, uint64_t *n_records, uint64_t *pool_used, uint32_t *deferred, uint64_t deferred_cap,
                            uint64_t *n_deferred);
/* pf_bfs_candidates_split in two steps, so that the copy of the records to the host (183 MB for the 5 M-unitig graph) runs
 * beside the caller's walk of the deferred traversals: _begin returns once the device tiers have run, with the deferred
 * candidates' indices and entrances (oriented vertices) and with records / pool on their way on a copy stream; _end waits for
 * them and marks the deferred records as pf_bfs_candidates_split does.  Between the two calls the caller must not touch
 * records / pool; it may call pf_side_components(records = NULL). */
int pf_bfs_candidates_begin(pf_ctx *, uint32_t u0, uint32_t u1, pf_bfs_record *records, uint64_t rec_cap, uint32_t *pool, uint64_t pool_cap,
                            uint64_t *n_records, uint64_t *pool_used, uint32_t *deferred, uint32_t *deferred_entrance, uint64_t deferred_cap,
                            uint64_t *n_deferred);
int pf_bfs_candidates_end(pf_ctx *);

/* ---- A1: SeqAlign::needlemanWunch + traceback (src/SeqAlign.cpp:480-549, 306-478) ------- */
/* One job = one pairwise alignment A x B.  Sequences are ASCII over {A,C,G,T,-}
 * ('-' only in A, when A is a row of an earlier alignment).  For every job the device
 * performs the full-matrix fill (scores in `int = long + double` arithmetic, +1 same-direction
 * bonus, look-ahead rule, all tying directions flagged) and the all-co-optimal traceback with
 * the shrinking 5-gap-open budgets, and returns the kept alignments in traceback order. */
typedef struct pf_align_job {
    uint64_t a_off, b_off; /* offsets into the text buffer */
    uint32_t a_len, b_len;
} pf_align_job;

typedef struct pf_align_hit {
    uint64_t text_off;   /* aligned rows: a at text_off, b at text_off + len (out_text) */
    uint64_t gap_off;    /* gap_pos entries at gap_off (out_gaps), traceback order */
    uint32_t len;        /* aligned length */
    uint32_t n_gaps;
    int64_t score;       /* variantAnalyze score (src/SeqAlign.cpp:255) */
    uint32_t n_pos;      /* variant positions */
    uint32_t n_indel;
} pf_align_hit;

/* hits of job j are hits[hit_first[j] .. hit_first[j] + hit_count[j]) (both arrays have n_jobs
 * entries; jobs are published in completion order, so the ranges are not sorted by j).
 * PF_ERR_OVERFLOW when a capacity is too small: the sizes needed are returned in used[3] =
 * {hits, text bytes, gap entries}. [text, jobs and all outputs: host|dev] */
int pf_align_batch(pf_ctx *, const char *text, uint64_t text_len, const pf_align_job *jobs, uint32_t n_jobs,
                   double match, double mismatch, double gap, uint64_t *hit_first, uint32_t *hit_count,
                   pf_align_hit *hits, uint64_t hit_cap, char *out_text, uint64_t text_cap, uint32_t *out_gaps,
                   uint64_t gap_cap, uint64_t used[3]);

/* ---- A1 complete: SeqAlign::SequenceAlignment (src/SeqAlign.cpp:550-640) -------------------- */
/* One task = one bubble: its N >= 2 path strings in the order the host sorted them.  The device
 * runs the whole multiple alignment for it on one wavefront: needlemanWunch + traceback of rows 0
 * and 1, then rows 2..N-1 progressively against row 0 of every kept alignment (re-opening the new
 * gaps in the older rows, re-scoring them as variantAnalyze does, keeping the best-ranked
 * candidates), and finally the selection ladder of compareStrPair (src/SeqAlign.cpp:8-236).
 * A path is either a slice of `text` (ASCII ACGT) or, with ov != PF_NONE, the whole oriented
 * unitig ov decoded from the packed graph (mappedSequenceToString). */
typedef struct pf_bubble_path {
    uint64_t text_off;
    uint32_t len;
    uint32_t ov;
} pf_bubble_path;

typedef struct pf_bubble_task {
    uint64_t path_first; /* index into paths */
    uint32_t n_paths;
    uint32_t pad_;
} pf_bubble_task;

typedef struct pf_bubble_site { /* one column with partition[col].back() > 0 */
    uint32_t col;
    uint8_t is_indel; /* col is in indel_pos (opens an indel) */
    uint8_t maxnum;   /* number of allele groups */
    uint16_t pad_;
} pf_bubble_site;

typedef struct pf_bubble_result {
    uint64_t rows_off;  /* n_rows * n_cols chars, row-major, in out_text */
    uint64_t site_off;  /* n_sites records in out_sites */
    uint64_t group_off; /* n_sites * n_rows bytes in out_groups: 1-based allele group of every row */
    uint64_t ilen_off;  /* n_indel_len u32 in out_ilen (indel_len_vec) */
    uint32_t n_rows;    /* 0: no alignment survived (the reference skips the bubble, src/CDBG.cpp:1254) */
    uint32_t n_cols;
    uint32_t n_sites;
    uint32_t n_indel_len;
} pf_bubble_result;

/* PF_ERR_OVERFLOW when a pool is too small: used[4] = {text bytes, sites, group bytes, ilen entries}
 * tells the sizes needed.  [all pointers host|dev] */
int pf_align_bubbles(pf_ctx *, const char *text, uint64_t text_len, const pf_bubble_path *paths, uint64_t n_paths,
                     const pf_bubble_task *tasks, uint32_t n_tasks, double match, double mismatch, double gap,
                     pf_bubble_result *results, char *out_text, uint64_t text_cap, pf_bubble_site *out_sites,
                     uint64_t site_cap, uint8_t *out_groups, uint64_t group_cap, uint32_t *out_ilen, uint64_t ilen_cap,
                     uint64_t used[4]);

/* ---- C2: CDBG::readCov(const string&, low, up) (src/CDBG.cpp:29-60) ---------------------- */
/* strings are ASCII ACGT, string i = text[str_off[i] .. str_off[i+1]).  sum[i] = sum of
 * canonical counts, ok[i] = 0 if any count is outside (low, up) (sum forced to 0), miss[i] = 1 if
 * a k-mer is absent. [host|dev] */
int pf_string_cov(pf_ctx *, const char *text, const uint64_t *str_off, uint32_t n_str, uint32_t low, uint32_t up,
                  uint64_t *sum, uint8_t *ok, uint8_t *miss);

/* ---- P1-P4 + O1 resident: CDBG::ploidyEstimation_ptr (src/CDBG.cpp:1101-1705) between the commit replay and the files -------
 * The caller keeps what is sequential by construction -- the commit replay of findSuperBubble and the light pass of the driver
 * loop that decides which endpoint sides are still open when reached (:1146-1186, 1656-1679) -- and the file system.  Everything
 * else runs on the device and comes back as text: owner / coverage-gate scan, path enumeration and sorting (strict: sortSeq_simple
 * :482-551; branching: two-stack walk :1364-1412 + sortSeq_branching :417-480), SequenceAlignment, per-site strings and their
 * readCov (:1448-1600, 29-60), and the rows of the ten result files with `ostream << double` formatting (:1259, 1303-1340,
 * 1552-1652).  Call order: pf_call_set_state, pf_call_coverage, pf_call_scan, pf_call_resolve (or pf_call_sides + pf_call_select), then pf_call_run /
 * pf_call_fetch per batch.  Single-sample path (CDBG); the colored twin keeps pf_align_bubbles + pf_string_cov_colored. */

/* The colored twin, CCDBG::ploidyEstimation_ptr (src/CCDBG.cpp:2759-3531), on the same pipeline.  pf_call_set_colours (once per
 * graph, after pf_upload_counts_colored) hands over what the calling phase asks of the colour sets: full_mask = the colours on every
 * k-mer of unitig u as (n_colors + 63) / 64 64-bit words per unitig (colour c = bit c % 64 of word u * words + c / 64); size_total[u] = UnitigColors::size() with the unitig's own mapping; and for a colour on part of a unitig one
 * bit per k-mer, reference orientation: entries part_first[u] .. part_first[u + 1] (N + 1 prefix) = {part_colour[e], first word
 * part_word[e] in part_bits}.  [host|dev]  With colours set, pf_call_coverage leaves readCovUni (src/CCDBG.cpp:123-156) of every
 * (colour, unitig) resident; pf_call_scan applies the per-colour gates of :2838-2931 (its lower / upper arguments are ignored:
 * pf_call_set_cutoffs gives one pair per colour, before every pf_call_scan) and the colored sortSeq_simple (:368-480); K-PATHS
 * records the unitigs each bubble's walks visit, K-SITES maps every site string to its unitig (findUnitig, :3251, 3390), asks the
 * colour sets which colours cover it (UnitigColors::contains) and reads every colour's count of its k-mers from the joined table
 * (readCov(string, low, up, colour), :89-122); K-TEXT writes one row per colour that sees two allele groups or more, with the
 * colour id and the largest Cramer's V over the colour pairs (:330-366, 2964-3059, 3285-3339).  n_colors = 0: back to the
 * single-sample path.  The -t > 1 format (pf_call_set_format) and pf_call_peek are the single-sample path's. */
int pf_call_set_colours(pf_ctx *, uint32_t n_colors, const uint64_t *full_mask, const uint64_t *size_total, const uint32_t *part_first,
                        const uint32_t *part_colour, const uint64_t *part_word, const uint64_t *part_bits, uint64_t n_part, uint64_t n_words);
int pf_call_set_cutoffs(pf_ctx *, uint32_t n_colors, const uint32_t *lower, const uint32_t *upper);

/* T1: the MyUnitig state after findSuperBubble (src/MyUnitig.hpp:37-130): b bits; plus / minus partners (0 = NULL, id = u + 1).
 * n_unitigs entries each. [host|dev] */
int pf_call_set_state(pf_ctx *, const uint8_t *flags, const uint32_t *plus, const uint32_t *minus);
/* Which of the reference's two text formats the rows follow: 0 (default) = its `-t 1` functions (ids and var_count from 1,
 * allele_frequency.txt in site order: src/CDBG.cpp:222-252, 1259-1340, 1552-1652); 1 = its `-t > 1` functions (BubbleId and
 * var_count from 0 -- fetch_add returns the old value, :1829, 2056 --, the allele_frequency rows of a bubble grouped by arity,
 * bi then tri then tetra then, in the strict branch only, penta, rows of other arities absent: :2158-2162, 2550).  Row ORDER
 * across bubbles is the deterministic `-t 1` order in both (the reference's own order with threads depends on timing). */
int pf_call_set_format(pf_ctx *, int reference_mt);
/* S1, second half: the rows of <outpre>_super_bubble.txt (src/CDBG.cpp:222-252) from that state, formatted on the device: one row
 * `BubbleId<TAB>Entrance<TAB>Strand<TAB>Exit<TAB>isSimple<TAB>isComplex` per open endpoint side in unitig order, ids from 1 (the
 * header line is the caller's).  colored_rule = 1: CCDBG's rule -- an open unitig lists every side whose partner is set, self
 * included (src/CCDBG.cpp:2106-2132).  pf_superbubble_fetch copies the text out on a stream of its own (any thread). */
int pf_superbubble_rows(pf_ctx *, int colored_rule, uint64_t *n_rows, uint64_t *text_len);
int pf_superbubble_fetch(pf_ctx *, char *dst, uint64_t len);
/* C1 for every unitig into device-resident arrays the scan reads (pf_unitig_cov / pf_unitig_cov_exact without the copy back; a
 * missing k-mer is reported by the scan only where the driver loop reads that unitig's coverage). */
int pf_call_coverage(pf_ctx *);

typedef struct pf_call_side { /* one open endpoint side, in the driver loop's order: unitig ascending, '+' side first */
    uint32_t u;          /* unitig index */
    uint32_t exit_ov;    /* the bubble's other endpoint as an oriented vertex; PF_NONE: not found (err = 2) */
    uint32_t err_unitig; /* err = 1: the unitig with a k-mer missing from the count table */
    uint8_t plus_side;
    uint8_t kind;        /* 1 complex (skipped, :1163), 2 the other endpoint owns the bubble (:1190, 1352), 3 owner: called here */
    uint8_t aligned;     /* kind 3: passes the coverage gate of the strict branch (:1210-1220); branching bubbles always do */
    uint8_t err;         /* 0, 1 missing k-mer, 2 exit unreachable -- raised by the caller only if the side is still open */
} pf_call_side;
/* part A of the driver loop for every open side, from static state only; *n_sides = number of records */
int pf_call_scan(pf_ctx *, uint32_t lower, uint32_t upper, uint64_t *n_sides);
int pf_call_sides(pf_ctx *, pf_call_side *out, uint64_t cap); /* [host|dev] */
/* part B of the driver loop on the device, exactly: which sides are still open when their unitig comes up (a handled owner closes
 * the side its exit faces, :1656-1679) is a recursion over smaller indices, settled by rounds of a monotone propagation instead of
 * a walk over the sides.  Leaves the selection (the called bubbles, in output order) in the context.  *err = 1 / 2 with *err_unitig
 * when the first open side in order carries a missing k-mer / an unreachable exit: the reference exits there. */
int pf_call_resolve(pf_ctx *, uint64_t *n_bubbles, uint32_t *err, uint32_t *err_unitig);
/* the same verdict from the caller (it walked pf_call_sides' records itself): the records, ascending, whose bubble is called [host|dev] */
int pf_call_select(pf_ctx *, const uint32_t *side_index, uint64_t n_bubbles);

#define PF_CALL_STREAMS 10
enum pf_call_stream { /* <outpre>_<name>.txt */
    PF_OUT_ALLELE_FREQUENCY = 0, PF_OUT_ALIGNSEQ = 1, PF_OUT_BIFRE = 2, PF_OUT_TRIFRE = 3, PF_OUT_TETRAFRE = 4, PF_OUT_PENTAFRE = 5,
    PF_OUT_BICOV = 6, PF_OUT_TRICOV = 7, PF_OUT_TETRACOV = 8, PF_OUT_PENTACOV = 9
};
typedef struct pf_call_result {
    uint64_t text_len[PF_CALL_STREAMS]; /* bytes of each stream this batch produced */
    uint64_t allele[4];                 /* sites with 2, 3, 4, 5 alleles */
    uint64_t core_cov, core_num;        /* coreCov / coreNum (:1261-1262) */
    uint64_t n_called;                  /* bubbles whose alignment left rows: var_count advances by this */
    uint64_t align_jobs, site_strings, n_branching;
    uint64_t snp_jobs, pair_jobs, wave_jobs; /* of align_jobs: finished by K-SNP, by K-PAIR, sent to K-BUBBLE */
    uint64_t stack_jobs;                     /* ... finished by K-STACK (paths of one length, alignment = the paths stacked) */
    uint64_t alignseq_packed_len;            /* pf_call_set_alignseq_packed: bytes stream PF_OUT_ALIGNSEQ takes in the slab (text_len keeps the text's) */
    uint64_t numeric_packed;                 /* pf_call_set_numeric_packed: 1 when the slab's fetches deliver the numeric streams at four bits a character */
} pf_call_result;
#define PF_CALL_SLABS 4 /* text slabs of a context: a slab is free again once pf_call_fetch has copied it */
#define PF_CALL_LANES 4 /* aligned ranges a context keeps resident side by side (pf_call_align_lane) */
/* Bubbles [t0, t1) of the selection (at most 2^24): everything up to the text of the ten streams, left in slab 0 .. 3 of the
 * context; var_count_base = bubbles called by earlier batches.  complex_size = -z (bounds the walk stacks). */
int pf_call_run(pf_ctx *, int slab, uint64_t t0, uint64_t t1, uint64_t var_count_base, uint32_t complex_size, double match,
                double mismatch, double gap, pf_call_result *out);
/* The same in two steps, for a bubble list cut over several GPUs: a rank learns how many of its bubbles are
 * called (pf_call_align: paths, alignment, site coverages; fills n_called, align_jobs, site_strings, n_branching), the ranks
 * exchange those counts, and only then is the text written with the rank's var_count_base (pf_call_text: text_len, allele, core_*). */
int pf_call_align(pf_ctx *, uint64_t t0, uint64_t t1, uint32_t complex_size, double match, double mismatch, double gap,
                  pf_call_result *out);
int pf_call_text(pf_ctx *, int slab, uint64_t var_count_base, pf_call_result *out);
/* alignseq.txt is more than half of the text of a pass and every row of a bubble repeats the bubble's four numbers in front of a row
 * over five letters.  on != 0: the pf_call_text* calls that follow leave stream PF_OUT_ALIGNSEQ of a slab PACKED -- an index, then per
 * bubble one header and its rows at 3 bits per character (csrc/pf_alnpack.hpp: layout, and the host routine that writes the text out
 * of it, pf::alnpack_expand) -- 2.5 x fewer bytes over PCIe.  pf_call_result.text_len[PF_OUT_ALIGNSEQ] stays the length of the TEXT
 * (file offsets); .alignseq_packed_len is what the stream takes in the slab, i.e. the length to pass to pf_call_fetch /
 * pf_call_fetch_slab for it.  Default off. */
int pf_call_set_alignseq_packed(pf_ctx *, int on);
/* The other nine streams are numbers: text over sixteen characters -- the digits, '.', tab, newline, '-', 'e', '+' -- and, with
 * alignseq packed, half of what a pass sends over PCIe.  on != 0: the pf_call_text* calls that follow leave a second form of the slab
 * for the fetches (K-NIB): every numeric stream at FOUR BITS a character (first character in the low nibble of a byte; code c of
 * "0123456789.\t\n-e+"), PF_NUMERIC_PACKED_LEN(text_len) bytes; alignseq as before, its length rounded up to 16; and, behind the ten
 * streams, 16 bytes whose first word has bit s set when stream s holds a character outside the sixteen ("nan", "inf": a frequency
 * of 0 / 0) -- that stream's nibbles are then not its text, and pf_call_fetch_text hands its text over as K-TEXT wrote it.
 * pf_call_fetch / pf_call_fetch_slab / pf_call_fetch_range speak of this form (lengths as above; pf_call_fetch_slab with every
 * stream at its full length delivers the 16 bytes of the tail as well); pf_call_result.text_len stays the text's, .numeric_packed
 * says the form is there.  Default off. */
#define PF_NUMERIC_PACKED_LEN(text_len) (((((uint64_t)(text_len)) + 1) / 2 + 15) & ~15ull)
int pf_call_set_numeric_packed(pf_ctx *, int on);
int pf_call_fetch_text(pf_ctx *, int slab, int stream, char *dst, uint64_t len);
/* Takes the device buffers pf_call_align(_lane) would take on its first call for ranges of up to n_bubbles bubbles -- to be called
 * beside the load, so that a one-shot run does not pay its first alignment launch with two dozen allocations.  A hint: sizes that
 * turn out too small grow in pf_call_align as before.  Needs nothing but the context: it may run on a helper thread while another
 * thread loads graph and count table into the same context (pf_gfa_parse, pf_upload_graph, pf_kmc_decode, pf_upload_counts) -- it
 * touches none of what those calls read or write; the first pf_call_* call of the run must come after it has returned. */
int pf_call_reserve(pf_ctx *, uint64_t n_bubbles, uint32_t complex_size);
/* The same for the first n_lanes lanes (1 .. PF_CALL_LANES; pf_call_reserve takes two): a caller that aligns several ranges side by
 * side (pf_call_align_lane below) takes every lane's working set beside the load. */
int pf_call_reserve_lanes(pf_ctx *, uint64_t n_bubbles, uint32_t complex_size, int n_lanes);
/* The same for the text stage: what the first pf_call_text_range(_lane) of a run would take for pieces of up to piece_bubbles
 * bubbles (its two streams, the size tables, the text slabs by an estimate for k = 31), and K-TEXT's two passes once over no bubbles on
 * each stream: the first launch of a kernel that spills pays for the queue's scratch (2 ms inside a first piece otherwise).
 * pf_call_reserve does the same for K-BUBBLE's class streams (13 ms of a first PloidyEstimation at 5 M unitigs).  Beside the load
 * like pf_call_reserve (single-sample path; with colours, after pf_call_set_colours). */
int pf_call_reserve_text(pf_ctx *, uint64_t piece_bubbles);
/* What pf_call_align(_lane) left resident for the bubbles of its range, before any text is made of it -- the kernel-level view the
 * parity tests hold against SeqAlign::SequenceAlignment (src/SeqAlign.cpp:550-640) bubble by bubble: per bubble its endpoints and,
 * for a strict one, its sorted inner unitigs with their mean coverages (sortSeq_simple, src/CDBG.cpp:482-551); its pf_bubble_result
 * (aligned rows, variant columns, allele groups, indel lengths: offsets into the pools below, as pf_align_bubbles returns them;
 * pf_bubble_site.pad_ of a branching bubble's site = 1 when every site string passed readCov's range test, src/CDBG.cpp:1548-1551);
 * and for a branching bubble, from sv[sv_off[bubble]], per site maxnum group coverages and their sum (src/CDBG.cpp:1527-1551).
 * cap[5] / used[1..5]: row text bytes, sites, group bytes, indel lengths, site values; used[0] = bubbles.  All pointers NULL:
 * sizes only.  [host] */
typedef struct pf_call_bubble {
    uint32_t entrance_ov, exit_ov;
    uint32_t strict, n_inner;
    uint32_t inner[4];
    double cov[4], core_mean, cov_sum;
} pf_call_bubble;
int pf_call_peek(pf_ctx *, int lane, pf_call_bubble *bubbles, pf_bubble_result *results, uint64_t *sv_off, uint64_t bubble_cap, char *text,
                 pf_bubble_site *sites, uint8_t *groups, uint32_t *ilen, double *sv, const uint64_t cap[5], uint64_t used[6]);
/* K-TEXT for bubbles [first, first + count) of the batch pf_call_align left resident: one large alignment launch (its kernels'
 * tails are paid once) can be formatted, fetched and written in pieces.  var_count_base is the same for every piece: the bubbles
 * called before the aligned batch; out->n_called = those called inside the piece. */
int pf_call_text_range(pf_ctx *, int slab, uint64_t first, uint64_t count, uint64_t var_count_base, pf_call_result *out);
/* The two with a choice of where the aligned batch lies ("lane" 0 .. PF_CALL_LANES - 1; the calls above use lane 0): the rows of
 * one range of bubbles are formatted from one host thread (pf_call_text_range_lane: a stream, scratch and counters of its own)
 * while others align the next ranges into other lanes -- the copy of the text to the host, the slowest stage of a large pass,
 * then runs beside the alignment kernels instead of after them.  A lane owns its results AND the working set of its alignment
 * (lists, pools, per-wavefront scratch, counters, streams), so pf_call_align_lane calls on DIFFERENT lanes may run at the same time
 * from different host threads: every kernel of a range ends in a tail of a few slow bubbles, which the kernels of the next range
 * fill.  One pf_call_text_range_lane at a time; never two calls on the same lane.  Their error messages are set under a lock, and
 * pf_last_error hands every calling thread a copy of its own. */
int pf_call_align_lane(pf_ctx *, int lane, uint64_t t0, uint64_t t1, uint32_t complex_size, double match, double mismatch,
                       double gap, pf_call_result *out);
int pf_call_text_range_lane(pf_ctx *, int lane, int slab, uint64_t first, uint64_t count, uint64_t var_count_base,
                            pf_call_result *out);
/* The count pass of pf_call_text_range_lane alone: text_len, allele, core_* and n_called of the range, nothing written.  A rank of a
 * sharded run learns the sizes of its whole slice with it, the ranks exchange them, and the slice is then written piece by piece
 * (pf_call_text_range_lane) at offsets that are known -- format, fetch and file copy of consecutive pieces side by side. */
int pf_call_text_sizes(pf_ctx *, int lane, uint64_t first, uint64_t count, uint64_t var_count_base, pf_call_result *out);
/* Copies the first len bytes of one stream of a slab to dst (host memory, pinned for speed) on a stream of its own: may be
 * called from another thread while pf_call_run fills the other slab. */
int pf_call_fetch(pf_ctx *, int slab, int stream, char *dst, uint64_t len);
/* All PF_CALL_STREAMS streams of a slab, packed one after the other into dst (len[s] bytes of stream s): the copies are issued
 * together and waited for once. */
int pf_call_fetch_slab(pf_ctx *, int slab, char *dst, const uint64_t *len);
/* The same packed slab in byte ranges, without waiting: range `slot` (0 / 1, alternating) is complete when pf_call_fetch_wait(slot)
 * returns, so that the caller copies one range into its files while the next crosses PCIe (a rank's one large slab of a sliced run). */
int pf_call_fetch_range(pf_ctx *, int slab, uint64_t first_byte, char *dst, uint64_t len, int slot);
int pf_call_fetch_wait(pf_ctx *, int slot);
/* O1's number formatting alone (test hook): text + 32 * i receives printf("%g", values[i]) without terminator, len[i] its length
 * [host|dev] */
int pf_format_doubles(pf_ctx *, const double *values, uint64_t n, char *text, uint8_t *len);

/* ---- K-CC: which traversal records may be committed side by side (pf_cc.hip) ------------------------------------------
 * The commits of findSuperBubble are order-dependent (src/CDBG.cpp:206-214: candidate entrances in unitig order behind the
 * `partner == NULL` gate), but two records whose footprints -- the unitig sides they can touch -- are disjoint commute.
 * pf_side_components adds the records of one slice of a pass to a union-find over the 2N sides (reset != 0 starts a pass):
 *   records == NULL: the n_records records and the vertex pool the last pf_bfs_candidates* call left on the device;
 *   otherwise records / pool [host|dev] as pf_bfs_candidates returns them (list_off into pool).
 *   extra: n_extra records [host|dev] with lists in extra_pool -- the traversals a caller of pf_bfs_candidates_split walked
 *   itself; they add their footprints only (their labels come from their entrances like everyone's).  A second call with
 *   records == NULL for the same K-BFS call adds only its `extra` (the device-resident records went in with the first).
 * pf_replay_order then labels every record of that slice with its component and deals the components into n_classes (<= 1024)
 * work units: order[n_records] = record indices grouped by class, ascending inside a class; class_off[n_classes + 1] (host);
 * labels (optional, [host|dev]) = the component label of each record's entrance side.  A record naming a vertex outside the
 * graph or a list outside its pool is refused (PF_ERR_ARG). */
int pf_side_components(pf_ctx *, int reset, const pf_bfs_record *records, uint64_t n_records, const uint32_t *pool, uint64_t pool_len,
                       const pf_bfs_record *extra, uint64_t n_extra, const uint32_t *extra_pool, uint64_t extra_pool_len);
int pf_replay_order(pf_ctx *, uint32_t n_classes, uint32_t *order, uint32_t *class_off, uint32_t *labels);

/* The commits on the device.  After pf_side_components over ALL records of a pass (one call with reset != 0, records == NULL:
 * the records pf_bfs_candidates_resident left on the device; `extra` = the traversals the caller walked itself), every component
 * of at most small_limit list entries is committed by one device thread, in record order, into the context's T1 state
 * (MyUnitig bits and partner slots: what pf_call_set_state uploads).  Components above the limit and components holding an
 * `extra` record are left to the caller: pf_replay_device returns how many device-resident records they have (and their list
 * entries), pf_replay_big_fetch hands them over (index[] = their positions among the records, ascending; records with list_off
 * into pool), the caller commits them together with its own records on a zeroed state of its own and passes what it changed to
 * pf_replay_finish: sides[] (2u = plus side of unitig u, 2u + 1 = minus side), the partner slot of each and its per-side flag
 * byte (csrc/host/pf_state_ops.hpp: S_LINK, S_STRICT, S_COMPLEX, S_NON_SUPER).  pf_replay_finish merges the per-side bytes into
 * MyUnitig's and leaves the state resident for pf_superbubble_rows / pf_call_scan; pf_call_get_state copies it to the host. */
/* The candidates the device tiers give up on (traversals of more than 128 vertices: the caller walks them on host cores) are also
 * reported AHEAD of the call's return: *list = cap entries of pinned host memory, zeroed by THIS call (call it once per pass,
 * before the threads that poll the list are started), that the next pf_bfs_candidates* call of this context with a `deferred` array
 * fills while its kernels run -- entry = entrance (oriented vertex) << 32 | (candidate index + 1), in no particular order -- so that
 * the caller's walkers start on the first long traversal while the device is still busy instead of after the call has returned.  A
 * pf_bfs_candidates_resident that repeats itself after a pool overflow writes the list again from its first slot.  An entry is a NOTICE, written when a traversal reaches 48 vertices: the
 * traversal goes on on the device and may yet end there (it is then absent from `deferred`, and the caller drops its walk); every
 * candidate of `deferred` has an entry unless the list ran out of room.  pf_bfs_live_count: the entries the last call wrote (may
 * exceed cap: the surplus was dropped).  cap = 0 switches the list off. */
int pf_bfs_live_deferred(pf_ctx *, uint64_t cap, volatile uint64_t **list);
int pf_bfs_live_count(pf_ctx *, uint64_t *n);
int pf_bfs_candidates_resident(pf_ctx *, uint32_t u0, uint32_t u1, uint64_t *n_records, uint64_t *pool_used, uint32_t *deferred,
                               uint32_t *deferred_entrance, uint64_t deferred_cap, uint64_t *n_deferred);
/* Takes the device buffers the first pf_side_components / pf_replay_device pass over n_records records would take (a hint, to be
 * called beside the load like pf_call_reserve; the adjacency must be resident). */
int pf_find_reserve(pf_ctx *, uint64_t n_records);
int pf_replay_device(pf_ctx *, uint32_t complex_size, uint32_t small_limit, uint64_t *n_big, uint64_t *big_entries);
/* Colored path (CCDBG): the colour gate of the accept commit (src/CCDBG.cpp:2530-2621) reads, per unitig, the set of colours
 * present on every k-mer (full_mask: (n_colors + 63) / 64 words per unitig, as for pf_call_set_colours), UnitigColors::size() with the unitig's own mapping, and how many colours the pair encoding stores as
 * "full" [host|dev].  Set once per graph; pf_side_components and pf_replay_device then apply the colored commits
 * (n_colors == 0: back to the single-sample ones). */
int pf_replay_set_colours(pf_ctx *, uint32_t n_colors, const uint64_t *full_mask, const uint64_t *size_total, const uint32_t *n_full_enc);
int pf_replay_big_fetch(pf_ctx *, uint32_t *index, pf_bfs_record *records, uint32_t *pool);
int pf_replay_finish(pf_ctx *, const uint32_t *sides, const uint32_t *links, const uint8_t *side_flags, uint64_t n_patch);
int pf_call_get_state(pf_ctx *, uint8_t *flags, uint32_t *plus, uint32_t *minus);

/* ---- colored (multi-sample) coverage: reference src/CCDBG.cpp ------------------------------------- */
/* CCDBG::CCDBG (src/CCDBG.cpp:13-43) opens one KMC database per colour.  Here the records of all colours
 * are joined into one HBM table keyed by the stored k-mer with one count per colour, so a k-mer costs one
 * random access whatever the number of colours.  kmers[c] / counts[c]: n[c] records of colour c (exact
 * k-mers as stored, any order); min_count / max_count / both_strands per colour as in pf_upload_counts.
 * Requires pf_upload_graph first (k).  [host|dev per array] */
#define PF_MAX_COLORS_TABLE 1024  /* colours of the joined table, K-COV-C and K-STRCOV-C */
#define PF_MAX_COLORS PF_MAX_COLORS_TABLE /* colours of the resident calling pipeline (pf_call_set_colours) and of the commits on the device
                                     (pf_replay_set_colours): colour sets of (n_colors + 63) / 64 words, a lane per colour of a word in K-SITES */
int pf_upload_counts_colored(pf_ctx *, uint32_t n_colors, const uint64_t *const *kmers, const uint32_t *const *counts,
                             const uint64_t *n, const uint64_t *min_count, const uint64_t *max_count, const int *both_strands);
uint32_t pf_num_colors(const pf_ctx *);
/* CCDBG::readCovUni (src/CCDBG.cpp:123-156) for unitigs [u0, u1) and every colour at once; arrays are
 * colour-major: x[c * (u1 - u0) + (u - u0)].  sum / min / max over the k-mers colour c's database holds,
 * miss = 1 when it lacks one.  readCovUni(u, low, up, c) == (sum / len, true) iff !miss && min > low &&
 * max < up, else (0, false) -- the caller applies its cutoffs.  [host|dev] */
int pf_unitig_cov_colored(pf_ctx *, uint32_t u0, uint32_t u1, uint64_t *sum, uint32_t *min_count, uint32_t *max_count,
                          uint8_t *miss);
/* pf_unitig_cov_colored streams a colour-major per-k-mer coverage SoA that the device joins once when graph and databases are
 * both resident (as pf_unitig_cov does, see there); this is the same function with every k-mer looked up at call time -- the
 * independent check, and the route for a max_count of 2^32 - 1.  [host|dev] */
int pf_unitig_cov_colored_probe(pf_ctx *, uint32_t u0, uint32_t u1, uint64_t *sum, uint32_t *min_count, uint32_t *max_count,
                          uint8_t *miss);
/* CCDBG::readCov(string, low, up, colour) (src/CCDBG.cpp:89-122) for every (string, colour): low / up hold one
 * cutoff per colour; arrays are string-major: x[i * n_colors + c].  ok = 0 (and sum = 0) when a k-mer is
 * missing from colour c's database or a count lies outside (low[c], up[c]).  [host|dev] */
int pf_string_cov_colored(pf_ctx *, const char *text, const uint64_t *str_off, uint32_t n_str, const uint32_t *low,
                          const uint32_t *up, uint64_t *sum, uint8_t *ok);

/* ---- (e) one graph over the GPUs of a node: the exchange step -------------------------------------------------------------
 * the graph, the CSR and the count table are replicated in every GPU's HBM, every rank (one process per GPU) runs
 * findSuperBubble and the owner scan, and rank r aligns, formats and writes its contiguous slice of the bubble list.  What the ranks
 * must tell each other is small: how many bubbles each of them called (var_count numbers bubbles across the whole run, src/CDBG.cpp:
 * 1254-1258), then the sizes of their ten text slabs with their allele histograms and coverage counters -- two all-gathers of a few
 * 64-bit words, over RCCL (xGMI between the GPUs).  No payload crosses ranks.
 * pf_comm_unique_id: called by one rank, the 128 bytes reach the others by any means (the CLI: a socket pair made before fork);
 * pf_comm_init: every rank, on its context (= its GPU); collective.  librccl is loaded here, not at start-up.
 * pf_gather: all[r * n + i] = word i of rank r, n <= 64, host pointers; collective. */
#define PF_COMM_ID_BYTES 128
int pf_comm_unique_id(unsigned char id[PF_COMM_ID_BYTES]);
int pf_comm_init(pf_ctx *, const unsigned char id[PF_COMM_ID_BYTES], int rank, int world);
int pf_gather(pf_ctx *, const uint64_t *mine, uint32_t n, uint64_t *all);
void pf_comm_destroy(pf_ctx *);

/* ---- `PloidyFrost model`: the Gaussian-mixture fit of the allele frequencies (src/GmmModel.cpp, src/Main.cpp:636-692) ----
 * pf_gmm_upload keeps the values (GmmModel::allele_fre, after the reader's frequency filter) in HBM; pf_gmm_fit is
 * GmmModel::resize(gauss) + emIterate() (src/GmmModel.cpp:8-20, 371-385) on them: means fixed at i/(gauss+1), weights and
 * variances re-estimated until the log-likelihood gains at most max_delta or max_iter iterations ran, with emStep's weight
 * thresholds m_thre / n_thre (:318-330).  Outputs: gauss weights / means / variances, the log-likelihood (the sum, not the
 * average) and the number of iterations.  fp64; sums are tree-shaped (the reference's are sequential), same order every run. */
#define PF_GMM_MAX_GAUSS 16
int pf_gmm_upload(pf_ctx *, const double *values, uint64_t n);
uint64_t pf_gmm_count(const pf_ctx *);
int pf_gmm_fit(pf_ctx *, uint32_t gauss, double m_thre, double n_thre, int32_t max_iter, double max_delta, double *weights,
               double *means, double *vars, double *loglik, uint32_t *iterations);

/* ---- pinned host memory for the exchange buffers (optional: pageable memory works, slower) ---- */
int pf_host_alloc(pf_ctx *, size_t bytes, void **out);
/* bytes from device memory to (pinned) host memory, synchronous: for callers that hold device pointers but no HIP runtime */
int pf_fetch(pf_ctx *, void *dst_host, const void *src_dev, uint64_t bytes);
void pf_host_free(pf_ctx *, void *p);

/* ---- self tests ---------------------------------------------------------------------------- */
/* The prefix sums and the flag selection the library runs between its kernels (csrc/pf_scan.hpp), over n pseudo-random elements
 * against sequential host arithmetic: PF_OK, or PF_ERR_HIP with the first difference in pf_last_error. */
int pf_selftest_scan(pf_ctx *, uint64_t n, uint32_t seed);

/* ---- introspection ---------------------------------------------------------------------- */
int pf_device_name(pf_ctx *, char *buf, size_t cap);
/* "domain:bus:device.function" of the context's GPU (the host layer looks up its NUMA node with it) */
int pf_device_pci_bus_id(pf_ctx *, char *buf, size_t cap);
uint64_t pf_table_capacity(const pf_ctx *);
uint64_t pf_num_kmers(const pf_ctx *);

#ifdef __cplusplus
}
#endif
#endif /* PLOIDYFROST_HIP_H_ */
