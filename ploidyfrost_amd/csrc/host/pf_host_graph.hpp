// Host-side substrate of the product: the unitig set of a Bifrost GFA in the reference's id
// order, 2-bit packed for the device, and the records of a KMC database.
//
// Replaces, for this path only, what the reference gets from CompactedDBG<MyUnitig>::read
// (bifrost/src/CompactedDBG.tcc:823-960, 7888-7907) and CKMCFile::OpenForRA
// (KMC/kmc_api/kmc_file.cpp:27-58, 185-302): no minimizer index, no Bloom filters -- neighbour
// discovery and k-mer lookups happen on the GPU (include/ploidyfrost_hip.h).
#pragma once
#include <cstdint>
#include <memory>
#include <string>
#include <string_view>
#include <vector>

struct pf_ctx;

namespace pfh {

struct GfaSource;   // the mapped GFA file of a device ingest (pf_host_graph.cpp)

struct UnitigSet {
    int k = 31;
    int g = 23;
    // reference-orientation sequences, flat: unitig u = text[off[u] .. off[u+1])
    std::vector<char> text;
    std::vector<uint64_t> off;
    // device layout (pf_upload_graph)
    std::vector<uint64_t> words, word_off;
    std::vector<uint32_t> len_bp;
    uint64_t n_kmers = 0;
    uint64_t n_short = 0;
    // Bifrost's per-segment "DA:Z:<n>" tag (which hash seed places the unitig's colour set,
    // bifrost/src/ColoredCDBG.tcc:496-533); -1 where a segment has none.  Empty when no segment has one.
    std::vector<int16_t> da_tag;
    // k-length unitigs Bifrost files as "abundant" k-mers (pf_host_minz.hpp): they are the last n_abundant unitigs
    uint64_t n_abundant = 0;
    uint32_t numbering_replays = 0;
    // load_gfa(..., defer_numbering = true) leaves the abundant k-mers undecided: the unitigs stand in provisional order
    // (long, then k-length, in file order -- final unless a minimizer bucket can get crowded) and file_rank[u] is the S-line
    // rank of unitig u.  The owner of the device decides with K-MINZ and calls finish_numbering() only if it has to.
    bool numbering_deferred = false;
    std::vector<uint32_t> file_rank;
    // host replay (pf_host_minz.hpp); repacks when the order changed.  dev_counters / dev_flags: pf_minimizer_replay_inputs' two
    // arrays (flags per unitig in the current order), sparing the replay its two passes over every unitig
    void finish_numbering(std::vector<uint8_t> *counters = nullptr, const uint8_t *dev_counters = nullptr, const uint8_t *dev_flags = nullptr, uint64_t dev_counters_len = 0);
    void numbering_settled() { numbering_deferred = false; std::vector<uint32_t>().swap(file_rank); }

    uint32_t n() const { return (uint32_t)len_bp.size(); }
    std::string_view seq(uint32_t u) const { return std::string_view(text.data() + off[u], len_bp[u]); }
    uint32_t size_bp(uint32_t u) const { return len_bp[u]; }
    uint32_t len_km(uint32_t u) const { return len_bp[u] - (uint32_t)k + 1; }
    // oriented sequence of ov = 2u + (strand ? 0 : 1), appended to dst
    void append_mapped(uint32_t ov, std::string &dst) const;
    std::string mapped(uint32_t ov) const { std::string s; append_mapped(ov, s); return s; }

    // Loads the S-lines of a GFA 1/2 file in the reference's unitig order: long unitigs
    // (length > k) in file order, then k-length ones, each stored as min(seq, revcomp)
    // (SURVEY.md 3.1), the abundant ones last (pf_host_minz.hpp).  A last line without '\n' is ignored, as in bifrost/src/GFA_Parser.cpp:486.
    bool load_gfa(const std::string &path, std::string &err, bool defer_numbering = false);
    // Device ingest (K-GFA, csrc/pf_gfa.hip): open_gfa maps the file and reads its header line; the owner of the device context
    // calls ingest_on_device, which parses and packs the segments THERE (the graph is resident afterwards: no pf_upload_graph)
    // and keeps on the host what it needs per unitig -- length, place in the file, file rank, DA tag.  `text` (and with it
    // seq(), append_mapped(), words) is made from the mapped file only when somebody asks: ensure_text().  Same unitig order,
    // same error messages as load_gfa(defer_numbering = true).
    bool open_gfa(const std::string &path, std::string &err);
    bool ingest_pending() const { return src_ != nullptr && !ingested_; }
    uint64_t estimated_unitigs() const;   // before the ingest: from the size of the file
    // parse_on_device: the parse + pack half alone (pf_gfa_parse: may run while another thread feeds the count table to the same
    // context); ingest_on_device then only adopts the packed arrays as the context's graph and fetches the segment table
    int parse_on_device(pf_ctx *ctx);
    int ingest_on_device(pf_ctx *ctx, std::string &err);
    bool on_device() const { return ingested_; }
    void ensure_text();
    // the sequence of unitig u (len_bp[u] characters) copied to dst -- from `text` when that exists, else straight from the mapped
    // file (what ensure_text would store): a caller that only streams the sequences out need not build `text` at all
    void copy_seq(uint32_t u, char *dst) const;
    // builds from already ordered sequences (tests, generators)
    void from_sequences(const std::vector<std::string> &seqs, int k_);
    void pack();

private:
    void text_from_file(uint32_t u, char *dst) const;
    std::shared_ptr<GfaSource> src_;
    bool ingested_ = false;
    pf_ctx *parsed_on_ = nullptr;   // parse_on_device ran on this context
    int parse_status_ = 0;
    uint32_t parsed_n_ = 0, parsed_short_ = 0;
};

// A KMC database opened for the device: header and prefix table parsed on the host (a few KB .. MB), the record area of
// <db>.kmc_suf mapped and handed to pf_kmc_decode as it lies on disk (KMC1 layout kmc_file.cpp:246-299, KMC2 :196-245).
struct KmcRecords {
    uint32_t k = 0, counter_size = 0, lut_prefix_len = 0, mode = 0, suffix_bytes = 0;
    uint64_t min_count = 0, max_count = 0, total = 0;
    bool both_strands = true;
    std::vector<uint64_t> lut;          // first record of every prefix-table entry, plus lut[n_lut()] = total
    const uint8_t *records = nullptr;   // total * (suffix_bytes + counter_size) bytes inside the mapped file
    uint64_t n_lut() const { return lut.empty() ? 0 : lut.size() - 1; }

    KmcRecords() = default;
    KmcRecords(const KmcRecords &) = delete;
    KmcRecords &operator=(const KmcRecords &) = delete;
    ~KmcRecords();
    bool load(const std::string &prefix, std::string &err);

private:
    void *map_ = nullptr;
    size_t map_n_ = 0;
};

}  // namespace pfh
