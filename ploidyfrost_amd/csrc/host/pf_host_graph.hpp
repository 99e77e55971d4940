// Host-side substrate of the product: the unitig set of a Bifrost GFA in the reference's id
// order, 2-bit packed for the device, and the records of a KMC database.
//
// Replaces, for this path only, what the reference gets from CompactedDBG<MyUnitig>::read
// (bifrost/src/CompactedDBG.tcc:823-960, 7888-7907) and CKMCFile::OpenForRA
// (KMC/kmc_api/kmc_file.cpp:27-58, 185-302): no minimizer index, no Bloom filters -- neighbour
// discovery and k-mer lookups happen on the GPU (include/ploidyfrost_hip.h).
#pragma once
#include <cstdint>
#include <string>
#include <string_view>
#include <vector>

namespace pfh {

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
    bool load_gfa(const std::string &path, std::string &err);
    // builds from already ordered sequences (tests, generators)
    void from_sequences(const std::vector<std::string> &seqs, int k_);
    void pack();
};

struct KmcRecords {
    uint32_t k = 0, counter_size = 0, lut_prefix_len = 0, mode = 0;
    uint64_t min_count = 0, max_count = 0, total = 0;
    bool both_strands = true;
    std::vector<uint64_t> kmers;   // exact k-mers as stored, ascending inside each prefix
    std::vector<uint32_t> counts;
    // Parses prefix.kmc_pre / prefix.kmc_suf (KMC1 layout, kmc_file.cpp:246-299).
    bool load(const std::string &prefix, std::string &err);
};

}  // namespace pfh
