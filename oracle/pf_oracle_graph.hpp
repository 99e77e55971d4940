// ORACLE (test infrastructure only -- see pf_oracle.h).  Graph + k-mer database substrate:
// a CPU restatement of the *semantics* the hot path relies on from the vendored Bifrost
// graph and KMC reader.  Not a port of either library: the
// graph is a vector of strings plus a hash-joined adjacency table, the database a sorted
// record array.
#pragma once
#include <cstdint>
#include <string>
#include <unordered_map>
#include <vector>

namespace pfo {

constexpr uint32_t NONE = 0xFFFFFFFFu;

inline int base_code(char c) {
    switch (c) {
        case 'A': case 'a': return 0;
        case 'C': case 'c': return 1;
        case 'G': case 'g': return 2;
        case 'T': case 't': return 3;
        default: return -1;
    }
}
inline char comp(char c) {
    switch (c) {
        case 'A': return 'T';
        case 'C': return 'G';
        case 'G': return 'C';
        case 'T': return 'A';
        default: return c;
    }
}
std::string revcomp(const std::string &s);
// 2 bits per base, first base most significant, right aligned (A0 C1 G2 T3:
// bifrost/src/Kmer.cpp set_kmer; KMC/kmc_api/kmer_api.h:156-258 use the same code)
uint64_t pack_kmer(const char *s, int k);
uint64_t rc_kmer(uint64_t x, int k);

// ---- KMC database (K1-K3) -------------------------------------------------------------
struct KmcDb {
    bool loaded = false;
    uint32_t k = 0, mode = 0, counter_size = 0, p = 0;
    uint64_t min_count = 0, max_count = 0, total = 0;
    bool both_strands = true;
    uint32_t version = 0;          // 0 = KMC1 layout, 0x200 = KMC2 layout
    uint32_t sig_len = 0;
    uint64_t single_lut = 0;       // 4^p
    std::vector<uint32_t> sig_map; // KMC2: signature -> bin
    std::vector<uint32_t> norm;    // KMC2: m-mer normalisation table (mmer.h)
    std::vector<uint64_t> lut;     // as the reader indexes it: lut[x], lut[x+1]
    uint64_t lut_words = 0;        // prefix_file_buf_size analogue
    std::vector<uint64_t> suffix;  // total records, (k-p) symbols each, numeric
    std::vector<uint32_t> count;
    std::string err;

    bool load(const std::string &prefix);  // OpenForRA, kmc_file.cpp:27-58
    // CheckKmer (kmc_file.cpp:330-366) + BinarySearch (:1383-1462) on an exact (non
    // canonicalised) k-mer
    bool check(uint64_t kmer, uint32_t &cnt) const;
    // the hot path's composite (CDBG.cpp:38-56): if (!IsKmer(fwd)) reverse(); CheckKmer()
    bool canonical_count(uint64_t fwd, uint32_t &cnt) const;
    uint32_t signature(uint64_t kmer) const;  // CKmerAPI::get_signature (kmer_api.h:653-673)
};

// readCov(string)'s look-up of the k characters at s (src/CDBG.cpp:36-43, src/CCDBG.cpp:96-103): ONE CKmerAPI object per call,
// created as k times 'A'; CKmerAPI::from_string (kmer_api.h:502-510) returns false and leaves the object UNTOUCHED when the k
// characters hold anything but ACGT -- the '-' of an aligned row, which reaches a site string when it takes raw columns up to the
// row's end (substr with a negative count, src/CDBG.cpp:1499).  Such a window is then looked up with what the object held before:
// the previous window's k-mer in the form that was found (reversed or not), or poly-A before any.
struct StringProbe {
    const KmcDb &db;
    int k;
    uint64_t held = 0;  // the object's content, 2 bits per base
    StringProbe(const KmcDb &d, int kk) : db(d), k(kk) {}
    bool count(const char *s, uint32_t &cnt) {
        bool acgt = true;
        for (int i = 0; i < k; ++i) acgt = acgt && (s[i] == 'A' || s[i] == 'C' || s[i] == 'G' || s[i] == 'T');
        if (acgt) held = pack_kmer(s, k);
        uint32_t c0;
        if (!db.check(held, c0)) held = rc_kmer(held, k);  // if (!IsKmer(kmer_object)) kmer_object.reverse();
        return db.check(held, cnt);                        // CheckKmer(kmer_object, count)
    }
};

// ---- graph (G1, G2) -------------------------------------------------------------------
struct Graph {
    int k = 31;
    int g = 23;
    std::vector<std::string> seq;  // reference orientation, index = id-1
    uint64_t n_kmers = 0;
    std::vector<uint32_t> succ, pred;  // [2N][4]
    std::string err;

    bool load_gfa(const std::string &path);  // CompactedDBG::read + UnitigIterator order
    void build_adjacency();                  // NeighborIterator.tcc:25-47 + find(extremities_only)
    uint32_t n() const { return (uint32_t)seq.size(); }
    uint32_t size_bp(uint32_t u) const { return (uint32_t)seq[u].size(); }
    uint32_t len_km(uint32_t u) const { return (uint32_t)seq[u].size() - k + 1; }
    std::string mapped(uint32_t ov) const;  // mappedSequenceToString (UnitigMap.tcc:37-55)
    const uint32_t *succ_row(uint32_t ov) const { return &succ[(size_t)ov * 4]; }
    const uint32_t *pred_row(uint32_t ov) const { return &pred[(size_t)ov * 4]; }
    int out_degree(uint32_t ov) const;
    int in_degree(uint32_t ov) const;
    uint32_t first_succ(uint32_t ov) const;
    uint32_t first_pred(uint32_t ov) const;
};

}  // namespace pfo
