// ORACLE (test infrastructure only -- see pf_oracle.h).  CPU restatement of
// SeqAlign (reference: src/SeqAlign.cpp, src/SeqAlign.hpp).
#pragma once
#include <cstdint>
#include <string>
#include <vector>

namespace pfo {

// One co-optimal pairwise alignment kept by the traceback (AlignUnit, SeqAlign.hpp:30-68;
// only the fields that influence results are kept).
struct Aln {
    std::string a, b;               // aligned rows ('-' = gap)
    std::vector<uint32_t> gap_pos;  // rows of A at which gaps were opened, traceback order
    long score = 0;
    uint32_t n_pos = 0;             // variant positions (snp + gap openings)
    uint32_t indel = 0;
};

struct AlignResult {
    std::vector<std::string> rows;  // empty => no alignment survived (bubble skipped)
    std::vector<uint32_t> snp_pos, indel_pos, indel_len;
    std::vector<std::vector<uint16_t>> partition;  // [col][row]
};

// indel_len[i] as the callers print it (src/CDBG.cpp:1310, 1565; src/CCDBG.cpp:3034, 3315, 3450).  An indel run that is still
// open at the last column of the alignment never has its length pushed (the column loop of compareStrPair, SeqAlign.cpp:56-157,
// closes a run only on a later column), so for that last indel site the reference's `indel_len[indel - 1]` reads one element past
// the vector: heap garbage that changes from run to run (ASLR), or a null dereference (SIGSEGV) when the vector is empty.  It can
// only happen with gap-friendly scores (rows ending in gaps).  UNDEFINED in the reference; this restatement and the product
// define it as the length of the open run (columns - its first column) and flag the cell (*ub) so the tests can mask it.
inline uint32_t indel_len_at(const AlignResult &ar, uint32_t i, bool *ub) {
    if (i < ar.indel_len.size()) return ar.indel_len[i];
    *ub = true;
    const uint32_t cols = (uint32_t)ar.partition.size();
    return i < ar.indel_pos.size() ? cols - ar.indel_pos[i] : 0;
}

struct Scoring {
    double match = 2, mismatch = -1, gap = -3;
};

// needlemanWunch + traceback (SeqAlign.cpp:480-549, 306-478)
std::vector<Aln> pairwise_all_optimal(const Scoring &sc, const std::string &A, const std::string &B);
// variantAnalyze (SeqAlign.cpp:237-305)
Aln score_rows(const Scoring &sc, const std::string &A, const std::string &B);
// long AlignUnit::operator-(x) (SeqAlign.hpp:43-67)
long aln_minus(const Aln &l, const Aln &r);
// SequenceAlignment (SeqAlign.cpp:550-640) incl. compareStrPair (:8-236)
AlignResult align_paths(const Scoring &sc, const std::vector<std::string> &strs);

}  // namespace pfo
