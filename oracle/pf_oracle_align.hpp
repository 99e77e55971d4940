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
