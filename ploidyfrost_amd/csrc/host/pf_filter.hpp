// `ploidyfrost filter` / `ploidyfrost filter-multi`: the row predicates of the reference's R scripts over the *cov.txt files of
// the path (reference script/Filter.R:1-159, script/Filter-multi.R:1-186) -- the step between
// PloidyEstimation's files and `PloidyFrost model`.  Same option letters, same predicates, same four filtered tables and
// <outprefix>_allele_frequency.txt, written the way R's read.table / write.table / round would (column types, 15 significant
// digits, fixed or scientific by width).
//
// PARITY UNPINNED: R is not installed in the build image, so no output of the reference's scripts exists to hold this to; the
// expected rows of the tests are worked out by hand from the scripts' text (tests/test_filter_cpu.py).  One divergence is closed by
// refusal rather than emulation: a cell that is not a finite decimal number ("nan", "-nan", "inf", "NA", hex floats) -- R would read
// some of these as NA and write rows of NAs -- ends the run with an error that names the cell.
#pragma once
#include <string>
#include <vector>

namespace pfh {

struct FilterOptions {            // optparse defaults of Filter.R:5-28 / Filter-multi.R:5-32
    bool simple = false;          // -S  only rows of strict bubbles (isStrict == 1)
    std::string outprefix = "filtered", inprefix = "input";   // -o, -i
    long low = 0, up = 10000;     // -l, -u  every allele coverage in (low, up)
    bool indel = false;           // -I  "filter indel": keeps VarType == 0
    bool snp = false;             // -P  "filter snp":   keeps VarType > 0
    long num = 10000;             // -n  VarNum < num
    long distance = -1;           // -d  VarDis > distance
    long size = 10000;            // -s  VarType < size
    double frequency = 0.05;      // -q  frequencies kept in (q, 1 - q)
    // Filter-multi.R only
    bool multi = false;
    long color = -1;              // -c  rows of this colour only (< 0: all)
    double cramer = 0.0;          // -v  Cramer's V > cramer
};

// 0 = done (also after the scripts' own early exits, which leave with status 0 and a message); 1 = an R error (message in err)
int run_filter(const FilterOptions &opt, std::string &messages, std::string &err);

// R's rendering of one double in write.table (formatReal with 15 significant digits + EncodeReal0, src/main/format.c): the
// fewest significant digits (<= 15) that give the value, fixed notation unless scientific is narrower
std::string r_format_double(double x);
// round(x, 7) of Filter.R:159
double r_round7(double x);

int filter_main(int argc, char **argv, bool multi);

}  // namespace pfh
