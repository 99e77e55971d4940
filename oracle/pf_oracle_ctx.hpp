// ORACLE (test infrastructure only -- see pf_oracle.h).  Context and helpers shared by the single-sample
// restatement (pf_oracle.cpp, reference src/CDBG.cpp) and the colored one (pf_oracle_colored.cpp,
// reference src/CCDBG.cpp).
#pragma once
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <unordered_map>
#include <vector>

#include "pf_oracle.h"
#include "pf_oracle_graph.hpp"

namespace pfo_int {

// MyUnitig flag bits (src/MyUnitig.hpp:37-46, 52-85, 97-130)
enum : uint8_t {
    F_PLUS_OPEN = 0x01,
    F_MINUS_OPEN = 0x02,
    F_NON_SUPER = 0x04,
    F_STRICT_MINUS = 0x08,
    F_STRICT_PLUS = 0x10,
    F_COMPLEX_MINUS = 0x20,
    F_COMPLEX_PLUS = 0x40,
};

// What the colored path asks of Bifrost's per-unitig colour sets (bifrost/src/ColorSet.cpp), loaded from the
// dump the real library produced for the fixture (oracle/ref_colors_dump.cpp).
struct ColorData {
    uint32_t n_colors = 0;
    std::vector<std::string> names;
    // per unitig, per colour: one '0'/'1' per k-mer, reference orientation
    std::vector<std::vector<std::string>> bits;
    std::vector<uint64_t> size_total;  // UnitigColors::size(um) with the unitig's own mapping
    std::vector<uint32_t> n_full_enc;  // colours the encoding keeps as "full" (ptrUnitigColors, ColorSet.cpp:902-907)
    std::string err;

    bool load_dump(const std::string &path, const pfo::Graph &g);
    // UnitigColors::contains(um, colour) (ColorSet.cpp:776-823): colour on every k-mer of [dist, dist+len)
    bool contains(uint32_t u, uint32_t colour, uint32_t dist, uint32_t len) const {
        const std::string &b = bits[u][colour];
        for (uint32_t i = dist; i < dist + len; ++i)
            if (b[i] != '1') return false;
        return true;
    }
    bool full(uint32_t u, uint32_t colour) const { return contains(u, colour, 0, (uint32_t)bits[u][colour].size()); }
    // UnitigColors::size(um) of unitig u's set, evaluated with a mapping of km_of k-mers (ColorSet.cpp:898-927):
    // only the ptrUnitigColors encoding looks at the mapping
    uint64_t size_with(uint32_t u, uint64_t km_own, uint64_t km_of) const {
        return size_total[u] - (uint64_t)n_full_enc[u] * km_own + (uint64_t)n_full_enc[u] * km_of;
    }
};

// CompactedDBG::findUnitig(const char*, pos, len) (bifrost/src/CompactedDBG.tcc:3815-3837) needs every k-mer
struct KmerIndex {
    std::unordered_map<uint64_t, uint64_t> where;  // canonical k-mer -> (u << 32 | pos), as stored
    void build(const pfo::Graph &g);
    // longest match of s from its first k-mer along one unitig: unitig, forward dist, #k-mers; false = not found
    bool find_unitig(const pfo::Graph &g, const std::string &s, uint32_t &u, uint32_t &dist, uint32_t &len) const;
};

struct Traversal {
    int outcome = PFO_BFS_NONE;
    uint32_t exit_ov = pfo::NONE;
    std::vector<uint32_t> seen;  // vec_km_seen
    std::vector<uint32_t> cyc;   // cycle_unitig_set, insertion order, deduplicated
    bool flag_cycle = false, flag_tip = false;
};

extern std::string g_err;

// Cells of the *cov.txt files whose value is UNDEFINED in the reference (pfo::indel_len_at): when the environment variable
// PFO_UB_LOG names a file, one line "<bi|tri|tetra|penta>cov\t<1-based line>" per such cell is appended to it, so a test can hold
// every other byte of the reference binary's output to this restatement and leave exactly these cells out.
struct UbLog {
    FILE *f = nullptr;
    UbLog() {
        const char *p = getenv("PFO_UB_LOG");
        if (p && *p) f = fopen(p, "a");
    }
    ~UbLog() { if (f) fclose(f); }
    void cell(int arity_idx, uint64_t line) {
        static const char *name[4] = {"bicov", "tricov", "tetracov", "pentacov"};
        if (f) fprintf(f, "%s\t%llu\n", name[arity_idx], (unsigned long long)line);
    }
};

}  // namespace pfo_int

struct pfo_ctx {
    pfo::Graph g;
    pfo::KmcDb db;
    // colored path (src/CCDBG.cpp): one database per colour, colour sets, k-mer index
    bool colored = false;
    std::vector<pfo::KmcDb> dbs;
    pfo_int::ColorData col;
    pfo_int::KmerIndex kidx;
    // MyUnitig state: partner ids are 1-based, 0 = NULL
    std::vector<uint8_t> flags;
    std::vector<uint32_t> plus, minus;
    uint32_t complex_size = 8;

    uint32_t id(uint32_t ov) const { return (ov >> 1) + 1; }
    bool strand(uint32_t ov) const { return (ov & 1) == 0; }

    // --- MyUnitig mutators on unitig index d
    void set_plus_self(uint32_t d) { plus[d] = d + 1; flags[d] &= 0xFE; }
    void set_minus_self(uint32_t d) { minus[d] = d + 1; flags[d] &= 0xFD; }
    void set_side_self(uint32_t d, bool plus_side) { plus_side ? set_plus_self(d) : set_minus_self(d); }
    uint32_t &side(uint32_t d, bool plus_side) { return plus_side ? plus[d] : minus[d]; }
    // "if (ex->get_plus() == me) ex->set_plus_self(); else ex->set_minus_self();"
    void release_partner(uint32_t ex, uint32_t me) {
        if (plus[ex] == me + 1) set_plus_self(ex); else set_minus_self(ex);
    }
    // interior treatment shared by the commits (e.g. CDBG.cpp:800-826)
    void poison(uint32_t d) {
        for (int s = 0; s < 2; ++s) {
            bool ps = (s == 0);
            uint32_t p = side(d, ps);
            if (p != 0 && p != d + 1) release_partner(p - 1, d);
            set_side_self(d, ps);
        }
        flags[d] |= pfo_int::F_NON_SUPER;
    }
    // the colored cycle commit's variant (CCDBG.cpp:2351-2384): a side is self-marked only when it held a real partner
    void poison_linked_only(uint32_t d) {
        for (int s = 0; s < 2; ++s) {
            bool ps = (s == 0);
            uint32_t p = side(d, ps);
            if (p != 0 && p != d + 1) {
                release_partner(p - 1, d);
                set_side_self(d, ps);
            }
        }
        flags[d] |= pfo_int::F_NON_SUPER;
    }
};

namespace pfo_int {
Traversal traverse(const pfo_ctx &c, uint32_t s);
void commit_reject(pfo_ctx &c, const Traversal &t, uint32_t s);
void commit_no_exit(pfo_ctx &c, const Traversal &t, uint32_t s);
void sort_branching(std::vector<std::string> &v, int low, int high);
std::string strip_gaps(const std::string &s);
bool ensure_dir(const std::string &d);
// colored twins (pf_oracle_colored.cpp)
void commit_cycle_exit_colored(pfo_ctx &c, const Traversal &t, uint32_t s);
void commit_accept_colored(pfo_ctx &c, const Traversal &t, uint32_t s);
}  // namespace pfo_int
