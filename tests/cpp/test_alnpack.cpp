// The packed form of alignseq.txt (ploidyfrost_amd/csrc/pf_alnpack.hpp) on its own, no GPU: random bubbles are packed the way the
// device packs them (restated here from the layout in the header's text) and pf::alnpack_expand must give the rows
// `var_count \t strict \t entrance \t exit \t row \n` (reference src/CDBG.cpp:1259, 1428) byte for byte, group by group.
#include <cstdio>
#include <random>
#include <string>
#include <vector>

#include "pf_alnpack.hpp"

int main() {
    std::mt19937_64 rng(12345);
    const char letters[5] = {'-', 'A', 'C', 'G', 'T'};
    for (int trial = 0; trial < 40; ++trial) {
        const uint64_t nb = 1 + rng() % 3000;
        std::string text;
        std::vector<uint8_t> rec;
        std::vector<uint64_t> text_off, rec_off;
        for (uint64_t j = 0; j < nb; ++j) {
            if (j % pf::ALNPACK_GROUP == 0) { text_off.push_back(text.size()); rec_off.push_back(rec.size()); }
            if (rng() % 7 == 0) continue;   // a bubble whose alignment left no rows
            const uint64_t vc = (trial == 3 ? (1ull << 33) : 0) + j + 1;
            const uint32_t ent = 1 + (uint32_t)(rng() % 4000000000u), ext = 1 + (uint32_t)(rng() % 4000000000u);
            const uint32_t L = 1 + (uint32_t)(rng() % (trial % 5 == 0 ? 700 : 90)), R = 2 + (uint32_t)(rng() % (trial % 9 == 0 ? 40 : 4));
            const bool strict = rng() & 1;
            uint8_t h[24];
            const uint32_t rr = R | (strict ? 0x80000000u : 0u);
            memcpy(h, &vc, 8); memcpy(h + 8, &ent, 4); memcpy(h + 12, &ext, 4); memcpy(h + 16, &L, 4); memcpy(h + 20, &rr, 4);
            rec.insert(rec.end(), h, h + 24);
            for (uint32_t r = 0; r < R; ++r) {
                std::string row(L, 'A');
                for (auto &c : row) c = letters[rng() % 5];
                text += std::to_string(vc) + "\t" + (strict ? "1" : "0") + "\t" + std::to_string(ent) + "\t" + std::to_string(ext) + "\t" + row + "\n";
                std::vector<uint8_t> bytes(pf::alnpack_row_bytes(L), 0);
                for (uint32_t i = 0; i < L; ++i) {
                    const uint32_t bit = 3 * i, code = pf::alnpack_code(row[i]);
                    for (int b = 0; b < 3; ++b)
                        if (code >> b & 1) bytes[(bit + b) >> 3] |= (uint8_t)(1u << ((bit + b) & 7));
                }
                rec.insert(rec.end(), bytes.begin(), bytes.end());
            }
        }
        text_off.push_back(text.size());
        rec_off.push_back(rec.size());
        const uint64_t n_groups = (nb + pf::ALNPACK_GROUP - 1) / pf::ALNPACK_GROUP, gb = pf::ALNPACK_GROUP;
        if (text_off.size() != n_groups + 1) { printf("index size\n"); return 1; }
        std::vector<uint8_t> piece(pf::alnpack_index_bytes(nb) + rec.size());
        memcpy(piece.data(), &n_groups, 8);
        memcpy(piece.data() + 8, &gb, 8);
        for (uint64_t g = 0; g <= n_groups; ++g) { memcpy(piece.data() + 16 + 16 * g, &text_off[g], 8); memcpy(piece.data() + 24 + 16 * g, &rec_off[g], 8); }
        memcpy(piece.data() + pf::alnpack_index_bytes(nb), rec.data(), rec.size());
        pf::AlnPackPiece p;
        if (!p.parse(piece.data(), piece.size()) || p.n_groups != n_groups || p.records != piece.data() + pf::alnpack_index_bytes(nb)) { printf("parse\n"); return 1; }
        std::string got(text.size(), '?');
        for (uint64_t g = n_groups; g-- > 0;) {   // (any order: every group knows where its text goes)
            uint64_t t0, r0, t1, r1;
            p.entry(g, t0, r0);
            p.entry(g + 1, t1, r1);
            char *end = pf::alnpack_expand(p.records + r0, p.records + r1, &got[0] + t0, &got[0] + t1);
            if (end != &got[0] + t1) { printf("group %llu ends at %lld, not %llu\n", (unsigned long long)g, (long long)(end - &got[0]), (unsigned long long)t1); return 1; }
        }
        if (got != text) { printf("text differs in trial %d\n", trial); return 1; }
        // a destination one byte short, and records cut one byte short: refused before anything is written beyond the bounds
        if (n_groups) {
            uint64_t t0, r0, t1, r1;
            p.entry(0, t0, r0);
            p.entry(1, t1, r1);
            if (t1 > t0 && r1 > r0) {
                std::string small(t1 - t0, '?');
                if (pf::alnpack_expand(p.records + r0, p.records + r1, &small[0], &small[0] + small.size() - 1) != nullptr) { printf("short destination accepted\n"); return 1; }
                if (pf::alnpack_expand(p.records + r0, p.records + r1 - 1, &small[0], &small[0] + small.size()) != nullptr) { printf("cut records accepted\n"); return 1; }
            }
        }
    }
    printf("ok\n");
    return 0;
}
