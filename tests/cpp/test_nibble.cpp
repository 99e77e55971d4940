// The packed form of the numeric result streams (ploidyfrost_amd/csrc/pf_nibble.hpp) on its own, no GPU: text over the sixteen
// characters is packed the way the device packs it (restated here: code of a character = its place in the alphabet, first character of a
// pair in the low nibble) and pf::nibble_expand must give the text back, for every length around the sixteen-character steps of the
// expansion and for rows as the path writes them.
#include <cstdio>
#include <random>
#include <string>
#include <vector>

#include "pf_nibble.hpp"

static std::vector<uint8_t> pack(const std::string &t) {
    std::vector<uint8_t> out((t.size() + 1) / 2 + 16, 0);
    for (size_t i = 0; i < t.size(); ++i) {
        const unsigned c = pf::nibble_code(t[i]);
        if (c > 15) { fprintf(stderr, "character %d is outside the alphabet\n", t[i]); exit(2); }
        out[i >> 1] |= (uint8_t)(c << (4 * (i & 1)));
    }
    return out;
}

int main() {
    std::mt19937_64 rng(2025);
    int bad = 0;
    for (unsigned c = 0; c < 16; ++c)
        if (pf::nibble_code(pf::kNibbleAlphabet[c]) != c) ++bad;
    for (const char c : std::string("naifxEN ,:;/")) if (pf::nibble_code(c) != 16) ++bad;   // "nan", "inf", hex, ... : not numbers
    for (size_t n = 0; n <= 200 && !bad; ++n)
        for (int trial = 0; trial < 20; ++trial) {
            std::string t(n, '0');
            for (auto &ch : t) ch = pf::kNibbleAlphabet[rng() % 16];
            const std::vector<uint8_t> p = pack(t);
            std::string back(n + 8, '#');
            pf::nibble_expand(&back[0], p.data(), n);
            if (back.substr(0, n) != t || back.substr(n) != std::string(8, '#')) { fprintf(stderr, "length %zu: expansion differs or writes past its end\n", n); ++bad; break; }
        }
    // rows as K-TEXT writes them
    std::string rows;
    for (int i = 0; i < 5000; ++i) {
        char buf[128];
        snprintf(buf, sizeof buf, "%g\t%g\t%d\t%d\t%d\t%d\t%g\t%d\t\n", (double)(rng() % 100000) / 7.0, 1e-5 * (double)(rng() % 977), (int)(rng() % 2), (int)(rng() % 50), (int)(rng() % 4000000),
                 (int)(rng() % 9), (double)(rng() % 1000) / 999.0, (int)(rng() % 300));
        rows += buf;
    }
    const std::vector<uint8_t> p = pack(rows);
    std::string back(rows.size(), '#');
    for (size_t at = 0; at < rows.size(); at += 128 << 10)   // in the writer's tasks: each starts on an even character
        pf::nibble_expand(&back[at], p.data() + at / 2, std::min<size_t>(128 << 10, rows.size() - at));
    if (back != rows) { fprintf(stderr, "rows: expansion differs\n"); ++bad; }
    printf("%s\n", bad ? "FAILED" : "ok");
    return bad ? 1 : 0;
}
