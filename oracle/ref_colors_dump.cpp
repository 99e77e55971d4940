// TEST INFRASTRUCTURE ONLY.  A small program of this repository that links against the reference's
// vendored Bifrost (built from /root/reference by oracle/Makefile.ref, objects under oracle/_ref/obj)
// and prints, through Bifrost's public API only, how the real library interprets a colored graph:
// for every unitig in iterator order (the order CCDBG::setUnitigId numbers them, src/CCDBG.cpp)
//
//   id \t head k-mer \t #k-mers \t size(um) \t n_full_enc \t <colour 0> , <colour 1> , ...
//
//   size(um)     UnitigColors::size(const UnitigMapBase&)  (bifrost/src/ColorSet.cpp:898-927)
//   n_full_enc   number of colours the on-disk encoding stores as "full" (the ptrUnitigColors form,
//                ColorSet.cpp:902-907): recovered as (size(um) - size(um with size=k)) / (#k-mers - 1);
//                0 for every other encoding.  It matters because CCDBG.cpp:2552 calls size() of the exit's
//                colour set with the *entrance's* mapping.
//   <colour c>   'F' = on every k-mer, 'E' = on none, else one '0'/'1' per k-mer (reference orientation)
//
// The dump is data about a fixture (committed under tests/golden/<case>/colors.txt); it pins both the
// oracle's colour semantics and the product's own .bfg_colors reader to the real Bifrost.
#include <ColoredCDBG.hpp>

#include <iostream>
#include <string>

int main(int argc, char **argv) {
    if (argc < 3) {
        std::cerr << "usage: colors_dump graph.gfa graph.bfg_colors" << std::endl;
        return 2;
    }
    ColoredCDBG<> g;
    if (!g.read(argv[1], argv[2], 1, false)) {
        std::cerr << "colors_dump: cannot read the colored graph" << std::endl;
        return 1;
    }
    const size_t C = g.getNbColors();
    const int k = g.getK();
    std::cout << "#colors\t" << C << "\tk\t" << k << "\tunitigs\t" << g.size() << "\n";
    for (const auto &name : g.getColorNames()) std::cout << "#name\t" << name << "\n";
    size_t id = 0;
    for (const auto &u : g) {
        ++id;
        const UnitigColors *uc = u.getData()->getUnitigColors(u);
        const size_t km = u.size - k + 1;
        const size_t sz = uc->size(u);
        size_t n_full = 0;
        if (km > 1) {
            UnitigColorMap<void> one(u);
            one.size = k;
            n_full = (sz - uc->size(one)) / (km - 1);
        }
        std::cout << id << "\t" << u.referenceUnitigToString().substr(0, k) << "\t" << km << "\t" << sz << "\t" << n_full << "\t";
        for (size_t c = 0; c < C; ++c) {
            std::string bits(km, '0');
            size_t n = 0;
            for (size_t p = 0; p < km; ++p) {
                UnitigColorMap<void> one(u);
                one.dist = p;
                one.len = 1;
                if (uc->contains(one, c)) { bits[p] = '1'; ++n; }
            }
            if (c) std::cout << ",";
            if (n == km) std::cout << "F";
            else if (n == 0) std::cout << "E";
            else std::cout << bits;
        }
        std::cout << "\n";
    }
    return 0;
}
