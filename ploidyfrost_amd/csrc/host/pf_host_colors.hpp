// Per-unitig colour sets of a Bifrost colored graph, read from its .bfg_colors file.
//
// Replaces, for the CCDBG path only, what the reference gets from ColoredCDBG<MyUnitig>::read
// (bifrost/src/ColoredCDBG.tcc:428-600 -> DataStorage<U>::read, bifrost/src/DataStorage.tcc:790-1043 ->
// UnitigColors::read, bifrost/src/ColorSet.cpp:1228-1283) and answers the three questions src/CCDBG.cpp
// asks of a colour set: UnitigColors::contains(um, colour) (ColorSet.cpp:776-823), UnitigColors::size(um)
// (:898-927) and getNbColors().  File layout understood: format versions 1 and 2; the colour-set
// encodings TinyBitmap (bitmap / list / run-length modes, bifrost/src/TinyBitmap.cpp:825-880), 61-bit
// vector, single integer, Roaring bitmap (portable serialisation) and the {full colours, rest} pair;
// hash-placed sets (Kmer::hash = wyhash of the head k-mer, bifrost/src/Kmer.hpp:120-123, seeds from the
// file, "DA:Z:<n>" tag of the GFA segment) and the overflow table.  Shared colour sets are refused:
// Bifrost 1.0.6 never writes them (their construction is commented out, ColoredCDBG.tcc:1188-1236).
#pragma once
#include <cstdint>
#include <string>
#include <unordered_map>
#include <vector>

#include "pf_host_graph.hpp"

namespace pfh {

struct ColorSets {
    uint32_t n_colors = 0;
    std::vector<std::string> names;
    // per unitig (graph id order)
    uint32_t words = 1;                // 64-bit words per unitig in the two masks below: (n_colors + 63) / 64
    std::vector<uint64_t> full_mask;   // [u * words + c / 64] bit c % 64: colour c on every k-mer
    std::vector<uint64_t> any_mask;    // ... colour c on at least one k-mer
    std::vector<uint64_t> size_total;  // UnitigColors::size(um) with the unitig's own mapping
    std::vector<uint32_t> n_full_enc;  // colours the file's encoding stores as "full" (the pair form); see size_with
    // colours present on part of a unitig: one bit per k-mer, reference orientation
    struct Partial { uint32_t colour; std::vector<uint64_t> bits; };
    std::unordered_map<uint32_t, std::vector<Partial>> partial;

    // (the joined count table of the device holds as many, PF_MAX_COLORS_TABLE; graphs of more than 62 colours are called by the
    // host-threaded pipeline and committed on host threads: the resident pipeline keeps a colour set in one 64-bit register)
    static constexpr uint32_t kMaxColors = 1024;

    // threads: worker threads for decoding (the result does not depend on it)
    bool load(const std::string &path, const UnitigSet &g, unsigned threads, std::string &err);

    // UnitigColors::contains(um, colour): colour on every k-mer of [dist, dist + len)
    bool contains(uint32_t u, uint32_t colour, uint32_t dist, uint32_t len) const;
    bool full(uint32_t u, uint32_t colour) const { return (full_mask[(size_t)u * words + (colour >> 6)] >> (colour & 63)) & 1; }
    bool any(uint32_t u, uint32_t colour) const { return (any_mask[(size_t)u * words + (colour >> 6)] >> (colour & 63)) & 1; }
    uint32_t n_full(uint32_t u) const {
        uint32_t n = 0;
        for (uint32_t w = 0; w < words; ++w) n += (uint32_t)__builtin_popcountll(full_mask[(size_t)u * words + w]);
        return n;
    }
    // UnitigColors::size(um) of unitig u's set evaluated with a mapping of km_of k-mers: only the pair
    // encoding looks at the mapping (ColorSet.cpp:902-907) -- CCDBG.cpp:2552 passes the *entrance's*
    // mapping to the exit's set
    uint64_t size_with(uint32_t u, uint64_t km_own, uint64_t km_of) const {
        return size_total[u] - (uint64_t)n_full_enc[u] * km_own + (uint64_t)n_full_enc[u] * km_of;
    }
};

// Kmer::hash(seed) of Bifrost built with MAX_KMER_SIZE=32 (the reference's build, CMakeLists.txt): wyhash
// (final version 3, default secret) over the 8 bytes of the left-aligned 2-bit k-mer
uint64_t bifrost_kmer_hash(uint64_t left_aligned_kmer, uint64_t seed);

}  // namespace pfh
