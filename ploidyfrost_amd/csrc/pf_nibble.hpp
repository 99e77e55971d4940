// The packed form of the nine numeric result streams (K-NIB, pf_call_set_numeric_packed): their text is written in sixteen
// characters, so it crosses PCIe at four bits a character -- code = the character's place in kNibbleAlphabet, first character of a
// pair in the low nibble of its byte.  The device packs (k_text_nibbles, pf_call_text.hip); the host expands, here.
#pragma once
#include <stdint.h>
#include <string.h>

namespace pf {

constexpr char kNibbleAlphabet[17] = "0123456789.\t\n-e+";

// the code of a character, 16 for one outside the alphabet (the stream then travels as text)
inline unsigned nibble_code(char c) {
    for (unsigned i = 0; i < 16; ++i)
        if (kNibbleAlphabet[i] == c) return i;
    return 16;
}

// n characters from (n + 1) / 2 bytes of nibbles
inline void nibble_expand(char *dst, const uint8_t *src, uint64_t n) {
    static const struct Lut {
        uint16_t two[256];   // the two characters of a byte, first one in the low byte (little-endian store)
        Lut() {
            for (int b = 0; b < 256; ++b) two[b] = (uint16_t)((uint8_t)kNibbleAlphabet[b & 15] | ((uint16_t)(uint8_t)kNibbleAlphabet[b >> 4] << 8));
        }
    } lut;
    uint64_t i = 0;
    for (; i + 16 <= n; i += 16) {   // eight bytes of nibbles -> sixteen characters
        uint64_t in;
        memcpy(&in, src + (i >> 1), 8);
        uint16_t w[8];
        for (int x = 0; x < 8; ++x) w[x] = lut.two[(in >> (8 * x)) & 0xFF];
        memcpy(dst + i, w, 16);
    }
    for (; i + 2 <= n; i += 2) {
        const uint16_t w = lut.two[src[i >> 1]];
        memcpy(dst + i, &w, 2);
    }
    if (i < n) dst[i] = (char)(lut.two[src[i >> 1]] & 0xFF);
}

}  // namespace pf
