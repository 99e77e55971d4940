// The rows of <outpre>_alignseq.txt as they leave the device when pf_call_set_alignseq_packed is on.
//
// alignseq.txt is more than half of a pass's result text (245 of 438 MB at BASELINE.json configs[2]) and the text crosses PCIe at
// 55 GB/s: the pass ends when the last piece has arrived.  Its rows are `var_count \t strict \t entrance \t exit \t <aligned row> \n`
// (reference src/CDBG.cpp:1259, 1428): every row of a bubble repeats the same four numbers, and the aligned row is written in a
// five-letter alphabet.  Packed, a bubble is one header and its rows at 3 bits per character (2.5 x smaller at k = 25); the host
// threads that copy a piece into the mapped file -- they touch every byte of it anyway -- write the text out of it.
//
// A piece of alignseq in stream PF_OUT_ALIGNSEQ of a text slab:
//     u64 n_groups, u64 group_bubbles
//     (n_groups + 1) x { u64 text_off, u64 rec_off }    where the text / the records of bubble g * group_bubbles begin; the last
//                                                       entry = the piece's text length / the records' length
//     records, bubble after bubble (bubbles without rows have none):
//         u64 var_count, u32 entrance id, u32 exit id, u32 n_cols, u32 n_rows | strict << 31
//         n_rows x row_bytes(n_cols) bytes: character i of a row in bits [3i, 3i + 3) of the row's bytes read as one little-endian
//         number; 0 '-', 1 'A', 2 'C', 3 'G', 4 'T'
// Nothing is aligned: headers are read and written with memcpy.
#pragma once
#include <cstdint>
#include <cstring>

#ifdef __HIPCC__
#define PF_ALNPACK_HD __host__ __device__
#else
#define PF_ALNPACK_HD
#endif

namespace pf {

constexpr uint32_t ALNPACK_GROUP = 256;       // bubbles per index entry
constexpr uint32_t ALNPACK_HEADER = 24;       // bytes of a record's header

PF_ALNPACK_HD inline uint32_t alnpack_row_bytes(uint32_t n_cols) { return 3u * ((n_cols + 7u) >> 3); }   // (8 characters = 3 bytes)
PF_ALNPACK_HD inline uint64_t alnpack_index_bytes(uint64_t n_bubbles) { return 16 + ((n_bubbles + ALNPACK_GROUP - 1) / ALNPACK_GROUP + 1) * 16; }
PF_ALNPACK_HD inline uint32_t alnpack_code(char c) { return c == '-' ? 0u : c == 'A' ? 1u : c == 'C' ? 2u : c == 'G' ? 3u : 4u; }

#ifndef __HIP_DEVICE_COMPILE__
// ---- host side: text out of a piece ---------------------------------------------------------------------------------------------
struct AlnPackTables {
    uint32_t four[4096];   // 12 bits = 4 characters, first character in the lowest byte
    AlnPackTables() {
        static const char letter[8] = {'-', 'A', 'C', 'G', 'T', '?', '?', '?'};
        for (uint32_t v = 0; v < 4096; ++v)
            four[v] = (uint32_t)(uint8_t)letter[v & 7] | (uint32_t)(uint8_t)letter[(v >> 3) & 7] << 8 | (uint32_t)(uint8_t)letter[(v >> 6) & 7] << 16 |
                      (uint32_t)(uint8_t)letter[(v >> 9) & 7] << 24;
    }
};
inline const AlnPackTables &alnpack_tables() {
    static const AlnPackTables t;
    return t;
}

inline char *alnpack_put_uint(char *p, uint64_t x) {
    char tmp[24];
    int n = 0;
    do { tmp[n++] = (char)('0' + x % 10); x /= 10; } while (x);
    while (n) *p++ = tmp[--n];
    return p;
}

struct AlnPackPiece {
    const uint8_t *base = nullptr;
    uint64_t n_groups = 0, group_bubbles = 0;
    const uint8_t *index = nullptr, *records = nullptr;
    bool parse(const uint8_t *p, uint64_t len) {
        if (len < 32) return false;
        memcpy(&n_groups, p, 8);
        memcpy(&group_bubbles, p + 8, 8);
        if (n_groups > (len - 16) / 16 - 1) return false;
        base = p;
        index = p + 16;
        records = index + (n_groups + 1) * 16;
        return true;
    }
    void entry(uint64_t g, uint64_t &text_off, uint64_t &rec_off) const {
        memcpy(&text_off, index + g * 16, 8);
        memcpy(&rec_off, index + g * 16 + 8, 8);
    }
};

// the text of the records in [rec, rec_end) to [dst, dst_end); returns the end of what was written, or nullptr when a record does not
// lie inside [rec, rec_end) or its rows would cross dst_end (a damaged record must not write over the bytes of other pieces or ranks
// in the mapped result file: nothing beyond the row before the damage is written)
inline char *alnpack_expand(const uint8_t *rec, const uint8_t *rec_end, char *dst, const char *dst_end) {
    const AlnPackTables &T = alnpack_tables();
    char prefix[96];
    while (rec < rec_end) {
        if ((size_t)(rec_end - rec) < ALNPACK_HEADER) return nullptr;
        uint64_t vc;
        uint32_t ent, ext, L, R;
        memcpy(&vc, rec, 8);
        memcpy(&ent, rec + 8, 4);
        memcpy(&ext, rec + 12, 4);
        memcpy(&L, rec + 16, 4);
        memcpy(&R, rec + 20, 4);
        const bool strict = (R >> 31) != 0;
        R &= 0x7FFFFFFFu;
        rec += ALNPACK_HEADER;
        char *q = alnpack_put_uint(prefix, vc);
        *q++ = '\t'; *q++ = strict ? '1' : '0'; *q++ = '\t';
        q = alnpack_put_uint(q, ent);
        *q++ = '\t';
        q = alnpack_put_uint(q, ext);
        *q++ = '\t';
        const size_t np = (size_t)(q - prefix);
        const uint32_t rb = alnpack_row_bytes(L);
        if ((uint64_t)R * rb > (uint64_t)(rec_end - rec) || (uint64_t)R * (np + L + 1) > (uint64_t)(dst_end - dst)) return nullptr;
        for (uint32_t r = 0; r < R; ++r, rec += rb) {
            memcpy(dst, prefix, np);
            dst += np;
            uint32_t x = 0;
            const uint8_t *s = rec;
            for (; x + 8 <= L; x += 8, s += 3) {   // 3 bytes -> 8 characters, whole
                const uint32_t v = (uint32_t)s[0] | (uint32_t)s[1] << 8 | (uint32_t)s[2] << 16;
                const uint32_t a = T.four[v & 4095], b = T.four[v >> 12];
                memcpy(dst + x, &a, 4);
                memcpy(dst + x + 4, &b, 4);
            }
            if (x < L) {
                const uint32_t v = (uint32_t)s[0] | (uint32_t)s[1] << 8 | (uint32_t)s[2] << 16;
                char tail[8];
                const uint32_t a = T.four[v & 4095], b = T.four[v >> 12];
                memcpy(tail, &a, 4);
                memcpy(tail + 4, &b, 4);
                memcpy(dst + x, tail, L - x);
            }
            dst += L;
            *dst++ = '\n';
        }
    }
    return dst;
}
#endif

}  // namespace pf
