// Text formatting on the device for the result rows (reference output writers, src/CDBG.cpp:1259, 1303-1340,
// 1552-1652): unsigned integers and `ostream << double` with default flags, i.e. printf("%g") -- 6 significant
// digits, correctly rounded from the EXACT binary value (round-half-even on ties, as glibc and std::to_chars do),
// fixed notation for decimal exponents -4..5, otherwise d.ddddde+XX, trailing zeros and a bare point removed.
//
// The conversion never divides a wide number: a double is m * 2^E with a 53-bit m.  For E < 0 the integer part is
// m >> -E and the fraction F = m mod 2^-E is turned into decimal digits by "F *= 10; digit = F >> -E", which is exact
// as long as F fits its limbs.  Values in [2^-68, 2^53) -- every coverage, frequency and coefficient the path prints
// -- keep F in two 64-bit registers; anything else (huge integers, tiny fractions, subnormals) takes the generic
// limb-array route below, exact for every finite double.  Sink is a template so that the same row writers run once
// to measure (CountSink) and once to write (BufSink).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace pf {

__host__ __device__ inline uint64_t fmt_bits(double v) { return __builtin_bit_cast(uint64_t, v); }
__host__ __device__ inline double fmt_double(uint64_t b) { return __builtin_bit_cast(double, b); }
__host__ __device__ inline uint64_t fmt_mulhi(uint64_t a, uint64_t b) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __umul64hi(a, b);
#else  // the host build of this header exists for the CPU unit test of the conversion (tests/cpp/test_format.cpp)
    return (uint64_t)(((unsigned __int128)a * b) >> 64);
#endif
}

struct CountSink {
    uint64_t n = 0;
    __host__ __device__ inline void put(char) { ++n; }
    __host__ __device__ inline void put_n(const char *, uint32_t len) { n += len; }
    __host__ __device__ inline void skip(uint64_t len) { n += len; }
};
struct BufSink {
    char *p;
    __host__ __device__ inline void put(char c) { *p++ = c; }
    __host__ __device__ inline void put_n(const char *s, uint32_t len) { for (uint32_t i = 0; i < len; ++i) p[i] = s[i]; p += len; }
};

// Decimal digits without an array: a local buffer indexed by a running count lives in private memory on the device, and every
// digit then costs a store and a load with memory latency behind each other (K-TEXT's threads spent most of their time there).
// The digits are collected least significant first into the low byte of a register that is shifted up as they come, so the
// most significant one ends in the lowest byte and the bytes leave in printing order.
template <class Sink>
__host__ __device__ inline void put_uint(Sink &s, uint64_t x) {
    if (x <= 0xFFFFFFFFull) {   // (ids, counts, lengths: at most ten digits, 32-bit arithmetic)
        uint32_t y = (uint32_t)x;
        uint64_t w0 = 0;
        uint32_t w1 = 0;
        int n = 0;
        do {
            const uint32_t q = y / 10u, d = y - q * 10u;
            if (n < 8) w0 = (w0 << 8) | (uint64_t)('0' + d);
            else w1 = (w1 << 8) | (uint32_t)('0' + d);
            y = q;
            ++n;
        } while (y);
        for (int i = 8; i < n; ++i) { s.put((char)(w1 & 0xFFu)); w1 >>= 8; }
        for (int i = 0; i < (n < 8 ? n : 8); ++i) { s.put((char)(w0 & 0xFFu)); w0 >>= 8; }
        return;
    }
    uint64_t w0 = 0, w1 = 0, w2 = 0;
    int n = 0;
    do {
        const uint64_t q = x / 10u, d = x - q * 10u;
        if (n < 8) w0 = (w0 << 8) | (uint64_t)('0' + d);
        else if (n < 16) w1 = (w1 << 8) | (uint64_t)('0' + d);
        else w2 = (w2 << 8) | (uint64_t)('0' + d);
        x = q;
        ++n;
    } while (x);
    for (int i = 16; i < n; ++i) { s.put((char)(w2 & 0xFFu)); w2 >>= 8; }
    for (int i = 8; i < (n < 16 ? n : 16); ++i) { s.put((char)(w1 & 0xFFu)); w1 >>= 8; }
    for (int i = 0; i < 8; ++i) { s.put((char)(w0 & 0xFFu)); w0 >>= 8; }
}
__host__ __device__ inline uint32_t uint_digits(uint64_t x) {
    uint32_t n = 1;
    while (x >= 10) { x /= 10; ++n; }
    return n;
}

// 6 significant digits (as an integer 100000..999999) and the decimal exponent of the first one, from the digit
// stream of the exact value: `lead` = digits collected so far (k of them), then the tail status decides the rounding
struct G6 {
    uint32_t n;  // 100000 .. 999999
    int x;       // decimal exponent of the leading digit
};
__host__ __device__ inline G6 g6_round(uint32_t n, int x, int tail /* -1 below half, 0 exactly half, 1 above */) {
    if (tail > 0 || (tail == 0 && (n & 1))) {
        if (++n == 1000000u) { n = 100000u; ++x; }
    }
    return G6{n, x};
}

// generic route: any finite non-zero double; 32-bit limbs, little endian.  1100 bits cover 2^-1074 .. 2^1024.
__host__ __device__ __noinline__ inline G6 g6_wide(uint64_t m, int E) {
    constexpr int LIMBS = 36;
    uint32_t a[LIMBS];
    for (int i = 0; i < LIMBS; ++i) a[i] = 0;
    if (E >= 0) {
        // integer m << E: all decimal digits by repeated division by 10^9, most significant group last
        const int w = E >> 5, sh = E & 31;
        const uint64_t v0 = (m & 0xFFFFFFFFull) << sh, v1 = (m >> 32) << sh;  // 63 and 52 bits at most
        a[w] = (uint32_t)v0;
        a[w + 1] = (uint32_t)(v0 >> 32) | (uint32_t)v1;
        a[w + 2] = (uint32_t)(v1 >> 32);
        int top = w + 2;
        while (top > 0 && a[top] == 0) --top;
        uint32_t groups[36];
        int ng = 0;
        for (;;) {
            uint64_t rem = 0;
            bool nz = false;
            for (int i = top; i >= 0; --i) {
                const uint64_t cur = (rem << 32) | a[i];
                a[i] = (uint32_t)(cur / 1000000000u);
                rem = cur % 1000000000u;
                nz |= a[i] != 0;
            }
            groups[ng++] = (uint32_t)rem;
            while (top > 0 && a[top] == 0) --top;
            if (!nz) break;
        }
        // digits, most significant first
        uint32_t n = 0;
        int k = 0, total = 0, tail = -1;
        bool tail_nz = false, tail_first_set = false;
        int tail_first = 0;
        for (int g = ng - 1; g >= 0; --g) {
            uint32_t v = groups[g], p = 100000000u;
            for (int d = 0; d < 9; ++d, p /= 10) {
                const uint32_t dig = v / p;
                v %= p;
                if (total == 0 && dig == 0) continue;  // leading zeros of the top group
                ++total;
                if (k < 6) { n = n * 10 + dig; ++k; }
                else if (!tail_first_set) { tail_first = (int)dig; tail_first_set = true; }
                else if (dig) tail_nz = true;
            }
        }
        while (k < 6) { n *= 10; ++k; }  // (cannot happen for m >= 2^52, kept for completeness)
        if (tail_first_set) tail = tail_first > 5 ? 1 : (tail_first < 5 ? -1 : (tail_nz ? 1 : 0));
        return g6_round(n, total - 1, tail);
    }
    // E < 0: integer part I = m >> s, fraction F = m mod 2^s in limbs
    const int s = -E;
    uint64_t I = s >= 64 ? 0 : (m >> s);
    uint64_t f = s >= 64 ? m : (m & ((1ull << s) - 1));
    a[0] = (uint32_t)f;
    a[1] = (uint32_t)(f >> 32);
    const int nl = (s + 31) / 32 + 1;  // limbs that hold F * 10
    uint32_t n = 0;
    int k = 0, x = 0;
    if (I) {
        char buf[20];
        int nd = 0;
        while (I) { buf[nd++] = (char)(I % 10); I /= 10; }
        x = nd - 1;
        // (I < 2^53 has at most 16 digits; with s > 120 it has at most ... but stay general)
        int i = nd - 1;
        for (; i >= 0 && k < 6; --i) { n = n * 10 + (uint32_t)buf[i]; ++k; }
        if (k == 6 && i >= 0) {
            const int first = buf[i];
            bool nz = false;
            for (int j = i - 1; j >= 0; --j) nz |= buf[j] != 0;
            for (int j = 0; j < nl; ++j) nz |= a[j] != 0;
            return g6_round(n, x, first > 5 ? 1 : (first < 5 ? -1 : (nz ? 1 : 0)));
        }
    }
    int zeros = 0;
    while (k < 6) {
        uint64_t carry = 0;
        for (int j = 0; j < nl; ++j) {
            const uint64_t cur = (uint64_t)a[j] * 10 + carry;
            a[j] = (uint32_t)cur;
            carry = cur >> 32;
        }
        // digit = F >> s
        const int w = s >> 5, sh = s & 31;
        uint32_t dig = a[w] >> sh;
        if (sh && w + 1 < LIMBS) dig |= a[w + 1] << (32 - sh);
        dig &= 15;
        a[w] &= sh ? ((1u << sh) - 1) : 0u;
        for (int j = w + 1; j < nl; ++j) a[j] = 0;
        if (k == 0 && dig == 0) { ++zeros; continue; }
        if (k == 0 && x == 0 && n == 0) x = -(zeros + 1);
        n = n * 10 + dig;
        ++k;
    }
    // remainder F against half = 2^(s-1)
    const int hb = s - 1, hw = hb >> 5;
    const uint32_t hbit = 1u << (hb & 31);
    int tail;
    if (!(a[hw] & hbit)) tail = -1;
    else {
        bool nz = (a[hw] & (hbit - 1)) != 0;
        for (int j = 0; j < hw; ++j) nz |= a[j] != 0;
        tail = nz ? 1 : 0;
    }
    return g6_round(n, x, tail);
}

// the common route: 2^-68 <= x < 2^53 (E in [-120, 0)), F in two registers
__host__ __device__ inline G6 g6_of(double v) {  // v finite, > 0
    const uint64_t bits = fmt_bits(v);
    const int be = (int)((bits >> 52) & 0x7FF);
    uint64_t m = bits & 0xFFFFFFFFFFFFFull;
    int E;
    if (be == 0) E = -1074;
    else { m |= 1ull << 52; E = be - 1075; }
    if (E >= 0 || E < -120) return g6_wide(m, E);
    const int s = -E;  // 1 .. 120
    uint64_t I, flo, fhi;
    if (s >= 64) { I = 0; flo = m; fhi = 0; }
    else { I = m >> s; flo = m & ((1ull << s) - 1); fhi = 0; }
    uint32_t n = 0;
    int k = 0, x = 0;
    if (I) {
        // decimal digits of the integer part (at most 16)
        // (an integer part below 2^32 -- every coverage -- counts its digits by comparisons instead of divisions)
        uint32_t nd = I > 0xFFFFFFFFull ? uint_digits(I)
                                        : 1u + ((uint32_t)I >= 10u) + ((uint32_t)I >= 100u) + ((uint32_t)I >= 1000u) + ((uint32_t)I >= 10000u) + ((uint32_t)I >= 100000u) +
                                              ((uint32_t)I >= 1000000u) + ((uint32_t)I >= 10000000u) + ((uint32_t)I >= 100000000u) + ((uint32_t)I >= 1000000000u);
        x = (int)nd - 1;
        if (nd > 6) {
            uint64_t p = 1;
            for (uint32_t i = 6; i < nd; ++i) p *= 10;
            n = (uint32_t)(I / p);
            const uint64_t r = I % p, half = p / 2;
            const bool fnz = (flo | fhi) != 0;
            return g6_round(n, x, r > half ? 1 : (r < half ? -1 : (fnz ? 1 : 0)));
        }
        n = (uint32_t)I;
        k = (int)nd;
    }
    int zeros = 0;
    while (k < 6) {
        // F *= 10 (F < 2^s, s <= 120: the product stays below 2^124)
        const uint64_t lo10 = flo * 10, c = fmt_mulhi(flo, 10);
        fhi = fhi * 10 + c;
        flo = lo10;
        uint32_t dig;
        if (s >= 64) { dig = (uint32_t)(fhi >> (s - 64)); fhi &= (s == 64) ? 0 : ((1ull << (s - 64)) - 1); }
        else { dig = (uint32_t)((flo >> s) | (fhi << (64 - s))); flo &= (1ull << s) - 1; fhi = 0; }
        if (k == 0 && dig == 0) { ++zeros; continue; }
        if (k == 0 && n == 0) x = -(zeros + 1);
        n = n * 10 + dig;
        ++k;
    }
    int tail;
    if (s > 64) {
        const uint64_t hb = 1ull << (s - 65);
        tail = !(fhi & hb) ? -1 : (((fhi & (hb - 1)) | flo) ? 1 : 0);
    } else {
        const uint64_t hb = 1ull << (s - 1);
        tail = !(flo & hb) ? -1 : ((flo & (hb - 1)) ? 1 : 0);
    }
    return g6_round(n, x, tail);
}

// printf("%g", v).  A NaN prints as "-nan": the only NaN the path can produce is 0.0 / 0.0, which the reference's x86
// host delivers as the negative default NaN.
template <class Sink>
__host__ __device__ inline void put_double(Sink &s, double v) {
    const uint64_t bits = fmt_bits(v);
    const bool neg = (bits >> 63) != 0;
    const uint64_t mag = bits & 0x7FFFFFFFFFFFFFFFull;
    if (mag > 0x7FF0000000000000ull) { s.put('-'); s.put('n'); s.put('a'); s.put('n'); return; }
    if (neg) s.put('-');
    if (mag == 0x7FF0000000000000ull) { s.put('i'); s.put('n'); s.put('f'); return; }
    if (mag == 0) { s.put('0'); return; }
    const G6 g = g6_of(fmt_double(mag));
    // the six digits in one register (no array: see put_uint), the most significant in the lowest byte; digit i = DG(i)
    uint64_t dw = 0;
    int last = 5;   // significant digits kept: 0 .. last (trailing zeros dropped, the first digit stays)
    {
        uint32_t n = g.n;
        bool tz = true;
#pragma unroll
        for (int i = 5; i >= 0; --i) {
            const uint32_t q = n / 10u, dg = n - q * 10u;
            n = q;
            dw = (dw << 8) | (uint64_t)('0' + dg);
            if (tz && dg == 0 && i > 0) last = i - 1;
            else tz = false;
        }
    }
#define PF_DG(i) ((char)((dw >> (8 * (i))) & 0xFFu))
    const int X = g.x;
    if (X < -4 || X >= 6) {
        s.put(PF_DG(0));
        if (last > 0) { s.put('.'); for (int i = 1; i <= last; ++i) s.put(PF_DG(i)); }
        s.put('e');
        uint32_t ax;
        if (X < 0) { s.put('-'); ax = (uint32_t)(-X); } else { s.put('+'); ax = (uint32_t)X; }
        if (ax < 10) s.put('0');
        put_uint(s, ax);
    } else if (X >= 0) {
        for (int i = 0; i <= X; ++i) s.put(i <= last ? PF_DG(i) : '0');
        if (last > X) { s.put('.'); for (int i = X + 1; i <= last; ++i) s.put(PF_DG(i)); }
    } else {
        s.put('0');
        s.put('.');
        for (int i = 0; i < -X - 1; ++i) s.put('0');
        for (int i = 0; i <= last; ++i) s.put(PF_DG(i));
    }
#undef PF_DG
}

}  // namespace pf
