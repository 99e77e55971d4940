// CPU check of the device text formatter (ploidyfrost_amd/csrc/pf_format_dev.hpp is host-compilable for exactly this):
// put_double must equal printf("%g") for every double -- random bit patterns over the whole range, values near
// rounding ties, the magnitudes the path prints (coverages, frequencies, coefficients), integers, subnormals.
//   hipcc -x hip --offload-arch=gfx950 -O2 -I ploidyfrost_amd/csrc tests/cpp/test_format.cpp -o test_format && ./test_format
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>

#include "pf_format_dev.hpp"

static long g_bad = 0, g_n = 0;
static void check(double v) {
    char a[64], b[64];
    snprintf(a, sizeof a, "%g", v);
    pf::BufSink s{b};
    pf::put_double(s, v);
    *s.p = 0;
    pf::CountSink c;
    pf::put_double(c, v);
    ++g_n;
    const bool nan = v != v;
    if (nan) { if (strcmp(b, "-nan")) { ++g_bad; printf("nan -> %s\n", b); } return; }
    if (strcmp(a, b) || c.n != strlen(b)) {
        if (++g_bad < 20) printf("MISMATCH %a: printf %s  ours %s (count %llu)\n", v, a, b, (unsigned long long)c.n);
    }
}

int main(int argc, char **argv) {
    const long N = argc > 1 ? atol(argv[1]) : 2000000;
    std::mt19937_64 rng(12345);
    for (long i = 0; i < N; ++i) {  // any bit pattern
        uint64_t b = rng();
        double v;
        memcpy(&v, &b, 8);
        check(v);
    }
    for (long i = 0; i < N; ++i) {  // the magnitudes of the path: 10^-8 .. 10^10
        const double e = (double)(rng() % 18000) / 1000.0 - 8.0;
        const double v = std::pow(10.0, e) * (1.0 + (double)(rng() % 1000003) / 1000003.0);
        check(v);
        check(std::floor(v));
        check((double)(rng() % 100000) / (double)(1 + rng() % 5000));  // sum / length
    }
    for (long i = 0; i < N / 4; ++i) {  // exact ties and their neighbours: 6 digits + a trailing 5
        const uint64_t n = 100000 + rng() % 900000;
        for (int e = -12; e <= 12; ++e) {
            const double v = ((double)n + 0.5) * std::pow(10.0, e);
            check(v);
            check(std::nextafter(v, 0.0));
            check(std::nextafter(v, 1e300));
        }
    }
    const double special[] = {0.0, -0.0, 1.0, 0.5, 0.1, 100000.0, 999999.0, 999999.5, 1e6, 1e-4, 9.99995e-5, 1e-5, 123456.5, 1234565.0, 0.0001234565,
                              2.5, 5e-324, 2.2250738585072014e-308, 1.7976931348623157e308, 4503599627370496.0, 9007199254740992.0, 1e22, 1e23,
                              INFINITY, -INFINITY, 33.333333333333336, 0.66666666666666663, 20.0, 0.25};
    for (double v : special) { check(v); check(-v); }
    check(std::nan(""));
    printf("%ld values, %ld mismatches\n", g_n, g_bad);
    return g_bad != 0;
}
