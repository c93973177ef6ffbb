// Test harness (CPU): the integer model of the x87 operations (popdel_amd/csrc/hip/x87soft.h, the code the kernels run)
// against the FPU itself on random and on adversarial operands. Prints the number of mismatches.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>

#include "x87soft.h"

static pd::X87 fromLongDouble(long double v)
{
    // 80-bit format: 64-bit mantissa with explicit integer bit, 15-bit exponent
    unsigned char raw[16] = {0};
    std::memcpy(raw, &v, 10);
    uint64_t m;
    uint16_t se;
    std::memcpy(&m, raw, 8);
    std::memcpy(&se, raw + 8, 2);
    pd::X87 r;
    r.pad = 0;
    const int ex = se & 0x7fff;
    if (m == 0)
        return pd::x87Zero();
    if (ex == 0)
    {
        const int lz = __builtin_clzll(m);
        r.m = m << lz;
        r.e = -16382 - lz;
    }
    else
    {
        r.m = m;
        r.e = ex - 16383;
    }
    return r;
}

static double reference(long double e1, long double e0, double noDel, double del)
{
    volatile double s = del + noDel;
    double p = e1 * noDel / s + e0;
    return p;
}

int main(int argc, char ** argv)
{
    const long n = argc > 1 ? atol(argv[1]) : 2000000;
    std::mt19937_64 rng(12345);
    std::uniform_real_distribution<double> u(0.0, 1.0);
    long bad = 0, total = 0;
    auto check = [&](long double e1, long double e0, double noDel, double del) {
        const double want = reference(e1, e0, noDel, del);
        const double got = pd::x87ProbRef(fromLongDouble(e1), fromLongDouble(e0), noDel, del + noDel);
        ++total;
        if (std::memcmp(&want, &got, sizeof(double)) != 0)
        {
            if (bad < 10)
                std::fprintf(stderr, "mismatch: e1 %.21Lg e0 %.21Lg noDel %.17g del %.17g want %.17g got %.17g\n", e1, e0, noDel, del,
                             want, got);
            ++bad;
        }
    };
    for (long i = 0; i < n; ++i)
    {
        // weights as the EM produces them: exp() of sums of logs, anywhere between 1 and far below the double range
        const long double e1 = expl(-(long double)std::pow(10.0, u(rng) * 4.2 - 1.5) * (rng() & 1 ? 1.0L : 1e-3L));
        const long double e0 = (rng() % 7 == 0) ? 1.0L : expl(-(long double)std::pow(10.0, u(rng) * 4.2 - 1.5) * (rng() & 1 ? 1.0L : 1e-3L));
        const double noDel = std::pow(10.0, -u(rng) * 9.0), del = std::pow(10.0, -u(rng) * 9.0);
        check(e1, e0, noDel, del);
        check(e1, e0, noDel, noDel);                    // equal densities
        check(e1, 0.0L, noDel, del);
        check(0.0L, e0, noDel, del);
    }
    // ties and boundaries: values that are exactly representable, halfway cases of the second rounding, tiny results
    const long double specials[] = {1.0L, 0.5L, 0.75L, 1.0L - 0x1p-64L, 1.0L - 0x1p-54L, 0x1p-53L, 0x1p-54L, 0x1p-63L, 0x1p-64L, 0x1p-65L,
                                    0x1p-1022L, 0x1p-1030L, 0x1p-1070L, 0x1p-1074L, 0x1p-1075L, 0x1.8p-1075L, 0x1p-1076L, 0x1p-2000L,
                                    0x1p-16382L, 0x1p-16400L, 0x1.fffffffffffff8p-1L, 0x1.00000000000008p0L, 0x1.000000000000080002p0L,
                                    0x1.000000000000180000p0L, 0.0L};
    const double dens[] = {1.0, 0.5, 0.25, 1e-10, 3.0, 1e-300, 0x1p-1022, 0x1p-1074, 0.1, 0.3};
    for (long double e1 : specials)
        for (long double e0 : specials)
            for (double a : dens)
                for (double b : dens)
                    check(e1, e0, a, b);
    std::printf("%ld %ld\n", bad, total);
    return bad ? 1 : 0;
}
