// Software model of the x87 extended-precision operations the reference performs per read pair when it weighs the pairs
// of a read group for the reference shift (genotype_deletion_popdel_call.h:438-445):
//
//     double prob_ref = exp(a1Rg) * no_del / (del + no_del) + exp(a0Rg);
//
// with exp(a1Rg), exp(a0Rg) `long double` and no_del, del `double`: an x87 multiply, divide and add, each rounded to a
// 64-bit mantissa (round to nearest even, the Linux default precision control), then the store to `double` rounds a
// second time. IEEE-754 fixes these results bit for bit, so they can be reproduced with integer arithmetic on the GPU;
// what cannot be reproduced there are the two transcendental values themselves (x87 fyl2x / f2xm1 microcode), which
// the host works out per cell (host/rg0_x87.cpp) and hands to the kernels as X87 values.
//
// Only what that expression needs: non-negative finite operands. A value is m * 2^(e - 63) with bit 63 of m set, or
// zero (m == 0). The exponent is not limited to the x87 range: results below the x87 normal range are far below the
// smallest double as well, and the conversion to double is where they end up as 0.
//
// Compiled for the device by hipcc and for the host by g++ (tests/test_x87soft.py checks it against the FPU).
#ifndef PD_X87SOFT_H_
#define PD_X87SOFT_H_

#include <stdint.h>
#include <string.h>

#if defined(__HIPCC__)
#define PD_X87_FN __host__ __device__ __forceinline__
#else
#define PD_X87_FN inline
#endif

namespace pd {

struct X87
{
    uint64_t m;
    int32_t  e;
    int32_t  pad;
};

PD_X87_FN uint64_t x87MulHi(uint64_t a, uint64_t b)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __umul64hi(a, b);
#else
    return (uint64_t)(((unsigned __int128)a * b) >> 64);
#endif
}

PD_X87_FN int x87Clz(uint64_t v)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __clzll((long long)v);
#else
    return __builtin_clzll(v);
#endif
}

// round to nearest even on the bits below m: guard = the first dropped bit, sticky = any later one
PD_X87_FN X87 x87Round(uint64_t m, int32_t e, bool guard, bool sticky)
{
    if (guard && (sticky || (m & 1ull)))
    {
        ++m;
        if (m == 0)
        {
            m = 1ull << 63;
            ++e;
        }
    }
    X87 r;
    r.m = m;
    r.e = e;
    r.pad = 0;
    return r;
}

PD_X87_FN X87 x87Zero()
{
    X87 r;
    r.m = 0;
    r.e = 0;
    r.pad = 0;
    return r;
}

// exact (a double has 53 significant bits); d >= 0 and finite
PD_X87_FN X87 x87FromDouble(double d)
{
    uint64_t bits;
    memcpy(&bits, &d, sizeof(bits));
    const uint64_t frac = bits & ((1ull << 52) - 1ull);
    const int ex = (int)((bits >> 52) & 0x7ffu);
    X87 r;
    r.pad = 0;
    if (ex == 0)
    {
        if (frac == 0)
            return x87Zero();
        const int lz = x87Clz(frac);
        r.m = frac << lz;
        r.e = 63 - lz - 1074;
    }
    else
    {
        r.m = (1ull << 63) | (frac << 11);
        r.e = ex - 1023;
    }
    return r;
}

PD_X87_FN X87 x87Mul(X87 a, X87 b)
{
    if (a.m == 0 || b.m == 0)
        return x87Zero();
    uint64_t hi = x87MulHi(a.m, b.m), lo = a.m * b.m;
    int32_t e = a.e + b.e;
    if (hi >> 63)
        ++e;
    else
    {
        hi = (hi << 1) | (lo >> 63);
        lo <<= 1;
    }
    return x87Round(hi, e, (lo >> 63) != 0, (lo << 1) != 0);
}

// a / b, b != 0: the quotient's bits by restoring division (the mantissas' ratio lies in (1/2, 2))
PD_X87_FN X87 x87Div(X87 a, X87 b)
{
    if (a.m == 0)
        return x87Zero();
    uint64_t r = a.m, q = 0;
    int32_t e = a.e - b.e;
    int steps = 64;
    if (r >= b.m)
    {
        r -= b.m;
        q = 1;
        steps = 63;
    }
    else
        --e;
    for (int i = 0; i < steps; ++i)
    {
        const bool carry = (r >> 63) != 0;
        r <<= 1;
        q <<= 1;
        if (carry || r >= b.m)
        {
            r -= b.m;
            q |= 1ull;
        }
    }
    const bool carry = (r >> 63) != 0;
    r <<= 1;
    bool guard = false;
    if (carry || r >= b.m)
    {
        r -= b.m;
        guard = true;
    }
    return x87Round(q, e, guard, r != 0);
}

PD_X87_FN X87 x87Add(X87 a, X87 b)
{
    if (a.m == 0)
        return b;
    if (b.m == 0)
        return a;
    if (a.e < b.e)
    {
        const X87 t = a;
        a = b;
        b = t;
    }
    const int d = a.e - b.e;
    if (d >= 65)
        return a;                                   // below half an ulp of a: rounds away
    uint64_t bhi, blo;
    if (d == 0)
    {
        bhi = b.m;
        blo = 0;
    }
    else if (d < 64)
    {
        bhi = b.m >> d;
        blo = b.m << (64 - d);
    }
    else
    {
        bhi = 0;
        blo = b.m;
    }
    uint64_t hi = a.m + bhi, lo = blo;
    int32_t e = a.e;
    bool sticky = false;
    if (hi < a.m)                                   // carry out of bit 63
    {
        sticky = (lo & 1ull) != 0;
        lo = (lo >> 1) | (hi << 63);
        hi = (hi >> 1) | (1ull << 63);
        ++e;
    }
    return x87Round(hi, e, (lo >> 63) != 0, sticky || (lo << 1) != 0);
}

// the store to a double: second rounding, gradual underflow included
PD_X87_FN double x87ToDouble(X87 x)
{
    if (x.m == 0)
        return 0.0;
    int ex = x.e + 1023;
    uint64_t bits;
    if (ex >= 1)
    {
        uint64_t mant = x.m >> 11;
        const uint64_t rem = x.m & 0x7ffull;
        if ((rem >> 10) && ((rem & 0x3ffull) || (mant & 1ull)))
            ++mant;
        if (mant >> 53)
        {
            mant >>= 1;
            ++ex;
        }
        if (ex >= 2047)
            bits = 0x7ffull << 52;
        else
            bits = ((uint64_t)ex << 52) | (mant & ((1ull << 52) - 1ull));
    }
    else
    {
        const int shift = 11 + (1 - ex);            // >= 12
        uint64_t mant;
        bool guard, sticky;
        if (shift > 64)
            return 0.0;
        if (shift == 64)
        {
            mant = 0;
            guard = (x.m >> 63) != 0;
            sticky = (x.m << 1) != 0;
        }
        else
        {
            mant = x.m >> shift;
            guard = ((x.m >> (shift - 1)) & 1ull) != 0;
            sticky = (x.m & ((1ull << (shift - 1)) - 1ull)) != 0;
        }
        if (guard && (sticky || (mant & 1ull)))
            ++mant;
        bits = mant;                                // reaching 2^52 is the smallest normal number, as it should be
    }
    double d;
    memcpy(&d, &bits, sizeof(d));
    return d;
}

// What the host hands to the kernels per (window, candidate) cell and EM iteration (host/rg0_x87.cpp): read group 0's
// weights of the reference-shift update, which the reference uses for every read group.
struct Rg0Weights
{
    X87      expA0, expA1;     // exp(a0Rg), exp(a1Rg) as the FPU holds them
    double   expA0d, expA1d;   // the same rounded to double (for quotients that are nowhere near an integer)
    uint32_t invalid;          // one of them is NaN or infinite: every quotient is NaN, every shift becomes 0
    uint32_t pad;
};

// What the host needs per cell for that: the state the cell is in when its next reference-shift update runs.
struct Rg0Query
{
    int32_t  active;           // 0: the cell does not iterate any more
    int32_t  window;
    uint32_t len;              // current deletion-length estimate
    int32_t  shift0;           // current reference shift of read group 0 (meaningful on the context that owns it)
    double   gt[3];            // genotype priors
};

// The reference's expression above.
PD_X87_FN double x87ProbRef(X87 expA1Rg, X87 expA0Rg, double noDel, double delPlusNoDel)
{
    X87 t = x87Mul(expA1Rg, x87FromDouble(noDel));
    t = x87Div(t, x87FromDouble(delPlusNoDel));
    t = x87Add(t, expA0Rg);
    return x87ToDouble(t);
}

}  // namespace pd
#endif
