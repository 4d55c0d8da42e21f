// x**2 and x**0.5 as upstream's Python computes them: through the C library's pow() (sphere.py:18-19, util.py:29-32).
// pow() is correct to 0.52 ulp, not correctly rounded: about one result in 1,200 is an ulp off x*x or sqrt(x), and the
// sphere's b**2 - 4ac then moves a root by a few ulps of |o - c| + r - enough to change places with a triangle whose
// depth agrees to 15 digits, or to take a root of 1e-8 a thousandth of itself away.  The double-precision exact code
// therefore restates the library's algorithm (glibc 2.28 and later, sysdeps/ieee754/dbl-64/e_pow.c as built for fused
// multiply-add): log(x) as a double-double from a 128-entry table, times y, exp() from a 128-entry table.  Every
// fused multiply-add the library's build contains is spelled out here, so the result is the library's to the last bit
// (the build holds it against pow() on four million results before it switches this on: backend/build.py).
// The tables are the library's own, read out of the build machine's libm (rt_libm_tables_gen.h); without them -
// RT_LIBM_POW 0 - squares are products and roots are sqrt, which is within an ulp of upstream everywhere.
//
// Only y = 2 and y = 1/2 occur.  Arguments outside 2^-500 .. 2^500, zeros, infinities and NaNs take the plain forms
// (the library has special cases there; no ray of a scene comes near them).
#pragma once
#include <stdint.h>

#if defined(RT_LIBM_HOST_TEST)
#define RT_LIBM_FN static inline
#define RT_LIBM_CONST static const
#else
#define RT_LIBM_FN __device__ __forceinline__
#define RT_LIBM_CONST static __device__ const
#endif

#if defined(RT_LIBM_TABLES_GENERATED) || defined(RT_LIBM_HOST_TEST)
#define RT_LIBM_POW 1
#include "rt_libm_tables_gen.h"
#else
#define RT_LIBM_POW 0
#endif

namespace rt {

#if RT_LIBM_POW
namespace libm {
RT_LIBM_FN uint64_t bits(double x) { return __builtin_bit_cast(uint64_t, x); }
RT_LIBM_FN double from_bits(uint64_t u) { return __builtin_bit_cast(double, u); }

// pow(x, y) for finite x in [2^-500, 2^500], y = 2 or 1/2.  (Not inlined on the device: the settling code calls it
// from a dozen places, and inlined it made the kernels that hold that code two thirds longer.)
#if defined(RT_LIBM_HOST_TEST)
static inline
#else
__device__ __attribute__((noinline))
#endif
double pow_core(double x, double y) {
    const uint64_t ix = bits(x);
    // log(x) = k ln2 + log(c) + log1p(z / c - 1): hi + tail
    const uint64_t tmp = ix - 0x3fe6955500000000ull;
    const int i = (int)((tmp >> (52 - 7)) % 128u);
    const int k = (int)((int64_t)tmp >> 52);
    const uint64_t iz = ix - (tmp & (0xfffull << 52));
    const double z = from_bits(iz), kd = (double)k;
    const double invc = kLogTab[i][0], logc = kLogTab[i][1], logctail = kLogTab[i][2];
    const double r = __builtin_fma(z, invc, -1.0);
    const double t1 = __builtin_fma(kd, kLn2Hi, logc), t2 = t1 + r;
    const double lo1 = __builtin_fma(kd, kLn2Lo, logctail), lo2 = (t1 - t2) + r;
    const double ar = kLogPoly[0] * r, ar2 = r * ar, ar3 = r * ar2;
    const double hi = t2 + ar2, lo3 = __builtin_fma(ar, r, -ar2), lo4 = (t2 - hi) + ar2;
    const double q = __builtin_fma(ar2, __builtin_fma(ar2, __builtin_fma(r, kLogPoly[6], kLogPoly[5]), __builtin_fma(r, kLogPoly[4], kLogPoly[3])),
                                   __builtin_fma(r, kLogPoly[2], kLogPoly[1]));
    const double lo = __builtin_fma(ar3, q, ((lo1 + lo2) + lo3) + lo4);
    const double lh = hi + lo, lt = (hi - lh) + lo;
    // y log(x) as a double-double
    const double ehi = y * lh, elo = __builtin_fma(y, lt, __builtin_fma(y, lh, -ehi));
    // exp
    const uint32_t abstop = (uint32_t)(bits(ehi) >> 52) & 0x7ffu;
    if (abstop - 0x3c9u >= 0x408u - 0x3c9u) {
        if (abstop - 0x3c9u >= 0x80000000u) return 1.0 + ehi;   // |y log x| < 2^-54
        return y == 2.0 ? x * x : __builtin_sqrt(x);            // (over- and underflow: out of the callers' range)
    }
    double kd2 = __builtin_fma(kInvLn2N, ehi, kShift);
    const uint64_t ki = bits(kd2);
    kd2 -= kShift;
    double rr = __builtin_fma(kd2, kNegLn2LoN, __builtin_fma(kd2, kNegLn2HiN, ehi));
    rr += elo;
    const uint64_t idx = 2 * (ki % 128u), top = ki << (52 - 7);
    const double tail = from_bits(kExpTab[idx]);
    const uint64_t sbits = kExpTab[idx + 1] + top;
    const double r2 = rr * rr;
    const double t = __builtin_fma(r2 * r2, __builtin_fma(rr, kExpPoly[3], kExpPoly[2]),
                                   __builtin_fma(r2, __builtin_fma(rr, kExpPoly[1], kExpPoly[0]), tail + rr));
    const double scale = from_bits(sbits);
    return __builtin_fma(scale, t, scale);
}
RT_LIBM_FN bool in_range(double ax) { return ax >= 0x1p-500 && ax <= 0x1p500; }   // (false for a NaN)
}  // namespace libm

// Nine times in ten the plain result is provably the library's: pow() is good to 0.52 ulp, so it can only name the
// plain result's NEIGHBOUR when the exact value lies within 0.02 ulp of the half-way point between the two (on 200
// million arguments each, pow(x, 2) != x * x and pow(x, 0.5) != sqrt(x) only within 0.009 ulp of it).  The exact
// residual of the plain result says how far it is: below 0.45 ulp, the plain result stands and the tables stay unread.
RT_LIBM_FN double libm_square(double x) {
    const double p = x * x, e = __builtin_fma(x, x, -p);                                  // x^2 = p + e exactly
    const double scale = libm::from_bits(libm::bits(p) & 0x7ff0000000000000ull);          // ulp(p) = scale 2^-52
    if (__builtin_fabs(e) * 0x1p52 <= 0.45 * scale) return p;
    const double ax = __builtin_fabs(x);
    return libm::in_range(ax) ? libm::pow_core(ax, 2.0) : p;
}
RT_LIBM_FN double libm_root(double x) {
    const double q = __builtin_sqrt(x), e = __builtin_fma(q, q, -x);                      // q^2 - x exactly; q - sqrt(x) = e / 2q to first order
    const double scale = libm::from_bits(libm::bits(q) & 0x7ff0000000000000ull);
    if (__builtin_fabs(e) * 0x1p52 <= 0.9 * q * scale) return q;
    return libm::in_range(x) ? libm::pow_core(x, 0.5) : q;
}
#else
RT_LIBM_FN double libm_square(double x) { return x * x; }
RT_LIBM_FN double libm_root(double x) { return __builtin_sqrt(x); }
#endif

}  // namespace rt
