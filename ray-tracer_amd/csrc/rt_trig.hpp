// rt_trig.hpp -- sinf / cosf as the host's C library computes them.
//
// The reference draws a lens sample (src/extra.cpp:27-44) and a glossy sample (:353-356) at an
// angle of 2 pi u and takes its sine and cosine with the C library.  glibc (2.28 and later)
// evaluates both in double precision -- the argument reduced by a multiple of pi/2 found with
// one multiplication, then a degree-7 (sine) or degree-8 (cosine) polynomial -- and rounds the
// double to float once.  The device library's own sinf / cosf are float routines and differ
// from that in the last bit for every fifth argument, which moves a lens sample by an ulp and,
// once in some ten thousand rays, a ray across a triangle's edge.  Evaluated the same way here,
// the results are the host's bit for bit: checked against the C library for every float in
// [0, 2 pi] (tests/cpp/trig_check.cpp), the only range the path asks for.  Double precision
// runs at half the float rate on this GPU and two calls per lens sample do not show in a frame.
#pragma once
#include <cstdint>
#include <cstring>

#if defined(__HIPCC__)
#define RT_TRIG_HD __host__ __device__ __forceinline__
#else
#define RT_TRIG_HD inline
#endif

namespace rt {

// n even: sine of x, n odd: cosine of x, for |x| <= pi/4 (x2 = x * x); `flip` negates the cosine
RT_TRIG_HD float trigPoly(double x, double x2, bool flip, int n)
{
    if ((n & 1) == 0) {
        const double s1 = -0x1.555545995a603p-3, s2 = 0x1.1107605230bc4p-7, s3 = -0x1.994eb3774cf24p-13;
        const double x3 = x * x2;
        const double hi = s2 + x2 * s3;
        const double x7 = x3 * x2;
        const double lo = x + x3 * s1;
        return (float)(lo + x7 * hi);
    }
    const double g = flip ? -1.0 : 1.0;
    const double c0 = g * 0x1p0, c1 = g * -0x1.ffffffd0c621cp-2, c2 = g * 0x1.55553e1068f19p-5, c3 = g * -0x1.6c087e89a359dp-10,
                 c4 = g * 0x1.99343027bf8c3p-16;
    const double x4 = x2 * x2;
    const double hi = c3 + x2 * c4;
    const double lo = c0 + x2 * c1;
    const double x6 = x4 * x2;
    const double mid = lo + x4 * c2;
    return (float)(mid + x6 * hi);
}

RT_TRIG_HD uint32_t trigTop12(float f)
{
    uint32_t u;
    memcpy(&u, &f, 4);
    return (u >> 20) & 0x7ffu;
}

// sine (cosine = false) or cosine of y, |y| < 120; the callers' arguments lie in [0, 2 pi]
RT_TRIG_HD float trigLikeHost(float y, bool cosine)
{
    const double x = (double)y;
    if (trigTop12(y) < trigTop12(0x1.921FB6p-1f)) { // below pi/4 (by exponent and top mantissa bits): no reduction
        if (trigTop12(y) < trigTop12(0x1p-12f))
            return cosine ? 1.0f : y;
        return trigPoly(x, x * x, false, cosine ? 1 : 0);
    }
    // quadrant: 2/pi scaled by 2^24, so that rounding to nearest is an integer add and a shift
    const double r = x * 0x1.45F306DC9C883p+23;
    const int n = ((int32_t)r + 0x800000) >> 24;
    const double xr = x - (double)n * 0x1.921FB54442D18p0;
    const double sign = ((n & 3) == 1 || (n & 3) == 2) ? -1.0 : 1.0;
    return trigPoly(xr * sign, xr * xr, (n & 2) != 0, cosine ? (n ^ 1) : n);
}

RT_TRIG_HD float sinLikeHost(float y) { return trigLikeHost(y, false); }
RT_TRIG_HD float cosLikeHost(float y) { return trigLikeHost(y, true); }

} // namespace rt
