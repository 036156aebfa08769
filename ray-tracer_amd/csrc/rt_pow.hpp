// rt_pow.hpp -- powf as the host's C library computes it.
//
// The specular terms of the Phong and Blinn-Phong models (src/shading.cpp:92-153) raise a cosine to
// the material's shininess with the C library's powf.  glibc (2.28 and later) evaluates it in double
// precision -- log2 of x from a 16-entry table and a degree-5 polynomial, times y, then 2^ of that
// from a 32-entry table and a cubic -- and rounds the double to float once.  The device library's
// powf is a float routine and differs from that in the last bit of some results, which was the
// last source of differing pixel bits between this path and the CPU oracle.  Evaluated the same
// way here (the same tables and coefficients, fused multiply-adds where the library's build for
// FMA-capable processors has them), the results are the host's bit for bit: tests/cpp/pow_check.cpp
// compares hundreds of millions of (cosine, shininess) pairs.  Arguments outside the plain case
// (x zero, negative, infinite or NaN; y zero, infinite or NaN) have exactly defined results and go
// to the platform's own powf.  On the GPU this costs no more than the float routine did: some
// twenty double-precision operations and two table reads.
#pragma once
#include <cmath>
#include <cstdint>
#include <cstring>

#if defined(__HIPCC__)
#define RT_POW_HD __host__ __device__ __forceinline__
#define RT_POW_TABLE __device__ static const
#else
#define RT_POW_HD inline
#define RT_POW_TABLE static const
#endif

namespace rt {

struct PowTables {
    double invc[16], logc[16]; // 1/c and log2(c) for the sixteen subintervals of [0x1.66p-1, 0x1.66p0)
    double a[5];               // log1p(r)/ln2 = r * (a[4] + r * (...))
    uint64_t exp2[32];         // bits of 2^(i/32), the exponent field lowered by i << 47
};

RT_POW_HD const PowTables& powTables()
{
    static constexpr PowTables t = {
        { 0x1.661ec79f8f3bep+0, 0x1.571ed4aaf883dp+0, 0x1.49539f0f010b0p+0, 0x1.3c995b0b80385p+0,
        0x1.30d190c8864a5p+0, 0x1.25e227b0b8ea0p+0, 0x1.1bb4a4a1a343fp+0, 0x1.12358f08ae5bap+0,
        0x1.0953f419900a7p+0, 0x1.0000000000000p+0, 0x1.e608cfd9a47acp-1, 0x1.ca4b31f026aa0p-1,
        0x1.b2036576afce6p-1, 0x1.9c2d163a1aa2dp-1, 0x1.886e6037841edp-1, 0x1.767dcf5534862p-1 },
        { -0x1.efec65b963019p-2, -0x1.b0b6832d4fca4p-2, -0x1.7418b0a1fb77bp-2, -0x1.39de91a6dcf7bp-2,
        -0x1.01d9bf3f2b631p-2, -0x1.97c1d1b3b7af0p-3, -0x1.2f9e393af3c9fp-3, -0x1.960cbbf788d5cp-4,
        -0x1.a6f9db6475fcep-5, 0x0.0p+0, 0x1.338ca9f24f53dp-4, 0x1.476a9543891bap-3,
        0x1.e840b4ac4e4d2p-3, 0x1.40645f0c6651cp-2, 0x1.88e9c2c1b9ff8p-2, 0x1.ce0a44eb17bccp-2 },
        { 0x1.27616c9496e0bp-2, -0x1.71969a075c67ap-2, 0x1.ec70a6ca7baddp-2, -0x1.7154748bef6c8p-1, 0x1.71547652ab82bp+0 },
        { 0x3ff0000000000000ull, 0x3fefd9b0d3158574ull, 0x3fefb5586cf9890full, 0x3fef9301d0125b51ull,
        0x3fef72b83c7d517bull, 0x3fef54873168b9aaull, 0x3fef387a6e756238ull, 0x3fef1e9df51fdee1ull,
        0x3fef06fe0a31b715ull, 0x3feef1a7373aa9cbull, 0x3feedea64c123422ull, 0x3feece086061892dull,
        0x3feebfdad5362a27ull, 0x3feeb42b569d4f82ull, 0x3feeab07dd485429ull, 0x3feea47eb03a5585ull,
        0x3feea09e667f3bcdull, 0x3fee9f75e8ec5f74ull, 0x3feea11473eb0187ull, 0x3feea589994cce13ull,
        0x3feeace5422aa0dbull, 0x3feeb737b0cdc5e5ull, 0x3feec49182a3f090ull, 0x3feed503b23e255dull,
        0x3feee89f995ad3adull, 0x3feeff76f2fb5e47ull, 0x3fef199bdd85529cull, 0x3fef3720dcef9069ull,
        0x3fef5818dcfba487ull, 0x3fef7c97337b9b5full, 0x3fefa4afa2a490daull, 0x3fefd0765b6e4540ull },
    };
    return t;
}

RT_POW_HD float powLikeHost(float x, float y)
{
    uint32_t ix, iy;
    memcpy(&ix, &x, 4);
    memcpy(&iy, &y, 4);
    const bool ySpecial = 2u * iy - 1u >= 2u * 0x7f800000u - 1u; // zero, infinity or NaN
    if (ix - 0x00800000u >= 0x7f800000u - 0x00800000u || ySpecial) {
        if (!(x > 0.0f && ix < 0x00800000u && !ySpecial))
            return powf(x, y);
        const float scaled = x * 0x1p23f; // a positive subnormal: normalise it
        memcpy(&ix, &scaled, 4);
        ix &= 0x7fffffffu;
        ix -= 23u << 23;
    }
    const PowTables& T = powTables();
    // log2(x) = log1p(z/c - 1)/ln2 + log2(c) + k, with x = 2^k z and c the centre of z's subinterval
    const uint32_t tmp = ix - 0x3f330000u;
    const uint32_t i = (tmp >> 19) & 15u;
    const uint32_t top = tmp & 0xff800000u;
    const uint32_t iz = ix - top;
    const int k = (int32_t)top >> 23;
    float zf;
    memcpy(&zf, &iz, 4);
    const double r = fma((double)zf, T.invc[i], -1.0);
    const double y0 = T.logc[i] + (double)k;
    const double r2 = r * r;
    double l = fma(T.a[0], r, T.a[1]);
    const double p = fma(T.a[2], r, T.a[3]);
    const double r4 = r2 * r2;
    double q = fma(T.a[4], r, y0);
    q = fma(p, r2, q);
    l = fma(l, r4, q);
    const double ylogx = (double)y * l;
    uint64_t yb;
    memcpy(&yb, &ylogx, 8);
    if (((yb >> 47) & 0xffffu) >= (0x405f800000000000ull >> 47)) { // |y log2 x| >= 126
        if (ylogx > 0x1.fffffffd1d571p+6)
            return INFINITY;
        if (ylogx <= -150.0)
            return 0.0f;
    }
    // 2^(y log2 x) = 2^(n/32) * 2^rr, n = round(32 y log2 x) taken from the low bits of the shifted sum
    const double shift = 0x1.8p+47;
    double kd = ylogx + shift;
    uint64_t ki;
    memcpy(&ki, &kd, 8);
    kd -= shift;
    const double rr = ylogx - kd;
    const uint64_t sb = T.exp2[ki & 31u] + (ki << 47);
    double s;
    memcpy(&s, &sb, 8);
    const double z = fma(0x1.c6af84b912394p-5, rr, 0x1.ebfce50fac4f3p-3);
    const double rr2 = rr * rr;
    double o = fma(0x1.62e42ff0c52d6p-1, rr, 1.0);
    o = fma(z, rr2, o);
    return (float)(o * s);
}

} // namespace rt
