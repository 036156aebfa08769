// rt_fastdiv.hpp -- unsigned 32-bit division by a divisor fixed per launch, as a multiply-high and
// two shifts (the round-up method of Granlund & Montgomery in its branch-free form): the GPU has no
// integer divide, a general `n / d` expands to some twenty instructions, and the lane-per-item
// kernels each need three or four of them with per-launch divisors (rays per pixel, light samples
// per hit, lens samples, tiles per row).  Exact for every n < 2^32 and every d >= 1.
#pragma once
#include <cstdint>

#if defined(__HIPCC__)
#define RT_FD_HD __host__ __device__ __forceinline__
#else
#define RT_FD_HD inline
#endif

namespace rt {

struct FastDiv {
    uint32_t magic; // 0 for a power of two
    uint32_t shift;
    uint32_t d;     // the divisor itself (for the remainder); d == 1 is answered directly
};

inline FastDiv makeFastDiv(uint32_t d)
{
    FastDiv f { 0u, 0u, d };
    if (d <= 1u) // 0 never divides anything here; treated like 1
        return FastDiv { 0u, 0u, 1u };
    uint32_t log2d = 31u;
    while (!(d >> log2d))
        log2d--;
    if ((d & (d - 1u)) == 0u) { // power of two: q = (n >> 1) >> (log2d - 1)
        f.magic = 0u;
        f.shift = log2d - 1u;
        return f;
    }
    // magic = floor(2^(33 + log2d) / d) - 2^32 + 1, built from floor(2^(32 + log2d) / d) and its remainder
    const uint64_t num = (uint64_t)1 << (32u + log2d);
    uint32_t m = (uint32_t)(num / d);
    const uint32_t rem = (uint32_t)(num % d);
    m += m;
    const uint32_t twice = rem + rem;
    if (twice >= d || twice < rem)
        m += 1u;
    f.magic = m + 1u;
    f.shift = log2d;
    return f;
}

RT_FD_HD uint32_t fastDiv(uint32_t n, const FastDiv& f)
{
#if defined(__HIP_DEVICE_COMPILE__)
    const uint32_t q = __umulhi(f.magic, n);
#else
    const uint32_t q = (uint32_t)(((uint64_t)f.magic * n) >> 32);
#endif
    const uint32_t t = ((n - q) >> 1) + q;
    return f.d == 1u ? n : (t >> f.shift);
}

RT_FD_HD uint32_t fastMod(uint32_t n, const FastDiv& f) { return n - fastDiv(n, f) * f.d; }

} // namespace rt
