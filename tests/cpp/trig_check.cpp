// rt_trig.hpp against the C library's sinf / cosf: every float from 0 up to just past 2 pi.
#include "../../ray-tracer_amd/csrc/rt_trig.hpp"
#include <cmath>
#include <cstdio>
#include <cstring>

int main()
{
    const float twoPi = 6.28318530717958647692528676655900576f;
    uint32_t last;
    std::memcpy(&last, &twoPi, 4);
    last += 16u;
    long bad = 0, total = 0;
    for (uint32_t u = 0; u <= last; u++) {
        float y;
        std::memcpy(&y, &u, 4);
        const float s = rt::sinLikeHost(y), c = rt::cosLikeHost(y);
        const float rs = std::sin(y), rc = std::cos(y);
        if (std::memcmp(&s, &rs, 4) != 0 || std::memcmp(&c, &rc, 4) != 0) {
            if (bad++ < 5)
                std::printf("wrong at %a: sin %a (libm %a), cos %a (libm %a)\n", y, s, rs, c, rc);
        }
        total++;
    }
    std::printf("%ld arguments, %ld wrong\n", total, bad);
    return bad != 0;
}
