// rt_pow.hpp against the C library's powf: cosines in [0, 1] (also bunched towards 0 and towards 1)
// raised to shininess values of every kind a material file holds, and the special arguments.
#include "../../ray-tracer_amd/csrc/rt_pow.hpp"
#include <cmath>
#include <cstdio>
#include <cstring>

int main()
{
    uint32_t st = 777u;
    long bad = 0, total = 0;
    auto next = [&]() { st = st * 747796405u + 2891336453u; return (float)(st >> 8) * (1.0f / 16777216.0f); };
    auto check = [&](float x, float y) {
        const float a = rt::powLikeHost(x, y), b = std::pow(x, y);
        if (std::memcmp(&a, &b, 4) != 0 && !(a != a && b != b)) {
            if (bad++ < 8)
                std::printf("wrong: pow(%a, %a) = %a, libm %a\n", x, y, a, b);
        }
        total++;
    };
    for (long it = 0; it < 200000000L; it++) {
        float x = next();
        const float u = next();
        float y;
        switch (it & 3) {
        case 0: y = u * 100.0f; break;
        case 1: y = 1.0f + std::floor(u * 64.0f); break;
        case 2: y = u * 2000.0f; x = 1.0f - x * x * 0.25f; break;
        default: y = u * 10.0f; x = x * x * x * x; break;
        }
        check(x, y);
    }
    const float sp[] = { 0.0f, -0.0f, 1.0f, -1.0f, 0.5f, 2.0f, 1e-40f, 1e-45f, 3.4e38f, INFINITY, -INFINITY, NAN, 10.000002f, 96.078431f };
    for (float x : sp)
        for (float y : sp)
            check(x, y);
    std::printf("%ld pairs, %ld wrong\n", total, bad);
    return 0;
}
