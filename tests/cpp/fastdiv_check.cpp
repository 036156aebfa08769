// Exhaustive-in-d, sampled-in-n check of rt_fastdiv.hpp against the machine's own division.
#include "../../ray-tracer_amd/csrc/rt_fastdiv.hpp"
#include <cstdio>
#include <random>

int main()
{
    std::mt19937 gen(1);
    long bad = 0;
    auto check = [&](uint32_t n, uint32_t d) {
        const rt::FastDiv f = rt::makeFastDiv(d);
        if (rt::fastDiv(n, f) != n / d || rt::fastMod(n, f) != n % d) {
            if (bad++ < 5)
                std::printf("wrong: %u / %u\n", n, d);
        }
    };
    for (uint32_t d = 1; d <= 70000; d++) {
        const uint32_t edge[] = { 0u, 1u, d - 1u, d, d + 1u, 2u * d - 1u, 2u * d, 0x7fffffffu, 0x80000000u, 0xfffffffeu, 0xffffffffu,
            0xffffffffu / d * d, 0xffffffffu / d * d - 1u };
        for (uint32_t n : edge)
            check(n, d);
        for (int k = 0; k < 100; k++)
            check(gen(), d);
    }
    for (int k = 0; k < 1000000; k++) { // divisors of every magnitude
        uint32_t d = gen() >> (gen() % 32u);
        check(gen(), d ? d : 1u);
    }
    std::printf("%ld wrong\n", bad);
    return bad != 0;
}
