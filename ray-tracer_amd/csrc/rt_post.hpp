// rt_post.hpp -- launch interface of the frame post-processing kernels (rt_post.hip).
#pragma once

#include <cstdint>
#include <hip/hip_runtime.h>

namespace rt {

constexpr int kBloomMinSize = 3;  // below 3 the reference's loop writes outside the image
constexpr int kBloomMaxSize = 20; // its factorial() overflows a long long from 21 on

struct BloomWeights {
    float w[kBloomMaxSize];
};

// computeGaussianFilter, src/extra.cpp:239-271: the second (vertical) pass overwrites every
// entry the first one wrote, so filter[x][y] = C(k, y) / sum_{y' < k} C(k, y') whatever x is
// (each column sums to one, the whole filter to k).  Host arithmetic as there: the binomial in
// long long integer division, the sum and the quotient in float.
inline void bloomWeights(int k, BloomWeights& out)
{
    auto factorial = [](int n) {
        long long f = 1;
        for (int i = 1; i <= n; i++)
            f *= i;
        return f;
    };
    float total = 0;
    for (int y = 0; y < k; y++) {
        const float val = float(factorial(k) / (factorial(y) * factorial(k - y)));
        total += val;
        out.w[y] = val;
    }
    for (int y = 0; y < k; y++)
        out.w[y] /= total;
    for (int y = k; y < kBloomMaxSize; y++)
        out.w[y] = 0.0f;
}

hipError_t launchBloom(hipStream_t stream, float* frame, float* bright, int W, int H, int k, float threshold, const BloomWeights& w);
hipError_t launchPackBmp(hipStream_t stream, const float* frame, uint8_t* out, int W, int H);

} // namespace rt
