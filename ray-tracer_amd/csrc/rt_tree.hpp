// rt_tree.hpp -- the library's own traversal hierarchy: the items it is built over, the grid its boxes are
// quantised on, and the device-side builder's interface (rt_tree.hip).  The host-side builder (full-sweep
// surface-area heuristic) lives in rt_capi_build.cpp.
#pragma once

#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <string>
#include <vector>

namespace rt {

// One leaf of the hierarchy: a primitive of its own, or a leaf of the caller's tree with its primitive range.
struct TreeLeaf {
    float lo[3], hi[3], c[3];
    uint32_t ref, count; // child reference the pair records carry for it; primitives it holds
};

// Quantisation grid: 65535 cells per axis across the root box.  A plane q stands
// for the real number gridMin + q * gridCell (both floats); lower planes round
// down, upper planes round up, checked in double so the quantised box contains
// the float box whatever the rounding of the division did.
struct QuantGrid {
    float gridMin[3], gridCell[3];
    __host__ __device__ uint32_t quantLo(float v, int a) const
    {
        if (!(gridCell[a] > 0.0f))
            return 0u;
        double q = floor((static_cast<double>(v) - gridMin[a]) / gridCell[a]);
        q = q < 0.0 ? 0.0 : (q > 65535.0 ? 65535.0 : q);
        while (q > 0.0 && static_cast<double>(gridMin[a]) + q * gridCell[a] > static_cast<double>(v))
            q -= 1.0;
        return static_cast<uint32_t>(q);
    }
    __host__ __device__ uint32_t quantHi(float v, int a) const
    {
        if (!(gridCell[a] > 0.0f))
            return 65535u; // flat or empty axis: A = 0, every plane sits at gridMin
        double q = ceil((static_cast<double>(v) - gridMin[a]) / gridCell[a]);
        q = q < 0.0 ? 0.0 : (q > 65535.0 ? 65535.0 : q);
        while (q < 65535.0 && static_cast<double>(gridMin[a]) + q * gridCell[a] < static_cast<double>(v))
            q += 1.0;
        return static_cast<uint32_t>(q);
    }
    __host__ __device__ void packBox(const float* lo, const float* hi, uint32_t* w) const
    {
        for (int a = 0; a < 3; a++)
            w[a] = quantLo(lo[a], a) | (quantHi(hi[a], a) << 16);
    }
};

// The hierarchy over `leaves` built on the current HIP device with the binned surface-area heuristic
// (32 bins per axis over the node's box, all nodes of a level at once): sibling-pair records in
// breadth-first order, as buildOwnTree makes them.  kTreeUnusable: the tree came out in a shape the
// traversal cannot take (deeper than its stack), the caller builds another way; kTreeFailed: a HIP call
// failed.  `error` says what happened in either case.
enum : int { kTreeBuilt = 0, kTreeUnusable = 1, kTreeFailed = 2 };
int deviceOwnTree(const std::vector<TreeLeaf>& leaves, const QuantGrid& grid, std::vector<uint4>& pairs, uint32_t& rootRef,
    uint32_t& levels, std::string& error);

} // namespace rt
