// rt_build.hpp -- interface of the device-side BVH build (rt_build.hip).
#pragma once

#include <cstdint>
#include <string>

#include "../../include/rt_amd.h"

namespace rt {

struct DeviceBuildStats {
    uint32_t levels = 0;
    uint32_t rangesSortedOnHost = 0;     // ranges with equal coordinates, re-sorted by the host's std::sort
    uint32_t primitivesSortedOnHost = 0; // primitives in those ranges
};

// BVH::BVH with the median split (src/bvh.cpp:52-101, :208-223, :424-493) on the current HIP device;
// same outputs as rt_bvh_build, bit for bit.  False with `error` set on failure.
bool deviceMedianBuild(const rt_primitive* in, uint32_t n, rt_node* nodesOut, rt_primitive* primsOut, uint32_t* nNodes,
    uint32_t* nLevels, uint32_t* nLeaves, DeviceBuildStats* stats, std::string& error);

// The same with splitPrimitivesBySAHBin (src/extra.cpp:493-543) as the split; same outputs as rt_bvh_build with
// sah_binning set, bit for bit.
bool deviceSahBinBuild(const rt_primitive* in, uint32_t n, rt_node* nodesOut, rt_primitive* primsOut, uint32_t* nNodes,
    uint32_t* nLevels, uint32_t* nLeaves, DeviceBuildStats* stats, std::string& error);

} // namespace rt
