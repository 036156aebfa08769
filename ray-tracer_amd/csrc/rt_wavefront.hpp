// rt_wavefront.hpp -- buffers and launch interface of the queue-based renderer
// (rt_wavefront.hip), shared with the host side (rt_capi_render.cpp).
#pragma once

#include "rt_device.hpp"

namespace rt {

// One shaded hit.  A: what shading needs.  B: links and bookkeeping.  The hits of a camera
// ray form a tree: every hit may have a mirror child and, with transparency on, a pass-through
// child (src/recursive.cpp:43-63).
struct NodeB {
    uint32_t path;    // path index inside the batch
    uint32_t ordinal; // position of the hit among its camera ray's hits in the order renderRay visits them
                      // (glossy reflections, WfParams::absDraws: the position of the hit's light-sample draws in its pixel's
                      // stream, counted from the first such draw; childR = first glossy child, childP = next sibling;
                      // path == kDeadNode with cvR set: a child that left the scene, carrying the environment's colour)
    int32_t depth;    // rayDepth of the ray that made the hit
    int32_t childR;   // mirror child: kChildNone, kChildMiss (its value in cvR), or node index
    int32_t childP;   // pass-through child, likewise (cvP)
    float cvR[3];
    float cvP[3];     // (without transparency: the hit's ks, read by the combine)
    float tr;         // material transparency
    float mkd[3];     // material kd (not the texture's): where the transparent shadow test starts, src/light.cpp:126
    uint32_t pad;     // (glossy reflections with transparency: the pass-through child, as childR / childP elsewhere)
};
static_assert(sizeof(NodeB) == 64, "NodeB is one 64-B record");
constexpr int32_t kChildNone = -1, kChildMiss = -2;
constexpr int32_t kPathMiss = -1;
constexpr uint32_t kDeadNode = 0xffffffffu; // NodeB::path of a pool slot no hit was written to
constexpr uint32_t kNodeBlock = 256;         // node slots a wavefront reserves at a time (>= 64)

struct WfBuffers {
    int32_t* pathHead;   // [paths]  first node or kPathMiss
    uint8_t* pathCount;  // [paths]  nodes of the path
    uint32_t* pathBase;  // [paths]  nodes of earlier paths of the same pixel
    float* pathEnv;      // [paths*3] environment colour of a camera ray that missed (env map on)
    float4* nodeA;       // [nodes*4] P.xyz N.x | N.yz rd.xy | rd.z kd.xyz | ks.xyz shininess
    NodeB* nodeB;        // [nodes]
    float* contrib;      // [nodes * slots * 3]
    float4* srays;       // [nodes * slots * 3] shadow ray: origin.xyz length | direction.xyz 0 | light colour.xyz 0
    uint8_t* svis;       // [nodes * slots]     1 = light sample visible (transparency only)
    uint32_t* redo;      // [nodes * slots]     without transparency, in svis' place: samples to be decided by the replay (shadowRedoKernel)
    float4* camRays;     // [paths*2] camera rays that need tracing: origin.xyz path(bits) | direction.xyz enters-the-scene-box flag
    uint32_t maxNodes;
};

// Motion blur (src/extra.cpp:121-211; DevFeat::motion, ::motionSamples = M): M time samples per pixel, each rendered like
// a pixel of its own -- its own stretch of the pixel's sampler stream, its own copy of the scene -- and the pixel's colour
// is their mean.  Every (pixel slot, time sample) pair is a "sample slot", pair = slot * M + m, and takes a pixel slot's
// place wherever paths are indexed: path = pair * R + camera ray.  Without motion blur a sample slot is a pixel slot.
// (Nothing about it is kept in WfParams: the struct sits in the kernels' argument block ahead of WfBuffers, and the
// path kernel's register allocation did not take its growing well.)
struct WfParams {
    uint32_t slotBegin;  // first local pixel slot of this batch
    uint32_t nSlots;     // pixel slots in this batch (multiple of 64)
    uint32_t R;          // camera rays per pixel
    uint32_t slots;      // light-sample slots per hit: sum over lights of (point ? 1 : shadowSamples)
    FastDiv divR, divSlots, divPerRound; // by R, by max(slots, 1), by min(R, 256)
    uint32_t drawsPerHit;
    uint32_t pixDraws;   // draws of the pixel-sample stretch
    uint32_t mainStart;  // position of the first light-sample draw
    uint32_t chunk;      // shadow rays per queue fetch, at most
    uint32_t chunkPath;  // camera rays per queue fetch, at most
    uint32_t leastPath;  // and at least (small frames)
    uint32_t nodeBlockFixed; // experiments: 1 = node blocks of nodeBlockMax slots whatever the frame's size
    uint32_t nodeBlockMax;   // node slots a wavefront reserves at a time, at most (64 <= power of two <= kNodeBlock)
    uint32_t ownShare;   // sixteenths of a queue that its waves split up by position instead of fetching
    uint32_t validPaths; // camera rays of the batch's real (non-padding) pixels
    uint32_t absDraws;   // 1: NodeB::ordinal is the position of the hit's light-sample draws itself (glossy reflections)
};

hipError_t wavefrontOccupancy(bool stats, bool transp, bool motion, int blockThreads, size_t ldsBytes, int* pathBlocks, int* shadowBlocks);
hipError_t launchWavefront(bool dof, bool stats, bool transp, bool motion, int pathBlocks, int shadowBlocks, int blockThreads, size_t ldsBytes,
    hipStream_t stream, const DevScene& sc, const DevCamera& cam, const DevFeat& ft, const DevFrame& fr, const WfParams& wp,
    const WfBuffers& wb, DevCounters* ctr, float* out, int minActive, int minActiveShadow, hipEvent_t* marks, int numCU);

hipError_t glossyOccupancy(bool dof, bool stats, bool motion, bool transp, int blockThreads, size_t ldsBytes, int* treeBlocks);
hipError_t launchWavefrontGlossy(bool dof, bool stats, bool motion, bool transp, int treeBlocks, int shadowBlocks, int blockThreads, size_t ldsBytes, hipStream_t stream,
    const DevScene& sc, const DevCamera& cam, const DevFeat& ft, const DevFrame& fr, const WfParams& wp, const WfBuffers& wb, DevCounters* ctr,
    float* out, int minActive, int minActiveShadow, hipEvent_t* marks, int numCU);

} // namespace rt
