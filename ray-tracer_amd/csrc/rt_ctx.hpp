// rt_ctx.hpp -- what the host-side sources of the library share: the context behind the opaque rt_ctx of
// include/rt_amd.h, error reporting, and the helpers one source defines and another calls.  Nothing here is
// part of the C ABI.
//   rt_capi.cpp         context, options, lights, features, post-processing, BMP, single-ray and formula entry points
//   rt_capi_build.cpp   the host builders of the reference's hierarchy and of the library's own
//   rt_capi_upload.cpp  rt_scene_upload: the wire format checked and flattened into the device layout
//   rt_capi_render.cpp  the render entry points: pixel batches, working memory, launches, statistics
#pragma once

#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdarg>
#include <cstddef>
#include <cstdio>
#include <cstring>
#include <limits>
#include <string>
#include <memory>
#include <mutex>
#include <new>
#include <exception>
#include <vector>

#include "rt_device.hpp"
#include "rt_wavefront.hpp"
#include "rt_post.hpp"
#include "rt_build.hpp"
#include "rt_tree.hpp"

namespace rt {
hipError_t renderOccupancy(int variant, int blockThreads, size_t ldsBytes, int* blocksPerCU);
hipError_t launchRender(int variant, int blocks, int blockThreads, size_t ldsBytes, hipStream_t stream, const DevScene& sc,
    const DevCamera& cam, const DevFeat& ft, const DevFrame& fr, DevCounters* ctr, float* out, int minActive, int logicRounds);
hipError_t launchIntersect(int blockThreads, size_t ldsBytes, hipStream_t stream, const DevScene& sc, const DevFeat& ft,
    rt_ray* rays, rt_hit* hits, uint32_t n);
hipError_t launchDeinterleave(hipStream_t stream, const float* packed, float* frame, int width, int height, int tilesX,
    int tilesY, int nShard, size_t shardStride);
hipError_t launchEval(hipStream_t stream, const DevScene& sc, const DevCamera& cam, const DevFeat& ft, int kind,
    const float* in, float* out, uint32_t n);

} // namespace rt

using namespace rt;
using namespace rt; // (an internal header: every source that includes it is host code of this library)

constexpr size_t kScratchBytes = RT_NOALLOC_SCRATCH_BYTES; // the static scratch array of rt_render_device_noalloc (rt_capi.cpp)
constexpr int kMaxDevices = 64; // device ordinals the table of scratch addresses covers
constexpr size_t kMaxTimedLaunches = 4096; // event pairs (and batch quadruples) kept before they are folded into sums

struct rt_ctx {
    int device = 0;
    int numCU = 256;
    std::string error;
    // device buffers
    float* dBloom = nullptr; // thresholded copy of the frame for the bloom pass
    size_t bloomCap = 0;     // floats
    uint4* dPairs = nullptr;
    bool strictIntersect = false; // rt_intersect_batch replays every ray in the reference's order of tests
    uint4* dRefNodes = nullptr; // the caller's nodes as they came (replay of the reference's order of tests)
    float4* dLeafBoxes = nullptr; // per primitive, the box of the caller's leaf that holds it (Tracer::surelyReached)
    uint32_t numPairs = 0; // pair records in dPairs
    float4* dTris = nullptr;
    float4* dAttrs = nullptr;
    DevMaterial* dMats = nullptr;
    DevTex* dTexs = nullptr;
    float* dTexels = nullptr;
    DevSphere* dSpheres = nullptr;
    // Lights on the device: kLightSlots copies' worth of room in one allocation.  rt_lights_set writes the NEXT slot,
    // at once, over a stream of the context's own, and frames enqueued from then on are handed that slot's address;
    // frames enqueued before keep the slot they were given.  So a render call enqueues no copy from the caller's
    // (pageable) memory, which is what a HIP graph recorded from it could not replay, and re-sending the lights does
    // not wait for frames in flight.  A slot comes round again after kLightSlots calls; the event recorded behind the
    // last frame that read it is waited for first.
    static constexpr int kLightSlots = 4;
    rt_light* dLights = nullptr;
    uint32_t lightCap = 0; // lights per slot
    int lightSlot = 0;     // the slot the next frame reads
    hipEvent_t lightSlotUsed[kLightSlots] = {};
    bool lightSlotPending[kLightSlots] = {};
    hipStream_t copyStream = nullptr;
    std::vector<rt_light> hostLights;
    bool anyMirror = true, anyClear = true, anyMirrorAndClear = true; // what the scene's materials can ask of a hit
    DevCounters* dCounters = nullptr;
    float* dFrame = nullptr;
    size_t frameCap = 0;
    // five HIP events per render launch since the last stats read: start, after the path
    // kernel, before and after the shadow kernel, end (the middle three only on the
    // wavefront path, first batch)
    // timing: two events around every render (evPool), four per pixel batch of the wavefront
    // path (markPool: batch start, after the path kernel, before and after the shadow traversal)
    std::unique_ptr<DevCounters> hostCounters; // where readStats copies the device counters to
    std::vector<hipEvent_t> evPool, markPool;
    size_t evUsed = 0, marksUsed = 0;
    // what the events recorded so far have been folded into (foldTimings), until the next stats read
    double sumKernelMs = 0.0, sumPathMs = 0.0, sumShadowMs = 0.0;
    uint64_t sumLaunches = 0, sumBatches = 0;
    bool timing = true; // RT_OPT_TIMING: record events around the render launches
    DevScene scene {};
    bool haveScene = false;
    uint32_t numLevels = 1;
    uint64_t generation = 0;
    rt_features features {};
    bool haveFeatures = false;
    // tuning knobs of the render kernel (environment-overridable for experiments)
    int minActive = 32;       // per-pixel and path kernels: leave the traversal loop below this many rays in flight
    int minActiveShadow = 32; // shadow-ray kernel
    // switch points of the traversal loop between inner steps and triangle tests (rt_trace.hpp):
    // lanes blocked before triangles are tested; lanes still holding a leaf when inner steps resume
    int parkA = 16, parkB = 32, parkAShadow = 16, parkBShadow = 32;
    int chunkPath = 128, chunkShadow = 512; // rays a wavefront takes from the queue at a time
    int leastPath = 32;                     // the least the path kernel takes (small frames)
    int nodeBlockFixed = 0;                 // experiments: node blocks of full size whatever the frame's size
    int ownShare = 0;                       // sixteenths of a queue its waves split up by position (beyond the first batch)
    int logicRounds = 2;
    int blocksPerCU = 0; // 0 = ask the occupancy API
    int pathBlocksPerCU = 0, shadowBlocksPerCU = 0; // the same for the path / shadow kernel alone
    int forcePathPerCU = 0, forceShadowPerCU = 0; // experiments: launch this many blocks per CU whatever the query says
    int forcePerPixel = 0; // 1 = always use the per-pixel kernel
    // 0 = traverse the caller's inner nodes as they are; 1 = own SAH hierarchy over the caller's leaves;
    // 2 = own SAH hierarchy over the caller's primitives, each a leaf of its own under its own box
    int treeMode = 2;
    bool treeBuildOnDevice = false; // RT_OPT_TREE_BUILD: the library's own hierarchy by the device's binned builder
    bool referenceBoxes = true;     // RT_OPT_REFERENCE_BOXES: lose the hits the reference's box routine loses
    // What a frame's geometry and the memory at hand say about its pixel batches, kept from the last render call:
    // a renderer asked for the same frame again (a still camera, a benchmark) has nothing to work out per tile.
    struct Batch {
        uint32_t firstTile, tiles;
        uint64_t cand;
    };
    struct FramePlan {
        std::vector<uint8_t> tilesKey, planKey; // the inputs the two parts below were made from
        std::vector<uint16_t> tileCand;         // per tile of the shard: pixels that can see the scene's box
        std::vector<uint32_t> validBefore;      // per tile: real (non-padding) pixels of the tiles before it; one more entry at the end
        int rect[4] = { 0, 0, 0, 0 };           // x0, x1, y0, y1 (inclusive): the pixels that can see the scene's box
        std::vector<Batch> batches;
        uint64_t maxTiles = 0, maxCand = 0;
    } plan;
    size_t workspaceCap = 192ull << 30; // bytes the wavefront path may hold (two thirds of an MI355X; fewer, larger pixel batches: the 4K configuration 12.1 s against 12.9 s under 64 GiB)
    // working memory of the queue-based path (one allocation, grow-only)
    void* wsBase = nullptr;
    size_t wsBytes = 0;
    // this device's copy of the static scratch array (rt_render_device_noalloc); null on a device
    // ordinal beyond the table
    void* scratchPtr = nullptr;
    DevCounters* lastCounters = nullptr; // where the last render's counters live (dCounters or the scratch)
    size_t wsClamp = 0;                  // the working-memory limit a frame was last cut down to (0 = none) and the scene it held for
    uint64_t wsClampGeneration = 0;
    // resident blocks per compute unit of the kernels the current scene and features select, asked
    // of the runtime when either changes so that a render call has nothing left to ask
    struct Occupancy {
        bool valid = false;
        size_t lds = 0;
        int transp = -1, variant = -1;
        int path = 0, shadow = 0, render = 0;
    } occ;
};

namespace rt {

// rt_capi.cpp
int fail(rt_ctx* ctx, int code, const char* fmt, ...);
int caught(rt_ctx* ctx, const char* where) noexcept;

// rt_capi_build.cpp
bool buildOwnTree(const std::vector<TreeLeaf>& leaves, const QuantGrid& grid, std::vector<uint4>& pairs, uint32_t& rootRef,
    uint32_t& levels);

// rt_capi_render.cpp
int warmOccupancy(rt_ctx* ctx);
DevFeat makeFeat(const rt_features& f);
DevCamera makeCamera(const rt_camera& c);
size_t shardTiles(int width, int height, int shard, int nShard);
size_t traversalLds(const rt_ctx* ctx);
int bloomDevice(rt_ctx* ctx, float* dFrame, int width, int height, hipStream_t stream, float* copy = nullptr);
int readStats(rt_ctx* ctx, rt_stats* stats, bool timed);

} // namespace rt

#define HIP_TRY(ctx, expr)                                                                                  \
    do {                                                                                                    \
        hipError_t e_ = (expr);                                                                             \
        if (e_ != hipSuccess)                                                                               \
            return fail(ctx, RT_ERR_DEVICE, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
    } while (0)

