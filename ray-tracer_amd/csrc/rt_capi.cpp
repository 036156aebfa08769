// rt_capi.cpp -- the C ABI of include/rt_amd.h: context, host-side BVH build,
// flattening of the reference's wire format into the device layout, and the
// launches.  Built with hipcc (-ffp-contract=off) so the per-triangle
// constants computed here carry the same roundings as the scalar code they
// stand in for.

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

static thread_local std::string g_lastError;

// Scratch of the entry point that allocates nothing (rt_render_device_noalloc): counters, queues,
// the hit and light-sample pools of one pixel batch and the bloom copy live here instead of in
// buffers from hipMalloc.  Every GPU the library runs on holds its own copy of the array, at its
// own address, until the process ends; host code never uses the array's name as an address but the
// per-device pointer hipGetSymbolAddress returned when the first context on that device was made.
constexpr size_t kScratchBytes = RT_NOALLOC_SCRATCH_BYTES;
static __device__ __align__(256) unsigned char scratch[kScratchBytes];
constexpr int kMaxDevices = 64;
static void* g_scratchPtr[kMaxDevices] = {};
static std::mutex g_scratchMutex;

// The device address of `scratch` on `device`, which must be the current device.
static hipError_t scratchOnDevice(int device, void** out)
{
    std::lock_guard<std::mutex> lock(g_scratchMutex);
    if (device < 0 || device >= kMaxDevices)
        return hipErrorInvalidDevice;
    if (!g_scratchPtr[device]) {
        void* p = nullptr;
        const hipError_t e = hipGetSymbolAddress(&p, HIP_SYMBOL(scratch));
        if (e != hipSuccess)
            return e;
        g_scratchPtr[device] = p;
    }
    *out = g_scratchPtr[device];
    return hipSuccess;
}

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

static int fail(rt_ctx* ctx, int code, const char* fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    if (ctx)
        ctx->error = buf;
    g_lastError = buf;
    return code;
}

// No C++ exception leaves the library (its callers bind a C ABI: ctypes, cgo, a C program): what a
// std::vector or std::string throws inside an entry point becomes a status like any other failure.
static int caught(rt_ctx* ctx, const char* where) noexcept
{
    int code = RT_ERR_INTERNAL;
    try {
        try {
            throw;
        } catch (const std::bad_alloc&) {
            code = RT_ERR_NOMEM;
            return fail(ctx, code, "%s: out of host memory", where);
        } catch (const std::exception& e) {
            return fail(ctx, code, "%s: %s", where, e.what());
        } catch (...) {
            return fail(ctx, code, "%s: unknown failure", where);
        }
    } catch (...) { // not even the message could be stored
        return code;
    }
}

#define HIP_TRY(ctx, expr)                                                                                  \
    do {                                                                                                    \
        hipError_t e_ = (expr);                                                                             \
        if (e_ != hipSuccess)                                                                               \
            return fail(ctx, RT_ERR_DEVICE, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
    } while (0)

static int warmOccupancy(rt_ctx* ctx);

constexpr size_t kMaxTimedLaunches = 4096; // event pairs (and batch quadruples) kept before they are folded into sums

// Adds the elapsed times between the events recorded so far to the context's running sums and
// hands the events back for reuse.  The work they bracket must have finished.
static int foldTimings(rt_ctx* ctx)
{
    for (size_t i = 0; i + 1 < ctx->evUsed; i += 2) {
        float ms = 0.0f;
        HIP_TRY(ctx, hipEventElapsedTime(&ms, ctx->evPool[i], ctx->evPool[i + 1]));
        ctx->sumKernelMs += ms;
        ctx->sumLaunches++;
    }
    for (size_t i = 0; i + 3 < ctx->marksUsed; i += 4) {
        float ms = 0.0f;
        HIP_TRY(ctx, hipEventElapsedTime(&ms, ctx->markPool[i], ctx->markPool[i + 1]));
        ctx->sumPathMs += ms;
        HIP_TRY(ctx, hipEventElapsedTime(&ms, ctx->markPool[i + 2], ctx->markPool[i + 3]));
        ctx->sumShadowMs += ms;
        ctx->sumBatches++;
    }
    ctx->evUsed = ctx->marksUsed = 0;
    return RT_OK;
}

// Behind the launches of a frame: marks the slot of lights they read as in use until they have run (see rt_ctx).
// Nothing is recorded on a stream that is being captured into a graph: the graph replays with the slot it was
// recorded with, and whoever replays it keeps the lights as they were.
static int lightsUsedBy(rt_ctx* ctx, hipStream_t stream)
{
    if (!ctx->dLights)
        return RT_OK;
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    if (stream != nullptr && hipStreamIsCapturing(stream, &cap) == hipSuccess && cap != hipStreamCaptureStatusNone)
        return RT_OK;
    hipEvent_t ev = ctx->lightSlotUsed[ctx->lightSlot]; // (made by rt_lights_set: a render call creates nothing)
    if (!ev)
        return RT_OK;
    HIP_TRY(ctx, hipEventRecord(ev, stream));
    ctx->lightSlotPending[ctx->lightSlot] = true;
    return RT_OK;
}

template <typename T>
static hipError_t uploadVec(T** dptr, const std::vector<T>& v)
{
    if (*dptr) {
        (void)hipFree(*dptr);
        *dptr = nullptr;
    }
    const size_t bytes = std::max<size_t>(v.size(), 1) * sizeof(T);
    hipError_t e = hipMalloc(reinterpret_cast<void**>(dptr), bytes);
    if (e != hipSuccess)
        return e;
    if (!v.empty())
        e = hipMemcpy(*dptr, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice);
    return e;
}

// Scheduling parameters of the traversing kernels, overridable from the environment for the sweeps under
// tools/ (only when RT_AMD_EXPERIMENTS is set).  What changes results or is of use to a caller is an
// rt_option instead.
static void readExperimentKnobs(rt_ctx* ctx)
{
    struct Knob {
        const char* name;
        int* a;
        int* b; // second field set to the same value, or null
        int lo, hi;
    };
    const Knob knobs[] = {
        { "RT_AMD_MIN_ACTIVE", &ctx->minActive, &ctx->minActiveShadow, 0, 64 },
        { "RT_AMD_MIN_ACTIVE_SHADOW", &ctx->minActiveShadow, nullptr, 0, 64 },
        { "RT_AMD_PARK_A", &ctx->parkA, &ctx->parkAShadow, 0, 255 },
        { "RT_AMD_PARK_B", &ctx->parkB, &ctx->parkBShadow, 0, 255 },
        { "RT_AMD_PARK_A_SHADOW", &ctx->parkAShadow, nullptr, 0, 255 },
        { "RT_AMD_PARK_B_SHADOW", &ctx->parkBShadow, nullptr, 0, 255 },
        { "RT_AMD_CHUNK_PATH", &ctx->chunkPath, nullptr, 64, 1 << 20 },
        { "RT_AMD_CHUNK_SHADOW", &ctx->chunkShadow, nullptr, 64, 1 << 20 },
        { "RT_AMD_NODE_BLOCK_FIXED", &ctx->nodeBlockFixed, nullptr, 0, 1 },
        { "RT_AMD_LEAST_PATH", &ctx->leastPath, nullptr, 1, 64 },
        { "RT_AMD_OWN_SHARE", &ctx->ownShare, nullptr, 0, 16 },
        { "RT_AMD_LOGIC_ROUNDS", &ctx->logicRounds, nullptr, 1, 64 },
        { "RT_AMD_BLOCKS_PER_CU", &ctx->blocksPerCU, nullptr, 0, 32 },
        { "RT_AMD_PATH_BLOCKS_PER_CU", &ctx->pathBlocksPerCU, nullptr, -1, 32 },
        { "RT_AMD_SHADOW_BLOCKS_PER_CU", &ctx->shadowBlocksPerCU, nullptr, 0, 32 },
        { "RT_AMD_FORCE_PATH_PER_CU", &ctx->forcePathPerCU, nullptr, 0, 32 },
        { "RT_AMD_FORCE_SHADOW_PER_CU", &ctx->forceShadowPerCU, nullptr, 0, 32 },
    };
    for (const Knob& k : knobs)
        if (const char* s = getenv(k.name)) {
            const int v = std::min(k.hi, std::max(k.lo, atoi(s)));
            *k.a = v;
            if (k.b)
                *k.b = v;
        }
}

extern "C" {

const char* rt_last_error(const rt_ctx* ctx) { return ctx ? ctx->error.c_str() : g_lastError.c_str(); }

int rt_ctx_create(int device_ordinal, rt_ctx** out)
try {
    if (!out)
        return fail(nullptr, RT_ERR_INVALID, "rt_ctx_create: out is NULL");
    *out = nullptr;
    int count = 0;
    HIP_TRY(nullptr, hipGetDeviceCount(&count));
    if (device_ordinal < 0 || device_ordinal >= count)
        return fail(nullptr, RT_ERR_INVALID, "rt_ctx_create: device %d not in [0, %d)", device_ordinal, count);
    HIP_TRY(nullptr, hipSetDevice(device_ordinal));
    hipDeviceProp_t prop;
    HIP_TRY(nullptr, hipGetDeviceProperties(&prop, device_ordinal));
    rt_ctx* ctx = new rt_ctx;
    ctx->device = device_ordinal;
    ctx->numCU = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    if (hipMalloc(reinterpret_cast<void**>(&ctx->dCounters), sizeof(DevCounters)) != hipSuccess) {
        delete ctx;
        return fail(nullptr, RT_ERR_DEVICE, "rt_ctx_create: device allocation failed");
    }
    ctx->lastCounters = ctx->dCounters;
    if (device_ordinal < kMaxDevices) { // this device's address of the static scratch array, once per device
        const hipError_t e = scratchOnDevice(device_ordinal, &ctx->scratchPtr);
        if (e != hipSuccess) {
            (void)hipFree(ctx->dCounters);
            delete ctx;
            return fail(nullptr, RT_ERR_DEVICE, "rt_ctx_create: hipGetSymbolAddress(scratch) failed: %s", hipGetErrorString(e));
        }
    }
    if (getenv("RT_AMD_EXPERIMENTS")) // scheduling experiments (tools/*.sh); nothing a caller needs
        readExperimentKnobs(ctx);
    *out = ctx;
    return RT_OK;
} catch (...) {
    return caught(nullptr, "rt_ctx_create");
}

int rt_ctx_set_option(rt_ctx* ctx, int option, int64_t value)
try {
    if (!ctx)
        return fail(nullptr, RT_ERR_INVALID, "rt_ctx_set_option: ctx is NULL");
    switch (option) {
    case RT_OPT_TREE:
        if (value < 0 || value > 2)
            return fail(ctx, RT_ERR_INVALID, "rt_ctx_set_option: RT_OPT_TREE takes 0, 1 or 2");
        ctx->treeMode = static_cast<int>(value);
        return RT_OK;
    case RT_OPT_PER_PIXEL:
        ctx->forcePerPixel = value != 0;
        return RT_OK;
    case RT_OPT_STRICT_INTERSECT:
        ctx->strictIntersect = value != 0;
        ctx->scene.strictIntersect = ctx->strictIntersect ? 1 : 0;
        return RT_OK;
    case RT_OPT_WORKSPACE_BYTES:
        if (value < (1 << 20))
            return fail(ctx, RT_ERR_INVALID, "rt_ctx_set_option: RT_OPT_WORKSPACE_BYTES takes at least 1 MiB");
        ctx->workspaceCap = static_cast<size_t>(value);
        return RT_OK;
    case RT_OPT_TIMING:
        ctx->timing = value != 0;
        return RT_OK;
    case RT_OPT_TREE_BUILD:
        ctx->treeBuildOnDevice = value != 0;
        return RT_OK;
    case RT_OPT_REFERENCE_BOXES:
        ctx->referenceBoxes = value != 0;
        ctx->scene.leafBoxes = ctx->referenceBoxes ? ctx->dLeafBoxes : nullptr;
        return RT_OK;
    default:
        return fail(ctx, RT_ERR_INVALID, "rt_ctx_set_option: unknown option %d", option);
    }
} catch (...) {
    return caught(ctx, "rt_ctx_set_option");
}

void rt_ctx_destroy(rt_ctx* ctx)
{
    if (!ctx)
        return;
    (void)hipSetDevice(ctx->device);
    (void)hipDeviceSynchronize();
    void* ptrs[] = { ctx->dPairs, ctx->dRefNodes, ctx->dLeafBoxes, ctx->dTris, ctx->dAttrs, ctx->dMats, ctx->dTexs, ctx->dTexels, ctx->dSpheres, ctx->dLights,
        ctx->dCounters, ctx->dFrame, ctx->dBloom };
    for (void* p : ptrs)
        if (p)
            (void)hipFree(p);
    if (ctx->wsBase)
        (void)hipFree(ctx->wsBase);
    for (hipEvent_t e : ctx->lightSlotUsed)
        if (e)
            (void)hipEventDestroy(e);
    if (ctx->copyStream)
        (void)hipStreamDestroy(ctx->copyStream);
    for (hipEvent_t e : ctx->markPool)
        (void)hipEventDestroy(e);
    for (hipEvent_t e : ctx->evPool)
        (void)hipEventDestroy(e);
    delete ctx;
}

// ---------------------------------------------------------------------------
// Host BVH build.  Same tree as BVH::BVH (src/bvh.cpp:52-101, :424-493): leaf
// at <= 4 primitives, split along the longest axis of the span box at
// (n+1)/2 after sorting by centroid (:208-223), children allocated left then
// right, leaves appended in depth-first order (:369-387).  The sort runs on
// (centroid key, index) pairs with std::sort: the permutation a comparison
// sort produces depends only on the comparison results, so this is the
// permutation std::ranges::sort yields on the 100-byte primitives themselves
// with the same toolchain, ties included, at a fraction of the data movement.
// ---------------------------------------------------------------------------
namespace {
struct BuildPrim {
    float lo[3], hi[3], c[3];
};
struct KeyIdx {
    float key;
    uint32_t idx;
};

// ---------------------------------------------------------------------------
// The library's own traversal hierarchy (see rt_scene_upload)
// ---------------------------------------------------------------------------
// Full-sweep SAH build in O(n log n): the leaves are sorted by centre once per axis; a
// node owns the same range [b, e) of all three orders, sweeps each for the cheapest split,
// and hands its halves down by a stable partition of the other two orders.
struct TreeBuild {
    const std::vector<TreeLeaf>& L;
    std::vector<uint4>& pairs;
    const QuantGrid& grid;
    std::vector<uint32_t> order[3], tmp, suffixCount;
    std::vector<uint8_t> goesLeft;
    std::vector<float> rightArea;
    uint32_t maxDepth = 0;

    TreeBuild(const std::vector<TreeLeaf>& leaves, std::vector<uint4>& out, const QuantGrid& g)
        : L(leaves), pairs(out), grid(g)
    {
    }
    float area(const float* lo, const float* hi) const
    {
        const float x = hi[0] - lo[0], y = hi[1] - lo[1], z = hi[2] - lo[2];
        return 2.0f * (x * y + x * z + y * z);
    }
    void pack(const float* lo, const float* hi, uint32_t* w) const { grid.packBox(lo, hi, w); }
    void bounds(uint32_t b, uint32_t e, float* lo, float* hi) const
    {
        const std::vector<uint32_t>& idx = order[0];
        for (int a = 0; a < 3; a++)
            lo[a] = L[idx[b]].lo[a], hi[a] = L[idx[b]].hi[a];
        for (uint32_t i = b + 1; i < e; i++)
            for (int a = 0; a < 3; a++) {
                lo[a] = std::min(lo[a], L[idx[i]].lo[a]);
                hi[a] = std::max(hi[a], L[idx[i]].hi[a]);
            }
    }
    void start()
    {
        const uint32_t n = static_cast<uint32_t>(L.size());
        for (int axis = 0; axis < 3; axis++) {
            order[axis].resize(n);
            for (uint32_t i = 0; i < n; i++)
                order[axis][i] = i;
            std::sort(order[axis].begin(), order[axis].end(), [&](uint32_t x, uint32_t y) {
                return L[x].c[axis] < L[y].c[axis] || (L[x].c[axis] == L[y].c[axis] && x < y);
            });
        }
        goesLeft.assign(n, 0);
        tmp.resize(n);
        rightArea.resize(n);
        suffixCount.resize(n + 1);
    }
    // returns the child reference of the subtree over the range [b, e) of the three orders
    uint32_t build(uint32_t b, uint32_t e, uint32_t depth)
    {
        maxDepth = std::max(maxDepth, depth);
        if (e - b == 1)
            return L[order[0][b]].ref;
        const uint32_t n = e - b;
        float bestCost = std::numeric_limits<float>::infinity();
        int bestAxis = -1;
        uint32_t bestSplit = 0;
        for (int axis = 0; axis < 3; axis++) {
            const uint32_t* idx = order[axis].data() + b;
            float lo[3], hi[3];
            uint32_t cnt = 0;
            suffixCount[n] = 0;
            for (uint32_t i = n; i-- > 0;) { // suffix boxes and primitive counts
                const TreeLeaf& q = L[idx[i]];
                suffixCount[i] = suffixCount[i + 1] + q.count;
                if (i == n - 1)
                    for (int a = 0; a < 3; a++)
                        lo[a] = q.lo[a], hi[a] = q.hi[a];
                else
                    for (int a = 0; a < 3; a++)
                        lo[a] = std::min(lo[a], q.lo[a]), hi[a] = std::max(hi[a], q.hi[a]);
                rightArea[i] = area(lo, hi);
            }
            for (uint32_t i = 0; i + 1 < n; i++) { // split after element i
                const TreeLeaf& q = L[idx[i]];
                if (i == 0)
                    for (int a = 0; a < 3; a++)
                        lo[a] = q.lo[a], hi[a] = q.hi[a];
                else
                    for (int a = 0; a < 3; a++)
                        lo[a] = std::min(lo[a], q.lo[a]), hi[a] = std::max(hi[a], q.hi[a]);
                cnt += q.count;
                const float cost = area(lo, hi) * (float)std::max(1u, cnt) + rightArea[i + 1] * (float)std::max(1u, suffixCount[i + 1]);
                if (cost < bestCost)
                    bestCost = cost, bestAxis = axis, bestSplit = i + 1;
            }
        }
        // keep the depth bounded: fall back to the middle once a branch runs deep
        if (bestAxis < 0 || depth + 2 + (uint32_t)std::ceil(std::log2((double)n)) > (uint32_t)kMaxStack) {
            bestAxis = 0;
            bestSplit = n / 2;
        }
        const uint32_t mid = b + bestSplit;
        for (uint32_t i = b; i < e; i++)
            goesLeft[order[bestAxis][i]] = i < mid;
        for (int axis = 0; axis < 3; axis++) {
            if (axis == bestAxis)
                continue;
            uint32_t* idx = order[axis].data();
            uint32_t nl = b, nr = 0;
            for (uint32_t i = b; i < e; i++) {
                if (goesLeft[idx[i]])
                    idx[nl++] = idx[i];
                else
                    tmp[nr++] = idx[i];
            }
            std::copy(tmp.begin(), tmp.begin() + nr, idx + nl);
        }
        const size_t me = pairs.size();
        pairs.resize(me + 2);
        float lo[3], hi[3];
        uint32_t wl[3], wr[3];
        bounds(b, mid, lo, hi);
        pack(lo, hi, wl);
        bounds(mid, e, lo, hi);
        pack(lo, hi, wr);
        const uint32_t rl = build(b, mid, depth + 1);
        const uint32_t rr = build(mid, e, depth + 1);
        pairs[me] = make_uint4(wl[0], wl[1], wl[2], wr[0]);
        pairs[me + 1] = make_uint4(wr[1], wr[2], rl, rr);
        return static_cast<uint32_t>(me / 2);
    }
};

// Builds the hierarchy over `leaves` into sibling-pair records in breadth-first order (the records
// every ray starts with sit next to each other).  False if it came out deeper than a traversal stack.
static bool buildOwnTree(const std::vector<TreeLeaf>& leaves, const QuantGrid& grid, std::vector<uint4>& pairs, uint32_t& rootRef,
    uint32_t& levels)
{
    pairs.clear();
    TreeBuild bld(leaves, pairs, grid);
    bld.start();
    pairs.reserve(leaves.size() * 2);
    rootRef = bld.build(0, static_cast<uint32_t>(leaves.size()), 1);
    if (bld.maxDepth + 1 > (uint32_t)kMaxStack) {
        pairs.clear();
        return false;
    }
    levels = bld.maxDepth; // nodes on the longest path from the root to a leaf, as for the caller's tree
    if ((rootRef & kLeafFlag) == 0) {
        const uint32_t nPairs = static_cast<uint32_t>(pairs.size() / 2);
        std::vector<uint32_t> order, newIndex(nPairs, 0);
        order.reserve(nPairs);
        order.push_back(rootRef);
        for (size_t h = 0; h < order.size(); h++) {
            const uint4 w1 = pairs[2 * static_cast<size_t>(order[h]) + 1];
            if ((w1.z & kLeafFlag) == 0)
                order.push_back(w1.z);
            if ((w1.w & kLeafFlag) == 0)
                order.push_back(w1.w);
        }
        for (uint32_t k = 0; k < order.size(); k++)
            newIndex[order[k]] = k;
        std::vector<uint4> sorted(pairs.size());
        for (uint32_t k = 0; k < order.size(); k++) {
            uint4 w0 = pairs[2 * static_cast<size_t>(order[k])], w1 = pairs[2 * static_cast<size_t>(order[k]) + 1];
            if ((w1.z & kLeafFlag) == 0)
                w1.z = newIndex[w1.z];
            if ((w1.w & kLeafFlag) == 0)
                w1.w = newIndex[w1.w];
            sorted[2 * static_cast<size_t>(k)] = w0, sorted[2 * static_cast<size_t>(k) + 1] = w1;
        }
        pairs.swap(sorted);
        rootRef = 0;
    }
    return true;
}

struct HostBuilder {
    const rt_primitive* in;
    std::vector<BuildPrim> info;
    std::vector<uint32_t> order; // current arrangement of primitive indices
    std::vector<KeyIdx> scratch;
    std::vector<rt_node> nodes;
    std::vector<uint32_t> outOrder;
    bool sahBinning = false;
    std::vector<uint32_t> binOf, tmpOrder;
    std::vector<float> costs;

    static float area(const float* lo, const float* hi) // computeSurfaceArea, src/extra.cpp:475-483
    {
        const float x = hi[0] - lo[0], y = hi[1] - lo[1], z = hi[2] - lo[2];
        return 2 * (x * y + x * z + y * z);
    }

    // splitPrimitivesBySAHBin, src/extra.cpp:493-543, on primitive indices: a counting sort into
    // the reference's eleven buckets (stable, like its bucket-by-bucket copy back), its cost of
    // every split position -- the right-hand box one primitive behind its count, as there -- and
    // the first cheapest position.  No extent along the axis (a division by zero there): bucket 0.
    uint32_t sahSplit(uint32_t begin, uint32_t n, int axis, const rt_node& node)
    {
        const size_t numBins = 10;
        const float binSize = (node.upper[axis] - node.lower[axis]) / numBins;
        uint32_t count[12] = {};
        binOf.resize(n);
        for (uint32_t i = 0; i < n; i++) {
            const float q = std::floor((info[order[begin + i]].c[axis] - node.lower[axis]) / binSize);
            uint32_t b = 0;
            if (q >= 0.0f)
                b = q >= float(numBins) ? static_cast<uint32_t>(numBins) : static_cast<uint32_t>(q);
            binOf[i] = b;
            count[b + 1]++;
        }
        for (int b = 0; b < 11; b++)
            count[b + 1] += count[b];
        tmpOrder.resize(n);
        for (uint32_t i = 0; i < n; i++)
            tmpOrder[count[binOf[i]]++] = order[begin + i];
        std::copy(tmpOrder.begin(), tmpOrder.end(), order.begin() + begin);

        costs.assign(n - 2, 0.0f);
        float lo[3], hi[3];
        auto start = [&](uint32_t k) {
            const BuildPrim& p = info[order[begin + k]];
            for (int a = 0; a < 3; a++)
                lo[a] = p.lo[a], hi[a] = p.hi[a];
        };
        auto merge = [&](uint32_t k) {
            const BuildPrim& p = info[order[begin + k]];
            for (int a = 0; a < 3; a++)
                lo[a] = std::min(lo[a], p.lo[a]), hi[a] = std::max(hi[a], p.hi[a]);
        };
        start(0);
        for (uint32_t i = 1; i < n - 1; i++) {
            costs[i - 1] = (float)i * area(lo, hi);
            merge(i);
        }
        start(n - 1);
        for (uint32_t i = n - 2; i > 0; i--) {
            costs[i - 1] += (float)(n - i) * area(lo, hi);
            merge(i);
        }
        uint32_t best = 0;
        for (uint32_t i = 0; i < n - 2; i++)
            if (costs[i] < costs[best])
                best = i;
        return best + 1;
    }

    void build(uint32_t begin, uint32_t n, uint32_t nodeIndex)
    {
        rt_node node;
        if (n == 0) { // computeSpanAABB of an empty span: AxisAlignedBox{} (src/bvh.cpp:142-144)
            for (int a = 0; a < 3; a++)
                node.lower[a] = 0.0f, node.upper[a] = 1.0f;
        } else {
            const BuildPrim& f = info[order[begin]];
            for (int a = 0; a < 3; a++)
                node.lower[a] = f.lo[a], node.upper[a] = f.hi[a];
            for (uint32_t i = 1; i < n; i++) {
                const BuildPrim& b = info[order[begin + i]];
                for (int a = 0; a < 3; a++) {
                    node.lower[a] = std::min(node.lower[a], b.lo[a]);
                    node.upper[a] = std::max(node.upper[a], b.hi[a]);
                }
            }
        }
        if (n <= 4) {
            node.data[0] = static_cast<uint32_t>(outOrder.size()) | RT_LEAF_BIT;
            node.data[1] = n;
            for (uint32_t i = 0; i < n; i++)
                outOrder.push_back(order[begin + i]);
            nodes[nodeIndex] = node;
            return;
        }
        const float xl = node.upper[0] - node.lower[0], yl = node.upper[1] - node.lower[1], zl = node.upper[2] - node.lower[2];
        const int axis = (xl >= yl && xl >= zl) ? 0 : (yl >= zl ? 1 : 2); // src/bvh.cpp:183-196
        uint32_t split;
        if (sahBinning) {
            split = sahSplit(begin, n, axis, node);
        } else {
            scratch.resize(n);
            for (uint32_t i = 0; i < n; i++)
                scratch[i] = { info[order[begin + i]].c[axis], order[begin + i] };
            std::sort(scratch.begin(), scratch.end(), [](const KeyIdx& a, const KeyIdx& b) { return a.key < b.key; });
            for (uint32_t i = 0; i < n; i++)
                order[begin + i] = scratch[i].idx;
            split = (n + 1) / 2;
        }
        const uint32_t left = static_cast<uint32_t>(nodes.size());
        nodes.emplace_back();
        const uint32_t right = static_cast<uint32_t>(nodes.size());
        nodes.emplace_back();
        node.data[0] = left;
        node.data[1] = right;
        nodes[nodeIndex] = node;
        build(begin, split, left);
        build(begin + split, n - split, right);
    }
};
} // namespace

int rt_bvh_build(const rt_primitive* prims_in, uint32_t n, int sah_binning, rt_node* nodes_out, rt_primitive* prims_out,
    uint32_t* n_nodes, uint32_t* n_levels, uint32_t* n_leaves)
try {
    if ((n && !prims_in) || !nodes_out || (n && !prims_out) || !n_nodes || !n_levels || !n_leaves)
        return fail(nullptr, RT_ERR_INVALID, "rt_bvh_build: NULL argument");
    HostBuilder b;
    b.in = prims_in;
    b.sahBinning = sah_binning != 0;
    b.info.resize(n);
    b.order.resize(n);
    for (uint32_t i = 0; i < n; i++) {
        const rt_primitive& p = prims_in[i];
        BuildPrim& q = b.info[i];
        for (int a = 0; a < 3; a++) {
            const float x = p.v0.position[a], y = p.v1.position[a], z = p.v2.position[a];
            q.lo[a] = std::min(x, std::min(y, z)); // src/bvh.cpp:117-133
            q.hi[a] = std::max(x, std::max(y, z));
            q.c[a] = ((x + y) + z) / 3.0f; // src/bvh.cpp:169-175
        }
        b.order[i] = i;
    }
    b.nodes.reserve(static_cast<size_t>(n) + 2);
    b.nodes.emplace_back();
    b.nodes.emplace_back(); // dummy node 1 keeps sibling pairs adjacent (src/bvh.cpp:88-89)
    std::memset(&b.nodes[1], 0, sizeof(rt_node));
    b.nodes[1].upper[0] = b.nodes[1].upper[1] = b.nodes[1].upper[2] = 1.0f;
    b.outOrder.reserve(n);
    b.build(0, n, 0);
    std::copy(b.nodes.begin(), b.nodes.end(), nodes_out);
    for (uint32_t i = 0; i < n; i++)
        prims_out[i] = prims_in[b.outOrder[i]];
    *n_nodes = static_cast<uint32_t>(b.nodes.size());
    // numLevels / numLeaves, src/bvh.cpp:498-547
    std::vector<uint32_t> levels(b.nodes.size(), 0);
    uint32_t numLevels = 1, numLeaves = 0;
    for (size_t i = 0; i < b.nodes.size(); i++) {
        if (i == 1)
            continue;
        const rt_node& nd = b.nodes[i];
        if (nd.data[0] & RT_LEAF_BIT) {
            numLeaves++;
        } else {
            levels[nd.data[0]] = levels[i] + 1;
            levels[nd.data[1]] = levels[i] + 1;
        }
    }
    for (size_t i = 2; i < b.nodes.size(); i++)
        numLevels = std::max(numLevels, levels[i] + 1);
    *n_levels = numLevels;
    *n_leaves = numLeaves;
    return RT_OK;
} catch (...) {
    return caught(nullptr, "rt_bvh_build");
}

int rt_bvh_build_device(rt_ctx* ctx, const rt_primitive* prims_in, uint32_t n, int sah_binning, rt_node* nodes_out,
    rt_primitive* prims_out, uint32_t* n_nodes, uint32_t* n_levels, uint32_t* n_leaves, uint32_t* host_sorted)
try {
    if (!ctx)
        return fail(nullptr, RT_ERR_INVALID, "rt_bvh_build_device: ctx is NULL");
    if ((n && !prims_in) || !nodes_out || (n && !prims_out) || !n_nodes || !n_levels || !n_leaves)
        return fail(ctx, RT_ERR_INVALID, "rt_bvh_build_device: NULL argument");
    if (host_sorted)
        *host_sorted = 0;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    DeviceBuildStats st;
    std::string err;
    const bool ok = sah_binning ? deviceSahBinBuild(prims_in, n, nodes_out, prims_out, n_nodes, n_levels, n_leaves, &st, err)
                                : deviceMedianBuild(prims_in, n, nodes_out, prims_out, n_nodes, n_levels, n_leaves, &st, err);
    if (!ok)
        return fail(ctx, RT_ERR_DEVICE, "rt_bvh_build_device: %s", err.c_str());
    if (host_sorted)
        *host_sorted = st.primitivesSortedOnHost;
    return RT_OK;
} catch (...) {
    return caught(ctx, "rt_bvh_build_device");
}

// ---------------------------------------------------------------------------
// Scene upload: validate the wire-format BVH, then flatten it.
// ---------------------------------------------------------------------------
int rt_scene_upload(rt_ctx* ctx, const rt_node* nodes, uint32_t n_nodes, const rt_primitive* prims, uint32_t n_prims,
    const rt_material* mesh_materials, uint32_t n_meshes, const rt_image* images, uint32_t n_images,
    const int32_t env_faces[6], const rt_sphere* spheres, uint32_t n_spheres, uint64_t* generation)
try {
    if (!ctx)
        return fail(nullptr, RT_ERR_INVALID, "rt_scene_upload: ctx is NULL");
    if (!nodes || n_nodes == 0)
        return fail(ctx, RT_ERR_INVALID, "rt_scene_upload: a BVH needs at least its root node");
    if ((n_prims && !prims) || (n_meshes && !mesh_materials) || (n_images && !images) || (n_spheres && !spheres))
        return fail(ctx, RT_ERR_INVALID, "rt_scene_upload: NULL array with non-zero count");
    if (n_prims > kLeafOffsetMask)
        return fail(ctx, RT_ERR_UNSUPPORTED, "rt_scene_upload: %u primitives exceed the %u supported", n_prims, kLeafOffsetMask);
    for (uint32_t i = 0; i < n_prims; i++)
        if (prims[i].meshID >= n_meshes)
            return fail(ctx, RT_ERR_INVALID, "rt_scene_upload: primitive %u names mesh %u of %u", i, prims[i].meshID, n_meshes);
    for (uint32_t i = 0; i < n_meshes; i++)
        if (mesh_materials[i].kdTexture >= (int32_t)n_images)
            return fail(ctx, RT_ERR_INVALID, "rt_scene_upload: material %u names image %d of %u", i, mesh_materials[i].kdTexture, n_images);
    for (uint32_t i = 0; i < n_images; i++)
        if (!images[i].pixels || images[i].width <= 0 || images[i].height <= 0)
            return fail(ctx, RT_ERR_INVALID, "rt_scene_upload: image %u is empty", i);
    for (int i = 0; i < 6; i++)
        if (env_faces && env_faces[i] >= (int32_t)n_images)
            return fail(ctx, RT_ERR_INVALID, "rt_scene_upload: environment face %d names image %d of %u", i, env_faces[i], n_images);

    // Walk the tree from the root: every child index in range, every node
    // reached once, leaves inside the primitive array and small enough for the
    // packed child reference.
    std::vector<int32_t> pairIndex(n_nodes, -1);
    std::vector<uint8_t> seen(n_nodes, 0);
    std::vector<uint32_t> innerOrder; // inner nodes in depth-first pre-order
    struct Item {
        uint32_t node, level;
    };
    std::vector<Item> work { { 0u, 1u } };
    uint32_t maxLevel = 1;
    while (!work.empty()) {
        const Item it = work.back();
        work.pop_back();
        if (it.node >= n_nodes)
            return fail(ctx, RT_ERR_INVALID, "rt_scene_upload: child index %u out of %u nodes", it.node, n_nodes);
        if (seen[it.node])
            return fail(ctx, RT_ERR_INVALID, "rt_scene_upload: node %u is reachable twice (not a tree)", it.node);
        seen[it.node] = 1;
        maxLevel = std::max(maxLevel, it.level);
        const rt_node& nd = nodes[it.node];
        if (nd.data[0] & RT_LEAF_BIT) {
            const uint32_t off = nd.data[0] & ~RT_LEAF_BIT, cnt = nd.data[1];
            if (cnt > (uint32_t)kMaxLeafCount)
                return fail(ctx, RT_ERR_UNSUPPORTED, "rt_scene_upload: leaf %u holds %u primitives (max %d)", it.node, cnt, kMaxLeafCount);
            if ((uint64_t)off + cnt > n_prims)
                return fail(ctx, RT_ERR_INVALID, "rt_scene_upload: leaf %u spans [%u, %u) of %u primitives", it.node, off, off + cnt, n_prims);
        } else {
            pairIndex[it.node] = static_cast<int32_t>(innerOrder.size());
            innerOrder.push_back(it.node);
            work.push_back({ nd.data[1], it.level + 1 });
            work.push_back({ nd.data[0], it.level + 1 });
        }
    }
    const uint32_t callerLevels = maxLevel; // only a limit when the caller's own inner nodes are traversed (below)

    auto childRef = [&](uint32_t child) -> uint32_t {
        const rt_node& c = nodes[child];
        if (c.data[0] & RT_LEAF_BIT)
            return kLeafFlag | (c.data[1] << 27) | (c.data[0] & ~RT_LEAF_BIT);
        return static_cast<uint32_t>(pairIndex[child]);
    };

    const rt_node& rootNode = nodes[0];
    QuantGrid grid;
    for (int a = 0; a < 3; a++) {
        grid.gridMin[a] = rootNode.lower[a];
        const double ext = static_cast<double>(rootNode.upper[a]) - static_cast<double>(rootNode.lower[a]);
        grid.gridCell[a] = (ext > 0.0 && std::isfinite(ext)) ? static_cast<float>(ext / 65535.0) : 0.0f;
    }
    const float* gridMin = grid.gridMin;
    const float* gridCell = grid.gridCell;
    auto packBox = [&](const float* lo, const float* hi, uint32_t* w) { grid.packBox(lo, hi, w); };
    std::vector<uint4> pairs;
    uint32_t rootRef = 0;
    bool ownTree = false;
    if (ctx->treeMode != 0 && !innerOrder.empty()) {
        // Traversal tree of the library's own, chosen by the surface-area heuristic: over the
        // caller's primitives, every one a leaf under its own tight box (mode 2), or over the
        // caller's leaves with their primitive ranges and boxes as they are (mode 1).  Which
        // primitive a ray hits and its t do not depend on the hierarchy; only the number of boxes
        // and primitives a ray has to test does.  (Of two primitives hit at exactly the same t
        // the reference keeps the one it tests later; a ray that meets such a pair is traced once
        // more in the reference's own order over the caller's nodes, Tracer::settleTie.)  The median
        // split of src/bvh.cpp:208-223 leaves heavily overlapping boxes, and a leaf of four
        // primitives makes a ray that grazes its box test all four.
        std::vector<TreeLeaf> leaves;
        for (uint32_t i = 0; i < n_nodes; i++) {
            if (!seen[i] || !(nodes[i].data[0] & RT_LEAF_BIT))
                continue;
            TreeLeaf l;
            for (int a = 0; a < 3; a++) {
                l.lo[a] = nodes[i].lower[a], l.hi[a] = nodes[i].upper[a];
                l.c[a] = 0.5f * (l.lo[a] + l.hi[a]);
            }
            l.count = nodes[i].data[1];
            l.ref = kLeafFlag | (nodes[i].data[1] << 27) | (nodes[i].data[0] & ~RT_LEAF_BIT);
            if (ctx->treeMode == 2) { // every primitive a leaf of its own, under its own box
                const uint32_t first = nodes[i].data[0] & ~RT_LEAF_BIT;
                for (uint32_t k = 0; k < nodes[i].data[1]; k++) {
                    const rt_primitive& q = prims[first + k];
                    TreeLeaf t;
                    for (int a = 0; a < 3; a++) {
                        t.lo[a] = std::min(q.v0.position[a], std::min(q.v1.position[a], q.v2.position[a]));
                        t.hi[a] = std::max(q.v0.position[a], std::max(q.v1.position[a], q.v2.position[a]));
                        t.c[a] = 0.5f * (t.lo[a] + t.hi[a]);
                    }
                    t.count = 1;
                    t.ref = kLeafFlag | (1u << 27) | (first + k);
                    leaves.push_back(t);
                }
                continue;
            }
            leaves.push_back(l);
        }
        if (ctx->treeBuildOnDevice) {
            HIP_TRY(ctx, hipSetDevice(ctx->device));
            std::string err;
            const int built = deviceOwnTree(leaves, grid, pairs, rootRef, maxLevel, err);
            if (built == kTreeFailed)
                return fail(ctx, RT_ERR_DEVICE, "rt_scene_upload: %s", err.c_str());
            // a shape the traversal cannot take: the host's builder has its say, then the caller's own nodes,
            // exactly as without the option (results do not depend on which tree is traversed)
            ownTree = built == kTreeBuilt || buildOwnTree(leaves, grid, pairs, rootRef, maxLevel);
        } else {
            ownTree = buildOwnTree(leaves, grid, pairs, rootRef, maxLevel);
        }
    }
    if (!ownTree) {
        if (callerLevels > (uint32_t)kMaxStack)
            return fail(ctx, RT_ERR_UNSUPPORTED, "rt_scene_upload: BVH has %u levels (max %d)", callerLevels, kMaxStack);
        maxLevel = callerLevels;
        pairs.resize(innerOrder.size() * 2);
        for (size_t k = 0; k < innerOrder.size(); k++) {
            const rt_node& nd = nodes[innerOrder[k]];
            uint32_t wl[3], wr[3];
            packBox(nodes[nd.data[0]].lower, nodes[nd.data[0]].upper, wl);
            packBox(nodes[nd.data[1]].lower, nodes[nd.data[1]].upper, wr);
            pairs[2 * k] = make_uint4(wl[0], wl[1], wl[2], wr[0]);
            pairs[2 * k + 1] = make_uint4(wr[1], wr[2], childRef(nd.data[0]), childRef(nd.data[1]));
        }
        const rt_node& r0 = nodes[0];
        rootRef = (r0.data[0] & RT_LEAF_BIT) ? (kLeafFlag | (r0.data[1] << 27) | (r0.data[0] & ~RT_LEAF_BIT)) : 0u;
    }

    // per primitive, the box of the caller's leaf that holds it (Tracer::surelyReached): lower.xyz upper.x | upper.yz 0 0
    std::vector<float4> leafBoxes(static_cast<size_t>(n_prims) * 2, make_float4(0.0f, 0.0f, 0.0f, 0.0f));
    for (uint32_t i = 0; i < n_nodes; i++) {
        if (!seen[i] || !(nodes[i].data[0] & RT_LEAF_BIT))
            continue;
        const uint32_t first = nodes[i].data[0] & ~RT_LEAF_BIT;
        for (uint32_t k = 0; k < nodes[i].data[1]; k++) {
            leafBoxes[2 * static_cast<size_t>(first + k)] = make_float4(nodes[i].lower[0], nodes[i].lower[1], nodes[i].lower[2], nodes[i].upper[0]);
            leafBoxes[2 * static_cast<size_t>(first + k) + 1] = make_float4(nodes[i].upper[1], nodes[i].upper[2], 0.0f, 0.0f);
        }
    }

    std::vector<float4> tris(static_cast<size_t>(n_prims) * 6), attrs(static_cast<size_t>(n_prims) * 4);
    for (uint32_t i = 0; i < n_prims; i++) {
        const rt_primitive& q = prims[i];
        const f3 v0 = { q.v0.position[0], q.v0.position[1], q.v0.position[2] };
        const f3 v1 = { q.v1.position[0], q.v1.position[1], q.v1.position[2] };
        const f3 v2 = { q.v2.position[0], q.v2.position[1], q.v2.position[2] };
        const f3 c = cross(v1 - v0, v2 - v0);
        const f3 n = c * (1.0f / sqrtf(dot(c, c)));
        const float D = dot(v0, n);
        const f3 E0 = cross(v2 - v0, n), E1 = cross(v0 - v1, n), E2 = cross(v1 - v2, n);
        float4* t = &tris[6 * static_cast<size_t>(i)];
        t[0] = make_float4(n.x, n.y, n.z, D);
        t[1] = make_float4(E0.x, E0.y, E0.z, v0.x);
        t[2] = make_float4(v0.y, v0.z, E1.x, E1.y);
        t[3] = make_float4(E1.z, v1.x, v1.y, v1.z);
        t[4] = make_float4(E2.x, E2.y, E2.z, v2.x);
        t[5] = make_float4(v2.y, v2.z, 0.0f, 0.0f);
        float4* a = &attrs[4 * static_cast<size_t>(i)];
        float meshBits;
        std::memcpy(&meshBits, &q.meshID, 4);
        a[0] = make_float4(q.v0.normal[0], q.v0.normal[1], q.v0.normal[2], q.v1.normal[0]);
        a[1] = make_float4(q.v1.normal[1], q.v1.normal[2], q.v2.normal[0], q.v2.normal[1]);
        a[2] = make_float4(q.v2.normal[2], q.v0.texCoord[0], q.v0.texCoord[1], q.v1.texCoord[0]);
        a[3] = make_float4(q.v1.texCoord[1], q.v2.texCoord[0], q.v2.texCoord[1], meshBits);
    }

    std::vector<DevMaterial> mats(n_meshes + n_spheres);
    auto toDev = [](const rt_material& m) {
        DevMaterial d {};
        std::memcpy(d.kd, m.kd, 12);
        std::memcpy(d.ks, m.ks, 12);
        d.shininess = m.shininess;
        d.transparency = m.transparency;
        d.tex = m.kdTexture;
        return d;
    };
    for (uint32_t i = 0; i < n_meshes; i++)
        mats[i] = toDev(mesh_materials[i]);
    std::vector<DevSphere> sph(n_spheres);
    for (uint32_t i = 0; i < n_spheres; i++) {
        if (spheres[i].material.kdTexture >= (int32_t)n_images)
            return fail(ctx, RT_ERR_INVALID, "rt_scene_upload: sphere %u names image %d of %u", i, spheres[i].material.kdTexture, n_images);
        mats[n_meshes + i] = toDev(spheres[i].material);
        std::memcpy(sph[i].center, spheres[i].center, 12);
        sph[i].radius = spheres[i].radius;
        sph[i].material = static_cast<int32_t>(n_meshes + i);
    }
    std::vector<DevTex> texs(n_images);
    size_t texFloats = 0;
    for (uint32_t i = 0; i < n_images; i++) {
        texs[i].offset = static_cast<uint32_t>(texFloats);
        texs[i].width = images[i].width;
        texs[i].height = images[i].height;
        texFloats += static_cast<size_t>(images[i].width) * images[i].height * 3;
        if (texFloats > 0xffffffffull)
            return fail(ctx, RT_ERR_UNSUPPORTED, "rt_scene_upload: textures exceed 2^32 floats");
    }
    std::vector<float> texels(texFloats);
    for (uint32_t i = 0; i < n_images; i++)
        std::memcpy(texels.data() + texs[i].offset, images[i].pixels, static_cast<size_t>(images[i].width) * images[i].height * 12);

    ctx->anyMirror = ctx->anyClear = ctx->anyMirrorAndClear = false;
    for (const DevMaterial& m : mats) {
        const bool mirror = !(m.ks[0] == 0.0f && m.ks[1] == 0.0f && m.ks[2] == 0.0f), clear = m.transparency != 1.0f;
        ctx->anyMirror |= mirror;
        ctx->anyClear |= clear;
        ctx->anyMirrorAndClear |= mirror && clear;
    }

    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, hipDeviceSynchronize());
    // On the device an inner child reference is the BYTE offset of the child's record (index * 32), so that a
    // traversal step forms no address: the loads take it as their 32-bit offset from the array's base.
    if (pairs.size() / 2 > (size_t(1) << 26))
        return fail(ctx, RT_ERR_UNSUPPORTED, "rt_scene_upload: %zu sibling-pair records exceed the %u supported", pairs.size() / 2, 1u << 26);
    for (size_t k = 1; k < pairs.size(); k += 2) {
        if ((pairs[k].z & kLeafFlag) == 0)
            pairs[k].z *= kPairBytes;
        if ((pairs[k].w & kLeafFlag) == 0)
            pairs[k].w *= kPairBytes;
    }
    if ((rootRef & kLeafFlag) == 0)
        rootRef *= kPairBytes;
    HIP_TRY(ctx, uploadVec(&ctx->dPairs, pairs));
    ctx->numPairs = static_cast<uint32_t>(pairs.size() / 2);
    // The caller's nodes, verbatim: a ray that meets two primitives at exactly the same distance is
    // replayed over them in the reference's order of tests, which decides the one it keeps.  The
    // replay's stack holds at most one entry per level of the caller's tree.
    static_assert(sizeof(rt_node) == 2 * sizeof(uint4), "rt_node is two 16-byte words");
    const bool replayable = callerLevels <= static_cast<uint32_t>(kMaxStack);
    {
        std::vector<uint4> ref(replayable ? static_cast<size_t>(n_nodes) * 2 : 0);
        if (replayable)
            std::memcpy(ref.data(), nodes, static_cast<size_t>(n_nodes) * sizeof(rt_node));
        HIP_TRY(ctx, uploadVec(&ctx->dRefNodes, ref));
    }
    if (replayable)
        maxLevel = std::max(maxLevel, callerLevels);
    HIP_TRY(ctx, uploadVec(&ctx->dLeafBoxes, leafBoxes));
    HIP_TRY(ctx, uploadVec(&ctx->dTris, tris));
    HIP_TRY(ctx, uploadVec(&ctx->dAttrs, attrs));
    HIP_TRY(ctx, uploadVec(&ctx->dMats, mats));
    HIP_TRY(ctx, uploadVec(&ctx->dTexs, texs));
    HIP_TRY(ctx, uploadVec(&ctx->dTexels, texels));
    HIP_TRY(ctx, uploadVec(&ctx->dSpheres, sph));

    DevScene& s = ctx->scene;
    const int32_t keepLights = s.nLights;
    const rt_light* keepLightPtr = s.lights;
    s = DevScene {};
    s.pairs = ctx->dPairs;
    s.tris = ctx->dTris;
    s.attrs = ctx->dAttrs;
    s.mats = ctx->dMats;
    s.texs = ctx->dTexs;
    s.texels = ctx->dTexels;
    s.spheres = ctx->dSpheres;
    s.lights = keepLightPtr;
    s.nLights = keepLights;
    const rt_node& root = nodes[0];
    s.rootRef = rootRef;
    std::memcpy(s.rootLo, root.lower, 12);
    std::memcpy(s.rootHi, root.upper, 12);
    std::memcpy(s.gridMin, gridMin, 12);
    std::memcpy(s.gridCell, gridCell, 12);
    s.nPrims = n_prims;
    s.tieByIndex = (ownTree && ctx->treeMode == 2) ? 1 : 0;
    s.refNodes = replayable ? ctx->dRefNodes : nullptr;
    s.leafBoxes = ctx->referenceBoxes ? ctx->dLeafBoxes : nullptr;
    s.strictIntersect = ctx->strictIntersect ? 1 : 0;
    s.nMeshes = static_cast<int32_t>(n_meshes);
    s.nSpheres = static_cast<int32_t>(n_spheres);
    s.nTex = static_cast<int32_t>(n_images);
    for (int i = 0; i < 6; i++)
        s.env[i] = env_faces ? env_faces[i] : -1;
    ctx->numLevels = maxLevel;
    ctx->haveScene = true;
    ctx->generation++;
    if (generation)
        *generation = ctx->generation;
    return warmOccupancy(ctx);
} catch (...) {
    return caught(ctx, "rt_scene_upload");
}

int rt_lights_set(rt_ctx* ctx, const rt_light* lights, uint32_t n)
try {
    if (!ctx)
        return fail(nullptr, RT_ERR_INVALID, "rt_lights_set: ctx is NULL");
    if (n && !lights)
        return fail(ctx, RT_ERR_INVALID, "rt_lights_set: NULL array with non-zero count");
    for (uint32_t i = 0; i < n; i++)
        if (lights[i].type < 0 || lights[i].type > 2)
            return fail(ctx, RT_ERR_INVALID, "rt_lights_set: light %u has type %d", i, lights[i].type);
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    if (!ctx->copyStream) {
        HIP_TRY(ctx, hipStreamCreateWithFlags(&ctx->copyStream, hipStreamNonBlocking));
        for (hipEvent_t& e : ctx->lightSlotUsed)
            HIP_TRY(ctx, hipEventCreateWithFlags(&e, hipEventDisableTiming));
    }
    if (n > ctx->lightCap || !ctx->dLights) {
        HIP_TRY(ctx, hipDeviceSynchronize());
        if (ctx->dLights)
            (void)hipFree(ctx->dLights);
        ctx->dLights = nullptr;
        ctx->lightCap = std::max<uint32_t>(n, 16);
        HIP_TRY(ctx, hipMalloc(reinterpret_cast<void**>(&ctx->dLights), static_cast<size_t>(rt_ctx::kLightSlots) * ctx->lightCap * sizeof(rt_light)));
        for (bool& pending : ctx->lightSlotPending)
            pending = false;
    }
    // The next slot, written now (frames already enqueued keep the slot they were enqueued with; see rt_ctx).
    const int slot = (ctx->lightSlot + 1) % rt_ctx::kLightSlots;
    if (ctx->lightSlotPending[slot]) {
        HIP_TRY(ctx, hipEventSynchronize(ctx->lightSlotUsed[slot]));
        ctx->lightSlotPending[slot] = false;
    }
    rt_light* dst = ctx->dLights + static_cast<size_t>(slot) * ctx->lightCap;
    if (n) {
        HIP_TRY(ctx, hipMemcpyAsync(dst, lights, n * sizeof(rt_light), hipMemcpyHostToDevice, ctx->copyStream));
        HIP_TRY(ctx, hipStreamSynchronize(ctx->copyStream));
    }
    ctx->lightSlot = slot;
    ctx->hostLights.assign(lights, lights + n);
    ctx->scene.lights = dst;
    ctx->scene.nLights = static_cast<int32_t>(n);
    return RT_OK;
} catch (...) {
    return caught(ctx, "rt_lights_set");
}

int rt_features_set(rt_ctx* ctx, const rt_features* f)
try {
    if (!ctx || !f)
        return fail(ctx, RT_ERR_INVALID, "rt_features_set: NULL argument");
    if (f->enableMotionBlur && !f->enableDepthOfField && f->motionBlurSamples > 65535)
        return fail(ctx, RT_ERR_UNSUPPORTED, "rt_features_set: %d motion blur samples", f->motionBlurSamples);
    if (f->enableBloomEffect && (f->bloomFilterSize < (uint32_t)kBloomMinSize || f->bloomFilterSize > (uint32_t)kBloomMaxSize))
        return fail(ctx, RT_ERR_UNSUPPORTED,
            "rt_features_set: bloomFilterSize %u: below %d the reference's filter loop writes outside the image, above %d its factorial overflows",
            f->bloomFilterSize, kBloomMinSize, kBloomMaxSize);
    if (f->shadingModel < 0 || f->shadingModel > 3)
        return fail(ctx, RT_ERR_INVALID, "rt_features_set: shadingModel %d", f->shadingModel);
    if (f->maxRayDepth < 0 || f->maxRayDepth > kMaxDepth)
        return fail(ctx, RT_ERR_INVALID, "rt_features_set: maxRayDepth %d not in [0, %d]", f->maxRayDepth, kMaxDepth);
    ctx->features = *f;
    ctx->haveFeatures = true;
    return warmOccupancy(ctx);
} catch (...) {
    return caught(ctx, "rt_features_set");
}

} // extern "C"

// state after k draws of the LCG (src/sampler.h:18) = mul * seed + add
static void lcgJump(uint64_t k, uint32_t& mul, uint32_t& add)
{
    mul = 1u;
    add = 0u;
    uint32_t stepMul = 747796405u, stepAdd = 2891336453u; // one draw; squared each round
    while (k) {
        if (k & 1) {
            mul = mul * stepMul;
            add = add * stepMul + stepAdd;
        }
        stepAdd = stepAdd * stepMul + stepAdd;
        stepMul = stepMul * stepMul;
        k >>= 1;
    }
}

// renderImage takes the depth-of-field renderer when both are asked for (src/render.cpp:22-25)
static bool motionBlurOn(const rt_features& f) { return f.enableMotionBlur && !f.enableDepthOfField; }

static DevFeat makeFeat(const rt_features& f)
{
    DevFeat d {};
    d.shading = f.enableShading, d.reflections = f.enableReflections, d.shadows = f.enableShadows;
    d.normalInterp = f.enableNormalInterp, d.texMapping = f.enableTextureMapping, d.accel = f.enableAccelStructure;
    d.bilinear = f.enableBilinearTextureFiltering, d.transparency = f.enableTransparency, d.jitter = f.enableJitteredSampling;
    d.shadingModel = f.shadingModel;
    d.spp = f.numPixelSamples, d.shadowSamples = f.numShadowSamples;
    d.dof = f.enableDepthOfField, d.envMap = f.enableEnvironmentMap, d.glossy = f.enableGlossyReflection;
    d.glossySamples = f.numGlossySamples, d.glossyExponent = f.glossyExponent;
    d.dofDistance = f.depthOfFieldDistance, d.dofDiameter = f.depthOfFieldCircleDiameter, d.dofSamples = f.numDepthOfFieldSamples;
    d.divDof = makeFastDiv(std::max(1u, f.numDepthOfFieldSamples));
    d.maxDepth = f.maxRayDepth;
    d.motion = motionBlurOn(f), d.motionSamples = f.motionBlurSamples, d.movement = f.movement;
    // src/render.cpp:58-70, :118: ray count and draws of the pixel-sample stretch
    d.jitterN = static_cast<uint32_t>(std::round(std::sqrt(static_cast<float>(f.numPixelSamples))));
    uint32_t drawsPerCamRay = 0;
    if (f.numPixelSamples > 1) {
        d.camRays = f.enableJitteredSampling ? d.jitterN * d.jitterN : f.numPixelSamples;
        drawsPerCamRay = 2;
    } else {
        d.camRays = 1;
    }
    d.raysPerPixel = f.enableDepthOfField ? d.camRays * f.numDepthOfFieldSamples : d.camRays;
    const uint64_t pixDraws = static_cast<uint64_t>(drawsPerCamRay) * d.camRays;
    const uint64_t lensDraws = f.enableDepthOfField ? 2ull * d.camRays * f.numDepthOfFieldSamples : 0ull;
    lcgJump(pixDraws, d.lensMul, d.lensAdd);
    lcgJump(pixDraws + lensDraws, d.mainMul, d.mainAdd);
    return d;
}

// Trackball accessors, framework/src/trackball.cpp:26-27, :71-74, :112-126,
// with glm::quat(eulerAngles) (glm detail/type_quat.inl:208-217).
static DevCamera makeCamera(const rt_camera& c)
{
    DevCamera d {};
    d.halfH = std::tan(c.fovy / 2.0f);
    d.halfW = c.aspect * d.halfH;
    const float cx = std::cos(c.rotation[0] * 0.5f), cy = std::cos(c.rotation[1] * 0.5f), cz = std::cos(c.rotation[2] * 0.5f);
    const float sx = std::sin(c.rotation[0] * 0.5f), sy = std::sin(c.rotation[1] * 0.5f), sz = std::sin(c.rotation[2] * 0.5f);
    d.qw = cx * cy * cz + sx * sy * sz;
    d.q.x = sx * cy * cz - cx * sy * sz;
    d.q.y = cx * sy * cz + sx * cy * sz;
    d.q.z = cx * cy * sz - sx * sy * cz;
    d.pos = mk3(c.lookAt[0], c.lookAt[1], c.lookAt[2]) + quatRotate(d.qw, d.q, mk3(0.0f, 0.0f, -c.distance));
    d.fwd = quatRotate(d.qw, d.q, mk3(0.0f, 0.0f, 1.0f));
    d.up = quatRotate(d.qw, d.q, mk3(0.0f, 1.0f, 0.0f));
    d.left = quatRotate(d.qw, d.q, mk3(1.0f, 0.0f, 0.0f));
    return d;
}

static size_t shardTiles(int width, int height, int shard, int nShard)
{
    const size_t tiles = static_cast<size_t>((width + kTile - 1) / kTile) * ((height + kTile - 1) / kTile);
    return (tiles > static_cast<size_t>(shard)) ? (tiles - shard + nShard - 1) / nShard : 0;
}

// A fixed piece of device memory a render call carves its working buffers from instead of
// allocating them (rt_render_device_noalloc: this device's copy of the static scratch array).
struct Arena {
    char* base;
    size_t bytes, used;
    void* take(size_t n)
    {
        const size_t at = (used + 255) & ~size_t(255);
        if (at + n > bytes)
            return nullptr;
        used = at + n;
        return base + at;
    }
};

static size_t traversalLds(const rt_ctx* ctx)
{
    // A ray's stack holds the bottom marker and at most one waiting sibling per inner node on its path:
    // a tree whose longest root-to-leaf path has L nodes needs L entries.
    return static_cast<size_t>(std::max<uint32_t>(ctx->numLevels, 2)) * kBlockThreads * 4;
}

static int perPixelVariant(const rt_features& F, bool collect)
{
    int variant = 0;
    if (motionBlurOn(F))
        return 16 | 1 | 2 | (collect ? 8 : 0);
    if (F.enableTransparency)
        variant |= 1;
    if (F.enableGlossyReflection && F.enableReflections)
        variant |= 2;
    if (F.enableDepthOfField)
        variant |= 4;
    if (collect)
        variant |= 8;
    return variant;
}

// Resident blocks per compute unit of the traversing kernels under the current scene and features
// (non-counting variants), asked of the runtime once per change of either: rt_scene_upload and
// rt_features_set call this, so a render call finds the answer ready.
static int warmOccupancy(rt_ctx* ctx)
{
    ctx->occ.valid = false;
    if (!ctx->haveScene || !ctx->haveFeatures)
        return RT_OK;
    const rt_features& F = ctx->features;
    rt_ctx::Occupancy o;
    o.lds = traversalLds(ctx);
    o.transp = F.enableTransparency != 0;
    o.variant = perPixelVariant(F, false);
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, wavefrontOccupancy(false, o.transp != 0, kBlockThreads, o.lds, &o.path, &o.shadow));
    HIP_TRY(ctx, renderOccupancy(o.variant, kBlockThreads, o.lds, &o.render));
    o.valid = true;
    ctx->occ = o;
    return RT_OK;
}

// postprocessImageWithBloom (src/extra.cpp:276-316) on a whole frame in Screen order, in place.
// `copy`: room for the thresholded copy of the frame, or null to use the context's own buffer.
static int bloomDevice(rt_ctx* ctx, float* dFrame, int width, int height, hipStream_t stream, float* copy = nullptr)
{
    const rt_features& F = ctx->features;
    if (!F.enableBloomEffect)
        return RT_OK;
    const size_t floats = static_cast<size_t>(width) * height * 3;
    if (copy) {
        BloomWeights w;
        bloomWeights(static_cast<int>(F.bloomFilterSize), w);
        HIP_TRY(ctx, launchBloom(stream, dFrame, copy, width, height, static_cast<int>(F.bloomFilterSize), F.bloomThreshold, w));
        return RT_OK;
    }
    if (floats > ctx->bloomCap) {
        HIP_TRY(ctx, hipStreamSynchronize(stream));
        if (ctx->dBloom)
            (void)hipFree(ctx->dBloom);
        ctx->dBloom = nullptr;
        ctx->bloomCap = 0;
        HIP_TRY(ctx, hipMalloc(reinterpret_cast<void**>(&ctx->dBloom), floats * sizeof(float)));
        ctx->bloomCap = floats;
    }
    BloomWeights w;
    bloomWeights(static_cast<int>(F.bloomFilterSize), w);
    HIP_TRY(ctx, launchBloom(stream, dFrame, ctx->dBloom, width, height, static_cast<int>(F.bloomFilterSize), F.bloomThreshold, w));
    return RT_OK;
}

static int traceKnobs(int minActive, int a, int b)
{
    auto c = [](int v) { return std::max(0, std::min(v, 255)); };
    return c(minActive) | (c(a) << 8) | (c(b) << 16);
}

// For every 8x8 tile of this shard, how many of its pixels can have a camera ray that enters the scene's box
// (the root box test of BVHInterface::intersect, src/bvh.cpp:253-259): the pixels inside the bounding
// rectangle of the box's eight corners as the camera sees them (Trackball::generateRay,
// framework/src/trackball.cpp:101-110), a pixel wider on every side.  Only such pixels can make hits, so
// only they need room in the pools of hits and light samples.  Every real pixel counts when the bound
// does not apply: a corner beside or behind the camera, rays that start on a lens instead of the camera's
// position (depth of field), spheres in the scene or the acceleration structure off (every ray is traced).
static void candidatePixelsPerTile(const rt_ctx* ctx, const rt_camera& camera, const DevFrame& fr, int width, int height,
    std::vector<uint16_t>& out, std::vector<uint32_t>& validBefore, int rect[4])
{
    const uint32_t nTiles = fr.nWork / 64u;
    out.assign(nTiles, 0);
    validBefore.assign(nTiles + 1, 0u);
    const rt_features& F = ctx->features;
    const DevScene& sc = ctx->scene;
    bool bounded = F.enableAccelStructure && !F.enableDepthOfField && sc.nSpheres == 0 && sc.nPrims != 0;
    int x0 = 0, x1 = width - 1, y0 = 0, y1 = height - 1; // inclusive pixel rectangle
    if (bounded) {
        const DevCamera cam = makeCamera(camera);
        const f3 qInv = { -cam.q.x, -cam.q.y, -cam.q.z };
        double lox = 1e300, hix = -1e300, loy = 1e300, hiy = -1e300;
        for (int k = 0; k < 8 && bounded; k++) {
            const f3 c = { (k & 1) ? sc.rootHi[0] : sc.rootLo[0], (k & 2) ? sc.rootHi[1] : sc.rootLo[1], (k & 4) ? sc.rootHi[2] : sc.rootLo[2] };
            const f3 v = quatRotate(cam.qw, qInv, c - cam.pos); // the corner in the camera's frame: rays leave along +z
            if (!(v.z > 1.0e-3f) || !std::isfinite(v.x) || !std::isfinite(v.y)) {
                bounded = false;
                break;
            }
            // direction (-x halfW, y halfH, 1) of normalised device coordinates (x, y); pixel = (ndc + 1) / 2 * size
            const double nx = -double(v.x) / (double(v.z) * cam.halfW), ny = double(v.y) / (double(v.z) * cam.halfH);
            const double px = (nx + 1.0) * 0.5 * width, py = (ny + 1.0) * 0.5 * height;
            lox = std::min(lox, px), hix = std::max(hix, px), loy = std::min(loy, py), hiy = std::max(hiy, py);
        }
        if (bounded) {
            // a pixel holds the samples [p, p + 1); one more pixel on every side for the roundings of either computation
            x0 = static_cast<int>(std::max(0.0, std::floor(lox) - 1.0)), x1 = static_cast<int>(std::min(double(width - 1), std::floor(hix) + 1.0));
            y0 = static_cast<int>(std::max(0.0, std::floor(loy) - 1.0)), y1 = static_cast<int>(std::min(double(height - 1), std::floor(hiy) + 1.0));
        }
    }
    rect[0] = x0, rect[1] = x1, rect[2] = y0, rect[3] = y1;
    // tile t of the shard is tile t * nShard + shard of the frame: column and row kept by stepping, not by division
    int col = fr.shard % fr.tilesX, row = fr.shard / fr.tilesX;
    const int colStep = fr.nShard % fr.tilesX, rowStep = fr.nShard / fr.tilesX;
    uint32_t valid = 0;
    for (uint32_t t = 0; t < nTiles; t++) {
        const int tx = col * kTile, ty = row * kTile;
        const int ax = std::max(tx, x0), bx = std::min(std::min(tx + kTile, width) - 1, x1);
        const int ay = std::max(ty, y0), by = std::min(std::min(ty + kTile, height) - 1, y1);
        out[t] = (bx >= ax && by >= ay) ? static_cast<uint16_t>((bx - ax + 1) * (by - ay + 1)) : 0;
        validBefore[t] = valid;
        valid += static_cast<uint32_t>(std::min(kTile, width - tx) * std::min(kTile, height - ty));
        col += colStep, row += rowStep;
        if (col >= fr.tilesX)
            col -= fr.tilesX, row++;
    }
    validBefore[nTiles] = valid;
}

// bytes of plain values, appended: the keys of rt_ctx::FramePlan
template <typename T>
static void keyAdd(std::vector<uint8_t>& key, const T& v)
{
    const uint8_t* p = reinterpret_cast<const uint8_t*>(&v);
    key.insert(key.end(), p, p + sizeof(T));
}

// `arena`: null = working buffers are the context's own (allocated and grown as needed); else they
// are carved from it and the call allocates nothing, times nothing (events are allocations too) and
// splits the frame into as many pixel batches as the arena's size asks for.
static int renderCommon(rt_ctx* ctx, const rt_camera* camera, int width, int height, int shard, int nShard, float* dOut,
    size_t dOutFloats, hipStream_t stream, bool collect, bool timeIt, Arena* arena = nullptr)
{
    if (!ctx->haveScene)
        return fail(ctx, RT_ERR_STATE, "render: no scene uploaded (rt_scene_upload)");
    if (!ctx->haveFeatures)
        return fail(ctx, RT_ERR_STATE, "render: no features set (rt_features_set)");
    if (!camera || width <= 0 || height <= 0 || nShard <= 0 || shard < 0 || shard >= nShard)
        return fail(ctx, RT_ERR_INVALID, "render: bad camera / size / shard (%dx%d, shard %d of %d)", width, height, shard, nShard);
    if (static_cast<uint64_t>(width) * height > 0x3fffffffull)
        return fail(ctx, RT_ERR_UNSUPPORTED, "render: %dx%d exceeds the supported pixel count", width, height);
    const size_t need = (nShard == 1) ? static_cast<size_t>(width) * height * 3 : shardTiles(width, height, shard, nShard) * 64 * 3;
    if (!dOut || dOutFloats < need)
        return fail(ctx, RT_ERR_INVALID, "render: output holds %zu floats, %zu needed", dOutFloats, need);

    const DevFeat ft = makeFeat(ctx->features);
    const DevCamera cam = makeCamera(*camera);
    DevFrame fr {};
    fr.width = width, fr.height = height;
    fr.tilesX = (width + kTile - 1) / kTile, fr.tilesY = (height + kTile - 1) / kTile;
    fr.divTilesX = makeFastDiv(static_cast<uint32_t>(fr.tilesX));
    fr.shard = shard, fr.nShard = nShard;
    fr.nWork = static_cast<uint32_t>(shardTiles(width, height, shard, nShard) * 64);
    fr.packed = (nShard == 1) ? 0 : 1;
    fr.candX0 = 0, fr.candX1 = width - 1, fr.candY0 = 0, fr.candY1 = height - 1;

    const int blockThreads = kBlockThreads;
    const int maxPerCU = 32 / kBlockWaves; // a compute unit holds 32 wavefronts
    const size_t lds = traversalLds(ctx);
    const rt_features& F = ctx->features;

    DevCounters* dCounters = ctx->dCounters;
    float* bloomCopy = nullptr;
    if (arena) {
        dCounters = static_cast<DevCounters*>(arena->take(sizeof(DevCounters)));
        if (F.enableBloomEffect && nShard == 1) {
            bloomCopy = static_cast<float*>(arena->take(static_cast<size_t>(width) * height * 12));
            if (!bloomCopy)
                return fail(ctx, RT_ERR_UNSUPPORTED, "render: the bloom copy of a %dx%d frame does not fit the %zu-byte scratch", width, height, arena->bytes);
        }
        if (!dCounters)
            return fail(ctx, RT_ERR_INTERNAL, "render: scratch smaller than the counters");
    }
    ctx->lastCounters = dCounters;
    HIP_TRY(ctx, hipMemsetAsync(dCounters, 0, collect ? sizeof(DevCounters) : offsetof(DevCounters, waveLog), stream));
    if (timeIt) {
        if (ctx->evUsed + 2 > ctx->evPool.size()) {
            if (ctx->evPool.size() >= 2 * kMaxTimedLaunches) { // unread launches pile up: wait once and fold them into the sums
                HIP_TRY(ctx, hipStreamSynchronize(stream));
                const int rf = foldTimings(ctx);
                if (rf != RT_OK)
                    return rf;
            } else {
                for (int i = 0; i < 2; i++) {
                    hipEvent_t a;
                    HIP_TRY(ctx, hipEventCreate(&a));
                    ctx->evPool.push_back(a);
                }
            }
        }
        HIP_TRY(ctx, hipEventRecord(ctx->evPool[ctx->evUsed], stream));
    }

    // the kernels' resident blocks per compute unit: as found when the scene / features were set, unless
    // this is a counting launch (other variants) or something changed behind that
    const bool occReady = !collect && ctx->occ.valid && ctx->occ.lds == lds && ctx->occ.transp == (F.enableTransparency != 0)
        && ctx->occ.variant == perPixelVariant(F, false);
    if (arena && !occReady)
        return fail(ctx, RT_ERR_STATE, "render: launch geometry not prepared (rt_scene_upload and rt_features_set do that)");
    // Glossy reflections take the queue-based renderer too (a tree kernel walks each pixel's radiance rays, everything
    // else is shared with the mirror case), except together with transparency, which only the per-pixel kernel combines.
    const bool glossyOn = F.enableGlossyReflection && F.enableReflections;
    const bool glossyWf = glossyOn && !ctx->forcePerPixel && !F.enableTransparency && !motionBlurOn(F);
    const bool perPixel = ctx->forcePerPixel || (glossyOn && !glossyWf) || motionBlurOn(F);
    if (perPixel) {
        const int variant = perPixelVariant(F, collect);
        int perCU = ctx->blocksPerCU;
        if (perCU <= 0) {
            if (occReady)
                perCU = ctx->occ.render;
            else
                HIP_TRY(ctx, renderOccupancy(variant, blockThreads, lds, &perCU));
            perCU = std::max(1, std::min(perCU, maxPerCU));
        }
        const uint32_t wavesNeeded = fr.nWork / 64;
        int blocks = std::min<int64_t>(static_cast<int64_t>(ctx->numCU) * perCU, (wavesNeeded + kBlockWaves - 1) / kBlockWaves);
        blocks = std::max(blocks, 1);
        HIP_TRY(ctx, launchRender(variant, blocks, blockThreads, lds, stream, ctx->scene, cam, ft, fr, dCounters, dOut,
                          traceKnobs(ctx->minActive, ctx->parkA, ctx->parkB), ctx->logicRounds));
    } else {
        // ---- wavefront path: work items are camera rays and light samples
        WfParams wp {};
        wp.R = ft.raysPerPixel;
        uint32_t slots = 0, drawsPerHit = 0;
        {
            for (const rt_light& l : ctx->hostLights) {
                slots += (l.type == 0) ? 1u : F.numShadowSamples;
                drawsPerHit += (l.type == 0) ? 0u : (l.type == 1 ? F.numShadowSamples : 2u * F.numShadowSamples);
            }
        }
        wp.slots = slots;
        wp.divSlots = makeFastDiv(std::max(1u, slots));
        wp.divR = makeFastDiv(wp.R);
        wp.divPerRound = makeFastDiv(std::min(wp.R, 256u));
        wp.drawsPerHit = drawsPerHit;
        wp.pixDraws = (F.numPixelSamples > 1 ? 2u : 0u) * ft.camRays;
        wp.mainStart = wp.pixDraws + (F.enableDepthOfField ? 2u * ft.camRays * F.numDepthOfFieldSamples : 0u);
        wp.chunk = ctx->chunkShadow;
        wp.chunkPath = ctx->chunkPath;
        wp.leastPath = ctx->leastPath;
        wp.nodeBlockFixed = ctx->nodeBlockFixed ? 1u : 0u;
        wp.ownShare = ctx->ownShare;
        if (wp.R == 0 || wp.R > 65535u)
            return fail(ctx, RT_ERR_UNSUPPORTED, "render: %u camera rays per pixel", wp.R);
        const bool dof = F.enableDepthOfField != 0;
        int pathPerCU = 0, shadowPerCU = 0, pathBlocksFixed = 0;
        const bool transp = F.enableTransparency != 0;
        wp.absDraws = glossyWf ? 1u : 0u;
        if (occReady)
            pathPerCU = ctx->occ.path, shadowPerCU = ctx->occ.shadow;
        else
            HIP_TRY(ctx, wavefrontOccupancy(collect, transp, blockThreads, lds, &pathPerCU, &shadowPerCU));
        if (glossyWf) // (the tree kernel stands in the path kernel's place; asking is no allocation)
            HIP_TRY(ctx, glossyOccupancy(dof, collect, blockThreads, lds, &pathPerCU));
        if (pathPerCU < 1 || shadowPerCU < 1)
            return fail(ctx, RT_ERR_DEVICE, "render: the traversing kernels do not fit a compute unit (%zu B of LDS)", lds);
        pathPerCU = std::min(pathPerCU, maxPerCU), shadowPerCU = std::min(shadowPerCU, maxPerCU);
        if (ctx->blocksPerCU > 0) { // experiments: fewer blocks than fit
            pathPerCU = std::min(pathPerCU, ctx->blocksPerCU);
            shadowPerCU = std::min(shadowPerCU, ctx->blocksPerCU);
        }
        if (ctx->pathBlocksPerCU > 0) {
            pathPerCU = std::min(pathPerCU, ctx->pathBlocksPerCU);
        } else if (ctx->pathBlocksPerCU == 0) {
            // The path kernel's waves spend much of their time waiting (a hit's shading is a chain of
            // dependent loads, a mirror chain a chain of rays), so on a small frame more waves only
            // means emptier ones that hold their slots until the last chain is done: a block per CU
            // for every 32 K camera rays per CU, at least one (measured: eighth to half of the headline
            // frame 4-5 % faster with frames in flight, 6 % alone; the whole frame keeps what fits).
            const uint64_t perCU = (static_cast<uint64_t>(fr.nWork) * wp.R + static_cast<uint64_t>(ctx->numCU) * 32768 - 1)
                / (static_cast<uint64_t>(ctx->numCU) * 32768);
            pathPerCU = static_cast<int>(std::min<uint64_t>(pathPerCU, std::max<uint64_t>(1, perCU)));
        }
        if (ctx->shadowBlocksPerCU > 0)
            shadowPerCU = std::min(shadowPerCU, ctx->shadowBlocksPerCU);
        if (ctx->forcePathPerCU > 0) // experiments: more blocks than the occupancy query answers
            pathPerCU = std::min(ctx->forcePathPerCU, maxPerCU);
        if (ctx->forceShadowPerCU > 0)
            shadowPerCU = std::min(ctx->forceShadowPerCU, maxPerCU);
        // per light sample: its contribution (12 B) and its ray, 32 B; with transparency the ray record also
        // carries the light's colour (48 B) and a visibility byte goes with it
        // (without it, four bytes for the list of samples the replay has to decide)
        const uint64_t perItem = transp ? (12ull + 48 + 1) : (12ull + 32 + 4);
        const uint64_t perNodeBytes = 64 + sizeof(NodeB) + perItem * std::max(1u, slots);
        // node slots a wavefront of the path kernel reserves at a time; from a fixed arena the smallest
        // the kernel knows, and only as many path blocks as keep the unused ends of those blocks within
        // an eighth of the arena
        wp.nodeBlockMax = arena ? 64u : kNodeBlock;
        size_t workspaceCap = ctx->workspaceCap;
        if (arena) {
            workspaceCap = arena->bytes - arena->used;
            const uint64_t perBlock = static_cast<uint64_t>(blockThreads / 64) * wp.nodeBlockMax * perNodeBytes;
            const uint64_t blocksFit = std::max<uint64_t>(1, (workspaceCap / 8) / std::max<uint64_t>(1, perBlock));
            if (static_cast<uint64_t>(ctx->numCU) * pathPerCU > blocksFit) {
                pathPerCU = static_cast<int>(blocksFit / ctx->numCU); // 0: fewer blocks than compute units
                if (pathPerCU == 0)
                    pathBlocksFixed = static_cast<int>(blocksFit);
            }
        }
        const int pathBlocksAll = pathBlocksFixed ? pathBlocksFixed : ctx->numCU * pathPerCU;
        // every wavefront of the path kernel may leave part of its last block of node slots unused
        const uint64_t nodeSlack = static_cast<uint64_t>(pathBlocksAll) * (blockThreads / 64) * wp.nodeBlockMax;
        // Hits one camera ray can make.  A hit below the depth cap has a mirror child if its material
        // has a ks, and with transparency a pass-through child if the material is not opaque
        // (src/recursive.cpp:43-63): a chain of maxRayDepth + 1 hits, unless some material asks for both.
        const bool mirrors = F.enableReflections && ctx->anyMirror, passes = transp && ctx->anyClear;
        uint64_t nodesPerPath = 1;
        if (glossyWf) {
            // every hit below the depth cap fans out into numGlossySamples children (src/extra.cpp:348-362), each a hit or,
            // with an environment map, a node that carries the colour it left the scene with
            if (mirrors) {
                uint64_t level = 1;
                for (int d = 0; d < F.maxRayDepth && nodesPerPath < (1ull << 40); d++) {
                    level *= std::max<uint64_t>(1, F.numGlossySamples);
                    nodesPerPath += F.numGlossySamples ? level : 0;
                }
            }
        } else if (mirrors && passes && ctx->anyMirrorAndClear)
            nodesPerPath = (2ull << F.maxRayDepth) - 1;
        else if (mirrors || passes)
            nodesPerPath = static_cast<uint64_t>(F.maxRayDepth) + 1;
        const uint64_t nodesPerPixel = static_cast<uint64_t>(wp.R) * nodesPerPath;
        // what does not shrink with the batch: the unused ends of the node blocks, the buffers' alignment
        const uint64_t fixedBytes = nodeSlack * perNodeBytes + 16 * 256;
        // item index (node * slots + slot) in 32 bits, with room for the queue head to run past the end
        // by one batch per wavefront
        const uint64_t nodeCap = (0xffffffffull - (1ull << 26)) / std::max(1u, slots);
        if (nodeCap <= nodeSlack)
            return fail(ctx, RT_ERR_UNSUPPORTED, "render: %u light samples per hit exceed the supported count", slots);

        // Which pixels can have a camera ray that enters the scene's box at all: only those make hits, and the
        // pools of hits and light samples -- nine tenths of the working memory -- are sized for them alone.
        rt_ctx::FramePlan& fp = ctx->plan;
        {
            std::vector<uint8_t> key;
            keyAdd(key, *camera), keyAdd(key, width), keyAdd(key, height), keyAdd(key, shard), keyAdd(key, nShard);
            keyAdd(key, ctx->generation), keyAdd(key, F.enableAccelStructure), keyAdd(key, F.enableDepthOfField);
            if (key != fp.tilesKey) {
                candidatePixelsPerTile(ctx, *camera, fr, width, height, fp.tileCand, fp.validBefore, fp.rect);
                fp.tilesKey.swap(key);
                fp.planKey.clear();
            }
        }
        fr.candX0 = fp.rect[0], fr.candX1 = fp.rect[1], fr.candY0 = fp.rect[2], fr.candY1 = fp.rect[3];
        const std::vector<uint16_t>& tileCand = fp.tileCand;
        const uint32_t nTiles = fr.nWork / 64u;
        // per pixel slot: head, count, prefix (and environment colour) of each of its camera rays; per candidate
        // pixel: its queued camera rays and the worst case of its hits with their light samples
        const uint64_t perSlot = static_cast<uint64_t>(wp.R) * (4 + 1 + 4 + (F.enableEnvironmentMap ? 12 : 0));
        const uint64_t perCand = static_cast<uint64_t>(wp.R) * 32 + nodesPerPixel * perNodeBytes;
        typedef rt_ctx::Batch Batch;
        std::vector<Batch>& batches = fp.batches;
        uint64_t& maxTiles = fp.maxTiles;
        uint64_t& maxCand = fp.maxCand;
        // consecutive tiles, as many as the cap lets in; false when one tile alone exceeds it
        auto plan = [&](uint64_t cap, uint64_t fixed) {
            std::vector<uint8_t> key;
            keyAdd(key, cap), keyAdd(key, fixed), keyAdd(key, perSlot), keyAdd(key, perCand), keyAdd(key, wp.R), keyAdd(key, nodesPerPixel);
            keyAdd(key, nodeSlack), keyAdd(key, nodeCap);
            if (key == fp.planKey && !batches.empty())
                return true; // the same frame under the same limits as last time
            fp.planKey.clear();
            batches.clear();
            maxTiles = maxCand = 0;
            const uint64_t room = cap > fixed ? cap - fixed : 0;
            Batch cur { 0u, 0u, 0ull };
            for (uint32_t t = 0; t < nTiles; t++) {
                const uint64_t tiles = cur.tiles + 1ull, cand = cur.cand + tileCand[t];
                const bool fits = tiles * 64 * perSlot + cand * perCand <= room && tiles * 64 * wp.R <= 0x7fffffffull // path index in 31 bits
                    && cand * nodesPerPixel + nodeSlack <= nodeCap;
                if (!fits) {
                    if (cur.tiles == 0)
                        return false;
                    batches.push_back(cur);
                    cur = Batch { t, 0u, 0ull };
                    t--; // this tile opens the next batch
                    continue;
                }
                cur.tiles = static_cast<uint32_t>(tiles), cur.cand = cand;
                maxTiles = std::max<uint64_t>(maxTiles, tiles), maxCand = std::max<uint64_t>(maxCand, cand);
            }
            if (cur.tiles)
                batches.push_back(cur);
            fp.planKey.swap(key);
            return true;
        };
        size_t want[10];
        uint64_t maxNodes = 0;
        auto sizes = [&]() { // the ten buffers for the largest batch of the plan; returns their sum
            const uint64_t paths = maxTiles * 64 * wp.R, queued = std::max<uint64_t>(1, maxCand * wp.R);
            maxNodes = maxCand * nodesPerPixel + nodeSlack;
            const size_t items = static_cast<size_t>(maxNodes) * std::max(1u, slots);
            const size_t w[10] = { static_cast<size_t>(paths) * 4, static_cast<size_t>(paths), static_cast<size_t>(paths) * 4,
                F.enableEnvironmentMap ? static_cast<size_t>(paths) * 12 : 16, static_cast<size_t>(maxNodes) * 64,
                static_cast<size_t>(maxNodes) * sizeof(NodeB), items * 12, items * (transp ? 48 : 32), transp ? items : items * 4,
                static_cast<size_t>(queued) * 32 };
            size_t total = 0;
            for (int i = 0; i < 10; i++)
                want[i] = w[i], total += (w[i] + 255) & ~size_t(255);
            return total;
        };
        // (RT_OPT_WORKSPACE_BYTES caps what grows with the batch; memory that is really there -- an arena, or
        // what the device has free, below -- has to hold the fixed part as well)
        // (a frame that had to be cut down to the memory the device had left is planned under that limit again, so that
        // the plan kept from it is found and its workspace is not given back and asked for anew frame after frame;
        // the limit is forgotten with the scene)
        const bool clamped = !arena && ctx->wsClamp != 0 && ctx->wsClampGeneration == ctx->generation;
        if (!plan(clamped ? std::min<uint64_t>(workspaceCap, ctx->wsClamp) : workspaceCap, (arena || clamped) ? fixedBytes : 0))
            return fail(ctx, RT_ERR_UNSUPPORTED, "render: one 8x8 tile of this frame needs %llu bytes of working memory, more than the %zu-byte %s",
                static_cast<unsigned long long>(64 * (perSlot + perCand) + fixedBytes), workspaceCap, arena ? "scratch left" : "workspace");
        size_t total = sizes();
        Arena own { nullptr, 0, 0 };
        if (!arena) {
            // The context's own working memory: one allocation, grown when a frame asks for more.  All sizes are
            // known before anything is freed, and a frame never asks for more than the device has left
            // (with what the context already holds): it is rendered in more pixel batches instead.
            if (total > ctx->wsBytes) {
                size_t freeB = 0, totalB = 0;
                HIP_TRY(ctx, hipMemGetInfo(&freeB, &totalB));
                const size_t allowed = ctx->wsBytes + freeB - freeB / 8;
                if (total > allowed) {
                    if (!plan(allowed, fixedBytes))
                        return fail(ctx, RT_ERR_DEVICE, "render: %zu bytes of device memory are free, one 8x8 tile of this frame needs %llu",
                            freeB, static_cast<unsigned long long>(64 * (perSlot + perCand) + fixedBytes));
                    total = sizes();
                    ctx->wsClamp = allowed;
                    ctx->wsClampGeneration = ctx->generation;
                }
                HIP_TRY(ctx, hipStreamSynchronize(stream)); // (frames still in flight on this stream use the old buffer)
                if (ctx->wsBase)
                    (void)hipFree(ctx->wsBase);
                ctx->wsBase = nullptr;
                ctx->wsBytes = 0;
                HIP_TRY(ctx, hipMalloc(&ctx->wsBase, total));
                ctx->wsBytes = total;
                if (getenv("RT_AMD_DEBUG_COUNTERS")) {
                    uint64_t cand = 0;
                    for (uint16_t c : tileCand)
                        cand += c;
                    fprintf(stderr, "[rt_amd] working memory %.2f GB for %zu batch(es): %llu of %llu pixels can see the scene's box\n", total / 1e9,
                        batches.size(), static_cast<unsigned long long>(cand), static_cast<unsigned long long>(fr.nWork));
                }
            }
            own = Arena { static_cast<char*>(ctx->wsBase), ctx->wsBytes, 0 };
            arena = &own;
        }
        void* ws[10];
        for (int i = 0; i < 10; i++) {
            ws[i] = arena->take(want[i]);
            if (!ws[i])
                return fail(ctx, RT_ERR_INTERNAL, "render: working buffers exceed their memory (%zu of %zu bytes used, %zu more asked)",
                    arena->used, arena->bytes, want[i]);
        }
        WfBuffers wb {};
        wb.pathHead = static_cast<int32_t*>(ws[0]);
        wb.pathCount = static_cast<uint8_t*>(ws[1]);
        wb.pathBase = static_cast<uint32_t*>(ws[2]);
        wb.pathEnv = static_cast<float*>(ws[3]);
        wb.nodeA = static_cast<float4*>(ws[4]);
        wb.nodeB = static_cast<NodeB*>(ws[5]);
        wb.contrib = static_cast<float*>(ws[6]);
        wb.srays = static_cast<float4*>(ws[7]);
        wb.svis = static_cast<uint8_t*>(ws[8]);
        wb.redo = static_cast<uint32_t*>(ws[8]);
        wb.camRays = static_cast<float4*>(ws[9]);
        wb.maxNodes = static_cast<uint32_t>(maxNodes);

        for (const Batch& bt : batches) {
            const uint64_t begin = static_cast<uint64_t>(bt.firstTile) * 64;
            wp.slotBegin = static_cast<uint32_t>(begin);
            wp.nSlots = bt.tiles * 64u;
            if (begin != 0) // queue heads and the node allocator start over; the ray counters keep adding up
                HIP_TRY(ctx, hipMemsetAsync(reinterpret_cast<char*>(dCounters) + offsetof(DevCounters, next), 0,
                    offsetof(DevCounters, loopIters) - offsetof(DevCounters, next), stream));
            // real pixels of the batch's tiles (edge tiles are padded to 8x8 slots)
            const uint64_t validPixels = fp.validBefore[bt.firstTile + bt.tiles] - fp.validBefore[bt.firstTile];
            wp.validPaths = static_cast<uint32_t>(validPixels * wp.R);
            const int pathBlocks = pathBlocksAll;
            const int shadowBlocks = ctx->numCU * shadowPerCU;
            hipEvent_t* marks = nullptr;
            if (timeIt && ctx->marksUsed + 4 > 4 * kMaxTimedLaunches) { // a frame of very many batches, or many unread frames
                HIP_TRY(ctx, hipStreamSynchronize(stream));
                // this frame's own start event sits at evPool[evUsed], behind the finished pairs: those and the
                // batches so far are folded, the start event moves to the front of the emptied pool
                const size_t open = ctx->evUsed;
                const int rf = foldTimings(ctx);
                if (rf != RT_OK)
                    return rf;
                std::swap(ctx->evPool[0], ctx->evPool[open]);
            }
            if (timeIt) {
                while (ctx->markPool.size() < ctx->marksUsed + 4) {
                    hipEvent_t a;
                    HIP_TRY(ctx, hipEventCreate(&a));
                    ctx->markPool.push_back(a);
                }
                HIP_TRY(ctx, hipEventRecord(ctx->markPool[ctx->marksUsed], stream));
                marks = &ctx->markPool[ctx->marksUsed + 1];
                ctx->marksUsed += 4;
            }
            if (glossyWf)
                HIP_TRY(ctx, launchWavefrontGlossy(dof, collect, pathBlocks, shadowBlocks, blockThreads, lds, stream, ctx->scene, cam, ft, fr, wp,
                                  wb, dCounters, dOut, traceKnobs(ctx->minActive, ctx->parkA, ctx->parkB),
                                  traceKnobs(ctx->minActiveShadow, ctx->parkAShadow, ctx->parkBShadow), marks, ctx->numCU));
            else
                HIP_TRY(ctx, launchWavefront(dof, collect, transp, pathBlocks, shadowBlocks, blockThreads, lds, stream, ctx->scene, cam, ft, fr, wp,
                                  wb, dCounters, dOut, traceKnobs(ctx->minActive, ctx->parkA, ctx->parkB),
                                  traceKnobs(ctx->minActiveShadow, ctx->parkAShadow, ctx->parkBShadow), marks, ctx->numCU));
        }
    }
    if (nShard == 1) { // a shard has no neighbours: a sharded frame is post-processed after its gather (rt_postprocess_device)
        const int rb = bloomDevice(ctx, dOut, width, height, stream, bloomCopy);
        if (rb != RT_OK)
            return rb;
    }
    if (timeIt) {
        HIP_TRY(ctx, hipEventRecord(ctx->evPool[ctx->evUsed + 1], stream));
        ctx->evUsed += 2;
    }
    return lightsUsedBy(ctx, stream);
}

static int readStats(rt_ctx* ctx, rt_stats* stats, bool timed)
{
    if (!ctx->hostCounters) // large (per-wave log), so not on the stack; one per context
        ctx->hostCounters.reset(new DevCounters());
    DevCounters& c = *ctx->hostCounters;
    const bool debug = getenv("RT_AMD_DEBUG_COUNTERS") != nullptr;
    HIP_TRY(ctx, hipMemcpy(&c, ctx->lastCounters, debug ? sizeof c : offsetof(DevCounters, waveLog), hipMemcpyDeviceToHost));
    if (debug)
        fprintf(stderr, "[rt_amd] light samples of the last batch left to the replay (their occluder's leaf may be out of the reference's reach): %u\n", c.redo.v);
    if (getenv("RT_AMD_DEBUG_COUNTERS") && (c.loopIters[0][2] | c.loopIters[1][2])) {
        // wave iterations of the traversal loop and the mean number of lanes that had work in them
        const char* names[2] = { "path", "shadow" };
        for (int k = 0; k < 2; k++) {
            const double wi = c.loopIters[k][0] / 64.0, wl = c.loopIters[k][1] / 64.0, wr = c.loopIters[k][2] / 64.0;
            const double li = k ? double(c.sInner) : double(c.inner - c.sInner), lt = k ? double(c.sTris) : double(c.tris - c.sTris);
            fprintf(stderr, "[rt_amd] %s: inner wave-iters %.0f (%.1f lanes busy), leaf wave-iters %.0f (%.1f lanes testing), run() calls %.0f\n",
                names[k], wi, wi > 0 ? li / wi : 0.0, wl, wl > 0 ? lt / wl : 0.0, wr);
            if (c.cycles[k][3])
                fprintf(stderr, "[rt_amd] %s: of a wave's clocks: inner block %.3f (%.0f per iteration), leaf block %.3f (%.0f per iteration), rest of run() %.3f, outside run() %.3f\n",
                    names[k], double(c.cycles[k][0]) / c.cycles[k][3], wi > 0 ? c.cycles[k][0] / wi : 0.0, double(c.cycles[k][1]) / c.cycles[k][3],
                    wl > 0 ? c.cycles[k][1] / wl : 0.0, (double(c.cycles[k][2]) - c.cycles[k][0] - c.cycles[k][1]) / c.cycles[k][3],
                    1.0 - double(c.cycles[k][2]) / c.cycles[k][3]);
            fprintf(stderr, "[rt_amd] %s: %llu waves traced rays; longest wave %.0f clocks, mean of all waves %.0f\n", names[k],
                (unsigned long long)c.busyWaves[k], double(c.maxCycles[k]), wr > 0 ? double(c.cycles[k][3]) / double(std::max<unsigned long long>(1, c.busyWaves[k])) : 0.0);
            { // when the waves of the kernel started and ended, in 10 ns ticks after the first start
                std::vector<unsigned long long> st, en;
                unsigned long long t0 = ~0ull;
                for (int w = 0; w < 8192; w++)
                    if (c.waveLog[k][w][1])
                        t0 = std::min(t0, c.waveLog[k][w][0]);
                for (int w = 0; w < 8192; w++)
                    if (c.waveLog[k][w][1])
                        st.push_back(c.waveLog[k][w][0] - t0), en.push_back(c.waveLog[k][w][1] - t0);
                if (!st.empty()) {
                    std::sort(st.begin(), st.end());
                    std::sort(en.begin(), en.end());
                    auto q = [](const std::vector<unsigned long long>& v, double f) { return (double)v[std::min(v.size() - 1, (size_t)(f * v.size()))]; };
                    fprintf(stderr, "[rt_amd] %s: %zu waves; start ticks p50 %.0f p90 %.0f max %.0f; end ticks p10 %.0f p50 %.0f p90 %.0f p99 %.0f max %.0f (10 ns each)\n",
                        names[k], st.size(), q(st, 0.5), q(st, 0.9), (double)st.back(), q(en, 0.1), q(en, 0.5), q(en, 0.9), q(en, 0.99), (double)en.back());
                }
            }
            fprintf(stderr, "[rt_amd] %s: share of inner steps at pair records below 1024 / 2048 / 4096: %.3f %.3f %.3f (%u levels)\n", names[k],
                li > 0 ? c.topHits[k][0] / li : 0.0, li > 0 ? c.topHits[k][1] / li : 0.0, li > 0 ? c.topHits[k][2] / li : 0.0, ctx->numLevels);
        }
    }
    if (c.overflow) {
        ctx->evUsed = ctx->marksUsed = 0;
        return fail(ctx, RT_ERR_INTERNAL, "render: %llu hits found no slot in the hit pool (the frame is invalid; this is a bug in the pool's sizing)",
            static_cast<unsigned long long>(c.overflow));
    }
    if (stats) {
        stats->rays = c.rays;
        stats->inner_nodes = c.inner;
        stats->tri_tests = c.tris;
        stats->shadow_rays = c.sRays;
        stats->shadow_inner_nodes = c.sInner;
        stats->shadow_tri_tests = c.sTris;
        stats->kernel_ms = stats->path_ms = stats->shadow_ms = 0.0f;
        stats->launches = stats->batches = 0;
        if (timed) {
            const int rf = foldTimings(ctx);
            if (rf != RT_OK)
                return rf;
            stats->kernel_ms = static_cast<float>(ctx->sumKernelMs);
            stats->path_ms = static_cast<float>(ctx->sumPathMs);
            stats->shadow_ms = static_cast<float>(ctx->sumShadowMs);
            stats->launches = static_cast<uint32_t>(ctx->sumLaunches);
            stats->batches = static_cast<uint32_t>(ctx->sumBatches);
            ctx->sumKernelMs = ctx->sumPathMs = ctx->sumShadowMs = 0.0;
            ctx->sumLaunches = ctx->sumBatches = 0;
        }
    }
    ctx->evUsed = ctx->marksUsed = 0;
    return RT_OK;
}

extern "C" {

size_t rt_shard_floats(int width, int height, int shard_index, int shard_count)
{
    if (width <= 0 || height <= 0 || shard_count <= 0 || shard_index < 0 || shard_index >= shard_count)
        return 0;
    if (shard_count == 1)
        return static_cast<size_t>(width) * height * 3;
    return shardTiles(width, height, shard_index, shard_count) * 64 * 3;
}

int rt_render(rt_ctx* ctx, const rt_camera* camera, int width, int height, float* out_rgb, rt_stats* stats)
try {
    if (!ctx)
        return fail(nullptr, RT_ERR_INVALID, "rt_render: ctx is NULL");
    if (!out_rgb)
        return fail(ctx, RT_ERR_INVALID, "rt_render: out_rgb is NULL");
    if (width <= 0 || height <= 0)
        return fail(ctx, RT_ERR_INVALID, "rt_render: bad size %dx%d", width, height);
    const auto t0 = std::chrono::steady_clock::now();
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const size_t floats = static_cast<size_t>(width) * height * 3;
    if (floats > ctx->frameCap) {
        if (ctx->dFrame)
            (void)hipFree(ctx->dFrame);
        ctx->dFrame = nullptr;
        ctx->frameCap = 0;
        HIP_TRY(ctx, hipMalloc(reinterpret_cast<void**>(&ctx->dFrame), floats * sizeof(float)));
        ctx->frameCap = floats;
    }
    const int rc = renderCommon(ctx, camera, width, height, 0, 1, ctx->dFrame, floats, nullptr, false, ctx->timing || stats != nullptr);
    if (rc != RT_OK)
        return rc;
    HIP_TRY(ctx, hipMemcpy(out_rgb, ctx->dFrame, floats * sizeof(float), hipMemcpyDeviceToHost));
    const int rs = readStats(ctx, stats, true);
    if (rs != RT_OK)
        return rs;
    if (stats)
        stats->total_ms = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t0).count();
    return RT_OK;
} catch (...) {
    return caught(ctx, "rt_render");
}

int rt_render_device(rt_ctx* ctx, const rt_camera* camera, int width, int height, int shard_index, int shard_count,
    void* d_out, size_t d_out_floats, void* stream, int collect_stats, rt_stats* stats)
try {
    if (!ctx)
        return fail(nullptr, RT_ERR_INVALID, "rt_render_device: ctx is NULL");
    const auto t0 = std::chrono::steady_clock::now();
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const int rc = renderCommon(ctx, camera, width, height, shard_index, shard_count, static_cast<float*>(d_out), d_out_floats,
        static_cast<hipStream_t>(stream), collect_stats != 0, ctx->timing || collect_stats != 0);
    if (rc != RT_OK)
        return rc;
    if (collect_stats) {
        HIP_TRY(ctx, hipStreamSynchronize(static_cast<hipStream_t>(stream)));
        const int rs = readStats(ctx, stats, true);
        if (rs != RT_OK)
            return rs;
        if (stats)
            stats->total_ms = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t0).count();
    }
    return RT_OK;
} catch (...) {
    return caught(ctx, "rt_render_device");
}

int rt_render_device_noalloc(rt_ctx* ctx, const rt_camera* camera, int width, int height, int shard_index, int shard_count,
    void* d_out, size_t d_out_floats, void* stream)
try {
    if (!ctx)
        return fail(nullptr, RT_ERR_INVALID, "rt_render_device_noalloc: ctx is NULL");
    if (!ctx->scratchPtr)
        return fail(ctx, RT_ERR_UNSUPPORTED, "rt_render_device_noalloc: no scratch on device %d (ordinals up to %d)", ctx->device, kMaxDevices - 1);
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    Arena arena { static_cast<char*>(ctx->scratchPtr), kScratchBytes, 0 };
    return renderCommon(ctx, camera, width, height, shard_index, shard_count, static_cast<float*>(d_out), d_out_floats,
        static_cast<hipStream_t>(stream), false, false, &arena);
} catch (...) {
    return caught(ctx, "rt_render_device_noalloc");
}

int rt_postprocess_device(rt_ctx* ctx, void* d_frame, int width, int height, void* stream)
try {
    if (!ctx)
        return fail(nullptr, RT_ERR_INVALID, "rt_postprocess_device: ctx is NULL");
    if (!ctx->haveFeatures)
        return fail(ctx, RT_ERR_STATE, "rt_postprocess_device: no features set (rt_features_set)");
    if (!d_frame || width <= 0 || height <= 0)
        return fail(ctx, RT_ERR_INVALID, "rt_postprocess_device: bad argument");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    return bloomDevice(ctx, static_cast<float*>(d_frame), width, height, static_cast<hipStream_t>(stream));
} catch (...) {
    return caught(ctx, "rt_postprocess_device");
}

int rt_pack_bmp_device(rt_ctx* ctx, const void* d_frame, int width, int height, void* d_bgra, void* stream)
try {
    if (!ctx)
        return fail(nullptr, RT_ERR_INVALID, "rt_pack_bmp_device: ctx is NULL");
    if (!d_frame || !d_bgra || width <= 0 || height <= 0)
        return fail(ctx, RT_ERR_INVALID, "rt_pack_bmp_device: bad argument");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, launchPackBmp(static_cast<hipStream_t>(stream), static_cast<const float*>(d_frame), static_cast<uint8_t*>(d_bgra), width, height));
    return RT_OK;
} catch (...) {
    return caught(ctx, "rt_pack_bmp_device");
}

int rt_write_bmp(rt_ctx* ctx, const float* rgb, int width, int height, const char* path)
try {
    if (!ctx)
        return fail(nullptr, RT_ERR_INVALID, "rt_write_bmp: ctx is NULL");
    if (!rgb || !path || width <= 0 || height <= 0)
        return fail(ctx, RT_ERR_INVALID, "rt_write_bmp: bad argument");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const size_t pixels = static_cast<size_t>(width) * height;
    float* dRgb = nullptr;
    uint8_t* dOut = nullptr;
    std::vector<uint8_t> file(122 + pixels * 4);
    hipError_t e = hipMalloc(reinterpret_cast<void**>(&dRgb), pixels * 12);
    if (e == hipSuccess)
        e = hipMalloc(reinterpret_cast<void**>(&dOut), pixels * 4);
    if (e == hipSuccess)
        e = hipMemcpy(dRgb, rgb, pixels * 12, hipMemcpyHostToDevice);
    if (e == hipSuccess)
        e = launchPackBmp(nullptr, dRgb, dOut, width, height);
    if (e == hipSuccess)
        e = hipMemcpy(file.data() + 122, dOut, pixels * 4, hipMemcpyDeviceToHost);
    if (dRgb)
        (void)hipFree(dRgb);
    if (dOut)
        (void)hipFree(dOut);
    if (e != hipSuccess)
        return fail(ctx, RT_ERR_DEVICE, "rt_write_bmp: %s", hipGetErrorString(e));
    rt_bmp_header(width, height, file.data(), 122);
    FILE* fh = std::fopen(path, "wb");
    if (!fh)
        return fail(ctx, RT_ERR_INVALID, "rt_write_bmp: cannot open %s", path);
    const size_t wrote = std::fwrite(file.data(), 1, file.size(), fh);
    std::fclose(fh);
    if (wrote != file.size())
        return fail(ctx, RT_ERR_INVALID, "rt_write_bmp: short write to %s", path);
    return RT_OK;
} catch (...) {
    return caught(ctx, "rt_write_bmp");
}

size_t rt_bmp_header(int width, int height, uint8_t* out, size_t out_bytes)
{
    // stbi_write_bmp for 4 components: 14-byte file header + 108-byte V4 header,
    // BI_BITFIELDS, 32 bpp, masks R 0xff0000 G 0xff00 B 0xff A 0xff000000
    const size_t kBytes = 14 + 108;
    if (!out || out_bytes < kBytes || width <= 0 || height <= 0)
        return kBytes;
    std::memset(out, 0, kBytes);
    auto put16 = [&](size_t at, uint32_t v) { out[at] = v & 255, out[at + 1] = (v >> 8) & 255; };
    auto put32 = [&](size_t at, uint32_t v) { put16(at, v & 0xffff), put16(at + 2, v >> 16); };
    out[0] = 'B', out[1] = 'M';
    put32(2, static_cast<uint32_t>(kBytes + static_cast<size_t>(width) * height * 4));
    put32(10, static_cast<uint32_t>(kBytes));
    put32(14, 108);
    put32(18, static_cast<uint32_t>(width));
    put32(22, static_cast<uint32_t>(height));
    put16(26, 1);
    put16(28, 32);
    put32(30, 3);
    put32(54, 0xff0000u);
    put32(58, 0xff00u);
    put32(62, 0xffu);
    put32(66, 0xff000000u);
    return kBytes;
}

int rt_sync(rt_ctx* ctx, void* stream, rt_stats* stats)
try {
    if (!ctx)
        return fail(nullptr, RT_ERR_INVALID, "rt_sync: ctx is NULL");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, hipStreamSynchronize(static_cast<hipStream_t>(stream)));
    if (stats) {
        std::memset(stats, 0, sizeof *stats);
        return readStats(ctx, stats, true);
    }
    // No statistics asked for: the one counter that says whether the frame is valid is looked at all the same.
    if (ctx->lastCounters) {
        unsigned long long overflow = 0;
        HIP_TRY(ctx, hipMemcpy(&overflow, reinterpret_cast<const char*>(ctx->lastCounters) + offsetof(DevCounters, overflow), sizeof overflow,
            hipMemcpyDeviceToHost));
        if (overflow) {
            ctx->evUsed = ctx->marksUsed = 0;
            return fail(ctx, RT_ERR_INTERNAL, "render: %llu hits found no slot in the hit pool (the frame is invalid; this is a bug in the pool's sizing)",
                overflow);
        }
    }
    return RT_OK;
} catch (...) {
    return caught(ctx, "rt_sync");
}

int rt_deinterleave(rt_ctx* ctx, const void* d_packed, int width, int height, int shard_count, void* d_frame, void* stream)
try {
    if (!ctx)
        return fail(nullptr, RT_ERR_INVALID, "rt_deinterleave: ctx is NULL");
    if (!d_packed || !d_frame || width <= 0 || height <= 0 || shard_count <= 0)
        return fail(ctx, RT_ERR_INVALID, "rt_deinterleave: bad argument");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const int tilesX = (width + kTile - 1) / kTile, tilesY = (height + kTile - 1) / kTile;
    const size_t stride = shardTiles(width, height, 0, shard_count) * 64 * 3;
    HIP_TRY(ctx, launchDeinterleave(static_cast<hipStream_t>(stream), static_cast<const float*>(d_packed),
                      static_cast<float*>(d_frame), width, height, tilesX, tilesY, shard_count, stride));
    return RT_OK;
} catch (...) {
    return caught(ctx, "rt_deinterleave");
}

int rt_intersect_batch(rt_ctx* ctx, rt_ray* rays, uint32_t n, rt_hit* hits)
try {
    if (!ctx)
        return fail(nullptr, RT_ERR_INVALID, "rt_intersect_batch: ctx is NULL");
    if (!ctx->haveScene)
        return fail(ctx, RT_ERR_STATE, "rt_intersect_batch: no scene uploaded");
    if (!ctx->haveFeatures)
        return fail(ctx, RT_ERR_STATE, "rt_intersect_batch: no features set");
    if (n == 0)
        return RT_OK;
    if (!rays || !hits)
        return fail(ctx, RT_ERR_INVALID, "rt_intersect_batch: NULL array");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    rt_ray* dRays = nullptr;
    rt_hit* dHits = nullptr;
    HIP_TRY(ctx, hipMalloc(reinterpret_cast<void**>(&dRays), n * sizeof(rt_ray)));
    hipError_t e = hipMalloc(reinterpret_cast<void**>(&dHits), n * sizeof(rt_hit));
    if (e == hipSuccess)
        e = hipMemcpy(dRays, rays, n * sizeof(rt_ray), hipMemcpyHostToDevice);
    if (e == hipSuccess) {
        const DevFeat ft = makeFeat(ctx->features);
        const int blockThreads = kBlockThreads;
        const size_t lds = static_cast<size_t>(std::max<uint32_t>(ctx->numLevels, 2)) * blockThreads * 4; // as in renderCommon
        e = launchIntersect(blockThreads, lds, nullptr, ctx->scene, ft, dRays, dHits, n);
    }
    if (e == hipSuccess)
        e = hipDeviceSynchronize();
    if (e == hipSuccess)
        e = hipMemcpy(rays, dRays, n * sizeof(rt_ray), hipMemcpyDeviceToHost);
    if (e == hipSuccess)
        e = hipMemcpy(hits, dHits, n * sizeof(rt_hit), hipMemcpyDeviceToHost);
    (void)hipFree(dRays);
    if (dHits)
        (void)hipFree(dHits);
    if (e != hipSuccess)
        return fail(ctx, RT_ERR_DEVICE, "rt_intersect_batch: %s", hipGetErrorString(e));
    return RT_OK;
} catch (...) {
    return caught(ctx, "rt_intersect_batch");
}

int rt_eval_batch(rt_ctx* ctx, int kind, const float* in, uint32_t n, float* out, const void* aux)
try {
    if (!ctx)
        return fail(nullptr, RT_ERR_INVALID, "rt_eval_batch: ctx is NULL");
    static const int inW[] = { 12, 12, 9, 3, 3, 2, 3, 2, 2, 20 };
    static const int outW[] = { 3, 3, 2, 3, 3, 6, 6, 0, 6, 3 };
    if (kind < 0 || kind > RT_EVAL_SHADING)
        return fail(ctx, RT_ERR_INVALID, "rt_eval_batch: unknown kind %d", kind);
    if (n == 0)
        return RT_OK;
    if (!in || !out)
        return fail(ctx, RT_ERR_INVALID, "rt_eval_batch: NULL array");
    size_t inFloats = static_cast<size_t>(inW[kind]) * n, outFloats = static_cast<size_t>(outW[kind]) * n;
    if (kind == RT_EVAL_SAMPLER) {
        if (n != 1)
            return fail(ctx, RT_ERR_INVALID, "rt_eval_batch: the sampler kind takes one record");
        uint32_t cnt;
        std::memcpy(&cnt, in + 1, 4);
        if (cnt == 0 || cnt > (1u << 24))
            return fail(ctx, RT_ERR_INVALID, "rt_eval_batch: sampler count %u", cnt);
        outFloats = cnt;
    }
    if ((kind == RT_EVAL_TEX_NEAREST || kind == RT_EVAL_TEX_BILINEAR)) {
        if (!ctx->haveScene)
            return fail(ctx, RT_ERR_STATE, "rt_eval_batch: texture kinds need an uploaded scene");
        for (uint32_t i = 0; i < n; i++)
            if (!(in[3 * i] >= 0.0f && in[3 * i] < (float)ctx->scene.nTex))
                return fail(ctx, RT_ERR_INVALID, "rt_eval_batch: record %u names image %g of %d", i, in[3 * i], ctx->scene.nTex);
    }
    if (kind == RT_EVAL_SEGMENT_LIGHT || kind == RT_EVAL_PARALLELOGRAM) {
        const int w = inW[kind];
        for (uint32_t i = 0; i < n; i++)
            if (!(in[w * i] >= 0.0f && in[w * i] < (float)ctx->scene.nLights))
                return fail(ctx, RT_ERR_INVALID, "rt_eval_batch: record %u names light %g of %d", i, in[w * i], ctx->scene.nLights);
    }
    DevCamera cam {};
    if (kind == RT_EVAL_CAMERA_RAY) {
        if (!aux)
            return fail(ctx, RT_ERR_INVALID, "rt_eval_batch: the camera-ray kind needs aux = rt_camera*");
        cam = makeCamera(*static_cast<const rt_camera*>(aux));
    }
    rt_features f = ctx->features;
    if (!ctx->haveFeatures)
        std::memset(&f, 0, sizeof f);
    const DevFeat ft = makeFeat(f);
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    float *dIn = nullptr, *dOut = nullptr;
    HIP_TRY(ctx, hipMalloc(reinterpret_cast<void**>(&dIn), inFloats * sizeof(float)));
    hipError_t e = hipMalloc(reinterpret_cast<void**>(&dOut), outFloats * sizeof(float));
    if (e == hipSuccess)
        e = hipMemcpy(dIn, in, inFloats * sizeof(float), hipMemcpyHostToDevice);
    if (e == hipSuccess)
        e = launchEval(nullptr, ctx->scene, cam, ft, kind, dIn, dOut, n);
    if (e == hipSuccess)
        e = hipDeviceSynchronize();
    if (e == hipSuccess)
        e = hipMemcpy(out, dOut, outFloats * sizeof(float), hipMemcpyDeviceToHost);
    (void)hipFree(dIn);
    if (dOut)
        (void)hipFree(dOut);
    if (e != hipSuccess)
        return fail(ctx, RT_ERR_DEVICE, "rt_eval_batch: %s", hipGetErrorString(e));
    return RT_OK;
} catch (...) {
    return caught(ctx, "rt_eval_batch");
}

} // extern "C"
