// rt_tree.hip -- the library's own traversal hierarchy built on the device.
//
// Not the reference's tree (that is rt_build.hip / rt_bvh_build) but the one the rays are
// actually walked through: a surface-area-heuristic hierarchy over the caller's primitives
// (or leaves), stored as 32-byte sibling-pair records.  Which primitive a ray hits does not
// depend on it; how many boxes the ray tests does.  The host builder sweeps every split
// position of every axis; this one bins the items of a node into 32 slabs per axis of the
// node's box and evaluates the 31 splits between them, all nodes of a level at once:
//
//   per level   binKernel    one lane per item: its bin on each axis, bins updated with atomics
//               splitKernel  one lane per node: cheapest split, child boxes (unions of bins),
//                            children allocated, the node's pair record written
//               moveKernel   one lane per item: which child it goes to; an item alone in its
//                            child is a leaf and writes its reference into the pair record
//
// An inner node is exactly one pair record, so node number = record number; the nodes of a
// level are allocated together, which is the breadth-first order the traversal likes (the
// records every ray starts with sit next to each other).  Nothing is sorted or moved: an
// item only remembers the node it currently belongs to.

#include <hip/hip_runtime.h>

#include <cstdint>
#include <string>
#include <vector>

#include "rt_device.hpp"
#include "rt_tree.hpp"

namespace rt {

namespace {

constexpr uint32_t kBins = 32; // (16: 34.2 pair records per shadow ray of the headline frame, 32: 32.9, the host's full sweep: 32.2)
constexpr uint32_t kDone = 0xffffffffu;   // itemNode of an item that has become a leaf
constexpr uint32_t kNoChild = 0xfffffffeu; // child slot a single item will fill in

struct Item {
    float lo[3];
    uint32_t ref;
    float hi[3];
    uint32_t count;
};

struct Bin { // one slab of one axis of one node
    uint32_t items, prims;
    uint32_t lo[3], hi[3]; // box of the items in it, floats as order-preserving integers (atomic min / max)
};

struct Node {
    float lo[3], hi[3];
    uint32_t items, depth; // items in the subtree; nodes on the path from the root to here (root = 1)
    // the split chosen for it
    int32_t axis;          // -1: by ticket (first `leftItems` items to ask go left)
    uint32_t splitBin, leftItems, ticket;
    uint32_t child[2];     // node number, or kNoChild
};

__device__ __forceinline__ uint32_t ordered(float f)
{
    const uint32_t u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float unordered(uint32_t u)
{
    return __uint_as_float((u & 0x80000000u) ? (u & 0x7fffffffu) : ~u);
}

// the slab of `axis` of the node's box the item's centre falls in
__device__ __forceinline__ uint32_t binOf(const Node& nd, const Item& it, int axis)
{
    const float ext = nd.hi[axis] - nd.lo[axis];
    if (!(ext > 0.0f))
        return 0u;
    const float c = 0.5f * (it.lo[axis] + it.hi[axis]);
    const float f = (c - nd.lo[axis]) / ext * (float)kBins;
    const int b = (int)f;
    return b < 0 ? 0u : (b >= (int)kBins ? kBins - 1u : (uint32_t)b);
}

__global__ __launch_bounds__(256) void clearBinsKernel(Bin* bins, uint32_t count)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count)
        return;
    Bin b;
    b.items = b.prims = 0u;
    for (int a = 0; a < 3; a++)
        b.lo[a] = 0xffffffffu, b.hi[a] = 0u;
    bins[i] = b;
}

__global__ __launch_bounds__(256) void binKernel(const Item* __restrict__ items, uint32_t n, const uint32_t* __restrict__ itemNode,
    const Node* __restrict__ nodes, uint32_t levelBase, Bin* __restrict__ bins)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n)
        return;
    const uint32_t node = itemNode[i];
    if (node == kDone)
        return;
    const Item it = items[i];
    const Node nd = nodes[node];
    for (int axis = 0; axis < 3; axis++) {
        Bin& b = bins[((size_t)(node - levelBase) * 3u + axis) * kBins + binOf(nd, it, axis)];
        atomicAdd(&b.items, 1u);
        atomicAdd(&b.prims, it.count);
        for (int a = 0; a < 3; a++) {
            atomicMin(&b.lo[a], ordered(it.lo[a]));
            atomicMax(&b.hi[a], ordered(it.hi[a]));
        }
    }
}

__device__ __forceinline__ float halfArea(const float* lo, const float* hi)
{
    const float x = hi[0] - lo[0], y = hi[1] - lo[1], z = hi[2] - lo[2];
    return x * y + x * z + y * z;
}

__global__ __launch_bounds__(64) void splitKernel(Node* __restrict__ nodes, uint32_t levelBase, uint32_t levelCount, const Bin* __restrict__ bins,
    uint4* __restrict__ pairs, uint32_t* __restrict__ nextNode, uint32_t* __restrict__ maxDepth, QuantGrid grid, int maxStack, uint32_t capacity)
{
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= levelCount)
        return;
    const uint32_t id = levelBase + k;
    Node nd = nodes[id];
    float bestCost = 3.402823466e+38f;
    int bestAxis = -1;
    uint32_t bestSplit = 0, bestLeft = 0;
    float bl[2][3], bh[2][3]; // boxes of the two sides of the best split
    for (int axis = 0; axis < 3; axis++) {
        const Bin* b = bins + ((size_t)k * 3u + axis) * kBins;
        // boxes and counts of the slabs from the right end: suffix[s] covers slabs s .. kBins - 1
        float slo[kBins][3], shi[kBins][3];
        uint32_t sItems[kBins + 1], sPrims[kBins + 1];
        sItems[kBins] = sPrims[kBins] = 0;
        float lo[3] = { 3.402823466e+38f, 3.402823466e+38f, 3.402823466e+38f }, hi[3] = { -3.402823466e+38f, -3.402823466e+38f, -3.402823466e+38f };
        for (int s = (int)kBins - 1; s >= 0; s--) {
            if (b[s].items)
                for (int a = 0; a < 3; a++)
                    lo[a] = fminf(lo[a], unordered(b[s].lo[a])), hi[a] = fmaxf(hi[a], unordered(b[s].hi[a]));
            for (int a = 0; a < 3; a++)
                slo[s][a] = lo[a], shi[s][a] = hi[a];
            sItems[s] = sItems[s + 1] + b[s].items;
            sPrims[s] = sPrims[s + 1] + b[s].prims;
        }
        for (int a = 0; a < 3; a++)
            lo[a] = 3.402823466e+38f, hi[a] = -3.402823466e+38f;
        uint32_t lItems = 0, lPrims = 0;
        for (uint32_t s = 1; s < kBins; s++) { // split between slab s - 1 and slab s
            if (b[s - 1].items)
                for (int a = 0; a < 3; a++)
                    lo[a] = fminf(lo[a], unordered(b[s - 1].lo[a])), hi[a] = fmaxf(hi[a], unordered(b[s - 1].hi[a]));
            lItems += b[s - 1].items;
            lPrims += b[s - 1].prims;
            if (lItems == 0 || sItems[s] == 0)
                continue;
            const float cost = halfArea(lo, hi) * (float)lPrims + halfArea(slo[s], shi[s]) * (float)sPrims[s];
            if (cost < bestCost) {
                bestCost = cost, bestAxis = axis, bestSplit = s, bestLeft = lItems;
                for (int a = 0; a < 3; a++)
                    bl[0][a] = lo[a], bh[0][a] = hi[a], bl[1][a] = slo[s][a], bh[1][a] = shi[s][a];
            }
        }
    }
    // Every item in one slab on every axis (centres that coincide at this resolution), or a branch running so deep
    // that a traversal stack could not hold it: halve by ticket instead, both halves under the node's own box.
    uint32_t lg = 0;
    while ((1u << lg) < nd.items)
        lg++;
    if (bestAxis < 0 || nd.depth + 2u + lg > (uint32_t)maxStack) {
        bestAxis = -1;
        bestLeft = nd.items / 2u;
        for (int s = 0; s < 2; s++)
            for (int a = 0; a < 3; a++)
                bl[s][a] = nd.lo[a], bh[s][a] = nd.hi[a];
    }
    nd.axis = bestAxis;
    nd.splitBin = bestSplit;
    nd.leftItems = bestLeft;
    nd.ticket = 0;
    const uint32_t sideItems[2] = { bestLeft, nd.items - bestLeft };
    uint32_t refs[2], w[2][3];
    for (int s = 0; s < 2; s++) {
        grid.packBox(bl[s], bh[s], w[s]);
        if (sideItems[s] >= 2u) {
            const uint32_t c = atomicAdd(nextNode, 1u);
            if (c >= capacity) { // cannot happen with n items (at most n - 1 inner nodes); the host sees the count and gives up
                nd.child[s] = kNoChild;
                refs[s] = kRefNone;
                continue;
            }
            Node ch {};
            for (int a = 0; a < 3; a++)
                ch.lo[a] = bl[s][a], ch.hi[a] = bh[s][a];
            ch.items = sideItems[s];
            ch.depth = nd.depth + 1u;
            nodes[c] = ch;
            nd.child[s] = c;
            refs[s] = c;
        } else {
            nd.child[s] = kNoChild;
            refs[s] = kRefNone; // the one item on this side writes its own reference (moveKernel)
            atomicMax(maxDepth, nd.depth + 1u);
        }
    }
    nodes[id] = nd;
    pairs[2u * (size_t)id] = make_uint4(w[0][0], w[0][1], w[0][2], w[1][0]);
    pairs[2u * (size_t)id + 1u] = make_uint4(w[1][1], w[1][2], refs[0], refs[1]);
}

__global__ __launch_bounds__(256) void moveKernel(const Item* __restrict__ items, uint32_t n, uint32_t* __restrict__ itemNode,
    Node* __restrict__ nodes, uint32_t* __restrict__ pairWords)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n)
        return;
    const uint32_t node = itemNode[i];
    if (node == kDone)
        return;
    const Item it = items[i];
    Node& nd = nodes[node];
    int side;
    if (nd.axis >= 0)
        side = binOf(nd, it, nd.axis) < nd.splitBin ? 0 : 1;
    else
        side = atomicAdd(&nd.ticket, 1u) < nd.leftItems ? 0 : 1;
    const uint32_t child = nd.child[side];
    if (child == kNoChild) {
        pairWords[8u * (size_t)node + 6u + (uint32_t)side] = it.ref;
        itemNode[i] = kDone;
    } else {
        itemNode[i] = child;
    }
}

struct Buffers {
    std::vector<void*> all;
    template <typename T>
    hipError_t get(T** p, size_t count)
    {
        void* v = nullptr;
        const hipError_t e = hipMalloc(&v, (count ? count : 1) * sizeof(T));
        if (e == hipSuccess)
            all.push_back(v);
        *p = static_cast<T*>(v);
        return e;
    }
    ~Buffers()
    {
        for (void* v : all)
            (void)hipFree(v);
    }
};

} // namespace

#define TREE_TRY(expr)                                                                                       \
    do {                                                                                                     \
        const hipError_t e_ = (expr);                                                                        \
        if (e_ != hipSuccess) {                                                                              \
            error = std::string(#expr) + " failed: " + hipGetErrorString(e_);                                \
            return kTreeFailed;                                                                              \
        }                                                                                                    \
    } while (0)

// Returns kTreeBuilt, kTreeUnusable (the tree came out in a shape the traversal cannot take, e.g. deeper than its stack:
// the caller builds another way) or kTreeFailed (a HIP call failed); `error` says what happened.
int deviceOwnTree(const std::vector<TreeLeaf>& leaves, const QuantGrid& grid, std::vector<uint4>& pairs, uint32_t& rootRef,
    uint32_t& levels, std::string& error)
{
    const uint32_t n = static_cast<uint32_t>(leaves.size());
    pairs.clear();
    if (n == 0) {
        error = "no leaves";
        return kTreeUnusable;
    }
    if (n == 1) {
        rootRef = leaves[0].ref;
        levels = 1;
        return kTreeBuilt;
    }
    std::vector<Item> host(n);
    Node root {};
    for (int a = 0; a < 3; a++)
        root.lo[a] = leaves[0].lo[a], root.hi[a] = leaves[0].hi[a];
    for (uint32_t i = 0; i < n; i++) {
        const TreeLeaf& l = leaves[i];
        for (int a = 0; a < 3; a++) {
            host[i].lo[a] = l.lo[a], host[i].hi[a] = l.hi[a];
            root.lo[a] = std::min(root.lo[a], l.lo[a]), root.hi[a] = std::max(root.hi[a], l.hi[a]);
        }
        host[i].ref = l.ref, host[i].count = l.count;
    }
    root.items = n;
    root.depth = 1;

    Buffers buf;
    Item* dItems;
    uint32_t *dItemNode, *dCounters; // [0] next free node number, [1] deepest leaf
    Node* dNodes;
    Bin* dBins;
    uint4* dPairs;
    // an inner node has at least two items below it: at most n - 1 of them, at most n / 2 on one level
    const size_t maxPerLevel = n / 2u + 1u;
    TREE_TRY(buf.get(&dItems, n));
    TREE_TRY(buf.get(&dItemNode, n));
    TREE_TRY(buf.get(&dCounters, 2));
    TREE_TRY(buf.get(&dNodes, n));
    TREE_TRY(buf.get(&dBins, maxPerLevel * 3u * kBins));
    TREE_TRY(buf.get(&dPairs, 2 * (size_t)n));
    TREE_TRY(hipMemcpy(dItems, host.data(), n * sizeof(Item), hipMemcpyHostToDevice));
    TREE_TRY(hipMemset(dItemNode, 0, n * sizeof(uint32_t))); // every item starts in the root, node 0
    TREE_TRY(hipMemcpy(dNodes, &root, sizeof root, hipMemcpyHostToDevice));
    const uint32_t init[2] = { 1u, 0u };
    TREE_TRY(hipMemcpy(dCounters, init, sizeof init, hipMemcpyHostToDevice));

    uint32_t levelBase = 0, levelEnd = 1;
    const uint32_t itemGrid = (n + 255u) / 256u;
    for (int guard = 0; levelEnd > levelBase; guard++) {
        if (guard > 2 * kMaxStack) {
            error = "device tree build did not terminate";
            return kTreeUnusable;
        }
        const uint32_t count = levelEnd - levelBase;
        if (count > maxPerLevel) {
            error = "device tree build: more nodes on a level than items allow";
            return kTreeUnusable;
        }
        const uint32_t nBins = count * 3u * kBins;
        hipLaunchKernelGGL(clearBinsKernel, dim3((nBins + 255u) / 256u), dim3(256), 0, 0, dBins, nBins);
        hipLaunchKernelGGL(binKernel, dim3(itemGrid), dim3(256), 0, 0, dItems, n, dItemNode, dNodes, levelBase, dBins);
        hipLaunchKernelGGL(splitKernel, dim3((count + 63u) / 64u), dim3(64), 0, 0, dNodes, levelBase, count, dBins, dPairs, dCounters, dCounters + 1,
            grid, kMaxStack, n);
        hipLaunchKernelGGL(moveKernel, dim3(itemGrid), dim3(256), 0, 0, dItems, n, dItemNode, dNodes, reinterpret_cast<uint32_t*>(dPairs));
        uint32_t c[2];
        TREE_TRY(hipMemcpy(c, dCounters, sizeof c, hipMemcpyDeviceToHost));
        if (c[0] > n) {
            error = "device tree build: node numbers ran past the item count";
            return kTreeUnusable;
        }
        levelBase = levelEnd;
        levelEnd = c[0];
        levels = c[1];
    }
    TREE_TRY(hipGetLastError());
    pairs.resize(2 * (size_t)levelEnd);
    TREE_TRY(hipMemcpy(pairs.data(), dPairs, pairs.size() * sizeof(uint4), hipMemcpyDeviceToHost));
    rootRef = 0;
    if (levels + 1 > (uint32_t)kMaxStack) {
        error = "device tree deeper than a traversal stack";
        return kTreeUnusable;
    }
    return kTreeBuilt;
}

} // namespace rt
