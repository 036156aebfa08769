// rt_build.hip -- BVH::BVH (src/bvh.cpp:52-101) on the device: the reference's
// tree -- recursive median split along the longest axis of the span box, leaves of
// at most four primitives (src/bvh.cpp:208-223, :424-493) -- built level by level
// instead of depth first, with the same node and primitive arrays as a result.
//
// What makes that possible: with the split at (n + 1) / 2 the shape of the tree --
// every node's index, its range of primitives, which nodes are leaves -- follows
// from the number of primitives alone and is laid out on the host in microseconds.
// What depends on the data is which primitive ends up where and the boxes: per
// level, every node's span box (a min/max reduction, exact in any order) and, for
// the inner nodes, a sort of the node's range by the centroid coordinate of its
// longest axis.  All ranges of a level are sorted at once by one radix sort on
// (range number, coordinate); deep levels, whose ranges hold a few dozen
// primitives, by one lane per range.
//
// The reference sorts with std::ranges::sort, which is not stable: how it arranges
// primitives whose coordinates are EQUAL is a property of that one algorithm run
// on that one arrangement.  A strict order has only one sorted arrangement, so a
// range without equal coordinates comes out the same from any sort.  Ranges in which
// two coordinates are equal (or one is not a number) are flagged by the device and
// sorted again on the host, from the arrangement they had before, by the very
// std::sort call the host builder uses.  Meshes from a modelling tool rarely have
// such ranges; a box of axis-aligned quads has little else and is tiny.
//
// The other builder, Features::extra.enableBvhSahBinning (splitPrimitivesBySAHBin,
// src/extra.cpp:493-543), is here too (deviceSahBinBuild).  There the split position
// of a node depends on the data, so the shape of the tree is only known level by
// level: per level the device buckets every range into the reference's eleven
// buckets (a stable sort, like its bucket-by-bucket copy: no ties to worry about),
// evaluates its cost at every position of every range (prefix and suffix boxes by a
// blocked scan; min and max are exact in any order) and reports the first cheapest
// position per range; the host turns those into the next level's ranges, and at the
// end numbers the nodes the way the recursion would have.

#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/rt_amd.h"
#include "rt_build.hpp"

namespace rt {

namespace {

struct Seg {
    uint32_t begin, count, node, inner; // range of primitive positions, node index, 1 = the node is split, 2 = a leaf finished on an earlier level
};

constexpr uint32_t kPrimWords = sizeof(rt_primitive) / 4; // 25
constexpr uint32_t kSortTile = 1024;                      // keys per block of a radix pass (256 threads x 4)

// a float as an unsigned integer that orders the same way (negative zero below positive zero)
__device__ __forceinline__ uint32_t orderable(float f)
{
    const uint32_t u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float fromOrderable(uint32_t u)
{
    return __uint_as_float((u & 0x80000000u) ? (u & 0x7fffffffu) : ~u);
}

// computePrimitiveAABB / computePrimitiveCentroid, src/bvh.cpp:117-133, :169-175; arrays are [axis][primitive]
__global__ __launch_bounds__(256) void primInfoKernel(const uint32_t* __restrict__ prims, uint32_t n, float* __restrict__ lo,
    float* __restrict__ hi, float* __restrict__ c)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n)
        return;
    const uint32_t* p = prims + (size_t)i * kPrimWords; // meshID, then three vertices of 8 floats, position first
    for (uint32_t a = 0; a < 3; a++) {
        const float x = __uint_as_float(p[1 + a]), y = __uint_as_float(p[9 + a]), z = __uint_as_float(p[17 + a]);
        // std::min / std::max as the reference calls them: of two equal values (a zero of either sign) the first stays
        const float mnYZ = (z < y) ? z : y, mxYZ = (y < z) ? z : y;
        lo[(size_t)a * n + i] = (mnYZ < x) ? mnYZ : x;
        hi[(size_t)a * n + i] = (x < mxYZ) ? mxYZ : x;
        c[(size_t)a * n + i] = ((x + y) + z) / 3.0f;
    }
}

__global__ __launch_bounds__(256) void iotaKernel(uint32_t* order, uint32_t n)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n)
        order[i] = i;
}

// One block per node of the level: its span box (computeSpanAABB, src/bvh.cpp:137-165) into the node record,
// the longest axis (src/bvh.cpp:183-196), and the sort key of every primitive of its range: the range's number
// above the coordinate, so that one sort of the whole array sorts every range in place.  Ranges of leaves keep
// their arrangement (their keys are the positions).
__global__ __launch_bounds__(256) void levelKernel(const Seg* __restrict__ segs, const uint32_t* __restrict__ order, uint32_t n,
    const float* __restrict__ lo, const float* __restrict__ hi, const float* __restrict__ c, rt_node* __restrict__ nodes,
    unsigned long long* __restrict__ keys, float* __restrict__ keyf, int buckets)
{
    __shared__ float red[6][256];
    __shared__ uint32_t at[6][256];
    __shared__ int axisShared;
    const Seg s = segs[blockIdx.x];
    if (s.inner == 2u) { // its box was made when it became a leaf; it only has to keep its place in the sort
        for (uint32_t k = threadIdx.x; k < s.count; k += blockDim.x)
            keys[s.begin + k] = ((unsigned long long)blockIdx.x << 32) | k;
        return;
    }
    // The reference folds the boxes in the order of the range with std::min / std::max, which keep what they have
    // when the next value is equal: the result is the value that came FIRST among equal ones -- which matters for
    // nothing but the sign of a zero, and the node records are compared bit for bit.  So every partial result
    // carries the position it came from, and of two equal values the earlier one wins.
    float val[6];
    uint32_t pos[6];
    for (int q = 0; q < 6; q++)
        val[q] = 0.0f, pos[q] = 0xffffffffu; // nothing yet
    for (uint32_t k = threadIdx.x; k < s.count; k += blockDim.x) {
        const uint32_t i = order[s.begin + k];
        for (uint32_t a = 0; a < 3; a++) {
            const float l = lo[(size_t)a * n + i], h = hi[(size_t)a * n + i];
            if (pos[a] == 0xffffffffu || l < val[a])
                val[a] = l, pos[a] = k;
            if (pos[3 + a] == 0xffffffffu || val[3 + a] < h)
                val[3 + a] = h, pos[3 + a] = k;
        }
    }
    for (int q = 0; q < 6; q++)
        red[q][threadIdx.x] = val[q], at[q][threadIdx.x] = pos[q];
    __syncthreads();
    for (uint32_t w = 128; w > 0; w >>= 1) {
        if (threadIdx.x < w)
            for (int q = 0; q < 6; q++) {
                const float mine = red[q][threadIdx.x], other = red[q][threadIdx.x + w];
                const uint32_t pm = at[q][threadIdx.x], po = at[q][threadIdx.x + w];
                bool take = false; // take the other partial result?
                if (po != 0xffffffffu) {
                    if (pm == 0xffffffffu)
                        take = true;
                    else if (q < 3)
                        take = other < mine || (!(mine < other) && po < pm);
                    else
                        take = mine < other || (!(other < mine) && po < pm);
                }
                if (take)
                    red[q][threadIdx.x] = other, at[q][threadIdx.x] = po;
            }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        rt_node& nd = nodes[s.node];
        for (int a = 0; a < 3; a++)
            nd.lower[a] = red[a][0], nd.upper[a] = red[3 + a][0];
        const float xl = red[3][0] - red[0][0], yl = red[4][0] - red[1][0], zl = red[5][0] - red[2][0];
        axisShared = (xl >= yl && xl >= zl) ? 0 : (yl >= zl ? 1 : 2);
    }
    __syncthreads();
    const int axis = axisShared;
    // splitPrimitivesBySAHBin's bucket of a centroid, src/extra.cpp:501-506 (ten bins' width, eleven buckets; a
    // quotient that is negative or not a number, which is undefined there, goes to bucket 0, one beyond the last
    // bucket to the last)
    const float axisLo = red[axis][0];
    const float binSize = (red[3 + axis][0] - axisLo) / 10.0f;
    for (uint32_t k = threadIdx.x; k < s.count; k += blockDim.x) {
        const uint32_t p = s.begin + k;
        uint32_t low = k;
        if (s.inner) {
            const float f = c[(size_t)axis * n + order[p]];
            if (buckets) {
                const float q = floorf((f - axisLo) / binSize);
                uint32_t b = 0;
                if (q >= 0.0f)
                    b = q >= 10.0f ? 10u : (uint32_t)q;
                keyf[p] = (float)b;
                low = b;
            } else {
                keyf[p] = f;
                low = orderable(f);
            }
        }
        keys[p] = ((unsigned long long)blockIdx.x << 32) | low;
    }
}

// ---- least-significant-digit radix sort of (64-bit key, 32-bit value), eight bits a pass ----
__global__ __launch_bounds__(256) void radixCountKernel(const unsigned long long* __restrict__ keys, uint32_t n, uint32_t shift,
    uint32_t* __restrict__ hist, uint32_t nBlocks)
{
    __shared__ uint32_t h[256];
    h[threadIdx.x] = 0;
    __syncthreads();
    const uint32_t base = blockIdx.x * kSortTile;
    for (uint32_t k = threadIdx.x; k < kSortTile; k += 256) {
        const uint32_t i = base + k;
        if (i < n)
            atomicAdd(&h[(uint32_t)(keys[i] >> shift) & 255u], 1u);
    }
    __syncthreads();
    hist[threadIdx.x * nBlocks + blockIdx.x] = h[threadIdx.x]; // digit-major: a scan over it yields each block's bases
}

__global__ __launch_bounds__(1024) void exclusiveScanKernel(uint32_t* __restrict__ data, uint32_t m)
{
    __shared__ uint32_t sums[1024];
    const uint32_t per = (m + 1023u) / 1024u, b = threadIdx.x * per, e = (b + per < m) ? b + per : m;
    uint32_t s = 0;
    for (uint32_t i = b; i < e; i++)
        s += data[i];
    sums[threadIdx.x] = s;
    __syncthreads();
    for (uint32_t off = 1; off < 1024; off <<= 1) { // inclusive scan of the per-thread sums
        const uint32_t v = (threadIdx.x >= off) ? sums[threadIdx.x - off] : 0u;
        __syncthreads();
        sums[threadIdx.x] += v;
        __syncthreads();
    }
    uint32_t run = sums[threadIdx.x] - s;
    for (uint32_t i = b; i < e; i++) {
        const uint32_t v = data[i];
        data[i] = run;
        run += v;
    }
}

// Stable scatter: a key's place is its digit's base for this block (from the scanned histogram) plus the
// number of keys with the same digit before it in the block.  Every wave ranks 4 x 64 consecutive keys with
// ballots (which lanes hold my digit?) and a running count per digit; the waves' counts are then offset by
// those of the waves before them.
__global__ __launch_bounds__(256) void radixScatterKernel(const unsigned long long* __restrict__ keysIn, const uint32_t* __restrict__ valsIn,
    unsigned long long* __restrict__ keysOut, uint32_t* __restrict__ valsOut, uint32_t n, uint32_t shift, const uint32_t* __restrict__ hist,
    uint32_t nBlocks)
{
    // (volatile: lanes of a wave read what other lanes of it have just written)
    __shared__ volatile uint32_t cnt[4][256];
    for (uint32_t w = 0; w < 4; w++)
        cnt[w][threadIdx.x] = 0;
    __syncthreads();
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const uint32_t base = blockIdx.x * kSortTile + wave * 256u;
    unsigned long long key[4];
    uint32_t val[4], digit[4], rank[4];
    bool valid[4];
    for (uint32_t k = 0; k < 4; k++) {
        const uint32_t i = base + k * 64u + lane;
        valid[k] = i < n;
        key[k] = valid[k] ? keysIn[i] : 0ull;
        val[k] = valid[k] ? valsIn[i] : 0u;
        digit[k] = (uint32_t)(key[k] >> shift) & 255u;
        unsigned long long same = __ballot(valid[k]);
        for (uint32_t bit = 0; bit < 8; bit++) {
            const bool one = (digit[k] >> bit) & 1u;
            const unsigned long long b = __ballot(one);
            same &= one ? b : ~b;
        }
        const uint32_t before = (uint32_t)__popcll(same & ((1ull << lane) - 1ull));
        rank[k] = 0;
        if (valid[k]) {
            rank[k] = cnt[wave][digit[k]] + before; // (every lane of the group reads before its first lane adds)
            if (before == 0)
                cnt[wave][digit[k]] = rank[k] + (uint32_t)__popcll(same);
        }
    }
    __syncthreads();
    { // per digit: the waves' counts become the counts of the waves before them
        uint32_t run = 0;
        for (uint32_t w = 0; w < 4; w++) {
            const uint32_t v = cnt[w][threadIdx.x];
            cnt[w][threadIdx.x] = run;
            run += v;
        }
    }
    __syncthreads();
    for (uint32_t k = 0; k < 4; k++)
        if (valid[k]) {
            const uint32_t pos = hist[digit[k] * nBlocks + blockIdx.x] + cnt[wave][digit[k]] + rank[k];
            keysOut[pos] = key[k];
            valsOut[pos] = val[k];
        }
}

// After the sort: two equal coordinates next to each other in one range, or a coordinate that is not a
// number, and the range's arrangement is the reference's sort algorithm's to decide: flag it.
__global__ __launch_bounds__(256) void tieKernel(const unsigned long long* __restrict__ keys, uint32_t n, const Seg* __restrict__ segs,
    uint32_t* __restrict__ flags, uint32_t* __restrict__ any)
{
    const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n)
        return;
    const unsigned long long k = keys[p];
    const uint32_t seg = (uint32_t)(k >> 32);
    if (!segs[seg].inner)
        return;
    const float f = fromOrderable((uint32_t)k);
    bool tie = f != f;
    if (p > 0) {
        const unsigned long long q = keys[p - 1];
        if ((uint32_t)(q >> 32) == seg)
            tie = tie || fromOrderable((uint32_t)q) == f;
    }
    if (tie) {
        flags[seg] = 1u;
        *any = 1u;
    }
}

// Deep levels: ranges of a few dozen primitives, one lane each, sorted where they lie.
__global__ __launch_bounds__(256) void smallSortKernel(const Seg* __restrict__ segs, uint32_t nSegs, float* __restrict__ keyf,
    uint32_t* __restrict__ order, uint32_t* __restrict__ flags, uint32_t* __restrict__ any)
{
    const uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= nSegs || segs[s].inner != 1u)
        return;
    const uint32_t b = segs[s].begin, cnt = segs[s].count;
    bool tie = false;
    for (uint32_t i = 1; i < cnt; i++) {
        const float f = keyf[b + i];
        const uint32_t v = order[b + i];
        uint32_t j = i;
        while (j > 0 && f < keyf[b + j - 1]) {
            keyf[b + j] = keyf[b + j - 1];
            order[b + j] = order[b + j - 1];
            j--;
        }
        keyf[b + j] = f;
        order[b + j] = v;
    }
    for (uint32_t i = 0; i < cnt; i++) {
        const float f = keyf[b + i];
        tie = tie || f != f || (i > 0 && keyf[b + i - 1] == f);
    }
    if (tie) {
        flags[s] = 1u;
        *any = 1u;
    }
}

// computeSurfaceArea, src/extra.cpp:476-483
__device__ __forceinline__ float boxArea(const float* b)
{
    const float x = b[3] - b[0], y = b[4] - b[1], z = b[5] - b[2];
    return 2.0f * (x * y + x * z + y * z);
}
__device__ __forceinline__ void boxMerge(float* b, const float* o)
{
    for (int a = 0; a < 3; a++) {
        b[a] = (o[a] < b[a]) ? o[a] : b[a];
        b[3 + a] = (b[3 + a] < o[3 + a]) ? o[3 + a] : b[3 + a];
    }
}

// The cost loop of splitPrimitivesBySAHBin (src/extra.cpp:519-542) for every split range of a level, one block
// per range, on the bucketed arrangement: cost of position i = i * area(box of the first i primitives) +
// (m - i) * area(box of the primitives AFTER the i-th) -- the right-hand box lags one primitive behind its
// count, as in the reference -- for i in [1, m - 2]; the first cheapest position is the split.  Every thread
// owns a stretch of the range; the boxes before and after a stretch come from the other threads' stretch boxes.
__global__ __launch_bounds__(256) void sahSplitKernel(const Seg* __restrict__ segs, const uint32_t* __restrict__ order, uint32_t n,
    const float* __restrict__ lo, const float* __restrict__ hi, float* __restrict__ costLeft, uint32_t* __restrict__ splits)
{
    __shared__ float part[256][6];
    __shared__ float bestCost[256];
    __shared__ uint32_t bestAt[256];
    __shared__ int firstIsNaN;
    const Seg s = segs[blockIdx.x];
    if (s.inner != 1u)
        return;
    const uint32_t m = s.count, stretch = (m + 255u) / 256u;
    const uint32_t kb = min(threadIdx.x * stretch, m), ke = min(kb + stretch, m);
    const float inf = __builtin_inff();
    auto primBox = [&](uint32_t k, float* b) {
        const uint32_t i = order[s.begin + k];
        for (uint32_t a = 0; a < 3; a++)
            b[a] = lo[(size_t)a * n + i], b[3 + a] = hi[(size_t)a * n + i];
    };
    float mine[6] = { inf, inf, inf, -inf, -inf, -inf };
    for (uint32_t k = kb; k < ke; k++) {
        float b[6];
        primBox(k, b);
        boxMerge(mine, b);
    }
    for (int q = 0; q < 6; q++)
        part[threadIdx.x][q] = mine[q];
    if (threadIdx.x == 0)
        firstIsNaN = 0;
    __syncthreads();
    float before[6] = { inf, inf, inf, -inf, -inf, -inf }, after[6] = { inf, inf, inf, -inf, -inf, -inf };
    for (uint32_t t = 0; t < 256u; t++) {
        if (t < threadIdx.x)
            boxMerge(before, part[t]);
        else if (t > threadIdx.x)
            boxMerge(after, part[t]);
    }
    // left to right: the box of everything before position k
    for (uint32_t k = kb; k < ke; k++) {
        if (k >= 1u && k + 2u <= m)
            costLeft[s.begin + k] = (float)k * boxArea(before);
        float b[6];
        primBox(k, b);
        boxMerge(before, b);
    }
    // right to left: the box of everything after position k; going down, an equal cost at a lower position wins
    float best = 0.0f;
    uint32_t at = 0xffffffffu;
    for (uint32_t k = ke; k > kb; k--) {
        const uint32_t kk = k - 1u;
        if (kk >= 1u && kk + 2u <= m) {
            const float cost = costLeft[s.begin + kk] + (float)(m - kk) * boxArea(after);
            if (kk == 1u && cost != cost)
                firstIsNaN = 1; // the reference's running minimum starts here and nothing compares below it
            if (cost == cost && (at == 0xffffffffu || cost <= best))
                best = cost, at = kk;
        }
        float b[6];
        primBox(kk, b);
        boxMerge(after, b);
    }
    bestCost[threadIdx.x] = best;
    bestAt[threadIdx.x] = at;
    __syncthreads();
    for (uint32_t w = 128; w > 0; w >>= 1) {
        if (threadIdx.x < w) {
            const uint32_t oa = bestAt[threadIdx.x + w], ma = bestAt[threadIdx.x];
            const float oc = bestCost[threadIdx.x + w], mc = bestCost[threadIdx.x];
            if (oa != 0xffffffffu && (ma == 0xffffffffu || oc < mc || (oc == mc && oa < ma)))
                bestCost[threadIdx.x] = oc, bestAt[threadIdx.x] = oa;
        }
        __syncthreads();
    }
    if (threadIdx.x == 0)
        splits[blockIdx.x] = (firstIsNaN || bestAt[0] == 0xffffffffu) ? 1u : bestAt[0];
}

__global__ __launch_bounds__(256) void gatherPrimsKernel(const uint32_t* __restrict__ in, const uint32_t* __restrict__ order, uint32_t n,
    uint32_t* __restrict__ out)
{
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (size_t)n * kPrimWords)
        return;
    const uint32_t i = (uint32_t)(t / kPrimWords), w = (uint32_t)(t % kPrimWords);
    out[t] = in[(size_t)order[i] * kPrimWords + w];
}

struct KeyIdx {
    float key;
    uint32_t idx;
};

struct DeviceBuffers {
    std::vector<void*> all;
    template <typename T>
    hipError_t get(T** p, size_t count)
    {
        void* v = nullptr;
        const hipError_t e = hipMalloc(&v, std::max<size_t>(count, 1) * sizeof(T));
        if (e == hipSuccess)
            all.push_back(v);
        *p = static_cast<T*>(v);
        return e;
    }
    ~DeviceBuffers()
    {
        for (void* v : all)
            (void)hipFree(v);
    }
};

} // namespace

#define BUILD_TRY(expr)                                                                                      \
    do {                                                                                                     \
        const hipError_t e_ = (expr);                                                                        \
        if (e_ != hipSuccess) {                                                                              \
            error = std::string(#expr) + " failed: " + hipGetErrorString(e_);                                \
            return false;                                                                                    \
        }                                                                                                    \
    } while (0)

bool deviceMedianBuild(const rt_primitive* in, uint32_t n, rt_node* nodesOut, rt_primitive* primsOut, uint32_t* nNodes,
    uint32_t* nLevels, uint32_t* nLeaves, DeviceBuildStats* stats, std::string& error)
{
    // ---- the tree's shape, from n alone: node indices in the order buildRecursive hands them out (children
    // allocated when their parent is reached, left before right, src/bvh.cpp:482-483), ranges, levels
    std::vector<rt_node> nodes(2);
    std::memset(nodes.data(), 0, 2 * sizeof(rt_node));
    nodes[1].upper[0] = nodes[1].upper[1] = nodes[1].upper[2] = 1.0f; // the unused node that keeps siblings adjacent (src/bvh.cpp:88-89)
    std::vector<std::vector<Seg>> levels;
    uint32_t leaves = 0;
    struct Item {
        uint32_t begin, count, node, level;
    };
    std::vector<Item> work { { 0u, n, 0u, 0u } };
    while (!work.empty()) {
        const Item it = work.back();
        work.pop_back();
        if (levels.size() <= it.level)
            levels.resize(it.level + 1);
        const bool inner = it.count > 4;
        levels[it.level].push_back({ it.begin, it.count, it.node, inner ? 1u : 0u });
        if (!inner) {
            nodes[it.node].data[0] = it.begin | RT_LEAF_BIT; // leaves append their primitives in depth-first order: the positions
            nodes[it.node].data[1] = it.count;
            leaves++;
            continue;
        }
        const uint32_t split = (it.count + 1) / 2; // src/bvh.cpp:222
        const uint32_t left = static_cast<uint32_t>(nodes.size());
        nodes.emplace_back();
        nodes.emplace_back();
        nodes[it.node].data[0] = left;
        nodes[it.node].data[1] = left + 1;
        // depth first, left subtree before the right one: the right child waits on the stack
        work.push_back({ it.begin + split, it.count - split, left + 1, it.level + 1 });
        work.push_back({ it.begin, split, left, it.level + 1 });
    }
    *nNodes = static_cast<uint32_t>(nodes.size());
    *nLevels = static_cast<uint32_t>(levels.size());
    *nLeaves = leaves;
    if (stats)
        *stats = DeviceBuildStats {};
    if (n == 0) { // computeSpanAABB of an empty span: AxisAlignedBox{} (src/bvh.cpp:142-144)
        nodes[0].upper[0] = nodes[0].upper[1] = nodes[0].upper[2] = 1.0f;
        std::copy(nodes.begin(), nodes.end(), nodesOut);
        return true;
    }

    // ---- device side
    DeviceBuffers buf;
    uint32_t *dPrims, *dPrimsOut, *dOrder[2], *dOrderPrev, *dHist, *dFlags, *dAny;
    float *dLo, *dHi, *dC, *dKeyf;
    unsigned long long* dKeys[2];
    rt_node* dNodes;
    Seg* dSegs;
    const uint32_t nBlocks = (n + kSortTile - 1) / kSortTile;
    size_t maxSegs = 0;
    for (const auto& l : levels)
        maxSegs = std::max(maxSegs, l.size());
    BUILD_TRY(buf.get(&dPrims, (size_t)n * kPrimWords));
    BUILD_TRY(buf.get(&dPrimsOut, (size_t)n * kPrimWords));
    BUILD_TRY(buf.get(&dOrder[0], n));
    BUILD_TRY(buf.get(&dOrder[1], n));
    BUILD_TRY(buf.get(&dOrderPrev, n));
    BUILD_TRY(buf.get(&dKeys[0], n));
    BUILD_TRY(buf.get(&dKeys[1], n));
    BUILD_TRY(buf.get(&dKeyf, n));
    BUILD_TRY(buf.get(&dLo, (size_t)3 * n));
    BUILD_TRY(buf.get(&dHi, (size_t)3 * n));
    BUILD_TRY(buf.get(&dC, (size_t)3 * n));
    BUILD_TRY(buf.get(&dHist, (size_t)256 * nBlocks));
    BUILD_TRY(buf.get(&dFlags, maxSegs));
    BUILD_TRY(buf.get(&dAny, 1));
    BUILD_TRY(buf.get(&dNodes, nodes.size()));
    BUILD_TRY(buf.get(&dSegs, maxSegs));
    BUILD_TRY(hipMemcpy(dPrims, in, (size_t)n * sizeof(rt_primitive), hipMemcpyHostToDevice));
    BUILD_TRY(hipMemcpy(dNodes, nodes.data(), nodes.size() * sizeof(rt_node), hipMemcpyHostToDevice));
    const uint32_t grid = (n + 255) / 256;
    hipLaunchKernelGGL(primInfoKernel, dim3(grid), dim3(256), 0, 0, dPrims, n, dLo, dHi, dC);
    hipLaunchKernelGGL(iotaKernel, dim3(grid), dim3(256), 0, 0, dOrder[0], n);
    int cur = 0; // which of dOrder / dKeys holds the current arrangement

    std::vector<uint32_t> flags, hostPrev, hostOrder;
    std::vector<KeyIdx> scratch;
    std::vector<rt_node> hostNodes;
    for (size_t lv = 0; lv < levels.size(); lv++) {
        const std::vector<Seg>& segs = levels[lv];
        const uint32_t nSegs = static_cast<uint32_t>(segs.size());
        // (ranges of a level are disjoint but need not cover the array: leaves higher up keep their part)
        uint32_t maxCount = 0, inners = 0;
        for (const Seg& s : segs)
            if (s.inner)
                maxCount = std::max(maxCount, s.count), inners++;
        BUILD_TRY(hipMemcpyAsync(dSegs, segs.data(), nSegs * sizeof(Seg), hipMemcpyHostToDevice, 0));
        if (inners == 0) { // a level of leaves: boxes only
            hipLaunchKernelGGL(levelKernel, dim3(nSegs), dim3(256), 0, 0, dSegs, dOrder[cur], n, dLo, dHi, dC, dNodes, dKeys[cur], dKeyf, 0);
            continue;
        }
        BUILD_TRY(hipMemsetAsync(dFlags, 0, nSegs * sizeof(uint32_t), 0));
        BUILD_TRY(hipMemsetAsync(dAny, 0, sizeof(uint32_t), 0));
        BUILD_TRY(hipMemcpyAsync(dOrderPrev, dOrder[cur], n * sizeof(uint32_t), hipMemcpyDeviceToDevice, 0));
        if (maxCount <= 64) {
            hipLaunchKernelGGL(levelKernel, dim3(nSegs), dim3(256), 0, 0, dSegs, dOrder[cur], n, dLo, dHi, dC, dNodes, dKeys[cur], dKeyf, 0);
            hipLaunchKernelGGL(smallSortKernel, dim3((nSegs + 255) / 256), dim3(256), 0, 0, dSegs, nSegs, dKeyf, dOrder[cur], dFlags, dAny);
        } else {
            // Ranges this large only exist while every node of the level above was split, so they tile the whole
            // array, in the order they were listed (left to right): the block number is the range number.
            uint32_t at = 0;
            for (const Seg& s : segs) {
                if (s.begin != at || !s.inner) {
                    error = "device build: ranges of a level with large nodes do not tile the primitive array";
                    return false;
                }
                at += s.count;
            }
            hipLaunchKernelGGL(levelKernel, dim3(nSegs), dim3(256), 0, 0, dSegs, dOrder[cur], n, dLo, dHi, dC, dNodes, dKeys[cur], dKeyf, 0);
            uint32_t segBits = 0;
            while ((1u << segBits) < nSegs)
                segBits++;
            const uint32_t passes = 4 + (segBits + 7) / 8;
            for (uint32_t p = 0; p < passes; p++) {
                const uint32_t shift = 8 * p;
                hipLaunchKernelGGL(radixCountKernel, dim3(nBlocks), dim3(256), 0, 0, dKeys[cur], n, shift, dHist, nBlocks);
                hipLaunchKernelGGL(exclusiveScanKernel, dim3(1), dim3(1024), 0, 0, dHist, 256u * nBlocks);
                hipLaunchKernelGGL(radixScatterKernel, dim3(nBlocks), dim3(256), 0, 0, dKeys[cur], dOrder[cur], dKeys[cur ^ 1], dOrder[cur ^ 1], n,
                    shift, dHist, nBlocks);
                cur ^= 1;
            }
            hipLaunchKernelGGL(tieKernel, dim3(grid), dim3(256), 0, 0, dKeys[cur], n, dSegs, dFlags, dAny);
        }
        {
            uint32_t any = 0;
            BUILD_TRY(hipMemcpy(&any, dAny, sizeof any, hipMemcpyDeviceToHost));
            if (!any)
                continue;
            flags.resize(nSegs);
            BUILD_TRY(hipMemcpy(flags.data(), dFlags, nSegs * sizeof(uint32_t), hipMemcpyDeviceToHost));
        }
        // the flagged ranges, sorted the reference's way from the arrangement they had before the device's sort;
        // one transfer each way per level, however many ranges there are
        hostNodes.resize(nodes.size());
        hostPrev.resize(n);
        hostOrder.resize(n);
        BUILD_TRY(hipMemcpy(hostNodes.data(), dNodes, nodes.size() * sizeof(rt_node), hipMemcpyDeviceToHost));
        BUILD_TRY(hipMemcpy(hostPrev.data(), dOrderPrev, n * sizeof(uint32_t), hipMemcpyDeviceToHost));
        BUILD_TRY(hipMemcpy(hostOrder.data(), dOrder[cur], n * sizeof(uint32_t), hipMemcpyDeviceToHost));
        for (uint32_t s = 0; s < nSegs; s++) {
            if (!flags[s])
                continue;
            const Seg& g = segs[s];
            const rt_node& nd = hostNodes[g.node];
            const float xl = nd.upper[0] - nd.lower[0], yl = nd.upper[1] - nd.lower[1], zl = nd.upper[2] - nd.lower[2];
            const int axis = (xl >= yl && xl >= zl) ? 0 : (yl >= zl ? 1 : 2);
            scratch.resize(g.count);
            for (uint32_t i = 0; i < g.count; i++) {
                const uint32_t idx = hostPrev[g.begin + i];
                const rt_primitive& p = in[idx];
                const float x = p.v0.position[axis], y = p.v1.position[axis], z = p.v2.position[axis];
                scratch[i] = { ((x + y) + z) / 3.0f, idx };
            }
            std::sort(scratch.begin(), scratch.end(), [](const KeyIdx& a, const KeyIdx& b) { return a.key < b.key; });
            for (uint32_t i = 0; i < g.count; i++)
                hostOrder[g.begin + i] = scratch[i].idx;
            if (stats)
                stats->rangesSortedOnHost++, stats->primitivesSortedOnHost += g.count;
        }
        BUILD_TRY(hipMemcpy(dOrder[cur], hostOrder.data(), n * sizeof(uint32_t), hipMemcpyHostToDevice));
    }
    const size_t words = (size_t)n * kPrimWords;
    hipLaunchKernelGGL(gatherPrimsKernel, dim3((uint32_t)((words + 255) / 256)), dim3(256), 0, 0, dPrims, dOrder[cur], n, dPrimsOut);
    BUILD_TRY(hipGetLastError());
    BUILD_TRY(hipMemcpy(primsOut, dPrimsOut, words * 4, hipMemcpyDeviceToHost));
    BUILD_TRY(hipMemcpy(nodes.data(), dNodes, nodes.size() * sizeof(rt_node), hipMemcpyDeviceToHost));
    std::copy(nodes.begin(), nodes.end(), nodesOut);
    if (stats)
        stats->levels = static_cast<uint32_t>(levels.size());
    return true;
}

bool deviceSahBinBuild(const rt_primitive* in, uint32_t n, rt_node* nodesOut, rt_primitive* primsOut, uint32_t* nNodes,
    uint32_t* nLevels, uint32_t* nLeaves, DeviceBuildStats* stats, std::string& error)
{
    if (stats)
        *stats = DeviceBuildStats {};
    if (n <= 4) { // a single leaf (or the empty tree): nothing is split, the median builder's tree is this builder's
        return deviceMedianBuild(in, n, nodesOut, primsOut, nNodes, nLevels, nLeaves, stats, error);
    }
    // Nodes get provisional numbers in the order the levels create them; the recursion's numbers follow at the end.
    struct Prov {
        uint32_t begin, count, left, right, level; // left == 0: a leaf
    };
    std::vector<Prov> prov { { 0u, n, 0u, 0u, 0u } };
    std::vector<Seg> segs { { 0u, n, 0u, 1u } }, next;

    DeviceBuffers buf;
    uint32_t *dPrims, *dPrimsOut, *dOrder[2], *dHist, *dSplits, *dFlags, *dAny;
    float *dLo, *dHi, *dC, *dKeyf, *dCost;
    unsigned long long* dKeys[2];
    rt_node* dNodes;
    Seg* dSegs;
    const uint32_t nBlocks = (n + kSortTile - 1) / kSortTile;
    BUILD_TRY(buf.get(&dPrims, (size_t)n * kPrimWords));
    BUILD_TRY(buf.get(&dPrimsOut, (size_t)n * kPrimWords));
    BUILD_TRY(buf.get(&dOrder[0], n));
    BUILD_TRY(buf.get(&dOrder[1], n));
    BUILD_TRY(buf.get(&dKeys[0], n));
    BUILD_TRY(buf.get(&dKeys[1], n));
    BUILD_TRY(buf.get(&dKeyf, n));
    BUILD_TRY(buf.get(&dCost, n));
    BUILD_TRY(buf.get(&dLo, (size_t)3 * n));
    BUILD_TRY(buf.get(&dHi, (size_t)3 * n));
    BUILD_TRY(buf.get(&dC, (size_t)3 * n));
    BUILD_TRY(buf.get(&dHist, (size_t)256 * nBlocks));
    BUILD_TRY(buf.get(&dSplits, n));
    BUILD_TRY(buf.get(&dFlags, n));
    BUILD_TRY(buf.get(&dAny, 1));
    BUILD_TRY(buf.get(&dNodes, (size_t)2 * n)); // a tree over n primitives has at most 2n - 1 nodes
    BUILD_TRY(buf.get(&dSegs, n));              // and a level at most n ranges
    BUILD_TRY(hipMemcpy(dPrims, in, (size_t)n * sizeof(rt_primitive), hipMemcpyHostToDevice));
    const uint32_t grid = (n + 255) / 256;
    hipLaunchKernelGGL(primInfoKernel, dim3(grid), dim3(256), 0, 0, dPrims, n, dLo, dHi, dC);
    hipLaunchKernelGGL(iotaKernel, dim3(grid), dim3(256), 0, 0, dOrder[0], n);
    int cur = 0;
    std::vector<uint32_t> splits;
    uint32_t levelsDone = 0;
    for (;;) {
        const uint32_t nSegs = static_cast<uint32_t>(segs.size());
        uint32_t maxCount = 0, inners = 0, fresh = 0;
        for (const Seg& s : segs) {
            if (s.inner == 1u)
                maxCount = std::max(maxCount, s.count), inners++;
            if (s.inner != 2u)
                fresh++;
        }
        if (fresh == 0)
            break;
        levelsDone++;
        BUILD_TRY(hipMemcpyAsync(dSegs, segs.data(), nSegs * sizeof(Seg), hipMemcpyHostToDevice, 0));
        hipLaunchKernelGGL(levelKernel, dim3(nSegs), dim3(256), 0, 0, dSegs, dOrder[cur], n, dLo, dHi, dC, dNodes, dKeys[cur], dKeyf, 1);
        next.clear();
        if (inners == 0) { // the last level: boxes of the leaves only
            for (Seg& s : segs)
                s.inner = 2u;
            continue;
        }
        if (maxCount <= 64) { // one lane per range: an insertion sort on the bucket numbers is stable
            hipLaunchKernelGGL(smallSortKernel, dim3((nSegs + 255) / 256), dim3(256), 0, 0, dSegs, nSegs, dKeyf, dOrder[cur], dFlags, dAny);
        } else { // the bucket digit, then the range number: ranges tile the array in order, finished leaves included
            uint32_t segBits = 0;
            while ((1u << segBits) < nSegs)
                segBits++;
            const uint32_t passes = 1 + (segBits + 7) / 8;
            for (uint32_t p = 0; p < passes; p++) {
                const uint32_t shift = p == 0 ? 0u : 32u + 8u * (p - 1);
                hipLaunchKernelGGL(radixCountKernel, dim3(nBlocks), dim3(256), 0, 0, dKeys[cur], n, shift, dHist, nBlocks);
                hipLaunchKernelGGL(exclusiveScanKernel, dim3(1), dim3(1024), 0, 0, dHist, 256u * nBlocks);
                hipLaunchKernelGGL(radixScatterKernel, dim3(nBlocks), dim3(256), 0, 0, dKeys[cur], dOrder[cur], dKeys[cur ^ 1], dOrder[cur ^ 1], n,
                    shift, dHist, nBlocks);
                cur ^= 1;
            }
        }
        hipLaunchKernelGGL(sahSplitKernel, dim3(nSegs), dim3(256), 0, 0, dSegs, dOrder[cur], n, dLo, dHi, dCost, dSplits);
        BUILD_TRY(hipGetLastError());
        splits.resize(nSegs);
        BUILD_TRY(hipMemcpy(splits.data(), dSplits, nSegs * sizeof(uint32_t), hipMemcpyDeviceToHost));
        for (uint32_t i = 0; i < nSegs; i++) {
            const Seg& s = segs[i];
            if (s.inner != 1u) {
                next.push_back({ s.begin, s.count, s.node, 2u });
                continue;
            }
            const uint32_t split = splits[i];
            if (split < 1u || split + 2u > s.count) {
                error = "device build: a split position outside its range";
                return false;
            }
            const uint32_t left = static_cast<uint32_t>(prov.size()), level = prov[s.node].level + 1;
            prov[s.node].left = left, prov[s.node].right = left + 1;
            prov.push_back({ s.begin, split, 0u, 0u, level });
            prov.push_back({ s.begin + split, s.count - split, 0u, 0u, level });
            next.push_back({ s.begin, split, left, split > 4u ? 1u : 0u });
            next.push_back({ s.begin + split, s.count - split, left + 1, s.count - split > 4u ? 1u : 0u });
        }
        segs.swap(next);
    }
    const size_t words = (size_t)n * kPrimWords;
    hipLaunchKernelGGL(gatherPrimsKernel, dim3((uint32_t)((words + 255) / 256)), dim3(256), 0, 0, dPrims, dOrder[cur], n, dPrimsOut);
    BUILD_TRY(hipGetLastError());
    BUILD_TRY(hipMemcpy(primsOut, dPrimsOut, words * 4, hipMemcpyDeviceToHost));
    std::vector<rt_node> boxes(prov.size());
    BUILD_TRY(hipMemcpy(boxes.data(), dNodes, prov.size() * sizeof(rt_node), hipMemcpyDeviceToHost));

    // ---- the recursion's numbering (src/bvh.cpp:424-493): a node's children take the next two indices when the
    // node is reached, depth first, left subtree before the right one; index 1 is the unused node (src/bvh.cpp:88-89)
    std::vector<rt_node> nodes(2);
    std::memset(nodes.data(), 0, 2 * sizeof(rt_node));
    nodes[1].upper[0] = nodes[1].upper[1] = nodes[1].upper[2] = 1.0f;
    struct Visit {
        uint32_t prov, index;
    };
    std::vector<Visit> work { { 0u, 0u } };
    uint32_t leaves = 0, deepest = 0;
    while (!work.empty()) {
        const Visit v = work.back();
        work.pop_back();
        const Prov& pn = prov[v.prov];
        rt_node nd = boxes[v.prov];
        deepest = std::max(deepest, pn.level);
        if (pn.left == 0u) {
            nd.data[0] = pn.begin | RT_LEAF_BIT;
            nd.data[1] = pn.count;
            leaves++;
        } else {
            const uint32_t left = static_cast<uint32_t>(nodes.size());
            nodes.emplace_back();
            nodes.emplace_back();
            nd.data[0] = left;
            nd.data[1] = left + 1;
            work.push_back({ pn.right, left + 1 });
            work.push_back({ pn.left, left });
        }
        nodes[v.index] = nd;
    }
    std::copy(nodes.begin(), nodes.end(), nodesOut);
    *nNodes = static_cast<uint32_t>(nodes.size());
    *nLevels = deepest + 1;
    *nLeaves = leaves;
    if (stats)
        stats->levels = levelsDone;
    return true;
}

} // namespace rt
