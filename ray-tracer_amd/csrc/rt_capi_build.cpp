// rt_capi_build.cpp -- host-side builders: the reference's hierarchy (rt_bvh_build, rt_bvh_build_device) and the
// library's own over the caller's leaves (buildOwnTree, used by the scene upload).

#include "rt_ctx.hpp"

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

} // namespace

namespace rt {

// Builds the hierarchy over `leaves` into sibling-pair records in breadth-first order (the records
// every ray starts with sit next to each other).  False if it came out deeper than a traversal stack.
bool buildOwnTree(const std::vector<TreeLeaf>& leaves, const QuantGrid& grid, std::vector<uint4>& pairs, uint32_t& rootRef,
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

} // namespace rt

namespace {
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

extern "C" {

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

} // extern "C"
