// treeopt.hpp -- (experiment, not part of the library) a traversal hierarchy as an explicit binary tree, and its
// re-optimisation by removing and re-inserting subtrees (after Bittner, Hapala, Havran: "Fast
// insertion-based optimization of bounding volume hierarchies", 2013).  Host code only.
//
// A greedy top-down build fixes every split before it knows what the subtrees will look like; afterwards
// the surface-area cost of the finished tree -- the expected number of boxes a random ray enters, the sum
// of the inner boxes' areas over the root's -- can still be lowered by local surgery: take an inner node
// whose box is large for what it holds, cut it out, and put its two subtrees back one by one wherever
// they raise the areas of the boxes above them least (a branch-and-bound search from the root).  Which
// primitive a ray hits does not depend on the hierarchy, so results are unchanged; only the number of
// traversal steps per ray moves.  The depth of the tree is kept within `maxDepth` nodes on a path (the
// traversal's LDS stack holds one entry per level).
#pragma once

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <queue>
#include <vector>

namespace rt {

struct OptNode {
    float lo[3], hi[3];
    int32_t parent = -1, child[2] = { -1, -1 }; // child[0] < 0: a leaf
    uint32_t ref = 0;                           // leaf: the reference the pair record carries
    uint32_t height = 1;                        // nodes on the longest path from here down to a leaf
};

struct TreeOptimiser {
    std::vector<OptNode>& N;
    int32_t& root;
    uint32_t maxDepth;

    TreeOptimiser(std::vector<OptNode>& nodes, int32_t& rootIndex, uint32_t depthLimit)
        : N(nodes), root(rootIndex), maxDepth(depthLimit)
    {
    }

    static float area(const float* lo, const float* hi)
    {
        const float x = hi[0] - lo[0], y = hi[1] - lo[1], z = hi[2] - lo[2];
        return 2.0f * (x * y + x * z + y * z);
    }
    float area(int32_t i) const { return area(N[i].lo, N[i].hi); }
    float mergedArea(int32_t a, const OptNode& x) const
    {
        float lo[3], hi[3];
        for (int k = 0; k < 3; k++)
            lo[k] = std::min(N[a].lo[k], x.lo[k]), hi[k] = std::max(N[a].hi[k], x.hi[k]);
        return area(lo, hi);
    }
    bool leaf(int32_t i) const { return N[i].child[0] < 0; }
    uint32_t depthOf(int32_t i) const // nodes on the path from the root to i, both counted
    {
        uint32_t d = 1;
        while (N[i].parent >= 0)
            i = N[i].parent, d++;
        return d;
    }
    // box and height of an inner node from its children, then of everything above it
    void refitFrom(int32_t i)
    {
        for (; i >= 0; i = N[i].parent) {
            const OptNode &a = N[N[i].child[0]], &b = N[N[i].child[1]];
            for (int k = 0; k < 3; k++)
                N[i].lo[k] = std::min(a.lo[k], b.lo[k]), N[i].hi[k] = std::max(a.hi[k], b.hi[k]);
            N[i].height = 1u + std::max(a.height, b.height);
        }
    }
    double cost() const // sum of the inner boxes' areas over the root's
    {
        double s = 0.0;
        for (size_t i = 0; i < N.size(); i++)
            if (N[i].child[0] >= 0 && (N[i].parent >= 0 || (int32_t)i == root))
                s += area((int32_t)i);
        return s / area(root);
    }

    // where the subtree x costs least: the node it should share a new parent with
    int32_t bestSibling(int32_t x) const
    {
        struct Entry {
            float induced;
            int32_t node;
            uint32_t depth;
            bool operator<(const Entry& o) const { return induced > o.induced; } // smallest induced cost first
        };
        std::priority_queue<Entry> q;
        q.push({ 0.0f, root, 1u });
        const float ax = area(x);
        float best = std::numeric_limits<float>::infinity();
        int32_t bestNode = root;
        while (!q.empty()) {
            const Entry e = q.top();
            q.pop();
            if (e.induced + ax >= best)
                break; // nothing further down the queue can do better
            const float direct = mergedArea(e.node, N[x]);
            // the new parent takes e.node's place: x ends up one level below it
            const bool fits = e.depth + 1u + N[x].height - 1u <= maxDepth && e.depth + N[e.node].height <= maxDepth;
            if (fits && e.induced + direct < best)
                best = e.induced + direct, bestNode = e.node;
            if (!leaf(e.node)) {
                const float below = e.induced + direct - area(e.node); // what every box from here up has grown by
                if (below + ax < best) {
                    q.push({ below, N[e.node].child[0], e.depth + 1u });
                    q.push({ below, N[e.node].child[1], e.depth + 1u });
                }
            }
        }
        return bestNode;
    }

    // x (a detached subtree) and `at` become the children of the free node `shell`, which takes `at`'s place
    void insertBeside(int32_t x, int32_t at, int32_t shell)
    {
        const int32_t p = N[at].parent;
        N[shell].parent = p;
        N[shell].child[0] = at;
        N[shell].child[1] = x;
        N[at].parent = shell;
        N[x].parent = shell;
        if (p >= 0)
            N[p].child[N[p].child[0] == at ? 0 : 1] = shell;
        else
            root = shell;
        refitFrom(shell);
    }

    // One node cut out, its two subtrees put back.  False if the node cannot be processed (too near the root).
    bool reinsert(int32_t n)
    {
        if (leaf(n) || n == root)
            return false;
        const int32_t p = N[n].parent;
        if (p == root || p < 0)
            return false;
        const int32_t g = N[p].parent;
        const int32_t s = N[p].child[N[p].child[0] == n ? 1 : 0];
        int32_t a = N[n].child[0], b = N[n].child[1];
        // the sibling takes the parent's place
        N[g].child[N[g].child[0] == p ? 0 : 1] = s;
        N[s].parent = g;
        refitFrom(g);
        N[a].parent = N[b].parent = -1;
        if (area(a) < area(b))
            std::swap(a, b); // the larger one first
        insertBeside(a, bestSibling(a), n);
        insertBeside(b, bestSibling(b), p);
        return true;
    }

    // Passes over the inner nodes that look worst (large box, small or lopsided children), `fraction` of them per
    // pass, until a pass gains less than `minGain` of the cost or `maxPasses` are done.  Returns passes done.
    int run(int maxPasses, double fraction, double minGain)
    {
        std::vector<std::pair<float, int32_t>> cand;
        double before = cost();
        int pass = 0;
        for (; pass < maxPasses; pass++) {
            cand.clear();
            for (size_t i = 0; i < N.size(); i++) {
                const OptNode& nd = N[i];
                if (nd.child[0] < 0 || (int32_t)i == root || nd.parent < 0 || nd.parent == root)
                    continue;
                const float an = area((int32_t)i), a0 = area(nd.child[0]), a1 = area(nd.child[1]);
                const float mn = std::min(a0, a1), sum = a0 + a1;
                if (!(mn > 0.0f) || !(sum > 0.0f))
                    continue;
                cand.push_back({ an * (an / mn) * (an / (0.5f * sum)), (int32_t)i });
            }
            if (cand.empty())
                break;
            const size_t take = std::max<size_t>(1, (size_t)(cand.size() * fraction));
            std::partial_sort(cand.begin(), cand.begin() + std::min(take, cand.size()), cand.end(),
                [](const std::pair<float, int32_t>& x, const std::pair<float, int32_t>& y) { return x.first > y.first || (x.first == y.first && x.second < y.second); });
            for (size_t k = 0; k < take && k < cand.size(); k++) {
                const int32_t n = cand[k].second;
                // (an earlier reinsertion of this pass may have made it a child of the root, or moved it: it is still an inner node)
                reinsert(n);
            }
            const double after = cost();
            const bool done = before - after < minGain * before;
            before = after;
            if (done) {
                pass++;
                break;
            }
        }
        return pass;
    }
};

} // namespace rt
