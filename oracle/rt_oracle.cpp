// rt_oracle.cpp -- TEST INFRASTRUCTURE, NOT PRODUCT CODE (see rt_oracle.h).
//
// Plain C++ restatement of the reference's per-pixel rendering path, written
// recursively and in the same operation order as the reference so that, built
// with -ffp-contract=off, every float is rounded where the reference rounds
// it.  Vector helpers follow glm 0.9.9.9 (framework/third_party/glm):
// dot = (x+y)+z, cross as in detail/func_geometric.inl:74-77,
// normalize = v * (1/sqrt(dot)), vec/scalar = per-component division.
//
// Every function cites the reference file:line it follows.  The five
// intersection primitives exist in the reference only as a prebuilt archive
// (src/intersect.cpp:13-53 is all stubs); they are restated from their
// documented formulas and pinned by the reference's own pyramid test values
// (tests/bvh_tests.cpp:903-1008).

#include "rt_oracle.h"

#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#ifdef _OPENMP
#include <omp.h>
#endif

namespace {

struct V2 {
    float x, y;
};
struct V3 {
    float x, y, z;
    float operator[](int i) const { return i == 0 ? x : (i == 1 ? y : z); }
};

inline V3 v3(const float* p) { return { p[0], p[1], p[2] }; }
inline V3 v3(float s) { return { s, s, s }; }
inline V3 operator+(V3 a, V3 b) { return { a.x + b.x, a.y + b.y, a.z + b.z }; }
inline V3 operator-(V3 a, V3 b) { return { a.x - b.x, a.y - b.y, a.z - b.z }; }
inline V3 operator-(V3 a) { return { -a.x, -a.y, -a.z }; }
inline V3 operator*(V3 a, V3 b) { return { a.x * b.x, a.y * b.y, a.z * b.z }; }
inline V3 operator*(V3 a, float s) { return { a.x * s, a.y * s, a.z * s }; }
inline V3 operator*(float s, V3 a) { return { s * a.x, s * a.y, s * a.z }; }
inline V3 operator/(V3 a, float s) { return { a.x / s, a.y / s, a.z / s }; }
inline V3& operator+=(V3& a, V3 b)
{
    a = a + b;
    return a;
}
inline bool operator==(V3 a, V3 b) { return a.x == b.x && a.y == b.y && a.z == b.z; }
inline V2 operator+(V2 a, V2 b) { return { a.x + b.x, a.y + b.y }; }
inline V2 operator*(V2 a, float s) { return { a.x * s, a.y * s }; }

inline float dot(V3 a, V3 b)
{
    const float tx = a.x * b.x, ty = a.y * b.y, tz = a.z * b.z;
    return tx + ty + tz;
}
inline V3 cross(V3 x, V3 y)
{
    return { x.y * y.z - y.y * x.z, x.z * y.x - y.z * x.x, x.x * y.y - y.x * x.y };
}
inline float length(V3 v) { return std::sqrt(dot(v, v)); }
inline V3 normalize(V3 v) { return v * (1.0f / std::sqrt(dot(v, v))); }
inline float gmin(float x, float y) { return (y < x) ? y : x; } // glm::min
inline float gmax(float x, float y) { return (x < y) ? y : x; } // glm::max
inline V3 clamp01(V3 v)
{
    return { gmin(gmax(v.x, 0.0f), 1.0f), gmin(gmax(v.y, 0.0f), 1.0f), gmin(gmax(v.z, 0.0f), 1.0f) };
}
inline V3 mix(V3 x, V3 y, float a) { return x * (1.0f - a) + y * a; } // glm::mix
inline bool allEpsilonEqual(V3 a, V3 b, float eps)
{
    return std::fabs(a.x - b.x) < eps && std::fabs(a.y - b.y) < eps && std::fabs(a.z - b.z) < eps;
}

// src/sampler.h:13-46
struct Sampler {
    uint32_t state;
    uint32_t hash()
    {
        state = state * 747796405u + 2891336453u;
        uint32_t v = state;
        v ^= v >> ((v >> 28u) + 4u);
        v *= 277803737u;
        v ^= v >> 22u;
        return v;
    }
    float next_1d() { return static_cast<float>(hash()) / 4294967295.f; }
    V2 next_2d()
    {
        const float a = next_1d();
        const float b = next_1d();
        return { a, b };
    }
};

struct Ray {
    V3 origin { 0.0f, 0.0f, 0.0f };
    V3 direction { 0.0f, 0.0f, -1.0f };
    float t = FLT_MAX;
};

struct HitInfo {
    V3 normal { 0, 0, 0 };
    V3 bary { 0, 0, 0 };
    V2 texCoord { 0, 0 };
    orc_material material { { 0, 0, 0 }, { 0, 0, 0 }, 1.0f, 1.0f, -1 };
    int prim = -1;
    int matSlot = -1;
};

constexpr uint32_t LeafBit = 1u << 31;

} // namespace

struct orc_scene {
    std::vector<orc_node> nodes;
    std::vector<orc_primitive> prims;
    std::vector<orc_material> materials;
    std::vector<std::vector<float>> imagePixels;
    std::vector<orc_image> images;
    int envFaces[6];
    std::vector<orc_sphere> spheres;
    std::vector<orc_light> lights;
};

namespace {

struct State {
    const orc_scene& scene;
    const orc_features& f;
    Sampler sampler;
    orc_stats stats {};
    int policy = 0;
    FILE* dump = nullptr; // debugging aid: every intersect call of one pixel (ORC_DUMP_PIXEL / ORC_DUMP_FILE)
};

// ---------------------------------------------------------------- intersect primitives
// trianglePlane / intersectRayWithPlane / pointInTriangle / intersectRayWithTriangle
// (declared src/intersect.h:8-12; bodies only in the prebuilt archive).
// Plane through the triangle: n = normalize(cross(v1-v0, v2-v0)), D = dot(v0, n).
// t = (D - dot(o,n)) / dot(d,n); accepted iff 0 <= t <= ray.t (NaN rejects).
// Inside test: three inclusive half-space tests against cross(edge, n).
bool intersectRayWithTriangle(V3 v0, V3 v1, V3 v2, Ray& ray)
{
    const V3 c = cross(v1 - v0, v2 - v0);
    const V3 n = c * (1.0f / std::sqrt(dot(c, c)));
    const float D = dot(v0, n);
    const float t = (D - dot(ray.origin, n)) / dot(ray.direction, n);
    if (!(t >= 0.0f && t <= ray.t))
        return false;
    const V3 p = ray.origin + ray.direction * t;
    if (!(dot(cross(v2 - v0, n), p - v0) >= 0.0f))
        return false;
    if (!(dot(cross(v0 - v1, n), p - v1) >= 0.0f))
        return false;
    if (!(dot(cross(v1 - v2, n), p - v2) >= 0.0f))
        return false;
    ray.t = t;
    return true;
}

// intersectRayWithShape(AxisAlignedBox) (src/intersect.h:16).  Per-axis slabs by
// true division; a zero direction component ignores its slab; an origin inside
// the box reports the exit distance.  `entry` receives max(tIn, 0) before the
// inside rewrite (used only by the culling policy, not by the reference).
bool intersectRayWithAABB(const float* lower, const float* upper, Ray& ray, float* entry = nullptr)
{
    float tIn = -FLT_MAX, tOut = FLT_MAX;
    const float o[3] = { ray.origin.x, ray.origin.y, ray.origin.z };
    const float d[3] = { ray.direction.x, ray.direction.y, ray.direction.z };
    for (int a = 0; a < 3; a++) {
        float tLo, tHi;
        if (d[a] == 0.0f) {
            tHi = FLT_MAX;
            tLo = FLT_MIN;
        } else {
            tLo = (lower[a] - o[a]) / d[a];
            tHi = (upper[a] - o[a]) / d[a];
        }
        const float mn = std::min(tLo, tHi);
        const float mx = std::max(tLo, tHi);
        tIn = (a == 0) ? mn : std::max(tIn, mn);
        tOut = (a == 0) ? mx : std::min(tOut, mx);
    }
    if (entry)
        *entry = tIn < 0.0f ? 0.0f : tIn;
    if (tIn < 0.0f) {
        if (tOut <= 0.0f)
            return false;
        tIn = tOut;
        tOut = FLT_MAX;
    }
    if (tIn > tOut || tIn < 0.0f || tIn > ray.t)
        return false;
    ray.t = tIn;
    return true;
}

// intersectRayWithShape(Sphere) (src/intersect.h:14): unit-direction quadratic,
// nearest non-negative root, strict t < ray.t.
bool intersectRayWithSphere(const orc_sphere& s, Ray& ray, HitInfo& hitInfo, int matSlot)
{
    const V3 oc = ray.origin - v3(s.center);
    const float B = 2.0f * dot(ray.direction, oc);
    const float C = dot(oc, oc) - s.radius * s.radius;
    const float disc = B * B - 4.0f * C;
    const float sq = std::sqrt(disc);
    float t0 = (-B - sq) / 2.0f;
    float t1 = (-B + sq) / 2.0f;
    if (t0 < 0.0f)
        t0 = FLT_MAX;
    if (t1 < 0.0f)
        t1 = FLT_MAX;
    const float t = std::min(t0, t1);
    if (!(t >= 0.0f && t < ray.t))
        return false;
    ray.t = t;
    const V3 p = ray.origin + ray.direction * t;
    hitInfo.material = s.material;
    hitInfo.normal = normalize(p - v3(s.center));
    hitInfo.prim = -1;
    hitInfo.matSlot = matSlot;
    return true;
}

// ---------------------------------------------------------------- interpolation
// src/interpolate.cpp:13-24
V3 computeBarycentricCoord(V3 v0, V3 v1, V3 v2, V3 p)
{
    const V3 n = cross(v1 - v0, v2 - v0);
    const V3 na = cross(v2 - v1, p - v1);
    const V3 nb = cross(v0 - v2, p - v2);
    const float alpha = dot(n, na) / dot(n, n);
    const float beta = dot(n, nb) / dot(n, n);
    const float gamma = 1 - alpha - beta;
    return { alpha, beta, gamma };
}
// src/interpolate.cpp:34-40 (no renormalisation)
V3 interpolateNormal(V3 n0, V3 n1, V3 n2, V3 bc) { return bc.x * n0 + bc.y * n1 + bc.z * n2; }
// src/interpolate.cpp:50-55
V2 interpolateTexCoord(V2 t0, V2 t1, V2 t2, V3 bc) { return t0 * bc.x + t1 * bc.y + t2 * bc.z; }

// ---------------------------------------------------------------- textures
// The reference indexes `image.pixels[float]` with no wrap or clamp; coordinates
// outside [0,1] are undefined behaviour there.  Both this restatement and the
// device path clamp the final texel index into the image.
V3 texel(const orc_image& im, float index)
{
    long i = static_cast<long>(index);
    const long last = static_cast<long>(im.width) * im.height - 1;
    if (!(index == index) || i < 0)
        i = 0;
    if (i > last)
        i = last;
    return v3(im.pixels + 3 * i);
}
// src/texture.cpp:12-35
V3 sampleTextureNearest(const orc_image& image, V2 texCoord)
{
    float i = texCoord.x * image.width;
    float j = (1 - texCoord.y) * image.height;
    if (i == image.width)
        i--;
    if (j == image.height)
        j--;
    const float downJ = std::floor(j);
    const float downI = std::floor(i);
    const float index = downJ * image.width + downI;
    return texel(image, index);
}
// src/texture.cpp:44-95; abs() on floats is the float overload (the reference's
// own bilinear8 test, tests/texture_tests.cpp:130-134, only passes that way).
V3 sampleTextureBilinear(const orc_image& image, V2 texCoord)
{
    float x = texCoord.x * image.width;
    float y = (1 - texCoord.y) * image.height;
    if ((x == 0 && y == 0) || (x == image.width && y == 0) || (x == 0 && y == image.height)
        || (x == image.width && y == image.height) || x <= 0.5f || x >= image.width - 0.5f || y <= 0.5f
        || y >= image.height - 0.5f) {
        if (x >= image.width)
            x--;
        if (y >= image.height)
            y--;
        const float index = std::floor(y) * image.width + std::floor(x);
        return texel(image, index);
    }
    const float i1 = std::round(x) - 0.5f;
    const float i2 = std::round(x) + 0.5f;
    const float j1 = std::round(y) - 0.5f;
    const float j2 = std::round(y) + 0.5f;
    const float distI1 = std::fabs(x - i1);
    const float distI2 = std::fabs(i2 - x);
    const V3 pix1 = texel(image, std::floor(j1) * image.width + std::floor(i1));
    const V3 pix2 = texel(image, std::floor(j1) * image.width + std::floor(i2));
    const V3 interpolatedX1 = distI2 * pix1 + distI1 * pix2;
    const V3 pix3 = texel(image, std::floor(j2) * image.width + std::floor(i1));
    const V3 pix4 = texel(image, std::floor(j2) * image.width + std::floor(i2));
    const V3 interpolatedX2 = distI2 * pix3 + distI1 * pix4;
    const float distJ1 = std::fabs(y - j1);
    const float distJ2 = std::fabs(j2 - y);
    return distJ2 * interpolatedX1 + distJ1 * interpolatedX2;
}

// ---------------------------------------------------------------- BVH build
struct AABB {
    V3 lower, upper;
};
// src/bvh.cpp:117-133
AABB computePrimitiveAABB(const orc_primitive& p)
{
    float mn[3], mx[3];
    for (int c = 0; c < 3; c++) {
        const float a = p.v0.position[c], b = p.v1.position[c], d = p.v2.position[c];
        mn[c] = std::min(a, std::min(b, d));
        mx[c] = std::max(a, std::max(b, d));
    }
    return { v3(mn), v3(mx) };
}
// src/bvh.cpp:140-162
AABB computeSpanAABB(const orc_primitive* prims, size_t n)
{
    if (n == 0)
        return { v3(0.0f), v3(1.0f) };
    AABB box = computePrimitiveAABB(prims[0]);
    float mn[3] = { box.lower.x, box.lower.y, box.lower.z }, mx[3] = { box.upper.x, box.upper.y, box.upper.z };
    for (size_t i = 1; i < n; i++) {
        const AABB b = computePrimitiveAABB(prims[i]);
        for (int c = 0; c < 3; c++) {
            mn[c] = std::min(mn[c], b.lower[c]);
            mx[c] = std::max(mx[c], b.upper[c]);
        }
    }
    return { v3(mn), v3(mx) };
}
// src/bvh.cpp:169-175
V3 computePrimitiveCentroid(const orc_primitive& p)
{
    return (v3(p.v0.position) + v3(p.v1.position) + v3(p.v2.position)) / 3.0f;
}
// src/bvh.cpp:183-196
uint32_t computeAABBLongestAxis(const AABB& b)
{
    const float xl = b.upper.x - b.lower.x, yl = b.upper.y - b.lower.y, zl = b.upper.z - b.lower.z;
    if (xl >= yl && xl >= zl)
        return 0;
    else if (yl >= zl)
        return 1;
    return 2;
}
// src/bvh.cpp:208-223; std::sort is what std::ranges::sort runs in libstdc++.
size_t splitPrimitivesByMedian(uint32_t axis, orc_primitive* prims, size_t n)
{
    std::sort(prims, prims + n, [axis](const orc_primitive& a, const orc_primitive& b) {
        return computePrimitiveCentroid(a)[axis] < computePrimitiveCentroid(b)[axis];
    });
    return (n + 1) / 2;
}

// src/extra.cpp:475-483
float computeSurfaceArea(const AABB& aabb)
{
    const float x = aabb.upper.x - aabb.lower.x;
    const float y = aabb.upper.y - aabb.lower.y;
    const float z = aabb.upper.z - aabb.lower.z;
    return 2 * (x * y + x * z + y * z);
}
AABB mergeAABBs(const AABB& a, const AABB& b)
{
    return { { std::min(a.lower.x, b.lower.x), std::min(a.lower.y, b.lower.y), std::min(a.lower.z, b.lower.z) },
        { std::max(a.upper.x, b.upper.x), std::max(a.upper.y, b.upper.y), std::max(a.upper.z, b.upper.z) } };
}
// splitPrimitivesBySAHBin, src/extra.cpp:493-543, statement by statement: eleven buckets of
// ten bins' width along the axis, primitives re-ordered bucket by bucket (order inside a
// bucket kept), then the cost of EVERY split position, where the box on the right lags one
// primitive behind its count, and the first cheapest position wins.  A node without extent
// along its longest axis divides by zero there (undefined); everything goes to bucket 0 here.
size_t splitPrimitivesBySAHBin(const AABB& aabb, uint32_t axis, orc_primitive* primitives, size_t n)
{
    const size_t numBins = 10;
    std::vector<std::vector<orc_primitive>> bins(numBins + 1);
    const float binSize = (aabb.upper[(int)axis] - aabb.lower[(int)axis]) / numBins;
    for (size_t k = 0; k < n; k++) {
        const float q = std::floor((computePrimitiveCentroid(primitives[k])[(int)axis] - aabb.lower[(int)axis]) / binSize);
        size_t binIndex = 0;
        if (q >= 0.0f)
            binIndex = q >= float(numBins) ? numBins : static_cast<size_t>(q);
        bins[binIndex].push_back(primitives[k]);
    }
    size_t index = 0;
    for (auto& bin : bins)
        for (orc_primitive& primitive : bin)
            primitives[index++] = primitive;

    std::vector<float> costs(n - 2);
    AABB leftAABB = computeSpanAABB(primitives, 1);
    for (size_t i = 1; i < n - 1; i++) {
        const float leftArea = computeSurfaceArea(leftAABB);
        costs[i - 1] = (float)i * leftArea;
        leftAABB = mergeAABBs(leftAABB, computePrimitiveAABB(primitives[i]));
    }
    AABB rightAABB = computeSpanAABB(primitives + (n - 1), 1);
    for (size_t i = n - 2; i > 0; i--) {
        const float rightArea = computeSurfaceArea(rightAABB);
        costs[i - 1] += (float)(n - i) * rightArea;
        rightAABB = mergeAABBs(rightAABB, computePrimitiveAABB(primitives[i]));
    }
    size_t optimalSplit = 0;
    for (size_t i = 0; i < costs.size(); i++)
        if (costs[i] < costs[optimalSplit])
            optimalSplit = i;
    return optimalSplit + 1;
}

struct Builder {
    std::vector<orc_node> nodes;
    std::vector<orc_primitive> out;
    bool sahBinning = false;
    // src/bvh.cpp:424-493
    void buildRecursive(orc_primitive* prims, size_t n, uint32_t nodeIndex)
    {
        const AABB aabb = computeSpanAABB(prims, n);
        orc_node node;
        node.lower[0] = aabb.lower.x, node.lower[1] = aabb.lower.y, node.lower[2] = aabb.lower.z;
        node.upper[0] = aabb.upper.x, node.upper[1] = aabb.upper.y, node.upper[2] = aabb.upper.z;
        if (n <= 4) { // BVH::LeafSize, src/bvh.h:65; leaf data src/bvh.cpp:369-387
            node.data[0] = static_cast<uint32_t>(out.size()) | LeafBit;
            node.data[1] = static_cast<uint32_t>(n);
            out.insert(out.end(), prims, prims + n);
            nodes[nodeIndex] = node;
            return;
        }
        const uint32_t axis = computeAABBLongestAxis(aabb);
        const size_t split = sahBinning ? splitPrimitivesBySAHBin(aabb, axis, prims, n) : splitPrimitivesByMedian(axis, prims, n);
        const uint32_t left = static_cast<uint32_t>(nodes.size());
        nodes.emplace_back();
        const uint32_t right = static_cast<uint32_t>(nodes.size());
        nodes.emplace_back();
        node.data[0] = left;
        node.data[1] = right;
        nodes[nodeIndex] = node;
        buildRecursive(prims, split, left);
        buildRecursive(prims + split, n - split, right);
    }
};

// ---------------------------------------------------------------- traversal
// src/bvh.cpp:18-48
void updateHitInfo(State& st, const orc_primitive& prim, int primIndex, const Ray& ray, HitInfo& hitInfo)
{
    const V3 p0 = v3(prim.v0.position), p1 = v3(prim.v1.position), p2 = v3(prim.v2.position);
    const V3 n = normalize(cross(p1 - p0, p2 - p0));
    const V3 p = ray.origin + ray.t * ray.direction;
    hitInfo.material = st.scene.materials[prim.meshID];
    hitInfo.matSlot = static_cast<int>(prim.meshID);
    hitInfo.prim = primIndex;
    hitInfo.normal = n;
    hitInfo.bary = computeBarycentricCoord(p0, p1, p2, p);
    if (st.f.enableNormalInterp)
        hitInfo.normal = interpolateNormal(v3(prim.v0.normal), v3(prim.v1.normal), v3(prim.v2.normal), hitInfo.bary);
    if (st.f.enableTextureMapping)
        hitInfo.texCoord = interpolateTexCoord({ prim.v0.texCoord[0], prim.v0.texCoord[1] },
            { prim.v1.texCoord[0], prim.v1.texCoord[1] }, { prim.v2.texCoord[0], prim.v2.texCoord[1] }, hitInfo.bary);
    if (dot(ray.direction, n) > 0.0f)
        hitInfo.normal = -hitInfo.normal;
}

// src/bvh.cpp:228-239
bool intersectRayWithPrimitive(State& st, int primIndex, Ray& ray, HitInfo& hitInfo)
{
    const orc_primitive& prim = st.scene.prims[primIndex];
    st.stats.tri_tests++;
    if (intersectRayWithTriangle(v3(prim.v0.position), v3(prim.v1.position), v3(prim.v2.position), ray)) {
        updateHitInfo(st, prim, primIndex, ray, hitInfo);
        return true;
    }
    return false;
}

// src/bvh.cpp:245-316
bool intersectRayWithBVHWhenEnabledAccel(State& st, Ray& ray, HitInfo& hitInfo)
{
    const std::vector<orc_node>& nodes = st.scene.nodes;
    Ray rayThisBox = ray;
    rayThisBox.t = FLT_MAX;
    st.stats.box_tests++;
    if (!intersectRayWithAABB(nodes[0].lower, nodes[0].upper, rayThisBox))
        return false;
    bool isHit = false;
    std::vector<uint32_t> nodeStack { 0u };
    while (!nodeStack.empty()) {
        const uint32_t nodeIndex = nodeStack.back();
        nodeStack.pop_back();
        const orc_node& node = nodes[nodeIndex];
        if (node.data[0] & LeafBit) {
            const uint32_t off = node.data[0] & ~LeafBit;
            for (uint32_t pi = off; pi < off + node.data[1]; pi++)
                isHit |= intersectRayWithPrimitive(st, static_cast<int>(pi), ray, hitInfo);
            continue;
        }
        st.stats.inner_nodes++;
        st.stats.box_tests += 2;
        Ray rayLeftBox = ray, rayRightBox = ray;
        rayLeftBox.t = FLT_MAX;
        rayRightBox.t = FLT_MAX;
        const uint32_t l = node.data[0], r = node.data[1];
        float entryL = 0.0f, entryR = 0.0f;
        bool hitL = intersectRayWithAABB(nodes[l].lower, nodes[l].upper, rayLeftBox, &entryL);
        bool hitR = intersectRayWithAABB(nodes[r].lower, nodes[r].upper, rayRightBox, &entryR);
        if (st.policy == 1) {
            // Device policy: a child whose entry distance lies beyond the closest
            // accepted hit cannot hold a closer one; ties (==) are still visited.
            hitL = hitL && !(entryL > ray.t);
            hitR = hitR && !(entryR > ray.t);
        }
        if (hitL && hitR) {
            if (rayLeftBox.t < rayRightBox.t) {
                nodeStack.push_back(r);
                nodeStack.push_back(l);
            } else {
                nodeStack.push_back(l);
                nodeStack.push_back(r);
            }
        } else if (hitL) {
            nodeStack.push_back(l);
        } else if (hitR) {
            nodeStack.push_back(r);
        }
    }
    return isHit;
}

// src/bvh.cpp:337-358 == BVHInterface::intersect (src/bvh.h:67-70)
bool intersect(State& st, Ray& ray, HitInfo& hitInfo)
{
    st.stats.rays++;
    const Ray in = ray;
    bool isHit = false;
    if (st.f.enableAccelStructure) {
        isHit |= intersectRayWithBVHWhenEnabledAccel(st, ray, hitInfo);
    } else {
        for (size_t i = 0; i < st.scene.prims.size(); i++)
            isHit |= intersectRayWithPrimitive(st, static_cast<int>(i), ray, hitInfo);
    }
    for (size_t i = 0; i < st.scene.spheres.size(); i++)
        isHit |= intersectRayWithSphere(st.scene.spheres[i], ray, hitInfo, static_cast<int>(st.scene.materials.size() + i));
    if (st.dump) { // origin, direction, t before, t after as bit patterns; hit flag; primitive
        auto bits = [](float f) { uint32_t u; std::memcpy(&u, &f, 4); return u; };
        std::fprintf(st.dump, "%08x %08x %08x %08x %08x %08x %08x %08x %d %d\n", bits(in.origin.x), bits(in.origin.y), bits(in.origin.z),
            bits(in.direction.x), bits(in.direction.y), bits(in.direction.z), bits(in.t), bits(ray.t), isHit ? 1 : 0, isHit ? hitInfo.prim : -1);
    }
    return isHit;
}

// ---------------------------------------------------------------- shading
// src/shading.cpp:13-24
V3 sampleMaterialKd(State& st, const HitInfo& hitInfo)
{
    if (st.f.enableTextureMapping && hitInfo.material.kdTexture >= 0) {
        const orc_image& im = st.scene.images[hitInfo.material.kdTexture];
        if (st.f.enableBilinearTextureFiltering)
            return sampleTextureBilinear(im, hitInfo.texCoord);
        return sampleTextureNearest(im, hitInfo.texCoord);
    }
    return v3(hitInfo.material.kd);
}
// src/shading.cpp:60-74
V3 computeLambertianModel(State& st, V3, V3 lightDirection, V3 lightColor, const HitInfo& hitInfo)
{
    const float d = dot(hitInfo.normal, lightDirection);
    if (d < 0)
        return v3(0.0f);
    const V3 materialKd = sampleMaterialKd(st, hitInfo);
    return lightColor * materialKd * d;
}
// src/shading.cpp:90-111; glm::reflect(I,N) = I - N*dot(N,I)*2
V3 computePhongModel(State& st, V3 cameraDirection, V3 lightDirection, V3 lightColor, const HitInfo& hitInfo)
{
    const V3 diffuse = computeLambertianModel(st, cameraDirection, lightDirection, lightColor, hitInfo);
    const V3 reflected = lightDirection - hitInfo.normal * dot(hitInfo.normal, lightDirection) * 2.0f;
    const float d = dot(reflected, cameraDirection);
    if (d < 0)
        return diffuse;
    const V3 specular = lightColor * v3(hitInfo.material.ks) * std::pow(d, hitInfo.material.shininess);
    return diffuse + specular;
}
// src/shading.cpp:127-147
V3 computeBlinnPhongModel(State& st, V3 cameraDirection, V3 lightDirection, V3 lightColor, const HitInfo& hitInfo)
{
    const V3 diffuse = computeLambertianModel(st, cameraDirection, lightDirection, lightColor, hitInfo);
    const V3 H = normalize(lightDirection + cameraDirection);
    const float d = dot(hitInfo.normal, H);
    if (d < 0)
        return diffuse;
    const V3 specular = lightColor * v3(hitInfo.material.ks) * std::pow(d, hitInfo.material.shininess);
    return diffuse + specular;
}
// src/shading.cpp:32-40 (the fixed gradient) and :155-189 (sample)
V3 linearGradientSample(float ti)
{
    static const float ts[5] = { 0.1f, 0.22f, 0.5f, 0.78f, 0.9f };
    static const V3 cs[5] = {
        { 215.f / 256.f, 210.f / 256.f, 203.f / 256.f },
        { 250.f / 256.f, 250.f / 256.f, 240.f / 256.f },
        { 145.f / 256.f, 170.f / 256.f, 175.f / 256.f },
        { 255.f / 256.f, 250.f / 256.f, 205.f / 256.f },
        { 170.f / 256.f, 170.f / 256.f, 170.f / 256.f },
    };
    if (ti <= ts[0])
        return cs[0];
    else if (ti >= ts[4])
        return cs[4];
    int it = 0; // lower_bound with `component.t <= value`: first component with t > ti
    while (ts[it] <= ti)
        it++;
    const float t = (ti - ts[it - 1]) / (ts[it] - ts[it - 1]);
    return mix(cs[it - 1], cs[it], t);
}
// src/shading.cpp:205-213
V3 computeLinearGradientModel(V3 lightDirection, V3 lightColor, const HitInfo& hitInfo)
{
    const float cos_theta = dot(lightDirection, hitInfo.normal);
    return lightColor * linearGradientSample(cos_theta);
}
// src/shading.cpp:29-56
V3 computeShading(State& st, V3 cameraDirection, V3 lightDirection, V3 lightColor, const HitInfo& hitInfo)
{
    if (st.f.enableShading) {
        switch (st.f.shadingModel) {
        case 0:
            return computeLambertianModel(st, cameraDirection, lightDirection, lightColor, hitInfo);
        case 1:
            return computePhongModel(st, cameraDirection, lightDirection, lightColor, hitInfo);
        case 2:
            return computeBlinnPhongModel(st, cameraDirection, lightDirection, lightColor, hitInfo);
        case 3:
            return computeLinearGradientModel(lightDirection, lightColor, hitInfo);
        }
    }
    return lightColor * sampleMaterialKd(st, hitInfo);
}

// ---------------------------------------------------------------- lights
// src/light.cpp:26-36
void sampleSegmentLight(float sample, const orc_light& l, V3& position, V3& color)
{
    position = mix(v3(l.v + 0), v3(l.v + 3), sample);
    color = mix(v3(l.v + 6), v3(l.v + 9), sample);
}
// src/light.cpp:38-74
void sampleParallelogramLight(V2 sample, const orc_light& l, V3& position, V3& color)
{
    const V3 v0 = v3(l.v + 0), e1 = v3(l.v + 3), e2 = v3(l.v + 6);
    position = v0 + sample.x * e1 + sample.y * e2;
    const float x = sample.x, y = sample.y;
    const float area0 = length(cross(e1 * x, e2 * y));
    const float area1 = length(cross(e1 * (1 - x), e2 * y));
    const float area2 = length(cross(e1 * x, e2 * (1 - y)));
    const float area3 = length(cross(e1 * (1 - x), e2 * (1 - y)));
    const float total = area0 + area1 + area2 + area3;
    const float bx = area0 / total, by = area1 / total, bz = area2 / total, bw = area3 / total;
    color = bw * v3(l.v + 9) + bz * v3(l.v + 12) + by * v3(l.v + 15) + bx * v3(l.v + 18);
}

// src/recursive.cpp:98-111: t is left at its default FLT_MAX
Ray generatePassthroughRay(const Ray& ray)
{
    const V3 intersection = ray.origin + ray.t * ray.direction;
    Ray r;
    r.origin = intersection + ray.direction * 1.0e-5f;
    r.direction = ray.direction;
    return r;
}
// src/recursive.cpp:72-89
Ray generateReflectionRay(const Ray& ray, const HitInfo& hitInfo)
{
    const V3 intersection = ray.origin + (ray.direction * ray.t);
    const V3 nd = normalize(ray.direction);
    const V3 reflected = normalize(nd - 2.0f * dot(nd, hitInfo.normal) * hitInfo.normal);
    Ray r;
    r.origin = intersection + hitInfo.normal * 1.0e-5f;
    r.direction = reflected;
    r.t = FLT_MAX;
    return r;
}

// src/light.cpp:87-109
bool visibilityOfLightSampleBinary(State& st, V3 lightPosition, const Ray& ray)
{
    if (!st.f.enableShadows)
        return true;
    const V3 hitPoint = ray.origin + ray.t * ray.direction;
    Ray shadowRay;
    shadowRay.origin = lightPosition;
    shadowRay.direction = normalize(hitPoint - lightPosition);
    const float shadowRayLength = length(hitPoint - lightPosition);
    shadowRay.t = shadowRayLength;
    HitInfo shadowHitInfo;
    if (intersect(st, shadowRay, shadowHitInfo))
        return std::fabs(shadowRay.t - shadowRayLength) <= 0.0001f;
    return true;
}
// src/light.cpp:126-151
V3 visibilityOfLightSampleTransparency(State& st, V3 lightPosition, V3 lightColor, const Ray& ray, const HitInfo& hitInfo)
{
    const V3 hitPoint = ray.origin + ray.t * ray.direction;
    Ray shadowRay;
    shadowRay.origin = lightPosition;
    shadowRay.direction = normalize(hitPoint - lightPosition);
    shadowRay.t = length(hitPoint - lightPosition);
    HitInfo shadowHitInfo;
    V3 finalColor = v3(0.0f);
    float transparency = 1.0f;
    orc_material currentMaterial = hitInfo.material;
    while (intersect(st, shadowRay, shadowHitInfo)
        && !allEpsilonEqual(shadowRay.origin + shadowRay.t * shadowRay.direction, hitPoint, 0.0001f)
        && shadowHitInfo.material.transparency != 1.0f) {
        currentMaterial = shadowHitInfo.material;
        transparency *= currentMaterial.transparency;
        shadowRay = generatePassthroughRay(shadowRay);
    }
    if (allEpsilonEqual(shadowRay.origin + shadowRay.t * shadowRay.direction, hitPoint, 0.0001f))
        finalColor = lightColor * v3(currentMaterial.kd) * transparency;
    return finalColor;
}
// src/light.cpp:268-280
V3 visibilityOfLightSample(State& st, V3 lightPosition, V3 lightColor, const Ray& ray, const HitInfo& hitInfo)
{
    if (!st.f.enableShadows)
        return lightColor;
    else if (!st.f.enableTransparency)
        return visibilityOfLightSampleBinary(st, lightPosition, ray) ? lightColor : v3(0.0f);
    return visibilityOfLightSampleTransparency(st, lightPosition, lightColor, ray, hitInfo);
}
// src/light.cpp:166-186
V3 computeContributionPointLight(State& st, V3 lightPosition, V3 lightColor, const Ray& ray, const HitInfo& hitInfo)
{
    const V3 visibleColor = visibilityOfLightSample(st, lightPosition, lightColor, ray, hitInfo);
    if (visibleColor == v3(0.0f))
        return v3(0.0f);
    const V3 p = ray.origin + ray.t * ray.direction;
    const V3 l = normalize(lightPosition - p);
    const V3 v = -ray.direction;
    HitInfo lightHitInfo = hitInfo;
    if (st.f.enableTransparency && hitInfo.material.transparency < 1.0f && dot(hitInfo.normal, l) < 0.0f)
        lightHitInfo.normal = -hitInfo.normal;
    return computeShading(st, v, l, visibleColor, lightHitInfo);
}
// src/light.cpp:283-298 with :205-219 and :239-253 inlined
V3 computeLightContribution(State& st, const Ray& ray, const HitInfo& hitInfo)
{
    V3 Lo = v3(0.0f);
    for (const orc_light& light : st.scene.lights) {
        if (light.type == 0) {
            Lo += computeContributionPointLight(st, v3(light.v + 0), v3(light.v + 3), ray, hitInfo);
        } else if (light.type == 1 || light.type == 2) {
            const uint32_t numSamples = st.f.numShadowSamples;
            V3 acc = v3(0.0f);
            for (uint32_t i = 0; i < numSamples; i++) {
                V3 position, color;
                if (light.type == 1)
                    sampleSegmentLight(st.sampler.next_1d(), light, position, color);
                else
                    sampleParallelogramLight(st.sampler.next_2d(), light, position, color);
                acc += computeContributionPointLight(st, position, color, ray, hitInfo);
            }
            Lo += acc * (1.0f / (float)numSamples);
        }
    }
    return clamp01(Lo);
}

// ---------------------------------------------------------------- camera
struct Camera {
    V3 lookAt;
    float dist, halfH, halfW;
    float qw, qx, qy, qz;
    // glm::quat(eulerAngles), detail/type_quat.inl:208-217
    explicit Camera(const orc_camera& c)
    {
        lookAt = v3(c.lookAt);
        dist = c.distance;
        halfH = std::tan(c.fovy / 2.0f); // framework/src/trackball.cpp:26-27
        halfW = c.aspect * halfH;
        const float cx = std::cos(c.rotation[0] * 0.5f), cy = std::cos(c.rotation[1] * 0.5f), cz = std::cos(c.rotation[2] * 0.5f);
        const float sx = std::sin(c.rotation[0] * 0.5f), sy = std::sin(c.rotation[1] * 0.5f), sz = std::sin(c.rotation[2] * 0.5f);
        qw = cx * cy * cz + sx * sy * sz;
        qx = sx * cy * cz - cx * sy * sz;
        qy = cx * sy * cz + sx * cy * sz;
        qz = cx * cy * sz - sx * sy * cz;
    }
    // glm quat * vec3, detail/type_quat.inl:347-354
    V3 rotate(V3 v) const
    {
        const V3 q { qx, qy, qz };
        const V3 uv = cross(q, v);
        const V3 uuv = cross(q, uv);
        return v + ((uv * qw) + uuv) * 2.0f;
    }
    V3 position() const { return lookAt + rotate({ 0, 0, -dist }); } // trackball.cpp:71-74
    V3 forward() const { return rotate({ 0, 0, 1 }); }               // :112-115
    V3 up() const { return rotate({ 0, 1, 0 }); }                    // :117-120
    V3 left() const { return rotate({ 1, 0, 0 }); }                  // :122-126
    // trackball.cpp:101-110
    Ray generateRay(V2 pixel) const
    {
        const V3 camDir = normalize(V3 { -pixel.x * halfW, pixel.y * halfH, 1.0f });
        Ray r;
        r.origin = position();
        r.direction = rotate(camDir);
        r.t = FLT_MAX;
        return r;
    }
};

// ---------------------------------------------------------------- recursion
V3 renderRay(State& st, Ray ray, int rayDepth);

// src/extra.cpp:378-459
V3 sampleEnvironmentMap(State& st, const Ray& ray)
{
    if (!st.f.enableEnvironmentMap)
        return v3(0.0f);
    int chosen = 0;
    float u = 0.0f, v = 0.0f;
    const float x = ray.direction.x, y = ray.direction.y, z = ray.direction.z;
    const float absX = std::fabs(x), absY = std::fabs(y), absZ = std::fabs(z);
    if (absX >= absY && absX >= absZ) {
        if (x > 0.0f) {
            chosen = 0, u = z, v = y;
        } else {
            chosen = 1, u = -z, v = y;
        }
    } else if (absY >= absZ) {
        if (y > 0.0f) {
            chosen = 2, u = -x, v = -z;
        } else {
            chosen = 3, u = -x, v = z;
        }
    } else {
        if (z > 0.0f) {
            chosen = 4, u = -x, v = y;
        } else {
            chosen = 5, u = x, v = y;
        }
    }
    if (st.scene.envFaces[chosen] < 0)
        return v3(0.0f);
    const float biggest = std::max(absX, std::max(absY, absZ));
    u = (u / biggest + 1.0f) * 0.5f;
    v = (v / biggest + 1.0f) * 0.5f;
    const orc_image& im = st.scene.images[st.scene.envFaces[chosen]];
    if (st.f.enableBilinearTextureFiltering)
        return sampleTextureBilinear(im, { u, v });
    return sampleTextureNearest(im, { u, v });
}

// src/extra.cpp:329-368
void renderRayGlossyComponent(State& st, const Ray& ray, const HitInfo& hitInfo, V3& hitColor, int rayDepth)
{
    V3 glossyColor = v3(0.0f);
    const V3 intersection = ray.origin + ray.direction * ray.t;
    const Ray initial = generateReflectionRay(ray, hitInfo);
    V3 arbitrary { 1, 0, 0 };
    if (length(arbitrary - initial.direction) < 0.01f)
        arbitrary = { 0, 1, 0 };
    const V3 u = normalize(cross(arbitrary, initial.direction));
    const V3 v = normalize(cross(initial.direction, u));
    const int numSamples = (int)st.f.numGlossySamples;
    const float diskRadius = st.f.glossyExponent * hitInfo.material.shininess / 64;
    for (int i = 0; i < numSamples; i++) {
        const V2 samples = st.sampler.next_2d();
        const float sampledDiskRadius = diskRadius * samples.x;
        const float theta = 6.28318530717958647692528676655900576f * samples.y;
        const V3 sampledU = u * sampledDiskRadius * std::cos(theta);
        const V3 sampledV = v * sampledDiskRadius * std::sin(theta);
        const V3 sampledDirection = normalize(initial.direction + sampledU + sampledV);
        Ray glossyRay;
        glossyRay.origin = intersection + sampledDirection * 1.0e-5f;
        glossyRay.direction = sampledDirection;
        glossyRay.t = FLT_MAX;
        glossyColor += renderRay(st, glossyRay, rayDepth + 1);
    }
    hitColor += (glossyColor / float(numSamples)) * v3(hitInfo.material.ks);
}

// src/recursive.cpp:25-64 (+ :123-127, :139-144)
V3 renderRay(State& st, Ray ray, int rayDepth)
{
    HitInfo hitInfo;
    if (!intersect(st, ray, hitInfo))
        return sampleEnvironmentMap(st, ray);
    V3 Lo = computeLightContribution(st, ray, hitInfo);
    if (rayDepth < st.f.maxRayDepth) {
        const V3 ks = v3(hitInfo.material.ks);
        const bool isReflective = ks.x != 0.0f || ks.y != 0.0f || ks.z != 0.0f;
        const bool isTransparent = hitInfo.material.transparency != 1.f;
        if (st.f.enableReflections && !st.f.enableGlossyReflection && isReflective) {
            const Ray reflection = generateReflectionRay(ray, hitInfo);
            Lo += renderRay(st, reflection, rayDepth + 1) * ks;
        }
        if (st.f.enableReflections && st.f.enableGlossyReflection && isReflective)
            renderRayGlossyComponent(st, ray, hitInfo, Lo, rayDepth);
        if (st.f.enableTransparency && isTransparent) {
            const Ray passthrough = generatePassthroughRay(ray);
            Lo = (renderRay(st, passthrough, rayDepth + 1) * hitInfo.material.transparency)
                + (Lo * (1.0f - hitInfo.material.transparency));
        }
    }
    return Lo;
}

// src/recursive.cpp:10-17
V3 renderRays(State& st, const std::vector<Ray>& rays)
{
    V3 L = v3(0.0f);
    for (const Ray& r : rays)
        L += renderRay(st, r, 0);
    return L / static_cast<float>(rays.size());
}

// src/render.cpp:56-131
std::vector<Ray> generatePixelRays(State& st, const Camera& cam, int px, int py, int resX, int resY)
{
    std::vector<Ray> rays;
    if (st.f.numPixelSamples > 1) {
        if (st.f.enableJitteredSampling) {
            const uint32_t numSamples = static_cast<uint32_t>(std::round(std::sqrt(float(st.f.numPixelSamples))));
            for (int p = 0; p < (int)numSamples; p++) {
                for (int q = 0; q < (int)numSamples; q++) {
                    const float jx = px + (p + st.sampler.next_1d()) / numSamples;
                    const float jy = py + (q + st.sampler.next_1d()) / numSamples;
                    const V2 position { jx / float(resX) * 2.f - 1.f, jy / float(resY) * 2.f - 1.f };
                    rays.push_back(cam.generateRay(position));
                }
            }
        } else {
            for (uint32_t i = 0; i < st.f.numPixelSamples; i++) {
                const V2 s = st.sampler.next_2d();
                const V2 position { (float(px) + s.x) / float(resX) * 2.f - 1.f, (float(py) + s.y) / float(resY) * 2.f - 1.f };
                rays.push_back(cam.generateRay(position));
            }
        }
    } else {
        const V2 position { (float(px) + 0.5f) / float(resX) * 2.f - 1.f, (float(py) + 0.5f) / float(resY) * 2.f - 1.f };
        rays.push_back(cam.generateRay(position));
    }
    return rays;
}

// src/extra.cpp:15-23
V3 getPointOfFocus(const Camera& cam, const Ray& cameraRay, float focalDistance)
{
    const float cosAngle = dot(cameraRay.direction, cam.forward());
    const float curDistance = focalDistance / cosAngle;
    return cameraRay.origin + curDistance * cameraRay.direction;
}

// src/extra.cpp:27-81
std::vector<Ray> generatePixelRaysForDepthOfField(State& st, const Camera& cam, int px, int py, int resX, int resY)
{
    const float focalDistance = st.f.depthOfFieldDistance;
    const float circleRadius = st.f.depthOfFieldCircleDiameter * 0.5f;
    const uint32_t numOfSamples = st.f.numDepthOfFieldSamples;
    const std::vector<Ray> generated = generatePixelRays(st, cam, px, py, resX, resY);
    std::vector<Ray> finalRays;
    for (const Ray& cameraRay : generated) {
        const V3 pointOfFocus = getPointOfFocus(cam, cameraRay, focalDistance);
        for (uint32_t i = 0; i < numOfSamples; i++) {
            const V2 rv = st.sampler.next_2d();
            const float angle = rv.x * 6.28318530717958647692528676655900576f;
            const float distance = std::sqrt(rv.y) * circleRadius;
            const float ox = std::cos(angle) * distance;
            const float oy = std::sin(angle) * distance;
            Ray r;
            r.origin = cameraRay.origin + ox * cam.up() + oy * cam.left();
            r.direction = normalize(pointOfFocus - r.origin);
            r.t = cameraRay.t;
            finalRays.push_back(r);
        }
    }
    return finalRays;
}

void fillHit(orc_hit& o, bool hit, const Ray& ray, const HitInfo& hi)
{
    std::memset(&o, 0, sizeof(o));
    o.hit = hit ? 1 : 0;
    o.t = ray.t;
    o.primitive = hit ? hi.prim : -1;
    o.material = hit ? hi.matSlot : -1;
    if (hit) {
        o.normal[0] = hi.normal.x, o.normal[1] = hi.normal.y, o.normal[2] = hi.normal.z;
        o.barycentricCoord[0] = hi.bary.x, o.barycentricCoord[1] = hi.bary.y, o.barycentricCoord[2] = hi.bary.z;
        o.texCoord[0] = hi.texCoord.x, o.texCoord[1] = hi.texCoord.y;
    }
}

void addStats(orc_stats& a, const orc_stats& b)
{
    a.rays += b.rays;
    a.inner_nodes += b.inner_nodes;
    a.box_tests += b.box_tests;
    a.tri_tests += b.tri_tests;
}

Ray toRay(const orc_ray& r)
{
    Ray q;
    q.origin = v3(r.origin);
    q.direction = v3(r.direction);
    q.t = r.t;
    return q;
}

} // namespace

extern "C" {

int orc_bvh_build(const orc_primitive* prims_in, uint32_t n, int sah_binning, orc_node* nodes_out,
    orc_primitive* prims_out, uint32_t* n_nodes, uint32_t* n_levels, uint32_t* n_leaves)
{
    std::vector<orc_primitive> prims(prims_in, prims_in + n);
    Builder b;
    b.sahBinning = sah_binning != 0;
    b.out.reserve(n);
    b.nodes.reserve(n + 1);
    b.nodes.emplace_back(); // root
    b.nodes.emplace_back(); // dummy, src/bvh.cpp:88-89
    std::memset(&b.nodes[1], 0, sizeof(orc_node));
    // The reference's dummy is a value-initialised Node: aabb {0},{1}, data {0,0}.
    b.nodes[1].upper[0] = b.nodes[1].upper[1] = b.nodes[1].upper[2] = 1.0f;
    b.buildRecursive(prims.data(), n, 0);
    std::copy(b.nodes.begin(), b.nodes.end(), nodes_out);
    std::copy(b.out.begin(), b.out.end(), prims_out);
    *n_nodes = static_cast<uint32_t>(b.nodes.size());
    // src/bvh.cpp:498-547
    std::vector<uint32_t> levels(b.nodes.size(), 0);
    uint32_t numLevels = 1, numLeaves = 0;
    for (size_t i = 0; i < b.nodes.size(); i++) {
        if (i == 1)
            continue;
        const orc_node& nd = b.nodes[i];
        if (!(nd.data[0] & LeafBit)) {
            levels[nd.data[0]] = levels[i] + 1;
            levels[nd.data[1]] = levels[i] + 1;
        } else {
            numLeaves++;
        }
    }
    for (size_t i = 2; i < b.nodes.size(); i++)
        numLevels = std::max(numLevels, levels[i] + 1);
    *n_levels = numLevels;
    *n_leaves = numLeaves;
    return 0;
}

orc_scene* orc_scene_create(const orc_node* nodes, uint32_t n_nodes, const orc_primitive* prims, uint32_t n_prims,
    const orc_material* mesh_materials, uint32_t n_meshes, const orc_image* images, uint32_t n_images,
    const int32_t env_faces[6], const orc_sphere* spheres, uint32_t n_spheres, const orc_light* lights, uint32_t n_lights)
{
    orc_scene* s = new orc_scene;
    s->nodes.assign(nodes, nodes + n_nodes);
    s->prims.assign(prims, prims + n_prims);
    s->materials.assign(mesh_materials, mesh_materials + n_meshes);
    s->imagePixels.resize(n_images);
    s->images.resize(n_images);
    for (uint32_t i = 0; i < n_images; i++) {
        const size_t cnt = size_t(images[i].width) * images[i].height * 3;
        s->imagePixels[i].assign(images[i].pixels, images[i].pixels + cnt);
        s->images[i] = { s->imagePixels[i].data(), images[i].width, images[i].height };
    }
    for (int i = 0; i < 6; i++)
        s->envFaces[i] = env_faces ? env_faces[i] : -1;
    s->spheres.assign(spheres, spheres + n_spheres);
    s->lights.assign(lights, lights + n_lights);
    return s;
}

void orc_scene_destroy(orc_scene* scene) { delete scene; }

void orc_intersect_batch(const orc_scene* scene, const orc_features* features, orc_ray* rays, uint32_t n, orc_hit* out,
    int policy, orc_stats* stats)
{
    orc_stats total {};
#pragma omp parallel
    {
        State st { *scene, *features, Sampler { 0 } };
        st.policy = policy;
#pragma omp for schedule(static)
        for (int64_t i = 0; i < (int64_t)n; i++) {
            Ray r = toRay(rays[i]);
            HitInfo hi;
            const bool hit = intersect(st, r, hi);
            rays[i].t = r.t;
            fillHit(out[i], hit, r, hi);
        }
#pragma omp critical
        addStats(total, st.stats);
    }
    if (stats)
        *stats = total;
}

// postprocessImageWithBloom and its helpers, src/extra.cpp:214-316, on a frame in Screen order
// (pixel (x, y) at flat index (H - 1 - y) * W + x, src/screen.cpp:36-42, :128-131), statement by
// statement and in place like the reference.  Two things there are undefined behaviour and have
// to be given a meaning: filter sizes below 3 write outside the image and above 20 overflow
// factorial() (refused, -1), and the filter window reaches one row above the image (flat index
// below zero): such a read counts as black.  A flat index inside the buffer is read whatever
// column arithmetic produced it, as there.
int orc_bloom(const orc_features* features, float* rgb, int width, int height)
{
    if (!features->enableBloomEffect)
        return 0;
    const int k = int(features->bloomFilterSize);
    if (k < 3 || k > 20)
        return -1;
    const long long n = (long long)width * height;
    std::vector<V3> image(n), thresholded(n);
    for (long long i = 0; i < n; i++)
        image[i] = thresholded[i] = V3 { rgb[3 * i], rgb[3 * i + 1], rgb[3 * i + 2] };
    auto indexAt = [&](int x, int y) { return (long long)(height - 1 - y) * width + x; };

    // computeHighpassFilter, :214-226
    const V3 luminanceVector { 0.2126f, 0.7152f, 0.0722f };
    for (V3& pixel : thresholded) {
        const float luminance = dot(luminanceVector, pixel);
        if (luminance < features->bloomThreshold)
            pixel = V3 { 0.f, 0.f, 0.f };
    }

    // computeGaussianFilter, :239-271 (both passes, the second overwrites the first)
    auto factorial = [](int m) {
        long long f = 1;
        for (int i = 1; i <= m; i++)
            f *= i;
        return f;
    };
    std::vector<std::vector<float>> filter(k, std::vector<float>(k));
    for (int y = 0; y < k; y++) {
        float total = 0;
        for (int x = 0; x < k; x++) {
            const float val = float(factorial(k) / (factorial(x) * factorial(k - x)));
            total += val;
            filter[x][y] = val;
        }
        for (int x = 0; x < k; x++)
            filter[x][y] /= total;
    }
    for (int x = 0; x < k; x++) {
        float total = 0;
        for (int y = 0; y < k; y++) {
            const float val = float(factorial(k) / (factorial(y) * factorial(k - y)));
            total += val;
            filter[x][y] = val;
        }
        for (int y = 0; y < k; y++)
            filter[x][y] /= total;
    }

    // :296-308
    for (int x = k - 2; x < width - k + 2; x++) {
        for (int y = k - 2; y < height - k + 2; y++) {
            V3 color { 0.f, 0.f, 0.f };
            for (int i = 0; i < k; i++) {
                for (int j = 0; j < k; j++) {
                    const long long idx = indexAt(x + i, y + j);
                    const V3 px = (idx >= 0 && idx < n) ? thresholded[idx] : V3 { 0.f, 0.f, 0.f };
                    color = color + filter[i][j] * px;
                }
            }
            thresholded[indexAt(x, y)] = color;
        }
    }
    // :311-315
    for (int x = 0; x < width; x++)
        for (int y = 0; y < height; y++)
            image[indexAt(x, y)] = image[indexAt(x, y)] + thresholded[indexAt(x, y)];
    for (long long i = 0; i < n; i++)
        rgb[3 * i] = image[i].x, rgb[3 * i + 1] = image[i].y, rgb[3 * i + 2] = image[i].z;
    return 0;
}

// Screen::writeBitmapToFile, src/screen.cpp:44-55, through stbi_write_bmp with 4 components
// (framework/third_party/stb/include/stb/stb_image_write.h:492-510): 14-byte file header, 108-byte
// V4 header (BI_BITFIELDS, 32 bpp), then the rows bottom-up, each pixel as B, G, R, A.
// `out` receives 122 + width * height * 4 bytes.
void orc_bmp_bytes(const float* rgb, int width, int height, uint8_t* out)
{
    auto put16 = [&](size_t at, uint32_t v) { out[at] = uint8_t(v & 255), out[at + 1] = uint8_t((v >> 8) & 255); };
    auto put32 = [&](size_t at, uint32_t v) { put16(at, v & 0xffff), put16(at + 2, v >> 16); };
    std::memset(out, 0, 122);
    out[0] = 'B', out[1] = 'M';
    put32(2, uint32_t(14 + 108 + (size_t)width * height * 4));
    put32(10, 14 + 108);
    put32(14, 108), put32(18, uint32_t(width)), put32(22, uint32_t(height));
    put16(26, 1), put16(28, 32);
    put32(30, 3);
    put32(54, 0xff0000u), put32(58, 0xff00u), put32(62, 0xffu), put32(66, 0xff000000u);
    uint8_t* px = out + 122;
    for (int row = height - 1; row >= 0; row--) { // stb: vdir -1, the last row of the data first
        for (int x = 0; x < width; x++) {
            const float* c = rgb + 3 * ((size_t)row * width + x);
            uint8_t v[3];
            for (int ch = 0; ch < 3; ch++) {
                const float clamped = std::min(std::max(c[ch], 0.0f), 1.0f); // glm::clamp
                v[ch] = uint8_t(clamped * 255.0f);                             // glm::u8vec4(vec4): truncation
            }
            *px++ = v[2], *px++ = v[1], *px++ = v[0], *px++ = 255;
        }
    }
}

int orc_render(const orc_scene* scene, const orc_features* features, const orc_camera* camera, int width, int height,
    float* out_rgb, orc_stats* stats, int policy, int threads)
{
    if (features->enableMotionBlur && !features->enableDepthOfField)
        return -1; // orc_render_motion_blur, which needs the primitives in the scene's order on top
    if (features->enableBloomEffect && (features->bloomFilterSize < 3 || features->bloomFilterSize > 20))
        return -1;
    const Camera cam(*camera);
    orc_stats total {};
    // debugging aid: ORC_DUMP_PIXEL="x y" ORC_DUMP_FILE=path writes every intersect call of that pixel
    FILE* dumpFile = nullptr;
    int dumpX = -1, dumpY = -1;
    if (const char* px = std::getenv("ORC_DUMP_PIXEL"))
        if (const char* path = std::getenv("ORC_DUMP_FILE"))
            if (std::sscanf(px, "%d %d", &dumpX, &dumpY) == 2)
                dumpFile = std::fopen(path, "w");
#ifdef _OPENMP
    const int nthreads = threads > 0 ? threads : omp_get_max_threads();
#else
    const int nthreads = 1;
#endif
    // src/render.cpp:27-45 and src/extra.cpp:95-112: rows in parallel, guided.
#pragma omp parallel num_threads(nthreads)
    {
        orc_stats local {};
#pragma omp for schedule(guided)
        for (int y = 0; y < height; y++) {
            for (int x = 0; x != width; x++) {
                State st { *scene, *features, Sampler { static_cast<uint32_t>(height * x + y) } };
                st.policy = policy;
                if (dumpFile && x == dumpX && y == dumpY)
                    st.dump = dumpFile;
                const std::vector<Ray> rays = features->enableDepthOfField
                    ? generatePixelRaysForDepthOfField(st, cam, x, y, width, height)
                    : generatePixelRays(st, cam, x, y, width, height);
                const V3 L = renderRays(st, rays);
                float* px = out_rgb + 3 * (size_t(height - 1 - y) * width + x); // src/screen.cpp:36-42
                px[0] = L.x, px[1] = L.y, px[2] = L.z;
                addStats(local, st.stats);
            }
        }
#pragma omp critical
        addStats(total, local);
    }
    if (stats)
        *stats = total;
    if (dumpFile)
        std::fclose(dumpFile);
    return orc_bloom(features, out_rgb, width, height); // src/render.cpp:48-50
}

namespace {

// splineMat (src/extra.cpp:546-567) applied to a position the way renderImageWithMotionBlur applies it
// (src/extra.cpp:152-154, :171-173): the Bezier point of the five control points offset * movement +
// position, glm::translate(mat4(1), point) (ext/matrix_transform.inl:13), the mat4 * vec4 product
// (detail/type_mat4x4.inl:561-572) and the division by w.
V3 movedPosition(float t, V3 currentCenter, float movement)
{
    const V3 p0 = (V3 { 0.0f, 0.0f, 0.0f } * movement) + currentCenter;
    const V3 p1 = (V3 { 0.0f, 1.0f, 1.0f } * movement) + currentCenter;
    const V3 p2 = (V3 { 1.0f, 1.0f, -1.0f } * movement) + currentCenter;
    const V3 p3 = (V3 { 1.0f, 0.0f, 0.0f } * movement) + currentCenter;
    const V3 p4 = (V3 { 1.5f, 1.0f, 2.0f } * movement) + currentCenter;
    const float oneMinusT = 1.0f - t;
    const float oneMinusTSquared = oneMinusT * oneMinusT;
    const float tSquared = t * t;
    const float tCubed = tSquared * t;
    const V3 posBezier = (oneMinusTSquared * oneMinusTSquared * p0) + (4.0f * oneMinusTSquared * oneMinusT * t * p1)
        + (6.0f * oneMinusTSquared * tSquared * p2) + (4.0f * oneMinusT * tCubed * p3) + (tSquared * tSquared * p4);
    // columns of the identity, then Result[3] = m[0] * v[0] + m[1] * v[1] + m[2] * v[2] + m[3]
    float m[4][4] = { { 1, 0, 0, 0 }, { 0, 1, 0, 0 }, { 0, 0, 1, 0 }, { 0, 0, 0, 1 } };
    float col3[4];
    for (int r = 0; r < 4; r++)
        col3[r] = ((m[0][r] * posBezier.x + m[1][r] * posBezier.y) + m[2][r] * posBezier.z) + m[3][r];
    for (int r = 0; r < 4; r++)
        m[3][r] = col3[r];
    const float v[4] = { currentCenter.x, currentCenter.y, currentCenter.z, 1.0f };
    float out[4];
    for (int r = 0; r < 4; r++) {
        const float add0 = m[0][r] * v[0] + m[1][r] * v[1];
        const float add1 = m[2][r] * v[2] + m[3][r] * v[3];
        out[r] = add0 + add1;
    }
    return V3 { out[0] / out[3], out[1] / out[3], out[2] / out[3] };
}

} // namespace

// renderImageWithMotionBlur (src/extra.cpp:121-211) followed by the bloom pass of renderImage
// (src/render.cpp:48-50).  Per pixel and per time sample the reference copies the scene with every
// vertex and sphere centre moved, builds a BVH over the copy (BVH::BVH takes the meshes' triangles in
// the scene's order, which is why this function wants the primitives in THAT order, not the order a
// built tree leaves them in) and renders the pixel's rays in it with a copy of the pixel's sampler.
// One deviation, which the reference's undefined behaviour forces: the sum over the time samples
// starts at zero here, where the reference adds into an uninitialised vector (src/extra.cpp:144, :205).
int orc_render_motion_blur(const orc_scene* scene, const orc_primitive* scene_order_prims, uint32_t n_prims,
    const orc_features* features, const orc_camera* camera, int width, int height, float* out_rgb, orc_stats* stats,
    int threads)
{
    if (!features->enableMotionBlur || features->enableDepthOfField)
        return -1;
    if (features->enableBloomEffect && (features->bloomFilterSize < 3 || features->bloomFilterSize > 20))
        return -1;
    const Camera cam(*camera);
    const int samples = features->motionBlurSamples;
    const float movement = features->movement;
    orc_stats total {};
#ifdef _OPENMP
    const int nthreads = threads > 0 ? threads : omp_get_max_threads();
#else
    const int nthreads = 1;
#endif
#pragma omp parallel num_threads(nthreads)
    {
        orc_stats local {};
        // the copy of the scene: everything but nodes, primitives and spheres stays as it is
        orc_scene newScene;
        newScene.materials = scene->materials;
        newScene.images = scene->images; // texel storage stays with the original
        for (int i = 0; i < 6; i++)
            newScene.envFaces[i] = scene->envFaces[i];
        newScene.lights = scene->lights;
        std::vector<orc_primitive> movedPrims(n_prims);
#pragma omp for schedule(guided)
        for (int y = 0; y < height; y++) {
            for (int x = 0; x != width; x++) {
                State state { *scene, *features, Sampler { static_cast<uint32_t>(height * x + y) } };
                V3 L = v3(0.0f);
                for (int i = 0; i < samples; i++) {
                    const float time = state.sampler.next_1d();
                    newScene.spheres = scene->spheres;
                    for (orc_sphere& sp : newScene.spheres) {
                        const V3 c = movedPosition(time, v3(sp.center), movement);
                        sp.center[0] = c.x, sp.center[1] = c.y, sp.center[2] = c.z;
                    }
                    for (uint32_t k = 0; k < n_prims; k++) {
                        orc_primitive q = scene_order_prims[k];
                        for (orc_vertex* vtx : { &q.v0, &q.v1, &q.v2 }) {
                            const V3 c = movedPosition(time, v3(vtx->position), movement);
                            vtx->position[0] = c.x, vtx->position[1] = c.y, vtx->position[2] = c.z;
                        }
                        movedPrims[k] = q;
                    }
                    Builder b; // BVH newBvh = BVH(newScene, features), src/extra.cpp:196
                    b.sahBinning = features->enableBvhSahBinning != 0;
                    b.out.reserve(n_prims);
                    b.nodes.reserve(n_prims + 1);
                    b.nodes.emplace_back();
                    b.nodes.emplace_back();
                    std::memset(&b.nodes[1], 0, sizeof(orc_node));
                    b.nodes[1].upper[0] = b.nodes[1].upper[1] = b.nodes[1].upper[2] = 1.0f;
                    b.buildRecursive(movedPrims.data(), n_prims, 0);
                    newScene.nodes.swap(b.nodes);
                    newScene.prims.swap(b.out);
                    State newState { newScene, *features, state.sampler };
                    const std::vector<Ray> rays = generatePixelRays(newState, cam, x, y, width, height);
                    L += renderRays(newState, rays);
                    addStats(local, newState.stats);
                }
                L = L / static_cast<float>(samples);
                float* px = out_rgb + 3 * (size_t(height - 1 - y) * width + x);
                px[0] = L.x, px[1] = L.y, px[2] = L.z;
            }
        }
#pragma omp critical
        addStats(total, local);
    }
    if (stats)
        *stats = total;
    return orc_bloom(features, out_rgb, width, height);
}

void orc_primary_hits(const orc_scene* scene, const orc_features* features, const orc_camera* camera, int width,
    int height, int32_t* prim_out, float* t_out)
{
    const Camera cam(*camera);
#pragma omp parallel for schedule(guided)
    for (int y = 0; y < height; y++) {
        State st { *scene, *features, Sampler { 0 } };
        for (int x = 0; x < width; x++) {
            const V2 position { (float(x) + 0.5f) / float(width) * 2.f - 1.f, (float(y) + 0.5f) / float(height) * 2.f - 1.f };
            Ray r = cam.generateRay(position);
            HitInfo hi;
            const bool hit = intersect(st, r, hi);
            const size_t idx = size_t(height - 1 - y) * width + x;
            prim_out[idx] = hit ? hi.prim : -1;
            t_out[idx] = r.t;
        }
    }
}

void orc_generate_rays(const orc_camera* camera, const float* ndc_xy, uint32_t n, orc_ray* out)
{
    const Camera cam(*camera);
    for (uint32_t i = 0; i < n; i++) {
        const Ray r = cam.generateRay({ ndc_xy[2 * i], ndc_xy[2 * i + 1] });
        out[i] = { { r.origin.x, r.origin.y, r.origin.z }, { r.direction.x, r.direction.y, r.direction.z }, r.t };
    }
}

void orc_sampler_next(uint32_t seed, uint32_t n, float* out)
{
    Sampler s { seed };
    for (uint32_t i = 0; i < n; i++)
        out[i] = s.next_1d();
}

void orc_barycentric(const float v0[3], const float v1[3], const float v2[3], const float p[3], float out[3])
{
    const V3 r = computeBarycentricCoord(v3(v0), v3(v1), v3(v2), v3(p));
    out[0] = r.x, out[1] = r.y, out[2] = r.z;
}
void orc_interpolate_normal(const float n0[3], const float n1[3], const float n2[3], const float bc[3], float out[3])
{
    const V3 r = interpolateNormal(v3(n0), v3(n1), v3(n2), v3(bc));
    out[0] = r.x, out[1] = r.y, out[2] = r.z;
}
void orc_interpolate_texcoord(const float t0[2], const float t1[2], const float t2[2], const float bc[3], float out[2])
{
    const V2 r = interpolateTexCoord({ t0[0], t0[1] }, { t1[0], t1[1] }, { t2[0], t2[1] }, v3(bc));
    out[0] = r.x, out[1] = r.y;
}
void orc_sample_texture(const orc_image* image, const float uv[2], int bilinear, float out[3])
{
    const V3 r = bilinear ? sampleTextureBilinear(*image, { uv[0], uv[1] }) : sampleTextureNearest(*image, { uv[0], uv[1] });
    out[0] = r.x, out[1] = r.y, out[2] = r.z;
}
void orc_sample_segment_light(float sample, const orc_light* light, float pos[3], float color[3])
{
    V3 p, c;
    sampleSegmentLight(sample, *light, p, c);
    pos[0] = p.x, pos[1] = p.y, pos[2] = p.z;
    color[0] = c.x, color[1] = c.y, color[2] = c.z;
}
void orc_sample_parallelogram_light(const float sample[2], const orc_light* light, float pos[3], float color[3])
{
    V3 p, c;
    sampleParallelogramLight({ sample[0], sample[1] }, *light, p, c);
    pos[0] = p.x, pos[1] = p.y, pos[2] = p.z;
    color[0] = c.x, color[1] = c.y, color[2] = c.z;
}
void orc_point_of_focus(const orc_camera* camera, const orc_ray* ray, float focal_distance, float out[3])
{
    const Camera cam(*camera);
    const V3 r = getPointOfFocus(cam, toRay(*ray), focal_distance);
    out[0] = r.x, out[1] = r.y, out[2] = r.z;
}
void orc_camera_basis(const orc_camera* camera, float position[3], float forward[3], float up[3], float left[3])
{
    const Camera cam(*camera);
    const V3 p = cam.position(), f = cam.forward(), u = cam.up(), l = cam.left();
    position[0] = p.x, position[1] = p.y, position[2] = p.z;
    forward[0] = f.x, forward[1] = f.y, forward[2] = f.z;
    up[0] = u.x, up[1] = u.y, up[2] = u.z;
    left[0] = l.x, left[1] = l.y, left[2] = l.z;
}
int orc_triangle_test(const float v0[3], const float v1[3], const float v2[3], orc_ray* ray)
{
    Ray r = toRay(*ray);
    const bool h = intersectRayWithTriangle(v3(v0), v3(v1), v3(v2), r);
    ray->t = r.t;
    return h;
}
int orc_aabb_test(const float lower[3], const float upper[3], orc_ray* ray)
{
    Ray r = toRay(*ray);
    const bool h = intersectRayWithAABB(lower, upper, r);
    ray->t = r.t;
    return h;
}
int orc_sphere_test(const orc_sphere* s, orc_ray* ray, float normal_out[3])
{
    Ray r = toRay(*ray);
    HitInfo hi;
    const bool h = intersectRayWithSphere(*s, r, hi, 0);
    ray->t = r.t;
    normal_out[0] = hi.normal.x, normal_out[1] = hi.normal.y, normal_out[2] = hi.normal.z;
    return h;
}
void orc_compute_shading(const orc_features* features, const orc_material* material, const float normal[3],
    const float camera_dir[3], const float light_dir[3], const float light_color[3], float out[3])
{
    orc_scene empty;
    State st { empty, *features, Sampler { 0 } };
    HitInfo hi;
    hi.material = *material;
    hi.material.kdTexture = -1;
    hi.normal = v3(normal);
    const V3 r = computeShading(st, v3(camera_dir), v3(light_dir), v3(light_color), hi);
    out[0] = r.x, out[1] = r.y, out[2] = r.z;
}
float orc_linear_gradient_sample_r(float ti, float out[3])
{
    const V3 r = linearGradientSample(ti);
    out[0] = r.x, out[1] = r.y, out[2] = r.z;
    return r.x;
}

} // extern "C"
