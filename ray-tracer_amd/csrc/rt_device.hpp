// rt_device.hpp -- device-side data layout and per-ray / per-hit arithmetic
// for gfx950.  Everything is FP32 VALU work (no dense contraction anywhere on
// this path, so no MFMA).  The translation unit is built with
// -ffp-contract=off: each float is rounded where the reference's scalar code
// rounds it, and the only fused operations are the explicit ones in the
// bounding-box test, whose result is made conservative instead of exact.
#pragma once

#include <hip/hip_runtime.h>
#include "rt_fastdiv.hpp"
#include "rt_trig.hpp"
#include "rt_pow.hpp"
#include <stdint.h>

#include "../../include/rt_amd.h"

namespace rt {

// ----------------------------------------------------------------------------
// glm-shaped vector helpers (operation order as in glm 0.9.9.9:
// dot = (x+y)+z, cross per detail/func_geometric.inl:74-77,
// normalize = v * (1/sqrt(dot)), vec/scalar = per-component division).
// ----------------------------------------------------------------------------
struct f3 {
    float x, y, z;
};
struct f2 {
    float x, y;
};

#define RT_HD __host__ __device__ __forceinline__

RT_HD f3 mk3(float x, float y, float z) { return f3 { x, y, z }; }
RT_HD f3 mk3(float s) { return f3 { s, s, s }; }
RT_HD f3 operator+(f3 a, f3 b) { return { a.x + b.x, a.y + b.y, a.z + b.z }; }
RT_HD f3 operator-(f3 a, f3 b) { return { a.x - b.x, a.y - b.y, a.z - b.z }; }
RT_HD f3 operator-(f3 a) { return { -a.x, -a.y, -a.z }; }
RT_HD f3 operator*(f3 a, f3 b) { return { a.x * b.x, a.y * b.y, a.z * b.z }; }
RT_HD f3 operator*(f3 a, float s) { return { a.x * s, a.y * s, a.z * s }; }
RT_HD f3 operator*(float s, f3 a) { return { s * a.x, s * a.y, s * a.z }; }
RT_HD f3 operator/(f3 a, float s) { return { a.x / s, a.y / s, a.z / s }; }
RT_HD float dot(f3 a, f3 b)
{
    const float tx = a.x * b.x, ty = a.y * b.y, tz = a.z * b.z;
    return tx + ty + tz;
}
RT_HD f3 cross(f3 x, f3 y) { return { x.y * y.z - y.y * x.z, x.z * y.x - y.z * x.x, x.x * y.y - y.x * x.y }; }
RT_HD float length(f3 v) { return sqrtf(dot(v, v)); }
RT_HD f3 normalize(f3 v) { return v * (1.0f / sqrtf(dot(v, v))); }
RT_HD float gmin(float x, float y) { return (y < x) ? y : x; } // glm::min, NaN behaviour included
RT_HD float gmax(float x, float y) { return (x < y) ? y : x; } // glm::max
RT_HD f3 clamp01(f3 v)
{
    return { gmin(gmax(v.x, 0.0f), 1.0f), gmin(gmax(v.y, 0.0f), 1.0f), gmin(gmax(v.z, 0.0f), 1.0f) };
}
RT_HD f3 mix3(f3 x, f3 y, float a) { return x * (1.0f - a) + y * a; }
RT_HD bool allEpsEq(f3 a, f3 b, float eps)
{
    return fabsf(a.x - b.x) < eps && fabsf(a.y - b.y) < eps && fabsf(a.z - b.z) < eps;
}
RT_HD bool isZero(f3 a) { return a.x == 0.0f && a.y == 0.0f && a.z == 0.0f; }

// glm quat * vec3, detail/type_quat.inl:347-354
RT_HD f3 quatRotate(float qw, f3 q, f3 v)
{
    const f3 uv = cross(q, v);
    const f3 uuv = cross(q, uv);
    return v + ((uv * qw) + uuv) * 2.0f;
}

// ----------------------------------------------------------------------------
// Device-resident scene (HBM; the whole working set of the bundled scenes is a
// few MB and lives in L2 / Infinity Cache after the first touches).
//
// pairs : one 32-B record per inner node = both children's boxes + child refs,
//         so one inner-node visit is two 16-B fetches per lane.  Box planes are
//         16-bit positions on a per-axis grid spanning the root box (plane =
//         gridMin + q * gridCell), lower planes rounded down and upper planes
//         rounded up, so a quantised box always contains the float box; the
//         traversal only has to be conservative, the triangle test decides.
//           [0] Lx Ly Lz Rx   [1] Ry Rz refL refR     each box dword = lo | hi << 16
//         child ref: inner -> byte offset of the child's own pair record (index * 32: a step's
//                             two loads take it as their offset from the array's base);
//                    leaf  -> kLeafFlag | count << 27 | offset   (count <= 15)
// tris  : one 96-B record per primitive (6 x dwordx4), holding the
//         ray-independent part of the reference's plane + three edge tests:
//           [0] n.xyz D        n = normalize(cross(v1-v0, v2-v0)), D = dot(v0, n)
//           [1] E0.xyz v0.x    E0 = cross(v2-v0, n)
//           [2] v0.yz E1.xy    E1 = cross(v0-v1, n)
//           [3] E1.z v1.xyz
//           [4] E2.xyz v2.x    E2 = cross(v1-v2, n)
//           [5] v2.yz 0 0
// attrs : one 64-B record per primitive, read once per accepted closest hit:
//           [0] n0.xyz n1.x  [1] n1.yz n2.xy  [2] n2.z uv0.xy uv1.x
//           [3] uv1.y uv2.xy meshID(bits)
// ----------------------------------------------------------------------------
constexpr uint32_t kLeafFlag = 0x80000000u;
constexpr uint32_t kPairBytes = 32u;
constexpr uint32_t kRefNone = 0x7fffffffu; // "no node": traversal finished
constexpr uint32_t kLeafOffsetMask = (1u << 27) - 1u;
constexpr int kMaxLeafCount = 15;
constexpr int kMaxStack = 32;   // traversal stack entries per lane (LDS)
constexpr int kMaxDepth = 7;    // deepest accepted maxRayDepth
constexpr int kTile = 8;        // sharding / work-queue tile edge (8x8 = one wave)

struct DevMaterial {
    float kd[3];
    float ks[3];
    float shininess;
    float transparency;
    int32_t tex;
    int32_t pad[3];
};
struct DevTex {
    uint32_t offset; // in floats into the texel pool
    int32_t width, height;
    int32_t pad;
};
struct DevSphere {
    float center[3];
    float radius;
    int32_t material; // slot
    int32_t pad[3];
};

struct DevScene {
    const uint4* pairs;
    const float4* tris;
    const float4* attrs;
    const DevMaterial* mats;
    const DevTex* texs;
    const float* texels;
    const DevSphere* spheres;
    const rt_light* lights;
    uint32_t rootRef;
    float rootLo[3], rootHi[3];
    float gridMin[3], gridCell[3];
    uint32_t nPrims;
    int32_t nMeshes, nSpheres, nLights, nTex;
    int32_t env[6];
    int32_t tieByIndex; // equal-t hits: the higher primitive index wins, whatever the order of the tests
    // The caller's own nodes as they came (src/bvh_interface.h:28-54, two 16-B words each: lower,
    // upper, data[2]); only read when a ray has to be replayed in the reference's order of tests
    // (Tracer::settleTie).  Null when the caller's tree is deeper than a lane's stack.
    const uint4* refNodes;
    // Per primitive, the box of the caller's leaf that holds it (two 16-B words: lower.xyz upper.x | upper.yz 0 0):
    // what Tracer::surelyReached asks the reference's box routine about.
    const float4* leafBoxes;
    // rt_intersect_batch only: replay every ray that way (RT_OPT_STRICT_INTERSECT), not only those that
    // Tracer::settleTie picks out (equal distances; a hit the reference's box routine may have lost).
    int32_t strictIntersect;
};

struct DevCamera {
    f3 pos, fwd, up, left;
    f3 q;
    float qw, halfW, halfH;
};

struct DevFeat {
    int32_t shading, reflections, shadows, normalInterp, texMapping, accel, bilinear, transparency, jitter;
    int32_t shadingModel;
    uint32_t spp, shadowSamples;
    int32_t dof, envMap, glossy;
    uint32_t glossySamples;
    float glossyExponent, dofDistance, dofDiameter;
    uint32_t dofSamples;
    FastDiv divDof; // by max(dofSamples, 1)
    int32_t maxDepth;
    // renderImageWithMotionBlur (src/extra.cpp:121-211); taken only without depth of field (src/render.cpp:22-25)
    int32_t motion, motionSamples;
    float movement;
    // derived per frame
    uint32_t raysPerPixel, jitterN, camRays;
    // LCG jump constants: state after k draws = mul*seed + add
    uint32_t lensMul, lensAdd, mainMul, mainAdd;
};

struct DevFrame {
    int32_t width, height, tilesX, tilesY;
    FastDiv divTilesX;
    int32_t shard, nShard;
    uint32_t nWork; // local pixels = local tiles * 64
    int32_t packed; // 0: write Screen order, 1: write shard-packed tiles
    // The pixels (inclusive rectangle) outside of which no camera ray can enter the scene's box: the whole frame
    // where no such bound holds (queue-based renderer; filled in by the host with the frame's batch plan).
    int32_t candX0, candX1, candY0, candY1;
};

// Threads per block of every kernel that traverses (their LDS stacks are laid out for it).
#ifndef RT_BLOCK_THREADS
#define RT_BLOCK_THREADS 256
#endif
constexpr uint32_t kBlockThreads = RT_BLOCK_THREADS;
constexpr uint32_t kBlockWaves = kBlockThreads / 64u;
static_assert(kBlockThreads % 64u == 0 && kBlockThreads >= 64u && kBlockThreads <= 256u, "whole wavefronts, at most four");

// A counter that many waves update at once, alone in its 128-B line: atomics on one line are
// served one after the other, whatever their addresses within it.
struct alignas(128) HotCounter {
    uint32_t v;
    uint32_t pad[31];
};

struct DevCounters {
    unsigned long long rays, inner, tris;       // accumulated over the frame, all kernels
    unsigned long long sRays, sInner, sTris;    // the shadow kernel's share of them
    unsigned long long overflow;                // hits that found no slot in the pool (the host's bound was wrong: the frame is invalid)
    // work-queue heads, the node allocator, the number of queued camera rays, and the flags the first
    // wave to find a queue empty raises for the others ([0] path, [1] shadow); cleared before every batch
    HotCounter next, nextShadow, nodes, camQueued, dry[2];
    HotCounter redo; // light samples whose occluder the reference may not have come to (Tracer::surelyReached): entries of WfBuffers::redo
    HotCounter pathRedo; // camera rays whose path met a hit that takes the reference's own walk to decide: entries of the path kernel's hand-over list
    // counting launches only: iterations of the traversal loop summed over all 64 lanes of
    // every wave (divide by 64 for wave iterations), [0] inner [1] leaf [2] calls of run()
    unsigned long long loopIters[2][3]; // [0] path kernel, [1] shadow kernel
    unsigned long long topHits[2][3];   // inner steps at pair records below 1024 / 2048 / 4096
    unsigned long long cycles[2][4];    // shader clocks summed over waves: inner block, leaf block, run(), whole kernel
    unsigned long long maxCycles[2], busyWaves[2]; // longest wave; waves that traced at least one ray
    // counting launches only, not cleared otherwise: start and end clock of every wave of the two traversing kernels
    unsigned long long waveLog[2][8192][2];
};

// ----------------------------------------------------------------------------
// Sampler, src/sampler.h:13-46.  float(v)/4294967295.f: the divisor rounds to
// 2^32 as a float, so the division is an exact scaling by 2^-32.
// ----------------------------------------------------------------------------
RT_HD float next1d(uint32_t& state)
{
    state = state * 747796405u + 2891336453u;
    uint32_t v = state;
    v ^= v >> ((v >> 28u) + 4u);
    v *= 277803737u;
    v ^= v >> 22u;
    return static_cast<float>(v) * 2.3283064365386963e-10f;
}

// ----------------------------------------------------------------------------
// Textures, src/texture.cpp:12-95.  The reference indexes without wrap or
// clamp (out-of-range coordinates are undefined behaviour there); the final
// texel index is clamped into the image here.
// ----------------------------------------------------------------------------
__device__ __forceinline__ f3 texelAt(const DevScene& sc, const DevTex& t, float index)
{
    const int last = t.width * t.height - 1;
    int i = (index == index) ? (index >= 2147483520.0f ? last : (int)index) : 0;
    i = i < 0 ? 0 : (i > last ? last : i);
    const float* p = sc.texels + t.offset + 3u * (uint32_t)i;
    return { p[0], p[1], p[2] };
}
__device__ __forceinline__ f3 sampleTextureNearest(const DevScene& sc, const DevTex& im, f2 uv)
{
    float i = uv.x * (float)im.width;
    float j = (1.0f - uv.y) * (float)im.height;
    if (i == (float)im.width)
        i -= 1.0f;
    if (j == (float)im.height)
        j -= 1.0f;
    const float index = floorf(j) * (float)im.width + floorf(i);
    return texelAt(sc, im, index);
}
__device__ __forceinline__ f3 sampleTextureBilinear(const DevScene& sc, const DevTex& im, f2 uv)
{
    const float W = (float)im.width, H = (float)im.height;
    float x = uv.x * W;
    float y = (1.0f - uv.y) * H;
    if ((x == 0 && y == 0) || (x == W && y == 0) || (x == 0 && y == H) || (x == W && y == H) || x <= 0.5f
        || x >= W - 0.5f || y <= 0.5f || y >= H - 0.5f) {
        if (x >= W)
            x -= 1.0f;
        if (y >= H)
            y -= 1.0f;
        return texelAt(sc, im, floorf(y) * W + floorf(x));
    }
    const float i1 = roundf(x) - 0.5f, i2 = roundf(x) + 0.5f;
    const float j1 = roundf(y) - 0.5f, j2 = roundf(y) + 0.5f;
    const float distI1 = fabsf(x - i1), distI2 = fabsf(i2 - x);
    const f3 pix1 = texelAt(sc, im, floorf(j1) * W + floorf(i1));
    const f3 pix2 = texelAt(sc, im, floorf(j1) * W + floorf(i2));
    const f3 ix1 = distI2 * pix1 + distI1 * pix2;
    const f3 pix3 = texelAt(sc, im, floorf(j2) * W + floorf(i1));
    const f3 pix4 = texelAt(sc, im, floorf(j2) * W + floorf(i2));
    const f3 ix2 = distI2 * pix3 + distI1 * pix4;
    const float distJ1 = fabsf(y - j1), distJ2 = fabsf(j2 - y);
    return distJ2 * ix1 + distJ1 * ix2;
}
__device__ __forceinline__ f3 sampleTexture(const DevScene& sc, int tex, f2 uv, bool bilinear)
{
    const DevTex t = sc.texs[tex];
    return bilinear ? sampleTextureBilinear(sc, t, uv) : sampleTextureNearest(sc, t, uv);
}

// src/extra.cpp:378-459
__device__ __forceinline__ f3 sampleEnvironmentMap(const DevScene& sc, const DevFeat& ft, f3 d)
{
    if (!ft.envMap)
        return mk3(0.0f);
    int chosen;
    float u, v;
    const float ax = fabsf(d.x), ay = fabsf(d.y), az = fabsf(d.z);
    if (ax >= ay && ax >= az) {
        if (d.x > 0.0f) {
            chosen = 0, u = d.z, v = d.y;
        } else {
            chosen = 1, u = -d.z, v = d.y;
        }
    } else if (ay >= az) {
        if (d.y > 0.0f) {
            chosen = 2, u = -d.x, v = -d.z;
        } else {
            chosen = 3, u = -d.x, v = d.z;
        }
    } else {
        if (d.z > 0.0f) {
            chosen = 4, u = -d.x, v = d.y;
        } else {
            chosen = 5, u = d.x, v = d.y;
        }
    }
    const int tex = sc.env[chosen];
    if (tex < 0)
        return mk3(0.0f);
    const float biggest = fmaxf(ax, fmaxf(ay, az));
    u = (u / biggest + 1.0f) * 0.5f;
    v = (v / biggest + 1.0f) * 0.5f;
    return sampleTexture(sc, tex, { u, v }, ft.bilinear != 0);
}

// ----------------------------------------------------------------------------
// Lights, src/light.cpp:26-74
// ----------------------------------------------------------------------------
__device__ __forceinline__ f3 ld3(const float* p) { return { p[0], p[1], p[2] }; }

__device__ __forceinline__ void sampleSegmentLight(float s, const rt_light& l, f3& pos, f3& col)
{
    pos = mix3(ld3(l.v + 0), ld3(l.v + 3), s);
    col = mix3(ld3(l.v + 6), ld3(l.v + 9), s);
}
__device__ __forceinline__ void sampleParallelogramLight(f2 s, const rt_light& l, f3& pos, f3& col)
{
    const f3 v0 = ld3(l.v + 0), e1 = ld3(l.v + 3), e2 = ld3(l.v + 6);
    pos = v0 + s.x * e1 + s.y * e2;
    const float x = s.x, y = s.y;
    const float a0 = length(cross(e1 * x, e2 * y));
    const float a1 = length(cross(e1 * (1.0f - x), e2 * y));
    const float a2 = length(cross(e1 * x, e2 * (1.0f - y)));
    const float a3 = length(cross(e1 * (1.0f - x), e2 * (1.0f - y)));
    const float total = a0 + a1 + a2 + a3;
    const float bx = a0 / total, by = a1 / total, bz = a2 / total, bw = a3 / total;
    col = bw * ld3(l.v + 9) + bz * ld3(l.v + 12) + by * ld3(l.v + 15) + bx * ld3(l.v + 18);
}

// ----------------------------------------------------------------------------
// Shading, src/shading.cpp:29-213.  `kd` is sampleMaterialKd's result (texture
// or material colour), which depends only on the hit and is fetched once.
// ----------------------------------------------------------------------------
__device__ __forceinline__ f3 linearGradientSample(float ti)
{
    const float ts[5] = { 0.1f, 0.22f, 0.5f, 0.78f, 0.9f };
    const f3 cs[5] = {
        { 215.f / 256.f, 210.f / 256.f, 203.f / 256.f },
        { 250.f / 256.f, 250.f / 256.f, 240.f / 256.f },
        { 145.f / 256.f, 170.f / 256.f, 175.f / 256.f },
        { 255.f / 256.f, 250.f / 256.f, 205.f / 256.f },
        { 170.f / 256.f, 170.f / 256.f, 170.f / 256.f },
    };
    if (ti <= ts[0])
        return cs[0];
    if (ti >= ts[4])
        return cs[4];
    f3 c1 = cs[0], c2 = cs[1];
    float t1 = ts[0], t2 = ts[1];
#pragma unroll
    for (int i = 1; i < 4; i++) {
        if (ts[i] <= ti) {
            c1 = cs[i], c2 = cs[i + 1];
            t1 = ts[i], t2 = ts[i + 1];
        }
    }
    const float t = (ti - t1) / (t2 - t1);
    return mix3(c1, c2, t);
}

__device__ __forceinline__ f3 computeShading(const DevFeat& ft, f3 N, f3 kd, f3 ks, float shininess, f3 camDir,
    f3 lightDir, f3 lightColor)
{
    if (!ft.shading)
        return lightColor * kd;
    if (ft.shadingModel == 3) {
        const float cosTheta = dot(lightDir, N);
        return lightColor * linearGradientSample(cosTheta);
    }
    // Lambertian term, src/shading.cpp:60-74
    const float nl = dot(N, lightDir);
    const f3 diffuse = (nl < 0.0f) ? mk3(0.0f) : lightColor * kd * nl;
    if (ft.shadingModel == 0)
        return diffuse;
    float d;
    if (ft.shadingModel == 1) {
        const f3 reflected = lightDir - N * dot(N, lightDir) * 2.0f; // glm::reflect(l, n)
        d = dot(reflected, camDir);
    } else {
        const f3 Hv = normalize(lightDir + camDir);
        d = dot(N, Hv);
    }
    if (d < 0.0f)
        return diffuse;
    const f3 specular = lightColor * ks * powLikeHost(d, shininess); // the C library's result, bit for bit (rt_pow.hpp)
    return diffuse + specular;
}

// ----------------------------------------------------------------------------
// Intersection arithmetic
// ----------------------------------------------------------------------------
struct TriRec {
    f3 n;
    float D;
    f3 E0, v0, E1, v1, E2, v2;
};

__device__ __forceinline__ TriRec loadTri(const float4* __restrict__ tris, uint32_t i)
{
    const float4* p = tris + 6u * i;
    const float4 a = p[0], b = p[1], c = p[2], d = p[3], e = p[4], f = p[5];
    TriRec r;
    r.n = { a.x, a.y, a.z };
    r.D = a.w;
    r.E0 = { b.x, b.y, b.z };
    r.v0 = { b.w, c.x, c.y };
    r.E1 = { c.z, c.w, d.x };
    r.v1 = { d.y, d.z, d.w };
    r.E2 = { e.x, e.y, e.z };
    r.v2 = { e.w, f.x, f.y };
    return r;
}

// intersectRayWithTriangle (src/intersect.h:12; body only in the prebuilt
// archive): plane hit t = (D - dot(o,n)) / dot(d,n) accepted iff 0 <= t <= tmax
// (NaN rejects), then three inclusive edge tests dot(cross(edge, n), p - v) >= 0.
// cross(edge, n), n and D do not depend on the ray and come from the record,
// computed with the same operations in the same order.
__device__ __forceinline__ bool triangleTest(const TriRec& r, f3 o, f3 d, float tmax, float& tOut)
{
    const float t = (r.D - dot(o, r.n)) / dot(d, r.n);
    if (!(t >= 0.0f && t <= tmax))
        return false;
    const f3 p = o + d * t;
    if (!(dot(r.E0, p - r.v0) >= 0.0f))
        return false;
    if (!(dot(r.E1, p - r.v1) >= 0.0f))
        return false;
    if (!(dot(r.E2, p - r.v2) >= 0.0f))
        return false;
    tOut = t;
    return true;
}

// The same test reading the record in two steps: the 16-B plane first, the
// remaining 80 B only for rays whose plane hit lies inside [0, tmax].
__device__ __forceinline__ bool triangleTestLazy(const float4* __restrict__ p4, f3 o, f3 d, float tmax, float& tOut)
{
    const float4 a = p4[0];
    const f3 n = { a.x, a.y, a.z };
    const float t = (a.w - dot(o, n)) / dot(d, n);
    if (!(t >= 0.0f && t <= tmax))
        return false;
    const float4 b = p4[1], c = p4[2], e = p4[3], f = p4[4], g = p4[5];
    const f3 p = o + d * t;
    if (!(dot(mk3(b.x, b.y, b.z), p - mk3(b.w, c.x, c.y)) >= 0.0f))
        return false;
    if (!(dot(mk3(c.z, c.w, e.x), p - mk3(e.y, e.z, e.w)) >= 0.0f))
        return false;
    if (!(dot(mk3(f.x, f.y, f.z), p - mk3(f.w, g.x, g.y)) >= 0.0f))
        return false;
    tOut = t;
    return true;
}

// intersectRayWithShape(Sphere) (src/intersect.h:14)
__device__ __forceinline__ bool sphereTest(const DevSphere& s, f3 o, f3 d, float tmax, float& tOut)
{
    const f3 oc = o - ld3(s.center);
    const float B = 2.0f * dot(d, oc);
    const float C = dot(oc, oc) - s.radius * s.radius;
    const float disc = B * B - 4.0f * C;
    const float sq = sqrtf(disc);
    float t0 = (-B - sq) / 2.0f;
    float t1 = (-B + sq) / 2.0f;
    if (t0 < 0.0f)
        t0 = 3.402823466e+38f;
    if (t1 < 0.0f)
        t1 = 3.402823466e+38f;
    const float t = (t1 < t0) ? t1 : t0; // std::min(t0, t1)
    if (!(t >= 0.0f && t < tmax))
        return false;
    tOut = t;
    return true;
}

// ----------------------------------------------------------------------------
// Motion blur (src/extra.cpp:121-211): every pixel sample draws a time and renders a copy of the
// scene whose vertices and sphere centres have been moved by splineMat (src/extra.cpp:546-567).
// A Mover holds what that move needs for one time; moved() is the move of one position, operation
// for operation (the Bezier sum, glm::translate's column, the mat4 * vec4 product and the division
// by w), so a moved vertex has the bits the reference's copy of the scene holds.
// The control points are offset + position and the weights sum to one, so the move is, up to
// rounding, position -> 2 * position + shift: the copy is twice the size of the scene.
// ----------------------------------------------------------------------------
struct Mover {
    float w0, w1, w2, w3, w4; // weights of the five control points at this time
    f3 c0, c1, c2, c3, c4;    // the control points' offsets, times `movement`
    f3 shift;                 // sum of weight * offset: where the origin goes, to rounding
};

__device__ __forceinline__ Mover makeMover(float t, float movement)
{
    Mover m;
    m.c0 = mk3(0.0f, 0.0f, 0.0f) * movement;
    m.c1 = mk3(0.0f, 1.0f, 1.0f) * movement;
    m.c2 = mk3(1.0f, 1.0f, -1.0f) * movement;
    m.c3 = mk3(1.0f, 0.0f, 0.0f) * movement;
    m.c4 = mk3(1.5f, 1.0f, 2.0f) * movement;
    const float oneMinusT = 1.0f - t;
    const float oneMinusTSquared = oneMinusT * oneMinusT;
    const float tSquared = t * t;
    const float tCubed = tSquared * t;
    m.w0 = oneMinusTSquared * oneMinusTSquared;
    m.w1 = 4.0f * oneMinusTSquared * oneMinusT * t;
    m.w2 = 6.0f * oneMinusTSquared * tSquared;
    m.w3 = 4.0f * oneMinusT * tCubed;
    m.w4 = tSquared * tSquared;
    m.shift = (m.w0 * m.c0) + (m.w1 * m.c1) + (m.w2 * m.c2) + (m.w3 * m.c3) + (m.w4 * m.c4);
    return m;
}

__device__ __forceinline__ f3 moved(const Mover& m, f3 v)
{
    const f3 p0 = m.c0 + v, p1 = m.c1 + v, p2 = m.c2 + v, p3 = m.c3 + v, p4 = m.c4 + v;
    const f3 b = (m.w0 * p0) + (m.w1 * p1) + (m.w2 * p2) + (m.w3 * p3) + (m.w4 * p4);
    // glm::translate(mat4(1), b), ext/matrix_transform.inl:13: column 3 = m[0]*b.x + m[1]*b.y + m[2]*b.z + m[3]
    const float tx = ((1.0f * b.x + 0.0f * b.y) + 0.0f * b.z) + 0.0f;
    const float ty = ((0.0f * b.x + 1.0f * b.y) + 0.0f * b.z) + 0.0f;
    const float tz = ((0.0f * b.x + 0.0f * b.y) + 1.0f * b.z) + 0.0f;
    const float tw = ((0.0f * b.x + 0.0f * b.y) + 0.0f * b.z) + 1.0f;
    // mat4 * vec4(v, 1), detail/type_mat4x4.inl:561-572: (m[0]*v.x + m[1]*v.y) + (m[2]*v.z + m[3]*1)
    const float x = (1.0f * v.x + 0.0f * v.y) + (0.0f * v.z + tx * 1.0f);
    const float y = (0.0f * v.x + 1.0f * v.y) + (0.0f * v.z + ty * 1.0f);
    const float z = (0.0f * v.x + 0.0f * v.y) + (1.0f * v.z + tz * 1.0f);
    const float w = (0.0f * v.x + 0.0f * v.y) + (0.0f * v.z + tw * 1.0f);
    return { x / w, y / w, z / w };
}

// The plane and edge terms of a triangle from its three vertices, as the scene upload computes
// them for the record (and as the reference's triangle routine computes them per test).
__device__ __forceinline__ TriRec triFromVertices(f3 v0, f3 v1, f3 v2)
{
    TriRec r;
    const f3 c = cross(v1 - v0, v2 - v0);
    r.n = c * (1.0f / sqrtf(dot(c, c)));
    r.D = dot(v0, r.n);
    r.E0 = cross(v2 - v0, r.n);
    r.E1 = cross(v0 - v1, r.n);
    r.E2 = cross(v1 - v2, r.n);
    r.v0 = v0, r.v1 = v1, r.v2 = v2;
    return r;
}

// Conservative slab test.  The reference's box routine divides per axis and
// ignores an axis whose direction component is zero; the result there only
// decides which nodes are visited and in which order, never the hit itself.
// Here the slabs use one subtract and one multiply by 1/d per plane and the
// interval is widened by a few ulp, so a box is never skipped that the exact
// test would have entered.  Returns the entry distance clamped to 0.
constexpr float kPadLo = 1.0f - 4.76837158e-7f; // 1 - 2^-21
constexpr float kPadHi = 1.0f + 4.76837158e-7f; // 1 + 2^-21

// 1/d for the box tests: an infinite one (a direction component of zero) is held to +-1e18.  The ray then
// counts as creeping along that axis: planes on either side of it lie at -+(distance) * 1e18, far beyond any real t,
// so a box the ray's plane does not pass through is turned away as it should be; a plane the origin lies exactly IN
// lies at zero, where infinity would have made "0 * inf" of it and lost the ray, and the additive widening of the
// tests (proportional to |1/d|) takes it from there.  Leaving such an axis out altogether, as the reference does,
// is conservative too, but lets a ray that runs along an axis into every box it passes above or below: a single
// lane then walks thousands of boxes while its kernel waits (measured: seven such rays in the headline frame
// doubled the path kernel's time).
__device__ __forceinline__ f3 boundedRcp(f3 inv)
{
    return { __builtin_amdgcn_fmed3f(inv.x, -1.0e18f, 1.0e18f), __builtin_amdgcn_fmed3f(inv.y, -1.0e18f, 1.0e18f),
        __builtin_amdgcn_fmed3f(inv.z, -1.0e18f, 1.0e18f) };
}

__device__ __forceinline__ bool boxTest(f3 lo, f3 hi, f3 o, f3 invIn, float tmax, float& entry)
{
    const f3 inv = boundedRcp(invIn);
    const float ax = (lo.x - o.x) * inv.x, bx = (hi.x - o.x) * inv.x;
    const float ay = (lo.y - o.y) * inv.y, by = (hi.y - o.y) * inv.y;
    const float az = (lo.z - o.z) * inv.z, bz = (hi.z - o.z) * inv.z;
    // additive part of the widening: 2^-20 of the coordinates involved, as a distance along the ray
    const float k = 9.5367431640625e-7f;
    const float px = fabsf(inv.x) * ((fabsf(lo.x) + fabsf(hi.x) + fabsf(o.x)) * k);
    const float py = fabsf(inv.y) * ((fabsf(lo.y) + fabsf(hi.y) + fabsf(o.y)) * k);
    const float pz = fabsf(inv.z) * ((fabsf(lo.z) + fabsf(hi.z) + fabsf(o.z)) * k);
    const float tIn = fmaxf(fmaxf(fminf(ax, bx) - px, fminf(ay, by) - py), fmaxf(fminf(az, bz) - pz, 0.0f)) * kPadLo;
    const float tOut = fminf(fminf(fmaxf(ax, bx) + px, fmaxf(ay, by) + py), fmaxf(az, bz) + pz) * kPadHi;
    entry = tIn;
    return tIn <= tOut && tIn <= tmax;
}


// intersectRayWithShape(AxisAlignedBox) the way the reference computes it (src/intersect.h:16;
// body only in the prebuilt archive): slabs by true division, a zero direction component
// leaves its slab out, an origin inside the box reports the distance to the exit.  Neither
// conservative nor used for culling here: it exists to replay the reference's order of tests
// (Tracer::settleTie).  `t` comes in as the limit and goes out as the reported distance.
__device__ __forceinline__ bool refBoxTest(const float lo[3], const float hi[3], const float o[3], const float d[3], float& t)
{
    float tIn = 0.0f, tOut = 0.0f;
#pragma unroll
    for (int a = 0; a < 3; a++) {
        float tLo, tHi;
        if (d[a] == 0.0f) {
            tHi = 3.402823466e+38f;
            tLo = 1.17549435e-38f;
        } else {
            tLo = (lo[a] - o[a]) / d[a];
            tHi = (hi[a] - o[a]) / d[a];
        }
        const float mn = (tHi < tLo) ? tHi : tLo; // std::min(tLo, tHi)
        const float mx = (tLo < tHi) ? tHi : tLo; // std::max(tLo, tHi)
        tIn = (a == 0) ? mn : ((tIn < mn) ? mn : tIn);   // std::max(tIn, mn)
        tOut = (a == 0) ? mx : ((mx < tOut) ? mx : tOut); // std::min(tOut, mx)
    }
    if (tIn < 0.0f) {
        if (tOut <= 0.0f)
            return false;
        tIn = tOut;
        tOut = 3.402823466e+38f;
    }
    if (tIn > tOut || tIn < 0.0f || tIn > t)
        return false;
    t = tIn;
    return true;
}

} // namespace rt
