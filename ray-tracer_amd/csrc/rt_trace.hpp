// rt_trace.hpp -- the shared BVH traversal: every ray of every kernel (camera,
// reflection, pass-through, glossy, shadow) runs through this one while-while
// loop.  Per-lane state is a RayState; the short traversal stack lives in LDS,
// one column per lane (entry e of lane t at stackBase[e * blockDim.x + t], so
// the 64 lanes of a wave always hit 64 different banks).
#pragma once

#include "rt_device.hpp"

#ifndef RT_INNER_STEPS
#define RT_INNER_STEPS 4 // sibling pairs a lane may visit between two votes of the traversal loop
#endif

#ifndef RT_STACK_TOS
#define RT_STACK_TOS 0 // the top entry of a lane's traversal stack lives in a register (see Tracer::pop)
#endif
#ifndef RT_ROLLED_STEPS
#define RT_ROLLED_STEPS 0 // the inner steps between two votes as a loop instead of unrolled code
#endif
#ifndef RT_PARK_IN_BLOCK
#define RT_PARK_IN_BLOCK 0 // a lane that reaches a leaf between two votes parks it there and then
#endif

namespace rt {

enum RayKind : int { K_RADIANCE = 0, K_SHADOW_ANY = 1, K_SHADOW_CLOSEST = 2 };

constexpr int kHitNone = -1;
constexpr float kFltMax = 3.402823466e+38f;
constexpr float kTwoPi = 6.28318530717958647692528676655900576f;

typedef __attribute__((address_space(3))) uint32_t* LdsWords;

struct RayState {
    f3 o, d;
    // Slab arithmetic against the quantised boxes.  A plane q (16-bit grid position)
    // is turned into the float 2^23 + q by gluing the exponent byte 0x4B on top of its
    // two bytes, so its distance along the ray is one fma: t = (2^23 + q) * A + B with
    // B = (gridMin - o) / d - 2^23 * A, widened by the rounding bound of that
    // expression (Bn for the plane the ray enters through, Bf for the one it leaves
    // through).  selN / selF are the byte selectors that pick the entering / leaving
    // half of a box dword (lo | hi << 16) for this ray's direction signs.
    f3 A, Bn, Bf;
    uint32_t selNx, selNy, selNz, selFx, selFy, selFz;
    float tmax, slen; // current closest t; the ray's original length (shadow rays)
    int hitPrim;      // primitive index, kHitNone, or -2 - sphere index
    int kind;
    uint32_t cur;     // node reference to visit next, kRefNone when the stack has run dry
    uint32_t pend;    // parked leaf reference whose primitives are still to be tested, 0 = none
    uint32_t li, le;  // next primitive of the parked leaf and its end, as indices into the float4 array (6 per primitive)
    LdsWords top;     // first free entry of this lane's traversal stack
    uint32_t tos;     // the stack's top entry (RT_STACK_TOS): what the next pop returns
    bool active;      // still inside the traversal loop
    bool tie;         // two primitives were met at exactly the same distance: see Tracer::settleTie
    uint32_t nInner, nTris;
    uint32_t itInner, itLeaf, itRuns; // counting builds: loop iterations this lane sat through
    uint32_t topHits[3];              // counting builds: inner steps at pair records below 1024 / 2048 / 4096
    unsigned long long cycInner, cycLeaf, cycRun; // counting builds: shader clocks spent in the two blocks and in run() as a whole
    Mover mv; // motion blur only: the move of the scene copy this ray is traced in
};

__device__ __forceinline__ f3 rcp3(f3 d)
{
    return { __builtin_amdgcn_rcpf(d.x), __builtin_amdgcn_rcpf(d.y), __builtin_amdgcn_rcpf(d.z) };
}

// BVHInterface::intersect over the caller's nodes, test for test as src/bvh.cpp:245-316 runs it
// (see Tracer::settleTie).  Out of line on purpose: it runs for one ray in millions and must not
// weigh on the register budget of the loops that call it.  `stk` is the calling lane's column of
// the LDS stack (entries `stride` words apart), empty by now; returns (closest t as bits, primitive or kHitNone).
// MOVED (with a Mover): the nodes and triangles are those of the moved copy of the scene: the move is monotone per
// coordinate, so the box of a set of moved vertices is the moved box, and the copy's tree has the rest
// tree's shape wherever the centroids keep their order (they do but for pairs a rounding apart).
// (A function of its own per case: what the callers of the plain one must keep out of its way is what IT uses.)
template <bool MOVED = false>
static __device__ __attribute__((noinline)) uint2 replayReference(const uint4* __restrict__ nodes, const float4* __restrict__ tris, LdsWords stk,
    uint32_t stride, float ox, float oy, float oz, float dx, float dy, float dz, float tmax, const Mover* mv = nullptr)
{
    const float o[3] = { ox, oy, oz }, d[3] = { dx, dy, dz };
    const f3 O = { ox, oy, oz }, D = { dx, dy, dz };
    int hitPrim = kHitNone;
    auto box = [&](uint32_t node, float& t) {
        const uint4 a = nodes[2u * node], b = nodes[2u * node + 1u];
        f3 l3 = { __uint_as_float(a.x), __uint_as_float(a.y), __uint_as_float(a.z) };
        f3 h3 = { __uint_as_float(a.w), __uint_as_float(b.x), __uint_as_float(b.y) };
        if (MOVED) {
            l3 = moved(*mv, l3);
            h3 = moved(*mv, h3);
        }
        const float lo[3] = { l3.x, l3.y, l3.z };
        const float hi[3] = { h3.x, h3.y, h3.z };
        t = kFltMax;
        return refBoxTest(lo, hi, o, d, t);
    };
    float tRoot;
    if (!box(0u, tRoot))
        return make_uint2(__float_as_uint(tmax), (uint32_t)kHitNone);
    uint32_t depth = 0;
    stk[0] = 0u;
    depth = 1;
    while (depth != 0u) {
        const uint32_t node = stk[--depth * stride];
        const uint4 w = nodes[2u * node + 1u];
        if (w.z & RT_LEAF_BIT) {
            const uint32_t first = w.z & ~RT_LEAF_BIT;
            for (uint32_t i = first; i < first + w.w; i++) {
                float t;
                bool hit;
                if (MOVED) {
                    const TriRec rest = loadTri(tris, i);
                    hit = triangleTest(triFromVertices(moved(*mv, rest.v0), moved(*mv, rest.v1), moved(*mv, rest.v2)), O, D, tmax, t);
                } else {
                    hit = triangleTestLazy(tris + 6u * i, O, D, tmax, t);
                }
                if (hit) {
                    tmax = t;
                    hitPrim = (int)i;
                }
            }
            continue;
        }
        float tL, tR;
        const bool hL = box(w.z, tL), hR = box(w.w, tR);
        if (hL && hR) {
            const bool leftFirst = tL < tR;
            stk[depth++ * stride] = leftFirst ? w.w : w.z;
            stk[depth++ * stride] = leftFirst ? w.z : w.w;
        } else if (hL) {
            stk[depth++ * stride] = w.z;
        } else if (hR) {
            stk[depth++ * stride] = w.w;
        }
    }
    return make_uint2(__float_as_uint(tmax), (uint32_t)hitPrim);
}

// MOTION: every ray carries a Mover (RayState::mv) and meets the moved copy of the scene: the triangle
// and sphere tests run on moved positions, computed on the spot, while the boxes stay the rest scene's.
// Since the move is position -> 2 * position + shift up to a few roundings, a box test against the
// moved box is a test of the ray (origin - shift) / 2, direction / 2 -- same distances along it --
// against the rest box, widened by what those roundings can amount to.
template <bool STATS, bool MOTION = false>
struct Tracer {
    const DevScene& sc;
    int accel;
    LdsWords stk; // this lane's column of the LDS stack
    static constexpr uint32_t stride = kBlockThreads; // every traversing kernel runs blocks of this size

    // The bottom entry of every stack holds kRefNone (begin()), so a pop needs no
    // emptiness check: the last pop of a traversal returns "no node".
    // With RT_STACK_TOS the top entry is held in a register: a pop hands it out at once and the LDS read
    // that fetches the entry below is only waited for at the lane's next stack operation, a box test or
    // more later, instead of in the middle of the step (every step of a wave has some lane popping).
    __device__ __forceinline__ uint32_t pop(RayState& r) const
    {
#if RT_STACK_TOS
        const uint32_t ref = r.tos;
        r.top -= stride;
        r.tos = *r.top;
        return ref;
#else
        r.top -= stride;
        return *r.top;
#endif
    }
    __device__ __forceinline__ void push(RayState& r, uint32_t ref) const
    {
#if RT_STACK_TOS
        *r.top = r.tos;
        r.top += stride;
        r.tos = ref;
#else
        *r.top = ref;
        r.top += stride;
#endif
    }

    __device__ __forceinline__ void begin(RayState& r, f3 o, f3 d, float tmax, int kind) const
    {
        r.o = o;
        r.d = d;
        r.tmax = tmax;
        r.slen = tmax;
        r.kind = kind;
        r.hitPrim = kHitNone;
#if RT_STACK_TOS
        *stk = kRefNone; // in memory below the register-held top, so that the pop of the last entry reads "no node" again
        r.top = stk + stride;
        r.tos = kRefNone;
#else
        r.top = stk;
        push(r, kRefNone);
#endif
        r.cur = kRefNone;
        r.pend = 0u;
        r.active = false;
        r.tie = false;
    }

    // The root box test of src/bvh.cpp:253-259: does the ray enter the scene's box at all?
    __device__ __forceinline__ bool entersScene(f3 o, f3 inv, float pad = 0.0f) const
    {
        float e;
        const f3 lo = { sc.rootLo[0] - pad, sc.rootLo[1] - pad, sc.rootLo[2] - pad };
        const f3 hi = { sc.rootHi[0] + pad, sc.rootHi[1] + pad, sc.rootHi[2] + pad };
        return sc.nPrims != 0 && boxTest(lo, hi, o, inv, kFltMax, e);
    }

    // Motion blur: how far, in the rest scene's space, a moved position can lie from 2 * position + shift
    // halved back.  The move is some twenty roundings of numbers no larger than the scene's largest
    // coordinate plus twice the movement; 2^-18 of that bound leaves a factor of ten.
    __device__ __forceinline__ float motionPad(const RayState& r) const
    {
        float big = 0.0f;
#pragma unroll
        for (int a = 0; a < 3; a++)
            big = fmaxf(big, fmaxf(fabsf(sc.rootLo[a]), fabsf(sc.rootHi[a])));
        big += 2.0f * fmaxf(fmaxf(fabsf(r.mv.c4.x), fabsf(r.mv.c4.y)), fabsf(r.mv.c4.z));
        return big * 3.814697265625e-6f; // 2^-18
    }

    // Start the traversal at the root and set up the slab arithmetic of this ray.
    __device__ __forceinline__ void enter(RayState& r, f3 invIn, f3 o, float pad = 0.0f) const
    {
        setCur(r, sc.rootRef);
        const f3 inv = boundedRcp(invIn); // (a zero direction component: see there)
        // t(q) = ((gridMin + q*cell) - o) * inv = q*A + B.  Rounding bound of the fused
        // form: (|q*A| + |B|) * 2^-22 for the inputs plus half an ulp of the result;
        // (extent*|inv| + |B|) * 2^-20 covers it with room to spare.  Moving 2^23 * A
        // from the product into B costs one more rounding, of at most half an ulp of
        // 2^23 * |A| + |B|, i.e. |A| / 2 on top of what the first term already covers.
        const f3 gmin = { sc.gridMin[0], sc.gridMin[1], sc.gridMin[2] };
        const f3 cell = { sc.gridCell[0], sc.gridCell[1], sc.gridCell[2] };
        f3 A = cell * inv;
        const f3 B = (gmin - o) * inv;
        const float k = 9.5367431640625e-7f; // 2^-20
        f3 err = { (65535.0f * fabsf(A.x) + fabsf(B.x)) * k + 0.5005f * fabsf(A.x),
            (65535.0f * fabsf(A.y) + fabsf(B.y)) * k + 0.5005f * fabsf(A.y),
            (65535.0f * fabsf(A.z) + fabsf(B.z)) * k + 0.5005f * fabsf(A.z) };
        if (MOTION) // planes `pad` further out on either side
            err = err + mk3(fabsf(inv.x), fabsf(inv.y), fabsf(inv.z)) * (pad * 1.001f);
        const f3 C = 8388608.0f * A;
        f3 Bn = (B - err) - C, Bf = (B + err) - C;
        // What is still out of range after that (a direction that is not a number): the slab places no bound.
        const bool okx = fabsf(A.x) < 1.0e30f && fabsf(B.x) < 1.0e30f, oky = fabsf(A.y) < 1.0e30f && fabsf(B.y) < 1.0e30f,
                   okz = fabsf(A.z) < 1.0e30f && fabsf(B.z) < 1.0e30f;
        r.A = { okx ? A.x : 0.0f, oky ? A.y : 0.0f, okz ? A.z : 0.0f };
        r.Bn = { okx ? Bn.x : -kFltMax, oky ? Bn.y : -kFltMax, okz ? Bn.z : -kFltMax };
        r.Bf = { okx ? Bf.x : kFltMax, oky ? Bf.y : kFltMax, okz ? Bf.z : kFltMax };
        const uint32_t selLo = 0x030c0504u, selHi = 0x030c0706u; // {0x4B, 0x00, w.b1, w.b0} / {0x4B, 0x00, w.b3, w.b2}
        r.selNx = inv.x < 0.0f ? selHi : selLo, r.selFx = inv.x < 0.0f ? selLo : selHi;
        r.selNy = inv.y < 0.0f ? selHi : selLo, r.selFy = inv.y < 0.0f ? selLo : selHi;
        r.selNz = inv.z < 0.0f ? selHi : selLo, r.selFz = inv.z < 0.0f ? selLo : selHi;
    }

    // One BVHInterface::intersect call begins (src/bvh.cpp:337-358).  The root
    // box test happens here, so a ray that misses the whole scene never enters
    // the traversal loop.
    __device__ __forceinline__ void launch(RayState& r, f3 o, f3 d, float tmax, int kind) const
    {
        begin(r, o, d, tmax, kind);
        if (accel) {
            if (MOTION) {
                const f3 inv = rcp3(d) * 2.0f;
                const f3 oRest = (o - r.mv.shift) * 0.5f;
                const float pad = motionPad(r);
                if (entersScene(oRest, inv, pad))
                    enter(r, inv, oRest, pad);
            } else {
                const f3 inv = rcp3(d);
                if (entersScene(o, inv))
                    enter(r, inv, o);
            }
        } else {
            // enableAccelStructure off: every primitive in order (src/bvh.cpp:344-351)
            for (uint32_t i = 0; i < sc.nPrims; i++) {
                TriRec tr = loadTri(sc.tris, i);
                if (MOTION)
                    tr = triFromVertices(moved(r.mv, tr.v0), moved(r.mv, tr.v1), moved(r.mv, tr.v2));
                float t;
                if (STATS)
                    r.nTris++;
                if (triangleTest(tr, r.o, r.d, r.tmax, t)) {
                    r.tmax = t;
                    r.hitPrim = (int)i;
                }
            }
        }
    }

    // One BVHInterface::intersect call answered at once by the reference's own walk over the caller's nodes
    // (replayReference), not by the traversal loop: for the rays of a path one of whose hits takes that walk to decide
    // (settleTie), followed to its end by a kernel of its own.  Needs the acceleration structure on and the caller's nodes.
    __device__ __forceinline__ void launchExact(RayState& r, f3 o, f3 d, float tmax, int kind) const
    {
        begin(r, o, d, tmax, kind);
        const uint2 res = replayReference<MOTION>(sc.refNodes, sc.tris, stk, stride, o.x, o.y, o.z, d.x, d.y, d.z, tmax, MOTION ? &r.mv : nullptr);
        r.tmax = __uint_as_float(res.x);
        r.hitPrim = (int)res.y;
    }

    // Does the finished traversal's answer take the reference's own walk to decide (see settleTie, which makes that
    // walk on the spot)?  Clears the mark either way.
    __device__ __forceinline__ bool doubtful(RayState& r) const
    {
        const bool d = accel && sc.refNodes != nullptr && (r.tie || (sc.leafBoxes != nullptr && r.hitPrim >= 0 && !surelyReached(r)));
        r.tie = false;
        return d;
    }

    // The same for a ray already known to pass entersScene() (acceleration structure on).
    __device__ __forceinline__ void launchInside(RayState& r, f3 o, f3 d, float tmax, int kind) const
    {
        begin(r, o, d, tmax, kind);
        enter(r, rcp3(d), o);
    }

    // Conservative slab test against one quantised child box (three dwords, lo | hi << 16
    // per axis).  One byte permute per plane picks the half the ray enters (leaves)
    // through and makes it a float; no min/max per axis is needed.
    __device__ __forceinline__ bool qboxTest(uint32_t wx, uint32_t wy, uint32_t wz, const RayState& r, float& entry) const
    {
        const uint32_t K = 0x4b000000u;
        const float nx = __builtin_fmaf(__uint_as_float(__builtin_amdgcn_perm(wx, K, r.selNx)), r.A.x, r.Bn.x);
        const float fx = __builtin_fmaf(__uint_as_float(__builtin_amdgcn_perm(wx, K, r.selFx)), r.A.x, r.Bf.x);
        const float ny = __builtin_fmaf(__uint_as_float(__builtin_amdgcn_perm(wy, K, r.selNy)), r.A.y, r.Bn.y);
        const float fy = __builtin_fmaf(__uint_as_float(__builtin_amdgcn_perm(wy, K, r.selFy)), r.A.y, r.Bf.y);
        const float nz = __builtin_fmaf(__uint_as_float(__builtin_amdgcn_perm(wz, K, r.selNz)), r.A.z, r.Bn.z);
        const float fz = __builtin_fmaf(__uint_as_float(__builtin_amdgcn_perm(wz, K, r.selFz)), r.A.z, r.Bf.z);
        // The ray meets the box between max(tIn, 0) and min(tOut, ray.t): tIn <= tOut, tIn <= ray.t and
        // 0 <= tOut, asked as one sign (the sign of a difference of two floats is that of their comparison).
        // Comparisons, minima and permutes issue at half the rate of additions and multiply-adds on gfx950
        // (tools/valu_probe.hip), so two subtractions and a three-way minimum are cheaper than two compares
        // and a clamp.  `entry` only orders the two children; it may be negative (origin inside the box).
        const float tIn = fmaxf(fmaxf(nx, ny), nz);
        const float tOut = fminf(fminf(fx, fy), fz);
        entry = tIn;
        return fminf(fminf(tOut - tIn, r.tmax - tIn), tOut) >= 0.0f;
    }

    // Make `ref` the node to visit next.
    __device__ __forceinline__ void setCur(RayState& r, uint32_t ref) const
    {
        r.cur = ref;
        r.active = true;
    }

    // Shared traversal loop.  A lane that reaches a leaf does not test it on the
    // spot: it parks the leaf in `pend` and goes on down the tree (pops the next
    // node), so that most lanes can still take inner steps while they already
    // hold triangle work; only with a second leaf in hand does a lane have to
    // wait.  Per iteration the wave runs ONE of the two code blocks -- up to
    // RT_INNER_STEPS inner steps (one sibling pair each) for every lane that has
    // a node, or one triangle test for every lane that has a parked leaf --
    // whichever more lanes are ready for (ballot vote; a few steps per vote
    // spread the cost of the vote itself).  Along one ray the near child still goes first
    // and the primitives of a leaf are tested in order; a parked leaf only sees
    // its ray's t a few steps later than it would have, which changes how many
    // boxes are culled but not which hit is the closest.
    // `knobs`: bits 0-7 minActive -- leave as soon as fewer lanes than this still have a
    // ray in flight (0 = run every ray to completion); bits 8-15 and 16-23 the two
    // switch points between inner steps and triangle tests described in the loop.
    __device__ __forceinline__ void run(RayState& r, int knobs) const
    {
        const int minActive = knobs & 0xff, voteScale = (knobs >> 8) & 0xff, voteBias = (knobs >> 16) & 0xff;
        long long tRun = 0;
        if (STATS) {
            r.itRuns++;
            tRun = clock64();
        }
        for (;;) {
            park(r);
            // every lane with a ray in flight can take an inner step, a triangle test, or both:
            // a lane whose node is a second leaf holds a parked one
            const bool canInner = r.cur < kRefNone; // leaf refs have the top bit set, kRefNone is the largest inner value
            const bool canLeaf = r.pend != 0u;
            const unsigned long long mInner = __ballot(canInner), mLeaf = __ballot(canLeaf);
            const int nInner = __popcll(mInner), nLeaf = __popcll(mLeaf), alive = __popcll(mInner | mLeaf);
            if (alive == 0 || alive < minActive)
                break;
            // vote: triangles when the lanes holding a leaf outnumber (scaled, biased) those with a node
            bool leafMode = 16 * nLeaf > voteScale * nInner + 16 * (voteBias - 32);
            if (nInner == 0)
                leafMode = true;
            if (nLeaf == 0)
                leafMode = false;
            long long t0 = 0;
            if (STATS)
                t0 = clock64();
            if (!leafMode) {
                if (canInner)
                    innerStep(r);
#if RT_ROLLED_STEPS
#pragma unroll 1
#else
#pragma unroll
#endif
                for (int k = 1; k < RT_INNER_STEPS; k++) {
#if RT_PARK_IN_BLOCK
                    park(r);
#endif
                    if (r.cur < kRefNone)
                        innerStep(r);
                }
                if (STATS) {
                    r.itInner++;
                    r.cycInner += (unsigned long long)(clock64() - t0);
                }
            } else {
                if (canLeaf)
                    leafStep(r);
                if (STATS) {
                    r.itLeaf++;
                    r.cycLeaf += (unsigned long long)(clock64() - t0);
                }
            }
        }
        if (STATS)
            r.cycRun += (unsigned long long)(clock64() - tRun);
        r.active = r.pend != 0u || r.cur != kRefNone;
    }

    // A lane whose node is a leaf, with its slot for one free: park the leaf, take the next node.
    __device__ __forceinline__ void park(RayState& r) const
    {
        if ((int32_t)r.cur < 0 && r.pend == 0u) {
            const uint32_t count = (r.cur >> 27) & 15u, first = r.cur & kLeafOffsetMask;
            r.pend = count ? r.cur : 0u; // an empty leaf is dropped here
            r.li = (first << 2) + (first << 1);
            r.le = r.li + (count << 2) + (count << 1);
            r.cur = pop(r);
        }
    }

    // One sibling-pair record (32 B): test both children, descend into the nearer.
    __device__ __forceinline__ void innerStep(RayState& r) const
    {
        const uint4* p = reinterpret_cast<const uint4*>(reinterpret_cast<const char*>(sc.pairs) + r.cur);
        const uint4 w0 = p[0], w1 = p[1];
        if (STATS) {
            r.nInner++;
            r.topHits[0] += r.cur < 1024u * kPairBytes, r.topHits[1] += r.cur < 2048u * kPairBytes, r.topHits[2] += r.cur < 4096u * kPairBytes;
        }
#ifdef RT_PROBE_EXTRA_LOADS // sensitivity probes (tools/sensitivity.sh): work added to every step, results unused
        {
            uint4 x;
            for (int k = 0; k < RT_PROBE_EXTRA_LOADS; k++)
                asm volatile("global_load_dwordx4 %0, %1, %2 offset:32\n s_waitcnt vmcnt(0)" : "=v"(x) : "v"(r.cur), "s"(sc.pairs) : "memory");
        }
#endif
#ifdef RT_PROBE_EXTRA_VALU
        {
            float x = r.tmax;
            for (int k = 0; k < RT_PROBE_EXTRA_VALU; k++)
                asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(x) : "v"(r.A.x));
        }
#endif
#ifdef RT_PROBE_EXTRA_VALU_B
        {
            float x = r.tmax;
            for (int k = 0; k < RT_PROBE_EXTRA_VALU_B; k++)
                asm volatile("v_max3_f32 %0, %0, %1, %1" : "+v"(x) : "v"(r.A.x));
        }
#endif
#ifdef RT_PROBE_EXTRA_SALU
        for (int k = 0; k < RT_PROBE_EXTRA_SALU; k++)
            asm volatile("s_add_u32 s100, s100, 1" ::: "s100");
#endif
#ifdef RT_PROBE_EXTRA_LDS
        {
            uint32_t x;
            for (int k = 0; k < RT_PROBE_EXTRA_LDS; k++)
                asm volatile("ds_read_b32 %0, %1\n s_waitcnt lgkmcnt(0)" : "=v"(x) : "v"((uint32_t)(size_t)stk) : "memory");
        }
#endif
        float eL, eR;
        const bool hL = qboxTest(w0.x, w0.y, w0.z, r, eL);
        const bool hR = qboxTest(w0.w, w1.x, w1.y, r, eR);
        // nearer child first; on a tie the right child, as in src/bvh.cpp:301-307
        const bool takeLeft = hL && (!hR || eL < eR);
        const uint32_t nearRef = takeLeft ? w1.z : w1.w, farRef = takeLeft ? w1.w : w1.z;
        if (hL && hR)
            push(r, farRef);
        r.cur = (hL || hR) ? nearRef : pop(r);
    }

    // One primitive of the parked leaf, in order (src/bvh.cpp:272-279).
    __device__ __forceinline__ void leafStep(RayState& r) const
    {
        float t;
        if (STATS)
            r.nTris++;
        // Of two primitives hit at exactly the same t the reference keeps the one it tests later
        // (t <= ray.t, src/bvh.cpp:272-279): inside a leaf that is the one with the higher index.
        // A hierarchy of the library's own visits them in another order, so it decides by the index
        // for the time being, and the ray is marked: which of the two the reference tests later depends on
        // its own walk through the caller's tree, which settleTie() repeats once the traversal is over.
        bool accept;
        if (MOTION) {
            const TriRec rest = loadTri(sc.tris, r.li / 6u);
            accept = triangleTest(triFromVertices(moved(r.mv, rest.v0), moved(r.mv, rest.v1), moved(r.mv, rest.v2)), r.o, r.d, r.tmax, t);
        } else {
            accept = triangleTestLazy(sc.tris + r.li, r.o, r.d, r.tmax, t);
        }
        if (accept && t == r.tmax && r.hitPrim >= 0) {
            // (binary visibility asks for no primitive; a light sample under transparency looks at
            // nothing but the material of what it meets)
            if (r.kind == K_RADIANCE)
                r.tie = true;
            else if (r.kind == K_SHADOW_CLOSEST)
                r.tie = r.tie || __float_as_uint(sc.attrs[4u * (r.li / 6u) + 3u].w) != __float_as_uint(sc.attrs[4u * (uint32_t)r.hitPrim + 3u].w); // mesh ids
            accept = !(sc.tieByIndex && (int)(r.li / 6u) < r.hitPrim);
        }
        if (accept) {
            r.tmax = t;
            r.hitPrim = (int)(r.li / 6u);
            // Binary visibility only asks whether the closest hit lies more
            // than 1e-4 before the target (src/light.cpp:101-104); any hit
            // that does settles it, the closest one is at least as near.
            if (r.kind == K_SHADOW_ANY && fabsf(t - r.slen) > 0.0001f) {
                r.cur = kRefNone;
                r.pend = 0u;
                return;
            }
        }
        r.li += 6u;
        if (r.li >= r.le)
            r.pend = 0u;
    }

    // Which of two primitives met at exactly the same t the reference keeps is a matter of its order
    // of tests (it accepts t <= ray.t, src/bvh.cpp:272-279): the leaf it reaches later wins, and that
    // order follows from its own box routine over the caller's nodes -- entry distances by true
    // division, the exit distance for an origin inside a box, no culling by the closest hit
    // (src/bvh.cpp:283-313).  Such ties come about once in some millions of rays (coplanar or
    // edge-sharing primitives), so the marked ray is simply traced again the reference's way; the
    // closest t is the same, the primitive is the reference's.  Call when the traversal has ended
    // and before the spheres.
    // The other thing that takes the reference's own walk to decide: its box routine is not conservative.  It
    // divides per plane and compares the quotients as they round, so a ray that starts on a face of a box and
    // leaves it, or grazes an edge, can be turned away from a box whose primitive it does hit -- and the
    // reference then reports the next hit, or none.  The traversal here never loses a hit; so once it has
    // one, it asks whether the reference would have come to the leaf that holds it.  Boxes nest and every
    // operation of the routine is monotone, so the reference reaches a leaf whenever its routine lets the ray
    // into the leaf's own box at a distance above zero (at zero -- a ray that starts on the box -- the boxes
    // around it may decide otherwise: doubtful).  Nearly every hit is cleared at once: its point lies inside
    // the leaf's box, and when it does so by a margin far above any rounding, every quotient the routine forms
    // lies on the right side of the hit's own distance.  The rest (a hit at a vertex that is a corner of the
    // box; a triangle that lies IN a face of the box, as in axis-aligned scenes) gets the routine itself on that
    // one box, and a ray it turns away there is settled by the replay.
    // (the first, quick part on its own: the hit's point inside the box of the caller's leaf by a margin)
    __device__ __forceinline__ bool insideLeafBox(const RayState& r, f3& lo, f3& hi) const
    {
        const float4 b0 = sc.leafBoxes[2u * (uint32_t)r.hitPrim], b1 = sc.leafBoxes[2u * (uint32_t)r.hitPrim + 1u];
        lo = { b0.x, b0.y, b0.z }, hi = { b0.w, b1.x, b1.y };
        if (MOTION) {
            lo = moved(r.mv, lo);
            hi = moved(r.mv, hi);
        }
        const f3 td = r.tmax * r.d;
        const f3 X = r.o + td;
        const float k = 3.814697265625e-6f; // 2^-18
        const float mx = (fabsf(r.o.x) + fabsf(td.x) + fabsf(lo.x) + fabsf(hi.x)) * k;
        const float my = (fabsf(r.o.y) + fabsf(td.y) + fabsf(lo.y) + fabsf(hi.y)) * k;
        const float mz = (fabsf(r.o.z) + fabsf(td.z) + fabsf(lo.z) + fabsf(hi.z)) * k;
        // (written so that a value that is not a number anywhere answers "not sure")
        return lo.x + mx < X.x && X.x < hi.x - mx && lo.y + my < X.y && X.y < hi.y - my && lo.z + mz < X.z && X.z < hi.z - mz;
    }

    __device__ __forceinline__ bool surelyReached(const RayState& r) const
    {
        f3 lo, hi;
        if (insideLeafBox(r, lo, hi))
            return true;
        const float l3[3] = { lo.x, lo.y, lo.z }, h3[3] = { hi.x, hi.y, hi.z };
        const float o3[3] = { r.o.x, r.o.y, r.o.z }, d3[3] = { r.d.x, r.d.y, r.d.z };
        float t = kFltMax;
        return refBoxTest(l3, h3, o3, d3, t) && t > 0.0f;
    }

    __device__ __forceinline__ void settleTie(RayState& r) const
    {
        if (accel && sc.refNodes != nullptr && sc.leafBoxes != nullptr && r.hitPrim >= 0 && !r.tie && !surelyReached(r))
            r.tie = true;
#ifndef RT_PROBE_NO_REPLAY // (experiment: what the call costs the kernels that hold it; results wrong where ties occur)
        if (__builtin_expect(r.tie, 0) && sc.refNodes != nullptr && accel) {
            const uint2 res = replayReference<MOTION>(sc.refNodes, sc.tris, stk, stride, r.o.x, r.o.y, r.o.z, r.d.x, r.d.y, r.d.z, r.slen,
                MOTION ? &r.mv : nullptr);
            r.tmax = __uint_as_float(res.x);
            r.hitPrim = (int)res.y;
        }
#endif
        r.tie = false;
    }

    // Spheres are tested after the BVH, in order, against the current ray.t
    // (src/bvh.cpp:354-355); a sphere hit is recorded as -2 - index.
    __device__ __forceinline__ void spheres(RayState& r) const
    {
        for (int s = 0; s < sc.nSpheres; s++) {
            float t;
            DevSphere sp = sc.spheres[s];
            if (MOTION) { // src/extra.cpp:149-161
                const f3 c = moved(r.mv, ld3(sp.center));
                sp.center[0] = c.x, sp.center[1] = c.y, sp.center[2] = c.z;
            }
            if (sphereTest(sp, r.o, r.d, r.tmax, t)) {
                r.tmax = t;
                r.hitPrim = -2 - s;
            }
        }
    }
};

// What updateHitInfo (src/bvh.cpp:18-48) and sampleMaterialKd
// (src/shading.cpp:13-24) leave behind for the accepted closest hit.
struct HitCtx {
    f3 P, N, kd, mkd, ks;
    float shin, tr;
};

template <bool MOTION = false>
__device__ __forceinline__ HitCtx shadeHit(const DevScene& sc, const DevFeat& ft, const RayState& r)
{
    HitCtx h;
    h.P = r.o + r.tmax * r.d;
    DevMaterial m;
    if (r.hitPrim >= 0) {
        const uint32_t i = (uint32_t)r.hitPrim;
        // flat, untextured shading needs the plane's normal and the mesh id only; the vertices and
        // their attributes are loaded where interpolation asks for them
        const float4 plane = sc.tris[6u * i];
        const float4 a3 = sc.attrs[4u * i + 3];
        m = sc.mats[__float_as_uint(a3.w)];
        f3 planeN = { plane.x, plane.y, plane.z };
        TriRec tr;
        if (MOTION) {
            const TriRec rest = loadTri(sc.tris, i);
            tr = triFromVertices(moved(r.mv, rest.v0), moved(r.mv, rest.v1), moved(r.mv, rest.v2));
            planeN = tr.n;
        }
        f3 N = planeN;
        f2 uv = { 0.0f, 0.0f };
        if (ft.normalInterp || ft.texMapping) {
            if (!MOTION)
                tr = loadTri(sc.tris, i);
            const float4 a0 = sc.attrs[4u * i + 0], a1 = sc.attrs[4u * i + 1], a2 = sc.attrs[4u * i + 2];
            // computeBarycentricCoord, src/interpolate.cpp:13-24
            const f3 n = cross(tr.v1 - tr.v0, tr.v2 - tr.v0);
            const f3 na = cross(tr.v2 - tr.v1, h.P - tr.v1);
            const f3 nb = cross(tr.v0 - tr.v2, h.P - tr.v2);
            const float alpha = dot(n, na) / dot(n, n);
            const float beta = dot(n, nb) / dot(n, n);
            const float gamma = 1.0f - alpha - beta;
            if (ft.normalInterp) // src/interpolate.cpp:34-40
                N = alpha * mk3(a0.x, a0.y, a0.z) + beta * mk3(a0.w, a1.x, a1.y) + gamma * mk3(a1.z, a1.w, a2.x);
            if (ft.texMapping) { // src/interpolate.cpp:50-55
                uv.x = a2.y * alpha + a2.w * beta + a3.y * gamma;
                uv.y = a2.z * alpha + a3.x * beta + a3.z * gamma;
            }
        }
        if (dot(r.d, planeN) > 0.0f)
            N = -N;
        h.N = N;
        h.kd = (ft.texMapping && m.tex >= 0) ? sampleTexture(sc, m.tex, uv, ft.bilinear != 0) : ld3(m.kd);
    } else {
        const DevSphere s = sc.spheres[-2 - r.hitPrim];
        m = sc.mats[s.material];
        h.N = normalize(h.P - (MOTION ? moved(r.mv, ld3(s.center)) : ld3(s.center)));
        h.kd = ld3(m.kd);
    }
    h.mkd = ld3(m.kd);
    h.ks = ld3(m.ks);
    h.shin = m.shininess;
    h.tr = m.transparency;
    return h;
}

// generateReflectionRay, src/recursive.cpp:72-89
__device__ __forceinline__ void reflectionRay(const f3& P, const f3& rd, const f3& N, f3& o, f3& d)
{
    const f3 nd = normalize(rd);
    d = normalize(nd - 2.0f * dot(nd, N) * N);
    o = P + N * 1.0e-5f;
}

// Camera ray i of pixel (px, py): generatePixelRays* (src/render.cpp:56-131)
// plus the lens fan-out of src/extra.cpp:48-81.  sPix / sLens are the sampler
// cursors of the pixel-sample and lens stretches of the pixel's stream; `pof`
// carries the point of focus from a camera ray to its lens samples.
template <bool DOF>
__device__ __forceinline__ void cameraRay(const DevCamera& cam, const DevFeat& ft, const DevFrame& fr, int px, int py,
    uint32_t rayI, uint32_t& sPix, uint32_t& sLens, f3& pof, f3& o, f3& d)
{
    const uint32_t dofN = (DOF && ft.dof) ? ft.dofSamples : 1u;
    const bool lens = DOF && ft.dof;
    const uint32_t c = lens ? fastDiv(rayI, ft.divDof) : rayI;
    f3 dir = { 0.0f, 0.0f, 0.0f };
    if (!lens || rayI - c * dofN == 0) {
        float ndcX, ndcY;
        const float W = (float)fr.width, H = (float)fr.height;
        if (ft.spp > 1) {
            if (ft.jitter) {
                const int p = (int)(c / ft.jitterN), q = (int)(c % ft.jitterN);
                const float a = next1d(sPix);
                const float b = next1d(sPix);
                const float jx = (float)px + ((float)p + a) / (float)ft.jitterN;
                const float jy = (float)py + ((float)q + b) / (float)ft.jitterN;
                ndcX = jx / W * 2.0f - 1.0f;
                ndcY = jy / H * 2.0f - 1.0f;
            } else {
                const float a = next1d(sPix);
                const float b = next1d(sPix);
                ndcX = ((float)px + a) / W * 2.0f - 1.0f;
                ndcY = ((float)py + b) / H * 2.0f - 1.0f;
            }
        } else {
            ndcX = ((float)px + 0.5f) / W * 2.0f - 1.0f;
            ndcY = ((float)py + 0.5f) / H * 2.0f - 1.0f;
        }
        // Trackball::generateRay, framework/src/trackball.cpp:101-110
        const f3 cd = normalize(mk3(-ndcX * cam.halfW, ndcY * cam.halfH, 1.0f));
        dir = quatRotate(cam.qw, cam.q, cd);
        if (DOF && ft.dof) { // getPointOfFocus, src/extra.cpp:15-23
            const float cosAngle = dot(dir, cam.fwd);
            const float curDistance = ft.dofDistance / cosAngle;
            pof = cam.pos + curDistance * dir;
        }
    }
    if (DOF && ft.dof) { // src/extra.cpp:27-44, :70-76
        const float r0 = next1d(sLens);
        const float r1 = next1d(sLens);
        const float angle = r0 * kTwoPi;
        const float dist = sqrtf(r1) * (ft.dofDiameter * 0.5f);
        const float ox = cosLikeHost(angle) * dist; // the C library's results, bit for bit (rt_trig.hpp)
        const float oy = sinLikeHost(angle) * dist;
        o = cam.pos + ox * cam.up + oy * cam.left;
        d = normalize(pof - o);
    } else {
        o = cam.pos;
        d = dir;
    }
}

// Can a camera ray of this pixel enter the scene's box at all (DevFrame::cand*)?
__device__ __forceinline__ bool seesScene(const DevFrame& fr, int px, int py)
{
    return px >= fr.candX0 && px <= fr.candX1 && py >= fr.candY0 && py <= fr.candY1;
}

// Pixel of local work slot `slot` (64 consecutive slots = one 8x8 tile; tile k
// of the frame belongs to shard k % nShard).  Returns false for the padding
// pixels of partial edge tiles.
__device__ __forceinline__ bool slotToPixel(const DevFrame& fr, uint32_t slot, int& px, int& py)
{
    const uint32_t tileLocal = slot >> 6, within = slot & 63u;
    const uint32_t tile = tileLocal * (uint32_t)fr.nShard + (uint32_t)fr.shard;
    const uint32_t row = fastDiv(tile, fr.divTilesX);
    const int tx = (int)(tile - row * (uint32_t)fr.tilesX), ty = (int)row;
    px = tx * kTile + (int)(within & 7u);
    py = ty * kTile + (int)(within >> 3);
    return px < fr.width && py < fr.height;
}

} // namespace rt
