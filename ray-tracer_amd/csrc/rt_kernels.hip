// rt_kernels.hip -- the general render kernel, the batch intersect kernel and
// the small helper kernels of the rendering hot path (gfx950).
//
// renderKernel: persistent 64-lane wavefronts.  Each lane owns one pixel at a
// time and walks that pixel's whole ray tree in the reference's depth-first
// order, because one per-pixel Sampler stream is consumed across pixel
// samples, lens samples, light samples and glossy samples
// (src/render.cpp:34-39), and how many draws a ray consumes depends on what it
// hits.  The recursion of src/recursive.cpp:25-64 is unrolled into a per-lane
// state machine with an explicit frame stack; every ray of every kind goes
// through ONE shared while-while BVH traversal loop (rt_trace.hpp), so lanes
// that are in different places of their ray trees still traverse together.
// Lanes that finish a pixel pull the next one from an atomic work queue of 8x8
// tiles (wave-aggregated: one atomic per 64 pixels).
//
// This kernel handles every feature combination (transparency, glossy
// reflections included).  Frames without those two take the wavefront path of
// rt_wavefront.hip instead, which spreads the rays of one pixel over many
// lanes; both produce the same bits.
//
// Motion blur (src/extra.cpp:121-211) is a variant of this kernel only: the
// reference renders every time sample of every pixel in a moved copy of the
// whole scene with a tree built for that copy; here the lane keeps the rest
// scene's tree and moves the vertices it meets (Tracer<., MOTION>), and the
// loop over time samples wraps the loop over the pixel's camera rays.
//
// No MFMA: nothing here is a dense contraction.

#include "rt_trace.hpp"

namespace rt {

enum Phase : int {
    PH_NEWPIXEL = 0, // wants a pixel from the queue
    PH_NEXTRAY,      // generate the pixel's next camera ray, or finish the pixel
    PH_TRACE,        // ray in flight in the shared traversal loop
    PH_LIGHT,        // direct-light loop over lights and light samples
    PH_RECURSE,      // direct light done: spawn reflection / glossy / pass-through
    PH_RETURN,       // a renderRay value is ready at `depth`
    PH_DEAD
};
enum FrameStage : int { FR_REFLECT = 0, FR_GLOSSY = 1, FR_TRANSP = 2 };

constexpr int V_TRANSP = 1, V_GLOSSY = 2, V_DOF = 4, V_STATS = 8, V_MOTION = 16;

// What renderRay keeps of its own hit while a child ray is traced
// (src/recursive.cpp:36-61): Lo, the material terms the combine needs, and the
// hit point / incoming direction the pass-through ray is built from.
struct Frame {
    f3 Lo, ks, P, rd;
    float tr;
    int stage;
};
struct GlossyFrame {
    f3 R, u, v, acc;
    float radius;
    uint32_t i;
};

struct Lane {
    int phase;
    // pixel
    int px, py;
    uint32_t widx;
    uint32_t sMain, sPix, sLens;
    uint32_t rayI;
    f3 Lpix;
    f3 pof;
    // motion blur: the sampler the times are drawn from, the time sample under way, the sum over them
    uint32_t sTime, mi;
    f3 Lmo;
    // the hit being shaded
    int depth;
    HitCtx h;
    f3 rd;
    // direct-light loop
    int li;
    uint32_t si;
    f3 Lo, acc, lcol;
    float tprod;
    f3 curKd;
    // value handed back by PH_RETURN
    f3 value;
    // ray in flight
    RayState r;
    uint32_t nrays;
};

template <int V>
struct Shader {
    const DevScene& sc;
    const DevCamera& cam;
    const DevFeat& ft;
    const DevFrame& fr;
    const Tracer<(V & V_STATS) != 0, (V & V_MOTION) != 0>& tracer;
    float* out;
    Frame* frames; // per-lane private arrays (scratch)
    GlossyFrame* gframes;

    __device__ __forceinline__ void launch(Lane& L, f3 o, f3 d, float tmax, int kind) const
    {
        tracer.launch(L.r, o, d, tmax, kind);
        L.nrays++;
        L.phase = PH_TRACE;
    }

    __device__ __forceinline__ DevMaterial materialOfHit(int hitPrim) const
    {
        if (hitPrim >= 0) {
            const uint32_t meshID = __float_as_uint(sc.attrs[4u * (uint32_t)hitPrim + 3u].w);
            return sc.mats[meshID];
        }
        return sc.mats[sc.spheres[-2 - hitPrim].material];
    }

    // computeContributionPointLight after visibility is known (src/light.cpp:171-186),
    // then the per-light bookkeeping of src/light.cpp:205-219 / :239-253 / :283-298.
    __device__ __forceinline__ void addSample(Lane& L, f3 visible, f3 l, int type) const
    {
        if (!isZero(visible)) {
            f3 N = L.h.N;
            if (ft.transparency && L.h.tr < 1.0f && dot(L.h.N, l) < 0.0f)
                N = -L.h.N;
            L.acc = L.acc + computeShading(ft, N, L.h.kd, L.h.ks, L.h.shin, -L.rd, l, visible);
        } else {
            L.acc = L.acc + mk3(0.0f);
        }
        L.si++;
        const uint32_t nS = (type == 0) ? 1u : ft.shadowSamples;
        if (L.si >= nS) {
            L.Lo = L.Lo + ((type == 0) ? L.acc : L.acc * (1.0f / (float)nS));
            L.li++;
            L.si = 0;
        }
        L.phase = PH_LIGHT;
    }

    __device__ __forceinline__ void lightStep(Lane& L) const
    {
        if (L.li >= sc.nLights) {
            L.Lo = clamp01(L.Lo); // src/light.cpp:297
            L.phase = PH_RECURSE;
            return;
        }
        const rt_light& light = sc.lights[L.li];
        const int type = light.type;
        if (L.si == 0)
            L.acc = mk3(0.0f);
        if (type != 0 && ft.shadowSamples == 0) { // zero samples: 0 * (1/0), as the reference computes it
            L.Lo = L.Lo + L.acc * (1.0f / 0.0f);
            L.li++;
            return;
        }
        f3 pos, col;
        if (type == 0) {
            pos = ld3(light.v + 0);
            col = ld3(light.v + 3);
        } else if (type == 1) {
            sampleSegmentLight(next1d(L.sMain), light, pos, col);
        } else {
            const float sx = next1d(L.sMain);
            const float sy = next1d(L.sMain);
            sampleParallelogramLight({ sx, sy }, light, pos, col);
        }
        if (!ft.shadows) { // src/light.cpp:270-272
            addSample(L, col, normalize(pos - L.h.P), type);
            return;
        }
        // shadow ray from the light to the hit point, src/light.cpp:94-97 / :128-131
        L.lcol = col;
        const f3 toHit = L.h.P - pos;
        const f3 sd = normalize(toHit);
        const float len = length(toHit);
        if ((V & V_TRANSP) && ft.transparency) {
            L.tprod = 1.0f;
            L.curKd = L.h.mkd;
            launch(L, pos, sd, len, K_SHADOW_CLOSEST);
        } else {
            launch(L, pos, sd, len, K_SHADOW_ANY);
        }
    }

    __device__ __forceinline__ void shadowDone(Lane& L) const
    {
        const int type = sc.lights[L.li].type;
        const f3 l = -L.r.d; // normalize(light - p) == -normalize(p - light), exactly
        if (L.r.kind == K_SHADOW_ANY) {
            // src/light.cpp:101-107
            const bool visible = (L.r.hitPrim == kHitNone) ? true : (fabsf(L.r.tmax - L.r.slen) <= 0.0001f);
            addSample(L, visible ? L.lcol : mk3(0.0f), l, type);
            return;
        }
        // src/light.cpp:138-150
        if (V & V_TRANSP) {
            const f3 hp = L.r.o + L.r.tmax * L.r.d;
            bool again = false;
            if (L.r.hitPrim != kHitNone && !allEpsEq(hp, L.h.P, 0.0001f)) {
                const DevMaterial hm = materialOfHit(L.r.hitPrim);
                if (hm.transparency != 1.0f) {
                    L.curKd = ld3(hm.kd);
                    L.tprod *= hm.transparency;
                    again = true;
                }
            }
            if (again) {
                launch(L, hp + L.r.d * 1.0e-5f, L.r.d, kFltMax, K_SHADOW_CLOSEST); // src/recursive.cpp:98-111
                return;
            }
            f3 visible = mk3(0.0f);
            if (allEpsEq(hp, L.h.P, 0.0001f))
                visible = L.lcol * L.curKd * L.tprod;
            addSample(L, visible, l, type);
        }
    }

    __device__ __forceinline__ void glossyLaunch(Lane& L, int fi) const
    {
        if (V & V_GLOSSY) {
            GlossyFrame& g = gframes[fi];
            const Frame& f = frames[fi];
            const float sx = next1d(L.sMain);
            const float sy = next1d(L.sMain);
            const float r = g.radius * sx;
            const float theta = kTwoPi * sy;
            const f3 sU = g.u * r * cosLikeHost(theta);
            const f3 sV = g.v * r * sinLikeHost(theta);
            const f3 dir = normalize(g.R + sU + sV);
            L.depth = fi + 1;
            launch(L, f.P + dir * 1.0e-5f, dir, kFltMax, K_RADIANCE);
        }
    }

    // Pass-through child of frame fi, or hand the frame's value back (src/recursive.cpp:57-63).
    __device__ __forceinline__ void afterReflection(Lane& L, int fi, f3 Lo) const
    {
        Frame& f = frames[fi];
        if ((V & V_TRANSP) && ft.transparency && f.tr != 1.0f) {
            f.Lo = Lo;
            f.stage = FR_TRANSP;
            L.depth = fi + 1;
            launch(L, f.P + f.rd * 1.0e-5f, f.rd, kFltMax, K_RADIANCE); // src/recursive.cpp:98-111
            return;
        }
        L.value = Lo;
        L.depth = fi;
        L.phase = PH_RETURN;
    }

    // src/recursive.cpp:43-63
    __device__ __forceinline__ void recurse(Lane& L) const
    {
        const int fi = L.depth;
        if (fi < ft.maxDepth) {
            const bool reflective = !(L.h.ks.x == 0.0f && L.h.ks.y == 0.0f && L.h.ks.z == 0.0f);
            const bool transparent = L.h.tr != 1.0f;
            const bool wantRefl = ft.reflections && reflective;
            const bool wantTr = (V & V_TRANSP) && ft.transparency && transparent;
            if (wantRefl || wantTr) {
                Frame& f = frames[fi];
                f.Lo = L.Lo;
                f.ks = L.h.ks;
                f.P = L.h.P;
                f.rd = L.rd;
                f.tr = L.h.tr;
                if (wantRefl && !ft.glossy) {
                    f.stage = FR_REFLECT;
                    f3 o, d;
                    reflectionRay(L.h.P, L.rd, L.h.N, o, d);
                    L.depth = fi + 1;
                    launch(L, o, d, kFltMax, K_RADIANCE);
                    return;
                }
                if ((V & V_GLOSSY) && wantRefl && ft.glossy) { // src/extra.cpp:329-368
                    f.stage = FR_GLOSSY;
                    GlossyFrame& g = gframes[fi];
                    f3 o;
                    reflectionRay(L.h.P, L.rd, L.h.N, o, g.R);
                    f3 arbitrary = { 1.0f, 0.0f, 0.0f };
                    if (length(arbitrary - g.R) < 0.01f)
                        arbitrary = { 0.0f, 1.0f, 0.0f };
                    g.u = normalize(cross(arbitrary, g.R));
                    g.v = normalize(cross(g.R, g.u));
                    g.radius = ft.glossyExponent * L.h.shin / 64.0f;
                    g.acc = mk3(0.0f);
                    g.i = 0;
                    if ((int)ft.glossySamples > 0) {
                        glossyLaunch(L, fi);
                        return;
                    }
                    const f3 Lo = L.Lo + (g.acc / (float)(int)ft.glossySamples) * L.h.ks;
                    afterReflection(L, fi, Lo);
                    return;
                }
                afterReflection(L, fi, L.Lo);
                return;
            }
        }
        L.value = L.Lo;
        L.phase = PH_RETURN;
    }

    // A renderRay call at L.depth returned L.value.
    __device__ __forceinline__ void ret(Lane& L) const
    {
        if (L.depth == 0) { // src/recursive.cpp:13-15
            L.Lpix = L.Lpix + L.value;
            L.rayI++;
            L.phase = PH_NEXTRAY;
            return;
        }
        const int fi = L.depth - 1;
        Frame& f = frames[fi];
        if (f.stage == FR_REFLECT) { // src/recursive.cpp:126
            afterReflection(L, fi, f.Lo + L.value * f.ks);
        } else if (f.stage == FR_TRANSP) { // src/recursive.cpp:142
            L.value = (L.value * f.tr) + (f.Lo * (1.0f - f.tr));
            L.depth = fi;
        } else if (V & V_GLOSSY) {
            GlossyFrame& g = gframes[fi];
            g.acc = g.acc + L.value;
            g.i++;
            if (g.i < ft.glossySamples) {
                glossyLaunch(L, fi);
                return;
            }
            afterReflection(L, fi, f.Lo + (g.acc / (float)(int)ft.glossySamples) * f.ks);
        }
    }

    // One time sample begins (src/extra.cpp:145-204): its time is the next draw of the pixel's sampler,
    // and the sample is rendered with a copy of that sampler as it stands after the draw, so sample
    // i + 1 sees the stream of sample i moved on by one.
    __device__ __forceinline__ void beginMotionSample(Lane& L) const
    {
        const float time = next1d(L.sTime);
        L.r.mv = makeMover(time, ft.movement);
        L.sPix = L.sTime;
        L.sMain = ft.mainMul * L.sTime + ft.mainAdd;
        L.rayI = 0;
        L.Lpix = mk3(0.0f);
    }

    // A pixel begins.  The sum over the time samples starts at zero here; the reference adds into a
    // vector it never initialises (src/extra.cpp:144, :205).
    __device__ __forceinline__ void beginMotionPixel(Lane& L, uint32_t seed) const
    {
        L.sTime = seed;
        L.mi = 0;
        L.Lmo = mk3(0.0f);
        if (ft.motionSamples > 0) {
            beginMotionSample(L);
        } else { // no sample at all: 0 / samples, through the same code as the last sample of a pixel
            L.rayI = ft.raysPerPixel;
            L.Lpix = mk3(0.0f);
            L.mi = 0xffffffffu;
        }
    }

    __device__ __forceinline__ void nextRay(Lane& L) const
    {
        if (L.rayI >= ft.raysPerPixel) {
            f3 c = L.Lpix / (float)ft.raysPerPixel; // src/recursive.cpp:16
            bool done = true;
            if (V & V_MOTION) { // src/extra.cpp:205-208
                L.Lmo = L.Lmo + c;
                L.mi++;
                done = (int)L.mi >= ft.motionSamples;
                if (done)
                    c = L.Lmo / (float)ft.motionSamples;
                else
                    beginMotionSample(L);
            }
            if (done) {
                const size_t idx = fr.packed ? (size_t)L.widx : ((size_t)(fr.height - 1 - L.py) * fr.width + L.px);
                out[3 * idx + 0] = c.x;
                out[3 * idx + 1] = c.y;
                out[3 * idx + 2] = c.z;
                L.phase = PH_NEWPIXEL;
                return;
            }
        }
        f3 o, d;
        cameraRay<(V & V_DOF) != 0>(cam, ft, fr, L.px, L.py, L.rayI, L.sPix, L.sLens, L.pof, o, d);
        L.depth = 0;
        launch(L, o, d, kFltMax, K_RADIANCE);
    }

    // Run this lane's state machine until it has a ray in the traversal loop,
    // wants a new pixel, or is dead.
    __device__ __forceinline__ void advance(Lane& L) const
    {
        for (;;) {
            switch (L.phase) {
            case PH_NEXTRAY:
                nextRay(L);
                break;
            case PH_TRACE:
                if (L.r.active)
                    return;
                tracer.settleTie(L.r);
                if (sc.nSpheres)
                    tracer.spheres(L.r);
                if (L.r.kind == K_RADIANCE) {
                    if (L.r.hitPrim == kHitNone) { // src/recursive.cpp:29-32
                        L.value = sampleEnvironmentMap(sc, ft, L.r.d);
                        L.phase = PH_RETURN;
                    } else {
                        L.h = shadeHit<(V & V_MOTION) != 0>(sc, ft, L.r);
                        L.rd = L.r.d;
                        L.li = 0;
                        L.si = 0;
                        L.Lo = mk3(0.0f);
                        L.acc = mk3(0.0f);
                        L.phase = PH_LIGHT;
                    }
                } else {
                    shadowDone(L);
                }
                break;
            case PH_LIGHT:
                lightStep(L);
                break;
            case PH_RECURSE:
                recurse(L);
                break;
            case PH_RETURN:
                ret(L);
                break;
            default:
                return;
            }
        }
    }
};

#ifndef RT_MIN_WAVES
#define RT_MIN_WAVES 1
#endif

template <int V>
__global__ __launch_bounds__(kBlockThreads, RT_MIN_WAVES) void renderKernel(const DevScene sc, const DevCamera cam, const DevFeat ft,
    const DevFrame fr, DevCounters* __restrict__ ctr, float* __restrict__ out, int minActive, int logicRounds)
{
    extern __shared__ uint32_t stackMem[];
    Frame frames[kMaxDepth + 1];
    GlossyFrame gframes[(V & V_GLOSSY) ? (kMaxDepth + 1) : 1];

    const Tracer<(V & V_STATS) != 0, (V & V_MOTION) != 0> tracer { sc, ft.accel, (LdsWords)(stackMem + threadIdx.x) };
    const Shader<V> shader { sc, cam, ft, fr, tracer, out, frames, gframes };

    Lane L;
    L.phase = PH_NEWPIXEL;
    L.nrays = 0;
    L.r.nInner = L.r.nTris = 0;
    L.r.itInner = L.r.itLeaf = L.r.itRuns = 0;
    L.r.topHits[0] = L.r.topHits[1] = L.r.topHits[2] = 0;
    L.r.cycInner = L.r.cycLeaf = L.r.cycRun = 0;
    L.r.cur = kRefNone;
    L.r.pend = 0u;
    L.r.top = tracer.stk;
    L.r.tos = kRefNone;
    L.r.active = false;
    L.depth = 0;

    // Work queue: 64-pixel chunks (one 8x8 tile).  The first chunk of every
    // wave is its own global wave index; later ones come from one atomic add
    // per chunk, issued by one lane and broadcast.
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wavesPerBlock = blockDim.x >> 6;
    const uint32_t waveId = blockIdx.x * wavesPerBlock + (threadIdx.x >> 6);
    const uint32_t nWaves = gridDim.x * wavesPerBlock;
    uint32_t cursor = waveId * 64u;
    uint32_t cend = (cursor < fr.nWork) ? cursor + 64u : cursor;
    bool exhausted = false;

    for (;;) {
        for (int round = 0; round < logicRounds; round++) {
            // ---- hand pixels to the lanes that want one
            for (;;) {
                const bool want = (L.phase == PH_NEWPIXEL);
                const unsigned long long wm = __ballot(want);
                if (wm == 0ull)
                    break;
                if (cursor == cend) {
                    if (exhausted) {
                        if (want)
                            L.phase = PH_DEAD;
                        break;
                    }
                    uint32_t base = 0;
                    if (lane == (uint32_t)__builtin_ctzll(wm))
                        base = atomicAdd(&ctr->next.v, 64u);
                    base = __shfl(base, __builtin_ctzll(wm)) + nWaves * 64u;
                    if (base >= fr.nWork) {
                        exhausted = true;
                        continue;
                    }
                    cursor = base;
                    cend = base + 64u;
                }
                const uint32_t rank = __popcll(wm & ((1ull << lane) - 1ull));
                const uint32_t avail = cend - cursor;
                const uint32_t nWant = __popcll(wm);
                if (want && rank < avail) {
                    const uint32_t w = cursor + rank;
                    L.widx = w;
                    if (slotToPixel(fr, w, L.px, L.py)) {
                        const uint32_t seed = (uint32_t)(fr.height * L.px + L.py); // src/render.cpp:38
                        L.sPix = seed;
                        L.sLens = ft.lensMul * seed + ft.lensAdd;
                        L.sMain = ft.mainMul * seed + ft.mainAdd;
                        L.rayI = 0;
                        L.Lpix = mk3(0.0f);
                        L.phase = PH_NEXTRAY;
                        if (V & V_MOTION)
                            shader.beginMotionPixel(L, seed);
                    }
                    // padding pixels of a partial edge tile are skipped: the lane stays in
                    // PH_NEWPIXEL and takes part in the next hand-out
                }
                cursor += (nWant < avail) ? nWant : avail;
            }
            // ---- per-lane logic until a ray is in flight / a new pixel is wanted
            shader.advance(L);
            if (__ballot(L.phase == PH_NEWPIXEL) == 0ull)
                break;
        }
        if (__ballot(L.phase == PH_TRACE) == 0ull) {
            if (__ballot(L.phase != PH_DEAD) == 0ull)
                break;
            continue;
        }
        tracer.run(L.r, exhausted ? (minActive & ~0xff) : minActive);
    }

    // ---- counters: one atomic per wave
    unsigned long long r = L.nrays, ni = L.r.nInner, nt = L.r.nTris;
    for (int off = 32; off > 0; off >>= 1) {
        r += __shfl_down(r, off);
        if (V & V_STATS) {
            ni += __shfl_down(ni, off);
            nt += __shfl_down(nt, off);
        }
    }
    if (lane == 0) {
        atomicAdd(&ctr->rays, r);
        if (V & V_STATS) {
            atomicAdd(&ctr->inner, ni);
            atomicAdd(&ctr->tris, nt);
        }
    }
}

// ----------------------------------------------------------------------------
// BVHInterface::intersect for a batch of rays: one lane per ray, same traversal.
// ----------------------------------------------------------------------------
__global__ __launch_bounds__(kBlockThreads) void intersectKernel(const DevScene sc, const DevFeat ft, rt_ray* __restrict__ rays,
    rt_hit* __restrict__ hits, uint32_t n)
{
    extern __shared__ uint32_t stackMem[];
    const Tracer<false> tracer { sc, ft.accel, (LdsWords)(stackMem + threadIdx.x) };
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    RayState L;
    L.cur = kRefNone;
    L.pend = 0u;
    L.top = tracer.stk;
    L.tos = kRefNone;
    L.active = false;
    L.nInner = L.nTris = 0;
    if (i < n) {
        const rt_ray r = rays[i];
        tracer.launch(L, ld3(r.origin), ld3(r.direction), r.t, K_RADIANCE);
    }
    tracer.run(L, (16 << 8) | (32 << 16)); // every ray to completion, default switch points
    if (i >= n)
        return;
    if (sc.strictIntersect)
        L.tie = true;
    tracer.settleTie(L);
    tracer.spheres(L);
    rt_hit h;
    h.hit = (L.hitPrim != kHitNone) ? 1 : 0;
    h.t = L.tmax;
    h.primitive = L.hitPrim >= 0 ? L.hitPrim : -1;
    h.material = -1;
    h.normal[0] = h.normal[1] = h.normal[2] = 0.0f;
    h.barycentricCoord[0] = h.barycentricCoord[1] = h.barycentricCoord[2] = 0.0f;
    h.texCoord[0] = h.texCoord[1] = 0.0f;
    const f3 P = L.o + L.tmax * L.d;
    if (L.hitPrim >= 0) { // updateHitInfo, src/bvh.cpp:18-48
        const uint32_t k = (uint32_t)L.hitPrim;
        const TriRec r = loadTri(sc.tris, k);
        const float4 a0 = sc.attrs[4u * k + 0], a1 = sc.attrs[4u * k + 1], a2 = sc.attrs[4u * k + 2], a3 = sc.attrs[4u * k + 3];
        h.material = (int)__float_as_uint(a3.w);
        const f3 nn = cross(r.v1 - r.v0, r.v2 - r.v0);
        const f3 na = cross(r.v2 - r.v1, P - r.v1);
        const f3 nb = cross(r.v0 - r.v2, P - r.v2);
        const float alpha = dot(nn, na) / dot(nn, nn);
        const float beta = dot(nn, nb) / dot(nn, nn);
        const float gamma = 1.0f - alpha - beta;
        f3 N = r.n;
        if (ft.normalInterp)
            N = alpha * mk3(a0.x, a0.y, a0.z) + beta * mk3(a0.w, a1.x, a1.y) + gamma * mk3(a1.z, a1.w, a2.x);
        if (ft.texMapping) {
            h.texCoord[0] = a2.y * alpha + a2.w * beta + a3.y * gamma;
            h.texCoord[1] = a2.z * alpha + a3.x * beta + a3.z * gamma;
        }
        if (dot(L.d, r.n) > 0.0f)
            N = -N;
        h.normal[0] = N.x, h.normal[1] = N.y, h.normal[2] = N.z;
        h.barycentricCoord[0] = alpha, h.barycentricCoord[1] = beta, h.barycentricCoord[2] = gamma;
    } else if (L.hitPrim <= -2) {
        const DevSphere s = sc.spheres[-2 - L.hitPrim];
        const f3 N = normalize(P - ld3(s.center));
        h.material = s.material;
        h.normal[0] = N.x, h.normal[1] = N.y, h.normal[2] = N.z;
    }
    hits[i] = h;
    rays[i].t = L.tmax;
}

// Shard-packed tiles -> frame in Screen order (src/screen.cpp:36-42).
__global__ void deinterleaveKernel(const float* __restrict__ packed, float* __restrict__ frame, int width, int height,
    int tilesX, int tilesY, int nShard, size_t shardStride)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; // global pixel slot over all tiles
    const size_t nTiles = (size_t)tilesX * tilesY;
    if (i >= nTiles * 64)
        return;
    const uint32_t tile = (uint32_t)(i >> 6), within = (uint32_t)(i & 63);
    const int x = (int)(tile % (uint32_t)tilesX) * kTile + (int)(within & 7u);
    const int y = (int)(tile / (uint32_t)tilesX) * kTile + (int)(within >> 3);
    if (x >= width || y >= height)
        return;
    const uint32_t shard = tile % (uint32_t)nShard, tileLocal = tile / (uint32_t)nShard;
    const float* src = packed + (size_t)shard * shardStride + 3 * ((size_t)tileLocal * 64 + within);
    float* dst = frame + 3 * ((size_t)(height - 1 - y) * width + x);
    dst[0] = src[0];
    dst[1] = src[1];
    dst[2] = src[2];
}

// Pure unit-tested functions on the device (one lane per record).
__global__ void evalKernel(const DevScene sc, const DevCamera cam, const DevFeat ft, int kind, const float* __restrict__ in,
    float* __restrict__ out, uint32_t n)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n)
        return;
    switch (kind) {
    case RT_EVAL_BARYCENTRIC: {
        const float* a = in + 12 * i;
        const f3 v0 = ld3(a), v1 = ld3(a + 3), v2 = ld3(a + 6), p = ld3(a + 9);
        const f3 nn = cross(v1 - v0, v2 - v0);
        const f3 na = cross(v2 - v1, p - v1);
        const f3 nb = cross(v0 - v2, p - v2);
        const float alpha = dot(nn, na) / dot(nn, nn);
        const float beta = dot(nn, nb) / dot(nn, nn);
        out[3 * i + 0] = alpha, out[3 * i + 1] = beta, out[3 * i + 2] = 1.0f - alpha - beta;
    } break;
    case RT_EVAL_INTERP_NORMAL: {
        const float* a = in + 12 * i;
        const f3 r = a[9] * ld3(a) + a[10] * ld3(a + 3) + a[11] * ld3(a + 6);
        out[3 * i + 0] = r.x, out[3 * i + 1] = r.y, out[3 * i + 2] = r.z;
    } break;
    case RT_EVAL_INTERP_TEXCOORD: {
        const float* a = in + 9 * i;
        out[2 * i + 0] = a[0] * a[6] + a[2] * a[7] + a[4] * a[8];
        out[2 * i + 1] = a[1] * a[6] + a[3] * a[7] + a[5] * a[8];
    } break;
    case RT_EVAL_TEX_NEAREST:
    case RT_EVAL_TEX_BILINEAR: {
        const float* a = in + 3 * i;
        const f3 r = sampleTexture(sc, (int)a[0], { a[1], a[2] }, kind == RT_EVAL_TEX_BILINEAR);
        out[3 * i + 0] = r.x, out[3 * i + 1] = r.y, out[3 * i + 2] = r.z;
    } break;
    case RT_EVAL_SEGMENT_LIGHT: {
        f3 p, c;
        sampleSegmentLight(in[2 * i + 1], sc.lights[(int)in[2 * i]], p, c);
        float* o = out + 6 * i;
        o[0] = p.x, o[1] = p.y, o[2] = p.z, o[3] = c.x, o[4] = c.y, o[5] = c.z;
    } break;
    case RT_EVAL_PARALLELOGRAM: {
        f3 p, c;
        sampleParallelogramLight({ in[3 * i + 1], in[3 * i + 2] }, sc.lights[(int)in[3 * i]], p, c);
        float* o = out + 6 * i;
        o[0] = p.x, o[1] = p.y, o[2] = p.z, o[3] = c.x, o[4] = c.y, o[5] = c.z;
    } break;
    case RT_EVAL_SAMPLER: {
        uint32_t s = __float_as_uint(in[0]);
        const uint32_t cnt = __float_as_uint(in[1]);
        for (uint32_t k = 0; k < cnt; k++)
            out[k] = next1d(s);
    } break;
    case RT_EVAL_CAMERA_RAY: {
        const f3 c = normalize(mk3(-in[2 * i] * cam.halfW, in[2 * i + 1] * cam.halfH, 1.0f));
        const f3 d = quatRotate(cam.qw, cam.q, c);
        float* o = out + 6 * i;
        o[0] = cam.pos.x, o[1] = cam.pos.y, o[2] = cam.pos.z, o[3] = d.x, o[4] = d.y, o[5] = d.z;
    } break;
    case RT_EVAL_SHADING: {
        const float* a = in + 20 * i;
        const f3 r = computeShading(ft, ld3(a + 8), ld3(a), ld3(a + 3), a[6], ld3(a + 11), ld3(a + 14), ld3(a + 17));
        out[3 * i + 0] = r.x, out[3 * i + 1] = r.y, out[3 * i + 2] = r.z;
    } break;
    default:
        break;
    }
}

} // namespace rt

// ----------------------------------------------------------------------------
// Launchers used by the C ABI (rt_capi.cpp, rt_capi_render.cpp)
// ----------------------------------------------------------------------------
namespace rt {

typedef void (*RenderFn)(const DevScene, const DevCamera, const DevFeat, const DevFrame, DevCounters*, float*, int, int);

static RenderFn renderVariant(int v)
{
#ifdef RT_FAST_BUILD // experiments: only the plain and the counting variant
    return (v & 8) ? renderKernel<8> : renderKernel<0>;
#endif
    if (v & V_MOTION) // one variant for every feature set: motion blur is slow by what it asks for, not by dead branches
        return (v & V_STATS) ? renderKernel<V_MOTION | V_TRANSP | V_GLOSSY | V_STATS> : renderKernel<V_MOTION | V_TRANSP | V_GLOSSY>;
    switch (v) {
    case 0: return renderKernel<0>;
    case 1: return renderKernel<1>;
    case 2: return renderKernel<2>;
    case 3: return renderKernel<3>;
    case 4: return renderKernel<4>;
    case 5: return renderKernel<5>;
    case 6: return renderKernel<6>;
    case 7: return renderKernel<7>;
    case 8: return renderKernel<8>;
    case 9: return renderKernel<9>;
    case 10: return renderKernel<10>;
    case 11: return renderKernel<11>;
    case 12: return renderKernel<12>;
    case 13: return renderKernel<13>;
    case 14: return renderKernel<14>;
    default: return renderKernel<15>;
    }
}

hipError_t renderOccupancy(int variant, int blockThreads, size_t ldsBytes, int* blocksPerCU)
{
    return hipOccupancyMaxActiveBlocksPerMultiprocessor(blocksPerCU, renderVariant(variant), blockThreads, ldsBytes);
}

hipError_t launchRender(int variant, int blocks, int blockThreads, size_t ldsBytes, hipStream_t stream, const DevScene& sc,
    const DevCamera& cam, const DevFeat& ft, const DevFrame& fr, DevCounters* ctr, float* out, int minActive, int logicRounds)
{
    hipLaunchKernelGGL(renderVariant(variant), dim3(blocks), dim3(blockThreads), ldsBytes, stream, sc, cam, ft, fr, ctr, out,
        minActive, logicRounds);
    return hipGetLastError();
}

hipError_t launchIntersect(int blockThreads, size_t ldsBytes, hipStream_t stream, const DevScene& sc, const DevFeat& ft,
    rt_ray* rays, rt_hit* hits, uint32_t n)
{
    const int blocks = (int)((n + blockThreads - 1) / blockThreads);
    hipLaunchKernelGGL(intersectKernel, dim3(blocks), dim3(blockThreads), ldsBytes, stream, sc, ft, rays, hits, n);
    return hipGetLastError();
}

hipError_t launchDeinterleave(hipStream_t stream, const float* packed, float* frame, int width, int height, int tilesX,
    int tilesY, int nShard, size_t shardStride)
{
    const size_t total = (size_t)tilesX * tilesY * 64;
    const int blocks = (int)((total + 255) / 256);
    hipLaunchKernelGGL(deinterleaveKernel, dim3(blocks), dim3(256), 0, stream, packed, frame, width, height, tilesX, tilesY,
        nShard, shardStride);
    return hipGetLastError();
}

hipError_t launchEval(hipStream_t stream, const DevScene& sc, const DevCamera& cam, const DevFeat& ft, int kind,
    const float* in, float* out, uint32_t n)
{
    const uint32_t threads = (kind == RT_EVAL_SAMPLER) ? 1u : n;
    const int blocks = (int)((threads + 63) / 64);
    hipLaunchKernelGGL(evalKernel, dim3(blocks), dim3(64), 0, stream, sc, cam, ft, kind, in, out, threads);
    return hipGetLastError();
}

} // namespace rt
