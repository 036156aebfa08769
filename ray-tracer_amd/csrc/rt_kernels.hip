// rt_kernels.hip -- gfx950 kernels of the rendering hot path.
//
// renderKernel: persistent 64-lane wavefronts.  Each lane owns one pixel at a
// time and walks that pixel's whole ray tree in the reference's depth-first
// order, because one per-pixel Sampler stream is consumed across pixel
// samples, lens samples, light samples and glossy samples
// (src/render.cpp:34-39), and how many draws a ray consumes depends on what it
// hits.  The recursion of src/recursive.cpp:25-64 is unrolled into a per-lane
// state machine with an explicit frame stack; every ray of every kind
// (camera, reflection, pass-through, glossy, shadow) goes through ONE shared
// while-while BVH traversal loop, so lanes that are in different places of
// their ray trees still traverse together.  Lanes that finish a pixel pull the
// next one from an atomic work queue of 8x8 tiles (wave-aggregated: one
// atomic per 64 pixels), and the traversal loop is left early, by ballot
// count, when too few lanes still have a ray in flight.
//
// No MFMA: nothing here is a dense contraction.  The bound is FP32 VALU issue
// under divergence plus L2 latency of dependent node fetches.

#include "rt_device.hpp"

namespace rt {

enum Phase : int {
    PH_NEWPIXEL = 0, // wants a pixel from the queue
    PH_NEXTRAY,      // generate the pixel's next camera ray, or finish the pixel
    PH_TRACE,        // ray in flight in the shared traversal loop
    PH_TRACED,       // traversal finished, result in tmax / hitPrim
    PH_LIGHT,        // direct-light loop over lights and light samples
    PH_RECURSE,      // direct light done: spawn reflection / glossy / pass-through
    PH_RETURN,       // a renderRay value is ready at `depth`
    PH_DEAD
};
enum RayKind : int { K_RADIANCE = 0, K_SHADOW_ANY = 1, K_SHADOW_CLOSEST = 2 };
enum FrameStage : int { FR_REFLECT = 0, FR_GLOSSY = 1, FR_TRANSP = 2 };

constexpr int kHitNone = -1;
constexpr float kFltMax = 3.402823466e+38f;
constexpr float kTwoPi = 6.28318530717958647692528676655900576f;

constexpr int V_TRANSP = 1, V_GLOSSY = 2, V_DOF = 4, V_STATS = 8;

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
    f3 pof; // depth of field: point of focus of the current camera ray
    // the hit being shaded
    int depth;
    f3 P, rd, N, kd, mkd, ks;
    float shin, tr;
    // direct-light loop
    int li;
    uint32_t si;
    f3 Lo, acc, lcol;
    float tprod;
    f3 curKd;
    // value handed back by PH_RETURN
    f3 value;
    // ray in flight
    f3 o, d, inv;
    float tmax, slen;
    int hitPrim, kind;
    uint32_t cur;
    int sp;
    uint32_t nrays, nInner, nTris;
};

__device__ __forceinline__ f3 rcp3(f3 d)
{
    return { __builtin_amdgcn_rcpf(d.x), __builtin_amdgcn_rcpf(d.y), __builtin_amdgcn_rcpf(d.z) };
}

template <int V>
struct Tracer {
    const DevScene& sc;
    const DevFeat& ft;
    uint32_t* stk; // this lane's column of the LDS stack: entry e at stk[e * blockDim.x]
    uint32_t stride;

    __device__ __forceinline__ uint32_t pop(Lane& L) const
    {
        if (L.sp == 0)
            return kRefNone;
        L.sp--;
        return stk[L.sp * stride];
    }
    __device__ __forceinline__ void push(Lane& L, uint32_t ref) const
    {
        stk[L.sp * stride] = ref;
        L.sp++;
    }

    // One BVHInterface::intersect call begins (src/bvh.cpp:337-358).  The root
    // box test of src/bvh.cpp:253-259 happens here, so a ray that misses the
    // whole scene never enters the traversal loop.
    __device__ __forceinline__ void launch(Lane& L, f3 o, f3 d, float tmax, int kind) const
    {
        L.o = o;
        L.d = d;
        L.inv = rcp3(d);
        L.tmax = tmax;
        L.slen = tmax;
        L.kind = kind;
        L.hitPrim = kHitNone;
        L.sp = 0;
        L.nrays++;
        L.phase = PH_TRACE;
        if (ft.accel) {
            float e;
            const f3 lo = { sc.rootLo[0], sc.rootLo[1], sc.rootLo[2] }, hi = { sc.rootHi[0], sc.rootHi[1], sc.rootHi[2] };
            if (sc.nPrims != 0 && boxTest(lo, hi, o, L.inv, kFltMax, e)) {
                L.cur = sc.rootRef;
            } else {
                L.cur = kRefNone;
                L.phase = PH_TRACED;
            }
        } else {
            // enableAccelStructure off: every primitive in order (src/bvh.cpp:344-351)
            for (uint32_t i = 0; i < sc.nPrims; i++) {
                const TriRec r = loadTri(sc.tris, i);
                float t;
                if (V & V_STATS)
                    L.nTris++;
                if (triangleTest(r, L.o, L.d, L.tmax, t)) {
                    L.tmax = t;
                    L.hitPrim = (int)i;
                }
            }
            L.cur = kRefNone;
            L.phase = PH_TRACED;
        }
    }

    // Shared while-while traversal.  `minActive`: leave as soon as fewer lanes
    // than this still have a ray in flight (0 = run every ray to completion).
    __device__ __forceinline__ void run(Lane& L, int minActive) const
    {
        if (__ballot(L.phase == PH_TRACE) == 0ull)
            return;
        for (;;) {
            if (L.phase == PH_TRACE)
                step(L);
            const int n = __popcll(__ballot(L.phase == PH_TRACE));
            if (n == 0 || n < minActive)
                break;
        }
    }

    // Inner nodes until a leaf is reached, then that leaf.
    __device__ __forceinline__ void step(Lane& L) const
    {
        {
            // ---- inner nodes: one 64-B sibling-pair record per visit
            while (!(L.cur & kLeafFlag) && L.cur != kRefNone) {
                const float4* p = sc.pairs + 4u * L.cur;
                const float4 a = p[0], b = p[1], c = p[2];
                const uint2 refs = *reinterpret_cast<const uint2*>(p + 3);
                if (V & V_STATS)
                    L.nInner++;
                float eL, eR;
                const bool hL = boxTest({ a.x, a.y, a.z }, { a.w, b.x, b.y }, L.o, L.inv, L.tmax, eL);
                const bool hR = boxTest({ b.z, b.w, c.x }, { c.y, c.z, c.w }, L.o, L.inv, L.tmax, eR);
                if (hL && hR) {
                    // nearer child first; on a tie the right child, as in src/bvh.cpp:301-307
                    const bool leftFirst = eL < eR;
                    push(L, leftFirst ? refs.y : refs.x);
                    L.cur = leftFirst ? refs.x : refs.y;
                } else if (hL) {
                    L.cur = refs.x;
                } else if (hR) {
                    L.cur = refs.y;
                } else {
                    L.cur = pop(L);
                }
            }
            // ---- leaf: every primitive of the leaf, in order (src/bvh.cpp:272-279)
            if (L.cur != kRefNone) {
                const uint32_t off = L.cur & kLeafOffsetMask;
                const uint32_t cnt = (L.cur >> 27) & 15u;
                bool done = false;
                for (uint32_t i = 0; i < cnt; i++) {
                    const TriRec r = loadTri(sc.tris, off + i);
                    float t;
                    if (V & V_STATS)
                        L.nTris++;
                    if (triangleTest(r, L.o, L.d, L.tmax, t)) {
                        L.tmax = t;
                        L.hitPrim = (int)(off + i);
                        // Binary visibility only asks whether the closest hit lies more
                        // than 1e-4 before the target (src/light.cpp:101-104); any hit
                        // that does settles it, the closest one is at least as near.
                        if (L.kind == K_SHADOW_ANY && fabsf(t - L.slen) > 0.0001f) {
                            done = true;
                            break;
                        }
                    }
                }
                L.cur = done ? kRefNone : pop(L);
            }
            if (L.cur == kRefNone)
                L.phase = PH_TRACED;
        }
    }
};

template <int V>
struct Shader {
    const DevScene& sc;
    const DevCamera& cam;
    const DevFeat& ft;
    const DevFrame& fr;
    const Tracer<V>& tracer;
    float* out;
    Frame* frames;          // per-lane private arrays (scratch)
    GlossyFrame* gframes;

    // Spheres are tested after the BVH, in order, against the current ray.t
    // (src/bvh.cpp:354-355); a sphere hit is recorded as -2 - index.
    __device__ __forceinline__ void spheres(Lane& L) const
    {
        for (int s = 0; s < sc.nSpheres; s++) {
            float t;
            if (sphereTest(sc.spheres[s], L.o, L.d, L.tmax, t)) {
                L.tmax = t;
                L.hitPrim = -2 - s;
            }
        }
    }

    __device__ __forceinline__ DevMaterial materialOfHit(int hitPrim) const
    {
        if (hitPrim >= 0) {
            const uint32_t meshID = __float_as_uint(sc.attrs[4u * (uint32_t)hitPrim + 3u].w);
            return sc.mats[meshID];
        }
        return sc.mats[sc.spheres[-2 - hitPrim].material];
    }

    // updateHitInfo (src/bvh.cpp:18-48) for the accepted closest hit, followed by
    // sampleMaterialKd (src/shading.cpp:13-24), which depends only on the hit.
    __device__ __forceinline__ void shadeSetup(Lane& L) const
    {
        L.P = L.o + L.tmax * L.d;
        L.rd = L.d;
        DevMaterial m;
        if (L.hitPrim >= 0) {
            const uint32_t i = (uint32_t)L.hitPrim;
            const TriRec r = loadTri(sc.tris, i);
            const float4 a0 = sc.attrs[4u * i + 0], a1 = sc.attrs[4u * i + 1], a2 = sc.attrs[4u * i + 2], a3 = sc.attrs[4u * i + 3];
            m = sc.mats[__float_as_uint(a3.w)];
            f3 N = r.n;
            f2 uv = { 0.0f, 0.0f };
            if (ft.normalInterp || ft.texMapping) {
                // computeBarycentricCoord, src/interpolate.cpp:13-24
                const f3 n = cross(r.v1 - r.v0, r.v2 - r.v0);
                const f3 na = cross(r.v2 - r.v1, L.P - r.v1);
                const f3 nb = cross(r.v0 - r.v2, L.P - r.v2);
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
            if (dot(L.d, r.n) > 0.0f)
                N = -N;
            L.N = N;
            L.kd = (ft.texMapping && m.tex >= 0) ? sampleTexture(sc, m.tex, uv, ft.bilinear != 0) : ld3(m.kd);
        } else {
            const DevSphere s = sc.spheres[-2 - L.hitPrim];
            m = sc.mats[s.material];
            L.N = normalize(L.P - ld3(s.center));
            L.kd = ld3(m.kd);
        }
        L.mkd = ld3(m.kd);
        L.ks = ld3(m.ks);
        L.shin = m.shininess;
        L.tr = m.transparency;
        L.li = 0;
        L.si = 0;
        L.Lo = mk3(0.0f);
        L.acc = mk3(0.0f);
        L.phase = PH_LIGHT;
    }

    // computeContributionPointLight after visibility is known (src/light.cpp:171-186),
    // then the per-light bookkeeping of src/light.cpp:205-219 / :239-253 / :283-298.
    __device__ __forceinline__ void addSample(Lane& L, f3 visible, f3 l, int type) const
    {
        if (!isZero(visible)) {
            f3 N = L.N;
            if (ft.transparency && L.tr < 1.0f && dot(L.N, l) < 0.0f)
                N = -L.N;
            L.acc = L.acc + computeShading(ft, N, L.kd, L.ks, L.shin, -L.rd, l, visible);
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
            addSample(L, col, normalize(pos - L.P), type);
            return;
        }
        // shadow ray from the light to the hit point, src/light.cpp:94-97 / :128-131
        L.lcol = col;
        const f3 toHit = L.P - pos;
        const f3 sd = normalize(toHit);
        const float len = length(toHit);
        if ((V & V_TRANSP) && ft.transparency) {
            L.tprod = 1.0f;
            L.curKd = L.mkd;
            tracer.launch(L, pos, sd, len, K_SHADOW_CLOSEST);
        } else {
            tracer.launch(L, pos, sd, len, K_SHADOW_ANY);
        }
    }

    __device__ __forceinline__ void shadowDone(Lane& L) const
    {
        const int type = sc.lights[L.li].type;
        const f3 l = -L.d; // normalize(light - p) == -normalize(p - light), exactly
        if (L.kind == K_SHADOW_ANY) {
            // src/light.cpp:101-107
            const bool visible = (L.hitPrim == kHitNone) ? true : (fabsf(L.tmax - L.slen) <= 0.0001f);
            addSample(L, visible ? L.lcol : mk3(0.0f), l, type);
            return;
        }
        // src/light.cpp:138-150
        if (V & V_TRANSP) {
            const f3 hp = L.o + L.tmax * L.d;
            bool again = false;
            if (L.hitPrim != kHitNone && !allEpsEq(hp, L.P, 0.0001f)) {
                const DevMaterial hm = materialOfHit(L.hitPrim);
                if (hm.transparency != 1.0f) {
                    L.curKd = ld3(hm.kd);
                    L.tprod *= hm.transparency;
                    again = true;
                }
            }
            if (again) {
                tracer.launch(L, hp + L.d * 1.0e-5f, L.d, kFltMax, K_SHADOW_CLOSEST); // src/recursive.cpp:98-111
                return;
            }
            f3 visible = mk3(0.0f);
            if (allEpsEq(hp, L.P, 0.0001f))
                visible = L.lcol * L.curKd * L.tprod;
            // l must be the direction to the light sample; pass-through rays keep the direction
            addSample(L, visible, l, type);
        }
    }

    // src/recursive.cpp:72-89
    __device__ __forceinline__ void reflectionRay(const f3& P, const f3& rd, const f3& N, f3& o, f3& d) const
    {
        const f3 nd = normalize(rd);
        d = normalize(nd - 2.0f * dot(nd, N) * N);
        o = P + N * 1.0e-5f;
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
            const f3 sU = g.u * r * cosf(theta);
            const f3 sV = g.v * r * sinf(theta);
            const f3 dir = normalize(g.R + sU + sV);
            L.depth = fi + 1;
            tracer.launch(L, f.P + dir * 1.0e-5f, dir, kFltMax, K_RADIANCE);
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
            tracer.launch(L, f.P + f.rd * 1.0e-5f, f.rd, kFltMax, K_RADIANCE); // src/recursive.cpp:98-111
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
            const bool reflective = !(L.ks.x == 0.0f && L.ks.y == 0.0f && L.ks.z == 0.0f);
            const bool transparent = L.tr != 1.0f;
            const bool wantRefl = ft.reflections && reflective;
            const bool wantTr = (V & V_TRANSP) && ft.transparency && transparent;
            if (wantRefl || wantTr) {
                Frame& f = frames[fi];
                f.Lo = L.Lo;
                f.ks = L.ks;
                f.P = L.P;
                f.rd = L.rd;
                f.tr = L.tr;
                if (wantRefl && !ft.glossy) {
                    f.stage = FR_REFLECT;
                    f3 o, d;
                    reflectionRay(L.P, L.rd, L.N, o, d);
                    L.depth = fi + 1;
                    tracer.launch(L, o, d, kFltMax, K_RADIANCE);
                    return;
                }
                if ((V & V_GLOSSY) && wantRefl && ft.glossy) { // src/extra.cpp:329-368
                    f.stage = FR_GLOSSY;
                    GlossyFrame& g = gframes[fi];
                    f3 o;
                    reflectionRay(L.P, L.rd, L.N, o, g.R);
                    f3 arbitrary = { 1.0f, 0.0f, 0.0f };
                    if (length(arbitrary - g.R) < 0.01f)
                        arbitrary = { 0.0f, 1.0f, 0.0f };
                    g.u = normalize(cross(arbitrary, g.R));
                    g.v = normalize(cross(g.R, g.u));
                    g.radius = ft.glossyExponent * L.shin / 64.0f;
                    g.acc = mk3(0.0f);
                    g.i = 0;
                    if ((int)ft.glossySamples > 0) {
                        glossyLaunch(L, fi);
                        return;
                    }
                    const f3 Lo = L.Lo + (g.acc / (float)(int)ft.glossySamples) * L.ks;
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

    // Trackball::generateRay, framework/src/trackball.cpp:101-110
    __device__ __forceinline__ f3 cameraDir(float ndcX, float ndcY) const
    {
        const f3 c = normalize(mk3(-ndcX * cam.halfW, ndcY * cam.halfH, 1.0f));
        return quatRotate(cam.qw, cam.q, c);
    }

    // generatePixelRays* (src/render.cpp:56-131) and the lens fan-out of
    // src/extra.cpp:48-81, one ray at a time: the pixel-sample draws, the lens
    // draws and the draws of the ray trees are three consecutive stretches of
    // the pixel's stream, so each gets its own cursor.
    __device__ __forceinline__ void nextRay(Lane& L) const
    {
        if (L.rayI >= ft.raysPerPixel) {
            const f3 c = L.Lpix / (float)ft.raysPerPixel; // src/recursive.cpp:16
            const size_t idx = fr.packed ? (size_t)L.widx : ((size_t)(fr.height - 1 - L.py) * fr.width + L.px);
            out[3 * idx + 0] = c.x;
            out[3 * idx + 1] = c.y;
            out[3 * idx + 2] = c.z;
            L.phase = PH_NEWPIXEL;
            return;
        }
        const uint32_t dofN = ((V & V_DOF) && ft.dof) ? ft.dofSamples : 1u;
        const uint32_t c = L.rayI / dofN;
        f3 dir;
        if (!((V & V_DOF) && ft.dof) || (L.rayI % dofN) == 0) {
            float ndcX, ndcY;
            const float W = (float)fr.width, H = (float)fr.height;
            if (ft.spp > 1) {
                if (ft.jitter) {
                    const int p = (int)(c / ft.jitterN), q = (int)(c % ft.jitterN);
                    const float a = next1d(L.sPix);
                    const float b = next1d(L.sPix);
                    const float jx = (float)L.px + ((float)p + a) / (float)ft.jitterN;
                    const float jy = (float)L.py + ((float)q + b) / (float)ft.jitterN;
                    ndcX = jx / W * 2.0f - 1.0f;
                    ndcY = jy / H * 2.0f - 1.0f;
                } else {
                    const float a = next1d(L.sPix);
                    const float b = next1d(L.sPix);
                    ndcX = ((float)L.px + a) / W * 2.0f - 1.0f;
                    ndcY = ((float)L.py + b) / H * 2.0f - 1.0f;
                }
            } else {
                ndcX = ((float)L.px + 0.5f) / W * 2.0f - 1.0f;
                ndcY = ((float)L.py + 0.5f) / H * 2.0f - 1.0f;
            }
            dir = cameraDir(ndcX, ndcY);
            if ((V & V_DOF) && ft.dof) { // getPointOfFocus, src/extra.cpp:15-23
                const float cosAngle = dot(dir, cam.fwd);
                const float curDistance = ft.dofDistance / cosAngle;
                L.pof = cam.pos + curDistance * dir;
            }
        }
        L.depth = 0;
        if ((V & V_DOF) && ft.dof) { // src/extra.cpp:27-44, :70-76
            const float r0 = next1d(L.sLens);
            const float r1 = next1d(L.sLens);
            const float angle = r0 * kTwoPi;
            const float dist = sqrtf(r1) * (ft.dofDiameter * 0.5f);
            const float ox = cosf(angle) * dist;
            const float oy = sinf(angle) * dist;
            const f3 o = cam.pos + ox * cam.up + oy * cam.left;
            tracer.launch(L, o, normalize(L.pof - o), kFltMax, K_RADIANCE);
        } else {
            tracer.launch(L, cam.pos, dir, kFltMax, K_RADIANCE);
        }
    }

    // Run this lane's state machine until it has a ray in flight, wants a new
    // pixel, or is dead.
    __device__ __forceinline__ void advance(Lane& L) const
    {
        for (;;) {
            switch (L.phase) {
            case PH_NEXTRAY:
                nextRay(L);
                break;
            case PH_TRACED:
                if (sc.nSpheres)
                    spheres(L);
                if (L.kind == K_RADIANCE) {
                    if (L.hitPrim == kHitNone) { // src/recursive.cpp:29-32
                        L.value = sampleEnvironmentMap(sc, ft, L.d);
                        L.phase = PH_RETURN;
                    } else {
                        shadeSetup(L);
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
__global__ __launch_bounds__(256, RT_MIN_WAVES) void renderKernel(const DevScene sc, const DevCamera cam, const DevFeat ft,
    const DevFrame fr, DevCounters* __restrict__ ctr, float* __restrict__ out, int minActive, int logicRounds)
{
    extern __shared__ uint32_t stackMem[];
    Frame frames[kMaxDepth + 1];
    GlossyFrame gframes[(V & V_GLOSSY) ? (kMaxDepth + 1) : 1];

    const Tracer<V> tracer { sc, ft, stackMem + threadIdx.x, blockDim.x };
    const Shader<V> shader { sc, cam, ft, fr, tracer, out, frames, gframes };

    Lane L;
    L.phase = PH_NEWPIXEL;
    L.nrays = L.nInner = L.nTris = 0;
    L.cur = kRefNone;
    L.sp = 0;
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
                        base = atomicAdd(&ctr->next, 64u);
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
                    const uint32_t tileLocal = w >> 6, within = w & 63u;
                    const uint32_t tile = tileLocal * (uint32_t)fr.nShard + (uint32_t)fr.shard;
                    const int tx = (int)(tile % (uint32_t)fr.tilesX), ty = (int)(tile / (uint32_t)fr.tilesX);
                    L.px = tx * kTile + (int)(within & 7u);
                    L.py = ty * kTile + (int)(within >> 3);
                    L.widx = w;
                    if (L.px < fr.width && L.py < fr.height) {
                        const uint32_t seed = (uint32_t)(fr.height * L.px + L.py); // src/render.cpp:38
                        L.sPix = seed;
                        L.sLens = ft.lensMul * seed + ft.lensAdd;
                        L.sMain = ft.mainMul * seed + ft.mainAdd;
                        L.rayI = 0;
                        L.Lpix = mk3(0.0f);
                        L.phase = PH_NEXTRAY;
                    }
                    // pixels of a partial edge tile that fall outside the frame are skipped:
                    // the lane stays in PH_NEWPIXEL and takes part in the next hand-out
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
        tracer.run(L, exhausted ? 0 : minActive);
    }

    // ---- counters: one atomic per wave
    unsigned long long r = L.nrays, ni = L.nInner, nt = L.nTris;
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
__global__ __launch_bounds__(256) void intersectKernel(const DevScene sc, const DevFeat ft, rt_ray* __restrict__ rays,
    rt_hit* __restrict__ hits, uint32_t n)
{
    extern __shared__ uint32_t stackMem[];
    const Tracer<0> tracer { sc, ft, stackMem + threadIdx.x, blockDim.x };
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    Lane L;
    L.phase = PH_DEAD;
    L.nrays = 0;
    L.cur = kRefNone;
    L.sp = 0;
    if (i < n) {
        const rt_ray r = rays[i];
        tracer.launch(L, ld3(r.origin), ld3(r.direction), r.t, K_RADIANCE);
    }
    tracer.run(L, 0);
    if (i >= n)
        return;
    for (int s = 0; s < sc.nSpheres; s++) {
        float t;
        if (sphereTest(sc.spheres[s], L.o, L.d, L.tmax, t)) {
            L.tmax = t;
            L.hitPrim = -2 - s;
        }
    }
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
// Launchers used by the C ABI (rt_capi.cpp)
// ----------------------------------------------------------------------------
namespace rt {

typedef void (*RenderFn)(const DevScene, const DevCamera, const DevFeat, const DevFrame, DevCounters*, float*, int, int);

static RenderFn renderVariant(int v)
{
#ifdef RT_FAST_BUILD // experiments: only the plain and the counting variant
    return (v & 8) ? renderKernel<8> : renderKernel<0>;
#endif
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
