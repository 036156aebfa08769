// rt_wavefront.hip -- the queue-based (wavefront) form of renderImage for gfx950: mirror reflections,
// transparency, glossy reflections (glossyTreeKernel) and motion blur (the MOTION variants; rt_wavefront.hpp).
//
// The reference consumes ONE Sampler stream per pixel, depth first
// (src/render.cpp:34-39).  Which rays a pixel's ray tree contains, though, does
// not depend on the draws: camera rays take their draws from fixed positions at
// the head of the stream, reflection and pass-through directions are pure geometry, and only
// the light samples (src/light.cpp:205-253) draw from the rest, a fixed number
// D per shaded hit.  So the position of every draw is known once the hits of
// the pixel are counted: hit number k of the pixel (camera rays in order, each
// chain root first) owns draws [main + D*k, main + D*(k+1)).  That turns the
// frame into independent work items, each on its own lane:
//
//   1. cameraKernel  one lane per camera ray: generate it, test it against the scene's
//                    box, finish the ones that miss it and queue the others;
//      pathKernel    the queued rays: trace each and the rays its hits spawn (the mirror
//                    chain, with transparency the tree of mirror and pass-through rays,
//                    src/recursive.cpp:25-64), append one node per hit;
//   2. prefixKernel  per pixel, exclusive prefix of its paths' hit counts;
//   3. shadowGenKernel    one lane per (hit, light sample): jump the LCG to the sample's
//                         draws, sample the light, build the shadow ray, shade the sample;
//      shadowTraceKernel  trace the shadow rays (any-hit; closest-hit through
//                         non-opaque surfaces with transparency), clear what is occluded;
//      shadowShadeKernel  with transparency only, where the light is tinted on its way:
//                         shade the samples after the traversal (src/light.cpp:166-186);
//   4. combineKernel per pixel: sums, clamp and mirror / pass-through combine in exactly
//                    the reference's order of operations, then the pixel mean.
//
// pathKernel and shadowTraceKernel are persistent 64-lane wavefronts pulling batches of
// work items from an atomic queue; lanes whose ray has finished take their next
// item while the others keep traversing (rt_trace.hpp).  All arithmetic is the
// same as in the per-pixel kernel of rt_kernels.hip, in the same order, so
// both produce the same bits.

#include <cstdio>
#include <cstdlib>
#include "rt_trace.hpp"
#include "rt_wavefront.hpp"

namespace rt {

// state after k draws from `state` (src/sampler.h:18), O(log k)
__device__ __forceinline__ uint32_t lcgSkip(uint32_t state, uint32_t k)
{
    uint32_t mul = 747796405u, add = 2891336453u;
    while (k) {
        if (k & 1u)
            state = state * mul + add;
        add = add * mul + add;
        mul = mul * mul;
        k >>= 1;
    }
    return state;
}

// Wave-level queue: every lane whose `want` is set receives the next work item
// in `item`, or is marked dead when the queue is empty.  All members are wave-uniform.
// One address takes fewer than a hundred atomics per microsecond, so a round of several
// thousand waves asking at once costs more than a small frame's whole traversal.  Hence:
// the head of the queue is each wave's own by position (wave w owns [w*own, (w+1)*own), no
// atomic), the rest is handed out in batches by the atomic counter, and the first wave to find
// it empty raises a flag (in another cache line) that the others read instead of asking the
// counter themselves.  `own` is one batch by default: a larger positional share was measured
// and lost more to uneven finishing times (rays differ widely in cost) than it saved in atomics.
struct WaveQueue {
    uint32_t cursor = 0, cend = 0;
    uint32_t shared = 0; // where the atomically handed-out part begins
    uint32_t chunk = 0;  // its batch size
    bool exhausted = false;

    // sixteenths: the part of the queue split up by position; configured/least: bounds of the batch size
    __device__ __forceinline__ void start(uint32_t total, uint32_t sixteenths, uint32_t configured, uint32_t least)
    {
        const uint32_t wavesPerBlock = blockDim.x / 64u;
        const uint32_t nWaves = gridDim.x * wavesPerBlock;
        const uint32_t wave = blockIdx.x * wavesPerBlock + (threadIdx.x >> 6);
        uint32_t own = (uint32_t)(((unsigned long long)total * sixteenths / 16u) / nWaves);
        if (own < least)
            own = least;
        shared = nWaves * own; // below total + 2^26 with the batch sizes in use
        // a few batches per wave, so that the waves still finish close together
        const uint32_t rest = shared < total ? total - shared : 0u;
        const uint32_t share = rest / (nWaves * 2u);
        chunk = share >= configured ? configured : (share < least ? least : share);
        cursor = wave * own;
        cend = (cursor + own < total) ? cursor + own : total;
        if (cursor >= total) {
            cursor = cend = 0;
            exhausted = true;
        }
    }

    __device__ __forceinline__ void fetch(bool& want, uint32_t& item, bool& dead, HotCounter* counter, HotCounter* dry, uint32_t total,
        uint32_t lane)
    {
        for (;;) {
            const unsigned long long wm = __ballot(want);
            if (wm == 0ull)
                return;
            if (cursor == cend) {
                if (!exhausted && (shared >= total || __hip_atomic_load(&dry->v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u))
                    exhausted = true;
                if (exhausted) {
                    if (want) {
                        dead = true;
                        want = false;
                    }
                    return;
                }
                uint32_t base = 0;
                const int leader = __builtin_ctzll(wm);
                if (lane == (uint32_t)leader)
                    base = atomicAdd(&counter->v, chunk);
                base = shared + __shfl(base, leader);
                if (base >= total) {
                    if (lane == (uint32_t)leader)
                        __hip_atomic_store(&dry->v, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    exhausted = true;
                    continue;
                }
                cursor = base;
                cend = (base + chunk < total) ? base + chunk : total;
            }
            const uint32_t rank = __popcll(wm & ((1ull << lane) - 1ull));
            const uint32_t avail = cend - cursor;
            const uint32_t nWant = __popcll(wm);
            if (want && rank < avail) {
                item = cursor + rank;
                want = false;
            }
            cursor += (nWant < avail) ? nWant : avail;
        }
    }
};

__device__ __forceinline__ unsigned long long waveSum(unsigned long long v)
{
    for (int off = 32; off > 0; off >>= 1)
        v += __shfl_down(v, off);
    return v;
}

// ---------------------------------------------------------------------------
// 1a. camera rays, one lane per ray: generatePixelRays* and the root box test of
//     BVHInterface::intersect.  With the default camera most rays never enter
//     the scene's box; those are finished here (a miss, src/recursive.cpp:29-32)
//     and only the others are queued for the traversal kernel, so its
//     wavefronts spend their time on rays that have something to traverse.
// ---------------------------------------------------------------------------
#ifndef RT_CAM_PER_LANE
#define RT_CAM_PER_LANE 4
#endif
constexpr uint32_t kCamPerLane = RT_CAM_PER_LANE; // camera rays per lane: one queue reservation per 256 times as many rays

// MOTION: a path belongs to a (pixel, time sample) pair.  The pair's time is draw m + 1 of the pixel's stream and the
// pair is rendered with the stream as it stands after that draw (src/extra.cpp:145-204); the scene's box moves with
// the time, so every ray is queued and the traversal kernel does the root test on the moved copy.
template <bool DOF, bool MOTION = false>
__global__ __launch_bounds__(256) void cameraKernel(const DevScene sc, const DevCamera cam, const DevFeat ft, const DevFrame fr,
    const WfParams wp, const WfBuffers wb, DevCounters* __restrict__ ctr)
{
    __shared__ uint32_t waveCount[kCamPerLane][4];
    __shared__ uint32_t blockBase;
    const Tracer<false> tracer { sc, ft.accel, (LdsWords)0 };
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const uint32_t M = MOTION ? (uint32_t)ft.motionSamples : 1u; // time samples per pixel
    const uint32_t total = wp.nSlots * M * wp.R;
    bool queue[kCamPerLane], enters[kCamPerLane];
    f3 o[kCamPerLane], d[kCamPerLane];
    float when[MOTION ? kCamPerLane : 1u];
    unsigned long long qm[kCamPerLane];
#pragma unroll
    for (uint32_t j = 0; j < kCamPerLane; j++) {
        const uint32_t path = (blockIdx.x * kCamPerLane + j) * 256u + threadIdx.x;
        queue[j] = enters[j] = false;
        o[j] = d[j] = mk3(0.0f);
        if (path < total) {
            const uint32_t pixel = fastDiv(path, wp.divR); // (MOTION: the sample slot)
            const uint32_t i = path - pixel * wp.R;
            uint32_t pslot = pixel, m = 0;
            if (MOTION) {
                pslot = pixel / M;
                m = pixel - pslot * M;
            }
            const uint32_t slot = wp.slotBegin + pslot;
            int px, py;
            if (slotToPixel(fr, slot, px, py)) { // padding pixels of partial edge tiles have no rays
                // A pixel from which the scene's box cannot be seen: every ray of it misses, and unless the
                // background has a colour of its own there is nothing to work out for it.
                // (combineKernel knows the same rectangle and writes such a pixel's black without reading anything)
                if (!ft.envMap && !seesScene(fr, px, py)) {
                } else {
                uint32_t seed = (uint32_t)(fr.height * px + py); // src/render.cpp:38
                if (MOTION) {
                    seed = lcgSkip(seed, m);
                    when[j] = next1d(seed);
                }
                const uint32_t dofN = (DOF && ft.dof) ? ft.dofSamples : 1u;
                const uint32_t sample = (DOF && ft.dof) ? fastDiv(i, ft.divDof) : i; // i / dofN
                uint32_t sPix = (ft.spp > 1) ? lcgSkip(seed, 2u * sample) : seed;
                uint32_t sLens = (DOF && ft.dof) ? lcgSkip(seed, wp.pixDraws + 2u * i) : seed;
                f3 pof = { 0.0f, 0.0f, 0.0f };
                cameraRay<DOF>(cam, ft, fr, px, py, sample * dofN, sPix, sLens, pof, o[j], d[j]);
                // (the root test is widened against rounding; the host sized the ray queue and the hit pools by the pixels
                // of the rectangle alone, and no ray from outside it can enter the box itself)
                enters[j] = !MOTION && ft.accel && seesScene(fr, px, py) && tracer.entersScene(o[j], rcp3(d[j]));
                // without the acceleration structure, or with spheres in the scene, every ray has work left
                queue[j] = MOTION || enters[j] || !ft.accel || sc.nSpheres != 0;
                if (!queue[j]) {
                    wb.pathHead[path] = kPathMiss;
                    wb.pathCount[path] = 0;
                    if (ft.envMap) {
                        const f3 env = sampleEnvironmentMap(sc, ft, d[j]);
                        wb.pathEnv[3u * path + 0] = env.x;
                        wb.pathEnv[3u * path + 1] = env.y;
                        wb.pathEnv[3u * path + 2] = env.z;
                    }
                }
                }
            }
        }
        qm[j] = __ballot(queue[j]);
        if (lane == 0)
            waveCount[j][wave] = (uint32_t)__popcll(qm[j]);
    }
    // one queue reservation per block
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t n = 0;
        for (uint32_t j = 0; j < kCamPerLane; j++)
            n += waveCount[j][0] + waveCount[j][1] + waveCount[j][2] + waveCount[j][3];
        blockBase = n ? atomicAdd(&ctr->camQueued.v, n) : 0u;
    }
    __syncthreads();
    uint32_t before = blockBase; // queued rays of the block before round j, wave `wave`
#pragma unroll
    for (uint32_t j = 0; j < kCamPerLane; j++) {
        uint32_t idx = before + (uint32_t)__popcll(qm[j] & ((1ull << lane) - 1ull));
        for (uint32_t w = 0; w < 4u; w++) {
            const uint32_t c = waveCount[j][w];
            if (w < wave)
                idx += c;
            before += c;
        }
        if (queue[j]) {
            const uint32_t path = (blockIdx.x * kCamPerLane + j) * 256u + threadIdx.x;
            wb.camRays[2u * (size_t)idx + 0] = make_float4(o[j].x, o[j].y, o[j].z, __uint_as_float(path));
            wb.camRays[2u * (size_t)idx + 1] = make_float4(d[j].x, d[j].y, d[j].z, MOTION ? when[MOTION ? j : 0u] : (enters[j] ? 1.0f : 0.0f));
        }
    }
}

// ---------------------------------------------------------------------------
// 1b. the queued camera rays and their mirror chains
// ---------------------------------------------------------------------------
#ifdef RT_PATH_WAVES // experiments: force the register budget of this many waves per SIMD
#define RT_PATH_ATTR __attribute__((amdgpu_waves_per_eu(RT_PATH_WAVES, RT_PATH_WAVES)))
#else
// (four waves per SIMD is what the kernel's registers allowed before the check of Tracer::surelyReached came to
// sit in its shading path; held to that, it keeps them for two spilled registers outside the loops)
#ifndef RT_MOTION_PATH_WAVES
#define RT_MOTION_PATH_WAVES 1 // (the motion-blur variant: 1 = whatever its registers leave)
#endif
#define RT_PATH_ATTR __attribute__((amdgpu_waves_per_eu(STATS ? 1 : (MOTION ? RT_MOTION_PATH_WAVES : 4))))
#endif
// MOTION: the queued ray's record carries its time; the ray and the mirror chain behind it meet the copy of the scene
// moved to that time (Tracer<., true>, RayState::mv).
// Hits that take the reference's own walk over the caller's nodes to decide (two primitives at one distance, a hit its
// box routine may have lost: Tracer::settleTie) are one in millions, but the out-of-line function that makes the walk
// costs the kernel that calls it registers all the time (the compiler keeps what the callee uses free across the
// call): 1.5-1.9 % of the headline frame, a sixth of a frame with motion blur, measured by leaving the call out.
// RT_DEFER_TIES=1 is the experiment that tried to have that back: without transparency (a chain of hits, which can be
// picked up again from its records alone) the kernel makes no such call; a path that meets such a hit is handed over
// -- its camera ray's index goes on a list (in the memory of WfBuffers::contrib, unused until the light samples are
// made) -- and the EXACT variant of the kernel, launched behind it over that list, finds where the path stopped from
// the nodes it left, answers that ray and every later one of the path by the reference's walk (Tracer::launchExact)
// and writes the rest of the chain.  Bit-equal (every GPU test, 1 200 random configurations), and slower: the handful
// of walks a frame needs, hidden among thousands of busy waves when made on the spot, stand alone at the end of the
// path stage and take 0.26 ms there (headline frame 5.70 ms against 5.55).  Off.
#ifndef RT_DEFER_TIES
#define RT_DEFER_TIES 0
#endif
template <bool STATS, bool TRANSP, bool MOTION = false, bool EXACT = false>
__global__ __launch_bounds__(kBlockThreads) RT_PATH_ATTR void pathKernel(const DevScene sc, const DevFeat ft, const WfParams wp, const WfBuffers wb,
    DevCounters* __restrict__ ctr, int minActive)
{
    static_assert(!EXACT || !TRANSP, "a tree of hits is not picked up again from its records");
    constexpr bool DEFER = RT_DEFER_TIES && !TRANSP && !EXACT;
    extern __shared__ uint32_t stackMem[];
    const Tracer<STATS, MOTION> tracer { sc, ft.accel, (LdsWords)(stackMem + threadIdx.x) };
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t total = EXACT ? ctr->pathRedo.v : ctr->camQueued.v;
    uint32_t* handedOver = reinterpret_cast<uint32_t*>(wb.contrib); // (at most one entry per queued camera ray: 4 B against 12 B per light sample)
    uint32_t mine = blockIdx.x * blockDim.x + threadIdx.x;           // EXACT: this lane's next entry of the list
    // node slots a wave reserves per atomic: about what its share of the rays will fill, so that a small
    // frame does not leave most of the pool's touched part empty (every slot is a work item further on)
    uint32_t nodeBlock = wp.nodeBlockMax;
    while (wp.nodeBlockFixed == 0u && nodeBlock > 64u && nodeBlock * (gridDim.x * (blockDim.x / 64u)) > 2u * total)
        nodeBlock >>= 1;

    RayState r;
    r.active = false;
    r.cur = kRefNone;
    r.pend = 0u;
    r.top = tracer.stk;
    r.tos = kRefNone;
    r.nInner = r.nTris = 0;
    r.itInner = r.itLeaf = r.itRuns = 0;
    r.topHits[0] = r.topHits[1] = r.topHits[2] = 0;
    r.cycInner = r.cycLeaf = r.cycRun = 0;
    const long long tKernel = STATS ? clock64() : 0;
    const long long tWall = STATS ? wall_clock64() : 0; // 100 MHz, the same on every XCD (the shader clock is not)
    uint32_t path = 0, item = 0, nrays = 0, made = 0;
    __shared__ uint32_t nodeBlkAll[kBlockWaves][2]; // per wave: next free slot and end of its block of node slots
    volatile __attribute__((address_space(3))) uint32_t* nodeBlk = (volatile __attribute__((address_space(3))) uint32_t*)nodeBlkAll[threadIdx.x >> 6];
    if (lane == 0)
        nodeBlk[0] = nodeBlk[1] = 0u;
    // With transparency a hit can have two children.  The mirror child is followed first, like
    // renderRay does; the hit itself waits on this per-lane stack (entry-major in LDS) for its
    // pass-through ray, which is rebuilt from the hit's record when its turn comes.
    __shared__ uint32_t waitingAll[TRANSP ? (kMaxDepth + 1) * kBlockThreads : 1];
    uint32_t* waiting = waitingAll + (TRANSP ? threadIdx.x : 0u);
    int nWaiting = 0;
    int depth = 0, prev = -1, count = 0;
    bool viaPass = false; // the ray in flight is the pass-through child of `prev` (else its mirror child)
    bool have = false, dead = false;
    WaveQueue q;
    q.start(total, wp.ownShare, wp.chunkPath, wp.leastPath);

    for (;;) {
        // ---- lanes without a path take the next queued camera ray
        for (int round = 0; round < 2; round++) {
            bool want = !have && !dead;
            if (EXACT) { // (a handful of entries: by position)
                if (want) {
                    want = false;
                    if (mine < total) {
                        item = handedOver[mine];
                        mine += gridDim.x * blockDim.x;
                    } else {
                        dead = true;
                    }
                }
            } else {
                q.fetch(want, item, dead, &ctr->next, &ctr->dry[0], total, lane);
            }
            if (!have && !dead && !want) {
                const float4 c0 = wb.camRays[2u * (size_t)item + 0], c1 = wb.camRays[2u * (size_t)item + 1];
                path = __float_as_uint(c0.w);
                if (EXACT) {
                    // where the path stopped: behind the last node of the chain it left (none: at its camera ray)
                    if (MOTION)
                        r.mv = makeMover(c1.w, ft.movement);
                    f3 o = { c0.x, c0.y, c0.z }, d = { c1.x, c1.y, c1.z };
                    depth = 0;
                    prev = -1;
                    count = 0;
                    for (int32_t cur = wb.pathHead[path]; cur >= 0 && depth <= kMaxDepth; cur = wb.nodeB[cur].childR) {
                        prev = cur;
                        depth++;
                        count++;
                    }
                    if (prev >= 0) {
                        const float4* a = wb.nodeA + 4u * (size_t)prev;
                        const float4 a0 = a[0], a1 = a[1], a2 = a[2];
                        reflectionRay({ a0.x, a0.y, a0.z }, { a1.z, a1.w, a2.x }, { a0.w, a1.x, a1.y }, o, d);
                    }
                    tracer.launchExact(r, o, d, kFltMax, K_RADIANCE);
                } else if (MOTION) {
                    r.mv = makeMover(c1.w, ft.movement);
                    tracer.launch(r, { c0.x, c0.y, c0.z }, { c1.x, c1.y, c1.z }, kFltMax, K_RADIANCE);
                } else if (c1.w != 0.0f)
                    tracer.launchInside(r, { c0.x, c0.y, c0.z }, { c1.x, c1.y, c1.z }, kFltMax, K_RADIANCE);
                else
                    tracer.launch(r, { c0.x, c0.y, c0.z }, { c1.x, c1.y, c1.z }, kFltMax, K_RADIANCE);
                have = true;
                if (!EXACT) {
                    depth = 0;
                    prev = -1;
                    count = 0;
                }
                viaPass = false;
                nWaiting = 0;
            }
            // ---- paths whose ray has finished: record the hit, go on with a child ray or retire
            while (have && !r.active) {
                if (DEFER) {
                    if (tracer.doubtful(r)) { // the path is handed over as it stands
                        handedOver[atomicAdd(&ctr->pathRedo.v, 1u)] = item;
                        if (count == 0)
                            wb.pathHead[path] = kPathMiss; // (no node yet: says so to whoever picks the path up)
                        have = false;
                        break;
                    }
                } else if (!EXACT) {
                    tracer.settleTie(r);
                }
                if (sc.nSpheres)
                    tracer.spheres(r);
                bool childLaunched = false;
                if (r.hitPrim == kHitNone) { // src/recursive.cpp:29-32
                    const f3 env = sampleEnvironmentMap(sc, ft, r.d);
                    if (depth == 0) {
                        wb.pathHead[path] = kPathMiss;
                        if (ft.envMap) {
                            wb.pathEnv[3u * path + 0] = env.x;
                            wb.pathEnv[3u * path + 1] = env.y;
                            wb.pathEnv[3u * path + 2] = env.z;
                        }
                    } else {
                        NodeB& pb = wb.nodeB[prev];
                        if (viaPass) {
                            pb.childP = kChildMiss;
                            pb.cvP[0] = env.x, pb.cvP[1] = env.y, pb.cvP[2] = env.z;
                        } else {
                            pb.childR = kChildMiss;
                            pb.cvR[0] = env.x, pb.cvR[1] = env.y, pb.cvR[2] = env.z;
                        }
                    }
                } else {
                    const HitCtx h = shadeHit<MOTION>(sc, ft, r);
                    // One node per shaded hit.  Every wave owns a block of nodeBlock (>= 64) consecutive slots of
                    // the pool and hands them out itself; only a new block costs an atomic (one address
                    // takes well under a hundred of them per microsecond, far fewer than there are hits).
                    const unsigned long long am = __ballot(true);
                    const uint32_t n = (uint32_t)__popcll(am), rank = (uint32_t)__popcll(am & ((1ull << lane) - 1ull));
                    // (kept in LDS: only some lanes are here, all of the wave must see the update)
                    const uint32_t nodeCur = nodeBlk[0], nodeEnd = nodeBlk[1];
                    const uint32_t rem = nodeEnd - nodeCur;
                    uint32_t fresh = 0;
                    if (n > rem) {
                        const int leader = __builtin_ctzll(am);
                        if (lane == (uint32_t)leader)
                            fresh = atomicAdd(&ctr->nodes.v, nodeBlock);
                        fresh = __shfl(fresh, leader);
                    }
                    const uint32_t idx = rank < rem ? nodeCur + rank : fresh + (rank - rem);
                    if (rank == 0) {
                        nodeBlk[0] = n > rem ? fresh + (n - rem) : nodeCur + n;
                        if (n > rem)
                            nodeBlk[1] = fresh + nodeBlock;
                    }
                    if (idx < wb.maxNodes) { // always: the pool holds the worst case of the batch
                        made++;
                        float4* a = wb.nodeA + 4u * (size_t)idx;
                        a[0] = make_float4(h.P.x, h.P.y, h.P.z, h.N.x);
                        a[1] = make_float4(h.N.y, h.N.z, r.d.x, r.d.y);
                        a[2] = make_float4(r.d.z, h.kd.x, h.kd.y, h.kd.z);
                        a[3] = make_float4(h.ks.x, h.ks.y, h.ks.z, h.shin);
                        NodeB nb;
                        nb.path = path;
                        nb.ordinal = (uint32_t)count;
                        nb.depth = depth;
                        nb.childR = nb.childP = kChildNone;
                        nb.cvR[0] = nb.cvR[1] = nb.cvR[2] = 0.0f;
                        if (TRANSP)
                            nb.cvP[0] = nb.cvP[1] = nb.cvP[2] = 0.0f;
                        else // no pass-through child: the slot carries ks, so that the combine reads one record per hit
                            nb.cvP[0] = h.ks.x, nb.cvP[1] = h.ks.y, nb.cvP[2] = h.ks.z;
                        nb.tr = h.tr;
                        nb.mkd[0] = h.mkd.x, nb.mkd[1] = h.mkd.y, nb.mkd[2] = h.mkd.z;
                        nb.pad = 0u;
                        wb.nodeB[idx] = nb;
                        if (depth == 0)
                            wb.pathHead[path] = (int32_t)idx;
                        else if (viaPass)
                            wb.nodeB[prev].childP = (int32_t)idx;
                        else
                            wb.nodeB[prev].childR = (int32_t)idx;
                        count++;
                        if (depth < ft.maxDepth) { // src/recursive.cpp:43-63
                            const bool reflective = !(h.ks.x == 0.0f && h.ks.y == 0.0f && h.ks.z == 0.0f);
                            const bool wantRefl = ft.reflections && reflective;
                            const bool wantPass = TRANSP && ft.transparency && h.tr != 1.0f;
                            if (wantRefl) {
                                if (wantPass)
                                    waiting[kBlockThreads * nWaiting++] = idx;
                                f3 o, d;
                                reflectionRay(h.P, r.d, h.N, o, d);
                                if (EXACT)
                                    tracer.launchExact(r, o, d, kFltMax, K_RADIANCE);
                                else
                                    tracer.launch(r, o, d, kFltMax, K_RADIANCE);
                                viaPass = false;
                                childLaunched = true;
                            } else if (wantPass) { // generatePassthroughRay, src/recursive.cpp:98-111
                                const f3 rd = r.d;
                                tracer.launch(r, h.P + rd * 1.0e-5f, rd, kFltMax, K_RADIANCE);
                                viaPass = true;
                                childLaunched = true;
                            }
                            if (childLaunched) {
                                nrays++;
                                prev = (int)idx;
                                depth++;
                            }
                        }
                    } else {
                        // The pool is sized for the worst case of the batch, so this cannot happen unless the
                        // host's bound is wrong.  Then the frame is invalid and says so (rt_sync fails); the
                        // hit is dropped in a way that leaves the records the later kernels follow consistent.
                        if (depth == 0)
                            wb.pathHead[path] = kPathMiss;
                        atomicAdd(&ctr->overflow, 1ull);
                    }
                }
                if (!childLaunched) {
                    // this branch of the ray tree is done: the deepest hit still waiting gets its pass-through ray
                    if (TRANSP && nWaiting > 0) {
                        const uint32_t w = waiting[kBlockThreads * --nWaiting];
                        const float4* a = wb.nodeA + 4u * (size_t)w;
                        const float4 a0 = a[0], a1 = a[1], a2 = a[2];
                        const f3 P = { a0.x, a0.y, a0.z }, rd = { a1.z, a1.w, a2.x };
                        depth = wb.nodeB[w].depth + 1;
                        prev = (int)w;
                        viaPass = true;
                        tracer.launch(r, P + rd * 1.0e-5f, rd, kFltMax, K_RADIANCE);
                        nrays++;
                    } else {
                        wb.pathCount[path] = (uint8_t)count;
                        have = false;
                        break;
                    }
                }
            }
            if (__ballot(!have && !dead) == 0ull)
                break;
        }
        if (__ballot(r.active && have) == 0ull) {
            if (__ballot(!dead) == 0ull)
                break;
            continue;
        }
        tracer.run(r, q.exhausted ? (minActive & ~0xff) : minActive);
    }

    // the unused tail of the wave's last block: slots that hold no hit
    for (uint32_t i = nodeBlk[0] + lane, e = nodeBlk[1]; i < e && i < wb.maxNodes; i += 64u)
        wb.nodeB[i].path = kDeadNode;

    // every (hit, light sample) will be one intersect call of the shadow stage
    const unsigned long long srays = ft.shadows ? waveSum(made) * wp.slots : 0ull;
    const unsigned long long rs = waveSum(nrays), ni = waveSum(r.nInner), nt = waveSum(r.nTris);
    const unsigned long long ii = waveSum(r.itInner), il = waveSum(r.itLeaf), ir = waveSum(r.itRuns);
    const unsigned long long th[3] = { waveSum(r.topHits[0]), waveSum(r.topHits[1]), waveSum(r.topHits[2]) };
    // ray counts: one pair of atomics per block that has something to add (thousands of idle
    // waves adding zero to one address would take longer than a small frame's traversal)
    __shared__ unsigned long long raySums[kBlockWaves][2];
    if (lane == 0)
        raySums[threadIdx.x >> 6][0] = rs + srays, raySums[threadIdx.x >> 6][1] = srays;
    __syncthreads();
    if (threadIdx.x == 0) {
        // reflection rays counted here; every camera ray of a real pixel is one intersect call, traced or not
        unsigned long long all = (!EXACT && blockIdx.x == 0) ? (unsigned long long)wp.validPaths : 0ull, shadow = 0ull;
        for (uint32_t w = 0; w < kBlockWaves; w++)
            all += raySums[w][0], shadow += raySums[w][1];
        if (all)
            atomicAdd(&ctr->rays, all);
        if (shadow)
            atomicAdd(&ctr->sRays, shadow);
    }
    if (lane == 0) {
        if (STATS && !EXACT) {
            atomicAdd(&ctr->inner, ni);
            atomicAdd(&ctr->tris, nt);
            atomicAdd(&ctr->loopIters[0][0], ii);
            atomicAdd(&ctr->loopIters[0][1], il);
            atomicAdd(&ctr->loopIters[0][2], ir);
            for (int k = 0; k < 3; k++)
                atomicAdd(&ctr->topHits[0][k], th[k]);
            atomicAdd(&ctr->cycles[0][0], r.cycInner);
            atomicAdd(&ctr->cycles[0][1], r.cycLeaf);
            atomicAdd(&ctr->cycles[0][2], r.cycRun);
            atomicAdd(&ctr->cycles[0][3], (unsigned long long)(clock64() - tKernel));
            atomicMax(&ctr->maxCycles[0], (unsigned long long)(clock64() - tKernel));
            if (ir)
                atomicAdd(&ctr->busyWaves[0], 1ull);
            const uint32_t waveId = blockIdx.x * (blockDim.x / 64u) + (threadIdx.x >> 6);
            if (waveId < 8192u)
                ctr->waveLog[0][waveId][0] = (unsigned long long)tWall, ctr->waveLog[0][waveId][1] = (unsigned long long)wall_clock64();
        }
    }
}

// ---------------------------------------------------------------------------
// 1c. glossy reflections (src/extra.cpp:329-368): the ray tree of a whole pixel, one lane per pixel.
//     A glossy bounce fans out into numGlossySamples children whose directions are DRAWN, each child's
//     two draws coming after everything its elder sibling's subtree drew -- so which rays a pixel has
//     depends on the draws, and every ray of the pixel on all the rays before it in the recursion's
//     order (src/recursive.cpp:25-64): that order is walked here, ray by ray, by one lane.  But only the
//     rays that carry radiance (camera and glossy rays) are bound that way.  What a hit draws for its
//     light samples is a fixed number of draws at a position known the moment the hit is made, so the
//     kernel only writes the tree down -- a node per hit with the position of its draws, its first child
//     and its next sibling -- and the light samples of all hits (most of a frame's rays), their shading
//     and the sums are left to the coherent kernels below, as for mirror reflections.
//     Frames of a lane's walk (the hits that still have children to come) sit in LDS, one per depth.
// ---------------------------------------------------------------------------
//     MOTION: the work items are the batch's sample slots, (pixel slot, time sample) pairs, each walked like a pixel
//     with the pair's own stretch of the stream, every ray in the copy of the scene moved to the pair's time.
//     TRANSP: a hit on a material that is not opaque also has a pass-through child, traced after its glossy children
//     (src/recursive.cpp:48-60) and linked through NodeB::pad: a node index, kChildNone (none wanted) or kChildMiss
//     (it left the scene without a colour).
constexpr uint32_t kFrameCount = 0x0fffffffu;   // TRANSP, frameDone: glossy children done ...
constexpr uint32_t kFramePassWanted = 1u << 31; // ... the hit wants a pass-through child ...
constexpr uint32_t kFramePassFlying = 1u << 30; // ... and that child is the ray under way (the glossy children are done)
template <bool DOF, bool STATS, bool MOTION = false, bool TRANSP = false>
__global__ __launch_bounds__(kBlockThreads) void glossyTreeKernel(const DevScene sc, const DevCamera cam,
    const DevFeat ft, const DevFrame fr, const WfParams wp, const WfBuffers wb, DevCounters* __restrict__ ctr, int minActive)
{
    extern __shared__ uint32_t stackMem[];
    const Tracer<STATS, MOTION> tracer { sc, ft.accel, (LdsWords)(stackMem + threadIdx.x) };
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t M = MOTION ? (uint32_t)ft.motionSamples : 1u;
    const uint32_t total = wp.nSlots * M; // work items are the batch's pixel slots (MOTION: sample slots)
    uint32_t nodeBlock = wp.nodeBlockMax;
    while (wp.nodeBlockFixed == 0u && nodeBlock > 64u && nodeBlock * (gridDim.x * (blockDim.x / 64u)) > 2u * total * wp.R)
        nodeBlock >>= 1;

    RayState r;
    r.active = false;
    r.cur = kRefNone;
    r.pend = 0u;
    r.top = tracer.stk;
    r.tos = kRefNone;
    r.nInner = r.nTris = 0;
    r.itInner = r.itLeaf = r.itRuns = 0;
    r.topHits[0] = r.topHits[1] = r.topHits[2] = 0;
    r.cycInner = r.cycLeaf = r.cycRun = 0;
    __shared__ uint32_t nodeBlkAll[kBlockWaves][2]; // per wave: next free slot and end of its block of node slots
    volatile __attribute__((address_space(3))) uint32_t* nodeBlk = (volatile __attribute__((address_space(3))) uint32_t*)nodeBlkAll[threadIdx.x >> 6];
    if (lane == 0)
        nodeBlk[0] = nodeBlk[1] = 0u;
    // the open hits of this lane's walk, by depth: the hit's node, how many of its children are done, its last child's node
    __shared__ uint32_t frameNodeAll[kMaxDepth * kBlockThreads], frameDoneAll[kMaxDepth * kBlockThreads], frameLastAll[kMaxDepth * kBlockThreads];
    uint32_t* frameNode = frameNodeAll + threadIdx.x;
    uint32_t* frameDone = frameDoneAll + threadIdx.x;
    uint32_t* frameLast = frameLastAll + threadIdx.x;
    const uint32_t g = ft.glossySamples;

    uint32_t item = 0, nrays = 0, made = 0;
    int px = 0, py = 0;
    uint32_t rayI = 0, sPix = 0, sLens = 0, sMain = 0, draws = 0;
    f3 pof = mk3(0.0f);
    int sp = 0;          // open hits; the ray in flight is a child of frame sp - 1 (a camera ray when sp == 0)
    bool needRay = false; // the pixel's next camera ray is to be made
    bool have = false, dead = false;
    WaveQueue q;
    q.start(total, wp.ownShare, 64u, 64u);

    // node slots: every wave owns a block of consecutive slots and hands them out itself (see pathKernel)
    auto allocNode = [&]() -> uint32_t {
        const unsigned long long am = __ballot(true);
        const uint32_t n = (uint32_t)__popcll(am), rank = (uint32_t)__popcll(am & ((1ull << lane) - 1ull));
        const uint32_t nodeCur = nodeBlk[0], nodeEnd = nodeBlk[1];
        const uint32_t rem = nodeEnd - nodeCur;
        uint32_t fresh = 0;
        if (n > rem) {
            const int leader = __builtin_ctzll(am);
            if (lane == (uint32_t)leader)
                fresh = atomicAdd(&ctr->nodes.v, nodeBlock);
            fresh = __shfl(fresh, leader);
        }
        const uint32_t idx = rank < rem ? nodeCur + rank : fresh + (rank - rem);
        if (rank == 0) {
            nodeBlk[0] = n > rem ? fresh + (n - rem) : nodeCur + n;
            if (n > rem)
                nodeBlk[1] = fresh + nodeBlock;
        }
        return idx;
    };
    // the node just made becomes the next child of the innermost open hit
    auto linkChild = [&](uint32_t idx) {
        const uint32_t top = kBlockThreads * (uint32_t)(sp - 1);
        if (TRANSP && (frameDone[top] & kFramePassFlying)) {
            wb.nodeB[frameNode[top]].pad = idx;
            return;
        }
        const uint32_t last = frameLast[top];
        if (last == (uint32_t)kChildNone)
            wb.nodeB[frameNode[top]].childR = (int32_t)idx;
        else
            wb.nodeB[last].childP = (int32_t)idx;
        frameLast[top] = idx;
    };
    // the next glossy ray of the innermost open hit (src/extra.cpp:333-362), rebuilt from the hit's record
    auto spawnChild = [&]() {
        const uint32_t n = frameNode[kBlockThreads * (uint32_t)(sp - 1)];
        const float4* a = wb.nodeA + 4u * (size_t)n;
        const float4 a0 = a[0], a1 = a[1], a2 = a[2], a3 = a[3];
        const f3 P = { a0.x, a0.y, a0.z }, N = { a0.w, a1.x, a1.y }, rd = { a1.z, a1.w, a2.x };
        f3 o, R;
        reflectionRay(P, rd, N, o, R);
        f3 arbitrary = { 1.0f, 0.0f, 0.0f };
        if (length(arbitrary - R) < 0.01f)
            arbitrary = { 0.0f, 1.0f, 0.0f };
        const f3 u = normalize(cross(arbitrary, R));
        const f3 v = normalize(cross(R, u));
        const float radius = ft.glossyExponent * a3.w / 64.0f;
        const float sx = next1d(sMain);
        const float sy = next1d(sMain);
        draws += 2u;
        const float rr = radius * sx;
        const float theta = kTwoPi * sy;
        const f3 sU = u * rr * cosLikeHost(theta);
        const f3 sV = v * rr * sinLikeHost(theta);
        const f3 dir = normalize(R + sU + sV);
        tracer.launch(r, P + dir * 1.0e-5f, dir, kFltMax, K_RADIANCE);
        nrays++;
    };
    // the pass-through ray of the innermost open hit (generatePassthroughRay, src/recursive.cpp:98-111), likewise
    auto spawnPass = [&]() {
        const uint32_t n = frameNode[kBlockThreads * (uint32_t)(sp - 1)];
        const float4* a = wb.nodeA + 4u * (size_t)n;
        const float4 a0 = a[0], a1 = a[1], a2 = a[2];
        const f3 P = { a0.x, a0.y, a0.z }, rd = { a1.z, a1.w, a2.x };
        tracer.launch(r, P + rd * 1.0e-5f, rd, kFltMax, K_RADIANCE);
        nrays++;
    };

    for (;;) {
        for (int round = 0; round < 2; round++) {
            bool want = !have && !dead;
            q.fetch(want, item, dead, &ctr->next, &ctr->dry[0], total, lane);
            if (!have && !dead && !want) {
                const uint32_t pslot = MOTION ? item / M : item;
                if (slotToPixel(fr, wp.slotBegin + pslot, px, py)) { // (padding pixels of partial edge tiles have no rays)
                    uint32_t seed = (uint32_t)(fr.height * px + py); // src/render.cpp:38
                    if (MOTION) { // the pair's time is draw m + 1 of the pixel's stream; the pair has the stream from there (src/extra.cpp:145-204)
                        seed = lcgSkip(seed, item - pslot * M);
                        r.mv = makeMover(next1d(seed), ft.movement);
                    }
                    sPix = seed;
                    sLens = ft.lensMul * seed + ft.lensAdd;
                    sMain = ft.mainMul * seed + ft.mainAdd;
                    draws = 0;
                    rayI = 0;
                    sp = 0;
                    needRay = true;
                    have = true;
                    r.active = false;
                }
            }
            while (have && !r.active) {
                if (needRay) {
                    if (rayI >= wp.R) { // the pixel's tree is written down
                        have = false;
                        break;
                    }
                    f3 o, d;
                    cameraRay<DOF>(cam, ft, fr, px, py, rayI, sPix, sLens, pof, o, d);
                    tracer.launch(r, o, d, kFltMax, K_RADIANCE); // (counted with the batch's camera rays, below)
                    needRay = false;
                    sp = 0;
                    continue; // a ray that misses the scene's box has finished already
                }
                tracer.settleTie(r);
                if (sc.nSpheres)
                    tracer.spheres(r);
                const uint32_t path = item * wp.R + rayI;
                bool spawned = false;
                if (r.hitPrim == kHitNone) { // src/recursive.cpp:29-32
                    if (sp == 0) {
                        wb.pathHead[path] = kPathMiss;
                        if (ft.envMap) {
                            const f3 env = sampleEnvironmentMap(sc, ft, r.d);
                            wb.pathEnv[3u * path + 0] = env.x;
                            wb.pathEnv[3u * path + 1] = env.y;
                            wb.pathEnv[3u * path + 2] = env.z;
                        }
                    } else if (TRANSP && !ft.envMap && (frameDone[kBlockThreads * (uint32_t)(sp - 1)] & kFramePassFlying)) {
                        // a pass-through child that left the scene: its (black) value is still blended with the hit's
                        wb.nodeB[frameNode[kBlockThreads * (uint32_t)(sp - 1)]].pad = (uint32_t)kChildMiss;
                    } else if (ft.envMap) {
                        // a child that left the scene with a colour of its own: a node without a hit, which carries the
                        // colour (without an environment map such a glossy child adds exactly zero and is left out)
                        const f3 env = sampleEnvironmentMap(sc, ft, r.d);
                        const uint32_t idx = allocNode();
                        if (idx < wb.maxNodes) {
                            NodeB nb {};
                            nb.path = kDeadNode;
                            nb.childR = nb.childP = kChildNone;
                            nb.cvR[0] = env.x, nb.cvR[1] = env.y, nb.cvR[2] = env.z;
                            wb.nodeB[idx] = nb;
                            linkChild(idx);
                        } else {
                            atomicAdd(&ctr->overflow, 1ull);
                        }
                    }
                } else {
                    const HitCtx h = shadeHit<MOTION>(sc, ft, r);
                    const uint32_t idx = allocNode();
                    if (idx < wb.maxNodes) { // always: the pool holds the worst case of the batch
                        made++;
                        float4* a = wb.nodeA + 4u * (size_t)idx;
                        a[0] = make_float4(h.P.x, h.P.y, h.P.z, h.N.x);
                        a[1] = make_float4(h.N.y, h.N.z, r.d.x, r.d.y);
                        a[2] = make_float4(r.d.z, h.kd.x, h.kd.y, h.kd.z);
                        a[3] = make_float4(h.ks.x, h.ks.y, h.ks.z, h.shin);
                        NodeB nb {};
                        nb.path = path;
                        nb.ordinal = draws; // where this hit's light samples draw (counted from the first such draw of the pixel)
                        nb.depth = sp;
                        nb.childR = nb.childP = kChildNone;
                        nb.tr = h.tr;
                        nb.mkd[0] = h.mkd.x, nb.mkd[1] = h.mkd.y, nb.mkd[2] = h.mkd.z;
                        if (TRANSP)
                            nb.pad = (uint32_t)kChildNone;
                        wb.nodeB[idx] = nb;
                        if (sp == 0)
                            wb.pathHead[path] = (int32_t)idx;
                        else
                            linkChild(idx);
                        // computeLightContribution draws before the glossy rays do (src/recursive.cpp:36, :53)
                        draws += wp.drawsPerHit;
                        sMain = lcgSkip(sMain, wp.drawsPerHit);
                        const bool reflective = !(h.ks.x == 0.0f && h.ks.y == 0.0f && h.ks.z == 0.0f);
                        const bool fan = ft.reflections && reflective && (int)g > 0;
                        const bool pass = TRANSP && ft.transparency && h.tr != 1.0f;
                        if (sp < ft.maxDepth && (fan || pass)) {
                            const uint32_t top = kBlockThreads * (uint32_t)sp;
                            frameNode[top] = idx;
                            frameDone[top] = TRANSP ? ((pass ? kFramePassWanted : 0u) | (fan ? 0u : kFramePassFlying)) : 0u;
                            frameLast[top] = (uint32_t)kChildNone;
                            sp++;
                            if (fan)
                                spawnChild();
                            else
                                spawnPass();
                            spawned = true;
                        }
                    } else {
                        if (sp == 0)
                            wb.pathHead[path] = kPathMiss;
                        atomicAdd(&ctr->overflow, 1ull);
                    }
                }
                if (!spawned) {
                    // this ray's subtree is complete: its next sibling, or the next sibling of the nearest open hit that has one
                    for (;;) {
                        if (sp == 0) {
                            rayI++;
                            needRay = true;
                            break;
                        }
                        const uint32_t top = kBlockThreads * (uint32_t)(sp - 1);
                        if (TRANSP) {
                            const uint32_t st = frameDone[top];
                            if (st & kFramePassFlying) { // the pass-through child was the hit's last
                                sp--;
                                continue;
                            }
                            const uint32_t done = (st & kFrameCount) + 1u;
                            if (done < g) {
                                frameDone[top] = (st & ~kFrameCount) | done;
                                spawnChild();
                                break;
                            }
                            if (st & kFramePassWanted) {
                                frameDone[top] = st | kFramePassFlying;
                                spawnPass();
                                break;
                            }
                            sp--;
                            continue;
                        }
                        const uint32_t done = frameDone[top] + 1u;
                        frameDone[top] = done;
                        if (done < g) {
                            spawnChild();
                            break;
                        }
                        sp--;
                    }
                }
            }
            if (__ballot(!have && !dead) == 0ull)
                break;
        }
        if (__ballot(r.active && have) == 0ull) {
            if (__ballot(!dead) == 0ull)
                break;
            continue;
        }
        tracer.run(r, q.exhausted ? (minActive & ~0xff) : minActive);
    }

    // the unused tail of the wave's last block: slots that hold no hit
    for (uint32_t i = nodeBlk[0] + lane, e = nodeBlk[1]; i < e && i < wb.maxNodes; i += 64u)
        wb.nodeB[i].path = kDeadNode;

    const unsigned long long srays = ft.shadows ? waveSum(made) * wp.slots : 0ull;
    const unsigned long long rs = waveSum(nrays), ni = waveSum(r.nInner), nt = waveSum(r.nTris);
    __shared__ unsigned long long raySums[kBlockWaves][2];
    if (lane == 0)
        raySums[threadIdx.x >> 6][0] = rs + srays, raySums[threadIdx.x >> 6][1] = srays;
    __syncthreads();
    if (threadIdx.x == 0) {
        // glossy rays counted here; every camera ray of a real pixel is one intersect call
        unsigned long long all = blockIdx.x == 0 ? (unsigned long long)wp.validPaths : 0ull, shadow = 0ull;
        for (uint32_t w = 0; w < kBlockWaves; w++)
            all += raySums[w][0], shadow += raySums[w][1];
        if (all)
            atomicAdd(&ctr->rays, all);
        if (shadow)
            atomicAdd(&ctr->sRays, shadow);
    }
    if (STATS && lane == 0) {
        atomicAdd(&ctr->inner, ni);
        atomicAdd(&ctr->tris, nt);
    }
}

// ---------------------------------------------------------------------------
// 2. per pixel: how many hits precede each of its paths
// ---------------------------------------------------------------------------
// M: time samples per pixel with motion blur, else 0
__global__ void prefixKernel(const DevFrame fr, const WfParams wp, const WfBuffers wb, uint32_t M)
{
    const uint32_t s = blockIdx.x * blockDim.x + threadIdx.x; // sample slot
    if (s >= wp.nSlots * (M ? M : 1u))
        return;
    int px, py;
    if (!slotToPixel(fr, wp.slotBegin + (M ? s / M : s), px, py) || !seesScene(fr, px, py)) // (no hits, no bases to count)
        return;
    uint32_t base = 0;
    if ((wp.R & 3u) == 0u) { // four paths at a time: one 4-B load of counts, one 16-B store of bases
        const uint32_t* cnt = reinterpret_cast<const uint32_t*>(wb.pathCount) + (size_t)s * (wp.R / 4u);
        uint4* dst = reinterpret_cast<uint4*>(wb.pathBase) + (size_t)s * (wp.R / 4u);
        for (uint32_t i = 0; i < wp.R / 4u; i++) {
            const uint32_t c = cnt[i];
            uint4 b;
            b.x = base;
            b.y = b.x + (c & 0xffu);
            b.z = b.y + ((c >> 8) & 0xffu);
            b.w = b.z + ((c >> 16) & 0xffu);
            base = b.w + (c >> 24);
            dst[i] = b;
        }
        return;
    }
    for (uint32_t i = 0; i < wp.R; i++) {
        const uint32_t path = s * wp.R + i;
        wb.pathBase[path] = base;
        base += wb.pathCount[path];
    }
}

// ---------------------------------------------------------------------------
// 3. one light sample of one hit: computeContributionPointLight, src/light.cpp:166-186,
//    in separate kernels so that the divergent part (the traversal) carries nothing
//    but the ray: (a) sample the light, build the shadow ray and, without transparency,
//    shade the sample as if its light arrived, one lane per (hit, sample), fully
//    coherent; (b) any-hit traversal, persistent wavefronts pulling batches of rays
//    from an atomic queue (every ray ends on a surface, so every ray enters the scene's
//    box: nothing to compact away), which clears the contribution of an occluded
//    sample; (c) with transparency only: shade the samples with the light that got
//    through, one lane per (hit, sample) again.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void shadowGenKernel(const DevScene sc, const DevFeat ft, const DevFrame fr, const WfParams wp,
    const WfBuffers wb, DevCounters* __restrict__ ctr, const rt_light* __restrict__ lightsGlobal)
{
    // `lights` is sc.lights again, seen through the constant address space: nothing this kernel
    // stores changes a light, and only so are they read through the scalar cache
    typedef __attribute__((address_space(4))) const uint32_t* ConstWords;
    const ConstWords lights = (ConstWords)(uintptr_t)lightsGlobal;
    constexpr uint32_t kLightWords = sizeof(rt_light) / 4;
    // With transparency the light that arrives is only known after the traversal (it is tinted on the
    // way): the sample is shaded afterwards, by shadowShadeKernel.  Otherwise it is shaded here, as if
    // it arrived, and the traversal clears the contribution of an occluded sample: one kernel and a
    // third of the record less.
    const bool late = ft.transparency != 0;
    const uint32_t nNodes = ctr->nodes.v < wb.maxNodes ? ctr->nodes.v : wb.maxNodes;
    const uint32_t total = nNodes * wp.slots; // host guarantees maxNodes * slots < 2^32
    for (uint32_t base = blockIdx.x * blockDim.x; base < total; base += gridDim.x * blockDim.x) {
        const uint32_t item = base + threadIdx.x;
        // which light and which of its samples; draws consumed by the lights before it
        // (the loop runs the same for every lane, so the lights are read through the scalar cache)
        int l = -1;
        uint32_t slot = 0, drawPrefix = 0, perSample = 0;
        uint32_t node = 0, path = 0, ordinal = 0;
        bool live = false;
        if (item < total) {
            node = fastDiv(item, wp.divSlots);
            const uint32_t sl = item - node * wp.slots;
            uint32_t first = 0, draws = 0;
            for (int k = 0; k < sc.nLights; k++) {
                const int type = (int)lights[k * kLightWords]; // rt_light::type
                const uint32_t n = (type == 0) ? 1u : ft.shadowSamples;
                const uint32_t per = (type == 0) ? 0u : (type == 1 ? 1u : 2u);
                if (l < 0 && sl < first + n)
                    l = k, slot = sl - first, drawPrefix = draws, perSample = per;
                first += n;
                draws += n * per;
            }
            const uint2 po = *reinterpret_cast<const uint2*>(&wb.nodeB[node]); // path, ordinal
            path = po.x, ordinal = po.y;
            live = l >= 0 && path != kDeadNode;
            if (!live) { // a slot no hit was written to: no ray, no light
                if (late) {
                    float4* sr = wb.srays + 3u * (size_t)item;
                    sr[0] = make_float4(0.0f, 0.0f, 0.0f, -1.0f);
                    sr[2] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
                    wb.svis[item] = 0;
                } else {
                    wb.srays[2u * (size_t)item] = make_float4(0.0f, 0.0f, 0.0f, -1.0f);
                    float* dst = wb.contrib + 3u * (size_t)item;
                    dst[0] = dst[1] = dst[2] = 0.0f;
                }
            }
        }
        // One pass per distinct light among the wave's lanes (one, unless a wave straddles
        // two lights), with the light's index uniform so that its data is read once per wave.
        for (;;) {
            const unsigned long long todo = __ballot(live);
            if (todo == 0ull)
                break;
            const int lu = __builtin_amdgcn_readlane(l, __builtin_ctzll(todo));
            if (live && l == lu) {
                live = false;
                // every lane in here has the same l; taking it from the first one keeps the index
                // (and so the loads) scalar, which the lane's own, equal, l would not
                const uint32_t lw = (uint32_t)__builtin_amdgcn_readfirstlane(l) * kLightWords;
                rt_light light;
                light.type = (int32_t)lights[lw];
#pragma unroll
                for (uint32_t k = 0; k < 21; k++)
                    light.v[k] = __uint_as_float(lights[lw + 1u + k]);
                const float4 a0 = wb.nodeA[4u * (size_t)node];
                const f3 P = { a0.x, a0.y, a0.z };
                int px, py;
                uint32_t pslot = fastDiv(path, wp.divR), before = 0;
                if (ft.motion) { // motion blur: the draws of the pixel's stream up to and with the time of the path's sample slot
                    const uint32_t pair = pslot, M = (uint32_t)ft.motionSamples;
                    pslot = pair / M;
                    before = pair - pslot * M + 1u;
                }
                slotToPixel(fr, wp.slotBegin + pslot, px, py);
                const uint32_t seed = (uint32_t)(fr.height * px + py);
                float when = 0.0f; // the time the shadow ray is traced at
                if (ft.motion) {
                    uint32_t st = lcgSkip(seed, before - 1u);
                    when = next1d(st);
                }
                // where the hit's light-sample draws begin: with glossy reflections the tree kernel knew and
                // wrote it down; otherwise it follows from the number of hits before this one in its pixel
                const uint32_t hitDraws = wp.absDraws ? ordinal : wp.drawsPerHit * (wb.pathBase[path] + ordinal);
                uint32_t s = lcgSkip(seed, before + wp.mainStart + hitDraws + drawPrefix + slot * perSample);
                f3 pos, col;
                if (light.type == 0) {
                    pos = ld3(light.v + 0);
                    col = ld3(light.v + 3);
                } else if (light.type == 1) {
                    sampleSegmentLight(next1d(s), light, pos, col);
                } else {
                    const float sx = next1d(s);
                    const float sy = next1d(s);
                    sampleParallelogramLight({ sx, sy }, light, pos, col);
                }
                f3 d;
                float len = 0.0f;
                if (ft.shadows) { // src/light.cpp:94-97: from the light to the hit point
                    const f3 toHit = P - pos;
                    d = normalize(toHit);
                    len = length(toHit);
                } else { // src/light.cpp:270-272: no ray; keep minus the direction to the light
                    d = -normalize(pos - P);
                }
                if (late) {
                    float4* sr = wb.srays + 3u * (size_t)item;
                    sr[0] = make_float4(pos.x, pos.y, pos.z, len);
                    sr[1] = make_float4(d.x, d.y, d.z, when);
                    sr[2] = make_float4(col.x, col.y, col.z, 0.0f);
                    wb.svis[item] = 1; // visible unless the traversal finds an occluder
                } else {
                    float4* sr = wb.srays + 2u * (size_t)item;
                    sr[0] = make_float4(pos.x, pos.y, pos.z, len);
                    sr[1] = make_float4(d.x, d.y, d.z, when);
                    // the sample's light as if it arrived (src/light.cpp:171-186); the traversal clears it if not
                    f3 c = { 0.0f, 0.0f, 0.0f };
                    if (!isZero(col)) {
                        const float4* a = wb.nodeA + 4u * (size_t)node;
                        const float4 a1 = a[1], a2 = a[2], a3 = a[3];
                        const f3 N = { a0.w, a1.x, a1.y }, rd = { a1.z, a1.w, a2.x }, kd = { a2.y, a2.z, a2.w }, ks = { a3.x, a3.y, a3.z };
                        c = computeShading(ft, N, kd, ks, a3.w, -rd, -d, col);
                    }
                    float* dst = wb.contrib + 3u * (size_t)item;
                    dst[0] = c.x, dst[1] = c.y, dst[2] = c.z;
                }
            }
        }
    }
}

// TRANSP: testVisibilityLightSampleTransparency (src/light.cpp:112-151) instead of the binary
// test: the closest hit along the ray; a hit short of the shaded point on a material that is
// not opaque tints the light and the ray goes on from there.
// Register budget: six waves per SIMD (80 registers) for both variants.  The plain one needs 79 to 81 depending
// on details of the box test, and an 81st register would cost a sixth of the resident waves; the variant that
// follows light through transparent surfaces needs a register or two more and keeps the budget at the price of a
// spill outside the loop.  The counting builds take what they need.  (RT_SHADOW_WAVES, RT_TRANSP_WAVES: experiments.)
#ifndef RT_SHADOW_WAVES
#define RT_SHADOW_WAVES 6
#endif
#ifndef RT_TRANSP_WAVES
#define RT_TRANSP_WAVES 6
#endif
#ifndef RT_MOTION_SHADOW_WAVES
#define RT_MOTION_SHADOW_WAVES 1
#endif
#define RT_SHADOW_ATTR(stats, transp) __attribute__((amdgpu_waves_per_eu((stats) ? 1 : ((transp) ? RT_TRANSP_WAVES : RT_SHADOW_WAVES))))
// MOTION: the ray's record carries its time and the ray meets the copy of the scene moved to that time.
// (No register budget forced: the move needs its own.)
template <bool STATS, bool TRANSP, bool MOTION = false>
__global__ __launch_bounds__(kBlockThreads) __attribute__((amdgpu_waves_per_eu(STATS ? 1 : (MOTION ? RT_MOTION_SHADOW_WAVES : (TRANSP ? RT_TRANSP_WAVES : RT_SHADOW_WAVES))))) void shadowTraceKernel(const DevScene sc, const DevFeat ft, const WfParams wp, const WfBuffers wb,
    DevCounters* __restrict__ ctr, int minActive)
{
    extern __shared__ uint32_t stackMem[];
    const Tracer<STATS, MOTION> tracer { sc, ft.accel, (LdsWords)(stackMem + threadIdx.x) };
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t nNodesAll = ctr->nodes.v < wb.maxNodes ? ctr->nodes.v : wb.maxNodes;
    const uint32_t total = ft.shadows ? nNodesAll * wp.slots : 0u; // shadows off: no ray at all (src/light.cpp:270-272)

    RayState r;
    r.active = false;
    r.cur = kRefNone;
    r.pend = 0u;
    r.top = tracer.stk;
    r.tos = kRefNone;
    r.nInner = r.nTris = 0;
    r.itInner = r.itLeaf = r.itRuns = 0;
    r.topHits[0] = r.topHits[1] = r.topHits[2] = 0;
    r.cycInner = r.cycLeaf = r.cycRun = 0;
    const long long tKernel = STATS ? clock64() : 0;
    const long long tWall = STATS ? wall_clock64() : 0; // 100 MHz, the same on every XCD (the shader clock is not)
    uint32_t item = 0, extra = 0, extraThis = 0; // rays that went on through a surface: of this wave's lane, of its current sample
    bool have = false, dead = false;
    WaveQueue q;
    q.start(total, wp.ownShare, wp.chunk, 64u);
    f3 curKd = mk3(0.0f); // TRANSP: kd of the last surface the light went through, and the product of
    float tprod = 1.0f;   // the transparencies on the way

    for (;;) {
        bool want = !have && !dead;
        q.fetch(want, item, dead, &ctr->nextShadow, &ctr->dry[1], total, lane);
        if (!have && !dead && !want) {
            const float4* sr = wb.srays + (TRANSP ? 3u : 2u) * (size_t)item;
            const float4 a = sr[0], b = sr[1];
            if (a.w >= 0.0f) { // a negative length marks the slot of a node that holds no hit
                if (MOTION)
                    r.mv = makeMover(b.w, ft.movement);
                tracer.launch(r, { a.x, a.y, a.z }, { b.x, b.y, b.z }, a.w, TRANSP ? K_SHADOW_CLOSEST : K_SHADOW_ANY);
                have = true;
                if (TRANSP) {
                    curKd = ld3(wb.nodeB[fastDiv(item, wp.divSlots)].mkd); // src/light.cpp:126
                    tprod = 1.0f;
                    extraThis = 0;
                }
            }
        }
        if (TRANSP && have && !r.active && !r.tie && r.hitPrim >= 0 && ft.accel && sc.refNodes != nullptr && sc.leafBoxes != nullptr
            && !tracer.surelyReached(r))
            r.tie = true; // the reference's own boxes may have kept it from this hit: same way out as a tie
        if (TRANSP && have && !r.active && r.tie) {
                // Two materials met at exactly the same distance: which of them the reference keeps takes
                // its own walk through the caller's tree (Tracer::settleTie).  Not here, where a call would
                // weigh on every ray: the sample is left to the shading kernel, which follows it again that way.
                r.tie = false;
                if (sc.refNodes != nullptr && ft.accel) {
                    wb.svis[item] = 2;
                    extra -= extraThis;
                    have = false;
                }
        }
        if (have && !r.active) {
            if (sc.nSpheres)
                tracer.spheres(r);
            if (!TRANSP) {
                // src/light.cpp:101-107
                const bool visible = (r.hitPrim == kHitNone) ? true : (fabsf(r.tmax - r.slen) <= 0.0001f);
                if (!visible) {
                    // An occluder the reference's own box tests may have kept it from (Tracer::surelyReached) proves
                    // nothing: the sample goes on a list and shadowRedoKernel follows it the reference's way.
                    // (A visible sample needs no such care: what the reference finds is among what was found here.)
                    if (r.hitPrim >= 0 && ft.accel && sc.refNodes != nullptr && sc.leafBoxes != nullptr && !tracer.surelyReached(r)) {
                        wb.redo[atomicAdd(&ctr->redo.v, 1u)] = item;
                    } else {
                        float* dst = wb.contrib + 3u * (size_t)item;
                        dst[0] = dst[1] = dst[2] = 0.0f;
                    }
                }
                have = false;
            } else { // src/light.cpp:138-150
                const float4 a0 = wb.nodeA[4u * (size_t)fastDiv(item, wp.divSlots)];
                const f3 P = { a0.x, a0.y, a0.z };
                const f3 hp = r.o + r.tmax * r.d;
                const bool atPoint = allEpsEq(hp, P, 0.0001f);
                bool again = false;
                if (r.hitPrim != kHitNone && !atPoint) {
                    const DevMaterial hm = r.hitPrim >= 0 ? sc.mats[__float_as_uint(sc.attrs[4u * (uint32_t)r.hitPrim + 3u].w)]
                                                          : sc.mats[sc.spheres[-2 - r.hitPrim].material];
                    if (hm.transparency != 1.0f) {
                        curKd = ld3(hm.kd);
                        tprod *= hm.transparency;
                        again = true;
                    }
                }
                if (again) {
                    const f3 d = r.d;
                    tracer.launch(r, hp + d * 1.0e-5f, d, kFltMax, K_SHADOW_CLOSEST); // src/recursive.cpp:98-111
                    extra++;
                    extraThis++;
                } else {
                    float4* sr = wb.srays + 3u * (size_t)item;
                    if (atPoint) {
                        const float4 c = sr[2];
                        const f3 v = mk3(c.x, c.y, c.z) * curKd * tprod;
                        sr[2] = make_float4(v.x, v.y, v.z, 0.0f);
                    } else {
                        wb.svis[item] = 0;
                    }
                    have = false;
                }
            }
        }
        if (__ballot(have) == 0ull) {
            if (__ballot(!dead) == 0ull)
                break;
            continue;
        }
        tracer.run(r, q.exhausted ? (minActive & ~0xff) : minActive);
    }

    const unsigned long long ni = waveSum(r.nInner), nt = waveSum(r.nTris);
    const unsigned long long ii = waveSum(r.itInner), il = waveSum(r.itLeaf), ir = waveSum(r.itRuns);
    const unsigned long long th[3] = { waveSum(r.topHits[0]), waveSum(r.topHits[1]), waveSum(r.topHits[2]) };
    const unsigned long long nx = TRANSP ? waveSum(extra) : 0ull;
    if (lane == 0) {
        if (TRANSP && nx) { // the rays that went on through a surface; the first ray of every sample is counted where it is made
            atomicAdd(&ctr->rays, nx);
            atomicAdd(&ctr->sRays, nx);
        }
        if (STATS) {
            atomicAdd(&ctr->inner, ni);
            atomicAdd(&ctr->tris, nt);
            atomicAdd(&ctr->sInner, ni);
            atomicAdd(&ctr->sTris, nt);
            atomicAdd(&ctr->loopIters[1][0], ii);
            atomicAdd(&ctr->loopIters[1][1], il);
            atomicAdd(&ctr->loopIters[1][2], ir);
            for (int k = 0; k < 3; k++)
                atomicAdd(&ctr->topHits[1][k], th[k]);
            atomicAdd(&ctr->cycles[1][0], r.cycInner);
            atomicAdd(&ctr->cycles[1][1], r.cycLeaf);
            atomicAdd(&ctr->cycles[1][2], r.cycRun);
            atomicAdd(&ctr->cycles[1][3], (unsigned long long)(clock64() - tKernel));
            atomicMax(&ctr->maxCycles[1], (unsigned long long)(clock64() - tKernel));
            if (ir)
                atomicAdd(&ctr->busyWaves[1], 1ull);
            const uint32_t waveId = blockIdx.x * (blockDim.x / 64u) + (threadIdx.x >> 6);
            if (waveId < 8192u)
                ctr->waveLog[1][waveId][0] = (unsigned long long)tWall, ctr->waveLog[1][waveId][1] = (unsigned long long)wall_clock64();
        }
    }
}

// A light sample under transparency the traversal kernel left undone (two materials met at exactly
// the same distance): testVisibilityLightSampleTransparency (src/light.cpp:112-151) once more, every
// intersect call replayed in the reference's own order of tests.  Leaves the tinted light and the
// visibility flag where the traversal kernel would have, and counts the rays that went on through a surface.
// MOTION: in the copy of the scene moved to the sample's time (its ray's record carries it).
template <bool MOTION>
__device__ __attribute__((noinline)) void redoTransparentSample(const DevScene& sc, const WfParams& wp, const WfBuffers& wb, DevCounters* ctr,
    uint32_t item, LdsWords stk, uint32_t stride, float movement)
{
    float4* sr = wb.srays + 3u * (size_t)item;
    const float4 a = sr[0], b = sr[1];
    const uint32_t node = fastDiv(item, wp.divSlots);
    const float4 a0 = wb.nodeA[4u * (size_t)node];
    const f3 P = { a0.x, a0.y, a0.z };
    f3 o = { a.x, a.y, a.z };
    const f3 d = { b.x, b.y, b.z };
    float tmax = a.w;
    f3 curKd = ld3(wb.nodeB[node].mkd); // src/light.cpp:126
    float tprod = 1.0f;
    unsigned long long extra = 0;
    Mover mv;
    if (MOTION)
        mv = makeMover(b.w, movement);
    for (;;) {
        const uint2 res = replayReference<MOTION>(sc.refNodes, sc.tris, stk, stride, o.x, o.y, o.z, d.x, d.y, d.z, tmax, MOTION ? &mv : nullptr);
        float t = __uint_as_float(res.x);
        int prim = (int)res.y;
        for (int k = 0; k < sc.nSpheres; k++) { // src/bvh.cpp:354-355
            float ts;
            DevSphere sp = sc.spheres[k];
            if (MOTION) { // src/extra.cpp:149-161
                const f3 c = moved(mv, ld3(sp.center));
                sp.center[0] = c.x, sp.center[1] = c.y, sp.center[2] = c.z;
            }
            if (sphereTest(sp, o, d, t, ts)) {
                t = ts;
                prim = -2 - k;
            }
        }
        const f3 hp = o + t * d;
        const bool atPoint = allEpsEq(hp, P, 0.0001f);
        if (prim != kHitNone && !atPoint) { // src/light.cpp:138-150
            const DevMaterial hm = prim >= 0 ? sc.mats[__float_as_uint(sc.attrs[4u * (uint32_t)prim + 3u].w)] : sc.mats[sc.spheres[-2 - prim].material];
            if (hm.transparency != 1.0f) {
                curKd = ld3(hm.kd);
                tprod *= hm.transparency;
                o = hp + d * 1.0e-5f; // src/recursive.cpp:98-111
                tmax = kFltMax;
                extra++;
                continue;
            }
        }
        if (atPoint) {
            const float4 c = sr[2];
            const f3 v = mk3(c.x, c.y, c.z) * curKd * tprod;
            sr[2] = make_float4(v.x, v.y, v.z, 0.0f);
            wb.svis[item] = 1;
        } else {
            wb.svis[item] = 0;
        }
        break;
    }
    if (extra) {
        atomicAdd(&ctr->rays, extra);
        atomicAdd(&ctr->sRays, extra);
    }
}

// Without transparency: the light samples the traversal kernel would not decide (their occluder lies in a leaf the
// reference's own box tests may have turned the ray away from), each traced again in the reference's own order
// of tests over the caller's nodes; src/light.cpp:101-107 on that result.  A handful per frame, if any.
// MOTION: in the copy of the scene moved to the ray's time (its record carries it).
template <bool MOTION = false>
__global__ __launch_bounds__(256) void shadowRedoKernel(const DevScene sc, const WfBuffers wb, DevCounters* __restrict__ ctr, float movement)
{
    __shared__ uint32_t redoStack[kMaxStack * 256];
    const uint32_t n = ctr->redo.v;
    for (uint32_t k = blockIdx.x * blockDim.x + threadIdx.x; k < n; k += gridDim.x * blockDim.x) {
        const uint32_t item = wb.redo[k];
        const float4* sr = wb.srays + 2u * (size_t)item;
        const float4 a = sr[0], b = sr[1];
        const f3 o = { a.x, a.y, a.z }, d = { b.x, b.y, b.z };
        Mover mv;
        if (MOTION)
            mv = makeMover(b.w, movement);
        const uint2 res = replayReference<MOTION>(sc.refNodes, sc.tris, (LdsWords)(redoStack + threadIdx.x), 256u, o.x, o.y, o.z, d.x, d.y, d.z, a.w,
            MOTION ? &mv : nullptr);
        float t = __uint_as_float(res.x);
        bool hit = (int)res.y != kHitNone;
        for (int s = 0; s < sc.nSpheres; s++) { // src/bvh.cpp:354-355
            float ts;
            DevSphere sp = sc.spheres[s];
            if (MOTION) { // src/extra.cpp:149-161
                const f3 c = moved(mv, ld3(sp.center));
                sp.center[0] = c.x, sp.center[1] = c.y, sp.center[2] = c.z;
            }
            if (sphereTest(sp, o, d, t, ts))
                t = ts, hit = true;
        }
        const bool visible = !hit || fabsf(t - a.w) <= 0.0001f;
        if (!visible) {
            float* dst = wb.contrib + 3u * (size_t)item;
            dst[0] = dst[1] = dst[2] = 0.0f;
        }
    }
}

template <bool MOTION = false>
__global__ __launch_bounds__(256) void shadowShadeKernel(const DevScene sc, const DevFeat ft, const WfParams wp, const WfBuffers wb,
    DevCounters* __restrict__ ctr)
{
    __shared__ uint32_t redoStack[kMaxStack * 256]; // only for redoTransparentSample: a column of kMaxStack entries per lane
    const uint32_t nNodes = ctr->nodes.v < wb.maxNodes ? ctr->nodes.v : wb.maxNodes;
    const uint32_t total = nNodes * wp.slots;
    for (uint32_t item = blockIdx.x * blockDim.x + threadIdx.x; item < total; item += gridDim.x * blockDim.x) {
        if (wb.svis[item] == 2)
            redoTransparentSample<MOTION>(sc, wp, wb, ctr, item, (LdsWords)(redoStack + threadIdx.x), 256u, ft.movement);
        f3 c = { 0.0f, 0.0f, 0.0f };
        const float4* sr = wb.srays + 3u * (size_t)item;
        const float4 col = sr[2];
        const f3 vis = wb.svis[item] ? mk3(col.x, col.y, col.z) : mk3(0.0f);
        if (!isZero(vis)) { // src/light.cpp:171-186
            const float4 d = sr[1];
            const uint32_t node = fastDiv(item, wp.divSlots);
            const float4* a = wb.nodeA + 4u * (size_t)node;
            const float4 a0 = a[0], a1 = a[1], a2 = a[2], a3 = a[3];
            f3 N = { a0.w, a1.x, a1.y };
            const f3 rd = { a1.z, a1.w, a2.x }, kd = { a2.y, a2.z, a2.w }, ks = { a3.x, a3.y, a3.z };
            const f3 l = -mk3(d.x, d.y, d.z);
            if (ft.transparency && wb.nodeB[node].tr < 1.0f && dot(N, l) < 0.0f) // src/light.cpp:174-176: lit from behind
                N = -N;
            c = computeShading(ft, N, kd, ks, a3.w, -rd, l, vis);
        }
        float* dst = wb.contrib + 3u * (size_t)item;
        dst[0] = c.x, dst[1] = c.y, dst[2] = c.z;
    }
}

// ---------------------------------------------------------------------------
// 4. per pixel: computeLightContribution's sums and clamp (src/light.cpp:283-298),
//    the mirror combine (src/recursive.cpp:126) and renderRays' mean (:10-17)
// ---------------------------------------------------------------------------
__device__ __forceinline__ f3 nodeLo(const DevScene& sc, const DevFeat& ft, const WfBuffers& wb, uint32_t node, uint32_t slots)
{
    const float* c = wb.contrib + 3u * (size_t)node * slots;
    f3 Lo = mk3(0.0f);
    for (int l = 0; l < sc.nLights; l++) {
        if (sc.lights[l].type == 0) {
            Lo = Lo + (mk3(0.0f) + ld3(c));
            c += 3;
        } else {
            f3 acc = mk3(0.0f);
            for (uint32_t s = 0; s < ft.shadowSamples; s++) {
                acc = acc + ld3(c);
                c += 3;
            }
            Lo = Lo + acc * (1.0f / (float)ft.shadowSamples);
        }
    }
    return clamp01(Lo);
}

// The value renderRay returns for one camera ray: its chain of hits folded from the deepest one up.
__device__ __forceinline__ f3 pathValue(const DevScene& sc, const DevFeat& ft, const WfParams& wp, const WfBuffers& wb, uint32_t path)
{
    const int32_t head = wb.pathHead[path];
    if (head == kPathMiss)
        return ft.envMap ? ld3(wb.pathEnv + 3u * path) : mk3(0.0f);
    int32_t chain[kMaxDepth + 1];
    int n = 0;
    int32_t cur = head, link = kChildNone;
    f3 cv = mk3(0.0f);
    while (n <= kMaxDepth) {
        chain[n++] = cur;
        const NodeB& nb = wb.nodeB[cur];
        link = nb.childR;
        cv = ld3(nb.cvR);
        if (link < 0)
            break;
        cur = link;
    }
    // deepest hit first (without transparency a hit's ks sits in its record's cvP slot)
    f3 v = nodeLo(sc, ft, wb, (uint32_t)chain[n - 1], wp.slots);
    if (link == kChildMiss)
        v = v + cv * ld3(wb.nodeB[chain[n - 1]].cvP);
    for (int k = n - 2; k >= 0; k--)
        v = nodeLo(sc, ft, wb, (uint32_t)chain[k], wp.slots) + v * ld3(wb.nodeB[chain[k]].cvP);
    return v;
}

// The same with transparency: a tree of hits, evaluated like the recursion does (src/recursive.cpp:43-63):
// a hit's own light, plus its mirror child times ks, then blended with its pass-through child.
__device__ __forceinline__ f3 pathValueTree(const DevScene& sc, const DevFeat& ft, const WfParams& wp, const WfBuffers& wb, uint32_t path)
{
    const int32_t head = wb.pathHead[path];
    if (head == kPathMiss)
        return ft.envMap ? ld3(wb.pathEnv + 3u * path) : mk3(0.0f);
    int32_t nodeAt[kMaxDepth + 1];
    int stage[kMaxDepth + 1]; // 0 new, 1 waiting for the mirror child, 2 waiting for the pass-through child
    f3 acc[kMaxDepth + 1];
    int sp = 0;
    nodeAt[0] = head;
    stage[0] = 0;
    f3 ret = mk3(0.0f);
    for (;;) {
        const uint32_t n = (uint32_t)nodeAt[sp];
        const NodeB& nb = wb.nodeB[n];
        const int st = stage[sp];
        bool passNext = false;
        if (st == 0) {
            acc[sp] = nodeLo(sc, ft, wb, n, wp.slots);
            const int32_t cr = nb.childR;
            if (cr >= 0 && sp < kMaxDepth) {
                stage[sp] = 1;
                sp++;
                nodeAt[sp] = cr;
                stage[sp] = 0;
                continue;
            }
            if (cr == kChildMiss) {
                const float4 a3 = wb.nodeA[4u * (size_t)n + 3];
                acc[sp] = acc[sp] + ld3(nb.cvR) * mk3(a3.x, a3.y, a3.z);
            }
            passNext = true;
        } else if (st == 1) { // src/recursive.cpp:126
            const float4 a3 = wb.nodeA[4u * (size_t)n + 3];
            acc[sp] = acc[sp] + ret * mk3(a3.x, a3.y, a3.z);
            passNext = true;
        }
        if (passNext) {
            const int32_t cp = nb.childP;
            if (cp >= 0 && sp < kMaxDepth) {
                stage[sp] = 2;
                sp++;
                nodeAt[sp] = cp;
                stage[sp] = 0;
                continue;
            }
            const float tr = nb.tr;
            ret = (cp == kChildMiss) ? (ld3(nb.cvP) * tr) + (acc[sp] * (1.0f - tr)) : acc[sp];
        } else { // src/recursive.cpp:142
            const float tr = nb.tr;
            ret = (ret * tr) + (acc[sp] * (1.0f - tr));
        }
        if (sp == 0)
            return ret;
        sp--;
    }
}

// With glossy reflections: a hit's own light plus the mean of its glossy children's values times ks
// (src/extra.cpp:364-367), children in the order they were traced (first child, then next siblings).  A
// node without a hit is a child that left the scene and carries the environment's colour.
__device__ __forceinline__ f3 pathValueGlossy(const DevScene& sc, const DevFeat& ft, const WfParams& wp, const WfBuffers& wb, uint32_t path)
{
    const int32_t head = wb.pathHead[path];
    if (head == kPathMiss)
        return ft.envMap ? ld3(wb.pathEnv + 3u * path) : mk3(0.0f);
    int32_t nodeAt[kMaxDepth + 1], child[kMaxDepth + 1]; // child: the child under way, or kChildNone before the first
    f3 Lo[kMaxDepth + 1], acc[kMaxDepth + 1];
    int sp = 0;
    nodeAt[0] = head;
    bool entering = true;
    f3 ret = mk3(0.0f);
    for (;;) {
        const uint32_t n = (uint32_t)nodeAt[sp];
        int32_t next;
        if (entering) {
            const NodeB& nb = wb.nodeB[n];
            bool leaf = true;
            if (nb.path == kDeadNode) {
                ret = ld3(nb.cvR);
            } else {
                Lo[sp] = nodeLo(sc, ft, wb, n, wp.slots);
                const float4 a3 = wb.nodeA[4u * (size_t)n + 3];
                const bool reflective = !(a3.x == 0.0f && a3.y == 0.0f && a3.z == 0.0f);
                if (nb.depth < ft.maxDepth && ft.reflections && reflective) {
                    acc[sp] = mk3(0.0f);
                    leaf = false;
                } else {
                    ret = Lo[sp];
                }
            }
            if (leaf) {
                if (sp == 0)
                    return ret;
                sp--;
                entering = false;
                continue;
            }
            next = nb.childR;
        } else { // a child's value has come back
            acc[sp] = acc[sp] + ret;
            next = wb.nodeB[child[sp]].childP;
        }
        if (next >= 0 && sp < kMaxDepth) {
            child[sp] = next;
            sp++;
            nodeAt[sp] = next;
            entering = true;
            continue;
        }
        const float4 a3 = wb.nodeA[4u * (size_t)n + 3];
        ret = Lo[sp] + (acc[sp] / (float)(int)ft.glossySamples) * mk3(a3.x, a3.y, a3.z);
        if (sp == 0)
            return ret;
        sp--;
        entering = false;
    }
}

// The same with transparency on top: the hit's value so far is blended with its pass-through child's
// (src/recursive.cpp:142), which NodeB::pad names: a node (a hit, or one that carries the colour the ray left the
// scene with), kChildMiss (it left without a colour) or kChildNone (the material is opaque, or the depth cap).
__device__ __forceinline__ f3 pathValueGlossyClear(const DevScene& sc, const DevFeat& ft, const WfParams& wp, const WfBuffers& wb, uint32_t path)
{
    const int32_t head = wb.pathHead[path];
    if (head == kPathMiss)
        return ft.envMap ? ld3(wb.pathEnv + 3u * path) : mk3(0.0f);
    int32_t nodeAt[kMaxDepth + 1], child[kMaxDepth + 1];
    int stage[kMaxDepth + 1]; // 1 glossy children under way, 2 pass-through child under way
    f3 Lo[kMaxDepth + 1], acc[kMaxDepth + 1]; // Lo: the hit's own light; in stage 2 its value before the blend
    int sp = 0;
    nodeAt[0] = head;
    bool entering = true;
    f3 ret = mk3(0.0f);
    for (;;) {
        const uint32_t n = (uint32_t)nodeAt[sp];
        const NodeB& nb = wb.nodeB[n];
        bool fanStage = false, closed = false;
        int32_t next = kChildNone;
        f3 val = mk3(0.0f);
        if (entering) {
            if (nb.path == kDeadNode) {
                ret = ld3(nb.cvR);
                closed = true;
            } else {
                Lo[sp] = nodeLo(sc, ft, wb, n, wp.slots);
                const float4 a3 = wb.nodeA[4u * (size_t)n + 3];
                const bool reflective = !(a3.x == 0.0f && a3.y == 0.0f && a3.z == 0.0f);
                if (nb.depth < ft.maxDepth && ft.reflections && reflective) {
                    acc[sp] = mk3(0.0f);
                    stage[sp] = 1;
                    fanStage = true;
                    next = nb.childR;
                } else {
                    val = Lo[sp];
                }
            }
        } else if (stage[sp] == 1) { // a glossy child's value has come back
            acc[sp] = acc[sp] + ret;
            fanStage = true;
            next = wb.nodeB[child[sp]].childP;
        } else { // the pass-through child's
            const float tr = nb.tr;
            ret = (ret * tr) + (Lo[sp] * (1.0f - tr));
            closed = true;
        }
        if (!closed) {
            if (fanStage) {
                if (next >= 0 && sp < kMaxDepth) {
                    child[sp] = next;
                    sp++;
                    nodeAt[sp] = next;
                    entering = true;
                    continue;
                }
                const float4 a3 = wb.nodeA[4u * (size_t)n + 3];
                val = Lo[sp] + (acc[sp] / (float)(int)ft.glossySamples) * mk3(a3.x, a3.y, a3.z);
            }
            const int32_t pc = (int32_t)nb.pad;
            if (pc >= 0 && sp < kMaxDepth) {
                Lo[sp] = val;
                stage[sp] = 2;
                sp++;
                nodeAt[sp] = pc;
                entering = true;
                continue;
            }
            if (pc == kChildMiss) {
                const float tr = nb.tr;
                val = (mk3(0.0f) * tr) + (val * (1.0f - tr));
            }
            ret = val;
        }
        if (sp == 0)
            return ret;
        sp--;
        entering = false;
    }
}

// One lane per camera ray evaluates its chain (the chains of a pixel sit all over the node
// pool, so their latencies overlap instead of queueing up in one lane); the first lane of
// each pixel then adds the pixel's values in ray order, exactly as renderRays does.
// MODE 0: chains of mirror hits, 1: trees with pass-through children (transparency), 2: trees of glossy children,
// 3: of glossy and pass-through children
template <int MODE>
__global__ __launch_bounds__(256) void combineKernel(const DevScene sc, const DevFeat ft, const DevFrame fr, const WfParams wp,
    const WfBuffers wb, float* __restrict__ out)
{
    __shared__ float vals[256 * 3];
    const uint32_t perRound = wp.R < 256u ? wp.R : 256u; // rays of one pixel handled side by side
    const uint32_t pixelsPerBlock = 256u / perRound;
    const uint32_t pl = fastDiv(threadIdx.x, wp.divPerRound), j = threadIdx.x - pl * perRound;
    const uint32_t s = blockIdx.x * pixelsPerBlock + pl;
    int px = 0, py = 0;
    const bool valid = pl < pixelsPerBlock && s < wp.nSlots && slotToPixel(fr, wp.slotBegin + s, px, py);
    // a pixel that cannot see the scene's box: every ray of it missed; black unless the background has a colour of
    // its own (the camera kernel wrote nothing for it)
    const bool sees = valid && (ft.envMap || seesScene(fr, px, py));
    // motion blur: the pixel's sample slots one after the other, each the mean of its camera rays; then their mean
    // (src/extra.cpp:205-208)
    const uint32_t nSamples = ft.motion ? (uint32_t)ft.motionSamples : 1u;
    f3 L = mk3(0.0f), Lmo = mk3(0.0f);
    for (uint32_t m = 0; m < nSamples; m++) {
        const uint32_t first = (s * nSamples + m) * wp.R; // the sample slot's first path
        L = mk3(0.0f);
        for (uint32_t base = 0; base < wp.R; base += perRound) {
            const uint32_t i = base + j;
            f3 v = mk3(0.0f);
            if (sees && i < wp.R)
                v = MODE == 3 ? pathValueGlossyClear(sc, ft, wp, wb, first + i)
                    : (MODE == 2 ? pathValueGlossy(sc, ft, wp, wb, first + i)
                        : (MODE == 1 ? pathValueTree(sc, ft, wp, wb, first + i) : pathValue(sc, ft, wp, wb, first + i)));
            vals[3u * threadIdx.x + 0] = v.x;
            vals[3u * threadIdx.x + 1] = v.y;
            vals[3u * threadIdx.x + 2] = v.z;
            __syncthreads();
            if (valid && j == 0) {
                const uint32_t n = (wp.R - base < perRound) ? wp.R - base : perRound;
                for (uint32_t k = 0; k < n; k++)
                    L = L + ld3(vals + 3u * (threadIdx.x + k));
            }
            __syncthreads();
        }
        if (ft.motion)
            Lmo = Lmo + L / (float)wp.R;
    }
    if (valid && j == 0) {
        const f3 c = ft.motion ? Lmo / (float)ft.motionSamples : L / (float)wp.R;
        const uint32_t slot = wp.slotBegin + s;
        const size_t idx = fr.packed ? (size_t)slot : ((size_t)(fr.height - 1 - py) * fr.width + px);
        out[3 * idx + 0] = c.x;
        out[3 * idx + 1] = c.y;
        out[3 * idx + 2] = c.z;
    }
}

// ---------------------------------------------------------------------------
// launchers
// ---------------------------------------------------------------------------
typedef void (*ShadowFn)(const DevScene, const DevFeat, const WfParams, const WfBuffers, DevCounters*, int);

static ShadowFn pathVariant(bool stats, bool transp, bool motion = false)
{
    if (motion && transp)
        return stats ? pathKernel<true, true, true> : pathKernel<false, true, true>;
    if (motion)
        return stats ? pathKernel<true, false, true> : pathKernel<false, false, true>;
    if (transp)
        return stats ? pathKernel<true, true> : pathKernel<false, true>;
    return stats ? pathKernel<true, false> : pathKernel<false, false>;
}
#if RT_DEFER_TIES
// the kernel that follows handed-over paths to their end (pathKernel, EXACT)
static ShadowFn pathExactVariant(bool stats, bool motion)
{
    if (motion)
        return stats ? pathKernel<true, false, true, true> : pathKernel<false, false, true, true>;
    return stats ? pathKernel<true, false, false, true> : pathKernel<false, false, false, true>;
}
#endif
static ShadowFn shadowVariant(bool stats, bool transp, bool motion = false)
{
    if (motion && transp)
        return stats ? shadowTraceKernel<true, true, true> : shadowTraceKernel<false, true, true>;
    if (motion)
        return stats ? shadowTraceKernel<true, false, true> : shadowTraceKernel<false, false, true>;
    if (transp)
        return stats ? shadowTraceKernel<true, true> : shadowTraceKernel<false, true>;
    return stats ? shadowTraceKernel<true, false> : shadowTraceKernel<false, false>;
}

hipError_t wavefrontOccupancy(bool stats, bool transp, bool motion, int blockThreads, size_t ldsBytes, int* pathBlocks, int* shadowBlocks)
{
    hipError_t e = hipOccupancyMaxActiveBlocksPerMultiprocessor(pathBlocks, pathVariant(stats, transp, motion), blockThreads, ldsBytes);
    if (e != hipSuccess)
        return e;
    if (getenv("RT_AMD_DEBUG_OCCUPANCY")) { // what bounds the resident blocks: registers or LDS
        for (int k = 0; k < 2; k++) {
            const void* fn = k ? (const void*)shadowVariant(stats, transp, motion) : (const void*)pathVariant(stats, transp, motion);
            hipFuncAttributes a {};
            int noLds = 0, withLds = 0;
            (void)hipFuncGetAttributes(&a, fn);
            (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&noLds, fn, blockThreads, 0);
            (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&withLds, fn, blockThreads, ldsBytes);
            fprintf(stderr, "[rt_amd] %s kernel: %d registers, %zu B static LDS, %zu B dynamic: %d blocks per CU without the dynamic LDS, %d with\n",
                k ? "shadow" : "path", a.numRegs, a.sharedSizeBytes, ldsBytes, noLds, withLds);
        }
    }
    return hipOccupancyMaxActiveBlocksPerMultiprocessor(shadowBlocks, shadowVariant(stats, transp, motion), blockThreads, ldsBytes);
}

hipError_t launchWavefront(bool dof, bool stats, bool transp, bool motion, int pathBlocks, int shadowBlocks, int blockThreads, size_t ldsBytes,
    hipStream_t stream, const DevScene& sc, const DevCamera& cam, const DevFeat& ft, const DevFrame& fr, const WfParams& wp,
    const WfBuffers& wb, DevCounters* ctr, float* out, int minActive, int minActiveShadow, hipEvent_t* marks, int numCU)
{
    // marks (optional, 3 events): after the path kernel, before and after the shadow-ray traversal
    const uint32_t nSampleSlots = wp.nSlots * (motion ? (uint32_t)ft.motionSamples : 1u);
    const int cb = (int)(((size_t)nSampleSlots * wp.R + 256 * kCamPerLane - 1) / (256 * kCamPerLane));
    if (motion)
        hipLaunchKernelGGL((cameraKernel<false, true>), dim3(cb), dim3(256), 0, stream, sc, cam, ft, fr, wp, wb, ctr);
    else if (dof)
        hipLaunchKernelGGL(cameraKernel<true>, dim3(cb), dim3(256), 0, stream, sc, cam, ft, fr, wp, wb, ctr);
    else
        hipLaunchKernelGGL(cameraKernel<false>, dim3(cb), dim3(256), 0, stream, sc, cam, ft, fr, wp, wb, ctr);
    hipLaunchKernelGGL(pathVariant(stats, transp, motion), dim3(pathBlocks), dim3(blockThreads), ldsBytes, stream, sc, ft, wp, wb, ctr, minActive);
#if RT_DEFER_TIES
    if (!transp && ft.accel && sc.refNodes != nullptr) // (no list otherwise)
        hipLaunchKernelGGL(pathExactVariant(stats, motion), dim3(numCU), dim3(blockThreads), ldsBytes, stream, sc, ft, wp, wb, ctr, minActive);
#endif
    if (marks)
        (void)hipEventRecord(marks[0], stream);
    const int pb = (int)((nSampleSlots + 255) / 256);
    hipLaunchKernelGGL(prefixKernel, dim3(pb), dim3(256), 0, stream, fr, wp, wb, motion ? (uint32_t)ft.motionSamples : 0u);
    // Six blocks of this kernel fit a compute unit, so a grid of eight per unit runs a full round and then a third of
    // one: 0.70 ms for the headline frame's samples alone, against 0.57 with 6, 12 or 24 per unit.  With several
    // frames in flight that last third is where the other frames' traversal waves get in, and the order reverses:
    // 5.56-5.59 ms a frame with 8, 5.61 with 6, 5.65-5.66 with 12 (three frames in flight, one box, two rounds of
    // each).  The default serves frames in flight; build with -DRT_GEN_BLOCKS=12 for one frame at a time.
#ifndef RT_GEN_BLOCKS
#define RT_GEN_BLOCKS 8
#endif
    hipLaunchKernelGGL(shadowGenKernel, dim3(numCU * RT_GEN_BLOCKS), dim3(256), 0, stream, sc, ft, fr, wp, wb, ctr, sc.lights);
    if (marks)
        (void)hipEventRecord(marks[1], stream);
    hipLaunchKernelGGL(shadowVariant(stats, transp, motion), dim3(shadowBlocks), dim3(blockThreads), ldsBytes, stream, sc, ft, wp, wb, ctr, minActiveShadow);
    if (marks)
        (void)hipEventRecord(marks[2], stream);
    if (transp && motion) // (without transparency the samples were shaded when they were generated)
        hipLaunchKernelGGL(shadowShadeKernel<true>, dim3(numCU * 8), dim3(256), 0, stream, sc, ft, wp, wb, ctr);
    else if (transp)
        hipLaunchKernelGGL(shadowShadeKernel<false>, dim3(numCU * 8), dim3(256), 0, stream, sc, ft, wp, wb, ctr);
    else if (ft.shadows && ft.accel && sc.refNodes != nullptr && sc.leafBoxes != nullptr) {
        if (motion)
            hipLaunchKernelGGL(shadowRedoKernel<true>, dim3(numCU), dim3(256), 0, stream, sc, wb, ctr, ft.movement);
        else
            hipLaunchKernelGGL(shadowRedoKernel<false>, dim3(numCU), dim3(256), 0, stream, sc, wb, ctr, 0.0f);
    }
    const uint32_t pixelsPerBlock = 256u / (wp.R < 256u ? wp.R : 256u);
    const dim3 combineGrid((wp.nSlots + pixelsPerBlock - 1) / pixelsPerBlock);
    if (transp)
        hipLaunchKernelGGL(combineKernel<1>, combineGrid, dim3(256), 0, stream, sc, ft, fr, wp, wb, out);
    else
        hipLaunchKernelGGL(combineKernel<0>, combineGrid, dim3(256), 0, stream, sc, ft, fr, wp, wb, out);
    return hipGetLastError();
}

typedef void (*GlossyFn)(const DevScene, const DevCamera, const DevFeat, const DevFrame, const WfParams, const WfBuffers, DevCounters*, int);
static GlossyFn glossyVariant(bool dof, bool stats, bool motion = false, bool transp = false)
{
    if (transp) {
        if (motion)
            return stats ? glossyTreeKernel<false, true, true, true> : glossyTreeKernel<false, false, true, true>;
        if (dof)
            return stats ? glossyTreeKernel<true, true, false, true> : glossyTreeKernel<true, false, false, true>;
        return stats ? glossyTreeKernel<false, true, false, true> : glossyTreeKernel<false, false, false, true>;
    }
    if (motion) // (depth of field takes the frame when both are asked for)
        return stats ? glossyTreeKernel<false, true, true> : glossyTreeKernel<false, false, true>;
    if (dof)
        return stats ? glossyTreeKernel<true, true> : glossyTreeKernel<true, false>;
    return stats ? glossyTreeKernel<false, true> : glossyTreeKernel<false, false>;
}

hipError_t glossyOccupancy(bool dof, bool stats, bool motion, bool transp, int blockThreads, size_t ldsBytes, int* treeBlocks)
{
    return hipOccupancyMaxActiveBlocksPerMultiprocessor(treeBlocks, glossyVariant(dof, stats, motion, transp), blockThreads, ldsBytes);
}

// The frame with glossy reflections: the tree kernel writes every pixel's hits down, the light samples of all
// hits are generated, traced and shaded as for mirror reflections, the combine folds the trees.
hipError_t launchWavefrontGlossy(bool dof, bool stats, bool motion, bool transp, int treeBlocks, int shadowBlocks, int blockThreads, size_t ldsBytes, hipStream_t stream,
    const DevScene& sc, const DevCamera& cam, const DevFeat& ft, const DevFrame& fr, const WfParams& wp, const WfBuffers& wb, DevCounters* ctr,
    float* out, int minActive, int minActiveShadow, hipEvent_t* marks, int numCU)
{
    hipLaunchKernelGGL(glossyVariant(dof, stats, motion, transp), dim3(treeBlocks), dim3(blockThreads), ldsBytes, stream, sc, cam, ft, fr, wp, wb, ctr, minActive);
    if (marks)
        (void)hipEventRecord(marks[0], stream);
    hipLaunchKernelGGL(shadowGenKernel, dim3(numCU * RT_GEN_BLOCKS), dim3(256), 0, stream, sc, ft, fr, wp, wb, ctr, sc.lights);
    if (marks)
        (void)hipEventRecord(marks[1], stream);
    hipLaunchKernelGGL(shadowVariant(stats, transp, motion), dim3(shadowBlocks), dim3(blockThreads), ldsBytes, stream, sc, ft, wp, wb, ctr, minActiveShadow);
    if (marks)
        (void)hipEventRecord(marks[2], stream);
    if (transp && motion) // (as in launchWavefront)
        hipLaunchKernelGGL(shadowShadeKernel<true>, dim3(numCU * 8), dim3(256), 0, stream, sc, ft, wp, wb, ctr);
    else if (transp)
        hipLaunchKernelGGL(shadowShadeKernel<false>, dim3(numCU * 8), dim3(256), 0, stream, sc, ft, wp, wb, ctr);
    else if (ft.shadows && ft.accel && sc.refNodes != nullptr && sc.leafBoxes != nullptr) {
        if (motion)
            hipLaunchKernelGGL(shadowRedoKernel<true>, dim3(numCU), dim3(256), 0, stream, sc, wb, ctr, ft.movement);
        else
            hipLaunchKernelGGL(shadowRedoKernel<false>, dim3(numCU), dim3(256), 0, stream, sc, wb, ctr, 0.0f);
    }
    const uint32_t pixelsPerBlock = 256u / (wp.R < 256u ? wp.R : 256u);
    const dim3 combineGrid((wp.nSlots + pixelsPerBlock - 1) / pixelsPerBlock);
    if (transp)
        hipLaunchKernelGGL(combineKernel<3>, combineGrid, dim3(256), 0, stream, sc, ft, fr, wp, wb, out);
    else
        hipLaunchKernelGGL(combineKernel<2>, combineGrid, dim3(256), 0, stream, sc, ft, fr, wp, wb, out);
    return hipGetLastError();
}

} // namespace rt
