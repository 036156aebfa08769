// rt_capi_render.cpp -- the render entry points (rt_render, rt_render_device, rt_render_device_noalloc): what a
// frame needs of working memory, its pixel batches, the launches of the queue-based and the per-pixel renderer, and
// the statistics read back afterwards.

#include "rt_ctx.hpp"

namespace {

// Adds the elapsed times between the events recorded so far to the context's running sums and
// hands the events back for reuse.  The work they bracket must have finished.
static int foldTimings(rt_ctx* ctx)
{
    for (size_t i = 0; i + 1 < ctx->evUsed; i += 2) {
        float ms = 0.0f;
        HIP_TRY(ctx, hipEventElapsedTime(&ms, ctx->evPool[i], ctx->evPool[i + 1]));
        ctx->sumKernelMs += ms;
        ctx->sumLaunches++;
    }
    for (size_t i = 0; i + 3 < ctx->marksUsed; i += 4) {
        float ms = 0.0f;
        HIP_TRY(ctx, hipEventElapsedTime(&ms, ctx->markPool[i], ctx->markPool[i + 1]));
        ctx->sumPathMs += ms;
        HIP_TRY(ctx, hipEventElapsedTime(&ms, ctx->markPool[i + 2], ctx->markPool[i + 3]));
        ctx->sumShadowMs += ms;
        ctx->sumBatches++;
    }
    ctx->evUsed = ctx->marksUsed = 0;
    return RT_OK;
}

// Behind the launches of a frame: marks the slot of lights they read as in use until they have run (see rt_ctx).
// Nothing is recorded on a stream that is being captured into a graph: the graph replays with the slot it was
// recorded with, and whoever replays it keeps the lights as they were.
static int lightsUsedBy(rt_ctx* ctx, hipStream_t stream)
{
    if (!ctx->dLights)
        return RT_OK;
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    if (stream != nullptr && hipStreamIsCapturing(stream, &cap) == hipSuccess && cap != hipStreamCaptureStatusNone)
        return RT_OK;
    hipEvent_t ev = ctx->lightSlotUsed[ctx->lightSlot]; // (made by rt_lights_set: a render call creates nothing)
    if (!ev)
        return RT_OK;
    HIP_TRY(ctx, hipEventRecord(ev, stream));
    ctx->lightSlotPending[ctx->lightSlot] = true;
    return RT_OK;
}

// state after k draws of the LCG (src/sampler.h:18) = mul * seed + add
static void lcgJump(uint64_t k, uint32_t& mul, uint32_t& add)
{
    mul = 1u;
    add = 0u;
    uint32_t stepMul = 747796405u, stepAdd = 2891336453u; // one draw; squared each round
    while (k) {
        if (k & 1) {
            mul = mul * stepMul;
            add = add * stepMul + stepAdd;
        }
        stepAdd = stepAdd * stepMul + stepAdd;
        stepMul = stepMul * stepMul;
        k >>= 1;
    }
}

// renderImage takes the depth-of-field renderer when both are asked for (src/render.cpp:22-25)
static bool motionBlurOn(const rt_features& f) { return f.enableMotionBlur && !f.enableDepthOfField; }

} // namespace

namespace rt {

DevFeat makeFeat(const rt_features& f)
{
    DevFeat d {};
    d.shading = f.enableShading, d.reflections = f.enableReflections, d.shadows = f.enableShadows;
    d.normalInterp = f.enableNormalInterp, d.texMapping = f.enableTextureMapping, d.accel = f.enableAccelStructure;
    d.bilinear = f.enableBilinearTextureFiltering, d.transparency = f.enableTransparency, d.jitter = f.enableJitteredSampling;
    d.shadingModel = f.shadingModel;
    d.spp = f.numPixelSamples, d.shadowSamples = f.numShadowSamples;
    d.dof = f.enableDepthOfField, d.envMap = f.enableEnvironmentMap, d.glossy = f.enableGlossyReflection;
    d.glossySamples = f.numGlossySamples, d.glossyExponent = f.glossyExponent;
    d.dofDistance = f.depthOfFieldDistance, d.dofDiameter = f.depthOfFieldCircleDiameter, d.dofSamples = f.numDepthOfFieldSamples;
    d.divDof = makeFastDiv(std::max(1u, f.numDepthOfFieldSamples));
    d.maxDepth = f.maxRayDepth;
    d.motion = motionBlurOn(f), d.motionSamples = f.motionBlurSamples, d.movement = f.movement;
    // src/render.cpp:58-70, :118: ray count and draws of the pixel-sample stretch
    d.jitterN = static_cast<uint32_t>(std::round(std::sqrt(static_cast<float>(f.numPixelSamples))));
    uint32_t drawsPerCamRay = 0;
    if (f.numPixelSamples > 1) {
        d.camRays = f.enableJitteredSampling ? d.jitterN * d.jitterN : f.numPixelSamples;
        drawsPerCamRay = 2;
    } else {
        d.camRays = 1;
    }
    d.raysPerPixel = f.enableDepthOfField ? d.camRays * f.numDepthOfFieldSamples : d.camRays;
    const uint64_t pixDraws = static_cast<uint64_t>(drawsPerCamRay) * d.camRays;
    const uint64_t lensDraws = f.enableDepthOfField ? 2ull * d.camRays * f.numDepthOfFieldSamples : 0ull;
    lcgJump(pixDraws, d.lensMul, d.lensAdd);
    lcgJump(pixDraws + lensDraws, d.mainMul, d.mainAdd);
    return d;
}

// Trackball accessors, framework/src/trackball.cpp:26-27, :71-74, :112-126,
// with glm::quat(eulerAngles) (glm detail/type_quat.inl:208-217).
DevCamera makeCamera(const rt_camera& c)
{
    DevCamera d {};
    d.halfH = std::tan(c.fovy / 2.0f);
    d.halfW = c.aspect * d.halfH;
    const float cx = std::cos(c.rotation[0] * 0.5f), cy = std::cos(c.rotation[1] * 0.5f), cz = std::cos(c.rotation[2] * 0.5f);
    const float sx = std::sin(c.rotation[0] * 0.5f), sy = std::sin(c.rotation[1] * 0.5f), sz = std::sin(c.rotation[2] * 0.5f);
    d.qw = cx * cy * cz + sx * sy * sz;
    d.q.x = sx * cy * cz - cx * sy * sz;
    d.q.y = cx * sy * cz + sx * cy * sz;
    d.q.z = cx * cy * sz - sx * sy * cz;
    d.pos = mk3(c.lookAt[0], c.lookAt[1], c.lookAt[2]) + quatRotate(d.qw, d.q, mk3(0.0f, 0.0f, -c.distance));
    d.fwd = quatRotate(d.qw, d.q, mk3(0.0f, 0.0f, 1.0f));
    d.up = quatRotate(d.qw, d.q, mk3(0.0f, 1.0f, 0.0f));
    d.left = quatRotate(d.qw, d.q, mk3(1.0f, 0.0f, 0.0f));
    return d;
}

size_t shardTiles(int width, int height, int shard, int nShard)
{
    const size_t tiles = static_cast<size_t>((width + kTile - 1) / kTile) * ((height + kTile - 1) / kTile);
    return (tiles > static_cast<size_t>(shard)) ? (tiles - shard + nShard - 1) / nShard : 0;
}

} // namespace rt

namespace {

// A fixed piece of device memory a render call carves its working buffers from instead of
// allocating them (rt_render_device_noalloc: this device's copy of the static scratch array).
struct Arena {
    char* base;
    size_t bytes, used;
    void* take(size_t n)
    {
        const size_t at = (used + 255) & ~size_t(255);
        if (at + n > bytes)
            return nullptr;
        used = at + n;
        return base + at;
    }
};

} // namespace

namespace rt {

size_t traversalLds(const rt_ctx* ctx)
{
    // A ray's stack holds the bottom marker and at most one waiting sibling per inner node on its path:
    // a tree whose longest root-to-leaf path has L nodes needs L entries.
    return static_cast<size_t>(std::max<uint32_t>(ctx->numLevels, 2)) * kBlockThreads * 4;
}

} // namespace rt

namespace {

// Motion blur on the queue-based renderer: every (pixel, time sample) pair takes a pixel's place in the queues
// (rt_wavefront.hpp).  Without a single time sample (0 / 0 in every pixel, src/extra.cpp:208) the per-pixel kernel
// renders it.
static bool motionOnQueues(const rt_features& F)
{
    return motionBlurOn(F) && F.motionBlurSamples > 0;
}

static int perPixelVariant(const rt_features& F, bool collect)
{
    int variant = 0;
    if (motionBlurOn(F))
        return 16 | 1 | 2 | (collect ? 8 : 0);
    if (F.enableTransparency)
        variant |= 1;
    if (F.enableGlossyReflection && F.enableReflections)
        variant |= 2;
    if (F.enableDepthOfField)
        variant |= 4;
    if (collect)
        variant |= 8;
    return variant;
}

} // namespace

namespace rt {

// Resident blocks per compute unit of the traversing kernels under the current scene and features
// (non-counting variants), asked of the runtime once per change of either: rt_scene_upload and
// rt_features_set call this, so a render call finds the answer ready.
int warmOccupancy(rt_ctx* ctx)
{
    ctx->occ.valid = false;
    if (!ctx->haveScene || !ctx->haveFeatures)
        return RT_OK;
    const rt_features& F = ctx->features;
    rt_ctx::Occupancy o;
    o.lds = traversalLds(ctx);
    o.transp = F.enableTransparency != 0;
    o.variant = perPixelVariant(F, false);
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, wavefrontOccupancy(false, o.transp != 0, motionOnQueues(F), kBlockThreads, o.lds, &o.path, &o.shadow));
    HIP_TRY(ctx, renderOccupancy(o.variant, kBlockThreads, o.lds, &o.render));
    o.valid = true;
    ctx->occ = o;
    return RT_OK;
}

// postprocessImageWithBloom (src/extra.cpp:276-316) on a whole frame in Screen order, in place.
// `copy`: room for the thresholded copy of the frame, or null to use the context's own buffer.
int bloomDevice(rt_ctx* ctx, float* dFrame, int width, int height, hipStream_t stream, float* copy)
{
    const rt_features& F = ctx->features;
    if (!F.enableBloomEffect)
        return RT_OK;
    const size_t floats = static_cast<size_t>(width) * height * 3;
    if (copy) {
        BloomWeights w;
        bloomWeights(static_cast<int>(F.bloomFilterSize), w);
        HIP_TRY(ctx, launchBloom(stream, dFrame, copy, width, height, static_cast<int>(F.bloomFilterSize), F.bloomThreshold, w));
        return RT_OK;
    }
    if (floats > ctx->bloomCap) {
        HIP_TRY(ctx, hipStreamSynchronize(stream));
        if (ctx->dBloom)
            (void)hipFree(ctx->dBloom);
        ctx->dBloom = nullptr;
        ctx->bloomCap = 0;
        HIP_TRY(ctx, hipMalloc(reinterpret_cast<void**>(&ctx->dBloom), floats * sizeof(float)));
        ctx->bloomCap = floats;
    }
    BloomWeights w;
    bloomWeights(static_cast<int>(F.bloomFilterSize), w);
    HIP_TRY(ctx, launchBloom(stream, dFrame, ctx->dBloom, width, height, static_cast<int>(F.bloomFilterSize), F.bloomThreshold, w));
    return RT_OK;
}

} // namespace rt

namespace {

static int traceKnobs(int minActive, int a, int b)
{
    auto c = [](int v) { return std::max(0, std::min(v, 255)); };
    return c(minActive) | (c(a) << 8) | (c(b) << 16);
}

// For every 8x8 tile of this shard, how many of its pixels can have a camera ray that enters the scene's box
// (the root box test of BVHInterface::intersect, src/bvh.cpp:253-259): the pixels inside the bounding
// rectangle of the box's eight corners as the camera sees them (Trackball::generateRay,
// framework/src/trackball.cpp:101-110), a pixel wider on every side.  Only such pixels can make hits, so
// only they need room in the pools of hits and light samples.  Every real pixel counts when the bound
// does not apply: a corner beside or behind the camera, rays that start on a lens instead of the camera's
// position (depth of field), a scene that moves (motion blur), spheres in the scene or the acceleration
// structure off (every ray is traced).
static void candidatePixelsPerTile(const rt_ctx* ctx, const rt_camera& camera, const DevFrame& fr, int width, int height,
    std::vector<uint16_t>& out, std::vector<uint32_t>& validBefore, int rect[4])
{
    const uint32_t nTiles = fr.nWork / 64u;
    out.assign(nTiles, 0);
    validBefore.assign(nTiles + 1, 0u);
    const rt_features& F = ctx->features;
    const DevScene& sc = ctx->scene;
    bool bounded = F.enableAccelStructure && !F.enableDepthOfField && !motionBlurOn(F) && sc.nSpheres == 0 && sc.nPrims != 0;
    int x0 = 0, x1 = width - 1, y0 = 0, y1 = height - 1; // inclusive pixel rectangle
    if (bounded) {
        const DevCamera cam = makeCamera(camera);
        const f3 qInv = { -cam.q.x, -cam.q.y, -cam.q.z };
        double lox = 1e300, hix = -1e300, loy = 1e300, hiy = -1e300;
        for (int k = 0; k < 8 && bounded; k++) {
            const f3 c = { (k & 1) ? sc.rootHi[0] : sc.rootLo[0], (k & 2) ? sc.rootHi[1] : sc.rootLo[1], (k & 4) ? sc.rootHi[2] : sc.rootLo[2] };
            const f3 v = quatRotate(cam.qw, qInv, c - cam.pos); // the corner in the camera's frame: rays leave along +z
            if (!(v.z > 1.0e-3f) || !std::isfinite(v.x) || !std::isfinite(v.y)) {
                bounded = false;
                break;
            }
            // direction (-x halfW, y halfH, 1) of normalised device coordinates (x, y); pixel = (ndc + 1) / 2 * size
            const double nx = -double(v.x) / (double(v.z) * cam.halfW), ny = double(v.y) / (double(v.z) * cam.halfH);
            const double px = (nx + 1.0) * 0.5 * width, py = (ny + 1.0) * 0.5 * height;
            lox = std::min(lox, px), hix = std::max(hix, px), loy = std::min(loy, py), hiy = std::max(hiy, py);
        }
        if (bounded) {
            // a pixel holds the samples [p, p + 1); one more pixel on every side for the roundings of either computation
            x0 = static_cast<int>(std::max(0.0, std::floor(lox) - 1.0)), x1 = static_cast<int>(std::min(double(width - 1), std::floor(hix) + 1.0));
            y0 = static_cast<int>(std::max(0.0, std::floor(loy) - 1.0)), y1 = static_cast<int>(std::min(double(height - 1), std::floor(hiy) + 1.0));
        }
    }
    rect[0] = x0, rect[1] = x1, rect[2] = y0, rect[3] = y1;
    // tile t of the shard is tile t * nShard + shard of the frame: column and row kept by stepping, not by division
    int col = fr.shard % fr.tilesX, row = fr.shard / fr.tilesX;
    const int colStep = fr.nShard % fr.tilesX, rowStep = fr.nShard / fr.tilesX;
    uint32_t valid = 0;
    for (uint32_t t = 0; t < nTiles; t++) {
        const int tx = col * kTile, ty = row * kTile;
        const int ax = std::max(tx, x0), bx = std::min(std::min(tx + kTile, width) - 1, x1);
        const int ay = std::max(ty, y0), by = std::min(std::min(ty + kTile, height) - 1, y1);
        out[t] = (bx >= ax && by >= ay) ? static_cast<uint16_t>((bx - ax + 1) * (by - ay + 1)) : 0;
        validBefore[t] = valid;
        valid += static_cast<uint32_t>(std::min(kTile, width - tx) * std::min(kTile, height - ty));
        col += colStep, row += rowStep;
        if (col >= fr.tilesX)
            col -= fr.tilesX, row++;
    }
    validBefore[nTiles] = valid;
}

// bytes of plain values, appended: the keys of rt_ctx::FramePlan
template <typename T>
static void keyAdd(std::vector<uint8_t>& key, const T& v)
{
    const uint8_t* p = reinterpret_cast<const uint8_t*>(&v);
    key.insert(key.end(), p, p + sizeof(T));
}

constexpr int kMaxBlocksPerCU = 32 / kBlockWaves; // a compute unit holds 32 wavefronts

// The first of the two events around a render call (rt_stats' kernel time).  The pair comes from the context's pool,
// which grows by a pair at a time; once very many launches have gone unread the call waits for them and folds them
// into the sums instead.
static int beginTimedLaunch(rt_ctx* ctx, hipStream_t stream)
{
    if (ctx->evUsed + 2 > ctx->evPool.size()) {
        if (ctx->evPool.size() >= 2 * kMaxTimedLaunches) {
            HIP_TRY(ctx, hipStreamSynchronize(stream));
            const int rf = foldTimings(ctx);
            if (rf != RT_OK)
                return rf;
        } else {
            for (int i = 0; i < 2; i++) {
                hipEvent_t a;
                HIP_TRY(ctx, hipEventCreate(&a));
                ctx->evPool.push_back(a);
            }
        }
    }
    HIP_TRY(ctx, hipEventRecord(ctx->evPool[ctx->evUsed], stream));
    return RT_OK;
}

// The renderer that keeps a pixel in one lane from its first camera ray to its colour: motion blur, glossy
// reflections together with transparency, and RT_OPT_PER_PIXEL.
static int launchPerPixel(rt_ctx* ctx, const DevCamera& cam, const DevFeat& ft, const DevFrame& fr, size_t lds, bool collect, bool occReady,
    DevCounters* dCounters, float* dOut, hipStream_t stream)
{
    const int variant = perPixelVariant(ctx->features, collect);
    int perCU = ctx->blocksPerCU;
    if (perCU <= 0) {
        if (occReady)
            perCU = ctx->occ.render;
        else
            HIP_TRY(ctx, renderOccupancy(variant, kBlockThreads, lds, &perCU));
        perCU = std::max(1, std::min(perCU, kMaxBlocksPerCU));
    }
    const uint32_t wavesNeeded = fr.nWork / 64;
    int blocks = std::min<int64_t>(static_cast<int64_t>(ctx->numCU) * perCU, (wavesNeeded + kBlockWaves - 1) / kBlockWaves);
    blocks = std::max(blocks, 1);
    HIP_TRY(ctx, launchRender(variant, blocks, kBlockThreads, lds, stream, ctx->scene, cam, ft, fr, dCounters, dOut,
                      traceKnobs(ctx->minActive, ctx->parkA, ctx->parkB), ctx->logicRounds));
    return RT_OK;
}

// What the queue-based renderer works out about a frame before any memory is touched: the kernels' parameters and
// launch sizes, and what a pixel costs in working memory.
struct QueueFrame {
    WfParams wp {};
    uint32_t slots = 0;                   // light samples per hit
    bool dof = false, transp = false, glossy = false, motion = false, envMap = false;
    uint32_t raysPerSlot = 0;             // camera rays of a pixel slot: R, with motion blur times its time samples
    int pathBlocks = 0, shadowBlocks = 0; // blocks of the path (or glossy tree) kernel and of the shadow-ray kernels
    uint64_t perNodeBytes = 0;            // a hit with its light samples
    uint64_t nodeSlack = 0;               // node slots the path kernel's wavefronts may leave unused at the ends of their blocks
    uint64_t nodesPerPixel = 0;           // hits a pixel can make, at most
    uint64_t fixedBytes = 0;              // working memory that does not shrink with the batch
    uint64_t nodeCap = 0;                 // nodes a batch may hold: the index of a light sample stays in 32 bits
    uint64_t perSlot = 0, perCand = 0;    // bytes per pixel slot; more per pixel that can see the scene's box
    size_t workspaceCap = 0;              // bytes the batches' buffers may take
};

static int queueFrame(rt_ctx* ctx, const DevFeat& ft, const DevFrame& fr, const Arena* arena, size_t lds, bool glossyWf, bool motionWf,
    bool collect, bool occReady, QueueFrame& q)
{
    const rt_features& F = ctx->features;
    WfParams& wp = q.wp;
    wp.R = ft.raysPerPixel;
    const uint32_t timeSamples = motionWf ? static_cast<uint32_t>(F.motionBlurSamples) : 1u;
    uint32_t slots = 0, drawsPerHit = 0;
    for (const rt_light& l : ctx->hostLights) {
        slots += (l.type == 0) ? 1u : F.numShadowSamples;
        drawsPerHit += (l.type == 0) ? 0u : (l.type == 1 ? F.numShadowSamples : 2u * F.numShadowSamples);
    }
    q.slots = wp.slots = slots;
    wp.divSlots = makeFastDiv(std::max(1u, slots));
    wp.divR = makeFastDiv(wp.R);
    wp.divPerRound = makeFastDiv(std::min(wp.R, 256u));
    wp.drawsPerHit = drawsPerHit;
    wp.pixDraws = (F.numPixelSamples > 1 ? 2u : 0u) * ft.camRays;
    wp.mainStart = wp.pixDraws + (F.enableDepthOfField ? 2u * ft.camRays * F.numDepthOfFieldSamples : 0u);
    wp.chunk = ctx->chunkShadow;
    wp.chunkPath = ctx->chunkPath;
    wp.leastPath = ctx->leastPath;
    wp.nodeBlockFixed = ctx->nodeBlockFixed ? 1u : 0u;
    wp.ownShare = ctx->ownShare;
    wp.absDraws = glossyWf ? 1u : 0u;
    if (wp.R == 0 || wp.R > 65535u || static_cast<uint64_t>(wp.R) * timeSamples > 0xffffffull)
        return fail(ctx, RT_ERR_UNSUPPORTED, "render: %u camera rays per pixel (times %u time samples)", wp.R, timeSamples);
    q.raysPerSlot = wp.R * timeSamples;
    q.motion = motionWf;
    q.dof = F.enableDepthOfField != 0;
    q.transp = F.enableTransparency != 0;
    q.glossy = glossyWf;
    q.envMap = F.enableEnvironmentMap != 0;

    int pathPerCU = 0, shadowPerCU = 0, pathBlocksFixed = 0;
    if (occReady)
        pathPerCU = ctx->occ.path, shadowPerCU = ctx->occ.shadow;
    else
        HIP_TRY(ctx, wavefrontOccupancy(collect, q.transp, motionWf, kBlockThreads, lds, &pathPerCU, &shadowPerCU));
    if (glossyWf) // (the tree kernel stands in the path kernel's place; asking is no allocation)
        HIP_TRY(ctx, glossyOccupancy(q.dof, collect, motionWf, q.transp, kBlockThreads, lds, &pathPerCU));
    if (pathPerCU < 1 || shadowPerCU < 1)
        return fail(ctx, RT_ERR_DEVICE, "render: the traversing kernels do not fit a compute unit (%zu B of LDS)", lds);
    pathPerCU = std::min(pathPerCU, kMaxBlocksPerCU), shadowPerCU = std::min(shadowPerCU, kMaxBlocksPerCU);
    if (ctx->blocksPerCU > 0) { // experiments: fewer blocks than fit
        pathPerCU = std::min(pathPerCU, ctx->blocksPerCU);
        shadowPerCU = std::min(shadowPerCU, ctx->blocksPerCU);
    }
    if (ctx->pathBlocksPerCU > 0) {
        pathPerCU = std::min(pathPerCU, ctx->pathBlocksPerCU);
    } else if (ctx->pathBlocksPerCU == 0) {
        // The path kernel's waves spend much of their time waiting (a hit's shading is a chain of
        // dependent loads, a mirror chain a chain of rays), so on a small frame more waves only
        // means emptier ones that hold their slots until the last chain is done: a block per CU
        // for every 32 K camera rays per CU, at least one (measured: eighth to half of the headline
        // frame 4-5 % faster with frames in flight, 6 % alone; the whole frame keeps what fits).
        const uint64_t perCU = (static_cast<uint64_t>(fr.nWork) * q.raysPerSlot + static_cast<uint64_t>(ctx->numCU) * 32768 - 1)
            / (static_cast<uint64_t>(ctx->numCU) * 32768);
        pathPerCU = static_cast<int>(std::min<uint64_t>(pathPerCU, std::max<uint64_t>(1, perCU)));
    }
    if (ctx->shadowBlocksPerCU > 0)
        shadowPerCU = std::min(shadowPerCU, ctx->shadowBlocksPerCU);
    if (ctx->forcePathPerCU > 0) // experiments: more blocks than the occupancy query answers
        pathPerCU = std::min(ctx->forcePathPerCU, kMaxBlocksPerCU);
    if (ctx->forceShadowPerCU > 0)
        shadowPerCU = std::min(ctx->forceShadowPerCU, kMaxBlocksPerCU);

    // per light sample: its contribution (12 B) and its ray, 32 B; with transparency the ray record also
    // carries the light's colour (48 B) and a visibility byte goes with it
    // (without it, four bytes for the list of samples the replay has to decide)
    const uint64_t perItem = q.transp ? (12ull + 48 + 1) : (12ull + 32 + 4);
    q.perNodeBytes = 64 + sizeof(NodeB) + perItem * std::max(1u, slots);
    // node slots a wavefront of the path kernel reserves at a time; from a fixed arena the smallest
    // the kernel knows, and only as many path blocks as keep the unused ends of those blocks within
    // an eighth of the arena
    wp.nodeBlockMax = arena ? 64u : kNodeBlock;
    q.workspaceCap = ctx->workspaceCap;
    if (arena) {
        q.workspaceCap = arena->bytes - arena->used;
        const uint64_t perBlock = static_cast<uint64_t>(kBlockThreads / 64) * wp.nodeBlockMax * q.perNodeBytes;
        const uint64_t blocksFit = std::max<uint64_t>(1, (q.workspaceCap / 8) / std::max<uint64_t>(1, perBlock));
        if (static_cast<uint64_t>(ctx->numCU) * pathPerCU > blocksFit) {
            pathPerCU = static_cast<int>(blocksFit / ctx->numCU); // 0: fewer blocks than compute units
            if (pathPerCU == 0)
                pathBlocksFixed = static_cast<int>(blocksFit);
        }
    }
    q.pathBlocks = pathBlocksFixed ? pathBlocksFixed : ctx->numCU * pathPerCU;
    q.shadowBlocks = ctx->numCU * shadowPerCU;
    // every wavefront of the path kernel may leave part of its last block of node slots unused
    q.nodeSlack = static_cast<uint64_t>(q.pathBlocks) * (kBlockThreads / 64) * wp.nodeBlockMax;

    // Hits one camera ray can make.  A hit below the depth cap has a mirror child if its material
    // has a ks, and with transparency a pass-through child if the material is not opaque
    // (src/recursive.cpp:43-63): a chain of maxRayDepth + 1 hits, unless some material asks for both.
    const bool mirrors = F.enableReflections && ctx->anyMirror, passes = q.transp && ctx->anyClear;
    uint64_t nodesPerPath = 1;
    if (glossyWf) {
        // every hit below the depth cap fans out into numGlossySamples children (src/extra.cpp:348-362) and, on a material
        // that is not opaque, a pass-through child, each a hit or, with an environment map, a node that carries the
        // colour it left the scene with
        const uint64_t fan = (mirrors ? F.numGlossySamples : 0u) + (passes ? 1u : 0u);
        uint64_t level = 1;
        for (int d = 0; d < F.maxRayDepth && level != 0 && nodesPerPath < (1ull << 40); d++) {
            level *= fan;
            nodesPerPath += level;
        }
    } else if (mirrors && passes && ctx->anyMirrorAndClear)
        nodesPerPath = (2ull << F.maxRayDepth) - 1;
    else if (mirrors || passes)
        nodesPerPath = static_cast<uint64_t>(F.maxRayDepth) + 1;
    q.nodesPerPixel = static_cast<uint64_t>(q.raysPerSlot) * nodesPerPath;
    // what does not shrink with the batch: the unused ends of the node blocks, the buffers' alignment
    q.fixedBytes = q.nodeSlack * q.perNodeBytes + 16 * 256;
    // item index (node * slots + slot) in 32 bits, with room for the queue head to run past the end
    // by one batch per wavefront
    q.nodeCap = (0xffffffffull - (1ull << 26)) / std::max(1u, slots);
    if (q.nodeCap <= q.nodeSlack)
        return fail(ctx, RT_ERR_UNSUPPORTED, "render: %u light samples per hit exceed the supported count", slots);
    // per pixel slot: head, count, prefix (and environment colour) of each of its camera rays; per candidate
    // pixel: its queued camera rays and the worst case of its hits with their light samples
    q.perSlot = static_cast<uint64_t>(q.raysPerSlot) * (4 + 1 + 4 + (q.envMap ? 12 : 0));
    q.perCand = static_cast<uint64_t>(q.raysPerSlot) * 32 + q.nodesPerPixel * q.perNodeBytes;
    return RT_OK;
}

// Which pixels can have a camera ray that enters the scene's box at all: only those make hits, and the pools of hits
// and light samples -- nine tenths of the working memory -- are sized for them alone.  Kept from the last call for
// the same camera, frame, shard and scene.
static void candidateTiles(rt_ctx* ctx, const rt_camera& camera, DevFrame& fr, int width, int height)
{
    rt_ctx::FramePlan& fp = ctx->plan;
    const rt_features& F = ctx->features;
    std::vector<uint8_t> key;
    keyAdd(key, camera), keyAdd(key, width), keyAdd(key, height), keyAdd(key, fr.shard), keyAdd(key, fr.nShard);
    keyAdd(key, ctx->generation), keyAdd(key, F.enableAccelStructure), keyAdd(key, F.enableDepthOfField), keyAdd(key, F.enableMotionBlur);
    if (key != fp.tilesKey) {
        candidatePixelsPerTile(ctx, camera, fr, width, height, fp.tileCand, fp.validBefore, fp.rect);
        fp.tilesKey.swap(key);
        fp.planKey.clear();
    }
    fr.candX0 = fp.rect[0], fr.candX1 = fp.rect[1], fr.candY0 = fp.rect[2], fr.candY1 = fp.rect[3];
}

// Cuts the shard's tiles into pixel batches: consecutive tiles, as many as `cap` bytes (less `fixed`) let in.
// False when one tile alone exceeds that.  The plan of the last call is kept if nothing it depends on has changed.
static bool planBatches(rt_ctx::FramePlan& fp, const QueueFrame& q, uint32_t nTiles, uint64_t cap, uint64_t fixed)
{
    typedef rt_ctx::Batch Batch;
    std::vector<uint8_t> key;
    keyAdd(key, cap), keyAdd(key, fixed), keyAdd(key, q.perSlot), keyAdd(key, q.perCand), keyAdd(key, q.raysPerSlot), keyAdd(key, q.nodesPerPixel);
    keyAdd(key, q.nodeSlack), keyAdd(key, q.nodeCap);
    if (key == fp.planKey && !fp.batches.empty())
        return true; // the same frame under the same limits as last time
    fp.planKey.clear();
    fp.batches.clear();
    fp.maxTiles = fp.maxCand = 0;
    const uint64_t room = cap > fixed ? cap - fixed : 0;
    Batch cur { 0u, 0u, 0ull };
    for (uint32_t t = 0; t < nTiles; t++) {
        const uint64_t tiles = cur.tiles + 1ull, cand = cur.cand + fp.tileCand[t];
        const bool fits = tiles * 64 * q.perSlot + cand * q.perCand <= room && tiles * 64 * q.raysPerSlot <= 0x7fffffffull // path index in 31 bits
            && cand * q.nodesPerPixel + q.nodeSlack <= q.nodeCap;
        if (!fits) {
            if (cur.tiles == 0)
                return false;
            fp.batches.push_back(cur);
            cur = Batch { t, 0u, 0ull };
            t--; // this tile opens the next batch
            continue;
        }
        cur.tiles = static_cast<uint32_t>(tiles), cur.cand = cand;
        fp.maxTiles = std::max<uint64_t>(fp.maxTiles, tiles), fp.maxCand = std::max<uint64_t>(fp.maxCand, cand);
    }
    if (cur.tiles)
        fp.batches.push_back(cur);
    fp.planKey.swap(key);
    return true;
}

constexpr int kQueueBuffers = 10;

// Bytes of the working buffers for the largest batch of the plan, in WfBuffers' order; returns their sum with each
// rounded up to the 256 B it is aligned to.
static size_t bufferSizes(const rt_ctx::FramePlan& fp, const QueueFrame& q, size_t want[kQueueBuffers], uint64_t& maxNodes)
{
    const uint64_t paths = fp.maxTiles * 64 * q.raysPerSlot, queued = std::max<uint64_t>(1, fp.maxCand * q.raysPerSlot);
    maxNodes = fp.maxCand * q.nodesPerPixel + q.nodeSlack;
    const size_t items = static_cast<size_t>(maxNodes) * std::max(1u, q.slots);
    const size_t w[kQueueBuffers] = { static_cast<size_t>(paths) * 4, static_cast<size_t>(paths), static_cast<size_t>(paths) * 4,
        q.envMap ? static_cast<size_t>(paths) * 12 : 16, static_cast<size_t>(maxNodes) * 64,
        static_cast<size_t>(maxNodes) * sizeof(NodeB), items * 12, items * (q.transp ? 48 : 32), q.transp ? items : items * 4,
        static_cast<size_t>(queued) * 32 };
    size_t total = 0;
    for (int i = 0; i < kQueueBuffers; i++)
        want[i] = w[i], total += (w[i] + 255) & ~size_t(255);
    return total;
}

// Plans the frame's pixel batches and hands out their working buffers: from `arena` (nothing is allocated), or
// from the context's own memory, grown when the frame asks for more than it holds.
static int workingBuffers(rt_ctx* ctx, const QueueFrame& q, const DevFrame& fr, Arena* arena, hipStream_t stream, WfBuffers& wb)
{
    rt_ctx::FramePlan& fp = ctx->plan;
    const uint32_t nTiles = fr.nWork / 64u;
    const unsigned long long oneTile = static_cast<unsigned long long>(64 * (q.perSlot + q.perCand) + q.fixedBytes);
    // (RT_OPT_WORKSPACE_BYTES caps what grows with the batch; memory that is really there -- an arena, or
    // what the device has free, below -- has to hold the fixed part as well)
    // (a frame that had to be cut down to the memory the device had left is planned under that limit again, so that
    // the plan kept from it is found and its workspace is not given back and asked for anew frame after frame;
    // the limit is forgotten with the scene)
    const bool clamped = !arena && ctx->wsClamp != 0 && ctx->wsClampGeneration == ctx->generation;
    if (!planBatches(fp, q, nTiles, clamped ? std::min<uint64_t>(q.workspaceCap, ctx->wsClamp) : q.workspaceCap, (arena || clamped) ? q.fixedBytes : 0))
        return fail(ctx, RT_ERR_UNSUPPORTED, "render: one 8x8 tile of this frame needs %llu bytes of working memory, more than the %zu-byte %s",
            oneTile, q.workspaceCap, arena ? "scratch left" : "workspace");
    size_t want[kQueueBuffers];
    uint64_t maxNodes = 0;
    size_t total = bufferSizes(fp, q, want, maxNodes);
    Arena own { nullptr, 0, 0 };
    if (!arena) {
        // The context's own working memory: one allocation, grown when a frame asks for more.  All sizes are
        // known before anything is freed, and a frame never asks for more than the device has left
        // (with what the context already holds): it is rendered in more pixel batches instead.
        if (total > ctx->wsBytes) {
            size_t freeB = 0, totalB = 0;
            HIP_TRY(ctx, hipMemGetInfo(&freeB, &totalB));
            const size_t allowed = ctx->wsBytes + freeB - freeB / 8;
            if (total > allowed) {
                if (!planBatches(fp, q, nTiles, allowed, q.fixedBytes))
                    return fail(ctx, RT_ERR_DEVICE, "render: %zu bytes of device memory are free, one 8x8 tile of this frame needs %llu", freeB, oneTile);
                total = bufferSizes(fp, q, want, maxNodes);
                ctx->wsClamp = allowed;
                ctx->wsClampGeneration = ctx->generation;
            }
            HIP_TRY(ctx, hipStreamSynchronize(stream)); // (frames still in flight on this stream use the old buffer)
            if (ctx->wsBase)
                (void)hipFree(ctx->wsBase);
            ctx->wsBase = nullptr;
            ctx->wsBytes = 0;
            HIP_TRY(ctx, hipMalloc(&ctx->wsBase, total));
            ctx->wsBytes = total;
            if (getenv("RT_AMD_DEBUG_COUNTERS")) {
                uint64_t cand = 0;
                for (uint16_t c : fp.tileCand)
                    cand += c;
                fprintf(stderr, "[rt_amd] working memory %.2f GB for %zu batch(es): %llu of %llu pixels can see the scene's box\n", total / 1e9,
                    fp.batches.size(), static_cast<unsigned long long>(cand), static_cast<unsigned long long>(fr.nWork));
            }
        }
        own = Arena { static_cast<char*>(ctx->wsBase), ctx->wsBytes, 0 };
        arena = &own;
    }
    void* ws[kQueueBuffers];
    for (int i = 0; i < kQueueBuffers; i++) {
        ws[i] = arena->take(want[i]);
        if (!ws[i])
            return fail(ctx, RT_ERR_INTERNAL, "render: working buffers exceed their memory (%zu of %zu bytes used, %zu more asked)",
                arena->used, arena->bytes, want[i]);
    }
    wb.pathHead = static_cast<int32_t*>(ws[0]);
    wb.pathCount = static_cast<uint8_t*>(ws[1]);
    wb.pathBase = static_cast<uint32_t*>(ws[2]);
    wb.pathEnv = static_cast<float*>(ws[3]);
    wb.nodeA = static_cast<float4*>(ws[4]);
    wb.nodeB = static_cast<NodeB*>(ws[5]);
    wb.contrib = static_cast<float*>(ws[6]);
    wb.srays = static_cast<float4*>(ws[7]);
    wb.svis = static_cast<uint8_t*>(ws[8]);
    wb.redo = static_cast<uint32_t*>(ws[8]);
    wb.camRays = static_cast<float4*>(ws[9]);
    wb.maxNodes = static_cast<uint32_t>(maxNodes);
    return RT_OK;
}

// The queue-based renderer, one pixel batch of the plan after another on `stream`.
static int launchBatches(rt_ctx* ctx, QueueFrame& q, const WfBuffers& wb, const DevCamera& cam, const DevFeat& ft, const DevFrame& fr, size_t lds,
    bool collect, bool timeIt, DevCounters* dCounters, float* dOut, hipStream_t stream)
{
    const rt_ctx::FramePlan& fp = ctx->plan;
    WfParams& wp = q.wp;
    const int pathKnobs = traceKnobs(ctx->minActive, ctx->parkA, ctx->parkB);
    const int shadowKnobs = traceKnobs(ctx->minActiveShadow, ctx->parkAShadow, ctx->parkBShadow);
    for (const rt_ctx::Batch& bt : fp.batches) {
        const uint64_t begin = static_cast<uint64_t>(bt.firstTile) * 64;
        wp.slotBegin = static_cast<uint32_t>(begin);
        wp.nSlots = bt.tiles * 64u;
        if (begin != 0) // queue heads and the node allocator start over; the ray counters keep adding up
            HIP_TRY(ctx, hipMemsetAsync(reinterpret_cast<char*>(dCounters) + offsetof(DevCounters, next), 0,
                offsetof(DevCounters, loopIters) - offsetof(DevCounters, next), stream));
        // real pixels of the batch's tiles (edge tiles are padded to 8x8 slots)
        const uint64_t validPixels = fp.validBefore[bt.firstTile + bt.tiles] - fp.validBefore[bt.firstTile];
        wp.validPaths = static_cast<uint32_t>(validPixels * q.raysPerSlot);
        hipEvent_t* marks = nullptr;
        if (timeIt && ctx->marksUsed + 4 > 4 * kMaxTimedLaunches) { // a frame of very many batches, or many unread frames
            HIP_TRY(ctx, hipStreamSynchronize(stream));
            // this frame's own start event sits at evPool[evUsed], behind the finished pairs: those and the
            // batches so far are folded, the start event moves to the front of the emptied pool
            const size_t open = ctx->evUsed;
            const int rf = foldTimings(ctx);
            if (rf != RT_OK)
                return rf;
            std::swap(ctx->evPool[0], ctx->evPool[open]);
        }
        if (timeIt) {
            while (ctx->markPool.size() < ctx->marksUsed + 4) {
                hipEvent_t a;
                HIP_TRY(ctx, hipEventCreate(&a));
                ctx->markPool.push_back(a);
            }
            HIP_TRY(ctx, hipEventRecord(ctx->markPool[ctx->marksUsed], stream));
            marks = &ctx->markPool[ctx->marksUsed + 1];
            ctx->marksUsed += 4;
        }
        if (q.glossy)
            HIP_TRY(ctx, launchWavefrontGlossy(q.dof, collect, q.motion, q.transp, q.pathBlocks, q.shadowBlocks, kBlockThreads, lds, stream, ctx->scene, cam, ft, fr,
                              wp, wb, dCounters, dOut, pathKnobs, shadowKnobs, marks, ctx->numCU));
        else
            HIP_TRY(ctx, launchWavefront(q.dof, collect, q.transp, q.motion, q.pathBlocks, q.shadowBlocks, kBlockThreads, lds, stream, ctx->scene, cam, ft,
                              fr, wp, wb, dCounters, dOut, pathKnobs, shadowKnobs, marks, ctx->numCU));
    }
    return RT_OK;
}

// `arena`: null = working buffers are the context's own (allocated and grown as needed); else they
// are carved from it and the call allocates nothing, times nothing (events are allocations too) and
// splits the frame into as many pixel batches as the arena's size asks for.
static int renderCommon(rt_ctx* ctx, const rt_camera* camera, int width, int height, int shard, int nShard, float* dOut,
    size_t dOutFloats, hipStream_t stream, bool collect, bool timeIt, Arena* arena = nullptr)
{
    if (!ctx->haveScene)
        return fail(ctx, RT_ERR_STATE, "render: no scene uploaded (rt_scene_upload)");
    if (!ctx->haveFeatures)
        return fail(ctx, RT_ERR_STATE, "render: no features set (rt_features_set)");
    if (!camera || width <= 0 || height <= 0 || nShard <= 0 || shard < 0 || shard >= nShard)
        return fail(ctx, RT_ERR_INVALID, "render: bad camera / size / shard (%dx%d, shard %d of %d)", width, height, shard, nShard);
    if (static_cast<uint64_t>(width) * height > 0x3fffffffull)
        return fail(ctx, RT_ERR_UNSUPPORTED, "render: %dx%d exceeds the supported pixel count", width, height);
    const size_t need = (nShard == 1) ? static_cast<size_t>(width) * height * 3 : shardTiles(width, height, shard, nShard) * 64 * 3;
    if (!dOut || dOutFloats < need)
        return fail(ctx, RT_ERR_INVALID, "render: output holds %zu floats, %zu needed", dOutFloats, need);

    const DevFeat ft = makeFeat(ctx->features);
    const DevCamera cam = makeCamera(*camera);
    DevFrame fr {};
    fr.width = width, fr.height = height;
    fr.tilesX = (width + kTile - 1) / kTile, fr.tilesY = (height + kTile - 1) / kTile;
    fr.divTilesX = makeFastDiv(static_cast<uint32_t>(fr.tilesX));
    fr.shard = shard, fr.nShard = nShard;
    fr.nWork = static_cast<uint32_t>(shardTiles(width, height, shard, nShard) * 64);
    fr.packed = (nShard == 1) ? 0 : 1;
    fr.candX0 = 0, fr.candX1 = width - 1, fr.candY0 = 0, fr.candY1 = height - 1;

    const size_t lds = traversalLds(ctx);
    const rt_features& F = ctx->features;

    DevCounters* dCounters = ctx->dCounters;
    float* bloomCopy = nullptr;
    if (arena) {
        dCounters = static_cast<DevCounters*>(arena->take(sizeof(DevCounters)));
        if (F.enableBloomEffect && nShard == 1) {
            bloomCopy = static_cast<float*>(arena->take(static_cast<size_t>(width) * height * 12));
            if (!bloomCopy)
                return fail(ctx, RT_ERR_UNSUPPORTED, "render: the bloom copy of a %dx%d frame does not fit the %zu-byte scratch", width, height, arena->bytes);
        }
        if (!dCounters)
            return fail(ctx, RT_ERR_INTERNAL, "render: scratch smaller than the counters");
    }
    ctx->lastCounters = dCounters;
    HIP_TRY(ctx, hipMemsetAsync(dCounters, 0, collect ? sizeof(DevCounters) : offsetof(DevCounters, waveLog), stream));
    if (timeIt) {
        const int rs = beginTimedLaunch(ctx, stream);
        if (rs != RT_OK)
            return rs;
    }

    // the kernels' resident blocks per compute unit: as found when the scene / features were set, unless
    // this is a counting launch (other variants) or something changed behind that
    const bool occReady = !collect && ctx->occ.valid && ctx->occ.lds == lds && ctx->occ.transp == (F.enableTransparency != 0)
        && ctx->occ.variant == perPixelVariant(F, false);
    if (arena && !occReady)
        return fail(ctx, RT_ERR_STATE, "render: launch geometry not prepared (rt_scene_upload and rt_features_set do that)");
    // Glossy reflections take the queue-based renderer too (a tree kernel walks each pixel's radiance rays, everything
    // else is shared with the mirror case).  So does motion blur (motionOnQueues).
    const bool glossyOn = F.enableGlossyReflection && F.enableReflections;
    const bool glossyWf = glossyOn && !ctx->forcePerPixel && (!motionBlurOn(F) || motionOnQueues(F));
    const bool motionWf = motionOnQueues(F) && !ctx->forcePerPixel && (!glossyOn || glossyWf);
    const bool perPixel = ctx->forcePerPixel || (glossyOn && !glossyWf) || (motionBlurOn(F) && !motionWf);
    int rc = RT_OK;
    if (perPixel) {
        rc = launchPerPixel(ctx, cam, ft, fr, lds, collect, occReady, dCounters, dOut, stream);
    } else { // the queue-based renderer: work items are camera rays and light samples
        QueueFrame q;
        WfBuffers wb {};
        rc = queueFrame(ctx, ft, fr, arena, lds, glossyWf, motionWf, collect, occReady, q);
        if (rc == RT_OK) {
            candidateTiles(ctx, *camera, fr, width, height);
            rc = workingBuffers(ctx, q, fr, arena, stream, wb);
        }
        if (rc == RT_OK)
            rc = launchBatches(ctx, q, wb, cam, ft, fr, lds, collect, timeIt, dCounters, dOut, stream);
    }
    if (rc != RT_OK)
        return rc;
    if (nShard == 1) { // a shard has no neighbours: a sharded frame is post-processed after its gather (rt_postprocess_device)
        const int rb = bloomDevice(ctx, dOut, width, height, stream, bloomCopy);
        if (rb != RT_OK)
            return rb;
    }
    if (timeIt) {
        HIP_TRY(ctx, hipEventRecord(ctx->evPool[ctx->evUsed + 1], stream));
        ctx->evUsed += 2;
    }
    return lightsUsedBy(ctx, stream);
}

} // namespace

namespace rt {

int readStats(rt_ctx* ctx, rt_stats* stats, bool timed)
{
    if (!ctx->hostCounters) // large (per-wave log), so not on the stack; one per context
        ctx->hostCounters.reset(new DevCounters());
    DevCounters& c = *ctx->hostCounters;
    const bool debug = getenv("RT_AMD_DEBUG_COUNTERS") != nullptr;
    HIP_TRY(ctx, hipMemcpy(&c, ctx->lastCounters, debug ? sizeof c : offsetof(DevCounters, waveLog), hipMemcpyDeviceToHost));
    if (debug)
        fprintf(stderr, "[rt_amd] light samples of the last batch left to the replay (their occluder's leaf may be out of the reference's reach): %u\n", c.redo.v);
    if (debug)
        fprintf(stderr, "[rt_amd] paths of the last batch handed over to the kernel that follows them by the reference's own walk: %u\n", c.pathRedo.v);
    if (getenv("RT_AMD_DEBUG_COUNTERS") && (c.loopIters[0][2] | c.loopIters[1][2])) {
        // wave iterations of the traversal loop and the mean number of lanes that had work in them
        const char* names[2] = { "path", "shadow" };
        for (int k = 0; k < 2; k++) {
            const double wi = c.loopIters[k][0] / 64.0, wl = c.loopIters[k][1] / 64.0, wr = c.loopIters[k][2] / 64.0;
            const double li = k ? double(c.sInner) : double(c.inner - c.sInner), lt = k ? double(c.sTris) : double(c.tris - c.sTris);
            fprintf(stderr, "[rt_amd] %s: inner wave-iters %.0f (%.1f lanes busy), leaf wave-iters %.0f (%.1f lanes testing), run() calls %.0f\n",
                names[k], wi, wi > 0 ? li / wi : 0.0, wl, wl > 0 ? lt / wl : 0.0, wr);
            if (c.cycles[k][3])
                fprintf(stderr, "[rt_amd] %s: of a wave's clocks: inner block %.3f (%.0f per iteration), leaf block %.3f (%.0f per iteration), rest of run() %.3f, outside run() %.3f\n",
                    names[k], double(c.cycles[k][0]) / c.cycles[k][3], wi > 0 ? c.cycles[k][0] / wi : 0.0, double(c.cycles[k][1]) / c.cycles[k][3],
                    wl > 0 ? c.cycles[k][1] / wl : 0.0, (double(c.cycles[k][2]) - c.cycles[k][0] - c.cycles[k][1]) / c.cycles[k][3],
                    1.0 - double(c.cycles[k][2]) / c.cycles[k][3]);
            fprintf(stderr, "[rt_amd] %s: %llu waves traced rays; longest wave %.0f clocks, mean of all waves %.0f\n", names[k],
                (unsigned long long)c.busyWaves[k], double(c.maxCycles[k]), wr > 0 ? double(c.cycles[k][3]) / double(std::max<unsigned long long>(1, c.busyWaves[k])) : 0.0);
            { // when the waves of the kernel started and ended, in 10 ns ticks after the first start
                std::vector<unsigned long long> st, en;
                unsigned long long t0 = ~0ull;
                for (int w = 0; w < 8192; w++)
                    if (c.waveLog[k][w][1])
                        t0 = std::min(t0, c.waveLog[k][w][0]);
                for (int w = 0; w < 8192; w++)
                    if (c.waveLog[k][w][1])
                        st.push_back(c.waveLog[k][w][0] - t0), en.push_back(c.waveLog[k][w][1] - t0);
                if (!st.empty()) {
                    std::sort(st.begin(), st.end());
                    std::sort(en.begin(), en.end());
                    auto q = [](const std::vector<unsigned long long>& v, double f) { return (double)v[std::min(v.size() - 1, (size_t)(f * v.size()))]; };
                    fprintf(stderr, "[rt_amd] %s: %zu waves; start ticks p50 %.0f p90 %.0f max %.0f; end ticks p10 %.0f p50 %.0f p90 %.0f p99 %.0f max %.0f (10 ns each)\n",
                        names[k], st.size(), q(st, 0.5), q(st, 0.9), (double)st.back(), q(en, 0.1), q(en, 0.5), q(en, 0.9), q(en, 0.99), (double)en.back());
                }
            }
            fprintf(stderr, "[rt_amd] %s: share of inner steps at pair records below 1024 / 2048 / 4096: %.3f %.3f %.3f (%u levels)\n", names[k],
                li > 0 ? c.topHits[k][0] / li : 0.0, li > 0 ? c.topHits[k][1] / li : 0.0, li > 0 ? c.topHits[k][2] / li : 0.0, ctx->numLevels);
        }
    }
    if (c.overflow) {
        ctx->evUsed = ctx->marksUsed = 0;
        return fail(ctx, RT_ERR_INTERNAL, "render: %llu hits found no slot in the hit pool (the frame is invalid; this is a bug in the pool's sizing)",
            static_cast<unsigned long long>(c.overflow));
    }
    if (stats) {
        stats->rays = c.rays;
        stats->inner_nodes = c.inner;
        stats->tri_tests = c.tris;
        stats->shadow_rays = c.sRays;
        stats->shadow_inner_nodes = c.sInner;
        stats->shadow_tri_tests = c.sTris;
        stats->kernel_ms = stats->path_ms = stats->shadow_ms = 0.0f;
        stats->launches = stats->batches = 0;
        if (timed) {
            const int rf = foldTimings(ctx);
            if (rf != RT_OK)
                return rf;
            stats->kernel_ms = static_cast<float>(ctx->sumKernelMs);
            stats->path_ms = static_cast<float>(ctx->sumPathMs);
            stats->shadow_ms = static_cast<float>(ctx->sumShadowMs);
            stats->launches = static_cast<uint32_t>(ctx->sumLaunches);
            stats->batches = static_cast<uint32_t>(ctx->sumBatches);
            ctx->sumKernelMs = ctx->sumPathMs = ctx->sumShadowMs = 0.0;
            ctx->sumLaunches = ctx->sumBatches = 0;
        }
    }
    ctx->evUsed = ctx->marksUsed = 0;
    return RT_OK;
}

} // namespace rt

extern "C" {

size_t rt_shard_floats(int width, int height, int shard_index, int shard_count)
{
    if (width <= 0 || height <= 0 || shard_count <= 0 || shard_index < 0 || shard_index >= shard_count)
        return 0;
    if (shard_count == 1)
        return static_cast<size_t>(width) * height * 3;
    return shardTiles(width, height, shard_index, shard_count) * 64 * 3;
}

int rt_render(rt_ctx* ctx, const rt_camera* camera, int width, int height, float* out_rgb, rt_stats* stats)
try {
    if (!ctx)
        return fail(nullptr, RT_ERR_INVALID, "rt_render: ctx is NULL");
    if (!out_rgb)
        return fail(ctx, RT_ERR_INVALID, "rt_render: out_rgb is NULL");
    if (width <= 0 || height <= 0)
        return fail(ctx, RT_ERR_INVALID, "rt_render: bad size %dx%d", width, height);
    const auto t0 = std::chrono::steady_clock::now();
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const size_t floats = static_cast<size_t>(width) * height * 3;
    if (floats > ctx->frameCap) {
        if (ctx->dFrame)
            (void)hipFree(ctx->dFrame);
        ctx->dFrame = nullptr;
        ctx->frameCap = 0;
        HIP_TRY(ctx, hipMalloc(reinterpret_cast<void**>(&ctx->dFrame), floats * sizeof(float)));
        ctx->frameCap = floats;
    }
    const int rc = renderCommon(ctx, camera, width, height, 0, 1, ctx->dFrame, floats, nullptr, false, ctx->timing || stats != nullptr);
    if (rc != RT_OK)
        return rc;
    HIP_TRY(ctx, hipMemcpy(out_rgb, ctx->dFrame, floats * sizeof(float), hipMemcpyDeviceToHost));
    const int rs = readStats(ctx, stats, true);
    if (rs != RT_OK)
        return rs;
    if (stats)
        stats->total_ms = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t0).count();
    return RT_OK;
} catch (...) {
    return caught(ctx, "rt_render");
}

int rt_render_device(rt_ctx* ctx, const rt_camera* camera, int width, int height, int shard_index, int shard_count,
    void* d_out, size_t d_out_floats, void* stream, int collect_stats, rt_stats* stats)
try {
    if (!ctx)
        return fail(nullptr, RT_ERR_INVALID, "rt_render_device: ctx is NULL");
    const auto t0 = std::chrono::steady_clock::now();
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const int rc = renderCommon(ctx, camera, width, height, shard_index, shard_count, static_cast<float*>(d_out), d_out_floats,
        static_cast<hipStream_t>(stream), collect_stats != 0, ctx->timing || collect_stats != 0);
    if (rc != RT_OK)
        return rc;
    if (collect_stats) {
        HIP_TRY(ctx, hipStreamSynchronize(static_cast<hipStream_t>(stream)));
        const int rs = readStats(ctx, stats, true);
        if (rs != RT_OK)
            return rs;
        if (stats)
            stats->total_ms = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t0).count();
    }
    return RT_OK;
} catch (...) {
    return caught(ctx, "rt_render_device");
}

int rt_render_device_noalloc(rt_ctx* ctx, const rt_camera* camera, int width, int height, int shard_index, int shard_count,
    void* d_out, size_t d_out_floats, void* stream)
try {
    if (!ctx)
        return fail(nullptr, RT_ERR_INVALID, "rt_render_device_noalloc: ctx is NULL");
    if (!ctx->scratchPtr)
        return fail(ctx, RT_ERR_UNSUPPORTED, "rt_render_device_noalloc: no scratch on device %d (ordinals up to %d)", ctx->device, kMaxDevices - 1);
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    Arena arena { static_cast<char*>(ctx->scratchPtr), kScratchBytes, 0 };
    return renderCommon(ctx, camera, width, height, shard_index, shard_count, static_cast<float*>(d_out), d_out_floats,
        static_cast<hipStream_t>(stream), false, false, &arena);
} catch (...) {
    return caught(ctx, "rt_render_device_noalloc");
}

} // extern "C"
