// rt_capi.cpp -- the C ABI of include/rt_amd.h, first part: context and options, lights, features, and the entry
// points beside the renderer (post-processing, BMP, single rays, formulas).  The builders, the scene upload and the
// render entry points are in rt_capi_build.cpp, rt_capi_upload.cpp and rt_capi_render.cpp (see rt_ctx.hpp).
// All of them are built with hipcc (-ffp-contract=off) so the constants computed on the host carry the same
// roundings as the scalar code they stand in for.

#include "rt_ctx.hpp"

static thread_local std::string g_lastError;

// Scratch of the entry point that allocates nothing (rt_render_device_noalloc): counters, queues,
// the hit and light-sample pools of one pixel batch and the bloom copy live here instead of in
// buffers from hipMalloc.  Every GPU the library runs on holds its own copy of the array, at its
// own address, until the process ends; host code never uses the array's name as an address but the
// per-device pointer hipGetSymbolAddress returned when the first context on that device was made.
static __device__ __align__(256) unsigned char scratch[kScratchBytes];
static void* g_scratchPtr[kMaxDevices] = {};
static std::mutex g_scratchMutex;

// The device address of `scratch` on `device`, which must be the current device.
static hipError_t scratchOnDevice(int device, void** out)
{
    std::lock_guard<std::mutex> lock(g_scratchMutex);
    if (device < 0 || device >= kMaxDevices)
        return hipErrorInvalidDevice;
    if (!g_scratchPtr[device]) {
        void* p = nullptr;
        const hipError_t e = hipGetSymbolAddress(&p, HIP_SYMBOL(scratch));
        if (e != hipSuccess)
            return e;
        g_scratchPtr[device] = p;
    }
    *out = g_scratchPtr[device];
    return hipSuccess;
}

namespace rt {

int fail(rt_ctx* ctx, int code, const char* fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    if (ctx)
        ctx->error = buf;
    g_lastError = buf;
    return code;
}

// No C++ exception leaves the library (its callers bind a C ABI: ctypes, cgo, a C program): what a
// std::vector or std::string throws inside an entry point becomes a status like any other failure.
int caught(rt_ctx* ctx, const char* where) noexcept
{
    int code = RT_ERR_INTERNAL;
    try {
        try {
            throw;
        } catch (const std::bad_alloc&) {
            code = RT_ERR_NOMEM;
            return fail(ctx, code, "%s: out of host memory", where);
        } catch (const std::exception& e) {
            return fail(ctx, code, "%s: %s", where, e.what());
        } catch (...) {
            return fail(ctx, code, "%s: unknown failure", where);
        }
    } catch (...) { // not even the message could be stored
        return code;
    }
}

} // namespace rt

// Scheduling parameters of the traversing kernels, overridable from the environment for the sweeps under
// tools/ (only when RT_AMD_EXPERIMENTS is set).  What changes results or is of use to a caller is an
// rt_option instead.
static void readExperimentKnobs(rt_ctx* ctx)
{
    struct Knob {
        const char* name;
        int* a;
        int* b; // second field set to the same value, or null
        int lo, hi;
    };
    const Knob knobs[] = {
        { "RT_AMD_MIN_ACTIVE", &ctx->minActive, &ctx->minActiveShadow, 0, 64 },
        { "RT_AMD_MIN_ACTIVE_SHADOW", &ctx->minActiveShadow, nullptr, 0, 64 },
        { "RT_AMD_PARK_A", &ctx->parkA, &ctx->parkAShadow, 0, 255 },
        { "RT_AMD_PARK_B", &ctx->parkB, &ctx->parkBShadow, 0, 255 },
        { "RT_AMD_PARK_A_SHADOW", &ctx->parkAShadow, nullptr, 0, 255 },
        { "RT_AMD_PARK_B_SHADOW", &ctx->parkBShadow, nullptr, 0, 255 },
        { "RT_AMD_CHUNK_PATH", &ctx->chunkPath, nullptr, 64, 1 << 20 },
        { "RT_AMD_CHUNK_SHADOW", &ctx->chunkShadow, nullptr, 64, 1 << 20 },
        { "RT_AMD_NODE_BLOCK_FIXED", &ctx->nodeBlockFixed, nullptr, 0, 1 },
        { "RT_AMD_LEAST_PATH", &ctx->leastPath, nullptr, 1, 64 },
        { "RT_AMD_OWN_SHARE", &ctx->ownShare, nullptr, 0, 16 },
        { "RT_AMD_LOGIC_ROUNDS", &ctx->logicRounds, nullptr, 1, 64 },
        { "RT_AMD_BLOCKS_PER_CU", &ctx->blocksPerCU, nullptr, 0, 32 },
        { "RT_AMD_PATH_BLOCKS_PER_CU", &ctx->pathBlocksPerCU, nullptr, -1, 32 },
        { "RT_AMD_SHADOW_BLOCKS_PER_CU", &ctx->shadowBlocksPerCU, nullptr, 0, 32 },
        { "RT_AMD_FORCE_PATH_PER_CU", &ctx->forcePathPerCU, nullptr, 0, 32 },
        { "RT_AMD_FORCE_SHADOW_PER_CU", &ctx->forceShadowPerCU, nullptr, 0, 32 },
    };
    for (const Knob& k : knobs)
        if (const char* s = getenv(k.name)) {
            const int v = std::min(k.hi, std::max(k.lo, atoi(s)));
            *k.a = v;
            if (k.b)
                *k.b = v;
        }
}

extern "C" {

const char* rt_last_error(const rt_ctx* ctx) { return ctx ? ctx->error.c_str() : g_lastError.c_str(); }

int rt_ctx_create(int device_ordinal, rt_ctx** out)
try {
    if (!out)
        return fail(nullptr, RT_ERR_INVALID, "rt_ctx_create: out is NULL");
    *out = nullptr;
    int count = 0;
    HIP_TRY(nullptr, hipGetDeviceCount(&count));
    if (device_ordinal < 0 || device_ordinal >= count)
        return fail(nullptr, RT_ERR_INVALID, "rt_ctx_create: device %d not in [0, %d)", device_ordinal, count);
    HIP_TRY(nullptr, hipSetDevice(device_ordinal));
    hipDeviceProp_t prop;
    HIP_TRY(nullptr, hipGetDeviceProperties(&prop, device_ordinal));
    rt_ctx* ctx = new rt_ctx;
    ctx->device = device_ordinal;
    ctx->numCU = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    if (hipMalloc(reinterpret_cast<void**>(&ctx->dCounters), sizeof(DevCounters)) != hipSuccess) {
        delete ctx;
        return fail(nullptr, RT_ERR_DEVICE, "rt_ctx_create: device allocation failed");
    }
    ctx->lastCounters = ctx->dCounters;
    if (device_ordinal < kMaxDevices) { // this device's address of the static scratch array, once per device
        const hipError_t e = scratchOnDevice(device_ordinal, &ctx->scratchPtr);
        if (e != hipSuccess) {
            (void)hipFree(ctx->dCounters);
            delete ctx;
            return fail(nullptr, RT_ERR_DEVICE, "rt_ctx_create: hipGetSymbolAddress(scratch) failed: %s", hipGetErrorString(e));
        }
    }
    if (getenv("RT_AMD_EXPERIMENTS")) // scheduling experiments (tools/*.sh); nothing a caller needs
        readExperimentKnobs(ctx);
    *out = ctx;
    return RT_OK;
} catch (...) {
    return caught(nullptr, "rt_ctx_create");
}

int rt_ctx_set_option(rt_ctx* ctx, int option, int64_t value)
try {
    if (!ctx)
        return fail(nullptr, RT_ERR_INVALID, "rt_ctx_set_option: ctx is NULL");
    switch (option) {
    case RT_OPT_TREE:
        if (value < 0 || value > 2)
            return fail(ctx, RT_ERR_INVALID, "rt_ctx_set_option: RT_OPT_TREE takes 0, 1 or 2");
        ctx->treeMode = static_cast<int>(value);
        return RT_OK;
    case RT_OPT_PER_PIXEL:
        ctx->forcePerPixel = value != 0;
        return RT_OK;
    case RT_OPT_STRICT_INTERSECT:
        ctx->strictIntersect = value != 0;
        ctx->scene.strictIntersect = ctx->strictIntersect ? 1 : 0;
        return RT_OK;
    case RT_OPT_WORKSPACE_BYTES:
        if (value < (1 << 20))
            return fail(ctx, RT_ERR_INVALID, "rt_ctx_set_option: RT_OPT_WORKSPACE_BYTES takes at least 1 MiB");
        ctx->workspaceCap = static_cast<size_t>(value);
        return RT_OK;
    case RT_OPT_TIMING:
        ctx->timing = value != 0;
        return RT_OK;
    case RT_OPT_TREE_BUILD:
        ctx->treeBuildOnDevice = value != 0;
        return RT_OK;
    case RT_OPT_REFERENCE_BOXES:
        ctx->referenceBoxes = value != 0;
        ctx->scene.leafBoxes = ctx->referenceBoxes ? ctx->dLeafBoxes : nullptr;
        return RT_OK;
    default:
        return fail(ctx, RT_ERR_INVALID, "rt_ctx_set_option: unknown option %d", option);
    }
} catch (...) {
    return caught(ctx, "rt_ctx_set_option");
}

void rt_ctx_destroy(rt_ctx* ctx)
{
    if (!ctx)
        return;
    (void)hipSetDevice(ctx->device);
    (void)hipDeviceSynchronize();
    void* ptrs[] = { ctx->dPairs, ctx->dRefNodes, ctx->dLeafBoxes, ctx->dTris, ctx->dAttrs, ctx->dMats, ctx->dTexs, ctx->dTexels, ctx->dSpheres, ctx->dLights,
        ctx->dCounters, ctx->dFrame, ctx->dBloom };
    for (void* p : ptrs)
        if (p)
            (void)hipFree(p);
    if (ctx->wsBase)
        (void)hipFree(ctx->wsBase);
    for (hipEvent_t e : ctx->lightSlotUsed)
        if (e)
            (void)hipEventDestroy(e);
    if (ctx->copyStream)
        (void)hipStreamDestroy(ctx->copyStream);
    for (hipEvent_t e : ctx->markPool)
        (void)hipEventDestroy(e);
    for (hipEvent_t e : ctx->evPool)
        (void)hipEventDestroy(e);
    delete ctx;
}

int rt_lights_set(rt_ctx* ctx, const rt_light* lights, uint32_t n)
try {
    if (!ctx)
        return fail(nullptr, RT_ERR_INVALID, "rt_lights_set: ctx is NULL");
    if (n && !lights)
        return fail(ctx, RT_ERR_INVALID, "rt_lights_set: NULL array with non-zero count");
    for (uint32_t i = 0; i < n; i++)
        if (lights[i].type < 0 || lights[i].type > 2)
            return fail(ctx, RT_ERR_INVALID, "rt_lights_set: light %u has type %d", i, lights[i].type);
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    if (!ctx->copyStream) {
        HIP_TRY(ctx, hipStreamCreateWithFlags(&ctx->copyStream, hipStreamNonBlocking));
        for (hipEvent_t& e : ctx->lightSlotUsed)
            HIP_TRY(ctx, hipEventCreateWithFlags(&e, hipEventDisableTiming));
    }
    if (n > ctx->lightCap || !ctx->dLights) {
        HIP_TRY(ctx, hipDeviceSynchronize());
        if (ctx->dLights)
            (void)hipFree(ctx->dLights);
        ctx->dLights = nullptr;
        ctx->lightCap = std::max<uint32_t>(n, 16);
        HIP_TRY(ctx, hipMalloc(reinterpret_cast<void**>(&ctx->dLights), static_cast<size_t>(rt_ctx::kLightSlots) * ctx->lightCap * sizeof(rt_light)));
        for (bool& pending : ctx->lightSlotPending)
            pending = false;
    }
    // The next slot, written now (frames already enqueued keep the slot they were enqueued with; see rt_ctx).
    const int slot = (ctx->lightSlot + 1) % rt_ctx::kLightSlots;
    if (ctx->lightSlotPending[slot]) {
        HIP_TRY(ctx, hipEventSynchronize(ctx->lightSlotUsed[slot]));
        ctx->lightSlotPending[slot] = false;
    }
    rt_light* dst = ctx->dLights + static_cast<size_t>(slot) * ctx->lightCap;
    if (n) {
        HIP_TRY(ctx, hipMemcpyAsync(dst, lights, n * sizeof(rt_light), hipMemcpyHostToDevice, ctx->copyStream));
        HIP_TRY(ctx, hipStreamSynchronize(ctx->copyStream));
    }
    ctx->lightSlot = slot;
    ctx->hostLights.assign(lights, lights + n);
    ctx->scene.lights = dst;
    ctx->scene.nLights = static_cast<int32_t>(n);
    return RT_OK;
} catch (...) {
    return caught(ctx, "rt_lights_set");
}

int rt_features_set(rt_ctx* ctx, const rt_features* f)
try {
    if (!ctx || !f)
        return fail(ctx, RT_ERR_INVALID, "rt_features_set: NULL argument");
    if (f->enableMotionBlur && !f->enableDepthOfField && f->motionBlurSamples > 65535)
        return fail(ctx, RT_ERR_UNSUPPORTED, "rt_features_set: %d motion blur samples", f->motionBlurSamples);
    if (f->enableBloomEffect && (f->bloomFilterSize < (uint32_t)kBloomMinSize || f->bloomFilterSize > (uint32_t)kBloomMaxSize))
        return fail(ctx, RT_ERR_UNSUPPORTED,
            "rt_features_set: bloomFilterSize %u: below %d the reference's filter loop writes outside the image, above %d its factorial overflows",
            f->bloomFilterSize, kBloomMinSize, kBloomMaxSize);
    if (f->shadingModel < 0 || f->shadingModel > 3)
        return fail(ctx, RT_ERR_INVALID, "rt_features_set: shadingModel %d", f->shadingModel);
    if (f->maxRayDepth < 0 || f->maxRayDepth > kMaxDepth)
        return fail(ctx, RT_ERR_INVALID, "rt_features_set: maxRayDepth %d not in [0, %d]", f->maxRayDepth, kMaxDepth);
    ctx->features = *f;
    ctx->haveFeatures = true;
    return warmOccupancy(ctx);
} catch (...) {
    return caught(ctx, "rt_features_set");
}

} // extern "C"

extern "C" {

int rt_postprocess_device(rt_ctx* ctx, void* d_frame, int width, int height, void* stream)
try {
    if (!ctx)
        return fail(nullptr, RT_ERR_INVALID, "rt_postprocess_device: ctx is NULL");
    if (!ctx->haveFeatures)
        return fail(ctx, RT_ERR_STATE, "rt_postprocess_device: no features set (rt_features_set)");
    if (!d_frame || width <= 0 || height <= 0)
        return fail(ctx, RT_ERR_INVALID, "rt_postprocess_device: bad argument");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    return bloomDevice(ctx, static_cast<float*>(d_frame), width, height, static_cast<hipStream_t>(stream));
} catch (...) {
    return caught(ctx, "rt_postprocess_device");
}

int rt_pack_bmp_device(rt_ctx* ctx, const void* d_frame, int width, int height, void* d_bgra, void* stream)
try {
    if (!ctx)
        return fail(nullptr, RT_ERR_INVALID, "rt_pack_bmp_device: ctx is NULL");
    if (!d_frame || !d_bgra || width <= 0 || height <= 0)
        return fail(ctx, RT_ERR_INVALID, "rt_pack_bmp_device: bad argument");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, launchPackBmp(static_cast<hipStream_t>(stream), static_cast<const float*>(d_frame), static_cast<uint8_t*>(d_bgra), width, height));
    return RT_OK;
} catch (...) {
    return caught(ctx, "rt_pack_bmp_device");
}

int rt_write_bmp(rt_ctx* ctx, const float* rgb, int width, int height, const char* path)
try {
    if (!ctx)
        return fail(nullptr, RT_ERR_INVALID, "rt_write_bmp: ctx is NULL");
    if (!rgb || !path || width <= 0 || height <= 0)
        return fail(ctx, RT_ERR_INVALID, "rt_write_bmp: bad argument");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const size_t pixels = static_cast<size_t>(width) * height;
    float* dRgb = nullptr;
    uint8_t* dOut = nullptr;
    std::vector<uint8_t> file(122 + pixels * 4);
    hipError_t e = hipMalloc(reinterpret_cast<void**>(&dRgb), pixels * 12);
    if (e == hipSuccess)
        e = hipMalloc(reinterpret_cast<void**>(&dOut), pixels * 4);
    if (e == hipSuccess)
        e = hipMemcpy(dRgb, rgb, pixels * 12, hipMemcpyHostToDevice);
    if (e == hipSuccess)
        e = launchPackBmp(nullptr, dRgb, dOut, width, height);
    if (e == hipSuccess)
        e = hipMemcpy(file.data() + 122, dOut, pixels * 4, hipMemcpyDeviceToHost);
    if (dRgb)
        (void)hipFree(dRgb);
    if (dOut)
        (void)hipFree(dOut);
    if (e != hipSuccess)
        return fail(ctx, RT_ERR_DEVICE, "rt_write_bmp: %s", hipGetErrorString(e));
    rt_bmp_header(width, height, file.data(), 122);
    FILE* fh = std::fopen(path, "wb");
    if (!fh)
        return fail(ctx, RT_ERR_INVALID, "rt_write_bmp: cannot open %s", path);
    const size_t wrote = std::fwrite(file.data(), 1, file.size(), fh);
    std::fclose(fh);
    if (wrote != file.size())
        return fail(ctx, RT_ERR_INVALID, "rt_write_bmp: short write to %s", path);
    return RT_OK;
} catch (...) {
    return caught(ctx, "rt_write_bmp");
}

size_t rt_bmp_header(int width, int height, uint8_t* out, size_t out_bytes)
{
    // stbi_write_bmp for 4 components: 14-byte file header + 108-byte V4 header,
    // BI_BITFIELDS, 32 bpp, masks R 0xff0000 G 0xff00 B 0xff A 0xff000000
    const size_t kBytes = 14 + 108;
    if (!out || out_bytes < kBytes || width <= 0 || height <= 0)
        return kBytes;
    std::memset(out, 0, kBytes);
    auto put16 = [&](size_t at, uint32_t v) { out[at] = v & 255, out[at + 1] = (v >> 8) & 255; };
    auto put32 = [&](size_t at, uint32_t v) { put16(at, v & 0xffff), put16(at + 2, v >> 16); };
    out[0] = 'B', out[1] = 'M';
    put32(2, static_cast<uint32_t>(kBytes + static_cast<size_t>(width) * height * 4));
    put32(10, static_cast<uint32_t>(kBytes));
    put32(14, 108);
    put32(18, static_cast<uint32_t>(width));
    put32(22, static_cast<uint32_t>(height));
    put16(26, 1);
    put16(28, 32);
    put32(30, 3);
    put32(54, 0xff0000u);
    put32(58, 0xff00u);
    put32(62, 0xffu);
    put32(66, 0xff000000u);
    return kBytes;
}

int rt_sync(rt_ctx* ctx, void* stream, rt_stats* stats)
try {
    if (!ctx)
        return fail(nullptr, RT_ERR_INVALID, "rt_sync: ctx is NULL");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, hipStreamSynchronize(static_cast<hipStream_t>(stream)));
    if (stats) {
        std::memset(stats, 0, sizeof *stats);
        return readStats(ctx, stats, true);
    }
    // No statistics asked for: the one counter that says whether the frame is valid is looked at all the same.
    if (ctx->lastCounters) {
        unsigned long long overflow = 0;
        HIP_TRY(ctx, hipMemcpy(&overflow, reinterpret_cast<const char*>(ctx->lastCounters) + offsetof(DevCounters, overflow), sizeof overflow,
            hipMemcpyDeviceToHost));
        if (overflow) {
            ctx->evUsed = ctx->marksUsed = 0;
            return fail(ctx, RT_ERR_INTERNAL, "render: %llu hits found no slot in the hit pool (the frame is invalid; this is a bug in the pool's sizing)",
                overflow);
        }
    }
    return RT_OK;
} catch (...) {
    return caught(ctx, "rt_sync");
}

int rt_deinterleave(rt_ctx* ctx, const void* d_packed, int width, int height, int shard_count, void* d_frame, void* stream)
try {
    if (!ctx)
        return fail(nullptr, RT_ERR_INVALID, "rt_deinterleave: ctx is NULL");
    if (!d_packed || !d_frame || width <= 0 || height <= 0 || shard_count <= 0)
        return fail(ctx, RT_ERR_INVALID, "rt_deinterleave: bad argument");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const int tilesX = (width + kTile - 1) / kTile, tilesY = (height + kTile - 1) / kTile;
    const size_t stride = shardTiles(width, height, 0, shard_count) * 64 * 3;
    HIP_TRY(ctx, launchDeinterleave(static_cast<hipStream_t>(stream), static_cast<const float*>(d_packed),
                      static_cast<float*>(d_frame), width, height, tilesX, tilesY, shard_count, stride));
    return RT_OK;
} catch (...) {
    return caught(ctx, "rt_deinterleave");
}

int rt_intersect_batch(rt_ctx* ctx, rt_ray* rays, uint32_t n, rt_hit* hits)
try {
    if (!ctx)
        return fail(nullptr, RT_ERR_INVALID, "rt_intersect_batch: ctx is NULL");
    if (!ctx->haveScene)
        return fail(ctx, RT_ERR_STATE, "rt_intersect_batch: no scene uploaded");
    if (!ctx->haveFeatures)
        return fail(ctx, RT_ERR_STATE, "rt_intersect_batch: no features set");
    if (n == 0)
        return RT_OK;
    if (!rays || !hits)
        return fail(ctx, RT_ERR_INVALID, "rt_intersect_batch: NULL array");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    rt_ray* dRays = nullptr;
    rt_hit* dHits = nullptr;
    HIP_TRY(ctx, hipMalloc(reinterpret_cast<void**>(&dRays), n * sizeof(rt_ray)));
    hipError_t e = hipMalloc(reinterpret_cast<void**>(&dHits), n * sizeof(rt_hit));
    if (e == hipSuccess)
        e = hipMemcpy(dRays, rays, n * sizeof(rt_ray), hipMemcpyHostToDevice);
    if (e == hipSuccess) {
        const DevFeat ft = makeFeat(ctx->features);
        const int blockThreads = kBlockThreads;
        const size_t lds = static_cast<size_t>(std::max<uint32_t>(ctx->numLevels, 2)) * blockThreads * 4; // as in renderCommon
        e = launchIntersect(blockThreads, lds, nullptr, ctx->scene, ft, dRays, dHits, n);
    }
    if (e == hipSuccess)
        e = hipDeviceSynchronize();
    if (e == hipSuccess)
        e = hipMemcpy(rays, dRays, n * sizeof(rt_ray), hipMemcpyDeviceToHost);
    if (e == hipSuccess)
        e = hipMemcpy(hits, dHits, n * sizeof(rt_hit), hipMemcpyDeviceToHost);
    (void)hipFree(dRays);
    if (dHits)
        (void)hipFree(dHits);
    if (e != hipSuccess)
        return fail(ctx, RT_ERR_DEVICE, "rt_intersect_batch: %s", hipGetErrorString(e));
    return RT_OK;
} catch (...) {
    return caught(ctx, "rt_intersect_batch");
}

int rt_eval_batch(rt_ctx* ctx, int kind, const float* in, uint32_t n, float* out, const void* aux)
try {
    if (!ctx)
        return fail(nullptr, RT_ERR_INVALID, "rt_eval_batch: ctx is NULL");
    static const int inW[] = { 12, 12, 9, 3, 3, 2, 3, 2, 2, 20 };
    static const int outW[] = { 3, 3, 2, 3, 3, 6, 6, 0, 6, 3 };
    if (kind < 0 || kind > RT_EVAL_SHADING)
        return fail(ctx, RT_ERR_INVALID, "rt_eval_batch: unknown kind %d", kind);
    if (n == 0)
        return RT_OK;
    if (!in || !out)
        return fail(ctx, RT_ERR_INVALID, "rt_eval_batch: NULL array");
    size_t inFloats = static_cast<size_t>(inW[kind]) * n, outFloats = static_cast<size_t>(outW[kind]) * n;
    if (kind == RT_EVAL_SAMPLER) {
        if (n != 1)
            return fail(ctx, RT_ERR_INVALID, "rt_eval_batch: the sampler kind takes one record");
        uint32_t cnt;
        std::memcpy(&cnt, in + 1, 4);
        if (cnt == 0 || cnt > (1u << 24))
            return fail(ctx, RT_ERR_INVALID, "rt_eval_batch: sampler count %u", cnt);
        outFloats = cnt;
    }
    if ((kind == RT_EVAL_TEX_NEAREST || kind == RT_EVAL_TEX_BILINEAR)) {
        if (!ctx->haveScene)
            return fail(ctx, RT_ERR_STATE, "rt_eval_batch: texture kinds need an uploaded scene");
        for (uint32_t i = 0; i < n; i++)
            if (!(in[3 * i] >= 0.0f && in[3 * i] < (float)ctx->scene.nTex))
                return fail(ctx, RT_ERR_INVALID, "rt_eval_batch: record %u names image %g of %d", i, in[3 * i], ctx->scene.nTex);
    }
    if (kind == RT_EVAL_SEGMENT_LIGHT || kind == RT_EVAL_PARALLELOGRAM) {
        const int w = inW[kind];
        for (uint32_t i = 0; i < n; i++)
            if (!(in[w * i] >= 0.0f && in[w * i] < (float)ctx->scene.nLights))
                return fail(ctx, RT_ERR_INVALID, "rt_eval_batch: record %u names light %g of %d", i, in[w * i], ctx->scene.nLights);
    }
    DevCamera cam {};
    if (kind == RT_EVAL_CAMERA_RAY) {
        if (!aux)
            return fail(ctx, RT_ERR_INVALID, "rt_eval_batch: the camera-ray kind needs aux = rt_camera*");
        cam = makeCamera(*static_cast<const rt_camera*>(aux));
    }
    rt_features f = ctx->features;
    if (!ctx->haveFeatures)
        std::memset(&f, 0, sizeof f);
    const DevFeat ft = makeFeat(f);
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    float *dIn = nullptr, *dOut = nullptr;
    HIP_TRY(ctx, hipMalloc(reinterpret_cast<void**>(&dIn), inFloats * sizeof(float)));
    hipError_t e = hipMalloc(reinterpret_cast<void**>(&dOut), outFloats * sizeof(float));
    if (e == hipSuccess)
        e = hipMemcpy(dIn, in, inFloats * sizeof(float), hipMemcpyHostToDevice);
    if (e == hipSuccess)
        e = launchEval(nullptr, ctx->scene, cam, ft, kind, dIn, dOut, n);
    if (e == hipSuccess)
        e = hipDeviceSynchronize();
    if (e == hipSuccess)
        e = hipMemcpy(out, dOut, outFloats * sizeof(float), hipMemcpyDeviceToHost);
    (void)hipFree(dIn);
    if (dOut)
        (void)hipFree(dOut);
    if (e != hipSuccess)
        return fail(ctx, RT_ERR_DEVICE, "rt_eval_batch: %s", hipGetErrorString(e));
    return RT_OK;
} catch (...) {
    return caught(ctx, "rt_eval_batch");
}

} // extern "C"
