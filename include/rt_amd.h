/*
 * rt_amd.h -- C ABI of the MI355X rendering hot path.
 *
 * The reference has no plugin/FFI layer; its seam for this path is the free
 * function renderImage (src/render.h:34) plus the virtual BVHInterface
 * (src/bvh_interface.h:12-72).  Each entry point below names the reference
 * interface it stands in for.  Only plain pointers, sizes and POD structs
 * cross this boundary; the library owns all device memory.
 *
 * Conventions: every call returns 0 on success and a negative rt_status on
 * failure; rt_last_error() returns the message of the last failure on that
 * context (or of the last context-less failure when ctx is NULL).  Calls on
 * one context must be serialised by the caller; one context drives one GPU
 * and carries the work of ONE stream at a time: its lights, counters and
 * working memory are shared by the frames enqueued through it, which the
 * order of a single stream keeps apart (lights re-sent between two frames
 * reach the device in that stream's order, behind the earlier frame).  Frames
 * meant to overlap -- several in flight on several streams -- take one
 * context each (bench.py does that).
 */
#ifndef RT_AMD_H
#define RT_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum rt_status {
    RT_OK = 0,
    RT_ERR_INVALID = -1,     /* bad argument / malformed BVH */
    RT_ERR_UNSUPPORTED = -2, /* outside this path (bloom sizes the reference cannot run, sample counts beyond a counter's width) */
    RT_ERR_DEVICE = -3,      /* HIP runtime failure */
    RT_ERR_STATE = -4,       /* call order (render before scene upload, ...) */
    RT_ERR_NOMEM = -5,       /* host memory ran out inside the library */
    RT_ERR_INTERNAL = -6     /* any other failure inside the library's host code (no exception crosses the C ABI) */
};

/* Vertex, framework/include/framework/mesh.h:14-20 (32 B) */
typedef struct {
    float position[3];
    float normal[3];
    float texCoord[2];
} rt_vertex;

/* BVHInterface::Primitive, src/bvh_interface.h:14-23 (100 B) */
typedef struct {
    uint32_t meshID;
    rt_vertex v0, v1, v2;
} rt_primitive;

/* BVHInterface::Node, src/bvh_interface.h:28-54 (32 B): inner {left,right},
 * leaf {RT_LEAF_BIT|offset, count}. */
#define RT_LEAF_BIT 0x80000000u
typedef struct {
    float lower[3];
    float upper[3];
    uint32_t data[2];
} rt_node;

/* Material, framework/include/framework/mesh.h:22-34; kdTexture indexes the
 * image table passed to rt_scene_upload, -1 = none. */
typedef struct {
    float kd[3];
    float ks[3];
    float shininess;
    float transparency;
    int32_t kdTexture;
} rt_material;

/* Image, framework/include/framework/image.h:14-24: row-major RGB floats. */
typedef struct {
    const float* pixels;
    int32_t width;
    int32_t height;
} rt_image;

/* Sphere, src/common.h:39-43 */
typedef struct {
    float center[3];
    float radius;
    rt_material material;
} rt_sphere;

/* PointLight / SegmentLight / ParallelogramLight, src/common.h:45-60.
 * type 0: v = position, color.  type 1: endpoint0, endpoint1, color0, color1.
 * type 2: v0, edge01, edge02, color0, color1, color2, color3. */
typedef struct {
    int32_t type;
    float v[21];
} rt_light;

/* Features + ExtraFeatures, src/common.h:62-106, one field per member in
 * declaration order.  maxRayDepth: the reference hard-codes `rayDepth < 6`
 * (src/recursive.cpp:43); 6 reproduces it, at most 7 is accepted. */
typedef struct {
    int32_t enableShading;
    int32_t enableReflections;
    int32_t enableShadows;
    int32_t enableNormalInterp;
    int32_t enableTextureMapping;
    int32_t enableAccelStructure;
    int32_t enableBilinearTextureFiltering;
    int32_t enableTransparency;
    int32_t enableJitteredSampling;
    int32_t shadingModel;
    uint32_t numPixelSamples;
    uint32_t numShadowSamples;
    int32_t enableBvhSahBinning;
    int32_t enableBloomEffect;
    int32_t enableDepthOfField;
    int32_t enableEnvironmentMap;
    int32_t enableGlossyReflection;
    int32_t enableMipmapTextureFiltering;
    int32_t enableMotionBlur;  /* renderImageWithMotionBlur, src/extra.cpp:121-211; depth of field, if also on, takes the frame (src/render.cpp:22-25) */
    int32_t motionBlurSamples; /* time samples per pixel, at most 65535 */
    float movement;            /* splineMat's scale, src/extra.cpp:546-567 */
    uint32_t numGlossySamples;
    float glossyExponent;
    uint32_t bloomFilterSize;
    float bloomThreshold;
    float depthOfFieldDistance;
    float depthOfFieldCircleDiameter;
    uint32_t numDepthOfFieldSamples;
    int32_t maxRayDepth;
} rt_features;

/* Trackball state, framework/src/trackball.cpp:23-31,64-69: angles in
 * radians, aspect = window width / height (framework/src/window.cpp:413-418). */
typedef struct {
    float lookAt[3];
    float rotation[3];
    float distance;
    float fovy;
    float aspect;
} rt_camera;

/* Ray, framework/include/framework/ray.h:9-13 */
typedef struct {
    float origin[3];
    float direction[3];
    float t;
} rt_ray;

/* What BVHInterface::intersect leaves behind (src/bvh_interface.h:61): the
 * bool result, ray.t, and the HitInfo fields (src/common.h:22-27), with the
 * material given as its slot (mesh id, or n_meshes + sphere index) and the
 * index into primitives() of the accepted hit (-1 for none / a sphere). */
typedef struct {
    int32_t hit;
    float t;
    int32_t primitive;
    int32_t material;
    float normal[3];
    float barycentricCoord[3];
    float texCoord[2];
} rt_hit;

typedef struct {
    uint64_t rays;        /* BVHInterface::intersect invocations */
    uint64_t inner_nodes; /* sibling-pair records fetched (0 unless collected) */
    uint64_t tri_tests;   /* triangle records tested   (0 unless collected) */
    float kernel_ms;      /* device time of the render kernel launches since the last
                             stats read, summed; HIP events on the launch stream */
    float total_ms;       /* wall time of the call */
    uint32_t launches;    /* render launches kernel_ms covers */
    uint32_t batches;     /* pixel batches of those launches (wavefront path): path_ms and shadow_ms
                             sum over them; a frame whose working set exceeds the workspace is
                             rendered in several */
    /* the share of the shadow-ray kernel (wavefront path; zero on the per-pixel path) */
    uint64_t shadow_rays;
    uint64_t shadow_inner_nodes;
    uint64_t shadow_tri_tests;
    float path_ms;        /* camera-ray / mirror-chain kernel, summed like kernel_ms */
    float shadow_ms;      /* shadow-ray kernel, summed like kernel_ms */
} rt_stats;

typedef struct rt_ctx rt_ctx;

/* Context on one HIP device.  No reference counterpart (the reference is a
 * single-process CPU program); it owns what `Scene`/`BVH` own there. */
int rt_ctx_create(int device_ordinal, rt_ctx** out);
void rt_ctx_destroy(rt_ctx* ctx);

/* Behaviour of a context that the reference has no notion of (no reference counterpart).
 *  RT_OPT_TREE              which hierarchy the rays are walked through, from the next rt_scene_upload on:
 *                           2 (default) the library's own over single primitives, 1 its own over the
 *                           caller's leaves, 0 the caller's inner nodes as they are.  Results do not depend on it.
 *  RT_OPT_PER_PIXEL         1 = every frame on the kernel that walks a pixel in renderRay's own depth-first
 *                           order; 0 (default) = the queue-based renderer.  Same bits either way: a cross-check.
 *  RT_OPT_STRICT_INTERSECT  1 = rt_intersect_batch replays every ray in the reference's own order of box
 *                           and triangle tests (src/bvh.cpp:245-316).  A cross-check: by default only rays
 *                           are replayed that met equal distances, or whose hit the reference's box routine
 *                           (which loses hits by rounding) may have kept it from; the answers are the same.
 *  RT_OPT_WORKSPACE_BYTES   cap on the working memory of the queue-based renderer (default 192 GiB, never more
 *                           than the device has free); a frame that needs more is rendered in pixel batches.
 *  RT_OPT_TIMING            0 = record no events around the render launches (rt_stats times read 0).
 *  RT_OPT_TREE_BUILD        who builds the library's own hierarchy at rt_scene_upload: 0 (default) the host, sweeping
 *                           every split position of every axis with the surface-area heuristic; 1 the device, with
 *                           32 bins per axis and all nodes of a level at once -- an upload several times faster
 *                           (scenes edited between frames) for a tree a few per cent more expensive to traverse.
 *                           Results do not depend on it.
 *  RT_OPT_REFERENCE_BOXES   1 (default) = a hit is reported only if the reference's own box routine, which is not
 *                           conservative, lets the ray into the caller's leaf that holds it (checked per hit; the
 *                           doubtful ones are replayed in the reference's order of tests): rays aimed exactly at
 *                           vertices or starting in a face of a box get the reference's answer.  0 = the nearest hit
 *                           whatever that routine would have done (one or two per cent faster on large frames,
 *                           more on frames of a tenth of a millisecond; differs from the reference on such rays only). */
enum rt_option {
    RT_OPT_TREE = 1,
    RT_OPT_PER_PIXEL = 2,
    RT_OPT_STRICT_INTERSECT = 3,
    RT_OPT_WORKSPACE_BYTES = 4,
    RT_OPT_TIMING = 5,
    RT_OPT_TREE_BUILD = 6,
    RT_OPT_REFERENCE_BOXES = 7
};
int rt_ctx_set_option(rt_ctx* ctx, int option, int64_t value);
const char* rt_last_error(const rt_ctx* ctx);

/* BVH::BVH(scene, features), src/bvh.cpp:52-101 with buildRecursive
 * (:424-493) and the median split (:208-223) or, with sah_binning != 0
 * (Features::extra.enableBvhSahBinning), splitPrimitivesBySAHBin
 * (src/extra.cpp:493-543); host code.  prims_in is the scene-order primitive
 * list (:66-77).  nodes_out needs max(2, 2*n) entries, prims_out n entries. */
int rt_bvh_build(const rt_primitive* prims_in, uint32_t n, int sah_binning, rt_node* nodes_out,
                 rt_primitive* prims_out, uint32_t* n_nodes, uint32_t* n_levels, uint32_t* n_leaves);

/* The same constructor with its work on the GPU of `ctx`.  Median split: the tree's shape follows from n alone
 * under the split at (n + 1) / 2 and is laid out on the host; boxes and the per-node sorts by centroid
 * (src/bvh.cpp:208-223) run level by level on the device.  A range in which two centroid coordinates are equal
 * has no unique sorted arrangement (the reference's sort is not stable), so such ranges are sorted once more by
 * the host's std::sort from the arrangement they had; host_sorted (optional) receives how many primitives
 * went that way.  sah_binning != 0: splitPrimitivesBySAHBin (src/extra.cpp:493-543) on the device -- buckets,
 * the cost of every split position, the first cheapest one -- with the host only collecting the split positions
 * of a level to lay out the next.  Either way the node and primitive arrays are rt_bvh_build's, bit for bit.
 * Host pointers in and out, like rt_bvh_build. */
int rt_bvh_build_device(rt_ctx* ctx, const rt_primitive* prims_in, uint32_t n, int sah_binning, rt_node* nodes_out,
                        rt_primitive* prims_out, uint32_t* n_nodes, uint32_t* n_levels, uint32_t* n_leaves,
                        uint32_t* host_sorted);

/* Hands the library what renderImage reads through `scene` and `bvh`
 * (src/render.h:17-26): bvh.nodes(), bvh.primitives() (src/bvh_interface.h:64-67),
 * scene.meshes[i].material, the textures they name, the six
 * environmentMapTextures (src/scene.h:40-48; image index or -1) and
 * scene.spheres.  All host memory stays caller-owned and is copied; the
 * device layout is built here.  The caller's nodes are validated and define
 * which primitives exist and in which order; the rays are walked through a
 * hierarchy of the library's own over those primitives (which primitive a ray
 * hits and its t do not depend on the hierarchy; of two primitives hit at
 * exactly the same t the reference reports the one it tests later, and so
 * does the library: the nodes are also kept as they came, and such a ray is
 * traced once more over them in the reference's order of tests,
 * src/bvh.cpp:245-316).  generation (optional) receives a counter that
 * changes with every upload. */
int rt_scene_upload(rt_ctx* ctx, const rt_node* nodes, uint32_t n_nodes, const rt_primitive* prims,
                    uint32_t n_prims, const rt_material* mesh_materials, uint32_t n_meshes,
                    const rt_image* images, uint32_t n_images, const int32_t env_faces[6],
                    const rt_sphere* spheres, uint32_t n_spheres, uint64_t* generation);

/* scene.lights (src/scene.h:36) and the Features argument of renderImage;
 * cheap, may be re-sent before every frame (the GUI mutates both between
 * calls, src/main.cpp:179-270,340-418).  rt_lights_set copies the lights to
 * the device before it returns (to one of four slots in turn; frames already
 * enqueued keep reading the slot they were enqueued with, so it does not wait
 * for them unless four further calls have come round to their slot). */
int rt_lights_set(rt_ctx* ctx, const rt_light* lights, uint32_t n);
int rt_features_set(rt_ctx* ctx, const rt_features* features);

/* renderImage(scene, bvh, features, camera, screen), src/render.cpp:19-51,
 * including the depth-of-field variant (src/extra.cpp:89-113).  Synchronous.
 * out_rgb (host) receives width*height*3 floats in Screen order, index
 * (height-1-y)*width + x (src/screen.cpp:36-42).  stats may be NULL. */
int rt_render(rt_ctx* ctx, const rt_camera* camera, int width, int height, float* out_rgb, rt_stats* stats);

/* Same frame, device-resident and sharded by interleaved 8x8 pixel tiles:
 * tile k (row-major over ceil(width/8) x ceil(height/8)) belongs to shard
 * k % shard_count.  With shard_count == 1 d_out is the whole frame in Screen
 * order (width*height*3 floats); otherwise it is this shard's packed tiles,
 * rt_shard_floats() floats.  The kernel is enqueued on `stream` (a
 * hipStream_t, NULL = default stream) and the call returns without waiting;
 * collect_stats != 0 makes it wait and fill *stats including node/triangle
 * counters.  No reference counterpart: pixels are independent there
 * (src/render.cpp:34-42), which is what makes the split legal. */
int rt_render_device(rt_ctx* ctx, const rt_camera* camera, int width, int height, int shard_index,
                     int shard_count, void* d_out, size_t d_out_floats, void* stream,
                     int collect_stats, rt_stats* stats);
size_t rt_shard_floats(int width, int height, int shard_index, int shard_count);

/* rt_render_device for callers that cannot allocate device memory while they run: code under HIP
 * graph capture, tight serving loops, harnesses that forbid hipMalloc.  No counterpart in the
 * reference.  Same frame, same arguments, same bits as rt_render_device, but the call allocates
 * nothing (no hipMalloc, no event, no stream): counters, ray queues, the pools of hits and light
 * samples and the bloom copy are carved from a static array in device memory,
 * RT_NOALLOC_SCRATCH_BYTES per GPU, which every GPU the library has a context on holds until the
 * process ends.  A frame whose working set exceeds it is rendered in more pixel batches (slower, the
 * same result); refused with RT_ERR_UNSUPPORTED are a frame of which one 8x8 tile alone needs more
 * than the scratch (hundreds of camera rays per pixel times many light samples at a deep ray depth)
 * and, with bloom on and shard_count == 1, a frame whose copy (width * height * 12 bytes) does not
 * fit.  No timing is taken (events are allocations); rt_sync still returns the ray count.  The
 * call enqueues kernels and memsets on `stream`, nothing else; replaying it from a HIP graph is not
 * claimed (it worked in a process of its own and faulted after other work in one process).  The
 * scratch is one per GPU, not per context: calls that overlap on one GPU (two streams, two contexts)
 * race, and so do a call here and the stats read of an earlier one. */
#define RT_NOALLOC_SCRATCH_BYTES ((size_t)128 << 20)
int rt_render_device_noalloc(rt_ctx* ctx, const rt_camera* camera, int width, int height, int shard_index,
                             int shard_count, void* d_out, size_t d_out_floats, void* stream);

/* Scatter shard_count packed tile buffers (as laid end to end by a gather,
 * each rt_shard_floats(.., 0, shard_count) floats apart) into a frame in
 * Screen order.  Device pointers; enqueued on `stream`. */
int rt_deinterleave(rt_ctx* ctx, const void* d_packed, int width, int height, int shard_count,
                    void* d_frame, void* stream);

/* postprocessImageWithBloom (src/extra.cpp:276-316, called at the end of
 * renderImage, src/render.cpp:48-50) on a whole frame in Screen order in device
 * memory, in place, under the features last set; nothing happens when
 * enableBloomEffect is off.  rt_render and rt_render_device with one shard run
 * it themselves; a sharded frame needs it once, on the rank that holds the
 * frame after rt_deinterleave (a shard has no neighbours).  Filter sizes the
 * reference cannot run without writing outside its image (below 3) or
 * overflowing its factorial (above 20) are refused by rt_features_set.  The
 * reference's filter window reaches one row above the image (undefined
 * behaviour there): those reads count as black here. */
int rt_postprocess_device(rt_ctx* ctx, void* d_frame, int width, int height, void* stream);

/* Screen::writeBitmapToFile (src/screen.cpp:44-55): clamp to [0,1], times 255,
 * truncated to 8 bits, alpha 255 -- written as the pixel array of the file
 * stbi_write_bmp produces for 4 components: rows bottom-up, bytes B G R A,
 * width * height * 4 bytes at d_bgra (device memory).  rt_bmp_header writes the
 * 122 bytes that precede them in the file (14-byte file header + 108-byte V4
 * header, BI_BITFIELDS) into `out` and returns that size; with out == NULL it
 * only returns the size. */
int rt_pack_bmp_device(rt_ctx* ctx, const void* d_frame, int width, int height, void* d_bgra, void* stream);
size_t rt_bmp_header(int width, int height, uint8_t* out, size_t out_bytes);

/* Screen::writeBitmapToFile (src/screen.cpp:44-55) for a frame in host memory
 * (width * height * 3 floats, Screen order): the same conversion on the device,
 * then the file.  Not a hot path: allocates and frees its device buffers. */
int rt_write_bmp(rt_ctx* ctx, const float* rgb, int width, int height, const char* path);

/* Blocks until the work rt_render_device / rt_deinterleave enqueued on
 * `stream` has finished; stats (optional) receives the ray count of the last
 * render and the summed device time of all render launches since the last
 * stats read. */
int rt_sync(rt_ctx* ctx, void* stream, rt_stats* stats);

/* BVHInterface::intersect(state, ray, hitInfo), src/bvh_interface.h:61 ->
 * intersectRayWithBVH, src/bvh.cpp:337-358, for n rays at once.  rays[i].t is
 * updated in place like `ray.t`; hits[i] receives the HitInfo.  Uses the
 * features last set (enableAccelStructure, enableNormalInterp,
 * enableTextureMapping). */
int rt_intersect_batch(rt_ctx* ctx, rt_ray* rays, uint32_t n, rt_hit* hits);

/* The unit-tested pure functions, evaluated on the device.  in/out are packed
 * float records, n of them:
 *  RT_EVAL_BARYCENTRIC      in v0 v1 v2 p (12)      out bc (3)      src/interpolate.cpp:13-24
 *  RT_EVAL_INTERP_NORMAL    in n0 n1 n2 bc (12)     out n (3)       src/interpolate.cpp:34-40
 *  RT_EVAL_INTERP_TEXCOORD  in t0 t1 t2 bc (9)      out uv (2)      src/interpolate.cpp:50-55
 *  RT_EVAL_TEX_NEAREST      in image, u v (3)       out rgb (3)     src/texture.cpp:12-35
 *  RT_EVAL_TEX_BILINEAR     in image, u v (3)       out rgb (3)     src/texture.cpp:44-95
 *  RT_EVAL_SEGMENT_LIGHT    in light, sample (2)    out pos col (6) src/light.cpp:26-36
 *  RT_EVAL_PARALLELOGRAM    in light, sx sy (3)     out pos col (6) src/light.cpp:60-74
 *  RT_EVAL_SAMPLER          in seed, count (2, as uint32 bits; n must be 1)
 *                           out count floats                         src/sampler.h:36-39
 *  RT_EVAL_CAMERA_RAY       in ndc x y (2)          out o d (6)     framework/src/trackball.cpp:101-110
 *                           (camera = aux)
 *  RT_EVAL_SHADING          in material(8: kd ks shininess transparency) normal cameraDir
 *                           lightDir lightColor (20) out rgb (3)    src/shading.cpp:29-56
 * "image"/"light" are indices (float-encoded) into the uploaded image table /
 * the lights last set.  aux is NULL except where noted. */
enum rt_eval_kind {
    RT_EVAL_BARYCENTRIC = 0,
    RT_EVAL_INTERP_NORMAL = 1,
    RT_EVAL_INTERP_TEXCOORD = 2,
    RT_EVAL_TEX_NEAREST = 3,
    RT_EVAL_TEX_BILINEAR = 4,
    RT_EVAL_SEGMENT_LIGHT = 5,
    RT_EVAL_PARALLELOGRAM = 6,
    RT_EVAL_SAMPLER = 7,
    RT_EVAL_CAMERA_RAY = 8,
    RT_EVAL_SHADING = 9
};
int rt_eval_batch(rt_ctx* ctx, int kind, const float* in, uint32_t n, float* out, const void* aux);

#ifdef __cplusplus
}
#endif
#endif
