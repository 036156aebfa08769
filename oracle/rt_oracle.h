/*
 * rt_oracle.h -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * CPU restatement of the per-pixel rendering path of the reference ray tracer
 * (renderImage -> BVH traversal -> ray/triangle test -> lights/shadows ->
 * shading -> interpolation/textures/environment map -> recursive bounces).
 * It exists to check the HIP path and to be timed as the CPU baseline; only
 * tests, the smoke check and the benchmark's cpu_baseline leg may load it.
 * The product library never links or calls anything in this directory.
 *
 * Parity pinning: the restatement is checked against every known-answer value
 * the reference's own tests hold for this path (pyramid t values and node
 * layout, BVH shape table, interpolation / texture / light KATs).  The
 * reference itself cannot be built in this image (it needs GLU/GLFW and ships
 * its intersection arithmetic only as a prebuilt archive, which is never
 * loaded), so whole-image results have no reference-generated vectors:
 * image-level parity is "parity unpinned" beyond those KATs.
 *
 * All structs are plain C, laid out exactly like the reference's own types
 * where one exists (Vertex 32 B, BVHInterface::Primitive 100 B,
 * BVHInterface::Node 32 B).
 */
#ifndef RT_ORACLE_H
#define RT_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* framework/include/framework/mesh.h:14-20 */
typedef struct {
    float position[3];
    float normal[3];
    float texCoord[2];
} orc_vertex;

/* src/bvh_interface.h:14-23 */
typedef struct {
    uint32_t meshID;
    orc_vertex v0, v1, v2;
} orc_primitive;

/* src/bvh_interface.h:28-54; data[0] carries LeafBit (1u<<31) */
typedef struct {
    float lower[3];
    float upper[3];
    uint32_t data[2];
} orc_node;

/* framework/include/framework/mesh.h:22-34; kdTexture is an index into the
 * image table, -1 for "no texture" (the reference holds a shared_ptr). */
typedef struct {
    float kd[3];
    float ks[3];
    float shininess;
    float transparency;
    int32_t kdTexture;
} orc_material;

/* framework/include/framework/image.h:14-24: row-major RGB float texels */
typedef struct {
    const float* pixels;
    int32_t width;
    int32_t height;
} orc_image;

/* src/common.h:39-43 */
typedef struct {
    float center[3];
    float radius;
    orc_material material;
} orc_sphere;

/* src/common.h:45-60, flattened: type 0 point {position,color}; 1 segment
 * {endpoint0,endpoint1,color0,color1}; 2 parallelogram {v0,edge01,edge02,
 * color0,color1,color2,color3}; unused floats are ignored. */
typedef struct {
    int32_t type;
    float v[21];
} orc_light;

/* src/common.h:62-106, one int32/float per member, same order.
 * maxRayDepth has no counterpart: the reference hard-codes `rayDepth < 6`
 * (src/recursive.cpp:43); 6 reproduces it. */
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
    /* ExtraFeatures */
    int32_t enableBvhSahBinning;
    int32_t enableBloomEffect;
    int32_t enableDepthOfField;
    int32_t enableEnvironmentMap;
    int32_t enableGlossyReflection;
    int32_t enableMipmapTextureFiltering;
    int32_t enableMotionBlur;
    int32_t motionBlurSamples;
    float movement;
    uint32_t numGlossySamples;
    float glossyExponent;
    uint32_t bloomFilterSize;
    float bloomThreshold;
    float depthOfFieldDistance;
    float depthOfFieldCircleDiameter;
    uint32_t numDepthOfFieldSamples;
    int32_t maxRayDepth;
} orc_features;

/* Trackball state (framework/src/trackball.cpp:23-31,64-69): angles in
 * radians, aspect = width/height of the window the camera was made for. */
typedef struct {
    float lookAt[3];
    float rotation[3];
    float distance;
    float fovy;
    float aspect;
} orc_camera;

/* framework/include/framework/ray.h:9-13 */
typedef struct {
    float origin[3];
    float direction[3];
    float t;
} orc_ray;

/* Result of BVHInterface::intersect (src/bvh_interface.h:61) plus the index
 * of the accepted closest primitive in primitives() (-1: none or a sphere)
 * and the material slot (mesh id, or n_meshes + sphere index). */
typedef struct {
    int32_t hit;
    float t;
    int32_t primitive;
    int32_t material;
    float normal[3];
    float barycentricCoord[3];
    float texCoord[2];
} orc_hit;

typedef struct {
    uint64_t rays;        /* BVHInterface::intersect invocations */
    uint64_t inner_nodes; /* inner nodes popped (sibling-pair fetches) */
    uint64_t box_tests;
    uint64_t tri_tests;
} orc_stats;

typedef struct orc_scene orc_scene;

/* BVH::BVH (src/bvh.cpp:52-101) with the median split (src/bvh.cpp:208-223) or,
 * sah_binning != 0, splitPrimitivesBySAHBin (src/extra.cpp:493-543).
 * nodes_out needs capacity max(2, 2*n) entries, prims_out n entries.  Returns 0. */
int orc_bvh_build(const orc_primitive* prims_in, uint32_t n, int sah_binning,
                  orc_node* nodes_out, orc_primitive* prims_out,
                  uint32_t* n_nodes, uint32_t* n_levels, uint32_t* n_leaves);

orc_scene* orc_scene_create(const orc_node* nodes, uint32_t n_nodes,
                            const orc_primitive* prims, uint32_t n_prims,
                            const orc_material* mesh_materials, uint32_t n_meshes,
                            const orc_image* images, uint32_t n_images,
                            const int32_t env_faces[6],
                            const orc_sphere* spheres, uint32_t n_spheres,
                            const orc_light* lights, uint32_t n_lights);
void orc_scene_destroy(orc_scene* scene);

/* policy 0: the reference's traversal (src/bvh.cpp:245-316, children tested
 * with t reset to FLT_MAX, no culling); policy 1: same tests, but a child whose
 * entry distance exceeds the current ray.t is skipped (the device policy).
 * Rays are updated in place (ray.t), hits written to out[i]. */
void orc_intersect_batch(const orc_scene* scene, const orc_features* features,
                         orc_ray* rays, uint32_t n, orc_hit* out, int policy,
                         orc_stats* stats);

/* renderImage (src/render.cpp:19-51).  out_rgb: width*height*3 floats in
 * Screen order, index (height-1-y)*width+x (src/screen.cpp:36-42).
 * threads <= 0: all OpenMP threads.  Returns 0, -1 for what it does not do
 * (motion blur without depth of field: orc_render_motion_blur; bloom filter
 * sizes outside [3, 20], which the reference itself cannot run without
 * undefined behaviour). */
/* postprocessImageWithBloom (src/extra.cpp:276-316) on a frame in Screen order,
 * in place; no-op when enableBloomEffect is off.  Returns 0, or -1 for a filter
 * size outside [3, 20]. */
int orc_bloom(const orc_features* features, float* rgb, int width, int height);

/* The bytes of the file Screen::writeBitmapToFile writes (src/screen.cpp:44-55):
 * 122 header bytes + width * height * 4 pixel bytes into `out`. */
void orc_bmp_bytes(const float* rgb, int width, int height, uint8_t* out);

int orc_render(const orc_scene* scene, const orc_features* features,
               const orc_camera* camera, int width, int height, float* out_rgb,
               orc_stats* stats, int policy, int threads);

/* renderImageWithMotionBlur (src/extra.cpp:121-211) + the bloom pass.  The
 * reference rebuilds a BVH over a moved copy of the scene for every time sample
 * of every pixel, from the meshes' triangles in the scene's own order:
 * scene_order_prims holds them in that order (what was handed to orc_bvh_build,
 * not what it returned).  The sum over the time samples starts at zero; the
 * reference leaves it uninitialised.  Returns 0, -1 when motion blur is off or
 * depth of field is on (renderImage then takes the other renderer). */
int orc_render_motion_blur(const orc_scene* scene, const orc_primitive* scene_order_prims, uint32_t n_prims,
                           const orc_features* features, const orc_camera* camera, int width, int height,
                           float* out_rgb, orc_stats* stats, int threads);

/* Pixel-centre primary rays of the given camera (src/render.cpp:68-69):
 * closest-hit primitive index and t per pixel, Screen order. */
void orc_primary_hits(const orc_scene* scene, const orc_features* features,
                      const orc_camera* camera, int width, int height,
                      int32_t* prim_out, float* t_out);

/* Camera rays, for checking ray generation on its own: Trackball::generateRay
 * (framework/src/trackball.cpp:101-110) at NDC positions xy[2*i], xy[2*i+1]. */
void orc_generate_rays(const orc_camera* camera, const float* ndc_xy, uint32_t n, orc_ray* out);

/* Unit-tested free functions. */
void orc_sampler_next(uint32_t seed, uint32_t n, float* out);                    /* src/sampler.h:36-39 */
void orc_barycentric(const float v0[3], const float v1[3], const float v2[3],
                     const float p[3], float out[3]);                            /* src/interpolate.cpp:13-24 */
void orc_interpolate_normal(const float n0[3], const float n1[3], const float n2[3],
                            const float bc[3], float out[3]);                    /* src/interpolate.cpp:34-40 */
void orc_interpolate_texcoord(const float t0[2], const float t1[2], const float t2[2],
                              const float bc[3], float out[2]);                  /* src/interpolate.cpp:50-55 */
void orc_sample_texture(const orc_image* image, const float uv[2], int bilinear,
                        float out[3]);                                           /* src/texture.cpp:12-95 */
void orc_sample_segment_light(float sample, const orc_light* light, float pos[3],
                              float color[3]);                                   /* src/light.cpp:26-36 */
void orc_sample_parallelogram_light(const float sample[2], const orc_light* light,
                                    float pos[3], float color[3]);               /* src/light.cpp:60-74 */
void orc_point_of_focus(const orc_camera* camera, const orc_ray* ray, float focal_distance,
                        float out[3]);                                           /* src/extra.cpp:15-23 */
void orc_camera_basis(const orc_camera* camera, float position[3], float forward[3],
                      float up[3], float left[3]);                               /* framework/src/trackball.cpp:71-74,112-126 */
int orc_triangle_test(const float v0[3], const float v1[3], const float v2[3], orc_ray* ray);
int orc_aabb_test(const float lower[3], const float upper[3], orc_ray* ray);
int orc_sphere_test(const orc_sphere* s, orc_ray* ray, float normal_out[3]);
/* computeShading (src/shading.cpp:29-56) for a material without texture. */
void orc_compute_shading(const orc_features* features, const orc_material* material,
                         const float normal[3], const float camera_dir[3],
                         const float light_dir[3], const float light_color[3], float out[3]);
float orc_linear_gradient_sample_r(float ti, float out[3]);                       /* src/shading.cpp:155-189 */

#ifdef __cplusplus
}
#endif
#endif
