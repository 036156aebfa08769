// rt_host.hpp -- C++ host side above the C ABI (include/rt_amd.h), mirroring the
// reference's interface for the rendering path: same type and function names,
// argument meaning and error behaviour (exceptions), so code written against
// src/render.h, src/bvh.h, src/scene.h and framework/{mesh,ray,image,trackball}.h
// reads the same here.  Header-only; link against librt_amd.so.
//
// Differences that are deliberate: the vector types are small local structs with
// glm's member names (glm itself is third-party code of the reference and is not
// carried), `Trackball` is headless (the reference's needs a GLFW window only for
// its aspect ratio, framework/src/trackball.cpp:27), and BVH::intersect /
// renderImage run on the GPU.
#pragma once

#include <array>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <limits>
#include <memory>
#include <random>
#include <concepts>
#include <type_traits>
#include <span>
#include <stdexcept>
#include <string>
#include <variant>
#include <filesystem>
#include <vector>

#include "../../include/rt_amd.h"

namespace glmlike {
struct vec2 {
    float x = 0, y = 0;
    vec2() = default;
    explicit vec2(float s) : x(s), y(s) {}
    vec2(float x_, float y_) : x(x_), y(y_) {}
    float& operator[](int i) { return i == 0 ? x : y; }
    float operator[](int i) const { return i == 0 ? x : y; }
    bool operator==(const vec2&) const = default;
};
struct vec3 {
    float x = 0, y = 0, z = 0;
    vec3() = default;
    explicit vec3(float s) : x(s), y(s), z(s) {}
    vec3(float x_, float y_, float z_) : x(x_), y(y_), z(z_) {}
    float& operator[](int i) { return i == 0 ? x : (i == 1 ? y : z); }
    float operator[](int i) const { return i == 0 ? x : (i == 1 ? y : z); }
    bool operator==(const vec3&) const = default;
};
struct ivec2 {
    int x = 0, y = 0;
    ivec2() = default;
    ivec2(int x_, int y_) : x(x_), y(y_) {}
};
struct uvec3 {
    uint32_t x = 0, y = 0, z = 0;
    uvec3() = default;
    uvec3(uint32_t x_, uint32_t y_, uint32_t z_) : x(x_), y(y_), z(z_) {}
};
inline float radians(float deg) { return deg * 0.01745329251994329576923690768489f; }
inline vec3 radians(const vec3& d) { return { radians(d.x), radians(d.y), radians(d.z) }; }
} // namespace glmlike

namespace rthost {
using glmlike::ivec2;
using glmlike::uvec3;
using glmlike::vec2;
using glmlike::vec3;

// framework/include/framework/image.h:14-24 (pixels only; decoding files is scene ingest)
struct Image {
    int width = 0, height = 0;
    std::vector<vec3> pixels;
};

// framework/include/framework/mesh.h:14-43
struct Vertex {
    vec3 position;
    vec3 normal;
    vec2 texCoord;
    bool operator==(const Vertex&) const = default;
};
struct Material {
    vec3 kd;
    vec3 ks { 0.0f };
    float shininess { 1.0f };
    float transparency { 1.0f };
    std::shared_ptr<Image> kdTexture;
};
struct Mesh {
    std::vector<Vertex> vertices;
    std::vector<uvec3> triangles;
    Material material;
};

// framework/include/framework/ray.h:9-13
struct Ray {
    vec3 origin { 0.0f };
    vec3 direction { 0.0f, 0.0f, -1.0f };
    float t { std::numeric_limits<float>::max() };
};

// src/common.h:15-106
enum class ShadingModel { Lambertian = 0, Phong = 1, BlinnPhong = 2, LinearGradient = 3 };
struct HitInfo {
    vec3 normal;
    vec3 barycentricCoord;
    vec2 texCoord;
    Material material;
};
struct AxisAlignedBox {
    vec3 lower { 0.0f };
    vec3 upper { 1.0f };
};
struct Sphere {
    vec3 center { 0.0f };
    float radius = 1.0f;
    Material material;
};
struct PointLight {
    vec3 position, color;
};
struct SegmentLight {
    vec3 endpoint0, endpoint1, color0, color1;
};
struct ParallelogramLight {
    vec3 v0, edge01, edge02, color0, color1, color2, color3;
};
struct ExtraFeatures {
    bool enableBvhSahBinning = false;
    bool enableBloomEffect = false;
    bool enableDepthOfField = false;
    bool enableEnvironmentMap = false;
    bool enableGlossyReflection = false;
    bool enableMipmapTextureFiltering = false;
    bool enableMotionBlur = false;
    int motionBlurSamples = 0;
    float movement = 0.0f;
    uint32_t numGlossySamples = 1;
    float glossyExponent = 1.f;
    uint32_t bloomFilterSize = 3;
    float bloomThreshold = 0.5f;
    float depthOfFieldDistance = 1.0f;
    float depthOfFieldCircleDiameter = 0.1f;
    uint32_t numDepthOfFieldSamples = 5;
};
struct Features {
    bool enableShading = false;
    bool enableReflections = false;
    bool enableShadows = false;
    bool enableNormalInterp = false;
    bool enableTextureMapping = false;
    bool enableAccelStructure = false;
    bool enableBilinearTextureFiltering = false;
    bool enableTransparency = false;
    bool enableJitteredSampling = false;
    ShadingModel shadingModel = ShadingModel::Lambertian;
    uint32_t numPixelSamples = 1;
    uint32_t numShadowSamples = 4;
    ExtraFeatures extra = {};
};

// src/scene.h:15-49
enum SceneType { SingleTriangle, Cube, CubeTextured, CornellBox, CornellBoxTransparency, CornellBoxParallelogramLight, Monkey, Teapot, Dragon, Spheres, Custom };
struct Scene {
    using SceneLight = std::variant<PointLight, SegmentLight, ParallelogramLight>;
    SceneType type = Custom;
    std::vector<Mesh> meshes;
    std::vector<Sphere> spheres;
    std::vector<SceneLight> lights;
    std::shared_ptr<Image> environmentMapTextures[6];
};

// src/sampler.h:13-46: the generator behind every random draw of the path, for host code that
// wants the same stream (the device seeds its own per pixel, src/render.cpp:38).
class Sampler {
public:
    Sampler(uint32_t seed = std::random_device()()) : m_state(seed) {}
    float next_1d()
    {
        m_state = m_state * 747796405u + 2891336453u;
        uint32_t v = m_state;
        v ^= v >> ((v >> 28u) + 4u);
        v *= 277803737u;
        v ^= v >> 22u;
        return static_cast<float>(v) / 4294967295.f;
    }
    vec2 next_2d()
    {
        const float a = next_1d(); // (first draw is x: the reference's braced initialiser evaluates left to right)
        const float b = next_1d();
        return { a, b };
    }

private:
    uint32_t m_state;
};

struct BVHInterface;
// src/render.h:17-26
struct RenderState {
    const Scene& scene;
    const Features& features;
    const BVHInterface& bvh;
    Sampler sampler;
};

// The one rt_ctx of the process (device RT_AMD_DEVICE, default 0).  Failures of the
// C ABI become exceptions, like the reference's loaders throw.
class Device {
public:
    static Device& get()
    {
        static Device d;
        return d;
    }
    rt_ctx* ctx() { return m_ctx; }
    void check(int rc, const char* what)
    {
        if (rc != RT_OK)
            throw std::runtime_error(std::string(what) + ": " + rt_last_error(m_ctx));
    }
    uint64_t uploadedHash = 0;
    // Releases the context now (a program that wants its device memory back, or that does not want
    // the release to happen among the static destructors at exit, where the GPU runtime may be gone).
    void shutdown()
    {
        if (m_ctx)
            rt_ctx_destroy(m_ctx);
        m_ctx = nullptr;
        uploadedHash = 0;
    }

private:
    Device()
    {
        int dev = 0;
        if (const char* s = std::getenv("RT_AMD_DEVICE"))
            dev = std::atoi(s);
        if (rt_ctx_create(dev, &m_ctx) != RT_OK)
            throw std::runtime_error(std::string("rt_ctx_create: ") + rt_last_error(nullptr));
    }
    ~Device() { shutdown(); }
    rt_ctx* m_ctx = nullptr;
};

namespace detail {
inline void put(float* d, const vec3& v) { d[0] = v.x, d[1] = v.y, d[2] = v.z; }
inline rt_vertex toC(const Vertex& v)
{
    rt_vertex o;
    put(o.position, v.position);
    put(o.normal, v.normal);
    o.texCoord[0] = v.texCoord.x, o.texCoord[1] = v.texCoord.y;
    return o;
}
inline Vertex fromC(const rt_vertex& v)
{
    return { { v.position[0], v.position[1], v.position[2] }, { v.normal[0], v.normal[1], v.normal[2] }, { v.texCoord[0], v.texCoord[1] } };
}
inline rt_features toC(const Features& f, int maxRayDepth = 6)
{
    rt_features o {};
    o.enableShading = f.enableShading, o.enableReflections = f.enableReflections, o.enableShadows = f.enableShadows;
    o.enableNormalInterp = f.enableNormalInterp, o.enableTextureMapping = f.enableTextureMapping;
    o.enableAccelStructure = f.enableAccelStructure, o.enableBilinearTextureFiltering = f.enableBilinearTextureFiltering;
    o.enableTransparency = f.enableTransparency, o.enableJitteredSampling = f.enableJitteredSampling;
    o.shadingModel = static_cast<int>(f.shadingModel);
    o.numPixelSamples = f.numPixelSamples, o.numShadowSamples = f.numShadowSamples;
    const ExtraFeatures& e = f.extra;
    o.enableBvhSahBinning = e.enableBvhSahBinning, o.enableBloomEffect = e.enableBloomEffect;
    o.enableDepthOfField = e.enableDepthOfField, o.enableEnvironmentMap = e.enableEnvironmentMap;
    o.enableGlossyReflection = e.enableGlossyReflection, o.enableMipmapTextureFiltering = e.enableMipmapTextureFiltering;
    o.enableMotionBlur = e.enableMotionBlur, o.motionBlurSamples = e.motionBlurSamples, o.movement = e.movement;
    o.numGlossySamples = e.numGlossySamples, o.glossyExponent = e.glossyExponent;
    o.bloomFilterSize = e.bloomFilterSize, o.bloomThreshold = e.bloomThreshold;
    o.depthOfFieldDistance = e.depthOfFieldDistance, o.depthOfFieldCircleDiameter = e.depthOfFieldCircleDiameter;
    o.numDepthOfFieldSamples = e.numDepthOfFieldSamples;
    o.maxRayDepth = maxRayDepth; // src/recursive.cpp:43
    return o;
}
inline uint64_t fnv(const void* p, size_t n, uint64_t h)
{
    const unsigned char* b = static_cast<const unsigned char*>(p);
    for (size_t i = 0; i < n; i++)
        h = (h ^ b[i]) * 1099511628211ull;
    return h;
}
} // namespace detail

// src/bvh_interface.h:12-72
struct BVHInterface {
    struct Primitive {
        uint32_t meshID;
        Vertex v0, v1, v2;
        bool operator==(const Primitive&) const = default;
    };
    struct Node {
        static constexpr uint32_t LeafBit = 1u << 31;
        AxisAlignedBox aabb;
        std::array<uint32_t, 2> data;
        bool isLeaf() const { return (data[0] & LeafBit) == LeafBit; }
        uint32_t primitiveOffset() const { return data[0] & (~LeafBit); }
        uint32_t primitiveCount() const { return data[1]; }
        uint32_t leftChild() const { return data[0]; }
        uint32_t rightChild() const { return data[1]; }
    };
    static_assert(sizeof(Node) == 32);
    static_assert(sizeof(Primitive) == 100);

    virtual ~BVHInterface() {}
    virtual bool intersect(RenderState& state, Ray& ray, HitInfo& hitInfo) const = 0;
    virtual std::span<const Node> nodes() const = 0;
    virtual std::span<Node> nodes() = 0;
    virtual std::span<const Primitive> primitives() const = 0;
    virtual std::span<Primitive> primitives() = 0;
    virtual uint32_t numLevels() const = 0;
    virtual uint32_t numLeaves() const = 0;
};

// Copies what renderImage reads through `scene` and `bvh` to the device when its
// content changed since the last upload (the GUI mutates scenes between calls).
inline void syncScene(const Scene& scene, const BVHInterface& bvh, const Features& features)
{
    Device& dev = Device::get();
    const auto nodes = bvh.nodes();
    const auto prims = bvh.primitives();
    std::vector<rt_material> mats;
    std::vector<const Image*> imagePtrs;
    std::vector<rt_image> images;
    auto imageIndex = [&](const std::shared_ptr<Image>& im) -> int32_t {
        if (!im)
            return -1;
        for (size_t i = 0; i < imagePtrs.size(); i++)
            if (imagePtrs[i] == im.get())
                return static_cast<int32_t>(i);
        imagePtrs.push_back(im.get());
        images.push_back({ reinterpret_cast<const float*>(im->pixels.data()), im->width, im->height });
        return static_cast<int32_t>(imagePtrs.size() - 1);
    };
    auto toMat = [&](const Material& m) {
        rt_material o;
        detail::put(o.kd, m.kd);
        detail::put(o.ks, m.ks);
        o.shininess = m.shininess, o.transparency = m.transparency;
        o.kdTexture = imageIndex(m.kdTexture);
        return o;
    };
    for (const Mesh& m : scene.meshes)
        mats.push_back(toMat(m.material));
    std::vector<rt_sphere> spheres;
    for (const Sphere& s : scene.spheres) {
        rt_sphere o;
        detail::put(o.center, s.center);
        o.radius = s.radius;
        o.material = toMat(s.material);
        spheres.push_back(o);
    }
    int32_t env[6];
    for (int i = 0; i < 6; i++)
        env[i] = imageIndex(scene.environmentMapTextures[i]);

    // What is on the device is known by the identity of the large arrays (which BVH object, where its
    // node and primitive arrays and the images' pixels live, how large they are) and by the content of
    // the small tables (materials, spheres, environment faces): cheap enough for every frame.  A caller
    // that rewrites nodes, primitives or texels in place says so with invalidateScene().
    uint64_t h = 1469598103934665603ull;
    const void* ids[] = { &bvh, nodes.data(), prims.data() };
    const size_t sizes[] = { nodes.size(), prims.size(), scene.meshes.size(), scene.spheres.size() };
    h = detail::fnv(ids, sizeof ids, h);
    h = detail::fnv(sizes, sizeof sizes, h);
    h = detail::fnv(mats.data(), mats.size() * sizeof(rt_material), h);
    h = detail::fnv(spheres.data(), spheres.size() * sizeof(rt_sphere), h);
    h = detail::fnv(env, sizeof env, h);
    for (const rt_image& im : images) {
        h = detail::fnv(&im.pixels, sizeof im.pixels, h);
        h = detail::fnv(&im.width, 2 * sizeof(int32_t), h);
    }
    if (h == 0)
        h = 1; // 0 stands for "nothing uploaded"
    if (h != dev.uploadedHash) {
        dev.check(rt_scene_upload(dev.ctx(), reinterpret_cast<const rt_node*>(nodes.data()), static_cast<uint32_t>(nodes.size()),
                      reinterpret_cast<const rt_primitive*>(prims.data()), static_cast<uint32_t>(prims.size()), mats.data(),
                      static_cast<uint32_t>(mats.size()), images.data(), static_cast<uint32_t>(images.size()), env, spheres.data(),
                      static_cast<uint32_t>(spheres.size()), nullptr),
            "rt_scene_upload");
        dev.uploadedHash = h;
    }
    std::vector<rt_light> lights;
    for (const auto& l : scene.lights) {
        rt_light o {};
        if (const auto* p = std::get_if<PointLight>(&l)) {
            o.type = 0;
            detail::put(o.v + 0, p->position), detail::put(o.v + 3, p->color);
        } else if (const auto* s = std::get_if<SegmentLight>(&l)) {
            o.type = 1;
            detail::put(o.v + 0, s->endpoint0), detail::put(o.v + 3, s->endpoint1), detail::put(o.v + 6, s->color0), detail::put(o.v + 9, s->color1);
        } else {
            const auto& q = std::get<ParallelogramLight>(l);
            o.type = 2;
            detail::put(o.v + 0, q.v0), detail::put(o.v + 3, q.edge01), detail::put(o.v + 6, q.edge02);
            detail::put(o.v + 9, q.color0), detail::put(o.v + 12, q.color1), detail::put(o.v + 15, q.color2), detail::put(o.v + 18, q.color3);
        }
        lights.push_back(o);
    }
    dev.check(rt_lights_set(dev.ctx(), lights.data(), static_cast<uint32_t>(lights.size())), "rt_lights_set");
    const rt_features f = detail::toC(features);
    dev.check(rt_features_set(dev.ctx(), &f), "rt_features_set");
}

// The next renderImage / intersect uploads the scene again whatever it looks like (after writing
// into nodes(), primitives() or an image's pixels in place).
inline void invalidateScene() { Device::get().uploadedHash = 0; }

// src/bvh.h:60-125
struct BVH : public BVHInterface {
    static constexpr uint32_t LeafSize = 4;
    static constexpr uint32_t RootIndex = 0;

    BVH(const Scene& scene, const Features& features)
    {
        std::vector<Primitive> in;
        for (uint32_t meshID = 0; meshID < scene.meshes.size(); meshID++) { // src/bvh.cpp:66-77
            const Mesh& mesh = scene.meshes[meshID];
            for (const auto& t : mesh.triangles)
                in.push_back(Primitive { meshID, mesh.vertices[t.x], mesh.vertices[t.y], mesh.vertices[t.z] });
        }
        const uint32_t n = static_cast<uint32_t>(in.size());
        m_nodes.resize(std::max<size_t>(2, 2 * size_t(n)));
        m_primitives.resize(n);
        uint32_t nNodes = 0;
        const int rc = rt_bvh_build(reinterpret_cast<const rt_primitive*>(in.data()), n, features.extra.enableBvhSahBinning,
            reinterpret_cast<rt_node*>(m_nodes.data()), reinterpret_cast<rt_primitive*>(m_primitives.data()), &nNodes, &m_numLevels,
            &m_numLeaves);
        if (rc != RT_OK)
            throw std::runtime_error(std::string("rt_bvh_build: ") + rt_last_error(nullptr));
        m_nodes.resize(nNodes);
    }

    // BVHInterface::intersect -> intersectRayWithBVH (src/bvh.cpp:337-358), on the device
    bool intersect(RenderState& state, Ray& ray, HitInfo& hitInfo) const override
    {
        syncScene(state.scene, *this, state.features);
        Device& dev = Device::get();
        rt_ray r;
        detail::put(r.origin, ray.origin), detail::put(r.direction, ray.direction);
        r.t = ray.t;
        rt_hit h;
        dev.check(rt_intersect_batch(dev.ctx(), &r, 1, &h), "rt_intersect_batch");
        if (!h.hit)
            return false;
        ray.t = r.t;
        hitInfo.normal = { h.normal[0], h.normal[1], h.normal[2] };
        if (h.primitive >= 0) { // sphere hits leave these two untouched, like intersectRayWithShape(Sphere)
            hitInfo.barycentricCoord = { h.barycentricCoord[0], h.barycentricCoord[1], h.barycentricCoord[2] };
            if (state.features.enableTextureMapping)
                hitInfo.texCoord = { h.texCoord[0], h.texCoord[1] };
        }
        const size_t nMeshes = state.scene.meshes.size();
        hitInfo.material = (size_t(h.material) < nMeshes) ? state.scene.meshes[h.material].material
                                                          : state.scene.spheres[h.material - nMeshes].material;
        return true;
    }

    std::span<const Node> nodes() const override { return m_nodes; }
    std::span<Node> nodes() override { return m_nodes; }
    std::span<const Primitive> primitives() const override { return m_primitives; }
    std::span<Primitive> primitives() override { return m_primitives; }
    uint32_t numLevels() const override { return m_numLevels; }
    uint32_t numLeaves() const override { return m_numLeaves; }

private:
    uint32_t m_numLevels = 0, m_numLeaves = 0;
    std::vector<Node> m_nodes;
    std::vector<Primitive> m_primitives;
};

// src/screen.h:10-33 without the OpenGL presentation
class Screen {
public:
    explicit Screen(const ivec2& resolution, bool = false)
        : m_resolution(resolution), m_textureData(size_t(resolution.x) * resolution.y, vec3(0.0f))
    {
    }
    void clear(const vec3& color) { std::fill(m_textureData.begin(), m_textureData.end(), color); }
    void setPixel(int x, int y, const vec3& color) { m_textureData[indexAt(x, y)] = color; }
    ivec2 resolution() const { return m_resolution; }
    int indexAt(int x, int y) const { return (m_resolution.y - 1 - y) * m_resolution.x + x; } // src/screen.cpp:119-122
    const std::vector<vec3>& pixels() const { return m_textureData; }
    std::vector<vec3>& pixels() { return m_textureData; }
    // src/screen.cpp:44-55; the 8-bit conversion runs on the device
    void writeBitmapToFile(const std::filesystem::path& filePath)
    {
        Device& dev = Device::get();
        dev.check(rt_write_bmp(dev.ctx(), &m_textureData[0].x, m_resolution.x, m_resolution.y, filePath.string().c_str()), "rt_write_bmp");
    }

private:
    ivec2 m_resolution;
    std::vector<vec3> m_textureData;
};

// framework/include/framework/trackball.h:14-59, headless: the window only ever
// supplied the aspect ratio (framework/src/trackball.cpp:27).
class Trackball {
public:
    Trackball(float aspectRatio, float fovy, float distFromLookAt, float rotationX = 0.0f, float rotationY = 0.0f)
        : m_aspect(aspectRatio), m_fovy(fovy), m_distanceFromLookAt(distFromLookAt), m_rotationEulerAngles(rotationX, rotationY, 0.0f)
    {
    }
    void setCamera(const vec3 lookAt, const vec3 rotations, const float dist)
    {
        m_lookAt = lookAt, m_rotationEulerAngles = rotations, m_distanceFromLookAt = dist;
    }
    vec3 lookAt() const { return m_lookAt; }
    vec3 rotationEulerAngles() const { return m_rotationEulerAngles; }
    float distanceFromLookAt() const { return m_distanceFromLookAt; }
    rt_camera toC() const
    {
        rt_camera c;
        detail::put(c.lookAt, m_lookAt), detail::put(c.rotation, m_rotationEulerAngles);
        c.distance = m_distanceFromLookAt, c.fovy = m_fovy, c.aspect = m_aspect;
        return c;
    }

private:
    float m_aspect, m_fovy, m_distanceFromLookAt;
    vec3 m_lookAt { 0.0f };
    vec3 m_rotationEulerAngles;
};

// src/render.h:34 / src/render.cpp:19-51
inline void renderImage(const Scene& scene, const BVHInterface& bvh, const Features& features, const Trackball& camera, Screen& screen)
{
    syncScene(scene, bvh, features);
    Device& dev = Device::get();
    const rt_camera cam = camera.toC();
    const ivec2 res = screen.resolution();
    dev.check(rt_render(dev.ctx(), &cam, res.x, res.y, reinterpret_cast<float*>(screen.pixels().data()), nullptr), "rt_render");
}

// The shape of the interface that code written against the reference relies on, as the
// reference's src/verification.cpp:11-144 pins it: member names and exact member types of Scene,
// RenderState, Features, BVHInterface and its Node / Primitive records, and BVH deriving from
// BVHInterface.  A change to this header that breaks one of them does not compile.
namespace contract {
template <typename T, typename M>
concept exactly = std::same_as<T, M>;

template <typename S>
concept scene_shape = requires(S s) {
    { s.type } -> exactly<SceneType&>;
    { s.meshes } -> exactly<std::vector<Mesh>&>;
    { s.spheres } -> exactly<std::vector<Sphere>&>;
    { s.lights } -> exactly<std::vector<typename S::SceneLight>&>;
};
template <typename R>
concept render_state_shape = requires(R r) {
    { r.scene } -> exactly<const Scene&>;
    { r.features } -> exactly<const Features&>;
    { r.bvh } -> exactly<const BVHInterface&>;
    { r.sampler } -> exactly<Sampler&>;
};
template <typename F>
concept features_shape = requires(F f) {
    { f.enableShading } -> exactly<bool&>;
    { f.enableReflections } -> exactly<bool&>;
    { f.enableShadows } -> exactly<bool&>;
    { f.enableNormalInterp } -> exactly<bool&>;
    { f.enableTextureMapping } -> exactly<bool&>;
    { f.enableAccelStructure } -> exactly<bool&>;
    { f.enableBilinearTextureFiltering } -> exactly<bool&>;
    { f.enableTransparency } -> exactly<bool&>;
    { f.shadingModel } -> exactly<ShadingModel&>;
    { f.numPixelSamples } -> exactly<uint32_t&>;
    { f.numShadowSamples } -> exactly<uint32_t&>;
    { f.extra } -> exactly<ExtraFeatures&>;
};
template <typename N>
concept node_shape = requires(N n) {
    { n.aabb } -> exactly<AxisAlignedBox&>;
    { n.data } -> exactly<std::array<uint32_t, 2>&>;
};
template <typename P>
concept primitive_shape = requires(P p) {
    { p.meshID } -> exactly<uint32_t&>;
    { p.v0 } -> exactly<Vertex&>;
};
template <typename B>
concept bvh_interface_shape = requires(B b, RenderState& state, Ray& ray, HitInfo& hitInfo) {
    { b.nodes() } -> exactly<std::span<BVHInterface::Node>>;
    { b.primitives() } -> exactly<std::span<BVHInterface::Primitive>>;
    { b.numLevels() } -> exactly<uint32_t>;
    { b.numLeaves() } -> exactly<uint32_t>;
    { b.intersect(state, ray, hitInfo) } -> exactly<bool>;
};
static_assert(scene_shape<Scene>, "Scene: type / meshes / spheres / lights");
static_assert(render_state_shape<RenderState>, "RenderState: scene / features / bvh / sampler");
static_assert(features_shape<Features>, "Features: the eight switches, shadingModel, sample counts, extra");
static_assert(bvh_interface_shape<BVHInterface>, "BVHInterface: nodes / primitives / numLevels / numLeaves / intersect");
static_assert(node_shape<BVHInterface::Node>, "BVHInterface::Node: aabb + data[2]");
static_assert(primitive_shape<BVHInterface::Primitive>, "BVHInterface::Primitive: meshID + vertices");
static_assert(std::is_base_of_v<BVHInterface, BVH>, "BVH implements BVHInterface");
} // namespace contract

} // namespace rthost
