// host_tests.cpp -- the reference's BVH / render tests written against the C++
// host mirror (rt_host.hpp); everything below runs on the GPU through the C ABI.
// A frame rendered through renderImage is compared with the CPU oracle
// (test infrastructure, linked only into this test binary).
#include <cstdio>
#include <cstdlib>

#include "../../oracle/rt_oracle.h"
#include "../../ray-tracer_amd/host/rt_host.hpp"

using namespace rthost;

static int g_failed = 0, g_checks = 0;
#define CHECK(cond)                                                          \
    do {                                                                     \
        g_checks++;                                                          \
        if (!(cond)) {                                                       \
            g_failed++;                                                      \
            std::printf("FAILED %s:%d: %s\n", __FILE__, __LINE__, #cond);    \
        }                                                                    \
    } while (0)

static Features getFeatures(bool enableBVH = true)
{
    Features f;
    f.enableAccelStructure = enableBVH;
    return f;
}

// tests/bvh_tests.cpp:817-1009
static void pyramidInsidePyramid()
{
    const std::vector<Vertex> v {
        Vertex { vec3(1, -2, -2) }, Vertex { vec3(5, -1, 1) }, Vertex { vec3(-4, 0, 2) }, Vertex { vec3(2, 4, 0) },
        Vertex { vec3(0.1f, -0.2f, -0.2f) }, Vertex { vec3(0.5f, -0.1f, 0.1f) }, Vertex { vec3(-0.4f, 0.0f, 0.2f) }, Vertex { vec3(0.2f, 0.4f, 0.0f) }
    };
    const std::vector<uvec3> tris { uvec3(0, 1, 2), uvec3(0, 1, 3), uvec3(0, 2, 3), uvec3(1, 2, 3) };
    Scene scene;
    scene.meshes = { Mesh { { v[0], v[1], v[2], v[3] }, tris, Material { vec3(0.0f) } },
        Mesh { { v[4], v[5], v[6], v[7] }, tris, Material { vec3(0.0f) } } };
    Features on = getFeatures(), off = getFeatures(false);
    BVH bvh(scene, on);
    CHECK(bvh.numLeaves() == 2);
    CHECK(bvh.numLevels() == 2);
    const auto nodes = bvh.nodes();
    CHECK(nodes[0].aabb.lower == vec3(-4, -2, -2) && nodes[0].aabb.upper == vec3(5, 4, 2));
    CHECK(!nodes[0].isLeaf() && nodes[0].leftChild() == 2 && nodes[0].rightChild() == 3);
    CHECK(nodes[2].isLeaf() && nodes[2].primitiveCount() == 4 && nodes[2].primitiveOffset() == 0);
    CHECK(nodes[2].aabb.upper == vec3(2, 4, 2));
    CHECK(nodes[3].isLeaf() && nodes[3].primitiveCount() == 4 && nodes[3].primitiveOffset() == 4);
    const int expect[8][4] = { { 0, 0, 2, 3 }, { 1, 4, 6, 7 }, { 1, 4, 5, 6 }, { 1, 5, 6, 7 }, { 1, 4, 5, 7 }, { 0, 0, 1, 2 }, { 0, 1, 2, 3 }, { 0, 0, 1, 3 } };
    const auto prims = bvh.primitives();
    CHECK(prims.size() == 8);
    for (int i = 0; i < 8 && prims.size() == 8; i++)
        CHECK(prims[i].meshID == uint32_t(expect[i][0]) && prims[i].v0 == v[expect[i][1]] && prims[i].v1 == v[expect[i][2]] && prims[i].v2 == v[expect[i][3]]);

    RenderState stateOn { scene, on, bvh }, stateOff { scene, off, bvh };
    const float ox[5] = { -10, -1, 0, 1, 10 };
    const float expectT[4] = { 9.199999809265f, 0.200000092387f, 0.3249999880790f, 2.25f };
    for (int i = 0; i < 5; i++) {
        Ray r1 { vec3(ox[i], 0, 0), vec3(1, 0, 0) }, r2 = r1;
        HitInfo h1, h2;
        const bool hit1 = bvh.intersect(stateOn, r1, h1), hit2 = bvh.intersect(stateOff, r2, h2);
        CHECK(hit1 == (i < 4) && hit2 == (i < 4));
        if (i < 4) {
            CHECK(r1.t == expectT[i] && r2.t == expectT[i]);
            CHECK(h1.normal == h2.normal && h1.barycentricCoord == h2.barycentricCoord);
        }
    }
}

// tests/your_tests.cpp:35-73 in spirit: a small lit scene renders, and the frame
// equals the CPU oracle's (same scene handed to both).
static void renderAgainstOracle()
{
    Scene scene;
    Mesh floor;
    floor.vertices = { Vertex { vec3(-1, -0.5f, -1), vec3(0, 1, 0), vec2(0, 0) }, Vertex { vec3(1, -0.5f, -1), vec3(0, 1, 0), vec2(1, 0) },
        Vertex { vec3(1, -0.5f, 1), vec3(0, 1, 0), vec2(1, 1) }, Vertex { vec3(-1, -0.5f, 1), vec3(0, 1, 0), vec2(0, 1) } };
    floor.triangles = { uvec3(0, 1, 2), uvec3(0, 2, 3) };
    floor.material = Material { vec3(0.7f, 0.7f, 0.6f), vec3(0.3f), 20.0f };
    Mesh blocker;
    blocker.vertices = { Vertex { vec3(-0.3f, 0.0f, -0.3f), vec3(0, 1, 0) }, Vertex { vec3(0.3f, 0.0f, -0.3f), vec3(0, 1, 0) },
        Vertex { vec3(0.0f, 0.4f, 0.3f), vec3(0, 1, 0) } };
    blocker.triangles = { uvec3(0, 1, 2) };
    blocker.material = Material { vec3(0.9f, 0.2f, 0.2f), vec3(0.0f), 1.0f };
    scene.meshes = { floor, blocker };
    scene.lights.emplace_back(PointLight { vec3(0.2f, 1.5f, 0.3f), vec3(1.0f) });
    scene.lights.emplace_back(ParallelogramLight { vec3(-0.2f, 1.2f, 0), vec3(0.4f, 0, 0), vec3(0, 0, 0.4f), vec3(1, 0, 0), vec3(0, 1, 0), vec3(0, 0, 1), vec3(0, 1, 1) });
    Features f = getFeatures();
    f.enableShading = f.enableShadows = f.enableReflections = true;
    f.shadingModel = ShadingModel::BlinnPhong;
    f.numPixelSamples = 4;
    BVH bvh(scene, f);
    const int W = 96, H = 64;
    Screen screen(ivec2(W, H));
    Trackball camera(float(W) / float(H), glmlike::radians(50.0f), 3.0f);
    camera.setCamera(vec3(0.0f), glmlike::radians(vec3(20.0f, 20.0f, 0.0f)), 3.0f);
    renderImage(scene, bvh, f, camera, screen);

    // the same scene through the oracle
    std::vector<orc_material> mats(2);
    for (int i = 0; i < 2; i++) {
        const Material& m = scene.meshes[i].material;
        mats[i] = { { m.kd.x, m.kd.y, m.kd.z }, { m.ks.x, m.ks.y, m.ks.z }, m.shininess, m.transparency, -1 };
    }
    std::vector<orc_light> lights(2);
    lights[0].type = 0;
    const float l0[6] = { 0.2f, 1.5f, 0.3f, 1, 1, 1 };
    std::memcpy(lights[0].v, l0, sizeof l0);
    lights[1].type = 2;
    const float l1[21] = { -0.2f, 1.2f, 0, 0.4f, 0, 0, 0, 0, 0.4f, 1, 0, 0, 0, 1, 0, 0, 0, 1, 0, 1, 1 };
    std::memcpy(lights[1].v, l1, sizeof l1);
    const int32_t env[6] = { -1, -1, -1, -1, -1, -1 };
    orc_scene* os = orc_scene_create(reinterpret_cast<const orc_node*>(bvh.nodes().data()), uint32_t(bvh.nodes().size()),
        reinterpret_cast<const orc_primitive*>(bvh.primitives().data()), uint32_t(bvh.primitives().size()), mats.data(), 2, nullptr, 0, env,
        nullptr, 0, lights.data(), 2);
    const rt_features cf = detail::toC(f);
    orc_features of;
    static_assert(sizeof(orc_features) == sizeof(rt_features));
    std::memcpy(&of, &cf, sizeof of);
    const rt_camera cc = camera.toC();
    orc_camera oc;
    std::memcpy(&oc, &cc, sizeof oc);
    std::vector<float> ref(size_t(W) * H * 3);
    orc_stats st;
    CHECK(orc_render(os, &of, &oc, W, H, ref.data(), &st, 0, 0) == 0);

    // renderImageWithMotionBlur through the same entry (src/render.cpp:24-25, src/extra.cpp:121-211); the
    // oracle builds a tree per time sample, from the triangles in the order the meshes hold them
    {
        Features blur = f;
        blur.numPixelSamples = 1;
        blur.extra.enableMotionBlur = true;
        blur.extra.motionBlurSamples = 2;
        blur.extra.movement = 0.4f;
        const int w = 32, h = 24;
        Screen moving(ivec2(w, h));
        renderImage(scene, bvh, blur, camera, moving);
        std::vector<orc_primitive> sceneOrder;
        for (uint32_t m = 0; m < scene.meshes.size(); m++) {
            for (const uvec3& t : scene.meshes[m].triangles) {
                orc_primitive q;
                static_assert(sizeof(orc_vertex) == sizeof(Vertex));
                q.meshID = m;
                std::memcpy(&q.v0, &scene.meshes[m].vertices[t.x], sizeof(Vertex));
                std::memcpy(&q.v1, &scene.meshes[m].vertices[t.y], sizeof(Vertex));
                std::memcpy(&q.v2, &scene.meshes[m].vertices[t.z], sizeof(Vertex));
                sceneOrder.push_back(q);
            }
        }
        const rt_features cb = detail::toC(blur);
        orc_features ob;
        std::memcpy(&ob, &cb, sizeof ob);
        std::vector<float> want(size_t(w) * h * 3);
        CHECK(orc_render_motion_blur(os, sceneOrder.data(), uint32_t(sceneOrder.size()), &ob, &oc, w, h, want.data(), nullptr, 0) == 0);
        CHECK(std::memcmp(want.data(), moving.pixels().data(), want.size() * sizeof(float)) == 0);
    }
    orc_scene_destroy(os);
    double maxAbs = 0, sum = 0;
    const float* got = reinterpret_cast<const float*>(screen.pixels().data());
    for (size_t i = 0; i < ref.size(); i++) {
        maxAbs = std::max(maxAbs, std::fabs(double(got[i]) - double(ref[i])));
        sum += got[i];
    }
    std::printf("renderImage vs oracle: max abs %.3g, pixel sum %.6f, %llu oracle rays\n", maxAbs, sum, (unsigned long long)st.rays);
    CHECK(sum > 100.0);
    CHECK(maxAbs <= 2e-5);

    // bloom inside renderImage (src/render.cpp:48-50) and Screen::writeBitmapToFile, against the oracle
    Features bloom = f;
    bloom.extra.enableBloomEffect = true;
    bloom.extra.bloomFilterSize = 4;
    bloom.extra.bloomThreshold = 0.3f;
    Screen bright(ivec2(W, H));
    renderImage(scene, bvh, bloom, camera, bright);
    rt_features bf = detail::toC(bloom);
    orc_features obf;
    std::memcpy(&obf, &bf, sizeof obf);
    std::vector<float> expect(got, got + ref.size()); // the un-bloomed device frame through the oracle's bloom
    CHECK(orc_bloom(&obf, expect.data(), W, H) == 0);
    CHECK(std::memcmp(expect.data(), bright.pixels().data(), expect.size() * sizeof(float)) == 0);
    CHECK(std::memcmp(expect.data(), got, expect.size() * sizeof(float)) != 0);
    const std::filesystem::path bmp = std::filesystem::temp_directory_path() / "rt_amd_host_test.bmp";
    bright.writeBitmapToFile(bmp);
    std::vector<uint8_t> want(122 + size_t(W) * H * 4), have(want.size() + 1);
    orc_bmp_bytes(reinterpret_cast<const float*>(bright.pixels().data()), W, H, want.data());
    FILE* fh = std::fopen(bmp.string().c_str(), "rb");
    CHECK(fh != nullptr);
    if (fh) {
        const size_t n = std::fread(have.data(), 1, have.size(), fh);
        std::fclose(fh);
        CHECK(n == want.size() && std::memcmp(have.data(), want.data(), n) == 0);
        std::filesystem::remove(bmp);
    }

    // error behaviour: what the library refuses throws, like the reference's loaders do
    Features wide = f;
    wide.extra.enableBloomEffect = true;
    wide.extra.bloomFilterSize = 21; // the reference's factorial overflows there
    bool threw = false;
    try {
        renderImage(scene, bvh, wide, camera, screen);
    } catch (const std::exception&) {
        threw = true;
    }
    CHECK(threw);
}

// src/sampler.h:16-45: the host-side Sampler draws the oracle's stream
static void samplerStream()
{
    for (uint32_t seed : { 0u, 123u, 1080u * 700u + 3u, 0xffffffffu }) {
        float want[64];
        orc_sampler_next(seed, 64, want);
        Sampler s(seed);
        bool same = true;
        for (int i = 0; i < 64; i += 2) {
            const vec2 v = s.next_2d();
            same = same && v.x == want[i] && v.y == want[i + 1];
        }
        CHECK(same);
    }
}

int main()
{
    samplerStream();
    pyramidInsidePyramid();
    renderAgainstOracle();
    Device::get().shutdown(); // not among the static destructors at exit
    std::printf("%d checks, %d failed\n", g_checks, g_failed);
    return g_failed ? 1 : 0;
}
