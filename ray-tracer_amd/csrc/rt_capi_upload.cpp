// rt_capi_upload.cpp -- rt_scene_upload: the reference's wire format (nodes, primitives, materials, images,
// spheres) checked, flattened into the device layout and sent.

#include "rt_ctx.hpp"

namespace {

template <typename T>
static hipError_t uploadVec(T** dptr, const std::vector<T>& v)
{
    if (*dptr) {
        (void)hipFree(*dptr);
        *dptr = nullptr;
    }
    const size_t bytes = std::max<size_t>(v.size(), 1) * sizeof(T);
    hipError_t e = hipMalloc(reinterpret_cast<void**>(dptr), bytes);
    if (e != hipSuccess)
        return e;
    if (!v.empty())
        e = hipMemcpy(*dptr, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice);
    return e;
}

} // namespace

extern "C" {

// ---------------------------------------------------------------------------
// Scene upload: validate the wire-format BVH, then flatten it.
// ---------------------------------------------------------------------------
int rt_scene_upload(rt_ctx* ctx, const rt_node* nodes, uint32_t n_nodes, const rt_primitive* prims, uint32_t n_prims,
    const rt_material* mesh_materials, uint32_t n_meshes, const rt_image* images, uint32_t n_images,
    const int32_t env_faces[6], const rt_sphere* spheres, uint32_t n_spheres, uint64_t* generation)
try {
    if (!ctx)
        return fail(nullptr, RT_ERR_INVALID, "rt_scene_upload: ctx is NULL");
    if (!nodes || n_nodes == 0)
        return fail(ctx, RT_ERR_INVALID, "rt_scene_upload: a BVH needs at least its root node");
    if ((n_prims && !prims) || (n_meshes && !mesh_materials) || (n_images && !images) || (n_spheres && !spheres))
        return fail(ctx, RT_ERR_INVALID, "rt_scene_upload: NULL array with non-zero count");
    if (n_prims > kLeafOffsetMask)
        return fail(ctx, RT_ERR_UNSUPPORTED, "rt_scene_upload: %u primitives exceed the %u supported", n_prims, kLeafOffsetMask);
    for (uint32_t i = 0; i < n_prims; i++)
        if (prims[i].meshID >= n_meshes)
            return fail(ctx, RT_ERR_INVALID, "rt_scene_upload: primitive %u names mesh %u of %u", i, prims[i].meshID, n_meshes);
    for (uint32_t i = 0; i < n_meshes; i++)
        if (mesh_materials[i].kdTexture >= (int32_t)n_images)
            return fail(ctx, RT_ERR_INVALID, "rt_scene_upload: material %u names image %d of %u", i, mesh_materials[i].kdTexture, n_images);
    for (uint32_t i = 0; i < n_images; i++)
        if (!images[i].pixels || images[i].width <= 0 || images[i].height <= 0)
            return fail(ctx, RT_ERR_INVALID, "rt_scene_upload: image %u is empty", i);
    for (int i = 0; i < 6; i++)
        if (env_faces && env_faces[i] >= (int32_t)n_images)
            return fail(ctx, RT_ERR_INVALID, "rt_scene_upload: environment face %d names image %d of %u", i, env_faces[i], n_images);

    // Walk the tree from the root: every child index in range, every node
    // reached once, leaves inside the primitive array and small enough for the
    // packed child reference.
    std::vector<int32_t> pairIndex(n_nodes, -1);
    std::vector<uint8_t> seen(n_nodes, 0);
    std::vector<uint32_t> innerOrder; // inner nodes in depth-first pre-order
    struct Item {
        uint32_t node, level;
    };
    std::vector<Item> work { { 0u, 1u } };
    uint32_t maxLevel = 1;
    while (!work.empty()) {
        const Item it = work.back();
        work.pop_back();
        if (it.node >= n_nodes)
            return fail(ctx, RT_ERR_INVALID, "rt_scene_upload: child index %u out of %u nodes", it.node, n_nodes);
        if (seen[it.node])
            return fail(ctx, RT_ERR_INVALID, "rt_scene_upload: node %u is reachable twice (not a tree)", it.node);
        seen[it.node] = 1;
        maxLevel = std::max(maxLevel, it.level);
        const rt_node& nd = nodes[it.node];
        if (nd.data[0] & RT_LEAF_BIT) {
            const uint32_t off = nd.data[0] & ~RT_LEAF_BIT, cnt = nd.data[1];
            if (cnt > (uint32_t)kMaxLeafCount)
                return fail(ctx, RT_ERR_UNSUPPORTED, "rt_scene_upload: leaf %u holds %u primitives (max %d)", it.node, cnt, kMaxLeafCount);
            if ((uint64_t)off + cnt > n_prims)
                return fail(ctx, RT_ERR_INVALID, "rt_scene_upload: leaf %u spans [%u, %u) of %u primitives", it.node, off, off + cnt, n_prims);
        } else {
            pairIndex[it.node] = static_cast<int32_t>(innerOrder.size());
            innerOrder.push_back(it.node);
            work.push_back({ nd.data[1], it.level + 1 });
            work.push_back({ nd.data[0], it.level + 1 });
        }
    }
    const uint32_t callerLevels = maxLevel; // only a limit when the caller's own inner nodes are traversed (below)

    auto childRef = [&](uint32_t child) -> uint32_t {
        const rt_node& c = nodes[child];
        if (c.data[0] & RT_LEAF_BIT)
            return kLeafFlag | (c.data[1] << 27) | (c.data[0] & ~RT_LEAF_BIT);
        return static_cast<uint32_t>(pairIndex[child]);
    };

    const rt_node& rootNode = nodes[0];
    QuantGrid grid;
    for (int a = 0; a < 3; a++) {
        grid.gridMin[a] = rootNode.lower[a];
        const double ext = static_cast<double>(rootNode.upper[a]) - static_cast<double>(rootNode.lower[a]);
        grid.gridCell[a] = (ext > 0.0 && std::isfinite(ext)) ? static_cast<float>(ext / 65535.0) : 0.0f;
    }
    const float* gridMin = grid.gridMin;
    const float* gridCell = grid.gridCell;
    auto packBox = [&](const float* lo, const float* hi, uint32_t* w) { grid.packBox(lo, hi, w); };
    std::vector<uint4> pairs;
    uint32_t rootRef = 0;
    bool ownTree = false;
    if (ctx->treeMode != 0 && !innerOrder.empty()) {
        // Traversal tree of the library's own, chosen by the surface-area heuristic: over the
        // caller's primitives, every one a leaf under its own tight box (mode 2), or over the
        // caller's leaves with their primitive ranges and boxes as they are (mode 1).  Which
        // primitive a ray hits and its t do not depend on the hierarchy; only the number of boxes
        // and primitives a ray has to test does.  (Of two primitives hit at exactly the same t
        // the reference keeps the one it tests later; a ray that meets such a pair is traced once
        // more in the reference's own order over the caller's nodes, Tracer::settleTie.)  The median
        // split of src/bvh.cpp:208-223 leaves heavily overlapping boxes, and a leaf of four
        // primitives makes a ray that grazes its box test all four.
        std::vector<TreeLeaf> leaves;
        for (uint32_t i = 0; i < n_nodes; i++) {
            if (!seen[i] || !(nodes[i].data[0] & RT_LEAF_BIT))
                continue;
            TreeLeaf l;
            for (int a = 0; a < 3; a++) {
                l.lo[a] = nodes[i].lower[a], l.hi[a] = nodes[i].upper[a];
                l.c[a] = 0.5f * (l.lo[a] + l.hi[a]);
            }
            l.count = nodes[i].data[1];
            l.ref = kLeafFlag | (nodes[i].data[1] << 27) | (nodes[i].data[0] & ~RT_LEAF_BIT);
            if (ctx->treeMode == 2) { // every primitive a leaf of its own, under its own box
                const uint32_t first = nodes[i].data[0] & ~RT_LEAF_BIT;
                for (uint32_t k = 0; k < nodes[i].data[1]; k++) {
                    const rt_primitive& q = prims[first + k];
                    TreeLeaf t;
                    for (int a = 0; a < 3; a++) {
                        t.lo[a] = std::min(q.v0.position[a], std::min(q.v1.position[a], q.v2.position[a]));
                        t.hi[a] = std::max(q.v0.position[a], std::max(q.v1.position[a], q.v2.position[a]));
                        t.c[a] = 0.5f * (t.lo[a] + t.hi[a]);
                    }
                    t.count = 1;
                    t.ref = kLeafFlag | (1u << 27) | (first + k);
                    leaves.push_back(t);
                }
                continue;
            }
            leaves.push_back(l);
        }
        if (ctx->treeBuildOnDevice) {
            HIP_TRY(ctx, hipSetDevice(ctx->device));
            std::string err;
            const int built = deviceOwnTree(leaves, grid, pairs, rootRef, maxLevel, err);
            if (built == kTreeFailed)
                return fail(ctx, RT_ERR_DEVICE, "rt_scene_upload: %s", err.c_str());
            // a shape the traversal cannot take: the host's builder has its say, then the caller's own nodes,
            // exactly as without the option (results do not depend on which tree is traversed)
            ownTree = built == kTreeBuilt || buildOwnTree(leaves, grid, pairs, rootRef, maxLevel);
        } else {
            ownTree = buildOwnTree(leaves, grid, pairs, rootRef, maxLevel);
        }
    }
    if (!ownTree) {
        if (callerLevels > (uint32_t)kMaxStack)
            return fail(ctx, RT_ERR_UNSUPPORTED, "rt_scene_upload: BVH has %u levels (max %d)", callerLevels, kMaxStack);
        maxLevel = callerLevels;
        pairs.resize(innerOrder.size() * 2);
        for (size_t k = 0; k < innerOrder.size(); k++) {
            const rt_node& nd = nodes[innerOrder[k]];
            uint32_t wl[3], wr[3];
            packBox(nodes[nd.data[0]].lower, nodes[nd.data[0]].upper, wl);
            packBox(nodes[nd.data[1]].lower, nodes[nd.data[1]].upper, wr);
            pairs[2 * k] = make_uint4(wl[0], wl[1], wl[2], wr[0]);
            pairs[2 * k + 1] = make_uint4(wr[1], wr[2], childRef(nd.data[0]), childRef(nd.data[1]));
        }
        const rt_node& r0 = nodes[0];
        rootRef = (r0.data[0] & RT_LEAF_BIT) ? (kLeafFlag | (r0.data[1] << 27) | (r0.data[0] & ~RT_LEAF_BIT)) : 0u;
    }

    // per primitive, the box of the caller's leaf that holds it (Tracer::surelyReached): lower.xyz upper.x | upper.yz 0 0
    std::vector<float4> leafBoxes(static_cast<size_t>(n_prims) * 2, make_float4(0.0f, 0.0f, 0.0f, 0.0f));
    for (uint32_t i = 0; i < n_nodes; i++) {
        if (!seen[i] || !(nodes[i].data[0] & RT_LEAF_BIT))
            continue;
        const uint32_t first = nodes[i].data[0] & ~RT_LEAF_BIT;
        for (uint32_t k = 0; k < nodes[i].data[1]; k++) {
            leafBoxes[2 * static_cast<size_t>(first + k)] = make_float4(nodes[i].lower[0], nodes[i].lower[1], nodes[i].lower[2], nodes[i].upper[0]);
            leafBoxes[2 * static_cast<size_t>(first + k) + 1] = make_float4(nodes[i].upper[1], nodes[i].upper[2], 0.0f, 0.0f);
        }
    }

    std::vector<float4> tris(static_cast<size_t>(n_prims) * 6), attrs(static_cast<size_t>(n_prims) * 4);
    for (uint32_t i = 0; i < n_prims; i++) {
        const rt_primitive& q = prims[i];
        const f3 v0 = { q.v0.position[0], q.v0.position[1], q.v0.position[2] };
        const f3 v1 = { q.v1.position[0], q.v1.position[1], q.v1.position[2] };
        const f3 v2 = { q.v2.position[0], q.v2.position[1], q.v2.position[2] };
        const f3 c = cross(v1 - v0, v2 - v0);
        const f3 n = c * (1.0f / sqrtf(dot(c, c)));
        const float D = dot(v0, n);
        const f3 E0 = cross(v2 - v0, n), E1 = cross(v0 - v1, n), E2 = cross(v1 - v2, n);
        float4* t = &tris[6 * static_cast<size_t>(i)];
        t[0] = make_float4(n.x, n.y, n.z, D);
        t[1] = make_float4(E0.x, E0.y, E0.z, v0.x);
        t[2] = make_float4(v0.y, v0.z, E1.x, E1.y);
        t[3] = make_float4(E1.z, v1.x, v1.y, v1.z);
        t[4] = make_float4(E2.x, E2.y, E2.z, v2.x);
        t[5] = make_float4(v2.y, v2.z, 0.0f, 0.0f);
        float4* a = &attrs[4 * static_cast<size_t>(i)];
        float meshBits;
        std::memcpy(&meshBits, &q.meshID, 4);
        a[0] = make_float4(q.v0.normal[0], q.v0.normal[1], q.v0.normal[2], q.v1.normal[0]);
        a[1] = make_float4(q.v1.normal[1], q.v1.normal[2], q.v2.normal[0], q.v2.normal[1]);
        a[2] = make_float4(q.v2.normal[2], q.v0.texCoord[0], q.v0.texCoord[1], q.v1.texCoord[0]);
        a[3] = make_float4(q.v1.texCoord[1], q.v2.texCoord[0], q.v2.texCoord[1], meshBits);
    }

    std::vector<DevMaterial> mats(n_meshes + n_spheres);
    auto toDev = [](const rt_material& m) {
        DevMaterial d {};
        std::memcpy(d.kd, m.kd, 12);
        std::memcpy(d.ks, m.ks, 12);
        d.shininess = m.shininess;
        d.transparency = m.transparency;
        d.tex = m.kdTexture;
        return d;
    };
    for (uint32_t i = 0; i < n_meshes; i++)
        mats[i] = toDev(mesh_materials[i]);
    std::vector<DevSphere> sph(n_spheres);
    for (uint32_t i = 0; i < n_spheres; i++) {
        if (spheres[i].material.kdTexture >= (int32_t)n_images)
            return fail(ctx, RT_ERR_INVALID, "rt_scene_upload: sphere %u names image %d of %u", i, spheres[i].material.kdTexture, n_images);
        mats[n_meshes + i] = toDev(spheres[i].material);
        std::memcpy(sph[i].center, spheres[i].center, 12);
        sph[i].radius = spheres[i].radius;
        sph[i].material = static_cast<int32_t>(n_meshes + i);
    }
    std::vector<DevTex> texs(n_images);
    size_t texFloats = 0;
    for (uint32_t i = 0; i < n_images; i++) {
        texs[i].offset = static_cast<uint32_t>(texFloats);
        texs[i].width = images[i].width;
        texs[i].height = images[i].height;
        texFloats += static_cast<size_t>(images[i].width) * images[i].height * 3;
        if (texFloats > 0xffffffffull)
            return fail(ctx, RT_ERR_UNSUPPORTED, "rt_scene_upload: textures exceed 2^32 floats");
    }
    std::vector<float> texels(texFloats);
    for (uint32_t i = 0; i < n_images; i++)
        std::memcpy(texels.data() + texs[i].offset, images[i].pixels, static_cast<size_t>(images[i].width) * images[i].height * 12);

    ctx->anyMirror = ctx->anyClear = ctx->anyMirrorAndClear = false;
    for (const DevMaterial& m : mats) {
        const bool mirror = !(m.ks[0] == 0.0f && m.ks[1] == 0.0f && m.ks[2] == 0.0f), clear = m.transparency != 1.0f;
        ctx->anyMirror |= mirror;
        ctx->anyClear |= clear;
        ctx->anyMirrorAndClear |= mirror && clear;
    }

    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, hipDeviceSynchronize());
    // On the device an inner child reference is the BYTE offset of the child's record (index * 32), so that a
    // traversal step forms no address: the loads take it as their 32-bit offset from the array's base.
    if (pairs.size() / 2 > (size_t(1) << 26))
        return fail(ctx, RT_ERR_UNSUPPORTED, "rt_scene_upload: %zu sibling-pair records exceed the %u supported", pairs.size() / 2, 1u << 26);
    for (size_t k = 1; k < pairs.size(); k += 2) {
        if ((pairs[k].z & kLeafFlag) == 0)
            pairs[k].z *= kPairBytes;
        if ((pairs[k].w & kLeafFlag) == 0)
            pairs[k].w *= kPairBytes;
    }
    if ((rootRef & kLeafFlag) == 0)
        rootRef *= kPairBytes;
    HIP_TRY(ctx, uploadVec(&ctx->dPairs, pairs));
    ctx->numPairs = static_cast<uint32_t>(pairs.size() / 2);
    // The caller's nodes, verbatim: a ray that meets two primitives at exactly the same distance is
    // replayed over them in the reference's order of tests, which decides the one it keeps.  The
    // replay's stack holds at most one entry per level of the caller's tree.
    static_assert(sizeof(rt_node) == 2 * sizeof(uint4), "rt_node is two 16-byte words");
    const bool replayable = callerLevels <= static_cast<uint32_t>(kMaxStack);
    {
        std::vector<uint4> ref(replayable ? static_cast<size_t>(n_nodes) * 2 : 0);
        if (replayable)
            std::memcpy(ref.data(), nodes, static_cast<size_t>(n_nodes) * sizeof(rt_node));
        HIP_TRY(ctx, uploadVec(&ctx->dRefNodes, ref));
    }
    if (replayable)
        maxLevel = std::max(maxLevel, callerLevels);
    HIP_TRY(ctx, uploadVec(&ctx->dLeafBoxes, leafBoxes));
    HIP_TRY(ctx, uploadVec(&ctx->dTris, tris));
    HIP_TRY(ctx, uploadVec(&ctx->dAttrs, attrs));
    HIP_TRY(ctx, uploadVec(&ctx->dMats, mats));
    HIP_TRY(ctx, uploadVec(&ctx->dTexs, texs));
    HIP_TRY(ctx, uploadVec(&ctx->dTexels, texels));
    HIP_TRY(ctx, uploadVec(&ctx->dSpheres, sph));

    DevScene& s = ctx->scene;
    const int32_t keepLights = s.nLights;
    const rt_light* keepLightPtr = s.lights;
    s = DevScene {};
    s.pairs = ctx->dPairs;
    s.tris = ctx->dTris;
    s.attrs = ctx->dAttrs;
    s.mats = ctx->dMats;
    s.texs = ctx->dTexs;
    s.texels = ctx->dTexels;
    s.spheres = ctx->dSpheres;
    s.lights = keepLightPtr;
    s.nLights = keepLights;
    const rt_node& root = nodes[0];
    s.rootRef = rootRef;
    std::memcpy(s.rootLo, root.lower, 12);
    std::memcpy(s.rootHi, root.upper, 12);
    std::memcpy(s.gridMin, gridMin, 12);
    std::memcpy(s.gridCell, gridCell, 12);
    s.nPrims = n_prims;
    s.tieByIndex = (ownTree && ctx->treeMode == 2) ? 1 : 0;
    s.refNodes = replayable ? ctx->dRefNodes : nullptr;
    s.leafBoxes = ctx->referenceBoxes ? ctx->dLeafBoxes : nullptr;
    s.strictIntersect = ctx->strictIntersect ? 1 : 0;
    s.nMeshes = static_cast<int32_t>(n_meshes);
    s.nSpheres = static_cast<int32_t>(n_spheres);
    s.nTex = static_cast<int32_t>(n_images);
    for (int i = 0; i < 6; i++)
        s.env[i] = env_faces ? env_faces[i] : -1;
    ctx->numLevels = maxLevel;
    ctx->haveScene = true;
    ctx->generation++;
    if (generation)
        *generation = ctx->generation;
    return warmOccupancy(ctx);
} catch (...) {
    return caught(ctx, "rt_scene_upload");
}

} // extern "C"
