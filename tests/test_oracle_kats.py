"""The CPU oracle against every known-answer value the reference's own tests
hold for this path, plus the two whole-frame figures recorded from the
reference build when the reference was surveyed.  No GPU involved."""
import itertools

import numpy as np
import pytest

import common
import orc
from common import sio


def close(a, b, eps):
    return np.all(np.abs(np.asarray(a, dtype=np.float64) - np.asarray(b, dtype=np.float64)) <= eps)


# ---- tests/interpolation_tests.cpp:24-61 (eps 1e-6)
def test_barycentric_cases():
    assert close(orc.barycentric((3, 4, 5), (0, 5, 3), (3, 0, -2), (2, 3, 2)), (1 / 3, 1 / 3, 1 / 3), 1e-6)
    assert close(orc.barycentric((0, 0, 0), (1, 0, 0), (0.4, 1, 0), (0.6, 0, 0)), (0.4, 0.6, 0.0), 1e-6)
    assert close(orc.barycentric((0, 0, 0), (1, 0, 0), (0.4, 1, 0), (0.4, 1, 0)), (0, 0, 1), 1e-6)


def test_interpolate_normals():
    b = (1 / 3, 1 / 3, 1 / 3)
    assert close(orc.interpolate_normal((1, 0, 0), (0, 1, 0), (0, 0, 1), b), b, 1e-6)


# ---- tests/texture_tests.cpp:46-152 (eps 1e-6)
def img(n):
    v = np.arange(1, n * n + 1, dtype=np.float32)
    return np.repeat(v[:, None], 3, axis=1).reshape(n, n, 3)


NEAREST = [(3, (0, 0), 7), (3, (1, 0), 9), (3, (1, 1), 3), (3, (0, 1), 1), (3, (0.8, 0.5), 6), (5, (0.5, 0.5), 13),
           (5, (0.3, 0.7), 7)]
BILINEAR = [((0, 0), 21), ((0, 1), 1), ((1, 0), 25), ((1, 1), 5), ((0.2, 0.2), 19), ((0.1, 0.1), 21), ((0.5, 0.0), 23),
            ((0.25, 0.15), 20.5), ((0.65, 0.45), 15), ((0.9, 0.5), 15), ((0.01, 0.5), 11)]


@pytest.mark.parametrize("n,uv,expect", NEAREST)
def test_texture_nearest(n, uv, expect):
    assert close(orc.sample_texture(img(n), uv, False), (expect,) * 3, 1e-6)


@pytest.mark.parametrize("uv,expect", BILINEAR)
def test_texture_bilinear(uv, expect):
    assert close(orc.sample_texture(img(5), uv, True), (expect,) * 3, 1e-6)


# ---- tests/light_tests.cpp:24-156
def test_segment_light():
    l = sio.segment_light((0, 0, 0), (1, 1, 1), (0.2,) * 3, (0.8,) * 3)
    for s, p, c in [(0.0, 0.0, 0.2), (0.3, 0.3, 0.38), (0.5, 0.5, 0.5), (1.0, 1.0, 0.8)]:
        pos, col = orc.sample_segment_light(s, l)
        assert close(pos, (p,) * 3, 1e-4) and close(col, (c,) * 3, 1e-4)


def test_parallelogram_light_exact():
    l = sio.parallelogram_light((0, 0, 0), (1, 0, 0), (0, 1, 0), (1, 0, 0), (0, 1, 0), (0, 0, 1), (1, 1, 1))
    f = np.float32

    def mix(x, y, a):  # glm::mix
        return (np.asarray(x, f) * (f(1) - f(a)) + np.asarray(y, f) * f(a)).astype(f)
    cases = {(0.0, 0.0): ((0, 0, 0), (1, 0, 0)), (0.5, 0.5): ((0.5, 0.5, 0), (0.5, 0.5, 0.5)), (1.0, 1.0): ((1, 1, 0), (1, 1, 1)),
             (0.3, 0.7): ((f(0.3), f(0.7), 0), mix(mix((1, 0, 0), (0, 1, 0), 0.3), mix((0, 0, 1), (1, 1, 1), 0.3), 0.7))}
    for s, (p, c) in cases.items():
        pos, col = orc.sample_parallelogram_light(s, l)
        assert np.array_equal(pos, np.asarray(p, f))  # the reference test uses ==
        assert np.array_equal(col, np.asarray(c, f))


def test_parallelogram_weights():
    """calculateBarycentricCoordinates via the colour it produces (tests/light_tests.cpp:56-105)."""
    l = sio.parallelogram_light((0, 0, 0), (1, 0, 0), (0, 1, 0), (1, 0, 0), (0, 1, 0), (0, 0, 1), (0, 0, 0))
    _, c = orc.sample_parallelogram_light((0.5, 0.5), l)
    assert close(c, (0.25, 0.25, 0.25), 1e-6)


# ---- tests/depth_of_field_tests.cpp:30-82 (eps 1e-6)
def dof_camera():
    c = np.zeros((), dtype=sio.CAMERA)
    c["lookAt"], c["rotation"], c["distance"] = (1, 2, 5), (1.0, 1.5, 2.0), 3.0
    c["fovy"] = np.float32(50.0) * np.float32(0.01745329251994329576923690768489)
    c["aspect"] = 1.0
    return c


def test_camera_basis_and_focus_points():
    cam = dof_camera()
    pos, fwd, up, left = orc.camera_basis(cam)
    assert close(pos, (-0.622596637, -0.520720849, 4.885341580), 1e-6)
    assert close(fwd, (0.540865545, 0.840240283, 0.038219473), 1e-6)
    assert close(left, (-0.029437062, 0.064321155, -0.99749486), 1e-6)
    assert close(up, (-0.840593790, 0.538385601, 0.059523302), 1e-6)
    cases = [(fwd, (1.540865545, 2.840240283, 5.038219473)),
             ((0.707073733, 0.675492054, -0.209182263), (2.379799376, 2.347572165, 3.997106076)),
             ((0.007410816, 0.919139747, -0.393861909), (-0.583656555, 4.308892342, 2.815797010)),
             ((0.302661460, 0.791110773, 0.531544714), (0.803810963, 3.207690485, 7.390448834))]
    for d, expect in cases:
        r = np.zeros((), dtype=sio.RAY)
        r["origin"], r["direction"], r["t"] = pos, d, common.FLT_MAX
        assert close(orc.point_of_focus(cam, r, 4.0), expect, 1e-6)


# ---- tests/bvh_tests.cpp:676-754: nodes / primitives / leaves / levels
SHAPES = [("SingleTriangle", 2, 1, 1, 1), ("Cube", 8, 12, 4, 3), ("CornellBox", 16, 32, 8, 4), ("Monkey", 512, 967, 256, 9),
          ("Teapot", 8192, 15704, 4096, 13), ("Dragon", 65536, 87130, 32768, 16)]


@pytest.mark.parametrize("name,nodes,prims,leaves,levels", SHAPES)
def test_bvh_shape_table(name, nodes, prims, leaves, levels):
    n, p, nl, nf = orc.bvh_build(common.scene(name).primitives())
    assert (len(n), len(p), nf, nl) == (nodes, prims, leaves, levels)


# ---- tests/bvh_tests.cpp:817-1009
def test_pyramid_layout_order_and_t_values():
    scn, V = common.pyramid_scene()
    osc = orc.OracleScene(scn)
    assert (osc.num_leaves, osc.num_levels) == (2, 2)
    n = osc.nodes
    assert n[0]["lower"].tolist() == [-4, -2, -2] and n[0]["upper"].tolist() == [5, 4, 2]
    assert n[0]["data"].tolist() == [2, 3]
    assert n[2]["upper"].tolist() == [2, 4, 2] and n[2]["data"].tolist() == [sio.LEAF_BIT | 0, 4]
    assert n[3]["upper"].tolist() == [5, 4, 2] and n[3]["data"].tolist() == [sio.LEAF_BIT | 4, 4]
    expect = [(0, 0, 2, 3), (1, 4, 6, 7), (1, 4, 5, 6), (1, 5, 6, 7), (1, 4, 5, 7), (0, 0, 1, 2), (0, 1, 2, 3), (0, 0, 1, 3)]
    for prim, (m, a, b, c) in zip(osc.prims, expect):
        assert prim["meshID"] == m
        assert np.array_equal(prim["v0"]["position"], V[a]) and np.array_equal(prim["v1"]["position"], V[b])
        assert np.array_equal(prim["v2"]["position"], V[c])
    rays = np.zeros(5, dtype=sio.RAY)
    rays["direction"], rays["t"] = (1, 0, 0), common.FLT_MAX
    rays["origin"][:, 0] = [-10, -1, 0, 1, 10]
    want = np.array([9.199999809265, 0.200000092387, 0.3249999880790, 2.25], dtype=np.float32)
    for accel in (1, 0):
        for policy in (0, 1):
            r, h, _ = osc.intersect(common.features(enableAccelStructure=accel), rays, policy)
            assert h["hit"].tolist() == [1, 1, 1, 1, 0]
            assert np.array_equal(r["t"][:4], want)


# ---- tests/bvh_tests.cpp:756-815: BVH on == BVH off
@pytest.mark.parametrize("name,n", [("SingleTriangle", 5), ("Cube", 20), ("CornellBox", 100), ("Monkey", 300), ("Teapot", 300)])
def test_bvh_on_equals_off(name, n):
    osc = orc.OracleScene(common.scene(name))
    rays = common.random_rays(n)
    r1, h1, _ = osc.intersect(common.features(enableAccelStructure=1), rays)
    r0, h0, _ = osc.intersect(common.features(enableAccelStructure=0), rays)
    r2, h2, _ = osc.intersect(common.features(enableAccelStructure=1), rays, policy=1)
    for r, h in ((r0, h0), (r2, h2)):
        assert np.array_equal(h1["hit"], h["hit"])
        assert np.array_equal(r1["t"].view(np.uint32), r["t"].view(np.uint32))
        for k in ("normal", "barycentricCoord", "material"):
            assert np.array_equal(h1[k], h[k])


# ---- tests/bvh_tests.cpp:197-674, through the builder (single-leaf and two-leaf trees)
def _prims(tris):
    p = np.zeros(len(tris), dtype=sio.PRIMITIVE)
    for i, t in enumerate(tris):
        p[i]["v0"]["position"], p[i]["v1"]["position"], p[i]["v2"]["position"] = t
    return p


def test_aabb_of_primitive_all_vertex_orders():
    tri = [(-2.0, 2.0, -3.0), (4.0, -1.0, 0.5), (1.0, 5.0, 2.0)]
    for perm in itertools.permutations(tri):
        nodes, _, _, _ = orc.bvh_build(_prims([perm]))
        assert nodes[0]["lower"].tolist() == [-2.0, -1.0, -3.0] and nodes[0]["upper"].tolist() == [4.0, 5.0, 2.0]


def test_median_split_sizes_and_order():
    # 6 triangles spread along x: sorted by centroid, split at (n+1)/2 = 3
    tris = [[(x, 0, 0), (x + 0.5, 1, 0), (x + 1, 0, 1)] for x in (5.0, 1.0, 3.0, 0.0, 4.0, 2.0)]
    nodes, prims, levels, leaves = orc.bvh_build(_prims(tris))
    assert (levels, leaves) == (2, 2)
    assert [float(p["v0"]["position"][0]) for p in prims] == [0.0, 1.0, 2.0, 3.0, 4.0, 5.0]
    assert nodes[2]["data"].tolist() == [sio.LEAF_BIT | 0, 3] and nodes[3]["data"].tolist() == [sio.LEAF_BIT | 3, 3]


def test_longest_axis_ties_take_the_first():
    """src/bvh.cpp:189-195: x wins ties with y and z, y wins a tie with z."""
    px = [0, 7, 3, 5, 1, 6, 2, 4]
    py = [4, 0, 6, 2, 7, 1, 5, 3]
    pz = [2, 5, 0, 7, 3, 4, 6, 1]

    def split_axis(ext):
        pts = [(px[k] * ext[0] / 7, py[k] * ext[1] / 7, pz[k] * ext[2] / 7) for k in range(8)]
        _, prims, _, _ = orc.bvh_build(_prims([[q, q, q] for q in pts]))
        c = np.array([p["v0"]["position"] for p in prims])
        return [axis for axis in range(3) if np.all(np.diff(c[:, axis]) > 0)]
    assert split_axis((2.0, 2.0, 2.0)) == [0]
    assert split_axis((1.0, 2.0, 2.0)) == [1]
    assert split_axis((1.0, 1.0, 2.0)) == [2]
    assert split_axis((2.0, 1.0, 2.0)) == [0]


# ---- sampler (src/sampler.h:16-45): an independent numpy restatement
def test_sampler_stream():
    def ref(seed, n):
        out, s = [], np.uint32(seed)
        with np.errstate(over="ignore"):
            for _ in range(n):
                s = np.uint32(s * np.uint32(747796405) + np.uint32(2891336453))
                v = np.uint32(s)
                v ^= v >> np.uint32((v >> np.uint32(28)) + np.uint32(4))
                v = np.uint32(v * np.uint32(277803737))
                v ^= v >> np.uint32(22)
                out.append(np.float32(v) / np.float32(4294967295.0))
        return np.array(out, dtype=np.float32)
    for seed in (0, 123, 1080 * 17 + 5, 2**32 - 1):
        assert np.array_equal(orc.sampler_next(seed, 16), ref(seed, 16))


# ---- whole frames recorded from the reference build itself
def test_cornell_pixel_sum_of_the_reference_build():
    """Recorded from a build of the reference: Cornell box 128x128, shading (Blinn-Phong) + shadows +
    reflections + BVH, 1 spp, default camera: sum of all pixels = 2570.015468."""
    osc = orc.OracleScene(common.scene("CornellBox"))
    f = common.features(enableShading=1, enableReflections=1, enableShadows=1, shadingModel=2)
    img, st = osc.render(f, sio.default_camera(128, 128), 128, 128)
    assert abs(float(img.astype(np.float64).sum()) - 2570.015468) < 5e-6


def test_teapot_ray_count_of_the_reference_build():
    """Recorded from a build of the reference: teapot 960x540, same features: 596 998 intersect calls."""
    osc = orc.OracleScene(common.scene("Teapot"))
    f = common.features(enableShading=1, enableReflections=1, enableShadows=1, shadingModel=2)
    _, st = osc.render(f, sio.default_camera(960, 540), 960, 540)
    assert st["rays"] == 596998


def test_culling_policy_changes_no_pixel():
    """The device's traversal policy (skip children entered beyond the current hit)
    against the reference's: same image, same ray count, fewer node visits."""
    osc = orc.OracleScene(common.scene("Monkey"))
    f = common.features(enableShading=1, enableReflections=1, enableShadows=1, shadingModel=2, enableNormalInterp=1)
    cam = sio.default_camera(160, 90)
    a, sa = osc.render(f, cam, 160, 90, policy=0)
    b, sb = osc.render(f, cam, 160, 90, policy=1)
    assert np.array_equal(a.view(np.uint32), b.view(np.uint32))
    assert sa["rays"] == sb["rays"] and sb["inner_nodes"] <= sa["inner_nodes"]


# ---------------------------------------------------------------------------
# post-processing (src/extra.cpp:214-316, src/screen.cpp:44-55): no test or output of the
# reference covers these, so the restatement is checked against an independent numpy form of
# the same statements (parity with the reference itself: unpinned)
# ---------------------------------------------------------------------------
def _numpy_bloom(frame, k, threshold):
    from math import factorial
    h, w = frame.shape[:2]
    flat = frame.reshape(-1, 3).astype(np.float32)
    lum = (np.float32(0.2126) * flat[:, 0] + np.float32(0.7152) * flat[:, 1]) + np.float32(0.0722) * flat[:, 2]
    t = flat.copy()
    t[lum < np.float32(threshold)] = 0
    vals = [np.float32(factorial(k) // (factorial(y) * factorial(k - y))) for y in range(k)]
    total = np.float32(0)
    for v in vals:
        total = np.float32(total + v)
    wgt = [np.float32(v / total) for v in vals]
    out = t.copy()
    xs = np.arange(k - 2, w - k + 2)
    ys = np.arange(k - 2, h - k + 2)
    if len(xs) and len(ys):
        X, Y = np.meshgrid(xs, ys, indexing="ij")
        acc = np.zeros(X.shape + (3,), dtype=np.float32)
        for i in range(k):
            for j in range(k):
                idx = (h - 1 - (Y + j)).astype(np.int64) * w + (X + i)
                ok = (idx >= 0) & (idx < w * h)
                px = np.where(ok[..., None], t[np.clip(idx, 0, w * h - 1)], np.float32(0))
                acc = (acc + wgt[j] * px).astype(np.float32)
        out[(h - 1 - Y) * w + X] = acc
    return (flat + out).astype(np.float32).reshape(h, w, 3)


@pytest.mark.parametrize("k,threshold,w,h", [(3, 0.5, 23, 17), (4, 0.3, 16, 9), (7, 0.0, 31, 20), (20, 0.4, 45, 41), (5, 0.5, 6, 5)])
def test_bloom_restatement_against_numpy_form(k, threshold, w, h):
    rng = np.random.default_rng(100 + k)
    frame = (rng.random((h, w, 3), dtype=np.float32) * np.float32(1.4)).astype(np.float32)
    f = sio.default_features(enableBloomEffect=1, bloomFilterSize=k, bloomThreshold=threshold)
    got = orc.bloom(f, frame)
    want = _numpy_bloom(frame, k, threshold)
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))
    off = sio.default_features(enableBloomEffect=0)
    assert np.array_equal(orc.bloom(off, frame), frame)
    for bad in (1, 2, 21):
        with pytest.raises(RuntimeError):
            orc.bloom(sio.default_features(enableBloomEffect=1, bloomFilterSize=bad), frame)


def test_bloom_filter_columns_sum_to_one_and_weights_are_binomial():
    # computeGaussianFilter: filter[x][y] = C(k, y) / sum_{y<k} C(k, y); for k = 3: 1/7, 3/7, 3/7.
    # A frame that is white everywhere, threshold 0: an inner pixel becomes 1 + k * (column sum) = 1 + k.
    k, w, h = 3, 12, 12
    frame = np.ones((h, w, 3), dtype=np.float32)
    out = orc.bloom(sio.default_features(enableBloomEffect=1, bloomFilterSize=k, bloomThreshold=0.0), frame)
    assert abs(float(out[h // 2, w // 2, 0]) - (1.0 + k)) < 1e-5
    assert float(out[0, 0, 0]) == 2.0  # outside the filtered range: the frame plus its thresholded self
    one = np.zeros((h, w, 3), dtype=np.float32)
    one[h - 1 - 6, 6] = 1.0  # pixel (6, 6)
    o = orc.bloom(sio.default_features(enableBloomEffect=1, bloomFilterSize=k, bloomThreshold=0.5), one)
    col = [float(o[h - 1 - (6 - j), 6, 0]) for j in range(k)]  # pixels (6, 6 - j) see it through filter[0][j]
    assert np.allclose(col, [1.0 + 1.0 / 7.0, 3.0 / 7.0, 3.0 / 7.0], atol=1e-6)


def test_bmp_bytes_layout():
    import struct
    w, h = 5, 3
    frame = np.zeros((h, w, 3), dtype=np.float32)
    frame[0, 1] = (1.0, 0.5, 0.25)   # top row, second pixel
    frame[2, 4] = (2.0, -1.0, 0.999)  # bottom row, clamped
    b = orc.bmp_bytes(frame)
    assert b[:2] == b"BM" and len(b) == 122 + w * h * 4
    size, _, off = struct.unpack_from("<IIi", b, 2)
    assert size == len(b) and off == 122
    hdr, bw, bh, planes, bpp, comp = struct.unpack_from("<IiiHHI", b, 14)
    assert (hdr, bw, bh, planes, bpp, comp) == (108, w, h, 1, 32, 3)
    assert struct.unpack_from("<IIII", b, 54) == (0xff0000, 0xff00, 0xff, 0xff000000)
    px = np.frombuffer(b, dtype=np.uint8, offset=122).reshape(h, w, 4)  # file rows: bottom-up
    assert tuple(px[h - 1, 1]) == (63, 127, 255, 255)  # B G R A of the top row's pixel, truncated
    assert tuple(px[0, 4]) == (254, 0, 255, 255)
