"""The oracle's restatement of renderImageWithMotionBlur (src/extra.cpp:121-211) against its own plain
renderer.  The reference holds no motion-blur output (and leaves the pixel sum uninitialised), so this
is what pins the restatement: with `movement` zero the moved copy of the scene is the scene at twice
its size -- the five control points of splineMat (src/extra.cpp:553-557) are then the position itself,
the Bezier point is the position to a rounding, and the translation by it doubles the position -- and
with point lights and one ray per pixel no sampler draw reaches the frame, so the one time sample of
every pixel must give the plain render of the doubled scene."""
import copy

import numpy as np

import common
from common import sio
import orc


def test_zero_movement_is_the_scene_at_twice_its_size():
    scn = common.scene("CornellBox")
    osc = orc.OracleScene(scn)
    w, h = 40, 30
    cam = sio.default_camera(w, h)
    kw = dict(enableShading=1, enableShadows=1, enableReflections=1, shadingModel=2)
    f = common.features(enableMotionBlur=1, motionBlurSamples=1, movement=0.0, **kw)
    blurred, st = osc.render_motion_blur(scn.primitives(), f, cam, w, h)
    doubled = copy.deepcopy(scn)
    for m in doubled.meshes:
        m.vertices["position"] *= np.float32(2)
    plain, st2 = orc.OracleScene(doubled).render(common.features(**kw), cam, w, h)
    assert st["rays"] == st2["rays"] and plain.any()
    assert common.image_error(blurred, plain)[0] <= 1e-5
    # several samples of the same still scene average to the same frame
    f["motionBlurSamples"] = 3
    again, st3 = osc.render_motion_blur(scn.primitives(), f, cam, w, h)
    assert st3["rays"] == 3 * st["rays"]
    assert common.image_error(again, plain)[0] <= 1e-5


def test_moved_scene_follows_the_curve():
    """At time t the copy is 2 * position + movement * (the Bezier point of the five offsets): a frame of a
    single far-away triangle, moved, equals the plain frame of that triangle put there by hand."""
    s = sio.Scene()
    va = np.zeros(3, dtype=sio.VERTEX)
    va["position"] = np.array([[-0.5, -0.5, 0.0], [0.5, -0.5, 0.0], [0.0, 0.5, 0.0]], dtype=np.float32)
    va["normal"] = (0, 0, -1)
    s.meshes.append(sio.Mesh(va, np.array([[0, 1, 2]], dtype=np.uint32), sio.make_material((0.8, 0.3, 0.2))))
    s.lights = [sio.point_light((0, 0, -3), (1, 1, 1))]
    w, h = 24, 24
    cam = sio.default_camera(w, h, distance=6.0)
    movement = np.float32(0.75)
    f = common.features(enableMotionBlur=1, motionBlurSamples=1, movement=float(movement), enableShading=1)
    blurred, _ = orc.OracleScene(s).render_motion_blur(s.primitives(), f, cam, w, h)
    # per pixel: the time is the first draw of Sampler(height * x + y)
    offsets = np.array([[0, 0, 0], [0, 1, 1], [1, 1, -1], [1, 0, 0], [1.5, 1, 2]], dtype=np.float64)
    checked = 0
    for (x, y) in [(3, 5), (12, 12), (20, 7), (8, 18)]:
        t = float(orc.sampler_next(h * x + y, 1)[0])
        wts = np.array([(1 - t) ** 4, 4 * (1 - t) ** 3 * t, 6 * (1 - t) ** 2 * t * t, 4 * (1 - t) * t ** 3, t ** 4])
        shift = (wts[:, None] * offsets).sum(0) * float(movement)
        there = copy.deepcopy(s)
        there.meshes[0].vertices["position"] = (2.0 * s.meshes[0].vertices["position"].astype(np.float64) + shift).astype(np.float32)
        plain, _ = orc.OracleScene(there).render(common.features(enableShading=1), cam, w, h)
        assert abs(float(blurred[h - 1 - y, x, 0]) - float(plain[h - 1 - y, x, 0])) <= 1e-4
        checked += plain[h - 1 - y, x].any()
    assert checked >= 1  # at least one of the pixels looked at the triangle
