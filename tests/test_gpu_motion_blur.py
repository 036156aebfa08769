"""GPU parity of renderImageWithMotionBlur (src/extra.cpp:121-211, splineMat :546-567): the device's
motion-blur variant of the per-pixel kernel against the oracle, which does what the reference does --
a moved copy of the scene and a tree built over it for every time sample of every pixel.

Parity unpinned against the reference itself: it holds no motion-blur output, and its own result is
undefined (the sum over the time samples is never initialised, src/extra.cpp:144); device and oracle
both start it at zero.  The oracle's restatement is checked on the CPU against its plain renderer
(tests/test_motion_blur_cpu.py)."""
import importlib

import numpy as np
import pytest

import common
from common import sio

pytestmark = pytest.mark.gpu

capi = importlib.import_module("ray-tracer_amd.capi")
import orc  # noqa: E402


@pytest.fixture(scope="module")
def ctx():
    c = capi.Context(0)
    yield c
    c.close()


def _compare(ctx, scn, f, cam, w, h):
    osc = orc.OracleScene(scn)
    ctx.upload_scene(scn)
    ctx.set_features(f)
    want, so = osc.render_motion_blur(scn.primitives(), f, cam, w, h)
    got, sg = ctx.render(cam, w, h)
    assert sg["rays"] == so["rays"]
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), common.image_error(got, want)
    return got


CASES = {
    # scene, frame, camera distance, features
    "cornell_mirrors": ("CornellBox", (40, 30), 3.0, dict(enableShading=1, enableShadows=1, enableReflections=1, shadingModel=2,
                                                          motionBlurSamples=3, movement=0.5)),
    "cornell_area_light_4spp": ("CornellBoxParallelogramLight", (24, 18), 3.0,
                                dict(enableShading=1, enableShadows=1, shadingModel=1, numPixelSamples=4, numShadowSamples=3,
                                     motionBlurSamples=2, movement=1.3)),
    "cube_textured_jittered": ("CubeTextured", (32, 32), 4.0,
                               dict(enableShading=1, enableShadows=1, enableNormalInterp=1, enableTextureMapping=1,
                                    enableBilinearTextureFiltering=1, enableJitteredSampling=1, numPixelSamples=4,
                                    motionBlurSamples=2, movement=0.25)),
    "monkey_glossy": ("Monkey", (24, 24), 5.0, dict(enableShading=1, enableShadows=1, enableReflections=1, enableGlossyReflection=1,
                                                    numGlossySamples=3, glossyExponent=0.5, enableNormalInterp=1, shadingModel=2,
                                                    motionBlurSamples=2, movement=0.4)),
    "cornell_no_accel": ("CornellBox", (20, 16), 3.0, dict(enableShading=1, enableShadows=1, enableAccelStructure=0,
                                                          motionBlurSamples=2, movement=0.8)),
    "teapot_sah_binned_flag": ("Teapot", (16, 12), 5.0, dict(enableShading=1, enableShadows=1, enableBvhSahBinning=1, shadingModel=3,
                                                           motionBlurSamples=1, movement=0.3)),
}


@pytest.mark.parametrize("key", list(CASES))
def test_motion_blur_matches_oracle(ctx, key):
    name, (w, h), distance, kw = CASES[key]
    scn = common.scene(name)
    f = common.features(enableMotionBlur=1, **kw)
    got = _compare(ctx, scn, f, sio.default_camera(w, h, distance=distance), w, h)
    assert got.any()
    # the effect did something: the same frame without it differs
    plain = f.copy()
    plain["enableMotionBlur"] = 0
    ctx.set_features(plain)
    still, _ = ctx.render(sio.default_camera(w, h, distance=distance), w, h)
    assert not np.array_equal(got, still)


def test_motion_blur_with_transparency_and_spheres(ctx):
    """Pass-through rays and light samples through transparent surfaces in the moved copy; spheres move with their centres."""
    import copy
    scn = copy.deepcopy(common.scene("CornellBox"))
    for i, m in enumerate(scn.meshes):
        m.material["transparency"] = 1.0 if i == 0 else 0.6  # one mesh stays opaque
    f = common.features(enableMotionBlur=1, motionBlurSamples=2, movement=0.6, enableShading=1, enableShadows=1, enableTransparency=1,
                        enableReflections=1, shadingModel=2, maxRayDepth=4)
    _compare(ctx, scn, f, sio.default_camera(28, 20), 28, 20)
    sp = sio.load_scene_prebuilt("Spheres", common.DATA)
    f = common.features(enableMotionBlur=1, motionBlurSamples=3, movement=0.5, enableShading=1, enableShadows=1, shadingModel=1)
    cam = sio.default_camera(32, 32, distance=6.0, look_at=(0, 0, 12), rotation_deg=(0, 180, 0))
    got = _compare(ctx, sp, f, cam, 32, 32)
    assert got.any()


def test_motion_blur_corner_cases(ctx):
    scn = common.scene("CornellBox")
    w, h = 16, 12
    cam = sio.default_camera(w, h)
    # depth of field wins when both are asked for (src/render.cpp:22-25)
    both = common.features(enableMotionBlur=1, motionBlurSamples=2, movement=0.5, enableShading=1, enableDepthOfField=1,
                           depthOfFieldDistance=3.0, depthOfFieldCircleDiameter=0.1, numDepthOfFieldSamples=3)
    dof = both.copy()
    dof["enableMotionBlur"] = 0
    ctx.upload_scene(scn)
    ctx.set_features(both)
    a, sa = ctx.render(cam, w, h)
    ctx.set_features(dof)
    b, sb = ctx.render(cam, w, h)
    assert np.array_equal(a.view(np.uint32), b.view(np.uint32)) and sa["rays"] == sb["rays"]
    # no time sample at all: zero divided by zero in every pixel, as in the oracle; no ray is traced
    none = common.features(enableMotionBlur=1, motionBlurSamples=0, movement=0.5, enableShading=1)
    osc = orc.OracleScene(scn)
    want, so = osc.render_motion_blur(scn.primitives(), none, cam, w, h)
    ctx.set_features(none)
    got, sg = ctx.render(cam, w, h)
    assert np.isnan(want).all() and np.isnan(got).all() and sg["rays"] == so["rays"] == 0
    # movement zero still doubles the scene (the control points are offsets from the position itself)
    zero = common.features(enableMotionBlur=1, motionBlurSamples=1, movement=0.0, enableShading=1, enableShadows=1)
    _compare(ctx, scn, zero, cam, w, h)
    # sharded like any other frame
    import torch
    f = common.features(enableMotionBlur=1, motionBlurSamples=2, movement=0.5, enableShading=1, enableShadows=1, enableBloomEffect=0)
    w, h = 37, 21
    cam = sio.default_camera(w, h)
    whole = _compare(ctx, scn, f, cam, w, h)
    world = 3
    stride = capi.shard_floats(w, h, 0, world)
    packed = torch.zeros((world, stride), dtype=torch.float32, device="cuda")
    for r in range(world):
        ctx.render_device(cam, w, h, packed[r].data_ptr(), stride, r, world, 0)
    frame = torch.empty(h * w * 3, dtype=torch.float32, device="cuda")
    ctx.deinterleave(packed.data_ptr(), w, h, world, frame.data_ptr(), 0)
    ctx.sync(0)
    assert np.array_equal(frame.cpu().numpy().reshape(h, w, 3).view(np.uint32), whole.view(np.uint32))


def test_motion_blur_on_the_queue_based_renderer():
    """Motion blur runs on the queue-based renderer: every (pixel, time sample) pair takes a pixel's place in the
    queues, its rays carry the pair's time and meet the scene moved to it, and the combine takes the mean over a
    pixel's pairs.  Bit for bit against the per-pixel kernel (which the tests above hold against the oracle, as they
    do the default renderer): one camera ray per sample and several, mirror chains, transparency, glossy reflections,
    all three kinds of light, an environment map, spheres, pixel batches (a small workspace), the counting kernels,
    shards, and the entry point that allocates nothing."""
    import copy
    import torch
    scn = copy.deepcopy(common.scene("CornellBox", env_map=True))
    scn.lights = [sio.cornell_parallelogram_light(), sio.point_light((0.2, 0.3, 0.4), (0.6, 0.5, 0.4)),
                  sio.segment_light((-0.3, 0.4, -0.2), (0.3, 0.4, 0.2), (1, 0, 0), (0, 0, 1))]
    spheres = sio.load_scene_prebuilt("Spheres", common.DATA)
    clear = copy.deepcopy(scn)
    for i, m in enumerate(clear.meshes):
        m.material["transparency"] = 1.0 if i == 0 else 0.55  # (one mesh stays opaque)
    w, h = 45, 27  # (partial tiles on both edges)
    cases = [
        (scn, sio.default_camera(w, h), dict(motionBlurSamples=1, movement=0.5, numShadowSamples=1)),
        (scn, sio.default_camera(w, h), dict(motionBlurSamples=3, movement=0.7, numShadowSamples=2, enableReflections=1, maxRayDepth=3,
                                             enableEnvironmentMap=1)),
        (scn, sio.default_camera(w, h), dict(motionBlurSamples=5, movement=0.3, numShadowSamples=3, numPixelSamples=4,
                                             enableJitteredSampling=1, enableReflections=1, maxRayDepth=2)),
        (scn, sio.default_camera(w, h), dict(motionBlurSamples=2, movement=1.1, numShadowSamples=2, numPixelSamples=3,
                                             enableShadows=0)),
        (spheres, sio.default_camera(w, h, distance=6.0, look_at=(0, 0, 12), rotation_deg=(0, 180, 0)),
         dict(motionBlurSamples=4, movement=0.5, numShadowSamples=1, enableReflections=1)),
        # pass-through rays and light tinted on its way through surfaces, in the moved copy
        (clear, sio.default_camera(w, h), dict(motionBlurSamples=3, movement=0.6, numShadowSamples=2, enableTransparency=1,
                                               enableReflections=1, maxRayDepth=3, numPixelSamples=2)),
        (clear, sio.default_camera(w, h), dict(motionBlurSamples=2, movement=0.4, numShadowSamples=1, enableTransparency=1,
                                               enableEnvironmentMap=1, maxRayDepth=4)),
        # glossy reflections: the tree kernel walks every (pixel, time sample) pair
        (scn, sio.default_camera(w, h), dict(motionBlurSamples=3, movement=0.5, numShadowSamples=2, enableReflections=1, maxRayDepth=2,
                                             enableGlossyReflection=1, numGlossySamples=3, glossyExponent=0.5, numPixelSamples=2,
                                             enableEnvironmentMap=1)),
        (scn, sio.default_camera(w, h), dict(motionBlurSamples=2, movement=0.9, numShadowSamples=1, enableReflections=1, maxRayDepth=3,
                                             enableGlossyReflection=1, numGlossySamples=2, glossyExponent=1.0)),
        # ... and all of it at once
        (clear, sio.default_camera(w, h), dict(motionBlurSamples=2, movement=0.5, numShadowSamples=2, enableReflections=1, maxRayDepth=2,
                                               enableGlossyReflection=1, numGlossySamples=2, glossyExponent=0.5, enableTransparency=1,
                                               enableEnvironmentMap=1)),
    ]
    batched = {False: [], True: []}  # pixel batches of the small-workspace frames, without and with transparency
    for k, (s, cam, kw) in enumerate(cases):
        kw = dict(dict(enableShading=1, enableShadows=1, shadingModel=2, enableNormalInterp=1), **kw)
        f = common.features(enableMotionBlur=1, **kw)
        images, rays = {}, {}
        # (a workspace that makes several pixel batches of the frame; a tree of hits under transparency takes more per tile)
        small = (8 << 20) if kw.get("enableTransparency") else (2 << 20)
        for name, opts in (("queues", dict(per_pixel=0)), ("queues, small workspace", dict(per_pixel=0, workspace_bytes=small)),
                           ("per pixel", dict(per_pixel=1))):
            c = capi.Context(0, **opts)
            c.upload_scene(s)
            c.set_features(f)
            images[name], st = c.render(cam, w, h)
            rays[name] = st["rays"]
            if name == "queues, small workspace":
                batched[bool(kw.get("enableTransparency"))].append(st["batches"])
            if name == "queues":  # the counting kernels, packed shards, and the static scratch
                buf = torch.zeros(h * w * 3, dtype=torch.float32, device="cuda")
                st2 = c.render_device(cam, w, h, buf.data_ptr(), buf.numel(), 0, 1, 0, collect_stats=True)
                assert st2["rays"] == st["rays"], (k, st2["rays"], st["rays"])
                images["queues, counting"] = buf.cpu().numpy().reshape(h, w, 3)
                buf.zero_()
                c.render_device_noalloc(cam, w, h, buf.data_ptr(), buf.numel(), 0, 1, 0)
                c.sync(0)
                images["queues, no allocation"] = buf.cpu().numpy().reshape(h, w, 3)
            c.close()
        assert images["per pixel"].any(), k
        assert rays["queues"] == rays["per pixel"] == rays["queues, small workspace"], (k, rays)
        for name in images:
            assert np.array_equal(images[name].view(np.uint32), images["per pixel"].view(np.uint32)), \
                (k, name, common.image_error(images[name], images["per pixel"]))
    assert max(batched[False]) > 1 and max(batched[True]) > 1, batched
    # shards of the frame, packed
    s, cam, kw = cases[2]
    f = common.features(enableMotionBlur=1, enableShading=1, enableShadows=1, shadingModel=2, **kw)
    for shard in range(3):
        outs = []
        for opts in (dict(per_pixel=0), dict(per_pixel=1)):
            c = capi.Context(0, **opts)
            c.upload_scene(s)
            c.set_features(f)
            n = capi.shard_floats(w, h, shard, 3)
            buf = torch.zeros(n, dtype=torch.float32, device="cuda")
            c.render_device(cam, w, h, buf.data_ptr(), n, shard, 3, 0)
            c.sync(0)
            outs.append(buf.cpu().numpy())
            c.close()
        assert np.array_equal(outs[0].view(np.uint32), outs[1].view(np.uint32)) and outs[0].any(), shard
