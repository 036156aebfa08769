"""GPU parity: the HIP path (through the C ABI) against the CPU oracle."""
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


def _both(ctx, scn):
    osc = orc.OracleScene(scn)
    ctx.upload_scene(scn)
    return osc


@pytest.mark.parametrize("name,n", [("SingleTriangle", 5), ("Cube", 20), ("CornellBox", 100), ("Monkey", 300),
                                    ("Teapot", 300), ("Dragon", 300)])
def test_bvh_intersection_matches_oracle(ctx, name, n):
    """tests/bvh_tests.cpp:756-815 against the device path: same hit, ray.t and HitInfo
    with the acceleration structure on and off, bit for bit."""
    scn = common.scene(name)
    osc = _both(ctx, scn)
    rays = common.random_rays(max(n, 2000))
    for accel in (1, 0):
        f = common.features(enableAccelStructure=accel, enableNormalInterp=1, enableTextureMapping=1)
        if accel == 0 and len(osc.prims) > 20000:
            rays_ = rays[:300]
        else:
            rays_ = rays
        r_o, h_o, _ = osc.intersect(f, rays_)
        ctx.set_features(f)
        r_g, h_g = ctx.intersect(rays_)
        assert np.array_equal(h_o["hit"], h_g["hit"])
        assert np.array_equal(r_o["t"].view(np.uint32), r_g["t"].view(np.uint32))
        assert np.array_equal(h_o["primitive"], h_g["primitive"])
        for k in ("normal", "barycentricCoord", "texCoord"):
            assert np.array_equal(h_o[k].view(np.uint32), h_g[k].view(np.uint32)), k
        assert np.array_equal(h_o["material"], h_g["material"])


def test_pyramid_known_answers(ctx):
    """tests/bvh_tests.cpp:901-1008"""
    scn, _ = common.pyramid_scene()
    ctx.upload_scene(scn)
    rays = np.zeros(5, dtype=sio.RAY)
    rays["direction"] = (1, 0, 0)
    rays["t"] = common.FLT_MAX
    rays["origin"][:, 0] = [-10, -1, 0, 1, 10]
    expect = np.array([9.199999809265, 0.200000092387, 0.3249999880790, 2.25], dtype=np.float32)
    for accel in (1, 0):
        ctx.set_features(common.features(enableAccelStructure=accel))
        r, h = ctx.intersect(rays)
        assert list(h["hit"]) == [1, 1, 1, 1, 0]
        assert np.array_equal(r["t"][:4], expect)


CONFIGS = {
    "cornell_mirror": ("CornellBox", dict(enableShading=1, enableReflections=1, enableShadows=1, shadingModel=2), 128, 128),
    "teapot_blinn_shadows": ("Teapot", dict(enableShading=1, enableShadows=1, shadingModel=2), 320, 180),
    "monkey_interp_env": ("Monkey", dict(enableShading=1, enableShadows=1, shadingModel=1, enableNormalInterp=1,
                                         enableTextureMapping=1, enableBilinearTextureFiltering=1,
                                         enableEnvironmentMap=1), 160, 90),
    "cube_textured_env": ("CubeTextured", dict(enableShading=1, enableShadows=1, shadingModel=2, enableNormalInterp=1,
                                               enableTextureMapping=1, enableBilinearTextureFiltering=1,
                                               enableEnvironmentMap=1, enableReflections=1), 160, 90),
    "dragon_area_16spp": ("Dragon", dict(enableShading=1, enableShadows=1, enableReflections=1, shadingModel=2,
                                         numPixelSamples=16, numShadowSamples=4), 96, 54),
    "dragon_dof_transp": ("Dragon", dict(enableShading=1, enableShadows=1, enableReflections=1, shadingModel=2,
                                         numPixelSamples=4, numShadowSamples=2, enableTransparency=1,
                                         enableDepthOfField=1, numDepthOfFieldSamples=3,
                                         depthOfFieldDistance=2.5), 64, 36),
}


@pytest.mark.parametrize("key", list(CONFIGS))
def test_render_matches_oracle(ctx, key):
    name, fkw, w, h = CONFIGS[key]
    scn = common.scene(name, env_map=bool(fkw.get("enableEnvironmentMap")))
    if name == "Dragon":
        scn = common.scene(name)
        scn.lights = [sio.cornell_parallelogram_light()]
        if fkw.get("enableTransparency"):
            scn.meshes[0].material["transparency"] = 0.6
    osc = _both(ctx, scn)
    f = common.features(**fkw)
    cam = sio.default_camera(w, h)
    ref, st_o = osc.render(f, cam, w, h)
    ctx.set_features(f)
    img, st_g = ctx.render(cam, w, h)
    prim_o, t_o = osc.primary_hits(f, cam, w, h)
    maxabs, rmse = common.image_error(img, ref)
    exact = float((img.view(np.uint32) == ref.view(np.uint32)).mean())
    print(f"{key}: rays gpu {st_g['rays']} oracle {st_o['rays']} maxabs {maxabs:.3g} rmse {rmse:.3g} exact {exact:.4f} "
          f"kernel {st_g['kernel_ms']:.3f} ms")
    assert st_g["rays"] == st_o["rays"]
    assert maxabs <= 2e-5 and rmse <= 2e-6
