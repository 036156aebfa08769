"""A model of a million triangles (the dragon stand-in's generator at 12x the resolution): BVH build on host and device,
upload, a small frame against the oracle and a 1080p frame's time.   python tools/big_scene.py [n_u n_v]"""
import importlib, json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")]
import common, orc
sio = common.sio
capi = importlib.import_module("ray-tracer_amd.capi")
n_u = int(sys.argv[1]) if len(sys.argv) > 1 else 2520
n_v = int(sys.argv[2]) if len(sys.argv) > 2 else 200
s = sio.Scene()
s.meshes.append(sio.dragon_standin(s, n_u=n_u, n_v=n_v))
s.lights = [sio.cornell_parallelogram_light()]
prims = s.primitives()
out = {"triangles": int(len(prims))}
t = time.perf_counter(); n_h, p_h, lv, lf = capi.bvh_build(prims); out["host_build_s"] = round(time.perf_counter() - t, 3)
c = capi.Context(0)
t = time.perf_counter(); n_d, p_d, lv2, lf2, hs = c.bvh_build_device(prims); out["device_build_s"] = round(time.perf_counter() - t, 3)
out["device_build_equal"] = bool(n_d.tobytes() == n_h.tobytes() and p_d.tobytes() == p_h.tobytes())
out["levels"], out["leaves"], out["host_sorted_primitives"] = lv, lf, hs
t = time.perf_counter(); c.upload_scene(s, n_h, p_h); out["upload_s"] = round(time.perf_counter() - t, 3)
f = common.features(enableShading=1, enableShadows=1, enableReflections=1, shadingModel=2, numPixelSamples=4, numShadowSamples=2, maxRayDepth=3)
c.set_features(f)
w, h = 96, 64
cam = sio.default_camera(w, h, distance=2.0)
got, sg = c.render(cam, w, h)
osc = orc.OracleScene(s, n_h, p_h)
want, so = osc.render(f, cam, w, h)
out["small_frame"] = {"rays": sg["rays"], "rays_equal": sg["rays"] == so["rays"], "bit_equal": float((got.view(np.uint32) == want.view(np.uint32)).mean())}
import torch
W, H = 1920, 1080
cam = sio.default_camera(W, H, distance=2.0)
frame = torch.empty(W * H * 3, dtype=torch.float32, device="cuda")
c.render_device(cam, W, H, frame.data_ptr(), frame.numel(), 0, 1, 0); c.sync(0)
t = time.perf_counter()
for _ in range(3):
    c.render_device(cam, W, H, frame.data_ptr(), frame.numel(), 0, 1, 0)
st = c.sync(0)
out["frame_1080p_4spp"] = {"ms": round((time.perf_counter() - t) / 3 * 1e3, 2), "rays": st["rays"], "Mrays_s": round(st["rays"] * 3 / (time.perf_counter() - t) / 1e6, 1)}
print(json.dumps(out))
