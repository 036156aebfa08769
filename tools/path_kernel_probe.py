"""The path kernel of the headline frame alone, with and without mirror bounces, and at a shorter depth cap:
how much of its time is the dependent chain of bounces.   python tools/path_kernel_probe.py"""
import importlib, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
capi = importlib.import_module("ray-tracer_amd.capi")
sio = importlib.import_module("ray-tracer_amd.scene_io")
scene, features, w, h, desc = bench.build_workload("dragon")
cam = sio.default_camera(w, h)
frame = torch.empty(w * h * 3, dtype=torch.float32, device="cuda")
for label, kw in (("as configured", {}), ("no reflections", dict(enableReflections=0)), ("depth cap 1", dict(maxRayDepth=1)), ("depth cap 2", dict(maxRayDepth=2))):
    f = features.copy()
    for k, v in kw.items():
        f[k] = v
    c = capi.Context(0)
    c.upload_scene(scene); c.set_features(f)
    for _ in range(3):
        c.render_device(cam, w, h, frame.data_ptr(), frame.numel(), 0, 1, 0)
    c.sync(0)
    for _ in range(5):
        c.render_device(cam, w, h, frame.data_ptr(), frame.numel(), 0, 1, 0)
    st = c.sync(0)
    n = max(1, st["launches"])
    print(json.dumps({"features": label, "rays": st["rays"], "camera+path_ms": round(st["path_ms"] / n, 3), "shadow_ms": round(st["shadow_ms"] / n, 3),
                      "frame_ms": round(st["kernel_ms"] / n, 3)}))
    c.close()
